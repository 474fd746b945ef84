"""Seeded random parity sweep (the always-on part of tools/fuzz_parity.py): GPU render through the C ABI against the oracle over
random configurations -- every K_A path, all eight channel modes, interpolation, view scaling / zoom, window functions, heights,
pairs.  This sweep is what found the mono modes' complex csf entries and the Phase last-pixel case."""
import numpy as np
import pytest

from fuzzcfg import random_config
from parity_chain import check_render
from signalizer_amd import api, config, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,wild", [(1, False), (2, False), (3, False), (11, True), (12, True)])
def test_random_configurations_match_the_oracle(gpu, oracle, seed, wild):
    import torch
    po = oracle
    rng = np.random.default_rng(seed)
    bad = []
    for it in range(40):
        cfg = random_config(rng, wild)
        frames = int(rng.integers(1, 12))
        W, hop = cfg["window_size"], cfg["hop"]
        S = W + (frames - 1) * hop + int(rng.integers(0, hop))
        x = synth.gen(100 + it, cfg["sample_rate"], S, 2 * cfg["num_pairs"])
        try:
            plan = api.Plan(cfg)
        except api.SgzError:
            continue                                       # a configuration the reference's assertions reject as well
        plan.upload()
        # the parity chain (tests/parity_chain.py): mapped pixels within the FFT tolerance of the oracle's (Phase near-tie flips
        # verified one by one against the oracle's own bins), colour bytes EXACT given the mapped pixels
        problems, stats = check_render(po, plan, cfg, x, gpu, want_lines=(it % 4 == 0))
        if problems:
            bad.append((it, plan.N, plan.path, cfg["channel_mode"], problems[:3], stats))
    assert not bad, bad


@pytest.mark.parametrize("tool,count", [("fuzz_stages.py", 40), ("fuzz_realtime.py", 30), ("fuzz_scope.py", 30), ("fuzz_rsnt.py", 40)])
def test_stage_realtime_and_scope_sweeps(gpu, tool, count):
    """the other seeded sweeps (tools/): bit-exact stages, the per-block path and split renders, the Oscilloscope / Vectorscope
    kernels -- each prints one line per case and exits non-zero on any mismatch"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", tool), str(count), "1"], capture_output=True, text=True, timeout=900)
    bad = [l for l in r.stdout.splitlines() if " BAD " in l or l.startswith("BAD") or l.startswith("EXC")]
    assert r.returncode == 0 and not bad, (bad[:5], r.stderr[-500:])
    # ... and the sweep must have RUN: every tool ends with "bad: <b> of <n>" -- a crash that prints nothing and exits 0 fails here
    import re
    tally = [re.match(r"bad: (\d+) of (\d+)", l) for l in r.stdout.splitlines()]
    tally = [m for m in tally if m]
    assert len(tally) == 1, (r.stdout[-400:], r.stderr[-400:])
    assert int(tally[0].group(1)) == 0 and int(tally[0].group(2)) >= count, tally[0].group(0)
