"""Seeded random parity sweep (the always-on part of tools/fuzz_parity.py): GPU render through the C ABI against the oracle over
random configurations -- every K_A path, all eight channel modes, interpolation, view scaling / zoom, window functions, heights,
pairs.  This sweep is what found the mono modes' complex csf entries and the Phase last-pixel case."""
import numpy as np
import pytest

from fuzzcfg import random_config
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
        ref = po.spectrogram(po.params_from_dict(cfg), x)["rgba"]
        got = plan.render(torch.from_numpy(x).to(gpu)).cpu().numpy()
        d = np.abs(got.astype(int) - ref.astype(int))
        phase = cfg["channel_mode"] == config.CH_PHASE      # the cancellation ratio amplifies FFT rounding
        # (Phase: isolated arg-max near-ties between adjacent bins may flip with the FFT's rounding -- see tools/fuzz_parity.py)
        if got.shape != ref.shape or (d > 0).sum() > max(2, (2e-2 if phase else 5e-3) * d.size) or \
                (d.max() > 1 if not phase else (d > 2).sum() > max(8, 1e-3 * d.size)):
            bad.append((it, plan.N, plan.path, cfg["channel_mode"], int(d.max()), float((d > 0).mean())))
    assert not bad, bad


@pytest.mark.parametrize("tool,count", [("fuzz_stages.py", 40), ("fuzz_realtime.py", 30), ("fuzz_scope.py", 30)])
def test_stage_realtime_and_scope_sweeps(gpu, tool, count):
    """the other seeded sweeps (tools/): bit-exact stages, the per-block path and split renders, the Oscilloscope / Vectorscope
    kernels -- each prints one line per case and exits non-zero on any mismatch"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", tool), str(count), "1"], capture_output=True, text=True, timeout=900)
    bad = [l for l in r.stdout.splitlines() if " BAD " in l]
    assert r.returncode == 0 and not bad, (bad[:5], r.stderr[-500:])
