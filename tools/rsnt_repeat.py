#!/usr/bin/env python3
"""Run-to-run determinism of one fuzz_rsnt.py case: the case's configuration (seed, index) is rendered `n` times on fresh and on reused plans;
every stage_mapped / render output must equal the first run's bit for bit.      usage: rsnt_repeat.py <seed> <case> [n]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from signalizer_amd import api, config as cf, synth

seed, want = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 500
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
for case in range(want + 1):
    mode = int(rng.integers(0, 8))
    d = cf.spectrum_config(algorithm=cf.ALGO_RSNT, channel_mode=mode, window_type=int(rng.integers(0, 13)),
                           window_size=int(rng.choice([512, 4096, 32768])), hop=int(rng.choice([int(rng.integers(40, 3000)), 1024, 2048, 3072])),
                           axis_points=int(rng.integers(2, 1500)), num_pairs=int(rng.integers(1, 4)), free_q=int(rng.integers(0, 2)),
                           view_scaling=int(rng.integers(0, 2)), sample_rate=float(rng.choice([44100.0, 48000.0, 96000.0])),
                           view_left=float(rng.uniform(0, 0.3)), view_right=float(rng.uniform(0.5, 1.0)),
                           pole=(float(rng.uniform(0.5, 0.999)), float(rng.uniform(0.5, 0.999))))
    F = int(rng.integers(1, 20))
    x = synth.gen(int(rng.integers(1, 1000)), int(d["sample_rate"]), F * d["hop"] + int(rng.integers(0, d["hop"])), 2 * d["num_pairs"])
print("case", want, "frames", F, {k: d[k] for k in ("window_size", "hop", "axis_points", "channel_mode", "num_pairs", "window_type", "free_q")})
xs = torch.from_numpy(x).to(dev)
plan = api.Plan(d).upload()
m0 = plan.stage_mapped(xs).cpu().numpy()
r0 = plan.render(xs).cpu().numpy()
bad_m = bad_r = 0
for it in range(n):
    p = plan if it % 4 else api.Plan(d).upload()                  # every fourth run on a fresh plan
    m = p.stage_mapped(xs).cpu().numpy()
    r = p.render(xs).cpu().numpy()
    if not np.array_equal(m, m0, equal_nan=True):
        bad_m += 1
        w = np.argwhere(m != m0)
        print("  run", it, "mapped differs in", len(w), "entries, first", tuple(int(v) for v in w[0]), float(m[tuple(w[0])]), "vs", float(m0[tuple(w[0])]), "fresh plan" if it % 4 == 0 else "reused plan")
    if not np.array_equal(r, r0):
        bad_r += 1
        w = np.argwhere(r != r0)
        print("  run", it, "image differs in", len(w), "bytes, first", tuple(int(v) for v in w[0]), "fresh plan" if it % 4 == 0 else "reused plan")
print(f"runs whose magnitudes / image differ from the first run's: {bad_m} / {bad_r} of {n}")
