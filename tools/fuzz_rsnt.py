#!/usr/bin/env python3
"""random RSNT (resonator algorithm) configurations against oracle/resonator.c: windowed magnitudes within the chain tolerance (frame 0
bit-exact wherever it continues the state sample by sample: one-frame launches, hops the matrix kernels do not take), decay / dB / colour byte for byte given the device's own magnitudes.   usage: fuzz_rsnt.py <cases> <seed>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po
from signalizer_amd import api, config as cf, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_resonator import _planes, check_planes

cases, seed = int(sys.argv[1]), int(sys.argv[2])
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
po.build()
bad = 0
worst = 0.0
for case in range(cases):
    mode = int(rng.integers(0, 8))
    d = cf.spectrum_config(algorithm=cf.ALGO_RSNT, channel_mode=mode, window_type=int(rng.integers(0, 13)),
                           window_size=int(rng.choice([512, 4096, 32768])), hop=int(rng.choice([int(rng.integers(40, 3000)), 1024, 2048, 3072])),   # (multiples of 1024: the matrix-core kernel)
                           axis_points=int(rng.integers(2, 1500)), num_pairs=int(rng.integers(1, 4)), free_q=int(rng.integers(0, 2)),
                           view_scaling=int(rng.integers(0, 2)), sample_rate=float(rng.choice([44100.0, 48000.0, 96000.0])),
                           view_left=float(rng.uniform(0, 0.3)), view_right=float(rng.uniform(0.5, 1.0)),
                           pole=(float(rng.uniform(0.5, 0.999)), float(rng.uniform(0.5, 0.999))))
    F = int(rng.integers(1, 20))
    x = synth.gen(int(rng.integers(1, 1000)), int(d["sample_rate"]), F * d["hop"] + int(rng.integers(0, d["hop"])), 2 * d["num_pairs"])
    p = po.params_from_dict(d)
    if only >= 0 and case != only:
        continue
    try:
        plan = api.Plan(d).upload()
        xs = torch.from_numpy(x).to(dev)
        got = plan.stage_mapped(xs).cpu().numpy()
        r = po.resonator_spectrogram(p, x, want_mapped=True, want_scale=True)
        ref = _planes(r["mapped"], mode, d["axis_points"])
        ok = got.shape == ref.shape
        if not ok:
            print("   shape", got.shape, ref.shape)
        if F == 1 or d["hop"] % 1024:                             # (a launch of several frames on the matrix cores starts every frame from rest)
            same0 = np.array_equal(got[0], ref[0])
            if not same0:
                print("   frame 0 differs from the oracle's (sample-by-sample continuation): frames", F, "hop", d["hop"], "entries", int((got[0] != ref[0]).sum()),
                      "max |diff|", float(np.nanmax(np.abs(got[0] - ref[0]))))
            ok &= same0
        problems, worst_case = check_planes(got, ref, r["scale"], mode, po.resonator_map(p)[1])
        ok &= not problems
        if problems:
            f_, c_, s_, ratio = max(problems, key=lambda t: t[3])
            e = np.abs(got[f_, c_, s_] - ref[f_, c_, s_]); i_ = int(np.argmax(e / np.maximum(r["scale"][f_, c_, s_ if mode != cf.CH_PHASE else 0], 1e-37)))
            print("   worst", (f_, c_, s_, ratio), "pixel", i_, "err", float(e[i_]), "ref", float(ref[f_, c_, s_, i_]), "scale", float(r["scale"][f_, c_, min(s_, 1), i_]),
                  "gain", float(po.resonator_map(p)[1][i_]), "frames", F, "top", float(np.max(np.abs(ref[f_, c_]))))
        worst = max(worst, worst_case)
        rgba = plan.render(xs).cpu().numpy()
        want, _ = po.decay_colour(p, got)
        if not np.array_equal(rgba, want):
            dd = np.abs(rgba.astype(int) - want.astype(int))
            print("   image differs from the oracle's K_B on the device's own magnitudes: bytes", int((dd > 0).sum()), "max", int(dd.max()), "frames", F,
                  "first at", tuple(int(v) for v in np.argwhere(dd > 0)[0]))
        ok &= np.array_equal(rgba, want)
    except Exception as e:                                         # noqa: BLE001
        ok = False
        print("EXC", e)
    if not ok:
        bad += 1
        print("BAD", case, d)
print(f"bad: {bad} of {cases}   (largest |error| / bar: {worst:.3f})")
sys.exit(1 if bad else 0)
