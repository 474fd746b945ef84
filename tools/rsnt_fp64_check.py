#!/usr/bin/env python3
"""Which of the two fp32 evaluations of an RSNT render is closer to the exact one?  For one fuzz_rsnt.py case (seed, index): the device's
windowed magnitudes (chained frames, matrix-core block sums at hops that are multiples of 1024) and the oracle's (sequential fp32 recurrence)
against an fp64 walk of the SAME resonators (the plan's fp32 poles, gains and window weights taken as exact): per frame of one pair / signal,
the largest and the RMS error of either, in units of the test's bar.      usage: rsnt_fp64_check.py <seed> <case> [pair] [signal]"""
import ctypes as C
import os
import sys

import numpy as np
import torch
from scipy.signal import lfilter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po
from signalizer_amd import api, config as cf, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_resonator import _planes, CHAIN_TOL, STATE_K, EPS

seed, want = int(sys.argv[1]), int(sys.argv[2])
pair = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sig = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rng = np.random.default_rng(seed)
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
assert mode != cf.CH_PHASE, "magnitude planes only"
po.build()
p = po.params_from_dict(d)
P, hop = d["axis_points"], d["hop"]
print("case", want, "frames", F, {k: d[k] for k in ("window_size", "hop", "axis_points", "channel_mode", "num_pairs", "window_type", "free_q", "sample_rate")})
plan = api.Plan(d).upload()
got = plan.stage_mapped(torch.from_numpy(x).cuda()).cpu().numpy()
r = po.resonator_spectrogram(p, x, want_mapped=True, want_scale=True)
ref = _planes(r["mapped"], mode, P)
coeff, gain, weights = po.resonator_map(p)[:3]
coeff = np.asarray(coeff); V = coeff.shape[0]
# the signal the resonators of (pair, sig) see, as the reference's dispatch mixes it (fp32, like both evaluations)
L, R = np.ascontiguousarray(x[2 * pair]), np.ascontiguousarray(x[2 * pair + 1])
w0, w1 = np.zeros_like(L), np.zeros_like(L)
nsig = po.lib().sgzo_resonator_dispatch(C.c_uint32(mode), L.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p), C.c_size_t(L.size),
                                       w0.ctypes.data_as(C.c_void_p), w1.ctypes.data_as(C.c_void_p))
xin = (w0 if sig == 0 else w1).astype(np.float64)
n = F * hop
truth = np.zeros((F, P))
for i in range(P):
    acc = np.zeros(F, np.complex128)
    for v in range(V):
        c = complex(float(coeff[v, i].real), float(coeff[v, i].imag)) if np.iscomplexobj(coeff) else complex(float(coeff[v, i, 0]), float(coeff[v, i, 1]))
        s = lfilter([1.0], [1.0, -c], xin[:n].astype(np.complex128))
        acc += float(weights[v]) * s[hop - 1::hop][:F]
    truth[:, i] = np.abs(acc) * float(gain[i])
state_tol = STATE_K * EPS * np.sqrt(1.0 / np.maximum(np.asarray(gain, np.float64), 1e-12))
print("frame  bar-units: device max / rms | oracle max / rms   (bar = CHAIN_TOL x frame top + STATE_K eps sqrt(1/(1-r)) x scale)")
for f in range(F):
    top = max(float(np.max(np.abs(ref[f, pair]))), 1e-30)
    bar = CHAIN_TOL * top + state_tol * r["scale"][f, pair, sig]
    ed = np.abs(got[f, pair, sig] - truth[f]) / bar
    eo = np.abs(ref[f, pair, sig] - truth[f]) / bar
    dd = np.abs(got[f, pair, sig].astype(np.float64) - ref[f, pair, sig]) / bar
    print(f"{f:5d}   device {ed.max():6.3f} / {np.sqrt((ed ** 2).mean()):6.3f} | oracle {eo.max():6.3f} / {np.sqrt((eo ** 2).mean()):6.3f} | device - oracle {dd.max():6.3f} at pixel {int(dd.argmax())}")
