#!/usr/bin/env python3
"""Does any result depend on what freshly allocated device memory held?  Fills the allocator's free memory with a byte pattern (hipMalloc -> hipMemset
-> hipFree of many blocks, in this process) in front of every case of a fuzz_rsnt.py seed and compares the device's outputs between two patterns
(0x00 and 0xFF = NaNs).  A difference is a read of uninitialised memory.     usage: poison_probe.py <cases> <seed>"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from signalizer_amd import api, config as cf, synth

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]


def poison(byte):
    """blocks of the sizes a plan allocates, from a few KB to 256 MB: what the next hipMalloc of such a size returns has this pattern"""
    held = []
    for size in [1 << k for k in range(10, 29)] * 3:
        p = C.c_void_p()
        if hip.hipMalloc(C.byref(p), size) != 0:
            break
        hip.hipMemset(p, byte, size)
        held.append(p)
    hip.hipDeviceSynchronize()
    for p in held:
        hip.hipFree(p)


def fresh_bytes(size):
    """what a fresh allocation of `size` bytes holds (fraction of 0xFF bytes)"""
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    p = C.c_void_p()
    hip.hipMalloc(C.byref(p), size)
    host = np.empty(size, np.uint8)
    hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), p, size, 2)
    hip.hipFree(p)
    return float((host == 0xFF).mean())


poison(0xFF)
print("fresh allocations after poisoning with 0xFF hold 0xFF in", {sz: round(fresh_bytes(sz), 3) for sz in (4096, 1 << 16, 1 << 20, 1 << 24, 3 << 20, 100000)}, "of their bytes")
cases, seed = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
bad = 0
for pattern_pair in [(0x00, 0xFF)]:
    outs = {}
    for byte in pattern_pair:
        rng = np.random.default_rng(seed)
        res = []
        for case in range(cases):
            mode = int(rng.integers(0, 8))
            d = cf.spectrum_config(algorithm=cf.ALGO_RSNT, channel_mode=mode, window_type=int(rng.integers(0, 13)),
                                   window_size=int(rng.choice([512, 4096, 32768])), hop=int(rng.choice([int(rng.integers(40, 3000)), 1024, 2048, 3072])),
                                   axis_points=int(rng.integers(2, 1500)), num_pairs=int(rng.integers(1, 4)), free_q=int(rng.integers(0, 2)),
                                   view_scaling=int(rng.integers(0, 2)), sample_rate=float(rng.choice([44100.0, 48000.0, 96000.0])),
                                   view_left=float(rng.uniform(0, 0.3)), view_right=float(rng.uniform(0.5, 1.0)),
                                   pole=(float(rng.uniform(0.5, 0.999)), float(rng.uniform(0.5, 0.999))))
            F = int(rng.integers(1, 20))
            x = synth.gen(int(rng.integers(1, 1000)), int(d["sample_rate"]), F * d["hop"] + int(rng.integers(0, d["hop"])), 2 * d["num_pairs"])
            xs = torch.from_numpy(x).to(dev)
            torch.cuda.synchronize()
            poison(byte)
            plan = api.Plan(d).upload()
            got = plan.stage_mapped(xs).cpu().numpy()
            poison(byte)
            rgba = plan.render(xs).cpu().numpy()
            res.append((got, rgba, d, F))
            del plan
        outs[byte] = res
    a, b = outs[pattern_pair[0]], outs[pattern_pair[1]]
    for case, ((g0, r0, d, F), (g1, r1, _, _)) in enumerate(zip(a, b)):
        same_m = np.array_equal(g0, g1, equal_nan=True)
        same_r = np.array_equal(r0, r1)
        if not (same_m and same_r):
            bad += 1
            print("DIFF case", case, "mapped same:", same_m, "image same:", same_r, "frames", F, {k: d[k] for k in ("window_size", "hop", "axis_points", "channel_mode", "num_pairs", "window_type")})
print(f"cases whose outputs depend on the contents of fresh device memory: {bad} of {cases}")
