#!/usr/bin/env python3
"""Does any result depend on what freshly allocated device memory (or a caller's output buffer) held before the call?  A fresh process
sees zeroed device memory; a host that has been running for hours does not -- a read of an uninitialised scratch word, or an output byte
the kernels never write, shows only there.  tools/poison_probe.py asked this of RSNT (round 5); this is the same question of everything:
  * random FFT configurations (tests/fuzzcfg.random_config, every K_A / K_B path) rendered with line results and a carried state on a
    plan created right after the driver's free memory was filled with a byte pattern, into outputs pre-filled with the pattern;
  * the spectrum stream, the Oscilloscope and the Vectorscope: handle created after the poisoning, fed, every read call recorded.
Everything is run twice, with 0x00 and with 0xFF (= NaNs) as the pattern; any difference between the two runs is a dependence.
usage: poison_probe_all.py [spectrum cases] [seed]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from fuzzcfg import random_config

from signalizer_amd import api, config as cf, synth

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
dev = torch.device("cuda", 0)


def poison(byte):
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


def filled(shape, dtype, byte):
    t = torch.empty(shape, dtype=torch.uint8 if dtype == torch.uint8 else dtype, device=dev)
    t.view(torch.uint8).fill_(byte)
    return t


def spectrum(byte, cases, seed):
    rng = np.random.default_rng(seed)
    res = []
    for case in range(cases):
        cfg = random_config(rng, wild=bool(case % 3 == 2))
        if case % 7 == 6:
            cfg["algorithm"] = cf.ALGO_RSNT
            cfg["window_size"] = int(rng.choice([512, 4096]))
            cfg["hop"] = int(rng.choice([1024, 700]))
        W, hop = cfg["window_size"], cfg["hop"]
        frames = int(rng.integers(1, 14))
        S = W + (frames - 1) * hop + int(rng.integers(0, hop))
        x = torch.from_numpy(synth.gen(300 + case, int(cfg["sample_rate"]), S, 2 * cfg["num_pairs"])).to(dev)
        torch.cuda.synchronize()
        poison(byte)
        try:
            plan = api.Plan(cfg).upload()
        except api.SgzError:
            res.append(None)
            continue
        F = plan.num_frames(S)
        rgba = filled((F, plan.P, 4), torch.uint8, byte)
        lines = filled((F, cfg["num_pairs"], 2, plan.P, 2), torch.float32, byte)
        state = torch.zeros((cfg["num_pairs"], 2, plan.P, 2), dtype=torch.float32, device=dev)      # (an input: must be defined)
        plan.render(x, rgba=rgba, lines=lines, state=state)
        poison(byte)
        rgba2 = filled((F, plan.P, 4), torch.uint8, byte)
        plan.render(x, rgba=rgba2)                                   # the image-only K_B forms, on the plan's grown scratch
        mapped = plan.stage_mapped(x)
        torch.cuda.synchronize()
        res.append((cfg, [t.cpu().numpy().copy() for t in (rgba, lines.view(torch.int32), state.view(torch.int32), rgba2, mapped.view(torch.int32))]))
        plan.close()
    return res


def spectrum_stream(byte):
    L = api.lib()
    out = []
    for cfg in (cf.spectrum_config(window_size=4096, hop=1024, axis_points=300), cf.spectrum_config(window_size=2048, hop=512, axis_points=300, channel_mode=cf.CH_PHASE),
                cf.spectrum_config(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1024, axis_points=300)):
        x = synth.gen(8, 48000, 24 * 512, 2)
        poison(byte)
        c = api.config_from_dict(cfg)
        h = C.c_void_p()
        api.check(L.sgz_spectrum_create(C.byref(c), C.byref(h)))
        cols = []
        buf = np.full((300, 4), byte, np.uint8)
        ap = C.c_uint32(0)
        for b in range(x.shape[1] // 512):
            blk = np.ascontiguousarray(x[:, b * 512:(b + 1) * 512])
            ptrs = (C.c_void_p * 2)(blk[0].ctypes.data, blk[1].ctypes.data)
            while L.sgz_spectrum_push(h, ptrs, 2, 512) == api.SGZ_BUSY:
                pass
        L.sgz_spectrum_flush.argtypes = [C.c_void_p]
        api.check(L.sgz_spectrum_flush(h))
        t0 = time.time()
        while time.time() - t0 < 5.0 and len(cols) < x.shape[1] // cfg["hop"]:
            if L.sgz_spectrum_pop_column(h, buf.ctypes.data_as(C.c_void_p), C.byref(ap)) == api.SGZ_OK:
                cols.append(buf.copy())
            else:
                time.sleep(0.001)
        L.sgz_spectrum_destroy(h)
        out.append(np.stack(cols))
    return out


def scope(byte):
    SR = 192000.0
    t = np.arange(40000) / SR
    x = np.stack([0.6 * np.sin(2 * np.pi * 441.7 * t), 0.5 * np.sin(2 * np.pi * 577.0 * t + 0.3)]).astype(np.float32)
    out = []
    for over in (dict(), dict(trigger_mode=1), dict(trigger_mode=3, interpolation=1), dict(envelope_mode=2, interpolation=2), dict(window_size=480.3, trigger_threshold=0.0)):
        cfg = dict(sample_rate=SR, window_size=19200.0, num_channels=2, trigger_mode=4, channel_mode=0, envelope_mode=1, interpolation=3,
                   max_block=4096, trigger_threshold=0.05, trigger_channel=1.0, envelope_window=0.3)
        cfg.update(over)
        poison(byte)
        try:
            d = api.Scope(**cfg)
        except api.SgzError:
            continue
        for p in range(0, x.shape[1], 1000):
            while d.push(x[:, p:p + 1000]) != api.SGZ_OK:
                pass
        st = d.state()
        g = d.peak_filter(1.0 / 60.0)
        view = api.ScopeView(cfg["window_size"], 0.0, 1.0, 1.0, 1200, 0)
        n = api.lib().sgz_scope_vertex_count(d.h, C.byref(view))
        xyz = np.full((n, 3), 0, np.float32); xyz.view(np.uint8)[:] = byte
        rgba = np.full((n, 4), byte, np.uint8)
        xyz, rgba = d.vertices(view, 0, 0, out=(xyz, rgba))
        front, cur = d.front(0)
        out.append((sorted(st.items()), np.float64(g), xyz.view(np.uint32).copy(), rgba.copy(), front.view(np.uint32).copy(), cur))
        d.close()
    return out


def vector(byte):
    x = synth.gen(4, 96000, 30000, 8)
    out = []
    for env_mode, fade in ((0, 1), (1, 1), (2, 0)):
        poison(byte)
        d = api.Vector(sample_rate=96000.0, num_channels=8, window_size=9600, envelope_mode=env_mode, lanes=8, fade_history=fade,
                       max_block=4096, envelope_window=0.3, stereo_window=0.05)
        for p in range(0, x.shape[1], 777):
            while d.push(x[:, p:p + 777]) != api.SGZ_OK:
                pass
        g = d.peak_filter(1.0 / 60.0)
        xyz = np.zeros((4, 9600, 3), np.float32); xyz.view(np.uint8)[:] = byte
        rgb = np.zeros((4, 9600, 3), np.float32); rgb.view(np.uint8)[:] = byte
        d.vertices_all(out=(xyz, rgb))
        mem, cur = d.history(3)
        out.append((np.float64(g), xyz.view(np.uint32).copy(), rgb.view(np.uint32).copy(), mem.view(np.uint32).copy(), cur))
        d.close()
    return out


def same(a, b):
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(same(u, v) for u, v in zip(a, b))
    if isinstance(a, np.ndarray):
        return a.shape == b.shape and np.array_equal(a, b, equal_nan=True) if a.dtype.kind != "f" else np.array_equal(a.view(np.uint8), b.view(np.uint8))
    if isinstance(a, dict):
        return a == b
    if isinstance(a, float) or isinstance(a, np.floating):
        return np.float64(a).tobytes() == np.float64(b).tobytes()
    return a == b


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
A, B = spectrum(0x00, cases, seed), spectrum(0xFF, cases, seed)
names = ("image", "line results", "carried state", "image (image-only render)", "mapped magnitudes")
ran = 0
for i, (a, b) in enumerate(zip(A, B)):
    if a is None or b is None:
        continue
    ran += 1
    diff = [names[k] for k in range(5) if not np.array_equal(a[1][k], b[1][k])]
    if diff:
        bad += 1
        print("DIFF spectrum case", i, diff, {k: a[0][k] for k in ("algorithm", "window_size", "hop", "axis_points", "channel_mode", "num_pairs", "bin_interp")}, flush=True)
print(f"spectrum: {ran} configurations rendered twice (plan scratch and outputs 0x00 / 0xFF before the call): {bad} depend on it", flush=True)
for name, fn in (("spectrum stream", spectrum_stream), ("oscilloscope", scope), ("vectorscope", vector)):
    a, b = fn(0x00), fn(0xFF)
    d = sum(0 if same(u, v) else 1 for u, v in zip(a, b))
    bad += d
    print(f"{name}: {len(a)} handles run twice: {d} depend on it", flush=True)
print("dependences on uninitialised memory:", bad)
sys.exit(1 if bad else 0)
