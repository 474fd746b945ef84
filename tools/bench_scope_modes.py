#!/usr/bin/env python3
"""per rendered frame (host clock): sgz_scope_analyse in Spectral mode, peak filter, one channel's vertices -- by trigger mode and
interpolation, 192 kHz stereo, 19 200-sample window (cfg3's sizes)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from signalizer_amd import api, synth
sr, W = 192000.0, 19200
x = synth.gen(31, int(sr), 3200 * 40, 2)
L = api.lib()
for name, kw in [("ZeroCrossing, Lanczos", dict(trigger_mode=4, interpolation=3)), ("Spectral, Lanczos", dict(trigger_mode=1, interpolation=3)),
                 ("Spectral, Linear", dict(trigger_mode=1, interpolation=1)), ("ZeroCrossing, Linear", dict(trigger_mode=4, interpolation=1)),
                 ("ZeroCrossing, Rectangular", dict(trigger_mode=4, interpolation=2)), ("Window, Lanczos", dict(trigger_mode=2, interpolation=3))]:
    cfg = dict(sample_rate=sr, window_size=float(W), num_channels=2, channel_mode=0, envelope_mode=2, max_block=512, trigger_threshold=0.05,
               trigger_channel=1.0, envelope_window=0.3)
    cfg.update(kw)
    h = api.Scope(**cfg)
    view = api.ScopeView(float(W), 0.0, 1.0, 1.0, 8 * W + 1, 0)
    ts = {"analyse": [], "vertices": []}
    out = None
    for frame in range(60):
        a = (frame % 40) * 3200
        for pos in range(a, a + 3200, 512):
            while h.push(x[:, pos:min(pos + 512, a + 3200)]) == api.SGZ_BUSY: pass
        h.flush()
        tr = api.TriggerState() if hasattr(api, "TriggerState") else None
        t0 = time.perf_counter()
        if tr is not None: api.check(L.sgz_scope_analyse(h.h, 0, 0, C.byref(tr)))
        t1 = time.perf_counter()
        h.peak_filter(1 / 60, 8)
        if out is None:
            n = L.sgz_scope_vertex_count(h.h, C.byref(view))
            out = (torch.empty((n, 3), dtype=torch.float32).pin_memory().numpy(), torch.empty((n, 4), dtype=torch.uint8).pin_memory().numpy())
        t2 = time.perf_counter()
        xyz, _ = h.vertices(view, 0, 0, out=out)
        t3 = time.perf_counter()
        if frame >= 10: ts["analyse"].append(t1 - t0); ts["vertices"].append(t3 - t2)
    print(f"{name:28s} analyse {np.median(ts['analyse']) * 1e6:7.1f} us   vertices ({xyz.shape[0]}) {np.median(ts['vertices']) * 1e6:7.1f} us")
    h.close()
