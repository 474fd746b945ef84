#!/usr/bin/env python3
"""RSNT (resonator algorithm) on the cfg2 buffer: 60 s stereo 48 kHz, hop 8192, P = 1024 -> 351 frames.  Whole render and K_A alone
(sgz_stage_mapped), achieved fp32 rate (8 flops per sample, axis point, vector and signal) and the oracle on a bounded sample."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from signalizer_amd import api, config, synth


def main():
    dev = torch.device("cuda", 0)
    out = {}
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None       # one window only (profiling runs)
    for name, over in (("hann", dict()), ("rect", dict(window_type=config.WIN_RECT)), ("blackman_harris", dict(window_type=config.WIN_BLACKMAN_HARRIS))):
        if only and name != only:
            continue
        cfg = config.spectrum_config(algorithm=config.ALGO_RSNT, **over)
        x = synth.gen(config.CFG2_SEED, 48000, int(config.CFG2_SECONDS * 48000), 2)
        xs = torch.from_numpy(x).to(dev)
        plan = api.Plan(cfg)
        if "--valu" in sys.argv:
            plan.set_option(api.OPT_MATRIX_RESONATOR, 0)
        if "--fp32-matrix" in sys.argv:
            plan.set_option(api.OPT_MATRIX_RESONATOR, 2)           # (the default since round 6)
        if "--bf16" in sys.argv:
            plan.set_option(api.OPT_MATRIX_RESONATOR, 1)           # the opt-in three-part bf16 kernel
        plan.upload()
        F = plan.num_frames(x.shape[1])
        rgba = torch.empty((F, plan.P, 4), dtype=torch.uint8, device=dev)
        V = plan.resonator()[0].shape[0]

        def timeit(fn, n=10):
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(n):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fn(); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            return float(np.median(ts))
        ms_render = timeit(lambda: plan.render(xs, rgba=rgba))
        ms_ka = timeit(lambda: plan.stage_mapped(xs))
        flops = 8.0 * F * cfg["hop"] * 2 * V * plan.P
        out[name] = {"frames": F, "vectors": V, "ms_render": ms_render, "ms_stage_mapped": ms_ka, "frames_per_s": F / ms_render * 1e3,
                     "realtime_factor": 60.0 / (ms_render * 1e-3), "tflops_fp32": flops / (ms_ka * 1e-3) / 1e12, "frac_of_157_tf": flops / (ms_ka * 1e-3) / 157.3e12}
    if "--cpu" in sys.argv:
        from oracle import pyoracle as po
        po.build()
        cfg = config.spectrum_config(algorithm=config.ALGO_RSNT)
        n = 8
        t0 = time.time()
        po.resonator_spectrogram(po.params_from_dict(cfg), x[:, :n * cfg["hop"]])
        dt = time.time() - t0
        out["cpu_oracle"] = {"frames": n, "seconds": dt, "frames_per_s": n / dt, "cores": 1}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
