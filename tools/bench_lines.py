#!/usr/bin/env python3
"""the line-graph display path per video frame (sgz_spectrum_render_lines): per call, 8 host blocks of 100 samples pushed (48 kHz / 60)
then one render of the ring's newest window for every pair; host clock, median over frames.   usage: bench_lines.py [N] [pairs]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from signalizer_amd import api, config, synth

def main():
    nums = [a for a in sys.argv[1:] if a.isdigit()]
    N = int(nums[0]) if nums else 4096
    pairs = int(nums[1]) if len(nums) > 1 else 1
    P = 1024
    extra = dict(algorithm=config.ALGO_RSNT) if "rsnt" in sys.argv else {}
    blk_len = 512 if "b512" in sys.argv else 100
    cfg = config.spectrum_config(window_size=N, hop=1024, axis_points=P, num_pairs=pairs, display_mode=config.DISPLAY_LINE_GRAPH, **extra)
    L = api.lib()
    c = api.config_from_dict(cfg)
    h = C.c_void_p()
    api.check(L.sgz_spectrum_create(C.byref(c), C.byref(h)))
    x = synth.gen(5, 48000, 48000 * 4, 2 * pairs)
    out = np.zeros((pairs, 2, P, 2), np.float32)
    pos = 0
    ts, tp = [], []
    for frame in range(400):
        t0 = time.perf_counter()
        for b in range(8):
            blk = np.ascontiguousarray(x[:, pos:pos + blk_len]); pos = (pos + blk_len) % (x.shape[1] - blk_len)
            ptrs = (C.c_void_p * (2 * pairs))(*[blk[i].ctypes.data for i in range(2 * pairs)])
            api.check(L.sgz_spectrum_push(h, ptrs, 2 * pairs, blk_len))
        t1 = time.perf_counter()
        api.check(L.sgz_spectrum_render_lines(h, None, out.ctypes.data_as(C.c_void_p)))
        t2 = time.perf_counter()
        if frame >= 50:
            tp.append(t1 - t0); ts.append(t2 - t1)
    print({"N": N, "pairs": pairs, "push_8_blocks_us": round(float(np.median(tp)) * 1e6, 1), "render_lines_us": round(float(np.median(ts)) * 1e6, 1),
           "render_lines_p90_us": round(float(np.percentile(ts, 90)) * 1e6, 1)})
    L.sgz_spectrum_destroy(h)

if __name__ == "__main__":
    main()
