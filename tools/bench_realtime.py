"""Real-time per-block path (sgz_spectrum_push / sgz_spectrum_pop_column, single GPU): host-side cost of one audio callback
and the latency from the push that completes a frame to the popped RGBA8 column.  cfg2 settings (N = 32768, hop 8192, P = 1024),
512-sample callbacks at 48 kHz (one callback = 10.7 ms of audio)."""
import sys, os, time, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from signalizer_amd import api, config, synth

def run(cfg, block, nblocks=400):
    c = api.config_from_dict(cfg)
    h = C.c_void_p()
    L = api.lib()
    api.check(L.sgz_spectrum_create(C.byref(c), C.byref(h)))
    P, hop = cfg["axis_points"], cfg["hop"]
    x = synth.gen(8, cfg["sample_rate"], block * nblocks, 2 * cfg["num_pairs"])
    col = np.zeros((P, 4), np.uint8)
    ap = C.c_uint32(0)
    push_plain, push_frame, latency = [], [], []
    fed = 0
    try:
        for b in range(nblocks):
            blk = np.ascontiguousarray(x[:, b * block:(b + 1) * block])
            ptrs = (C.c_void_p * blk.shape[0])(*[blk[i].ctypes.data for i in range(blk.shape[0])])
            fires = (fed + block) // hop > fed // hop
            t0 = time.perf_counter()
            api.check(L.sgz_spectrum_push(h, ptrs, blk.shape[0], block))
            t1 = time.perf_counter()
            fed += block
            (push_frame if fires else push_plain).append((t1 - t0) * 1e6)
            if fires:
                while L.sgz_spectrum_pop_column(h, col.ctypes.data_as(C.c_void_p), C.byref(ap)) != api.SGZ_OK:
                    pass
                latency.append((time.perf_counter() - t0) * 1e6)
    finally:
        L.sgz_spectrum_destroy(h)
    med = lambda v: float(np.median(v[3:])) if len(v) > 3 else float("nan")
    print(json.dumps({"window": cfg["window_size"], "hop": hop, "block": block, "pairs": cfg["num_pairs"],
                      "push_us_no_frame": med(push_plain), "push_us_frame": med(push_frame),
                      "push_to_column_us": med(latency), "frames": len(latency)}, default=str))

run(config.cfg2(), 512)
run(config.spectrum_config(window_size=4096, hop=1024), 256)
run(config.cfg5(pairs=4), 1024, nblocks=200)
