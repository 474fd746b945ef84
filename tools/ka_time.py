"""K_A timing loop for kernel work: average launch duration (HIP events on the launch stream, sustained clock) of sgz_stage_mapped at
cfg2 (348 frames, 2 rounds on 256 CUs) and at a tail-free size (8 pairs x 348 = 2784 tasks), plus the whole step.
usage: [SGZ_WHOLE_FRAME=1] ka_time.py [iters]     (SGZ_WHOLE_FRAME=1: plan option SGZ_OPT_CHANNEL_SPLIT = 0, the whole-frame / halves kernels)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth

def timeit(fn, iters, spin_ms=60.0, batches=5):
    """(median, min) over `batches` of the average duration (us) of `iters` calls back to back between ONE event pair, after `spin_ms` of
    untimed calls: the device's clock settles only under sustained load (tools/clock_ramp_probe.py; an event pair around every call
    also adds ~2 us to each)"""
    import time
    hip = ctypes.CDLL("libamdhip64.so")
    stream = torch.cuda.current_stream().cuda_stream
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    hip.hipEventCreate(ctypes.byref(e0)); hip.hipEventCreate(ctypes.byref(e1))
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < spin_ms:
        for _ in range(32): fn()
        torch.cuda.synchronize()
    tot = []
    for _ in range(batches):
        hip.hipEventRecord(e0, ctypes.c_void_p(stream))
        for _ in range(iters): fn()
        hip.hipEventRecord(e1, ctypes.c_void_p(stream))
        hip.hipEventSynchronize(e1)
        ms = ctypes.c_float(); hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1); tot.append(ms.value * 1e3 / iters)
    return float(np.median(tot)), float(np.min(tot))

def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    out = {}
    for name, pairs in (("cfg2_348", 1), ("notail_2784", 8), ("cfg5_20s", 32)):
        cfg = config.cfg2() if name != "cfg5_20s" else config.cfg5(); cfg["num_pairs"] = pairs
        sr = int(cfg["sample_rate"])
        S = int(config.CFG2_SECONDS * 48000) if name != "cfg5_20s" else 20 * sr
        x = torch.from_numpy(synth.gen(2, sr, S, 2 * pairs)).cuda()
        if os.environ.get('SGZ_ZERO_INPUT') == '1': x.zero_()      # (clock experiment: no toggling in the data path)
        plan = api.Plan(cfg)
        if os.environ.get('SGZ_WHOLE_FRAME') == '1': plan.set_option(api.OPT_CHANNEL_SPLIT, 0)
        if os.environ.get('SGZ_FETCH_WINDOW') == '1': plan.set_option(3, 1)
        if os.environ.get('SGZ_WIDE') == '1': plan.set_option(api.OPT_WIDE_GROUPS, 1)     # the 1024-thread form at N = 32768
        plan.upload()
        F = plan.num_frames(S)
        mapped = torch.empty((F, pairs, 2, plan.P), dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        # SGZ_BUFFERS=n (cfg2 only): the launches rotate over n distinct copies of the audio -- 16 x 23 MB is past the 256 MB Infinity Cache,
        # so every launch streams its input from HBM (the default loop re-renders one buffer, which stays cache-resident)
        nbuf = max(1, int(os.environ.get('SGZ_BUFFERS', '1'))) if name == "cfg2_348" else 1
        xs = [x] + [x.clone() for _ in range(nbuf - 1)]
        turn = [0]
        def nextx():
            turn[0] = (turn[0] + 1) % nbuf
            return xs[turn[0]]
        def ka():
            b = nextx()
            api.check(api.lib().sgz_stage_mapped_dominant(plan.h, b.data_ptr(), b.stride(0), S, mapped.data_ptr(), stream))
        rgba = torch.empty((F, plan.P, 4), dtype=torch.uint8, device="cuda")
        full = lambda: plan.render(nextx(), rgba=rgba)
        m, mn = timeit(ka, iters)
        fm, fmn = timeit(full, iters)
        byts = F * pairs * (2 * cfg['window_size'] * 4 + 4 * 1024)
        del xs
        out[name] = dict(ka_us=round(m, 2), ka_min_us=round(mn, 2), frac=round(byts / (m * 1e-6) / 8e12, 4), step_us=round(fm, 2),
                         per_task_ns=round(m * 1e3 / (F * pairs), 1))
    print(out)
if __name__ == "__main__":
    main()
