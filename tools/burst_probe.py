"""how much of a K_A launch is the fetch of its samples: the same launch shapes with hop = 2 (every frame reads the same 128 KB: cache
resident) against hop = N / 4 (cfg2: 32 KB of new samples per frame and channel from HBM / MALL)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
hip = ctypes.CDLL("libamdhip64.so")
stream = torch.cuda.current_stream().cuda_stream
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
hip.hipEventCreate(ctypes.byref(e0)); hip.hipEventCreate(ctypes.byref(e1))
x = torch.from_numpy(synth.gen(2, 48000, 32768 + 8192 * 1100, 2)).cuda()
for hop in (8192, 2):
    cfg = config.cfg2(); cfg["hop"] = hop
    plan = api.Plan(cfg).upload()
    for F in (128, 256, 348, 512, 1024):
        S = 32768 + hop * (F - 1)
        mapped = torch.empty((F, 1, 2, 1024), dtype=torch.float32, device="cuda")
        fn = lambda: api.check(api.lib().sgz_stage_mapped(plan.h, x.data_ptr(), x.stride(0), S, mapped.data_ptr(), stream))
        for _ in range(5): fn()
        t = []
        for _ in range(40):
            hip.hipEventRecord(e0, ctypes.c_void_p(stream)); fn(); hip.hipEventRecord(e1, ctypes.c_void_p(stream)); hip.hipEventSynchronize(e1)
            ms = ctypes.c_float(); hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1); t.append(ms.value * 1e3)
        print(f"hop {hop:5d} frames {F:5d} path {plan.path}  {np.mean(t):7.1f} us  (min {np.min(t):6.1f})")
