"""K_A launch time against the number of frames (cfg2 settings): shows how many workgroups a CU really holds at a time"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
cfg = config.cfg2()
plan = api.Plan(cfg).upload()
hip = ctypes.CDLL("libamdhip64.so")
x = torch.from_numpy(synth.gen(2, 48000, 32768 + 8192 * 1100, 2)).cuda()
stream = torch.cuda.current_stream().cuda_stream
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
hip.hipEventCreate(ctypes.byref(e0)); hip.hipEventCreate(ctypes.byref(e1))
for F in (32, 64, 128, 192, 256, 320, 384, 512, 768, 1024):
    S = 32768 + 8192 * (F - 1)
    mapped = torch.empty((F, 1, 2, 1024), dtype=torch.float32, device="cuda")
    fn = lambda: api.check(api.lib().sgz_stage_mapped_dominant(plan.h, x.data_ptr(), x.stride(0), S, mapped.data_ptr(), stream))
    for _ in range(5): fn()
    t = []
    for _ in range(30):
        hip.hipEventRecord(e0, ctypes.c_void_p(stream)); fn(); hip.hipEventRecord(e1, ctypes.c_void_p(stream)); hip.hipEventSynchronize(e1)
        ms = ctypes.c_float(); hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1); t.append(ms.value * 1e3)
    print(f"frames {F:5d}  {np.mean(t):8.1f} us   {np.mean(t) * 1e3 / F:7.1f} ns/frame")
