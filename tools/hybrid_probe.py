"""what a hybrid K_A launch could give at cfg2: the 256 first frames by the whole-frame kernel (one per CU) and the 92 remaining frames as
184 channel-split half-tasks (one per CU), timed as two back-to-back launches"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
cfg = config.cfg2()
os.environ["SGZ_CHANNEL_SPLIT"] = "0"
whole = api.Plan(cfg).upload()
os.environ["SGZ_CHANNEL_SPLIT"] = "1"
split = api.Plan(cfg).upload()
S = int(60 * 48000)
x = torch.from_numpy(synth.gen(2, 48000, S, 2)).cuda()
hip = ctypes.CDLL("libamdhip64.so")
stream = torch.cuda.current_stream().cuda_stream
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
hip.hipEventCreate(ctypes.byref(e0)); hip.hipEventCreate(ctypes.byref(e1))
F = whole.num_frames(S)
mapped = torch.empty((F, 1, 2, 1024), dtype=torch.float32, device="cuda")
def run(plan, f0, nf):
    off = f0 * 8192
    n = 32768 + (nf - 1) * 8192
    api.check(api.lib().sgz_stage_mapped(plan.h, x.data_ptr() + off * 4, x.stride(0), n, mapped.data_ptr() + f0 * 2 * 1024 * 4, stream))
def timeit(fn):
    for _ in range(10): fn()
    t = []
    for _ in range(100):
        hip.hipEventRecord(e0, ctypes.c_void_p(stream)); fn(); hip.hipEventRecord(e1, ctypes.c_void_p(stream)); hip.hipEventSynchronize(e1)
        ms = ctypes.c_float(); hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1); t.append(ms.value * 1e3)
    return round(float(np.mean(t)), 2)
print("whole-frame, 348 frames      ", timeit(lambda: run(whole, 0, F)))
print("whole-frame, 256 frames      ", timeit(lambda: run(whole, 0, 256)))
print("whole-frame, 92 frames       ", timeit(lambda: run(whole, 256, 92)))
print("channel-split, 92 frames     ", timeit(lambda: run(split, 256, 92)))
print("channel-split, 348 frames    ", timeit(lambda: run(split, 0, F)))
print("whole 256 + split 92 (2 launches)", timeit(lambda: (run(whole, 0, 256), run(split, 256, 92))))
print("empty-ish launch (1 frame)   ", timeit(lambda: run(whole, 0, 1)), timeit(lambda: run(split, 0, 1)))
