"""Does the GPU need sustained load to reach its clock?  K_A launched back to back with an event pair around every launch (no host
wait in between): duration against launch index.  usage: clock_ramp_probe.py [launches]"""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
cfg = config.cfg2(); S = int(60 * 48000)
x = torch.from_numpy(synth.gen(2, 48000, S, 2)).cuda()
plan = api.Plan(cfg).upload()
F = plan.num_frames(S)
mapped = torch.empty((F, 1, 2, plan.P), dtype=torch.float32, device="cuda")
rgba = torch.empty((F, plan.P, 4), dtype=torch.uint8, device="cuda")
hip = ctypes.CDLL("libamdhip64.so")
st = torch.cuda.current_stream().cuda_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ev = [ctypes.c_void_p() for _ in range(n + 1)]
for e in ev: hip.hipEventCreate(ctypes.byref(e))
L = api.lib()
torch.cuda.synchronize(); time.sleep(0.5)                      # idle first
for i in range(n):
    hip.hipEventRecord(ev[i], ctypes.c_void_p(st))
    api.check(L.sgz_stage_mapped_dominant(plan.h, x.data_ptr(), x.stride(0), S, mapped.data_ptr(), st))
hip.hipEventRecord(ev[n], ctypes.c_void_p(st))
torch.cuda.synchronize()
d = []
for i in range(n):
    ms = ctypes.c_float(); hip.hipEventElapsedTime(ctypes.byref(ms), ev[i], ev[i + 1]); d.append(ms.value * 1e3)
d = np.array(d)
for a, b in ((0, 20), (20, 100), (100, 300), (300, 600), (600, 1000), (1000, 1500), (1500, 2000)):
    if b <= n: print(f"launches {a:4d}..{b:4d}: K_A back to back {np.median(d[a:b]):6.2f} us (median, event to event)")
