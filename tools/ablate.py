import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth, sharding
cfg = config.cfg2()
S = int(60 * 48000)
x = torch.from_numpy(synth.gen(2, 48000, S, 2)).cuda()
plan = api.Plan(cfg).upload()
r = sharding.TimeChunkRenderer(plan, x)
L = api.lib()
names = {0: "full", 512: "empty kernel", 63|1024: "nothing, no M write", 63|1024|64|128: "nothing, no M write, only L", 63 | 64: "nothing, no window", 63 | 128: "nothing, no R", 63 | 64 | 128: "nothing, only L", 63 | 256: "nothing, hot frames", 63|64|256: "nothing, hot, no window", 256: "full, hot frames", 64: "full, no window",  1: "no dif", 2: "no ex1", 4: "no ex2", 8: "no mirror", 16: "no map", 32: "no twiddle", 63: "nothing (loads+barriers only)",
         62: "only dif", 1 | 32: "no dif, no tw", 2 | 4 | 8: "no lds", 16 | 8: "no mirror,no map"}
for bits, name in names.items():
    L.sgz_debug_set_ablate(bits)
    t = min(r.time_stft_kernel(iters=20) for _ in range(3))
    print(f"{name:32s} {t*1e3:8.1f} us")
L.sgz_debug_set_ablate(0)
