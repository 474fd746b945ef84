import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth, sharding
cfg = config.cfg2()
L = api.lib()
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 0
L.sgz_debug_set_ablate(bits)
for F in (1, 8, 64, 256, 257, 348, 512, 768, 1024, 2048):
    S = 32768 + (F - 1) * 8192
    x = torch.from_numpy(synth.gen(2, 48000, S, 2)).cuda()
    plan = api.Plan(cfg).upload()
    r = sharding.TimeChunkRenderer(plan, x)
    t = min(r.time_stft_kernel(iters=20) for _ in range(3))
    print(f"ablate={bits} frames={F:5d}  {t*1e3:8.1f} us   {t*1e3/max(1,-(-F//256)):8.1f} us/round")
