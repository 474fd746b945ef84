import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from signalizer_amd import api, config, synth, sharding
cfg = config.cfg2(); sr = 48000; S = int(60 * sr)
x_host = synth.gen(2, sr, S, 2)
plan = api.Plan(cfg).upload()
x = torch.from_numpy(x_host).cuda()
tr = sharding.TimeChunkRenderer(plan, x, rank=0, world=1)
F = plan.num_frames(S)
rgba = torch.empty((F, plan.P, 4), dtype=torch.uint8, device="cuda")
def loop(fn, n=400):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
for name, fn in (("TimeChunkRenderer.render", tr.render), ("plan.render(x)", lambda: plan.render(x, rgba=rgba)),
                 ("plan.render(view of padded buf)", lambda: plan.render(tr._view(), rgba=rgba))):
    for r in range(3):
        enq, tot = loop(fn)
        print(f"{name:34s} enqueue {enq:6.2f} us/step   total {tot:6.2f} us/step")
