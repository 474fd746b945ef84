"""The step with line results and end state (K_A + the full fused K_B), and that K_B alone on precomputed magnitudes: HIP events around batches at the sustained clock.
usage: [SGZ_LIB=...] state_time.py [iters]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ka_time import timeit
cfg = config.cfg2()
S = int(60 * 48000)
x = torch.from_numpy(synth.gen(2, 48000, S, 2)).cuda()
plan = api.Plan(cfg).upload()
F = plan.num_frames(S)
lines = torch.empty((F, 1, 2, plan.P, 2), dtype=torch.float32, device="cuda")
state = torch.zeros((1, 2, plan.P, 2), dtype=torch.float32, device="cuda")
rgba = torch.empty((F, plan.P, 4), dtype=torch.uint8, device="cuda")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
full, fmin = timeit(lambda: plan.render(x, rgba=rgba, lines=lines, state=state), iters)
img, imin = timeit(lambda: plan.render(x, rgba=rgba), iters)
print(f"with lines+state {full:.2f}/{fmin:.2f} us   image only {img:.2f}/{imin:.2f} us")
