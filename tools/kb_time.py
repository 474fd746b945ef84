"""K_B alone on precomputed magnitudes (sgz_stage_decay_colour), image only and with line results + state: HIP events around batches of launches at the sustained clock.  usage: [SGZ_LIB=...] kb_time.py [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from signalizer_amd import api, config, synth
from ka_time import timeit
cfg = config.cfg2()
S = int(60 * 48000)
x = torch.from_numpy(synth.gen(2, 48000, S, 2)).cuda()
plan = api.Plan(cfg).upload()
F = plan.num_frames(S)
mapped = plan.stage_mapped(x)
rgba = torch.empty((F, plan.P, 4), dtype=torch.uint8, device="cuda")
lines = torch.empty((F, 1, 2, plan.P, 2), dtype=torch.float32, device="cuda")
state = torch.zeros((1, 2, plan.P, 2), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
L = api.lib()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
img = timeit(lambda: api.check(L.sgz_stage_decay_colour(plan.h, mapped.data_ptr(), F, rgba.data_ptr(), None, None, st)), iters)
full = timeit(lambda: api.check(L.sgz_stage_decay_colour(plan.h, mapped.data_ptr(), F, rgba.data_ptr(), lines.data_ptr(), state.data_ptr(), st)), iters)
ends = timeit(lambda: api.check(L.sgz_stage_decay_colour(plan.h, mapped.data_ptr(), F, rgba.data_ptr(), None, state.data_ptr(), st)), iters)
print(f"K_B image + end state (no per-frame lines) {ends[0]:.2f}/{ends[1]:.2f} us")
print(f"K_B image only {img[0]:.2f}/{img[1]:.2f} us   with lines + state {full[0]:.2f}/{full[1]:.2f} us")
