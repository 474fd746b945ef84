"""kernel-trace probe (run under rocprofv3 --kernel-trace --stats): which kernels a render with / without line results and state launches"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
cfg = config.cfg2()
S = int(60 * 48000)
x = torch.from_numpy(synth.gen(2, 48000, S, 2)).cuda()
plan = api.Plan(cfg).upload()
F = plan.num_frames(S)
lines = torch.empty((F, 1, 2, plan.P, 2), dtype=torch.float32, device="cuda")
state = torch.zeros((1, 2, plan.P, 2), dtype=torch.float32, device="cuda")
rgba = torch.empty((F, plan.P, 4), dtype=torch.uint8, device="cuda")
mode = sys.argv[1] if len(sys.argv) > 1 else "state"
for _ in range(20):
    if mode == "state": plan.render(x, rgba=rgba, lines=lines, state=state)
    elif mode == "mapped": plan.stage_mapped(x)
    else: plan.render(x, rgba=rgba)
torch.cuda.synchronize()
