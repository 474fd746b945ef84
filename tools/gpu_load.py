"""A process that renders flat out for N seconds: background load on the device from ANOTHER process (the driver time-slices the
queues of several processes and saves / restores waves mid-kernel, which reorders the workgroups of one launch far more than other
streams of the same process do -- the RSNT carried-state race of round 6 showed only this way).
usage: gpu_load.py [seconds] [kind: spectrum|rsnt|scope]          prints READY once it renders"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from signalizer_amd import api, config, synth

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30
kind = sys.argv[2] if len(sys.argv) > 2 else "spectrum"
cfg = config.cfg2()
if kind == "rsnt":
    cfg = config.spectrum_config(algorithm=config.ALGO_RSNT, window_size=4096, hop=1024)
cfg["num_pairs"] = 4 if kind == "spectrum" else 1
frames = 348 if kind == "spectrum" else 200
S = cfg["window_size"] + cfg["hop"] * (frames - 1)
x = torch.from_numpy(synth.gen(9, int(cfg["sample_rate"]), S, 2 * cfg["num_pairs"])).cuda()
plan = api.Plan(cfg).upload()
out = plan.render(x)
print("READY", flush=True)
t0 = time.time()
while time.time() - t0 < seconds:
    for _ in range(32):
        plan.render(x, rgba=out)
    torch.cuda.synchronize()
