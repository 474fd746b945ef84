"""one K_A stage call in a loop, for counter runs: run_stage.py bins|mapped [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signalizer_amd import api, config, synth
cfg = config.cfg2()
S = int(config.CFG2_SECONDS * 48000)
x = torch.from_numpy(synth.gen(config.CFG2_SEED, 48000, S, 2)).cuda()
plan = api.Plan(cfg).upload()
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    out = plan.stage_bins(x) if sys.argv[1] == "bins" else plan.stage_mapped(x)
torch.cuda.synchronize()
