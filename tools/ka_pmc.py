"""K_A alone, a few launches, for counter collection: ka_pmc.py [cfg2|notail] [launches]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signalizer_amd import api, config, synth
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
pairs = 8 if which == "notail" else 1
cfg = config.cfg2(); cfg["num_pairs"] = pairs
S = int(config.CFG2_SECONDS * 48000)
x = torch.from_numpy(synth.gen(2, 48000, S, 2 * pairs)).cuda()
plan = api.Plan(cfg)
if os.environ.get("SGZ_WIDE") == "1": plan.set_option(api.OPT_WIDE_GROUPS, 1)
plan.upload()
F = plan.num_frames(S)
mapped = torch.empty((F, pairs, 2, plan.P), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for _ in range(n):
    api.check(api.lib().sgz_stage_mapped_dominant(plan.h, x.data_ptr(), x.stride(0), S, mapped.data_ptr(), stream))
torch.cuda.synchronize()
