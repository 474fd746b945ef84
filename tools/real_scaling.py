"""K_A launch time against the number of frames (cfg2 settings), both forms of the N = 32768 channel-split kernel, sustained clock:
the staircase shows what a workgroup costs alone on its CU, as one of two, and per dispatch generation"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signalizer_amd import api, config, synth
from ka_time import timeit
cfg = config.cfg2()
plans = {"wide(1024x16)": api.Plan(cfg).upload(), "narrow(512x32)": api.Plan(cfg).set_option(api.OPT_WIDE_GROUPS, 0).upload()}
x = torch.from_numpy(synth.gen(2, 48000, 32768 + 8192 * 1100, 2)).cuda()
stream = torch.cuda.current_stream().cuda_stream
frames = [int(a) for a in sys.argv[1:]] or [1, 16, 64, 128, 192, 256, 320, 348, 384, 512, 768, 1024]
print("frames  workgroups  " + "  ".join(f"{k:>16s}" for k in plans))
for F in frames:
    S = 32768 + 8192 * (F - 1)
    mapped = torch.empty((F, 1, 2, 1024), dtype=torch.float32, device="cuda")
    row = []
    for name, plan in plans.items():
        fn = lambda: api.check(api.lib().sgz_stage_mapped_dominant(plan.h, x.data_ptr(), x.stride(0), S, mapped.data_ptr(), stream))
        m, mn = timeit(fn, 40, spin_ms=30.0, batches=3)
        row.append(f"{m:13.2f} us")
    print(f"{F:6d}  {2 * F:10d}  " + "  ".join(row), flush=True)
