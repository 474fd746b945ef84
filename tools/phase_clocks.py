"""Per-wave phase clocks of one K_A workgroup (debug hook sgz_debug_phase_clocks): for every phase boundary the
earliest / latest wave, relative to the workgroup's first clock.  usage: phase_clocks.py [samples] [ablate bits] [task]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
# SGZ_CFG5=1: the halves kernel of cfg5 (32 pairs, N = 65536); ablate bit 15 selects the odd half
cfg5 = os.environ.get("SGZ_CFG5") == "1"
cfg = config.cfg5(pairs=32) if cfg5 else config.cfg2()
S = int(sys.argv[1]) if len(sys.argv) > 1 else (int(4 * 96000) if cfg5 else int(60 * 48000))
x = torch.from_numpy(synth.gen(2, cfg["sample_rate"], S, 2 * cfg["num_pairs"])).cuda()
plan = api.Plan(cfg).upload()
F = plan.num_frames(S)
mapped = torch.empty((F, cfg['num_pairs'], 2, 1024), dtype=torch.float32, device="cuda")
clk = torch.zeros(16 * 16, dtype=torch.int64, device="cuda")
L = api.lib()
L.sgz_debug_set_ablate((int(sys.argv[2]) if len(sys.argv) > 2 else 0) | ((int(sys.argv[3]) if len(sys.argv) > 3 else 0) << 16))
L.sgz_debug_phase_clocks.argtypes = [C.c_void_p] * 2 + [C.c_size_t] * 2 + [C.c_void_p] * 3
names = ["start", "pass1 done", "ex1 done", "pass2 done", "ex2 done", "pass3 done", "mirror done", "M in LDS", "binsOut", "end",
         "map: pieces scanned", "map: interp done", "map: barrier", "windowed", "dif1 done"]
# SGZ_BUFFERS=n: the launches rotate over n copies of the audio (n x 23 MB at cfg2: 16 of them are past the Infinity Cache -- the input of the
# launch that is reported then comes from HBM)
xs = [x] + [x.clone() for _ in range(max(1, int(os.environ.get("SGZ_BUFFERS", "1"))) - 1)]
for rep in range(max(3, 2 * len(xs))):
    b = xs[rep % len(xs)]
    api.check(L.sgz_debug_phase_clocks(plan.h, b.data_ptr(), b.stride(0), S, mapped.data_ptr(), clk.data_ptr(), None))
    torch.cuda.synchronize()
c = clk.cpu().numpy().reshape(16, 16)
c = c[:int(os.environ.get('SGZ_WAVES', '16'))]      # (the channel-split kernel at N = 32768 has 8 waves)
t0 = c[:, 0].min()
order = [0, 13, 14, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 9]
if cfg5 and os.environ.get("SGZ_REALCLK") != "1":   # halves kernel: the map slots are then written by mapSideKernel (a later launch): print the two kernels
    # separately.  (SGZ_REALCLK=1: the plan runs the channel-split kernel at N = 65536 -- the default since round 2 --, one kernel, the full list)
    order = [0, 7, 10, 11, 12, 9] if os.environ.get("SGZ_MAPCLK") == "1" else [13, 14, 1, 2, 3, 4, 5, 6, 8]
    t0 = c[:, 0].min() if os.environ.get("SGZ_MAPCLK") == "1" else c[:, 13].min()
print(f"{'boundary':22s} {'first wave':>10s} {'last wave':>10s} {'wave 0':>8s} {'wave 15':>8s}")
for i in order:
    col = c[:, i] - t0
    print(f"{names[i]:22s} {col.min():10d} {col.max():10d} {col[0]:8d} {col[-1]:8d}   " + " ".join(str(v) for v in col))
