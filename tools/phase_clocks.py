import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
cfg = config.cfg2()
S = int(sys.argv[1]) if len(sys.argv) > 1 else int(60 * 48000)
x = torch.from_numpy(synth.gen(2, 48000, S, 2)).cuda()
plan = api.Plan(cfg).upload()
F = plan.num_frames(S)
mapped = torch.empty((F, 1, 2, 1024), dtype=torch.float32, device="cuda")
clk = torch.zeros(16 + 64, dtype=torch.int64, device="cuda")
L = api.lib()
L.sgz_debug_set_ablate((int(sys.argv[2]) if len(sys.argv) > 2 else 0) | ((int(sys.argv[3]) if len(sys.argv) > 3 else 0) << 16))
L.sgz_debug_phase_clocks.argtypes = [C.c_void_p] * 2 + [C.c_size_t] * 2 + [C.c_void_p] * 3
names = ["0:top", "1:dif1+tw1", "2:ex1", "3:dif2+tw2", "4:ex2", "5:dif3", "6:mirror", "7:Mwrite+fix", "8:binsOut+prefetch", "9:map+window"]
for rep in range(3):
    api.check(L.sgz_debug_phase_clocks(plan.h, x.data_ptr(), x.stride(0), S, mapped.data_ptr(), clk.data_ptr(), None))
    torch.cuda.synchronize()
    c = clk.cpu().numpy()
    d = np.diff(c[:10])
    print('   map detail: items', c[10]-c[8], 'interp', c[11]-c[10], 'barrier', c[12]-c[11], 'resolve', c[9]-c[12])
    w = c[16:].reshape(16, 4).astype(np.int64) - int(c[0])
    print("   per-wave [start, pass3 done, map start, end]:", " ".join(f"w{i}:{w[i,0]}/{w[i,1]}/{w[i,2]}/{w[i,3]}" for i in range(16)))
    print("rep", rep, "total cycles", c[9] - c[0], " ".join(f"{n}={int(v)}" for n, v in zip(names, d)))
