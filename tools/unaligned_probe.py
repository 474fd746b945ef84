"""rows at their natural 4-byte alignment (odd row stride, base shifted by one sample) on the real-input kernels: bits against the 8-byte
aligned layout, and the time.  The SGZ_ALLOW_UNALIGNED switch it once flipped is gone: the kernels take any row layout."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
for N, sr in ((32768, 48000), (16384, 48000), (65536, 96000)):
    cfg = config.spectrum_config(window_size=N, hop=N // 4, sample_rate=float(sr))
    plan = api.Plan(cfg).upload()
    S = N + 40 * (N // 4)
    x = synth.gen(3, sr, S, 2)
    even = torch.zeros((2, S + 2), dtype=torch.float32, device="cuda"); even[:, :S] = torch.from_numpy(x).cuda()
    odd = torch.zeros((2, S + 1), dtype=torch.float32, device="cuda"); odd[:, :S] = torch.from_numpy(x).cuda()
    a = plan.stage_mapped(even[:, :S]).cpu().numpy()
    os.environ.pop("SGZ_ALLOW_UNALIGNED", None)
    b0 = plan.stage_mapped(odd[:, :S]).cpu().numpy()
    os.environ["SGZ_ALLOW_UNALIGNED"] = "1"
    b1 = plan.stage_mapped(odd[:, :S]).cpu().numpy()
    # offset start by one sample too (base pointer 4-byte aligned only)
    c = plan.stage_mapped(even[:, 1:S + 1]).cpu().numpy()
    os.environ.pop("SGZ_ALLOW_UNALIGNED", None)
    c0 = plan.stage_mapped(even[:, 1:S + 1]).cpu().numpy()
    print(N, "odd stride: fallback == aligned bits", np.array_equal(a, b0), " unaligned-real == aligned bits", np.array_equal(a, b1),
          " shifted base: real vs fallback max rel", np.abs(c - c0).max() / np.abs(c0).max())
    def t(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    os.environ["SGZ_ALLOW_UNALIGNED"] = "1"
    tu = t(lambda: plan.stage_mapped(odd[:, :S]))
    os.environ.pop("SGZ_ALLOW_UNALIGNED", None)
    tf = t(lambda: plan.stage_mapped(odd[:, :S]))
    ta = t(lambda: plan.stage_mapped(even[:, :S]))
    print("   us: aligned", round(ta, 1), " odd stride real kernel", round(tu, 1), " odd stride fallback", round(tf, 1))
