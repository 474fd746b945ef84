"""The schedule of one K_A launch (debug build, -DSGZ_DEBUG): per workgroup the start / end wall clock (100 MHz), the CU it ran on.
Prints, per dispatch wave, how the workgroups were spread over the CUs and what the last ones cost.  usage: unit_trace.py [seconds]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
cfg = config.cfg2()
S = int(float(sys.argv[1]) * 48000) if len(sys.argv) > 1 else int(60 * 48000)
x = torch.from_numpy(synth.gen(2, cfg["sample_rate"], S, 2)).cuda()
plan = api.Plan(cfg).upload()
F = plan.num_frames(S)
units = 2 * F
mapped = torch.empty((F, 1, 2, 1024), dtype=torch.float32, device="cuda")
clk = torch.zeros(256 + 4 * units, dtype=torch.int64, device="cuda")
L = api.lib()
L.sgz_debug_set_ablate(0xffff << 16)
L.sgz_debug_phase_clocks.argtypes = [C.c_void_p] * 2 + [C.c_size_t] * 2 + [C.c_void_p] * 3
for rep in range(4):
    api.check(L.sgz_debug_phase_clocks(plan.h, x.data_ptr(), x.stride(0), S, mapped.data_ptr(), clk.data_ptr(), None))
    torch.cuda.synchronize()
t = clk.cpu().numpy()[256:].reshape(units, 4)
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01            # us
hw, xcc = t[:, 2], t[:, 3] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
order = np.argsort(start)
print(f"units {units}  distinct CUs {len(set(cuid.tolist()))}  launch span {end.max():.2f} us (first start -> last end)")
dur = end - start
print(f"workgroup duration: min {dur.min():.2f} median {np.median(dur):.2f} max {dur.max():.2f} us")
late = order[512:] if units > 512 else order[:0]
first = order[:512]
print(f"first 512 starts: {start[first].min():.2f} .. {start[first].max():.2f} us; ends {end[first].min():.2f} .. {end[first].max():.2f}; duration median {np.median(dur[first]):.2f}")
if len(late):
    print(f"late {len(late)} starts: {start[late].min():.2f} .. {start[late].max():.2f} us; ends {end[late].min():.2f} .. {end[late].max():.2f}; duration median {np.median(dur[late]):.2f}")
    lc = cuid[late]
    cnt = np.bincount(np.unique(lc, return_counts=True)[1])
    print("late workgroups per CU (count of CUs with k late workgroups):", {k: int(v) for k, v in enumerate(cnt) if v})
    # for each late workgroup: was another workgroup resident on its CU during its run?
    shared = 0
    for u in late:
        same = np.where(cuid == cuid[u])[0]
        ov = [(min(end[u], end[v]) - max(start[u], start[v])) for v in same if v != u]
        if ov and max(ov) > 0.5 * dur[u]: shared += 1
    print(f"late workgroups that shared their CU for more than half their run: {shared}")
hist, edges = np.histogram(end, bins=12)
print("end-time histogram:", " ".join(f"{edges[i]:.0f}:{hist[i]}" for i in range(len(hist))))
per_xcc = [int((xcc == k).sum()) for k in range(8)]
print("workgroups per XCC:", per_xcc)
# idle: per CU, busy time = union of intervals
busy = 0.0
for c in set(cuid.tolist()):
    iv = sorted((start[u], end[u]) for u in np.where(cuid == c)[0])
    cur_s, cur_e = iv[0]
    for s_, e_ in iv[1:]:
        if s_ > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s_, e_
        else: cur_e = max(cur_e, e_)
    busy += cur_e - cur_s
print(f"CU busy (union of resident intervals) {busy / 256:.2f} us average per CU of {end.max():.2f}")
np.save("gpurun_out/unit_trace.npy", np.stack([start, end, cuid.astype(np.float64)], 1))
