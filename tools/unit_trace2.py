"""Do the workgroups of launch i + 1 start while launch i's last generation runs?  (round-5 review item 2a)
Debug build (tools/mkdebug.sh; SGZ_LIB=tools/ab/lib_dbg.so): every K_A workgroup leaves (start, end) in the 100 MHz wall clock all CUs
share.  K_A launches of cfg2 (696 workgroups; two resident per CU, so 512 + 184) are enqueued back to back
  (a) on ONE stream (in order: the runtime puts a barrier between them),
  (b) alternating between TWO streams / two plans (bench.py's `two_in_flight`),
  (c) like (b) with K_B (the fused decay / colour kernel) behind every K_A on its stream -- the whole step,
and for every launch the tool prints when its first / 512th / last workgroup started and when its last one ended relative to the previous
launch's last end, and how many of its workgroups started before that end.
usage: SGZ_LIB=tools/ab/lib_dbg.so python tools/unit_trace2.py [launches]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from signalizer_amd import api, config, synth

N_LAUNCH = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = config.cfg2()
S = int(60 * 48000)
L = api.lib()
L.sgz_debug_set_ablate(0xffff << 16)
L.sgz_debug_phase_clocks.argtypes = [C.c_void_p] * 2 + [C.c_size_t] * 2 + [C.c_void_p] * 3
plans = [api.Plan(cfg).upload() for _ in range(2)]
xs = [torch.from_numpy(synth.gen(2 + k, cfg["sample_rate"], S, 2)).cuda() for k in range(2)]
F = plans[0].num_frames(S)
units = 2 * F
mapped = [torch.empty((F, 1, 2, 1024), dtype=torch.float32, device="cuda") for _ in range(2)]
rgba = [torch.empty((F, 1024, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(mode: str):
    clks = [torch.zeros(256 + 4 * units, dtype=torch.int64, device="cuda") for _ in range(N_LAUNCH)]
    torch.cuda.synchronize()
    # spin the clock up first (untimed launches), then the traced ones without a host wait in between
    for rep in range(200):
        k = rep & 1 if mode != "one" else 0
        api.check(L.sgz_debug_phase_clocks(plans[k].h, xs[k].data_ptr(), xs[k].stride(0), S, mapped[k].data_ptr(), clks[0].data_ptr(), streams[k].cuda_stream))
    for j in range(N_LAUNCH):
        k = j & 1 if mode != "one" else 0
        api.check(L.sgz_debug_phase_clocks(plans[k].h, xs[k].data_ptr(), xs[k].stride(0), S, mapped[k].data_ptr(), clks[j].data_ptr(), streams[k].cuda_stream))
        if mode == "step":
            api.check(L.sgz_stage_decay_colour(plans[k].h, mapped[k].data_ptr(), F, rgba[k].data_ptr(), None, None, streams[k].cuda_stream))
    torch.cuda.synchronize()
    t = [c.cpu().numpy()[256:].reshape(units, 4) for c in clks]
    t0 = min(int(a[:, 0].min()) for a in t)
    print(f"== {mode}: {N_LAUNCH} K_A launches" + {"one": " on one stream", "two": " alternating on two streams", "step": " + K_B behind each, alternating on two streams"}[mode])
    print(" launch | first start | 512th start | last start | last end | span | starts before the previous launch's last end | gap after it (us)")
    prev_end = None
    spans = []
    for j, a in enumerate(t):
        st = np.sort((a[:, 0] - t0) * 0.01)
        en = (a[:, 1] - t0) * 0.01
        early = int((st < prev_end).sum()) if prev_end is not None else 0
        gap = st[0] - prev_end if prev_end is not None else float("nan")
        print(f"  {j:4d}  | {st[0]:9.2f} | {st[min(511, units - 1)]:9.2f} | {st[-1]:9.2f} | {en.max():8.2f} | {en.max() - st[0]:6.2f} | {early:5d} | {gap:7.2f}")
        spans.append((st[0], en.max()))
        prev_end = en.max()
    ends = [e for _, e in spans]
    per = (ends[-1] - ends[1]) / (len(ends) - 2) if len(ends) > 2 else float("nan")
    print(f"   steady state: {per:.2f} us per launch (last end to last end, launches 1 .. {N_LAUNCH - 1})")


L.sgz_stage_decay_colour.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
for mode in ("one", "two", "step"):
    run(mode)
