"""Bit-identity under concurrency for every kernel family: whole renders (K_A + K_B) of three fixed buffers over four plans / streams,
against the quiet run, for a list of configurations that together reach every K_A / K_B form.   usage: overlap_stress_cfgs.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config as cf, synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
gpu = torch.device("cuda", 0)
CASES = {
    "real N=32768 separate (bench)": dict(),
    "real N=16384 midside": dict(window_size=16384, hop=4096, channel_mode=cf.CH_MIDSIDE),
    "real N=65536 two pairs": dict(window_size=65536, hop=16384, num_pairs=2, sample_rate=96000.0),
    "real mono merge N=32768": dict(channel_mode=cf.CH_MERGE),
    "wide groups N=32768": dict(_wide=1),
    "whole-frame complex N=32768": dict(channel_mode=cf.CH_COMPLEX),
    "whole-frame N=4096 zero-padded": dict(window_size=3000, hop=750),
    "halves N=8192": dict(window_size=8192, hop=2048),
    "generic N=2048": dict(window_size=2048, hop=512),
    "phase N=4096": dict(window_size=4096, hop=1024, channel_mode=cf.CH_PHASE),
    "phase N=32768": dict(channel_mode=cf.CH_PHASE),
    "rsnt hop 1024 (matrix cores)": dict(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1024),
    "rsnt hop 1000 (vector form)": dict(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1000),
    "three pairs N=4096 (scan/emit K_B)": dict(window_size=4096, hop=1024, num_pairs=3),
    "fetch window blackman N=32768": dict(window_type=cf.WIN_BLACKMAN),
}
total_bad = 0
for name, over in CASES.items():
    over = dict(over)
    wide = over.pop("_wide", 0)
    cfg = cf.spectrum_config(**over)
    frames = 120 if cfg["window_size"] >= 16384 else 200
    S = cfg["window_size"] + cfg["hop"] * (frames - 1)
    xs = [torch.from_numpy(synth.gen(500 + k, int(cfg["sample_rate"]), S, 2 * cfg["num_pairs"])).to(gpu) for k in range(3)]
    def mk():
        p = api.Plan(cfg)
        if wide: p.set_option(api.OPT_WIDE_GROUPS, 1)
        return p.upload()
    ref = mk()
    want = [ref.render(x).clone() for x in xs]
    torch.cuda.synchronize()
    plans = [mk() for _ in range(4)]
    streams = [torch.cuda.Stream(device=gpu) for _ in range(4)]
    bad = 0
    n = rounds if cfg["algorithm"] == 0 else max(20, rounds // 4)
    for r in range(n):
        outs = []
        torch.cuda.synchronize()
        for k in range(9):
            outs.append(plans[k % 4].render(xs[k % 3], stream=streams[k % 4].cuda_stream))
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(outs[k], want[k % 3]) else 1 for k in range(9))
    total_bad += bad
    print(f"{name:40s}: {bad} of {n * 9} renders differ (path {ref.path})", flush=True)
print("total differing:", total_bad)
