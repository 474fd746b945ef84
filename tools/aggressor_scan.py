"""Does any kernel family of the library disturb a bystander?  (The bf16 RSNT kernel does, on these MI355X boxes: NOTES.md round 6.)
Every configuration of tools/overlap_stress_cfgs.py in turn renders flat out on one stream while PyTorch's rocFFT transform and this
library's bench kernel (K_A, N = 32768) run on another; every bystander result is compared with the quiet run's.
usage: aggressor_scan.py [iterations per aggressor]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signalizer_amd import api, config as cf, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
gpu = torch.device("cuda", 0)
AGGRESSORS = {
    "real N=32768 separate (bench)": dict(),
    "real N=16384 midside": dict(window_size=16384, hop=4096, channel_mode=cf.CH_MIDSIDE),
    "real N=65536 two pairs (walking)": dict(window_size=65536, hop=16384, num_pairs=2, sample_rate=96000.0),
    "wide groups N=32768": dict(_wide=1),
    "whole-frame complex N=32768": dict(channel_mode=cf.CH_COMPLEX),
    "whole-frame N=4096 zero-padded": dict(window_size=3000, hop=750),
    "halves N=8192": dict(window_size=8192, hop=2048),
    "generic N=2048": dict(window_size=2048, hop=512),
    "phase N=32768": dict(channel_mode=cf.CH_PHASE),
    "three pairs N=4096 (scan/emit K_B)": dict(window_size=4096, hop=1024, num_pairs=3),
    "rsnt default (fp32 matrix kernel)": dict(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1024),
    "rsnt vector-ALU form (hop 1000)": dict(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1000),
    "rsnt bf16 matrix kernel (opt-in)": dict(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1024, _form=1),
}
g = torch.Generator(device="cpu").manual_seed(5)
xt = torch.randn((64, 32768), generator=g).to(gpu)
x2 = torch.from_numpy(synth.gen(9, 48000, 32768 + 8192 * 99, 2)).to(gpu)
ka = api.Plan(cf.cfg2()).upload()
want_fft = torch.view_as_real(torch.fft.rfft(xt)).clone()
want_ka = ka.stage_mapped(x2).view(torch.int32).clone()
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(device=gpu), torch.cuda.Stream(device=gpu)
for name, over in AGGRESSORS.items():
    over = dict(over)
    wide, form = over.pop("_wide", 0), over.pop("_form", None)
    cfg = cf.spectrum_config(**over)
    frames = 100 if cfg["window_size"] >= 16384 else 200
    xa = torch.from_numpy(synth.gen(7, int(cfg["sample_rate"]), cfg["window_size"] + cfg["hop"] * (frames - 1), 2 * cfg["num_pairs"])).to(gpu)
    p = api.Plan(cfg)
    if wide: p.set_option(api.OPT_WIDE_GROUPS, 1)
    if form is not None: p.set_option(api.OPT_MATRIX_RESONATOR, form)
    p.upload()
    out = p.render(xa)
    torch.cuda.synchronize()
    bad_fft = bad_ka = 0
    for it in range(0, n, 8):
        outs = []
        for k in range(8):
            p.render(xa, rgba=out, stream=s1.cuda_stream)
            p.render(xa, rgba=out, stream=s1.cuda_stream)
            with torch.cuda.stream(s2):
                outs.append((torch.view_as_real(torch.fft.rfft(xt)), ka.stage_mapped(x2).view(torch.int32)))
        torch.cuda.synchronize()
        for f, m in outs:
            bad_fft += 0 if torch.equal(f.view(torch.int32), want_fft.view(torch.int32)) else 1
            bad_ka += 0 if torch.equal(m, want_ka) else 1
    print(f"{name:38s} as the neighbour: rocFFT {bad_fft} of {n} differ, K_A {bad_ka} of {n} differ", flush=True)
