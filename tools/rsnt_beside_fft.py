"""In ONE process: the RSNT bf16 matrix-core kernel on one stream, FFT work on another (PyTorch's rocFFT transform; this library's K_A),
every FFT result compared with the quiet run.  (tools/mp_control.py found that RSNT renders in OTHER processes disturb rocFFT.)
usage: rsnt_beside_fft.py [iterations] [kind: rsnt (the default form)|rsnt_bf16|rsnt_valu|rsnt_fp32mfma|none]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signalizer_amd import api, config as cf, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
kind = sys.argv[2] if len(sys.argv) > 2 else "rsnt"
gpu = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(5)
xt = torch.randn((64, 32768), generator=g).to(gpu)
cfg2 = cf.cfg2()
S2 = 32768 + 8192 * 99
x2 = torch.from_numpy(synth.gen(9, 48000, S2, 2)).to(gpu)
ka = api.Plan(cfg2).upload()
rc = cf.spectrum_config(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1024)
xr = torch.from_numpy(synth.gen(9, 48000, 4096 + 1024 * 199, 2)).to(gpu)
rp = api.Plan(rc)
if kind == "rsnt_bf16": rp.set_option(api.OPT_MATRIX_RESONATOR, 1)
if kind == "rsnt_valu": rp.set_option(api.OPT_MATRIX_RESONATOR, 0)
if kind == "rsnt_fp32mfma": rp.set_option(api.OPT_MATRIX_RESONATOR, 2)
rp.upload()
rout = rp.render(xr)
want_fft = torch.view_as_real(torch.fft.rfft(xt)).clone()
want_ka = ka.stage_mapped(x2).view(torch.int32).clone()
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(device=gpu), torch.cuda.Stream(device=gpu)
bad_fft = bad_ka = 0
for it in range(0, n, 8):
    outs = []
    for k in range(8):
        if kind != "none":
            rp.render(xr, rgba=rout, stream=s1.cuda_stream)
            rp.render(xr, rgba=rout, stream=s1.cuda_stream)
        with torch.cuda.stream(s2):
            outs.append((torch.view_as_real(torch.fft.rfft(xt)), ka.stage_mapped(x2).view(torch.int32)))
    torch.cuda.synchronize()
    for f, m in outs:
        bad_fft += 0 if torch.equal(f.view(torch.int32), want_fft.view(torch.int32)) else 1
        bad_ka += 0 if torch.equal(m, want_ka) else 1
print(f"{kind:14s} on the other stream: rocFFT rfft {bad_fft} of {n} differ, this library's K_A (N = 32768, 100 frames) {bad_ka} of {n} differ", flush=True)
