"""A process that renders flat out for N seconds: background load on the device from ANOTHER process (the driver time-slices the
queues of several processes and saves / restores waves mid-kernel, which reorders the workgroups of one launch far more than other
streams of the same process do -- the RSNT carried-state race of round 6 showed only this way).
usage: gpu_load.py [seconds] [kind: spectrum|rsnt|rsnt_bf16|rsnt_valu|rsnt_fp32mfma|torch|torch_bf16]          prints READY once it renders
kind torch: PyTorch's own operators only (rocFFT transforms and matrix products; this library is not even loaded) -- the control for
tools/mp_control.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30
kind = sys.argv[2] if len(sys.argv) > 2 else "spectrum"
if kind == "torch_bf16":                                      # matrix cores flat out (hipBLASLt bf16): the power / clock regime of the RSNT kernels
    at = torch.randn((8192, 8192), device="cuda", dtype=torch.bfloat16)
    at @ at
    torch.cuda.synchronize()
    print("READY", flush=True)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(8):
            at @ at
        torch.cuda.synchronize()
    sys.exit(0)
if kind == "torch":
    xt = torch.randn((256, 32768), device="cuda")
    at = torch.randn((2048, 2048), device="cuda")
    torch.fft.rfft(xt); at @ at
    torch.cuda.synchronize()
    print("READY", flush=True)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(16):
            torch.fft.rfft(xt)
            at @ at
        torch.cuda.synchronize()
    sys.exit(0)

from signalizer_amd import api, config, synth

cfg = config.cfg2()
if kind.startswith("rsnt"):
    cfg = config.spectrum_config(algorithm=config.ALGO_RSNT, window_size=4096, hop=1024)
cfg["num_pairs"] = 4 if kind == "spectrum" else 1
frames = 348 if kind == "spectrum" else 200
S = cfg["window_size"] + cfg["hop"] * (frames - 1)
x = torch.from_numpy(synth.gen(9, int(cfg["sample_rate"]), S, 2 * cfg["num_pairs"])).cuda()
plan = api.Plan(cfg)
if kind == "rsnt_valu":
    plan.set_option(api.OPT_MATRIX_RESONATOR, 0)          # the vector-ALU block form: no matrix instructions
if kind == "rsnt_fp32mfma":
    plan.set_option(api.OPT_MATRIX_RESONATOR, 2)          # v_mfma_f32_32x32x2_f32 (the default since round 6)
if kind == "rsnt_bf16":
    plan.set_option(api.OPT_MATRIX_RESONATOR, 1)          # v_mfma_f32_32x32x16_bf16: the kernel that disturbs its neighbours (NOTES.md round 6)
plan.upload()
out = plan.render(x)
print("READY", flush=True)
t0 = time.time()
while time.time() - t0 < seconds:
    for _ in range(32):
        plan.render(x, rgba=out)
    torch.cuda.synchronize()
