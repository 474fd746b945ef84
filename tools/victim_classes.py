"""Which instruction class of a bystander kernel is disturbed while the bf16 matrix-core RSNT kernel runs on another stream?  Synthetic
victims (tools/ubench/victims.hip: register-only chains of one instruction class each, and an LDS transpose loop), every launch compared
with the quiet run.  usage: [SGZ_LIB=variant] victim_classes.py [launches per class]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signalizer_amd import api, config as cf, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
V = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", "libvictims.so"))
V.victim_run.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
gpu = torch.device("cuda", 0)
rc = cf.spectrum_config(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1024)
xr = torch.from_numpy(synth.gen(9, 48000, 4096 + 1024 * 199, 2)).to(gpu)
rp = api.Plan(rc)
rp.set_option(api.OPT_MATRIX_RESONATOR, 1)                   # the bf16 kernel (opt-in since round 6)
rp.upload()
rout = rp.render(xr)
s1, s2 = torch.cuda.Stream(device=gpu), torch.cuda.Stream(device=gpu)
V.victim_load.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
gsrc = torch.Generator(device="cpu").manual_seed(3)
small = torch.randint(0, 2**31 - 1, (1 << 18,), generator=gsrc, dtype=torch.int32).to(gpu)          # 1 MB
big = torch.randint(0, 2**31 - 1, (1 << 26,), generator=gsrc, dtype=torch.int32).to(gpu)            # 256 MB
names = ["v_pk_fma_f32 chains", "v_fma_f32 chains", "LDS transposes (ds_write_b64 / ds_read_b64, barriers)", "v_pk_mul_f32 + v_pk_add_f32", "64-bit integer multiply-adds"]
BLOCKS, ITERS = 1024, 600
names += ["global loads, 1 MB buffer (caches)", "global loads, 256 MB buffer (HBM)"]
if os.environ.get("KINDS"):
    only = [int(k) for k in os.environ["KINDS"].split(",")]
else:
    only = list(range(len(names)))
def launch(kind, o):
    if kind < 5:
        return V.victim_run(kind, BLOCKS, ITERS, C.c_void_p(o.data_ptr()), C.c_void_p(s2.cuda_stream))
    b = small if kind == 5 else big
    return V.victim_load(C.c_void_p(b.data_ptr()), b.numel() // 4, BLOCKS, 128, C.c_void_p(o.data_ptr()), C.c_void_p(s2.cuda_stream))
for kind, name in enumerate(names):
    if kind not in only:
        continue
    want = torch.zeros(BLOCKS * 256, dtype=torch.int32, device=gpu)
    assert launch(kind, want) == 0
    torch.cuda.synchronize()
    bad = lanes = 0
    for it in range(0, n, 8):
        outs = [torch.zeros(BLOCKS * 256, dtype=torch.int32, device=gpu) for _ in range(8)]
        torch.cuda.synchronize()
        for o in outs:
            rp.render(xr, rgba=rout, stream=s1.cuda_stream)
            rp.render(xr, rgba=rout, stream=s1.cuda_stream)
            assert launch(kind, o) == 0
        torch.cuda.synchronize()
        for o in outs:
            d = int((o != want).sum())
            bad += d > 0
            lanes += d
    print(f"{name:58s}: {bad} of {n} launches differ ({lanes} lanes in all)", flush=True)
