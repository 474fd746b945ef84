"""Does any result depend on what the LDS or the vector registers held when a workgroup started?  Neither is cleared between workgroups:
a fresh workgroup sees what earlier workgroups -- of any kernel, of any process -- left there.  Every case is run after the whole chip's
LDS and register files were filled with zeros and again after they were filled with 0xFFFFFFFF (NaNs), 0x7F800000 (+Inf), 0x7F7FFFFF
(tools/ubench/lds_poison.hip, vgpr_poison.hip); any difference is a read of an LDS word or a register the kernel never wrote.  (Found with other PROCESSES rendering beside the tests: their leftovers are what a workgroup
inherits then -- tools/mp_control.py showed the same for rocFFT.)
usage: lds_poison_probe.py [rounds]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from signalizer_amd import api, config as cf, synth
here = os.path.dirname(os.path.abspath(__file__))
for lib, src in (("liblds_poison.so", "lds_poison.hip"), ("libvgpr_poison.so", "vgpr_poison.hip")):      # (two tiny kernels: built on first use)
    if not os.path.exists(os.path.join(here, "ab", lib)):
        import subprocess
        os.makedirs(os.path.join(here, "ab"), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(here, "ubench", src), "-o", os.path.join(here, "ab", lib)], check=True)
P = C.CDLL(os.path.join(here, "ab", "liblds_poison.so"))
P.lds_poison.argtypes = [C.c_uint32, C.c_void_p]
gpu = torch.device("cuda", 0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6


R = C.CDLL(os.path.join(here, "ab", "libvgpr_poison.so"))
R.vgpr_poison.argtypes = [C.c_uint32, C.c_void_p]
R.vgpr_peek.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]


def poison(pattern):
    """the LDS of every CU and the vector registers (arch + acc) of every SIMD hold `pattern` for whoever starts next"""
    torch.cuda.synchronize()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert P.lds_poison(pattern, s) == 0
    assert R.vgpr_poison(pattern, s) == 0
    torch.cuda.synchronize()


P.lds_peek.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
for pattern in (0xFFFFFFFF, 0x7F800000):                    # the tool's own check: fresh workgroups DO find the pattern
    poison(pattern)
    cnt = torch.zeros(2, dtype=torch.int64, device=gpu)
    assert P.lds_peek(pattern, C.c_void_p(cnt.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    cnt2 = torch.zeros(2, dtype=torch.int64, device=gpu)
    assert R.vgpr_peek(pattern, C.c_void_p(cnt2.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    print(f"after a fill with {pattern:#010x}: workgroups that only read find it in {cnt[0].item() / cnt[1].item():.3f} of their LDS words, "
          f"waves in {cnt2[0].item() / max(1, cnt2[1].item()):.3f} of the registers they never wrote", flush=True)
if os.environ.get("CONTROL") == "1":                        # the same question of PyTorch's rocFFT transform (the multi-process control, tools/mp_control.py)
    g = torch.Generator(device="cpu").manual_seed(5)
    xt = torch.randn((64, 32768), generator=g).to(gpu)
    poison(0)
    want = torch.view_as_real(torch.fft.rfft(xt)).clone()
    bad = 0
    for r in range(200):
        for pattern in (0xFFFFFFFF, 0x7F800000, 0x7F7FFFFF, 0x3C003C00, 0):
            poison(pattern)
            bad += 0 if torch.equal(torch.view_as_real(torch.fft.rfft(xt)).view(torch.int32), want.view(torch.int32)) else 1
    print(f"rocFFT rfft 64 x 32768: {bad} of 1000 results differ after an LDS / register fill", flush=True)
    sys.exit(0)
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
    "generic N=1024 linear": dict(window_size=1024, hop=256, bin_interp=cf.INTERP_LINEAR),
    "phase N=4096": dict(window_size=4096, hop=1024, channel_mode=cf.CH_PHASE),
    "phase N=32768": dict(channel_mode=cf.CH_PHASE),
    "rsnt hop 1024 (matrix cores)": dict(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1024),
    "rsnt hop 1000 (vector form)": dict(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1000),
    "three pairs N=4096 (scan/emit K_B)": dict(window_size=4096, hop=1024, num_pairs=3),
    "fetch window blackman N=32768": dict(window_type=cf.WIN_BLACKMAN),
    "lanczos interpolation N=4096": dict(window_size=4096, hop=1024, bin_interp=cf.INTERP_LANCZOS),
    "few pixels N=32768": dict(axis_points=77),
}
total = 0
for name, over in CASES.items():
    over = dict(over)
    wide = over.pop("_wide", 0)
    cfg = cf.spectrum_config(**over)
    frames = 60
    S = cfg["window_size"] + cfg["hop"] * (frames - 1)
    x = torch.from_numpy(synth.gen(500, int(cfg["sample_rate"]), S, 2 * cfg["num_pairs"])).to(gpu)
    plan = api.Plan(cfg)
    if wide:
        plan.set_option(api.OPT_WIDE_GROUPS, 1)
    plan.upload()
    F = plan.num_frames(S)

    def run():
        state = torch.zeros((cfg["num_pairs"], 2, plan.P, 2), dtype=torch.float32, device=gpu)
        lines = torch.zeros((F, cfg["num_pairs"], 2, plan.P, 2), dtype=torch.float32, device=gpu)
        if cfg["algorithm"]:
            plan.reset_resonator()
        rgba = plan.render(x, lines=lines, state=state).clone()
        if cfg["algorithm"]:
            plan.reset_resonator()
        m = plan.stage_mapped(x).view(torch.int32).clone()
        img = plan.render(x).clone()
        torch.cuda.synchronize()
        return m, rgba, lines.view(torch.int32), state.view(torch.int32), img
    poison(0)
    want = run()
    bad = [0] * 5
    for r in range(rounds):
        for pattern in (0xFFFFFFFF, 0x7F800000, 0x7F7FFFFF, 0):
            poison(pattern)
            got = run()
            for k in range(5):
                bad[k] += 0 if torch.equal(got[k], want[k]) else 1
    total += sum(bad)
    print(f"{name:38s}: differing after an LDS / register fill -- mapped {bad[0]}, image+lines+state render: image {bad[1]} lines {bad[2]} state {bad[3]}, image-only render {bad[4]}   (of {4 * rounds} each)", flush=True)
print("dependences on the earlier contents of the LDS or the registers:", total)
sys.exit(1 if total else 0)
