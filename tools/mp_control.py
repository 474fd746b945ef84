"""Control experiment for the multi-process stress (tools/overlap_stress_cfgs.py LOAD=n): the same three load processes, but the work
under test is NOT this library's -- plain PyTorch operators (rocFFT real transform, a matrix product, an elementwise chain, a sort) on fixed
inputs, each result compared bit for bit with the quiet run's.  If these differ at a similar rate, the rare wrong workgroup under
multi-process time-slicing is the platform's (driver / firmware / hardware), not a race in the library's kernels.
LOADKIND=torch: the load processes run PyTorch operators only as well (rocFFT + matrix products): nothing of this library on the device.
usage: [LOAD=3] [LOADKIND=torch] mp_control.py [iterations]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
load = int(os.environ.get("LOAD", "3"))
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn((64, 32768), generator=g).to(dev)
a = torch.randn((1024, 1024), generator=g).to(dev)
b = torch.randn((1024, 1024), generator=g).to(dev)
ops = {
    "rfft 64 x 32768 (rocFFT)": lambda: torch.view_as_real(torch.fft.rfft(x)),
    "matmul 1024^3 fp32": lambda: a @ b,
    "elementwise chain": lambda: torch.sin(x) * torch.cos(x * 1.7) + torch.sqrt(x.abs()),
    "sort 64 x 32768": lambda: torch.sort(x, dim=1).values,
    "cumsum 64 x 32768": lambda: torch.cumsum(x, dim=1),
}
if os.environ.get("OPS"):                                   # OPS=rfft: only the operators whose name contains one of the words
    ops = {k: f for k, f in ops.items() if any(w in k for w in os.environ["OPS"].split(","))}
want = {k: f().clone() for k, f in ops.items()}
torch.cuda.synchronize()
procs = []
if load:
    here = os.path.dirname(os.path.abspath(__file__))
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "gpu_load.py"), "1200", os.environ.get("LOADKIND") or ("rsnt" if k % 2 else "spectrum")], stdout=subprocess.PIPE, text=True) for k in range(load)]
    for p in procs:
        assert p.stdout.readline().strip() == "READY"
    what = os.environ.get("LOADKIND") or "renders of this library"
    print(f"{load} load processes ({what}) beside PyTorch's own operators", flush=True)
try:
    for name, f in ops.items():
        bad = 0
        for it in range(n):
            y = f()
            if not torch.equal(y.view(torch.int32), want[name].view(torch.int32)):
                bad += 1
                d = (y != want[name])
                print(f"    iteration {it}: {int(d.sum())} of {d.numel()} values differ", flush=True)
        print(f"{name:28s}: {bad} of {n} results differ from the quiet run", flush=True)
    assert all(p.poll() is None for p in procs)
finally:
    for p in procs:
        p.kill(); p.wait()
