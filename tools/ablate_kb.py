"""K_B timing: single-launch path vs the three-kernel path (forced with debug bit 0x1000)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signalizer_amd import api, config, synth, sharding
cfg = config.cfg2()
x = torch.from_numpy(synth.gen(2, 48000, int(60 * 48000), 2)).cuda()
plan = api.Plan(cfg).upload()
r = sharding.TimeChunkRenderer(plan, x)
L = api.lib()
def step_time(iters=50):
    for _ in range(5): r.render()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): r.render()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for bits, name in {0: "single-launch K_B", 0x1000: "three-kernel K_B"}.items():
    L.sgz_debug_set_ablate(bits)
    print(f"{name:32s} step {min(step_time() for _ in range(3)):8.1f} us")
L.sgz_debug_set_ablate(0)
