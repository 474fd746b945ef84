"""Stress of sgz_render_queue against single renders at the bench size: `reps` rounds of 9 buffers over 3 lanes, every image compared byte
for byte.  mode "fill": the output tensors are zero-filled on torch's stream right before the submits and NOT waited for (the way the
first version of the tests did it -- a fill that runs late wipes an image: the test's race, not the queue's); mode "sync": a device
synchronisation between the fills and the submits.      usage: queue_stress.py [reps] [depth]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = config.cfg2(); S = 32768 + 8192 * 347; gpu = torch.device("cuda", 0)
plan = api.Plan(cfg).upload()
xs = [torch.from_numpy(synth.gen(200 + k, 48000, S, 2)).to(gpu) for k in range(3)]
want = [plan.render(x).clone() for x in xs]
torch.cuda.synchronize()
for mode in ("sync", "fill"):
    q = api.RenderQueue(cfg, depth)
    if os.environ.get("SGZ_Q_PX"):
        q.set_option(api.OPT_FUSED_COLOUR, int(os.environ["SGZ_Q_PX"]))
    bad = zeros = 0
    for r in range(reps):
        outs = [torch.zeros((348, 1024, 4), dtype=torch.uint8, device=gpu) for _ in range(9)]
        if mode == "sync":
            torch.cuda.synchronize()
        for k in range(9):
            q.submit(xs[k % 3], outs[k])
        q.wait()
        torch.cuda.synchronize()
        for k in range(9):
            if not torch.equal(outs[k], want[k % 3]):
                bad += 1
                d = (outs[k] != want[k % 3])
                zeros += int(((outs[k] == 0) & d).sum() == d.sum())
                if bad <= 6:
                    idx = d.nonzero()
                    fr, px = idx[:, 0], idx[:, 1]
                    diff = (outs[k].int() - want[k % 3].int()).abs()
                    print(f"   round {r} buffer {k} (lane {k % depth}, input {k % 3}): {int(d.sum())} bytes differ, frames {int(fr.min())}..{int(fr.max())} "
                          f"({len(set(fr.tolist()))} distinct), pixels {int(px.min())}..{int(px.max())} ({len(set(px.tolist()))} distinct), max |diff| {int(diff.max())}")
    q.close()
    print(f"mode {mode}: {bad} of {reps * 9} images differ ({zeros} of them only by zeroed bytes)")
