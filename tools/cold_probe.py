"""Is the FIRST launch of a kernel in a process bit-identical to the later ones?  (a race inside a kernel shows when its waves run
skewed -- cold instruction cache)  usage: cold_probe.py [W] [frames] [pairs] [mode]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from signalizer_amd import api, config, synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 5
cfg = config.spectrum_config(window_size=W, hop=W, axis_points=300, num_pairs=pairs, channel_mode=mode)
x = torch.from_numpy(synth.gen(44, 48000, F * W, 2 * pairs)).cuda()
plan = api.Plan(cfg).upload()
outs = []
for i in range(4):
    m = plan.stage_mapped(x).cpu().numpy()
    outs.append(m)
for i in range(1, 4):
    d = np.argwhere(outs[i].view(np.uint32) != outs[0].view(np.uint32))
    print(f"K_A launch 0 vs {i}: {len(d)} differ", (sorted(set(d[:, 0])), sorted(set(d[:, 2])), sorted(set(d[:, 3]))[:3], sorted(set(d[:, 3]))[-3:]) if len(d) else "")
plan2 = api.Plan(cfg).upload()
r = []
for i in range(3):
    lines = torch.empty((F, pairs, 2, 300, 2), dtype=torch.float32, device="cuda")
    rgba = plan2.render(x, lines=lines).cpu().numpy(); r.append((rgba, lines.cpu().numpy()))
for i in range(1, 3):
    print(f"render 0 vs {i}: rgba {int((r[i][0] != r[0][0]).sum())} lines {int((r[i][1].view(np.uint32) != r[0][1].view(np.uint32)).sum())}")
