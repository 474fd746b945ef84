"""Chunk-scan map against the whole-frame kernel's map on the same audio (cfg2-like): mismatching pixels of `mapped`.
usage: debug_chunkmap.py [N] [frames]"""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
def run(split):
    code = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, %r)
from signalizer_amd import api, config, synth
N = int(sys.argv[1]); F = int(sys.argv[2])
cfg = config.spectrum_config(window_size=N, hop=N // 4)
S = N + (F - 1) * (N // 4)
x = torch.from_numpy(synth.gen(2, 48000, S, 2)).cuda()
plan = api.Plan(cfg).set_option(api.OPT_CHANNEL_SPLIT, int(os.environ['SGZ_SPLIT'])).upload()
print("path", plan.path, file=sys.stderr)
m = plan.stage_mapped(x).cpu().numpy()
np.save(sys.argv[3], m)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = f"/tmp/m_{split}.npy"
    env = dict(os.environ, SGZ_SPLIT=str(split))
    subprocess.run([sys.executable, "-c", code, sys.argv[1] if len(sys.argv) > 1 else "32768", sys.argv[2] if len(sys.argv) > 2 else "8", out], env=env, check=True)
    return np.load(out)
a, b = run(0), run(1)
bad = np.nonzero(np.abs(a - b) > 1e-4 * np.abs(a).max())
print("shape", a.shape, "nbad", len(bad[0]))
for i in range(min(40, len(bad[0]))):
    idx = tuple(int(v[i]) for v in bad)
    print(idx, "whole-frame", a[idx], "channel-split", b[idx])
