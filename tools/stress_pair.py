"""the channel workgroups' pair exchange under load: the same channel-split render many times, every result identical to the first
(a lost settlement, a stale published value or a torn write would show as a differing pixel).  usage: stress_pair.py [N] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SGZ_CHANNEL_SPLIT"] = "1"
import numpy as np, torch
from signalizer_amd import api, config, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
sr = {16384: 48000, 32768: 48000, 65536: 96000}[N]
pairs = 4
cfg = config.spectrum_config(sample_rate=float(sr), window_size=N, hop=N // 4, num_pairs=pairs)
x = torch.from_numpy(synth.gen(5, sr, N + 200 * (N // 4), 2 * pairs)).cuda()
plan = api.Plan(cfg).upload()
assert plan.path & 8
ref = plan.stage_mapped(x).clone()
torch.cuda.synchronize()
os.environ["SGZ_CHANNEL_SPLIT"] = "0"
whole = api.Plan(cfg).upload()
assert not whole.path & 8
bad = 0
for r in range(reps):
    m = plan.stage_mapped(x)
    torch.cuda.synchronize()
    if not torch.equal(m.view(torch.int32), ref.view(torch.int32)):
        bad += 1
        d = (m.view(torch.int32) != ref.view(torch.int32)).nonzero()
        print("rep", r, "differs at", d[:5].tolist(), int(d.shape[0]))
print(f"N {N}: {reps - bad} of {reps} repetitions identical; tasks per render {ref.shape[0] * pairs * 2}")
# and the settled pixels agree with the whole-frame kernel's (same bins within the FFT tolerance -> compare loosely)
w = whole.stage_mapped(x)
torch.cuda.synchronize()
rel = ((w - ref).abs().max() / ref.abs().max()).item()
print("max |split - whole| / max:", rel)
