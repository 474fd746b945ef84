"""channel-split K_A (spectrum_real.hip) against numpy: csf magnitudes of one frame, worst bins.  usage: debug_real.py [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
cfg = config.spectrum_config(window_size=N, hop=N // 4, sample_rate=48000.0)
x = synth.gen(2, 48000, N + N // 4, 2)
plan = api.Plan(cfg).upload()
bins = plan.stage_bins(torch.from_numpy(x).cuda()).cpu().numpy()       # [F][1][N+1]
w = plan.window().astype(np.float64)
for f in range(bins.shape[0]):
    seg = x[:, f * (N // 4):f * (N // 4) + N].astype(np.float64)
    XL = np.fft.fft(seg[0] * w); XR = np.fft.fft(seg[1] * w)
    M = N // 2
    ref = np.zeros(N + 1)
    ref[:M + 1] = np.abs(XL[:M + 1]); ref[M:] = np.abs(XR[:M + 1])[::-1]
    ref[0] = 0.5 * XL[0].real; ref[N] = 0.5 * XR[0].real
    ref[M] = 0.5 * np.hypot(XL[M].real, XR[M].real); ref[M - 1] *= 0.5
    got = bins[f, 0].astype(np.float64)
    err = np.abs(got - ref)
    bad = np.nonzero(err > 4e-6 * np.abs(ref).max())[0]
    print("frame", f, "max err", err.max(), "scale", np.abs(ref).max(), "bad bins", len(bad), bad[:20])
    if len(bad):
        k = bad[:8]
        print("   got", got[k]); print("   ref", ref[k])
