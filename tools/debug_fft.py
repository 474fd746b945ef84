import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config
for N in (4096, 32768):
    cfg = config.spectrum_config(window_size=N, hop=N, window_type=config.WIN_RECT, channel_mode=config.CH_COMPLEX,
                                 view_scaling=config.VIEW_LINEAR, bin_interp=config.INTERP_NONE)
    plan = api.Plan(cfg).upload()
    rng = np.random.default_rng(0)
    for name in ("tone", "impulse", "random"):
        x = np.zeros((2, N), np.float32)
        if name == "tone":
            k0 = 37
            x[0] = np.cos(2 * np.pi * k0 * np.arange(N) / N); x[1] = np.sin(2 * np.pi * k0 * np.arange(N) / N)
        elif name == "impulse":
            x[0, 5] = 1.0
        else:
            x = rng.standard_normal((2, N)).astype(np.float32)
        bins = plan.stage_bins(torch.from_numpy(x).cuda()).cpu().numpy()[0, 0]
        ref = np.abs(np.fft.fft(x[0].astype(np.float64) + 1j * x[1]))
        ref[0] *= 0.5
        got = bins[:N]
        err = np.abs(got - ref)
        print(N, name, "max err", err.max(), "ref max", ref.max(), "argmax got/ref", got.argmax(), ref.argmax(),
              "n bad", int((err > 1e-3 * ref.max()).sum()), "first bad", np.nonzero(err > 1e-3 * ref.max())[0][:8])
