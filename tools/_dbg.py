import sys; sys.path.insert(0, '.')
import numpy as np, torch
from signalizer_amd import api, config, synth
sys.path.insert(0, 'tests')

from oracle import pyoracle as po; po.build()
for W, mode in ((65536, config.CH_COMPLEX), (32768, config.CH_COMPLEX), (16384, config.CH_COMPLEX), (8192, config.CH_COMPLEX)):
    cfg = config.spectrum_config(window_size=W, hop=W // 4, channel_mode=mode)
    x = synth.gen(11, 48000, W, 2)
    p = po.params_from_dict(cfg)
    plan = api.Plan(cfg).upload()
    bins = plan.stage_bins(torch.from_numpy(x).cuda()).cpu().numpy()[0, 0]
    raw, csf, csp = po.frame_bins(p, x[0, :W], x[1, :W])
    ref = csf.real
    d = np.abs(bins - ref)[: W // 2]
    k = int(d.argmax())
    print(W, "max err", d.max(), "at", k, bins[k], ref[k], "n bad", int((d > 4e-6 * np.abs(ref).max()).sum()), np.nonzero(d > 4e-6 * np.abs(ref).max())[0][:10])
