import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
from oracle import pyoracle as po
for interp in (0, 1, 2):
    for view in (1, 0):
        cfg = config.spectrum_config(window_size=4096, hop=4096, bin_interp=interp, view_scaling=view, axis_points=700)
        p = po.params_from_dict(cfg)
        x = synth.gen(5, 48000, 4096 * 2, 2)
        plan = api.Plan(cfg).upload()
        raw, csf, csp = po.frame_bins(p, x[0, :4096], x[1, :4096])
        csfs = csf.real.astype(np.float32).reshape(1, 1, -1).copy()
        v = csp.reshape(2, plan.P)
        want = np.sqrt((v.real * v.real + v.imag * v.imag).astype(np.float32)).astype(np.float32)
        got = plan.stage_map_from_bins(torch.from_numpy(csfs).cuda()).cpu().numpy()[0, 0]
        bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))
        print("interp", interp, "view", view, "break", plan.break_pixel, "nbad", len(bad[0]), "sides", np.unique(bad[0]),
              "pix", bad[1][:10], "got", got[bad][:4], "want", want[bad][:4], "imag max", np.abs(v.imag).max())
