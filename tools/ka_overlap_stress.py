"""Which part of several renders in flight can produce a wrong frame?  `depth` plans on `depth` streams, inputs fixed per buffer, outputs
compared with a quiet run's.  Modes: K_A alone (sgz_stage_mapped: K_A + the late-pixel kernel) / whole renders; with / without
SGZ_OPT_PIPELINED.      usage: ka_overlap_stress.py [rounds] [depth]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = config.cfg2(); S = 32768 + 8192 * 347; gpu = torch.device("cuda", 0)
xs = [torch.from_numpy(synth.gen(200 + k, 48000, S, 2)).to(gpu) for k in range(3)]
ref = api.Plan(cfg).upload()
want_m = [ref.stage_mapped(x).clone() for x in xs]
want_r = [ref.render(x).clone() for x in xs]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=gpu) for _ in range(depth)]
for pipelined in (0, 1):
    plans = []
    for _ in range(depth):
        pl = api.Plan(cfg); pl.set_option(api.OPT_PIPELINED, pipelined); plans.append(pl.upload())
    for what in ("mapped",):
        bad = 0
        for r in range(rounds):
            outs = []
            torch.cuda.synchronize()
            for k in range(9):
                lane = k % depth
                with torch.cuda.stream(streams[lane]):
                    if what == "mapped":
                        outs.append(plans[lane].stage_mapped(xs[k % 3]))
                    else:
                        outs.append(plans[lane].render(xs[k % 3], stream=streams[lane].cuda_stream))
            torch.cuda.synchronize()
            for k in range(9):
                w = (want_m if what == "mapped" else want_r)[k % 3]
                if not torch.equal(outs[k].view(torch.uint8), w.view(torch.uint8)):
                    bad += 1
                    if bad <= 4:
                        d = (outs[k] != w)
                        idx = d.nonzero()
                        print(f"   {what} pipelined={pipelined} round {r} buffer {k}: {int(d.sum())} values differ, frames {sorted(set(idx[:, 0].tolist()))[:8]}, "
                              f"dims beyond frame: {[sorted(set(idx[:, j].tolist()))[:6] for j in range(1, idx.shape[1])]}")
                        if what == "mapped":
                            f0 = int(idx[0, 0]); sd = int(idx[0, 2])
                            g, w_ = outs[k][f0, 0, sd].float().cpu().numpy(), w[f0, 0, sd].float().cpu().numpy()
                            px = np.nonzero(g != w_)[0]
                            rel = np.abs(g[px] - w_[px]) / np.maximum(np.abs(w_[px]), 1e-30)
                            print(f"      pixels {px[:12].tolist()} ... {px[-6:].tolist()}; rel diff min {rel.min():.2e} median {np.median(rel):.2e} max {rel.max():.2e}; "
                                  f"got/want at first: {g[px[0]]:.6e} / {w_[px[0]]:.6e}; nan {int(np.isnan(g).sum())}")
                            np.save(f"gpurun_out/ka_bad_{pipelined}_{r}_{k}.npy", np.stack([g, w_]))
        print(f"{what:7s} pipelined={pipelined}: {bad} of {rounds * 9} outputs differ")
