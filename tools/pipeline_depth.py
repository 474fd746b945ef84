"""How much of the 348-frame launch's tail does pipelining hide?  Independent renders of cfg2 buffers (one plan + one stream each -- a plan
owns its scratch) submitted round-robin over `depth` streams, no host wait in between: ms per render for depth 1, 2, 3, 4, 6, 8, beside the
tail-free K_A rate (what a machine that is always full would do).  Host-paced from Python (a render is one C call, two launches).
usage: pipeline_depth.py [renders per measurement]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from signalizer_amd import api, config, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
cfg = config.cfg2()
S = int(60 * 48000)
dev = torch.device("cuda", 0)
DEPTHS = tuple(int(v) for v in os.environ.get('SGZ_DEPTHS', '1,2,3,4,6,8').split(','))
PX = int(os.environ.get("SGZ_KB_PX", "4"))                    # pixels per workgroup of the fused K_B (SGZ_OPT_FUSED_COLOUR = 4 / 8 / 16)
KA_ONLY = os.environ.get("SGZ_KA_ONLY") == "1"                # K_A's dominant launch alone (sgz_stage_mapped_dominant)
plans = []
for _ in range(max(DEPTHS)):
    pl = api.Plan(cfg)
    pl.set_option(api.OPT_FUSED_COLOUR, PX)
    plans.append(pl.upload())
print(f"fused K_B pixels per workgroup: {PX}; K_A only: {KA_ONLY}")
# 13 distinct input copies (288 MB: past the Infinity Cache, bench.py's protocol)
xs = [torch.from_numpy(synth.gen(2 + k, 48000, S, 2)).to(dev) for k in range(13)]
F = plans[0].num_frames(S)
outs = [torch.empty((F, 1024, 4), dtype=torch.uint8, device=dev) for _ in range(max(DEPTHS))]
streams = [torch.cuda.Stream(device=dev) for _ in range(max(DEPTHS))]
mapped = [torch.empty((F, 1, 2, 1024), dtype=torch.float32, device=dev) for _ in range(max(DEPTHS))]
# SGZ_DUMMY_STREAMS=k: k other streams exist (and have run something) before the lanes are made -- the runtime maps streams onto a few
# hardware queues (GPU_MAX_HW_QUEUES, default 4), and two lanes that share one run one after the other
dummies = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("SGZ_DUMMY_STREAMS", "0")))]
for ds in dummies:
    with torch.cuda.stream(ds):
        torch.zeros(16, device=dev).add_(1)
torch.cuda.synchronize()
print(f"dummy streams: {len(dummies)}; GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}")
QUEUE = os.environ.get("SGZ_QUEUE") == "1"                    # through sgz_render_queue instead of plans / streams of this tool
for d in DEPTHS:
    if QUEUE:
        q = api.RenderQueue(cfg, d)
        print(f"   (depth {d}: {q.distinct_lanes} lanes on hardware queues of their own)")
        def go(count):
            for i in range(count):
                q.submit(xs[i % 13], outs[i % d])
    else:
      def go(count):
        for i in range(count):
            k = i % d
            if KA_ONLY:
                x = xs[i % 13]
                api.check(api.lib().sgz_stage_mapped_dominant(plans[k].h, x.data_ptr(), x.stride(0), S, mapped[k].data_ptr(), streams[k].cuda_stream))
            else:
                plans[k].render(xs[i % 13], rgba=outs[k], stream=streams[k].cuda_stream)
    go(1500)                                                      # spin the clock up
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        go(n)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        res.append(((time.perf_counter() - t0) / n * 1e6, host / n * 1e6))
    us, host_us = min(res)
    if QUEUE: q.close()
    print(f"depth {d}: {us:6.2f} us per render ({F / us:5.2f} M frames/s); host enqueue {host_us:5.2f} us per render")
