"""The multi-GPU time-chunk path with the REAL HIP stages (GpuBackend through the C ABI), two ranks sharing one GPU:
the collectives run over gloo (RCCL refuses two ranks on one device), everything else is the product path.
The sharded render must be bit-identical to a single-device render of the whole stream."""
import os
import socket

import numpy as np
import pytest

from signalizer_amd import config, synth

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, cfg, S, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from signalizer_amd import api
    from signalizer_amd.sharding import TimeChunkRenderer
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    full = synth.gen(78, 48000, S * world, 2 * cfg["num_pairs"])
    chunk = torch.from_numpy(full[:, rank * S:(rank + 1) * S].copy()).to(dev)
    plan = api.Plan(cfg).upload()
    r = TimeChunkRenderer(plan, chunk, rank=rank, world=world)
    out = r.render()[:r.local_frames]
    torch.cuda.synchronize()
    out2 = r.render()[:r.local_frames]              # a second pass must give the same columns (no stale carry)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    q.put((rank, out.cpu().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("window,hop,pairs,S,world,mode", [
    (4096, 1024, 1, 4096 * 5 + 300, 2, config.CH_SEPARATE), (32768, 8192, 1, 32768 * 3 + 1234, 2, config.CH_SEPARATE),
    (8192, 2048, 2, 8192 * 4 + 77, 3, config.CH_SEPARATE),            # halves path, three ranks
    (2048, 700, 1, 2048 * 6 + 5, 2, config.CH_MERGE),                # generic path, a mono mode, hop not dividing the chunk
    (4096, 1024, 3, 4096 * 3 + 1, 4, config.CH_MIDSIDE),              # four ranks, three pairs
    (65536, 16384, 4, 65536 * 3 + 999, 2, config.CH_SEPARATE)])       # cfg5's transform (N = 65536, halves path), sharded
def test_sharded_render_equals_single_device(gpu, window, hop, pairs, S, world, mode):
    import torch
    import torch.multiprocessing as mp
    from signalizer_amd import api
    cfg = config.spectrum_config(window_size=window, hop=hop, num_pairs=pairs, axis_points=300, pole=(0.97, 0.5), channel_mode=mode)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cfg, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = torch.from_numpy(synth.gen(78, 48000, S * world, 2 * pairs)).to(gpu)
    ref = api.Plan(cfg).upload().render(full).cpu().numpy()
    out = np.concatenate([got[r] for r in range(world)])
    assert out.shape == ref.shape
    assert np.array_equal(out, ref)
