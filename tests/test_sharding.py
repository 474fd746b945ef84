"""Time-chunk sharding: partition arithmetic, and the N>1 exchange path run with
world_size 2 over gloo on CPU.  On CPU the per-stage compute is supplied by the oracle (TEST stand-in for
the HIP kernels -- the choreography under test is signalizer_amd.sharding: neighbour halo send / recv (or the
all-gather form), zero-carry scan, end-state all-gather, exact carry fold, emit with the carry)."""
import os
import socket

import numpy as np
import pytest

from signalizer_amd import config, synth
from signalizer_amd.sharding import ShardPlan


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("S,W,hop", [(2880000, 32768, 8192), (5760000, 65536, 16384), (1920000, 65536, 16384), (10000, 4096, 1000), (4096, 4096, 4096)])
def test_partition_covers_every_global_frame_once(world, S, W, hop):
    sps = [ShardPlan(r, world, S, W, hop) for r in range(world)]
    total = (world * S - W) // hop + 1
    assert sum(s.local_frames for s in sps) == total == sps[0].total_frames
    f = 0
    for s in sps:
        assert s.first_frame(s.rank) == f
        if s.local_frames:
            assert 0 <= s.local_offset < hop
            start = s.rank * S + s.local_offset
            assert start == f * hop
            assert s.halo <= W and s.local_offset + s.local_samples <= S + s.halo
        f += s.local_frames
    assert sps[-1].halo == 0


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("S,W,hop", [(2880000, 32768, 8192), (5760000, 65536, 16384), (1920000, 65536, 16384), (10000, 4096, 1000), (4096, 4096, 4096)])
def test_c_abi_shard_layout_equals_shard_plan(world, S, W, hop):
    """sgz_shard_layout (csrc/sharded.hip: the partition sgz_spectrogram_render_sharded runs on) against ShardPlan (the Python twin the
    gloo test and bench.py's torch path run on), over the same sweep: frames, first frame, halo in / out of every rank.  Host arithmetic
    only -- no GPU."""
    import ctypes as C
    from signalizer_amd import api
    cfg = config.spectrum_config(window_size=W, hop=hop, sample_rate=96000.0)
    plan = api.Plan(cfg)
    halos = [ShardPlan(r, world, S, W, hop).halo for r in range(world)]
    for r in range(world):
        sp = ShardPlan(r, world, S, W, hop)
        vals = [C.c_uint64() for _ in range(4)]
        api.check(api.lib().sgz_shard_layout(plan.h, r, world, S, *[C.byref(v) for v in vals]))
        local, first, halo_in, halo_out = [int(v.value) for v in vals]
        assert (local, first) == (sp.local_frames, sp.first_frame(r)), (r, local, first)
        assert halo_in == sp.halo, (r, halo_in, sp.halo)
        assert halo_out == (halos[r - 1] if r > 0 else 0), (r, halo_out)


def test_bench_workload_partition():
    sp = [ShardPlan(r, 8, 2880000, 32768, 8192) for r in range(8)]
    assert [s.local_frames for s in sp] == [352, 352, 351, 352, 351, 352, 351, 348]
    assert ShardPlan(0, 1, 2880000, 32768, 8192).local_frames == 348


class OracleBackend:
    """CPU stand-in for the HIP stages, used only by this test."""

    def __init__(self, cfg):
        from oracle import pyoracle as po
        self.po = po
        self.cfg = cfg
        self.p = po.params_from_dict(cfg)
        self.P, self.C = cfg["axis_points"], cfg["num_pairs"]

    def _mapped(self, x):
        import torch
        po = self.po
        xs = x.numpy()
        F = po.lib().sgzo_num_frames(xs.shape[1], self.p.window_size, self.p.hop)
        out = np.zeros((F, self.C, 2, self.P), np.float32)
        for f in range(F):
            for c in range(self.C):
                seg = xs[:, f * self.p.hop:f * self.p.hop + self.p.window_size]
                _, _, csp = po.frame_bins(self.p, seg[2 * c], seg[2 * c + 1])
                v = csp.reshape(2, self.P)
                out[f, c] = np.sqrt((v.real * v.real + v.imag * v.imag).astype(np.float32))
        return out

    def stage_mapped(self, x, mapped):
        import torch
        m = self._mapped(x)
        mapped[:m.shape[0]].copy_(torch.from_numpy(m))

    def stage_decay_colour(self, mapped, frames, rgba, state):
        import torch
        po = self.po
        st = state.numpy()                                            # [C][G][P][2]
        m = mapped.numpy()
        for f in range(frames):
            fr = np.zeros((self.C, self.P), np.complex64)
            for c in range(self.C):
                csp = np.zeros(2 * self.P, np.complex64)
                csp.real = m[f, c].reshape(-1)
                states = np.ascontiguousarray(st[c, :, :, 0] + 1j * st[c, :, :, 1]).astype(np.complex64)
                res = po.filters(self.p, csp, states)
                st[c, :, :, 0], st[c, :, :, 1] = states.real, states.imag
                fr[c] = res[0]
            if rgba is not None:                                      # None: state-only pass (the decay carry exchange)
                rgba[f].copy_(torch.from_numpy(po.blend_column(self.p, fr)))

    def render(self, x, rgba, state):
        import torch
        F = self.po.lib().sgzo_num_frames(x.shape[1], self.p.window_size, self.p.hop)
        mapped = torch.from_numpy(self._mapped(x))
        self.stage_decay_colour(mapped, F, rgba, state)

    def decay_scan(self, mapped, frames, end_state):
        end_state.zero_()
        self.stage_decay_colour(mapped, frames, None, end_state)

    def decay_emit(self, mapped, frames, carry, rgba):
        import torch
        state = carry.clone() if carry is not None else torch.zeros((self.C, 2, self.P, 2), dtype=torch.float32)
        self.stage_decay_colour(mapped, frames, rgba, state)

    def fold_carry(self, aggs, frames_per_rank, rank, carry):
        # same identity as sgz_decay_fold_carry: sequential fp32 decay, max with each predecessor's end state
        a = aggs.numpy()
        c = np.zeros_like(a[0])
        poles = np.array(self.cfg["pole"], np.float32)
        for q in range(rank):
            for _ in range(frames_per_rank[q]):
                c = (c * poles[None, :, None, None]).astype(np.float32)
            c = np.maximum(c, a[q])
        carry.copy_(__import__("torch").from_numpy(c))


class _FakePlan:
    def __init__(self, cfg):
        self.P, self.C, self.sides = cfg["axis_points"], cfg["num_pairs"], 2

        class _C:
            window_size, hop = cfg["window_size"], cfg["hop"]
        self.cfg = _C


def _worker(rank, world, port, cfg, S, q, halo="p2p"):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from signalizer_amd.sharding import TimeChunkRenderer
    full = synth.gen(77, 48000, S * world, 2 * cfg["num_pairs"])
    chunk = torch.from_numpy(full[:, rank * S:(rank + 1) * S].copy())
    r = TimeChunkRenderer(_FakePlan(cfg), chunk, rank=rank, world=world, backend=OracleBackend(cfg), halo=halo)
    out = r.render()[:r.local_frames].numpy().copy()
    assert r.time_collectives(iters=2) > 0.0          # bench.py's collective-share probe runs on this backend too
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("halo,world", [("p2p", 2), ("allgather", 2), ("p2p", 3)])
def test_gloo_render_equals_single_process(halo, world):
    import torch.multiprocessing as mp
    from oracle import pyoracle as po
    cfg = config.spectrum_config(window_size=512, hop=96, axis_points=40, num_pairs=2, pole=(0.97, 0.5))
    S = 1500                                # S is not a multiple of hop: frames straddle the chunk boundary
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cfg, S, q, halo)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = synth.gen(77, 48000, S * world, 4)
    ref = po.spectrogram(po.params_from_dict(cfg), full)["rgba"]
    out = np.concatenate([got[r] for r in range(world)])
    assert out.shape == ref.shape
    # the decay carry is folded exactly, so the sharded result is bit-identical to the single-process one
    assert np.array_equal(out, ref)
