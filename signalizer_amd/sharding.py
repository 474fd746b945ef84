"""Time-chunk sharding of long spectrogram renders across the GPUs of one node.

The global stream is the concatenation of every rank's chunk (rank r holds samples [r*S, (r+1)*S)).
A frame belongs to the rank that owns its first sample.  Per render:
  A1  halo: rank r needs the samples its last frames reach into rank r+1's chunk (W - hop when hop divides S; ShardPlan.halo) --
      a neighbour send / recv of exactly those samples (one xGMI link each way), or (halo="allgather", the north star's wording) an
      all-gather of every rank's leading max-halo samples
  K_A window x FFT x map over the local frames (no collective)
  K_B scan: chunk scans from a zero carry-in -> this rank's zero-carry end state (sgz_stage_decay_scan)
  A2  all-gather of the end states; exact carry fold (sgz_decay_fold_carry)
  K_B emit: the carry folded into the kept aggregates, dB, colour (sgz_stage_decay_emit) -- the magnitudes are scanned once
With world == 1 this degenerates to a single sgz_spectrogram_render_device call.
One process per GPU; torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" for the CPU plan tests).  A C++ host uses
sgz_spectrogram_render_sharded (include/sgz.h) on its own ncclComm_t instead: the same protocol without torch.
"""
from __future__ import annotations

import math
from dataclasses import dataclass


@dataclass
class ShardPlan:
    """Pure host arithmetic of the partition (tested on CPU with gloo, world_size 2)."""
    rank: int
    world: int
    chunk: int          # samples per rank S
    window: int         # W
    hop: int

    @property
    def total_frames(self) -> int:
        total = self.chunk * self.world
        return 0 if total < self.window else (total - self.window) // self.hop + 1

    def first_frame(self, r: int) -> int:
        return min(self.total_frames, -(-(r * self.chunk) // self.hop))      # ceil(r*S/hop)

    def frames_of(self, r: int) -> int:
        return self.first_frame(r + 1) - self.first_frame(r) if r + 1 < self.world else self.total_frames - self.first_frame(r)

    @property
    def local_frames(self) -> int:
        return self.frames_of(self.rank)

    @property
    def local_offset(self) -> int:
        """offset of this rank's first frame inside its own chunk, in samples"""
        return self.first_frame(self.rank) * self.hop - self.rank * self.chunk

    @property
    def halo(self) -> int:
        """samples needed from the next rank's chunk"""
        if self.local_frames == 0:
            return 0
        last_end = (self.first_frame(self.rank) + self.local_frames - 1) * self.hop + self.window
        return max(0, last_end - (self.rank + 1) * self.chunk)

    @property
    def local_samples(self) -> int:
        """length of the buffer [local_offset, ...) that render_device must be given"""
        return 0 if self.local_frames == 0 else (self.local_frames - 1) * self.hop + self.window


class GpuBackend:
    """the product path: stages run as HIP kernels through the C ABI (libsgz.so)"""

    def __init__(self, plan):
        self.plan = plan

    def _stream(self):
        import torch
        return torch.cuda.current_stream().cuda_stream

    def render(self, x, rgba, state):
        self.plan.render(x, rgba=rgba, state=state)

    def stage_mapped(self, x, mapped):
        from . import api
        api.check(api.lib().sgz_stage_mapped(self.plan.h, x.data_ptr(), x.stride(0), x.shape[1], mapped.data_ptr(), self._stream()))

    def decay_scan(self, mapped, frames, end_state):
        from . import api
        api.check(api.lib().sgz_stage_decay_scan(self.plan.h, mapped.data_ptr(), frames, end_state.data_ptr(), self._stream()))

    def decay_emit(self, mapped, frames, carry, rgba):
        from . import api
        api.check(api.lib().sgz_stage_decay_emit(self.plan.h, mapped.data_ptr(), frames, carry.data_ptr() if carry is not None else None,
                                                 rgba.data_ptr(), None, None, self._stream()))

    def fold_carry(self, aggs, frames_per_rank, rank, carry):
        self.plan.fold_carry(aggs, frames_per_rank, rank, carry)


class TimeChunkRenderer:
    def __init__(self, plan, chunk_audio, rank: int = 0, world: int = 1, backend=None, halo: str = "p2p", always_collective: bool = False,
                 rotate_bytes: int = 0):
        import torch
        self.torch = torch
        self.plan = plan
        self.backend = backend if backend is not None else GpuBackend(plan)
        self.rank, self.world = rank, world
        self.halo_mode = halo
        self.collective = world > 1 or always_collective
        self.nch, S = chunk_audio.shape
        W, hop = plan.cfg.window_size, plan.cfg.hop
        assert S >= W, "chunk must hold at least one window"
        self.sp = ShardPlan(rank, world, S, W, hop)
        dev = chunk_audio.device
        halos = [ShardPlan(r, world, S, W, hop).halo for r in range(world)]
        self.halo_in = halos[rank]                                   # samples wanted from rank + 1
        self.halo_out = halos[rank - 1] if rank > 0 else 0           # samples rank - 1 wants from this rank's head
        self.halo_max = max(halos) if halos else 0
        # local buffer = own chunk followed by the next rank's leading samples
        # (row stride a multiple of 64 samples: the channel-split K_A fetches sample pairs and needs 8-byte aligned rows)
        self.buf = torch.zeros((self.nch, (S + max(self.halo_max, 1) + 63) // 64 * 64), dtype=torch.float32, device=dev)
        self.buf[:, :S] = chunk_audio
        # rotate_bytes > 0 (bench.py): consecutive renders read DISTINCT copies of the audio, enough of them to exceed `rotate_bytes` -- past
        # the 256 MB Infinity Cache every render then streams its input from HBM instead of re-reading a cache-resident buffer
        self._bufs = [self.buf]
        if rotate_bytes > 0 and self.buf.is_cuda:
            per = self.buf.numel() * 4
            self._bufs += [self.buf.clone() for _ in range(max(0, -(-rotate_bytes // per) - 1))]
        self._turn = 0
        self.S, self.W = S, W
        self.local_frames = self.sp.local_frames
        P, C = plan.P, plan.C
        self.rgba = torch.empty((max(self.local_frames, 1), P, 4), dtype=torch.uint8, device=dev)
        if self.collective:
            self.mapped = torch.empty((max(self.local_frames, 1), C, plan.sides, P), dtype=torch.float32, device=dev)
            self.send_halo = torch.empty((self.nch, max(self.halo_out, 1)), dtype=torch.float32, device=dev)
            self.recv_halo = torch.empty((self.nch, max(self.halo_in, 1)), dtype=torch.float32, device=dev)
            self.halo_send_all = torch.empty((self.nch, max(self.halo_max, 1)), dtype=torch.float32, device=dev)
            self.halo_all = torch.empty((world, self.nch, max(self.halo_max, 1)), dtype=torch.float32, device=dev)
            self.end_state = torch.zeros((C, 2, P, 2), dtype=torch.float32, device=dev)
            self.agg_all = torch.empty((world, C, 2, P, 2), dtype=torch.float32, device=dev)
            self.carry = torch.zeros((C, 2, P, 2), dtype=torch.float32, device=dev)
            self.frames_per_rank = [self.sp.frames_of(r) for r in range(world)]

    def _rotate(self):
        if len(self._bufs) > 1:
            self._turn = (self._turn + 1) % len(self._bufs)
            self.buf = self._bufs[self._turn]

    def _view(self):
        off = self.sp.local_offset
        return self.buf[:, off:off + self.sp.local_samples]

    def _exchange_halo(self):
        import torch.distributed as dist
        if self.halo_mode == "allgather":
            if self.halo_max == 0:
                return
            self.halo_send_all.copy_(self.buf[:, :self.halo_max])
            dist.all_gather_into_tensor(self.halo_all.view(-1), self.halo_send_all.view(-1))
            if self.rank + 1 < self.world and self.halo_in:
                self.buf[:, self.S:self.S + self.halo_in] = self.halo_all[self.rank + 1][:, :self.halo_in]
            return
        reqs = self._post_halo()
        self._finish_halo(reqs)

    def _post_halo(self):
        """neighbour form: queue the send of this rank's leading samples and the recv of the next rank's; returns the requests"""
        import torch.distributed as dist
        ops = []
        if self.rank > 0 and self.halo_out:
            self.send_halo.copy_(self.buf[:, :self.halo_out])
            ops.append(dist.P2POp(dist.isend, self.send_halo, self.rank - 1))
        if self.rank + 1 < self.world and self.halo_in:
            ops.append(dist.P2POp(dist.irecv, self.recv_halo, self.rank + 1))
        return dist.batch_isend_irecv(ops) if ops else []

    def _finish_halo(self, reqs):
        for req in reqs:
            req.wait()
        if self.rank + 1 < self.world and self.halo_in:
            self.buf[:, self.S:self.S + self.halo_in] = self.recv_halo

    @property
    def early_frames(self) -> int:
        """local frames whose window ends inside this rank's own chunk: they do not wait for the halo"""
        if not (self.rank + 1 < self.world and self.halo_in):
            return self.local_frames
        off = self.sp.local_offset
        return 0 if self.S < off + self.W else min(self.local_frames, (self.S - off - self.W) // self.sp.hop + 1)

    def render(self):
        """one full pass; returns this rank's RGBA8 columns [local_frames, P, 4]"""
        self._rotate()
        if not self.collective:
            # single device: start from a zero decay state, nobody needs the end state -> no memset, no snapshot
            self.backend.render(self._view(), self.rgba, None)
            return self.rgba
        import torch.distributed as dist
        F = self.local_frames
        if self.halo_mode == "allgather":
            self._exchange_halo()                                     # A1 (north star's wording), then K_A over every frame
            if F:
                self.backend.stage_mapped(self._view(), self.mapped)
        else:
            # A1 as a neighbour send / recv that runs WHILE K_A transforms the frames inside the chunk; only the last
            # ceil((W - hop) / hop) frames -- the ones that reach into the next rank's samples -- wait for it
            reqs = self._post_halo()
            E, off, hop = self.early_frames, self.sp.local_offset, self.sp.hop
            if E:
                self.backend.stage_mapped(self.buf[:, off:off + (E - 1) * hop + self.W], self.mapped[:E])
            self._finish_halo(reqs)
            if F > E:
                self.backend.stage_mapped(self.buf[:, off + E * hop:off + self.sp.local_samples], self.mapped[E:F])
        if F:
            self.backend.decay_scan(self.mapped, F, self.end_state)   # zero-carry scans -> end state; aggregates stay in the plan
        else:
            self.end_state.zero_()
        dist.all_gather_into_tensor(self.agg_all.view(-1), self.end_state.view(-1))      # A2
        carry = None
        if self.rank > 0:
            self.backend.fold_carry(self.agg_all, self.frames_per_rank, self.rank, self.carry)
            carry = self.carry
        if F:
            self.backend.decay_emit(self.mapped, F, carry, self.rgba)
        return self.rgba

    def time_collectives(self, iters: int = 20) -> float:
        """average wall time (ms) of one render's collectives (A1 halo, A2 decay carry) on their own"""
        import time
        import torch.distributed as dist
        torch = self.torch
        if not self.collective:
            return 0.0
        sync = torch.cuda.synchronize if self.buf.is_cuda else (lambda: None)
        for i in range(iters + 3):
            if i == 3:
                sync(); dist.barrier(); sync()
                t0 = time.perf_counter()
            self._exchange_halo()
            dist.all_gather_into_tensor(self.agg_all.view(-1), self.end_state.view(-1))
        sync(); dist.barrier(); sync()
        return (time.perf_counter() - t0) / iters * 1e3

    def time_stft_kernel(self, iters: int = 50, spin_ms: float = 60.0) -> float:
        """average launch duration (ms) of the dominant kernel -- K_A alone, sgz_stage_mapped_dominant -- HIP events on the launch stream
        around batches of `iters` launches, median of five batches, at the device's sustained clock"""
        import ctypes
        torch = self.torch
        from . import api
        hip = ctypes.CDLL("libamdhip64.so")
        x = self._view()
        F = self.local_frames
        mapped = torch.empty((F, self.plan.C, self.plan.sides, self.plan.P), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream().cuda_stream

        def call():                                                   # (rotating over the input copies like render())
            self._rotate()
            v = self._view()
            api.check(api.lib().sgz_stage_mapped_dominant(self.plan.h, v.data_ptr(), v.stride(0), v.shape[1], mapped.data_ptr(), stream))
        # `spin_ms` of untimed launches first: the device's clock settles only under sustained load (tools/clock_ramp_probe.py), and a
        # host wait after every launch would idle the GPU for ~20 us each time
        import time
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < spin_ms:
            for _ in range(32):
                call()
            torch.cuda.synchronize()
        # one event pair around `iters` launches back to back (an event between every two launches costs ~2 us of its own): the
        # average launch duration of the batch; the median of five batches
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        hip.hipEventCreate(ctypes.byref(e0)); hip.hipEventCreate(ctypes.byref(e1))
        samples = []
        for _ in range(5):
            hip.hipEventRecord(e0, ctypes.c_void_p(stream))
            for i in range(iters):
                call()
            hip.hipEventRecord(e1, ctypes.c_void_p(stream))
            hip.hipEventSynchronize(e1)
            ms = ctypes.c_float()
            hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1)
            samples.append(ms.value / iters)
        hip.hipEventDestroy(e0); hip.hipEventDestroy(e1)
        samples.sort()
        return samples[len(samples) // 2]
