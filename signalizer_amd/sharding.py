"""Time-chunk sharding of long spectrogram renders across the GPUs of one node.

The global stream is the concatenation of every rank's chunk (rank r holds samples [r*S, (r+1)*S)).
A frame belongs to the rank that owns its first sample.  Per render:
  A1  all-gather of every rank's leading W samples (the "overlap frames" halo); rank r appends rank r+1's
  K_A window x FFT x map over the local frames (no collective)
  A2  all-gather of every rank's zero-carry decay end state; exact carry fold (sgz_decay_fold_carry)
  K_B decay + dB + colour with the folded carry as state
With world == 1 this degenerates to a single sgz_spectrogram_render_device call.
One process per GPU; torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" for the CPU plan tests).
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

    def stage_decay_colour(self, mapped, frames, rgba, state):
        from . import api
        api.check(api.lib().sgz_stage_decay_colour(self.plan.h, mapped.data_ptr(), frames,
                                                   rgba.data_ptr() if rgba is not None else None, None,
                                                   state.data_ptr(), self._stream()))

    def fold_carry(self, aggs, frames_per_rank, rank, carry):
        self.plan.fold_carry(aggs, frames_per_rank, rank, carry)


class TimeChunkRenderer:
    def __init__(self, plan, chunk_audio, rank: int = 0, world: int = 1, backend=None):
        import torch
        self.torch = torch
        self.plan = plan
        self.backend = backend if backend is not None else GpuBackend(plan)
        self.rank, self.world = rank, world
        self.nch, S = chunk_audio.shape
        W, hop = plan.cfg.window_size, plan.cfg.hop
        assert S >= W, "chunk must hold at least one window"
        self.sp = ShardPlan(rank, world, S, W, hop)
        dev = chunk_audio.device
        # local buffer = own chunk followed by the next rank's leading W samples
        self.buf = torch.zeros((self.nch, S + W), dtype=torch.float32, device=dev)
        self.buf[:, :S] = chunk_audio
        self.S, self.W = S, W
        self.local_frames = self.sp.local_frames
        P, C = plan.P, plan.C
        self.rgba = torch.empty((max(self.local_frames, 1), P, 4), dtype=torch.uint8, device=dev)
        self.state = torch.zeros((C, 2, P, 2), dtype=torch.float32, device=dev)
        if world > 1:
            self.mapped = torch.empty((max(self.local_frames, 1), C, plan.sides, P), dtype=torch.float32, device=dev)
            self.halo_send = torch.empty((self.nch, W), dtype=torch.float32, device=dev)
            self.halo_all = torch.empty((world, self.nch, W), dtype=torch.float32, device=dev)
            self.agg_all = torch.empty((world, C, 2, P, 2), dtype=torch.float32, device=dev)
            self.carry = torch.zeros((C, 2, P, 2), dtype=torch.float32, device=dev)
            self.frames_per_rank = [self.sp.frames_of(r) for r in range(world)]

    def _view(self):
        off = self.sp.local_offset
        return self.buf[:, off:off + self.sp.local_samples]

    def render(self):
        """one full pass; returns this rank's RGBA8 columns [local_frames, P, 4]"""
        torch = self.torch
        if self.world == 1:
            # single device: start from a zero decay state, nobody needs the end state -> no memset, no snapshot
            self.backend.render(self._view(), self.rgba, None)
            return self.rgba
        import torch.distributed as dist
        # A1: halo = everybody's leading W samples
        self.halo_send.copy_(self.buf[:, :self.W])
        dist.all_gather_into_tensor(self.halo_all.view(-1), self.halo_send.view(-1))
        if self.rank + 1 < self.world:
            self.buf[:, self.S:] = self.halo_all[self.rank + 1]
        x = self._view()
        # K_A once; K_B as a state-only pass (zero carry -> this chunk's end state), then the full pass with the folded carry
        self.backend.stage_mapped(x, self.mapped)
        self.state.zero_()
        self.backend.stage_decay_colour(self.mapped, self.local_frames, None if self.rank > 0 else self.rgba, self.state)
        # A2: decay carry
        dist.all_gather_into_tensor(self.agg_all.view(-1), self.state.view(-1))
        if self.rank > 0:
            self.backend.fold_carry(self.agg_all, self.frames_per_rank, self.rank, self.carry)
            self.backend.stage_decay_colour(self.mapped, self.local_frames, self.rgba, self.carry)
        return self.rgba

    def time_collectives(self, iters: int = 20) -> float:
        """average wall time (ms) of one render's two all-gathers (A1 halo, A2 decay carry) on their own"""
        import time
        import torch.distributed as dist
        torch = self.torch
        if self.world == 1:
            return 0.0
        sync = torch.cuda.synchronize if self.buf.is_cuda else (lambda: None)
        for i in range(iters + 3):
            if i == 3:
                sync(); dist.barrier(); sync()
                t0 = time.perf_counter()
            dist.all_gather_into_tensor(self.halo_all.view(-1), self.halo_send.view(-1))
            dist.all_gather_into_tensor(self.agg_all.view(-1), self.state.view(-1))
        sync(); dist.barrier(); sync()
        return (time.perf_counter() - t0) / iters * 1e3

    def time_stft_kernel(self, iters: int = 50) -> float:
        """average duration (ms) of the dominant kernel's launches, HIP events on the launch stream"""
        import ctypes
        torch = self.torch
        from . import api
        hip = ctypes.CDLL("libamdhip64.so")
        x = self._view()
        F = self.local_frames
        mapped = torch.empty((F, self.plan.C, self.plan.sides, self.plan.P), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream().cuda_stream
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        hip.hipEventCreate(ctypes.byref(e0)); hip.hipEventCreate(ctypes.byref(e1))
        call = lambda: api.check(api.lib().sgz_stage_mapped(self.plan.h, x.data_ptr(), x.stride(0), x.shape[1],
                                                           mapped.data_ptr(), stream))
        for _ in range(5):
            call()
        total = 0.0
        for _ in range(iters):
            hip.hipEventRecord(e0, ctypes.c_void_p(stream))
            call()
            hip.hipEventRecord(e1, ctypes.c_void_p(stream))
            hip.hipEventSynchronize(e1)
            ms = ctypes.c_float()
            hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1)
            total += ms.value
        return total / iters
