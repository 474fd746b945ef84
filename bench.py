#!/usr/bin/env python3
"""bench.py -- 32768-pt stereo STFT frames/sec (75 % overlap) on MI355X, BASELINE.json metric.

A "step" = one pass of the whole Spectrum hot path (window x audio -> FFT -> split -> |X| -> log-frequency
pixel mapping -> peak decay -> dB -> colour map -> RGBA8 columns) over BASELINE.json configs[1]:
stereo 48 kHz, 60 s, N = W = 32768, hop 8192 => 348 frames, P = 1024.  Audio is resident in HBM when the
timed region starts.  N GPUs: time-chunk sharding -- every rank renders its own 60 s chunk of a
60*N s stream (weak scaling), halo frames and the decay carry exchanged with RCCL all-gathers.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def _hip():
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    raise RuntimeError("libamdhip64.so not found")


class HipEvents:
    """HIP events on an explicit stream (torch.cuda.Event only sees torch's current stream)."""

    def __init__(self, n: int):
        self.h = _hip()
        self.ev = [ctypes.c_void_p() for _ in range(n)]
        for e in self.ev:
            assert self.h.hipEventCreate(ctypes.byref(e)) == 0

    def record(self, i: int, stream: int):
        assert self.h.hipEventRecord(self.ev[i], ctypes.c_void_p(stream)) == 0

    def elapsed_ms(self, a: int, b: int) -> float:
        ms = ctypes.c_float()
        assert self.h.hipEventSynchronize(self.ev[b]) == 0
        assert self.h.hipEventElapsedTime(ctypes.byref(ms), self.ev[a], self.ev[b]) == 0
        return float(ms.value)


def measured_traffic():
    """HBM bytes per K_A launch from the last committed PMC pass (tools/profile.sh -> profiles/traffic_latest.json;
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as fh:
            return int(json.load(fh)["traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(cfg: dict, x: np.ndarray, budget_s: float = 12.0) -> dict:
    """The CPU oracle (a restatement: 'port') timed on this host, 1 thread, on a bounded sample of the
    same workload: the first frames of the same buffer until ~budget_s of CPU work."""
    from oracle import pyoracle as po
    p = po.params_from_dict(cfg)
    t0 = time.perf_counter()
    po.spectrogram_range(p, x, 0, 4)
    per = (time.perf_counter() - t0) / 4
    nfr = int(max(8, min(348, budget_s / max(per, 1e-6))))
    t0 = time.perf_counter()
    po.spectrogram_range(p, x, 0, nfr)
    dt = time.perf_counter() - t0
    return {"value": nfr / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"first {nfr} of 348 frames of the same 60 s stereo buffer, oracle/libsgz_oracle.so (gcc -O3, strict fp), 1 thread of {os.cpu_count()}"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=("cfg2", "cfg5"), default="cfg2",
                    help="cfg2 (default): BASELINE.json's metric; cfg5: the 64-channel 65536-pt job of BASELINE configs[4], for the "
                         "1/2/4/8-GPU time-chunk scaling curve of SURVEY 8(e) (20 s of 32 pairs per rank)")
    args = ap.parse_args()

    import torch
    from signalizer_amd import api, config, synth
    from signalizer_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    if args.workload == "cfg5":
        cfg = config.cfg5()
        sr = 96000
        S = 20 * sr                                   # per-rank chunk: 20 s of 64 channels (491 MB)
    else:
        cfg = config.cfg2()
        sr = 48000
        S = int(config.CFG2_SECONDS * sr)             # per-rank chunk: 2 880 000 samples
    hop, W = cfg["hop"], cfg["window_size"]
    pairs = cfg["num_pairs"]
    bytes_per_frame = 2 * W * 4 + 4 * cfg["axis_points"]      # algorithmic bytes per stereo frame
    # rank r owns samples [r*S, (r+1)*S) of a world-times-longer stream (weak scaling)
    x_host = synth.gen(config.CFG2_SEED + 100 * rank, sr, S, 2 * pairs)
    plan = api.Plan(cfg).upload()
    shard = sharding.TimeChunkRenderer(plan, torch.from_numpy(x_host).to(dev), rank=rank, world=world)
    frames_per_rank = shard.local_frames
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        shard.render()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = HipEvents(2)
    t0 = time.perf_counter()
    ev.record(0, stream)
    for _ in range(args.steps):
        step()
    ev.record(1, stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu_ms = ev.elapsed_ms(0, 1)

    # dominant kernel (stftMapKernel): its own launches timed with HIP events on the launch stream
    kern_ms = shard.time_stft_kernel(iters=50)
    # outside the timed region: latency of one render from an idle GPU, and the two all-gathers on their own
    shots = []
    for _ in range(20):
        torch.cuda.synchronize()
        ts = time.perf_counter()
        step()
        torch.cuda.synchronize()
        shots.append((time.perf_counter() - ts) * 1e3)
    single_shot_ms = float(np.median(shots))
    coll_ms = shard.time_collectives(iters=20) if world > 1 else 0.0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        fr = torch.tensor([frames_per_rank], dtype=torch.float64, device=dev)
        dist.all_reduce(fr, op=dist.ReduceOp.SUM)
        total_frames = int(fr.item())
    else:
        total_frames = frames_per_rank

    if rank == 0:
        value = total_frames * pairs * args.steps / dt
        achieved = frames_per_rank * pairs * bytes_per_frame / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "32768-pt stereo STFT frames/sec (75% overlap); achieved HBM GB/s vs peak" if args.workload == "cfg2" else
                      "65536-pt stereo-pair STFT frames/sec, 32 pairs (75% overlap), time-chunk sharded; achieved HBM GB/s vs peak",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[1]: stereo 48 kHz spectrogram, 32768-pt FFT, 75% overlap "
                                    "(hop 8192), 60 s buffer => 348 frames/GPU, P=1024, Hann, Separate, Lanczos, log view")
                       if args.workload == "cfg2" else
                       ("BASELINE.json configs[4]: 64-channel 96 kHz spectrogram, 65536-pt FFT, 75% overlap (hop 16384), 20 s per GPU "
                        "=> 114 frames x 32 pairs per GPU, P=1024, Hann, Separate, Lanczos, log view"),
                       "frames_per_gpu": frames_per_rank, "parallelism": f"time-chunk x{world}",
                       "gpu_ms_per_step_rank0": gpu_ms / args.steps, "single_shot_ms": single_shot_ms,
                       "collectives_ms_per_step": coll_ms},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": measured_traffic(),
                         "kernel": "stftMapKernel<5, 0, true>" if args.workload == "cfg2" else
                                   "stftHalfKernel<5, 0, true> + mapSideKernel<1024> (all slabs of one K_A pass)",
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": frames_per_rank * pairs * bytes_per_frame},
        }
        if args.workload != "cfg2":
            out["roofline"]["traffic"] = None          # the committed PMC numbers are cfg2's
        if not args.no_cpu_baseline and world == 1 and args.workload == "cfg2":
            out["cpu_baseline"] = cpu_baseline(cfg, x_host)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
