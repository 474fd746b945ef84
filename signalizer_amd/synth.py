"""Deterministic synthetic audio for parity tests and bench.py.

channel c = 0.5*sin(2*pi*f_c*t + phi_c) + 0.25*logsweep(20 Hz -> 0.45*sr over T) + 0.1*U(-1,1),
f_c = 997*(1 + c/7) Hz, phi_c = c*pi/5, noise from numpy Generator(PCG64(seed + c)); fp32 planar.
"""
from __future__ import annotations

import numpy as np


def gen(seed: int, sample_rate: float, nsamples: int, channels: int) -> np.ndarray:
    t = np.arange(nsamples, dtype=np.float64) / float(sample_rate)
    T = nsamples / float(sample_rate)
    f0, f1 = 20.0, 0.45 * sample_rate
    k = np.log(f1 / f0)
    # phase of an exponential sweep: 2*pi*f0*T/k * (exp(k*t/T) - 1)
    sweep = np.sin(2.0 * np.pi * f0 * T / k * np.expm1(k * t / T))
    out = np.empty((channels, nsamples), np.float32)
    for c in range(channels):
        rng = np.random.Generator(np.random.PCG64(seed + c))
        fc = 997.0 * (1.0 + c / 7.0)
        x = 0.5 * np.sin(2.0 * np.pi * fc * t + c * np.pi / 5.0) + 0.25 * sweep
        x += 0.1 * rng.uniform(-1.0, 1.0, nsamples)
        out[c] = x.astype(np.float32)
    return out
