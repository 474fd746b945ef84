"""Index-level emulation (numpy, CPU) of the 1024-thread, 16-values-per-thread channel transform (spectrum_real16.hip):
M = 16384 complex points = 16 (pass 1, in registers) x 16 (pass 2) x [4 across a lane quad x 16] (pass 3).  Every array below is
indexed the way the kernel indexes threads / registers / LDS, so the role formulas of the kernel can be checked against numpy's FFT
before anything runs on a GPU.  usage: python tools/emulate_real16.py"""
import numpy as np

M, N = 16384, 32768
rng = np.random.default_rng(1)
x = rng.standard_normal(N)
z = x[0::2] + 1j * x[1::2]
W = lambda n, e: np.exp(-2j * np.pi * (np.asarray(e) % n) / n)

def dft16(v, axis):
    return np.fft.fft(v, axis=axis)

# ---- pass 1: thread c, registers j: z[c + 1024 j] -> q1, times W_M^{c q1}
c = np.arange(1024)
a = z[c[:, None] + 1024 * np.arange(16)[None, :]]            # [c][j]
A = dft16(a, 1) * W(M, c[:, None] * np.arange(16)[None, :])  # [c][q1]
# ---- exchange 1: wave q1, lane = c_lo6, register c_hi4
B = np.empty((16, 64, 16), complex)                          # [q1][lane][c_hi4]
for q1 in range(16):
    for h in range(16):
        B[q1, :, h] = A[np.arange(64) + 64 * h, q1]
# ---- pass 2: DFT over c_hi4 -> q2, times W_1024^{c_lo6 q2}
C = dft16(B, 2) * W(1024, np.arange(64)[None, :, None] * np.arange(16)[None, None, :])   # [q1][c_lo6][q2]
# ---- exchange 2 (wave-local): lane = 4 q2 + l, register r: c_lo6 = 16 l + r
D = np.empty((16, 64, 16), complex)                          # [q1][lane][r]
for q2 in range(16):
    for l in range(4):
        D[:, 4 * q2 + l, :] = C[:, 16 * l:16 * l + 16, q2]
# ---- pass 3a: radix 4 across the quad with own + sigma * partner steps (v_fmac_f32_dpp), lane 3 rotated by i in between
lane = np.arange(64)
l = lane & 3
s1 = np.where(l < 2, 1.0, -1.0)[None, :, None]               # partner = lane ^ 2
s2 = np.array([1.0, -1.0, -1.0, 1.0])[l][None, :, None]      # partner = lane ^ 1
E = D + s1 * D[:, lane ^ 2, :]
E = np.where((l == 3)[None, :, None], 1j * E, E)
E = E + s2 * E[:, lane ^ 1, :]
# lane l now holds sign(l) a_m, m = brev2(l), signs (+, -, -, -)
brev2 = np.array([0, 2, 1, 3])
m = brev2[l]
sign = np.array([1.0, -1.0, -1.0, -1.0])[l]
r = np.arange(16)
tab3 = sign[:, None] * W(64, r[None, :] * m[:, None])        # [lane][r] (depends on lane & 3 only)
# ---- pass 3b: DFT over r -> s
Z = dft16(E * tab3[None, :, :], 2)                           # [q1][lane][s]
# k = q1 + 16 q2 + 256 m + 1024 s
ref = np.fft.fft(z)
q1 = np.arange(16)[:, None, None]; q2 = (lane >> 2)[None, :, None]; mm = m[None, :, None]; s = np.arange(16)[None, None, :]
k = q1 + 16 * q2 + 256 * mm + 1024 * s
err = np.abs(Z - ref[k]).max() / np.abs(ref).max()
print("transform: max rel err", err)
assert err < 1e-12

# ---- mirror partner: bins k and M - k
def partner(q1, ln):
    """(wave, lane) holding Z[M - k] in register 15 - s, for the thread (q1, ln) and its register s (generic columns)"""
    q2, ll = ln >> 2, ln & 3
    if q1 != 0:
        return (16 - q1) % 16, 63 - ln
    if q2 != 0:
        return 0, 4 * (16 - q2) + (3 - ll)
    return None
bad = 0
for w in range(16):
    for ln in range(64):
        p = partner(w, ln)
        if p is None:
            continue
        for s_ in range(16):
            kk = int(k[w, ln, s_])
            kp = int(k[p[0], p[1], 15 - s_])
            bad += (kk + kp) != M
print("mirror mismatches:", bad)
assert bad == 0
# column 0: quad 0 of wave 0 holds k = 256 (m + 4 s) = 256 q3; mirror 256 (64 - q3)
k0 = k[0, 0:4, :]
assert sorted(k0.ravel().tolist()) == [256 * i for i in range(64)]

# ---- recombination: 2 X[k] = (a + conj b) - i w (a - conj b), a = Z[k], b = Z[M - k], w = W_N^k
X = np.fft.rfft(x)
kk = np.arange(1, M)
aa, bb = ref[kk], ref[M - kk]
Xk = 0.5 * ((aa + np.conj(bb)) - 1j * W(N, kk) * (aa - np.conj(bb)))
print("recombination err", np.abs(Xk - X[kk]).max() / np.abs(X).max())
# W_N^k = W_N^{kb} W_32^s with kb = q1 + 16 q2 + 256 m
kb = (q1 + 16 * q2 + 256 * mm)
assert np.allclose(W(N, k), W(N, kb) * W(32, s))
print("ok")
