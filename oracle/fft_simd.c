/*
 * fft_simd.c -- a transform the compiler vectorises, for bench.py's cpu_baseline ONLY ("an optional -DSGZ_CPU_SIMD
 * path").  TEST INFRASTRUCTURE / timing only (see sgz_oracle.h): nothing compares the product against it; the parity oracle's
 * transform is primitives.c's radix-2.
 *
 * Why it exists: the reference's transform is cpl::dsp::UniFFT -> pffft (SSE / AVX; Source/Spectrum/TransformDSP.inl:487-502,
 * Make/Skeleton/licenses/pffft.txt), which is absent here.  The restatement's scalar radix-2 (bit reversal, one complex multiply per
 * butterfly through an array of structures) is several times slower than any SIMD library transform, so a GPU / CPU ratio quoted
 * against it alone is inflated by about the host's vector width.  This file is a plain radix-4 Stockham autosort transform on split
 * (re[], im[]) arrays whose inner loops are unit-stride and dependence-free: gcc -O3 -march=native turns them into AVX2 / AVX-512.
 * Same definition as sgzo_fft_forward (forward, unnormalised, natural order); tests/test_oracle_math.py holds it to that one and to
 * numpy's fp64 transform.
 */
#include "sgz_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct {
    uint32_t N;
    float *wr, *wi;            /* per radix-4 stage, per p: (w1, w2, w3) as six planes of n/4 floats, stages laid end to end */
    float *ar, *ai, *br, *bi;  /* ping-pong split buffers */
} simd_plan;

static _Thread_local simd_plan g_plan;       /* one cached size per thread (the baseline transforms one N) */

static void plan_free(simd_plan *pl)
{
    free(pl->wr); free(pl->wi); free(pl->ar); free(pl->ai); free(pl->br); free(pl->bi);
    memset(pl, 0, sizeof *pl);
}

static int plan_make(simd_plan *pl, uint32_t N)
{
    plan_free(pl);
    size_t tw = 0;
    for (uint32_t n = N; n >= 4; n >>= 2) tw += 3 * (size_t)(n / 4);
    pl->wr = (float *)aligned_alloc(64, ((tw + 16) * sizeof(float) + 63) / 64 * 64);
    pl->wi = (float *)aligned_alloc(64, ((tw + 16) * sizeof(float) + 63) / 64 * 64);
    const size_t bytes = ((size_t)N * sizeof(float) + 63) / 64 * 64;
    pl->ar = (float *)aligned_alloc(64, bytes); pl->ai = (float *)aligned_alloc(64, bytes);
    pl->br = (float *)aligned_alloc(64, bytes); pl->bi = (float *)aligned_alloc(64, bytes);
    if (!pl->wr || !pl->wi || !pl->ar || !pl->ai || !pl->br || !pl->bi) { plan_free(pl); return -1; }
    size_t o = 0;
    for (uint32_t n = N; n >= 4; n >>= 2) {
        const uint32_t n1 = n / 4;
        for (uint32_t k = 1; k <= 3; ++k)
            for (uint32_t p = 0; p < n1; ++p) {
                const double a = -2.0 * M_PI * (double)k * (double)p / (double)n;
                pl->wr[o + (size_t)(k - 1) * n1 + p] = (float)cos(a);
                pl->wi[o + (size_t)(k - 1) * n1 + p] = (float)sin(a);
            }
        o += 3 * (size_t)n1;
    }
    pl->N = N;
    return 0;
}

/* one radix-4 Stockham stage: length n sub-transforms at stride s (n * s == N).  x -> y.
 *   y[q + s(4p + k)] = w_p^k * sum_j x[q + s(p + j n/4)] * (-i)^(jk) */
static void stage4(uint32_t n, uint32_t s, const float *restrict xr, const float *restrict xi, float *restrict yr, float *restrict yi,
                   const float *restrict wr, const float *restrict wi)
{
    const uint32_t n1 = n / 4;
    if (s >= 8) {
        for (uint32_t p = 0; p < n1; ++p) {
            const float w1r = wr[p], w1i = wi[p], w2r = wr[n1 + p], w2i = wi[n1 + p], w3r = wr[2 * n1 + p], w3i = wi[2 * n1 + p];
            const float *ar = xr + (size_t)s * p, *ai = xi + (size_t)s * p;
            const float *br = ar + (size_t)s * n1, *bi = ai + (size_t)s * n1;
            const float *cr = br + (size_t)s * n1, *ci = bi + (size_t)s * n1;
            const float *dr = cr + (size_t)s * n1, *di = ci + (size_t)s * n1;
            float *y0r = yr + (size_t)s * 4 * p, *y0i = yi + (size_t)s * 4 * p;
            float *y1r = y0r + s, *y1i = y0i + s, *y2r = y1r + s, *y2i = y1i + s, *y3r = y2r + s, *y3i = y2i + s;
#pragma GCC ivdep
            for (uint32_t q = 0; q < s; ++q) {
                const float apcr = ar[q] + cr[q], apci = ai[q] + ci[q], amcr = ar[q] - cr[q], amci = ai[q] - ci[q];
                const float bpdr = br[q] + dr[q], bpdi = bi[q] + di[q];
                const float jr = -(bi[q] - di[q]), ji = br[q] - dr[q];                   /* i (b - d) */
                y0r[q] = apcr + bpdr; y0i[q] = apci + bpdi;
                const float t1r = amcr - jr, t1i = amci - ji, t2r = apcr - bpdr, t2i = apci - bpdi, t3r = amcr + jr, t3i = amci + ji;
                y1r[q] = t1r * w1r - t1i * w1i; y1i[q] = t1r * w1i + t1i * w1r;
                y2r[q] = t2r * w2r - t2i * w2i; y2i[q] = t2r * w2i + t2i * w2r;
                y3r[q] = t3r * w3r - t3i * w3i; y3i[q] = t3r * w3i + t3i * w3r;
            }
        }
    } else {                   /* the first stages (s = 1, 4): vectorise over p, the stores are strided */
        for (uint32_t q = 0; q < s; ++q) {
#pragma GCC ivdep
            for (uint32_t p = 0; p < n1; ++p) {
                const size_t ia = q + (size_t)s * p, ib = ia + (size_t)s * n1, ic = ib + (size_t)s * n1, id = ic + (size_t)s * n1;
                const float apcr = xr[ia] + xr[ic], apci = xi[ia] + xi[ic], amcr = xr[ia] - xr[ic], amci = xi[ia] - xi[ic];
                const float bpdr = xr[ib] + xr[id], bpdi = xi[ib] + xi[id];
                const float jr = -(xi[ib] - xi[id]), ji = xr[ib] - xr[id];
                const float t1r = amcr - jr, t1i = amci - ji, t2r = apcr - bpdr, t2i = apci - bpdi, t3r = amcr + jr, t3i = amci + ji;
                const size_t o = q + (size_t)s * 4 * p;
                yr[o] = apcr + bpdr; yi[o] = apci + bpdi;
                yr[o + s] = t1r * wr[p] - t1i * wi[p]; yi[o + s] = t1r * wi[p] + t1i * wr[p];
                yr[o + 2 * s] = t2r * wr[n1 + p] - t2i * wi[n1 + p]; yi[o + 2 * s] = t2r * wi[n1 + p] + t2i * wr[n1 + p];
                yr[o + 3 * s] = t3r * wr[2 * n1 + p] - t3i * wi[2 * n1 + p]; yi[o + 3 * s] = t3r * wi[2 * n1 + p] + t3i * wr[2 * n1 + p];
            }
        }
    }
}

/* forward, unnormalised, natural order; N a power of two >= 2.  Returns 0, or -1 when the plan cannot be allocated. */
int sgzo_fft_forward_simd(sgzo_cf *buf, uint32_t N)
{
    if (N < 2 || (N & (N - 1))) return -1;
    if (g_plan.N != N && plan_make(&g_plan, N)) return -1;
    float *xr = g_plan.ar, *xi = g_plan.ai, *yr = g_plan.br, *yi = g_plan.bi;
    for (uint32_t i = 0; i < N; ++i) { xr[i] = buf[i].re; xi[i] = buf[i].im; }
    size_t o = 0;
    uint32_t n = N, s = 1;
    for (; n >= 4; n >>= 2, s <<= 2) {
        stage4(n, s, xr, xi, yr, yi, g_plan.wr + o, g_plan.wi + o);
        o += 3 * (size_t)(n / 4);
        float *t = xr; xr = yr; yr = t;
        t = xi; xi = yi; yi = t;
    }
    if (n == 2) {              /* one radix-2 stage left (odd log2 N): stride s = N / 2, no twiddle */
        for (uint32_t q = 0; q < s; ++q) {
            const float ar = xr[q], ai = xi[q], br = xr[q + s], bi = xi[q + s];
            yr[q] = ar + br; yi[q] = ai + bi; yr[q + s] = ar - br; yi[q + s] = ai - bi;
        }
        xr = yr; xi = yi;
    }
    for (uint32_t i = 0; i < N; ++i) { buf[i].re = xr[i]; buf[i].im = xi[i]; }
    return 0;
}
