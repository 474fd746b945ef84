/*
 * primitives.c -- restatements of the cpl primitives the Signalizer hot path calls.
 * TEST INFRASTRUCTURE (see sgz_oracle.h).  cpl is absent from /root/reference, so every function
 * here follows the *published definition* of the algorithm and the semantics the Signalizer call
 * sites rely on; each is tagged UNVERIFIED vs cpl.
 */
#include "sgz_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* cpl::Math::nextPow2Inc + TransformConstant::setStorage, Source/Spectrum/TransformConstant.h:84:
 * transformSize = max(32, nextPow2Inc(windowSize))  (smallest power of two >= W). */
uint32_t sgzo_transform_size(uint32_t W)
{
    uint32_t n = 1;
    while (n < W) n <<= 1;
    return n < 32 ? 32 : n;
}

/* ---- window design -------------------------------------------------------------------------
 * Call site: windowDesigner.generateWindow<T>(windowKernel, windowSize) -> windowKernelScale,
 * Source/Spectrum/TransformConstant.h:104-107.  UNVERIFIED vs cpl.  Shapes follow the Octave
 * `signal` package definitions (cpl bundles Octave-derived window code per
 * Make/Skeleton/licenses); periodic = the W+1 point symmetric window with the last point dropped.
 * The returned scale is defined by the property the call site relies on (TransformDSP.inl:537-540):
 * a unit sine on an exact bin must read 1.0 after `invSize = scale / (W/2)`, i.e. scale = W / sum(w).
 */
static double bessel_i0(double x)
{
    double sum = 1.0, term = 1.0;
    const double q = x * x * 0.25;
    for (int k = 1; k < 200; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < sum * 1e-17) break;
    }
    return sum;
}

double sgzo_window(uint32_t type, uint32_t symmetry, double alpha, double beta, uint32_t W, float *out)
{
    if (W == 0) return 1.0;
    /* denominator of the phase ramp: W-1 for symmetric, W for periodic */
    const double D = (symmetry == SGZO_WIN_PERIODIC) ? (double)W : (W > 1 ? (double)(W - 1) : 1.0);
    double sum = 0.0;
    for (uint32_t n = 0; n < W; ++n) {
        const double x = (double)n / D;          /* 0..1 */
        const double t = 2.0 * M_PI * x;
        double w;
        switch (type) {
        default:
        case SGZO_WIN_RECT: w = 1.0; break;
        case SGZO_WIN_HANN: w = 0.5 - 0.5 * cos(t); break;
        case SGZO_WIN_HAMMING: w = 0.54 - 0.46 * cos(t); break;
        case SGZO_WIN_FLATTOP:
            w = 0.21557895 - 0.41663158 * cos(t) + 0.277263158 * cos(2 * t)
              - 0.083578947 * cos(3 * t) + 0.006947368 * cos(4 * t);
            break;
        case SGZO_WIN_BLACKMAN: w = 0.42 - 0.5 * cos(t) + 0.08 * cos(2 * t); break;
        case SGZO_WIN_EXACT_BLACKMAN:
            w = 7938.0 / 18608.0 - 9240.0 / 18608.0 * cos(t) + 1430.0 / 18608.0 * cos(2 * t);
            break;
        case SGZO_WIN_NUTTALL:
            w = 0.355768 - 0.487396 * cos(t) + 0.144232 * cos(2 * t) - 0.012604 * cos(3 * t);
            break;
        case SGZO_WIN_BLACKMAN_NUTTALL:
            w = 0.3635819 - 0.4891775 * cos(t) + 0.1365995 * cos(2 * t) - 0.0106411 * cos(3 * t);
            break;
        case SGZO_WIN_BLACKMAN_HARRIS:
            w = 0.35875 - 0.48829 * cos(t) + 0.14128 * cos(2 * t) - 0.01168 * cos(3 * t);
            break;
        case SGZO_WIN_TRIANGULAR: w = 1.0 - fabs(2.0 * x - 1.0); break;
        case SGZO_WIN_WELCH: { const double u = 2.0 * x - 1.0; w = 1.0 - u * u; break; }
        case SGZO_WIN_GAUSSIAN: { /* alpha = sigma as a fraction of the half width; default 0.4 */
            const double s = alpha > 0 ? alpha : 0.4;
            const double u = (2.0 * x - 1.0) / s;
            w = exp(-0.5 * u * u);
            break;
        }
        case SGZO_WIN_KAISER: { /* beta = pi*alpha shape parameter */
            const double u = 2.0 * x - 1.0;
            const double r = 1.0 - u * u;
            w = bessel_i0(beta * sqrt(r > 0 ? r : 0)) / bessel_i0(beta);
            break;
        }
        }
        out[n] = (float)w;
        sum += (double)out[n];
    }
    return sum > 0 ? (double)W / sum : 1.0;
}

/* ---- FFT --------------------------------------------------------------------------------------
 * Call site: constant.fft.forward(in, out, work), Source/Spectrum/TransformDSP.inl:494-498 ->
 * cpl::dsp::UniFFT<std::complex<float>> (pffft behind it).  Semantics relied upon: forward,
 * unnormalised (a unit sine on an exact bin yields N/2, :538-540), natural order (bins k and N-k
 * are indexed directly, :684,:793).  UNVERIFIED vs cpl: pffft's butterfly order (and so its exact
 * fp32 rounding) is not reproduced; this is a textbook in-place radix-2 DIT with a fp64-derived
 * twiddle table rounded to fp32.  fp32 parity with any other correct FFT is therefore a tolerance
 * (see tests: <= 3e-6 * max|X| absolute per bin at N = 32768), never bit-exactness.
 */
static uint32_t ilog2(uint32_t n) { uint32_t l = 0; while ((1u << l) < n) ++l; return l; }

void sgzo_fft_forward(sgzo_cf *buf, uint32_t N)
{
    if (N < 2) return;
    const uint32_t lg = ilog2(N);
    for (uint32_t i = 0; i < N; ++i) {
        uint32_t r = 0;
        for (uint32_t b = 0; b < lg; ++b) r |= ((i >> b) & 1u) << (lg - 1 - b);
        if (r > i) { sgzo_cf t = buf[i]; buf[i] = buf[r]; buf[r] = t; }
    }
    sgzo_cf *tw = (sgzo_cf *)malloc(sizeof(sgzo_cf) * (N / 2));
    for (uint32_t k = 0; k < N / 2; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)N;
        tw[k].re = (float)cos(a);
        tw[k].im = (float)sin(a);
    }
    for (uint32_t len = 2; len <= N; len <<= 1) {
        const uint32_t half = len >> 1, step = N / len;
        for (uint32_t base = 0; base < N; base += len) {
            for (uint32_t j = 0; j < half; ++j) {
                const sgzo_cf w = tw[j * step];
                sgzo_cf *a = &buf[base + j], *b = &buf[base + j + half];
                const float tr = b->re * w.re - b->im * w.im;
                const float ti = b->re * w.im + b->im * w.re;
                b->re = a->re - tr; b->im = a->im - ti;
                a->re = a->re + tr; a->im = a->im + ti;
            }
        }
    }
    free(tw);
}

void sgzo_fft_forward_f64(sgzo_cd *buf, uint32_t N)
{
    if (N < 2) return;
    const uint32_t lg = ilog2(N);
    for (uint32_t i = 0; i < N; ++i) {
        uint32_t r = 0;
        for (uint32_t b = 0; b < lg; ++b) r |= ((i >> b) & 1u) << (lg - 1 - b);
        if (r > i) { sgzo_cd t = buf[i]; buf[i] = buf[r]; buf[r] = t; }
    }
    for (uint32_t len = 2; len <= N; len <<= 1) {
        const uint32_t half = len >> 1;
        for (uint32_t base = 0; base < N; base += len) {
            for (uint32_t j = 0; j < half; ++j) {
                const double ang = -2.0 * M_PI * (double)j / (double)len;
                const double wr = cos(ang), wi = sin(ang);
                sgzo_cd *a = &buf[base + j], *b = &buf[base + j + half];
                const double tr = b->re * wr - b->im * wi;
                const double ti = b->re * wi + b->im * wr;
                b->re = a->re - tr; b->im = a->im - ti;
                a->re = a->re + tr; a->im = a->im + ti;
            }
        }
    }
}

/* ---- two-for-one split ------------------------------------------------------------------------
 * Call site: dsp::separateTransformsIPL(csf), Source/Spectrum/TransformDSP.inl:646,:858.
 * Semantics relied upon (:649-652,:861-864,:892-896): afterwards csf[k] is the first channel's
 * bin k and csf[N-k] the second channel's bin k for 1 <= k < N/2; csf[0] still packs
 * (DC_first, DC_second) as (re, im); bin N/2 is left packed.  UNVERIFIED vs cpl: the 1/2 that
 * yields true per-channel spectra is assumed (a Separate-mode sine must read the same level as
 * the Left-mode transform of the same channel), as is storing the second channel unconjugated.
 *   X1[k] = (Z[k] + conj(Z[N-k])) / 2 ;  X2[k] = (Z[k] - conj(Z[N-k])) / (2i)
 */
void sgzo_separate_transforms_ipl(sgzo_cf *csf, uint32_t N)
{
    for (uint32_t k = 1; k < N / 2; ++k) {
        const sgzo_cf a = csf[k], b = csf[N - k];
        sgzo_cf x1, x2;
        x1.re = (a.re + b.re) * 0.5f;
        x1.im = (a.im - b.im) * 0.5f;
        x2.re = (a.im + b.im) * 0.5f;
        x2.im = (b.re - a.re) * 0.5f;
        csf[k] = x1;
        csf[N - k] = x2;
    }
}

/* ---- Lanczos / linear bin filters -------------------------------------------------------------
 * Call sites: dsp::lanczosFilter<std::complex<T>, true>(csf, x, 5) TransformDSP.inl:599,:742,:911;
 * dsp::linearFilter<std::complex<T>>(csf, x) :588,:683,:892;
 * cpl::dsp::lanczosFilter<double>(kernel, 21, 10 + delta, 10) OscilloscopeRendering.cpp:874.
 * UNVERIFIED vs cpl.  Definition used (Lanczos resampling, Duchon 1979):
 *   y(x) = sum_{i = floor(x)-a+1}^{floor(x)+a} v[i] * L(x - i),  L(d) = sinc(d) * sinc(d/a), L(0)=1
 * kernel evaluated in fp64 and rounded to the data type; accumulation in the data type, ascending i.
 * Template argument `true` is taken as periodic indexing over the passed array (size N+1 for csf,
 * which is what makes csf[N] - the second channel's DC - the left neighbour of csf[0]); the
 * non-wrapping variant skips taps outside [0,size).
 */
double sgzo_lanczos_kernel(double d, int a)
{
    if (d == 0.0) return 1.0;
    if (d <= -(double)a || d >= (double)a) return 0.0;
    const double pd = M_PI * d;
    return (double)a * sin(pd) * sin(pd / (double)a) / (pd * pd);
}

sgzo_cf sgzo_lanczos_filter_wrap(const sgzo_cf *v, size_t size, float x, int a)
{
    sgzo_cf acc = {0.0f, 0.0f};
    const double xd = (double)x;
    const long fl = (long)floor(xd);
    for (long i = fl - a + 1; i <= fl + a; ++i) {
        const float w = (float)sgzo_lanczos_kernel(xd - (double)i, a);
        long idx = i % (long)size;
        if (idx < 0) idx += (long)size;
        acc.re = acc.re + v[idx].re * w;
        acc.im = acc.im + v[idx].im * w;
    }
    return acc;
}

sgzo_cf sgzo_linear_filter(const sgzo_cf *v, size_t size, float x)
{
    const double xd = (double)x;
    const long fl = (long)floor(xd);
    const float frac = (float)(xd - (double)fl);
    long i0 = fl % (long)size; if (i0 < 0) i0 += (long)size;
    long i1 = (fl + 1) % (long)size; if (i1 < 0) i1 += (long)size;
    sgzo_cf r;
    r.re = v[i0].re * (1.0f - frac) + v[i1].re * frac;
    r.im = v[i0].im * (1.0f - frac) + v[i1].im * frac;
    return r;
}

double sgzo_lanczos_filter_f64(const float *v, size_t size, double x, int a)
{
    double acc = 0.0;
    const long fl = (long)floor(x);
    for (long i = fl - a + 1; i <= fl + a; ++i) {
        if (i < 0 || i >= (long)size) continue;
        acc += (double)v[i] * sgzo_lanczos_kernel(x - (double)i, a);
    }
    return acc;
}
