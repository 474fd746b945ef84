// plan.cpp -- host construction of the Spectrum constant block (no GPU needed).
// Mirrors, in the product's own code, what Spectrum::handleFlagUpdates rebuilds
// (Source/Spectrum/Spectrum.cpp:351-616): setStorage (TransformConstant.h:81-92), remapFrequencies
// (:125-180), generateSlopeMap (:109-118), regenerateWindowKernel (:104-107), colour ratios
// (Spectrum.cpp:226-246), colour rotation (TransformConstant.h:55-65), and it pre-resolves the
// pixel->bin mapping decisions of mapToLinearSpace (TransformDSP.inl:565-639, :871-985, :995-1096)
// into per-pixel records so the device does no fp64 index math ("hard parts").
#include "plan.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace sgz {

uint32_t transformSizeFor(uint32_t W)
{
    uint32_t n = 1;
    while (n < W && n < (1u << 31)) n <<= 1;
    return n < 32 ? 32u : n;           // max(32, nextPow2Inc(W)), TransformConstant.h:84
}

// ---- window design (stand-in for cpl's windowDesigner.generateWindow; scale = W / sum(w)) ----------
static double besselI0(double x)
{
    double sum = 1.0, term = 1.0;
    const double q = x * x * 0.25;
    for (int k = 1; k < 200; ++k) {
        term *= q / (double(k) * double(k));
        sum += term;
        if (term < sum * 1e-17) break;
    }
    return sum;
}

double designWindow(uint32_t type, uint32_t symmetry, double alpha, double beta, uint32_t W, float *out)
{
    if (W == 0) return 1.0;
    const double kPi = 3.14159265358979323846;
    const double D = (symmetry == SGZ_WIN_PERIODIC) ? double(W) : (W > 1 ? double(W - 1) : 1.0);
    static const double cosCoeffs[][5] = {
        /* HANN */              {0.5, 0.5, 0, 0, 0},
        /* HAMMING */           {0.54, 0.46, 0, 0, 0},
        /* FLATTOP */           {0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368},
        /* BLACKMAN */          {0.42, 0.5, 0.08, 0, 0},
        /* EXACT_BLACKMAN */    {7938.0 / 18608.0, 9240.0 / 18608.0, 1430.0 / 18608.0, 0, 0},
        /* NUTTALL */           {0.355768, 0.487396, 0.144232, 0.012604, 0},
        /* BLACKMAN_NUTTALL */  {0.3635819, 0.4891775, 0.1365995, 0.0106411, 0},
        /* BLACKMAN_HARRIS */   {0.35875, 0.48829, 0.14128, 0.01168, 0},
    };
    double sum = 0.0;
    for (uint32_t n = 0; n < W; ++n) {
        const double x = double(n) / D;
        const double t = 2.0 * kPi * x;
        double w = 1.0;
        if (type >= SGZ_WIN_HANN && type <= SGZ_WIN_BLACKMAN_HARRIS) {
            const double *c = cosCoeffs[type - SGZ_WIN_HANN];
            // a0 - a1 cos t + a2 cos 2t - a3 cos 3t + a4 cos 4t, evaluated left to right
            switch (type) {
            case SGZ_WIN_HANN: case SGZ_WIN_HAMMING: w = c[0] - c[1] * std::cos(t); break;
            case SGZ_WIN_BLACKMAN: case SGZ_WIN_EXACT_BLACKMAN:
                w = c[0] - c[1] * std::cos(t) + c[2] * std::cos(2 * t); break;
            case SGZ_WIN_FLATTOP:
                w = c[0] - c[1] * std::cos(t) + c[2] * std::cos(2 * t) - c[3] * std::cos(3 * t) + c[4] * std::cos(4 * t); break;
            default:
                w = c[0] - c[1] * std::cos(t) + c[2] * std::cos(2 * t) - c[3] * std::cos(3 * t); break;
            }
        } else if (type == SGZ_WIN_TRIANGULAR) {
            w = 1.0 - std::fabs(2.0 * x - 1.0);
        } else if (type == SGZ_WIN_WELCH) {
            const double u = 2.0 * x - 1.0; w = 1.0 - u * u;
        } else if (type == SGZ_WIN_GAUSSIAN) {
            const double s = alpha > 0 ? alpha : 0.4;
            const double u = (2.0 * x - 1.0) / s;
            w = std::exp(-0.5 * u * u);
        } else if (type == SGZ_WIN_KAISER) {
            const double u = 2.0 * x - 1.0;
            const double r = 1.0 - u * u;
            w = besselI0(beta * std::sqrt(r > 0 ? r : 0)) / besselI0(beta);
        }
        out[n] = float(w);
        sum += double(out[n]);
    }
    return sum > 0 ? double(W) / sum : 1.0;
}

double lanczosKernel(double d, int a)
{
    const double kPi = 3.14159265358979323846;
    if (d == 0.0) return 1.0;
    if (d <= -double(a) || d >= double(a)) return 0.0;
    const double pd = kPi * d;
    return double(a) * std::sin(pd) * std::sin(pd / double(a)) / (pd * pd);
}

// juce::Colour::withRotatedHue through ColourHelpers::HSB (juce_Colour.cpp:33-107, :331-336); SURVEY Q12.
static inline int roundToInt(float v) { return int(std::lrint(double(v))); }
void rotateHueRgb8(const uint8_t rgb[3], float amount, uint8_t out[3])
{
    const int r = rgb[0], g = rgb[1], b = rgb[2];
    const int hi = std::max(r, std::max(g, b)), lo = std::min(r, std::min(g, b));
    float hue = 0, sat = 0;
    if (hi != 0) {
        sat = float(hi - lo) / float(hi);
        if (sat > 0) {
            const float invDiff = 1.0f / float(hi - lo);
            const float red = float(hi - r) * invDiff, green = float(hi - g) * invDiff, blue = float(hi - b) * invDiff;
            if (r == hi) hue = blue - green;
            else if (g == hi) hue = 2.0f + red - blue;
            else hue = 4.0f + green - red;
            hue *= 1.0f / 6.0f;
            if (hue < 0) ++hue;
        }
    }
    const float brightness = float(hi) / 255.0f;
    float h = hue + amount, s = sat, v = brightness * 255.0f;
    v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
    const uint8_t intV = uint8_t(roundToInt(v));
    if (s <= 0) { out[0] = out[1] = out[2] = intV; return; }
    s = std::min(1.0f, s);
    h = (h - std::floor(h)) * 6.0f + 0.00001f;
    const float f = h - std::floor(h);
    const uint8_t x = uint8_t(roundToInt(v * (1.0f - s)));
    const uint8_t up = uint8_t(roundToInt(v * (1.0f - (s * (1.0f - f)))));
    const uint8_t dn = uint8_t(roundToInt(v * (1.0f - s * f)));
    if (h < 1.0f)      { out[0] = intV; out[1] = up;   out[2] = x; }
    else if (h < 2.0f) { out[0] = dn;   out[1] = intV; out[2] = x; }
    else if (h < 3.0f) { out[0] = x;    out[1] = intV; out[2] = up; }
    else if (h < 4.0f) { out[0] = x;    out[1] = dn;   out[2] = intV; }
    else if (h < 5.0f) { out[0] = up;   out[1] = x;    out[2] = intV; }
    else               { out[0] = intV; out[1] = x;    out[2] = dn; }
}

// ---- pixel records ------------------------------------------------------------------------------------
namespace {

struct RecBuilder {
    Plan &p;
    long N;
    size_t csfSize;
    explicit RecBuilder(Plan &pl) : p(pl), N(long(pl.N)), csfSize(size_t(pl.N) + 1) {}

    int32_t wrap(long i) const { long m = i % long(csfSize); if (m < 0) m += long(csfSize); return int32_t(m); }

    PixelRec interp(float pos, size_t clampHi) {
        PixelRec r{};
        r.kind = 0;
        r.c = int32_t(p.weights.size());
        const double xd = double(pos);
        switch (p.cfg.bin_interp) {
        case SGZ_INTERP_LINEAR: {
            const long fl = long(std::floor(xd));
            const float frac = float(xd - double(fl));
            r.a = wrap(fl); r.b = 2;
            p.weights.push_back(1.0f - frac);
            p.weights.push_back(frac);
            break;
        }
        case SGZ_INTERP_LANCZOS: {
            const int a = 5;
            const long fl = long(std::floor(xd));
            r.a = wrap(fl - a + 1); r.b = 2 * a;
            for (long i = fl - a + 1; i <= fl + a; ++i) p.weights.push_back(float(lanczosKernel(xd - double(i), a)));
            break;
        }
        default: {   // None: +0.5 to centre bins, TransformDSP.inl:577
            size_t idx = size_t(xd + 0.5);
            idx = idx > clampHi ? clampHi : idx;
            r.a = int32_t(idx); r.b = 1;
            p.weights.push_back(1.0f);
            break;
        }
        }
        return r;
    }
};

}  // namespace

static void buildPixelRecords(Plan &p)
{
    const long N = long(p.N), P = long(p.P);
    const size_t numBins = size_t(N >> 1);
    const float *mf = p.mapped.data();
    const float topFrequency = p.cfg.sample_rate / 2;                  // TransformDSP.inl:523
    const float freqToBin = float(float(numBins) / topFrequency);      // :524
    RecBuilder rb(p);
    p.recs.assign(size_t(p.sides) * size_t(P), PixelRec{});
    p.weights.clear();
    PixelRec *left = p.recs.data();
    PixelRec *right = p.sides == 2 ? p.recs.data() + P : nullptr;
    long x = 0, oldBin = 0;

    auto maxRec = [&](long xx) {
        const long bin = long(size_t(mf[xx] * freqToBin));              // :612 / :948
        long diff = bin - oldBin;
        long counter = diff ? 1 : 0;
        long first = oldBin + counter, cnt = 0;
        do { ++cnt; ++counter; --diff; } while (diff > 0);
        PixelRec r{}; r.kind = 1; r.a = int32_t(first); r.b = int32_t(cnt); r.c = int32_t(bin);
        oldBin = bin;
        return r;
    };

    const uint32_t mode = p.cfg.channel_mode;
    if (mode == SGZ_CH_COMPLEX) {                                      // TransformDSP.inl:987-1097
        const double fftBandwidth = 1.0 / double(numBins * 2);
        p.breakPixel = uint32_t(P);
        bool firstBreak = true;
        while (x < P) {
            for (; x < P; ++x) {
                if (x != P - 1) {
                    const double bw = double((mf[x + 1] - mf[x]) / topFrequency);
                    if (bw > fftBandwidth) break;
                }
                left[x] = rb.interp(mf[x] * freqToBin, size_t(N));
            }
            if (firstBreak) { p.breakPixel = uint32_t(x); firstBreak = false; }
            if (x != P) oldBin = long(mf[x] * freqToBin);
            for (; x < P; ++x) {
                if (x != P - 1) {
                    const double bw = double((mf[x + 1] - mf[x]) / topFrequency);
                    if (bw < fftBandwidth) break;
                }
                left[x] = maxRec(x);
            }
        }
        return;
    }

    const double fftBandwidth = 1.0 / double(numBins);
    for (x = 0; x < P - 1; ++x) {                                       // :568-601 / :878-936
        const double bw = double((mf[x + 1] - mf[x]) / topFrequency);
        if (bw > fftBandwidth) break;
        const float pos = mf[x] * freqToBin;
        left[x] = rb.interp(pos, numBins);
        if (right) {
            if (p.cfg.bin_interp == SGZ_INTERP_NONE) {
                PixelRec r = left[x];
                r.a = int32_t(N - long(left[x].a));                       // csf[N - index], :933
                r.c = int32_t(p.weights.size());
                p.weights.push_back(1.0f);
                right[x] = r;
            } else {
                right[x] = rb.interp(float(N) - pos, numBins);            // N - (float), :893,:912
            }
        }
    }
    p.breakPixel = uint32_t(x);
    oldBin = long(mf[x] * freqToBin);                                    // :606 / :942
    for (; x < P; ++x) {
        left[x] = maxRec(x);
        if (right) {                                                     // offsets mirrored: csf[N - offset], :963
            PixelRec r = left[x];
            right[x] = r;                                                // device applies N - offset for side 1
        }
    }
}

// SpectrumChannels::Phase (TransformDSP.inl:643-853).  recs[0..P) = left record, recs[P..2P) = right record of the
// interpolated pixels; phaseType[x]: 0 = interpolated (cancellation + magnitude), 1 = arg-max run, 2 = interpolated
// magnitude only (the last pixel of a view that never leaves the interpolation branch: its cancellation is never written);
// phaseNorm[x] = normalizedPosition when pixel x's magnitude is evaluated (bins k < phaseNorm and N-k with k < phaseNorm
// have been replaced by their magnitudes by then, Q6); phaseNormFinal = its value for the arg-max pixels.
static void buildPhaseRecords(Plan &p)
{
    const long N = long(p.N), P = long(p.P);
    const size_t numBins = size_t(N >> 1);
    const float *mf = p.mapped.data();
    const float topFrequency = p.cfg.sample_rate / 2;
    const float freqToBin = float(float(numBins) / topFrequency);
    RecBuilder rb(p);
    p.recs.assign(size_t(2) * size_t(P), PixelRec{});
    p.weights.clear();
    p.phaseType.assign(size_t(P), 1u);
    p.phaseNorm.assign(size_t(P), 0u);
    PixelRec *left = p.recs.data(), *right = p.recs.data() + P;
    const double fftBandwidth = 1.0 / double(numBins);
    const bool filtered = p.cfg.bin_interp == SGZ_INTERP_LINEAR || p.cfg.bin_interp == SGZ_INTERP_LANCZOS;
    size_t breakingPoint = size_t(P);                                    // bandWidthBreakingPoint, :665
    long x = 0;
    for (x = 0; x < P - 1; ++x) {
        const double bw = double((mf[x + 1] - mf[x]) / topFrequency);
        if (bw > fftBandwidth) { breakingPoint = size_t(x); break; }
        const float pos = mf[x] * freqToBin;
        if (filtered) {
            left[x] = rb.interp(pos, numBins);
            right[x] = rb.interp(float(N) - pos, numBins);
        } else {                                                         // None: clamp to numBins - 1 here (:791, Q5)
            left[x] = rb.interp(pos, numBins - 1);
            PixelRec r = left[x];
            r.a = int32_t(N - long(left[x].a));
            r.c = int32_t(p.weights.size());
            p.weights.push_back(1.0f);
            right[x] = r;
        }
        p.phaseType[size_t(x)] = 0u;
    }
    uint32_t normalizedPosition = 0;
    if (filtered) {
        const size_t filterSize = p.cfg.bin_interp == SGZ_INTERP_LINEAR ? 1 : 5;
        if (breakingPoint == size_t(P)) {                                // no break: the magnitude pass also covers pixel P-1
            const float pos = mf[P - 1] * freqToBin;
            left[P - 1] = rb.interp(pos, numBins);
            right[P - 1] = rb.interp(float(N) - pos, numBins);
            p.phaseType[size_t(P - 1)] = 2u;
        }
        for (long xx = 0; xx < long(breakingPoint); ++xx) {              // :714-727 / :759-775
            const float binPosition = mf[xx] * freqToBin;
            while ((binPosition + float(filterSize)) > float(normalizedPosition) && size_t(xx) < breakingPoint - (filterSize + 1))
                ++normalizedPosition;
            p.phaseNorm[size_t(xx)] = normalizedPosition;
        }
    }
    p.phaseNormFinal = normalizedPosition;
    if (filtered) x = long(breakingPoint);      // the magnitude pass leaves x there (:714 / :759): P when the view never breaks
    p.breakPixel = uint32_t(x);
    long oldBin = 0;
    if (x < P) oldBin = long(mf[x] * freqToBin);                         // :806-807
    for (; x < P; ++x) {
        const long bin = long(size_t(mf[x] * freqToBin));
        long diff = bin - oldBin;
        long counter = diff ? 1 : 0;
        const long first = oldBin + counter;
        long cnt = 0;
        do { ++cnt; ++counter; --diff; } while (diff > 0);
        PixelRec r{}; r.kind = 1; r.a = int32_t(first); r.b = int32_t(cnt); r.c = 0;   // maxBin starts at 0 (:813)
        left[x] = r; right[x] = r;
        p.phaseType[size_t(x)] = 1u;
        oldBin = bin;
    }
}

static void buildTwiddles(Plan &p)
{
    // N = R^3 fast path (R = 16 or 32).  Factorised tables (see TwFactors in spectrum_fft.hip):
    //   tw1 rows [W_N^{t b}] b = 1..3, then [W_N^{t 4a}] a = 1..R/4-1, t < R^2
    //   tw2 rows [W_T^{t2 b}], [W_T^{t2 4a}], t2 < R, T = R^2
    const double kTwoPi = 6.28318530717958647692;
    int R = 0;
    if (p.N == 32768) R = 32; else if (p.N == 4096) R = 16;
    // N = 2 R^3: two half-frame workgroups (decimation in frequency) + the generic map kernel
    int halfR = 0;
    if (p.N == 65536) halfR = 32; else if (p.N == 8192) halfR = 16;
    // Phase keeps complex bins: its split and map run as HBM-resident kernels (spectrum_generic.hip); at N = R^3 the transform
    // itself still comes from the in-register FFT (stftComplexKernel), which needs the R tables
    const int phaseR = p.cfg.channel_mode == SGZ_CH_PHASE ? R : 0;
    if (p.cfg.channel_mode == SGZ_CH_PHASE) R = halfR = 0;
    p.fused = R != 0;
    p.halves = halfR != 0;
    p.phaseFusedFft = phaseR != 0;
    if (!R) {   // generic Stockham path: one table W_N^i, i < N/2 (also behind the halves path's map-from-bins test hook)
        p.twN.resize(size_t(p.N / 2) * 2);
        for (uint32_t i = 0; i < p.N / 2; ++i) {
            const double ang = -kTwoPi * double(i) / double(p.N);
            p.twN[size_t(i) * 2 + 0] = float(std::cos(ang));
            p.twN[size_t(i) * 2 + 1] = float(std::sin(ang));
        }
        if (!halfR && !phaseR) return;
        R = halfR ? halfR : phaseR;
    }
    const uint32_t fftN = p.halves ? p.N / 2 : p.N;     // size of the in-register transform
    const uint32_t T = uint32_t(R * R);
    const int rows = 3 + R / 4 - 1;
    auto mult = [&](int row) { return row < 3 ? row + 1 : 4 * (row - 3 + 1); };
    p.tw1.resize(size_t(rows) * T * 2);
    for (int row = 0; row < rows; ++row)
        for (uint32_t t = 0; t < T; ++t) {
            const uint64_t m = (uint64_t(t) * uint64_t(mult(row))) % fftN;
            const double ang = -kTwoPi * double(m) / double(fftN);
            p.tw1[(size_t(row) * T + t) * 2 + 0] = float(std::cos(ang));
            p.tw1[(size_t(row) * T + t) * 2 + 1] = float(std::sin(ang));
        }
    if (p.halves) {
        // odd half (TwFactors<LR, true>): rows A_1..A_3 = W_M^{t b}, then B_a U = W_N^{t (8a + 1)}, a = 0..R/4-1  (M = N/2)
        const int rowsOdd = 3 + R / 4;
        p.tw1odd.resize(size_t(rowsOdd) * T * 2);
        for (int row = 0; row < rowsOdd; ++row)
            for (uint32_t t = 0; t < T; ++t) {
                const uint64_t m = row < 3 ? (uint64_t(t) * uint64_t(2 * (row + 1))) % p.N
                                           : (uint64_t(t) * uint64_t(8 * (row - 3) + 1)) % p.N;
                const double ang = -kTwoPi * double(m) / double(p.N);
                p.tw1odd[(size_t(row) * T + t) * 2 + 0] = float(std::cos(ang));
                p.tw1odd[(size_t(row) * T + t) * 2 + 1] = float(std::sin(ang));
            }
    }
    p.tw2.resize(size_t(rows) * R * 2);
    for (int row = 0; row < rows; ++row)
        for (int t2 = 0; t2 < R; ++t2) {
            const uint32_t m = uint32_t(t2 * mult(row)) % T;
            const double ang = -kTwoPi * double(m) / double(T);
            p.tw2[(size_t(row) * R + t2) * 2 + 0] = float(std::cos(ang));
            p.tw2[(size_t(row) * R + t2) * 2 + 1] = float(std::sin(ang));
        }
}

// Tables of the chunk-scan pixel map (chunk_map.hpp) for the channel-split kernels: `recs` = the records those kernels map
// (recsReal when some pixels are settled elsewhere), nSides = 2 (pairs) or 1 (mono).  Returns false when the map does not apply
// (overlapping runs, a tap window that does not fit kTapFloats: neither occurs with the reference's mapping, but the kernel must not
// be handed tables it would misread).
static bool buildChunkMap(Plan &p, const std::vector<PixelRec> &recs, int nSides)
{
    const long N = long(p.N), M = N / 2;
    const uint32_t T = uint32_t(M / 32);
    p.chunkEnds.assign(size_t(nSides) * T, 0u);
    p.chunkReBase.assign(size_t(nSides) * T, 0u);
    p.chunkRec.assign(size_t(nSides) * p.P * 2, 0u);
    p.weights12.assign(size_t(kTapFloats), 0.0f);                      // row 0: what the pixels without taps read
    for (int side = 0; side < nSides; ++side) {
        const long off = side ? M : 0;
        std::vector<uint8_t> end(size_t(M), 0), owned(size_t(M), 0);
        struct Tile { long lo, hi; uint32_t x; };
        std::vector<Tile> tiles;
        for (uint32_t x = 0; x < p.P; ++x) {
            const PixelRec &rec = recs[size_t(side) * p.P + x];
            uint32_t *cr = &p.chunkRec[(size_t(side) * p.P + x) * 2];
            if (rec.kind == 0) {
                const long i0 = long(rec.a) - off;
                if (i0 < 0 || i0 + rec.b - 1 > M || rec.b > kMaxTaps) return false;
                cr[0] = uint32_t(p.weights12.size() / kTapFloats);
                if (chunkPos(int(i0)) > 0xFFFF) return false;
                cr[1] = uint32_t(chunkPos(int(i0))) | kChunkInterp;
                p.weights12.insert(p.weights12.end(), size_t(kTapFloats), 0.0f);
                float *w = &p.weights12[p.weights12.size() - kTapFloats];
                for (int t = 0; t < rec.b; ++t) {
                    const int d = chunkPos(int(i0 + t)) - chunkPos(int(i0));
                    if (d < 0 || d >= kTapFloats) return false;
                    w[d] = p.weights[size_t(rec.c) + size_t(t)];
                }
            } else if (rec.kind & 1) {
                // offsets [a, a + b): csf index = offset (left) or N - offset (right); side-local entry i = index - off
                long lo = side ? M - (long(rec.a) + rec.b - 1) : long(rec.a);
                long hi = side ? M - long(rec.a) : long(rec.a) + rec.b - 1;
                if (lo < 0 || hi > M) return false;
                uint32_t flags = 0;
                if (hi == M) { flags |= kChunkPlusM; hi = M - 1; }
                if (lo > hi) { cr[0] = 0; cr[1] = flags | kChunkNoScan; }
                else if (lo == hi) { cr[0] = uint32_t(chunkPos(int(lo))); cr[1] = flags | kChunkDirect; }
                else {
                    for (long i = lo; i <= hi; ++i) { if (owned[size_t(i)]) return false; owned[size_t(i)] = 1; }
                    tiles.push_back(Tile{lo, hi, x});
                    cr[1] = flags;                                       // completed below
                }
            } else cr[1] = kChunkOff;                                    // (a pixel the channel-split kernels leave to realLateKernel)
        }
        for (const Tile &t : tiles) {
            end[size_t(t.hi)] = 1;
            if (t.lo >= 1 && !owned[size_t(t.lo - 1)]) end[size_t(t.lo - 1)] = 1;      // an unowned entry right before a tile: restart the maximum behind it
        }
        std::vector<uint32_t> rank(size_t(M) + 1, 0);
        for (long i = 0; i < M; ++i) rank[size_t(i) + 1] = rank[size_t(i)] + end[size_t(i)];
        if (rank[size_t(M)] > 0xFFFFu) return false;
        p.chunkSlots[side] = rank[size_t(M)];
        for (uint32_t t = 0; t < T; ++t) p.chunkReBase[size_t(side) * T + t] = rank[size_t(t) * 32];
        for (long i = 0; i < M; ++i)
            if (end[size_t(i)]) p.chunkEnds[size_t(side) * T + size_t(i >> 5)] |= 1u << (i & 31);
        for (const Tile &t : tiles) {
            uint32_t *cr = &p.chunkRec[(size_t(side) * p.P + t.x) * 2];
            const uint32_t c0 = uint32_t(t.lo >> 5), nC = uint32_t(t.hi >> 5) - c0;
            if (nC > 0xFFFFu) return false;
            cr[0] = rank[size_t(t.hi)] | (c0 << 16);
            cr[1] |= nC;
        }
    }
    return true;
}

// ---- RSNT: CComplexResonator::Constant::mapSystemHz(mappedFrequencies, size, numVectors, sampleRate, freeQ, 8, windowSize)
// (TransformConstant.h:120-123; numVectors = cpl::dsp::windowCoefficients(window).second, Spectrum.cpp:593).  cpl is absent: restated
// from the mathematics it implements (UNVERIFIED vs cpl; oracle/resonator.c states the same choices and is the checker):
//   - filter i sits on mappedFrequencies[i]; its bandwidth B_i is the spacing to the next axis point (the last reuses the one before);
//     unless Q is free the equivalent window length fs / B_i is bounded by the window size;  pole radius r = exp(-pi B / fs);
//   - a K-term cosine-sum window  w[n] = sum (-1)^m a_m cos(2 pi m n / N)  becomes the frequency-domain kernel
//     a_0 X[k] - a_1/2 (X[k-1] + X[k+1]) + a_2/2 (...) - ...  on resonators detuned by m B_i: V = 2K - 1 "vectors";
//   - gain 1 - r: a full-scale sine on the centre reads 1/2 unwindowed (the two-sided convention of the FFT branch).
static int windowCosineTerms(uint32_t type, double a[5])
{
    for (int i = 0; i < 5; ++i) a[i] = 0.0;
    switch (type) {
    case SGZ_WIN_HANN: a[0] = 0.5; a[1] = 0.5; return 2;
    case SGZ_WIN_HAMMING: a[0] = 0.54; a[1] = 0.46; return 2;
    case SGZ_WIN_BLACKMAN: a[0] = 0.42; a[1] = 0.5; a[2] = 0.08; return 3;
    case SGZ_WIN_EXACT_BLACKMAN: a[0] = 7938.0 / 18608.0; a[1] = 9240.0 / 18608.0; a[2] = 1430.0 / 18608.0; return 3;
    case SGZ_WIN_NUTTALL: a[0] = 0.355768; a[1] = 0.487396; a[2] = 0.144232; a[3] = 0.012604; return 4;
    case SGZ_WIN_BLACKMAN_NUTTALL: a[0] = 0.3635819; a[1] = 0.4891775; a[2] = 0.1365995; a[3] = 0.0106411; return 4;
    case SGZ_WIN_BLACKMAN_HARRIS: a[0] = 0.35875; a[1] = 0.48829; a[2] = 0.14128; a[3] = 0.01168; return 4;
    case SGZ_WIN_FLATTOP: a[0] = 0.21557895; a[1] = 0.41663158; a[2] = 0.277263158; a[3] = 0.083578947; a[4] = 0.006947368; return 5;
    default: a[0] = 1.0; return 1;       // no cosine-sum form: unwindowed
    }
}

static void buildResonator(Plan &p)
{
    const sgz_spectrum_config &cfg = p.cfg;
    double a[5];
    const int K = windowCosineTerms(cfg.window_type, a), V = 2 * K - 1;
    const uint32_t P = p.P;
    const double fs = double(cfg.sample_rate);
    p.resV = V;
    for (int v = 0; v < V; ++v) {
        const int m = v - (K - 1), am = m < 0 ? -m : m;
        p.resWeights[v] = float(am == 0 ? a[0] : ((am & 1) ? -0.5 : 0.5) * a[am]);
    }
    p.resCoeff.assign(size_t(V) * P * 2, 0.f);
    p.resPow.assign(size_t(V) * P * 4, 0.f);
    p.resPowB.assign(size_t(V) * P * 8 * 2, 0.f);
    p.resPowBLo.assign(size_t(V) * P * 2 * 2, 0.f);
    // matrix-core form of the frames from rest (resonator.hip resonateMfmaKernel): needs whole tiles of 32 blocks x 32 samples per frame
    const bool mfma = cfg.hop % 1024u == 0;
    p.resW1.clear(); p.resW2.clear(); p.resTile.clear(); p.resW1b.clear();
    if (mfma) { p.resW1.assign(size_t(V) * P * 32 * 2, 0.f); p.resW2.assign(size_t(V) * P * 32 * 2, 0.f); p.resTile.assign(size_t(V) * P * 4, 0.f); p.resW1b.assign(size_t(24) * V * P * 4, 0u); }
    p.resGain.assign(P, 0.f);
    for (uint32_t i = 0; i < P; ++i) {
        const uint32_t k = i + 1 >= P ? P - 2 : i;
        const double hDiff = std::fabs(double(p.mapped[k + 1]) - double(p.mapped[k]));
        double length = hDiff > 0 ? fs / hDiff : double(cfg.window_size);
        if (!cfg.free_q && length > double(cfg.window_size)) length = double(cfg.window_size);
        if (length < 2.0) length = 2.0;
        const double B = fs / length;
        const double r = std::exp(-3.14159265358979323846 * B / fs);
        p.resGain[i] = float(1.0 - r);
        for (int v = 0; v < V; ++v) {
            const double omega = 2.0 * 3.14159265358979323846 * (double(p.mapped[i]) + double(v - (K - 1)) * B) / fs;
            const float cr = float(r * std::cos(omega)), ci = float(r * std::sin(omega));
            p.resCoeff[(size_t(v) * P + i) * 2] = cr;
            p.resCoeff[(size_t(v) * P + i) * 2 + 1] = ci;
            // (the fp32 pole)^hop in double by repeated squaring: what `hop` steps of the fp32 recurrence multiply an old state by, up
            // to the recurrence's own rounding -- the rounding of the pole itself (6e-8 x hop) must NOT be left out of the power
            double br = cr, bi = ci, pr = 1.0, pi = 0.0;
            for (uint32_t e = cfg.hop; e; e >>= 1) {
                if (e & 1u) { const double t = pr * br - pi * bi; pi = pr * bi + pi * br; pr = t; }
                const double t = br * br - bi * bi; bi = 2.0 * br * bi; br = t;
            }
            // (hi, lo) pairs: a power rounded to fp32 is a slightly different pole -- the same relative 3e-8 at EVERY step of the chain, which
            // a resonator that remembers 1e5 samples turns into 1e-4 of its state; the low words take that systematic part out
            p.resPow[(size_t(v) * P + i) * 4] = float(pr);
            p.resPow[(size_t(v) * P + i) * 4 + 1] = float(pi);
            p.resPow[(size_t(v) * P + i) * 4 + 2] = float(pr - double(float(pr)));
            p.resPow[(size_t(v) * P + i) * 4 + 3] = float(pi - double(float(pi)));
            if (mfma) {
                // W1[b] = pole^(31 - b): the weight of sample b of a 32-sample block;  W2[a] = pole^(32 (31 - a)): the weight of block a of a
                // 32-block tile;  pole^1024 (hi, lo): what a tile multiplies the state by.  All from the fp32 pole in double, rounded once.
                auto cpowd = [&](uint32_t e, double &xr, double &xi) {
                    double br = cr, bi = ci; xr = 1.0; xi = 0.0;
                    for (; e; e >>= 1) {
                        if (e & 1u) { const double t = xr * br - xi * bi; xi = xr * bi + xi * br; xr = t; }
                        const double t = br * br - bi * bi; bi = 2.0 * br * bi; br = t;
                    }
                };
                for (uint32_t k = 0; k < 32; ++k) {
                    double xr, xi;
                    cpowd(31 - k, xr, xi);
                    p.resW1[((size_t(k) * V + v) * P + i) * 2] = float(xr); p.resW1[((size_t(k) * V + v) * P + i) * 2 + 1] = float(xi);
                    cpowd(32 * (31 - k), xr, xi);
                    p.resW2[((size_t(k) * V + v) * P + i) * 2] = float(xr); p.resW2[((size_t(k) * V + v) * P + i) * 2 + 1] = float(xi);
                }
                // the same weights as three bfloat16 parts each: h = bf16(x), m = bf16(x - h), l = x - h - m (at most eight significant bits
                // are left: exact), round to nearest even; entry [kh][h][part][re / im] holds samples 16 kh + 8 h + 0 .. 7 of a block
                auto bf16High = [](float x) { uint32_t u; std::memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return u & 0xffff0000u; };
                auto asFloat = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
                for (uint32_t kh = 0; kh < 2; ++kh)
                    for (uint32_t hh = 0; hh < 2; ++hh)
                        for (uint32_t c = 0; c < 2; ++c) {
                            uint32_t part[3][8];
                            for (uint32_t e = 0; e < 8; ++e) {
                                const float x = p.resW1[((size_t(16 * kh + 8 * hh + e) * V + v) * P + i) * 2 + c];
                                part[0][e] = bf16High(x);
                                const float r1 = x - asFloat(part[0][e]);
                                part[1][e] = bf16High(r1);
                                const float r2 = r1 - asFloat(part[1][e]);
                                std::memcpy(&part[2][e], &r2, 4);
                                part[2][e] &= 0xffff0000u;
                            }
                            for (uint32_t q = 0; q < 3; ++q) {
                                uint32_t *dst = &p.resW1b[((size_t(((kh * 2 + hh) * 3 + q) * 2 + c) * V + v) * P + i) * 4];
                                for (uint32_t w = 0; w < 4; ++w) dst[w] = (part[q][2 * w] >> 16) | part[q][2 * w + 1];
                            }
                        }
                double tr, ti;
                cpowd(1024, tr, ti);
                float *q = &p.resTile[(size_t(v) * P + i) * 4];
                q[0] = float(tr); q[1] = float(ti); q[2] = float(tr - double(q[0])); q[3] = float(ti - double(q[1]));
            }
            // (the fp32 pole)^1 .. ^8, each rounded once from double: what the block steps of the frames from rest multiply by -- powers
            // built up in fp32 would be a slightly different pole, and a resonator that remembers 1e5 samples notices 1e-7 of that
            double qr = cr, qi = ci;
            for (int k = 0; k < 8; ++k) {
                p.resPowB[((size_t(v) * P + i) * 8 + k) * 2] = float(qr);
                p.resPowB[((size_t(v) * P + i) * 8 + k) * 2 + 1] = float(qi);
                if (k == 3 || k == 7) {                       // the block lengths in use (4, 8): low words of pole^B
                    p.resPowBLo[((size_t(v) * P + i) * 2 + (k == 7 ? 1 : 0)) * 2] = float(qr - double(float(qr)));
                    p.resPowBLo[((size_t(v) * P + i) * 2 + (k == 7 ? 1 : 0)) * 2 + 1] = float(qi - double(float(qi)));
                }
                const double t = qr * double(cr) - qi * double(ci); qi = qr * double(ci) + qi * double(cr); qr = t;
            }
        }
    }
}

sgz_status buildPlan(const sgz_spectrum_config &cfg, Plan &p, std::string &err)
{
    if (cfg.window_size < 1 || cfg.axis_points < 2 || cfg.num_pairs < 1 || cfg.hop < 1 || !(cfg.sample_rate >= 1)) {
        err = "invalid spectrum config (window_size>=1, axis_points>=2 (TransformConstant.h:127), num_pairs>=1, hop>=1, sample_rate>=1)";
        return SGZ_EINVAL;
    }
    if (cfg.channel_mode > SGZ_CH_COMPLEX || cfg.bin_interp > SGZ_INTERP_LANCZOS || cfg.view_scaling > SGZ_VIEW_LOG ||
        cfg.window_type >= SGZ_WIN_END) {
        err = "invalid enum value in spectrum config";
        return SGZ_EINVAL;
    }
    // The reference's GUI clamps the view to [0, 1] and only asserts on the rest in debug builds (TransformDSP.inl:571-577);
    // a raw C ABI cannot rely on that: a frequency beyond Nyquist would index past csf[N] in every map kernel.
    const double reals[] = {cfg.view_left, cfg.view_right, cfg.min_log_freq, cfg.low_db, cfg.high_db, cfg.clip_db, cfg.slope_a,
                            cfg.slope_b, cfg.window_alpha, cfg.window_beta, double(cfg.pole[0]), double(cfg.pole[1]),
                            cfg.ratios[0], cfg.ratios[1], cfg.ratios[2], cfg.ratios[3], cfg.ratios[4], double(cfg.sample_rate)};
    for (double v : reals)
        if (!std::isfinite(v)) { err = "non-finite value in spectrum config"; return SGZ_EINVAL; }
    if (!(cfg.view_left >= 0.0) || !(cfg.view_right <= 1.0) || !(cfg.view_right > cfg.view_left)) {
        err = "view must satisfy 0 <= view_left < view_right <= 1";
        return SGZ_EINVAL;
    }
    if (cfg.view_scaling == SGZ_VIEW_LOG && (!(cfg.min_log_freq > 0.0) || !(cfg.min_log_freq < double(cfg.sample_rate) * 0.5))) {
        err = "min_log_freq must lie in (0, sample_rate / 2)";
        return SGZ_EINVAL;
    }
    if (cfg.window_size > (1u << 24) || cfg.axis_points > (1u << 20) || cfg.num_pairs > 4096) {
        err = "window_size <= 2^24, axis_points <= 2^20, num_pairs <= 4096";
        return SGZ_EINVAL;
    }
    if (!(cfg.high_db > cfg.low_db)) { err = "high_db must exceed low_db"; return SGZ_EINVAL; }
    if (cfg.algorithm > SGZ_ALGO_RSNT) { err = "algorithm must be SGZ_ALGO_FFT or SGZ_ALGO_RSNT"; return SGZ_EINVAL; }
    p.cfg = cfg;
    p.W = cfg.window_size;
    p.N = transformSizeFor(p.W);
    p.log2N = 0; while ((1u << p.log2N) < p.N) ++p.log2N;
    p.P = cfg.axis_points;
    p.C = cfg.num_pairs;
    // two planes of P values per (frame, pair): left / right magnitudes, or (Phase) magnitude / cancellation
    p.sides = (cfg.channel_mode == SGZ_CH_SEPARATE || cfg.channel_mode == SGZ_CH_MIDSIDE || cfg.channel_mode == SGZ_CH_PHASE) ? 2 : 1;
    p.stateChannels = cfg.channel_mode > SGZ_CH_SIDE ? 2 : 1;

    // windowKernel (N entries; [W,N) stay zero = the zero padding of prepareTransform :220-223)
    p.window.assign(p.N, 0.0f);
    p.windowScale = designWindow(cfg.window_type, cfg.window_symmetry, cfg.window_alpha, cfg.window_beta, p.W, p.window.data());

    // mappedFrequencies, TransformConstant.h:125-180
    p.mapped.resize(p.P);
    {
        const double viewSize = cfg.view_right - cfg.view_left;
        const double sampleRate = double(cfg.sample_rate);
        if (cfg.view_scaling == SGZ_VIEW_LINEAR) {
            const double halfSampleRate = sampleRate * 0.5;
            const double complexFactor = cfg.channel_mode == SGZ_CH_COMPLEX ? 2.0 : 1.0;
            const double freqPerPixel = halfSampleRate / double(p.P - 1);
            for (uint32_t i = 0; i < p.P; ++i)
                p.mapped[i] = float(complexFactor * cfg.view_left * halfSampleRate + complexFactor * viewSize * double(i) * freqPerPixel);
        } else {
            const double sampleSize = double(p.P - 1);
            const double minFreq = cfg.min_log_freq;
            const double end = double(cfg.sample_rate / 2);
            for (uint32_t i = 0; i < p.P; ++i) {
                if (cfg.channel_mode != SGZ_CH_COMPLEX) {
                    p.mapped[i] = float(minFreq * std::pow(end / minFreq, cfg.view_left + viewSize * (double(i) / sampleSize)));
                } else {
                    double arg = cfg.view_left + viewSize * double(i) / sampleSize;
                    if (arg < 0.5) p.mapped[i] = float(minFreq * std::pow(end / minFreq, arg * 2));
                    else {
                        arg -= 0.5;
                        const double power = minFreq * std::pow(end / minFreq, 1.0 - arg * 2);
                        p.mapped[i] = float(end + (end - power));
                    }
                }
            }
        }
    }
    // slopeMap, TransformConstant.h:109-118
    p.slope.resize(p.P);
    {
        const float a = float(cfg.slope_a), b = float(cfg.slope_b);
        for (uint32_t i = 0; i < p.P; ++i) p.slope[i] = b * powf(p.mapped[i], a);   // std::pow(float, float) is powf
    }
    // scalars
    DeviceScalars &s = p.scalars;
    s.invSize = float(p.windowScale / (double(p.W) * 0.5));
    for (int k = 0; k < SGZ_NUM_GRAPHS; ++k) { s.pole[k] = cfg.pole[k]; s.phasePole[k] = std::pow(cfg.pole[k], 0.3f); }
    {
        const double lowerFraction = std::pow(10.0, cfg.low_db / 20.0);
        const double upperFraction = std::pow(10.0, cfg.high_db / 20.0);
        s.deltaYRecip = float(1.0 / std::log(upperFraction / lowerFraction));
        s.minFracRecip = float(1.0 / lowerFraction);
        s.lowerClip = float(cfg.clip_db);
    }
    {   // Spectrum::calculateSpectrumColourRatios, Spectrum.cpp:226-246
        double acc = 0.0, vals[SGZ_NUM_SPEC_COLOURS];
        for (int i = 0; i < SGZ_NUM_SPEC_COLOURS; ++i) { vals[i] = std::max(0.0001, cfg.ratios[i]); acc += vals[i]; }
        acc += double(FLT_EPSILON);
        s.ratios[0] = 0;
        for (int i = 0; i < SGZ_NUM_SPEC_COLOURS; ++i) s.ratios[i + 1] = float(vals[i] / acc);
    }
    // colour stops per pair: generateSpectrogramColourRotation (TransformConstant.h:55-65) through
    // ColourRotation::operator[] (CommonSignalizer.h:931-937) and FloatColour (:1139-1163)
    p.colourTables.resize(size_t(p.C) * (SGZ_NUM_SPEC_COLOURS + 1) * 3);
    for (uint32_t pair = 0; pair < p.C; ++pair)
        for (int i = 0; i <= SGZ_NUM_SPEC_COLOURS; ++i) {
            const uint32_t rotation = i == 0 ? 0u : pair;
            uint8_t rgb[3];
            rotateHueRgb8(cfg.colours[i], float(rotation) / float(p.C), rgb);
            for (int c = 0; c < 3; ++c)
                p.colourTables[(size_t(pair) * (SGZ_NUM_SPEC_COLOURS + 1) + i) * 3 + c] = float(rgb[c]) / 255.0f;
        }
    if (cfg.algorithm == SGZ_ALGO_RSNT) {
        // no transform tables: the axis points ARE the filters (mapToLinearSpace's RSNT branch copies the resonator state, :1103-1133)
        buildResonator(p);
        return SGZ_OK;
    }
    if (cfg.channel_mode == SGZ_CH_PHASE) buildPhaseRecords(p);
    else buildPixelRecords(p);
    p.weights.insert(p.weights.end(), size_t(kMaxTaps), 0.0f);    // padding: the kernel reads kMaxTaps weights unconditionally
    // The fused kernel (N = R^3) reads a tap window as kMaxTaps + 1 CONTIGUOUS floats of its bank-padded csf array (one pad slot,
    // holding 0, after every R entries; csf[0 .. 9] repeated behind csf[N] for the periodic indexing).  Its weights: the record's
    // taps in order, +0 at the position of the pad slot the window steps over (if any), +0 behind the last tap.
    p.weights11.clear();
    {
        const uint32_t R = p.N == 32768u ? 32u : (p.N == 4096u ? 16u : 0u);
        if (R && cfg.channel_mode != SGZ_CH_PHASE) {
            p.weights11.assign(p.recs.size() * size_t(kMaxTaps + 1), 0.0f);
            for (size_t r = 0; r < p.recs.size(); ++r) {
                const PixelRec &rec = p.recs[r];
                if (rec.kind != 0) continue;
                float *w = &p.weights11[r * size_t(kMaxTaps + 1)];
                // taps i at csf index a + i; LDS offset from the first tap: i, +1 from the first i >= 1 with (a + i) % R == 0 on
                const uint32_t rem = uint32_t(rec.a) % R;
                const int cross = rem ? int(R - rem) : kMaxTaps + 1;       // first tap behind a pad slot (none if the window starts a block)
                for (int i = 0; i < rec.b; ++i) w[i < cross ? i : i + 1] = p.weights[size_t(rec.c) + size_t(i)];
            }
        }
    }
    // balanced arg-max work list: every kind-1 record cut at 16-aligned csf windows (see MaxItem)
    p.items.clear();
    p.nItemsLeft = 0;
    for (size_t r = 0; r < p.recs.size() && cfg.channel_mode != SGZ_CH_PHASE; ++r) {   // (Phase runs the generic map kernel)
        PixelRec &rec = p.recs[r];
        if (rec.kind != 1) continue;
        const bool right = r >= size_t(p.P);
        rec.kind = 1 + 2 * int32_t(p.items.size());      // kind: bit 0 = arg-max record, bits 1.. = index of its first piece
        int32_t pieces = 0;
        // scan order = ascending offset; k = offset (left) or N - offset (right)
        long o = rec.a;
        const long oEnd = long(rec.a) + rec.b;            // exclusive
        while (o < oEnd) {
            const long k = right ? long(p.N) - o : o;
            const long w = k >> 4;
            long lo, hi, n;
            if (!right) { lo = k & 15; n = std::min<long>(16 - lo, oEnd - o); hi = lo + n - 1; }
            else { hi = k & 15; n = std::min<long>(hi + 1, oEnd - o); lo = hi - n + 1; }
            MaxItem it;
            it.win = uint32_t(w) | (uint32_t(lo) << 16) | (uint32_t(hi) << 20);
            p.items.push_back(it);
            ++pieces;
            o += n;
        }
        (void)pieces;                                      // = number of 16-windows the run spans (recomputed on the device)
        if (!right) p.nItemsLeft = uint32_t(p.items.size());
    }
    // The csf entries the reference leaves complex (Complex: csf[0]; Left / Right / Merge / Side: csf[N/2 .. N-1], reached by
    // filter windows that wrap below bin 0 or sit at Nyquist): list the pixels that touch one (complex_dc.hpp redoes them)
    p.dcPixels.clear();
    {
        const uint32_t mode = cfg.channel_mode;
        const bool mono = mode == SGZ_CH_LEFT || mode == SGZ_CH_RIGHT || mode == SGZ_CH_MERGE || mode == SGZ_CH_SIDE;
        const long N = long(p.N);
        auto isComplex = [&](long k) { return mode == SGZ_CH_COMPLEX ? k == 0 : (mono && k >= N / 2 && k < N); };
        for (size_t r = 0; r < p.recs.size() && (mono || mode == SGZ_CH_COMPLEX); ++r) {
            const PixelRec &rec = p.recs[r];
            bool hit = false;
            if (rec.kind == 0) {
                long k = rec.a;
                for (int i = 0; i < rec.b; ++i) { hit = hit || isComplex(k); k = (k == N) ? 0 : k + 1; }
            } else if (rec.kind & 1) {
                for (long k = rec.a; k < long(rec.a) + rec.b; ++k) hit = hit || isComplex(k);
            }
            if (hit) p.dcPixels.push_back(uint32_t(r));
        }
    }
    buildTwiddles(p);
    // halves and generic paths: mapSideKernel stages k in [N-15, N] + [0, N/2+31] for the left side, [N/2-16, N] + [0, 30] for the right
    // (Complex: the generic map kernel, whose csf order the complex DC redo reads)
    p.sideMapOk = !p.fused && p.cfg.channel_mode != SGZ_CH_PHASE && p.cfg.channel_mode != SGZ_CH_COMPLEX;
    for (size_t r = 0; r < p.recs.size() && p.sideMapOk; ++r) {
        const PixelRec &rec = p.recs[r];
        const bool right = r >= size_t(p.P);
        const long N = long(p.N), half = N / 2;
        auto inside = [&](long k) {
            return right ? ((k >= half - 16 && k <= N) || (k >= 0 && k <= 30)) : ((k >= 0 && k <= half + 31) || (k >= N - 15 && k <= N));
        };
        if (rec.kind == 0) {
            long k = rec.a;
            for (int i = 0; i < rec.b; ++i) { p.sideMapOk = p.sideMapOk && inside(k); k = (k == N) ? 0 : k + 1; }
        } else if (rec.kind & 1) {
            for (long o = rec.a; o < long(rec.a) + rec.b; ++o) p.sideMapOk = p.sideMapOk && inside(right ? N - o : o);
        }
    }
    // the window evaluated inside the kernels (Hann / Hamming, periodic, W == N): w[n] = a0 - a1 cos(2 pi n / N)
    const bool cosWindow = (cfg.window_type == SGZ_WIN_HANN || cfg.window_type == SGZ_WIN_HAMMING) && cfg.window_symmetry == SGZ_WIN_PERIODIC &&
                           p.W == p.N;
    if (cosWindow) {
        p.winP0 = cfg.window_type == SGZ_WIN_HANN ? 0.5f : 0.54f;
        p.winP1 = cfg.window_type == SGZ_WIN_HANN ? -0.5f : -0.46f;
    }
    if (cosWindow && p.fused) {
        const uint32_t T = p.N == 32768 ? 1024u : 256u;
        p.winPhaseT.resize(size_t(T) * 2);
        for (uint32_t t = 0; t < T; ++t) {
            const double ang = 6.28318530717958647692 * double(t) / double(p.N);
            p.winPhaseT[size_t(t) * 2 + 0] = float(std::cos(ang));
            p.winPhaseT[size_t(t) * 2 + 1] = float(std::sin(ang));
        }
    }
    // channel-split path (spectrum_real.hip): eligibility and tables
    p.realSplit = (cfg.channel_mode == SGZ_CH_SEPARATE || cfg.channel_mode == SGZ_CH_MIDSIDE) && (p.N == 16384 || p.N == 32768 || p.N == 65536) && p.W == p.N && (cfg.hop % 2u) == 0u &&
                  p.dcPixels.empty() && !p.items.empty();
    std::vector<uint32_t> lowFix[2];
    if (p.realSplit) {
        const long N = long(p.N), M = N / 2;
        for (size_t r = 0; r < p.recs.size() && p.realSplit; ++r) {
            const PixelRec &rec = p.recs[r];
            const bool right = r >= size_t(p.P);
            auto inside = [&](long k) { return right ? (k >= M && k <= N) : (k >= 0 && k <= M); };
            if (rec.kind == 0) {
                // interpolation taps stay clear of csf[N/2] (the entry that needs both channels) and either on their own side, or -- a
                // window that reaches over bin 0: ..., csf[N-1], csf[N], csf[0], csf[1], ... -- inside the kLowBins lowest entries of
                // the two channels (left csf[j], right csf[N - j], j < kLowBins), which the channels publish: such a pixel is settled
                // by the later workgroup (spectrum_real.hip) and switched off in the kernels' own mapping (recsReal)
                constexpr long kLow = 24;                                // = kLowBins (kernels.hpp)
                long k = rec.a;
                bool foreign = false, nearZero = true;
                for (int i = 0; i < rec.b; ++i) {
                    p.realSplit = p.realSplit && k != M;
                    if (!inside(k)) foreign = true;
                    if (!(k < kLow || k > N - kLow)) nearZero = false;
                    k = (k == N) ? 0 : k + 1;
                }
                if (foreign) {
                    if (nearZero) lowFix[right ? 1 : 0].push_back(uint32_t(r - (right ? size_t(p.P) : 0)));
                    else p.realSplit = false;
                }
            } else if (rec.kind & 1) {
                for (long o = rec.a; o < long(rec.a) + rec.b; ++o) p.realSplit = p.realSplit && inside(right ? N - o : o);
            }
        }
        // arg-max runs that include offset N/2 (csf[N/2] on either side: it is the LAST offset of a scan): the top pixels of a side
        for (int side = 0; side < 2 && p.realSplit; ++side) {
            uint32_t from = p.P;
            for (uint32_t x = 0; x < p.P; ++x) {
                const PixelRec &rec = p.recs[size_t(side) * p.P + x];
                const bool hit = (rec.kind & 1) && long(rec.a) <= M && M < long(rec.a) + rec.b;
                if (hit && from == p.P) from = x;
                if (!hit && from != p.P) p.realSplit = false;            // (not a suffix of the pixel axis: leave it to the whole-frame kernel)
                if (!hit && (rec.kind & 1) && rec.c == M) p.realSplit = false;   // fallback bin N/2 without N/2 in the run
            }
            if (p.P - from > 64) p.realSplit = false;
            p.realFixFrom[side] = from;
        }
        const size_t nLeft = p.nItemsLeft, nRight = p.items.size() - p.nItemsLeft;
        const size_t ldsFloats = size_t(M + 1) + size_t((M + 1) >> 5) + 2;
        const size_t budget = (p.N == 16384 ? size_t(40) : p.N == 32768 ? size_t(80) : size_t(160)) * 1024 - ldsFloats * 4 - 16;
        if (std::max(nLeft, nRight) * 4 > budget) p.realSplit = false;
    }
    // Every eligible plan takes it.  Measured on MI355X (tools/ka_time.py; the hybrid whole-frame + half-task launch of round 2 is in NOTES.md), since the pair exchange stopped costing
    // cache maintenance: N = 65536 (cfg5, 32 pairs) 693 us per K_A pass against 1199 us for the half-frame kernels + map kernel;
    // N = 32768 (cfg2, 348 frames = 696 channel workgroups, two per CU) 38.2 us against 44.0 us for the whole-frame kernel; N = 16384
    // 6.9 M against 2.9 M transforms/s for the generic passes.  sgz_plan_set_option(SGZ_OPT_CHANNEL_SPLIT, 0) keeps a plan off it (A/B runs, parity tests against the other kernels).
    // The mono modes transform ONE real signal per task: the same kernel, one workgroup per (frame, pair), no pair exchange.  It holds
    // csf[0 .. N/2] as magnitudes (csf[N/2] = X[N/2] / 2 is real for a real signal: arg-max runs may end on it) and the kSpecBins entries
    // of complex_dc.hpp the reference leaves complex (csf[N-8 .. N-1] = conj X[8 .. 1], csf[N/2 .. N/2+7]); csf[N] is 0 in these modes.
    // Eligible when every arg-max run stays inside csf[0 .. N/2] and every tap window inside those entries; the pixels whose taps leave the
    // magnitudes (windows that wrap below bin 0 -- the lowest pixels of a view from 0 Hz or of the default view at N = 16384 -- or sit at
    // Nyquist) are redone by complexDcPixel from the complex entries, like the whole-frame kernel does (monoFix).
    std::vector<uint32_t> monoFix;
    p.realMono = (cfg.channel_mode == SGZ_CH_LEFT || cfg.channel_mode == SGZ_CH_RIGHT || cfg.channel_mode == SGZ_CH_MERGE ||
                  cfg.channel_mode == SGZ_CH_SIDE) &&
                 (p.N == 16384 || p.N == 32768 || p.N == 65536) && p.W == p.N && (cfg.hop % 2u) == 0u && !p.items.empty();
    if (p.realMono) {
        const long N = long(p.N), M = N / 2;
        for (size_t r = 0; r < p.recs.size() && p.realMono; ++r) {
            const PixelRec &rec = p.recs[r];
            if (rec.kind == 0) {
                long k = rec.a;
                bool special = false;
                for (int i = 0; i < rec.b; ++i) {
                    const bool plain = k >= 0 && k < M;
                    const bool held = k == N || (k >= N - 8 && k < N) || (k >= M && k < M + 8);      // complex_dc.hpp specSlot (mono modes)
                    p.realMono = p.realMono && (plain || held);
                    special = special || !plain;
                    k = (k == N) ? 0 : k + 1;                                               // periodic over the N + 1 entries
                }
                if (special) monoFix.push_back(uint32_t(r));
            } else if (rec.kind & 1) {
                p.realMono = p.realMono && rec.a >= 0 && long(rec.a) + rec.b - 1 <= M && rec.c <= M;
            }
        }
        if (monoFix.size() > 256) p.realMono = false;
        const size_t ldsFloats = size_t(M + 1) + size_t((M + 1) >> 5) + 2;
        const size_t budget = (p.N == 16384 ? size_t(40) : p.N == 32768 ? size_t(80) : size_t(160)) * 1024 - ldsFloats * 4 - 16 - 2 * 16 * 4;     // (2 x kSpecBins floats of complex entries)
        if (p.items.size() * 4 > budget) p.realMono = false;
    }
    p.recsReal.clear(); p.realLowPixels.clear(); p.realLowCount[0] = p.realLowCount[1] = 0;
    if (p.realSplit && lowFix[0].size() + lowFix[1].size() > 128) p.realSplit = false;      // (one thread settles them)
    if (p.realMono && !monoFix.empty()) {                                                   // (mono: the same fields carry the redone pixels)
        p.recsReal = p.recs;
        p.realLowCount[0] = uint32_t(monoFix.size());
        for (uint32_t x : monoFix) {
            p.realLowPixels.push_back(x);
            p.recsReal[x] = PixelRec{2, 0, 0, 0};
        }
    }
    if (p.realSplit && !(lowFix[0].empty() && lowFix[1].empty())) {
        p.recsReal = p.recs;
        for (int side = 0; side < 2; ++side) {
            p.realLowCount[side] = uint32_t(lowFix[side].size());
            for (uint32_t x : lowFix[side]) {
                p.realLowPixels.push_back(x);
                p.recsReal[size_t(side) * p.P + x] = PixelRec{2, 0, 0, 0};               // neither interpolated nor arg-max: nothing is written
            }
        }
    }
    p.chunkEnds.clear(); p.chunkReBase.clear(); p.chunkRec.clear(); p.weights12.clear(); p.chunkSlots[0] = p.chunkSlots[1] = 0;
    if (p.realSplit || p.realMono) {
        // the pixel map of these kernels (chunk_map.hpp) and its LDS budget: magnitudes + max(pass-2 twiddle table, tile and chunk maxima)
        const bool ok = buildChunkMap(p, p.recsReal.empty() ? p.recs : p.recsReal, p.realSplit ? 2 : 1);
        const size_t Mh = p.N / 2, Tt = Mh / 32;
        const size_t sFloats = size_t(chunkPos(int(Mh))) + 32;
        const size_t extra = std::max<size_t>(size_t(std::max(p.chunkSlots[0], p.chunkSlots[1])) + 1 + Tt, p.N >= 32768 ? 2180 : 0) + 64;   // (+ column 0's scratch)
        const size_t budget = (p.N == 16384 ? size_t(40) : p.N == 32768 ? size_t(80) : size_t(160)) * 1024;    // (the kernels have no static LDS)
        if (!ok || (sFloats + extra + (p.realMono ? 2 * 16 : 0)) * 4 > budget) { p.realSplit = false; p.realMono = false; p.recsReal.clear(); p.realLowPixels.clear(); p.realLowCount[0] = p.realLowCount[1] = 0; }
    }
    if (p.realSplit || p.realMono) {
        const double kTwoPi = 6.28318530717958647692;
        const uint32_t M = p.N / 2, R1 = M / 1024, T = R1 * 32;
        const int rows = 3 + int(R1) / 4 - 1;
        auto mult = [&](int row) { return row < 3 ? row + 1 : 4 * (row - 3 + 1); };
        p.twReal1.resize(size_t(rows) * 1024 * 2);
        for (int row = 0; row < rows; ++row)
            for (uint32_t c = 0; c < 1024; ++c) {
                const uint64_t m = (uint64_t(c) * uint64_t(mult(row))) % M;
                const double ang = -kTwoPi * double(m) / double(M);
                p.twReal1[(size_t(row) * 1024 + c) * 2 + 0] = float(std::cos(ang));
                p.twReal1[(size_t(row) * 1024 + c) * 2 + 1] = float(std::sin(ang));
            }
        p.windowHalf.resize(p.window.size());
        for (size_t i = 0; i < p.window.size(); ++i) p.windowHalf[i] = 0.5f * p.window[i];
        if (cosWindow) {
            p.winPhase.resize(1024 * 4);
            for (uint32_t c = 0; c < 1024; ++c)
                for (int e = 0; e < 2; ++e) {
                    const double ang = kTwoPi * double(2 * c + e) / double(p.N);
                    // p1 x (cos even, cos odd, sin even, sin odd): two aligned pairs, the window's cosine coefficient folded in
                    // (halved: these kernels transform x w / 2, see real_common.hpp realBinMag -- exact, the factor is a power of two)
                    p.winPhase[size_t(c) * 4 + e] = 0.5f * float(double(p.winP1) * std::cos(ang));
                    p.winPhase[size_t(c) * 4 + 2 + e] = 0.5f * float(double(p.winP1) * std::sin(ang));
                }
        }
        if (p.tw2.empty()) {                                   // N = 16384 has no R^3 tables of its own: the channel transform's passes 2 / 3 are radix 32
            const int R = 32, rows2 = 3 + R / 4 - 1;
            p.tw2.resize(size_t(rows2) * R * 2);
            for (int row = 0; row < rows2; ++row)
                for (int t2 = 0; t2 < R; ++t2) {
                    const uint32_t m = uint32_t(t2 * mult(row)) % 1024u;
                    const double ang = -kTwoPi * double(m) / 1024.0;
                    p.tw2[(size_t(row) * R + t2) * 2 + 0] = float(std::cos(ang));
                    p.tw2[(size_t(row) * R + t2) * 2 + 1] = float(std::sin(ang));
                }
        }
        p.tw2Full.clear();
        if (p.N >= 32768) {
            // [c < 32][34] float2: a thread's 32 factors W_1024^{c q} are contiguous (16 ds_read_b128), rows of 32 + 2 entries keep the 16 lanes
            // an LDS cycle serves on 16 different bank quads
            p.tw2Full.assign(size_t(32) * 34 * 2, 0.0f);
            for (uint32_t q = 0; q < 32; ++q)
                for (uint32_t c = 0; c < 32; ++c) {
                    const double ang = -kTwoPi * double((q * c) % 1024u) / 1024.0;
                    p.tw2Full[(size_t(c) * 34 + q) * 2 + 0] = float(std::cos(ang));
                    p.tw2Full[(size_t(c) * 34 + q) * 2 + 1] = float(std::sin(ang));
                }
        }
        p.twRealPost.resize(size_t(T) * 2);
        for (uint32_t kc = 0; kc < T; ++kc) {
            const double ang = -kTwoPi * double(kc) / double(p.N);
            p.twRealPost[size_t(kc) * 2 + 0] = float(std::cos(ang));
            p.twRealPost[size_t(kc) * 2 + 1] = float(std::sin(ang));
        }
        // The 1024-thread form of the N = 32768 channel transform (spectrum_real16.hip): M = 16 x 16 x (4 across a lane quad x 16).
        //   tw16: [16 q2][64 c_lo] W_1024^{c_lo q2} (pass 2), then [4 l][16 r] sign_l W_64^{r m}, m = brev2(l) (pass 3: the quad's
        //         radix-4 stage leaves lane l with sign_l a_m, signs (+, -, -, -): tools/emulate_real16.py), staged in LDS by every workgroup
        //   twPost16: [1024] W_N^{kb}, kb = q1 + 16 q2 + 256 m: the recombination twiddle of a thread's first bin
        p.tw16.clear(); p.twPost16.clear();
        if (p.N == 32768 && p.realSplit) {
            p.tw16.resize(size_t(16 * 64 + 4 * 16) * 2);
            for (uint32_t q = 0; q < 16; ++q)
                for (uint32_t c = 0; c < 64; ++c) {
                    const double ang = -kTwoPi * double((q * c) % 1024u) / 1024.0;
                    p.tw16[(size_t(q) * 64 + c) * 2 + 0] = float(std::cos(ang));
                    p.tw16[(size_t(q) * 64 + c) * 2 + 1] = float(std::sin(ang));
                }
            for (uint32_t l = 0; l < 4; ++l) {
                const uint32_t m = ((l & 1u) << 1) | (l >> 1);
                const double sign = l == 0 ? 1.0 : -1.0;
                for (uint32_t r = 0; r < 16; ++r) {
                    const double ang = -kTwoPi * double((r * m) % 64u) / 64.0;
                    p.tw16[(size_t(16 * 64) + size_t(l) * 16 + r) * 2 + 0] = float(sign * std::cos(ang));
                    p.tw16[(size_t(16 * 64) + size_t(l) * 16 + r) * 2 + 1] = float(sign * std::sin(ang));
                }
            }
            p.twPost16.resize(size_t(1024) * 2);
            for (uint32_t kb = 0; kb < 1024; ++kb) {
                const double ang = -kTwoPi * double(kb) / double(p.N);
                p.twPost16[size_t(kb) * 2 + 0] = float(std::cos(ang));
                p.twPost16[size_t(kb) * 2 + 1] = float(std::sin(ang));
            }
        }
    }
    return SGZ_OK;
}

}  // namespace sgz
