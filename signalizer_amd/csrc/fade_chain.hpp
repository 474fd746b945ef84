// fade_chain.hpp -- the Vectorscope's fade ramp without its dependent additions.
//
// The reference's ramp is a running fp32 sum per SIMD lane, f <- fl(f + incr) once per SIMD iteration (VectorscopeRendering.cpp:528-543,
// :592): 1 200 dependent additions at cfg4, 18-20 us on a lone wave whatever is done around them.  The sum is sequential only in
// appearance.  While f stays inside one binade [2^e, 2^(e+1)) it is an integer multiple m of u = 2^(e-23), and
//     fl(m u + incr) = (m + q) u,   q = round_to_nearest(incr / u),
// the SAME q at every step unless incr / u ends in exactly .5 (ties go to even: then q depends on m's parity).  So a chain is a
// handful of arithmetic progressions of integers -- one per binade it crosses, about fourteen from 1 / size to 1 -- joined by single
// real additions where the binade changes, where f is zero or tiny, or where a tie makes the step depend on m.  One thread per lane
// finds the progressions (a few dozen operations each); every thread then evaluates its outputs directly: (m + j q) u, exact (an
// integer below 2^24 times a power of two).  Bit for bit the reference's sums: tests/test_gpu_vector_stream.py, test_gpu_scope_vector.py.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace sgz {

struct FadeSeg { uint32_t k0, count, m, q; float u; };        // outputs k0 .. k0 + count - 1: (m + j q) u;  u == 0: the single value in `m`'s bits
constexpr int kFadeSegs = 48;                                  // per chain; a chain that needs more (ties all the way) is finished step by step

// the progressions of  value[k] = f_k, f_{k+1} = fl(f_k + incr), k < steps, from f_0 = f (in / out: f_steps, what the next step would
// write); `last`: the last value written (untouched when steps == 0).  segs[0 .. n) in order, covering [0, steps).  false: the table
// is full (a chain with ties at every step) -- the caller walks the chain the reference's way.
__device__ inline bool fadeChainSegments(float &f, float incr, uint32_t steps, FadeSeg *segs, int &n, float &last)
{
    uint32_t k = 0;
    n = 0;
    while (k < steps) {
        if (n >= kFadeSegs) return false;
        const uint32_t bits = __float_as_uint(f);
        const int e = int((bits >> 23) & 0xffu) - 127;
        bool single = !(f > 0.f) || e - 23 < -126 || e >= 127 || !(incr > 0.f);
        uint32_t m = 0; uint64_t q = 0; float u = 0.f;
        if (!single) {
            u = __uint_as_float(uint32_t(e - 23 + 127) << 23);
            m = (bits & 0x7fffffu) | 0x800000u;
            const double t = double(incr) * double(__uint_as_float(uint32_t(127 - (e - 23)) << 23));      // incr / u, exact: u is a power of two
            const double a = floor(t), r = t - a;
            if (r == 0.5 || a >= 16777216.0) single = true;    // a tie (the step depends on m's parity), or a jump across binades at once
            else q = uint64_t(a) + (r > 0.5 ? 1u : 0u);
        }
        if (single) {                                          // one real addition
            segs[n++] = FadeSeg{k, 1u, bits, 0u, 0.f};
            last = f; f = f + incr; ++k;
            continue;
        }
        const uint32_t room = q ? uint32_t((0xffffffu - m) / q) : 0xffffffffu;         // further steps that stay inside the binade
        const uint32_t count = (steps - k - 1u < room) ? steps - k : room + 1u;
        segs[n++] = FadeSeg{k, count, m, uint32_t(q), u};
        const float fl = float(m + uint32_t(q) * (count - 1u)) * u;
        last = fl; f = fl + incr; k += count;
    }
    return true;
}

__device__ inline float fadeChainValueAt(const FadeSeg *segs, int n, uint32_t k)
{
    int s = 0;
    while (s + 1 < n && segs[s + 1].k0 <= k) ++s;
    const FadeSeg g = segs[s];
    return g.u == 0.f ? __uint_as_float(g.m) : float(g.m + g.q * (k - g.k0)) * g.u;
}

// every thread's share of the outputs of `lanes` chains of `steps` values each: threads tid % lanes == lane take consecutive runs of
// steps of their lane's chain (the segment is found once per run), out[k * lanes + lane] = value + bias.  32-bit arithmetic only.
__device__ inline void fadeChainFill(const FadeSeg *segs /*[lanes][kFadeSegs]*/, const int *nseg /*[lanes]*/, uint32_t lanes, uint32_t steps, uint32_t tid,
                                     uint32_t threads, float bias, float *out)
{
    const uint32_t groups = threads / lanes;
    if (!steps || tid >= groups * lanes) return;
    const uint32_t lane = tid % lanes, c = tid / lanes;
    const uint32_t per = (steps + groups - 1u) / groups;
    const uint32_t k0 = c * per, k1 = k0 + per < steps ? k0 + per : steps;
    const FadeSeg *sg = segs + size_t(lane) * kFadeSegs;
    const int n = nseg[lane];
    int s = 0;
    for (uint32_t k = k0; k < k1; ++k) {
        while (s + 1 < n && sg[s + 1].k0 <= k) ++s;
        const FadeSeg g = sg[s];
        const float v = g.u == 0.f ? __uint_as_float(g.m) : float(g.m + g.q * (k - g.k0)) * g.u;
        out[size_t(k) * lanes + lane] = v + bias;
    }
}

}  // namespace sgz
