// rt_driver.cpp -- the host loop of bench.py's Oscilloscope / Vectorscope workloads in the reference's host language.
//
// A plugin drives the real-time handles from C++ (audio callbacks: Oscilloscope.cpp:333-392 / Vectorscope.cpp:264-300; the paint:
// OscilloscopeRendering.cpp / VectorscopeRendering.cpp); bench.py is Python, and a rendered frame's seven to nine C-ABI calls cost it
// 60-100 us of interpreter and ctypes time on top of what the library spends.  This is the same loop -- per rendered frame the audio
// thread's callbacks from host buffers, then the peak filter and every channel's / pair's vertex stream into the caller's pinned
// buffers -- compiled against include/sgz.h and nothing else.  Not part of the product: built into signalizer_amd/librtdriver.so by
// build.py, loaded by bench.py only.
#include <chrono>
#include <cstdint>
#include <vector>

#include "../include/sgz.h"

namespace {
double seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}

extern "C" {

// x: planar [channels][total]; a rendered frame takes `perFrame` samples in callbacks of `block`; the signal is walked in laps of
// `laps` frames.  Returns seconds for `frames` frames after `warm` untimed ones, < 0 on a failing call (the status, negated - 1000).
double sgz_bench_scope_loop(sgz_scope *h, const float *x, size_t total, uint32_t channels, uint32_t perFrame, uint32_t block, uint32_t laps,
                            const sgz_scope_view *view, uint32_t items, const uint32_t *evaluators, const uint32_t *chans, float *const *xyz,
                            uint8_t *const *rgba, uint32_t capacity, double deltaTime, uint32_t lanes, int warm, int frames, uint64_t *refused,
                            uint32_t *vertices)
{
    std::vector<const float *> ptr(channels);
    std::vector<uint32_t> counts(items);
    uint64_t busy = 0;
    uint32_t frame = 0;
    double t0 = 0;
    for (int it = -warm; it < frames; ++it) {
        if (it == 0) { if (sgz_scope_flush(h) != SGZ_OK) return -1001; t0 = seconds(); }
        const size_t a = size_t(frame % laps) * perFrame;
        for (size_t pos = a; pos < a + perFrame; pos += block) {
            const uint32_t n = uint32_t(pos + block <= a + perFrame ? block : a + perFrame - pos);
            for (uint32_t c = 0; c < channels; ++c) ptr[c] = x + size_t(c) * total + pos;
            sgz_status st;
            while ((st = sgz_scope_push(h, ptr.data(), channels, n)) == SGZ_BUSY) ++busy;      // never waits; a refused block is offered again
            if (st != SGZ_OK) return -1000.0 - double(st);
        }
        if (sgz_status st = sgz_scope_peak_filter(h, deltaTime, lanes, nullptr); st != SGZ_OK) return -1000.0 - double(st);
        for (uint32_t k = 0; k < items; ++k) counts[k] = capacity;
        if (sgz_status st = sgz_scope_vertices_all(h, view, items, evaluators, chans, xyz, rgba, counts.data()); st != SGZ_OK) return -1000.0 - double(st);
        ++frame;
    }
    const double dt = seconds() - t0;
    if (refused) *refused = busy;
    if (vertices) { uint32_t v = 0; for (uint32_t c : counts) v += c; *vertices = v; }
    return dt;
}

double sgz_bench_vector_loop(sgz_vector *h, const float *x, size_t total, uint32_t channels, uint32_t perFrame, uint32_t block, uint32_t laps,
                             float *xyz, float *rgb, uint32_t capacity, double deltaTime, int warm, int frames, uint64_t *refused, uint32_t *vertices)
{
    std::vector<const float *> ptr(channels);
    uint64_t busy = 0;
    uint32_t frame = 0, count = 0;
    double t0 = 0;
    for (int it = -warm; it < frames; ++it) {
        if (it == 0) { if (sgz_vector_flush(h) != SGZ_OK) return -1001; t0 = seconds(); }
        const size_t a = size_t(frame % laps) * perFrame;
        for (size_t pos = a; pos < a + perFrame; pos += block) {
            const uint32_t n = uint32_t(pos + block <= a + perFrame ? block : a + perFrame - pos);
            for (uint32_t c = 0; c < channels; ++c) ptr[c] = x + size_t(c) * total + pos;
            sgz_status st;
            while ((st = sgz_vector_push(h, ptr.data(), channels, n)) == SGZ_BUSY) ++busy;
            if (st != SGZ_OK) return -1000.0 - double(st);
        }
        if (sgz_status st = sgz_vector_peak_filter(h, deltaTime, nullptr); st != SGZ_OK) return -1000.0 - double(st);
        count = capacity;
        if (sgz_status st = sgz_vector_vertices_all(h, xyz, rgb, &count); st != SGZ_OK) return -1000.0 - double(st);
        ++frame;
    }
    const double dt = seconds() - t0;
    if (refused) *refused = busy;
    if (vertices) *vertices = count * (channels / 2);
    return dt;
}

}  // extern "C"
