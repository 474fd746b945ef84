"""Known answers for the oracle's restatement of the Oscilloscope's spectral trigger and frequency colouring (oracle/scope_spectral.c;
OscilloscopeDSP.inl:62-308, :445-647).  The cpl / DustFFT pieces are absent from the reference tree, so these tests pin the restatement
against what the reference's own formulas must produce on signals with closed-form answers: a tone's fundamental, the 5 Hz start-up
quirk of the median filter, a display that starts on the sine's rising zero crossing wherever the ring happens to stand, -6 dB at a
Linkwitz-Riley crossover, a low tone painted in the low colour.  libstdc++'s std::nth_element is compared with the real one (g++)."""
import ctypes as C
import shutil
import subprocess

import numpy as np
import pytest

SR = 48000.0


def _tone_ring(po, f0, phase, newest_time, size):
    """ring memory with cursor 0 (oldest first): sample k is time newest_time - (size - 1 - k)"""
    t = newest_time - (size - 1 - np.arange(size))
    return np.sin(2 * np.pi * f0 * t / SR + phase).astype(np.float32)


def test_nth_element_is_libstdcxx(oracle, tmp_path):
    po = oracle
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    src = tmp_path / "nth.cpp"
    src.write_text("""
#include <algorithm>
#include <cstdint>
struct R { std::uint64_t index; double value, offset; };
extern "C" void real_nth(R* v, int n, int nth) {
    std::nth_element(v, v + nth, v + n, [](const R& a, const R& b) { return a.index < b.index; });
}
""")
    so = tmp_path / "libnth.so"
    subprocess.run([gxx, "-O2", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    real = C.CDLL(str(so))
    rng = np.random.default_rng(5)
    dt = np.dtype([("index", np.uint64), ("value", np.float64), ("offset", np.float64)])
    for trial in range(3000):
        r = np.zeros(8, dt)
        r["index"] = rng.integers(0, rng.integers(1, 12), 8)          # few distinct keys: ties decide which record lands in the middle
        r["value"] = rng.random(8)
        r["offset"] = np.arange(8)                                     # identifies the record
        want = r.copy()
        real.real_nth(want.ctypes.data_as(C.c_void_p), 8, 4)
        got = po.nth_element_by_index(r, 4)
        assert np.array_equal(got, want), (trial, r, got, want)


@pytest.mark.parametrize("f0", [110.0, 441.3, 1234.5])
def test_fundamental_of_a_tone_and_the_startup_quirk(oracle, f0):
    po = oracle
    size = 20000
    ts = po.SpectralState()
    out = []
    for frame in range(10):
        mem = _tone_ring(po, f0, 0.3, 50000 + 800 * frame, size)
        po.scope_analyse(ts, mem, mem, 0, 0, 2000.0, SR, 0.02, 0.0, 0.0)
        out.append((ts.fundamental, ts.cycle_samples))
    # medianTriggerFilter starts value-initialised: until four records are stored the median record is bin 0 and the fundamental is
    # clamped to 5 Hz (OscilloscopeDSP.inl:196-214)
    assert all(f == 5.0 for f, _ in out[:4])
    for f, c in out[4:]:
        assert abs(f - f0) < 0.02 * f0 + 0.5, (f, f0)
        assert c == SR / f


@pytest.mark.parametrize("f0,window", [(441.3, 2000.0), (97.0, 1500.5), (2000.0, 9000.0)])
def test_display_starts_on_the_rising_zero_crossing(oracle, f0, window):
    """drawWavePlot's Lanczos branch starts at samplePos = 2 cycleSamples + window - sampleOffset samples before the cursor
    (OscilloscopeRendering.cpp:810): with the phase correction of calculateTriggeringOffset that point must be the same point of
    the cycle whatever the ring's position -- and, "phase correct to sines", a rising zero crossing."""
    po = oracle
    size = 24000
    ts = po.SpectralState()
    phases = []
    for frame in range(14):
        newest = 60000 + 1237 * frame
        mem = _tone_ring(po, f0, 1.1, newest, size)
        po.scope_analyse(ts, mem, mem, 0, 0, window, SR, 0.02, 0.0, 0.0)
        if frame < 5:
            continue
        sample_pos = 2 * ts.cycle_samples + window - ts.sample_offset
        t_start = newest + 1 - sample_pos                      # "cursor - k" is k samples before the write position = newest + 1
        ph = (2 * np.pi * f0 * t_start / SR + 1.1) % (2 * np.pi)
        phases.append(ph if ph < np.pi else ph - 2 * np.pi)
    phases = np.array(phases)
    per_sample = 2 * np.pi * f0 / SR
    assert np.ptp(phases) < 0.25 * per_sample + 0.02, phases                      # stable from frame to frame
    assert np.abs(phases).max() < 0.25 * per_sample + 0.03, phases                # and on the zero crossing


def test_linkwitz_riley_bands(oracle):
    po = oracle
    n = 24000
    t = np.arange(n) / SR
    for f, band in [(60.0, 0), (1000.0, 1), (9000.0, 2)]:
        y = po.lr_bands(np.sin(2 * np.pi * f * t), SR)[n // 2:]
        e = (y.astype(np.float64) ** 2).mean(0)
        assert e.argmax() == band and e[band] > 20 * np.delete(e, band).max(), (f, e)
    # LR4: both branches are 6 dB down at the crossover and the branches sum to an all-pass
    y = po.lr_bands(np.sin(2 * np.pi * 300.0 * t), SR)[n // 2:].astype(np.float64)
    amp = np.sqrt(2 * (y ** 2).mean(0))
    assert abs(amp[0] - 0.5) < 0.01
    y = po.lr_bands(np.sin(2 * np.pi * 3000.0 * t), SR)[n // 2:].astype(np.float64)
    amp = np.sqrt(2 * (y ** 2).mean(0))
    assert abs(amp[1] - 0.5) < 0.02 and abs(amp[2] - 0.5) < 0.02


def test_colours_follow_the_band(oracle):
    po = oracle
    bands = [(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)]
    keys = [(10, 20, 30, 255), (200, 100, 50, 255)]
    n = 30000
    t = np.arange(n) / SR
    for f, want in [(60.0, 0), (1000.0, 1), (9000.0, 2)]:
        st = po.ScopeStream(2, SR, 4000.0, po.TRIG_NONE, 0.0, po.OSC_LEFT, 1.0, po.ENV_NONE, 0.3)
        st.enable_colours(bands, 1.0, 5.0, keys)                  # blend parameter 1: the key colour is blended out completely
        x = np.stack([np.sin(2 * np.pi * f * t), 0.5 * np.sin(2 * np.pi * f * t)]).astype(np.float32)
        for pos in range(0, n, 1000):
            st.audio(x[:, pos:pos + 1000])
        for c in (0, 1):
            for aux in (False, True):
                col, cur = st.front_colours(c, aux)
                px = col.view(np.uint8).reshape(-1, 4)
                assert (px[:, 3] == 255).all()
                # side of (x, 0.5 x) = 0.25 x, mid = 0.75 x: every signal is the same tone
                assert (px[:, want] >= 254).all() and (np.delete(px[:, :3], want, axis=1) <= 12).all(), (f, c, aux, px[:4])
    # blend parameter 0 (blend = 1): the pixel is the channel's key
    st = po.ScopeStream(2, SR, 4000.0, po.TRIG_NONE, 0.0, po.OSC_LEFT, 1.0, po.ENV_NONE, 0.3)
    st.enable_colours(bands, 0.0, 5.0, keys)
    x = np.random.default_rng(1).standard_normal((2, 5000)).astype(np.float32)
    st.audio(x)
    for c in (0, 1):
        for aux in (False, True):
            px = st.front_colours(c, aux)[0].view(np.uint8).reshape(-1, 4)
            assert (px == np.array(keys[c], np.uint8)).all()


def test_wave_plot_spectral_and_colours(oracle):
    """the vertex stream in Spectral mode reads cycleSamples / sampleOffset; colours: Linear = the sample's, Lanczos = lerp of the two
    newest kernel samples"""
    po = oracle
    size = 12000
    mem = _tone_ring(po, 500.0, 0.0, 40000, size)
    col = (np.arange(size, dtype=np.uint32) % 251) * 0x01010101
    v = po.ScopeView(1000.0, 0.0, 1.0, 1.0, 4001, 0)
    cyc, off = SR / 500.0, 17.25
    xyz, rgba = po.scope_wave_plot_ex(v, po.TRIG_SPECTRAL, 3, mem, mem, 0, 0, cyc, off, col)
    # 4 vertices per sample; vertex j sits at sample position samplePos0 + j / 4 behind the cursor
    sample_pos = 2 * cyc + 1000.0 - off
    j = np.arange(xyz.shape[0])
    t = 40001 - sample_pos + j * (1000.0 - 1) / 4000.0
    want = np.sin(2 * np.pi * 500.0 * t / SR)
    assert np.abs(xyz[:, 1] - want).max() < 2e-3                                  # Lanczos-10 of a 500 Hz tone at 48 kHz
    assert rgba is not None and rgba.shape == (xyz.shape[0], 4) and (rgba[:, 0] == rgba[:, 3]).all()
    # Linear: ceil(window) + ceil(cycleSamples) vertices from cursor - (ceil(window) + ceil(cycleSamples))
    v2 = po.ScopeView(1000.0, 0.0, 1.0, 1.0, 400, 0)
    xyz2, rgba2 = po.scope_wave_plot_ex(v2, po.TRIG_SPECTRAL, 2, mem, mem, 0, 0, cyc, off, col)
    n = 1000 + int(np.ceil(cyc))
    assert xyz2.shape[0] == n
    assert np.array_equal(xyz2[:, 1], mem[size - n:]) and np.array_equal(rgba2.view(np.uint32)[:, 0], col[size - n:])
