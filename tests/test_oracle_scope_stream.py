"""Known answers for the oracle's restatement of the Oscilloscope's audio-thread state machine (oracle/scope_stream.c): the
reference ships no tests, so the restatement is pinned against what the code it follows must do by construction."""
import numpy as np
import pytest

SR = 48000.0


def _sine(n, f=440.3, amp=0.8):
    t = np.arange(n) / SR
    return np.stack([amp * np.sin(2 * np.pi * f * t), 0.5 * np.sin(2 * np.pi * 100 * t)]).astype(np.float32)


def _feed(s, x, seed=0, max_block=700):
    rng = np.random.default_rng(seed)
    pos = 0
    while pos < x.shape[1]:
        n = int(rng.integers(1, max_block))
        s.audio(x[:, pos:pos + n])
        pos += n


def test_triggering_off_front_buffer_is_the_newest_window(oracle):
    """KA: TriggeringMode::None -> audioProcessing writes straight into the front ring (OscilloscopeDSP.inl:415-418): in time order
    the ring holds the newest ceil(W + 1) samples"""
    po = oracle
    x = _sine(5000)
    s = po.ScopeStream(2, SR, 480.0, trigger_mode=po.TRIG_NONE)
    _feed(s, x)
    assert s.size == 481
    for c in range(2):
        assert np.array_equal(s.front_in_time_order(c), x[c, -481:])


def test_threshold_above_the_peak_never_swaps(oracle):
    """KA11: threshold > peak => no trigger => the front buffer is never written"""
    po = oracle
    s = po.ScopeStream(2, SR, 480.0, threshold=2.0)
    _feed(s, _sine(20000))
    st = s.state()
    assert st["swaps"] == 0 and st["peaks"] == 0
    assert not s.front(0)[0].any()
    assert st["bufferedSamples"] == 481 and st["frontOrigin"] + st["bufferedSamples"] == st["steadyClock"] == 20000


def test_front_buffer_is_contiguous_audio_ending_after_the_trigger(oracle):
    """processMutating always hands the OLDEST buffered samples to the front ring (swapBuffers(cappedSize, -bufferedSamples)), so in
    time order the ring is one contiguous slice of the input ending at frontOrigin, and the last trigger lies inside it with at
    least half a window behind it"""
    po = oracle
    x = _sine(30000)
    s = po.ScopeStream(2, SR, 480.0, threshold=0.1)
    _feed(s, x, seed=3)
    st = s.state()
    assert st["swaps"] > 200
    end = st["frontOrigin"]
    for c in range(2):
        assert np.array_equal(s.front_in_time_order(c), x[c, end - 481:end])
    assert end - 481 <= st["oldPeak"] - 240 and st["oldPeak"] < end
    # the trigger is an upward zero crossing of channel 0 followed by a sample above the threshold
    k = st["oldPeak"]
    assert x[0, k] > 0 and x[0, k - 1] < 0


def test_trigger_indices_match_the_stateless_detector(oracle):
    """the stream's detector (executeSamplingWindows on the block, state carried in TriggeringProcessor) fires where the stand-alone
    ZeroCrossingProcessor restatement does: count of swaps + pending == number of triggers"""
    po = oracle
    x = _sine(12000, f=997.0)
    s = po.ScopeStream(2, SR, 256.0, threshold=0.3)
    _feed(s, x, seed=5)
    zs = po.ZeroCrossingState()
    zs.threshold = 0.3
    trig = po.zero_crossing(zs, po.OSC_LEFT, x[0])
    st = s.state()
    assert st["swaps"] + st["peaks"] == len(trig)


@pytest.mark.parametrize("mode", range(6))
def test_rms_envelope_known_answer(oracle, mode):
    """KA13: y_n = x^2 + a (y_{n-1} - x^2) on a constant input converges to x^2; envelopeGain = 1 / sqrt(max env)"""
    po = oracle
    x = np.full((2, 40000), 0.5, np.float32)
    x[1] = 0.25
    s = po.ScopeStream(2, SR, 480.0, trigger_mode=po.TRIG_NONE, osc_mode=mode, env_mode=po.ENV_RMS, envelope_window=0.01)
    _feed(s, x)
    expect = {0: 0.5, 1: 0.25, 2: 0.375, 3: 0.125, 4: 0.5, 5: np.sqrt(0.5 * 0.75 ** 2)}[mode]
    assert abs(1.0 / s.envelope_gain - expect) < 1e-4


def test_envelope_hold_fires_one_sample_after_a_rising_run_ends(oracle):
    """PeakHoldProcessor (StreamPreprocessing.h:282-310) on squares: a run of rising squares arms, the first sample that falls fires a
    trigger at the previous sample; the held level decays by 0.9999 per sample and never below threshold^2"""
    po = oracle
    sr, window = 48000.0, 64.0
    s = po.ScopeStream(2, sr, window, 3, 0.1, 0, 1.0, 0, 0.3)
    s.set_hysteresis(0.0)
    x = np.zeros((2, 4000), np.float32)
    x[0, 1000:1011] = np.linspace(0.2, 1.0, 11)            # rising to a peak at sample 1010, then silence
    x[0, 3000:3006] = np.linspace(0.3, 1.5, 6)             # a second burst above the held (slowly decaying) level, peak at 3005
    s.audio(x)
    st = s.state()
    assert st["swaps"] == 2                                # one window per burst ...
    assert st["currentPeak"] == 3005 and st["oldPeak"] == 3005   # ... placed on the last rising sample


def test_rectangular_and_none_vertex_lists(oracle):
    po = oracle
    mem = np.arange(10, dtype=np.float32) / 10
    v = po.ScopeView(8.0, 0.0, 1.0, 1.0, 800, 0)
    lin, _ = po.scope_wave_plot_ex2(v, 0, 2, mem, mem, 0, 3)
    dots, _ = po.scope_wave_plot_ex2(v, 0, 0, mem, mem, 0, 3)
    rect, col = po.scope_wave_plot_ex2(v, 0, 1, mem, mem, 0, 3, key=0x11223344)
    assert np.array_equal(lin, dots) and lin.shape[0] == 8
    assert rect.shape[0] == 16 and np.array_equal(rect[0::2, 1], lin[:, 1]) and np.array_equal(rect[1::2, 1], lin[:, 1])
    assert np.array_equal(rect[0::2, 0], lin[:, 0]) and np.array_equal(rect[1::2, 0], lin[:, 0] + 1)
    assert (col.view(np.uint32) == 0x11223344).all()
    # Window mode: the drawn window starts ceil(fmod(transport, window)) samples back
    a, _ = po.scope_wave_plot_ex2(v, 2, 2, mem, mem, 0, 3, transport_position=8 * 5 + 3)
    b, _ = po.scope_wave_plot_ex2(v, 2, 2, mem, mem, 0, 3, transport_position=3)
    assert np.array_equal(a, b) and a[0, 1] == mem[(3 - 3) % 10]
