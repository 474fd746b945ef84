"""CPU tests of the product's host logic and of the C-ABI surface (no compute calls: no GPU here).
The host tables of libsgz.so (window, mapped frequencies, slope map, colour ratios/tables) are built by the
product's own C++ (signalizer_amd/csrc/plan.cpp) and must equal the oracle's bit for bit."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from signalizer_amd import api, config, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cases():
    yield "cfg1", config.cfg1()
    yield "cfg2", config.cfg2()
    yield "cfg5", config.cfg5()
    yield "linear_kaiser", config.spectrum_config(view_scaling=config.VIEW_LINEAR, window_type=config.WIN_KAISER,
                                                  window_beta=9.5, window_symmetry=config.WIN_SYMMETRIC, window_size=3000,
                                                  hop=700, num_pairs=3, channel_mode=config.CH_LEFT, bin_interp=config.INTERP_LINEAR)
    yield "complex_log", config.spectrum_config(channel_mode=config.CH_COMPLEX, axis_points=999, window_type=config.WIN_FLATTOP)
    yield "complex_lin", config.spectrum_config(channel_mode=config.CH_COMPLEX, view_scaling=config.VIEW_LINEAR, axis_points=512,
                                                window_size=1024, hop=256)
    yield "zoom_slope", config.spectrum_config(view_left=0.3, view_right=0.65, slope_a=0.5, slope_b=0.01, axis_points=777,
                                               window_type=config.WIN_GAUSSIAN, window_alpha=0.3, num_pairs=5,
                                               colours=[(12, 34, 56), (200, 10, 90), (1, 2, 3), (255, 255, 255), (90, 180, 45), (250, 128, 7)],
                                               ratios=(0.0, 1.0, 0.5, 0.00001, 0.3))
    for w in range(13):
        yield f"win{w}", config.spectrum_config(window_type=w, window_size=500, hop=100, window_alpha=0.35, window_beta=6.0,
                                                window_symmetry=w % 2)


@pytest.mark.parametrize("name,cfg", list(_cases()))
def test_plan_tables_equal_oracle(oracle, name, cfg):
    po = oracle
    plan = api.Plan(cfg)
    p = po.params_from_dict(cfg)
    w, scale = po.window(p.window_type, p.window_symmetry, p.window_size, p.window_alpha, p.window_beta)
    assert plan.N == po.lib().sgzo_transform_size(p.window_size)
    win = plan.window()
    assert np.array_equal(win[:p.window_size].view(np.uint32), w.view(np.uint32)) and (win[p.window_size:] == 0).all()
    assert plan.window_scale == scale
    mf = po.remap_frequencies(p)
    assert np.array_equal(plan.mapped_frequencies().view(np.uint32), mf.view(np.uint32))
    assert np.array_equal(plan.slope_map().view(np.uint32), po.slope_map(p, mf).view(np.uint32))
    assert np.array_equal(plan.colour_ratios().view(np.uint32), po.colour_ratios(cfg["ratios"]).view(np.uint32))
    for pair in range(p.num_pairs):
        assert np.array_equal(plan.colour_table(pair), po.colour_table(p, pair))
    plan.close()


def test_frame_count_matches_reference_cadence():
    assert api.lib().sgz_num_frames(2880000, 32768, 8192) == 348
    assert api.lib().sgz_num_frames(5760000, 65536, 16384) == 348
    assert api.lib().sgz_num_frames(4095, 4096, 1) == 0


def test_break_pixel_is_where_pixel_bandwidth_exceeds_bin_bandwidth():
    plan = api.Plan(config.cfg2())
    mf = plan.mapped_frequencies()
    b = plan.break_pixel
    bw = (mf[1:] - mf[:-1]) / np.float32(24000.0)
    assert 0 < b < plan.P - 1
    assert bw[b].astype(np.float64) > 1.0 / 16384 and (bw[:b].astype(np.float64) <= 1.0 / 16384).all()


def test_config_validation_mirrors_reference_assertions():
    for bad in (dict(axis_points=1), dict(window_size=0), dict(num_pairs=0), dict(hop=0), dict(sample_rate=0.5),
                dict(channel_mode=9), dict(window_type=99), dict(bin_interp=3)):
        with pytest.raises(api.SgzError) as e:
            api.Plan(config.spectrum_config(**bad))
        assert e.value.status == api.SGZ_EINVAL
    api.Plan(config.spectrum_config(channel_mode=config.CH_PHASE))     # every channel mode has host tables


def test_path_selection():
    """which K_A implementation a configuration selects"""
    P = lambda **kw: api.Plan(config.spectrum_config(**kw)).path
    assert P(window_size=4096, hop=1024) == 1 and P(window_size=32768, hop=8192) == 1 | 8   # cfg2: channel-split workgroups (whole-frame kernel behind them)
    assert P(window_size=32768, hop=8192, channel_mode=config.CH_MIDSIDE) == 1 | 8 and P(window_size=32768, hop=8192, channel_mode=config.CH_COMPLEX) == 1
    assert P(window_size=3000, hop=750) == 1                                  # zero-padded to 4096
    assert P(window_size=65536, hop=16384, sample_rate=96000.0) == 2 | 4 | 8    # cfg5: channel-split workgroups (halves + per-side LDS map behind them)
    assert P(window_size=8192, hop=2048, channel_mode=config.CH_MERGE) == 2 | 4
    assert P(window_size=8192, hop=2048, channel_mode=config.CH_COMPLEX) == 2   # whole-spectrum view: generic map kernel
    assert P(window_size=16384, hop=4096) == 0 | 4 | 8                       # Separate: channel-split workgroups (generic passes behind them)
    # mono at the default view: its lowest pixels' tap windows wrap below bin 0 -- redone from the kernel's complex entries, eligible
    assert P(window_size=16384, hop=4096, channel_mode=config.CH_MERGE) == 0 | 4 | 8 and P(window_size=20, hop=7, axis_points=16) == 0 | 4
    for W in (4096, 8192, 32768):                                              # Phase keeps complex bins: generic at any size
        assert P(window_size=W, hop=W // 4, channel_mode=config.CH_PHASE) == 0


def test_complex_mode_dc_pixel_list(oracle):
    """the pixels of a Complex view whose csp is complex in the oracle (csf[0] stays complex, TransformDSP.inl:993) are
    exactly the ones the plan lists for the complex redo"""
    po = oracle
    for W, interp in ((4096, config.INTERP_LANCZOS), (2048, config.INTERP_LINEAR), (8192, config.INTERP_LANCZOS), (4096, config.INTERP_NONE)):
        cfg = config.spectrum_config(sample_rate=96000.0, window_size=W, hop=W // 4, axis_points=300, channel_mode=config.CH_COMPLEX,
                                     bin_interp=interp)
        x = synth.gen(41, 96000, W, 2) + 0.25
        x[1] -= 0.6
        r = po.spectrogram(po.params_from_dict(cfg), x.astype(np.float32), want_mapped=True)
        complex_px = set(np.nonzero(r["mapped"][0, 0, :300].imag != 0)[0].tolist())
        listed = set(api.Plan(cfg).dc_pixels().tolist())
        assert complex_px <= listed, (W, interp, sorted(complex_px - listed))
        # listed but real in the oracle: only a tap weight of exactly zero can do that
        assert len(listed - complex_px) <= 2, (W, interp, sorted(listed - complex_px))
    assert api.Plan(config.spectrum_config(window_size=4096, hop=1024)).dc_pixels().size == 0


def test_library_exports_every_symbol_the_header_declares():
    hdr = open(os.path.join(ROOT, "include", "sgz.h")).read()
    declared = sorted(set(re.findall(r"\b(sgz_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    L = api.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(api.EXPORTS) == declared
    m = re.search(r"#define SGZ_ABI_VERSION (\d+)", hdr)
    assert m and L.sgz_abi_version() == int(m.group(1)) == 5          # the binary is the header's (5: scope / vector handle options, line-results tracker)


def test_header_is_plain_c(tmp_path):
    """include/sgz.h is the drop-in boundary: it must compile as C99 and as C++11 on its own (no torch / HIP types in the signatures)"""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text('#include "sgz.h"\nint main(void) { sgz_spectrum_config c; sgz_scope_config s; sgz_vector_config v; (void)c; (void)s; (void)v; return 0; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror"], ["g++", "-std=c++11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-x", "c++"]):
        r = subprocess.run(cmd + ["-I", inc, "-fsyntax-only", str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_struct_layout_matches_header():
    """ctypes mirrors must have the C layout (the parity tests pass these structs across the ABI)"""
    assert C.sizeof(api.SpectrumConfig) == 4 * 10 + 8 * 10 + 8 + 18 + 2 + 40 + 4 + 8 + 8   # incl. align padding; algorithm, free_q; display_mode, _reserved
    # ... and against the compiler's own view of include/sgz.h
    import shutil
    import subprocess
    import tempfile
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc:
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "layout.c")
            open(src, "w").write('#include <stdio.h>\n#include <stddef.h>\n#include "sgz.h"\nint main(void) { printf("%zu %zu %zu %zu %zu", '
                                 'sizeof(sgz_spectrum_config), offsetof(sgz_spectrum_config, display_mode), offsetof(sgz_spectrum_config, ratios), '
                                 'sizeof(sgz_scope_config), sizeof(sgz_vector_config)); return 0; }\n')
            exe = os.path.join(d, "layout")
            subprocess.check_call([cc, "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
            got = [int(v) for v in subprocess.check_output([exe]).split()]
        assert got == [C.sizeof(api.SpectrumConfig), api.SpectrumConfig.display_mode.offset, api.SpectrumConfig.ratios.offset,
                       C.sizeof(api.ScopeConfig), C.sizeof(api.VectorConfig)], got
    assert api.SpectrumConfig.ratios.offset % 8 == 0 and api.SpectrumConfig.window_alpha.offset == 40
    assert C.sizeof(api.ScopeView) == 40 and C.sizeof(api.ZeroCrossingState) == 48 and C.sizeof(api.VectorFilters) == 32


@pytest.mark.skipif(api.lib().sgz_device_count() > 0, reason="only meaningful without a GPU")
def test_no_cpu_fallback_compute_fails_loudly_without_gpu():
    plan = api.Plan(config.cfg1())
    with pytest.raises(api.SgzError) as e:
        plan.upload()
    assert e.value.status == api.SGZ_EHIP
    x = np.zeros((2, 8192), np.float32)
    with pytest.raises(api.SgzError) as e:
        api.render_spectrogram(config.cfg1(), x)
    assert e.value.status == api.SGZ_EHIP
    h = C.c_void_p()
    c = api.config_from_dict(config.cfg1())
    assert api.lib().sgz_spectrum_create(C.byref(c), C.byref(h)) == api.SGZ_EHIP


def test_mono_views_on_the_real_input_kernel():
    """which mono views the real-input kernel takes (plan.cpp realMono): tap windows may wrap below bin 0 or reach csf[N/2 .. N/2+7] -- the
    kernel keeps those 16 complex entries --, arg-max runs must stay inside csf[0 .. N/2]"""
    P = lambda **kw: api.Plan(config.spectrum_config(**kw)).path
    mono = dict(window_size=32768, hop=8192, channel_mode=config.CH_MERGE)
    assert P(**mono) & 8                                                                   # default log view
    assert P(**mono, view_scaling=0, view_left=0.0, view_right=1.0) & 8                    # linear from 0 Hz to Nyquist
    assert P(**mono, view_scaling=0, view_left=0.0, view_right=0.001, bin_interp=2) & 8    # deep zoom at 0 Hz: every window wraps
    assert P(window_size=16384, hop=4096, channel_mode=config.CH_LEFT) & 8                 # default view at N = 16384: its lowest windows wrap
    assert not P(window_size=30000, hop=7500, channel_mode=config.CH_MERGE) & 8            # zero-padded window
    assert not P(window_size=8192, hop=2048, channel_mode=config.CH_MERGE) & 8             # no real-input kernel at this size
