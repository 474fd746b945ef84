"""Randomised sweep of the Oscilloscope / Vectorscope kernels against the oracle: Lanczos-10 resampler over random views (window length, zoom,
rendering scale, width, ring length) and the zero-crossing trigger over random modes, thresholds and block splits.
usage: fuzz_scope.py [count] [seed]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, synth
from oracle import pyoracle as po

def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    po.build()
    rng = np.random.default_rng(seed)
    L = api.lib()
    stream = torch.cuda.current_stream().cuda_stream
    bad = 0
    for it in range(count):
        # ---- Lanczos view
        W = float(rng.integers(16, 40000)) if rng.random() < 0.7 else float(rng.uniform(16, 5000))
        left = float(rng.choice([0.0, 0.0, rng.uniform(0, 0.7)]))
        right = float(rng.choice([1.0, 1.0, min(1.0, left + rng.uniform(0.05, 1.0))]))
        scale = float(rng.choice([1.0, 2.0, 4.0, 8.0, rng.uniform(0.5, 8.0)]))
        width = int(rng.integers(64, 4000))
        nch = int(rng.integers(1, 4))
        ringlen = int(np.ceil(W)) + int(rng.integers(0, 64))
        ring = synth.gen(700 + it, 192000, ringlen, nch)
        vo = po.ScopeView(window_size=W, left=left, right=right, rendering_scale=scale, width=width)
        vg = api.ScopeView(window_size=W, left=left, right=right, rendering_scale=scale, width=width)
        npts = L.sgz_scope_num_points(C.byref(vg))
        ok = npts == po.lib().sgzo_scope_num_points(C.byref(vo))
        ey = ex = 0.0
        if ok and npts > 0:
            d = torch.from_numpy(ring).cuda()
            out = torch.zeros((nch, npts, 2), dtype=torch.float32, device="cuda")
            st = L.sgz_scope_lanczos_device(C.byref(vg), d.data_ptr(), ring.shape[1], d.stride(0), nch, out.data_ptr(), stream)
            if st < 0:
                print(it, "lanczos status", st, api.lib().sgz_last_error()); ok = False
            else:
                got = out.cpu().numpy()
                for c in range(nch):
                    x, y = po.scope_lanczos(vo, ring[c])
                    ey = max(ey, float(np.abs(got[c, :, 1] - y).max())); ex = max(ex, float(np.abs(got[c, :, 0] - x).max()))
                ok = ey <= 2e-6 and ex <= 1e-6
        print(it, "ok " if ok else "BAD", "lanczos W", W, "left", round(left, 3), "right", round(right, 3), "scale", round(scale, 3), "width", width,
              "ring", ringlen, "points", npts, "ey", ey, "ex", ex)
        bad += 0 if ok else 1
        # ---- zero crossing
        mode = int(rng.integers(0, 6))
        thr = float(rng.choice([0.0, 0.01, 0.3, rng.uniform(0, 1.5)]))
        n = int(rng.integers(1000, 200000))
        xs = synth.gen(900 + it, 96000, n, 2)
        if rng.random() < 0.5:
            z0 = int(rng.integers(0, n - 100)); xs[:, z0:z0 + int(rng.integers(1, 100))] = 0.0
        a, b = (xs[1] if mode == 1 else xs[0]), xs[1]
        st_o = po.ZeroCrossingState(state=0.0, threshold=thr, steady_clock=int(rng.integers(0, 5000)), cross_origin=0, count=0, armed=0)
        st_g = api.ZeroCrossingState(state=0.0, threshold=thr, steady_clock=st_o.steady_clock, cross_origin=0, count=0, armed=0)
        da, db = torch.from_numpy(a.copy()).cuda(), torch.from_numpy(b.copy()).cuda()
        dtrig = torch.zeros(1 << 17, dtype=torch.int64, device="cuda")
        pos, okz = 0, True
        while pos < n and okz:
            blk = int(min(n - pos, rng.choice([1, 7, 64, 1000, 4096, 30000, 100000])))
            want = po.zero_crossing(st_o, mode, a[pos:pos + blk], b[pos:pos + blk])
            cnt = C.c_size_t(0)
            api.check(L.sgz_scope_zero_crossing_device(C.byref(st_g), mode, da.data_ptr() + 4 * pos, db.data_ptr() + 4 * pos, blk,
                                                       dtrig.data_ptr(), dtrig.numel(), C.byref(cnt), stream))
            got = dtrig[:cnt.value].cpu().numpy().astype(np.uint64)
            okz = cnt.value == want.size and np.array_equal(got, want) and (st_g.armed != 0) == (st_o.armed != 0) and \
                st_g.count == st_o.count and st_g.state == st_o.state and st_g.cross_origin == st_o.cross_origin
            pos += blk
        print(it, "ok " if okz else "BAD", "zero-crossing mode", mode, "thr", round(thr, 4), "n", n)
        bad += 0 if okz else 1
    # ---- Vectorscope: polar transform, one-pole filters, peak envelope (random lengths, including SIMD tails and tiny inputs)
    for it in range(count):
        pairs = int(rng.integers(1, 5))
        n = int(rng.choice([1, 7, 8, 9, 63, 64, 65, 1000, rng.integers(10, 30000)]))
        x = synth.gen(1300 + it, 96000, max(n, 16), 2 * pairs)[:, :n].copy()
        if n > 4 and rng.random() < 0.5: x[:, int(rng.integers(0, n))] = 0.0
        d = torch.from_numpy(x).cuda()
        out = torch.zeros((pairs, n, 3), dtype=torch.float32, device="cuda")
        api.check(L.sgz_vector_polar_device(d.data_ptr(), d.stride(0), pairs, n, 8, out.data_ptr(), stream))
        got = out.cpu().numpy()
        okp = True
        for p_ in range(pairs):
            want = po.vector_polar(x[2 * p_], x[2 * p_ + 1])
            okp = okp and np.abs(got[p_, :, :2] - want[:, :2]).max() <= 2e-6 and np.abs(got[p_, :, 2] - want[:, 2]).max() <= 1e-6
        fo, fg = po.VectorFilters(), api.VectorFilters()
        env = float(np.exp(-1.0 / (rng.uniform(0.01, 1.0) * 96000))); ste = float(np.exp(-1.0 / (rng.uniform(0.01, 1.0) * 96000)))
        dl, dr = torch.from_numpy(x[0].copy()).cuda(), torch.from_numpy(x[1].copy()).cuda()
        oka = True
        for rep in range(2):
            g_o = po.vector_audio_processing(fo, x[0], x[1], env, ste)
            g = C.c_float(float("nan"))
            api.check(L.sgz_vector_audio_processing_device(C.byref(fg), dl.data_ptr(), dr.data_ptr(), n, 8, env, ste, 0.25, 1, C.byref(g), stream))
            oka = oka and [fg.env[0], fg.env[1]] == [fo.env[0], fo.env[1]] and all(fg.balance[i][j] == fo.balance[i][j] for i in range(2) for j in range(2)) \
                and abs(fg.phase[0] - fo.phase[0]) <= 1e-5 and abs(fg.phase[1] - fo.phase[1]) <= 1e-5 and (g.value == g_o or (np.isnan(g.value) and np.isnan(g_o)))
        env_o = rng.uniform(0, 1, 2 * pairs); env_g = env_o.copy()
        coeff = float(rng.uniform(0.9, 0.9999))
        want = po.peak_filter(x, coeff, env_o)
        gain = C.c_double(0)
        api.check(L.sgz_peak_filter_device(d.data_ptr(), d.stride(0), 2 * pairs, n, 8, coeff, env_g.ctypes.data_as(C.c_void_p), C.byref(gain), stream))
        okk = np.array_equal(env_g, env_o) and (gain.value == want or (np.isnan(gain.value) and np.isnan(want)))
        print(it, "ok " if okp and oka and okk else "BAD", "vector pairs", pairs, "n", n, "polar", okp, "filters", oka, "peak", okk)
        bad += 0 if okp and oka and okk else 1
    print("bad:", bad, "of", 3 * count)
    sys.exit(1 if bad else 0)

main()
