"""Randomised sweep of the stateful paths: (i) the real-time per-block path (sgz_spectrum_push / pop_column, callbacks of varying
length) against the oracle's render of [W zeros ++ audio]; (ii) a render cut at random frame boundaries with the decay state
carried between the calls against the single-call render (must be identical).   usage: fuzz_realtime.py [count] [seed]"""
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from signalizer_amd import api, config, synth
from oracle import pyoracle as po
from fuzzcfg import random_config
from parity_chain import check_render

def pop_all(h, P, want, timeout=5.0):
    cols, t0 = [], time.time()
    buf, ap = np.zeros((P, 4), np.uint8), C.c_uint32(0)
    while len(cols) < want and time.time() - t0 < timeout:
        st = api.lib().sgz_spectrum_pop_column(h, buf.ctypes.data_as(C.c_void_p), C.byref(ap))
        if st == api.SGZ_OK: cols.append(buf.copy())
    return cols

def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    po.build()
    rng = np.random.default_rng(seed)
    bad = 0
    L = api.lib()
    for it in range(count):
        cfg = random_config(rng)
        cfg["window_size"] = int(min(cfg["window_size"], 20000))
        W = cfg["window_size"]
        cfg["hop"] = hop = max(16, int(W * rng.choice([0.25, 0.5, 1.0])))
        P, nch = cfg["axis_points"], 2 * cfg["num_pairs"]
        frames = int(rng.integers(2, 14))
        S = frames * hop + int(rng.integers(0, hop))
        x = synth.gen(500 + it, cfg["sample_rate"], S, nch)
        phase = cfg["channel_mode"] == config.CH_PHASE
        try:
            plan = api.Plan(cfg).upload()
        except api.SgzError:
            continue
        # (i) real-time path
        c = api.config_from_dict(cfg)
        h = C.c_void_p()
        api.check(L.sgz_spectrum_create(C.byref(c), C.byref(h)))
        cols, fed = [], 0
        while fed < S:
            n = int(min(S - fed, rng.integers(1, max(2, 2 * hop))))
            blk = np.ascontiguousarray(x[:, fed:fed + n])
            ptrs = (C.c_void_p * nch)(*[blk[i].ctypes.data for i in range(nch)])
            api.check(L.sgz_spectrum_push(h, ptrs, nch, n))
            fed += n
            cols += pop_all(h, P, fed // hop - len(cols))            # drain: the column queue holds 10 (frameQueue's depth)
        want = S // hop
        L.sgz_spectrum_destroy(h)
        padded = np.concatenate([np.zeros((nch, W), np.float32), x], axis=1)[:, hop:]
        # the per-block path must reproduce the batch render of [W zeros ++ audio] byte for byte (same kernels, decay state carried
        # exactly); the batch render itself is held against the oracle by the parity chain (tests/parity_chain.py)
        padded = np.ascontiguousarray(padded)
        # rows of a 64-sample multiple, like the handle's ring (before the real-input kernels took dword-aligned rows a plan fell back to
        # the complex ones on odd strides -- another rounding, while "byte for byte" must compare one kernel with itself)
        def aligned(a):
            t = torch.zeros((a.shape[0], (a.shape[1] + 63) // 64 * 64), dtype=torch.float32, device="cuda")
            t[:, :a.shape[1]] = torch.from_numpy(np.ascontiguousarray(a)).cuda()
            return t[:, :a.shape[1]]
        ref = plan.render(aligned(padded)).cpu().numpy()[:want]
        ok1 = len(cols) == want and (want == 0 or np.array_equal(np.stack(cols), ref))
        if ok1 and plan.num_frames(padded.shape[1]) > 0:
            problems, _ = check_render(po, plan, cfg, padded, torch.device("cuda:0"))
            ok1 = not problems
            if problems: print("   ", problems[:3])
        # (ii) split render with carried state
        ok2 = True
        if not phase or True:
            y = aligned(padded)
            F = plan.num_frames(padded.shape[1])
            if F >= 2:
                full = plan.render(y).cpu().numpy()
                cut = int(rng.integers(1, F))
                state = torch.zeros((cfg["num_pairs"], 2, P, 2), dtype=torch.float32, device="cuda")
                a = plan.render(aligned(padded[:, :W + (cut - 1) * hop]), state=state).cpu().numpy()
                b = plan.render(aligned(padded[:, cut * hop:]), state=state).cpu().numpy()
                ok2 = np.array_equal(np.concatenate([a, b]), full)
        if not ok1 and len(cols) == want and want:
            dd = np.abs(np.stack(cols).astype(int) - ref.astype(int))
            print("    rt max", int(dd.max()), "frac", float((dd > 0).mean()), "frames with diffs", np.unique(np.nonzero(dd)[0]).tolist()[:10],
                  "pixels", np.unique(np.nonzero(dd)[1]).tolist()[:10], "block sizes unknown; cfg:", {k: v for k, v in cfg.items() if k != "colours"})
        print(it, "ok " if ok1 and ok2 else "BAD", "N", plan.N, "path", plan.path, "mode", cfg["channel_mode"], "W", W, "hop", hop, "P", P,
              "pairs", cfg["num_pairs"], "cols", len(cols), "/", want, "rt", ok1, "split", ok2)
        bad += 0 if (ok1 and ok2) else 1
    print("bad:", bad, "of", count)
    sys.exit(1 if bad else 0)

main()
