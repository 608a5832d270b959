#!/usr/bin/env python3
"""Does a result depend on what its context ran before?  ONE long-lived solver context takes a random sequence of calls of every kind and
size -- local-BA batches of 1 .. 200 windows (both packers, windows of 1 .. 60 keyframes, a 300-keyframe map through the global-memory
factorisation), LocalInertialBA windows alone and in batches, FullInertialBA-shaped maps in the dense and in the banded layout -- and
every result is compared, bit for bit, with the same call in a FRESH context (buffers that grow and are reused, sections a kernel
assumes to be zero, state left by the previous call).  Found in r03: the banded layout of an inertial map read 15 columns past the band it
had written (profiles/r03_soak.txt).
Usage: python profiles/soak_reuse.py [seconds] [seed]"""
import dataclasses
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from orb_slam3_study_kr_amd import lba, synth  # noqa: E402
from orb_slam3_study_kr_amd import synth_inertial as si  # noqa: E402

LBA_FIELDS = ("pose_qt", "points", "edge_chi2", "chi2_trace")
LIBA_FIELDS = ("chi2_trace", "pose_twb", "vel", "bias_g", "bias_a", "points")


def same(a, b, fields):
    if a.iterations != b.iterations or a.trials != b.trials:
        return f"iterations {a.iterations}/{b.iterations} trials {a.trials}/{b.trials}"
    for f in fields:
        if not np.array_equal(getattr(a, f), getattr(b, f)):
            return f"{f}: max difference {np.abs(np.asarray(getattr(a, f), dtype=np.float64) - np.asarray(getattr(b, f), dtype=np.float64)).max():.3g}"
    return None


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31)
    t_end = time.time() + seconds
    pool = {}

    def lba_window(k):
        if ("l", k) not in pool:
            r = np.random.default_rng(k)
            pool[("l", k)] = synth.make_window(3000 + k, n_free=int(r.choice([1, 3, 8, 20, 47, 60])), n_fixed=int(r.integers(1, 6)),
                                               n_points=int(r.choice([60, 400, 1500, 5000])), stereo=bool(r.random() < 0.7),
                                               track_len=(3, int(r.integers(5, 20))), obs_dropout=float(r.choice([0.0, 0.2])), max_iterations=int(r.choice([3, 6])))
        return pool[("l", k)]

    def inertial_window(k, n_opt=None):
        key = ("i", k, n_opt)
        if key not in pool:
            if n_opt is None:
                pool[key] = si.make_inertial_window(4000 + k, n_opt=int(np.random.default_rng(k).choice([4, 10, 14])), n_points=int(np.random.default_rng(k + 1).choice([600, 2000, 3600])))
            else:
                w = si.make_inertial_window(900 + n_opt, n_opt=n_opt, n_fixed=0, n_points=40 * n_opt, large=True)
                pool[key] = dataclasses.replace(w, lambda_init=1e-5, max_iterations=4, link_robust=np.ones_like(w.link_robust))
        return pool[key]

    n = 0
    kinds = {}
    with lba.LbaSolver(0) as live:
        while time.time() < t_end:
            kind = str(rng.choice(["lba_small", "lba_batch", "lba_map", "liba_one", "liba_batch", "liba_dense_map", "liba_banded_map"],
                              p=[0.25, 0.2, 0.05, 0.15, 0.15, 0.1, 0.1]))
            if kind in ("lba_small", "lba_batch", "lba_map"):
                if kind == "lba_small":
                    ws = [lba_window(int(rng.integers(0, 40))) for _ in range(int(rng.integers(1, 6)))]
                elif kind == "lba_batch":
                    ws = [lba_window(int(rng.integers(0, 40))) for _ in range(int(rng.choice([24, 60, 200])))]
                else:
                    if ("m",) not in pool:
                        pool[("m",)] = synth.make_window(77, n_free=300, n_fixed=2, n_points=12000, stereo=True, track_len=(3, 25), max_iterations=3)
                    ws = [pool[("m",)]]
                mode = int(rng.choice([-1, 0, 1]))
                live.set_pack_mode(mode)
                got = live.solve(ws)
                with lba.LbaSolver(0) as fresh:
                    fresh.set_pack_mode(mode)
                    ref = fresh.solve(ws)
                fields = LBA_FIELDS
            else:
                if kind == "liba_one":
                    ws = [inertial_window(int(rng.integers(0, 12)))]
                elif kind == "liba_batch":
                    ws = [inertial_window(int(rng.integers(0, 12))) for _ in range(int(rng.choice([3, 20, 130])))]
                elif kind == "liba_dense_map":
                    ws = [inertial_window(0, n_opt=int(rng.choice([60, 100])))]
                else:
                    ws = [inertial_window(0, n_opt=int(rng.choice([150, 250, 400])))]
                got = live.solve_inertial(ws)
                with lba.LbaSolver(0) as fresh:
                    ref = fresh.solve_inertial(ws)
                fields = LIBA_FIELDS
            for i, (a, b) in enumerate(zip(got, ref)):
                err = same(a, b, fields)
                if err:
                    print(f"MISMATCH call {n} ({kind}, {len(ws)} windows), window {i}: {err}", flush=True)
                    return 1
            kinds[kind] = kinds.get(kind, 0) + 1
            n += 1
            if n % 10 == 0:
                print(f"{n} calls ok {kinds}", flush=True)
    print(f"soak ok: {n} calls in one context, each equal bit for bit to the same call in a fresh context: {kinds}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
