#!/usr/bin/env python3
"""Randomised soak of the local-BA path (not a benchmark): for a few minutes, batches of random size of windows of random shape
(1 .. 60 optimisable keyframes, 0 .. 8 fixed, 20 .. 4000 landmarks, mono / stereo / mixed / fisheye / fisheye rig, track lengths up to 30,
missed detections, outliers, shuffled edges) go through
  * osh_lba_pack_compare: the device packer against the host packer, every section byte for byte,
  * osh_lba_solve with the batch packed on the device against the batch packed on the host: bitwise equal; against the same windows
    solved one at a time: the same cost trace (alone a window may take another panel width in k_solve),
  * the CPU oracle on the small windows of the batch: iterations, trials, cost trace, translations to 1e-6 relative (gauge-free windows
    -- no fixed keyframe -- and windows with fewer than twenty landmarks per keyframe are compared on the cost trace only).
Usage: python profiles/soak.py [seconds] [seed]     (prints one line per batch and a summary; exit code 1 on the first mismatch)"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from helpers import rel_translation_error, rotation_error  # noqa: E402
from orb_slam3_study_kr_amd import lba, synth  # noqa: E402
from oracle import binding as ob  # noqa: E402


def random_window(rng, k):
    kind = rng.choice(["stereo", "mono", "mixed", "fisheye", "rig"], p=[0.4, 0.2, 0.15, 0.15, 0.1])
    n_free = int(rng.choice([1, 2, 3, 5, 8, 12, 20, 33, 47, 60], p=[0.05, 0.05, 0.1, 0.15, 0.2, 0.15, 0.1, 0.1, 0.05, 0.05]))
    n_fixed = int(rng.integers(1 if kind in ("mono", "fisheye") else 0, 9))
    if kind in ("mono", "fisheye"):
        n_fixed = max(n_fixed, 2)            # scale needs two fixed views
    n_points = int(rng.choice([20, 80, 300, 900, 2000, 4000], p=[0.1, 0.2, 0.3, 0.2, 0.15, 0.05]))
    lo = int(rng.integers(2 if kind in ("stereo", "mixed") else 3, 6))   # two-view monocular landmarks: weak geometry, rounding amplified 1e3 times
    hi = int(rng.integers(lo + 1, 31))
    args = dict(n_free=n_free, n_fixed=n_fixed, n_points=n_points, track_len=(lo, hi), obs_dropout=float(rng.choice([0.0, 0.1, 0.3])),
                outlier_frac=float(rng.choice([0.0, 0.03, 0.1])), max_iterations=int(rng.choice([3, 5, 10])))
    seed = 100000 + k
    if kind == "rig":
        w = synth.make_rig_window(seed, n_free=n_free, n_fixed=max(n_fixed, 1), n_points=n_points, track_len=(lo, min(hi, 12)))
    elif kind == "fisheye":
        w = synth.make_window(seed, stereo=False, fisheye=True, **args)
    elif kind == "mono":
        w = synth.make_window(seed, stereo=False, **args)
    elif kind == "mixed":
        w = synth.make_window(seed, stereo=True, mixed_mono_frac=0.4, **args)
    else:
        w = synth.make_window(seed, stereo=True, **args)
    if rng.random() < 0.3 and kind != "rig":
        perm = rng.permutation(w.n_edges)
        w = synth.LbaWindow(n_free=w.n_free, n_fixed=w.n_fixed, pose_qt=w.pose_qt, pose_cam=w.pose_cam, points=w.points,
                            edge_pose=np.ascontiguousarray(w.edge_pose[perm]), edge_point=np.ascontiguousarray(w.edge_point[perm]),
                            edge_kind=np.ascontiguousarray(w.edge_kind[perm]), edge_obs=np.ascontiguousarray(w.edge_obs[perm]),
                            edge_info=np.ascontiguousarray(w.edge_info[perm]), kb8=w.kb8, max_iterations=w.max_iterations).normalise()
    return kind, w


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    t_end = time.time() + seconds
    n_batches = n_windows = n_oracle = 0
    k = 0
    with lba.LbaSolver(0) as sv:
        while time.time() < t_end:
            B = int(rng.choice([1, 2, 5, 24, 31, 40]))
            ws, kinds = [], []
            for _ in range(B):
                kind, w = random_window(rng, k)
                k += 1
                ws.append(w); kinds.append(kind)
            st = sv.pack_compare(ws)
            sv.set_pack_mode(1)
            dev = sv.solve(ws)
            sv.set_pack_mode(0)
            hst = sv.solve(ws)
            for i, (a, b) in enumerate(zip(dev, hst)):
                if not (a.iterations == b.iterations and a.trials == b.trials and np.array_equal(a.pose_qt, b.pose_qt)
                        and np.array_equal(a.points, b.points) and np.array_equal(a.edge_chi2, b.edge_chi2)):
                    print(f"MISMATCH batch {n_batches} window {i} ({kinds[i]}): batch packed on the device vs on the host", flush=True)
                    return 1
            picks = sorted(set(int(i) for i in rng.integers(0, B, size=min(B, 3))))
            for i in picks:
                # alone, the window may take another panel width in k_solve (chosen by the largest system of the batch): same path, other rounding
                one = sv.solve([ws[i]])[0]
                a = dev[i]
                if a.iterations == one.iterations and a.trials == one.trials:
                    n = a.iterations
                    if not np.allclose(a.chi2_trace[:n], one.chi2_trace[:n], rtol=1e-6):
                        print(f"MISMATCH batch {n_batches} window {i} ({kinds[i]}): in the batch vs alone, cost trace", flush=True)
                        return 1
                w = ws[i]
                if w.n_edges <= 12000:
                    ref = ob.lba_solve(w)
                    n_oracle += 1
                    n = min(a.iterations, ref.iterations)
                    same_path = a.iterations == ref.iterations and a.trials == ref.trials
                    tr_ok = np.allclose(a.chi2_trace[:n], ref.chi2_trace[:n], rtol=2e-5)
                    anchored = w.n_fixed >= 1 and w.n_points >= 20 * w.n_free    # (a few landmarks under many keyframes: the poses are barely determined)
                    pose_ok = True
                    if anchored and same_path and w.n_free > 0:
                        pose_ok = (rel_translation_error(a.pose_qt[:w.n_free], ref.pose_qt[:w.n_free]) < 5e-6
                                   and rotation_error(a.pose_qt[:w.n_free], ref.pose_qt[:w.n_free]) < 5e-6)
                    if not (tr_ok and pose_ok and abs(a.iterations - ref.iterations) <= 1):
                        print(f"ORACLE MISMATCH batch {n_batches} window {i} ({kinds[i]}, seed {100000 + k - B + i}): iterations {a.iterations}/{ref.iterations} "
                              f"trials {a.trials}/{ref.trials} trace ok {tr_ok} pose ok {pose_ok}", flush=True)
                        print("  shape:", w.n_free, w.n_fixed, w.n_points, w.n_edges, "errors:", rel_translation_error(a.pose_qt[:w.n_free], ref.pose_qt[:w.n_free]),
                              rotation_error(a.pose_qt[:w.n_free], ref.pose_qt[:w.n_free]), flush=True)
                        return 1
            sv.set_pack_mode(-1)
            n_batches += 1
            n_windows += B
            print(f"batch {n_batches}: {B} windows ({', '.join(sorted(set(kinds)))}), {st['records']} records, {st['items']} items: ok", flush=True)
    print(f"soak ok: {n_batches} batches, {n_windows} windows packed twice and solved, {n_oracle} checked against the oracle")
    return 0


if __name__ == "__main__":
    sys.exit(main())
