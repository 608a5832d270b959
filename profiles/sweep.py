#!/usr/bin/env python3
"""SURVEY.md 8(d) batch-size sweeps on one GPU (run on the GPU box, writes gpurun_out/sweep_<tag>.json):
   * local BA (config 2): windows/s and ms per optimize() for B in {1, 8, 64, 256, 512} resident windows, one stream;
     plus the end-to-end time of one window through upload + optimize + download (the live-SLAM call pattern)
   * ORB matching (config 3): frame pairs/s and pair evaluations/s for B in {1, 64, 1024} pairs per launch
   * LocalInertialBA (config 4): windows/s for B in {1, 8, 128}
   * CPU baseline of config 2 on ALL host cores (one independent window per process, oracle restatement) next to the
     single-thread figure bench.py reports.
usage: python3 profiles/sweep.py <tag>"""
import json
import multiprocessing as mp
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from orb_slam3_study_kr_amd import synth  # noqa: E402


def _mk(seed):
    return synth.make_config2(seed)


def _cpu_worker(args):
    seed, seconds = args
    from oracle import binding as ob
    w = synth.make_config2(seed)
    ob.lba_solve(w, native=True)          # warm (and build check)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        ob.lba_solve(w, native=True)
        n += 1
    return n, time.perf_counter() - t0


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    out = {"tag": tag}
    cores = len(os.sched_getaffinity(0))
    # ---- everything that forks runs BEFORE the GPU is touched
    from oracle import binding as ob
    ob.load(native=True)                  # build the -march=native oracle once, in the parent
    with mp.get_context("fork").Pool(cores) as pool:
        windows = pool.map(_mk, [100 + k for k in range(512)])
        res = pool.map(_cpu_worker, [(100 + k, 10.0) for k in range(cores)])
    out["cpu_all_cores"] = {"cores": cores, "windows_per_s": sum(n / t for n, t in res),
                            "kind": "port", "sample": f"{sum(n for n, _ in res)} solves, one independent config-2 window per process, 10 s"}
    n1, t1 = _cpu_worker((100, 10.0))
    out["cpu_one_thread"] = {"cores": 1, "windows_per_s": n1 / t1}

    from orb_slam3_study_kr_amd import lba, orb
    from orb_slam3_study_kr_amd import synth_inertial as si
    solver = lba.LbaSolver(0)
    rows = []
    for B in (1, 8, 64, 256, 512):
        ws = windows[:B]
        t0 = time.perf_counter(); solver.upload(ws); t_up = time.perf_counter() - t0
        solver.optimize()
        reps = 20 if B <= 8 else (5 if B <= 64 else 3)
        t0 = time.perf_counter()
        for _ in range(reps):
            solver.optimize()
        t_opt = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter(); solver.download(); t_dn = time.perf_counter() - t0
        rows.append({"windows": B, "optimize_ms": t_opt * 1e3, "windows_per_s": B / t_opt, "upload_ms": t_up * 1e3, "download_ms": t_dn * 1e3})
        print(rows[-1], flush=True)
    out["lba_batch_sweep"] = rows
    t0 = time.perf_counter()
    for k in range(10):
        solver.solve([windows[k]])
    out["lba_single_window_end_to_end_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    solver.close()

    base = synth.make_orb_pair(7, 2000, 2000)
    m = orb.OrbMatcher(0)
    rows = []
    for B in (1, 64, 1024):
        m.upload([base] * B)
        m.match(); m.download()
        reps = 50 if B == 1 else (20 if B == 64 else 5)
        t0 = time.perf_counter()
        for _ in range(reps):
            m.match()
        m.download()
        dt = (time.perf_counter() - t0) / reps
        rows.append({"pairs": B, "ms_per_launch": dt * 1e3, "frame_pairs_per_s": B / dt, "pair_evals_per_s": B * 4.0e6 / dt})
        print(rows[-1], flush=True)
    out["orb_batch_sweep"] = rows
    m.close()

    solver = lba.LbaSolver(0)
    iw = [si.make_inertial_window(11 + k) for k in range(8)]
    rows = []
    for B in (1, 8, 128):
        ws = [iw[k % 8] for k in range(B)]
        solver.solve_inertial(ws)
        reps = 10 if B <= 8 else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            solver.solve_inertial(ws)
        dt = (time.perf_counter() - t0) / reps
        rows.append({"windows": B, "ms_per_call": dt * 1e3, "windows_per_s": B / dt})
        print(rows[-1], flush=True)
    out["inertial_batch_sweep"] = rows
    solver.close()
    # ---- Optimizer::PoseOptimization: one frame (the live call pattern) and a batch, next to the single-thread oracle
    frames = [synth.make_pose_frame(60 + k, n_points=1200) for k in range(256)]
    t0 = time.perf_counter(); n_cpu = 0
    while time.perf_counter() - t0 < 3.0:
        ob.pose_optimize(frames[n_cpu % 256], native=True); n_cpu += 1
    cpu_fps = n_cpu / (time.perf_counter() - t0)
    solver = lba.LbaSolver(0)
    rows = []
    for B in (1, 16, 256):
        fs = frames[:B]
        solver.optimize_poses(fs)
        reps = 20 if B == 1 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            solver.optimize_poses(fs)
        dt = (time.perf_counter() - t0) / reps
        rows.append({"frames": B, "ms_per_call": dt * 1e3, "frames_per_s": B / dt})
        print(rows[-1], flush=True)
    out["pose_optimization"] = {"edges_per_frame": int(np.mean([f.n_edges for f in frames])), "sweep": rows,
                                "cpu_one_thread_frames_per_s": cpu_fps, "includes": "H2D upload + D2H download"}
    solver.close()
    dst = ROOT / "gpurun_out"
    dst.mkdir(exist_ok=True)
    (dst / f"sweep_{tag}.json").write_text(json.dumps(out, indent=1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
