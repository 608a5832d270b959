#!/usr/bin/env python3
"""The kernels of the path besides the local-BA loop, run once each for rocprofv3 (profiles/run_profiles.sh, no worker
processes): ORB brute-force + windowed search (k_orb_bruteforce / k_orb_grid), LocalInertialBA (k_liba), PoseOptimization
(k_pose_opt), the frustum projection (k_frustum).  Prints the HIP-event / wall timings it measured itself."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from orb_slam3_study_kr_amd import lba, orb, synth  # noqa: E402
from orb_slam3_study_kr_amd import synth_inertial as si  # noqa: E402


def main():
    out = {}
    uniform = "--uniform" in sys.argv   # counter passes: every launch of a kernel does the same work (its per-launch mean is then a per-launch value)
    base = synth.make_orb_pair(7, 2000, 2000)
    pairs = [base] * 64
    m = orb.OrbMatcher(0)
    m.upload(pairs)
    m.match()
    m.set_profiling(True)
    for _ in range(5):
        m.match()
    launches, ms = m.profile()
    out["orb_bruteforce_ms_per_launch_64_pairs"] = ms / max(launches, 1)
    if uniform:
        m.close()
        ws = [si.make_inertial_window(11 + k, n_points=3600) for k in range(8)]
        sv = lba.LbaSolver(0)
        batch = [ws[k % 8] for k in range(128)]
        for _ in range(3):
            sv.solve_inertial(batch)
        sv.close()
        print(json.dumps(out))
        return
    # the whole matching loop with the sequential slot occupancy resolved on the device (k_orb_claim / k_orb_research rounds)
    m.match_local_points()
    t0 = time.perf_counter()
    for _ in range(5):
        n_m, _, _, rounds = m.match_local_points()
    out["orb_match_local_points_ms_64_pairs"] = (time.perf_counter() - t0) / 5 * 1e3
    out["orb_occupancy_rounds"] = rounds
    m.close()
    ws = [si.make_inertial_window(11 + k, n_points=3600) for k in range(8)]   # BASELINE.json configs[3] at ~2 000 landmarks
    sv = lba.LbaSolver(0)
    sv.solve_inertial(ws[:1])
    t0 = time.perf_counter()
    for _ in range(5):
        sv.solve_inertial(ws[:1])
    out["liba_single_window_ms"] = (time.perf_counter() - t0) / 5 * 1e3
    batch = [ws[k % 8] for k in range(128)]
    sv.solve_inertial(batch)
    t0 = time.perf_counter()
    sv.solve_inertial(batch)
    out["liba_128_windows_ms"] = (time.perf_counter() - t0) * 1e3
    frames = [synth.make_pose_frame(300 + k) for k in range(64)] if hasattr(synth, "make_pose_frame") else []
    if frames:
        sv.optimize_poses(frames[:1])
        t0 = time.perf_counter()
        for _ in range(5):
            sv.optimize_poses(frames[:1])
        out["pose_single_frame_ms"] = (time.perf_counter() - t0) / 5 * 1e3
        sv.optimize_poses(frames)
        t0 = time.perf_counter()
        sv.optimize_poses(frames)
        out["pose_64_frames_ms"] = (time.perf_counter() - t0) * 1e3
    # PoseInertialOptimizationLastKeyFrame / LastFrame (k_posei): one frame (the tracker's call) and a batch
    pf = [si.make_posei_frame(70 + k, mode=k % 2, n_points=800) for k in range(64)]
    sv.optimize_poses_inertial(pf[:1])
    t0 = time.perf_counter()
    for _ in range(5):
        sv.optimize_poses_inertial(pf[:1])
    out["posei_single_frame_ms"] = (time.perf_counter() - t0) / 5 * 1e3
    sv.optimize_poses_inertial(pf)
    t0 = time.perf_counter()
    sv.optimize_poses_inertial(pf)
    out["posei_64_frames_ms"] = (time.perf_counter() - t0) * 1e3
    # global BA of a 600-keyframe map: the reduced system (n = 3594) goes through csrc/big_solve.h
    big = synth.make_window(900, n_free=599, n_fixed=1, n_points=18000, stereo=True, max_iterations=5)
    sv.upload([big])
    t0 = time.perf_counter()
    sv.optimize()
    out["gba_600_keyframes_optimize_ms"] = (time.perf_counter() - t0) * 1e3
    r = sv.download()[0]
    out["gba_600_keyframes_iterations"] = int(r.iterations)
    sv.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
