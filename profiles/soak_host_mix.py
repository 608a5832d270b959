#!/usr/bin/env python3
"""The host layer keeps ONE device context per thread (csrc/host/Optimizer.cc) for every Optimizer call a SLAM session makes -- local BA,
LocalInertialBA, FullInertialBA, global BA, of whatever size comes next.  This script makes such a session: a random sequence of calls
through the reference signatures, each on a freshly built synthetic map from a fixed list of scenarios, and compares what every call
wrote into its map (keyframe poses, velocities, biases, map points: the float32 values) BIT FOR BIT with what the same scenario wrote
in a process of its own (one subprocess per scenario, run first).  A difference means the result depended on the context's history.
Usage: python profiles/soak_host_mix.py [seconds] [seed]        (internal: --ref K OUT.npz)"""
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from orb_slam3_study_kr_amd import host, synth  # noqa: E402
from orb_slam3_study_kr_amd import synth_inertial as si  # noqa: E402


def lba_state(g, w):
    return dict(kf=np.stack([g.kf_pose(k) for k in range(w.n_free + w.n_fixed)]), mp=np.stack([g.mp_pos(j) for j in range(w.n_points)]))


def inertial_state(g):
    n = len(g.kf_id)
    return dict(kf=np.stack([g.kf_pose(k) for k in range(n)]), vel=np.stack([g.kf_velocity(k) for k in range(n)]),
                bias=np.stack([g.kf_bias(k) for k in range(n)]), mp=np.stack([g.mp_pos(j) for j in range(len(g.mp_id))]))


def run_lba(w):
    with host.HostGraph(w) as g:
        g.run_lba()
        return lba_state(g, w)


def run_gba(w, its):
    with host.HostGraph(w, init_kf_id_index=w.n_free) as g:
        g.run_gba(its, 0)
        return lba_state(g, w)


def run_liba(w, large=False):
    with host.HostInertialGraph(w) as g:
        assert g.run(large=large) == 0
        return inertial_state(g)


def run_full(w, its, init=False):
    with host.HostInertialGraph(w) as g:
        assert g.run_full(its, 0, init=init) == 0
        return inertial_state(g)


SCENARIOS = [
    ("local BA, 6 + 2 keyframes stereo", lambda: run_lba(synth.make_window(501, n_free=6, n_fixed=2, n_points=400, stereo=True))),
    ("local BA, 24 + 5 keyframes stereo", lambda: run_lba(synth.make_window(502, n_free=24, n_fixed=5, n_points=3000, stereo=True))),
    ("local BA, fisheye", lambda: run_lba(synth.make_window(503, n_free=8, n_fixed=3, n_points=700, stereo=False, fisheye=True, track_len=(3, 8)))),
    ("local BA, fisheye rig", lambda: run_lba(synth.make_rig_window(504, n_free=7, n_fixed=3, n_points=500, track_len=(3, 8)))),
    ("local BA, 50 + 6 keyframes", lambda: run_lba(synth.make_window(505, n_free=50, n_fixed=6, n_points=5000, stereo=True))),
    ("LocalInertialBA, 10 keyframes", lambda: run_liba(si.make_inertial_window(51, n_opt=10, n_fixed=8, n_points=1200))),
    ("LocalInertialBA, 25 keyframes (bLarge)", lambda: run_liba(si.make_inertial_window(52, n_opt=25, n_fixed=8, n_points=1500, large=True), large=True)),
    ("FullInertialBA, 19 keyframes", lambda: run_full(si.make_inertial_window(81, n_opt=14, n_fixed=4, n_points=900), 7)),
    ("FullInertialBA, 75 keyframes (group factorisation)", lambda: run_full(si.make_inertial_window(81, n_opt=70, n_fixed=4, n_points=3000), 5)),
    ("FullInertialBA, 160 keyframes (banded)", lambda: run_full(si.make_inertial_window(906, n_opt=155, n_fixed=4, n_points=6000, large=True), 4)),
    ("FullInertialBA with bInit", lambda: run_full(si.make_inertial_window(81, n_opt=14, n_fixed=4, n_points=900), 7, init=True)),
    ("global BA, 80 keyframes", lambda: run_gba(synth.make_window(46, n_free=80, n_fixed=1, n_points=4000, stereo=True, track_len=(3, 10)), 5)),
    ("global BA, 320 keyframes (global-memory factorisation)", lambda: run_gba(synth.make_window(46, n_free=319, n_fixed=1, n_points=5000, stereo=True), 3)),
]


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--ref":
        k = int(sys.argv[2])
        np.savez(sys.argv[3], **SCENARIOS[k][1]())
        return 0
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
    refs = []
    for k in range(len(SCENARIOS)):
        out = f"/tmp/soak_host_mix_ref_{k}.npz"
        r = subprocess.run([sys.executable, __file__, "--ref", str(k), out], capture_output=True, text=True)
        if r.returncode != 0:
            print("reference process failed:", SCENARIOS[k][0], r.stderr[-600:])
            return 1
        refs.append({n: v for n, v in np.load(out).items()})
    print(f"{len(refs)} references from processes of their own", flush=True)
    t_end = time.time() + seconds
    n = 0
    counts = [0] * len(SCENARIOS)
    while time.time() < t_end:
        k = int(rng.integers(0, len(SCENARIOS)))
        got = SCENARIOS[k][1]()
        for name, v in refs[k].items():
            if not np.array_equal(got[name], v):
                print(f"MISMATCH call {n}: {SCENARIOS[k][0]}: {name} differs from the fresh process by {np.abs(got[name].astype(np.float64) - v.astype(np.float64)).max():.3g}", flush=True)
                return 1
        counts[k] += 1
        n += 1
        if n % 20 == 0:
            print(f"{n} calls ok", flush=True)
    print(f"soak ok: {n} Optimizer calls of {len(SCENARIOS)} kinds in one thread (one device context), each wrote the bits it writes in a process of its own: {counts}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
