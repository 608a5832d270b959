#!/usr/bin/env python3
"""LocalInertialBA single-window latency: wall time of osh_liba_solve and the phase cycle counters of the window's block group
(osh_liba_get_profile).  OSH_LIBA_GROUP=1|2|4|8 overrides the blocks per window."""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from orb_slam3_study_kr_amd import lba  # noqa: E402
from orb_slam3_study_kr_amd import synth_inertial as si  # noqa: E402


def main():
    ws = [si.make_inertial_window(11 + k) for k in range(8)]
    sv = lba.LbaSolver(0)
    out = {"group_env": os.environ.get("OSH_LIBA_GROUP")}
    sv.solve_inertial(ws[:1])
    t0 = time.perf_counter()
    for _ in range(10):
        r = sv.solve_inertial(ws[:1])
    out["single_window_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    grp, cyc = sv.inertial_profile()
    out["group"] = grp
    out["iterations"] = int(r[0].iterations)
    out["trials"] = int(r[0].trials)
    out["cycles"] = cyc
    out["cycles_total"] = sum(cyc.values())
    batch = [ws[k % 8] for k in range(128)]
    sv.solve_inertial(batch)
    t0 = time.perf_counter()
    sv.solve_inertial(batch)
    out["batch_128_ms"] = (time.perf_counter() - t0) * 1e3
    out["batch_group"] = sv.inertial_profile()[0]
    sv.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
