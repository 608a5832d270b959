"""Per-phase shader-clock breakdown of the inertial kernel on map-sized problems (FullInertialBA-shaped windows of 25..1000 keyframes):
`PYTHONPATH=. python profiles/liba_map_profile.py` on the GPU box (OSH_LIBA_DENSE=1: the dense layout of round 2, up to 400 keyframes).
Output of r02: profiles/r02_liba_map_profile.txt, of r03: profiles/r03_liba_map_profile.txt."""
import os
import numpy as np, dataclasses, time
from orb_slam3_study_kr_amd import synth_inertial as si, lba
names = ["linearise", "assembly", "Dinv", "Schur", "LDLT", "backsub", "errors", "outputs"]
with lba.LbaSolver(0) as s:
    for n_opt in (25, 50, 100, 150, 200, 400) + (() if os.environ.get("OSH_LIBA_DENSE") else (1000,)):
        w = si.make_inertial_window(900 + n_opt, n_opt=n_opt, n_fixed=0, n_points=40 * n_opt, large=True)
        w = dataclasses.replace(w, lambda_init=1e-5, max_iterations=7, link_robust=np.ones_like(w.link_robust))
        s.solve_inertial([w]); t0 = time.perf_counter(); r = s.solve_inertial([w])[0]; ms = (time.perf_counter() - t0) * 1e3
        g, cyc = s.inertial_profile()
        tot = sum(cyc.values())
        print(n_opt, "ms %.1f" % ms, "its", r.iterations, "trials", int(np.sum(r.trials_trace[:r.iterations])), "group", g, {n: round(c / max(tot, 1), 3) for n, c in cyc.items()}, "Mcycles", tot / 1e6)
