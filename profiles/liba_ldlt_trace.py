"""Cycles per phase of the group LDL^T of k_liba on a 400-keyframe map problem (build with -DOSH_LIBA_LDLT_TRACE)."""
import dataclasses, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from orb_slam3_study_kr_amd import synth_inertial as si, lba
with lba.LbaSolver(0) as s:
    w = si.make_inertial_window(1300, n_opt=400, n_fixed=0, n_points=16000, large=True)
    w = dataclasses.replace(w, lambda_init=1e-5, max_iterations=1, link_robust=np.ones_like(w.link_robust))
    s.solve_inertial([w])
