import numpy as np, dataclasses, sys
from orb_slam3_study_kr_amd import synth_inertial as si, lba
from oracle import binding as ob
for n_opt, its in ((12, 1), (12, 3)):
    w = si.with_shared_bias(si.make_inertial_window(300 + n_opt, n_opt=n_opt, n_fixed=6, n_points=60 * n_opt + 200))
    w = dataclasses.replace(w, lambda_init=1e-5, max_iterations=its)
    ref = ob.liba_solve(w)
    with lba.LbaSolver(0) as s:
        got = s.solve_inertial([w])[0]
    print("its", its, got.chi2_trace[:its], ref.chi2_trace[:its], got.chi2_final, ref.chi2_final)
    print(" dt", np.linalg.norm(got.pose_tcw - ref.pose_tcw, axis=1))
    print(" dv", np.linalg.norm(got.vel - ref.vel, axis=1))
    print(" dbg", np.linalg.norm(got.bias_g - ref.bias_g, axis=1), "dba", np.linalg.norm(got.bias_a - ref.bias_a, axis=1))
    print(" dpts", np.abs(got.points - ref.points).max())
