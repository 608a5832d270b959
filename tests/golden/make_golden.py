"""Generates the committed golden fixtures under tests/golden/.

The reference ships no golden vectors for this path and cannot be built here
(SURVEY.md section 8c), so the expected outputs come from the INDEPENDENT numpy
implementation oracle/lm_numpy.py (dense normal equations, matrix-exponential
updates) for local BA and from numpy.unpackbits popcounts for Hamming distances.
Run from the repo root:  python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from orb_slam3_study_kr_amd import synth  # noqa: E402
from oracle import lm_numpy  # noqa: E402
from oracle import liba_numpy  # noqa: E402
from oracle import pose_numpy  # noqa: E402
from orb_slam3_study_kr_amd import synth_inertial  # noqa: E402

OUT = Path(__file__).resolve().parent


def window_arrays(w):
    return dict(n_free=w.n_free, n_fixed=w.n_fixed, pose_qt=w.pose_qt, pose_cam=w.pose_cam, points=w.points,
                edge_pose=w.edge_pose, edge_point=w.edge_point, edge_kind=w.edge_kind, edge_obs=w.edge_obs,
                edge_info=w.edge_info, huber_mono=w.huber_mono, huber_stereo=w.huber_stereo,
                lambda_init=w.lambda_init, max_iterations=w.max_iterations,
                kb8=w.kb8 if w.kb8 is not None else np.zeros(0), cam2=w.cam2 if w.cam2 is not None else np.zeros(0),
                trl=w.trl if w.trl is not None else np.zeros(0))


def lba_fixture(name, w):
    st0 = lm_numpy.State(w)
    errs = lm_numpy.errors(w, st0)
    chi0, per_edge0 = lm_numpy.robust_chi2(w, errs)
    H, b = lm_numpy.build_dense_system(w, st0, errs)
    lam = 1e-5 * np.max(np.abs(np.diag(H)))
    x = np.linalg.solve(H + lam * np.eye(H.shape[0]), b)
    st, trace, per_edge = lm_numpy.lm_optimize(w)
    np.savez_compressed(
        OUT / f"{name}.npz", **window_arrays(w),
        exp_chi2_initial=chi0, exp_edge_chi2_initial=per_edge0, exp_H=H, exp_b=b, exp_lambda0=lam, exp_x0=x,
        exp_T=st.T[:w.n_free], exp_points=st.X, exp_chi2_trace=np.array(trace["chi2"]),
        exp_lambda_trace=np.array(trace["lam"]), exp_trials_trace=np.array(trace["trials"]),
        exp_iterations=trace["iterations"], exp_edge_chi2_final=per_edge)
    print(name, "edges", w.n_edges, "iters", trace["iterations"], "trials", trace["trials"], "chi2", chi0, "->", trace["chi2"][-1])


def orb_fixture():
    rng = np.random.Generator(np.random.PCG64(2024))
    a = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    b[5] = a[7]                      # distance 0
    b[9] = ~a[3]                     # distance 256
    b[11] = a[2]; b[11, 0] ^= 0x81   # distance 2
    dist = (np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(axis=2)).astype(np.int32)
    np.savez_compressed(OUT / "orb_64x64.npz", a=a, b=b, dist=dist)
    print("orb_64x64", dist.min(), dist.max())


POSEI_INPUTS = ("Rcw", "tcw", "Rwb", "twb", "vel", "bias_g", "bias_a", "prev_Rwb", "prev_twb", "prev_vel", "prev_bias_g", "prev_bias_a", "Rcb", "tcb",
                "tbc", "cam", "preint", "info_inertial", "info_g", "info_a", "points", "edge_kind", "edge_obs", "edge_info", "edge_close", "prior_Rwb",
                "prior_twb", "prior_vel", "prior_bg", "prior_ba", "prior_H", "kb8", "cam2", "trl")


def pose_fixture(name, f):
    """PoseOptimization: the flat frame + the outputs of the independent numpy model (oracle/pose_numpy.py: numeric Jacobians, own
    Levenberg-Marquardt loop and classification rounds)."""
    r = pose_numpy.pose_optimize(f)
    np.savez_compressed(OUT / f"{name}.npz", pose_qt=f.pose_qt, cam=f.cam, points=f.points, edge_kind=f.edge_kind, edge_obs=f.edge_obs, edge_info=f.edge_info,
                        huber=np.array([f.huber_mono, f.huber_stereo]), chi2_mono=np.array(f.chi2_mono), chi2_stereo=np.array(f.chi2_stereo),
                        iterations=np.array(f.iterations), exp_T=r["T"], exp_outlier=r["outlier"], exp_n_bad=r["n_bad"], exp_chi2_final=r["chi2_final"],
                        exp_edge_chi2=r["edge_chi2"], exp_rounds=r["rounds"])
    print(name, "edges", f.n_edges, "iterations", r["iterations"], "bad", r["n_bad"], "chi2", r["chi2_final"])


def posei_fixture(name, f):
    """PoseInertialOptimizationLastKeyFrame / LastFrame: inputs + the outputs of the independent numpy model (oracle/liba_numpy.py:
    central-difference Jacobians, numpy.linalg.solve, its own classification loop)."""
    g = liba_numpy.posei_optimize(f)
    arrays = {k: (getattr(f, k) if getattr(f, k) is not None else np.zeros(0)) for k in POSEI_INPUTS}
    np.savez_compressed(OUT / f"{name}.npz", mode=f.mode, rec_init=int(f.rec_init), chi2_mono=np.array(f.chi2_mono), chi2_stereo=np.array(f.chi2_stereo),
                        iterations=np.array(f.iterations), huber=np.array([f.huber_mono, f.huber_stereo, f.huber_prior]), **arrays,
                        **{"exp_" + k: np.asarray(v) for k, v in g.items()})
    print(name, "mode", f.mode, "edges", f.n_edges, "rounds", g["rounds"], "n_bad", g["n_bad"])


LIBA_INPUTS = ("pose_Rcw", "pose_tcw", "pose_Rwb", "pose_twb", "Rcb", "tcb", "tbc", "cam", "vel", "bias_g", "bias_a", "points", "edge_pose",
               "edge_point", "edge_kind", "edge_obs", "edge_info", "link_prev", "link_cur", "link_preint", "link_info", "link_info_g",
               "link_info_a", "link_robust", "kb8", "cam2", "trl")


def liba_fixture(name, w):
    """LocalInertialBA: inputs + the outputs of the independent numpy Levenberg-Marquardt (oracle/liba_numpy.py:lm_optimize: full dense
    system from central-difference Jacobians, numpy.linalg.solve, g2o's controller)."""
    st, tr = liba_numpy.lm_optimize(w)
    N = w.n_opt
    arrays = {k: (getattr(w, k) if getattr(w, k) is not None else np.zeros(0)) for k in LIBA_INPUTS}
    np.savez_compressed(OUT / f"{name}.npz", n_opt=w.n_opt, n_fixed_imu=w.n_fixed_imu, n_fixed=w.n_fixed, lambda_init=w.lambda_init,
                        max_iterations=w.max_iterations, huber=np.array([w.huber_mono, w.huber_stereo, w.huber_inertial]), **arrays,
                        exp_chi2_initial=tr["chi2_initial"], exp_chi2_trace=np.array(tr["chi2"]), exp_lambda_trace=np.array(tr["lam"]),
                        exp_trials_trace=np.array(tr["trials"]), exp_iterations=tr["iterations"], exp_Rwb=st.Rwb[:N], exp_twb=st.twb[:N],
                        exp_Rcw=st.Rcw[:N], exp_tcw=st.tcw[:N], exp_vel=st.vel[:N], exp_bg=st.bg[:N], exp_ba=st.ba[:N], exp_points=st.X)
    print(name, "edges", w.n_edges, "iters", tr["iterations"], "trials", tr["trials"], "chi2", tr["chi2_initial"], "->", tr["chi2"][-1])


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "pose":
    pose_fixture("pose_tiny", synth.make_pose_frame(401, n_points=70, mixed_mono_frac=0.4, outlier_frac=0.15))
    pose_fixture("pose_tiny_mono", synth.make_pose_frame(402, n_points=60, stereo=False, outlier_frac=0.1))
elif __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "liba":
    liba_fixture("liba_tiny", synth_inertial.make_inertial_window(5, n_opt=3, n_fixed=2, n_points=40))
    liba_fixture("liba_tiny_rig", synth_inertial.make_inertial_rig_window(7, n_opt=3, n_fixed=2, n_points=40))
elif __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "posei":
    posei_fixture("posei_tiny_keyframe", synth_inertial.make_posei_frame(9, mode=0, n_points=60))
    posei_fixture("posei_tiny_frame", synth_inertial.make_posei_frame(9, mode=1, n_points=60))
    posei_fixture("posei_tiny_rig", synth_inertial.make_posei_frame(10, mode=1, n_points=60, rig=True))
elif __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "rig":
    lba_fixture("lba_tiny_rig", synth.make_rig_window(71, n_free=3, n_fixed=2, n_points=40, track_len=(2, 5)))
elif __name__ == "__main__":
    lba_fixture("lba_tiny_mono", synth.make_window(11, n_free=3, n_fixed=2, n_points=40, stereo=False, track_len=(2, 5)))
    lba_fixture("lba_tiny_stereo", synth.make_window(12, n_free=3, n_fixed=2, n_points=40, stereo=True, track_len=(2, 5)))
    lba_fixture("lba_tiny_mixed", synth.make_window(13, n_free=4, n_fixed=1, n_points=30, stereo=True, track_len=(2, 5),
                                                     mixed_mono_frac=0.4, outlier_frac=0.1))
    # genuine LM rejections (large initial error, small user lambda): trials_trace > 1
    rej = dict(n_free=3, n_fixed=2, n_points=40, track_len=(2, 5), pose_noise=(0.08, 0.4), point_noise=1.5, lambda_init=1e-4)
    lba_fixture("lba_tiny_reject_stereo", synth.make_window(40, stereo=True, **rej))
    lba_fixture("lba_tiny_reject_mono", synth.make_window(51, stereo=False, **rej))
    # monocular KannalaBrandt8 (fisheye) window: the independent numpy model of oracle/lm_numpy.py (own Jacobian derivation)
    lba_fixture("lba_tiny_fisheye", synth.make_window(61, n_free=3, n_fixed=2, n_points=40, stereo=False, track_len=(2, 5), fisheye=True))
    # fisheye STEREO rig: left KannalaBrandt8 edges + right-camera body edges (EdgeSE3ProjectXYZToBody) sharing Hessian blocks
    lba_fixture("lba_tiny_rig", synth.make_rig_window(71, n_free=3, n_fixed=2, n_points=40, track_len=(2, 5)))
    orb_fixture()
    pose_fixture("pose_tiny", synth.make_pose_frame(401, n_points=70, mixed_mono_frac=0.4, outlier_frac=0.15))
    pose_fixture("pose_tiny_mono", synth.make_pose_frame(402, n_points=60, stereo=False, outlier_frac=0.1))
    liba_fixture("liba_tiny", synth_inertial.make_inertial_window(5, n_opt=3, n_fixed=2, n_points=40))
    liba_fixture("liba_tiny_rig", synth_inertial.make_inertial_rig_window(7, n_opt=3, n_fixed=2, n_points=40))
    posei_fixture("posei_tiny_keyframe", synth_inertial.make_posei_frame(9, mode=0, n_points=60))
    posei_fixture("posei_tiny_frame", synth_inertial.make_posei_frame(9, mode=1, n_points=60))
    posei_fixture("posei_tiny_rig", synth_inertial.make_posei_frame(10, mode=1, n_points=60, rig=True))
