"""Shared helpers for the parity tests."""
from pathlib import Path

import numpy as np

from orb_slam3_study_kr_amd import synth

GOLDEN = Path(__file__).resolve().parent / "golden"
LBA_FIXTURES = ["lba_tiny_mono", "lba_tiny_stereo", "lba_tiny_mixed", "lba_tiny_reject_stereo", "lba_tiny_reject_mono", "lba_tiny_fisheye", "lba_tiny_rig"]


def load_lba_fixture(name):
    z = np.load(GOLDEN / f"{name}.npz")
    w = synth.LbaWindow(
        n_free=int(z["n_free"]), n_fixed=int(z["n_fixed"]), pose_qt=z["pose_qt"], pose_cam=z["pose_cam"],
        points=z["points"], edge_pose=z["edge_pose"], edge_point=z["edge_point"], edge_kind=z["edge_kind"],
        edge_obs=z["edge_obs"], edge_info=z["edge_info"], huber_mono=float(z["huber_mono"]),
        huber_stereo=float(z["huber_stereo"]), lambda_init=float(z["lambda_init"]),
        max_iterations=int(z["max_iterations"]),
        kb8=(z["kb8"] if "kb8" in z.files and z["kb8"].size == 4 else None),
        cam2=(z["cam2"] if "cam2" in z.files and z["cam2"].size == 8 else None),
        trl=(z["trl"] if "trl" in z.files and z["trl"].size == 7 else None)).normalise()
    return w, z


def quat_to_R(q):
    return synth.quat_to_R(np.asarray(q) / np.linalg.norm(q))


def rel_translation_error(a_qt, b_qt):
    """max_i |t_a - t_b| / max(|t_b|, 1e-12)  -- the north_star tolerance is on SE3 translations."""
    ta, tb = np.asarray(a_qt)[:, 4:], np.asarray(b_qt)[:, 4:]
    return float(np.max(np.linalg.norm(ta - tb, axis=1) / np.maximum(np.linalg.norm(tb, axis=1), 1e-12)))


def rotation_error(a_qt, b_qt):
    out = 0.0
    for qa, qb in zip(np.asarray(a_qt)[:, :4], np.asarray(b_qt)[:, :4]):
        out = max(out, float(np.abs(quat_to_R(qa) - quat_to_R(qb)).max()))
    return out


def dense_blocks_from_H(H, b, w):
    """Split the dense (6P+3L)^2 system of oracle/lm_numpy.py into g2o's block layout."""
    P, L, E = w.n_free, w.n_points, w.n_edges
    Hpp = np.stack([H[6 * i:6 * i + 6, 6 * i:6 * i + 6] for i in range(P)]) if P else np.zeros((0, 6, 6))
    Hll = np.stack([H[6 * P + 3 * j:6 * P + 3 * j + 3, 6 * P + 3 * j:6 * P + 3 * j + 3] for j in range(L)])
    Hpl = np.zeros((E, 6, 3))
    for e in range(E):
        ip, il = w.edge_pose[e], w.edge_point[e]
        if ip < P:
            Hpl[e] = H[6 * ip:6 * ip + 6, 6 * P + 3 * il:6 * P + 3 * il + 3]
    return Hpp, b[:6 * P].reshape(P, 6), Hll, b[6 * P:].reshape(L, 3), Hpl


POSEI_FIXTURES = ["posei_tiny_keyframe", "posei_tiny_frame", "posei_tiny_rig"]


def load_posei_fixture(name):
    """A committed PoseInertialOptimization fixture (tests/golden/make_golden.py posei): the flat frame and the numpy model's outputs."""
    from orb_slam3_study_kr_amd import synth_inertial as si
    z = np.load(GOLDEN / f"{name}.npz")
    opt = lambda k: (z[k] if z[k].size else None)   # noqa: E731
    f = si.PoseiFrame(
        mode=int(z["mode"]), Rcw=z["Rcw"], tcw=z["tcw"], Rwb=z["Rwb"], twb=z["twb"], vel=z["vel"], bias_g=z["bias_g"], bias_a=z["bias_a"],
        prev_Rwb=z["prev_Rwb"], prev_twb=z["prev_twb"], prev_vel=z["prev_vel"], prev_bias_g=z["prev_bias_g"], prev_bias_a=z["prev_bias_a"],
        Rcb=z["Rcb"], tcb=z["tcb"], tbc=z["tbc"], cam=z["cam"], preint=z["preint"], info_inertial=z["info_inertial"], info_g=z["info_g"],
        info_a=z["info_a"], points=z["points"], edge_kind=z["edge_kind"], edge_obs=z["edge_obs"], edge_info=z["edge_info"],
        edge_close=z["edge_close"], prior_Rwb=opt("prior_Rwb"), prior_twb=opt("prior_twb"), prior_vel=opt("prior_vel"), prior_bg=opt("prior_bg"),
        prior_ba=opt("prior_ba"), prior_H=opt("prior_H"), kb8=opt("kb8"), cam2=opt("cam2"), trl=opt("trl"), rec_init=bool(int(z["rec_init"])),
        huber_mono=float(z["huber"][0]), huber_stereo=float(z["huber"][1]), huber_prior=float(z["huber"][2]),
        chi2_mono=tuple(float(x) for x in z["chi2_mono"]), chi2_stereo=tuple(float(x) for x in z["chi2_stereo"]),
        iterations=tuple(int(x) for x in z["iterations"])).normalise()
    return f, z


LIBA_FIXTURES = ["liba_tiny", "liba_tiny_rig"]


def load_liba_fixture(name):
    """A committed LocalInertialBA fixture (tests/golden/make_golden.py liba): the flat window and the numpy LM's outputs."""
    from orb_slam3_study_kr_amd import synth_inertial as si
    z = np.load(GOLDEN / f"{name}.npz")
    opt = lambda k: (z[k] if z[k].size else None)   # noqa: E731
    w = si.LibaWindow(
        n_opt=int(z["n_opt"]), n_fixed_imu=int(z["n_fixed_imu"]), n_fixed=int(z["n_fixed"]), pose_Rcw=z["pose_Rcw"], pose_tcw=z["pose_tcw"],
        pose_Rwb=z["pose_Rwb"], pose_twb=z["pose_twb"], Rcb=z["Rcb"], tcb=z["tcb"], tbc=z["tbc"], cam=z["cam"], vel=z["vel"], bias_g=z["bias_g"],
        bias_a=z["bias_a"], points=z["points"], edge_pose=z["edge_pose"], edge_point=z["edge_point"], edge_kind=z["edge_kind"], edge_obs=z["edge_obs"],
        edge_info=z["edge_info"], link_prev=z["link_prev"], link_cur=z["link_cur"], link_preint=z["link_preint"], link_info=z["link_info"],
        link_info_g=z["link_info_g"], link_info_a=z["link_info_a"], link_robust=z["link_robust"], huber_mono=float(z["huber"][0]),
        huber_stereo=float(z["huber"][1]), huber_inertial=float(z["huber"][2]), lambda_init=float(z["lambda_init"]),
        max_iterations=int(z["max_iterations"]), kb8=opt("kb8"), cam2=opt("cam2"), trl=opt("trl")).normalise()
    return w, z


def check_against_liba_fixture(got, z, fisheye):
    """A LocalInertialBA result against the numpy LM's committed outputs: LM trace, poses, velocities, biases, landmarks."""
    assert got.iterations == int(z["exp_iterations"])
    np.testing.assert_array_equal(got.trials_trace, z["exp_trials_trace"])
    np.testing.assert_allclose(got.chi2_initial, z["exp_chi2_initial"], rtol=1e-6)
    np.testing.assert_allclose(got.chi2_trace, z["exp_chi2_trace"], rtol=5e-6)
    # the lambda left behind by the LAST iteration is never used, and at convergence its gain ratio is a quotient of two 1e-6-relative
    # cost differences: compare the lambdas that steer an iteration
    np.testing.assert_allclose(got.lambda_trace[:-1], z["exp_lambda_trace"][:-1], rtol=1e-4)
    np.testing.assert_allclose(got.pose_Rwb, z["exp_Rwb"], atol=2e-7)
    np.testing.assert_allclose(got.pose_twb, z["exp_twb"], atol=2e-7 * max(1.0, np.abs(z["exp_twb"]).max()))
    np.testing.assert_allclose(got.pose_Rcw, z["exp_Rcw"], atol=2e-7)
    np.testing.assert_allclose(got.vel, z["exp_vel"], atol=5e-7)
    np.testing.assert_allclose(got.bias_g, z["exp_bg"], atol=1e-7)
    np.testing.assert_allclose(got.bias_a, z["exp_ba"], atol=1e-6)
    # weakly observed depths of a monocular / fisheye window move with the 1e-8 differences of the float32 preintegration getters
    np.testing.assert_allclose(got.points, z["exp_points"], atol=2e-4 if fisheye else 1e-5)


POSE_FIXTURES = ["pose_tiny", "pose_tiny_mono"]


def load_pose_fixture(name):
    """A committed PoseOptimization fixture (tests/golden/make_golden.py pose): the flat frame and the numpy model's outputs."""
    from orb_slam3_study_kr_amd import synth
    z = np.load(GOLDEN / f"{name}.npz")
    f = synth.PoseFrame(pose_qt=z["pose_qt"], cam=z["cam"], points=z["points"], edge_kind=z["edge_kind"], edge_obs=z["edge_obs"], edge_info=z["edge_info"],
                        huber_mono=float(z["huber"][0]), huber_stereo=float(z["huber"][1]), chi2_mono=tuple(z["chi2_mono"]), chi2_stereo=tuple(z["chi2_stereo"]),
                        iterations=tuple(int(k) for k in z["iterations"])).normalise()
    return f, z


def check_against_pose_fixture(got, z):
    """A PoseOptimization result (pose_qt, outlier, n_bad, chi2_final, edge_chi2) against the numpy model's committed outputs."""
    from oracle import lm_numpy as lm
    T = np.eye(4)
    T[:3, :3] = lm.quat_to_R(np.asarray(got.pose_qt[:4], dtype=np.float64))
    T[:3, 3] = got.pose_qt[4:]
    np.testing.assert_allclose(T, z["exp_T"], atol=5e-8)
    np.testing.assert_array_equal(got.outlier, z["exp_outlier"])
    assert got.n_bad == int(z["exp_n_bad"]) and got.rounds == int(z["exp_rounds"])
    np.testing.assert_allclose(got.chi2_final, z["exp_chi2_final"], rtol=1e-6)
    inl = z["exp_outlier"] == 0
    np.testing.assert_allclose(got.edge_chi2[inl], z["exp_edge_chi2"][inl], rtol=1e-5, atol=1e-7)
