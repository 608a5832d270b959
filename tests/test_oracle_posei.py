"""CPU tests of the restatement of Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame (oracle/liba_oracle.c, second half),
pinned by the independent numpy model oracle/liba_numpy.py (central-difference Jacobians of every residual block, numpy float32
preintegration getters).  Parity unpinned against a reference binary (see the oracle header)."""
import numpy as np
import pytest

from oracle import binding as ob
from oracle import liba_numpy as ln
from orb_slam3_study_kr_amd import synth_inertial as si
from helpers import POSEI_FIXTURES, load_posei_fixture


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("kw", [dict(), dict(stereo=False), dict(fisheye=True), dict(rig=True)], ids=["stereo", "mono", "fisheye", "rig"])
def test_first_gauss_newton_system_vs_numeric_jacobians(mode, kw):
    f = si.make_posei_frame(5, mode=mode, n_points=60, **kw)
    H, b = ob.posei_linearize(f)
    Hn, bn = ln.posei_numeric_system(ln.PoseiState(f))
    assert np.abs(H - Hn).max() < 2e-6 * np.abs(Hn).max()
    assert np.abs(b - bn).max() < 2e-6 * np.abs(bn).max()
    np.testing.assert_allclose(H, H.T, rtol=0, atol=1e-9 * np.abs(H).max())


def test_last_keyframe_variant_converges_and_classifies_outliers():
    f = si.make_posei_frame(3, mode=0, n_points=400)
    r = ob.posei_optimize(f)
    assert r.rounds == 4 and r.status == 0
    assert np.abs(r.twb - f.gt["twb"]).max() < 2e-3 and np.abs(f.twb - f.gt["twb"]).max() > 1e-2
    assert np.abs(r.vel - f.gt["vel"]).max() < 5e-3
    # every gross outlier is classified as one; camera pose stays the body pose seen through T_cb
    assert r.outlier[f.gt["outliers"]].mean() > 0.9 and r.n_bad == int(r.outlier.sum()) and r.n_inliers == f.n_edges - r.n_bad
    Rcb = f.Rcb.reshape(3, 3)
    np.testing.assert_allclose(r.Rcw, Rcb @ r.Rwb.T, atol=1e-12)
    np.testing.assert_allclose(r.tcw, Rcb @ (-r.Rwb.T @ r.twb) + f.tcb, atol=1e-12)
    # the Hessian handed to ConstraintPoseImu: EdgeInertial::GetHessian2 + random walks + the inlier visual edges, at the final state
    assert r.H.shape == (15, 15) and np.linalg.eigvalsh((r.H + r.H.T) / 2).min() > 0
    np.testing.assert_allclose(r.H[9:12, 9:12], f.info_g.reshape(3, 3), rtol=1e-12)
    np.testing.assert_allclose(r.H[12:15, 12:15], f.info_a.reshape(3, 3), rtol=1e-12)
    assert np.all(r.H[:9, 9:] == 0)


def test_last_frame_variant_prior_and_marginalisation():
    f = si.make_posei_frame(4, mode=1, n_points=300)
    r = ob.posei_optimize(f)
    assert r.rounds == 4 and r.H.shape == (30, 30)
    assert np.abs(r.twb - f.gt["twb"]).max() < np.abs(f.twb - f.gt["twb"]).max()
    # Optimizer::Marginalize(H, 0, 14).block<15,15>(15,15) == Schur complement with the pseudo-inverse of the previous frame's block
    Hs = (r.H + r.H.T) / 2
    ref = Hs[15:, 15:] - Hs[15:, :15] @ np.linalg.pinv(Hs[:15, :15], rcond=0, hermitian=True) @ Hs[:15, 15:]
    got = ob.marginalize_previous(r.H)
    np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-9 * np.abs(ref).max())
    # ConstraintPoseImu: symmetrised, eigenvalues below 1e-12 dropped
    A = np.diag([5.0, 2.0, 1e-13] + [1.0] * 12)
    Q, _ = np.linalg.qr(np.random.default_rng(0).standard_normal((15, 15)))
    M = Q @ A @ Q.T
    C = ob.constraint_pose_imu_H(M)
    w = np.linalg.eigvalsh(C)
    assert abs(w[0]) < 1e-14 and np.allclose(np.sort(w)[1:], np.sort(np.diag(A))[1:], rtol=1e-10)


def test_few_inliers_recovery_pass_and_rec_init():
    # 25 map points: fewer than 30 inliers -> the recovery pass re-admits every edge under 18 / 24 (src/Optimizer.cc:4821-4848)
    f = si.make_posei_frame(6, mode=0, n_points=25, outlier_frac=0.3)
    r = ob.posei_optimize(f)
    assert r.n_inliers < 30 and r.n_bad == int((r.edge_chi2 >= np.where(f.edge_kind == 1, 24.0, 18.0)).sum())
    f2 = si.make_posei_frame(6, mode=0, n_points=25, outlier_frac=0.3, rec_init=True)
    r2 = ob.posei_optimize(f2)
    assert r2.n_bad == int(r2.outlier.sum()) and r2.n_bad >= r.n_bad
    # fewer than 10 graph edges: one round only (:4814-4817)
    f3 = si.make_posei_frame(7, mode=0, n_points=5, outlier_frac=0.0)
    assert ob.posei_optimize(f3).rounds == 1


def check_against_posei_fixture(got, z, f, state_tol=1e-7, chi_tol=2e-5):
    """Outputs of a PoseInertialOptimization run against the committed outputs of the independent numpy model (numeric Jacobians,
    numpy.linalg.solve): final state, per-edge chi2, identical classification and the ConstraintPoseImu Hessian."""
    assert got.rounds == int(z["exp_rounds"])
    np.testing.assert_allclose(got.Rwb, z["exp_Rwb"], atol=state_tol)
    np.testing.assert_allclose(got.twb, z["exp_twb"], atol=state_tol * max(1.0, np.abs(z["exp_twb"]).max()))
    np.testing.assert_allclose(got.vel, z["exp_vel"], atol=state_tol * 10)
    np.testing.assert_allclose(got.bias_g, z["exp_bias_g"], atol=state_tol / 10)
    np.testing.assert_allclose(got.bias_a, z["exp_bias_a"], atol=state_tol)
    np.testing.assert_allclose(got.edge_chi2, z["exp_edge_chi2"], rtol=chi_tol, atol=chi_tol)
    thr = np.where(f.edge_kind == 1, f.chi2_stereo[3], np.where(f.edge_close == 1, 1.5 * f.chi2_mono[3], f.chi2_mono[3]))
    near = np.abs(z["exp_edge_chi2"] - thr) < 10 * chi_tol * thr
    np.testing.assert_array_equal(got.outlier[~near], z["exp_outlier"][~near])
    if not near.any():
        assert (got.n_bad, got.n_inliers) == (int(z["exp_n_bad"]), int(z["exp_n_inliers"]))
    np.testing.assert_allclose(got.H, z["exp_H"], rtol=0, atol=2e-6 * np.abs(z["exp_H"]).max())


@pytest.mark.parametrize("name", POSEI_FIXTURES)
def test_oracle_matches_the_numpy_models_golden_outputs(name):
    f, z = load_posei_fixture(name)
    fish = f.kb8 is not None      # float32 theta / psi staircase: the numpy Jacobians differentiate the smooth projection
    check_against_posei_fixture(ob.posei_optimize(f), z, f, state_tol=2e-6 if fish else 1e-7, chi_tol=2e-3 if fish else 2e-5)
