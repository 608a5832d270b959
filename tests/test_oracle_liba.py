"""CPU tests of the stereo-inertial oracle (oracle/liba_oracle.c), pinned by the independent numpy check
oracle/liba_numpy.py (numeric Jacobians, real SVD, numpy float32 preintegration getters).  Parity unpinned
against a reference binary (see the oracle header)."""
import numpy as np
import pytest

from oracle import binding as ob
from oracle import liba_numpy as ln
from orb_slam3_study_kr_amd import synth_inertial as si


@pytest.fixture(scope="module")
def tiny():
    return si.make_inertial_window(5, n_opt=3, n_fixed=2, n_points=60)


def test_so3_helpers_match_numpy():
    rng = np.random.default_rng(0)
    for scale in (1e-7, 1e-3, 0.3, 2.0):
        for _ in range(10):
            w = rng.standard_normal(3) * scale
            R = ob.exp_so3(w)
            np.testing.assert_allclose(R, ln.exp_so3(w), atol=1e-14)
            np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-14)
            np.testing.assert_allclose(ob.log_so3(R), ln.log_so3(R), atol=1e-14)
            if 1e-4 < scale < 1:   # |w| < pi so the log is the principal value
                np.testing.assert_allclose(ob.log_so3(R), w, atol=1e-9)


def test_preintegration_recovers_noise_free_motion():
    w = si.make_inertial_window(6, n_opt=3, n_fixed=2, n_points=60, imu_noise=False)
    # with the linearisation bias ~ true bias and no noise, dP/dV/dR predict the ground-truth relative motion
    gt = w.gt
    for l in range(w.n_links):
        a, c = int(w.link_prev[l]), int(w.link_cur[l])
        rec = w.link_preint[l]
        dT = float(rec[0])
        R1 = gt["Rwb"][a]
        dV = R1.T @ (gt["vel"][c] - gt["vel"][a] - ln.G * dT)
        dP = R1.T @ (gt["twb"][c] - gt["twb"][a] - gt["vel"][a] * dT - ln.G * dT * dT / 2)
        assert abs(dT - 0.25) < 1e-6
        np.testing.assert_allclose(rec[10:13], dV, atol=2e-2)     # bias linearisation offset + Euler integration
        np.testing.assert_allclose(rec[13:16], dP, atol=5e-3)
        np.testing.assert_allclose(rec[1:10].reshape(3, 3), R1.T @ gt["Rwb"][c], atol=2e-3)


def test_inertial_residual_and_jacobians_vs_numeric(tiny):
    st = ln.State(tiny)
    N = tiny.n_opt
    for l in range(tiny.n_links):
        r, J = ob.liba_inertial_edge(tiny, l)
        np.testing.assert_allclose(r, ln.inertial_residual(st, l), rtol=0, atol=2e-6)   # float32 SVD vs polar Newton
        a, c = int(tiny.link_prev[l]), int(tiny.link_cur[l])
        # numeric Jacobian wrt the optimisable vertices of the link (columns P1 V1 G1 A1 P2 V2 of the oracle)
        cols = {0: (6 * a, 6) if a < N else None, 6: (6 * N + 9 * a, 3) if a < N else None,
                9: (6 * N + 9 * a + 3, 3) if a < N else None, 12: (6 * N + 9 * a + 6, 3) if a < N else None,
                15: (6 * c, 6), 21: (6 * N + 9 * c, 3)}
        for col, tgt in cols.items():
            if tgt is None:
                continue
            off, dim = tgt
            is_bias = col in (9, 12)
            d = 2e-3 if is_bias else 1e-6
            for k in range(dim):
                e = np.zeros(15 * N + 3 * tiny.n_points); e[off + k] = d
                sp, sm = st.copy(), st.copy()
                sp.oplus(e); sm.oplus(-e)
                num = (ln.inertial_residual(sp, l) - ln.inertial_residual(sm, l)) / (2 * d)
                tol = 2e-3 if is_bias else 2e-5
                np.testing.assert_allclose(J[:, col + k], num, rtol=tol, atol=tol * max(1.0, np.abs(num).max()))


def test_linearisation_matches_numeric_dense_system(tiny):
    lin = ob.liba_linearize(tiny)
    st = ln.State(tiny)
    # float32 NormalizeRotation (real SVD here, polar Newton in the oracle) differs by one float ulp in dR; the
    # inertial information is ~1e8, so chi2 agrees to 1e-7 rather than 1e-12
    np.testing.assert_allclose(lin["chi2"], ln.robust_chi2(st), rtol=1e-6)
    Hn, bn = ln.numeric_dense_system(st)
    N, L = tiny.n_opt, tiny.n_points
    n = 15 * N
    # assemble the oracle's blocks into the same dense layout
    H = np.zeros_like(Hn)
    H[:n, :n] = lin["H"]
    for j in range(L):
        H[n + 3 * j:n + 3 * j + 3, n + 3 * j:n + 3 * j + 3] = lin["Hll"][j]
    for e in range(tiny.n_edges):
        k, j = int(tiny.edge_pose[e]), int(tiny.edge_point[e])
        if k < N:
            H[6 * k:6 * k + 6, n + 3 * j:n + 3 * j + 3] += lin["Hpl"][e]
            H[n + 3 * j:n + 3 * j + 3, 6 * k:6 * k + 6] += lin["Hpl"][e].T
    scale = np.abs(Hn).max()
    assert np.abs(H - Hn).max() < 2e-4 * scale
    assert np.abs(lin["b"] - bn).max() < 2e-4 * np.abs(bn).max()


def test_shared_bias_pair_and_its_priors_vs_numeric_dense_system():
    """FullInertialBA with bInit (src/Optimizer.cc:452-462,514-518,581-601): every EdgeInertial on one (gyro, acc) bias pair, no random
    walks, EdgePriorGyro / EdgePriorAcc written as the random-walk edges of a link without inertial information from a fixed keyframe that
    holds the prior value (osh_liba_problem.link_bias).  The C restatement against the numpy model's cost and numeric Jacobians."""
    from orb_slam3_study_kr_amd import synth_inertial as si
    w = si.with_shared_bias(si.make_inertial_window(5, n_opt=3, n_fixed=2, n_points=25, outlier_frac=0.0), prior_g=1e2, prior_a=1e6)
    lin, st = ob.liba_linearize(w), ln.State(w)
    np.testing.assert_allclose(lin["chi2"], ln.robust_chi2(st), rtol=1e-6)
    prior = float(1e2 * st.bg[0] @ st.bg[0] + 1e6 * st.ba[0] @ st.ba[0])          # (bprior - b)^T info (bprior - b), bprior = 0
    plain = si.make_inertial_window(5, n_opt=3, n_fixed=2, n_points=25, outlier_frac=0.0)
    assert prior > 100.0 and lin["chi2"] > prior
    Hn, bn = ln.numeric_dense_system(st)
    n = 15 * w.n_opt
    assert np.abs(lin["H"] - Hn[:n, :n]).max() < 2e-4 * np.abs(Hn).max()
    assert np.abs(lin["b"][:n] - bn[:n]).max() < 2e-4 * np.abs(bn).max()
    o = 6 * w.n_opt + 9 * 0 + 3                                                   # the shared pair sits with the oldest keyframe
    for k in (1, 2):                                                              # the other keyframes' bias slots take part in nothing
        oo = 6 * w.n_opt + 9 * k + 3
        assert not lin["H"][oo:oo + 6].any() and not lin["b"][oo:oo + 6].any()
    assert np.abs(lin["H"][o:o + 6, :6 * w.n_opt]).max() > 0                      # the pair couples with the poses of every link
    r = ob.liba_solve(w)
    assert r.chi2_final < 0.05 * r.chi2_initial and np.abs(r.bias_a[0]).max() < 1e-4      # information 1e6 pulls the acc bias to the prior
    np.testing.assert_array_equal(r.bias_g[1:], np.asarray(w.bias_g).reshape(-1, 3)[1:3])
    assert plain.n_links == w.n_links


def test_full_inertial_lm_converges_and_is_consistent():
    w = si.make_inertial_window(11)
    r = ob.liba_solve(w)
    assert 0 < r.iterations <= 10 and r.chi2_final < 0.02 * r.chi2_initial
    N = w.n_opt
    assert np.abs(r.pose_twb - w.gt["twb"][:N]).max() < 0.01
    assert np.abs(r.vel - w.gt["vel"][:N]).max() < 0.02
    # camera pose stays the body pose seen through T_cb (ImuCamPose::Update)
    Rcb = w.Rcb.reshape(3, 3)
    for k in range(N):
        np.testing.assert_allclose(r.pose_Rcw[k], Rcb @ r.pose_Rwb[k].T, atol=1e-12)
        np.testing.assert_allclose(r.pose_tcw[k], Rcb @ (-r.pose_Rwb[k].T @ r.pose_twb[k]) + w.tcb, atol=1e-12)
    # bLarge variant: lambda 1e-2, 4 iterations
    wl = si.make_inertial_window(12, large=True)
    rl = ob.liba_solve(wl)
    assert rl.iterations <= 4 and abs(rl.lambda_trace[0] - 1e-2 / 3) < 1e-12


def test_fisheye_rig_right_camera_edges_vs_numeric_dense_system():
    """LocalInertialBA on a fisheye stereo rig (src/Optimizer.cc:2798-2835): EdgeMono(1) edges on camera 1 of ImuCamPose, alone or
    sharing their (keyframe, landmark) Hessian block with the left EdgeMono(0).  The oracle's analytic blocks against central
    differences of the numpy model (which forms camera 1 as Trl * Tc0w)."""
    w = si.make_inertial_rig_window(7, n_opt=3, n_fixed=2, n_points=60, right_frac=0.5, right_only_frac=0.2)
    kinds = w.edge_kind
    pairs = np.sum((kinds[1:] == 2) & (kinds[:-1] == 0) & (w.edge_pose[1:] == w.edge_pose[:-1]) & (w.edge_point[1:] == w.edge_point[:-1]))
    assert pairs > 20 and np.sum(kinds == 2) > pairs + 5
    lin = ob.liba_linearize(w)
    st = ln.State(w)
    np.testing.assert_allclose(lin["chi2"], ln.robust_chi2(st), rtol=1e-6)
    Hn, bn = ln.numeric_dense_system(st)
    N, L = w.n_opt, w.n_points
    n = 15 * N
    H = np.zeros_like(Hn)
    H[:n, :n] = lin["H"]
    for j in range(L):
        H[n + 3 * j:n + 3 * j + 3, n + 3 * j:n + 3 * j + 3] = lin["Hll"][j]
    for e in range(w.n_edges):
        k, j = int(w.edge_pose[e]), int(w.edge_point[e])
        if k < N:   # the second edge of a pair leaves its own slot zero: its block was added to the first's
            H[6 * k:6 * k + 6, n + 3 * j:n + 3 * j + 3] += lin["Hpl"][e]
            H[n + 3 * j:n + 3 * j + 3, 6 * k:6 * k + 6] += lin["Hpl"][e].T
    assert np.abs(H - Hn).max() < 2e-4 * np.abs(Hn).max()
    assert np.abs(lin["b"] - bn).max() < 2e-4 * np.abs(bn).max()
    # the full LM loop converges and keeps both cameras consistent with the body pose
    wf = si.make_inertial_rig_window(13)
    r = ob.liba_solve(wf)
    assert 0 < r.iterations <= 10 and r.chi2_final < 0.05 * r.chi2_initial
    assert np.abs(r.pose_twb - wf.gt["twb"][:wf.n_opt]).max() < 0.01


@pytest.mark.parametrize("name", ["liba_tiny", "liba_tiny_rig"])
def test_oracle_matches_the_numpy_lm_golden_outputs(name):
    """tests/golden/liba_tiny*.npz: the whole optimize() of LocalInertialBA by the independent numpy Levenberg-Marquardt (full dense
    system, central-difference Jacobians).  Same iteration / trial trace, cost trace to 5e-6, states to 2e-7."""
    from helpers import check_against_liba_fixture, load_liba_fixture
    w, z = load_liba_fixture(name)
    check_against_liba_fixture(ob.liba_solve(w), z, fisheye=w.kb8 is not None)
