"""GPU parity tests of the local-BA hot path: HIP kernels (through the C-ABI) vs the CPU oracle
and the committed golden fixtures.  Tolerances: SE3 translations <= 1e-6 relative (north_star);
assembled blocks <= 1e-11 relative (only re-association / FMA contraction differ)."""
import numpy as np
import pytest

from helpers import LBA_FIXTURES, dense_blocks_from_H, load_lba_fixture, quat_to_R, rel_translation_error, rotation_error
from orb_slam3_study_kr_amd import lba, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver(hip_lib):
    with lba.LbaSolver(0) as s:
        yield s


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    return binding


def _assert_blocks(got, exp, tol=1e-11):
    for k in ("Hpp", "bp", "Hll", "bl", "Hpl"):
        scale = max(np.abs(exp[k]).max(), 1e-300)
        assert np.abs(got[k] - exp[k]).max() <= tol * scale, (k, np.abs(got[k] - exp[k]).max() / scale)
    np.testing.assert_allclose(got["chi2"], exp["chi2"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(got["robust_chi2"], exp["robust_chi2"], rtol=1e-12)


@pytest.mark.parametrize("name", LBA_FIXTURES)
def test_linearisation_matches_oracle_and_golden(solver, ob, name):
    w, z = load_lba_fixture(name)
    solver.upload([w])
    got = solver.linearize(0)
    # fisheye Jacobians divide by rho^3 and rho^2 (rho^2 + z^2): 1e-10 there (as in test_fisheye_monocular_window_kannala_brandt8)
    _assert_blocks(got, ob.lba_linearize(w), tol=1e-10 if w.kb8 is not None else 1e-11)
    Hpp, bp, Hll, bl, Hpl = dense_blocks_from_H(z["exp_H"], z["exp_b"], w)
    scale = np.abs(z["exp_H"]).max()
    np.testing.assert_allclose(got["Hpp"], Hpp, rtol=1e-9, atol=1e-9 * scale)
    np.testing.assert_allclose(got["Hll"], Hll, rtol=1e-9, atol=1e-9 * scale)
    np.testing.assert_allclose(got["Hpl"], Hpl, rtol=1e-9, atol=1e-9 * scale)
    np.testing.assert_allclose(got["robust_chi2"], float(z["exp_chi2_initial"]), rtol=1e-12)


@pytest.mark.parametrize("name", LBA_FIXTURES)
def test_one_trial_schur_and_solution_match_oracle(solver, ob, name):
    w, z = load_lba_fixture(name)
    lam = float(z["exp_lambda0"])
    solver.upload([w])
    S, bs, x = solver.debug_trial(0, lam)
    So, bso, xo = ob.lba_schur_step(w, lam)
    iu = np.triu_indices(S.shape[0])
    np.testing.assert_allclose(S[iu], So[iu], rtol=1e-10, atol=1e-11 * np.abs(So).max())
    np.testing.assert_allclose(bs, bso, rtol=1e-10, atol=1e-11 * np.abs(bso).max())
    np.testing.assert_allclose(x, xo, rtol=1e-7, atol=1e-9 * np.abs(xo).max())
    np.testing.assert_allclose(x, z["exp_x0"], rtol=1e-6, atol=1e-8 * np.abs(z["exp_x0"]).max())


def _check_result(got, ref, w, t_tol=1e-6, chi_tol=1e-7, pts_tol=1e-6, lam_tol=None):
    assert got.iterations == ref.iterations
    np.testing.assert_array_equal(got.trials_trace, ref.trials_trace)
    np.testing.assert_allclose(got.chi2_initial, ref.chi2_initial, rtol=1e-11)
    np.testing.assert_allclose(got.chi2_trace, ref.chi2_trace, rtol=chi_tol)
    np.testing.assert_allclose(got.lambda_trace, ref.lambda_trace, rtol=lam_tol or max(1e-6, 10 * chi_tol))
    assert rel_translation_error(got.pose_qt, ref.pose_qt) < t_tol
    assert rotation_error(got.pose_qt, ref.pose_qt) < 1e-6
    np.testing.assert_allclose(got.points, ref.points, rtol=pts_tol, atol=pts_tol)
    np.testing.assert_allclose(got.edge_chi2, ref.edge_chi2, rtol=max(1e-5, 100 * pts_tol), atol=max(1e-6, 100 * pts_tol))
    # identical outlier decisions (Optimizer.cc:1413-1460) except within rounding of the threshold
    thr = np.where(w.edge_kind == 1, synth.CHI2_STEREO, synth.CHI2_MONO)   # the body edge is a two-row edge like the mono one
    near = np.abs(ref.edge_chi2 - thr) < max(1e-5, 100 * pts_tol) * thr
    a = (got.edge_chi2 > thr) | (got.edge_depth_pos == 0)
    b = (ref.edge_chi2 > thr) | (ref.edge_depth_pos == 0)
    assert np.array_equal(a[~near], b[~near])


@pytest.mark.parametrize("name", LBA_FIXTURES)
def test_full_lm_matches_oracle_and_golden(solver, ob, name):
    w, z = load_lba_fixture(name)
    got = solver.solve([w])[0]
    if w.cam2 is not None:
        # fisheye rig: float32 theta / psi of both cameras put a staircase into the residual; the C oracle, the numpy model and
        # the device each round slightly differently on it (on this fixture the device's cost trace equals the numpy model's
        # to 1e-9 and the C oracle's to 2.5e-7); translations / rotations are still held to 1e-6
        _check_result(got, ob.lba_solve(w), w, chi_tol=2e-6, pts_tol=1e-4)
    else:
        _check_result(got, ob.lba_solve(w), w)
    # and the independent numpy implementation the fixture was generated with
    assert got.iterations == int(z["exp_iterations"])
    np.testing.assert_array_equal(got.trials_trace, z["exp_trials_trace"])
    T = z["exp_T"]
    t_rel = np.max(np.linalg.norm(got.pose_qt[:, 4:] - T[:, :3, 3], axis=1) / np.linalg.norm(T[:, :3, 3], axis=1))
    assert t_rel < 1e-6
    for i in range(w.n_free):
        np.testing.assert_allclose(quat_to_R(got.pose_qt[i, :4]), T[i, :3, :3], atol=1e-6)


def test_config1_mono_plumbing_graph(solver, ob):
    w = synth.make_config1(1)
    _check_result(solver.solve([w])[0], ob.lba_solve(w), w)


def test_config2_stereo_window_full_size(solver, ob):
    """BASELINE.json configs[1]: 50 free + 10 fixed KF, ~10k landmarks, ~75k stereo edges."""
    w = synth.make_config2(100)
    got = solver.solve([w])[0]
    _check_result(got, ob.lba_solve(w), w)


def test_window_beyond_the_lds_factorisation_uses_the_global_memory_solver(solver, ob):
    """A reduced system of 300 optimisable keyframes (n = 1800) does not fit the LDS-resident LDL^T of k_solve: it is factored in
    global memory by big_solve.h (global BA of a long session, src/Optimizer.cc:53-392).  Same LM trace as the oracle's dense solve;
    a small window in the same batch takes the same path."""
    big = synth.make_window(900, n_free=300, n_fixed=3, n_points=6000, stereo=True, max_iterations=5)
    small = synth.make_window(901, n_free=6, n_fixed=2, n_points=300, stereo=True)
    got = solver.solve([big, small])
    _check_result(got[0], ob.lba_solve(big), big, t_tol=1e-5, chi_tol=1e-6, pts_tol=1e-5)
    _check_result(got[1], ob.lba_solve(small), small)


def test_batch_of_heterogeneous_windows_equals_single_solves(solver, ob):
    ws = [synth.make_window(200 + i, n_free=4 + 3 * i, n_fixed=1 + i, n_points=150 + 90 * i, stereo=bool(i % 2),
                            mixed_mono_frac=0.3 if i == 3 else 0.0, lambda_init=[0.0, 100.0, 0.0, 1e-3][i],
                            max_iterations=[10, 5, 3, 10][i]) for i in range(4)]
    got = solver.solve(ws)
    for w, g in zip(ws, got):
        _check_result(g, ob.lba_solve(w), w)
    # repeated optimize() on the resident batch is idempotent (resets to the uploaded estimates)
    solver.optimize()
    again = solver.download()
    for g, a in zip(got, again):
        np.testing.assert_array_equal(g.pose_qt, a.pose_qt)
        np.testing.assert_array_equal(g.points, a.points)


def test_rejections_and_early_termination_paths(solver, ob):
    ws = [synth.make_window(s, n_free=5, n_fixed=2, n_points=120, stereo=(s % 2 == 0), track_len=(2, 6),
                            pose_noise=(0.08, 0.4), point_noise=1.5, lambda_init=1e-4) for s in range(60, 68)]
    refs = [ob.lba_solve(w) for w in ws]
    assert any((r.trials_trace > 1).any() for r in refs), "fixture set should exercise rejected trials"
    for w, g, r in zip(ws, solver.solve(ws), refs):
        # badly conditioned ON PURPOSE (120 points, user lambda 1e-4, 0.4 m / 1.5 m initial error, up to 7
        # rejected trials per iteration): 1e-16 re-association differences are amplified by cond(S) through
        # ten nonlinear iterations, so this stress case is held to 2e-5; every realistic window (configs 1, 2,
        # the golden fixtures incl. the two rejection fixtures) is held to the north-star 1e-6 above.
        # (The two independent CPU implementations, oracle/lba_oracle.c and oracle/lm_numpy.py, disagree by
        # the same amount on these windows: seed 64 -> chi2 2e-6, points 3.5e-4, translations 8e-7.)
        # lambda is the most sensitive trace here: its update is 1 - (2 rho - 1)^3 with rho a ratio of two small numbers
        _check_result(g, r, w, t_tol=2e-5, chi_tol=2e-5, pts_tol=2e-3, lam_tol=1e-3)


def test_stop_flag_zero_iterations_and_empty_cases(solver, ob):
    w = synth.make_window(5, n_free=3, n_fixed=2, n_points=40, track_len=(2, 5))
    w.stop_flag = np.ones(1, dtype=np.uint8)
    g = solver.solve([w])[0]
    assert g.iterations == 0 and g.trials == 0
    np.testing.assert_allclose(g.points, w.points)
    np.testing.assert_allclose(g.pose_qt[:, 4:], w.pose_qt[:w.n_free, 4:])
    w.stop_flag = None
    w.max_iterations = 0
    assert solver.solve([w])[0].iterations == 0
    # landmark observed only by fixed keyframes; pose without any edge is rejected by a zero pivot -> 10 failed trials
    w2 = synth.make_window(6, n_free=3, n_fixed=2, n_points=40, track_len=(2, 5))
    fixed_seen = np.unique(w2.edge_point[w2.edge_pose >= w2.n_free])
    j = int(fixed_seen[0])
    keep = ~((w2.edge_point == j) & (w2.edge_pose < w2.n_free))   # landmark j is now seen by fixed keyframes only
    for name in ("edge_pose", "edge_point", "edge_kind", "edge_obs", "edge_info"):
        setattr(w2, name, np.ascontiguousarray(getattr(w2, name)[keep]))
    assert not np.any((w2.edge_point == j) & (w2.edge_pose < w2.n_free))
    _check_result(solver.solve([w2])[0], ob.lba_solve(w2), w2)


def test_duplicate_pose_landmark_edge_is_reported_unsupported(solver):
    from orb_slam3_study_kr_amd import capi
    w = synth.make_window(8, n_free=3, n_fixed=2, n_points=40, track_len=(2, 5))
    w.edge_pose[1] = w.edge_pose[0]
    w.edge_point[1] = w.edge_point[0]
    with pytest.raises(capi.OshError) as ei:
        solver.upload([w])
    assert ei.value.code == capi.OSH_ERR_UNSUPPORTED


def test_zero_noise_window_converges_to_ground_truth(solver):
    w = synth.make_window(77, n_free=6, n_fixed=3, n_points=300, stereo=True, pixel_noise=False, outlier_frac=0.0,
                          max_iterations=40)
    g = solver.solve([w])[0]
    assert rel_translation_error(g.pose_qt, w.gt_pose_qt[:w.n_free]) < 1e-5
    assert np.abs(g.points - w.gt_points).max() < 1e-3


@pytest.mark.parametrize("kwargs", [
    dict(seed=31, n_free=12, n_fixed=3, n_points=800, track_len=(2, 9), obs_dropout=0.2),     # ragged observer sets
    dict(seed=32, n_free=30, n_fixed=4, n_points=1500, track_len=(10, 28)),                   # > 8 and > 16 observers: cross items
    dict(seed=33, n_free=30, n_fixed=4, n_points=1500, track_len=(10, 28), obs_dropout=0.3, stereo=False),
])
def test_observer_set_grouping_ragged_and_long_tracks(solver, ob, kwargs):
    """The Schur work plan (csrc/schur_plan.h) on graphs whose landmarks do not share observer sets:
    one trial (S, b_s, x) and the full optimisation against the oracle."""
    kw = dict(kwargs)
    w = synth.make_window(kw.pop("seed"), **kw)
    solver.upload([w])
    S, bs, x = solver.debug_trial(0, 1e-3)
    So, bso, xo = ob.lba_schur_step(w, 1e-3)
    iu = np.triu_indices(S.shape[0])
    np.testing.assert_allclose(S[iu], So[iu], rtol=1e-10, atol=1e-11 * np.abs(So).max())
    np.testing.assert_allclose(bs, bso, rtol=1e-10, atol=1e-11 * np.abs(bso).max())
    np.testing.assert_allclose(x, xo, rtol=1e-7, atol=1e-9 * np.abs(xo).max())
    _check_result(solver.solve([w])[0], ob.lba_solve(w), w)


def test_randomised_small_windows_in_one_batch(solver, ob):
    """Forty random windows of assorted shapes in one batch (1-9 optimisable poses, 0-4 fixed, 5-400 landmarks, mono / stereo /
    mixed, ragged observer sets, short iteration limits): every one against the oracle.  Exercises items without row poses,
    landmarks seen by fixed keyframes only, single-pose windows and windows without any fixed keyframe."""
    rng = np.random.Generator(np.random.PCG64(2024))
    ws = []
    for k in range(40):
        n_free = int(rng.integers(1, 10))
        n_fixed = int(rng.integers(0, 5)) if n_free > 1 else int(rng.integers(1, 4))
        w = synth.make_window(900 + k, n_free=n_free, n_fixed=n_fixed, n_points=int(rng.integers(5, 400)), stereo=bool(rng.integers(0, 2)),
                              mixed_mono_frac=float(rng.choice([0.0, 0.3])), track_len=(2, int(rng.integers(3, 14))),
                              obs_dropout=float(rng.choice([0.0, 0.15, 0.35])), max_iterations=int(rng.integers(1, 11)),
                              lambda_init=float(rng.choice([0.0, 1e-3, 10.0])))
        if w.n_edges == 0 or w.n_points == 0:
            continue
        ws.append(w)
    assert len(ws) > 30
    got = solver.solve(ws)
    worst = 0.0
    n_numpy_checked = 0
    for w, g in zip(ws, got):
        r = ob.lba_solve(w)
        assert g.iterations == r.iterations
        np.testing.assert_array_equal(g.trials_trace, r.trials_trace)
        # the estimates are only comparable where the gauge is fixed: a stereo window needs one fixed keyframe, a window
        # with monocular edges only needs two (scale); otherwise the cost agrees but the state may drift along the gauge.
        # In a gauge-free window S is singular up to lambda (cond ~1e8 with a user lambda of 1e-3), so the cost trace carries
        # the rounding of the implementation: on window 901 (9 free, 0 fixed keyframes) oracle/lba_oracle.c and the
        # independent oracle/lm_numpy.py disagree by 1.35e-6 on one entry, and the device agrees with the numpy model to 4e-8
        all_mono = bool((w.edge_kind == 0).all())
        gauge_fixed = w.n_fixed >= (2 if all_mono else 1)
        np.testing.assert_allclose(g.chi2_trace, r.chi2_trace, rtol=1e-6 if gauge_fixed else 2e-5)
        if not gauge_fixed and w.n_points <= 250:
            # the wider bound above is the C oracle's own rounding (see the comment): against the independent numpy model
            # (oracle/lm_numpy.py: dense full system, its own LM loop) the device holds the 1e-6 of every other window, so a
            # regression of the fast-math edge cores cannot hide under the 2e-5
            from oracle import lm_numpy
            _, tr, _ = lm_numpy.lm_optimize(w)
            assert tr["iterations"] == g.iterations and tr["trials"] == list(g.trials_trace[:g.iterations])
            np.testing.assert_allclose(g.chi2_trace[:g.iterations], tr["chi2"], rtol=1e-6)
            n_numpy_checked += 1
        if gauge_fixed:
            worst = max(worst, rel_translation_error(g.pose_qt, r.pose_qt))
            np.testing.assert_allclose(g.points, r.points, rtol=1e-5, atol=1e-5, err_msg=f"window {w.n_free}+{w.n_fixed} KF, {w.n_points} points")
    assert worst < 1e-6
    assert n_numpy_checked >= 2


def test_fisheye_monocular_window_kannala_brandt8(solver, ob):
    """Monocular KannalaBrandt8 windows (osh_lba_problem.kb8; EdgeSE3ProjectXYZ through KannalaBrandt8::project / projectJac,
    src/CameraModels/KannalaBrandt8.cpp:45-63,147-175): assembled blocks, the full LM run, and a batch that mixes a fisheye
    window with pinhole ones (the KB8 kernel instantiations take the pinhole path per window)."""
    w = synth.make_window(31, n_free=12, n_fixed=4, n_points=1500, stereo=False, track_len=(3, 10), fisheye=True)
    solver.upload([w])
    _assert_blocks(solver.linearize(0), ob.lba_linearize(w), tol=1e-10)
    got = solver.solve([w])[0]
    ref = ob.lba_solve(w)
    assert ref.iterations >= 5 and ref.chi2_trace[ref.iterations - 1] < 0.3 * ref.chi2_initial
    _check_result(got, ref, w)
    ws = [w, synth.make_config1(3), synth.make_window(32, n_free=5, n_fixed=2, n_points=300, stereo=True, track_len=(2, 6)),
          synth.make_window(33, n_free=7, n_fixed=3, n_points=600, stereo=False, track_len=(3, 8), fisheye=True, obs_dropout=0.2)]
    for g, wi in zip(solver.solve(ws), ws):
        _check_result(g, ob.lba_solve(wi), wi)
    # stereo edges cannot live in a fisheye window
    bad = synth.make_window(34, n_free=4, n_fixed=2, n_points=100, stereo=False, track_len=(2, 5), fisheye=True)
    bad.edge_kind[3] = 1
    with pytest.raises(RuntimeError, match="KannalaBrandt8"):
        solver.upload([bad])


def test_fisheye_stereo_rig_body_edges(solver, ob):
    """Fisheye STEREO rig: left KannalaBrandt8 edges and right-camera EdgeSE3ProjectXYZToBody edges
    (include/OptimizableTypes.h:117-144, src/OptimizableTypes.cpp:192-213, created at src/Optimizer.cc:1365-1399), the two edges
    of a (keyframe, landmark) pair sharing one Hessian block: assembled blocks, one trial, the full LM run, per-edge chi2 /
    isDepthPositive of BOTH edges of a pair, and a batch that mixes a rig window with pinhole and monocular-fisheye ones."""
    from orb_slam3_study_kr_amd import capi
    w = synth.make_rig_window(81, n_free=10, n_fixed=3, n_points=1200, track_len=(3, 9))
    n_body = int((w.edge_kind == capi.OSH_EDGE_BODY).sum())
    assert n_body > 2000 and int((w.edge_kind == capi.OSH_EDGE_MONO).sum()) > 2000
    solver.upload([w])
    _assert_blocks(solver.linearize(0), ob.lba_linearize(w), tol=1e-10)
    S, bs, x = solver.debug_trial(0, 1e-3)
    So, bso, xo = ob.lba_schur_step(w, 1e-3)
    iu = np.triu_indices(S.shape[0])
    np.testing.assert_allclose(S[iu], So[iu], rtol=1e-9, atol=1e-10 * np.abs(So).max())
    np.testing.assert_allclose(bs, bso, rtol=1e-9, atol=1e-10 * np.abs(bso).max())
    got = solver.solve([w])[0]
    ref = ob.lba_solve(w)
    assert ref.iterations >= 5 and ref.chi2_trace[ref.iterations - 1] < 0.5 * ref.chi2_initial
    # float32 theta / psi of both cameras: the staircase tolerances of the fisheye path, translations at 1e-6
    # (lambda: its update is 1 - (2 rho - 1)^3 with rho a ratio of two cost differences, which near convergence are of the
    #  size of the staircase itself)
    _check_result(got, ref, w, chi_tol=2e-6, pts_tol=1e-4, lam_tol=5e-3)
    ws = [w, synth.make_config1(3), synth.make_window(33, n_free=7, n_fixed=3, n_points=600, stereo=False, track_len=(3, 8), fisheye=True),
          synth.make_rig_window(82, n_free=4, n_fixed=2, n_points=200, track_len=(2, 6), right_frac=0.3, right_only_frac=0.4)]
    for g, wi in zip(solver.solve(ws), ws):
        _check_result(g, ob.lba_solve(wi), wi, chi_tol=2e-6, pts_tol=1e-4, lam_tol=5e-3)
    # a body edge without the rig description is an argument error
    bad = synth.make_window(34, n_free=4, n_fixed=2, n_points=100, stereo=False, track_len=(2, 5), fisheye=True)
    bad.edge_kind[3] = capi.OSH_EDGE_BODY
    with pytest.raises(RuntimeError, match="body edge"):
        solver.upload([bad])


def test_observations_that_are_not_float32_values_take_the_double_records(solver, ob):
    """osh_lba_upload ships the observation records as float32 when every value is one (what the reference stores) and as
    doubles otherwise: both paths against the oracle."""
    w = synth.make_window(91, n_free=6, n_fixed=2, n_points=300, stereo=True)
    _check_result(solver.solve([w])[0], ob.lba_solve(w), w)
    w.edge_obs = w.edge_obs + 1e-7 * np.sin(np.arange(w.edge_obs.size)).reshape(w.edge_obs.shape)
    w.edge_info = w.edge_info * (1 + 1e-9)
    assert np.any(w.edge_obs.astype(np.float32).astype(np.float64) != w.edge_obs)
    _check_result(solver.solve([w])[0], ob.lba_solve(w), w)


def test_envelope_factorisation_equals_the_dense_one_bit_for_bit(hip_lib, monkeypatch):
    """k_solve skips the tile columns outside the column envelope of S (blocks no landmark connects: exact zeros that stay zeros under
    LDL^T without pivoting).  OSH_LBA_DENSE_SOLVE=1 factors every tile column as before: same bits out, for a banded window (contiguous
    tracks), a window with missed detections (ragged envelope) and a batch."""
    ws = [synth.make_window(61, n_free=30, n_fixed=5, n_points=3000, stereo=True, track_len=(3, 8)),
          synth.make_window(62, n_free=24, n_fixed=4, n_points=2000, stereo=True, track_len=(4, 20), obs_dropout=0.3),
          synth.make_config1(5)]
    out = []
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("OSH_LBA_DENSE_SOLVE", "1")
        with lba.LbaSolver(0) as s:
            out.append(s.solve(ws))
    for a, b in zip(*out):
        assert a.iterations == b.iterations and a.trials == b.trials
        np.testing.assert_array_equal(a.chi2_trace, b.chi2_trace)
        np.testing.assert_array_equal(a.pose_qt, b.pose_qt)
        np.testing.assert_array_equal(a.points, b.points)
