"""GPU parity tests of the stereo-inertial local BA (BASELINE.json configs[3]): the persistent-block HIP kernel
(csrc/liba_device.hip) through osh_liba_solve vs the CPU oracle (oracle/liba_oracle.c) on the same windows."""
import numpy as np
import pytest

from orb_slam3_study_kr_amd import lba
from orb_slam3_study_kr_amd import synth_inertial as si

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver(hip_lib):
    with lba.LbaSolver(0) as s:
        yield s


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    return binding


def _check(got, ref, w, tol=1e-6, pts_tol=1e-6, edge_tol=1e-4, lam_tol=1e-5):
    assert got.iterations == ref.iterations
    np.testing.assert_array_equal(got.trials_trace, ref.trials_trace)
    # float32 sinf/cosf of the preintegration getters differ by an ulp between device and host libm; the
    # inertial information is ~1e8, so chi2 agrees to ~1e-9..1e-8, not to 1e-12
    np.testing.assert_allclose(got.chi2_initial, ref.chi2_initial, rtol=1e-7)
    np.testing.assert_allclose(got.chi2_trace, ref.chi2_trace, rtol=1e-6)
    np.testing.assert_allclose(got.chi2_final, ref.chi2_final, rtol=1e-6)
    np.testing.assert_allclose(got.lambda_trace, ref.lambda_trace, rtol=lam_tol)
    t_rel = np.max(np.linalg.norm(got.pose_tcw - ref.pose_tcw, axis=1) / np.linalg.norm(ref.pose_tcw, axis=1))
    assert t_rel < tol, t_rel                                  # north star: SE3 translations <= 1e-6 relative
    assert np.abs(got.pose_Rcw - ref.pose_Rcw).max() < tol
    assert np.abs(got.pose_twb - ref.pose_twb).max() < tol * max(1.0, np.abs(ref.pose_twb).max())
    np.testing.assert_allclose(got.vel, ref.vel, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got.bias_g, ref.bias_g, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(got.bias_a, ref.bias_a, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got.points, ref.points, rtol=pts_tol, atol=pts_tol)
    np.testing.assert_allclose(got.edge_chi2, ref.edge_chi2, rtol=edge_tol, atol=edge_tol / 10)
    np.testing.assert_array_equal(got.edge_depth_pos, ref.edge_depth_pos)


def test_config4_stereo_inertial_window(solver, ob):
    w = si.make_inertial_window(11)            # 10 temporal KFs + predecessor + 20 fixed observers, ~1.1k landmarks
    _check(solver.solve_inertial([w])[0], ob.liba_solve(w), w)


def test_small_and_large_variants_in_one_batch(solver, ob):
    ws = [si.make_inertial_window(21, n_opt=3, n_fixed=2, n_points=80),
          si.make_inertial_window(22, n_opt=6, n_fixed=5, n_points=400, rec_init=True),
          si.make_inertial_window(23, n_opt=10, n_fixed=8, n_points=900, large=True),
          si.make_inertial_window(24, n_opt=5, n_fixed=0, n_points=300)]
    got = solver.solve_inertial(ws)
    for w, g in zip(ws, got):
        _check(g, ob.liba_solve(w), w)
    again = solver.solve_inertial(ws)           # bitwise reproducible: fixed reduction orders, no atomics
    for g, a in zip(got, again):
        np.testing.assert_array_equal(g.pose_tcw, a.pose_tcw)
        np.testing.assert_array_equal(g.points, a.points)


def test_result_is_physically_sensible(solver):
    w = si.make_inertial_window(31)
    g = solver.solve_inertial([w])[0]
    N = w.n_opt
    assert np.abs(g.pose_twb - w.gt["twb"][:N]).max() < 0.01
    assert np.abs(g.vel - w.gt["vel"][:N]).max() < 0.02
    assert g.chi2_final < 0.02 * g.chi2_initial


def test_monocular_fisheye_inertial_window(solver, ob):
    """Mono-inertial window whose camera is a KannalaBrandt8 (TUM-VI style): EdgeMono projects through
    KannalaBrandt8::project / projectJac (osh_liba_problem.kb8); a stereo edge in such a window is refused."""
    w = si.make_inertial_window(13, n_opt=6, n_fixed=6, n_points=700, fisheye=True)
    assert w.kb8 is not None and (w.edge_kind == 0).all()
    ref = ob.liba_solve(w)
    assert ref.iterations >= 3 and ref.chi2_final < 0.1 * ref.chi2_initial
    # monocular: the depth of a low-parallax landmark is weakly constrained, so the 1e-8 noise of the float32 preintegration
    # getters (see _check) shows up at the 1e-5 level in a few landmark coordinates; poses, velocities and biases stay at 1e-6
    # per-edge chi2: the reference rounds theta and psi to float32, so the fisheye residual is a staircase with ~3e-5 px steps;
    # a state difference of 1e-9 can move an edge to the next step (1e-4 .. 1e-3 of a chi2 near 1)
    _check(solver.solve_inertial([w])[0], ref, w, pts_tol=2e-4, edge_tol=2e-3)
    ws = [w, si.make_inertial_window(14, n_opt=4, n_fixed=4, n_points=400)]        # fisheye + pinhole stereo in one batch
    for g, wi in zip(solver.solve_inertial(ws), ws):
        _check(g, ob.liba_solve(wi), wi, pts_tol=2e-4 if wi.kb8 is not None else 1e-6, edge_tol=2e-3 if wi.kb8 is not None else 1e-4)
    w.edge_kind[5] = 1
    with pytest.raises(RuntimeError, match="KannalaBrandt8"):
        solver.solve_inertial([w])


def test_fisheye_stereo_rig_right_camera_edges(solver, ob):
    """LocalInertialBA of a fisheye stereo rig (KeyFrame::mpCamera2, src/Optimizer.cc:2798-2835): EdgeMono(1) edges on camera 1 of
    ImuCamPose, alone or on the same (keyframe, landmark) Hessian block as the left EdgeMono(0); the right edge's information is the
    reference's `mvInvLevelSigma2[kpUn.octave]` of the LEFT keypoint variable."""
    w = si.make_inertial_rig_window(17, n_opt=6, n_fixed=6, n_points=700)
    k = w.edge_kind
    assert (k == 2).sum() > 500 and ((k[1:] == 2) & (k[:-1] == 0) & (w.edge_pose[1:] == w.edge_pose[:-1]) & (w.edge_point[1:] == w.edge_point[:-1])).sum() > 300
    ref = ob.liba_solve(w)
    assert ref.iterations >= 3 and ref.chi2_final < 0.1 * ref.chi2_initial
    _check(solver.solve_inertial([w])[0], ref, w, pts_tol=2e-4, edge_tol=2e-3)      # fisheye tolerances, see the monocular test
    ws = [w, si.make_inertial_window(14, n_opt=4, n_fixed=4, n_points=400), si.make_inertial_rig_window(18, n_opt=3, n_fixed=2, n_points=200)]
    for g, wi in zip(solver.solve_inertial(ws), ws):
        _check(g, ob.liba_solve(wi), wi, pts_tol=2e-4 if wi.kb8 is not None else 1e-6, edge_tol=2e-3 if wi.kb8 is not None else 1e-4)
    w2 = si.make_inertial_window(13, n_opt=3, n_fixed=2, n_points=100, fisheye=True)
    w2.edge_kind[3] = 2                                            # a right-camera edge without cam2 / trl
    with pytest.raises(RuntimeError, match="right-camera"):
        solver.solve_inertial([w2])


@pytest.mark.parametrize("name", ["liba_tiny", "liba_tiny_rig"])
def test_device_matches_the_numpy_lm_golden_outputs(solver, name):
    """The committed fixtures of the independent numpy LM (tests/golden/make_golden.py liba): no oracle in the loop."""
    from helpers import check_against_liba_fixture, load_liba_fixture
    w, z = load_liba_fixture(name)
    check_against_liba_fixture(solver.solve_inertial([w])[0], z, fisheye=w.kb8 is not None)


@pytest.mark.parametrize("group", ["1", "2", "4", "8", "16", "32"])
def test_every_group_size_takes_the_same_path(solver, ob, monkeypatch, group):
    """A window is optimised by a group of 1..32 thread blocks (OSH_LIBA_GROUP; default: 32 for a single window, 1 for a large
    batch).  The group size only changes how the sums are split, so the Levenberg-Marquardt trace is the oracle's for each."""
    monkeypatch.setenv("OSH_LIBA_GROUP", group)
    ws = [si.make_inertial_window(41), si.make_inertial_rig_window(43, n_opt=4, n_fixed=3, n_points=150)]
    got = solver.solve_inertial(ws)
    assert solver.inertial_profile()[0] == int(group)
    for w, g in zip(ws, got):
        rig = w.kb8 is not None
        _check(g, ob.liba_solve(w), w, **(dict(pts_tol=2e-4, edge_tol=2e-3) if rig else {}))   # fisheye tolerances as above


def test_agent_scope_barrier_path_agrees(solver, monkeypatch):
    """OSH_LIBA_HEAVY_BARRIER=1 keeps the agent-scope fences (L2 write-back) a group uses when its blocks are not all on one XCD."""
    w = si.make_inertial_window(45)
    a = solver.solve_inertial([w])[0]
    monkeypatch.setenv("OSH_LIBA_HEAVY_BARRIER", "1")
    b = solver.solve_inertial([w])[0]
    np.testing.assert_array_equal(a.chi2_trace, b.chi2_trace)
    np.testing.assert_array_equal(a.pose_twb, b.pose_twb)
    np.testing.assert_array_equal(a.points, b.points)


def test_a_barrier_that_gives_up_is_retried_with_one_block_per_window(solver, monkeypatch):
    """A group barrier that times out (blocks of other streams kept part of the group off the device) sets the abort word; the call
    then runs the same problem once more with one block per window instead of failing.  OSH_LIBA_TEST_ABORT makes the first barrier
    of the first launch give up at once: the result must be the one-block result, bit for bit."""
    w = si.make_inertial_window(47)
    monkeypatch.setenv("OSH_LIBA_GROUP", "1")
    one = solver.solve_inertial([w])[0]
    monkeypatch.delenv("OSH_LIBA_GROUP")
    monkeypatch.setenv("OSH_LIBA_TEST_ABORT", "1")
    got = solver.solve_inertial([w])[0]
    assert solver.inertial_profile()[0] == 1          # the launch that produced the result ran one block per window
    np.testing.assert_array_equal(got.chi2_trace, one.chi2_trace)
    np.testing.assert_array_equal(got.pose_twb, one.pose_twb)
    np.testing.assert_array_equal(got.points, one.points)
    monkeypatch.delenv("OSH_LIBA_TEST_ABORT")
    assert solver.solve_inertial([w])[0].iterations == one.iterations and solver.inertial_profile()[0] == 32


def test_banded_layout_of_a_map_sized_problem_equals_the_dense_one(solver, monkeypatch):
    """FullInertialBA over a map (src/Optimizer.cc:393-814): with the keyframes in temporal order the reduced system is banded (landmarks and
    IMU links couple nearby keyframes), so the unknowns are interleaved per keyframe and only the band is formed and factored.
    OSH_LIBA_DENSE=1 keeps the dense [poses | velocities, biases] system of round 2: same iterations, the same trace and the same states
    up to the rounding of another elimination order.  One keyframe is fixed (n_fixed_imu) so the minimum is unique."""
    import dataclasses
    w = si.make_inertial_window(905, n_opt=90, n_fixed=0, n_points=3000, large=True)
    w = dataclasses.replace(w, lambda_init=1e-5, max_iterations=5, link_robust=np.ones_like(w.link_robust))
    a = solver.solve_inertial([w])[0]
    monkeypatch.setenv("OSH_LIBA_DENSE", "1")
    b = solver.solve_inertial([w])[0]
    assert a.iterations == b.iterations and list(a.trials_trace[:a.iterations]) == list(b.trials_trace[:b.iterations])
    np.testing.assert_allclose(a.chi2_trace[:a.iterations], b.chi2_trace[:b.iterations], rtol=1e-9)
    np.testing.assert_allclose(a.pose_twb, b.pose_twb, rtol=0, atol=1e-8)
    np.testing.assert_allclose(a.vel, b.vel, rtol=0, atol=1e-7)
    np.testing.assert_allclose(a.points, b.points, rtol=0, atol=1e-7)


def test_banded_problem_does_not_depend_on_what_the_context_ran_before(hip_lib):
    """The banded layout forms only the band of S, but a 16-pivot step of the group factorisation reads its first row 15 columns further:
    structural zeros that have to be written (r03: left unwritten, they were whatever an earlier call had put in the arena -- a map solved
    after a batch of 128 windows in the same context ended on another cost; `bench.py` does exactly that).  Same bits in a fresh context
    and after a batch that has filled the arena."""
    import dataclasses
    w = si.make_inertial_window(1300, n_opt=400, n_fixed=0, n_points=16000, large=True)      # (the map bench.py solves)
    w = dataclasses.replace(w, lambda_init=1e-5, max_iterations=7, link_robust=np.ones_like(w.link_robust))
    with lba.LbaSolver(0) as s:
        a = s.solve_inertial([w])[0]
    with lba.LbaSolver(0) as s:
        base = [si.make_inertial_window(11 + k, n_points=3600) for k in range(8)]
        s.solve_inertial([base[k % 8] for k in range(128)])
        b = s.solve_inertial([w])[0]
        c = s.solve_inertial([w])[0]
    for x in (b, c):
        assert x.iterations == a.iterations and x.trials == a.trials
        for f in ("chi2_trace", "pose_twb", "vel", "bias_g", "bias_a", "points"):
            np.testing.assert_array_equal(getattr(x, f), getattr(a, f), err_msg=f)


def test_large_window_of_25_keyframes(solver, ob):
    """LocalInertialBA's bLarge case: 25 temporal keyframes (src/Optimizer.cc:2394-2400), a 375 x 375 reduced system."""
    w = si.make_inertial_window(61, n_opt=25, n_fixed=10, n_points=1500, large=True)
    _check(solver.solve_inertial([w])[0], ob.liba_solve(w), w)
    with pytest.raises(RuntimeError, match="up to 1200"):
        solver.solve_inertial([si.make_inertial_window(62, n_opt=1201, n_fixed=0, n_points=300, large=True)])


def _without_links_at(w, kfs):
    """The window with every inertial link that touches one of the keyframes `kfs` removed: those keyframes keep their 15-dof
    state, but nothing constrains its velocity / bias part (the pose-only optimisable keyframes of MergeInertialBA and of
    FullInertialBA with !bImu keyframes, whose velocity / bias vertices have no active edge, src/Optimizer.cc:4077-4100)."""
    import dataclasses
    keep = ~(np.isin(w.link_prev, kfs) | np.isin(w.link_cur, kfs))
    return dataclasses.replace(w, link_prev=w.link_prev[keep], link_cur=w.link_cur[keep], link_preint=w.link_preint[keep], link_info=w.link_info[keep],
                               link_info_g=w.link_info_g[keep], link_info_a=w.link_info_a[keep], link_robust=w.link_robust[keep])


@pytest.mark.parametrize("n_opt,lam,its", [(40, 1e-5, 7), (51, 1e3, 8), (100, 1e-5, 3), (150, 1e-5, 3), (200, 1e-5, 2)])
def test_map_sized_windows_and_keyframes_without_links(solver, ob, n_opt, lam, its):
    """Windows beyond LocalInertialBA's 25 keyframes (FullInertialBA over a small map: lambda 1e-5, src/Optimizer.cc:408; MergeInertialBA:
    lambda 1e3, optimize(8), :4121,4388): up to 51 keyframes (a 765 x 765 reduced system) with the LDS-resident LDL^T of the tracker's
    windows, 100 / 150 / 200 keyframes (3000 x 3000) with the factorisation by the whole block group; no fixed observers, some keyframes without inertial links
    (their velocity / bias columns carry lambda only and stay where they are)."""
    import dataclasses
    w = si.make_inertial_window(70 + n_opt, n_opt=n_opt, n_fixed=0, n_points=max(2500, 40 * n_opt), large=True)
    w = _without_links_at(w, [5, 17, 18])
    w = dataclasses.replace(w, lambda_init=lam, max_iterations=its, link_robust=np.ones_like(w.link_robust))
    got, ref = solver.solve_inertial([w])[0], ob.liba_solve(w)
    _check(got, ref, w, lam_tol=1e-3, pts_tol=2e-4, edge_tol=1e-3)
    for k in (5, 17, 18):
        np.testing.assert_array_equal(got.vel[k], w.vel[k])
        np.testing.assert_array_equal(got.bias_g[k], w.bias_g[k])


@pytest.mark.parametrize("n_opt", [4, 12, 30])
def test_one_bias_pair_for_every_inertial_edge_with_priors(solver, ob, n_opt):
    """FullInertialBA with bInit (src/Optimizer.cc:452-462,514-518,581-601) as a flat problem: link_bias names one keyframe for all links
    (every link then shares a keyframe with every other: one colour each), no random walks, the two priors as one link without
    inertial information.  The bias slots of the other keyframes stay where they are."""
    import dataclasses
    w = si.with_shared_bias(si.make_inertial_window(300 + n_opt, n_opt=n_opt, n_fixed=6, n_points=60 * n_opt + 200))
    w = dataclasses.replace(w, lambda_init=1e-5, max_iterations=3)   # at the minimum the gain ratio, and lambda with it, is rounding noise
    got, ref = solver.solve_inertial([w])[0], ob.liba_solve(w)
    _check(got, ref, w, lam_tol=1e-3, pts_tol=2e-4, edge_tol=1e-3)
    np.testing.assert_array_equal(got.bias_g[1:], np.asarray(w.bias_g).reshape(-1, 3)[1:n_opt])
    np.testing.assert_array_equal(got.bias_a[1:], np.asarray(w.bias_a).reshape(-1, 3)[1:n_opt])
    assert np.linalg.norm(got.bias_a[0]) < 0.6 * np.linalg.norm(np.asarray(w.bias_a).reshape(-1, 3)[0])   # information 1e6 pulls it towards the prior
    bad = dataclasses.replace(w, link_info_g=np.tile(np.eye(3).ravel(), (w.n_links, 1)))
    with pytest.raises(RuntimeError, match="no random-walk edges"):
        solver.solve_inertial([bad])
    lb = w.link_bias.copy()
    lb[0] = w.link_cur[0]
    with pytest.raises(RuntimeError, match="later keyframe"):
        solver.solve_inertial([dataclasses.replace(w, link_bias=lb)])


def test_many_window_shapes_through_the_block_groups(solver, ob, monkeypatch):
    """Windows of many shapes (3..12 keyframes, 60..1300 landmarks, with and without fixed observers), one at a time (a group of 32
    blocks each) and as one batch (groups of 16): the barriers, the lane teams and the chunked sums see every remainder case."""
    rng = np.random.Generator(np.random.PCG64(2024))
    ws = []
    for k in range(14):
        ws.append(si.make_inertial_window(200 + k, n_opt=int(rng.integers(3, 13)), n_fixed=int(rng.integers(0, 12)), n_points=int(rng.integers(60, 1300)),
                                          large=bool(k % 3 == 0), rec_init=bool(k % 4 == 1)))
    refs = [ob.liba_solve(w) for w in ws]
    # near convergence the gain ratio is a quotient of two 1e-6-relative cost differences, and lambda follows its cube: 1e-3 there;
    # landmarks seen from two or three nearby keyframes only (the small windows) have a depth the float32 preintegration noise moves
    for w, r in zip(ws, refs):
        _check(solver.solve_inertial([w])[0], r, w, lam_tol=1e-3, pts_tol=2e-4, edge_tol=1e-3)
    assert solver.inertial_profile()[0] == 32
    for g, w, r in zip(solver.solve_inertial(ws), ws, refs):
        _check(g, r, w, lam_tol=1e-3, pts_tol=2e-4, edge_tol=1e-3)
    assert solver.inertial_profile()[0] == 16
