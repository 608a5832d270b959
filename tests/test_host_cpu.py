"""CPU tests of the C++ host layer: the graph walk + packing of Optimizer::LocalBundleAdjustment
(steps 1-6, src/Optimizer.cc:1118-1404) against the synthetic window it was built from; no GPU involved."""
import re
from pathlib import Path

import numpy as np

from orb_slam3_study_kr_amd import capi, host, synth

ROOT = Path(__file__).resolve().parent.parent


def test_host_header_symbols_exported():
    text = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "orbslam3_hip_host.h").read_text(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(osh_host_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(capi.HOST_EXPORTED_SYMBOLS)
    lib = capi.load_library()
    for name in declared:
        assert hasattr(lib, name)


def _check_pack(w, g):
    rc, sizes, o = g.pack()
    assert rc == 0
    P, F, L, E = (int(x) for x in sizes[:4])
    assert (P, F, L, E) == (w.n_free, w.n_fixed, w.n_points, w.n_edges)
    assert sizes[4] == w.n_fixed                       # num_fixedKF = |lFixedCameras| (no init keyframe in the window)
    # optimisable poses come first in ascending keyframe id (Hessian order), fixed ones after
    assert np.all(np.diff(o["pose_kf_id"][:P]) > 0) and np.all(o["pose_kf_id"][:P] >= 100) and np.all(o["pose_kf_id"][P:] < 100)
    assert np.all(np.diff(o["point_mp_id"]) > 0)
    kf_of_pose = {int(i): k for k, i in enumerate(g.kf_id)}
    order = np.array([kf_of_pose[int(i)] for i in o["pose_kf_id"]])
    # float32 storage widened to double (src/Optimizer.cc:1217-1218,1286)
    exp_qt = w.pose_qt[order].astype(np.float32).astype(np.float64)
    np.testing.assert_array_equal(o["pose_qt"][:, 4:], exp_qt[:, 4:])
    # Sophus::SE3f re-normalises the float quaternion it is constructed from (one float ulp)
    np.testing.assert_allclose(o["pose_qt"][:, :4], exp_qt[:, :4], rtol=0, atol=2e-7)
    np.testing.assert_array_equal(o["points"], w.points.astype(np.float32).astype(np.float64))
    np.testing.assert_array_equal(o["pose_cam"], w.pose_cam[order])
    # same edge multiset: (keyframe, point, kind, obs, info); order is landmark-list x std::map<KeyFrame*> order
    def key(pose, point, kind, obs, info):
        return sorted(zip(pose.tolist(), point.tolist(), kind.tolist(), map(tuple, np.round(obs, 4).tolist()), info.tolist()))
    obs_w = w.edge_obs.copy()
    got = key(order[o["edge_pose"]], o["edge_point"], o["edge_kind"], o["edge_obs"], o["edge_info"])
    exp = key(w.edge_pose, w.edge_point, w.edge_kind, obs_w, w.edge_info)
    assert got == exp


def test_pack_matches_window_stereo_and_mono():
    for stereo in (True, False):
        w = synth.make_window(31, n_free=6, n_fixed=3, n_points=200, stereo=stereo)
        with host.HostGraph(w) as g:
            _check_pack(w, g)


def test_pack_mixed_kinds_and_repeatability():
    w = synth.make_window(32, n_free=5, n_fixed=2, n_points=150, stereo=True, mixed_mono_frac=0.4)
    with host.HostGraph(w) as g:
        _check_pack(w, g)
        _check_pack(w, g)   # marks are reset by the harness, a second call sees the same window


def test_init_keyframe_is_a_fixed_vertex():
    w = synth.make_window(33, n_free=5, n_fixed=2, n_points=150)
    with host.HostGraph(w, init_kf_fixed=True) as g:
        rc, sizes, o = g.pack()
        assert rc == 0
        assert (int(sizes[0]), int(sizes[1])) == (w.n_free - 1, w.n_fixed + 1)   # vSE3->setFixed(mnId==InitKFid) :1220
        assert int(sizes[4]) == w.n_fixed + 1                                     # num_fixedKF counts it (:1143-1146,1179)
        assert int(o["pose_kf_id"][int(sizes[0])]) == int(g.kf_id[0])             # first of the fixed block


def test_window_without_fixed_keyframe_aborts():
    w = synth.make_window(34, n_free=5, n_fixed=2, n_points=150)
    keep = w.edge_pose < w.n_free
    for name in ("edge_pose", "edge_point", "edge_kind", "edge_obs", "edge_info"):
        setattr(w, name, np.ascontiguousarray(getattr(w, name)[keep]))
    with host.HostGraph(w) as g:
        rc, sizes, _ = g.pack()
        assert rc == 1 and sizes[4] == 0      # "LM-LBA: There are 0 fixed KF" path (:1182-1186)


def test_global_ba_pack_selects_every_keyframe_and_skips_bad_and_unobserved_points():
    """PackBundleAdjustment (src/Optimizer.cc:112-300): every non-bad keyframe is a vertex, the map's initial keyframe the only
    fixed one; bad points are no vertices; a point whose observers are all bad is removed again (vbNotIncludedMP)."""
    w = synth.make_window(51, n_free=5, n_fixed=2, n_points=120, stereo=True, track_len=(2, 4))
    with host.HostGraph(w, init_kf_id_index=w.n_free) as g:
        pw, o = g.packed_global_window()
        # both fixed poses of the window are keyframes of the map: one is the init keyframe (fixed), the other is optimised
        assert (pw.n_free, pw.n_fixed) == (w.n_free + 1, 1)
        assert list(o["pose_kf_id"][:pw.n_free]) == sorted(o["pose_kf_id"][:pw.n_free])
        assert o["pose_kf_id"][pw.n_free] == g.kf_id[w.n_free]
        assert pw.n_points == w.n_points and pw.n_edges == w.n_edges and o["n_not_included"] == 0
        g.lib.osh_host_set_bad(g.g, -1, 3)
        pw2, o2 = g.packed_global_window()
        assert pw2.n_points == w.n_points - 1 and int(g.mp_id[3]) not in set(o2["point_mp_id"].tolist())
        # make every observer of point 7 bad -> the point has no edge and is dropped from the problem
        obs7 = set(int(p) for p, l in zip(w.edge_pose, w.edge_point) if l == 7)
        for k in obs7:
            g.lib.osh_host_set_bad(g.g, k, -1)
        pw3, o3 = g.packed_global_window()
        assert o3["n_not_included"] >= 1 and int(g.mp_id[7]) not in set(o3["point_mp_id"].tolist())
        assert pw3.n_free + pw3.n_fixed == w.n_free + w.n_fixed - len(obs7)


def test_gba_failure_writes_the_identity_result():
    """When the device solve cannot run (no GPU in this container -> HostSolverContext() fails) BundleAdjustment must leave
    what LoopClosing::RunGlobalBundleAdjustment (src/LoopClosing.cc:2330-2386) reads in a state where its propagation
    `SetPose(mTcwGBA)` / `SetWorldPos(mPosGBA)` is a no-op: mTcwGBA = pose, mPosGBA = position, stamped with nLoopKF."""
    lib = capi.load_library()
    if lib.osh_device_count() > 0:
        import pytest
        pytest.skip("a GPU is visible: the failure path is exercised by tests/test_gpu_host.py instead")
    w = synth.make_window(5, n_free=4, n_fixed=2, n_points=60, stereo=True)
    with host.HostGraph(w, init_kf_fixed=True) as g:
        n_kf = w.n_free + w.n_fixed
        before_kf = [g.kf_pose(i).copy() for i in range(n_kf)]
        before_mp = [g.mp_pos(j).copy() for j in range(w.n_points)]
        g.run_gba(n_iterations=3, n_loop_kf=77, robust=True)
        for i in range(n_kf):
            mark, T = g.kf_pose_gba(i)
            assert mark == 77
            np.testing.assert_array_equal(T, before_kf[i])
            np.testing.assert_array_equal(g.kf_pose(i), before_kf[i])
        for j in range(w.n_points):
            mark, X = g.mp_pos_gba(j)
            assert mark == 77
            np.testing.assert_array_equal(X, before_mp[j])


def test_pack_fisheye_stereo_rig_window_adds_body_edges():
    """a keyframe with mpCamera2 contributes, besides the left edge, an EdgeSE3ProjectXYZToBody for a
    right-camera observation (rightIndex - NLeft into mvKeysRight, src/Optimizer.cc:1365-1399): same multiset of edges as the
    window the graph was built from, body edges carrying Trl and the right camera."""
    w = synth.make_rig_window(73, n_free=5, n_fixed=2, n_points=150, track_len=(2, 6))
    with host.HostGraph(w) as g:
        pw, o = g.packed_window()
        assert pw.n_edges == w.n_edges
        assert np.array_equal(np.bincount(pw.edge_kind, minlength=3), np.bincount(w.edge_kind, minlength=3))
        np.testing.assert_array_equal(pw.cam2, w.cam2.astype(np.float32).astype(np.float64))
        # Sophus::SE3f re-normalises the float quaternion of Trl (one float ulp)
        np.testing.assert_allclose(pw.trl, w.trl, rtol=0, atol=2e-7)
        kf_of_pose = {int(i): k for k, i in enumerate(g.kf_id)}
        order = np.array([kf_of_pose[int(i)] for i in o["pose_kf_id"]])
        mp_of_point = {int(i): k for k, i in enumerate(g.mp_id)}
        pts = np.array([mp_of_point[int(i)] for i in o["point_mp_id"]])

        def key(pose, point, kind, obs, info):
            return sorted(zip(pose.tolist(), point.tolist(), kind.tolist(), map(tuple, np.round(obs[:, :2], 3).tolist()), np.round(info, 6).tolist()))
        assert key(order[pw.edge_pose], pts[pw.edge_point], pw.edge_kind, pw.edge_obs, pw.edge_info) == \
            key(w.edge_pose, w.edge_point, w.edge_kind, w.edge_obs.astype(np.float32).astype(np.float64), w.edge_info)
        # the reference inserts the right edge directly after the pair's left edge
        body = np.nonzero(pw.edge_kind == capi.OSH_EDGE_BODY)[0]
        paired = [e for e in body if e > 0 and pw.edge_kind[e - 1] == 0 and pw.edge_pose[e - 1] == pw.edge_pose[e] and pw.edge_point[e - 1] == pw.edge_point[e]]
        assert len(paired) > 0.5 * len(body)
