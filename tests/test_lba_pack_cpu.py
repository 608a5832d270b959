"""Host-side batch packer of osh_lba_upload (csrc/lba_pack.h) on CPU: ``osh_lba_pack_check`` packs the windows exactly as the
upload does (into malloc'ed instead of pinned staging) and verifies the layout the kernels rely on -- landmark renumbering
along the Schur plan, landmark-major edge order, sign-coded observation records, merged fisheye-rig edge pairs
(src/Optimizer.cc:1305-1399), chunks, rebased contribution slots."""
import ctypes as C

import numpy as np
import pytest

from orb_slam3_study_kr_amd import capi, synth


def pack_check(windows, threads=0):
    lib = capi.load_library()
    arr = (capi.LbaProblem * len(windows))()
    for i, w in enumerate(windows):
        arr[i] = w.as_struct()
    st = np.zeros(8, dtype=np.int64)
    ms = C.c_double(0.0)
    rc = lib.osh_lba_pack_check(len(windows), arr, threads, capi.ptr(st, capi.c_int64_p), C.byref(ms))
    return rc, lib.osh_last_error().decode(), dict(items=int(st[0]), sym=int(st[1]), recs=int(st[2]), contrib=int(st[3]), chunks=int(st[4]),
                                                   staging_bytes=int(st[5]), merged=int(st[6]), reduce_entries=int(st[7])), ms.value


def test_single_windows_pack():
    for w in (synth.make_config1(1), synth.make_window(3, n_free=6, n_fixed=2, n_points=400, stereo=True),
              synth.make_window(9, n_free=12, n_fixed=3, n_points=300, stereo=False, mixed_mono_frac=0.4)):
        rc, msg, st, _ = pack_check([w])
        assert rc == 0, msg
        assert st["sym"] >= 1 and st["recs"] >= w.n_points


def test_config2_window_packs_and_long_tracks_split():
    w = synth.make_config2(100)
    rc, msg, st, ms = pack_check([w])
    assert rc == 0, msg
    assert st["items"] > st["sym"] > 0            # landmarks with more than 8 optimisable observers -> cross items
    assert st["chunks"] >= w.n_edges // 1024


def test_heterogeneous_batch_is_thread_count_independent():
    ws = [synth.make_window(200 + i, n_free=4 + 3 * i, n_fixed=1 + i, n_points=150 + 90 * i, stereo=bool(i % 2)) for i in range(5)]
    a = pack_check(ws, threads=1)
    b = pack_check(ws, threads=4)
    assert a[0] == 0 and b[0] == 0, (a[1], b[1])
    assert a[2] == b[2]


def test_empty_and_degenerate_windows():
    w = synth.make_window(5, n_free=3, n_fixed=1, n_points=40, stereo=True)
    # a landmark without any edge, and a window whose edges all sit on fixed keyframes
    keep = w.edge_point != 7
    w2 = synth.LbaWindow(n_free=w.n_free, n_fixed=w.n_fixed, pose_qt=w.pose_qt, pose_cam=w.pose_cam, points=w.points,
                         edge_pose=w.edge_pose[keep], edge_point=w.edge_point[keep], edge_kind=w.edge_kind[keep],
                         edge_obs=w.edge_obs[keep], edge_info=w.edge_info[keep]).normalise()
    fixed_only = w.edge_pose >= w.n_free
    w3 = synth.LbaWindow(n_free=w.n_free, n_fixed=w.n_fixed, pose_qt=w.pose_qt, pose_cam=w.pose_cam, points=w.points,
                         edge_pose=w.edge_pose[fixed_only], edge_point=w.edge_point[fixed_only], edge_kind=w.edge_kind[fixed_only],
                         edge_obs=w.edge_obs[fixed_only], edge_info=w.edge_info[fixed_only]).normalise()
    rc, msg, st, _ = pack_check([w2, w3])
    assert rc == 0, msg


def _rig_window(seed=11):
    """A fisheye window whose free observations also carry a right-camera (body) edge on the same (keyframe, landmark)."""
    w = synth.make_window(seed, n_free=5, n_fixed=2, n_points=120, stereo=False, fisheye=True)
    E = w.n_edges
    rng = np.random.default_rng(seed)
    pick = np.nonzero(rng.uniform(size=E) < 0.6)[0]
    body_obs = w.edge_obs[pick].copy()
    body_obs[:, :2] += rng.normal(0, 0.5, (len(pick), 2))
    w2 = synth.LbaWindow(
        n_free=w.n_free, n_fixed=w.n_fixed, pose_qt=w.pose_qt, pose_cam=w.pose_cam, points=w.points,
        edge_pose=np.concatenate([w.edge_pose, w.edge_pose[pick]]), edge_point=np.concatenate([w.edge_point, w.edge_point[pick]]),
        edge_kind=np.concatenate([w.edge_kind, np.full(len(pick), capi.OSH_EDGE_BODY, dtype=np.uint8)]),
        edge_obs=np.concatenate([w.edge_obs, body_obs]), edge_info=np.concatenate([w.edge_info, w.edge_info[pick]]),
        kb8=w.kb8, cam2=np.concatenate([w.pose_cam[0, :4], w.kb8]), trl=np.array([0, 0, 0, 1.0, -0.1, 0, 0])).normalise()
    return w2, len(pick)


def test_fisheye_rig_pairs_merge_into_one_sorted_edge():
    w, n_pairs = _rig_window()
    rc, msg, st, _ = pack_check([w])
    assert rc == 0, msg
    assert st["merged"] == n_pairs


def test_duplicate_pair_without_a_rig_is_refused():
    w = synth.make_window(3, n_free=4, n_fixed=1, n_points=50, stereo=True)
    w2 = synth.LbaWindow(n_free=w.n_free, n_fixed=w.n_fixed, pose_qt=w.pose_qt, pose_cam=w.pose_cam, points=w.points,
                         edge_pose=np.concatenate([w.edge_pose, w.edge_pose[:1]]), edge_point=np.concatenate([w.edge_point, w.edge_point[:1]]),
                         edge_kind=np.concatenate([w.edge_kind, w.edge_kind[:1]]), edge_obs=np.concatenate([w.edge_obs, w.edge_obs[:1]]),
                         edge_info=np.concatenate([w.edge_info, w.edge_info[:1]])).normalise()
    rc, msg, _, _ = pack_check([w2])
    assert rc == capi.OSH_ERR_UNSUPPORTED and "observed twice" in msg
    # a body edge needs the rig description
    w3 = synth.make_window(3, n_free=4, n_fixed=1, n_points=50, stereo=False)
    w3.edge_kind = w3.edge_kind.copy()
    w3.edge_kind[0] = capi.OSH_EDGE_BODY
    rc, msg, _, _ = pack_check([w3])
    assert rc == capi.OSH_ERR_INVALID and "body edge" in msg


def test_shuffled_edges_and_double_records_take_the_general_paths():
    """Edges in any order pack to the same structure as the reference's landmark-major order (counting sort + insertion sort
    instead of the fast path); observations that are not float32 values repack the batch with double records."""
    w = synth.make_window(12, n_free=7, n_fixed=3, n_points=500, stereo=True)
    rc, msg, st, _ = pack_check([w])
    assert rc == 0, msg
    rng = np.random.default_rng(1)
    perm = rng.permutation(w.n_edges)
    w2 = synth.make_window(12, n_free=7, n_fixed=3, n_points=500, stereo=True)
    for k in ("edge_pose", "edge_point", "edge_kind", "edge_obs", "edge_info"):
        setattr(w2, k, np.ascontiguousarray(getattr(w2, k)[perm]))
    rc2, msg2, st2, _ = pack_check([w2])
    assert rc2 == 0, msg2
    assert {k: st2[k] for k in ("items", "sym", "recs", "contrib", "chunks", "staging_bytes")} == {k: st[k] for k in ("items", "sym", "recs", "contrib", "chunks", "staging_bytes")}
    w3 = synth.make_window(12, n_free=7, n_fixed=3, n_points=500, stereo=True)
    w3.edge_obs = w3.edge_obs + 1e-9                     # no longer float32 values
    rc3, msg3, st3, _ = pack_check([w, w3])
    assert rc3 == 0, msg3
    rc4, _, st4, _ = pack_check([w, w])
    assert st3["staging_bytes"] > st4["staging_bytes"]   # 32-byte records instead of 16-byte ones
