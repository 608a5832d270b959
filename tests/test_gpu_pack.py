"""The device packer of osh_lba_upload (csrc/lba_pack_device.hip) against the host packer (csrc/lba_pack.h, the CPU checker of
tests/test_lba_pack_cpu.py): ``osh_lba_pack_compare`` packs the same problems with both and compares every section of the two
layouts byte for byte -- landmark renumbering, landmark-major edge order, observation records, chunks, Schur items, records, slot
bytes, contribution slots, block ranges -- i.e. what SparseOptimizer::initializeOptimization + BlockSolver::buildStructure
(Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:199-267, block_solver.hpp:143-295) build inside the call this library replaces."""
import numpy as np
import pytest

from orb_slam3_study_kr_amd import capi, lba, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver(hip_lib):
    with lba.LbaSolver(0) as s:
        yield s


def _shuffled(w, seed=1):
    perm = np.random.default_rng(seed).permutation(w.n_edges)
    return synth.LbaWindow(n_free=w.n_free, n_fixed=w.n_fixed, pose_qt=w.pose_qt, pose_cam=w.pose_cam, points=w.points,
                           edge_pose=np.ascontiguousarray(w.edge_pose[perm]), edge_point=np.ascontiguousarray(w.edge_point[perm]),
                           edge_kind=np.ascontiguousarray(w.edge_kind[perm]), edge_obs=np.ascontiguousarray(w.edge_obs[perm]),
                           edge_info=np.ascontiguousarray(w.edge_info[perm]), kb8=w.kb8).normalise()


def test_small_windows_pack_identically(solver):
    for w in (synth.make_config1(1), synth.make_window(3, n_free=6, n_fixed=2, n_points=400, stereo=True),
              synth.make_window(9, n_free=12, n_fixed=3, n_points=300, stereo=False, mixed_mono_frac=0.0),
              synth.make_window(10, n_free=12, n_fixed=3, n_points=300, stereo=True, mixed_mono_frac=0.4)):
        st = solver.pack_compare([w])
        assert st["records"] >= w.n_points and st["sections"] >= 20


def test_config2_window_with_cross_items_packs_identically(solver):
    w = synth.make_config2(100)
    st = solver.pack_compare([w])
    assert st["items"] > 300 and st["records"] > w.n_points     # long tracks are cut into parts: cross items exist


def test_heterogeneous_batch_and_shuffled_edges(solver):
    ws = [synth.make_window(200 + i, n_free=4 + 3 * i, n_fixed=1 + i, n_points=150 + 90 * i, stereo=bool(i % 2), obs_dropout=0.1 * (i % 3)) for i in range(7)]
    ws.append(_shuffled(ws[3]))
    ws.append(_shuffled(synth.make_config2(101), 2))
    solver.pack_compare(ws)


def test_double_records_and_degenerate_windows(solver):
    w = synth.make_window(12, n_free=7, n_fixed=3, n_points=500, stereo=True)
    w3 = synth.make_window(12, n_free=7, n_fixed=3, n_points=500, stereo=True)
    w3.edge_obs = w3.edge_obs + 1e-9                     # not float32 values: the batch keeps 32-byte records
    solver.pack_compare([w, w3])
    base = synth.make_window(5, n_free=3, n_fixed=1, n_points=40, stereo=True)
    keep = base.edge_point != 7                          # a landmark without any edge
    w4 = synth.LbaWindow(n_free=base.n_free, n_fixed=base.n_fixed, pose_qt=base.pose_qt, pose_cam=base.pose_cam, points=base.points,
                         edge_pose=base.edge_pose[keep], edge_point=base.edge_point[keep], edge_kind=base.edge_kind[keep],
                         edge_obs=base.edge_obs[keep], edge_info=base.edge_info[keep]).normalise()
    fixed_only = base.edge_pose >= base.n_free           # every edge on a fixed keyframe: no optimisable observer anywhere
    w5 = synth.LbaWindow(n_free=base.n_free, n_fixed=base.n_fixed, pose_qt=base.pose_qt, pose_cam=base.pose_cam, points=base.points,
                         edge_pose=base.edge_pose[fixed_only], edge_point=base.edge_point[fixed_only], edge_kind=base.edge_kind[fixed_only],
                         edge_obs=base.edge_obs[fixed_only], edge_info=base.edge_info[fixed_only]).normalise()
    solver.pack_compare([w4, w5, base])


def test_sliced_upload_of_a_large_batch_and_its_restaging_with_double_records(solver):
    """From 128 windows on the staging pass and the host-to-device copy overlap slice by slice (lba_pack_device.hip:device_pack_batch).  A batch of
    140 windows packs to the host packer's bytes; so does the same batch with ONE observation that is not a float32 value in its last window
    -- found while the earlier slices' float32 records are already on their way, the whole batch is staged and sent again as doubles."""
    ws = [synth.make_window(700 + i, n_free=3 + i % 4, n_fixed=1 + i % 2, n_points=60 + 7 * (i % 9), stereo=bool(i % 2)) for i in range(140)]
    st = solver.pack_compare(ws)
    assert st["records"] > 140 * 60
    ws[-1].edge_obs = ws[-1].edge_obs + 1e-9
    solver.pack_compare(ws)
    a = solver.solve(ws)
    solver.set_pack_mode(0)
    b = solver.solve(ws)
    solver.set_pack_mode(-1)
    for x, y in zip(a, b):
        assert x.iterations == y.iterations and np.array_equal(x.pose_qt, y.pose_qt) and np.array_equal(x.points, y.points)


def test_fisheye_monocular_window(solver):
    solver.pack_compare([synth.make_window(21, n_free=8, n_fixed=2, n_points=500, stereo=False, fisheye=True)])


def test_fisheye_stereo_rig_pairs_merge_on_the_device(solver):
    """A fisheye stereo rig puts two edges on one (keyframe, landmark) Hessian block (left EdgeSE3ProjectXYZ + right EdgeSE3ProjectXYZToBody,
    src/Optimizer.cc:1305-1399): the packers merge them into ONE sorted edge with two observation records.  On the device: the pair is
    found by the rank counting (a pose twice on a landmark is only allowed as mono + body), the merged edges leave the sorted lists
    through a prefix count, lone right-camera edges stay.  Same bytes as the host packer, alone and in a batch with other camera models;
    same optimisation results."""
    rig = synth.make_rig_window(71, n_free=7, n_fixed=2, n_points=400)
    assert (rig.edge_kind == capi.OSH_EDGE_BODY).sum() > 100
    solver.pack_compare([rig])
    solver.pack_compare([rig, synth.make_window(72, n_free=5, n_fixed=2, n_points=300, stereo=False, fisheye=True),
                         synth.make_window(73, n_free=6, n_fixed=2, n_points=300, stereo=True), _shuffled_rig(rig)])
    out = []
    for mode in (0, 1):
        solver.set_pack_mode(mode)
        out.append(solver.solve([rig, synth.make_rig_window(74, n_free=4, n_fixed=2, n_points=200)]))
    solver.set_pack_mode(-1)
    for a, b in zip(*out):
        assert a.iterations == b.iterations
        assert np.array_equal(a.pose_qt, b.pose_qt) and np.array_equal(a.points, b.points) and np.array_equal(a.edge_chi2, b.edge_chi2)


def _shuffled_rig(w, seed=3):
    perm = np.random.default_rng(seed).permutation(w.n_edges)
    return synth.LbaWindow(n_free=w.n_free, n_fixed=w.n_fixed, pose_qt=w.pose_qt, pose_cam=w.pose_cam, points=w.points,
                           edge_pose=np.ascontiguousarray(w.edge_pose[perm]), edge_point=np.ascontiguousarray(w.edge_point[perm]),
                           edge_kind=np.ascontiguousarray(w.edge_kind[perm]), edge_obs=np.ascontiguousarray(w.edge_obs[perm]),
                           edge_info=np.ascontiguousarray(w.edge_info[perm]), kb8=w.kb8, cam2=w.cam2, trl=w.trl).normalise()


def test_map_sized_window_takes_the_global_memory_paths(solver):
    """300 optimisable keyframes, 30 k landmarks with missed detections: thousands of distinct observer sets (sorted in global
    memory instead of LDS) and more landmarks than the LDS chunk walk holds."""
    w = synth.make_window(900, n_free=300, n_fixed=3, n_points=30000, stereo=True, track_len=(3, 30), obs_dropout=0.25, max_iterations=5)
    st = solver.pack_compare([w])
    assert st["items"] > 1024


def test_refusals_match_the_host_packer(solver):
    w = synth.make_window(3, n_free=4, n_fixed=1, n_points=50, stereo=True)
    dup = synth.LbaWindow(n_free=w.n_free, n_fixed=w.n_fixed, pose_qt=w.pose_qt, pose_cam=w.pose_cam, points=w.points,
                          edge_pose=np.concatenate([w.edge_pose, w.edge_pose[:1]]), edge_point=np.concatenate([w.edge_point, w.edge_point[:1]]),
                          edge_kind=np.concatenate([w.edge_kind, w.edge_kind[:1]]), edge_obs=np.concatenate([w.edge_obs, w.edge_obs[:1]]),
                          edge_info=np.concatenate([w.edge_info, w.edge_info[:1]])).normalise()
    for mode in (0, 1):
        solver.set_pack_mode(mode)
        with pytest.raises(RuntimeError, match="observed twice"):
            solver.upload([w, dup])
    bad = synth.make_window(3, n_free=4, n_fixed=1, n_points=50, stereo=True)
    bad.edge_point = bad.edge_point.copy()
    bad.edge_point[5] = bad.n_points                     # landmark index out of range
    msgs = []
    for mode in (0, 1):
        solver.set_pack_mode(mode)
        with pytest.raises(RuntimeError, match="index or kind out of range") as ei:
            solver.upload([bad])
        msgs.append(str(ei.value))
    assert msgs[0] == msgs[1]
    solver.set_pack_mode(-1)
    solver.upload([w])                                   # the context is usable after a refusal


def test_both_packers_give_bitwise_equal_optimisations(solver):
    ws = [synth.make_window(40 + i, n_free=5 + 2 * i, n_fixed=2, n_points=300 + 100 * i, stereo=True) for i in range(4)]
    out = []
    for mode in (0, 1):
        solver.set_pack_mode(mode)
        out.append(solver.solve(ws))
    solver.set_pack_mode(-1)
    for a, b in zip(*out):
        assert a.iterations == b.iterations and a.trials == b.trials
        assert np.array_equal(a.pose_qt, b.pose_qt) and np.array_equal(a.points, b.points) and np.array_equal(a.edge_chi2, b.edge_chi2)


def test_default_rule_batches_on_the_device_single_windows_on_the_host(solver):
    """The default rule sends batches to the device packer and a handful of windows to host threads.  A window solved alone is cut into
    smaller Schur items than in a batch (schur_plan.h:item_max_lm: the latency of a single call), so its sums run in another order: the
    same iterations and trials, the same results to rounding (the two packers give the same BITS for the same batch: the tests above)."""
    ws = [synth.make_window(300 + i, n_free=4 + i % 5, n_fixed=1 + i % 3, n_points=120 + 15 * i, stereo=bool(i % 2)) for i in range(32)]
    solver.set_pack_mode(-1)
    batch = solver.solve(ws)
    assert solver.pack_profile()["on_device"]
    for i in (0, 7, 19, 31):
        one = solver.solve([ws[i]])[0]
        assert not solver.pack_profile()["on_device"]
        assert one.iterations == batch[i].iterations and one.trials == batch[i].trials
        np.testing.assert_allclose(one.pose_qt, batch[i].pose_qt, rtol=0, atol=1e-9)
        np.testing.assert_allclose(one.points, batch[i].points, rtol=0, atol=1e-8)
        np.testing.assert_allclose(one.edge_chi2, batch[i].edge_chi2, rtol=1e-6, atol=1e-9)
