"""Host side of Optimizer::FullInertialBA / MergeInertialBA (csrc/host/OptimizerInertialMap.cc) without a GPU: the keyframe sets,
inertial links and visual edges the graph walk selects, checked against what the reference's walk gives by hand for the synthetic map
(src/Optimizer.cc:417-470, 480-579, 604-727; 3958-4114, 4203-4262, 4290-4382), and that the CPU restatement solves the packed problem."""
import dataclasses

import numpy as np
import pytest

from orb_slam3_study_kr_amd import host, synth, synth_inertial as si


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    return binding


def test_full_inertial_ba_packs_every_keyframe_of_the_map(ob):
    w = si.make_inertial_window(81, n_opt=14, n_fixed=4, n_points=900)
    with host.HostInertialGraph(w) as g:
        pw, kid, mid, idle = g.packed_full(7)
        # every keyframe is an optimisable vertex, in id order; the four keyframes without IMU are pose-only (no link touches them)
        assert (pw.n_opt, pw.n_fixed_imu, pw.n_fixed, idle) == (19, 0, 0, 0)
        np.testing.assert_array_equal(kid, np.sort(g.kf_id))
        np.testing.assert_array_equal(np.sort(kid[pw.link_cur]), 100 + np.arange(14))
        np.testing.assert_array_equal(kid[pw.link_prev], kid[pw.link_cur] - 1)
        assert pw.link_robust.all() and pw.lambda_init == 1e-5 and pw.max_iterations == 7
        # all observations become edges; information = invSigma2 of the octave alone (:646,672), Huber deltas as floats
        assert pw.edge_pose.size == w.edge_pose.size
        key = lambda ww, ids_k, ids_p: sorted(zip(ids_k[ww.edge_pose].tolist(), ids_p[ww.edge_point].tolist()))
        assert key(pw, kid, mid) == key(w, g.kf_id, g.mp_id)
        lv = np.float32(synth.INV_LEVEL_SIGMA2).astype(np.float64)
        assert np.isin(pw.edge_info, lv).all()
        # the inertial information of a link does not depend on where the link sits (no 1e-2 factor on the oldest one, :523-549)
        l_old = int(np.argmin(kid[pw.link_cur]))
        np.testing.assert_allclose(pw.link_info[l_old].ravel(), w.link_info[int(np.nonzero(w.link_cur == 0)[0][0])].ravel() * 100.0, rtol=1e-6, atol=1e-3)
        ref = ob.liba_solve(pw)
        assert ref.iterations == 7 and ref.chi2_final < 0.2 * ref.chi2_initial
        assert g.packed_full(7, fix_local=True) == -3
        # bInit (:452-462,514-518,551,581-601): one bias pair in the slot of a keyframe no link ends at, started from the biases of the
        # last keyframe of the map's list; no random walks; the priors as the random-walk pair of a link from a virtual fixed keyframe
        pi, kidi, _, _ = g.packed_full(7, init=True, prior_g=3.0, prior_a=5.0)
        slot = int(pi.link_bias[0])
        assert (pi.n_opt, pi.n_fixed_imu, pi.n_links) == (19, 1, 15) and kidi[19] == -1
        assert (pi.link_bias[:-1] == slot).all() and slot not in pi.link_cur[:-1].tolist() and pi.link_bias[-1] == 19
        assert not pi.link_info_g[:-1].any() and not pi.link_info_a[:-1].any() and not pi.link_info[-1].any() and pi.link_robust[:-1].all()
        np.testing.assert_array_equal(pi.link_info_g[-1].reshape(3, 3), 3.0 * np.eye(3))
        np.testing.assert_array_equal(pi.link_info_a[-1].reshape(3, 3), 5.0 * np.eye(3))
        np.testing.assert_array_equal(pi.link_info[:-1], pw.link_info)
        last = len(g.kf_id) - 1                                        # Map::GetAllKeyFrames order of the test double = creation order
        np.testing.assert_array_equal(pi.bias_a.reshape(-1, 3)[slot], g.kf_bias(last)[:3].astype(np.float64))
        assert not pi.bias_a.reshape(-1, 3)[19].any() and not pi.vel.reshape(-1, 3)[19].any()
        ri = ob.liba_solve(pi)
        assert ri.chi2_final < 0.2 * ri.chi2_initial


def test_merge_inertial_ba_selects_both_chains_and_the_covisible_keyframes(ob):
    w = si.make_inertial_window(83, n_opt=20, n_fixed=3, n_points=1500)
    keep = w.link_cur != 9
    w = dataclasses.replace(w, **{f: getattr(w, f)[keep] for f in ("link_prev", "link_cur", "link_preint", "link_info", "link_info_g", "link_info_a", "link_robust")})
    w.gt["link_cov"] = w.gt["link_cov"][keep]
    with host.HostInertialGraph(w, no_prev=(9,)) as g:
        pw, kid, mid, tid, cid = g.packed_merge(19, 4)
        # current keyframe + 5 predecessors; merge keyframe + 2 predecessors; then its successors up to 12 (:3972-4049)
        assert tid.tolist() == [119, 118, 117, 116, 115, 114, 104, 103, 102, 105, 106, 107]
        assert cid[0] == 113 and len(set(cid.tolist())) == len(cid) and not set(cid.tolist()) & set(tid.tolist())   # :3988-3992
        assert kid[-1] == 101 and (pw.n_fixed_imu, pw.n_fixed) == (1, 0) and pw.n_opt == len(tid) + len(cid)          # :4018-4022
        np.testing.assert_array_equal(kid[:pw.n_opt], np.sort(np.concatenate([tid, cid])))
        # one inertial link per temporal keyframe, to its predecessor wherever that one sits (temporal, covisible or fixed)
        np.testing.assert_array_equal(kid[pw.link_cur], tid)
        np.testing.assert_array_equal(kid[pw.link_prev], tid - 1)
        assert pw.link_robust.all() and pw.lambda_init == 1e3 and pw.max_iterations == 8
        # points: the matches of the temporal keyframes; edges: their observations from keyframes that carry a vertex
        kf_of = {int(i): k for k, i in enumerate(g.kf_id)}
        temporal = np.array([kf_of[i] for i in tid.tolist()])
        in_problem = np.array([kf_of[i] for i in kid.tolist()])
        pts = np.unique(w.edge_point[np.isin(w.edge_pose, temporal)])
        np.testing.assert_array_equal(mid, np.sort(g.mp_id[pts]))
        assert pw.edge_pose.size == int((np.isin(w.edge_point, pts) & np.isin(w.edge_pose, in_problem)).sum())
        ref = ob.liba_solve(pw)
        assert ref.iterations == 8 and np.isfinite(ref.chi2_final) and ref.chi2_final < ref.chi2_initial
