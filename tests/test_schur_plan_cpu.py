"""Host-side Schur work plan (csrc/schur_plan.h): coverage self-check on CPU, no GPU needed.

``osh_lba_schur_plan_stats`` rebuilds the plan exactly as ``osh_lba_upload`` does and verifies
that every pair of optimisable observers of every landmark (the products of
Thirdparty/g2o/g2o/core/block_solver.hpp:381-432) is covered by exactly one item record and
that every contribution slot is written once and belongs to the right block of S.
"""
import ctypes as C

import numpy as np
import pytest

from orb_slam3_study_kr_amd import capi, synth


def plan_stats(w):
    lib = capi.load_library()
    pr = w.as_struct()
    st = np.zeros(8, dtype=np.int64)
    rc = lib.osh_lba_schur_plan_stats(C.byref(pr), capi.ptr(st, capi.c_int64_p))
    assert rc == 0, lib.osh_last_error().decode()
    return dict(items=int(st[0]), sym=int(st[1]), recs=int(st[2]), contrib=int(st[3]), ccontrib=int(st[4]),
                mfma=int(st[5]), blocks=int(st[6]), reduce_entries=int(st[7]))


def observer_counts(w):
    free = w.edge_pose < w.n_free
    return np.bincount(w.edge_point[free], minlength=w.n_points)


@pytest.mark.parametrize("maker", [
    lambda: synth.make_config1(1),
    lambda: synth.make_config2(100),
    lambda: synth.make_window(3, n_free=12, n_fixed=3, n_points=800, stereo=True, track_len=(2, 9), obs_dropout=0.2),
    lambda: synth.make_window(5, n_free=40, n_fixed=4, n_points=2000, track_len=(10, 30)),           # > 8 and > 16 observers
    lambda: synth.make_window(6, n_free=40, n_fixed=4, n_points=2000, track_len=(10, 30), obs_dropout=0.3),
    lambda: synth.make_window(7, n_free=1, n_fixed=2, n_points=50, track_len=(2, 3)),
])
@pytest.mark.parametrize("item_max", [8, 24, 64])
def test_plan_covers_every_observer_pair_once(maker, item_max, monkeypatch):
    # (the landmarks per item follow the number of windows of the call -- schur_plan.h:item_max_lm -- 24, 32 or 64)
    monkeypatch.setenv("OSH_LBA_ITEM_MAX", str(item_max))
    w = maker()
    st = plan_stats(w)
    k = observer_counts(w)
    assert st["blocks"] == int((k * (k + 1) // 2).sum())
    assert st["reduce_entries"] == w.n_free * (w.n_free + 1) // 2 + w.n_free
    # one record per (landmark, part pair)
    parts = np.maximum(1, (k + 7) // 8)
    assert st["recs"] == int((parts * (parts + 1) // 2).sum())
    assert st["sym"] <= st["items"]
    if (k > 8).any():
        assert st["items"] > st["sym"]
    else:
        assert st["items"] == st["sym"]
    # every landmark with an optimisable observer feeds the rhs of each of them exactly once per item
    assert st["ccontrib"] >= np.count_nonzero(np.bincount(w.edge_pose[w.edge_pose < w.n_free], minlength=w.n_free))


def test_plan_landmark_without_optimisable_observer():
    """A landmark seen only by fixed keyframes still needs its Dinv for the back-substitution."""
    w = synth.make_window(9, n_free=4, n_fixed=3, n_points=120, track_len=(2, 5))
    keep = ~((w.edge_point == 0) & (w.edge_pose < w.n_free))
    extra_pose = np.array([w.n_free, w.n_free + 1], dtype=np.int32)
    w.edge_pose = np.concatenate([w.edge_pose[keep], extra_pose])
    w.edge_point = np.concatenate([w.edge_point[keep], np.zeros(2, dtype=np.int32)])
    w.edge_kind = np.concatenate([w.edge_kind[keep], w.edge_kind[:2]])
    w.edge_obs = np.concatenate([w.edge_obs[keep], w.edge_obs[:2]])
    w.edge_info = np.concatenate([w.edge_info[keep], w.edge_info[:2]])
    # drop a possible duplicate (pose, landmark) pair created by the append
    _, first = np.unique(np.stack([w.edge_pose, w.edge_point], axis=1), axis=0, return_index=True)
    first.sort()
    for name in ("edge_pose", "edge_point", "edge_kind", "edge_obs", "edge_info"):
        setattr(w, name, getattr(w, name)[first])
    st = plan_stats(w)
    assert observer_counts(w)[0] == 0
    assert st["recs"] >= w.n_points


def test_plan_efficiency_of_the_headline_window():
    """The MFMA work of config 2 stays within 2x of the useful 6x6x3 products (tile padding + zero fill)."""
    st = plan_stats(synth.make_config2(100))      # (a single window: items of at most 24 landmarks)
    useful = st["blocks"] * 36 * 3
    issued = st["mfma"] * 16 * 16 * 4
    assert issued < 2.0 * useful
