"""Python driver over the ORB matching C-ABI (include/orbslam3_hip.h, osh_orb_*)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .synth import OrbPair


class OrbMatcher:
    def __init__(self, device: int = 0):
        self.lib = capi.load_library()
        self.ctx = C.c_void_p()
        capi.check(self.lib.osh_orb_create(device, C.byref(self.ctx)), "osh_orb_create", self.lib)
        self._shape = None
        self._keep = None

    def close(self):
        if self.ctx:
            self.lib.osh_orb_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def upload(self, pairs: list[OrbPair], windowed: bool = False):
        n_pairs = len(pairs)
        nq, nt = pairs[0].query_desc.shape[0], pairs[0].train_desc.shape[0]
        q = np.ascontiguousarray(np.stack([p.query_desc for p in pairs]), dtype=np.uint8)
        t = np.ascontiguousarray(np.stack([p.train_desc for p in pairs]), dtype=np.uint8)
        lev = np.ascontiguousarray(np.stack([p.train_level for p in pairs]), dtype=np.int32)
        b = capi.OrbBatch()
        b.n_pairs, b.n_query, b.n_train = n_pairs, nq, nt
        b.query_desc, b.train_desc = capi.ptr(q, capi.c_uint8_p), capi.ptr(t, capi.c_uint8_p)
        b.train_level = capi.ptr(lev, capi.c_int32_p)
        keep = [q, t, lev]
        if windowed:
            off = np.ascontiguousarray(np.stack([p.cand_off for p in pairs]), dtype=np.int32)
            idx = np.ascontiguousarray(np.concatenate([p.cand_idx for p in pairs]), dtype=np.int32)
            lens = np.array([p.cand_idx.shape[0] for p in pairs], dtype=np.int64)
            base = np.ascontiguousarray(np.concatenate([[0], np.cumsum(lens)[:-1]]), dtype=np.int64)
            if idx.size == 0:
                idx = np.zeros(1, dtype=np.int32)
            b.cand_off, b.cand_idx = capi.ptr(off, capi.c_int32_p), capi.ptr(idx, capi.c_int32_p)
            b.pair_cand_base = capi.ptr(base, capi.c_int64_p)
            keep += [off, idx, base]
            self._list_total = int(lens.sum())
        self._keep = keep
        self._shape = (n_pairs, nq)
        self._n_train = nt
        capi.check(self.lib.osh_orb_upload(self.ctx, C.byref(b)), "osh_orb_upload", self.lib)

    def upload_grid(self, query_desc, train_desc, train_level, train_xy, query_window, query_levels, train_uright=None,
                    train_skip=None, query_uright=None):
        """One frame pair whose candidates are generated on the device from the train frame's grid
        (Frame::GetFeaturesInArea, src/Frame.cc:658-722): query_window [nq,3] = x, y, r; query_levels [nq,2] = min, max."""
        from . import synth
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        q = np.ascontiguousarray(query_desc, dtype=np.uint8)
        t = np.ascontiguousarray(train_desc, dtype=np.uint8)
        lev = np.ascontiguousarray(train_level, dtype=np.int32)
        b = capi.OrbBatch()
        b.n_pairs, b.n_query, b.n_train = 1, q.shape[0], t.shape[0]
        b.query_desc, b.train_desc, b.train_level = capi.ptr(q, capi.c_uint8_p), capi.ptr(t, capi.c_uint8_p), capi.ptr(lev, capi.c_int32_p)
        g = capi.OrbGrid()
        keep = [q, t, lev, f32(train_xy), f32(query_window), np.ascontiguousarray(query_levels, dtype=np.int32)]
        g.train_xy, g.query_window, g.query_levels = capi.ptr(keep[3], capi.c_float_p), capi.ptr(keep[4], capi.c_float_p), capi.ptr(keep[5], capi.c_int32_p)
        if train_uright is not None:
            keep.append(f32(train_uright)); g.train_uright = capi.ptr(keep[-1], capi.c_float_p)
        if train_skip is not None:
            keep.append(np.ascontiguousarray(train_skip, dtype=np.uint8)); g.train_skip = capi.ptr(keep[-1], capi.c_uint8_p)
        if query_uright is not None:
            keep.append(f32(query_uright)); g.query_uright = capi.ptr(keep[-1], capi.c_float_p)
        g.min_x, g.min_y = 0.0, 0.0
        g.cell_w_inv = np.float32(synth.FRAME_GRID_COLS) / np.float32(synth.IMG_W)
        g.cell_h_inv = np.float32(synth.FRAME_GRID_ROWS) / np.float32(synth.IMG_H)
        g.cols, g.rows = synth.FRAME_GRID_COLS, synth.FRAME_GRID_ROWS
        self._keep = keep
        self._shape = (1, q.shape[0])
        self._n_train = t.shape[0]
        capi.check(self.lib.osh_orb_upload_grid(self.ctx, C.byref(b), C.byref(g)), "osh_orb_upload_grid", self.lib)

    def frustum(self, frame: "capi.FrustumFrame", pos, normal, min_dist, max_dist) -> dict:
        """Frame::isInFrustum (src/Frame.cc:513-587) for every map point of `pos` [n,3]; see osh_orb_frustum."""
        args, res, outs = frustum_args(pos, normal, min_dist, max_dist)
        capi.check(self.lib.osh_orb_frustum(self.ctx, C.byref(frame), C.byref(args[0]), C.byref(res)), "osh_orb_frustum", self.lib)
        return outs

    def list_distances(self):
        """osh_orb_list_distances: the Hamming distance of every (query, candidate) entry of the uploaded lists, pairs concatenated."""
        out = np.zeros(max(self._list_total, 1), dtype=np.int32)
        capi.check(self.lib.osh_orb_list_distances(self.ctx, capi.ptr(out, capi.c_int32_p)), "osh_orb_list_distances", self.lib)
        return out[:self._list_total]

    def match(self):
        capi.check(self.lib.osh_orb_match(self.ctx), "osh_orb_match", self.lib)

    def match_local_points(self, nn_ratio: float = 0.8, th_high: int = 100, occupied=None, query_blocks=None, n_train: int | None = None):
        """SearchByProjection(Frame&, vector<MapPoint*>&) with the sequential slot occupancy resolved on the device
        (osh_orb_match_local_points): (n_matches[n_pairs], assignment[n_pairs, n_train], query_slot[n_pairs, n_query], rounds)."""
        n_pairs, n_query = self._shape
        n_train = int(n_train if n_train is not None else self._n_train)
        occ = None if occupied is None else np.ascontiguousarray(occupied, dtype=np.uint8).reshape(n_pairs, n_train)
        blk = None if query_blocks is None else np.ascontiguousarray(query_blocks, dtype=np.uint8).reshape(n_pairs, n_query)
        assign = np.zeros((n_pairs, n_train), dtype=np.int32)
        n = np.zeros(n_pairs, dtype=np.int32)
        slot = np.zeros((n_pairs, n_query), dtype=np.int32)
        rounds = C.c_int32(0)
        capi.check(self.lib.osh_orb_match_local_points(self.ctx, float(nn_ratio), int(th_high), capi.ptr(occ, capi.c_uint8_p),
                                                       capi.ptr(blk, capi.c_uint8_p), capi.ptr(assign, capi.c_int32_p),
                                                       capi.ptr(n, capi.c_int32_p), capi.ptr(slot, capi.c_int32_p), C.byref(rounds)),
                   "osh_orb_match_local_points", self.lib)
        return n, assign, slot, int(rounds.value)

    def download(self) -> dict:
        names = ["best_idx", "best_dist", "second_dist", "best_level", "second_level", "second_idx"]
        outs = [np.zeros(self._shape, dtype=np.int32) for _ in names]
        capi.check(self.lib.osh_orb_download(self.ctx, *[capi.ptr(o, capi.c_int32_p) for o in outs]), "osh_orb_download", self.lib)
        return dict(zip(names, outs))

    def search(self, pairs: list[OrbPair], windowed: bool = False) -> dict:
        self.upload(pairs, windowed)
        self.match()
        return self.download()

    def distance_matrix(self, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint8)
        b = np.ascontiguousarray(b, dtype=np.uint8)
        out = np.zeros((a.shape[0], b.shape[0]), dtype=np.int32)
        capi.check(self.lib.osh_orb_distance_matrix(self.ctx, a.shape[0], b.shape[0], capi.ptr(a, capi.c_uint8_p),
                                                    capi.ptr(b, capi.c_uint8_p), capi.ptr(out, capi.c_int32_p)),
                   "osh_orb_distance_matrix", self.lib)
        return out

    def set_profiling(self, enable: bool):
        capi.check(self.lib.osh_orb_set_profiling(self.ctx, int(enable)), "osh_orb_set_profiling", self.lib)

    def profile(self):
        n, ms = C.c_int64(0), C.c_double(0)
        capi.check(self.lib.osh_orb_get_profile(self.ctx, C.byref(n), C.byref(ms)), "osh_orb_get_profile", self.lib)
        return int(n.value), float(ms.value)


    def resolve_profile(self):
        n, ms = C.c_int64(0), C.c_double(0)
        capi.check(self.lib.osh_orb_get_resolve_profile(self.ctx, C.byref(n), C.byref(ms)), "osh_orb_get_resolve_profile", self.lib)
        return int(n.value), float(ms.value)


def accept_local_points(res: dict, pair_index: int, nn_ratio: float = 0.8, th_high: int = 100) -> np.ndarray:
    """Acceptance rule of SearchByProjection(Frame&, vector<MapPoint*>&), src/ORBmatcher.cc:123-139,
    applied independently per query (no occupancy): bool mask of accepted queries."""
    bd = res["best_dist"][pair_index]
    sd = res["second_dist"][pair_index]
    bl, sl = res["best_level"][pair_index], res["second_level"][pair_index]
    ratio_fail = (bl == sl) & (bd.astype(np.float32) > np.float32(nn_ratio) * sd.astype(np.float32))
    return (bd <= th_high) & ~ratio_fail


def replay_local_points(res: dict, pair_index: int, rescan, nn_ratio: float = 0.8, th_high: int = 100,
                        occupied: np.ndarray | None = None, n_train: int | None = None):
    """Host side of SearchByProjection(Frame&, vector<MapPoint*>&) (src/ORBmatcher.cc:43-141) on top of the
    device search (SURVEY.md 8a "bit-exactness rule"): walk the queries in order; a query whose best or
    second-best slot was claimed by an earlier accepted query is re-scanned by `rescan(q, occupied)`
    -> (best_idx, best_dist, second_dist, best_level, second_level); occupancy only ever removes candidates,
    so every other query keeps its device result.  Returns (nmatches, assignment[n_train], n_rescans)."""
    bi, bd, sd = res["best_idx"][pair_index], res["best_dist"][pair_index], res["second_dist"][pair_index]
    bl, sl, si = res["best_level"][pair_index], res["second_level"][pair_index], res["second_idx"][pair_index]
    n_train = int(n_train if n_train is not None else max(int(bi.max()), int(si.max())) + 1)
    occ = np.zeros(n_train, dtype=np.uint8) if occupied is None else occupied
    pre_occupied = bool(occ.any())
    assign = -np.ones(n_train, dtype=np.int32)
    nmatches = rescans = 0
    ratio = np.float32(nn_ratio)
    for q in range(bi.shape[0]):
        b, d1, d2, l1, l2 = int(bi[q]), int(bd[q]), int(sd[q]), int(bl[q]), int(sl[q])
        if pre_occupied or (b >= 0 and occ[b]) or (si[q] >= 0 and occ[si[q]]):
            b, d1, d2, l1, l2 = rescan(q, occ)
            rescans += 1
        if b < 0 or d1 > th_high:
            continue
        if l1 == l2 and np.float32(d1) > ratio * np.float32(d2):
            continue
        assign[b] = q
        occ[b] = 1
        nmatches += 1
    return nmatches, assign, rescans


def frustum_frame(Rcw, tcw, fx, fy, cx, cy, bf, bounds, log_scale_factor, n_scale_levels, viewing_cos_limit=0.5, kb8=None):
    """osh_frustum_frame from a float32 camera pose; Ow = -Rcw^T tcw formed in float32 like Frame::UpdatePoseMatrices
    (src/Frame.cc:298-307)."""
    R = np.asarray(Rcw, dtype=np.float32).reshape(3, 3)
    t = np.asarray(tcw, dtype=np.float32).reshape(3)
    Ow = (-(R.T.astype(np.float32) @ t)).astype(np.float32)
    f = capi.FrustumFrame()
    f.Rcw[:] = [float(x) for x in R.reshape(9)]
    f.tcw[:] = [float(x) for x in t]
    f.Ow[:] = [float(x) for x in Ow]
    f.fx, f.fy, f.cx, f.cy, f.bf = fx, fy, cx, cy, bf
    f.min_x, f.max_x, f.min_y, f.max_y = bounds
    f.log_scale_factor, f.n_scale_levels, f.viewing_cos_limit = log_scale_factor, n_scale_levels, viewing_cos_limit
    if kb8 is not None:      # KannalaBrandt8 frame (monocular fisheye)
        f.fisheye = 1
        f.kb8[:] = [float(np.float32(k)) for k in kb8]
    return f


def frustum_args(pos, normal, min_dist, max_dist):
    """ctypes argument / result structs of osh_orb_frustum (also taken by the oracle) and the dict of output arrays."""
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    pos, normal, min_dist, max_dist = f32(pos), f32(normal), f32(min_dist), f32(max_dist)
    n = pos.shape[0]
    pts = capi.FrustumPoints()
    pts.n = n
    pts.pos, pts.normal = capi.ptr(pos, capi.c_float_p), capi.ptr(normal, capi.c_float_p)
    pts.min_dist, pts.max_dist = capi.ptr(min_dist, capi.c_float_p), capi.ptr(max_dist, capi.c_float_p)
    outs = dict(stage=np.zeros(n, np.uint8), proj_x=np.zeros(n, np.float32), proj_y=np.zeros(n, np.float32),
                proj_xr=np.zeros(n, np.float32), depth=np.zeros(n, np.float32), view_cos=np.zeros(n, np.float32),
                level=np.zeros(n, np.int32))
    res = capi.FrustumResult()
    res.stage = capi.ptr(outs["stage"], capi.c_uint8_p)
    for k in ("proj_x", "proj_y", "proj_xr", "depth", "view_cos"):
        setattr(res, k, capi.ptr(outs[k], capi.c_float_p))
    res.level = capi.ptr(outs["level"], capi.c_int32_p)
    return (pts, pos, normal, min_dist, max_dist), res, outs
