"""ctypes driver of the C++ host layer (include/orbslam3_hip_host.h): builds the reference's pointer graph
(KeyFrame / MapPoint / Map, Frame) from flat arrays and calls ORB_SLAM3::Optimizer::LocalBundleAdjustment /
ORB_SLAM3::ORBmatcher::SearchByProjection through their own C++ signatures."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi, synth
from .synth import LbaWindow


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class HostGraph:
    """A map built from a synthetic LbaWindow: keyframe k <-> pose index k of the window, point j <-> point j.

    Keyframe ids: optimisable poses get ids 100+i (ascending with the pose index, so the Hessian order of the
    reference equals the window's order), fixed poses get ids 10+i (older keyframes).  The current keyframe is
    the newest optimisable one; every other optimisable keyframe is covisible with it."""

    def __init__(self, w: LbaWindow, init_kf_fixed: bool = False, inertial: bool = False, init_kf_id_index: int | None = None):
        self.lib = capi.load_library()
        self.w = w
        P, F = w.n_free, w.n_fixed
        self.kf_id = np.array([100 + i for i in range(P)] + [10 + i for i in range(F)], dtype=np.int64)
        self.mp_id = np.arange(w.n_points, dtype=np.int64) + 1000
        pose = _f32(w.pose_qt)
        cam5 = _f32(w.pose_cam[0])
        inv = _f32(synth.INV_LEVEL_SIGMA2)
        # a fisheye rig window: OSH_EDGE_BODY edges are right-camera observations, added after the left ones (set_rig below)
        body = w.edge_kind == capi.OSH_EDGE_BODY
        left = ~body
        all_octave = _i32(np.round(np.log(1.0 / w.edge_info) / np.log(1.44)).astype(np.int32))
        octave = _i32(all_octave[left])
        obs = _f32(w.edge_obs[left])
        obs[w.edge_kind[left] == capi.OSH_EDGE_MONO, 2] = -1.0
        init_id = int(self.kf_id[0]) if init_kf_fixed else -1 & 0x7FFFFFFF
        if init_kf_id_index is not None:      # the map's initial keyframe (Map::GetInitKFid / GetOriginKF) is keyframe number ...
            init_id = int(self.kf_id[init_kf_id_index])
        self.init_kf_fixed = init_kf_fixed
        mp_pos = _f32(w.points)
        self._keep = [pose, cam5, inv, octave, obs, mp_pos]
        self.g = C.c_void_p(self.lib.osh_host_graph_create(
            P + F, capi.ptr(self.kf_id, capi.c_int64_p), capi.ptr(pose, capi.c_float_p), capi.ptr(cam5, capi.c_float_p),
            capi.ptr(inv, capi.c_float_p), len(inv), w.n_points, capi.ptr(self.mp_id, capi.c_int64_p),
            capi.ptr(mp_pos, capi.c_float_p), int(left.sum()), capi.ptr(_i32(w.edge_pose[left]), capi.c_int32_p),
            capi.ptr(_i32(w.edge_point[left]), capi.c_int32_p), capi.ptr(obs, capi.c_float_p), capi.ptr(octave, capi.c_int32_p),
            init_id, int(inertial)))
        if w.kb8 is not None:     # monocular fisheye map: every keyframe's mpCamera is one KannalaBrandt8
            self.lib.osh_host_graph_set_fisheye(self.g, capi.ptr(_f32(w.kb8), capi.c_float_p))
        if w.cam2 is not None:    # fisheye stereo rig: right camera, Trl and the right-camera observations
            r_uv = _f32(w.edge_obs[body][:, :2])
            rc = self.lib.osh_host_graph_set_rig(self.g, capi.ptr(_f32(w.cam2), capi.c_float_p), capi.ptr(_f32(w.trl), capi.c_float_p),
                                                 int(body.sum()), capi.ptr(_i32(w.edge_pose[body]), capi.c_int32_p),
                                                 capi.ptr(_i32(w.edge_point[body]), capi.c_int32_p), capi.ptr(r_uv, capi.c_float_p),
                                                 capi.ptr(_i32(all_octave[body]), capi.c_int32_p))
            assert rc == 0
        self.cur = P - 1
        cov = _i32([i for i in range(P) if i != self.cur])
        self.lib.osh_host_graph_set_covisible(self.g, self.cur, len(cov), capi.ptr(cov, capi.c_int32_p))

    def close(self):
        if self.g:
            self.lib.osh_host_graph_destroy(self.g)
            self.g = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def pack(self):
        sizes = np.zeros(5, dtype=np.int32)
        null = [None] * 10
        rc = self.lib.osh_host_pack_lba(self.g, self.cur, capi.ptr(sizes, capi.c_int32_p), *[C.cast(None, t) for t in (
            capi.c_double_p, capi.c_double_p, capi.c_double_p, capi.c_int32_p, capi.c_int32_p, capi.c_uint8_p, capi.c_double_p,
            capi.c_double_p, capi.c_int64_p, capi.c_int64_p)])
        del null
        if rc != 0:
            return rc, sizes, None
        P, F, L, E = (int(x) for x in sizes[:4])
        out = dict(pose_qt=np.zeros((P + F, 7)), pose_cam=np.zeros((P + F, 5)), points=np.zeros((L, 3)),
                   edge_pose=np.zeros(E, dtype=np.int32), edge_point=np.zeros(E, dtype=np.int32), edge_kind=np.zeros(E, dtype=np.uint8),
                   edge_obs=np.zeros((E, 3)), edge_info=np.zeros(E), pose_kf_id=np.zeros(P + F, dtype=np.int64),
                   point_mp_id=np.zeros(L, dtype=np.int64))
        d, i32, u8, i64 = capi.c_double_p, capi.c_int32_p, capi.c_uint8_p, capi.c_int64_p
        rc = self.lib.osh_host_pack_lba(self.g, self.cur, capi.ptr(sizes, i32), capi.ptr(out["pose_qt"], d), capi.ptr(out["pose_cam"], d),
                                        capi.ptr(out["points"], d), capi.ptr(out["edge_pose"], i32), capi.ptr(out["edge_point"], i32),
                                        capi.ptr(out["edge_kind"], u8), capi.ptr(out["edge_obs"], d), capi.ptr(out["edge_info"], d),
                                        capi.ptr(out["pose_kf_id"], i64), capi.ptr(out["point_mp_id"], i64))
        return rc, sizes, out

    def packed_window(self) -> LbaWindow:
        """The osh_lba_problem the host layer builds for the current keyframe, as an LbaWindow (for the oracle)."""
        rc, sizes, o = self.pack()
        assert rc == 0, rc
        return LbaWindow(n_free=int(sizes[0]), n_fixed=int(sizes[1]), pose_qt=o["pose_qt"], pose_cam=o["pose_cam"], points=o["points"],
                         edge_pose=o["edge_pose"], edge_point=o["edge_point"], edge_kind=o["edge_kind"], edge_obs=o["edge_obs"],
                         edge_info=o["edge_info"], lambda_init=0.0, max_iterations=10, kb8=self._last_kb8(),
                         cam2=self._last_rig()[0], trl=self._last_rig()[1]).normalise(), o

    def _last_rig(self):
        c, t = np.zeros(8), np.zeros(7)
        return (c, t) if self.lib.osh_host_last_pack_rig(self.g, capi.ptr(c, capi.c_double_p), capi.ptr(t, capi.c_double_p)) else (None, None)

    def _last_kb8(self):
        k = np.zeros(4)
        return k if self.lib.osh_host_last_pack_kb8(self.g, capi.ptr(k, capi.c_double_p)) else None

    def run_lba(self, stop_flag: np.ndarray | None = None):
        counts = np.zeros(4, dtype=np.int32)
        rc = self.lib.osh_host_run_lba(self.g, self.cur, capi.ptr(stop_flag, capi.c_uint8_p) if stop_flag is not None else
                                       C.cast(None, capi.c_uint8_p), capi.ptr(counts, capi.c_int32_p))
        assert rc == 0
        return counts

    # ---- Optimizer::GlobalBundleAdjustemnt
    def packed_global_window(self, max_iterations=5, robust=True):
        """The osh_lba_problem the host layer builds for BundleAdjustment over the whole graph, as an LbaWindow."""
        sizes = np.zeros(5, dtype=np.int32)
        d, i32, u8, i64 = capi.c_double_p, capi.c_int32_p, capi.c_uint8_p, capi.c_int64_p
        rc = self.lib.osh_host_pack_gba(self.g, capi.ptr(sizes, i32), *[C.cast(None, t) for t in (d, d, d, i32, i32, u8, d, d, i64, i64)])
        assert rc == 0, rc
        P, F, L, E = (int(x) for x in sizes[:4])
        o = dict(pose_qt=np.zeros((P + F, 7)), pose_cam=np.zeros((P + F, 5)), points=np.zeros((L, 3)),
                 edge_pose=np.zeros(E, dtype=np.int32), edge_point=np.zeros(E, dtype=np.int32), edge_kind=np.zeros(E, dtype=np.uint8),
                 edge_obs=np.zeros((E, 3)), edge_info=np.zeros(E), pose_kf_id=np.zeros(P + F, dtype=np.int64),
                 point_mp_id=np.zeros(L, dtype=np.int64), n_not_included=int(sizes[4]))
        rc = self.lib.osh_host_pack_gba(self.g, capi.ptr(sizes, i32), capi.ptr(o["pose_qt"], d), capi.ptr(o["pose_cam"], d),
                                        capi.ptr(o["points"], d), capi.ptr(o["edge_pose"], i32), capi.ptr(o["edge_point"], i32),
                                        capi.ptr(o["edge_kind"], u8), capi.ptr(o["edge_obs"], d), capi.ptr(o["edge_info"], d),
                                        capi.ptr(o["pose_kf_id"], i64), capi.ptr(o["point_mp_id"], i64))
        assert rc == 0, rc
        w = LbaWindow(n_free=P, n_fixed=F, pose_qt=o["pose_qt"], pose_cam=o["pose_cam"], points=o["points"],
                      edge_pose=o["edge_pose"], edge_point=o["edge_point"], edge_kind=o["edge_kind"], edge_obs=o["edge_obs"],
                      edge_info=o["edge_info"], lambda_init=0.0, max_iterations=max_iterations, kb8=self._last_kb8(),
                      cam2=self._last_rig()[0], trl=self._last_rig()[1]).normalise()
        # const float thHuber2D = sqrt(5.99), thHuber3D = sqrt(7.815) (src/Optimizer.cc:130-131); none unless bRobust
        w.huber_mono = float(np.float32(np.sqrt(5.99))) if robust else float("inf")
        w.huber_stereo = float(np.float32(np.sqrt(7.815))) if robust else float("inf")
        return w, o

    def run_gba(self, n_iterations=5, n_loop_kf=0, robust=True, stop_flag=None):
        rc = self.lib.osh_host_run_gba(self.g, n_iterations, capi.ptr(stop_flag, capi.c_uint8_p) if stop_flag is not None else
                                       C.cast(None, capi.c_uint8_p), int(n_loop_kf), int(robust))
        assert rc == 0

    # ---- welding Optimizer::LocalBundleAdjustment(pMainKF, vpAdjustKF, vpFixedKF, pbStopFlag)
    def packed_welding_window(self, main, adjust, fixed):
        sizes = np.zeros(5, dtype=np.int32)
        d, i32, u8, i64 = capi.c_double_p, capi.c_int32_p, capi.c_uint8_p, capi.c_int64_p
        adj, fix = _i32(adjust), _i32(fixed)
        head = (self.g, int(main), len(adj), capi.ptr(adj, i32), len(fix), capi.ptr(fix, i32), capi.ptr(sizes, i32))
        rc = self.lib.osh_host_pack_welding(*head, *[C.cast(None, t) for t in (d, d, d, i32, i32, u8, d, d, i64, i64)])
        assert rc == 0, rc
        P, F, L, E = (int(x) for x in sizes[:4])
        o = dict(pose_qt=np.zeros((P + F, 7)), pose_cam=np.zeros((P + F, 5)), points=np.zeros((L, 3)),
                 edge_pose=np.zeros(E, dtype=np.int32), edge_point=np.zeros(E, dtype=np.int32), edge_kind=np.zeros(E, dtype=np.uint8),
                 edge_obs=np.zeros((E, 3)), edge_info=np.zeros(E), pose_kf_id=np.zeros(P + F, dtype=np.int64),
                 point_mp_id=np.zeros(L, dtype=np.int64))
        rc = self.lib.osh_host_pack_welding(*head, capi.ptr(o["pose_qt"], d), capi.ptr(o["pose_cam"], d), capi.ptr(o["points"], d),
                                            capi.ptr(o["edge_pose"], i32), capi.ptr(o["edge_point"], i32), capi.ptr(o["edge_kind"], u8),
                                            capi.ptr(o["edge_obs"], d), capi.ptr(o["edge_info"], d), capi.ptr(o["pose_kf_id"], i64),
                                            capi.ptr(o["point_mp_id"], i64))
        assert rc == 0, rc
        w = LbaWindow(n_free=P, n_fixed=F, pose_qt=o["pose_qt"], pose_cam=o["pose_cam"], points=o["points"],
                      edge_pose=o["edge_pose"], edge_point=o["edge_point"], edge_kind=o["edge_kind"], edge_obs=o["edge_obs"],
                      edge_info=o["edge_info"], lambda_init=0.0, max_iterations=5, kb8=self._last_kb8()).normalise()
        w.huber_mono = float(np.float32(np.sqrt(5.99)))      # const float thHuber2D = sqrt(5.99) (src/Optimizer.cc:3625)
        w.huber_stereo = float(np.float32(np.sqrt(7.815)))
        return w, o

    def run_welding(self, main, adjust, fixed, stop_flag=None):
        adj, fix = _i32(adjust), _i32(fixed)
        rc = self.lib.osh_host_run_welding(self.g, int(main), len(adj), capi.ptr(adj, capi.c_int32_p), len(fix), capi.ptr(fix, capi.c_int32_p),
                                           capi.ptr(stop_flag, capi.c_uint8_p) if stop_flag is not None else C.cast(None, capi.c_uint8_p))
        assert rc == 0

    def kf_pose_gba(self, i):
        o = np.zeros(7, dtype=np.float32)
        mark = self.lib.osh_host_get_kf_pose_gba(self.g, i, capi.ptr(o, capi.c_float_p))
        return int(mark), o

    def mp_pos_gba(self, j):
        o = np.zeros(3, dtype=np.float32)
        mark = self.lib.osh_host_get_mp_pos_gba(self.g, j, capi.ptr(o, capi.c_float_p))
        return int(mark), o

    def kf_pose(self, i):
        o = np.zeros(7, dtype=np.float32)
        self.lib.osh_host_get_kf_pose(self.g, i, capi.ptr(o, capi.c_float_p))
        return o

    def mp_pos(self, j):
        o = np.zeros(3, dtype=np.float32)
        self.lib.osh_host_get_mp_pos(self.g, j, capi.ptr(o, capi.c_float_p))
        return o


class HostFrame:
    def __init__(self, xy, octave, desc, angle=None, uright=None, pose_qt=None, mbf=float(synth.BF), mb=0.110078, kb8=None):
        self.lib = capi.load_library()
        n = len(octave)
        self.n = n
        pose = _f32(pose_qt if pose_qt is not None else [0, 0, 0, 1, 0, 0, 0])
        cam4 = _f32([synth.FX, synth.FY, synth.CX, synth.CY])
        a = _f32(angle) if angle is not None else None
        u = _f32(uright) if uright is not None else None
        self.f = C.c_void_p(self.lib.osh_host_frame_create(
            n, capi.ptr(_f32(xy), capi.c_float_p), capi.ptr(_i32(octave), capi.c_int32_p),
            capi.ptr(a, capi.c_float_p) if a is not None else C.cast(None, capi.c_float_p),
            capi.ptr(u, capi.c_float_p) if u is not None else C.cast(None, capi.c_float_p),
            capi.ptr(np.ascontiguousarray(desc, dtype=np.uint8), capi.c_uint8_p), capi.ptr(pose, capi.c_float_p),
            capi.ptr(cam4, capi.c_float_p), mbf, mb, synth.N_LEVELS, np.float32(synth.SCALE_FACTOR)))
        if kb8 is not None:      # monocular fisheye frame: mpCamera is a KannalaBrandt8
            self.lib.osh_host_frame_set_fisheye(self.f, capi.ptr(_f32(kb8), capi.c_float_p))

    def close(self):
        if self.f:
            self.lib.osh_host_frame_destroy(self.f)
            self.f = C.c_void_p()

    def set_rig(self, n_left, left_to_right, right_to_left, trl=(0, 0, 0, 1, -0.1, 0, 0)):
        """Fisheye stereo layout: keypoints [0, n_left) left camera, the rest right camera."""
        keep = [_i32(left_to_right), _i32(right_to_left), _f32(np.asarray(trl))]
        rc = self.lib.osh_host_frame_set_rig(self.f, int(n_left), capi.ptr(keep[0], capi.c_int32_p), capi.ptr(keep[1], capi.c_int32_p),
                                             capi.ptr(keep[2], capi.c_float_p))
        assert rc == 0

    def search_local_points_rig(self, mp_desc, in_l, proj_l, level_l, viewcos_l, in_r, proj_r, level_r, viewcos_r, n_obs=None, nnratio=0.8, th=1.0):
        n_mp = len(mp_desc)
        assign = -np.ones(self.n, dtype=np.int32)
        keep = [np.ascontiguousarray(mp_desc, dtype=np.uint8), np.ascontiguousarray(in_l, dtype=np.uint8), _f32(proj_l), _i32(level_l), _f32(viewcos_l),
                np.ascontiguousarray(in_r, dtype=np.uint8), _f32(proj_r), _i32(level_r), _f32(viewcos_r),
                _i32(n_obs) if n_obs is not None else None]
        u8, fp, i32 = capi.c_uint8_p, capi.c_float_p, capi.c_int32_p
        n = self.lib.osh_host_search_local_points_rig(self.f, n_mp, capi.ptr(keep[0], u8), capi.ptr(keep[1], u8), capi.ptr(keep[2], fp),
                                                      capi.ptr(keep[3], i32), capi.ptr(keep[4], fp), capi.ptr(keep[5], u8), capi.ptr(keep[6], fp),
                                                      capi.ptr(keep[7], i32), capi.ptr(keep[8], fp), capi.ptr(keep[9], i32), nnratio, th,
                                                      capi.ptr(assign, i32))
        return int(n), assign

    def search_local_points(self, mp_desc, proj_xy, level, viewcos=None, proj_xr=None, depth=None, n_obs=None, nnratio=0.8, th=1.0):
        n_mp = len(level)
        assign = -np.ones(self.n, dtype=np.int32)
        fp = capi.c_float_p

        def opt(a, typ, conv):
            return capi.ptr(conv(a), typ) if a is not None else C.cast(None, typ)
        keep = [np.ascontiguousarray(mp_desc, dtype=np.uint8), _f32(proj_xy), _i32(level)]
        n = self.lib.osh_host_search_local_points(self.f, n_mp, capi.ptr(keep[0], capi.c_uint8_p), capi.ptr(keep[1], fp),
                                                  opt(proj_xr, fp, _f32), capi.ptr(keep[2], capi.c_int32_p), opt(viewcos, fp, _f32),
                                                  opt(depth, fp, _f32), opt(n_obs, capi.c_int32_p, _i32), nnratio, th,
                                                  capi.ptr(assign, capi.c_int32_p))
        return n, assign

    def search_local_points_projected(self, mp_pos, mp_normal, mp_min_dist, mp_max_dist, viewing_cos_limit=0.5, mp_desc=None,
                                      n_obs=None, nnratio=0.8, th=1.0):
        """Frame::isInFrustum for every map point on the device (src/Frame.cc:513-587), then, when descriptors are given,
        SearchByProjection(F, vpMapPoints, th) as in Tracking::SearchLocalPoints (src/Tracking.cc:3411-3460)."""
        n_mp = len(mp_min_dist)
        fp, ip = capi.c_float_p, capi.c_int32_p
        keep = [_f32(mp_pos), _f32(mp_normal), _f32(mp_min_dist), _f32(mp_max_dist)]
        out = dict(in_view=np.zeros(n_mp, np.uint8), proj_xy=np.zeros((n_mp, 2), np.float32), proj_xr=np.zeros(n_mp, np.float32),
                   depth=np.zeros(n_mp, np.float32), view_cos=np.zeros(n_mp, np.float32), level=np.zeros(n_mp, np.int32))
        assign = -np.ones(self.n, dtype=np.int32) if mp_desc is not None else None
        nm = C.c_int32(0)
        desc = np.ascontiguousarray(mp_desc, dtype=np.uint8) if mp_desc is not None else None
        nobs = _i32(n_obs) if n_obs is not None else None
        n_in = self.lib.osh_host_frame_search_local_points_projected(
            self.f, n_mp, capi.ptr(keep[0], fp), capi.ptr(keep[1], fp), capi.ptr(keep[2], fp), capi.ptr(keep[3], fp), viewing_cos_limit,
            capi.ptr(out["in_view"], capi.c_uint8_p), capi.ptr(out["proj_xy"], fp), capi.ptr(out["proj_xr"], fp), capi.ptr(out["depth"], fp),
            capi.ptr(out["view_cos"], fp), capi.ptr(out["level"], ip), capi.ptr(desc, capi.c_uint8_p), capi.ptr(nobs, ip), nnratio, th,
            capi.ptr(assign, ip), C.cast(C.byref(nm), ip) if assign is not None else C.cast(None, ip))
        if n_in < 0:
            raise RuntimeError("osh_host_frame_search_local_points_projected failed: " + self.lib.osh_last_error().decode())
        out.update(n_in_view=n_in, assignment=assign, n_matches=nm.value)
        return out

    def search_last_frame(self, last: "HostFrame", last_mp, mp_pos, mp_desc, th=15.0, mono=True, check_ori=True):
        assign = -np.ones(self.n, dtype=np.int32)
        keep = [_i32(last_mp), _f32(mp_pos), np.ascontiguousarray(mp_desc, dtype=np.uint8)]
        n = self.lib.osh_host_search_last_frame(self.f, last.f, capi.ptr(keep[0], capi.c_int32_p), len(mp_pos),
                                                capi.ptr(keep[1], capi.c_float_p), capi.ptr(keep[2], capi.c_uint8_p), th, int(mono),
                                                int(check_ori), capi.ptr(assign, capi.c_int32_p))
        return n, assign


    def search_keyframe(self, kf_angle, kf_mp, mp_pos, mp_desc, mp_min_max_dist, mp_found=None, mp_bad=None, cur_mp=None,
                        th=10.0, orb_dist=100, check_ori=True):
        """ORBmatcher(0.9, check_ori).SearchByProjection(self, pKF, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1889-2010)."""
        assign = -np.ones(self.n, dtype=np.int32)
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
        keep = [_f32(kf_angle), _i32(kf_mp), _f32(mp_pos), u8(mp_desc), _f32(mp_min_max_dist),
                u8(mp_found if mp_found is not None else np.zeros(len(mp_pos))), u8(mp_bad if mp_bad is not None else np.zeros(len(mp_pos))),
                _i32(cur_mp if cur_mp is not None else -np.ones(self.n))]
        n = self.lib.osh_host_search_keyframe(self.f, len(kf_mp), capi.ptr(keep[0], capi.c_float_p), capi.ptr(keep[1], capi.c_int32_p),
                                              len(mp_pos), capi.ptr(keep[2], capi.c_float_p), capi.ptr(keep[3], capi.c_uint8_p),
                                              capi.ptr(keep[4], capi.c_float_p), capi.ptr(keep[5], capi.c_uint8_p),
                                              capi.ptr(keep[6], capi.c_uint8_p), capi.ptr(keep[7], capi.c_int32_p), th, int(orb_dist),
                                              int(check_ori), capi.ptr(assign, capi.c_int32_p))
        return n, assign


    def search_by_bow(self, kf_desc, kf_angle, kf_has_mp, kf_fv, f_fv, nnratio=0.7, check_ori=True):
        """ORBmatcher(nnratio, check_ori).SearchByBoW(pKF, F, vpMapPointMatches) (src/ORBmatcher.cc:223-420).  kf_fv / f_fv: feature
        vectors as (node_id[n], node_off[n + 1], node_feat) with ascending node ids.  Returns (nmatches, assignment[F.N])."""
        kf_desc = np.ascontiguousarray(kf_desc, dtype=np.uint8)
        keep = [kf_desc, _f32(kf_angle), np.ascontiguousarray(kf_has_mp, dtype=np.uint8)] + [_i32(a) for a in kf_fv] + [_i32(a) for a in f_fv]
        assign = np.zeros(self.n, dtype=np.int32)
        n = self.lib.osh_host_search_by_bow(self.f, len(kf_desc), capi.ptr(keep[0], capi.c_uint8_p), capi.ptr(keep[1], capi.c_float_p),
                                            capi.ptr(keep[2], capi.c_uint8_p), len(keep[3]), capi.ptr(keep[3], capi.c_int32_p),
                                            capi.ptr(keep[4], capi.c_int32_p), capi.ptr(keep[5], capi.c_int32_p), len(keep[6]),
                                            capi.ptr(keep[6], capi.c_int32_p), capi.ptr(keep[7], capi.c_int32_p), capi.ptr(keep[8], capi.c_int32_p),
                                            float(nnratio), int(check_ori), capi.ptr(assign, capi.c_int32_p))
        return n, assign

    def set_camera2(self, cam2):
        """Give the right camera of a fisheye stereo frame its own KannalaBrandt8 (fx fy cx cy k1..k4)."""
        assert self.lib.osh_host_frame_set_camera2(self.f, capi.ptr(_f32(cam2), capi.c_float_p)) == 0

    def pose_optimization(self, kp_mp, mp_pos):
        """Optimizer::PoseOptimization(&frame) (src/Optimizer.cc:815-1114); returns (n_inliers, pose_qt, mvbOutlier)."""
        pose = np.zeros(7, dtype=np.float32)
        outlier = np.zeros(self.n, dtype=np.uint8)
        keep = [_f32(mp_pos), _i32(kp_mp), _f32(synth.INV_LEVEL_SIGMA2)]
        n = self.lib.osh_host_frame_pose_optimization(self.f, len(mp_pos), capi.ptr(keep[0], capi.c_float_p), capi.ptr(keep[1], capi.c_int32_p),
                                                      capi.ptr(keep[2], capi.c_float_p), len(keep[2]), capi.ptr(pose, capi.c_float_p),
                                                      capi.ptr(outlier, capi.c_uint8_p))
        return n, pose, outlier

    def search_sim3(self, scw, mp_pos, mp_desc, mp_min_max_dist, mp_normal, mp_bad=None, matched_in=None, th=3, ratio_hamming=1.0,
                    with_keyframes=False):
        """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints[, vpPointsKFs], vpMatched[, vpMatchedKF], th, ratioHamming)
        (src/ORBmatcher.cc:427-646) with this frame turned into the keyframe."""
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
        out, out_kf = -np.ones(self.n, dtype=np.int32), -np.ones(self.n, dtype=np.int32)
        keep = [_f32(scw), _f32(mp_pos), u8(mp_desc), _f32(mp_min_max_dist), _f32(mp_normal),
                u8(mp_bad if mp_bad is not None else np.zeros(len(mp_pos))), _i32(matched_in if matched_in is not None else -np.ones(self.n))]
        fp = capi.c_float_p
        n = self.lib.osh_host_search_sim3(self.f, capi.ptr(keep[0], fp), len(mp_pos), capi.ptr(keep[1], fp), capi.ptr(keep[2], capi.c_uint8_p),
                                          capi.ptr(keep[3], fp), capi.ptr(keep[4], fp), capi.ptr(keep[5], capi.c_uint8_p),
                                          capi.ptr(keep[6], capi.c_int32_p), int(th), float(ratio_hamming), int(with_keyframes),
                                          capi.ptr(out, capi.c_int32_p), capi.ptr(out_kf, capi.c_int32_p))
        return n, out, out_kf


    def fuse(self, mp_pos, mp_desc, mp_min_max_dist, mp_normal, mp_nobs, slot_res, res_nobs, mp_bad=None, null_mask=None, res_bad=None, th=3.0):
        """ORBmatcher::Fuse(pKF, vpMapPoints, th) (src/ORBmatcher.cc:1148-1338) with this frame turned into the keyframe.  Returns
        nFused and a dict: slot (the keyframe's matches afterwards), cand_bad / cand_replaced / cand_nobs, res_bad / res_replaced /
        res_nobs (ids: candidate j -> j, resident r -> 100000 + r, none -> -1)."""
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
        n_mp, n_res = len(mp_pos), len(res_nobs)
        zeros = np.zeros(max(n_mp, n_res, 1))
        keep = [_f32(mp_pos), u8(mp_desc), _f32(mp_min_max_dist), _f32(mp_normal), u8(mp_bad if mp_bad is not None else zeros[:n_mp]),
                u8(null_mask if null_mask is not None else zeros[:n_mp]), _i32(mp_nobs), _i32(slot_res), _i32(res_nobs),
                u8(res_bad if res_bad is not None else zeros[:n_res])]
        out = dict(slot=-np.ones(self.n, dtype=np.int32), cand_bad=np.zeros(n_mp, dtype=np.uint8), cand_replaced=-np.ones(n_mp, dtype=np.int32),
                   cand_nobs=np.zeros(n_mp, dtype=np.int32), res_bad=np.zeros(max(n_res, 1), dtype=np.uint8),
                   res_replaced=-np.ones(max(n_res, 1), dtype=np.int32), res_nobs=np.zeros(max(n_res, 1), dtype=np.int32))
        fp, ip, bp = capi.c_float_p, capi.c_int32_p, capi.c_uint8_p
        n = self.lib.osh_host_fuse(self.f, n_mp, capi.ptr(keep[0], fp), capi.ptr(keep[1], bp), capi.ptr(keep[2], fp), capi.ptr(keep[3], fp),
                                   capi.ptr(keep[4], bp), capi.ptr(keep[5], bp), capi.ptr(keep[6], ip), n_res, capi.ptr(keep[7], ip),
                                   capi.ptr(keep[8], ip), capi.ptr(keep[9], bp), float(th), capi.ptr(out["slot"], ip), capi.ptr(out["cand_bad"], bp),
                                   capi.ptr(out["cand_replaced"], ip), capi.ptr(out["cand_nobs"], ip), capi.ptr(out["res_bad"], bp),
                                   capi.ptr(out["res_replaced"], ip), capi.ptr(out["res_nobs"], ip))
        for k in ("res_bad", "res_replaced", "res_nobs"):
            out[k] = out[k][:n_res]
        return n, out


    def fuse_sim3(self, scw, mp_pos, mp_desc, mp_min_max_dist, mp_normal, mp_nobs, slot_res, n_res, mp_bad=None, found_slot=None, res_bad=None, th=3.0):
        """ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (src/ORBmatcher.cc:1340-1455) with this frame turned into the
        keyframe.  Returns nFused, the keyframe's matches afterwards, vpReplacePoint as ids, Observations() of the candidates."""
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
        n_mp = len(mp_pos)
        keep = [_f32(scw), _f32(mp_pos), u8(mp_desc), _f32(mp_min_max_dist), _f32(mp_normal), u8(mp_bad if mp_bad is not None else np.zeros(n_mp)),
                _i32(mp_nobs), _i32(found_slot if found_slot is not None else -np.ones(n_mp)), _i32(slot_res),
                u8(res_bad if res_bad is not None else np.zeros(max(n_res, 1)))]
        slot, repl, nobs = -np.ones(self.n, dtype=np.int32), -np.ones(n_mp, dtype=np.int32), np.zeros(n_mp, dtype=np.int32)
        fp, ip, bp = capi.c_float_p, capi.c_int32_p, capi.c_uint8_p
        n = self.lib.osh_host_fuse_sim3(self.f, capi.ptr(keep[0], fp), n_mp, capi.ptr(keep[1], fp), capi.ptr(keep[2], bp), capi.ptr(keep[3], fp),
                                        capi.ptr(keep[4], fp), capi.ptr(keep[5], bp), capi.ptr(keep[6], ip), capi.ptr(keep[7], ip), int(n_res),
                                        capi.ptr(keep[8], ip), capi.ptr(keep[9], bp), float(th), capi.ptr(slot, ip), capi.ptr(repl, ip), capi.ptr(nobs, ip))
        return n, slot, repl, nobs


    def search_by_sim3(self, other: "HostFrame", s12, pts1: dict, pts2: dict, matches_in=None, th=7.5):
        """ORBmatcher(0.75, true).SearchBySim3(pKF1, pKF2, vpMatches12, S12, th) (src/ORBmatcher.cc:1457-1674), this frame as keyframe 1
        and ``other`` as keyframe 2.  ``pts_k``: pos [n,3], desc [n,32], min_max [n,2], bad [n], slot [N_k] (point index per keypoint
        slot or -1).  Returns nFound and, per slot of keyframe 1, the matched keyframe-2 point index or -1."""
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
        fp, ip, bp = capi.c_float_p, capi.c_int32_p, capi.c_uint8_p
        keep = [_f32(s12)]
        for p in (pts1, pts2):
            keep += [_f32(p["pos"]), u8(p["desc"]), _f32(p["min_max"]), u8(p["bad"]), _i32(p["slot"])]
        keep.append(_i32(matches_in if matches_in is not None else -np.ones(self.n)))
        out = -np.ones(self.n, dtype=np.int32)
        n = self.lib.osh_host_search_by_sim3(self.f, other.f, capi.ptr(keep[0], fp), float(th),
                                             len(pts1["pos"]), capi.ptr(keep[1], fp), capi.ptr(keep[2], bp), capi.ptr(keep[3], fp), capi.ptr(keep[4], bp), capi.ptr(keep[5], ip),
                                             len(pts2["pos"]), capi.ptr(keep[6], fp), capi.ptr(keep[7], bp), capi.ptr(keep[8], fp), capi.ptr(keep[9], bp), capi.ptr(keep[10], ip),
                                             capi.ptr(keep[11], ip), capi.ptr(out, ip))
        return n, out


def _quat_from_R(R):
    return synth._quat_from_R(np.asarray(R, dtype=np.float64))


class HostInertialGraph:
    """A map with IMU state built from a synth_inertial.LibaWindow: keyframe k <-> pose index k of the window
    (temporal keyframes ids 100+i, the fixed predecessor id 99, fixed observers ids 10+i)."""

    def __init__(self, w, no_prev=()):
        """``no_prev``: pose indices of temporal keyframes whose mPrevKF stays null (the chain of keyframes breaks before them;
        the window must not hold an inertial link that ends there)."""
        self.lib = capi.load_library()
        self.w = w
        assert not np.isin(w.link_cur, list(no_prev)).any()
        N, K = w.n_opt, w.n_opt + w.n_fixed_imu + w.n_fixed
        self.kf_id = np.array([100 + i for i in range(N)] + [99] * w.n_fixed_imu + [10 + i for i in range(w.n_fixed)], dtype=np.int64)
        self.mp_id = np.arange(w.n_points, dtype=np.int64) + 1000
        pose = np.zeros((K, 7), dtype=np.float32)
        for k in range(K):
            pose[k, :4] = _quat_from_R(w.pose_Rcw.reshape(-1, 3, 3)[k])
            pose[k, 4:] = w.pose_tcw.reshape(-1, 3)[k]
        cam5, inv = _f32(w.cam), _f32(synth.INV_LEVEL_SIGMA2)
        # a fisheye rig window: OSH_EDGE_RIGHT edges are right-camera observations, registered after the left ones (set_rig below)
        right = w.edge_kind == capi.OSH_EDGE_RIGHT
        left = ~right
        all_octave = _i32(np.round(np.log(1.0 / w.edge_info) / np.log(1.44)).astype(np.int32))
        octave = _i32(all_octave[left])
        obs, mp_pos = _f32(w.edge_obs[left]), _f32(w.points)
        self.g = C.c_void_p(self.lib.osh_host_graph_create(
            K, capi.ptr(self.kf_id, capi.c_int64_p), capi.ptr(pose, capi.c_float_p), capi.ptr(cam5, capi.c_float_p),
            capi.ptr(inv, capi.c_float_p), len(inv), w.n_points, capi.ptr(self.mp_id, capi.c_int64_p), capi.ptr(mp_pos, capi.c_float_p),
            int(left.sum()), capi.ptr(_i32(w.edge_pose[left]), capi.c_int32_p), capi.ptr(_i32(w.edge_point[left]), capi.c_int32_p),
            capi.ptr(obs, capi.c_float_p), capi.ptr(octave, capi.c_int32_p), -1 & 0x7FFFFFFF, 1))
        n_imu = N + w.n_fixed_imu
        kf_index = _i32(np.arange(n_imu))
        prev = _i32([-1 if i in no_prev else N if (i == 0 and w.n_fixed_imu) else i - 1 for i in range(N)] + [-1] * w.n_fixed_imu)
        vel = _f32(w.vel.reshape(-1, 3)[:n_imu])
        bias6 = _f32(np.concatenate([w.bias_a.reshape(-1, 3)[:n_imu], w.bias_g.reshape(-1, 3)[:n_imu]], axis=1))
        pre = np.zeros((n_imu, capi.OSH_PREINT_FLOATS), dtype=np.float32)
        cov = np.zeros((n_imu, 225), dtype=np.float32)
        for l in range(w.n_links):
            pre[int(w.link_cur[l])] = w.link_preint[l]
            cov[int(w.link_cur[l])] = w.gt["link_cov"][l].ravel()
        Tbc = w.gt["Tbc"]
        tbc_qt = _f32(np.concatenate([_quat_from_R(Tbc[:3, :3]), Tbc[:3, 3]]))
        self.lib.osh_host_graph_set_inertial(self.g, n_imu, capi.ptr(kf_index, capi.c_int32_p), capi.ptr(prev, capi.c_int32_p),
                                             capi.ptr(vel, capi.c_float_p), capi.ptr(bias6, capi.c_float_p), capi.ptr(pre, capi.c_float_p),
                                             capi.ptr(cov, capi.c_float_p), capi.ptr(tbc_qt, capi.c_float_p))
        if w.kb8 is not None:     # monocular fisheye map: every keyframe's mpCamera is one KannalaBrandt8
            self.lib.osh_host_graph_set_fisheye(self.g, capi.ptr(_f32(w.kb8), capi.c_float_p))
        if w.cam2 is not None:    # fisheye stereo rig: right camera, Trl and the right-camera observations
            r_uv = _f32(w.edge_obs[right][:, :2])
            rc = self.lib.osh_host_graph_set_rig(self.g, capi.ptr(_f32(w.cam2), capi.c_float_p), capi.ptr(_f32(w.gt["trl_qt"]), capi.c_float_p),
                                                 int(right.sum()), capi.ptr(_i32(w.edge_pose[right]), capi.c_int32_p),
                                                 capi.ptr(_i32(w.edge_point[right]), capi.c_int32_p), capi.ptr(r_uv, capi.c_float_p),
                                                 capi.ptr(_i32(all_octave[right]), capi.c_int32_p))
            assert rc == 0
        self.cur = N - 1

    def close(self):
        if self.g:
            self.lib.osh_host_graph_destroy(self.g)
            self.g = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def packed_window(self, large=False, rec_init=False):
        """The osh_liba_problem the host layer builds, copied into a LibaWindow (for the oracle) + id maps."""
        from .synth_inertial import LibaWindow
        p = capi.LibaProblem()
        K = len(self.kf_id)
        kid, mid = np.zeros(K, dtype=np.int64), np.zeros(self.w.n_points, dtype=np.int64)
        rc = self.lib.osh_host_pack_liba(self.g, self.cur, int(large), int(rec_init), C.byref(p), capi.ptr(kid, capi.c_int64_p),
                                         capi.ptr(mid, capi.c_int64_p))
        assert rc == 0, rc

        w = self._window_of(p, lambda_init=1e-2 if large else 1.0, max_iterations=4 if large else 10)
        Kp = p.n_opt + p.n_fixed_imu + p.n_fixed
        return w, kid[:Kp], mid[:p.n_points]

    @staticmethod
    def _window_of(p, lambda_init, max_iterations):
        from .synth_inertial import LibaWindow

        def arr(ptr, n, dt=np.float64):
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt).copy() if n else np.zeros(0, dtype=dt)
        Kp, NV, L, E, NL = p.n_opt + p.n_fixed_imu + p.n_fixed, p.n_opt + p.n_fixed_imu, p.n_points, p.n_edges, p.n_links
        w = LibaWindow(
            n_opt=p.n_opt, n_fixed_imu=p.n_fixed_imu, n_fixed=p.n_fixed, pose_Rcw=arr(p.pose_Rcw, Kp * 9), pose_tcw=arr(p.pose_tcw, Kp * 3),
            pose_Rwb=arr(p.pose_Rwb, Kp * 9), pose_twb=arr(p.pose_twb, Kp * 3), Rcb=arr(p.Rcb, 9), tcb=arr(p.tcb, 3), tbc=arr(p.tbc, 3),
            cam=arr(p.cam, 5), vel=arr(p.vel, NV * 3), bias_g=arr(p.bias_g, NV * 3), bias_a=arr(p.bias_a, NV * 3),
            points=arr(p.points, L * 3).reshape(-1, 3), edge_pose=arr(p.edge_pose, E, np.int32), edge_point=arr(p.edge_point, E, np.int32),
            edge_kind=arr(p.edge_kind, E, np.uint8), edge_obs=arr(p.edge_obs, E * 3).reshape(-1, 3), edge_info=arr(p.edge_info, E),
            link_prev=arr(p.link_prev, NL, np.int32), link_cur=arr(p.link_cur, NL, np.int32),
            link_preint=arr(p.link_preint, NL * capi.OSH_PREINT_FLOATS, np.float32).reshape(NL, -1),
            link_info=arr(p.link_info, NL * 81).reshape(NL, 81), link_info_g=arr(p.link_info_g, NL * 9).reshape(NL, 9),
            link_info_a=arr(p.link_info_a, NL * 9).reshape(NL, 9), link_robust=arr(p.link_robust, NL, np.uint8),
            lambda_init=lambda_init, max_iterations=max_iterations,
            kb8=arr(p.kb8, 4) if p.kb8 else None, cam2=arr(p.cam2, 8) if p.cam2 else None, trl=arr(p.trl, 12) if p.trl else None,
            link_bias=arr(p.link_bias, NL, np.int32) if p.link_bias else None).normalise()
        return w

    def run(self, large=False, rec_init=False):
        return self.lib.osh_host_run_liba(self.g, self.cur, int(large), int(rec_init))

    def packed_full(self, its, fix_local=False, init=False, prior_g=1e2, prior_a=1e6):
        """The problem Optimizer::FullInertialBA(map, its, bFixLocal, ., ., bInit) solves, as (LibaWindow, keyframe ids, map point ids,
        number of keyframes no edge touches); the host layer's return code instead when it declines the case."""
        p = capi.LibaProblem()
        kid, mid = np.zeros(len(self.kf_id) + 1, dtype=np.int64), np.zeros(self.w.n_points, dtype=np.int64)
        idle = np.zeros(1, dtype=np.int32)
        rc = self.lib.osh_host_pack_full_inertial(self.g, int(its), int(fix_local), int(init), prior_g, prior_a, C.byref(p), capi.ptr(kid, capi.c_int64_p),
                                                  capi.ptr(mid, capi.c_int64_p), capi.ptr(idle, capi.c_int32_p))
        if rc != 0:
            return rc
        return self._window_of(p, 1e-5, int(its)), kid[:p.n_opt + p.n_fixed_imu + p.n_fixed], mid[:p.n_points], int(idle[0])

    def run_full(self, its, loop_id=0, fix_local=False, init=False, prior_g=1e2, prior_a=1e6):
        return self.lib.osh_host_run_full_inertial(self.g, int(its), int(fix_local), int(loop_id), int(init), prior_g, prior_a)

    def packed_merge(self, curr, merge):
        """The problem Optimizer::MergeInertialBA(kf[curr], kf[merge], ...) solves + (temporal keyframe ids, covisible keyframe ids) in
        the reference's order."""
        p = capi.LibaProblem()
        K = len(self.kf_id)
        kid, mid = np.zeros(K, dtype=np.int64), np.zeros(self.w.n_points, dtype=np.int64)
        sets, tid, cid = np.zeros(2, dtype=np.int32), np.zeros(K, dtype=np.int64), np.zeros(K, dtype=np.int64)
        rc = self.lib.osh_host_pack_merge_inertial(self.g, int(curr), int(merge), C.byref(p), capi.ptr(kid, capi.c_int64_p), capi.ptr(mid, capi.c_int64_p),
                                                   capi.ptr(sets, capi.c_int32_p), capi.ptr(tid, capi.c_int64_p), capi.ptr(cid, capi.c_int64_p))
        if rc != 0:
            return rc
        return (self._window_of(p, 1e3, 8), kid[:p.n_opt + p.n_fixed_imu + p.n_fixed], mid[:p.n_points], tid[:sets[0]], cid[:sets[1]])

    def run_merge(self, curr, merge):
        """Optimizer::MergeInertialBA; returns corrPoses as {keyframe id: [qx qy qz qw tx ty tz s]}."""
        K = len(self.kf_id)
        ids, sim = np.zeros(K, dtype=np.int64), np.zeros((K, 8))
        n = self.lib.osh_host_run_merge_inertial(self.g, int(curr), int(merge), K, capi.ptr(ids, capi.c_int64_p), capi.ptr(sim, capi.c_double_p))
        assert 0 <= n <= K, n
        return {int(ids[i]): sim[i].copy() for i in range(n)}

    def kf_inertial_gba(self, i):
        """(mnBAGlobalForKF, mTcwGBA [qx qy qz qw tx ty tz], mVwbGBA, mBiasGBA [ba, bg]) of keyframe i."""
        pose, vel, bias = np.zeros(7, dtype=np.float32), np.zeros(3, dtype=np.float32), np.zeros(6, dtype=np.float32)
        self.lib.osh_host_get_kf_pose_gba(self.g, i, capi.ptr(pose, capi.c_float_p))
        lid = self.lib.osh_host_get_kf_inertial_gba(self.g, i, capi.ptr(vel, capi.c_float_p), capi.ptr(bias, capi.c_float_p))
        return int(lid), pose, vel, bias

    def kf_pose(self, i):
        o = np.zeros(7, dtype=np.float32)
        self.lib.osh_host_get_kf_pose(self.g, i, capi.ptr(o, capi.c_float_p))
        return o

    def kf_velocity(self, i):
        o = np.zeros(3, dtype=np.float32)
        self.lib.osh_host_get_kf_velocity(self.g, i, capi.ptr(o, capi.c_float_p))
        return o

    def kf_bias(self, i):
        o = np.zeros(6, dtype=np.float32)
        self.lib.osh_host_get_kf_bias(self.g, i, capi.ptr(o, capi.c_float_p))
        return o

    def mp_pos(self, j):
        o = np.zeros(3, dtype=np.float32)
        self.lib.osh_host_get_mp_pos(self.g, j, capi.ptr(o, capi.c_float_p))
        return o


def host_preintegrate(acc, gyr, dt, bias6, nga6, walk6):
    lib = capi.load_library()
    acc, gyr = _f32(acc), _f32(gyr)
    rec, cov = np.zeros(capi.OSH_PREINT_FLOATS, dtype=np.float32), np.zeros(225, dtype=np.float32)
    fp = capi.c_float_p
    lib.osh_host_preintegrate(len(acc), capi.ptr(acc, fp), capi.ptr(gyr, fp), np.float32(dt), capi.ptr(_f32(bias6), fp),
                              capi.ptr(_f32(nga6), fp), capi.ptr(_f32(walk6), fp), capi.ptr(rec, fp), capi.ptr(cov, fp))
    return rec, cov.reshape(15, 15)


def host_inertial_information(cov):
    lib = capi.load_library()
    out = np.zeros(81)
    lib.osh_host_inertial_information(capi.ptr(_f32(cov).ravel(), capi.c_float_p), capi.ptr(out, capi.c_double_p))
    return out.reshape(9, 9)


class HostPoseiFrame:
    """A tracked frame + its IMU link built from a synth_inertial.PoseiFrame, for Optimizer::PoseInertialOptimizationLastKeyFrame
    (mode 0) / LastFrame (mode 1) through the reference signatures.  Keypoint k <-> edge order[k] of the flat frame (a fisheye rig
    frame lists its left keypoints first)."""

    def __init__(self, f):
        self.lib = capi.load_library()
        self.f = f
        right = f.edge_kind == capi.OSH_EDGE_RIGHT
        rig = f.cam2 is not None
        self.order = np.concatenate([np.nonzero(~right)[0], np.nonzero(right)[0]]) if rig else np.arange(f.n_edges)
        o = self.order
        n_left = int((~right).sum()) if rig else -1
        Tbc = np.eye(4)
        Tbc[:3, :3], Tbc[:3, 3] = f.Rcb.reshape(3, 3).T, f.tbc
        tbc_qt = _f32(np.concatenate([_quat_from_R(Tbc[:3, :3]), Tbc[:3, 3]]))

        def tcw_qt(Rwb, twb):      # Tcw = Tcb * Tbw
            Rcb = f.Rcb.reshape(3, 3)
            Rcw = Rcb @ Rwb.reshape(3, 3).T
            return _f32(np.concatenate([_quat_from_R(Rcw), Rcb @ (-Rwb.reshape(3, 3).T @ twb) + f.tcb]))
        pose = _f32(np.concatenate([_quat_from_R(f.Rcw.reshape(3, 3)), f.tcw]))
        prev_pose = tcw_qt(f.prev_Rwb, f.prev_twb)
        octave = _i32(np.round(np.log(1.0 / f.edge_info[o]) / np.log(1.44)).astype(np.int32))
        uright = _f32(np.where(f.edge_kind[o] == capi.OSH_EDGE_STEREO, f.edge_obs[o, 2], -1.0))
        cov = np.zeros((15, 15), dtype=np.float32)
        cov[:9, :9] = np.linalg.inv(f.info_inertial.reshape(9, 9))
        cov[9:12, 9:12] = np.linalg.inv(f.info_g.reshape(3, 3))
        cov[12:15, 12:15] = np.linalg.inv(f.info_a.reshape(3, 3))
        fp = lambda a: capi.ptr(_f32(a), capi.c_float_p) if a is not None else None   # noqa: E731
        dp = lambda a: capi.ptr(np.ascontiguousarray(a, dtype=np.float64), capi.c_double_p) if a is not None else None   # noqa: E731
        self._keep = [pose, prev_pose, octave, uright, tbc_qt, cov]
        self.h = C.c_void_p(self.lib.osh_host_posei_create(
            int(f.mode), f.n_edges, fp(f.edge_obs[o, :2]), capi.ptr(octave, capi.c_int32_p), capi.ptr(uright, capi.c_float_p), n_left,
            capi.ptr(pose, capi.c_float_p), fp(f.cam), fp(f.kb8), fp(f.cam2), fp(f.gt["trl_qt"]) if rig else None, fp(synth.INV_LEVEL_SIGMA2),
            len(synth.INV_LEVEL_SIGMA2), fp(f.points[o]), capi.ptr(np.ascontiguousarray(f.edge_close[o]), capi.c_uint8_p),
            capi.ptr(tbc_qt, capi.c_float_p), fp(f.vel), fp(np.concatenate([f.bias_a, f.bias_g])), capi.ptr(prev_pose, capi.c_float_p), fp(f.prev_vel),
            fp(np.concatenate([f.prev_bias_a, f.prev_bias_g])), capi.ptr(np.ascontiguousarray(f.preint), capi.c_float_p),
            capi.ptr(np.ascontiguousarray(cov.ravel()), capi.c_float_p), dp(f.prior_Rwb), dp(f.prior_twb), dp(f.prior_vel), dp(f.prior_bg), dp(f.prior_ba),
            dp(f.prior_H)))
        assert self.h

    def close(self):
        if self.h:
            self.lib.osh_host_posei_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def packed(self, rec_init=False):
        """The osh_posei_problem the host layer builds, copied into a PoseiFrame (for the oracle) + the keypoint of every edge."""
        from .synth_inertial import PoseiFrame
        p = capi.PoseiProblem()
        kp = np.zeros(self.f.n_edges, dtype=np.int32)
        rc = self.lib.osh_host_posei_pack(self.h, int(rec_init), C.byref(p), capi.ptr(kp, capi.c_int32_p))
        assert rc == 0, rc

        def arr(ptr, n, dt=np.float64):
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt).copy() if ptr else None
        E = p.n_edges
        g = PoseiFrame(
            mode=p.mode, Rcw=arr(p.Rcw, 9), tcw=arr(p.tcw, 3), Rwb=arr(p.Rwb, 9), twb=arr(p.twb, 3), vel=arr(p.vel, 3), bias_g=arr(p.bias_g, 3),
            bias_a=arr(p.bias_a, 3), prev_Rwb=arr(p.prev_Rwb, 9), prev_twb=arr(p.prev_twb, 3), prev_vel=arr(p.prev_vel, 3),
            prev_bias_g=arr(p.prev_bias_g, 3), prev_bias_a=arr(p.prev_bias_a, 3), Rcb=arr(p.Rcb, 9), tcb=arr(p.tcb, 3), tbc=arr(p.tbc, 3), cam=arr(p.cam, 5),
            preint=arr(p.preint, capi.OSH_PREINT_FLOATS, np.float32), info_inertial=arr(p.info_inertial, 81), info_g=arr(p.info_g, 9), info_a=arr(p.info_a, 9),
            points=arr(p.points, E * 3).reshape(-1, 3), edge_kind=arr(p.edge_kind, E, np.uint8), edge_obs=arr(p.edge_obs, E * 3).reshape(-1, 3),
            edge_info=arr(p.edge_info, E), edge_close=arr(p.edge_close, E, np.uint8), prior_Rwb=arr(p.prior_Rwb, 9), prior_twb=arr(p.prior_twb, 3),
            prior_vel=arr(p.prior_vel, 3), prior_bg=arr(p.prior_bg, 3), prior_ba=arr(p.prior_ba, 3), prior_H=arr(p.prior_H, 225), kb8=arr(p.kb8, 4),
            cam2=arr(p.cam2, 8), trl=arr(p.trl, 12), rec_init=bool(p.rec_init), huber_mono=p.huber_mono, huber_stereo=p.huber_stereo, huber_prior=p.huber_prior,
            chi2_mono=tuple(p.chi2_mono), chi2_stereo=tuple(p.chi2_stereo), iterations=tuple(p.iterations)).normalise()
        return g, kp[:E]

    def run(self, rec_init=False):
        pose, Rwb, twb, vel, bias = (np.zeros(k, dtype=np.float32) for k in (7, 9, 3, 3, 6))
        outlier = np.zeros(self.f.n_edges, dtype=np.uint8)
        H = np.zeros(225)
        gone = C.c_int32(0)
        n = self.lib.osh_host_posei_run(self.h, int(rec_init), capi.ptr(pose, capi.c_float_p), capi.ptr(Rwb, capi.c_float_p), capi.ptr(twb, capi.c_float_p),
                                        capi.ptr(vel, capi.c_float_p), capi.ptr(bias, capi.c_float_p), capi.ptr(outlier, capi.c_uint8_p),
                                        capi.ptr(H, capi.c_double_p), C.byref(gone))
        return dict(n=n, pose_qt=pose, Rwb=Rwb.reshape(3, 3), twb=twb, vel=vel, bias_a=bias[:3], bias_g=bias[3:], outlier=outlier, H=H.reshape(15, 15),
                    prev_cpi_deleted=bool(gone.value))


def search_by_bow_keyframes(desc1, angle1, has_mp1, fv1, desc2, angle2, has_mp2, fv2, nnratio=0.7, check_ori=True):
    """ORBmatcher(nnratio, check_ori).SearchByBoW(pKF1, pKF2, vpMatches12) (src/ORBmatcher.cc:765-905) on two keyframes built from flat
    features; returns (nmatches, match12[n1]) with match12[i] = feature of keyframe 2 whose map point was matched to feature i."""
    lib = capi.load_library()
    d1, d2 = np.ascontiguousarray(desc1, dtype=np.uint8), np.ascontiguousarray(desc2, dtype=np.uint8)
    keep = [d1, _f32(angle1), np.ascontiguousarray(has_mp1, dtype=np.uint8)] + [_i32(a) for a in fv1] + \
           [d2, _f32(angle2), np.ascontiguousarray(has_mp2, dtype=np.uint8)] + [_i32(a) for a in fv2]
    m = np.zeros(len(d1), dtype=np.int32)
    n = lib.osh_host_search_by_bow_kf(len(d1), capi.ptr(keep[0], capi.c_uint8_p), capi.ptr(keep[1], capi.c_float_p), capi.ptr(keep[2], capi.c_uint8_p),
                                      len(keep[3]), capi.ptr(keep[3], capi.c_int32_p), capi.ptr(keep[4], capi.c_int32_p), capi.ptr(keep[5], capi.c_int32_p),
                                      len(d2), capi.ptr(keep[6], capi.c_uint8_p), capi.ptr(keep[7], capi.c_float_p), capi.ptr(keep[8], capi.c_uint8_p),
                                      len(keep[9]), capi.ptr(keep[9], capi.c_int32_p), capi.ptr(keep[10], capi.c_int32_p), capi.ptr(keep[11], capi.c_int32_p),
                                      float(nnratio), int(check_ori), capi.ptr(m, capi.c_int32_p))
    return n, m


def search_for_triangulation(kp1, octave1, desc1, has_mp1, pose1_qt, fv1, kp2, octave2, desc2, has_mp2, pose2_qt, fv2, only_stereo=False, coarse=False,
                             check_ori=True):
    """ORBmatcher(0.6, check_ori).SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse) (src/ORBmatcher.cc:907-1146) on
    two pinhole keyframes built from flat features (kp = x, y, angle, uright); returns (nmatches, match12[n1])."""
    lib = capi.load_library()
    u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
    cam4 = _f32([synth.FX, synth.FY, synth.CX, synth.CY])
    keep = [cam4, _f32(kp1), _i32(octave1), u8(desc1), u8(has_mp1), _f32(pose1_qt)] + [_i32(a) for a in fv1] + \
           [_f32(kp2), _i32(octave2), u8(desc2), u8(has_mp2), _f32(pose2_qt)] + [_i32(a) for a in fv2]
    m = -np.ones(len(octave1), dtype=np.int32)
    fp, ip, bp = capi.c_float_p, capi.c_int32_p, capi.c_uint8_p
    n = lib.osh_host_search_for_triangulation(capi.ptr(keep[0], fp), int(synth.N_LEVELS), float(synth.SCALE_FACTOR), len(octave1), capi.ptr(keep[1], fp),
                                              capi.ptr(keep[2], ip), capi.ptr(keep[3], bp), capi.ptr(keep[4], bp), capi.ptr(keep[5], fp), len(keep[6]),
                                              capi.ptr(keep[6], ip), capi.ptr(keep[7], ip), capi.ptr(keep[8], ip), len(octave2), capi.ptr(keep[9], fp),
                                              capi.ptr(keep[10], ip), capi.ptr(keep[11], bp), capi.ptr(keep[12], bp), capi.ptr(keep[13], fp), len(keep[14]),
                                              capi.ptr(keep[14], ip), capi.ptr(keep[15], ip), capi.ptr(keep[16], ip), int(only_stereo), int(coarse),
                                              int(check_ori), capi.ptr(m, ip))
    return n, m


def search_for_initialization(f1, f2, prev_xy, window=100, nnratio=0.9, check_ori=True):
    """ORBmatcher(nnratio, check_ori).SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (src/ORBmatcher.cc:648-763)
    on two HostFrames; returns (nmatches, vnMatches12, the updated vbPrevMatched)."""
    lib = capi.load_library()
    prev = np.array(prev_xy, dtype=np.float32)
    m = -np.ones(f1.n, dtype=np.int32)
    n = lib.osh_host_search_for_initialization(f1.f, f2.f, capi.ptr(prev, capi.c_float_p), int(window), float(nnratio), int(check_ori),
                                               capi.ptr(m, capi.c_int32_p))
    return n, m, prev
