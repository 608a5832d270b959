"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: see oracle/lba_oracle.c.  The struct layouts are shared with
the product header (include/orbslam3_hip.h) through orb_slam3_study_kr_amd.capi.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from orb_slam3_study_kr_amd import capi
from orb_slam3_study_kr_amd.synth import LbaResultArrays, LbaWindow

HERE = Path(__file__).resolve().parent
_libs: dict = {}


def build(native: bool = False) -> Path:
    """Compile the oracle with gcc (portable x86-64-v3, or -march=native in place)."""
    out = HERE / ("liboracle_native.so" if native else "liboracle.so")
    args = ["make", "-s", "-C", str(HERE), f"OUT={out}"]
    if native:
        args.append("ARCH=native")
    subprocess.run(args, check=True)
    return out


def load(native: bool = False) -> C.CDLL:
    key = "native" if native else "portable"
    if key in _libs:
        return _libs[key]
    path = HERE / ("liboracle_native.so" if native else "liboracle.so")
    if native or not path.exists():
        path = build(native)
    lib = C.CDLL(str(path))
    d, i32, u8 = capi.c_double_p, capi.c_int32_p, capi.c_uint8_p
    lib.oracle_lba_solve.restype = C.c_int
    lib.oracle_lba_solve.argtypes = [C.POINTER(capi.LbaProblem), C.POINTER(capi.LbaResult)]
    lib.oracle_pose_optimize.restype = C.c_int
    lib.oracle_pose_optimize.argtypes = [C.POINTER(capi.PoseProblem), C.POINTER(capi.PoseResult)]
    lib.oracle_frustum.restype = None
    lib.oracle_frustum.argtypes = [C.POINTER(capi.FrustumFrame), C.POINTER(capi.FrustumPoints), C.POINTER(capi.FrustumResult)]
    lib.oracle_lba_linearize.restype = C.c_int
    lib.oracle_lba_linearize.argtypes = [C.POINTER(capi.LbaProblem)] + [d] * 7
    lib.oracle_lba_schur_step.restype = C.c_int
    lib.oracle_lba_schur_step.argtypes = [C.POINTER(capi.LbaProblem), C.c_double, d, d, d]
    lib.oracle_pose_oplus.restype = None
    lib.oracle_pose_oplus.argtypes = [d, d]
    lib.oracle_edge_error.restype = None
    lib.oracle_edge_error.argtypes = [C.c_int, d, d, d, d, d]
    lib.oracle_edge_jacobians.restype = None
    lib.oracle_edge_jacobians.argtypes = [C.c_int, d, d, d, d, d]
    lib.oracle_edge_error_kb8.restype = None
    lib.oracle_edge_error_kb8.argtypes = [d] * 6
    lib.oracle_edge_jacobians_kb8.restype = None
    lib.oracle_edge_jacobians_kb8.argtypes = [d] * 6
    lib.oracle_edge_depth_positive.restype = C.c_int
    lib.oracle_edge_depth_positive.argtypes = [d, d]
    lib.oracle_huber.restype = None
    lib.oracle_huber.argtypes = [C.c_double, C.c_double, d]
    lib.oracle_ldlt_solve.restype = C.c_int
    lib.oracle_ldlt_solve.argtypes = [C.c_int, d, d, d, d]
    lib.oracle_descriptor_distance.restype = C.c_int
    lib.oracle_descriptor_distance.argtypes = [u8, u8]
    lib.oracle_distance_matrix.restype = None
    lib.oracle_distance_matrix.argtypes = [C.c_int, C.c_int, u8, u8, i32]
    lib.oracle_orb_search.restype = None
    lib.oracle_orb_search.argtypes = [C.c_int, C.c_int, u8, u8, i32, i32, i32, u8, i32, i32, i32, i32, i32]
    lib.oracle_orb_match_local_points.restype = C.c_int
    lib.oracle_orb_match_local_points.argtypes = [C.c_int, C.c_int, u8, u8, i32, i32, i32, C.c_float, C.c_int, u8, i32]
    lib.oracle_orb_match_last_frame.restype = C.c_int
    lib.oracle_orb_match_last_frame.argtypes = [C.c_int, C.c_int, u8, u8, i32, i32, capi.c_float_p, capi.c_float_p,
                                                C.c_int, C.c_int, u8, i32]
    lib.oracle_liba_solve.restype = C.c_int
    lib.oracle_liba_solve.argtypes = [C.POINTER(capi.LibaProblem), C.POINTER(capi.LibaResult)]
    lib.oracle_liba_linearize.restype = C.c_int
    lib.oracle_liba_linearize.argtypes = [C.POINTER(capi.LibaProblem), d, d, d, d, d]
    lib.oracle_liba_inertial_edge.restype = C.c_int
    lib.oracle_liba_inertial_edge.argtypes = [C.POINTER(capi.LibaProblem), C.c_int, d, d]
    lib.oracle_exp_so3.restype = None
    lib.oracle_exp_so3.argtypes = [d, d]
    lib.oracle_log_so3.restype = None
    lib.oracle_log_so3.argtypes = [d, d]
    _libs[key] = lib
    return lib


def _d(a):
    return capi.ptr(a, capi.c_double_p)


# ------------------------------------------------------------------ local BA
def lba_solve(w: LbaWindow, native: bool = False) -> LbaResultArrays:
    lib = load(native)
    res = LbaResultArrays(w)
    prob = w.as_struct()
    rc = lib.oracle_lba_solve(C.byref(prob), C.byref(res.struct))
    if rc != 0:
        raise RuntimeError(f"oracle_lba_solve failed: {rc}")
    return res.read_scalars(res.struct)


def lba_linearize(w: LbaWindow):
    lib = load()
    P, L, E = w.n_free, w.n_points, w.n_edges
    out = dict(Hpp=np.zeros((P, 6, 6)), bp=np.zeros((P, 6)), Hll=np.zeros((L, 3, 3)), bl=np.zeros((L, 3)),
               Hpl=np.zeros((E, 6, 3)), chi2=np.zeros(E))
    rc2 = C.c_double(0)
    prob = w.as_struct()
    rc = lib.oracle_lba_linearize(C.byref(prob), _d(out["Hpp"]), _d(out["bp"]), _d(out["Hll"]), _d(out["bl"]),
                                  _d(out["Hpl"]), _d(out["chi2"]), C.cast(C.byref(rc2), capi.c_double_p))
    if rc != 0:
        raise RuntimeError(f"oracle_lba_linearize failed: {rc}")
    out["robust_chi2"] = rc2.value
    return out


def lba_schur_step(w: LbaWindow, lam: float):
    lib = load()
    n = 6 * w.n_free
    S = np.zeros((n, n))
    bs = np.zeros(n)
    x = np.zeros(n + 3 * w.n_points)
    prob = w.as_struct()
    rc = lib.oracle_lba_schur_step(C.byref(prob), lam, _d(S), _d(bs), _d(x))
    if rc != 0:
        raise RuntimeError(f"oracle_lba_schur_step failed: {rc}")
    return S, bs, x


def pose_oplus(update, qt):
    lib = load()
    u = np.ascontiguousarray(update, dtype=np.float64)
    o = np.array(qt, dtype=np.float64)
    lib.oracle_pose_oplus(_d(u), _d(o))
    return o


def edge_error(kind, qt, cam, X, obs):
    lib = load()
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (qt, cam, X, obs)]
    err = np.zeros(3)
    lib.oracle_edge_error(int(kind), _d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), _d(err))
    return err


def edge_jacobians(kind, qt, cam, X):
    lib = load()
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (qt, cam, X)]
    Jxi, Jxj = np.zeros((3, 3)), np.zeros((3, 6))
    lib.oracle_edge_jacobians(int(kind), _d(a[0]), _d(a[1]), _d(a[2]), _d(Jxi), _d(Jxj))
    return Jxi, Jxj


def edge_error_kb8(qt, cam, kb, X, obs):
    lib = load()
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (qt, cam, kb, X, obs)]
    err = np.zeros(3)
    lib.oracle_edge_error_kb8(_d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), _d(a[4]), _d(err))
    return err


def edge_jacobians_kb8(qt, cam, kb, X):
    lib = load()
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (qt, cam, kb, X)]
    Jxi, Jxj = np.zeros((3, 3)), np.zeros((3, 6))
    lib.oracle_edge_jacobians_kb8(_d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), _d(Jxi), _d(Jxj))
    return Jxi, Jxj


def huber(e, delta):
    lib = load()
    rho = np.zeros(3)
    lib.oracle_huber(float(e), float(delta), _d(rho))
    return rho


def ldlt_solve(A, b):
    lib = load()
    A = np.array(A, dtype=np.float64)
    n = A.shape[0]
    b = np.ascontiguousarray(b, dtype=np.float64)
    x, tmp = np.zeros(n), np.zeros(n)
    ok = lib.oracle_ldlt_solve(n, _d(A), _d(b), _d(x), _d(tmp))
    return bool(ok), x


# ------------------------------------------------------------------ ORB matching
def _u8(a):
    return capi.ptr(a, capi.c_uint8_p)


def _i32(a):
    return capi.ptr(a, capi.c_int32_p)


def descriptor_distance(a, b) -> int:
    lib = load()
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return int(lib.oracle_descriptor_distance(_u8(a), _u8(b)))


def distance_matrix(a, b):
    lib = load()
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.int32)
    lib.oracle_distance_matrix(a.shape[0], b.shape[0], _u8(a), _u8(b), _i32(out))
    return out


def orb_search(query, train, train_level=None, cand_off=None, cand_idx=None, occupied=None, native=False):
    lib = load(native)
    nq, nt = query.shape[0], train.shape[0]
    outs = [np.zeros(nq, dtype=np.int32) for _ in range(5)]
    lib.oracle_orb_search(nq, nt, _u8(query), _u8(train), _i32(train_level), _i32(cand_off), _i32(cand_idx),
                          _u8(occupied), *[_i32(o) for o in outs])
    return dict(zip(["best_idx", "best_dist", "second_dist", "best_level", "second_level"], outs))


def orb_match_local_points(query, train, train_level=None, cand_off=None, cand_idx=None, nn_ratio=0.8, th_high=100,
                           occupied=None):
    lib = load()
    nq, nt = query.shape[0], train.shape[0]
    occ = np.zeros(nt, dtype=np.uint8) if occupied is None else np.array(occupied, dtype=np.uint8)
    assign = -np.ones(nt, dtype=np.int32)
    n = lib.oracle_orb_match_local_points(nq, nt, _u8(query), _u8(train), _i32(train_level), _i32(cand_off),
                                          _i32(cand_idx), C.c_float(nn_ratio), th_high, _u8(occ), _i32(assign))
    return int(n), assign, occ


def orb_match_local_points_rig(query, desc, n_left, level_left, level_right, in_l, candl, in_r, candr, left_to_right, right_to_left,
                               nn_ratio=0.8, th_high=100, occupied=None):
    """SearchByProjection(Frame&, vector<MapPoint*>) on a fisheye stereo frame: both camera passes, sequential (orb_oracle.c)."""
    lib = load()
    nq, n = query.shape[0], desc.shape[0]
    occ = np.zeros(n, dtype=np.uint8) if occupied is None else np.array(occupied, dtype=np.uint8)
    assign = -np.ones(n, dtype=np.int32)
    cnt = lib.oracle_orb_match_local_points_rig(nq, int(n_left), int(n - n_left), _u8(query), _u8(desc), _i32(level_left), _i32(level_right),
                                                _u8(np.asarray(in_l, dtype=np.uint8)), _i32(candl[0]), _i32(candl[1]),
                                                _u8(np.asarray(in_r, dtype=np.uint8)), _i32(candr[0]), _i32(candr[1]),
                                                _i32(left_to_right), _i32(right_to_left), C.c_float(nn_ratio), th_high, _u8(occ), _i32(assign))
    return int(cnt), assign, occ


def orb_match_last_frame(query, train, cand_off, cand_idx, query_angle, train_angle, th_high=100,
                         check_orientation=True, occupied=None):
    lib = load()
    nq, nt = query.shape[0], train.shape[0]
    occ = np.zeros(nt, dtype=np.uint8) if occupied is None else np.array(occupied, dtype=np.uint8)
    assign = -np.ones(nt, dtype=np.int32)
    qa = np.ascontiguousarray(query_angle, dtype=np.float32)
    ta = np.ascontiguousarray(train_angle, dtype=np.float32)
    n = lib.oracle_orb_match_last_frame(nq, nt, _u8(query), _u8(train), _i32(cand_off), _i32(cand_idx),
                                        capi.ptr(qa, capi.c_float_p), capi.ptr(ta, capi.c_float_p), th_high,
                                        int(check_orientation), _u8(occ), _i32(assign))
    return int(n), assign, occ


# ------------------------------------------------------------------ frustum projection
def frustum(frame, pos, normal, min_dist, max_dist, native: bool = False) -> dict:
    from orb_slam3_study_kr_amd.orb import frustum_args
    lib = load(native)
    args, res, outs = frustum_args(pos, normal, min_dist, max_dist)
    lib.oracle_frustum(C.byref(frame), C.byref(args[0]), C.byref(res))
    return outs


# ------------------------------------------------------------------ pose-only optimisation
def pose_optimize(f, native: bool = False):
    from orb_slam3_study_kr_amd.synth import PoseResultArrays
    lib = load(native)
    res = PoseResultArrays(f)
    prob = f.as_struct()
    rc = lib.oracle_pose_optimize(C.byref(prob), C.byref(res.struct))
    if rc != 0:
        raise RuntimeError(f"oracle_pose_optimize failed: {rc}")
    return res.read_scalars(res.struct)


# ------------------------------------------------------------------ local inertial BA
def liba_solve(w, native: bool = False):
    from orb_slam3_study_kr_amd.synth_inertial import LibaResultArrays
    lib = load(native)
    res = LibaResultArrays(w)
    prob = w.as_struct()
    rc = lib.oracle_liba_solve(C.byref(prob), C.byref(res.struct))
    if rc != 0:
        raise RuntimeError(f"oracle_liba_solve failed: {rc}")
    return res.read_scalars(res.struct)


def liba_linearize(w):
    lib = load()
    n = 15 * w.n_opt
    H, b = np.zeros((n, n)), np.zeros(n + 3 * w.n_points)
    Hll, Hpl = np.zeros((w.n_points, 3, 3)), np.zeros((w.n_edges, 6, 3))
    chi = C.c_double(0)
    prob = w.as_struct()
    lib.oracle_liba_linearize(C.byref(prob), _d(H), _d(b), _d(Hll), _d(Hpl), C.cast(C.byref(chi), capi.c_double_p))
    return dict(H=H, b=b, Hll=Hll, Hpl=Hpl, chi2=chi.value)


def liba_inertial_edge(w, link):
    lib = load()
    r, J = np.zeros(9), np.zeros((9, 24))
    prob = w.as_struct()
    lib.oracle_liba_inertial_edge(C.byref(prob), int(link), _d(r), _d(J))
    return r, J


def exp_so3(wv):
    lib = load()
    wv = np.ascontiguousarray(wv, dtype=np.float64)
    R = np.zeros((3, 3))
    lib.oracle_exp_so3(_d(wv), _d(R))
    return R


def log_so3(R):
    lib = load()
    R = np.ascontiguousarray(R, dtype=np.float64)
    o = np.zeros(3)
    lib.oracle_log_so3(_d(R), _d(o))
    return o


def orb_match_last_frame_rig(query, desc, n_left, candl, candr, query_angle, angle_left, angle_right, th_high=100, check_orientation=True):
    """SearchByProjection(CurrentFrame, LastFrame) with a fisheye stereo current frame: both camera searches (orb_oracle.c)."""
    lib = load()
    nq, n = query.shape[0], desc.shape[0]
    occ = np.zeros(n, dtype=np.uint8)
    assign = -np.ones(n, dtype=np.int32)
    qa, al, ar = (np.ascontiguousarray(a, dtype=np.float32) for a in (query_angle, angle_left, angle_right))
    fp = capi.c_float_p
    cnt = lib.oracle_orb_match_last_frame_rig(nq, int(n_left), int(n - n_left), _u8(query), _u8(desc), _i32(candl[0]), _i32(candl[1]),
                                              _i32(candr[0]), _i32(candr[1]), capi.ptr(qa, fp), capi.ptr(al, fp), capi.ptr(ar, fp), th_high, int(check_orientation), _u8(occ), _i32(assign))
    return int(cnt), assign, occ


def posei_optimize(f):
    """Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame restatement (liba_oracle.c)."""
    from orb_slam3_study_kr_amd import synth_inertial as si
    lib = load()
    lib.oracle_posei_optimize.restype = C.c_int
    lib.oracle_posei_optimize.argtypes = [C.POINTER(capi.PoseiProblem), C.POINTER(capi.PoseiResult)]
    prob = f.as_struct()
    res = si.PoseiResultArrays(f)
    rc = lib.oracle_posei_optimize(C.byref(prob), C.byref(res.struct))
    if rc != 0:
        raise RuntimeError(f"oracle_posei_optimize failed: {rc}")
    return res.read(res.struct, f.mode)


def posei_linearize(f):
    lib = load()
    d = C.POINTER(C.c_double)
    lib.oracle_posei_linearize.restype = C.c_int
    lib.oracle_posei_linearize.argtypes = [C.POINTER(capi.PoseiProblem), d, d]
    n = 30 if f.mode == 1 else 15
    H, b = np.zeros((n, n)), np.zeros(n)
    prob = f.as_struct()
    lib.oracle_posei_linearize(C.byref(prob), _d(H), _d(b))
    return H, b


def marginalize_previous(H30):
    lib = load()
    d = C.POINTER(C.c_double)
    lib.oracle_marginalize_previous.restype = None
    lib.oracle_marginalize_previous.argtypes = [d, d]
    H30 = np.ascontiguousarray(H30, dtype=np.float64)
    out = np.zeros((15, 15))
    lib.oracle_marginalize_previous(_d(H30), _d(out))
    return out


def constraint_pose_imu_H(H15):
    lib = load()
    lib.oracle_constraint_pose_imu_H.restype = None
    lib.oracle_constraint_pose_imu_H.argtypes = [C.POINTER(C.c_double)]
    H = np.array(H15, dtype=np.float64).reshape(15, 15).copy()
    lib.oracle_constraint_pose_imu_H(_d(H))
    return H


def orb_search_by_bow(kf_desc, f_desc, kf_has_mp, kf_fv, f_fv, kf_angle, f_angle, n_left_f=-1, nn_ratio=0.7, th_low=50, check_ori=True):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) restated sequentially (orb_oracle.c); feature vectors as (node_id, node_off, node_feat)."""
    lib = load()
    i32 = C.POINTER(C.c_int32)
    lib.oracle_orb_search_by_bow.restype = C.c_int
    lib.oracle_orb_search_by_bow.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8),
                                             C.c_int, i32, i32, i32, C.c_int, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                             C.c_float, C.c_int, C.c_int, i32]
    kf_desc, f_desc = np.ascontiguousarray(kf_desc, dtype=np.uint8), np.ascontiguousarray(f_desc, dtype=np.uint8)
    has = np.ascontiguousarray(kf_has_mp, dtype=np.uint8)
    kfv = [np.ascontiguousarray(a, dtype=np.int32) for a in kf_fv]
    ffv = [np.ascontiguousarray(a, dtype=np.int32) for a in f_fv]
    ka, fa = np.ascontiguousarray(kf_angle, dtype=np.float32), np.ascontiguousarray(f_angle, dtype=np.float32)
    assign = -np.ones(f_desc.shape[0], dtype=np.int32)
    n = lib.oracle_orb_search_by_bow(kf_desc.shape[0], f_desc.shape[0], int(n_left_f), _u8(kf_desc), _u8(f_desc), _u8(has),
                                     len(kfv[0]), _i32(kfv[0]), _i32(kfv[1]), _i32(kfv[2]), len(ffv[0]), _i32(ffv[0]), _i32(ffv[1]), _i32(ffv[2]),
                                     ka.ctypes.data_as(C.POINTER(C.c_float)), fa.ctypes.data_as(C.POINTER(C.c_float)),
                                     C.c_float(nn_ratio), int(th_low), int(check_ori), _i32(assign))
    return int(n), assign


def orb_search_by_bow_kf(desc1, desc2, has_mp1, has_mp2, fv1, fv2, angle1, angle2, lim1=-1, lim2=-1, nn_ratio=0.7, th_low=50, check_ori=True):
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...) restated sequentially (orb_oracle.c)."""
    lib = load()
    i32 = C.POINTER(C.c_int32)
    u8 = C.POINTER(C.c_uint8)
    lib.oracle_orb_search_by_bow_kf.restype = C.c_int
    lib.oracle_orb_search_by_bow_kf.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, u8, u8, u8, u8, C.c_int, i32, i32, i32, C.c_int, i32, i32, i32,
                                                C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int, C.c_int, i32]
    d1, d2 = np.ascontiguousarray(desc1, dtype=np.uint8), np.ascontiguousarray(desc2, dtype=np.uint8)
    h1, h2 = np.ascontiguousarray(has_mp1, dtype=np.uint8), np.ascontiguousarray(has_mp2, dtype=np.uint8)
    f1 = [np.ascontiguousarray(a, dtype=np.int32) for a in fv1]
    f2 = [np.ascontiguousarray(a, dtype=np.int32) for a in fv2]
    a1, a2 = np.ascontiguousarray(angle1, dtype=np.float32), np.ascontiguousarray(angle2, dtype=np.float32)
    m = -np.ones(d1.shape[0], dtype=np.int32)
    n = lib.oracle_orb_search_by_bow_kf(d1.shape[0], d2.shape[0], int(lim1), int(lim2), _u8(d1), _u8(d2), _u8(h1), _u8(h2), len(f1[0]), _i32(f1[0]),
                                        _i32(f1[1]), _i32(f1[2]), len(f2[0]), _i32(f2[0]), _i32(f2[1]), _i32(f2[2]),
                                        a1.ctypes.data_as(C.POINTER(C.c_float)), a2.ctypes.data_as(C.POINTER(C.c_float)), C.c_float(nn_ratio),
                                        int(th_low), int(check_ori), _i32(m))
    return int(n), m


def orb_fuse(q_desc, feat_desc, skip, cand_off, cand_idx, stereo, slot, nobs, bad, th_low=50):
    """ORBmatcher::Fuse after its projection gates, restated sequentially (orb_oracle.c:oracle_orb_fuse).  skip: 0 candidate passes
    the gates, 1 fails one, 2 null entry.  slot[k] = id at feature k (-1 none); nobs / bad indexed [candidates..., residents...]
    (resident r has id 100000 + r).  Returns nFused and the final slot, nobs, bad, replaced arrays."""
    lib = load()
    i32, u8 = C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    lib.oracle_orb_fuse.restype = C.c_int
    lib.oracle_orb_fuse.argtypes = [C.c_int, C.c_int, C.c_int, u8, u8, u8, i32, i32, u8, C.c_int, i32, i32, u8, i32, u8]
    qd, fd = np.ascontiguousarray(q_desc, dtype=np.uint8), np.ascontiguousarray(feat_desc, dtype=np.uint8)
    sk, st = np.ascontiguousarray(skip, dtype=np.uint8), np.ascontiguousarray(stereo, dtype=np.uint8)
    off, idx = np.ascontiguousarray(cand_off, dtype=np.int32), np.ascontiguousarray(cand_idx if len(cand_idx) else [0], dtype=np.int32)
    slot = np.array(slot, dtype=np.int32)
    nobs = np.array(nobs, dtype=np.int32)
    bad = np.array(bad, dtype=np.uint8)
    n_q = qd.shape[0]
    n_res = len(nobs) - n_q
    replaced = -np.ones(len(nobs), dtype=np.int32)
    in_kf = np.zeros(len(nobs), dtype=np.uint8)
    for k in range(len(slot)):
        if slot[k] >= 100000:
            in_kf[n_q + slot[k] - 100000] = 1
    n = lib.oracle_orb_fuse(n_q, n_res, fd.shape[0], _u8(qd), _u8(fd), _u8(sk), _i32(off), _i32(idx), _u8(st), int(th_low), _i32(slot), _i32(nobs),
                            _u8(bad), _i32(replaced), _u8(in_kf))
    return int(n), slot, nobs, bad, replaced


def orb_fuse_sim3(q_desc, feat_desc, skip, cand_off, cand_idx, stereo, slot, slot_bad, nobs, th_low=50):
    """The Sim3 overload of ORBmatcher::Fuse after its gates (orb_oracle.c:oracle_orb_fuse_sim3).  Returns nFused, slot, nobs, replace."""
    lib = load()
    i32, u8 = C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    lib.oracle_orb_fuse_sim3.restype = C.c_int
    lib.oracle_orb_fuse_sim3.argtypes = [C.c_int, u8, u8, u8, i32, i32, u8, u8, C.c_int, i32, i32, i32]
    qd, fd = np.ascontiguousarray(q_desc, dtype=np.uint8), np.ascontiguousarray(feat_desc, dtype=np.uint8)
    sk, st, sb = (np.ascontiguousarray(a, dtype=np.uint8) for a in (skip, stereo, slot_bad))
    off, idx = np.ascontiguousarray(cand_off, dtype=np.int32), np.ascontiguousarray(cand_idx if len(cand_idx) else [0], dtype=np.int32)
    slot, nobs = np.array(slot, dtype=np.int32), np.array(nobs, dtype=np.int32)
    replace = -np.ones(qd.shape[0], dtype=np.int32)
    n = lib.oracle_orb_fuse_sim3(qd.shape[0], _u8(qd), _u8(fd), _u8(sk), _i32(off), _i32(idx), _u8(st), _u8(sb), int(th_low), _i32(slot), _i32(nobs),
                                 _i32(replace))
    return int(n), slot, nobs, replace


def orb_search_by_sim3(desc_mp1, desc_mp2, desc_kf1, desc_kf2, skip1, off1, idx1, skip2, off2, idx2, th_high=100):
    """ORBmatcher::SearchBySim3 after its projection gates (orb_oracle.c:oracle_orb_search_by_sim3): per keypoint slot of each keyframe the
    descriptor of the map point it holds and the candidate slots of the other keyframe.  Returns nFound and match12 (slot of keyframe 2 or -1)."""
    lib = load()
    i32, u8 = C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    lib.oracle_orb_search_by_sim3.restype = C.c_int
    lib.oracle_orb_search_by_sim3.argtypes = [C.c_int, C.c_int, u8, u8, u8, u8, u8, i32, i32, u8, i32, i32, C.c_int, i32]
    a = [np.ascontiguousarray(x, dtype=np.uint8) for x in (desc_mp1, desc_mp2, desc_kf1, desc_kf2)]
    s1, s2 = np.ascontiguousarray(skip1, dtype=np.uint8), np.ascontiguousarray(skip2, dtype=np.uint8)
    o1, o2 = np.ascontiguousarray(off1, dtype=np.int32), np.ascontiguousarray(off2, dtype=np.int32)
    x1 = np.ascontiguousarray(idx1 if len(idx1) else [0], dtype=np.int32)
    x2 = np.ascontiguousarray(idx2 if len(idx2) else [0], dtype=np.int32)
    n1, n2 = len(s1), len(s2)
    m12 = -np.ones(n1, dtype=np.int32)
    n = lib.oracle_orb_search_by_sim3(n1, n2, _u8(a[0]), _u8(a[1]), _u8(a[2]), _u8(a[3]), _u8(s1), _i32(o1), _i32(x1), _u8(s2), _i32(o2), _i32(x2), int(th_high),
                                      _i32(m12))
    return int(n), m12


def orb_search_for_triangulation(desc1, desc2, has_mp1, has_mp2, fv1, fv2, kp1, kp2, octave2, F12, ep, scale_factors, level_sigma2, only_stereo=False,
                                 coarse=False, th_low=50, check_ori=True):
    """ORBmatcher::SearchForTriangulation on two pinhole keyframes, restated (orb_oracle.c).  kp = x y angle uright per feature."""
    lib = load()
    i32, u8, f32p = C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_float)
    lib.oracle_orb_search_for_triangulation.restype = C.c_int
    lib.oracle_orb_search_for_triangulation.argtypes = [C.c_int, C.c_int, u8, u8, u8, u8, C.c_int, i32, i32, i32, C.c_int, i32, i32, i32, f32p, f32p, i32,
                                                        f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, i32]
    d1, d2 = np.ascontiguousarray(desc1, dtype=np.uint8), np.ascontiguousarray(desc2, dtype=np.uint8)
    h1, h2 = np.ascontiguousarray(has_mp1, dtype=np.uint8), np.ascontiguousarray(has_mp2, dtype=np.uint8)
    f1 = [np.ascontiguousarray(a, dtype=np.int32) for a in fv1]
    f2 = [np.ascontiguousarray(a, dtype=np.int32) for a in fv2]
    fl = [np.ascontiguousarray(a, dtype=np.float32) for a in (kp1, kp2, np.asarray(F12).reshape(9), ep, scale_factors, level_sigma2)]
    o2 = np.ascontiguousarray(octave2, dtype=np.int32)
    fp = lambda a: a.ctypes.data_as(f32p)
    m = -np.ones(d1.shape[0], dtype=np.int32)
    n = lib.oracle_orb_search_for_triangulation(d1.shape[0], d2.shape[0], _u8(d1), _u8(d2), _u8(h1), _u8(h2), len(f1[0]), _i32(f1[0]), _i32(f1[1]),
                                                _i32(f1[2]), len(f2[0]), _i32(f2[0]), _i32(f2[1]), _i32(f2[2]), fp(fl[0]), fp(fl[1]), _i32(o2), fp(fl[2]),
                                                fp(fl[3]), fp(fl[4]), fp(fl[5]), int(only_stereo), int(coarse), int(th_low), int(check_ori), _i32(m))
    return int(n), m


def orb_search_for_initialization(desc1, desc2, skip, cand_off, cand_idx, angle1, angle2, xy2, prev_xy, nn_ratio=0.9, th_low=50, check_ori=True):
    """ORBmatcher::SearchForInitialization after its candidate generation, restated sequentially (orb_oracle.c)."""
    lib = load()
    i32, u8, f32p = C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_float)
    lib.oracle_orb_search_for_initialization.restype = C.c_int
    lib.oracle_orb_search_for_initialization.argtypes = [C.c_int, C.c_int, u8, u8, u8, i32, i32, f32p, f32p, f32p, C.c_float, C.c_int, C.c_int, i32, f32p]
    d1, d2, sk = (np.ascontiguousarray(a, dtype=np.uint8) for a in (desc1, desc2, skip))
    off, idx = np.ascontiguousarray(cand_off, dtype=np.int32), np.ascontiguousarray(cand_idx if len(cand_idx) else [0], dtype=np.int32)
    a1, a2, x2 = (np.ascontiguousarray(a, dtype=np.float32) for a in (angle1, angle2, xy2))
    prev = np.array(prev_xy, dtype=np.float32)
    fp = lambda a: a.ctypes.data_as(f32p)
    m = -np.ones(d1.shape[0], dtype=np.int32)
    n = lib.oracle_orb_search_for_initialization(d1.shape[0], d2.shape[0], _u8(d1), _u8(d2), _u8(sk), _i32(off), _i32(idx), fp(a1), fp(a2), fp(x2),
                                                 C.c_float(nn_ratio), int(th_low), int(check_ori), _i32(m), fp(prev))
    return int(n), m, prev
