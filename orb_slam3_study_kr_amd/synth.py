"""Deterministic synthetic inputs for the hot path.

No dataset ships with the reference (``.MISSING_LARGE_BLOBS``), so every
benchmark / parity input is generated here: local-BA windows shaped like the
graphs ``Optimizer::LocalBundleAdjustment`` builds (src/Optimizer.cc:1116-1402)
and ORB descriptor sets shaped like ``ORBmatcher::SearchByProjection`` inputs.

Everything the reference stores as ``float`` (poses as unit quaternion + t,
points, pixel observations, ``mvuRight``, ``mvInvLevelSigma2``, intrinsics,
``mbf``) is rounded to float32 *before* it is widened to double, exactly as the
graph construction does (src/Optimizer.cc:1217-1218,1286,1309,1316,1352-1356).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import capi

# EuRoC cam0 (Examples/Stereo-Inertial/EuRoC.yaml:23-26,43-44,58-60)
FX, FY, CX, CY = np.float32(458.654), np.float32(457.296), np.float32(367.215), np.float32(248.375)
IMG_W, IMG_H = 752, 480
BF = np.float32(np.float32(458.654) * np.float32(0.110078))
# KannalaBrandt8 k1..k4 (the float32 values an ORB-SLAM3 TUM-VI settings file stores as Camera1.k1..k4)
KB8_K = np.array([0.0034823894022493434, 0.0007150348452162257, -0.0020532361418706202, 0.00020293673591811182],
                 dtype=np.float32).astype(np.float64)
N_LEVELS, SCALE_FACTOR = 8, 1.2
# mvInvLevelSigma2[l] = 1/(1.2^l)^2 as float (src/ORBextractor.cc:414-429)
_sf = np.ones(N_LEVELS, dtype=np.float32)
for _i in range(1, N_LEVELS):
    _sf[_i] = np.float32(_sf[_i - 1] * np.float32(SCALE_FACTOR))
LEVEL_SIGMA2 = (_sf * _sf).astype(np.float32)
INV_LEVEL_SIGMA2 = (np.float32(1.0) / LEVEL_SIGMA2).astype(np.float32)
SCALE_FACTORS = _sf

# Huber deltas: `const float thHuberMono = sqrt(5.991)` (src/Optimizer.cc:1275-1276)
HUBER_MONO = float(np.float32(np.sqrt(5.991)))
HUBER_STEREO = float(np.float32(np.sqrt(7.815)))
CHI2_MONO, CHI2_STEREO = 5.991, 7.815


@dataclass
class LbaWindow:
    """One local-BA window as the flat arrays of ``osh_lba_problem``."""

    n_free: int
    n_fixed: int
    pose_qt: np.ndarray      # [(P+F),7] f64: qx qy qz qw tx ty tz (Tcw)
    pose_cam: np.ndarray     # [(P+F),5] f64
    points: np.ndarray       # [L,3] f64
    edge_pose: np.ndarray    # [E] i32
    edge_point: np.ndarray   # [E] i32
    edge_kind: np.ndarray    # [E] u8
    edge_obs: np.ndarray     # [E,3] f64
    edge_info: np.ndarray    # [E] f64
    huber_mono: float = HUBER_MONO
    huber_stereo: float = HUBER_STEREO
    lambda_init: float = 0.0
    max_iterations: int = 10
    stop_flag: np.ndarray | None = None   # u8[1] or None
    kb8: np.ndarray | None = None         # [4] f64 KannalaBrandt8 k1..k4: mono edges project through the fisheye model
    cam2: np.ndarray | None = None        # [8] f64 right camera of a fisheye stereo rig: fx fy cx cy k1..k4
    trl: np.ndarray | None = None         # [7] f64 Trl (left -> right camera): qx qy qz qw tx ty tz
    gt_pose_qt: np.ndarray | None = None  # ground truth (not part of the problem)
    gt_points: np.ndarray | None = None
    outlier_mask: np.ndarray | None = None
    _keep: list = field(default_factory=list, repr=False)

    @property
    def n_points(self) -> int:
        return int(self.points.shape[0])

    @property
    def n_edges(self) -> int:
        return int(self.edge_pose.shape[0])

    @property
    def n_free_edges(self) -> int:
        return int(np.count_nonzero(self.edge_pose < self.n_free))

    def normalise(self) -> "LbaWindow":
        self.pose_qt = np.ascontiguousarray(self.pose_qt, dtype=np.float64).reshape(-1, 7)
        self.pose_cam = np.ascontiguousarray(self.pose_cam, dtype=np.float64).reshape(-1, 5)
        self.points = np.ascontiguousarray(self.points, dtype=np.float64).reshape(-1, 3)
        self.edge_pose = np.ascontiguousarray(self.edge_pose, dtype=np.int32)
        self.edge_point = np.ascontiguousarray(self.edge_point, dtype=np.int32)
        self.edge_kind = np.ascontiguousarray(self.edge_kind, dtype=np.uint8)
        self.edge_obs = np.ascontiguousarray(self.edge_obs, dtype=np.float64).reshape(-1, 3)
        self.edge_info = np.ascontiguousarray(self.edge_info, dtype=np.float64)
        return self

    def as_struct(self) -> capi.LbaProblem:
        self.normalise()
        p = capi.LbaProblem()
        p.n_free, p.n_fixed = self.n_free, self.n_fixed
        p.n_points, p.n_edges = self.n_points, self.n_edges
        p.pose_qt = capi.ptr(self.pose_qt, capi.c_double_p)
        p.pose_cam = capi.ptr(self.pose_cam, capi.c_double_p)
        p.points = capi.ptr(self.points, capi.c_double_p)
        p.edge_pose = capi.ptr(self.edge_pose, capi.c_int32_p)
        p.edge_point = capi.ptr(self.edge_point, capi.c_int32_p)
        p.edge_kind = capi.ptr(self.edge_kind, capi.c_uint8_p)
        p.edge_obs = capi.ptr(self.edge_obs, capi.c_double_p)
        p.edge_info = capi.ptr(self.edge_info, capi.c_double_p)
        p.huber_mono, p.huber_stereo = self.huber_mono, self.huber_stereo
        p.lambda_init, p.max_iterations = self.lambda_init, self.max_iterations
        p.stop_flag = capi.ptr(self.stop_flag, capi.c_uint8_p) if self.stop_flag is not None else C.cast(None, capi.c_uint8_p)
        if self.kb8 is not None:
            self.kb8 = np.ascontiguousarray(self.kb8, dtype=np.float64)
        p.kb8 = capi.ptr(self.kb8, capi.c_double_p)
        if self.cam2 is not None:
            self.cam2 = np.ascontiguousarray(self.cam2, dtype=np.float64)
        if self.trl is not None:
            self.trl = np.ascontiguousarray(self.trl, dtype=np.float64)
        p.cam2 = capi.ptr(self.cam2, capi.c_double_p)
        p.trl = capi.ptr(self.trl, capi.c_double_p)
        return p

    def algorithmic_bytes(self) -> dict:
        """Per-pass algorithmic byte counts of SURVEY.md section 8(d) for this window."""
        P, F, L, E = self.n_free, self.n_fixed, self.n_points, self.n_edges
        Ef = self.n_free_edges
        d = np.where(self.edge_kind == capi.OSH_EDGE_MONO, 2, 3).astype(np.int64)
        resid = int((8 * d + 16).sum()) + L * 24 + (P + F) * 56
        lin = resid + Ef * 144 + L * 72 + P * 216
        schur = Ef * 144 + L * 72 + (6 * P) * (6 * P + 1) * 8
        back = Ef * 144 + L * 96 + 6 * P * 8
        update = 2 * (P * 56 + L * 24)
        return {"resid": resid, "lin": lin, "schur": schur, "back": back, "update": update}


class LbaResultArrays:
    """Owns the output arrays of one ``osh_lba_result``."""

    def __init__(self, w: LbaWindow):
        self.pose_qt = np.zeros((w.n_free, 7), dtype=np.float64)
        self.points = np.zeros((w.n_points, 3), dtype=np.float64)
        self.edge_chi2 = np.zeros(w.n_edges, dtype=np.float64)
        self.edge_depth_pos = np.zeros(w.n_edges, dtype=np.uint8)
        self.struct = capi.LbaResult()
        self.bind(self.struct)

    def bind(self, r: capi.LbaResult):
        r.pose_qt = capi.ptr(self.pose_qt, capi.c_double_p)
        r.points = capi.ptr(self.points, capi.c_double_p)
        r.edge_chi2 = capi.ptr(self.edge_chi2, capi.c_double_p)
        r.edge_depth_pos = capi.ptr(self.edge_depth_pos, capi.c_uint8_p)

    def read_scalars(self, r: capi.LbaResult):
        n = r.n_trace
        self.status, self.iterations, self.trials = r.status, r.iterations, r.trials
        self.chi2_initial = r.chi2_initial
        self.chi2_trace = np.array(r.chi2_trace[:n])
        self.lambda_trace = np.array(r.lambda_trace[:n])
        self.trials_trace = np.array(r.trials_trace[:n])
        return self


# --------------------------------------------------------------------------- geometry helpers
def _quat_from_R(R: np.ndarray) -> np.ndarray:
    """Rotation matrix -> unit quaternion (x,y,z,w), w>=0 (generator side only)."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[3] = (R[k, j] - R[j, k]) / s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def _quat_to_R(q: np.ndarray) -> np.ndarray:
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _rodrigues(w: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * (K @ K)


def quat_to_R(q: np.ndarray) -> np.ndarray:
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def _f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def make_window(seed: int, n_free: int = 50, n_fixed: int = 10, n_points: int = 10000, stereo: bool = True,
                track_len=(4, 12), pose_noise=(0.01, 0.05), point_noise: float = 0.1,
                outlier_frac: float = 0.03, pixel_noise: bool = True, kf_spacing: float = 0.25,
                yaw_drift: float = 0.02, lambda_init: float = 0.0, max_iterations: int = 10,
                mixed_mono_frac: float = 0.0, obs_dropout: float = 0.0, fisheye: bool = False) -> LbaWindow:
    """A synthetic local-BA window.

    Keyframes move along +x looking down +z with a slow yaw drift; the oldest
    ``n_fixed`` keyframes are the fixed observers, the rest are optimisable.
    Landmarks fill a 4-12 m deep slab; each is observed by a contiguous run of
    ``track_len`` keyframes that see it; ``obs_dropout`` removes that fraction of the
    observations at random (missed detections: the observer sets stop being contiguous runs).
    Order of poses: optimisable first.  ``fisheye``: a monocular KannalaBrandt8 camera (KB8_K, same fx fy cx cy):
    the observations are the fisheye projections and the window carries ``kb8``.
    """
    assert not (fisheye and stereo), "the fisheye window is monocular"
    rng = np.random.Generator(np.random.PCG64(seed))
    K = n_free + n_fixed
    # ---- ground-truth keyframe poses, Twc then Tcw
    Rcw = np.zeros((K, 3, 3))
    tcw = np.zeros((K, 3))
    centers = np.zeros((K, 3))
    for k in range(K):
        yaw = yaw_drift * (k - K / 2) * 0.25
        Rwc = _rodrigues(np.array([0.0, yaw, 0.0]))
        c = np.array([kf_spacing * k, 0.02 * np.sin(0.3 * k), 0.05 * np.cos(0.2 * k)])
        centers[k] = c
        Rcw[k] = Rwc.T
        tcw[k] = -Rwc.T @ c
    # time index -> pose index (optimisable first, then fixed = the oldest)
    pose_index = np.concatenate([np.arange(n_free, K), np.arange(0, n_free)]).astype(np.int32)
    # ---- landmarks
    span = kf_spacing * (K - 1)
    Xw = np.stack([
        rng.uniform(-3.0, span + 3.0, n_points),
        rng.uniform(-2.0, 2.0, n_points),
        rng.uniform(4.0, 12.0, n_points),
    ], axis=1)
    # project into all keyframes
    Xc = np.einsum("kij,lj->kli", Rcw, Xw) + tcw[:, None, :]          # [K,L,3]
    z = Xc[..., 2]
    if fisheye:
        theta = np.arctan2(np.hypot(Xc[..., 0], Xc[..., 1]), z)
        psi = np.arctan2(Xc[..., 1], Xc[..., 0])
        rr = theta + KB8_K[0] * theta**3 + KB8_K[1] * theta**5 + KB8_K[2] * theta**7 + KB8_K[3] * theta**9
        u = float(FX) * rr * np.cos(psi) + float(CX)
        v = float(FY) * rr * np.sin(psi) + float(CY)
    else:
        u = float(FX) * Xc[..., 0] / z + float(CX)
        v = float(FY) * Xc[..., 1] / z + float(CY)
    vis = (z > 0.5) & (u >= 0) & (u < IMG_W) & (v >= 0) & (v < IMG_H)
    if stereo:
        vis &= (u - float(BF) / z) >= 0
    any_vis = vis.any(axis=0)
    first = np.where(any_vis, vis.argmax(axis=0), 0)
    last = np.where(any_vis, K - 1 - vis[::-1].argmax(axis=0), -1)
    count = last - first + 1
    want = rng.integers(track_len[0], track_len[1] + 1, n_points)
    tlen = np.minimum(want, np.maximum(count, 0))
    start = first + np.floor(rng.uniform(0, 1, n_points) * (count - tlen + 1)).astype(np.int64)
    tt = np.arange(K)[:, None]
    obs_mask = vis & (tt >= start[None, :]) & (tt < (start + tlen)[None, :])
    if obs_dropout > 0:
        obs_mask &= np.random.Generator(np.random.PCG64(seed + 7919)).uniform(0, 1, obs_mask.shape) >= obs_dropout
    # every landmark needs >= 2 observations and >= 1 optimisable keyframe
    ok = (obs_mask.sum(axis=0) >= 2) & (obs_mask[n_fixed:].sum(axis=0) >= 1)
    obs_mask[:, ~ok] = False
    keep_l = np.nonzero(ok)[0]
    remap = -np.ones(n_points, dtype=np.int64)
    remap[keep_l] = np.arange(keep_l.size)
    # edges in the reference's insertion order: landmark-major, observers ascending
    ll, kk = np.nonzero(obs_mask.T)                                    # sorted by landmark, then time
    E = ll.size
    octave = rng.integers(0, N_LEVELS, E)
    sig = SCALE_FACTORS[octave].astype(np.float64)
    noise = rng.standard_normal((E, 3)) * sig[:, None] if pixel_noise else np.zeros((E, 3))
    is_out = rng.uniform(0, 1, E) < outlier_frac
    gross = rng.standard_normal((E, 3)) * 20.0 * is_out[:, None]
    uu, vv, zz = u[kk, ll], v[kk, ll], z[kk, ll]
    ur = uu - float(BF) / zz
    obs = np.stack([uu, vv, ur], axis=1) + noise + gross
    kind = np.full(E, capi.OSH_EDGE_STEREO if stereo else capi.OSH_EDGE_MONO, dtype=np.uint8)
    if stereo and mixed_mono_frac > 0:
        kind[rng.uniform(0, 1, E) < mixed_mono_frac] = capi.OSH_EDGE_MONO
    obs[kind == capi.OSH_EDGE_MONO, 2] = -1.0
    # ---- initial estimates = ground truth + perturbation, float32-quantised
    gt_qt = np.zeros((K, 7))
    init_qt = np.zeros((K, 7))
    for k in range(K):
        gt_qt[pose_index[k], :4] = _quat_from_R(Rcw[k])
        gt_qt[pose_index[k], 4:] = tcw[k]
        if pose_index[k] < n_free:
            dR = _rodrigues(rng.standard_normal(3) * pose_noise[0])
            Rn = dR @ Rcw[k]
            tn = tcw[k] + rng.standard_normal(3) * pose_noise[1]
        else:
            Rn, tn = Rcw[k], tcw[k]
        init_qt[pose_index[k], :4] = _quat_from_R(Rn)
        init_qt[pose_index[k], 4:] = tn
    init_qt = _f32(init_qt)  # Sophus::SE3f storage (KeyFrame::GetPose)
    pts_gt = Xw[keep_l]
    pts = _f32(pts_gt + rng.standard_normal(pts_gt.shape) * point_noise)
    cam = np.tile(np.array([FX, FY, CX, CY, BF], dtype=np.float32).astype(np.float64), (K, 1))
    w = LbaWindow(
        n_free=n_free, n_fixed=n_fixed, pose_qt=init_qt, pose_cam=cam, points=pts,
        edge_pose=pose_index[kk].astype(np.int32), edge_point=remap[ll].astype(np.int32), edge_kind=kind,
        edge_obs=_f32(obs), edge_info=INV_LEVEL_SIGMA2[octave].astype(np.float64),
        lambda_init=lambda_init, max_iterations=max_iterations,
        gt_pose_qt=gt_qt, gt_points=pts_gt, outlier_mask=is_out, kb8=KB8_K.copy() if fisheye else None,
    )
    return w.normalise()


def make_rig_window(seed: int, right_frac: float = 0.6, right_only_frac: float = 0.1, **kwargs) -> LbaWindow:
    """A fisheye STEREO rig window (KannalaBrandt8 left + right camera, src/Optimizer.cc:1305-1399): the monocular fisheye window
    of ``make_window(fisheye=True)`` whose observations gain, with probability ``right_frac``, a right-camera observation of the
    same landmark in the same keyframe (OSH_EDGE_BODY on the same Hessian block) and are, with probability ``right_only_frac``,
    replaced by a right-camera observation alone.  Right camera: slightly different intrinsics, Trl = a 10 cm baseline with a
    small rotation.  Edge order as the reference inserts them: left edge, then right edge, landmark by landmark."""
    w = make_window(seed, stereo=False, fisheye=True, **kwargs)
    rng = np.random.Generator(np.random.PCG64(seed + 104729))
    cam2 = np.array([float(FX) * 1.01, float(FY) * 0.99, float(CX) + 3.0, float(CY) - 2.0, *(KB8_K * np.array([1.05, 0.9, 1.1, 1.0]))])
    cam2[:4] = cam2[:4].astype(np.float32).astype(np.float64)
    cam2[4:] = cam2[4:].astype(np.float32).astype(np.float64)
    rv = np.array([0.01, -0.02, 0.005])
    ang = np.linalg.norm(rv)
    q = np.concatenate([np.sin(ang / 2) * rv / ang, [np.cos(ang / 2)]])
    trl = np.concatenate([q, [-0.1, 0.002, 0.001]]).astype(np.float32).astype(np.float64)   # Sophus::SE3f storage
    Rrl = _quat_to_R(trl[:4] / np.linalg.norm(trl[:4]))
    E = w.n_edges
    u = rng.uniform(0, 1, E)
    add_right = u < right_frac
    right_only = (u >= right_frac) & (u < right_frac + right_only_frac)
    ep, el, ek, eo, ei = [], [], [], [], []
    for e in range(E):
        ip, il = int(w.edge_pose[e]), int(w.edge_point[e])
        if add_right[e] or right_only[e]:
            qt = w.gt_pose_qt[ip]
            Xl = _quat_to_R(qt[:4]) @ w.gt_points[il] + qt[4:]
            Xr = Rrl @ Xl + trl[4:]
            rho = np.hypot(Xr[0], Xr[1])
            theta = np.arctan2(rho, Xr[2])
            rr = theta + cam2[4] * theta**3 + cam2[5] * theta**5 + cam2[6] * theta**7 + cam2[7] * theta**9
            octave = int(rng.integers(0, N_LEVELS))
            uv = np.array([cam2[0] * rr * Xr[0] / rho + cam2[2], cam2[1] * rr * Xr[1] / rho + cam2[3]])
            uv = uv + rng.standard_normal(2) * float(SCALE_FACTORS[octave]) + (rng.standard_normal(2) * 20.0 if rng.uniform() < 0.03 else 0.0)
        if not right_only[e]:
            ep.append(ip); el.append(il); ek.append(capi.OSH_EDGE_MONO); eo.append(w.edge_obs[e]); ei.append(w.edge_info[e])
        if add_right[e] or right_only[e]:
            ep.append(ip); el.append(il); ek.append(capi.OSH_EDGE_BODY); eo.append([uv[0], uv[1], -1.0]); ei.append(float(INV_LEVEL_SIGMA2[octave]))
    return LbaWindow(n_free=w.n_free, n_fixed=w.n_fixed, pose_qt=w.pose_qt, pose_cam=w.pose_cam, points=w.points,
                     edge_pose=np.array(ep, dtype=np.int32), edge_point=np.array(el, dtype=np.int32), edge_kind=np.array(ek, dtype=np.uint8),
                     edge_obs=_f32(np.array(eo, dtype=np.float64)), edge_info=np.array(ei, dtype=np.float64),
                     lambda_init=w.lambda_init, max_iterations=w.max_iterations, gt_pose_qt=w.gt_pose_qt, gt_points=w.gt_points,
                     kb8=w.kb8, cam2=cam2, trl=trl).normalise()


def make_config1(seed: int = 1) -> LbaWindow:
    """BASELINE.json configs[0]: 10 KF / ~1k landmark monocular plumbing graph."""
    return make_window(seed, n_free=7, n_fixed=3, n_points=1000, stereo=False, track_len=(2, 6))


def make_config2(seed: int = 100) -> LbaWindow:
    """BASELINE.json configs[1]: 50 free + 10 fixed KF, 10k landmarks, stereo, <=10 LM iterations."""
    return make_window(seed, n_free=50, n_fixed=10, n_points=10000, stereo=True, track_len=(4, 12))


# --------------------------------------------------------------------------- ORB descriptors (config 3)
@dataclass
class OrbPair:
    query_desc: np.ndarray     # [Nq,32] u8
    train_desc: np.ndarray     # [Nt,32] u8
    train_level: np.ndarray    # [Nt] i32
    cand_off: np.ndarray | None = None   # [Nq+1] i32
    cand_idx: np.ndarray | None = None   # [sum] i32
    query_angle: np.ndarray | None = None
    train_angle: np.ndarray | None = None


def make_orb_pair(seed: int = 7, n_query: int = 2000, n_train: int = 2000, match_frac: float = 0.7,
                  flip_prob: float = 0.08, windowed: bool = False, same_level: bool = True) -> OrbPair:
    """70 % of the queries are noisy copies of a train descriptor."""
    rng = np.random.Generator(np.random.PCG64(seed))
    train = rng.integers(0, 256, (n_train, 32), dtype=np.uint8)
    query = rng.integers(0, 256, (n_query, 32), dtype=np.uint8)
    perm = rng.permutation(max(n_train, n_query))[:n_query] % n_train
    is_copy = rng.uniform(0, 1, n_query) < match_frac
    flips = np.packbits(rng.uniform(0, 1, (n_query, 256)) < flip_prob, axis=1)
    query[is_copy] = train[perm[is_copy]] ^ flips[is_copy]
    level = np.zeros(n_train, dtype=np.int32) if same_level else rng.integers(0, N_LEVELS, n_train).astype(np.int32)
    pair = OrbPair(np.ascontiguousarray(query), np.ascontiguousarray(train), level)
    pair.query_angle = rng.uniform(0, 360, n_query).astype(np.float32)
    pair.train_angle = rng.uniform(0, 360, n_train).astype(np.float32)
    if windowed:
        # keypoints on a 752x480 image, candidate lists in Frame::GetFeaturesInArea order
        # (src/Frame.cc:658-722): grid cells ix-major then iy, insertion order inside a cell.
        tx = rng.uniform(0, IMG_W, n_train).astype(np.float32)
        ty = rng.uniform(0, IMG_H, n_train).astype(np.float32)
        qx = np.where(is_copy, tx[perm] + rng.normal(0, 3, n_query), rng.uniform(0, IMG_W, n_query)).astype(np.float32)
        qy = np.where(is_copy, ty[perm] + rng.normal(0, 3, n_query), rng.uniform(0, IMG_H, n_query)).astype(np.float32)
        qlevel = level[perm] if not same_level else np.zeros(n_query, dtype=np.int32)
        off, idx = features_in_area_lists(tx, ty, level, qx, qy, np.float32(15.0) * SCALE_FACTORS[qlevel], None, None)
        pair.cand_off, pair.cand_idx = off, idx
    return pair


FRAME_GRID_COLS, FRAME_GRID_ROWS = 64, 48  # include/Frame.h


def features_in_area_lists(tx, ty, tlevel, qx, qy, r, min_level, max_level):
    """Candidate lists in the order Frame::GetFeaturesInArea produces them
    (src/Frame.cc:658-722; float32 arithmetic, square window, strict '<')."""
    n_train = len(tx)
    winv = np.float32(FRAME_GRID_COLS) / np.float32(IMG_W)
    hinv = np.float32(FRAME_GRID_ROWS) / np.float32(IMG_H)
    grid = [[[] for _ in range(FRAME_GRID_ROWS)] for _ in range(FRAME_GRID_COLS)]
    for i in range(n_train):  # Frame::AssignFeaturesToGrid / PosInGrid (round)
        gx = int(np.round(np.float32(tx[i]) * winv))
        gy = int(np.round(np.float32(ty[i]) * hinv))
        if 0 <= gx < FRAME_GRID_COLS and 0 <= gy < FRAME_GRID_ROWS:
            grid[gx][gy].append(i)
    off = [0]
    idx = []
    for q in range(len(qx)):
        x, y, rr = np.float32(qx[q]), np.float32(qy[q]), np.float32(r[q])
        c0 = max(0, int(np.floor((x - rr) * winv)))
        c1 = min(FRAME_GRID_COLS - 1, int(np.ceil((x + rr) * winv)))
        r0 = max(0, int(np.floor((y - rr) * hinv)))
        r1 = min(FRAME_GRID_ROWS - 1, int(np.ceil((y + rr) * hinv)))
        if c0 < FRAME_GRID_COLS and c1 >= 0 and r0 < FRAME_GRID_ROWS and r1 >= 0:
            for ix in range(c0, c1 + 1):
                for iy in range(r0, r1 + 1):
                    for j in grid[ix][iy]:
                        if min_level is not None and tlevel[j] < min_level[q]:
                            continue
                        if max_level is not None and max_level[q] >= 0 and tlevel[j] > max_level[q]:
                            continue
                        if abs(np.float32(tx[j]) - x) < rr and abs(np.float32(ty[j]) - y) < rr:
                            idx.append(j)
        off.append(len(idx))
    return np.asarray(off, dtype=np.int32), np.asarray(idx, dtype=np.int32)


# --------------------------------------------------------------------------- pose-only optimisation of a frame
@dataclass
class PoseFrame:
    """Flat problem of Optimizer::PoseOptimization (src/Optimizer.cc:815-1114): one pose, unary edges."""
    pose_qt: np.ndarray       # [7] initial Tcw
    cam: np.ndarray           # [5]
    points: np.ndarray        # [E,3]
    edge_kind: np.ndarray     # [E] u8
    edge_obs: np.ndarray      # [E,3]
    edge_info: np.ndarray     # [E]
    gt_pose_qt: np.ndarray | None = None
    outlier_mask: np.ndarray | None = None
    huber_mono: float = HUBER_MONO
    huber_stereo: float = HUBER_STEREO
    chi2_mono: tuple = (5.991, 5.991, 5.991, 5.991)
    chi2_stereo: tuple = (7.815, 7.815, 7.815, 7.815)
    iterations: tuple = (10, 10, 10, 10)
    kb8: np.ndarray | None = None   # [4] KannalaBrandt8 k1..k4: mono edges project through the fisheye model
    cam2: np.ndarray | None = None  # [8] right camera of a fisheye stereo frame (OSH_EDGE_BODY edges: EdgeSE3ProjectXYZOnlyPoseToBody)
    trl: np.ndarray | None = None   # [7] Trl qx qy qz qw tx ty tz

    @property
    def n_edges(self) -> int:
        return int(self.edge_kind.shape[0])

    def normalise(self):
        self.pose_qt = np.ascontiguousarray(self.pose_qt, dtype=np.float64)
        self.cam = np.ascontiguousarray(self.cam, dtype=np.float64)
        self.points = np.ascontiguousarray(self.points, dtype=np.float64).reshape(-1, 3)
        self.edge_kind = np.ascontiguousarray(self.edge_kind, dtype=np.uint8)
        self.edge_obs = np.ascontiguousarray(self.edge_obs, dtype=np.float64).reshape(-1, 3)
        self.edge_info = np.ascontiguousarray(self.edge_info, dtype=np.float64)
        return self

    def as_struct(self) -> "capi.PoseProblem":
        p = capi.PoseProblem()
        p.n_edges = self.n_edges
        p.pose_qt = capi.ptr(self.pose_qt, capi.c_double_p)
        p.cam = capi.ptr(self.cam, capi.c_double_p)
        p.points = capi.ptr(self.points, capi.c_double_p)
        p.edge_kind = capi.ptr(self.edge_kind, capi.c_uint8_p)
        p.edge_obs = capi.ptr(self.edge_obs, capi.c_double_p)
        p.edge_info = capi.ptr(self.edge_info, capi.c_double_p)
        p.huber_mono, p.huber_stereo = self.huber_mono, self.huber_stereo
        for k in range(4):
            p.chi2_mono[k] = self.chi2_mono[k]
            p.chi2_stereo[k] = self.chi2_stereo[k]
            p.iterations[k] = self.iterations[k]
        if self.kb8 is not None:
            self.kb8 = np.ascontiguousarray(self.kb8, dtype=np.float64)
        p.kb8 = capi.ptr(self.kb8, capi.c_double_p)
        if self.cam2 is not None:
            self.cam2 = np.ascontiguousarray(self.cam2, dtype=np.float64)
            self.trl = np.ascontiguousarray(self.trl, dtype=np.float64)
        p.cam2, p.trl = capi.ptr(self.cam2, capi.c_double_p), capi.ptr(self.trl, capi.c_double_p)
        return p


class PoseResultArrays:
    def __init__(self, f: PoseFrame):
        self.outlier = np.zeros(f.n_edges, dtype=np.uint8)
        self.edge_chi2 = np.zeros(f.n_edges, dtype=np.float64)
        self.struct = capi.PoseResult()
        self.bind(self.struct)

    def bind(self, r):
        r.outlier = capi.ptr(self.outlier, capi.c_uint8_p)
        r.edge_chi2 = capi.ptr(self.edge_chi2, capi.c_double_p)

    def read_scalars(self, r):
        self.pose_qt = np.array(r.pose_qt[:7])
        self.n_bad, self.rounds, self.status = int(r.n_bad), int(r.rounds), int(r.status)
        self.iterations = np.array(r.iterations[:4])
        self.chi2_final = np.array(r.chi2_final[:4])
        return self


def make_pose_frame(seed: int = 21, n_points: int = 800, stereo: bool = True, mixed_mono_frac: float = 0.3, outlier_frac: float = 0.1,
                    pose_noise=(0.01, 0.05), fisheye: bool = False, rig: bool = False) -> PoseFrame:
    """A tracked frame: map points in front of the camera, pixel noise by pyramid level, gross outliers (wrong matches),
    the initial pose = ground truth + perturbation, everything the reference stores as float rounded to float32.
    ``rig``: a fisheye stereo frame (Nleft != -1): about 45 % of the matched keypoints belong to the right camera (OSH_EDGE_BODY)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    if rig:
        fisheye, stereo = True, False
    yaw = 0.1
    Rcw = _rodrigues(np.array([0.02, yaw, -0.01]))
    tcw = np.array([0.3, -0.1, 0.2])
    Xc = np.stack([rng.uniform(-4, 4, n_points), rng.uniform(-2.5, 2.5, n_points), rng.uniform(3, 14, n_points)], axis=1)
    if fisheye:      # monocular KannalaBrandt8 camera (KB8_K), same fx fy cx cy
        assert not stereo, "the fisheye frame is monocular"
        theta = np.arctan2(np.hypot(Xc[:, 0], Xc[:, 1]), Xc[:, 2])
        psi = np.arctan2(Xc[:, 1], Xc[:, 0])
        rr = theta + KB8_K[0] * theta**3 + KB8_K[1] * theta**5 + KB8_K[2] * theta**7 + KB8_K[3] * theta**9
        u = float(FX) * rr * np.cos(psi) + float(CX)
        v = float(FY) * rr * np.sin(psi) + float(CY)
    else:
        u = float(FX) * Xc[:, 0] / Xc[:, 2] + float(CX)
        v = float(FY) * Xc[:, 1] / Xc[:, 2] + float(CY)
    vis = (u >= 0) & (u < IMG_W) & (v >= 0) & (v < IMG_H) & (fisheye | ((u - float(BF) / Xc[:, 2]) >= 0))
    Xc, u, v = Xc[vis], u[vis], v[vis]
    E = Xc.shape[0]
    Xw = (Xc - tcw) @ Rcw                              # Rcw^T (Xc - tcw)
    octave = rng.integers(0, N_LEVELS, E)
    sig = SCALE_FACTORS[octave].astype(np.float64)
    is_out = rng.uniform(0, 1, E) < outlier_frac
    noise = rng.standard_normal((E, 3)) * sig[:, None] + rng.standard_normal((E, 3)) * 25.0 * is_out[:, None]
    obs = np.stack([u, v, u - float(BF) / Xc[:, 2]], axis=1) + noise
    kind = np.full(E, capi.OSH_EDGE_STEREO if stereo else capi.OSH_EDGE_MONO, dtype=np.uint8)
    if stereo and mixed_mono_frac > 0:
        kind[rng.uniform(0, 1, E) < mixed_mono_frac] = capi.OSH_EDGE_MONO
    obs[kind == capi.OSH_EDGE_MONO, 2] = -1.0
    gt = np.concatenate([_quat_from_R(Rcw), tcw])
    dR = _rodrigues(rng.standard_normal(3) * pose_noise[0])
    init = np.concatenate([_quat_from_R(dR @ Rcw), tcw + rng.standard_normal(3) * pose_noise[1]])
    cam = np.array([FX, FY, CX, CY, BF], dtype=np.float32).astype(np.float64)
    cam2 = trl = None
    if rig:
        cam2 = np.array([float(FX) * 1.01, float(FY) * 0.99, float(CX) + 3.0, float(CY) - 2.0, *(KB8_K * np.array([1.05, 0.9, 1.1, 1.0]))]).astype(np.float32).astype(np.float64)
        rv = np.array([0.01, -0.02, 0.005])
        ang = np.linalg.norm(rv)
        trl = np.concatenate([np.sin(ang / 2) * rv / ang, [np.cos(ang / 2)], [-0.1, 0.002, 0.001]]).astype(np.float32).astype(np.float64)
        Xr = Xc @ _quat_to_R(trl[:4] / np.linalg.norm(trl[:4])).T + trl[4:]
        th = np.arctan2(np.hypot(Xr[:, 0], Xr[:, 1]), Xr[:, 2])
        ps = np.arctan2(Xr[:, 1], Xr[:, 0])
        rr = th + cam2[4] * th**3 + cam2[5] * th**5 + cam2[6] * th**7 + cam2[7] * th**9
        right = rng.uniform(0, 1, E) < 0.45
        obs[right, 0] = (cam2[0] * rr * np.cos(ps) + cam2[2] + noise[:, 0])[right]
        obs[right, 1] = (cam2[1] * rr * np.sin(ps) + cam2[3] + noise[:, 1])[right]
        kind[right] = capi.OSH_EDGE_BODY
    return PoseFrame(pose_qt=_f32(init), cam=cam, points=_f32(Xw), edge_kind=kind, edge_obs=_f32(obs),
                     edge_info=INV_LEVEL_SIGMA2[octave].astype(np.float64), gt_pose_qt=gt, outlier_mask=is_out,
                     kb8=KB8_K.copy() if fisheye else None, cam2=cam2, trl=trl).normalise()


def make_bow_pair(seed: int = 5, n_kf: int = 900, n_f: int = 1000, n_nodes: int = 120, match_frac: float = 0.6, flip: float = 0.04, n_left_f: int = -1):
    """A keyframe and a frame described by DBoW2-style feature vectors (vocabulary node -> feature indices, ids ascending) for
    ORBmatcher::SearchByBoW: ``match_frac`` of the frame features are noisy copies of keyframe features in the SAME node (so they
    can match), several keyframe features share near-identical descriptors (ratio-test failures and contested frame features),
    some nodes exist on one side only.  Returns a dict of arrays; feature vectors as (node_id, node_off, node_feat)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    kf_desc = rng.integers(0, 256, (n_kf, 32), dtype=np.uint8)
    dup = rng.permutation(n_kf)[:n_kf // 6]                        # near-duplicates: a source of ties and contested slots
    kf_desc[dup] = kf_desc[(dup + 1) % n_kf] ^ np.packbits(rng.uniform(0, 1, (len(dup), 256)) < 0.01, axis=1)
    kf_node = rng.integers(0, n_nodes, n_kf)
    kf_node[dup] = kf_node[(dup + 1) % n_kf]
    f_desc = rng.integers(0, 256, (n_f, 32), dtype=np.uint8)
    f_node = rng.integers(0, n_nodes + 15, n_f)                     # nodes n_nodes .. n_nodes+14 exist in the frame only
    src = rng.permutation(n_kf)[:int(match_frac * n_f)]
    tgt = rng.permutation(n_f)[:len(src)]
    f_desc[tgt] = kf_desc[src] ^ np.packbits(rng.uniform(0, 1, (len(src), 256)) < flip, axis=1)
    f_node[tgt] = kf_node[src]
    kf_node[kf_node % 11 == 3] += 1000                              # nodes that exist in the keyframe only
    kf_angle = rng.uniform(0, 360, n_kf).astype(np.float32)
    f_angle = rng.uniform(0, 360, n_f).astype(np.float32)
    f_angle[tgt] = (kf_angle[src] - 20.0 + rng.normal(0, 3.0, len(src))).astype(np.float32) % np.float32(360.0)   # a dominant rotation

    def fv(node):
        ids = np.unique(node)
        feats = [np.nonzero(node == i)[0] for i in ids]
        off = np.concatenate([[0], np.cumsum([len(f) for f in feats])])
        return ids.astype(np.int32), off.astype(np.int32), np.concatenate(feats).astype(np.int32)
    return dict(kf_desc=kf_desc, f_desc=f_desc, kf_has_mp=(rng.uniform(size=n_kf) < 0.8).astype(np.uint8), kf_fv=fv(kf_node), f_fv=fv(f_node),
                kf_angle=kf_angle, f_angle=f_angle, n_left_f=n_left_f, src=src, tgt=tgt)
