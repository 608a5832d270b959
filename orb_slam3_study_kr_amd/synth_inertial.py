"""Synthetic stereo-inertial local-BA windows (BASELINE.json configs[3], SURVEY.md 8d "Config 4").

An EuRoC-shaped window: N temporal keyframes 0.25 s apart (+1 fixed predecessor, + fixed covisible keyframes),
~2000 stereo landmarks, 200 Hz IMU with the EuRoC noise densities (Examples/Stereo-Inertial/EuRoC.yaml:74-78)
scaled as src/Tracking.cc:613-614 does, T_bc from EuRoC.yaml:64-71.  The IMU stream is preintegrated with the
reference's FLOAT32 recursion (IMU::Preintegrated::IntegrateNewMeasurement, src/ImuTypes.cc:177-237), restated
here in numpy float32; the 9x9 information follows EdgeInertial's constructor (src/G2oTypes.cc:492-511).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import capi, synth

f32 = np.float32
GRAVITY = float(f32(9.81))
# T_b_c1 of EuRoC cam0 (Examples/Stereo-Inertial/EuRoC.yaml:64-71)
T_BC = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
                 [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
                 [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949],
                 [0.0, 0.0, 0.0, 1.0]])
IMU_FREQ = 200.0
NG, NA, NGW, NAW = 1.7e-4, 2.0e-3, 1.9393e-5, 3.0e-3


def hat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.asarray(v).dtype)


def exp_so3(w):
    th = np.linalg.norm(w)
    W = hat(np.asarray(w, dtype=np.float64))
    if th < 1e-10:
        return np.eye(3) + W
    return np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th**2 * (W @ W)


def log_so3(R):
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
    return w if th < 1e-10 else th * w / np.sin(th)


def _normalize_f32(R):
    u, _, vt = np.linalg.svd(R.astype(f32))
    return (u @ vt).astype(f32)


def preintegrate(acc, gyr, dt, bias, nga_diag, walk_diag):
    """IMU::Preintegrated::IntegrateNewMeasurement over a measurement list, float32 throughout.
    bias = (bax bay baz bwx bwy bwz).  Returns the OSH_PREINT_FLOATS record and the 15x15 covariance C."""
    b = np.asarray(bias, dtype=f32)
    dR, dV, dP = np.eye(3, dtype=f32), np.zeros(3, f32), np.zeros(3, f32)
    JRg, JVg, JVa, JPg, JPa = (np.zeros((3, 3), f32) for _ in range(5))
    Cm = np.zeros((15, 15), f32)
    Nga, Walk = np.diag(np.asarray(nga_diag, f32)), np.diag(np.asarray(walk_diag, f32))
    dT = f32(0)
    dt = f32(dt)
    I3 = np.eye(3, dtype=f32)
    for a_m, w_m in zip(np.asarray(acc, f32), np.asarray(gyr, f32)):
        A, B = np.eye(9, dtype=f32), np.zeros((9, 6), f32)
        a = (a_m - b[:3]).astype(f32)
        dP = (dP + dV * dt + f32(0.5) * (dR @ a) * dt * dt).astype(f32)
        dV = (dV + (dR @ a) * dt).astype(f32)
        Wacc = hat(a).astype(f32)
        A[3:6, 0:3] = -dR * dt @ Wacc
        A[6:9, 0:3] = f32(-0.5) * dR * dt * dt @ Wacc
        A[6:9, 3:6] = I3 * dt
        B[3:6, 3:6] = dR * dt
        B[6:9, 3:6] = f32(0.5) * dR * dt * dt
        JPa = (JPa + JVa * dt - f32(0.5) * dR * dt * dt).astype(f32)
        JPg = (JPg + JVg * dt - f32(0.5) * dR * dt * dt @ Wacc @ JRg).astype(f32)
        JVa = (JVa - dR * dt).astype(f32)
        JVg = (JVg - dR * dt @ Wacc @ JRg).astype(f32)
        # IntegratedRotation (src/ImuTypes.cc:86-108)
        v = ((w_m - b[3:]) * dt).astype(f32)
        d2 = f32(v @ v)
        d = f32(np.sqrt(d2))
        W = hat(v).astype(f32)
        if d < f32(1e-4):
            deltaR, rightJ = (I3 + W).astype(f32), I3.copy()
        else:
            deltaR = (I3 + W * f32(np.sin(d)) / d + W @ W * (f32(1.0) - f32(np.cos(d))) / d2).astype(f32)
            rightJ = (I3 - W * (f32(1.0) - f32(np.cos(d))) / d2 + W @ W * (d - f32(np.sin(d))) / (d2 * d)).astype(f32)
        dR = _normalize_f32(dR @ deltaR)
        A[0:3, 0:3] = deltaR.T
        B[0:3, 0:3] = rightJ * dt
        Cm[0:9, 0:9] = (A @ Cm[0:9, 0:9] @ A.T + B @ Nga @ B.T).astype(f32)
        Cm[9:15, 9:15] += Walk
        JRg = (deltaR.T @ JRg - rightJ * dt).astype(f32)
        dT = f32(dT + dt)
    rec = np.zeros(capi.OSH_PREINT_FLOATS, f32)
    rec[0] = dT
    rec[1:10], rec[10:13], rec[13:16] = dR.ravel(), dV, dP
    rec[16:25], rec[25:34], rec[34:43], rec[43:52], rec[52:61] = JRg.ravel(), JVg.ravel(), JVa.ravel(), JPg.ravel(), JPa.ravel()
    rec[61:67] = b
    return rec, Cm


def inertial_information(Cm, downweight=False):
    """EdgeInertial information (src/G2oTypes.cc:500-508): inverse, symmetrise, zero eigenvalues < 1e-12."""
    info = np.linalg.inv(Cm[:9, :9].astype(np.float64))
    info = (info + info.T) / 2
    w, V = np.linalg.eigh(info)
    w[w < 1e-12] = 0
    info = V @ np.diag(w) @ V.T
    return info * 1e-2 if downweight else info


@dataclass
class LibaWindow:
    n_opt: int
    n_fixed_imu: int
    n_fixed: int
    pose_Rcw: np.ndarray
    pose_tcw: np.ndarray
    pose_Rwb: np.ndarray
    pose_twb: np.ndarray
    Rcb: np.ndarray
    tcb: np.ndarray
    tbc: np.ndarray
    cam: np.ndarray
    vel: np.ndarray
    bias_g: np.ndarray
    bias_a: np.ndarray
    points: np.ndarray
    edge_pose: np.ndarray
    edge_point: np.ndarray
    edge_kind: np.ndarray
    edge_obs: np.ndarray
    edge_info: np.ndarray
    link_prev: np.ndarray
    link_cur: np.ndarray
    link_preint: np.ndarray
    link_info: np.ndarray
    link_info_g: np.ndarray
    link_info_a: np.ndarray
    link_robust: np.ndarray
    huber_mono: float = synth.HUBER_MONO
    huber_stereo: float = synth.HUBER_STEREO
    huber_inertial: float = float(np.sqrt(16.92))
    lambda_init: float = 1.0
    max_iterations: int = 10
    kb8: np.ndarray | None = None   # [4] KannalaBrandt8 k1..k4 (monocular fisheye window)
    cam2: np.ndarray | None = None  # [8] right camera of a fisheye stereo rig: fx fy cx cy k1..k4 (edges of kind OSH_EDGE_RIGHT)
    trl: np.ndarray | None = None   # [12] rows of [Rrl | trl] (float32 values)
    link_bias: np.ndarray | None = None   # [n_links] keyframe that stores the bias vertices of each inertial edge (None: link_prev)
    gt: dict | None = None

    _F64 = ("pose_Rcw", "pose_tcw", "pose_Rwb", "pose_twb", "Rcb", "tcb", "tbc", "cam", "vel", "bias_g", "bias_a", "points",
            "edge_obs", "edge_info", "link_info", "link_info_g", "link_info_a")

    def normalise(self):
        for k in self._F64:
            setattr(self, k, np.ascontiguousarray(getattr(self, k), dtype=np.float64))
        for k in ("edge_pose", "edge_point", "link_prev", "link_cur"):
            setattr(self, k, np.ascontiguousarray(getattr(self, k), dtype=np.int32))
        for k in ("edge_kind", "link_robust"):
            setattr(self, k, np.ascontiguousarray(getattr(self, k), dtype=np.uint8))
        self.link_preint = np.ascontiguousarray(self.link_preint, dtype=np.float32)
        return self

    @property
    def n_points(self):
        return self.points.shape[0]

    @property
    def n_edges(self):
        return self.edge_pose.shape[0]

    @property
    def n_links(self):
        return self.link_prev.shape[0]

    def as_struct(self) -> capi.LibaProblem:
        self.normalise()
        p = capi.LibaProblem()
        p.n_opt, p.n_fixed_imu, p.n_fixed = self.n_opt, self.n_fixed_imu, self.n_fixed
        p.n_points, p.n_edges, p.n_links = self.n_points, self.n_edges, self.n_links
        for k in self._F64:
            setattr(p, k, capi.ptr(getattr(self, k), capi.c_double_p))
        p.edge_pose, p.edge_point = capi.ptr(self.edge_pose, capi.c_int32_p), capi.ptr(self.edge_point, capi.c_int32_p)
        p.edge_kind = capi.ptr(self.edge_kind, capi.c_uint8_p)
        p.link_prev, p.link_cur = capi.ptr(self.link_prev, capi.c_int32_p), capi.ptr(self.link_cur, capi.c_int32_p)
        p.link_preint = capi.ptr(self.link_preint, capi.c_float_p)
        p.link_robust = capi.ptr(self.link_robust, capi.c_uint8_p)
        p.huber_mono, p.huber_stereo, p.huber_inertial = self.huber_mono, self.huber_stereo, self.huber_inertial
        p.lambda_init, p.max_iterations = self.lambda_init, self.max_iterations
        if self.kb8 is not None:
            self.kb8 = np.ascontiguousarray(self.kb8, dtype=np.float64)
        p.kb8 = capi.ptr(self.kb8, capi.c_double_p)
        if self.cam2 is not None:
            self.cam2 = np.ascontiguousarray(self.cam2, dtype=np.float64)
            self.trl = np.ascontiguousarray(self.trl, dtype=np.float64)
        p.cam2, p.trl = capi.ptr(self.cam2, capi.c_double_p), capi.ptr(self.trl, capi.c_double_p)
        if self.link_bias is not None:
            self.link_bias = np.ascontiguousarray(self.link_bias, dtype=np.int32)
        p.link_bias = capi.ptr(self.link_bias, capi.c_int32_p)
        return p


class LibaResultArrays:
    def __init__(self, w: LibaWindow):
        N, L, E = w.n_opt, w.n_points, w.n_edges
        self.pose_Rcw, self.pose_tcw = np.zeros((N, 3, 3)), np.zeros((N, 3))
        self.pose_Rwb, self.pose_twb = np.zeros((N, 3, 3)), np.zeros((N, 3))
        self.vel, self.bias_g, self.bias_a = np.zeros((N, 3)), np.zeros((N, 3)), np.zeros((N, 3))
        self.points = np.zeros((L, 3))
        self.edge_chi2 = np.zeros(E)
        self.edge_depth_pos = np.zeros(E, dtype=np.uint8)
        self.struct = capi.LibaResult()
        self.bind(self.struct)

    def bind(self, r):
        for k in ("pose_Rcw", "pose_tcw", "pose_Rwb", "pose_twb", "vel", "bias_g", "bias_a", "points", "edge_chi2"):
            setattr(r, k, capi.ptr(getattr(self, k), capi.c_double_p))
        r.edge_depth_pos = capi.ptr(self.edge_depth_pos, capi.c_uint8_p)

    def read_scalars(self, r):
        n = r.n_trace
        self.status, self.iterations, self.trials = r.status, r.iterations, r.trials
        self.chi2_initial, self.chi2_final = r.chi2_initial, r.chi2_final
        self.chi2_trace = np.array(r.chi2_trace[:n])
        self.lambda_trace = np.array(r.lambda_trace[:n])
        self.trials_trace = np.array(r.trials_trace[:n])
        return self


def _trajectory(t):
    """Smooth body trajectory: world z up, camera (= body z) looking along world +x."""
    R0 = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    th = np.array([0.05 * np.sin(0.9 * t), 0.12 * np.sin(0.6 * t + 0.3), 0.04 * np.sin(1.1 * t)])
    R = R0 @ exp_so3(th)
    p = np.array([0.9 * t + 0.1 * np.sin(1.3 * t), 0.35 * np.sin(0.8 * t), 0.12 * np.sin(1.7 * t)])
    v = np.array([0.9 + 0.13 * np.cos(1.3 * t), 0.28 * np.cos(0.8 * t), 0.204 * np.cos(1.7 * t)])
    a = np.array([-0.169 * np.sin(1.3 * t), -0.224 * np.sin(0.8 * t), -0.3468 * np.sin(1.7 * t)])
    return R, p, v, a


def with_shared_bias(w: "LibaWindow", prior_g: float = 1e2, prior_a: float = 1e6) -> "LibaWindow":
    """The window in the form Optimizer::FullInertialBA takes with bInit (src/Optimizer.cc:452-462, 514-518, 551, 581-601): every
    EdgeInertial hangs on ONE (gyro, acc) bias pair -- stored with the oldest keyframe, which no link ends at --, there are no random-walk edges, and
    EdgePriorGyro / EdgePriorAcc (prior value 0, information prior_g I / prior_a I) hold that pair.  The priors are written as the
    random-walk edges of one extra link with zero inertial information, from the fixed predecessor (made to hold the prior value, and
    cut loose from the window) to the keyframe that stores the pair."""
    import dataclasses
    assert w.n_fixed_imu == 1
    N = w.n_opt
    keep = w.link_prev != N
    nl = int(keep.sum())
    rec = np.zeros(w.link_preint.shape[1], dtype=np.float32)
    rec[1:10] = np.eye(3, dtype=np.float32).ravel()              # dR = I, dT = 0: a finite residual that the zero information drops
    vel, bg, ba = (np.array(x, dtype=np.float64).reshape(-1, 3).copy() for x in (w.vel, w.bias_g, w.bias_a))
    vel[N] = 0.0; bg[N] = 0.0; ba[N] = 0.0                        # the prior value (bprior = 0, :586)
    cat = lambda a, b: np.concatenate([np.asarray(a)[keep], np.asarray(b)[None]], axis=0)
    out = dataclasses.replace(
        w, vel=vel, bias_g=bg, bias_a=ba,
        link_prev=cat(w.link_prev, np.int32(N)), link_cur=cat(w.link_cur, np.int32(0)), link_preint=cat(w.link_preint, rec),
        link_info=cat(np.asarray(w.link_info).reshape(len(keep), -1), np.zeros(81)),
        link_info_g=cat(np.zeros((len(keep), 9)), prior_g * np.eye(3).ravel()), link_info_a=cat(np.zeros((len(keep), 9)), prior_a * np.eye(3).ravel()),
        link_robust=cat(np.ones(len(keep), dtype=np.uint8), np.uint8(0)),
        link_bias=np.concatenate([np.full(nl, 0), [N]]).astype(np.int32))
    if out.gt is not None and "link_cov" in out.gt:
        out.gt = dict(out.gt)
        out.gt["link_cov"] = np.concatenate([np.asarray(w.gt["link_cov"])[keep], np.zeros((1, 15, 15))], axis=0)
    return out.normalise()


def make_inertial_window(seed: int = 11, n_opt: int = 10, n_fixed: int = 20, n_points: int = 2000, kf_dt: float = 0.25,
                         pixel_noise: bool = True, outlier_frac: float = 0.02, rec_init: bool = False, large: bool = False,
                         imu_noise: bool = True, fisheye: bool = False) -> LibaWindow:
    """``fisheye``: a monocular KannalaBrandt8 camera (synth.KB8_K): mono edges only, the window carries ``kb8``."""
    rng = np.random.Generator(np.random.PCG64(seed))
    per = int(round(kf_dt * IMU_FREQ))
    dt = 1.0 / IMU_FREQ
    sf = np.sqrt(IMU_FREQ)
    nga = np.array([(NG * sf) ** 2] * 3 + [(NA * sf) ** 2] * 3)
    walk = np.array([(NGW / sf) ** 2] * 3 + [(NAW / sf) ** 2] * 3)
    bg_true, ba_true = np.array([0.002, -0.0015, 0.003]), np.array([0.03, -0.02, 0.015])
    Rbc, tbc = T_BC[:3, :3], T_BC[:3, 3]
    Rcb, tcb = Rbc.T, -Rbc.T @ tbc
    # keyframe times: fixed observers (oldest), the fixed predecessor, then the temporal window
    K_imu = n_opt + 1
    times = np.concatenate([-(np.arange(n_fixed, 0, -1) + 0.0) * kf_dt - kf_dt, np.arange(K_imu) * kf_dt - kf_dt])
    # time index order: [fixed observers..., predecessor, opt_0 (oldest) ... opt_{N-1}]
    def cam_pose(t):
        Rwb, twb, v, _ = _trajectory(t)
        Rwc = Rwb @ Rbc
        twc = Rwb @ tbc + twb
        return Rwb, twb, v, Rwc.T, -Rwc.T @ twc
    gt = [cam_pose(t) for t in times]
    # landmarks in front of the path (world +x), 4-12 m ahead / around
    x0, x1 = _trajectory(times[0])[1][0], _trajectory(times[-1])[1][0]
    Xw = np.stack([rng.uniform(x0 + 3.0, x1 + 12.0, n_points), rng.uniform(-5.0, 5.0, n_points), rng.uniform(-2.5, 2.5, n_points)], axis=1)
    fx, fy, cx, cy, bf = (float(v) for v in (synth.FX, synth.FY, synth.CX, synth.CY, synth.BF))
    # pose index of the problem: opt (ascending time), predecessor, fixed observers
    T = len(times)
    pidx = np.zeros(T, dtype=np.int32)
    pidx[n_fixed + 1:] = np.arange(n_opt)
    pidx[n_fixed] = n_opt
    pidx[:n_fixed] = n_opt + 1 + np.arange(n_fixed)
    ep, el, eobs, einfo, eout = [], [], [], [], []
    for ti in range(T):
        Rcw, tcw = gt[ti][3], gt[ti][4]
        Xc = Xw @ Rcw.T + tcw
        z = Xc[:, 2]
        if fisheye:
            theta = np.arctan2(np.hypot(Xc[:, 0], Xc[:, 1]), z)
            psi = np.arctan2(Xc[:, 1], Xc[:, 0])
            kb = synth.KB8_K
            rr = theta + kb[0] * theta**3 + kb[1] * theta**5 + kb[2] * theta**7 + kb[3] * theta**9
            u = fx * rr * np.cos(psi) + cx
            v = fy * rr * np.sin(psi) + cy
            ur = np.full_like(u, -1.0)
        else:
            u = fx * Xc[:, 0] / z + cx
            v = fy * Xc[:, 1] / z + cy
            ur = u - bf / z
        vis = (z > 1.0) & (z < 14.0) & (u >= 0) & (u < synth.IMG_W) & (v >= 0) & (v < synth.IMG_H) & (fisheye | (ur >= 0))
        idx = np.nonzero(vis)[0]
        if ti < n_fixed:   # a fixed observer keeps a random third of what it sees
            idx = idx[rng.uniform(0, 1, idx.size) < 0.35]
        octave = rng.integers(0, synth.N_LEVELS, idx.size)
        sig = synth.SCALE_FACTORS[octave].astype(np.float64)
        noise = rng.standard_normal((idx.size, 3)) * sig[:, None] if pixel_noise else np.zeros((idx.size, 3))
        is_out = rng.uniform(0, 1, idx.size) < outlier_frac
        noise += rng.standard_normal((idx.size, 3)) * 20.0 * is_out[:, None]
        ep.append(np.full(idx.size, pidx[ti])); el.append(idx)
        eobs.append(np.stack([u[idx], v[idx], ur[idx]], axis=1) + noise)
        einfo.append(synth.INV_LEVEL_SIGMA2[octave].astype(np.float64)); eout.append(is_out)
    ep, el, eobs, einfo, eout = (np.concatenate(a) for a in (ep, el, eobs, einfo, eout))
    if fisheye:
        eobs[:, 2] = -1.0        # monocular keypoints: mvuRight < 0
    # keep landmarks seen by >= 2 keyframes of which >= 1 optimisable (local points come from the temporal window)
    cnt = np.bincount(el, minlength=n_points)
    cnt_opt = np.bincount(el[ep < n_opt], minlength=n_points)
    keep_l = (cnt >= 2) & (cnt_opt >= 1)
    sel = keep_l[el]
    remap = -np.ones(n_points, dtype=np.int64)
    remap[keep_l] = np.arange(keep_l.sum())
    ep, el, eobs, einfo, eout = ep[sel], remap[el[sel]], eobs[sel], einfo[sel], eout[sel]
    order = np.lexsort((ep, el))     # insertion order: landmark-major
    ep, el, eobs, einfo, eout = ep[order], el[order], eobs[order], einfo[order], eout[order]
    Xw = Xw[keep_l]
    # IMU stream + preintegration per link (predecessor -> opt_0, opt_0 -> opt_1, ...)
    recs, infos, infog, infoa, covs = [], [], [], [], []
    bias_lin = []
    for l in range(n_opt):
        t_a = times[n_fixed + l]
        acc, gyr = [], []
        for k in range(per):
            t = t_a + k * dt
            R0_, _, _, a0 = _trajectory(t)
            R1_ = _trajectory(t + dt)[0]
            w = log_so3(R0_.T @ R1_) / dt
            f = R0_.T @ (a0 - np.array([0, 0, -GRAVITY]))
            if imu_noise:
                w = w + rng.standard_normal(3) * NG * sf
                f = f + rng.standard_normal(3) * NA * sf
            gyr.append(w + bg_true); acc.append(f + ba_true)
        b_lin = np.concatenate([ba_true + rng.standard_normal(3) * 2e-3, bg_true + rng.standard_normal(3) * 2e-4])
        rec, Cm = preintegrate(acc, gyr, dt, b_lin, nga, walk)
        recs.append(rec); bias_lin.append(b_lin); covs.append(Cm)
        infos.append(inertial_information(Cm, downweight=(l == 0)))   # i == N-1 in the reference's newest-first order
        infog.append(np.linalg.inv(Cm[9:12, 9:12].astype(np.float64)))
        infoa.append(np.linalg.inv(Cm[12:15, 12:15].astype(np.float64)))
    # initial estimates: float32 keyframe storage; optimisable keyframes perturbed in the body frame
    K = n_opt + 1 + n_fixed
    Rcw_a, tcw_a, Rwb_a, twb_a = np.zeros((K, 3, 3)), np.zeros((K, 3)), np.zeros((K, 3, 3)), np.zeros((K, 3))
    vel, bg, ba = np.zeros((K_imu, 3)), np.zeros((K_imu, 3)), np.zeros((K_imu, 3))
    gt_Rwb, gt_twb, gt_vel = np.zeros((K, 3, 3)), np.zeros((K, 3)), np.zeros((K_imu, 3))
    for ti in range(T):
        k = pidx[ti]
        Rwb, twb, v = gt[ti][0], gt[ti][1], gt[ti][2]
        gt_Rwb[k], gt_twb[k] = Rwb, twb
        if k < n_opt:
            Rwb = Rwb @ exp_so3(rng.standard_normal(3) * 0.01)
            twb = twb + rng.standard_normal(3) * 0.03
        Rwc = Rwb @ Rbc
        twc = Rwb @ tbc + twb
        Rcw_a[k], tcw_a[k], Rwb_a[k], twb_a[k] = Rwc.T, -Rwc.T @ twc, Rwb, twb
        if k < K_imu:
            gt_vel[k] = v
            vel[k] = v + (rng.standard_normal(3) * 0.05 if k < n_opt else 0)
            bg[k] = bg_true + rng.standard_normal(3) * 1e-4
            ba[k] = ba_true + rng.standard_normal(3) * 1e-3
    q = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)   # noqa: E731  float storage of the map
    w = LibaWindow(
        n_opt=n_opt, n_fixed_imu=1, n_fixed=n_fixed, pose_Rcw=q(Rcw_a), pose_tcw=q(tcw_a), pose_Rwb=q(Rwb_a), pose_twb=q(twb_a),
        Rcb=q(Rcb), tcb=q(tcb), tbc=q(tbc), cam=q([fx, fy, cx, cy, bf]), vel=q(vel), bias_g=q(bg), bias_a=q(ba),
        points=q(Xw + rng.standard_normal(Xw.shape) * 0.05), edge_pose=ep, edge_point=el,
        edge_kind=np.full(ep.size, capi.OSH_EDGE_MONO if fisheye else capi.OSH_EDGE_STEREO, dtype=np.uint8), edge_obs=q(eobs), edge_info=einfo,
        kb8=synth.KB8_K.copy() if fisheye else None,
        link_prev=np.array([n_opt] + list(range(n_opt - 1)), dtype=np.int32), link_cur=np.arange(n_opt, dtype=np.int32),
        link_preint=np.stack(recs), link_info=np.stack(infos), link_info_g=np.stack(infog), link_info_a=np.stack(infoa),
        link_robust=np.array([1 if (l == 0 or rec_init) else 0 for l in range(n_opt)], dtype=np.uint8),
        lambda_init=1e-2 if large else 1.0, max_iterations=4 if large else 10,
        gt=dict(Rwb=gt_Rwb, twb=gt_twb, vel=gt_vel, points=Xw, outliers=eout, bg=bg_true, ba=ba_true, link_cov=np.stack(covs),
                Tbc=T_BC, nga=nga, walk=walk))
    return w.normalise()


def make_inertial_rig_window(seed: int = 11, right_frac: float = 0.5, right_only_frac: float = 0.1, **kwargs) -> LibaWindow:
    """A LocalInertialBA window of a fisheye STEREO rig (KeyFrame::mpCamera2 != NULL, src/Optimizer.cc:2798-2835): the monocular
    fisheye window of ``make_inertial_window(fisheye=True)`` whose observations gain, with probability ``right_frac``, an
    EdgeMono(1) of the same landmark in the same keyframe (OSH_EDGE_RIGHT) and are, with probability ``right_only_frac``, replaced
    by the right-camera observation alone.  The right edge's information follows the reference's expression
    ``mvInvLevelSigma2[kpUn.octave]`` with kpUn the LEFT keypoint declared in the same loop iteration (:2821): the left
    observation's level when there is one, level 0 (a default-constructed cv::KeyPoint) when there is none."""
    w = make_inertial_window(seed, fisheye=True, **kwargs)
    rng = np.random.Generator(np.random.PCG64(seed + 15485863))
    cam2 = np.array([float(synth.FX) * 1.01, float(synth.FY) * 0.99, float(synth.CX) + 3.0, float(synth.CY) - 2.0,
                     *(synth.KB8_K * np.array([1.05, 0.9, 1.1, 1.0]))]).astype(np.float32).astype(np.float64)
    rv = np.array([0.01, -0.02, 0.005])
    ang = np.linalg.norm(rv)
    q = np.concatenate([np.sin(ang / 2) * rv / ang, [np.cos(ang / 2)]])
    Rrl = synth._quat_to_R(q).astype(np.float32).astype(np.float64)     # Sophus::SE3f::matrix() holds float32 values
    trl_t = np.array([-0.1, 0.002, 0.001], dtype=np.float32).astype(np.float64)
    trl = np.concatenate([Rrl, trl_t[:, None]], axis=1).reshape(12)
    gt = w.gt
    Rbc, tbc = gt["Tbc"][:3, :3], gt["Tbc"][:3, 3]
    E = w.n_edges
    u = rng.uniform(0, 1, E)
    add_right = u < right_frac
    right_only = (u >= right_frac) & (u < right_frac + right_only_frac)
    ep, el, ek, eo, ei = [], [], [], [], []
    for e in range(E):
        ip, il = int(w.edge_pose[e]), int(w.edge_point[e])
        if add_right[e] or right_only[e]:
            Rwb, twb = gt["Rwb"][ip], gt["twb"][ip]
            Rwc = Rwb @ Rbc
            twc = Rwb @ tbc + twb
            Xl = Rwc.T @ (gt["points"][il] - twc)
            Xr = Rrl @ Xl + trl_t
            rho = np.hypot(Xr[0], Xr[1])
            theta = np.arctan2(rho, Xr[2])
            rr = theta + cam2[4] * theta**3 + cam2[5] * theta**5 + cam2[6] * theta**7 + cam2[7] * theta**9
            sigma = float(synth.SCALE_FACTORS[int(rng.integers(0, synth.N_LEVELS))])
            uv = np.array([cam2[0] * rr * Xr[0] / rho + cam2[2], cam2[1] * rr * Xr[1] / rho + cam2[3]])
            uv = uv + rng.standard_normal(2) * sigma + (rng.standard_normal(2) * 20.0 if rng.uniform() < 0.03 else 0.0)
        if not right_only[e]:
            ep.append(ip); el.append(il); ek.append(capi.OSH_EDGE_MONO); eo.append(w.edge_obs[e]); ei.append(w.edge_info[e])
        if add_right[e] or right_only[e]:
            ep.append(ip); el.append(il); ek.append(capi.OSH_EDGE_RIGHT); eo.append([uv[0], uv[1], -1.0])
            ei.append(float(w.edge_info[e]) if not right_only[e] else float(synth.INV_LEVEL_SIGMA2[0]))
    w.edge_pose, w.edge_point, w.edge_kind = np.array(ep, dtype=np.int32), np.array(el, dtype=np.int32), np.array(ek, dtype=np.uint8)
    w.edge_obs = np.asarray(eo, dtype=np.float32).astype(np.float64)
    w.edge_info = np.array(ei, dtype=np.float64)
    w.cam2, w.trl = cam2, trl
    w.gt["trl_qt"] = np.concatenate([q, trl_t]).astype(np.float32)      # the Sophus::SE3f a KeyFrame stores (mTrl)
    return w.normalise()


# --------------------------------------------------------------------------- PoseInertialOptimizationLastKeyFrame / LastFrame
@dataclass
class PoseiFrame:
    """Flat problem of Optimizer::PoseInertialOptimizationLastKeyFrame (mode 0) / ...LastFrame (mode 1), src/Optimizer.cc:4499-5299."""
    mode: int
    Rcw: np.ndarray
    tcw: np.ndarray
    Rwb: np.ndarray
    twb: np.ndarray
    vel: np.ndarray
    bias_g: np.ndarray
    bias_a: np.ndarray
    prev_Rwb: np.ndarray
    prev_twb: np.ndarray
    prev_vel: np.ndarray
    prev_bias_g: np.ndarray
    prev_bias_a: np.ndarray
    Rcb: np.ndarray
    tcb: np.ndarray
    tbc: np.ndarray
    cam: np.ndarray
    preint: np.ndarray
    info_inertial: np.ndarray
    info_g: np.ndarray
    info_a: np.ndarray
    points: np.ndarray
    edge_kind: np.ndarray
    edge_obs: np.ndarray
    edge_info: np.ndarray
    edge_close: np.ndarray
    prior_Rwb: np.ndarray | None = None
    prior_twb: np.ndarray | None = None
    prior_vel: np.ndarray | None = None
    prior_bg: np.ndarray | None = None
    prior_ba: np.ndarray | None = None
    prior_H: np.ndarray | None = None
    kb8: np.ndarray | None = None
    cam2: np.ndarray | None = None
    trl: np.ndarray | None = None
    rec_init: bool = False
    huber_mono: float = synth.HUBER_MONO
    huber_stereo: float = synth.HUBER_STEREO
    huber_prior: float = 5.0
    chi2_mono: tuple = (12.0, 7.5, 5.991, 5.991)
    chi2_stereo: tuple = (15.6, 9.8, 7.815, 7.815)
    iterations: tuple = (10, 10, 10, 10)
    gt: dict | None = None

    _F64 = ("Rcw", "tcw", "Rwb", "twb", "vel", "bias_g", "bias_a", "prev_Rwb", "prev_twb", "prev_vel", "prev_bias_g", "prev_bias_a", "Rcb", "tcb",
            "tbc", "cam", "info_inertial", "info_g", "info_a", "points", "edge_obs", "edge_info", "prior_Rwb", "prior_twb", "prior_vel",
            "prior_bg", "prior_ba", "prior_H", "kb8", "cam2", "trl")

    @property
    def n_edges(self):
        return int(self.edge_kind.shape[0])

    def normalise(self):
        for k in self._F64:
            if getattr(self, k) is not None:
                setattr(self, k, np.ascontiguousarray(getattr(self, k), dtype=np.float64))
        self.preint = np.ascontiguousarray(self.preint, dtype=np.float32)
        self.edge_kind = np.ascontiguousarray(self.edge_kind, dtype=np.uint8)
        self.edge_close = np.ascontiguousarray(self.edge_close, dtype=np.uint8)
        return self

    def as_struct(self) -> capi.PoseiProblem:
        self.normalise()
        p = capi.PoseiProblem()
        p.mode, p.n_edges, p.rec_init = int(self.mode), self.n_edges, int(self.rec_init)
        for k in self._F64:
            setattr(p, k, capi.ptr(getattr(self, k), capi.c_double_p))
        p.preint = capi.ptr(self.preint, capi.c_float_p)
        p.edge_kind, p.edge_close = capi.ptr(self.edge_kind, capi.c_uint8_p), capi.ptr(self.edge_close, capi.c_uint8_p)
        p.huber_mono, p.huber_stereo, p.huber_prior = self.huber_mono, self.huber_stereo, self.huber_prior
        for k in range(4):
            p.chi2_mono[k], p.chi2_stereo[k], p.iterations[k] = self.chi2_mono[k], self.chi2_stereo[k], self.iterations[k]
        return p


class PoseiResultArrays:
    def __init__(self, f: PoseiFrame):
        self.outlier = np.zeros(f.n_edges, dtype=np.uint8)
        self.edge_chi2 = np.zeros(f.n_edges, dtype=np.float64)
        self.struct = capi.PoseiResult()
        self.bind(self.struct)

    def bind(self, r):
        r.outlier = capi.ptr(self.outlier, capi.c_uint8_p)
        r.edge_chi2 = capi.ptr(self.edge_chi2, capi.c_double_p)

    def read(self, r, mode):
        for k, shape in (("Rcw", (3, 3)), ("tcw", (3,)), ("Rwb", (3, 3)), ("twb", (3,)), ("vel", (3,)), ("bias_g", (3,)), ("bias_a", (3,))):
            setattr(self, k, np.array(getattr(r, k)[:]).reshape(shape))
        self.n_bad, self.n_inliers, self.rounds, self.status = int(r.n_bad), int(r.n_inliers), int(r.rounds), int(r.status)
        n = 30 if mode == 1 else 15
        self.H = np.array(r.H[:n * n]).reshape(n, n)
        return self


def make_posei_frame(seed: int = 3, mode: int = 0, n_points: int = 400, stereo: bool = True, fisheye: bool = False, rig: bool = False,
                     outlier_frac: float = 0.05, rec_init: bool = False, link_dt: float | None = None) -> PoseiFrame:
    """A tracked frame with its IMU preintegration since the last keyframe (mode 0, 0.25 s) or since the previous frame (mode 1,
    0.05 s), matched to ``n_points`` map points: rectified-stereo observations, monocular ones (``stereo=False``), a monocular
    KannalaBrandt8 camera (``fisheye``) or a fisheye stereo rig whose right keypoints (index >= Nleft) are OSH_EDGE_RIGHT edges."""
    rng = np.random.Generator(np.random.PCG64(seed))
    fisheye = fisheye or rig
    stereo = stereo and not fisheye
    link_dt = link_dt if link_dt is not None else (0.25 if mode == 0 else 0.05)
    per = int(round(link_dt * IMU_FREQ))
    dt = 1.0 / IMU_FREQ
    sf = np.sqrt(IMU_FREQ)
    nga = np.array([(NG * sf) ** 2] * 3 + [(NA * sf) ** 2] * 3)
    walk = np.array([(NGW / sf) ** 2] * 3 + [(NAW / sf) ** 2] * 3)
    bg_true, ba_true = np.array([0.002, -0.0015, 0.003]), np.array([0.03, -0.02, 0.015])
    Rbc, tbc = T_BC[:3, :3], T_BC[:3, 3]
    Rcb, tcb = Rbc.T, -Rbc.T @ tbc
    t1 = 1.0 + 0.37 * seed
    t2 = t1 + per * dt
    acc, gyr = [], []
    for k in range(per):
        t = t1 + k * dt
        R0_, _, _, a0 = _trajectory(t)
        R1_ = _trajectory(t + dt)[0]
        w = log_so3(R0_.T @ R1_) / dt + rng.standard_normal(3) * NG * sf
        f = R0_.T @ (a0 - np.array([0, 0, -GRAVITY])) + rng.standard_normal(3) * NA * sf
        gyr.append(w + bg_true); acc.append(f + ba_true)
    b_lin = np.concatenate([ba_true + rng.standard_normal(3) * 2e-3, bg_true + rng.standard_normal(3) * 2e-4])
    rec, Cm = preintegrate(acc, gyr, dt, b_lin, nga, walk)
    q = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)   # noqa: E731  float storage of frames / keyframes
    R1, p1, v1, _ = _trajectory(t1)
    R2, p2, v2, _ = _trajectory(t2)
    # current frame: perturbed (the tracker's prediction), previous state: close to the truth
    Rwb = R2 @ exp_so3(rng.standard_normal(3) * 0.01)
    twb = p2 + rng.standard_normal(3) * 0.03
    Rwc, twc = Rwb @ Rbc, Rwb @ tbc + twb
    pRwb = R1 @ exp_so3(rng.standard_normal(3) * (1e-3 if mode == 1 else 0))
    ptwb = p1 + rng.standard_normal(3) * (3e-3 if mode == 1 else 0)
    # map points in front of the true camera
    Rwc_t, twc_t = R2 @ Rbc, R2 @ tbc + p2
    fx, fy, cx, cy, bf = (float(v) for v in (synth.FX, synth.FY, synth.CX, synth.CY, synth.BF))
    Xc = np.stack([rng.uniform(-4, 4, n_points), rng.uniform(-2.5, 2.5, n_points), rng.uniform(2.0, 25.0, n_points)], axis=1)
    Xw = Xc @ Rwc_t.T + twc_t
    kb = synth.KB8_K

    def project(cam, kbk, P):
        if kbk is None:
            return cam[0] * P[:, 0] / P[:, 2] + cam[2], cam[1] * P[:, 1] / P[:, 2] + cam[3]
        th = np.arctan2(np.hypot(P[:, 0], P[:, 1]), P[:, 2])
        ps = np.arctan2(P[:, 1], P[:, 0])
        rr = th + kbk[0] * th**3 + kbk[1] * th**5 + kbk[2] * th**7 + kbk[3] * th**9
        return cam[0] * rr * np.cos(ps) + cam[2], cam[1] * rr * np.sin(ps) + cam[3]
    cam = np.array([fx, fy, cx, cy, bf])
    u, v = project(cam, kb if fisheye else None, Xc)
    octave = rng.integers(0, synth.N_LEVELS, n_points)
    sig = synth.SCALE_FACTORS[octave].astype(np.float64)
    kind = np.full(n_points, capi.OSH_EDGE_STEREO if stereo else capi.OSH_EDGE_MONO, dtype=np.uint8)
    if stereo:
        kind[rng.uniform(size=n_points) < 0.2] = capi.OSH_EDGE_MONO          # mvuRight < 0: no stereo match
    obs = np.stack([u, v, u - bf / Xc[:, 2]], axis=1)
    cam2 = trl = None
    if rig:
        cam2 = np.array([fx * 1.01, fy * 0.99, cx + 3.0, cy - 2.0, *(kb * np.array([1.05, 0.9, 1.1, 1.0]))]).astype(np.float32).astype(np.float64)
        rv = np.array([0.01, -0.02, 0.005])
        ang = np.linalg.norm(rv)
        qq = np.concatenate([np.sin(ang / 2) * rv / ang, [np.cos(ang / 2)]])
        Rrl = synth._quat_to_R(qq).astype(np.float32).astype(np.float64)
        trl_t = np.array([-0.1, 0.002, 0.001], dtype=np.float32).astype(np.float64)
        trl = np.concatenate([Rrl, trl_t[:, None]], axis=1).reshape(12)
        trl_qt = np.concatenate([qq, trl_t]).astype(np.float32)
        right = rng.uniform(size=n_points) < 0.45                               # keypoints of the right image
        ur, vr = project(cam2, cam2[4:], Xc @ Rrl.T + trl_t)
        obs[right, 0], obs[right, 1] = ur[right], vr[right]
        kind[right] = capi.OSH_EDGE_RIGHT
    obs = obs + rng.standard_normal((n_points, 3)) * sig[:, None]
    is_out = rng.uniform(size=n_points) < outlier_frac
    obs += rng.standard_normal((n_points, 3)) * 25.0 * is_out[:, None]
    kind[(kind == capi.OSH_EDGE_STEREO) & (obs[:, 2] < 0)] = capi.OSH_EDGE_MONO     # a negative u_right reads as "no stereo match"
    obs[kind != capi.OSH_EDGE_STEREO, 2] = -1.0
    pts = q(Xw + rng.standard_normal(Xw.shape) * 0.02)
    f = PoseiFrame(
        mode=mode, Rcw=q(Rwc.T), tcw=q(-Rwc.T @ twc), Rwb=q(Rwb), twb=q(twb), vel=q(v2 + rng.standard_normal(3) * 0.05),
        bias_g=q(bg_true + rng.standard_normal(3) * 1e-4), bias_a=q(ba_true + rng.standard_normal(3) * 1e-3),
        prev_Rwb=q(pRwb), prev_twb=q(ptwb), prev_vel=q(v1 + rng.standard_normal(3) * (5e-3 if mode == 1 else 0)),
        prev_bias_g=q(bg_true + rng.standard_normal(3) * 1e-4), prev_bias_a=q(ba_true + rng.standard_normal(3) * 1e-3),
        Rcb=q(Rcb), tcb=q(tcb), tbc=q(tbc), cam=q(cam), preint=rec, info_inertial=inertial_information(Cm),
        info_g=np.linalg.inv(Cm[9:12, 9:12].astype(np.float64)), info_a=np.linalg.inv(Cm[12:15, 12:15].astype(np.float64)),
        points=pts, edge_kind=kind, edge_obs=q(obs), edge_info=synth.INV_LEVEL_SIGMA2[octave].astype(np.float64),
        edge_close=(Xc[:, 2] < 10.0).astype(np.uint8), kb8=kb.copy() if fisheye else None, cam2=cam2, trl=trl, rec_init=rec_init,
        chi2_mono=(12.0, 7.5, 5.991, 5.991) if mode == 0 else (5.991, 5.991, 5.991, 5.991),
        gt=dict(Rwb=R2, twb=p2, vel=v2, outliers=is_out))
    if rig:
        f.gt["trl_qt"] = trl_qt
    if mode == 1:
        # mpcpi of the previous frame: its state estimate at the time + a symmetric positive semi-definite information
        Q, _ = np.linalg.qr(rng.standard_normal((15, 15)))
        lam = 10.0 ** rng.uniform(2, 6, 15)
        f.prior_H = (Q * lam) @ Q.T
        f.prior_Rwb = q(R1 @ exp_so3(rng.standard_normal(3) * 1e-3))
        f.prior_twb = q(p1 + rng.standard_normal(3) * 3e-3)
        f.prior_vel = q(v1 + rng.standard_normal(3) * 5e-3)
        f.prior_bg = q(bg_true + rng.standard_normal(3) * 1e-4)
        f.prior_ba = q(ba_true + rng.standard_normal(3) * 1e-3)
    return f.normalise()
