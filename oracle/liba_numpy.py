"""Independent numpy check of the stereo-inertial local-BA maths (TEST INFRASTRUCTURE ONLY).

Pins oracle/liba_oracle.c without the (unbuildable) reference: residuals are re-derived with numpy (rotation
matrices, numpy float32 for the preintegration getters, real SVD for NormalizeRotation) and EVERY Jacobian is
numeric (central differences through the vertices' own oplus, g2o's recipe base_multi_edge.hpp:114-169), so
nothing is shared with the analytic Jacobians of the C restatement.  PARITY UNPINNED (see liba_oracle.c).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
G = np.array([0.0, 0.0, -float(f32(9.81))])


def hat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def normalize(R):
    u, _, vt = np.linalg.svd(R)
    return u @ vt


def exp_so3(w):
    d2 = float(w @ w)
    d = np.sqrt(d2)
    W = hat(w)
    if d < 1e-5:
        return normalize(np.eye(3) + W + 0.5 * W @ W)
    return normalize(np.eye(3) + W * np.sin(d) / d + W @ W * (1.0 - np.cos(d)) / d2)


def log_so3(R):
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
    c = (np.trace(R) - 1.0) * 0.5
    if c > 1 or c < -1:
        return w
    th = np.arccos(c)
    s = np.sin(th)
    return w if abs(s) < 1e-5 else th * w / s


def so3f_exp(v):
    v = v.astype(f32)
    th2 = f32(v @ v)
    if th2 < f32(1e-5) * f32(1e-5):
        im = f32(0.5) - f32(1.0 / 48.0) * th2
        re = f32(1) - f32(1.0 / 8.0) * th2
    else:
        th = f32(np.sqrt(th2))
        im = f32(np.sin(f32(0.5) * th)) / th
        re = f32(np.cos(f32(0.5) * th))
    x, y, z, w = (im * v[0], im * v[1], im * v[2], re)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=f32)


class State:
    def __init__(self, w):
        self.w = w
        self.Rwb, self.twb = w.pose_Rwb.reshape(-1, 3, 3).copy(), w.pose_twb.reshape(-1, 3).copy()
        self.Rcw, self.tcw = w.pose_Rcw.reshape(-1, 3, 3).copy(), w.pose_tcw.reshape(-1, 3).copy()
        self.vel, self.bg, self.ba = w.vel.reshape(-1, 3).copy(), w.bias_g.reshape(-1, 3).copy(), w.bias_a.reshape(-1, 3).copy()
        self.X = w.points.copy()

    def copy(self):
        s = State.__new__(State)
        s.w = self.w
        for k in ("Rwb", "twb", "Rcw", "tcw", "vel", "bg", "ba", "X"):
            setattr(s, k, getattr(self, k).copy())
        return s

    def oplus(self, x):
        w, N = self.w, self.w.n_opt
        Rcb, tcb = w.Rcb.reshape(3, 3), w.tcb
        for k in range(N):
            ur, ut = x[6 * k:6 * k + 3], x[6 * k + 3:6 * k + 6]
            if np.any(ur != 0) or np.any(ut != 0):
                self.twb[k] = self.twb[k] + self.Rwb[k] @ ut
                self.Rwb[k] = self.Rwb[k] @ exp_so3(ur)
                Rbw = self.Rwb[k].T
                self.Rcw[k] = Rcb @ Rbw
                self.tcw[k] = Rcb @ (-Rbw @ self.twb[k]) + tcb
            o = 6 * N + 9 * k
            self.vel[k] += x[o:o + 3]; self.bg[k] += x[o + 3:o + 6]; self.ba[k] += x[o + 6:o + 9]
        self.X = self.X + x[15 * N:].reshape(-1, 3)


def inertial_residual(st, l):
    w = st.w
    rec = w.link_preint[l]
    a, c = int(w.link_prev[l]), int(w.link_cur[l])
    dT = float(rec[0])
    dR, dV, dP = rec[1:10].reshape(3, 3), rec[10:13], rec[13:16]
    JRg, JVg, JVa, JPg, JPa = (rec[o:o + 9].reshape(3, 3) for o in (16, 25, 34, 43, 52))
    b = rec[61:67]
    ab = a if getattr(w, "link_bias", None) is None else int(w.link_bias[l])   # the keyframe whose bias vertices the edge hangs on
    bg1, ba1 = st.bg[ab].astype(f32), st.ba[ab].astype(f32)   # IMU::Bias holds floats
    dbg, dba = (bg1 - b[3:6]).astype(f32), (ba1 - b[0:3]).astype(f32)
    u, _, vt = np.linalg.svd((dR @ so3f_exp(JRg @ dbg)).astype(f32))
    dRb = (u @ vt).astype(np.float64)
    dVb = (dV + JVg @ dbg + JVa @ dba).astype(f32).astype(np.float64)
    dPb = (dP + JPg @ dbg + JPa @ dba).astype(f32).astype(np.float64)
    R1, R2 = st.Rwb[a], st.Rwb[c]
    er = log_so3(dRb.T @ R1.T @ R2)
    ev = R1.T @ (st.vel[c] - st.vel[a] - G * dT) - dVb
    ep = R1.T @ (st.twb[c] - st.twb[a] - st.vel[a] * dT - G * dT * dT / 2) - dPb
    return np.concatenate([er, ev, ep])


def _kb8_project(cam, kb, Xc, smooth):
    """KannalaBrandt8::project(Vector3d) (src/CameraModels/KannalaBrandt8.cpp:46-65): theta and psi are float32 values there
    (atan2f / sqrtf); ``smooth`` keeps them in double so that central differences see a differentiable function."""
    fx, fy, cx, cy = cam[:4]
    if smooth:
        theta = np.arctan2(np.hypot(Xc[0], Xc[1]), Xc[2])
        psi = np.arctan2(Xc[1], Xc[0])
    else:
        rho = np.float32(np.sqrt(np.float32(Xc[0] * Xc[0] + Xc[1] * Xc[1])))
        theta = float(np.float32(np.arctan2(float(rho), float(np.float32(Xc[2])))))
        psi = float(np.float32(np.arctan2(float(np.float32(Xc[1])), float(np.float32(Xc[0])))))
    r = theta + kb[0] * theta**3 + kb[1] * theta**5 + kb[2] * theta**7 + kb[3] * theta**9
    return fx * r * np.cos(psi) + cx, fy * r * np.sin(psi) + cy


def visual_residual(st, e, smooth=False):
    w = st.w
    k, l = int(w.edge_pose[e]), int(w.edge_point[e])
    fx, fy, cx, cy, bf = w.cam
    if w.edge_kind[e] == 2:
        # EdgeMono(1), the right camera of a fisheye rig: Tc1w = Trl Tc0w (ImuCamPose camera 1, src/G2oTypes.cc:56-66,212-218)
        T = np.asarray(w.trl).reshape(3, 4)
        Xc = T[:, :3] @ (st.Rcw[k] @ st.X[l] + st.tcw[k]) + T[:, 3]
        u, v = _kb8_project(w.cam2[:4], w.cam2[4:], Xc, smooth)
        return np.array([w.edge_obs[e, 0] - u, w.edge_obs[e, 1] - v])
    Xc = st.Rcw[k] @ st.X[l] + st.tcw[k]
    if getattr(w, "kb8", None) is not None:
        u, v = _kb8_project(w.cam[:4], w.kb8, Xc, smooth)
        return np.array([w.edge_obs[e, 0] - u, w.edge_obs[e, 1] - v])
    u, v = fx * Xc[0] / Xc[2] + cx, fy * Xc[1] / Xc[2] + cy
    if w.edge_kind[e] == 0:
        return np.array([w.edge_obs[e, 0] - u, w.edge_obs[e, 1] - v])
    return np.array([w.edge_obs[e, 0] - u, w.edge_obs[e, 1] - v, w.edge_obs[e, 2] - (u - bf / Xc[2])])


def huber(c, delta):
    d2 = delta * delta
    if c <= d2:
        return c, 1.0
    s = np.sqrt(c)
    return 2 * s * delta - d2, delta / s


def all_residual_blocks(st, smooth=False):
    """[(residual, information, huber delta or None)] in g2o edge order: inertial links first, then visual."""
    w = st.w
    out = []
    for l in range(w.n_links):
        a, c = int(w.link_prev[l]), int(w.link_cur[l])
        out.append((inertial_residual(st, l), w.link_info[l].reshape(9, 9), w.huber_inertial if w.link_robust[l] else None))
        out.append((st.bg[c] - st.bg[a], w.link_info_g[l].reshape(3, 3), None))
        out.append((st.ba[c] - st.ba[a], w.link_info_a[l].reshape(3, 3), None))
    for e in range(w.n_edges):
        r = visual_residual(st, e, smooth)
        out.append((r, np.eye(len(r)) * w.edge_info[e], w.huber_stereo if w.edge_kind[e] == 1 else w.huber_mono))
    return out


def robust_chi2(st):
    chi = 0.0
    for r, Om, delta in all_residual_blocks(st):
        c = float(r @ Om @ r)
        chi += huber(c, delta)[0] if delta is not None else c
    return chi


def numeric_dense_system(st, delta_pose=1e-6, delta_bias=2e-3):
    """Dense H, b over (poses, v/bg/ba, points) from central-difference Jacobians of every residual block."""
    w = st.w
    N = w.n_opt
    nall = 15 * N + 3 * w.n_points
    blocks0 = all_residual_blocks(st)
    H, b = np.zeros((nall, nall)), np.zeros(nall)
    # which state entries can influence anything: all optimisable ones
    J = [np.zeros((len(r), nall)) for r, _, _ in blocks0]
    for j in range(nall):
        is_bias = 6 * N <= j < 15 * N and ((j - 6 * N) % 9) >= 3
        d = delta_bias if is_bias else delta_pose
        e = np.zeros(nall); e[j] = d
        sp, sm = st.copy(), st.copy()
        sp.oplus(e); sm.oplus(-e)
        bp, bm = all_residual_blocks(sp, smooth=True), all_residual_blocks(sm, smooth=True)
        for i in range(len(blocks0)):
            J[i][:, j] = (bp[i][0] - bm[i][0]) / (2 * d)
    for (r, Om, delta), Ji in zip(blocks0, J):
        rho1 = huber(float(r @ Om @ r), delta)[1] if delta is not None else 1.0
        H += Ji.T @ (rho1 * Om) @ Ji
        b += -Ji.T @ (rho1 * Om @ r)
    return H, b


# --------------------------------------------------------------------------- PoseInertialOptimizationLastKeyFrame / LastFrame
class PoseiState:
    """State of Optimizer::PoseInertialOptimization* (src/Optimizer.cc:4499-5299): the frame's ImuCamPose / velocity / biases and the
    previous state; unknown order [cur pose 6, v 3, bg 3, ba 3] then, in mode 1, the same for the previous frame."""

    def __init__(self, f):
        self.f = f
        self.Rwb, self.twb, self.Rcw, self.tcw = f.Rwb.reshape(3, 3).copy(), f.twb.copy(), f.Rcw.reshape(3, 3).copy(), f.tcw.copy()
        self.v, self.bg, self.ba = f.vel.copy(), f.bias_g.copy(), f.bias_a.copy()
        self.pRwb, self.ptwb = f.prev_Rwb.reshape(3, 3).copy(), f.prev_twb.copy()
        self.pv, self.pbg, self.pba = f.prev_vel.copy(), f.prev_bias_g.copy(), f.prev_bias_a.copy()

    def copy(self):
        s = PoseiState.__new__(PoseiState)
        s.f = self.f
        for k in ("Rwb", "twb", "Rcw", "tcw", "v", "bg", "ba", "pRwb", "ptwb", "pv", "pbg", "pba"):
            setattr(s, k, getattr(self, k).copy())
        return s

    def oplus(self, x):
        f = self.f
        Rcb, tcb = f.Rcb.reshape(3, 3), f.tcb
        if np.any(x[:6] != 0):
            self.twb = self.twb + self.Rwb @ x[3:6]
            self.Rwb = self.Rwb @ exp_so3(x[:3])
            self.Rcw = Rcb @ self.Rwb.T
            self.tcw = Rcb @ (-self.Rwb.T @ self.twb) + tcb
        self.v, self.bg, self.ba = self.v + x[6:9], self.bg + x[9:12], self.ba + x[12:15]
        if f.mode == 1:
            self.ptwb = self.ptwb + self.pRwb @ x[18:21]
            self.pRwb = self.pRwb @ exp_so3(x[15:18])
            self.pv, self.pbg, self.pba = self.pv + x[21:24], self.pbg + x[24:27], self.pba + x[27:30]


class _LinkView:
    """Adapter: the two states of a PoseiState as a one-link window for ``inertial_residual``."""

    def __init__(self, st):
        f = st.f

        class W:
            pass
        self.w = W()
        self.w.link_preint, self.w.link_prev, self.w.link_cur = f.preint.reshape(1, -1), np.array([0]), np.array([1])
        self.Rwb, self.twb = np.stack([st.pRwb, st.Rwb]), np.stack([st.ptwb, st.twb])
        self.vel, self.bg, self.ba = np.stack([st.pv, st.v]), np.stack([st.pbg, st.bg]), np.stack([st.pba, st.ba])


def posei_visual_residual(st, e, smooth=False):
    f = st.f
    kind = int(f.edge_kind[e])
    fx, fy, cx, cy, bf = f.cam
    Xc = st.Rcw @ f.points[e] + st.tcw
    if kind == 2:
        T = np.asarray(f.trl).reshape(3, 4)
        Xc = T[:, :3] @ Xc + T[:, 3]
        u, v = _kb8_project(f.cam2[:4], f.cam2[4:], Xc, smooth)
        return np.array([f.edge_obs[e, 0] - u, f.edge_obs[e, 1] - v])
    if f.kb8 is not None:
        u, v = _kb8_project(f.cam[:4], f.kb8, Xc, smooth)
    else:
        u, v = fx * Xc[0] / Xc[2] + cx, fy * Xc[1] / Xc[2] + cy
    if kind == 1:
        return np.array([f.edge_obs[e, 0] - u, f.edge_obs[e, 1] - v, f.edge_obs[e, 2] - (u - bf / Xc[2])])
    return np.array([f.edge_obs[e, 0] - u, f.edge_obs[e, 1] - v])


def posei_residual_blocks(st, smooth=False):
    """[(residual, information, huber delta or None)]: visual edges, EdgeInertial, EdgeGyroRW, EdgeAccRW, EdgePriorPoseImu."""
    f = st.f
    out = []
    for e in range(f.n_edges):
        r = posei_visual_residual(st, e, smooth)
        out.append((r, np.eye(len(r)) * f.edge_info[e], f.huber_stereo if f.edge_kind[e] == 1 else f.huber_mono))
    out.append((inertial_residual(_LinkView(st), 0), f.info_inertial.reshape(9, 9), None))
    out.append((st.bg - st.pbg, f.info_g.reshape(3, 3), None))
    out.append((st.ba - st.pba, f.info_a.reshape(3, 3), None))
    if f.mode == 1:
        Rp = f.prior_Rwb.reshape(3, 3)
        r = np.concatenate([log_so3(Rp.T @ st.pRwb), Rp.T @ (st.ptwb - f.prior_twb), st.pv - f.prior_vel, st.pbg - f.prior_bg, st.pba - f.prior_ba])
        out.append((r, f.prior_H.reshape(15, 15), f.huber_prior))
    return out


def posei_numeric_system(st, delta_pose=1e-6, delta_bias=2e-3):
    """Dense Gauss-Newton system from central differences of every residual block (Huber weights at the current state)."""
    n = 30 if st.f.mode == 1 else 15
    blocks0 = posei_residual_blocks(st)
    J = [np.zeros((len(r), n)) for r, _, _ in blocks0]
    for j in range(n):
        d = delta_bias if (j % 15) >= 9 else delta_pose
        e = np.zeros(n); e[j] = d
        sp, sm = st.copy(), st.copy()
        sp.oplus(e); sm.oplus(-e)
        bp, bm = posei_residual_blocks(sp, smooth=True), posei_residual_blocks(sm, smooth=True)
        for i in range(len(blocks0)):
            J[i][:, j] = (bp[i][0] - bm[i][0]) / (2 * d)
    H, b = np.zeros((n, n)), np.zeros(n)
    for (r, Om, delta), Ji in zip(blocks0, J):
        rho1 = huber(float(r @ Om @ r), delta)[1] if delta is not None else 1.0
        H += Ji.T @ (rho1 * Om) @ Ji
        b += -Ji.T @ (rho1 * Om @ r)
    return H, b


def posei_optimize(f, delta_pose=1e-6, delta_bias=2e-3):
    """The whole of PoseInertialOptimizationLastKeyFrame / LastFrame (src/Optimizer.cc:4499-5299) in numpy, independent of the C
    restatement: Gauss-Newton with CENTRAL-DIFFERENCE Jacobians of every residual block (smooth fisheye angles for the differences),
    numpy.linalg.solve for the step, four classify rounds with g2o's stale-error rule, the recovery pass, and the Hessian of the
    frame's ConstraintPoseImu (mode 1: over [previous, current], before Marginalize).  Returns a dict."""
    st = PoseiState(f)
    E = f.n_edges
    n = 30 if f.mode == 1 else 15
    level = np.zeros(E, dtype=bool)
    outlier = np.zeros(E, dtype=bool)
    chi2 = np.zeros(E)
    robust = True
    n_extra = 4 if f.mode == 1 else 3

    def depth_ok(s, e):
        X = f.points[e]
        R, t = s.Rcw, s.tcw
        if f.edge_kind[e] == 2:
            T = np.asarray(f.trl).reshape(3, 4)
            R, t = T[:, :3] @ R, T[:, :3] @ t + T[:, 3]
        return (R[2] @ X + t[2]) > 0.0

    def jac(s, block_fn):
        r0 = block_fn(s)
        J = np.zeros((len(r0), n))
        for j in range(n):
            d = delta_bias if (j % 15) >= 9 else delta_pose
            e = np.zeros(n); e[j] = d
            sp, sm = s.copy(), s.copy()
            sp.oplus(e); sm.oplus(-e)
            J[:, j] = (block_fn(sp) - block_fn(sm)) / (2 * d)
        return J

    def all_jacobians(s, skip):
        """Central-difference Jacobian of every residual block that is not skipped, one state perturbation per column."""
        base = posei_residual_blocks(s, smooth=True)
        J = [None if skip(i) else np.zeros((len(r), n)) for i, (r, _, _) in enumerate(base)]
        for j in range(n):
            d = delta_bias if (j % 15) >= 9 else delta_pose
            e = np.zeros(n); e[j] = d
            sp, sm = s.copy(), s.copy()
            sp.oplus(e); sm.oplus(-e)
            bp, bm = posei_residual_blocks(sp, smooth=True), posei_residual_blocks(sm, smooth=True)
            for i in range(len(base)):
                if J[i] is not None:
                    J[i][:, j] = (bp[i][0] - bm[i][0]) / (2 * d)
        return J
    rounds = n_bad = n_inl = 0
    for rnd in range(4):
        for _ in range(f.iterations[rnd]):
            H, b = np.zeros((n, n)), np.zeros(n)
            blocks = posei_residual_blocks(st)                       # reference residuals (float32 fisheye angles)
            J = all_jacobians(st, lambda i: i < E and level[i])
            for i, (r, Om, delta) in enumerate(blocks):
                if i < E:
                    if level[i]:
                        continue
                    chi2[i] = float(r @ Om @ r)
                    rho1 = huber(chi2[i], delta)[1] if robust else 1.0
                else:
                    rho1 = huber(float(r @ Om @ r), delta)[1] if delta is not None else 1.0
                H += J[i].T @ (rho1 * Om) @ J[i]
                b += -J[i].T @ (rho1 * Om @ r)
            st.oplus(np.linalg.solve(H, b))
        bad = inl = 0
        chi2close = np.float32(1.5 * np.float32(f.chi2_mono[rnd]))
        for e in range(E):
            if outlier[e]:
                r = posei_visual_residual(st, e)
                chi2[e] = float(r @ r) * f.edge_info[e]
            c = np.float32(chi2[e])
            if f.edge_kind[e] != 1:
                close = bool(f.edge_close[e])
                o = (c > np.float32(f.chi2_mono[rnd]) and not close) or (close and c > chi2close) or not depth_ok(st, e)
            else:
                o = c > np.float32(f.chi2_stereo[rnd])
            outlier[e] = level[e] = o
            bad += int(o); inl += int(not o)
        n_bad, n_inl, rounds = bad, inl, rnd + 1
        if rnd == 2:
            robust = False
        if E + n_extra < 10:
            break
    if n_inl < 30 and not f.rec_init:
        n_bad = 0
        for e in range(E):
            r = posei_visual_residual(st, e)
            chi2[e] = float(r @ r) * f.edge_info[e]
            if chi2[e] < (24.0 if f.edge_kind[e] == 1 else 18.0):
                outlier[e] = False
            else:
                n_bad += 1
    # Hessian for ConstraintPoseImu: plain information, every block re-linearised at the final state; reference order [prev | cur]
    Hc = np.zeros((n, n))
    blocks = posei_residual_blocks(st)
    J = all_jacobians(st, lambda i: i < E and outlier[i])
    for i, (r, Om, delta) in enumerate(blocks):
        if J[i] is not None:
            Hc += J[i].T @ Om @ J[i]
    if f.mode == 1:
        perm = np.concatenate([np.arange(15, 30), np.arange(0, 15)])
        Hc = Hc[np.ix_(perm, perm)]
    return dict(Rwb=st.Rwb, twb=st.twb, Rcw=st.Rcw, tcw=st.tcw, vel=st.v, bias_g=st.bg, bias_a=st.ba, outlier=outlier.astype(np.uint8),
                edge_chi2=chi2, n_bad=n_bad, n_inliers=n_inl, rounds=rounds, H=Hc)


def lm_optimize(w, max_iterations=None):
    """Optimizer::LocalInertialBA's optimizer.optimize(opt_it) in numpy, independent of the C restatement: the full (not Schur-reduced)
    system over poses, velocities, biases and landmarks from central-difference Jacobians, numpy.linalg.solve, and g2o's
    Levenberg-Marquardt controller (optimization_algorithm_levenberg.cpp:61-169: user lambda, rho with the +1e-3 scale, lambda
    update by alpha / the ni doubling, at most 10 trials, Raul's three-bad-iterations stop).  Returns (state, trace dict)."""
    st = State(w)
    n = 15 * w.n_opt + 3 * w.n_points
    lam = w.lambda_init if w.lambda_init > 0 else None
    ni, n_bad = 2.0, 0
    trace = dict(chi2=[], lam=[], trials=[], chi2_initial=robust_chi2(st))
    its = w.max_iterations if max_iterations is None else max_iterations
    for it in range(its):
        current = robust_chi2(st)
        ini = current
        H, b = numeric_dense_system(st)
        if it == 0 and lam is None:
            lam = 1e-5 * np.max(np.abs(np.diag(H)))
        rho, q = 0.0, 0
        while True:
            try:
                x = np.linalg.solve(H + lam * np.eye(n), b)
                ok = True
            except np.linalg.LinAlgError:
                x, ok = np.zeros(n), False
            trial = st.copy()
            trial.oplus(x)
            temp = robust_chi2(trial) if ok else np.inf
            scale = float(x @ (lam * x + b)) + 1e-3
            rho = (current - temp) / scale
            if rho > 0 and np.isfinite(temp):
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni = 2.0
                current = temp
                st = trial
            else:
                lam *= ni
                ni *= 2
            q += 1
            if not (rho < 0 and q < 10):
                break
        trace["chi2"].append(current); trace["lam"].append(lam); trace["trials"].append(q)
        if q == 10 or rho == 0:
            break
        n_bad = n_bad + 1 if (ini - current) * 1e3 < ini else 0
        if n_bad >= 3:
            break
    trace["iterations"] = len(trace["chi2"])
    return st, trace
