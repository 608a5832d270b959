"""Independent numpy re-derivation of the local-BA maths (TEST INFRASTRUCTURE ONLY).

Purpose: pin oracle/lba_oracle.c without the (unbuildable) reference binary.
It is deliberately written differently from the C restatement:

* rotations are 3x3 matrices obtained with scipy's matrix exponential of the
  4x4 twist (not quaternions / closed-form Rodrigues),
* Jacobians come from g2o's own central-difference recipe
  (Thirdparty/g2o/g2o/core/base_binary_edge.hpp:147-197, delta = 1e-9) as well
  as from an independent analytic chain rule,
* the normal equations are assembled as ONE dense (6P+3L)^2 matrix and solved
  with numpy.linalg.solve (no Schur complement, no LDL^T).

The Levenberg-Marquardt controller follows
Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-169 and
sparse_optimizer.cpp:354-419.  PARITY UNPINNED (see oracle/lba_oracle.c).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import expm

MONO, STEREO, BODY = 0, 1, 2


def quat_to_R(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def se3_exp_matrix(update):
    """exp of the twist (omega, upsilon) as a 4x4 matrix (g2o tangent order: rotation first)."""
    xi = np.zeros((4, 4))
    xi[:3, :3] = hat(update[:3])
    xi[:3, 3] = update[3:]
    return expm(xi)


class State:
    def __init__(self, w):
        self.P, self.F = w.n_free, w.n_fixed
        self.T = np.zeros((self.P + self.F, 4, 4))
        for i, qt in enumerate(w.pose_qt):
            self.T[i] = np.eye(4)
            self.T[i, :3, :3] = quat_to_R(qt[:4])
            self.T[i, :3, 3] = qt[4:]
        self.X = w.points.copy()

    def copy(self):
        s = State.__new__(State)
        s.P, s.F, s.T, s.X = self.P, self.F, self.T.copy(), self.X.copy()
        return s

    def oplus(self, x):
        """SparseOptimizer::update: T <- exp(d) T ; X <- X + d."""
        for i in range(self.P):
            self.T[i] = se3_exp_matrix(x[6 * i:6 * i + 6]) @ self.T[i]
        self.X = self.X + x[6 * self.P:].reshape(-1, 3)


def kb8_edge_error(T, cam, kb, X, obs):
    """Monocular edge through KannalaBrandt8::project(Vector3d) (src/CameraModels/KannalaBrandt8.cpp:45-63): theta and psi are
    float32 values there (atan2f / sqrtf on double arguments); float32 atan2 is taken as the correctly rounded value."""
    fx, fy, cx, cy = cam[:4]
    Xc = T[:3, :3] @ X + T[:3, 3]
    rho = np.float32(np.sqrt(np.float32(Xc[0] * Xc[0] + Xc[1] * Xc[1])))
    theta = float(np.float32(np.arctan2(float(rho), float(np.float32(Xc[2])))))
    psi = float(np.float32(np.arctan2(float(np.float32(Xc[1])), float(np.float32(Xc[0])))))
    r = theta + kb[0] * theta**3 + kb[1] * theta**5 + kb[2] * theta**7 + kb[3] * theta**9
    return np.array([obs[0] - (fx * r * np.cos(psi) + cx), obs[1] - (fy * r * np.sin(psi) + cy)])


def kb8_analytic_jacobians(T, cam, kb, X):
    """Chain rule on u = fx r(theta) x / rho + cx, v = fy r(theta) y / rho + cy with theta = atan2(rho, z), rho = |(x, y)|
    (an independent derivation; the reference's closed form is KannalaBrandt8::projectJac, KannalaBrandt8.cpp:147-175)."""
    fx, fy = cam[0], cam[1]
    R = T[:3, :3]
    Xc = R @ X + T[:3, 3]
    x, y, z = Xc
    rho = np.hypot(x, y)
    theta = np.arctan2(rho, z)
    r = theta + kb[0] * theta**3 + kb[1] * theta**5 + kb[2] * theta**7 + kb[3] * theta**9
    dr = 1 + 3 * kb[0] * theta**2 + 5 * kb[1] * theta**4 + 7 * kb[2] * theta**6 + 9 * kb[3] * theta**8
    d2 = rho * rho + z * z
    dtheta = np.array([z * x / (rho * d2), z * y / (rho * d2), -rho / d2])
    dcos = np.array([1 / rho - x * x / rho**3, -x * y / rho**3, 0.0])      # d(x / rho)
    dsin = np.array([-x * y / rho**3, 1 / rho - y * y / rho**3, 0.0])      # d(y / rho)
    dpi = np.vstack([fx * (dr * dtheta * x / rho + r * dcos), fy * (dr * dtheta * y / rho + r * dsin)])
    J_X = -dpi @ R
    J_xi = -dpi @ np.hstack([-hat(Xc), np.eye(3)])
    return J_X, J_xi


def _kb8_project_f32(cam, kb, Xc):
    """KannalaBrandt8::project(Vector3d) with its float32 theta / psi (see kb8_edge_error)."""
    fx, fy, cx, cy = cam[:4]
    rho = np.float32(np.sqrt(np.float32(Xc[0] * Xc[0] + Xc[1] * Xc[1])))
    theta = float(np.float32(np.arctan2(float(rho), float(np.float32(Xc[2])))))
    psi = float(np.float32(np.arctan2(float(np.float32(Xc[1])), float(np.float32(Xc[0])))))
    r = theta + kb[0] * theta**3 + kb[1] * theta**5 + kb[2] * theta**7 + kb[3] * theta**9
    return np.array([fx * r * np.cos(psi) + cx, fy * r * np.sin(psi) + cy])


def _kb8_dpi(cam, kb, Xc):
    """d (u, v) / d Xc of the KannalaBrandt8 projection by the chain rule (as in kb8_analytic_jacobians)."""
    fx, fy = cam[0], cam[1]
    x, y, z = Xc
    rho = np.hypot(x, y)
    theta = np.arctan2(rho, z)
    r = theta + kb[0] * theta**3 + kb[1] * theta**5 + kb[2] * theta**7 + kb[3] * theta**9
    dr = 1 + 3 * kb[0] * theta**2 + 5 * kb[1] * theta**4 + 7 * kb[2] * theta**6 + 9 * kb[3] * theta**8
    d2 = rho * rho + z * z
    dtheta = np.array([z * x / (rho * d2), z * y / (rho * d2), -rho / d2])
    dcos = np.array([1 / rho - x * x / rho**3, -x * y / rho**3, 0.0])
    dsin = np.array([-x * y / rho**3, 1 / rho - y * y / rho**3, 0.0])
    return np.vstack([fx * (dr * dtheta * x / rho + r * dcos), fy * (dr * dtheta * y / rho + r * dsin)])


def _trl_matrix(trl):
    T = np.eye(4)
    T[:3, :3] = quat_to_R(np.asarray(trl[:4], dtype=np.float64))
    T[:3, 3] = trl[4:]
    return T


def body_edge_error(T, cam2, trl, X, obs):
    """Right-camera observation of a fisheye stereo rig (ORB_SLAM3::EdgeSE3ProjectXYZToBody, include/OptimizableTypes.h:125-130):
    obs - project2(Trl * T * X), as one 4x4 product here (the reference multiplies the two SE3Quat first as well)."""
    Trw = _trl_matrix(trl) @ T
    Xr = Trw[:3, :3] @ X + Trw[:3, 3]
    uv = _kb8_project_f32(cam2[:4], cam2[4:], Xr)
    return np.array([obs[0] - uv[0], obs[1] - uv[1]])


def body_analytic_jacobians(T, cam2, trl, X):
    """Chain rule through Xr = Rrl (R X + t) + trl:  d err / d X = -dpi2 Rrl R,  d err / d xi = -dpi2 Rrl [ -hat(Xl) | I ]."""
    Trl = _trl_matrix(trl)
    R = T[:3, :3]
    Xl = R @ X + T[:3, 3]
    Xr = Trl[:3, :3] @ Xl + Trl[:3, 3]
    dpi = _kb8_dpi(cam2[:4], cam2[4:], Xr) @ Trl[:3, :3]
    return -dpi @ R, -dpi @ np.hstack([-hat(Xl), np.eye(3)])


def edge_error(kind, T, cam, X, obs):
    """obs - projection.  The stereo residual reproduces the float32 1/z and bf
    of g2o::EdgeStereoSE3ProjectXYZ::cam_project (types_six_dof_expmap.cpp:190-197)."""
    fx, fy, cx, cy, bf = cam
    Xc = T[:3, :3] @ X + T[:3, 3]
    if kind == MONO:
        return np.array([obs[0] - (fx * Xc[0] / Xc[2] + cx), obs[1] - (fy * Xc[1] / Xc[2] + cy)])
    invz = np.float32(1.0 / Xc[2])
    u = Xc[0] * float(invz) * fx + cx
    v = Xc[1] * float(invz) * fy + cy
    bfz = np.float32(bf) * invz
    return np.array([obs[0] - u, obs[1] - v, obs[2] - (u - float(bfz))])


def edge_error_smooth(kind, T, cam, X, obs):
    """Same residual in pure double (for differentiation)."""
    fx, fy, cx, cy, bf = cam
    Xc = T[:3, :3] @ X + T[:3, 3]
    u = fx * Xc[0] / Xc[2] + cx
    v = fy * Xc[1] / Xc[2] + cy
    if kind == MONO:
        return np.array([obs[0] - u, obs[1] - v])
    return np.array([obs[0] - u, obs[1] - v, obs[2] - (u - bf / Xc[2])])


def analytic_jacobians(kind, T, cam, X):
    """Chain rule: d err / d X and d err / d (omega, upsilon) with T <- exp(d) T."""
    fx, fy, cx, cy, bf = cam
    R = T[:3, :3]
    Xc = R @ X + T[:3, 3]
    x, y, z = Xc
    dpi = np.array([[fx / z, 0, -fx * x / z**2], [0, fy / z, -fy * y / z**2]])
    if kind == STEREO:
        dpi = np.vstack([dpi, [fx / z, 0, -fx * x / z**2 + bf / z**2]])
    J_X = -dpi @ R
    dXc_dxi = np.hstack([-hat(Xc), np.eye(3)])  # d(exp(d) Xc)/d d at 0
    J_xi = -dpi @ dXc_dxi
    return J_X, J_xi


def numeric_jacobians(kind, T, cam, X, obs, delta=1e-9):
    """g2o's BaseBinaryEdge::linearizeOplus numeric recipe (central differences)."""
    d = 2 if kind == MONO else 3
    J_X = np.zeros((d, 3))
    J_xi = np.zeros((d, 6))
    scalar = 1.0 / (2 * delta)
    for k in range(3):
        e = np.zeros(3)
        e[k] = delta
        J_X[:, k] = scalar * (edge_error_smooth(kind, T, cam, X + e, obs) - edge_error_smooth(kind, T, cam, X - e, obs))
    for k in range(6):
        e = np.zeros(6)
        e[k] = delta
        Tp = se3_exp_matrix(e) @ T
        Tm = se3_exp_matrix(-e) @ T
        J_xi[:, k] = scalar * (edge_error_smooth(kind, Tp, cam, X, obs) - edge_error_smooth(kind, Tm, cam, X, obs))
    return J_X, J_xi


def huber(e, delta):
    """RobustKernelHuber::robustify (robust_kernel_impl.cpp:78-91)."""
    dsqr = delta * delta
    if e <= dsqr:
        return e, 1.0
    s = np.sqrt(e)
    return 2 * s * delta - dsqr, delta / s


def errors(w, st):
    out = []
    for e in range(w.n_edges):
        ip, il = w.edge_pose[e], w.edge_point[e]
        if w.edge_kind[e] == BODY:
            out.append(body_edge_error(st.T[ip], w.cam2, w.trl, st.X[il], w.edge_obs[e]))
        elif getattr(w, "kb8", None) is not None and w.edge_kind[e] == MONO:
            out.append(kb8_edge_error(st.T[ip], w.pose_cam[ip], w.kb8, st.X[il], w.edge_obs[e]))
        else:
            out.append(edge_error(w.edge_kind[e], st.T[ip], w.pose_cam[ip], st.X[il], w.edge_obs[e]))
    return out


def robust_chi2(w, errs):
    chi = 0.0
    per_edge = np.zeros(w.n_edges)
    for e, r in enumerate(errs):
        c = float(r @ (w.edge_info[e] * r))
        per_edge[e] = c
        delta = w.huber_stereo if w.edge_kind[e] == STEREO else w.huber_mono   # the body edge takes thHuberMono (Optimizer.cc:1386)
        chi += huber(c, delta)[0]
    return chi, per_edge


def build_dense_system(w, st, errs):
    """Dense H (6P+3L)^2 and b, robust-weighted (first-order, base_edge.h:96-102)."""
    P, L = w.n_free, w.n_points
    n = 6 * P + 3 * L
    H = np.zeros((n, n))
    b = np.zeros(n)
    for e in range(w.n_edges):
        ip, il = w.edge_pose[e], w.edge_point[e]
        kind = w.edge_kind[e]
        r = errs[e]
        if kind == BODY:
            J_X, J_xi = body_analytic_jacobians(st.T[ip], w.cam2, w.trl, st.X[il])
        elif getattr(w, "kb8", None) is not None and kind == MONO:
            J_X, J_xi = kb8_analytic_jacobians(st.T[ip], w.pose_cam[ip], w.kb8, st.X[il])
        else:
            J_X, J_xi = analytic_jacobians(kind, st.T[ip], w.pose_cam[ip], st.X[il])
        info = w.edge_info[e]
        c = float(r @ (info * r))
        delta = w.huber_stereo if kind == STEREO else w.huber_mono
        _, rho1 = huber(c, delta)
        W = rho1 * info
        sl = slice(6 * P + 3 * il, 6 * P + 3 * il + 3)
        H[sl, sl] += J_X.T @ (W * J_X)
        b[sl] += -J_X.T @ (W * r)
        if ip < P:
            sp = slice(6 * ip, 6 * ip + 6)
            H[sp, sp] += J_xi.T @ (W * J_xi)
            H[sp, sl] += J_xi.T @ (W * J_X)
            H[sl, sp] += J_X.T @ (W * J_xi)
            b[sp] += -J_xi.T @ (W * r)
    return H, b


def lm_optimize(w, max_iterations=None):
    """SparseOptimizer::optimize + OptimizationAlgorithmLevenberg::solve with a dense solver."""
    st = State(w)
    iters = w.max_iterations if max_iterations is None else max_iterations
    lam, ni, n_bad = -1.0, 2.0, 0
    trace = dict(chi2=[], lam=[], trials=[], chi2_initial=None)
    ok = True
    it = 0
    while it < iters and ok:
        errs = errors(w, st)
        current, _ = robust_chi2(w, errs)
        ini = current
        if it == 0:
            trace["chi2_initial"] = current
        H, b = build_dense_system(w, st, errs)
        if it == 0:
            lam = w.lambda_init if w.lambda_init > 0 else 1e-5 * np.max(np.abs(np.diag(H)))
            ni, n_bad = 2.0, 0
        rho, qmax = 0.0, 0
        while True:
            backup = st.copy()
            Hd = H + lam * np.eye(H.shape[0])
            try:
                x = np.linalg.solve(Hd, b)
                ok2 = True
            except np.linalg.LinAlgError:
                x, ok2 = np.zeros_like(b), False
            st.oplus(x)
            errs_t = errors(w, st)
            temp, _ = robust_chi2(w, errs_t)
            if not ok2:
                temp = np.finfo(np.float64).max
            rho = current - temp
            scale = float(x @ (lam * x + b)) + 1e-3
            rho /= scale
            if rho > 0 and np.isfinite(temp):
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni = 2.0
                current = temp
                last_errs = errs_t
            else:
                lam *= ni
                ni *= 2
                st = backup
                last_errs = errs_t
            qmax += 1
            if not (rho < 0 and qmax < 10):
                break
        it += 1
        trace["chi2"].append(current)
        trace["lam"].append(lam)
        trace["trials"].append(qmax)
        if qmax == 10 or rho == 0:
            ok = False
            continue
        n_bad = n_bad + 1 if (ini - current) * 1e3 < ini else 0
        if n_bad >= 3:
            ok = False
    trace["iterations"] = it
    _, per_edge = robust_chi2(w, last_errs)
    return st, trace, per_edge
