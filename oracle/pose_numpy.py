"""Independent numpy model of Optimizer::PoseOptimization (reference src/Optimizer.cc:815-1114) -- TEST INFRASTRUCTURE ONLY.

It shares nothing with oracle/pose_oracle.c: the pose is a 4x4 matrix updated through scipy's matrix exponential, the Jacobian of
every unary edge is a central difference of a smooth double residual (g2o's own numeric recipe, base_unary_edge.hpp), the 6x6 system
is solved with numpy, and the Levenberg-Marquardt controller and the four classification rounds are written out again here:
  * round r: the estimate is reset to the frame's pose (:1024-1025), optimize(its[r]) on the level-0 edges
  * an outlier's error is recomputed at the round's final pose, an inlier keeps the error the optimiser computed last -- the errors
    of the last TRIAL, accepted or not (:1030-1105, sparse_optimizer.cpp:354-419)
  * chi2 is compared as a float with the float thresholds; after round 2 the robust kernels are dropped (:1054); fewer than 10 edges
    end after the first round (:1107)
Pinhole monocular and rectified-stereo edges (EdgeSE3ProjectXYZOnlyPose, EdgeStereoSE3ProjectXYZOnlyPose: the stereo residual with the
float32 1/z and bf of types_six_dof_expmap.cpp).  Used by tests/golden/make_golden.py (fixture pose_tiny) and tests/test_oracle_pose.py."""
import numpy as np

from . import lm_numpy as lm


def _T_of(qt):
    T = np.eye(4)
    T[:3, :3] = lm.quat_to_R(np.asarray(qt[:4], dtype=np.float64))
    T[:3, 3] = qt[4:]
    return T


def _jac(kind, T, cam, X, obs, delta=1e-9):
    d = 2 if kind == lm.MONO else 3
    J = np.zeros((d, 6))
    for k in range(6):
        e = np.zeros(6)
        e[k] = delta
        J[:, k] = (lm.edge_error_smooth(kind, lm.se3_exp_matrix(e) @ T, cam, X, obs) -
                   lm.edge_error_smooth(kind, lm.se3_exp_matrix(-e) @ T, cam, X, obs)) / (2 * delta)
    return J


def _errors(f, T, level, err):
    for e in range(f.n_edges):
        if not level[e]:
            err[e] = lm.edge_error(int(f.edge_kind[e]), T, f.cam, f.points[e], f.edge_obs[e])


def _chi(f, e, r):
    return float(r @ (f.edge_info[e] * r))


def _robust_chi2(f, level, err, robust):
    s = 0.0
    for e in range(f.n_edges):
        if level[e]:
            continue
        c = _chi(f, e, err[e])
        s += lm.huber(c, f.huber_stereo if f.edge_kind[e] == lm.STEREO else f.huber_mono)[0] if robust else c
    return s


def _optimize(f, T, level, err, robust, max_iterations):
    """OptimizationAlgorithmLevenberg::solve on the one pose vertex; returns T, iterations, chi2 of the last accepted state.  `err` is
    left holding the errors of the last trial."""
    if all(level):
        return T, 0, 0.0
    lam, ni, n_bad, cj, last_chi = -1.0, 2.0, 0, 0, 0.0
    ok = True
    it = 0
    while it < max_iterations and ok:
        it += 1
        _errors(f, T, level, err)
        current = _robust_chi2(f, level, err, robust)
        ini = current
        H, b = np.zeros((6, 6)), np.zeros(6)
        for e in range(f.n_edges):
            if level[e]:
                continue
            kind = int(f.edge_kind[e])
            J = _jac(kind, T, f.cam, f.points[e], f.edge_obs[e])
            r = err[e]
            w = 1.0
            if robust:
                w = lm.huber(_chi(f, e, r), f.huber_stereo if kind == lm.STEREO else f.huber_mono)[1]
            H += J.T @ (w * f.edge_info[e] * J)
            b -= J.T @ (w * f.edge_info[e] * r)
        if cj == 0:
            lam, ni, n_bad = 1e-5 * np.max(np.abs(np.diag(H))), 2.0, 0
        rho, qmax = 0.0, 0
        while True:
            A = H + lam * np.eye(6)
            good = True
            try:
                np.linalg.cholesky(A)                      # the solver fails unless every pivot is positive
                x = np.linalg.solve(A, b)
            except np.linalg.LinAlgError:
                good, x = False, np.zeros(6)
            Tn = lm.se3_exp_matrix(x) @ T
            _errors(f, Tn, level, err)
            temp = _robust_chi2(f, level, err, robust) if good else np.inf
            scale = float(x @ (lam * x + b)) + 1e-3
            rho = (current - temp) / scale
            if rho > 0 and np.isfinite(temp):
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni = 2.0
                current = temp
                T = Tn
            else:
                lam *= ni
                ni *= 2
            qmax += 1
            if not (rho < 0 and qmax < 10):
                break
        cj += 1
        last_chi = current
        if qmax == 10 or rho == 0:
            ok = False
            continue
        n_bad = n_bad + 1 if (ini - current) * 1e3 < ini else 0
        if n_bad >= 3:
            ok = False
    return T, cj, last_chi


def pose_optimize(f):
    """Returns dict(T, outlier, n_bad, iterations[4], chi2_final[4], rounds, edge_chi2)."""
    E = f.n_edges
    level = np.zeros(E, dtype=bool)
    outlier = np.zeros(E, dtype=np.uint8)
    err = [np.zeros(2 if f.edge_kind[e] != lm.STEREO else 3) for e in range(E)]
    iters, chis = np.zeros(4, dtype=np.int64), np.zeros(4)
    robust = True
    T0 = _T_of(f.pose_qt)
    T = T0
    n_bad, rounds = 0, 0
    edge_chi2 = np.zeros(E)
    for r in range(4):
        T, iters[r], chis[r] = _optimize(f, T0.copy(), level, err, robust, int(f.iterations[r]))
        rounds = r + 1
        n_bad = 0
        for e in range(E):
            if outlier[e]:
                err[e] = lm.edge_error(int(f.edge_kind[e]), T, f.cam, f.points[e], f.edge_obs[e])
            c = _chi(f, e, err[e])
            edge_chi2[e] = c
            th = np.float32(f.chi2_stereo[r] if f.edge_kind[e] == lm.STEREO else f.chi2_mono[r])
            if np.float32(c) > th:
                outlier[e], level[e] = 1, True
                n_bad += 1
            else:
                outlier[e], level[e] = 0, False
        if r == 2:
            robust = False
        if E < 10:
            break
    return dict(T=T, outlier=outlier, n_bad=n_bad, iterations=iters, chi2_final=chis, rounds=rounds, edge_chi2=edge_chi2)
