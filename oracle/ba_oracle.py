"""CPU restatement (numpy) of the reference's dense bundle adjustment `droid_backends.ba`
(src/droid_kernels.cu:1314-1434 `ba_cuda` and the kernels it launches).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference implementation needs Eigen (src/droid_kernels.cu:15-17), which is not in this image, so
it cannot be built here, and the reference ships no fixtures for it.  This file follows the source line by line
(citations below) and is checked by self-consistency tests only (tests/test_ba.py: Jacobians against finite
differences, zero residual => zero update, cost decrease, recovery of perturbed poses / depths).

Conventions (reference): poses (N,7) = [tx,ty,tz, qx,qy,qz,qw] world-to-camera; disps (N,H,W) inverse depth;
intrinsics (4,) = fx,fy,cx,cy; targets / weights (E,2,H,W); eta (K,H,W) or broadcastable; ii, jj (E,) int64.
Per-pixel arithmetic is done in float32 like the kernels; sums and the linear solve in float64 (the reference sums in
float32 with a block reduction and solves in double with Eigen::SimplicialLLT).
"""
import numpy as np

MIN_DEPTH = 0.25  # droid_kernels.cu:26
f32 = np.float32


def act_so3(q, X):  # :56-67
    uv = 2.0 * np.cross(q[:3], X)
    return X + q[3] * uv + np.cross(q[:3], uv)


def rel_se3(ti, qi, tj, qj):  # :95-107  (T_ij = T_j * T_i^-1)
    qij = np.array([
        -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1],
        -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2],
        -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0],
        qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2]], dtype=f32)
    tij = (tj - act_so3(qij, ti)).astype(f32)
    return tij, qij


def adj_se3_batch(t, q, X):  # :78-93 on (..., 6) arrays
    qinv = np.array([-q[0], -q[1], -q[2], q[3]], dtype=f32)

    def act(v):
        uv = 2.0 * np.cross(qinv[:3], v)
        return v + qinv[3] * uv + np.cross(qinv[:3], uv)

    Y0 = act(X[..., :3])
    Y1 = act(X[..., 3:])
    u = np.stack([t[2] * X[..., 1] - t[1] * X[..., 2], t[0] * X[..., 2] - t[2] * X[..., 0], t[1] * X[..., 0] - t[0] * X[..., 1]], -1)
    return np.concatenate([Y0, Y1 + act(u)], -1).astype(f32)


def exp_so3(phi):  # :110-131
    theta_sq = float(np.dot(phi, phi))
    theta = np.sqrt(theta_sq)
    if theta_sq < 1e-8:
        imag = 0.5 - theta_sq / 48.0 + theta_sq * theta_sq / 3840.0
        real = 1.0 - theta_sq / 8.0 + theta_sq * theta_sq / 384.0
    else:
        imag = np.sin(0.5 * theta) / theta
        real = np.cos(0.5 * theta)
    return np.array([imag * phi[0], imag * phi[1], imag * phi[2], real])


def exp_se3(xi):  # :147-174
    q = exp_so3(xi[3:])
    tau, phi = xi[:3].copy(), xi[3:]
    theta_sq = float(np.dot(phi, phi))
    theta = np.sqrt(theta_sq)
    t = tau.copy()
    if theta > 1e-4:
        a = (1 - np.cos(theta)) / theta_sq
        tau = np.cross(phi, tau)
        t = t + a * tau
        b = (theta - np.sin(theta)) / (theta * theta_sq)
        tau = np.cross(phi, tau)
        t = t + b * tau
    return t, q


def retr_se3(xi, t, q):  # :877-895
    dt, dq = exp_se3(xi)
    q1 = np.array([
        dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1],
        dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2],
        dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0],
        dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2]])
    t1 = act_so3(dq, t) + dt
    return t1, q1


def projective_transform(targets, weights, poses, disps, intrinsics, ii, jj):
    """droid_kernels.cu:176-425.  Returns Hs (4,E,6,6), vs (2,E,6), Eii, Eij (E,6,HW), Cii, wi (E,HW) (float64 sums)."""
    E = ii.shape[0]
    H, W = disps.shape[1:]
    HW = H * W
    fx, fy, cx, cy = [f32(v) for v in intrinsics]
    Hs = np.zeros((4, E, 6, 6)); vs = np.zeros((2, E, 6))
    Eii = np.zeros((E, 6, HW), f32); Eij = np.zeros((E, 6, HW), f32)
    Cii = np.zeros((E, HW), f32); wi = np.zeros((E, HW), f32)
    v, u = np.meshgrid(np.arange(H, dtype=f32), np.arange(W, dtype=f32), indexing="ij")
    for e in range(E):
        ix, jx = int(ii[e]), int(jj[e])
        if ix == jx:  # stereo pair: fixed baseline (:218-229)
            tij = np.array([-0.1, 0, 0], f32); qij = np.array([0, 0, 0, 1], f32)
        else:
            tij, qij = rel_se3(poses[ix, :3].astype(f32), poses[ix, 3:].astype(f32), poses[jx, :3].astype(f32), poses[jx, 3:].astype(f32))
        X = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1).reshape(HW, 3).astype(f32)
        h = disps[ix].reshape(HW).astype(f32)
        uv = 2.0 * np.cross(qij[:3], X)
        Xj = (X + qij[3] * uv + np.cross(qij[:3], uv) + h[:, None] * tij[None]).astype(f32)  # actSE3 :69-76
        x, y, z = Xj[:, 0], Xj[:, 1], Xj[:, 2]
        ok = z >= MIN_DEPTH
        d = np.where(ok, 1.0 / np.where(ok, z, 1), 0).astype(f32)
        d2 = d * d
        wu = np.where(ok, f32(.001) * weights[e, 0].reshape(HW), 0).astype(f32)
        wv = np.where(ok, f32(.001) * weights[e, 1].reshape(HW), 0).astype(f32)
        ru = (targets[e, 0].reshape(HW) - (fx * d * x + cx)).astype(f32)
        rv = (targets[e, 1].reshape(HW) - (fy * d * y + cy)).astype(f32)
        zero = np.zeros_like(x)
        acc = np.zeros((12, 12)); avi = np.zeros(6); avj = np.zeros(6)
        for (f, Jj, Jz, w_, r_) in (
                (fx, np.stack([h * d, zero, -x * h * d2, -x * y * d2, 1 + x * x * d2, -y * d], -1) * fx,
                 fx * (tij[0] * d - tij[2] * (x * d2)), wu, ru),
                (fy, np.stack([zero, h * d, -y * h * d2, -1 - y * y * d2, x * y * d2, x * d], -1) * fy,
                 fy * (tij[1] * d - tij[2] * (y * d2)), wv, rv)):
            Jj = Jj.astype(f32); Jz = Jz.astype(f32)
            Cii[e] += w_ * Jz * Jz          # :301,341 (uses the weight BEFORE the stereo zeroing)
            wi[e] += w_ * r_ * Jz           # :302,342
            wk = zero if ix == jx else w_   # :304,344
            Ji = -adj_se3_batch(tij, qij, Jj)
            Jx = np.concatenate([Ji, Jj], -1).astype(np.float64)   # (HW,12): Ji first (:258-259)
            acc += np.einsum("k,kn,km->nm", wk.astype(np.float64), Jx, Jx)
            avi += np.einsum("k,kn->n", (wk * r_).astype(np.float64), Jx[:, :6])
            avj += np.einsum("k,kn->n", (wk * r_).astype(np.float64), Jx[:, 6:])
            Eii[e] += (wk * Jz)[None] * Ji.T
            Eij[e] += (wk * Jz)[None] * Jj.T
        Hs[0, e] = acc[:6, :6]; Hs[1, e] = acc[:6, 6:]; Hs[2, e] = acc[6:, :6]; Hs[3, e] = acc[6:, 6:]   # :400-421
        vs[0, e] = avi; vs[1, e] = avj
    return Hs, vs, Eii, Eij, Cii, wi


def accum(data, ix, jx):
    """accum_cuda (:948-998): out[j] = sum of data[n] over n with ix[n] == jx[j]."""
    out = np.zeros((len(jx),) + data.shape[1:], np.float64)
    for j, fr in enumerate(jx):
        sel = np.nonzero(ix == fr)[0]
        if len(sel):
            out[j] = data[sel].astype(np.float64).sum(0)
    return out


def solve_block(A, b, lm, ep):
    """SparseBlock::solve (:1206-1231): (A + diag(ep + lm*diag(A))) x = b by Cholesky; zeros when not SPD."""
    L = A.copy()
    dg = np.diag(L).copy()
    L[np.diag_indices_from(L)] = dg + ep + lm * dg
    try:
        c = np.linalg.cholesky(L)
    except np.linalg.LinAlgError:
        return np.zeros_like(b)
    return np.linalg.solve(c.T, np.linalg.solve(c, b))


def ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep, motion_only):
    """ba_cuda (:1314-1434).  poses and disps are updated IN PLACE (float32 arrays); returns (dx, dz) of the last iteration."""
    E = ii.shape[0]
    H, W = disps.shape[1:]
    HW = H * W
    P = t1 - t0
    ts = np.arange(t0, t1)
    ii_exp = np.concatenate([ts, ii]); jj_exp = np.concatenate([ts, jj])
    kx, kk_exp = np.unique(ii_exp, return_inverse=True)   # :1340-1344
    dx = dz = None
    for _ in range(iterations):
        Hs, vs, Eii, Eij, Cii, wi = projective_transform(targets, weights, poses, disps, intrinsics, ii, jj)
        A = np.zeros((6 * P, 6 * P)); b = np.zeros(6 * P)
        # :1375-1382 (update_lhs / update_rhs skip negative block indices = poses before t0)
        bi = np.concatenate([ii, ii, jj, jj]) - t0
        bj = np.concatenate([ii, jj, ii, jj]) - t0
        blocks = Hs.reshape(-1, 6, 6)
        for n in range(4 * E):
            if bi[n] >= 0 and bj[n] >= 0:
                A[6 * bi[n]:6 * bi[n] + 6, 6 * bj[n]:6 * bj[n] + 6] += blocks[n]
        vi = np.concatenate([ii, jj]) - t0
        vv = vs.reshape(-1, 6)
        for n in range(2 * E):
            if vi[n] >= 0:
                b[6 * vi[n]:6 * vi[n] + 6] += vv[n]
        if motion_only:
            dx = solve_block(A, b, lm, ep).reshape(P, 6)
        else:
            alpha = 0.05
            m = (disps_sens[kx] > 0).astype(np.float64).reshape(-1, HW)               # :1395
            C = accum(Cii, ii, kx) + m * alpha + (1 - m) * np.asarray(eta, np.float64).reshape(-1, HW)   # :1396
            w = accum(wi, ii, kx) - m * alpha * (disps[kx] - disps_sens[kx]).reshape(-1, HW)          # :1397
            Q = 1.0 / C
            Ei = accum(Eii.reshape(E, 6 * HW), ii, ts).reshape(P, 6, HW)                # :1400
            Eall = np.concatenate([Ei, Eij.astype(np.float64)], 0)                      # :1401
            # schur_block (:1240-1311)
            S = np.zeros((6 * P, 6 * P)); sv = np.zeros(6 * P)
            graph = [[] for _ in range(P)]; index = [[] for _ in range(P)]
            for n in range(len(ii_exp)):
                j = jj_exp[n]
                if t0 <= j <= t1:                                                       # :1267 (sic: <= t1)
                    t = j - t0
                    if t < P:
                        graph[t].append(kk_exp[n]); index[t].append(n)
            for i in range(P):
                for j in range(P):
                    for a, ka in zip(index[i], graph[i]):
                        for c_, kc in zip(index[j], graph[j]):
                            if ka == kc:
                                S[6 * i:6 * i + 6, 6 * j:6 * j + 6] += (Eall[a] * Q[ka][None]) @ Eall[c_].T   # EEt6x6 :1001-1056
            for n in range(len(ii_exp)):                                                # Ev6x1 :1059-1093 + update_rhs(v, jj - t0)
                t = jj_exp[n] - t0
                if t >= 0:
                    sv[6 * t:6 * t + 6] += Eall[n] @ (Q[kk_exp[n]] * w[kk_exp[n]])
            dx = solve_block(A - S, b - sv, lm, ep).reshape(P, 6)                       # :1404
            ixs = jj_exp - t0
            dw = np.zeros((len(ixs), HW))
            for n in range(len(ixs)):                                                   # EvT6x1 :1095-1115 (sic: skips index <= 0)
                if ixs[n] <= 0 or ixs[n] >= P:
                    continue
                dw[n] = Eall[n].T @ dx[ixs[n]]
            dz = Q * (w - accum(dw, ii_exp, kx))                                        # :1415
        for k in range(t0, t1):                                                         # pose_retr_kernel :898-931
            t_, q_ = retr_se3(dx[k - t0], poses[k, :3].astype(np.float64), poses[k, 3:].astype(np.float64))
            poses[k, :3] = t_; poses[k, 3:] = q_
        if not motion_only:
            for n, fr in enumerate(kx):                                                 # disp_retr_kernel :933-946
                disps[fr] = (disps[fr].reshape(HW) + dz[n]).reshape(H, W)
    return dx, dz
