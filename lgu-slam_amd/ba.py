"""Dense bundle adjustment on the device: this build's counterpart of `droid_backends.ba`
(reference src/droid.cpp:88-107 -> src/droid_kernels.cu:1314-1434 `ba_cuda`).  SURVEY §8 row f3, first version.

Same signature, same in-place updates of `poses` and `disps`, same return value `[dx, dz]`.  What differs is the
host side: the reference copies every 6x6 block to the CPU, assembles an Eigen sparse matrix in double and solves with
SimplicialLLT once per iteration (host round trips inside the loop); here the reduced camera system is assembled
(a dense (6P)^2 double matrix, summed in a fixed order so that replicated runs agree bit for bit), damped and solved
by Cholesky on the device — every stage is a HIP kernel of csrc/ba.hip.  The only host work is the graph bookkeeping (which E blocks meet in which
depth frame), built once per call from ii / jj exactly as the reference's schur_block / accum_cuda do on the CPU.

PARITY UNPINNED (the reference needs Eigen, absent in this image): checked against oracle/ba_oracle.py, which is itself
pinned only by self-consistency tests (tests/test_ba.py).
"""
import torch

from . import _lib
from .ops import _check, _check_dtype, _ptr, _stream

_ALPHA = 0.05  # droid_kernels.cu:1394


def _segments(ix, jx):
    """accum_cuda's bookkeeping (:948-982): for every j, the rows n with ix[n] == jx[j]."""
    order = sorted(range(len(ix)), key=lambda n: ix[n])
    ptrs, cols, i = [0], [], 0
    for j in jx:
        while i < len(order) and ix[order[i]] <= j:
            if ix[order[i]] == j:
                cols.append(order[i])
            i += 1
        ptrs.append(len(cols))
    return ptrs, cols


class _Accum:
    """out[j] = sum of data[n] over ix[n] == jx[j] (accum_cuda :948-998); the segment tables are built once."""

    def __init__(self, lib, ix, jx, dev):
        ptrs, cols = _segments(ix, jx)
        self.lib, self.n = lib, len(jx)
        self.ptrs = torch.tensor(ptrs, dtype=torch.int64, device=dev)
        self.cols = torch.tensor(cols if cols else [0], dtype=torch.int64, device=dev)

    def __call__(self, data, st):
        out = torch.empty((self.n, data.shape[1]), dtype=torch.float32, device=data.device)
        if self.n:
            _lib.check(self.lib.lgu_ba_accum_f32(_ptr(data), _ptr(self.ptrs), _ptr(self.cols), _ptr(out), self.n, data.shape[1], st),
                       "ba accum")
        return out


class _ScatterSum:
    """Deterministic `out[dst] += sign * sum(rows)` in double (lgu_ba_scatter_sum_f64): `dest[n]` = destination row of
    input row n, or negative to drop it.  Tables built once per call; rows are summed in input order."""

    def __init__(self, lib, dest, dev):
        groups = {}
        for n, d in enumerate(dest):
            if d >= 0:
                groups.setdefault(d, []).append(n)
        keys = sorted(groups)
        ptrs, idxs = [0], []
        for k in keys:
            idxs += groups[k]
            ptrs.append(len(idxs))
        self.lib, self.m = lib, len(keys)
        self.ptrs = torch.tensor(ptrs, dtype=torch.int64, device=dev)
        self.idxs = torch.tensor(idxs if idxs else [0], dtype=torch.int64, device=dev)
        self.dst = torch.tensor(keys if keys else [0], dtype=torch.int64, device=dev)

    def __call__(self, inp, out, sign, st):
        if self.m:
            D = inp.shape[1]
            _lib.check(self.lib.lgu_ba_scatter_sum_f64(_ptr(inp), _ptr(self.ptrs), _ptr(self.idxs), _ptr(self.dst), _ptr(out), self.m, D,
                                                       float(sign), st), "ba assembly")


def _solve(A, b, lm, ep):
    """SparseBlock::solve (:1206-1231): (A + diag(ep + lm * diag A)) x = b by Cholesky in double; zeros if not SPD."""
    L = A.clone()
    dg = torch.diagonal(L)
    dg += ep + lm * dg
    chol, info = torch.linalg.cholesky_ex(L)
    if int(info) != 0:
        return torch.zeros_like(b)
    return torch.cholesky_solve(b[:, None], chol)[:, 0]


def ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep, motion_only):
    _check(poses, "poses", disps, "disps", intrinsics, "intrinsics", disps_sens, "disps_sens", targets, "targets",
           weights, "weights")
    _check_dtype(ii, "ii", torch.int64)
    _check_dtype(jj, "jj", torch.int64)
    lib = _lib.load()
    dev = poses.device
    E = ii.shape[0]
    ht, wd = disps.shape[1:]
    HW = ht * wd
    P = t1 - t0
    f32, f64 = torch.float32, torch.float64
    with torch.cuda.device(dev):
        st = _stream(poses)
        ii_h, jj_h = ii.tolist(), jj.tolist()     # graph bookkeeping on the host, once per call
        ts_h = list(range(t0, t1))
        ii_exp_h, jj_exp_h = ts_h + ii_h, ts_h + jj_h
        kx_h = sorted(set(ii_exp_h))
        kpos = {f: n for n, f in enumerate(kx_h)}
        kk_h = [kpos[f] for f in ii_exp_h]
        kx = torch.tensor(kx_h, dtype=torch.int64, device=dev)
        kk = torch.tensor(kk_h, dtype=torch.int64, device=dev)
        # block indices of the pose-pose system; blocks of poses before t0 are dropped (update_lhs / update_rhs)
        bi_h = [v - t0 for v in ii_h + ii_h + jj_h + jj_h]
        bj_h = [v - t0 for v in ii_h + jj_h + ii_h + jj_h]
        asm_H = _ScatterSum(lib, [a * P + c if (a >= 0 and c >= 0) else -1 for a, c in zip(bi_h, bj_h)], dev)
        asm_v = _ScatterSum(lib, [v - t0 for v in ii_h + jj_h], dev)
        if not motion_only:
            # schur_block's pair enumeration (:1260-1290): E entries n, m meeting in the same depth frame
            # (grouped by depth frame: the same set of (n, m) pairs as the reference's P x P double loop, in O(pairs))
            by_k = {}
            for n, (j, k) in enumerate(zip(jj_exp_h, kk_h)):
                if t0 <= j < t1:
                    by_k.setdefault(k, []).append((j - t0, n))
            trip, pi, pj = [], [], []
            for k, lst in by_k.items():
                for ta, a in lst:
                    for tc, c in lst:
                        trip += [a, c, k]
                        pi.append(ta)
                        pj.append(tc)
            trip_t = torch.tensor(trip if trip else [0, 0, 0], dtype=torch.int64, device=dev).view(-1, 3)
            asm_S = _ScatterSum(lib, [a * P + c for a, c in zip(pi, pj)], dev)
            asm_sv = _ScatterSum(lib, [v - t0 for v in jj_exp_h], dev)
            jpose = torch.tensor(jj_exp_h, dtype=torch.int64, device=dev) - t0
            m = (disps_sens[kx] > 0).to(f32).view(-1, HW)
            eta_v = eta.reshape(-1, HW).to(f32)
            acc_ii_kx = _Accum(lib, ii_h, kx_h, dev)
            acc_ii_ts = _Accum(lib, ii_h, ts_h, dev)
            acc_exp_kx = _Accum(lib, ii_exp_h, kx_h, dev)

        Hs = torch.empty((4, E, 6, 6), dtype=f32, device=dev)
        vs = torch.empty((2, E, 6), dtype=f32, device=dev)
        Eii = torch.empty((E, 6, HW), dtype=f32, device=dev)
        Eij = torch.empty((E, 6, HW), dtype=f32, device=dev)
        Cii = torch.empty((E, HW), dtype=f32, device=dev)
        wi = torch.empty((E, HW), dtype=f32, device=dev)
        dx = torch.zeros((P, 6), dtype=f32, device=dev)
        dz = None
        for _ in range(iterations):
            _lib.check(lib.lgu_ba_build_f32(_ptr(targets), _ptr(weights), _ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(ii),
                                            _ptr(jj), _ptr(Hs), _ptr(vs), _ptr(Eii), _ptr(Eij), _ptr(Cii), _ptr(wi), E, ht, wd, st),
                       "ba build")
            A = torch.zeros((P * P, 36), dtype=f64, device=dev)
            asm_H(Hs.view(-1, 36), A, 1.0, st)
            b = torch.zeros((P, 6), dtype=f64, device=dev)
            asm_v(vs.view(-1, 6), b, 1.0, st)
            if not motion_only:
                C = acc_ii_kx(Cii, st) + m * _ALPHA + (1 - m) * eta_v                       # :1396
                w = acc_ii_kx(wi, st) - m * _ALPHA * (disps[kx] - disps_sens[kx]).view(-1, HW)   # :1397
                Q = (1.0 / C).contiguous()
                w = w.contiguous()
                Ei = acc_ii_ts(Eii.view(E, 6 * HW), st).view(P, 6, HW)                     # :1400
                Eall = torch.cat([Ei, Eij], 0).contiguous()                                             # :1401
                nE = Eall.shape[0]
                S = torch.empty((trip_t.shape[0], 6, 6), dtype=f32, device=dev)
                if trip:
                    _lib.check(lib.lgu_ba_eet_f32(_ptr(Eall), _ptr(Q), _ptr(trip_t), _ptr(S), trip_t.shape[0], HW, st), "ba EEt")
                    asm_S(S.view(-1, 36), A, -1.0, st)
                sv = torch.empty((nE, 6), dtype=f32, device=dev)
                _lib.check(lib.lgu_ba_ev_f32(_ptr(Eall), _ptr(Q), _ptr(w), _ptr(kk), _ptr(sv), nE, HW, st), "ba Ev")
                asm_sv(sv, b, -1.0, st)
            Ad = A.view(P, P, 6, 6).permute(0, 2, 1, 3).reshape(6 * P, 6 * P).contiguous()
            dx = torch.empty((P, 6), dtype=f32, device=dev)
            rc = lib.lgu_ba_solve_f64(_ptr(Ad), _ptr(b), _ptr(dx), P, float(lm), float(ep), st)   # one workgroup, matrix in LDS
            if rc == _lib.LGU_E_UNSUPPORTED:   # more than 21 poses in the window: library Cholesky on the device
                dx = _solve(Ad, b.view(-1), lm, ep).view(P, 6).to(f32).contiguous()
            else:
                _lib.check(rc, "ba solve")
            if not motion_only:
                dw = torch.empty((nE, HW), dtype=f32, device=dev)
                _lib.check(lib.lgu_ba_evt_f32(_ptr(Eall), _ptr(dx), _ptr(jpose.contiguous()), _ptr(dw), nE, HW, P, st), "ba EvT")
                dz = (Q * (w - acc_exp_kx(dw, st))).contiguous()                       # :1415
            _lib.check(lib.lgu_ba_pose_retr_f32(_ptr(poses), _ptr(dx), t0, t1, st), "ba pose retraction")
            if not motion_only:
                _lib.check(lib.lgu_ba_disp_retr_f32(_ptr(disps), _ptr(dz), _ptr(kx), kx.shape[0], HW, st), "ba disp retraction")
    return [dx, dz]
