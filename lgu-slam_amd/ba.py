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
import os

import numpy as np
import torch

from . import _lib
from .ops import _check, _check_dtype, _ptr, _stream

_ALPHA = 0.05  # droid_kernels.cu:1394


def _segments(ix, jx):
    """accum_cuda's bookkeeping (:948-982): for every j, the rows n with ix[n] == jx[j] (rows in ascending n)."""
    ix = np.asarray(ix, dtype=np.int64)
    jx = np.asarray(jx, dtype=np.int64)
    order = np.argsort(ix, kind="stable")
    sx = ix[order]
    lo = np.searchsorted(sx, jx, side="left")
    hi = np.searchsorted(sx, jx, side="right")
    cnt = hi - lo
    ptrs = np.concatenate([[0], np.cumsum(cnt)])
    cols = np.concatenate([order[a:b] for a, b in zip(lo, hi)]) if len(jx) and cnt.sum() else np.zeros(0, np.int64)
    return ptrs, cols


class _Accum:
    """out[j] = sum of data[n] over ix[n] == jx[j] (accum_cuda :948-998); the segment tables are built once."""

    def __init__(self, lib, ix, jx, dev):
        ptrs, cols = _segments(ix, jx)
        self.lib, self.n = lib, len(jx)
        self.ptrs = torch.from_numpy(np.ascontiguousarray(ptrs, dtype=np.int64)).to(dev)
        self.cols = torch.from_numpy(np.ascontiguousarray(cols if len(cols) else [0], dtype=np.int64)).to(dev)

    def __call__(self, data, st, out=None):
        if out is None:
            out = torch.empty((self.n, data.shape[1]), dtype=torch.float32, device=data.device)
        if self.n:
            _lib.check(self.lib.lgu_ba_accum_f32(_ptr(data), _ptr(self.ptrs), _ptr(self.cols), _ptr(out), self.n, data.shape[1], st),
                       "ba accum")
        return out


class _ScatterSum:
    """Deterministic `out[dst] += sign * sum(rows)` in double (lgu_ba_scatter_sum_f64): `dest[n]` = destination row of
    input row n, or negative to drop it.  Tables built once per call; rows are summed in input order."""

    def __init__(self, lib, dest, dev):
        dest = np.asarray(dest, dtype=np.int64)
        rows = np.nonzero(dest >= 0)[0]
        order = rows[np.argsort(dest[rows], kind="stable")]          # rows grouped by destination, ascending row inside
        keys, counts = np.unique(dest[order], return_counts=True)
        ptrs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.lib, self.m = lib, len(keys)
        self.ptrs = torch.from_numpy(ptrs).to(dev)
        self.idxs = torch.from_numpy(np.ascontiguousarray(order if len(order) else [0], dtype=np.int64)).to(dev)
        self.dst = torch.from_numpy(np.ascontiguousarray(keys if len(keys) else [0], dtype=np.int64)).to(dev)

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


def _csr(dest, ndst, dev):
    """Rows grouped by destination (ascending row inside a group) as CSR over ALL ndst destinations: (ptr, idx) on dev.
    Rows with a destination outside [0, ndst) are dropped."""
    dest = np.asarray(dest, dtype=np.int64)
    rows = np.nonzero((dest >= 0) & (dest < ndst))[0]
    order = rows[np.argsort(dest[rows], kind="stable")]
    ptr = np.concatenate([[0], np.cumsum(np.bincount(dest[order], minlength=ndst))]).astype(np.int64)
    idx = np.ascontiguousarray(order if len(order) else [0], dtype=np.int64)
    return torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)


class _Plan:
    """Everything of a BA call that depends only on the graph (ii, jj, t0, t1, motion_only): the index tables of the
    deterministic assembly, of the accumulations and of the Schur pair enumeration, resident on the device."""

    def __init__(self, lib, ii_h, jj_h, t0, t1, motion_only, dev):
        P = t1 - t0
        ts_h = np.arange(t0, t1, dtype=np.int64)
        ii_exp_h, jj_exp_h = np.concatenate([ts_h, ii_h]), np.concatenate([ts_h, jj_h])
        kx_h, kk_h = np.unique(ii_exp_h, return_inverse=True)           # :1340-1344
        kk_h = kk_h.astype(np.int64)
        self.kx = torch.from_numpy(kx_h).to(dev)
        self.kk = torch.from_numpy(kk_h).to(dev)
        self.K = int(kx_h.shape[0])
        # block indices of the pose-pose system; blocks of poses before t0 are dropped (update_lhs / update_rhs)
        bi_h = np.concatenate([ii_h, ii_h, jj_h, jj_h]) - t0
        bj_h = np.concatenate([ii_h, jj_h, ii_h, jj_h]) - t0
        self.csr_H = _csr(np.where((bi_h >= 0) & (bj_h >= 0) & (bi_h < P) & (bj_h < P), bi_h * P + bj_h, -1), P * P, dev)
        self.csr_v = _csr(np.concatenate([ii_h, jj_h]) - t0, P, dev)
        self.csr_S = self.csr_sv = None
        if motion_only:
            return
        # schur_block's pair enumeration (:1260-1290): E entries n, m meeting in the same depth frame
        # (grouped by depth frame: the same set of (n, m) pairs as the reference's P x P double loop, in O(pairs))
        ent = np.nonzero((jj_exp_h >= t0) & (jj_exp_h < t1))[0]
        ent = ent[np.argsort(kk_h[ent], kind="stable")]
        _, starts, sizes = np.unique(kk_h[ent], return_index=True, return_counts=True)
        a_l, c_l, k_l = [], [], []
        for s0, m_ in zip(starts, sizes):                      # one small outer product per depth frame, upper triangle:
            g = ent[s0:s0 + m_]                                # S_ca = S_ac^T is read transposed by the assembly
            ua, uc = np.triu_indices(m_)
            a_l.append(g[ua]); c_l.append(g[uc])
            k_l.append(np.full(len(ua), kk_h[g[0]], np.int64))
        self.have_pairs = len(a_l) > 0
        if self.have_pairs:
            a_n, c_n, k_n = np.concatenate(a_l), np.concatenate(c_l), np.concatenate(k_l)
        else:
            a_n = c_n = k_n = np.zeros(1, np.int64)
        self.trip_t = torch.from_numpy(np.ascontiguousarray(np.stack([a_n, c_n, k_n], 1))).to(dev)
        if self.have_pairs:
            # destinations of S_ac (block (pose of a, pose of c)) and, for a != c, of its transpose (block (pose of c, pose of
            # a)); lgu_ba_assemble_f64 takes the transposed rows as negative indices -row - 1
            npair = len(a_n)
            d_ac = (jj_exp_h[a_n] - t0) * P + (jj_exp_h[c_n] - t0)
            d_ca = np.where(a_n != c_n, (jj_exp_h[c_n] - t0) * P + (jj_exp_h[a_n] - t0), -1)
            ptr, idx = _csr(np.concatenate([d_ac, d_ca]), P * P, dev)
            self.csr_S = (ptr, torch.where(idx < npair, idx, npair - 1 - idx))
        else:
            self.csr_S = None
        self.csr_sv = _csr(jj_exp_h - t0, P, dev)
        self.jpose = torch.from_numpy(jj_exp_h - t0).to(dev).contiguous()
        self.acc_ii_kx = _Accum(lib, ii_h, kx_h, dev)
        self.acc_ii_ts = _Accum(lib, ii_h, ts_h, dev)
        self.acc_exp_kx = _Accum(lib, ii_exp_h, kx_h, dev)


_PLANS = {}       # graph -> _Plan: the factor graph runs many BA calls on one edge set (8-16 per keyframe in the frontend)
_PLANS_MAX = 8


def _plan_for(lib, ii, jj, t0, t1, motion_only, dev):
    ii_h, jj_h = ii.cpu().numpy().astype(np.int64), jj.cpu().numpy().astype(np.int64)  # the call's one host round trip
    key = (ii_h.tobytes(), jj_h.tobytes(), int(t0), int(t1), bool(motion_only), str(dev))
    plan = _PLANS.pop(key, None)
    if plan is None:
        plan = _Plan(lib, ii_h, jj_h, int(t0), int(t1), bool(motion_only), dev)
        while len(_PLANS) >= _PLANS_MAX:
            _PLANS.pop(next(iter(_PLANS)))
    _PLANS[key] = plan   # most recently used last
    return plan


def _run(pl, lib, poses, disps, intrinsics, disps_sens, targets, weights, eta_v, ii, jj, t0, t1, iterations, lm, ep,
         motion_only, hooks=None):
    """The iterations of one call on the current stream.  (Capturable when the window fits the LDS solver; replaying a
    whole frontend-sized call as one HIP graph was measured and gives nothing — 0.59 ms either way: the call is bound
    by its kernels, not by its ~40 launches.)"""
    dev = poses.device
    E = ii.shape[0]
    ht, wd = disps.shape[1:]
    HW = ht * wd
    P = t1 - t0
    f32, f64 = torch.float32, torch.float64
    st = _stream(poses)
    kx, kk = pl.kx, pl.kk
    if not motion_only:
        have_pairs, trip_t, jpose = pl.have_pairs, pl.trip_t, pl.jpose
        acc_ii_kx, acc_ii_ts, acc_exp_kx = pl.acc_ii_kx, pl.acc_ii_ts, pl.acc_exp_kx
        K = pl.K
        Q = torch.empty((K, HW), dtype=f32, device=dev)
        w = torch.empty((K, HW), dtype=f32, device=dev)
    reduce_system, after_depth = hooks if hooks is not None else (None, None)   # sharded.sharded_ba_split
    Hs = torch.empty((4, max(E, 1), 6, 6), dtype=f32, device=dev)   # (a rank of a split BA may own no edge: E = 0)
    vs = torch.empty((2, max(E, 1), 6), dtype=f32, device=dev)
    Eii = torch.empty((max(E, 1), 6, HW), dtype=f32, device=dev)[:E]
    Eall = torch.empty((P + E, 6, HW), dtype=f32, device=dev)   # E = cat(Ei, Eij) (:1401) without the copy:
    Eij = Eall[P:]                                               # the build kernel writes Eij in place, Ei is summed into the head
    scratch = torch.empty((max(E, 1) * lib.lgu_ba_build_slices(E) * 90,), dtype=f32, device=dev)
    Cii = torch.empty((max(E, 1), HW), dtype=f32, device=dev)[:E]
    wi = torch.empty((max(E, 1), HW), dtype=f32, device=dev)[:E]
    dx = torch.zeros((P, 6), dtype=f32, device=dev)
    dz = None
    for _ in range(iterations):
        if E:
          _lib.check(lib.lgu_ba_build_f32(_ptr(targets), _ptr(weights), _ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(ii),
                                          _ptr(jj), _ptr(Hs), _ptr(vs), _ptr(Eii), _ptr(Eij), _ptr(Cii), _ptr(wi), _ptr(scratch), E, ht, wd,
                                          st),
                     "ba build")
        S = sv = None
        if not motion_only:
            _lib.check(lib.lgu_ba_depth_system_f32(_ptr(Cii), _ptr(wi), _ptr(acc_ii_kx.ptrs), _ptr(acc_ii_kx.cols), _ptr(kx), _ptr(disps),
                                                   _ptr(disps_sens), _ptr(eta_v), eta_v.shape[0], _ptr(Q), _ptr(w), K, HW, st),
                       "ba depth system")                                                          # :1394-1398
            acc_ii_ts(Eii.view(E, 6 * HW), st, out=Eall[:P].view(P, 6 * HW))             # :1400-1401
            nE = P + E
            if have_pairs:
                S = torch.empty((trip_t.shape[0], 6, 6), dtype=f32, device=dev)
                _lib.check(lib.lgu_ba_eet_f32(_ptr(Eall), _ptr(Q), _ptr(trip_t), _ptr(S), trip_t.shape[0], HW, st), "ba EEt")
            sv = torch.empty((nE, 6), dtype=f32, device=dev)
            _lib.check(lib.lgu_ba_ev_f32(_ptr(Eall), _ptr(Q), _ptr(w), _ptr(kk), _ptr(sv), nE, HW, st), "ba Ev")
        # the reduced camera system: H - S and v - sv summed per block in fixed order (double), dense layout, one launch
        Ad = torch.empty((6 * P, 6 * P), dtype=f64, device=dev)
        b = torch.empty((P, 6), dtype=f64, device=dev)
        cS, cs = pl.csr_S if S is not None else None, pl.csr_sv if sv is not None else None
        _lib.check(lib.lgu_ba_assemble_f64(_ptr(Hs), _ptr(pl.csr_H[0]), _ptr(pl.csr_H[1]),
                                           _ptr(S) if S is not None else None, _ptr(cS[0]) if cS else None, _ptr(cS[1]) if cS else None,
                                           _ptr(vs), _ptr(pl.csr_v[0]), _ptr(pl.csr_v[1]),
                                           _ptr(sv) if sv is not None else None, _ptr(cs[0]) if cs else None, _ptr(cs[1]) if cs else None,
                                           _ptr(Ad), _ptr(b), P, st), "ba assembly")
        if reduce_system is not None:   # split BA: every rank assembled its own edges' part of the system
            reduce_system(Ad, b)
        dx = torch.empty((P, 6), dtype=f32, device=dev)
        rc = lib.lgu_ba_solve_f64(_ptr(Ad), _ptr(b), _ptr(dx), P, float(lm), float(ep), st)   # one workgroup, matrix in LDS
        if rc == _lib.LGU_E_UNSUPPORTED:   # more than 32 poses in the window: blocked Cholesky over the matrix in HBM (csrc/ba_chol.hip)
            if os.environ.get("LGU_BA_LIBRARY_SOLVE", "0") != "0":   # debug / A-B only: the library factorisation of rounds 1-2
                dx = _solve(Ad, b.view(-1), lm, ep).view(P, 6).to(f32).contiguous()
            else:
                work = torch.empty(int(lib.lgu_ba_solve_blocked_work_doubles(P)), dtype=f64, device=dev)
                _lib.check(lib.lgu_ba_solve_blocked_f64(_ptr(Ad), _ptr(b), _ptr(dx), _ptr(work), P, float(lm), float(ep), st),
                           "ba blocked solve")   # Ad is this iteration's own copy: overwritten with the factor
        else:
            _lib.check(rc, "ba solve")
        if not motion_only:
            dw = torch.empty((nE, HW), dtype=f32, device=dev)
            _lib.check(lib.lgu_ba_evt_f32(_ptr(Eall), _ptr(dx), _ptr(jpose), _ptr(dw), nE, HW, P, st), "ba EvT")
            dz = torch.empty((K, HW), dtype=f32, device=dev)
            _lib.check(lib.lgu_ba_depth_update_f32(_ptr(Q), _ptr(w), _ptr(dw), _ptr(acc_exp_kx.ptrs), _ptr(acc_exp_kx.cols), _ptr(kx),
                                                   _ptr(dz), _ptr(disps), K, HW, st), "ba depth update")  # :1415, :933-946
            if after_depth is not None:   # split BA: the owners' rows of disps replace this rank's partial ones
                after_depth(disps)
        _lib.check(lib.lgu_ba_pose_retr_f32(_ptr(poses), _ptr(dx), t0, t1, st), "ba pose retraction")
    return [dx, dz]


def ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep, motion_only, _hooks=None):
    """`_hooks` (sharded.sharded_ba_split only): (reduce_system(Ad, b), after_depth(disps)) called once per iteration, and
    `eta` may then be a callable kx -> (K, ht, wd) giving the damping rows of this call's depth frames."""
    named = [poses, "poses", disps, "disps", intrinsics, "intrinsics", disps_sens, "disps_sens"]
    if ii.numel():   # (a rank of a split BA may own no edge)
        named += [targets, "targets", weights, "weights"]
    _check(*named)
    _check_dtype(ii, "ii", torch.int64)
    _check_dtype(jj, "jj", torch.int64)
    lib = _lib.load()
    dev = poses.device
    HW = disps.shape[1] * disps.shape[2]
    with torch.cuda.device(dev):
        # graph bookkeeping on the host (numpy): built once per edge set, reused while the graph does not change
        pl = _plan_for(lib, ii, jj, t0, t1, motion_only, dev)
        eta_v = None
        if not motion_only:
            if callable(eta):
                eta = eta(pl.kx)
            eta_v = eta.reshape(-1, HW).to(torch.float32).contiguous()
            if eta_v.shape[0] not in (1, pl.K):
                raise RuntimeError("ba: eta must have one row per depth frame (%d) or one row, got %d" % (pl.K, eta_v.shape[0]))
        return _run(pl, lib, poses, disps, intrinsics, disps_sens, targets, weights, eta_v, ii, jj, t0, t1, iterations, lm, ep,
                    motion_only, _hooks)
