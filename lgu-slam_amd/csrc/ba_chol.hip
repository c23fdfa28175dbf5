// ba_chol.hip — reduced camera system of a large window: damping + blocked Cholesky + both triangular solves on the
// device in double, for 6 P > 192 (the single-workgroup solver of ba.hip holds the matrix in LDS and stops there).
//
// Reference: SparseBlock::solve (src/droid_kernels.cu:1206-1231) — Eigen SimplicialLLT on the HOST, in double, zero update
// when the factorisation fails.  Round 1 / 2 used the library factorisation (rocSOLVER through torch.linalg.cholesky_ex):
// 3.5 ms per iteration at 1194 x 1194, a chain of ~60 small unblocked-panel launches.  Here, per block column of 32:
//   panel   every workgroup factorises the 32 x 32 diagonal block redundantly in LDS (no extra launch, no dependency on
//           another workgroup), then solves 32 rows below it against the block, eight lanes per row;
//   update  64 x 64 tiles of the trailing lower triangle, one workgroup each, panels staged in LDS.
// The right-hand side rides along as row n of the matrix, so the forward substitution is part of the factorisation;
// the back substitution is one workgroup (column blocks from the last to the first, 1024 threads on the updates).
// Measured at 1194 x 1194 (200 keyframes): panel launches 38 x 29 us — 26 of them the 32 barrier-paced pivots of the
// diagonal block (double-precision root and divisions on the critical path), whatever the number of rows behind it —
// updates 38 x 14 us, back substitution 0.32 ms: 2.1 ms per solve against 3.5 ms for the library.
// Parity: unpinned, like the rest of the bundle adjustment (the reference needs Eigen, absent here); held to an fp64
// library solve in tests/test_ba.py.
#include "lgu_common.hpp"

namespace lgu {

constexpr int CH_NB = 32;    // block column width
constexpr int CH_T = 256;    // threads of the panel / update kernels
constexpr int CH_TILE = 64;  // trailing-update tile
constexpr int CH_BT = 1024;  // threads of the back substitution

// A[i][i] += ep + lm * A[i][i]  (L.diagonal() += ep + lm * L.diagonal(), :1213); row n := b; flag := 0
__global__ void chol_prepare_kernel(double* A, const double* __restrict__ b, double* ext, int* flag, int n, double lm, double ep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *flag = 0;
  if (i < n) {
    const double d = A[(size_t)i * n + i];
    A[(size_t)i * n + i] = d + (ep + lm * d);
    ext[i] = b[i];
  }
}

// Row r of the working matrix: r < n is a row of A, r == n is the right-hand side.
__device__ __forceinline__ double* chol_row(double* A, double* ext, int n, int r) { return r < n ? A + (size_t)r * n : ext; }

// Sum over each group of 8 consecutive lanes (every lane of the group gets the total): three DPP steps on the two halves
// of the double (xor 1, xor 2 inside quads; the half-row mirror pairs lane i with 7 - i of the other quad).
__device__ __forceinline__ double chol_sum8(double v) {
#define LGU_DPP_ADD(ctrl)                                                                     \
  {                                                                                           \
    const long long bits = __builtin_bit_cast(long long, v);                                  \
    const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, ctrl, 0xf, 0xf, false);          \
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), ctrl, 0xf, 0xf, false);  \
    v += __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);                    \
  }
  LGU_DPP_ADD(0xB1)   // quad_perm:[1,0,3,2]
  LGU_DPP_ADD(0x4E)   // quad_perm:[2,3,0,1]
  LGU_DPP_ADD(0x141)  // row_half_mirror
#undef LGU_DPP_ADD
  return v;
}

constexpr int CH_RPW = CH_T / 8;  // rows a panel workgroup solves: eight lanes per row

__global__ __launch_bounds__(CH_T) void chol_panel_kernel(double* A, double* ext, double* stage, int* flag, int n, int k0) {
  __shared__ double D[CH_NB][CH_NB + 1];
  __shared__ double dsq[CH_NB];
  const int w = n - k0 < CH_NB ? n - k0 : CH_NB;
  const int t = threadIdx.x;
  // this lane's share of its row behind the block (r == n: the right-hand side): columns l, l + 8, l + 16, l + 24 — the
  // eight lanes of a row read 64 contiguous bytes per load; requested before the block is factorised
  const int l = t & 7;
  const int r = k0 + w + blockIdx.x * CH_RPW + (t >> 3);
  double* const rp = r <= n ? chol_row(A, ext, n, r) + k0 : nullptr;
  double x[4];
#pragma unroll
  for (int a = 0; a < 4; a++) x[a] = (rp && 8 * a + l < w) ? rp[8 * a + l] : 0.0;
  // diagonal block, lower triangle; padded to CH_NB with the identity
  for (int idx = t; idx < CH_NB * CH_NB; idx += CH_T) {
    const int i = idx / CH_NB, j = idx - i * CH_NB;
    double v = i == j ? 1.0 : 0.0;
    if (i < w && j < w && j <= i) v = A[(size_t)(k0 + i) * n + k0 + j];
    D[i][j] = v;
  }
  __syncthreads();
  // The 32 x 32 block is factorised by ONE WAVE in registers: lane i holds row i, a pivot column reaches the other lanes
  // through v_readlane (its entries are wave-uniform scalars there), no barrier and no LDS round trip inside the 32
  // pivots.  (The barrier-paced forms — two barriers per pivot, or one with running pivots in registers — cost 26 / 33 us
  // per block, whatever the number of rows behind it: 38 blocks of a 200-keyframe system.)
  if (t < kWave) {
    const int i = t & (CH_NB - 1);  // lanes 32..63 mirror lanes 0..31 and store nothing
    double x[CH_NB];
#pragma unroll
    for (int k = 0; k < CH_NB; k++) x[k] = D[i][k];
    auto bcast = [](double v, int lane) {
      const long long bits = __builtin_bit_cast(long long, v);
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)bits, lane);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), lane);
      return __builtin_bit_cast(double, ((long long)hi << 32) | lo);
    };
    bool bad = false;
#pragma unroll
    for (int j = 0; j < CH_NB; j++) {
      double d = bcast(x[j], j);  // pivot (wave-uniform)
      if (!(d > 0.0) || !(d < 1e300)) { bad = true; d = 1.0; }  // not positive definite: finish on a harmless pivot, x = 0 at the end
      d = sqrt(d);
      x[j] = i == j ? d : x[j] / d;  // column j: L_ij for i > j (rows above the diagonal carry don't-care values)
#pragma unroll
      for (int k = j + 1; k < CH_NB; k++) x[k] = __builtin_fma(-x[j], bcast(x[j], k), x[k]);  // a_ik -= L_ij L_kj
    }
    if (bad && t == 0) *flag = 1;
    if (t < CH_NB) {
#pragma unroll
      for (int k = 0; k < CH_NB; k++)
        if (k <= i) D[i][k] = x[k];
    }
  }
  __syncthreads();
  // The factor of the block goes to a staging area, not into A: other workgroups of this launch may still be reading the
  // block.  The update launch that follows copies it in.
  if (blockIdx.x == 0)
    for (int idx = t; idx < CH_NB * CH_NB; idx += CH_T) stage[idx] = D[idx / CH_NB][idx % CH_NB];
  if (t < CH_NB) dsq[t] = 1.0 / D[t][t];  // reciprocal pivots for the row solve (one division per pivot instead of one per row and pivot)
  __syncthreads();
  // rows behind the block: X L_kk^T = A_ik by forward substitution, EIGHT LANES PER ROW: lane l holds x_l, x_{l+8}, x_{l+16},
  // x_{l+24}; step j sums the products x_p L_jp (p < j) over the group with three DPP adds and the owner of column j takes
  // x_j = (a_j - sum) * (1 / L_jj).  (One thread per row was a chain of 496 LDS round trips: 39 us per launch.)
#pragma unroll
  for (int j = 0; j < CH_NB; j++) {
    constexpr int dummy = 0; (void)dummy;
    const int aj = j >> 3, lj = j & 7;
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < 4; a++) {
      if (a < aj) s += x[a] * D[j][8 * a + l];
      else if (a == aj) s += (l < lj) ? x[a] * D[j][8 * a + l] : 0.0;
    }
    s = chol_sum8(s);
    const double v = (x[aj] - s) * dsq[j];
    x[aj] = (l == lj) ? v : x[aj];
  }
  if (rp) {
#pragma unroll
    for (int a = 0; a < 4; a++)
      if (8 * a + l < w) rp[8 * a + l] = x[a];
  }
}

// A[i][j] -= sum_t L[i][k0 + t] L[j][k0 + t] for the rows / columns behind the panel (rows include the right-hand side),
// tiles on or below the diagonal.  Thread (ty, tx) owns a 4 x 4 sub-tile.
__global__ __launch_bounds__(CH_T) void chol_update_kernel(double* A, double* ext, const double* __restrict__ stage, int n, int k0, int w) {
  if (blockIdx.x > blockIdx.y) return;  // x = column tile, y = row tile
  if (blockIdx.x == 0 && blockIdx.y == 0)  // the factor of the diagonal block, staged by the panel launch
    for (int idx = threadIdx.x; idx < CH_NB * CH_NB; idx += CH_T) {
      const int i = idx / CH_NB, j = idx - i * CH_NB;
      if (i < w && j <= i) A[(size_t)(k0 + i) * n + k0 + j] = stage[idx];
    }
  __shared__ double Li[CH_TILE][CH_NB + 1], Lj[CH_TILE][CH_NB + 1];
  const int k1 = k0 + w;
  const int i0 = k1 + blockIdx.y * CH_TILE, j0 = k1 + blockIdx.x * CH_TILE;
  const int t = threadIdx.x;
  for (int idx = t; idx < CH_TILE * CH_NB; idx += CH_T) {
    const int r = idx / CH_NB, c = idx - r * CH_NB;
    const int gi = i0 + r, gj = j0 + r;
    Li[r][c] = (gi <= n && c < w) ? chol_row(A, ext, n, gi)[k0 + c] : 0.0;
    Lj[r][c] = (gj < n && c < w) ? A[(size_t)gj * n + k0 + c] : 0.0;
  }
  __syncthreads();
  const int ty = t / 16, tx = t % 16;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = 0.0;
#pragma unroll 4
  for (int c = 0; c < CH_NB; c++) {
    double li[4], lj[4];
#pragma unroll
    for (int a = 0; a < 4; a++) { li[a] = Li[ty * 4 + a][c]; lj[a] = Lj[tx * 4 + a][c]; }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) acc[a][b] += li[a] * lj[b];
  }
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const int gi = i0 + ty * 4 + a;
    if (gi > n) continue;
    double* const rp = chol_row(A, ext, n, gi);
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int gj = j0 + tx * 4 + b;
      if (gj < n && gj <= gi) rp[gj] -= acc[a][b];
    }
  }
}

// L^T x = y with y = row n after the factorisation; x as float (P, 6); zeros when the factorisation failed.
__global__ __launch_bounds__(CH_BT) void chol_backsolve_kernel(const double* __restrict__ A, const double* __restrict__ ext,
                                                              const int* __restrict__ flag, float* __restrict__ x, int n) {
  extern __shared__ double ys[];  // [n] right-hand side, then the current diagonal block and its solution
  double* const Dk = ys + n;      // [CH_NB][CH_NB + 1]
  double* const xb = Dk + CH_NB * (CH_NB + 1);
  const int t = threadIdx.x;
  if (*flag) {
    for (int i = t; i < n; i += CH_BT) x[i] = 0.f;
    return;
  }
  for (int i = t; i < n; i += CH_BT) ys[i] = ext[i];
  __syncthreads();
  const int nblk = (n + CH_NB - 1) / CH_NB;
  for (int kb = nblk - 1; kb >= 0; kb--) {
    const int k0 = kb * CH_NB, w = n - k0 < CH_NB ? n - k0 : CH_NB;
    for (int idx = t; idx < CH_NB * CH_NB; idx += CH_BT) {
      const int i = idx / CH_NB, j = idx - i * CH_NB;
      Dk[i * (CH_NB + 1) + j] = (i < w && j <= i) ? A[(size_t)(k0 + i) * n + k0 + j] : 0.0;
    }
    __syncthreads();
    if (t < kWave) {  // one wave: x_j = (y_j - sum_{i > j} L_ij x_i) / L_jj, j descending; lane i holds the running y_i
      double v = t < w ? ys[k0 + t] : 0.0;
      for (int j = w - 1; j >= 0; j--) {
        const double xj = __shfl(v, j, kWave) / Dk[j * (CH_NB + 1) + j];
        if (t == j) v = xj;
        else if (t < j) v -= Dk[j * (CH_NB + 1) + t] * xj;
      }
      if (t < w) { xb[t] = v; ys[k0 + t] = v; }
    }
    __syncthreads();
    // y[c] -= sum_i L[k0 + i][c] x_i for the columns in front of the block (rows of L are contiguous along c)
    for (int c = t; c < k0; c += CH_BT) {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      const double* const col = A + (size_t)k0 * n + c;
      int i = 0;
      for (; i + 4 <= w; i += 4) {
        s0 += col[(size_t)i * n] * xb[i];
        s1 += col[(size_t)(i + 1) * n] * xb[i + 1];
        s2 += col[(size_t)(i + 2) * n] * xb[i + 2];
        s3 += col[(size_t)(i + 3) * n] * xb[i + 3];
      }
      for (; i < w; i++) s0 += col[(size_t)i * n] * xb[i];
      ys[c] -= (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
  }
  for (int i = t; i < n; i += CH_BT) x[i] = (float)ys[i];
}

}  // namespace lgu

extern "C" {

/* Doubles of workspace lgu_ba_solve_blocked_f64 needs (the right-hand side row, a flag, one staged diagonal block). */
long long lgu_ba_solve_blocked_work_doubles(int P) { return P < 1 ? 0 : 6LL * P + 2 + 32 * 32; }

/* (A + diag(ep + lm diag A)) x = b for any window size; A (6P x 6P row-major double, symmetric) is OVERWRITTEN with the
 * factor, b is left alone, x (P, 6) float, work >= lgu_ba_solve_blocked_work_doubles(P) doubles.  x = 0 if not positive
 * definite (the reference's behaviour when Eigen reports failure).  Enqueues ~2 launches per 32 columns on `stream`. */
int lgu_ba_solve_blocked_f64(double* A, const double* b, float* x, double* work, int P, double lm, double ep, void* stream) {
  using namespace lgu;
  if (!A || !b || !x || !work || P < 1) return LGU_E_BADARG;
  const int n = 6 * P;
  if (n > 16000) return LGU_E_UNSUPPORTED;  // the back substitution keeps the right-hand side in LDS
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  double* ext = work;
  int* flag = reinterpret_cast<int*>(work + n);
  double* stage = work + n + 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_backsolve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(chol_prepare_kernel, dim3((n + 255) / 256), dim3(256), 0, st, A, b, ext, flag, n, lm, ep);
  for (int k0 = 0; k0 < n; k0 += CH_NB) {
    const int w = n - k0 < CH_NB ? n - k0 : CH_NB;
    const int rows = n + 1 - (k0 + w);  // rows behind the block, the right-hand side included
    hipLaunchKernelGGL(chol_panel_kernel, dim3((rows + CH_RPW - 1) / CH_RPW), dim3(CH_T), 0, st, A, ext, stage, flag, n, k0);
    const int nt = (rows + CH_TILE - 1) / CH_TILE;
    hipLaunchKernelGGL(chol_update_kernel, dim3(nt, nt), dim3(CH_T), 0, st, A, ext, stage, n, k0, w);
  }
  const size_t lds = sizeof(double) * ((size_t)n + CH_NB * (CH_NB + 1) + CH_NB);
  hipLaunchKernelGGL(chol_backsolve_kernel, dim3(1), dim3(CH_BT), lds, st, A, ext, flag, x, n);
  return launch_status();
}

}  // extern "C"
