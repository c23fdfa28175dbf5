// ba_chol.hip — reduced camera system of a large window: damping + blocked Cholesky + both triangular solves on the
// device in double, for 6 P > 192 (the single-workgroup solver of ba.hip holds the matrix in LDS and stops there).
//
// Reference: SparseBlock::solve (src/droid_kernels.cu:1206-1231) — Eigen SimplicialLLT on the HOST, in double, zero update
// when the factorisation fails.  Round 1 / 2 used the library factorisation (rocSOLVER through torch.linalg.cholesky_ex):
// 3.5 ms per iteration at 1194 x 1194, a chain of ~60 small unblocked-panel launches.  Here, per block column of 32:
//   panel   every workgroup factorises the 32 x 32 diagonal block redundantly (no extra launch, no dependency on another
//           workgroup) — ONE WAVE in registers, pivot columns broadcast with v_readlane, 1 / sqrt(pivot) from the hardware
//           estimate + two Newton steps instead of a library root and 32 divisions — then solves 32 rows below it against
//           the block, eight lanes per row;
//   update  the trailing lower triangle on the double-precision matrix cores (v_mfma_f64_16x16x4_f64), one wave per
//           16 x 64 block, operands from memory straight into the instruction's registers: no LDS, no barrier.
// The right-hand side rides along as row n of the matrix, so the forward substitution is part of the factorisation;
// the back substitution is one workgroup (column blocks from the last to the first, 1024 threads on the updates, the next
// diagonal block in flight during the current one, the 32-step triangular solve of a block in one wave's registers).
// Measured at 1194 x 1194 (200 keyframes, profiles/r02_ba_kernel_stats.csv): panel launches 38 x 13.5 us, updates
// 38 x 7.3 us, back substitution 0.19 ms: 0.98 ms per solve (1.48 ms with LDS-staged 64 x 64 update tiles, library root /
// divisions and LDS reads inside the two 32-step chains; 2.1 ms with barrier-paced pivots) against 3.5 ms for the library.
// Parity: unpinned, like the rest of the bundle adjustment (the reference needs Eigen, absent here); held to an fp64
// library solve in tests/test_ba.py.
#include "lgu_common.hpp"

namespace lgu {

constexpr int CH_NB = 32;    // block column width
constexpr int CH_T = 256;    // threads of the panel / update kernels
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
      if (!(d > 1e-290) || !(d < 1e290)) { bad = true; d = 1.0; }  // not positive definite: finish on a harmless pivot, x = 0 at the end
      // 1 / sqrt(d): the hardware estimate and two Newton steps (to the last bits of a double), then L_jj = d * r and
      // L_ij = a_ij * r — the library root and division are ~50 dependent instructions per pivot on this critical path
      double rs = __builtin_amdgcn_rsq(d);
      rs = rs * __builtin_fma(-0.5 * d * rs, rs, 1.5);
      rs = rs * __builtin_fma(-0.5 * d * rs, rs, 1.5);
      x[j] = (i == j ? d : x[j]) * rs;  // column j: L_ij for i > j (rows above the diagonal carry don't-care values)
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
  // (row j + 1 of the block is read from LDS while step j computes: the LDS round trip is off the 32-step chain)
  double dn[4];
#pragma unroll
  for (int a = 0; a < 4; a++) dn[a] = D[0][8 * a + l];
#pragma unroll
  for (int j = 0; j < CH_NB; j++) {
    const int aj = j >> 3, lj = j & 7;
    double dc[4];
#pragma unroll
    for (int a = 0; a < 4; a++) dc[a] = dn[a];
    if (j + 1 < CH_NB) {
#pragma unroll
      for (int a = 0; a < 4; a++)
        if (a <= ((j + 1) >> 3)) dn[a] = D[j + 1][8 * a + l];
    }
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < 4; a++) {
      if (a < aj) s += x[a] * dc[a];
      else if (a == aj) s += (l < lj) ? x[a] * dc[a] : 0.0;
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

// A[i][j] -= sum_t L[i][k0 + t] L[j][k0 + t] for the rows / columns behind the panel (rows include the right-hand side), on
// or below the diagonal.  ONE WAVE per 16-row x 64-column block on the double-precision matrix cores
// (v_mfma_f64_16x16x4_f64; layout found by tools/diag/mfma_f64_layout.hip: A lane = (row l % 16, k l / 16), B lane = (k l / 16,
// column l % 16), D register r of lane l = (row 4 r + l / 16, column l % 16)): operands go from memory straight into the
// registers the instruction reads — lane (i, q) takes the 64 contiguous bytes of columns k0 + 8 q .. + 7 of its row, and step
// s multiplies element s of every lane (the order of the 32 products of a sum is free, A and B use the same one) — the
// accumulators start as the tile itself and the panel enters negated, so there is no LDS, no barrier and no separate
// read-modify-write pass.  (The 64 x 64 LDS-staged tiles this replaces took 14 us per launch, a chain of global -> LDS ->
// barrier -> 256 LDS reads per thread -> read-modify-write; whole panels only: w == CH_NB.)
typedef double chd4 __attribute__((ext_vector_type(4)));
typedef double chd2 __attribute__((ext_vector_type(2)));
constexpr int CH_UR = 16, CH_UC = 64;
__global__ __launch_bounds__(kWave) void chol_update_kernel(double* A, double* ext, const double* __restrict__ stage, int n, int k0, int w) {
  const int lane = threadIdx.x;
  if (blockIdx.x == 0 && blockIdx.y == 0)  // the factor of the diagonal block, staged by the panel launch
    for (int idx = lane; idx < CH_NB * CH_NB; idx += kWave) {
      const int i = idx / CH_NB, j = idx - i * CH_NB;
      if (i < w && j <= i) A[(size_t)(k0 + i) * n + k0 + j] = stage[idx];
    }
  const int k1 = k0 + w;
  const int i0 = k1 + blockIdx.y * CH_UR, j0 = k1 + blockIdx.x * CH_UC;  // y = row strip, x = column block
  if (w != CH_NB || j0 > i0 + CH_UR - 1 || i0 > n) return;  // wave-uniform
  const int li = lane & 15, q = lane >> 4;
  double av[8], bv[4][8];
  {
    const int gi = i0 + li;
    // 16-byte loads: n = 6 P is even, k0 a multiple of 32, A and ext 16-byte aligned (checked by the host entry)
    const chd2* const ar = gi <= n ? reinterpret_cast<const chd2*>(chol_row(A, ext, n, gi) + k0 + 8 * q) : nullptr;
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) {
      const chd2 v = ar ? ar[s2] : chd2{0.0, 0.0};
      av[2 * s2] = -v[0]; av[2 * s2 + 1] = -v[1];
    }
  }
#pragma unroll
  for (int jb = 0; jb < 4; jb++) {
    const int gj = j0 + 16 * jb + li;
    const chd2* const br = gj < n ? reinterpret_cast<const chd2*>(A + (size_t)gj * n + k0 + 8 * q) : nullptr;
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) {
      const chd2 v = br ? br[s2] : chd2{0.0, 0.0};
      bv[jb][2 * s2] = v[0]; bv[jb][2 * s2 + 1] = v[1];
    }
  }
  chd4 acc[4];
  double* cp[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int gi = i0 + 4 * r + q;
    cp[r] = gi <= n ? chol_row(A, ext, n, gi) : nullptr;
#pragma unroll
    for (int jb = 0; jb < 4; jb++) {
      const int gj = j0 + 16 * jb + li;
      acc[jb][r] = (cp[r] && gj < n && gj <= gi) ? cp[r][gj] : 0.0;
    }
  }
#pragma unroll
  for (int s2 = 0; s2 < 8; s2++)
#pragma unroll
    for (int jb = 0; jb < 4; jb++) acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s2], bv[jb][s2], acc[jb], 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int gi = i0 + 4 * r + q;
#pragma unroll
    for (int jb = 0; jb < 4; jb++) {
      const int gj = j0 + 16 * jb + li;
      if (cp[r] && gj < n && gj <= gi) cp[r][gj] = acc[jb][r];
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
  static_assert(CH_BT == CH_NB * CH_NB, "one thread per entry of a diagonal block");
  const int di = t / CH_NB, dj = t - di * CH_NB;
  auto diag_entry = [&](int kb) {  // this thread's entry of diagonal block kb
    const int k0 = kb * CH_NB, w = n - k0 < CH_NB ? n - k0 : CH_NB;
    return (di < w && dj <= di) ? A[(size_t)(k0 + di) * n + k0 + dj] : 0.0;
  };
  double nd = diag_entry(nblk - 1);
  for (int kb = nblk - 1; kb >= 0; kb--) {
    const int k0 = kb * CH_NB, w = n - k0 < CH_NB ? n - k0 : CH_NB;
    Dk[di * (CH_NB + 1) + dj] = nd;
    __syncthreads();
    if (kb > 0) nd = diag_entry(kb - 1);  // the next block travels during this block's solve and update
    if (t < kWave) {  // one wave: x_j = (y_j - sum_{i > j} L_ij x_i) / L_jj, j descending; lane i holds the running y_i
      double v = t < w ? ys[k0 + t] : 0.0;
      const double rinv = t < w ? 1.0 / Dk[t * (CH_NB + 1) + t] : 0.0;  // one division per lane, off the 32-step chain
      auto bcast = [](double q, int src) {  // src is wave-uniform: v_readlane, not a trip through the LDS crossbar
        const long long bits = __builtin_bit_cast(long long, q);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)bits, src);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), src);
        return __builtin_bit_cast(double, ((long long)hi << 32) | lo);
      };
      // column t of the block in registers before the chain starts (rows past w hold zeros and identity pivots do no harm:
      // their v is 0), so a step is four v_readlane, a multiplication and an FMA with no LDS round trip
      double lcol[CH_NB];
      const int tc = t & (CH_NB - 1);
#pragma unroll
      for (int j = 0; j < CH_NB; j++) lcol[j] = Dk[j * (CH_NB + 1) + tc];
#pragma unroll
      for (int j = CH_NB - 1; j >= 0; j--) {
        const double xj = bcast(v, j) * bcast(rinv, j);
        v = t == j ? xj : (t < j ? __builtin_fma(-lcol[j], xj, v) : v);
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
  if (((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(work)) & 15) != 0) return LGU_E_BADARG;  // 16-byte operand loads
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  double* ext = work;
  int* flag = reinterpret_cast<int*>(work + n);
  double* stage = work + n + 2;
  allow_max_dynamic_lds<&chol_backsolve_kernel>();
  hipLaunchKernelGGL(chol_prepare_kernel, dim3((n + 255) / 256), dim3(256), 0, st, A, b, ext, flag, n, lm, ep);
  for (int k0 = 0; k0 < n; k0 += CH_NB) {
    const int w = n - k0 < CH_NB ? n - k0 : CH_NB;
    const int rows = n + 1 - (k0 + w);  // rows behind the block, the right-hand side included
    hipLaunchKernelGGL(chol_panel_kernel, dim3((rows + CH_RPW - 1) / CH_RPW), dim3(CH_T), 0, st, A, ext, stage, flag, n, k0);
    hipLaunchKernelGGL(chol_update_kernel, dim3((rows + CH_UC - 1) / CH_UC, (rows + CH_UR - 1) / CH_UR), dim3(kWave), 0, st, A, ext,
                       stage, n, k0, w);
  }
  const size_t lds = sizeof(double) * ((size_t)n + CH_NB * (CH_NB + 1) + CH_NB);
  hipLaunchKernelGGL(chol_backsolve_kernel, dim3(1), dim3(CH_BT), lds, st, A, ext, flag, x, n);
  return launch_status();
}

}  // extern "C"
