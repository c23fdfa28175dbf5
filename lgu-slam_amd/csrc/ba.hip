// ba.hip — data-parallel kernels of the dense bundle adjustment (first version).
//
// Replaces, in reference src/droid_kernels.cu: projective_transform_kernel (:176-425), accum_kernel (:854-874),
// EEt6x6_kernel (:1001-1056), Ev6x1_kernel (:1059-1093), EvT6x1_kernel (:1095-1115), pose_retr_kernel (:898-931),
// disp_retr_kernel (:933-946).  The reference's host side (ba_cuda :1314-1434) ships every Hessian block to the CPU
// and solves with Eigen::SimplicialLLT; here the reduced camera system is assembled and solved on the device by the
// host glue (lgu-slam_amd/ba.py), these kernels produce and consume its operands.
//
// PARITY UNPINNED: the reference BA cannot be built in this image (Eigen absent); the kernels are held to
// oracle/ba_oracle.py, a line-by-line restatement checked by self-consistency tests (tests/test_ba.py).
//
// Mapping: one workgroup of 256 threads (4 waves) per edge / per output block, lanes over pixels with fully coalesced
// reads of the per-pixel planes; per-thread partial sums are reduced with DPP inside the wave and through LDS
// across the 4 waves (the reference's 256-entry shared-memory tree per scalar is 78 + 12 block reductions).
#include "lgu_common.hpp"

namespace lgu {

constexpr float BA_MIN_DEPTH = 0.25f;  // droid_kernels.cu:26
constexpr int BA_THREADS = 256;

__device__ __forceinline__ void cross3(const float* a, const float* b, float* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void act_so3(const float* q, const float* X, float* Y) {  // :56-67
  float uv[3], t[3];
  cross3(q, X, uv);
  uv[0] *= 2.0f; uv[1] *= 2.0f; uv[2] *= 2.0f;
  cross3(q, uv, t);
  Y[0] = X[0] + q[3] * uv[0] + t[0];
  Y[1] = X[1] + q[3] * uv[1] + t[1];
  Y[2] = X[2] + q[3] * uv[2] + t[2];
}
__device__ __forceinline__ void adj_se3(const float* t, const float* q, const float* X, float* Y) {  // :78-93
  const float qinv[4] = {-q[0], -q[1], -q[2], q[3]};
  act_so3(qinv, X, Y);
  act_so3(qinv, X + 3, Y + 3);
  const float u[3] = {t[2] * X[1] - t[1] * X[2], t[0] * X[2] - t[2] * X[0], t[1] * X[0] - t[0] * X[1]};
  float v[3];
  act_so3(qinv, u, v);
  Y[3] += v[0]; Y[4] += v[1]; Y[5] += v[2];
}
__device__ __forceinline__ void rel_se3(const float* ti, const float* qi, const float* tj, const float* qj, float* tij,
                                        float* qij) {  // :95-107
  qij[0] = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  qij[1] = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  qij[2] = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  qij[3] = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  act_so3(qij, ti, tij);
  tij[0] = tj[0] - tij[0]; tij[1] = tj[1] - tij[1]; tij[2] = tj[2] - tij[2];
}

// N sums over the workgroup's 256 threads behind ONE pair of barriers: every wave reduces its N values (DPP), lane 0 parks them
// in red[wave][i], thread i adds the four wave sums in wave order ((w0 + w1) + w2) + w3 (one sum at a time
// behind its own pair of barriers, as the first version did, cost 180 barriers per workgroup of the build kernel; same results).  Returns, in thread
// i < N, the sum of entry i.
template <int N>
__device__ __forceinline__ float block_sums(const float (&v)[N], float* red /* [4 * N] */) {
  static_assert(N <= BA_THREADS, "one result per thread");
  const int w = threadIdx.x >> 6;
  __syncthreads();  // `red` may still be read from a previous call
#pragma unroll
  for (int i = 0; i < N; i++) {
    const float sw = wave_sum_f32(v[i]);
    if ((threadIdx.x & 63) == 0) red[w * N + i] = sw;
  }
  __syncthreads();
  const int t = threadIdx.x < N ? threadIdx.x : 0;
  return red[t] + red[N + t] + red[2 * N + t] + red[3 * N + t];
}

// ---- projective transform, residuals, Jacobians, per-edge Hessian blocks (:176-425) ----
__global__ __launch_bounds__(BA_THREADS) void ba_build_kernel(
    const float* __restrict__ target, const float* __restrict__ weight, const float* __restrict__ poses,
    const float* __restrict__ disps, const float* __restrict__ intrinsics, const long long* __restrict__ ii,
    const long long* __restrict__ jj, float* __restrict__ part, float* __restrict__ Eii,
    float* __restrict__ Eij, float* __restrict__ Cii, float* __restrict__ bz, int E, int HW, int wd, int slices) {
  // grid = (edges, slices): an edge's pixels are split over `slices` workgroups so that a frontend-sized graph
  // (tens of edges) still fills the 256 CUs; each writes its 90 partial sums (78 Hessian + 6 + 6 gradient entries)
  // to part[e][slice][:], ba_build_finalize_kernel adds them in slice order (deterministic) into Hs / vs.
  __shared__ float red[4 * 78];
  const int e = blockIdx.x;
  const int chunk = (HW + slices - 1) / slices;
  const int kbeg = blockIdx.y * chunk, kend = kbeg + chunk < HW ? kbeg + chunk : HW;
  const int ix = (int)ii[e], jx = (int)jj[e];
  const float fx = intrinsics[0], fy = intrinsics[1], cx = intrinsics[2], cy = intrinsics[3];
  float tij[3], qij[4];
  if (ix == jx) {  // stereo pair: fixed baseline (:218-229)
    tij[0] = -0.1f; tij[1] = 0.f; tij[2] = 0.f;
    qij[0] = 0.f; qij[1] = 0.f; qij[2] = 0.f; qij[3] = 1.f;
  } else {
    rel_se3(poses + ix * 7, poses + ix * 7 + 3, poses + jx * 7, poses + jx * 7 + 3, tij, qij);  // uniform: scalar math
  }
  float hij[78], vi[6], vj[6];
#pragma unroll
  for (int l = 0; l < 78; l++) hij[l] = 0.f;
#pragma unroll
  for (int n = 0; n < 6; n++) vi[n] = vj[n] = 0.f;
  const size_t eo = (size_t)e * HW;
  for (int k = kbeg + threadIdx.x; k < kend; k += BA_THREADS) {
    const int i = k / wd, j = k - i * wd;
    float Xi[3] = {((float)j - cx) / fx, ((float)i - cy) / fy, 1.0f}, Xj[3];
    const float h = disps[(size_t)ix * HW + k];
    act_so3(qij, Xi, Xj);  // actSE3 :69-76
    Xj[0] += h * tij[0]; Xj[1] += h * tij[1]; Xj[2] += h * tij[2];
    const float x = Xj[0], y = Xj[1];
    const bool ok = !(Xj[2] < BA_MIN_DEPTH);
    const float d = ok ? 1.0f / Xj[2] : 0.0f, d2 = d * d;
    float wu = ok ? .001f * weight[(eo * 2) + k] : 0.0f;
    float wv = ok ? .001f * weight[(eo * 2) + HW + k] : 0.0f;
    const float ru = target[(eo * 2) + k] - (fx * d * x + cx);
    const float rv = target[(eo * 2) + HW + k] - (fy * d * y + cy);
    float Jx[12];
    float* const Ji = Jx;
    float* const Jj = Jx + 6;
    // x coordinate (:292-325)
    Jj[0] = fx * (h * d); Jj[1] = fx * 0.f; Jj[2] = fx * (-x * h * d2);
    Jj[3] = fx * (-x * y * d2); Jj[4] = fx * (1 + x * x * d2); Jj[5] = fx * (-y * d);
    float Jz = fx * (tij[0] * d - tij[2] * (x * d2));
    float cii = wu * Jz * Jz, b = wu * ru * Jz;
    if (ix == jx) wu = 0.f;
    adj_se3(tij, qij, Jj, Ji);
#pragma unroll
    for (int n = 0; n < 6; n++) Ji[n] = -Ji[n];
    {
      int l = 0;
#pragma unroll
      for (int n = 0; n < 12; n++)
#pragma unroll
        for (int m = 0; m <= n; m++) hij[l++] += wu * Jx[n] * Jx[m];
    }
    float ei[6], ej[6];
#pragma unroll
    for (int n = 0; n < 6; n++) {
      vi[n] += wu * ru * Ji[n];
      vj[n] += wu * ru * Jj[n];
      ei[n] = wu * Jz * Ji[n];
      ej[n] = wu * Jz * Jj[n];
    }
    // y coordinate (:328-365)
    Jj[0] = fy * 0.f; Jj[1] = fy * (h * d); Jj[2] = fy * (-y * h * d2);
    Jj[3] = fy * (-1 - y * y * d2); Jj[4] = fy * (x * y * d2); Jj[5] = fy * (x * d);
    Jz = fy * (tij[1] * d - tij[2] * (y * d2));
    cii += wv * Jz * Jz;
    b += wv * rv * Jz;
    if (ix == jx) wv = 0.f;
    adj_se3(tij, qij, Jj, Ji);
#pragma unroll
    for (int n = 0; n < 6; n++) Ji[n] = -Ji[n];
    {
      int l = 0;
#pragma unroll
      for (int n = 0; n < 12; n++)
#pragma unroll
        for (int m = 0; m <= n; m++) hij[l++] += wv * Jx[n] * Jx[m];
    }
#pragma unroll
    for (int n = 0; n < 6; n++) {
      vi[n] += wv * rv * Ji[n];
      vj[n] += wv * rv * Jj[n];
      Eii[(eo * 6) + (size_t)n * HW + k] = ei[n] + wv * Jz * Ji[n];
      Eij[(eo * 6) + (size_t)n * HW + k] = ej[n] + wv * Jz * Jj[n];
    }
    Cii[eo + k] = cii;
    bz[eo + k] = b;
  }
  // block reductions (:369-424) into this slice's partial record: [0,78) hij, [78,84) vi, [84,90) vj
  float* const rec = part + ((size_t)e * slices + blockIdx.y) * 90;
  {
    const float sh = block_sums(hij, red);
    if (threadIdx.x < 78) rec[threadIdx.x] = sh;
    const float si = block_sums(vi, red);
    if (threadIdx.x < 6) rec[78 + threadIdx.x] = si;
    const float sj = block_sums(vj, red);
    if (threadIdx.x < 6) rec[84 + threadIdx.x] = sj;
  }
}

// Adds the slice records of every edge in slice order and lays the blocks out as the reference does (:400-421):
// Hs (4,E,6,6) = Hii, Hij, Hji, Hjj; vs (2,E,6).  One thread per (edge, entry).
__global__ __launch_bounds__(BA_THREADS) void ba_build_finalize_kernel(const float* __restrict__ part, float* __restrict__ Hs,
                                                                       float* __restrict__ vs, int E, int slices) {
  const int t = blockIdx.x * BA_THREADS + threadIdx.x;
  if (t >= E * 90) return;
  const int e = t / 90, l = t - e * 90;
  float sum = 0.f;
  for (int sl = 0; sl < slices; sl++) sum += part[((size_t)e * slices + sl) * 90 + l];
  if (l >= 78) {
    const int n = l - 78;
    vs[((size_t)(n / 6) * E + e) * 6 + n % 6] = sum;
    return;
  }
  int n = 0;
  while ((n + 1) * (n + 2) / 2 <= l) n++;  // packed lower-triangular index l -> (n, m), m <= n
  const int m = l - n * (n + 1) / 2;
  float* const H0 = Hs + ((size_t)0 * E + e) * 36, * const H1 = Hs + ((size_t)1 * E + e) * 36;
  float* const H2 = Hs + ((size_t)2 * E + e) * 36, * const H3 = Hs + ((size_t)3 * E + e) * 36;
  if (n < 6) { H0[n * 6 + m] = sum; H0[m * 6 + n] = sum; }
  else if (m < 6) { H1[m * 6 + (n - 6)] = sum; H2[(n - 6) * 6 + m] = sum; }
  else { H3[(n - 6) * 6 + (m - 6)] = sum; H3[(m - 6) * 6 + (n - 6)] = sum; }
}

// ---- segment sums: out[j][:] = sum of inp[idxs[i]][:] for i in [ptrs[j], ptrs[j+1])  (:854-874) ----
__global__ __launch_bounds__(BA_THREADS) void ba_accum_kernel(const float* __restrict__ inp, const long long* __restrict__ ptrs,
                                                              const long long* __restrict__ idxs, float* __restrict__ out, int D) {
  // grid = (output rows, 1024-element chunks of a row): enough workgroups to fill the chip when there are few frames
  const int start = (int)ptrs[blockIdx.x], end = (int)ptrs[blockIdx.x + 1];
  const int k0 = blockIdx.y * (BA_THREADS * 4) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int k = k0 + u * BA_THREADS;
    if (k >= D) break;
    float x = 0.f;
    for (int i = start; i < end; i++) x += inp[(size_t)idxs[i] * D + k];
    out[(size_t)blockIdx.x * D + k] = x;
  }
}

// ---- depth system in one pass (ba_cuda :1394-1398): for depth frame j and pixel k
//   C = sum_{edges with ii == kx[j]} Cii + m*alpha + (1 - m)*eta,   w = sum wi - m*alpha*(disp - disp_sens),   Q = 1/C
// with m = (disp_sens > 0).  Replaces two segment sums and eight elementwise tensor passes.
__global__ __launch_bounds__(BA_THREADS) void ba_depth_system_kernel(const float* __restrict__ Cii, const float* __restrict__ wi,
                                                                     const long long* __restrict__ ptrs, const long long* __restrict__ idxs,
                                                                     const long long* __restrict__ kx, const float* __restrict__ disps,
                                                                     const float* __restrict__ sens, const float* __restrict__ eta,
                                                                     int eta_rows, float alpha, float* __restrict__ Q,
                                                                     float* __restrict__ w, int D) {
  const int j = blockIdx.x;
  const int start = (int)ptrs[j], end = (int)ptrs[j + 1];
  const size_t f = (size_t)kx[j];
  const float* const er = eta + (size_t)(eta_rows == 1 ? 0 : j) * D;
  const int k0 = blockIdx.y * (BA_THREADS * 4) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int k = k0 + u * BA_THREADS;
    if (k >= D) break;
    float c = 0.f, b = 0.f;
    for (int i = start; i < end; i++) {
      c += Cii[(size_t)idxs[i] * D + k];
      b += wi[(size_t)idxs[i] * D + k];
    }
    const float ds = sens[f * D + k], dd = disps[f * D + k];
    const float m = ds > 0.f ? 1.0f : 0.0f;
    const float C = c + m * alpha + (1.0f - m) * er[k];
    w[(size_t)j * D + k] = b - m * alpha * (dd - ds);
    Q[(size_t)j * D + k] = 1.0f / C;
  }
}

// ---- depth update in one pass (:1415, :933-946): dz = Q (w - sum_{entries of frame j} dw); disps[kx[j]] += dz ----
__global__ __launch_bounds__(BA_THREADS) void ba_depth_update_kernel(const float* __restrict__ Q, const float* __restrict__ w,
                                                                     const float* __restrict__ dw, const long long* __restrict__ ptrs,
                                                                     const long long* __restrict__ idxs, const long long* __restrict__ kx,
                                                                     float* __restrict__ dz, float* disps, int D) {
  const int j = blockIdx.x;
  const int start = (int)ptrs[j], end = (int)ptrs[j + 1];
  const size_t f = (size_t)kx[j];
  const int k0 = blockIdx.y * (BA_THREADS * 4) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int k = k0 + u * BA_THREADS;
    if (k >= D) break;
    float a = 0.f;
    for (int i = start; i < end; i++) a += dw[(size_t)idxs[i] * D + k];
    const float z = Q[(size_t)j * D + k] * (w[(size_t)j * D + k] - a);
    dz[(size_t)j * D + k] = z;
    disps[f * D + k] += z;
  }
}

// ---- deterministic assembly: out[dst[j]][:] += sign * sum of inp[idxs[i]][:], i in [ptrs[j], ptrs[j+1]), in double.
// One thread per (destination, component), rows summed in table order: the replicated BA of a sharded run must
// produce the same bits on every rank, which atomics-based index_add would not guarantee.
__global__ __launch_bounds__(BA_THREADS) void ba_scatter_sum_kernel(const float* __restrict__ inp, const long long* __restrict__ ptrs,
                                                                    const long long* __restrict__ idxs, const long long* __restrict__ dst,
                                                                    double* __restrict__ out, int m, int D, double sign) {
  const int t = blockIdx.x * BA_THREADS + threadIdx.x;
  if (t >= m * D) return;
  const int j = t / D, k = t - j * D;
  double acc = 0.0;
  for (long long i = ptrs[j]; i < ptrs[j + 1]; i++) acc += (double)inp[(size_t)idxs[i] * D + k];
  out[(size_t)dst[j] * D + k] += sign * acc;
}

// ---- the whole reduced camera system in one launch: every 6 x 6 block (and 6-vector) = (sum of its rows of the first
// source) - (sum of its rows of the second source), each sum in double in table order, written in the dense (6P x 6P)
// row-major form the solver takes.  Replaces two zero-fills, four scatter-sum launches and a permuting copy; the
// arithmetic per entry is the same ((0 + 1 * a) + (-1) * c), so the bits are too.  CSR tables over ALL destinations.
__global__ __launch_bounds__(BA_THREADS) void ba_assemble_kernel(const float* __restrict__ Hs, const long long* __restrict__ hptr,
                                                                 const long long* __restrict__ hidx, const float* __restrict__ S,
                                                                 const long long* __restrict__ sptr, const long long* __restrict__ sidx,
                                                                 const float* __restrict__ vs, const long long* __restrict__ vptr,
                                                                 const long long* __restrict__ vidx, const float* __restrict__ sv,
                                                                 const long long* __restrict__ svptr, const long long* __restrict__ svidx,
                                                                 double* __restrict__ Ad, double* __restrict__ b, int P) {
  const long long t = (long long)blockIdx.x * BA_THREADS + threadIdx.x;
  const long long nA = (long long)P * P * 36;
  if (t < nA) {
    const int d = (int)(t / 36), k = (int)(t - (long long)d * 36);
    double a = 0.0;
    for (long long i = hptr[d]; i < hptr[d + 1]; i++) a += (double)Hs[(size_t)hidx[i] * 36 + k];
    double out = 0.0;
    if (hptr[d + 1] > hptr[d]) out += 1.0 * a;
    if (S) {
      // entries e >= 0: row e of S as it is; e < 0: row -e - 1 TRANSPOSED (S_ca = S_ac^T: the Schur products are computed for
      // a <= c only).  Direct rows come first in a segment (the tables are built that way) and the two kinds are summed
      // apart, as two scatter-sum passes would
      double c = 0.0, ct = 0.0;
      bool anyd = false, anyt = false;
      const int kt = (k % 6) * 6 + k / 6;
      for (long long i = sptr[d]; i < sptr[d + 1]; i++) {
        const long long e = sidx[i];
        if (e >= 0) { c += (double)S[(size_t)e * 36 + k]; anyd = true; }
        else { ct += (double)S[(size_t)(-e - 1) * 36 + kt]; anyt = true; }
      }
      if (anyd) out += -1.0 * c;
      if (anyt) out += -1.0 * ct;
    }
    const int bi = d / P, bj = d - bi * P, r = k / 6, cidx = k - r * 6;
    Ad[((size_t)bi * 6 + r) * (6 * P) + bj * 6 + cidx] = out;
  } else if (t < nA + (long long)P * 6) {
    const int u = (int)(t - nA), d = u / 6, k = u - d * 6;
    double a = 0.0;
    for (long long i = vptr[d]; i < vptr[d + 1]; i++) a += (double)vs[(size_t)vidx[i] * 6 + k];
    double out = 0.0;
    if (vptr[d + 1] > vptr[d]) out += 1.0 * a;
    if (sv) {
      double c = 0.0;
      for (long long i = svptr[d]; i < svptr[d + 1]; i++) c += (double)sv[(size_t)svidx[i] * 6 + k];
      if (svptr[d + 1] > svptr[d]) out += -1.0 * c;
    }
    b[u] = out;
  }
}

// ---- S[b] = (E[ix] * Q[kx]) E[jx]^T over the pixels (:1001-1056) ----
__global__ __launch_bounds__(BA_THREADS) void ba_eet_kernel(const float* __restrict__ Em, const float* __restrict__ Q,
                                                            const long long* __restrict__ idx, float* __restrict__ S, int D) {
  __shared__ float red[4 * 36];
  const int ix = (int)idx[blockIdx.x * 3 + 0], jx = (int)idx[blockIdx.x * 3 + 1], kx = (int)idx[blockIdx.x * 3 + 2];
  float dS[36];
#pragma unroll
  for (int i = 0; i < 36; i++) dS[i] = 0.f;
  for (int k = threadIdx.x; k < D; k += BA_THREADS) {
    const float q = Q[(size_t)kx * D + k];
    float ei[6], ej[6];
#pragma unroll
    for (int n = 0; n < 6; n++) {
      ei[n] = Em[((size_t)ix * 6 + n) * D + k] * q;
      ej[n] = Em[((size_t)jx * 6 + n) * D + k];
    }
#pragma unroll
    for (int n = 0; n < 6; n++)
#pragma unroll
      for (int m = 0; m < 6; m++) dS[n * 6 + m] += ei[n] * ej[m];
  }
  const float ss = block_sums(dS, red);
  if (threadIdx.x < 36) S[(size_t)blockIdx.x * 36 + threadIdx.x] = ss;
}

// ---- v[n] = E[n] (Q[k(n)] * w[k(n)])  (:1059-1093; v is written, the caller zero-initialises nothing) ----
__global__ __launch_bounds__(BA_THREADS) void ba_ev_kernel(const float* __restrict__ Em, const float* __restrict__ Q,
                                                           const float* __restrict__ w, const long long* __restrict__ kk,
                                                           float* __restrict__ v, int D) {
  __shared__ float red[4 * 6];
  const int kx = (int)kk[blockIdx.x];
  float b[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = threadIdx.x; k < D; k += BA_THREADS) {
    const float qw = Q[(size_t)kx * D + k] * w[(size_t)kx * D + k];
#pragma unroll
    for (int n = 0; n < 6; n++) b[n] += qw * Em[((size_t)blockIdx.x * 6 + n) * D + k];
  }
  const float sb = block_sums(b, red);
  if (threadIdx.x < 6) v[(size_t)blockIdx.x * 6 + threadIdx.x] = sb;
}

// ---- dw[n][:] = E[n]^T x[idx[n]]; rows whose pose index is <= 0 or >= P stay zero (:1095-1115, sic) ----
__global__ __launch_bounds__(BA_THREADS) void ba_evt_kernel(const float* __restrict__ Em, const float* __restrict__ x,
                                                            const long long* __restrict__ idx, float* __restrict__ dw, int D,
                                                            int P) {
  const int ix = (int)idx[blockIdx.x];
  const bool skip = ix <= 0 || ix >= P;
  float xr[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (!skip)
#pragma unroll
    for (int n = 0; n < 6; n++) xr[n] = x[ix * 6 + n];
  for (int k = threadIdx.x; k < D; k += BA_THREADS) {
    float s = 0.f;
    if (!skip)
#pragma unroll
      for (int n = 0; n < 6; n++) s += Em[((size_t)blockIdx.x * 6 + n) * D + k] * xr[n];
    dw[(size_t)blockIdx.x * D + k] = s;
  }
}

// ---- SE3 retraction of the poses t0 .. t1-1 (:877-931) ----
__device__ __forceinline__ void exp_so3(const float* phi, float* q) {  // :110-131
  const float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float theta_p4 = theta_sq * theta_sq, theta = sqrtf(theta_sq);
  float imag, real;
  if (theta_sq < 1e-8f) {
    imag = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * theta_p4;
    real = 1.0f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * theta_p4;
  } else {
    imag = sinf(0.5f * theta) / theta;
    real = cosf(0.5f * theta);
  }
  q[0] = imag * phi[0]; q[1] = imag * phi[1]; q[2] = imag * phi[2]; q[3] = real;
}
__global__ void ba_pose_retr_kernel(float* poses, const float* __restrict__ dx, int t0, int t1) {
  for (int k = t0 + blockIdx.x * blockDim.x + threadIdx.x; k < t1; k += gridDim.x * blockDim.x) {
    float xi[6], t[3], q[4], dt[3], dq[4], t1v[3], q1[4];
#pragma unroll
    for (int n = 0; n < 6; n++) xi[n] = dx[(k - t0) * 6 + n];
#pragma unroll
    for (int n = 0; n < 3; n++) t[n] = poses[k * 7 + n];
#pragma unroll
    for (int n = 0; n < 4; n++) q[n] = poses[k * 7 + 3 + n];
    // expSE3 :147-174
    exp_so3(xi + 3, dq);
    float tau[3] = {xi[0], xi[1], xi[2]}, tmp[3];
    const float* phi = xi + 3;
    const float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2], theta = sqrtf(theta_sq);
    dt[0] = tau[0]; dt[1] = tau[1]; dt[2] = tau[2];
    if (theta > 1e-4f) {
      const float a = (1 - cosf(theta)) / theta_sq;
      cross3(phi, tau, tmp); tau[0] = tmp[0]; tau[1] = tmp[1]; tau[2] = tmp[2];
      dt[0] += a * tau[0]; dt[1] += a * tau[1]; dt[2] += a * tau[2];
      const float b = (theta - sinf(theta)) / (theta * theta_sq);
      cross3(phi, tau, tmp); tau[0] = tmp[0]; tau[1] = tmp[1]; tau[2] = tmp[2];
      dt[0] += b * tau[0]; dt[1] += b * tau[1]; dt[2] += b * tau[2];
    }
    q1[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
    q1[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
    q1[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
    q1[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
    act_so3(dq, t, t1v);
#pragma unroll
    for (int n = 0; n < 3; n++) poses[k * 7 + n] = t1v[n] + dt[n];
#pragma unroll
    for (int n = 0; n < 4; n++) poses[k * 7 + 3 + n] = q1[n];
  }
}

// ---- disps[inds[b]][:] += dz[b][:]  (:933-946) ----
__global__ __launch_bounds__(BA_THREADS) void ba_disp_retr_kernel(float* disps, const float* __restrict__ dz,
                                                                  const long long* __restrict__ inds, int HW) {
  const size_t f = (size_t)inds[blockIdx.x];
  for (int k = threadIdx.x; k < HW; k += BA_THREADS) disps[f * HW + k] += dz[(size_t)blockIdx.x * HW + k];
}

// ---- reduced camera system: damping + blocked Cholesky + solve in ONE workgroup (SparseBlock::solve :1206-1231) ----
// A (n x n, n = 6 P, row-major double, symmetric) and b (n) stay untouched; x (P,6) float.  The matrix lives in LDS
// (packed lower triangle, n <= 192 = 32 poses: 148 KB), factorised by 6 x 6 block columns: diagonal block by one thread in registers, panel solve with
// lanes over rows, trailing update with lanes over (row, column) pairs; then the two triangular solves by one wave.  Not positive
// definite (a pivot <= 0 or not finite): x = 0, as the reference does when Eigen reports failure.
constexpr int BA_SOLVE_MAXN = 192;
constexpr int BA_SOLVE_THREADS = 1024;  // one workgroup = the whole CU: 16 waves to hide the LDS latency of the updates
#define LT(i, j) Ls[(((i) * ((i) + 1)) >> 1) + (j)]  // packed lower triangle, j <= i
__global__ __launch_bounds__(BA_SOLVE_THREADS) void ba_solve_kernel(const double* __restrict__ A, const double* __restrict__ b,
                                                              float* __restrict__ x, int n, double lm, double ep) {
  extern __shared__ double Ls[];  // packed lower triangle n (n + 1) / 2, then [n] rhs
  double* const y = Ls + (((size_t)n * (n + 1)) >> 1);
  int& bad = *reinterpret_cast<int*>(y + n);
  if (threadIdx.x == 0) bad = 0;
  for (int idx = threadIdx.x; idx < n * n; idx += BA_SOLVE_THREADS) {
    const int i = idx / n, j = idx - i * n;
    if (j > i) continue;
    double v = A[idx];
    if (i == j) v += ep + lm * v;  // L.diagonal() += ep + lm * L.diagonal()
    LT(i, j) = v;
  }
  for (int i = threadIdx.x; i < n; i += BA_SOLVE_THREADS) y[i] = b[i];
  __syncthreads();
  const int nb = n / 6;
  for (int kb = 0; kb < nb; kb++) {
    const int k0 = kb * 6;
    if (threadIdx.x == 0) {  // 6 x 6 diagonal block, factorised in registers (no dependent LDS round trips)
      double a[6][6];
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) a[i][j] = LT(k0 + i, k0 + j);
#pragma unroll
      for (int k = 0; k < 6; k++) {
        double d = a[k][k];
#pragma unroll
        for (int p = 0; p < k; p++) d -= a[k][p] * a[k][p];
        if (!(d > 0.0) || !(d < 1e300)) { bad = 1; d = 1.0; }
        d = sqrt(d);
        a[k][k] = d;
#pragma unroll
        for (int i = k + 1; i < 6; i++) {
          double v = a[i][k];
#pragma unroll
          for (int p = 0; p < k; p++) v -= a[i][p] * a[k][p];
          a[i][k] = v / d;
        }
      }
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) LT(k0 + i, k0 + j) = a[i][j];
    }
    __syncthreads();
    // panel: rows below the block solve L_ik L_kk^T = A_ik; the factor block and the row live in registers, so the six
    // columns of a row are not a chain of LDS round trips.  (Factorising the block redundantly in every thread to save
    // this barrier was measured and is slower: 1024 copies of the double-precision sqrt / divide sequences.)
    if (k0 + 6 + (int)threadIdx.x < n) {
      double a[6][6];
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) a[i][j] = LT(k0 + i, k0 + j);
      for (int i = k0 + 6 + threadIdx.x; i < n; i += BA_SOLVE_THREADS) {
        double r[6];
#pragma unroll
        for (int k = 0; k < 6; k++) r[k] = LT(i, k0 + k);
#pragma unroll
        for (int k = 0; k < 6; k++) {
          double v = r[k];
#pragma unroll
          for (int p = 0; p < k; p++) v -= r[p] * a[k][p];
          r[k] = v / a[k][k];
        }
#pragma unroll
        for (int k = 0; k < 6; k++) LT(i, k0 + k) = r[k];
      }
    }
    __syncthreads();
    // trailing update of the lower triangle: A_ij -= L_i,kb L_j,kb^T for j <= i, both below the block.  2 x 2 entries
    // per thread (24 LDS reads for 24 multiply-adds instead of 48); every entry's sum keeps the order p = 0..5.
    const int m = n - (k0 + 6), T = (m + 1) >> 1;
    for (int idx = threadIdx.x; idx < T * T; idx += BA_SOLVE_THREADS) {
      const int ti = idx / T, tj = idx - ti * T;
      if (tj > ti) continue;
      const int i0 = k0 + 6 + 2 * ti, j0 = k0 + 6 + 2 * tj;
      const bool i1v = i0 + 1 < n, j1v = j0 + 1 < n;
      double li0[6], li1[6], lj0[6], lj1[6];
#pragma unroll
      for (int p = 0; p < 6; p++) {
        li0[p] = LT(i0, k0 + p);
        lj0[p] = LT(j0, k0 + p);
        li1[p] = i1v ? LT(i0 + 1, k0 + p) : 0.0;
        lj1[p] = j1v ? LT(j0 + 1, k0 + p) : 0.0;
      }
      double v00 = 0.0, v01 = 0.0, v10 = 0.0, v11 = 0.0;
#pragma unroll
      for (int p = 0; p < 6; p++) {
        v00 += li0[p] * lj0[p];
        v01 += li0[p] * lj1[p];
        v10 += li1[p] * lj0[p];
        v11 += li1[p] * lj1[p];
      }
      LT(i0, j0) -= v00;                                   // j0 <= i0 always
      if (j1v && j0 + 1 <= i0) LT(i0, j0 + 1) -= v01;       // above the diagonal in a diagonal tile
      if (i1v) LT(i0 + 1, j0) -= v10;
      if (i1v && j1v) LT(i0 + 1, j0 + 1) -= v11;
    }
    __syncthreads();
  }
  // L y = b, L^T x = y by ONE wave, column by column: lane l holds entries l, l + 64 and l + 128 of the right-hand side
  // (n <= 192), the pivot entry is broadcast with v_readlane and every lane updates its own entries — 2 n dependent
  // steps of ~200 cycles instead of n^2 serial operations of one thread.  Forward: the subtractions reach every entry
  // in the order of the row-by-row loop (ascending column), so L y = b is bit-identical to it.
  if (threadIdx.x < kWave && !bad) {
    const int l = threadIdx.x;
    double yv[3];
#pragma unroll
    for (int q = 0; q < 3; q++) yv[q] = l + q * kWave < n ? y[l + q * kWave] : 0.0;
    auto bcast = [](double v, int src) {
      const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
      const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src);
      return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    };
    for (int i = 0; i < n; i++) {  // forward
      const int qi = i >> 6, li = i & (kWave - 1);  // wave-uniform
      const double piv = qi == 0 ? yv[0] : qi == 1 ? yv[1] : yv[2];
      const double yi = bcast(piv, li) / LT(i, i);
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const int r = l + q * kWave;
        if (q == qi && l == li) yv[q] = yi;
        if (r > i && r < n) yv[q] -= LT(r, i) * yi;
      }
    }
    for (int i = n - 1; i >= 0; i--) {  // backward: row i of L is column i of L^T
      const int qi = i >> 6, li = i & (kWave - 1);
      const double piv = qi == 0 ? yv[0] : qi == 1 ? yv[1] : yv[2];
      const double xi = bcast(piv, li) / LT(i, i);
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const int r = l + q * kWave;
        if (q == qi && l == li) yv[q] = xi;
        if (r < i) yv[q] -= LT(i, r) * xi;
      }
    }
#pragma unroll
    for (int q = 0; q < 3; q++)
      if (l + q * kWave < n) y[l + q * kWave] = yv[q];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += BA_SOLVE_THREADS) x[i] = bad ? 0.0f : (float)y[i];
}
#undef LT

}  // namespace lgu

extern "C" {

/* Pixel slices per edge of lgu_ba_build_f32 (the caller sizes `scratch` = E * slices * 90 floats with it). */
int lgu_ba_build_slices(int E) {
  if (E <= 0) return 1;
  const int s = (512 + E - 1) / E;
  return s < 1 ? 1 : (s > 8 ? 8 : s);
}


int lgu_ba_solve_f64(const double* A, const double* b, float* x, int P, double lm, double ep, void* stream) {
  using namespace lgu;
  if (!A || !b || !x || P < 1) return LGU_E_BADARG;
  const int n = 6 * P;
  if (n > BA_SOLVE_MAXN) return LGU_E_UNSUPPORTED;  // larger systems: the caller uses a library factorisation
  const size_t lds = sizeof(double) * ((((size_t)n * (n + 1)) >> 1) + n + 1);
  allow_max_dynamic_lds<&ba_solve_kernel>();
  hipLaunchKernelGGL(ba_solve_kernel, dim3(1), dim3(BA_SOLVE_THREADS), lds, reinterpret_cast<hipStream_t>(stream), A, b, x, n, lm, ep);
  return launch_status();
}

int lgu_ba_build_f32(const float* targets, const float* weights, const float* poses, const float* disps,
                     const float* intrinsics, const long long* ii, const long long* jj, float* Hs, float* vs, float* Eii,
                     float* Eij, float* Cii, float* wi, float* scratch, int E, int ht, int wd, void* stream) {
  using namespace lgu;
  if (!targets || !weights || !poses || !disps || !intrinsics || !ii || !jj || !Hs || !vs || !Eii || !Eij || !Cii || !wi || !scratch)
    return LGU_E_BADARG;
  if (E < 0 || ht < 1 || wd < 1) return LGU_E_BADARG;
  if (E == 0) return LGU_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int slices = lgu_ba_build_slices(E);
  hipLaunchKernelGGL(ba_build_kernel, dim3(E, slices), dim3(BA_THREADS), 0, st, targets, weights, poses, disps, intrinsics, ii, jj,
                     scratch, Eii, Eij, Cii, wi, E, ht * wd, wd, slices);
  hipLaunchKernelGGL(ba_build_finalize_kernel, dim3((E * 90 + BA_THREADS - 1) / BA_THREADS), dim3(BA_THREADS), 0, st, scratch, Hs, vs, E,
                     slices);
  return launch_status();
}

int lgu_ba_accum_f32(const float* inp, const long long* ptrs, const long long* idxs, float* out, int nout, int D, void* stream) {
  using namespace lgu;
  if (!inp || !ptrs || !out || nout < 0 || D < 1) return LGU_E_BADARG;
  if (nout == 0) return LGU_OK;
  hipLaunchKernelGGL(ba_accum_kernel, dim3(nout, (D + BA_THREADS * 4 - 1) / (BA_THREADS * 4)), dim3(BA_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), inp, ptrs, idxs, out, D);
  return launch_status();
}

int lgu_ba_depth_system_f32(const float* Cii, const float* wi, const long long* ptrs, const long long* idxs, const long long* kx,
                            const float* disps, const float* disps_sens, const float* eta, int eta_rows, float* Q, float* w, int K,
                            int HW, void* stream) {
  using namespace lgu;
  if (!Cii || !wi || !ptrs || !idxs || !kx || !disps || !disps_sens || !eta || !Q || !w || K < 0 || HW < 1) return LGU_E_BADARG;
  if (eta_rows != 1 && eta_rows != K) return LGU_E_BADARG;
  if (K == 0) return LGU_OK;
  hipLaunchKernelGGL(ba_depth_system_kernel, dim3(K, (HW + BA_THREADS * 4 - 1) / (BA_THREADS * 4)), dim3(BA_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), Cii, wi, ptrs, idxs, kx, disps, disps_sens, eta, eta_rows, 0.05f, Q, w, HW);
  return launch_status();
}

int lgu_ba_depth_update_f32(const float* Q, const float* w, const float* dw, const long long* ptrs, const long long* idxs,
                            const long long* kx, float* dz, float* disps, int K, int HW, void* stream) {
  using namespace lgu;
  if (!Q || !w || !dw || !ptrs || !idxs || !kx || !dz || !disps || K < 0 || HW < 1) return LGU_E_BADARG;
  if (K == 0) return LGU_OK;
  hipLaunchKernelGGL(ba_depth_update_kernel, dim3(K, (HW + BA_THREADS * 4 - 1) / (BA_THREADS * 4)), dim3(BA_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), Q, w, dw, ptrs, idxs, kx, dz, disps, HW);
  return launch_status();
}

int lgu_ba_scatter_sum_f64(const float* inp, const long long* ptrs, const long long* idxs, const long long* dst, double* out, int m,
                           int D, double sign, void* stream) {
  using namespace lgu;
  if (!inp || !ptrs || !idxs || !dst || !out || m < 0 || D < 1) return LGU_E_BADARG;
  if (m == 0) return LGU_OK;
  const unsigned grid = (unsigned)(((size_t)m * D + BA_THREADS - 1) / BA_THREADS);
  hipLaunchKernelGGL(ba_scatter_sum_kernel, dim3(grid), dim3(BA_THREADS), 0, reinterpret_cast<hipStream_t>(stream), inp, ptrs, idxs,
                     dst, out, m, D, sign);
  return launch_status();
}

int lgu_ba_assemble_f64(const float* Hs, const long long* hptr, const long long* hidx, const float* S, const long long* sptr,
                        const long long* sidx, const float* vs, const long long* vptr, const long long* vidx, const float* sv,
                        const long long* svptr, const long long* svidx, double* Ad, double* b, int P, void* stream) {
  using namespace lgu;
  if (!Hs || !hptr || !hidx || !vs || !vptr || !vidx || !Ad || !b || P < 1) return LGU_E_BADARG;
  if ((S && (!sptr || !sidx)) || (sv && (!svptr || !svidx))) return LGU_E_BADARG;
  const long long n = (long long)P * P * 36 + (long long)P * 6;
  hipLaunchKernelGGL(ba_assemble_kernel, dim3((unsigned)((n + BA_THREADS - 1) / BA_THREADS)), dim3(BA_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), Hs, hptr, hidx, S, sptr, sidx, vs, vptr, vidx, sv, svptr, svidx, Ad, b, P);
  return launch_status();
}

int lgu_ba_eet_f32(const float* Em, const float* Q, const long long* idx, float* S, int nblocks, int D, void* stream) {
  using namespace lgu;
  if (!Em || !Q || !S || nblocks < 0 || D < 1) return LGU_E_BADARG;
  if (nblocks == 0) return LGU_OK;
  if (!idx) return LGU_E_BADARG;
  hipLaunchKernelGGL(ba_eet_kernel, dim3(nblocks), dim3(BA_THREADS), 0, reinterpret_cast<hipStream_t>(stream), Em, Q, idx, S, D);
  return launch_status();
}

int lgu_ba_ev_f32(const float* Em, const float* Q, const float* w, const long long* kk, float* v, int n, int D, void* stream) {
  using namespace lgu;
  if (!Em || !Q || !w || !kk || !v || n < 0 || D < 1) return LGU_E_BADARG;
  if (n == 0) return LGU_OK;
  hipLaunchKernelGGL(ba_ev_kernel, dim3(n), dim3(BA_THREADS), 0, reinterpret_cast<hipStream_t>(stream), Em, Q, w, kk, v, D);
  return launch_status();
}

int lgu_ba_evt_f32(const float* Em, const float* x, const long long* idx, float* dw, int n, int D, int P, void* stream) {
  using namespace lgu;
  if (!Em || !x || !idx || !dw || n < 0 || D < 1 || P < 1) return LGU_E_BADARG;
  if (n == 0) return LGU_OK;
  hipLaunchKernelGGL(ba_evt_kernel, dim3(n), dim3(BA_THREADS), 0, reinterpret_cast<hipStream_t>(stream), Em, x, idx, dw, D, P);
  return launch_status();
}

int lgu_ba_pose_retr_f32(float* poses, const float* dx, int t0, int t1, void* stream) {
  using namespace lgu;
  if (!poses || !dx || t0 < 0 || t1 < t0) return LGU_E_BADARG;
  if (t1 == t0) return LGU_OK;
  hipLaunchKernelGGL(ba_pose_retr_kernel, dim3((t1 - t0 + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), poses, dx,
                     t0, t1);
  return launch_status();
}

int lgu_ba_disp_retr_f32(float* disps, const float* dz, const long long* inds, int n, int HW, void* stream) {
  using namespace lgu;
  if (!disps || !dz || !inds || n < 0 || HW < 1) return LGU_E_BADARG;
  if (n == 0) return LGU_OK;
  hipLaunchKernelGGL(ba_disp_retr_kernel, dim3(n), dim3(BA_THREADS), 0, reinterpret_cast<hipStream_t>(stream), disps, dz, inds, HW);
  return launch_status();
}

}  // extern "C"
