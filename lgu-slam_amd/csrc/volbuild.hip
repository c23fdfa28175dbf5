// volbuild.hip — the all-pairs correlation volume of CorrBlock.__init__ built ON the matrix cores straight into the tiled
// 4-level pyramid: no raw volume ever reaches HBM.
//
// Replaces, per new edge (reference, relative to /root/reference):
//   droid_slam/modules/corr.py:145-152      CorrBlock.corr: matmul((fmap1/4)^T, fmap2/4)            -> (h*w, h, w) volume
//   droid_slam/modules/corr.py:64           .float()
//   droid_slam/gaussianMask_cuda.py:84-86   gaussianMask kernel, corr1 / (6.28*sqrt(det)) + corr
//   droid_slam/modules/corr.py:79-86        3 x avg_pool2d over the target dims
// The library GEMM + csrc/gaussmask.hip's fused builder (lgu_volume_pyramid_*) write the raw volume (37.7 MB per edge at
// 48 x 64) and read it back once; here the product tile goes from the MFMA accumulators through LDS into the pyramid:
// HBM sees the feature maps (3 MB per edge, L2-resident) and ONE write of the pyramid (50 MB per edge).
//
// Decomposition.  Workgroup = 4 waves = 32 source pixels x one STRIP of 8 target rows (8 W positions): a strip holds whole
// 8 x 8 pooling blocks, so all four levels of it are formed in the workgroup, and in the tiled slice layout (4 x 8 element
// tiles, tile rows consecutive) a strip is ONE contiguous run of a slice at every level (2 KB / 512 B / 2 x 64 B / 32 B at
// W = 64): the stores are whole lines.  Wave w owns the strip's columns [w * 2W, (w + 1) * 2W) = NTW tiles of 32 positions;
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation: the GEMM the reference runs in fp32), A = 32 source
// pixels, B = 32 target positions, K = the channels two at a time; the (L2-resident) NCHW maps arrive 16 channels at a
// time through LDS, fetched with 16-byte loads under the previous chunk's MFMAs.  Epilogue in two halves of 16 source pixels (LDS: 16 x (8W + 4) floats of level 0 + the
// pooled levels = 44 KB at W = 64, three workgroups per CU): accumulators -> LDS, / 16 (both maps carry the reference's
// / 4: exact), Gaussian re-weighting in gaussmask.hip's arithmetic (t = (v*3*e)/den + v inside the 9 x 9 window, v
// outside), level 0 out in tiled order, then the pooled levels in ATen's order ((a00 + a01) + a10 + a11) / 4, each from the
// level before it.
// Results differ from the library GEMM + fused builder by the GEMM's summation order only (tests: 1e-5 of the scale).
#include "lgu_common.hpp"

namespace lgu {

typedef float vbf32x16 __attribute__((ext_vector_type(16)));
typedef float vbf32x4 __attribute__((ext_vector_type(4)));   // (HIP's float4 struct in a loop-carried array stays on the stack)
typedef _Float16 vbh8 __attribute__((ext_vector_type(8)));

constexpr int VB_M = 32;        // source pixels per workgroup
constexpr int VB_ROWS = 8;      // target rows per strip
constexpr int VB_THREADS = 256;
constexpr int VB_L = 4;
constexpr int VB_KC = 16;       // channels per chunk staged in LDS (fp32 maps)
constexpr int VB_EP = 16;       // source pixels per epilogue pass (8 or 16): sets the LDS footprint
// (measured, profiles/r03_ab_volbuild_half.txt: passes of 8 pixels + a register cap for 4 or 5 workgroups per CU make both
// kernels SLOWER — 0.50 / 0.97 ms for the half kernel against 0.38 — because under the cap the compiler gives up the operand
// prefetch: load, wait, MFMA.  Deeper chunks (4 or 8 k steps in flight) move the half kernel by -8 .. +6 % only.  A launch
// takes the SUM of a product-only and an epilogue-only launch (half: 0.126 + 0.225 ms at 20 edges): the phases of the workgroups
// sharing a CU do not overlap, and neither halving the operand traffic (two source blocks per workgroup) nor staggering the
// workgroups' phases changes that — what is left is overlap inside a wave.)
constexpr int VB_KH = 32;       // channels per register chunk (half maps): two v_mfma_f32_32x32x16_f16 per tile

struct VolBuildParams {
  const float* f1;      // (E, C, H*W) source maps            (fp32 kernel)
  const float* f2;      // (E, C, H*W) target maps
  const _Float16* th;   // both maps packed in MFMA fragment order by volume_pack_kernel (half kernel)
  const float* means;   // (E*H*W, 2)
  const float* covs;    // (E*H*W, 2)
  const void* det;      // (E*H*W) fp32 or half, or null (= cov0 * cov1)
  int det_half;
  float* out[VB_L];     // tiled levels
  int E, C, H, W, r;
};

__host__ __device__ constexpr int vb_pitch(int W) { return VB_ROWS * W + 4; }
// LDS floats: the larger of one epilogue pass (VB_EP source pixels: level 0 strip + levels 1..3 of it) and, for fp32 maps, one
// operand chunk
__host__ __device__ constexpr int vb_lds_floats(int W, bool half) {
  const int ep = VB_EP * (vb_pitch(W) + (VB_ROWS / 2) * (W / 2) + (VB_ROWS / 4) * (W / 4) + (VB_ROWS / 8) * (W / 8));
  const int chunk = half ? 0 : VB_KC * (VB_ROWS * W + VB_M);
  return ep > chunk ? ep : chunk;
}

// exp(f1) of gaussianAttn.cu:59-62 (see gaussmask.hip)
__device__ __forceinline__ float vb_gauss_e(int x1, int y1, float mx, float my, float c1, float c2) {
  const float ddx = (float)x1 - mx, ddy = (float)y1 - my;
  const float temp1 = ddx / c1, temp2 = ddy / c2;
  const float f1 = -0.5f * (temp1 * ddx + temp2 * ddy);
  return expf(f1);
}

template <int NTW, bool HALF>   // 32-position tiles per wave: W = 16 * NTW; HALF: half maps, product rounded to half
__global__ __launch_bounds__(VB_THREADS, 2) void volume_build_kernel(const VolBuildParams p) {
  constexpr int W = 16 * NTW, N = VB_ROWS * W, PITCH = vb_pitch(W);
  constexpr int W1 = W / 2, W2 = W / 4, W3 = W / 8;            // level widths of the strip
  constexpr int N1 = (VB_ROWS / 2) * W1, N2 = (VB_ROWS / 4) * W2, N3 = (VB_ROWS / 8) * W3;
  extern __shared__ float4 vb_smem4[];
  float* const st0 = reinterpret_cast<float*>(vb_smem4);      // [VB_EP][PITCH]  level-0 strip, row-major (y, x)
  float* const st1 = st0 + VB_EP * PITCH;                      // [VB_EP][N1]
  float* const st2 = st1 + VB_EP * N1;                         // [VB_EP][N2]
  float* const st3 = st2 + VB_EP * N2;                         // [VB_EP][N3]

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int HW = p.H * p.W;
  const int mblocks = HW / VB_M, strips = p.H / VB_ROWS;
  // source-pixel block fastest: consecutive workgroups share the strip of the target map (256 KB at W = 64, K = 128)
  unsigned bid = blockIdx.x;
  const int mb = (int)(bid % (unsigned)mblocks);
  bid /= (unsigned)mblocks;
  const int s = (int)(bid % (unsigned)strips);
  const int e = (int)(bid / (unsigned)strips);
  const size_t mapbase = (size_t)e * p.C * HW;

  // ---- product: acc[t][r] = sum_c f1[c][p1] * f2[c][p2],  p1 = mb*32 + row(r, lane),  p2 = s*8W + wv*2W + t*32 + (lane & 31)
  // The channels arrive VB_KC at a time: every thread fetches its share of the next chunk of both maps with 16-byte loads
  // (rows of the NCHW maps are contiguous in the positions) into registers WHILE the matrix cores work on the chunk that is in
  // LDS; operands are then single ds_read_b32s (lane = position, the k slot selects the row).  The chunk buffers share LDS
  // with the epilogue's staging area (used after the last chunk).
  vbf32x16 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
  if constexpr (HALF) {
    // half maps (autocast: the reference's matmul is a half GEMM — exact half x half products, fp32 accumulation, ONE
    // rounding of the sum to half, which the epilogue applies).  The maps arrive PACKED in MFMA fragment order
    // (volume_pack_kernel below): [map][tile of 32 positions][k step of 16 channels][lane][8 halves], lane (h, li) holding
    // channels 16 s + 8 h + j of position 32 tile + li — the operand of v_mfma_f32_32x32x16_f16 for that tile and step is ONE
    // contiguous KB, 16 bytes per lane, straight into the fragment registers.  (Read channel-last, where a lane's 16 bytes
    // sit 4 C bytes from its neighbour's, every load touched 64 lines and the texture path, at a line per clock, took as
    // long as the whole epilogue: 0.21 of 0.42 ms at 20 edges.)  Per 32-channel chunk: 2 loads per tile, the next chunk's
    // travelling under this chunk's MFMAs.  The maps are L2-resident (0.75 MB per edge).
    const int S = p.C >> 4, T = HW / 32;
    const vbh8* const fa = reinterpret_cast<const vbh8*>(p.th);
    // (16-byte units; the host checks that the packed maps hold fewer than 2^32 of them)
    const unsigned abase = ((unsigned)(e * 2 + 0) * T + mb) * S * 64 + lane;
    unsigned bbase[NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++) bbase[t] = ((unsigned)(e * 2 + 1) * T + s * (N / 32) + wv * NTW + t) * S * 64 + lane;
    vbh8 a[2], b[NTW][2], na[2], nb[NTW][2];
#define VB_FETCH_H(s0)                                                                          \
  {                                                                                             \
    na[0] = fa[abase + (unsigned)(s0) * 64u];                                                   \
    na[1] = fa[abase + (unsigned)((s0) + 1) * 64u];                                             \
    _Pragma("unroll") for (int t = 0; t < NTW; t++) {                                           \
      nb[t][0] = fa[bbase[t] + (unsigned)(s0) * 64u];                                           \
      nb[t][1] = fa[bbase[t] + (unsigned)((s0) + 1) * 64u];                                     \
    }                                                                                           \
  }
    VB_FETCH_H(0)
    for (int s0 = 0; s0 < S; s0 += 2) {
      a[0] = na[0]; a[1] = na[1];
#pragma unroll
      for (int t = 0; t < NTW; t++) { b[t][0] = nb[t][0]; b[t][1] = nb[t][1]; }
      {
        const int sn = s0 + 2 < S ? s0 + 2 : s0;                 // (after the last chunk: the same one again, unused)
        VB_FETCH_H(sn)
      }
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int t = 0; t < NTW; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], b[t][ks], acc[t], 0, 0, 0);
    }
#undef VB_FETCH_H
  } else {
    float* const Bs = reinterpret_cast<float*>(vb_smem4);       // [VB_KC][N]
    float* const As = Bs + VB_KC * N;                            // [VB_KC][VB_M]
    constexpr int BQ = VB_KC * N / 4 / VB_THREADS;               // float4s of the target chunk per thread (8 at W = 64)
    static_assert(VB_KC * N / 4 % VB_THREADS == 0 && VB_KC * VB_M / 4 <= VB_THREADS, "chunk shares");
    const float* const f1e = p.f1 + mapbase + mb * VB_M;
    const float* const f2e = p.f2 + mapbase + s * N;
    const int tid = threadIdx.x;
    vbf32x4 gb[BQ], ga;
    // this thread's share of a chunk: BQ float4s of the target rows, one of the source rows (threads 0 .. 127)
    int brow[BQ], bc4[BQ];
#pragma unroll
    for (int j = 0; j < BQ; j++) {
      const int idx = tid + VB_THREADS * j;
      brow[j] = idx / (N / 4);
      bc4[j] = (idx - brow[j] * (N / 4)) * 4;
    }
    const int arow = (tid & (VB_KC * VB_M / 4 - 1)) / (VB_M / 4), ac4 = ((tid & (VB_KC * VB_M / 4 - 1)) % (VB_M / 4)) * 4;  // (upper threads repeat the lower ones' loads: no branch around a load)
#define VB_FETCH(c0)                                                                                              \
  {                                                                                                               \
    _Pragma("unroll") for (int j = 0; j < BQ; j++)                                                                \
        gb[j] = *reinterpret_cast<const vbf32x4*>(f2e + (size_t)((c0) + brow[j]) * HW + bc4[j]);                   \
    ga = *reinterpret_cast<const vbf32x4*>(f1e + (size_t)((c0) + arow) * HW + ac4);                                \
  }
    const int kk = lane >> 5, li = lane & 31;                   // operand lane: k slot, row / column
    VB_FETCH(0)
    for (int c0 = 0; c0 < p.C; c0 += VB_KC) {
      __syncthreads();                                           // the chunk in LDS has been consumed
#pragma unroll
      for (int j = 0; j < BQ; j++) reinterpret_cast<vbf32x4*>(Bs)[tid + VB_THREADS * j] = gb[j];
      if (tid < VB_KC * VB_M / 4) reinterpret_cast<vbf32x4*>(As)[tid] = ga;
      __syncthreads();
      {
        const int cn = c0 + VB_KC < p.C ? c0 + VB_KC : c0;       // (after the last chunk: the same one again, unused)
        VB_FETCH(cn)                                             // travels while this chunk is multiplied
      }
#pragma unroll
      for (int ks = 0; ks < VB_KC / 2; ks++) {
        const float a = As[(2 * ks + kk) * VB_M + li];
        float b[NTW];
#pragma unroll
        for (int t = 0; t < NTW; t++) b[t] = Bs[(2 * ks + kk) * N + wv * (NTW * 32) + t * 32 + li];
#pragma unroll
        for (int t = 0; t < NTW; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
      }
    }
    __syncthreads();                                             // every wave is done with the chunk buffers: the epilogue reuses them
#undef VB_FETCH
  }

  // ---- epilogue, VB_EP source pixels per pass (accumulator registers 4q .. 4q+3 hold rows 8q .. 8q+7) ----
  constexpr int WK = VB_THREADS / VB_EP;                        // workers per pixel (16 or 32)
  constexpr int RP = VB_EP / 2;                                  // accumulator registers per pass
  const int tp = threadIdx.x / WK, ts = threadIdx.x % WK;       // epilogue thread: pixel of the pass, 1 of WK workers on it
  const int tpr[VB_L] = {p.W >> 3, ((p.W >> 1) + 7) >> 3, ((p.W >> 2) + 7) >> 3, ((p.W >> 3) + 7) >> 3};
#pragma unroll
  for (int half = 0; half < VB_M / VB_EP; half++) {
    if (half) __syncthreads();                                   // the previous pass's LDS reads are done
    // accumulators -> st0[row][col]:  row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5),  col = lane & 31
#pragma unroll
    for (int t = 0; t < NTW; t++)
#pragma unroll
      for (int r8 = 0; r8 < RP; r8++) {
        const int r = half * RP + r8;
        const int row = (r8 & 3) + 8 * (r8 >> 2) + 4 * (lane >> 5);      // 0 .. VB_EP-1 within the pass
        float v = acc[t][r] * 0.0625f;                           // (f1/4)(f2/4): exact scaling
        if constexpr (HALF) v = (float)(_Float16)v;              // the half GEMM's output rounding, then corr.py:64 .float()
        st0[row * PITCH + wv * (NTW * 32) + t * 32 + (lane & 31)] = v;
      }
    __syncthreads();

    // per-pixel Gaussian parameters (a pixel's workers read the same)
    const size_t pix = (size_t)e * HW + mb * VB_M + half * VB_EP + tp;
    const float mx = p.means[pix * 2 + 0], my = p.means[pix * 2 + 1];
    const float c1 = p.covs[pix * 2 + 0], c2 = p.covs[pix * 2 + 1];
    float den;
    if (p.det && p.det_half) {   // the reference's half roundings under autocast (gaussmask.hip, volume_pyramid_kernel)
      const _Float16 sq = (_Float16)sqrtf((float)static_cast<const _Float16*>(p.det)[pix]);
      float prod = (float)sq * 6.28f;
      asm volatile("" : "+v"(prod));
      den = (float)(_Float16)prod;
    } else {
      den = 6.28f * sqrtf(p.det ? static_cast<const float*>(p.det)[pix] : c1 * c2);
    }
    const int cx = (int)floorf(mx), cy = (int)floorf(my);
    const int xa = cx - p.r, xb = cx + p.r, ya = cy - p.r, yb = cy + p.r;
    float* const row0 = st0 + tp * PITCH;

    // level 0: re-weight in place, write the strip out in tiled order (one contiguous run of the slice)
    {
      const int ssz0 = ((p.H + 3) >> 2) * tpr[0] * 32;
      float4* const g0 = reinterpret_cast<float4*>(p.out[0] + pix * (size_t)ssz0 + (size_t)s * (2 * tpr[0] * 32));
#pragma unroll
      for (int j = 0; j < N / (4 * WK); j++) {
        const int t4 = (ts + WK * j) * 4;                        // tiled index of this float4 inside the strip
        const int tile = t4 >> 5, ty = tile / (W / 8), tx = tile - ty * (W / 8);
        const int yl = ty * 4 + ((t4 & 31) >> 3), x4 = tx * 8 + (t4 & 7);
        const int row = s * VB_ROWS + yl;                        // target row in the map
        float4 v = *reinterpret_cast<const float4*>(row0 + yl * W + x4);
        if (row >= ya && row <= yb && x4 + 3 >= xa && x4 <= xb) {
          if (x4 + 0 >= xa && x4 + 0 <= xb) v.x = (v.x * 3.0f * vb_gauss_e(x4 + 0, row, mx, my, c1, c2)) / den + v.x;
          if (x4 + 1 >= xa && x4 + 1 <= xb) v.y = (v.y * 3.0f * vb_gauss_e(x4 + 1, row, mx, my, c1, c2)) / den + v.y;
          if (x4 + 2 >= xa && x4 + 2 <= xb) v.z = (v.z * 3.0f * vb_gauss_e(x4 + 2, row, mx, my, c1, c2)) / den + v.z;
          if (x4 + 3 >= xa && x4 + 3 <= xb) v.w = (v.w * 3.0f * vb_gauss_e(x4 + 3, row, mx, my, c1, c2)) / den + v.w;
          *reinterpret_cast<float4*>(row0 + yl * W + x4) = v;
        }
        g0[ts + WK * j] = v;
      }
    }
    __syncthreads();
    // level 1 (4 x W/2 of the strip = one tile row of the level-1 slice), from level 0
    {
      const int ssz1 = (((p.H >> 1) + 3) >> 2) * tpr[1] * 32;
      float* const g1 = p.out[1] + pix * (size_t)ssz1 + (size_t)s * (tpr[1] * 32);
      float* const d1 = st1 + tp * N1;
      for (int t = ts; t < tpr[1] * 32; t += WK) {              // tiled order inside the tile row (tiles may be padded in x)
        const int tile = t >> 5, y = (t & 31) >> 3, x = tile * 8 + (t & 7);
        float o = 0.0f;
        if (x < W1) {
          const float* q = row0 + (2 * y) * W + 2 * x;
          o = (((q[0] + q[1]) + q[W]) + q[W + 1]) / 4.0f;
          d1[y * W1 + x] = o;
        }
        g1[t] = o;
      }
    }
    __syncthreads();
    // level 2 (2 x W/4: rows (s & 1) * 2 .. + 1 of tile row s / 2), from level 1; the slice's padding — columns up to the
    // tile width, and the rows below the map in its last tile row — is written as zeros by the strip next to it, as the
    // fused builder of gaussmask.hip writes it (the lookup's clamped loads may touch it)
    {
      const int ssz2 = (((p.H >> 2) + 3) >> 2) * tpr[2] * 32;
      float* const g2 = p.out[2] + pix * (size_t)ssz2 + (size_t)(s >> 1) * (tpr[2] * 32);
      const float* const s1 = st1 + tp * N1;
      float* const d2 = st2 + tp * N2;
      const int r0 = (s & 1) * 2, nr = (s == strips - 1) ? 4 - r0 : 2;   // rows of the tile row this strip writes
      for (int t = ts; t < nr * tpr[2] * 8; t += WK) {
        const int y = t / (tpr[2] * 8), xx = t - y * (tpr[2] * 8);          // row (relative to r0), padded column
        float o = 0.0f;
        if (y < 2 && xx < W2) {
          const float* q = s1 + (2 * y) * W1 + 2 * xx;
          o = (((q[0] + q[1]) + q[W1]) + q[W1 + 1]) / 4.0f;
          d2[y * W2 + xx] = o;
        }
        g2[((xx >> 3) << 5) + ((r0 + y) << 3) + (xx & 7)] = o;
      }
    }
    __syncthreads();
    // level 3 (1 x W/8: row s & 3 of tile row s / 4), from level 2; padding as above
    {
      const int ssz3 = (((p.H >> 3) + 3) >> 2) * tpr[3] * 32;
      float* const g3 = p.out[3] + pix * (size_t)ssz3 + (size_t)(s >> 2) * (tpr[3] * 32);
      const float* const s2 = st2 + tp * N2;
      const int r0 = s & 3, nr = (s == strips - 1) ? 4 - r0 : 1;
      for (int t = ts; t < nr * tpr[3] * 8; t += WK) {
        const int y = t / (tpr[3] * 8), xx = t - y * (tpr[3] * 8);
        float o = 0.0f;
        if (y < 1 && xx < W3) {
          const float* q = s2 + 2 * xx;
          o = (((q[0] + q[1]) + q[W2]) + q[W2 + 1]) / 4.0f;
        }
        g3[((xx >> 3) << 5) + ((r0 + y) << 3) + (xx & 7)] = o;
      }
    }
    (void)st3; (void)N3;
  }
}

// (E, H*W, 2C) half, channel-last (source channels first) -> [e][map][tile][k step][lane = 32 h + li][8]: one thread per 16 bytes
__global__ __launch_bounds__(256) void volume_pack_kernel(const _Float16* __restrict__ feats, _Float16* __restrict__ packed,
                                                          int C, int HW, unsigned total) {
  const unsigned o = blockIdx.x * 256u + threadIdx.x;
  if (o >= total) return;
  const int S = C >> 4, T = HW / 32;
  const unsigned lane = o & 63u;
  unsigned q = o >> 6;
  const unsigned st = q % (unsigned)S;  q /= (unsigned)S;
  const unsigned tile = q % (unsigned)T;  q /= (unsigned)T;
  const unsigned map = q & 1u, e = q >> 1;
  const unsigned h = lane >> 5, li = lane & 31u;
  const size_t src = ((size_t)e * HW + tile * 32u + li) * (size_t)(2 * C) + map * (unsigned)C + 16u * st + 8u * h;
  reinterpret_cast<vbh8*>(packed)[o] = *reinterpret_cast<const vbh8*>(feats + src);
}

template <int NTW, bool HALF>
static int launch_volume_build(const VolBuildParams& p, hipStream_t st) {
  const size_t lds = sizeof(float) * (size_t)vb_lds_floats(16 * NTW, HALF);
  if (lds > 64 * 1024) allow_max_dynamic_lds<&volume_build_kernel<NTW, HALF>>();
  const size_t grid = (size_t)p.E * (p.H / VB_ROWS) * ((size_t)p.H * p.W / VB_M);
  if (grid >= (1ull << 31)) return LGU_E_UNSUPPORTED;
  hipLaunchKernelGGL((volume_build_kernel<NTW, HALF>), dim3((unsigned)grid), dim3(VB_THREADS), lds, st, p);
  return launch_status();
}

// argument checks both entries share; UNSUPPORTED: whole strips of 8 target rows, whole blocks of 32 source pixels, 2 W
// positions per wave in 32-position MFMA tiles, whole channel chunks, 16-byte loads and stores
static int volume_build_checks(const void* m1, const void* m2, const float* means, const float* covs, float* const* levels, int L,
                               int E, int C, int H, int W, int radius, int kchunk) {
  if (!m1 || !m2 || !means || !covs || !levels || E < 0 || C < 1 || H < 1 || W < 1 || radius < 0) return LGU_E_BADARG;
  if (L != VB_L) return LGU_E_UNSUPPORTED;
  for (int l = 0; l < L; l++)
    if (!levels[l]) return LGU_E_BADARG;
  if (H % VB_ROWS != 0 || (W != 16 && W != 32 && W != 64) || (H * W) % VB_M != 0 || C % kchunk != 0) return LGU_E_UNSUPPORTED;
  if ((size_t)C * H * W >= (1u << 30)) return LGU_E_UNSUPPORTED;
  uintptr_t al = reinterpret_cast<uintptr_t>(m1) | reinterpret_cast<uintptr_t>(m2);
  for (int l = 0; l < L; l++) al |= reinterpret_cast<uintptr_t>(levels[l]);
  return (al & 15) ? LGU_E_UNSUPPORTED : LGU_OK;
}

}  // namespace lgu

extern "C" {

int lgu_volume_build_pyramid_f32(const float* fmap1, const float* fmap2, const float* means, const float* covs, const void* det,
                                 int det_half, float* const* levels, int L, int E, int C, int H, int W, int radius, void* stream) {
  using namespace lgu;
  const int rc = volume_build_checks(fmap1, fmap2, means, covs, levels, L, E, C, H, W, radius, VB_KC);
  if (rc != LGU_OK) return rc;
  if (E == 0) return LGU_OK;
  VolBuildParams p;
  p.f1 = fmap1; p.f2 = fmap2; p.th = nullptr; p.means = means; p.covs = covs; p.det = det; p.det_half = det_half ? 1 : 0;
  for (int l = 0; l < VB_L; l++) p.out[l] = levels[l];
  p.E = E; p.C = C; p.H = H; p.W = W; p.r = radius;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (W) {
    case 16: return launch_volume_build<1, false>(p, st);
    case 32: return launch_volume_build<2, false>(p, st);
    default: return launch_volume_build<4, false>(p, st);
  }
}

int lgu_volume_build_pyramid_h16(const void* feats, void* workspace, const float* means, const float* covs, const void* det,
                                 int det_half, float* const* levels, int L, int E, int C, int H, int W, int radius, void* stream) {
  using namespace lgu;
  const int rc = volume_build_checks(feats, workspace, means, covs, levels, L, E, C, H, W, radius, VB_KH);
  if (rc != LGU_OK) return rc;
  if (E == 0) return LGU_OK;
  const size_t units = (size_t)E * H * W * 2 * C / 8;             // 16-byte pieces of the maps
  if (units >= (1ull << 32)) return LGU_E_UNSUPPORTED;
  hipLaunchKernelGGL(volume_pack_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     static_cast<const _Float16*>(feats), static_cast<_Float16*>(workspace), C, H * W, (unsigned)units);
  VolBuildParams p;
  p.f1 = nullptr; p.f2 = nullptr; p.th = static_cast<const _Float16*>(workspace);
  p.means = means; p.covs = covs; p.det = det; p.det_half = det_half ? 1 : 0;
  for (int l = 0; l < VB_L; l++) p.out[l] = levels[l];
  p.E = E; p.C = C; p.H = H; p.W = W; p.r = radius;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (W) {
    case 16: return launch_volume_build<1, true>(p, st);
    case 32: return launch_volume_build<2, true>(p, st);
    default: return launch_volume_build<4, true>(p, st);
  }
}

}  // extern "C"
