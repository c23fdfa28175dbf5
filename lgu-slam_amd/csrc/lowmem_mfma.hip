// lowmem_mfma.hip — on-the-fly deformable correlation from HALF feature maps on the matrix cores.
//
// Serves the mixed-precision entry points lgu_lowmem_defsample_fwd_h16 / lgu_altcorr_fwd_h16, i.e. the
// reference call sites lowMem_defSample(fmap1.float(), fmap2.float(), ...) (droid_slam/modules/corr.py:209)
// and altcorr_forward(fmap1.float(), fmap2.float(), ...) (:202) for feature maps the SLAM system stores
// in half precision.  Kernels replaced: offersample_LGS/lowMem_defSample.cu:27-134,
// src/altcorr_kernel.cu:27-149.
//
// Unlike the volume path this operator IS a contraction: dot(fmap1[p], bilerp(fmap2)(x,y)) equals the
// bilinear blend of the four corner dots, so a pixel's 49 taps only need its LOCAL correlation patch
//     D_p[y2][x2] = sum_c fmap1[p][c] * fmap2[y2][x2][c]     over its tap box (<= 16x16 positions)
// and neighbouring pixels' boxes overlap almost entirely.  One WAVE owns a 4x4 pixel block:
//   0. lanes = taps: tap box of each of the 16 pixels (packed DPP min/max), their union = the block window;
//   1. the window is swept in groups of 16 x-adjacent positions: v_mfma_f32_16x16x32_f16 with
//      A = fmap1 of the 16 pixels (resident in registers for the whole sweep, K = C channels in C/32 steps),
//      B = fmap2 of the 16 positions, each lane loading 16 contiguous bytes of channels straight from the
//      channel-last map (both operands are "k-contiguous per lane" in this layout: no LDS staging, no
//      transposition).  Half products are exact in fp32 and the accumulation is fp32, so the result differs
//      from the reference's `.float()` + fp32 chain by summation order only (tests: 1e-5).
//      Each 16x16 result tile is scattered into the pixels' patches in LDS (only entries inside a pixel's box);
//   2. lanes = taps again: the four corners come from the patch with the reference's per-corner zero
//      padding (lowMem_defSample.cu:102-117), blended in its evaluation order, transposed through LDS
//      and written as 16-byte row segments.
// Every wave is independent (no workgroup barrier anywhere); a workgroup is 4 waves = an 8x8 pixel tile
// so that neighbouring windows share L1/L2 lines, and edges are dealt to XCDs (workgroup id % 8) so one
// edge's fmap2 (a few MB) stays in one XCD's L2.
// Pixels whose box exceeds 16x16 (|offset| >= 4: never produced by corr.py:126-131) take a per-tap fallback.
#include "lgu_common.hpp"

namespace lgu {

typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int MM_BP = 16;            // pixels per wave = 4 x 4 block (MFMA rows)
constexpr int MM_BOX = 16;           // patch row pitch / largest box side
constexpr int MM_PP = MM_BOX * MM_BOX + 4;  // patch pitch per pixel: the 4 pixel groups of one scatter land on different bank quarters
constexpr int MM_MAXMT = 2;          // 4 x 4 pixel sub-blocks (MFMA row tiles) one wave can own
__host__ __device__ constexpr int mm_outp(int MT) { return 16 * MT + 4; }  // output transpose pitch (pixels + pad, 16-byte aligned rows)
// Zero-offset levels (ZO kernels): every tap box is the (2R+2)^2 <= 8 x 8 integer lattice around the level coords, so a
// patch is 8 x 8 (+ 4 pad): 4.6 KB per 16 pixels instead of 16.9 KB, and no offset is loaded or held.
constexpr int MM_GUARD = 20;         // floats in front of the patches (>= MM_BOX + 1): see the sampling phase
constexpr int MM_ZBOX = 8;
constexpr int MM_ZPP = MM_ZBOX * MM_ZBOX + 4;
__host__ __device__ constexpr int mm_lds_floats(int MT, bool ZO = false) {
  return MM_GUARD + MT * (MM_BP * (ZO ? MM_ZPP : MM_PP) + MM_BP * 4);
}

// Element-type traits.  One k-step covers CPS channels; lane (lg, lx) holds EPL consecutive channels starting at
// EPL * lg of row/column lx.  half: one v_mfma_f32_16x16x32_f16 per step.  float: four v_mfma_f32_16x16x4_f32 per
// step (component t of the lane's float4 is k-slot lg of MFMA t) — exact fp32, a k-ordered fmaf chain per the ISA,
// at the fp32 matrix rate (= the packed-VALU peak, but without the per-product LDS reads and VALU issue of the
// tile kernel).
template <typename T> struct MmT;
template <> struct MmT<_Float16> {
  typedef half8v frag;
  static constexpr int CPS = 32, EPL = 8;
  static __device__ __forceinline__ f32x4v mma(const frag& a, const frag& b, f32x4v d) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
  }
  static __device__ __forceinline__ float dot(const frag& f, const frag& a, float s) {
#pragma unroll
    for (int i = 0; i < 8; i++) s = __builtin_fmaf((float)f[i], (float)a[i], s);
    return s;
  }
};
template <> struct MmT<float> {
  typedef f32x4v frag;
  static constexpr int CPS = 16, EPL = 4;
  static __device__ __forceinline__ f32x4v mma(const frag& a, const frag& b, f32x4v d) {
#pragma unroll
    for (int t = 0; t < 4; t++) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], d, 0, 0, 0);
    return d;
  }
  static __device__ __forceinline__ float dot(const frag& f, const frag& a, float s) {
#pragma unroll
    for (int i = 0; i < 4; i++) s = __builtin_fmaf(f[i], a[i], s);
    return s;
  }
};

// Fallback for boxes larger than the patch: the four corner dots of one tap straight from memory.  F2 is addressed
// as position * pstride + chunk * cstride (+ element in chunk): channel-last maps have pstride = C, cstride = EPL; the
// chunk-planar form (MmParams::f2_chunked) has pstride = EPL, cstride = H2 * W2 * EPL.
template <typename T>
__device__ __noinline__ float4 corner_dots(const T* f1p, const T* F2, ptrdiff_t pos11, int C, int W2, int mask,
                                           ptrdiff_t pstride, ptrdiff_t cstride) {
  typedef typename MmT<T>::frag frag;
  constexpr int EPL = MmT<T>::EPL;
  float q11 = 0.f, q21 = 0.f, q12 = 0.f, q22 = 0.f;
  for (int c = 0; c < C; c += EPL) {
    const frag f = *reinterpret_cast<const frag*>(f1p + c);
    const T* base = F2 + (ptrdiff_t)(c / EPL) * cstride;
    if (mask & 1) q11 = MmT<T>::dot(f, *reinterpret_cast<const frag*>(base + pos11 * pstride), q11);
    if (mask & 2) q21 = MmT<T>::dot(f, *reinterpret_cast<const frag*>(base + (pos11 + 1) * pstride), q21);
    if (mask & 4) q12 = MmT<T>::dot(f, *reinterpret_cast<const frag*>(base + (pos11 + W2) * pstride), q12);
    if (mask & 8) q22 = MmT<T>::dot(f, *reinterpret_cast<const frag*>(base + (pos11 + W2 + 1) * pstride), q22);
  }
  return make_float4(q11, q21, q12, q22);
}

// Packed (x low, y high) int16 min / max over each ROW of 16 lanes (every lane of the row gets the result).
template <bool IS_MIN>
__device__ __forceinline__ int row_pk_reduce(int v) {
#define LGU_DPP_STEP(ctrl)                                                  \
  {                                                                         \
    const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false); \
    v = IS_MIN ? pk_min(v, o) : pk_max(v, o);                               \
  }
  LGU_DPP_STEP(0xB1)   // quad_perm:[1,0,3,2]
  LGU_DPP_STEP(0x4E)   // quad_perm:[2,3,0,1]
  LGU_DPP_STEP(0x141)  // row_half_mirror
  LGU_DPP_STEP(0x140)  // row_mirror
#undef LGU_DPP_STEP
  return v;
}

// Diagnostic builds only (tools/diag/run_mm_stamps.py defines LGU_MM_STAMPS): per-wave phase stamps.
#ifdef LGU_MM_STAMPS
__device__ unsigned long long* g_mm_stamps;  // [wave][8]
#define MM_STAMP(i)                                                                                \
  do {                                                                                             \
    if (lane == 0 && blockIdx.y == 0) g_mm_stamps[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define MM_STAMP(i)
#endif

constexpr int MM_MAXL = 4;  // pyramid levels one launch can serve

// Launch parameters.  L == 1: one operator call (lowMem_defSample / altcorr_forward).  L > 1: the per-level loop of
// AltCorrBlock.corr_fn (reference corr.py:192-213) in ONE launch: work items are (level, edge, block) with level 0
// (the largest windows) first, level l samples fmap2[l] at coords / 2^l with offset[l] and writes channels
// l*NT .. (l+1)*NT - 1 of the concatenated output.
struct MmParams {
  const void* fmap1;  // element type = the kernel's T (half or float)
  const void* fmap2[MM_MAXL];
  float* offset[MM_MAXL];  // null = zero offsets for that level (altcorr)
  const float* coords;
  float* corr;
  int H2[MM_MAXL], W2[MM_MAXL];
  int L, B, S, H1, W1, blocks_x, blocks_y, xcd_map, vec_out;
  int lbase;             // pyramid level of fmap2[0]: level l of the launch samples at coords / 2^(lbase + l)
  int lvl0, Ltot;        // output: level l of this launch writes channels (lvl0 + l) * NT .. of Ltot * NT (a pyramid call may be
                         // split into a launch for the levels with offsets and one for the zero-offset levels)
  // fmap2 storage.  0: channel-last (F,H2,W2,C), the operators' layout.  1: chunk-planar (F, C/EPL, H2, W2, EPL) with
  // EPL = 16 bytes of channels: the 16 x-adjacent positions an MFMA B fragment covers are then 256 CONTIGUOUS bytes per
  // 16-byte channel chunk, where channel-last puts them 2C bytes apart.  The vector L1 serves a load quad by quad
  // (4 lanes), one access per distinct 128-byte line in the quad: 64 accesses per fragment load channel-last, 16-20
  // chunk-planar — the access rate, not L2 bandwidth, is what bounds the sweep (AltCorrBlock keeps its pyramid in this form).
  int f2_chunked;
  const long long* ii;   // optional frame indices (device, int64): edge b reads fmap1[ii[b]] and fmap2[l][jj[b]]
  const long long* jj;   // straight from the frame buffers — no gathered per-edge copies; null = fmap*[b]
  const int* orow;       // optional (device, B ints; cooperative kernel only): offset row of edge b, see lgu_lowmem_pyramid_calls_fwd_h16
  int n_orow;
};

// One wave = one workgroup = a block of MT 4 x 4 pixel sub-blocks side by side (4 rows x 4 MT columns).  Every B
// fragment (16 window positions x all channels) fetched by the sweep feeds MT MFMA row tiles, and the union window of
// a 4 x 8 block is barely larger than that of a 4 x 4 one (the tap boxes are ~16 x 16 either way): MT = 2 moves
// 28 positions per pixel-level through L2 -> CU instead of 47.  The price is LDS: 16.6 KB of patches per sub-block,
// i.e. 4 waves per CU (one per SIMD) at MT = 2 against 9 at MT = 1 — which costs more than the traffic saves (see
// launch_mfma below), so MT = 1 is the default and MT = 2 is kept for A/B.
// Lane layout outside the sweep: row = lane / 16 is a pixel of the current pass (pass (m, q) serves row q of sub-block
// m: pixel column 4 m + row), j = lane % 16 carries the taps j, j + 16, j + 32, j + 48 of that pixel, so tap boxes
// reduce inside 16-lane rows with DPP only.
// ZO: a launch whose levels all have structurally zero offsets (levels >= 2 of AltCorrBlock, corr.py:232-235; the r = 1
// probe; altcorr_forward).  All taps of a pixel then share one fractional part and its box is the lattice — known from
// the coords alone: no offset loads, no per-tap box reduction, 8 x 8 patches.  The small LDS and register footprint
// (no offsets held across the sweep) doubles the resident waves, which is what these overhead-bound levels need
// (their sweep is 2-3 us of an 11 us wave life: tools/diag/run_mm_stamps.py).
template <int R, int KS, typename T, int MT, bool ZO>
__global__ __launch_bounds__(kWave, ZO ? 4 : (MT == 1 ? 2 : 1)) void lowmem_mfma_kernel(const MmParams p) {
  typedef typename MmT<T>::frag frag;
  constexpr int CPS = MmT<T>::CPS, EPL = MmT<T>::EPL;
  constexpr int RD = 2 * R + 1, NT = RD * RD, C = CPS * KS;
  constexpr int TI = (NT + 15) / 16;   // tap slots per lane
  constexpr int CEN = R * RD + R;      // centre tap
  constexpr int MM_PF = (KS <= 4 && !ZO) ? 4 : 2;  // position groups in flight (16 bytes x KS per lane each)
  constexpr int BOXP = ZO ? MM_ZBOX : MM_BOX;      // patch row pitch
  constexpr int PP = ZO ? MM_ZPP : MM_PP;          // patch pitch per pixel
  static_assert(!ZO || 2 * R + 2 <= MM_ZBOX, "zero-offset lattice must fit the small patch");
  static_assert(sizeof(frag) == 16, "one 16-byte load per lane and k-step");
  constexpr int NP = MM_BP * MT;        // pixels of the block
  constexpr int MM_OUTP = mm_outp(MT);
  extern __shared__ float smem[];
  float* const patch = smem + MM_GUARD;                               // [NP][PP]
  int* const pbox = reinterpret_cast<int*>(patch + NP * PP);          // [NP][xlo,ylo,bw,bh]
  const int lane = threadIdx.x;
  const int lx = lane & 15, lg = lane >> 4;
  const int B = p.B, S = p.S, H1 = p.H1, W1 = p.W1, blocks_x = p.blocks_x;

  // ---- workgroup -> (level, edge, block) ----
  int b, blk;
  const int blocks = blocks_x * p.blocks_y;
  const int per_level = (p.xcd_map ? ((B + 7) >> 3) * 8 : B) * blocks;
  const int lvl = p.L > 1 ? (int)blockIdx.x / per_level : 0;
  const int item = p.L > 1 ? (int)blockIdx.x - lvl * per_level : (int)blockIdx.x;
  if (p.xcd_map) {  // consecutive workgroup ids go to consecutive XCDs: keep an edge on one XCD
    const int xcd = item & 7, slot = item >> 3;
    b = (slot / blocks) * 8 + xcd;
    blk = slot % blocks;
    if (b >= B) return;
  } else {
    b = item / blocks;
    blk = item % blocks;
  }
  // per-level operands (uniform selects, no dynamic indexing of the argument struct)
  const T* fmap2 = static_cast<const T*>(p.fmap2[0]);
  float* offset = p.offset[0];
  int H2 = p.H2[0], W2 = p.W2[0];
#pragma unroll
  for (int l = 1; l < MM_MAXL; l++)
    if (lvl == l) { fmap2 = static_cast<const T*>(p.fmap2[l]); offset = p.offset[l]; H2 = p.H2[l]; W2 = p.W2[l]; }
  const float cscale = __builtin_ldexpf(1.0f, -(p.lbase + lvl));  // coords / 2^l (corr.py:197): exact in fp32
  const int n = blockIdx.y;
  const int by = blk / blocks_x, bx = blk % blocks_x;
  const size_t HW1 = (size_t)H1 * W1;
  const size_t f1i = p.ii ? (size_t)p.ii[b] : (size_t)b, f2i = p.jj ? (size_t)p.jj[b] : (size_t)b;  // wave-uniform
  const T* const F1 = static_cast<const T*>(p.fmap1) + f1i * HW1 * C;
  const T* const F2 = fmap2 + f2i * H2 * W2 * C;
  // reference indexing kept: offset[b*n] (lowMem_defSample.cu:80-83); null = zero offsets (altcorr)
  float* const obase = (!ZO && offset) ? offset + (size_t)(b * n) * HW1 * NT * 2 : nullptr;
  const float2* const cbase = reinterpret_cast<const float2*>(p.coords) + ((size_t)b * S + n) * HW1;

  MM_STAMP(0);
  // ---- phase 0: sample positions and tap boxes (4 pixels per pass, one per lane row) ----
  int tix[TI], tiy[TI];         // offset / output index [ix][iy] of this lane's tap slots
#pragma unroll
  for (int i = 0; i < TI; i++) {
    const int t = lx + 16 * i;
    tix[i] = t / RD;
    tiy[i] = t - tix[i] * RD;
  }
  float2 cvv[MT][4], o0[MT][4][ZO ? 1 : TI];   // ZO: no offsets (the one dummy entry is never read)
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int h1 = by * 4 + q, w1r = (bx * MT + m) * 4 + lg;  // this lane row's pixel
      const bool pv = h1 < H1 && w1r < W1;
      const size_t pix = pv ? (size_t)h1 * W1 + w1r : 0;
      cvv[m][q] = cbase[pix];
      cvv[m][q].x *= cscale; cvv[m][q].y *= cscale;
      if constexpr (!ZO) {
#pragma unroll
        for (int i = 0; i < TI; i++) {
          const int t = lx + 16 * i;
          o0[m][q][i] = make_float2(0.f, 0.f);
          if (obase && pv && t < NT) o0[m][q][i] = reinterpret_cast<const float2*>(obase + pix * NT * 2)[t];
        }
      }
    }
  // A fragments: lane (lg, lx) holds channels 32 s + 8 lg .. + 7 of pixel lx; requested right behind the
  // coords / offsets (loads return in order: the box computation must not wait for these 4 KB), consumed by the sweep
  frag a[MT][KS];
#pragma unroll
  for (int m = 0; m < MT; m++) {
    int h1 = by * 4 + (lx >> 2), w1 = (bx * MT + m) * 4 + (lx & 3);
    h1 = h1 < H1 ? h1 : H1 - 1; w1 = w1 < W1 ? w1 : W1 - 1;
    const T* ap = F1 + ((size_t)h1 * W1 + w1) * C + EPL * lg;
#pragma unroll
    for (int s = 0; s < KS; s++) a[m][s] = *reinterpret_cast<const frag*>(ap + CPS * s);
  }
  if constexpr (!ZO) if (obase) {  // reference side effect (:80-81): offset[centre] = 0, stored only where the bits are not +0 already
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int h1 = by * 4 + q, w1r = (bx * MT + m) * 4 + lg;
        constexpr int ci = CEN / 16;
        if (lx == CEN % 16 && h1 < H1 && w1r < W1) {
          if ((__builtin_bit_cast(unsigned, o0[m][q][ci].x) | __builtin_bit_cast(unsigned, o0[m][q][ci].y)) != 0u)
            reinterpret_cast<float2*>(obase + ((size_t)h1 * W1 + w1r) * NT * 2)[CEN] = make_float2(0.f, 0.f);
          o0[m][q][ci] = make_float2(0.f, 0.f);
        }
      }
  }
  int blo[MT][4], bwh[MT][4];  // per pass, row-uniform: packed (xlo, ylo); bw | bh << 8 (0 = no patch) | fallback << 16
  int ulo = 0x7fff7fff, uhi = (int)0x80008000;
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int h1 = by * 4 + q, w1r = (bx * MT + m) * 4 + lg;
      const bool pv = h1 < H1 && w1r < W1;
      int lo = 0x7fff7fff, hi = (int)0x80008000;
      if constexpr (ZO) {
        // taps i, j = -R..R at floor(c) + (i, j), corners one further: the lattice [f - R, f + R + 1]^2 clipped to the map
        // (= what the per-tap reduction below yields for zero offsets; the x and y ranges clip independently)
        const int fx = (int)floorf(cvv[m][q].x), fy = (int)floorf(cvv[m][q].y);
        const int xa = fx - R > 0 ? fx - R : 0, xb = fx + R + 1 < W2 ? fx + R + 1 : W2 - 1;
        const int ya = fy - R > 0 ? fy - R : 0, yb = fy + R + 1 < H2 ? fy + R + 1 : H2 - 1;
        if (pv && xa <= xb && ya <= yb) { lo = pk16(xa, ya); hi = pk16(xb, yb); }
      } else {
#pragma unroll
        for (int i = 0; i < TI; i++) {
          const float xs = cvv[m][q].x + o0[m][q][i].x, ys = cvv[m][q].y + o0[m][q][i].y;  // :82-83
          const int w2 = (int)floorf(xs) - R + tix[i], h2 = (int)floorf(ys) - R + tiy[i];
          const int xa = w2 > 0 ? w2 : 0, xb = w2 + 1 < W2 ? w2 + 1 : W2 - 1;
          const int ya = h2 > 0 ? h2 : 0, yb = h2 + 1 < H2 ? h2 + 1 : H2 - 1;
          const bool part = pv && lx + 16 * i < NT && xa <= xb && ya <= yb;  // at least one corner in bounds
          lo = part ? pk_min(lo, pk16(xa, ya)) : lo;
          hi = part ? pk_max(hi, pk16(xb, yb)) : hi;
        }
        lo = row_pk_reduce<true>(lo);
        hi = row_pk_reduce<false>(hi);
      }
      const int xlo = pk_lo(lo), ylo = pk_hi(lo), xhi = pk_lo(hi), yhi = pk_hi(hi);
      const bool any = xhi >= xlo && yhi >= ylo;
      const bool boxed = any && xhi - xlo < BOXP && yhi - ylo < BOXP;
      ulo = boxed ? pk_min(ulo, lo) : ulo;
      uhi = boxed ? pk_max(uhi, hi) : uhi;
      blo[m][q] = lo;
      bwh[m][q] = boxed ? (xhi - xlo + 1) | ((yhi - ylo + 1) << 8) : (any ? 1 << 16 : 0);
      if (lx == 0) {
        int* pb = pbox + (m * MM_BP + q * 4 + lg) * 4;
        pb[0] = xlo; pb[1] = ylo;
        pb[2] = boxed ? xhi - xlo + 1 : 0;
        pb[3] = boxed ? yhi - ylo + 1 : 0;
      }
    }
  // union over the four lane rows = the block window
  ulo = pk_min(pk_min(__builtin_amdgcn_readlane(ulo, 0), __builtin_amdgcn_readlane(ulo, 16)),
               pk_min(__builtin_amdgcn_readlane(ulo, 32), __builtin_amdgcn_readlane(ulo, 48)));
  uhi = pk_max(pk_max(__builtin_amdgcn_readlane(uhi, 0), __builtin_amdgcn_readlane(uhi, 16)),
               pk_max(__builtin_amdgcn_readlane(uhi, 32), __builtin_amdgcn_readlane(uhi, 48)));
  const int UX0 = __builtin_amdgcn_readfirstlane(pk_lo(ulo)), UY0 = __builtin_amdgcn_readfirstlane(pk_hi(ulo));
  const int UX1 = __builtin_amdgcn_readfirstlane(pk_lo(uhi)), UY1 = __builtin_amdgcn_readfirstlane(pk_hi(uhi));
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  MM_STAMP(1);

  // ---- phase 1: sweep the block window with the matrix cores, scatter into the patches ----
  if (UX1 >= UX0 && UY1 >= UY0) {
    // result register r of sub-block m, lane (lg, lx) is pixel 4*lg + r of that sub-block at window position gx0 + lx
    int sbase[MT][4], sxv[MT][4], sylo[MT][4], sbw[MT][4], sbh[MT][4];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int p = m * MM_BP + lg * 4 + r;
        const int xlo = pbox[p * 4 + 0], ylo = pbox[p * 4 + 1];
        sbw[m][r] = pbox[p * 4 + 2]; sbh[m][r] = pbox[p * 4 + 3];
        sxv[m][r] = lx - xlo; sylo[m][r] = ylo;
        sbase[m][r] = p * PP - ylo * BOXP - xlo + lx;
      }
    const int ngx = (UX1 - UX0 + 16) >> 4;  // groups of 16 positions per window row
    const int nit = ngx * (UY1 - UY0 + 1);
    // The window is swept column of groups by column of groups (rows innermost): whether this lane's position lies in
    // the box columns of its four pixels then changes only ngx times per sweep, not every group.
    // k-step s of lane (lg, lx) holds channels CPS s + EPL lg .. + EPL - 1 of position gx0 + lx
    const size_t bstep = p.f2_chunked ? (size_t)4 * H2 * W2 * EPL : (size_t)CPS;  // elements from k-step s to s + 1
    auto bptr = [&](int y, int gx0) {
      int x = gx0 + lx;
      x = x < W2 ? x : W2 - 1;  // padded positions re-read the last column; their results land in no box
      return p.f2_chunked ? F2 + (((size_t)lg * H2 + y) * W2 + x) * EPL : F2 + ((size_t)(y * W2 + x)) * C + EPL * lg;
    };
    // MM_PF groups in flight: slot j holds iteration it + j; it is refilled for it + j + MM_PF right after use.
    // (The sweep is bound by the L2 -> CU read rate, ~70 GB/s per CU; a hand-counted s_waitcnt variant with
    // unconditional loads kept more loads in flight and was not faster: profiles/README.md.)
    frag bq[MM_PF][KS];
    int yl = UY0, gxl = UX0;  // load cursor
#pragma unroll
    for (int j = 0; j < MM_PF; j++) {
      if (j < nit) {
        const T* p = bptr(yl, gxl);
#pragma unroll
        for (int s = 0; s < KS; s++) bq[j][s] = *reinterpret_cast<const frag*>(p + bstep * s);
        yl++;
        if (yl > UY1) { yl = UY0; gxl += 16; }
      }
    }
    MM_STAMP(2);
    int y = UY0, gx0 = UX0;   // compute cursor
    bool xok[MT][4];          // position gx0 + lx inside the box columns of pixel 4 lg + r (lane masks, per column of groups)
    auto column = [&](int g0) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int r = 0; r < 4; r++) xok[m][r] = (unsigned)(g0 + sxv[m][r]) < (unsigned)sbw[m][r];
    };
    column(UX0);
    for (int it = 0; it < nit; it += MM_PF) {
#pragma unroll
      for (int j = 0; j < MM_PF; j++) {
        if (it + j < nit) {  // wave-uniform
          f32x4v d[MT];
#pragma unroll
          for (int m = 0; m < MT; m++) {
            d[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; s++) d[m] = MmT<T>::mma(a[m][s], bq[j][s], d[m]);
          }
          if (it + j + MM_PF < nit) {
            const T* p = bptr(yl, gxl);
#pragma unroll
            for (int s = 0; s < KS; s++) bq[j][s] = *reinterpret_cast<const frag*>(p + bstep * s);
            yl++;
            if (yl > UY1) { yl = UY0; gxl += 16; }
          }
          const int yg = y * BOXP + gx0;
#pragma unroll
          for (int m = 0; m < MT; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const unsigned qy = (unsigned)(y - sylo[m][r]);
              if (xok[m][r] && qy < (unsigned)sbh[m][r]) patch[sbase[m][r] + yg] = d[m][r];
            }
          y++;
          if (y > UY1) {  // next column of groups (wave-uniform)
            y = UY0;
            gx0 += 16;
            column(gx0);
          }
        }
      }
    }
  }
  MM_STAMP(3);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- phase 2: sample the patches (same lane layout as phase 0) ----
  {  // tap indices are recomputed (from a laundered lane id) rather than kept live across the sweep
    int lxo = lx;
    asm volatile("" : "+v"(lxo));
#pragma unroll
    for (int i = 0; i < TI; i++) {
      const int t = lxo + 16 * i;
      tix[i] = t / RD;
      tiy[i] = t - tix[i] * RD;
    }
  }
  float res[MT][4][TI];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int h1 = by * 4 + q, w1r = (bx * MT + m) * 4 + lg;
      const bool pv = h1 < H1 && w1r < W1;
      const int xlo = pk_lo(blo[m][q]), ylo = pk_hi(blo[m][q]);
      const bool has_patch = (bwh[m][q] & 0xff) != 0, fallback = (bwh[m][q] >> 16) != 0;
      const float* const Dp = patch + (m * MM_BP + q * 4 + lg) * PP;
      // zero-offset levels: one sample position per pixel — floor, fraction and the four weights once per pass
      // (the products and their order are bilerp()'s)
      const float zfx = floorf(cvv[m][q].x), zfy = floorf(cvv[m][q].y);
      const float zdx = cvv[m][q].x - zfx, zdy = cvv[m][q].y - zfy;
      const float zw11 = (1.0f - zdy) * (1.0f - zdx), zw21 = (1.0f - zdy) * zdx, zw12 = zdy * (1.0f - zdx), zw22 = zdy * zdx;
#pragma unroll
      for (int i = 0; i < TI; i++) {
        const bool tv = pv && lx + 16 * i < NT;
        float fxs, fys, dx, dy;
        if constexpr (ZO) {
          fxs = zfx; fys = zfy; dx = zdx; dy = zdy;
        } else {
          const float xs = cvv[m][q].x + o0[m][q][i].x, ys = cvv[m][q].y + o0[m][q][i].y;
          fxs = floorf(xs); fys = floorf(ys);
          dx = xs - fxs; dy = ys - fys;  // :87-88
        }
        const int w2 = (int)fxs - R + tix[i], h2 = (int)fys - R + tiy[i];
        // per-axis range checks (one unsigned compare each), combined per corner
        const bool bx0 = (unsigned)w2 < (unsigned)W2, bx1 = (unsigned)(w2 + 1) < (unsigned)W2;
        const bool by0 = (unsigned)h2 < (unsigned)H2, by1 = (unsigned)(h2 + 1) < (unsigned)H2;
        const bool b11 = by0 && bx0, b21 = by0 && bx1, b12 = by1 && bx0, b22 = by1 && bx1;
        float q11 = 0.f, q21 = 0.f, q12 = 0.f, q22 = 0.f;
        if (tv && has_patch) {
          // the four corners are read unconditionally (two 8-byte LDS reads) and the per-corner zero padding is a
          // select: an in-bounds corner always lies inside the pixel's box, so its entry is where the index says; the
          // others read whatever is there (inside the LDS allocation: MM_GUARD floats in front of the patches cover a
          // top-left one row / column before the box, the box table behind them covers the bottom-right overshoot)
          const int idx = ((by0 || by1) && (bx0 || bx1)) ? (h2 - ylo) * BOXP + (w2 - xlo) : 0;
          const float* D = Dp + idx;
          const float d0 = D[0], d1 = D[1], d2 = D[BOXP], d3 = D[BOXP + 1];
          q11 = b11 ? d0 : 0.f;
          q21 = b21 ? d1 : 0.f;
          q12 = b12 ? d2 : 0.f;
          q22 = b22 ? d3 : 0.f;
        } else if (tv && fallback) {  // box larger than 16 x 16: this tap's four corner dots, channels in order
          const float4 qq = corner_dots<T>(F1 + ((size_t)h1 * W1 + w1r) * C, F2, (ptrdiff_t)h2 * W2 + w2, C, W2,
                                          (b11 ? 1 : 0) | (b21 ? 2 : 0) | (b12 ? 4 : 0) | (b22 ? 8 : 0),
                                          p.f2_chunked ? EPL : C, p.f2_chunked ? (ptrdiff_t)H2 * W2 * EPL : EPL);
          q11 = qq.x; q21 = qq.y; q12 = qq.z; q22 = qq.w;
        }
        if constexpr (ZO) res[m][q][i] = q11 * zw11 + q21 * zw21 + q12 * zw12 + q22 * zw22;
        else res[m][q][i] = bilerp(q11, q21, q12, q22, dx, dy);  // :114-117, per-corner zero padding
      }
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  MM_STAMP(4);

  // ---- write-out: corr[b][n][ix][iy][h1][w1]; the patch region becomes the [tap][pixel] transpose tile, pixel index
  // inside a tap row = q * 4 MT + m * 4 + column: a block row is 4 MT consecutive floats ----
  float* const outt = patch;
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int i = 0; i < TI; i++)
        if (lx + 16 * i < NT) outt[(lx + 16 * i) * MM_OUTP + q * 4 * MT + m * 4 + lg] = res[m][q][i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float* const cb = p.corr + (((size_t)b * S + n) * p.Ltot + p.lvl0 + lvl) * NT * HW1;
  for (int idx = lane; idx < NT * 4 * MT; idx += kWave) {
    const int t = idx / (4 * MT), qm = idx - t * (4 * MT);
    const int q = qm / MT, m = qm - q * MT;
    const int h1 = by * 4 + q, w1 = (bx * MT + m) * 4;
    if (h1 >= H1 || w1 >= W1) continue;
    const float4 v = *reinterpret_cast<const float4*>(outt + t * MM_OUTP + q * 4 * MT + m * 4);
    float* dst = cb + ((size_t)t * H1 + h1) * W1 + w1;
    if (p.vec_out) {
      *reinterpret_cast<float4*>(dst) = v;
    } else {
      dst[0] = v.x;
      if (w1 + 1 < W1) dst[1] = v.y;
      if (w1 + 2 < W1) dst[2] = v.z;
      if (w1 + 3 < W1) dst[3] = v.w;
    }
  }
  MM_STAMP(5);
}

template <int R, int KS, typename T, int MT, bool ZO>
static int launch_mfma_mt(MmParams p, hipStream_t st) {
  const size_t lds = sizeof(float) * (size_t)mm_lds_floats(MT, ZO);
  auto kern = lowmem_mfma_kernel<R, KS, T, MT, ZO>;
  p.blocks_x = (p.W1 + 4 * MT - 1) / (4 * MT);
  p.blocks_y = (p.H1 + 3) / 4;
  const int blocks = p.blocks_x * p.blocks_y;
  p.xcd_map = p.B >= 8 ? 1 : 0;
  const size_t nwg = (size_t)p.L * (p.xcd_map ? (size_t)((p.B + 7) / 8) * 8 * blocks : (size_t)p.B * blocks);
  if (nwg >= (1u << 31)) return -1;
  p.vec_out = (p.W1 % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.corr) & 15) == 0);
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg, (unsigned)p.S), dim3(kWave), lds, st, p);
  return launch_status();
}

// One pyramid call = one launch per run of levels of equal kind (offsets / structurally zero offsets): AltCorrBlock's
// [off, off, null, null] is two launches, the second with the ZO kernel.
// LGU_LOWMEM_MT (debug / A-B only): pixel sub-blocks per wave, 1 (default) or 2.  Measured (tools/ab_lowmem.py,
// profiles/r02_ab_lowmem_mt.jsonl, config 4 shapes): MT = 2 moves 1.7x fewer bytes from L2 and wins only at level 0
// (108 vs 116 us); at 4 waves per CU the latency of the box and sampling phases is exposed and the coarse levels lose
// (51 vs 37 us, 49 vs 32 us): 229 vs 220 us for the four levels in one launch.
// LGU_LOWMEM_ZO (debug / A-B only): 0 = zero-offset levels through the general kernel too.
// (Running the two launches of a call concurrently — the second on a side stream forked and joined with events — was
// measured and lost: 163.5 vs 146.3 us for config 4; the launches stay in order on the caller's stream.)
template <int R, int KS, typename T>
static int launch_mfma(const MmParams& p, hipStream_t st) {
  const bool mt2 = env_int("LGU_LOWMEM_MT", 1) >= 2 && p.W1 > 4;
  const bool zo_on = env_int("LGU_LOWMEM_ZO", 1) != 0;
  int l0 = 0;
  while (l0 < p.L) {
    const bool zo = zo_on && p.offset[l0] == nullptr;
    int l1 = l0 + 1;
    while (l1 < p.L && (zo_on && p.offset[l1] == nullptr) == zo) l1++;
    MmParams q = p;
    q.L = l1 - l0;
    q.lbase = p.lbase + l0;
    q.lvl0 = p.lvl0 + l0;
    for (int l = 0; l < MM_MAXL; l++) {
      const bool on = l < q.L;
      q.fmap2[l] = on ? p.fmap2[l0 + l] : nullptr;
      q.offset[l] = on ? p.offset[l0 + l] : nullptr;
      q.H2[l] = on ? p.H2[l0 + l] : 0;
      q.W2[l] = on ? p.W2[l0 + l] : 0;
    }
    int rc;
    if (zo) rc = mt2 ? launch_mfma_mt<R, KS, T, 2, true>(q, st) : launch_mfma_mt<R, KS, T, 1, true>(q, st);
    else rc = mt2 ? launch_mfma_mt<R, KS, T, 2, false>(q, st) : launch_mfma_mt<R, KS, T, 1, false>(q, st);
    if (rc != LGU_OK) return rc;
    l0 = l1;
  }
  return LGU_OK;
}

// lowmem_coop.hip: four waves share the swept windows, all levels in one wave life (half maps, C <= 128)
int lowmem_coop_dispatch(const void* fmap1, const void* const* fmap2, float* const* offset, const float* coords, float* corr,
                         const int* H2, const int* W2, int L, int B, int S, int H1, int W1, int C, int radius, int lbase,
                         int lvl0, int Ltot, int f2_chunked, const long long* ii, const long long* jj, const int* orow, int n_orow,
                         hipStream_t st);

template <typename T>
static int mfma_dispatch(const MmParams& p, int C, int radius, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    const int rc = lowmem_coop_dispatch(p.fmap1, p.fmap2, p.offset, p.coords, p.corr, p.H2, p.W2, p.L, p.B, p.S, p.H1, p.W1, C,
                                        radius, p.lbase, p.lvl0, p.Ltot, p.f2_chunked, p.ii, p.jj, p.orow, p.n_orow, st);
    if (rc >= 0) return rc;
  }
  if (p.orow) return -1;  // only the cooperative kernel knows offset rows per edge
  uintptr_t al = reinterpret_cast<uintptr_t>(p.fmap1);
  for (int l = 0; l < p.L; l++) al |= reinterpret_cast<uintptr_t>(p.fmap2[l]);
  if (radius < 1 || radius > 3 || (al & 15) != 0 || p.S > 65535) return -1;
  if ((size_t)p.H1 * p.W1 * C >= (1u << 31)) return -1;
  for (int l = 0; l < p.L; l++)
    if ((size_t)p.H2[l] * p.W2[l] * C >= (1u << 31) || p.H2[l] > 32767 || p.W2[l] > 32767) return -1;
  constexpr int cps = MmT<T>::CPS;  // half: C in {32,64,128,256}; float: C in {16,32,64,128}
#define LGU_MM_CASE(RV, KSV) \
  if (radius == RV && C == cps * KSV) return launch_mfma<RV, KSV, T>(p, st);
  LGU_MM_CASE(3, 4) LGU_MM_CASE(1, 4) LGU_MM_CASE(2, 4)
  LGU_MM_CASE(3, 1) LGU_MM_CASE(1, 1) LGU_MM_CASE(2, 1)
  LGU_MM_CASE(3, 2) LGU_MM_CASE(1, 2) LGU_MM_CASE(2, 2)
  LGU_MM_CASE(3, 8) LGU_MM_CASE(1, 8) LGU_MM_CASE(2, 8)
#undef LGU_MM_CASE
  return -1;
}

// Return -1 when the matrix-core kernel does not serve the arguments (the caller then uses the VALU tile kernel).
static MmParams single_level(const void* fmap1, const void* fmap2, const float* coords, float* offset, float* corr, int B,
                             int S, int H1, int W1, int H2, int W2) {
  MmParams p = {};
  p.fmap1 = fmap1; p.fmap2[0] = fmap2; p.offset[0] = offset; p.coords = coords; p.corr = corr;
  p.H2[0] = H2; p.W2[0] = W2;
  p.L = 1; p.B = B; p.S = S; p.H1 = H1; p.W1 = W1;
  p.lvl0 = 0; p.Ltot = 1;
  return p;
}
int lowmem_mfma_dispatch(const _Float16* fmap1, const _Float16* fmap2, const float* coords, float* offset, float* corr,
                         int B, int S, int H1, int W1, int H2, int W2, int C, int radius, hipStream_t st) {
  return mfma_dispatch<_Float16>(single_level(fmap1, fmap2, coords, offset, corr, B, S, H1, W1, H2, W2), C, radius, st);
}
int lowmem_mfma_dispatch_f32(const float* fmap1, const float* fmap2, const float* coords, float* offset, float* corr,
                             int B, int S, int H1, int W1, int H2, int W2, int C, int radius, hipStream_t st) {
  return mfma_dispatch<float>(single_level(fmap1, fmap2, coords, offset, corr, B, S, H1, W1, H2, W2), C, radius, st);
}

}  // namespace lgu

extern "C" {

static int pyramid_entry(bool half, const void* fmap1, const void* const* fmap2, const float* coords, float* const* offsets,
                         float* out, int L, int lbase, int B, int S, int H1, int W1, const int* H2, const int* W2, int C, int NO,
                         int radius, const long long* ii, const long long* jj, void* stream, bool chunked = false,
                         const int* orow = nullptr) {
  using namespace lgu;
  if (orow && (S != 1 || NO < 1)) return LGU_E_BADARG;
  if (!fmap1 || !fmap2 || !coords || !offsets || !out || !H2 || !W2) return LGU_E_BADARG;
  if (L < 1 || L > MM_MAXL || lbase < 0 || lbase > 16 || B < 0 || S < 1 || H1 < 1 || W1 < 1 || C < 1 || radius < 0)
    return LGU_E_BADARG;
  if ((long long)(B - 1) * (S - 1) >= (long long)NO || (ii == nullptr) != (jj == nullptr)) return LGU_E_BADARG;
  MmParams p = {};
  p.orow = orow; p.n_orow = NO;
  p.fmap1 = fmap1;
  for (int l = 0; l < L; l++) {
    if (!fmap2[l] || H2[l] < 1 || W2[l] < 1) return LGU_E_BADARG;
    p.fmap2[l] = fmap2[l];
    p.offset[l] = offsets[l];
    p.H2[l] = H2[l]; p.W2[l] = W2[l];
  }
  p.coords = coords; p.corr = out;
  p.L = L; p.B = B; p.S = S; p.H1 = H1; p.W1 = W1;
  p.lbase = lbase; p.ii = ii; p.jj = jj;
  p.lvl0 = 0; p.Ltot = L;
  p.f2_chunked = chunked ? 1 : 0;
  if (B == 0) return LGU_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rc = half ? mfma_dispatch<_Float16>(p, C, radius, st) : mfma_dispatch<float>(p, C, radius, st);
  return rc < 0 ? LGU_E_UNSUPPORTED : rc;
}

int lgu_lowmem_pyramid_fwd_h16(const void* fmap1, const void* const* fmap2, const float* coords, float* const* offsets,
                               float* out, int L, int lbase, int B, int S, int H1, int W1, const int* H2, const int* W2, int C,
                               int NO, int radius, const long long* ii, const long long* jj, void* stream) {
  return pyramid_entry(true, fmap1, fmap2, coords, offsets, out, L, lbase, B, S, H1, W1, H2, W2, C, NO, radius, ii, jj, stream);
}

int lgu_lowmem_pyramid_fwd_f32(const float* fmap1, const float* const* fmap2, const float* coords, float* const* offsets,
                               float* out, int L, int lbase, int B, int S, int H1, int W1, const int* H2, const int* W2, int C,
                               int NO, int radius, const long long* ii, const long long* jj, void* stream) {
  return pyramid_entry(false, fmap1, reinterpret_cast<const void* const*>(fmap2), coords, offsets, out, L, lbase, B, S, H1, W1,
                       H2, W2, C, NO, radius, ii, jj, stream);
}

int lgu_lowmem_pyramid_chunked_fwd_h16(const void* fmap1, const void* const* fmap2, const float* coords,
                                       float* const* offsets, float* out, int L, int lbase, int B, int S, int H1, int W1,
                                       const int* H2, const int* W2, int C, int NO, int radius, const long long* ii,
                                       const long long* jj, void* stream) {
  return pyramid_entry(true, fmap1, fmap2, coords, offsets, out, L, lbase, B, S, H1, W1, H2, W2, C, NO, radius, ii, jj, stream, true);
}

int lgu_lowmem_pyramid_chunked_fwd_f32(const float* fmap1, const float* const* fmap2, const float* coords,
                                       float* const* offsets, float* out, int L, int lbase, int B, int S, int H1, int W1,
                                       const int* H2, const int* W2, int C, int NO, int radius, const long long* ii,
                                       const long long* jj, void* stream) {
  return pyramid_entry(false, fmap1, reinterpret_cast<const void* const*>(fmap2), coords, offsets, out, L, lbase, B, S, H1, W1,
                       H2, W2, C, NO, radius, ii, jj, stream, true);
}

int lgu_lowmem_pyramid_calls_fwd_h16(const void* fmap1, const void* const* fmap2, const float* coords, float* const* offsets,
                                     float* out, int L, int lbase, int B, int H1, int W1, const int* H2, const int* W2, int C,
                                     int NO, const int* off_row, int radius, const long long* ii, const long long* jj,
                                     int chunked, void* stream) {
  if (!off_row) return LGU_E_BADARG;
  return pyramid_entry(true, fmap1, fmap2, coords, offsets, out, L, lbase, B, 1, H1, W1, H2, W2, C, NO, radius, ii, jj, stream,
                       chunked != 0, off_row);
}

}  // extern "C"

#ifdef LGU_MM_STAMPS
// Diagnostic build only (tools/diag): never part of liblgu_corr.so.
extern "C" {
int lgu_mm_diag_set_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(lgu::g_mm_stamps), &p, sizeof(p)); }
int lgu_mm_diag_lowmem(const void* fmap1, const void* fmap2, const float* coords, float* offset, float* corr, int B, int S,
                       int H1, int W1, int H2, int W2, int C, int radius, int chunked, void* stream) {
  lgu::MmParams p = lgu::single_level(fmap1, fmap2, coords, offset, corr, B, S, H1, W1, H2, W2);
  p.f2_chunked = chunked;
  return lgu::mfma_dispatch<_Float16>(p, C, radius, reinterpret_cast<hipStream_t>(stream));
}
}
#endif
