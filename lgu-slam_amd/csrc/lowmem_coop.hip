// lowmem_coop.hip — on-the-fly deformable correlation from HALF feature maps on the matrix cores, with the swept
// windows of the target map SHARED by the four waves of a workgroup and all pyramid levels served by one wave life.
//
// Same operator and same patch formulation as lowmem_mfma.hip (reference kernels replaced:
// offersample_LGS/lowMem_defSample.cu:27-134, src/altcorr_kernel.cu:27-149; per-level loop of AltCorrBlock.corr_fn,
// droid_slam/modules/corr.py:192-213): a pixel's 49 taps only need its LOCAL correlation patch
//     D_p[y2][x2] = sum_c fmap1[p][c] * fmap2[y2][x2][c]      over its tap box (<= 16 x 16 positions),
// computed with v_mfma_f32_16x16x32_f16 and sampled from LDS with the reference's per-corner zero padding.
//
// What bounds the one-wave-per-block kernel is the rate at which window positions reach the matrix cores (L2 -> CU:
// 11 KB per pixel at level 0, each 16-position fragment feeding 16 pixels).  Here
//   * a workgroup = 4 waves = a 4 x 16 pixel tile (four 4 x 4 sub-blocks side by side).  The union window of the tile is
//     only ~1.5 x that of one sub-block (tap boxes are ~14 wide whatever the tile), and its rows are dealt to the four
//     waves: every wave keeps the fmap1 fragments of ALL 64 pixels in registers and multiplies each position fragment
//     it loads with the four sub-blocks -> 64 pixels per fragment instead of 16 (4.3 KB per pixel at level 0), with
//     no staging of the window in LDS and no more loads per wave;
//   * the MFMA is issued with the POSITIONS as rows (A) and the pixels as columns (B): a lane then holds four
//     x-adjacent positions of ONE pixel, so the scatter into the pixel's patch is two 8-byte LDS stores with one row
//     and two column range checks (the pixel-rows form needs four 4-byte stores to four different pixels, each with
//     its own checks: ~45 VALU instructions per fragment).  Patches are 16 rows x 18 columns with the origin at the
//     box corner rounded down to an even column, so the pairs stay 8-byte aligned;
//   * one wave life serves every level of the call: coords and the fmap1 fragments are loaded once, and the latency of
//     one level's phases overlaps the other workgroup of the CU (2 workgroups = 8 waves per CU, 75 KB of LDS each).
// Per level: boxes (own sub-block) | barrier | cooperative sweep | barrier | sampling + write-out (own sub-block).
// Half products are exact in fp32 and the accumulation is fp32: the result differs from the `.float()` call site by
// summation order only (tests: 1e-5).  Boxes wider than 16 (|offset| >= 4: never produced by corr.py:126-131) take the
// per-tap fallback, as in lowmem_mfma.hip.
#include "lgu_common.hpp"

namespace lgu {

typedef _Float16 cohalf8 __attribute__((ext_vector_type(8)));
typedef float cof32x4 __attribute__((ext_vector_type(4)));

constexpr int CO_SB = 2;                       // 4 x 4 sub-blocks per workgroup (tile = 4 rows x 8 columns of pixels)
constexpr int CO_NW = 4;                       // waves per workgroup: two per sub-block in the box / sampling phases
constexpr int CO_QP = 16 * CO_SB / (4 * CO_NW);  // passes (pixel rows) per wave = 2
constexpr int CO_NPX = 16 * CO_SB;             // pixels per workgroup
constexpr int CO_BOXP = 18;                    // patch row pitch: 16 columns + the even-alignment slack, pairs 8-byte aligned
constexpr int CO_PP = 16 * CO_BOXP + 2;        // floats per patch (+2: consecutive pixels start 8 banks apart)
constexpr int CO_GUARD = 20;                   // floats in front of the patches (>= CO_BOXP + 1): see the sampling phase
constexpr int CO_OUTP = 4 * CO_QP + 4;          // output transpose pitch
constexpr int CO_MAXL = 4;
// Alignment mask of the tile window's first column: pairs of positions start at even columns (needed).  (Round 2 measured
// 7 — fragment loads starting on 128-byte lines in the chunk-planar form: L2 read requests 4.22 -> 3.74 KB per pixel-level,
// but the wider windows cost 2.5 % of time.)
constexpr int CO_XALIGN = 1;
constexpr int CO_LDS_FLOATS = CO_GUARD + CO_NPX * CO_PP + CO_NPX * 4 + 2 * CO_NW + 8;

struct CoParams {
  const _Float16* fmap1;
  const _Float16* fmap2[CO_MAXL];
  float* offset[CO_MAXL];   // null = zero offsets for that level
  const float* coords;
  float* corr;
  int H2[CO_MAXL], W2[CO_MAXL];
  int L, B, S, H1, W1, tiles_x, tiles_y, xcd_map, vec_out;
  int lbase, lvl0, Ltot;
  int f2_chunked;
  // Work units.  The first n_fused workgroups serve ALL levels of their (edge, tile) in one wave life; the remaining
  // n_split (edge, tile) items are served level group by level group (group k = levels gl0[k] .. gl0[k + 1] - 1), groups in
  // launch order: n_fused is a whole number of rounds over the device's workgroup slots, and the short units fill the
  // last, partial round (launch_coop).
  int n_fused, n_split, ngroups, gl0[CO_MAXL + 1];
  const long long* ii;
  const long long* jj;
  // Several reference calls in one launch (lgu_lowmem_pyramid_calls_fwd_h16): edge b samples with offset row orow[b] — the
  // first edge of ITS call — instead of row b*n.  Null = one call.  Values are clamped to n_orow - 1 (no wild reads).
  const int* orow;
  int n_orow;
};

// one corner dot of one tap straight from memory (boxes larger than the patch): channels in order, as the reference sums
// them.  One corner per call and not inlined: the rare path must not set the register count of the sampling phase.
__device__ __noinline__ float co_corner_dot(const _Float16* f1p, const _Float16* f2p, int C, ptrdiff_t cstride) {
  float s = 0.f;
  for (int c = 0; c < C; c += 8) {
    const cohalf8 f = *reinterpret_cast<const cohalf8*>(f1p + c);
    const cohalf8 a = *reinterpret_cast<const cohalf8*>(f2p + (ptrdiff_t)(c / 8) * cstride);
#pragma unroll
    for (int i = 0; i < 8; i++) s = __builtin_fmaf((float)f[i], (float)a[i], s);
  }
  return s;
}
__device__ __forceinline__ float4 co_corner_dots(const _Float16* f1p, const _Float16* F2, ptrdiff_t pos11, int C, int W2, int mask,
                                                 ptrdiff_t pstride, ptrdiff_t cstride) {
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (mask & 1) q.x = co_corner_dot(f1p, F2 + pos11 * pstride, C, cstride);
  if (mask & 2) q.y = co_corner_dot(f1p, F2 + (pos11 + 1) * pstride, C, cstride);
  if (mask & 4) q.z = co_corner_dot(f1p, F2 + (pos11 + W2) * pstride, C, cstride);
  if (mask & 8) q.w = co_corner_dot(f1p, F2 + (pos11 + W2 + 1) * pstride, C, cstride);
  return q;
}

template <bool IS_MIN>
__device__ __forceinline__ int co_row_pk_reduce(int v) {
#define LGU_DPP_STEP(ctrl)                                                  \
  {                                                                         \
    const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false); \
    v = IS_MIN ? pk_min(v, o) : pk_max(v, o);                               \
  }
  LGU_DPP_STEP(0xB1)
  LGU_DPP_STEP(0x4E)
  LGU_DPP_STEP(0x141)
  LGU_DPP_STEP(0x140)
#undef LGU_DPP_STEP
  return v;
}

#ifdef LGU_MM_STAMPS
__device__ unsigned long long* g_co_stamps;  // [workgroup][wave][8 per level]
#define CO_STAMP(i)                                                                                                        \
  do {                                                                                                                     \
    if (lane == 0 && blockIdx.y == 0) g_co_stamps[((size_t)blockIdx.x * CO_NW + wv) * 32 + lvl * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define CO_STAMP(i)
#endif

// Lane layout outside the sweep (as in lowmem_mfma.hip): row = lane / 16 is a pixel of the current pass (pass q serves row
// q of the wave's 4 x 4 sub-block: pixel column = lane row), j = lane % 16 carries the taps j, j + 16, j + 32, j + 48.
// Inside the sweep lane (lg, lx) holds, per sub-block m, positions gx0 + 4 lg .. + 3 of pixel lx (= row lx / 4, column
// lx % 4 of the sub-block).
// Every phase derives its lane indices from a freshly laundered lane id: left alone, the compiler hoists the
// level-invariant per-lane address arithmetic of ALL phases out of the level loop and shares it between the phases,
// which keeps ~100 registers live across the sweep and spills the fmap1 fragments to scratch.
#define CO_FRESH_LANE()                 \
  int lane = lane0;                     \
  asm volatile("" : "+v"(lane));        \
  const int lx = lane & 15, lg = lane >> 4;

// Compiled for 3 waves per SIMD (168 registers).  Build-time experiments on this kernel are made with tools/build_variant.py
// from modified copies and recorded under profiles/ (r02_lowmem_coop_ablation.txt, r03_ab_lowmem_coop_variants.txt) — what
// they established, in one place, so that the source carries no dead switches:
//   * occupancy is not the bound: builds for four waves per SIMD (fragments / offsets re-requested: 121 registers; tap boxes
//     of all levels in a prologue: 124) run no faster than three, two waves per SIMD 27 % slower;
//   * more loads in flight make it slower (3 / 4 position fragments: + 10 / + 17 %); s_setprio around the sweep, skipping
//     the surplus slot of odd trip counts, exec set directly around the patch stores, kernarg scalars pinned in registers,
//     -fno-slp-vectorize / other schedulers: all within +- 2 %;
//   * ablating fragment loads, patch stores, MFMAs and patch reads TOGETHER removes 16 % of the time;
//   * what pays is removing a wait: a branch around a load, a loop with a load-use chain per trip.
// CO_WPS_N: waves per SIMD the kernel is compiled for (also sizes the launcher's round rule).
#ifndef CO_WPS_N
#define CO_WPS_N 3
#endif
constexpr int CO_WPS = CO_WPS_N;
template <int R, int KS>
__global__ __launch_bounds__(kWave* CO_NW, CO_WPS) void lowmem_coop_kernel(const CoParams p_arg) {
  // The parameter block is read where it lies (the kernarg segment: the only argument, at offset 0), so per-level
  // fields are scalar loads at a computed offset instead of select chains over registers that do not fit the SGPR file.
  typedef const CoParams __attribute__((address_space(4))) CoParamsK;
  CoParamsK& p = *(CoParamsK*)__builtin_amdgcn_kernarg_segment_ptr();
  (void)p_arg;
  typedef _Float16 T;
  typedef cohalf8 frag;
  constexpr int CPS = 32, EPL = 8;
  constexpr int RD = 2 * R + 1, NT = RD * RD, C = CPS * KS;
  constexpr int TI = (NT + 15) / 16;
  constexpr int CEN = R * RD + R;
  constexpr int PF = 2;  // position fragments in flight per wave (3 do not fit the registers: 120 against 107 us)
  extern __shared__ float smem[];
  float* const patch = smem + CO_GUARD;                                   // [CO_NPX][CO_PP]
  int* const pbox = reinterpret_cast<int*>(patch + CO_NPX * CO_PP);       // [CO_NPX][xlo, ylo, bw, bh]
  int* const swin = pbox + CO_NPX * 4;                                    // [CO_NW][lo, hi] packed: window of each wave's pixels
  const int lane0 = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave id: scalar
  const int B = p.B, S = p.S, H1 = p.H1, W1 = p.W1;

  // ---- workgroup -> (levels, edge, tile) ----
  int b, blk;
  const int tiles = p.tiles_x * p.tiles_y;
  int item = (int)blockIdx.x, lv0 = 0, lv1 = p.L;
  if (item >= p.n_fused) {
    const int u = item - p.n_fused, grp = u / p.n_split;
    item = p.n_fused + (u - grp * p.n_split);
    lv0 = p.gl0[grp]; lv1 = p.gl0[grp + 1];
  }
  if (p.xcd_map) {
    const int xcd = item & 7, slot = item >> 3;
    b = (slot / tiles) * 8 + xcd;
    blk = slot % tiles;
    if (b >= B) return;  // workgroup-uniform: no barrier is skipped by part of a workgroup
  } else {
    b = item / tiles;
    blk = item % tiles;
  }
  const int n = blockIdx.y;
  const int by = blk / p.tiles_x, bx = blk % p.tiles_x;
  const int msb = wv >> 1, qr0 = (wv & 1) * CO_QP;  // this wave's sub-block and the first of its pixel rows in it
  const int px0 = bx * (4 * CO_SB) + msb * 4;  // first pixel column of this wave's sub-block
  const size_t HW1 = (size_t)H1 * W1;
  const size_t f1i = p.ii ? (size_t)p.ii[b] : (size_t)b, f2i = p.jj ? (size_t)p.jj[b] : (size_t)b;
  const T* const F1 = p.fmap1 + f1i * HW1 * C;
  const float2* const cbase = reinterpret_cast<const float2*>(p.coords) + ((size_t)b * S + n) * HW1;

  // Levels are served coarse to fine.  A level's offsets (16 registers) are requested one step ahead of their use: right
  // after the sampling of the level before it, so they travel during that level's write-out and the barrier (held across
  // a whole zero-offset level they cost 16 registers where the kernel has none to spare).
  const int orow_b = p.orow ? min(max(p.orow[b], 0), p.n_orow - 1) : b * n;
  auto level_offsets = [&](int l) -> float* {  // workgroup-uniform; reference indexing kept: offset[b*n] (lowMem_defSample.cu:80-83)
    float* o = p.offset[l];
    return o ? o + (size_t)orow_b * HW1 * NT * 2 : nullptr;
  };
  float2 o0[CO_QP][TI];
  auto request_offsets = [&](int l) __attribute__((always_inline)) {  // offsets of level l, if it is served here and has any
    if (l < lv0) return;
    const float* const ob = level_offsets(l);
    if (!ob) return;
    CO_FRESH_LANE();
    // Unconditional loads from clamped positions (pixels outside the image read pixel 0, lanes past the last tap read it
    // again: their values are never used): a load behind a lane condition becomes an exec-mask branch with a full wait for
    // every outstanding load behind it, which serialises the request with whatever else is in flight.
#pragma unroll
    for (int q = 0; q < CO_QP; q++) {
      const int h1 = by * 4 + qr0 + q, w1r = px0 + lg;
      const bool pv = h1 < H1 && w1r < W1;
      const unsigned pixo = pv ? ((unsigned)h1 * (unsigned)W1 + (unsigned)w1r) * (unsigned)(NT * 2 * sizeof(float)) : 0u;  // < 2^32 (host-checked)
#pragma unroll
      for (int i = 0; i < TI; i++) {
        int t = lx + 16 * i;
        t = t < NT ? t : NT - 1;
        o0[q][i] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(ob) + (size_t)(pixo + (unsigned)t * 8u));
      }
    }
  };

  // ---- once per wave life: coords of the own pixels, the first offsets, fmap1 fragments of the whole tile (requested
  // last: loads return in order, and the boxes must not wait for these 8 KB) ----
  float2 cv0[CO_QP];
  frag a[CO_SB][KS];
#pragma unroll
  for (int q = 0; q < CO_QP; q++)
#pragma unroll
    for (int i = 0; i < TI; i++) o0[q][i] = make_float2(0.f, 0.f);
#define CO_LOAD_COORDS()                                               \
  {                                                                    \
    CO_FRESH_LANE();                                                   \
    _Pragma("unroll") for (int q = 0; q < CO_QP; q++) {                \
      const int h1 = by * 4 + qr0 + q, w1r = px0 + lg;                 \
      const bool pv = h1 < H1 && w1r < W1;                             \
      cv0[q] = cbase[pv ? (size_t)h1 * W1 + w1r : 0];                  \
    }                                                                  \
  }
  CO_LOAD_COORDS()
  request_offsets(lv1 - 1);
#define CO_LOAD_A()                                                                         \
  {                                                                                         \
    CO_FRESH_LANE();                                                                        \
    _Pragma("unroll") for (int m = 0; m < CO_SB; m++) {                                     \
      int h1 = by * 4 + (lx >> 2), w1 = bx * (4 * CO_SB) + m * 4 + (lx & 3);                \
      h1 = h1 < H1 ? h1 : H1 - 1; w1 = w1 < W1 ? w1 : W1 - 1;                               \
      const T* ap = F1 + ((size_t)h1 * W1 + w1) * C + EPL * lg;                             \
      _Pragma("unroll") for (int s = 0; s < KS; s++) a[m][s] = *reinterpret_cast<const frag*>(ap + CPS * s); \
    }                                                                                       \
  }
  CO_LOAD_A()

  // (Serving the two zero-offset levels in one box / sweep / sampling pass — their 8-row patches fit one 16-row patch — was
  // built and measured: two barriers less per wave life, but 109 against 105 us; the levels stay one pass each.)
  for (int lvl = lv1 - 1; lvl >= lv0; lvl--) {
    const T* const fmap2 = p.fmap2[lvl];
    const int H2 = p.H2[lvl], W2 = p.W2[lvl];
    const float cscale = __builtin_ldexpf(1.0f, -(p.lbase + lvl));
    const T* const F2 = fmap2 + f2i * H2 * W2 * C;
    float* const obase = level_offsets(lvl);  // null = zero offsets
    const bool zo = obase == nullptr;  // workgroup-uniform

    // ---- phase 0: tap boxes of the wave's own pixels ----
    // A pixel's box is the extent of ALL its taps' corners, NOT clipped to the map (<= 16 x 16 for |offset| < 4, whatever
    // the coords): the patch then holds every corner of every tap, the ones outside the map as zeros (patches of pixels
    // whose box crosses the border are zero-filled here, before the sweep stores the in-map entries), and the sampling
    // phase needs neither range checks nor selects for the reference's per-corner zero padding (:102-117).
    {
      CO_FRESH_LANE();
      CO_STAMP(0);
      if (!zo) {  // reference side effect (:80-81): offset[centre] = 0, stored only where the bits are not +0 already
#pragma unroll
        for (int q = 0; q < CO_QP; q++) {
          const int h1 = by * 4 + qr0 + q, w1r = px0 + lg;
          constexpr int ci = CEN / 16;
          if (lx == CEN % 16 && h1 < H1 && w1r < W1) {
            if ((__builtin_bit_cast(unsigned, o0[q][ci].x) | __builtin_bit_cast(unsigned, o0[q][ci].y)) != 0u)
              reinterpret_cast<float2*>(obase + ((size_t)h1 * W1 + w1r) * NT * 2)[CEN] = make_float2(0.f, 0.f);
            o0[q][ci] = make_float2(0.f, 0.f);
          }
        }
      }
      float tfx[TI], tfy[TI];  // tap index - R as floats (lanes past the last tap repeat it: neutral for the extent)
#pragma unroll
      for (int i = 0; i < TI; i++) {
        int t = lx + 16 * i;
        t = t < NT ? t : NT - 1;
        const int ix = t / RD;
        tfx[i] = (float)(ix - R);
        tfy[i] = (float)(t - ix * RD - R);
      }
      int ulo = 0x7fff7fff, uhi = (int)0x80008000;
      bool border = false;
#pragma unroll
      for (int q = 0; q < CO_QP; q++) {
        const int h1 = by * 4 + qr0 + q, w1r = px0 + lg;
        const bool pv = h1 < H1 && w1r < W1;
        const float cx = cv0[q].x * cscale, cy = cv0[q].y * cscale;
        int lo, hi;  // packed (x, y) of the first / last column and row the taps' top-left corners reach
        if (zo) {
          const int fx = (int)floorf(cx), fy = (int)floorf(cy);
          lo = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(fx - R, fy - R));
          hi = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(fx + R, fy + R));
        } else {
          float x0 = __builtin_inff(), y0 = __builtin_inff(), x1 = -__builtin_inff(), y1 = -__builtin_inff();
#pragma unroll
          for (int i = 0; i < TI; i++) {
            const float wx = floorf(cx + o0[q][i].x) + tfx[i], wy = floorf(cy + o0[q][i].y) + tfy[i];  // :82-83, exact integers
            x0 = fminf(x0, wx); x1 = fmaxf(x1, wx);
            y0 = fminf(y0, wy); y1 = fmaxf(y1, wy);
          }
          // saturating conversion: absurd or non-finite coords give a box that fails the size test below
          lo = co_row_pk_reduce<true>(__builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16((int)x0, (int)y0)));
          hi = co_row_pk_reduce<false>(__builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16((int)x1, (int)y1)));
        }
        const int xlo = pk_lo(lo), ylo = pk_hi(lo), xhi = pk_lo(hi) + 1, yhi = pk_hi(hi) + 1;  // + 1: the right / bottom corners
        const bool any = pv && xhi > xlo && yhi > ylo;
        // a conversion that hit the int16 rails (coords beyond +-32767: points behind the camera) no longer says where the
        // taps are: such a pixel takes the per-tap fallback, whose range checks work on the full integers
        const bool sat = xlo <= -32768 || ylo <= -32768 || xhi >= 32767 || yhi >= 32767;
        const bool boxed = any && !sat && xhi - xlo < 16 && yhi - ylo < 16;
        // what the sweep has to visit: the box clipped to the map
        const int cx0 = xlo > 0 ? xlo : 0, cy0 = ylo > 0 ? ylo : 0, cx1 = xhi < W2 ? xhi : W2 - 1, cy1 = yhi < H2 ? yhi : H2 - 1;
        const bool inmap = boxed && cx0 <= cx1 && cy0 <= cy1;
        ulo = inmap ? pk_min(ulo, pk16(cx0, cy0)) : ulo;
        uhi = inmap ? pk_max(uhi, pk16(cx1, cy1)) : uhi;
        border = border || (boxed && (xlo < 0 || ylo < 0 || xhi >= W2 || yhi >= H2));
        if (lx == 0) {  // box table: (bw, bh) of a patch; (-1, 0) = box larger than a patch (per-tap fallback); (0, 0) = nothing to sample
          int* pb = pbox + (msb * 16 + (qr0 + q) * 4 + lg) * 4;
          pb[0] = xlo; pb[1] = ylo;
          pb[2] = boxed ? xhi - xlo + 1 : (any ? -1 : 0);
          pb[3] = boxed ? yhi - ylo + 1 : 0;
        }
      }
      if (__builtin_amdgcn_ballot_w64(border) != 0) {  // wave-uniform: zero the patches of the wave's own pixels
        float4* z = reinterpret_cast<float4*>(patch + (msb * 16 + qr0 * 4) * CO_PP);
        constexpr int NZ = 4 * CO_QP * CO_PP / 4;  // float4s
#pragma unroll
        for (int k = 0; k < (NZ + kWave - 1) / kWave; k++)
          if (k * kWave + lane < NZ) z[k * kWave + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      ulo = pk_min(pk_min(__builtin_amdgcn_readlane(ulo, 0), __builtin_amdgcn_readlane(ulo, 16)),
                   pk_min(__builtin_amdgcn_readlane(ulo, 32), __builtin_amdgcn_readlane(ulo, 48)));
      uhi = pk_max(pk_max(__builtin_amdgcn_readlane(uhi, 0), __builtin_amdgcn_readlane(uhi, 16)),
                   pk_max(__builtin_amdgcn_readlane(uhi, 32), __builtin_amdgcn_readlane(uhi, 48)));
      if (lane == 0) { swin[wv * 2] = ulo; swin[wv * 2 + 1] = uhi; }
      CO_STAMP(1);
    }
    __syncthreads();

    // ---- phase 1: the tile window, its rows dealt to the four waves; every fragment meets all four sub-blocks ----
    {
      CO_FRESH_LANE();
      int tlo = 0x7fff7fff, thi = (int)0x80008000;  // the tile window = union of the sub-block windows (empty ones are neutral)
#pragma unroll
      for (int m = 0; m < CO_NW; m++) {
        tlo = pk_min(tlo, swin[m * 2]);
        thi = pk_max(thi, swin[m * 2 + 1]);
      }
      tlo = __builtin_amdgcn_readfirstlane(tlo); thi = __builtin_amdgcn_readfirstlane(thi);
      int TX0 = pk_lo(tlo);
      const int TY0 = pk_hi(tlo), TX1 = pk_lo(thi), TY1 = pk_hi(thi);
      if (TX1 >= TX0 && TY1 >= TY0) {
        TX0 &= ~CO_XALIGN;  // pairs of positions start at even columns
        // per sub-block m, this lane's pixel (m, lx): byte address in LDS of the patch entry of (row 0 of the map, column
        // 4 lg of group 0), and the row / column ranges a store must fall in
        float* sbp[CO_SB];
        int sylo[CO_SB], sbh[CO_SB], sq0[CO_SB], swl[CO_SB];
#pragma unroll
        for (int m = 0; m < CO_SB; m++) {
          const int4 pb = *reinterpret_cast<const int4*>(pbox + (m * 16 + lx) * 4);
          const int xal = pb.x & ~1;
          sylo[m] = pb.y;
          sbh[m] = pb.w;                    // 0 = no patch: no row passes
          swl[m] = pb.x + pb.z - 1 - xal;   // last patch column a pair may start at or before
          sq0[m] = TX0 + 4 * lg - xal;      // patch column of this lane's first pair in group 0
          sbp[m] = patch + __mul24(m * 16 + lx, CO_PP) - __mul24(pb.y, CO_BOXP) + sq0[m];
        }
        const int ngx = (TX1 - TX0 + 16) >> 4;
        const int y0w = TY0 + wv;
        const int nrow = y0w <= TY1 ? ((TY1 - y0w) / CO_NW) + 1 : 0;
        const int nit = ngx * nrow;
        // fragment addresses: uniform base per k-step (SGPR pair) + a 32-bit lane offset (host-checked: a level has
        // < 2^31 elements).  Element strides of a position / a map row / a lane group / a k-step in the two storage forms:
        const unsigned PSTR = p.f2_chunked ? EPL : C, YSTR = (unsigned)W2 * PSTR;
        const unsigned GSTR = p.f2_chunked ? (unsigned)H2 * W2 * EPL : EPL;
        const size_t KSTR = p.f2_chunked ? (size_t)4 * H2 * W2 * EPL : (size_t)CPS;
        auto col_off = [&](int gx) {  // lane byte offset of row 0 in the column of groups starting at gx
          int x = gx + lx;
          x = x < W2 ? x : W2 - 1;  // padded positions re-read the last column; their results land in no box column that is read
          return ((unsigned)lg * GSTR + (unsigned)x * PSTR) * (unsigned)sizeof(T);
        };
        auto load_group = [&](frag (&dstf)[KS], unsigned coff, int y) __attribute__((always_inline)) {
          const unsigned off = coff + (unsigned)y * (YSTR * (unsigned)sizeof(T));  // bytes; the row term is scalar
#pragma unroll
          for (int s = 0; s < KS; s++)
            dstf[s] = *reinterpret_cast<const frag*>(reinterpret_cast<const char*>(F2 + KSTR * s) + (size_t)off);
        };
        if (nit > 0) {
          // The loop body is branch-free around its loads (a branch around a load makes hipcc wait for nearly every
          // outstanding load at each use): the trip count is rounded up to a multiple of PF and the load cursor stops at
          // the last group, which the surplus steps load and store again.  Slot j remembers which group it holds.
          frag bq[PF][KS];
          int ys[PF], gqs[PF];              // row and column (relative to TX0) of the group in slot j (scalar)
          int yl = y0w, gql = 0, li = 0;    // load cursor
          unsigned cofl = col_off(TX0);
          auto issue = [&](int j) __attribute__((always_inline)) {
            load_group(bq[j], cofl, yl);
            ys[j] = yl; gqs[j] = gql;
            if (li < nit - 1) {
              li++;
              yl += CO_NW;
              if (yl > TY1) { yl = y0w; gql += 16; cofl = col_off(TX0 + gql); }
            }
          };
#pragma unroll
          for (int j = 0; j < PF; j++) issue(j);
          CO_STAMP(2);
          // Column masks (wave-uniform 64-bit values, recomputed when the column of groups changes): which lanes' first /
          // second pair of positions falls into the columns of their pixel's box.  A store then costs one row compare.
          // Positions right of the map (the last column of groups re-reads column W2 - 1 for them) are not stored: their
          // patch entries keep the zeros of phase 0.  A pair straddling the edge (odd W2) stores a zero as its second half.
          unsigned long long colA[CO_SB], colB[CO_SB];
          bool strad1 = false, strad3 = false;  // this lane's position 1 / 3 is the first one right of the map
          bool strad_any = false;
          int curq = -1;
          for (int it = 0; it < nit; it += PF) {
#pragma unroll
            for (int j = 0; j < PF; j++) {
              cof32x4 d[CO_SB];
#pragma unroll
              for (int m = 0; m < CO_SB; m++) d[m] = cof32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int s = 0; s < KS; s++)
#pragma unroll
                for (int m = 0; m < CO_SB; m++) d[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[j][s], a[m][s], d[m], 0, 0, 0);
              const int y = ys[j], gq = gqs[j];
              issue(j);
              if (gq != curq) {  // wave-uniform
                curq = gq;
                const int xa = TX0 + gq + 4 * lg;  // map column of this lane's first position
#pragma unroll
                for (int m = 0; m < CO_SB; m++) {
                  const int qx = sq0[m] + gq;
                  colA[m] = __builtin_amdgcn_ballot_w64((unsigned)qx <= (unsigned)swl[m] && xa < W2);
                  colB[m] = __builtin_amdgcn_ballot_w64((unsigned)(qx + 2) <= (unsigned)swl[m] && xa + 2 < W2);
                }
                strad1 = xa + 1 == W2; strad3 = xa + 3 == W2;
                strad_any = __builtin_amdgcn_ballot_w64(strad1 || strad3) != 0;  // scalar, once per column of groups
              }
              if (strad_any) {  // wave-uniform; only in the last column of groups of an odd-width map
                asm volatile("");  // a real (scalar) branch: if-converted, the eight selects run in every step of every map
#pragma unroll
                for (int m = 0; m < CO_SB; m++) {
                  d[m][1] = strad1 ? 0.f : d[m][1];
                  d[m][3] = strad3 ? 0.f : d[m][3];
                }
              }
              const int yg = y * CO_BOXP + gq;  // scalar
#pragma unroll
              for (int m = 0; m < CO_SB; m++) {
                const unsigned long long rowm = __builtin_amdgcn_ballot_w64((unsigned)(y - sylo[m]) < (unsigned)sbh[m]);
                float* dst = sbp[m] + yg;
                if (__builtin_amdgcn_inverse_ballot_w64(rowm & colA[m])) *reinterpret_cast<float2*>(dst) = make_float2(d[m][0], d[m][1]);
                if (__builtin_amdgcn_inverse_ballot_w64(rowm & colB[m])) *reinterpret_cast<float2*>(dst + 2) = make_float2(d[m][2], d[m][3]);
              }
            }
          }
        }
      }
      CO_STAMP(3);
    }
    __syncthreads();

    // ---- phase 2: sample the patches of the wave's own pixels ----
    // Straight-line for the pixels that have a patch (all of them, in practice): every patch read of a pass is issued
    // before the first blend.  A pixel WITHOUT a patch — outside the image, non-finite coords, or a box larger than a
    // patch (per-tap fallback) — reads its own patch's first entries instead (a legal address) and is put right
    // afterwards, behind a wave-uniform branch that is not taken unless some lane of the wave is such a pixel: left inside
    // the tap loop, that rare path turns every tap into a chain of exec-mask branches with a wait behind each read.
    float res[CO_QP][TI];
    {
      CO_FRESH_LANE();
      int trel[TI];  // (tap row - R) * pitch + (tap column - R); lanes past the last tap repeat it (their results are not stored)
#pragma unroll
      for (int i = 0; i < TI; i++) {
        int t = lx + 16 * i;
        t = t < NT ? t : NT - 1;
        const int ix = t / RD;
        trel[i] = (t - ix * RD - R) * CO_BOXP + (ix - R);
      }
#pragma unroll
      for (int q = 0; q < CO_QP; q++) {
        const int h1 = by * 4 + qr0 + q, w1r = px0 + lg;
        const int pixl = msb * 16 + (qr0 + q) * 4 + lg;
        const int4 pb = *reinterpret_cast<const int4*>(pbox + pixl * 4);  // row-uniform
        const bool has_patch = pb.w != 0, fallback = pb.z < 0;
        const bool odd = __builtin_amdgcn_ballot_w64(!has_patch) != 0;  // wave-uniform
        // patch entry of map position (0, 0): every corner of every tap lies inside the patch (phase 0)
        const int d0 = __mul24(pixl, CO_PP) - __mul24(pb.y, CO_BOXP) - (pb.x & ~1);  // 24-bit products: full rate
        const float cx = cv0[q].x * cscale, cy = cv0[q].y * cscale;
        float q11[TI], q21[TI], q12[TI], q22[TI];  // all reads of the pass in flight before the first blend
        if (zo) {
          // one sample position per pixel: floor, fraction and the four weights once per pass (products and order are bilerp()'s)
          const float zfx = floorf(cx), zfy = floorf(cy);
          const float zdx = cx - zfx, zdy = cy - zfy;
          const float zw11 = (1.0f - zdy) * (1.0f - zdx), zw21 = (1.0f - zdy) * zdx, zw12 = zdy * (1.0f - zdx), zw22 = zdy * zdx;
          const int dz = has_patch ? d0 + __mul24((int)zfy, CO_BOXP) + (int)zfx : __mul24(pixl, CO_PP) + R * CO_BOXP + R;  // one select per pass
#pragma unroll
          for (int i = 0; i < TI; i++) {
            const float* D = patch + dz + trel[i];
            q11[i] = D[0]; q21[i] = D[1]; q12[i] = D[CO_BOXP]; q22[i] = D[CO_BOXP + 1];
          }
#pragma unroll
          for (int i = 0; i < TI; i++) res[q][i] = q11[i] * zw11 + q21[i] * zw21 + q12[i] * zw12 + q22[i] * zw22;
          if (odd) {
            asm volatile("");  // a real (scalar) branch
#pragma unroll
            for (int i = 0; i < TI; i++) res[q][i] = has_patch ? res[q][i] : 0.f;
          }
        } else {
          float dxs[TI], dys[TI];
#pragma unroll
          for (int i = 0; i < TI; i++) {
            const float xs = cx + o0[q][i].x, ys = cy + o0[q][i].y;
            const float fxs = floorf(xs), fys = floorf(ys);
            dxs[i] = xs - fxs; dys[i] = ys - fys;  // :87-88
            const int rel = __mul24((int)fys, CO_BOXP) + (int)fxs + trel[i];
            const float* D = patch + (has_patch ? d0 + rel : __mul24(pixl, CO_PP));
            q11[i] = D[0]; q21[i] = D[1]; q12[i] = D[CO_BOXP]; q22[i] = D[CO_BOXP + 1];
          }
          if (odd) {
            asm volatile("");  // a real (scalar) branch: the rare pixels of this pass
#pragma unroll
            for (int i = 0; i < TI; i++) {
              if (!has_patch) {
                q11[i] = q21[i] = q12[i] = q22[i] = 0.f;
                if (fallback) {  // box larger than a patch: this tap's four corner dots, channels in order, per-corner zero padding
                  int t = lx + 16 * i;
                  t = t < NT ? t : NT - 1;
                  const int ix = t / RD;
                  const int w2 = (int)floorf(cx + o0[q][i].x) - R + ix, h2 = (int)floorf(cy + o0[q][i].y) - R + (t - ix * RD);
                  const bool bx0 = (unsigned)w2 < (unsigned)W2, bx1 = (unsigned)(w2 + 1) < (unsigned)W2;
                  const bool by0 = (unsigned)h2 < (unsigned)H2, by1 = (unsigned)(h2 + 1) < (unsigned)H2;
                  const float4 qq = co_corner_dots(F1 + ((size_t)h1 * W1 + w1r) * C, F2, (ptrdiff_t)h2 * W2 + w2, C, W2,
                                                   (by0 && bx0 ? 1 : 0) | (by0 && bx1 ? 2 : 0) | (by1 && bx0 ? 4 : 0) | (by1 && bx1 ? 8 : 0),
                                                   p.f2_chunked ? EPL : C, p.f2_chunked ? (ptrdiff_t)H2 * W2 * EPL : EPL);
                  q11[i] = qq.x; q21[i] = qq.y; q12[i] = qq.z; q22[i] = qq.w;
                }
              }
            }
          }
#pragma unroll
          for (int i = 0; i < TI; i++) res[q][i] = bilerp(q11[i], q21[i], q12[i], q22[i], dxs[i], dys[i]);  // :114-117
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      CO_STAMP(4);
    }
    request_offsets(lvl - 1);  // this level's offsets are consumed: the next level's travel during the write-out

    // ---- write-out: corr[b][n][ix][iy][h1][w1]; the wave's own patches become its [tap][pixel] transpose tile ----
    {
      CO_FRESH_LANE();
      float* const outt = patch + (msb * 16 + qr0 * 4) * CO_PP;  // the patches of the wave's own pixels
#pragma unroll
      for (int q = 0; q < CO_QP; q++)
#pragma unroll
        for (int i = 0; i < TI; i++)
          if (lx + 16 * i < NT) outt[(lx + 16 * i) * CO_OUTP + q * 4 + lg] = res[q][i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float* const cb = p.corr + (((size_t)b * S + n) * p.Ltot + p.lvl0 + lvl) * NT * HW1;
      if (p.vec_out) {
        // NT x 2 (tap, row) items of 4 pixels each (98 at radius 3): lane -> tap lane / 2 (and 32 taps on: a scalar step of the base), row lane % 2;
        // one 32-bit lane offset serves both (host-checked: a level's output is below 4 GB)
        static_assert(CO_QP == 2, "item layout");
        const int t0 = lane >> 1, qq = lane & 1;
        const int h1 = by * 4 + qr0 + qq;
        const unsigned loff = ((unsigned)(t0 * H1 + h1) * (unsigned)W1 + (unsigned)px0) * 4u;
        const float* const src = outt + t0 * CO_OUTP + qq * 4;
        if (h1 < H1 && px0 < W1 && t0 < NT) {   // (t0 < NT: radius 1 / 2 have 9 / 25 taps, fewer than the 32 of the first item)
          const float4 v0 = *reinterpret_cast<const float4*>(src);
          *reinterpret_cast<float4*>(reinterpret_cast<char*>(cb) + (size_t)loff) = v0;
          if (t0 + 32 < NT) {
            const float4 v1 = *reinterpret_cast<const float4*>(src + 32 * CO_OUTP);
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(cb + 32 * HW1) + (size_t)loff) = v1;
          }
        }
      } else {   // W1 % 4 != 0 or an unaligned output: element stores with their own range checks
        for (int idx = lane; idx < NT * CO_QP; idx += kWave) {
          const int t = idx / CO_QP, q = idx - t * CO_QP;
          const int h1 = by * 4 + qr0 + q, w1 = px0;
          if (h1 >= H1 || w1 >= W1) continue;
          const float4 v = *reinterpret_cast<const float4*>(outt + t * CO_OUTP + q * 4);
          float* dst = cb + ((size_t)t * H1 + h1) * W1 + w1;
          dst[0] = v.x;
          if (w1 + 1 < W1) dst[1] = v.y;
          if (w1 + 2 < W1) dst[2] = v.z;
          if (w1 + 3 < W1) dst[3] = v.w;
        }
      }
      CO_STAMP(5);
    }
  }
}
#undef CO_FRESH_LANE

template <int R, int KS>
static int launch_coop(CoParams p, hipStream_t st) {
  auto kern = lowmem_coop_kernel<R, KS>;
  // LGU_LOWMEM_COOP_LDS_PAD (debug / diagnosis only): extra dynamic LDS in KB per workgroup, clamped to the 160 KB of a CU —
  // fewer resident workgroups per CU, to read a workgroup's life against the number of workgroups sharing the CU
  size_t lds = sizeof(float) * (size_t)CO_LDS_FLOATS;
  {
    const int pad_kb = env_int("LGU_LOWMEM_COOP_LDS_PAD", 0);
    if (pad_kb > 0) lds = lds + (size_t)pad_kb * 1024 <= 160 * 1024 ? lds + (size_t)pad_kb * 1024 : 160 * 1024;
  }
  allow_max_dynamic_lds<&lowmem_coop_kernel<R, KS>>();
  p.tiles_x = (p.W1 + 4 * CO_SB - 1) / (4 * CO_SB);
  p.tiles_y = (p.H1 + 3) / 4;
  const int tiles = p.tiles_x * p.tiles_y;
  p.xcd_map = p.B >= 8 ? 1 : 0;
  // level groups of the split items: every level with offsets is a unit of its own, a run of zero-offset levels is one
  p.ngroups = 0;
  for (int l = 0; l < p.L;) {
    p.gl0[p.ngroups++] = l;
    if (p.offset[l]) l++;
    else while (l < p.L && !p.offset[l]) l++;
  }
  for (int k = p.ngroups; k <= CO_MAXL; k++) p.gl0[k] = p.L;
  // Fused workgroups in whole rounds over the device's slots (4 resident workgroups per CU), the remainder split by
  // level group so that the last round is made of short units: a call of 2.3 rounds costs ~2.4 instead of 3.
  // LGU_LOWMEM_COOP_SPLIT (debug / A-B only): 0 = all fused, 1 = all split, default = the rule above.
  const int slots = CO_WPS * device_cu_count();
  const size_t items = p.xcd_map ? (size_t)((p.B + 7) / 8) * 8 * tiles : (size_t)p.B * tiles;
  if (items >= (1u << 30)) return -1;
  const int mode = env_int("LGU_LOWMEM_COOP_SPLIT", -1);
  size_t fused = items;
  if (p.ngroups > 1) {
    if (mode == 1) fused = 0;
    else if (mode != 0 && items > (size_t)slots && items < (size_t)8 * slots) fused = items / slots * slots;
  }
  p.n_fused = (int)fused;
  p.n_split = (int)(items - fused);
  if (p.n_split == 0) p.n_split = 1;  // never divided by when unused
  const size_t nwg = fused + (items - fused) * p.ngroups;
  if (nwg >= (1u << 31)) return -1;
  p.vec_out = (p.W1 % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.corr) & 15) == 0);
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg, (unsigned)p.S), dim3(kWave * CO_NW), lds, st, p);
  return launch_status();
}

// Serves half feature maps with C in {32, 64, 128} and radius 1..3; returns -1 otherwise (the caller then takes the
// one-wave-per-block kernel of lowmem_mfma.hip).  LGU_LOWMEM_COOP=0 (debug / A-B only) disables it.
int lowmem_coop_dispatch(const void* fmap1, const void* const* fmap2, float* const* offset, const float* coords, float* corr,
                         const int* H2, const int* W2, int L, int B, int S, int H1, int W1, int C, int radius, int lbase,
                         int lvl0, int Ltot, int f2_chunked, const long long* ii, const long long* jj, const int* orow, int n_orow,
                         hipStream_t st) {
  if (env_int("LGU_LOWMEM_COOP", 1) == 0) return -1;
  if (orow && (S != 1 || n_orow < 1)) return -1;
  if (L < 1 || L > CO_MAXL || radius < 1 || radius > 3 || S > 65535) return -1;
  // a single zero-offset level (altcorr_forward; the r = 1 probe of AltCorrBlock) has little window to share and no
  // offsets to wait for: the independent waves of the one-wave kernel serve it faster (probe: 18 against 25 us)
  if (L == 1 && !offset[0] && env_int("LGU_LOWMEM_COOP_SINGLE", 0) == 0) return -1;
  uintptr_t al = reinterpret_cast<uintptr_t>(fmap1);
  for (int l = 0; l < L; l++) al |= reinterpret_cast<uintptr_t>(fmap2[l]);
  if ((al & 15) != 0) return -1;
  if ((size_t)H1 * W1 * C >= (1u << 31) || (size_t)H1 * W1 * 49 * 8 >= (1ull << 32)) return -1;
  CoParams p = {};
  p.fmap1 = static_cast<const _Float16*>(fmap1);
  for (int l = 0; l < L; l++) {
    if ((size_t)H2[l] * W2[l] * C >= (1u << 31) || H2[l] > 32767 || W2[l] > 32767) return -1;
    p.fmap2[l] = static_cast<const _Float16*>(fmap2[l]);
    p.offset[l] = offset[l];
    p.H2[l] = H2[l]; p.W2[l] = W2[l];
  }
  p.coords = coords; p.corr = corr;
  p.L = L; p.B = B; p.S = S; p.H1 = H1; p.W1 = W1;
  p.lbase = lbase; p.lvl0 = lvl0; p.Ltot = Ltot; p.f2_chunked = f2_chunked; p.ii = ii; p.jj = jj;
  p.orow = orow; p.n_orow = n_orow;
#define LGU_CO_CASE(RV, KSV) \
  if (radius == RV && C == 32 * KSV) return launch_coop<RV, KSV>(p, st);
  LGU_CO_CASE(3, 4) LGU_CO_CASE(1, 4) LGU_CO_CASE(2, 4)
  LGU_CO_CASE(3, 2) LGU_CO_CASE(1, 2) LGU_CO_CASE(2, 2)
  LGU_CO_CASE(3, 1) LGU_CO_CASE(1, 1) LGU_CO_CASE(2, 1)
#undef LGU_CO_CASE
  return -1;
}

}  // namespace lgu

#ifdef LGU_MM_STAMPS
// Diagnostic build only (tools/diag/run_co_stamps.py): never part of liblgu_corr.so.
extern "C" {
int lgu_co_diag_set_stamps(void* q) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(lgu::g_co_stamps), &q, sizeof(q)); }
int lgu_co_diag_pyramid(const void* fmap1, const void* const* fmap2, float* const* offset, const float* coords, float* corr,
                        const int* H2, const int* W2, int L, int B, int S, int H1, int W1, int C, int radius, int chunked) {
  return lgu::lowmem_coop_dispatch(fmap1, fmap2, offset, coords, corr, H2, W2, L, B, S, H1, W1, C, radius, 0, 0, L, chunked,
                                   nullptr, nullptr, nullptr, 0, nullptr);
}
}
#endif
