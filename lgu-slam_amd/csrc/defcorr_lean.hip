// defcorr_lean.hip — the production form of the fused pyramid sampler, written for instruction count.
//
// Same operator as csrc/defcorr.hip's defcorr_gather_kernel (reference, relative to /root/reference:
// offersample_LGS/defCorrSample_kernel.cu:25-91 per level, offersample_LGS/corrSample_kernel.cu:24-82 for the
// zero-offset levels and the probe, droid_slam/modules/corr.py:88-109 for the composition) and bit-identical to it,
// for the one configuration CorrBlock launches: radius 3, four levels, learned offsets on levels 0 and 1, structurally
// zero offsets on levels 2 and 3 (corr.py:132-135), planar fp32 output.  Everything else stays with defcorr.hip.
//
// Why a second kernel: profiles/r02_pmc_cold.txt — the general kernel executes 709 vector + 521 scalar instructions
// per wave for 24 memory reads; a wave spends 35 % of its life waiting for an issue slot and 22 % issuing.  Here
//   * every uniform quantity is scalar: coords arrive by s_load, slice / offset bases are SGPR pairs and every
//     vector memory access is  SGPR base + 32-bit lane offset (+ immediate), so there is no 64-bit vector address
//     arithmetic and one pixel's second neighbour costs an immediate, not an add;
//   * validity / edge masks live in SGPR pairs from the compare to the select (no pack / unpack through a VGPR);
//   * corners come as two 8-byte x-pairs per tap (defcorr.hip, PAIR) — the right neighbour across a tile boundary is
//     an exec-masked load into a register that otherwise holds the 0 the reference pads with;
//   * level geometry is compile-time where CorrBlock fixes it (levels halve, tiles are 4 x 8) and kernarg otherwise;
//   * the write-out advances an SGPR base per channel group: no per-element address arithmetic.
// Decomposition is unchanged: workgroup = 16 x-adjacent pixels of one row (8 waves), wave = 2 pixels, lanes = taps for
// the offset levels and lattice points for the zero-offset levels, LDS transpose tile, 64-byte row segments, tiles dealt
// to XCDs in contiguous runs.
#include "lgu_common.hpp"

namespace lgu {

namespace lean {

constexpr int R = 3, RD = 7, NT = 49, NL = 4, GP = 2, TPX = 16, NWV = TPX / GP;
constexpr int PITCH = TPX + 1;            // transpose tile pitch (floats)
constexpr int CH = NL * NT;               // 196 output channels

struct Geo {
  int H2[NL], W2[NL];   // logical slice sizes per level
  int ssz[NL];          // floats per (edge, pixel) slice as stored
  int tpr[NL];          // tiled layout: 4 x 8 tiles per tile row
};

typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x2a __attribute__((ext_vector_type(2)));
typedef float f32x4a __attribute__((ext_vector_type(4)));

// element position inside a slice, in floats (separable: fy(y) + fx(x))
template <bool TILED>
__device__ __forceinline__ unsigned pos_x(unsigned x) { return TILED ? x + __umul24(x >> 3, 24u) : x; }
template <bool TILED>
__device__ __forceinline__ unsigned pos_y(unsigned y, unsigned W2, unsigned tpr) {
  return TILED ? (y << 3) + __umul24(y >> 2, tpr * 32u - 32u) : __umul24(y, W2);   // slices are far below 2^24 elements
}

// clamp to [0, hi] in one instruction (hi wave-uniform)
__device__ __forceinline__ unsigned clamp0(int v, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "s"(hi));
  return (unsigned)r;
}
__device__ __forceinline__ float uniform(float v) {   // a wave-uniform value computed on the vector ALU, moved to an SGPR
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

template <class T>
__device__ __forceinline__ T ldg(const float* base, unsigned byte_off) {
  // SGPR base + zero-extended 32-bit lane offset: the saddr form of global_load
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)byte_off);
}

// Launch bound: 6 waves per SIMD = 3 workgroups per CU at <= 80 VGPRs.  The live set is 76 (the in-flight corners of
// 2 pixels x 2 levels, their fractions, the lattice values, the offsets); capped at 64 the allocator spills loaded
// values back to back with the loads (75.7 us).  Measured with LDS padding: a fourth resident workgroup buys the
// general kernel 3 % (profiles/r02_ab_occupancy.jsonl), the third buys this one 4 %.
template <bool PROBE, bool TILED>
__global__ __launch_bounds__(NWV* kWave, 6) void defcorr_lean_kernel(
    const float* __restrict__ v0, const float* __restrict__ v1, const float* __restrict__ v2, const float* __restrict__ v3,
    float* __restrict__ off0, float* __restrict__ off1, const float* __restrict__ coords, float* __restrict__ out,
    const int* __restrict__ edge_slot, const Geo g, const int H1, const int W1, const int tiles_per_row,
    const unsigned magic_tiles, const unsigned magic_h1, const int coords_last, const int xcd_remap) {
  extern __shared__ float4 lds4[];
  float* const outst = reinterpret_cast<float*>(lds4);  // [CH][PITCH]

  const unsigned lane = threadIdx.x & (kWave - 1);
  const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned bid = blockIdx.x;
  if (xcd_remap) {  // see defcorr.hip: contiguous runs of tiles per XCD, so that half-line writes merge in one L2
    const unsigned n8 = gridDim.x & ~7u;
    if (bid < n8) bid = (bid & 7u) * (n8 >> 3) + (bid >> 3);
  }
  // divisions by magic multiplication (exact in the host-checked range; 0 = divisor 1)
  const unsigned row = magic_tiles ? __umulhi(bid, magic_tiles) : bid;   // = bid / tiles_per_row = e * H1 + y
  const unsigned tile = bid - row * tiles_per_row;
  const unsigned e = magic_h1 ? __umulhi(row, magic_h1) : row;           // = row / H1
  const unsigned y = row - e * H1;
  const unsigned xbase = tile * TPX;
  const unsigned px0 = xbase + w * GP;                    // first of this wave's two pixels
  const unsigned W1m1 = W1 - 1;
  // pixels beyond the row end (W1 % 16 != 0) are computed on the last pixel's data and never stored
  const unsigned pxa = px0 < W1m1 ? px0 : W1m1, pxb = px0 + 1 < W1m1 ? px0 + 1 : W1m1;
  const bool pv[GP] = {px0 < (unsigned)W1, px0 + 1 < (unsigned)W1};   // stores of a clamped (duplicate) pixel are suppressed
  const unsigned row_pix = row * W1;
  const unsigned vrow_pix = edge_slot ? ((unsigned)edge_slot[e] * H1 + y) * W1 : row_pix;

  // lane roles
  const unsigned ti = (lane * 37u) >> 8, tj = lane - ti * 7u;     // tap (i moves in x, j in y), lane < 49
  const int tix = (int)ti - R, tjy = (int)tj - R;
  const bool tap = lane < NT;
  const unsigned lane_c = tap ? lane : NT - 1;                    // clamped: lanes 49..63 re-read tap 48 (discarded)
  const int lxm = (int)(lane & 7u) - R, lym = (int)(lane >> 3) - R;  // 8 x 8 lattice point of this lane
  const unsigned lsrc4 = (tj * 8u + ti) * 4u;                     // byte address (ds_bpermute) of the tap's top-left lattice lane

  // ---- phase 0: this wave's offsets (two pixels x two levels, 392 B per pixel-level: the second pixel is an
  // immediate further) and coords (scalar) ----
  const float* ob0 = off0 + (size_t)(row_pix + pxa) * (NT * 2);
  const float* ob1 = off1 + (size_t)(row_pix + pxa) * (NT * 2);
  const unsigned dpx = (pxb - pxa) * (NT * 2 * 4);                // 392 or 0 bytes (uniform)
  f32x2a of[GP][2];
  of[0][0] = ldg<f32x2a>(ob0, lane_c * 8u);
  of[1][0] = ldg<f32x2a>(ob0, lane_c * 8u + dpx);
  of[0][1] = ldg<f32x2a>(ob1, lane_c * 8u);
  of[1][1] = ldg<f32x2a>(ob1, lane_c * 8u + dpx);
  float cx[GP], cy[GP];
  {
    const unsigned HW1 = (unsigned)H1 * W1;
    if (coords_last) {   // (E,H1,W1,2)
      const float* c = coords + (size_t)(row_pix + pxa) * 2;
      cx[0] = c[0]; cy[0] = c[1];
      const float* d = coords + (size_t)(row_pix + pxb) * 2;
      cx[1] = d[0]; cy[1] = d[1];
    } else {             // (E,2,H1,W1)
      const float* c = coords + (size_t)e * 2 * HW1 + (size_t)y * W1;
      cx[0] = c[pxa]; cx[1] = c[pxb];
      cy[0] = c[HW1 + pxa]; cy[1] = c[HW1 + pxb];
    }
  }
  // reference side effect (defCorrSample_kernel.cu:51-52): the centre tap's offset is 0 in memory and in use.  The
  // tensor persists over the 8-16 lookups of a volume: stored only while its bits are not already +0.
  const bool centre = lane == (R * RD + R);
  {
    unsigned nz = 0;
#pragma unroll
    for (int k = 0; k < GP; k++)
#pragma unroll
      for (int l = 0; l < 2; l++) nz |= __builtin_bit_cast(unsigned, of[k][l].x) | __builtin_bit_cast(unsigned, of[k][l].y);
    if (centre && nz != 0u) {   // first lookup of a volume only
      const f32x2a z = {0.0f, 0.0f};
      if (pv[0]) {
        reinterpret_cast<f32x2a*>(off0 + (size_t)(row_pix + pxa) * (NT * 2))[R * RD + R] = z;
        reinterpret_cast<f32x2a*>(off1 + (size_t)(row_pix + pxa) * (NT * 2))[R * RD + R] = z;
      }
      if (pv[1]) {
        reinterpret_cast<f32x2a*>(off0 + (size_t)(row_pix + pxb) * (NT * 2))[R * RD + R] = z;
        reinterpret_cast<f32x2a*>(off1 + (size_t)(row_pix + pxb) * (NT * 2))[R * RD + R] = z;
      }
    }
#pragma unroll
    for (int k = 0; k < GP; k++)
#pragma unroll
      for (int l = 0; l < 2; l++) {
        of[k][l].x = centre ? 0.0f : of[k][l].x;
        of[k][l].y = centre ? 0.0f : of[k][l].y;
      }
  }

  const float* const vb[NL] = {v0, v1, v2, v3};

  // ---- probe lattices (level 1, 4 x 4 around coords / 2): pixel 0 on lanes 0..15, pixel 1 on lanes 16..31 — one
  // register and one load for both pixels, issued first: level 1 waits for it ----
  float platv = 0.0f;
  const bool pr1 = (lane & 16u) != 0u;                    // row 0 / row 1 of the wave = this wave's first / second pixel
  float psx = 0.0f, psy = 0.0f;
  if constexpr (PROBE) {
    psx = (pr1 ? cx[1] : cx[0]) * 0.5f;
    psy = (pr1 ? cy[1] : cy[0]) * 0.5f;
    const int X = (int)floorf(psx) - 1 + (int)(lane & 3u), Y = (int)floorf(psy) - 1 + (int)((lane >> 2) & 3u);
    const float* sl = vb[1] + (size_t)(vrow_pix + pxa) * (unsigned)g.ssz[1];
    const unsigned dsl = pr1 ? (pxb - pxa) * (unsigned)g.ssz[1] * 4u : 0u;   // the second pixel's slice: 0 or one slice on
    if (lane < 32 && (unsigned)X < (unsigned)g.W2[1] && (unsigned)Y < (unsigned)g.H2[1])
      platv = ldg<float>(sl, (pos_y<TILED>(Y, g.W2[1], g.tpr[1]) + pos_x<TILED>(X)) * 4u + dsl);
  }

  // ---- phase A: issue every load before any result is used ----
  // offset levels: taps
  f32x2u qa[GP][2], qb[GP][2];      // top / bottom x-pair
  float qc[GP][2], qd[GP][2];       // right neighbours across a tile boundary (tiled), else the reference's 0 padding
  float gdx[GP][2], gdy[GP][2];
  bool gvalid[GP][2], gshift[GP][2], gyin[GP][2];

  auto issue_offset_level = [&](int k, int l) __attribute__((always_inline)) {
    const int H2 = g.H2[l], W2 = g.W2[l];
    const float sc = l ? 0.5f : 1.0f;
    const float ofsX = of[k][l].x + cx[k] * sc, ofsY = of[k][l].y + cy[k] * sc;     // :56-57 (coords / 2^l is exact)
    const int fx = (int)floorf(ofsX), fy = (int)floorf(ofsY);
    gdx[k][l] = ofsX - (float)fx;                                                     // :60-61
    gdy[k][l] = ofsY - (float)fy;
    const int x1 = fx + tix, y1 = fy + tjy;                                           // :63-66
    gvalid[k][l] = tap && (unsigned)x1 < (unsigned)W2 && (unsigned)y1 < (unsigned)H2; // :67 whole-tap rule
    // clamped so that every lane forms a legal address; clamped lanes are discarded in phase B
    const unsigned xc = clamp0(x1, W2 - 1), yc = clamp0(y1, H2 - 1);
    const unsigned yb = yc + 1 < (unsigned)H2 ? yc + 1 : yc;
    gyin[k][l] = y1 + 1 < H2;
    // pair start: one element to the left in the last column of a tile (tiled) / of the slice (row-major).  With
    // W2 % 8 == 0 (host-checked) a valid tap whose right neighbour is outside the slice is always such a lane.
    const bool shift = TILED ? ((xc & 7u) == 7u) : (xc == (unsigned)(W2 - 1));
    gshift[k][l] = shift;
    const unsigned xa = xc - (shift ? 1u : 0u);
    const unsigned px_ = pos_x<TILED>(xa);
    const unsigned pa = (pos_y<TILED>(yc, W2, g.tpr[l]) + px_) * 4u, pb = (pos_y<TILED>(yb, W2, g.tpr[l]) + px_) * 4u;
    const float* sl = vb[l] + (size_t)(vrow_pix + (k ? pxb : pxa)) * (unsigned)g.ssz[l];
    qa[k][l] = ldg<f32x2u>(sl, pa);
    qb[k][l] = ldg<f32x2u>(sl, pb);
    qc[k][l] = qd[k][l] = 0.0f;
    if constexpr (TILED) {
      if (shift && x1 + 1 < W2) {  // first column of the next tile: 26 floats after the pair
        qc[k][l] = ldg<float>(sl, pa + 104u);
        qd[k][l] = ldg<float>(sl, pb + 104u);
      }
    }
  };

  // zero-offset levels: all taps share one fractional part and sit on an 8 x 8 integer lattice = one wave
  float latv[GP][2];
  bool latin[GP][2];
  int lfx[GP][2], lfy[GP][2];       // floor of the level coords (SGPRs)
  float lw[GP][2][4];               // the four bilinear weights, common to all taps of the pixel-level (SGPRs)
  auto issue_lattice_level = [&](int k, int l) __attribute__((always_inline)) {   // l = 2, 3
    const float sc = l == 2 ? 0.25f : 0.125f;
    const float sx = cx[k] * sc, sy = cy[k] * sc;
    const float fxs = floorf(sx), fys = floorf(sy);
    const float dx = sx - fxs, dy = sy - fys;                          // corrSample_kernel.cu:52-53
    // wave-uniform: kept in SGPRs from here to phase B (held in VGPRs these eight values per job would spill)
    lfx[k][l - 2] = __builtin_amdgcn_readfirstlane((int)fxs);
    lfy[k][l - 2] = __builtin_amdgcn_readfirstlane((int)fys);
    lw[k][l - 2][0] = uniform((1.0f - dy) * (1.0f - dx));              // same products as bilerp()
    lw[k][l - 2][1] = uniform((1.0f - dy) * dx);
    lw[k][l - 2][2] = uniform(dy * (1.0f - dx));
    lw[k][l - 2][3] = uniform(dy * dx);
    const int X = lfx[k][l - 2] + lxm, Y = lfy[k][l - 2] + lym;
    const float* sl = vb[l] + (size_t)(vrow_pix + (k ? pxb : pxa)) * (unsigned)g.ssz[l];
    // every lane loads from a clamped (legal) position; out-of-range lattice points become the reference's 0 padding
    // when the value is used (phase B), so no lane is branched around a load
    latin[k][l - 2] = (unsigned)X < (unsigned)g.W2[l] && (unsigned)Y < (unsigned)g.H2[l];
    const unsigned Xc = clamp0(X, g.W2[l] - 1), Yc = clamp0(Y, g.H2[l] - 1);
    latv[k][l - 2] = ldg<float>(sl, (pos_y<TILED>(Yc, g.W2[l], g.tpr[l]) + pos_x<TILED>(Xc)) * 4u);
  };

  // Issue order: the lattice levels need only the coords, so they go out while the offsets are still in flight; then
  // one offset job at a time.  The scheduling barriers keep the compiler from hoisting every job's address arithmetic
  // above the first load (which costs more registers than the 64 that eight waves per SIMD allow).
#pragma unroll
  for (int k = 0; k < GP; k++) {
    issue_lattice_level(k, 2);
    issue_lattice_level(k, 3);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int l = 0; l < (PROBE ? 1 : 2); l++)
#pragma unroll
    for (int k = 0; k < GP; k++) {
      issue_offset_level(k, l);
      __builtin_amdgcn_sched_barrier(0);
    }

  // ---- probe -> mask -> level-1 offsets (corr.py:94-99), then the level-1 gathers.  Both pixels' 3 x 3 probes are
  // blended and reduced at once, one per row of 16 lanes (the row sums below are row-relative). ----
  if constexpr (PROBE) {
    const int H2 = g.H2[1], W2 = g.W2[1];
    const unsigned l16 = lane & 15u;
    const int pi = (int)l16 / 3, pj = (int)l16 - pi * 3;
    const int src = (int)(lane & 16u) + ((pj * 4 + pi) & 15);
    const float q11 = __shfl(platv, src, kWave), q21 = __shfl(platv, src + 1, kWave);
    const float q12 = __shfl(platv, src + 4, kWave), q22 = __shfl(platv, src + 5, kWave);
    const float fxs = floorf(psx), fys = floorf(psy);
    const float dx = psx - fxs, dy = psy - fys;
    const int x1 = (int)fxs - 1 + pi, y1 = (int)fys - 1 + pj;
    float v = 0.0f;
    if (l16 < 9 && in_bounds(y1, x1, H2, W2)) v = bilerp(q11, q21, q12, q22, dx, dy);
    const float mean = row16_sum(v) / 9.0f;
    const float dd = l16 < 9 ? v - mean : 0.0f;
    const float var = row16_sum(dd * dd) / 8.0f;  // unbiased, torch.var default
    const float m = 1.0f / (1.0f + expf(-var));
    const float mk[GP] = {__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 0)),
                          __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 16))};
#pragma unroll
    for (int k = 0; k < GP; k++) {
      of[k][1].x *= mk[k];
      of[k][1].y *= mk[k];
      // persistent offset[1] *= mask (corr.py:99); the centre stays 0
      if (tap && !centre && pv[k])
        reinterpret_cast<f32x2a*>(off1 + (size_t)(row_pix + (k ? pxb : pxa)) * (NT * 2))[lane] = of[k][1];
      issue_offset_level(k, 1);
    }
  }

  // ---- phase B: blend in the reference's order (:83-86), park in the transpose tile ----
  const unsigned lrow = lane * (PITCH * 4u) + w * (GP * 4u);   // byte address of (channel = lane, pixel = w * GP)
  char* const lds = reinterpret_cast<char*>(outst);
  auto blend_offset_level = [&](int k, int l) __attribute__((always_inline)) {
    const bool sh = gshift[k][l];
    const float q11 = sh ? qa[k][l].y : qa[k][l].x;
    const float q12r = sh ? qb[k][l].y : qb[k][l].x;
    const float q21 = sh ? qc[k][l] : qa[k][l].y;        // out-of-range corners read as 0 (:76-80)
    const float q22r = sh ? qd[k][l] : qb[k][l].y;
    const float q12 = gyin[k][l] ? q12r : 0.0f;
    const float q22 = gyin[k][l] ? q22r : 0.0f;
    const float val = gvalid[k][l] ? bilerp(q11, q21, q12, q22, gdx[k][l], gdy[k][l]) : 0.0f;
    if (tap) *reinterpret_cast<float*>(lds + lrow + (l * NT * PITCH + k) * 4) = val;
  };
  auto blend_lattice_level = [&](int k, int l) __attribute__((always_inline)) {   // l = 2, 3
    const int x1 = lfx[k][l - 2] + tix, y1 = lfy[k][l - 2] + tjy;
    const int lv = __builtin_bit_cast(int, latin[k][l - 2] ? latv[k][l - 2] : 0.0f);
    const float q11 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)lsrc4, lv));
    const float q21 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)lsrc4 + 4, lv));
    const float q12 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)lsrc4 + 32, lv));
    const float q22 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)lsrc4 + 36, lv));
    // out-of-bounds corners arrive as 0 from the lattice; the whole-tap rule (:60) on top; bilerp()'s sum order
    const float* wq = lw[k][l - 2];
    const float val = ((unsigned)x1 < (unsigned)g.W2[l] && (unsigned)y1 < (unsigned)g.H2[l])
                          ? q11 * wq[0] + q21 * wq[1] + q12 * wq[2] + q22 * wq[3] : 0.0f;
    if (tap) *reinterpret_cast<float*>(lds + lrow + (l * NT * PITCH + k) * 4) = val;
  };
  if constexpr (PROBE) {
    // level 1 was requested last, behind the probe -> mask chain: everything else is blended while it travels
#pragma unroll
    for (int k = 0; k < GP; k++) blend_offset_level(k, 0);
#pragma unroll
    for (int k = 0; k < GP; k++) {
      blend_lattice_level(k, 2);
      blend_lattice_level(k, 3);
    }
#pragma unroll
    for (int k = 0; k < GP; k++) blend_offset_level(k, 1);
  } else {
#pragma unroll
    for (int k = 0; k < GP; k++) {
      blend_offset_level(k, 0);
      blend_offset_level(k, 1);
      blend_lattice_level(k, 2);
      blend_lattice_level(k, 3);
    }
  }
  __syncthreads();

  // ---- write-out: 64-byte row segments.  Thread t owns pixel t % 16 of channels t / 16 + 32 i ----
  const unsigned pc = threadIdx.x & (TPX - 1), c0 = threadIdx.x >> 4;
  const unsigned HW1 = (unsigned)H1 * W1;
  float* ob = out + ((size_t)e * CH * H1 + y) * W1 + xbase;      // channel 0, this row, first pixel of the tile (uniform)
  const unsigned voff = (c0 * HW1 + pc) * 4u;
  const char* lsrc = lds + (c0 * PITCH + pc) * 4u;
  if (xbase + pc < (unsigned)W1) {
#pragma unroll
    for (int i = 0; i < 6; i++)
      *reinterpret_cast<float*>(reinterpret_cast<char*>(ob + (size_t)i * 32 * HW1) + (size_t)voff) =
          *reinterpret_cast<const float*>(lsrc + i * 32 * PITCH * 4);
    if (c0 < CH - 192)
      *reinterpret_cast<float*>(reinterpret_cast<char*>(ob + (size_t)6 * 32 * HW1) + (size_t)voff) =
          *reinterpret_cast<const float*>(lsrc + 6 * 32 * PITCH * 4);
  }
}

static unsigned magic_u32(unsigned d) { return d == 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }

}  // namespace lean

// Launcher, called by pyramid_forward (defcorr.hip) when the launch is the production configuration.  Returns
// LGU_E_UNSUPPORTED for anything else: the caller then takes the general kernel.
int lean_pyramid_forward(const float* const* volumes, const float* coords, float* const* offsets, float* out, int E,
                         int H1, int W1, const int* H2, const int* W2, int flags, const int* edge_slot, hipStream_t st) {
  using namespace lean;
  const bool tiled = (flags & LGU_PYR_TILED) != 0, probe = (flags & LGU_PYR_PROBE) != 0;
  if (!offsets[0] || !offsets[1] || offsets[2] || offsets[3]) return LGU_E_UNSUPPORTED;
  if (flags & (LGU_PYR_OUT_NHWC | LGU_PYR_OUT_F16)) return LGU_E_UNSUPPORTED;
  Geo g;
  for (int l = 0; l < NL; l++) {
    g.H2[l] = H2[l]; g.W2[l] = W2[l];
    g.tpr[l] = (W2[l] + 7) >> 3;
    g.ssz[l] = tiled ? ((H2[l] + 3) >> 2) * g.tpr[l] * 32 : H2[l] * W2[l];
    if (!tiled && W2[l] % 4 != 0) return LGU_E_UNSUPPORTED;
  }
  // pairs: W2 % 8 == 0 on the offset levels (see the kernel); 32-bit pixel / slice arithmetic
  if (W2[0] % 8 != 0 || W2[1] % 8 != 0 || W2[0] < 8 || W2[1] < 8) return LGU_E_UNSUPPORTED;
  const unsigned tiles_per_row = (unsigned)(W1 + TPX - 1) / TPX;
  const unsigned long long rows = (unsigned long long)E * H1, grid = rows * tiles_per_row;
  if (grid >= (1ull << 31) / tiles_per_row || rows * (unsigned)H1 >= (1ull << 31) || rows * W1 >= (1ull << 31) / (NT * 2))
    return LGU_E_UNSUPPORTED;
  for (int l = 0; l < NL; l++)
    if ((unsigned long long)g.ssz[l] * 4 >= (1ull << 31)) return LGU_E_UNSUPPORTED;
  const size_t lds = sizeof(float) * CH * PITCH + lds_pad();  // pad: occupancy experiments only
  const int remap = 1;
  const unsigned mt = magic_u32(tiles_per_row), mh = magic_u32((unsigned)H1);
#define LGU_LEAN(PR, TL)                                                                                               \
  hipLaunchKernelGGL((defcorr_lean_kernel<PR, TL>), dim3((unsigned)grid), dim3(NWV* kWave), lds, st, volumes[0],        \
                     volumes[1], volumes[2], volumes[3], offsets[0], offsets[1], coords, out, edge_slot, g, H1, W1,    \
                     (int)tiles_per_row, mt, mh, (flags & LGU_PYR_COORDS_LAST) ? 1 : 0, remap)
  if (probe) { if (tiled) LGU_LEAN(true, true); else LGU_LEAN(true, false); }
  else { if (tiled) LGU_LEAN(false, true); else LGU_LEAN(false, false); }
#undef LGU_LEAN
  return launch_status();
}

}  // namespace lgu
