// defcorr.hip — deformable / plain bilinear sampling of the correlation-volume pyramid.
//
// Replaces (reference, relative to /root/reference):
//   offersample_LGS/defCorrSample_kernel.cu:25-91   defCorr_index_forward_kernel
//   offersample_LGS/corrSample_kernel.cu:24-82      corr_index_forward_kernel (LGU variant)
//   droid_slam/modules/corr.py:88-109               CorrBlock.__call__ (probe + 4 launches + cat)
//
// Design (gfx950, wave64) — see DESIGN.md §3:
//   Every (edge, pixel) owns a private slice of H2*W2 floats per level; adjacent pixels
//   share nothing, so the only reuse is among the rd*rd taps of one pixel.  The
//   reference maps one thread per pixel (lane stride = one slice = 12 KiB at level 0:
//   196 dependent 4-byte gathers per thread).  Here one WAVE serves a (pixel, level) job:
//     * levels with offsets: LANES = TAPS — lane t loads its own offset pair (the pixel's
//       rd*rd*2 floats are one coalesced read) and its four corners; the ~15-20 lines a
//       pixel touches are shared by the 49 lanes of each corner load;
//     * zero-offset levels: LANES = LATTICE POINTS — all taps share one fractional part and
//       sit on a (2r+2)^2 integer lattice = one wave; one 4-byte load per lane, taps pull
//       their corners with ds_bpermute;
//   a workgroup covers 16 x-adjacent pixels (tiles dealt to the XCDs in contiguous runs so that the
//   half-line writes of neighbours merge in one L2), issues every load of its jobs before
//   the first is consumed, blends in the reference's fp32 order, parks results in an LDS
//   transpose tile [channel][pixel] and writes 64-byte row segments straight into the
//   concatenated (E, L*rd*rd, H1, W1) tensor.  The uncertainty probe of corr.py:94-99 is
//   optionally fused.  Other output forms (OUTM): channel-last fp32 / half rows stored from the
//   registers, or the first corr_encoder layer applied on the matrix cores (droid_net.py:76-77).
//   Kernels in this file:
//     defcorr_gather_kernel   production (register gather)                  [variant 0/3/4/5]
//     defcorr_pyr_kernel      LDS-DMA staged variant: tap box by packed DPP reduction,
//                             global_load_lds_dwordx4 into an LDS pool — same HBM bytes,
//                             1.8x the instructions, kept for A/B              [variant 1]
//     defcorr_generic_kernel  one thread per output: any radius / shape         [variant 2]
#include "lgu_common.hpp"

namespace lgu {

// csrc/defcorr_lean.hip: the production configuration (radius 3, 4 levels, offsets on levels 0-1, planar output)
int lean_pyramid_forward(const float* const* volumes, const float* coords, float* const* offsets, float* out, int E,
                         int H1, int W1, const int* H2, const int* W2, int flags, const int* edge_slot, hipStream_t st);

constexpr int TP = 16;               // pixels (along x) per workgroup of the staged kernel / narrow gather tiles
constexpr int NWAVE = 4;             // waves per workgroup
constexpr int PPW = TP / NWAVE;      // pixels per wave
constexpr int FASTL = 4;             // levels served by one launch of the fast kernel
constexpr int POOL_FLOATS = 2560;    // 10 KiB of staging pool per wave: 8 worst-case boxes of 16 rows x 5 granules
constexpr int OUT_PITCH = TP + 1;    // transpose tile pitch (conflict-free column writes)
constexpr int MAX_GRAN = 2 * kWave;  // granules per staged job (two DMA instructions per lane)

struct PyrParams {
  const float* vol[FASTL];
  float* off[FASTL];
  int H2[FASTL];
  int W2[FASTL];
  const float* coords;  // (E,2,H1,W1), level-0 units
  float* out;           // (E, Ctot, H1, W1)
  int L, E, H1, W1;
  int tiles_per_row;
  int Ctot;   // channels of `out` per edge
  int cbase;  // first channel this launch writes
  int lbase;  // pyramid level of vol[0] (coords are divided by 2^(lbase+l))
  int flags;
  // slice geometry: floats per (edge,pixel) slice and, for the tiled layout, 4x8 tiles per tile row
  int ssz[FASTL];
  int tpr[FASTL];
  // optional storage indirection: edge e's volume slices live at slot edge_slot[e] of the level buffers (the state
  // container appends / drops edges without moving the pyramid); null = slot e.  Offsets, coords, out stay by e.
  const int* edge_slot;
  // fused consumer (OUTM 3): the first layer of corr_encoder, relu(W1 x + b1), applied to every pixel's Ctot samples
  const void* enc_w;  // half (ENC_N, enc_kp): W1 rows zero-padded from Ctot to enc_kp = ceil(Ctot / 32) * 32 entries
  const void* enc_b;  // half (ENC_N)
  int enc_kp;
};

constexpr int PYR_INT_XCD_REMAP = 1 << 16;  // internal flag, set by the launcher

constexpr int ENC_N = 128;       // output channels of the fused 1x1 convolution (UpdateModule.corr_encoder[0])
constexpr int ENC_XPITCH = 200;  // halves per pixel row of the LDS operand tile: 100 dwords -> the 16 rows of a fragment
                                 // read start 36 banks apart, conflict-free
constexpr int ENC_MAXKS = 7;     // k-steps of 32: up to 224 input channels (4 levels x 49 taps = 196)
constexpr int ENC_XBYTES = (32 * ENC_XPITCH + 64) * 2;             // operand tile (+ the over-read of the last row)
constexpr int ENC_LDS_BYTES = ENC_XBYTES + 8 * ENC_MAXKS * 1024;   // + the weight fragments: 70 272 bytes, 2 workgroups per CU
typedef _Float16 enc_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 enc_half4 __attribute__((ext_vector_type(4)));
typedef float enc_f32x4 __attribute__((ext_vector_type(4)));

// Position of target element (y, x) inside a slice.  Reference layout: row-major H2 x W2.
// Tiled layout (LGU_PYR_TILED): 4 x 8 element tiles = one 128-byte line each, tiles row-major over the slice
// padded to multiples of (4, 8).  HBM is fetched in whole 128-byte lines and a pixel's tap footprint is a
// ~13 x 13 blob: in 4 x 8 tiles it touches 8.3 lines at level 0 instead of 14.9 as row segments.
// Both forms are separable, addr = fy(y) + fx(x), so a neighbour is reached by adding a per-axis step.
template <bool TILED>
__device__ __forceinline__ int slice_pos(int y, int x, int W2, int tpr) {
  if (!TILED) return y * W2 + x;
  return (((y >> 2) * tpr + (x >> 3)) << 5) + ((y & 3) << 3) + (x & 7);
}
template <bool TILED>
__device__ __forceinline__ int step_x(int x) { return TILED ? ((x & 7) == 7 ? 25 : 1) : 1; }
template <bool TILED>
__device__ __forceinline__ int step_y(int y, int W2, int tpr) { return TILED ? ((y & 3) == 3 ? tpr * 32 - 24 : 8) : W2; }

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// One (pixel, level) job served with direct global gathers: the rare case of a tap box
// that does not fit the wave's LDS pool (offsets far outside the +-4 the network emits).
// Kept out of line so that the unrolled fast-path bodies stay small.  (ox, oy) is this
// lane's offset pair, centre already 0 and probe mask already applied.
template <int R>
__device__ __noinline__ void direct_job(const float* __restrict__ slice, float ox, float oy, float cx, float cy,
                                         int H2, int W2, float* outcol, int lane) {
  constexpr int RD = 2 * R + 1, NT = RD * RD;
  if (lane >= NT) return;
  const int ti = lane / RD, tj = lane - ti * RD;
  const float ofsX = ox + cx, ofsY = oy + cy;
  const int fx = (int)floorf(ofsX), fy = (int)floorf(ofsY);
  const float dx = ofsX - (float)fx, dy = ofsY - (float)fy;
  const int x1 = fx - R + ti, y1 = fy - R + tj;
  float val = 0.0f;
  if (in_bounds(y1, x1, H2, W2)) {
    const float* s = slice + (size_t)y1 * W2 + x1;
    const bool xin = x1 + 1 < W2, yin = y1 + 1 < H2;
    const float q11 = s[0];
    const float q21 = xin ? s[1] : 0.0f;
    const float q12 = yin ? s[W2] : 0.0f;
    const float q22 = (xin && yin) ? s[W2 + 1] : 0.0f;
    val = bilerp(q11, q21, q12, q22, dx, dy);
  }
  outcol[lane * OUT_PITCH] = val;
}

// The fused sampler.  Per (pixel, level) job one of two modes:
//  * STAGED  (level has an offset tensor): lanes = taps; tap box by a packed DPP reduction;
//    the box's 16-byte granules go to LDS by LDS-DMA; taps blend from LDS.
//  * LATTICE (offsets structurally zero: all taps share one fractional part and sit on a
//    (2R+2)^2 integer lattice): lanes = lattice points, ONE 4-byte load per lane, taps
//    gather their four corners from the lattice with ds_bpermute.  No LDS, no reduction.
//    For R = 1 the 4x4 lattices of the wave's four pixels share one wave instruction.
// PROBE additionally evaluates the 3x3 plain sample of level 1 (lattice mode, lanes 0..15),
// its unbiased variance, mask = sigmoid(var), scales this pixel's level-1 offsets by it
// (written back: the reference's persistent offset[1] *= mask, corr.py:94-99) before level
// 1 is sampled.  The level-1 box is made conservative (covers mask = 0..1) so staging never
// waits for the probe.
// ZMASK: bit l set = level l of this launch has structurally zero offsets (LATTICE mode),
// fixed at compile time so that each job carries the code of one mode only.
template <int R, bool PROBE, int ZMASK>
__global__ __launch_bounds__(NWAVE * kWave) void defcorr_pyr_kernel(const PyrParams p) {
  constexpr int RD = 2 * R + 1, NT = RD * RD;
  constexpr int LAT = 2 * R + 2;                 // lattice side
  constexpr int LATP = LAT <= 4 ? 4 : 8;         // lattice pitch in lanes
  constexpr int PIXOP = kWave / (LATP * LATP);   // pixels per lattice instruction: 4 (R=1) or 1
  static_assert(NT <= kWave && LAT <= LATP, "lanes = taps / lattice points must fit one wave");
  extern __shared__ float4 lds4[];
  float* const lds = reinterpret_cast<float*>(lds4);
  // layout: [NWAVE][POOL_FLOATS] staging pools, then the [L*NT][OUT_PITCH] transpose tile
  float* const outst = lds + NWAVE * POOL_FLOATS;

  const int lane = threadIdx.x & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* const pool = lds + w * POOL_FLOATS;

  int bid = blockIdx.x;
  const int tile = bid % p.tiles_per_row;
  bid /= p.tiles_per_row;
  const int y = bid % p.H1;
  const int e = bid / p.H1;
  const int xbase = tile * TP;

  // staged-mode lane roles: lane t = tap (i = t / RD moves in x, j = t % RD in y)
  const bool tap = lane < NT;
  const int ti = lane / RD, tj = lane - ti * RD;
  const bool centre = (ti == R) && (tj == R);
  // lattice-mode lane roles
  const int lpix = PIXOP == 1 ? 0 : lane / (LATP * LATP);   // pixel slot of this lane (R = 1 only)
  const int lq = lane & (LATP * LATP - 1);
  const int ly = lq / LATP, lx = lq & (LATP - 1);
  const bool lat_on = ly < LAT && lx < LAT;
  const bool ltap = lq < NT;                                 // lattice-mode tap lanes (per pixel slot)
  const int lti = PIXOP == 1 ? ti : lq / RD, ltj = PIXOP == 1 ? tj : lq - (lq / RD) * RD;
  const int lsrc = (lane - lq) + ltj * LATP + lti;           // lattice lane holding this tap's top-left

  const size_t HW1 = (size_t)p.H1 * p.W1;
  const size_t row_pix = ((size_t)e * p.H1 + y) * p.W1;  // pixel index of (e, y, 0)

  // ---- phase 0: coords + every offset pair of this wave's jobs (all loads in flight) ----
  float x0[PPW], y0[PPW];
  float2 off[PPW][FASTL];
#pragma unroll
  for (int k = 0; k < PPW; k++) {
    const int px = xbase + w * PPW + k;
    const bool pv = px < p.W1;
    x0[k] = pv ? p.coords[((size_t)e * 2 + 0) * HW1 + (size_t)y * p.W1 + px] : 0.0f;
    y0[k] = pv ? p.coords[((size_t)e * 2 + 1) * HW1 + (size_t)y * p.W1 + px] : 0.0f;
#pragma unroll
    for (int l = 0; l < FASTL; l++) {
      off[k][l] = make_float2(0.0f, 0.0f);
      // the centre tap's offset is forced to 0 (defCorrSample_kernel.cu:51-52): never read it
      if (!((ZMASK >> l) & 1) && l < p.L && pv && tap && !centre)
        off[k][l] = reinterpret_cast<const float2*>(p.off[l] + (row_pix + px) * (NT * 2))[lane];
    }
  }
  // reference side effect: offset[e][y][x][r][r][:] = 0 in global memory (stores only, no
  // dependence on the loads above, so nothing waits here)
  if (centre) {
#pragma unroll
    for (int k = 0; k < PPW; k++) {
      const int px = xbase + w * PPW + k;
      if (px >= p.W1) continue;
#pragma unroll
      for (int l = 0; l < FASTL; l++)
        if (!((ZMASK >> l) & 1) && l < p.L)
          reinterpret_cast<float2*>(p.off[l] + (row_pix + px) * (NT * 2))[lane] = make_float2(0.0f, 0.0f);
    }
  }

  // ---- phase 0.5 (unconditional, top level): consume every phase-0 register once, outside
  // any branch, so the compiler retires those loads here instead of draining the LDS-DMA
  // queue in front of each job.  cs = coords / 2^l (corr.py:102; the scale is exact).
  float2 cs[PPW][FASTL], ofs[PPW][FASTL];
#pragma unroll
  for (int l = 0; l < FASTL; l++) {
    const float sc = __builtin_ldexpf(1.0f, -(p.lbase + l));
#pragma unroll
    for (int k = 0; k < PPW; k++) {
      cs[k][l] = make_float2(x0[k] * sc, y0[k] * sc);
      ofs[k][l] = make_float2(off[k][l].x + cs[k][l].x, off[k][l].y + cs[k][l].y);  // defCorrSample_kernel.cu:56-57
    }
  }

  // ---- phase A: issue every job's loads ----
  int jbase[PPW][FASTL], jorg[PPW][FASTL], jrowp[PPW][FASTL];   // staged-job records (wave-uniform)
  float latv[PPW][FASTL];                                       // lattice-job values (one per lane)
  float platv[PPW];                                             // probe lattice values
  unsigned staged_mask = 0, direct_mask = 0, lattice_mask = 0;
  int pool_used = 0;
#pragma unroll
  for (int k = 0; k < PPW; k++) {
    const int px = xbase + w * PPW + k;
    const bool pv = px < p.W1;
    platv[k] = 0.0f;
    if (PROBE && pv) {  // 4x4 lattice of level 1 around coords/2 on lanes 0..15
      const int H2 = p.H2[1], W2 = p.W2[1];
      const int X = (int)floorf(cs[k][1].x) - 1 + (lane & 3), Y = (int)floorf(cs[k][1].y) - 1 + ((lane >> 2) & 3);
      if (lane < 16 && in_bounds(Y, X, H2, W2))
        platv[k] = p.vol[1][(row_pix + px) * ((size_t)H2 * W2) + (size_t)Y * W2 + X];
    }
#pragma unroll
    for (int l = 0; l < FASTL; l++) {
      jbase[k][l] = 0; jorg[k][l] = 0; jrowp[k][l] = 4; latv[k][l] = 0.0f;
      if (l >= p.L) continue;
      const int H2 = p.H2[l], W2 = p.W2[l];
      if ((ZMASK >> l) & 1) {
        // ---- LATTICE job ----
        if (PIXOP == 1) {
          if (!pv) continue;
          lattice_mask |= 1u << (k * FASTL + l);
          const int X = (int)floorf(cs[k][l].x) - R + lx, Y = (int)floorf(cs[k][l].y) - R + ly;
          if (lat_on && in_bounds(Y, X, H2, W2))
            latv[k][l] = p.vol[l][(row_pix + px) * ((size_t)H2 * W2) + (size_t)Y * W2 + X];
        } else if (k == 0) {  // one instruction covers the wave's PIXOP pixels
          lattice_mask |= 1u << (k * FASTL + l);
          const int pxl = xbase + w * PPW + lpix;
          float cxl = cs[0][l].x, cyl = cs[0][l].y;
#pragma unroll
          for (int kk = 1; kk < PPW; kk++)
            if (lpix == kk) { cxl = cs[kk][l].x; cyl = cs[kk][l].y; }
          const int X = (int)floorf(cxl) - R + lx, Y = (int)floorf(cyl) - R + ly;
          if (pxl < p.W1 && lpix < PPW && lat_on && in_bounds(Y, X, H2, W2))
            latv[k][l] = p.vol[l][(row_pix + pxl) * ((size_t)H2 * W2) + (size_t)Y * W2 + X];
        }
        continue;
      }
      // ---- STAGED job ----
      if (!pv) continue;
      int ylo, yhi, ga, gb;
      if (H2 * W2 <= 4 * kWave) {
        // small slice (<= 1 KiB): stage all of it with one wave instruction, no reduction
        ylo = 0; yhi = H2 - 1; ga = 0; gb = (W2 >> 2) - 1;
      } else {
        const int fx = (int)floorf(ofs[k][l].x), fy = (int)floorf(ofs[k][l].y);
        int xa = fx - R + ti, ya = fy - R + tj;   // top-left of this tap
        int xb = xa, yb = ya;
        bool part = tap && in_bounds(ya, xa, H2, W2);
        if (PROBE && l == 1) {
          // the mask in (0.5, 1] moves the tap between its zero-offset and full-offset
          // positions: cover both, clamped, unless the whole span is out of bounds
          const int xz = (int)floorf(cs[k][l].x) - R + ti, yz = (int)floorf(cs[k][l].y) - R + tj;
          const int xmn = xa < xz ? xa : xz, xmx = xa < xz ? xz : xa;
          const int ymn = ya < yz ? ya : yz, ymx = ya < yz ? yz : ya;
          part = tap && !(xmx < 0 || xmn >= W2 || ymx < 0 || ymn >= H2);
          xa = clampi(xmn, 0, W2 - 1); xb = clampi(xmx, 0, W2 - 1);
          ya = clampi(ymn, 0, H2 - 1); yb = clampi(ymx, 0, H2 - 1);
        }
        const int xh = xb + 1 < W2 ? xb + 1 : W2 - 1;
        const int yh = yb + 1 < H2 ? yb + 1 : H2 - 1;
        const int lo = wave_pk_reduce<true>(part ? pk16(xa, ya) : 0x7fff7fff);
        const int hi = wave_pk_reduce<false>(part ? pk16(xh, yh) : (int)0x80008000);
        ylo = pk_hi(lo); yhi = pk_hi(hi);
        ga = pk_lo(lo) >> 2; gb = pk_lo(hi) >> 2;
      }
      if (yhi < ylo) continue;  // no tap of this pixel touches the slice: outputs are all 0
      const int pitch = gb - ga + 1;
      const int n = (yhi - ylo + 1) * pitch;  // 16-byte granules in the box
      if (n > MAX_GRAN || pool_used + n * 4 > POOL_FLOATS) {
        direct_mask |= 1u << (k * FASTL + l);
        continue;
      }
      staged_mask |= 1u << (k * FASTL + l);
      jbase[k][l] = pool_used;
      jrowp[k][l] = pitch * 4;
      jorg[k][l] = ylo * pitch * 4 + ga * 4;
      const float* slice = p.vol[l] + (row_pix + px) * ((size_t)H2 * W2);  // wave-uniform
      const float rp = 1.0f / (float)pitch;
#pragma unroll
      for (int q = 0; q < MAX_GRAN / kWave; q++) {
        const int kk = q * kWave + lane;
        const int row = (int)(((float)kk + 0.5f) * rp);
        const int gq = kk - row * pitch;
        const unsigned voff = (unsigned)((ylo + row) * W2 + (ga + gq) * 4);
        // LDS destination = wave-uniform base + lane*16 (the DMA's own addressing)
        if (kk < n)
          __builtin_amdgcn_global_load_lds((glb_void*)(slice + voff), (lds_void*)(pool + pool_used + q * kWave * 4), 16,
                                           0, 0);
      }
      pool_used += n * 4;
    }
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();

  // ---- phase B: blend, park in the transpose tile ----
  float pmask[PPW];
#pragma unroll
  for (int k = 0; k < PPW; k++) {
    const int px = xbase + w * PPW + k;
    const bool pv = px < p.W1;
    pmask[k] = 1.0f;
    if (PROBE && pv) {
      // 3x3 plain sample of level 1 (corrSample_kernel.cu:52-77) on lanes 0..8 from the 4x4 lattice
      const int H2 = p.H2[1], W2 = p.W2[1];
      const int pi = lane / 3, pj = lane - pi * 3;
      const int src = (pj * 4 + pi) & 15;
      const float q11 = __shfl(platv[k], src, kWave), q21 = __shfl(platv[k], src + 1, kWave);
      const float q12 = __shfl(platv[k], src + 4, kWave), q22 = __shfl(platv[k], (src + 5) & 15, kWave);
      const float fxs = floorf(cs[k][1].x), fys = floorf(cs[k][1].y);
      const float dx = cs[k][1].x - fxs, dy = cs[k][1].y - fys;
      const int x1 = (int)fxs - 1 + pi, y1 = (int)fys - 1 + pj;
      float v = 0.0f;
      if (lane < 9 && in_bounds(y1, x1, H2, W2)) v = bilerp(q11, q21, q12, q22, dx, dy);
      // unbiased variance over the nine taps (torch.var, corr.py:96), sigmoid (corr.py:97)
      const float mean = row16_sum(lane < 9 ? v : 0.0f) / 9.0f;
      const float d = lane < 9 ? v - mean : 0.0f;
      const float var = row16_sum(d * d) / 8.0f;
      const float m = 1.0f / (1.0f + expf(-var));
      pmask[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
      // offset[1] *= mask, persistent (corr.py:99); the centre stays 0
      off[k][1].x *= pmask[k];
      off[k][1].y *= pmask[k];
      ofs[k][1] = make_float2(off[k][1].x + cs[k][1].x, off[k][1].y + cs[k][1].y);
      if (tap && !centre) reinterpret_cast<float2*>(p.off[1] + (row_pix + px) * (NT * 2))[lane] = off[k][1];
    }
#pragma unroll
    for (int l = 0; l < FASTL; l++) {
      if (l >= p.L) continue;
      const int H2 = p.H2[l], W2 = p.W2[l];
      if ((ZMASK >> l) & 1) {
        if (!(lattice_mask & (1u << (k * FASTL + l)))) continue;
        float cxl = cs[k][l].x, cyl = cs[k][l].y;
        int pxl = px;
        if (PIXOP > 1) {
          pxl = xbase + w * PPW + lpix;
#pragma unroll
          for (int kk = 1; kk < PPW; kk++)
            if (lpix == kk) { cxl = cs[kk][l].x; cyl = cs[kk][l].y; }
        }
        const float q11 = __shfl(latv[k][l], lsrc, kWave), q21 = __shfl(latv[k][l], lsrc + 1, kWave);
        const float q12 = __shfl(latv[k][l], lsrc + LATP, kWave), q22 = __shfl(latv[k][l], lsrc + LATP + 1, kWave);
        const float fxs = floorf(cxl), fys = floorf(cyl);
        const float dx = cxl - fxs, dy = cyl - fys;  // corrSample_kernel.cu:52-53
        const int x1 = (int)fxs - R + lti, y1 = (int)fys - R + ltj;
        float val = 0.0f;
        // out-of-bounds corners arrive as 0 from the lattice; the whole-tap rule (:60) on top
        if (in_bounds(y1, x1, H2, W2)) val = bilerp(q11, q21, q12, q22, dx, dy);
        if (ltap && pxl < p.W1 && (PIXOP == 1 || lpix < PPW))
          outst[(l * NT + (PIXOP == 1 ? lane : lq)) * OUT_PITCH + (w * PPW + (PIXOP == 1 ? k : lpix))] = val;
        continue;
      }
      if (!pv) continue;
      float val = 0.0f;  // masked taps stay 0 like torch::zeros in the reference (:181-183)
      if (staged_mask & (1u << (k * FASTL + l))) {
        const int fx = (int)floorf(ofs[k][l].x), fy = (int)floorf(ofs[k][l].y);
        const float dx = ofs[k][l].x - (float)fx, dy = ofs[k][l].y - (float)fy;  // :60-61
        const int x1 = fx - R + ti, y1 = fy - R + tj;                             // :63-66
        if (tap && in_bounds(y1, x1, H2, W2)) {                                   // :67 whole-tap rule
          const bool xin = x1 + 1 < W2, yin = y1 + 1 < H2;  // x2,y2 >= 0 follow from x1,y1 >= 0
          const int rowp = jrowp[k][l];
          const float* s = pool + jbase[k][l] + (y1 * rowp + x1 - jorg[k][l]);
          const float q11 = s[0];
          const float q21 = xin ? s[1] : 0.0f;
          const float q12 = yin ? s[rowp] : 0.0f;
          const float q22 = (xin && yin) ? s[rowp + 1] : 0.0f;
          val = bilerp(q11, q21, q12, q22, dx, dy);
        }
      }
      if (tap) outst[(l * NT + lane) * OUT_PITCH + (w * PPW + k)] = val;
    }
  }
  while (direct_mask) {  // wave-uniform, normally never entered
    const int j = __builtin_ctz(direct_mask);
    direct_mask &= direct_mask - 1;
    const int k = j / FASTL, l = j % FASTL;
    float2 o = make_float2(0.f, 0.f), c = make_float2(0.f, 0.f);
#pragma unroll
    for (int kk = 0; kk < PPW; kk++)
#pragma unroll
      for (int ll = 0; ll < FASTL; ll++)
        if (j == kk * FASTL + ll) { o = off[kk][ll]; c = cs[kk][ll]; }
    const int px = xbase + w * PPW + k;
    direct_job<R>(p.vol[l] + (row_pix + px) * ((size_t)p.H2[l] * p.W2[l]), o.x, o.y, c.x, c.y, p.H2[l], p.W2[l],
                  outst + (l * NT) * OUT_PITCH + (w * PPW + k), lane);
  }
  __syncthreads();

  // ---- coalesced write-out: 16 pixels (64 B) per channel row ----
  const int nout = p.L * NT * TP;
  float* const orow = p.out + (((size_t)e * p.Ctot + p.cbase) * p.H1 + y) * p.W1 + xbase;
  for (int idx = threadIdx.x; idx < nout; idx += NWAVE * kWave) {
    const int c = idx >> 4, pc = idx & (TP - 1);
    if (xbase + pc < p.W1) orow[(size_t)c * HW1 + pc] = outst[c * OUT_PITCH + pc];
  }
}

// ---------------------------------------------------------------------------------------
// Register-gather variant of the fused sampler: same decomposition (workgroup = 16 pixels of
// one row, wave = 4 pixels, lanes = taps / lattice points, LDS transpose tile for the
// write-out) but offset levels fetch their four corners with plain per-lane global loads
// instead of staging the tap box in LDS.  All taps of a pixel sit in one wave instruction,
// so the ~20 lines a pixel touches are shared by the 49 lanes of each load and stay in the
// CU's L1 for the other three corners.  Every load of the wave's 16 jobs is issued before
// the first result is used.  Far fewer instructions per pixel than the LDS-DMA kernel; the
// price is L1/TA work per corner instead of per line.
// GP = pixels per wave, TPX = pixels (along x) per workgroup: TPX / GP waves per workgroup
// OUTM = output form: 0 = planar fp32 (E, L*NT, H1, W1), the reference's tensor, written through the LDS transpose
// tile; 1 / 2 = channel-last (E, H1, W1, L*NT) in fp32 / half (LGU_PYR_OUT_NHWC [| LGU_PYR_OUT_F16]) — the form the
// consumer 1x1 convolution of the update operator prefers (droid_net.py:76-80 under autocast).  With lanes = taps a
// wave's store is already one contiguous run of the pixel's channels, so these forms need no LDS and no barrier.
// 3 = the consumer fused in (lgu_defcorr_pyramid_enc_fwd_f32): the samples of the workgroup's 32 pixels are parked in LDS
// as half rows (the cast autocast applies) and multiplied by the first corr_encoder layer on the matrix cores —
// v_mfma_f32_16x16x32_f16, A = 16 output channels of W1 (16 contiguous bytes per lane from the zero-padded weight
// rows), B = 16 pixels of the LDS tile, one 16 x 16 output tile per wave (8 channel tiles x 2 pixel tiles = the 16
// waves), fp32 accumulation, + bias, ReLU, half — and only the (E, H1, W1, 128) half result goes to HBM.
// PAIR: an offset level's corners are fetched as two 8-byte x-pairs (top row, bottom row) instead of four 4-byte
// gathers.  The vector L1 serves a gather quad by quad (4 lanes), one access per distinct line in the quad, and at
// ~39 accesses per corner instruction that access rate — not HBM — bounds the 4-byte form (profiles/r02_pmc_cold.txt:
// 771 L1 accesses per wave, 0.86 per CU cycle).  A pair never leaves its line except when the tap's x is the last
// column of a 4 x 8 tile (1 lane in 8): those lanes fetch the pair one element to the left and take the right
// neighbours with two extra instructions that only they execute.
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

template <int R, bool PROBE, int ZMASK, int GP, int TPX, bool TILED, int OUTM = 0, bool PAIR = false>
__global__ __launch_bounds__((TPX / GP) * kWave, GP == 4 ? (PROBE ? 4 : 5) : 8) void defcorr_gather_kernel(const PyrParams p) {
  constexpr int RD = 2 * R + 1, NT = RD * RD;
  constexpr int LAT = 2 * R + 2;
  constexpr int LATP = LAT <= 4 ? 4 : 8;
  constexpr int PIXOP = kWave / (LATP * LATP);
  extern __shared__ float4 lds4[];
  float* const outst = reinterpret_cast<float*>(lds4);  // [L*NT][(TPX + 1)]

  const int lane = threadIdx.x & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bid = blockIdx.x;
  if (p.flags & PYR_INT_XCD_REMAP) {
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Consecutive tiles share output lines
    // (a 16-pixel tile writes half of each 128-byte row segment): give every XCD a contiguous run of tiles so that
    // the two halves of a line meet in ONE L2 and leave it as a full-line write.
    const int n8 = (int)(gridDim.x & ~7u);
    if (bid < n8) bid = (bid & 7) * (n8 >> 3) + (bid >> 3);
  }
  const int tile = bid % p.tiles_per_row;
  bid /= p.tiles_per_row;
  const int y = bid % p.H1;
  const int e = bid / p.H1;
  const int xbase = tile * TPX;

  const bool tap = lane < NT;
  const int ti = lane / RD, tj = lane - ti * RD;
  const bool centre = (ti == R) && (tj == R);
  const int lpix = PIXOP == 1 ? 0 : lane / (LATP * LATP);
  const int lq = lane & (LATP * LATP - 1);
  const int ly = lq / LATP, lx = lq & (LATP - 1);
  const bool lat_on = ly < LAT && lx < LAT;
  const bool ltap = lq < NT;
  const int lti = PIXOP == 1 ? ti : lq / RD, ltj = PIXOP == 1 ? tj : lq - (lq / RD) * RD;
  const int lsrc = (lane - lq) + ltj * LATP + lti;

  const size_t HW1 = (size_t)p.H1 * p.W1;
  const size_t row_pix = ((size_t)e * p.H1 + y) * p.W1;
  const size_t vrow_pix = p.edge_slot ? ((size_t)p.edge_slot[e] * p.H1 + y) * p.W1 : row_pix;  // where the volume slices live

  if constexpr (OUTM == 3) {
    // The weight matrix goes to LDS by LDS-DMA right away: no registers, and its latency is over long before the
    // gathers are.  Fragment order: chunk (channel tile mt, k-step ks) = 1 KiB, lane i of the DMA instruction supplies
    // the 16 bytes lane i of the MFMA will want (row mt*16 + i%16, k = ks*32 + (i/16)*8), so the read-back is
    // lane-linear (conflict-free).  Channel tile mt is fetched by waves mt (k-steps 0-3) and mt+8 (4-6).
    const int mt = w & 7, nks = p.enc_kp >> 5;
    char* const wl = reinterpret_cast<char*>(lds4) + ENC_XBYTES + mt * (ENC_MAXKS * 1024);
    const char* const wg = reinterpret_cast<const char*>(p.enc_w) + ((size_t)(mt * 16 + (lane & 15)) * p.enc_kp + (lane >> 4) * 8) * 2;
#pragma unroll
    for (int ks = 0; ks < ENC_MAXKS; ks++)
      if (ks < nks && (ks < 4) == (w < 8))
        __builtin_amdgcn_global_load_lds((glb_void*)(wg + ks * 64), (lds_void*)(wl + ks * 1024), 16, 0, 0);
  }

  // ---- phase 0: coords + offsets ----
  float x0[GP], y0[GP];
  float2 off[GP][FASTL];
#pragma unroll
  for (int k = 0; k < GP; k++) {
    const int px = xbase + w * GP + k;
    const bool pv = px < p.W1;
    if (p.flags & LGU_PYR_COORDS_LAST) {  // (E,H1,W1,2): x, y interleaved, as the SLAM system holds them
      const float2 c = pv ? reinterpret_cast<const float2*>(p.coords)[(size_t)e * HW1 + (size_t)y * p.W1 + px] : make_float2(0.f, 0.f);
      x0[k] = c.x;
      y0[k] = c.y;
    } else {
      x0[k] = pv ? p.coords[((size_t)e * 2 + 0) * HW1 + (size_t)y * p.W1 + px] : 0.0f;
      y0[k] = pv ? p.coords[((size_t)e * 2 + 1) * HW1 + (size_t)y * p.W1 + px] : 0.0f;
    }
#pragma unroll
    for (int l = 0; l < FASTL; l++) {
      off[k][l] = make_float2(0.0f, 0.0f);
      if (!((ZMASK >> l) & 1) && l < p.L && pv && tap)
        off[k][l] = reinterpret_cast<const float2*>(p.off[l] + (row_pix + px) * (NT * 2))[lane];
    }
  }
  if (centre) {
    // reference side effect (defCorrSample_kernel.cu:51-52): offset[centre] = 0.  The tensor persists across the
    // 8-16 lookups of one volume, so after the first call the centre already holds +0: it is stored only when its
    // bits are not already that (same final tensor, no 32-byte partial write per pixel-level in the steady state)
#pragma unroll
    for (int k = 0; k < GP; k++) {
      const int px = xbase + w * GP + k;
#pragma unroll
      for (int l = 0; l < FASTL; l++) {
        if (((ZMASK >> l) & 1) || l >= p.L || px >= p.W1) continue;
        if ((__builtin_bit_cast(unsigned, off[k][l].x) | __builtin_bit_cast(unsigned, off[k][l].y)) != 0u)
          reinterpret_cast<float2*>(p.off[l] + (row_pix + px) * (NT * 2))[lane] = make_float2(0.0f, 0.0f);
        off[k][l] = make_float2(0.0f, 0.0f);
      }
    }
  }
  float2 cs[GP][FASTL];
#pragma unroll
  for (int l = 0; l < FASTL; l++) {
    const float sc = __builtin_ldexpf(1.0f, -(p.lbase + l));
#pragma unroll
    for (int k = 0; k < GP; k++) cs[k][l] = make_float2(x0[k] * sc, y0[k] * sc);
  }

  // ---- phase A: issue every load.  Order: probe lattices first (the level-1 gathers wait
  // for them), then everything that does not depend on the probe, then level 1.
  float platv[GP];
  float latv[GP][FASTL];
  float q[GP][FASTL][PAIR && TILED ? 6 : 4];
  float gdx[GP][FASTL], gdy[GP][FASTL];
  int gflag[GP][FASTL];  // bit0 tap valid, bit1 x2 in bounds, bit2 y2 in bounds, bit3 (PAIR) pair fetched one element to the left
#pragma unroll
  for (int k = 0; k < GP; k++) {
    platv[k] = 0.0f;
    const int px = xbase + w * GP + k;
    if (PROBE && px < p.W1) {
      const int H2 = p.H2[1], W2 = p.W2[1];
      const int X = (int)floorf(cs[k][1].x) - 1 + (lane & 3), Y = (int)floorf(cs[k][1].y) - 1 + ((lane >> 2) & 3);
      if (lane < 16 && in_bounds(Y, X, H2, W2))
        platv[k] = p.vol[1][(vrow_pix + px) * (size_t)p.ssz[1] + slice_pos<TILED>(Y, X, W2, p.tpr[1])];
    }
  }

  auto issue_level = [&](int k, int l) __attribute__((always_inline)) {
    const int px = xbase + w * GP + k;
    const bool pv = px < p.W1;
    const int H2 = p.H2[l], W2 = p.W2[l];
    const float ofsX = off[k][l].x + cs[k][l].x, ofsY = off[k][l].y + cs[k][l].y;  // :56-57
    const int fx = (int)floorf(ofsX), fy = (int)floorf(ofsY);
    gdx[k][l] = ofsX - (float)fx;
    gdy[k][l] = ofsY - (float)fy;
    const int x1 = fx - R + ti, y1 = fy - R + tj;
    const bool valid = tap && pv && in_bounds(y1, x1, H2, W2);  // :67
    // clamp so that every lane forms a legal address; results of clamped lanes are discarded
    // in phase B (the loaded registers are NOT touched here, so nothing waits in phase A)
    const int xc = clampi(x1, 0, W2 - 1), yc = clampi(y1, 0, H2 - 1);
    const bool xin = x1 + 1 < W2, yin = y1 + 1 < H2;
    gflag[k][l] = (valid ? 1 : 0) | (xin ? 2 : 0) | (yin ? 4 : 0);
    const int dxo = xin ? step_x<TILED>(xc) : 0, dyo = yin ? step_y<TILED>(yc, W2, p.tpr[l]) : 0;
    if constexpr (PAIR) {
      // last column of a tile (tiled) / of the slice (row-major): the pair starts one element to the left
      const bool shift = TILED ? ((xc & 7) == 7) : (xc == W2 - 1);
      const int xa = xc - (shift ? 1 : 0);
      const int yb = yc + 1 < H2 ? yc + 1 : yc;
      gflag[k][l] |= shift ? 8 : 0;
      const float* base = p.vol[l] + (vrow_pix + (pv ? px : 0)) * (size_t)p.ssz[l];
      const int pa = slice_pos<TILED>(yc, xa, W2, p.tpr[l]);
      const int pb = TILED ? pa + (yb != yc ? step_y<TILED>(yc, W2, p.tpr[l]) : 0) : pa + (yb != yc ? W2 : 0);
      const f32x2u a = *reinterpret_cast<const f32x2u*>(base + pa);
      const f32x2u b = *reinterpret_cast<const f32x2u*>(base + pb);
      q[k][l][0] = a.x; q[k][l][1] = a.y; q[k][l][2] = b.x; q[k][l][3] = b.y;
      if constexpr (TILED) {
        // right neighbours across the tile boundary: element (y, x + 1) = first column of the next tile, 26 floats on
        q[k][l][4] = q[k][l][5] = 0.0f;
        if (shift && xin) {
          q[k][l][4] = base[pa + 26];
          q[k][l][5] = base[pb + 26];
        }
      }
      return;
    }
    const float* s = p.vol[l] + (vrow_pix + (pv ? px : 0)) * (size_t)p.ssz[l] + slice_pos<TILED>(yc, xc, W2, p.tpr[l]);
    q[k][l][0] = s[0];
    q[k][l][1] = s[dxo];
    q[k][l][2] = s[dyo];
    q[k][l][3] = s[dyo + dxo];
  };

#pragma unroll
  for (int k = 0; k < GP; k++) {
    const int px = xbase + w * GP + k;
    const bool pv = px < p.W1;
#pragma unroll
    for (int l = 0; l < FASTL; l++) {
      latv[k][l] = 0.0f;
      gdx[k][l] = gdy[k][l] = 0.0f;
      gflag[k][l] = 0;
      q[k][l][0] = q[k][l][1] = q[k][l][2] = q[k][l][3] = 0.0f;
      if constexpr (PAIR && TILED) q[k][l][4] = q[k][l][5] = 0.0f;
      if (l >= p.L) continue;
      const int H2 = p.H2[l], W2 = p.W2[l];
      if ((ZMASK >> l) & 1) {
        if (PIXOP == 1) {
          const int X = (int)floorf(cs[k][l].x) - R + lx, Y = (int)floorf(cs[k][l].y) - R + ly;
          if (pv && lat_on && in_bounds(Y, X, H2, W2))
            latv[k][l] = p.vol[l][(vrow_pix + px) * (size_t)p.ssz[l] + slice_pos<TILED>(Y, X, W2, p.tpr[l])];
        } else if (k == 0) {
          const int pxl = xbase + w * GP + lpix;
          float cxl = cs[0][l].x, cyl = cs[0][l].y;
#pragma unroll
          for (int kk = 1; kk < GP; kk++)
            if (lpix == kk) { cxl = cs[kk][l].x; cyl = cs[kk][l].y; }
          const int X = (int)floorf(cxl) - R + lx, Y = (int)floorf(cyl) - R + ly;
          if (pxl < p.W1 && lpix < GP && lat_on && in_bounds(Y, X, H2, W2))
            latv[k][l] = p.vol[l][(vrow_pix + pxl) * (size_t)p.ssz[l] + slice_pos<TILED>(Y, X, W2, p.tpr[l])];
        }
      } else if (!(PROBE && l == 1)) {
        issue_level(k, l);
      }
    }
  }

  // ---- probe -> mask -> level-1 offsets (corr.py:94-99), then the level-1 gathers ----
  if (PROBE) {
#pragma unroll
    for (int k = 0; k < GP; k++) {
      const int px = xbase + w * GP + k;
      const bool pv = px < p.W1;
      const int H2 = p.H2[1], W2 = p.W2[1];
      const int pi = lane / 3, pj = lane - pi * 3;
      const int src = (pj * 4 + pi) & 15;
      const float q11 = __shfl(platv[k], src, kWave), q21 = __shfl(platv[k], src + 1, kWave);
      const float q12 = __shfl(platv[k], src + 4, kWave), q22 = __shfl(platv[k], (src + 5) & 15, kWave);
      const float fxs = floorf(cs[k][1].x), fys = floorf(cs[k][1].y);
      const float dx = cs[k][1].x - fxs, dy = cs[k][1].y - fys;
      const int x1 = (int)fxs - 1 + pi, y1 = (int)fys - 1 + pj;
      float v = 0.0f;
      if (lane < 9 && in_bounds(y1, x1, H2, W2)) v = bilerp(q11, q21, q12, q22, dx, dy);
      const float mean = row16_sum(lane < 9 ? v : 0.0f) / 9.0f;
      const float dd = lane < 9 ? v - mean : 0.0f;
      const float var = row16_sum(dd * dd) / 8.0f;  // unbiased, torch.var default
      const float m = 1.0f / (1.0f + expf(-var));
      const float mk = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
      off[k][1].x *= mk;
      off[k][1].y *= mk;
      if (pv && tap && !centre) reinterpret_cast<float2*>(p.off[1] + (row_pix + px) * (NT * 2))[lane] = off[k][1];
      issue_level(k, 1);
    }
  }

  // ---- phase B: blend, park in the transpose tile (or store the channel-last row directly) ----
  auto put = [&](int l, int ch, int pc, float val) __attribute__((always_inline)) {
    if constexpr (OUTM == 0) {
      outst[(l * NT + ch) * (TPX + 1) + pc] = val;
    } else if constexpr (OUTM == 3) {
      reinterpret_cast<_Float16*>(lds4)[pc * ENC_XPITCH + l * NT + ch] = (_Float16)val;
    } else {
      const size_t o = (row_pix + xbase + pc) * (size_t)p.Ctot + p.cbase + l * NT + ch;
      if constexpr (OUTM == 1) p.out[o] = val;
      else reinterpret_cast<_Float16*>(p.out)[o] = (_Float16)val;  // v_cvt_f16_f32, round-to-nearest-even = Tensor.half()
    }
  };
#pragma unroll
  for (int k = 0; k < GP; k++) {
    const int px = xbase + w * GP + k;
    const bool pv = px < p.W1;
#pragma unroll
    for (int l = 0; l < FASTL; l++) {
      if (l >= p.L) continue;
      const int H2 = p.H2[l], W2 = p.W2[l];
      if ((ZMASK >> l) & 1) {
        if (PIXOP > 1 && k > 0) continue;
        float cxl = cs[k][l].x, cyl = cs[k][l].y;
        int pxl = px;
        if (PIXOP > 1) {
          pxl = xbase + w * GP + lpix;
#pragma unroll
          for (int kk = 1; kk < GP; kk++)
            if (lpix == kk) { cxl = cs[kk][l].x; cyl = cs[kk][l].y; }
        }
        const float q11 = __shfl(latv[k][l], lsrc, kWave), q21 = __shfl(latv[k][l], lsrc + 1, kWave);
        const float q12 = __shfl(latv[k][l], lsrc + LATP, kWave), q22 = __shfl(latv[k][l], lsrc + LATP + 1, kWave);
        const float fxs = floorf(cxl), fys = floorf(cyl);
        const float dx = cxl - fxs, dy = cyl - fys;
        const int x1 = (int)fxs - R + lti, y1 = (int)fys - R + ltj;
        float val = 0.0f;
        if (in_bounds(y1, x1, H2, W2)) val = bilerp(q11, q21, q12, q22, dx, dy);
        if (ltap && pxl < p.W1 && (PIXOP == 1 || lpix < GP))
          put(l, PIXOP == 1 ? lane : lq, w * GP + (PIXOP == 1 ? k : lpix), val);
        continue;
      }
      if (!pv) continue;
      const int fl = gflag[k][l];
      float q11, q21, q12, q22;
      if constexpr (PAIR) {
        const bool sh = (fl & 8) != 0;
        q11 = sh ? q[k][l][1] : q[k][l][0];
        q12 = sh ? q[k][l][3] : q[k][l][2];
        if constexpr (TILED) {
          q21 = sh ? q[k][l][4] : q[k][l][1];
          q22 = sh ? q[k][l][5] : q[k][l][3];
        } else {  // row-major: a shifted pair means x + 1 is outside the slice, masked below
          q21 = q[k][l][1];
          q22 = q[k][l][3];
        }
      } else {
        q11 = q[k][l][0]; q21 = q[k][l][1]; q12 = q[k][l][2]; q22 = q[k][l][3];
      }
      q21 = (fl & 2) ? q21 : 0.0f;          // :76-80 out-of-range corners read as 0
      q12 = (fl & 4) ? q12 : 0.0f;
      q22 = ((fl & 6) == 6) ? q22 : 0.0f;
      const float val = (fl & 1) ? bilerp(q11, q21, q12, q22, gdx[k][l], gdy[k][l]) : 0.0f;
      if (tap) put(l, lane, w * GP + k, val);
    }
  }
  if constexpr (OUTM == 3) {
    static_assert(OUTM != 3 || (TPX == 32 && GP == 2), "fused encoder: 32-pixel tiles, 16 waves");
    const int mt = w & 7, ntile = w >> 3;  // channel tile, pixel tile of this wave
    const int lr = lane & 15, kg = lane >> 4;
    const int nks = p.enc_kp >> 5;
    const enc_half4 bias = *reinterpret_cast<const enc_half4*>(reinterpret_cast<const _Float16*>(p.enc_b) + mt * 16 + kg * 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's weight DMA (the compiler does not track it)
    __syncthreads();
    const char* const wl = reinterpret_cast<const char*>(lds4) + ENC_XBYTES + mt * (ENC_MAXKS * 1024) + lane * 16;
    const _Float16* xrow = reinterpret_cast<const _Float16*>(lds4) + (ntile * 16 + lr) * ENC_XPITCH + kg * 8;
    // Entries at k >= Ctot of the last k-step are not this pixel's samples (row padding / the next row).  W1 is zero
    // there, but 0 * NaN is NaN, so they are cleared on the B side as well: one 128-bit lane mask, built once.
    typedef unsigned enc_u32x4 __attribute__((ext_vector_type(4)));
    const int nvalid = p.Ctot - ((nks - 1) * 32 + kg * 8);  // valid entries of this lane's 8 in the last k-step
    enc_u32x4 lastmask;
#pragma unroll
    for (int j = 0; j < 4; j++) lastmask[j] = nvalid >= 2 * j + 2 ? 0xffffffffu : nvalid == 2 * j + 1 ? 0x0000ffffu : 0u;
    enc_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ks = 0; ks < ENC_MAXKS; ks++) {
      if (ks >= nks) break;
      const enc_half8 a = *reinterpret_cast<const enc_half8*>(wl + ks * 1024);
      enc_half8 b = *reinterpret_cast<const enc_half8*>(xrow + ks * 32);
      if (ks == nks - 1) b = __builtin_bit_cast(enc_half8, __builtin_bit_cast(enc_u32x4, b) & lastmask);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    }
    const int pxo = xbase + ntile * 16 + lr;
    if (pxo < p.W1) {
      enc_half4 o;
#pragma unroll
      for (int t = 0; t < 4; t++) o[t] = (_Float16)fmaxf(acc[t] + (float)bias[t], 0.0f);
      *reinterpret_cast<enc_half4*>(reinterpret_cast<_Float16*>(p.out) + (row_pix + pxo) * ENC_N + mt * 16 + kg * 4) = o;
    }
    return;
  }
  if constexpr (OUTM != 0) return;
  __syncthreads();

  const int nout = p.L * NT * TPX;
  float* const orow = p.out + (((size_t)e * p.Ctot + p.cbase) * p.H1 + y) * p.W1 + xbase;
  for (int idx = threadIdx.x; idx < nout; idx += (TPX / GP) * kWave) {
    const int c = idx / TPX, pc = idx & (TPX - 1);
    if (xbase + pc < p.W1) orow[(size_t)c * HW1 + pc] = outst[c * (TPX + 1) + pc];
  }
}

// Generic fallback: one thread per output element (x fastest -> coalesced stores).
// Serves any radius / any W2 / unaligned buffers; same arithmetic.
__global__ __launch_bounds__(256) void defcorr_generic_kernel(const float* __restrict__ vol,
                                                              const float* __restrict__ coords,
                                                              float* offs, float* __restrict__ out,
                                                              int E, int H1, int W1, int H2, int W2, int r,
                                                              float sc, int Ctot, int cbase) {
  const int rd = 2 * r + 1, nt = rd * rd;
  const size_t HW1 = (size_t)H1 * W1;
  const size_t total = (size_t)E * nt * HW1;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % W1);
    size_t rest = idx / W1;
    const int y = (int)(rest % H1);
    rest /= H1;
    const int t = (int)(rest % nt);
    const int e = (int)(rest / nt);
    const int i = t / rd, j = t - i * rd;
    const size_t pix = ((size_t)e * H1 + y) * W1 + x;
    const float x0 = coords[((size_t)e * 2 + 0) * HW1 + (size_t)y * W1 + x] * sc;
    const float y0 = coords[((size_t)e * 2 + 1) * HW1 + (size_t)y * W1 + x] * sc;
    float ox = 0.0f, oy = 0.0f;
    if (offs) {
      float* op = offs + (pix * nt + t) * 2;
      if (i == r && j == r) {
        op[0] = 0.0f;
        op[1] = 0.0f;
      } else {
        ox = op[0];
        oy = op[1];
      }
    }
    const float ofsX = ox + x0, ofsY = oy + y0;
    const int fx = (int)floorf(ofsX), fy = (int)floorf(ofsY);
    const float dx = ofsX - (float)fx, dy = ofsY - (float)fy;
    const int x1 = fx - r + i, y1 = fy - r + j;
    float val = 0.0f;
    if (in_bounds(y1, x1, H2, W2)) {
      const float* s = vol + pix * ((size_t)H2 * W2) + (size_t)y1 * W2 + x1;
      const bool xin = x1 + 1 < W2, yin = y1 + 1 < H2;
      const float q11 = s[0];
      const float q21 = xin ? s[1] : 0.0f;
      const float q12 = yin ? s[W2] : 0.0f;
      const float q22 = (xin && yin) ? s[W2 + 1] : 0.0f;
      val = bilerp(q11, q21, q12, q22, dx, dy);
    }
    out[(((size_t)e * Ctot + cbase + t) * H1 + y) * W1 + x] = val;
  }
}

static size_t pyr_lds_bytes(int L, int radius) {
  const int nt = (2 * radius + 1) * (2 * radius + 1);
  return sizeof(float) * ((size_t)NWAVE * POOL_FLOATS + (size_t)L * nt * OUT_PITCH);
}

// KIND 0: LDS-DMA staged kernel; 1-3: register-gather kernel (4 px/wave; 2 px/wave with 16- / 32-pixel tiles);
// 4 / 5: the 2 px/wave gather kernel over the TILED volume layout (16- / 32-pixel tiles);
// 8: KIND 5 with the fused corr_encoder layer; 9 / 10: the tiled gather kernel with channel-last half / fp32 output
// (no LDS, no barrier: 8-pixel tiles = 4-wave workgroups, which refill freed wave slots sooner than 16-wave ones);
// 11 / 12: KIND 4 / 2 fetching the corners as 8-byte x-pairs (production for the tiled / row-major planar output)
template <int R, bool PROBE, int ZMASK, int KIND>
static int launch_fast(const PyrParams& p, hipStream_t st) {
  const int nt_ = (2 * R + 1) * (2 * R + 1);
  constexpr int tpx = KIND >= 11 ? 16 : KIND >= 9 ? 8 : (KIND == 3 || KIND >= 5) ? 32 : TP;
  const size_t lds = KIND == 0 ? pyr_lds_bytes(p.L, R) : KIND == 8 ? (size_t)ENC_LDS_BYTES
                     : (KIND == 9 || KIND == 10) ? 0 : sizeof(float) * ((size_t)p.L * nt_ * (tpx + 1)) + lds_pad();  // pad: occupancy experiments (tools/ab_cold.py)
  // only the kernel of this KIND is instantiated
  constexpr auto kern = [] {
    if constexpr (KIND == 0) return &defcorr_pyr_kernel<R, PROBE, ZMASK>;
    else if constexpr (KIND == 1) return &defcorr_gather_kernel<R, PROBE, ZMASK, 4, 16, false>;
    else if constexpr (KIND == 2) return &defcorr_gather_kernel<R, PROBE, ZMASK, 2, 16, false>;
    else if constexpr (KIND == 3) return &defcorr_gather_kernel<R, PROBE, ZMASK, 2, 32, false>;
    else if constexpr (KIND == 4) return &defcorr_gather_kernel<R, PROBE, ZMASK, 2, 16, true>;
    else if constexpr (KIND == 5) return &defcorr_gather_kernel<R, PROBE, ZMASK, 2, 32, true>;
    else if constexpr (KIND == 8) return &defcorr_gather_kernel<R, PROBE, ZMASK, 2, 32, true, 3>;
    else if constexpr (KIND == 9) return &defcorr_gather_kernel<R, PROBE, ZMASK, 2, 8, true, 2>;
    else if constexpr (KIND == 10) return &defcorr_gather_kernel<R, PROBE, ZMASK, 2, 8, true, 1>;
    else if constexpr (KIND == 11) return &defcorr_gather_kernel<R, PROBE, ZMASK, 2, 16, true, 0, true>;
    else return &defcorr_gather_kernel<R, PROBE, ZMASK, 2, 16, false, 0, true>;
  }();
  const int nthreads = KIND >= 11 ? 8 * kWave : KIND >= 9 ? (tpx / 2) * kWave : (KIND == 3 || KIND >= 5) ? 16 * kWave : (KIND == 2 || KIND == 4) ? 8 * kWave : NWAVE * kWave;
  PyrParams q = p;
  q.tiles_per_row = (p.W1 + tpx - 1) / tpx;
  allow_max_dynamic_lds<kern>();
  const unsigned grid = (unsigned)((size_t)q.E * q.H1 * q.tiles_per_row);
  if (KIND == 2 || KIND == 4 || KIND >= 11) q.flags |= PYR_INT_XCD_REMAP;  // 16-pixel tiles with planar output
  hipLaunchKernelGGL(kern, dim3(grid), dim3(nthreads), lds, st, q);
  return launch_status();
}

static bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// Host dispatcher shared by the three forward entry points.
static int pyramid_forward(const float* const* volumes, const float* coords, float* const* offsets, float* out,
                           int L, int E, int H1, int W1, const int* H2, const int* W2, int radius, int flags,
                           void* stream, const int* edge_slot = nullptr, const void* enc_w = nullptr,
                           const void* enc_b = nullptr, int enc_n = 0) {
  if (!volumes || !coords || !offsets || !out || !H2 || !W2) return LGU_E_BADARG;
  const bool enc = enc_w != nullptr;
  if (enc && (!enc_b || (flags & (LGU_PYR_OUT_NHWC | LGU_PYR_OUT_F16)))) return LGU_E_BADARG;
  if (L < 1 || L > LGU_MAX_LEVELS || E < 0 || H1 < 1 || W1 < 1 || radius < 0 || radius > LGU_MAX_RADIUS)
    return LGU_E_BADARG;
  for (int l = 0; l < L; l++)
    if (!volumes[l] || H2[l] < 1 || W2[l] < 1) return LGU_E_BADARG;
  const bool probe = (flags & LGU_PYR_PROBE) != 0;
  const bool tiled = (flags & LGU_PYR_TILED) != 0;
  const bool coords_last = (flags & LGU_PYR_COORDS_LAST) != 0;
  const bool out_nhwc = (flags & LGU_PYR_OUT_NHWC) != 0, out_f16 = (flags & LGU_PYR_OUT_F16) != 0;
  if (out_f16 && !out_nhwc) return LGU_E_BADARG;
  if (probe && (L < 2 || offsets[1] == nullptr)) return LGU_E_BADARG;
  if (E == 0) return LGU_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nt = (2 * radius + 1) * (2 * radius + 1);
  const int Ctot = L * nt;
  // LGU_DEFCORR_VARIANT (debug/A-B only): 0 = production (csrc/defcorr_lean.hip for CorrBlock's configuration, else the
  // register-gather kernel with 8-byte pair fetches, 2 px/wave, 16-pixel tiles in XCD-aware order; 8-pixel tiles for the
  // channel-last output forms), 7 = the general pair-fetch kernel everywhere, 6 = 4-byte corner gathers (round 1's
  // production), 4 = 32-pixel tiles, 5 = 6,
  // 3 = 4 px/wave with 16-pixel tiles, 1 = LDS-DMA staged kernel, 2 = generic one-thread-per-output kernel
  // (independent cross-check)
  const int variant = env_int("LGU_DEFCORR_VARIANT", 0);

  int l0 = 0;
  while (l0 < L) {
    // group up to FASTL consecutive levels into one launch; the kernel is specialised on which
    // levels of the group have null (= zero) offsets: none, all, or "levels >= 2" (CorrBlock)
    int nl = (L - l0) < FASTL ? (L - l0) : FASTL;
    int zm = 0;
    for (int l = 0; l < nl; l++)
      if (offsets[l0 + l] == nullptr) zm |= 1 << l;
    const int full = (1 << nl) - 1;
    int tmpl;
    if (zm == 0) tmpl = 0x0;
    else if (zm == full) tmpl = 0xF;
    else if (nl >= 3 && zm == (full & 0xC)) tmpl = 0xC;
    else {  // mixed pattern: take the longest prefix of equal kind
      const bool first = offsets[l0] == nullptr;
      int m = 1;
      while (m < nl && (offsets[l0 + m] == nullptr) == first) m++;
      nl = m;
      tmpl = first ? 0xF : 0x0;
    }
    const bool pr = probe && l0 == 0;
    if (pr && (nl < 2 || tmpl == 0xF)) return LGU_E_UNSUPPORTED;  // host glue runs the probe as separate ops

    bool fast = (radius >= 1 && radius <= 3) && variant != 2 && aligned16(coords);
    for (int l = l0; l < l0 + nl; l++)
      fast = fast && (tiled || W2[l] % 4 == 0) && aligned16(volumes[l]) && (offsets[l] == nullptr || aligned16(offsets[l]));
    // the tiled layout is served by the production gather kernel at radius 3 only (what CorrBlock builds)
    if (tiled && !(fast && radius == 3 && (variant == 0 || (variant >= 4 && variant <= 7)))) return LGU_E_UNSUPPORTED;
    // interleaved coords and slot-indirected volumes are served by the register-gather kernels only
    if ((coords_last || edge_slot) && !(fast && variant != 1)) return LGU_E_UNSUPPORTED;
    // channel-last / half output: production tiled kernel only
    if (out_nhwc && !(tiled && variant == 0)) return LGU_E_UNSUPPORTED;
    // fused encoder layer: production tiled kernel, all levels in this one launch, 128 output channels
    if (enc && !(tiled && variant == 0 && nl == L && enc_n == ENC_N && aligned16(enc_w) && aligned16(out) &&
                 (reinterpret_cast<uintptr_t>(enc_b) & 7) == 0))
      return LGU_E_UNSUPPORTED;
    if (fast && variant == 0 && radius == 3 && L == FASTL && nl == FASTL && tmpl == 0xC && !enc && !out_nhwc) {
      const int rc = lean_pyramid_forward(volumes, coords, offsets, out, E, H1, W1, H2, W2, flags, edge_slot, st);
      if (rc == LGU_OK) { l0 += nl; continue; }
      if (rc != LGU_E_UNSUPPORTED) return rc;
    }
    if (fast) {
      PyrParams p;
      for (int l = 0; l < FASTL; l++) {
        const bool on = l < nl;
        p.vol[l] = on ? volumes[l0 + l] : nullptr;
        p.off[l] = on ? offsets[l0 + l] : nullptr;
        p.H2[l] = on ? H2[l0 + l] : 1;
        p.W2[l] = on ? W2[l0 + l] : 4;
        p.tpr[l] = (p.W2[l] + 7) >> 3;
        p.ssz[l] = tiled ? ((p.H2[l] + 3) >> 2) * p.tpr[l] * 32 : p.H2[l] * p.W2[l];
      }
      p.coords = coords; p.out = out; p.edge_slot = edge_slot;
      p.enc_w = enc_w; p.enc_b = enc_b; p.enc_kp = ((Ctot + 31) / 32) * 32;
      p.L = nl; p.E = E; p.H1 = H1; p.W1 = W1;
      p.tiles_per_row = (W1 + TP - 1) / TP;
      p.Ctot = Ctot; p.cbase = l0 * nt; p.lbase = l0; p.flags = flags;
      int rc;
#define LGU_LAUNCH_K(PR, ZM, KD)                                                                   \
  (radius == 3 ? launch_fast<3, PR, ZM, KD>(p, st)                                                 \
               : radius == 2 ? launch_fast<2, PR, ZM, KD>(p, st) : launch_fast<1, PR, ZM, KD>(p, st))
      // production choice of tile width: 16 pixels (8-wave workgroups) with the XCD-aware tile order that lets the
      // two half-line writes of neighbouring tiles merge in one L2; variant 4 = 32-pixel tiles (full-line writes
      // from one workgroup, 16-wave workgroups), 3.5 % slower
      const bool wide = variant == 4;
      const bool pair = variant == 0 || variant == 7;  // 8-byte x-pair fetches; variants 5 / 6 = the 4-byte corner gathers (A/B)
#define LGU_LAUNCH(PR, ZM)                                                                                     \
  (enc ? launch_fast<3, PR, ZM, 8>(p, st) :                                                                    \
   out_f16 ? launch_fast<3, PR, ZM, 9>(p, st) : out_nhwc ? launch_fast<3, PR, ZM, 10>(p, st) :                  \
   tiled ? (wide ? launch_fast<3, PR, ZM, 5>(p, st) : pair ? launch_fast<3, PR, ZM, 11>(p, st) : launch_fast<3, PR, ZM, 4>(p, st)) \
         : variant == 1 ? LGU_LAUNCH_K(PR, ZM, 0)                                                              \
                        : variant == 3 ? LGU_LAUNCH_K(PR, ZM, 1) : wide ? LGU_LAUNCH_K(PR, ZM, 3)              \
                        : pair ? LGU_LAUNCH_K(PR, ZM, 12) : LGU_LAUNCH_K(PR, ZM, 2))
      if (pr) rc = tmpl == 0xC ? LGU_LAUNCH(true, 0xC) : LGU_LAUNCH(true, 0x0);
      else rc = tmpl == 0xC ? LGU_LAUNCH(false, 0xC) : tmpl == 0xF ? LGU_LAUNCH(false, 0xF) : LGU_LAUNCH(false, 0x0);
#undef LGU_LAUNCH
#undef LGU_LAUNCH_K
      if (rc != LGU_OK) return rc;
    } else {
      // the generic kernel has no fused probe: the host glue then runs the probe as separate ops
      if (pr) return LGU_E_UNSUPPORTED;
      for (int l = l0; l < l0 + nl; l++) {
        const size_t total = (size_t)E * nt * H1 * W1;
        const unsigned grid = (unsigned)((total + 255) / 256 < 65535u * 16 ? (total + 255) / 256 : 65535u * 16);
        hipLaunchKernelGGL(defcorr_generic_kernel, dim3(grid), dim3(256), 0, st, volumes[l], coords, offsets[l], out,
                           E, H1, W1, H2[l], W2[l], radius, 1.0f / (float)(1 << l), Ctot, l * nt);
        const int rc = launch_status();
        if (rc != LGU_OK) return rc;
      }
    }
    l0 += nl;
  }
  return LGU_OK;
}

}  // namespace lgu

extern "C" {

int lgu_defcorr_pyramid_fwd_f32(const float* const* volumes, const float* coords, float* const* offsets, float* out,
                                int L, int E, int H1, int W1, const int* H2, const int* W2, int radius, int flags,
                                void* stream) {
  return lgu::pyramid_forward(volumes, coords, offsets, out, L, E, H1, W1, H2, W2, radius, flags, stream);
}

int lgu_defcorr_pyramid_slots_fwd_f32(const float* const* volumes, const int* edge_slot, const float* coords,
                                      float* const* offsets, float* out, int L, int E, int H1, int W1, const int* H2,
                                      const int* W2, int radius, int flags, void* stream) {
  return lgu::pyramid_forward(volumes, coords, offsets, out, L, E, H1, W1, H2, W2, radius, flags, stream, edge_slot);
}

int lgu_defcorr_pyramid_enc_fwd_f32(const float* const* volumes, const int* edge_slot, const float* coords,
                                    float* const* offsets, const void* enc_w, const void* enc_b, void* out, int L, int E,
                                    int H1, int W1, const int* H2, const int* W2, int radius, int enc_n, int flags,
                                    void* stream) {
  if (!enc_w || !enc_b) return LGU_E_BADARG;
  return lgu::pyramid_forward(volumes, coords, offsets, reinterpret_cast<float*>(out), L, E, H1, W1, H2, W2, radius, flags,
                              stream, edge_slot, enc_w, enc_b, enc_n);
}

int lgu_defcorr_fwd_f32(const float* volume, const float* coords, float* offset, float* corr, int E, int H1, int W1,
                        int H2, int W2, int radius, void* stream) {
  if (!offset) return LGU_E_BADARG;
  const float* vols[1] = {volume};
  float* offs[1] = {offset};
  return lgu::pyramid_forward(vols, coords, offs, corr, 1, E, H1, W1, &H2, &W2, radius, 0, stream);
}

int lgu_corridx_fwd_f32(const float* volume, const float* coords, float* corr, int E, int H1, int W1, int H2, int W2,
                        int radius, void* stream) {
  // defCorr with zero offsets is bit-identical to the LGU corr_index kernel
  // (corrSample_kernel.cu:52-60 vs defCorrSample_kernel.cu:56-67 with offset == 0).
  const float* vols[1] = {volume};
  float* offs[1] = {nullptr};
  return lgu::pyramid_forward(vols, coords, offs, corr, 1, E, H1, W1, &H2, &W2, radius, 0, stream);
}

}  // extern "C"
