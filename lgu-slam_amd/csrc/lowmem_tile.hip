// lowmem_tile.hip — tile-staged on-the-fly deformable correlation (lowMem_defSample).
//
// Replaces offersample_LGS/lowMem_defSample.cu:27-134 for radius <= 3 (the production case).
//
// dot(fmap1[p], bilerp(fmap2)(x,y)) == bilerp over the four corners of dot(fmap1[p], fmap2[corner]):
// a pixel's 49 taps only ever touch the integer positions inside its tap box (<= 16x16 for
// |offset| < 4, r = 3), so the kernel builds that LOCAL correlation patch
//     D_p[y2][x2] = sum_c fmap1[p][c] * fmap2[y2][x2][c]
// once per pixel and then samples it exactly like the volume path.  Neighbouring pixels'
// boxes overlap almost entirely, which the reference (one thread per pixel, 49*4*C global
// loads each) and a wave-per-pixel kernel cannot exploit.  Here a workgroup owns a 4x16
// pixel tile:
//   0. every wave finds the tap boxes of its 8 pixels (packed DPP reduction); their union
//      is the tile's staging window;
//   1. per 8-channel chunk the window is staged in LDS ONCE (16-byte coalesced loads, 48-byte
//      position pitch = conflict-free ds_read_b128 with lanes = positions) and every pixel
//      accumulates its <= 256 patch entries from it: lanes = patch positions, fmap1[p] chunk as
//      an LDS broadcast, 4 accumulators per lane per pixel;
//   2. the patch goes through a per-wave LDS scratch, the 49 taps blend their corners with
//      per-corner zero padding (lowMem_defSample.cu:102-117) and park the result in a
//      [tap][pixel] tile that is written out with 64-byte segments.
// fmap2 traffic per pixel drops from 49*4*512 B to (window/64 pixels)*512 B; LDS reads and
// fp32 FMAs (32.8 k per pixel-level at full boxes) become the bound.
// Tiles whose window exceeds the LDS budget, and pixels whose box exceeds 256 positions,
// take a per-tap fallback (lanes = taps, sequential channels) — correct for any offsets.
// Channel sums run sequentially with FMA per position; the reference sums 32-channel chunks
// of pre-blended values: equal to fp32 rounding (tests: 1e-5).
#include <type_traits>

#include "lgu_common.hpp"

namespace lgu {

constexpr int LT_W = 16, LT_H = 4, LT_PIX = LT_W * LT_H;
constexpr int LT_WAVES = 16, LT_PPW = LT_PIX / LT_WAVES;  // 1024 threads, 4 pixels per wave, one workgroup per CU
constexpr int LT_CH = 16;                                 // channels per staged chunk (8 for windows that only fit at the smaller pitch)
constexpr int LT_STAGE_FLOATS = 1536 * (LT_CH + 4);       // stage: 1536 positions at the 80-byte pitch = 2560 at the 48-byte pitch (122 880 B)
constexpr int LT_MAXPOS = LT_STAGE_FLOATS / (8 + 4);      // largest padded window served (2560 positions)
constexpr int LT_BOXW = 16;                               // patch row pitch: boxes up to 16 x 16 positions
constexpr int LT_MAXBOX = LT_BOXW * LT_BOXW;              // patch entries per pixel (4 per lane)
constexpr int LT_OUTP = LT_PIX + 1;
constexpr int LT_SCMAX = 4;                               // chunks staged per barrier pair when the window is small


// T = feature-map element type: float, or _Float16 (features kept in half precision as the
// SLAM system stores them; products and sums stay fp32, i.e. exactly what the reference call site
// `lowMem_defSample(fmap1.float(), fmap2.float(), ...)` computes, without the conversion passes and
// with half the LDS bytes per multiply-add).  One staged "piece" is 16 bytes = EPP elements.
template <typename T>
__device__ __forceinline__ float piece_dot(const float4& fa, const float4& va, float s);
template <>
__device__ __forceinline__ float piece_dot<float>(const float4& f, const float4& a, float s) {
  s = __builtin_fmaf(f.x, a.x, s); s = __builtin_fmaf(f.y, a.y, s);
  s = __builtin_fmaf(f.z, a.z, s); s = __builtin_fmaf(f.w, a.w, s);
  return s;
}
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
template <>
__device__ __forceinline__ float piece_dot<_Float16>(const float4& f, const float4& a, float s) {
  const half8_t fh = __builtin_bit_cast(half8_t, f), ah = __builtin_bit_cast(half8_t, a);
#pragma unroll
  for (int i = 0; i < 8; i++) s = __builtin_fmaf((float)fh[i], (float)ah[i], s);  // channel order, fp32 accumulate
  return s;
}

template <int R, typename T>
__global__ __launch_bounds__(LT_WAVES * kWave) void lowmem_tile_kernel(const T* __restrict__ fmap1,
                                                                       const T* __restrict__ fmap2,
                                                                       const float* __restrict__ coords, float* offset,
                                                                       float* __restrict__ corr, int B, int S, int H1,
                                                                       int W1, int H2, int W2, int C, int tiles_x,
                                                                       int tiles_y) {
  constexpr int RD = 2 * R + 1, NT = RD * RD;
  constexpr int EPP = 16 / (int)sizeof(T);  // elements per 16-byte piece
  extern __shared__ float4 smem4[];
  float* const stage = reinterpret_cast<float*>(smem4);       // [positions][pitch], LT_STAGE_FLOATS
  float* const dscr = stage + 4096;                           // [LT_WAVES][LT_MAXBOX] patch scratch, aliases the stage too
  int* const pbox = reinterpret_cast<int*>(stage + LT_STAGE_FLOATS);  // [LT_PIX][4] xlo,ylo,bw,bh
  int* const ubox = pbox + LT_PIX * 4;                        // xmin,ymin,xmax,ymax of the tile window
  float* const f1s = reinterpret_cast<float*>(ubox + 8);      // [LT_PIX][SC*LT_CH] fmap1 chunk(s) of the tile's pixels
  float* const outt = stage;                                  // [NT][LT_OUTP] (< 4096 floats), aliases the stage after the chunk loop

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int n = blockIdx.y;
  const size_t HW1 = (size_t)H1 * W1;
  const T* F2 = fmap2 + (size_t)b * H2 * W2 * C;
  const T* F1 = fmap1 + (size_t)b * H1 * W1 * C;  // per-edge bases keep the 32-bit offsets below small
  // reference indexing kept: offset[b*n] (lowMem_defSample.cu:80-83)
  // offset == nullptr: plain (zero-offset) sampling = altcorr_forward (src/altcorr_kernel.cu:27-149)
  float* const obase = offset ? offset + (size_t)(b * n) * HW1 * NT * 2 : nullptr;

  const bool tap = lane < NT;
  const int ix = lane / RD, iy = lane - ix * RD;  // offset / output index [ix][iy]
  const bool centre = (ix == R) && (iy == R);

  if (tid == 0) { ubox[0] = 0x7fffffff; ubox[1] = 0x7fffffff; ubox[2] = -1; ubox[3] = -1; }
  __syncthreads();

  // ---- phase 0: sample positions and tap boxes of this wave's pixels ----
  float2 o0[LT_PPW], c0v[LT_PPW];
#pragma unroll
  for (int k = 0; k < LT_PPW; k++) {  // all loads first, then the reductions
    const int pw = w * LT_PPW + k;
    const int h1 = ty * LT_H + (pw >> 4), w1 = tx * LT_W + (pw & 15);
    const bool pv = h1 < H1 && w1 < W1;
    const size_t pix = pv ? (size_t)h1 * W1 + w1 : 0;
    c0v[k] = reinterpret_cast<const float2*>(coords)[((size_t)b * S + n) * HW1 + pix];
    o0[k] = make_float2(0.f, 0.f);
    if (obase && pv && tap && !centre) o0[k] = reinterpret_cast<const float2*>(obase + pix * NT * 2)[lane];
  }
  if (obase && centre) {
#pragma unroll
    for (int k = 0; k < LT_PPW; k++) {
      const int pw = w * LT_PPW + k;
      const int h1 = ty * LT_H + (pw >> 4), w1 = tx * LT_W + (pw & 15);
      if (h1 < H1 && w1 < W1)
        reinterpret_cast<float2*>(obase + ((size_t)h1 * W1 + w1) * NT * 2)[lane] = make_float2(0.f, 0.f);  // :80-81
    }
  }
#pragma unroll
  for (int k = 0; k < LT_PPW; k++) {
    const int pw = w * LT_PPW + k;
    const int h1 = ty * LT_H + (pw >> 4), w1 = tx * LT_W + (pw & 15);
    const bool pv = h1 < H1 && w1 < W1;
    int lo = 0x7fff7fff, hi = (int)0x80008000;
    {
      const float xs = c0v[k].x + o0[k].x, ys = c0v[k].y + o0[k].y;  // :82-83
      const int w2 = (int)floorf(xs) - R + ix, h2 = (int)floorf(ys) - R + iy;
      const int xa = w2 > 0 ? w2 : 0, xb = w2 + 1 < W2 ? w2 + 1 : W2 - 1;
      const int ya = h2 > 0 ? h2 : 0, yb = h2 + 1 < H2 ? h2 + 1 : H2 - 1;
      const bool part = pv && tap && xa <= xb && ya <= yb;  // at least one corner in bounds
      lo = part ? pk16(xa, ya) : lo;
      hi = part ? pk16(xb, yb) : hi;
    }
    lo = wave_pk_reduce<true>(lo);
    hi = wave_pk_reduce<false>(hi);
    const int xlo = pk_lo(lo), ylo = pk_hi(lo), xhi = pk_lo(hi), yhi = pk_hi(hi);
    const bool any = pv && xhi >= xlo && yhi >= ylo;
    if (lane == 0) {
      pbox[pw * 4 + 0] = xlo; pbox[pw * 4 + 1] = ylo;
      pbox[pw * 4 + 2] = any ? xhi - xlo + 1 : 0;
      pbox[pw * 4 + 3] = any ? yhi - ylo + 1 : 0;
      if (any && xhi - xlo < LT_BOXW && yhi - ylo < LT_BOXW) {  // oversize boxes use the fallback, keep them out of the window
        atomicMin(&ubox[0], xlo); atomicMin(&ubox[1], ylo);
        atomicMax(&ubox[2], xhi); atomicMax(&ubox[3], yhi);
      }
    }
  }
  __syncthreads();
  const int UX0 = ubox[0], UY0 = ubox[1];
  const int UW = ubox[2] - UX0 + 1, UH = ubox[3] - UY0 + 1;
  const bool have_window = ubox[2] >= 0;
  // window rows are padded to a multiple of 16 positions: with the 80-byte position pitch and the
  // fixed 16-wide patch rows below, every ds_read_b128 lane group then covers all 64 banks once
  const int UWp = (UW + 15) & ~15;
  const int npos = have_window ? UWp * UH : 0;
  const bool tiled = have_window && npos <= LT_MAXPOS;  // workgroup-uniform

  float acc[LT_PPW][4];
  unsigned lpos[LT_PPW][2];  // stage position of this lane's patch entries, two 16-bit fields each, 0xffff = none
#pragma unroll
  for (int k = 0; k < LT_PPW; k++) {
    const int pw = w * LT_PPW + k;
    const int xlo = pbox[pw * 4 + 0], ylo = pbox[pw * 4 + 1], bw = pbox[pw * 4 + 2], bh = pbox[pw * 4 + 3];
    const bool boxed = tiled && bw > 0 && bw <= LT_BOXW && bh <= LT_BOXW;
    lpos[k][0] = lpos[k][1] = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      acc[k][j] = 0.0f;
      const int q = lane + j * kWave;            // patch entry (qy, qx) = (q / 16, q % 16)
      const int qy = q >> 4, qx = q & (LT_BOXW - 1);
      const unsigned ps = (boxed && qx < bw && qy < bh) ? (unsigned)((ylo + qy - UY0) * UWp + (xlo + qx - UX0)) : 0xffffu;
      lpos[k][j >> 1] |= ps << (16 * (j & 1));
    }
  }

  // ---- phase 1: stage the window chunk by chunk, accumulate every pixel's patch ----
  // A staged position holds 4 pieces (64 B of channels, 80-byte pitch) when the padded window fits
  // the stage at that pitch and 2 pieces (48-byte pitch, conflict-free by the same argument) for
  // larger windows: the stage holds LT_STAGE_FLOATS either way.
  auto chunk_loop = [&](auto pc_tag) __attribute__((always_inline)) {
    constexpr int Q4 = decltype(pc_tag)::value;  // 16-byte pieces per staged position: 4 (80-byte pitch) or 2 (48-byte pitch)
    constexpr int CH = Q4 * EPP;                  // channels per chunk
    constexpr int PITCH = Q4 * 4 + 4;             // floats per staged position
    constexpr int CAP = LT_STAGE_FLOATS / PITCH;  // positions the stage holds at this pitch
    const float rUW = 1.0f / (float)UWp;
    // small windows (coarse pyramid levels): stage SC chunks per barrier pair
    int SC = 1;
    if (npos * 4 <= CAP && C % (CH * 4) == 0 && Q4 * 4 <= 4 * LT_SCMAX) SC = 4;
    else if (npos * 2 <= CAP && C % (CH * 2) == 0) SC = 2;
    const int nvp = npos * SC;  // virtual positions: (sub-chunk, position)
    const float rnpos = 1.0f / (float)npos;
    // each thread's share of the window (chunk-invariant): source offsets relative to F2 + c0.
    // The NEXT chunk's global loads are issued before the current chunk's FMAs (register
    // prefetch), so only the LDS write sits between the two barriers.
    constexpr int NPRE = (CAP * Q4 + LT_WAVES * kWave - 1) / (LT_WAVES * kWave);
    int soff[NPRE];
    float4 pre[NPRE], pf1 = make_float4(0.f, 0.f, 0.f, 0.f);
    int f1off = -1;
    if (tid < LT_PIX * Q4 * SC) {  // fmap1 chunk(s) of the tile's 64 pixels (read back as LDS broadcasts)
      const int pw = tid / (Q4 * SC);
      const int h1 = ty * LT_H + (pw >> 4), w1 = tx * LT_W + (pw & 15);
      if (h1 < H1 && w1 < W1) f1off = (h1 * W1 + w1) * C + (tid - pw * Q4 * SC) * EPP;
    }
    if (f1off >= 0) pf1 = *reinterpret_cast<const float4*>(F1 + f1off);
#pragma unroll
    for (int i = 0; i < NPRE; i++) {
      const int idx = tid + i * LT_WAVES * kWave;
      const int vp = idx / Q4;
      const int sc = (int)(((float)vp + 0.5f) * rnpos);
      const int pos = vp - sc * npos;
      const int uy = (int)(((float)pos + 0.5f) * rUW);
      const int ux = pos - uy * UWp;
      soff[i] = (idx < nvp * Q4 && ux < UW) ? ((UY0 + uy) * W2 + (UX0 + ux)) * C + sc * CH + (idx % Q4) * EPP : -1;
      pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (soff[i] >= 0) pre[i] = *reinterpret_cast<const float4*>(F2 + soff[i]);  // chunk 0
    }
    const int cstep = CH * SC;
    for (int c0 = 0; c0 < C; c0 += cstep) {
      __syncthreads();  // previous chunk fully consumed
      if (tid < LT_PIX * Q4 * SC) *reinterpret_cast<float4*>(f1s + tid * 4) = pf1;
#pragma unroll
      for (int i = 0; i < NPRE; i++) {
        const int idx = tid + i * LT_WAVES * kWave;
        if (soff[i] >= 0) *reinterpret_cast<float4*>(stage + (idx / Q4) * PITCH + (idx % Q4) * 4) = pre[i];
      }
      __syncthreads();
      if (c0 + cstep < C) {
        if (f1off >= 0) pf1 = *reinterpret_cast<const float4*>(F1 + f1off + c0 + cstep);
#pragma unroll
        for (int i = 0; i < NPRE; i++)
          if (soff[i] >= 0) pre[i] = *reinterpret_cast<const float4*>(F2 + soff[i] + c0 + cstep);
      }
      for (int sc = 0; sc < SC; sc++) {
        const float* stg = stage + sc * npos * PITCH;
#pragma unroll
        for (int k = 0; k < LT_PPW; k++) {
          const int pw = w * LT_PPW + k;
          const int h1 = ty * LT_H + (pw >> 4), w1 = tx * LT_W + (pw & 15);
          if (h1 >= H1 || w1 >= W1) continue;  // wave-uniform
          float4 f1[Q4];  // same address in every lane: LDS broadcast
#pragma unroll
          for (int i = 0; i < Q4; i++) f1[i] = *reinterpret_cast<const float4*>(f1s + ((pw * SC + sc) * Q4 + i) * 4);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const unsigned ps = (lpos[k][j >> 1] >> (16 * (j & 1))) & 0xffffu;
            if (ps != 0xffffu) {
              float s = acc[k][j];
#pragma unroll
              for (int i = 0; i < Q4; i++)
                s = piece_dot<T>(f1[i], *reinterpret_cast<const float4*>(stg + ps * PITCH + i * 4), s);
              acc[k][j] = s;
            }
          }
        }
      }
    }
    __syncthreads();  // the stage is dead from here on: it becomes the output tile
  };
  if (tiled) {
    if (npos <= LT_STAGE_FLOATS / 20 && C % (4 * EPP) == 0) chunk_loop(std::integral_constant<int, 4>{});
    else chunk_loop(std::integral_constant<int, 2>{});
  }

  // ---- phase 2: sample every pixel's patch (or fall back to per-tap dots) ----
  float* const D = dscr + w * LT_MAXBOX;
#pragma unroll
  for (int k = 0; k < LT_PPW; k++) {  // sample positions again (L2 hits; the centre was zeroed in phase 0)
    const int pw = w * LT_PPW + k;
    const int h1 = ty * LT_H + (pw >> 4), w1 = tx * LT_W + (pw & 15);
    const bool pv = h1 < H1 && w1 < W1;
    const size_t pix = pv ? (size_t)h1 * W1 + w1 : 0;
    c0v[k] = reinterpret_cast<const float2*>(coords)[((size_t)b * S + n) * HW1 + pix];
    o0[k] = make_float2(0.f, 0.f);
    if (obase && pv && tap && !centre) o0[k] = reinterpret_cast<const float2*>(obase + pix * NT * 2)[lane];
  }
#pragma unroll
  for (int k = 0; k < LT_PPW; k++) {
    const int pw = w * LT_PPW + k;
    const int h1 = ty * LT_H + (pw >> 4), w1 = tx * LT_W + (pw & 15);
    if (h1 >= H1 || w1 >= W1) continue;
    const int xlo = pbox[pw * 4 + 0], ylo = pbox[pw * 4 + 1], bw = pbox[pw * 4 + 2], bh = pbox[pw * 4 + 3];
    const float xs = c0v[k].x + o0[k].x, ys = c0v[k].y + o0[k].y;
    const float fxs = floorf(xs), fys = floorf(ys);
    const float dx = xs - fxs, dy = ys - fys;  // :87-88
    const int w2 = (int)fxs - R + ix, h2 = (int)fys - R + iy;
    const bool b11 = in_bounds(h2, w2, H2, W2), b21 = in_bounds(h2, w2 + 1, H2, W2);
    const bool b12 = in_bounds(h2 + 1, w2, H2, W2), b22 = in_bounds(h2 + 1, w2 + 1, H2, W2);
    float q11 = 0.f, q21 = 0.f, q12 = 0.f, q22 = 0.f;
    if (tiled && bw <= LT_BOXW && bh <= LT_BOXW) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 4; j++) D[lane + j * kWave] = acc[k][j];  // [qy][16]
      __builtin_amdgcn_wave_barrier();
      if (tap) {
        const int o = (h2 - ylo) * LT_BOXW + (w2 - xlo);
        if (b11) q11 = D[o];
        if (b21) q21 = D[o + 1];
        if (b12) q12 = D[o + LT_BOXW];
        if (b22) q22 = D[o + LT_BOXW + 1];
      }
    } else if (tap) {
      // fallback: this lane's four corner dots, channels in order
      const T* f1p = F1 + ((size_t)h1 * W1 + w1) * C;
      const T* p11 = F2 + ((ptrdiff_t)h2 * W2 + w2) * C;
      for (int c = 0; c < C; c += EPP) {
        const float4 f = *reinterpret_cast<const float4*>(f1p + c);
        if (b11) q11 = piece_dot<T>(f, *reinterpret_cast<const float4*>(p11 + c), q11);
        if (b21) q21 = piece_dot<T>(f, *reinterpret_cast<const float4*>(p11 + C + c), q21);
        if (b12) q12 = piece_dot<T>(f, *reinterpret_cast<const float4*>(p11 + (size_t)W2 * C + c), q12);
        if (b22) q22 = piece_dot<T>(f, *reinterpret_cast<const float4*>(p11 + (size_t)W2 * C + C + c), q22);
      }
    }
    if (tap) outt[lane * LT_OUTP + pw] = bilerp(q11, q21, q12, q22, dx, dy);  // :114-117, per-corner zero padding
  }
  __syncthreads();

  // ---- write-out: corr[b][n][ix][iy][h1][w1], 16 pixels (64 B) per (tap, row) ----
  for (int idx = tid; idx < NT * LT_PIX; idx += LT_WAVES * kWave) {
    const int t = idx >> 6, pw = idx & (LT_PIX - 1);
    const int h1 = ty * LT_H + (pw >> 4), w1 = tx * LT_W + (pw & 15);
    if (h1 < H1 && w1 < W1) corr[((((size_t)b * S + n) * NT + t) * H1 + h1) * W1 + w1] = outt[t * LT_OUTP + pw];
  }
}

template <int R, typename T>
static int launch_tile(const T* fmap1, const T* fmap2, const float* coords, float* offset, float* corr, int B, int S,
                       int H1, int W1, int H2, int W2, int C, hipStream_t st) {
  const size_t lds = sizeof(float) * ((size_t)LT_STAGE_FLOATS + LT_PIX * 4 + 8 + LT_PIX * LT_CH * LT_SCMAX);
  auto kern = lowmem_tile_kernel<R, T>;
  allow_max_dynamic_lds<&lowmem_tile_kernel<R, T>>();
  const int tiles_x = (W1 + LT_W - 1) / LT_W, tiles_y = (H1 + LT_H - 1) / LT_H;
  hipLaunchKernelGGL(kern, dim3((unsigned)((size_t)B * tiles_x * tiles_y), (unsigned)S), dim3(LT_WAVES * kWave), lds, st,
                     fmap1, fmap2, coords, offset, corr, B, S, H1, W1, H2, W2, C, tiles_x, tiles_y);
  return launch_status();
}

template <typename T>
static int tile_dispatch(const T* fmap1, const T* fmap2, const float* coords, float* offset, float* corr, int B, int S,
                         int H1, int W1, int H2, int W2, int C, int radius, hipStream_t st) {
  const bool aligned = ((reinterpret_cast<uintptr_t>(fmap1) | reinterpret_cast<uintptr_t>(fmap2)) & 15) == 0;
  constexpr int epp = 16 / (int)sizeof(T);
  if (radius < 1 || radius > 3 || C % (2 * epp) != 0 || !aligned || S > 65535) return -1;
  if ((size_t)H2 * W2 * C >= (1u << 31) || (size_t)H1 * W1 * C >= (1u << 31)) return -1;  // 32-bit offsets inside one edge
  switch (radius) {
    case 1: return launch_tile<1, T>(fmap1, fmap2, coords, offset, corr, B, S, H1, W1, H2, W2, C, st);
    case 2: return launch_tile<2, T>(fmap1, fmap2, coords, offset, corr, B, S, H1, W1, H2, W2, C, st);
    default: return launch_tile<3, T>(fmap1, fmap2, coords, offset, corr, B, S, H1, W1, H2, W2, C, st);
  }
}

// Called from lgu_lowmem_defsample_fwd_f32 / lgu_altcorr_fwd_f32 (lowmem.hip).  Returns -1 when this
// kernel does not serve the arguments (the caller then uses the wave-per-pixel kernel).
int lowmem_tile_dispatch(const float* fmap1, const float* fmap2, const float* coords, float* offset, float* corr, int B,
                         int S, int H1, int W1, int H2, int W2, int C, int radius, hipStream_t st) {
  return tile_dispatch<float>(fmap1, fmap2, coords, offset, corr, B, S, H1, W1, H2, W2, C, radius, st);
}

// lowmem_mfma.hip
int lowmem_mfma_dispatch(const _Float16* fmap1, const _Float16* fmap2, const float* coords, float* offset, float* corr,
                         int B, int S, int H1, int W1, int H2, int W2, int C, int radius, hipStream_t st);

// Half feature maps: the matrix-core kernel when it serves the shape, else the VALU tile kernel above.
// LGU_LOWMEM_H16_VARIANT (debug/A-B only): 0 = matrix-core kernel, 1 = VALU tile kernel.
static int h16_dispatch(const _Float16* fmap1, const _Float16* fmap2, const float* coords, float* offset, float* corr,
                        int B, int S, int H1, int W1, int H2, int W2, int C, int radius, hipStream_t st) {
  if (env_int("LGU_LOWMEM_H16_VARIANT", 0) == 0) {
    const int rc = lowmem_mfma_dispatch(fmap1, fmap2, coords, offset, corr, B, S, H1, W1, H2, W2, C, radius, st);
    if (rc >= 0) return rc;
  }
  return tile_dispatch<_Float16>(fmap1, fmap2, coords, offset, corr, B, S, H1, W1, H2, W2, C, radius, st);
}

}  // namespace lgu

extern "C" {

// Mixed-precision entry points: fp16 feature maps, fp32 coords / offsets / accumulation / output.
// Numerically these ARE the reference call sites `lowMem_defSample(fmap1.float(), fmap2.float(), ...)`
// (droid_slam/modules/corr.py:209) and `altcorr_forward(fmap1.float(), fmap2.float(), ...)` (:202) for
// feature maps stored in half precision, as droid_slam/depth_video.py keeps them.
int lgu_lowmem_defsample_fwd_h16(const void* fmap1, const void* fmap2, const float* coords, float* offset, float* corr,
                                 int B, int S, int H1, int W1, int H2, int W2, int C, int NO, int radius, void* stream) {
  if (!fmap1 || !fmap2 || !coords || !offset || !corr) return LGU_E_BADARG;
  if (B < 0 || S < 1 || H1 < 1 || W1 < 1 || H2 < 1 || W2 < 1 || C < 1 || radius < 0) return LGU_E_BADARG;
  if ((long long)(B - 1) * (S - 1) >= (long long)NO) return LGU_E_BADARG;
  if (B == 0) return LGU_OK;
  const int rc = lgu::h16_dispatch(static_cast<const _Float16*>(fmap1), static_cast<const _Float16*>(fmap2), coords,
                                   offset, corr, B, S, H1, W1, H2, W2, C, radius, reinterpret_cast<hipStream_t>(stream));
  return rc < 0 ? LGU_E_UNSUPPORTED : rc;
}

int lgu_altcorr_fwd_h16(const void* fmap1, const void* fmap2, const float* coords, float* corr, int B, int S, int H1,
                        int W1, int H2, int W2, int C, int radius, void* stream) {
  if (!fmap1 || !fmap2 || !coords || !corr) return LGU_E_BADARG;
  if (B < 0 || S < 1 || H1 < 1 || W1 < 1 || H2 < 1 || W2 < 1 || C < 1 || radius < 0) return LGU_E_BADARG;
  if (B == 0) return LGU_OK;
  const int rc = lgu::h16_dispatch(static_cast<const _Float16*>(fmap1), static_cast<const _Float16*>(fmap2), coords,
                                   nullptr, corr, B, S, H1, W1, H2, W2, C, radius, reinterpret_cast<hipStream_t>(stream));
  return rc < 0 ? LGU_E_UNSUPPORTED : rc;
}

}  // extern "C"
