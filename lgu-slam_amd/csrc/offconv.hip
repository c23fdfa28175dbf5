// offconv.hip — the offset heads of AltCorrBlock on the matrix cores (gfx950).
//
// Reference (droid_slam/modules/corr.py:174-189, :217-220): every call of AltCorrBlock.corr_fn gathers the two frames of
// each edge, multiplies by 4, concatenates to (E, 256, H, W), casts to fp32 and runs ofsMap = Conv2d(256, 98, 3,
// padding=1) in fp32 (update_lowmem disables autocast) — 35 GFLOP per 16 edges at 60x80, 0.48-0.89 ms in the library's
// fp32 kernels, the largest single piece of the low-memory lookup.
//
// The inputs of that convolution are exactly representable in half: they are the stored half feature maps (times 4,
// which is folded into the weights here, exactly).  So the convolution is evaluated as an implicit GEMM on the half
// matrix cores with fp32-accurate weights:  W' = 4 W = hi + lo, two half parts (22 significant bits);  x . W' =
// x . hi + x . lo with exact half x half products and fp32 accumulation (v_mfma_f32_16x16x32_f16).  The result differs
// from an fp32 convolution by the weights' 2^-22 truncation — below the fp32 convolution's own summation noise (measured
// against an fp64 evaluation: 2.2e-6 here, 1.1e-6 for the library kernel, on outputs of magnitude 1.6).
//
//   M = pixels (flattened y*W + x, 16 per MFMA tile), N = output channels (98, padded to 112 = 7 tiles),
//   K = 9 taps x 256 channels = 72 steps of 32.
// Workgroup = 4 waves x 2 (or 1) pixel tiles = 128 (64) pixels.  A (pixels) comes straight from the channel-last frame buffers —
// 16 contiguous bytes per lane, frame ii[e] for channels 0-127 and jj[e] for 128-255, zero outside the image — no
// gather, no concatenation, no cast.  B (weights) is prepacked on the host in MFMA fragment order, 14 KiB per K step
// (2 parts x 7 tiles x 1 KiB), streamed into a three-slot LDS ring by LDS-DMA two steps ahead and read back
// lane-linearly.  Output (E, 98, H, W) fp32 with the bias added: a lane owns 4 consecutive pixels of one channel = one
// 16-byte store.  The residual head (corr.py:219-220: ofs_residual on the 2 x 2 average of that input) is the same
// kernel over frames pooled once per block, with the input in two half parts as well (template LO).
#include "lgu_common.hpp"

namespace lgu {

constexpr int OC_NT = 7;                      // output-channel tiles (<= 112 channels)
constexpr int OC_CHUNK = 2 * OC_NT * 1024;    // bytes of weight fragments per K step (hi + lo)
constexpr int OC_WAVES = 4;                   // waves per workgroup
constexpr int OC_RING = 3;                    // LDS ring of weight chunks: chunk s + 2 streams in while chunk s is multiplied

typedef _Float16 oc_half8 __attribute__((ext_vector_type(8)));
typedef float oc_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned oc_u32x4 __attribute__((ext_vector_type(4)));   // a fragment as the four dwords it is loaded as
typedef __attribute__((address_space(3))) void oc_lds_void;
typedef const __attribute__((address_space(1))) void oc_glb_void;

// 16 bytes of zeros in device memory: what a tap outside the image loads (the select is on the ADDRESS, so the loaded
// fragment is used as it arrives and no wait sits next to the load)
// (16 fragments: the small-grid kernel adds a K-step offset of up to 12 fragments to whichever pointer the select produced)
__device__ oc_u32x4 g_oc_zero[16] = {};  // not const: keeps it in the global address space (a constant-space pointer would turn the select into flat loads)

struct OffConvParams {
  const _Float16* frames;  // (NF, H, W, C) channel-last, C = 128
  const _Float16* frames_lo;  // LO kernels: second half part of the input (input = frames + frames_lo), same layout
  const long long* ii;     // (E) frame of channels 0..C-1
  const long long* jj;     // (E) frame of channels C..2C-1
  const _Float16* wpack;   // [9][KS][2][OC_NT][64][8] halves, KS = 2C / 32
  const float* bias;       // (Cout)
  float* out;              // (E, Cout, H, W)
  int E, H, W, C, Cout, KS;
  int ksh;                 // K steps per tap that read frame ii[e]; the rest read jj[e] (two-source form: KS / 2)
  // Worklist form (per-frame partial convolutions, lgu_offset_conv_worklist_h16): "edge" k is entry k of the worklist —
  // frame worklist[k] >> 1, input half worklist[k] & 1 — as long as k < *wcount; the single source is that frame, the
  // weights / bias / output buffer those of its half, and the output row is the FRAME.
  const int* worklist;
  const int* wcount;
  const _Float16* wpack_b;
  const float* bias_b;
  float* out_b;
};

// LO: the input has two half parts (e.g. 2 x 2 averages of half values, which need up to 24 bits): x = hi + lo and
// x . W' = hi . whi + hi . wlo + lo . whi + lo . wlo.
// MT: pixel tiles (of 16) per wave: a workgroup covers 64 MT pixels.  2 for full-resolution maps; 1 when that would
// leave CUs idle (the residual head runs at half resolution: 10 x 16 workgroups of 128 pixels on 256 CUs).
//
// K loop (9 taps x KS steps): the weight chunk of step s + 2 is in flight by LDS-DMA and the pixel fragments of step
// s + 2 in registers while step s is multiplied — one raw barrier per step, a COUNTED vmcnt in front of it (never 0:
// __syncthreads() would drain the DMA queue, cdna_hip_programming.md "Pipelining across barriers").  Round 1's form
// (two buffers, vmcnt(0) + __syncthreads() per step) ran at 28 % of the matrix peak.  Pixel loads are unconditional
// (zero padding by pointing the lane at a zero fragment), so every wave issues the same number of vector-memory
// operations per step and the count is exact.
template <bool LO, int MT>
__global__ __launch_bounds__(OC_WAVES* kWave) void offconv_frames_kernel(const OffConvParams p) {
  constexpr int PIX = OC_WAVES * MT * 16;
  extern __shared__ float4 oc_smem[];  // OC_RING x OC_CHUNK
  char* const wbuf = reinterpret_cast<char*>(oc_smem);
  const int lane = threadIdx.x & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int e = blockIdx.y;
  const int HW = p.H * p.W;
  const int lr = lane & 15, kg = lane >> 4;
  const size_t fstride = (size_t)HW * p.C;
  size_t o1, o2;
  const _Float16* wpack = p.wpack;
  const float* bias = p.bias;
  float* outp = p.out;
  if (p.worklist) {
    if ((int)blockIdx.y >= *p.wcount) return;  // workgroup-uniform
    const int entry = p.worklist[blockIdx.y];
    e = entry >> 1;
    o1 = o2 = (size_t)e * fstride;
    if (entry & 1) { wpack = p.wpack_b; bias = p.bias_b; outp = p.out_b; }
  } else {
    o1 = (size_t)p.ii[e] * fstride;
    o2 = (size_t)p.jj[e] * fstride;
  }
  const int ksh = p.ksh;  // K steps per tap of the first frame

  // this lane's A-row pixel in each of the wave's tiles
  int py[MT], px[MT];
  bool pv[MT];
#pragma unroll
  for (int t = 0; t < MT; t++) {
    const int pix = blockIdx.x * PIX + (w * MT + t) * 16 + lr;
    pv[t] = pix < HW;
    const int pc = pv[t] ? pix : HW - 1;
    py[t] = pc / p.W;
    px[t] = pc - py[t] * p.W;
  }

  oc_f32x4 acc[MT][OC_NT];
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int n = 0; n < OC_NT; n++) acc[t][n] = oc_f32x4{0.f, 0.f, 0.f, 0.f};

  const int nsteps = 9 * p.KS;
  // weight chunk `s` -> ring slot s % 3: 14 KiB = 3.5 x (256 threads x 16 bytes), contiguous in wpack
  // (slot = ring slot to fill, s = chunk to read: they differ only past the last step, where the pipeline keeps issuing
  // so that every step carries the same number of operations; those fills land in a slot nobody reads any more)
  auto stage = [&](int slot, int s) {
    const char* src = reinterpret_cast<const char*>(wpack) + (size_t)s * OC_CHUNK + (size_t)w * 1024 + lane * 16;
    char* dst = wbuf + slot * OC_CHUNK + w * 1024;  // wave-uniform; the DMA adds lane * 16
    // four pieces per wave, branch-free: waves 2, 3 have only three (14 KiB = 3.5 x 4 KiB) and repeat their last one —
    // the same bytes to the same place.  (A branch around a DMA makes the compiler's wait-count pass lose the counts and
    // drain the queue before later uses of prefetched registers.)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int kk = (k * 4096 + w * 1024 < OC_CHUNK) ? k : k - 1;
      __builtin_amdgcn_global_load_lds((oc_glb_void*)(src + kk * 4096), (oc_lds_void*)(dst + kk * 4096), 16, 0, 0);
    }
  };
  // pixel fragments of step s: always loaded — taps outside the image read the zero fragment
  auto load_a = [&](int s, oc_u32x4 (&a)[MT], oc_u32x4 (&al)[MT]) {
    const int tap = s / p.KS, ks = s - tap * p.KS;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    const size_t fo = (ks < ksh ? o1 : o2) + (ks < ksh ? ks : ks - ksh) * 32 + kg * 8;
#pragma unroll
    for (int t = 0; t < MT; t++) {
      const int yy = py[t] + dy, xx = px[t] + dx;
      const bool ok = pv[t] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      const size_t off = fo + ((size_t)(ok ? yy : 0) * p.W + (ok ? xx : 0)) * p.C;
      a[t] = *(ok ? reinterpret_cast<const oc_u32x4*>(p.frames + off) : g_oc_zero);
      if (LO) al[t] = *(ok ? reinterpret_cast<const oc_u32x4*>(p.frames_lo + off) : g_oc_zero);
    }
  };

  // three register sets in fixed roles (step s uses set s % 3): no copies between them — a copy would have to wait for
  // the youngest load, i.e. drain the DMA queue behind it
  oc_u32x4 ar[OC_RING][MT], lr_[OC_RING][MT];
  stage(0, 0);
  load_a(0, ar[0], lr_[0]);
  stage(1, 1);
  load_a(1, ar[1], lr_[1]);
  auto step = [&](int s, int slot, oc_u32x4 (&a)[MT], oc_u32x4 (&al)[MT], oc_u32x4 (&an)[MT], oc_u32x4 (&aln)[MT]) __attribute__((always_inline)) {
    // chunk s has landed (this wave's part): everything older than the DMA of chunk s + 1 and the pixel loads of
    // step s + 1 is complete when at most that many operations are outstanding (loads return in order)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + MT * (LO ? 2 : 1)) : "memory");
    __builtin_amdgcn_s_barrier();  // ... everyone's part; ring slot (s + 2) % 3 = (s - 1) % 3 is no longer read
    asm volatile("" ::: "memory");  // (the barrier intrinsic itself does not order memory operations for the compiler)
    const int s2 = s + 2 < nsteps ? s + 2 : nsteps - 1;   // past the end: refetch the last chunk (see stage)
    stage(slot == 0 ? 2 : slot - 1, s2);
    load_a(s2, an, aln);
    const char* const wb = wbuf + slot * OC_CHUNK + lane * 16;
#pragma unroll
    for (int n = 0; n < OC_NT; n++) {
      const oc_half8 bh = *reinterpret_cast<const oc_half8*>(wb + n * 1024);
      const oc_half8 bl = *reinterpret_cast<const oc_half8*>(wb + (OC_NT + n) * 1024);
#pragma unroll
      for (int t = 0; t < MT; t++) {
        const oc_half8 av = __builtin_bit_cast(oc_half8, a[t]);
        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bh, acc[t][n], 0, 0, 0);
        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bl, acc[t][n], 0, 0, 0);
        if (LO) {
          const oc_half8 alv = __builtin_bit_cast(oc_half8, al[t]);
          acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alv, bh, acc[t][n], 0, 0, 0);
          acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alv, bl, acc[t][n], 0, 0, 0);
        }
      }
    }
  };
  // nsteps = 9 KS with KS = C / 16 even (host-checked): a multiple of 6.  Six steps per trip: at the loop header the
  // compiler's wait-count pass loses the counts and drains the queue once (checked in the ISA), so the header is made rare.
#pragma unroll 1
  for (int s = 0; s < nsteps; s += 6) {
    step(s, 0, ar[0], lr_[0], ar[2], lr_[2]);
    step(s + 1, 1, ar[1], lr_[1], ar[0], lr_[0]);
    step(s + 2, 2, ar[2], lr_[2], ar[1], lr_[1]);
    step(s + 3, 0, ar[0], lr_[0], ar[2], lr_[2]);
    step(s + 4, 1, ar[1], lr_[1], ar[0], lr_[0]);
    step(s + 5, 2, ar[2], lr_[2], ar[1], lr_[1]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing fills, before the workgroup's LDS is released

  // C layout: lane (kg, lr) holds pixels 4 kg .. 4 kg + 3 of the tile for channel tile*16 + lr
#pragma unroll
  for (int t = 0; t < MT; t++) {
    const int pix0 = blockIdx.x * PIX + (w * MT + t) * 16 + kg * 4;
#pragma unroll
    for (int n = 0; n < OC_NT; n++) {
      const int ch = n * 16 + lr;
      if (ch >= p.Cout || pix0 >= HW) continue;
      const float b = bias ? bias[ch] : 0.f;
      float* dst = outp + ((size_t)e * p.Cout + ch) * HW + pix0;
      const oc_f32x4 v = acc[t][n];
      if (pix0 + 3 < HW && (HW & 3) == 0) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0] + b, v[1] + b, v[2] + b, v[3] + b);
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (pix0 + r < HW) dst[r] = v[r] + b;
      }
    }
  }
}

// ---- small grids: one or a few edges per launch (AltCorrBlock's first-edge offsets) --------------------------------
// One edge at 60 x 80 is 75 workgroups of 64 pixels for the full-resolution head and 19 for the residual head: most CUs idle,
// and the kernel above, one wave per SIMD, is paced per K step by (a) the latency of the weight chunk two steps ahead and (b)
// ~150 instructions of address arithmetic in load_a (0.55 us per step measured: 40 / 55 us per launch).  This kernel is the
// same implicit GEMM — every output element sees the same MFMA sequence, results are bit-identical — arranged for that case:
//   * the OUTPUT CHANNELS are split over blockIdx.z, OC_NTS = 2 tiles per workgroup: 4 x the workgroups, and a step's weights
//     are 4 KiB = one 1 KiB LDS-DMA per wave (fragment: part w >> 1, tile n0 + (w & 1));
//   * the ring is OC_RING_S = 8 deep = the K steps of one tap (C = 128, two source frames: KS = 8), so a chunk has seven steps
//     to arrive and one trip of the unrolled loop is one tap: the tap's pixel pointers (image bounds, zero fragment) are formed
//     once per trip and a step's pixel source is that pointer plus a compile-time offset;
//   * per step and wave: 1 weight DMA + 1 (LO: 2) pixel-fragment DMAs (no load targets a register, see OC_STAGE), a counted wait,
//     one barrier, 1 (2) + 4 LDS reads, 4 (LO: 8) MFMAs.
constexpr int OC_NTS = 2;
constexpr int OC_RING_S = 8;

template <bool LO>
__global__ __launch_bounds__(OC_WAVES* kWave) void offconv_small_kernel(const OffConvParams p) {
  constexpr int RING = OC_RING_S, KS = 8, CHUNK = 2 * OC_NTS * 1024;
  constexpr int AW = LO ? 2 : 1;                       // pixel fragments per wave and step
  constexpr int ASLOT = OC_WAVES * AW * 1024;          // bytes of pixel fragments per ring slot
  static_assert(CHUNK == OC_WAVES * 1024 && RING == KS, "one weight fragment per wave and step; one trip per tap");
  extern __shared__ float4 oc_smem[];  // RING x (CHUNK + ASLOT)
  char* const wbuf = reinterpret_cast<char*>(oc_smem);
  char* const abuf = wbuf + RING * CHUNK;
  const int lane = threadIdx.x & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int e = blockIdx.y, n0 = (int)blockIdx.z * OC_NTS;
  const int HW = p.H * p.W;
  const int lr = lane & 15, kg = lane >> 4;
  const size_t fstride = (size_t)HW * p.C;
  const size_t o1 = (size_t)p.ii[e] * fstride, o2 = (size_t)p.jj[e] * fstride;

  const int pix = blockIdx.x * (OC_WAVES * 16) + w * 16 + lr;   // this lane's A-row pixel
  const bool pv = pix < HW;
  const int pc = pv ? pix : HW - 1;
  const int py = pc / p.W, px = pc - py * p.W;

  oc_f32x4 acc[OC_NTS];
#pragma unroll
  for (int n = 0; n < OC_NTS; n++) acc[n] = oc_f32x4{0.f, 0.f, 0.f, 0.f};

  // this wave's weight fragment of a chunk (a tile past the last one — odd tile count — re-reads the last tile into an LDS
  // position whose products are never stored)
  const int wtile = n0 + (w & 1) < OC_NT ? n0 + (w & 1) : OC_NT - 1;
  const char* const wsrc = reinterpret_cast<const char*>(p.wpack) + (size_t)((w >> 1) * OC_NT + wtile) * 1024 + lane * 16;
  char* const wdst = wbuf + w * 1024;        // wave-uniform; the DMA adds lane * 16
  char* const adst = abuf + w * (AW * 1024);  // this wave's private pixel fragments
  // pixel pointers of a tap: frame ii (K steps 0..3) and frame jj (4..7); a tap outside the image points at zeros
  struct TapPtrs { const char *q1, *q2, *l1, *l2; };
  auto tap_ptrs = [&](int tap) {
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    const int yy = py + dy, xx = px + dx;
    const bool ok = pv && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
    const size_t po = ((size_t)(ok ? yy : 0) * p.W + (ok ? xx : 0)) * p.C + kg * 8;
    const char* const z = reinterpret_cast<const char*>(g_oc_zero);
    TapPtrs t;
    t.q1 = ok ? reinterpret_cast<const char*>(p.frames + o1 + po) : z;
    t.q2 = ok ? reinterpret_cast<const char*>(p.frames + o2 + po) : z;
    t.l1 = t.l2 = z;
    if (LO) {
      t.l1 = ok ? reinterpret_cast<const char*>(p.frames_lo + o1 + po) : z;
      t.l2 = ok ? reinterpret_cast<const char*>(p.frames_lo + o2 + po) : z;
    }
    return t;
  };
  // everything a step needs arrives by LDS-DMA — the weight fragment AND the pixel fragment(s) of K step ks (compile-time
  // after unrolling; 32 channels = 64 bytes further along the pixel's row).  No load targets a register, so the compiler's
  // wait-count pass has nothing pending to protect and inserts no waits of its own: with register-targeted pixel loads it
  // drained the queue (vmcnt(0)) once per tap in front of an MFMA — a full load latency with seven steps of prefetch in flight.
#define OC_STAGE(slot, s, tp, ks)                                                                                              \
  {                                                                                                                            \
    __builtin_amdgcn_global_load_lds((oc_glb_void*)(wsrc + (size_t)(s) * OC_CHUNK), (oc_lds_void*)(wdst + (slot) * CHUNK), 16, 0, 0);   \
    __builtin_amdgcn_global_load_lds((oc_glb_void*)((ks) < KS / 2 ? (tp).q1 + (ks) * 64 : (tp).q2 + ((ks) - KS / 2) * 64),      \
                                     (oc_lds_void*)(adst + (slot) * ASLOT), 16, 0, 0);                                         \
    if (LO)                                                                                                                    \
      __builtin_amdgcn_global_load_lds((oc_glb_void*)((ks) < KS / 2 ? (tp).l1 + (ks) * 64 : (tp).l2 + ((ks) - KS / 2) * 64),    \
                                       (oc_lds_void*)(adst + (slot) * ASLOT + 1024), 16, 0, 0);                                \
  }

  TapPtrs cur = tap_ptrs(0);
#pragma unroll
  for (int k = 0; k < RING - 1; k++) OC_STAGE(k, k, cur, k)
  constexpr int NSTEPS = 9 * KS;
  // fully unrolled (72 steps): at a loop header the wait-count pass loses the counts and drains the queue
#pragma unroll
  for (int tap = 0; tap < 9; tap++) {
    const TapPtrs nxt = tap_ptrs(tap < 8 ? tap + 1 : 8);   // (past the last tap: fills that nobody reads, so that every
                                                           //  step carries the same number of operations)
#pragma unroll
    for (int k = 0; k < KS; k++) {
      const int s = tap * KS + k;
      // chunk s has landed (this wave's parts) when at most the operations of steps s + 1 .. s + RING - 2 are outstanding
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * (1 + AW)) : "memory");
      __builtin_amdgcn_s_barrier();  // ... everyone's part; ring slot (s - 1) % RING is no longer read
      asm volatile("" ::: "memory");  // (the barrier intrinsic itself does not order memory operations for the compiler)
      const int s2 = s + RING - 1 < NSTEPS ? s + RING - 1 : NSTEPS - 1;
      if (k == 0) OC_STAGE(RING - 1, s2, cur, KS - 1)
      else OC_STAGE(k - 1, s2, nxt, k - 1)
      const char* const wb = wbuf + k * CHUNK + lane * 16;
      const oc_half8 av = *reinterpret_cast<const oc_half8*>(adst + k * ASLOT + lane * 16);
      oc_half8 alv;
      if (LO) alv = *reinterpret_cast<const oc_half8*>(adst + k * ASLOT + 1024 + lane * 16);
#pragma unroll
      for (int n = 0; n < OC_NTS; n++) {
        const oc_half8 bh = *reinterpret_cast<const oc_half8*>(wb + n * 1024);
        const oc_half8 bl = *reinterpret_cast<const oc_half8*>(wb + (OC_NTS + n) * 1024);
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bh, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bl, acc[n], 0, 0, 0);
        if (LO) {
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alv, bh, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alv, bl, acc[n], 0, 0, 0);
        }
      }
    }
    cur = nxt;
  }
#undef OC_STAGE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing fills, before the workgroup's LDS is released

  // C layout: lane (kg, lr) holds pixels 4 kg .. 4 kg + 3 of the tile for channel tile*16 + lr
  const int pix0 = blockIdx.x * (OC_WAVES * 16) + w * 16 + kg * 4;
#pragma unroll
  for (int n = 0; n < OC_NTS; n++) {
    const int ch = (n0 + n) * 16 + lr;
    if (n0 + n >= OC_NT || ch >= p.Cout || pix0 >= HW) continue;
    const float b = p.bias ? p.bias[ch] : 0.f;
    float* dst = p.out + ((size_t)e * p.Cout + ch) * HW + pix0;
    const oc_f32x4 v = acc[n];
    if (pix0 + 3 < HW && (HW & 3) == 0) {
      *reinterpret_cast<float4*>(dst) = make_float4(v[0] + b, v[1] + b, v[2] + b, v[3] + b);
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (pix0 + r < HW) dst[r] = v[r] + b;
    }
  }
}

// ---- per-frame partial convolutions, cached across the calls of one AltCorrBlock ----------------------------------
// The offset heads are linear in their input cat(frame ii, frame jj): conv(cat(a, b)) = conv_A(a) + conv_B(b).  A frame is
// the source of ~10 edges and the target of ~10 more, and one AltCorrBlock serves every chunk of an update_lowmem pass
// (reference factor_graph.py:272-300), so the partial results P_A[f] (+ bias) and P_B[f] are computed once per frame and
// pass, and an edge's head output is P_A[ii] + P_B[jj]: 200 frame convolutions instead of 1970 edge convolutions for the
// global BA of BASELINE config 5.  Which frames are still missing is decided on the device (no host round trip):
//   mark     every (edge, half) entry claims its frame with an atomic on the frame's flag; winners append the frame to
//            the worklist;
//   conv     offconv_frames_kernel in worklist form (grid sized for the worst case, surplus workgroups leave at once);
//   combine  out[e] = P_A[ii[e]] + P_B[jj[e]] (fixed order: bit-reproducible) and reset of the worklist counter.
__global__ void offconv_mark_kernel(const long long* __restrict__ ii, const long long* __restrict__ jj, int E, int* done,
                                    int NF, int* worklist, int* count) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * E) return;
  const int half = t >= E, e = half ? t - E : t;
  const long long f = half ? jj[e] : ii[e];
  if (f < 0 || f >= NF) return;
  if (atomicCAS(&done[half * NF + (int)f], 0, 1) == 0) worklist[atomicAdd(count, 1)] = (int)f * 2 + half;
}

__global__ void offconv_combine_kernel(const float4* __restrict__ PA, const float4* __restrict__ PB,
                                       const long long* __restrict__ ii, const long long* __restrict__ jj,
                                       float4* __restrict__ out, int n4, int* count_reset) {
  const int e = blockIdx.y;
  const float4* a = PA + (size_t)ii[e] * n4;
  const float4* b = PB + (size_t)jj[e] * n4;
  float4* o = out + (size_t)e * n4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    const float4 x = a[i], y = b[i];
    o[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
  if (count_reset && e == 0 && blockIdx.x == 0 && threadIdx.x == 0) *count_reset = 0;
}

}  // namespace lgu

extern "C" {

int lgu_offset_conv_frames_h16(const void* frames, const void* frames_lo, const long long* ii, const long long* jj,
                               const void* wpack, const float* bias, float* out, int E, int H, int W, int C, int Cout,
                               void* stream) {
  using namespace lgu;
  if (!frames || !ii || !jj || !wpack || !bias || !out) return LGU_E_BADARG;
  if (E < 0 || H < 1 || W < 1 || C < 1 || Cout < 1) return LGU_E_BADARG;
  if (C % 32 != 0 || Cout > OC_NT * 16 || E > 65535 ||
      ((reinterpret_cast<uintptr_t>(frames) | reinterpret_cast<uintptr_t>(frames_lo) | reinterpret_cast<uintptr_t>(wpack) |
        reinterpret_cast<uintptr_t>(out)) & 15) != 0)
    return LGU_E_UNSUPPORTED;
  if (E == 0) return LGU_OK;
  OffConvParams p;
  p.frames = static_cast<const _Float16*>(frames); p.frames_lo = static_cast<const _Float16*>(frames_lo); p.ii = ii; p.jj = jj;
  p.wpack = static_cast<const _Float16*>(wpack); p.bias = bias; p.out = out;
  p.E = E; p.H = H; p.W = W; p.C = C; p.Cout = Cout; p.KS = 2 * C / 32;
  p.ksh = p.KS / 2; p.worklist = nullptr; p.wcount = nullptr; p.wpack_b = nullptr; p.bias_b = nullptr; p.out_b = nullptr;
  // pixel tiles per wave: 2 (128 pixels per workgroup) unless that leaves fewer than two workgroups per CU
  const int mt = ((H * W + 127) / 128) * E >= 512 ? 2 : 1;
  const dim3 grid((H * W + 64 * mt - 1) / (64 * mt), E);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // fewer workgroups than half the CUs (one edge per call): the small-grid kernel
  const bool no_small = env_int("LGU_OFFCONV_NOSMALL", 0) != 0;   // debug switch (honoured with LGU_DEBUG_KNOBS=1 only): the full form, for A/B runs and the identity test
  if (mt == 1 && (int)(grid.x * grid.y) * 2 <= device_cu_count() && p.KS == 8 && !no_small) {
    const dim3 g3(grid.x, grid.y, (OC_NT + OC_NTS - 1) / OC_NTS);
    const size_t lds_s = (size_t)OC_RING_S * (2 * OC_NTS * 1024 + OC_WAVES * (frames_lo ? 2 : 1) * 1024);
    if (frames_lo) {
      if (lds_s > 64 * 1024) allow_max_dynamic_lds<&offconv_small_kernel<true>>();
      hipLaunchKernelGGL(offconv_small_kernel<true>, g3, dim3(OC_WAVES * kWave), lds_s, st, p);
    } else {
      if (lds_s > 64 * 1024) allow_max_dynamic_lds<&offconv_small_kernel<false>>();
      hipLaunchKernelGGL(offconv_small_kernel<false>, g3, dim3(OC_WAVES * kWave), lds_s, st, p);
    }
    return launch_status();
  }
  const size_t lds = (size_t)OC_RING * OC_CHUNK;
#define LGU_OC(LOV, MTV) hipLaunchKernelGGL((offconv_frames_kernel<LOV, MTV>), grid, dim3(OC_WAVES * kWave), lds, st, p)
  if (frames_lo) { if (mt == 2) LGU_OC(true, 2); else LGU_OC(true, 1); }
  else { if (mt == 2) LGU_OC(false, 2); else LGU_OC(false, 1); }
#undef LGU_OC
  return launch_status();
}

/* Claims the frames of ii (half 0) / jj (half 1) whose partial convolutions are still missing: done (2, NF) int32 flags
 * (0 = missing; set to 1 here), worklist (>= 2E ints) gets frame * 2 + half per claimed frame, *count their number (the
 * caller zero-initialises it once; lgu_offset_heads_combine_f32 resets it). */
int lgu_offset_heads_mark(const long long* ii, const long long* jj, int E, int* done, int NF, int* worklist, int* count,
                          void* stream) {
  using namespace lgu;
  if (!ii || !jj || !done || !worklist || !count || E < 0 || NF < 1) return LGU_E_BADARG;
  if (E == 0) return LGU_OK;
  hipLaunchKernelGGL(offconv_mark_kernel, dim3((2 * E + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ii, jj, E,
                     done, NF, worklist, count);
  return launch_status();
}

/* Partial convolutions of the worklist's frames: P_A[f] = conv_A(frames[f]) + bias for entries of half 0, P_B[f] =
 * conv_B(frames[f]) for half 1 (wpack_a / wpack_b: pack_offset_conv of the two input halves of the weight, C input
 * channels each; C % 64 == 0).  PA, PB (NF, Cout, H, W) fp32, rows of unlisted frames untouched.  maxwork >= the
 * worklist's length bound (2E): the grid is sized with it. */
int lgu_offset_conv_worklist_h16(const void* frames, const void* frames_lo, const int* worklist, const int* count, int maxwork,
                                 const void* wpack_a, const void* wpack_b, const float* bias_a, float* PA, float* PB, int H,
                                 int W, int C, int Cout, void* stream) {
  using namespace lgu;
  if (!frames || !worklist || !count || !wpack_a || !wpack_b || !bias_a || !PA || !PB) return LGU_E_BADARG;
  if (maxwork < 0 || H < 1 || W < 1 || C < 1 || Cout < 1) return LGU_E_BADARG;
  if (C % 64 != 0 || Cout > OC_NT * 16 || maxwork > 65535 ||
      ((reinterpret_cast<uintptr_t>(frames) | reinterpret_cast<uintptr_t>(frames_lo) | reinterpret_cast<uintptr_t>(wpack_a) |
        reinterpret_cast<uintptr_t>(wpack_b) | reinterpret_cast<uintptr_t>(PA) | reinterpret_cast<uintptr_t>(PB)) & 15) != 0)
    return LGU_E_UNSUPPORTED;
  if (maxwork == 0) return LGU_OK;
  OffConvParams p;
  p.frames = static_cast<const _Float16*>(frames); p.frames_lo = static_cast<const _Float16*>(frames_lo); p.ii = nullptr; p.jj = nullptr;
  p.wpack = static_cast<const _Float16*>(wpack_a); p.bias = bias_a; p.out = PA;
  p.wpack_b = static_cast<const _Float16*>(wpack_b); p.bias_b = nullptr; p.out_b = PB;
  p.E = maxwork; p.H = H; p.W = W; p.C = C; p.Cout = Cout; p.KS = C / 32; p.ksh = p.KS;
  p.worklist = worklist; p.wcount = count;
  const int mt = ((H * W + 127) / 128) * maxwork >= 512 ? 2 : 1;
  const dim3 grid((H * W + 64 * mt - 1) / (64 * mt), maxwork);
  const size_t lds = (size_t)OC_RING * OC_CHUNK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define LGU_OC(LOV, MTV) hipLaunchKernelGGL((offconv_frames_kernel<LOV, MTV>), grid, dim3(OC_WAVES * kWave), lds, st, p)
  if (frames_lo) { if (mt == 2) LGU_OC(true, 2); else LGU_OC(true, 1); }
  else { if (mt == 2) LGU_OC(false, 2); else LGU_OC(false, 1); }
#undef LGU_OC
  return launch_status();
}

/* out[e] = PA[ii[e]] + PB[jj[e]], rows of n floats (n % 4 == 0, 16-byte aligned buffers); *count_reset = 0 if given. */
int lgu_offset_heads_combine_f32(const float* PA, const float* PB, const long long* ii, const long long* jj, float* out, int E,
                                 int n, int* count_reset, void* stream) {
  using namespace lgu;
  if (!PA || !PB || !ii || !jj || !out || E < 0 || n < 4) return LGU_E_BADARG;
  if (n % 4 != 0 || E > 65535 ||
      ((reinterpret_cast<uintptr_t>(PA) | reinterpret_cast<uintptr_t>(PB) | reinterpret_cast<uintptr_t>(out)) & 15) != 0)
    return LGU_E_UNSUPPORTED;
  if (E == 0) return LGU_OK;
  const int n4 = n / 4;
  int bx = (n4 + 255) / 256;
  bx = bx > 64 ? 64 : bx;
  hipLaunchKernelGGL(offconv_combine_kernel, dim3(bx, E), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(PA), reinterpret_cast<const float4*>(PB), ii, jj, reinterpret_cast<float4*>(out), n4,
                     count_reset);
  return launch_status();
}

}  // extern "C"
