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
// Workgroup = 4 waves x 2 pixel tiles = 128 pixels.  A (pixels) comes straight from the channel-last frame buffers —
// 16 contiguous bytes per lane, frame ii[e] for channels 0-127 and jj[e] for 128-255, zero outside the image — no
// gather, no concatenation, no cast.  B (weights) is prepacked on the host in MFMA fragment order, 14 KiB per K step
// (2 parts x 7 tiles x 1 KiB), streamed into a double-buffered LDS area by LDS-DMA one step ahead and read back
// lane-linearly.  Output (E, 98, H, W) fp32 with the bias added: a lane owns 4 consecutive pixels of one channel = one
// 16-byte store.  The residual head (corr.py:219-220: ofs_residual on the 2 x 2 average of that input) is the same
// kernel over frames pooled once per block, with the input in two half parts as well (template LO).
#include "lgu_common.hpp"

namespace lgu {

constexpr int OC_NT = 7;                      // output-channel tiles (<= 112 channels)
constexpr int OC_CHUNK = 2 * OC_NT * 1024;    // bytes of weight fragments per K step (hi + lo)
constexpr int OC_WAVES = 4, OC_MT = 2;        // waves per workgroup, pixel tiles per wave
constexpr int OC_PIX = OC_WAVES * OC_MT * 16; // pixels per workgroup

typedef _Float16 oc_half8 __attribute__((ext_vector_type(8)));
typedef float oc_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void oc_lds_void;
typedef const __attribute__((address_space(1))) void oc_glb_void;

struct OffConvParams {
  const _Float16* frames;  // (NF, H, W, C) channel-last, C = 128
  const _Float16* frames_lo;  // LO kernels: second half part of the input (input = frames + frames_lo), same layout
  const long long* ii;     // (E) frame of channels 0..C-1
  const long long* jj;     // (E) frame of channels C..2C-1
  const _Float16* wpack;   // [9][KS][2][OC_NT][64][8] halves, KS = 2C / 32
  const float* bias;       // (Cout)
  float* out;              // (E, Cout, H, W)
  int E, H, W, C, Cout, KS;
};

// LO: the input has two half parts (e.g. 2 x 2 averages of half values, which need up to 24 bits): x = hi + lo and
// x . W' = hi . whi + hi . wlo + lo . whi + lo . wlo.
template <bool LO>
__global__ __launch_bounds__(OC_WAVES* kWave) void offconv_frames_kernel(const OffConvParams p) {
  extern __shared__ float4 oc_smem[];  // 2 x OC_CHUNK
  char* const wbuf = reinterpret_cast<char*>(oc_smem);
  const int lane = threadIdx.x & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int e = blockIdx.y;
  const int HW = p.H * p.W;
  const int lr = lane & 15, kg = lane >> 4;
  const size_t fstride = (size_t)HW * p.C;
  const size_t o1 = (size_t)p.ii[e] * fstride, o2 = (size_t)p.jj[e] * fstride;
  const int ksh = p.KS >> 1;  // K steps per frame

  // this lane's A-row pixel in each of the wave's tiles
  int py[OC_MT], px[OC_MT];
  bool pv[OC_MT];
#pragma unroll
  for (int t = 0; t < OC_MT; t++) {
    const int pix = blockIdx.x * OC_PIX + (w * OC_MT + t) * 16 + lr;
    pv[t] = pix < HW;
    py[t] = pix / p.W;
    px[t] = pix - py[t] * p.W;
  }

  oc_f32x4 acc[OC_MT][OC_NT];
#pragma unroll
  for (int t = 0; t < OC_MT; t++)
#pragma unroll
    for (int n = 0; n < OC_NT; n++) acc[t][n] = oc_f32x4{0.f, 0.f, 0.f, 0.f};

  const int nsteps = 9 * p.KS;
  // weight chunk `s` -> LDS buffer s & 1: 14 KiB = 3.5 x (256 threads x 16 bytes), contiguous in wpack
  auto stage = [&](int s) {
    const char* src = reinterpret_cast<const char*>(p.wpack) + (size_t)s * OC_CHUNK + (size_t)w * 1024 + lane * 16;
    char* dst = wbuf + (s & 1) * OC_CHUNK + w * 1024;  // wave-uniform; the DMA adds lane * 16
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k * 4096 + w * 1024 < OC_CHUNK)
        __builtin_amdgcn_global_load_lds((oc_glb_void*)(src + k * 4096), (oc_lds_void*)(dst + k * 4096), 16, 0, 0);
  };
  auto load_a = [&](int s, oc_half8 (&a)[OC_MT], oc_half8 (&al)[OC_MT]) {
    const int tap = s / p.KS, ks = s - tap * p.KS;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    const size_t fo = (ks < ksh ? o1 : o2) + (ks < ksh ? ks : ks - ksh) * 32 + kg * 8;
#pragma unroll
    for (int t = 0; t < OC_MT; t++) {
      const int yy = py[t] + dy, xx = px[t] + dx;
      const bool ok = pv[t] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      a[t] = oc_half8{0, 0, 0, 0, 0, 0, 0, 0};
      if (LO) al[t] = a[t];
      if (ok) {
        const size_t off = fo + ((size_t)yy * p.W + xx) * p.C;
        a[t] = *reinterpret_cast<const oc_half8*>(p.frames + off);
        if (LO) al[t] = *reinterpret_cast<const oc_half8*>(p.frames_lo + off);
      }
    }
  };

  oc_half8 a_cur[OC_MT], a_nxt[OC_MT], l_cur[OC_MT], l_nxt[OC_MT];
  stage(0);
  load_a(0, a_cur, l_cur);
  for (int s = 0; s < nsteps; s++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // chunk s has landed (this wave's part) and a_cur is here
    __syncthreads();                                  // ... everyone's part; buffer (s + 1) & 1 is no longer read
    if (s + 1 < nsteps) {
      stage(s + 1);
      load_a(s + 1, a_nxt, l_nxt);
    }
    const char* const wb = wbuf + (s & 1) * OC_CHUNK + lane * 16;
#pragma unroll
    for (int n = 0; n < OC_NT; n++) {
      const oc_half8 bh = *reinterpret_cast<const oc_half8*>(wb + n * 1024);
      const oc_half8 bl = *reinterpret_cast<const oc_half8*>(wb + (OC_NT + n) * 1024);
#pragma unroll
      for (int t = 0; t < OC_MT; t++) {
        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur[t], bh, acc[t][n], 0, 0, 0);
        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur[t], bl, acc[t][n], 0, 0, 0);
        if (LO) {
          acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(l_cur[t], bh, acc[t][n], 0, 0, 0);
          acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(l_cur[t], bl, acc[t][n], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < OC_MT; t++) {
      a_cur[t] = a_nxt[t];
      if (LO) l_cur[t] = l_nxt[t];
    }
  }

  // C layout: lane (kg, lr) holds pixels 4 kg .. 4 kg + 3 of the tile for channel tile*16 + lr
#pragma unroll
  for (int t = 0; t < OC_MT; t++) {
    const int pix0 = blockIdx.x * OC_PIX + (w * OC_MT + t) * 16 + kg * 4;
#pragma unroll
    for (int n = 0; n < OC_NT; n++) {
      const int ch = n * 16 + lr;
      if (ch >= p.Cout || pix0 >= HW) continue;
      const float b = p.bias[ch];
      float* dst = p.out + ((size_t)e * p.Cout + ch) * HW + pix0;
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
  const dim3 grid((H * W + OC_PIX - 1) / OC_PIX, E);
  if (frames_lo)
    hipLaunchKernelGGL(offconv_frames_kernel<true>, grid, dim3(OC_WAVES * kWave), 2 * OC_CHUNK, reinterpret_cast<hipStream_t>(stream), p);
  else
    hipLaunchKernelGGL(offconv_frames_kernel<false>, grid, dim3(OC_WAVES * kWave), 2 * OC_CHUNK, reinterpret_cast<hipStream_t>(stream), p);
  return launch_status();
}

}  // extern "C"
