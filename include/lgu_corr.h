/*
 * lgu_corr.h — C ABI of the MI355X (gfx950) deformable correlation-sampling library.
 *
 * This is the drop-in boundary for LGU-SLAM's hot path: every entry point below
 * replaces one Python-visible operator of the reference's two CUDA extensions
 * (`defCorrSample`, reference offersample_LGS/droid.cpp:138-147, and the two
 * `altcorr_*` operators of `droid_backends`, reference src/droid.cpp:246-247).
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer into a dense, contiguous fp32 buffer owned by
 *     the caller (PyTorch's allocator in practice); the library never allocates,
 *     frees or retains memory;
 *   - the work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     legacy default stream, which is what the reference launches on);
 *   - the return value is 0 on success, a hipError_t value if the launch failed, or
 *     one of the LGU_E_* codes below for arguments the kernels cannot serve;
 *     nothing throws; lgu_error_string() names any code;
 *   - re-entrant; no mutable global state beyond idempotent per-device launch caches ("this kernel may use > 64 KB of
 *     LDS on device d", the device's CU count: the same values whoever writes them) and one flag read when the library
 *     is loaded: the LGU_* debug environment variables that select superseded kernels for tests / A-B tools are
 *     honoured only if LGU_DEBUG_KNOBS=1 was set at load time — otherwise no entry point reads the environment;
 *   - "fully written" outputs need no initialisation by the caller; "accumulated"
 *     outputs must be zero-filled by the caller before the call (the reference
 *     allocates them with torch::zeros / zeros_like).
 *
 * Index conventions follow the reference: tap index i moves in x (width), j in y
 * (height); rd = 2*radius+1.
 */
#ifndef LGU_CORR_H
#define LGU_CORR_H

#ifdef __cplusplus
extern "C" {
#endif

#define LGU_OK 0
#define LGU_E_BADARG 100001     /* null pointer / non-positive size / radius out of range */
#define LGU_E_UNSUPPORTED 100002 /* shape the kernels do not serve (e.g. C % 32 != 0) */

#define LGU_MAX_LEVELS 8
#define LGU_MAX_RADIUS 7

/* Library identification: "lgu_corr <semver> gfx950", followed by " [<extra compiler flags>]" for a non-default
 * (experiment) build. */
const char* lgu_version(void);
/* 1 if the library was loaded with LGU_DEBUG_KNOBS=1 in the environment (debug / A-B knobs live), else 0. */
int lgu_debug_knobs_enabled(void);
/* Static string for a code returned by any entry point (LGU_E_* or hipError_t). */
const char* lgu_error_string(int code);

/* ---- volume path -------------------------------------------------------------- */

/* defCorrSample.defCorr_index_forward   (reference offersample_LGS/droid.cpp:53-63,
 * defCorrSample_kernel.cu:25-91,165-196).
 *   volume (E,H1,W1,H2,W2)  coords (E,2,H1,W1)  offset (E,H1,W1,rd,rd,2) IN/OUT
 *   corr   (E,rd,rd,H1,W1)  fully written (masked taps are written as 0).
 * Side effect kept from the reference: offset[e][y][x][r][r][0:2] = 0. */
int lgu_defcorr_fwd_f32(const float* volume, const float* coords, float* offset, float* corr,
                        int E, int H1, int W1, int H2, int W2, int radius, void* stream);

/* defCorrSample.defCorr_index_backward  (droid.cpp:65-77, defCorrSample_kernel.cu:93-162,198-231).
 *   corr_grad (E,rd,rd,H1,W1); volume_grad like volume — ACCUMULATED (caller zero-fills);
 *   offset_grad like offset — fully written (0 for masked taps). offset centre re-zeroed. */
int lgu_defcorr_bwd_f32(const float* volume, const float* coords, float* offset,
                        const float* corr_grad, float* volume_grad, float* offset_grad,
                        int E, int H1, int W1, int H2, int W2, int radius, void* stream);

/* defCorrSample.corr_index_forward      (droid.cpp:79-87, corrSample_kernel.cu:24-82,139-168). */
int lgu_corridx_fwd_f32(const float* volume, const float* coords, float* corr,
                        int E, int H1, int W1, int H2, int W2, int radius, void* stream);

/* defCorrSample.corr_index_backward     (droid.cpp:89-99, corrSample_kernel.cu:84-136,170-199).
 *   volume_grad ACCUMULATED (caller zero-fills). `volume` is accepted for signature
 *   fidelity and never read (the reference kernel does not read it either). */
int lgu_corridx_bwd_f32(const float* volume, const float* coords, const float* corr_grad,
                        float* volume_grad,
                        int E, int H1, int W1, int H2, int W2, int radius, void* stream);

/* defCorrSample.gaussianMask            (droid.cpp:100-110, gaussianAttn.cu:19-68,134-163).
 *   means, covs (E,H1,W1,2); volume, volume1 (E,H1,W1,H2,W2); volume1 fully written
 *   (zero outside the (2*radius+1)^2 window around floor(mean)). */
int lgu_gaussmask_fwd_f32(const float* means, const float* covs, const float* volume,
                          float* volume1,
                          int E, int H1, int W1, int H2, int W2, int radius, void* stream);

/* defCorrSample.gaussianMask_backward   (droid.cpp:112-123, gaussianAttn.cu:72-131,165-200).
 *   means_grad, covs_grad (E,H1,W1,2) fully written. */
int lgu_gaussmask_bwd_f32(const float* means, const float* covs, const float* volume,
                          const float* volume1_grad, float* means_grad, float* covs_grad,
                          int E, int H1, int W1, int H2, int W2, int radius, void* stream);

/* Fused multi-level deformable sample = the body of CorrBlock.__call__
 * (reference droid_slam/modules/corr.py:88-109): L calls of defCorr_index_forward on the
 * L pyramid levels with coords / 2^l, written straight into the concatenated tensor
 *   out (E, L*rd*rd, H1, W1)   channel = l*rd*rd + i*rd + j     (corr.py:103,109).
 * volumes[l] (E,H1,W1,H2[l],W2[l]); offsets[l] (E,H1,W1,rd,rd,2) IN/OUT (centre zeroed) or
 * NULL = structurally zero offsets for that level (corr.py:132-135), nothing is read.
 * The pointer/size tables are HOST arrays of length L (copied at launch).
 * flags: LGU_PYR_PROBE fuses the uncertainty probe of corr.py:94-99 as well: the 3x3
 *   plain sample of level 1 at coords/2, its unbiased variance over the 9 taps,
 *   mask = sigmoid(var), offsets[1] *= mask written back (the reference's stateful
 *   update) before level 1 is sampled. Requires L >= 2 and offsets[1] != NULL. */
#define LGU_PYR_PROBE 1
/* LGU_PYR_TILED: volumes[l] are in this library's tiled slice layout instead of the reference's row-major
 *   H2 x W2 slices: every (edge,pixel) slice is stored as 4 x 8 element tiles (one 128-byte line each), tiles
 *   row-major over the slice padded to multiples of (4, 8):
 *     pos(y, x) = ((y/4) * ceil(W2/8) + x/8) * 32 + (y%4) * 8 + x%8,   slice pitch = ceil(H2/4)*ceil(W2/8)*32 floats.
 *   HBM is fetched in whole 128-byte lines; a pixel's tap footprint touches ~40 % fewer lines in this layout.
 *   H2[l], W2[l] stay the LOGICAL sizes.  Produced by lgu_volume_pyramid_tiled_f32 / lgu_volume_retile_f32.
 *   Served for radius 3 (the production case); otherwise LGU_E_UNSUPPORTED.  Results are bit-identical to the
 *   reference layout. */
#define LGU_PYR_TILED 2
/* LGU_PYR_COORDS_LAST: coords is (E,H1,W1,2) with x, y interleaved — how factor_graph.py holds them — instead of the
 *   operator's (E,2,H1,W1) planes: saves the permute().contiguous() pass of corr.py:91 in front of every lookup.
 *   Served by the fast kernels (radius 1..3, 16-byte aligned operands); otherwise LGU_E_UNSUPPORTED. */
#define LGU_PYR_COORDS_LAST 4
/* LGU_PYR_OUT_NHWC: out is channel-last, (E, H1, W1, L*rd*rd) — the memory format the consumer of the lookup, the 1x1
 *   convolution of UpdateModule.corr_encoder (droid_net.py:76-80), prefers — instead of (E, L*rd*rd, H1, W1); same
 *   values.  LGU_PYR_OUT_F16 (with LGU_PYR_OUT_NHWC only, else LGU_E_BADARG): out holds IEEE half, each value the
 *   round-to-nearest-even of the fp32 result, i.e. exactly the cast autocast applies in front of that convolution
 *   (factor_graph.py wraps the update operator in autocast); `out` then points at E*H1*W1*L*rd*rd 2-byte elements.
 *   Served by the production kernel (LGU_PYR_TILED, radius 3); otherwise LGU_E_UNSUPPORTED. */
#define LGU_PYR_OUT_NHWC 8
#define LGU_PYR_OUT_F16 16
int lgu_defcorr_pyramid_fwd_f32(const float* const* volumes, const float* coords,
                                float* const* offsets, float* out,
                                int L, int E, int H1, int W1, const int* H2, const int* W2,
                                int radius, int flags, void* stream);

/* The same with a storage indirection for the volumes: edge e's slices are at slot edge_slot[e] (device array of E
 * int32) of the level buffers, which may hold more slots than E.  This is what lets the CorrBlock state container
 * implement `cat` (factor_graph.py:123) and `__getitem__` (:139) by editing a small index array instead of copying the
 * multi-GB pyramid as the reference's tensor concatenation / boolean indexing do.  Offsets, coords and out are indexed by e as before.
 * Served by the fast kernels; otherwise LGU_E_UNSUPPORTED. */
int lgu_defcorr_pyramid_slots_fwd_f32(const float* const* volumes, const int* edge_slot, const float* coords,
                                      float* const* offsets, float* out,
                                      int L, int E, int H1, int W1, const int* H2, const int* W2,
                                      int radius, int flags, void* stream);

/* The lookup with its consumer fused in: lgu_defcorr_pyramid_slots_fwd_f32 followed by the first layer of
 * UpdateModule.corr_encoder (reference droid_slam/droid_net.py:76-77,116: Conv2d(L*rd*rd, 128, 1) + ReLU, evaluated
 * under autocast: half inputs and weights, fp32 accumulation, half result) on the matrix cores, in the same launch:
 *   out (E, H1, W1, enc_n) IEEE half, channel-last = relu(W1 . half(samples) + b1) per pixel
 * so the L*rd*rd samples of a pixel never go to HBM.
 *   enc_w  half (enc_n, Kp): the convolution weight (enc_n, L*rd*rd, 1, 1) with every row zero-padded to
 *          Kp = ceil(L*rd*rd / 32) * 32 entries, 16-byte aligned;   enc_b  half (enc_n).
 * edge_slot may be NULL (slot e = edge e).  flags: LGU_PYR_TILED is required; LGU_PYR_PROBE and LGU_PYR_COORDS_LAST as
 * above; the LGU_PYR_OUT_* flags do not apply (LGU_E_BADARG).  Served for radius 3, enc_n = 128, zero-offset patterns
 * the sampler handles in one launch (none / all / levels >= 2); otherwise LGU_E_UNSUPPORTED.  Offsets get the same in-place side
 * effects as in the unfused entry. */
int lgu_defcorr_pyramid_enc_fwd_f32(const float* const* volumes, const int* edge_slot, const float* coords,
                                    float* const* offsets, const void* enc_w, const void* enc_b, void* out,
                                    int L, int E, int H1, int W1, const int* H2, const int* W2,
                                    int radius, int enc_n, int flags, void* stream);

/* Tail of GaussianMask.gaussian_parameters (reference droid_slam/gaussianMask_cuda.py:69-83) after the two linear heads:
 *   mean_ofs, cov_raw (E, H*W, 2) fp32 or half (is_half: the heads ran under autocast; steps rounded to half like the
 *   framework's half kernels);  mean (E,H,W,2) fp32 = pixel grid (x, y) + mean_ofs;  cov (E,H,W,2) fp32 =
 *   sigmoid(per-sample standardised cov_raw) * 5 + 0.05;  det (E, H*W) = cov.x * cov.y in the input's dtype. */
int lgu_gaussian_params(const void* mean_ofs, const void* cov_raw, float* mean, float* cov, void* det,
                        int E, int H, int W, int is_half, float eps, void* stream);

/* The uncertainty mask of AltCorrBlock.corr_fn (reference droid_slam/modules/corr.py:203-207) applied in place:
 *   probe (E, T, H*W): the T = 9 plain level-1 samples of every pixel (altcorr_forward, radius 1);
 *   offset (E, H*W, C) IN/OUT: offset[e][p][:] *= sigmoid(unbiased variance of probe[e][:][p]). */
int lgu_probe_mask_scale_f32(const float* probe, float* offset, int E, int HW, int T, int C, void* stream);

/* The level-0 offset head of AltCorrBlock.corr_fn (reference droid_slam/modules/corr.py:174-189, :220:
 * ofsMap(cat(fmap[ii] * 4, fmap[jj] * 4).float()), a Conv2d(2C, Cout, 3, padding=1) evaluated in fp32) straight from the
 * stored half frame buffers, on the half matrix cores with fp32-accurate weights:
 *   frames (NF, H, W, C) half, channel-last (AltCorrBlock.pyramid[0], = fmap / 4);  ii, jj (E) int64 frame indices;
 *   wpack: the weight times 4 (exact), split into two half parts hi + lo (22 significant bits) and laid out in MFMA
 *          fragment order [9 taps][2C/32][hi, lo][7][64 lanes][8] (lgu_slam_amd.ops.pack_offset_conv builds it);
 *   bias (Cout) fp32;  out (E, Cout, H, W) fp32 fully written.
 * Half x half products are exact in fp32 and accumulation is fp32, so the result equals the fp32 convolution up to the
 * 2^-22 truncation of the weights (below that convolution's own summation noise).  frames_lo (may be NULL): a second
 * half part of the input, same layout — input = frames + frames_lo — for inputs that need up to 24 bits (the residual
 * head of :219-220 takes 2 x 2 averages of the frames; split as hi = half(x), lo = half(x - hi)).  C % 32 == 0,
 * Cout <= 112; otherwise LGU_E_UNSUPPORTED. */
int lgu_offset_conv_frames_h16(const void* frames, const void* frames_lo, const long long* ii, const long long* jj,
                               const void* wpack, const float* bias, float* out, int E, int H, int W, int C, int Cout,
                               void* stream);

/* The same heads through per-FRAME partial convolutions (csrc/offconv.hip).  The convolution is linear in its input
 * cat(frame ii, frame jj): conv(cat(a, b)) = conv_A(a) + conv_B(b), a frame is source / target of ~10 edges each, and one
 * AltCorrBlock serves every chunk of an update_lowmem pass (reference factor_graph.py:272-300): P_A[f] = conv_A(frames[f])
 * + bias and P_B[f] = conv_B(frames[f]) are computed once per frame and kept, an edge's output is P_A[ii] + P_B[jj].
 *   lgu_offset_heads_mark         claims, on the device, the frames of ii (half 0) / jj (half 1) whose partials are
 *                                 missing: done (2, NF) int32 flags (0 = missing, set to 1), worklist (>= 2E ints)
 *                                 receives frame * 2 + half per claimed frame, *count their number (zero before the first
 *                                 call; reset by the combine).
 *   lgu_offset_conv_worklist_h16  the partial convolutions of the worklist: wpack_a / wpack_b = the weight's two input halves
 *                                 packed like wpack above (C input channels each, C % 64 == 0), bias added to P_A only;
 *                                 PA, PB (NF, Cout, H, W) fp32; maxwork >= the worklist's possible length (sizes the grid).
 *   lgu_offset_heads_combine_f32  out[e] = PA[ii[e]] + PB[jj[e]] over rows of n floats (n % 4 == 0); *count_reset = 0. */
int lgu_offset_heads_mark(const long long* ii, const long long* jj, int E, int* done, int NF, int* worklist, int* count,
                          void* stream);
int lgu_offset_conv_worklist_h16(const void* frames, const void* frames_lo, const int* worklist, const int* count, int maxwork,
                                 const void* wpack_a, const void* wpack_b, const float* bias_a, float* PA, float* PB, int H,
                                 int W, int C, int Cout, void* stream);
int lgu_offset_heads_combine_f32(const float* PA, const float* PB, const long long* ii, const long long* jj, float* out, int E,
                                 int n, int* count_reset, void* stream);

/* Post-processing of the learned sampling offsets (reference droid_slam/modules/corr.py:117-135 and :217-235 with
 * per_Corr_Normalization, gaussianMask_cuda.py:26-33) in one pass:
 *   o0 (E,C,H,W), o1 (E,C,Hl,Wl): the outputs of the two offset convolutions (o1 still at the pooled resolution; the
 *   nearest-neighbour upsampling to (H,W) is an index map inside the kernel), fp32 or IEEE half (is_half);
 *   out0 = 4 tanh((o0 - mean) / sqrt(var + eps)),  out1 = (4 tanh((o1 - mean1) / sqrt(var1 + eps)) + out0) / 2,
 *   statistics per edge over (C,H,W), biased variance; both written as (E,H,W,C) fp32 — the (E,H,W,rd,rd,2) tensors
 *   the samplers take.  is_half = 1: every step is rounded to half as the framework's half kernels do; is_half = 2: the
 *   same for level 0, level 1 in fp32 — what autocast yields, which promotes the nearest upsampling to fp32.
 * scratch: lgu_offsets_finalize_scratch_bytes(E) bytes of device memory (partial sums; no initialisation needed). */
long long lgu_offsets_finalize_scratch_bytes(int E);
int lgu_offsets_finalize(const void* o0, const void* o1, float* out0, float* out1, void* scratch,
                         int E, int C, int H, int W, int Hl, int Wl, int is_half, float eps, void* stream);
/* The same with the uncertainty mask of AltCorrBlock.corr_fn (reference corr.py:203-207) folded in: probe (E, T, H, W) fp32,
 * the T >= 2 plain level-1 samples of every pixel; out1 = level-1 offsets * sigmoid(unbiased variance over T) — bit for bit
 * lgu_offsets_finalize followed by lgu_probe_mask_scale_f32, without the extra read-modify-write pass over out1. */
int lgu_offsets_finalize_masked(const void* o0, const void* o1, const float* probe, int T, float* out0, float* out1,
                                void* scratch, int E, int C, int H, int W, int Hl, int Wl, int is_half, float eps,
                                void* stream);

/* Fused volume post-processing of CorrBlock.__init__ (reference droid_slam/gaussianMask_cuda.py:84-86
 * and droid_slam/modules/corr.py:79-86): in ONE pass over the raw all-pairs volume
 *   level0 = gaussianMask(means, covs, volume, radius) / (6.28*sqrt(covs.x*covs.y)) + volume
 *   level l = avg_pool2d(level l-1, 2, stride 2) over the target dims, l = 1..L-1
 * levels[l] (E,H1,W1,H2>>l,W2>>l) fully written; levels[0] may alias `volume` (in place).
 * `levels` is a HOST array of L device pointers.  Requires W2 % 4 == 0 and a slice pyramid
 * that fits LDS (<= 96 KiB); otherwise LGU_E_UNSUPPORTED and the caller composes the ops. */
int lgu_volume_pyramid_f32(const float* means, const float* covs, const float* volume, float* const* levels, int L,
                           int E, int H1, int W1, int H2, int W2, int radius, void* stream);

/* Same as lgu_volume_pyramid_f32 but every level is written in the tiled slice layout (LGU_PYR_TILED above).
 * levels[0] may alias `volume` when H2 % 4 == 0 and W2 % 8 == 0 (the slice is converted in place). */
int lgu_volume_pyramid_tiled_f32(const float* means, const float* covs, const float* volume, float* const* levels,
                                 int L, int E, int H1, int W1, int H2, int W2, int radius, void* stream);

/* The same over a HALF raw volume (what the all-pairs product of the half feature maps returns, corr.py:145-152): the
 * `.float()` of corr.py:64 becomes this kernel's load (exact) instead of a pass of its own over the volume.  Levels are
 * fp32 as above; tiled != 0 writes them in the tiled slice layout.  levels[0] cannot alias `volume`. */
int lgu_volume_pyramid_h16(const float* means, const float* covs, const void* volume, float* const* levels, int L,
                           int E, int H1, int W1, int H2, int W2, int radius, int tiled, void* stream);

/* The same builder with the determinant handed over: det (E*H1*W1) is what GaussianMask.forward computes as
 * `det = cov[:,:,0] * cov[:,:,1]` (gaussianMask_cuda.py:79), fp32 or IEEE half (det_half != 0).  Inside the reference's
 * autocast region (factor_graph.py:90) det IS a half tensor, and the denominator `6.28 * sqrt(det)` (:85) is rounded to half after
 * the square root and after the product; a half det reproduces those roundings, an fp32 det is the fp32 evaluation
 * (= lgu_volume_pyramid_f32 / _tiled_f32 / _h16, which form det = cov0 * cov1 themselves).  volume_half / tiled as above. */
int lgu_volume_pyramid_det(const float* means, const float* covs, const void* det, int det_half, const void* volume,
                           int volume_half, float* const* levels, int L, int E, int H1, int W1, int H2, int W2, int radius,
                           int tiled, void* stream);

/* CorrBlock.__init__'s volume, BUILT into the tiled pyramid (reference droid_slam/modules/corr.py:145-152 matmul of the
 * two feature maps / 4 each, :64 .float(), gaussianMask_cuda.py:84-86 Gaussian re-weighting and "/ denominator + corr",
 * corr.py:79-86 three average poolings) in ONE launch on the fp32 matrix cores: the raw all-pairs volume never reaches HBM.
 *   fmap1, fmap2 (E, C, H, W) fp32, the reference's NCHW maps (un-scaled: the kernel applies the / 16 exactly)
 *   means, covs (E, H, W, 2), det (E*H*W) fp32 / half (det_half) or NULL: as lgu_volume_pyramid_det
 *   levels[l] (E, H, W, <tiled slice of (H >> l, W >> l)>) fp32, fully written incl. the slices' zero padding; L must be 4
 * Served: H % 8 == 0, W in {16, 32, 64}, C % 16 == 0; anything else LGU_E_UNSUPPORTED (callers take the library GEMM +
 * lgu_volume_pyramid_*).  Equal to that composition up to the GEMM's fp32 summation order. */
int lgu_volume_build_pyramid_f32(const float* fmap1, const float* fmap2, const float* means, const float* covs, const void* det,
                                 int det_half, float* const* levels, int L, int E, int C, int H, int W, int radius, void* stream);

/* The same for HALF feature maps (the reference under autocast, factor_graph.py:90: corr.py:145-152 is then a half GEMM —
 * exact half x half products, fp32 accumulation, one rounding of each sum to half, which this kernel applies before the
 * .float() of corr.py:64).
 *   feats (E, H, W, 2C) half, channel-last, the source map's C channels first then the target map's (un-scaled;
 *   CorrBlock's `t`, corr.py:57-62)
 *   workspace: E*H*W*2C halves of scratch, 16-byte aligned, distinct from feats (a first small launch re-orders the maps
 *   into MFMA fragment order there; contents afterwards unspecified)
 * Served: H % 8 == 0, W in {16, 32, 64}, C % 32 == 0.  Differs from the library half GEMM + lgu_volume_pyramid_det only where
 * a different fp32 summation order moves a sum across a half rounding boundary (one half ulp of that raw product). */
int lgu_volume_build_pyramid_h16(const void* feats, void* workspace, const float* means, const float* covs, const void* det,
                                 int det_half, float* const* levels, int L, int E, int C, int H, int W, int radius, void* stream);

/* Layout conversion of `nslices` slices of H2 x W2 floats: to_tiled != 0: row-major -> tiled, else tiled -> row-major
 * (padding elements of the tiled form are written as 0).  src and dst must not overlap. */
int lgu_volume_retile_f32(const float* src, float* dst, long long nslices, int H2, int W2, int to_tiled, void* stream);

/* ---- low-memory (on-the-fly correlation) path ---------------------------------- */

/* The channel contraction of the two forward operators below runs on the matrix cores for radius 1..3 and
 * C in {16, 32, 64, 128} (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 fmaf-chain accumulation; the channel
 * summation order differs from the reference's, results agree to fp32 rounding, tests: 1e-5); other shapes take
 * VALU kernels.
 *
 * defCorrSample.lowMem_defSample        (droid.cpp:124-136, lowMem_defSample.cu:27-134,137-168).
 *   fmap1 (B,H1,W1,C)  fmap2 (B,H2,W2,C)  coords (B,S,H1,W1,2) [x,y interleaved]
 *   offset (NO,H1,W1,rd,rd,2) IN/OUT — indexed with b*s exactly as the reference does
 *   (lowMem_defSample.cu:80-83), i.e. offset[0] for every b when S == 1;
 *   corr (B,S,rd,rd,H1,W1) fully written. Requires C % 32 == 0 and (B-1)*(S-1) < NO. */
int lgu_lowmem_defsample_fwd_f32(const float* fmap1, const float* fmap2, const float* coords,
                                 float* offset, float* corr,
                                 int B, int S, int H1, int W1, int H2, int W2, int C, int NO,
                                 int radius, void* stream);

/* droid_backends.altcorr_forward        (src/droid.cpp:193-203, src/altcorr_kernel.cu:27-149,290-319).
 *   corr (B,S,rd*rd,H1,W1), channel = ix*rd + iy, fully written. Requires C % 32 == 0. */
int lgu_altcorr_fwd_f32(const float* fmap1, const float* fmap2, const float* coords, float* corr,
                        int B, int S, int H1, int W1, int H2, int W2, int C,
                        int radius, void* stream);

/* droid_backends.altcorr_backward       (src/droid.cpp:205-217, src/altcorr_kernel.cu:152-286,321-356).
 *   fmap1_grad (B,H1,W1,C) fully written; fmap2_grad (B,H2,W2,C) ACCUMULATED with float
 *   atomics (caller zero-fills); coords_grad is never written by the reference and is
 *   not part of this ABI (the host shim returns zeros). */
int lgu_altcorr_bwd_f32(const float* fmap1, const float* fmap2, const float* coords,
                        const float* corr_grad, float* fmap1_grad, float* fmap2_grad,
                        int B, int S, int H1, int W1, int H2, int W2, int C,
                        int radius, void* stream);

/* Mixed-precision forms of the two low-memory forward operators: feature maps in IEEE half
 * (as droid_slam/depth_video.py stores them), everything else — coords, offsets, products,
 * sums, output — fp32.  The reference call sites (droid_slam/modules/corr.py:202,209) convert with
 * .float() and run the fp32 operator; these entries skip the conversion passes and, for
 * C in {32, 64, 128, 256}, run the channel contraction on the matrix cores
 * (v_mfma_f32_16x16x32_f16: half products are exact in fp32, accumulation is fp32), so the result
 * equals the _f32 entry on fmap.float() up to fp32 summation order (<= 1e-5; measured 2.4e-7).
 * Other C (multiples of 16) take a VALU kernel that is bit-identical to the _f32 entry.
 * Same shapes/side effects as the _f32 forms; radius in 1..3, otherwise LGU_E_UNSUPPORTED. */
int lgu_lowmem_defsample_fwd_h16(const void* fmap1_half, const void* fmap2_half, const float* coords,
                                 float* offset, float* corr,
                                 int B, int S, int H1, int W1, int H2, int W2, int C, int NO,
                                 int radius, void* stream);
int lgu_altcorr_fwd_h16(const void* fmap1_half, const void* fmap2_half, const float* coords, float* corr,
                        int B, int S, int H1, int W1, int H2, int W2, int C,
                        int radius, void* stream);

/* The per-level loop of AltCorrBlock.corr_fn (reference droid_slam/modules/corr.py:192-213) in ONE launch:
 *   for l in 0..L-1   out[:, :, l*rd*rd:(l+1)*rd*rd] =
 *       lowMem_defSample(fmap1[ii].float(), fmap2[l][jj].float(), coords / 2^(lbase+l), offsets[l], radius)
 * written straight into the concatenated tensor out (B,S,L*rd*rd,H1,W1) (what corr.py:211-213 builds for S == 1).
 *   fmap1 (F,H1,W1,C), fmap2[l] (F,H2[l],W2[l],C): the FRAME buffers of the pyramid; ii, jj: device arrays of B int64
 *   frame indices (corr.py:193-194 `self.pyramid[i][:, jj]`) read in place — no gathered per-edge copies.  ii == jj ==
 *   NULL: fmap1 / fmap2[l] are already per-edge, (B,...).
 *   coords (B,S,H1,W1,2) in level-0 units; lbase = pyramid level of fmap2[0] (the level-1 probe of corr.py:201-202 is
 *   this entry with L = 1, lbase = 1, offsets = {NULL}, radius = 1).
 *   offsets[l] (NO,H1,W1,rd,rd,2) IN/OUT with the reference's offset[b*s] indexing, or NULL = zero offsets.
 *   `fmap2`, `offsets`, `H2`, `W2` are HOST arrays of length L <= 4.
 * _h16: half feature maps (C in {32,64,128,256}); _f32: float feature maps, exact fp32 on v_mfma_f32_16x16x4_f32
 * (C in {16,32,64,128}); radius in 1..3; otherwise LGU_E_UNSUPPORTED (compose the per-level entries instead). */
int lgu_lowmem_pyramid_fwd_h16(const void* fmap1_half, const void* const* fmap2_half, const float* coords,
                               float* const* offsets, float* out,
                               int L, int lbase, int B, int S, int H1, int W1, const int* H2, const int* W2, int C, int NO,
                               int radius, const long long* ii, const long long* jj, void* stream);
int lgu_lowmem_pyramid_fwd_f32(const float* fmap1, const float* const* fmap2, const float* coords,
                               float* const* offsets, float* out,
                               int L, int lbase, int B, int S, int H1, int W1, const int* H2, const int* W2, int C, int NO,
                               int radius, const long long* ii, const long long* jj, void* stream);

/* The same with every fmap2[l] in this library's CHUNK-PLANAR form (F, C/k, H2l, W2l, k), k = 8 halves / 4 floats (16
 * bytes of channels); fmap1 stays channel-last.  Same results bit for bit (same products, same summation order); the
 * sweep's operand loads then read 256 contiguous bytes per 16 x-adjacent positions instead of 16 separate lines, which
 * is what bounds it.  AltCorrBlock keeps its feature pyramid in this form. */
int lgu_lowmem_pyramid_chunked_fwd_h16(const void* fmap1_half, const void* const* fmap2_half, const float* coords,
                                       float* const* offsets, float* out,
                                       int L, int lbase, int B, int S, int H1, int W1, const int* H2, const int* W2, int C, int NO,
                                       int radius, const long long* ii, const long long* jj, void* stream);
int lgu_lowmem_pyramid_chunked_fwd_f32(const float* fmap1, const float* const* fmap2, const float* coords,
                                       float* const* offsets, float* out,
                                       int L, int lbase, int B, int S, int H1, int W1, const int* H2, const int* W2, int C, int NO,
                                       int radius, const long long* ii, const long long* jj, void* stream);

/* SEVERAL calls of AltCorrBlock.corr_fn in ONE launch (the chunk loop of update_lowmem, reference
 * droid_slam/factor_graph.py:272-279, runs one corr_fn call per chunk of source frames only to bound memory).  The B
 * edges are the calls' edges back to back, one sample per pixel (S = 1); `offsets[l]` holds NO rows (NO = number of
 * calls), row k = the offsets of call k's FIRST edge — the only row the reference's sampler reads in a call:
 * offset[b*n] with n = 0 (offersample_LGS/lowMem_defSample.cu:80-83) — and off_row (device, B ints, values in [0, NO))
 * names each edge's call.  Edge b's results are, bit for bit, those of lgu_lowmem_pyramid_(chunked_)fwd_h16 over its
 * call alone.  coords (B,1,H1,W1,2), out (B,1,L*rd*rd,H1,W1).  chunked != 0: fmap2 levels in the chunk-planar form.
 * Half maps with C in {32,64,128}, radius 1..3 (the cooperative kernel); LGU_E_UNSUPPORTED otherwise — the caller
 * then issues the calls one by one.  Out-of-range off_row values are clamped to [0, NO) on the device. */
int lgu_lowmem_pyramid_calls_fwd_h16(const void* fmap1_half, const void* const* fmap2_half, const float* coords,
                                     float* const* offsets, float* out,
                                     int L, int lbase, int B, int H1, int W1, const int* H2, const int* W2, int C, int NO,
                                     const int* off_row, int radius, const long long* ii, const long long* jj, int chunked,
                                     void* stream);

/* ---- dense bundle adjustment: device kernels (first version) ---------------------------
 * The data-parallel kernels of droid_backends.ba (reference src/droid.cpp:88-107 -> src/droid_kernels.cu:1314-1434).
 * The reference's host driver copies every block to the CPU and solves with Eigen; here lgu-slam_amd/ba.py assembles
 * and solves the reduced camera system on the device and calls these for its operands.  All index arrays are int64
 * device arrays (the dtype the reference's tensors have).  PARITY UNPINNED: see oracle/ba_oracle.py.
 *
 * lgu_ba_build_f32        projective_transform_kernel (:176-425): per edge e = (ii[e] -> jj[e]): reprojection residual of
 *   targets (E,2,ht,wd) with weights (E,2,ht,wd), pose / depth Jacobians; Hs (4,E,6,6) = Hii,Hij,Hji,Hjj, vs (2,E,6),
 *   Eii, Eij (E,6,ht*wd), Cii, wi (E,ht*wd), all fully written.  poses (N,7) = t, q(xyzw); disps (N,ht,wd); intrinsics (4).
 *   An edge's pixels are split over lgu_ba_build_slices(E) workgroups (so that tens of edges still fill 256 CUs);
 *   `scratch` (device, E * slices * 90 floats) holds their partial sums, which a second kernel adds in slice order.
 * lgu_ba_accum_f32        accum_kernel (:854-874): out[j] = sum of inp rows idxs[ptrs[j] .. ptrs[j+1]), rows of D floats.
 * lgu_ba_depth_system_f32 the depth block of the normal equations in one pass (ba_cuda :1394-1398): for depth frame kx[j],
 *                         C = sum_{edges of the frame} Cii + m*0.05 + (1-m)*eta, w = sum wi - m*0.05*(disps - disps_sens),
 *                         m = (disps_sens > 0); writes Q = 1/C and w, both (K,ht*wd).  Segment tables as lgu_ba_accum_f32;
 *                         eta (eta_rows, ht*wd) with eta_rows in {1, K}.
 * lgu_ba_depth_update_f32 dz = Q (w - sum_{E entries of the frame} dw) (:1415) and disps[kx[j]] += dz (:933-946) in one pass.
 * lgu_ba_scatter_sum_f64  assembly of the reduced camera system (SparseBlock::update_lhs / update_rhs :1137-1179, on the CPU
 *                         in the reference): out[dst[j]] += sign * sum of inp rows idxs[ptrs[j] .. ptrs[j+1]) in double, rows of
 *                         D floats; one thread per output component, fixed summation order (bit-reproducible).
 * lgu_ba_eet_f32          EEt6x6_kernel (:1001-1056): S[b] = (E[idx[b][0]] * Q[idx[b][2]]) E[idx[b][1]]^T, idx (nblocks,3).
 * lgu_ba_ev_f32           Ev6x1_kernel (:1059-1093): v[n] = E[n] (Q[kk[n]] * w[kk[n]]), fully written.
 * lgu_ba_evt_f32          EvT6x1_kernel (:1095-1115): dw[n] = E[n]^T x[idx[n]], zero rows where idx[n] <= 0 or >= P
 *                         (the reference's condition, kept).
 * lgu_ba_solve_f64        SparseBlock::solve (:1206-1231), which the reference runs on the CPU with Eigen: damping
 *                         diag += ep + lm*diag, blocked Cholesky and both triangular solves in one workgroup with the
 *                         matrix in LDS.  A (6P x 6P, row-major double, symmetric), b (6P) double; x (P,6) float; x = 0
 *                         if the damped matrix is not positive definite.  6P <= 192 (the matrix lives in LDS as a packed lower triangle), otherwise LGU_E_UNSUPPORTED.
 * lgu_ba_solve_blocked_f64  the same solve for any window size (csrc/ba_chol.hip): blocked Cholesky over 32-column panels
 *                         (every workgroup factorises the diagonal block in LDS, one thread per row below it; 64 x 64 tiles
 *                         for the trailing update), the right-hand side carried as row 6P of the matrix, back substitution
 *                         in one workgroup.  A is OVERWRITTEN with the factor; work >= lgu_ba_solve_blocked_work_doubles(P)
 *                         doubles; x = 0 if not positive definite.
 * lgu_ba_pose_retr_f32    pose_retr_kernel (:898-931): poses[k] <- exp(dx[k-t0]) * poses[k], k in [t0, t1).
 * lgu_ba_disp_retr_f32    disp_retr_kernel (:933-946): disps[inds[b]] += dz[b]. */
int lgu_ba_build_f32(const float* targets, const float* weights, const float* poses, const float* disps,
                     const float* intrinsics, const long long* ii, const long long* jj,
                     float* Hs, float* vs, float* Eii, float* Eij, float* Cii, float* wi, float* scratch,
                     int E, int ht, int wd, void* stream);
int lgu_ba_build_slices(int E);
int lgu_ba_accum_f32(const float* inp, const long long* ptrs, const long long* idxs, float* out, int nout, int D, void* stream);
int lgu_ba_depth_system_f32(const float* Cii, const float* wi, const long long* ptrs, const long long* idxs, const long long* kx,
                            const float* disps, const float* disps_sens, const float* eta, int eta_rows, float* Q, float* w,
                            int K, int HW, void* stream);
int lgu_ba_depth_update_f32(const float* Q, const float* w, const float* dw, const long long* ptrs, const long long* idxs,
                            const long long* kx, float* dz, float* disps, int K, int HW, void* stream);
int lgu_ba_scatter_sum_f64(const float* inp, const long long* ptrs, const long long* idxs, const long long* dst, double* out,
                           int m, int D, double sign, void* stream);
int lgu_ba_eet_f32(const float* E, const float* Q, const long long* idx, float* S, int nblocks, int D, void* stream);
int lgu_ba_ev_f32(const float* E, const float* Q, const float* w, const long long* kk, float* v, int n, int D, void* stream);
int lgu_ba_evt_f32(const float* E, const float* x, const long long* idx, float* dw, int n, int D, int P, void* stream);
int lgu_ba_solve_f64(const double* A, const double* b, float* x, int P, double lm, double ep, void* stream);
long long lgu_ba_solve_blocked_work_doubles(int P);
int lgu_ba_solve_blocked_f64(double* A, const double* b, float* x, double* work, int P, double lm, double ep, void* stream);
/* lgu_ba_assemble_f64: the four scatter sums of one iteration, their zero-fills and the blocked -> dense permutation in one
 * launch: block d = bi * P + bj of the system = sum of Hs rows [hptr[d], hptr[d+1]) of hidx - sum of S rows (sptr, sidx;
 * S may be NULL: motion only), written to Ad (6P x 6P row-major, double); b (6P) likewise from vs / sv.  CSR tables cover
 * all P*P (resp. P) destinations.  An entry e < 0 of sidx stands for row -e - 1 of S TRANSPOSED (S_ca = S_ac^T: the caller
 * computes the Schur products for a <= c only); direct entries first in a segment.  Per-entry arithmetic and summation
 * order are those of lgu_ba_scatter_sum_f64 (direct rows, then transposed rows). */
int lgu_ba_assemble_f64(const float* Hs, const long long* hptr, const long long* hidx, const float* S, const long long* sptr,
                        const long long* sidx, const float* vs, const long long* vptr, const long long* vidx, const float* sv,
                        const long long* svptr, const long long* svidx, double* Ad, double* b, int P, void* stream);
int lgu_ba_pose_retr_f32(float* poses, const float* dx, int t0, int t1, void* stream);
int lgu_ba_disp_retr_f32(float* disps, const float* dz, const long long* inds, int n, int HW, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LGU_CORR_H */
