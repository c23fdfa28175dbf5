/*
 * lgu_oracle.c — CPU restatement of LGU-SLAM's deformable correlation-sampling operators.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP kernels in
 * lgu-slam_amd/csrc: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may build, load or call it.  The product path never does.
 *
 * Every function restates one reference kernel thread-for-thread (one loop iteration
 * = one reference thread), in the reference's own fp32 evaluation order, so that a
 * build with `-O2 -ffp-contract=off` is the reference arithmetic without FMA
 * contraction.  Reference citations are relative to /root/reference.
 *
 * Parity pin: see oracle/README.md — the restatement is checked against the golden
 * vectors in tests/golden/ that were produced by running the reference's own kernel
 * sources (built by oracle/build_ref.py) on an MI355X.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define IDX2(a, b, B) ((size_t)(a) * (B) + (b))

static int within_bounds(int h, int w, int H, int W) { return h >= 0 && h < H && w >= 0 && w < W; }

void oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int oracle_get_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---------------------------------------------------------------------------------
 * defCorr_index_forward — offersample_LGS/defCorrSample_kernel.cu:25-91 (host :165-196)
 * volume (E,H1,W1,H2,W2) coords (E,2,H1,W1) offset (E,H1,W1,rd,rd,2) in/out
 * corr (E,rd,rd,H1,W1), zero-initialised here as torch::zeros does (:181-183).
 * --------------------------------------------------------------------------------- */
void oracle_defcorr_fwd(const float* volume, const float* coords, float* offset, float* corr,
                        int E, int H1, int W1, int H2, int W2, int r) {
  const int rd = 2 * r + 1;
  const size_t HW1 = (size_t)H1 * W1, HW2 = (size_t)H2 * W2;
  memset(corr, 0, sizeof(float) * (size_t)E * rd * rd * HW1);
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < E; n++)
    for (int y = 0; y < H1; y++)
      for (int x = 0; x < W1; x++) {
        const float x0 = coords[((size_t)n * 2 + 0) * HW1 + (size_t)y * W1 + x]; /* :47 */
        const float y0 = coords[((size_t)n * 2 + 1) * HW1 + (size_t)y * W1 + x]; /* :48 */
        float* off = offset + (((size_t)n * H1 + y) * W1 + x) * rd * rd * 2;
        const float* V = volume + (((size_t)n * H1 + y) * W1 + x) * HW2;
        off[((rd / 2) * rd + rd / 2) * 2 + 0] = 0.0f; /* :51 */
        off[((rd / 2) * rd + rd / 2) * 2 + 1] = 0.0f; /* :52 */
        for (int i = 0; i < rd; i++)
          for (int j = 0; j < rd; j++) {
            const float ofsX = off[(i * rd + j) * 2 + 0] + x0; /* :56 */
            const float ofsY = off[(i * rd + j) * 2 + 1] + y0; /* :57 */
            const int fx = (int)floorf(ofsX);                  /* :58 */
            const int fy = (int)floorf(ofsY);
            const float dx = ofsX - (float)fx; /* :60 */
            const float dy = ofsY - (float)fy;
            const int x1 = fx - r + i, x2 = x1 + 1; /* :63-66 */
            const int y1 = fy - r + j, y2 = y1 + 1;
            if (within_bounds(y1, x1, H2, W2)) { /* :67 whole-tap rule */
              float Q11, Q21 = 0.0f, Q12 = 0.0f, Q22 = 0.0f;
              Q11 = V[(size_t)y1 * W2 + x1];
              if (x2 >= 0 && x2 < W2) Q21 = V[(size_t)y1 * W2 + x2];
              if (y2 >= 0 && y2 < H2) Q12 = V[(size_t)y2 * W2 + x1];
              if (y2 >= 0 && y2 < H2 && x2 >= 0 && x2 < W2) Q22 = V[(size_t)y2 * W2 + x2];
              const float w11 = (1.0f - dy) * (1.0f - dx);
              const float w21 = (1.0f - dy) * dx;
              const float w12 = dy * (1.0f - dx);
              const float w22 = dy * dx;
              corr[(((size_t)n * rd + i) * rd + j) * HW1 + (size_t)y * W1 + x] =
                  Q11 * w11 + Q21 * w21 + Q12 * w12 + Q22 * w22; /* :83-86 */
            }
          }
      }
}

/* ---------------------------------------------------------------------------------
 * defCorr_index_backward — defCorrSample_kernel.cu:93-162 (host :198-231)
 * volume_grad, offset_grad zero-initialised here (zeros_like, :210-211).
 * --------------------------------------------------------------------------------- */
void oracle_defcorr_bwd(const float* volume, const float* coords, float* offset,
                        const float* corr_grad, float* volume_grad, float* offset_grad,
                        int E, int H1, int W1, int H2, int W2, int r) {
  const int rd = 2 * r + 1;
  const size_t HW1 = (size_t)H1 * W1, HW2 = (size_t)H2 * W2;
  memset(volume_grad, 0, sizeof(float) * (size_t)E * HW1 * HW2);
  memset(offset_grad, 0, sizeof(float) * (size_t)E * HW1 * rd * rd * 2);
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < E; n++)
    for (int y = 0; y < H1; y++)
      for (int x = 0; x < W1; x++) {
        const float x0 = coords[((size_t)n * 2 + 0) * HW1 + (size_t)y * W1 + x];
        const float y0 = coords[((size_t)n * 2 + 1) * HW1 + (size_t)y * W1 + x];
        const size_t pix = ((size_t)n * H1 + y) * W1 + x;
        float* off = offset + pix * rd * rd * 2;
        float* og = offset_grad + pix * rd * rd * 2;
        const float* V = volume + pix * HW2;
        float* VG = volume_grad + pix * HW2;
        off[((rd / 2) * rd + rd / 2) * 2 + 0] = 0.0f; /* :122 */
        off[((rd / 2) * rd + rd / 2) * 2 + 1] = 0.0f; /* :123 */
        for (int i = 0; i < rd; i++)
          for (int j = 0; j < rd; j++) {
            const float ofsX = off[(i * rd + j) * 2 + 0] + x0;
            const float ofsY = off[(i * rd + j) * 2 + 1] + y0;
            const int fx = (int)floorf(ofsX);
            const int fy = (int)floorf(ofsY);
            const float dx = ofsX - (float)fx;
            const float dy = ofsY - (float)fy;
            const int x1 = fx - r + i, x2 = x1 + 1;
            const int y1 = fy - r + j, y2 = y1 + 1;
            if (within_bounds(y1, x1, H2, W2)) {
              const float g = corr_grad[(((size_t)n * rd + i) * rd + j) * HW1 + (size_t)y * W1 + x];
              float Q11, Q21 = 0.0f, Q12 = 0.0f, Q22 = 0.0f;
              Q11 = V[(size_t)y1 * W2 + x1];
              VG[(size_t)y1 * W2 + x1] += ((1.0f - dy) * (1.0f - dx)) * g; /* :145 */
              if (x2 >= 0 && x2 < W2) {
                Q21 = V[(size_t)y1 * W2 + x2];
                VG[(size_t)y1 * W2 + x2] += ((1.0f - dy) * dx) * g; /* :147-148 */
              }
              if (y2 >= 0 && y2 < H2) {
                Q12 = V[(size_t)y2 * W2 + x1];
                VG[(size_t)y2 * W2 + x1] += (dy * (1.0f - dx)) * g; /* :150-151 */
              }
              if (y2 >= 0 && y2 < H2 && x2 >= 0 && x2 < W2) {
                Q22 = V[(size_t)y2 * W2 + x2];
                VG[(size_t)y2 * W2 + x2] += (dy * dx) * g; /* :153-154 */
              }
              /* :156-157 — [1] is d/dy, [0] is d/dx */
              og[(i * rd + j) * 2 + 1] =
                  (-Q11 * (1.0f - dx) - Q21 * dx + Q12 * (1.0f - dx) + Q22 * dx) * g;
              og[(i * rd + j) * 2 + 0] =
                  (-Q11 * (1.0f - dy) + Q21 * (1.0f - dy) - Q12 * dy + Q22 * dy) * g;
            }
          }
      }
}

/* ---------------------------------------------------------------------------------
 * corr_index_forward (LGU variant) — offersample_LGS/corrSample_kernel.cu:24-82 (host :139-168)
 * --------------------------------------------------------------------------------- */
void oracle_corridx_fwd(const float* volume, const float* coords, float* corr,
                        int E, int H1, int W1, int H2, int W2, int r) {
  const int rd = 2 * r + 1;
  const size_t HW1 = (size_t)H1 * W1, HW2 = (size_t)H2 * W2;
  memset(corr, 0, sizeof(float) * (size_t)E * rd * rd * HW1);
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < E; n++)
    for (int y = 0; y < H1; y++)
      for (int x = 0; x < W1; x++) {
        const float x0 = coords[((size_t)n * 2 + 0) * HW1 + (size_t)y * W1 + x];
        const float y0 = coords[((size_t)n * 2 + 1) * HW1 + (size_t)y * W1 + x];
        const float* V = volume + (((size_t)n * H1 + y) * W1 + x) * HW2;
        for (int i = 0; i < rd; i++)
          for (int j = 0; j < rd; j++) {
            const float dx = x0 - floorf(x0); /* :52-53 */
            const float dy = y0 - floorf(y0);
            const int x1 = (int)floorf(x0) - r + i, x2 = x1 + 1; /* :55-59 */
            const int y1 = (int)floorf(y0) - r + j, y2 = y1 + 1;
            if (within_bounds(y1, x1, H2, W2)) { /* :60 */
              float Q11, Q21 = 0.0f, Q12 = 0.0f, Q22 = 0.0f;
              Q11 = V[(size_t)y1 * W2 + x1];
              if (x2 >= 0 && x2 < W2) Q21 = V[(size_t)y1 * W2 + x2];
              if (y2 >= 0 && y2 < H2) Q12 = V[(size_t)y2 * W2 + x1];
              if (y2 >= 0 && y2 < H2 && x2 >= 0 && x2 < W2) Q22 = V[(size_t)y2 * W2 + x2];
              corr[(((size_t)n * rd + i) * rd + j) * HW1 + (size_t)y * W1 + x] =
                  Q11 * ((1.0f - dy) * (1.0f - dx)) + Q21 * ((1.0f - dy) * dx) +
                  Q12 * (dy * (1.0f - dx)) + Q22 * (dy * dx); /* :74-77 */
            }
          }
      }
}

/* corr_index_backward (LGU variant) — corrSample_kernel.cu:84-136 (host :170-199) */
void oracle_corridx_bwd(const float* coords, const float* corr_grad, float* volume_grad,
                        int E, int H1, int W1, int H2, int W2, int r) {
  const int rd = 2 * r + 1;
  const size_t HW1 = (size_t)H1 * W1, HW2 = (size_t)H2 * W2;
  memset(volume_grad, 0, sizeof(float) * (size_t)E * HW1 * HW2);
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < E; n++)
    for (int y = 0; y < H1; y++)
      for (int x = 0; x < W1; x++) {
        const float x0 = coords[((size_t)n * 2 + 0) * HW1 + (size_t)y * W1 + x];
        const float y0 = coords[((size_t)n * 2 + 1) * HW1 + (size_t)y * W1 + x];
        float* VG = volume_grad + (((size_t)n * H1 + y) * W1 + x) * HW2;
        for (int i = 0; i < rd; i++)
          for (int j = 0; j < rd; j++) {
            const float dx = x0 - floorf(x0);
            const float dy = y0 - floorf(y0);
            const int x1 = (int)floorf(x0) - r + i, x2 = x1 + 1;
            const int y1 = (int)floorf(y0) - r + j, y2 = y1 + 1;
            if (within_bounds(y1, x1, H2, W2)) {
              const float g = corr_grad[(((size_t)n * rd + i) * rd + j) * HW1 + (size_t)y * W1 + x];
              VG[(size_t)y1 * W2 + x1] += ((1.0f - dy) * (1.0f - dx)) * g; /* :121 */
              if (x2 >= 0 && x2 < W2) VG[(size_t)y1 * W2 + x2] += ((1.0f - dy) * dx) * g;
              if (y2 >= 0 && y2 < H2) VG[(size_t)y2 * W2 + x1] += (dy * (1.0f - dx)) * g;
              if (y2 >= 0 && y2 < H2 && x2 >= 0 && x2 < W2) VG[(size_t)y2 * W2 + x2] += (dy * dx) * g;
            }
          }
      }
}

/* ---------------------------------------------------------------------------------
 * gaussianMask — offersample_LGS/gaussianAttn.cu:19-68 (host :134-163)
 * means/covs (E,H1,W1,2) volume/volume1 (E,H1,W1,H2,W2); volume1 = zeros_like (:150).
 * --------------------------------------------------------------------------------- */
void oracle_gaussmask_fwd(const float* means, const float* covs, const float* volume, float* volume1,
                          int E, int H1, int W1, int H2, int W2, int r) {
  const int rd = 2 * r + 1;
  const size_t HW2 = (size_t)H2 * W2;
  memset(volume1, 0, sizeof(float) * (size_t)E * H1 * W1 * HW2);
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < E; n++)
    for (int y = 0; y < H1; y++)
      for (int x = 0; x < W1; x++) {
        const size_t pix = ((size_t)n * H1 + y) * W1 + x;
        const float mean_x = means[pix * 2 + 0], mean_y = means[pix * 2 + 1]; /* :41-42 */
        const float cov_1 = covs[pix * 2 + 0], cov_2 = covs[pix * 2 + 1];     /* :43-44 */
        const float* V = volume + pix * HW2;
        float* V1 = volume1 + pix * HW2;
        for (int i = 0; i < rd; i++)
          for (int j = 0; j < rd; j++) {
            const int cx = (int)floorf(mean_x), cy = (int)floorf(mean_y); /* :51-52 */
            const int x1 = cx - r + i, y1 = cy - r + j;
            if (within_bounds(y1, x1, H2, W2)) {
              const float temp1 = ((float)x1 - mean_x) / cov_1; /* :59 */
              const float temp2 = ((float)y1 - mean_y) / cov_2; /* :60 */
              /* :61 — the -0.5 literal is a double: float sum promoted, scaled, narrowed */
              const float f1 =
                  (float)(-0.5 * (double)(temp1 * ((float)x1 - mean_x) + temp2 * ((float)y1 - mean_y)));
              const float e = expf(f1);                                         /* :62 */
              V1[(size_t)y1 * W2 + x1] = V[(size_t)y1 * W2 + x1] * 3.0f * e;    /* :65 */
            }
          }
      }
}

/* gaussianMask_backward — gaussianAttn.cu:72-131 (host :165-200) */
void oracle_gaussmask_bwd(const float* means, const float* covs, const float* volume,
                          const float* volume1_grad, float* means_grad, float* covs_grad,
                          int E, int H1, int W1, int H2, int W2, int r) {
  const int rd = 2 * r + 1;
  const size_t HW2 = (size_t)H2 * W2;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < E; n++)
    for (int y = 0; y < H1; y++)
      for (int x = 0; x < W1; x++) {
        const size_t pix = ((size_t)n * H1 + y) * W1 + x;
        const float mean_x = means[pix * 2 + 0], mean_y = means[pix * 2 + 1];
        const float cov_1 = covs[pix * 2 + 0], cov_2 = covs[pix * 2 + 1];
        const float* V = volume + pix * HW2;
        const float* G = volume1_grad + pix * HW2;
        float mg0 = 0.0f, mg1 = 0.0f, cg0 = 0.0f, cg1 = 0.0f; /* zeros_like, :176-177 */
        for (int i = 0; i < rd; i++)
          for (int j = 0; j < rd; j++) {
            const int cx = (int)floorf(mean_x), cy = (int)floorf(mean_y);
            const int x1 = cx - r + i, y1 = cy - r + j;
            if (within_bounds(y1, x1, H2, W2)) {
              const float ddx = (float)x1 - mean_x, ddy = (float)y1 - mean_y;
              const float temp1 = ddx / cov_1, temp2 = ddy / cov_2;
              const float f1 = (float)(-0.5 * (double)(temp1 * ddx + temp2 * ddy));
              const float e = expf(f1);
              const float v = V[(size_t)y1 * W2 + x1], g = G[(size_t)y1 * W2 + x1];
              mg0 += 3.0f * v * (e * ddx / cov_1) * g; /* :116 */
              mg1 += 3.0f * v * (e * ddy / cov_2) * g; /* :117 */
              /* :119,121 — the 0.5 literal makes the whole product a double expression */
              const float dE1 = (float)((double)e * 0.5 * (double)ddx * (double)ddx / (double)(cov_1 * cov_1));
              const float dE2 = (float)((double)e * 0.5 * (double)ddy * (double)ddy / (double)(cov_2 * cov_2));
              cg0 += (3.0f * v * dE1) * g; /* :124 */
              cg1 += (3.0f * v * dE2) * g; /* :125 */
            }
          }
        means_grad[pix * 2 + 0] = mg0;
        means_grad[pix * 2 + 1] = mg1;
        covs_grad[pix * 2 + 0] = cg0;
        covs_grad[pix * 2 + 1] = cg1;
      }
}

/* ---------------------------------------------------------------------------------
 * lowMem_defSample — offersample_LGS/lowMem_defSample.cu:27-134 (host :137-168)
 * fmap1 (B,H1,W1,C) fmap2 (B,H2,W2,C) coords (B,S,H1,W1,2) offset (NO,H1,W1,rd,rd,2)
 * corr (B,S,rd,rd,H1,W1) zero-initialised (:150).  offset is indexed with b*n (:80-83).
 * Pixels are independent in the reference (each thread touches only its own LDS
 * column), so one loop iteration per (b,h1,w1) is the same computation.
 * --------------------------------------------------------------------------------- */
void oracle_lowmem_defsample_fwd(const float* fmap1, const float* fmap2, const float* coords,
                                 float* offset, float* corr,
                                 int B, int S, int H1, int W1, int H2, int W2, int C, int r) {
  const int rd = 2 * r + 1;
  const size_t HW1 = (size_t)H1 * W1;
  memset(corr, 0, sizeof(float) * (size_t)B * S * rd * rd * HW1);
  /* centre zeroing touches offset[b*n] for every (b,n): do it first, it is idempotent
   * and every reader in the reference zeroes its own copy before reading (:80-81). */
  for (int b = 0; b < B; b++)
    for (int n = 0; n < S; n++) {
      float* ob = offset + (size_t)(b * n) * HW1 * rd * rd * 2;
      for (size_t p = 0; p < HW1; p++) {
        ob[p * rd * rd * 2 + ((rd / 2) * rd + rd / 2) * 2 + 0] = 0.0f;
        ob[p * rd * rd * 2 + ((rd / 2) * rd + rd / 2) * 2 + 1] = 0.0f;
      }
    }
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; b++)
    for (int h1 = 0; h1 < H1; h1++)
      for (int w1 = 0; w1 < W1; w1++) {
        const float* f1 = fmap1 + (((size_t)b * H1 + h1) * W1 + w1) * C;
        const float* F2 = fmap2 + (size_t)b * H2 * W2 * C;
        for (int c = 0; c < C; c += 32) /* :56 CHANNEL_STRIDE chunks, outermost */
          for (int n = 0; n < S; n++) {
            const float* co = coords + ((((size_t)b * S + n) * H1 + h1) * W1 + w1) * 2;
            const float* off = offset + ((size_t)(b * n) * HW1 + (size_t)h1 * W1 + w1) * rd * rd * 2;
            for (int iy = 0; iy < rd; iy++)
              for (int ix = 0; ix < rd; ix++) {
                const float xs = co[0] + off[(ix * rd + iy) * 2 + 0]; /* :82 */
                const float ys = co[1] + off[(ix * rd + iy) * 2 + 1]; /* :83 */
                const float dx = xs - floorf(xs);                     /* :87 */
                const float dy = ys - floorf(ys);
                const int h2 = (int)floorf(ys) - r + iy, h2h = h2 + 1; /* :91-94 */
                const int w2 = (int)floorf(xs) - r + ix, w2h = w2 + 1;
                const float w11 = (1.0f - dy) * (1.0f - dx), w21 = (1.0f - dy) * dx;
                const float w12 = dy * (1.0f - dx), w22 = dy * dx;
                float Q = 0.0f;
                for (int k = 0; k < 32; k++) {
                  float Q11 = 0.0f, Q21 = 0.0f, Q12 = 0.0f, Q22 = 0.0f; /* per-corner zero padding :102-112 */
                  if (within_bounds(h2, w2, H2, W2)) Q11 = F2[((size_t)h2 * W2 + w2) * C + c + k];
                  if (within_bounds(h2, w2h, H2, W2)) Q21 = F2[((size_t)h2 * W2 + w2h) * C + c + k];
                  if (within_bounds(h2h, w2, H2, W2)) Q12 = F2[((size_t)h2h * W2 + w2) * C + c + k];
                  if (within_bounds(h2h, w2h, H2, W2)) Q22 = F2[((size_t)h2h * W2 + w2h) * C + c + k];
                  const float f2 = Q11 * w11 + Q21 * w21 + Q12 * w12 + Q22 * w22; /* :114-117 */
                  Q += f1[c + k] * f2;                                             /* :122-125 */
                }
                corr[((((size_t)b * S + n) * rd + ix) * rd + iy) * HW1 + (size_t)h1 * W1 + w1] += Q; /* :128 */
              }
          }
      }
}

/* ---------------------------------------------------------------------------------
 * altcorr_forward — src/altcorr_kernel.cu:27-149 (host :290-319)
 * corr (B,S,rd*rd,H1,W1) zero-initialised (:302).  Output channel = ix*rd + iy.
 * --------------------------------------------------------------------------------- */
void oracle_altcorr_fwd(const float* fmap1, const float* fmap2, const float* coords, float* corr,
                        int B, int S, int H1, int W1, int H2, int W2, int C, int r) {
  const int rd = 2 * r + 1;
  const size_t HW1 = (size_t)H1 * W1;
  memset(corr, 0, sizeof(float) * (size_t)B * S * rd * rd * HW1);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; b++)
    for (int h1 = 0; h1 < H1; h1++)
      for (int w1 = 0; w1 < W1; w1++) {
        const float* f1 = fmap1 + (((size_t)b * H1 + h1) * W1 + w1) * C;
        const float* F2 = fmap2 + (size_t)b * H2 * W2 * C;
        for (int c = 0; c < C; c += 32)
          for (int n = 0; n < S; n++) {
            const float* co = coords + ((((size_t)b * S + n) * H1 + h1) * W1 + w1) * 2;
            const float xs = co[0], ys = co[1]; /* :73-74 */
            const float dx = xs - floorf(xs), dy = ys - floorf(ys);
            float* out = corr + (((size_t)b * S + n) * rd * rd) * HW1 + (size_t)h1 * W1 + w1;
            for (int iy = 0; iy < rd + 1; iy++)
              for (int ix = 0; ix < rd + 1; ix++) {
                const int h2 = (int)floorf(ys) - r + iy;
                const int w2 = (int)floorf(xs) - r + ix;
                float s = 0.0f;
                if (within_bounds(h2, w2, H2, W2)) {
                  const float* f2 = F2 + ((size_t)h2 * W2 + w2) * C;
                  for (int k = 0; k < 32; k++) s += f1[c + k] * f2[c + k]; /* :98-100 */
                }
                const float nw = s * (dy * dx), ne = s * (dy * (1 - dx));    /* :112-115 */
                const float sw = s * ((1 - dy) * dx), se = s * ((1 - dy) * (1 - dx));
                if (iy > 0 && ix > 0) out[(size_t)((iy - 1) + rd * (ix - 1)) * HW1] += nw; /* :132-142 */
                if (iy > 0 && ix < rd) out[(size_t)((iy - 1) + rd * ix) * HW1] += ne;
                if (iy < rd && ix > 0) out[(size_t)(iy + rd * (ix - 1)) * HW1] += sw;
                if (iy < rd && ix < rd) out[(size_t)(iy + rd * ix) * HW1] += se;
              }
          }
      }
}

/* altcorr_backward — src/altcorr_kernel.cu:152-286 (host :321-356); float only.
 * fmap2_grad is accumulated with atomicAdd in the reference (:267): the sum order is
 * unspecified there; this restatement adds in (b,h1,w1,c,n,iy,ix) order, serially. */
void oracle_altcorr_bwd(const float* fmap1, const float* fmap2, const float* coords,
                        const float* corr_grad, float* fmap1_grad, float* fmap2_grad,
                        int B, int S, int H1, int W1, int H2, int W2, int C, int r) {
  const int rd = 2 * r + 1;
  const size_t HW1 = (size_t)H1 * W1;
  memset(fmap1_grad, 0, sizeof(float) * (size_t)B * HW1 * C);
  memset(fmap2_grad, 0, sizeof(float) * (size_t)B * H2 * W2 * C);
  for (int b = 0; b < B; b++)
    for (int h1 = 0; h1 < H1; h1++)
      for (int w1 = 0; w1 < W1; w1++) {
        const float* f1 = fmap1 + (((size_t)b * H1 + h1) * W1 + w1) * C;
        float* f1g = fmap1_grad + (((size_t)b * H1 + h1) * W1 + w1) * C;
        const float* F2 = fmap2 + (size_t)b * H2 * W2 * C;
        float* F2G = fmap2_grad + (size_t)b * H2 * W2 * C;
        for (int c = 0; c < C; c += 32)
          for (int n = 0; n < S; n++) {
            const float* co = coords + ((((size_t)b * S + n) * H1 + h1) * W1 + w1) * 2;
            const float xs = co[0], ys = co[1];
            const float dx = xs - floorf(xs), dy = ys - floorf(ys);
            const float* gp = corr_grad + (((size_t)b * S + n) * rd * rd) * HW1 + (size_t)h1 * W1 + w1;
            for (int iy = 0; iy < rd + 1; iy++)
              for (int ix = 0; ix < rd + 1; ix++) {
                const int h2 = (int)floorf(ys) - r + iy;
                const int w2 = (int)floorf(xs) - r + ix;
                float g = 0.0f; /* :238-250 */
                if (iy > 0 && ix > 0) g += gp[(size_t)((iy - 1) + rd * (ix - 1)) * HW1] * dy * dx;
                if (iy > 0 && ix < rd) g += gp[(size_t)((iy - 1) + rd * ix) * HW1] * dy * (1 - dx);
                if (iy < rd && ix > 0) g += gp[(size_t)(iy + rd * (ix - 1)) * HW1] * (1 - dy) * dx;
                if (iy < rd && ix < rd) g += gp[(size_t)(iy + rd * ix) * HW1] * (1 - dy) * (1 - dx);
                const int inb = within_bounds(h2, w2, H2, W2);
                for (int k = 0; k < 32; k++) {
                  const float f2v = inb ? F2[((size_t)h2 * W2 + w2) * C + c + k] : 0.0f;
                  f1g[c + k] += g * f2v;                                              /* :253 */
                  if (inb) F2G[((size_t)h2 * W2 + w2) * C + c + k] += 0.0f + g * f1[c + k]; /* :254,267 */
                }
              }
          }
      }
}

/* ---------------------------------------------------------------------------------
 * Fused CorrBlock.__call__ body — droid_slam/modules/corr.py:88-109 restated on top of
 * the two kernels above (probe: corr.py:94-99; pyramid sample: :101-103; cat: :109).
 * volumes/offsets are arrays of L host pointers; offsets[l] may be NULL (= zeros).
 * out (E, L*rd*rd, H1, W1).  scratch_off must hold E*H1*W1*rd*rd*2 floats (used for
 * NULL offsets), scratch_corr E*rd*rd*H1*W1 floats, scratch_probe E*9*H1*W1 floats.
 * --------------------------------------------------------------------------------- */
void oracle_defcorr_pyramid_fwd(const float* const* volumes, const float* coords, float* const* offsets,
                                float* out, int L, int E, int H1, int W1, const int* H2, const int* W2,
                                int r, int probe, float* scratch_coords, float* scratch_off,
                                float* scratch_corr, float* scratch_probe) {
  const int rd = 2 * r + 1;
  const size_t HW1 = (size_t)H1 * W1;
  const size_t ncoords = (size_t)E * 2 * HW1;
  if (probe) {
    /* corr.py:94  CorrSampler.apply(pyr[1], coords/2, 1) */
    for (size_t k = 0; k < ncoords; k++) scratch_coords[k] = coords[k] / 2.0f;
    oracle_corridx_fwd(volumes[1], scratch_coords, scratch_probe, E, H1, W1, H2[1], W2[1], 1);
    /* corr.py:95-99  unbiased var over the 9 taps -> sigmoid -> offset[1] *= mask */
    for (int n = 0; n < E; n++)
      for (size_t p = 0; p < HW1; p++) {
        float v[9], mean = 0.0f;
        for (int t = 0; t < 9; t++) { v[t] = scratch_probe[((size_t)n * 9 + t) * HW1 + p]; mean += v[t]; }
        mean = mean / 9.0f;
        float ss = 0.0f;
        for (int t = 0; t < 9; t++) ss += (v[t] - mean) * (v[t] - mean);
        const float var = ss / 8.0f;
        const float mask = 1.0f / (1.0f + expf(-var));
        float* off = offsets[1] + ((size_t)n * HW1 + p) * rd * rd * 2;
        for (int t = 0; t < rd * rd * 2; t++) off[t] = off[t] * mask;
      }
  }
  for (int l = 0; l < L; l++) {
    const float scale = (float)(1 << l);
    for (size_t k = 0; k < ncoords; k++) scratch_coords[k] = coords[k] / scale; /* corr.py:102 */
    float* off = offsets[l];
    if (!off) { off = scratch_off; memset(off, 0, sizeof(float) * (size_t)E * HW1 * rd * rd * 2); }
    oracle_defcorr_fwd(volumes[l], scratch_coords, off, scratch_corr, E, H1, W1, H2[l], W2[l], r);
    for (int n = 0; n < E; n++)
      memcpy(out + ((size_t)n * L + l) * rd * rd * HW1, scratch_corr + (size_t)n * rd * rd * HW1,
             sizeof(float) * rd * rd * HW1);
  }
}

/* ---------------------------------------------------------------------------------
 * Volume post-processing of CorrBlock.__init__, restated op by op:
 *   corr1 = gaussianMask(mean, cov, corr, r)                 (gaussianMask_cuda.py:84)
 *   corr  = corr1 / (6.28*sqrt(cov0*cov1)) + corr            (gaussianMask_cuda.py:79,85-86)
 *   level l = avg_pool2d(level l-1, 2, stride 2), l >= 1     (corr.py:83-86; ATen sums the
 *             2x2 window row-major in fp32 and divides by 4)
 * levels[l] (E,H1,W1,H2>>l,W2>>l); scratch holds one volume.
 * --------------------------------------------------------------------------------- */
void oracle_volume_pyramid(const float* means, const float* covs, const float* volume, float* const* levels, int L,
                           int E, int H1, int W1, int H2, int W2, int r, float* scratch) {
  const size_t npix = (size_t)E * H1 * W1, HW2 = (size_t)H2 * W2;
  oracle_gaussmask_fwd(means, covs, volume, scratch, E, H1, W1, H2, W2, r);
#pragma omp parallel for schedule(static)
  for (size_t p = 0; p < npix; p++) {
    const float den = 6.28f * sqrtf(covs[p * 2 + 0] * covs[p * 2 + 1]);
    for (size_t k = 0; k < HW2; k++) levels[0][p * HW2 + k] = scratch[p * HW2 + k] / den + volume[p * HW2 + k];
  }
  int Hs = H2, Ws = W2;
  for (int l = 1; l < L; l++) {
    const int Hd = Hs / 2, Wd = Ws / 2;
    const float* src = levels[l - 1];
    float* dst = levels[l];
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < npix; p++)
      for (int y = 0; y < Hd; y++)
        for (int x = 0; x < Wd; x++) {
          const float* s = src + p * (size_t)Hs * Ws + (size_t)(2 * y) * Ws + 2 * x;
          dst[p * (size_t)Hd * Wd + (size_t)y * Wd + x] = (((s[0] + s[1]) + s[Ws]) + s[Ws + 1]) / 4.0f;
        }
    Hs = Hd;
    Ws = Wd;
  }
}
