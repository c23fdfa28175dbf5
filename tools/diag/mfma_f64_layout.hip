// Operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, found by running it (tools/diag, not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
  const int l = threadIdx.x;
  // 1: rows.  A[i][k] = i + 1 for the lanes that (by hypothesis) hold k = 0, B[k][j] = 1 at k = 0
  d4 c = {0, 0, 0, 0};
  double a = (l / 16 == 0) ? (double)(l % 16 + 1) : 0.0, b = (l / 16 == 0) ? 1.0 : 0.0;
  d4 r = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int q = 0; q < 4; q++) out[l * 4 + q] = r[q];
  // 2: columns
  a = (l / 16 == 0) ? 1.0 : 0.0; b = (l / 16 == 0) ? (double)(l % 16 + 1) : 0.0;
  r = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int q = 0; q < 4; q++) out[256 + l * 4 + q] = r[q];
  // 3: k pairing: A one-hot in lanes of group 2, B one-hot in lanes of group g
  for (int g = 0; g < 4; g++) {
    a = (l / 16 == 2) ? 1.0 : 0.0; b = (l / 16 == g) ? 1.0 : 0.0;
    r = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    if (l == 0) out[512 + g] = r[0];
  }
}
int main() {
  double* d; hipMalloc(&d, 1024 * 8);
  probe<<<1, 64>>>(d);
  double h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("row of (lane, reg):\n");
  for (int l = 0; l < 64; l += 16) printf(" lane %2d: %g %g %g %g\n", l, h[l * 4] - 1, h[l * 4 + 1] - 1, h[l * 4 + 2] - 1, h[l * 4 + 3] - 1);
  printf("col of (lane, reg 0): lane 0 %g lane 1 %g lane 15 %g lane 16 %g lane 17 %g\n", h[256] - 1, h[256 + 4] - 1, h[256 + 60] - 1, h[256 + 64] - 1, h[256 + 68] - 1);
  printf("k pairing (A group 2 x B group g): %g %g %g %g\n", h[512], h[513], h[514], h[515]);
  return 0;
}
