// Which contraction index does element j of lane l of the A / B operands of v_mfma_f32_16x16x32_bf16 carry?  One wave,
// A[16][32] and B[32][16] of small integers (exact in bf16); the device result is compared with the host product under
// candidate layouts.  Build: hipcc --offload-arch=gfx950 -O2 tools/probe/mfma16_layout_probe.hip -o /tmp/mfma16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const float* A, const float* B, float* D, int mode) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  bf16x8 a, b;
  for (int j = 0; j < 8; j++) {
    int ka = 8 * g + j, kb = 8 * g + j;
    if (mode == 1) kb = 4 * g + (j & 3) + 16 * (j >> 2);
    if (mode == 2) { ka = 4 * g + (j & 3) + 16 * (j >> 2); kb = ka; }
    a[j] = (__bf16)A[i * 32 + ka];
    b[j] = (__bf16)B[kb * 16 + i];
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[(4 * g + r) * 16 + i] = c[r];
}
int main() {
  float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
  for (int i = 0; i < 512; i++) hA[i] = (float)((i * 7 + 3) % 11 - 5), hB[i] = (float)((i * 5 + 1) % 13 - 6);
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) {
      float s = 0;
      for (int k2 = 0; k2 < 32; k2++) s += hA[i * 32 + k2] * hB[k2 * 16 + j];
      ref[i * 16 + j] = s;
    }
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  for (int mode = 0; mode < 3; mode++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, mode);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; i++) bad += hD[i] != ref[i];
    printf("mode %d (0: k = 8g + j for A and B; 1: B as two 4-blocks; 2: both as two 4-blocks): %d of 256 wrong; D[0][0..3] = %g %g %g %g ref %g %g %g %g\n",
           mode, bad, hD[0], hD[1], hD[2], hD[3], ref[0], ref[1], ref[2], ref[3]);
  }
  return 0;
}
