// Which bf16 MFMA shape sustains more FLOP/s on this MI355X under its power cap?  Two bare loops, one wave per SIMD (256
// threads per workgroup, one workgroup per CU), operands re-read from LDS by ds_read_b128 every step, random data:
//   A: v_mfma_f32_32x32x16_bf16, wave tile 64 x 32 (2 accumulators of 16 regs), per k-step of 16: 2 A + 1 B reads, 2 MFMAs
//   B: v_mfma_f32_16x16x32_bf16, wave tile 64 x 32 (8 accumulators of 4 regs),  per k-step of 32: 4 A + 2 B reads, 8 MFMAs
// Same FLOPs and the same LDS bytes per FLOP.  Prints TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int SHAPE>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ src, float* __restrict__ out, int iters,
                                              unsigned long long* __restrict__ stamps) {
  __shared__ u32x4 lds[4096];  // 64 KB of operands
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[(blockIdx.x * 4096 + i) & 0xFFFFF];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if constexpr (SHAPE == 0) {
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; i++) acc0[i] = acc1[i] = 0.f;
    u32x4 a0[2], a1[2], b[2];
    auto rd = [&](int step, int s) {
      const int base = (step * 192 + wave * 48) & 4095;
      a0[s] = lds[(base + lane) & 4095], a1[s] = lds[(base + 64 + lane) & 4095], b[s] = lds[(base + 128 + lane) & 4095];
    };
    rd(0, 0);
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 16; k++) {  // 16 k-steps of 16; the reads of step k+1 are issued above the MFMAs of step k
        rd(it * 16 + k + 1, (k + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0[k & 1]), __builtin_bit_cast(bf16x8, b[k & 1]), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1[k & 1]), __builtin_bit_cast(bf16x8, b[k & 1]), acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (int i = 0; i < 16; i++) s += acc0[i] + acc1[i];
  } else {
    f32x4 acc[8];
    for (int j = 0; j < 8; j++)
      for (int i = 0; i < 4; i++) acc[j][i] = 0.f;
    u32x4 a[2][4], b[2][2];
    auto rd = [&](int step, int s) {
      const int base = (step * 384 + wave * 96) & 4095;
#pragma unroll
      for (int j = 0; j < 4; j++) a[s][j] = lds[(base + 64 * j + lane) & 4095];
#pragma unroll
      for (int j = 0; j < 2; j++) b[s][j] = lds[(base + 256 + 64 * j + lane) & 4095];
    };
    rd(0, 0);
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {  // 8 k-steps of 32
        rd(it * 8 + k + 1, (k + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
          for (int n = 0; n < 2; n++)
            acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[k & 1][m]), __builtin_bit_cast(bf16x8, b[k & 1][n]), acc[m * 2 + n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (int j = 0; j < 8; j++)
      for (int i = 0; i < 4; i++) s += acc[j][i];
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

int main() {
  const int nb = 256, iters = 2000;
  std::vector<uint32_t> h(4 << 20);
  srand(1);
  for (auto& v : h) {  // random bf16 pairs in [-1, 1)
    auto bf = [](float f) { union { float f; uint32_t u; } c; c.f = f; return (c.u + 0x7FFF + ((c.u >> 16) & 1)) >> 16; };
    v = bf(rand() / (float)RAND_MAX * 2 - 1) | (bf(rand() / (float)RAND_MAX * 2 - 1) << 16);
  }
  u32x4* src; float* out; unsigned long long* st;
  hipMalloc(&src, h.size() * 4); hipMalloc(&out, nb * 256 * 4); hipMalloc(&st, nb * 16);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = (double)nb * 4 * iters * 16 * 2 * (2.0 * 32 * 32 * 16);
  for (int round = 0; round < 3; round++)
    for (int shape = 0; shape < 2; shape++) {
      const int reps = 60;
      for (int w = 0; w < 20; w++) {
        if (shape == 0) hipLaunchKernelGGL(probe<0>, dim3(nb), dim3(256), 0, 0, src, out, iters, st);
        else hipLaunchKernelGGL(probe<1>, dim3(nb), dim3(256), 0, 0, src, out, iters, st);
      }
      hipEventRecord(e0);
      for (int r = 0; r < reps; r++) {
        if (shape == 0) hipLaunchKernelGGL(probe<0>, dim3(nb), dim3(256), 0, 0, src, out, iters, st);
        else hipLaunchKernelGGL(probe<1>, dim3(nb), dim3(256), 0, 0, src, out, iters, st);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> hs(nb * 2);
      hipMemcpy(hs.data(), st, nb * 16, hipMemcpyDeviceToHost);
      std::vector<double> clk;
      for (int b = 0; b < nb; b++) clk.push_back((double)hs[2 * b] / (double)hs[2 * b + 1] * 0.1);
      std::sort(clk.begin(), clk.end());
      printf("round %d %s: %.1f us per launch, %.0f TFLOP/s, in-kernel clock median %.2f GHz, cycles per WG %.0f\n", round,
             shape == 0 ? "32x32x16" : "16x16x32", ms / reps * 1e3, flop / (ms / reps * 1e-3) / 1e12, clk[nb / 2],
             (double)hs[0]);
    }
  return 0;
}
