// Stand-alone bandwidth experiments for the InstanceNorm-backward "apply" access pattern (2 x 16-byte loads + 1 store
// per thread-chunk over three 268 MB bf16 tensors).  hipcc --offload-arch=gfx950 -O3 tools/ew_bench.hip -o ew_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ void unpack(const u32x4& c, float* f) {
  for (int i = 0; i < 4; i++) { f[2*i] = __uint_as_float(c[i] << 16); f[2*i+1] = __uint_as_float(c[i] & 0xffff0000u); }
}
__device__ __forceinline__ u32x4 pack(const float* f) {
  u32x4 c;
  for (int i = 0; i < 4; i++) { f32x2 v = {f[2*i], f[2*i+1]}; c[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }
  return c;
}
// MODE 0: copy-like (a+b raw xor), 1: full math; U chunks per iteration; grid-stride
template <int MODE, int U>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ o,
                                         const float* __restrict__ co, int64_t total) {
  float c[7][8];
  const int col = threadIdx.x & 3;
  if (MODE == 1) for (int j = 0; j < 7; j++) for (int e = 0; e < 8; e++) c[j][e] = co[j * 32 + col * 8 + e];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += stride * U) {
    u32x4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; u++) { int64_t i = min(i0 + u * stride, total - 1); va[u] = a[i]; vb[u] = b[i]; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      u32x4 r;
      if (MODE == 0) { r = va[u] ^ vb[u]; }
      else {
        float g[8], f[8];
        unpack(va[u], g); unpack(vb[u], f);
#pragma unroll
        for (int e = 0; e < 8; e++) {
          float gg = (f[e] * c[0][e] + c[1][e] > 0.f) ? g[e] : 0.f;
          float xh = (f[e] - c[2][e]) * c[3][e];
          g[e] = c[4][e] * (gg - c[5][e] - xh * c[6][e]);
        }
        r = pack(g);
      }
      if (i0 + u * stride < total) o[i0 + u * stride] = r;
    }
  }
}
template <int MODE, int U>
float run(const u32x4* a, const u32x4* b, u32x4* o, const float* co, int64_t total, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<MODE, U>), dim3(blocks), dim3(256), 0, 0, a, b, o, co, total);
  hipEventRecord(e0);
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<MODE, U>), dim3(blocks), dim3(256), 0, 0, a, b, o, co, total);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 20;
}
int main() {
  const int64_t total = (int64_t)2 * 128 * 128 * 128 * 4;  // 16-byte chunks of a [2,128^3,32] bf16 tensor
  u32x4 *a, *b, *o; float* co;
  hipMalloc(&a, total * 16); hipMalloc(&b, total * 16); hipMalloc(&o, total * 16); hipMalloc(&co, 7 * 32 * 4);
  hipMemset(a, 0x11, total * 16); hipMemset(b, 0x22, total * 16); hipMemset(co, 0, 7 * 32 * 4);
  const double gb = 3.0 * total * 16 / 1e9;
  int bl[] = {1024, 2048, 4096, 8192, 16384, 65536};
  for (int blocks : bl) {
    float t0 = run<0, 4>(a, b, o, co, total, blocks), t1 = run<1, 4>(a, b, o, co, total, blocks);
    float t2 = run<1, 2>(a, b, o, co, total, blocks), t3 = run<1, 8>(a, b, o, co, total, blocks), t4 = run<0, 1>(a, b, o, co, total, blocks);
    printf("blocks %6d: copy U4 %.0f us %.2f TB/s | math U4 %.0f us %.2f | math U2 %.0f us %.2f | math U8 %.0f us %.2f | copy U1 %.0f us %.2f\n",
           blocks, t0 * 1e3, gb / t0, t1 * 1e3, gb / t1, t2 * 1e3, gb / t2, t3 * 1e3, gb / t3, t4 * 1e3, gb / t4);
  }
  return 0;
}
