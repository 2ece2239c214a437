// CU-mask probe: where do the workgroups of a launch on a hipExtStreamCreateWithCUMask stream run?
// Every workgroup records (XCC id, HW_ID) and then idles for `spin` shader cycles so that a launch with many
// workgroups spreads over every CU its queue may use.  C ABI for tools/cumask_probe.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void where_kernel(uint32_t* out, long long spin) {
  if (threadIdx.x == 0) {
    uint32_t xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
    uint32_t hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
  long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}

extern "C" int probe_create_stream(const uint32_t* mask, int words, void** stream) {
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
  *stream = (void*)s;
  return (int)e;
}
extern "C" int probe_get_mask(void* stream, uint32_t* mask, int words) {
  return (int)hipExtStreamGetCUMask((hipStream_t)stream, (uint32_t)words, mask);
}
extern "C" int probe_where(void* stream, uint32_t* out, int blocks, int threads, long long spin) {
  hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, spin);
  return (int)hipGetLastError();
}
