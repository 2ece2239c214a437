// Pieces shared by the convolution kernels of conv_igemm.hip and conv_wr.hip: the MFMA wrapper per storage type, the
// MFMA-row -> tile-voxel map of the 4x8x8 tile family, the LDS-only workgroup barrier and the persistent kernels' tile
// record.
#pragma once
#include "hdf_common.h"

namespace {

template <typename T>
struct Mma;
template <>
struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mma<f16_t> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mma<float> {
  // lane (r, h) holds channels 4h..4h+3 of an 8-channel group: step s contracts channels {s, 4+s}
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
#pragma unroll
    for (int s = 0; s < 4; s++)
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[s]), __uint_as_float(b[s]), c, 0, 0, 0);
  }
};

// MFMA row (0..31) -> (dz in 0..3, x in 0..7).  ds_read_b128 services lanes {0-3,12-15,20-27} and {4-11,16-19,
// 28-31} (and the same +32) as groups; group 1 gets z in {0,2}, group 2 z in {1,3}: box row = 100*z + 10*y + x
// (BH = BW = 10) is then distinct mod 16 inside each group.
constexpr int WS_STAT_ROWS = 512;  // conv_ws2_kernel: InstanceNorm partial rows per sample (2 passes x 256 workgroup slots)
// (g2, rank): which of the two ds_read_b128 service groups MFMA row r (0..31) belongs to, and its position 0..15 there
__device__ __forceinline__ void ds128_group(int r, int& g2, int& rank) {
  g2 = ((r >= 4 && r < 12) || (r >= 16 && r < 20) || r >= 28) ? 1 : 0;
  rank = g2 ? (r < 12 ? r - 4 : (r < 20 ? r - 8 : r - 16)) : (r < 4 ? r : (r < 16 ? r - 8 : r - 12));
}
__device__ __forceinline__ void ws_row_to_zx(int r, int& dz, int& x) {
  int g2, rank;
  ds128_group(r, g2, rank);
  dz = 2 * (rank >> 3) + g2;
  x = rank & 7;
}

// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain the vector-memory
// counter, so prefetch loads and epilogue stores stay in flight across it
#define WS_BARRIER()                                     \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_s_barrier();                        \
    asm volatile("" ::: "memory");                       \
  } while (0)

struct WsTile {
  int n, z0, y0, x0, tile;
  int k;  // index in the list the tile came from (border pass)
};


}  // namespace
