// Common device/host helpers for the gfx950 (CDNA4) kernels of the H-DenseFormer hot path.
// Everything here is MI355X-only: wave64, MFMA, 160 KiB LDS.  No portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define HDF_OK 0
#define HDF_ERR_ARG 1
#define HDF_ERR_HIP 2
#define HDF_ERR_UNSUPPORTED 3
#define HDF_ERR_CHAIN_TIMEOUT 4   // a persistent transformer launch of an earlier call gave up (see include/hdf.h)

enum { HDF_F32 = 0, HDF_BF16 = 1, HDF_F16 = 2 };
static inline int hdf_esz(int dtype) { return dtype == HDF_F32 ? 4 : 2; }  // bytes per stored activation element

void hdf_set_error(const char* fmt, ...);

// Persistent kernels size their grid to the compute units they may use: 256 = the whole chip (default).  A
// process-wide diagnostic knob (hdf_set_cu_budget; tools/cu_budget_sweep.py, tools/cumask_probe.py): the plan itself
// never changes it and creates no CU-masked streams (measured: no gain).
int hdf_cu_budget();

#define HDF_CHECK_ARG(cond, ...)              \
  do {                                        \
    if (!(cond)) {                            \
      hdf_set_error(__VA_ARGS__);             \
      return HDF_ERR_ARG;                     \
    }                                         \
  } while (0)

#define HDF_LAUNCH_CHECK()                                                        \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      hdf_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return HDF_ERR_HIP;                                                         \
    }                                                                             \
  } while (0)

#define HDF_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != HDF_OK) return rc__; \
  } while (0)

// ---------------------------------------------------------------------------------------------
// storage types
struct bf16_t {
  uint16_t v;
};
struct f16_t {  // IEEE binary16 storage (torch.float16 autocast, trainer.py:20-21,369): v_mfma_f32_32x32x16_f16
  uint16_t v;
};

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN preserved
  return __builtin_bit_cast(uint16_t, h);
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// one v_cvt_pk_bf16_f32 for the pair (two scalar converts + shift + or cost three times the issue slots)
__device__ __forceinline__ uint32_t pack_bf2(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
__device__ __forceinline__ float h2f(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }  // RNE
__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}

template <typename T>
struct ST;  // storage traits
template <>
struct ST<float> {
  static constexpr int EPC = 4;  // elements per 16-byte chunk
  static constexpr int DT = HDF_F32;
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
  // four consecutive elements in one store / load (p aligned to the access)
  __device__ static __forceinline__ void st4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
  }
  __device__ static __forceinline__ void ld4(const float* p, float* f) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    f[0] = v[0], f[1] = v[1], f[2] = v[2], f[3] = v[3];
  }
  // unpack a 16-B chunk into EPC floats / pack back
  __device__ static __forceinline__ void unpack(const u32x4& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; i++) f[i] = __uint_as_float(c[i]);
  }
  __device__ static __forceinline__ u32x4 pack(const float* f) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = __float_as_uint(f[i]);
    return c;
  }
};
template <>
struct ST<bf16_t> {
  static constexpr int EPC = 8;
  static constexpr int DT = HDF_BF16;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(p->v); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { p->v = f2bf(v); }
  __device__ static __forceinline__ void st4(bf16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf2(a, b), pack_bf2(c, d)};
  }
  __device__ static __forceinline__ void ld4(const bf16_t* p, float* f) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(p);
    f[0] = __uint_as_float(v[0] << 16), f[1] = __uint_as_float(v[0] & 0xffff0000u);
    f[2] = __uint_as_float(v[1] << 16), f[3] = __uint_as_float(v[1] & 0xffff0000u);
  }
  __device__ static __forceinline__ void unpack(const u32x4& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      f[2 * i] = __uint_as_float(c[i] << 16);
      f[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ u32x4 pack(const float* f) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
    return c;
  }
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf2(lo, hi); }
};

template <>
struct ST<f16_t> {
  static constexpr int EPC = 8;
  static constexpr int DT = HDF_F16;
  __device__ static __forceinline__ float ld(const f16_t* p) { return h2f(p->v); }
  __device__ static __forceinline__ void st(f16_t* p, float v) { p->v = f2h(v); }
  __device__ static __forceinline__ void st4(f16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<u32x2*>(p) = u32x2{pack_h2(a, b), pack_h2(c, d)};
  }
  __device__ static __forceinline__ void ld4(const f16_t* p, float* f) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(p);
    const uint32_t w0 = v[0], w1 = v[1];  // (through scalars: see unpack)
    f[0] = h2f((uint16_t)(w0 & 0xffffu)), f[1] = h2f((uint16_t)(w0 >> 16));
    f[2] = h2f((uint16_t)(w1 & 0xffffu)), f[3] = h2f((uint16_t)(w1 >> 16));
  }
  __device__ static __forceinline__ void unpack(const u32x4& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      // (bit_cast straight from the vector element c[i] makes hipcc read element 0 four times: go through a scalar)
      const uint32_t w = c[i];
      f[2 * i] = h2f((uint16_t)(w & 0xffffu));
      f[2 * i + 1] = h2f((uint16_t)(w >> 16));
    }
  }
  __device__ static __forceinline__ u32x4 pack(const float* f) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = pack_h2(f[2 * i], f[2 * i + 1]);
    return c;
  }
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) { return pack_h2(lo, hi); }
};

// Two MFMA accumulator rows (v0, v1: the same output channel = this lane, two different voxels) stored as ONE dword per
// lane: even-channel lanes write (channel, channel + 1) of row 0, odd-channel lanes (channel - 1, channel) of row 1, after
// exchanging one value with lane ^ 1 (DPP quad_perm [1,0,3,2]).  16 lanes x 4 B = one 64-byte voxel row of 32 channels;
// a wave instruction covers four rows.  `p` = odd ? row1 + channel - 1 : row0 + channel (4-byte aligned), 16-bit T.
template <typename T>
__device__ __forceinline__ void st_rows2(T* p, float v0, float v1, bool odd) {
  const float give = odd ? v0 : v1;
  const float got = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(give), 0xB1, 0xF, 0xF, false));
  *reinterpret_cast<uint32_t*>(p) = ST<T>::pack2(odd ? got : v0, odd ? v1 : got);
}

// one dispatch for every kernel family templated on the storage type
#define HDF_DISPATCH_T(dtype, ...)                      \
  do {                                                  \
    if ((dtype) == HDF_BF16) {                          \
      using T = bf16_t;                                 \
      __VA_ARGS__;                                      \
    } else if ((dtype) == HDF_F32) {                    \
      using T = float;                                  \
      __VA_ARGS__;                                      \
    } else if ((dtype) == HDF_F16) {                    \
      using T = f16_t;                                  \
      __VA_ARGS__;                                      \
    } else {                                            \
      hdf_set_error("unsupported dtype %d", (dtype));   \
      return HDF_ERR_UNSUPPORTED;                       \
    }                                                   \
  } while (0)

// ---------------------------------------------------------------------------------------------
// dropout hash (same integer recipe as oracle/detgen.py: mix32 / dropout_keep)
__host__ __device__ __forceinline__ uint32_t hdf_mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t hdf_site_key(uint32_t seed, uint32_t site) {
  return hdf_mix32(seed ^ (site * 0x9E3779B1u));
}
// keep decision for flat element index idx under a site key; thresh24 = round((1-p)*2^24)
__host__ __device__ __forceinline__ bool hdf_keep(uint32_t key, uint32_t idx, uint32_t thresh24) {
  return (hdf_mix32(idx + key) >> 8) < thresh24;
}
__host__ __device__ __forceinline__ uint32_t hdf_site_id(int m, int b, int l, int kind) {
  return (uint32_t)(((m * 64 + b) * 8 + l) * 8 + kind);
}

// ---------------------------------------------------------------------------------------------
// wave helpers (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// Second pass of the InstanceNorm(+ReLU) backward for one element: g = d(activation), f = the raw conv output y; scale /
// shift / mean / rstd are the forward's constants of the (sample, channel), k1 / ka / kb come from in_bwd_finalize.  The
// arithmetic is pinned (explicit fma: left to -ffp-contract hipcc fuses differently per context) because the pass exists
// in four places that must round identically -- the stand-alone kernels (unet_ops.hip), the first layer's weight gradient
// (conv_first.hip) and the 16-bit weight gradient's staging (conv_wgrad2_kernel<.,.,true>); tests compare them bit for bit.
__device__ __forceinline__ float in_bwd_elem(float g, float f, float scale, float shift, float mean, float rstd, float k1,
                                             float ka, float kb) {
  const float gg = (__builtin_fmaf(f, scale, shift) > 0.f) ? g : 0.f;
  const float xh = (f - mean) * rstd;
  return k1 * __builtin_fmaf(-xh, kb, gg - ka);
}

// The transformer branches are ~100 short, latency-bound launches on the critical path that run NEXT TO the heavy
// persistent convolutions of the other streams (plan.hip: forward and backward3d fork).  Their waves co-reside with the
// conv waves on a SIMD; raising the wave priority lets their few instructions issue ahead of the conv's stream of
// MFMA / LDS work (the stream priority only orders workgroup DISPATCH).  HDF_NO_CHAIN_PRIO: A/B builds.
#ifdef HDF_NO_CHAIN_PRIO
#define HDF_CHAIN_PRIO() ((void)0)
#else
#define HDF_CHAIN_PRIO() __builtin_amdgcn_s_setprio(3)
#endif
// (round 5) The same for every LIGHT kernel -- normalisation statistics, pooling / up-sampling, heads: memory-bound passes of
// a few hundred instructions per thread -- and, through ConvArgs::prio, for the small convolutions of the UpConv chain.
// They are dispatched next to a persistent convolution of another stream (its workgroup leaves 130-180 registers per lane
// free), but at equal priority the issue arbiter serves the OLDEST wave first, and a conv wave at one wave per SIMD always
// has something to issue: up1's trilinear up-sampling (7 us alone) took 268 us beside the encoder's second 128^3 conv and
// held the whole UpConv chain -- which the caller's stream then waited 220 us for (tools/timeline.py).  HDF_NO_LIGHT_PRIO:
// A/B builds.
#ifdef HDF_NO_LIGHT_PRIO
#define HDF_LIGHT_PRIO() ((void)0)
#else
#define HDF_LIGHT_PRIO() __builtin_amdgcn_s_setprio(3)
#endif


