// HBM-bound U-Net ops: layout conversion, InstanceNorm statistics/finalise, norm+ReLU(+skip)
// materialisation, MaxPool3d(2,2) fwd/bwd, trilinear x2 fwd/bwd, the 1x1x1 heads fwd/bwd and the
// InstanceNorm+ReLU backward (reduce / finalise / apply).  Every thread moves 16-byte channel chunks
// of channels-last rows (8 bf16 / 4 f32), the coalescing sweet spot on CDNA4.
//
// Reference semantics: HDenseFormer.py:148-175 (BasicConv3d / UpConv), :199-207 (MaxPool3d),
// :223-227 (heads); torch semantics restated in SURVEY.md appendix A items 7-9,18.
#include "unet_ops.h"

namespace {

constexpr int MAX_BLOCKS = 4096;

template <typename T>
__device__ __forceinline__ void load_chunk(const T* p, float* f) {
  u32x4 v = *reinterpret_cast<const u32x4*>(p);
  ST<T>::unpack(v, f);
}
template <typename T>
__device__ __forceinline__ void store_chunk(T* p, const float* f) {
  *reinterpret_cast<u32x4*>(p) = ST<T>::pack(f);
}

// ------------------------------------------------------------------------------ layout conversion
template <typename T>
__global__ void nchw_to_ndhwc_kernel(const float* __restrict__ x, T* __restrict__ out, int N, int C, int CP,
                                     int64_t vox) {
  HDF_LIGHT_PRIO();
  int64_t total = (int64_t)N * vox;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t n = i / vox, v = i - n * vox;
    for (int c0 = 0; c0 < CP; c0 += ST<T>::EPC) {
      float f[ST<T>::EPC];
#pragma unroll
      for (int e = 0; e < ST<T>::EPC; e++) f[e] = (c0 + e < C) ? x[(n * C + c0 + e) * vox + v] : 0.f;
      store_chunk<T>(out + i * CP + c0, f);
    }
  }
}

// ------------------------------------------------------------------------------ IN statistics
// one workgroup per (n, 8-channel group): 32 tile lanes x 8 channels, 4 independent loads in flight per
// thread, double accumulation in a fixed order (bitwise reproducible).  256 threads and few registers ON PURPOSE: the
// 1024-thread form (32 x 32, 128 registers) needed a compute unit with every SIMD empty, and next to a persistent conv /
// weight-gradient kernel of another stream (one wave per SIMD, 300-430 registers) this 5 us kernel waited 60-95 us for
// that kernel to END (r03c timeline: five such waits on the caller's stream in one backward).
constexpr int FIN_LANES = 32, FIN_CG = 8;
__global__ __launch_bounds__(256, 4) void in_finalize_kernel(const float* __restrict__ partials, int tiles, int C, int CP,
                                                           int64_t vox, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           float* __restrict__ mean, float* __restrict__ rstd,
                                                           float* __restrict__ scale, float* __restrict__ shift) {
  HDF_LIGHT_PRIO();
  __shared__ double red[FIN_LANES][FIN_CG][2];
  const int n = blockIdx.y, cg = blockIdx.x;
  const int cl = threadIdx.x & (FIN_CG - 1), c = cg * FIN_CG + cl, tl = threadIdx.x / FIN_CG;
  double s1 = 0.0, s2 = 0.0;
  if (c < CP) {
    const float* p = partials + ((int64_t)n * tiles * CP + c) * 2;
    const int64_t ts = (int64_t)CP * 2;
    int t = tl;
    for (; t + 3 * FIN_LANES < tiles; t += 4 * FIN_LANES) {
      float2 v0 = *reinterpret_cast<const float2*>(p + (int64_t)t * ts);
      float2 v1 = *reinterpret_cast<const float2*>(p + (int64_t)(t + FIN_LANES) * ts);
      float2 v2 = *reinterpret_cast<const float2*>(p + (int64_t)(t + 2 * FIN_LANES) * ts);
      float2 v3 = *reinterpret_cast<const float2*>(p + (int64_t)(t + 3 * FIN_LANES) * ts);
      s1 += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
      s2 += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    }
    for (; t < tiles; t += FIN_LANES) {
      float2 v = *reinterpret_cast<const float2*>(p + (int64_t)t * ts);
      s1 += (double)v.x;
      s2 += (double)v.y;
    }
  }
  red[tl][cl][0] = s1;
  red[tl][cl][1] = s2;
  __syncthreads();
  if (tl == 0 && c < C) {
    for (int k = 1; k < FIN_LANES; k++) {
      s1 += red[k][cl][0];
      s2 += red[k][cl][1];
    }
    double m = s1 / (double)vox;
    double var = s2 / (double)vox - m * m;  // biased variance
    if (var < 0.0) var = 0.0;
    float r = (float)(1.0 / sqrt(var + (double)eps));
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    int64_t o = (int64_t)n * C + c;
    mean[o] = (float)m;
    rstd[o] = r;
    scale[o] = g * r;
    shift[o] = b - (float)m * g * r;
  }
}

// ------------------------------------------------------------------------------ norm+relu(+skip)
template <typename T>
__global__ void norm_relu_add_kernel(const T* __restrict__ y, int64_t y_pitch, const float* __restrict__ scale,
                                     const float* __restrict__ shift, const T* __restrict__ skip, int64_t skip_pitch,
                                     T* __restrict__ out, int64_t out_pitch, int N, int C, int64_t vox) {
  HDF_LIGHT_PRIO();
  constexpr int EPC = ST<T>::EPC;
  const int cols = C / EPC;
  int64_t total = (int64_t)N * vox * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t row = i / cols;
    int c0 = (int)(i - row * cols) * EPC;
    int n = (int)(row / vox);
    float f[EPC], s[EPC];
    load_chunk<T>(y + row * y_pitch + c0, f);
    if (skip) load_chunk<T>(skip + row * skip_pitch + c0, s);
#pragma unroll
    for (int e = 0; e < EPC; e++) {
      float v = fmaxf(f[e] * scale[(int64_t)n * C + c0 + e] + shift[(int64_t)n * C + c0 + e], 0.f);
      f[e] = skip ? v + s[e] : v;
    }
    store_chunk<T>(out + row * out_pitch + c0, f);
  }
}

// ------------------------------------------------------------------------------ maxpool 2x2x2
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ in, int64_t in_pitch, T* __restrict__ out, int64_t out_pitch,
                                   uint8_t* __restrict__ idx, int N, int C, int Do, int Ho, int Wo) {
  constexpr int EPC = ST<T>::EPC;
  const int cols = C / EPC;
  const int Hi = Ho * 2, Wi = Wo * 2, Di = Do * 2;
  int64_t total = (int64_t)N * Do * Ho * Wo * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t row = i / cols;
    int c0 = (int)(i - row * cols) * EPC;
    int64_t t = row;
    int ow = t % Wo;
    t /= Wo;
    int oh = t % Ho;
    t /= Ho;
    int od = t % Do;
    int n = (int)(t / Do);
    float best[EPC];
    int bi[EPC];
#pragma unroll
    for (int e = 0; e < EPC; e++) {
      best[e] = -INFINITY;
      bi[e] = 0;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {  // scan order d,h,w; strict > keeps the FIRST maximum (torch tie rule)
      int dz = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
      int64_t irow = (((int64_t)n * Di + 2 * od + dz) * Hi + 2 * oh + dy) * Wi + 2 * ow + dx;
      float f[EPC];
      load_chunk<T>(in + irow * in_pitch + c0, f);
#pragma unroll
      for (int e = 0; e < EPC; e++) {
        if (f[e] > best[e] || f[e] != f[e]) {
          best[e] = f[e];
          bi[e] = k;
        }
      }
    }
    store_chunk<T>(out + row * out_pitch + c0, best);
    if constexpr (EPC == 8) {
      uint32_t lo = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
      uint32_t hi = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
      *reinterpret_cast<uint2*>(idx + row * C + c0) = make_uint2(lo, hi);
    } else {
      *reinterpret_cast<uint32_t*>(idx + row * C + c0) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    }
  }
}

template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dout, int64_t dout_pitch, const uint8_t* __restrict__ idx,
                                   T* __restrict__ din, int64_t din_pitch, int N, int C, int Do, int Ho, int Wo,
                                   int accumulate) {
  constexpr int EPC = ST<T>::EPC;
  const int cols = C / EPC;
  const int Hi = Ho * 2, Wi = Wo * 2, Di = Do * 2;
  int64_t total = (int64_t)N * Do * Ho * Wo * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t row = i / cols;
    int c0 = (int)(i - row * cols) * EPC;
    int64_t t = row;
    int ow = t % Wo;
    t /= Wo;
    int oh = t % Ho;
    t /= Ho;
    int od = t % Do;
    int n = (int)(t / Do);
    float g[EPC];
    int bi[EPC];
    load_chunk<T>(dout + row * dout_pitch + c0, g);
    if constexpr (EPC == 8) {
      uint2 pk = *reinterpret_cast<const uint2*>(idx + row * C + c0);
#pragma unroll
      for (int e = 0; e < 4; e++) bi[e] = (pk.x >> (8 * e)) & 255, bi[4 + e] = (pk.y >> (8 * e)) & 255;
    } else {
      uint32_t pk = *reinterpret_cast<const uint32_t*>(idx + row * C + c0);
#pragma unroll
      for (int e = 0; e < 4; e++) bi[e] = (pk >> (8 * e)) & 255;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      int dz = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
      int64_t irow = (((int64_t)n * Di + 2 * od + dz) * Hi + 2 * oh + dy) * Wi + 2 * ow + dx;
      T* p = din + irow * din_pitch + c0;
      float f[EPC];
      if (accumulate)
        load_chunk<T>(p, f);
      else {
#pragma unroll
        for (int e = 0; e < EPC; e++) f[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < EPC; e++)
        if (bi[e] == k) f[e] += g[e];
      store_chunk<T>(p, f);
    }
  }
}

// ------------------------------------------------------------------------------ fused encoder tail
// ds = relu(y*s+t) + skip ;  pooled, idx = MaxPool3d(2)(ds)        (HDenseFormer.py:237-243)
// One thread owns a 2x2x2 block of ds voxels (one pooled voxel) x one 16-byte channel chunk, so ds is written
// once and never re-read for pooling.  (Round 1 also evaluated the trilinear x2 of up3's output inside this kernel so
// that at3 was never materialised; with the 27-loads-per-8-outputs upsample kernel reading the materialised tensor is
// 0.09 ms per step faster, and that variant -- VALU-bound with register spills -- was removed in round 3.)
// FLAT (round 6): the 2-D form (models/HDenseFormer_2D.py:232-240: MaxPool2d(2)) on depth-1 tensors -- a thread owns a
// 1x2x2 block, the depth axis is not pooled; window index k = 2 dy + dx as in the 3-D form with dz = 0.
template <typename T, bool FLAT = false>
__global__ __launch_bounds__(256, 2) void enc_tail_kernel(const T* __restrict__ y, int64_t y_pitch,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const T* __restrict__ skip,
                                                          int64_t skip_pitch, T* __restrict__ ds,
                                                          int64_t ds_pitch, T* __restrict__ pooled,
                                                          int64_t pooled_pitch, uint8_t* __restrict__ idx, int N, int C,
                                                          int Do, int Ho, int Wo) {
  HDF_LIGHT_PRIO();
  constexpr int EPC = ST<T>::EPC;
  const int cols = C / EPC;
  const int Hi = 2 * Ho, Wi = 2 * Wo;
  const int64_t total = (int64_t)N * Do * Ho * Wo * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / cols;
    const int c0 = (int)(i - row * cols) * EPC;
    // (n, od) by one 64-bit division, the in-plane coordinates in 32 bits
    const int plane = Ho * Wo;
    const int64_t nz = row / plane;
    const int rem = (int)(row - nz * plane);
    const int oh = rem / Wo, ow = rem - oh * Wo;
    const int od = (int)(nz % Do), n = (int)(nz / Do);
    float sc[EPC], sh[EPC], best[EPC];
    int bi[EPC];
#pragma unroll
    for (int e = 0; e < EPC; e++) {
      sc[e] = scale[(int64_t)n * C + c0 + e];
      sh[e] = shift[(int64_t)n * C + c0 + e];
      best[e] = -INFINITY;
      bi[e] = 0;
    }
    // one output z plane (dz) of the 2x2x2 block: ds = relu(y*s+t) + skip, stored, and folded into the running
    // maximum in scan order d,h,w (strict > keeps the FIRST maximum: torch's tie rule)
    const int64_t row0 = (((int64_t)nz * (FLAT ? 1 : 2)) * Hi + 2 * oh) * Wi + 2 * ow;  // first voxel of the 2x2x2 block
    const T* const ybase = y + row0 * y_pitch + c0;
    T* const dbase = ds + row0 * ds_pitch + c0;
    auto finish_plane = [&](int dz, const float (&sk)[4][EPC]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int dy = q >> 1, dx = q & 1, k = dz * 4 + q;
        const int64_t orow = ((int64_t)dz * Hi + dy) * Wi + dx;  // uniform
        float f[EPC];
        load_chunk<T>(ybase + orow * y_pitch, f);
#pragma unroll
        for (int e = 0; e < EPC; e++) f[e] = fmaxf(f[e] * sc[e] + sh[e], 0.f) + sk[q][e];
        store_chunk<T>(dbase + orow * ds_pitch, f);
        // pool over the STORED (storage-rounded) values so that backward/recompute sees the same maxima
        float g[EPC];
        ST<T>::unpack(ST<T>::pack(f), g);
#pragma unroll
        for (int e = 0; e < EPC; e++) {
          if (g[e] > best[e] || g[e] != g[e]) {
            best[e] = g[e];
            bi[e] = k;
          }
        }
      }
    };
#pragma unroll
    for (int dz = 0; dz < (FLAT ? 1 : 2); dz++) {
      float sk[4][EPC];
#pragma unroll
      for (int q = 0; q < 4; q++)
        load_chunk<T>(skip + (row0 + ((int64_t)dz * Hi + (q >> 1)) * Wi + (q & 1)) * skip_pitch + c0, sk[q]);
      finish_plane(dz, sk);
    }
    store_chunk<T>(pooled + row * pooled_pitch + c0, best);
    if constexpr (EPC == 8) {
      uint32_t lo = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
      uint32_t hi = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
      *reinterpret_cast<uint2*>(idx + row * C + c0) = make_uint2(lo, hi);
    } else {
      *reinterpret_cast<uint32_t*>(idx + row * C + c0) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    }
  }
}

// ------------------------------------------------------------------------------ trilinear x2
// per dim, output o reads inputs (ia, wa), (ib, wb):  o=2i: (max(i-1,0), .25), (i, .75) ; o=2i+1: (i, .75), (min(i+1,n-1), .25)
__device__ __forceinline__ void up_taps(int o, int n, int& ia, float& wa, int& ib, float& wb) {
  int i = o >> 1;
  if (o & 1) {
    ia = i;
    wa = 0.75f;
    ib = min(i + 1, n - 1);
    wb = 0.25f;
  } else {
    ia = max(i - 1, 0);
    wa = 0.25f;
    ib = i;
    wb = 0.75f;
  }
}

// One thread per LOW-resolution voxel chunk: it produces the 2 x 2 x 2 block of outputs from the 3 x 3 x 3 input
// neighbourhood (27 loads and transforms per 8 outputs, separable interpolation plane by plane: x, then y, then z).  The
// first version gave every OUTPUT chunk its own 8 loads + 8 transforms (64 per block) and was VALU-bound: 204 us for the
// 268 MB of at3 (1.5 TB/s).  Edge voxels: the clamped neighbour index makes the .25 / .75 pair collapse onto the same
// voxel, which is torch's align_corners=False edge rule.
// FLAT (round 6): bilinear x2 of a depth-1 tensor (models/HDenseFormer_2D.py:166-170: F.interpolate(scale_factor=2,
// mode='bilinear', align_corners=False)): the y / x arithmetic of the 3-D form, the depth axis untouched.
template <typename T, bool FLAT = false>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const T* __restrict__ y, int64_t y_pitch,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, T* __restrict__ out,
                                                           int64_t out_pitch, int N, int C, int Di, int Hi, int Wi) {
  HDF_LIGHT_PRIO();
  constexpr int EPC = ST<T>::EPC;
  const int cols = C / EPC;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  const int64_t total = (int64_t)N * Di * Hi * Wi * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / cols;
    const int c0 = (int)(i - row * cols) * EPC;
    const int plane = Hi * Wi;
    const int64_t nz = row / plane;
    const int rem = (int)(row - nz * plane);
    const int ih = rem / Wi, iw = rem - ih * Wi;
    const int id = (int)(nz % Di), n = (int)(nz / Di);
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; e++) {
      sc[e] = scale[(int64_t)n * C + c0 + e];
      sh[e] = shift[(int64_t)n * C + c0 + e];
    }
    const int xs[3] = {max(iw - 1, 0), iw, min(iw + 1, Wi - 1)};
    const int ys[3] = {max(ih - 1, 0), ih, min(ih + 1, Hi - 1)};
    const int zs[3] = {max(id - 1, 0), id, min(id + 1, Di - 1)};
    const T* const lbase = y + (int64_t)n * Di * Hi * Wi * y_pitch + c0;
    const int lp = (int)y_pitch;
    // low-resolution plane a interpolated in y and x: P[dy][dx]
    auto plane_yx = [&](int a, float (&P)[4][EPC]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int e = 0; e < EPC; e++) P[q][e] = 0.f;
#pragma unroll
      for (int b = 0; b < 3; b++) {
        float L[3][EPC];
#pragma unroll
        for (int c = 0; c < 3; c++) {
          float f[EPC];
          load_chunk<T>(lbase + ((int64_t)(zs[a] * Hi + ys[b]) * Wi + xs[c]) * lp, f);
#pragma unroll
          for (int e = 0; e < EPC; e++) L[c][e] = fmaxf(f[e] * sc[e] + sh[e], 0.f);
        }
        const float wy0 = (b == 0) ? 0.25f : (b == 1 ? 0.75f : 0.f);  // weight of row b for dy = 0
        const float wy1 = (b == 0) ? 0.f : (b == 1 ? 0.75f : 0.25f);   // ... for dy = 1
#pragma unroll
        for (int e = 0; e < EPC; e++) {
          const float x0 = 0.25f * L[0][e] + 0.75f * L[1][e], x1 = 0.75f * L[1][e] + 0.25f * L[2][e];
          P[0][e] += wy0 * x0, P[1][e] += wy0 * x1;
          P[2][e] += wy1 * x0, P[3][e] += wy1 * x1;
        }
      }
    };
    T* const obase = out + ((((int64_t)nz * (FLAT ? 1 : 2)) * Ho + 2 * ih) * Wo + 2 * iw) * out_pitch + c0;
    auto store_plane = [&](int dz, const float (&A)[4][EPC], float wa, const float (&B)[4][EPC], float wb)
        __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float f[EPC];
#pragma unroll
        for (int e = 0; e < EPC; e++) f[e] = wa * A[q][e] + wb * B[q][e];
        store_chunk<T>(obase + (((int64_t)dz * Ho + (q >> 1)) * Wo + (q & 1)) * out_pitch, f);
      }
    };
    float P0[4][EPC], P1[4][EPC];
    if constexpr (FLAT) {
      plane_yx(1, P1);                       // zs[1] = id: the tensor's one plane
      store_plane(0, P1, 1.f, P1, 0.f);
    } else {
      plane_yx(0, P0);
      plane_yx(1, P1);
      store_plane(0, P0, 0.25f, P1, 0.75f);
      plane_yx(2, P0);
      store_plane(1, P1, 0.75f, P0, 0.25f);
    }
  }
}

// ------------------------------------------------------------------------------ encoder tail with the up-sampling inside
// ds = relu(y*s+t) + trilinear_x2(relu(low*ls+lt)) ;  pooled, idx = MaxPool3d(2)(ds)     (HDenseFormer.py:168-175,237-243)
// enc_tail_kernel and upsample_fwd_kernel share their decomposition -- a thread owns one LOW-resolution voxel chunk, i.e.
// one 2 x 2 x 2 block of ds -- so the level-0 feature at3 = up3(...) need not exist in memory: this kernel interpolates
// the block from the 3 x 3 x 3 low-resolution neighbourhood (the same separable x, y, z arithmetic as
// upsample_fwd_kernel, plane by plane) and adds it in registers.  What it is for: at3's up-sampling (268 MB written at
// 128^3, batch 2: 103 us) was the LAST launch of the transformer / UpConv chain the caller's stream waits for in the
// forward (plan.hip: forward3d); with it here the wait ends at up3's InstanceNorm statistics, and this pass reads the
// 33 MB low-resolution tensor (L2-resident neighbours) instead of 268 MB.  The interpolated value is added in fp32
// (the materialised at3 was rounded to the storage type first).
// Registers: two interpolated planes (64 floats at 8 channels) + four norm vectors + the running maxima: one workgroup
// of 256 per SIMD set, no spills (the round-1 attempt at this fusion kept all eight interpolated chunks live and spilled).
template <typename T>
__global__ __launch_bounds__(256, 2) void enc_tail_up_kernel(const T* __restrict__ y, int64_t y_pitch,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const T* __restrict__ low,
                                                             int64_t low_pitch, const float* __restrict__ lscale,
                                                             const float* __restrict__ lshift, T* __restrict__ ds,
                                                             int64_t ds_pitch, T* __restrict__ pooled,
                                                             int64_t pooled_pitch, uint8_t* __restrict__ idx, int N, int C,
                                                             int Do, int Ho, int Wo) {
  HDF_LIGHT_PRIO();
  constexpr int EPC = 4;  // four channels per thread (8-byte accesses in the 16-bit modes): two workgroups per SIMD set
  // grid (x tiles, oh, n * Do + od): the plane and row coordinates are uniform per workgroup, so every base address is
  // scalar arithmetic and a thread adds 32-bit offsets inside one row (the first version, a flat index with 64-bit
  // divisions and one 64-bit multiply per neighbour, spent a third of its vector cycles on addresses)
  const int cols = C / EPC;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Wo * cols) return;
  const int ow = t / cols, c0 = (t - ow * cols) * EPC;
  const int oh = blockIdx.y;
  const int nz = blockIdx.z, n = nz / Do, od = nz - n * Do;
  const int Hi = 2 * Ho, Wi = 2 * Wo;
  const int yp = (int)y_pitch, dp = (int)ds_pitch, lp = (int)low_pitch;
  float sc[EPC], sh[EPC], lsc[EPC], lsh[EPC], best[EPC];
  int bi[EPC];
#pragma unroll
  for (int e = 0; e < EPC; e++) {
    sc[e] = scale[(int64_t)n * C + c0 + e];
    sh[e] = shift[(int64_t)n * C + c0 + e];
    lsc[e] = lscale[(int64_t)n * C + c0 + e];
    lsh[e] = lshift[(int64_t)n * C + c0 + e];
    best[e] = -INFINITY;
    bi[e] = 0;
  }
  // ---- the low-resolution neighbourhood (upsample_fwd_kernel's plane_yx, same order of operations)
  const int zs[3] = {max(od - 1, 0), od, min(od + 1, Do - 1)};
  const int ys[3] = {max(oh - 1, 0), oh, min(oh + 1, Ho - 1)};
  const int xo[3] = {max(ow - 1, 0) * lp + c0, ow * lp + c0, min(ow + 1, Wo - 1) * lp + c0};
  auto plane_yx = [&](int a, float (&P)[4][EPC]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int e = 0; e < EPC; e++) P[q][e] = 0.f;
#pragma unroll
    for (int b = 0; b < 3; b++) {
      const T* const lrow = low + (((int64_t)n * Do + zs[a]) * Ho + ys[b]) * Wo * low_pitch;  // uniform
      float L[3][EPC];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        float f[EPC];
        ST<T>::ld4(lrow + xo[c], f);
#pragma unroll
        for (int e = 0; e < EPC; e++) L[c][e] = fmaxf(f[e] * lsc[e] + lsh[e], 0.f);
      }
      const float wy0 = (b == 0) ? 0.25f : (b == 1 ? 0.75f : 0.f);
      const float wy1 = (b == 0) ? 0.f : (b == 1 ? 0.75f : 0.25f);
#pragma unroll
      for (int e = 0; e < EPC; e++) {
        const float x0 = 0.25f * L[0][e] + 0.75f * L[1][e], x1 = 0.75f * L[1][e] + 0.25f * L[2][e];
        P[0][e] += wy0 * x0, P[1][e] += wy0 * x1;
        P[2][e] += wy1 * x0, P[3][e] += wy1 * x1;
      }
    }
  };
  // first voxel row of the workgroup's 2 x 2 x (2 Wo) slab (uniform) + this thread's x offset
  const int64_t slab = (((int64_t)nz * 2) * Hi + 2 * oh) * Wi;
  const T* const ybase = y + slab * y_pitch + 2 * ow * yp + c0;
  T* const dbase = ds + slab * ds_pitch + 2 * ow * dp + c0;
  // one output z plane of the block: ds = relu(y*s+t) + (wa A + wb B), stored, folded into the running maximum in scan
  // order d,h,w (strict > keeps the FIRST maximum: torch's tie rule; over the STORED values, as enc_tail_kernel)
  auto finish_plane = [&](int dz, const float (&A)[4][EPC], float wa, const float (&B)[4][EPC], float wb)
      __attribute__((always_inline)) {
    float yv[4][EPC];
#pragma unroll
    for (int q = 0; q < 4; q++) ST<T>::ld4(ybase + ((int64_t)(dz * Hi + (q >> 1)) * Wi) * y_pitch + (q & 1) * yp, yv[q]);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = dz * 4 + q;
      float f[EPC];
#pragma unroll
      for (int e = 0; e < EPC; e++) f[e] = fmaxf(yv[q][e] * sc[e] + sh[e], 0.f) + (wa * A[q][e] + wb * B[q][e]);
      ST<T>::st4(dbase + ((int64_t)(dz * Hi + (q >> 1)) * Wi) * ds_pitch + (q & 1) * dp, f[0], f[1], f[2], f[3]);
      float g[EPC];
#pragma unroll
      for (int e = 0; e < EPC; e++) {
        T tmp;
        ST<T>::st(&tmp, f[e]);
        g[e] = ST<T>::ld(&tmp);
      }
#pragma unroll
      for (int e = 0; e < EPC; e++) {
        if (g[e] > best[e] || g[e] != g[e]) {
          best[e] = g[e];
          bi[e] = k;
        }
      }
    }
  };
  float P0[4][EPC], P1[4][EPC];
  plane_yx(0, P0);
  plane_yx(1, P1);
  finish_plane(0, P0, 0.25f, P1, 0.75f);
  plane_yx(2, P0);
  finish_plane(1, P1, 0.75f, P0, 0.25f);
  const int64_t prow = ((int64_t)nz * Ho + oh) * Wo + ow;
  ST<T>::st4(pooled + prow * pooled_pitch + c0, best[0], best[1], best[2], best[3]);
  *reinterpret_cast<uint32_t*>(idx + prow * C + c0) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
}

// input i receives from outputs 2i-1 (.25, i>=1), 2i (.75 [+.25 at i==0]), 2i+1 (.75 [+.25 at i==n-1]), 2i+2 (.25, i<=n-2)
__device__ __forceinline__ void up_bwd_taps(int i, int n, int* o, float* w) {
  o[0] = 2 * i - 1;
  w[0] = (i >= 1) ? 0.25f : 0.f;
  o[1] = 2 * i;
  w[1] = (i == 0) ? 1.0f : 0.75f;
  o[2] = 2 * i + 1;
  w[2] = (i == n - 1) ? 1.0f : 0.75f;
  o[3] = 2 * i + 2;
  w[3] = (i <= n - 2) ? 0.25f : 0.f;
  if (i < 1) o[0] = 0;
  if (i > n - 2) o[3] = 2 * n - 1;
}

// XCD-aware workgroup order.  The dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, each with its own
// L2: with tiles in raster order, a tile's neighbours in y and z -- which share its halo -- run on OTHER XCDs and every
// halo row is fetched from the fabric once per XCD that touches it.  logical tile = xcd * (W / 8) + id / 8 gives every
// XCD one contiguous range of the raster instead (W a multiple of 8; else the identity).
__device__ __forceinline__ int xcd_slab_id(int L, int W) { return (W & 7) ? L : (L & 7) * (W >> 3) + (L >> 3); }

// One thread per low-resolution voxel chunk: 4 x 4 x 4 output taps.  Grid = x tiles * Hi * (N * Di) workgroups, one
// low-resolution row segment each: plane and row are uniform per workgroup (scalar address arithmetic; the flat-index
// version decoded its coordinates with 64-bit divisions), and the XCD slab order above keeps the 16 output rows a
// workgroup reads in the L2 that read them for the previous row (the raster order fetched 1.1 GB per step for 0.35 GB of
// operands: three XCDs per output row).
template <typename T, bool FLAT = false>   // FLAT: the adjoint of the bilinear x2 of a depth-1 tensor (4 x 4 taps)
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ dout, int64_t dout_pitch,
                                                           T* __restrict__ din, int64_t din_pitch, int N, int C, int Di,
                                                           int Hi, int Wi, int gx) {
  HDF_LIGHT_PRIO();
  constexpr int EPC = ST<T>::EPC;
  const int cols = C / EPC;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  const int vb = xcd_slab_id(blockIdx.x, gridDim.x);
  const int xt = vb % gx, r = vb / gx;
  const int ih = r % Hi, nz = r / Hi;
  const int id = nz % Di, n = nz / Di;
  const int t = xt * 256 + threadIdx.x;
  if (t >= Wi * cols) return;
  const int iw = t / cols, c0 = (t - iw * cols) * EPC;
  int oz[4], oy[4], ox[4];
  float wz[4], wy[4], wx[4];
  if constexpr (FLAT) {
#pragma unroll
    for (int q = 0; q < 4; q++) oz[q] = 0, wz[q] = q == 0 ? 1.f : 0.f;
  } else {
    up_bwd_taps(id, Di, oz, wz);
  }
  up_bwd_taps(ih, Hi, oy, wy);
  up_bwd_taps(iw, Wi, ox, wx);
  float acc[EPC];
#pragma unroll
  for (int e = 0; e < EPC; e++) acc[e] = 0.f;
  // all 64 taps unconditionally (indices are clamped into range, out-of-range taps carry weight 0): no branch around
  // any load, so the loads of a thread are all in flight together.  Addresses: one 64-bit sample base, 32-bit element
  // offsets z + y (uniform) + x (the launcher checks that a sample fits 2^31 elements)
  const T* const sbase = dout + (int64_t)n * ((FLAT ? 1 : 2) * Di) * Ho * Wo * dout_pitch + c0;
  const int pit = (int)dout_pitch;
  int zo[4], yo[4], xo[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    zo[q] = oz[q] * Ho * Wo * pit;
    yo[q] = oy[q] * Wo * pit;
    xo[q] = ox[q] * pit;
  }
#pragma unroll
  for (int a = 0; a < (FLAT ? 1 : 4); a++) {
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const float wzy = wz[a] * wy[b];
      const int zy = zo[a] + yo[b];
      float f[4][EPC];
#pragma unroll
      for (int c = 0; c < 4; c++) load_chunk<T>(sbase + (zy + xo[c]), f[c]);
#pragma unroll
      for (int c = 0; c < 4; c++)
#pragma unroll
        for (int e = 0; e < EPC; e++) acc[e] += (wzy * wx[c]) * f[c][e];
    }
  }
  store_chunk<T>(din + (((int64_t)nz * Hi + ih) * Wi + iw) * din_pitch + c0, acc);
}

// ------------------------------------------------------------------------------ 1x1x1 heads
constexpr int HEAD_MAXCLS = 8;

// Streaming form (as head_bwd below): thread = (voxel lane, 8-channel chunk) so a wave reads whole contiguous voxel
// rows; the per-chunk partial dot products are summed over the `cols` chunk lanes of a voxel with xor shuffles
// (cols is a power of two <= 32), the block's logits are staged in LDS as [class][voxel] and written plane by plane,
// coalesced.  (One thread per voxel walked a 64..512-byte row alone: 58 us for the 4 MB of the 16^3 level.)
template <typename T, int MC>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ in, int64_t in_pitch,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ w, const float* __restrict__ b,
                                                       T* __restrict__ logits, int N, int C, int ncls, int64_t vox,
                                                       int per, int vec4) {
  HDF_LIGHT_PRIO();
  constexpr int EPC = ST<T>::EPC;
  extern __shared__ float outs[];  // [MC][per], then [256][MC] partials when cols is not a power of two
  float* part = outs + MC * per;
  const int n = blockIdx.y;
  const int cols = C / EPC;
  // shuffle reduction when a voxel's chunk lanes are a power of two within one wave; else (n_filters = 48: 6 chunk
  // lanes straddle waves; fp32 storage at 512 channels: 128 lanes per voxel) through LDS
  const bool pow2 = (cols & (cols - 1)) == 0 && cols <= 64;
  const int vlanes = 256 / cols;
  const int col = threadIdx.x % cols, vl = min((int)threadIdx.x / cols, vlanes - 1);
  const bool lane_on = (int)threadIdx.x < vlanes * cols;
  const int c0 = col * EPC;
  float sc[EPC], sh[EPC], wv[MC][EPC];
#pragma unroll
  for (int e = 0; e < EPC; e++) {
    sc[e] = scale ? scale[(int64_t)n * C + c0 + e] : 1.f;
    sh[e] = scale ? shift[(int64_t)n * C + c0 + e] : 0.f;
#pragma unroll
    for (int o = 0; o < MC; o++) wv[o][e] = (o < ncls) ? w[min(o, ncls - 1) * C + c0 + e] : 0.f;
  }
  const int64_t vb = (int64_t)blockIdx.x * per, ve = min(vox, vb + per);
  constexpr int U = 4;
  const int iters = (per + U * vlanes - 1) / (U * vlanes);  // the same for every thread: shuffles / barriers below
  for (int it = 0; it < iters; it++) {
    const int64_t v0 = vb + vl + (int64_t)it * U * vlanes;
    float f[U][EPC];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t v = min(v0 + (int64_t)u * vlanes, ve - 1);  // clamped: never branch around a load
      load_chunk<T>(in + ((int64_t)n * vox + v) * in_pitch + c0, f[u]);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      float acc[MC];
#pragma unroll
      for (int o = 0; o < MC; o++) acc[o] = 0.f;
#pragma unroll
      for (int e = 0; e < EPC; e++) {
        float x = f[u][e];
        if (scale) x = fmaxf(x * sc[e] + sh[e], 0.f);
#pragma unroll
        for (int o = 0; o < MC; o++) acc[o] += x * wv[o][e];
      }
      if (pow2) {
        for (int off = 1; off < cols; off <<= 1) {
#pragma unroll
          for (int o = 0; o < MC; o++) acc[o] += __shfl_xor(acc[o], off, 64);
        }
      } else {  // uniform branch and trip count: the barriers are reached by every thread
        __syncthreads();
#pragma unroll
        for (int o = 0; o < MC; o++) part[threadIdx.x * MC + o] = acc[o];
        __syncthreads();
        if (col == 0) {
          for (int k = 1; k < cols; k++)
#pragma unroll
            for (int o = 0; o < MC; o++) acc[o] += part[(threadIdx.x + k) * MC + o];
        }
      }
      const int64_t lv = v0 + (int64_t)u * vlanes - vb;
      if (col == 0 && lane_on && lv < per) {
#pragma unroll
        for (int o = 0; o < MC; o++) outs[o * per + (int)lv] = acc[o];
      }
    }
  }
  __syncthreads();
  for (int o = 0; o < ncls; o++) {
    const float bo = b[o];
    if (vec4) {  // four voxels of the class plane per store
      for (int lv = threadIdx.x * 4; lv < per; lv += 1024)
        if (vb + lv < vox)
          ST<T>::st4(logits + ((int64_t)n * ncls + o) * vox + vb + lv, outs[o * per + lv] + bo, outs[o * per + lv + 1] + bo,
                     outs[o * per + lv + 2] + bo, outs[o * per + lv + 3] + bo);
    } else {
      for (int lv = threadIdx.x; lv < per; lv += 256)
        if (vb + lv < vox) ST<T>::st(logits + ((int64_t)n * ncls + o) * vox + vb + lv, outs[o * per + lv] + bo);
    }
  }
}

// Streaming form: thread = (voxel lane, 8-channel chunk), so a wave reads whole contiguous voxel rows; the weight-
// gradient outer products accumulate in registers over the thread's voxels (ncls x 8 accumulators) and are reduced
// across the block ONCE at the end (the first version rebuilt a 256-voxel LDS tile and ran a 256-deep serial LDS
// reduction per tile: 3x the HBM time).  grid (blocks, N); C <= 256 (cols = C/8 <= 32).
constexpr int HEAD_VOX_MAX = 2048;  // voxels per workgroup (fewer at the low-resolution levels: see head_vox)
// voxels per workgroup: a multiple of 64; at least 128 (backward: every workgroup ends with a block reduction and 132
// atomics; 512 per sample was slower, 39.8 -> 49.4 us at 64^3) / 1024 (forward: with 128 per sample the 64^3 level ran
// one workgroup per CU at 1.7 TB/s) workgroups per sample when the level has that many 64-voxel groups
inline int head_vox(int64_t vox, int wgs = 128) {
  return (int)std::max<int64_t>(64, std::min<int64_t>(HEAD_VOX_MAX, (vox / wgs) & ~63));
}
// storage rounding of one value (what a later pass would read back)
template <typename T>
__device__ __forceinline__ float storage_round(float v) {
  T t;
  ST<T>::st(&t, v);
  return ST<T>::ld(&t);
}
// MC: class slots held in registers (4 or 8); ACC: dx += (else dx =).  FOUR channels per thread for every storage type
// (16-bit: 8-byte loads, two voxel rows in flight, <= 128 registers).  The 8-channel form for 16-bit storage needed 238
// registers: below the top level the head gradient runs next to a persistent weight-gradient kernel of the side stream
// (one wave per SIMD, 301-376 registers) and WAITED for that kernel to end (179 us instead of 58 at 64^3 in the r03c
// timeline); alone the light form is as fast at 128^3 (151 vs 159 us), slower at 64^3 (52 vs 38 us).
template <typename T, int MC, bool ACC>
__global__ __launch_bounds__(256, (MC == 4 && sizeof(T) == 2) ? 4 : 1) void head_bwd_kernel(const T* __restrict__ dlogits, const T* __restrict__ in,
                                                       int64_t in_pitch, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ w,
                                                       T* __restrict__ dx, int64_t dx_pitch,
                                                       float* __restrict__ dw, float* __restrict__ db, int N, int C,
                                                       int ncls, int64_t vox, int per,
                                                       const float* __restrict__ in_mean,
                                                       const float* __restrict__ in_rstd,
                                                       float* __restrict__ inb_partials, int vec4) {
  HDF_LIGHT_PRIO();
  constexpr int EPC = 4;
  // inb_partials (optional): this kernel produces the complete gradient dx of the activation relu(IN(in)), so it
  // also writes the first pass of that InstanceNorm's backward -- per workgroup and channel (sum g, sum g*xhat) with
  // g = dx where the activation is positive, row layout of in_bwd_reduce_kernel with gridDim.x rows per sample -- and
  // saves a full read of dx and in.  The sums use the STORED (storage-rounded) dx, as a separate pass would.
  // LDS: first the block's logit gradients [voxel][MC] (loaded class plane by class plane, coalesced, once), then
  // reused as [vlanes][C + 1][MC] for the final reduction
  extern __shared__ float red[];
  const int n = blockIdx.y;
  const int cols = C / EPC;
  const int vlanes = 256 / cols;
  const int col = threadIdx.x % cols, vl = threadIdx.x / cols;
  const int c0 = col * EPC;
  float sc[EPC], sh[EPC], wv[MC][EPC], accw[MC][EPC], accb[MC];
  float mu[EPC], rs[EPC], s1[EPC], s2[EPC];
#pragma unroll
  for (int e = 0; e < EPC; e++) {
    sc[e] = scale ? scale[(int64_t)n * C + c0 + e] : 1.f;
    sh[e] = scale ? shift[(int64_t)n * C + c0 + e] : 0.f;
    mu[e] = inb_partials ? in_mean[(int64_t)n * C + c0 + e] : 0.f;
    rs[e] = inb_partials ? in_rstd[(int64_t)n * C + c0 + e] : 0.f;
    s1[e] = s2[e] = 0.f;
  }
#pragma unroll
  for (int o = 0; o < MC; o++) {
    accb[o] = 0.f;
#pragma unroll
    for (int e = 0; e < EPC; e++) {
      wv[o][e] = (o < ncls) ? w[min(o, ncls - 1) * C + c0 + e] : 0.f;
      accw[o][e] = 0.f;
    }
  }
  const int64_t vb = (int64_t)blockIdx.x * per, ve = min(vox, vb + per);
  if (vec4) {  // four voxels of a class plane per load (the plane rows are aligned and whole: see the launcher)
    for (int i = threadIdx.x; i < MC * (per >> 2); i += 256) {
      const int o = i / (per >> 2), lv = (i - o * (per >> 2)) * 4;
      float v[4];
      ST<T>::ld4(dlogits + ((int64_t)n * ncls + min(o, ncls - 1)) * vox + min(vb + lv, vox - 4), v);
#pragma unroll
      for (int j = 0; j < 4; j++) red[(lv + j) * MC + o] = (o < ncls && vb + lv < vox) ? v[j] : 0.f;
    }
  } else {
    for (int o = 0; o < MC; o++) {
      const T* src = dlogits + ((int64_t)n * ncls + min(o, ncls - 1)) * vox;
      for (int lv = threadIdx.x; lv < per; lv += 256) {
        const float v = ST<T>::ld(src + min(vb + lv, vox - 1));
        red[lv * MC + o] = (o < ncls && vb + lv < vox) ? v : 0.f;
      }
    }
  }
  __syncthreads();
  if (vl < vlanes) {
    constexpr int U = sizeof(T) == 2 ? 2 : 4;  // 16-bit storage: 128 registers (four rows in flight for the top level's
                                               // launch, alone on the chip: 153 vs 141 us)
    for (int64_t v0 = vb + vl; v0 < ve; v0 += (int64_t)U * vlanes) {
      float f[U][EPC], g[ACC ? U : 1][EPC], dl[U][MC];
      int64_t row[U];
#pragma unroll
      for (int u = 0; u < U; u++) {  // clamped: never branch around a load; the tail is masked below
        const int64_t v = min(v0 + (int64_t)u * vlanes, ve - 1);
        row[u] = (int64_t)n * vox + v;
        ST<T>::ld4(in + row[u] * in_pitch + c0, f[u]);
        if (ACC) ST<T>::ld4(dx + row[u] * dx_pitch + c0, g[ACC ? u : 0]);
#pragma unroll
        for (int o = 0; o < MC; o++) dl[u][o] = red[(int)(v - vb) * MC + o];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const bool live = v0 + (int64_t)u * vlanes < ve;
#pragma unroll
        for (int o = 0; o < MC; o++) dl[u][o] = live ? dl[u][o] : 0.f;
        float d[EPC];
#pragma unroll
        for (int e = 0; e < EPC; e++) {
          float x = f[u][e];
          if (scale) x = fmaxf(x * sc[e] + sh[e], 0.f);
          float t = ACC ? g[ACC ? u : 0][e] : 0.f;
#pragma unroll
          for (int o = 0; o < MC; o++) {
            t += dl[u][o] * wv[o][e];
            accw[o][e] += dl[u][o] * x;
          }
          // relu of the producing norm is handled by the IN backward of that layer (dx is d/d activation)
          d[e] = t;
        }
        if (live) ST<T>::st4(dx + row[u] * dx_pitch + c0, d[0], d[1], d[2], d[3]);
        if (inb_partials) {
#pragma unroll
          for (int e = 0; e < EPC; e++) {
            const float gg = (live && f[u][e] * sc[e] + sh[e] > 0.f) ? storage_round<T>(d[e]) : 0.f;
            s1[e] += gg;
            s2[e] += gg * ((f[u][e] - mu[e]) * rs[e]);
          }
        }
        if (col == 0) {
#pragma unroll
          for (int o = 0; o < MC; o++) accb[o] += dl[u][o];
        }
      }
    }
  }
  // ---- block reduction over the voxel lanes, then one atomic per (class, channel) and block
  __syncthreads();  // everybody is done with the logit gradients: the region is reused
  const int ld = (C + 1) * MC;
  if (vl < vlanes) {
#pragma unroll
    for (int o = 0; o < MC; o++) {
#pragma unroll
      for (int e = 0; e < EPC; e++) red[vl * ld + (c0 + e) * MC + o] = accw[o][e];
      if (col == 0) red[vl * ld + C * MC + o] = accb[o];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (C + 1) * MC; i += 256) {
    const int c = i / MC, o = i - c * MC;
    if (o >= ncls) continue;
    float t = 0.f;
    for (int k = 0; k < vlanes; k++) t += red[k * ld + i];
#ifdef HEAD_DBG_NOATOMIC  // attribution build
    if (t == 12345.678f) dw[0] = t;
#else
    if (c < C)
      atomicAdd(dw + o * C + c, t);
    else
      atomicAdd(db + o, t);
#endif
  }
  if (inb_partials) {  // same fixed-order reduction over the voxel lanes as in_bwd_reduce_kernel
    __syncthreads();
    if (vl < vlanes) {
#pragma unroll
      for (int e = 0; e < EPC; e++) {
        red[(vl * C + c0 + e) * 2 + 0] = s1[e];
        red[(vl * C + c0 + e) * 2 + 1] = s2[e];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 2; i += 256) {
      float t = 0.f;
      for (int k = 0; k < vlanes; k++) t += red[k * C * 2 + i];
      inb_partials[((int64_t)n * gridDim.x + blockIdx.x) * C * 2 + i] = t;
    }
  }
}

// ------------------------------------------------------------------------------ IN + ReLU backward
// grid (blocks, N).  thread = (voxel lane, channel chunk)
template <typename T>
__global__ __launch_bounds__(256) void in_bwd_reduce_kernel(const T* __restrict__ da, int64_t da_pitch,
                                                            const T* __restrict__ y, int64_t y_pitch,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ partials,
                                                            int blocks, int C, int64_t vox) {
  HDF_LIGHT_PRIO();
  constexpr int EPC = ST<T>::EPC;
  extern __shared__ float red[];  // [vlanes][C][2]
  const int n = blockIdx.y;
  const int cols = C / EPC;
  const int vlanes = 256 / cols;
  const int col = threadIdx.x % cols, vl = threadIdx.x / cols;
  const int c0 = col * EPC;
  float s1[EPC], s2[EPC], sc[EPC], sh[EPC], mu[EPC], rs[EPC];
#pragma unroll
  for (int e = 0; e < EPC; e++) {
    s1[e] = s2[e] = 0.f;
    int64_t o = (int64_t)n * C + c0 + e;
    sc[e] = scale[o];
    sh[e] = shift[o];
    mu[e] = mean[o];
    rs[e] = rstd[o];
  }
  if (vl < vlanes) {
    int64_t per = (vox + blocks - 1) / blocks;
    int64_t vb = (int64_t)blockIdx.x * per, ve = min(vox, vb + per);
    constexpr int U = 4;
    for (int64_t v0 = vb + vl; v0 < ve; v0 += U * vlanes) {
      float g[U][EPC], f[U][EPC];
#pragma unroll
      for (int u = 0; u < U; u++) {  // clamped (never branch around a load); the tail is masked below
        int64_t row = (int64_t)n * vox + min(v0 + (int64_t)u * vlanes, ve - 1);
        load_chunk<T>(da + row * da_pitch + c0, g[u]);
        load_chunk<T>(y + row * y_pitch + c0, f[u]);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const bool live = v0 + (int64_t)u * vlanes < ve;
#pragma unroll
        for (int e = 0; e < EPC; e++) {
          float gg = (live && f[u][e] * sc[e] + sh[e] > 0.f) ? g[u][e] : 0.f;
          s1[e] += gg;
          s2[e] += gg * ((f[u][e] - mu[e]) * rs[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < EPC; e++) {
      red[(vl * C + c0 + e) * 2 + 0] = s1[e];
      red[(vl * C + c0 + e) * 2 + 1] = s2[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * 2; i += 256) {
    float s = 0.f;
    for (int k = 0; k < vlanes; k++) s += red[k * C * 2 + i];
    partials[((int64_t)n * blocks + blockIdx.x) * C * 2 + i] = s;
  }
}

// Register-light forms of the two streaming passes for 16-bit storage: FOUR channels per thread (8-byte loads; the forms
// above take 8 with 16-byte loads and need 160 / 114 registers).  Backward runs these passes on the caller's stream NEXT
// TO a persistent weight-gradient kernel of the side stream that holds one wave per SIMD with 301-376 of its 512
// registers: the heavy forms then got one wave per SIMD or none at all (a reduce pass of 60 us took 237 us, and the
// weight gradient it was supposed to hide cost as much as it saved); these fit two to three waves into what is left.
template <typename T>
__global__ __launch_bounds__(256, 6) void in_bwd_reduce4_kernel(const T* __restrict__ da, int64_t da_pitch,
                                                                const T* __restrict__ y, int64_t y_pitch,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                float* __restrict__ partials, int blocks, int C,
                                                                int64_t vox) {
  HDF_LIGHT_PRIO();
  extern __shared__ float red[];  // [vlanes][C][2]
  const int n = blockIdx.y;
  const int cols = C >> 2, vlanes = 256 / cols;
  const int col = threadIdx.x % cols, vl = threadIdx.x / cols, c0 = col * 4;
  float s1[4], s2[4], sc[4], sh[4], mu[4], rs[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int64_t o = (int64_t)n * C + c0 + e;
    s1[e] = s2[e] = 0.f;
    sc[e] = scale[o], sh[e] = shift[o], mu[e] = mean[o], rs[e] = rstd[o];
  }
  if (vl < vlanes) {
    const int64_t per = (vox + blocks - 1) / blocks;
    const int64_t vb = (int64_t)blockIdx.x * per, ve = min(vox, vb + per);
    constexpr int U = 4;
    const T* dap = da + (int64_t)n * vox * da_pitch + c0;
    const T* yp = y + (int64_t)n * vox * y_pitch + c0;
    for (int64_t v0 = vb + vl; v0 < ve; v0 += U * vlanes) {
      float g[U][4], f[U][4];
#pragma unroll
      for (int u = 0; u < U; u++) {  // clamped (never branch around a load); the tail is masked below
        const int64_t v = min(v0 + (int64_t)u * vlanes, ve - 1);
        ST<T>::ld4(dap + v * da_pitch, g[u]);
        ST<T>::ld4(yp + v * y_pitch, f[u]);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const bool live = v0 + (int64_t)u * vlanes < ve;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float gg = (live && f[u][e] * sc[e] + sh[e] > 0.f) ? g[u][e] : 0.f;
          s1[e] += gg;
          s2[e] += gg * ((f[u][e] - mu[e]) * rs[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      red[(vl * C + c0 + e) * 2 + 0] = s1[e];
      red[(vl * C + c0 + e) * 2 + 1] = s2[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * 2; i += 256) {
    float s = 0.f;
    for (int k = 0; k < vlanes; k++) s += red[k * C * 2 + i];
    partials[((int64_t)n * blocks + blockIdx.x) * C * 2 + i] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(256, 8) void in_bwd_apply4_kernel(const T* __restrict__ da, int64_t da_pitch,
                                                               const T* __restrict__ y, int64_t y_pitch,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd,
                                                               const float* __restrict__ k1,
                                                               const float* __restrict__ ka,
                                                               const float* __restrict__ kb, T* __restrict__ dy,
                                                               int64_t dy_pitch, int C, int64_t vox) {
  HDF_LIGHT_PRIO();
  constexpr int U = 4;
  const int n = blockIdx.y;
  const int cols = C >> 2, vlanes = 256 / cols;
  const int col = threadIdx.x % cols, vl = threadIdx.x / cols, c0 = col * 4;
  if (vl >= vlanes) return;
  float sc[4], sh[4], mu[4], rs[4], c1[4], ca[4], cb[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int64_t o = (int64_t)n * C + c0 + e;
    sc[e] = scale[o], sh[e] = shift[o], mu[e] = mean[o], rs[e] = rstd[o];
    c1[e] = k1[o], ca[e] = ka[o], cb[e] = kb[o];
  }
  const int64_t per = (vox + gridDim.x - 1) / gridDim.x;
  const int64_t vb = (int64_t)blockIdx.x * per, ve = min(vox, vb + per);
  const T* dap = da + (int64_t)n * vox * da_pitch + c0;
  const T* yp = y + (int64_t)n * vox * y_pitch + c0;
  T* dyp = dy + (int64_t)n * vox * dy_pitch + c0;
  for (int64_t v0 = vb + vl; v0 < ve; v0 += (int64_t)U * vlanes) {
    float g[U][4], f[U][4];
#pragma unroll
    for (int u = 0; u < U; u++) {  // clamped: never branch around a load
      const int64_t v = min(v0 + (int64_t)u * vlanes, ve - 1);
      ST<T>::ld4(dap + v * da_pitch, g[u]);
      ST<T>::ld4(yp + v * y_pitch, f[u]);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t v = v0 + (int64_t)u * vlanes;
      if (v < ve) {
        float d[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          d[e] = in_bwd_elem(g[u][e], f[u][e], sc[e], sh[e], mu[e], rs[e], c1[e], ca[e], cb[e]);
        }
        ST<T>::st4(dyp + v * dy_pitch, d[0], d[1], d[2], d[3]);
      }
    }
  }
}

// grid (ceil(C/8), N), 256 threads = 32 block-lanes x 8 channels (with 8 lanes the 1024-row partial table of a
// 128^3 level cost 49 us of serial fp64 adds on two workgroups); fixed summation order; light on purpose (in_finalize)
__global__ __launch_bounds__(256, 4) void in_bwd_finalize_kernel(const float* __restrict__ partials, int blocks, int N,
                                                               int C, int64_t vox, const float* __restrict__ gamma,
                                                               const float* __restrict__ rstd, float* __restrict__ k1,
                                                               float* __restrict__ ka, float* __restrict__ kb,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
  HDF_LIGHT_PRIO();
  constexpr int BL = 32;
  __shared__ double red[BL][FIN_CG][2];
  const int n = blockIdx.y, cl = threadIdx.x & (FIN_CG - 1), c = blockIdx.x * FIN_CG + cl, bl = threadIdx.x / FIN_CG;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    const float* p = partials + ((int64_t)n * blocks * C + c) * 2;
    const int64_t bs = (int64_t)C * 2;
    int b = bl;
    for (; b + 3 * BL < blocks; b += 4 * BL) {  // four independent loads in flight (the rows are L2 round trips)
      const float2 v0 = *reinterpret_cast<const float2*>(p + (int64_t)b * bs);
      const float2 v1 = *reinterpret_cast<const float2*>(p + (int64_t)(b + BL) * bs);
      const float2 v2 = *reinterpret_cast<const float2*>(p + (int64_t)(b + 2 * BL) * bs);
      const float2 v3 = *reinterpret_cast<const float2*>(p + (int64_t)(b + 3 * BL) * bs);
      s1 += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
      s2 += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    }
    for (; b < blocks; b += BL) {
      const float2 v = *reinterpret_cast<const float2*>(p + (int64_t)b * bs);
      s1 += (double)v.x;
      s2 += (double)v.y;
    }
  }
  red[bl][cl][0] = s1;
  red[bl][cl][1] = s2;
  __syncthreads();
  if (bl == 0 && c < C) {
    for (int k = 1; k < BL; k++) {
      s1 += red[k][cl][0];
      s2 += red[k][cl][1];
    }
    int64_t o = (int64_t)n * C + c;
    float g = gamma ? gamma[c] : 1.f;
    k1[o] = g * rstd[o];
    ka[o] = (float)(s1 / (double)vox);
    kb[o] = (float)(s2 / (double)vox);
    if (dgamma) atomicAdd(dgamma + c, (float)s2);  // N adders per word
    if (dbeta) atomicAdd(dbeta + c, (float)s1);
  }
}

// grid (blocks, N); thread = (voxel lane, channel chunk): the 7 per-(n,channel) coefficient vectors are loaded ONCE
// per thread (re-reading them per chunk made the kernel load-issue bound: 56 scalar loads per 2 streaming loads)
template <typename T>
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(const T* __restrict__ da, int64_t da_pitch,
                                                           const T* __restrict__ y, int64_t y_pitch,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ k1, const float* __restrict__ ka,
                                                           const float* __restrict__ kb, T* __restrict__ dy,
                                                           int64_t dy_pitch, int C, int64_t vox) {
  HDF_LIGHT_PRIO();
  constexpr int EPC = ST<T>::EPC;
  constexpr int U = 4;  // chunks per thread per iteration: 8 independent 16-byte loads in flight
  const int n = blockIdx.y;
  const int cols = C / EPC;
  const int vlanes = 256 / cols;
  const int col = threadIdx.x % cols, vl = threadIdx.x / cols;
  const int c0 = col * EPC;
  if (vl >= vlanes) return;
  float sc[EPC], sh[EPC], mu[EPC], rs[EPC], c1[EPC], ca[EPC], cb[EPC];
#pragma unroll
  for (int e = 0; e < EPC; e++) {
    const int64_t o = (int64_t)n * C + c0 + e;
    sc[e] = scale[o];
    sh[e] = shift[o];
    mu[e] = mean[o];
    rs[e] = rstd[o];
    c1[e] = k1[o];
    ca[e] = ka[o];
    cb[e] = kb[o];
  }
  const int64_t per = (vox + gridDim.x - 1) / gridDim.x;
  const int64_t vb = (int64_t)blockIdx.x * per, ve = min(vox, vb + per);
  for (int64_t v0 = vb + vl; v0 < ve; v0 += (int64_t)U * vlanes) {
    float g[U][EPC], f[U][EPC];
    int64_t rowv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {  // clamped: never branch around a load
      rowv[u] = (int64_t)n * vox + min(v0 + (int64_t)u * vlanes, ve - 1);
      load_chunk<T>(da + rowv[u] * da_pitch + c0, g[u]);
      load_chunk<T>(y + rowv[u] * y_pitch + c0, f[u]);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (v0 + (int64_t)u * vlanes < ve) {
#pragma unroll
        for (int e = 0; e < EPC; e++) {
          g[u][e] = in_bwd_elem(g[u][e], f[u][e], sc[e], sh[e], mu[e], rs[e], c1[e], ca[e], cb[e]);
        }
        store_chunk<T>(dy + rowv[u] * dy_pitch + c0, g[u]);
      }
    }
  }
}

template <typename T>
__global__ void add_kernel(T* __restrict__ a, int64_t a_pitch, const T* __restrict__ b, int64_t b_pitch, int C,
                           int64_t rows, int accumulate) {
  constexpr int EPC = ST<T>::EPC;
  const int cols = C / EPC;
  int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t row = i / cols;
    int c0 = (int)(i - row * cols) * EPC;
    float f[EPC], g[EPC];
    load_chunk<T>(b + row * b_pitch + c0, g);
    if (accumulate) {
      load_chunk<T>(a + row * a_pitch + c0, f);
#pragma unroll
      for (int e = 0; e < EPC; e++) g[e] += f[e];
    }
    store_chunk<T>(a + row * a_pitch + c0, g);
  }
}

// db[c] += sum over rows; block partial in LDS then one atomic per channel per block
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_kernel(const T* __restrict__ dy, int64_t dy_pitch,
                                                        float* __restrict__ db, int C, int64_t rows) {
  constexpr int EPC = ST<T>::EPC;
  extern __shared__ float red[];  // [C]
  const int cols = C / EPC;
  const int vlanes = 256 / cols;
  const int col = threadIdx.x % cols, vl = threadIdx.x / cols;
  for (int i = threadIdx.x; i < C; i += 256) red[i] = 0.f;
  __syncthreads();
  float s[EPC];
#pragma unroll
  for (int e = 0; e < EPC; e++) s[e] = 0.f;
  if (vl < vlanes) {
    for (int64_t r = (int64_t)blockIdx.x * vlanes + vl; r < rows; r += (int64_t)gridDim.x * vlanes) {
      float f[EPC];
      load_chunk<T>(dy + r * dy_pitch + col * EPC, f);
#pragma unroll
      for (int e = 0; e < EPC; e++) s[e] += f[e];
    }
#pragma unroll
    for (int e = 0; e < EPC; e++) atomicAdd(&red[col * EPC + e], s[e]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) atomicAdd(db + i, red[i]);
}

inline unsigned grid_for(int64_t total, int block = 256) {
  return (unsigned)std::min<int64_t>(ceil_div64(total, block), MAX_BLOCKS);
}

}  // namespace

// dynamic LDS above 64 KiB has to be allowed per kernel
static int allow_big_lds(const void* kern, size_t bytes) {
  if (bytes <= 64 * 1024) return HDF_OK;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    hdf_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(e));
    return HDF_ERR_HIP;
  }
  return HDF_OK;
}

#define DISPATCH_T HDF_DISPATCH_T

int hdf_launch_nchw_to_ndhwc(int dtype, const float* x, void* out, int N, int C, int CP, int64_t vox, hipStream_t st) {
  HDF_CHECK_ARG(CP % 16 == 0 && CP >= C, "nchw_to_ndhwc: CP=%d", CP);
  DISPATCH_T(dtype, hipLaunchKernelGGL(nchw_to_ndhwc_kernel<T>, dim3(grid_for((int64_t)N * vox)), dim3(256), 0, st, x,
                                       (T*)out, N, C, CP, vox));
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

// out[c] += sum over `rows` partial rows of partials[row][c][0] (the per-channel SUM column of the conv kernels'
// InstanceNorm partial table), c < C.  grid ceil(C/8), 256 threads = 32 row lanes x 8 channels, fixed order.
__global__ __launch_bounds__(256, 6) void stat_rows_sum_kernel(const float* __restrict__ partials, int rows, int C, int CP,
                                                             float* __restrict__ out) {
  HDF_LIGHT_PRIO();
  __shared__ double red[32][FIN_CG];
  const int cl = threadIdx.x & (FIN_CG - 1), c = blockIdx.x * FIN_CG + cl, rl = threadIdx.x / FIN_CG;
  double s = 0.0;
  if (c < C) {
    const float* p = partials + (int64_t)c * 2;
    const int64_t rs = (int64_t)CP * 2;
    int r = rl;
    for (; r + 96 < rows; r += 128)
      s += ((double)p[(int64_t)r * rs] + (double)p[(int64_t)(r + 32) * rs]) +
           ((double)p[(int64_t)(r + 64) * rs] + (double)p[(int64_t)(r + 96) * rs]);
    for (; r < rows; r += 32) s += (double)p[(int64_t)r * rs];
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    for (int k = 1; k < 32; k++) s += red[k][cl];
    out[c] += (float)s;
  }
}

int hdf_launch_stat_rows_sum(const float* partials, int rows, int C, int CP, float* out, hipStream_t st) {
  hipLaunchKernelGGL(stat_rows_sum_kernel, dim3(ceil_div(C, FIN_CG)), dim3(256), 0, st, partials, rows, C, CP, out);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_in_finalize(const float* partials, int N, int tiles, int C, int CP, int64_t vox, const float* gamma,
                           const float* beta, float eps, float* mean, float* rstd, float* scale, float* shift,
                           hipStream_t st) {
  hipLaunchKernelGGL(in_finalize_kernel, dim3(ceil_div(CP, FIN_CG), N), dim3(256), 0, st, partials, tiles, C, CP, vox, gamma,
                     beta, eps, mean, rstd, scale, shift);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_norm_relu_add(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift,
                             const void* skip, int64_t skip_pitch, void* out, int64_t out_pitch, int N, int C,
                             int64_t vox, hipStream_t st) {
  HDF_CHECK_ARG(C % 16 == 0, "norm_relu_add: C=%d", C);
  DISPATCH_T(dtype, hipLaunchKernelGGL(norm_relu_add_kernel<T>, dim3(grid_for((int64_t)N * vox * (C / ST<T>::EPC))),
                                       dim3(256), 0, st, (const T*)y, y_pitch, scale, shift, (const T*)skip, skip_pitch,
                                       (T*)out, out_pitch, N, C, vox));
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_enc_tail(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift,
                        const void* skip, int64_t skip_pitch, void* ds, int64_t ds_pitch, void* pooled,
                        int64_t pooled_pitch, uint8_t* idx, int N, int C, int Do, int Ho, int Wo, hipStream_t st, int flat) {
  HDF_CHECK_ARG(C % 16 == 0, "enc_tail: C=%d", C);
  HDF_CHECK_ARG(!flat || Do == 1, "enc_tail: the 2-D form takes depth-1 tensors");
  // (a form with the two x neighbours of a pooled voxel on neighbouring lanes -- whole contiguous rows per instruction, the
  // partial maxima merged through one lane exchange -- was built and measured: 163 vs 155 us at 128^3; this form already
  // streams at 5.5 TB/s alone, the 206 us it shows inside a step come from what runs around it)
  DISPATCH_T(dtype, {
    unsigned g = grid_for((int64_t)N * Do * Ho * Wo * (C / ST<T>::EPC));
    if (flat)
      hipLaunchKernelGGL((enc_tail_kernel<T, true>), dim3(g), dim3(256), 0, st, (const T*)y, y_pitch, scale, shift,
                         (const T*)skip, skip_pitch, (T*)ds, ds_pitch, (T*)pooled, pooled_pitch, idx, N, C, Do, Ho, Wo);
    else
      hipLaunchKernelGGL((enc_tail_kernel<T>), dim3(g), dim3(256), 0, st, (const T*)y, y_pitch, scale, shift,
                         (const T*)skip, skip_pitch, (T*)ds, ds_pitch, (T*)pooled, pooled_pitch, idx, N, C, Do, Ho, Wo);
  });
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_enc_tail_up(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift,
                           const void* low, int64_t low_pitch, const float* lscale, const float* lshift, void* ds,
                           int64_t ds_pitch, void* pooled, int64_t pooled_pitch, uint8_t* idx, int N, int C, int Do, int Ho,
                           int Wo, hipStream_t st) {
  HDF_CHECK_ARG(C % 16 == 0, "enc_tail_up: C=%d", C);
  DISPATCH_T(dtype, {
    HDF_CHECK_ARG(Ho <= 65535 && (int64_t)N * Do <= 65535, "enc_tail_up: extent %d x %d x %d", Do, Ho, Wo);
    hipLaunchKernelGGL((enc_tail_up_kernel<T>), dim3(ceil_div(Wo * (C / 4), 256), Ho, N * Do), dim3(256), 0, st, (const T*)y, y_pitch, scale, shift,
                       (const T*)low, low_pitch, lscale, lshift, (T*)ds, ds_pitch, (T*)pooled, pooled_pitch, idx, N, C, Do,
                       Ho, Wo);
  });
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_maxpool_fwd(int dtype, const void* in, int64_t in_pitch, void* out, int64_t out_pitch, uint8_t* idx,
                           int N, int C, int Do, int Ho, int Wo, hipStream_t st) {
  HDF_CHECK_ARG(C % 16 == 0, "maxpool: C=%d", C);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(grid_for((int64_t)N * Do * Ho * Wo * (C / ST<T>::EPC))),
                                dim3(256), 0, st, (const T*)in, in_pitch, (T*)out, out_pitch, idx, N, C, Do, Ho, Wo));
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

// MaxPool3d(2) backward, accumulating into din, for the encoder levels: din (= the gradient of ds_k = relu(IN(y)) + at_k)
// is COMPLETE once the pooled branch's gradient has been added, so the pass that adds it also takes the first pass of that
// layer's InstanceNorm(+ReLU) backward -- per workgroup and channel (sum g, sum g * xhat) with g = the STORED din where
// relu(IN(y)) is positive, rows of in_bwd_reduce_kernel's layout with gridDim.x rows per sample -- and saves that pass its
// read of din (one of its two tensors; y is read here instead).  grid (blocks, N); thread = (pooled-voxel lane, 4 channels).
// FLAT (round 6): MaxPool2d(2) windows (1x2x2) of a depth-1 tensor, window index k = 2 dy + dx.
template <typename T, bool FLAT = false>
__global__ __launch_bounds__(256, 4) void maxpool_bwd_inb_kernel(const T* __restrict__ dout, int64_t dout_pitch,
                                                                 const uint8_t* __restrict__ idx, T* __restrict__ din,
                                                                 int64_t din_pitch, const T* __restrict__ y,
                                                                 int64_t y_pitch, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd,
                                                                 float* __restrict__ partials, int C, int Do, int Ho,
                                                                 int Wo) {
  HDF_LIGHT_PRIO();
  extern __shared__ float red[];  // [vlanes][C][2]
  const int n = blockIdx.y, blocks = gridDim.x;
  const int cols = C >> 2, vlanes = 256 / cols;
  const int col = threadIdx.x % cols, vl = threadIdx.x / cols, c0 = col * 4;
  const int Hi = 2 * Ho, Wi = 2 * Wo;
  const int pvox = Do * Ho * Wo;
  float sc[4], sh[4], mu[4], rs[4], s1[4], s2[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int64_t o = (int64_t)n * C + c0 + e;
    sc[e] = scale[o], sh[e] = shift[o], mu[e] = mean[o], rs[e] = rstd[o];
    s1[e] = s2[e] = 0.f;
  }
  if (vl < vlanes) {
    const int per = (pvox + blocks - 1) / blocks;
    const int vb = blockIdx.x * per, ve = min(pvox, vb + per);
    for (int v = vb + vl; v < ve; v += vlanes) {
      const int ow = v % Wo, oh = (v / Wo) % Ho, od = v / (Wo * Ho);
      const int64_t prow = (int64_t)n * pvox + v;
      float g[4];
      ST<T>::ld4(dout + prow * dout_pitch + c0, g);
      const uint32_t pk = *reinterpret_cast<const uint32_t*>(idx + prow * C + c0);
      constexpr int NZ = FLAT ? 1 : 2;
      const int64_t row0 = (((int64_t)n * NZ * Do + NZ * od) * Hi + 2 * oh) * Wi + 2 * ow;
      // both z planes of the 2x2x2 block: 16 loads in flight per thread (one plane at a time, 8 loads: 203 vs 177 us at 128^3)
      float f[NZ][4][4], yv[NZ][4][4];
#pragma unroll
      for (int half = 0; half < NZ; half++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int64_t irow = row0 + ((int64_t)half * Hi + (q >> 1)) * Wi + (q & 1);
          ST<T>::ld4(din + irow * din_pitch + c0, f[half][q]);
          ST<T>::ld4(y + irow * y_pitch + c0, yv[half][q]);
        }
#pragma unroll
      for (int half = 0; half < NZ; half++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int k = half * 4 + q;
          const int64_t irow = row0 + ((int64_t)half * Hi + (q >> 1)) * Wi + (q & 1);
#pragma unroll
          for (int e = 0; e < 4; e++)
            if ((int)((pk >> (8 * e)) & 255u) == k) f[half][q][e] += g[e];
          ST<T>::st4(din + irow * din_pitch + c0, f[half][q][0], f[half][q][1], f[half][q][2], f[half][q][3]);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float gg = (yv[half][q][e] * sc[e] + sh[e] > 0.f) ? storage_round<T>(f[half][q][e]) : 0.f;
            s1[e] += gg;
            s2[e] += gg * ((yv[half][q][e] - mu[e]) * rs[e]);
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      red[(vl * C + c0 + e) * 2 + 0] = s1[e];
      red[(vl * C + c0 + e) * 2 + 1] = s2[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * 2; i += 256) {
    float s = 0.f;
    for (int k = 0; k < vlanes; k++) s += red[k * C * 2 + i];
    partials[((int64_t)n * blocks + blockIdx.x) * C * 2 + i] = s;
  }
}

// rows per sample of the partials maxpool_bwd_inb_kernel writes: ~256 pooled voxels x chunk lanes per workgroup, <= 1024
int hdf_maxpool_bwd_in_blocks(int64_t pooled_vox, int C) {
  return (int)std::max<int64_t>(1, std::min<int64_t>(1024, pooled_vox * (C / 4) / 2048));
}

int hdf_launch_maxpool_bwd_in(int dtype, const void* dout, int64_t dout_pitch, const uint8_t* idx, void* din,
                              int64_t din_pitch, const void* y, int64_t y_pitch, const float* scale, const float* shift,
                              const float* mean, const float* rstd, float* partials, int N, int C, int Do, int Ho, int Wo,
                              hipStream_t st, int flat) {
  HDF_CHECK_ARG(C % 16 == 0 && C <= 1024, "maxpool_bwd_in: C=%d", C);
  HDF_CHECK_ARG((int64_t)Do * Ho * Wo < ((int64_t)1 << 28), "maxpool_bwd_in: %dx%dx%d pooled voxels per sample", Do, Ho, Wo);
  const int blocks = hdf_maxpool_bwd_in_blocks((int64_t)Do * Ho * Wo, C);
  const int vlanes = 256 / (C / 4);
  if (flat) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_bwd_inb_kernel<T, true>), dim3(blocks, N), dim3(256),
                                         (size_t)vlanes * C * 2 * sizeof(float), st, (const T*)dout, dout_pitch, idx, (T*)din,
                                         din_pitch, (const T*)y, y_pitch, scale, shift, mean, rstd, partials, C, Do, Ho, Wo));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool_bwd_inb_kernel<T>, dim3(blocks, N), dim3(256),
                                         (size_t)vlanes * C * 2 * sizeof(float), st, (const T*)dout, dout_pitch, idx, (T*)din,
                                         din_pitch, (const T*)y, y_pitch, scale, shift, mean, rstd, partials, C, Do, Ho, Wo));
  }
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_maxpool_bwd(int dtype, const void* dout, int64_t dout_pitch, const uint8_t* idx, void* din,
                           int64_t din_pitch, int N, int C, int Do, int Ho, int Wo, int accumulate, hipStream_t st) {
  DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<T>,
                                       dim3(grid_for((int64_t)N * Do * Ho * Wo * (C / ST<T>::EPC))), dim3(256), 0, st,
                                       (const T*)dout, dout_pitch, idx, (T*)din, din_pitch, N, C, Do, Ho, Wo,
                                       accumulate));
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_upsample_fwd(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift,
                            void* out, int64_t out_pitch, int N, int C, int Di, int Hi, int Wi, hipStream_t st, int flat) {
  HDF_CHECK_ARG(C % 16 == 0, "upsample: C=%d", C);
  HDF_CHECK_ARG(scale && shift, "upsample_fwd: the producer's InstanceNorm scale / shift are required");
  HDF_CHECK_ARG(!flat || Di == 1, "upsample_fwd: the 2-D form takes depth-1 tensors");
  if (flat) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((upsample_fwd_kernel<T, true>),
                                         dim3(grid_for((int64_t)N * Di * Hi * Wi * (C / ST<T>::EPC))), dim3(256), 0, st,
                                         (const T*)y, y_pitch, scale, shift, (T*)out, out_pitch, N, C, Di, Hi, Wi));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL(upsample_fwd_kernel<T>,
                                         dim3(grid_for((int64_t)N * Di * Hi * Wi * (C / ST<T>::EPC))), dim3(256), 0, st,
                                         (const T*)y, y_pitch, scale, shift, (T*)out, out_pitch, N, C, Di, Hi, Wi));
  }
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_upsample_bwd(int dtype, const void* dout, int64_t dout_pitch, void* din, int64_t din_pitch, int N,
                            int C, int Di, int Hi, int Wi, hipStream_t st, int flat) {
  HDF_CHECK_ARG(!flat || Di == 1, "upsample_bwd: the 2-D form takes depth-1 tensors");
  HDF_CHECK_ARG((int64_t)8 * Di * Hi * Wi * dout_pitch < ((int64_t)1 << 31),
                "upsample_bwd: a sample of %dx%dx%d voxels x pitch %lld exceeds 32-bit element offsets", 2 * Di, 2 * Hi,
                2 * Wi, (long long)dout_pitch);
  DISPATCH_T(dtype, {
    const int gx = ceil_div(Wi * (C / ST<T>::EPC), 256);
    const int64_t wgs = (int64_t)gx * Hi * N * Di;
    HDF_CHECK_ARG(wgs < ((int64_t)1 << 31), "upsample_bwd: %lld workgroups", (long long)wgs);
    if (flat)
      hipLaunchKernelGGL((upsample_bwd_kernel<T, true>), dim3((unsigned)wgs), dim3(256), 0, st, (const T*)dout, dout_pitch,
                         (T*)din, din_pitch, N, C, Di, Hi, Wi, gx);
    else
      hipLaunchKernelGGL(upsample_bwd_kernel<T>, dim3((unsigned)wgs), dim3(256), 0, st, (const T*)dout, dout_pitch, (T*)din,
                         din_pitch, N, C, Di, Hi, Wi, gx);
  });
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_head_fwd(int dtype, const void* in, int64_t in_pitch, const float* scale, const float* shift,
                        const float* w, const float* b, void* logits, int N, int C, int ncls, int64_t vox,
                        hipStream_t st) {
  HDF_CHECK_ARG(ncls <= HEAD_MAXCLS, "head: n_cls=%d > %d", ncls, HEAD_MAXCLS);
  HDF_CHECK_ARG(C % 16 == 0 && C <= 1024, "head_fwd: C=%d", C);
  // >= 1024 workgroups per sample where that still leaves 256 voxels each (64^3: 42 -> 32 us), else the backward's rule
  // (64-voxel workgroups at 32^3 were slower: 28 vs 18.5 us)
  const int per = head_vox(vox, 1024) >= 256 ? head_vox(vox, 1024) : head_vox(vox);
  const unsigned gx = (unsigned)ceil_div64(vox, per);
  const int vec4 = (vox % 4 == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0) ? 1 : 0;
  DISPATCH_T(dtype, {
    // 8 class slots at per = 2048 (the 128^3 level) need 72 KiB of dynamic LDS: above 64 KiB it is allowed per kernel
    auto go = [&](auto kern, int mc) -> int {
      const size_t shm = (size_t)mc * (per + 256) * sizeof(float);
      HDF_TRY(allow_big_lds((const void*)kern, shm));
      hipLaunchKernelGGL(kern, dim3(gx, N), dim3(256), shm, st, (const T*)in, in_pitch, scale, shift, w, b, (T*)logits, N,
                         C, ncls, vox, per, vec4);
      return HDF_OK;
    };
    if (ncls <= 4)
      HDF_TRY(go(head_fwd_kernel<T, 4>, 4));
    else
      HDF_TRY(go(head_fwd_kernel<T, 8>, 8));
  });
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

// workgroups per sample of head_bwd_kernel (= rows per sample of its optional InstanceNorm-backward partials)
int hdf_head_bwd_blocks(int64_t vox) { return (int)ceil_div64(vox, head_vox(vox)); }

int hdf_launch_head_bwd(int dtype, const void* dlogits, const void* in, int64_t in_pitch, const float* scale,
                        const float* shift, const float* w, void* dx, int64_t dx_pitch, int accumulate_dx, float* dw,
                        float* db, int N, int C, int ncls, int64_t vox, hipStream_t st, const float* in_mean,
                        const float* in_rstd, float* inb_partials) {
  HDF_CHECK_ARG(ncls <= HEAD_MAXCLS, "head: n_cls=%d > %d", ncls, HEAD_MAXCLS);
  HDF_CHECK_ARG(C % 16 == 0 && C <= 1024, "head_bwd: C=%d", C);  // (C / 4 <= 256 chunk lanes)
  HDF_CHECK_ARG(inb_partials == nullptr || (scale && in_mean && in_rstd), "head_bwd: IN partials need the layer's statistics");
  const int per = head_vox(vox);
  const unsigned gx = (unsigned)hdf_head_bwd_blocks(vox);
  // logit-gradient planes by 4-voxel loads: whole aligned groups (per is a multiple of 64)
  const int vec4 = (vox % 4 == 0 && (reinterpret_cast<uintptr_t>(dlogits) & 15) == 0) ? 1 : 0;
  DISPATCH_T(dtype, {
    const int cols = C / 4, vlanes = 256 / cols;
    const int mc = ncls <= 4 ? 4 : 8;
    const size_t shm =
        std::max(std::max((size_t)vlanes * (C + 1) * mc, (size_t)per * mc), (size_t)vlanes * C * 2) * sizeof(float);
    auto go = [&](auto kern) -> int {
      HDF_TRY(allow_big_lds((const void*)kern, shm));
      hipLaunchKernelGGL(kern, dim3(gx, N), dim3(256), shm, st, (const T*)dlogits, (const T*)in, in_pitch, scale, shift, w,
                         (T*)dx, dx_pitch, dw, db, N, C, ncls, vox, per, in_mean, in_rstd, inb_partials, vec4);
      return HDF_OK;
    };
    if (mc == 4)
      HDF_TRY(accumulate_dx ? go(head_bwd_kernel<T, 4, true>) : go(head_bwd_kernel<T, 4, false>));
    else
      HDF_TRY(accumulate_dx ? go(head_bwd_kernel<T, 8, true>) : go(head_bwd_kernel<T, 8, false>));
  });
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

// workgroups per sample of the IN-backward passes: ~2K 16-byte chunks each, so the low-resolution levels (few
// voxels, many channels) still fill the chip (with a voxel-only rule the 16^3 level ran on 4 workgroups: 85 us for 4 MB)
int hdf_in_bwd_blocks(int64_t vox, int C) {
  return (int)std::max<int64_t>(1, std::min<int64_t>(1024, vox * (C / 8) / 2048));
}

int hdf_launch_in_bwd_reduce(int dtype, const void* da, int64_t da_pitch, const void* y, int64_t y_pitch,
                             const float* scale, const float* shift, const float* mean, const float* rstd,
                             float* partials, int blocks, int N, int C, int64_t vox, hipStream_t st) {
  HDF_CHECK_ARG(C % 16 == 0 && C <= 1024, "in_bwd: C=%d", C);
  DISPATCH_T(dtype, {
#ifndef INB_HEAVY
    if constexpr (sizeof(T) == 2) {  // register-light form: co-runs with the side stream's weight gradients
      const int vlanes = 256 / (C / 4);
      hipLaunchKernelGGL(in_bwd_reduce4_kernel<T>, dim3(blocks, N), dim3(256), (size_t)vlanes * C * 2 * sizeof(float), st,
                         (const T*)da, da_pitch, (const T*)y, y_pitch, scale, shift, mean, rstd, partials, blocks, C, vox);
    } else
#endif
    {
      int cols = C / ST<T>::EPC;
      HDF_CHECK_ARG(cols <= 256, "in_bwd: C=%d too wide", C);
      int vlanes = 256 / cols;
      size_t shm = (size_t)vlanes * C * 2 * sizeof(float);
      hipLaunchKernelGGL(in_bwd_reduce_kernel<T>, dim3(blocks, N), dim3(256), shm, st, (const T*)da, da_pitch,
                         (const T*)y, y_pitch, scale, shift, mean, rstd, partials, blocks, C, vox);
    }
  });
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_in_bwd_finalize(const float* partials, int blocks, int N, int C, int64_t vox, const float* gamma,
                               const float* rstd, float* k1, float* ka, float* kb, float* dgamma, float* dbeta,
                               hipStream_t st) {
  hipLaunchKernelGGL(in_bwd_finalize_kernel, dim3(ceil_div(C, FIN_CG), N), dim3(256), 0, st, partials, blocks, N, C, vox, gamma,
                     rstd, k1, ka, kb, dgamma, dbeta);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_in_bwd_apply(int dtype, const void* da, int64_t da_pitch, const void* y, int64_t y_pitch,
                            const float* scale, const float* shift, const float* mean, const float* rstd,
                            const float* k1, const float* ka, const float* kb, void* dy, int64_t dy_pitch, int N, int C,
                            int64_t vox, hipStream_t st) {
  HDF_CHECK_ARG(C % 16 == 0 && C <= 1024, "in_bwd_apply: C=%d", C);
  // ~1K chunks per workgroup, at most 2048 workgroups per sample
  const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, vox * (C / 8) / 1024));
  DISPATCH_T(dtype, {
#ifndef INB_HEAVY
    if constexpr (sizeof(T) == 2)
      hipLaunchKernelGGL(in_bwd_apply4_kernel<T>, dim3(blocks, N), dim3(256), 0, st, (const T*)da, da_pitch, (const T*)y,
                         y_pitch, scale, shift, mean, rstd, k1, ka, kb, (T*)dy, dy_pitch, C, vox);
    else
#endif
      hipLaunchKernelGGL(in_bwd_apply_kernel<T>, dim3(blocks, N), dim3(256), 0, st, (const T*)da, da_pitch, (const T*)y,
                         y_pitch, scale, shift, mean, rstd, k1, ka, kb, (T*)dy, dy_pitch, C, vox);
  });
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_add(int dtype, void* a, int64_t a_pitch, const void* b, int64_t b_pitch, int N, int C, int64_t vox,
                   int accumulate, hipStream_t st) {
  DISPATCH_T(dtype, hipLaunchKernelGGL(add_kernel<T>, dim3(grid_for((int64_t)N * vox * (C / ST<T>::EPC))), dim3(256), 0,
                                       st, (T*)a, a_pitch, (const T*)b, b_pitch, C, (int64_t)N * vox, accumulate));
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_bias_grad(int dtype, const void* dy, int64_t dy_pitch, float* db, int C, int64_t nvox, hipStream_t st) {
  HDF_CHECK_ARG(C % 16 == 0 && C <= 1024, "bias_grad: C=%d", C);
  DISPATCH_T(dtype, {
    int cols = C / ST<T>::EPC;
    HDF_CHECK_ARG(cols <= 256, "bias_grad: C=%d too wide", C);
    int vlanes = 256 / cols;
    unsigned g = (unsigned)std::min<int64_t>(ceil_div64(nvox, (int64_t)vlanes * 8), 512);
    hipLaunchKernelGGL(bias_grad_kernel<T>, dim3(g), dim3(256), C * sizeof(float), st, (const T*)dy, dy_pitch, db, C,
                       nvox);
  });
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}
