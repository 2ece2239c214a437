// Token-tile helpers shared by the fused token kernels (transformer_fused.hip) and the persistent chain kernels
// (transformer_chain.hip): GEMM fragments on v_mfma_f32_16x16x4_f32, LayerNorm(32) forward / backward pieces, tapes.
// See the head of transformer_fused.hip for the GEMM form.  Thread maps assume the FIRST 256 threads of a workgroup.
#pragma once
#include "transformer.h"

namespace tftok {

constexpr int TT = 16;  // tokens per workgroup
constexpr int LD32 = 36, LD64 = 68;  // LDS row pitches (floats) of the 32- and 64-wide token tiles
constexpr size_t LDS_LIMIT_F = 160 * 1024;

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

struct DropF {
  int training;
  uint32_t seed, thresh;
  float scale;
  __device__ __forceinline__ float mask(uint32_t site, uint32_t idx) const {
    if (!training) return 1.f;
    return hdf_keep(hdf_site_key(seed, site), idx, thresh) ? scale : 0.f;
  }
};

__device__ __forceinline__ f32x4 zero4() {
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return z;
}

// per-lane weight fragment: NF4 float4 = this lane's (sub-)quarter of weight row n (lane & 15) of a 16-row tile
template <int NF4>
struct WFrag {
  float4 v[NF4];
};
// rows [n0, n0+16) of W[.][ldw]; lane group g owns columns [g*KQ + s0, g*KQ + s0 + 4*n4)
template <int NF4>
__device__ __forceinline__ void wload(WFrag<NF4>& f, const float* __restrict__ W, int ldw, int n0, int KQ, int s0, int n4) {
  const int lane = threadIdx.x & 63;
  const float4* p = reinterpret_cast<const float4*>(W + (int64_t)(n0 + (lane & 15)) * ldw + (lane >> 4) * KQ + s0);
#pragma unroll
  for (int j = 0; j < NF4; j++) f.v[j] = p[j < n4 ? j : 0];  // clamped: never branch around a load
}
// acc += A[16][.] * W^T over this lane group's columns; sA row pitch lda
template <int NF4>
__device__ __forceinline__ void wmma(f32x4& acc, const WFrag<NF4>& f, const float* sA, int lda, int KQ, int s0, int n4) {
  const int lane = threadIdx.x & 63;
  const float4* pa = reinterpret_cast<const float4*>(sA + (lane & 15) * lda + (lane >> 4) * KQ + s0);
#pragma unroll
  for (int j = 0; j < NF4; j++) {
    if (j < n4) {
      const float4 a = pa[j];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, f.v[j].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, f.v[j].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, f.v[j].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, f.v[j].w, acc, 0, 0, 0);
    }
  }
}
// streaming form for the wide block out-layer GEMMs: weights fetched in chunks of 4 float4 per lane, the next chunk in
// flight under the 16 MFMAs of the current one
__device__ __forceinline__ void wmma_stream(f32x4& acc, const float* __restrict__ W, int ldw, int n0, const float* sA,
                                            int lda, int KQ) {
  const int lane = threadIdx.x & 63;
  const float4* pw = reinterpret_cast<const float4*>(W + (int64_t)(n0 + (lane & 15)) * ldw + (lane >> 4) * KQ);
  const float4* pa = reinterpret_cast<const float4*>(sA + (lane & 15) * lda + (lane >> 4) * KQ);
  const int n4 = KQ >> 2;
  float4 w[4], wn[4];
#pragma unroll
  for (int j = 0; j < 4; j++) w[j] = pw[min(j, n4 - 1)];
  for (int c = 0; c < n4; c += 4) {
#pragma unroll
    for (int j = 0; j < 4; j++) wn[j] = pw[min(c + 4 + j, n4 - 1)];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (c + j < n4) {
        const float4 a = pa[c + j];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[j].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[j].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[j].w, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = wn[j];
  }
}

// LayerNorm(32) of the 16 token rows of sIn (pitch LD32) -> sOut; 16 lanes per token, 2 columns per lane
// small per-column parameters of a stage, requested at kernel entry (a load at its point of use would expose one
// memory round trip per stage: the stages themselves are a few hundred cycles long)
struct LnP {
  float g0, g1, b0, b1;
};
__device__ __forceinline__ LnP ln_load(const float* __restrict__ gam, const float* __restrict__ bet) {
  const int c = (threadIdx.x & 15) * 2;
  return LnP{gam[c], gam[c + 1], bet[c], bet[c + 1]};
}
// (every sum of products in these helpers is an explicit fma under contract(off): the token kernels of
// transformer_fused.hip and the persistent kernels of transformer_chain.hip must round identically, and left to the
// compiler `d0 * d0 + d1 * d1` was an fma in one file and two multiplies + an add -- SLP-vectorised -- in the other)
__device__ __forceinline__ void ln32(const float* sIn, float* sOut, const LnP& p) {
#pragma clang fp contract(off)
  const int row = threadIdx.x >> 4, c = (threadIdx.x & 15) * 2;
  const float v0 = sIn[row * LD32 + c], v1 = sIn[row * LD32 + c + 1];
  float s = v0 + v1;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s * (1.f / 32.f);
  const float d0 = v0 - mean, d1 = v1 - mean;
  float q = __builtin_fmaf(d1, d1, d0 * d0);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(__builtin_fmaf(q, 1.f / 32.f, 1e-5f));
  sOut[row * LD32 + c] = __builtin_fmaf(d0 * rstd, p.g0, p.b0);
  sOut[row * LD32 + c + 1] = __builtin_fmaf(d1 * rstd, p.g1, p.b1);
}

__device__ __forceinline__ float gelu_grad_f(float x) {
#pragma clang fp contract(off)
  return __builtin_fmaf(x * 0.3989422804014327f, __expf(-0.5f * x * x), 0.5f * (1.f + erff(x * 0.70710678118654752f)));
}

// Data-gradient GEMMs contract over the OUTPUT index of a Linear: dX[16][I] = dY[16][O] * W[O][I].  The B operand of
// lane (n = lane & 15, g = lane >> 4) is then W[o][n0 + n] for the o range of lane group g -- one scalar per step,
// 16 lanes reading 64 contiguous bytes.  NS scalars per lane, requested at kernel entry like the row fragments.
template <int NS>
struct CFrag {
  float v[NS];
};
// rows [g*OQ + s0, .. + ns) of W[.][ldw], column n0 + (lane & 15) (clamped to ncols - 1)
template <int NS>
__device__ __forceinline__ void cload(CFrag<NS>& f, const float* __restrict__ W, int ldw, int n0, int ncols, int OQ,
                                      int s0, int ns) {
  const int lane = threadIdx.x & 63;
  const float* p = W + (int64_t)((lane >> 4) * OQ + s0) * ldw + min(n0 + (lane & 15), ncols - 1);
#pragma unroll
  for (int s = 0; s < NS; s++) f.v[s] = p[(int64_t)(s < ns ? s : 0) * ldw];
}
// acc += dY[16][.] * W over this lane group's o range; sA = dY tile in LDS (pitch lda), ns % 4 == 0
template <int NS>
__device__ __forceinline__ void cmma(f32x4& acc, const CFrag<NS>& f, const float* sA, int lda, int OQ, int s0, int ns) {
  const int lane = threadIdx.x & 63;
  const float4* pa = reinterpret_cast<const float4*>(sA + (lane & 15) * lda + (lane >> 4) * OQ + s0);
#pragma unroll
  for (int j = 0; j < NS / 4; j++) {
    if (4 * j < ns) {
      const float4 a = pa[j];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, f.v[4 * j + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, f.v[4 * j + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, f.v[4 * j + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, f.v[4 * j + 3], acc, 0, 0, 0);
    }
  }
}
// the same with the weights fetched on the spot (block out-layer: wide, six launches per step)
__device__ __forceinline__ void cmma_stream(f32x4& acc, const float* __restrict__ W, int ldw, int n0, int ncols,
                                            const float* sA, int lda, int OQ) {
  const int lane = threadIdx.x & 63;
  const float* pw = W + (int64_t)((lane >> 4) * OQ) * ldw + min(n0 + (lane & 15), ncols - 1);
  const float4* pa = reinterpret_cast<const float4*>(sA + (lane & 15) * lda + (lane >> 4) * OQ);
  for (int c = 0; c < OQ; c += 8) {
    float w[8];
#pragma unroll
    for (int s = 0; s < 8; s++) w[s] = pw[(int64_t)min(c + s, OQ - 1) * ldw];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      if (c + 4 * j < OQ) {
        const float4 a = pa[(c >> 2) + j];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[4 * j + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[4 * j + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[4 * j + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[4 * j + 3], acc, 0, 0, 0);
      }
    }
  }
}
// Weight gradient of a Linear: gW[o][i] += sum over the tile's 16 tokens of dY[t][o] * X[t][i], all [O/16] x [I/16]
// tiles of it spread over the 4 waves.  The contraction runs over tokens (t = 4g + s at step s), both operands come
// from LDS rows (conflict-free: every pitch here is 4 mod 64 words or 36), the 16 x 16 result goes out with fp32 atomics
// (lane: 4 rows x 1 column, 16 lanes = 64 contiguous bytes).
// attribution builds: -DTF_DBG_NOATOMIC drops the weight-gradient atomics, -DTF_DBG_NOWGRAD the products too (measured on
// the inner backward kernel at N = 512: 31 us -> 25 us -> 19 us)
__device__ __forceinline__ void wgrad_emit(float* __restrict__ gW, int I, int o0, int i0, const f32x4& acc) {
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; r++) {
#ifdef TF_DBG_NOATOMIC
    if (acc[r] == 12345.678f) gW[0] = acc[r];   // keeps the products alive without the atomics
#else
    atomicAdd(gW + (int64_t)(o0 + 4 * g + r) * I + i0 + n, acc[r]);
#endif
  }
}
__device__ __forceinline__ void wgrad_tiles(float* __restrict__ gW, int O, int I, const float* sY, int ldy,
                                            const float* sX, int ldx, int wrot = 0) {
#ifdef TF_DBG_NOWGRAD
  return;
#endif
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  const int ti = I >> 4, nt = (O >> 4) * ti;
  // two tiles per trip: their 4-step MFMA chains (40 cycles of dependent latency per step) interleave
  for (int tile = (wave + wrot) & 3; tile < nt; tile += 8) {
    const int tile2 = tile + 4;
    const bool two = tile2 < nt;
    const int to = tile / ti, o0 = to * 16, i0 = (tile - to * ti) * 16;
    const int tq = two ? tile2 / ti : to, o1 = tq * 16, i1 = two ? (tile2 - tq * ti) * 16 : i0;
    const float* py = sY + 4 * g * ldy + o0 + n;
    const float* px = sX + 4 * g * ldx + i0 + n;
    const float* qy = sY + 4 * g * ldy + o1 + n;
    const float* qx = sX + 4 * g * ldx + i1 + n;
    float ya[4], xa[4], yb[4], xb[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) ya[s2] = py[s2 * ldy], xa[s2] = px[s2 * ldx], yb[s2] = qy[s2 * ldy], xb[s2] = qx[s2 * ldx];
    f32x4 acc = zero4(), acc2 = zero4();
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[s2], xa[s2], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(yb[s2], xb[s2], acc2, 0, 0, 0);
    }
    wgrad_emit(gW, I, o0, i0, acc);
    if (two) wgrad_emit(gW, I, o1, i1, acc2);
  }
}
// gb[c] += sum over the 16 token rows of s[row][c]  (threads [tbase, tbase + width))
__device__ __forceinline__ void colsum_atomic(float* __restrict__ gb, int width, const float* sv, int ld, int tbase) {
  const int c = (int)threadIdx.x - tbase;
  if (c >= 0 && c < width) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < TT; r++) a += sv[r * ld + c];
    atomicAdd(gb + c, a);
  }
}
// LayerNorm(32) forward pieces kept for its backward: u = xh*gamma + beta -> sU ; xh -> sXh ; rstd -> return value
__device__ __forceinline__ float ln32_keep(const float* sIn, float* sU, float* sXh, const LnP& p) {
#pragma clang fp contract(off)
  const int row = threadIdx.x >> 4, c = (threadIdx.x & 15) * 2;
  const float v0 = sIn[row * LD32 + c], v1 = sIn[row * LD32 + c + 1];
  float s = v0 + v1;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s * (1.f / 32.f);
  const float d0 = v0 - mean, d1 = v1 - mean;
  float q = __builtin_fmaf(d1, d1, d0 * d0);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(__builtin_fmaf(q, 1.f / 32.f, 1e-5f));
  sXh[row * LD32 + c] = d0 * rstd, sXh[row * LD32 + c + 1] = d1 * rstd;
  sU[row * LD32 + c] = __builtin_fmaf(d0 * rstd, p.g0, p.b0);
  sU[row * LD32 + c + 1] = __builtin_fmaf(d1 * rstd, p.g1, p.b1);
  return rstd;
}
// LayerNorm backward of this thread's two columns: du (gradient w.r.t. the LN output) -> dh; the products du*xh are left
// in sGx (for the gamma gradient column sums; du itself stays in sDu for beta)
__device__ __forceinline__ void ln32_bwd(const float* sDu, const float* sXh, float* sGx, float rstd, const LnP& p,
                                         float& dh0, float& dh1) {
#pragma clang fp contract(off)
  const int row = threadIdx.x >> 4, c = (threadIdx.x & 15) * 2;
  const float u0 = sDu[row * LD32 + c], u1 = sDu[row * LD32 + c + 1];
  const float x0 = sXh[row * LD32 + c], x1 = sXh[row * LD32 + c + 1];
  sGx[row * LD32 + c] = u0 * x0, sGx[row * LD32 + c + 1] = u1 * x1;
  const float a0 = u0 * p.g0, a1 = u1 * p.g1;
  float s1 = a0 + a1, s2 = __builtin_fmaf(a1, x1, a0 * x0);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64), s2 += __shfl_xor(s2, o, 64);
  const float m1 = s1 * (1.f / 32.f), m2 = s2 * (1.f / 32.f);
  dh0 = rstd * __builtin_fmaf(-x0, m2, a0 - m1);
  dh1 = rstd * __builtin_fmaf(-x1, m2, a1 - m1);
}

// 16-token LDS tile [16][width] -> the tile's rows of tape segment `col0`: a tape is segment-major, segment c of width w
// is the dense array [rows][w] at float offset rows * c (tf_wgrad then streams whole contiguous rows of a segment; with
// token-major 576-float rows its 128-byte column slices were 2.3 KB apart and the launch took 358 us)
__device__ __forceinline__ void tape_store(float* __restrict__ tape, int64_t rows, int col0, const float* sT, int lds_ld,
                                           int width, int64_t row0, int nvalid) {
  const int w4 = width >> 2;
  float* seg = tape + rows * col0 + row0 * width;
  for (int i = threadIdx.x; i < TT * w4; i += 256) {
    const int row = i / w4, c4 = (i - row * w4) * 4;
    if (row < nvalid) *reinterpret_cast<float4*>(seg + i * 4) = *reinterpret_cast<const float4*>(sT + row * lds_ld + c4);
  }
}

}  // namespace tftok
