// Fused token kernels of the multi-path dense Transformer branch on v_mfma_f32_16x16x4_f32 (exact fp32).
//
// Reference: models/HDenseFormer.py:33-145.  Everything of a dense layer except the attention core is local to a token,
// so between two attention launches ONE kernel runs, for a tile of 16 tokens (= the 16 rows of the MFMA tile):
//     POST(l-1): h1 = to_out(ob)*mask + h0 ; h2 = ff(LN2(h1)) + h1 ; feature(l-1) = ff(LN2(h2))        (:96-98)
//     OUT(b):    block out_layer DenseForward(DM+128 -> 64 -> DM) at a block boundary                     (:99-100)
//     PRE(l):    h0 = Linear(cat(features)) ; qkv = to_qkv(LN1(h0))                                       (:95, :66)
// (forward: 49 token launches + 24 attention launches per step instead of 24 x 3 + 6; backward likewise).
//
// GEMM form.  C[16 tokens][16 outputs] += A[16][K] * W[16 outputs][K]^T on v_mfma_f32_16x16x4_f32: lane (i = lane & 15,
// g = lane >> 4) supplies A[i][k] and B[k][i] for ONE k per step.  The contraction order is free, so lane group g owns
// the k range [g*K/4, (g+1)*K/4): a lane's B operands are then a CONTIGUOUS quarter of one row of the torch Linear
// weight [out][in] -- float4 loads straight from global memory into registers, every weight element read once per
// workgroup, no LDS staging -- and its A operands a contiguous quarter of one token row in LDS (ds_read_b128; row
// pitch = 4 mod 64 words keeps the 16 rows of a quarter-wave on distinct bank groups).  All weights of the POST and
// PRE stages are requested at kernel entry (<= 92 registers), so a launch pays one memory round trip, not one per stage.
// C/D layout: lane holds C[4g + r][i], r = 0..3: bias, dropout mask, GELU and residuals are applied in registers.
#include "transformer.h"

namespace {

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
__device__ __forceinline__ void ln32(const float* sIn, float* sOut, const float* __restrict__ gam,
                                     const float* __restrict__ bet) {
  const int row = threadIdx.x >> 4, c = (threadIdx.x & 15) * 2;
  const float v0 = sIn[row * LD32 + c], v1 = sIn[row * LD32 + c + 1];
  float s = v0 + v1;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s * (1.f / 32.f);
  const float d0 = v0 - mean, d1 = v1 - mean;
  float q = d0 * d0 + d1 * d1;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q * (1.f / 32.f) + 1e-5f);
  sOut[row * LD32 + c] = d0 * rstd * gam[c] + bet[c];
  sOut[row * LD32 + c + 1] = d1 * rstd * gam[c + 1] + bet[c + 1];
}

struct TokFwd {
  TfDims d;
  // POST: layer lp of block bp
  TfLayerP pp;
  int bp, lp;
  const float* h0_in;
  const float* ob;
  float* h1s;
  float* h2s;
  float* Fpost;  // block bp's feature buffer (also the OUT stage's input)
  // OUT: block bp's out_layer
  TfOutP po;
  float* next_F;   // next block's feature buffer (columns [0, DM)) or null
  void* attnall;   // or the channels-last attnall tensor (last block)
  // PRE: layer lq of block bq on feature buffer Fpre
  TfLayerP pq;
  int bq, lq;
  const float* Fpre;
  float* h0_out;
  float* qkv;
};

template <bool POST, bool OUT, bool PRE, typename T>
__global__ __launch_bounds__(256) void tok_fwd_kernel(TokFwd a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const TfDims& d = a.d;
  const int DM = d.DM, DMF = d.DMF, ldF = DMF + 4;
  float* s_F = sm;                       // [16][ldF] feature rows of the tile
  float* s_x = s_F + TT * ldF;           // [16][36]
  float* s_h = s_x + TT * LD32;          // [16][36]
  float* s_z = s_h + TT * LD32;          // [16][68]
  float* s_red = s_z + TT * LD64;        // [2][16][16] partial tiles of the K-split W0 product
  const int m = blockIdx.y, BN = d.B * d.N, t0 = blockIdx.x * TT;
  const int64_t mo = (int64_t)m * d.mstride, rb = (int64_t)m * BN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 15, g = lane >> 4;
  const DropF dr{d.training, d.seed, d.thresh24, d.keep_scale};
  auto tok = [&](int row) { return min(t0 + row, BN - 1); };  // clamped token of a tile row (loads)

  // ------------------------------------------------------------------ requests issued at kernel entry
  const int Kq = PRE ? (OUT ? DM : DM + 32 * a.lq) : 0;
  WFrag<2> f_wo, f_w1;     // to_out [32][32] / ff net.0 [64][32]: K = 32 -> 8 floats per lane
  WFrag<4> f_w2;           // ff net.3 [32][64]: 16 floats per lane
  WFrag<11> f_w0;          // Linear0 [32][Kq], split in two k halves over wave pairs: <= 44 floats per lane (Kq <= 352)
  WFrag<2> f_q0, f_q1;     // to_qkv [96][32]: tiles wave and wave + 4
  float h0r[4];
  if (POST) {
    if (wave < 2) {
      wload(f_wo, a.pp.wout + mo, 32, 16 * wave, 8, 0, 2);
      wload(f_w2, a.pp.w2 + mo, 64, 16 * wave, 16, 0, 4);
#pragma unroll
      for (int r = 0; r < 4; r++) h0r[r] = a.h0_in[(rb + tok(4 * g + r)) * 32 + 16 * wave + col];
    }
    wload(f_w1, a.pp.w1 + mo, 32, 16 * wave, 8, 0, 2);
  }
  if (PRE) {
    wload(f_w0, a.pq.w0 + mo, Kq, 16 * (wave & 1), Kq >> 2, (wave >> 1) * (Kq >> 3), Kq >> 5);
    wload(f_q0, a.pq.wqkv + mo, 32, 16 * wave, 8, 0, 2);
    if (wave < 2) wload(f_q1, a.pq.wqkv + mo, 32, 16 * (wave + 4), 8, 0, 2);
  }
  // token rows -> LDS.  POST: ob; the feature rows the later stages contract over (the columns written by this kernel
  // are filled in from registers)
  if (POST) {
    for (int i = threadIdx.x; i < TT * 8; i += 256) {
      const int row = i >> 3, c4 = (i & 7) * 4;
      *reinterpret_cast<float4*>(s_x + row * LD32 + c4) =
          *reinterpret_cast<const float4*>(a.ob + (rb + tok(row)) * 32 + c4);
    }
  }
  {
    // columns of the feature buffer needed from memory: OUT reads all of block bp's buffer but the feature computed
    // here; PRE without OUT reads [0, Kq) of the same buffer (minus that feature); PRE alone reads [0, Kq) of Fpre
    const float* Fsrc = (POST || OUT) ? a.Fpost : a.Fpre;
    const int ncol = OUT ? DMF : Kq;
    const int skip0 = POST ? DM + 32 * a.lp : ncol;  // [skip0, skip0 + 32) comes from this kernel
    if (ncol > 0) {
      const int c4n = ncol >> 2;
      for (int i = threadIdx.x; i < TT * c4n; i += 256) {
        const int row = i / c4n, c4 = (i - row * c4n) * 4;
        if (c4 < skip0 || c4 >= skip0 + 32)
          *reinterpret_cast<float4*>(s_F + row * ldF + c4) =
              *reinterpret_cast<const float4*>(Fsrc + (rb + tok(row)) * DMF + c4);
      }
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ POST(lp)
  if (POST) {
    const uint32_t site0 = hdf_site_id(m, a.bp, a.lp, 0);
    float h1r[4];
    if (wave < 2) {  // to_out + dropout + residual
      f32x4 acc = zero4();
      wmma(acc, f_wo, s_x, LD32, 8, 0, 2);
      const int c = 16 * wave + col;
      const float bo = a.pp.bout[mo + c];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 4 * g + r, t = t0 + row;
        const float v = (acc[r] + bo) * dr.mask(site0 + 0, (uint32_t)t * 32 + c) + h0r[r];
        h1r[r] = v;
        s_h[row * LD32 + c] = v;
        if (t < BN) a.h1s[(rb + t) * 32 + c] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {  // pass 0: h2 = ff(LN2(h1)) + h1 ; pass 1: feature = ff(LN2(h2))
      ln32(s_h, s_x, a.pp.ln2g + mo, a.pp.ln2b + mo);
      __syncthreads();
      {
        f32x4 acc = zero4();
        wmma(acc, f_w1, s_x, LD32, 8, 0, 2);
        const int c = 16 * wave + col;
        const float b1 = a.pp.b1[mo + c];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 4 * g + r, t = t0 + row;
          s_z[row * LD64 + c] = gelu_f(acc[r] + b1) * dr.mask(site0 + 1 + 2 * pass, (uint32_t)t * 64 + c);
        }
      }
      __syncthreads();
      if (wave < 2) {
        f32x4 acc = zero4();
        wmma(acc, f_w2, s_z, LD64, 16, 0, 4);
        const int c = 16 * wave + col;
        const float b2 = a.pp.b2[mo + c];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 4 * g + r, t = t0 + row;
          const float gv = (acc[r] + b2) * dr.mask(site0 + 2 + 2 * pass, (uint32_t)t * 32 + c);
          if (pass == 0) {
            const float h2 = gv + h1r[r];
            s_h[row * LD32 + c] = h2;
            if (t < BN) a.h2s[(rb + t) * 32 + c] = h2;
          } else {
            const int fc = DM + 32 * a.lp + c;
            s_F[row * ldF + fc] = gv;
            if (t < BN) a.Fpost[(rb + t) * DMF + fc] = gv;
          }
        }
      }
      __syncthreads();
    }
  }

  // ------------------------------------------------------------------ OUT(bp): DenseForward(DM+128 -> 64 -> DM)
  if (OUT) {
    const uint32_t siteo = hdf_site_id(m, a.bp, 4, 0);
    {
      f32x4 acc = zero4();
      wmma_stream(acc, a.po.wa + mo, DMF, 16 * wave, s_F, ldF, DMF >> 2);
      const int c = 16 * wave + col;
      const float ba = a.po.ba[mo + c];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 4 * g + r, t = t0 + row;
        s_z[row * LD64 + c] = gelu_f(acc[r] + ba) * dr.mask(siteo + 0, (uint32_t)t * 64 + c);
      }
    }
    __syncthreads();  // every wave is done reading s_F: its first DM columns become the next block's input
    for (int n0 = 16 * wave; n0 < DM; n0 += 64) {
      f32x4 acc = zero4();
      wmma_stream(acc, a.po.wb + mo, 64, n0, s_z, LD64, 16);
      const int c = n0 + col;
      const float bb = a.po.bb[mo + c];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 4 * g + r, t = t0 + row;
        const float v = (acc[r] + bb) * dr.mask(siteo + 1, (uint32_t)t * DM + c);
        if (a.next_F) {
          s_F[row * ldF + c] = v;
          if (t < BN) a.next_F[(rb + t) * DMF + c] = v;
        } else if (t < BN) {
          const int b = t / d.N, n = t - b * d.N;
          ST<T>::st(reinterpret_cast<T*>(a.attnall) + ((int64_t)b * d.N + n) * ((int64_t)d.M * DM) + (int64_t)m * DM + c, v);
        }
      }
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ PRE(lq): Linear0 + LN1 + to_qkv
  if (PRE) {
    {
      f32x4 acc = zero4();
      wmma(acc, f_w0, s_F, ldF, Kq >> 2, (wave >> 1) * (Kq >> 3), Kq >> 5);
      if (wave >= 2) {
#pragma unroll
        for (int r = 0; r < 4; r++) s_red[((wave & 1) * 16 + 4 * g + r) * 16 + col] = acc[r];
      }
      __syncthreads();
      if (wave < 2) {
        const int c = 16 * wave + col;
        const float b0 = a.pq.b0[mo + c];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 4 * g + r, t = t0 + row;
          const float h = acc[r] + s_red[(wave * 16 + row) * 16 + col] + b0;
          s_h[row * LD32 + c] = h;
          if (t < BN) a.h0_out[(rb + t) * 32 + c] = h;
        }
      }
    }
    __syncthreads();
    ln32(s_h, s_x, a.pq.ln1g + mo, a.pq.ln1b + mo);
    __syncthreads();
    {
      f32x4 acc = zero4();
      wmma(acc, f_q0, s_x, LD32, 8, 0, 2);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int t = t0 + 4 * g + r;
        if (t < BN) a.qkv[(rb + t) * 96 + 16 * wave + col] = acc[r];
      }
      if (wave < 2) {
        f32x4 acc2 = zero4();
        wmma(acc2, f_q1, s_x, LD32, 8, 0, 2);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int t = t0 + 4 * g + r;
          if (t < BN) a.qkv[(rb + t) * 96 + 16 * (wave + 4) + col] = acc2[r];
        }
      }
    }
  }
}

template <typename Kern>
int allow_lds_f(Kern kern, size_t bytes) {
  if (bytes <= 64 * 1024) return HDF_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)LDS_LIMIT_F);
  if (e != hipSuccess) {
    hdf_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(e));
    return HDF_ERR_HIP;
  }
  return HDF_OK;
}

template <bool POST, bool OUT, bool PRE, typename T>
int launch_tok_fwd(const TokFwd& a, hipStream_t st) {
  const TfDims& d = a.d;
  dim3 grid(ceil_div(d.B * d.N, TT), d.M);
  const size_t shm = (size_t)(TT * (d.DMF + 4) + 2 * TT * LD32 + TT * LD64 + 2 * 16 * 16) * sizeof(float);
  HDF_TRY(allow_lds_f(tok_fwd_kernel<POST, OUT, PRE, T>, shm));
  hipLaunchKernelGGL((tok_fwd_kernel<POST, OUT, PRE, T>), grid, dim3(256), shm, st, a);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

}  // namespace

// post: finish dense layer (bp, lp) [null layer pointers: none]; out: block bp's out_layer; pre: start layer (bq, lq)
int tf_token_fwd(const TfDims& d, const TfTokenFwd& t, int dtype, hipStream_t st) {
  HDF_CHECK_ARG(d.DM % 32 == 0 && d.DM >= 32 && d.DM <= 256, "token kernel: token dim %d unsupported", d.DM);
  TokFwd a{};
  a.d = d;
  if (t.post) {
    a.pp = *t.post, a.bp = t.bp, a.lp = t.lp;
    a.h0_in = t.post_save.h0, a.ob = t.post_save.ob, a.h1s = t.post_save.h1, a.h2s = t.post_save.h2;
  }
  a.Fpost = t.F_post;
  if (t.out) {
    a.po = *t.out, a.bp = t.bp;
    a.next_F = t.next_F, a.attnall = t.attnall;
  }
  if (t.pre) {
    a.pq = *t.pre, a.bq = t.bq, a.lq = t.lq;
    a.Fpre = t.out ? t.next_F : (t.post ? t.F_post : t.F_pre);
    a.h0_out = t.pre_save.h0, a.qkv = t.pre_save.qkv;
  }
  const bool P = t.post != nullptr, O = t.out != nullptr, Q = t.pre != nullptr;
  HDF_CHECK_ARG(!(O && Q) || t.next_F, "token kernel: OUT + PRE needs the next block's feature buffer");
  HDF_CHECK_ARG(!Q || O || !P || (t.bq == t.bp && t.lq == t.lp + 1), "token kernel: POST + PRE must be consecutive layers");
#define TOK_CASE(PP, OO, QQ)                                                   \
  if (P == PP && O == OO && Q == QQ) {                                         \
    if constexpr (OO && !QQ) {                                                 \
      if (!t.next_F) HDF_DISPATCH_T(dtype, return (launch_tok_fwd<PP, OO, QQ, T>(a, st))); \
    }                                                                          \
    return launch_tok_fwd<PP, OO, QQ, float>(a, st);                           \
  }
  TOK_CASE(false, false, true)
  TOK_CASE(true, false, false)
  TOK_CASE(true, false, true)
  TOK_CASE(true, true, true)
  TOK_CASE(true, true, false)
  TOK_CASE(false, true, false)
  TOK_CASE(false, true, true)
#undef TOK_CASE
  hdf_set_error("token kernel: empty stage selection");
  return HDF_ERR_ARG;
}
