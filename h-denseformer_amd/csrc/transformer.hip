// Multi-path densely-connected Transformer branch: forward and backward kernels (fp32).
//
// Reference: models/HDenseFormer.py:33-145.  Per dense layer l of block b (DensePreConv_AttentionBlock
// .forward, :91-101):   h0 = Linear(cat(features));  h1 = attn(LN1(h0)) + h0;  h2 = ff(LN2(h1)) + h1;
// feature = ff(LN2(h2))  (second evaluation of the same ff, :98).  The feature concat is never
// materialised: every block owns a dense buffer F[rows][DM+128] whose column ranges ARE the features.
//
// Work decomposition: token-parallel kernels give one token to each 32-lane half-wave (the layer
// width is 32 = growth_rate), 32 tokens per workgroup; attention is a flash-style pass with K,V of
// one (sample, modality, head) in LDS (N x 4 floats each), 4 lanes per query, no N x N tensor in
// HBM; backward recomputes the probabilities from the saved log-sum-exp.  Parameter gradients are
// reduced per workgroup through LDS outer products and then added with fp32 atomics.  The patch
// embedding (a 4096-deep contraction) and its weight gradient run on v_mfma_f32_32x32x2_f32.
#include "transformer.h"

namespace {

constexpr int TJ = 2;        // tokens per 32-lane half-wave group
constexpr int TB = 8 * TJ;   // tokens per workgroup
// The two forward token kernels of a dense layer have no per-workgroup weight-gradient work to amortise: one token
// per half-wave (8 per workgroup, 512 workgroups) is 10-20 % faster there; the backward kernels and the block
// out-layer (large weight panels, atomics per workgroup) are fastest at TJ = 2 (measured at 1, 2 and 4).
constexpr int FWD_TJ = 1, FWD_TB = 8 * FWD_TJ;
constexpr size_t LDS_LIMIT = 160 * 1024;

__device__ __forceinline__ float hsum32(float v) {  // sum over the 32 lanes of a half-wave
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float dot_row(const float* __restrict__ w, const float* sv, int K) {
  float acc = 0.f;
  for (int k = 0; k < K; k += 4) {
    float4 a = *reinterpret_cast<const float4*>(w + k);
    float4 b = *reinterpret_cast<const float4*>(sv + k);
    acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  return acc;
}
// sum_o W[o*ld + col] * sv[o]
__device__ __forceinline__ float dot_col(const float* __restrict__ w, int ld, int col, const float* sv, int O) {
  float acc = 0.f;
  for (int o = 0; o < O; o++) acc += w[(int64_t)o * ld + col] * sv[o];
  return acc;
}
// ---- weights through LDS ----------------------------------------------------------------------------------
// Every lane of a half-wave owns one output row of a Linear; reading that row straight from global memory makes
// each load touch 32 different cache lines.  Instead the [O][K] matrix is staged once per workgroup with
// coalesced loads into LDS rows of ODD stride (K+1 floats): "lane = row" reads (forward) and "lane = column"
// reads (transposed products in backward) are both bank-conflict free on the same image.
// The loads are float4 (every parameter starts on a 64-byte boundary, K % 4 == 0) and SU of them are in flight per
// thread before the first LDS store: one memory round trip per 8192 floats instead of one per 256.
constexpr int SU = 8;
__device__ __forceinline__ void stage_w(float* sW, const float* __restrict__ gW, int O, int K) {
  const int ldw = K + 1, total4 = (O * K) >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(gW);
  for (int base = threadIdx.x; base < total4; base += 256 * SU) {
    float4 v[SU];
#pragma unroll
    for (int u = 0; u < SU; u++) v[u] = g4[min(base + 256 * u, total4 - 1)];
#pragma unroll
    for (int u = 0; u < SU; u++) {
      const int i = (base + 256 * u) * 4;
      if (i < 4 * total4) {
        const int o = i / K, k = i - o * K;
        float* q = sW + o * ldw + k;
        q[0] = v[u].x, q[1] = v[u].y, q[2] = v[u].z, q[3] = v[u].w;
      }
    }
  }
}
// TB token rows of K floats (row t of the modality's [BN][ld] matrix, rows past BN: clamped or zero) -> sX[TB][K]
template <bool ZERO, int TBv = TB>
__device__ __forceinline__ void stage_rows(float* sX, const float* __restrict__ gX, int64_t ld, int K, int t0, int BN) {
  const int total4 = (TBv * K) >> 2;
  for (int base = threadIdx.x; base < total4; base += 256 * SU) {
    float4 v[SU];
#pragma unroll
    for (int u = 0; u < SU; u++) {
      const int i = min(base + 256 * u, total4 - 1) * 4, tl = i / K, k = i - tl * K;
      v[u] = *reinterpret_cast<const float4*>(gX + (int64_t)min(t0 + tl, BN - 1) * ld + k);
    }
#pragma unroll
    for (int u = 0; u < SU; u++) {
      const int i = (base + 256 * u) * 4;
      if (i < 4 * total4) {
        if (ZERO && t0 + i / K >= BN) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(sX + i) = v[u];
      }
    }
  }
}
// sum_k sWrow[k] * sv[k]   (sv 16-byte aligned, K % 4 == 0)
__device__ __forceinline__ float dot_lds(const float* sWrow, const float* sv, int K) {
  float acc = 0.f;
  for (int k = 0; k < K; k += 4) {
    float4 b = *reinterpret_cast<const float4*>(sv + k);
    acc += sWrow[k] * b.x + sWrow[k + 1] * b.y + sWrow[k + 2] * b.z + sWrow[k + 3] * b.w;
  }
  return acc;
}
// NT tokens (rows g, g+8, ... of sX) share every weight read: acc[j] += sum_k sWrow[k] * sX[(g+8j)*ldx + k]
template <int NT>
__device__ __forceinline__ void dot_lds_multi(const float* sWrow, const float* sX, int ldx, int g, int K, float* acc) {
  for (int k = 0; k < K; k += 4) {
    const float w0 = sWrow[k], w1 = sWrow[k + 1], w2 = sWrow[k + 2], w3 = sWrow[k + 3];
#pragma unroll
    for (int j = 0; j < NT; j++) {
      float4 b = *reinterpret_cast<const float4*>(sX + (g + 8 * j) * ldx + k);
      acc[j] += w0 * b.x + w1 * b.y + w2 * b.z + w3 * b.w;
    }
  }
}
// sum_o sW[o*ldw + col] * sv[o]
__device__ __forceinline__ float dot_col_lds(const float* sW, int ldw, int col, const float* sv, int O) {
  float acc = 0.f;
  for (int o = 0; o < O; o += 4) {
    float4 b = *reinterpret_cast<const float4*>(sv + o);
    acc += sW[o * ldw + col] * b.x + sW[(o + 1) * ldw + col] * b.y + sW[(o + 2) * ldw + col] * b.z +
           sW[(o + 3) * ldw + col] * b.w;
  }
  return acc;
}

// select AFTER an unconditional load (the argument is evaluated before the call): never branch around a load
__device__ __forceinline__ float sel0(bool ok, float v) { return ok ? v : 0.f; }

struct Drop {
  int training;
  uint32_t seed, thresh;
  float scale;
  __device__ __forceinline__ float mask(uint32_t site, uint32_t idx) const {
    if (!training) return 1.f;
    return hdf_keep(hdf_site_key(seed, site), idx, thresh) ? scale : 0.f;
  }
};
__device__ __forceinline__ Drop make_drop(const TfDims& d) { return Drop{d.training, d.seed, d.thresh24, d.keep_scale}; }

// gW[o*K + k] += sum_t sA[t*lda + o] * sB[t*ldb + k]   (t < TB; padded tokens hold zeros)
// One 32x32 block of gW per wave on v_mfma_f32_32x32x2_f32 (exact fp32 fma chains): TB/2 steps of two tokens, each
// lane feeding one sA and one sB element per step (conflict-free rows), then 16 coalesced atomics per lane.  The
// per-thread dot-product form of this read 2 * TB LDS words per output.  wrot rotates the block -> wave map so that
// two small products issued back to back land on different waves.
__device__ __forceinline__ void outer_acc(float* __restrict__ gW, int O, int K, const float* sA, int lda,
                                          const float* sB, int ldb, int wrot = 0) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int nbk = (K + 31) >> 5, nblk = ((O + 31) >> 5) * nbk;
  for (int blk = (wave - wrot) & 3; blk < nblk; blk += 4) {
    const int bo = blk / nbk, o0 = bo * 32, k0 = (blk - bo * nbk) * 32;
    const bool aok = o0 + r < O, bok = k0 + r < K;
    const float* pa = sA + h * lda + min(o0 + r, O - 1);
    const float* pb = sB + h * ldb + min(k0 + r, K - 1);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < TB / 2; s++) {
      const float av = aok ? pa[2 * s * lda] : 0.f;
      const float bv = bok ? pb[2 * s * ldb] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    if (bok) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int o = o0 + (i & 3) + 8 * (i >> 2) + 4 * h;  // accumulator i of this lane: row o, column k0 + r
        if (o < O) atomicAdd(gW + (int64_t)o * K + k0 + r, acc[i]);
      }
    }
  }
}
// gb[o] += sum_t sA[t*lda + o]
__device__ __forceinline__ void col_acc(float* __restrict__ gb, int O, const float* sA, int lda) {
  for (int o = threadIdx.x; o < O; o += 256) {
    float s = 0.f;
    for (int t = 0; t < TB; t++) s += sA[t * lda + o];
    atomicAdd(gb + o, s);
  }
}

#define TOK_LOOP(j, tl, t, ok, R)                       \
  _Pragma("unroll") for (int j = 0; j < TJ; j++)        \
    if (int tl = (threadIdx.x >> 5) + 8 * j; true)      \
      if (int t = blockIdx.x * TB + tl; true)           \
        if (bool ok = t < BN; true)                     \
          if (int64_t R = (int64_t)m * BN + (ok ? t : 0); true)

// ------------------------------------------------------------------------------ K1: Linear0 + LN1 + QKV
__global__ __launch_bounds__(256) void dense_pre_fwd_kernel(TfDims d, int K, TfLayerP p, const float* __restrict__ F,
                                                            float* __restrict__ h0, float* __restrict__ qkv) {
  constexpr int TJ = FWD_TJ, TB = FWD_TB;  // shadow the file-wide token grouping (TOK_LOOP picks these up)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_in = sm;                        // [TB][K]
  float* s_t = s_in + TB * K;              // [TB][32]
  float* s_w0 = s_t + TB * 32;             // [32][K+1]
  float* s_wq = s_w0 + 32 * (K + 1);       // [96][33]
  const int m = blockIdx.y, BN = d.B * d.N, o = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t mo = (int64_t)m * d.mstride;
  stage_rows<false, TB>(s_in, F + (int64_t)m * BN * d.DMF, d.DMF, K, blockIdx.x * TB, BN);
  stage_w(s_w0, p.w0 + mo, 32, K);
  stage_w(s_wq, p.wqkv + mo, 96, 32);
  __syncthreads();
  {
    float hacc[4] = {0.f, 0.f, 0.f, 0.f};
    dot_lds_multi<TJ>(s_w0 + o * (K + 1), s_in, K, grp, K, hacc);
    TOK_LOOP(j, tl, t, ok, R) {
      float h = p.b0[mo + o] + hacc[j];
      float mean = hsum32(h) * (1.f / 32.f);
      float dd = h - mean;
      float rstd = rsqrtf(hsum32(dd * dd) * (1.f / 32.f) + 1e-5f);
      s_t[tl * 32 + o] = dd * rstd * p.ln1g[mo + o] + p.ln1b[mo + o];
      if (ok) h0[R * 32 + o] = h;
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 3; c++) {
    int jj = o + 32 * c;
    float qacc[4] = {0.f, 0.f, 0.f, 0.f};
    dot_lds_multi<TJ>(s_wq + jj * 33, s_t, 32, grp, 32, qacc);
    TOK_LOOP(j, tl, t, ok, R) {
      if (ok) qkv[R * 96 + jj] = qacc[j];
    }
  }
}

// ------------------------------------------------------------------------------ K2: attention
// grid (ceil(N/AQ), 8 heads, M*B).  QL lanes per query, keys interleaved over the QL lanes.  The pair loops are
// VALU-bound (N^2 pairs x 8 heads x M*B sequences, head dim 4), so they are written on 2-wide fp32 vectors
// (v_pk_mul/fma_f32), the exponentials are raw v_exp_f32 on scores pre-scaled by log2(e) (folded into the 0.5
// query scale), and the LDS arrays are padded to whole trips so that only the fwd/dQ tail trip carries a mask
// (dK/dV pads with lse = +inf: the probability of a padded query is exp2(-inf) = 0).
constexpr int QL = 4, AQ = 256 / QL, AU = 4, ATRIP = QL * AU;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 lo2(const float4& v) { return f2{v.x, v.y}; }
__device__ __forceinline__ f2 hi2(const float4& v) { return f2{v.z, v.w}; }
__device__ __forceinline__ float dot4(f2 a01, f2 a23, f2 b01, f2 b23) {
  f2 t = a01 * b01;
  t = __builtin_elementwise_fma(a23, b23, t);
  return t.x + t.y;
}
__host__ __device__ inline int attn_rows(int N) { return (N + ATRIP - 1) / ATRIP * ATRIP; }

// K and V rows of one (sequence, head) -> LDS, padded rows zero; 4 row pairs in flight per thread
__device__ __forceinline__ void attn_stage_kv(int N, int NP, const float* __restrict__ qkv, int64_t rowbase, int head,
                                              float4* sK, float4* sV) {
  for (int base = threadIdx.x; base < NP; base += 256 * 4) {
    float4 k[4], v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const float* r = qkv + (rowbase + min(base + 256 * u, N - 1)) * 96 + head * 4;
      k[u] = *reinterpret_cast<const float4*>(r + 32);
      v[u] = *reinterpret_cast<const float4*>(r + 64);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = base + 256 * u;
      if (i < NP) {
        const bool real = i < N;
        sK[i] = real ? k[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        sV[i] = real ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
}

// QK^T on the matrix core.  One v_mfma_f32_16x16x4_f32 (exact fp32, K = 4 = the head width) gives the 16 x 16 scores of a
// block of 16 keys (A rows) against the wave's 16 queries (B columns): lane (q = lane & 15, g = lane >> 4) receives the
// scores of keys j0 + 4g .. 4g + 3 for its query -- exactly the "4 lanes per query, keys interleaved" decomposition of
// the VALU version, with the packed dot products (3 of its 14 instruction slots per pair) gone.  The exponentials, the
// running max / rescale and the 4-wide P.V update stay on the VALU (P.V as an MFMA would fill 4 of the 16 tile rows).
// The MFMA of block j0 + 16 is issued before the exponentials of block j0 (40 cycles of dependent latency).
__device__ __forceinline__ f32x4 attn_scores(const float* sKf, int j0, float bq) {
  const int lane = threadIdx.x & 63;
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return __builtin_amdgcn_mfma_f32_16x16x4f32(sKf[(j0 + (lane & 15)) * 4 + (lane >> 4)], bq, z, 0, 0, 0);
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(int N, const float* __restrict__ qkv, float* __restrict__ ob,
                                                       float* __restrict__ lse) {
  extern __shared__ float4 skv[];
  const int NP = attn_rows(N);
  float4* sK = skv;
  float4* sV = skv + NP;
  const float* sKf = reinterpret_cast<const float*>(sK);
  const int head = blockIdx.y;
  const int64_t rowbase = (int64_t)blockIdx.z * N;
  attn_stage_kv(N, NP, qkv, rowbase, head, sK, sV);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int qi = blockIdx.x * AQ + 16 * wave + (lane & 15);
  const bool ok = qi < N;
  // B operand of every score MFMA: this lane's query, component g, pre-scaled (dim_head^-0.5 = 0.5; log2 units)
  const float bq = qkv[(rowbase + (ok ? qi : 0)) * 96 + head * 4 + g] * (0.5f * LOG2E);
  float mx = -INFINITY, l = 0.f;
  f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
  auto trip = [&](int j0, const f32x4& sc4, auto masked) __attribute__((always_inline)) {
    float4 v[AU];
    float sc[AU];
#pragma unroll
    for (int u = 0; u < AU; u++) v[u] = sV[j0 + 4 * g + u];
    float mn = mx;
#pragma unroll
    for (int u = 0; u < AU; u++) {
      sc[u] = sc4[u];
      if (decltype(masked)::value) sc[u] = (j0 + 4 * g + u < N) ? sc[u] : -INFINITY;
      mn = fmaxf(mn, sc[u]);
    }
    const float mr = (mn == -INFINITY) ? 0.f : mn;  // a lane with no key yet: exp2(-inf - mr) = 0, not NaN
    const float c = __builtin_amdgcn_exp2f(mx - mr);
    float ps = 0.f;
    f2 b01 = {0.f, 0.f}, b23 = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < AU; u++) {
      const float pr = __builtin_amdgcn_exp2f(sc[u] - mr);
      const f2 pp = {pr, pr};
      ps += pr;
      b01 = __builtin_elementwise_fma(pp, lo2(v[u]), b01);
      b23 = __builtin_elementwise_fma(pp, hi2(v[u]), b23);
    }
    const f2 cc = {c, c};
    l = l * c + ps;
    a01 = __builtin_elementwise_fma(a01, cc, b01);
    a23 = __builtin_elementwise_fma(a23, cc, b23);
    mx = mn;
  };
  static_assert(AU == 4 && ATRIP == 16, "one 16-key MFMA block per trip, 4 keys per lane");
  const int nfull = N / ATRIP * ATRIP;
  f32x4 cur = attn_scores(sKf, 0, bq);
  int j0 = 0;
  for (; j0 < nfull; j0 += ATRIP) {
    const f32x4 nxt = attn_scores(sKf, min(j0 + ATRIP, NP - ATRIP), bq);
    __builtin_amdgcn_sched_barrier(0);   // keep the next block's MFMA ahead of this block's exponentials
    trip(j0, cur, std::false_type{});
    cur = nxt;
  }
  if (j0 < NP) trip(j0, cur, std::true_type{});
  // merge the 4 key subsets of a query (lanes q, q + 16, q + 32, q + 48)
#pragma unroll
  for (int off = 16; off < 64; off <<= 1) {
    const float m2 = __shfl_xor(mx, off, 64), l2 = __shfl_xor(l, off, 64);
    const f2 b01 = {__shfl_xor(a01.x, off, 64), __shfl_xor(a01.y, off, 64)};
    const f2 b23 = {__shfl_xor(a23.x, off, 64), __shfl_xor(a23.y, off, 64)};
    const float mn = fmaxf(mx, m2);
    const float mr = (mn == -INFINITY) ? 0.f : mn;
    const float ca = __builtin_amdgcn_exp2f(mx - mr), cb = __builtin_amdgcn_exp2f(m2 - mr);
    l = l * ca + l2 * cb;
    a01 = a01 * ca + b01 * cb;
    a23 = a23 * ca + b23 * cb;
    mx = mn;
  }
  if (ok && g == 0) {
    const float inv = 1.f / l;
    *reinterpret_cast<float4*>(ob + (rowbase + qi) * 32 + head * 4) =
        make_float4(a01.x * inv, a01.y * inv, a23.x * inv, a23.y * inv);
    lse[(rowbase + qi) * 8 + head] = mx * LN2 + __logf(l);  // natural-log units, as the backward expects
  }
}

// dQ: same decomposition as forward; the two 16 x 16 x 4 products per block (scores K.Q and T = V.dO) on the matrix core
__device__ __forceinline__ void attn_bwd_dq_body(int N, int seq, const float* __restrict__ qkv,
                                                 const float* __restrict__ ob, const float* __restrict__ lse,
                                                 const float* __restrict__ dO, float* __restrict__ dqkv, float4* skv) {
  const int NP = attn_rows(N);
  float4* sK = skv;
  float4* sV = skv + NP;
  const float* sKf = reinterpret_cast<const float*>(sK);
  const float* sVf = reinterpret_cast<const float*>(sV);
  const int head = blockIdx.y;
  const int64_t rowbase = (int64_t)seq * N;
  attn_stage_kv(N, NP, qkv, rowbase, head, sK, sV);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int qi = blockIdx.x * AQ + 16 * wave + (lane & 15);
  const bool ok = qi < N;
  const int64_t R = rowbase + (ok ? qi : 0);
  const float4 go = *reinterpret_cast<const float4*>(dO + R * 32 + head * 4);
  const float4 oo = *reinterpret_cast<const float4*>(ob + R * 32 + head * 4);
  const float delta = go.x * oo.x + go.y * oo.y + go.z * oo.z + go.w * oo.w;
  const float ls = lse[R * 8 + head] * LOG2E;
  const float bq = qkv[R * 96 + head * 4 + g] * (0.5f * LOG2E);   // B operands: query / dO component g of this lane's row
  const float bg = dO[R * 32 + head * 4 + g];
  f2 d01 = {0.f, 0.f}, d23 = {0.f, 0.f};
  auto trip = [&](int j0, const f32x4& s4, const f32x4& t4, auto masked) __attribute__((always_inline)) {
    float4 k[AU];
#pragma unroll
    for (int u = 0; u < AU; u++) k[u] = sK[j0 + 4 * g + u];
#pragma unroll
    for (int u = 0; u < AU; u++) {
      float pr = __builtin_amdgcn_exp2f(s4[u] - ls);
      if (decltype(masked)::value) pr = (j0 + 4 * g + u < N) ? pr : 0.f;
      const float ds = pr * (t4[u] - delta);
      const f2 dd = {ds, ds};
      d01 = __builtin_elementwise_fma(dd, lo2(k[u]), d01);
      d23 = __builtin_elementwise_fma(dd, hi2(k[u]), d23);
    }
  };
  const int nfull = N / ATRIP * ATRIP;
  f32x4 cs = attn_scores(sKf, 0, bq), ct = attn_scores(sVf, 0, bg);
  int j0 = 0;
  for (; j0 < nfull; j0 += ATRIP) {
    const int jn = min(j0 + ATRIP, NP - ATRIP);
    const f32x4 ns = attn_scores(sKf, jn, bq), nt = attn_scores(sVf, jn, bg);
    __builtin_amdgcn_sched_barrier(0);
    trip(j0, cs, ct, std::false_type{});
    cs = ns, ct = nt;
  }
  if (j0 < NP) trip(j0, cs, ct, std::true_type{});
#pragma unroll
  for (int off = 16; off < 64; off <<= 1) {
    d01.x += __shfl_xor(d01.x, off, 64), d01.y += __shfl_xor(d01.y, off, 64);
    d23.x += __shfl_xor(d23.x, off, 64), d23.y += __shfl_xor(d23.y, off, 64);
  }
  if (ok && g == 0)
    *reinterpret_cast<float4*>(dqkv + R * 96 + head * 4) =
        make_float4(0.5f * d01.x, 0.5f * d01.y, 0.5f * d23.x, 0.5f * d23.y);
}

// dK, dV: a lane owns one key (16 per wave) and every fourth group of 4 queries of each 16-query block; scores and
// T = dO.V per block on the matrix core (A = the staged query / dO block, B = this lane's key / value component)
__device__ __forceinline__ void attn_bwd_dkv_body(int N, int seq, const float* __restrict__ qkv,
                                                  const float* __restrict__ ob, const float* __restrict__ lse,
                                                  const float* __restrict__ dO, float* __restrict__ dqkv, float4* skv) {
  const int NP = attn_rows(N);
  float4* sQ = skv;        // pre-scaled by 0.5 log2(e)
  float4* sG = skv + NP;   // dO
  float2* sL = reinterpret_cast<float2*>(skv + 2 * NP);  // (lse log2(e), delta); padded rows: (+inf, 0)
  const float* sQf = reinterpret_cast<const float*>(sQ);
  const float* sGf = reinterpret_cast<const float*>(sG);
  const int head = blockIdx.y;
  const int64_t rowbase = (int64_t)seq * N;
  for (int base = threadIdx.x; base < NP; base += 256 * 4) {
    float4 q[4], go[4], oo[4];
    float lv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int64_t r = rowbase + min(base + 256 * u, N - 1);
      q[u] = *reinterpret_cast<const float4*>(qkv + r * 96 + head * 4);
      go[u] = *reinterpret_cast<const float4*>(dO + r * 32 + head * 4);
      oo[u] = *reinterpret_cast<const float4*>(ob + r * 32 + head * 4);
      lv[u] = lse[r * 8 + head];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = base + 256 * u;
      if (i < NP) {
        const bool real = i < N;
        const float sc = real ? 0.5f * LOG2E : 0.f;
        sQ[i] = make_float4(q[u].x * sc, q[u].y * sc, q[u].z * sc, q[u].w * sc);
        sG[i] = real ? go[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        sL[i] = real ? make_float2(lv[u] * LOG2E, go[u].x * oo[u].x + go[u].y * oo[u].y + go[u].z * oo[u].z +
                                                      go[u].w * oo[u].w)
                     : make_float2(INFINITY, 0.f);
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int kj = blockIdx.x * AQ + 16 * wave + (lane & 15);
  const bool ok = kj < N;
  const int64_t R = rowbase + (ok ? kj : 0);
  const float bk = qkv[R * 96 + 32 + head * 4 + g], bv = qkv[R * 96 + 64 + head * 4 + g];
  f2 dk01 = {0.f, 0.f}, dk23 = {0.f, 0.f}, dv01 = {0.f, 0.f}, dv23 = {0.f, 0.f};
  f32x4 cs = attn_scores(sQf, 0, bk), ct = attn_scores(sGf, 0, bv);
  for (int i0 = 0; i0 < NP; i0 += ATRIP) {
    const int in = min(i0 + ATRIP, NP - ATRIP);
    const f32x4 ns = attn_scores(sQf, in, bk), nt = attn_scores(sGf, in, bv);
    __builtin_amdgcn_sched_barrier(0);
    float4 q[AU], go[AU];
    float2 ld[AU];
#pragma unroll
    for (int u = 0; u < AU; u++) {
      q[u] = sQ[i0 + 4 * g + u];
      go[u] = sG[i0 + 4 * g + u];
      ld[u] = sL[i0 + 4 * g + u];
    }
#pragma unroll
    for (int u = 0; u < AU; u++) {
      const float pr = __builtin_amdgcn_exp2f(cs[u] - ld[u].x);   // padded queries: exp2(-inf) = 0
      const f2 pp = {pr, pr};
      dv01 = __builtin_elementwise_fma(pp, lo2(go[u]), dv01);
      dv23 = __builtin_elementwise_fma(pp, hi2(go[u]), dv23);
      const float ds = pr * (ct[u] - ld[u].y);
      const f2 dd = {ds, ds};
      dk01 = __builtin_elementwise_fma(dd, lo2(q[u]), dk01);
      dk23 = __builtin_elementwise_fma(dd, hi2(q[u]), dk23);
    }
    cs = ns, ct = nt;
  }
#pragma unroll
  for (int off = 16; off < 64; off <<= 1) {
    dk01.x += __shfl_xor(dk01.x, off, 64), dk01.y += __shfl_xor(dk01.y, off, 64);
    dk23.x += __shfl_xor(dk23.x, off, 64), dk23.y += __shfl_xor(dk23.y, off, 64);
    dv01.x += __shfl_xor(dv01.x, off, 64), dv01.y += __shfl_xor(dv01.y, off, 64);
    dv23.x += __shfl_xor(dv23.x, off, 64), dv23.y += __shfl_xor(dv23.y, off, 64);
  }
  if (ok && g == 0) {
    const float un = 1.f / LOG2E;  // the staged queries carry log2(e)
    *reinterpret_cast<float4*>(dqkv + R * 96 + 32 + head * 4) =
        make_float4(dk01.x * un, dk01.y * un, dk23.x * un, dk23.y * un);
    *reinterpret_cast<float4*>(dqkv + R * 96 + 64 + head * 4) = make_float4(dv01.x, dv01.y, dv23.x, dv23.y);
  }
}

// One launch for both halves of the attention backward: grid z = 2 * (M*B); the first M*B slices compute dQ, the
// rest dK/dV.  The two are independent (both only read qkv / dO / lse / ob), each alone fills the chip once.
__global__ __launch_bounds__(256) void attn_bwd_kernel(int N, int nseq, const float* __restrict__ qkv,
                                                       const float* __restrict__ ob, const float* __restrict__ lse,
                                                       const float* __restrict__ dO, float* __restrict__ dqkv) {
  extern __shared__ float4 skv[];
  if ((int)blockIdx.z < nseq)
    attn_bwd_dq_body(N, blockIdx.z, qkv, ob, lse, dO, dqkv, skv);
  else
    attn_bwd_dkv_body(N, blockIdx.z - nseq, qkv, ob, lse, dO, dqkv, skv);
}

// ------------------------------------------------------------------------------ K3: to_out + residual + ff + ff
__global__ __launch_bounds__(256) void dense_post_fwd_kernel(TfDims d, int block, int layer, TfLayerP p,
                                                             const float* __restrict__ h0, const float* __restrict__ ob,
                                                             float* __restrict__ h1s, float* __restrict__ h2s,
                                                             float* __restrict__ F) {
  constexpr int TJ = FWD_TJ, TB = FWD_TB;  // shadow the file-wide token grouping (TOK_LOOP picks these up)
  __shared__ __attribute__((aligned(16))) float s_a[TB][32], s_u[TB][32], s_f[TB][64];
  __shared__ float s_wo[32 * 33], s_w1[64 * 33], s_w2[32 * 65];
  const int m = blockIdx.y, BN = d.B * d.N, o = threadIdx.x & 31;
  const int64_t mo = (int64_t)m * d.mstride;
  const Drop dr = make_drop(d);
  const uint32_t site0 = hdf_site_id(m, block, layer, 0);
  float h1r[4], hcur[4];
  stage_w(s_wo, p.wout + mo, 32, 32);
  stage_w(s_w1, p.w1 + mo, 64, 32);
  stage_w(s_w2, p.w2 + mo, 32, 64);
  TOK_LOOP(j, tl, t, ok, R) { s_a[tl][o] = sel0(ok, ob[R * 32 + o]); }
  __syncthreads();
  TOK_LOOP(j, tl, t, ok, R) {
    float a = p.bout[mo + o] + dot_lds(s_wo + o * 33, s_a[tl], 32);
    a *= dr.mask(site0 + 0, (uint32_t)t * 32 + o);
    float h1 = a + sel0(ok, h0[R * 32 + o]);
    h1r[j] = h1;
    hcur[j] = h1;
    if (ok) h1s[R * 32 + o] = h1;
  }
  for (int pass = 0; pass < 2; pass++) {  // pass 0: h2 = ff(LN2(h1)) + h1 ; pass 1: feature = ff(LN2(h2))
    TOK_LOOP(j, tl, t, ok, R) {
      float mean = hsum32(hcur[j]) * (1.f / 32.f);
      float dd = hcur[j] - mean;
      float rstd = rsqrtf(hsum32(dd * dd) * (1.f / 32.f) + 1e-5f);
      s_u[tl][o] = dd * rstd * p.ln2g[mo + o] + p.ln2b[mo + o];
    }
    __syncthreads();
    TOK_LOOP(j, tl, t, ok, R) {
#pragma unroll
      for (int c = 0; c < 2; c++) {
        int jj = o + 32 * c;
        float z = p.b1[mo + jj] + dot_lds(s_w1 + jj * 33, s_u[tl], 32);
        s_f[tl][jj] = gelu_f(z) * dr.mask(site0 + 1 + 2 * pass, (uint32_t)t * 64 + jj);
      }
    }
    __syncthreads();
    TOK_LOOP(j, tl, t, ok, R) {
      float g = p.b2[mo + o] + dot_lds(s_w2 + o * 65, s_f[tl], 64);
      g *= dr.mask(site0 + 2 + 2 * pass, (uint32_t)t * 32 + o);
      if (pass == 0) {
        hcur[j] = g + h1r[j];
        if (ok) h2s[R * 32 + o] = hcur[j];
      } else if (ok) {
        F[R * d.DMF + d.DM + 32 * layer + o] = g;
      }
    }
    __syncthreads();
  }
}

// backward of K3.  Inputs: dfeat = dF[:, DM+32l .. +32], saved h1,h2,ob.  Outputs: dO (grad of attention
// output before to_out), dh0acc (= dh1, the residual path into h0), parameter gradients.
__global__ __launch_bounds__(256) void dense_post_bwd_kernel(TfDims d, int block, int layer, TfLayerP p, TfLayerP g,
                                                             const float* __restrict__ h1s,
                                                             const float* __restrict__ h2s,
                                                             const float* __restrict__ ob, const float* __restrict__ dF,
                                                             float* __restrict__ dO, float* __restrict__ dh0acc) {
  __shared__ __attribute__((aligned(16))) float s_u[TB][32], s_f[TB][64], s_dz[TB][64], s_dg[TB][32];
  __shared__ float s_red[8][32][2], s_wo[32 * 33], s_w1[64 * 33], s_w2[32 * 65];
  const int m = blockIdx.y, BN = d.B * d.N, o = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t mo = (int64_t)m * d.mstride;
  const Drop dr = make_drop(d);
  const uint32_t site0 = hdf_site_id(m, block, layer, 0);
  stage_w(s_wo, p.wout + mo, 32, 32);
  stage_w(s_w1, p.w1 + mo, 64, 32);
  stage_w(s_w2, p.w2 + mo, 32, 64);
  float dcur[4];   // gradient flowing into the current ff's output (post-dropout side)
  float dres[4];   // gradient of the residual input accumulated so far
  float gam = 0.f, bet = 0.f;  // LN2 gamma/beta gradient partials of this thread's channel
  TOK_LOOP(j, tl, t, ok, R) {
    dcur[j] = sel0(ok, dF[R * d.DMF + d.DM + 32 * layer + o]);
    dres[j] = 0.f;
  }
  for (int pass = 1; pass >= 0; pass--) {  // pass 1: second ff on h2 ; pass 0: first ff on h1
    const float* hs = pass ? h2s : h1s;
    float xh[4], rs[4], zz[4][2], mk[4][2];
    TOK_LOOP(j, tl, t, ok, R) {
      float h = sel0(ok, hs[R * 32 + o]);
      float mean = hsum32(h) * (1.f / 32.f);
      float dd = h - mean;
      rs[j] = rsqrtf(hsum32(dd * dd) * (1.f / 32.f) + 1e-5f);
      xh[j] = dd * rs[j];
      s_u[tl][o] = xh[j] * p.ln2g[mo + o] + p.ln2b[mo + o];
      s_dg[tl][o] = ok ? dcur[j] * dr.mask(site0 + 2 + 2 * pass, (uint32_t)t * 32 + o) : 0.f;
    }
    __syncthreads();
    TOK_LOOP(j, tl, t, ok, R) {
#pragma unroll
      for (int c = 0; c < 2; c++) {
        int jj = o + 32 * c;
        float z = p.b1[mo + jj] + dot_lds(s_w1 + jj * 33, s_u[tl], 32);
        float mkv = dr.mask(site0 + 1 + 2 * pass, (uint32_t)t * 64 + jj);
        zz[j][c] = z;
        mk[j][c] = mkv;
        s_f[tl][jj] = ok ? gelu_f(z) * mkv : 0.f;
        float df = dot_col_lds(s_w2, 65, jj, s_dg[tl], 32);
        s_dz[tl][jj] = ok ? df * mkv * gelu_grad(z) : 0.f;
      }
    }
    __syncthreads();
    outer_acc(g.w2 + mo, 32, 64, &s_dg[0][0], 32, &s_f[0][0], 64);
    col_acc(g.b2 + mo, 32, &s_dg[0][0], 32);
    outer_acc(g.w1 + mo, 64, 32, &s_dz[0][0], 64, &s_u[0][0], 32, 2);
    col_acc(g.b1 + mo, 64, &s_dz[0][0], 64);
    TOK_LOOP(j, tl, t, ok, R) {
      float du = dot_col_lds(s_w1, 33, o, s_dz[tl], 64);
      gam += du * xh[j];
      bet += du;
      float dxh = du * p.ln2g[mo + o];
      float m1 = hsum32(dxh) * (1.f / 32.f), m2 = hsum32(dxh * xh[j]) * (1.f / 32.f);
      float dh = rs[j] * (dxh - m1 - xh[j] * m2);
      if (pass == 1) {
        dcur[j] = ok ? dh : 0.f;   // h2 feeds only the second ff; its gradient flows into ff#1's output
        dres[j] = dcur[j];         // ... and into the residual h1
      } else {
        dres[j] += ok ? dh : 0.f;  // dh1 = dh2 + LN2-bwd path of ff#1
      }
    }
    __syncthreads();
  }
  // LN2 gamma/beta: reduce over the 8 token groups, one atomic per channel
  s_red[grp][o][0] = gam;
  s_red[grp][o][1] = bet;
  __syncthreads();
  if (grp == 0) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < 8; k++) a += s_red[k][o][0], b += s_red[k][o][1];
    atomicAdd(g.ln2g + mo + o, a);
    atomicAdd(g.ln2b + mo + o, b);
  }
  // to_out: a = (Wout.ob + bout) * mask ; h1 = a + h0
  TOK_LOOP(j, tl, t, ok, R) {
    s_dg[tl][o] = ok ? dres[j] * dr.mask(site0 + 0, (uint32_t)t * 32 + o) : 0.f;
    s_u[tl][o] = sel0(ok, ob[R * 32 + o]);
    if (ok) dh0acc[R * 32 + o] = dres[j];
  }
  __syncthreads();
  outer_acc(g.wout + mo, 32, 32, &s_dg[0][0], 32, &s_u[0][0], 32);
  col_acc(g.bout + mo, 32, &s_dg[0][0], 32);
  TOK_LOOP(j, tl, t, ok, R) {
    float v = dot_col_lds(s_wo, 33, o, s_dg[tl], 32);
    if (ok) dO[R * 32 + o] = v;
  }
}

// backward of K1.  dF[:, 0:K] += W0^T dh0
__global__ __launch_bounds__(256) void dense_pre_bwd_kernel(TfDims d, int K, TfLayerP p, TfLayerP g,
                                                            const float* __restrict__ F, const float* __restrict__ h0,
                                                            const float* __restrict__ dqkv,
                                                            const float* __restrict__ dh0acc, float* __restrict__ dF) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_in = sm;                 // [TB][K]
  float* s_t = s_in + TB * K;       // [TB][32]
  float* s_dq = s_t + TB * 32;      // [TB][96]
  float* s_dh = s_dq + TB * 96;     // [TB][32]
  float* s_red = s_dh + TB * 32;    // [8][32][2]
  float* s_w0 = s_red + 8 * 32 * 2; // [32][K+1]
  float* s_wq = s_w0 + 32 * (K + 1);  // [96][33]
  const int m = blockIdx.y, BN = d.B * d.N, o = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t mo = (int64_t)m * d.mstride;
  float xh[4], rs[4];
  stage_w(s_w0, p.w0 + mo, 32, K);
  stage_w(s_wq, p.wqkv + mo, 96, 32);
  stage_rows<true>(s_in, F + (int64_t)m * BN * d.DMF, d.DMF, K, blockIdx.x * TB, BN);
  TOK_LOOP(j, tl, t, ok, R) {
    float h = sel0(ok, h0[R * 32 + o]);
    float mean = hsum32(h) * (1.f / 32.f);
    float dd = h - mean;
    rs[j] = rsqrtf(hsum32(dd * dd) * (1.f / 32.f) + 1e-5f);
    xh[j] = dd * rs[j];
    s_t[tl * 32 + o] = ok ? xh[j] * p.ln1g[mo + o] + p.ln1b[mo + o] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) s_dq[tl * 96 + o + 32 * c] = sel0(ok, dqkv[R * 96 + o + 32 * c]);
  }
  __syncthreads();
  outer_acc(g.wqkv + mo, 96, 32, s_dq, 96, s_t, 32);
  float gam = 0.f, bet = 0.f;
  TOK_LOOP(j, tl, t, ok, R) {
    float dt = dot_col_lds(s_wq, 33, o, s_dq + tl * 96, 96);
    gam += dt * xh[j];
    bet += dt;
    float dxh = dt * p.ln1g[mo + o];
    float m1 = hsum32(dxh) * (1.f / 32.f), m2 = hsum32(dxh * xh[j]) * (1.f / 32.f);
    float dh = rs[j] * (dxh - m1 - xh[j] * m2) + sel0(ok, dh0acc[R * 32 + o]);
    s_dh[tl * 32 + o] = ok ? dh : 0.f;
  }
  s_red[(grp * 32 + o) * 2 + 0] = gam;
  s_red[(grp * 32 + o) * 2 + 1] = bet;
  __syncthreads();
  if (grp == 0) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < 8; k++) a += s_red[(k * 32 + o) * 2], b += s_red[(k * 32 + o) * 2 + 1];
    atomicAdd(g.ln1g + mo + o, a);
    atomicAdd(g.ln1b + mo + o, b);
  }
  outer_acc(g.w0 + mo, 32, K, s_dh, 32, s_in, K);
  col_acc(g.b0 + mo, 32, s_dh, 32);
  // dF[:, 0:K] += W0^T dh0, SU read-modify-writes in flight per thread (each element has one owner)
  for (int base = threadIdx.x; base < TB * K; base += 256 * SU) {
    float cur[SU];
    float* q[SU];
    bool okv[SU];
#pragma unroll
    for (int u = 0; u < SU; u++) {
      const int i = min(base + 256 * u, TB * K - 1), tl = i / K, k = i - tl * K, t = blockIdx.x * TB + tl;
      okv[u] = (base + 256 * u < TB * K) & (t < BN);
      q[u] = dF + ((int64_t)m * BN + min(t, BN - 1)) * d.DMF + k;
      cur[u] = *q[u];
    }
#pragma unroll
    for (int u = 0; u < SU; u++) {
      const int i = min(base + 256 * u, TB * K - 1), tl = i / K, k = i - tl * K;
      const float v = dot_col_lds(s_w0, K + 1, k, s_dh + tl * 32, 32);
      if (okv[u]) *q[u] = cur[u] + v;
    }
  }
}

// ------------------------------------------------------------------------------ K4: block out_layer
template <typename T>
__global__ __launch_bounds__(256) void block_out_fwd_kernel(TfDims d, int block, TfOutP p, const float* __restrict__ F,
                                                            float* __restrict__ next_F, T* __restrict__ attnall,
                                                            int stage_wb) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = d.DMF;
  float* s_in = sm;                    // [TB][K]
  float* s_f = s_in + TB * K;          // [TB][64]
  float* s_wa = s_f + TB * 64;         // [64][K+1]
  float* s_wb = s_wa + 64 * (K + 1);   // [DM][65] (only if it fits)
  const int m = blockIdx.y, BN = d.B * d.N, o = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t mo = (int64_t)m * d.mstride;
  const Drop dr = make_drop(d);
  const uint32_t site0 = hdf_site_id(m, block, 4, 0);
  stage_rows<false>(s_in, F + (int64_t)m * BN * d.DMF, d.DMF, K, blockIdx.x * TB, BN);
  stage_w(s_wa, p.wa + mo, 64, K);
  if (stage_wb) stage_w(s_wb, p.wb + mo, d.DM, 64);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 2; c++) {
    int jj = o + 32 * c;
    float zacc[4] = {0.f, 0.f, 0.f, 0.f};
    dot_lds_multi<TJ>(s_wa + jj * (K + 1), s_in, K, grp, K, zacc);
    TOK_LOOP(j, tl, t, ok, R) {
      float z = p.ba[mo + jj] + zacc[j];
      s_f[tl * 64 + jj] = gelu_f(z) * dr.mask(site0 + 0, (uint32_t)t * 64 + jj);
    }
  }
  __syncthreads();
  for (int c = o; c < d.DM; c += 32) {
    float vacc[4] = {0.f, 0.f, 0.f, 0.f};
    if (stage_wb) {
      dot_lds_multi<TJ>(s_wb + c * 65, s_f, 64, grp, 64, vacc);
    } else {
      TOK_LOOP(j, tl, t, ok, R) { vacc[j] = dot_row(p.wb + mo + c * 64, s_f + tl * 64, 64); }
    }
    TOK_LOOP(j, tl, t, ok, R) {
      float v = (p.bb[mo + c] + vacc[j]) * dr.mask(site0 + 1, (uint32_t)t * d.DM + c);
      if (ok) {
        if (next_F)
          next_F[R * d.DMF + c] = v;
        else {
          int b = t / d.N, n = t - b * d.N;
          ST<T>::st(attnall + ((int64_t)b * d.N + n) * ((int64_t)d.M * d.DM) + (int64_t)m * d.DM + c, v);
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void block_out_bwd_kernel(TfDims d, int block, TfOutP p, TfOutP g,
                                                            const float* __restrict__ F,
                                                            const float* __restrict__ dF_next,
                                                            const T* __restrict__ d_attnall, float* __restrict__ dF,
                                                            int stage_wb) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = d.DMF, DM = d.DM;
  float* s_in = sm;                // [TB][K]
  float* s_f = s_in + TB * K;      // [TB][64]
  float* s_dz = s_f + TB * 64;     // [TB][64]
  float* s_do = s_dz + TB * 64;    // [TB][DM]
  float* s_wa = s_do + TB * DM;    // [64][K+1]
  float* s_wb = s_wa + 64 * (K + 1);  // [DM][65] (only if it fits)
  const int m = blockIdx.y, BN = d.B * d.N, o = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t mo = (int64_t)m * d.mstride;
  const Drop dr = make_drop(d);
  const uint32_t site0 = hdf_site_id(m, block, 4, 0);
  stage_rows<true>(s_in, F + (int64_t)m * BN * d.DMF, d.DMF, K, blockIdx.x * TB, BN);
  // upstream gradient rows (next block's dF, or the attnall gradient in the conv dtype), SU loads in flight
  auto stage_do = [&](auto load) __attribute__((always_inline)) {
    for (int base = threadIdx.x; base < TB * DM; base += 256 * SU) {
      float v[SU];
#pragma unroll
      for (int u = 0; u < SU; u++) {
        const int i = min(base + 256 * u, TB * DM - 1), tl = i / DM, c = i - tl * DM;
        v[u] = load(min(blockIdx.x * TB + tl, BN - 1), c);
      }
#pragma unroll
      for (int u = 0; u < SU; u++) {
        const int i = base + 256 * u;
        if (i < TB * DM) {
          const int tl = i / DM, c = i - tl * DM, t = blockIdx.x * TB + tl;
          s_do[i] = t < BN ? v[u] * dr.mask(site0 + 1, (uint32_t)t * DM + c) : 0.f;
        }
      }
    }
  };
  if (dF_next)
    stage_do([&](int t, int c) { return dF_next[((int64_t)m * BN + t) * d.DMF + c]; });
  else
    stage_do([&](int t, int c) {
      const int b = t / d.N, n = t - b * d.N;
      return ST<T>::ld(d_attnall + ((int64_t)b * d.N + n) * ((int64_t)d.M * DM) + (int64_t)m * DM + c);
    });
  if (stage_wb) stage_w(s_wb, p.wb + mo, DM, 64);
  stage_w(s_wa, p.wa + mo, 64, K);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 2; c++) {
    int jj = o + 32 * c;
    float zacc[4] = {0.f, 0.f, 0.f, 0.f};
    dot_lds_multi<TJ>(s_wa + jj * (K + 1), s_in, K, grp, K, zacc);
    TOK_LOOP(j, tl, t, ok, R) {
      float z = p.ba[mo + jj] + zacc[j];
      float mk = dr.mask(site0 + 0, (uint32_t)t * 64 + jj);
      s_f[tl * 64 + jj] = ok ? gelu_f(z) * mk : 0.f;
      float df = stage_wb ? dot_col_lds(s_wb, 65, jj, s_do + tl * DM, DM)
                          : dot_col(p.wb + mo, 64, jj, s_do + tl * DM, DM);  // lanes = consecutive columns: coalesced
      s_dz[tl * 64 + jj] = ok ? df * mk * gelu_grad(z) : 0.f;
    }
  }
  __syncthreads();
  outer_acc(g.wb + mo, DM, 64, s_do, DM, s_f, 64);
  col_acc(g.bb + mo, DM, s_do, DM);
  outer_acc(g.wa + mo, 64, K, s_dz, 64, s_in, K);
  col_acc(g.ba + mo, 64, s_dz, 64);
  for (int i = threadIdx.x; i < TB * K; i += 256) {
    int tl = i / K, k = i - tl * K, t = blockIdx.x * TB + tl;
    if (t < BN) dF[((int64_t)m * BN + t) * d.DMF + k] = dot_col_lds(s_wa, K + 1, k, s_dz + tl * 64, 64);
  }
}

// ------------------------------------------------------------------------------ patch embedding (MFMA f32)
// tokens[32] x DM tile per workgroup, K = 4096 in chunks of 64 (4 rows of 16 voxels of the 16^3 brick)
__global__ __launch_bounds__(256) void patch_embed_fwd_kernel(TfDims d, const float* __restrict__ x, int D, int H,
                                                              int W, const float* __restrict__ wpe,
                                                              const float* __restrict__ bpe,
                                                              const float* __restrict__ pos, float* __restrict__ F) {
  extern __shared__ float sm[];
  constexpr int KC = 64, LD = KC + 1;
  float* sA = sm;             // [32][LD]
  float* sB = sm + 32 * LD;   // [DM][LD]
  const int m = blockIdx.y, BN = d.B * d.N, DM = d.DM;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int gh = H / 16, gw = W / 16;
  const float* wm = wpe + (int64_t)m * d.mstride;
  f32x16 acc[2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[a][i] = 0.f;
  // Software pipeline over the 64 chunks: the global loads of chunk k+1 are issued right after chunk k's tile is
  // in LDS and stay in flight under its 32 MFMAs per wave (the single-buffered loop exposed one HBM round trip per
  // chunk: 170 us for 4 GFLOP).  Tokens beyond BN are clamped to the last token: their outputs are never stored.
  float4 ra[2], rb[8];
  const int nbq = DM * KC / 4;  // float4 chunks of the B tile
  const float* xtok[2];         // this thread's two tokens: first voxel of their 16^3 bricks (loop invariant)
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int tl = ((threadIdx.x + 256 * u) * 4) >> 6, t = min(blockIdx.x * 32 + tl, BN - 1);
    const int b = t / d.N, n = t - b * d.N;
    const int gz = n / (gh * gw), gy = (n / gw) % gh, gx = n % gw;
    xtok[u] = x + ((((int64_t)b * d.M + m) * D + gz * 16) * H + gy * 16) * W + gx * 16;
  }
  auto load_chunk_regs = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int kk = ((threadIdx.x + 256 * u) * 4) & 63;  // element index in the [32][64] A chunk
      const int k = kc + kk, dz = k >> 8, dy = (k >> 4) & 15, dx = k & 15;
      ra[u] = *reinterpret_cast<const float4*>(xtok[u] + ((int64_t)dz * H + dy) * W + dx);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      int q = min((int)threadIdx.x + 256 * u, nbq - 1), i = q * 4;
      int c = i >> 6, kk = i & 63;
      rb[u] = *reinterpret_cast<const float4*>(wm + (int64_t)c * 4096 + kc + kk);
    }
  };
  load_chunk_regs(0);
  for (int kc = 0; kc < 4096; kc += KC) {
    __syncthreads();  // the previous chunk's MFMAs are done with the tile
#pragma unroll
    for (int u = 0; u < 2; u++) {
      int i = (threadIdx.x + 256 * u) * 4, tl = i >> 6, kk = i & 63;
      float* dst = sA + tl * LD + kk;
      dst[0] = ra[u].x, dst[1] = ra[u].y, dst[2] = ra[u].z, dst[3] = ra[u].w;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      int q = threadIdx.x + 256 * u;
      if (q < nbq) {
        int i = q * 4, c = i >> 6, kk = i & 63;
        float* dst = sB + c * LD + kk;
        dst[0] = rb[u].x, dst[1] = rb[u].y, dst[2] = rb[u].z, dst[3] = rb[u].w;
      }
    }
    for (int base = 2048; base < nbq; base += 2048) {  // token dims > 128: remaining rows of the weight tile
#pragma unroll
      for (int u = 0; u < 8; u++) {
        int q = min(base + (int)threadIdx.x + 256 * u, nbq - 1), i = q * 4;
        rb[u] = *reinterpret_cast<const float4*>(wm + (int64_t)(i >> 6) * 4096 + kc + (i & 63));
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        int q = base + threadIdx.x + 256 * u;
        if (q < nbq) {
          int i = q * 4;
          float* dst = sB + (i >> 6) * LD + (i & 63);
          dst[0] = rb[u].x, dst[1] = rb[u].y, dst[2] = rb[u].z, dst[3] = rb[u].w;
        }
      }
    }
    __syncthreads();
    if (kc + KC < 4096) load_chunk_regs(kc + KC);
#pragma unroll
    for (int a = 0; a < 2; a++) {
      int nb = wave + 4 * a;
      if (nb * 32 < DM) {
        for (int k2 = 0; k2 < KC / 2; k2++) {
          float av = sA[r * LD + 2 * k2 + h];
          float bv = sB[(nb * 32 + r) * LD + 2 * k2 + h];
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
        }
      }
    }
  }
  const Drop dr = make_drop(d);
  const uint32_t site = hdf_site_id(m, 63, 7, 7);
#pragma unroll
  for (int a = 0; a < 2; a++) {
    int nb = wave + 4 * a;
    if (nb * 32 < DM) {
      int c = nb * 32 + r;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        int tl = (i & 3) + 8 * (i >> 2) + 4 * h, t = blockIdx.x * 32 + tl;
        if (t < BN) {
          int n = t % d.N;
          float v = acc[a][i] + bpe[(int64_t)m * d.mstride + c] + pos[(int64_t)m * d.mstride + (int64_t)n * DM + c];
          v *= dr.mask(site, (uint32_t)t * DM + c);
          F[((int64_t)m * BN + t) * d.DMF + c] = v;
        }
      }
    }
  }
}

// Round-2 form: operands straight from global memory, no barrier in the main loop.  A workgroup owns 32 tokens x 64
// columns (grid = token tiles x DM/64 x modalities: 256 workgroups at the benchmark size; the LDS-staged kernel above
// had 128 and re-staged the whole 2 MB weight matrix of a modality in each).  The four waves SPLIT K: wave w contracts
// the 64-deep chunks c = w (mod 4) for the whole 32 x 64 tile (2 x 4 tiles of v_mfma_f32_16x16x4_f32), so no operand is
// loaded twice inside a workgroup (giving each wave 16 of the columns instead loaded the token rows four times and
// kept the vector-memory pipe, not the matrix pipe, busy: 73 us); the four partial tiles meet in LDS once, in a fixed
// order.  The contraction order is free, so lane group g = lane >> 4 owns k in [16 g, 16 g + 16) of a chunk: its A
// operands are 64 contiguous bytes of one brick row of its token, its B operands 64 contiguous bytes of one row of the
// Conv3d weight (float4 loads), prefetched one chunk (128 MFMAs = 4096 cycles) ahead.
__global__ __launch_bounds__(256) void patch_embed_fwd2_kernel(TfDims d, const float* __restrict__ x, int D, int H,
                                                               int W, const float* __restrict__ wpe,
                                                               const float* __restrict__ bpe,
                                                               const float* __restrict__ pos, float* __restrict__ F) {
  __shared__ float red[4][32][65];
  const int m = blockIdx.z, BN = d.B * d.N, DM = d.DM;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i16 = lane & 15, g = lane >> 4;
  const int gh = H / 16, gw = W / 16;
  const int col0 = blockIdx.y * 64;  // < DM (DM % 64 == 0)
  const float* const wrow = wpe + (int64_t)m * d.mstride + (int64_t)(col0 + i16) * 4096 + g * 16;  // + 16 nt rows
  const float* xt[2];  // first voxel of the 16^3 brick of this lane's token in each row tile (clamped: never stored)
#pragma unroll
  for (int mt = 0; mt < 2; mt++) {
    const int t = min(blockIdx.x * 32 + mt * 16 + i16, BN - 1);
    const int b = t / d.N, n = t - b * d.N;
    const int gz = n / (gh * gw), gy = (n / gw) % gh, gx = n % gw;
    xt[mt] = x + ((((int64_t)b * d.M + m) * D + gz * 16) * H + gy * 16) * W + gx * 16;
  }
  constexpr int NCH = 4096 / 64 / 4;  // chunks per wave
  f32x4 ra[2][2][4], rb[2][4][4];
  auto load_chunk = [&](int i, int slot) __attribute__((always_inline)) {
    const int c = i * 4 + wave;
    const int k16 = c * 4 + g, dz = k16 >> 4, dy = k16 & 15;  // brick row of this lane group's 16 k values
    const int64_t xo = ((int64_t)dz * H + dy) * W;
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) ra[slot][mt][s4] = *reinterpret_cast<const f32x4*>(xt[mt] + xo + 4 * s4);
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++)
        rb[slot][nt][s4] = *reinterpret_cast<const f32x4*>(wrow + (int64_t)nt * 16 * 4096 + c * 64 + 4 * s4);
  };
  f32x4 acc[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int nt = 0; nt < 4; nt++) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mma_chunk = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
          for (int nt = 0; nt < 4; nt++)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[slot][mt][s4][e], rb[slot][nt][s4][e], acc[mt][nt], 0, 0, 0);
  };
  load_chunk(0, 0);
  for (int i = 0; i < NCH; i += 2) {
    load_chunk(i + 1, 1);  // NCH is even
    mma_chunk(0);
    if (i + 2 < NCH) load_chunk(i + 2, 0);
    mma_chunk(1);
  }
  // C/D layout: row = 4 (lane >> 4) + i, column = lane & 15
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
      for (int i = 0; i < 4; i++) red[wave][mt * 16 + 4 * g + i][nt * 16 + i16] = acc[mt][nt][i];
  __syncthreads();
  const Drop dr = make_drop(d);
  const uint32_t site = hdf_site_id(m, 63, 7, 7);
  const int cl = threadIdx.x & 63, col = col0 + cl;
  const float bias = bpe[(int64_t)m * d.mstride + col];
  for (int tl = threadIdx.x >> 6; tl < 32; tl += 4) {
    const int t = blockIdx.x * 32 + tl;
    if (t < BN) {
      const int n = t % d.N;
      float v = ((red[0][tl][cl] + red[1][tl][cl]) + (red[2][tl][cl] + red[3][tl][cl])) + bias +
                pos[(int64_t)m * d.mstride + (int64_t)n * DM + col];
      v *= dr.mask(site, (uint32_t)t * DM + col);
      F[((int64_t)m * BN + t) * d.DMF + col] = v;
    }
  }
}

// masked token gradient + dpos + dbias
__global__ void patch_embed_bwd_prep_kernel(TfDims d, const float* __restrict__ dF, float* __restrict__ dtok,
                                            float* __restrict__ dbpe, float* __restrict__ dpos) {
  const int m = blockIdx.y, DM = d.DM, BN = d.B * d.N;
  const Drop dr = make_drop(d);
  const uint32_t site = hdf_site_id(m, 63, 7, 7);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.N * DM; i += gridDim.x * blockDim.x) {
    int n = i / DM, c = i - n * DM;
    float s = 0.f;
    for (int b = 0; b < d.B; b++) {
      int t = b * d.N + n;
      float v = dF[((int64_t)m * BN + t) * d.DMF + c] * dr.mask(site, (uint32_t)t * DM + c);
      dtok[((int64_t)m * BN + t) * DM + c] = v;
      s += v;
    }
    dpos[(int64_t)m * d.mstride + i] += s;
    atomicAdd(dbpe + (int64_t)m * d.mstride + c, s);
  }
}

// dW[c][k] = sum_t dtok[t][c] * patch[t][k].  grid (4096/128, ceil(DM/32), M); wave w owns k columns w*32..+32
__global__ __launch_bounds__(256) void patch_embed_wgrad_kernel(TfDims d, const float* __restrict__ x, int D, int H,
                                                                int W, const float* __restrict__ dtok,
                                                                float* __restrict__ dwpe) {
  constexpr int TT = 32, LDD = 33, LDP = 129;
  __shared__ float sD[TT * LDD], sP[TT * LDP];
  extern __shared__ int sTok[];  // [BN]: element offset of every token's brick in x (the per-load divisions by the
                                 // token grid cost more VALU time than the MFMAs: 25 runtime divisions per tile)
  const int m = blockIdx.z, BN = d.B * d.N, DM = d.DM;
  const int cb = blockIdx.y * 32, kb = blockIdx.x * 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int gh = H / 16, gw = W / 16;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  // software pipeline over the token tiles (as patch_embed_fwd_kernel): the loads of tile t+1 are in flight under
  // the 16 MFMAs of tile t; tokens beyond BN contribute zeros
  for (int t = threadIdx.x; t < BN; t += 256) {
    const int b = t / d.N, n = t - b * d.N;
    const int gz = n / (gh * gw), gy = (n / gw) % gh, gx = n % gw;
    sTok[t] = (int)(((((int64_t)b * d.M + m) * D + gz * 16) * H + gy * 16) * W + gx * 16);
  }
  int koff[4];  // brick-local offset of this thread's four k positions (loop invariant)
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int kk = ((threadIdx.x + 256 * u) * 4) & 127, k = kb + kk;
    koff[u] = ((k >> 8) * H + ((k >> 4) & 15)) * W + (k & 15);
  }
  __syncthreads();
  float4 rd, rp[4];
  auto load_tile_regs = [&](int t0) __attribute__((always_inline)) {
    {
      int i = threadIdx.x * 4, tl = i >> 5, c = i & 31, t = min(t0 + tl, BN - 1);   // [32 tok][32 c]
      rd = *reinterpret_cast<const float4*>(dtok + ((int64_t)m * BN + t) * DM + min(cb + c, DM - 4));
      if (t0 + tl >= BN || cb + c >= DM) rd = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int tl = ((threadIdx.x + 256 * u) * 4) >> 7, t = min(t0 + tl, BN - 1);  // [32 tok][128 kk]
      rp[u] = *reinterpret_cast<const float4*>(x + sTok[t] + koff[u]);
      if (t0 + tl >= BN) rp[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  load_tile_regs(0);
  for (int t0 = 0; t0 < BN; t0 += TT) {
    __syncthreads();  // the previous tile's MFMAs are done with the LDS tiles
    {
      int i = threadIdx.x * 4, tl = i >> 5, c = i & 31;
      float* dst = sD + tl * LDD + c;
      dst[0] = rd.x, dst[1] = rd.y, dst[2] = rd.z, dst[3] = rd.w;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      int i = (threadIdx.x + 256 * u) * 4, tl = i >> 7, kk = i & 127;
      float* dst = sP + tl * LDP + kk;
      dst[0] = rp[u].x, dst[1] = rp[u].y, dst[2] = rp[u].z, dst[3] = rp[u].w;
    }
    __syncthreads();
    if (t0 + TT < BN) load_tile_regs(t0 + TT);
    for (int t2 = 0; t2 < TT / 2; t2++) {
      float av = sD[(2 * t2 + h) * LDD + r];
      float bv = sP[(2 * t2 + h) * LDP + wave * 32 + r];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; i++) {
    int c = cb + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (c < DM) dwpe[(int64_t)m * d.mstride + (int64_t)c * 4096 + kb + wave * 32 + r] += acc[i];
  }
}

// dynamic LDS above 64 KB has to be opted into per kernel (once)
template <typename Kern>
int allow_lds(Kern kern, size_t bytes) {
  if (bytes <= 64 * 1024) return HDF_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)LDS_LIMIT);
  if (e != hipSuccess) {
    hdf_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(e));
    return HDF_ERR_HIP;
  }
  return HDF_OK;
}

}  // namespace

int tf_patch_embed_fwd(const TfDims& d, const float* x, int D, int H, int W, const float* wpe, const float* bpe,
                       const float* pos, float* F, hipStream_t st) {
  HDF_CHECK_ARG(d.DM <= 256 && d.DM % 32 == 0, "patch_embed: token dim %d unsupported", d.DM);
  static const bool pe_old = getenv("HDF_PE_OLD") != nullptr;  // A/B knob: the LDS-staged 32-token kernel
  if (d.DM % 64 == 0 && !pe_old) {
    hipLaunchKernelGGL(patch_embed_fwd2_kernel, dim3(ceil_div(d.B * d.N, 32), d.DM / 64, d.M), dim3(256), 0, st, d, x, D,
                       H, W, wpe, bpe, pos, F);
    HDF_LAUNCH_CHECK();
    return HDF_OK;
  }
  size_t shm = (size_t)(32 + d.DM) * 65 * sizeof(float);
  HDF_TRY(allow_lds(patch_embed_fwd_kernel, shm));
  hipLaunchKernelGGL(patch_embed_fwd_kernel, dim3(ceil_div(d.B * d.N, 32), d.M), dim3(256), shm, st, d, x, D, H, W, wpe,
                     bpe, pos, F);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int tf_patch_embed_bwd(const TfDims& d, const float* x, int D, int H, int W, const float* dF, float* dwpe, float* dbpe,
                       float* dpos, float* scratch, hipStream_t st) {
  hipLaunchKernelGGL(patch_embed_bwd_prep_kernel, dim3(ceil_div(d.N * d.DM, 256), d.M), dim3(256), 0, st, d, dF,
                     scratch, dbpe, dpos);
  HDF_LAUNCH_CHECK();
  HDF_CHECK_ARG((int64_t)d.B * d.M * D * H * W < ((int64_t)1 << 31), "patch_embed: volume exceeds 32-bit element offsets");
  hipLaunchKernelGGL(patch_embed_wgrad_kernel, dim3(4096 / 128, ceil_div(d.DM, 32), d.M), dim3(256),
                     (size_t)d.B * d.N * sizeof(int), st, d, x, D, H, W, scratch, dwpe);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int tf_attention_fwd(int N, int nseq, const float* qkv, float* ob, float* lse, hipStream_t st) {
  const size_t shm = (size_t)attn_rows(N) * 32;
  HDF_CHECK_ARG(N >= 1 && shm <= LDS_LIMIT, "attention: %d tokens need %zu B of LDS", N, shm);
  HDF_TRY(allow_lds(attn_fwd_kernel, shm));
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(ceil_div(N, AQ), 8, nseq), dim3(256), shm, st, N, qkv, ob, lse);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int tf_attention_bwd(int N, int nseq, const float* qkv, const float* ob, const float* lse, const float* dO, float* dqkv,
                     hipStream_t st) {
  const size_t shm = (size_t)attn_rows(N) * 40;
  HDF_CHECK_ARG(N >= 1 && shm <= LDS_LIMIT, "attention backward: %d tokens need %zu B of LDS", N, shm);
  HDF_TRY(allow_lds(attn_bwd_kernel, shm));
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(ceil_div(N, AQ), 8, 2 * nseq), dim3(256), shm, st, N, nseq, qkv, ob, lse, dO,
                     dqkv);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

bool tf_use_fused() {
  static const bool old = getenv("HDF_TF_OLD") != nullptr;  // A/B knob: the round-1 VALU token kernels
  return !old;
}

int tf_layer_fwd(const TfDims& d, int block, int layer, const TfLayerP& p, float* F, const TfLayerSave& s,
                 hipStream_t st) {
  if (tf_use_fused()) {  // one dense layer on its own: PRE, attention, POST as three launches of the fused kernels
    TfTokenFwd pre;
    pre.pre = &p, pre.pre_save = s, pre.bq = block, pre.lq = layer, pre.F_pre = F;
    HDF_TRY(tf_token_fwd(d, pre, HDF_F32, st));
    HDF_TRY(tf_attention_fwd(d.N, d.M * d.B, s.qkv, s.ob, s.lse, st));
    TfTokenFwd post;
    post.post = &p, post.post_save = s, post.bp = block, post.lp = layer, post.F_post = F;
    return tf_token_fwd(d, post, HDF_F32, st);
  }
  const int K = d.DM + 32 * layer, BN = d.B * d.N;
  dim3 grid(ceil_div(BN, FWD_TB), d.M);
  size_t shm = (size_t)(FWD_TB * K + FWD_TB * 32 + 32 * (K + 1) + 96 * 33) * sizeof(float);
  HDF_TRY(allow_lds(dense_pre_fwd_kernel, shm));
  hipLaunchKernelGGL(dense_pre_fwd_kernel, grid, dim3(256), shm, st, d, K, p, F, s.h0, s.qkv);
  HDF_LAUNCH_CHECK();
  HDF_TRY(tf_attention_fwd(d.N, d.M * d.B, s.qkv, s.ob, s.lse, st));
  hipLaunchKernelGGL(dense_post_fwd_kernel, grid, dim3(256), 0, st, d, block, layer, p, s.h0, s.ob, s.h1, s.h2, F);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int tf_layer_bwd(const TfDims& d, int block, int layer, const TfLayerP& p, const TfLayerP& g, const float* F, float* dF,
                 const TfLayerSave& s, float* scratch, hipStream_t st) {
  const int K = d.DM + 32 * layer, BN = d.B * d.N;
  const int64_t rows = (int64_t)d.M * BN;
  float* dO = scratch;
  float* dh0acc = scratch + rows * 32;
  float* dqkv = scratch + rows * 64;
  if (tf_use_fused()) {  // one dense layer on its own: POSTB, attention backward, PREB
    TfTokenBwd post;
    post.dF = dF, post.post = &p, post.post_grad = &g, post.post_save = s, post.bp = block, post.lp = layer;
    post.dO = dO, post.dh0acc_out = dh0acc;
    HDF_TRY(tf_token_bwd(d, post, HDF_F32, st));
    HDF_TRY(tf_attention_bwd(d.N, d.M * d.B, s.qkv, s.ob, s.lse, dO, dqkv, st));
    TfTokenBwd pre;
    pre.dF = dF, pre.pre = &p, pre.pre_grad = &g, pre.pre_save = s, pre.bq = block, pre.lq = layer, pre.F_pre = F;
    pre.dqkv = dqkv, pre.dh0acc = dh0acc;
    return tf_token_bwd(d, pre, HDF_F32, st);
  }
  dim3 grid(ceil_div(BN, TB), d.M);
  hipLaunchKernelGGL(dense_post_bwd_kernel, grid, dim3(256), 0, st, d, block, layer, p, g, s.h1, s.h2, s.ob, dF, dO,
                     dh0acc);
  HDF_LAUNCH_CHECK();
  HDF_TRY(tf_attention_bwd(d.N, d.M * d.B, s.qkv, s.ob, s.lse, dO, dqkv, st));
  size_t shm = (size_t)(TB * K + TB * 32 + TB * 96 + TB * 32 + 8 * 32 * 2 + 32 * (K + 1) + 96 * 33) * sizeof(float);
  HDF_TRY(allow_lds(dense_pre_bwd_kernel, shm));
  hipLaunchKernelGGL(dense_pre_bwd_kernel, grid, dim3(256), shm, st, d, K, p, g, F, s.h0, dqkv, dh0acc, dF);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int tf_block_out_fwd(const TfDims& d, int block, const TfOutP& p, const float* F, float* next_F, void* attnall,
                     int dtype, hipStream_t st) {
  if (tf_use_fused()) {
    TfTokenFwd o;
    o.out = &p, o.bp = block, o.F_post = const_cast<float*>(F), o.next_F = next_F, o.attnall = attnall;
    return tf_token_fwd(d, o, dtype, st);
  }
  dim3 grid(ceil_div(d.B * d.N, TB), d.M);
  size_t base = (size_t)(TB * d.DMF + TB * 64 + 64 * (d.DMF + 1)) * sizeof(float);
  size_t with_wb = base + (size_t)d.DM * 65 * sizeof(float);
  const int stage_wb = with_wb <= LDS_LIMIT ? 1 : 0;
  size_t shm = stage_wb ? with_wb : base;
  HDF_CHECK_ARG(shm <= LDS_LIMIT, "block_out: token dim %d needs %zu B of LDS", d.DM, shm);
  HDF_DISPATCH_T(dtype, {
    HDF_TRY(allow_lds(block_out_fwd_kernel<T>, shm));
    hipLaunchKernelGGL(block_out_fwd_kernel<T>, grid, dim3(256), shm, st, d, block, p, F, next_F, (T*)attnall, stage_wb);
  });
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int tf_block_out_bwd(const TfDims& d, int block, const TfOutP& p, const TfOutP& g, const float* F,
                     const float* dF_next, const void* d_attnall, int dtype, float* dF, hipStream_t st) {
  if (tf_use_fused()) {
    TfTokenBwd o;
    o.dF = dF, o.out = &p, o.out_grad = &g, o.bo = block, o.F_out = F, o.dF_next = dF_next, o.d_attnall = d_attnall;
    return tf_token_bwd(d, o, dtype, st);
  }
  dim3 grid(ceil_div(d.B * d.N, TB), d.M);
  size_t base = (size_t)(TB * d.DMF + TB * 64 * 2 + TB * d.DM + 64 * (d.DMF + 1)) * sizeof(float);
  size_t with_wb = base + (size_t)d.DM * 65 * sizeof(float);
  const int stage_wb = with_wb <= LDS_LIMIT ? 1 : 0;
  size_t shm = stage_wb ? with_wb : base;
  HDF_CHECK_ARG(shm <= LDS_LIMIT, "block_out_bwd: token dim %d needs %zu B of LDS", d.DM, shm);
  HDF_DISPATCH_T(dtype, {
    HDF_TRY(allow_lds(block_out_bwd_kernel<T>, shm));
    hipLaunchKernelGGL(block_out_bwd_kernel<T>, grid, dim3(256), shm, st, d, block, p, g, F, dF_next,
                       (const T*)d_attnall, dF, stage_wb);
  });
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}
