// Multi-path densely-connected Transformer branch: forward and backward kernels (fp32).
//
// Reference: models/HDenseFormer.py:33-145.  Per dense layer l of block b (DensePreConv_AttentionBlock
// .forward, :91-101):   h0 = Linear(cat(features));  h1 = attn(LN1(h0)) + h0;  h2 = ff(LN2(h1)) + h1;
// feature = ff(LN2(h2))  (second evaluation of the same ff, :98).  The feature concat is never
// materialised: every block owns a dense buffer F[rows][DM+128] whose column ranges ARE the features.
//
// This file: the attention core (a flash-style pass with K,V of one (sample, modality, head) in LDS, N x 4 floats
// each, 4 lanes per query, no N x N tensor in HBM; backward recomputes the probabilities from the saved
// log-sum-exp), the patch embedding (a 4096-deep contraction) and its weight gradient on the f32 MFMA, and the
// host-side entry points.  Everything of a dense layer between two attention launches lives in
// transformer_fused.hip (tok_fwd_kernel / tok_bwd_kernel / tf_wgrad_kernel); the round-1 VALU token kernels
// (dense_pre/post_*, block_out_*) were removed in round 3.
#include "transformer.h"

namespace {

constexpr size_t LDS_LIMIT = 160 * 1024;

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
// the same with a C operand: row r of the lane's column starts at c[r] (a subtraction folded into the matrix op)
__device__ __forceinline__ f32x4 attn_scores(const float* sKf, int j0, float bq, const f32x4& c) {
  const int lane = threadIdx.x & 63;
  return __builtin_amdgcn_mfma_f32_16x16x4f32(sKf[(j0 + (lane & 15)) * 4 + (lane >> 4)], bq, c, 0, 0, 0);
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(int N, const float* __restrict__ qkv, float* __restrict__ ob,
                                                       float* __restrict__ lse) {
  // Every sum of two products below is written as an explicit fma of one product into the other and contraction is
  // switched off: left to the compiler, `l * ca + l2 * cb` became fma(l, ca, l2 * cb) in one merge round and two
  // multiplies + an add in the next (SLP-vectorised products), i.e. the rounding depended on the surrounding code -- and
  // the persistent kernel of transformer_chain.hip must reproduce this kernel bit for bit.
#pragma clang fp contract(off)
  HDF_CHAIN_PRIO();
  extern __shared__ float4 skv[];
  const int NP = attn_rows(N);
  float4* sK = skv;
  float4* sV = skv + NP;
  const float* sKf = reinterpret_cast<const float*>(sK);
  const int head = blockIdx.y;
  const int64_t rowbase = (int64_t)blockIdx.z * N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int qi = blockIdx.x * AQ + 16 * wave + (lane & 15);
  const bool ok = qi < N;
  // B operand of every score MFMA: this lane's query, component g, pre-scaled (dim_head^-0.5 = 0.5; log2 units)
  // (requested before the staging loads: one global round trip less on this launch's critical path)
  const float bq = qkv[(rowbase + (ok ? qi : 0)) * 96 + head * 4 + g] * (0.5f * LOG2E);
  attn_stage_kv(N, NP, qkv, rowbase, head, sK, sV);
  __syncthreads();
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
    l = __builtin_fmaf(l, c, ps);
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
    const f2 ca2 = {ca, ca}, cb2 = {cb, cb};
    l = __builtin_fmaf(l, ca, l2 * cb);
    a01 = __builtin_elementwise_fma(a01, ca2, b01 * cb2);
    a23 = __builtin_elementwise_fma(a23, ca2, b23 * cb2);
    mx = mn;
  }
  if (ok && g == 0) {
    const float inv = 1.f / l;
    *reinterpret_cast<float4*>(ob + (rowbase + qi) * 32 + head * 4) =
        make_float4(a01.x * inv, a01.y * inv, a23.x * inv, a23.y * inv);
    lse[(rowbase + qi) * 8 + head] = __builtin_fmaf(mx, LN2, __logf(l));  // natural-log units, as the backward expects
  }
}

// ---- 16-bit storage modes: the three accumulations of the BACKWARD (dS.K, dS^T.Q, P^T.dO) on the matrix core.
// Under torch autocast the reference runs the matmuls of Dense_Attention with 16-bit operands and fp32 accumulation
// (trainer.py:369, HDenseFormer.py:70-73); the scores and the softmax stay on the exact-fp32 path here.  The dS / P values a
// lane holds after the exponentials -- 4 consecutive keys (queries) of the block for its query (key) -- are exactly the A
// operand of v_mfma_f32_16x16x16 (row = lane & 15, k = 4 (lane >> 4) + e), so they are packed to bf16 / f16 in place and
// ONE matrix instruction per 16 x 16 block replaces the lane's eight packed FMAs; the B operand comes from a TRANSPOSED
// 16-bit copy in LDS ([component][row]: 8 contiguous bytes per lane; lanes whose column is >= 4 read a row of zeros).
// The forward keeps the exact kernel in every mode: the same construction for P.V needs the row maximum before the first
// accumulation (the result tile has the query on the register index, which a running maximum cannot rescale), i.e. a
// second pass over the keys, and measured no faster (11.9 us against 11.5: the launch is bound by its prologue, not its
// 32-trip loop) -- so the forward output is exact and only the gradient sees the 16-bit operands.
template <int LP>
struct LpPack;
template <>
struct LpPack<1> {  // bf16
  static __device__ __forceinline__ uint16_t one(float v) { return f2bf(v); }
  static __device__ __forceinline__ u32x2 four(float a, float b, float c, float d) { return u32x2{pack_bf2(a, b), pack_bf2(c, d)}; }
  static __device__ __forceinline__ f32x4 mma(const u32x2& a, const u32x2& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
  }
};
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
template <>
struct LpPack<2> {  // f16
  static __device__ __forceinline__ uint16_t one(float v) { return f2h(v); }
  static __device__ __forceinline__ u32x2 four(float a, float b, float c, float d) { return u32x2{pack_h2(a, b), pack_h2(c, d)}; }
  static __device__ __forceinline__ f32x4 mma(const u32x2& a, const u32x2& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
  }
};
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
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int qi = blockIdx.x * AQ + 16 * wave + (lane & 15);
  const bool ok = qi < N;
  const int64_t R = rowbase + (ok ? qi : 0);
  const float4 go = *reinterpret_cast<const float4*>(dO + R * 32 + head * 4);
  const float4 oo = *reinterpret_cast<const float4*>(ob + R * 32 + head * 4);
  const float ls = lse[R * 8 + head] * LOG2E;
  const float bq = qkv[R * 96 + head * 4 + g] * (0.5f * LOG2E);   // B operands: query / dO component g of this lane's row
  const float bg = dO[R * 32 + head * 4 + g];
  attn_stage_kv(N, NP, qkv, rowbase, head, sK, sV);
  __syncthreads();
  const float delta = tf_dot4(go, oo);
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
        sL[i] = real ? make_float2(lv[u] * LOG2E, tf_dot4(go[u], oo[u]))
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
  HDF_CHAIN_PRIO();
  extern __shared__ float4 skv[];
  if ((int)blockIdx.z < nseq)
    attn_bwd_dq_body(N, blockIdx.z, qkv, ob, lse, dO, dqkv, skv);
  else
    attn_bwd_dkv_body(N, blockIdx.z - nseq, qkv, ob, lse, dO, dqkv, skv);
}

// ---- backward, 16-bit storage modes: the three accumulations on v_mfma_f32_16x16x16 (see attn_fwd_lp_kernel).
// dQ: A = dS of (query lane & 15, keys 4g + e), B = K^T 16-bit -> lane (c, g) holds dQ[query 4g + r][c].
// dK / dV: the lane's dS / P values are (key lane & 15, queries 4g + e), i.e. the A operand of dS^T.Q and P^T.dO as they
// stand; B = Q^T / dO^T 16-bit.  The exact version read 160 B of LDS per lane and trip in the dK/dV half (fp32 q, dO rows
// for the packed FMAs); this one reads 48.
__host__ __device__ inline size_t attn_bwd_lp_lds(int N) { return (size_t)attn_rows(N) * (16 + 16 + 8 + 10 + 10); }

template <int LP>
__device__ __forceinline__ void attn_bwd_dq_lp_body(int N, int seq, const float* __restrict__ qkv,
                                                    const float* __restrict__ ob, const float* __restrict__ lse,
                                                    const float* __restrict__ dO, float* __restrict__ dqkv, float4* skv) {
  const int NP = attn_rows(N);
  float4* sK = skv;
  float4* sV = skv + NP;
  uint16_t* sKt = reinterpret_cast<uint16_t*>(skv + 2 * NP);  // [5][NP]
  const float* sKf = reinterpret_cast<const float*>(sK);
  const float* sVf = reinterpret_cast<const float*>(sV);
  const int head = blockIdx.y;
  const int64_t rowbase = (int64_t)seq * N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int qi = blockIdx.x * AQ + 16 * wave + c;
  const bool ok = qi < N;
  const int64_t R = rowbase + (ok ? qi : 0);
  const float4 go = *reinterpret_cast<const float4*>(dO + R * 32 + head * 4);
  const float4 oo = *reinterpret_cast<const float4*>(ob + R * 32 + head * 4);
  const float ls = lse[R * 8 + head] * LOG2E;
  const float bq = qkv[R * 96 + head * 4 + g] * (0.5f * LOG2E);
  const float bg = dO[R * 32 + head * 4 + g];
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
        const float z = i < N ? 1.f : 0.f;
        const float4 kk = make_float4(k[u].x * z, k[u].y * z, k[u].z * z, k[u].w * z);
        sK[i] = kk;
        sV[i] = make_float4(v[u].x * z, v[u].y * z, v[u].z * z, v[u].w * z);
        sKt[0 * NP + i] = LpPack<LP>::one(kk.x), sKt[1 * NP + i] = LpPack<LP>::one(kk.y);
        sKt[2 * NP + i] = LpPack<LP>::one(kk.z), sKt[3 * NP + i] = LpPack<LP>::one(kk.w);
        sKt[4 * NP + i] = 0;
      }
    }
  }
  __syncthreads();
  const float delta = tf_dot4(go, oo);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const uint16_t* krow = sKt + (size_t)min(c, 4) * NP + 4 * g;
  const int nfull = N / ATRIP * ATRIP;
  // C operands: the scores come out as (score - lse), T as (dO.v - delta)
  const f32x4 nl = {-ls, -ls, -ls, -ls}, nd = {-delta, -delta, -delta, -delta};
  f32x4 cs = attn_scores(sKf, 0, bq, nl), ct = attn_scores(sVf, 0, bg, nd);
  int j0 = 0;
#pragma unroll 2
  for (; j0 < nfull; j0 += ATRIP) {
    const int jn = min(j0 + ATRIP, NP - ATRIP);
    const f32x4 ns = attn_scores(sKf, jn, bq, nl), nt = attn_scores(sVf, jn, bg, nd);
    const u32x2 kb = *reinterpret_cast<const u32x2*>(krow + j0);
    __builtin_amdgcn_sched_barrier(0);
    float ds[AU];
#pragma unroll
    for (int u = 0; u < AU; u++) ds[u] = __builtin_amdgcn_exp2f(cs[u]) * ct[u];
    acc = LpPack<LP>::mma(LpPack<LP>::four(ds[0], ds[1], ds[2], ds[3]), kb, acc);
    cs = ns, ct = nt;
  }
  if (j0 < NP) {
    const u32x2 kb = *reinterpret_cast<const u32x2*>(krow + j0);
    float ds[AU];
#pragma unroll
    for (int u = 0; u < AU; u++) ds[u] = (j0 + 4 * g + u < N) ? __builtin_amdgcn_exp2f(cs[u]) * ct[u] : 0.f;
    acc = LpPack<LP>::mma(LpPack<LP>::four(ds[0], ds[1], ds[2], ds[3]), kb, acc);
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int q2 = blockIdx.x * AQ + 16 * wave + 4 * g + r;
    if (c < 4 && q2 < N) dqkv[(rowbase + q2) * 96 + head * 4 + c] = 0.5f * acc[r];
  }
}

struct LsePair {
  f32x4 first, second;
};
template <int LP>
__device__ __forceinline__ void attn_bwd_dkv_lp_body(int N, int seq, const float* __restrict__ qkv,
                                                     const float* __restrict__ ob, const float* __restrict__ lse,
                                                     const float* __restrict__ dO, float* __restrict__ dqkv,
                                                     float4* skv) {
  const int NP = attn_rows(N);
  float4* sQ = skv;        // pre-scaled by 0.5 log2(e): score operand
  float4* sG = skv + NP;   // dO: operand of T = dO.V
  float* sLse = reinterpret_cast<float*>(skv + 2 * NP);  // -lse log2(e); padded rows: -inf
  float* sDel = sLse + NP;                                // -delta
  uint16_t* sQt = reinterpret_cast<uint16_t*>(sDel + NP);  // [5][NP], unscaled q
  uint16_t* sGt = sQt + 5 * NP;                           // [5][NP]
  const float* sQf = reinterpret_cast<const float*>(sQ);
  const float* sGf = reinterpret_cast<const float*>(sG);
  const int head = blockIdx.y;
  const int64_t rowbase = (int64_t)seq * N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int kj = blockIdx.x * AQ + 16 * wave + c;
  const bool ok = kj < N;
  const int64_t R = rowbase + (ok ? kj : 0);
  const float bk = qkv[R * 96 + 32 + head * 4 + g], bv = qkv[R * 96 + 64 + head * 4 + g];
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
        const float z = real ? 1.f : 0.f;
        sQ[i] = make_float4(q[u].x * sc, q[u].y * sc, q[u].z * sc, q[u].w * sc);
        sG[i] = make_float4(go[u].x * z, go[u].y * z, go[u].z * z, go[u].w * z);
        sLse[i] = real ? -lv[u] * LOG2E : -INFINITY;
        sDel[i] = -z * tf_dot4(go[u], oo[u]);
        sQt[0 * NP + i] = LpPack<LP>::one(q[u].x * z), sQt[1 * NP + i] = LpPack<LP>::one(q[u].y * z);
        sQt[2 * NP + i] = LpPack<LP>::one(q[u].z * z), sQt[3 * NP + i] = LpPack<LP>::one(q[u].w * z);
        sQt[4 * NP + i] = 0;
        sGt[0 * NP + i] = LpPack<LP>::one(go[u].x * z), sGt[1 * NP + i] = LpPack<LP>::one(go[u].y * z);
        sGt[2 * NP + i] = LpPack<LP>::one(go[u].z * z), sGt[3 * NP + i] = LpPack<LP>::one(go[u].w * z);
        sGt[4 * NP + i] = 0;
      }
    }
  }
  __syncthreads();
  f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
  const uint16_t* qrow = sQt + (size_t)min(c, 4) * NP + 4 * g;
  const uint16_t* grow = sGt + (size_t)min(c, 4) * NP + 4 * g;
  // the (-lse, -delta) of the block's queries are the C operands of the two score products (row r = query 4g + r):
  // scores arrive as (score - lse), T as (dO.v - delta); both are fetched one block ahead of the MFMA that takes them
  auto lrow = [&](int i0) __attribute__((always_inline)) {
    return LsePair{*reinterpret_cast<const f32x4*>(sLse + i0 + 4 * g), *reinterpret_cast<const f32x4*>(sDel + i0 + 4 * g)};
  };
  LsePair l0 = lrow(0);
  f32x4 cs = attn_scores(sQf, 0, bk, l0.first), ct = attn_scores(sGf, 0, bv, l0.second);
  LsePair ln = lrow(min(ATRIP, NP - ATRIP));
#pragma unroll 2
  for (int i0 = 0; i0 < NP; i0 += ATRIP) {
    const int in = min(i0 + ATRIP, NP - ATRIP);
    const f32x4 ns = attn_scores(sQf, in, bk, ln.first), nt = attn_scores(sGf, in, bv, ln.second);
    ln = lrow(min(i0 + 2 * ATRIP, NP - ATRIP));
    const u32x2 qb = *reinterpret_cast<const u32x2*>(qrow + i0);
    const u32x2 gb = *reinterpret_cast<const u32x2*>(grow + i0);
    __builtin_amdgcn_sched_barrier(0);
    const float p0 = __builtin_amdgcn_exp2f(cs[0]), p1 = __builtin_amdgcn_exp2f(cs[1]);
    const float p2 = __builtin_amdgcn_exp2f(cs[2]), p3 = __builtin_amdgcn_exp2f(cs[3]);
    dv = LpPack<LP>::mma(LpPack<LP>::four(p0, p1, p2, p3), gb, dv);
    dk = LpPack<LP>::mma(LpPack<LP>::four(p0 * ct[0], p1 * ct[1], p2 * ct[2], p3 * ct[3]), qb, dk);
    cs = ns, ct = nt;
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int k2 = blockIdx.x * AQ + 16 * wave + 4 * g + r;
    if (c < 4 && k2 < N) {
      dqkv[(rowbase + k2) * 96 + 32 + head * 4 + c] = 0.5f * dk[r];
      dqkv[(rowbase + k2) * 96 + 64 + head * 4 + c] = dv[r];
    }
  }
}

template <int LP>
__global__ __launch_bounds__(256) void attn_bwd_lp_kernel(int N, int nseq, const float* __restrict__ qkv,
                                                          const float* __restrict__ ob, const float* __restrict__ lse,
                                                          const float* __restrict__ dO, float* __restrict__ dqkv) {
  HDF_CHAIN_PRIO();
  extern __shared__ float4 skv[];
  if ((int)blockIdx.z < nseq)
    attn_bwd_dq_lp_body<LP>(N, blockIdx.z, qkv, ob, lse, dO, dqkv, skv);
  else
    attn_bwd_dkv_lp_body<LP>(N, blockIdx.z - nseq, qkv, ob, lse, dO, dqkv, skv);
}

// ------------------------------------------------------------------------------ patch embedding (MFMA f32)
// Operands straight from global memory, no barrier in the main loop.  A workgroup owns 32 tokens x 64
// columns (grid = token tiles x DM/64 x modalities: 256 workgroups at the benchmark size; round 1's LDS-staged kernel
// had 128 and re-staged the whole 2 MB weight matrix of a modality in each).  The four waves SPLIT K: wave w contracts
// the 64-deep chunks c = w (mod 4) for the whole 32 x 64 tile (2 x 4 tiles of v_mfma_f32_16x16x4_f32), so no operand is
// loaded twice inside a workgroup (giving each wave 16 of the columns instead loaded the token rows four times and
// kept the vector-memory pipe, not the matrix pipe, busy: 73 us); the four partial tiles meet in LDS once, in a fixed
// order.  The contraction order is free, so lane group g = lane >> 4 owns k in [16 g, 16 g + 16) of a chunk: its A
// operands are 64 contiguous bytes of one brick row of its token, its B operands 64 contiguous bytes of one row of the
// Conv3d weight (float4 loads), prefetched one chunk (128 MFMAs = 4096 cycles) ahead.
// LP = 1 / 2 (the plan's bf16 / f16 storage modes): the operands are rounded to that type and contracted with
// v_mfma_f32_16x16x16 -- what autocast does to the patch Conv3d (HDenseFormer.py:115-119 under trainer.py:369).  A lane's
// four consecutive fp32 values of a 16-byte load are exactly the four k of one 16x16x16 step (k = 4 g + e), for A and B
// alike: one matrix instruction per load pair instead of four, 32 per 64-deep chunk instead of 128 (this launch is the
// first of the forward's critical chain, and with fp32 steps it is bound by the matrix pipe: 2,048 steps of 32 cycles).
// kd (round 6): depth slices of the 16^3 kernel that are contracted -- 16, or 1 for the 2-D model's 16 x 16 patches on a
// depth-1 input (models/HDenseFormer_2D.py:112: the 2-D kernel sits on depth slice 0 of the embedded 16^3 one, i.e. the first
// 256 of a weight row's 4096 entries): K = 256, one 64-deep chunk per wave.
template <int LP>
__global__ __launch_bounds__(256) void patch_embed_fwd2_kernel(TfDims d, const float* __restrict__ x, int D, int H,
                                                               int W, const float* __restrict__ wpe,
                                                               const float* __restrict__ bpe,
                                                               const float* __restrict__ pos, float* __restrict__ F,
                                                               int kd) {
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
  const int NCH = kd;  // chunks per wave: kd * 256 / 64 / 4
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
    if constexpr (LP == 0) {
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
          for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
              acc[mt][nt] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(ra[slot][mt][s4][e], rb[slot][nt][s4][e], acc[mt][nt], 0, 0, 0);
    } else {
      constexpr int L = LP == 0 ? 1 : LP;
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) {
        u32x2 pa[2], pb[4];
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
          const f32x4& v = ra[slot][mt][s4];
          pa[mt] = LpPack<L>::four(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
          const f32x4& v = rb[slot][nt][s4];
          pb[nt] = LpPack<L>::four(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
          for (int nt = 0; nt < 4; nt++) acc[mt][nt] = LpPack<L>::mma(pa[mt], pb[nt], acc[mt][nt]);
      }
    }
  };
  load_chunk(0, 0);
  for (int i = 0; i < NCH; i += 2) {
    if (i + 1 < NCH) load_chunk(i + 1, 1);
    mma_chunk(0);
    if (i + 2 < NCH) load_chunk(i + 2, 0);
    if (i + 1 < NCH) mma_chunk(1);
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

// dW[c][k] = sum_t dtok[t][c] * patch[t][k].  grid (kblocks * tchunks, ceil(DM/32), M); wave w owns k columns w*32..+32.
// kblocks = 4096/128 (3-D) or 256/128 (round 6: the 2-D model's depth-1 patches, slice 0 of the embedded kernel); tchunks
// > 1 (the 2-D model: two K blocks would leave 16 workgroups walking 13,824 tokens each): a workgroup contracts its chunk of
// the tokens and ADDS with float atomics.
__global__ __launch_bounds__(256) void patch_embed_wgrad_kernel(TfDims d, const float* __restrict__ x, int D, int H,
                                                                int W, const float* __restrict__ dtok,
                                                                float* __restrict__ dwpe, int kblocks, int tchunks) {
  constexpr int TT = 32, LDD = 33, LDP = 129;
  __shared__ float sD[TT * LDD], sP[TT * LDP];
  extern __shared__ int sTok[];  // [BN]: element offset of every token's brick in x (the per-load divisions by the
                                 // token grid cost more VALU time than the MFMAs: 25 runtime divisions per tile)
  const int m = blockIdx.z, BN = d.B * d.N, DM = d.DM;
  const int tchunk = (int)blockIdx.x / kblocks;
  const int cb = blockIdx.y * 32, kb = ((int)blockIdx.x - tchunk * kblocks) * 128;
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
  // this workgroup's token range (whole tiles of TT)
  const int tper = ((BN + tchunks - 1) / tchunks + TT - 1) / TT * TT;
  const int tb = tchunk * tper, te = min(BN, tb + tper);
  if (tb >= te) return;
  load_tile_regs(tb);
  for (int t0 = tb; t0 < te; t0 += TT) {
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
    if (t0 + TT < te) load_tile_regs(t0 + TT);
    for (int t2 = 0; t2 < TT / 2; t2++) {
      float av = sD[(2 * t2 + h) * LDD + r];
      float bv = sP[(2 * t2 + h) * LDP + wave * 32 + r];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; i++) {
    int c = cb + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (c < DM) {
      float* q = dwpe + (int64_t)m * d.mstride + (int64_t)c * 4096 + kb + wave * 32 + r;
      if (tchunks > 1)
        atomicAdd(q, acc[i]);
      else
        *q += acc[i];
    }
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
                       const float* pos, float* F, hipStream_t st, int lp, int kd) {
  HDF_CHECK_ARG(kd == 16 || (kd == 1 && D == 1), "patch_embed: kernel depth %d on an input of depth %d", kd, D);
  // token dim = 4 * n_filters with n_filters % 16 == 0 (plan): a multiple of 64
  HDF_CHECK_ARG(d.DM <= 256 && d.DM % 64 == 0, "patch_embed: token dim %d unsupported (a multiple of 64, <= 256)", d.DM);
  const dim3 grid(ceil_div(d.B * d.N, 32), d.DM / 64, d.M);
  if (lp == HDF_BF16)
    hipLaunchKernelGGL(patch_embed_fwd2_kernel<1>, grid, dim3(256), 0, st, d, x, D, H, W, wpe, bpe, pos, F, kd);
  else if (lp == HDF_F16)
    hipLaunchKernelGGL(patch_embed_fwd2_kernel<2>, grid, dim3(256), 0, st, d, x, D, H, W, wpe, bpe, pos, F, kd);
  else
    hipLaunchKernelGGL(patch_embed_fwd2_kernel<0>, grid, dim3(256), 0, st, d, x, D, H, W, wpe, bpe, pos, F, kd);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int tf_patch_embed_bwd(const TfDims& d, const float* x, int D, int H, int W, const float* dF, float* dwpe, float* dbpe,
                       float* dpos, float* scratch, hipStream_t st, int kd) {
  HDF_CHECK_ARG(kd == 16 || (kd == 1 && D == 1), "patch_embed: kernel depth %d on an input of depth %d", kd, D);
  hipLaunchKernelGGL(patch_embed_bwd_prep_kernel, dim3(ceil_div(d.N * d.DM, 256), d.M), dim3(256), 0, st, d, dF,
                     scratch, dbpe, dpos);
  HDF_LAUNCH_CHECK();
  HDF_CHECK_ARG((int64_t)d.B * d.M * D * H * W < ((int64_t)1 << 31), "patch_embed: volume exceeds 32-bit element offsets");
  const int kblocks = kd * 256 / 128;
  // enough workgroups for the chip: the 3-D form has 32 K blocks x DM/32 x M; the depth-1 form cuts the tokens instead
  const int tchunks = kd == 16 ? 1 : std::max(1, std::min(ceil_div(d.B * d.N, 256), 512 / std::max(1, kblocks * ceil_div(d.DM, 32) * d.M)));
  hipLaunchKernelGGL(patch_embed_wgrad_kernel, dim3(kblocks * tchunks, ceil_div(d.DM, 32), d.M), dim3(256),
                     (size_t)d.B * d.N * sizeof(int), st, d, x, D, H, W, scratch, dwpe, kblocks, tchunks);
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
                     hipStream_t st, int lp) {
  if (lp == HDF_BF16 || lp == HDF_F16) {
    const size_t shm = attn_bwd_lp_lds(N);
    HDF_CHECK_ARG(N >= 1 && shm <= LDS_LIMIT, "attention backward: %d tokens need %zu B of LDS", N, shm);
    const dim3 grid(ceil_div(N, AQ), 8, 2 * nseq);
    if (lp == HDF_BF16) {
      HDF_TRY(allow_lds(attn_bwd_lp_kernel<1>, shm));
      hipLaunchKernelGGL(attn_bwd_lp_kernel<1>, grid, dim3(256), shm, st, N, nseq, qkv, ob, lse, dO, dqkv);
    } else {
      HDF_TRY(allow_lds(attn_bwd_lp_kernel<2>, shm));
      hipLaunchKernelGGL(attn_bwd_lp_kernel<2>, grid, dim3(256), shm, st, N, nseq, qkv, ob, lse, dO, dqkv);
    }
    HDF_LAUNCH_CHECK();
    return HDF_OK;
  }
  const size_t shm = (size_t)attn_rows(N) * 40;
  HDF_CHECK_ARG(N >= 1 && shm <= LDS_LIMIT, "attention backward: %d tokens need %zu B of LDS", N, shm);
  HDF_TRY(allow_lds(attn_bwd_kernel, shm));
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(ceil_div(N, AQ), 8, 2 * nseq), dim3(256), shm, st, N, nseq, qkv, ob, lse, dO,
                     dqkv);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int tf_layer_fwd(const TfDims& d, int block, int layer, const TfLayerP& p, float* F, const TfLayerSave& s,
                 hipStream_t st) {
  // one dense layer on its own (operator-level ABI): PRE, attention, POST as three launches of the fused token kernels
  TfTokenFwd pre;
  pre.pre = &p, pre.pre_save = s, pre.bq = block, pre.lq = layer, pre.F_pre = F;
  HDF_TRY(tf_token_fwd(d, pre, HDF_F32, st));
  HDF_TRY(tf_attention_fwd(d.N, d.M * d.B, s.qkv, s.ob, s.lse, st));
  TfTokenFwd post;
  post.post = &p, post.post_save = s, post.bp = block, post.lp = layer, post.F_post = F;
  return tf_token_fwd(d, post, HDF_F32, st);
}

int tf_layer_bwd(const TfDims& d, int block, int layer, const TfLayerP& p, const TfLayerP& g, const float* F, float* dF,
                 const TfLayerSave& s, float* scratch, hipStream_t st) {
  const int BN = d.B * d.N;
  const int64_t rows = (int64_t)d.M * BN;
  float* dO = scratch;
  float* dh0acc = scratch + rows * 32;
  float* dqkv = scratch + rows * 64;
  // one dense layer on its own: POSTB, attention backward, PREB
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

int tf_block_out_fwd(const TfDims& d, int block, const TfOutP& p, const float* F, float* next_F, void* attnall,
                     int dtype, hipStream_t st) {
  TfTokenFwd o;
  o.out = &p, o.bp = block, o.F_post = const_cast<float*>(F), o.next_F = next_F, o.attnall = attnall;
  return tf_token_fwd(d, o, dtype, st);
}

int tf_block_out_bwd(const TfDims& d, int block, const TfOutP& p, const TfOutP& g, const float* F,
                     const float* dF_next, const void* d_attnall, int dtype, float* dF, hipStream_t st) {
  TfTokenBwd o;
  o.dF = dF, o.out = &p, o.out_grad = &g, o.bo = block, o.F_out = F, o.dF_next = dF_next, o.d_attnall = d_attnall;
  return tf_token_bwd(d, o, dtype, st);
}
