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
#include <algorithm>

#include "tf_tok.h"

namespace {

using namespace tftok;


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
  HDF_CHAIN_PRIO();
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
  float p_bout = 0.f, p_b1 = 0.f, p_b2 = 0.f, p_b0 = 0.f;
  LnP ln2{}, ln1{};
  const int cw = 16 * (wave & 1) + col, c4w = 16 * wave + col;   // this lane's column in a 2-tile / 4-tile product
  if (POST) {
    if (wave < 2) {
      wload(f_wo, a.pp.wout + mo, 32, 16 * wave, 8, 0, 2);
      wload(f_w2, a.pp.w2 + mo, 64, 16 * wave, 16, 0, 4);
#pragma unroll
      for (int r = 0; r < 4; r++) h0r[r] = a.h0_in[(rb + tok(4 * g + r)) * 32 + 16 * wave + col];
    }
    wload(f_w1, a.pp.w1 + mo, 32, 16 * wave, 8, 0, 2);
    p_bout = a.pp.bout[mo + cw], p_b2 = a.pp.b2[mo + cw], p_b1 = a.pp.b1[mo + c4w];
    ln2 = ln_load(a.pp.ln2g + mo, a.pp.ln2b + mo);
  }
  if (PRE) {
    wload(f_w0, a.pq.w0 + mo, Kq, 16 * (wave & 1), Kq >> 2, (wave >> 1) * (Kq >> 3), Kq >> 5);
    wload(f_q0, a.pq.wqkv + mo, 32, 16 * wave, 8, 0, 2);
    if (wave < 2) wload(f_q1, a.pq.wqkv + mo, 32, 16 * (wave + 4), 8, 0, 2);
    p_b0 = a.pq.b0[mo + cw];
    ln1 = ln_load(a.pq.ln1g + mo, a.pq.ln1b + mo);
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
      const float bo = p_bout;
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
      ln32(s_h, s_x, ln2);
      __syncthreads();
      {
        f32x4 acc = zero4();
        wmma(acc, f_w1, s_x, LD32, 8, 0, 2);
        const int c = 16 * wave + col;
        const float b1 = p_b1;
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
        const float b2 = p_b2;
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
        const float b0 = p_b0;
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
    ln32(s_h, s_x, ln1);
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

// ================================================================================================ backward

struct TokBwd {
  TfDims d;
  // weight-gradient operands go to a tape (tf_wgrad: one batched product per weight matrix over ALL tokens) instead of
  // per-tile products + fp32 atomics when these are set
  float* tape_q;   // layer (bq, lq): [rows][TF_TAPE_W]
  float* tape_p;   // layer (bp, lp)
  float* otape;    // block bo: [rows][DM + 128]
  // PREB: layer lq of block bq (after its attention backward)
  TfLayerP pq, gq;
  int bq, lq;
  const float* Fq;       // block bq's feature buffer
  const float* h0;       // saved
  const float* dqkv;     // from the attention backward
  const float* dh0acc;   // residual-path gradient left by this layer's POSTB
  float* dF;             // [rows][DMF] gradient of the feature buffer (shared by all blocks)
  // OUTB: out_layer of block bo
  TfOutP po, go;
  int bo;
  const float* Fo;       // block bo's feature buffer
  const float* dF_next;  // gradient of the next block's input (fp32, pitch DMF) when OUTB runs without PREB, or
  const void* d_attnall; // the attnall gradient (storage dtype) for the last block
  // POSTB: layer lp of block bp
  TfLayerP pp, gp;
  int bp, lp;
  const float* h1s;
  const float* h2s;
  const float* ob;
  float* dO;             // -> attention backward of layer lp
  float* dh0acc_out;     // -> PREB of layer lp
};

template <bool PREB, bool OUTB, bool POSTB, typename T>
__global__ __launch_bounds__(256) void tok_bwd_kernel(TokBwd a) {
  HDF_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const TfDims& d = a.d;
  // (the inner-layer form stages only the Kq <= DMF - 32 feature columns Linear0 of layers 0..3 reads: 2 KB less LDS)
  const int DM = d.DM, DMF = d.DMF, ldF = (OUTB ? DMF : DMF - 32) + 4, ldD = DM + 4;
  float* s_F = sm;                      // [16][ldF]  feature rows (PREB: block bq; OUTB: block bo)
  float* s_do = s_F + TT * ldF;         // [16][ldD]  OUTB: masked gradient of the out_layer output (no room otherwise:
  float* s_dq = s_do + (OUTB ? TT * ldD : 0);  // [16][100]  PREB: dqkv tile      see launch_tok_bwd)
  float* s_a = s_dq + TT * 100;         // [16][36] x 6 small tiles
  float* s_b = s_a + TT * LD32;
  float* s_c = s_b + TT * LD32;
  float* s_e = s_c + TT * LD32;
  float* s_dg = s_e + TT * LD32;        // gradient of the feature column handed from PREB / OUTB to POSTB
  float* s_gx = s_dg + TT * LD32;
  float* s_f = s_gx + TT * LD32;        // [16][68]
  float* s_dz = s_f + TT * LD64;        // [16][68]
  float* s_red = s_dz + TT * LD64;      // [2][16][16]
  const int m = blockIdx.y, BN = d.B * d.N, t0 = blockIdx.x * TT;
  const int64_t mo = (int64_t)m * d.mstride, rb = (int64_t)m * BN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 15, g = lane >> 4;
  const int lrow = threadIdx.x >> 4, lc = (threadIdx.x & 15) * 2;   // the "LayerNorm" thread map: row, 2 columns
  const DropF dr{d.training, d.seed, d.thresh24, d.keep_scale};
  auto tok = [&](int row) { return min(t0 + row, BN - 1); };
  const int64_t trows = (int64_t)d.M * BN;   // rows of a tape segment

  // ------------------------------------------------------------------ requests issued at kernel entry
  const int Kq = PREB ? DM + 32 * a.lq : 0;
  CFrag<12> c_q;            // dt = dqkv * Wqkv: tile wave & 1, o half wave >> 1 (24 of 96 / 4 ... 12 per half)
  CFrag<8> c_w0[6];         // dF += dh0 * W0: tiles wave + 4j (Kq <= 352: 22 tiles)
  WFrag<2> f_w1;            // POSTB: ff net.0 rows (recompute)
  CFrag<8> c_w2, c_w1;      // df = dg * W2 (tile wave) ; du = dz * W1 (tile wave & 1, o half wave >> 1)
  CFrag<4> c_wo;            // dO = dgo * Wout (tile wave & 1, o half)
  LnP ln1{}, ln2{};
  float p_b1 = 0.f, r_acc0 = 0.f, r_acc1 = 0.f, r_h[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, r_ob[2] = {0.f, 0.f};
  float r_dF[6][4];        // the tile's dF entries this lane read-modify-writes (only this workgroup touches them)
  const bool lok = t0 + lrow < BN;
  const int64_t lr = (rb + tok(lrow)) * 32 + lc;   // this thread's two columns of a [rows][32] tensor
  if (PREB) {
    cload(c_q, a.pq.wqkv + mo, 32, 16 * (wave & 1), 32, 24, 12 * (wave >> 1), 12);
#pragma unroll
    for (int j = 0; j < 6; j++) {
      cload(c_w0[j], a.pq.w0 + mo, Kq, 16 * (wave + 4 * j), Kq, 8, 0, 8);
#pragma unroll
      for (int r = 0; r < 4; r++)
        r_dF[j][r] = a.dF[(rb + tok(4 * g + r)) * DMF + min(16 * (wave + 4 * j) + col, Kq - 1)];
    }
    ln1 = ln_load(a.pq.ln1g + mo, a.pq.ln1b + mo);
    r_acc0 = a.dh0acc[lr], r_acc1 = a.dh0acc[lr + 1];
  }
  if (POSTB) {
    wload(f_w1, a.pp.w1 + mo, 32, 16 * wave, 8, 0, 2);
    cload(c_w2, a.pp.w2 + mo, 64, 16 * wave, 64, 8, 0, 8);
    cload(c_w1, a.pp.w1 + mo, 32, 16 * (wave & 1), 32, 16, 8 * (wave >> 1), 8);
    cload(c_wo, a.pp.wout + mo, 32, 16 * (wave & 1), 32, 8, 4 * (wave >> 1), 4);
    ln2 = ln_load(a.pp.ln2g + mo, a.pp.ln2b + mo);
    p_b1 = a.pp.b1[mo + 16 * wave + col];
    r_h[0][0] = a.h1s[lr], r_h[0][1] = a.h1s[lr + 1], r_h[1][0] = a.h2s[lr], r_h[1][1] = a.h2s[lr + 1];
    r_ob[0] = a.ob[lr], r_ob[1] = a.ob[lr + 1];
  }
  if (PREB) {
    for (int i = threadIdx.x; i < TT * 24; i += 256) {      // dqkv tile (zero rows beyond BN)
      const int row = i / 24, c4 = (i - row * 24) * 4;
      float4 v = *reinterpret_cast<const float4*>(a.dqkv + (rb + tok(row)) * 96 + c4);
      if (t0 + row >= BN) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(s_dq + row * 100 + c4) = v;
    }
    const int c4n = Kq >> 2;
    for (int i = threadIdx.x; i < TT * c4n; i += 256) {     // feature rows [0, Kq)
      const int row = i / c4n, c4 = (i - row * c4n) * 4;
      *reinterpret_cast<float4*>(s_F + row * ldF + c4) =
          *reinterpret_cast<const float4*>(a.Fq + (rb + tok(row)) * DMF + c4);
    }
    s_a[lrow * LD32 + lc] = a.h0[lr];
    s_a[lrow * LD32 + lc + 1] = a.h0[lr + 1];
  }
  __syncthreads();

  // ------------------------------------------------------------------ PREB(lq)
  if (PREB) {
    // t = LN1(h0) -> s_b, xh -> s_c
    const float rs1 = ln32_keep(s_a, s_b, s_c, ln1);
    __syncthreads();
    const int nvalid = min(TT, BN - t0);
    if (a.tape_q) {
      tape_store(a.tape_q, trows, TF_T_DQ, s_dq, 100, 96, rb + t0, nvalid);
      tape_store(a.tape_q, trows, TF_T_T, s_b, LD32, 32, rb + t0, nvalid);
    } else {
      wgrad_tiles(a.gq.wqkv + mo, 96, 32, s_dq, 100, s_b, LD32);
    }
    {  // dt = dqkv * Wqkv -> s_e
      f32x4 acc = zero4();
      cmma(acc, c_q, s_dq, 100, 24, 12 * (wave >> 1), 12);
      if (wave >= 2) {
#pragma unroll
        for (int r = 0; r < 4; r++) s_red[((wave & 1) * 16 + 4 * g + r) * 16 + col] = acc[r];
      }
      __syncthreads();
      if (wave < 2) {
#pragma unroll
        for (int r = 0; r < 4; r++)
          s_e[(4 * g + r) * LD32 + 16 * wave + col] = acc[r] + s_red[(wave * 16 + 4 * g + r) * 16 + col];
      }
    }
    __syncthreads();
    {  // LN1 backward + the residual-path gradient -> dh0 (s_a)
      float dh0v, dh1v;
      ln32_bwd(s_e, s_c, s_gx, rs1, ln1, dh0v, dh1v);
      s_a[lrow * LD32 + lc] = lok ? dh0v + r_acc0 : 0.f;
      s_a[lrow * LD32 + lc + 1] = lok ? dh1v + r_acc1 : 0.f;
    }
    __syncthreads();
    colsum_atomic(a.gq.ln1g + mo, 32, s_gx, LD32, 0);
    colsum_atomic(a.gq.ln1b + mo, 32, s_e, LD32, 64);
    colsum_atomic(a.gq.b0 + mo, 32, s_a, LD32, 128);
    if (a.tape_q)
      tape_store(a.tape_q, trows, TF_T_DH0, s_a, LD32, 32, rb + t0, nvalid);
    else
      wgrad_tiles(a.gq.w0 + mo, 32, Kq, s_a, LD32, s_F, ldF, 2);
    // dF[:, 0:Kq] += dh0 * W0 ; the last 32 columns are the complete gradient of feature lq - 1 (POSTB), the first DM
    // ones at lq = 0 the gradient of the block input (OUTB of the previous block)
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const int n0 = 16 * (wave + 4 * j);
      if (n0 < Kq) {
        f32x4 acc = zero4();
        cmma(acc, c_w0[j], s_a, LD32, 8, 0, 8);
        const int c = n0 + col;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 4 * g + r, t = t0 + row;
          const bool ok = t < BN;
          float* q = a.dF + (rb + tok(row)) * DMF + c;
          const float v = ok ? r_dF[j][r] + acc[r] : 0.f;
          if (OUTB) {
            s_do[row * ldD + c] = v;          // (lq = 0: Kq = DM) consumed below, not written back
          } else {
            if (ok) *q = v;
            if (POSTB && c >= Kq - 32) s_dg[row * LD32 + c - (Kq - 32)] = v;
          }
        }
      }
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ OUTB(bo)
  if (OUTB) {
    const uint32_t siteo = hdf_site_id(m, a.bo, 4, 0);
    {  // feature rows of block bo; the upstream gradient (masked by the out_layer's second dropout)
      const int c4n = DMF >> 2;
      for (int i = threadIdx.x; i < TT * c4n; i += 256) {
        const int row = i / c4n, c4 = (i - row * c4n) * 4;
        *reinterpret_cast<float4*>(s_F + row * ldF + c4) =
            *reinterpret_cast<const float4*>(a.Fo + (rb + tok(row)) * DMF + c4);
      }
      for (int i = threadIdx.x; i < TT * DM; i += 256) {
        const int row = i / DM, c = i - row * DM, t = t0 + row;
        float v;
        if (PREB)
          v = s_do[row * ldD + c];
        else if (a.dF_next)
          v = a.dF_next[(rb + tok(row)) * DMF + c];
        else {
          const int tt = tok(row), b = tt / d.N, n = tt - b * d.N;
          v = ST<T>::ld(reinterpret_cast<const T*>(a.d_attnall) + ((int64_t)b * d.N + n) * ((int64_t)d.M * DM) +
                        (int64_t)m * DM + c);
        }
        s_do[row * ldD + c] = t < BN ? v * dr.mask(siteo + 1, (uint32_t)t * DM + c) : 0.f;
      }
    }
    __syncthreads();
    float zr[4], mk[4];
    {  // z = Wa F + ba (recomputed), f = gelu(z) * mask -> s_f
      f32x4 acc = zero4();
      wmma_stream(acc, a.po.wa + mo, DMF, 16 * wave, s_F, ldF, DMF >> 2);
      const int c = 16 * wave + col;
      const float ba = a.po.ba[mo + c];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 4 * g + r, t = t0 + row;
        zr[r] = acc[r] + ba;
        mk[r] = dr.mask(siteo + 0, (uint32_t)t * 64 + c);
        s_f[row * LD64 + c] = t < BN ? gelu_f(zr[r]) * mk[r] : 0.f;
      }
    }
    {  // df = do * Wb ; dz = df * mask * gelu'(z) -> s_dz
      f32x4 acc = zero4();
      cmma_stream(acc, a.po.wb + mo, 64, 16 * wave, 64, s_do, ldD, DM >> 2);
      const int c = 16 * wave + col;
#pragma unroll
      for (int r = 0; r < 4; r++) s_dz[(4 * g + r) * LD64 + c] = acc[r] * mk[r] * gelu_grad_f(zr[r]);
    }
    __syncthreads();
    if (a.otape) {
      const int nv = min(TT, BN - t0);
      tape_store(a.otape, trows, 0, s_do, ldD, DM, rb + t0, nv);
      tape_store(a.otape, trows, DM, s_f, LD64, 64, rb + t0, nv);
      tape_store(a.otape, trows, DM + 64, s_dz, LD64, 64, rb + t0, nv);
    } else {
      wgrad_tiles(a.go.wb + mo, DM, 64, s_do, ldD, s_f, LD64);
      wgrad_tiles(a.go.wa + mo, 64, DMF, s_dz, LD64, s_F, ldF, 1);
    }
    for (int c0 = 0; c0 < DM; c0 += 128) colsum_atomic(a.go.bb + mo + c0, min(128, DM - c0), s_do + c0, ldD, 0);
    colsum_atomic(a.go.ba + mo, 64, s_dz, LD64, 128);
    // dF[:, 0:DMF] = dz * Wa  (overwrites: the first writer of block bo's feature gradient)
    for (int n0 = 16 * wave; n0 < DMF; n0 += 64) {
      f32x4 acc = zero4();
      cmma_stream(acc, a.po.wa + mo, DMF, n0, DMF, s_dz, LD64, 16);
      const int c = n0 + col;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 4 * g + r, t = t0 + row;
        if (t < BN) a.dF[(rb + t) * DMF + c] = acc[r];
        if (POSTB && c >= DMF - 32) s_dg[row * LD32 + c - (DMF - 32)] = t < BN ? acc[r] : 0.f;
      }
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ POSTB(lp)
  if (POSTB) {
    const uint32_t site0 = hdf_site_id(m, a.bp, a.lp, 0);
    const int t = t0 + lrow;
    const bool ok = t < BN;
    if (!PREB && !OUTB) {  // stand-alone: the feature gradient comes from memory
      const int fc = DM + 32 * a.lp;
      s_dg[lrow * LD32 + lc] = ok ? a.dF[(rb + t) * DMF + fc + lc] : 0.f;
      s_dg[lrow * LD32 + lc + 1] = ok ? a.dF[(rb + t) * DMF + fc + lc + 1] : 0.f;
      __syncthreads();
    }
    float dcur0 = s_dg[lrow * LD32 + lc], dcur1 = s_dg[lrow * LD32 + lc + 1];  // gradient into the current ff's output
    float dres0 = 0.f, dres1 = 0.f;
#pragma unroll
    for (int pass = 1; pass >= 0; pass--) {  // pass 1: the second ff (on h2) ; pass 0: the first ff (on h1)
      __syncthreads();
      s_a[lrow * LD32 + lc] = r_h[pass][0];
      s_a[lrow * LD32 + lc + 1] = r_h[pass][1];
      // a thread reads back only the two values it wrote: no barrier needed before the LN
      const float rs = ln32_keep(s_a, s_b, s_c, ln2);   // u -> s_b, xh -> s_c
      s_dg[lrow * LD32 + lc] = ok ? dcur0 * dr.mask(site0 + 2 + 2 * pass, (uint32_t)t * 32 + lc) : 0.f;
      s_dg[lrow * LD32 + lc + 1] = ok ? dcur1 * dr.mask(site0 + 2 + 2 * pass, (uint32_t)t * 32 + lc + 1) : 0.f;
      __syncthreads();
      {
        f32x4 accz = zero4(), accd = zero4();
        wmma(accz, f_w1, s_b, LD32, 8, 0, 2);            // z = W1 u  (tile wave)
        cmma(accd, c_w2, s_dg, LD32, 8, 0, 8);           // df = dg * W2 (tile wave)
        const int c = 16 * wave + col;
        const float b1 = p_b1;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 4 * g + r, tt = t0 + row;
          const float z = accz[r] + b1, mkv = dr.mask(site0 + 1 + 2 * pass, (uint32_t)tt * 64 + c);
          s_f[row * LD64 + c] = tt < BN ? gelu_f(z) * mkv : 0.f;
          s_dz[row * LD64 + c] = tt < BN ? accd[r] * mkv * gelu_grad_f(z) : 0.f;
        }
      }
      __syncthreads();
      if (a.tape_p) {
        const int nv = min(TT, BN - t0), c0 = pass ? TF_T_P1 : TF_T_P0;
        tape_store(a.tape_p, trows, c0, s_dg, LD32, 32, rb + t0, nv);
        tape_store(a.tape_p, trows, c0 + 32, s_f, LD64, 64, rb + t0, nv);
        tape_store(a.tape_p, trows, c0 + 96, s_dz, LD64, 64, rb + t0, nv);
        tape_store(a.tape_p, trows, c0 + 160, s_b, LD32, 32, rb + t0, nv);
      } else {
        wgrad_tiles(a.gp.w2 + mo, 32, 64, s_dg, LD32, s_f, LD64);
        wgrad_tiles(a.gp.w1 + mo, 64, 32, s_dz, LD64, s_b, LD32, 2);
      }
      colsum_atomic(a.gp.b2 + mo, 32, s_dg, LD32, 0);
      colsum_atomic(a.gp.b1 + mo, 64, s_dz, LD64, 64);
      {  // du = dz * W1 -> s_e
        f32x4 acc = zero4();
        cmma(acc, c_w1, s_dz, LD64, 16, 8 * (wave >> 1), 8);
        if (wave >= 2) {
#pragma unroll
          for (int r = 0; r < 4; r++) s_red[((wave & 1) * 16 + 4 * g + r) * 16 + col] = acc[r];
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
          for (int r = 0; r < 4; r++)
            s_e[(4 * g + r) * LD32 + 16 * wave + col] = acc[r] + s_red[(wave * 16 + 4 * g + r) * 16 + col];
        }
      }
      __syncthreads();
      float dh0v, dh1v;
      ln32_bwd(s_e, s_c, s_gx, rs, ln2, dh0v, dh1v);
      dh0v = ok ? dh0v : 0.f, dh1v = ok ? dh1v : 0.f;
      __syncthreads();
      colsum_atomic(a.gp.ln2g + mo, 32, s_gx, LD32, 0);
      colsum_atomic(a.gp.ln2b + mo, 32, s_e, LD32, 64);
      if (pass == 1) {
        dcur0 = dh0v, dcur1 = dh1v;   // h2 feeds only the second ff: its gradient flows into ff#1's output ...
        dres0 = dh0v, dres1 = dh1v;   // ... and into the residual h1
      } else {
        dres0 += dh0v, dres1 += dh1v;
      }
    }
    // to_out: a = (Wout ob + bout) * mask ; h1 = a + h0
    __syncthreads();
    s_dg[lrow * LD32 + lc] = ok ? dres0 * dr.mask(site0 + 0, (uint32_t)t * 32 + lc) : 0.f;
    s_dg[lrow * LD32 + lc + 1] = ok ? dres1 * dr.mask(site0 + 0, (uint32_t)t * 32 + lc + 1) : 0.f;
    s_b[lrow * LD32 + lc] = r_ob[0];
    s_b[lrow * LD32 + lc + 1] = r_ob[1];
    if (ok) {
      a.dh0acc_out[(rb + t) * 32 + lc] = dres0;
      a.dh0acc_out[(rb + t) * 32 + lc + 1] = dres1;
    }
    __syncthreads();
    if (a.tape_p)
      tape_store(a.tape_p, trows, TF_T_DGO, s_dg, LD32, 32, rb + t0, min(TT, BN - t0));
    else
      wgrad_tiles(a.gp.wout + mo, 32, 32, s_dg, LD32, s_b, LD32);
    colsum_atomic(a.gp.bout + mo, 32, s_dg, LD32, 0);
    {
      f32x4 acc = zero4();
      cmma(acc, c_wo, s_dg, LD32, 8, 4 * (wave >> 1), 4);
      if (wave >= 2) {
#pragma unroll
        for (int r = 0; r < 4; r++) s_red[((wave & 1) * 16 + 4 * g + r) * 16 + col] = acc[r];
      }
      __syncthreads();
      if (wave < 2) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int tt = t0 + 4 * g + r;
          if (tt < BN) a.dO[(rb + tt) * 32 + 16 * wave + col] = acc[r] + s_red[(wave * 16 + 4 * g + r) * 16 + col];
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

// ------------------------------------------------------------------------------------------------ tf_wgrad
// Every weight-matrix gradient of the transformer branches as ONE launch over the tapes: gW[o][i] = sum over ALL tokens
// of a modality of dY[t][o] * X[t][i] (+ a second operand pair for the ff weights, which are used twice per layer).
// A workgroup owns one 32 x 32 tile of one matrix; its four waves split the token range and contract with
// v_mfma_f32_32x32x2_f32 straight from global memory (a half-wave reads 128 contiguous bytes of a tape row per step),
// then reduce through LDS in a fixed order: no atomics, bitwise reproducible.
__global__ __launch_bounds__(256) void tf_wgrad_kernel(TfWgradArgs a) {
  __shared__ float red[3][32][33];
  // grid (tiles of one block and modality, blocks, modalities): no empty workgroups, and consecutive workgroup ids --
  // which the dispatcher deals round-robin to the 8 XCDs -- are different tiles.  (The first version used
  // grid.x = 16 tile slots per matrix: slot x always landed on XCD x mod 8, XCD 0 received 7.7x the average work and
  // the launch took 360 us with the matrix pipes 11 % busy.)
  const int b = blockIdx.y, m = blockIdx.z;
  const int chunk = (int)blockIdx.x / a.ntiles;      // token chunk (tf_wgrad: one chunk up to 4096 tokens per modality)
  int en = 0, tile = (int)blockIdx.x - chunk * a.ntiles;
  for (; en < TF_WG_ENTRIES - 1; en++) {
    const int nt = (a.e[en].O >> 5) * (a.e[en].I >> 5);
    if (tile < nt) break;
    tile -= nt;
  }
  const TfWgradEntry& e = a.e[en];
  const int ti = e.I >> 5;
  const int to = tile / ti, o0 = to * 32, i0 = (tile - to * ti) * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)m * a.BN;
  const int bb = a.b0 + b;
  // operand sources: 0 a segment of the layer's tape, 1 the block's feature buffer, 2 the layer's saved ob, 3 a segment of
  // the block's out tape; (off, ld) = (segment column, segment width) for 0 / 3, (column, row pitch) for 1 / 2
  auto base = [&](int src, int off) -> const float* {
    switch (src) {
      case 0: return a.tape + ((int64_t)(bb * 4 + e.layer) * TF_TAPE_W + off) * a.rows;
      case 1: return a.F + ((int64_t)bb * a.rows) * a.DMF + off;
      case 2: return a.save + ((int64_t)(bb * 4 + e.layer) * a.rows) * 232 + a.rows * 128 + off;
      default: return a.otape + ((int64_t)bb * a.DMF + off) * a.rows;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  const int cper = (a.BN + a.kchunks - 1) / a.kchunks, c_lo = chunk * cper, c_hi = min(a.BN, c_lo + cper);
  const int per = (c_hi - c_lo + 3) / 4, t_lo = c_lo + wave * per, t_hi = min(c_hi, t_lo + per);
#pragma unroll
  for (int pair = 0; pair < 2; pair++) {
    if (pair == 1 && e.y1 < 0) break;
    const int ldy = e.yld, ldx = e.xld;
    const float* py = base(e.ysrc, pair ? e.y1 : e.y0) + row0 * ldy + o0 + r;
    const float* px = base(e.xsrc, pair ? e.x1 : e.x0) + row0 * ldx + i0 + r;
    constexpr int UN = 8;  // 16 loads per lane per batch; the next batch is requested before this batch's MFMAs
    float ya[UN], xa[UN], yn[UN], xn[UN];
    auto load = [&](float (&y)[UN], float (&x)[UN], int t) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < UN; u++) {
        const int tt = min(t + 2 * u + h, a.BN - 1);
        y[u] = py[(int64_t)tt * ldy];
        x[u] = px[(int64_t)tt * ldx];
      }
    };
    load(ya, xa, t_lo);
    for (int t = t_lo; t < t_hi; t += 2 * UN) {
      load(yn, xn, t + 2 * UN);
#pragma unroll
      for (int u = 0; u < UN; u++) {
        const bool live = t + 2 * u + h < t_hi;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(live ? ya[u] : 0.f, xa[u], acc, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < UN; u++) ya[u] = yn[u], xa[u] = xn[u];
    }
  }
  // fixed-order reduction over the four waves; accumulator i of a lane: row (i & 3) + 8 (i >> 2) + 4 h, column r
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 16; i++) red[wave - 1][(i & 3) + 8 * (i >> 2) + 4 * h][r] = acc[i];
  }
  __syncthreads();
  if (wave == 0) {
    float* gW = a.grads + (int64_t)m * a.mstride + a.block0 + (int64_t)bb * a.block_stride + e.poff;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      const float v = ((acc[i] + red[0][row][r]) + red[1][row][r]) + red[2][row][r];
      if (a.kchunks > 1)
        atomicAdd(&gW[(int64_t)(o0 + row) * e.I + i0 + r], v);   // (the backward zeroes the gradient buffer first)
      else
        gW[(int64_t)(o0 + row) * e.I + i0 + r] = v;
    }
  }
}

template <bool PREB, bool OUTB, bool POSTB, typename T>
int launch_tok_bwd(const TokBwd& a, hipStream_t st) {
  const TfDims& d = a.d;
  dim3 grid(ceil_div(d.B * d.N, TT), d.M);
  // The inner-layer form (no out_layer stage) leaves out that stage's [16][DM + 4] tile: 49.9 KB instead of 58.4 KB at
  // DM = 128, and with that it fits beside a level-0 weight-gradient workgroup of the side stream (110 KB of LDS, 304 of
  // the 512 registers per lane against this kernel's 208): the transformer backward then advances next to it instead
  // of queueing behind it.
  // ... and (round 4) only the DMF - 32 feature columns it can read: 47.9 KB, which also fits beside the 113 KB workgroups
  // of the level-0 data-gradient conv (conv_ws2<32,64>), i.e. on every CU during the level-0 encoder backward.
  const size_t shm = (size_t)(TT * ((OUTB ? d.DMF : d.DMF - 32) + 4) + (OUTB ? TT * (d.DM + 4) : 0) + TT * 100 +
                              7 * TT * LD32 + 2 * TT * LD64 + 2 * 16 * 16) * sizeof(float);
  HDF_TRY(allow_lds_f(tok_bwd_kernel<PREB, OUTB, POSTB, T>, shm));
  hipLaunchKernelGGL((tok_bwd_kernel<PREB, OUTB, POSTB, T>), grid, dim3(256), shm, st, a);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

}  // namespace

int tf_wgrad(const TfWgradArgs& a0, int nblocks, int M, hipStream_t st) {
  TfWgradArgs a = a0;
  int tiles = 0;
  for (int k = 0; k < TF_WG_ENTRIES; k++) tiles += (a.e[k].O >> 5) * (a.e[k].I >> 5);
  HDF_CHECK_ARG(tiles > 0 && a.e[TF_WG_ENTRIES - 1].O > 0, "tf_wgrad: empty entry table");
  a.ntiles = tiles;
  a.kchunks = a.BN <= 4096 ? 1 : std::min(16, (a.BN + 2047) / 2048);
  hipLaunchKernelGGL(tf_wgrad_kernel, dim3(tiles * a.kchunks, nblocks, M), dim3(256), 0, st, a);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int tf_token_bwd(const TfDims& d, const TfTokenBwd& t, int dtype, hipStream_t st) {
  HDF_CHECK_ARG(d.DM % 32 == 0 && d.DM >= 32 && d.DM <= 256, "token kernel: token dim %d unsupported", d.DM);
  TokBwd a{};
  a.d = d;
  a.dF = t.dF;
  a.tape_q = t.tape_pre, a.tape_p = t.tape_post, a.otape = t.tape_out;
  if (t.pre) {
    a.pq = *t.pre, a.gq = *t.pre_grad, a.bq = t.bq, a.lq = t.lq;
    a.Fq = t.F_pre, a.h0 = t.pre_save.h0, a.dqkv = t.dqkv, a.dh0acc = t.dh0acc;
  }
  if (t.out) {
    a.po = *t.out, a.go = *t.out_grad, a.bo = t.bo;
    a.Fo = t.F_out, a.dF_next = t.dF_next, a.d_attnall = t.d_attnall;
  }
  if (t.post) {
    a.pp = *t.post, a.gp = *t.post_grad, a.bp = t.bp, a.lp = t.lp;
    a.h1s = t.post_save.h1, a.h2s = t.post_save.h2, a.ob = t.post_save.ob;
    a.dO = t.dO, a.dh0acc_out = t.dh0acc_out;
  }
  const bool Q = t.pre != nullptr, O = t.out != nullptr, P = t.post != nullptr;
  HDF_CHECK_ARG(!(Q && O) || t.lq == 0, "token kernel: PREB + OUTB only at a block boundary (layer 0)");
  HDF_CHECK_ARG(!(Q && P) || O || (t.bp == t.bq && t.lp == t.lq - 1), "token kernel: PREB + POSTB must be consecutive layers");
#define TOKB_CASE(QQ, OO, PP)                                                  \
  if (Q == QQ && O == OO && P == PP) {                                         \
    if constexpr (OO && !QQ) {                                                 \
      if (!t.dF_next) HDF_DISPATCH_T(dtype, return (launch_tok_bwd<QQ, OO, PP, T>(a, st))); \
    }                                                                          \
    return launch_tok_bwd<QQ, OO, PP, float>(a, st);                           \
  }
  TOKB_CASE(true, false, false)
  TOKB_CASE(false, false, true)
  TOKB_CASE(true, false, true)
  TOKB_CASE(true, true, true)
  TOKB_CASE(false, true, true)
  TOKB_CASE(false, true, false)
  TOKB_CASE(true, true, false)
#undef TOKB_CASE
  hdf_set_error("token kernel: empty stage selection");
  return HDF_ERR_ARG;
}

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
