// Multi-path densely-connected Transformer branch (HDenseFormer.py:33-145), fp32 arithmetic.
// Row order of every token buffer: row = (m*B + b)*N + n  (modality outermost, so one workgroup only
// ever touches one modality's weights).  Per-modality parameters are addressed as base + m*mstride
// (the branches have identical structure, so their parameter blocks are equally spaced in the flat
// parameter buffer).
#pragma once
#include "hdf_common.h"

// delta of the attention backward: (dO . ob) over the 4 components of a head.  One explicit fma chain under contract(off):
// as a plain sum of four products its rounding depended on how the surrounding code was vectorised, and the persistent
// backward kernel (transformer_chain.hip) must reproduce the attention backward kernels bit for bit.
__device__ __forceinline__ float tf_dot4(const float4& a, const float4& b) {
#pragma clang fp contract(off)
  return __builtin_fmaf(a.w, b.w, __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)));
}

struct TfDims {
  int M;    // modalities (= in_channels)
  int B;    // batch
  int N;    // tokens per sample
  int DM;   // token dim = 4*n_filters
  int DMF;  // DM + 128: width of the dense feature buffer
  int64_t mstride;  // floats between the parameter blocks of consecutive modalities
  // dropout
  int training;
  uint32_t seed;
  uint32_t thresh24;  // round((1-p)*2^24)
  float keep_scale;   // 1/(1-p)
};

struct TfLayerP {  // pointers into modality 0's block (parameters or, for backward, gradients)
  float *w0, *b0, *ln1g, *ln1b, *wqkv, *wout, *bout, *ln2g, *ln2b, *w1, *b1, *w2, *b2;
};
struct TfOutP {
  float *wa, *ba, *wb, *bb;
};

// saved per layer (forward -> backward): all [rows][..] fp32
struct TfLayerSave {
  float* h0;   // [rows][32]
  float* qkv;  // [rows][96]
  float* ob;   // [rows][32] attention output before to_out
  float* lse;  // [rows][8]
  float* h1;   // [rows][32]
  float* h2;   // [rows][32]
};

int tf_patch_embed_fwd(const TfDims& d, const float* x /*[B][M][D][H][W]*/, int D, int H, int W, const float* wpe,
                       const float* bpe, const float* pos, float* F /*[rows][DMF]*/, hipStream_t st, int lp = 0,
                       int kd = 16 /* depth slices of the 16^3 kernel contracted: 1 = the 2-D model's 16 x 16 patches on a depth-1 input */);
int tf_patch_embed_bwd(const TfDims& d, const float* x, int D, int H, int W, const float* dF /*[rows][DMF]*/,
                       float* dwpe, float* dbpe, float* dpos, float* scratch /*[rows][DM]*/, hipStream_t st, int kd = 16);

// Dense_Attention core (HDenseFormer.py:67-74): qkv [nseq*N][96] (q | k | v, 8 heads x 4) -> ob [nseq*N][32] (heads
// merged, before to_out), lse [nseq*N][8] (natural-log row log-sum-exp of the 0.5-scaled scores)
int tf_attention_fwd(int N, int nseq, const float* qkv, float* ob, float* lse, hipStream_t st);
// dO [nseq*N][32] -> dqkv [nseq*N][96]
// lp: 0 = everything fp32 (fp32 storage); HDF_BF16 / HDF_F16 = the accumulations dS.k, dS^T.q, attn^T.dO with 16-bit operands
// on the matrix core (the plan's 16-bit storage modes: what torch autocast does to the matmuls of Dense_Attention)
int tf_attention_bwd(int N, int nseq, const float* qkv, const float* ob, const float* lse, const float* dO, float* dqkv,
                     hipStream_t st, int lp = 0);

int tf_layer_fwd(const TfDims& d, int block, int layer, const TfLayerP& p, float* F, const TfLayerSave& s,
                 hipStream_t st);
int tf_layer_bwd(const TfDims& d, int block, int layer, const TfLayerP& p, const TfLayerP& g, const float* F, float* dF,
                 const TfLayerSave& s, float* scratch /*[rows][32+32+96]*/, hipStream_t st);

// out_layer of a block.  next_F != null: write fp32 into next block's feature buffer columns [0,DM);
// else write storage-typed rows into the channels-last attnall buffer [B][N][M*DM].
int tf_block_out_fwd(const TfDims& d, int block, const TfOutP& p, const float* F, float* next_F, void* attnall,
                     int dtype, hipStream_t st);
// dout comes from dF_next[rows][0:DM] (fp32, pitch DMF) or from d_attnall (storage type).  Writes dF[rows][0:DMF].
int tf_block_out_bwd(const TfDims& d, int block, const TfOutP& p, const TfOutP& g, const float* F,
                     const float* dF_next, const void* d_attnall, int dtype, float* dF, hipStream_t st);

// ---- fused token kernels (transformer_fused.hip): everything between two attention launches in ONE launch -----------
// Forward: [finish dense layer (bp, lp): to_out + residual + ff + ff] -> [block bp's out_layer] -> [start layer (bq, lq):
// Linear0 + LN1 + to_qkv], any non-empty contiguous selection of the three stages (null pointer = stage absent).
struct TfTokenFwd {
  const TfLayerP* post = nullptr;  // layer (bp, lp); reads post_save.h0 / .ob, writes .h1 / .h2 and the feature column
  TfLayerSave post_save{};
  int bp = 0, lp = 0;
  float* F_post = nullptr;         // block bp's feature buffer [rows][DMF] (POST writes it, OUT reads it)
  const TfOutP* out = nullptr;     // block bp's out_layer
  float* next_F = nullptr;         // -> next block's feature buffer columns [0, DM) ...
  void* attnall = nullptr;         // ... or the channels-last attnall tensor in the storage dtype (last block)
  const TfLayerP* pre = nullptr;   // layer (bq, lq); writes pre_save.h0 / .qkv
  TfLayerSave pre_save{};
  int bq = 0, lq = 0;
  const float* F_pre = nullptr;    // feature buffer PRE reads when it runs alone (else next_F / F_post)
};
int tf_token_fwd(const TfDims& d, const TfTokenFwd& t, int dtype, hipStream_t st);
// Backward, in execution order: [PREB: Linear0 / LN1 / to_qkv backward of layer (bq, lq), after its attention backward]
// -> [OUTB: out_layer backward of block bo] -> [POSTB: ff / ff / to_out backward of layer (bp, lp), before its attention
// backward].  PREB + POSTB: consecutive layers of one block; PREB + OUTB (+ POSTB): block boundary (lq = 0, bo = bq - 1,
// POSTB = layer 3 of block bo).  Parameter gradients are accumulated with fp32 atomics.
// Tape of a dense layer's weight-gradient operands, TF_TAPE_W floats per token (written by the token kernels,
// contracted over all tokens by tf_wgrad).  Segment-major: segment c of width w is the dense array [rows][w] at float
// offset rows * c.  Segments:
constexpr int TF_TAPE_W = 576;
constexpr int TF_T_DQ = 0;      // [96] dqkv                         (x t       -> to_qkv.weight)
constexpr int TF_T_T = 96;      // [32] t = LN1(h0)
constexpr int TF_T_DH0 = 128;   // [32] dh0                          (x F[:, :K] -> Linear0.weight)
constexpr int TF_T_P1 = 160;    // second ff (on h2): dg [32] | f [64] | dz [64] | u [32]   (dg x f -> net.3, dz x u -> net.0)
constexpr int TF_T_P0 = 352;    // first ff (on h1):  same layout
constexpr int TF_T_DGO = 544;   // [32] masked gradient of the to_out output   (x ob -> to_out.weight)
// a block's out_layer tape (DM + 128 floats per token, segment-major): do [DM] at 0 | f [64] at DM | dz [64] at DM + 64
// (do x f -> net.3, dz x F -> net.0)

constexpr int TF_WG_ENTRIES = 22;  // 4 layers x 5 matrices + 2 out_layer matrices per block
struct TfWgradEntry {
  int64_t poff;        // floats from the block's first parameter to this matrix
  int O, I, layer;     // O = 0: entry switched off
  int ysrc, xsrc;      // operand sources: 0 layer tape, 1 block feature buffer, 2 saved ob, 3 block out tape
  int y0, x0, y1, x1;  // segment (sources 0 / 3) or column (1 / 2) of the (first, second) operand pair; y1 < 0: one pair
  int yld, xld;        // row pitch of the operands: the segment width (0 / 3), DMF (1), 32 (2)
};
struct TfWgradArgs {
  TfWgradEntry e[TF_WG_ENTRIES];
  float* grads;                    // modality 0's gradient block base is grads + block0
  int64_t mstride, block0, block_stride;
  const float* tape;               // [nb*4][rows][TF_TAPE_W], indexed from block b0
  const float* otape;              // [nb][rows][DMF]
  const float* F;                  // [nb][rows][DMF]
  const float* save;               // [nb*4][rows][232]
  int64_t rows;                    // M * B * N
  int BN, DMF, b0;
  int kchunks = 1, ntiles = 0;     // set by tf_wgrad: token chunks per tile (> 1: partial products added with fp32 atomics)
};
// all weight-matrix gradients of blocks [b0, b0 + nblocks) in one launch.  Up to 4096 tokens per modality a workgroup
// contracts ALL tokens of its 32 x 32 tile and overwrites the gradient (fixed order, bitwise reproducible); beyond that
// (round 6: the 2-D model at batch 24 has 13,824) the token range is cut into chunks of ~2048 over more workgroups, which
// ADD into the zeroed gradient buffer with float atomics -- one workgroup per tile walked 13.5x the tokens of the 3-D
// benchmark with the same four waves (806 us per launch at 0.3 TB/s).
int tf_wgrad(const TfWgradArgs& a, int nblocks, int M, hipStream_t st);

struct TfTokenBwd {
  float* tape_pre = nullptr;        // layer (bq, lq)'s tape / layer (bp, lp)'s tape / block bo's out tape: when set the
  float* tape_post = nullptr;       // weight-matrix gradients are left to tf_wgrad (biases and LayerNorm parameters are
  float* tape_out = nullptr;        // still accumulated here)
  float* dF = nullptr;              // [rows][DMF] gradient of the (current block's) feature buffer
  const TfLayerP* pre = nullptr;    // layer (bq, lq)
  const TfLayerP* pre_grad = nullptr;
  TfLayerSave pre_save{};
  int bq = 0, lq = 0;
  const float* F_pre = nullptr;     // block bq's feature buffer
  const float* dqkv = nullptr;      // [rows][96] from the attention backward
  const float* dh0acc = nullptr;    // [rows][32] residual-path gradient left by this layer's POSTB
  const TfOutP* out = nullptr;      // out_layer of block bo
  const TfOutP* out_grad = nullptr;
  int bo = 0;
  const float* F_out = nullptr;     // block bo's feature buffer
  const float* dF_next = nullptr;   // OUTB without PREB: gradient of the next block's input (fp32 [rows][DMF]) or
  const void* d_attnall = nullptr;  // the attnall gradient (storage dtype)
  const TfLayerP* post = nullptr;   // layer (bp, lp)
  const TfLayerP* post_grad = nullptr;
  TfLayerSave post_save{};
  int bp = 0, lp = 0;
  float* dO = nullptr;              // [rows][32] -> attention backward of layer (bp, lp)
  float* dh0acc_out = nullptr;      // [rows][32] -> PREB of layer (bp, lp)
};
int tf_token_bwd(const TfDims& d, const TfTokenBwd& t, int dtype, hipStream_t st);

// ---- persistent chain kernels (transformer_chain.hip): every dense layer of every block in ONE launch per direction.
// Parameter addressing of a branch: tensor k of layer l of block b sits at blk0 + b * blk_stride + loff[l][k] floats from the
// flat parameter (or gradient) buffer's base, + m * mstride for modality m.  loff order = the members of TfLayerP;
// ooff = the block's out_layer (wa, ba, wb, bb).
struct TfChainP {
  int32_t loff[4][13];
  int32_t ooff[4];
  int64_t blk0, blk_stride;
};
// the launch takes the shape (one workgroup per 16 tokens of a sequence, all resident at once: tiles <= hdf_cu_budget(),
// which is capped by the device's compute-unit count)
bool tf_chain_supported(const TfDims& d);
bool tf_chain_shape_ok(const TfDims& d);   // the shape alone (hdf_plan_force_persistent: tests of the give-up path)
// What a persistent launch does when a per-sequence barrier is not completed in time (the grid was not resident together:
// the device is shared).  `host_flag`: device address of a host-mapped word (the plan's; null for operator-level calls)
// that receives 1 + the id of the workgroup that gave up; `ticks`: deadline of one wait in 100 MHz real-time ticks.
// The launch then ends by itself (no trap) with NaN in the rows of the workgroups that gave up: see chain_wait.
struct TfChainCtl {
  unsigned* host_flag = nullptr;
  unsigned ticks = 150000000u;   // 1.5 s
};
size_t tf_chain_sync_bytes(const TfDims& d);
size_t tf_chain_wpack_bytes(const TfDims& d, int nb);
size_t tf_chain_frag_bytes(const TfDims& d, int nb);   // the forward's operand records for the backward (16-bit modes)
bool tf_chain_backward_supported(const TfDims& d, int dtype);
// fragment-major copies of every layer's and block's weight matrices for both directions (any stream: it only reads the
// parameters; tf_chain_forward / tf_chain_backward must be ordered behind it)
int tf_chain_pack(const TfDims& d, const TfChainP& cp, int nb, const float* params, void* wpack, hipStream_t st);
// Forward of all nb blocks: reads F0[block 0][:, 0:DM] (the patch embedding), writes every block's feature buffer, the
// saved tensors of every layer (tf_save layout) and the channels-last attnall tensor.  `sync`: tf_chain_sync_bytes of
// device memory owned by the caller (zeroed by the call on `st`); `wpack`: tf_chain_wpack_bytes of scratch for the
// fragment-major copies of the layers' weight matrices (tf_chain_pack).
int tf_chain_forward(const TfDims& d, const TfChainP& cp, int nb, const float* params, float* F0, float* save,
                     void* attnall, unsigned* sync, void* wpack, float* frag, int dtype, hipStream_t st,
                     const TfChainCtl& ctl = TfChainCtl{});
// Backward of all nb blocks (after the UpConv chain's backward left d(attnall)): every bias / LayerNorm-parameter gradient
// (fp32 atomics into `grads`), the weight-gradient tapes of every layer (tf_wgrad afterwards), and dF[:, 0:DM] = the
// gradient of block 0's input (tf_patch_embed_bwd afterwards).  `xchg`: 2 * rows * 40 floats of scratch (the dO | delta
// rows the workgroups of a sequence hand each other); `sync`: tf_chain_sync_bytes, zeroed by the call.
int tf_chain_backward(const TfDims& d, const TfChainP& cp, int nb, const float* params, float* grads, const float* F0,
                      const float* save, float* dF, const void* d_attnall, float* tape, float* otape, float* xchg,
                      const float* frag, const void* wpack, unsigned* sync, int dtype, hipStream_t st,
                      const TfChainCtl& ctl = TfChainCtl{});
