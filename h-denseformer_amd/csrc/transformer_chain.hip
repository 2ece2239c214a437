// Persistent "chain" kernels of the multi-path dense Transformer branch: ALL dense layers of all blocks of every
// (modality, sample) sequence in ONE launch per direction (round 5).
//
// Reference: models/HDenseFormer.py:78-145 (DensePreConv_AttentionBlock.forward, Dense_TransformerBlock.forward) and
// :47-75 (Dense_Attention).  The launch chain it replaces (transformer_fused.hip + the attention kernels of
// transformer.hip: token kernel, attention, token kernel, ... = 49 + 24 launches per direction) stays as the path of
// shapes this kernel does not take, and as the bit-for-bit reference of its tests: every stage below performs the SAME
// fp32 operations in the SAME order as the kernel it mirrors.
//
// Structure.  A workgroup (512 threads) owns 16 consecutive tokens of ONE sequence for the whole launch.  Everything of a
// dense layer except the attention core is local to a token: the feature rows (the growing concat of
// HDenseFormer.py:91-98), h0 and the attention output stay in LDS / registers from layer to layer, the saved tensors of
// the backward are written on the way.  The attention of layer L needs the keys and values of the whole sequence: the
// workgroups of a sequence publish their 16 rows of q|k|v (write-through stores), add to the sequence's arrival counter
// and wait until all of them have arrived -- a per-sequence barrier, one per layer; sequences never wait for each other
// (HDenseFormer.py:93-101 has no cross-sequence term).  Then each of the 8 waves takes one head: its 16 queries against
// all keys, streamed through LDS in chunks of 128 keys (double-buffered), scores on v_mfma_f32_16x16x4_f32, exponentials
// and P.V on the VALU exactly as attn_fwd_kernel.  The weights of the next token phase are requested before the wait.
//
// Hand-off protocol (cdna_hip_programming.md Guideline 16, R1 with a counter; MI355X_MICROARCH.md "Valid forms", first
// row of the sc1 table): every payload store is a 16-byte sc1 (write-through) store, every storing wave drains
// (s_waitcnt vmcnt(0)), the workgroup's barrier, ONE lane adds to the agent-scope counter; the consumer polls that word
// with relaxed agent-scope loads from ONE lane, joins the workgroup's barrier, and EVERY load of handed-off bytes is an
// sc1 buffer load to registers.  Counters are monotonic within a launch (target = tiles x (phase + 1)) and zeroed by a
// hipMemsetAsync in front of every launch.  All workgroups of the grid must be resident (grid <= compute units, checked
// by the launcher); every spin is bounded by the 100 MHz real-time counter and leaves a timeout word behind.
#include "tf_tok.h"

namespace {

using namespace tftok;

constexpr int CT = 512;     // threads per workgroup: 8 waves = the 8 heads of Dense_Attention
constexpr int KVC = 128;    // keys per staged chunk
constexpr int KVP = 68;     // floats per row of a chunk image: K (8 heads x 4) | V (8 heads x 4) | pad
constexpr int LDQ = 100;    // row pitch of the q|k|v tile
constexpr int SYNC_LINE = 32;  // unsigned words per counter (128-byte lines)
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr int ATRIP = 16;
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 lo2(const float4& v) { return f2{v.x, v.y}; }
__device__ __forceinline__ f2 hi2(const float4& v) { return f2{v.z, v.w}; }
__host__ __device__ inline int attn_rows(int N) { return (N + ATRIP - 1) / ATRIP * ATRIP; }

// ---- fragment-major weight copies.  A token-stage weight fragment (tf_tok.h: wload) is, per lane, a run of float4s of ONE
// row of the torch Linear weight: the 64 lanes of a load instruction touch 64 different cache lines, and 19 such loads per
// wave and layer made the REQUEST of a layer's weights cost 2-4 us of vector-cache line lookups per workgroup (measured:
// the phase scaled with the number of loading waves).  tf_chain_pack_kernel permutes every layer's matrices once per
// forward into units of 64 lanes x float4 = 1 KB in exactly the order the lanes consume them, so a fragment load reads 8
// whole lines.  Unit map of a layer's slot (CH_SLOT units):
constexpr int CH_WO = 0;     // to_out  [32][32]:  tile (0..1) x j (0..1)
constexpr int CH_W2 = 4;     // net.3   [32][64]:  tile (0..1) x j (0..3)
constexpr int CH_W1 = 12;    // net.0   [64][32]:  tile (0..3) x j (0..1)
constexpr int CH_WQ = 20;    // to_qkv  [96][32]:  tile (0..5) x j (0..1)
constexpr int CH_W0 = 32;    // Linear0 [32][Kq]:  (k half x tile) (0..3) x j (0 .. Kq/32 - 1 <= 10)
constexpr int CH_SLOT = 76;
// offsets (floats) of a layer's tensors behind its Linear0 weight [32][Kq] in the flat parameter buffer (state_dict order,
// entries padded to 16 floats; growth 32, mlp 64: HDenseFormer.py:79-89).  The launcher checks the plan's table against them.
constexpr int CO_B0 = 0, CO_LN1G = 32, CO_LN1B = 64, CO_WQKV = 96, CO_WOUT = 3168, CO_BOUT = 4192, CO_LN2G = 4224,
              CO_LN2B = 4256, CO_W1 = 4288, CO_B1 = 6336, CO_W2 = 6400, CO_B2 = 8448;

struct ChainW {              // device-side parameter addressing (a checked digest of TfChainP)
  int32_t l0[4];             // Linear0 weight of layer l, floats from the block's base
  int32_t ooff[4];           // out_layer wa, ba, wb, bb
  int64_t blk0, blk_stride;
};
__device__ __forceinline__ float* chain_w0(const ChainW& cw, float* base, int b, int l) {
  return base + cw.blk0 + (int64_t)b * cw.blk_stride + cw.l0[l];
}
__device__ __forceinline__ TfOutP chain_out(const ChainW& cw, float* base, int b) {
  float* q = base + cw.blk0 + (int64_t)b * cw.blk_stride;
  return TfOutP{q + cw.ooff[0], q + cw.ooff[1], q + cw.ooff[2], q + cw.ooff[3]};
}

// grid (layers, modalities), 256 threads: wave w copies the units u = w (mod 4) of its layer's slot
__global__ __launch_bounds__(256) void tf_chain_pack_kernel(ChainW cw, const float* __restrict__ params, int64_t mstride,
                                                            int DM, int nl, float4* __restrict__ wpack) {
  const int L = blockIdx.x, m = blockIdx.y, b = L >> 2, l = L & 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  const int Kq = DM + 32 * l;
  const float* w0 = chain_w0(cw, const_cast<float*>(params), b, l) + (int64_t)m * mstride;
  const float* rest = w0 + 32 * Kq;
  float4* dst = wpack + ((int64_t)m * nl + L) * (CH_SLOT * 64);
  for (int u = wave; u < CH_SLOT; u += 4) {
    const float* W;
    int ldw, n0, KQ, s0, j;
    if (u < CH_W2) {
      W = rest + CO_WOUT, ldw = 32, n0 = 16 * ((u - CH_WO) >> 1), KQ = 8, s0 = 0, j = (u - CH_WO) & 1;
    } else if (u < CH_W1) {
      W = rest + CO_W2, ldw = 64, n0 = 16 * ((u - CH_W2) >> 2), KQ = 16, s0 = 0, j = (u - CH_W2) & 3;
    } else if (u < CH_WQ) {
      W = rest + CO_W1, ldw = 32, n0 = 16 * ((u - CH_W1) >> 1), KQ = 8, s0 = 0, j = (u - CH_W1) & 1;
    } else if (u < CH_W0) {
      W = rest + CO_WQKV, ldw = 32, n0 = 16 * ((u - CH_WQ) >> 1), KQ = 8, s0 = 0, j = (u - CH_WQ) & 1;
    } else {
      const int f = (u - CH_W0) / 11;
      j = (u - CH_W0) - f * 11;
      if (j >= (Kq >> 5)) continue;
      W = w0, ldw = Kq, n0 = 16 * (f & 1), KQ = Kq >> 2, s0 = (f >> 1) * (Kq >> 3);
    }
    dst[u * 64 + lane] = *reinterpret_cast<const float4*>(W + (int64_t)(n0 + i) * ldw + g * KQ + s0 + 4 * j);
  }
}
// this lane's fragment: units [unit0, unit0 + n4) of the layer's slot (clamped like wload: never branch around a load)
template <int NF4>
__device__ __forceinline__ void pload(WFrag<NF4>& f, const float4* __restrict__ slot, int unit0, int n4) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < NF4; j++) f.v[j] = slot[(unit0 + (j < n4 ? j : 0)) * 64 + lane];
}

// ---- hand-off pieces
// 16-byte write-through store / L1-bypassing load (sc1): buffer instructions with a raw resource over the whole tensor
__device__ __forceinline__ __amdgpu_buffer_rsrc_t chain_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, const float4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ float4 ld16_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
// arrival: called by the whole workgroup after its payload stores
__device__ __forceinline__ void chain_arrive(unsigned* cnt) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until the counter reaches `target`; one lane polls, the workgroup's barrier releases the others.  `dead` is the
// workgroup's sticky give-up flag (lane 0's copy decides; after a timeout no further wait is attempted)
__device__ __forceinline__ void chain_wait(unsigned* cnt, unsigned target, unsigned* tmo, bool& dead) {
  if (threadIdx.x == 0 && !dead) {
    const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (__builtin_amdgcn_s_memrealtime() - t_start > 150000000ull) {   // 1.5 s at 100 MHz
        __hip_atomic_store(tmo, 1u + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dead = true;
        break;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: keeps the loads below the poll)
  __syncthreads();
}

// -DCHAIN_DBG_STAMPS: lane 0 of every workgroup leaves the 100 MHz real-time counter at the phase boundaries of every layer
// in the second megabyte of the sync region (tools/chain_stamps.py); measurement builds only
#ifdef CHAIN_DBG_STAMPS
#define CHAIN_STAMP(k)                                                                                   \
  do {                                                                                                   \
    if (threadIdx.x == 0)                                                                                \
      reinterpret_cast<uint64_t*>(a.sync + (1 << 18))[((size_t)blockIdx.x * 32 + L) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define CHAIN_STAMP(k) ((void)0)
#endif

// ================================================================================================ forward
struct ChainFwd {
  TfDims d;
  ChainW cw;
  const float* params;
  const float4* wpack;   // [M][nl][CH_SLOT][64] fragment-major weights (tf_chain_pack_kernel)
  int dtype;             // storage type of attnall
  float* F0;        // [nb][rows][DMF]
  float* save;      // [nb*4][rows][232] (tf_save layout: h0 | qkv | ob | lse | h1 | h2, segment-major)
  void* attnall;    // channels-last [B][N][M*DM], storage dtype
  unsigned* sync;   // [nseq + 1][SYNC_LINE]: arrival counters, then the timeout word
  int nb, ntile, nseq;
  int64_t rows;
};

template <bool TRAIN>
__global__ __launch_bounds__(CT) void tf_chain_fwd_kernel(ChainFwd a) {
  HDF_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const TfDims& d = a.d;
  const int DM = d.DM, DMF = d.DMF, ldF = DMF + 4, N = d.N, BN = d.B * N;
  float* s_F = sm;                        // [16][ldF] feature rows of the current block
  float* s_x = s_F + TT * ldF;            // [16][36]
  float* s_h = s_x + TT * LD32;           // [16][36]
  float* s_z = s_h + TT * LD32;           // [16][68]
  float* s_red = s_z + TT * LD64;         // [2][16][16]
  float* s_ob = s_red + 2 * 16 * 16;      // [16][36] attention output of the tile
  float* s_q = s_ob + TT * LD32;          // [16][100] q|k|v of the tile
  float* s_kv = s_q + TT * LDQ;           // [2][KVC][68] key / value chunks
  const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6, wave = wave8 & 3, col = lane & 15, g = lane >> 4;
  const bool cw = tid < 256;              // the token stages run on the first four waves (as tok_fwd_kernel)
  // blocks b and b + 8 share an XCD (speed only): the workgroups of a sequence meet in as few L2s as possible
  const int seq = blockIdx.x % a.nseq, tile = blockIdx.x / a.nseq;
  const int m = seq / d.B, bsm = seq - m * d.B;
  const int n0 = tile * TT, nvalid = min(TT, N - n0);
  const int t0k = bsm * N + n0;           // modality-local token index of tile row 0 (dropout index, row addresses)
  const int64_t mo = (int64_t)m * d.mstride, rb = (int64_t)m * BN;
  const DropF dr{TRAIN ? 1 : 0, d.seed, d.thresh24, d.keep_scale};   // (compile-time: no branch per element)
  auto tok = [&](int row) { return t0k + min(row, nvalid - 1); };
  float* pm = const_cast<float*>(a.params);
  unsigned* cnt = a.sync + seq * SYNC_LINE;
  unsigned* tmo = a.sync + a.nseq * SYNC_LINE;
  bool dead = false;
  const int nl = a.nb * 4;
  const int cw_col = 16 * (wave & 1) + col, c4w = 16 * wave + col;

  WFrag<2> f_wo{}, f_w1{};
  WFrag<4> f_w2{};
  WFrag<11> f_w0{};
  WFrag<2> f_q0{}, f_q1{};
  float p_bout = 0.f, p_b1 = 0.f, p_b2 = 0.f, p_b0 = 0.f;
  LnP ln2{}, ln1{};
  float h0r[4] = {0.f, 0.f, 0.f, 0.f};

  // Weights of the token phase that finishes layer L - 1 (POST): requested one attention phase ahead; those of the stages
  // that start layer L (PRE) at the top of the token phase, so that they arrive under the POST stages.  Only the four
  // waves that run the token stages load; the others ZERO their fragment registers (an `if` without that else keeps
  // the previous contents of all 92 registers live across the whole layer loop on the path that skips the request, and
  // the kernel spilled).  The last iteration re-requests the last layer for the same reason.
  auto request_post = [&](int L) __attribute__((always_inline)) {   // L >= 1
    const int Lp = L - 1;
    if (cw) {
      const float4* slot = a.wpack + ((int64_t)m * nl + Lp) * (CH_SLOT * 64);
      const float* rest = chain_w0(a.cw, pm, Lp >> 2, Lp & 3) + mo + 32 * (DM + 32 * (Lp & 3));
      pload(f_wo, slot, CH_WO + 2 * (wave & 1), 2);
      pload(f_w2, slot, CH_W2 + 4 * (wave & 1), 4);
      pload(f_w1, slot, CH_W1 + 2 * wave, 2);
      p_bout = rest[CO_BOUT + cw_col], p_b2 = rest[CO_B2 + cw_col], p_b1 = rest[CO_B1 + c4w];
      ln2 = ln_load(rest + CO_LN2G, rest + CO_LN2B);
    } else {
      f_wo = WFrag<2>{}, f_w2 = WFrag<4>{}, f_w1 = WFrag<2>{};
      p_bout = p_b2 = p_b1 = 0.f, ln2 = LnP{};
    }
  };
  auto request_pre = [&](int L) __attribute__((always_inline)) {
    const int Lc = min(L, nl - 1);
    if (cw) {
      const float4* slot = a.wpack + ((int64_t)m * nl + Lc) * (CH_SLOT * 64);
      const float* rest = chain_w0(a.cw, pm, Lc >> 2, Lc & 3) + mo + 32 * (DM + 32 * (Lc & 3));
      pload(f_w0, slot, CH_W0 + 11 * wave, (DM + 32 * (Lc & 3)) >> 5);   // fragment (k half = wave >> 1, tile = wave & 1)
      pload(f_q0, slot, CH_WQ + 2 * wave, 2);
      pload(f_q1, slot, CH_WQ + 2 * ((wave & 1) + 4), 2);
      p_b0 = rest[CO_B0 + cw_col];
      ln1 = ln_load(rest + CO_LN1G, rest + CO_LN1B);
    } else {
      f_w0 = WFrag<11>{}, f_q0 = WFrag<2>{}, f_q1 = WFrag<2>{};
      p_b0 = 0.f, ln1 = LnP{};
    }
  };

  {  // block 0's input rows (the patch embedding's output)
    const int c4n = DM >> 2;
    for (int i = tid; i < TT * c4n; i += CT) {
      const int row = i / c4n, c4 = (i - row * c4n) * 4;
      *reinterpret_cast<float4*>(s_F + row * ldF + c4) =
          *reinterpret_cast<const float4*>(a.F0 + (rb + tok(row)) * DMF + c4);
    }
  }

  for (int L = 0; L <= nl; L++) {
    const bool POST = L > 0, PRE = L < nl, OUT = POST && (L & 3) == 0;
    const int bp = (L - 1) >> 2, lp = (L - 1) & 3;   // layer finished here
    const int bq = L >> 2, lq = L & 3;               // layer started here
    // (laundered per iteration: left loop-invariant, hipcc hoists every row address of every stage out of the layer
    // loop as 64-bit values and spills them)
    int t0 = t0k;
    asm volatile("" : "+s"(t0));
    request_pre(L);
    CHAIN_STAMP(8);
    __syncthreads();
    CHAIN_STAMP(0);
    // ------------------------------------------------------------------ POST(bp, lp)
    if (POST) {
      float* sv = a.save + (int64_t)(L - 1) * a.rows * 232;
      float* h1s = sv + a.rows * 168;
      float* h2s = sv + a.rows * 200;
      float* Fp = a.F0 + (int64_t)bp * a.rows * DMF;
      const uint32_t site0 = hdf_site_id(m, bp, lp, 0);
      float h1r[4] = {0.f, 0.f, 0.f, 0.f};
      if (cw && wave < 2) {  // to_out + dropout + residual
        f32x4 acc = zero4();
        wmma(acc, f_wo, s_ob, LD32, 8, 0, 2);
        const int c = 16 * wave + col;
        const float bo = p_bout;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 4 * g + r, t = t0 + row;
          const float v = (acc[r] + bo) * dr.mask(site0 + 0, (uint32_t)t * 32 + c) + h0r[r];
          h1r[r] = v;
          s_h[row * LD32 + c] = v;
          if (row < nvalid) h1s[(rb + t) * 32 + c] = v;
        }
      }
      __syncthreads();
      CHAIN_STAMP(9);
#pragma unroll
      for (int pass = 0; pass < 2; pass++) {  // pass 0: h2 = ff(LN2(h1)) + h1 ; pass 1: feature = ff(LN2(h2))
        if (cw) ln32(s_h, s_x, ln2);
        __syncthreads();
        if (cw) {
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
        if (cw && wave < 2) {
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
              if (row < nvalid) h2s[(rb + t) * 32 + c] = h2;
            } else {
              const int fc = DM + 32 * lp + c;
              s_F[row * ldF + fc] = gv;
              if (row < nvalid) Fp[(rb + t) * DMF + fc] = gv;
            }
          }
        }
        __syncthreads();
        if (pass == 0) CHAIN_STAMP(10);
      }
    }
    CHAIN_STAMP(13);
    // ------------------------------------------------------------------ OUT(bp): DenseForward(DM+128 -> 64 -> DM)
    if (OUT) {
      const TfOutP po = chain_out(a.cw, pm, bp);
      float* next_F = PRE ? a.F0 + (int64_t)bq * a.rows * DMF : nullptr;
      const uint32_t siteo = hdf_site_id(m, bp, 4, 0);
      if (cw) {
        f32x4 acc = zero4();
        wmma_stream(acc, po.wa + mo, DMF, 16 * wave, s_F, ldF, DMF >> 2);
        const int c = 16 * wave + col;
        const float ba = po.ba[mo + c];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 4 * g + r, t = t0 + row;
          s_z[row * LD64 + c] = gelu_f(acc[r] + ba) * dr.mask(siteo + 0, (uint32_t)t * 64 + c);
        }
      }
      __syncthreads();  // every wave is done reading s_F: its first DM columns become the next block's input
      if (cw) {
        for (int nn = 16 * wave; nn < DM; nn += 64) {
          f32x4 acc = zero4();
          wmma_stream(acc, po.wb + mo, 64, nn, s_z, LD64, 16);
          const int c = nn + col;
          const float bb = po.bb[mo + c];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = 4 * g + r, t = t0 + row;
            const float v = (acc[r] + bb) * dr.mask(siteo + 1, (uint32_t)t * DM + c);
            if (next_F) {
              s_F[row * ldF + c] = v;
              if (row < nvalid) next_F[(rb + t) * DMF + c] = v;
            } else if (row < nvalid) {
              const int64_t ai = ((int64_t)bsm * N + n0 + row) * ((int64_t)d.M * DM) + (int64_t)m * DM + c;
              if (a.dtype == HDF_BF16)
                ST<bf16_t>::st(reinterpret_cast<bf16_t*>(a.attnall) + ai, v);
              else if (a.dtype == HDF_F16)
                ST<f16_t>::st(reinterpret_cast<f16_t*>(a.attnall) + ai, v);
              else
                reinterpret_cast<float*>(a.attnall)[ai] = v;
            }
          }
        }
      }
      __syncthreads();
    }
    CHAIN_STAMP(1);
    if (!PRE) break;
    // ------------------------------------------------------------------ PRE(bq, lq): Linear0 + LN1 + to_qkv
    float* sv = a.save + (int64_t)L * a.rows * 232;
    {
      float* h0_out = sv;
      const int Kq = DM + 32 * lq;
      f32x4 acc = zero4();
      if (cw) {
        wmma(acc, f_w0, s_F, ldF, Kq >> 2, (wave >> 1) * (Kq >> 3), Kq >> 5);
        if (wave >= 2) {
#pragma unroll
          for (int r = 0; r < 4; r++) s_red[((wave & 1) * 16 + 4 * g + r) * 16 + col] = acc[r];
        }
      }
      __syncthreads();
      if (cw && wave < 2) {
        const int c = 16 * wave + col;
        const float b0 = p_b0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 4 * g + r, t = t0 + row;
          const float h = acc[r] + s_red[(wave * 16 + row) * 16 + col] + b0;
          h0r[r] = h;
          s_h[row * LD32 + c] = h;
          if (row < nvalid) h0_out[(rb + t) * 32 + c] = h;
        }
      }
      __syncthreads();
      CHAIN_STAMP(11);
      if (cw) ln32(s_h, s_x, ln1);
      __syncthreads();
      CHAIN_STAMP(12);
      if (cw) {
        f32x4 acc1 = zero4();
        wmma(acc1, f_q0, s_x, LD32, 8, 0, 2);
#pragma unroll
        for (int r = 0; r < 4; r++) s_q[(4 * g + r) * LDQ + 16 * wave + col] = acc1[r];
        if (wave < 2) {
          f32x4 acc2 = zero4();
          wmma(acc2, f_q1, s_x, LD32, 8, 0, 2);
#pragma unroll
          for (int r = 0; r < 4; r++) s_q[(4 * g + r) * LDQ + 16 * (wave + 4) + col] = acc2[r];
        }
      }
      __syncthreads();
    }
    CHAIN_STAMP(2);
    // ------------------------------------------------------------------ publish q|k|v, per-sequence barrier
    float* qkvL = sv + a.rows * 32;                               // [rows][96] of this layer
    const __amdgpu_buffer_rsrc_t rq = chain_rsrc(qkvL + (rb + (int64_t)bsm * N) * 96);   // this sequence's rows
    if (tid < TT * 24) {
      const int row = tid / 24, c4 = (tid - row * 24) * 4;
      if (row < nvalid)
        st16_sc1(rq, (uint32_t)(((n0 + row) * 96 + c4) * 4), *reinterpret_cast<const float4*>(s_q + row * LDQ + c4));
    }
    chain_arrive(cnt);
    CHAIN_STAMP(3);
    request_post(L + 1);   // parameters: never written during the launch, plain loads
    const int head = wave8;
    const float bqv = s_q[col * LDQ + head * 4 + g] * (0.5f * LOG2E);
    chain_wait(cnt, (unsigned)(a.ntile * (L + 1)), tmo, dead);
    CHAIN_STAMP(4);
    // ------------------------------------------------------------------ attention of layer L: head = wave
    {
#pragma clang fp contract(off)   // explicit fmas only: attn_fwd_kernel's arithmetic, bit for bit (see there)
      const int NP = attn_rows(N), nchunk = (NP + KVC - 1) / KVC;
      // staging: thread -> rows (tid >> 3) and (tid >> 3) + 64 of the chunk, 16-byte part (tid & 7) of their K and V halves
      const int srow = tid >> 3, spart = tid & 7;
      float4 pk[2], pv[2];
      auto stage_load = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int j = min(c * KVC + srow + 64 * h, N - 1);
          pk[h] = ld16_sc1(rq, (uint32_t)((j * 96 + 32 + 4 * spart) * 4));
          pv[h] = ld16_sc1(rq, (uint32_t)((j * 96 + 64 + 4 * spart) * 4));
        }
      };
      auto stage_store = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const bool real = c * KVC + srow + 64 * h < N;
          float* dst = s_kv + (c & 1) * (KVC * KVP) + (srow + 64 * h) * KVP + 4 * spart;
          // (component selects: `real ? pk : zero` on the float4s is a select between two ADDRESSES and goes through scratch)
          *reinterpret_cast<float4*>(dst) =
              make_float4(real ? pk[h].x : 0.f, real ? pk[h].y : 0.f, real ? pk[h].z : 0.f, real ? pk[h].w : 0.f);
          *reinterpret_cast<float4*>(dst + 32) =
              make_float4(real ? pv[h].x : 0.f, real ? pv[h].y : 0.f, real ? pv[h].z : 0.f, real ? pv[h].w : 0.f);
        }
      };
      stage_load(0);
      stage_store(0);
      __syncthreads();
      CHAIN_STAMP(5);
      float mx = -INFINITY, l = 0.f;
      f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
      // one block of 16 keys against the wave's 16 queries (attn_fwd_kernel's trip, operation for operation)
      auto trip = [&](const float4 (&v)[4], int j0, const f32x4& sc4, auto masked) __attribute__((always_inline)) {
        float sc[4];
        float mn = mx;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          sc[u] = sc4[u];
          if (decltype(masked)::value) sc[u] = (j0 + 4 * g + u < N) ? sc[u] : -INFINITY;
          mn = fmaxf(mn, sc[u]);
        }
        const float mr = (mn == -INFINITY) ? 0.f : mn;
        const float cfac = __builtin_amdgcn_exp2f(mx - mr);
        float ps = 0.f;
        f2 b01 = {0.f, 0.f}, b23 = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const float pr = __builtin_amdgcn_exp2f(sc[u] - mr);
          const f2 pp = {pr, pr};
          ps += pr;
          b01 = __builtin_elementwise_fma(pp, lo2(v[u]), b01);
          b23 = __builtin_elementwise_fma(pp, hi2(v[u]), b23);
        }
        const f2 cc = {cfac, cfac};
        l = __builtin_fmaf(l, cfac, ps);
        a01 = __builtin_elementwise_fma(a01, cc, b01);
        a23 = __builtin_elementwise_fma(a23, cc, b23);
        mx = mn;
      };
      for (int c = 0; c < nchunk; c++) {
        if (c + 1 < nchunk) stage_load(c + 1);
        const float* sK = s_kv + (c & 1) * (KVC * KVP);
        const float* sKq = sK + col * KVP + head * 4 + g;         // + 16 k rows: the score MFMA's A operand of block k
        const float* sVq = sK + 4 * g * KVP + 32 + head * 4;      // + (16 k + u) rows: this lane's value rows of block k
        const int jbase = c * KVC;
        auto scores = [&](float ka) __attribute__((always_inline)) {
          f32x4 z = {0.f, 0.f, 0.f, 0.f};
          return __builtin_amdgcn_mfma_f32_16x16x4f32(ka, bqv, z, 0, 0, 0);
        };
        if (jbase + KVC <= N) {
          // whole chunk, no mask: all eight K operands requested at once, the value rows one block ahead of their use (with
          // one LDS round trip in front of the MFMA and one in front of the first P.V of every block the loop took 600
          // cycles per block at two waves per SIMD)
          constexpr int NT = KVC / ATRIP;
          float ka[NT];
#pragma unroll
          for (int k = 0; k < NT; k++) ka[k] = sKq[k * ATRIP * KVP];
          float4 v[2][4];
#pragma unroll
          for (int u = 0; u < 4; u++) v[0][u] = *reinterpret_cast<const float4*>(sVq + u * KVP);
          f32x4 cur = scores(ka[0]);
#pragma unroll
          for (int k = 0; k < NT; k++) {
            f32x4 nxt = cur;
            if (k + 1 < NT) {
              nxt = scores(ka[k + 1]);
#pragma unroll
              for (int u = 0; u < 4; u++)
                v[(k + 1) & 1][u] = *reinterpret_cast<const float4*>(sVq + ((k + 1) * ATRIP + u) * KVP);
            }
            __builtin_amdgcn_sched_barrier(0);
            trip(v[k & 1], jbase + k * ATRIP, cur, std::false_type{});
            cur = nxt;
          }
        } else {
          const int ntrip = (NP - jbase) / ATRIP;
          f32x4 cur = scores(sKq[0]);
          for (int k = 0; k < ntrip; k++) {
            const int j0 = jbase + k * ATRIP;
            const f32x4 nxt = scores(sKq[min(k + 1, ntrip - 1) * ATRIP * KVP]);
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const float4*>(sVq + (k * ATRIP + u) * KVP);
            __builtin_amdgcn_sched_barrier(0);
            if (j0 + ATRIP <= N)
              trip(v, j0, cur, std::false_type{});
            else
              trip(v, j0, cur, std::true_type{});
            cur = nxt;
          }
        }
        if (c + 1 < nchunk) stage_store(c + 1);
        __syncthreads();
      }
      CHAIN_STAMP(6);
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
      if (g == 0) {
        const float inv = 1.f / l;
        const float4 o = make_float4(a01.x * inv, a01.y * inv, a23.x * inv, a23.y * inv);
        *reinterpret_cast<float4*>(s_ob + col * LD32 + head * 4) = o;
        if (col < nvalid) {
          float* ob = sv + a.rows * 128;
          float* lse = sv + a.rows * 160;
          const int64_t R = rb + t0 + col;
          *reinterpret_cast<float4*>(ob + R * 32 + head * 4) = o;
          lse[R * 8 + head] = __builtin_fmaf(mx, LN2, __logf(l));
        }
      }
      CHAIN_STAMP(7);
    }
  }
}

size_t chain_fwd_lds(const TfDims& d) {
  return (size_t)(TT * (d.DMF + 4) + 2 * TT * LD32 + TT * LD64 + 2 * 16 * 16 + TT * LD32 + TT * LDQ + 2 * KVC * KVP) *
         sizeof(float);
}

template <typename Kern>
int chain_allow_lds(Kern kern, size_t bytes) {
  if (bytes <= 64 * 1024) return HDF_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)LDS_LIMIT_F);
  if (e != hipSuccess) {
    hdf_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(e));
    return HDF_ERR_HIP;
  }
  return HDF_OK;
}

}  // namespace

// the plan's parameter table in the form the kernels address it; false when it is not the regular layout they assume
static bool chain_digest(const TfChainP& cp, int DM, ChainW& w) {
  static const int rel[13] = {0, CO_B0, CO_LN1G, CO_LN1B, CO_WQKV, CO_WOUT, CO_BOUT, CO_LN2G, CO_LN2B, CO_W1, CO_B1, CO_W2, CO_B2};
  for (int l = 0; l < 4; l++) {
    w.l0[l] = cp.loff[l][0];
    for (int k = 1; k < 13; k++)
      if (cp.loff[l][k] != cp.loff[l][0] + 32 * (DM + 32 * l) + rel[k]) return false;
  }
  for (int k = 0; k < 4; k++) w.ooff[k] = cp.ooff[k];
  w.blk0 = cp.blk0, w.blk_stride = cp.blk_stride;
  return true;
}

size_t tf_chain_sync_bytes(const TfDims& d) { return (size_t)(d.M * d.B + 1) * SYNC_LINE * sizeof(unsigned); }
size_t tf_chain_wpack_bytes(const TfDims& d, int nb) { return (size_t)d.M * nb * 4 * CH_SLOT * 64 * sizeof(float4); }

bool tf_chain_supported(const TfDims& d) {
  const int ntile = ceil_div(d.N, TT);
  return d.DM % 32 == 0 && d.DM >= 32 && d.DM <= 256 && d.N >= 1 && d.M * d.B * ntile <= hdf_cu_budget() &&
         (int64_t)d.N * 96 * 4 < ((int64_t)1 << 31);
}

int tf_chain_forward(const TfDims& d, const TfChainP& cp, int nb, const float* params, float* F0, float* save,
                     void* attnall, unsigned* sync, void* wpack, int dtype, hipStream_t st) {
  HDF_CHECK_ARG(tf_chain_supported(d), "transformer chain: shape not supported (M %d B %d N %d DM %d)", d.M, d.B, d.N, d.DM);
  ChainFwd a{};
  HDF_CHECK_ARG(chain_digest(cp, d.DM, a.cw), "transformer chain: irregular parameter layout");
  a.d = d, a.params = params, a.F0 = F0, a.save = save, a.attnall = attnall, a.sync = sync;
  a.wpack = reinterpret_cast<const float4*>(wpack), a.dtype = dtype;
  a.nb = nb, a.ntile = ceil_div(d.N, TT), a.nseq = d.M * d.B, a.rows = (int64_t)d.M * d.B * d.N;
  const size_t shm = chain_fwd_lds(d);
  HDF_CHECK_ARG(shm <= LDS_LIMIT_F, "transformer chain: %zu B of LDS", shm);
  HDF_CHECK_ARG(dtype == HDF_F32 || dtype == HDF_BF16 || dtype == HDF_F16, "unsupported dtype %d", dtype);
  hipError_t e = hipMemsetAsync(sync, 0, tf_chain_sync_bytes(d), st);
  if (e != hipSuccess) {
    hdf_set_error("transformer chain: hipMemsetAsync failed: %s", hipGetErrorString(e));
    return HDF_ERR_HIP;
  }
  hipLaunchKernelGGL(tf_chain_pack_kernel, dim3(nb * 4, d.M), dim3(256), 0, st, a.cw, params, d.mstride, d.DM, nb * 4,
                     reinterpret_cast<float4*>(wpack));
  HDF_LAUNCH_CHECK();
  const dim3 grid(a.nseq * a.ntile);
  if (d.training) {
    HDF_TRY(chain_allow_lds(tf_chain_fwd_kernel<true>, shm));
    hipLaunchKernelGGL(tf_chain_fwd_kernel<true>, grid, dim3(CT), shm, st, a);
  } else {
    HDF_TRY(chain_allow_lds(tf_chain_fwd_kernel<false>, shm));
    hipLaunchKernelGGL(tf_chain_fwd_kernel<false>, grid, dim3(CT), shm, st, a);
  }
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}
