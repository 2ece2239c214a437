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
#include <mutex>

#include "tf_tok.h"

namespace {

using namespace tftok;

constexpr int CT = 512;     // threads per workgroup: 8 waves = the 8 heads of Dense_Attention
constexpr int KVC = 128;    // keys per staged chunk
constexpr int KVP = 68;     // floats per row of a chunk image: K (8 heads x 4) | V (8 heads x 4) | pad
constexpr int LDQ = 100;    // row pitch of the q|k|v tile
constexpr int SYNC_LINE = 32;  // unsigned words per counter (128-byte lines)
// Operand records of the attention BACKWARD in the 16-bit storage modes.  A workgroup's tile is exactly one block of 16
// tokens of the score / accumulation MFMAs, so the token's owner can leave every operand any wave of the sequence will
// ever want from that block in the order the 64 lanes consume it: the backward's key / query loops then run on coalesced
// loads straight into registers -- no LDS image, no barrier, a prefetch ring -- where the first version staged every
// chunk of the sequence through LDS in each of the sequence's 32 workgroups (30 us per layer; this form: see DESIGN).
// Forward record per (layer, sequence, block, head), FR_W floats (plain stores: read by the backward LAUNCH):
constexpr int FR_KA = 0;      // [64] k of key (lane & 15), component (lane >> 4), zero for padded rows: score MFMA A operand
constexpr int FR_VA = 64;     // [64] v likewise                                                   (T = dO . v)
constexpr int FR_QA = 128;    // [64] q x 0.5 log2(e)                                              (dK/dV half)
constexpr int FR_KB = 192;    // [5][4] x 8 B: 16-bit k of keys 4g .. 4g+3, component c (row 4: zeros): B operand of dS . K
constexpr int FR_QB = 232;    // [5][4] x 8 B: 16-bit q likewise                                    (dS^T . Q)
constexpr int FR_W = 288;      // (16 spare floats: whole 128-byte lines)
// -lse log2(e) of the queries (-inf for padded rows) follows the records as [layer][sequence][head][ntile * 16]: a wave
// stages its head's row with two 16-byte loads per lane
// hand-off record of the backward per (layer parity, sequence, block, head), XG_W floats (sc1 stores and loads):
constexpr int XG_GA = 0;      // [64] dO of query (lane & 15), component (lane >> 4)
constexpr int XG_GB = 64;     // [5][4] x 8 B: 16-bit dO
constexpr int XG_W = 128;
// -delta of the queries follows the hand-off records as [parity][sequence][head][ntile * 16]
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
// backward (tok_bwd_kernel's column fragments: cload), same slot:
constexpr int CB_Q = 76;     // dt = dqkv Wqkv:   (o half x tile) (0..3) x j (0..2)      (12 scalars per lane)
constexpr int CB_W0 = 88;    // dF += dh0 W0:     tile (0 .. Kq/16 - 1 <= 23) x j (0..1)
constexpr int CB_W2 = 136;   // df = dg W2:       tile (0..3) x j (0..1)
constexpr int CB_W1 = 144;   // du = dz W1:       (o half x tile) (0..3) x j (0..1)
constexpr int CB_WO = 152;   // dO = dgo Wout:    (o half x tile) (0..3)
constexpr int CH_SLOT = 156;
// a block's out_layer, behind all layer slots: [M][nb][CO_SLOT] units
constexpr int OA_F = 0;      // wa rows   (wmma_stream: z = Wa F):        tile (0..3) x j (0 .. DMF/16 - 1 <= 23)
constexpr int OB_F = 96;     // wb rows   (wmma_stream: out = Wb f):      tile (0 .. DM/16 - 1 <= 15) x j (0..3)
constexpr int OB_B = 160;    // wb columns (cmma_stream: df = do Wb):     tile (0..3) x j (0 .. DM/16 - 1 <= 15)
constexpr int OA_B = 224;    // wa columns (cmma_stream: dF = dz Wa):     tile (0 .. DMF/16 - 1 <= 23) x j (0..3)
constexpr int CO_SLOT = 320;
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

// grid (layers + blocks, modalities, z), 256 threads: wave w of slice z copies the units u = 4 z + w (mod 4 gridDim.z) of its slot.
// Row fragment (wload): lane (i, g) <- W[n0 + i][g KQ + s0 + 4 j .. + 3]; column fragment (cload): element e of the lane's
// float4 <- W[g OQ + s0 + 4 j + e][n0 + i].
__global__ __launch_bounds__(256) void tf_chain_pack_kernel(ChainW cw, const float* __restrict__ params, int64_t mstride,
                                                            int DM, int nl, float4* __restrict__ wpack) {
  HDF_LIGHT_PRIO();   // (runs beside the first level-0 conv since round 5: plan.hip forward3d)
  const int m = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  const int DMF = DM + 128, M = gridDim.y, nb = nl >> 2;
  auto rowf = [&](const float* W, int ldw, int n0, int KQ, int s0, int j) {
    return *reinterpret_cast<const float4*>(W + (int64_t)(n0 + i) * ldw + g * KQ + s0 + 4 * j);
  };
  auto colf = [&](const float* W, int ldw, int n0, int OQ, int s0, int j) {
    const float* q = W + (int64_t)(g * OQ + s0 + 4 * j) * ldw + n0 + i;
    return make_float4(q[0], q[ldw], q[2 * ldw], q[3 * ldw]);
  };
  if ((int)blockIdx.x < nl) {
    const int L = blockIdx.x, b = L >> 2, l = L & 3, Kq = DM + 32 * l;
    const float* w0 = chain_w0(cw, const_cast<float*>(params), b, l) + (int64_t)m * mstride;
    const float* rest = w0 + 32 * Kq;
    float4* dst = wpack + ((int64_t)m * nl + L) * (CH_SLOT * 64);
    for (int u = blockIdx.z * 4 + wave; u < CH_SLOT; u += 4 * gridDim.z) {
      float4 v;
      if (u < CH_W2) {
        v = rowf(rest + CO_WOUT, 32, 16 * ((u - CH_WO) >> 1), 8, 0, (u - CH_WO) & 1);
      } else if (u < CH_W1) {
        v = rowf(rest + CO_W2, 64, 16 * ((u - CH_W2) >> 2), 16, 0, (u - CH_W2) & 3);
      } else if (u < CH_WQ) {
        v = rowf(rest + CO_W1, 32, 16 * ((u - CH_W1) >> 1), 8, 0, (u - CH_W1) & 1);
      } else if (u < CH_W0) {
        v = rowf(rest + CO_WQKV, 32, 16 * ((u - CH_WQ) >> 1), 8, 0, (u - CH_WQ) & 1);
      } else if (u < CB_Q) {
        const int f = (u - CH_W0) / 11, j = (u - CH_W0) - f * 11;
        if (j >= (Kq >> 5)) continue;
        v = rowf(w0, Kq, 16 * (f & 1), Kq >> 2, (f >> 1) * (Kq >> 3), j);
      } else if (u < CB_W0) {
        const int f = (u - CB_Q) / 3, j = (u - CB_Q) - f * 3;
        v = colf(rest + CO_WQKV, 32, 16 * (f & 1), 24, 12 * (f >> 1), j);
      } else if (u < CB_W2) {
        const int t = (u - CB_W0) >> 1, j = (u - CB_W0) & 1;
        if (16 * t >= Kq) continue;
        v = colf(w0, Kq, 16 * t, 8, 0, j);
      } else if (u < CB_W1) {
        v = colf(rest + CO_W2, 64, 16 * ((u - CB_W2) >> 1), 8, 0, (u - CB_W2) & 1);
      } else if (u < CB_WO) {
        const int f = (u - CB_W1) >> 1;
        v = colf(rest + CO_W1, 32, 16 * (f & 1), 16, 8 * (f >> 1), (u - CB_W1) & 1);
      } else {
        const int f = u - CB_WO;
        v = colf(rest + CO_WOUT, 32, 16 * (f & 1), 8, 4 * (f >> 1), 0);
      }
      dst[u * 64 + lane] = v;
    }
  } else {
    const int b = blockIdx.x - nl;
    const TfOutP po = chain_out(cw, const_cast<float*>(params), b);
    const float* wa = po.wa + (int64_t)m * mstride;
    const float* wb = po.wb + (int64_t)m * mstride;
    float4* dst = wpack + (int64_t)M * nl * (CH_SLOT * 64) + ((int64_t)m * nb + b) * (CO_SLOT * 64);
    const int nja = DMF >> 4, ntb = DM >> 4;
    for (int u = blockIdx.z * 4 + wave; u < CO_SLOT; u += 4 * gridDim.z) {
      float4 v;
      if (u < OB_F) {
        const int t = u / 24, j = u - t * 24;
        if (j >= nja) continue;
        v = rowf(wa, DMF, 16 * t, DMF >> 2, 0, j);
      } else if (u < OB_B) {
        const int t = (u - OB_F) >> 2;
        if (t >= ntb) continue;
        v = rowf(wb, 64, 16 * t, 16, 0, (u - OB_F) & 3);
      } else if (u < OA_B) {
        const int t = (u - OB_B) >> 4, j = (u - OB_B) & 15;
        if (j >= ntb) continue;
        v = colf(wb, 64, 16 * t, DM >> 2, 0, j);
      } else {
        const int t = (u - OA_B) >> 2;
        if (t >= nja) continue;
        v = colf(wa, DMF, 16 * t, 16, 0, (u - OA_B) & 3);
      }
      dst[u * 64 + lane] = v;
    }
  }
}
// this lane's fragment: units [unit0, unit0 + n4) of the slot (clamped like wload: never branch around a load)
template <int NF4>
__device__ __forceinline__ void pload(WFrag<NF4>& f, const float4* __restrict__ slot, int unit0, int n4) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < NF4; j++) f.v[j] = slot[(unit0 + (j < n4 ? j : 0)) * 64 + lane];
}
// column fragment of NS scalars = NS / 4 units
template <int NS>
__device__ __forceinline__ void cpload(CFrag<NS>& f, const float4* __restrict__ slot, int unit0) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < NS / 4; j++) {
    const float4 v = slot[(unit0 + j) * 64 + lane];
    f.v[4 * j] = v.x, f.v[4 * j + 1] = v.y, f.v[4 * j + 2] = v.z, f.v[4 * j + 3] = v.w;
  }
}
// wmma_stream / cmma_stream (tf_tok.h) on packed units: the same MFMA order, the weights of the next chunk in flight
__device__ __forceinline__ void wmma_stream_p(f32x4& acc, const float4* __restrict__ slot, int unit0, int n4, const float* sA,
                                              int lda, int KQ) {
  const int lane = threadIdx.x & 63;
  const float4* pw = slot + unit0 * 64 + lane;
  const float4* pa = reinterpret_cast<const float4*>(sA + (lane & 15) * lda + (lane >> 4) * KQ);
  float4 w[4], wn[4];
#pragma unroll
  for (int j = 0; j < 4; j++) w[j] = pw[min(j, n4 - 1) * 64];
  for (int c = 0; c < n4; c += 4) {
#pragma unroll
    for (int j = 0; j < 4; j++) wn[j] = pw[min(c + 4 + j, n4 - 1) * 64];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (c + j < n4) {
        const float4 av = pa[c + j];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, w[j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, w[j].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, w[j].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, w[j].w, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = wn[j];
  }
}
// OQ scalars per lane = OQ / 4 units
__device__ __forceinline__ void cmma_stream_p(f32x4& acc, const float4* __restrict__ slot, int unit0, const float* sA, int lda,
                                              int OQ) {
  const int lane = threadIdx.x & 63;
  const float4* pw = slot + unit0 * 64 + lane;
  const float4* pa = reinterpret_cast<const float4*>(sA + (lane & 15) * lda + (lane >> 4) * OQ);
  const int n4 = OQ >> 2;
  for (int c = 0; c < n4; c += 2) {
    const float4 w0 = pw[c * 64], w1 = pw[min(c + 1, n4 - 1) * 64];
    {
      const float4 av = pa[c];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, w0.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, w0.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, w0.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, w0.w, acc, 0, 0, 0);
    }
    if (c + 1 < n4) {
      const float4 av = pa[c + 1];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, w1.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, w1.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, w1.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, w1.w, acc, 0, 0, 0);
    }
  }
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
__device__ __forceinline__ void chain_arrive(unsigned* cnt, bool storing = true) {
  if (storing) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until the counter reaches `target`; one lane polls, the workgroup's barrier releases the others.  `dead` is the
// workgroup's sticky give-up flag (lane 0's copy decides; after a timeout no further wait is attempted).
// Giving up (round 6; rounds 5 trapped here, which killed the caller's HIP context): a sibling that has not arrived within
// `c.ticks` of the 100 MHz real-time counter means the grid is not resident together -- the device is shared with something
// that holds compute units.  The workgroup then leaves 1 + its id in the sync region's timeout word AND in the plan's
// host-mapped flag word (system scope: the host reads it without synchronising, plan.hip chain_flag_check), stops waiting
// for the rest of the launch and runs to the end on whatever it reads; every other workgroup sees the word in its own poll
// loop and does the same, so the launch ends within about one timeout.  Its results are garbage BY DEFINITION: dead
// workgroups overwrite their rows of the branch output with NaN at the end of the kernel (which rows were lost: NaN does
// not survive relu(InstanceNorm(.)) = fmaxf(.., 0) downstream), the plan's last launch of the call turns the head of every
// output / of the gradient buffer into NaN when the timeout word is set (plan.hip: chain_poison_*_kernel), so the step's
// loss and optimizer step are NaN rather than plausible, and the plan's next call returns HDF_ERR_CHAIN_TIMEOUT once and
// routes this plan to the launch chain from then on.
using ChainCtl = TfChainCtl;   // (transformer.h: host-mapped flag word + give-up deadline)
__device__ __forceinline__ void chain_wait(unsigned* cnt, unsigned target, unsigned* tmo, const ChainCtl& c, bool& dead) {
  if (threadIdx.x == 0 && !dead) {
    const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 31u) == 0u && __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        dead = true;   // a sibling (of any sequence) gave up: the launch is lost, do not sit out a deadline of our own
        break;
      }
      if (__builtin_amdgcn_s_memrealtime() - t_start > (uint64_t)c.ticks) {
        __hip_atomic_store(tmo, 1u + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c.host_flag) __hip_atomic_store(c.host_flag, 1u + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        dead = true;
        break;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: keeps the loads below the poll)
  __syncthreads();
}
// lane 0's give-up flag -> the whole workgroup (uniform result); `word`: any LDS word nobody reads across this call
__device__ __forceinline__ bool chain_any_dead(bool dead, unsigned* word) {
  __syncthreads();
  if (threadIdx.x == 0) *word = dead ? 1u : 0u;
  __syncthreads();
  return *word != 0u;
}

// -DCHAIN_DBG_STAMPS: lane 0 of every workgroup leaves the 100 MHz real-time counter at the phase boundaries of every layer
// in the second megabyte of the sync region (tools/chain_stamps.py); measurement builds only
#ifdef CHAIN_DBG_STAMPS
#define CHAIN_STAMP(k)                                                                                   \
  do {                                                                                                   \
    if (threadIdx.x == 0)                                                                                \
      reinterpret_cast<uint64_t*>(a.sync + (1 << 18))[((size_t)blockIdx.x * 32 + L) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define CHAIN_STAMPB(k)                                                                                  \
  do {                                                                                                   \
    if (threadIdx.x == 0)                                                                                \
      reinterpret_cast<uint64_t*>(a.sync + (1 << 18) + (1 << 17))[((size_t)blockIdx.x * 32 + (L + 1)) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define CHAIN_STAMP(k) ((void)0)
#define CHAIN_STAMPB(k) ((void)0)
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
  TfChainCtl ctl;
  float* frag;      // [nl][nseq][ntile][8][FR_W] operand records for the backward (16-bit storage modes), or null
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
      const float4* oslot = a.wpack + (int64_t)d.M * nl * (CH_SLOT * 64) + ((int64_t)m * a.nb + bp) * (CO_SLOT * 64);
      float* next_F = PRE ? a.F0 + (int64_t)bq * a.rows * DMF : nullptr;
      const uint32_t siteo = hdf_site_id(m, bp, 4, 0);
      if (cw) {
        f32x4 acc = zero4();
        wmma_stream_p(acc, oslot, OA_F + 24 * wave, DMF >> 4, s_F, ldF, DMF >> 2);
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
          wmma_stream_p(acc, oslot, OB_F + 4 * (nn >> 4), 4, s_z, LD64, 16);
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
    if (a.frag) {   // the backward's operand records of this block: every wave its head (see FR_*)
      float* rec = a.frag + ((((int64_t)L * a.nseq + seq) * a.ntile + tile) * 8 + wave8) * FR_W;
      const bool vrow = col < nvalid;
      const float* qr = s_q + col * LDQ + wave8 * 4 + g;
      rec[FR_KA + lane] = vrow ? qr[32] : 0.f;
      rec[FR_VA + lane] = vrow ? qr[64] : 0.f;
      rec[FR_QA + lane] = vrow ? qr[0] * (0.5f * LOG2E) : 0.f;
      if (lane < 20) {
        const int c = lane >> 2, gg = lane & 3;
        float kq[4] = {0.f, 0.f, 0.f, 0.f}, qq[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < 4) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float* r = s_q + (4 * gg + e) * LDQ + wave8 * 4 + c;
            const bool v = 4 * gg + e < nvalid;
            kq[e] = v ? r[32] : 0.f, qq[e] = v ? r[0] : 0.f;
          }
        }
        u32x2* ru = reinterpret_cast<u32x2*>(rec);
        if (a.dtype == HDF_F16) {
          ru[FR_KB / 2 + lane] = u32x2{pack_h2(kq[0], kq[1]), pack_h2(kq[2], kq[3])};
          ru[FR_QB / 2 + lane] = u32x2{pack_h2(qq[0], qq[1]), pack_h2(qq[2], qq[3])};
        } else {
          ru[FR_KB / 2 + lane] = u32x2{pack_bf2(kq[0], kq[1]), pack_bf2(kq[2], kq[3])};
          ru[FR_QB / 2 + lane] = u32x2{pack_bf2(qq[0], qq[1]), pack_bf2(qq[2], qq[3])};
        }
      }
    }
    chain_arrive(cnt);
    CHAIN_STAMP(3);
    request_post(L + 1);   // parameters: never written during the launch, plain loads
    const int head = wave8;
    const float bqv = s_q[col * LDQ + head * 4 + g] * (0.5f * LOG2E);
    chain_wait(cnt, (unsigned)(a.ntile * (L + 1)), tmo, a.ctl, dead);
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
        const float lsv = __builtin_fmaf(mx, LN2, __logf(l));
        if (col < nvalid) {
          float* ob = sv + a.rows * 128;
          float* lse = sv + a.rows * 160;
          const int64_t R = rb + t0 + col;
          *reinterpret_cast<float4*>(ob + R * 32 + head * 4) = o;
          lse[R * 8 + head] = lsv;
        }
        if (a.frag)   // -lse log2(e) of every query, head-major: [layer][sequence][head][ntile * 16] behind the records
          a.frag[(int64_t)nl * a.nseq * a.ntile * 8 * FR_W + (((int64_t)L * a.nseq + seq) * 8 + head) * (a.ntile * TT) + n0 + col] =
              col < nvalid ? -lsv * LOG2E : -INFINITY;
      }
      CHAIN_STAMP(7);
    }
  }
  // a workgroup that gave up at a barrier overwrites its rows of the branch output with NaN (chain_wait; a marker of WHICH
  // rows were lost -- what makes the step's loss NaN is plan.hip's chain_poison_outputs_kernel)
  if (chain_any_dead(dead, reinterpret_cast<unsigned*>(s_red))) {
    const float qnan = __builtin_nanf("");
    for (int i = tid; i < nvalid * DM; i += CT) {
      const int row = i / DM, c = i - row * DM;
      const int64_t ai = ((int64_t)bsm * N + n0 + row) * ((int64_t)d.M * DM) + (int64_t)m * DM + c;
      if (a.dtype == HDF_BF16)
        ST<bf16_t>::st(reinterpret_cast<bf16_t*>(a.attnall) + ai, qnan);
      else if (a.dtype == HDF_F16)
        ST<f16_t>::st(reinterpret_cast<f16_t*>(a.attnall) + ai, qnan);
      else
        reinterpret_cast<float*>(a.attnall)[ai] = qnan;
    }
  }
}

// ================================================================================================ backward
// Mirror image of the forward kernel (HDenseFormer.py:78-145 differentiated): per layer L, from the last one down,
//     token phase:  PREB(L + 1)  Linear0 / LN1 / to_qkv backward of the layer whose attention backward just ran
//                   OUTB(b)      out_layer backward at a block boundary
//                   POSTB(L)     ff / ff / to_out backward of layer L -> dO
//     publish dO | delta (dO . ob per head) of the tile, per-sequence barrier
//     attention backward of layer L: dQ of the tile's 16 queries against all keys, dK / dV of its 16 keys against all
//     queries (q, k, v, lse come from the forward LAUNCH: plain loads; only dO | delta are handed off inside this one)
// and one last PREB(0).  The stages are tok_bwd_kernel's (same operations, same order), the attention halves
// attn_bwd_kernel's (LP = 0: exact fp32) or attn_bwd_lp_kernel's (LP = 1 / 2: the three accumulations on
// v_mfma_f32_16x16x16 with bf16 / f16 operands), a wave per head.  What no other workgroup needs stays on the chip: the
// gradient of the feature buffer (s_dF: only block 0's input gradient is written back, for the patch embedding),
// d(qkv), the residual-path gradient.  Weight-matrix gradients leave through the tapes (tf_wgrad after this launch).
struct ChainBwd {
  TfDims d;
  ChainW cw;
  const float* params;
  float* grads;
  const float* F0;        // [nb][rows][DMF]
  const float* save;      // tf_save layout
  float* dF;              // [rows][DMF]: columns [0, DM) = gradient of block 0's input on return
  const void* d_attnall;  // storage dtype
  const float4* wpack;    // fragment-major weights (written by the forward's pack launch: same parameters)
  float* tape;            // [nb*4][TF_TAPE_W segments][rows]
  float* otape;           // [nb][DMF segments][rows]
  float* xchg;            // hand-off scratch, by layer parity: [2][rows][40] dO | delta rows (exact mode) or
                          // [2][nseq][ntile][8][XG_W] operand records (16-bit modes)
  const float* frag;      // the forward's operand records (16-bit modes)
  unsigned* sync;         // [nseq + 1][SYNC_LINE]
  TfChainCtl ctl;
  int nb, ntile, nseq, dtype;
  int64_t rows;
};

constexpr int QC = 64;      // rows per staged chunk of the attention backward
constexpr int QP = 84;      // floats per row of the dK/dV chunk image: q (32) | dO (32) | pad; K|V image uses KVP
constexpr int XW = 40;      // hand-off row: dO (32) | delta (8)

template <int LP>
struct ChainLp;
template <>
struct ChainLp<1> {
  static __device__ __forceinline__ uint16_t one(float v) { return f2bf(v); }
  static __device__ __forceinline__ u32x2 four(float a, float b, float c, float d) { return u32x2{pack_bf2(a, b), pack_bf2(c, d)}; }
  static __device__ __forceinline__ f32x4 mma(const u32x2& a, const u32x2& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
  }
};
typedef __attribute__((ext_vector_type(4))) _Float16 chain_f16x4;
template <>
struct ChainLp<2> {
  static __device__ __forceinline__ uint16_t one(float v) { return f2h(v); }
  static __device__ __forceinline__ u32x2 four(float a, float b, float c, float d) { return u32x2{pack_h2(a, b), pack_h2(c, d)}; }
  static __device__ __forceinline__ f32x4 mma(const u32x2& a, const u32x2& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(chain_f16x4, a), __builtin_bit_cast(chain_f16x4, b), c, 0, 0, 0);
  }
};

__device__ __forceinline__ f32x4 chain_mfma4(float a, float b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <bool TRAIN, int LP>
__global__ __launch_bounds__(CT) void tf_chain_bwd_kernel(ChainBwd a) {
  HDF_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const TfDims& d = a.d;
  const int DM = d.DM, DMF = d.DMF, ldF = DMF + 4, ldD = DM + 4, N = d.N, BN = d.B * N;
  float* s_F = sm;                       // [16][ldF]  feature rows of block bo (OUTB)
  float* s_dF = s_F + TT * ldF;          // [16][ldF]  gradient of the current block's feature rows
  float* s_do = s_dF + TT * ldF;         // [16][ldD]  masked gradient of the out_layer output
  float* s_dq = s_do + TT * ldD;         // [16][100]  d(qkv) of the tile (written by the attention backward)
  float* s_a = s_dq + TT * 100;          // [16][36] x 7 small tiles
  float* s_b = s_a + TT * LD32;
  float* s_c = s_b + TT * LD32;
  float* s_e = s_c + TT * LD32;
  float* s_dg = s_e + TT * LD32;
  float* s_gx = s_dg + TT * LD32;
  float* s_dO = s_gx + TT * LD32;        // dO of the tile (attention backward operand)
  float* s_f = s_dO + TT * LD32;         // [16][68]
  float* s_dz = s_f + TT * LD64;         // [16][68]
  float* s_red = s_dz + TT * LD64;       // [2][16][16]
  float* s_dl = s_red + 2 * 16 * 16;     // [16][8] delta of the tile
  float* s_at = s_dl + TT * 8;           // attention chunk images: 2 x (QC x QP floats + 2 x [8][QC] floats + 2 x [8][5][QC] u16)
  constexpr int AT_IMG = QC * QP, AT_T = 8 * QC, AT_U = 8 * 5 * QC / 2;   // (u16 arrays counted in floats)
  constexpr int AT_BUF = AT_IMG + 2 * AT_T + 2 * AT_U;
  const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6, wave = wave8 & 3, col0 = lane & 15, g0 = lane >> 4;
  const bool cw = tid < 256;
  const int lrow0 = (tid >> 4) & 15, lc0 = (tid & 15) * 2;   // the "LayerNorm" thread map of the first 256 threads
  const int seq = blockIdx.x % a.nseq, tile = blockIdx.x / a.nseq;
  const int m = seq / d.B, bsm = seq - m * d.B;
  const int n0 = tile * TT, nvalid = min(TT, N - n0);
  const int t0k = bsm * N + n0;
  const int64_t mo = (int64_t)m * d.mstride, rb = (int64_t)m * BN;
  const int64_t trows = a.rows;
  const DropF dr{TRAIN ? 1 : 0, d.seed, d.thresh24, d.keep_scale};
  float* pm = const_cast<float*>(a.params);
  unsigned* cnt = a.sync + seq * SYNC_LINE;
  unsigned* tmo = a.sync + a.nseq * SYNC_LINE;
  bool dead = false;
  const int nl = a.nb * 4;
  float r_acc0 = 0.f, r_acc1 = 0.f;      // residual-path gradient of the layer in flight (POSTB -> PREB)
  // Side outputs of a stage -- tape segments, column sums for the bias / LayerNorm-parameter gradients -- are written by the
  // four waves that sit out the token stages, in the same barrier interval in which waves 0-3 run the next GEMM on the same
  // tiles (tok_bwd_kernel: the computing waves did both, one after the other).
  const int htid = tid - 256;

  for (int L = nl - 1; L >= -1; L--) {
    const bool PREB = L + 1 < nl, POSTB = L >= 0, OUTB = POSTB && (L & 3) == 3;
    const int Lq = L + 1, bq = Lq >> 2, lq = Lq & 3;   // PREB's layer
    const int bp = L >> 2, lp = L & 3, bo = bp;        // POSTB's layer, OUTB's block
    int t0 = t0k;
    asm volatile("" : "+s"(t0));   // (see the forward kernel: keeps the row addresses of every stage inside the loop)
    // the lane coordinates too: loop-invariant, hipcc hoists the 24 (row, column) addresses of the Linear0 data gradient
    // (and more) out of the layer loop, spills them at kernel entry and re-loads two per row from scratch in every layer
    int col = col0, g = g0, lrow = lrow0, lc = lc0;
    asm volatile("" : "+v"(col), "+v"(g), "+v"(lrow), "+v"(lc));
    auto tok = [&](int row) { return t0 + min(row, nvalid - 1); };
    auto tape_h = [&](float* tape, int col0, const float* sT, int lds_ld, int width) __attribute__((always_inline)) {
      const int w4 = width >> 2;
      float* seg = tape + trows * col0 + (rb + t0) * width;
      for (int i = htid; i < TT * w4; i += 256) {
        const int row = i / w4, c4 = (i - row * w4) * 4;
        if (row < nvalid) *reinterpret_cast<float4*>(seg + i * 4) = *reinterpret_cast<const float4*>(sT + row * lds_ld + c4);
      }
    };
    auto colsum_h = [&](float* gb, int width, const float* sv, int ld, int tbase) __attribute__((always_inline)) {
      const int c = htid - tbase;
      if (c >= 0 && c < width) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < TT; r++) acc += sv[r * ld + c];
        atomicAdd(gb + c, acc);
      }
    };
    const bool lok = lrow < nvalid;
    const int64_t lr = (rb + tok(lrow)) * 32 + lc;
    // ------------------------------------------------------------------ requests of the token phase
    const int Kq = PREB ? DM + 32 * lq : 0;
    CFrag<12> c_q{};
    CFrag<8> c_w0[6]{};
    WFrag<2> f_w1{};
    CFrag<8> c_w2{}, c_w1{};
    CFrag<4> c_wo{};
    LnP ln1{}, ln2{};
    float p_b1 = 0.f, r_h[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, r_ob[2] = {0.f, 0.f}, r_h0[2] = {0.f, 0.f};
    TfLayerP gq{}, gp{};
    if (PREB) {
      float* w0 = chain_w0(a.cw, pm, bq, lq) + mo;
      float* rest = w0 + 32 * Kq;
      float* gw0 = chain_w0(a.cw, a.grads, bq, lq) + mo;
      float* grest = gw0 + 32 * Kq;
      gq.b0 = grest + CO_B0, gq.ln1g = grest + CO_LN1G, gq.ln1b = grest + CO_LN1B;
      if (cw) {
        const float4* slot = a.wpack + ((int64_t)m * nl + Lq) * (CH_SLOT * 64);
        cpload(c_q, slot, CB_Q + 3 * wave);
#pragma unroll
        for (int j = 0; j < 6; j++) cpload(c_w0[j], slot, CB_W0 + 2 * min(wave + 4 * j, (Kq >> 4) - 1));
        ln1 = ln_load(rest + CO_LN1G, rest + CO_LN1B);
        const float* h0 = a.save + (int64_t)Lq * a.rows * 232;
        r_h0[0] = h0[lr], r_h0[1] = h0[lr + 1];
      }
    }
    if (POSTB) {
      float* w0 = chain_w0(a.cw, pm, bp, lp) + mo;
      float* rest = w0 + 32 * (DM + 32 * lp);
      float* grest = chain_w0(a.cw, a.grads, bp, lp) + mo + 32 * (DM + 32 * lp);
      gp.bout = grest + CO_BOUT, gp.ln2g = grest + CO_LN2G, gp.ln2b = grest + CO_LN2B, gp.b1 = grest + CO_B1, gp.b2 = grest + CO_B2;
      if (cw) {
        const float4* slot = a.wpack + ((int64_t)m * nl + L) * (CH_SLOT * 64);
        pload(f_w1, slot, CH_W1 + 2 * wave, 2);
        cpload(c_w2, slot, CB_W2 + 2 * wave);
        cpload(c_w1, slot, CB_W1 + 2 * wave);
        cpload(c_wo, slot, CB_WO + wave);
        ln2 = ln_load(rest + CO_LN2G, rest + CO_LN2B);
        p_b1 = rest[CO_B1 + 16 * wave + col];
        const float* sv = a.save + (int64_t)L * a.rows * 232;
        const float* h1s = sv + a.rows * 168;
        const float* h2s = sv + a.rows * 200;
        const float* ob = sv + a.rows * 128;
        r_h[0][0] = h1s[lr], r_h[0][1] = h1s[lr + 1], r_h[1][0] = h2s[lr], r_h[1][1] = h2s[lr + 1];
        r_ob[0] = ob[lr], r_ob[1] = ob[lr + 1];
      }
    }
    CHAIN_STAMPB(0);
    if (PREB && cw) s_a[lrow * LD32 + lc] = r_h0[0], s_a[lrow * LD32 + lc + 1] = r_h0[1];
    __syncthreads();
    CHAIN_STAMPB(1);

    // ------------------------------------------------------------------ PREB(bq, lq)
    if (PREB) {
      float* tape_q = a.tape + (int64_t)Lq * a.rows * TF_TAPE_W;
      float rs1 = 0.f;
      if (cw) rs1 = ln32_keep(s_a, s_b, s_c, ln1);   // t = LN1(h0) -> s_b, xh -> s_c
      __syncthreads();
      if (!cw) {
        tape_h(tape_q, TF_T_DQ, s_dq, 100, 96);
        tape_h(tape_q, TF_T_T, s_b, LD32, 32);
      }
      f32x4 accq = zero4();
      if (cw) {  // dt = dqkv * Wqkv -> s_e
        cmma(accq, c_q, s_dq, 100, 24, 12 * (wave >> 1), 12);
        if (wave >= 2) {
#pragma unroll
          for (int r = 0; r < 4; r++) s_red[((wave & 1) * 16 + 4 * g + r) * 16 + col] = accq[r];
        }
      }
      __syncthreads();
      if (cw && wave < 2) {
#pragma unroll
        for (int r = 0; r < 4; r++)
          s_e[(4 * g + r) * LD32 + 16 * wave + col] = accq[r] + s_red[(wave * 16 + 4 * g + r) * 16 + col];
      }
      __syncthreads();
      if (cw) {  // LN1 backward + the residual-path gradient -> dh0 (s_a)
        float dh0v, dh1v;
        ln32_bwd(s_e, s_c, s_gx, rs1, ln1, dh0v, dh1v);
        s_a[lrow * LD32 + lc] = lok ? dh0v + r_acc0 : 0.f;
        s_a[lrow * LD32 + lc + 1] = lok ? dh1v + r_acc1 : 0.f;
      }
      __syncthreads();
      if (!cw) {
        colsum_h(gq.ln1g, 32, s_gx, LD32, 0);
        colsum_h(gq.ln1b, 32, s_e, LD32, 64);
        colsum_h(gq.b0, 32, s_a, LD32, 128);
        tape_h(tape_q, TF_T_DH0, s_a, LD32, 32);
      } else {
        // dF[:, 0:Kq] += dh0 * W0
#pragma unroll
        for (int j = 0; j < 6; j++) {
          const int nn = 16 * (wave + 4 * j);
          if (nn < Kq) {
            f32x4 acc = zero4();
            cmma(acc, c_w0[j], s_a, LD32, 8, 0, 8);
            const int c = nn + col;
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int row = 4 * g + r;
              const bool ok = row < nvalid;
              const float v = ok ? s_dF[row * ldF + min(c, Kq - 1)] + acc[r] : 0.f;
              if (c < Kq) {
                if (OUTB) {
                  s_do[row * ldD + c] = v;          // (lq = 0: Kq = DM) consumed below
                } else {
                  if (!POSTB) {                     // the last token phase: block 0's input gradient leaves the chip
                    if (ok) a.dF[(rb + t0 + row) * DMF + c] = v;
                  } else {
                    s_dF[row * ldF + c] = v;
                    if (c >= Kq - 32) s_dg[row * LD32 + c - (Kq - 32)] = v;
                  }
                }
              }
            }
          }
        }
      }
      __syncthreads();
    }
    CHAIN_STAMPB(2);
    if (!POSTB) break;

    // ------------------------------------------------------------------ OUTB(bo)
    if (OUTB) {
      const TfOutP po = chain_out(a.cw, pm, bo);
      const TfOutP go = chain_out(a.cw, a.grads, bo);
      const float4* oslot = a.wpack + (int64_t)d.M * nl * (CH_SLOT * 64) + ((int64_t)m * a.nb + bo) * (CO_SLOT * 64);
      const float* Fo = a.F0 + (int64_t)bo * a.rows * DMF;
      float* otape = a.otape + (int64_t)bo * a.rows * DMF;
      const uint32_t siteo = hdf_site_id(m, bo, 4, 0);
      {
        const int c4n = DMF >> 2;
        for (int i = tid; i < TT * c4n; i += CT) {
          const int row = i / c4n, c4 = (i - row * c4n) * 4;
          *reinterpret_cast<float4*>(s_F + row * ldF + c4) = *reinterpret_cast<const float4*>(Fo + (rb + tok(row)) * DMF + c4);
        }
        for (int i = tid; i < TT * DM; i += CT) {
          const int row = i / DM, c = i - row * DM, t = t0 + row;
          float v;
          if (PREB) {
            v = s_do[row * ldD + c];
          } else {
            const int64_t ai = ((int64_t)bsm * N + n0 + min(row, nvalid - 1)) * ((int64_t)d.M * DM) + (int64_t)m * DM + c;
            if (a.dtype == HDF_BF16)
              v = ST<bf16_t>::ld(reinterpret_cast<const bf16_t*>(a.d_attnall) + ai);
            else if (a.dtype == HDF_F16)
              v = ST<f16_t>::ld(reinterpret_cast<const f16_t*>(a.d_attnall) + ai);
            else
              v = reinterpret_cast<const float*>(a.d_attnall)[ai];
          }
          s_do[row * ldD + c] = row < nvalid ? v * dr.mask(siteo + 1, (uint32_t)t * DM + c) : 0.f;
        }
      }
      __syncthreads();
      float zr[4] = {0.f, 0.f, 0.f, 0.f}, mk[4] = {0.f, 0.f, 0.f, 0.f};
      if (cw) {
        {  // z = Wa F + ba (recomputed), f = gelu(z) * mask -> s_f
          f32x4 acc = zero4();
          wmma_stream_p(acc, oslot, OA_F + 24 * wave, DMF >> 4, s_F, ldF, DMF >> 2);
          const int c = 16 * wave + col;
          const float ba = po.ba[mo + c];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = 4 * g + r, t = t0 + row;
            zr[r] = acc[r] + ba;
            mk[r] = dr.mask(siteo + 0, (uint32_t)t * 64 + c);
            s_f[row * LD64 + c] = row < nvalid ? gelu_f(zr[r]) * mk[r] : 0.f;
          }
        }
        {  // df = do * Wb ; dz = df * mask * gelu'(z) -> s_dz
          f32x4 acc = zero4();
          cmma_stream_p(acc, oslot, OB_B + 16 * wave, s_do, ldD, DM >> 2);
          const int c = 16 * wave + col;
#pragma unroll
          for (int r = 0; r < 4; r++) s_dz[(4 * g + r) * LD64 + c] = acc[r] * mk[r] * gelu_grad_f(zr[r]);
        }
      }
      __syncthreads();
      if (!cw) {
        tape_h(otape, 0, s_do, ldD, DM);
        tape_h(otape, DM, s_f, LD64, 64);
        tape_h(otape, DM + 64, s_dz, LD64, 64);
        for (int c0 = 0; c0 < DM; c0 += 128) colsum_h(go.bb + mo + c0, min(128, DM - c0), s_do + c0, ldD, 0);
        colsum_h(go.ba + mo, 64, s_dz, LD64, 128);
      } else {
        // dF[:, 0:DMF] = dz * Wa  (overwrites: the first writer of block bo's feature gradient)
        for (int nn = 16 * wave; nn < DMF; nn += 64) {
          f32x4 acc = zero4();
          cmma_stream_p(acc, oslot, OA_B + 4 * (nn >> 4), s_dz, LD64, 16);
          const int c = nn + col;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = 4 * g + r;
            s_dF[row * ldF + c] = acc[r];
            if (c >= DMF - 32) s_dg[row * LD32 + c - (DMF - 32)] = row < nvalid ? acc[r] : 0.f;
          }
        }
      }
      __syncthreads();
    }

    CHAIN_STAMPB(3);
    // ------------------------------------------------------------------ POSTB(bp, lp)
    {
      float* tape_p = a.tape + (int64_t)L * a.rows * TF_TAPE_W;
      const uint32_t site0 = hdf_site_id(m, bp, lp, 0);
      const int t = t0 + lrow;
      const bool ok = lok;
      float dcur0 = 0.f, dcur1 = 0.f;
      if (cw) dcur0 = s_dg[lrow * LD32 + lc], dcur1 = s_dg[lrow * LD32 + lc + 1];
      float dres0 = 0.f, dres1 = 0.f;
#pragma unroll
      for (int pass = 1; pass >= 0; pass--) {  // pass 1: the second ff (on h2) ; pass 0: the first ff (on h1)
        __syncthreads();
        float rs = 0.f;
        if (cw) {
          s_a[lrow * LD32 + lc] = r_h[pass][0];
          s_a[lrow * LD32 + lc + 1] = r_h[pass][1];
          rs = ln32_keep(s_a, s_b, s_c, ln2);   // u -> s_b, xh -> s_c
          s_dg[lrow * LD32 + lc] = ok ? dcur0 * dr.mask(site0 + 2 + 2 * pass, (uint32_t)t * 32 + lc) : 0.f;
          s_dg[lrow * LD32 + lc + 1] = ok ? dcur1 * dr.mask(site0 + 2 + 2 * pass, (uint32_t)t * 32 + lc + 1) : 0.f;
        }
        __syncthreads();
        if (pass == 1) CHAIN_STAMPB(9);
        if (cw) {
          f32x4 accz = zero4(), accd = zero4();
          wmma(accz, f_w1, s_b, LD32, 8, 0, 2);            // z = W1 u  (tile wave)
          cmma(accd, c_w2, s_dg, LD32, 8, 0, 8);           // df = dg * W2 (tile wave)
          const int c = 16 * wave + col;
          const float b1 = p_b1;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = 4 * g + r, tt = t0 + row;
            const float z = accz[r] + b1, mkv = dr.mask(site0 + 1 + 2 * pass, (uint32_t)tt * 64 + c);
            s_f[row * LD64 + c] = row < nvalid ? gelu_f(z) * mkv : 0.f;
            s_dz[row * LD64 + c] = row < nvalid ? accd[r] * mkv * gelu_grad_f(z) : 0.f;
          }
        }
        __syncthreads();
        if (pass == 1) CHAIN_STAMPB(10);
        f32x4 accu = zero4();
        if (!cw) {
          const int c0 = pass ? TF_T_P1 : TF_T_P0;
          tape_h(tape_p, c0, s_dg, LD32, 32);
          tape_h(tape_p, c0 + 32, s_f, LD64, 64);
          tape_h(tape_p, c0 + 96, s_dz, LD64, 64);
          tape_h(tape_p, c0 + 160, s_b, LD32, 32);
          colsum_h(gp.b2, 32, s_dg, LD32, 0);
          colsum_h(gp.b1, 64, s_dz, LD64, 64);
        } else {
          // du = dz * W1 -> s_e
          cmma(accu, c_w1, s_dz, LD64, 16, 8 * (wave >> 1), 8);
          if (wave >= 2) {
#pragma unroll
            for (int r = 0; r < 4; r++) s_red[((wave & 1) * 16 + 4 * g + r) * 16 + col] = accu[r];
          }
        }
        __syncthreads();
        if (pass == 1) CHAIN_STAMPB(11);
        if (cw && wave < 2) {
#pragma unroll
          for (int r = 0; r < 4; r++)
            s_e[(4 * g + r) * LD32 + 16 * wave + col] = accu[r] + s_red[(wave * 16 + 4 * g + r) * 16 + col];
        }
        __syncthreads();
        if (pass == 1) CHAIN_STAMPB(12);
        float dh0v = 0.f, dh1v = 0.f;
        if (cw) {
          ln32_bwd(s_e, s_c, s_gx, rs, ln2, dh0v, dh1v);
          dh0v = ok ? dh0v : 0.f, dh1v = ok ? dh1v : 0.f;
        }
        __syncthreads();
        if (pass == 1) CHAIN_STAMPB(13);
        if (!cw) {
          colsum_h(gp.ln2g, 32, s_gx, LD32, 0);
          colsum_h(gp.ln2b, 32, s_e, LD32, 64);
        }
        if (pass == 1) {
          dcur0 = dh0v, dcur1 = dh1v;   // h2 feeds only the second ff: its gradient flows into ff#1's output ...
          dres0 = dh0v, dres1 = dh1v;   // ... and into the residual h1
        } else {
          dres0 += dh0v, dres1 += dh1v;
        }
      }
      // to_out: a = (Wout ob + bout) * mask ; h1 = a + h0
      __syncthreads();
      if (cw) {
        s_dg[lrow * LD32 + lc] = ok ? dres0 * dr.mask(site0 + 0, (uint32_t)t * 32 + lc) : 0.f;
        s_dg[lrow * LD32 + lc + 1] = ok ? dres1 * dr.mask(site0 + 0, (uint32_t)t * 32 + lc + 1) : 0.f;
        s_b[lrow * LD32 + lc] = r_ob[0];
        s_b[lrow * LD32 + lc + 1] = r_ob[1];
        r_acc0 = dres0, r_acc1 = dres1;   // -> PREB of this layer, after its attention backward
      }
      __syncthreads();
      CHAIN_STAMPB(14);
      f32x4 acco = zero4();
      if (!cw) {
        tape_h(tape_p, TF_T_DGO, s_dg, LD32, 32);
        colsum_h(gp.bout, 32, s_dg, LD32, 0);
      } else {
        cmma(acco, c_wo, s_dg, LD32, 8, 4 * (wave >> 1), 4);
        if (wave >= 2) {
#pragma unroll
          for (int r = 0; r < 4; r++) s_red[((wave & 1) * 16 + 4 * g + r) * 16 + col] = acco[r];
        }
      }
      __syncthreads();
      if (cw && wave < 2) {
#pragma unroll
        for (int r = 0; r < 4; r++)
          s_dO[(4 * g + r) * LD32 + 16 * wave + col] = acco[r] + s_red[(wave * 16 + 4 * g + r) * 16 + col];
      }
      __syncthreads();
      CHAIN_STAMPB(15);
      // delta = dO . ob per (token, head): attn_bwd's arithmetic (tf_dot4)
      if (tid < TT * 8) {
        const int row = tid >> 3, h = tid & 7;
        s_dl[row * 8 + h] = tf_dot4(*reinterpret_cast<const float4*>(s_dO + row * LD32 + 4 * h),
                                    *reinterpret_cast<const float4*>(s_b + row * LD32 + 4 * h));
      }
      __syncthreads();
    }

    CHAIN_STAMPB(4);
    // ------------------------------------------------------------------ publish dO | delta, per-sequence barrier
    const int par = L & 1;
    float* xbase = a.xchg + ((int64_t)par * a.rows + rb + (int64_t)bsm * N) * XW;   // this sequence's rows, this parity
    const __amdgpu_buffer_rsrc_t rx = chain_rsrc(xbase);
    float* xgseq = a.xchg + ((int64_t)par * a.nseq + seq) * a.ntile * 8 * XG_W;    // LP: this sequence's records
    // The payload is stored by waves 0-3 only (two heads each in the 16-bit modes): waves 4-7 still have the tape stores and
    // the column-sum atomics of this token phase in flight -- an atomic stays counted in vmcnt for thousands of cycles when
    // every workgroup of a modality adds to the same 32 words -- and the drain in front of the arrival waits for a storing
    // wave's WHOLE queue.  (Guideline 16 R1: every STORING wave drains; the others only join the barrier.)
    if (LP) {
      // the 8 records of the tile (XG_GA | XG_GB) and its 8 x 16 -delta values are assembled in LDS by all waves (s_f is free
      // here) and leave as ONE 16-byte write-through store per lane of waves 0-3 (4-byte sc1 stores are one fabric write
      // each: ~6x the time per byte of 16-byte ones)
      constexpr int L1 = LP ? LP : 1;
      float* s_rec = s_f;                       // [8][XG_W]: GA (64) | GB (40 words) | -delta (16) | pad
      {
        const int hd = wave8;
        float* rec = s_rec + hd * XG_W;
        const bool vrow = col < nvalid;
        rec[XG_GA + lane] = vrow ? s_dO[col * LD32 + hd * 4 + g] : 0.f;
        if (lane < 20) {
          const int c = lane >> 2, gg = lane & 3;
          float gv[4] = {0.f, 0.f, 0.f, 0.f};
          if (c < 4) {
#pragma unroll
            for (int e = 0; e < 4; e++) gv[e] = 4 * gg + e < nvalid ? s_dO[(4 * gg + e) * LD32 + hd * 4 + c] : 0.f;
          }
          reinterpret_cast<u32x2*>(rec + XG_GB)[lane] = ChainLp<L1>::four(gv[0], gv[1], gv[2], gv[3]);
        }
        if (lane < 16) rec[XG_GB + 40 + lane] = lane < nvalid ? -s_dl[lane * 8 + hd] : 0.f;
      }
      __syncthreads();
      if (tid < 8 * 26) {          // GA | GB of head tid / 26: 104 floats = 26 pieces
        const int hd = tid / 26, pc = tid - hd * 26;
        const __amdgpu_buffer_rsrc_t rr = chain_rsrc(xgseq + (int64_t)tile * 8 * XG_W);
        st16_sc1(rr, (uint32_t)((hd * XG_W + 4 * pc) * 4), *reinterpret_cast<const float4*>(s_rec + hd * XG_W + 4 * pc));
      } else if (tid < 8 * 26 + 32) {   // -delta rows of the 8 heads: 4 pieces each, head-major [parity][sequence][head][NP]
        const int i = tid - 8 * 26, hd = i >> 2, pc = i & 3;
        const __amdgpu_buffer_rsrc_t rr = chain_rsrc(a.xchg + (int64_t)2 * a.nseq * a.ntile * 8 * XG_W +
                                                     ((int64_t)par * a.nseq + seq) * 8 * (a.ntile * TT));
        st16_sc1(rr, (uint32_t)((hd * (a.ntile * TT) + n0 + 4 * pc) * 4),
                 *reinterpret_cast<const float4*>(s_rec + hd * XG_W + XG_GB + 40 + 4 * pc));
      }
    } else if (tid < TT * 10) {
      const int row = tid / 10, c4 = (tid - row * 10) * 4;
      if (row < nvalid) {
        const float4 v = c4 < 32 ? *reinterpret_cast<const float4*>(s_dO + row * LD32 + c4)
                                 : *reinterpret_cast<const float4*>(s_dl + row * 8 + (c4 - 32));
        st16_sc1(rx, (uint32_t)(((n0 + row) * XW + c4) * 4), v);
      }
    }
    // (the forward launch's operands of the attention backward are requested BEFORE the drain: their round trip passes under
    // the write-through stores')
    // operands of this head (wave) for the tile's own 16 tokens, requested before the wait (forward launch's data)
    const int head = __builtin_amdgcn_readfirstlane(wave8);   // (uniform: per-wave pointers then live in SGPRs)
    const float* sv = a.save + (int64_t)L * a.rows * 232;
    const float* qkvL = sv + a.rows * 32 + (rb + (int64_t)bsm * N) * 96;   // the sequence's rows
    const float* lseL = sv + a.rows * 160 + (rb + (int64_t)bsm * N) * 8;
    const int rown = n0 + min(col, nvalid - 1);   // this lane's token (as a query in the dQ half, as a key in the other)
    const float own_q = qkvL[(int64_t)rown * 96 + head * 4 + g];
    const float own_k = qkvL[(int64_t)rown * 96 + 32 + head * 4 + g];
    const float own_v = qkvL[(int64_t)rown * 96 + 64 + head * 4 + g];
    const float own_lse = lseL[(int64_t)rown * 8 + head];
    // 16-bit modes: everything of the attention backward that comes from the forward launch is requested before the wait:
    // the dQ loop's first PD blocks of key operands and this head's -lse row
    constexpr int PD = 8, NLS = 4;   // prefetch ring depth (blocks); 16-byte pieces per lane of an [NP <= 1024] row
    const int NPq = attn_rows(N);
    const float* fr = nullptr;
    const int cb = min(col, 4) * 4 + g;      // this lane's entry of a 16-bit B operand table (row 4: zeros)
    float ka[PD], va[PD];
    u32x2 kb[PD];
    float4 r_ls[NLS];
    auto ldk = [&](int b, int i) __attribute__((always_inline)) {
      const float* r = fr + (int64_t)min(b, a.ntile - 1) * 8 * FR_W;
      ka[i] = r[FR_KA + lane], va[i] = r[FR_VA + lane];
      kb[i] = reinterpret_cast<const u32x2*>(r + FR_KB)[cb];
    };
    if constexpr (LP != 0) {
      fr = a.frag + ((int64_t)L * a.nseq + seq) * a.ntile * 8 * FR_W + (int64_t)head * FR_W;   // + blk * 8 * FR_W
      const float* lsrow = a.frag + (int64_t)nl * a.nseq * a.ntile * 8 * FR_W + (((int64_t)L * a.nseq + seq) * 8 + head) * NPq;
#pragma unroll
      for (int i = 0; i < PD; i++) ldk(i, i);
#pragma unroll
      for (int k = 0; k < NLS; k++) r_ls[k] = *reinterpret_cast<const float4*>(lsrow + min((k * 64 + lane) * 4, NPq - 4));
    }
    chain_arrive(cnt, cw);
    CHAIN_STAMPB(5);
    // (no wait here: the dQ half needs only this tile's own dO and the forward launch's k / v -- the hand-off's latency
    // passes under it; the wait sits in front of the dK / dV half, the first reader of the siblings' records)
    CHAIN_STAMPB(6);

    // ------------------------------------------------------------------ attention backward of layer L: head = wave
    if constexpr (LP != 0) {
      // 16-bit storage modes: attn_bwd_lp_kernel's arithmetic on operand records (FR_* / XG_*), prefetch rings of PD blocks
      const int NP = NPq, nblk = a.ntile;
      const float* xg = xgseq + (int64_t)head * XG_W;                                                        // + blk * 8 * XG_W
      // (-lse log2(e), -delta) of every query of the sequence for THIS wave's head: two [NP] rows of LDS, filled by the
      // wave itself from the head-major arrays (written behind the dQ loop: no workgroup barrier)
      float* s_ls = s_at + head * 2 * NP;
      float* s_dlq = s_ls + NP;
      const __amdgpu_buffer_rsrc_t rdl = chain_rsrc(a.xchg + (int64_t)2 * a.nseq * a.ntile * 8 * XG_W +
                                                    (((int64_t)par * a.nseq + seq) * 8 + head) * NP);
      // ---- dQ of the tile's queries
      {
        const float delta = s_dl[col * 8 + head];
        const float ls = own_lse * LOG2E;
        const float bqv = own_q * (0.5f * LOG2E);
        const float bg = s_dO[col * LD32 + head * 4 + g];
        const f32x4 nl4 = {-ls, -ls, -ls, -ls}, nd4 = {-delta, -delta, -delta, -delta};
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        auto step = [&](int i, int j0, auto masked) __attribute__((always_inline)) {
          const f32x4 cs = chain_mfma4(ka[i], bqv, nl4), ct = chain_mfma4(va[i], bg, nd4);
          float ds[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            ds[u] = __builtin_amdgcn_exp2f(cs[u]) * ct[u];
            if (decltype(masked)::value) ds[u] = (j0 + 4 * g + u < N) ? ds[u] : 0.f;
          }
          acc = ChainLp<LP>::mma(ChainLp<LP>::four(ds[0], ds[1], ds[2], ds[3]), kb[i], acc);
        };
        const int nfullb = N / ATRIP;   // whole blocks
        int bb = 0;
        for (; bb + PD <= nfullb; bb += PD) {   // no mask, no bounds test: the ring's loads are clamped, never branched around
#pragma unroll
          for (int i = 0; i < PD; i++) {
            step(i, 0, std::false_type{});
            ldk(bb + i + PD, i);
          }
        }
        for (int i = 0; bb < nblk; bb++, i++) {   // up to PD - 1 whole blocks and the partial one (i < PD: static after unrolling)
#pragma unroll
          for (int k = 0; k < PD; k++)
            if (k == i) {
              if ((bb + 1) * ATRIP <= N)
                step(k, 0, std::false_type{});
              else
                step(k, bb * ATRIP, std::true_type{});
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (col < 4) s_dq[(4 * g + r) * 100 + head * 4 + col] = (4 * g + r < nvalid) ? 0.5f * acc[r] : 0.f;
      }
      chain_wait(cnt, (unsigned)(a.ntile * (nl - L)), tmo, a.ctl, dead);
      {
        float4 r_dl[NLS];
#pragma unroll
        for (int k = 0; k < NLS; k++) r_dl[k] = ld16_sc1(rdl, (uint32_t)(min((k * 64 + lane) * 4, NP - 4) * 4));
#pragma unroll
        for (int k = 0; k < NLS; k++) {
          const int i4 = (k * 64 + lane) * 4;
          if (i4 < NP) {
            *reinterpret_cast<float4*>(s_ls + i4) = r_ls[k];
            *reinterpret_cast<float4*>(s_dlq + i4) = r_dl[k];
          }
        }
      }
      CHAIN_STAMPB(7);
      // ---- dK, dV of the tile's keys
      {
        const float bk = own_k, bv = own_v;
        const float* tl = s_ls + 4 * g;
        const float* td = s_dlq + 4 * g;
        float qa[PD], ga[PD];
        u32x2 qb[PD], gb[PD];
        auto ld = [&](int b, int i) __attribute__((always_inline)) {
          const int bc = min(b, nblk - 1);
          const float* r = fr + (int64_t)bc * 8 * FR_W;
          const float* x = xg + (int64_t)bc * 8 * XG_W;
          qa[i] = r[FR_QA + lane];
          qb[i] = reinterpret_cast<const u32x2*>(r + FR_QB)[cb];
          ga[i] = __hip_atomic_load(x + XG_GA + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(x + XG_GB) + cb,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gb[i] = u32x2{(uint32_t)w, (uint32_t)(w >> 32)};
        };
#pragma unroll
        for (int i = 0; i < PD; i++) ld(i, i);
        f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
        auto step = [&](int i, int b) __attribute__((always_inline)) {
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(tl + b * ATRIP), d4 = *reinterpret_cast<const f32x4*>(td + b * ATRIP);
          const f32x4 cs = chain_mfma4(qa[i], bk, l4), ct = chain_mfma4(ga[i], bv, d4);
          const float p0 = __builtin_amdgcn_exp2f(cs[0]), p1 = __builtin_amdgcn_exp2f(cs[1]);
          const float p2 = __builtin_amdgcn_exp2f(cs[2]), p3 = __builtin_amdgcn_exp2f(cs[3]);
          dv = ChainLp<LP>::mma(ChainLp<LP>::four(p0, p1, p2, p3), gb[i], dv);
          dk = ChainLp<LP>::mma(ChainLp<LP>::four(p0 * ct[0], p1 * ct[1], p2 * ct[2], p3 * ct[3]), qb[i], dk);
        };
        int bb = 0;
        for (; bb + PD <= nblk; bb += PD) {
#pragma unroll
          for (int i = 0; i < PD; i++) {
            step(i, bb + i);
            ld(bb + i + PD, i);
          }
        }
        for (int i = 0; bb < nblk; bb++, i++) {   // (padded queries: -inf in the lse term, zeros elsewhere)
#pragma unroll
          for (int k = 0; k < PD; k++)
            if (k == i) step(k, bb);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (col < 4) {
            const bool ok = 4 * g + r < nvalid;
            s_dq[(4 * g + r) * 100 + 32 + head * 4 + col] = ok ? 0.5f * dk[r] : 0.f;
            s_dq[(4 * g + r) * 100 + 64 + head * 4 + col] = ok ? dv[r] : 0.f;
          }
        }
      }
    } else {
      const int NP = attn_rows(N), nchunk = (NP + QC - 1) / QC;
      const int srow = tid >> 3, spart = tid & 7;   // staging: row of the chunk, 16-byte part
      const bool rvalid = col < nvalid;
      // ---- dQ of the tile's queries: K | V chunks (image rows of KVP floats: K (32) | V (32) | pad)
      {
        const float4 go = *reinterpret_cast<const float4*>(s_dO + col * LD32 + head * 4);
        const float delta = s_dl[col * 8 + head];
        const float ls = own_lse * LOG2E;
        const float bqv = own_q * (0.5f * LOG2E);
        const float bg = s_dO[col * LD32 + head * 4 + g];
        float4 pk, pv;
        auto stage_load = [&](int c) __attribute__((always_inline)) {
          const int j = min(c * QC + srow, N - 1);
          pk = *reinterpret_cast<const float4*>(qkvL + (int64_t)j * 96 + 32 + 4 * spart);
          pv = *reinterpret_cast<const float4*>(qkvL + (int64_t)j * 96 + 64 + 4 * spart);
        };
        auto stage_store = [&](int c) __attribute__((always_inline)) {
          const float z = c * QC + srow < N ? 1.f : 0.f;
          float* img = s_at + (c & 1) * AT_BUF;
          const float4 kk = make_float4(pk.x * z, pk.y * z, pk.z * z, pk.w * z);
          *reinterpret_cast<float4*>(img + srow * KVP + 4 * spart) = kk;
          *reinterpret_cast<float4*>(img + srow * KVP + 32 + 4 * spart) = make_float4(pv.x * z, pv.y * z, pv.z * z, pv.w * z);
          if (LP) {
            uint16_t* kt = reinterpret_cast<uint16_t*>(img + AT_IMG + 2 * AT_T) + spart * 5 * QC;   // [head][5][QC]
            constexpr int L1 = LP ? LP : 1;
            kt[0 * QC + srow] = ChainLp<L1>::one(kk.x), kt[1 * QC + srow] = ChainLp<L1>::one(kk.y);
            kt[2 * QC + srow] = ChainLp<L1>::one(kk.z), kt[3 * QC + srow] = ChainLp<L1>::one(kk.w);
            kt[4 * QC + srow] = 0;
          }
        };
        stage_load(0);
        stage_store(0);
        __syncthreads();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};          // LP: dQ[query 4g + r][component col] (col < 4)
        f2 d01 = {0.f, 0.f}, d23 = {0.f, 0.f};     // exact: this lane's partial dQ of its query
        const f32x4 nl4 = {-ls, -ls, -ls, -ls}, nd4 = {-delta, -delta, -delta, -delta}, z4 = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nchunk; c++) {
          if (c + 1 < nchunk) stage_load(c + 1);
          const float* img = s_at + (c & 1) * AT_BUF;
          const uint16_t* kt = reinterpret_cast<const uint16_t*>(img + AT_IMG + 2 * AT_T) + head * 5 * QC + min(col, 4) * QC + 4 * g;
          const int jbase = c * QC, ntrip = min(QC, NP - jbase) / ATRIP;
          for (int k = 0; k < ntrip; k++) {
            const int jj = k * ATRIP, j0 = jbase + jj;
            const float ka = img[(jj + col) * KVP + head * 4 + g], va = img[(jj + col) * KVP + 32 + head * 4 + g];
            if (LP) {
              constexpr int L1 = LP ? LP : 1;
              const f32x4 cs = chain_mfma4(ka, bqv, nl4), ct = chain_mfma4(va, bg, nd4);
              const u32x2 kb = *reinterpret_cast<const u32x2*>(kt + jj);
              float ds[4];
#pragma unroll
              for (int u = 0; u < 4; u++) {
                ds[u] = __builtin_amdgcn_exp2f(cs[u]) * ct[u];
                if (j0 + ATRIP > N) ds[u] = (j0 + 4 * g + u < N) ? ds[u] : 0.f;
              }
              acc = ChainLp<L1>::mma(ChainLp<L1>::four(ds[0], ds[1], ds[2], ds[3]), kb, acc);
            } else {
              const f32x4 s4 = chain_mfma4(ka, bqv, z4), t4 = chain_mfma4(va, bg, z4);
#pragma unroll
              for (int u = 0; u < 4; u++) {
                const float4 kr = *reinterpret_cast<const float4*>(img + (jj + 4 * g + u) * KVP + head * 4);
                float pr = __builtin_amdgcn_exp2f(s4[u] - ls);
                if (j0 + ATRIP > N) pr = (j0 + 4 * g + u < N) ? pr : 0.f;
                const float dsv = pr * (t4[u] - delta);
                const f2 dd = {dsv, dsv};
                d01 = __builtin_elementwise_fma(dd, lo2(kr), d01);
                d23 = __builtin_elementwise_fma(dd, hi2(kr), d23);
              }
            }
          }
          if (c + 1 < nchunk) stage_store(c + 1);
          __syncthreads();
        }
        if (LP) {
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (col < 4) s_dq[(4 * g + r) * 100 + head * 4 + col] = (4 * g + r < nvalid) ? 0.5f * acc[r] : 0.f;
        } else {
#pragma unroll
          for (int off = 16; off < 64; off <<= 1) {
            d01.x += __shfl_xor(d01.x, off, 64), d01.y += __shfl_xor(d01.y, off, 64);
            d23.x += __shfl_xor(d23.x, off, 64), d23.y += __shfl_xor(d23.y, off, 64);
          }
          if (g == 0)
            *reinterpret_cast<float4*>(s_dq + col * 100 + head * 4) =
                rvalid ? make_float4(0.5f * d01.x, 0.5f * d01.y, 0.5f * d23.x, 0.5f * d23.y) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      chain_wait(cnt, (unsigned)(a.ntile * (nl - L)), tmo, a.ctl, dead);
      CHAIN_STAMPB(7);
      // ---- dK, dV of the tile's keys: q | dO chunks (image rows of QP floats), (lse, delta) per head as [8][QC] arrays
      {
        const float bk = own_k, bv = own_v;
        float4 pq, pg, pl;   // q part, dO part, and for spart < 4: lse (0, 1) / delta (2, 3) halves
        auto stage_load = [&](int c) __attribute__((always_inline)) {
          const int j = min(c * QC + srow, N - 1);
          pq = *reinterpret_cast<const float4*>(qkvL + (int64_t)j * 96 + 4 * spart);
          pg = ld16_sc1(rx, (uint32_t)((j * XW + 4 * spart) * 4));
          if (spart < 2)
            pl = *reinterpret_cast<const float4*>(lseL + (int64_t)j * 8 + 4 * spart);
          else if (spart < 4)
            pl = ld16_sc1(rx, (uint32_t)((j * XW + 32 + 4 * (spart - 2)) * 4));
        };
        auto stage_store = [&](int c) __attribute__((always_inline)) {
          const bool real = c * QC + srow < N;
          const float z = real ? 1.f : 0.f, sc = real ? 0.5f * LOG2E : 0.f;
          float* img = s_at + (c & 1) * AT_BUF;
          float* tl = img + AT_IMG;           // [8][QC] lse term
          float* td = tl + AT_T;              // [8][QC] delta term
          *reinterpret_cast<float4*>(img + srow * QP + 4 * spart) = make_float4(pq.x * sc, pq.y * sc, pq.z * sc, pq.w * sc);
          if (LP) {
            constexpr int L1 = LP ? LP : 1;
            *reinterpret_cast<float4*>(img + srow * QP + 32 + 4 * spart) = make_float4(pg.x * z, pg.y * z, pg.z * z, pg.w * z);
            uint16_t* qt = reinterpret_cast<uint16_t*>(td + AT_T) + spart * 5 * QC;
            uint16_t* gt = qt + 8 * 5 * QC;
            qt[0 * QC + srow] = ChainLp<L1>::one(pq.x * z), qt[1 * QC + srow] = ChainLp<L1>::one(pq.y * z);
            qt[2 * QC + srow] = ChainLp<L1>::one(pq.z * z), qt[3 * QC + srow] = ChainLp<L1>::one(pq.w * z);
            qt[4 * QC + srow] = 0;
            gt[0 * QC + srow] = ChainLp<L1>::one(pg.x * z), gt[1 * QC + srow] = ChainLp<L1>::one(pg.y * z);
            gt[2 * QC + srow] = ChainLp<L1>::one(pg.z * z), gt[3 * QC + srow] = ChainLp<L1>::one(pg.w * z);
            gt[4 * QC + srow] = 0;
            if (spart < 2) {
              const int h4 = 4 * spart;
              tl[(h4 + 0) * QC + srow] = real ? -pl.x * LOG2E : -INFINITY, tl[(h4 + 1) * QC + srow] = real ? -pl.y * LOG2E : -INFINITY;
              tl[(h4 + 2) * QC + srow] = real ? -pl.z * LOG2E : -INFINITY, tl[(h4 + 3) * QC + srow] = real ? -pl.w * LOG2E : -INFINITY;
            } else if (spart < 4) {
              const int h4 = 4 * (spart - 2);
              td[(h4 + 0) * QC + srow] = -z * pl.x, td[(h4 + 1) * QC + srow] = -z * pl.y;
              td[(h4 + 2) * QC + srow] = -z * pl.z, td[(h4 + 3) * QC + srow] = -z * pl.w;
            }
          } else {
            *reinterpret_cast<float4*>(img + srow * QP + 32 + 4 * spart) =
                make_float4(real ? pg.x : 0.f, real ? pg.y : 0.f, real ? pg.z : 0.f, real ? pg.w : 0.f);
            if (spart < 2) {
              const int h4 = 4 * spart;
              tl[(h4 + 0) * QC + srow] = real ? pl.x * LOG2E : INFINITY, tl[(h4 + 1) * QC + srow] = real ? pl.y * LOG2E : INFINITY;
              tl[(h4 + 2) * QC + srow] = real ? pl.z * LOG2E : INFINITY, tl[(h4 + 3) * QC + srow] = real ? pl.w * LOG2E : INFINITY;
            } else if (spart < 4) {
              const int h4 = 4 * (spart - 2);
              td[(h4 + 0) * QC + srow] = real ? pl.x : 0.f, td[(h4 + 1) * QC + srow] = real ? pl.y : 0.f;
              td[(h4 + 2) * QC + srow] = real ? pl.z : 0.f, td[(h4 + 3) * QC + srow] = real ? pl.w : 0.f;
            }
          }
        };
        stage_load(0);
        stage_store(0);
        __syncthreads();
        f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
        f2 dk01 = {0.f, 0.f}, dk23 = {0.f, 0.f}, dv01 = {0.f, 0.f}, dv23 = {0.f, 0.f};
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nchunk; c++) {
          if (c + 1 < nchunk) stage_load(c + 1);
          const float* img = s_at + (c & 1) * AT_BUF;
          const float* tl = img + AT_IMG + head * QC;
          const float* td = img + AT_IMG + AT_T + head * QC;
          const uint16_t* qt = reinterpret_cast<const uint16_t*>(img + AT_IMG + 2 * AT_T) + head * 5 * QC + min(col, 4) * QC + 4 * g;
          const uint16_t* gt = qt + 8 * 5 * QC;
          const int ibase = c * QC, ntrip = min(QC, NP - ibase) / ATRIP;
          for (int k = 0; k < ntrip; k++) {
            const int ii = k * ATRIP;
            const float qa = img[(ii + col) * QP + head * 4 + g], ga = img[(ii + col) * QP + 32 + head * 4 + g];
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(tl + ii + 4 * g), d4 = *reinterpret_cast<const f32x4*>(td + ii + 4 * g);
            if (LP) {
              constexpr int L1 = LP ? LP : 1;
              const f32x4 cs = chain_mfma4(qa, bk, l4), ct = chain_mfma4(ga, bv, d4);
              const u32x2 qb = *reinterpret_cast<const u32x2*>(qt + ii);
              const u32x2 gb = *reinterpret_cast<const u32x2*>(gt + ii);
              const float p0 = __builtin_amdgcn_exp2f(cs[0]), p1 = __builtin_amdgcn_exp2f(cs[1]);
              const float p2 = __builtin_amdgcn_exp2f(cs[2]), p3 = __builtin_amdgcn_exp2f(cs[3]);
              dv = ChainLp<L1>::mma(ChainLp<L1>::four(p0, p1, p2, p3), gb, dv);
              dk = ChainLp<L1>::mma(ChainLp<L1>::four(p0 * ct[0], p1 * ct[1], p2 * ct[2], p3 * ct[3]), qb, dk);
            } else {
              const f32x4 cs = chain_mfma4(qa, bk, z4), ct = chain_mfma4(ga, bv, z4);
#pragma unroll
              for (int u = 0; u < 4; u++) {
                const float4 qr = *reinterpret_cast<const float4*>(img + (ii + 4 * g + u) * QP + head * 4);
                const float4 gr = *reinterpret_cast<const float4*>(img + (ii + 4 * g + u) * QP + 32 + head * 4);
                const float pr = __builtin_amdgcn_exp2f(cs[u] - l4[u]);   // padded queries: exp2(-inf) = 0
                const f2 pp = {pr, pr};
                dv01 = __builtin_elementwise_fma(pp, lo2(gr), dv01);
                dv23 = __builtin_elementwise_fma(pp, hi2(gr), dv23);
                const float dsv = pr * (ct[u] - d4[u]);
                const f2 dd = {dsv, dsv};
                dk01 = __builtin_elementwise_fma(dd, lo2(qr), dk01);
                dk23 = __builtin_elementwise_fma(dd, hi2(qr), dk23);
              }
            }
          }
          if (c + 1 < nchunk) stage_store(c + 1);
          __syncthreads();
        }
        if (LP) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            if (col < 4) {
              const bool ok = 4 * g + r < nvalid;
              s_dq[(4 * g + r) * 100 + 32 + head * 4 + col] = ok ? 0.5f * dk[r] : 0.f;
              s_dq[(4 * g + r) * 100 + 64 + head * 4 + col] = ok ? dv[r] : 0.f;
            }
          }
        } else {
#pragma unroll
          for (int off = 16; off < 64; off <<= 1) {
            dk01.x += __shfl_xor(dk01.x, off, 64), dk01.y += __shfl_xor(dk01.y, off, 64);
            dk23.x += __shfl_xor(dk23.x, off, 64), dk23.y += __shfl_xor(dk23.y, off, 64);
            dv01.x += __shfl_xor(dv01.x, off, 64), dv01.y += __shfl_xor(dv01.y, off, 64);
            dv23.x += __shfl_xor(dv23.x, off, 64), dv23.y += __shfl_xor(dv23.y, off, 64);
          }
          if (g == 0) {
            const float un = 1.f / LOG2E;  // the staged queries carry log2(e)
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(s_dq + col * 100 + 32 + head * 4) =
                rvalid ? make_float4(dk01.x * un, dk01.y * un, dk23.x * un, dk23.y * un) : z;
            *reinterpret_cast<float4*>(s_dq + col * 100 + 64 + head * 4) =
                rvalid ? make_float4(dv01.x, dv01.y, dv23.x, dv23.y) : z;
          }
        }
      }
    }
    CHAIN_STAMPB(8);
  }
  // a workgroup that gave up at a barrier (chain_wait) overwrites its rows of block 0's input gradient with NaN: the patch
  // embedding's gradients are then NaN (and plan.hip's chain_poison_grads_kernel marks the head of the gradient buffer)
  if (chain_any_dead(dead, reinterpret_cast<unsigned*>(s_red))) {
    const float qnan = __builtin_nanf("");
    for (int i = tid; i < nvalid * DM; i += CT) {
      const int row = i / DM, c = i - row * DM;
      a.dF[(rb + t0k + row) * DMF + c] = qnan;
    }
  }
}

size_t chain_bwd_lds(const TfDims& d, int dtype) {
  const size_t tiles = (size_t)2 * TT * (d.DMF + 4) + TT * (d.DM + 4) + TT * 100 + 7 * TT * LD32 + 2 * TT * LD64 + 2 * 16 * 16 + TT * 8;
  const size_t att = dtype == HDF_F32 ? (size_t)2 * (QC * QP + 2 * 8 * QC + 2 * (8 * 5 * QC / 2)) : (size_t)2 * 8 * attn_rows(d.N);
  return (tiles + att) * sizeof(float);
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
bool tf_chain_supported(const TfDims& d);
size_t tf_chain_frag_bytes(const TfDims& d, int nb) {
  const size_t nt = ceil_div(d.N, TT);
  return (size_t)nb * 4 * d.M * d.B * (nt * 8 * FR_W + 8 * nt * TT) * sizeof(float);   // records + the -lse rows
}
bool tf_chain_backward_supported(const TfDims& d, int dtype) {
  // (16-bit modes: the hand-off records must fit tf_scratch = rows x 160 floats; whole tiles always do)
  return tf_chain_supported(d) && chain_bwd_lds(d, dtype) <= LDS_LIMIT_F &&
         (dtype == HDF_F32 || ((size_t)2 * d.M * d.B * ceil_div(d.N, TT) * (8 * XG_W + 8 * TT) <= (size_t)d.M * d.B * d.N * 160 &&
                               attn_rows(d.N) <= 1024));
}
size_t tf_chain_wpack_bytes(const TfDims& d, int nb) {
  return ((size_t)d.M * nb * 4 * CH_SLOT + (size_t)d.M * nb * CO_SLOT) * 64 * sizeof(float4);
}

// what the kernels can address, whatever the grid
bool tf_chain_shape_ok(const TfDims& d) {
  return d.DM % 32 == 0 && d.DM >= 32 && d.DM <= 256 && d.N >= 1 && (int64_t)d.N * 96 * 4 < ((int64_t)1 << 31);
}
// ... and every workgroup of the grid resident at once: one compute unit each (512 threads at ~250 registers fill a unit's
// register files), so tiles <= the units the launch may assume -- hdf_cu_budget(), capped by the device's own count
bool tf_chain_supported(const TfDims& d) {
  return tf_chain_shape_ok(d) && d.M * d.B * ceil_div(d.N, TT) <= hdf_cu_budget();
}

int tf_chain_pack(const TfDims& d, const TfChainP& cp, int nb, const float* params, void* wpack, hipStream_t st) {
  ChainW w{};
  HDF_CHECK_ARG(chain_digest(cp, d.DM, w), "transformer chain: irregular parameter layout");
  hipLaunchKernelGGL(tf_chain_pack_kernel, dim3(nb * 4 + nb, d.M, 8), dim3(256), 0, st, w, params, d.mstride, d.DM, nb * 4,
                     reinterpret_cast<float4*>(wpack));
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

// At most ONE persistent transformer launch in flight per device and process: two of them resident together
// -- two plans or models on different streams -- can each hold a part of the compute units and wait for siblings that are
// never dispatched.  Every launch therefore waits for the event of the previous persistent launch of the process on the
// same device and records its own, under a mutex (launches of one stream are ordered anyway: the wait is then free).
// Another PROCESS on the same device is outside this chain: that case ends in the give-up path of chain_wait.
namespace {
struct ChainSerial {
  std::mutex mu;
  hipEvent_t last[16] = {};
  bool armed[16] = {};
};
ChainSerial g_chain_serial;
// call with the launch parameters ready: orders `st` behind the previous persistent launch; `done` must be called right
// after the launch (both under the caller-held lock)
int chain_serial_enter(hipStream_t st, int& dev) {
  dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (g_chain_serial.armed[dev] && hipStreamWaitEvent(st, g_chain_serial.last[dev], 0) != hipSuccess) {
    hdf_set_error("transformer chain: could not order the launch behind the previous persistent launch");
    return HDF_ERR_HIP;
  }
  return HDF_OK;
}
int chain_serial_done(hipStream_t st, int dev) {
  if (!g_chain_serial.last[dev] &&
      hipEventCreateWithFlags(&g_chain_serial.last[dev], hipEventDisableTiming) != hipSuccess) {
    g_chain_serial.last[dev] = nullptr;
    hdf_set_error("transformer chain: could not create the serialisation event");
    return HDF_ERR_HIP;
  }
  if (hipEventRecord(g_chain_serial.last[dev], st) != hipSuccess) {
    hdf_set_error("transformer chain: could not record the serialisation event");
    return HDF_ERR_HIP;
  }
  g_chain_serial.armed[dev] = true;
  return HDF_OK;
}
}  // namespace

int tf_chain_forward(const TfDims& d, const TfChainP& cp, int nb, const float* params, float* F0, float* save,
                     void* attnall, unsigned* sync, void* wpack, float* frag, int dtype, hipStream_t st,
                     const TfChainCtl& ctl) {
  HDF_CHECK_ARG(tf_chain_shape_ok(d), "transformer chain: shape not supported (M %d B %d N %d DM %d)", d.M, d.B, d.N, d.DM);
  ChainFwd a{};
  HDF_CHECK_ARG(chain_digest(cp, d.DM, a.cw), "transformer chain: irregular parameter layout");
  a.d = d, a.params = params, a.F0 = F0, a.save = save, a.attnall = attnall, a.sync = sync;
  a.wpack = reinterpret_cast<const float4*>(wpack), a.dtype = dtype, a.ctl = ctl;
  a.frag = dtype == HDF_F32 ? nullptr : frag;
  a.nb = nb, a.ntile = ceil_div(d.N, TT), a.nseq = d.M * d.B, a.rows = (int64_t)d.M * d.B * d.N;
  const size_t shm = chain_fwd_lds(d);
  HDF_CHECK_ARG(shm <= LDS_LIMIT_F, "transformer chain: %zu B of LDS", shm);
  HDF_CHECK_ARG(dtype == HDF_F32 || dtype == HDF_BF16 || dtype == HDF_F16, "unsupported dtype %d", dtype);
  HDF_TRY(chain_allow_lds(tf_chain_fwd_kernel<true>, shm));
  HDF_TRY(chain_allow_lds(tf_chain_fwd_kernel<false>, shm));
  std::lock_guard<std::mutex> lock(g_chain_serial.mu);
  int dev = 0;
  HDF_TRY(chain_serial_enter(st, dev));
  hipError_t e = hipMemsetAsync(sync, 0, tf_chain_sync_bytes(d), st);
  if (e != hipSuccess) {
    hdf_set_error("transformer chain: hipMemsetAsync failed: %s", hipGetErrorString(e));
    return HDF_ERR_HIP;
  }
  const dim3 grid(a.nseq * a.ntile);
  if (d.training)
    hipLaunchKernelGGL(tf_chain_fwd_kernel<true>, grid, dim3(CT), shm, st, a);
  else
    hipLaunchKernelGGL(tf_chain_fwd_kernel<false>, grid, dim3(CT), shm, st, a);
  HDF_LAUNCH_CHECK();
  return chain_serial_done(st, dev);
}

int tf_chain_backward(const TfDims& d, const TfChainP& cp, int nb, const float* params, float* grads, const float* F0,
                      const float* save, float* dF, const void* d_attnall, float* tape, float* otape, float* xchg,
                      const float* frag, const void* wpack, unsigned* sync, int dtype, hipStream_t st,
                      const TfChainCtl& ctl) {
  HDF_CHECK_ARG(tf_chain_backward_supported(d, dtype), "transformer chain: shape not supported (M %d B %d N %d DM %d)", d.M, d.B, d.N, d.DM);
  ChainBwd a{};
  HDF_CHECK_ARG(chain_digest(cp, d.DM, a.cw), "transformer chain: irregular parameter layout");
  HDF_CHECK_ARG(dtype == HDF_F32 || dtype == HDF_BF16 || dtype == HDF_F16, "unsupported dtype %d", dtype);
  a.d = d, a.params = params, a.grads = grads, a.F0 = F0, a.save = save, a.dF = dF, a.d_attnall = d_attnall;
  a.tape = tape, a.otape = otape, a.xchg = xchg, a.sync = sync, a.dtype = dtype, a.frag = frag;
  a.wpack = reinterpret_cast<const float4*>(wpack), a.ctl = ctl;
  HDF_CHECK_ARG(dtype == HDF_F32 || frag, "transformer chain backward: no operand records");
  a.nb = nb, a.ntile = ceil_div(d.N, TT), a.nseq = d.M * d.B, a.rows = (int64_t)d.M * d.B * d.N;
  const size_t shm = chain_bwd_lds(d, dtype);
  HDF_CHECK_ARG(shm <= LDS_LIMIT_F, "transformer chain backward: %zu B of LDS", shm);
#define CHAIN_BWD_LDS(TR, LPV) \
  if ((d.training != 0) == TR && dtype == LPV) HDF_TRY(chain_allow_lds(tf_chain_bwd_kernel<TR, LPV>, shm));
  CHAIN_BWD_LDS(true, 0)
  CHAIN_BWD_LDS(true, 1)
  CHAIN_BWD_LDS(true, 2)
  CHAIN_BWD_LDS(false, 0)
  CHAIN_BWD_LDS(false, 1)
  CHAIN_BWD_LDS(false, 2)
#undef CHAIN_BWD_LDS
  std::lock_guard<std::mutex> lock(g_chain_serial.mu);
  int dev = 0;
  HDF_TRY(chain_serial_enter(st, dev));
  hipError_t e = hipMemsetAsync(sync, 0, tf_chain_sync_bytes(d), st);
  if (e != hipSuccess) {
    hdf_set_error("transformer chain: hipMemsetAsync failed: %s", hipGetErrorString(e));
    return HDF_ERR_HIP;
  }
  const dim3 grid(a.nseq * a.ntile);
#define CHAIN_BWD_CASE(TR, LPV)                \
  if ((d.training != 0) == TR && dtype == LPV) \
    hipLaunchKernelGGL((tf_chain_bwd_kernel<TR, LPV>), grid, dim3(CT), shm, st, a);
  CHAIN_BWD_CASE(true, 0)
  CHAIN_BWD_CASE(true, 1)
  CHAIN_BWD_CASE(true, 2)
  CHAIN_BWD_CASE(false, 0)
  CHAIN_BWD_CASE(false, 1)
  CHAIN_BWD_CASE(false, 2)
#undef CHAIN_BWD_CASE
  HDF_LAUNCH_CHECK();
  return chain_serial_done(st, dev);
}
