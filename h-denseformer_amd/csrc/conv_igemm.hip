// Implicit-GEMM 3x3x3 convolution family on the gfx950 matrix cores.
//
// One kernel template covers
//   mode 0  Conv3d(k3,s1,p1)            -- forward of BasicConv3d / UpConv convs (HDenseFormer.py:151,167)
//                                          and, with flipped+transposed packed weights, their dgrad
//   mode 1  Conv3d(k3,s2,p1) gather     -- dgrad of ConvTranspose3d(k3,s2,p1,op1)   (HDenseFormer.py:211,215,219)
//   mode 2  ConvTranspose3d(k3,s2,p1,op1) forward, one launch z-slice per output parity class
// plus the weight-gradient kernel (wgrad) for stride 1 and 2.
//
// Data layout: activations are channels-last (NDHWC) with an explicit voxel pitch, storage bf16 or
// f32.  GEMM view: M = output voxels of a 3-D tile, N = output channels, K = taps x input channels.
// A workgroup (4 waves) stages the input halo box of its tile in LDS, 64 bytes of channels per
// voxel row (32 bf16 / 16 f32) padded to an 80-byte pitch (conflict-free for ds_read_b128), applies
// the producer's InstanceNorm+ReLU on the way in (so normalised activations are never written to
// HBM), then walks taps x channel chunks issuing v_mfma_f32_32x32x16_bf16 (bf16 storage) or
// v_mfma_f32_32x32x2_f32 (f32 storage: exact fp32 fma chains, the parity path).  The epilogue adds
// bias, stores the raw conv output once and emits per-tile (sum, sum^2) partials for the following
// InstanceNorm, reduced deterministically by in_finalize (no float atomics).
#include "conv_igemm.h"
#include "conv_tile.h"
#include <type_traits>

namespace {

constexpr int PITCH = 80;  // LDS bytes per voxel row: 64 payload + 16 pad

// Stage a box of voxels (channels [c0, c0 + chunk_elems)) of a pitched NDHWC tensor into LDS with the
// optional per-(n,channel) affine(+relu) transform.  Voxels outside the tensor and channels >= C
// become zeros (zero padding applies to the TRANSFORMED activation).
template <typename T, int BD, int BH, int BW, int ROWB /*payload bytes per row*/, int LPITCH>
__device__ __forceinline__ void stage_box(char* lds, const T* __restrict__ src, int64_t pitch, int C, int n, int D,
                                          int H, int W, int oz, int oy, int ox, int c0, int row_bytes,
                                          const float* __restrict__ scale, const float* __restrict__ shift, int relu) {
  constexpr int EPC = ST<T>::EPC;
  const int cpv = row_bytes >> 4;  // 16-B chunks per voxel row (power of two)
  const int cpv_shift = (cpv == 32) ? 5 : (cpv == 16) ? 4 : (cpv == 8) ? 3 : (cpv == 4) ? 2 : (cpv == 2) ? 1 : 0;
  const int total = (BD * BH * BW) << cpv_shift;
  const int part = threadIdx.x & (cpv - 1);  // constant per thread (256 % cpv == 0)
  const int cbase = c0 + part * EPC;
  float sc[EPC], sh[EPC];
  const bool xf = (scale != nullptr);
  if (xf) {
#pragma unroll
    for (int e = 0; e < EPC; e++) {
      bool ok = (cbase + e) < C;
      sc[e] = ok ? scale[(int64_t)n * C + cbase + e] : 0.f;
      sh[e] = ok ? shift[(int64_t)n * C + cbase + e] : 0.f;
    }
  }
  const bool chan_ok = cbase < C;  // C is a multiple of EPC
  // Batches of U chunks per thread: all U loads are issued back to back (UNCONDITIONAL, from a clamped address:
  // a per-element `if (ok) v = load` makes hipcc branch around every load and wait for each one in turn --
  // cdna_hip_programming.md, projection-GEMM trap (c)), then transformed and written to LDS.
  constexpr int U = 5;
  for (int id0 = threadIdx.x; id0 < total; id0 += 256 * U) {
    u32x4 v[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      int id = min(id0 + 256 * u, total - 1);
      int vox = id >> cpv_shift;
      int bz = vox / (BH * BW);
      int rem = vox - bz * (BH * BW);
      int by = rem / BW;
      int bx = rem - by * BW;
      int iz = oz + bz, iy = oy + by, ix = ox + bx;
      ok[u] = chan_ok && (unsigned)iz < (unsigned)D && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const T* p = ok[u] ? src + ((((int64_t)n * D + iz) * H + iy) * W + ix) * pitch + cbase : src;
      v[u] = *reinterpret_cast<const u32x4*>(p);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      int id = id0 + 256 * u;
      u32x4 w = v[u];
      if (xf) {
        float f[EPC];
        ST<T>::unpack(w, f);
#pragma unroll
        for (int e = 0; e < EPC; e++) {
          f[e] = f[e] * sc[e] + sh[e];
          if (relu) f[e] = fmaxf(f[e], 0.f);
        }
        w = ST<T>::pack(f);
      }
      if (!ok[u]) w = u32x4{0u, 0u, 0u, 0u};
      if (id < total) *reinterpret_cast<u32x4*>(lds + (id >> cpv_shift) * LPITCH + part * 16) = w;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward-style kernel (modes 0,1,2)
// FLAT (round 6): the 2-D operators of models/HDenseFormer_2D.py (Conv2d k3, ConvTranspose2d k3 s2 p1 op1 and its data
// gradient) on tensors of depth 1 -- TD = 1 tiles, only the 9 taps of the centre depth plane of the [27][CoutP][Cin] panel
// (a Conv2d kernel sits on depth tap 1 of the Conv3d panel: plan.hip "2-D embedding"), a one-plane box, and a depth axis
// that is never strided (stride-2 gather and transposed conv stride y and x only: 4 output-parity classes, not 8).  For the
// stride-1 conv this is exactly the 3-D operator on a depth-1 volume (the other 18 taps meet zero padding only).
template <typename T, int TD, int TH, int TW, int WM, int WN, int MB, int S, bool CONVT, bool FLAT = false>
// (256, 2): two workgroups per compute unit.  Left to itself hipcc let the big-tile instantiations grow past 256 registers
// (276 for the 256-voxel x 64-channel tile: one workgroup per unit, nobody to run while a workgroup stages its next channel
// chunk); with the bound it fits them into 178-201 without scratch.  Same-box pairs (round 6, call c37): bench step -0.09 ms
// (4 of 4), the 2-D PI-CAI step 16.8 -> 15.95 ms.
// The flat (2-D) stride-1 instantiations take three (<= 168 registers, no scratch; their step -0.1 ms of 15.9; the stride-2
// gather would spill); for the 3-D ones three cost the
// bench step +0.07 ms (call c38).
#ifndef HDF_IGEMM_WG_PER_CU   // (A/B builds)
#define HDF_IGEMM_WG_PER_CU ((FLAT && S == 1) ? 3 : 2)
#endif
__global__ __launch_bounds__(256, HDF_IGEMM_WG_PER_CU) void conv_igemm_kernel(ConvArgs a) {
  if (a.prio) HDF_LIGHT_PRIO();
  static_assert(WM * WN == 4, "4 waves");
  static_assert(WM * MB * 32 == TD * TH * TW, "tile/wave decomposition");
  static_assert(!FLAT || TD == 1, "flat tiles are one voxel deep");
  constexpr int BD = FLAT ? 1 : (CONVT ? TD + 1 : S * (TD - 1) + 3);
  constexpr int BH = CONVT ? TH + 1 : S * (TH - 1) + 3;
  constexpr int BW = CONVT ? TW + 1 : S * (TW - 1) + 3;
  constexpr int SS = CONVT ? 1 : S;
  constexpr int ESZ = sizeof(T);
  __shared__ __attribute__((aligned(16))) char lds[BD * BH * BW * PITCH + 4 * 32 * 2 * 4];
  float* s_red = reinterpret_cast<float*>(lds + BD * BH * BW * PITCH);  // [WN*32][2]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  // tile space: output voxels (conv) or input voxels m (convT)
  const int Td = CONVT ? a.Di : a.Do, Th = CONVT ? a.Hi : a.Ho, Tw = CONVT ? a.Wi : a.Wo;
  const int ntz = (Td + TD - 1) / TD, nty = (Th + TH - 1) / TH, ntx = (Tw + TW - 1) / TW;
  int t = blockIdx.x;
  const int tx = t % ntx;
  t /= ntx;
  const int ty = t % nty;
  t /= nty;
  const int tz = t % ntz;
  const int n = t / ntz;
  const int z0 = tz * TD, y0 = ty * TH, x0 = tx * TW;
  const int n_base = blockIdx.y * (WN * 32) + wn * 32;
  const int cls = CONVT ? blockIdx.z : 0;
  const int pz = FLAT ? 0 : (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
  const int ntapz = FLAT ? 1 : (CONVT ? (pz ? 2 : 1) : 3), ntapy = CONVT ? (py ? 2 : 1) : 3, ntapx = CONVT ? (px ? 2 : 1) : 3;

  const int oz = (CONVT || FLAT) ? z0 : S * z0 - 1, oy = CONVT ? y0 : S * y0 - 1, ox = CONVT ? x0 : S * x0 - 1;

  // MFMA row -> tile voxel.  Tile order (32 consecutive voxels per M-block) put the 16 lanes of a ds_read_b128 service
  // group (conv_tile.h: ds128_group) on box rows that repeat mod 16 -- with the 80-byte pitch that is the same 16-byte
  // bank slot up to three times (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.66 in every round's kernel table: the A
  // fragment reads of this kernel took 3x their LDS cycles).  Where the tile shape allows it the M-block is a slab whose
  // groups see 16 distinct rows mod 16:
  //  ZXMAP (8-wide tiles 4 deep, stride 1): M-block = one y, rows -> (z, x) as in conv_ws2_kernel (a group = z {0,2} or
  //        {1,3}: 2 BH BW = 8 mod 16 for BH BW = 100 and 60)
  //  YXMAP (flat 16-wide tiles): M-block = two y rows, one whole x row of 16 per group
  // Which voxel an accumulator row stands for changes nothing in a voxel's own arithmetic; only the order in which the
  // InstanceNorm partial sums of a workgroup add their voxels moves.
#ifdef HDF_IGEMM_LINEAR_MAP   // (A/B builds: tile order everywhere)
  constexpr bool ZXMAP = false, YXMAP = false;
#else
  constexpr bool ZXMAP = !CONVT && !FLAT && S == 1 && TD == 4 && TW == 8 && WM * MB == TH && (2 * BH * BW) % 16 == 8;
  constexpr bool YXMAP = FLAT && SS == 1 && TW == 16 && 2 * WM * MB == TH;
#endif
  auto mrow_voxel = [](int mblk, int row, int& lz, int& ly, int& lx) {
    if constexpr (ZXMAP) {
      ws_row_to_zx(row, lz, lx);
      ly = mblk;
    } else if constexpr (YXMAP) {
      int g2;
      ds128_group(row, g2, lx);
      lz = 0;
      ly = 2 * mblk + g2;
    } else {
      const int lin = mblk * 32 + row;
      lz = lin / (TH * TW), ly = (lin / TW) % TH, lx = lin % TW;
    }
  };
  int rowbase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; mb++) {
    int lz, ly, lx;
    mrow_voxel(wm * MB + mb, r, lz, ly, lx);
    rowbase[mb] = (((FLAT ? 0 : SS * lz)) * BH + SS * ly) * BW + SS * lx;
  }
  const bool n_active = n_base < a.CoutP;  // wave-uniform
  const char* wrow = reinterpret_cast<const char*>(a.w) + ((int64_t)(n_base + r) * a.Cin) * ESZ + h * 16;
  const int64_t wtap_stride = (int64_t)a.CoutP * a.Cin * ESZ;
  const int chunk_elems_max = 64 / ESZ;

  // weight fragment addressing: row-major panels [27][CoutP][Cin] or fragment-major (a.wfrag, see below)
  const char* const wfr = a.wfrag ? reinterpret_cast<const char*>(a.w) +
                                        (int64_t)(n_base >> 5) * ((a.Cin * ESZ) >> 5) * 1024 + r * 32 + h * 16
                                  : wrow;
  const int cstride = a.wfrag ? 2048 : 64, fstride = a.wfrag ? 1024 : 32;  // per 64-byte chunk / per 32-byte step

  bool done = false;
  f32x16 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[mb][i] = 0.f;
  if constexpr (!CONVT) {
    // Conv (stride 1 or the stride-2 gather) with whole 64-byte channel chunks: the 27 taps x 2 fragment steps of a chunk are unrolled and
    // software-pipelined.  The weight fragments come straight from L2 (~500+ cycles): a register ring keeps RING
    // steps of them in flight, across chunk boundaries too (the loads of the next chunk's first steps are issued
    // before its box is staged); the A fragments of step s+1 are read from LDS under the MFMAs of step s.  The
    // plain loop below waited for one L2 round trip per step, which at the low resolutions (one or two workgroups
    // per CU, nothing to switch to) left the matrix pipe idle 70-90 % of the time.
    if ((a.Cin * ESZ) % 64 == 0) {
      constexpr int NS = FLAT ? 18 : 54;           // fragment steps per chunk (taps x 2)
      constexpr int TAP0 = FLAT ? 9 : 0;           // first tap of the panel this kernel uses (flat: the centre depth plane)
#ifndef V1_RING_BIG
#define V1_RING_BIG 6
#endif
#ifndef V1_RING_SMALL
#define V1_RING_SMALL 9
#endif
      constexpr int RING = (MB >= 4) ? V1_RING_BIG : V1_RING_SMALL;  // divides 54: the ring position is the same in every chunk
      const int nchunk_all = a.Cin * ESZ / 64;
      // split-K (a.ksplit > 1: low-resolution layers with fewer tiles than CUs): this workgroup contracts chunks
      // [c_lo, c_hi) and leaves an fp32 partial tile to conv_ksplit_reduce_kernel
      const int kz = a.ksplit > 1 ? (int)blockIdx.z : 0;
      const int c_lo = a.ksplit > 1 ? kz * (nchunk_all / a.ksplit) : 0;
      const int nchunk = a.ksplit > 1 ? c_lo + nchunk_all / a.ksplit : nchunk_all;
      u32x4 bq[RING];
      // fragment-major panels (a.wfrag, see hdf_conv_weight_layout): the 32 rows x 32 B of one fragment step are one
      // contiguous 1 KB block, steps of a row follow each other, then the next 32-channel block; a tap plane has
      // the same size in both layouts.  With row-major panels a fragment load touched 32 cache lines for 1 KB and
      // the vector cache's line rate, not latency, bounded the low-resolution layers.
      auto b_load = [&](int chunk, int s_) -> u32x4 {
        const int tap = TAP0 + (s_ >> 1), fs = s_ & 1;
        const char* p = wfr + tap * wtap_stride + (int64_t)chunk * cstride + fs * fstride;
        return *reinterpret_cast<const u32x4*>(n_active ? p : reinterpret_cast<const char*>(a.w));
      };
      auto a_off = [&](int s_) {
        const int tap = s_ >> 1, fs = s_ & 1;   // (flat: tap = 3 ky + kx in the one staged plane)
        return ((tap / 9 * BH + (tap / 3) % 3) * BW + tap % 3) * PITCH + h * 16 + fs * 32;
      };
#pragma unroll
      for (int k = 0; k < RING; k++) bq[k] = b_load(c_lo, k);
      for (int chunk = c_lo; chunk < nchunk; chunk++) {
        if (chunk > c_lo) __syncthreads();
#ifdef V1_DBG_NOSTAGE  // attribution build: only the first chunk is staged
        if (chunk == c_lo)
#endif
        stage_box<T, BD, BH, BW, 64, PITCH>(lds, reinterpret_cast<const T*>(a.in), a.in_pitch, a.Cin, n, a.Di, a.Hi,
                                            a.Wi, oz, oy, ox, chunk * chunk_elems_max, 64, a.in_scale, a.in_shift,
                                            a.in_relu);
        __syncthreads();
        u32x4 af[2][MB];
#pragma unroll
        for (int mb = 0; mb < MB; mb++) af[0][mb] = *reinterpret_cast<const u32x4*>(lds + rowbase[mb] * PITCH + a_off(0));
        const bool more = chunk + 1 < nchunk;
#pragma unroll
        for (int s_ = 0; s_ < NS; s_++) {
          if (s_ + 1 < NS) {
#pragma unroll
            for (int mb = 0; mb < MB; mb++)
              af[(s_ + 1) & 1][mb] = *reinterpret_cast<const u32x4*>(lds + rowbase[mb] * PITCH + a_off(s_ + 1));
          }
          const u32x4 bcur = bq[s_ % RING];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mb = 0; mb < MB; mb++) Mma<T>::run(af[s_ & 1][mb], bcur, acc[mb]);
          // refill the slot just consumed: RING steps ahead, in this chunk or the next one
#ifndef V1_DBG_NOB  // attribution build: the weight ring is never refilled
          if (s_ + RING < NS)
            bq[s_ % RING] = b_load(chunk, s_ + RING);
          else if (more)
            bq[s_ % RING] = b_load(chunk + 1, s_ + RING - NS);
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      done = true;
    }
  }
  for (int c0 = 0; !done && c0 < a.Cin; c0 += chunk_elems_max) {
    const int chunk_elems = min(chunk_elems_max, a.Cin - c0);
    const int row_bytes = chunk_elems * ESZ;  // 64 or 32
    if (c0 > 0) __syncthreads();
    stage_box<T, BD, BH, BW, 64, PITCH>(lds, reinterpret_cast<const T*>(a.in), a.in_pitch, a.Cin, n, a.Di, a.Hi, a.Wi,
                                        oz, oy, ox, c0, row_bytes, a.in_scale, a.in_shift, a.in_relu);
    __syncthreads();
    if (n_active) {
      const int nfs = row_bytes >> 5;  // fragment steps (32 B each) in this chunk
      for (int jz = 0; jz < ntapz; jz++) {
        const int offz = FLAT ? 0 : (CONVT ? (pz ? 1 - jz : 0) : jz), wz = FLAT ? 1 : (CONVT ? (pz ? 2 * jz : 1) : jz);
        for (int jy = 0; jy < ntapy; jy++) {
          const int offy = CONVT ? (py ? 1 - jy : 0) : jy, wy = CONVT ? (py ? 2 * jy : 1) : jy;
#pragma unroll
          for (int jx = 0; jx < 3; jx++) {
            if (jx < ntapx) {
              const int offx = CONVT ? (px ? 1 - jx : 0) : jx, wx = CONVT ? (px ? 2 * jx : 1) : jx;
              const int tapoff = ((offz * BH + offy) * BW + offx) * PITCH + h * 16;
              const char* wp = wfr + ((wz * 3 + wy) * 3 + wx) * wtap_stride + (int64_t)(c0 * ESZ / 64) * cstride;
              for (int fs = 0; fs < nfs; fs++) {
                u32x4 bfrag = *reinterpret_cast<const u32x4*>(wp + fs * fstride);
                u32x4 afrag[MB];
#pragma unroll
                for (int mb = 0; mb < MB; mb++)
                  afrag[mb] = *reinterpret_cast<const u32x4*>(lds + rowbase[mb] * PITCH + tapoff + fs * 32);
#pragma unroll
                for (int mb = 0; mb < MB; mb++) Mma<T>::run(afrag[mb], bfrag, acc[mb]);
              }
            }
          }
        }
      }
    }
  }

  // ------------------------------------------------------------------------------- epilogue
  const int ch = n_base + r;
  if constexpr (!CONVT) {
    if (a.ksplit > 1) {  // fp32 partial tile [split][voxel][CoutP]: bias, storage rounding and statistics in the reduce pass
      if (n_active) {
        const int64_t mtot = (int64_t)a.N * a.Do * a.Ho * a.Wo;
#pragma unroll
        for (int mb = 0; mb < MB; mb++) {
#pragma unroll
          for (int i = 0; i < 16; i++) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            int lz, ly, lx;
            mrow_voxel(wm * MB + mb, row, lz, ly, lx);
            const int gz = z0 + lz, gy = y0 + ly, gx = x0 + lx;
            if (gz < Td && gy < Th && gx < Tw)
              a.kpart[((int64_t)blockIdx.z * mtot + (((int64_t)n * a.Do + gz) * a.Ho + gy) * a.Wo + gx) * a.CoutP + ch] =
                  acc[mb][i];
          }
        }
      }
      return;
    }
  }
  const bool ch_ok = n_active && ch < a.Cout;
  const float bias = (a.bias && ch_ok) ? a.bias[ch] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  // split output: this wave's 32-channel block lies entirely on one side (split is a multiple of 32)
  T* outp = (a.split && n_base >= a.split) ? reinterpret_cast<T*>(a.out2) - a.split : reinterpret_cast<T*>(a.out);
#pragma unroll
  for (int mb = 0; mb < MB; mb++) {
    // an accumulating launch requests the 16 old values of this row block together (one 2-byte load at a time behind
    // the bounds branch doubled the launch: 82 vs 40 us for the 64 -> 128 conv at 32^3)
    float old[16];
    if (a.accumulate) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        int lz, ly, lx;
        mrow_voxel(wm * MB + mb, row, lz, ly, lx);
        int gz = z0 + lz, gy = y0 + ly, gx = x0 + lx;
        const bool ok = ch_ok && gz < Td && gy < Th && gx < Tw;
        int qz = (CONVT && !FLAT) ? 2 * gz + pz : gz, qy = CONVT ? 2 * gy + py : gy, qx = CONVT ? 2 * gx + px : gx;
        const T* p = outp + ((((int64_t)n * a.Do + qz) * a.Ho + qy) * a.Wo + qx) * a.out_pitch + ch;
        old[i] = ST<T>::ld(ok ? p : outp);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++) old[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
      int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      int lz, ly, lx;
      mrow_voxel(wm * MB + mb, row, lz, ly, lx);
      int gz = z0 + lz, gy = y0 + ly, gx = x0 + lx;
      if (ch_ok && gz < Td && gy < Th && gx < Tw) {
        int qz = (CONVT && !FLAT) ? 2 * gz + pz : gz, qy = CONVT ? 2 * gy + py : gy, qx = CONVT ? 2 * gx + px : gx;
        T* p = outp + ((((int64_t)n * a.Do + qz) * a.Ho + qy) * a.Wo + qx) * a.out_pitch + ch;
        float v = acc[mb][i] + bias + old[i];
        ST<T>::st(p, v);
        s1 += v;
        s2 += v * v;
      }
    }
  }
  if (a.stat_partials) {
    // lanes r and r+32 hold the same channel (different rows): fold, then reduce over the WM waves
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    __syncthreads();
    if (h == 0) {  // one slot per wave, summed below in a fixed order: bitwise reproducible statistics
      s_red[((wm * WN + wn) * 32 + r) * 2 + 0] = s1;
      s_red[((wm * WN + wn) * 32 + r) * 2 + 1] = s2;
    }
    __syncthreads();
    if (threadIdx.x < WN * 32) {
      int c = blockIdx.y * (WN * 32) + threadIdx.x;
      if (c < a.CoutP) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < WM; k++) {
          t1 += s_red[(k * WN * 32 + threadIdx.x) * 2 + 0];
          t2 += s_red[(k * WN * 32 + threadIdx.x) * 2 + 1];
        }
        float* q = a.stat_partials + ((int64_t)blockIdx.x * a.CoutP + c) * 2;
        q[0] = t1;
        q[1] = t2;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// conv_ws2: weights-stationary persistent kernel with a DOUBLE-BUFFERED activation tile.
//
// One wave per SIMD leaves nothing to hide the staging work behind, and in the single-buffered predecessor of this
// kernel (round-1 conv_ws_kernel, see git history) the commit (global ->
// InstanceNorm/ReLU transform -> LDS), the bounds logic and the prefetch issue cost as many cycles as the MFMAs.
// Here a pass over item i (= one 4x8x8 tile x one CH-byte channel chunk) reads tile buffer i&1 while the SAME
// instruction stream, in the gaps between its MFMAs, transforms and writes item i+1 into the other buffer and
// re-loads the freed registers with item i+2: one workgroup barrier per pass, staging under the MFMAs.
//  * unpadded rows (pitch == CH) + XOR swizzle of the 16-byte slots keep both buffers and the whole weight
//    panel (27 x 32 x RB bytes) inside 160 KiB; a_swz makes every ds_read_b128 lane group conflict-free
//  * a wave owns the M-blocks y = 2w, 2w+1: the A fragment of box row y' serves (mb, jy) with mb + jy == y', so a
//    (jz, jx) group issues 4 A + 3 B fragment reads for 6 MFMAs (one fragment pair per MFMA: 9 reads)
//  * the NCH passes of a tile are unrolled inside one "tile phase": chunk indices, buffer parities and the
//    accumulators' lifetime are compile-time, tile coordinates advance incrementally once per tile (every
//    instruction outside the MFMA shadow costs ~5 cycles with one wave per SIMD), scale/shift sit in an LDS table
//  * epilogue: accumulators (+bias) -> storage type -> element stores (32 lanes = one voxel row of 32 channels), no
//    LDS staging and no barrier; the cross-wave reduction of the InstanceNorm partials rides on the next pass's
//    barrier
//  * schedule: an interior pass (unchecked copy of the phase) then a border pass (checked copy) per workgroup, both
//    split XCD-aware (see the kernel body)
template <int CH>
__device__ __forceinline__ int a_swz(int row) {
  return CH == 64 ? ((row >> 2) & 3) : ((row >> 3) & 1);
}

// 16 zero bytes: the source of LDS-DMA slots that lie outside the tensor (zero padding)
__device__ __attribute__((aligned(16))) uint32_t g_zero_line[4] = {0u, 0u, 0u, 0u};

// NB: 32-channel output blocks per workgroup.  NB = 2 (64-byte rows, Cout % 64 == 0) stages every tile once for both
// blocks (two workgroups with 32 channels each staged it twice: the 32->64 dgrad at 128^3 was the costliest launch of
// the step) and shares each A fragment between two MFMAs.
// (An LDS-DMA staging variant of the transform-free 32-byte-chunk layers -- global_load_lds_dwordx4 instead of the pf
// registers and ds_write commits -- was built at the end of round 2, measured neutral (536 vs 539 us) and removed in
// round 3: DESIGN.md section 6d.)
// TDP (round 6): tile depth, 4 or 8.  TDP = 8 -- an 8x8x8 tile, taken by the 64-byte-row single-block layers (32 -> 32
// channels at the top level: the four launches of a step that are co-bound by HBM and MFMA, VERDICT r05 #1a) -- stages a
// 10x10x10 box for 512 outputs instead of 6x10x10 for 256: 1.95 instead of 2.34 box voxels per output through L2 ->
// registers -> LDS, the z halo fetched 1.25x instead of 1.5x, 8 A + 3 B fragment reads per 12 MFMAs instead of 4 + 3 per
// 6, one epilogue / barrier / tile step per 108 MFMAs instead of per 54.  A wave then owns four M-blocks (z half zh, row
// mb): MFMA row -> (dz + 4 zh, x) with the same ws_row_to_zx map, so the fragment addresses of z half 1 are those of half
// 0 plus a compile-time offset.  Needs the padded tile-buffer layout (two 1000-row buffers + the panel: 150 KiB).
template <typename T, int CH, int RB, bool XF, int NB, bool BS = false, int TDP = 4>
__global__ __launch_bounds__(256) void conv_ws2_kernel(ConvArgs a) {
  if (a.prio) HDF_LIGHT_PRIO();
  constexpr int TD = TDP, ZH = TDP / 4, TH = 8, TW = 8, BD = TD + 2, BH = TH + 2, BW = TW + 2, BOX = BD * BH * BW;
  static_assert(TDP == 4 || TDP == 8, "tile depth");
  constexpr int ESZ = sizeof(T), EPC = ST<T>::EPC;
  constexpr int CIN = RB / ESZ;
  constexpr int NCH = RB / CH, NFS = CH / 32, NG = 9 * NFS;
  constexpr int CPV = CH / 16, CPV_SHIFT = (CPV == 4) ? 2 : 1;
  constexpr int TOTAL = BOX * CPV, NJ = (TOTAL + 255) / 256;
  // tile-buffer row pitch: padded by 16 bytes (conflict-free, see ws_row_to_zx; fragment addresses = one lane
  // base + compile-time offsets) whenever two padded buffers and the weight panel fit in 160 KiB; else unpadded
  // rows with XOR-swizzled 16-byte slots (a 36-entry per-lane address table)
  constexpr bool SWZ = (2 * BOX * (CH + 16) + 4096 + 512 + 27 * 32 * NB * RB > 160 * 1024);
  constexpr int AP = SWZ ? CH : CH + 16;
  constexpr int ABUF = BOX * AP;
  constexpr int OFF_RED = 2 * ABUF, OFF_XF = OFF_RED + 2048 * NB, OFF_W = OFF_XF + 512;
  constexpr int NC = 32 * NB;  // output channels of the workgroup
  constexpr int CPR = RB / 16, RP256 = 16 / CPR;
  static_assert(CH == 32 || CH == 64, "chunk bytes");
  static_assert(NCH == 1 || NCH % 2 == 0, "buffer parity at tile start must be compile-time");
  static_assert(2 * CIN * 4 <= 512, "scale/shift table");
  static_assert(OFF_W + 27 * NC * RB <= 160 * 1024, "LDS budget");
  static_assert(ZH == 1 || !SWZ, "the 8-deep tile needs the padded layout");
  __shared__ __attribute__((aligned(256))) char lds[OFF_W + 27 * NC * RB];
#ifdef WS_DBG_STAMPS
  const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
#endif
  char* const a_lds = lds;
  float* const s_red = reinterpret_cast<float*>(lds + OFF_RED);  // [2 halves][4 waves][NC][2]
  float* const s_xf = reinterpret_cast<float*>(lds + OFF_XF);
  char* const w_lds = lds + OFF_W;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * NC;
  const int part = tid & (CPV - 1);
  constexpr bool xf = XF;  // input transform x*scale+shift (+relu) on the way into LDS; else a plain copy
  const float relu_lo = (xf && a.in_relu) ? 0.f : -INFINITY;

  // ---- weights -> LDS, once (rows [tap][cout 32][RB bytes], 16-byte slots XOR-swizzled by row).  ALL of a thread's
  // loads are issued before its first LDS store: written as load / store pairs in a loop, hipcc waited for every load
  // in turn (s_waitcnt vmcnt(0) in front of each ds_write), and with all 256 workgroups asking the same L2 lines for
  // the same rows at the same moment one round trip took ~1.6 us -- 27 of them in a row were 45-55 us of a 480 us
  // launch (real-time stamps: 575 us per launch against 520 us from the first to the last instruction after this loop).
  // The batches start at a different 4 KB slice per workgroup so the requests of a round spread over the L2 channels.
  {
    constexpr int WTOT = 27 * NC * CPR, WIT = (WTOT + 255) / 256;
    u32x4 wv[WIT];
    const int rot = blockIdx.x % WIT;
#pragma unroll
    for (int u = 0; u < WIT; u++) {
      int it = u + rot;
      it -= (it >= WIT) ? WIT : 0;
      const int id = min(tid + 256 * it, WTOT - 1);
      const int row = id / CPR, ch = id - row * CPR;
      const int tap = row / NC, rr = row - tap * NC;
      wv[u] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.w) +
                                             ((int64_t)(tap * a.CoutP + n0 + rr) * CIN) * ESZ + ch * 16);
    }
#pragma unroll
    for (int u = 0; u < WIT; u++) {
      int it = u + rot;
      it -= (it >= WIT) ? WIT : 0;
      const int id = tid + 256 * it;
      const int row = id / CPR, ch = id - row * CPR;
      const int sw = ch ^ ((row / RP256) & (CPR - 1));
      if (id < WTOT) *reinterpret_cast<u32x4*>(w_lds + row * RB + sw * 16) = wv[u];
    }
  }
  const int bswz = (r / RP256) & (CPR - 1);

  // Tile schedule: every workgroup first runs a contiguous share of the INTERIOR tiles (halo box inside the volume:
  // no bounds logic, one long run inside the hot copy of the phase), then a share of the
  // BORDER tiles with the checked copy.  Walking all tiles in raster order switched between the copies at every
  // x-row (two border tiles per 16), and each switch re-fetched cold code and drained the staging pipeline.
  const int ntz = (a.Do + TD - 1) / TD, nty = (a.Ho + TH - 1) / TH, ntx = (a.Wo + TW - 1) / TW;
  const bool has_int = ntz >= 3 && nty >= 3 && ntx >= 3;
  const int ipz = ntz - 2, ipy = nty - 2, ipx = ntx - 2;                // interior tile grid per sample
  const int n_int = has_int ? a.N * ipz * ipy * ipx : 0;
  const int per_all = ntz * nty * ntx, per_bor = per_all - (has_int ? ipz * ipy * ipx : 0);
  const int n_bor = a.N * per_bor;
  // XCD-aware split: workgroup b runs on XCD b % 8 (round-robin dispatch), and each XCD has its own 4 MiB L2.  XCD x
  // owns the x-th eighth of the raster-ordered tile list and its G/8 workgroups walk that range interleaved (tile
  // r0 + slot, + G/8, ...): at any moment the XCD works on ~G/8 consecutive tiles, so the halo planes neighbouring
  // tiles share are fetched from HBM once and hit in that L2 (a contiguous chunk per workgroup re-fetched them:
  // FETCH_SIZE 2.8x the input).
  const int G = gridDim.x;
  const int NX = (G % 8 == 0) ? 8 : 1;  // ranges
  const int WPX = G / NX;               // workgroups per range = stride of a workgroup inside its range
  const int xcd = blockIdx.x % NX, slot = blockIdx.x / NX;
  auto split = [&](int total, int& begin, int& cnt) {
    const int r0 = (int)((int64_t)total * xcd / NX), r1 = (int)((int64_t)total * (xcd + 1) / NX);
    begin = r0 + slot;
    cnt = (r1 - r0 > slot) ? (r1 - r0 - slot + WPX - 1) / WPX : 0;
  };
  int int_begin, int_cnt, bor_begin, bor_cnt;
  split(n_int, int_begin, int_cnt);
  split(n_bor, bor_begin, bor_cnt);
  if (int_cnt + bor_cnt == 0) return;
  // mixed-radix digits of the stride WPX over the interior tile grid (x, y, z, n): int_next adds them with carries
  int sdx = 0, sdy = 0, sdz = 0, sdn = 0;
  if (has_int) {
    int t = WPX;
    sdx = t % ipx, t /= ipx;
    sdy = t % ipy, t /= ipy;
    sdz = t % ipz, sdn = t / ipz;
  }

  // ---- per-lane constants
  // fragment read addresses: [jz][y'][jx] -> byte offset of this lane's 16-byte slot inside a tile buffer
  // (swizzled layout: a table; padded layout: abase + a compile-time offset)
  int aaddr[SWZ ? 3 : 1][SWZ ? 4 : 1][SWZ ? 3 : 1];
  int abase;
  {
    int dz, x;
    ws_row_to_zx(r, dz, x);
    abase = ((dz * BH + 2 * wave) * BW + x) * AP + h * 16;
    if constexpr (SWZ) {
#pragma unroll
      for (int jz = 0; jz < 3; jz++)
#pragma unroll
        for (int yp = 0; yp < 4; yp++)
#pragma unroll
          for (int jx = 0; jx < 3; jx++) {
            int row = ((dz + jz) * BH + 2 * wave + yp) * BW + x + jx;
            aaddr[jz][yp][jx] = row * CH + ((h ^ a_swz<CH>(row)) << 4);
          }
    }
  }
  auto a_addr = [&](int jz, int yp, int jx, int fs, int zh = 0) {
    if constexpr (SWZ)
      return aaddr[jz][yp][jx] ^ (fs * 32);
    else
      return abase + (((jz + 4 * zh) * BH + yp) * BW + jx) * AP + fs * 32;
  };
  int bvar[NCH];  // this lane's 16-byte slot inside a weight row, per channel chunk
#pragma unroll
  for (int c = 0; c < NCH; c++) bvar[c] = r * RB + (((c * CPV + h) ^ bswz) << 4);
  // staging slots of this thread: source element offset, packed box coordinates, LDS byte offset
  int boff[NJ], bxyz[NJ], woff_t[SWZ ? NJ : 2];
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    // threads past the end of the last slot redo the box's last voxel (same part): identical bytes to the same
    // address, so neither the load nor the LDS write needs a predicate
    int vox = min(tid + 256 * j, TOTAL - CPV + part) >> CPV_SHIFT;
    int bz = vox / (BH * BW), rem = vox - bz * (BH * BW), by = rem / BW, bx = rem - by * BW;
    boff[j] = ((bz * a.Hi + by) * a.Wi + bx) * (int)a.in_pitch;
    bxyz[j] = (bz << 16) | (by << 8) | bx;
    if constexpr (SWZ) {
      woff_t[j] = vox * CH + ((part ^ a_swz<CH>(vox)) << 4);
    } else {  // slot j = slot 0 + 256/CPV voxels * j: a compile-time offset (the clamped last slot apart)
      if (j == 0) woff_t[0] = vox * AP + part * 16;
      if (j == NJ - 1) woff_t[1] = vox * AP + part * 16;
    }
  }
  auto woff = [&](int j) {
    if constexpr (SWZ)
      return woff_t[j];
    else
      return j == NJ - 1 ? woff_t[1] : woff_t[0] + j * (256 / CPV) * AP;
  };

  // linear index of a tile in raster order (the InstanceNorm partial row it owns)
  auto tile_lin = [&](const WsTile& c) { return ((c.n * ntz + c.z0 / TD) * nty + (c.y0 >> 3)) * ntx + (c.x0 >> 3); };
  // k-th interior tile / k-th border tile, in raster order
  auto int_init = [&](WsTile& c, int k) {
    int t = k;
    c.x0 = (t % ipx + 1) * TW;
    t /= ipx;
    c.y0 = (t % ipy + 1) * TH;
    t /= ipy;
    c.z0 = (t % ipz + 1) * TD;
    c.n = t / ipz;
    c.tile = tile_lin(c);
  };
  auto int_next = [&](WsTile& c) {  // + WPX tiles in raster order of the interior grid
    int xi = (c.x0 >> 3) - 1 + sdx, yi = (c.y0 >> 3) - 1 + sdy, zi = c.z0 / TD - 1 + sdz;
    c.n += sdn;
    if (xi >= ipx) xi -= ipx, yi++;
    if (yi >= ipy) yi -= ipy, zi++;
    if (zi >= ipz) zi -= ipz, c.n++;
    c.x0 = (xi + 1) * TW, c.y0 = (yi + 1) * TH, c.z0 = (zi + 1) * TD;
    c.tile = tile_lin(c);
  };
  auto bor_init = [&](WsTile& c, int k) {
    int tz, ty, tx;
    c.k = k;
    c.n = k / per_bor;
    int rem = k - c.n * per_bor;
    if (!has_int) {
      tx = rem % ntx, ty = (rem / ntx) % nty, tz = rem / (ntx * nty);
    } else {
      const int plane = nty * ntx, ring = plane - ipy * ipx;  // border tiles of a z-plane: all of it / its rim
      if (rem < plane) {
        tz = 0, ty = rem / ntx, tx = rem % ntx;
      } else if (rem - plane < ipz * ring) {
        rem -= plane;
        tz = 1 + rem / ring;
        rem %= ring;
        if (rem < ntx) {
          ty = 0, tx = rem;
        } else if (rem - ntx < 2 * ipy) {
          rem -= ntx;
          ty = 1 + (rem >> 1), tx = (rem & 1) ? ntx - 1 : 0;
        } else {
          ty = nty - 1, tx = rem - ntx - 2 * ipy;
        }
      } else {
        rem -= plane + ipz * ring;
        tz = ntz - 1, ty = rem / ntx, tx = rem % ntx;
      }
    }
    c.z0 = tz * TD, c.y0 = ty * TH, c.x0 = tx * TW;
    c.tile = tile_lin(c);
  };
  // border tiles are re-decoded from their list index (a few integer divisions per tile of the checked path)
  auto bor_next = [&](WsTile& c) { bor_init(c, c.k + WPX); };
  auto tile_interior = [&](const WsTile& c) {
    return c.z0 >= 1 && c.y0 >= 1 && c.x0 >= 1 && c.z0 + TD + 1 <= a.Di && c.y0 + TH + 1 <= a.Hi &&
           c.x0 + TW + 1 <= a.Wi;
  };
  const T* const src_safe = reinterpret_cast<const T*>(a.in) + part * EPC;
  // first box voxel of the tile (channel chunk 0, this thread's part); may lie outside the tensor for border tiles
  auto tile_org = [&](const WsTile& c) -> const T* {
    return src_safe + ((((int64_t)c.n * a.Di + (c.z0 - 1)) * a.Hi + (c.y0 - 1)) * a.Wi + (c.x0 - 1)) * a.in_pitch;
  };

  u32x4 pf[NJ];
  // box voxel of slot j inside the volume?  (border tiles only)
  auto slot_ok = [&](int j, const WsTile& c) {
    int iz = c.z0 - 1 + (bxyz[j] >> 16), iy = c.y0 - 1 + ((bxyz[j] >> 8) & 255), ix = c.x0 - 1 + (bxyz[j] & 255);
    return ((unsigned)iz < (unsigned)a.Di) & ((unsigned)iy < (unsigned)a.Hi) & ((unsigned)ix < (unsigned)a.Wi);
  };
  // loads are UNCONDITIONAL (clamped address + select): one load per slot on every path, so the compiler's
  // vmcnt bookkeeping stays exact and nothing ever branches around a load.
  // FAST: the tile is a valid interior tile (no test at all); else `inter` (uniform) short-cuts the test.
  auto load_one = [&](auto fast_tag, int j, int chunk, const WsTile& c, bool valid, bool inter, const T* org) {
    if constexpr (decltype(fast_tag)::value) {
      pf[j] = *reinterpret_cast<const u32x4*>(org + boff[j] + chunk * (CH / ESZ));
    } else {
      const bool ok = inter | (valid & slot_ok(j, c));  // bitwise: no short-circuit branches
      const T* p = ok ? org + boff[j] + chunk * (CH / ESZ) : src_safe;
      pf[j] = *reinterpret_cast<const u32x4*>(p);
    }
  };
  float sc[EPC], sh[EPC];
  auto read_xf = [&](int chunk) {
    const int cb = chunk * (CH / ESZ) + part * EPC;
#pragma unroll
    for (int e = 0; e < EPC; e += 4) {
      f32x4 u = *reinterpret_cast<const f32x4*>(s_xf + cb + e);
      f32x4 v = *reinterpret_cast<const f32x4*>(s_xf + CIN + cb + e);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        sc[e + k] = u[k];
        sh[e + k] = v[k];
      }
    }
  };
  // straight-line on purpose (no branch may split an MFMA group); relu_lo is -inf when there is no ReLU
  auto commit_one = [&](auto fast_tag, int j, const WsTile& c, bool inter, char* dst) {
    u32x4 v = pf[j];
    if constexpr (XF) {
      float f[EPC];
      ST<T>::unpack(v, f);
#pragma unroll
      for (int e = 0; e < EPC; e++) f[e] = fmaxf(f[e] * sc[e] + sh[e], relu_lo);
      v = ST<T>::pack(f);
    }
    if constexpr (!decltype(fast_tag)::value) {
      const bool ok = inter | slot_ok(j, c);
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = ok ? v[k] : 0u;
    }
    *reinterpret_cast<u32x4*>(dst + woff(j)) = v;
  };
  int tbl_n = -1;
  auto refresh_xf = [&](int n) {  // uniform; callers guarantee nobody still reads the old table
    if (tid < CIN) {
      s_xf[tid] = xf ? a.in_scale[(int64_t)n * CIN + tid] : 1.f;
      s_xf[CIN + tid] = xf ? a.in_shift[(int64_t)n * CIN + tid] : 0.f;
    }
    tbl_n = n;
    __syncthreads();
  };

  // T0: tile under the MFMAs, T1 / T2: the next two (T2 only feeds the loads when a tile is a single pass)
  WsTile T0, T1, T2;
  bool v1 = false, v2 = false, i0 = false, i1 = false, i2 = false;
  const T *org0 = src_safe, *org1 = src_safe, *org2 = src_safe;
  int left = 0;  // tiles of the current pass after T0
  // start a pass at its k-th tile: T0 -> buffer 0 (not overlapped), the next item -> registers
  auto begin_pass = [&](auto border_tag, int k, int count) __attribute__((always_inline)) {
    constexpr bool BORDER = decltype(border_tag)::value;
    auto next = [&](WsTile& c) {
      if constexpr (BORDER)
        bor_next(c);
      else
        int_next(c);
    };
    if constexpr (BORDER)
      bor_init(T0, k);
    else
      int_init(T0, k);
    left = count - 1;
    T1 = T0;
    next(T1);
    T2 = T1;
    next(T2);
    v1 = left >= 1, v2 = left >= 2;
    i0 = BORDER ? tile_interior(T0) : true;
    i1 = v1 && (BORDER ? tile_interior(T1) : true);
    i2 = v2 && (BORDER ? tile_interior(T2) : true);
    org0 = tile_org(T0);
    org1 = v1 ? tile_org(T1) : src_safe;
    org2 = v2 ? tile_org(T2) : src_safe;
    __syncthreads();  // the previous pass is done with both tile buffers
#pragma unroll
    for (int j = 0; j < NJ; j++) load_one(std::false_type{}, j, 0, T0, true, i0, org0);
    if constexpr (XF) {
      refresh_xf(T0.n);
      read_xf(0);
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) commit_one(std::false_type{}, j, T0, i0, a_lds);
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      if (NCH > 1)
        load_one(std::false_type{}, j, 1, T0, true, i0, org0);
      else
        load_one(std::false_type{}, j, 0, T1, v1, i1, org1);
    }
    // (round 4) drained once per pass: hipcc derives the N of every `s_waitcnt vmcnt(N)` in the tile loop from the
    // predecessor with the FEWEST younger operations -- entered with these loads pending and no store behind them, the
    // first commit of every tile waits for the acknowledgements of the previous tile's stores (see `epilogue`)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    WS_BARRIER();
  };

  const bool plain = !a.accumulate && n0 + NC <= a.Cout;  // uniform: see `epilogue`
  const int ch = n0 + r;                 // channel of output block 0 (block nb: + 32 nb)
  bool ch_ok[NB];
  float bias[NB];
#pragma unroll
  for (int nb = 0; nb < NB; nb++) {
    ch_ok[nb] = ch + 32 * nb < a.Cout;
    bias[nb] = (a.bias && ch_ok[nb]) ? a.bias[ch + 32 * nb] : 0.f;
  }
  T* outp[NB];  // split output: a 32-channel block lies entirely on one side (split is a multiple of 32)
#pragma unroll
  for (int nb = 0; nb < NB; nb++)
    outp[nb] = (a.split && n0 + 32 * nb >= a.split) ? reinterpret_cast<T*>(a.out2) - a.split : reinterpret_cast<T*>(a.out);
  // accumulator register i of this lane: tile voxel (dz, 2w + mb, x) with x = (i & 3) + 4 * ((i >> 2) & 1) and
  // dz = {0,1,3,2}[i >> 2] (h == 0) or {1,0,2,3}[i >> 2] (h == 1)   (ws_row_to_zx of row (i&3) + 8*(i>>2) + 4h)
  int edz[4], eplane[4];
#pragma unroll
  for (int q4 = 0; q4 < 4; q4++) {
    int dz, x;
    ws_row_to_zx(8 * q4 + 4 * h, dz, x);
    edz[q4] = dz;
    eplane[q4] = dz * a.Ho * a.Wo * (int)a.out_pitch;
  }
  auto eoff = [&](int i) { return eplane[i >> 2] + ((i & 3) + 4 * ((i >> 2) & 1)) * (int)a.out_pitch; };
  const int ezh = 4 * a.Ho * a.Wo * (int)a.out_pitch;   // z half 1 of an 8-deep tile: four planes further
  // InstanceNorm partial rows are per WORKGROUP, not per tile: sample n owns WS_STAT_ROWS rows, row
  // pass * 256 + blockIdx.x (+ k gridDim.x for the slots no workgroup has) holds this workgroup's sums over its
  // tiles of n in that pass.  Every lane keeps running sums of ITS channel over the voxels it has produced (lr1 / lr2:
  // two registers per output block, two adds per tile); lanes and waves are combined through s_red only when the sample
  // changes or the pass ends (stats_to_row, a handful of times per launch).  Round 2 sent every tile's per-wave sums
  // through LDS and had tid < NC collect them behind the next barrier: a shuffle pair, two LDS writes, eight LDS reads
  // and two branches per tile inside the hot loop.  The rows are zeroed here first, so in_finalize reads 512 rows
  // instead of one per tile.
  float lr1[NB], lr2[NB];
#pragma unroll
  for (int nb = 0; nb < NB; nb++) lr1[nb] = lr2[nb] = 0.f;
  int racc_n = -1, cur_pass = 0;
  auto stat_row = [&](int n, int pass, int b) {
    return a.stat_partials + (((int64_t)n * WS_STAT_ROWS + pass * 256 + b) * a.CoutP + n0 + tid) * 2;
  };
  if (a.stat_partials && tid < NC) {
    for (int n = 0; n < a.N; n++)
      for (int pass = 0; pass < 2; pass++)
        for (int b = blockIdx.x; b < 256; b += gridDim.x) {
          float* q = stat_row(n, pass, b);
          q[0] = 0.f;
          q[1] = 0.f;
        }
  }
  // uniform (every wave takes the same path): lanes -> waves -> this workgroup's row of sample racc_n, fixed order
  auto stats_to_row = [&]() __attribute__((always_inline)) {
    if (a.stat_partials && racc_n >= 0) {
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        const float u1 = lr1[nb] + __shfl_xor(lr1[nb], 32, 64), u2 = lr2[nb] + __shfl_xor(lr2[nb], 32, 64);
        if (h == 0) {
          s_red[(wave * NC + 32 * nb + r) * 2 + 0] = u1;
          s_red[(wave * NC + 32 * nb + r) * 2 + 1] = u2;
        }
      }
      __syncthreads();
      if (tid < NC) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          t1 += s_red[(k * NC + tid) * 2 + 0];
          t2 += s_red[(k * NC + tid) * 2 + 1];
        }
        float* q = stat_row(racc_n, cur_pass, blockIdx.x);
        q[0] += t1;
        q[1] += t2;
      }
      __syncthreads();
    }
#pragma unroll
    for (int nb = 0; nb < NB; nb++) lr1[nb] = lr2[nb] = 0.f;
    racc_n = -1;
  };
  // the tile about to be summed belongs to sample n (uniform)
  // BS (backward statistics, ConvArgs::bs_*): this lane's channel constants of the sample being summed
  float bsc = 0.f, bsh = 0.f, bmu = 0.f, brs = 0.f;
  auto stats_sample = [&](int n) __attribute__((always_inline)) {
    if (n != racc_n) {
      if (racc_n >= 0) stats_to_row();
      racc_n = n;
      if constexpr (BS) {
        const int64_t o = (int64_t)n * a.Cout + min(n0 + r, a.Cout - 1);
        bsc = a.bs_scale[o], bsh = a.bs_shift[o], bmu = a.bs_mean[o], brs = a.bs_rstd[o];
      }
    }
  };

#ifdef WS_DBG_STAMPS
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long treal0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz, the same counter on every CU
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
  const unsigned long long tcyc0 = tlast;
#define WS2_STAMP(k)                                       \
  {                                                        \
    __builtin_amdgcn_sched_barrier(0);                     \
    unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
    __builtin_amdgcn_s_waitcnt(0xC07F);                    \
    __builtin_amdgcn_sched_barrier(0);                     \
    tacc[k] += t_ - tlast;                                 \
    tlast = t_;                                            \
  }
#else
#define WS2_STAMP(k)
#endif

  // this lane's first output element of a tile: channel n0 + r, rows y0 + 2w (+ mb), voxel (z0, ., x0)
  auto out_base = [&](const WsTile& t, int nb) -> T* {
    return outp[nb] + ((((int64_t)t.n * a.Do + t.z0) * a.Ho + t.y0 + 2 * wave) * a.Wo + t.x0) * a.out_pitch + ch + 32 * nb;
  };
  // PLAIN (compile-time, round 4): the launch stores whole 32-channel blocks without accumulating (`plain`, uniform) and
  // the tile is whole (every interior tile is): the epilogue is then branch-free, a FIXED number of stores.  vmcnt counts
  // stores too and retires in order; with a conditional store path in the loop hipcc assumes that no store is younger than
  // the loads the next tile's first commit waits for, emits vmcnt(4) and thereby waits for the write acknowledgements of
  // this tile's 32 NB stores at every tile top (the "tile top" share of the round-3 stamps).
  auto epilogue = [&](f32x16 (&acc)[2 * NB * ZH], const WsTile& ET, auto plain_tag) __attribute__((always_inline)) {
    constexpr bool PLAIN = decltype(plain_tag)::value;
    const int z0 = ET.z0, y0 = ET.y0, x0 = ET.x0;
    const bool full = PLAIN || (z0 + TD <= a.Do && y0 + TH <= a.Ho && x0 + TW <= a.Wo);
    stats_sample(ET.n);
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      float s1 = 0.f, s2 = 0.f;
      T* const obase = out_base(ET, nb);
      if (PLAIN || (full && ch_ok[nb] && !a.accumulate)) {
        if constexpr (BS) {
          // the y values of the tile's 32 voxels are requested first, the outputs stored under their latency; the sums
          // use the STORED (storage-rounded) gradient, as in_bwd_reduce would read it back
#pragma unroll
          for (int zh = 0; zh < ZH; zh++) {
          const T* const ybase = reinterpret_cast<const T*>(a.bs_y) +
                                 ((((int64_t)ET.n * a.Do + z0 + 4 * zh) * a.Ho + y0 + 2 * wave) * a.Wo + x0) * a.bs_y_pitch + ch;
          float yv[2][16];
#pragma unroll
          for (int mb = 0; mb < 2; mb++)
#pragma unroll
            for (int i = 0; i < 16; i++)
              yv[mb][i] = ST<T>::ld(ybase + ((int64_t)(edz[i >> 2] * a.Ho + mb) * a.Wo + (i & 3) + 4 * ((i >> 2) & 1)) *
                                                a.bs_y_pitch);
#pragma unroll
          for (int mb = 0; mb < 2; mb++) {
            T* const orow = obase + (int64_t)mb * a.Wo * a.out_pitch + zh * ezh;
#pragma unroll
            for (int i = 0; i < 16; i++) {
              const float v = acc[(nb * ZH + zh) * 2 + mb][i] + bias[nb];
              ST<T>::st(orow + eoff(i), v);
              T tmp;
              ST<T>::st(&tmp, v);
              const float g = (yv[mb][i] * bsc + bsh > 0.f) ? ST<T>::ld(&tmp) : 0.f;
              s1 += g;
              s2 += g * ((yv[mb][i] - bmu) * brs);
            }
          }
          }
        } else {
#pragma unroll
        for (int zh = 0; zh < ZH; zh++)
#pragma unroll
        for (int mb = 0; mb < 2; mb++) {
          T* const orow = obase + (int64_t)mb * a.Wo * a.out_pitch + zh * ezh;
#pragma unroll
          for (int i = 0; i < 16; i++) {
            const float v = acc[(nb * ZH + zh) * 2 + mb][i] + bias[nb];
            ST<T>::st(orow + eoff(i), v);
            s1 += v;
            s2 += v * v;
          }
        }
        }
      } else if (full && ch_ok[nb]) {
        // accumulating launch (the UpConv chain's data gradients add into the skip gradient), whole tile: the 16 old
        // values of a row are requested together, then added and stored -- the generic path below asks for one
        // 2-byte value at a time behind a branch (up3's data gradient at 64^3: 252 us against 76 us for the same conv
        // without accumulation, on the critical path of the backward)
#pragma unroll
        for (int zh = 0; zh < ZH; zh++)
#pragma unroll
        for (int mb = 0; mb < 2; mb++) {
          T* const orow = obase + (int64_t)mb * a.Wo * a.out_pitch + zh * ezh;
          float old[16];
#pragma unroll
          for (int i = 0; i < 16; i++) old[i] = ST<T>::ld(orow + eoff(i));
#pragma unroll
          for (int i = 0; i < 16; i++) {
            const float v = acc[(nb * ZH + zh) * 2 + mb][i] + bias[nb];
            ST<T>::st(orow + eoff(i), v + old[i]);
            s1 += v;
            s2 += v * v;
          }
        }
      } else {
#pragma unroll
        for (int zh = 0; zh < ZH; zh++)
#pragma unroll
        for (int mb = 0; mb < 2; mb++) {
          const int gy = y0 + 2 * wave + mb;
          T* const orow = obase + (int64_t)mb * a.Wo * a.out_pitch + zh * ezh;
#pragma unroll
          for (int i = 0; i < 16; i++) {
            const int gz = z0 + edz[i >> 2] + 4 * zh, gx = x0 + (i & 3) + 4 * ((i >> 2) & 1);
            const float v = acc[(nb * ZH + zh) * 2 + mb][i] + bias[nb];
            const bool ok = gz < a.Do && gy < a.Ho && gx < a.Wo;
            if (ok && ch_ok[nb]) {
              float o = v;
              if (a.accumulate) o += ST<T>::ld(orow + eoff(i));
              ST<T>::st(orow + eoff(i), o);
            }
            const float mk = ok ? 1.f : 0.f;
            s1 += mk * v;
            s2 += mk * v * v;
          }
        }
      }
      if (PLAIN || ch_ok[nb]) {
        lr1[nb] += s1;
        lr2[nb] += s2;
      }
    }
  };

  // One tile: NCH passes.  Pass c runs the MFMAs of chunk c out of buffer (PAR0 + c) & 1 and, in their shadow,
  // commits item c+1 (chunk (c+1) % NCH of T0 or T1) into the other buffer and loads item c+2 into the freed pf.
  // (Rounds 1-2 deferred the epilogue of single-pass tiles into the MFMA gaps of the next tile, with two accumulator
  // register sets; once the 64-byte-row layers became two-pass that served only the 16->32 first layer, where it was
  // worth 3 % of the launch and nothing in the step, at 30 more registers: removed in round 3.)
  f32x16 acc[2 * NB * ZH];  // [(nb * ZH + zh) * 2 + mb]
  auto tile_phase = [&](auto par_tag, auto fast_tag, auto plain_tag) __attribute__((always_inline)) {
    constexpr int PAR0 = decltype(par_tag)::value;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      constexpr int dummy = 0;
      (void)dummy;
      const int PAR = (PAR0 + c) & 1;
      char* const a_rd = a_lds + PAR * ABUF;
      char* const a_wr = a_lds + (1 - PAR) * ABUF;
      // commit target: item c+1; load target: item c+2
      const bool c_next = (c + 1 >= NCH);                  // commit goes to T1
      const int c_chunk = (c + 1) % NCH;
      const int l_tile = (c + 2) / NCH;                    // 0: T0, 1: T1, 2: T2
      const int l_chunk = (c + 2) % NCH;
      const WsTile& CT = c_next ? T1 : T0;
      const bool c_int = c_next ? (i1 | !v1) : i0;         // an invalid next tile is committed as garbage, unchecked
      const WsTile& LT = l_tile == 0 ? T0 : (l_tile == 1 ? T1 : T2);
      const bool l_val = l_tile == 0 ? true : (l_tile == 1 ? v1 : v2);
      const bool l_int = l_tile == 0 ? i0 : (l_tile == 1 ? i1 : i2);
      const T* const l_org = l_tile == 0 ? org0 : (l_tile == 1 ? org1 : org2);
      WS2_STAMP(3)
      if constexpr (XF) {
        if (CT.n != tbl_n && (!c_next || v1)) refresh_xf(CT.n);
        read_xf(c_chunk);
      }
      WS2_STAMP(0)
      u32x4 af[2][4 * ZH], bf[2][3 * NB];
      // RB == 32: the 27 weight fragments of the single pass are the same LDS words for every tile, and hipcc hoists
      // their reads out of the tile loop into 108 AGPRs (384 registers in all: nothing of another stream fits beside
      // the workgroup -- the branch stream's first token kernels waited 150 us for the first encoder conv).  The
      // laundered pointer keeps the reads in the loop, as in the wider variants.
      const char* w_rd = w_lds;
      if constexpr (RB == 32) asm volatile("" : "+v"(w_rd));
      auto read_group = [&](int g, u32x4 (&A)[4 * ZH], u32x4 (&B)[3 * NB]) {
        const int t = g / NFS, fs = g % NFS, jz = t / 3, jx = t % 3;
        auto rdA = [&](int yp) {
#pragma unroll
          for (int zh = 0; zh < ZH; zh++) A[zh * 4 + yp] = *reinterpret_cast<const u32x4*>(a_rd + a_addr(jz, yp, jx, fs, zh));
        };
        auto rdB = [&](int jy) {
#pragma unroll
          for (int nb = 0; nb < NB; nb++)
            B[jy * NB + nb] = *reinterpret_cast<const u32x4*>(w_rd + ((jz * 9 + jy * 3 + jx) * NC + 32 * nb) * RB +
                                                              (bvar[c] ^ (fs * 32)));
        };
        // in order of first use (LDS returns in order, the waits are counted)
        rdB(0);
        rdA(0);
        rdA(1);
        rdB(1);
        rdA(2);
        rdB(2);
        rdA(3);
      };
      read_group(0, af[0], bf[0]);
#pragma unroll
      for (int g = 0; g < NG; g++) {
#ifdef WS2_DBG_NOREADS
        if (g + 1 < 2) read_group(g + 1, af[(g + 1) & 1], bf[(g + 1) & 1]);
#else
        if (g + 1 < NG) read_group(g + 1, af[(g + 1) & 1], bf[(g + 1) & 1]);
#endif
        __builtin_amdgcn_sched_barrier(0);  // look-ahead reads stay ABOVE this group's MFMAs
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          if ((j * NG) / NJ == g) {
#ifndef WS2_DBG_NOSTAGE  // energy/cycle attribution experiments: drop the staging or the fragment reads
            commit_one(fast_tag, j, CT, c_int, a_wr);
            load_one(fast_tag, j, l_chunk, LT, l_val, l_int, l_org);
#endif
          }
        }
        u32x4(&A)[4 * ZH] = af[g & 1];
        u32x4(&B)[3 * NB] = bf[g & 1];
        if (c == 0 && g == 0) {  // first MFMAs of the tile take a zero C operand: no accumulator clearing
#pragma unroll
          for (int q2 = 0; q2 < 2 * NB * ZH; q2++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[q2][i] = 0.f;
        }
#pragma unroll
        for (int jy = 0; jy < 3; jy++)
#pragma unroll
          for (int nb = 0; nb < NB; nb++)
#pragma unroll
            for (int zh = 0; zh < ZH; zh++) {
              Mma<T>::run(A[zh * 4 + jy], B[jy * NB + nb], acc[(nb * ZH + zh) * 2 + 0]);      // mb 0: box row y' = jy
              Mma<T>::run(A[zh * 4 + jy + 1], B[jy * NB + nb], acc[(nb * ZH + zh) * 2 + 1]);  // mb 1: box row y' = jy + 1
            }
        __builtin_amdgcn_sched_barrier(0);
      }
      WS2_STAMP(1)
      WS_BARRIER();  // buffer PAR fully read, buffer 1-PAR fully written
      WS2_STAMP(2)
    }
    epilogue(acc, T0, plain_tag);
    WS2_STAMP(4)
  };

  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using Yes = std::true_type;
  using No = std::false_type;
  bool more = true;
  auto step = [&](auto border_tag) __attribute__((always_inline)) {
    constexpr bool BORDER = decltype(border_tag)::value;
    more = v1;
    if (!more) return;
    left--;
    T0 = T1;
    T1 = T2;
    i0 = i1;
    i1 = i2;
    v1 = v2;
    org0 = org1;
    org1 = org2;
    if constexpr (BORDER)
      bor_next(T2);
    else
      int_next(T2);
    v2 = left >= 2;
    i2 = v2 && (BORDER ? tile_interior(T2) : true);
    org2 = v2 ? tile_org(T2) : src_safe;
  };

  // ---- pass A: interior tiles: the unchecked copy of the phase (single-pass tiles alternate the buffer parity)
  if (int_cnt > 0) {
    begin_pass(No{}, int_begin, int_cnt);
    more = true;
    int par = 0;
    // (two loops, not one loop with a branch: the two epilogues must not meet at one loop head, see `epilogue`)
    if (plain) {
      while (more) {
        if (NCH > 1 || par == 0)
          tile_phase(P0{}, Yes{}, Yes{});
        else
          tile_phase(P1{}, Yes{}, Yes{});
        if (NCH == 1) par ^= 1;
        step(No{});
      }
    } else {
      while (more) {
        if (NCH > 1 || par == 0)
          tile_phase(P0{}, Yes{}, No{});
        else
          tile_phase(P1{}, Yes{}, No{});
        if (NCH == 1) par ^= 1;
        step(No{});
      }
    }
  }
  // ---- pass B: border tiles (checked copy)
  if (bor_cnt > 0) {
    begin_pass(Yes{}, bor_begin, bor_cnt);
    stats_to_row();
    cur_pass = 1;
    more = true;
    int par = 0;
    while (more) {
      if (NCH > 1 || par == 0)
        tile_phase(P0{}, No{}, No{});
      else
        tile_phase(P1{}, No{}, No{});
      if (NCH == 1) par ^= 1;
      step(Yes{});
    }
  }
  stats_to_row();
#ifdef WS_DBG_STAMPS
  if (tid == 0 && blockIdx.y == 0) {
    for (int k = 0; k < 5; k++) a.stat_partials[(int64_t)blockIdx.x * 8 + k] = (float)tacc[k];
    // slots 5, 6: low words of the 100 MHz real-time counter at entry / exit (bit patterns); slot 7: shader cycles
    a.stat_partials[(int64_t)blockIdx.x * 8 + 5] = __uint_as_float((uint32_t)treal0);
    a.stat_partials[(int64_t)blockIdx.x * 8 + 6] = __uint_as_float((uint32_t)__builtin_amdgcn_s_memrealtime());
    a.stat_partials[(int64_t)blockIdx.x * 8 + 7] = (float)(__builtin_amdgcn_s_memtime() - tcyc0);
    a.stat_partials[2048 + blockIdx.x] = __uint_as_float((uint32_t)t_entry);  // kernel entry of this workgroup
  }
#endif
}

// ConvTranspose3d(k3,s2,p1,op1) forward with ALL 8 output-parity classes in one workgroup (Cin*sizeof(T) <= 128 B):
// the (TD+1)x(TH+1)x(TW+1) input box is staged ONCE with full-Cin rows, then each class runs its 1..8 taps and
// writes its 2x-strided outputs through an LDS staging tile as whole 16-byte chunks.  (The per-class launch of
// conv_igemm_kernel<CONVT> restaged the same box 8 times for ~3 taps of work each.)
// NFS = 32-byte fragment steps per voxel row (Cin*sizeof(T)/32): with the class and tap loops unrolled at compile time
// a class is straight-line code.  Written as run-time loops, hipcc carried the accumulators in VGPRs and bracketed
// EVERY pair of MFMAs with 64 v_accvgpr moves (SQ_INSTS_VALU 88 M against 57 M MFMA-busy cycles per launch: the kernel
// was VALU-bound at 280 TF); an `asm("" : "+a"(acc))` pin does not survive the dynamic-trip-count loop nest.
// (round 6) The weight fragments come straight from L2, one per pair of MFMAs, and hipcc kept ONE of them in flight
// (`s_waitcnt vmcnt(1)` in front of every pair: a 64-cycle pair waited out a 500+ cycle load -- 0.086 of the MFMA roof for
// upconv_2).  The 27 x NFS fragment steps of a workgroup are one compile-time sequence (class, tap, step): a register ring
// keeps CT_RING of them in flight across tap and class boundaries, as conv_igemm_kernel does for its chunks.
constexpr int CT_RING = 8;
// tap (in the [27] panel) of the k-th (class, tap) pair in the order the classes walk them, and the first pair of a class
constexpr int ct_class_base(int cls) {
  int n = 0;
  for (int c = 0; c < cls; c++) n += (((c >> 2) & 1) ? 2 : 1) * (((c >> 1) & 1) ? 2 : 1) * ((c & 1) ? 2 : 1);
  return n;
}
constexpr int ct_pair_tap(int k) {
  int idx = 0;
  for (int cls = 0; cls < 8; cls++) {
    const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
    for (int jz = 0; jz < (pz ? 2 : 1); jz++)
      for (int jy = 0; jy < (py ? 2 : 1); jy++)
        for (int jx = 0; jx < (px ? 2 : 1); jx++) {
          if (idx == k) return (((pz ? 2 * jz : 1) * 3 + (py ? 2 * jy : 1)) * 3 + (px ? 2 * jx : 1));
          idx++;
        }
  }
  return 0;
}
// FLAT (round 6): ConvTranspose2d(k3, s2, p1, op1) of the 2-D model on depth-1 tensors -- the four (y, x) parity classes, the
// 9 taps of the panel's centre depth plane, a one-plane box; until then the 2-D decoder ran one launch slice per class
// through conv_igemm_kernel's unpipelined fallback loop.
template <typename T, int TD, int TH, int TW, int MB, int NFS, bool FLAT = false>
__global__ __launch_bounds__(256) void convt_fused_kernel(ConvArgs a) {
  static_assert(4 * MB * 32 == TD * TH * TW, "tile/wave decomposition");
  static_assert(!FLAT || TD == 1, "flat tiles are one voxel deep");
  constexpr int BD = FLAT ? 1 : TD + 1, BH = TH + 1, BW = TW + 1, BOX = BD * BH * BW;
  constexpr int ESZ = sizeof(T), EPC = ST<T>::EPC;
  constexpr int ROWB = NFS > 8 ? 512 : (NFS > 4 ? 256 : 128);  // row payload the box is laid out for (full Cin: up to 128 / 256 / 512 B)
  constexpr int LP = ROWB + 16;              // box row pitch
  static_assert(BOX * LP <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char lds[BOX * LP];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ntz = (a.Di + TD - 1) / TD, nty = (a.Hi + TH - 1) / TH, ntx = (a.Wi + TW - 1) / TW;
  int t = blockIdx.x;
  const int tx = t % ntx;
  t /= ntx;
  const int ty = t % nty;
  t /= nty;
  const int tz = t % ntz;
  const int n = t / ntz;
  const int z0 = tz * TD, y0 = ty * TH, x0 = tx * TW;
  const int n0 = blockIdx.y * 32;
  const int RB = a.Cin * ESZ;                // 32, 64, 128 or 256 bytes of channels per voxel (= 32 * NFS)

  stage_box<T, BD, BH, BW, ROWB, LP>(lds, reinterpret_cast<const T*>(a.in), a.in_pitch, a.Cin, n, a.Di, a.Hi, a.Wi, z0,
                                    y0, x0, 0, RB, a.in_scale, a.in_shift, a.in_relu);
  __syncthreads();

  int rowbase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; mb++) {
    int lin = (wave * MB + mb) * 32 + r;
    int lz = lin / (TH * TW), ly = (lin / TW) % TH, lx = lin % TW;
    rowbase[mb] = ((lz * BH + ly) * BW + lx) * LP + h * 16;
  }
  const int ch = n0 + r;
  const bool ch_ok = ch < a.Cout;
  const float bias = (a.bias && ch_ok) ? a.bias[ch] : 0.f;
  const char* wrow = reinterpret_cast<const char*>(a.w) + ((int64_t)(n0 + r) * a.Cin) * ESZ + h * 16;
  const int64_t wtap_stride = (int64_t)a.CoutP * a.Cin * ESZ;
  // output voxel (2z, 2y, 2x) of every accumulator row of this lane, as an element offset inside the sample (-1: the
  // tile voxel lies outside the volume); a class adds its parity offset
  T* obase = reinterpret_cast<T*>(a.out) + (int64_t)n * a.Do * a.Ho * a.Wo * a.out_pitch + ch;
  int voff[MB][16];
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int lin = (wave * MB + mb) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      const int gz = z0 + lin / (TH * TW), gy = y0 + (lin / TW) % TH, gx = x0 + lin % TW;
      voff[mb][i] = (gz < a.Di && gy < a.Hi && gx < a.Wi) ? (((FLAT ? gz : 2 * gz) * a.Ho + 2 * gy) * a.Wo + 2 * gx) * (int)a.out_pitch : -1;
    }

  constexpr int NSTEP = (FLAT ? 9 : 27) * NFS;   // (flat: classes 0..3, the first 9 pairs of the sequence, all on depth tap 1)
  auto b_load = [&](int g) -> u32x4 {   // g: compile-time after unrolling
    return *reinterpret_cast<const u32x4*>(wrow + ct_pair_tap(g / NFS) * wtap_stride + (g % NFS) * 32);
  };
  u32x4 bq[CT_RING];
#pragma unroll
  for (int g = 0; g < CT_RING; g++) bq[g] = b_load(g);

  auto do_class = [&](auto cls_tag) __attribute__((always_inline)) {
    constexpr int cls = decltype(cls_tag)::value;
    constexpr int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
    constexpr int ntapz = pz ? 2 : 1, ntapy = py ? 2 : 1, ntapx = px ? 2 : 1;
    constexpr int g0 = ct_class_base(cls) * NFS;   // first fragment step of this class
    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[mb][i] = 0.f;
#pragma unroll
    for (int jz = 0; jz < ntapz; jz++) {
      constexpr int dummy = 0;
      (void)dummy;
      const int offz = pz ? 1 - jz : 0, wz = pz ? 2 * jz : 1;
#pragma unroll
      for (int jy = 0; jy < ntapy; jy++) {
        const int offy = py ? 1 - jy : 0, wy = py ? 2 * jy : 1;
#pragma unroll
        for (int jx = 0; jx < ntapx; jx++) {
          const int offx = px ? 1 - jx : 0, wx = px ? 2 * jx : 1;
          const int tapoff = ((offz * BH + offy) * BW + offx) * LP;
          (void)wz, (void)wy, (void)wx;   // (the tap's panel index is ct_pair_tap of its position in the sequence)
#pragma unroll
          for (int fs = 0; fs < NFS; fs++) {
            const int g = g0 + ((jz * ntapy + jy) * ntapx + jx) * NFS + fs;
            const u32x4 bfrag = bq[g % CT_RING];
            u32x4 afrag[MB];
#pragma unroll
            for (int mb = 0; mb < MB; mb++)
              afrag[mb] = *reinterpret_cast<const u32x4*>(lds + rowbase[mb] + tapoff + fs * 32);
#pragma unroll
            for (int mb = 0; mb < MB; mb++) Mma<T>::run(afrag[mb], bfrag, acc[mb]);
            if (g + CT_RING < NSTEP) bq[g % CT_RING] = b_load(g + CT_RING);
            __builtin_amdgcn_sched_barrier(0);   // (without it hipcc sinks the refill down to its use again)
          }
        }
      }
    }
    // epilogue straight from the accumulators (lane = channel r, 16 tile voxels per M-block): 2-byte stores, 32 lanes =
    // one 64-byte voxel row.  No LDS staging, no barrier: a wave's stores of class c run under its MFMAs of class c+1
    // and under the other waves' work (the staged form cost two workgroup barriers and an LDS round trip per class)
    const int coff = ((pz * a.Ho + py) * a.Wo + px) * (int)a.out_pitch;
#pragma unroll
    for (int mb = 0; mb < MB; mb++)
#pragma unroll
      for (int i = 0; i < 16; i++)
        if (ch_ok && voff[mb][i] >= 0) ST<T>::st(obase + voff[mb][i] + coff, acc[mb][i] + bias);
  };
  do_class(std::integral_constant<int, 0>{});
  do_class(std::integral_constant<int, 1>{});
  do_class(std::integral_constant<int, 2>{});
  do_class(std::integral_constant<int, 3>{});
  if constexpr (!FLAT) {
    do_class(std::integral_constant<int, 4>{});
    do_class(std::integral_constant<int, 5>{});
    do_class(std::integral_constant<int, 6>{});
    do_class(std::integral_constant<int, 7>{});
  }
}

// weight gradient:  D[tap][sc][lc] = sum_{n,i} S[n,i][sc] * L[n, STRIDE*i-1+tap][lc]
// bf16: both operands are contracted over VOXELS, which are the slow axis of the channels-last LDS
// rows -> read with ds_read_b64_tr_b16 (hardware transpose, 4 voxels x 16 channels per 16 lanes).
// f32: v_mfma_f32_32x32x2_f32 takes one scalar per lane, plain ds_read_b32.
template <typename T>
struct WG;
template <>
struct WG<bf16_t> {
  static constexpr int KV = 16;  // voxels contracted per MFMA step
};
template <>
struct WG<f16_t> {
  static constexpr int KV = 16;
};
template <>
struct WG<float> {
  static constexpr int KV = 2;
};

// FLAT (round 6): the 2-D weight gradients of models/HDenseFormer_2D.py (Conv2d, ConvTranspose2d) on depth-1 tensors: the 9
// taps of the centre depth plane (three per wave, the fourth wave only stages), a one-plane box of the large operand, a
// depth axis that is never strided; the other 18 taps of the [27] slab are written as zeros.
template <typename T, int TD, int TH, int TW, int S, bool FLAT = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  static_assert(!FLAT || TD == 1, "flat tiles are one voxel deep");
  constexpr int MT = TD * TH * TW;
  constexpr int BD = FLAT ? 1 : S * (TD - 1) + 3, BH = S * (TH - 1) + 3, BW = S * (TW - 1) + 3;
  constexpr int NTAP = FLAT ? 9 : 27, TAP0 = FLAT ? 9 : 0, ZO = FLAT ? 0 : 1;   // taps, first tap of the slab, z halo
  constexpr int ROWB = 32 * sizeof(T);  // 32 channels per LDS row
  // LDS row pitch.  bf16 transposed reads touch 4 voxel rows x 16 dwords per half-wave: with stride 2 those rows are
  // 2 box rows apart, and a 96-byte pitch (24 dwords: 0, 48, 32, 16 mod 64) tiles the 64 banks exactly; 80 bytes
  // overlapped the 1st and 4th row (2-way conflicts).  (Stride 1 bf16 runs conv_wgrad2_kernel with 64-byte rows.)
  constexpr int LP = (sizeof(T) == 2 && S == 2) ? ROWB + 32 : ROWB + 16;
  constexpr int TAPS_PER_WAVE = FLAT ? 3 : 7;
  __shared__ __attribute__((aligned(16))) char lds[(MT + BD * BH * BW) * LP];
  char* s_lds = lds;
  char* l_lds = lds + MT * LP;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int scb = blockIdx.y, lcb = blockIdx.z;
  const int ntz = (a.Ds + TD - 1) / TD, nty = (a.Hs + TH - 1) / TH, ntx = (a.Ws + TW - 1) / TW;

  f32x16 acc[TAPS_PER_WAVE];
#pragma unroll
  for (int j = 0; j < TAPS_PER_WAVE; j++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[j][i] = 0.f;

  int tapoff[TAPS_PER_WAVE];
#pragma unroll
  for (int j = 0; j < TAPS_PER_WAVE; j++) {
    int tap = wave * TAPS_PER_WAVE + j;
    if (tap >= NTAP) tap = 0;  // dummy slot of the last wave (never stored)
    int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    tapoff[j] = ((kz * BH + ky) * BW + kx) * LP;
  }
  const int ntaps_here = max(0, min(TAPS_PER_WAVE, NTAP - wave * TAPS_PER_WAVE));  // 7,7,7,6 (flat: 3,3,3,0)

  const int t_begin = blockIdx.x * a.tiles_per_group;
  const int t_end = min(a.num_tiles, t_begin + a.tiles_per_group);
  auto tile_origin = [&](int tile, int& n, int& z0, int& y0, int& x0) {
    int t = tile;
    x0 = (t % ntx) * TW;
    t /= ntx;
    y0 = (t % nty) * TH;
    t /= nty;
    z0 = (t % ntz) * TD;
    n = t / ntz;
  };

  if constexpr (sizeof(T) == 2) {
    // ---- bf16: register-prefetched tiles (the next tile's global loads fly during this tile's MFMAs) ----------
    constexpr int CPV = ROWB / 16;                       // 4 chunks of 16 B per voxel row
    constexpr int BOXL = BD * BH * BW;
    constexpr int NS_ = (MT * CPV + 255) / 256, NL_ = (BOXL * CPV + 255) / 256;
    constexpr int EPC = ST<T>::EPC;
    const int part = threadIdx.x & (CPV - 1);
    u32x4 ps[NS_], pl[NL_];
    uint32_t vs = 0, vl = 0;
    auto prefetch = [&](int tile) {
      int n, z0, y0, x0;
      tile_origin(tile, n, z0, y0, x0);
      const T* ssrc = reinterpret_cast<const T*>(a.sm) + scb * 32 + part * EPC;
      const T* lsrc = reinterpret_cast<const T*>(a.lg) + lcb * 32 + part * EPC;
      const bool sc_ok = scb * 32 + part * EPC < a.SC, lc_ok = lcb * 32 + part * EPC < a.LC;
      vs = vl = 0;
#pragma unroll
      for (int j = 0; j < NS_; j++) {
        int vox = min((int)threadIdx.x + 256 * j, MT * CPV - 1) / CPV;
        int bz = vox / (TH * TW), rem = vox - bz * (TH * TW), by = rem / TW, bx = rem - by * TW;
        int iz = z0 + bz, iy = y0 + by, ix = x0 + bx;
        bool ok = sc_ok && iz < a.Ds && iy < a.Hs && ix < a.Ws;
        const T* p = ok ? ssrc + ((((int64_t)n * a.Ds + iz) * a.Hs + iy) * a.Ws + ix) * a.sm_pitch : ssrc - part * EPC - scb * 32;
        ps[j] = *reinterpret_cast<const u32x4*>(p);
        vs |= (ok ? 1u : 0u) << j;
      }
#pragma unroll
      for (int j = 0; j < NL_; j++) {
        int vox = min((int)threadIdx.x + 256 * j, BOXL * CPV - 1) / CPV;
        int bz = vox / (BH * BW), rem = vox - bz * (BH * BW), by = rem / BW, bx = rem - by * BW;
        int iz = (FLAT ? z0 : S * z0 - ZO) + bz, iy = S * y0 - 1 + by, ix = S * x0 - 1 + bx;
        bool ok = lc_ok && (unsigned)iz < (unsigned)a.Dl && (unsigned)iy < (unsigned)a.Hl && (unsigned)ix < (unsigned)a.Wl;
        const T* p = ok ? lsrc + ((((int64_t)n * a.Dl + iz) * a.Hl + iy) * a.Wl + ix) * a.lg_pitch : lsrc - part * EPC - lcb * 32;
        pl[j] = *reinterpret_cast<const u32x4*>(p);
        vl |= (ok ? 1u : 0u) << j;
      }
    };
    auto xform = [&](u32x4 v, const float* sc, const float* sh, int relu) {
      float f[EPC];
      ST<T>::unpack(v, f);
#pragma unroll
      for (int e = 0; e < EPC; e++) {
        f[e] = f[e] * sc[e] + sh[e];
        if (relu) f[e] = fmaxf(f[e], 0.f);
      }
      return ST<T>::pack(f);
    };
    auto commit = [&](int tile) {
      int n, z0, y0, x0;
      tile_origin(tile, n, z0, y0, x0);
      float sc[EPC], sh[EPC];
      if (a.sm_scale) {
#pragma unroll
        for (int e = 0; e < EPC; e++) {
          int c = min(scb * 32 + part * EPC + e, a.SC - 1);
          sc[e] = a.sm_scale[(int64_t)n * a.SC + c];
          sh[e] = a.sm_shift[(int64_t)n * a.SC + c];
        }
      }
#pragma unroll
      for (int j = 0; j < NS_; j++) {
        int id = threadIdx.x + 256 * j;
        if (id < MT * CPV) {
          u32x4 v = ps[j];
          if (a.sm_scale) v = xform(v, sc, sh, a.sm_relu);
          if (!((vs >> j) & 1u)) v = u32x4{0u, 0u, 0u, 0u};
          *reinterpret_cast<u32x4*>(s_lds + (id / CPV) * LP + part * 16) = v;
        }
      }
      if (a.lg_scale) {
#pragma unroll
        for (int e = 0; e < EPC; e++) {
          int c = min(lcb * 32 + part * EPC + e, a.LC - 1);
          sc[e] = a.lg_scale[(int64_t)n * a.LC + c];
          sh[e] = a.lg_shift[(int64_t)n * a.LC + c];
        }
      }
#pragma unroll
      for (int j = 0; j < NL_; j++) {
        int id = threadIdx.x + 256 * j;
        if (id < BOXL * CPV) {
          u32x4 v = pl[j];
          if (a.lg_scale) v = xform(v, sc, sh, a.lg_relu);
          if (!((vl >> j) & 1u)) v = u32x4{0u, 0u, 0u, 0u};
          *reinterpret_cast<u32x4*>(l_lds + (id / CPV) * LP + part * 16) = v;
        }
      }
    };
    // lane roles for ds_read_b64_tr_b16: group g4 = lane>>4 ; within group i = lane&15, q = i>>2, p = i&3
    const int g4 = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3;
    const int hh = g4 >> 1, cb = (g4 & 1) * 16;
    const int colb = (cb + 4 * p4) * 2;
    using lds_s16x4 = s16x4 __attribute__((address_space(3)));
    auto tr_read = [&](const char* ptr) {
      s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)ptr);
      return __builtin_bit_cast(u32x2, v);
    };
    if (t_begin < t_end) prefetch(t_begin);
    for (int tile = t_begin; tile < t_end; tile++) {
      commit(tile);
      WS_BARRIER();
      if (tile + 1 < t_end) prefetch(tile + 1);
      for (int ks = 0; ks < MT / 16; ks++) {
        // the two 4-voxel groups this lane addresses: k = 8*hh + 4*t + q
        int sA[2], lB[2];
#pragma unroll
        for (int tt = 0; tt < 2; tt++) {
          int lin = ks * 16 + 8 * hh + 4 * tt + q;
          int lz = lin / (TH * TW), ly = (lin / TW) % TH, lx = lin % TW;
          sA[tt] = lin * LP + colb;
          lB[tt] = ((((FLAT ? 0 : S * lz)) * BH + S * ly) * BW + S * lx) * LP + colb;
        }
        // all 16 transposed reads of this k-step are requested before its 7 MFMAs (wave 3's 7th tap is a dummy
        // pointing at tap 0: uniform instruction stream, its accumulator is never stored)
        u32x2 a0u = tr_read(s_lds + sA[0]), a1u = tr_read(s_lds + sA[1]);
        u32x2 b0u[TAPS_PER_WAVE], b1u[TAPS_PER_WAVE];
#pragma unroll
        for (int j = 0; j < TAPS_PER_WAVE; j++) {
          b0u[j] = tr_read(l_lds + lB[0] + tapoff[j]);
          b1u[j] = tr_read(l_lds + lB[1] + tapoff[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        u32x4 af = {a0u[0], a0u[1], a1u[0], a1u[1]};
#pragma unroll
        for (int j = 0; j < TAPS_PER_WAVE; j++) {
          u32x4 bf = {b0u[j][0], b0u[j][1], b1u[j][0], b1u[j][1]};
          Mma<T>::run(af, bf, acc[j]);
        }
      }
      WS_BARRIER();
    }
  } else {
    // ---- f32 (parity path): plain stage -> barrier -> compute -> barrier ------------------------------------------
    for (int tile = t_begin; tile < t_end; tile++) {
      int n, z0, y0, x0;
      tile_origin(tile, n, z0, y0, x0);
      if (tile > t_begin) __syncthreads();
      stage_box<T, TD, TH, TW, ROWB, LP>(s_lds, reinterpret_cast<const T*>(a.sm), a.sm_pitch, a.SC, n, a.Ds, a.Hs, a.Ws,
                                         z0, y0, x0, scb * 32, ROWB, a.sm_scale, a.sm_shift, a.sm_relu);
      stage_box<T, BD, BH, BW, ROWB, LP>(l_lds, reinterpret_cast<const T*>(a.lg), a.lg_pitch, a.LC, n, a.Dl, a.Hl, a.Wl,
                                         FLAT ? z0 : S * z0 - 1, S * y0 - 1, S * x0 - 1, lcb * 32, ROWB, a.lg_scale, a.lg_shift,
                                         a.lg_relu);
      __syncthreads();
      const int r = lane & 31, hh = lane >> 5;
      for (int ks = 0; ks < MT / 2; ks++) {
        int lin = ks * 2 + hh;
        int lz = lin / (TH * TW), ly = (lin / TW) % TH, lx = lin % TW;
        float av = *reinterpret_cast<const float*>(s_lds + lin * LP + r * 4);
        const char* lb = l_lds + ((((FLAT ? 0 : S * lz)) * BH + S * ly) * BW + S * lx) * LP + r * 4;
#pragma unroll
        for (int j = 0; j < TAPS_PER_WAVE; j++) {
          if (j < ntaps_here) {
            float bv = *reinterpret_cast<const float*>(lb + tapoff[j]);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
          }
        }
      }
    }
  }

  // partial[g][tap][SCp][LCp]
  const int col = lane & 31, hh2 = lane >> 5;
#pragma unroll
  for (int j = 0; j < TAPS_PER_WAVE; j++) {
    if (j < ntaps_here) {
      int tap = TAP0 + wave * TAPS_PER_WAVE + j;
      float* base = a.partials + (((int64_t)blockIdx.x * 27 + tap) * a.SCp + scb * 32) * a.LCp + lcb * 32 + col;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        int row = (i & 3) + 8 * (i >> 2) + 4 * hh2;
        base[(int64_t)row * a.LCp] = acc[j][i];
      }
    }
  }
  if constexpr (FLAT) {   // the 18 taps off the centre depth plane: zero blocks (the slab reduction sums all 27)
    for (int t = wave; t < 18; t += 4) {
      const int tap = t < 9 ? t : t + 9;
      float* base = a.partials + (((int64_t)blockIdx.x * 27 + tap) * a.SCp + scb * 32) * a.LCp + lcb * 32 + col;
#pragma unroll
      for (int i = 0; i < 16; i++) base[(int64_t)((i & 3) + 8 * (i >> 2) + 4 * hh2) * a.LCp] = 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// conv_wgrad2: bf16 stride-1 weight gradient with DOUBLE-BUFFERED tiles (same idea as conv_ws2_kernel).
// conv_wgrad_kernel spends ~3 cycles outside the matrix pipe per MFMA cycle: per-slot index arithmetic in the
// prefetch, the transform + LDS commit, and a transposed-read latency exposed in every k-step.  Here the tile
// under the MFMAs (T0, buffer PAR) is read with a one-k-step look-ahead while, in the gaps of the same stream,
// tile T1 goes registers -> (InstanceNorm/ReLU) -> the other buffer and tile T2's loads refill the registers.
// One barrier per tile; slot offsets are per-thread constants; tile coordinates advance incrementally.
// AP (WgradArgs::ap_*): the small operand arrives as d(activation); its InstanceNorm(+ReLU) backward is applied on the way
// into LDS (in_bwd_elem) from a second staged stream (the layer's raw output y), and the rows leave for ap_out as well.
template <typename T, bool XFL, bool AP = false>
__global__ __launch_bounds__(256) void conv_wgrad2_kernel(WgradArgs a) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int TD = 4, TH = 8, TW = 8, MT = TD * TH * TW, BD = TD + 2, BH = TH + 2, BW = TW + 2, BOXL = BD * BH * BW;
  // unpadded 64-byte rows: the 4 voxel rows x 16 dwords a ds_read_b64_tr_b16 half-wave touches then tile the 64
  // banks exactly (an 80-byte pitch wraps the 4th row onto the 1st: 2-way conflicts on every read)
  constexpr int LP = 64, CPV = 4, EPC = 8;
  constexpr int SBUF = MT * LP, LBUF = BOXL * LP, BUF = SBUF + LBUF;
  constexpr int NS_ = MT * CPV / 256, NL_ = (BOXL * CPV + 255) / 256, NSLOT = NS_ + NL_;
  constexpr int KS = MT / 16, NT = 7;  // k-steps per tile, taps per wave (7,7,7,6 + one dummy)
  constexpr int PD = 6, NPF = PD + 1;  // a slot's global load is issued PD k-steps before its commit; register ring
  static_assert(NSLOT + 2 <= KS && NSLOT >= PD, "slot schedule: commits at k-steps 0..NSLOT-1, loads PD steps ahead");
  __shared__ __attribute__((aligned(256))) char lds[2 * BUF + 256 + (AP ? 7 * 32 * 4 : 0)];
  float* const s_xf = reinterpret_cast<float*>(lds + 2 * BUF);  // [32 scale][32 shift] of the large operand
  float* const s_ap = reinterpret_cast<float*>(lds + 2 * BUF + 256);  // AP: [7][32] constants of the small operand's block

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int scb = blockIdx.y, lcb = blockIdx.z;
  const int part = tid & (CPV - 1);
  const int ntz = (a.Ds + TD - 1) / TD, nty = (a.Hs + TH - 1) / TH, ntx = (a.Ws + TW - 1) / TW;
  const float relu_lo = (XFL && a.lg_relu) ? 0.f : -INFINITY;

  // ---- transposed-read addresses (lane roles of ds_read_b64_tr_b16: 4 voxels x 16 channels per 16 lanes)
  const int g4 = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3;
  const int hh = g4 >> 1, cb = (g4 & 1) * 16;
  const int colb = (cb + 4 * p4) * 2;
  // per tap; the voxel group tt and the k-step ride in the 16-bit immediate offset, the buffer is added per tile
  int sA, lB[NT];
  sA = (8 * hh + q) * LP + colb;
#pragma unroll
  for (int j = 0; j < NT; j++) {
    int tap = wave * NT + j;
    if (tap >= 27) tap = 0;  // dummy slot of the last wave (never stored)
    const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    lB[j] = SBUF + (((kz * BH + ky + hh) * BW) + kx + q) * LP + colb;
  }
  const int ntaps_here = min(NT, 27 - wave * NT);

  // ---- staging slots (per-thread constants): slot s < NS_ -> small tile, else large box
  // small tile: slot s holds voxel v0 + 64 s = one z-plane further (everything else is a compile-time offset);
  // large box: explicit per-slot constants
  const int v0 = tid >> 2, s_by = (v0 >> 3) & 7, s_bx = v0 & 7;  // v0 < 64: z-plane 0
  const int s_goff0 = (s_by * a.Ws + s_bx) * (int)a.sm_pitch;
  const int s_plane = a.Hs * a.Ws * (int)a.sm_pitch;
  const int w0 = v0 * LP + part * 16;
  int goffL[NL_], gxyzL[NL_];
#pragma unroll
  for (int k = 0; k < NL_; k++) {
    // threads past the end of the last slot redo the box's last voxel (same part): no predicate needed
    const int vox = min(tid + 256 * k, BOXL * CPV - CPV + part) >> 2;
    const int bz = vox / (BH * BW), rem = vox - bz * (BH * BW), by = rem / BW, bx = rem - by * BW;
    goffL[k] = ((bz * a.Hl + by) * a.Wl + bx) * (int)a.lg_pitch;
    gxyzL[k] = (bz << 16) | (by << 8) | bx;
  }
  const int w_last = SBUF + (min(tid + 256 * (NL_ - 1), BOXL * CPV - CPV + part) >> 2) * LP + part * 16;
  auto goff = [&](int s) { return s < NS_ ? s_goff0 + s * s_plane : goffL[s - NS_]; };
  auto woff = [&](int s) {
    return s < NS_ ? w0 + s * 64 * LP : (s < NSLOT - 1 ? SBUF + w0 + (s - NS_) * 64 * LP : w_last);
  };
  const bool sc_ok = scb * 32 + part * EPC < a.SC, lc_ok = lcb * 32 + part * EPC < a.LC;
  const bool chan_all = (a.SC % 32 == 0) && (a.LC % 32 == 0);
  // AP: y is read and dy written through buffer descriptors with 32-bit byte offsets (launcher: both tensors < 2 GiB); an
  // offset with bit 31 set is out of range -- such a load returns 0 and such a store is dropped, so neither is ever
  // branched around.  Slot s of a tile: tile offset (uniform) + v0 offset + s planes.
  constexpr uint32_t AP_OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t ap_ry =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(AP ? a.ap_y : nullptr), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t ap_ro = __builtin_amdgcn_make_buffer_rsrc(AP ? a.ap_out : nullptr, 0, 0x7fffffff, 0x00020000);
  const uint32_t ap_yv0 = AP ? (uint32_t)(((s_by * a.Ws + s_bx) * (int)a.ap_y_pitch + scb * 32 + part * EPC) * 2) : 0u;
  const uint32_t ap_yplane = AP ? (uint32_t)(a.Hs * a.Ws * (int)a.ap_y_pitch * 2) : 0u;
  const uint32_t ap_ov0 = AP ? (uint32_t)(((s_by * a.Ws + s_bx) * (int)a.ap_out_pitch + scb * 32 + part * EPC) * 2) : 0u;
  const uint32_t ap_oplane = AP ? (uint32_t)(a.Hs * a.Ws * (int)a.ap_out_pitch * 2) : 0u;
  auto ap_tile_vox = [&](const WsTile& c) { return (uint32_t)(((c.n * a.Ds + c.z0) * a.Hs + c.y0) * a.Ws + c.x0); };
  auto ap_ty_of = [&](const WsTile& c, bool valid) { return valid ? ap_tile_vox(c) * (uint32_t)a.ap_y_pitch * 2u : AP_OOB; };
  auto ap_to_of = [&](const WsTile& c, bool valid) {  // only the workgroups of large-channel block 0 store
    return (valid && lcb == 0) ? ap_tile_vox(c) * (uint32_t)a.ap_out_pitch * 2u : AP_OOB;
  };
  const T* const s_safe = reinterpret_cast<const T*>(a.sm);
  const T* const l_safe = reinterpret_cast<const T*>(a.lg);
  const T* const s_src = s_safe + scb * 32 + part * EPC;
  const T* const l_src = l_safe + lcb * 32 + part * EPC;

  // Tile schedule (as in conv_ws2_kernel): an interior pass (unchecked copy of the phase, one long run) and a
  // border pass (checked copy); inside each, XCD x owns the x-th eighth of the raster-ordered list and its
  // gridDim.x/8 workgroups walk it interleaved, so neighbouring tiles' halos meet in that XCD's L2.
  const bool has_int = chan_all && ntz >= 3 && nty >= 3 && ntx >= 3;
  const int ipz = ntz - 2, ipy = nty - 2, ipx = ntx - 2;
  const int n_int = has_int ? a.N * ipz * ipy * ipx : 0;
  const int per_bor = ntz * nty * ntx - (has_int ? ipz * ipy * ipx : 0);
  const int n_bor = a.N * per_bor;
  const int G = gridDim.x;
  const int NX = (G % 8 == 0) ? 8 : 1;
  const int WPX = G / NX;
  const int xcd = blockIdx.x % NX, slot = blockIdx.x / NX;
  auto split = [&](int total, int& begin, int& cnt) {
    const int r0 = (int)((int64_t)total * xcd / NX), r1 = (int)((int64_t)total * (xcd + 1) / NX);
    begin = r0 + slot;
    cnt = (r1 - r0 > slot) ? (r1 - r0 - slot + WPX - 1) / WPX : 0;
  };
  int int_begin, int_cnt, bor_begin, bor_cnt;
  split(n_int, int_begin, int_cnt);
  split(n_bor, bor_begin, bor_cnt);
  int sdx = 0, sdy = 0, sdz = 0, sdn = 0;  // mixed-radix digits of the stride WPX over the interior tile grid
  if (has_int) {
    int t = WPX;
    sdx = t % ipx, t /= ipx;
    sdy = t % ipy, t /= ipy;
    sdz = t % ipz, sdn = t / ipz;
  }
  auto int_init = [&](WsTile& c, int k) {
    int t = k;
    c.x0 = (t % ipx + 1) * TW;
    t /= ipx;
    c.y0 = (t % ipy + 1) * TH;
    t /= ipy;
    c.z0 = (t % ipz + 1) * TD;
    c.n = t / ipz;
    c.k = k;
  };
  auto int_next = [&](WsTile& c) {  // + WPX tiles in raster order of the interior grid
    int xi = (c.x0 >> 3) - 1 + sdx, yi = (c.y0 >> 3) - 1 + sdy, zi = (c.z0 >> 2) - 1 + sdz;
    c.n += sdn;
    if (xi >= ipx) xi -= ipx, yi++;
    if (yi >= ipy) yi -= ipy, zi++;
    if (zi >= ipz) zi -= ipz, c.n++;
    c.x0 = (xi + 1) * TW, c.y0 = (yi + 1) * TH, c.z0 = (zi + 1) * TD;
  };
  auto bor_init = [&](WsTile& c, int k) {
    int tz, ty, tx;
    c.k = k;
    c.n = k / per_bor;
    int rem = k - c.n * per_bor;
    if (!has_int) {
      tx = rem % ntx, ty = (rem / ntx) % nty, tz = rem / (ntx * nty);
    } else {
      const int plane = nty * ntx, ring = plane - ipy * ipx;  // border tiles of a z-plane: all of it / its rim
      if (rem < plane) {
        tz = 0, ty = rem / ntx, tx = rem % ntx;
      } else if (rem - plane < ipz * ring) {
        rem -= plane;
        tz = 1 + rem / ring;
        rem %= ring;
        if (rem < ntx) {
          ty = 0, tx = rem;
        } else if (rem - ntx < 2 * ipy) {
          rem -= ntx;
          ty = 1 + (rem >> 1), tx = (rem & 1) ? ntx - 1 : 0;
        } else {
          ty = nty - 1, tx = rem - ntx - 2 * ipy;
        }
      } else {
        rem -= plane + ipz * ring;
        tz = ntz - 1, ty = rem / ntx, tx = rem % ntx;
      }
    }
    c.z0 = tz * TD, c.y0 = ty * TH, c.x0 = tx * TW;
  };
  auto bor_next = [&](WsTile& c) { bor_init(c, c.k + WPX); };
  auto s_org_of = [&](const WsTile& c) -> const T* {
    return s_src + ((((int64_t)c.n * a.Ds + c.z0) * a.Hs + c.y0) * a.Ws + c.x0) * a.sm_pitch;
  };
  auto l_org_of = [&](const WsTile& c) -> const T* {
    return l_src + ((((int64_t)c.n * a.Dl + (c.z0 - 1)) * a.Hl + (c.y0 - 1)) * a.Wl + (c.x0 - 1)) * a.lg_pitch;
  };
  auto slot_ok = [&](int s, const WsTile& c) {
    if (s < NS_) return sc_ok & (c.z0 + s < a.Ds) & (c.y0 + s_by < a.Hs) & (c.x0 + s_bx < a.Ws);
    const int g = gxyzL[s - NS_];
    const int bz = g >> 16, by = (g >> 8) & 255, bx = g & 255;
    return lc_ok & ((unsigned)(c.z0 - 1 + bz) < (unsigned)a.Dl) & ((unsigned)(c.y0 - 1 + by) < (unsigned)a.Hl) &
           ((unsigned)(c.x0 - 1 + bx) < (unsigned)a.Wl);
  };

  u32x4 pf[NPF];  // slot s lives in pf[s % NPF] from its load to its commit PD k-steps later
  u32x4 py[AP ? NS_ : 1];  // AP: the y chunk of small slot s, requested together with its d(activation) chunk
  // unconditional loads from a clamped address (never branch around a load)
  // ty (AP): byte offset of the tile's first voxel in y, or AP_OOB for a tile past the end of the list
  auto load_one = [&](auto fast_tag, int s, const WsTile& c, bool valid, const T* sorg, const T* lorg, uint32_t ty) {
    const T* org = (s < NS_) ? sorg : lorg;
    if constexpr (decltype(fast_tag)::value) {
      pf[s % NPF] = *reinterpret_cast<const u32x4*>(org + goff(s));
      if constexpr (AP)
        if (s < NS_) py[s] = __builtin_amdgcn_raw_buffer_load_b128(ap_ry, ty + ap_yv0 + s * ap_yplane, 0, 0);
    } else {
      const bool ok = valid & slot_ok(s, c);
      const T* p = ok ? org + goff(s) : ((s < NS_) ? s_safe : l_safe);
      pf[s % NPF] = *reinterpret_cast<const u32x4*>(p);
      if constexpr (AP)
        if (s < NS_) py[s] = __builtin_amdgcn_raw_buffer_load_b128(ap_ry, ok ? ty + ap_yv0 + s * ap_yplane : AP_OOB, 0, 0);
    }
  };
  float sc[EPC], sh[EPC];
  auto read_xf = [&]() {
#pragma unroll
    for (int e = 0; e < EPC; e += 4) {
      f32x4 u = *reinterpret_cast<const f32x4*>(s_xf + part * EPC + e);
      f32x4 v = *reinterpret_cast<const f32x4*>(s_xf + 32 + part * EPC + e);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        sc[e + k] = u[k];
        sh[e + k] = v[k];
      }
    }
  };
  // AP: the seven constants of this thread's 8 channels, re-read from s_ap at the top of every tile (56 registers that live
  // for the four small-slot commits only)
  float apk[AP ? 7 : 1][EPC];
  auto read_ap = [&]() {
    if constexpr (AP) {
#pragma unroll
      for (int k = 0; k < 7; k++)
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
          const f32x4 u = *reinterpret_cast<const f32x4*>(s_ap + k * 32 + part * EPC + e);
#pragma unroll
          for (int i = 0; i < 4; i++) apk[k][e + i] = u[i];
        }
    }
  };
  // to (AP): byte offset of the tile's first voxel in ap_out, or AP_OOB (tile past the end / not this workgroup's to store)
  auto commit_one = [&](auto fast_tag, int s, const WsTile& c, char* dst, uint32_t to) {
    u32x4 v = pf[s % NPF];
    if constexpr (XFL) {
      if (s >= NS_) {
        float f[EPC];
        ST<T>::unpack(v, f);
#pragma unroll
        for (int e = 0; e < EPC; e++) f[e] = fmaxf(f[e] * sc[e] + sh[e], relu_lo);
        v = ST<T>::pack(f);
      }
    }
    if constexpr (AP) {
      if (s < NS_) {
        float g[EPC], f[EPC];
        ST<T>::unpack(v, g);
        ST<T>::unpack(py[s < NS_ ? s : 0], f);
#pragma unroll
        for (int e = 0; e < EPC; e++)
          g[e] = in_bwd_elem(g[e], f[e], apk[0][e], apk[1][e], apk[2][e], apk[3][e], apk[4][e], apk[5][e], apk[6][e]);
        v = ST<T>::pack(g);
      }
    }
    bool ok = true;
    if constexpr (!decltype(fast_tag)::value) {
      ok = slot_ok(s, c);
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = ok ? v[k] : 0u;
    }
    *reinterpret_cast<u32x4*>(dst + woff(s)) = v;
    if constexpr (AP)
      if (s < NS_) __builtin_amdgcn_raw_buffer_store_b128(v, ap_ro, ok ? to + ap_ov0 + s * ap_oplane : AP_OOB, 0, 0);
  };
  int tbl_n = -1;
  auto refresh_xf = [&](int n) {  // uniform; nobody reads the old tables any more (sc/sh hold theirs; apk is per tile)
    if constexpr (XFL) {
      if (tid < 32) {
        const int c = min(lcb * 32 + tid, a.LC - 1);
        s_xf[tid] = a.lg_scale[(int64_t)n * a.LC + c];
        s_xf[32 + tid] = a.lg_shift[(int64_t)n * a.LC + c];
      }
    }
    if constexpr (AP) {
      if (tid >= 32) {
        const int i = tid - 32, k = i >> 5, c = min(scb * 32 + (i & 31), a.SC - 1);
        s_ap[i] = a.ap_tab[k][(int64_t)n * a.SC + c];
      }
    }
    tbl_n = n;
    __syncthreads();
    if constexpr (XFL) read_xf();
  };

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; j++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[j][i] = 0.f;

#ifdef WS_DBG_STAMPS
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
  {
    WsTile T0, T1, T2;
    bool v1 = false, v2 = false;
    const T *so1 = s_safe, *lo1 = l_safe, *so2 = s_safe, *lo2 = l_safe;
    uint32_t ty1 = AP_OOB, ty2 = AP_OOB, to1 = AP_OOB, to2 = AP_OOB;  // AP: tile offsets in y / ap_out (see load_one, commit_one)
    int left = 0, par = 0;
    // start a pass at its k-th tile: T0 -> buffer 0 (not overlapped, NPF slots at a time), first PD slots of T1 ->
    // registers
    auto begin_pass = [&](auto border_tag, int k, int count) __attribute__((always_inline)) {
      constexpr bool BORDER = decltype(border_tag)::value;
      auto next = [&](WsTile& c) {
        if constexpr (BORDER)
          bor_next(c);
        else
          int_next(c);
      };
      if constexpr (BORDER)
        bor_init(T0, k);
      else
        int_init(T0, k);
      left = count - 1;
      T1 = T0;
      next(T1);
      T2 = T1;
      next(T2);
      v1 = left >= 1, v2 = left >= 2;
      so1 = v1 ? s_org_of(T1) : s_safe;
      lo1 = v1 ? l_org_of(T1) : l_safe;
      so2 = v2 ? s_org_of(T2) : s_safe;
      lo2 = v2 ? l_org_of(T2) : l_safe;
      const T* so0 = s_org_of(T0);
      const T* lo0 = l_org_of(T0);
      ty1 = ap_ty_of(T1, v1), ty2 = ap_ty_of(T2, v2);
      to1 = ap_to_of(T1, v1), to2 = ap_to_of(T2, v2);
      __syncthreads();  // the previous pass is done with both buffers
      par = 0;
      if constexpr (XFL || AP) refresh_xf(T0.n);
      read_ap();
#pragma unroll
      for (int s0 = 0; s0 < NSLOT; s0 += NPF) {
#pragma unroll
        for (int s = s0; s < s0 + NPF && s < NSLOT; s++) load_one(std::false_type{}, s, T0, true, so0, lo0, ap_ty_of(T0, true));
#pragma unroll
        for (int s = s0; s < s0 + NPF && s < NSLOT; s++) commit_one(std::false_type{}, s, T0, lds, ap_to_of(T0, true));
      }
#pragma unroll
      for (int s = 0; s < PD; s++) load_one(std::false_type{}, s, T1, v1, so1, lo1, ty1);
      WS_BARRIER();
    };

    using lds_s16x4 = s16x4 __attribute__((address_space(3)));
    auto tr_read = [&](int off) {
      s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off));
      return __builtin_bit_cast(u32x2, v);
    };
    // The buffer parity is a RUN-TIME value folded into the 8 read-address registers: with it as a template
    // parameter the copies of the phase disagreed on where the in-flight staging registers live, and the compiler
    // drained every outstanding load (s_waitcnt vmcnt(0)) at the loop's back edge.
    auto tile_phase = [&](auto fast_tag) __attribute__((always_inline)) {
      constexpr int FASTI = decltype(fast_tag)::value ? 0 : 4;
      (void)FASTI;
      char* const a_wr = lds + (1 - par) * BUF;
      const int sAw = sA + par * BUF;
      int lBw[NT];
#pragma unroll
      for (int j = 0; j < NT; j++) lBw[j] = lB[j] + par * BUF;
      if constexpr (XFL || AP) {
        if (v1 && T1.n != tbl_n) refresh_xf(T1.n);
      }
      read_ap();
      WS2_STAMP(0)
      // B fragments: ONE register set, re-read for k-step ks+1 right behind the MFMA that consumed them (the arch
      // VGPR file is 256 deep: a second set pushed the staging ring into AGPR/scratch spills); A: two sets
      u32x2 A0[2], A1[2], B0[NT], B1[NT];
      auto koff = [&](int ks) { return ((ks >> 2) * BH * BW + 2 * (ks & 3) * BW) * LP; };
      auto read_a = [&](int ks) {
        A0[ks & 1] = tr_read(sAw + ks * 16 * LP);
        A1[ks & 1] = tr_read(sAw + ks * 16 * LP + 4 * LP);
      };
      auto read_b = [&](int ks, int j) {
        B0[j] = tr_read(lBw[j] + koff(ks));
        B1[j] = tr_read(lBw[j] + koff(ks) + 4 * LP);
      };
      read_a(0);
#pragma unroll
      for (int j = 0; j < NT; j++) read_b(0, j);
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        if (ks + 1 < KS) read_a(ks + 1);
        // staging: commit slot ks of T1; load the slot that commits PD k-steps from now (T1's, or T2's when that
        // falls into the next tile phase)
        if (ks < NSLOT) commit_one(fast_tag, ks, T1, a_wr, to1);
        if (ks + PD < NSLOT)
          load_one(fast_tag, ks + PD, T1, v1, so1, lo1, ty1);
        else if (ks + PD >= KS)
          load_one(fast_tag, ks + PD - KS, T2, v2, so2, lo2, ty2);
        const u32x4 af = {A0[ks & 1][0], A0[ks & 1][1], A1[ks & 1][0], A1[ks & 1][1]};
#pragma unroll
        for (int j = 0; j < NT; j++) {
          const u32x4 bf = {B0[j][0], B0[j][1], B1[j][0], B1[j][1]};
          Mma<T>::run(af, bf, acc[j]);
          if (ks + 1 < KS) read_b(ks + 1, j);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      WS2_STAMP(1 + FASTI)
      WS_BARRIER();  // buffer PAR fully read, buffer 1-PAR fully written
      WS2_STAMP(2)
      // pin the loop-carried accumulators to AGPRs: left alone the compiler carries them in VGPRs between tile
      // phases and pays 2 x 112 v_accvgpr moves per tile
#pragma unroll
      for (int j = 0; j < NT; j++) asm volatile("" : "+a"(acc[j]));
    };

    bool more = true;
    auto step = [&](auto border_tag) __attribute__((always_inline)) {
      constexpr bool BORDER = decltype(border_tag)::value;
      par ^= 1;
      more = v1;
      if (!more) return;
      left--;
      T0 = T1;
      T1 = T2;
      v1 = v2;
      so1 = so2;
      lo1 = lo2;
      ty1 = ty2, to1 = to2;
      if constexpr (BORDER)
        bor_next(T2);
      else
        int_next(T2);
      v2 = left >= 2;
      so2 = v2 ? s_org_of(T2) : s_safe;
      lo2 = v2 ? l_org_of(T2) : l_safe;
      ty2 = ap_ty_of(T2, v2), to2 = ap_to_of(T2, v2);
    };
    if (int_cnt > 0) {  // interior pass: the unchecked copy, one run (its own back edge: nothing drained per tile)
      begin_pass(std::false_type{}, int_begin, int_cnt);
      more = true;
      while (more) {
        tile_phase(std::true_type{});
        step(std::false_type{});
      }
    }
    if (bor_cnt > 0) {  // border pass: the checked copy
      begin_pass(std::true_type{}, bor_begin, bor_cnt);
      more = true;
      while (more) {
        tile_phase(std::false_type{});
        step(std::true_type{});
      }
    }
  }

  // partial[g][tap][SCp][LCp]
  const int col = lane & 31, hh2 = lane >> 5;
#pragma unroll
  for (int j = 0; j < NT; j++) {
    if (j < ntaps_here) {
      int tap = wave * NT + j;
      float* base = a.partials + (((int64_t)blockIdx.x * 27 + tap) * a.SCp + scb * 32) * a.LCp + lcb * 32 + col;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        int row = (i & 3) + 8 * (i >> 2) + 4 * hh2;
        base[(int64_t)row * a.LCp] = acc[j][i];
      }
    }
  }
#ifdef WS_DBG_STAMPS
  __syncthreads();
  if (tid == 0 && blockIdx.y == 0 && blockIdx.z == 0)  // debug only: overwrites the head of this group's slab
    for (int k = 0; k < 8; k++) a.partials[(int64_t)blockIdx.x * 27 * a.SCp * a.LCp + k] = (float)tacc[k];
#endif
}

// ------------------------------------------------------------------------------------------------
// conv_wgrad_s2: 16-bit STRIDE-2 weight gradient (ConvTranspose3d: small = its input x, large = dy), double-buffered
// through LDS-DMA.  conv_wgrad_kernel<.,4,4,4,2> issued ~620 VALU instructions per tile and wave (per-slot div/mod,
// 64-bit addresses and bounds tests of 13 staging slots) for 28 MFMAs, with one tile of loads in flight and the LDS
// commit serialised with the MFMAs: 9 % matrix-pipe utilisation.  Here
//  * a workgroup owns SB = 2 small-channel blocks: a staged dy box and every B fragment feed two MFMAs (half the
//    L2->LDS bytes and LDS reads per FLOP).  The 2 x 7 x 16 accumulators fill the AGPR file, so nothing is staged through
//    registers: every 16-byte slot is a global_load_lds_dwordx4 (lane-linear LDS image = consecutive 64-byte rows,
//    4 lanes per row), slots outside the tensor / beyond the channel count read a zero line instead;
//  * tile t+1 is in flight while tile t is under the MFMAs (two LDS buffers, one barrier per tile); slot offsets are
//    per-thread constants (32-bit offsets from a per-tile origin);
//  * the InstanceNorm/ReLU transform of x is applied in place by the thread that loaded the chunk (its own 16 bytes
//    are visible to it once vmcnt says so; the small-operand loads are issued first, so they retire first);
//  * the 9x9x9 dy box is stored with each x-line split into even then odd positions: the 4 voxels a transposed
//    read addresses (2 apart in x) are then 4 consecutive 64-byte rows = all 64 banks once, without row padding.
// Tile = 4x4x4 small voxels (whole tiles only: launcher check); wave w owns taps 7w..7w+6 (27 + one dummy).
template <typename T, int SB>
__global__ __launch_bounds__(256) void conv_wgrad_s2_kernel(WgradArgs a) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int MT = 64, BX = 9, BOXL = BX * BX * BX;
  constexpr int LP = 64, CPV = 4, EPC = 8;
  constexpr int NL_ = (BOXL * CPV + 255) / 256, NSLOT = SB + NL_;
  constexpr int SBUF = SB * MT * LP, LBUF = NL_ * 256 * 16, BUF = SBUF + LBUF;  // the last slot's tail lanes land in padding
  constexpr int KS = MT / 16, NT = 7;  // k-steps per tile, taps per wave
  static_assert(NSLOT <= 32, "slot validity bits");
  __shared__ __attribute__((aligned(256))) char lds[2 * BUF + SB * 64 * 4];
  float* const s_xf = reinterpret_cast<float*>(lds + 2 * BUF);  // [SB*32 scale][SB*32 shift] of the small operand

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int scb = blockIdx.y, lcb = blockIdx.z;
  const int part = tid & (CPV - 1), v0 = tid >> 2;
  const int ntz = a.Ds / 4, nty = a.Hs / 4, ntx = a.Ws / 4;
  const bool xfs = a.sm_scale != nullptr;
  const float relu_lo = (xfs && a.sm_relu) ? 0.f : -INFINITY;

  // ---- transposed-read addresses (lane roles of ds_read_b64_tr_b16: 4 voxels x 16 channels per 16 lanes).  Voxel
  // lin = 16 ks + 8 hh + 4 tt + q of the tile is (lz, ly, lx) = (ks, 2 hh + tt, q); k-step and tt are immediates,
  // the buffer parity is folded into the registers (toggled once per tile)
  const int g4 = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3;
  const int hh = g4 >> 1, cb = (g4 & 1) * 16;
  const int colb = (cb + 4 * p4) * 2;
  int sA = (8 * hh + q) * LP + colb;
  int lB[NT];
#pragma unroll
  for (int j = 0; j < NT; j++) {
    int tap = wave * NT + j;
    if (tap >= 27) tap = 0;  // dummy slot of the last wave (never stored)
    const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    const int pb = kx == 0 ? 0 : (kx == 1 ? 5 : 1);  // line position of box x = kx (even x first: 0,2,4,6,8,1,3,5,7)
    lB[j] = SBUF + ((kz * BX + 4 * hh + ky) * BX + pb + q) * LP + colb;
  }
  const int ntaps_here = min(NT, 27 - wave * NT);

  // ---- staging slots (per-thread constants): slot s < SB -> channel block s of small voxel v0, else box row
  // v0 + 64 (s - SB).  Only the LOW faces of a box can leave the tensor (whole tiles, Dl = 2 Ds): emz/emy/emx flag
  // the slots on them; chan_ok the slots whose channels (and box row) exist.
  const int s_goff = ((((v0 >> 4) * a.Hs) + ((v0 >> 2) & 3)) * a.Ws + (v0 & 3)) * (int)a.sm_pitch;
  int goffL[NL_];
  uint32_t emz = 0, emy = 0, emx = 0, chan_ok = 0;
  const bool lc_ok = lcb * 32 + part * EPC < a.LC;
#pragma unroll
  for (int k = 0; k < NL_; k++) {
    const int row = v0 + 64 * k;  // LDS row
    const int rc = min(row, BOXL - 1);
    const int line = rc / BX, pos = rc - line * BX;
    const int bx = pos < 5 ? 2 * pos : 2 * pos - 9, by = line % BX, bz = line / BX;
    goffL[k] = ((bz * a.Hl + by) * a.Wl + bx) * (int)a.lg_pitch;
    emz |= (bz == 0 ? 1u : 0u) << (SB + k);
    emy |= (by == 0 ? 1u : 0u) << (SB + k);
    emx |= (bx == 0 ? 1u : 0u) << (SB + k);
    chan_ok |= ((lc_ok && row < BOXL) ? 1u : 0u) << (SB + k);
  }
#pragma unroll
  for (int b = 0; b < SB; b++) chan_ok |= ((scb * SB + b) * 32 + part * EPC < a.SC ? 1u : 0u) << b;
  const T* const zero_src = reinterpret_cast<const T*>(g_zero_line);
  const T* const s_src = reinterpret_cast<const T*>(a.sm) + scb * SB * 32 + part * EPC;
  const T* const l_src = reinterpret_cast<const T*>(a.lg) + lcb * 32 + part * EPC;

  struct Tl {
    int n, z0, y0, x0;
  };
  auto decode = [&](int t, Tl& c) {
    c.x0 = (t % ntx) * 4;
    t /= ntx;
    c.y0 = (t % nty) * 4;
    t /= nty;
    c.z0 = (t % ntz) * 4;
    c.n = t / ntz;
  };
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  // all slots of one tile, small operand first (its loads retire first: vmcnt(NL_) = "my x chunks have landed")
  auto issue_tile = [&](const Tl& c, int buf_off) __attribute__((always_inline)) {
    const T* const sorg = s_src + ((((int64_t)c.n * a.Ds + c.z0) * a.Hs + c.y0) * a.Ws + c.x0) * a.sm_pitch + s_goff;
    // box origin; may lie before the tensor (those slots read the zero line)
    const T* const lorg =
        l_src + ((((int64_t)c.n * a.Dl + (2 * c.z0 - 1)) * a.Hl + (2 * c.y0 - 1)) * a.Wl + (2 * c.x0 - 1)) * a.lg_pitch;
    const uint32_t off = (c.z0 == 0 ? emz : 0u) | (c.y0 == 0 ? emy : 0u) | (c.x0 == 0 ? emx : 0u);
    const uint32_t m = chan_ok & ~off;
    // wave-uniform LDS byte address of slot 0 (+ lane * 16 by the hardware)
    const uint32_t wbase = __builtin_amdgcn_readfirstlane(lds_base + buf_off + wave * 1024);
#pragma unroll
    for (int s = 0; s < NSLOT; s++) {
      const bool ok = (m >> s) & 1u;
      const T* p = (s < SB) ? sorg + s * 32 : lorg + goffL[s < SB ? 0 : s - SB];
      p = ok ? p : zero_src;
      // inline asm on purpose: hipcc drains a builtin LDS-DMA (vmcnt(0)) in front of the next LDS read of ANY buffer;
      // these are counted by hand (s_waitcnt vmcnt below).  M0 = LDS destination, saved and restored per statement.
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(p), "s"(wbase + (uint32_t)(s < SB ? s * MT * LP : SBUF + (s - SB) * 4096))
                   : "memory");
    }
  };
  int tbl_n = -1;
  auto refresh_xf = [&](int n) {  // uniform
    __syncthreads();              // nobody still reads the previous table
    if (tid < SB * 32) {
      const int c = scb * SB * 32 + tid;
      const bool live = c < a.SC;  // channels past the end stay exactly zero under the transform
      s_xf[tid] = live ? a.sm_scale[(int64_t)n * a.SC + c] : 0.f;
      s_xf[SB * 32 + tid] = live ? a.sm_shift[(int64_t)n * a.SC + c] : 0.f;
    }
    tbl_n = n;
    __syncthreads();
  };
  // x*scale+shift (+relu) on this thread's own chunks of the small tile in buffer buf_off
  auto transform_own = [&](int buf_off) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < SB; b++) {
      u32x4* const slot = reinterpret_cast<u32x4*>(lds + buf_off + b * MT * LP + tid * 16);
      float f[EPC];
      ST<T>::unpack(*slot, f);
      const float* tb = s_xf + b * 32 + part * EPC;
#pragma unroll
      for (int e = 0; e < EPC; e += 4) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(tb + e);
        const f32x4 w = *reinterpret_cast<const f32x4*>(tb + SB * 32 + e);
#pragma unroll
        for (int k = 0; k < 4; k++) f[e + k] = fmaxf(f[e + k] * u[k] + w[k], relu_lo);
      }
      *slot = ST<T>::pack(f);
    }
  };

  f32x16 acc[SB][NT];
#pragma unroll
  for (int b = 0; b < SB; b++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[b][j][i] = 0.f;

  const int t_begin = blockIdx.x * a.tiles_per_group;
  const int t_end = min(a.num_tiles, t_begin + a.tiles_per_group);
  if (t_begin < t_end) {
    using lds_s16x4 = s16x4 __attribute__((address_space(3)));
    auto tr_read = [&](int off) {
      s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off));
      return __builtin_bit_cast(u32x2, v);
    };
    Tl T1;
    decode(t_begin, T1);
    if (xfs) refresh_xf(T1.n);
    issue_tile(T1, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (xfs) transform_own(0);
    WS_BARRIER();
    int wr_off = BUF;  // byte offset of the buffer being filled (the other one is read)
    for (int t = t_begin; t < t_end; t++) {
      const bool v1 = t + 1 < t_end;
      if (v1) {  // uniform
        decode(t + 1, T1);
        issue_tile(T1, wr_off);
      }
      u32x2 A0[2][SB], A1[2][SB], B0[NT], B1[NT];
      auto read_a = [&](int ks) {
#pragma unroll
        for (int b = 0; b < SB; b++) {
          A0[ks & 1][b] = tr_read(sA + b * MT * LP + ks * 16 * LP);
          A1[ks & 1][b] = tr_read(sA + b * MT * LP + ks * 16 * LP + 4 * LP);
        }
      };
      auto read_b = [&](int ks, int j) {
        B0[j] = tr_read(lB[j] + ks * 2 * BX * BX * LP);
        B1[j] = tr_read(lB[j] + ks * 2 * BX * BX * LP + 2 * BX * LP);
      };
      read_a(0);
#pragma unroll
      for (int j = 0; j < NT; j++) read_b(0, j);
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        if (ks + 1 < KS) read_a(ks + 1);
#pragma unroll
        for (int j = 0; j < NT; j++) {
          const u32x4 bf = {B0[j][0], B0[j][1], B1[j][0], B1[j][1]};
#pragma unroll
          for (int b = 0; b < SB; b++) {
            const u32x4 af = {A0[ks & 1][b][0], A0[ks & 1][b][1], A1[ks & 1][b][0], A1[ks & 1][b][1]};
            Mma<T>::run(af, bf, acc[b][j]);
          }
          if (ks + 1 < KS) read_b(ks + 1, j);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (xfs && v1) {
        if (T1.n != tbl_n) refresh_xf(T1.n);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL_) : "memory");  // this thread's x chunks of tile t+1 have landed
        transform_own(wr_off);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WS_BARRIER();  // one buffer fully read, the other fully written
#pragma unroll
      for (int b = 0; b < SB; b++)
#pragma unroll
        for (int j = 0; j < NT; j++) asm volatile("" : "+a"(acc[b][j]));  // loop-carried accumulators stay in AGPRs
      // swap the buffers (the read addresses carry the parity)
      const int d = wr_off ? BUF : -BUF;
      sA += d;
#pragma unroll
      for (int j = 0; j < NT; j++) lB[j] += d;
      wr_off = BUF - wr_off;
    }
  }

  // partial[g][tap][SCp][LCp]
  const int col = lane & 31, hh2 = lane >> 5;
#pragma unroll
  for (int b = 0; b < SB; b++) {
    if ((scb * SB + b) * 32 >= a.SCp) continue;
#pragma unroll
    for (int j = 0; j < NT; j++) {
      if (j < ntaps_here) {
        const int tap = wave * NT + j;
        float* base = a.partials + (((int64_t)blockIdx.x * 27 + tap) * a.SCp + (scb * SB + b) * 32) * a.LCp + lcb * 32 + col;
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int row = (i & 3) + 8 * (i >> 2) + 4 * hh2;
          base[(int64_t)row * a.LCp] = acc[b][j][i];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// conv_gather_s2: 16-bit stride-2 gather conv with 64-byte input rows and 64 output channels -- the data gradient of
// the highest-resolution ConvTranspose3d (dX[v] = sum_tap W[tap]^T dY[2v - 1 + tap], 32 -> 64 channels): persistent
// workgroups, the same LDS-DMA double-buffered 9x9x9 box as conv_wgrad_s2_kernel, and WEIGHTS-STATIONARY REGISTERS.
// The generic kernel it replaces (conv_igemm_kernel<.,2,4,8,..,S=2>) ran one 64-voxel tile per workgroup (8192
// workgroups, each staging its box with per-slot index arithmetic and fetching every weight fragment from L2): 23 VALU
// instructions per MFMA, 63 % of the LDS cycles in bank conflicts (rows 2 apart), 268 TF.  Here
//  * wave (mb, nb) owns 32 of the tile's 64 voxels x 32 of the 64 output channels; its 27 x 2 weight fragments
//    (216 registers) are loaded once per launch;
//  * per tile a wave issues 54 MFMAs and 54 ds_read_b128 (two accumulator chains, A fragments two steps ahead) and
//    its share of the next tile's 12 LDS-DMA slots; nothing else;
//  * box rows keep the even-then-odd x order of conv_wgrad_s2_kernel (stride-2 neighbours = consecutive 64-byte rows),
//    and the 16-byte chunks of a row are XOR-swizzled by (box y >> 1) & 3 on the SOURCE side of the DMA (the LDS image
//    of an LDS-DMA is lane-linear), so the 16 lanes a ds_read_b128 services together (4 x positions x 4 different y)
//    cover all 64 banks.
template <typename T>
__global__ __launch_bounds__(256) void conv_gather_s2_kernel(ConvArgs a) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int BX = 9, BOXL = BX * BX * BX, LP = 64, EPC = 8;
  constexpr int NL_ = (BOXL * 4 + 255) / 256, LBUF = NL_ * 256 * 16;  // the last slot's tail lanes land in padding
  constexpr int NS = 54;                                              // fragment steps per tile: 27 taps x 2
  __shared__ __attribute__((aligned(256))) char lds[2 * LBUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int mb = wave >> 1, nb = wave & 1;
  const int part = tid & 3, v0 = tid >> 2;
  const int ntz = a.Do / 4, nty = a.Ho / 4, ntx = a.Wo / 4;  // whole tiles (launcher check)
  const int num_tiles = a.N * ntz * nty * ntx;
  const int per = (num_tiles + gridDim.x - 1) / gridDim.x;
  const int t_begin = blockIdx.x * per, t_end = min(num_tiles, t_begin + per);
  if (t_begin >= t_end) return;

  // ---- weights -> registers (B operand: lane r = output channel, 8 k values at h)
  u32x4 wf[NS];
  {
    const char* wb = a.wfrag ? reinterpret_cast<const char*>(a.w) + nb * 2 * 1024 + r * 32 + h * 16
                             : reinterpret_cast<const char*>(a.w) + ((int64_t)(nb * 32 + r) * a.Cin) * 2 + h * 16;
    const int fstride = a.wfrag ? 1024 : 32;
    const int64_t wtap_stride = (int64_t)a.CoutP * a.Cin * 2;
#pragma unroll
    for (int s_ = 0; s_ < NS; s_++) wf[s_] = *reinterpret_cast<const u32x4*>(wb + (s_ >> 1) * wtap_stride + (s_ & 1) * fstride);
  }

  // ---- A fragment addresses: M-block row r = (lz & 1) * 16 + ly * 4 + lx, lz = 2 mb + (r >> 4)
  const int a_ly = (r >> 2) & 3;
  const int abase = (((2 * (2 * mb + (r >> 4))) * BX + 2 * a_ly) * BX + (r & 3)) * LP;
  int aoff[2][2];  // [tap y == 2][k-step]: byte offset of this lane's 16-byte chunk inside its row
#pragma unroll
  for (int y2 = 0; y2 < 2; y2++)
#pragma unroll
    for (int ks = 0; ks < 2; ks++) aoff[y2][ks] = abase + (((2 * ks + h) ^ ((a_ly + y2) & 3)) << 4);

  // ---- staging slots (per-thread constants): slot k = box row v0 + 64 k, this lane's chunk = part ^ swizzle(row)
  int goffL[NL_];
  uint32_t emz = 0, emy = 0, emx = 0, row_ok = 0;
#pragma unroll
  for (int k = 0; k < NL_; k++) {
    const int row = v0 + 64 * k;
    const int rc = min(row, BOXL - 1);
    const int line = rc / BX, pos = rc - line * BX;
    const int bx = pos < 5 ? 2 * pos : 2 * pos - 9, by = line % BX, bz = line / BX;
    const int chunk = part ^ ((by >> 1) & 3);
    goffL[k] = ((bz * a.Hi + by) * a.Wi + bx) * (int)a.in_pitch + chunk * EPC;
    emz |= (bz == 0 ? 1u : 0u) << k;
    emy |= (by == 0 ? 1u : 0u) << k;
    emx |= (bx == 0 ? 1u : 0u) << k;
    row_ok |= (row < BOXL ? 1u : 0u) << k;
  }
  const T* const zero_src = reinterpret_cast<const T*>(g_zero_line);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  struct Tl {
    int n, z0, y0, x0;
  };
  auto decode = [&](int t, Tl& c) {
    c.x0 = (t % ntx) * 4;
    t /= ntx;
    c.y0 = (t % nty) * 4;
    t /= nty;
    c.z0 = (t % ntz) * 4;
    c.n = t / ntz;
  };
  auto issue_tile = [&](const Tl& c, int buf_off) __attribute__((always_inline)) {
    // box origin; may lie before the tensor (those slots read the zero line)
    const T* const lorg = reinterpret_cast<const T*>(a.in) +
                          ((((int64_t)c.n * a.Di + (2 * c.z0 - 1)) * a.Hi + (2 * c.y0 - 1)) * a.Wi + (2 * c.x0 - 1)) * a.in_pitch;
    const uint32_t off = (c.z0 == 0 ? emz : 0u) | (c.y0 == 0 ? emy : 0u) | (c.x0 == 0 ? emx : 0u);
    const uint32_t m = row_ok & ~off;
    const uint32_t wbase = __builtin_amdgcn_readfirstlane(lds_base + buf_off + wave * 1024);
#pragma unroll
    for (int k = 0; k < NL_; k++) {
      const T* p = ((m >> k) & 1u) ? lorg + goffL[k] : zero_src;
      uint32_t keep;  // inline asm: see conv_wgrad_s2_kernel
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(p), "s"(wbase + (uint32_t)(k * 4096))
                   : "memory");
    }
  };

  // ---- epilogue constants: accumulator register i = M-block row (i & 3) + 8 (i >> 2) + 4 h
  const int ch = nb * 32 + r;
  const bool ch_ok = ch < a.Cout;
  const float bias = (a.bias && ch_ok) ? a.bias[ch] : 0.f;
  const bool ch_odd = r & 1;
  int eoff[8];  // accumulator rows 2 j and 2 j + 1 leave as one dword per lane (st_rows2)
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int i = 2 * j + (ch_odd ? 1 : 0);
    const int rr = (i & 3) + 8 * (i >> 2) + 4 * h;
    eoff[j] = (((2 * mb + (rr >> 4)) * a.Ho + ((rr >> 2) & 3)) * a.Wo + (rr & 3)) * (int)a.out_pitch - (ch_odd ? 1 : 0);
  }

  Tl T1;
  decode(t_begin, T1);
  issue_tile(T1, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WS_BARRIER();
  int rd_off = 0;
  for (int t = t_begin; t < t_end; t++) {
    const Tl T0 = T1;
    if (t + 1 < t_end) {  // uniform
      decode(t + 1, T1);
      issue_tile(T1, LBUF - rd_off);
    }
    f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[c][i] = 0.f;
    auto rd = [&](int s_) {
      const int tap = s_ >> 1, ks = s_ & 1;
      const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
      const int pb = kx == 0 ? 0 : (kx == 1 ? 5 : 1);
      return *reinterpret_cast<const u32x4*>(lds + rd_off + aoff[ky == 2][ks] + ((kz * BX + ky) * BX + pb) * LP);
    };
    u32x4 af[3];
    af[0] = rd(0);
    af[1] = rd(1);
#pragma unroll
    for (int s_ = 0; s_ < NS; s_++) {
      if (s_ + 2 < NS) af[(s_ + 2) % 3] = rd(s_ + 2);
      Mma<T>::run(af[s_ % 3], wf[s_], acc[s_ & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // The next tile's box must have landed before the barrier.  vmcnt counts stores too and retires in order, so the
    // wait sits BEFORE this tile's stores (behind them it also waited for their write acknowledgements, ~1 us per
    // tile); the stores then drain under the next tile's MFMAs.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // epilogue: one dword per lane and pair of accumulator rows (16 lanes = 64 contiguous bytes of one voxel row)
    T* const obase = reinterpret_cast<T*>(a.out) +
                     ((((int64_t)T0.n * a.Do + T0.z0) * a.Ho + T0.y0) * a.Wo + T0.x0) * a.out_pitch + ch;
    if (ch_ok) {
#pragma unroll
      for (int j = 0; j < 8; j++)
        st_rows2<T>(obase + eoff[j], acc[0][2 * j] + acc[1][2 * j] + bias, acc[0][2 * j + 1] + acc[1][2 * j + 1] + bias, ch_odd);
    }
    WS_BARRIER();  // one buffer fully read, the other fully written
    rd_off = LBUF - rd_off;
  }
}

// ------------------------------------------------------------------------------------------------
// convt_ws: 16-bit ConvTranspose3d(k3,s2,p1,op1) forward for 128-byte input rows and <= 32 output channels (the
// highest-resolution up-convolution, 64 -> 32): persistent workgroups, LDS-DMA double-buffered input box, weights in
// registers.  convt_fused_kernel ran one 256-voxel tile per workgroup: generic per-slot staging, every weight
// fragment fetched from L2 at its point of use by all four waves, 3360 VALU instructions per wave for 216 MFMAs,
// SQ_WAIT_ANY 64 % of the wave cycles (310 TF).  Here
//  * the 8 output-parity classes (1,2,2,2,4,4,4,8 taps) are dealt to the waves as {7}, {6,5}, {3,4,0}, {1,2}: a
//    wave keeps the 16..32 weight fragments of ITS classes in registers for the whole launch and runs them over all
//    four 32-voxel M-blocks of a 4x4x8 tile (8 : 8 : 7 : 4 taps -- the matrix pipe is not the bound here);
//  * the (4+1)x(4+1)x(8+1) input box of tile t+1 lands by LDS-DMA while tile t is under the MFMAs; its
//    InstanceNorm/ReLU transform is applied in place, by the thread that loaded the chunk, between the two halves of
//    the wave's MFMA work (zero padding = slots that read the zero line and are skipped by the transform);
//  * 128-byte rows: the 16-byte chunk c of a box row sits in slot c ^ ((x >> 1) & 1 | (y & 3) << 1) (applied on the
//    source side of the DMA), so the 16 lanes one ds_read_b128 pass services (4 x 4 voxels in x, y) cover the 64 banks.
template <typename T>
__global__ __launch_bounds__(256) void convt_ws_kernel(ConvArgs a) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int BD = 5, BH = 5, BW = 9, BOX = BD * BH * BW, LP = 128, EPC = 8;
  constexpr int NJ = (BOX * 8 + 255) / 256, LBUF = NJ * 256 * 16;  // 8 slots per thread; tail lanes land in padding
  __shared__ __attribute__((aligned(256))) char lds[2 * LBUF + 128 * 4];
  float* const s_xf = reinterpret_cast<float*>(lds + 2 * LBUF);  // [64 scale][64 shift]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ntz = a.Di / 4, nty = a.Hi / 4, ntx = a.Wi / 8;  // whole tiles (launcher check)
  const int num_tiles = a.N * ntz * nty * ntx;
  const int per = (num_tiles + gridDim.x - 1) / gridDim.x;
  const int t_begin = blockIdx.x * per, t_end = min(num_tiles, t_begin + per);
  if (t_begin >= t_end) return;
  const bool xf = a.in_scale != nullptr;
  const float relu_lo = (xf && a.in_relu) ? 0.f : -INFINITY;

  // ---- A fragment addresses: M-block mb = tile z, row r = (ly, lx) = (r >> 3, r & 7)
  const int a_ly = r >> 3, a_lx = r & 7;
  const int abase = (a_ly * BW + a_lx) * LP;
  int aoff[2][2][4];  // [tap offset y][tap offset x][k-step]: byte offset of this lane's 16-byte chunk inside its row
#pragma unroll
  for (int oy = 0; oy < 2; oy++)
#pragma unroll
    for (int ox = 0; ox < 2; ox++) {
      const int gsw = (((a_lx + ox) >> 1) & 1) | (((a_ly + oy) & 3) << 1);
#pragma unroll
      for (int ks = 0; ks < 4; ks++) aoff[oy][ox][ks] = abase + (((2 * ks + h) ^ gsw) << 4);
    }

  // ---- staging slots (per-thread constants): slot k = 16-byte slot tid + 256 k of the lane-linear box image
  int goff[NJ], tboff[NJ];
  uint32_t emz = 0, emy = 0, emx = 0, row_ok = 0;
#pragma unroll
  for (int k = 0; k < NJ; k++) {
    const int q = tid + 256 * k, row = q >> 3, sl = q & 7;
    const int rc = min(row, BOX - 1);
    const int bz = rc / (BH * BW), by = (rc / BW) % BH, bx = rc % BW;
    const int chunk = sl ^ (((bx >> 1) & 1) | ((by & 3) << 1));
    goff[k] = ((bz * a.Hi + by) * a.Wi + bx) * (int)a.in_pitch + chunk * EPC;
    tboff[k] = chunk * EPC;
    emz |= (bz == BD - 1 ? 1u : 0u) << k;
    emy |= (by == BH - 1 ? 1u : 0u) << k;
    emx |= (bx == BW - 1 ? 1u : 0u) << k;
    row_ok |= (row < BOX ? 1u : 0u) << k;
  }
  const T* const zero_src = reinterpret_cast<const T*>(g_zero_line);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  struct Tl {
    int n, z0, y0, x0;
  };
  auto decode = [&](int t, Tl& c) {
    c.x0 = (t % ntx) * 8;
    t /= ntx;
    c.y0 = (t % nty) * 4;
    t /= nty;
    c.z0 = (t % ntz) * 4;
    c.n = t / ntz;
  };
  // only the HIGH faces of a box can leave the tensor (whole tiles)
  auto slots_ok = [&](const Tl& c) -> uint32_t {
    const uint32_t off = (c.z0 + 4 == a.Di ? emz : 0u) | (c.y0 + 4 == a.Hi ? emy : 0u) | (c.x0 + 8 == a.Wi ? emx : 0u);
    return row_ok & ~off;
  };
  auto issue_tile = [&](const Tl& c, uint32_t m, int buf_off) __attribute__((always_inline)) {
    const T* const org = reinterpret_cast<const T*>(a.in) + ((((int64_t)c.n * a.Di + c.z0) * a.Hi + c.y0) * a.Wi + c.x0) * a.in_pitch;
    const uint32_t wbase = __builtin_amdgcn_readfirstlane(lds_base + buf_off + wave * 1024);
#pragma unroll
    for (int k = 0; k < NJ; k++) {
      const T* p = ((m >> k) & 1u) ? org + goff[k] : zero_src;
      uint32_t keep;  // inline asm: see conv_wgrad_s2_kernel
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(p), "s"(wbase + (uint32_t)(k * 4096))
                   : "memory");
    }
  };
  int tbl_n = -1;
  auto refresh_xf = [&](int n) {  // uniform
    __syncthreads();              // nobody still reads the previous table
    if (tid < 64) {
      s_xf[tid] = a.in_scale[(int64_t)n * a.Cin + tid];
      s_xf[64 + tid] = a.in_shift[(int64_t)n * a.Cin + tid];
    }
    tbl_n = n;
    __syncthreads();
  };
  // x*scale+shift (+relu) on this thread's own chunks of the box in buffer buf_off; padding slots stay zero
  auto transform_own = [&](uint32_t m, int buf_off) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NJ; k++) {
      if ((m >> k) & 1u) {
        u32x4* const slot = reinterpret_cast<u32x4*>(lds + buf_off + (tid + 256 * k) * 16);
        float f[EPC];
        ST<T>::unpack(*slot, f);
        const float* tb = s_xf + tboff[k];
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
          const f32x4 u = *reinterpret_cast<const f32x4*>(tb + e);
          const f32x4 w = *reinterpret_cast<const f32x4*>(tb + 64 + e);
#pragma unroll
          for (int q = 0; q < 4; q++) f[e + q] = fmaxf(f[e + q] * u[q] + w[q], relu_lo);
        }
        *slot = ST<T>::pack(f);
      }
    }
  };

  // ---- epilogue constants: accumulator register i = M-block row (i & 3) + 8 (i >> 2) + 4 h = (ly, lx)
  const bool ch_ok = r < a.Cout;
  const float bias = (a.bias && ch_ok) ? a.bias[r] : 0.f;
  const bool ch_odd = r & 1;
  int eoff[8];  // accumulator rows 2 j and 2 j + 1 leave as one dword per lane (st_rows2)
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int i = 2 * j + (ch_odd ? 1 : 0);
    const int rr = (i & 3) + 8 * (i >> 2) + 4 * h;
    eoff[j] = ((2 * (rr >> 3)) * a.Wo + 2 * (rr & 7)) * (int)a.out_pitch - (ch_odd ? 1 : 0);
  }
  const int64_t wtap_stride = (int64_t)a.CoutP * a.Cin * 2;
  const char* const wrow = reinterpret_cast<const char*>(a.w) + ((int64_t)r * a.Cin) * 2 + h * 16;

  // One parity class = (pz, py, px); tap j of it = (jz, jy, jx) in [0, ntap) per axis: box offset (parity ? 1 - j : 0),
  // weight index (parity ? 2 j : 1).  The per-wave code below is straight-line: classes, taps and weight slots are
  // compile-time.
  u32x4 wf[32];
  int rd_off = 0;
  T* obase = nullptr;
  auto load_class_w = [&](auto cls_tag, auto slot0_tag) __attribute__((always_inline)) {
    constexpr int cls = decltype(cls_tag)::value, slot0 = decltype(slot0_tag)::value;
    constexpr int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
    constexpr int ny = py ? 2 : 1, nx = px ? 2 : 1, ntap = (pz ? 2 : 1) * ny * nx;
#pragma unroll
    for (int j = 0; j < ntap; j++) {
      const int jz = j / (ny * nx), jy = (j / nx) % ny, jx = j % nx;
      const int wz = pz ? 2 * jz : 1, wy = py ? 2 * jy : 1, wx = px ? 2 * jx : 1;
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
        wf[slot0 + 4 * j + ks] = *reinterpret_cast<const u32x4*>(wrow + ((wz * 3 + wy) * 3 + wx) * wtap_stride + ks * 32);
    }
  };
  // taps [J0, J1) of class cls into acc (zeroed first when J0 == 0)
  auto run_class = [&](auto cls_tag, auto slot0_tag, auto j0_tag, auto j1_tag, f32x16 (&acc)[4]) __attribute__((always_inline)) {
    constexpr int cls = decltype(cls_tag)::value, slot0 = decltype(slot0_tag)::value;
    constexpr int J0 = decltype(j0_tag)::value, J1 = decltype(j1_tag)::value;
    constexpr int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
    constexpr int ny = py ? 2 : 1, nx = px ? 2 : 1;
    if constexpr (J0 == 0) {
#pragma unroll
      for (int mb = 0; mb < 4; mb++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[mb][i] = 0.f;
    }
    // step s = (tap j, k-step ks); the four A fragments of step s+1 are read under the MFMAs of step s
    u32x4 af[2][4];
    auto rd = [&](int s_) {
      const int j = s_ >> 2, ks = s_ & 3;
      const int jz = j / (ny * nx), jy = (j / nx) % ny, jx = j % nx;
      const int oz = pz ? 1 - jz : 0, oy = py ? 1 - jy : 0, ox = px ? 1 - jx : 0;
#pragma unroll
      for (int mb = 0; mb < 4; mb++)
        af[s_ & 1][mb] = *reinterpret_cast<const u32x4*>(lds + rd_off + aoff[oy][ox][ks] + (((mb + oz) * BH + oy) * BW + ox) * LP);
    };
    rd(4 * J0);
#pragma unroll
    for (int s_ = 4 * J0; s_ < 4 * J1; s_++) {
      if (s_ + 1 < 4 * J1) rd(s_ + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mb = 0; mb < 4; mb++) Mma<T>::run(af[s_ & 1][mb], wf[slot0 + s_], acc[mb]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // stores of a finished class: one dword per lane and pair of accumulator rows (16 lanes = one 64-byte voxel row)
  auto store_class = [&](auto cls_tag, const f32x16 (&acc)[4]) __attribute__((always_inline)) {
    constexpr int cls = decltype(cls_tag)::value;
    constexpr int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
    const int coff = ((pz * a.Ho + py) * a.Wo + px) * (int)a.out_pitch;
    if (ch_ok) {
#pragma unroll
      for (int mb = 0; mb < 4; mb++) {
        T* const ob = obase + coff + (int64_t)(2 * mb) * a.Ho * a.Wo * a.out_pitch;
#pragma unroll
        for (int j = 0; j < 8; j++) st_rows2<T>(ob + eoff[j], acc[mb][2 * j] + bias, acc[mb][2 * j + 1] + bias, ch_odd);
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I4 = std::integral_constant<int, 4>;
  using I8 = std::integral_constant<int, 8>;
  using I16 = std::integral_constant<int, 16>;
  using I24 = std::integral_constant<int, 24>;
#define CLS(c) std::integral_constant<int, c>{}
  if (wave == 0) {
    load_class_w(CLS(7), I0{});
  } else if (wave == 1) {
    load_class_w(CLS(6), I0{});
    load_class_w(CLS(5), I16{});
  } else if (wave == 2) {
    load_class_w(CLS(3), I0{});
    load_class_w(CLS(4), I16{});
    load_class_w(CLS(0), I24{});
  } else {
    load_class_w(CLS(1), I0{});
    load_class_w(CLS(2), I8{});
  }

  Tl T1;
  decode(t_begin, T1);
  uint32_t m1 = slots_ok(T1);
  if (xf) refresh_xf(T1.n);
  issue_tile(T1, m1, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (xf) transform_own(m1, 0);
  WS_BARRIER();
  for (int t = t_begin; t < t_end; t++) {
    const Tl T0 = T1;
    const bool v1 = t + 1 < t_end;
    if (v1) {  // uniform
      decode(t + 1, T1);
      m1 = slots_ok(T1);
      issue_tile(T1, m1, LBUF - rd_off);
      if (xf && T1.n != tbl_n) refresh_xf(T1.n);
    }
    obase = reinterpret_cast<T*>(a.out) + ((((int64_t)T0.n * a.Do + 2 * T0.z0) * a.Ho + 2 * T0.y0) * a.Wo + 2 * T0.x0) * a.out_pitch + r;
    // The next tile's box must have landed (and be transformed) before the barrier.  vmcnt counts stores too and
    // retires in order, so the wait sits after the first half of the wave's MFMAs and BEFORE the phase's first store:
    // behind stores it would also wait for their write acknowledgements.  The stores drain under the MFMAs that follow
    // (this phase's second half, the next phase's first).
    auto mid = [&]() __attribute__((always_inline)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (xf && v1) transform_own(m1, LBUF - rd_off);
    };
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    f32x16 acc[4];
    if (wave == 0) {
      run_class(CLS(7), I0{}, I0{}, I4{}, acc);
      mid();
      run_class(CLS(7), I0{}, I4{}, I8{}, acc);
      store_class(CLS(7), acc);
    } else if (wave == 1) {
      run_class(CLS(6), I0{}, I0{}, I4{}, acc);
      mid();
      store_class(CLS(6), acc);
      run_class(CLS(5), I16{}, I0{}, I4{}, acc);
      store_class(CLS(5), acc);
    } else if (wave == 2) {
      run_class(CLS(3), I0{}, I0{}, I4{}, acc);
      mid();
      store_class(CLS(3), acc);
      run_class(CLS(4), I16{}, I0{}, I2{}, acc);
      store_class(CLS(4), acc);
      run_class(CLS(0), I24{}, I0{}, I1{}, acc);
      store_class(CLS(0), acc);
    } else {
      run_class(CLS(1), I0{}, I0{}, I2{}, acc);
      mid();
      store_class(CLS(1), acc);
      run_class(CLS(2), I8{}, I0{}, I2{}, acc);
      store_class(CLS(2), acc);
    }
    WS_BARRIER();  // one buffer fully read, the other fully written (and transformed)
    rd_off = LBUF - rd_off;
  }
#undef CLS
}

// out[(sc*LC + lc)*27 + tap] (+)= sum_g partial[g][tap][sc][lc].  256 threads = 32 group-lanes x 8 lanes of 4 entries
// (16-byte loads; a workgroup owns 32 consecutive entries = one 128-byte line per slab); fixed summation order
// (bitwise reproducible)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partials, float* __restrict__ dw,
                                                           int G, int SCp, int LCp, int SC, int LC, int accumulate) {
  __shared__ float red[32][36];
  const int e4 = threadIdx.x & 7, gl = threadIdx.x >> 3;
  const int64_t per = (int64_t)27 * SCp * LCp;  // a multiple of 32
  const float* src = partials + (int64_t)blockIdx.x * 32 + e4 * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int g = gl;
  for (; g + 32 < G; g += 64) {
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + (int64_t)g * per);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(src + (int64_t)(g + 32) * per);
    s += a0 + a1;
  }
  if (g < G) s += *reinterpret_cast<const f32x4*>(src + (int64_t)g * per);
  *reinterpret_cast<f32x4*>(&red[gl][e4 * 4]) = s;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int el = threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; k++) t += red[k][el];
    const int64_t idx = (int64_t)blockIdx.x * 32 + el;
    const int lc = idx % LCp;
    const int sc = (idx / LCp) % SCp;
    const int tap = idx / ((int64_t)LCp * SCp);
    if (lc < LC && sc < SC) {
      float* o = dw + ((int64_t)sc * LC + lc) * 27 + tap;
      *o = accumulate ? (*o + t) : t;
    }
  }
}

// The same reduction for FEW slabs and a LARGE matrix (the low-resolution layers: G <= 8, up to 27 x 512 x 512
// entries).  There the cost is the transposed store (4-byte writes 108 bytes apart: 7 M sectors for 512 x 512), not
// the slab reads: a workgroup owns one sc row x 32 lc columns x all 27 taps, sums into LDS and writes the 864 outputs
// as one contiguous run.
__global__ __launch_bounds__(256) void wgrad_reduce_rows_kernel(const float* __restrict__ partials,
                                                                float* __restrict__ dw, int G, int SCp, int LCp, int SC,
                                                                int LC, int accumulate) {
  __shared__ float red[27][33];
  const int sc = blockIdx.y, lc0 = blockIdx.x * 32;
  const int64_t per = (int64_t)27 * SCp * LCp;
  for (int i = threadIdx.x; i < 27 * 32; i += 256) {
    const int tap = i >> 5, l = i & 31;
    const float* src = partials + ((int64_t)tap * SCp + sc) * LCp + lc0 + l;
    float s = 0.f;
    for (int g = 0; g < G; g++) s += src[(int64_t)g * per];
    red[tap][l] = s;
  }
  __syncthreads();
  if (sc >= SC) return;
  float* const out = dw + ((int64_t)sc * LC + lc0) * 27;
  const int n_out = min(32, LC - lc0) * 27;
  for (int j = threadIdx.x; j < n_out; j += 256) {
    const int l = j / 27, tap = j - l * 27;
    out[j] = accumulate ? out[j] + red[tap][l] : red[tap][l];
  }
}

// dst[t][o][i] = src[o*so + i*si + (flip ? 26-t : t)]  (zero for o>=O or i>=I); dst is [27][OP][IP]
template <typename T>
__global__ void pack_w_kernel(const float* __restrict__ src, T* __restrict__ dst, int O, int I, int OP, int IP,
                              int64_t so, int64_t si, int flip) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)27 * OP * IP;
  if (idx >= total) return;
  int i = idx % IP;
  int o = (idx / IP) % OP;
  int t = idx / ((int64_t)IP * OP);
  float v = 0.f;
  if (o < O && i < I) v = src[o * so + i * si + (flip ? 26 - t : t)];
  ST<T>::st(dst + idx, v);
}

struct PackBatch {
  PackJob j[HDF_MAX_PACK_JOBS];
};
// grid (blocks, jobs): job blockIdx.y, grid-stride over its OP*IP (out, in) pairs; a thread reads the pair's 27 taps
// (both source layouts keep them contiguous: 108 bytes) and writes one element of each of the 27 tap planes, where
// consecutive threads are consecutive `in` indices, i.e. coalesced
template <typename T>
__global__ void pack_batch_kernel(PackBatch b, const float* __restrict__ params, char* __restrict__ ws) {
  HDF_LIGHT_PRIO();   // (runs beside the first level-0 conv since round 5: plan.hip forward3d)
  const PackJob& jb = b.j[blockIdx.y];
  const float* src = params + jb.src_off;
  T* dst = reinterpret_cast<T*>(ws + jb.dst_off);
  const int64_t pairs = (int64_t)jb.OP * jb.IP;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < pairs; idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = idx % jb.IP, o = idx / jb.IP;
    const bool live = o < jb.O && i < jb.I;
    const float* sp = src + (live ? (int64_t)o * jb.so + (int64_t)i * jb.si : 0);
    float v[27];
#pragma unroll
    for (int t = 0; t < 27; t++) v[t] = sp[t];
    constexpr int E32 = 32 / (int)sizeof(T);  // elements per 32-byte fragment step
    const int64_t at = jb.frag ? (((int64_t)(o >> 5) * (jb.IP / E32) + i / E32) * 32 + (o & 31)) * E32 + i % E32 : idx;
#pragma unroll
    for (int t = 0; t < 27; t++) ST<T>::st(dst + (int64_t)t * pairs + at, live ? (jb.flip ? v[26 - t] : v[t]) : 0.f);
  }
}

// Second pass of a split-K conv launch: one workgroup per (TILE of the conv's tiling, 64-channel block) -- so that the
// InstanceNorm partial rows keep their geometry -- thread = (voxel lane, 4 consecutive channels): the partial tiles are
// summed in split order, + bias (+ the old output when accumulating), stored in the storage type, and the per-channel
// (sum, sum of squares) of the tile reduced over the 16 voxel lanes in a fixed order.
template <typename T, int TD, int TH, int TW>
__global__ __launch_bounds__(256) void conv_ksplit_reduce_kernel(ConvArgs a) {
  if (a.prio) HDF_LIGHT_PRIO();
  __shared__ float red[16][64][2];
  const int ntz = (a.Do + TD - 1) / TD, nty = (a.Ho + TH - 1) / TH, ntx = (a.Wo + TW - 1) / TW;
  int t = blockIdx.x;
  const int tx = t % ntx;
  t /= ntx;
  const int ty = t % nty;
  t /= nty;
  const int tz = t % ntz, n = t / ntz;
  const int cl = ((int)threadIdx.x & 15) * 4, c0 = blockIdx.y * 64 + cl, vl = (int)threadIdx.x >> 4;
  const bool c_on = c0 < a.CoutP;
  const int64_t mtot = (int64_t)a.N * a.Do * a.Ho * a.Wo;
  float bias[4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; e++) bias[e] = (a.bias && c0 + e < a.Cout) ? a.bias[c0 + e] : 0.f;
  constexpr int TV = TD * TH * TW, IT = (TV + 15) / 16;
  static_assert(TV % 16 == 0, "tile voxels per voxel lane");
  if (c_on) {
#pragma unroll 2
    for (int it = 0; it < IT; it++) {
      const int lin = vl + 16 * it;
      const int gz = tz * TD + lin / (TH * TW), gy = ty * TH + (lin / TW) % TH, gx = tx * TW + lin % TW;
      const bool ok = gz < a.Do && gy < a.Ho && gx < a.Wo;
      const int64_t vox = ok ? (((int64_t)n * a.Do + gz) * a.Ho + gy) * a.Wo + gx : 0;
      f32x4 pv[4];
#pragma unroll
      for (int k = 0; k < 4; k++)   // (ksplit <= 4; clamped: never branch around a load)
        pv[k] = *reinterpret_cast<const f32x4*>(a.kpart + ((int64_t)min(k, a.ksplit - 1) * mtot + vox) * a.CoutP + c0);
      f32x4 v = pv[0];
      for (int k = 1; k < a.ksplit; k++) v += pv[k];
      if (ok) {
        T* p = reinterpret_cast<T*>(a.out) + vox * a.out_pitch + c0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          if (c0 + e < a.Cout) {
            float o = v[e] + bias[e];
            if (a.accumulate) o += ST<T>::ld(p + e);
            ST<T>::st(p + e, o);
            s1[e] += o;
            s2[e] += o * o;
          }
        }
      }
    }
  }
  if (a.stat_partials) {
#pragma unroll
    for (int e = 0; e < 4; e++) red[vl][cl + e][0] = s1[e], red[vl][cl + e][1] = s2[e];
    __syncthreads();
    if (threadIdx.x < 128) {
      const int c = threadIdx.x >> 1, j = threadIdx.x & 1;
      if (blockIdx.y * 64 + c < a.CoutP) {
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) sum += red[k][c][j];
        a.stat_partials[((int64_t)blockIdx.x * a.CoutP + blockIdx.y * 64 + c) * 2 + j] = sum;
      }
    }
  }
}

template <typename T, int TD, int TH, int TW, int WM, int WN, int MB, int S, bool CONVT, bool FLAT = false>
int launch_cfg(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  const int Td = CONVT ? a.Di : a.Do, Th = CONVT ? a.Hi : a.Ho, Tw = CONVT ? a.Wi : a.Wo;
  const int tiles = a.N * ceil_div(Td, TD) * ceil_div(Th, TH) * ceil_div(Tw, TW), nblk = ceil_div(a.CoutP, WN * 32);
  a.ksplit = 1;
  if constexpr (!CONVT && S == 1) {
    // split-K: a low-resolution layer whose tiles x output blocks leave CUs idle, whole 64-byte chunks (the pipelined
    // path), one plain output; 4 or 2 ways while every workgroup keeps >= 2 chunks and the partial tiles fit the scratch
    const int nchunk = (a.Cin * (int)sizeof(T)) % 64 == 0 ? a.Cin * (int)sizeof(T) / 64 : 0;
    if (a.kpart && !a.split && a.CoutP <= 1024 && a.CoutP % 4 == 0 && a.out_pitch % 4 == 0) {
      for (int ks = 4; ks >= 2; ks >>= 1) {
        const size_t need = (size_t)ks * a.N * a.Do * a.Ho * a.Wo * a.CoutP * sizeof(float);
        if (tiles * nblk * ks <= hdf_cu_budget() && nchunk % ks == 0 && nchunk / ks >= 2 && need <= a.kpart_bytes) {
          a.ksplit = ks;
          break;
        }
      }
    }
  }
  dim3 grid(tiles, nblk, CONVT ? (FLAT ? 4 : 8) : a.ksplit);   // transposed conv: one grid slice per output-parity class
  hipLaunchKernelGGL((conv_igemm_kernel<T, TD, TH, TW, WM, WN, MB, S, CONVT, FLAT>), grid, dim3(256), 0, st, a);
  HDF_LAUNCH_CHECK();
  if constexpr (!CONVT && S == 1) {
    if (a.ksplit > 1) {
      hipLaunchKernelGGL((conv_ksplit_reduce_kernel<T, TD, TH, TW>), dim3(tiles, ceil_div(a.CoutP, 64)), dim3(256), 0, st, a);
      HDF_LAUNCH_CHECK();
    }
  }
  return HDF_OK;
}

// tile-shape choice: shared by the launcher and by hdf_conv_stat_tiles (partials geometry)
inline bool small_tile(int Do, int Ho, int Wo) {
  return (int64_t)Do * Ho * Wo <= 32 * 32 * 32 / 2;
}
// 8^3-class volumes (the bottleneck UpConv): 64-voxel tiles, or a handful of workgroups would carry the whole layer
inline bool tiny_tile(int Do, int Ho, int Wo) {
  return (int64_t)Do * Ho * Wo <= 8 * 8 * 16;
}
// weights-stationary kernel: mode 0, whole Cin row <= 128 B, enough tiles to amortise the weight panel
inline int ws_cfg(int mode, int Do, int Ho, int Wo, int row_bytes) {
  if (mode != 0 || row_bytes > 128 || (int64_t)Do * Ho * Wo < 48 * 48 * 48 || Do == 1) return 0;   // (depth 1: the flat kernels)
  return (row_bytes == 32 || row_bytes == 64 || row_bytes == 128) ? 1 : 0;
}

// transposed conv: row widths convt_fused_kernel is instantiated for
inline bool convt_fused_rows(int rb) { return rb == 32 || rb == 64 || rb == 128 || rb == 256 || rb == 512; }

template <typename T, int CH, int RB, int NB>
int launch_ws2(const ConvArgs& a, hipStream_t st) {
  const int budget = a.cu_budget > 0 ? std::min(a.cu_budget, hdf_cu_budget()) : hdf_cu_budget();
  const int cout_tiles = a.CoutP / (32 * NB);
#ifndef HDF_NO_WS2_TD8   // (A/B builds: the 4-deep tile everywhere)
  // 64-byte rows, one output block, 16-bit storage, WITH an input transform (32 -> 32 channels at the top level, forward:
  // block_1_2_left / block_1_2_right): the 8x8x8 tile (TDP = 8, see the kernel) when the depth is a whole number of tiles.
  // Measured (tools/td8_ab.sh, same box, 32 -> 32 @128^3 batch 2): with the transform 303 vs 328 us; without 297 vs 300;
  // data gradient with the statistics epilogue 351 vs 356 -- but the 8-deep form needs all 512 registers (384 spill 203),
  // and a 512-register workgroup shares its compute unit with nothing: with all four launches of a step on it the step
  // was 0.1 ms SLOWER (2 of 3 pairs).  So only the two forward launches take it: they run with the other streams idle or
  // on their own quarter of the chip (forward3d), and they are where the gain is.
  if constexpr (sizeof(T) == 2 && NB == 1 && RB == 64) {
    if (a.in_scale && !a.bs_y && a.Do % 8 == 0 && a.Do >= 24 && a.Ho >= 24 && a.Wo >= 24) {
      const int tiles8 = a.N * (a.Do / 8) * ceil_div(a.Ho, 8) * ceil_div(a.Wo, 8);
      const int gx8 = std::min(tiles8, std::max(1, budget / cout_tiles));
      hipLaunchKernelGGL((conv_ws2_kernel<T, CH, RB, true, NB, false, 8>), dim3(gx8, cout_tiles), dim3(256), 0, st, a);
      HDF_LAUNCH_CHECK();
      return HDF_OK;
    }
  }
#endif
  const int tiles = a.N * ceil_div(a.Do, 4) * ceil_div(a.Ho, 8) * ceil_div(a.Wo, 8);
  const int gx = std::min(tiles, std::max(1, budget / cout_tiles));  // one workgroup per CU; the kernel splits the tiles
  if constexpr (sizeof(T) == 2 && NB == 1 && RB == 64) {   // the level-0 32 -> 32 data gradient (hdf_conv_bwd_stats_ok)
    if (a.bs_y && !a.in_scale) {
      hipLaunchKernelGGL((conv_ws2_kernel<T, CH, RB, false, NB, true>), dim3(gx, cout_tiles), dim3(256), 0, st, a);
      HDF_LAUNCH_CHECK();
      return HDF_OK;
    }
  }
  HDF_CHECK_ARG(a.bs_y == nullptr, "conv: backward statistics are not available for this launch (hdf_conv_bwd_stats_ok)");
  if (a.in_scale)
    hipLaunchKernelGGL((conv_ws2_kernel<T, CH, RB, true, NB>), dim3(gx, cout_tiles), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_ws2_kernel<T, CH, RB, false, NB>), dim3(gx, cout_tiles), dim3(256), 0, st, a);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

// Tile of a flat (depth-1, 2-D) stride-1 conv: 1x16x16 (256 voxels), 1x8x16 below 64x64, 1x8x8 below 32x32
inline int flat_tile(int Ho, int Wo) { return (int64_t)Ho * Wo <= 32 * 32 ? 0 : ((int64_t)Ho * Wo <= 64 * 64 ? 1 : 2); }

template <typename T>
int launch_conv_t(int mode, const ConvArgs& a, hipStream_t st) {
  if (a.Di == 1) {
    // ---- the 2-D operators (models/HDenseFormer_2D.py:125-170): depth-1 tensors, centre-plane taps, y / x strides only
    HDF_CHECK_ARG(a.Do == 1 && !a.bs_y, "conv: a depth-1 input selects the 2-D operator (output depth 1, no statistics epilogue of the 3-D data gradient)");
    if (mode == 0) {
      const int ft = flat_tile(a.Ho, a.Wo);
      if (ft == 0) return launch_cfg<T, 1, 8, 8, 2, 2, 1, 1, false, true>(a, st);        // 64 vox x 64 ch
      if (ft == 1) return launch_cfg<T, 1, 8, 16, 2, 2, 2, 1, false, true>(a, st);       // 128 vox x 64 ch
      if (a.CoutP <= 32) return launch_cfg<T, 1, 16, 16, 4, 1, 2, 1, false, true>(a, st);  // 256 vox x 32 ch
      return launch_cfg<T, 1, 16, 16, 2, 2, 4, 1, false, true>(a, st);                   // 256 vox x 64 ch
    } else if (mode == 1) {
      if (a.CoutP <= 64) return launch_cfg<T, 1, 8, 8, 2, 2, 1, 2, false, true>(a, st);   // 64 vox x 64 ch, stride 2 in y, x
      return launch_cfg<T, 1, 8, 8, 1, 4, 2, 2, false, true>(a, st);                      // 64 vox x 128 ch
    } else {
#ifndef HDF_NO_CONVT_FUSED_FLAT   // (A/B builds)
      // all four parity classes in one workgroup, the box staged once, the weight fragments through a register ring
      const int rbt = a.Cin * (int)sizeof(T);
      if (convt_fused_rows(rbt) && !a.accumulate && !a.wfrag) {
        dim3 grid(a.N * ceil_div(a.Hi, 16) * ceil_div(a.Wi, 16), a.CoutP / 32);
        const int nfs = rbt / 32;
        if (nfs == 16)
          hipLaunchKernelGGL((convt_fused_kernel<T, 1, 16, 16, 2, 16, true>), grid, dim3(256), 0, st, a);
        else if (nfs == 8)
          hipLaunchKernelGGL((convt_fused_kernel<T, 1, 16, 16, 2, 8, true>), grid, dim3(256), 0, st, a);
        else if (nfs == 4)
          hipLaunchKernelGGL((convt_fused_kernel<T, 1, 16, 16, 2, 4, true>), grid, dim3(256), 0, st, a);
        else if (nfs == 2)
          hipLaunchKernelGGL((convt_fused_kernel<T, 1, 16, 16, 2, 2, true>), grid, dim3(256), 0, st, a);
        else
          hipLaunchKernelGGL((convt_fused_kernel<T, 1, 16, 16, 2, 1, true>), grid, dim3(256), 0, st, a);
        HDF_LAUNCH_CHECK();
        return HDF_OK;
      }
#endif
      if (a.CoutP <= 32) return launch_cfg<T, 1, 16, 16, 4, 1, 2, 1, true, true>(a, st);
      return launch_cfg<T, 1, 8, 16, 2, 2, 2, 1, true, true>(a, st);
    }
  }
  if (mode == 0) {
    const int rb = a.Cin * (int)sizeof(T);
    const int ws = ws_cfg(mode, a.Do, a.Ho, a.Wo, rb);
    if (ws) {
      if (rb == 32) return launch_ws2<T, 32, 32, 1>(a, st);
      if (rb == 64 && a.CoutP % 64 == 0) return launch_ws2<T, 32, 64, 2>(a, st);
      // 64-byte rows, one output block: two 32-byte passes per tile.  (Rounds 1-2 ran these layers as ONE 64-byte pass
      // with the deferred epilogue: 512 registers, 500 B of scratch per lane in the border copies, 1.2x the output in
      // HBM writes; the two-pass form needs 303 registers and is 7 % faster without / 1 % with an input transform.)
      if (rb == 64) return launch_ws2<T, 32, 64, 1>(a, st);
      if (rb == 128) return launch_ws2<T, 32, 128, 1>(a, st);
    }
    if (tiny_tile(a.Do, a.Ho, a.Wo)) return launch_cfg<T, 2, 4, 8, 2, 2, 1, 1, false>(a, st);   // 64 vox x 64 ch
    if (small_tile(a.Do, a.Ho, a.Wo)) return launch_cfg<T, 4, 4, 8, 2, 2, 2, 1, false>(a, st);  // 128 vox x 64 ch
    if (a.CoutP <= 32) return launch_cfg<T, 4, 8, 8, 4, 1, 2, 1, false>(a, st);                  // 256 vox x 32 ch
    return launch_cfg<T, 4, 8, 8, 2, 2, 4, 1, false>(a, st);                                     // 256 vox x 64 ch
  } else if (mode == 1) {
    if constexpr (sizeof(T) == 2) {
      // (st_rows2: one dword per lane pair = channels (c, c + 1): even channel count / pitch, 4-byte aligned view)
      if (a.Cin * 2 == 64 && a.CoutP == 64 && a.Cout % 2 == 0 && a.out_pitch % 2 == 0 &&
          (reinterpret_cast<uintptr_t>(a.out) & 3) == 0 && !a.in_scale && !a.accumulate && a.Do % 4 == 0 && a.Ho % 4 == 0 &&
          a.Wo % 4 == 0 && a.Di == 2 * a.Do && a.Hi == 2 * a.Ho && a.Wi == 2 * a.Wo) {
        const int tiles = a.N * (a.Do / 4) * (a.Ho / 4) * (a.Wo / 4);
        hipLaunchKernelGGL((conv_gather_s2_kernel<T>), dim3(std::min(tiles, hdf_cu_budget())), dim3(256), 0, st, a);
        HDF_LAUNCH_CHECK();
        return HDF_OK;
      }
    }
    if (a.CoutP <= 64) return launch_cfg<T, 2, 4, 8, 2, 2, 1, 2, false>(a, st);  // 64 vox x 64 ch, stride 2
    return launch_cfg<T, 4, 4, 4, 1, 4, 2, 2, false>(a, st);  // 64 vox x 128 ch, stride 2
  } else {
    if constexpr (sizeof(T) == 2) {
      if (a.Cin * 2 == 128 && a.CoutP == 32 && a.Cout % 2 == 0 && a.out_pitch % 2 == 0 &&
          (reinterpret_cast<uintptr_t>(a.out) & 3) == 0 && !a.accumulate && a.Di % 4 == 0 && a.Hi % 4 == 0 && a.Wi % 8 == 0 &&
          a.Do == 2 * a.Di && a.Ho == 2 * a.Hi && a.Wo == 2 * a.Wi) {
        const int tiles = a.N * (a.Di / 4) * (a.Hi / 4) * (a.Wi / 8);
        hipLaunchKernelGGL((convt_ws_kernel<T>), dim3(std::min(tiles, hdf_cu_budget())), dim3(256), 0, st, a);
        HDF_LAUNCH_CHECK();
        return HDF_OK;
      }
    }
    // all 8 parity classes in one workgroup: rows of 32, 64, 128 or 256 bytes (n_filters = 48: 192-byte rows run per class)
    if (convt_fused_rows(a.Cin * (int)sizeof(T)) && !a.accumulate) {
      dim3 grid(a.N * ceil_div(a.Di, 4) * ceil_div(a.Hi, 8) * ceil_div(a.Wi, 8), a.CoutP / 32);
      const int nfs = a.Cin * (int)sizeof(T) / 32;
      if (nfs == 16) {   // 512-byte rows (16-bit upconv_1: 256 -> 128 channels at the bottom of the decoder): 2-deep tiles, a 128 KB box
        dim3 grid2(a.N * ceil_div(a.Di, 2) * ceil_div(a.Hi, 8) * ceil_div(a.Wi, 8), a.CoutP / 32);
        hipLaunchKernelGGL((convt_fused_kernel<T, 2, 8, 8, 1, 16>), grid2, dim3(256), 0, st, a);
      } else if (nfs == 8)
        hipLaunchKernelGGL((convt_fused_kernel<T, 4, 8, 8, 2, 8>), grid, dim3(256), 0, st, a);
      else if (nfs == 4)
        hipLaunchKernelGGL((convt_fused_kernel<T, 4, 8, 8, 2, 4>), grid, dim3(256), 0, st, a);
      else if (nfs == 2)
        hipLaunchKernelGGL((convt_fused_kernel<T, 4, 8, 8, 2, 2>), grid, dim3(256), 0, st, a);
      else
        hipLaunchKernelGGL((convt_fused_kernel<T, 4, 8, 8, 2, 1>), grid, dim3(256), 0, st, a);
      HDF_LAUNCH_CHECK();
      return HDF_OK;
    }
    if (a.CoutP <= 32) return launch_cfg<T, 4, 8, 8, 4, 1, 2, 1, true>(a, st);
    return launch_cfg<T, 4, 4, 8, 2, 2, 2, 1, true>(a, st);
  }
}

template <typename T, int TD, int TH, int TW, int S, bool FLAT = false>
int launch_wgrad_t(WgradArgs a, float* dw, int sc_store, int lc_store, int accumulate, void* ws, size_t ws_bytes,
                   hipStream_t st) {
  a.SCp = round_up(a.SC, 32);
  a.LCp = round_up(a.LC, 32);
  a.num_tiles = a.N * ceil_div(a.Ds, TD) * ceil_div(a.Hs, TH) * ceil_div(a.Ws, TW);
  // 16-bit stride 2 without a transform of the large operand: conv_wgrad_s2_kernel, SB small-channel blocks per workgroup
  const bool use_s2 = !FLAT && sizeof(T) == 2 && S == 2 && !a.lg_scale && a.Ds % 4 == 0 && a.Hs % 4 == 0 &&
                      a.Ws % 4 == 0 && a.Dl == 2 * a.Ds && a.Hl == 2 * a.Hs && a.Wl == 2 * a.Ws;
  const int sb = (use_s2 && a.SCp >= 64) ? 2 : 1;
  const int sblocks = ceil_div(a.SCp / 32, sb);
  const int pairs = sblocks * (a.LCp / 32);
  const int64_t per = (int64_t)27 * a.SCp * a.LCp * sizeof(float);
  const bool use_new = !FLAT && sizeof(T) == 2 && S == 1 && !a.sm_scale;
  const int wg_target = hdf_cu_budget();  // one workgroup per CU
  int G = ceil_div(wg_target, pairs);
  G = (int)std::min<int64_t>(G, std::max<int64_t>(1, (int64_t)ws_bytes / per));
  G = std::min(G, a.num_tiles);
  a.tiles_per_group = ceil_div(a.num_tiles, G);
  G = ceil_div(a.num_tiles, a.tiles_per_group);
  HDF_CHECK_ARG((size_t)(G * per) <= ws_bytes, "wgrad workspace too small: need %lld have %zu", (long long)(G * per),
                ws_bytes);
  a.partials = reinterpret_cast<float*>(ws);
  dim3 grid(G, sblocks, a.LCp / 32);
  if constexpr (sizeof(T) == 2 && S == 2) {
    if (use_s2) {
      if (sb == 2)
        hipLaunchKernelGGL((conv_wgrad_s2_kernel<T, 2>), grid, dim3(256), 0, st, a);
      else
        hipLaunchKernelGGL((conv_wgrad_s2_kernel<T, 1>), grid, dim3(256), 0, st, a);
    }
  }
  if constexpr (sizeof(T) == 2 && S == 1) {
    if (use_new) {
      if (a.ap_y) {
        if (a.lg_scale)
          hipLaunchKernelGGL((conv_wgrad2_kernel<T, true, true>), grid, dim3(256), 0, st, a);
        else
          hipLaunchKernelGGL((conv_wgrad2_kernel<T, false, true>), grid, dim3(256), 0, st, a);
      } else if (a.lg_scale)
        hipLaunchKernelGGL((conv_wgrad2_kernel<T, true>), grid, dim3(256), 0, st, a);
      else
        hipLaunchKernelGGL((conv_wgrad2_kernel<T, false>), grid, dim3(256), 0, st, a);
    }
  }
  if (!use_new && !use_s2) {
    hipLaunchKernelGGL((conv_wgrad_kernel<T, TD, TH, TW, S, FLAT>), grid, dim3(256), 0, st, a);
  }
  HDF_LAUNCH_CHECK();
  int64_t n = (int64_t)27 * a.SCp * a.LCp;
  if (G <= 8)
    hipLaunchKernelGGL(wgrad_reduce_rows_kernel, dim3(a.LCp / 32, a.SCp), dim3(256), 0, st, a.partials, dw, G, a.SCp,
                       a.LCp, sc_store, lc_store, accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div64(n, 32)), dim3(256), 0, st, a.partials, dw, G,
                       a.SCp, a.LCp, sc_store, lc_store, accumulate);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

}  // namespace

int hdf_conv_weight_layout(int dtype, int mode, int Cin, int Do, int Ho, int Wo) {
  const int rb = Cin * hdf_esz(dtype);
  if (rb % 64 != 0) return 0;                       // the pipelined path needs whole 64-byte chunks
  if (Do == 1)                                      // depth 1 (the 2-D operators): conv_igemm_kernel, but the transposed conv
    return (mode == 2 && convt_fused_rows(rb)) ? 0 : 1;   // runs convt_fused_kernel<FLAT> (row-major panels)
  if (mode == 0) return ws_cfg(mode, Do, Ho, Wo, rb) ? 0 : 1;  // conv_ws2_kernel stages row-major panels
  if (mode == 1) return 1;                                 // stride-2 gather conv: pipelined path
  return convt_fused_rows(rb) ? 0 : 1;  // transposed conv: convt_fused_kernel reads row-major panels, other widths run per class
}

bool hdf_conv_bwd_stats_ok(int dtype, const ConvArgs& a) {
  const int rb = a.Cin * hdf_esz(dtype);
  return dtype != HDF_F32 && rb == 64 && a.Cout == 32 && a.CoutP == 32 && !a.in_scale && !a.accumulate && !a.split &&
         a.Do % 4 == 0 && a.Ho % 8 == 0 && a.Wo % 8 == 0 && ws_cfg(0, a.Do, a.Ho, a.Wo, rb) && a.stat_partials &&
#ifndef HDF_NO_CONV_WR
         !hdf_conv_wr_takes(dtype, a) &&
#endif
         a.Di == a.Do && a.Hi == a.Ho && a.Wi == a.Wo;
}

int hdf_conv_stat_tiles(int mode, int Do, int Ho, int Wo, int row_bytes) {
  if (mode != 0) return 0;
  if (Do == 1) {   // the flat tiles of launch_conv_t
    const int ft = flat_tile(Ho, Wo);
    return ft == 0 ? ceil_div(Ho, 8) * ceil_div(Wo, 8) : (ft == 1 ? ceil_div(Ho, 8) * ceil_div(Wo, 16) : ceil_div(Ho, 16) * ceil_div(Wo, 16));
  }
  if (ws_cfg(mode, Do, Ho, Wo, row_bytes)) return WS_STAT_ROWS;  // per-workgroup rows (conv_ws2_kernel)
  if (tiny_tile(Do, Ho, Wo)) return ceil_div(Do, 2) * ceil_div(Ho, 4) * ceil_div(Wo, 8);
  if (small_tile(Do, Ho, Wo)) return ceil_div(Do, 4) * ceil_div(Ho, 4) * ceil_div(Wo, 8);
  return ceil_div(Do, 4) * ceil_div(Ho, 8) * ceil_div(Wo, 8);
}

int hdf_launch_conv(int dtype, int mode, const ConvArgs& a, hipStream_t st) {
  HDF_CHECK_ARG(a.Cin % 16 == 0, "conv: Cin=%d must be a multiple of 16 (pad the first layer)", a.Cin);
  HDF_CHECK_ARG(a.CoutP % 32 == 0 && a.CoutP >= a.Cout, "conv: bad CoutP=%d for Cout=%d", a.CoutP, a.Cout);
  HDF_CHECK_ARG(a.in_pitch % 8 == 0 && (((uintptr_t)a.in) & 15) == 0, "conv: input view must be 16-byte aligned");
  HDF_CHECK_ARG(mode == 0 || a.stat_partials == nullptr, "conv: stats only in mode 0");
  HDF_CHECK_ARG(a.wfrag == 0 || hdf_conv_weight_layout(dtype, mode, a.Cin, a.Do, a.Ho, a.Wo) == 1,
                "conv: fragment-major weights given to a launch that reads row-major panels (mode %d Cin %d)", mode, a.Cin);
  HDF_CHECK_ARG(a.split == 0 || (mode == 0 && a.split % 32 == 0 && a.split < a.Cout && a.out2 != nullptr),
                "conv: split output needs mode 0, split %% 32 == 0 and a second buffer (split=%d)", a.split);
#ifndef HDF_NO_CONV_WR  // (A/B builds: the weights-stationary kernels for every launch)
  if (mode == 0 && hdf_conv_wr_takes(dtype, a)) return hdf_launch_conv_wr(dtype, a, st);
#endif
  HDF_DISPATCH_T(dtype, return launch_conv_t<T>(mode, a, st));
  return HDF_ERR_UNSUPPORTED;
}

size_t hdf_wgrad_workspace_bytes(int stride, int N, int Ds, int Hs, int Ws, int SC, int LC) {
  int SCp = round_up(SC, 32), LCp = round_up(LC, 32);
  int64_t per = (int64_t)27 * SCp * LCp * sizeof(float);
  int pairs = (SCp / 32) * (LCp / 32);
  int tiles = stride == 1 ? N * ceil_div(Ds, 4) * ceil_div(Hs, 8) * ceil_div(Ws, 8)
                          : N * ceil_div(Ds, 4) * ceil_div(Hs, 4) * ceil_div(Ws, 4);
  if (Ds == 1) tiles = stride == 1 ? N * ceil_div(Hs, 16) * ceil_div(Ws, 16) : N * ceil_div(Hs, 8) * ceil_div(Ws, 8);   // flat tiles
  int G = std::min(ceil_div(1024, pairs), tiles);
  int64_t bytes = std::min<int64_t>((int64_t)G * per, std::max<int64_t>(per, (int64_t)96 << 20));
  return (size_t)bytes;
}

// the launches conv_wgrad2_kernel<., ., true> serves: 16-bit storage, stride 1, untransformed small operand, 16-byte rows
// on both extra tensors, and 32-bit byte offsets into them with bit 31 free for the out-of-range marker
bool hdf_wgrad_apply_takes(int dtype, int stride, const WgradArgs& a) {
  if (hdf_esz(dtype) != 2 || stride != 1 || a.sm_scale || !a.ap_y || !a.ap_out || a.Ds == 1) return false;
  for (int k = 0; k < 7; k++)
    if (!a.ap_tab[k]) return false;
  const int64_t vox = (int64_t)a.N * a.Ds * a.Hs * a.Ws;
  if (vox * a.ap_y_pitch * 2 >= (1ll << 31) || vox * a.ap_out_pitch * 2 >= (1ll << 31)) return false;
  if (a.ap_y_pitch % 8 || a.ap_out_pitch % 8 || a.SC % 16) return false;
  if ((reinterpret_cast<uintptr_t>(a.ap_y) | reinterpret_cast<uintptr_t>(a.ap_out)) & 15) return false;
  return a.Ds == a.Dl && a.Hs == a.Hl && a.Ws == a.Wl;
}

int hdf_launch_wgrad(int dtype, int stride, WgradArgs a, float* dw, int sc_store, int lc_store, int accumulate,
                     void* workspace, size_t workspace_bytes, hipStream_t st) {
  HDF_CHECK_ARG(a.SC % 16 == 0 && a.LC % 16 == 0, "wgrad: channel counts must be multiples of 16 (SC=%d LC=%d)", a.SC,
                a.LC);
  HDF_CHECK_ARG(stride == 1 || stride == 2, "wgrad: stride %d", stride);
  HDF_CHECK_ARG(!a.ap_y || hdf_wgrad_apply_takes(dtype, stride, a),
                "wgrad: the fused InstanceNorm backward does not take this launch (ask hdf_wgrad_apply_takes first)");
  HDF_DISPATCH_T(dtype, {
    if (a.Ds == 1) {   // depth-1 operands: the 2-D weight gradients (Conv2d / ConvTranspose2d, models/HDenseFormer_2D.py)
      HDF_CHECK_ARG(a.Dl == 1 && !a.ap_y, "wgrad: depth-1 small operand selects the 2-D operator (large operand of depth 1, no fused InstanceNorm backward)");
      if (stride == 1)
        return launch_wgrad_t<T, 1, 16, 16, 1, true>(a, dw, sc_store, lc_store, accumulate, workspace, workspace_bytes, st);
      return launch_wgrad_t<T, 1, 8, 8, 2, true>(a, dw, sc_store, lc_store, accumulate, workspace, workspace_bytes, st);
    }
    if (stride == 1)
      return launch_wgrad_t<T, 4, 8, 8, 1>(a, dw, sc_store, lc_store, accumulate, workspace, workspace_bytes, st);
    return launch_wgrad_t<T, 4, 4, 4, 2>(a, dw, sc_store, lc_store, accumulate, workspace, workspace_bytes, st);
  });
  return HDF_ERR_UNSUPPORTED;
}

int hdf_launch_pack_batch(int dtype, const float* params, char* ws, const PackJob* jobs, int njobs, hipStream_t st) {
  HDF_CHECK_ARG(njobs >= 0 && njobs <= HDF_MAX_PACK_JOBS, "pack batch: %d jobs", njobs);
  if (njobs == 0) return HDF_OK;
  PackBatch b;
  for (int k = 0; k < njobs; k++) b.j[k] = jobs[k];
  dim3 grid(64, njobs);
  HDF_DISPATCH_T(dtype, hipLaunchKernelGGL(pack_batch_kernel<T>, grid, dim3(256), 0, st, b, params, ws));
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_pack_w(int dtype, const float* src, void* dst, int O, int I, int OP, int IP, int64_t so, int64_t si,
                      int flip, hipStream_t st) {
  int64_t total = (int64_t)27 * OP * IP;
  dim3 grid((unsigned)ceil_div64(total, 256));
  HDF_DISPATCH_T(dtype, hipLaunchKernelGGL(pack_w_kernel<T>, grid, dim3(256), 0, st, src, (T*)dst, O, I, OP, IP, so, si,
                                           flip));
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}
