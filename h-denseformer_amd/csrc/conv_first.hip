// First layer of the encoder (block_1_1_left: Conv3d(in_channels <= 4 -> n_filters, k3, p1), HDenseFormer.py:152-158,190):
// tap-packed K.  The generic stride-1 kernels contract one tap at a time over the input's channel
// row, which the first layer pads from 4 to 16 channels: 27 k-steps of 16 of which three quarters multiply zeros
// (conv_ws2<32,32>: 276 us, matrix pipe 0.17 busy).  Here K = (tap, channel) = 27 x 4 = 108 -> 128, eight k-steps:
//   * the tile's 6 x 10 x 10 input box (the 4 real channels = 8 bytes of each 32-byte voxel row) goes to LDS once, 600
//     8-byte loads per tile (the first version had every thread fetch its 27 neighbours from global memory: 6,912
//     32-byte-strided requests per tile kept the texture path, not HBM, busy -- 238 us);
//   * every thread owns one voxel of the 4 x 8 x 8 tile and builds that voxel's im2col row from the box: 27 ds_read_b64
//     (consecutive lanes = consecutive x: conflict-free) written at byte 8 * tap of a 272-byte row (256 + 16: the 32 rows
//     of an A fragment then spread over all banks);
//   * the weights [K = 128][Cout] sit in registers for the whole launch as B fragments (32 registers per output block,
//     read once from the fp32 parameters and rounded to the storage type like the packed panels are);
//   * a wave owns two 32-voxel M-blocks: 8 ds_read_b128 + 8 MFMAs each per tile.
// The next tile's box loads (3 per thread) are in flight under the current tile's MFMAs and stores.  Bound: the 268 MB of
// output at 128^3, batch 2 (HBM).  InstanceNorm partial sums leave as in conv_ws2_kernel: one row per workgroup and
// sample (WS_STAT_ROWS rows per sample, the unused ones zeroed here).
#include "conv_igemm.h"
#include "conv_tile.h"
#include <type_traits>

namespace {

constexpr int CF_PITCH = 272;  // bytes per im2col row
constexpr int CF_TD = 4, CF_TH = 8, CF_TW = 8, CF_VOX = CF_TD * CF_TH * CF_TW;

struct ConvFirstArgs {
  const void* in;       // channels-last, >= 4 channels per voxel (the first 4 are read), in_pitch elements apart
  int64_t in_pitch;
  int Cin;              // real input channels, 1..4
  int N, D, H, W;
  const float* w32;     // torch layout [Cout][Cin][3][3][3], fp32
  const float* bias;    // [Cout] or null
  void* out;
  int64_t out_pitch;
  int Cout, CoutP;
  float* stat_partials;  // [N][WS_STAT_ROWS][CoutP][2] or null
};

template <typename T, int NB>
__global__ __launch_bounds__(256, NB == 1 ? 2 : 1) void conv_first_kernel(ConvFirstArgs a) {
  __shared__ __attribute__((aligned(16))) char s_col[CF_VOX * CF_PITCH];
  __shared__ u32x2 s_box[(CF_TD + 2) * (CF_TH + 2) * (CF_TW + 2)];
  // channels >= Cin of the 4-channel (8-byte) voxel load are masked when the box is committed: the op-level entry promises
  // that only the first Cin channels are read, and a NaN there would reach every output as 0 * NaN
  const uint32_t cmask0 = a.Cin >= 2 ? 0xffffffffu : 0x0000ffffu, cmask1 = a.Cin >= 4 ? 0xffffffffu : (a.Cin == 3 ? 0x0000ffffu : 0u);
  __shared__ float s_red[4 * 32 * NB * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int n = blockIdx.y;
  const int ntz = (a.D + CF_TD - 1) / CF_TD, nty = (a.H + CF_TH - 1) / CF_TH, ntx = (a.W + CF_TW - 1) / CF_TW;
  const int ntile = ntz * nty * ntx;

  // ---- B fragments: lane (co = r, kg = h) holds k = 16 ks + 8 kg + j, j = 0..7; k = 4 tap + ci
  u32x4 wf[NB][8];
#pragma unroll
  for (int nb = 0; nb < NB; nb++) {
    const int co = 32 * nb + r;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int k = 16 * ks + 8 * h + j, tap = k >> 2, ci = k & 3;
        const bool ok = co < a.Cout && tap < 27 && ci < a.Cin;
        f[j] = ok ? a.w32[((int64_t)co * a.Cin + ci) * 27 + tap] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; j++) wf[nb][ks][j] = ST<T>::pack2(f[2 * j], f[2 * j + 1]);
    }
  }
  float bias[NB];
#pragma unroll
  for (int nb = 0; nb < NB; nb++) bias[nb] = (a.bias && 32 * nb + r < a.Cout) ? a.bias[32 * nb + r] : 0.f;

  // ---- this thread's voxel of a tile and the zero columns [108, 128) of its row (never overwritten)
  const int lx = tid & 7, ly = (tid >> 3) & 7, lz = tid >> 6;
  char* const my_row = s_col + tid * CF_PITCH;
#pragma unroll
  for (int b = 216; b < 256; b += 8) *reinterpret_cast<u32x2*>(my_row + b) = u32x2{0u, 0u};  // k = 108..127
  const T* const xin = reinterpret_cast<const T*>(a.in) + (int64_t)n * a.D * a.H * a.W * a.in_pitch;
  T* const outp = reinterpret_cast<T*>(a.out) + (int64_t)n * a.D * a.H * a.W * a.out_pitch;
  const int ip = (int)a.in_pitch, op = (int)a.out_pitch;

  constexpr int BD = CF_TD + 2, BH = CF_TH + 2, BW = CF_TW + 2, BOX = BD * BH * BW, NSL = (BOX + 255) / 256;
  u32x2 pf[NSL];
  unsigned pf_ok = 0;   // bit j: slot j of the prefetched box lies inside the volume
  auto tile_origin = [&](int t, int& z0, int& y0, int& x0) {
    const int tx = t % ntx, ty = (t / ntx) % nty, tz = t / (ntx * nty);
    z0 = tz * CF_TD, y0 = ty * CF_TH, x0 = tx * CF_TW;
  };
  // box slots of this thread: s = tid + 256 j -> (bz, by, bx)
  int sbz[NSL], sby[NSL], sbx[NSL];
#pragma unroll
  for (int j = 0; j < NSL; j++) {
    const int sl = min(tid + 256 * j, BOX - 1);
    sbz[j] = sl / (BH * BW), sby[j] = (sl / BW) % BH, sbx[j] = sl % BW;
  }
  // the box of tile t -> registers (clamped addresses; voxels outside the volume are zeros: the conv's padding)
  auto prefetch = [&](int t) __attribute__((always_inline)) {
    int z0, y0, x0;
    tile_origin(t, z0, y0, x0);
#pragma unroll
    for (int j = 0; j < NSL; j++) {
      const int z = z0 - 1 + sbz[j], y = y0 - 1 + sby[j], x = x0 - 1 + sbx[j];
      const bool ok = (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
      const int off = ((min(max(z, 0), a.D - 1) * a.H + min(max(y, 0), a.H - 1)) * a.W + min(max(x, 0), a.W - 1)) * ip;
      pf[j] = *reinterpret_cast<const u32x2*>(xin + off);   // (no use of the value here: a select on it would make the
      pf_ok = ok ? (pf_ok | (1u << j)) : (pf_ok & ~(1u << j));  //  wave wait for the load it has just issued)
    }
  };
  auto commit_box = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NSL; j++) {
      const bool ok = (pf_ok >> j) & 1u;
      if (tid + 256 * j < BOX) s_box[tid + 256 * j] = u32x2{ok ? pf[j][0] & cmask0 : 0u, ok ? pf[j][1] & cmask1 : 0u};
    }
  };
  // the tile's 256 x 27 (voxel, tap) items, tap fastest, dealt to the threads in order: a wave then writes 64 CONSECUTIVE
  // 8-byte slots of the rows (conflict-free; one voxel per lane put 16 lanes on each bank: rows are 68 dwords apart)
  unsigned item[27];  // box slot (10 bits) | row byte offset << 10 (one register per item: 54 cost an occupancy step)
#pragma unroll
  for (int j = 0; j < 27; j++) {
    const int i = tid + 256 * j, v = i / 27, tap = i - 27 * v;
    const int vz = v >> 6, vy = (v >> 3) & 7, vx = v & 7;
    item[j] = (unsigned)(((vz + tap / 9) * BH + vy + (tap / 3) % 3) * BW + vx + tap % 3) |
              ((unsigned)(v * CF_PITCH + 8 * tap) << 10);
  }
  auto build_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j0 = 0; j0 < 27; j0 += 9) {   // three rounds of nine: 18 transient registers
      u32x2 v[9];
#pragma unroll
      for (int j = 0; j < 9; j++) v[j] = s_box[item[j0 + j] & 1023u];
#pragma unroll
      for (int j = 0; j < 9; j++) *reinterpret_cast<u32x2*>(s_col + (item[j0 + j] >> 10)) = v[j];
    }
  };

  float lr1[NB], lr2[NB];
#pragma unroll
  for (int nb = 0; nb < NB; nb++) lr1[nb] = lr2[nb] = 0.f;

  // two barriers per tile: rows(t) built from box(t) | barrier | box(t+1) written from registers loaded a tile ago, the
  // loads of box(t+2) issued, MFMAs + stores of tile t | barrier
  int t = blockIdx.x;
  if (t < ntile) {
    prefetch(t);
    commit_box();
    prefetch(min(t + (int)gridDim.x, ntile - 1));
  }
  // drain here: entered with loads pending, the loop's wait for the prefetched box would have to hold for the entry path
  // too (no store younger than the loads) and would wait for every store of the previous tile on every trip
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  // (barriers that order LDS traffic only: __syncthreads() would wait for the tile's stores to be acknowledged, 4.9 us
  // per tile; and two copies of the loop, because a conditional store path makes hipcc wait for every outstanding
  // store wherever it waits for the prefetched box -- DESIGN 6e, the vmcnt finding)
  auto run = [&](auto full_tag) __attribute__((always_inline)) {
  for (; t < ntile; t += gridDim.x) {
    build_rows();
    WS_BARRIER();   // rows complete, nobody reads the box any more
    if (t + (int)gridDim.x < ntile) commit_box();
    prefetch(min(t + 2 * (int)gridDim.x, ntile - 1));   // unconditional (a skipped prefetch is a second path to the
                                                        // next commit, and hipcc then waits for the stores as well)
    int z0, y0, x0;
    tile_origin(t, z0, y0, x0);
    // ---- 2 M-blocks per wave: voxels 64 wave + 32 mb + m, row m of the block = (ly & 3) * 8 + lx
    f32x16 acc[2][NB];
#pragma unroll
    for (int mb = 0; mb < 2; mb++)
#pragma unroll
      for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[mb][nb][i] = 0.f;
#pragma unroll
    for (int mb = 0; mb < 2; mb++) {
      const char* const arow = s_col + (64 * wave + 32 * mb + r) * CF_PITCH + 16 * h;
      u32x4 af[8];
#pragma unroll
      for (int ks = 0; ks < 8; ks++) af[ks] = *reinterpret_cast<const u32x4*>(arow + 32 * ks);
#pragma unroll
      for (int ks = 0; ks < 8; ks++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++) Mma<T>::run(af[ks], wf[nb][ks], acc[mb][nb]);
    }
    // ---- epilogue: lane (co = r, h): accumulator i = voxel row (i & 3) + 8 (i >> 2) + 4 h of the M-block, i.e.
    // y = 4 (mb-th half) + (i >> 2), x = (i & 3) + 4 h; the wave's M-blocks are (lz = wave, y half mb)
    const int gz = z0 + wave;
    if constexpr (decltype(full_tag)::value) {
      // whole tile, whole channel blocks (uniform): branch-free, one per-lane base + uniform 32-bit offsets
      // A lane holds ONE channel of 16 voxels: stored as they are that is 16 two-byte stores per lane and block, and the
      // launch is bound by the number of store instructions (2.1 M wave stores for 268 MB).  The 4 x 4 blocks (4 lanes of
      // a quad = 4 channels) x (4 voxels) are transposed inside the quad (two DPP butterfly rounds), after which a lane
      // holds 4 channels of ONE voxel per group: 4 eight-byte stores per lane and block.
      T* const obase = outp + (((int64_t)gz * a.H + y0) * a.W + x0 + 4 * h + (r & 3)) * op + (r & ~3);
      const bool l1 = r & 1, l2 = r & 2;
#pragma unroll
      for (int mb = 0; mb < 2; mb++) {
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
#pragma unroll
          for (int g = 0; g < 4; g++) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
              v[j] = acc[mb][nb][4 * g + j] + bias[nb];
              lr1[nb] += v[j];
              lr2[nb] += v[j] * v[j];
            }
#pragma unroll
            for (int j = 0; j < 4; j += 2) {   // round 1: partner = lane ^ 1
              const float snd = l1 ? v[j] : v[j + 1];
              const float rcv = __builtin_bit_cast(
                  float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, snd), 0xB1, 0xF, 0xF, false));
              if (l1) v[j] = rcv; else v[j + 1] = rcv;
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {      // round 2: partner = lane ^ 2
              const float snd = l2 ? v[j] : v[j + 2];
              const float rcv = __builtin_bit_cast(
                  float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, snd), 0x4E, 0xF, 0xF, false));
              if (l2) v[j] = rcv; else v[j + 2] = rcv;
            }
            // v[c] = channel (r & ~3) + c of voxel row (r & 3) + 8 g + 4 h: y = 4 mb + g, x = (r & 3) + 4 h
            ST<T>::st4(obase + ((4 * mb + g) * a.W) * op + 32 * nb, v[0], v[1], v[2], v[3]);
          }
        }
      }
    } else {
#pragma unroll
      for (int mb = 0; mb < 2; mb++) {
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
          const int co = 32 * nb + r;
#pragma unroll
          for (int i = 0; i < 16; i++) {
            const int gy = y0 + 4 * mb + (i >> 2), gx = x0 + (i & 3) + 4 * h;
            const float v = acc[mb][nb][i] + bias[nb];
            const bool ok = gz < a.D && gy < a.H && gx < a.W && co < a.Cout;
            if (ok) ST<T>::st(outp + (((int64_t)gz * a.H + gy) * a.W + gx) * op + co, v);
            const float mk = ok ? 1.f : 0.f;
            lr1[nb] += mk * v;
            lr2[nb] += mk * v * v;
          }
        }
      }
    }
    WS_BARRIER();  // every wave is done with the im2col rows; the next box is complete
  }
  };
  if (a.D % CF_TD == 0 && a.H % CF_TH == 0 && a.W % CF_TW == 0 && 32 * NB <= a.Cout)
    run(std::true_type{});
  else
    run(std::false_type{});
  // ---- InstanceNorm partial sums: row blockIdx.x of sample n; the rows no workgroup owns are zeroed
  if (a.stat_partials) {
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      const float u1 = lr1[nb] + __shfl_xor(lr1[nb], 32, 64), u2 = lr2[nb] + __shfl_xor(lr2[nb], 32, 64);
      if (h == 0) {
        s_red[((wave * NB + nb) * 32 + r) * 2 + 0] = u1;
        s_red[((wave * NB + nb) * 32 + r) * 2 + 1] = u2;
      }
    }
    __syncthreads();
    if (tid < 32 * NB) {
      const int nb = tid >> 5, c = tid & 31;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        t1 += s_red[((k * NB + nb) * 32 + c) * 2 + 0];
        t2 += s_red[((k * NB + nb) * 32 + c) * 2 + 1];
      }
      for (int row = blockIdx.x; row < WS_STAT_ROWS; row += gridDim.x) {
        float* q = a.stat_partials + (((int64_t)n * WS_STAT_ROWS + row) * a.CoutP + tid) * 2;
        q[0] = row == (int)blockIdx.x ? t1 : 0.f;
        q[1] = row == (int)blockIdx.x ? t2 : 0.f;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[co][ci][tap] = sum over (sample, voxel) of dy[voxel][co] * x[voxel + tap][ci], the same tap-packed K seen from the
// other side: a 32 x 32 x 16 MFMA contracts 16 VOXELS, M = co, N = (tap, ci) = 108 -> 128: four N-blocks, one per wave, for
// 1/6.75 of the matrix instructions of the generic kernel (27 taps x a 32-channel block of which 4 channels are real).
// Per 2 x 8 x 8 tile: the input box and the dy rows go to LDS (registers loaded a tile ahead), every (voxel, tap) pair is
// copied from the box into the voxel's im2col row (320-byte pitch: the four voxel rows x 16 dwords a half-wave's
// ds_read_b64_tr_b16 touches then tile the 64 banks), and both MFMA operands come out of LDS through transposed reads
// (lane roles as in conv_wgrad2_kernel).  60 KB of LDS, ~120 registers: two workgroups per CU, and room for the
// transformer branch's kernels beside it (the generic kernel: 110 KB, one per CU).  Each workgroup leaves one fp32
// partial [Cout][128]; wgrad_first_reduce_kernel sums them in a fixed order into the torch layout.
constexpr int WF_TD = 2, WF_VOX = WF_TD * CF_TH * CF_TW, WF_PITCH = 320;
constexpr int WF_BD = WF_TD + 2, WF_BH = CF_TH + 2, WF_BW = CF_TW + 2, WF_BOX = WF_BD * WF_BH * WF_BW;

struct WgradFirstArgs {
  const void* dy;      // [N][D][H][W] rows of dy_pitch elements, SC = Cout channels used
  int64_t dy_pitch;
  int SC;
  const void* x;       // the layer's input, x_pitch elements per voxel, the first 4 read
  int64_t x_pitch;
  int Cin, N, D, H, W;
  float* partials;     // [gridDim.y * gridDim.x][SCp][128]
  int SCp;
  // XF: `dy` holds the gradient w.r.t. the layer's ACTIVATION relu(IN(y)) and the kernel applies the second pass of the
  // InstanceNorm(+ReLU) backward while it stages a row (in_bwd_apply4_kernel's arithmetic, rounded to the storage type like
  // the tensor that pass would have written): dy = k1 * ((y*scale+shift > 0 ? da : 0) - ka - (y - mean) * rstd * kb)
  const void* y;
  int64_t y_pitch;
  const float *scale, *shift, *mean, *rstd, *k1, *ka, *kb;  // [N][SC]
};

template <typename T, int MB, bool XF>
__global__ __launch_bounds__(256) void wgrad_first_kernel(WgradFirstArgs a) {
  __shared__ __attribute__((aligned(16))) char s_col[WF_VOX * WF_PITCH];
  __shared__ __attribute__((aligned(16))) char s_dy[2][WF_VOX * 64 * MB];
  __shared__ u32x2 s_box[WF_BOX];
  // channels >= Cin of the 4-channel (8-byte) voxel load are masked when the box is committed: the op-level entry promises
  // that only the first Cin channels are read, and a NaN there would reach every output as 0 * NaN
  const uint32_t cmask0 = a.Cin >= 2 ? 0xffffffffu : 0x0000ffffu, cmask1 = a.Cin >= 4 ? 0xffffffffu : (a.Cin == 3 ? 0x0000ffffu : 0u);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.y;
  const int ntz = (a.D + WF_TD - 1) / WF_TD, nty = (a.H + CF_TH - 1) / CF_TH, ntx = (a.W + CF_TW - 1) / CF_TW;
  const int ntile = ntz * nty * ntx, G = gridDim.x;
  const T* const xin = reinterpret_cast<const T*>(a.x) + (int64_t)n * a.D * a.H * a.W * a.x_pitch;
  const T* const dyin = reinterpret_cast<const T*>(a.dy) + (int64_t)n * a.D * a.H * a.W * a.dy_pitch;
  const int xp = (int)a.x_pitch, yp = (int)a.dy_pitch;
  constexpr int ROWB = 64 * MB;             // bytes of a dy row in LDS
  constexpr int NBX = (WF_BOX + 255) / 256;  // box slots per thread (2)
  constexpr int NDY = WF_VOX * (ROWB / 16) / 256;  // 16-byte dy chunks per thread (2 MB)
  auto tile_origin = [&](int t, int& z0, int& y0, int& x0) {
    const int tx = t % ntx, ty = (t / ntx) % nty, tz = t / (ntx * nty);
    z0 = tz * WF_TD, y0 = ty * CF_TH, x0 = tx * CF_TW;
  };
  // ---- per-thread staging slots
  int sbz[NBX], sby[NBX], sbx[NBX];
#pragma unroll
  for (int j = 0; j < NBX; j++) {
    const int sl = min(tid + 256 * j, WF_BOX - 1);
    sbz[j] = sl / (WF_BH * WF_BW), sby[j] = (sl / WF_BW) % WF_BH, sbx[j] = sl % WF_BW;
  }
  u32x2 pfb[NBX];
  u32x4 pfd[NDY], pfy[XF ? NDY : 1];
  unsigned pf_ok = 0;  // bits 0..NBX-1: box slot inside the volume; bits 8..: dy chunk valid
  // XF: the thread's dy chunks all cover the same 8 channels (256 is a multiple of the chunks per row)
  float xsc[XF ? 8 : 1], xsh[XF ? 8 : 1], xmu[XF ? 8 : 1], xrs[XF ? 8 : 1], xk1[XF ? 8 : 1], xka[XF ? 8 : 1], xkb[XF ? 8 : 1];
  const T* const yin = XF ? reinterpret_cast<const T*>(a.y) + (int64_t)n * a.D * a.H * a.W * a.y_pitch : nullptr;
  const int ypit = (int)a.y_pitch;
  if constexpr (XF) {
    const int ch = min((tid % (ROWB / 16)) * 8, a.SC - 8);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int64_t o = (int64_t)n * a.SC + ch + e;
      xsc[e] = a.scale[o], xsh[e] = a.shift[o], xmu[e] = a.mean[o], xrs[e] = a.rstd[o];
      xk1[e] = a.k1[o], xka[e] = a.ka[o], xkb[e] = a.kb[o];
    }
  }
  auto prefetch = [&](int t) __attribute__((always_inline)) {
    int z0, y0, x0;
    tile_origin(t, z0, y0, x0);
#pragma unroll
    for (int j = 0; j < NBX; j++) {
      const int z = z0 - 1 + sbz[j], y = y0 - 1 + sby[j], x = x0 - 1 + sbx[j];
      const bool ok = (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
      const int off = ((min(max(z, 0), a.D - 1) * a.H + min(max(y, 0), a.H - 1)) * a.W + min(max(x, 0), a.W - 1)) * xp;
      pfb[j] = *reinterpret_cast<const u32x2*>(xin + off);
      pf_ok = ok ? (pf_ok | (1u << j)) : (pf_ok & ~(1u << j));
    }
#pragma unroll
    for (int j = 0; j < NDY; j++) {
      const int c = tid + 256 * j, v = c / (ROWB / 16), ch = (c - v * (ROWB / 16)) * 8;
      const int z = z0 + (v >> 6), y = y0 + ((v >> 3) & 7), x = x0 + (v & 7);
      const bool ok = z < a.D && y < a.H && x < a.W && ch < a.SC;
      const int off = ((min(z, a.D - 1) * a.H + min(y, a.H - 1)) * a.W + min(x, a.W - 1)) * yp + min(ch, a.SC - 8);
      pfd[j] = *reinterpret_cast<const u32x4*>(dyin + off);
      if constexpr (XF) {
        const int offy = ((min(z, a.D - 1) * a.H + min(y, a.H - 1)) * a.W + min(x, a.W - 1)) * ypit + min(ch, a.SC - 8);
        pfy[j] = *reinterpret_cast<const u32x4*>(yin + offy);
      }
      pf_ok = ok ? (pf_ok | (256u << j)) : (pf_ok & ~(256u << j));
    }
  };
  auto commit = [&](int par) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NBX; j++) {
      const bool ok = (pf_ok >> j) & 1u;
      if (tid + 256 * j < WF_BOX) s_box[tid + 256 * j] = u32x2{ok ? pfb[j][0] & cmask0 : 0u, ok ? pfb[j][1] & cmask1 : 0u};
    }
#pragma unroll
    for (int j = 0; j < NDY; j++) {
      const bool ok = (pf_ok >> (8 + j)) & 1u;
      u32x4 v = pfd[j];
      if constexpr (XF) {
        float g[8], f[8];
        ST<T>::unpack(v, g);
        ST<T>::unpack(pfy[j], f);
#pragma unroll
        for (int e = 0; e < 8; e++) {
          g[e] = in_bwd_elem(g[e], f[e], xsc[e], xsh[e], xmu[e], xrs[e], xk1[e], xka[e], xkb[e]);
        }
        v = ST<T>::pack(g);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = ok ? v[k] : 0u;
      *reinterpret_cast<u32x4*>(s_dy[par] + (tid + 256 * j) * 16) = v;
    }
  };
  const int v = tid & (WF_VOX - 1), half = tid >> 7;
  char* const my_row = s_col + v * WF_PITCH;
  if (half == 0) {
#pragma unroll
    for (int b = 216; b < 256; b += 8) *reinterpret_cast<u32x2*>(my_row + b) = u32x2{0u, 0u};  // k = 108..127
  }
  // the 128 x 27 (voxel, tap) items, tap fastest, dealt to the threads in order: consecutive lanes write consecutive
  // 8-byte slots (one voxel per lane put 16 lanes on each bank: rows are 80 dwords apart)
  constexpr int NIT = (WF_VOX * 27 + 255) / 256;  // 14 (the last round is half empty)
  int it_src[NIT], it_dst[NIT];
#pragma unroll
  for (int j = 0; j < NIT; j++) {
    const int i = min(tid + 256 * j, WF_VOX * 27 - 1), vv = i / 27, tap = i - 27 * vv;
    const int vz = vv >> 6, vy = (vv >> 3) & 7, vx = vv & 7;
    it_src[j] = ((vz + tap / 9) * WF_BH + vy + (tap / 3) % 3) * WF_BW + vx + tap % 3;
    it_dst[j] = vv * WF_PITCH + 8 * tap;
  }
  auto build_rows = [&]() __attribute__((always_inline)) {
    u32x2 t[NIT];
#pragma unroll
    for (int j = 0; j < NIT; j++) t[j] = s_box[it_src[j]];
#pragma unroll
    for (int j = 0; j < NIT; j++) *reinterpret_cast<u32x2*>(s_col + it_dst[j]) = t[j];   // (the clamped tail rewrites the last item)
  };
  // ---- transposed-read lane roles (conv_wgrad2_kernel): 16 lanes cover 4 voxel rows x 16 channels, lane i16 receives
  // channel cb + i16 of the 4 voxels
  const int g4 = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3, hh = g4 >> 1, cb = (g4 & 1) * 16;
  const int colb = (cb + 4 * p4) * 2;
  using lds_s16x4 = s16x4 __attribute__((address_space(3)));
  auto tr_read = [&](const char* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p));
  };
  f32x16 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[mb][i] = 0.f;

  int t = blockIdx.x, par = 0;
  if (t < ntile) {
    prefetch(t);
    commit(0);
    prefetch(min(t + G, ntile - 1));
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (; t < ntile; t += G) {
    build_rows();
    WS_BARRIER();   // rows complete; the box is free
    if (t + G < ntile) commit(par ^ 1);
    prefetch(min(t + 2 * G, ntile - 1));
    const char* const dyb = s_dy[par];
#pragma unroll
    for (int ks = 0; ks < WF_VOX / 16; ks++) {
      const int row0 = 16 * ks + 8 * hh + q;
      const u32x2 b0 = tr_read(s_col + row0 * WF_PITCH + 64 * wave + colb);
      const u32x2 b1 = tr_read(s_col + (row0 + 4) * WF_PITCH + 64 * wave + colb);
      const u32x4 bf = {b0[0], b0[1], b1[0], b1[1]};
#pragma unroll
      for (int mb = 0; mb < MB; mb++) {
        const u32x2 a0 = tr_read(dyb + row0 * ROWB + 64 * mb + colb);
        const u32x2 a1 = tr_read(dyb + (row0 + 4) * ROWB + 64 * mb + colb);
        Mma<T>::run(u32x4{a0[0], a0[1], a1[0], a1[1]}, bf, acc[mb]);
      }
    }
    par ^= 1;
    WS_BARRIER();   // every wave is done with the rows and this dy buffer; the next box / dy are complete
  }
  // ---- this workgroup's partial: rows co = 32 mb + (i & 3) + 8 (i >> 2) + 4 h, columns 32 wave + (lane & 31)
  float* const part = a.partials + ((int64_t)blockIdx.y * G + blockIdx.x) * a.SCp * 128;
  const int h = lane >> 5, col = 32 * wave + (lane & 31);
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int i = 0; i < 16; i++) part[(32 * mb + (i & 3) + 8 * (i >> 2) + 4 * h) * 128 + col] = acc[mb][i];
}

// dw[(co * Cin + ci) * 27 + tap] (+)= sum_g partial[g][co][4 tap + ci]: a workgroup owns 32 consecutive entries, its 8
// lane groups take every 8th partial (4 loads in flight each), summed through LDS in a fixed order: bitwise reproducible
// (one thread per entry walking all 512 partials: 103 us of dependent loads)
__global__ __launch_bounds__(256) void wgrad_first_reduce_kernel(const float* __restrict__ partials, float* __restrict__ dw,
                                                                 int G, int SCp, int SC, int Cin, int accumulate) {
  __shared__ float red[8][33];
  const int el = threadIdx.x & 31, gl = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + el;
  const int64_t per = (int64_t)SCp * 128;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  int g = gl;
  for (; g + 24 < G; g += 32) {
#pragma unroll
    for (int k = 0; k < 4; k++) s[k] += partials[(int64_t)(g + 8 * k) * per + idx];
  }
  for (; g < G; g += 8) s[0] += partials[(int64_t)g * per + idx];
  red[gl][el] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (threadIdx.x < 32) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) t += red[k][el];
    const int co = idx >> 7, k = idx & 127, tap = k >> 2, ci = k & 3;
    if (co < SC && tap < 27 && ci < Cin) {
      float* o = dw + ((int64_t)co * Cin + ci) * 27 + tap;
      *o = accumulate ? *o + t : t;
    }
  }
}

}  // namespace

bool hdf_conv_first_can(int dtype, int Cin, int Cout, int D, int H, int W, int64_t in_pitch) {
  return dtype != HDF_F32 && Cin >= 1 && Cin <= 4 && Cout >= 1 && Cout <= 64 && in_pitch % 4 == 0 && D >= 1 && H >= 1 &&
         W >= 1 && (int64_t)D * H * W * in_pitch < ((int64_t)1 << 31);
}

// the plan's routing rule: where the generic launcher would use conv_ws2_kernel (same InstanceNorm partials geometry)
bool hdf_conv_first_takes(int dtype, int Cin, int Cout, int D, int H, int W, int64_t in_pitch) {
  return hdf_conv_first_can(dtype, Cin, Cout, D, H, W, in_pitch) && D > 1 && (int64_t)D * H * W >= 48 * 48 * 48 &&
         hdf_conv_stat_tiles(0, D, H, W, 32) == WS_STAT_ROWS;
}

int hdf_launch_conv_first(int dtype, const void* in, int64_t in_pitch, int Cin, int N, int D, int H, int W,
                          const float* w32, const float* bias, void* out, int64_t out_pitch, int Cout,
                          float* stat_partials, hipStream_t st) {
  HDF_CHECK_ARG(hdf_conv_first_can(dtype, Cin, Cout, D, H, W, in_pitch), "conv_first: dtype %d Cin %d Cout %d %dx%dx%d",
                dtype, Cin, Cout, D, H, W);
  HDF_CHECK_ARG((reinterpret_cast<uintptr_t>(in) & 7) == 0, "conv_first: input must be 8-byte aligned");
  HDF_CHECK_ARG((reinterpret_cast<uintptr_t>(out) & 7) == 0 && out_pitch % 4 == 0,
                "conv_first: output must be 8-byte aligned with a pitch that is a multiple of 4 channels");
  ConvFirstArgs a{};
  a.in = in, a.in_pitch = in_pitch, a.Cin = Cin, a.N = N, a.D = D, a.H = H, a.W = W;
  a.w32 = w32, a.bias = bias, a.out = out, a.out_pitch = out_pitch, a.Cout = Cout, a.CoutP = (Cout + 31) / 32 * 32;
  a.stat_partials = stat_partials;
  const int tiles = ((D + CF_TD - 1) / CF_TD) * ((H + CF_TH - 1) / CF_TH) * ((W + CF_TW - 1) / CF_TW);
  // two workgroups per CU (70 KB of LDS each): one's barriers and stores hide under the other's tile
  const int gx = std::max(1, std::min(tiles, std::min(WS_STAT_ROWS, 2 * hdf_cu_budget() / std::max(1, N))));
  const dim3 grid(gx, N);
  if (a.CoutP == 32) {
    if (dtype == HDF_BF16)
      hipLaunchKernelGGL((conv_first_kernel<bf16_t, 1>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((conv_first_kernel<f16_t, 1>), grid, dim3(256), 0, st, a);
  } else {
    if (dtype == HDF_BF16)
      hipLaunchKernelGGL((conv_first_kernel<bf16_t, 2>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((conv_first_kernel<f16_t, 2>), grid, dim3(256), 0, st, a);
  }
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

bool hdf_wgrad_first_takes(int dtype, int Cin, int Cout, int D, int H, int W, int64_t x_pitch, int64_t dy_pitch) {
  return hdf_conv_first_can(dtype, Cin, Cout, D, H, W, x_pitch) && Cout % 8 == 0 && dy_pitch % 8 == 0 &&
         (int64_t)D * H * W * dy_pitch < ((int64_t)1 << 31) && D > 1 && (int64_t)D * H * W >= 48 * 48 * 48;
}

int hdf_launch_wgrad_first(int dtype, const void* dy, int64_t dy_pitch, int Cout, const void* x, int64_t x_pitch, int Cin,
                           int N, int D, int H, int W, float* dw, int accumulate, void* workspace, size_t workspace_bytes,
                           hipStream_t st, const WgradFirstIn* in_bwd) {
  HDF_CHECK_ARG(hdf_conv_first_can(dtype, Cin, Cout, D, H, W, x_pitch) && Cout % 8 == 0 && dy_pitch % 8 == 0 &&
                    (int64_t)D * H * W * dy_pitch < ((int64_t)1 << 31),
                "wgrad_first: dtype %d Cin %d Cout %d %dx%dx%d", dtype, Cin, Cout, D, H, W);
  HDF_CHECK_ARG(((reinterpret_cast<uintptr_t>(dy) & 15) | (reinterpret_cast<uintptr_t>(x) & 7)) == 0,
                "wgrad_first: operands must be 16- / 8-byte aligned");
  WgradFirstArgs a{};
  a.dy = dy, a.dy_pitch = dy_pitch, a.SC = Cout, a.x = x, a.x_pitch = x_pitch, a.Cin = Cin;
  a.N = N, a.D = D, a.H = H, a.W = W, a.SCp = (Cout + 31) / 32 * 32;
  const int tiles = ((D + WF_TD - 1) / WF_TD) * ((H + CF_TH - 1) / CF_TH) * ((W + CF_TW - 1) / CF_TW);
  const int64_t per = (int64_t)a.SCp * 128 * sizeof(float);
  int gx = std::max(1, std::min(tiles, 2 * hdf_cu_budget() / std::max(1, N)));  // 60 KB of LDS: two per CU
  gx = (int)std::min<int64_t>(gx, (int64_t)workspace_bytes / (per * N));
  HDF_CHECK_ARG(gx >= 1, "wgrad_first: workspace of %zu bytes too small", workspace_bytes);
  a.partials = reinterpret_cast<float*>(workspace);
  if (in_bwd) {
    HDF_CHECK_ARG(in_bwd->y && in_bwd->scale && in_bwd->shift && in_bwd->mean && in_bwd->rstd && in_bwd->k1 && in_bwd->ka &&
                      in_bwd->kb && in_bwd->y_pitch % 8 == 0 && (reinterpret_cast<uintptr_t>(in_bwd->y) & 15) == 0 &&
                      (int64_t)D * H * W * in_bwd->y_pitch < ((int64_t)1 << 31),
                  "wgrad_first: bad InstanceNorm-backward operands");
    a.y = in_bwd->y, a.y_pitch = in_bwd->y_pitch, a.scale = in_bwd->scale, a.shift = in_bwd->shift, a.mean = in_bwd->mean;
    a.rstd = in_bwd->rstd, a.k1 = in_bwd->k1, a.ka = in_bwd->ka, a.kb = in_bwd->kb;
  }
  const dim3 grid(gx, N);
  auto go = [&](auto t_tag, auto mb_tag) {
    using T = decltype(t_tag);
    constexpr int MB = decltype(mb_tag)::value;
    if (in_bwd)
      hipLaunchKernelGGL((wgrad_first_kernel<T, MB, true>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((wgrad_first_kernel<T, MB, false>), grid, dim3(256), 0, st, a);
  };
  if (a.SCp == 32) {
    if (dtype == HDF_BF16)
      go(bf16_t{}, std::integral_constant<int, 1>{});
    else
      go(f16_t{}, std::integral_constant<int, 1>{});
  } else {
    if (dtype == HDF_BF16)
      go(bf16_t{}, std::integral_constant<int, 2>{});
    else
      go(f16_t{}, std::integral_constant<int, 2>{});
  }
  HDF_LAUNCH_CHECK();
  hipLaunchKernelGGL(wgrad_first_reduce_kernel, dim3(a.SCp * 128 / 32), dim3(256), 0, st, a.partials, dw, gx * N,
                     a.SCp, Cout, Cin, accumulate);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}
