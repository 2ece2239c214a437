// conv_wr: Conv3d(k3,s1,p1) forward / data gradient with the WEIGHTS RESIDENT IN REGISTERS (16-bit storage, 64- or
// 128-byte channel rows at the two highest resolutions; BasicConv3d / UpConv convs, HDenseFormer.py:151,167).
//
// Why: conv_ws2_kernel (conv_igemm.hip) keeps the weight panel in LDS and feeds every MFMA with one A and one B fragment
// read -- 7 ds_read_b128 per 6 MFMAs plus the staging stores, four 32-byte channel passes per tile because the panel
// (110 KB for 64 input channels) leaves room for two 19 KB tile buffers only.  Three rounds of scheduling work left it at
// 0.39 of the bf16 MFMA roof with the matrix pipe busy 52 % of the cycles: the limit is LDS bytes and issue
// slots per MFMA at one wave per SIMD, and the short passes (54 MFMAs) pay their top / barrier / epilogue every 2,000
// cycles.  Here:
//  * a wave holds its 27 taps x KSW k-steps x NB output blocks = 54 weight fragments (216 registers, AGPRs: MFMA operands
//    may live there) for the whole launch: NO weight traffic after the prologue, the LDS holds two WHOLE-ROW tile buffers;
//  * a wave owns 4 consecutive-y M-blocks: the A fragment of box row y' serves (mb, jy) with mb + jy == y', so a (jz, jx,
//    k-step) group is 6 ds_read_b128 for 12 NB MFMAs (0.5 / 0.25 reads per MFMA against 1.17);
//  * one phase per tile: 216 MFMAs of straight-line code between two barriers, every input line fetched once per tile
//    (the four-pass form asked L2 for each 128-byte line four times);
//  * the four waves split K where the rows are wide (KSPLIT = 2: wave (kq, mq) contracts k-steps [kq KSW, (kq+1) KSW) for
//    M-blocks 4 mq .. 4 mq + 3) and the two K-halves of a tile are summed through LDS once per tile (each wave sends the
//    two M-blocks its partner keeps: 8 NB ds_write_b128 + 8 NB ds_read_b128), or split M only (KSPLIT = 1, 8x8x8 tile).
// Instantiations (all 54 fragments per wave):
//    A  RB 128, NB 1, KSPLIT 2, tile 4x8x8   64 -> 32k channels   (b11R, up3, the 64 -> 64 layers at 64^3 and their dgrads)
//    B  RB  64, NB 1, KSPLIT 1, tile 8x8x8   32 -> 32              (b12L, b12R and their dgrads)
//    C  RB  64, NB 2, KSPLIT 2, tile 4x8x8   32 -> 64k             (b21L, the dgrads of b11R and up3)
// LDS image: box voxel (bz, by, bx) = row (bz 10 + by) 10 + bx of RB bytes, unpadded; the 16-byte slots of a row are
// XOR-swizzled by a key of (bz, bx) only -- independent of by, so the six reads of a group share ONE address register and
// differ in their immediate offsets -- chosen so that each 16-lane group of a ds_read_b128 (ws_row_to_zx: z in {b, b+2}
// x eight consecutive x) covers all 64 banks: 128-byte rows alternate bank halves with bx & 1, key = (bx>>1)&3 | (bz>>1)&1
// << 2; 64-byte rows select a bank quarter with (2 by + bx) & 3, key = (bx>>2)&1 | (bz>>1)&1 << 1.
// Staging is LDS-DMA (global_load_lds_dwordx4, swizzle on the source side, the producer's InstanceNorm + ReLU applied in
// place afterwards): no staging registers beside the 216 weight registers, all rounds of the next tile in flight from the
// first groups of a phase on.  The epilogue of a whole tile is deferred into the next tile's phase (paired-row stores, one
// piece per group).  Everything else follows conv_ws2_kernel: persistent workgroups, interior tiles through an unchecked
// copy of the phase then border tiles through a checked one, XCD-aware tile split, InstanceNorm partial sums per lane.
// Measurements, the attribution that led here and why the plan routes only one layer class to this kernel: DESIGN.md 6e.
// Build flags (build.py EXTRA): -mllvm -amdgpu-mfma-vgpr-form, -mllvm -pragma-unroll-threshold=1000000, -fno-slp-vectorize.
#include "conv_igemm.h"
#include "conv_tile.h"
#include <type_traits>

namespace {

// 16 zero bytes: the source of DMA slots that lie outside the tensor (zero padding)
__device__ __attribute__((aligned(16))) uint32_t g_wr_zero_line[4] = {0u, 0u, 0u, 0u};

// s_waitcnt vmcnt(n) for a value that is a constant only after unrolling (the builtin wants a literal): the switch folds
#define WR_VMCNT_CASE(n) \
  case n:                \
    __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14)); \
    break;
__device__ __forceinline__ void wr_wait_vmcnt(int n) {
  switch (n) {
    WR_VMCNT_CASE(0) WR_VMCNT_CASE(1) WR_VMCNT_CASE(2) WR_VMCNT_CASE(3) WR_VMCNT_CASE(4) WR_VMCNT_CASE(5)
    WR_VMCNT_CASE(6) WR_VMCNT_CASE(7) WR_VMCNT_CASE(8) WR_VMCNT_CASE(9) WR_VMCNT_CASE(10) WR_VMCNT_CASE(11)
    WR_VMCNT_CASE(12) WR_VMCNT_CASE(13) WR_VMCNT_CASE(14) WR_VMCNT_CASE(15) WR_VMCNT_CASE(16) WR_VMCNT_CASE(17)
    WR_VMCNT_CASE(18) WR_VMCNT_CASE(19) WR_VMCNT_CASE(20) WR_VMCNT_CASE(21) WR_VMCNT_CASE(22) WR_VMCNT_CASE(23)
    WR_VMCNT_CASE(24) WR_VMCNT_CASE(25) WR_VMCNT_CASE(26) WR_VMCNT_CASE(27) WR_VMCNT_CASE(28) WR_VMCNT_CASE(29)
    WR_VMCNT_CASE(30) WR_VMCNT_CASE(31) WR_VMCNT_CASE(32) WR_VMCNT_CASE(33) WR_VMCNT_CASE(34) WR_VMCNT_CASE(35)
    WR_VMCNT_CASE(36) WR_VMCNT_CASE(37) WR_VMCNT_CASE(38) WR_VMCNT_CASE(39) WR_VMCNT_CASE(40) WR_VMCNT_CASE(41)
    WR_VMCNT_CASE(42) WR_VMCNT_CASE(43) WR_VMCNT_CASE(44) WR_VMCNT_CASE(45) WR_VMCNT_CASE(46) WR_VMCNT_CASE(47)
    WR_VMCNT_CASE(48) WR_VMCNT_CASE(49) WR_VMCNT_CASE(50) WR_VMCNT_CASE(51) WR_VMCNT_CASE(52) WR_VMCNT_CASE(53)
    WR_VMCNT_CASE(54) WR_VMCNT_CASE(55) WR_VMCNT_CASE(56) WR_VMCNT_CASE(57) WR_VMCNT_CASE(58) WR_VMCNT_CASE(59)
    WR_VMCNT_CASE(60)
    default:
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): always sufficient
  }
}

template <int RB>
__device__ __forceinline__ int wr_key(int bz, int bx) {
  return RB == 128 ? (((bx >> 1) & 3) | (((bz >> 1) & 1) << 2)) : (((bx >> 2) & 1) | (((bz >> 1) & 1) << 1));
}

template <typename T, int RB, int NB, int KSPLIT, int TD, bool XF>
__global__ __launch_bounds__(256) void conv_wr_kernel(ConvArgs a) {
  constexpr int TH = 8, TW = 8, BD = TD + 2, BH = TH + 2, BW = TW + 2, BOX = BD * BH * BW;
  constexpr int ESZ = sizeof(T), EPC = 16 / ESZ;
  static_assert(ESZ == 2, "16-bit storage only (fp32 fragments are four times as many registers)");
  constexpr int CIN = RB / ESZ;
  constexpr int NKS = RB / 32, KSW = NKS / KSPLIT, MSPLIT = 4 / KSPLIT, MBW = 4;
  static_assert(2 * TD / MSPLIT == MBW, "a wave owns four consecutive-y M-blocks");
  static_assert(27 * KSW * NB * 4 <= 216, "weight fragments per wave");
  constexpr int KEEP = MBW / KSPLIT;  // M-blocks a wave stores (after the K exchange)
  constexpr int NG = 9 * KSW;         // (jz, jx, k-step) groups per tile
  constexpr int CPV = RB / 16, VPS = 256 / CPV;  // 16-byte chunks per voxel row; voxels per staging round
  constexpr int NJ = (BOX + VPS - 1) / VPS;      // staging slots per thread and tile
  constexpr int ABUF = NJ * VPS * RB;            // tile buffer = whole staging rounds of 4 KB (the LDS-DMA image is lane-linear)
  constexpr int NC = 32 * NB;
  constexpr int OFF_RED = 2 * ABUF;              // [4 waves][NC][2] floats
  constexpr int XCH_BYTES = (KSPLIT == 2) ? 4 * KEEP * NB * 16 * 64 * 4 : 0;  // 4 waves x outgoing accumulators
  constexpr bool XCH_ALIAS = OFF_RED + 4 * NC * 8 + XCH_BYTES > 160 * 1024;   // scratch inside the tile buffer just read
  constexpr int OFF_XCH = OFF_RED + 4 * NC * 8;
  constexpr int LDS_BYTES = OFF_XCH + (XCH_ALIAS ? 0 : XCH_BYTES);
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert(!XCH_ALIAS || XCH_BYTES <= ABUF, "exchange scratch inside one tile buffer");
  __shared__ __attribute__((aligned(256))) char lds[LDS_BYTES];
  float* const s_red = reinterpret_cast<float*>(lds + OFF_RED);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int kq = wave % KSPLIT, mq = wave / KSPLIT;
  const int zb = (TD == 8) ? (mq & 1) : 0;                   // z half of the 8x8x8 tile
  const int ybase = (TD == 8) ? 4 * (mq >> 1) : 4 * mq;      // first of the wave's four y rows
  const int n0 = blockIdx.y * NC;
  const int part = tid & (CPV - 1), tv = tid / CPV;
  const float relu_lo = (XF && a.in_relu) ? 0.f : -INFINITY;

  // ---- weights -> registers, once: fragment (tap, k-step, output block) = 16 bytes of row n0 + 32 nb + r of the packed
  // [27][CoutP][Cin] panel.  54 loads per lane, all issued before the first use.
  u32x4 wreg[27][KSW][NB];
  {
    const char* wb = reinterpret_cast<const char*>(a.w) + ((int64_t)(n0 + r) * CIN + (kq * KSW) * 16 + h * 8) * ESZ;
#pragma unroll
    for (int tap = 0; tap < 27; tap++)
#pragma unroll
      for (int ks = 0; ks < KSW; ks++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
          wreg[tap][ks][nb] =
              *reinterpret_cast<const u32x4*>(wb + ((int64_t)(tap * a.CoutP + 32 * nb) * CIN + ks * 16) * ESZ);
    // AGPRs: MFMA operands may live there, and the 256 architectural registers are needed for everything the vector ALU
    // touches.  Without the pin the allocator treats the fragments as VGPRs "spilled" to AGPRs and copies each one back
    // (4 v_accvgpr_read) in front of every use.
#pragma unroll
    for (int tap = 0; tap < 27; tap++)
#pragma unroll
      for (int ks = 0; ks < KSW; ks++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++) asm volatile("" : "+a"(wreg[tap][ks][nb]));
  }

  // ---- tile schedule (conv_ws2_kernel's): interior tiles first (unchecked phase), then border tiles (checked phase),
  // both split XCD-aware: XCD x owns the x-th eighth of the raster-ordered list, its workgroups walk it interleaved.
  const int ntz = (a.Do + TD - 1) / TD, nty = (a.Ho + TH - 1) / TH, ntx = (a.Wo + TW - 1) / TW;
  // plain: every lane stores (whole 32-channel blocks), nothing is accumulated into the output: the epilogue of a whole
  // tile is then branch-free -- a fixed number of stores, see tile_phase.  Launches that are not plain run every tile
  // through the checked phase (no interior pass).
  const bool plain = !a.accumulate && n0 + NC <= a.Cout && a.out_pitch % 2 == 0 &&
                     (reinterpret_cast<uintptr_t>(a.out) & 3) == 0 && (!a.split || (reinterpret_cast<uintptr_t>(a.out2) & 3) == 0);
  const bool all_full = plain && a.Do % TD == 0 && a.Ho % TH == 0 && a.Wo % TW == 0;  // no ragged tile in the launch
  const bool has_int = plain && ntz >= 3 && nty >= 3 && ntx >= 3;
  const int ipz = ntz - 2, ipy = nty - 2, ipx = ntx - 2;
  const int n_int = has_int ? a.N * ipz * ipy * ipx : 0;
  const int per_all = ntz * nty * ntx, per_bor = per_all - (has_int ? ipz * ipy * ipx : 0);
  const int n_bor = a.N * per_bor;
  const int G = gridDim.x;
  const int NX = (G % 8 == 0) ? 8 : 1;
  const int WPX = G / NX;
  const int xcd = blockIdx.x % NX, slot = blockIdx.x / NX;
  auto split = [&](int total, int& begin, int& cnt) __attribute__((always_inline)) {
    const int r0 = (int)((int64_t)total * xcd / NX), r1 = (int)((int64_t)total * (xcd + 1) / NX);
    begin = r0 + slot;
    cnt = (r1 - r0 > slot) ? (r1 - r0 - slot + WPX - 1) / WPX : 0;
  };
  int int_begin, int_cnt, bor_begin, bor_cnt;
  split(n_int, int_begin, int_cnt);
  split(n_bor, bor_begin, bor_cnt);
  // InstanceNorm partial rows of this workgroup are zeroed even if it has no tile (in_finalize reads all of them)
  auto stat_row = [&](int n, int pass, int b) __attribute__((always_inline)) {
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
  if (int_cnt + bor_cnt == 0) return;
  int sdx = 0, sdy = 0, sdz = 0, sdn = 0;  // mixed-radix digits of the stride WPX over the interior grid
  if (has_int) {
    int t = WPX;
    sdx = t % ipx, t /= ipx;
    sdy = t % ipy, t /= ipy;
    sdz = t % ipz, sdn = t / ipz;
  }

  // ---- per-lane constants
  // A fragment addresses: ga[jz][jx] = this lane's 16-byte slot of box row (4 zb + dz + jz, ybase, x + jx) for the
  // wave's first k-step; k-step ks: ^ (ks * 32); box row y': + y' * BW * RB (immediate)
  int ga[3][3];
  {
    int dz, x;
    ws_row_to_zx(r, dz, x);
#pragma unroll
    for (int jz = 0; jz < 3; jz++)
#pragma unroll
      for (int jx = 0; jx < 3; jx++) {
        const int bz = 4 * zb + dz + jz, bx = x + jx;
        ga[jz][jx] = ((bz * BH + ybase) * BW + bx) * RB + (((2 * KSW * kq + h) ^ wr_key<RB>(bz, bx)) << 4);
      }
  }
  // staging slots: round j covers box voxels [VPS j, VPS j + VPS); this thread: voxel VPS j + tv, 16-byte chunk `part`
  // (threads past the end of the box redo its last voxel: identical bytes to the same address)
  // One 16-bit field per slot -- box coordinates bx | by << 4 | bz << 8 and the swizzled chunk << 12 -- two slots per
  // register: NJ / 2 registers instead of two NJ-entry tables.  (Tables of byte offsets, 2 NJ registers that hipcc widens
  // to 64 bits, do not fit beside 216 weight registers: they were spilled, and every reload in the phase is an
  // s_waitcnt vmcnt(0) -- a drain of the prefetch queue.)  A slot's global offset is then three 24-bit multiply-adds with
  // the uniform strides, its LDS offset affine in j plus the swizzled chunk, its bounds test needs no division.
  constexpr int NCP = (NJ + 1) / 2;
  uint32_t cpk[NCP];
#pragma unroll
  for (int k = 0; k < NCP; k++) cpk[k] = 0;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const unsigned vox = min(VPS * j + tv, BOX - 1);
    const unsigned bz = vox / (BH * BW), rem = vox - bz * (BH * BW), by = rem / BW, bx = rem - by * BW;
    const unsigned sw = part ^ wr_key<RB>((int)bz, (int)bx);
    cpk[j / 2] |= (bx | (by << 4) | (bz << 8) | (sw << 12)) << (16 * (j % 2));
  }
  const uint32_t XS = (uint32_t)a.in_pitch * ESZ, YS = XS * a.Wi, ZS = YS * a.Hi;  // < 2^24 (launcher)
  // the field of slot j, opaque to the optimiser: everything derived from it is recomputed where it is used (left
  // visible, all 3 NJ coordinates and NJ offsets are hoisted out of the tile loop into registers again)
  auto slot_field = [&](int j) __attribute__((always_inline)) {
    uint32_t c = cpk[j / 2];
    asm volatile("" : "+v"(c));
    return (j % 2) ? (c >> 16) : (c & 0xffffu);
  };
  // global byte offset of the chunk that belongs into THIS lane's 16 bytes of the (lane-linear) LDS image: the XOR
  // swizzle is applied on the source side of the DMA (chunk part ^ key of the voxel; an involution, so the same field is
  // also the LDS position of the thread's own chunk `part`, used by the in-place transform)
  auto slot_goff = [&](uint32_t f) __attribute__((always_inline)) {
    return __umul24(f & 15u, XS) + __umul24((f >> 4) & 15u, YS) + __umul24((f >> 8) & 15u, ZS) + ((f >> 12) << 4);
  };
  auto tile_lin = [&](const WsTile& c) __attribute__((always_inline)) { return ((c.n * ntz + c.z0 / TD) * nty + c.y0 / TH) * ntx + c.x0 / TW; };
  auto int_init = [&](WsTile& c, int k) __attribute__((always_inline)) {
    int t = k;
    c.x0 = (t % ipx + 1) * TW;
    t /= ipx;
    c.y0 = (t % ipy + 1) * TH;
    t /= ipy;
    c.z0 = (t % ipz + 1) * TD;
    c.n = t / ipz;
    c.tile = tile_lin(c);
  };
  auto int_next = [&](WsTile& c) __attribute__((always_inline)) {
    int xi = c.x0 / TW - 1 + sdx, yi = c.y0 / TH - 1 + sdy, zi = c.z0 / TD - 1 + sdz;
    c.n += sdn;
    if (xi >= ipx) xi -= ipx, yi++;
    if (yi >= ipy) yi -= ipy, zi++;
    if (zi >= ipz) zi -= ipz, c.n++;
    c.x0 = (xi + 1) * TW, c.y0 = (yi + 1) * TH, c.z0 = (zi + 1) * TD;
    c.tile = tile_lin(c);
  };
  auto bor_init = [&](WsTile& c, int k) __attribute__((always_inline)) {
    int tz, ty, tx;
    c.k = k;
    c.n = k / per_bor;
    int rem = k - c.n * per_bor;
    if (!has_int) {
      tx = rem % ntx, ty = (rem / ntx) % nty, tz = rem / (ntx * nty);
    } else {
      const int plane = nty * ntx, ring = plane - ipy * ipx;
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
  auto bor_next = [&](WsTile& c) __attribute__((always_inline)) { bor_init(c, c.k + WPX); };
  auto tile_interior = [&](const WsTile& c) __attribute__((always_inline)) {
    return c.z0 >= 1 && c.y0 >= 1 && c.x0 >= 1 && c.z0 + TD + 1 <= a.Di && c.y0 + TH + 1 <= a.Hi &&
           c.x0 + TW + 1 <= a.Wi;
  };

  // ---- staging.  Addresses are a UNIFORM base plus a 32-bit per-lane offset (global_load ... saddr): no 64-bit vector
  // arithmetic per load.  Interior tiles: base = the tile's first box voxel, offset = goff[j].  Border tiles: base = the
  // sample, offset = (tile offset inside the sample, possibly negative) + goff[j] where the voxel exists, else 0 -- and a
  // bit of `mask` remembers that the slot is zero padding (applied at commit time, after the transform).
  const char* const in_b = reinterpret_cast<const char*>(a.in);
  const int64_t sample_bytes = (int64_t)a.Di * a.Hi * a.Wi * a.in_pitch * ESZ;  // < 2^31 (launcher)
  // A staged tile is described by three uniform scalars: base pointer, byte offset of its first box voxel inside the sample
  // (border tiles) and `rng` = the box range that lies inside the volume, one nibble each: zlo | zn << 4 | ylo << 8 |
  // yn << 12 | xlo << 16 | xn << 20, bit 31 = interior tile (no test).  An invalid tile (past the end of the pass) has
  // rng = 0: every slot reads the first voxel of the tensor and is committed as zeros nobody reads.
  // (Plain scalars on purpose: as a struct passed by reference they stayed in scratch memory in the 128-byte-row
  // instantiation -- and took the staging registers and the masks with them.)
  auto src_base = [&](const WsTile& c, bool valid, bool inter) __attribute__((always_inline)) -> const char* {
    const int64_t tile_off = ((((int64_t)(c.z0 - 1)) * a.Hi + (c.y0 - 1)) * a.Wi + (c.x0 - 1)) * a.in_pitch * ESZ;
    return !valid ? in_b : in_b + (int64_t)c.n * sample_bytes + (inter ? tile_off : 0);
  };
  auto src_toff = [&](const WsTile& c, bool valid, bool inter) __attribute__((always_inline)) {
    const int64_t tile_off = ((((int64_t)(c.z0 - 1)) * a.Hi + (c.y0 - 1)) * a.Wi + (c.x0 - 1)) * a.in_pitch * ESZ;
    return (valid && !inter) ? (int)tile_off : 0;
  };
  auto src_rng = [&](const WsTile& c, bool valid, bool inter) __attribute__((always_inline)) {
    if (!valid) return 0u;
    if (inter) return 0x80000000u;
    const int zlo = max(0, 1 - c.z0), zn = min(BD, a.Di + 1 - c.z0) - zlo;
    const int ylo = max(0, 1 - c.y0), yn = min(BH, a.Hi + 1 - c.y0) - ylo;
    const int xlo = max(0, 1 - c.x0), xn = min(BW, a.Wi + 1 - c.x0) - xlo;
    return (uint32_t)(zlo | (zn << 4) | (ylo << 8) | (yn << 12) | (xlo << 16) | (xn << 20));
  };
  auto slot_ok = [&](uint32_t f, uint32_t rng) __attribute__((always_inline)) {
    const unsigned bx = f & 15u, by = (f >> 4) & 15u, bz = (f >> 8) & 15u;
    const bool in = ((bz - (rng & 15u)) < ((rng >> 4) & 15u)) & ((by - ((rng >> 8) & 15u)) < ((rng >> 12) & 15u)) &
                    ((bx - ((rng >> 16) & 15u)) < ((rng >> 20) & 15u));
    return in | ((int32_t)rng < 0);  // bitwise: no short-circuit branches
  };
  // Staging is LDS-DMA: global_load_lds_dwordx4 moves 16 bytes per lane from global memory straight into the tile buffer
  // (lane-linear: a wave writes 1 KB, a staging round of the workgroup 4 KB).  No staging registers -- the register-staged
  // first version of this kernel needed 10 x 4 of them beside the 216 weight registers and hipcc spilled weights --, no
  // commit instructions, and all NJ rounds of the next tile are in flight from the first groups of a phase on (76 KB per
  // CU instead of 40).  The instruction is inline asm and counted by hand (s_waitcnt vmcnt): issued through the builtin,
  // hipcc waits for vmcnt(0) in front of the next LDS read of ANY buffer.  M0 = LDS destination of the wave.
  const char* const zero_src = reinterpret_cast<const char*>(g_wr_zero_line);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  // issue round j of the tile (base, toff, rng) into the buffer at byte offset bufoff; returns the round's mask bit
  auto dma_one = [&](auto fast_tag, int j, const char* base, int toff, uint32_t rng, int bufoff) __attribute__((always_inline)) {
    const uint32_t f = slot_field(j);
    const char* p;
    uint32_t bit = 0u;
    if constexpr (decltype(fast_tag)::value) {
      p = base + slot_goff(f);
    } else {
      const bool ok = slot_ok(f, rng);
      p = ok ? base + (int64_t)toff + slot_goff(f) : zero_src;   // zero padding: a 16-byte line of zeros
      bit = ok ? (1u << j) : 0u;
    }
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)bufoff + (uint32_t)(j * 4096) + (uint32_t)(wave * 1024));
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(p), "s"(m0v)
                 : "memory");
    return bit;
  };
  float sc[EPC], sh[EPC];
  int xf_n = -1;
  auto load_xf = [&](int n) __attribute__((always_inline)) {  // uniform; this thread's 8 channels (chunk `part`) for every slot
    if constexpr (XF) {
      if (n != xf_n) {
        const f32x4* ps = reinterpret_cast<const f32x4*>(a.in_scale + (int64_t)n * CIN + part * EPC);
        const f32x4* ph = reinterpret_cast<const f32x4*>(a.in_shift + (int64_t)n * CIN + part * EPC);
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
          const f32x4 u = ps[e / 4], v = ph[e / 4];
#pragma unroll
          for (int k = 0; k < 4; k++) sc[e + k] = u[k], sh[e + k] = v[k];
        }
        xf_n = n;
      }
    }
  };
  // XF: the producer's InstanceNorm + ReLU, applied IN PLACE to round j once it has landed: thread (voxel, chunk `part`)
  // rewrites the 16 bytes at the chunk's swizzled position of its voxel's row -- written by a lane of its own wave (the
  // 8 or 4 lanes of a voxel row share a wave), so the wave's own vmcnt wait is all the ordering needed.  `landed` =
  // DMA rounds of this tile issued after j (they retire in order).  Rounds of zero padding stay zero (mask).
  auto xf_one = [&](auto fast_tag, int j, uint32_t mask, int bufoff, int younger) __attribute__((always_inline)) {
    if constexpr (XF) {
      const uint32_t f = slot_field(j);
      const int vox = (int)(((f >> 8) & 15u) * (BH * BW) + ((f >> 4) & 15u) * BW + (f & 15u));
      char* const q = lds + bufoff + vox * RB + ((int)(f >> 12) << 4);
      wr_wait_vmcnt(younger);
      asm volatile("" ::: "memory");
      u32x4 v = *reinterpret_cast<const u32x4*>(q);
      float fl[EPC];
      ST<T>::unpack(v, fl);
#pragma unroll
      for (int e = 0; e < EPC; e++) fl[e] = fmaxf(fl[e] * sc[e] + sh[e], relu_lo);
      u32x4 w = ST<T>::pack(fl);
      if constexpr (!decltype(fast_tag)::value) {
        const uint32_t keepm = (uint32_t)(((int32_t)(mask << (31 - j))) >> 31);  // all ones where the voxel exists
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] &= keepm;
      }
      // threads past the end of the box (last round) must not touch it: their field is the box's last voxel, and a
      // second read-modify-write of the same 16 bytes would transform them twice
      if (j < NJ - 1 || NJ * VPS == BOX || VPS * j + tv < BOX) *reinterpret_cast<u32x4*>(q) = w;
    }
  };

  // T0: tile under the MFMAs; T1: the tile landing in the other buffer
  WsTile T0, T1, T2;  // T2: the tile after T1, decoded under the MFMAs of T0 (scalar work off the loop's serial tail)
  const char *base1 = in_b, *base2 = in_b;
  int toff1 = 0, toff2 = 0;
  uint32_t rng1 = 0, rng2 = 0, m1 = 0;
  bool v1 = false, v2 = false;
  int left = 0, par = 0;
  using Yes = std::true_type;
  using No = std::false_type;
  auto begin_pass = [&](auto border_tag, int k, int count) __attribute__((always_inline)) {
    constexpr bool BORDER = decltype(border_tag)::value;
    if constexpr (BORDER)
      bor_init(T0, k);
    else
      int_init(T0, k);
    left = count - 1;
    T1 = T0;
    if constexpr (BORDER)
      bor_next(T1);
    else
      int_next(T1);
    v1 = left >= 1;
    const bool i0 = BORDER ? tile_interior(T0) : true, i1 = v1 && (BORDER ? tile_interior(T1) : true);
    const char* const base0 = src_base(T0, true, i0);
    const int toff0 = src_toff(T0, true, i0);
    const uint32_t rng0 = src_rng(T0, true, i0);
    base1 = src_base(T1, v1, i1), toff1 = src_toff(T1, v1, i1), rng1 = src_rng(T1, v1, i1);
    T2 = T1;
    __syncthreads();  // the previous pass is done with both tile buffers
    par = 0;
    load_xf(T0.n);
    // T0 -> buffer 0, not overlapped
    uint32_t m0 = 0;
#pragma unroll
    for (int j = 0; j < NJ; j++) m0 |= dma_one(No{}, j, base0, toff0, rng0, 0);
#pragma unroll
    for (int j = 0; j < NJ; j++) xf_one(No{}, j, m0, 0, NJ - 1 - j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    m1 = 0;
    WS_BARRIER();
  };

  const int ch = n0 + r;
  bool ch_ok[NB];
  float bias[NB];
#pragma unroll
  for (int nb = 0; nb < NB; nb++) {
    ch_ok[nb] = ch + 32 * nb < a.Cout;
    bias[nb] = (a.bias && ch_ok[nb]) ? a.bias[ch + 32 * nb] : 0.f;
  }
  T* outp[NB];
#pragma unroll
  for (int nb = 0; nb < NB; nb++)
    outp[nb] = (a.split && n0 + 32 * nb >= a.split) ? reinterpret_cast<T*>(a.out2) - a.split : reinterpret_cast<T*>(a.out);
  // accumulator register i of this lane: voxel (dz, y, x) of its M-block with x = (i & 3) + 4 ((i >> 2) & 1) and dz from
  // ws_row_to_zx of MFMA row (i & 3) + 8 (i >> 2) + 4 h
  int edz[4], eplane[4];
#pragma unroll
  for (int q4 = 0; q4 < 4; q4++) {
    int dz, x;
    ws_row_to_zx(8 * q4 + 4 * h, dz, x);
    edz[q4] = dz;
    eplane[q4] = dz * a.Ho * a.Wo * (int)a.out_pitch;
  }
  // element offset of accumulator register i inside its M-block: plane part per lane (edz depends on h), x part uniform
  int opitch_l = (int)a.out_pitch;
  auto eoff = [&](int i) __attribute__((always_inline)) { return eplane[i >> 2] + ((i & 3) + 4 * ((i >> 2) & 1)) * opitch_l; };
  const int ykeep = ybase + (KSPLIT == 2 ? KEEP * kq : 0);  // first y row this wave stores
  const bool ch_odd = r & 1;

  // InstanceNorm partial sums: per lane over the voxels it has produced, combined through LDS when the sample changes or
  // a pass ends (conv_ws2_kernel's scheme and row geometry: 2 passes x 256 workgroup slots per sample)
  float lr1[NB], lr2[NB];
#pragma unroll
  for (int nb = 0; nb < NB; nb++) lr1[nb] = lr2[nb] = 0.f;
  int racc_n = -1, cur_pass = 0;
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
  auto stats_sample = [&](int n) __attribute__((always_inline)) {
    if (n != racc_n) {
      if (racc_n >= 0) stats_to_row();
      racc_n = n;
    }
  };

  // PLAIN (compile-time): the launch is plain and the tile is whole.  Branch-free on purpose: vmcnt counts stores too and
  // retires in order, and hipcc derives the N of the next phase's `s_waitcnt vmcnt(N)` (first commit: the oldest prefetch
  // load) from the path with the FEWEST younger operations -- with a conditional store path it assumes none, emits
  // vmcnt(R - 2) and thereby waits for every prefetch load, the one issued a moment ago included, and for the write
  // acknowledgements of this tile's stores: a full memory round trip at the top of every tile.
  auto epilogue = [&](auto& fin, const WsTile& ET, auto plain_tag) __attribute__((always_inline)) {
    constexpr bool PLAIN = decltype(plain_tag)::value;
    const int z0 = ET.z0 + 4 * zb, y0 = ET.y0 + ykeep, x0 = ET.x0;
    opitch_l = (int)a.out_pitch;
    asm volatile("" : "+s"(opitch_l));
    const bool full = PLAIN || (ET.z0 + TD <= a.Do && ET.y0 + TH <= a.Ho && ET.x0 + TW <= a.Wo);
    stats_sample(ET.n);
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      float s1 = 0.f, s2 = 0.f;
      T* const obase = outp[nb] + ((((int64_t)ET.n * a.Do + z0) * a.Ho + y0) * a.Wo + x0) * a.out_pitch + ch + 32 * nb;
      if constexpr (PLAIN) {
        // two accumulator rows (voxels x, x + 1 of one channel) leave as ONE dword per lane after a DPP exchange with the
        // neighbouring channel's lane (st_rows2): half the store instructions -- 8 per M-block -- and the tile's stores plus
        // the prefetch loads in flight stay below the 63 operations vmcnt can count (with 64 two-byte stores per wave the
        // 8x8x8 tile stalled in its epilogue until the loads issued a moment earlier had landed).  The InstanceNorm sums
        // run as four independent chains: one chain of 16 KEEP dependent adds was ~1,000 cycles per tile.
        float p1[4] = {0.f, 0.f, 0.f, 0.f}, p2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mb = 0; mb < KEEP; mb++) {
          T* const orow = obase + (int64_t)mb * a.Wo * a.out_pitch - (ch_odd ? 1 : 0);
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const float v0 = fin[mb][nb][i] + bias[nb], v1 = fin[mb][nb][i + 1] + bias[nb];
#ifdef WR_DBG_NOSTORE  // attribution build: one store per M-block instead of eight
            if (i == 0)
#endif
            st_rows2<T>(orow + eoff(ch_odd ? i + 1 : i), v0, v1, ch_odd);
            p1[(i >> 1) & 3] += v0 + v1;
            p2[(i >> 1) & 3] += v0 * v0 + v1 * v1;
          }
        }
        s1 = (p1[0] + p1[1]) + (p1[2] + p1[3]);
        s2 = (p2[0] + p2[1]) + (p2[2] + p2[3]);
      } else if (full && ch_ok[nb] && !a.accumulate) {
#pragma unroll
        for (int mb = 0; mb < KEEP; mb++) {
          T* const orow = obase + (int64_t)mb * a.Wo * a.out_pitch;
#pragma unroll
          for (int i = 0; i < 16; i++) {
            const float v = fin[mb][nb][i] + bias[nb];
            ST<T>::st(orow + eoff(i), v);
            s1 += v;
            s2 += v * v;
          }
        }
      } else {
#pragma unroll
        for (int mb = 0; mb < KEEP; mb++) {
          const int gy = y0 + mb;
          T* const orow = obase + (int64_t)mb * a.Wo * a.out_pitch;
#pragma unroll
          for (int i = 0; i < 16; i++) {
            const int gz = z0 + edz[i >> 2], gx = x0 + (i & 3) + 4 * ((i >> 2) & 1);
            const float v = fin[mb][nb][i] + bias[nb];
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

  // DEFERRED EPILOGUE (plain phases).  With one wave per SIMD the stores, the 64-bit address arithmetic and the
  // InstanceNorm sums of a tile -- 15 % of a workgroup's cycles behind the MFMA phase of the 8x8x8 tile, and as much
  // again in the tail that hipcc sinks the sums into -- run with the matrix pipe idle.  Instead the kept accumulators
  // are copied to `pend` when a tile's phase ends and leave during the NEXT tile's phase, NPAIR paired-row stores spread
  // over its NG groups.  A pass starts with pend = 0 on its own first tile: the zeros it stores there are overwritten by
  // that tile's real values one phase later (same wave, same addresses, program order) and add nothing to the sums -- so
  // the phase never branches on "is there a pending tile".  The pass ends with an immediate epilogue of the last tile.
#ifdef WR_PLAIN_STORES
#define WR_STORES_PER_PIECE 2
#else
#define WR_STORES_PER_PIECE 1
#endif
  constexpr bool DEFER = KSPLIT == 2 && NB == 1;
  constexpr int NPAIR = KEEP * NB * 8;
  f32x16 pend[KEEP][NB];
  WsTile PT;
  T* pbase[NB];
  auto pend_clear = [&](const WsTile& c) __attribute__((always_inline)) {
#pragma unroll
    for (int mb = 0; mb < KEEP; mb++)
#pragma unroll
      for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int i = 0; i < 16; i++) pend[mb][nb][i] = 0.f;
    PT = c;
  };
  auto pend_begin = [&]() __attribute__((always_inline)) {
    stats_sample(PT.n);
    opitch_l = (int)a.out_pitch;
    asm volatile("" : "+s"(opitch_l));
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
      pbase[nb] = outp[nb] + ((((int64_t)PT.n * a.Do + PT.z0 + 4 * zb) * a.Ho + PT.y0 + ykeep) * a.Wo + PT.x0) * a.out_pitch +
                  ch + 32 * nb - (ch_odd ? 1 : 0);
  };
  auto epi_piece = [&](int k) __attribute__((always_inline)) {
    const int mb = k / (NB * 8), nb = (k / 8) % NB, i = 2 * (k % 8);
    const float v0 = pend[mb][nb][i], v1 = pend[mb][nb][i + 1];
#ifdef WR_PLAIN_STORES  // A/B build: two 2-byte stores per piece instead of the paired dword store
    T* const q0 = pbase[nb] + (ch_odd ? 1 : 0) + (int64_t)mb * a.Wo * opitch_l;
    ST<T>::st(q0 + eoff(i), v0);
    ST<T>::st(q0 + eoff(i + 1), v1);
#else
    st_rows2<T>(pbase[nb] + (int64_t)mb * a.Wo * opitch_l + eoff(ch_odd ? i + 1 : i), v0, v1, ch_odd);
#endif
    lr1[nb] += v0 + v1;
    lr2[nb] += v0 * v0 + v1 * v1;
  };
  auto pend_flush = [&]() __attribute__((always_inline)) {
    pend_begin();
#pragma unroll
    for (int k = 0; k < NPAIR; k++) epi_piece(k);
  };
  // VM operations a phase issues behind its last DMA round: the stores of the pieces scheduled in or after that group
  // (within a group: DMA rounds, transforms, then pieces).  vmcnt retires in order, so waiting until at most that many are
  // outstanding means every round of T1 has landed.
  constexpr int G_LAST_DMA = ((NJ - 1) * (NG / 2)) / NJ;
  constexpr int N_AFTER_DMA = [] {
    int n = 0;
    for (int k = 0; k < KEEP * NB * 8; k++)
      if ((k * NG) / (KEEP * NB * 8) >= G_LAST_DMA) n++;
#ifdef WR_PLAIN_STORES
    n *= 2;
#endif
    return n;
  }();

  // One tile.  Group g = (jz, jx, k-step): its six A fragments were read during group g - 1; in the gaps of its 12 NB
  // MFMAs the stream commits the slots of T1 scheduled for g (registers -> transform -> LDS, other buffer) and reloads
  // each freed register: first half of the phase with the second half of T1's slots, second half with T2's first half
  // (a load has half a phase, ~1.7 us, to land).
  // KQ (compile-time copy of kq): accumulator a of the wave is M-block (a + KEEP KQ) % 4, so that in BOTH K-halves the
  // accumulators a wave keeps are acc[0 .. KEEP) and the ones it sends acc[KEEP .. 4).  The two halves run two copies of
  // the whole tile loop under one wave-uniform branch: written as `kq ? acc[0] : acc[2]` inside one copy, hipcc turned
  // the accumulators into a dynamically indexed array in scratch memory (a scratch store of all 64 after every group).
#ifdef WR_DBG_STAMPS  // diagnostic build: s_memtime per section of the tile loop, summed per workgroup (wave 0)
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
  const unsigned long long tcyc0 = tlast, treal0 = __builtin_amdgcn_s_memrealtime();
#define WR_STAMP(k)                                        \
  {                                                        \
    __builtin_amdgcn_sched_barrier(0);                     \
    unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
    __builtin_amdgcn_s_waitcnt(0xC07F);                    \
    __builtin_amdgcn_sched_barrier(0);                     \
    tacc[k] += t_ - tlast;                                 \
    tlast = t_;                                            \
  }
#else
#define WR_STAMP(k)
#endif
  f32x16 acc[MBW][NB];
  auto tile_phase = [&](auto fast_tag, auto plain_tag, auto kq_tag) __attribute__((always_inline)) {
    constexpr int KQ = decltype(kq_tag)::value;
    // (free at run time: keeps the allocator from choosing the 54 weight fragments -- long live ranges, one use per
    // tile each -- as the values to spill when the phase needs registers)
#pragma unroll
    for (int tap = 0; tap < 27; tap++)
#pragma unroll
      for (int ks = 0; ks < KSW; ks++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++) asm volatile("" : "+a"(wreg[tap][ks][nb]));
    constexpr bool PLAINP = decltype(plain_tag)::value;
    // the epilogue is deferred into the next tile's phase only where its 16 KEEP NB pending registers fit beside the phase's
    // own (instantiation A); the 8x8x8 tile (64 accumulator + 64 pending registers) spilled and runs it immediately
    constexpr bool DEFERP = PLAINP && DEFER;
    constexpr bool BORDERP = !decltype(fast_tag)::value;
    char* const a_rd = lds + par * ABUF;
    const int a_wr = (1 - par) * ABUF;
    WR_STAMP(7)
    if (v1) load_xf(T1.n);
    if constexpr (DEFERP) pend_begin();
    // the tile after T1, for the next phase (uniform scalar work, here under the MFMAs instead of between two phases)
    T2 = T1;
    if constexpr (BORDERP)
      bor_next(T2);
    else
      int_next(T2);
    v2 = left >= 2;
    {
      const bool i2 = v2 && (BORDERP ? tile_interior(T2) : true);
      base2 = src_base(T2, v2, i2), toff2 = src_toff(T2, v2, i2), rng2 = src_rng(T2, v2, i2);
    }
    u32x4 af[2][6];
    auto read_group = [&](int g, u32x4 (&A)[6]) __attribute__((always_inline)) {
      const int t = g / KSW, ks = g % KSW, jz = t / 3, jx = t % 3;
      // (laundered: left visible, hipcc hoists the 9 KSW xor-ed addresses out of the tile loop into registers of their
      // own and spills some -- and a scratch reload inside the phase is an s_waitcnt vmcnt(0), a drain of the DMA queue)
      int g0 = ga[jz][jx];
      asm volatile("" : "+v"(g0));
      const char* p = a_rd + (g0 ^ (ks * 32));
#pragma unroll
      for (int yp = 0; yp < 6; yp++) A[yp] = *reinterpret_cast<const u32x4*>(p + yp * BW * RB);
    };
    read_group(0, af[0]);
#pragma unroll
    for (int g = 0; g < NG; g++) {
      if (g + 1 < NG) read_group(g + 1, af[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);  // look-ahead reads stay ABOVE this group's MFMAs
#ifndef WR_DBG_NOSTAGE  // attribution build (wrong results): no staging at all
      // first half of the phase: the DMA rounds of T1 (other buffer), NJ of them spread over NG / 2 groups; second half
      // (XF only): the in-place transform of each round once it has landed
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        if ((j * (NG / 2)) / NJ == g) m1 |= dma_one(fast_tag, j, base1, toff1, rng1, a_wr);
        if (XF && NG / 2 + (j * (NG - NG / 2)) / NJ == g) {
          // younger than round j: the rounds after it and the stores of the pieces between its group and this one
          int younger = NJ - 1 - j;
          if (DEFERP)
            for (int k = 0; k < NPAIR; k++)
              if ((k * NG) / NPAIR >= (j * (NG / 2)) / NJ && (k * NG) / NPAIR < g) younger += WR_STORES_PER_PIECE;
          xf_one(fast_tag, j, m1, a_wr, younger);
        }
      }
#endif
      if constexpr (DEFERP) {
#pragma unroll
        for (int k = 0; k < NPAIR; k++)
          if ((k * NG) / NPAIR == g) epi_piece(k);
      }
      const int t = g / KSW, ks = g % KSW, jz = t / 3, jx = t % 3;
      u32x4(&A)[6] = af[g & 1];
      if (g == 0) {  // the first MFMAs of the tile take a zero C operand
#pragma unroll
        for (int mb = 0; mb < MBW; mb++)
#pragma unroll
          for (int nb = 0; nb < NB; nb++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[mb][nb][i] = 0.f;
      }
#pragma unroll
      for (int jy = 0; jy < 3; jy++)
#pragma unroll
        for (int ai = 0; ai < MBW; ai++)
#pragma unroll
          for (int nb = 0; nb < NB; nb++)
#ifdef WR_DBG_NOMFMA  // attribution build: everything but the matrix instructions (one per group keeps the operands live)
            if (jy == 0 && ai == 0)
#endif
            Mma<T>::run(A[(ai + KEEP * KQ) % MBW + jy], wreg[jz * 9 + jy * 3 + jx][ks][nb], acc[ai][nb]);
      // spread the group's staging work over its MFMA gaps, two instructions per gap (left alone hipcc puts the whole
      // commit + address arithmetic + load -- ~100 issue cycles -- between the first two MFMAs and the matrix pipe idles)
#pragma unroll
      for (int k = 0; k < 12 * NB; k++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002 | 0x260, DEFERP ? 3 : 2, 0);  // of: VALU, VMEM, DS write
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // T1 has landed (and, with a transform, been rewritten).  vmcnt counts stores too and retires in order: the previous
    // tile's stores were issued before this phase's DMA rounds and are a whole phase old.
    wr_wait_vmcnt(DEFERP ? N_AFTER_DMA : 0);
    asm volatile("" ::: "memory");
    WR_STAMP(0)
    WS_BARRIER();  // buffer `par` fully read, the other one fully written
    WR_STAMP(1)
    if constexpr (KSPLIT == 2) {
      // sum the two K halves: a wave keeps acc[0 .. KEEP) and sends acc[KEEP .. 4) -- the M-blocks its partner (wave ^ 1)
      // keeps, in the partner's order -- through LDS, 1 KB per ds_write_b128 / ds_read_b128, lane-linear
      f32x4* const xs = reinterpret_cast<f32x4*>(XCH_ALIAS ? a_rd : lds + OFF_XCH);
      f32x4* const mine = xs + wave * (KEEP * NB * 4 * 64) + lane;
      const f32x4* const theirs = xs + (wave ^ 1) * (KEEP * NB * 4 * 64) + lane;
#pragma unroll
      for (int mb = 0; mb < KEEP; mb++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const f32x16& c = acc[KEEP + mb][nb];
            mine[((mb * NB + nb) * 4 + q) * 64] = f32x4{c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
          }
      WR_STAMP(2)
      WS_BARRIER();
      WR_STAMP(3)
#pragma unroll
      for (int mb = 0; mb < KEEP; mb++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const f32x4 v = theirs[((mb * NB + nb) * 4 + q) * 64];
#pragma unroll
            for (int k = 0; k < 4; k++) acc[mb][nb][4 * q + k] += v[k];
          }
      WR_STAMP(4)
      if constexpr (XCH_ALIAS) WS_BARRIER();  // the scratch is the next phase's staging target
      WR_STAMP(5)
    }
    if constexpr (DEFERP) {
#pragma unroll
      for (int mb = 0; mb < KEEP; mb++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
#pragma unroll
          for (int i = 0; i < 16; i++) pend[mb][nb][i] = acc[mb][nb][i] + bias[nb];  // (bias here: the zeros of a pass's start stay zeros)
      PT = T0;
    } else {
      epilogue(acc, T0, plain_tag);
    }
    WR_STAMP(6)
    par ^= 1;
  };

  bool more = true;
  auto step = [&]() __attribute__((always_inline)) {
    more = v1;
    if (!more) return;
    left--;
    T0 = T1;
    T1 = T2;
    v1 = v2;
    base1 = base2, toff1 = toff2, rng1 = rng2;
    m1 = 0;
  };

  auto run = [&](auto kq_tag) __attribute__((always_inline)) {
    if (int_cnt > 0) {
      begin_pass(No{}, int_begin, int_cnt);  // (interior tiles exist only in plain launches, and they are whole)
      if constexpr (DEFER) pend_clear(T0);
      more = true;
      while (more) {
        tile_phase(Yes{}, Yes{}, kq_tag);
        step();
      }
      if constexpr (DEFER) pend_flush();  // the pass's last tile
    }
    if (bor_cnt > 0) {
      begin_pass(Yes{}, bor_begin, bor_cnt);
      stats_to_row();
      cur_pass = 1;
      more = true;
      if (all_full) {
        if constexpr (DEFER) pend_clear(T0);
        while (more) {
          tile_phase(No{}, Yes{}, kq_tag);
          step();
        }
        if constexpr (DEFER) pend_flush();
      } else {
        while (more) {
          tile_phase(No{}, No{}, kq_tag);
          step();
        }
      }
    }
    stats_to_row();
  };
  if constexpr (KSPLIT == 2) {
    if (kq == 0)
      run(std::integral_constant<int, 0>{});
    else
      run(std::integral_constant<int, 1>{});
  } else {
    run(std::integral_constant<int, 0>{});
  }
#ifdef WR_DBG_STAMPS
  // wave 0 and wave 1 (the two K halves) of every workgroup: cycles per section, then total cycles and real time (10 ns)
  if (lane == 0 && wave < 2 && blockIdx.y == 0 && a.stat_partials) {
    float* q = a.stat_partials + ((int64_t)blockIdx.x * 2 + wave) * 16;
    for (int k = 0; k < 8; k++) q[k] = (float)tacc[k];
    q[8] = (float)(__builtin_amdgcn_s_memtime() - tcyc0);
    q[9] = (float)(__builtin_amdgcn_s_memrealtime() - treal0);
  }
#endif
}

template <typename T, int RB, int NB, int KSPLIT, int TD>
int launch_wr(const ConvArgs& a, hipStream_t st) {
  const int tiles = a.N * ceil_div(a.Do, TD) * ceil_div(a.Ho, 8) * ceil_div(a.Wo, 8);
  const int cout_tiles = a.CoutP / (32 * NB);
  const int gx = std::min(tiles, std::max(1, hdf_cu_budget() / cout_tiles));
  // (round 5: the only instantiation per storage type.  The 64-byte-row forms <64, 2, 2, 4> / <64, 1, 1, 8> and the
  // forms with the producer's InstanceNorm + ReLU applied on load were built, parity-tested and measured level with or
  // behind conv_ws2 in rounds 4-5 and are no longer compiled; the template keeps their parameters.)
  hipLaunchKernelGGL((conv_wr_kernel<T, RB, NB, KSPLIT, TD, false>), dim3(gx, cout_tiles), dim3(256), 0, st, a);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

}  // namespace

// Does the weights-in-registers kernel take this mode-0 launch?  (16-bit storage, 64- or 128-byte rows, a volume the old
// weights-stationary kernel would take, a sample below 2 GiB so that 32-bit byte offsets address it.)
// Measured against conv_ws2_kernel on one box (tools/conv_ab.sh, round 4): 64 -> 32 @128^3 without an input transform
// 482 vs 502 us; with a transform, at 64^3 and for 64-byte rows it was level or behind, so only
// that class exists.  hdf_op_conv3d_wr runs any eligible shape (>= 48^3) through it (tests, tools).
bool hdf_conv_wr_takes(int dtype, const ConvArgs& a) {
  return hdf_conv_wr_can(dtype, a) && (int64_t)a.Do * a.Ho * a.Wo >= 96 * 96 * 96;
}

bool hdf_conv_wr_can(int dtype, const ConvArgs& a) {
  if (dtype == HDF_F32 || a.wfrag || a.Di == 1) return false;   // (depth 1: the 2-D operators of conv_igemm.hip)
  const int rb = a.Cin * 2;
  if (rb != 128 || a.in_scale) return false;
  if ((int64_t)a.Do * a.Ho * a.Wo < 48 * 48 * 48) return false;
  if ((int64_t)a.Di * a.Hi * a.Wi * a.in_pitch * 2 >= ((int64_t)1 << 31)) return false;
  if ((int64_t)a.Hi * a.Wi * a.in_pitch * 2 >= (1 << 24)) return false;  // 24-bit multiplies by the z stride
  if (a.Di != a.Do || a.Hi != a.Ho || a.Wi != a.Wo) return false;
  return true;
}

int hdf_launch_conv_wr(int dtype, const ConvArgs& a, hipStream_t st) {
  const int rb = a.Cin * 2;
  if (dtype == HDF_BF16) {
    using T = bf16_t;
    if (rb == 128 && !a.in_scale) return launch_wr<T, 128, 1, 2, 4>(a, st);
  } else if (dtype == HDF_F16) {
    using T = f16_t;
    if (rb == 128 && !a.in_scale) return launch_wr<T, 128, 1, 2, 4>(a, st);
  }
  hdf_set_error("conv_wr: dtype %d, %d-byte rows%s: not built", dtype, rb, a.in_scale ? " with an input transform" : "");
  return HDF_ERR_UNSUPPORTED;
}
