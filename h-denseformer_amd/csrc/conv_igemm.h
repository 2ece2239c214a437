// Argument blocks of the implicit-GEMM 3x3x3 convolution family (see conv_igemm.hip).
#pragma once
#include "hdf_common.h"

// Pitched channels-last view: voxel (n,d,h,w) channel c lives at p + ((((n*D+d)*H+h)*W+w)*pitch + c)
// elements.  `p` already points at the first channel of the view, so a channel slice of a wider
// buffer (e.g. one half of a concat buffer) is just a different p with the same pitch.
struct ConvArgs {
  const void* in;
  int64_t in_pitch;
  int Cin;  // multiple of 16
  int N, Di, Hi, Wi;
  int Do, Ho, Wo;
  const void* w;        // packed [27][CoutP][Cin] storage type, CoutP = round_up(Cout,32), zero padded
  const float* bias;    // [Cout] or null
  const float* in_scale;  // [N][Cin] or null: x -> x*scale+shift (then relu if in_relu)
  const float* in_shift;
  int in_relu;
  void* out;
  int64_t out_pitch;
  int Cout;
  int CoutP;
  float* stat_partials;  // [N*tiles][CoutP][2] (sum, sum of squares) or null
  int accumulate;        // out += result
  void* out2;            // split output (mode 0): channels >= split go to out2[...][ch - split], same pitch as out
  int split;             // multiple of 32, 0 = single output
  int wfrag;             // weight layout: 0 = [27][CoutP][Cin] rows, 1 = fragment-major (hdf_conv_weight_layout)
  // split-K scratch (optional, mode 0): the launcher may split the input-channel chunks of a low-resolution layer over up
  // to 4 workgroups per tile (fp32 partial tiles here, summed in a fixed order by conv_ksplit_reduce_kernel)
  float* kpart;
  size_t kpart_bytes;
  int ksplit;            // set by the launcher
  // Backward statistics (optional, mode 0, 16-bit storage, whole 4x8x8 tiles, CoutP == Cout, no accumulate / split): the
  // launch computes the gradient w.r.t. the ACTIVATION relu(IN(bs_y)) of the layer below, and stat_partials receives the
  // first pass of that InstanceNorm(+ReLU)'s backward instead of (sum, sum of squares): per channel (sum g, sum g * xhat),
  // g = the STORED output where bs_y * bs_scale + bs_shift > 0, xhat = (bs_y - bs_mean) * bs_rstd -- the rows
  // hdf_launch_in_bwd_reduce would write (WS_STAT_ROWS per sample), without its pass over the two tensors.
  const void* bs_y;
  int64_t bs_y_pitch;
  const float *bs_scale, *bs_shift, *bs_mean, *bs_rstd;   // [N][Cout]
  // raised wave priority (HDF_LIGHT_PRIO, hdf_common.h): the launch belongs to a latency-bound chain that runs next to
  // another stream's persistent kernels (the UpConv chain on the plan's branch stream)
  int prio = 0;
  // workgroups of a persistent launch (conv_ws2_kernel), 0 = hdf_cu_budget(): a launch that should leave compute units to
  // another stream's chain asks for fewer
  int cu_budget = 0;
};
// split-K scratch a plan keeps per stream
constexpr size_t HDF_KSPLIT_BYTES = (size_t)16 << 20;

struct WgradArgs {
  // D[tap][sc][lc] = sum_{n,i} S[n,i][sc] * L[n, STRIDE*i - 1 + tap][lc]
  const void* sm;
  int64_t sm_pitch;
  int SC;  // channels of the small tensor (multiple of 16)
  const void* lg;
  int64_t lg_pitch;
  int LC;
  int N, Ds, Hs, Ws, Dl, Hl, Wl;
  const float* sm_scale;  // [N][SC] or null
  const float* sm_shift;
  int sm_relu;
  const float* lg_scale;  // [N][LC] or null
  const float* lg_shift;
  int lg_relu;
  float* partials;  // [G][27][SCp][LCp]
  int SCp, LCp;
  int tiles_per_group, num_tiles;
  // Fused InstanceNorm(+ReLU) backward (optional; 16-bit stride-1 launches that hdf_wgrad_apply_takes accepts): `sm` is then
  // the gradient w.r.t. the layer's ACTIVATION relu(IN(y)), and the kernel applies the second pass of that norm's backward
  // (in_bwd_elem, hdf_common.h: the arithmetic and storage rounding of hdf_launch_in_bwd_apply) to the rows it stages --
  // every voxel of the small operand is staged exactly once per (small, large) channel-block pair -- and the workgroups of
  // large-channel block 0 store the result to ap_out: the dy the data-gradient conv reads afterwards.  The stand-alone
  // pass (read da, read y, write dy: three tensors through HBM at its roof) and the weight gradient's own read of dy go.
  const void* ap_y = nullptr;  // raw conv output of the layer, [N][vox][ap_y_pitch]
  int64_t ap_y_pitch = 0;
  void* ap_out = nullptr;  // dy, [N][vox][ap_out_pitch]
  int64_t ap_out_pitch = 0;
  const float* ap_tab[7] = {};  // scale, shift, mean, rstd, k1, ka, kb: [N][SC] each
};
bool hdf_wgrad_apply_takes(int dtype, int stride, const WgradArgs& a);

// conv_wr.hip: weights-in-registers form of the mode-0 launches with 64- / 128-byte rows at >= 48^3 (16-bit storage)
// csrc/conv_first.hip: the encoder's first layer (1..4 real input channels, 16-bit storage) with tap-packed K.  Reads the
// fp32 torch-layout weights itself; stat_partials as hdf_launch_conv with WS_STAT_ROWS rows per sample.
bool hdf_conv_first_can(int dtype, int Cin, int Cout, int D, int H, int W, int64_t in_pitch);
bool hdf_conv_first_takes(int dtype, int Cin, int Cout, int D, int H, int W, int64_t in_pitch);
int hdf_launch_conv_first(int dtype, const void* in, int64_t in_pitch, int Cin, int N, int D, int H, int W,
                          const float* w32, const float* bias, void* out, int64_t out_pitch, int Cout,
                          float* stat_partials, hipStream_t st);
// the same layer's weight gradient (csrc/conv_first.hip): dw in the torch layout [Cout][Cin][27], fp32
bool hdf_wgrad_first_takes(int dtype, int Cin, int Cout, int D, int H, int W, int64_t x_pitch, int64_t dy_pitch);
// in_bwd (optional): `dy` is the gradient w.r.t. the layer's activation relu(IN(y)); the kernel applies the second pass of that
// InstanceNorm(+ReLU)'s backward (hdf_launch_in_bwd_apply's arithmetic and storage rounding) while it stages the rows, so
// that pass and the tensor it writes are not needed.  All vectors [N][Cout].
struct WgradFirstIn {
  const void* y;
  int64_t y_pitch;
  const float *scale, *shift, *mean, *rstd, *k1, *ka, *kb;
};
int hdf_launch_wgrad_first(int dtype, const void* dy, int64_t dy_pitch, int Cout, const void* x, int64_t x_pitch, int Cin,
                           int N, int D, int H, int W, float* dw, int accumulate, void* workspace, size_t workspace_bytes,
                           hipStream_t st, const WgradFirstIn* in_bwd = nullptr);
bool hdf_conv_wr_can(int dtype, const ConvArgs& a);    // the kernel handles this launch
bool hdf_conv_wr_takes(int dtype, const ConvArgs& a);  // ... and the plan routes it there
int hdf_launch_conv_wr(int dtype, const ConvArgs& a, hipStream_t st);
int hdf_launch_conv(int dtype, int mode /*0 conv s1, 1 conv s2, 2 convT*/, const ConvArgs& a, hipStream_t st);
// tiles per sample of the stat partials; row_bytes = Cin*sizeof(storage) selects the kernel variant (pass a
// large value for the upper bound used to size buffers)
int hdf_conv_stat_tiles(int mode, int Do, int Ho, int Wo, int row_bytes);
// can this mode-0 launch take the bs_* fields (else the caller runs hdf_launch_in_bwd_reduce)?
bool hdf_conv_bwd_stats_ok(int dtype, const ConvArgs& a);
int hdf_launch_wgrad(int dtype, int stride, WgradArgs a, float* dw, int sc_store, int lc_store, int accumulate,
                     void* workspace, size_t workspace_bytes, hipStream_t st);
size_t hdf_wgrad_workspace_bytes(int stride, int N, int Ds, int Hs, int Ws, int SC, int LC);
// every weight pack of a training step in ONE launch (42 separate 6-us launches otherwise)
struct PackJob {
  int64_t src_off;  // floats from the parameter base
  int64_t dst_off;  // bytes from the workspace base
  int O, I, OP, IP, so, si, flip;
  int frag;  // 1: fragment-major destination (hdf_conv_weight_layout)
};
// Weight layout the conv launch of this shape reads.  0: [27][CoutP][Cin] rows.  1: fragment-major, for the launches
// that take conv_igemm_kernel's pipelined path (weight fragments straight from L2): per tap, per 32-channel output
// block, per 32-byte step of the input-channel row, the 32 rows x 32 bytes one wave loads form ONE contiguous 1 KB
// block -- a fragment load touches 8 full cache lines instead of 32 lines of which it uses 32 bytes each.
int hdf_conv_weight_layout(int dtype, int mode, int Cin, int Do, int Ho, int Wo);
constexpr int HDF_MAX_PACK_JOBS = 64;
int hdf_launch_pack_batch(int dtype, const float* params, char* ws, const PackJob* jobs, int njobs, hipStream_t st);
int hdf_launch_pack_w(int dtype, const float* src, void* dst, int O, int I, int OP, int IP, int64_t so, int64_t si,
                      int flip, hipStream_t st);
