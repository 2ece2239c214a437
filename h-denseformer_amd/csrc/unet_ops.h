// HBM-bound U-Net ops around the convolutions (see unet_ops.hip).  All tensors channels-last with a
// voxel pitch; storage bf16 or f32 (dtype enum), statistics and parameters fp32.
#pragma once
#include "hdf_common.h"

// x [N,C,D,H,W] fp32 (the reference's input layout) -> [N,D,H,W,CP] storage type, channels >= C zero
int hdf_launch_nchw_to_ndhwc(int dtype, const float* x, void* out, int N, int C, int CP, int64_t vox, hipStream_t st);

// (sum,sumsq) partials [N][tiles][CP][2] -> per-(n,c) mean, rstd, scale = gamma*rstd, shift = beta - mean*scale
int hdf_launch_in_finalize(const float* partials, int N, int tiles, int C, int CP, int64_t vox, const float* gamma,
                           const float* beta, float eps, float* mean, float* rstd, float* scale, float* shift,
                           hipStream_t st);

// out = relu(y*scale+shift) + skip   (skip may be null)
int hdf_launch_norm_relu_add(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift,
                             const void* skip, int64_t skip_pitch, void* out, int64_t out_pitch, int N, int C,
                             int64_t vox, hipStream_t st);

// fused encoder tail: ds = relu(y*scale+shift) + skip, pooled/idx = MaxPool3d(2)(ds).  If lscale != null the skip
// is trilinear_x2(relu(skip_lo*lscale+lshift)) computed on the fly from the low-resolution tensor `skip`
// (dims Do,Ho,Wo); else `skip` is a materialised full-resolution tensor.
int hdf_launch_enc_tail(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift,
                        const void* skip, int64_t skip_pitch, void* ds, int64_t ds_pitch, void* pooled,
                        int64_t pooled_pitch, uint8_t* idx, int N, int C, int Do, int Ho, int Wo, hipStream_t st,
                        int flat = 0 /* 1: the 2-D form on depth-1 tensors (MaxPool2d(2)), round 6 */);
// the same with skip = trilinear x2 of relu(low * lscale + lshift), low at (Do, Ho, Wo): the skip tensor is never materialised
int hdf_launch_enc_tail_up(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift,
                           const void* low, int64_t low_pitch, const float* lscale, const float* lshift, void* ds,
                           int64_t ds_pitch, void* pooled, int64_t pooled_pitch, uint8_t* idx, int N, int C, int Do, int Ho,
                           int Wo, hipStream_t st);
int hdf_launch_maxpool_fwd(int dtype, const void* in, int64_t in_pitch, void* out, int64_t out_pitch, uint8_t* idx,
                           int N, int C, int Do, int Ho, int Wo, hipStream_t st);
// din[8 positions] (+)= (pos == idx) ? dout : 0
int hdf_launch_maxpool_bwd(int dtype, const void* dout, int64_t dout_pitch, const uint8_t* idx, void* din,
                           int64_t din_pitch, int N, int C, int Do, int Ho, int Wo, int accumulate, hipStream_t st);

// trilinear x2, align_corners=False, of relu(y*scale+shift)
int hdf_launch_upsample_fwd(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift,
                            void* out, int64_t out_pitch, int N, int C, int Di, int Hi, int Wi, hipStream_t st,
                            int flat = 0 /* 1: bilinear x2 of a depth-1 tensor */);
// transposed stencil: din[lo-res] = sum of weighted dout[hi-res]
int hdf_launch_upsample_bwd(int dtype, const void* dout, int64_t dout_pitch, void* din, int64_t din_pitch, int N,
                            int C, int Di, int Hi, int Wi, hipStream_t st, int flat = 0);

// 1x1x1 head: logits[N][ncls][vox] (NCDHW) = W[ncls][C] . act(in) + b ;  act = relu(in*scale+shift) if scale
int hdf_launch_head_fwd(int dtype, const void* in, int64_t in_pitch, const float* scale, const float* shift,
                        const float* w, const float* b, void* logits, int N, int C, int ncls, int64_t vox,
                        hipStream_t st);
// dX (+)= W^T dlogits ; dW += dlogits . act(in)^T ; db += sum dlogits   (dW, db accumulated with float atomics)
int hdf_head_bwd_blocks(int64_t vox);
// MaxPool3d(2) backward accumulating into din + the first pass of the InstanceNorm(+ReLU) backward of the layer whose
// activation gradient din then is (hdf_maxpool_bwd_in_blocks rows per sample in `partials`)
int hdf_maxpool_bwd_in_blocks(int64_t pooled_vox, int C);
int hdf_launch_maxpool_bwd_in(int dtype, const void* dout, int64_t dout_pitch, const uint8_t* idx, void* din,
                              int64_t din_pitch, const void* y, int64_t y_pitch, const float* scale, const float* shift,
                              const float* mean, const float* rstd, float* partials, int N, int C, int Do, int Ho, int Wo,
                              hipStream_t st, int flat = 0 /* 1: MaxPool2d(2) windows of a depth-1 tensor */);
int hdf_launch_head_bwd(int dtype, const void* dlogits, const void* in, int64_t in_pitch, const float* scale,
                        const float* shift, const float* w, void* dx, int64_t dx_pitch, int accumulate_dx, float* dw,
                        float* db, int N, int C, int ncls, int64_t vox, hipStream_t st, const float* in_mean = nullptr,
                        const float* in_rstd = nullptr, float* inb_partials = nullptr);

// InstanceNorm+ReLU backward, stage 1: g = da * [y*scale+shift > 0]; partial sums of g and g*xhat
int hdf_launch_in_bwd_reduce(int dtype, const void* da, int64_t da_pitch, const void* y, int64_t y_pitch,
                             const float* scale, const float* shift, const float* mean, const float* rstd,
                             float* partials /*[N][blocks][C][2]*/, int blocks, int N, int C, int64_t vox,
                             hipStream_t st);
// stage 2: per (n,c) coefficients + dgamma/dbeta (accumulated, may be null for non-affine norms)
int hdf_launch_in_bwd_finalize(const float* partials, int blocks, int N, int C, int64_t vox, const float* gamma,
                               const float* rstd, float* k1, float* ka, float* kb, float* dgamma, float* dbeta,
                               hipStream_t st);
// stage 3: dy = k1 * (g - ka - xhat*kb)
int hdf_launch_in_bwd_apply(int dtype, const void* da, int64_t da_pitch, const void* y, int64_t y_pitch,
                            const float* scale, const float* shift, const float* mean, const float* rstd,
                            const float* k1, const float* ka, const float* kb, void* dy, int64_t dy_pitch, int N, int C,
                            int64_t vox, hipStream_t st);
int hdf_in_bwd_blocks(int64_t vox, int C);

// a (+)= b over a pitched view;  or a = b when accumulate == 0
int hdf_launch_add(int dtype, void* a, int64_t a_pitch, const void* b, int64_t b_pitch, int N, int C, int64_t vox,
                   int accumulate, hipStream_t st);
// bias gradient of a conv output gradient: db[c] += sum_{n,v} dy[n,v,c]
// out[c] += column sums (sum column) of a conv InstanceNorm partial table [rows][CP][2], c < C
int hdf_launch_stat_rows_sum(const float* partials, int rows, int C, int CP, float* out, hipStream_t st);
int hdf_launch_bias_grad(int dtype, const void* dy, int64_t dy_pitch, float* db, int C, int64_t nvox, hipStream_t st);
