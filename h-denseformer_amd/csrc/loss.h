#pragma once
#include "hdf_common.h"

struct LossScales {
  float V[4];
  float weight[4];
  int blocks[4];  // partial rows written per (scale, sample)
};

int hdf_loss_blocks();
size_t hdf_loss_workspace_floats(int N, int nscale);
int hdf_launch_loss_fwd(int dtype, const void* const* logits, const float* target, int nscale, int N, int C, int D,
                        int H, int W, float* ws, float* loss_out, hipStream_t st, float w_ce = 1.f, float w_dice = 1.f,
                        const float* class_weight = nullptr, int dice_ignore = 0);
int hdf_launch_loss_bwd(int dtype, const void* const* logits, const float* target, int nscale, int N, int C, int D,
                        int H, int W, const float* ws, const float* grad_out, void* const* dlogits, hipStream_t st,
                        float w_ce = 1.f, float w_dice = 1.f, const float* class_weight = nullptr, int dice_ignore = 0);
int hdf_launch_dice_counts(int dtype, const void* logits, const float* target, int N, int C, int64_t V,
                           unsigned long long* counts, hipStream_t st);
int hdf_launch_confusion(int dtype, const void* logits, const float* target, int N, int C, int64_t V,
                         unsigned long long* conf, int accumulate, hipStream_t st);
int hdf_launch_sw_accumulate(int dtype, const void* logits, int C, int pd, int ph, int pw, float* psum, float* cnt,
                             int D, int H, int W, int z0, int y0, int x0, hipStream_t st);
int hdf_launch_sw_finalize(const float* psum, const float* cnt, int C, int64_t V, uint8_t* label, hipStream_t st);
int hdf_launch_onehot(const uint8_t* lab, float* oh, int N, int C, int64_t V, hipStream_t st);
int hdf_launch_adam(float* p, const float* g, float* m, float* v, const uint8_t* decay, int64_t n, float lr, float b1,
                    float b2, float eps, float wd, int step, float gscale, hipStream_t st);
int hdf_launch_confusion_labels(const uint8_t* tgt, const uint8_t* pred, int C, int64_t n, unsigned long long* conf,
                                int accumulate, hipStream_t st);
// in-place input normalisation of one sample [C][V] fp32 (data_utils/data_loader.py:39-68); mode 0 MR, 1 PET/CT
size_t hdf_norm_ws_bytes(int C);
int hdf_launch_normalize(float* img, int C, int64_t V, int mode, float pmean, float pw, void* ws, hipStream_t st);
