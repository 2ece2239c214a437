/* libhdf_hip.so -- C ABI of the MI355X-native H-DenseFormer 3D training hot path.
 *
 * Nothing like this exists in the reference (pure Python on torch ops); each entry point
 * replaces the torch call sequence cited next to it (file:line into the reference repository).  All
 * pointers are raw DEVICE pointers unless marked host.  The library never allocates or frees device
 * memory, never synchronises the device and enqueues everything on the caller's stream (hipStream_t
 * passed as void*), so it composes with torch's caching allocator, autograd streams and RCCL side
 * streams.  Errors: every function returns 0 on success, non-zero otherwise; hdf_last_error() gives the
 * message (thread-local).  No C++ exception crosses this boundary.
 *
 * dtype enum: 0 = float32 storage (v_mfma_f32_32x32x2_f32, exact fp32 -- the parity path),
 *             1 = bfloat16 storage with fp32 accumulation (v_mfma_f32_32x32x16_bf16 -- the bench path).
 */
#ifndef HDF_H
#define HDF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hdf_plan hdf_plan;
typedef void* hdf_stream; /* hipStream_t */

const char* hdf_version(void);
const char* hdf_last_error(void);

/* ---- model plan: models/HDenseFormer.py:177-227 (HDenseFormer.__init__) ------------------------------- */
int hdf_plan_create(int in_channels, int n_cls, int n_filters, int D, int H, int W, int transformer_depth, int dtype,
                    hdf_plan** out);
void hdf_plan_destroy(hdf_plan* p);
/* parameter table = the reference state_dict, in registration order; the flat fp32
 * parameter / gradient buffers place tensor i at offset_i (64-byte aligned) */
int64_t hdf_plan_num_params(const hdf_plan* p);
int64_t hdf_plan_param_floats(const hdf_plan* p);
int hdf_plan_param_info(const hdf_plan* p, int64_t idx, char* name, int name_cap, int64_t* offset, int64_t* numel,
                        int* ndim, int64_t* shape5);
int64_t hdf_plan_workspace_bytes(hdf_plan* p, int batch);
/* named activation buffers inside the workspace (debug / parity tests): channels-last views */
int hdf_plan_buffer_info(hdf_plan* p, int batch, const char* name, int64_t* byte_offset, int64_t* pitch_elems,
                         int* channels, int* d, int* h, int* w);

/* HDenseFormer.forward, models/HDenseFormer.py:229-255.  x: [B,Cin,D,H,W] fp32 NCDHW.  out_i: [B,n_cls,D/2^i,..]
 * NCDHW in the plan's storage dtype.  training!=0 enables the dropout sites (:39,41,61,138) with the
 * counter-hash masks of (seed).  The workspace keeps what backward needs. */
int hdf_forward(hdf_plan* p, const float* x, const float* params, void* workspace, int64_t workspace_bytes, void* out0,
                void* out1, void* out2, void* out3, int batch, int training, uint64_t seed, hdf_stream stream);
/* autograd of the above (trainer.py:374-380 loss.backward()).  dout_i: gradients w.r.t. the 4 outputs, same
 * layout/dtype.  grads: flat fp32 buffer, OVERWRITTEN with d loss / d params. */
int hdf_backward(hdf_plan* p, const float* x, const float* params, void* workspace, int64_t workspace_bytes,
                 const void* dout0, const void* dout1, const void* dout2, const void* dout3, float* grads, int batch,
                 hdf_stream stream);

/* the same in stages so a caller can overlap the gradient all-reduce with the rest of backward (a stage needs the
 * earlier ones to have run):
 * stages bit 0: zero grads + decoder/encoder/heads (their parameter gradients are final afterwards);
 * stages bit 1: UpConv chain (deep_conv, up1..3);
 * stages bit 2: transformer branches.  7 = everything.  Replaces nn.DataParallel's reduce (trainer.py:228-229). */
int hdf_backward_stages(hdf_plan* p, const float* x, const float* params, void* workspace, int64_t workspace_bytes,
                        const void* dout0, const void* dout1, const void* dout2, const void* dout3, float* grads,
                        int batch, int stages, hdf_stream stream);

/* ---- loss: loss/combine_loss.py:68-79 DeepSuperloss(CEPlusDice(weight=None, ignore_index=0)) --------- */
int64_t hdf_loss_workspace_bytes(int batch);
int hdf_loss_forward(int dtype, const void* out0, const void* out1, const void* out2, const void* out3, int nscale,
                     const float* target_onehot, int batch, int n_cls, int D, int H, int W, void* workspace,
                     float* loss_out, hdf_stream stream);
int hdf_loss_backward(int dtype, const void* out0, const void* out1, const void* out2, const void* out3, int nscale,
                      const float* target_onehot, int batch, int n_cls, int D, int H, int W, const void* workspace,
                      const float* grad_out, void* dout0, void* dout1, void* dout2, void* dout3, hdf_stream stream);
/* hard-argmax Dice counts of trainer.py:919-945: counts[batch][8][3] = (|P&T|, |P|, |T|) per class, uint64 */
int hdf_dice_counts(int dtype, const void* logits, const float* target_onehot, int batch, int n_cls, int64_t voxels,
                    uint64_t* counts, hdf_stream stream);

/* running confusion matrix of metrics.RunningDice.update_matrix (metrics.py:104-133; the reference moves the argmax
 * maps to the host and calls sklearn): confusion[8][8] uint64, rows = target class, cols = predicted class, summed
 * over the batch; accumulate != 0 keeps the previous counts (running matrix over steps) */
int hdf_confusion_matrix(int dtype, const void* logits, const float* target_onehot, int batch, int n_cls,
                         int64_t voxels, uint64_t* confusion, int accumulate, hdf_stream stream);

/* ---- sliding-window inference (trainer.py:488-593): the per-window tail of the loop and the final vote.
 * hdf_sw_accumulate: softmax over classes of one window's full-resolution logits [n_cls][pd][ph][pw] (NCDHW, the
 * model's out0 for batch 1), added into prob_sum[n_cls][D][H][W] at (z0,y0,x0); count[D][H][W] += 1
 * (trainer.py:559-576; the reference keeps n_cls identical count planes).
 * hdf_sw_finalize: label[v] = argmax_c softmax(prob_sum[c][v] / count[v]) (trainer.py:580-584), uint8, first
 * maximum wins; voxels no window covered (count 0) give 0. */
int hdf_sw_accumulate(int dtype, const void* logits, int n_cls, int pd, int ph, int pw, float* prob_sum, float* count,
                      int D, int H, int W, int z0, int y0, int x0, hdf_stream stream);
int hdf_sw_finalize(const float* prob_sum, const float* count, int n_cls, int64_t voxels, uint8_t* label,
                    hdf_stream stream);

/* ---- label staging (data_utils/data_loader.py:126-159, To_Tensor): uint8 class map [N][voxels] -> fp32 one-hot
 * [N][n_cls][voxels]; channel z>=1 is (label == z), channel 0 is "no other class" (so values >= n_cls count as
 * background).  Ships 1 byte per voxel over PCIe instead of 4*n_cls. */
int hdf_onehot_from_labels(const uint8_t* labels, float* onehot, int batch, int n_cls, int64_t voxels,
                           hdf_stream stream);

/* ---- optimizer: torch.optim.Adam as configured by trainer.py:793-840 (L2 weight decay on the mask) ---- */
int hdf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* decay_mask,
                  int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  float grad_scale, hdf_stream stream);

/* ---- operator level (what nn.Conv3d / ConvTranspose3d / InstanceNorm3d / MaxPool3d / F.interpolate bind
 *      in the reference, HDenseFormer.py:148-175,199-227).  Channels-last activations with a voxel pitch. -- */
int hdf_op_to_channels_last(int dtype, const float* x_ncdhw, void* out, int N, int C, int CP, int64_t voxels,
                            hdf_stream stream);
/* dst[27][OP][IP] = src[o*so + i*si + tap] (tap reversed if flip) */
int hdf_op_pack_weights(int dtype, const float* src, void* dst, int O, int I, int OP, int IP, int64_t so, int64_t si,
                        int flip, hdf_stream stream);
/* mode 0: Conv3d(k3,s1,p1); mode 1: Conv3d(k3,s2,p1); mode 2: ConvTranspose3d(k3,s2,p1,op1).
 * in_scale/in_shift ([N][Cin], may be null): input is relu?(x*scale+shift).  stat_partials may be null. */
int hdf_op_conv3d(int dtype, int mode, const void* in, int64_t in_pitch, int Cin, int N, int Di, int Hi, int Wi,
                  const void* w_packed, const float* bias, const float* in_scale, const float* in_shift, int in_relu,
                  void* out, int64_t out_pitch, int Cout, float* stat_partials, int accumulate, hdf_stream stream);
/* partial rows per sample of stat_partials ([N*rows][Cout rounded up to 32][2] floats: sum, sum of squares) for this
 * layer shape: one row per output tile, or 512 per-workgroup rows when the weights-stationary kernel takes the layer */
int hdf_op_conv3d_stat_tiles(int dtype, int Cin, int Do, int Ho, int Wo);
int64_t hdf_op_wgrad_workspace_bytes(int stride, int N, int Ds, int Hs, int Ws, int SC, int LC);
/* dW[sc][lc][27] = sum S[i][sc] * L[stride*i-1+tap][lc]  (torch weight layout for both Conv3d and ConvTranspose3d) */
int hdf_op_conv3d_wgrad(int dtype, int stride, const void* sm, int64_t sm_pitch, int SC, const void* lg,
                        int64_t lg_pitch, int LC, int N, int Ds, int Hs, int Ws, const float* sm_scale,
                        const float* sm_shift, int sm_relu, const float* lg_scale, const float* lg_shift, int lg_relu,
                        float* dw, int sc_store, int lc_store, int accumulate, void* workspace, int64_t workspace_bytes,
                        hdf_stream stream);
int hdf_op_in_finalize(const float* partials, int N, int tiles, int C, int CP, int64_t voxels, const float* gamma,
                       const float* beta, float eps, float* mean, float* rstd, float* scale, float* shift,
                       hdf_stream stream);
/* InstanceNorm3d(+ReLU) backward of one layer (HDenseFormer.py:152-158): da = d/d(relu(IN(y))) -> dy = d/dy, plus
 * dgamma / dbeta (accumulated; may be null).  workspace: hdf_op_in_bwd_workspace_floats(N, C, voxels) floats. */
int64_t hdf_op_in_bwd_workspace_floats(int N, int C, int64_t voxels);
int hdf_op_in_bwd(int dtype, const void* da, int64_t da_pitch, const void* y, int64_t y_pitch, const float* scale,
                  const float* shift, const float* mean, const float* rstd, const float* gamma, void* dy,
                  int64_t dy_pitch, float* dgamma, float* dbeta, int N, int C, int64_t voxels, float* workspace,
                  hdf_stream stream);
int hdf_op_norm_relu_add(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift,
                         const void* skip, int64_t skip_pitch, void* out, int64_t out_pitch, int N, int C,
                         int64_t voxels, hdf_stream stream);
int hdf_op_maxpool_fwd(int dtype, const void* in, int64_t in_pitch, void* out, int64_t out_pitch, uint8_t* idx, int N,
                       int C, int Do, int Ho, int Wo, hdf_stream stream);
int hdf_op_maxpool_bwd(int dtype, const void* dout, int64_t dout_pitch, const uint8_t* idx, void* din,
                       int64_t din_pitch, int N, int C, int Do, int Ho, int Wo, int accumulate, hdf_stream stream);
int hdf_op_upsample_fwd(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift, void* out,
                        int64_t out_pitch, int N, int C, int Di, int Hi, int Wi, hdf_stream stream);
int hdf_op_upsample_bwd(int dtype, const void* dout, int64_t dout_pitch, void* din, int64_t din_pitch, int N, int C,
                        int Di, int Hi, int Wi, hdf_stream stream);

#ifdef __cplusplus
}
#endif
#endif
