/* libhdf_hip.so -- C ABI of the MI355X-native H-DenseFormer 3D training hot path.
 *
 * Nothing like this exists in the reference (pure Python on torch ops); each entry point
 * replaces the torch call sequence cited next to it (file:line into the reference repository).  All
 * pointers are raw DEVICE pointers unless marked host.  The library never allocates or frees device
 * memory, never synchronises the device and enqueues everything on the caller's stream (hipStream_t
 * passed as void*; hdf_backward additionally forks work onto one internal stream of the plan and joins
 * it back before returning, see there), so it composes with torch's caching allocator, autograd
 * streams and RCCL side streams.  Errors: every function returns 0 on success, non-zero otherwise; hdf_last_error() gives the
 * message (thread-local).  No C++ exception crosses this boundary.
 *
 * dtype enum: 0 = float32 storage (v_mfma_f32_32x32x2_f32, exact fp32 -- the parity path),
 *             1 = bfloat16 storage with fp32 accumulation (v_mfma_f32_32x32x16_bf16 -- the bench path),
 *             2 = float16 storage with fp32 accumulation (v_mfma_f32_32x32x16_f16 -- what the reference's
 *                 torch.cuda.amp.autocast(True) + GradScaler run, trainer.py:20-21,257,369-377).
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
/* Compute units the persistent kernels (conv_ws2 / conv_wgrad2 / the transposed-conv kernels) may assume: their grids
 * are one workgroup per CU.  256 = the whole MI355X (default); a multiple of 8 below that for launches on a stream the
 * CALLER created with hipExtStreamCreateWithCUMask, so that every workgroup of a launch is resident at once.  Process-wide
 * (forward and autograd's backward thread see the same value: it also enters the split-K decision of the low-resolution
 * convs, i.e. their summation order); a diagnostic knob -- the plan never changes it and owns no masked streams.
 * Round 6: a DOWNWARD override only -- the effective budget is min(this, the current device's compute-unit count
 * (hipDeviceAttributeMultiprocessorCount)), so partitioned or smaller gfx950 devices size their grids, and decide whether
 * the persistent transformer kernels can be resident together, from what they have. */
int hdf_set_cu_budget(int cus);

/* ---- model plan: models/HDenseFormer.py:177-227 (HDenseFormer.__init__) ------------------------------- */
int hdf_plan_create(int in_channels, int n_cls, int n_filters, int D, int H, int W, int transformer_depth, int dtype,
                    hdf_plan** out);
/* models/HDenseFormer_2D.py:172-229 (HDenseFormer_2D.__init__): the 2-D model.  The plan's parameter table is then the
 * 2-D reference state_dict (Conv2d / ConvTranspose2d kernels [..,3,3], patch kernels [..,16,16]); hdf_forward takes
 * x [B,C,H,W] and returns out_i [B,n_cls,H/2^i,W/2^i]; hdf_backward* take 2-D logit gradients and write 2-D parameter
 * gradients.  Round 6: the model runs NATIVELY on depth-1 tensors -- 2-D convolutions / transposed convolutions / weight
 * gradients (the 9 centre-plane taps of the embedded 27-tap panels), MaxPool2d, bilinear x2, 16 x 16 patches against depth
 * slice 0 of the embedded patch kernels.  hdf_plan_create_2d_embedded keeps the exact depth-16 replicated 3-D embedding of
 * rounds 3-5 (16x ... 2x the arithmetic): the oracle of tests/test_gpu_model_2d.py. */
int hdf_plan_create_2d(int in_channels, int n_cls, int n_filters, int H, int W, int transformer_depth, int dtype,
                       hdf_plan** out);
int hdf_plan_create_2d_embedded(int in_channels, int n_cls, int n_filters, int H, int W, int transformer_depth, int dtype,
                                hdf_plan** out);
/* Operator level (hdf_op_conv3d, hdf_op_conv3d_wgrad, below): a tensor depth of 1 selects the 2-D operator -- Conv2d(k3,p1),
 * the stride-2 gather in y and x, ConvTranspose2d(k3,s2,p1,op1) and their weight gradients; the depth axis is never
 * strided (output depth 1), the packed panel keeps its 27-tap layout with the 2-D kernel on depth tap 1, and a weight
 * gradient comes back as [.., 27] with zeros off the centre plane (tests/test_gpu_ops_2d.py). */
void hdf_plan_destroy(hdf_plan* p);
/* parameter table = the reference state_dict, in registration order; the flat fp32
 * parameter / gradient buffers place tensor i at offset_i (64-byte aligned) */
int64_t hdf_plan_num_params(const hdf_plan* p);
int64_t hdf_plan_param_floats(const hdf_plan* p);
int hdf_plan_param_info(const hdf_plan* p, int64_t idx, char* name, int name_cap, int64_t* offset, int64_t* numel,
                        int* ndim, int64_t* shape5);
int64_t hdf_plan_workspace_bytes(hdf_plan* p, int batch);
/* the prefix of that arena a forward touches: enough for hdf_forward when no hdf_backward follows (eval, sliding-window
 * prediction); hdf_backward* refuse a workspace smaller than hdf_plan_workspace_bytes */
int64_t hdf_plan_inference_workspace_bytes(hdf_plan* p, int batch);
/* named activation buffers inside the workspace (debug / parity tests): channels-last views */
int hdf_plan_buffer_info(hdf_plan* p, int batch, const char* name, int64_t* byte_offset, int64_t* pitch_elems,
                         int* channels, int* d, int* h, int* w);

/* raw fp32 regions of the transformer branches inside the workspace (debug / parity tests): "tf_F" (feature buffers of
 * HDenseFormer.py:91-99, [block][row][DM+128]), "tf_save" (per layer h0 | qkv | ob | lse | h1 | h2), "tf_dF", "tf_tape",
 * "tf_otape" (backward), "tf_sync" (arrival counters and the timeout words of the persistent transformer kernels: word
 * 32 * (in_channels * batch) of each half is non-zero after a launch whose per-sequence barrier gave up) */
int hdf_plan_region_info(hdf_plan* p, int batch, const char* name, int64_t* byte_offset, int64_t* bytes);

/* HDenseFormer.forward, models/HDenseFormer.py:229-255.  x: [B,Cin,D,H,W] fp32 NCDHW.  out_i: [B,n_cls,D/2^i,..]
 * NCDHW in the plan's storage dtype.  training!=0 enables the dropout sites (:39,41,61,138) with the
 * counter-hash masks of (seed).  The workspace keeps what backward needs. */
int hdf_forward(hdf_plan* p, const float* x, const float* params, void* workspace, int64_t workspace_bytes, void* out0,
                void* out1, void* out2, void* out3, int batch, int training, uint64_t seed, hdf_stream stream);
/* autograd of the above (trainer.py:374-380 loss.backward()).  dout_i: gradients w.r.t. the 4 outputs, same
 * layout/dtype.  grads: flat fp32 buffer, OVERWRITTEN with d loss / d params.
 * Streams: the conv weight gradients run on one internal side stream of the plan, forked from `stream` with an event;
 * `stream` waits for them before the call's work is complete on it (HDF_NO_ASYNC_WGRAD=1 disables). */
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

/* hdf_backward (ONE call, with its internal side / branch streams) that also tells the caller when each gradient BUCKET is
 * final.  The flat gradient buffer is cut into HDF_NUM_GRAD_BUCKETS contiguous ranges (hdf_plan_grad_bucket, floats):
 *   0  decoder + heads      (upconv_3 .. conv1x1_d3)        final a third of the way into the backward
 *   1  UpConv chain         (deep_conv, up1..3)             final when the chain's backward is through
 *   2  transformer branches (attns.*)                       final at the end of the branch stream
 *   3  encoder levels 1-3   (block_2_1_left .. block_4_2_left)   final before the UpConv chain's backward starts
 *   4  encoder level 0      (block_1_1_left, block_1_2_left: 0.1 MB)   final with the last kernel of the call
 * (round 6: five buckets; rounds 4-5 had three, and their "encoder / decoder / heads" bucket -- 26 of 62 MB at n_filters 32
 * -- was final only at the very end, i.e. its all-reduce was exposed.)  bucket_events[k] receives an event owned by the plan
 * -- the handles are stable for the plan's lifetime and RE-RECORDED by every hdf_backward_events call, so a waiter must
 * enqueue its wait before the next call --, already recorded, on whichever internal stream finishes that bucket, when the
 * call returns.  A communication stream that waits for event k (hdf_stream_wait_event, or hipStreamWaitEvent on the
 * handle) may all-reduce bucket k while the rest of the backward is still running: no host round trip between the stages,
 * the branch-stream fork of the one-call backward stays.  Order of finality in the default arrangement: 0, 3, 1, 2, 4.
 * 2-D plans: all five events are recorded at the end (their gradients are extracted from the embedding in one pass), and
 * hdf_plan_grad_bucket is refused.  Replaces nn.DataParallel's reduce (trainer.py:228-229). */
#define HDF_NUM_GRAD_BUCKETS 5
int hdf_plan_grad_bucket(const hdf_plan* p, int k, int64_t* lo, int64_t* hi);
int hdf_backward_events(hdf_plan* p, const float* x, const float* params, void* workspace, int64_t workspace_bytes,
                        const void* dout0, const void* dout1, const void* dout2, const void* dout3, float* grads,
                        int batch, hdf_stream stream, void** bucket_events /* [HDF_NUM_GRAD_BUCKETS] hipEvent_t out */);
/* ---- the persistent transformer kernels and a shared device (round 6) ----------------------------------------------
 * The plan runs all dense layers of the multi-path transformer (models/HDenseFormer.py:78-145) as ONE persistent launch per
 * direction when every 16-token tile gets a compute unit of its own (tiles <= the device's compute units; else, and under
 * HDF_NO_TF_CHAIN=1, as a chain of ~100 small launches).  Those launches wait on each other inside the kernel, so all their
 * workgroups must be resident together.  The library guarantees that within the PROCESS (one such launch in flight per
 * device at a time: each waits for the previous one's event).  If something else holds compute units long enough that a
 * per-sequence barrier is not completed within the deadline (default 1.5 s), the launch does NOT trap any more: it ends by
 * itself and writes a host-mapped status word; the call's last launch then turns the head of every output (forward) / of
 * the gradient buffer (backward) into NaN, so the loss and the optimizer step of that iteration are NaN, never plausible
 * garbage.  The plan's NEXT hdf_forward / hdf_backward* call then launches
 * nothing and returns HDF_ERR_CHAIN_TIMEOUT (4) once, and the plan uses the launch chain from then on: redo the step.
 * hdf_plan_set_chain_timeout_us: the deadline of one barrier wait (100 us .. 30 s).
 * hdf_plan_chain_state: *persistent = 1 when the next forward of `batch` samples would take the persistent kernels;
 * *gave_up_workgroup = id of the workgroup whose give-up is pending or was last reported, -1 if none.  Meaningful right
 * after the caller synchronised the device; never synchronises itself. */
#define HDF_ERR_CHAIN_TIMEOUT 4
int hdf_plan_set_chain_timeout_us(hdf_plan* p, int64_t usec);
int hdf_plan_chain_state(hdf_plan* p, int batch, int* persistent, int* gave_up_workgroup);
/* Tests only: take the persistent kernels even when their grid cannot be resident together (more 16-token tiles than
 * compute units).  Such a forward gives up by construction: this is how tests/test_gpu_chain.py exercises the path above. */
int hdf_plan_force_persistent(hdf_plan* p, int on);
/* Tests and tools: a stand-in for a collective's kernel -- `workgroups` workgroups of 256 threads, each holding `lds_bytes`
 * of LDS (160 KiB: a compute unit of its own) and `vgprs` (0 or 128) vector registers per lane, spinning for `usec` on the
 * device's real-time counter. */
int hdf_op_occupy(int workgroups, int lds_bytes, int vgprs, int usec, hdf_stream stream);

/* Measurement hook: every following hdf_forward on this plan records ev_start immediately before and ev_stop immediately
 * after the launch of the forward's dominant convolution (block_1_1_right, 64 -> 32 channels at full resolution: the
 * kernel bench.py's `roofline` object reports) on the caller's stream.  Two caller-owned HIP events created with timing
 * enabled; NULL, NULL switches it off.  The launch itself is unchanged. */
int hdf_plan_set_probe(hdf_plan* p, void* ev_start, void* ev_stop);
/* hipStreamWaitEvent(stream, event) for callers that hold streams and events as opaque handles */
int hdf_stream_wait_event(hdf_stream stream, void* event);

/* ---- loss: loss/combine_loss.py:68-79 DeepSuperloss(CEPlusDice(weight=None, ignore_index=0)) --------- */
int64_t hdf_loss_workspace_bytes(int batch);
int hdf_loss_forward(int dtype, const void* out0, const void* out1, const void* out2, const void* out3, int nscale,
                     const float* target_onehot, int batch, int n_cls, int D, int H, int W, void* workspace,
                     float* loss_out, hdf_stream stream);
int hdf_loss_backward(int dtype, const void* out0, const void* out1, const void* out2, const void* out3, int nscale,
                      const float* target_onehot, int batch, int n_cls, int D, int H, int W, const void* workspace,
                      const float* grad_out, void* dout0, void* dout1, void* dout2, void* dout3, hdf_stream stream);
/* the same with the two terms weighted (ce_weight * CE + dice_weight * Dice per scale): 1,1 is CEPlusDice
 * (loss/combine_loss.py:8-35); 0,1 is the stand-alone DiceLoss(ignore_index=0) of loss/dice_loss.py:53-87; 1,0 is
 * CrossentropyLoss (loss/cross_entropy.py:8-22).  nscale = 1 gives the un-supervised (single output) forms. */
int hdf_loss_terms_forward(int dtype, const void* out0, const void* out1, const void* out2, const void* out3, int nscale,
                           const float* target_onehot, int batch, int n_cls, int D, int H, int W, float ce_weight,
                           float dice_weight, void* workspace, float* loss_out, hdf_stream stream);
int hdf_loss_terms_backward(int dtype, const void* out0, const void* out1, const void* out2, const void* out3,
                            int nscale, const float* target_onehot, int batch, int n_cls, int D, int H, int W,
                            float ce_weight, float dice_weight, const void* workspace, const float* grad_out,
                            void* dout0, void* dout1, void* dout2, void* dout3, hdf_stream stream);
/* the general forms trainer.py:743-771 (_get_loss) can build: class_weight = device [n_cls] fp32 or NULL
 * (DiceLoss multiplies class i's Dice term by weight[i], loss/dice_loss.py:79-82; CrossEntropyLoss(weight) is
 * sum_v w[t_v] nll_v / sum_v w[t_v], loss/cross_entropy.py:8-22 on torch.nn.CrossEntropyLoss); dice_ignore_index =
 * the class DiceLoss skips (then the class sum is divided by n_cls - 1) or -1 for ignore_index=None (divided by
 * n_cls), loss/dice_loss.py:75-87.  NULL, 0 reproduce hdf_loss_terms_* bit for bit. */
int hdf_loss_weighted_forward(int dtype, const void* out0, const void* out1, const void* out2, const void* out3,
                              int nscale, const float* target_onehot, int batch, int n_cls, int D, int H, int W,
                              float ce_weight, float dice_weight, const float* class_weight, int dice_ignore_index,
                              void* workspace, float* loss_out, hdf_stream stream);
int hdf_loss_weighted_backward(int dtype, const void* out0, const void* out1, const void* out2, const void* out3,
                               int nscale, const float* target_onehot, int batch, int n_cls, int D, int H, int W,
                               float ce_weight, float dice_weight, const float* class_weight, int dice_ignore_index,
                               const void* workspace, const float* grad_out, void* dout0, void* dout1, void* dout2,
                               void* dout3, hdf_stream stream);
/* hard-argmax Dice counts of trainer.py:919-945: counts[batch][8][3] = (|P&T|, |P|, |T|) per class, uint64 */
int hdf_dice_counts(int dtype, const void* logits, const float* target_onehot, int batch, int n_cls, int64_t voxels,
                    uint64_t* counts, hdf_stream stream);

/* running confusion matrix of metrics.RunningDice.update_matrix (metrics.py:104-133; the reference moves the argmax
 * maps to the host and calls sklearn): confusion[8][8] uint64, rows = target class, cols = predicted class, summed
 * over the batch; accumulate != 0 keeps the previous counts (running matrix over steps) */
int hdf_confusion_matrix(int dtype, const void* logits, const float* target_onehot, int batch, int n_cls,
                         int64_t voxels, uint64_t* confusion, int accumulate, hdf_stream stream);

/* the reference's own call signature, RunningDice.update_matrix(ground_truth, prediction) (metrics.py:104): two uint8
 * class maps of n voxels each; labels >= n_cls are dropped like sklearn's confusion_matrix(labels=...) does */
int hdf_confusion_matrix_labels(const uint8_t* target, const uint8_t* prediction, int n_cls, int64_t n,
                                uint64_t* confusion, int accumulate, hdf_stream stream);

/* ---- input normalisation on the device, in place on one fp32 sample [channels][voxels] -----------------------
 * hdf_normalize_mr: MRNormalize (data_utils/data_loader.py:39-50): every channel divided by its maximum when that is
 * non-zero, then negatives clamped to 0.  hdf_normalize_petct: PETandCTNormalize (:53-68): channel 0 ->
 * (clip(x, mean-w, mean+w) - mean) / w, channel 1 -> (x - mean_1) / (std_1 + 1e-3) (population std), others untouched.
 * workspace: hdf_normalize_workspace_bytes(channels) bytes. */
int64_t hdf_normalize_workspace_bytes(int channels);
int hdf_normalize_mr(float* image, int channels, int64_t voxels, void* workspace, hdf_stream stream);
int hdf_normalize_petct(float* image, int channels, int64_t voxels, float mean, float w, void* workspace,
                        hdf_stream stream);

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
/* The encoder's first layer as the plan runs it in a 16-bit storage mode (nn.Conv3d(in_channels <= 4, n_filters, 3, padding=1),
 * HDenseFormer.py:152-158,190; csrc/conv_first.hip: K = (tap, channel) instead of 27 quarter-empty channel rows).  in:
 * channels-last, the first Cin of in_pitch channels per voxel are read (in_pitch % 4 == 0); weight: the torch fp32 tensor
 * [Cout][Cin][3][3][3] itself (rounded to the storage type inside); stat_partials (optional): [N][512][round_up(Cout,32)][2]
 * InstanceNorm partial sums (sum, sum of squares of the fp32 results), every row written. */
int hdf_op_conv3d_first(int dtype, const void* in, int64_t in_pitch, int Cin, int N, int D, int H, int W, const float* weight,
                        const float* bias, void* out, int64_t out_pitch, int Cout, float* stat_partials,
                        hdf_stream stream);
/* ... and its weight gradient: dweight [Cout][Cin][3][3][3] fp32 (+)= sum over samples and voxels of dy (x) shifted x.
 * dy: channels-last, Cout (a multiple of 8) of dy_pitch channels; x as above; workspace: >= 2 * 256 * round_up(Cout,32) *
 * 512 bytes of scratch (fp32 partial sums per workgroup, reduced in a fixed order). */
int hdf_op_conv3d_first_wgrad(int dtype, const void* dy, int64_t dy_pitch, int Cout, const void* x, int64_t x_pitch,
                              int Cin, int N, int D, int H, int W, float* dweight, int accumulate, void* workspace,
                              int64_t workspace_bytes, hdf_stream stream);
/* The same with the second pass of the layer's InstanceNorm(+ReLU) backward applied to the rows as they are staged: da is
 * the gradient w.r.t. the activation relu(IN(y)), and dy = k1 * ((y*scale+shift > 0 ? da : 0) - ka - (y-mean)*rstd*kb)
 * (rounded to the storage type: what hdf_op_in_bwd's apply pass would have written) is what enters the weight gradient.
 * The first layer has no input gradient, so dy itself is never needed.  Vectors [N][Cout]. */
int hdf_op_conv3d_first_wgrad_in(int dtype, const void* da, int64_t da_pitch, int Cout, const void* y, int64_t y_pitch,
                                 const float* scale, const float* shift, const float* mean, const float* rstd,
                                 const float* k1, const float* ka, const float* kb, const void* x, int64_t x_pitch, int Cin,
                                 int N, int D, int H, int W, float* dweight, int accumulate, void* workspace,
                                 int64_t workspace_bytes, hdf_stream stream);
/* The same Conv3d(k3,s1,p1) forced through the weights-in-registers kernel (csrc/conv_wr.hip: 16-bit storage, Cin of 32
 * or 64, >= 48^3) whatever the plan's routing rule says; HDF_ERR_UNSUPPORTED for other shapes.  Tests and tools. */
int hdf_op_conv3d_wr(int dtype, const void* in, int64_t in_pitch, int Cin, int N, int D, int H, int W,
                     const void* w_packed, const float* bias, const float* in_scale, const float* in_shift, int in_relu,
                     void* out, int64_t out_pitch, int Cout, float* stat_partials, int accumulate, hdf_stream stream);
/* partial rows per sample of stat_partials ([N*rows][Cout rounded up to 32][2] floats: sum, sum of squares) for this
 * layer shape: one row per output tile, or 512 per-workgroup rows when the weights-stationary kernel takes the layer */
int hdf_op_conv3d_stat_tiles(int dtype, int Cin, int Do, int Ho, int Wo);
/* hdf_op_conv3d (mode 0) as a data gradient whose epilogue also takes the first pass of the NEXT InstanceNorm(+ReLU)
 * backward: out is the gradient w.r.t. the activation relu(IN(y)) of the layer below, and partials [N][512][Cout][2] receive
 * per channel (sum g, sum g * xhat), g = the stored out where y*scale+shift > 0, xhat = (y-mean)*rstd -- the rows
 * hdf_op_in_bwd's reduce pass would produce from (out, y), without that pass.  16-bit storage, Cin = Cout = 32, extents
 * multiples of (4, 8, 8) and >= 48^3 (the level-0 data gradients of the plan); other launches are refused. */
int hdf_op_conv3d_bwd_stats(int dtype, const void* in, int64_t in_pitch, int Cin, int N, int D, int H, int W,
                            const void* w_packed, void* out, int64_t out_pitch, int Cout, const void* y, int64_t y_pitch,
                            const float* scale, const float* shift, const float* mean, const float* rstd,
                            float* partials, hdf_stream stream);
/* Conv3d(k3,s1,p1) whose output channels [0, split) go to `out` and [split, Cout) to `out2` (two dense buffers of one
 * pitch; split % 32 == 0), the form the plan uses for the gradient of a decoder concat [upconv | skip]
 * (models/HDenseFormer.py:245-253 backward).  With stat_partials ([N * hdf_op_conv3d_stat_tiles][round_up(Cout,32)][2])
 * and colsum ([colsum_C] floats) the per-channel sums over all voxels of channels [0, colsum_C) are written too: the
 * ConvTranspose3d bias gradient taken from the conv's statistics epilogue. */
int hdf_op_conv3d_split(int dtype, const void* in, int64_t in_pitch, int Cin, int N, int D, int H, int W,
                        const void* w_packed, void* out, void* out2, int64_t out_pitch, int Cout, int split,
                        float* stat_partials, float* colsum, int colsum_C, hdf_stream stream);
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
/* The same backward with its second pass INSIDE the layer's weight gradient (round 5; 16-bit storage, Conv3d k3 s1 p1;
 * reference: models/HDenseFormer.py:148-159 backward): the reduce + finalize passes of hdf_op_in_bwd, then ONE launch that
 * applies the second pass to the rows of d(activation) it stages, writes dy (bit-identical to hdf_op_in_bwd's) and
 * contracts it with the layer's input x -- optionally x -> relu(x * x_scale + x_shift), [N][Cin] -- into
 * dw [Cout][Cin][27] (bit-identical to hdf_op_conv3d_wgrad on that dy).  What hdf_backward does for the 128^3 layers.
 * HDF_ERR_UNSUPPORTED where the fused kernel does not take the launch (fp32 storage, tensors of 2 GiB and more).
 * workspace: hdf_op_in_bwd_workspace_floats(N, Cout, D*H*W) floats; wgrad_workspace: hdf_op_wgrad_workspace_bytes(1, ...). */
int hdf_op_in_bwd_wgrad(int dtype, const void* da, int64_t da_pitch, const void* y, int64_t y_pitch, const float* scale,
                        const float* shift, const float* mean, const float* rstd, const float* gamma, void* dy,
                        int64_t dy_pitch, float* dgamma, float* dbeta, const void* x, int64_t x_pitch, int Cin,
                        const float* x_scale, const float* x_shift, int x_relu, int N, int Cout, int D, int H, int W,
                        float* dw, float* workspace, void* wgrad_workspace, int64_t wgrad_workspace_bytes,
                        hdf_stream stream);
int hdf_op_norm_relu_add(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift,
                         const void* skip, int64_t skip_pitch, void* out, int64_t out_pitch, int N, int C,
                         int64_t voxels, hdf_stream stream);
int hdf_op_maxpool_fwd(int dtype, const void* in, int64_t in_pitch, void* out, int64_t out_pitch, uint8_t* idx, int N,
                       int C, int Do, int Ho, int Wo, hdf_stream stream);
/* The encoder tail of a level in one pass (models/HDenseFormer.py:238-243): ds = relu(y * scale + shift) + skip (the
 * InstanceNorm + ReLU of the level's second conv plus the transformer feature at_k), stored, and MaxPool3d(2) of the stored
 * values with the arg-max byte per channel (0..7 = dz*4 + dy*2 + dx, first maximum in scan order as torch).
 * (Do, Ho, Wo) = the pooled size. */
int hdf_op_enc_tail(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift, const void* skip,
                    int64_t skip_pitch, void* ds, int64_t ds_pitch, void* pooled, int64_t pooled_pitch, uint8_t* idx,
                    int N, int C, int Do, int Ho, int Wo, hdf_stream stream);
/* hdf_op_enc_tail with skip = Upsample(x2, trilinear)(relu(low * lscale + lshift)) evaluated inside the pass (low at the
 * pooled extent Do x Ho x Wo): level 0 of the encoder, where the skip is the UpConv chain's last feature
 * (HDenseFormer.py:168-175, 237; the up-sampled tensor is not written). */
int hdf_op_enc_tail_up(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift, const void* low,
                       int64_t low_pitch, const float* lscale, const float* lshift, void* ds, int64_t ds_pitch,
                       void* pooled, int64_t pooled_pitch, uint8_t* idx, int N, int C, int Do, int Ho, int Wo,
                       hdf_stream stream);
int hdf_op_maxpool_bwd(int dtype, const void* dout, int64_t dout_pitch, const uint8_t* idx, void* din,
                       int64_t din_pitch, int N, int C, int Do, int Ho, int Wo, int accumulate, hdf_stream stream);
/* hdf_op_maxpool_bwd with accumulate = 1 for the encoder levels: din becomes the complete gradient of
 * ds = relu(y * scale + shift) + skip, and the same pass writes the first pass of that InstanceNorm(+ReLU)'s backward:
 * partials[n][row][c][2] = per-workgroup (sum g, sum g * (y - mean) * rstd) with g = the stored din where the activation
 * is positive, hdf_op_maxpool_bwd_in_rows(C, Do, Ho, Wo) rows per sample (what hdf_op_in_bwd's reduce pass would write). */
int hdf_op_maxpool_bwd_in_rows(int C, int Do, int Ho, int Wo);
int hdf_op_maxpool_bwd_in(int dtype, const void* dout, int64_t dout_pitch, const uint8_t* idx, void* din,
                          int64_t din_pitch, const void* y, int64_t y_pitch, const float* scale, const float* shift,
                          const float* mean, const float* rstd, float* partials, int N, int C, int Do, int Ho, int Wo,
                          hdf_stream stream);
int hdf_op_upsample_fwd(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift, void* out,
                        int64_t out_pitch, int N, int C, int Di, int Hi, int Wi, hdf_stream stream);
int hdf_op_upsample_bwd(int dtype, const void* dout, int64_t dout_pitch, void* din, int64_t din_pitch, int N, int C,
                        int Di, int Hi, int Wi, hdf_stream stream);


/* ---- operator level, transformer branch and heads (what nn.Linear / LayerNorm / GELU / Softmax / matmul / Dropout and
 *      the 1x1x1 nn.Conv3d bind in the reference, HDenseFormer.py:11-145,223-227).  All fp32.
 * Token rows are ordered row = (m*B + b)*N + n (modality, sample, token); the parameters of modality m live at
 * pointer + m*mstride floats (the flat state_dict layout of the plan; pass M = 1 for a single Linear stack).
 * A block's dense feature buffer F is [rows][DM+128]: columns [0,DM) the block input, [DM+32l, DM+32l+32) the feature
 * dense layer l appends (HDenseFormer.py:91-101).  training != 0 applies the p = 0.5 hash dropout of (seed). */

/* Dense_Attention core (:67-74): qkv [nseq*N][96] = (q | k | v), 8 heads x 4 -> ob [nseq*N][32] (heads merged, before
 * to_out) and lse [nseq*N][8] (row log-sum-exp of the 0.5-scaled scores, natural log) */
int hdf_op_attention_fwd(const float* qkv, int nseq, int N, float* ob, float* lse, hdf_stream stream);
int hdf_op_attention_bwd(const float* qkv, const float* ob, const float* lse, const float* d_ob, float* dqkv, int nseq,
                         int N, hdf_stream stream);
/* The backward as the plan runs it in storage mode `dtype` (HDF_F32: identical to hdf_op_attention_bwd).  In the 16-bit
 * modes the scores and the softmax stay fp32 and the accumulations dS.k, dS^T.q and attn^T.d_ob take bf16 / f16 operands
 * with fp32 accumulation on the matrix core: what torch.cuda.amp.autocast does to the torch.matmul calls of
 * Dense_Attention (HDenseFormer.py:70-73 under trainer.py:369).  The forward is the exact one in every mode.
 * Same buffers as above, all fp32. */
int hdf_op_attention_amp_bwd(int dtype, const float* qkv, const float* ob, const float* lse, const float* d_ob,
                             float* dqkv, int nseq, int N, hdf_stream stream);
/* Dense_TransformerBlock front (:115-119,133-138): Conv3d(1 -> DM, k16, s16) + flatten + position embedding + dropout.
 * x: [B][M][D][H][W]; writes F[:, 0:DM].  backward: dF -> dweight, dpos (+=), dbias (+=); scratch rows*DM floats */
int hdf_op_patch_embed_fwd(const float* x, int M, int B, int D, int H, int W, int DM, const float* weight,
                           const float* bias, const float* pos, int64_t mstride, float* F, int training, uint64_t seed,
                           hdf_stream stream);
int hdf_op_patch_embed_bwd(const float* x, int M, int B, int D, int H, int W, int DM, const float* dF, int64_t mstride,
                           float* dweight, float* dbias, float* dpos, float* scratch, int training, uint64_t seed,
                           hdf_stream stream);
/* one dense layer of DensePreConv_AttentionBlock (:93-98): h0 = Linear(F[:, 0:DM+32l]); h1 = attn(LN(h0)) + h0;
 * h2 = ff(LN(h1)) + h1; F[:, DM+32l : +32] = ff(LN(h2)).  params13 / grads13: HOST arrays of 13 device pointers in
 * state_dict order (0.weight, 0.bias, 1.norm.weight, 1.norm.bias, 1.fn.to_qkv.weight, 1.fn.to_out.0.weight,
 * 1.fn.to_out.0.bias, 2.norm.weight, 2.norm.bias, 2.fn.net.0.weight, .bias, 2.fn.net.3.weight, .bias).
 * save: rows*232 floats kept for backward.  backward: dF[:, DM+32l..] is read, dF[:, 0:DM+32l] += ; parameter
 * gradients are ACCUMULATED; scratch rows*160 floats. */
int hdf_op_dense_layer_fwd(int M, int B, int N, int DM, int block, int layer, const float* const* params13,
                           int64_t mstride, float* F, float* save, int training, uint64_t seed, hdf_stream stream);
int hdf_op_dense_layer_bwd(int M, int B, int N, int DM, int block, int layer, const float* const* params13,
                           float* const* grads13, int64_t mstride, const float* F, float* dF, const float* save,
                           float* scratch, int training, uint64_t seed, hdf_stream stream);
/* out_layer of a block (:89,99-100): DenseForward(DM+128 -> 64 -> DM) of the whole feature buffer.  Exactly one of
 * next_F (fp32, next block's F[:, 0:DM]) / attnall (storage dtype, channels-last [B][N][M*DM], the 'b (d h w) c'
 * rearrange + modality concat of :144,230) is written.  params4 / grads4: net.0.weight, net.0.bias, net.3.weight,
 * net.3.bias.  backward writes dF[:, 0:DM+128] (overwrites) and accumulates the parameter gradients. */
int hdf_op_block_out_fwd(int M, int B, int N, int DM, int block, const float* const* params4, int64_t mstride,
                         const float* F, float* next_F, void* attnall, int dtype, int training, uint64_t seed,
                         hdf_stream stream);
int hdf_op_block_out_bwd(int M, int B, int N, int DM, int block, const float* const* params4, float* const* grads4,
                         int64_t mstride, const float* F, const float* dF_next, const void* d_attnall, int dtype,
                         float* dF, int training, uint64_t seed, hdf_stream stream);
/* 1x1x1 head (:223-227): logits [N][n_cls][voxels] (NCDHW, storage dtype) = W . act(in) + b with in channels-last,
 * act = relu(in*scale+shift) when in_scale is given.  backward: dx (+)= W^T dlogits (gradient w.r.t. act(in)),
 * dweight / dbias accumulated. */
int hdf_op_head_fwd(int dtype, const void* in, int64_t in_pitch, const float* in_scale, const float* in_shift,
                    const float* weight, const float* bias, void* logits, int N, int C, int n_cls, int64_t voxels,
                    hdf_stream stream);
int hdf_op_head_bwd(int dtype, const void* dlogits, const void* in, int64_t in_pitch, const float* in_scale,
                    const float* in_shift, const float* weight, void* dx, int64_t dx_pitch, int accumulate_dx,
                    float* dweight, float* dbias, int N, int C, int n_cls, int64_t voxels, hdf_stream stream);

#ifdef __cplusplus
}
#endif
#endif
