// Fused DeepSuperloss(CEPlusDice) forward/backward, on-device Dice metric, flat Adam.
//
// Reference: loss/combine_loss.py:8-35,68-79 ; loss/dice_loss.py:5-87 ; loss/cross_entropy.py:8-22 ;
// metric trainer.py:891-945 ; optimizer torch.optim.Adam as built by trainer.py:793-840.
// One pass per scale reads the logits once (NCDHW, coalesced along voxels) and the fp32 one-hot target
// at the 2^i-strided positions (nearest down-sampling), producing per-(sample,class) sums
// sum(p*t), sum(p), sum(t) and the CE sum; backward recomputes the softmax from the logits.
#include "loss.h"

namespace {
constexpr int MAXC = 8;
constexpr int LOSS_BLOCKS = 1024;
constexpr int NSTAT = 3 * MAXC + 2;  // per class: sum p*t, sum p, sum t; then the (weighted) CE sum and the sum of CE weights

template <typename T>
__global__ __launch_bounds__(256) void loss_fwd_kernel(const T* __restrict__ logits, const float* __restrict__ target,
                                                       int C, int Ds, int Hs, int Ws, int stride, int D, int H, int W,
                                                       const float* __restrict__ cw /*[C] class weights or null*/,
                                                       float* __restrict__ partials /*[N][blocks][NSTAT]*/) {
  __shared__ float red[4][NSTAT];
  const int n = blockIdx.y;
  const int64_t V = (int64_t)Ds * Hs * Ws, Vf = (int64_t)D * H * W;
  float acc[NSTAT];
#pragma unroll
  for (int i = 0; i < NSTAT; i++) acc[i] = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
    int x = v % Ws, y = (v / Ws) % Hs, z = v / ((int64_t)Ws * Hs);
    int64_t vf = ((int64_t)z * stride * H + (int64_t)y * stride) * W + (int64_t)x * stride;
    float lg[MAXC], t[MAXC];
    float mx = -INFINITY, tbest = -INFINITY;
    int tc = 0;
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
      if (c < C) {
        lg[c] = ST<T>::ld(logits + ((int64_t)n * C + c) * V + v);
        t[c] = target[((int64_t)n * C + c) * Vf + vf];
        mx = fmaxf(mx, lg[c]);
        if (t[c] > tbest) {
          tbest = t[c];
          tc = c;
        }
      }
    }
    float se = 0.f, e[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; c++)
      if (c < C) {
        e[c] = __expf(lg[c] - mx);
        se += e[c];
      }
    float inv = 1.f / se;
    float lse = mx + __logf(se);
#pragma unroll
    for (int c = 0; c < MAXC; c++)
      if (c < C) {
        float p = e[c] * inv;
        acc[c] += p * t[c];
        acc[MAXC + c] += p;
        acc[2 * MAXC + c] += t[c];
        if (c == tc) {
          // torch CrossEntropyLoss(weight=w, reduction='mean'): sum_v w[t_v] * nll_v / sum_v w[t_v]
          const float wv = cw ? cw[c] : 1.f;
          acc[3 * MAXC] += cw ? wv * (lse - lg[c]) : lse - lg[c];
          acc[3 * MAXC + 1] += wv;
        }
      }
  }
#pragma unroll
  for (int i = 0; i < NSTAT; i++) acc[i] = wave_sum(acc[i]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < NSTAT; i++) red[wave][i] = acc[i];
  __syncthreads();
  if (threadIdx.x < NSTAT)
    partials[((int64_t)n * gridDim.x + blockIdx.x) * NSTAT + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// grid = nscale*N blocks of 256 threads (one per partial slot): per-(scale, sample) loss term + the backward
// coefficients coef[(i*N+n)*MAXC + c] = (A, B) of the Dice gradient; terms[i*N+n] is summed by loss_total_kernel
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ partials, int nscale, int N, int C,
                                                            int blocks, LossScales sc, float smooth, float w_ce,
                                                            float w_dice, const float* __restrict__ cw, int ignore,
                                                            float* __restrict__ terms, float* __restrict__ coefA,
                                                            float* __restrict__ coefB, float* __restrict__ wsum_out) {
  __shared__ double red[4][NSTAT];
  const int i = blockIdx.x / N, n = blockIdx.x % N;
  const float* base = partials + ((int64_t)i * N + n) * blocks * NSTAT;
  double s[NSTAT];
#pragma unroll
  for (int k = 0; k < NSTAT; k++) s[k] = 0.0;
  for (int b = threadIdx.x; b < blocks; b += 256)
#pragma unroll
    for (int k = 0; k < NSTAT; k++) s[k] += (double)base[(int64_t)b * NSTAT + k];
  if (cw) {  // weighted CE: the denominator is the weight sum over ALL samples of the scale (slot NSTAT-1 of every n)
    double wall = 0.0;
    for (int m = 0; m < N; m++) {
      const float* bm = partials + ((int64_t)i * N + m) * blocks * NSTAT;
      for (int b = threadIdx.x; b < blocks; b += 256) wall += (double)bm[(int64_t)b * NSTAT + 3 * MAXC + 1];
    }
    s[3 * MAXC + 1] = wall;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NSTAT; k++) {
    double v = s[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[NSTAT];
    for (int k = 0; k < NSTAT; k++) t[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    double V = (double)sc.V[i];
    double ce = cw ? t[3 * MAXC] / t[3 * MAXC + 1] : t[3 * MAXC] / (V * N);
    if (n == 0) wsum_out[i] = cw ? (float)t[3 * MAXC + 1] : (float)(V * N);
    double dice = 0.0;
    for (int c = 0; c < C; c++) {
      float A = 0.f, B = 0.f;
      if (c != ignore) {  // dice_loss.py:75-84: every class but ignore_index, times its class weight
        const double wc = cw ? (double)cw[c] : 1.0;
        double I = t[c], U = t[MAXC + c] + t[2 * MAXC + c];
        dice += wc * (1.0 - (2.0 * I + smooth) / (U + smooth)) / (double)N;
        A = (float)(wc * 2.0 / (U + smooth));
        B = (float)(wc * (2.0 * I + smooth) / ((U + smooth) * (U + smooth)));
      }
      coefA[((int64_t)i * N + n) * MAXC + c] = A;
      coefB[((int64_t)i * N + n) * MAXC + c] = B;
    }
    dice /= (double)(ignore >= 0 ? C - 1 : C);  // dice_loss.py:84-87
    // CE is already a mean over all N samples' voxels: every (scale, n) block contributes its own share
    terms[blockIdx.x] = (float)(((double)w_ce * ce + (double)w_dice * dice) * (double)sc.weight[i]);
  }
}

__global__ void loss_total_kernel(const float* __restrict__ terms, int count, float* __restrict__ loss_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < count; k++) t += (double)terms[k];   // fixed order
    *loss_out = (float)t;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const T* __restrict__ logits, const float* __restrict__ target,
                                                       int N, int C, int Ds, int Hs, int Ws, int stride, int D, int H,
                                                       int W, const float* __restrict__ coefA,
                                                       const float* __restrict__ coefB, float weight, float w_ce,
                                                       float w_dice, const float* __restrict__ cw, int ignore,
                                                       const float* __restrict__ wsum, const float* __restrict__ gup,
                                                       T* __restrict__ dlogits) {
  const int n = blockIdx.y;
  const int64_t V = (int64_t)Ds * Hs * Ws, Vf = (int64_t)D * H * W;
  const float g = (*gup) * weight;
  const float kce0 = cw ? w_ce * g / (*wsum) : w_ce * g / ((float)V * (float)N);
  const float kd = w_dice * g / ((float)(ignore >= 0 ? C - 1 : C) * (float)N);
  float cwr[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; c++) cwr[c] = (cw && c < C) ? cw[c] : 1.f;
  float cA[MAXC], cB[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; c++) {
    cA[c] = c < C ? coefA[(int64_t)n * MAXC + c] : 0.f;
    cB[c] = c < C ? coefB[(int64_t)n * MAXC + c] : 0.f;
  }
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
    int x = v % Ws, y = (v / Ws) % Hs, z = v / ((int64_t)Ws * Hs);
    int64_t vf = ((int64_t)z * stride * H + (int64_t)y * stride) * W + (int64_t)x * stride;
    float lg[MAXC], t[MAXC];
    float mx = -INFINITY, tbest = -INFINITY;
    int tc = 0;
#pragma unroll
    for (int c = 0; c < MAXC; c++)
      if (c < C) {
        lg[c] = ST<T>::ld(logits + ((int64_t)n * C + c) * V + v);
        t[c] = target[((int64_t)n * C + c) * Vf + vf];
        mx = fmaxf(mx, lg[c]);
        if (t[c] > tbest) {
          tbest = t[c];
          tc = c;
        }
      }
    float se = 0.f, p[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; c++)
      if (c < C) {
        p[c] = __expf(lg[c] - mx);
        se += p[c];
      }
    float inv = 1.f / se, dot = 0.f, G[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; c++)
      if (c < C) {
        p[c] *= inv;
        G[c] = -kd * (cA[c] * t[c] - cB[c]);  // dDice/dp_c (the coefficients of the ignored class are zero)
        dot += G[c] * p[c];
      }
    float kce = kce0;
    if (cw) {
#pragma unroll
      for (int c = 0; c < MAXC; c++)
        if (c == tc) kce = kce0 * cwr[c];
    }
#pragma unroll
    for (int c = 0; c < MAXC; c++)
      if (c < C) {
        float d = kce * (p[c] - (c == tc ? 1.f : 0.f)) + p[c] * (G[c] - dot);
        ST<T>::st(dlogits + ((int64_t)n * C + c) * V + v, d);
      }
  }
}

// ---------------------------------------------------------------------------------- Dice metric
// counts[n][c][3] = (|P=c & T=c|, |P=c|, |T=c|) from hard argmax of logits / one-hot (trainer.py:919-945)
template <typename T>
__global__ __launch_bounds__(256) void dice_count_kernel(const T* __restrict__ logits, const float* __restrict__ target,
                                                         int C, int64_t V, unsigned long long* __restrict__ counts) {
  __shared__ unsigned int red[MAXC * 3];
  const int n = blockIdx.y;
  if (threadIdx.x < MAXC * 3) red[threadIdx.x] = 0;
  __syncthreads();
  unsigned int loc[MAXC * 3];
#pragma unroll
  for (int i = 0; i < MAXC * 3; i++) loc[i] = 0;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
    float bl = -INFINITY, bt = -INFINITY;
    int pc = 0, tc = 0;
#pragma unroll
    for (int c = 0; c < MAXC; c++)
      if (c < C) {
        float l = ST<T>::ld(logits + ((int64_t)n * C + c) * V + v);
        float t = target[((int64_t)n * C + c) * V + v];
        if (l > bl) bl = l, pc = c;
        if (t > bt) bt = t, tc = c;
      }
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
      loc[c * 3 + 0] += (pc == c && tc == c);
      loc[c * 3 + 1] += (pc == c);
      loc[c * 3 + 2] += (tc == c);
    }
  }
#pragma unroll
  for (int i = 0; i < MAXC * 3; i++) atomicAdd(&red[i], loc[i]);
  __syncthreads();
  if (threadIdx.x < C * 3) atomicAdd(counts + (int64_t)n * MAXC * 3 + threadIdx.x, (unsigned long long)red[threadIdx.x]);
}

// confusion[t][p] += #voxels with target class t and predicted class p, summed over the batch (the matrix that
// metrics.RunningDice.update_matrix builds with sklearn on the CPU, metrics.py:104-133)
template <typename T>
__global__ __launch_bounds__(256) void confusion_kernel(const T* __restrict__ logits, const float* __restrict__ target,
                                                        int C, int64_t V, unsigned long long* __restrict__ conf) {
  __shared__ unsigned int red[MAXC * MAXC];
  const int n = blockIdx.y;
  if (threadIdx.x < MAXC * MAXC) red[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
    float bl = -INFINITY, bt = -INFINITY;
    int pc = 0, tc = 0;
#pragma unroll
    for (int c = 0; c < MAXC; c++)
      if (c < C) {
        float l = ST<T>::ld(logits + ((int64_t)n * C + c) * V + v);
        float t = target[((int64_t)n * C + c) * V + v];
        if (l > bl) bl = l, pc = c;
        if (t > bt) bt = t, tc = c;
      }
    atomicAdd(&red[tc * MAXC + pc], 1u);
  }
  __syncthreads();
  if (threadIdx.x < MAXC * MAXC && red[threadIdx.x])
    atomicAdd(conf + threadIdx.x, (unsigned long long)red[threadIdx.x]);
}

// the same from two uint8 class maps (the reference's call signature: RunningDice.update_matrix(ground_truth,
// prediction), metrics.py:104); labels >= C are not counted (sklearn's confusion_matrix(labels=...) drops them)
__global__ __launch_bounds__(256) void confusion_labels_kernel(const uint8_t* __restrict__ tgt,
                                                               const uint8_t* __restrict__ pred, int C, int64_t n,
                                                               unsigned long long* __restrict__ conf) {
  __shared__ unsigned int red[MAXC * MAXC];
  if (threadIdx.x < MAXC * MAXC) red[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (int64_t)gridDim.x * 256) {
    const int tc = tgt[v], pc = pred[v];
    if (tc < C && pc < C) atomicAdd(&red[tc * MAXC + pc], 1u);
  }
  __syncthreads();
  if (threadIdx.x < MAXC * MAXC && red[threadIdx.x])
    atomicAdd(conf + threadIdx.x, (unsigned long long)red[threadIdx.x]);
}

// ---------------------------------------------------------------------------------- input normalisation
// data_utils/data_loader.py:39-68.  Per-channel reductions over the volume in a fixed order (block partials in
// fp64, then one block), then one elementwise pass.  stats[c] = (max, sum, sum of squares, unused).
constexpr int NORM_BLOCKS = 512;
__global__ __launch_bounds__(256) void norm_reduce_kernel(const float* __restrict__ img, int64_t V,
                                                          double* __restrict__ part /*[C][NORM_BLOCKS][3]*/) {
  __shared__ double red[4][3];
  const int c = blockIdx.y;
  const float* p = img + (int64_t)c * V;
  double mx = -INFINITY, s = 0.0, ss = 0.0;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
    const double x = (double)p[v];
    mx = fmax(mx, x);
    s += x;
    ss += x * x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mx = fmax(mx, __shfl_xor(mx, o, 64));
    s += __shfl_xor(s, o, 64);
    ss += __shfl_xor(ss, o, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave][0] = mx, red[wave][1] = s, red[wave][2] = ss;
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = part + ((int64_t)c * gridDim.x + blockIdx.x) * 3;
    o[0] = fmax(fmax(red[0][0], red[1][0]), fmax(red[2][0], red[3][0]));
    o[1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    o[2] = red[0][2] + red[1][2] + red[2][2] + red[3][2];
  }
}
__global__ void norm_finalize_kernel(const double* __restrict__ part, int blocks, int64_t V,
                                     double* __restrict__ stats /*[C][4]: max, mean, std (population), 0*/) {
  const int c = blockIdx.x;
  if (threadIdx.x == 0) {
    double mx = -INFINITY, s = 0.0, ss = 0.0;
    for (int b = 0; b < blocks; b++) {
      const double* q = part + ((int64_t)c * blocks + b) * 3;
      mx = fmax(mx, q[0]);
      s += q[1];
      ss += q[2];
    }
    const double mean = s / (double)V;
    stats[c * 4 + 0] = mx;
    stats[c * 4 + 1] = mean;
    stats[c * 4 + 2] = sqrt(fmax(ss / (double)V - mean * mean, 0.0));
    stats[c * 4 + 3] = 0.0;
  }
}
// mode 0 (MRNormalize, data_loader.py:39-50): x / max(channel) when the max is non-zero, then negatives -> 0.
// mode 1 (PETandCTNormalize, :53-68): channel 0 -> (clip(x, mean-w, mean+w) - mean) / w ; channel 1 -> (x - mean_1)
//         / (std_1 + 1e-3) ; further channels untouched.
__global__ void norm_apply_kernel(float* __restrict__ img, int64_t V, const double* __restrict__ stats, int mode,
                                  float pmean, float pw) {
  const int c = blockIdx.y;
  float* p = img + (int64_t)c * V;
  const float mx = (float)stats[c * 4 + 0], mean = (float)stats[c * 4 + 1], sd = (float)stats[c * 4 + 2];
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
    float x = p[v];
    if (mode == 0) {
      if (mx != 0.f) x = x / mx;
      x = x < 0.f ? 0.f : x;
    } else if (c == 0) {
      x = (fminf(fmaxf(x, pmean - pw), pmean + pw) - pmean) / pw;
    } else if (c == 1) {
      x = (x - mean) / (sd + 1e-3f);
    }
    p[v] = x;
  }
}

// ---------------------------------------------------------------------------------- Adam
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, const uint8_t* __restrict__ decay, int64_t n, float lr, float b1,
                            float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pi = p[i];
    float gi = g[i] * gscale + ((decay && decay[i]) ? wd * pi : 0.f);
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}
}  // namespace

int hdf_loss_blocks() { return LOSS_BLOCKS; }
size_t hdf_loss_workspace_floats(int N, int nscale) {
  return (size_t)nscale * N * LOSS_BLOCKS * NSTAT + 2 * (size_t)nscale * N * MAXC + (size_t)nscale * N + 16;
}

int hdf_launch_loss_fwd(int dtype, const void* const* logits, const float* target, int nscale, int N, int C, int D,
                        int H, int W, float* ws, float* loss_out, hipStream_t st, float w_ce, float w_dice,
                        const float* class_weight, int dice_ignore) {
  HDF_CHECK_ARG(C <= MAXC && C >= 2, "loss: n_cls=%d unsupported (2..%d)", C, MAXC);
  HDF_CHECK_ARG(dice_ignore >= -1 && dice_ignore < C, "loss: ignore_index %d outside [-1, %d)", dice_ignore, C);
  HDF_CHECK_ARG(nscale <= 4 && nscale * N <= 256, "loss: nscale=%d N=%d", nscale, N);
  float* partials = ws;
  float* coefA = ws + (size_t)nscale * N * LOSS_BLOCKS * NSTAT;
  float* coefB = coefA + (size_t)nscale * N * MAXC;
  LossScales sc;
  for (int i = 0; i < nscale; i++) {
    int s = 1 << i;
    // D == 1: 2-D logits [N][C][H][W] (models/HDenseFormer_2D.py); the nearest down-sampling then strides H and W only
    HDF_CHECK_ARG((D == 1 || D % s == 0) && H % s == 0 && W % s == 0, "loss: size not divisible by %d", s);
    int Ds = D == 1 ? 1 : D / s, Hs = H / s, Ws = W / s;
    sc.V[i] = (float)((int64_t)Ds * Hs * Ws);
    sc.weight[i] = 1.f / (float)s;
    float* pi = partials + (size_t)i * N * LOSS_BLOCKS * NSTAT;
    HDF_DISPATCH_T(dtype, hipLaunchKernelGGL(loss_fwd_kernel<T>, dim3(LOSS_BLOCKS, N), dim3(256), 0, st,
                                             (const T*)logits[i], target, C, Ds, Hs, Ws, s, D, H, W, class_weight, pi));
    HDF_LAUNCH_CHECK();
  }
  float* terms = coefB + (size_t)nscale * N * MAXC;
  float* wsum = terms + (size_t)nscale * N;  // [nscale]: denominator of the cross-entropy mean (the 16 spare floats)
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(nscale * N), dim3(256), 0, st, partials, nscale, N, C, LOSS_BLOCKS, sc,
                     1e-5f, w_ce, w_dice, class_weight, dice_ignore, terms, coefA, coefB, wsum);
  HDF_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(64), 0, st, terms, nscale * N, loss_out);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_loss_bwd(int dtype, const void* const* logits, const float* target, int nscale, int N, int C, int D,
                        int H, int W, const float* ws, const float* grad_out, void* const* dlogits, hipStream_t st,
                        float w_ce, float w_dice, const float* class_weight, int dice_ignore) {
  const float* coefA = ws + (size_t)nscale * N * LOSS_BLOCKS * NSTAT;
  const float* coefB = coefA + (size_t)nscale * N * MAXC;
  const float* wsum = coefB + (size_t)nscale * N * MAXC + (size_t)nscale * N;
  for (int i = 0; i < nscale; i++) {
    int s = 1 << i;
    int Ds = D == 1 ? 1 : D / s, Hs = H / s, Ws = W / s;
    int64_t V = (int64_t)Ds * Hs * Ws;
    unsigned gx = (unsigned)std::min<int64_t>(ceil_div64(V, 256), 2048);
    HDF_DISPATCH_T(dtype, hipLaunchKernelGGL(loss_bwd_kernel<T>, dim3(gx, N), dim3(256), 0, st, (const T*)logits[i],
                                             target, N, C, Ds, Hs, Ws, s, D, H, W, coefA + (size_t)i * N * MAXC,
                                             coefB + (size_t)i * N * MAXC, 1.f / (float)s, w_ce, w_dice, class_weight,
                                             dice_ignore, wsum + i, grad_out, (T*)dlogits[i]));
    HDF_LAUNCH_CHECK();
  }
  return HDF_OK;
}

int hdf_launch_dice_counts(int dtype, const void* logits, const float* target, int N, int C, int64_t V,
                           unsigned long long* counts, hipStream_t st) {
  HDF_CHECK_ARG(C <= MAXC, "dice: n_cls=%d", C);
  hipError_t e = hipMemsetAsync(counts, 0, (size_t)N * MAXC * 3 * sizeof(unsigned long long), st);
  if (e != hipSuccess) {
    hdf_set_error("dice: memset failed: %s", hipGetErrorString(e));
    return HDF_ERR_HIP;
  }
  unsigned gx = (unsigned)std::min<int64_t>(ceil_div64(V, 256), 1024);
  HDF_DISPATCH_T(dtype, hipLaunchKernelGGL(dice_count_kernel<T>, dim3(gx, N), dim3(256), 0, st, (const T*)logits, target,
                                           C, V, counts));
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_confusion(int dtype, const void* logits, const float* target, int N, int C, int64_t V,
                         unsigned long long* conf, int accumulate, hipStream_t st) {
  HDF_CHECK_ARG(C <= MAXC, "confusion: n_cls=%d", C);
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(conf, 0, (size_t)MAXC * MAXC * sizeof(unsigned long long), st);
    if (e != hipSuccess) {
      hdf_set_error("confusion: memset failed: %s", hipGetErrorString(e));
      return HDF_ERR_HIP;
    }
  }
  unsigned gx = (unsigned)std::min<int64_t>(ceil_div64(V, 256), 1024);
  HDF_DISPATCH_T(dtype, hipLaunchKernelGGL(confusion_kernel<T>, dim3(gx, N), dim3(256), 0, st, (const T*)logits, target, C,
                                           V, conf));
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_confusion_labels(const uint8_t* tgt, const uint8_t* pred, int C, int64_t n, unsigned long long* conf,
                                int accumulate, hipStream_t st) {
  HDF_CHECK_ARG(C >= 1 && C <= MAXC, "confusion: n_cls=%d", C);
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(conf, 0, (size_t)MAXC * MAXC * sizeof(unsigned long long), st);
    if (e != hipSuccess) {
      hdf_set_error("confusion: memset failed: %s", hipGetErrorString(e));
      return HDF_ERR_HIP;
    }
  }
  unsigned gx = (unsigned)std::min<int64_t>(std::max<int64_t>(ceil_div64(n, 256), 1), 1024);
  hipLaunchKernelGGL(confusion_labels_kernel, dim3(gx), dim3(256), 0, st, tgt, pred, C, n, conf);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

size_t hdf_norm_ws_bytes(int C) { return ((size_t)C * NORM_BLOCKS * 3 + (size_t)C * 4) * sizeof(double); }
int hdf_launch_normalize(float* img, int C, int64_t V, int mode, float pmean, float pw, void* ws, hipStream_t st) {
  HDF_CHECK_ARG(C >= 1 && C <= 64 && V >= 1, "normalize: channels=%d voxels=%lld", C, (long long)V);
  HDF_CHECK_ARG(mode == 0 || mode == 1, "normalize: mode %d", mode);
  HDF_CHECK_ARG(mode == 0 || (C >= 2 && pw != 0.f), "normalize: PET/CT mode needs >= 2 channels and w != 0");
  double* part = (double*)ws;
  double* stats = part + (size_t)C * NORM_BLOCKS * 3;
  hipLaunchKernelGGL(norm_reduce_kernel, dim3(NORM_BLOCKS, C), dim3(256), 0, st, img, V, part);
  HDF_LAUNCH_CHECK();
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(C), dim3(64), 0, st, part, NORM_BLOCKS, V, stats);
  HDF_LAUNCH_CHECK();
  unsigned gx = (unsigned)std::min<int64_t>(ceil_div64(V, 256), 4096);
  hipLaunchKernelGGL(norm_apply_kernel, dim3(gx, C), dim3(256), 0, st, img, V, stats, mode, pmean, pw);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

// ------------------------------------------------------------------------------ sliding-window inference tail
namespace {
constexpr int SW_MAXC = 8;
// one thread per window voxel: softmax over classes (fp32, max-subtracted like F.softmax) and accumulate
template <typename T>
__global__ void sw_accumulate_kernel(const T* __restrict__ logits, int C, int pd, int ph, int pw,
                                     float* __restrict__ psum, float* __restrict__ cnt, int D, int H, int W, int z0,
                                     int y0, int x0) {
  const int64_t pv = (int64_t)pd * ph * pw;
  const int64_t V = (int64_t)D * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pv; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % pw), y = (int)((i / pw) % ph), z = (int)(i / ((int64_t)pw * ph));
    float v[SW_MAXC], mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < SW_MAXC; c++)
      if (c < C) {
        v[c] = ST<T>::ld(logits + c * pv + i);
        mx = fmaxf(mx, v[c]);
      }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < SW_MAXC; c++)
      if (c < C) {
        v[c] = expf(v[c] - mx);
        sum += v[c];
      }
    const float inv = 1.f / sum;
    const int64_t o = ((int64_t)(z0 + z) * H + (y0 + y)) * W + (x0 + x);
#pragma unroll
    for (int c = 0; c < SW_MAXC; c++)
      if (c < C) psum[c * V + o] += v[c] * inv;
    cnt[o] += 1.f;
  }
}
__global__ void sw_finalize_kernel(const float* __restrict__ psum, const float* __restrict__ cnt, int C, int64_t V,
                                   uint8_t* __restrict__ label) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x) {
    const float n = cnt[i];
    float best = -INFINITY;
    int bi = 0;
    if (n > 0.f) {
      // argmax(softmax(p / n)): softmax is monotonic, so the vote is the first maximum of the mean probabilities
#pragma unroll
      for (int c = 0; c < SW_MAXC; c++)
        if (c < C) {
          const float m = psum[c * V + i] / n;
          if (m > best) best = m, bi = c;
        }
    }
    label[i] = (uint8_t)bi;
  }
}
__global__ void onehot_kernel(const uint8_t* __restrict__ lab, float* __restrict__ oh, int C, int64_t V) {
  const int n = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = lab[(int64_t)n * V + i];
    float* o = oh + (int64_t)n * C * V + i;
    const bool fg = l >= 1 && l < C;
    o[0] = fg ? 0.f : 1.f;
    for (int c = 1; c < C; c++) o[(int64_t)c * V] = (l == c) ? 1.f : 0.f;
  }
}
}  // namespace

int hdf_launch_sw_accumulate(int dtype, const void* logits, int C, int pd, int ph, int pw, float* psum, float* cnt,
                             int D, int H, int W, int z0, int y0, int x0, hipStream_t st) {
  HDF_CHECK_ARG(C >= 1 && C <= SW_MAXC, "sw_accumulate: n_cls=%d (max %d)", C, SW_MAXC);
  HDF_CHECK_ARG(z0 >= 0 && y0 >= 0 && x0 >= 0 && z0 + pd <= D && y0 + ph <= H && x0 + pw <= W,
                "sw_accumulate: window (%d,%d,%d)+(%d,%d,%d) outside the %dx%dx%d volume", z0, y0, x0, pd, ph, pw, D, H,
                W);
  const int64_t pv = (int64_t)pd * ph * pw;
  dim3 grid((unsigned)std::min<int64_t>(ceil_div64(pv, 256), 4096));
  HDF_DISPATCH_T(dtype, hipLaunchKernelGGL(sw_accumulate_kernel<T>, grid, dim3(256), 0, st, (const T*)logits, C, pd, ph, pw,
                                           psum, cnt, D, H, W, z0, y0, x0));
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}
int hdf_launch_sw_finalize(const float* psum, const float* cnt, int C, int64_t V, uint8_t* label, hipStream_t st) {
  HDF_CHECK_ARG(C >= 1 && C <= SW_MAXC, "sw_finalize: n_cls=%d (max %d)", C, SW_MAXC);
  dim3 grid((unsigned)std::min<int64_t>(ceil_div64(V, 256), 8192));
  hipLaunchKernelGGL(sw_finalize_kernel, grid, dim3(256), 0, st, psum, cnt, C, V, label);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}
int hdf_launch_onehot(const uint8_t* lab, float* oh, int N, int C, int64_t V, hipStream_t st) {
  HDF_CHECK_ARG(C >= 2 && C <= 255 && N >= 1, "onehot: n_cls=%d batch=%d", C, N);
  dim3 grid((unsigned)std::min<int64_t>(ceil_div64(V, 256), 4096), N);
  hipLaunchKernelGGL(onehot_kernel, grid, dim3(256), 0, st, lab, oh, C, V);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_adam(float* p, const float* g, float* m, float* v, const uint8_t* decay, int64_t n, float lr, float b1,
                    float b2, float eps, float wd, int step, float gscale, hipStream_t st) {
  float bc1 = 1.f - powf(b1, (float)step);
  float bc2s = sqrtf(1.f - powf(b2, (float)step));
  unsigned gx = (unsigned)std::min<int64_t>(ceil_div64(n, 256), 4096);
  hipLaunchKernelGGL(adam_kernel, dim3(gx), dim3(256), 0, st, p, g, m, v, decay, n, lr, b1, b2, eps, wd, bc1, bc2s,
                     gscale);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}
