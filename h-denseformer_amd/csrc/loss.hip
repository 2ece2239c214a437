// Fused DeepSuperloss(CEPlusDice) forward/backward, on-device Dice metric, flat Adam.
//
// Reference: loss/combine_loss.py:8-35,68-79 ; loss/dice_loss.py:5-87 ; loss/cross_entropy.py:8-22 ;
// metric trainer.py:891-945 ; optimizer torch.optim.Adam as built by trainer.py:793-840.
// One pass per scale reads the logits once (NCDHW, coalesced along voxels) and the fp32 one-hot target
// at the 2^i-strided positions (nearest down-sampling), producing per-(sample,class) sums
// sum(p*t), sum(p), sum(t) and the CE sum; backward recomputes the softmax from the logits.
#include <type_traits>

#include "loss.h"

namespace {
constexpr int MAXC = 8;
constexpr int LOSS_BLOCKS = 1024;
constexpr int NSTAT = 3 * MAXC + 2;  // per class: sum p*t, sum p, sum t; then the (weighted) CE sum and the sum of CE weights

// All scales of the deep supervision in ONE launch: grid (sum of the scales' block counts, N); a block finds its scale from
// the block ranges in LossLevels.  A thread owns VEC = 4 consecutive voxels of a row of the scale's grid (one 8-byte load
// of 16-bit logits and -- at scale 0 -- one 16-byte load of the one-hot target per class: the scalar form had 2 + 4 bytes
// per lane and class in flight and ran at 1.1 TB/s); rows whose width is not a multiple of 4, or unaligned views, take
// VEC = 1.  The sums of a thread are taken in voxel order, of a block in a fixed order: bitwise reproducible.
struct LossLevel {
  const void* logits;
  void* dlogits;
  int Ds, Hs, Ws, stride;
  int blk0, nblk, vec;
};
struct LossLevels {
  LossLevel L[4];
  int nscale;
};

template <typename T>
__device__ __forceinline__ void ld_vox(const T* p, float* f, std::integral_constant<int, 1>) {
  f[0] = ST<T>::ld(p);
}
template <typename T>
__device__ __forceinline__ void ld_vox(const T* p, float* f, std::integral_constant<int, 4>) {
  ST<T>::ld4(p, f);
}
template <typename T>
__device__ __forceinline__ void st_vox(T* p, const float* f, std::integral_constant<int, 1>) {
  ST<T>::st(p, f[0]);
}
template <typename T>
__device__ __forceinline__ void st_vox(T* p, const float* f, std::integral_constant<int, 4>) {
  ST<T>::st4(p, f[0], f[1], f[2], f[3]);
}
// one-hot target of VEC voxels that are `stride` apart in the full-resolution row
template <int VEC>
__device__ __forceinline__ void ld_tgt(const float* p, int stride, float* t) {
  if (VEC == 4 && stride == 1) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    t[0] = v[0], t[1] = v[1], t[2] = v[2], t[3] = v[3];
  } else {
#pragma unroll
    for (int j = 0; j < VEC; j++) t[j] = p[(int64_t)j * stride];
  }
}

template <typename T, int VEC, int MC>
__device__ __forceinline__ void loss_fwd_body(const LossLevel& L, int bx, const float* __restrict__ target, int C, int D,
                                              int H, int W, const float* __restrict__ cw, float* acc) {
  const T* logits = reinterpret_cast<const T*>(L.logits);
  const int n = blockIdx.y, Ws = L.Ws, Hs = L.Hs, stride = L.stride;
  const int64_t V = (int64_t)L.Ds * Hs * Ws, Vf = (int64_t)D * H * W;
  const std::integral_constant<int, VEC> vt{};
  // (32-bit voxel indices inside a sample: the launcher checks D*H*W < 2^31; at scale 0 the two grids coincide)
  for (int v = (bx * 256 + (int)threadIdx.x) * VEC; v < (int)V; v += L.nblk * 256 * VEC) {
    int64_t vf = v;
    if (stride > 1) {
      const int x = v % Ws, y = (v / Ws) % Hs, z = v / (Ws * Hs);
      vf = ((int64_t)z * stride * H + (int64_t)y * stride) * W + (int64_t)x * stride;
    }
    float lg[MC][VEC], t[MC][VEC];
#pragma unroll
    for (int c = 0; c < MC; c++)
      if (c < C) {
        ld_vox(logits + ((int64_t)n * C + c) * V + v, lg[c], vt);
        ld_tgt<VEC>(target + ((int64_t)n * C + c) * Vf + vf, stride, t[c]);
      }
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      float mx = -INFINITY, tbest = -INFINITY;
      int tc = 0;
#pragma unroll
      for (int c = 0; c < MC; c++)
        if (c < C) {
          mx = fmaxf(mx, lg[c][j]);
          if (t[c][j] > tbest) {
            tbest = t[c][j];
            tc = c;
          }
        }
      float se = 0.f, e[MC];
#pragma unroll
      for (int c = 0; c < MC; c++)
        if (c < C) {
          e[c] = __expf(lg[c][j] - mx);
          se += e[c];
        }
      const float inv = 1.f / se;
      const float lse = mx + __logf(se);
#pragma unroll
      for (int c = 0; c < MC; c++)
        if (c < C) {
          const float p = e[c] * inv;
          acc[c] += p * t[c][j];
          acc[MAXC + c] += p;
          acc[2 * MAXC + c] += t[c][j];
          if (c == tc) {
            // torch CrossEntropyLoss(weight=w, reduction='mean'): sum_v w[t_v] * nll_v / sum_v w[t_v]
            const float wv = cw ? cw[c] : 1.f;
            acc[3 * MAXC] += cw ? wv * (lse - lg[c][j]) : lse - lg[c][j];
            acc[3 * MAXC + 1] += wv;
          }
        }
    }
  }
}

template <typename T, int MC>  // MC: class slots held in registers (4 or 8)
__global__ __launch_bounds__(256) void loss_fwd_kernel(LossLevels lv, const float* __restrict__ target, int C, int D,
                                                       int H, int W,
                                                       const float* __restrict__ cw /*[C] class weights or null*/,
                                                       float* __restrict__ partials /*[scale][N][LOSS_BLOCKS][NSTAT]*/) {
  __shared__ float red[4][NSTAT];
  int i = 0;
#pragma unroll
  for (int k = 1; k < 4; k++)
    if (k < lv.nscale && (int)blockIdx.x >= lv.L[k].blk0) i = k;
  const LossLevel& L = lv.L[i];
  const int bx = blockIdx.x - L.blk0;
  float acc[NSTAT];
#pragma unroll
  for (int k = 0; k < NSTAT; k++) acc[k] = 0.f;
  if (L.vec == 4)
    loss_fwd_body<T, 4, MC>(L, bx, target, C, D, H, W, cw, acc);
  else
    loss_fwd_body<T, 1, MC>(L, bx, target, C, D, H, W, cw, acc);
#pragma unroll
  for (int k = 0; k < NSTAT; k++) acc[k] = wave_sum(acc[k]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NSTAT; k++) red[wave][k] = acc[k];
  __syncthreads();
  if (threadIdx.x < NSTAT)
    partials[(((int64_t)i * gridDim.y + blockIdx.y) * LOSS_BLOCKS + bx) * NSTAT + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// grid = nscale*N blocks of 256 threads (one per partial slot): per-(scale, sample) loss term + the backward
// coefficients coef[(i*N+n)*MAXC + c] = (A, B) of the Dice gradient; terms[i*N+n] is summed by loss_total_kernel
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ partials, int nscale, int N, int C,
                                                            LossScales sc, float smooth, float w_ce,
                                                            float w_dice, const float* __restrict__ cw, int ignore,
                                                            float* __restrict__ terms, float* __restrict__ coefA,
                                                            float* __restrict__ coefB, float* __restrict__ wsum_out) {
  __shared__ double red[4][NSTAT];
  const int i = blockIdx.x / N, n = blockIdx.x % N, blocks = sc.blocks[i];
  const float* base = partials + ((int64_t)i * N + n) * LOSS_BLOCKS * NSTAT;
  double s[NSTAT];
#pragma unroll
  for (int k = 0; k < NSTAT; k++) s[k] = 0.0;
  for (int b = threadIdx.x; b < blocks; b += 256)
#pragma unroll
    for (int k = 0; k < NSTAT; k++) s[k] += (double)base[(int64_t)b * NSTAT + k];
  if (cw) {  // weighted CE: the denominator is the weight sum over ALL samples of the scale (slot NSTAT-1 of every n)
    double wall = 0.0;
    for (int m = 0; m < N; m++) {
      const float* bm = partials + ((int64_t)i * N + m) * LOSS_BLOCKS * NSTAT;
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

struct LossBwdP {
  const float *coefA, *coefB;  // [scale][N][MAXC]
  const float* wsum;           // [scale]
  const float* gup;
  const float* cw;
  float w_ce, w_dice;
  int ignore;
};

template <typename T, int VEC, int MC>
__device__ __forceinline__ void loss_bwd_body(const LossLevel& L, int i, int bx, const float* __restrict__ target, int N,
                                              int C, int D, int H, int W, const LossBwdP& q) {
  const T* logits = reinterpret_cast<const T*>(L.logits);
  T* dlogits = reinterpret_cast<T*>(L.dlogits);
  const int n = blockIdx.y, Ws = L.Ws, Hs = L.Hs, stride = L.stride;
  const int64_t V = (int64_t)L.Ds * Hs * Ws, Vf = (int64_t)D * H * W;
  const float g = (*q.gup) / (float)stride;   // scale weight 1 / 2^i (combine_loss.py:68-79)
  const float kce0 = q.cw ? q.w_ce * g / q.wsum[i] : q.w_ce * g / ((float)V * (float)N);
  const float kd = q.w_dice * g / ((float)(q.ignore >= 0 ? C - 1 : C) * (float)N);
  float cwr[MC], cA[MC], cB[MC];
#pragma unroll
  for (int c = 0; c < MC; c++) {
    cwr[c] = (q.cw && c < C) ? q.cw[c] : 1.f;
    cA[c] = c < C ? q.coefA[((int64_t)i * N + n) * MAXC + c] : 0.f;
    cB[c] = c < C ? q.coefB[((int64_t)i * N + n) * MAXC + c] : 0.f;
  }
  const std::integral_constant<int, VEC> vt{};
  // (32-bit voxel indices inside a sample: the launcher checks D*H*W < 2^31; at scale 0 the two grids coincide)
  for (int v = (bx * 256 + (int)threadIdx.x) * VEC; v < (int)V; v += L.nblk * 256 * VEC) {
    int64_t vf = v;
    if (stride > 1) {
      const int x = v % Ws, y = (v / Ws) % Hs, z = v / (Ws * Hs);
      vf = ((int64_t)z * stride * H + (int64_t)y * stride) * W + (int64_t)x * stride;
    }
    float lg[MC][VEC], t[MC][VEC];
#pragma unroll
    for (int c = 0; c < MC; c++)
      if (c < C) {
        ld_vox(logits + ((int64_t)n * C + c) * V + v, lg[c], vt);
        ld_tgt<VEC>(target + ((int64_t)n * C + c) * Vf + vf, stride, t[c]);
      }
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      float mx = -INFINITY, tbest = -INFINITY;
      int tc = 0;
#pragma unroll
      for (int c = 0; c < MC; c++)
        if (c < C) {
          mx = fmaxf(mx, lg[c][j]);
          if (t[c][j] > tbest) {
            tbest = t[c][j];
            tc = c;
          }
        }
      float se = 0.f, p[MC];
#pragma unroll
      for (int c = 0; c < MC; c++)
        if (c < C) {
          p[c] = __expf(lg[c][j] - mx);
          se += p[c];
        }
      float inv = 1.f / se, dot = 0.f, G[MC];
#pragma unroll
      for (int c = 0; c < MC; c++)
        if (c < C) {
          p[c] *= inv;
          G[c] = -kd * (cA[c] * t[c][j] - cB[c]);  // dDice/dp_c (the coefficients of the ignored class are zero)
          dot += G[c] * p[c];
        }
      float kce = kce0;
      if (q.cw) {
#pragma unroll
        for (int c = 0; c < MC; c++)
          if (c == tc) kce = kce0 * cwr[c];
      }
#pragma unroll
      for (int c = 0; c < MC; c++)
        if (c < C) lg[c][j] = kce * (p[c] - (c == tc ? 1.f : 0.f)) + p[c] * (G[c] - dot);
    }
#pragma unroll
    for (int c = 0; c < MC; c++)
      if (c < C) st_vox(dlogits + ((int64_t)n * C + c) * V + v, lg[c], vt);
  }
}

template <typename T, int MC>
__global__ __launch_bounds__(256) void loss_bwd_kernel(LossLevels lv, const float* __restrict__ target, int N, int C,
                                                       int D, int H, int W, LossBwdP q) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < 4; k++)
    if (k < lv.nscale && (int)blockIdx.x >= lv.L[k].blk0) i = k;
  const LossLevel& L = lv.L[i];
  const int bx = blockIdx.x - L.blk0;
  if (L.vec == 4)
    loss_bwd_body<T, 4, MC>(L, i, bx, target, N, C, D, H, W, q);
  else
    loss_bwd_body<T, 1, MC>(L, i, bx, target, N, C, D, H, W, q);
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

// geometry of the scales: nearest down-sampling by 2^i (combine_loss.py:68-79); D == 1: 2-D logits [N][C][H][W]
// (models/HDenseFormer_2D.py), the down-sampling then strides H and W only
static int loss_levels(const void* const* logits, void* const* dlogits, const float* target, int nscale, int D, int H,
                       int W, LossLevels& lv, LossScales& sc, int& blocks) {
  lv.nscale = nscale;
  int blk = 0;
  for (int i = 0; i < nscale; i++) {
    const int s = 1 << i;
    HDF_CHECK_ARG((D == 1 || D % s == 0) && H % s == 0 && W % s == 0, "loss: size not divisible by %d", s);
    HDF_CHECK_ARG((int64_t)D * H * W < ((int64_t)1 << 31) - 4096 * 256, "loss: %dx%dx%d voxels per sample", D, H, W);
    LossLevel& L = lv.L[i];
    L.logits = logits[i];
    L.dlogits = dlogits ? dlogits[i] : nullptr;
    L.Ds = D == 1 ? 1 : D / s, L.Hs = H / s, L.Ws = W / s, L.stride = s;
    const int64_t V = (int64_t)L.Ds * L.Hs * L.Ws;
    const uintptr_t al = reinterpret_cast<uintptr_t>(L.logits) | reinterpret_cast<uintptr_t>(L.dlogits) |
                         reinterpret_cast<uintptr_t>(target);
    L.vec = (L.Ws % 4 == 0 && (al & 15) == 0) ? 4 : 1;
    L.blk0 = blk;
    L.nblk = (int)std::min<int64_t>(std::max<int64_t>(ceil_div64(V, 256 * L.vec), 1), LOSS_BLOCKS);
    blk += L.nblk;
    sc.V[i] = (float)V;
    sc.weight[i] = 1.f / (float)s;
    sc.blocks[i] = L.nblk;
  }
  for (int i = nscale; i < 4; i++) lv.L[i] = lv.L[0], sc.V[i] = 0.f, sc.weight[i] = 0.f, sc.blocks[i] = 0;
  blocks = blk;
  return HDF_OK;
}

int hdf_launch_loss_fwd(int dtype, const void* const* logits, const float* target, int nscale, int N, int C, int D,
                        int H, int W, float* ws, float* loss_out, hipStream_t st, float w_ce, float w_dice,
                        const float* class_weight, int dice_ignore) {
  HDF_CHECK_ARG(C <= MAXC && C >= 2, "loss: n_cls=%d unsupported (2..%d)", C, MAXC);
  HDF_CHECK_ARG(dice_ignore >= -1 && dice_ignore < C, "loss: ignore_index %d outside [-1, %d)", dice_ignore, C);
  HDF_CHECK_ARG(nscale >= 1 && nscale <= 4 && nscale * N <= 256, "loss: nscale=%d N=%d", nscale, N);
  float* partials = ws;
  float* coefA = ws + (size_t)nscale * N * LOSS_BLOCKS * NSTAT;
  float* coefB = coefA + (size_t)nscale * N * MAXC;
  LossLevels lv;
  LossScales sc;
  int blocks = 0;
  HDF_TRY(loss_levels(logits, nullptr, target, nscale, D, H, W, lv, sc, blocks));
  HDF_DISPATCH_T(dtype, {
    if (C <= 4)
      hipLaunchKernelGGL((loss_fwd_kernel<T, 4>), dim3(blocks, N), dim3(256), 0, st, lv, target, C, D, H, W, class_weight,
                         partials);
    else
      hipLaunchKernelGGL((loss_fwd_kernel<T, MAXC>), dim3(blocks, N), dim3(256), 0, st, lv, target, C, D, H, W,
                         class_weight, partials);
  });
  HDF_LAUNCH_CHECK();
  float* terms = coefB + (size_t)nscale * N * MAXC;
  float* wsum = terms + (size_t)nscale * N;  // [nscale]: denominator of the cross-entropy mean (the 16 spare floats)
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(nscale * N), dim3(256), 0, st, partials, nscale, N, C, sc, 1e-5f, w_ce,
                     w_dice, class_weight, dice_ignore, terms, coefA, coefB, wsum);
  HDF_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(64), 0, st, terms, nscale * N, loss_out);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_launch_loss_bwd(int dtype, const void* const* logits, const float* target, int nscale, int N, int C, int D,
                        int H, int W, const float* ws, const float* grad_out, void* const* dlogits, hipStream_t st,
                        float w_ce, float w_dice, const float* class_weight, int dice_ignore) {
  HDF_CHECK_ARG(nscale >= 1 && nscale <= 4, "loss: nscale=%d", nscale);
  LossBwdP q;
  q.coefA = ws + (size_t)nscale * N * LOSS_BLOCKS * NSTAT;
  q.coefB = q.coefA + (size_t)nscale * N * MAXC;
  q.wsum = q.coefB + (size_t)nscale * N * MAXC + (size_t)nscale * N;
  q.gup = grad_out, q.cw = class_weight, q.w_ce = w_ce, q.w_dice = w_dice, q.ignore = dice_ignore;
  LossLevels lv;
  LossScales sc;
  int blocks = 0;
  HDF_TRY(loss_levels(logits, dlogits, target, nscale, D, H, W, lv, sc, blocks));
  HDF_DISPATCH_T(dtype, {
    if (C <= 4)
      hipLaunchKernelGGL((loss_bwd_kernel<T, 4>), dim3(blocks, N), dim3(256), 0, st, lv, target, N, C, D, H, W, q);
    else
      hipLaunchKernelGGL((loss_bwd_kernel<T, MAXC>), dim3(blocks, N), dim3(256), 0, st, lv, target, N, C, D, H, W, q);
  });
  HDF_LAUNCH_CHECK();
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
