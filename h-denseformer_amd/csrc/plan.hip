// Model plan / executor: owns the parameter table (= the reference state_dict, HDenseFormer.py:178-227),
// the workspace layout and the forward/backward launch sequences of HDenseFormer.forward
// (HDenseFormer.py:229-255) and its autograd.  Host-side C++; every device buffer is caller-owned.
#include <atomic>
#include <algorithm>
#include <cstdarg>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/hdf.h"
#include "conv_igemm.h"
#include "loss.h"
#include "transformer.h"
#include "unet_ops.h"

static thread_local char g_err[1024] = "";
// process-wide (relaxed atomic): forward runs on the caller's thread and backward on autograd's worker thread, and both
// must size their grids -- and take their split-K decisions -- from the same value
static std::atomic<int> g_cu_budget{256};
// The budget is capped by what the CURRENT DEVICE has (round 6): on a partitioned MI355X (DPX / QPX / CPX: 128 /
// 64 / 32 compute units) or any smaller gfx950 part a literal 256 let the persistent transformer kernels -- whose
// per-sequence barriers need every workgroup resident at once -- launch a grid that could never be resident together.
// hdf_set_cu_budget stays a DOWNWARD override.  No device (CPU-only layout queries): the literal.
static int device_cus() {
  static std::atomic<int> cache[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  int v = cache[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
  if (n >= 8) n &= ~7;   // grids of the persistent conv kernels are split over the 8 XCDs
  cache[dev].store(n, std::memory_order_relaxed);
  return n;
}
int hdf_cu_budget() { return std::min(g_cu_budget.load(std::memory_order_relaxed), device_cus()); }

void hdf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {

struct ParamInfo {
  std::string name;
  std::vector<int64_t> shape;
  int64_t offset, numel;
};

struct View {  // channels-last view into the workspace
  size_t off = 0;
  int64_t pitch = 0;
  int C = 0;
  int lvl = 0;
};

struct Stats {  // per conv layer InstanceNorm statistics, each [B][C] floats
  size_t mean = 0, rstd = 0, scale = 0, shift = 0;
};

struct Conv3 {  // 3x3x3 conv + InstanceNorm (+ReLU)
  std::string name;
  int Cin = 0, CinP = 0, Cout = 0, lvl = 0;
  int64_t w = -1, b = -1, gamma = -1, beta = -1;
  View y;
  Stats st;
  size_t wf = 0, wd = 0;  // packed forward / dgrad weights
  int wf_frag = 0, wd_frag = 0;  // their layout (hdf_conv_weight_layout of the launch that reads them)
};
struct ConvT3 {
  std::string name;
  int Cin = 0, Cout = 0, lvl_in = 0;
  int64_t w = -1, b = -1;
  size_t wf = 0, wd = 0;
  int wf_frag = 0, wd_frag = 0;
};
struct Head1 {
  std::string name;
  int C = 0, lvl = 0;
  int64_t w = -1, b = -1;
};

struct Bump {
  size_t cur = 0;
  size_t take(size_t bytes) {
    size_t o = cur;
    cur += (bytes + 255) & ~(size_t)255;
    return o;
  }
};

}  // namespace

// ---------------------------------------------------------------------------------------------- 2-D embedding
// HDenseFormer_2D (reference models/HDenseFormer_2D.py:172-250) is the 3-D graph with 2-D primitives.  It equals,
// EXACTLY, the 3-D network applied to the image replicated along a depth axis of 16 when its parameters are embedded as
//   Conv2d [o,i,3,3]            -> Conv3d [o,i,3,3,3]   with the 2-D kernel on depth tap 1, zeros on taps 0 and 2
//   ConvTranspose2d [i,o,3,3]   -> ConvTranspose3d      with the 2-D kernel on depth taps 1 AND 2 (output slice 2z
//                                  takes tap 1 of input slice z, slice 2z+1 takes tap 2 of the same slice), zero on tap 0
//   patch Conv2d [c,1,16,16]    -> Conv3d [c,1,16,16,16] with the 2-D kernel on depth slice 0, zeros elsewhere
//   everything else             -> unchanged
// Every activation then consists of identical depth slices (InstanceNorm statistics, MaxPool3d, trilinear x2 and the
// token grid all reduce to their 2-D forms), the 2-D logits are depth slice 0 of the 3-D logits, and by the chain
// rule the 2-D parameter gradient is the sum of the 3-D gradient over the embedded positions.  The cost is the 16
// (at level 0) .. 2 (level 3) redundant slices; a native depth-1 mode of the pooling / up-sampling / transposed-conv
// kernels would remove it.
struct Embed2dJob {
  int64_t off3, off2;  // float offsets in the 3-D / 2-D flat parameter (or gradient) buffers
  int64_t n3;          // 3-D elements of the job
  int inner;           // elements of one 2-D kernel (9, 256) or 1
  short rep;           // depth taps / slices of the 3-D kernel (3, 16) or 1
  char kind;           // 0 copy, 1 conv (tap 1), 2 transposed conv (taps 1 and 2), 3 patch (slice 0)
  char stage;          // backward stage bit (1 U-Net, 2 UpConv chain, 4 transformer) whose gradients it carries
};
constexpr int HDF_MAX_EMBED_JOBS = 96;
struct Embed2dBatch {
  Embed2dJob j[HDF_MAX_EMBED_JOBS];
};
__device__ __forceinline__ bool embed_live(int kind, int z) {
  return kind == 0 || (kind == 1 && z == 1) || (kind == 2 && (z == 1 || z == 2)) || (kind == 3 && z == 0);
}
// 3-D parameters from the 2-D ones (grid (blocks, jobs))
__global__ void embed2d_kernel(Embed2dBatch b, const float* __restrict__ p2, float* __restrict__ p3) {
  const Embed2dJob& jb = b.j[blockIdx.y];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < jb.n3; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t outer = e / ((int64_t)jb.rep * jb.inner);
    const int z = (int)((e / jb.inner) % jb.rep), r = (int)(e % jb.inner);
    p3[jb.off3 + e] = embed_live(jb.kind, z) ? p2[jb.off2 + outer * jb.inner + r] : 0.f;
  }
}
// 2-D gradients from the 3-D ones: the transpose of the embedding (sum over the embedded positions)
__global__ void extract2d_kernel(Embed2dBatch b, const float* __restrict__ g3, float* __restrict__ g2) {
  const Embed2dJob& jb = b.j[blockIdx.y];
  const int64_t n2 = jb.n3 / jb.rep;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n2; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t outer = e / jb.inner;
    const int r = (int)(e % jb.inner);
    float s = 0.f;
    for (int z = 0; z < jb.rep; z++)
      if (embed_live(jb.kind, z)) s += g3[jb.off3 + (outer * jb.rep + z) * jb.inner + r];
    g2[jb.off2 + e] = s;
  }
}
// x [rows][HW] -> [rows][reps][HW]
__global__ void replicate_depth_kernel(const float* __restrict__ x2, float* __restrict__ x3, int64_t rows, int reps,
                                       int64_t hw) {
  const int64_t total = rows * reps * hw;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    x3[e] = x2[(e / (reps * hw)) * hw + e % hw];
}
// depth slice 0 of [rows][reps][HW] -> [rows][HW]  (to2d) or its transpose: slice 0 <- src, other slices <- 0
template <typename T>
__global__ void depth_slice_kernel(T* __restrict__ t3, T* __restrict__ t2, int64_t rows, int reps, int64_t hw,
                                   int to2d) {
  if (to2d) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < rows * hw; e += (int64_t)gridDim.x * blockDim.x)
      t2[e] = t3[(e / hw) * reps * hw + e % hw];
  } else {
    T zero;
    zero.v = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < rows * reps * hw;
         e += (int64_t)gridDim.x * blockDim.x) {
      const int64_t row = e / (reps * hw), rem = e % (reps * hw);
      t3[e] = rem < hw ? t2[row * hw + rem] : zero;
    }
  }
}
struct f32w {  // float wrapper with the .v member the 16-bit storage structs have
  float v;
};

struct hdf_plan {
  int M, ncls, nf, D, H, W, td, nb, dtype;
  int esz;
  int dims[5][3];
  int DM, DMF, Ntok;
  std::vector<ParamInfo> params;
  std::map<std::string, int64_t> pidx;
  int64_t total_floats = 0;
  int64_t mstride = 0;
  // layers
  Conv3 deep, up[3], enc[4][2], dec[3][2];  // dec[k]: level k (0..2) right blocks
  ConvT3 upc[3];                            // upc[k] produces level k from level k+1
  Head1 head[4];
  std::vector<PackJob> pack_jobs;           // every conv's forward and dgrad weight pack (one launch per forward)
  // layout for the current batch
  int batch = -1;
  size_t ws_bytes = 0;
  size_t ws_fwd_bytes = 0;  // prefix of the workspace a forward-only (inference) call touches
  std::map<std::string, View> bufs;
  // forward buffers
  View xin, attnall, attnout, at[3] /*at[k] lives at level k*/, cat[3], pooled[3], x4;
  size_t pool_idx[3];
  size_t tf_F, tf_save, tf_scratch, tf_dF, tf_tape = 0, tf_otape = 0;
  size_t tf_sync = 0;   // arrival counters of the persistent transformer kernels (transformer_chain.hip)
  size_t tf_frag = 0;   // operand records the forward leaves for the attention backward (16-bit modes)
  size_t tf_wpack = 0;  // fragment-major copies of the dense layers' weight matrices for those kernels
  size_t stat_partials, wgrad_ws, inb_partials, inb_k;
  size_t stat_partials2 = 0, inb_partials2 = 0, inb_k2 = 0;  // the same scratch for the branch stream (see Exec::branch)
  size_t inb_k3 = 0;  // k1 / ka / kb of the first layer's InstanceNorm backward: read by its weight gradient on the SIDE stream,
                      // i.e. possibly after the caller's stream has run the next in_backward (which reuses inb_k)
  size_t ksplit_ws = 0, ksplit_ws2 = 0;                      // split-K partial tiles of the low-resolution convs, per stream
  size_t wgrad_ws_bytes = 0;
  // backward scratch
  View gA[4], gY[4], gY2[4], dCat[3], dUp[3], dSkip[3], dP[3], dUa[4], dUy[4], dX4, dAttnall;
  // Side stream of the backward pass (weight gradients; see Exec::wgrad_stream) and a ring of its events.  Created
  // lazily on first use, destroyed with the plan.
  hipStream_t side = nullptr;
  // Branch stream: the multi-path transformer + UpConv chain (forward), their backward (HDenseFormer.py:230-235), next
  // to the level-0 encoder convolutions the caller's stream runs meanwhile (forward3d / backward3d).
  hipStream_t branch = nullptr;
  std::vector<hipEvent_t> events;
  size_t ev_next = 0;
  // "gradient bucket k is final" (hdf_backward_events): recorded on whichever stream of the call finishes the bucket
  hipEvent_t bucket_ev[HDF_NUM_GRAD_BUCKETS] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // hdf_plan_set_probe: caller-owned events recorded around the dominant conv launch of the forward (measurement only)
  hipEvent_t probe_start = nullptr, probe_stop = nullptr;
  // Persistent transformer kernels (transformer_chain.hip).  chain_flag: one host-mapped word a launch writes (system
  // scope) when one of its per-sequence barriers gives up; read by the next forward / backward call of the plan without
  // synchronising (chain_flag_check).  chain_off: sticky -- after a give-up the plan runs the launch chain.
  // tf_fwd_chain: which arrangement the LAST forward ran; its backward follows it (the operand records and the
  // fragment-major weight copies of the persistent backward exist only behind a persistent forward).
  unsigned* chain_flag = nullptr;      // host address
  unsigned* chain_flag_dev = nullptr;  // device address of the same word
  bool chain_off = false;
  bool chain_forced = false;           // hdf_plan_force_persistent (tests): skip the residency check
  bool tf_fwd_chain = false;
  bool tf_bwd_chain = false;           // the last backward ran the persistent kernel (its timeout word is valid)
  unsigned chain_last_giveup = 0;      // 1 + workgroup id of the last give-up seen (hdf_plan_chain_state)
  unsigned chain_ticks = 150000000u;   // deadline of one barrier wait, 100 MHz ticks (hdf_plan_set_chain_timeout_us)
  ~hdf_plan() {
    if (chain_flag) (void)hipHostFree(chain_flag);
    for (hipEvent_t ev : bucket_ev)
      if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : events) (void)hipEventDestroy(ev);
    if (side) (void)hipStreamDestroy(side);
    if (branch) (void)hipStreamDestroy(branch);
  }
  bool dcat_split[3] = {false, false, false};
  // ---- 2-D model (models/HDenseFormer_2D.py) run as its exact depth-replicated 3-D embedding (see embed2d below)
  bool is2d = false;
  // Round 6: the 2-D model runs NATIVELY on depth-1 tensors (flat): every level has depth 1, the convolutions / transposed
  // convolutions / weight gradients are the FLAT instantiations of conv_igemm.hip (centre-plane taps of the embedded 27-tap
  // panels), pooling and up-sampling their 2-D forms (unet_ops.hip), the patch embedding contracts depth slice 0 of the
  // embedded 16^3 kernels with the input's 16 x 16 patches (K = 256).  flat = false keeps the depth-16
  // replicated embedding of rounds 3-5 (hdf_plan_create_2d_embedded: the oracle of tests/test_gpu_model_2d.py).
  bool flat = false;
  std::vector<ParamInfo> params2d;  // the 2-D reference state_dict: conv kernels [..,3,3], patch kernels [..,16,16]
  int64_t total_floats2d = 0;
  std::vector<Embed2dJob> ejobs;
  size_t e_x3d = 0, e_params3d = 0, e_grads3d = 0, e_out3d[4] = {0, 0, 0, 0}, e_dout3d[4] = {0, 0, 0, 0};
  // state carried from forward to backward
  int training = 0;
  uint32_t seed = 0;

  int64_t vox(int lvl) const { return (int64_t)dims[lvl][0] * dims[lvl][1] * dims[lvl][2]; }
  int64_t P(const std::string& n) const {
    auto it = pidx.find(n);
    return it == pidx.end() ? -1 : params[it->second].offset;
  }
};

namespace {

void add_param(hdf_plan* p, const std::string& name, std::vector<int64_t> shape) {
  ParamInfo pi;
  pi.name = name;
  pi.shape = shape;
  pi.numel = 1;
  for (auto s : shape) pi.numel *= s;
  pi.offset = p->total_floats;
  p->total_floats += (pi.numel + 15) / 16 * 16;
  p->pidx[name] = (int64_t)p->params.size();
  p->params.push_back(pi);
}

void build_params(hdf_plan* p) {
  const int nf = p->nf, DM = p->DM;
  char buf[256];
  for (int m = 0; m < p->M; m++) {
    int64_t start = p->total_floats;
    auto nm = [&](const char* fmt, ...) {
      va_list ap;
      va_start(ap, fmt);
      vsnprintf(buf, sizeof(buf), fmt, ap);
      va_end(ap);
      return std::string("attns.") + std::to_string(m) + "." + buf;
    };
    add_param(p, nm("position_embeddings"), {1, p->Ntok, DM});
    add_param(p, nm("patch_embeddings.weight"), {DM, 1, 16, 16, 16});
    add_param(p, nm("patch_embeddings.bias"), {DM});
    for (int b = 0; b < p->nb; b++) {
      for (int l = 0; l < 4; l++) {
        add_param(p, nm("blocks.%d.0.layers.%d.0.weight", b, l), {32, DM + 32 * l});
        add_param(p, nm("blocks.%d.0.layers.%d.0.bias", b, l), {32});
        add_param(p, nm("blocks.%d.0.layers.%d.1.norm.weight", b, l), {32});
        add_param(p, nm("blocks.%d.0.layers.%d.1.norm.bias", b, l), {32});
        add_param(p, nm("blocks.%d.0.layers.%d.1.fn.to_qkv.weight", b, l), {96, 32});
        add_param(p, nm("blocks.%d.0.layers.%d.1.fn.to_out.0.weight", b, l), {32, 32});
        add_param(p, nm("blocks.%d.0.layers.%d.1.fn.to_out.0.bias", b, l), {32});
        add_param(p, nm("blocks.%d.0.layers.%d.2.norm.weight", b, l), {32});
        add_param(p, nm("blocks.%d.0.layers.%d.2.norm.bias", b, l), {32});
        add_param(p, nm("blocks.%d.0.layers.%d.2.fn.net.0.weight", b, l), {64, 32});
        add_param(p, nm("blocks.%d.0.layers.%d.2.fn.net.0.bias", b, l), {64});
        add_param(p, nm("blocks.%d.0.layers.%d.2.fn.net.3.weight", b, l), {32, 64});
        add_param(p, nm("blocks.%d.0.layers.%d.2.fn.net.3.bias", b, l), {32});
      }
      add_param(p, nm("blocks.%d.0.out_layer.net.0.weight", b), {64, DM + 128});
      add_param(p, nm("blocks.%d.0.out_layer.net.0.bias", b), {64});
      add_param(p, nm("blocks.%d.0.out_layer.net.3.weight", b), {DM, 64});
      add_param(p, nm("blocks.%d.0.out_layer.net.3.bias", b), {DM});
    }
    if (m == 0) p->mstride = p->total_floats - start;
  }
  auto upc = [&](const std::string& n, int ci, int co) {
    add_param(p, n + ".double_conv.0.weight", {co, ci, 3, 3, 3});
    add_param(p, n + ".double_conv.0.bias", {co});
  };
  auto basic = [&](const std::string& n, int ci, int co) {
    add_param(p, n + ".conv.weight", {co, ci, 3, 3, 3});
    add_param(p, n + ".norm.weight", {co});
    add_param(p, n + ".norm.bias", {co});
  };
  auto convt = [&](const std::string& n, int ci, int co) {
    add_param(p, n + ".weight", {ci, co, 3, 3, 3});
    add_param(p, n + ".bias", {co});
  };
  auto head = [&](const std::string& n, int ci) {
    add_param(p, n + ".weight", {p->ncls, ci, 1, 1, 1});
    add_param(p, n + ".bias", {p->ncls});
  };
  upc("deep_conv", DM * p->M, 8 * nf);
  upc("up1", 8 * nf, 4 * nf);
  upc("up2", 4 * nf, 2 * nf);
  upc("up3", 2 * nf, nf);
  basic("block_1_1_left", p->M, nf);
  basic("block_1_2_left", nf, nf);
  basic("block_2_1_left", nf, 2 * nf);
  basic("block_2_2_left", 2 * nf, 2 * nf);
  basic("block_3_1_left", 2 * nf, 4 * nf);
  basic("block_3_2_left", 4 * nf, 4 * nf);
  basic("block_4_1_left", 4 * nf, 8 * nf);
  basic("block_4_2_left", 8 * nf, 8 * nf);
  convt("upconv_3", 8 * nf, 4 * nf);
  basic("block_3_1_right", 8 * nf, 4 * nf);
  basic("block_3_2_right", 4 * nf, 4 * nf);
  convt("upconv_2", 4 * nf, 2 * nf);
  basic("block_2_1_right", 4 * nf, 2 * nf);
  basic("block_2_2_right", 2 * nf, 2 * nf);
  convt("upconv_1", 2 * nf, nf);
  basic("block_1_1_right", 2 * nf, nf);
  basic("block_1_2_right", nf, nf);
  head("conv1x1", nf);
  head("conv1x1_d1", 2 * nf);
  head("conv1x1_d2", 4 * nf);
  head("conv1x1_d3", 8 * nf);
}

void init_conv(hdf_plan* p, Conv3& c, const std::string& name, int cin, int cout, int lvl, bool basic) {
  c.name = name;
  c.Cin = cin;
  c.CinP = round_up(cin, 16);
  c.Cout = cout;
  c.lvl = lvl;
  if (basic) {
    c.w = p->P(name + ".conv.weight");
    c.gamma = p->P(name + ".norm.weight");
    c.beta = p->P(name + ".norm.bias");
  } else {
    c.w = p->P(name + ".double_conv.0.weight");
    c.b = p->P(name + ".double_conv.0.bias");
  }
}

void build_layers(hdf_plan* p) {
  const int nf = p->nf;
  init_conv(p, p->deep, "deep_conv", p->DM * p->M, 8 * nf, 4, false);
  init_conv(p, p->up[0], "up1", 8 * nf, 4 * nf, 3, false);
  init_conv(p, p->up[1], "up2", 4 * nf, 2 * nf, 2, false);
  init_conv(p, p->up[2], "up3", 2 * nf, nf, 1, false);
  const int ch[4] = {nf, 2 * nf, 4 * nf, 8 * nf};
  for (int k = 0; k < 4; k++) {
    std::string b = "block_" + std::to_string(k + 1);
    init_conv(p, p->enc[k][0], b + "_1_left", k == 0 ? p->M : ch[k - 1], ch[k], k, true);
    init_conv(p, p->enc[k][1], b + "_2_left", ch[k], ch[k], k, true);
    if (k < 3) {
      init_conv(p, p->dec[k][0], b + "_1_right", 2 * ch[k], ch[k], k, true);
      init_conv(p, p->dec[k][1], b + "_2_right", ch[k], ch[k], k, true);
      ConvT3& t = p->upc[k];
      t.name = "upconv_" + std::to_string(k + 1);
      t.Cin = ch[k + 1];
      t.Cout = ch[k];
      t.lvl_in = k + 1;
      t.w = p->P(t.name + ".weight");
      t.b = p->P(t.name + ".bias");
    }
  }
  const char* hn[4] = {"conv1x1", "conv1x1_d1", "conv1x1_d2", "conv1x1_d3"};
  for (int k = 0; k < 4; k++) {
    p->head[k].name = hn[k];
    p->head[k].C = ch[k];
    p->head[k].lvl = k;
    p->head[k].w = p->P(std::string(hn[k]) + ".weight");
    p->head[k].b = p->P(std::string(hn[k]) + ".bias");
  }
}

// the 2-D state_dict (conv kernels lose their depth axis) and the embedding jobs; consecutive unchanged tensors are
// merged into one copy job (both flat layouts pad every tensor to 16 floats, so such runs have equal lengths)
int build_params2d(hdf_plan* p) {
  p->params2d.clear();
  p->ejobs.clear();
  p->total_floats2d = 0;
  int64_t unet0 = p->P("block_1_1_left.conv.weight"), chain0 = p->P("deep_conv.double_conv.0.weight");
  for (const ParamInfo& pi : p->params) {
    ParamInfo q = pi;
    int kind = 0, rep = 1, inner = 1;
    if (pi.shape.size() == 5) {
      rep = (int)pi.shape[2];
      inner = (int)(pi.shape[3] * pi.shape[4]);
      q.shape.erase(q.shape.begin() + 2);
      q.numel = pi.numel / rep;
      if (rep == 3)
        kind = pi.name.rfind("upconv_", 0) == 0 ? 2 : 1;
      else if (rep == 16)
        kind = 3;
      else
        rep = 1, inner = 1;  // 1x1x1 heads: plain copy
    }
    q.offset = p->total_floats2d;
    p->total_floats2d += (q.numel + 15) / 16 * 16;
    const char stage = pi.offset >= unet0 ? 1 : pi.offset >= chain0 ? 2 : 4;
    const int64_t n3 = kind ? pi.numel : (pi.numel + 15) / 16 * 16;
    if (kind == 0 && !p->ejobs.empty()) {
      Embed2dJob& last = p->ejobs.back();
      if (last.kind == 0 && last.stage == stage && last.off3 + last.n3 == pi.offset && last.off2 + last.n3 == q.offset) {
        last.n3 += n3;
        p->params2d.push_back(q);
        continue;
      }
    }
    p->ejobs.push_back(Embed2dJob{pi.offset, q.offset, n3, inner, (short)rep, (char)kind, stage});
    p->params2d.push_back(q);
  }
  HDF_CHECK_ARG((int)p->ejobs.size() <= HDF_MAX_EMBED_JOBS, "2-D plan: %d embedding jobs (max %d)", (int)p->ejobs.size(),
                HDF_MAX_EMBED_JOBS);
  return HDF_OK;
}

int launch_embed2d(hdf_plan* p, const float* p2, float* p3, hipStream_t st) {
  Embed2dBatch b;
  for (size_t k = 0; k < p->ejobs.size(); k++) b.j[k] = p->ejobs[k];
  hipLaunchKernelGGL(embed2d_kernel, dim3(64, (unsigned)p->ejobs.size()), dim3(256), 0, st, b, p2, p3);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}
int launch_extract2d(hdf_plan* p, int stages, const float* g3, float* g2, hipStream_t st) {
  Embed2dBatch b;
  unsigned n = 0;
  for (const Embed2dJob& j : p->ejobs)
    if (j.stage & stages) b.j[n++] = j;
  if (n == 0) return HDF_OK;
  hipLaunchKernelGGL(extract2d_kernel, dim3(64, n), dim3(256), 0, st, b, g3, g2);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}
// depth slice 0 of a [rows][reps][hw] tensor of the plan's storage type <-> [rows][hw]
int launch_depth_slice(int dtype, void* t3, void* t2, int64_t rows, int reps, int64_t hw, int to2d, hipStream_t st) {
  const unsigned gx = (unsigned)std::min<int64_t>(ceil_div64(rows * (to2d ? 1 : reps) * hw, 256), 4096);
  if (dtype == HDF_F32)
    hipLaunchKernelGGL(depth_slice_kernel<f32w>, dim3(gx), dim3(256), 0, st, (f32w*)t3, (f32w*)t2, rows, reps, hw, to2d);
  else  // bf16 / f16: same 2-byte moves
    hipLaunchKernelGGL(depth_slice_kernel<bf16_t>, dim3(gx), dim3(256), 0, st, (bf16_t*)t3, (bf16_t*)t2, rows, reps, hw,
                       to2d);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

View mkview(hdf_plan* p, Bump& bp, const std::string& name, int lvl, int C, int batch) {
  View v;
  v.C = C;
  v.pitch = C;
  v.lvl = lvl;
  v.off = bp.take((size_t)batch * p->vox(lvl) * C * p->esz);
  if (!name.empty()) p->bufs[name] = v;
  return v;
}
View subview(hdf_plan* p, const View& v, int c0, int C, const std::string& name = "") {
  View s = v;
  s.off = v.off + (size_t)c0 * p->esz;
  s.C = C;
  if (!name.empty()) p->bufs[name] = s;
  return s;
}

void layout(hdf_plan* p, int B) {
  if (p->batch == B) return;
  p->batch = B;
  p->bufs.clear();
  Bump bp;
  const int nf = p->nf;
  const int ch[4] = {nf, 2 * nf, 4 * nf, 8 * nf};
  auto conv_bufs = [&](Conv3& c) {
    c.y = mkview(p, bp, "y." + c.name, c.lvl, c.Cout, B);
    size_t s = (size_t)B * c.Cout * sizeof(float);
    c.st.mean = bp.take(s);
    c.st.rstd = bp.take(s);
    c.st.scale = bp.take(s);
    c.st.shift = bp.take(s);
    c.wf = bp.take((size_t)27 * round_up(c.Cout, 32) * c.CinP * p->esz);
    c.wd = bp.take((size_t)27 * round_up(c.CinP, 32) * round_up(c.Cout, 16) * p->esz);
  };
  // ---- forward (persistent until backward)
  p->xin = mkview(p, bp, "xin", 0, 16, B);
  const int64_t rows = (int64_t)p->M * B * p->Ntok;
  p->tf_F = bp.take((size_t)p->nb * rows * p->DMF * sizeof(float));
  p->tf_save = bp.take((size_t)p->nb * 4 * rows * 232 * sizeof(float));
  {
    TfDims dd{};
    dd.M = p->M;
    p->tf_wpack = bp.take(tf_chain_wpack_bytes(dd, p->nb));   // (forward region: the backward reads what the forward packed)
    dd.B = B, dd.N = p->Ntok;
    p->tf_frag = bp.take(p->dtype == HDF_F32 ? 256 : tf_chain_frag_bytes(dd, p->nb));
  }
  p->tf_sync = bp.take((size_t)3 << 20);  // forward | backward counters in the first megabyte (one half each), then
                                           // two megabytes of phase stamps in -DCHAIN_DBG_STAMPS builds
  p->attnall = mkview(p, bp, "attnall", 4, p->M * p->DM, B);
  conv_bufs(p->deep);
  p->attnout = mkview(p, bp, "attnout", 3, 8 * nf, B);
  for (int k = 0; k < 3; k++) conv_bufs(p->up[k]);
  p->at[2] = mkview(p, bp, "at1", 2, 4 * nf, B);
  p->at[1] = mkview(p, bp, "at2", 1, 2 * nf, B);
#ifdef HDF_NO_FUSED_AT3  // (A/B builds; the product evaluates at3 inside the level-0 encoder tail: forward3d)
  p->at[0] = mkview(p, bp, "at3", 0, nf, B);
#else
  if (p->flat) p->at[0] = mkview(p, bp, "at3", 0, nf, B);   // (the 2-D encoder tail reads a materialised at3)
#endif
  for (int k = 0; k < 4; k++) {
    conv_bufs(p->enc[k][0]);
    conv_bufs(p->enc[k][1]);
    if (k < 3) {
      p->cat[k] = mkview(p, bp, "cat" + std::to_string(k + 1), k, 2 * ch[k], B);
      subview(p, p->cat[k], ch[k], ch[k], "ds" + std::to_string(k));
      p->pooled[k] = mkview(p, bp, "pool" + std::to_string(k + 1), k + 1, ch[k], B);
      p->pool_idx[k] = bp.take((size_t)B * p->vox(k + 1) * ch[k]);
      conv_bufs(p->dec[k][0]);
      conv_bufs(p->dec[k][1]);
      ConvT3& t = p->upc[k];
      t.wf = bp.take((size_t)27 * round_up(t.Cout, 32) * t.Cin * p->esz);
      t.wd = bp.take((size_t)27 * round_up(t.Cin, 32) * t.Cout * p->esz);
    }
  }
  p->x4 = mkview(p, bp, "bottleneck", 3, 8 * nf, B);
  if (p->is2d) {  // depth-replicated input, embedded parameters / their gradients, 3-D logits and logit gradients
    p->e_x3d = bp.take((size_t)B * p->M * p->D * p->H * p->W * sizeof(float));   // depth-16 copy of the input
    p->e_params3d = bp.take((size_t)p->total_floats * sizeof(float));
    p->e_grads3d = bp.take((size_t)p->total_floats * sizeof(float));
    for (int i = 0; i < 4; i++) {
      p->e_out3d[i] = bp.take((size_t)B * p->ncls * p->vox(i) * p->esz);
      p->e_dout3d[i] = bp.take((size_t)B * p->ncls * p->vox(i) * p->esz);
    }
  }
  // ---- weight packs (see conv_forward / conv_backward / convt_* for the layouts)
  p->pack_jobs.clear();
  auto conv_jobs = [&](Conv3& c) {
    const int* d = p->dims[c.lvl];
    c.wf_frag = hdf_conv_weight_layout(p->dtype, 0, c.CinP, d[0], d[1], d[2]);
    c.wd_frag = hdf_conv_weight_layout(p->dtype, 0, c.Cout, d[0], d[1], d[2]);
    // forward [tap][CoutP][CinP] from torch [Cout][Cin][27]
    p->pack_jobs.push_back(
        PackJob{c.w, (int64_t)c.wf, c.Cout, c.Cin, round_up(c.Cout, 32), c.CinP, c.Cin * 27, 27, 0, c.wf_frag});
    // dgrad: taps reversed, channel roles swapped: Wd[t][ci][co] = W[co][ci][26-t]
    p->pack_jobs.push_back(
        PackJob{c.w, (int64_t)c.wd, c.Cin, c.Cout, round_up(c.Cin, 32), c.Cout, 27, c.Cin * 27, 1, c.wd_frag});
  };
  conv_jobs(p->deep);
  for (int k = 0; k < 3; k++) conv_jobs(p->up[k]);
  for (int k = 0; k < 4; k++) {
    conv_jobs(p->enc[k][0]);
    conv_jobs(p->enc[k][1]);
    if (k < 3) {
      conv_jobs(p->dec[k][0]);
      conv_jobs(p->dec[k][1]);
      ConvT3& t = p->upc[k];
      const int* d = p->dims[t.lvl_in];
      t.wf_frag = hdf_conv_weight_layout(p->dtype, 2, t.Cin, d[0], d[1], d[2]);
      t.wd_frag = hdf_conv_weight_layout(p->dtype, 1, t.Cout, d[0], d[1], d[2]);
      // forward: torch ConvTranspose3d weight [Cin][Cout][27] -> [tap][CoutP][Cin]
      p->pack_jobs.push_back(
          PackJob{t.w, (int64_t)t.wf, t.Cout, t.Cin, round_up(t.Cout, 32), t.Cin, 27, t.Cout * 27, 0, t.wf_frag});
      // input gradient: stride-2 gather conv, [tap][CinP][Cout]
      p->pack_jobs.push_back(
          PackJob{t.w, (int64_t)t.wd, t.Cin, t.Cout, round_up(t.Cin, 32), t.Cout, t.Cout * 27, 27, 0, t.wd_frag});
    }
  }
  // ---- scratch shared by forward and backward
  size_t maxtiles = 0;
  for (int l = 0; l < 5; l++)
    for (int rb : {32, 1 << 20})  // weights-stationary (per-workgroup rows) and tiled (per-tile rows) geometry
      maxtiles = std::max<size_t>(maxtiles, hdf_conv_stat_tiles(0, p->dims[l][0], p->dims[l][1], p->dims[l][2], rb));
  p->stat_partials = bp.take((size_t)B * maxtiles * round_up(8 * nf, 32) * 2 * sizeof(float));
  {  // branch stream: deep_conv (level 4) and up1..3 (levels 3, 2, 1) write their InstanceNorm partials here
    size_t mt = 0;
    for (int l = 1; l < 5; l++)
      for (int rb : {32, 1 << 20}) mt = std::max<size_t>(mt, hdf_conv_stat_tiles(0, p->dims[l][0], p->dims[l][1], p->dims[l][2], rb));
    p->stat_partials2 = bp.take((size_t)B * mt * round_up(8 * nf, 32) * 2 * sizeof(float));
  }
  p->ksplit_ws = bp.take(HDF_KSPLIT_BYTES);
  p->ksplit_ws2 = bp.take(HDF_KSPLIT_BYTES);
  // Everything above is what a forward touches: an inference-only caller (eval / sliding-window prediction) can hand
  // over just this prefix (hdf_plan_inference_workspace_bytes); the backward scratch below -- transformer tapes, second
  // dy buffers, the 128 MB weight-gradient workspace, ... -- is more than half of the arena at the benchmark size.
  p->ws_fwd_bytes = bp.cur;
  // ---- backward scratch
  p->tf_scratch = bp.take((size_t)rows * std::max(160, p->DM) * sizeof(float));
  p->tf_dF = bp.take((size_t)rows * p->DMF * sizeof(float));
  // weight-gradient operand tapes of the transformer branches (contracted by tf_wgrad at the end of backward)
  p->tf_tape = bp.take((size_t)p->nb * 4 * rows * TF_TAPE_W * sizeof(float));
  p->tf_otape = bp.take((size_t)p->nb * rows * p->DMF * sizeof(float));
  p->wgrad_ws_bytes = (size_t)128 << 20;
  p->wgrad_ws = bp.take(p->wgrad_ws_bytes);
  p->inb_partials = bp.take((size_t)B * 1024 * 8 * nf * 2 * sizeof(float));  // hdf_in_bwd_blocks <= 1024, C <= 8 nf
  p->inb_k = bp.take((size_t)3 * B * 8 * nf * sizeof(float));
  p->inb_partials2 = bp.take((size_t)B * 1024 * 8 * nf * 2 * sizeof(float));
  p->inb_k2 = bp.take((size_t)3 * B * 8 * nf * sizeof(float));
  p->inb_k3 = bp.take((size_t)3 * B * 8 * nf * sizeof(float));
  for (int k = 0; k < 4; k++) {
    // (named for tools/cos_probe.py: after a backward g.y_<k> holds the raw-output gradient of the encoder's SECOND conv of
    // level k and g.a_<k> the gradient of its input activation -- the last writers of the two buffers)
    p->gA[k] = mkview(p, bp, "g.a_" + std::to_string(k), k, ch[k], B);
    p->gY[k] = mkview(p, bp, "g.y_" + std::to_string(k), k, ch[k], B);
    p->gY2[k] = mkview(p, bp, "g.y2_" + std::to_string(k), k, ch[k], B);  // the level's second conv keeps its own dy (read by a side-stream wgrad)
    if (k < 3) {
      // gradient of cat_k = [upconv | ds]: two dense buffers when the halves are whole 32-channel blocks (every
      // consumer of a half -- InstanceNorm backward, max-pool backward, up-sampling backward, the transposed conv's
      // backward -- then streams whole lines instead of 64 of every 128 bytes), else one buffer with views
      p->dcat_split[k] = ch[k] % 32 == 0;
      if (p->dcat_split[k]) {
        p->dUp[k] = mkview(p, bp, "g.up" + std::to_string(k + 1), k, ch[k], B);
        p->dSkip[k] = mkview(p, bp, "g.ds" + std::to_string(k), k, ch[k], B);
        p->dCat[k] = p->dUp[k];
      } else {
        p->dCat[k] = mkview(p, bp, "g.cat" + std::to_string(k + 1), k, 2 * ch[k], B);
        p->dUp[k] = subview(p, p->dCat[k], 0, ch[k]);
        p->dSkip[k] = subview(p, p->dCat[k], ch[k], ch[k]);
      }
      p->dP[k] = mkview(p, bp, "g.pool" + std::to_string(k + 1), k + 1, ch[k], B);
    }
  }
  // UpConv chain: conv outputs live at levels 4,3,2,1 with channels 8nf,4nf,2nf,nf
  const int uc[4] = {8 * nf, 4 * nf, 2 * nf, nf};
  for (int k = 0; k < 4; k++) {
    p->dUa[k] = mkview(p, bp, "", 4 - k, uc[k], B);
    p->dUy[k] = mkview(p, bp, "", 4 - k, uc[k], B);
  }
  p->dX4 = mkview(p, bp, "g.attnout", 3, 8 * nf, B);
  p->dAttnall = mkview(p, bp, "g.attnall", 4, p->M * p->DM, B);
  p->ws_bytes = bp.cur;
}

// Backward runs its weight gradients on the plan's side stream.  They are off the critical path (nothing in backward
// reads a weight gradient), MFMA-bound, and leave wave slots and 50 KB of LDS per CU free, while the chain they would
// otherwise delay is full of HBM-bound passes (InstanceNorm backward, pooling / up-sampling backward, heads): with both in
// flight the memory-bound kernels run under the matrix kernels (tools/overlap_probe.py: a 64->32 weight gradient plus
// three elementwise passes over 268 MB tensors take 708 us on two streams against 880 us back to back).  Ordering:
//  * fork: the side stream waits for an event recorded on the main stream after the producers of the operands;
//  * a buffer a side-stream kernel still reads (the dy of a conv) is not overwritten: wait_readers() before its next
//    writer on the main stream (each level keeps two dy buffers so that the wait is normally already satisfied);
//  * join: the main stream waits for the side stream's last event at the end of every backward call, so at the ABI
//    boundary all work is ordered on the caller's stream as before.
// The shared weight-gradient workspace is only touched on the side stream (its kernels run in order).
struct Exec {
  hdf_plan* p;
  char* ws;
  const float* params;
  float* grads;
  int B;
  hipStream_t st;
  int conv_budget = 0;                          // ConvArgs::cu_budget of the convolutions issued through this Exec (0: all)
  bool async = false;                           // weight gradients on the side stream
  bool on_branch = false;                       // this Exec issues onto the plan's branch stream (own scratch)
  hipEvent_t last_side = nullptr;               // last event recorded on the side stream in this call
  hipEvent_t tf_packed = nullptr;               // branch Exec: the persistent transformer kernel's weight copies are ready (forward3d)
  std::map<size_t, hipEvent_t> readers;         // workspace offset of a buffer -> side-stream event after its last reader
  hipEvent_t next_event() {
    if (p->events.size() < 256) {
      hipEvent_t ev = nullptr;
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr;
      p->events.push_back(ev);
      return ev;
    }
    return p->events[p->ev_next++ % p->events.size()];
  }
  // stream for a weight-gradient launch whose operands are ready on the main stream now
  hipStream_t wgrad_stream() {
    if (!async) return st;
    hipEvent_t f = next_event();
    if (!f || hipEventRecord(f, st) != hipSuccess || hipStreamWaitEvent(p->side, f, 0) != hipSuccess) {
      // fall back to in-order execution: first order this stream behind EVERYTHING the side stream holds -- the main and
      // the branch Exec both feed it and share the one weight-gradient workspace this stream is about to reuse, so this
      // Exec's own last_side is not enough.  A fresh event on the side stream, or, if events are what fails, a host wait.
      hipEvent_t all = next_event();
      if (!all || hipEventRecord(all, p->side) != hipSuccess || hipStreamWaitEvent(st, all, 0) != hipSuccess)
        (void)hipStreamSynchronize(p->side);
      last_side = nullptr;
      readers.clear();
      async = false;
      return st;
    }
    return p->side;
  }
  // after the launch: remember that `buf` is read on the side stream until now
  // (an event that cannot be recorded would leave the launch outside every later join: a hard error, not a silent
  // loss of ordering)
  int wgrad_done(const View& buf) {
    if (!async) return HDF_OK;
    hipEvent_t d = next_event();
    if (!d || hipEventRecord(d, p->side) != hipSuccess) {
      (void)hipStreamSynchronize(p->side);
      hdf_set_error("backward: could not record the side stream's completion event");
      return HDF_ERR_HIP;
    }
    last_side = d;
    readers[buf.off] = d;
    return HDF_OK;
  }
  // the same for a side-stream launch whose operands are never overwritten inside this call: only join() waits for it
  int side_done() {
    if (!async) return HDF_OK;
    hipEvent_t d = next_event();
    if (!d || hipEventRecord(d, p->side) != hipSuccess) {
      (void)hipStreamSynchronize(p->side);
      hdf_set_error("backward: could not record the side stream's completion event");
      return HDF_ERR_HIP;
    }
    last_side = d;
    return HDF_OK;
  }
  void wait_readers(const View& buf) {
    auto it = readers.find(buf.off);
    if (it != readers.end()) {
      (void)hipStreamWaitEvent(st, it->second, 0);
      readers.erase(it);
    }
  }
  void join() {
    if (last_side) (void)hipStreamWaitEvent(st, last_side, 0);
    last_side = nullptr;
    readers.clear();
  }
  // scratch of this Exec's stream (two streams of one call must not share the per-launch partial-sum tables)
  float* statp() const { return f(on_branch ? p->stat_partials2 : p->stat_partials); }
  float* kspl() const { return f(on_branch ? p->ksplit_ws2 : p->ksplit_ws); }
  float* inbp() const { return f(on_branch ? p->inb_partials2 : p->inb_partials); }
  float* inbk() const { return f(on_branch ? p->inb_k2 : p->inb_k); }
  // fork: a second Exec on the plan's branch stream, ordered behind everything issued on this one so far.  nullptr
  // stream when the branch stream cannot be used (creation / event failure): the caller then stays in order.
  hipStream_t fork_branch() {
    if (!p->branch) {
      int least = 0, greatest = 0;  // the branch carries the longer dependency chain: highest priority
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
          hipStreamCreateWithPriority(&p->branch, hipStreamNonBlocking, greatest) != hipSuccess) {
        if (hipStreamCreateWithFlags(&p->branch, hipStreamNonBlocking) != hipSuccess) p->branch = nullptr;
      }
    }
    if (!p->branch) return nullptr;
    hipEvent_t f = next_event();
    if (!f || hipEventRecord(f, st) != hipSuccess || hipStreamWaitEvent(p->branch, f, 0) != hipSuccess) return nullptr;
    return p->branch;
  }
  // join a branch Exec back: this stream waits for everything issued on the branch (incl. its side-stream work)
  int join_branch(Exec& b) {
    b.join();
    hipEvent_t d = next_event();
    if (!d || hipEventRecord(d, b.st) != hipSuccess || hipStreamWaitEvent(st, d, 0) != hipSuccess) {
      hdf_set_error("branch stream: join failed");
      return HDF_ERR_HIP;
    }
    return HDF_OK;
  }
  void* at(const View& v) const { return ws + v.off; }
  float* f(size_t off) const { return reinterpret_cast<float*>(ws + off); }
  const float* P(int64_t off) const { return off < 0 ? nullptr : params + off; }
  float* G(int64_t off) const { return off < 0 ? nullptr : grads + off; }
  const int* dm(int lvl) const { return p->dims[lvl]; }
};

// per-(n,c) input transform of a consumer: the producer's InstanceNorm scale/shift (+ReLU)
struct Xf {
  const float* scale = nullptr;
  const float* shift = nullptr;
  int relu = 0;
};
Xf xf_of(const Exec& e, const Conv3& c) { return Xf{e.f(c.st.scale), e.f(c.st.shift), 1}; }

// probe: record the plan's probe events immediately around the convolution launch (hdf_plan_set_probe)
int conv_forward(Exec& e, Conv3& c, const View& in, Xf xf, bool probe = false) {
  hdf_plan* p = e.p;
  const int* d = e.dm(c.lvl);
  const int CoutP = round_up(c.Cout, 32);
  ConvArgs a{};  // weights: packed by hdf_forward's pack batch
  a.prio = e.on_branch;  // the UpConv chain's convolutions run next to the encoder's persistent ones (ConvArgs::prio)
  a.cu_budget = e.conv_budget;
  a.in = e.at(in);
  a.in_pitch = in.pitch;
  a.Cin = c.CinP;
  a.N = e.B;
  a.Di = a.Do = d[0];
  a.Hi = a.Ho = d[1];
  a.Wi = a.Wo = d[2];
  a.w = e.ws + c.wf;
  a.wfrag = c.wf_frag;
  a.bias = e.P(c.b);
  a.in_scale = xf.scale;
  a.in_shift = xf.shift;
  a.in_relu = xf.relu;
  a.out = e.at(c.y);
  a.out_pitch = c.y.pitch;
  a.Cout = c.Cout;
  a.CoutP = CoutP;
  a.stat_partials = e.statp();
  a.accumulate = 0;
  a.kpart = e.kspl(), a.kpart_bytes = HDF_KSPLIT_BYTES;
#ifndef HDF_NO_CONV_FIRST  // (A/B builds)
  // the encoder's first layer (<= 4 real channels in a 16-channel row): K = (tap, channel), csrc/conv_first.hip
  if (c.Cin <= 4 && !xf.scale && hdf_conv_first_takes(p->dtype, c.Cin, c.Cout, d[0], d[1], d[2], in.pitch)) {
    HDF_TRY(hdf_launch_conv_first(p->dtype, e.at(in), in.pitch, c.Cin, e.B, d[0], d[1], d[2], e.P(c.w), e.P(c.b),
                                  e.at(c.y), c.y.pitch, c.Cout, e.statp(), e.st));
  } else
#endif
  {
    const bool pr = probe && p->probe_start && p->probe_stop;
    if (pr && hipEventRecord(p->probe_start, e.st) != hipSuccess) {
      hdf_set_error("probe: hipEventRecord failed");
      return HDF_ERR_HIP;
    }
    HDF_TRY(hdf_launch_conv(p->dtype, 0, a, e.st));
    if (pr && hipEventRecord(p->probe_stop, e.st) != hipSuccess) {
      hdf_set_error("probe: hipEventRecord failed");
      return HDF_ERR_HIP;
    }
  }
  int tiles = hdf_conv_stat_tiles(0, d[0], d[1], d[2], c.CinP * p->esz);
  HDF_TRY(hdf_launch_in_finalize(e.statp(), e.B, tiles, c.Cout, CoutP, p->vox(c.lvl), e.P(c.gamma),
                                 e.P(c.beta), 1e-5f, e.f(c.st.mean), e.f(c.st.rstd), e.f(c.st.scale),
                                 e.f(c.st.shift), e.st));
  return HDF_OK;
}

int convt_forward(Exec& e, ConvT3& t, const View& in, Xf xf, const View& out) {
  hdf_plan* p = e.p;
  const int* d = e.dm(t.lvl_in);
  const int CoutP = round_up(t.Cout, 32);
  ConvArgs a{};
  a.in = e.at(in);
  a.in_pitch = in.pitch;
  a.Cin = t.Cin;
  a.N = e.B;
  a.Di = d[0], a.Hi = d[1], a.Wi = d[2];
  a.Do = (p->flat ? 1 : 2) * d[0], a.Ho = 2 * d[1], a.Wo = 2 * d[2];
  a.w = e.ws + t.wf;
  a.wfrag = t.wf_frag;
  a.bias = e.P(t.b);
  a.in_scale = xf.scale;
  a.in_shift = xf.shift;
  a.in_relu = xf.relu;
  a.out = e.at(out);
  a.out_pitch = out.pitch;
  a.Cout = t.Cout;
  a.CoutP = CoutP;
  return hdf_launch_conv(p->dtype, 2, a, e.st);
}

int head_forward(Exec& e, const Head1& h, const View& in, Xf xf, void* out) {
  return hdf_launch_head_fwd(e.p->dtype, e.at(in), in.pitch, xf.scale, xf.shift, e.P(h.w), e.P(h.b), out, e.B, h.C,
                             e.p->ncls, e.p->vox(h.lvl), e.st);
}

TfDims tf_dims(const hdf_plan* p, int B) {
  TfDims d;
  d.M = p->M;
  d.B = B;
  d.N = p->Ntok;
  d.DM = p->DM;
  d.DMF = p->DMF;
  d.mstride = p->mstride;
  d.training = p->training;
  d.seed = p->seed;
  d.thresh24 = 1u << 23;  // p = 0.5 (HDenseFormer.py:79,105)
  d.keep_scale = 2.0f;
  return d;
}

void tf_layer_ptrs(const hdf_plan* p, float* base, int b, int l, TfLayerP& q) {
  std::string pre = "attns.0.blocks." + std::to_string(b) + ".0.layers." + std::to_string(l);
  q.w0 = base + p->P(pre + ".0.weight");
  q.b0 = base + p->P(pre + ".0.bias");
  q.ln1g = base + p->P(pre + ".1.norm.weight");
  q.ln1b = base + p->P(pre + ".1.norm.bias");
  q.wqkv = base + p->P(pre + ".1.fn.to_qkv.weight");
  q.wout = base + p->P(pre + ".1.fn.to_out.0.weight");
  q.bout = base + p->P(pre + ".1.fn.to_out.0.bias");
  q.ln2g = base + p->P(pre + ".2.norm.weight");
  q.ln2b = base + p->P(pre + ".2.norm.bias");
  q.w1 = base + p->P(pre + ".2.fn.net.0.weight");
  q.b1 = base + p->P(pre + ".2.fn.net.0.bias");
  q.w2 = base + p->P(pre + ".2.fn.net.3.weight");
  q.b2 = base + p->P(pre + ".2.fn.net.3.bias");
}
void tf_out_ptrs(const hdf_plan* p, float* base, int b, TfOutP& q) {
  std::string pre = "attns.0.blocks." + std::to_string(b) + ".0.out_layer.net";
  q.wa = base + p->P(pre + ".0.weight");
  q.ba = base + p->P(pre + ".0.bias");
  q.wb = base + p->P(pre + ".3.weight");
  q.bb = base + p->P(pre + ".3.bias");
}
TfLayerSave tf_save(const hdf_plan* p, const Exec& e, int b, int l) {
  const int64_t rows = (int64_t)p->M * e.B * p->Ntok;
  float* base = e.f(p->tf_save) + (int64_t)(b * 4 + l) * rows * 232;
  TfLayerSave s;
  s.h0 = base;
  s.qkv = base + rows * 32;
  s.ob = base + rows * 128;
  s.lse = base + rows * 160;
  s.h1 = base + rows * 168;
  s.h2 = base + rows * 200;
  return s;
}

// Parameter addressing of the persistent transformer kernels (TfChainP, transformer.h): block-relative offsets of the 13
// tensors of each of a block's four dense layers and of its out_layer, the same for every block and modality.
TfChainP tf_chain_params(const hdf_plan* p) {
  TfChainP c{};
  c.blk0 = p->P("attns.0.blocks.0.0.layers.0.0.weight");
  c.blk_stride = p->nb > 1 ? p->P("attns.0.blocks.1.0.layers.0.0.weight") - c.blk0 : 0;
  TfLayerP q;
  TfOutP o;
  float* base = nullptr;
  for (int l = 0; l < 4; l++) {
    tf_layer_ptrs(p, base, 0, l, q);
    float* const f[13] = {q.w0, q.b0, q.ln1g, q.ln1b, q.wqkv, q.wout, q.bout, q.ln2g, q.ln2b, q.w1, q.b1, q.w2, q.b2};
    for (int k = 0; k < 13; k++) c.loff[l][k] = (int32_t)((f[k] - base) - c.blk0);
  }
  tf_out_ptrs(p, base, 0, o);
  float* const g[4] = {o.wa, o.ba, o.wb, o.bb};
  for (int k = 0; k < 4; k++) c.ooff[k] = (int32_t)((g[k] - base) - c.blk0);
  return c;
}
// The persistent kernels take the plan's transformer when every 16-token tile of every sequence gets a compute unit of
// its own (resident together: their per-sequence barriers need that).  HDF_NO_TF_CHAIN=1: the launch chain
// (tok_fwd / attention / tok_bwd ...) instead -- the third arrangement knob of tests/test_gpu_knobs.py.
// Decided ONCE per forward (forward3d stores the answer in p->tf_fwd_chain); the backward follows the forward it belongs
// to instead of reading the environment again (a knob flipped between the two calls made the persistent
// backward consume records the launch-chain forward never wrote).
bool tf_use_chain(const hdf_plan* p, int B) {
  const bool off = getenv("HDF_NO_TF_CHAIN") != nullptr;   // read per FORWARD call: tests switch it inside one process
  if (off || p->chain_off) return false;
  TfDims d = tf_dims(p, B);
  return p->chain_forced ? tf_chain_shape_ok(d) : tf_chain_supported(d);
}
// the plan's host-mapped give-up word, created on first use
int chain_flag_ensure(hdf_plan* p) {
  if (p->chain_flag) return HDF_OK;
  void* h = nullptr;
  if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) {
    hdf_set_error("transformer chain: could not allocate the host-mapped status word");
    return HDF_ERR_HIP;
  }
  void* dv = nullptr;
  if (hipHostGetDevicePointer(&dv, h, 0) != hipSuccess) {
    (void)hipHostFree(h);
    hdf_set_error("transformer chain: no device address for the host-mapped status word");
    return HDF_ERR_HIP;
  }
  memset(h, 0, 64);
  p->chain_flag = reinterpret_cast<unsigned*>(h);
  p->chain_flag_dev = reinterpret_cast<unsigned*>(dv);
  return HDF_OK;
}
// Called at the top of every forward / backward: a persistent launch of an EARLIER call gave up at a barrier (the device
// was shared: its grid was not resident together within the deadline).  That call's outputs are NaN-poisoned garbage; this
// call reports it once -- HDF_ERR_CHAIN_TIMEOUT, nothing launched -- and the plan runs the launch chain from now on.
// (Asynchronous by nature: the host is ahead of the device, so the report can be one or more calls late; a caller that
// synchronises can ask at once with hdf_plan_chain_state.)
int chain_flag_check(hdf_plan* p) {
  if (!p->chain_flag) return HDF_OK;
  const unsigned v = __atomic_load_n(p->chain_flag, __ATOMIC_ACQUIRE);
  if (v == 0) return HDF_OK;
  __atomic_store_n(p->chain_flag, 0u, __ATOMIC_RELEASE);
  p->chain_last_giveup = v;
  p->chain_off = true;
  hdf_set_error("persistent transformer kernel: workgroup %u gave up at a per-sequence barrier (the compute units were not "
                "all available to the launch); the outputs of that call are NaN; this plan uses the launch chain from now on",
                v - 1);
  return HDF_ERR_CHAIN_TIMEOUT;
}
TfChainCtl chain_ctl(const hdf_plan* p) {
  TfChainCtl c;
  c.host_flag = p->chain_flag_dev;
  c.ticks = p->chain_ticks;
  return c;
}

int transformer_forward(Exec& e, const float* x) {
  hdf_plan* p = e.p;
  TfDims d = tf_dims(p, e.B);
  float* pm = const_cast<float*>(e.params);
  const int64_t rows = (int64_t)p->M * e.B * p->Ntok;
  float* F0 = e.f(p->tf_F);
#ifdef HDF_PE_FP32  // (A/B builds: the exact fp32 patch embedding in every storage mode)
  const int PE_LP = 0;
#else
  const int PE_LP = p->dtype;
#endif
  // (flat: the 2-D input itself, depth 1, against depth slice 0 of the embedded 16^3 patch kernel)
  HDF_TRY(tf_patch_embed_fwd(d, x, p->flat ? 1 : p->D, p->H, p->W, pm + p->P("attns.0.patch_embeddings.weight"),
                             pm + p->P("attns.0.patch_embeddings.bias"), pm + p->P("attns.0.position_embeddings"), F0,
                             e.st, PE_LP, p->flat ? 1 : 16));
  if (p->tf_fwd_chain) {  // all layers of all blocks in one persistent launch (transformer_chain.hip)
    if (e.tf_packed && hipStreamWaitEvent(e.st, e.tf_packed, 0) != hipSuccess) {
      hdf_set_error("branch stream: wait failed");
      return HDF_ERR_HIP;
    }
    return tf_chain_forward(d, tf_chain_params(p), p->nb, pm, F0, e.f(p->tf_save), e.at(p->attnall),
                            reinterpret_cast<unsigned*>(e.ws + p->tf_sync), e.ws + p->tf_wpack, e.f(p->tf_frag), p->dtype, e.st,
                            chain_ctl(p));
  }
  // token kernel, attention, token kernel, ...: between two attention launches ONE kernel finishes the previous
  // dense layer (and, at a block boundary, runs the block's out_layer) and starts the next one
  TfLayerP prev{}, cur{};
  TfOutP o{};
  for (int b = 0; b < p->nb; b++) {
    float* F = F0 + (int64_t)b * rows * p->DMF;
    for (int l = 0; l < 4; l++) {
      tf_layer_ptrs(p, pm, b, l, cur);
      TfTokenFwd t;
      if (l > 0) {
        t.post = &prev, t.post_save = tf_save(p, e, b, l - 1), t.bp = b, t.lp = l - 1, t.F_post = F;
      } else if (b > 0) {
        float* Fp = F0 + (int64_t)(b - 1) * rows * p->DMF;
        t.post = &prev, t.post_save = tf_save(p, e, b - 1, 3), t.bp = b - 1, t.lp = 3, t.F_post = Fp;
        tf_out_ptrs(p, pm, b - 1, o);
        t.out = &o, t.next_F = F;
      }
      t.pre = &cur, t.pre_save = tf_save(p, e, b, l), t.bq = b, t.lq = l, t.F_pre = F;
      HDF_TRY(tf_token_fwd(d, t, p->dtype, e.st));
      TfLayerSave s = tf_save(p, e, b, l);
      HDF_TRY(tf_attention_fwd(d.N, d.M * d.B, s.qkv, s.ob, s.lse, e.st));
      prev = cur;
    }
  }
  TfTokenFwd t;
  const int b = p->nb - 1;
  t.post = &prev, t.post_save = tf_save(p, e, b, 3), t.bp = b, t.lp = 3, t.F_post = F0 + (int64_t)b * rows * p->DMF;
  tf_out_ptrs(p, pm, b, o);
  t.out = &o, t.attnall = e.at(p->attnall);
  return tf_token_fwd(d, t, p->dtype, e.st);
}

int transformer_backward(Exec& e, const float* x) {
  hdf_plan* p = e.p;
  TfDims d = tf_dims(p, e.B);
  float* pm = const_cast<float*>(e.params);
  const int64_t rows = (int64_t)p->M * e.B * p->Ntok;
  float* F0 = e.f(p->tf_F);
  float* dF = e.f(p->tf_dF);
  float* scratch = e.f(p->tf_scratch);
  // token kernel, attention backward, token kernel, ...: one launch runs the Linear0 / LN1 / to_qkv backward of the
  // layer whose attention backward just finished, (at a block boundary) the previous block's out_layer backward, and
  // the ff / to_out backward of the next layer down
  float* dO = scratch;
  float* dh0acc = scratch + rows * 32;
  float* dqkv = scratch + rows * 64;
  // Every weight-matrix gradient of the branches comes from the tapes the token kernels leave behind (fixed-order
  // reductions, no atomics): one launch per block, issued on the side stream as soon as that block's last tape segment
  // is written (the token kernel that also starts the next block down), so that only block 0's -- next to the patch
  // embedding's -- is left at the end of the chain.  (One launch for all blocks after the chain: 114 us with nothing
  // else left to run beside it.)
  TfWgradArgs w{};
  {
    const int64_t blk0 = p->P("attns.0.blocks.0.0.layers.0.0.weight");
    const int64_t blk_stride = p->nb > 1 ? p->P("attns.0.blocks.1.0.layers.0.0.weight") - blk0 : 0;
    int k = 0;
    auto rel = [&](const std::string& n) { return p->P("attns.0.blocks.0.0." + n) - blk0; };
    for (int l = 0; l < 4; l++) {
      const std::string pre = "layers." + std::to_string(l);
      w.e[k++] = TfWgradEntry{rel(pre + ".1.fn.to_qkv.weight"), 96, 32, l, 0, 0, TF_T_DQ, TF_T_T, -1, -1, 96, 32};
      w.e[k++] = TfWgradEntry{rel(pre + ".0.weight"), 32, p->DM + 32 * l, l, 0, 1, TF_T_DH0, 0, -1, -1, 32, p->DMF};
      w.e[k++] = TfWgradEntry{rel(pre + ".2.fn.net.3.weight"), 32, 64, l, 0, 0, TF_T_P1, TF_T_P1 + 32, TF_T_P0, TF_T_P0 + 32,
                              32, 64};
      w.e[k++] = TfWgradEntry{rel(pre + ".2.fn.net.0.weight"), 64, 32, l, 0, 0, TF_T_P1 + 96, TF_T_P1 + 160, TF_T_P0 + 96,
                              TF_T_P0 + 160, 64, 32};
      w.e[k++] = TfWgradEntry{rel(pre + ".1.fn.to_out.0.weight"), 32, 32, l, 0, 2, TF_T_DGO, 0, -1, -1, 32, 32};
    }
    w.e[k++] = TfWgradEntry{rel("out_layer.net.3.weight"), p->DM, 64, 0, 3, 3, 0, p->DM, -1, -1, p->DM, 64};
    w.e[k++] = TfWgradEntry{rel("out_layer.net.0.weight"), 64, p->DMF, 0, 3, 1, p->DM + 64, 0, -1, -1, 64, p->DMF};
    w.grads = e.grads, w.mstride = p->mstride, w.block0 = blk0, w.block_stride = blk_stride;
    w.tape = e.f(p->tf_tape), w.otape = e.f(p->tf_otape), w.F = F0, w.save = e.f(p->tf_save);
    w.rows = rows, w.BN = e.B * p->Ntok, w.DMF = p->DMF, w.b0 = 0;
  }
  auto wgrad_block = [&](int b) -> int {
    w.b0 = b;
    HDF_TRY(tf_wgrad(w, 1, p->M, e.wgrad_stream()));
    return e.side_done();
  };
  p->tf_bwd_chain = p->tf_fwd_chain && tf_chain_backward_supported(d, p->dtype);
  if (p->tf_bwd_chain) {
    // one persistent launch for all layers (transformer_chain.hip); then every block's weight-matrix gradients from the
    // tapes on the side stream, next to the patch embedding's backward on this one
    HDF_TRY(tf_chain_backward(d, tf_chain_params(p), p->nb, pm, e.grads, F0, e.f(p->tf_save), dF, e.at(p->dAttnall),
                              e.f(p->tf_tape), e.f(p->tf_otape), scratch, e.f(p->tf_frag), e.ws + p->tf_wpack,
                              reinterpret_cast<unsigned*>(e.ws + p->tf_sync) + (1 << 17), p->dtype, e.st, chain_ctl(p)));
    // (on this stream, not on the side stream: that one still holds the level-0 weight gradients, and tf_wgrad -- HBM-bound,
    // 110 us -- would run behind them as the last kernel of the step)
    HDF_TRY(tf_patch_embed_bwd(d, x, p->flat ? 1 : p->D, p->H, p->W, dF, e.grads + p->P("attns.0.patch_embeddings.weight"),
                               e.grads + p->P("attns.0.patch_embeddings.bias"),
                               e.grads + p->P("attns.0.position_embeddings"), scratch, e.st, p->flat ? 1 : 16));
    w.b0 = 0;
    HDF_TRY(tf_wgrad(w, p->nb, p->M, e.st));
    return HDF_OK;
  }
  TfLayerP up{}, gup{}, cur{}, gcur{};
  TfOutP o{}, go{};
  bool have_up = false;
  int ub = 0, ul = 0;
  for (int b = p->nb - 1; b >= 0; b--) {
    float* F = F0 + (int64_t)b * rows * p->DMF;
    for (int l = 3; l >= 0; l--) {
      tf_layer_ptrs(p, pm, b, l, cur);
      tf_layer_ptrs(p, e.grads, b, l, gcur);
      TfTokenBwd t;
      t.dF = dF;
      float* tape = e.f(p->tf_tape);
      float* otape = e.f(p->tf_otape);
      if (have_up) {
        t.pre = &up, t.pre_grad = &gup, t.pre_save = tf_save(p, e, ub, ul), t.bq = ub, t.lq = ul;
        t.F_pre = F0 + (int64_t)ub * rows * p->DMF, t.dqkv = dqkv, t.dh0acc = dh0acc;
        if (tape) t.tape_pre = tape + (int64_t)(ub * 4 + ul) * rows * TF_TAPE_W;
      }
      if (l == 3) {
        tf_out_ptrs(p, pm, b, o);
        tf_out_ptrs(p, e.grads, b, go);
        t.out = &o, t.out_grad = &go, t.bo = b, t.F_out = F;
        if (!have_up) t.d_attnall = e.at(p->dAttnall);
        if (otape) t.tape_out = otape + (int64_t)b * rows * p->DMF;
      }
      t.post = &cur, t.post_grad = &gcur, t.post_save = tf_save(p, e, b, l), t.bp = b, t.lp = l;
      if (tape) t.tape_post = tape + (int64_t)(b * 4 + l) * rows * TF_TAPE_W;
      t.dO = dO, t.dh0acc_out = dh0acc;
      HDF_TRY(tf_token_bwd(d, t, p->dtype, e.st));
      if (have_up && l == 3) HDF_TRY(wgrad_block(ub));  // block b + 1 is complete
      TfLayerSave s = tf_save(p, e, b, l);
#ifdef HDF_ATTN_FP32  // A/B builds: the exact-fp32 attention backward in every storage mode
      HDF_TRY(tf_attention_bwd(d.N, d.M * d.B, s.qkv, s.ob, s.lse, dO, dqkv, e.st, 0));
#else
      HDF_TRY(tf_attention_bwd(d.N, d.M * d.B, s.qkv, s.ob, s.lse, dO, dqkv, e.st, p->dtype));
#endif
      up = cur, gup = gcur, ub = b, ul = l, have_up = true;
    }
  }
  TfTokenBwd t;
  t.dF = dF, t.pre = &up, t.pre_grad = &gup, t.pre_save = tf_save(p, e, 0, 0), t.bq = 0, t.lq = 0, t.F_pre = F0;
  t.dqkv = dqkv, t.dh0acc = dh0acc;
  t.tape_pre = e.f(p->tf_tape);
  HDF_TRY(tf_token_bwd(d, t, p->dtype, e.st));
  HDF_TRY(wgrad_block(0));
  HDF_TRY(tf_patch_embed_bwd(d, x, p->flat ? 1 : p->D, p->H, p->W, dF, e.grads + p->P("attns.0.patch_embeddings.weight"),
                             e.grads + p->P("attns.0.patch_embeddings.bias"),
                             e.grads + p->P("attns.0.position_embeddings"), scratch, e.st, p->flat ? 1 : 16));
  return HDF_OK;
}

// InstanceNorm(+ReLU) backward of conv layer c: da (grad w.r.t. the activation) -> dy (grad w.r.t. raw conv out)
// pre_blocks > 0: the producer of da (head_backward) already wrote that many partial rows per sample
// apply = false: only the statistics passes (k1 / ka / kb in e.inbk()); the consumer applies them itself
// kbuf: where k1 | ka | kb go (default: the Exec's scratch, overwritten by its next in_backward)
int in_backward(Exec& e, const Conv3& c, const View& da, const View& dy, int pre_blocks = 0, bool apply = true,
                float* kbuf = nullptr) {
  hdf_plan* p = e.p;
  const int64_t vox = p->vox(c.lvl);
  const int blocks = pre_blocks > 0 ? pre_blocks : hdf_in_bwd_blocks(vox, c.Cout);
  float* k = kbuf ? kbuf : e.inbk();
  float* k1 = k;
  float* ka = k + (size_t)e.B * c.Cout;
  float* kb = k + (size_t)2 * e.B * c.Cout;
  if (pre_blocks == 0)
    HDF_TRY(hdf_launch_in_bwd_reduce(p->dtype, e.at(da), da.pitch, e.at(c.y), c.y.pitch, e.f(c.st.scale),
                                     e.f(c.st.shift), e.f(c.st.mean), e.f(c.st.rstd), e.inbp(), blocks, e.B,
                                     c.Cout, vox, e.st));
  HDF_TRY(hdf_launch_in_bwd_finalize(e.inbp(), blocks, e.B, c.Cout, vox, e.P(c.gamma), e.f(c.st.rstd), k1,
                                     ka, kb, e.G(c.gamma), e.G(c.beta), e.st));
  if (!apply) return HDF_OK;
  e.wait_readers(dy);  // a side-stream weight gradient may still read this buffer's previous contents
  HDF_TRY(hdf_launch_in_bwd_apply(p->dtype, e.at(da), da.pitch, e.at(c.y), c.y.pitch, e.f(c.st.scale), e.f(c.st.shift),
                                  e.f(c.st.mean), e.f(c.st.rstd), k1, ka, kb, e.at(dy), dy.pitch, e.B, c.Cout, vox,
                                  e.st));
  return HDF_OK;
}

// conv backward: weight (and bias) gradient from (dy, input) and optionally the input gradient
// din_colsum (optional, [colsum_C] floats): += the per-channel sums over (sample, voxel) of the first colsum_C channels of
// the input gradient, taken from the dgrad conv's own InstanceNorm-partials epilogue (fp32 accumulators): the
// ConvTranspose3d bias gradient of the layer that produced those channels, without a pass over the tensor
// bs_next / bs_rows (optional): the conv whose InstanceNorm(+ReLU) backward consumes *din next.  Where the data-gradient
// launch can (hdf_conv_bwd_stats_ok) its epilogue writes the first pass of that backward into e.inbp() and *bs_rows is
// set to the rows per sample (pass it to in_backward as pre_blocks); else *bs_rows = 0.
// ap (optional): dy has NOT been written yet.  ap->da is the gradient w.r.t. c's activation and in_backward(.., apply =
// false) has left k1 | ka | kb at ap->k: the weight-gradient launch applies the second pass of the InstanceNorm backward to
// the rows it stages and writes dy as it goes (WgradArgs::ap_*); this stream waits for it before the data gradient.
struct InApply {
  const View* da;
  const float* k;
};
// the weight-gradient launch of conv layer c (without the fused pass)
static WgradArgs wgrad_args(Exec& e, const Conv3& c, const View& dy, const View& in, Xf xf) {
  const int* d = e.dm(c.lvl);
  WgradArgs w{};
  w.sm = e.at(dy);
  w.sm_pitch = dy.pitch;
  w.SC = c.Cout;
  w.lg = e.at(in);
  w.lg_pitch = in.pitch;
  w.LC = c.CinP;
  w.N = e.B;
  w.Ds = w.Dl = d[0];
  w.Hs = w.Hl = d[1];
  w.Ws = w.Wl = d[2];
  w.lg_scale = xf.scale;
  w.lg_shift = xf.shift;
  w.lg_relu = xf.relu;
  return w;
}
static void wgrad_args_apply(WgradArgs& w, Exec& e, const Conv3& c, const View& dy, const InApply& ap) {
  w.sm = e.at(*ap.da);
  w.sm_pitch = ap.da->pitch;
  w.ap_y = e.at(c.y);
  w.ap_y_pitch = c.y.pitch;
  w.ap_out = e.at(dy);
  w.ap_out_pitch = dy.pitch;
  w.ap_tab[0] = e.f(c.st.scale), w.ap_tab[1] = e.f(c.st.shift), w.ap_tab[2] = e.f(c.st.mean), w.ap_tab[3] = e.f(c.st.rstd);
  w.ap_tab[4] = ap.k, w.ap_tab[5] = ap.k + (size_t)e.B * c.Cout, w.ap_tab[6] = ap.k + (size_t)2 * e.B * c.Cout;
}

int conv_backward(Exec& e, Conv3& c, const View& dy, const View& in, Xf xf, const View* din, int accumulate,
                  const View* din2 = nullptr, float* din_colsum = nullptr, int colsum_C = 0,
                  const Conv3* bs_next = nullptr, int* bs_rows = nullptr, const InApply* ap = nullptr) {
  hdf_plan* p = e.p;
  const int* d = e.dm(c.lvl);
  WgradArgs w = wgrad_args(e, c, dy, in, xf);
  if (ap) wgrad_args_apply(w, e, c, dy, *ap);
#if !defined(HDF_NO_CONV_FIRST) && !defined(HDF_NO_WGRAD_FIRST)  // (A/B builds)
  // the encoder's first layer: K = (tap, channel) from the other side, csrc/conv_first.hip
  if (!ap && c.Cin <= 4 && !xf.scale &&
      hdf_wgrad_first_takes(p->dtype, c.Cin, c.Cout, d[0], d[1], d[2], in.pitch, dy.pitch)) {
    HDF_TRY(hdf_launch_wgrad_first(p->dtype, e.at(dy), dy.pitch, c.Cout, e.at(in), in.pitch, c.Cin, e.B, d[0], d[1], d[2],
                                   e.G(c.w), 0, e.ws + p->wgrad_ws, p->wgrad_ws_bytes, e.wgrad_stream()));
  } else
#endif
  HDF_TRY(hdf_launch_wgrad(p->dtype, 1, w, e.G(c.w), c.Cout, c.Cin, 0, e.ws + p->wgrad_ws, p->wgrad_ws_bytes,
                           e.wgrad_stream()));
  HDF_TRY(e.wgrad_done(dy));
  // (the side stream runs its launches in order, so the earlier readers of dy's previous contents are done before this
  // launch writes it; the data gradient below is the first reader of the new contents)
  if (ap) e.wait_readers(dy);
  // Conv3 layers with a bias are the UpConvs (HDenseFormer.py:162-175): conv(bias) -> InstanceNorm3d(affine=False).
  // The norm subtracts the per-(sample, channel) mean, so dL/dbias = sum_voxels dy is identically zero (the reference
  // accumulates ~3e-8 of rounding noise there); the gradient buffer was zeroed at the start of backward,
  // so the four reduction passes over dy are simply not run.
  if (din) {
    // dgrad = the same conv with taps reversed and channel roles swapped: Wd[t][ci][co] = W[co][ci][26-t]
    const int OP = round_up(c.Cin, 32);
    ConvArgs a{};
    a.prio = e.on_branch;
    a.in = e.at(dy);
    a.in_pitch = dy.pitch;
    a.Cin = c.Cout;
    a.N = e.B;
    a.Di = a.Do = d[0];
    a.Hi = a.Ho = d[1];
    a.Wi = a.Wo = d[2];
    a.w = e.ws + c.wd;
    a.wfrag = c.wd_frag;
    a.out = e.at(*din);
    a.out_pitch = din->pitch;
    a.Cout = c.Cin;
    a.CoutP = OP;
    a.accumulate = accumulate;
    if (din2) {  // input channels [0, din->C) -> din, the rest -> din2 (two dense buffers of one pitch)
      a.out2 = e.at(*din2);
      a.split = din->C;
    }
    if (din_colsum) a.stat_partials = e.statp();  // forward scratch, free during backward
    a.kpart = e.kspl(), a.kpart_bytes = HDF_KSPLIT_BYTES;
    if (bs_rows) *bs_rows = 0;
#ifndef HDF_NO_CONV_BWD_STATS  // (A/B builds)
    if (bs_next && bs_rows && !din_colsum && !din2) {
      ConvArgs b = a;
      b.stat_partials = e.inbp();
      b.bs_y = e.at(bs_next->y), b.bs_y_pitch = bs_next->y.pitch;
      b.bs_scale = e.f(bs_next->st.scale), b.bs_shift = e.f(bs_next->st.shift);
      b.bs_mean = e.f(bs_next->st.mean), b.bs_rstd = e.f(bs_next->st.rstd);
      if (bs_next->Cout == a.Cout && hdf_conv_bwd_stats_ok(p->dtype, b)) {
        a = b;
        *bs_rows = hdf_conv_stat_tiles(0, d[0], d[1], d[2], a.Cin * p->esz);
      }
    }
#endif
    HDF_TRY(hdf_launch_conv(p->dtype, 0, a, e.st));
    if (din_colsum) {
      const int rows = e.B * hdf_conv_stat_tiles(0, d[0], d[1], d[2], a.Cin * p->esz);
      HDF_TRY(hdf_launch_stat_rows_sum(e.statp(), rows, colsum_C, OP, din_colsum, e.st));
    }
  }
  return HDF_OK;
}

// InstanceNorm(+ReLU) backward of conv layer c followed by the conv's own backward (in_backward + conv_backward).  Where the
// weight-gradient launch can take the norm's second pass along (16-bit stride-1 layers: conv_wgrad2_kernel<., ., true>) that
// pass need not run on its own: one launch, the read of d(activation) + y and the write + re-read of dy by a pass that does
// nothing else (in_bwd_apply4 at 128^3 x 32 channels: 86 us alone, 165 us inside the step, three times per step).
// Measured (round 5, same box, interleaved, bench geometry):
//   one stream, sum of kernel times: -0.27 ms with every level fused (apply -0.60 ms, weight gradients +0.33 ms: the pass
//     costs ~400 VALU instructions per tile and wave in a kernel with ONE wave per SIMD);
//   the step (three streams), when this was built: fused at 128^3 only 10.88 ms, not fused 10.90 ms, fused at >= 64^3
//     11.06 ms (6 rounds each); under the round's final schedule (light kernels prioritised, forward reordered): 10.42 vs
//     10.55 vs 10.60 ms, and 10.83 ms at >= 32^3 (7-8 rounds each, medians).  The stand-alone pass is HBM-bound and runs
//     UNDER the matrix kernels of the other streams; fused, its work sits in the matrix kernels' instruction stream, and the
//     data gradient waits for the weight gradient (i.e. for whatever the side stream still holds) -- at 128^3 x 32 channels
//     the saved traffic wins, at the smaller levels (more channels: every voxel's pass is redone per large-channel block) it
//     does not.
// So the default fuses the 128^3 layers only (0.8 GB of the step's HBM traffic and three launches less, -0.13 ms); HDF_FUSED_APPLY_MIN_VOX (environment, voxels per sample) moves the threshold, HDF_NO_FUSED_APPLY switches the
// fused form off.  The UpConv chain on the branch stream (the critical path) always keeps the stand-alone pass.
static int64_t fused_apply_min_vox() {
  static const int64_t v = [] {
    const char* s = getenv("HDF_FUSED_APPLY_MIN_VOX");
    return s ? (int64_t)atoll(s) : (int64_t)128 * 128 * 128;
  }();
  return v;
}
int norm_conv_backward(Exec& e, Conv3& c, const View& da, const View& dy, int pre_blocks, const View& in, Xf xf,
                       const View* din, int accumulate, const View* din2 = nullptr, float* din_colsum = nullptr,
                       int colsum_C = 0, const Conv3* bs_next = nullptr, int* bs_rows = nullptr) {
  static const bool off = getenv("HDF_NO_FUSED_APPLY") != nullptr;  // A/B knob (tests/test_gpu_knobs.py)
  hdf_plan* p = e.p;
  const int* d = e.dm(c.lvl);
  InApply ap{&da, e.inbk()};
  bool fuse = !off && !e.on_branch && p->vox(c.lvl) >= fused_apply_min_vox();
  if (fuse) {
#if !defined(HDF_NO_CONV_FIRST) && !defined(HDF_NO_WGRAD_FIRST)
    if (c.Cin <= 4 && !xf.scale && hdf_wgrad_first_takes(p->dtype, c.Cin, c.Cout, d[0], d[1], d[2], in.pitch, dy.pitch))
      fuse = false;  // the first layer's own kernel
#endif
    WgradArgs w = wgrad_args(e, c, dy, in, xf);
    wgrad_args_apply(w, e, c, dy, ap);
    fuse = fuse && hdf_wgrad_apply_takes(p->dtype, 1, w);
  }
  HDF_TRY(in_backward(e, c, da, dy, pre_blocks, !fuse));
  return conv_backward(e, c, dy, in, xf, din, accumulate, din2, din_colsum, colsum_C, bs_next, bs_rows,
                       fuse ? &ap : nullptr);
}

// ConvTranspose3d backward: dOut (hi-res) -> dIn (lo-res, grad w.r.t. the activation fed to the convT)
int convt_backward(Exec& e, ConvT3& t, const View& dout, const View& in, Xf xf, const View& din) {
  hdf_plan* p = e.p;
  const int* d = e.dm(t.lvl_in);
  // (the bias gradient comes out of the epilogue of the dgrad conv that produced dout: conv_backward)
  WgradArgs w{};
  w.sm = e.at(in);
  w.sm_pitch = in.pitch;
  w.SC = t.Cin;
  w.lg = e.at(dout);
  w.lg_pitch = dout.pitch;
  w.LC = t.Cout;
  w.N = e.B;
  w.Ds = d[0], w.Hs = d[1], w.Ws = d[2];
  w.Dl = (p->flat ? 1 : 2) * d[0], w.Hl = 2 * d[1], w.Wl = 2 * d[2];
  w.sm_scale = xf.scale;
  w.sm_shift = xf.shift;
  w.sm_relu = xf.relu;
  HDF_TRY(hdf_launch_wgrad(p->dtype, 2, w, e.G(t.w), t.Cin, t.Cout, 0, e.ws + p->wgrad_ws, p->wgrad_ws_bytes,
                           e.wgrad_stream()));
  HDF_TRY(e.wgrad_done(dout));
  // dX[i][ci] = sum_k sum_co dY[2i-1+k][co] * W[ci][co][k]  -> stride-2 gather conv, packed [tap][CinP][Cout]
  const int OP = round_up(t.Cin, 32);
  ConvArgs a{};
  a.in = e.at(dout);
  a.in_pitch = dout.pitch;
  a.Cin = t.Cout;
  a.N = e.B;
  a.Di = (p->flat ? 1 : 2) * d[0], a.Hi = 2 * d[1], a.Wi = 2 * d[2];
  a.Do = d[0], a.Ho = d[1], a.Wo = d[2];
  a.w = e.ws + t.wd;
  a.wfrag = t.wd_frag;
  a.out = e.at(din);
  a.out_pitch = din.pitch;
  a.Cout = t.Cin;
  a.CoutP = OP;
  return hdf_launch_conv(p->dtype, 1, a, e.st);
}

// fuse_in: the conv layer whose InstanceNorm+ReLU output the head reads -- the head gradient is then that activation's
// complete gradient, and the kernel also writes the first pass of the layer's InstanceNorm backward (*pre_blocks rows
// per sample in inb_partials; 0 when the table does not hold that many rows and the separate pass has to run)
int head_backward(Exec& e, const Head1& h, const void* dlogits, const View& in, Xf xf, const View& dx, int acc,
                  const Conv3* fuse_in = nullptr, int* pre_blocks = nullptr) {
  hdf_plan* p = e.p;
  const int hb = hdf_head_bwd_blocks(p->vox(h.lvl));
  const bool fuse = fuse_in && pre_blocks && hb <= 1024;
  if (pre_blocks) *pre_blocks = fuse ? hb : 0;
  return hdf_launch_head_bwd(p->dtype, dlogits, e.at(in), in.pitch, xf.scale, xf.shift, e.P(h.w), e.at(dx), dx.pitch,
                             acc, e.G(h.w), e.G(h.b), e.B, h.C, p->ncls, p->vox(h.lvl), e.st,
                             fuse ? e.f(fuse_in->st.mean) : nullptr, fuse ? e.f(fuse_in->st.rstd) : nullptr,
                             fuse ? e.inbp() : nullptr);
}

}  // namespace

// A persistent transformer launch that gave up (transformer_chain.hip: chain_wait) leaves 1 + a workgroup id in its timeout
// word.  NaN written into the branch output does NOT survive the network -- relu(InstanceNorm(.)) is fmaxf(x * scale +
// shift, 0), and fmaxf returns the operand that is not NaN -- so the step would end with finite, plausible-looking logits
// and gradients.  These two launches (one workgroup each, the LAST launch of a forward / of a backward on the caller's
// stream, only when the persistent kernels ran) make the failure visible in the data itself: the first rows of every output
// / the head of the flat gradient buffer become NaN, so the loss, and the optimizer step, are NaN.
template <typename T>
__global__ void chain_poison_outputs_kernel(const unsigned* __restrict__ tmo, T* o0, T* o1, T* o2, T* o3, int n0, int n1,
                                            int n2, int n3) {
  if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
  T* o[4] = {o0, o1, o2, o3};
  const int n[4] = {n0, n1, n2, n3};
  for (int k = 0; k < 4; k++)
    for (int i = threadIdx.x; i < n[k]; i += blockDim.x) ST<T>::st(o[k] + i, __builtin_nanf(""));
}
__global__ void chain_poison_grads_kernel(const unsigned* __restrict__ tmo, float* grads, int n) {
  if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
  for (int i = threadIdx.x; i < n; i += blockDim.x) grads[i] = __builtin_nanf("");
}

// Stand-in for a collective's kernel (tests / tools): `workgroups` workgroups of 256 threads that each hold `lds_bytes` of
// LDS (160 KiB = a compute unit of its own) and `vgprs` vector registers per lane (0: a handful; 128: what a RCCL
// all-reduce kernel holds -- next to it a 304-register conv wave still fits a SIMD, a 512-register conv_wr wave or the
// two 252-register waves of a persistent transformer workgroup do not), spinning on the 100 MHz real-time counter for `usec`.
template <bool FAT>
__global__ __launch_bounds__(256) void occupy_kernel(unsigned ticks) {
  extern __shared__ char occ_lds[];
  if (threadIdx.x == 0) occ_lds[0] = 1;
  if (FAT) asm volatile("v_mov_b32 v127, 0" ::: "v127");   // forces an allocation of 128 VGPRs
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)ticks) __builtin_amdgcn_s_sleep(8);
}

// ================================================================================================ C ABI
extern "C" {

const char* hdf_version(void) { return "hdf-hip 0.1 (gfx950)"; }
int hdf_set_cu_budget(int cus) {
  HDF_CHECK_ARG(cus >= 8 && cus <= 256 && cus % 8 == 0, "cu budget %d: a multiple of 8 in [8, 256]", cus);
  g_cu_budget.store(cus, std::memory_order_relaxed);
  return HDF_OK;
}
const char* hdf_last_error(void) { return g_err; }

static int create_plan(int in_channels, int n_cls, int n_filters, int D, int H, int W, int transformer_depth, int dtype,
                       bool is2d, hdf_plan** out, bool flat = false) {
  HDF_CHECK_ARG(out != nullptr, "plan_create: null out");
  HDF_CHECK_ARG(in_channels >= 1 && in_channels <= 8, "plan_create: in_channels=%d unsupported (1..8)", in_channels);
  HDF_CHECK_ARG(n_cls >= 2 && n_cls <= 8, "plan_create: n_cls=%d unsupported (2..8)", n_cls);
  HDF_CHECK_ARG(n_filters >= 16 && n_filters % 16 == 0 && n_filters <= 64,
                "plan_create: n_filters=%d unsupported (multiple of 16, 16..64)", n_filters);
  HDF_CHECK_ARG(D % 16 == 0 && H % 16 == 0 && W % 16 == 0 && (D >= 32 || is2d) && H >= 32 && W >= 32,
                "plan_create: image_size (%d,%d,%d) must be multiples of 16 and >= 32", D, H, W);
  HDF_CHECK_ARG(transformer_depth >= 4, "plan_create: transformer_depth=%d < 4", transformer_depth);
  HDF_CHECK_ARG(dtype == HDF_F32 || dtype == HDF_BF16 || dtype == HDF_F16, "plan_create: dtype %d", dtype);
  hdf_plan* p = new hdf_plan();
  p->M = in_channels;
  p->ncls = n_cls;
  p->nf = n_filters;
  p->D = D, p->H = H, p->W = W;
  p->td = transformer_depth;
  p->nb = transformer_depth / 4;
  p->dtype = dtype;
  p->esz = hdf_esz(dtype);
  for (int l = 0; l < 5; l++) p->dims[l][0] = D >> l, p->dims[l][1] = H >> l, p->dims[l][2] = W >> l;
  p->flat = is2d && flat;
  if (p->flat)   // (p->D stays 16: the depth of the patch embedding's input copy)
    for (int l = 0; l < 5; l++) p->dims[l][0] = 1;
  p->DM = 4 * n_filters;
  p->DMF = p->DM + 128;
  p->Ntok = (D / 16) * (H / 16) * (W / 16);
  build_params(p);
  build_layers(p);
  p->is2d = is2d;
  if (is2d) {
    int rc = build_params2d(p);
    if (rc != HDF_OK) {
      delete p;
      return rc;
    }
  }
  *out = p;
  return HDF_OK;
}

int hdf_plan_create(int in_channels, int n_cls, int n_filters, int D, int H, int W, int transformer_depth, int dtype,
                    hdf_plan** out) {
  return create_plan(in_channels, n_cls, n_filters, D, H, W, transformer_depth, dtype, false, out);
}
int hdf_plan_create_2d_embedded(int in_channels, int n_cls, int n_filters, int H, int W, int transformer_depth, int dtype,
                               hdf_plan** out) {
  return create_plan(in_channels, n_cls, n_filters, 16, H, W, transformer_depth, dtype, true, out, false);
}
int hdf_plan_create_2d(int in_channels, int n_cls, int n_filters, int H, int W, int transformer_depth, int dtype,
                       hdf_plan** out) {
  return create_plan(in_channels, n_cls, n_filters, 16, H, W, transformer_depth, dtype, true, out, true);
}

void hdf_plan_destroy(hdf_plan* p) { delete p; }
int64_t hdf_plan_num_params(const hdf_plan* p) { return (int64_t)(p->is2d ? p->params2d : p->params).size(); }
int64_t hdf_plan_param_floats(const hdf_plan* p) { return p->is2d ? p->total_floats2d : p->total_floats; }

int hdf_plan_param_info(const hdf_plan* p, int64_t idx, char* name, int name_cap, int64_t* offset, int64_t* numel,
                        int* ndim, int64_t* shape5) {
  const std::vector<ParamInfo>& tbl = p->is2d ? p->params2d : p->params;
  HDF_CHECK_ARG(idx >= 0 && idx < (int64_t)tbl.size(), "param_info: index %lld", (long long)idx);
  const ParamInfo& pi = tbl[idx];
  if (name && name_cap > 0) {
    strncpy(name, pi.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (offset) *offset = pi.offset;
  if (numel) *numel = pi.numel;
  if (ndim) *ndim = (int)pi.shape.size();
  if (shape5)
    for (size_t i = 0; i < 5; i++) shape5[i] = i < pi.shape.size() ? pi.shape[i] : 1;
  return HDF_OK;
}

int64_t hdf_plan_workspace_bytes(hdf_plan* p, int batch) {
  layout(p, batch);
  return (int64_t)p->ws_bytes;
}
int64_t hdf_plan_inference_workspace_bytes(hdf_plan* p, int batch) {
  layout(p, batch);
  return (int64_t)p->ws_fwd_bytes;
}

int hdf_plan_buffer_info(hdf_plan* p, int batch, const char* name, int64_t* byte_offset, int64_t* pitch_elems,
                         int* channels, int* d, int* h, int* w) {
  layout(p, batch);
  auto it = p->bufs.find(name);
  HDF_CHECK_ARG(it != p->bufs.end(), "buffer_info: no buffer named '%s'", name);
  const View& v = it->second;
  *byte_offset = (int64_t)v.off;
  *pitch_elems = v.pitch;
  *channels = v.C;
  *d = p->dims[v.lvl][0];
  *h = p->dims[v.lvl][1];
  *w = p->dims[v.lvl][2];
  return HDF_OK;
}

int hdf_plan_region_info(hdf_plan* p, int batch, const char* name, int64_t* byte_offset, int64_t* bytes) {
  HDF_CHECK_ARG(p && name && byte_offset && bytes, "region_info: null argument");
  layout(p, batch);
  const int64_t rows = (int64_t)p->M * batch * p->Ntok;
  const std::string n = name;
  if (n == "tf_F") *byte_offset = (int64_t)p->tf_F, *bytes = (int64_t)p->nb * rows * p->DMF * 4;
  else if (n == "tf_save") *byte_offset = (int64_t)p->tf_save, *bytes = (int64_t)p->nb * 4 * rows * 232 * 4;
  else if (n == "tf_sync") *byte_offset = (int64_t)p->tf_sync, *bytes = (int64_t)3 << 20;
  else if (n == "tf_dF") *byte_offset = (int64_t)p->tf_dF, *bytes = rows * p->DMF * 4;
  else if (n == "tf_tape") *byte_offset = (int64_t)p->tf_tape, *bytes = (int64_t)p->nb * 4 * rows * TF_TAPE_W * 4;
  else if (n == "tf_otape") *byte_offset = (int64_t)p->tf_otape, *bytes = (int64_t)p->nb * rows * p->DMF * 4;
  else {
    hdf_set_error("region_info: no region named '%s'", name);
    return HDF_ERR_ARG;
  }
  return HDF_OK;
}

static int forward3d(hdf_plan* p, const float* x, const float* params, void* workspace, int64_t workspace_bytes,
                     void* out0, void* out1, void* out2, void* out3, int batch, int training, uint64_t seed,
                     hdf_stream stream);
static int backward3d(hdf_plan* p, const float* x, const float* params, void* workspace, int64_t workspace_bytes,
                      const void* dout0, const void* dout1, const void* dout2, const void* dout3, float* grads,
                      int batch, int stages, hdf_stream stream, hipEvent_t* bev = nullptr);

int hdf_forward(hdf_plan* p, const float* x, const float* params, void* workspace, int64_t workspace_bytes, void* out0,
                void* out1, void* out2, void* out3, int batch, int training, uint64_t seed, hdf_stream stream) {
  HDF_CHECK_ARG(p && x && params && workspace, "forward: null argument");
  if (!p->is2d)
    return forward3d(p, x, params, workspace, workspace_bytes, out0, out1, out2, out3, batch, training, seed, stream);
  // 2-D model: x [B,C,H,W], params = the 2-D flat buffer, outputs [B,n_cls,H/2^i,W/2^i]
  layout(p, batch);
  HDF_CHECK_ARG((size_t)workspace_bytes >= p->ws_fwd_bytes, "forward: workspace %lld < %zu bytes",
                (long long)workspace_bytes, p->ws_fwd_bytes);
  char* ws = (char*)workspace;
  hipStream_t st = (hipStream_t)stream;
  float* p3 = (float*)(ws + p->e_params3d);
  float* x3 = (float*)(ws + p->e_x3d);
  HDF_TRY(launch_embed2d(p, params, p3, st));
  if (p->flat)
    // native 2-D path: depth-1 tensors throughout; the logits [B, n_cls, 1, H, W] ARE the 2-D outputs, and the patch
    // embedding contracts the input's 16 x 16 patches with depth slice 0 of the embedded kernels (kd = 1)
    return forward3d(p, x, p3, workspace, workspace_bytes, out0, out1, out2, out3, batch, training, seed, stream);
  const int64_t hw = (int64_t)p->H * p->W;
  hipLaunchKernelGGL(replicate_depth_kernel, dim3(2048), dim3(256), 0, st, x, x3, (int64_t)batch * p->M, p->D, hw);
  HDF_LAUNCH_CHECK();
  void* o3[4];
  for (int i = 0; i < 4; i++) o3[i] = ws + p->e_out3d[i];
  HDF_TRY(forward3d(p, x3, p3, workspace, workspace_bytes, o3[0], o3[1], o3[2], o3[3], batch, training, seed, stream));
  void* o2[4] = {out0, out1, out2, out3};
  for (int i = 0; i < 4; i++)
    HDF_TRY(launch_depth_slice(p->dtype, o3[i], o2[i], (int64_t)batch * p->ncls, p->dims[i][0],
                               (int64_t)p->dims[i][1] * p->dims[i][2], 1, st));
  return HDF_OK;
}

static int forward3d(hdf_plan* p, const float* x, const float* params, void* workspace, int64_t workspace_bytes,
                     void* out0, void* out1, void* out2, void* out3, int batch, int training, uint64_t seed,
                     hdf_stream stream) {
  layout(p, batch);
  HDF_CHECK_ARG((size_t)workspace_bytes >= p->ws_fwd_bytes, "forward: workspace %lld < %zu bytes",
                (long long)workspace_bytes, p->ws_fwd_bytes);
  HDF_TRY(chain_flag_check(p));
  p->training = training ? 1 : 0;
  p->seed = (uint32_t)(seed & 0xffffffffu);
  p->tf_fwd_chain = tf_use_chain(p, batch);
  if (p->tf_fwd_chain) HDF_TRY(chain_flag_ensure(p));
  Exec e{p, (char*)workspace, params, nullptr, batch, (hipStream_t)stream};
  const int nf = p->nf;
  const int ch[4] = {nf, 2 * nf, 4 * nf, 8 * nf};
  void* outs[4] = {out0, out1, out2, out3};
  Xf none;

  // ---- multi-path transformer (HDenseFormer.py:230) -> attnall, then the UpConv chain (:231-235).  Nothing on the
  // encoder's first level depends on it before ds0 = block_1_2_left(..) + at3 (:238), and it is ~100 launches of
  // latency-bound token / attention kernels plus low-resolution convs: it runs on the plan's BRANCH stream next to the
  // two 128^3 encoder convs of the caller's stream (matrix-bound at the package power cap, one wave per SIMD) and is
  // joined in front of the level-0 encoder tail.  HDF_NO_BRANCH_OVERLAP=1 keeps everything on the caller's stream.
  static const bool no_branch = getenv("HDF_NO_BRANCH_OVERLAP") != nullptr;  // A/B knob (tests/test_gpu_knobs.py)
  Exec eb = e;
  hipStream_t bst = no_branch ? nullptr : e.fork_branch();
  if (bst) eb.st = bst, eb.on_branch = true;
  struct Rejoin {  // error returns below must not leave branch-stream work unordered behind the caller's stream
    Exec &e, &eb;
    bool armed;
    ~Rejoin() {
      if (armed) (void)e.join_branch(eb);
    }
  } rejoin{e, eb, bst != nullptr};
#ifdef HDF_NO_FUSED_AT3  // A/B builds
  const bool fused_at3 = false;
#else
  const bool fused_at3 = !p->flat;   // (the 2-D model: materialised at3 + the 2-D encoder tail)
#endif
  const int flat = p->flat ? 1 : 0;
  // (round 5) Order of the first launches.  The first level-0 conv reads the fp32 weights itself (csrc/conv_first.hip) and
  // needs only the converted input, so where that kernel takes the layer the caller's stream starts with conversion +
  // conv, and the weight packs -- every conv's 16-bit panels and the persistent transformer kernel's fragment-major copies,
  // ~80 us of light kernels -- go to the BRANCH stream in front of the patch embedding: the conv runs beside them instead
  // of behind them, and the transformer kernel (which holds every unit and therefore effectively starts when that conv
  // ends) starts ~120 us earlier.  `packed` orders the second conv (and the branch's own convs, by stream order) behind the
  // packs.  Without a branch stream, or where the generic conv takes the first layer, the packs stay in front.
#if defined(HDF_NO_CONV_FIRST) || defined(HDF_NO_PACK_ON_BRANCH)
  const bool first_direct = false;
#else
  const bool first_direct = bst && p->enc[0][0].Cin <= 4 &&
                            hdf_conv_first_takes(p->dtype, p->enc[0][0].Cin, p->enc[0][0].Cout, p->dims[0][0], p->dims[0][1],
                                                 p->dims[0][2], p->xin.pitch);
#endif
  hipStream_t pst = first_direct ? bst : e.st;
  if (first_direct) {
    HDF_TRY(hdf_launch_nchw_to_ndhwc(p->dtype, x, e.at(p->xin), batch, p->M, 16, p->vox(0), e.st));
    HDF_TRY(conv_forward(e, p->enc[0][0], p->xin, none));
  }
  HDF_TRY(hdf_launch_pack_batch(p->dtype, params, e.ws, p->pack_jobs.data(), (int)p->pack_jobs.size(), pst));
  if (p->tf_fwd_chain)
    HDF_TRY(tf_chain_pack(tf_dims(p, batch), tf_chain_params(p), p->nb, params, e.ws + p->tf_wpack, pst));
  hipEvent_t packed = nullptr;
  if (bst) {
    packed = e.next_event();
    if (!packed || hipEventRecord(packed, pst) != hipSuccess) {
      hdf_set_error("branch stream: event failed");
      return HDF_ERR_HIP;
    }
    if (first_direct && hipStreamWaitEvent(e.st, packed, 0) != hipSuccess) {
      hdf_set_error("branch stream: wait failed");
      return HDF_ERR_HIP;
    }
  }
  // the caller's stream first (4 launches), then the ~65 launches of the branch: the host issues launches one after the
  // other, and whatever is issued second starts that much later when the host is not far ahead of the GPU
  if (!first_direct) {
    HDF_TRY(hdf_launch_nchw_to_ndhwc(p->dtype, x, e.at(p->xin), batch, p->M, 16, p->vox(0), e.st));
    HDF_TRY(conv_forward(e, p->enc[0][0], p->xin, none));
  }
  // (round 5) The second level-0 conv takes three quarters of the compute units: it runs while the branch stream works
  // through deep_conv / up1..3 (the persistent transformer kernel in front of them holds every unit, so the order on
  // the device is conv_first, transformer, then this conv NEXT TO the UpConv chain), and the chain's low-resolution
  // convs cannot share a unit with a persistent 128^3 workgroup (LDS, registers): at 256 workgroups they queued behind it
  // (deep_conv: 291 us instead of 72) and the caller's stream then waited 220 us for at3.  192 of 256: conv 277 -> 363 us,
  // at3 ready 85 us earlier (tools/timeline.py, profiles/r05_forward_timeline.txt).  The grid is the same in every
  // stream arrangement: the InstanceNorm partial sums are grouped per workgroup, and tests/test_gpu_knobs.py compares
  // arrangements bit for bit.  HDF_NO_L0_BUDGET: A/B builds.
  // With the persistent transformer kernel the device order conv_first -> transformer -> this conv is made explicit: the
  // kernel needs every unit, and this conv's 192 workgroups in front of it would leave it spinning on the other 64 for the
  // conv's whole duration.  (The launch chain of small kernels co-runs with the conv instead: no wait.)
  const bool chain_first = first_direct && p->tf_fwd_chain;
  eb.tf_packed = first_direct ? nullptr : packed;   // (packs on the branch stream itself: stream order)
  if (chain_first) {
    HDF_TRY(transformer_forward(eb, x));
    hipEvent_t tf_done = e.next_event();
    if (!tf_done || hipEventRecord(tf_done, bst) != hipSuccess || hipStreamWaitEvent(e.st, tf_done, 0) != hipSuccess) {
      hdf_set_error("branch stream: event failed");
      return HDF_ERR_HIP;
    }
  }
#ifndef HDF_NO_L0_BUDGET
  e.conv_budget = (hdf_cu_budget() * 3 / 4) & ~7;
#endif
  HDF_TRY(conv_forward(e, p->enc[0][1], p->enc[0][0].y, xf_of(e, p->enc[0][0])));
  e.conv_budget = 0;
  if (!chain_first) HDF_TRY(transformer_forward(eb, x));
  if (packed && hipStreamWaitEvent(bst, packed, 0) != hipSuccess) {
    hdf_set_error("branch stream: wait failed");
    return HDF_ERR_HIP;
  }
  HDF_TRY(conv_forward(eb, p->deep, p->attnall, none));
  HDF_TRY(hdf_launch_upsample_fwd(p->dtype, eb.at(p->deep.y), p->deep.y.pitch, eb.f(p->deep.st.scale),
                                  eb.f(p->deep.st.shift), eb.at(p->attnout), p->attnout.pitch, batch, 8 * nf,
                                  p->dims[4][0], p->dims[4][1], p->dims[4][2], eb.st, flat));
  {
    const View* src = &p->attnout;
    for (int k = 0; k < 3; k++) {  // up1 -> at1 (lvl 2), up2 -> at2 (lvl 1), up3 -> at3 (lvl 0)
      Conv3& c = p->up[k];
      HDF_TRY(conv_forward(eb, c, *src, none));
      const View& dst = p->at[2 - k];
      // at3 (k == 2) is not materialised: the level-0 encoder tail interpolates it from up3's output (enc_tail_up_kernel)
      if (k == 2 && fused_at3) break;
      HDF_TRY(hdf_launch_upsample_fwd(p->dtype, eb.at(c.y), c.y.pitch, eb.f(c.st.scale), eb.f(c.st.shift), eb.at(dst),
                                      dst.pitch, batch, c.Cout, p->dims[c.lvl][0], p->dims[c.lvl][1],
                                      p->dims[c.lvl][2], eb.st, flat));
      src = &dst;
    }
  }
  // ---- encoder (:237-244)
  const View* cur = &p->xin;
  for (int k = 0; k < 4; k++) {
    if (k > 0) {  // (level 0: issued above, in front of the branch)
      HDF_TRY(conv_forward(e, p->enc[k][0], *cur, none));
      HDF_TRY(conv_forward(e, p->enc[k][1], p->enc[k][0].y, xf_of(e, p->enc[k][0])));
    }
    Conv3& c = p->enc[k][1];
    if (k == 0 && bst) {  // at1..3 / attnout are needed from here on
      HDF_TRY(e.join_branch(eb));
      rejoin.armed = false;
    }
    if (k < 3) {
      View ds = subview(p, p->cat[k], ch[k], ch[k]);
      // ds_k = relu(IN(y)) + at_k ; pooled = MaxPool(ds_k): one fused pass
      if (k == 0 && fused_at3) {
        Conv3& u = p->up[2];  // at3 = Upsample(relu(IN(up3 conv))), evaluated inside the pass
        HDF_TRY(hdf_launch_enc_tail_up(p->dtype, e.at(c.y), c.y.pitch, e.f(c.st.scale), e.f(c.st.shift), e.at(u.y), u.y.pitch,
                                       e.f(u.st.scale), e.f(u.st.shift), e.at(ds), ds.pitch, e.at(p->pooled[k]),
                                       p->pooled[k].pitch, (uint8_t*)(e.ws + p->pool_idx[k]), batch, ch[k],
                                       p->dims[k + 1][0], p->dims[k + 1][1], p->dims[k + 1][2], e.st));
      } else
      HDF_TRY(hdf_launch_enc_tail(p->dtype, e.at(c.y), c.y.pitch, e.f(c.st.scale), e.f(c.st.shift), e.at(p->at[k]),
                                  p->at[k].pitch, e.at(ds), ds.pitch, e.at(p->pooled[k]), p->pooled[k].pitch,
                                  (uint8_t*)(e.ws + p->pool_idx[k]), batch, ch[k], p->dims[k + 1][0], p->dims[k + 1][1],
                                  p->dims[k + 1][2], e.st, flat));
      cur = &p->pooled[k];
    } else {
      HDF_TRY(hdf_launch_norm_relu_add(p->dtype, e.at(c.y), c.y.pitch, e.f(c.st.scale), e.f(c.st.shift),
                                       e.at(p->attnout), p->attnout.pitch, e.at(p->x4), p->x4.pitch, batch, ch[3],
                                       p->vox(3), e.st));
    }
  }
  // ---- decoder (:246-253)
  HDF_TRY(head_forward(e, p->head[3], p->x4, none, outs[3]));
  const View* dec_in = &p->x4;
  Xf dec_xf = none;
  for (int k = 2; k >= 0; k--) {
    View up_out = subview(p, p->cat[k], 0, ch[k]);
    HDF_TRY(convt_forward(e, p->upc[k], *dec_in, dec_xf, up_out));
    HDF_TRY(conv_forward(e, p->dec[k][0], p->cat[k], none, k == 0));   // k == 0: block_1_1_right, the probed launch
    HDF_TRY(conv_forward(e, p->dec[k][1], p->dec[k][0].y, xf_of(e, p->dec[k][0])));
    dec_in = &p->dec[k][1].y;
    dec_xf = xf_of(e, p->dec[k][1]);
    HDF_TRY(head_forward(e, p->head[k], *dec_in, dec_xf, outs[k]));
  }
  if (p->tf_fwd_chain) {  // (see chain_poison_outputs_kernel: a launch that gave up must not leave plausible outputs)
    const unsigned* tmo = reinterpret_cast<const unsigned*>(e.ws + p->tf_sync) + p->M * batch * 32;
    int n[4];
    for (int i = 0; i < 4; i++) n[i] = (int)std::min<int64_t>(p->vox(i) * p->ncls * batch, 4096);
    if (p->dtype == HDF_F32)
      hipLaunchKernelGGL(chain_poison_outputs_kernel<float>, dim3(1), dim3(256), 0, e.st, tmo, (float*)outs[0], (float*)outs[1],
                         (float*)outs[2], (float*)outs[3], n[0], n[1], n[2], n[3]);
    else if (p->dtype == HDF_BF16)
      hipLaunchKernelGGL(chain_poison_outputs_kernel<bf16_t>, dim3(1), dim3(256), 0, e.st, tmo, (bf16_t*)outs[0],
                         (bf16_t*)outs[1], (bf16_t*)outs[2], (bf16_t*)outs[3], n[0], n[1], n[2], n[3]);
    else
      hipLaunchKernelGGL(chain_poison_outputs_kernel<f16_t>, dim3(1), dim3(256), 0, e.st, tmo, (f16_t*)outs[0], (f16_t*)outs[1],
                         (f16_t*)outs[2], (f16_t*)outs[3], n[0], n[1], n[2], n[3]);
    HDF_LAUNCH_CHECK();
  }
  return HDF_OK;
}

int hdf_backward(hdf_plan* p, const float* x, const float* params, void* workspace, int64_t workspace_bytes,
                 const void* dout0, const void* dout1, const void* dout2, const void* dout3, float* grads, int batch,
                 hdf_stream stream) {
  return hdf_backward_stages(p, x, params, workspace, workspace_bytes, dout0, dout1, dout2, dout3, grads, batch, 7,
                             stream);
}

static int backward_any(hdf_plan* p, const float* x, const float* params, void* workspace, int64_t workspace_bytes,
                        const void* dout0, const void* dout1, const void* dout2, const void* dout3, float* grads,
                        int batch, int stages, hdf_stream stream, hipEvent_t* bev);

int hdf_backward_stages(hdf_plan* p, const float* x, const float* params, void* workspace, int64_t workspace_bytes,
                        const void* dout0, const void* dout1, const void* dout2, const void* dout3, float* grads,
                        int batch, int stages, hdf_stream stream) {
  return backward_any(p, x, params, workspace, workspace_bytes, dout0, dout1, dout2, dout3, grads, batch, stages, stream,
                      nullptr);
}

int hdf_backward_events(hdf_plan* p, const float* x, const float* params, void* workspace, int64_t workspace_bytes,
                        const void* dout0, const void* dout1, const void* dout2, const void* dout3, float* grads,
                        int batch, hdf_stream stream, void** bucket_events) {
  HDF_CHECK_ARG(p && bucket_events, "backward_events: null argument");
  for (int k = 0; k < HDF_NUM_GRAD_BUCKETS; k++) {
    if (!p->bucket_ev[k] && hipEventCreateWithFlags(&p->bucket_ev[k], hipEventDisableTiming) != hipSuccess) {
      p->bucket_ev[k] = nullptr;
      hdf_set_error("backward_events: could not create an event");
      return HDF_ERR_HIP;
    }
    bucket_events[k] = (void*)p->bucket_ev[k];
  }
  return backward_any(p, x, params, workspace, workspace_bytes, dout0, dout1, dout2, dout3, grads, batch, 7, stream,
                      p->bucket_ev);
}

int hdf_plan_set_chain_timeout_us(hdf_plan* p, int64_t usec) {
  HDF_CHECK_ARG(p && usec >= 100 && usec <= 30000000, "plan_set_chain_timeout_us: 100 us .. 30 s");
  p->chain_ticks = (unsigned)(usec * 100);   // s_memrealtime: 100 MHz
  return HDF_OK;
}

int hdf_plan_force_persistent(hdf_plan* p, int on) {
  HDF_CHECK_ARG(p != nullptr, "plan_force_persistent: null plan");
  p->chain_forced = on != 0;
  return HDF_OK;
}

int hdf_plan_chain_state(hdf_plan* p, int batch, int* persistent, int* gave_up_workgroup) {
  HDF_CHECK_ARG(p && batch >= 1, "plan_chain_state: null plan / batch < 1");
  // (reads the host-mapped word like the next forward would, without consuming it: that call still reports the error)
  const unsigned pending = p->chain_flag ? __atomic_load_n(p->chain_flag, __ATOMIC_ACQUIRE) : 0u;
  if (gave_up_workgroup) *gave_up_workgroup = pending ? (int)pending - 1 : (p->chain_last_giveup ? (int)p->chain_last_giveup - 1 : -1);
  if (persistent) *persistent = (!pending && tf_use_chain(p, batch)) ? 1 : 0;
  return HDF_OK;
}

int hdf_op_occupy(int workgroups, int lds_bytes, int vgprs, int usec, hdf_stream stream) {
  HDF_CHECK_ARG(workgroups >= 1 && workgroups <= 4096 && lds_bytes >= 0 && lds_bytes <= 160 * 1024 && usec >= 1 &&
                    usec <= 10000000 && (vgprs == 0 || vgprs == 128),
                "op_occupy: workgroups 1..4096, lds 0..160 KiB, vgprs 0 or 128, 1 us..10 s");
  const void* fn = vgprs ? reinterpret_cast<const void*>(occupy_kernel<true>) : reinterpret_cast<const void*>(occupy_kernel<false>);
  if (lds_bytes > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
    hdf_set_error("op_occupy: hipFuncSetAttribute failed");
    return HDF_ERR_HIP;
  }
  if (vgprs)
    hipLaunchKernelGGL(occupy_kernel<true>, dim3(workgroups), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (unsigned)usec * 100u);
  else
    hipLaunchKernelGGL(occupy_kernel<false>, dim3(workgroups), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (unsigned)usec * 100u);
  HDF_LAUNCH_CHECK();
  return HDF_OK;
}

int hdf_plan_grad_bucket(const hdf_plan* p, int k, int64_t* lo, int64_t* hi) {
  HDF_CHECK_ARG(p && lo && hi && k >= 0 && k < HDF_NUM_GRAD_BUCKETS, "plan_grad_bucket: bucket 0..%d", HDF_NUM_GRAD_BUCKETS - 1);
  HDF_CHECK_ARG(!p->is2d, "plan_grad_bucket: the 2-D plan's gradients are final together (one bucket: the whole buffer)");
  // state_dict order: attns.* | deep_conv, up1..3 | block_1_*_left | block_2_*_left .. block_4_*_left | upconv_3 .. heads
  const int64_t chain = p->P("deep_conv.double_conv.0.weight"), enc0 = p->P("block_1_1_left.conv.weight"),
                enc1 = p->P("block_2_1_left.conv.weight"), dec = p->P("upconv_3.weight"), end = p->total_floats;
  HDF_CHECK_ARG(0 < chain && chain < enc0 && enc0 < enc1 && enc1 < dec && dec < end, "plan_grad_bucket: unexpected parameter order");
  const int64_t b[HDF_NUM_GRAD_BUCKETS][2] = {{dec, end}, {chain, enc0}, {0, chain}, {enc1, dec}, {enc0, enc1}};
  *lo = b[k][0], *hi = b[k][1];
  return HDF_OK;
}

int hdf_plan_set_probe(hdf_plan* p, void* ev_start, void* ev_stop) {
  HDF_CHECK_ARG(p && ((ev_start == nullptr) == (ev_stop == nullptr)), "plan_set_probe: both events or none");
  p->probe_start = (hipEvent_t)ev_start, p->probe_stop = (hipEvent_t)ev_stop;
  return HDF_OK;
}

int hdf_stream_wait_event(hdf_stream stream, void* event) {
  HDF_CHECK_ARG(event != nullptr, "stream_wait_event: null event");
  if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0) != hipSuccess) {
    hdf_set_error("hipStreamWaitEvent failed");
    return HDF_ERR_HIP;
  }
  return HDF_OK;
}

static int backward_any(hdf_plan* p, const float* x, const float* params, void* workspace, int64_t workspace_bytes,
                        const void* dout0, const void* dout1, const void* dout2, const void* dout3, float* grads,
                        int batch, int stages, hdf_stream stream, hipEvent_t* bev) {
  HDF_CHECK_ARG(p && x && params && workspace && grads, "backward: null argument");
  if (!p->is2d)
    return backward3d(p, x, params, workspace, workspace_bytes, dout0, dout1, dout2, dout3, grads, batch, stages, stream,
                      bev);
  // 2-D model: the forward left the replicated input and the embedded parameters in the workspace
  HDF_CHECK_ARG(p->batch == batch && (size_t)workspace_bytes >= p->ws_bytes, "backward: batch / workspace mismatch");
  char* ws = (char*)workspace;
  hipStream_t st = (hipStream_t)stream;
  void* d3[4];
  const void* d2[4] = {dout0, dout1, dout2, dout3};
  for (int i = 0; i < 4; i++) {
    if (p->flat) {   // the 2-D logit gradients are the depth-1 tensors the native path reads
      d3[i] = const_cast<void*>(d2[i]);
      continue;
    }
    d3[i] = ws + p->e_dout3d[i];
    if (stages & 1)  // the loss sees depth slice 0 only
      HDF_TRY(launch_depth_slice(p->dtype, d3[i], const_cast<void*>(d2[i]), (int64_t)batch * p->ncls, p->dims[i][0],
                                 (int64_t)p->dims[i][1] * p->dims[i][2], 0, st));
  }
  float* g3 = (float*)(ws + p->e_grads3d);
  if (p->flat)   // (x: the patch embedding's weight gradient reads the 2-D input itself)
    HDF_TRY(backward3d(p, x, (const float*)(ws + p->e_params3d), workspace, workspace_bytes, d3[0], d3[1], d3[2], d3[3], g3,
                       batch, stages, stream));
  else
  HDF_TRY(backward3d(p, (const float*)(ws + p->e_x3d), (const float*)(ws + p->e_params3d), workspace, workspace_bytes,
                     d3[0], d3[1], d3[2], d3[3], g3, batch, stages, stream));
  HDF_TRY(launch_extract2d(p, stages, g3, grads, st));
  if (bev) {  // the 2-D gradients exist only after the extraction: every bucket is final here
    for (int k = 0; k < HDF_NUM_GRAD_BUCKETS; k++)
      if (hipEventRecord(bev[k], st) != hipSuccess) {
        hdf_set_error("backward: could not record a bucket event");
        return HDF_ERR_HIP;
      }
  }
  return HDF_OK;
}

// ---- UpConv chain backward: at3 <- up3 <- at2 <- up2 <- at1 <- up1 <- attnout <- deep_conv <- attnall
static int upconv_chain_backward(Exec& e, int batch) {
  hdf_plan* p = e.p;
  Xf none;
  for (int k = 2; k >= 0; k--) {
    Conv3& c = p->up[k];                                   // up[k] output level c.lvl, upsampled to level c.lvl-1
    View dat = p->dSkip[c.lvl - 1];  // gradient of at_{..} == of ds
    const int* d = p->dims[c.lvl];
    View& da = p->dUa[4 - c.lvl];
    View& dy = p->dUy[4 - c.lvl];
    HDF_TRY(hdf_launch_upsample_bwd(p->dtype, e.at(dat), dat.pitch, e.at(da), da.pitch, batch, c.Cout, d[0], d[1], d[2],
                                    e.st, p->flat ? 1 : 0));
    HDF_TRY(in_backward(e, c, da, dy));
    // input of up[k]: attnout (k==0) or at_{lvl} ; its gradient buffer already holds the skip-path gradient
    const View& cin = (k == 0) ? p->attnout : p->at[c.lvl];
    View din = (k == 0) ? p->dX4 : p->dSkip[c.lvl];
    HDF_TRY(conv_backward(e, c, dy, cin, none, &din, 1));
  }
  {
    Conv3& c = p->deep;
    const int* d = p->dims[4];
    HDF_TRY(hdf_launch_upsample_bwd(p->dtype, e.at(p->dX4), p->dX4.pitch, e.at(p->dUa[0]), p->dUa[0].pitch, batch,
                                    c.Cout, d[0], d[1], d[2], e.st, p->flat ? 1 : 0));
    HDF_TRY(in_backward(e, c, p->dUa[0], p->dUy[0]));
    HDF_TRY(conv_backward(e, c, p->dUy[0], p->attnall, none, &p->dAttnall, 0));
  }
  return HDF_OK;
}

// bev (optional, stages == 7): HDF_NUM_GRAD_BUCKETS events, recorded where the parameter gradients of a bucket
// (hdf_plan_grad_bucket: 0 decoder + heads, 1 UpConv chain, 2 transformer branches, 3 encoder levels 1-3, 4 encoder
// level 0) are final -- on the caller's stream, the side stream or the branch stream, whichever finishes them -- so that a
// communication stream can start a bucket's all-reduce while the rest of this one call is still running (no staged calls,
// the branch-stream fork stays).  Round 6: five buckets instead of three.  With one "encoder / decoder / heads" bucket 26
// of the 62 MB became final with the LAST kernel of the backward (the first encoder layer's weight gradient) and their
// all-reduce was fully exposed (profiles/r06_timeline_standin_32cu_300us.txt: two of three stand-in collectives ran
// behind the backward); now the decoder + heads (final a third of the way into the backward) and the encoder's levels
// 1-3 (final before the UpConv chain's backward starts) are reduced under the rest, and what is final at the very end is
// the first level's two layers: 0.1 MB.
static int backward3d(hdf_plan* p, const float* x, const float* params, void* workspace, int64_t workspace_bytes,
                      const void* dout0, const void* dout1, const void* dout2, const void* dout3, float* grads,
                      int batch, int stages, hdf_stream stream, hipEvent_t* bev) {
  HDF_CHECK_ARG(p->batch == batch, "backward: batch %d differs from the forward's %d", batch, p->batch);
  HDF_TRY(chain_flag_check(p));
  auto record = [&](int k, hipStream_t s) -> int {
    if (bev && hipEventRecord(bev[k], s) != hipSuccess) {
      hdf_set_error("backward: could not record the event of gradient bucket %d", k);
      return HDF_ERR_HIP;
    }
    return HDF_OK;
  };
  enum { BK_DEC = 0, BK_CHAIN = 1, BK_TF = 2, BK_ENC = 3, BK_ENC0 = 4 };
  // "everything enqueued so far on the caller's stream AND on the side stream": the side stream (where the bucket's conv
  // weight gradients run) waits for the caller's position (InstanceNorm / head / bias gradients) and carries the event.
  // Every side-stream launch already waits for the caller's position of its own launch point, so this orders nothing new.
  auto record_joined = [&](int k, Exec& ex) -> int {
    if (!bev) return HDF_OK;
    if (ex.async && p->side) {
      hipEvent_t f = ex.next_event();
      if (!f || hipEventRecord(f, ex.st) != hipSuccess || hipStreamWaitEvent(p->side, f, 0) != hipSuccess) {
        hdf_set_error("backward: could not order the side stream for the event of gradient bucket %d", k);
        return HDF_ERR_HIP;
      }
      return record(k, p->side);
    }
    return record(k, ex.st);
  };
  HDF_CHECK_ARG((size_t)workspace_bytes >= p->ws_bytes,
                "backward: workspace of %lld bytes holds a forward only (hdf_plan_workspace_bytes = %zu)",
                (long long)workspace_bytes, p->ws_bytes);
  Exec e{p, (char*)workspace, params, grads, batch, (hipStream_t)stream};
  static const bool no_async = getenv("HDF_NO_ASYNC_WGRAD") != nullptr;  // A/B knob: everything on the caller's stream
  if (!no_async) {
    if (!p->side) {
      // lowest priority: when a data-gradient conv (critical path) and a weight gradient are both ready, the data
      // gradient gets the CUs first and the weight gradient then runs next to the memory-bound passes that follow it
      int least = 0, greatest = 0;
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
          hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, least) != hipSuccess) {
        if (hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) != hipSuccess) p->side = nullptr;
      }
    }
    e.async = p->side != nullptr;
  }
  const int nf = p->nf;
  const int ch[4] = {nf, 2 * nf, 4 * nf, 8 * nf};
  const void* douts[4] = {dout0, dout1, dout2, dout3};
  Xf none;
  // one call for all three stages: stages 2 and 4 fork onto the branch stream inside stage 1 (see the encoder loop).
  // Both streams send their weight gradients through the side stream (in order: one shared workspace), so the fork
  // needs it.  Staged calls (gradient buckets of hdf_rt.parallel) keep the three stages in order on the caller's stream.
  static const bool no_branch = getenv("HDF_NO_BRANCH_OVERLAP") != nullptr;  // A/B knob (tests/test_gpu_knobs.py)
  const bool fork_ok = stages == 7 && e.async && !no_branch;
  bool forked = false;
  Exec eb = e;
  eb.last_side = nullptr;
  // every return path (also the HDF_TRY error returns) orders the side and branch streams behind the caller's stream:
  // the caller may free or reuse the workspace / gradient buffers as soon as its own stream gets there
  struct Rejoin {
    Exec &e, &eb;
    bool& forked;
    ~Rejoin() {
      if (forked) (void)e.join_branch(eb);
      e.join();
    }
  } rejoin{e, eb, forked};
  if (stages & 1) {
  hipError_t me = hipMemsetAsync(grads, 0, (size_t)p->total_floats * sizeof(float), e.st);
  if (me != hipSuccess) {
    hdf_set_error("backward: memset failed: %s", hipGetErrorString(me));
    return HDF_ERR_HIP;
  }

  // ---- decoder, top (level 0) down to level 2
  for (int k = 0; k <= 2; k++) {
    Conv3 &c1 = p->dec[k][0], &c2 = p->dec[k][1];
    // gA[k] holds d/d(activation of c2): head gradient (+ convT input gradient from the level above, k>0)
    int pre = 0;
    HDF_TRY(head_backward(e, p->head[k], douts[k], c2.y, xf_of(e, c2), p->gA[k], k > 0 ? 1 : 0, &c2, &pre));
    int bsr = 0;  // the data-gradient conv may leave the first pass of c1's InstanceNorm backward behind (level 0)
    HDF_TRY(norm_conv_backward(e, c2, p->gA[k], p->gY[k], pre, c1.y, xf_of(e, c1), &p->gA[k], 0, nullptr, nullptr, 0, &c1,
                               &bsr));
    // the upconv half of d(cat) is the gradient of upconv_{k+1}'s output: its bias gradient rides on this conv
    float* up_db = e.G(p->upc[k].b);
    if (p->dcat_split[k])
      HDF_TRY(norm_conv_backward(e, c1, p->gA[k], p->gY2[k], bsr, p->cat[k], none, &p->dUp[k], 0, &p->dSkip[k], up_db,
                                 ch[k]));
    else
      HDF_TRY(norm_conv_backward(e, c1, p->gA[k], p->gY2[k], bsr, p->cat[k], none, &p->dCat[k], 0, nullptr, up_db, ch[k]));
    // upconv_{k+1}: input is dec[k+1][1] activation (k<2) or the bottleneck x4 (k==2)
    const View& dup = p->dUp[k];
    if (k < 2)
      HDF_TRY(convt_backward(e, p->upc[k], dup, p->dec[k + 1][1].y, xf_of(e, p->dec[k + 1][1]), p->gA[k + 1]));
    else
      HDF_TRY(convt_backward(e, p->upc[k], dup, p->x4, none, p->dX4));
  }
  HDF_TRY(head_backward(e, p->head[3], douts[3], p->x4, none, p->dX4, 1));
  HDF_TRY(record_joined(BK_DEC, e));   // upconv_1..3, block_*_right, the four heads: nothing below touches their gradients

  // ---- encoder, bottom (level 3) up to level 0.  dskip: gradient of ds_k (= of the transformer feature at_k too)
  for (int k = 3; k >= 0; k--) {
    Conv3 &c1 = p->enc[k][0], &c2 = p->enc[k][1];
    View dskip = (k == 3) ? p->dX4 : p->dSkip[k];
    int pre = 0;
    if (k < 3) {
      // ds_k also feeds pool_{k+1}: with that gradient added d(ds_k) is complete, and the pass that adds it takes the first
      // pass of c2's InstanceNorm backward along (pre rows per sample in the partials table)
      pre = hdf_maxpool_bwd_in_blocks((int64_t)p->dims[k + 1][0] * p->dims[k + 1][1] * p->dims[k + 1][2], ch[k]);
      HDF_TRY(hdf_launch_maxpool_bwd_in(p->dtype, e.at(p->dP[k]), p->dP[k].pitch, (const uint8_t*)(e.ws + p->pool_idx[k]),
                                        e.at(dskip), dskip.pitch, e.at(c2.y), c2.y.pitch, e.f(c2.st.scale),
                                        e.f(c2.st.shift), e.f(c2.st.mean), e.f(c2.st.rstd), e.inbp(), batch, ch[k],
                                        p->dims[k + 1][0], p->dims[k + 1][1], p->dims[k + 1][2], e.st, p->flat ? 1 : 0));
    }
    if (k == 0 && fork_ok) {
      // d(ds_0) = d(at3) is final.  What is left: (1) the UpConv chain backward, (2) the transformer branches' backward,
      // (3) the level-0 encoder backward (two InstanceNorm backward passes at 128^3, a 32->32 data-gradient conv, two
      // weight gradients).  (1) -> (2) is the critical path (~2.1 ms, of which (2) is ~100 latency-bound launches that
      // leave most of the chip idle); (3) is 1.4 ms of heavy kernels nothing waits for.  So (1) + (2) go to the BRANCH
      // stream, and the caller's stream runs (3) NEXT TO (2): it waits for the end of (1) first -- issued together, the
      // persistent convs of (3) held every CU while the chain's small kernels queued behind them (a 5 us
      // in_bwd_finalize waited 208 us for a slot; 3.5 ms from here to the end of the step instead of 2.4).
      // The chain only READS d(ds_k) of the levels the encoder has already finished with (it accumulates into
      // dSkip[1], dSkip[2] and dX4, which the loop above consumed at k = 1, 2, 3).
      hipStream_t bst = e.fork_branch();
      if (bst) {
        eb.st = bst, eb.on_branch = true, eb.async = e.async;
        forked = true;
        HDF_TRY(upconv_chain_backward(eb, batch));
        hipEvent_t chain_done = e.next_event();
        if (!chain_done || hipEventRecord(chain_done, bst) != hipSuccess ||
            hipStreamWaitEvent(e.st, chain_done, 0) != hipSuccess) {
          hdf_set_error("branch stream: event failed");
          return HDF_ERR_HIP;
        }
        if (bev) {
          // bucket 2 = the chain's conv weight gradients: launched on the side stream (every one of them is enqueued by
          // now), the rest of the chain on the branch stream.  The side stream's later launches belong to the level-0
          // encoder work, which the caller's stream starts behind chain_done anyway.
          if (hipStreamWaitEvent(p->side, chain_done, 0) != hipSuccess) {
            hdf_set_error("side stream: event failed");
            return HDF_ERR_HIP;
          }
          HDF_TRY(record(BK_CHAIN, p->side));
        }
      }
    }
    int bsr = 0;
    HDF_TRY(norm_conv_backward(e, c2, dskip, p->gY[k], pre, c1.y, xf_of(e, c1), &p->gA[k], 0, nullptr, nullptr, 0, &c1,
                               &bsr));
    bool first_fused = false;
#if !defined(HDF_NO_CONV_FIRST) && !defined(HDF_NO_WGRAD_FIRST) && !defined(HDF_NO_WGRAD_FIRST_IN)
    // The first layer has no input gradient: the second pass of its InstanceNorm backward would write dy (268 MB at the
    // benchmark size) only for the weight gradient to read it back.  wgrad_first_kernel applies that pass to the rows it
    // stages (from d(activation) and y) instead: one pass over two tensors less on the caller's stream.
    if (k == 0 && hdf_wgrad_first_takes(p->dtype, c1.Cin, c1.Cout, p->dims[0][0], p->dims[0][1], p->dims[0][2],
                                        p->xin.pitch, p->gA[k].pitch)) {
      float* kk = e.f(p->inb_k3);
      HDF_TRY(in_backward(e, c1, p->gA[k], p->gY2[k], bsr, false, kk));
      const WgradFirstIn fi{e.at(c1.y), c1.y.pitch, e.f(c1.st.scale), e.f(c1.st.shift), e.f(c1.st.mean), e.f(c1.st.rstd),
                            kk, kk + (size_t)e.B * c1.Cout, kk + (size_t)2 * e.B * c1.Cout};
      HDF_TRY(hdf_launch_wgrad_first(p->dtype, e.at(p->gA[k]), p->gA[k].pitch, c1.Cout, e.at(p->xin), p->xin.pitch, c1.Cin,
                                     e.B, p->dims[0][0], p->dims[0][1], p->dims[0][2], e.G(c1.w), 0, e.ws + p->wgrad_ws,
                                     p->wgrad_ws_bytes, e.wgrad_stream(), &fi));
      HDF_TRY(e.wgrad_done(p->gA[k]));
      first_fused = true;
    }
#endif
    if (!first_fused) {
      if (k > 0)
        HDF_TRY(norm_conv_backward(e, c1, p->gA[k], p->gY2[k], bsr, p->pooled[k - 1], none, &p->dP[k - 1], 0));
      else
        HDF_TRY(norm_conv_backward(e, c1, p->gA[k], p->gY2[k], bsr, p->xin, none, nullptr, 0));
    }
    if (k == 1) HDF_TRY(record_joined(BK_ENC, e));   // block_2_* .. block_4_*_left: the encoder below the top level is done
    // (host order: the ten level-0 launches of the caller's stream first, then the ~100 of the transformer backward)
    if (k == 0 && forked) {
      HDF_TRY(transformer_backward(eb, x));
      if (bev) {
        eb.join();  // (the branch's own side-stream launches, if any)
        HDF_TRY(record(BK_TF, eb.st));
      }
    }
  }

  if (!(stages & 6) || bev) e.join();  // staged call (gradient buckets) / bucket event: final here
  HDF_TRY(record(BK_ENC0, e.st));
  }  // stage 1: every gradient of the encoder / decoder / head parameters is final here
  if (forked) {
    HDF_TRY(e.join_branch(eb));
    forked = false;
  } else {
    if (stages & 2) {
      HDF_TRY(upconv_chain_backward(e, batch));
      if (!(stages & 4) || bev) e.join();  // staged call: final when it returns; else the transformer branches run under them
      HDF_TRY(record(BK_CHAIN, e.st));
    }  // stage 2: deep_conv / up1..3 gradients are final
    if (stages & 4) {
      HDF_TRY(transformer_backward(e, x));
      if (bev) {
        e.join();
        HDF_TRY(record(BK_TF, e.st));
      }
    }
  }
  e.join();
  if ((stages & 4) && p->tf_bwd_chain) {  // (chain_poison_grads_kernel: the persistent backward's timeout word, second half)
    const unsigned* tmo = reinterpret_cast<const unsigned*>(e.ws + p->tf_sync) + (1 << 17) + p->M * batch * 32;
    hipLaunchKernelGGL(chain_poison_grads_kernel, dim3(1), dim3(256), 0, e.st, tmo, grads,
                       (int)std::min<int64_t>(p->total_floats, 4096));
    HDF_LAUNCH_CHECK();
  }
  return HDF_OK;
}

// ---------------------------------------------------------------------------------------- loss / metric / adam
int64_t hdf_loss_workspace_bytes(int batch) { return (int64_t)hdf_loss_workspace_floats(batch, 4) * sizeof(float); }

int hdf_loss_forward(int dtype, const void* out0, const void* out1, const void* out2, const void* out3, int nscale,
                     const float* target_onehot, int batch, int n_cls, int D, int H, int W, void* workspace,
                     float* loss_out, hdf_stream stream) {
  const void* outs[4] = {out0, out1, out2, out3};
  return hdf_launch_loss_fwd(dtype, outs, target_onehot, nscale, batch, n_cls, D, H, W, (float*)workspace, loss_out,
                             (hipStream_t)stream);
}
int hdf_loss_backward(int dtype, const void* out0, const void* out1, const void* out2, const void* out3, int nscale,
                      const float* target_onehot, int batch, int n_cls, int D, int H, int W, const void* workspace,
                      const float* grad_out, void* dout0, void* dout1, void* dout2, void* dout3, hdf_stream stream) {
  const void* outs[4] = {out0, out1, out2, out3};
  void* douts[4] = {dout0, dout1, dout2, dout3};
  return hdf_launch_loss_bwd(dtype, outs, target_onehot, nscale, batch, n_cls, D, H, W, (const float*)workspace,
                             grad_out, douts, (hipStream_t)stream);
}
int hdf_loss_terms_forward(int dtype, const void* out0, const void* out1, const void* out2, const void* out3, int nscale,
                           const float* target_onehot, int batch, int n_cls, int D, int H, int W, float ce_weight,
                           float dice_weight, void* workspace, float* loss_out, hdf_stream stream) {
  const void* outs[4] = {out0, out1, out2, out3};
  return hdf_launch_loss_fwd(dtype, outs, target_onehot, nscale, batch, n_cls, D, H, W, (float*)workspace, loss_out,
                             (hipStream_t)stream, ce_weight, dice_weight);
}
int hdf_loss_terms_backward(int dtype, const void* out0, const void* out1, const void* out2, const void* out3,
                            int nscale, const float* target_onehot, int batch, int n_cls, int D, int H, int W,
                            float ce_weight, float dice_weight, const void* workspace, const float* grad_out,
                            void* dout0, void* dout1, void* dout2, void* dout3, hdf_stream stream) {
  const void* outs[4] = {out0, out1, out2, out3};
  void* douts[4] = {dout0, dout1, dout2, dout3};
  return hdf_launch_loss_bwd(dtype, outs, target_onehot, nscale, batch, n_cls, D, H, W, (const float*)workspace,
                             grad_out, douts, (hipStream_t)stream, ce_weight, dice_weight);
}
int hdf_loss_weighted_forward(int dtype, const void* out0, const void* out1, const void* out2, const void* out3,
                              int nscale, const float* target_onehot, int batch, int n_cls, int D, int H, int W,
                              float ce_weight, float dice_weight, const float* class_weight, int dice_ignore_index,
                              void* workspace, float* loss_out, hdf_stream stream) {
  const void* outs[4] = {out0, out1, out2, out3};
  return hdf_launch_loss_fwd(dtype, outs, target_onehot, nscale, batch, n_cls, D, H, W, (float*)workspace, loss_out,
                             (hipStream_t)stream, ce_weight, dice_weight, class_weight, dice_ignore_index);
}
int hdf_loss_weighted_backward(int dtype, const void* out0, const void* out1, const void* out2, const void* out3,
                               int nscale, const float* target_onehot, int batch, int n_cls, int D, int H, int W,
                               float ce_weight, float dice_weight, const float* class_weight, int dice_ignore_index,
                               const void* workspace, const float* grad_out, void* dout0, void* dout1, void* dout2,
                               void* dout3, hdf_stream stream) {
  const void* outs[4] = {out0, out1, out2, out3};
  void* douts[4] = {dout0, dout1, dout2, dout3};
  return hdf_launch_loss_bwd(dtype, outs, target_onehot, nscale, batch, n_cls, D, H, W, (const float*)workspace,
                             grad_out, douts, (hipStream_t)stream, ce_weight, dice_weight, class_weight,
                             dice_ignore_index);
}
int hdf_dice_counts(int dtype, const void* logits, const float* target_onehot, int batch, int n_cls, int64_t voxels,
                    uint64_t* counts, hdf_stream stream) {
  return hdf_launch_dice_counts(dtype, logits, target_onehot, batch, n_cls, voxels, (unsigned long long*)counts,
                                (hipStream_t)stream);
}
int hdf_confusion_matrix(int dtype, const void* logits, const float* target_onehot, int batch, int n_cls,
                         int64_t voxels, uint64_t* confusion, int accumulate, hdf_stream stream) {
  return hdf_launch_confusion(dtype, logits, target_onehot, batch, n_cls, voxels, (unsigned long long*)confusion,
                              accumulate, (hipStream_t)stream);
}
int hdf_confusion_matrix_labels(const uint8_t* target, const uint8_t* prediction, int n_cls, int64_t n,
                                uint64_t* confusion, int accumulate, hdf_stream stream) {
  HDF_CHECK_ARG(target && prediction && confusion, "confusion_matrix_labels: null argument");
  return hdf_launch_confusion_labels(target, prediction, n_cls, n, (unsigned long long*)confusion, accumulate,
                                     (hipStream_t)stream);
}
int64_t hdf_normalize_workspace_bytes(int channels) { return (int64_t)hdf_norm_ws_bytes(channels); }
int hdf_normalize_mr(float* image, int channels, int64_t voxels, void* workspace, hdf_stream stream) {
  HDF_CHECK_ARG(image && workspace, "normalize_mr: null argument");
  return hdf_launch_normalize(image, channels, voxels, 0, 0.f, 1.f, workspace, (hipStream_t)stream);
}
int hdf_normalize_petct(float* image, int channels, int64_t voxels, float mean, float w, void* workspace,
                        hdf_stream stream) {
  HDF_CHECK_ARG(image && workspace, "normalize_petct: null argument");
  return hdf_launch_normalize(image, channels, voxels, 1, mean, w, workspace, (hipStream_t)stream);
}
int hdf_sw_accumulate(int dtype, const void* logits, int n_cls, int pd, int ph, int pw, float* prob_sum, float* count,
                      int D, int H, int W, int z0, int y0, int x0, hdf_stream stream) {
  HDF_CHECK_ARG(logits && prob_sum && count, "sw_accumulate: null argument");
  return hdf_launch_sw_accumulate(dtype, logits, n_cls, pd, ph, pw, prob_sum, count, D, H, W, z0, y0, x0,
                                  (hipStream_t)stream);
}
int hdf_sw_finalize(const float* prob_sum, const float* count, int n_cls, int64_t voxels, uint8_t* label,
                    hdf_stream stream) {
  HDF_CHECK_ARG(prob_sum && count && label, "sw_finalize: null argument");
  return hdf_launch_sw_finalize(prob_sum, count, n_cls, voxels, label, (hipStream_t)stream);
}
int hdf_onehot_from_labels(const uint8_t* labels, float* onehot, int batch, int n_cls, int64_t voxels,
                           hdf_stream stream) {
  HDF_CHECK_ARG(labels && onehot, "onehot_from_labels: null argument");
  return hdf_launch_onehot(labels, onehot, batch, n_cls, voxels, (hipStream_t)stream);
}
int hdf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* decay_mask,
                  int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  float grad_scale, hdf_stream stream) {
  HDF_CHECK_ARG(step >= 1, "adam: step=%d must start at 1", step);
  return hdf_launch_adam(params, grads, exp_avg, exp_avg_sq, decay_mask, n, lr, beta1, beta2, eps, weight_decay, step,
                         grad_scale, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------- operator level
int hdf_op_to_channels_last(int dtype, const float* x, void* out, int N, int C, int CP, int64_t voxels,
                            hdf_stream stream) {
  return hdf_launch_nchw_to_ndhwc(dtype, x, out, N, C, CP, voxels, (hipStream_t)stream);
}
int hdf_op_pack_weights(int dtype, const float* src, void* dst, int O, int I, int OP, int IP, int64_t so, int64_t si,
                        int flip, hdf_stream stream) {
  return hdf_launch_pack_w(dtype, src, dst, O, I, OP, IP, so, si, flip, (hipStream_t)stream);
}
int hdf_op_conv3d(int dtype, int mode, const void* in, int64_t in_pitch, int Cin, int N, int Di, int Hi, int Wi,
                  const void* w_packed, const float* bias, const float* in_scale, const float* in_shift, int in_relu,
                  void* out, int64_t out_pitch, int Cout, float* stat_partials, int accumulate, hdf_stream stream) {
  ConvArgs a{};
  a.in = in;
  a.in_pitch = in_pitch;
  a.Cin = Cin;
  a.N = N;
  a.Di = Di, a.Hi = Hi, a.Wi = Wi;
  if (mode == 0)
    a.Do = Di, a.Ho = Hi, a.Wo = Wi;
  else if (mode == 1)
    a.Do = Di / 2, a.Ho = Hi / 2, a.Wo = Wi / 2;
  else
    a.Do = 2 * Di, a.Ho = 2 * Hi, a.Wo = 2 * Wi;
  if (Di == 1) a.Do = 1;   // depth 1 selects the 2-D operator: the depth axis is never strided (include/hdf.h)
  a.w = w_packed;
  a.bias = bias;
  a.in_scale = in_scale;
  a.in_shift = in_shift;
  a.in_relu = in_relu;
  a.out = out;
  a.out_pitch = out_pitch;
  a.Cout = Cout;
  a.CoutP = round_up(Cout, 32);
  a.stat_partials = stat_partials;
  a.accumulate = accumulate;
  return hdf_launch_conv(dtype, mode, a, (hipStream_t)stream);
}
int hdf_op_conv3d_bwd_stats(int dtype, const void* in, int64_t in_pitch, int Cin, int N, int D, int H, int W,
                            const void* w_packed, void* out, int64_t out_pitch, int Cout, const void* y, int64_t y_pitch,
                            const float* scale, const float* shift, const float* mean, const float* rstd,
                            float* partials, hdf_stream stream) {
  HDF_CHECK_ARG(in && w_packed && out && y && scale && shift && mean && rstd && partials, "conv3d_bwd_stats: null argument");
  ConvArgs a{};
  a.in = in, a.in_pitch = in_pitch, a.Cin = Cin, a.N = N;
  a.Di = a.Do = D, a.Hi = a.Ho = H, a.Wi = a.Wo = W;
  a.w = w_packed, a.out = out, a.out_pitch = out_pitch, a.Cout = Cout, a.CoutP = round_up(Cout, 32);
  a.stat_partials = partials;
  a.bs_y = y, a.bs_y_pitch = y_pitch, a.bs_scale = scale, a.bs_shift = shift, a.bs_mean = mean, a.bs_rstd = rstd;
  HDF_CHECK_ARG(hdf_conv_bwd_stats_ok(dtype, a), "conv3d_bwd_stats: this launch cannot take the statistics epilogue "
                "(16-bit storage, 32 -> 32 channels, whole 4x8x8 tiles, >= 48^3)");
  return hdf_launch_conv(dtype, 0, a, (hipStream_t)stream);
}
int hdf_op_conv3d_split(int dtype, const void* in, int64_t in_pitch, int Cin, int N, int D, int H, int W,
                        const void* w_packed, void* out, void* out2, int64_t out_pitch, int Cout, int split,
                        float* stat_partials, float* colsum, int colsum_C, hdf_stream stream) {
  HDF_CHECK_ARG(colsum == nullptr || (stat_partials != nullptr && colsum_C > 0 && colsum_C <= Cout),
                "conv3d_split: column sums need the statistics buffer and 0 < colsum_C <= Cout");
  ConvArgs a{};
  a.in = in;
  a.in_pitch = in_pitch;
  a.Cin = Cin;
  a.N = N;
  a.Di = a.Do = D, a.Hi = a.Ho = H, a.Wi = a.Wo = W;
  a.w = w_packed;
  a.out = out;
  a.out2 = out2;
  a.split = split;
  a.out_pitch = out_pitch;
  a.Cout = Cout;
  a.CoutP = round_up(Cout, 32);
  a.stat_partials = stat_partials;
  HDF_TRY(hdf_launch_conv(dtype, 0, a, (hipStream_t)stream));
  if (colsum) {
    const int rows = N * hdf_conv_stat_tiles(0, D, H, W, Cin * hdf_esz(dtype));
    HDF_TRY(hdf_launch_stat_rows_sum(stat_partials, rows, colsum_C, a.CoutP, colsum, (hipStream_t)stream));
  }
  return HDF_OK;
}
int hdf_op_conv3d_first(int dtype, const void* in, int64_t in_pitch, int Cin, int N, int D, int H, int W, const float* weight,
                        const float* bias, void* out, int64_t out_pitch, int Cout, float* stat_partials,
                        hdf_stream stream) {
  HDF_CHECK_ARG(in && weight && out, "conv3d_first: null argument");
  return hdf_launch_conv_first(dtype, in, in_pitch, Cin, N, D, H, W, weight, bias, out, out_pitch, Cout, stat_partials,
                               (hipStream_t)stream);
}
int hdf_op_conv3d_first_wgrad(int dtype, const void* dy, int64_t dy_pitch, int Cout, const void* x, int64_t x_pitch,
                              int Cin, int N, int D, int H, int W, float* dweight, int accumulate, void* workspace,
                              int64_t workspace_bytes, hdf_stream stream) {
  HDF_CHECK_ARG(dy && x && dweight && workspace, "conv3d_first_wgrad: null argument");
  return hdf_launch_wgrad_first(dtype, dy, dy_pitch, Cout, x, x_pitch, Cin, N, D, H, W, dweight, accumulate, workspace,
                                (size_t)workspace_bytes, (hipStream_t)stream);
}
int hdf_op_conv3d_first_wgrad_in(int dtype, const void* da, int64_t da_pitch, int Cout, const void* y, int64_t y_pitch,
                                 const float* scale, const float* shift, const float* mean, const float* rstd,
                                 const float* k1, const float* ka, const float* kb, const void* x, int64_t x_pitch, int Cin,
                                 int N, int D, int H, int W, float* dweight, int accumulate, void* workspace,
                                 int64_t workspace_bytes, hdf_stream stream) {
  HDF_CHECK_ARG(da && x && dweight && workspace, "conv3d_first_wgrad_in: null argument");
  const WgradFirstIn fi{y, y_pitch, scale, shift, mean, rstd, k1, ka, kb};
  return hdf_launch_wgrad_first(dtype, da, da_pitch, Cout, x, x_pitch, Cin, N, D, H, W, dweight, accumulate, workspace,
                                (size_t)workspace_bytes, (hipStream_t)stream, &fi);
}
int hdf_op_conv3d_wr(int dtype, const void* in, int64_t in_pitch, int Cin, int N, int D, int H, int W,
                     const void* w_packed, const float* bias, const float* in_scale, const float* in_shift, int in_relu,
                     void* out, int64_t out_pitch, int Cout, float* stat_partials, int accumulate, hdf_stream stream) {
  ConvArgs a{};
  a.in = in, a.in_pitch = in_pitch, a.Cin = Cin, a.N = N;
  a.Di = a.Do = D, a.Hi = a.Ho = H, a.Wi = a.Wo = W;
  a.w = w_packed, a.bias = bias, a.in_scale = in_scale, a.in_shift = in_shift, a.in_relu = in_relu;
  a.out = out, a.out_pitch = out_pitch, a.Cout = Cout, a.CoutP = round_up(Cout, 32);
  a.stat_partials = stat_partials, a.accumulate = accumulate;
  HDF_CHECK_ARG(a.in_pitch % 8 == 0 && (((uintptr_t)a.in) & 15) == 0, "conv_wr: input view must be 16-byte aligned");
  if (!hdf_conv_wr_can(dtype, a)) {
    hdf_set_error("conv3d_wr: shape not handled by the weights-in-registers kernel (dtype %d Cin %d %dx%dx%d)", dtype, Cin, D, H, W);
    return HDF_ERR_UNSUPPORTED;
  }
  return hdf_launch_conv_wr(dtype, a, (hipStream_t)stream);
}
int hdf_op_conv3d_stat_tiles(int dtype, int Cin, int Do, int Ho, int Wo) {
  return hdf_conv_stat_tiles(0, Do, Ho, Wo, Cin * hdf_esz(dtype));
}
int64_t hdf_op_wgrad_workspace_bytes(int stride, int N, int Ds, int Hs, int Ws, int SC, int LC) {
  return (int64_t)hdf_wgrad_workspace_bytes(stride, N, Ds, Hs, Ws, SC, LC);
}
int hdf_op_conv3d_wgrad(int dtype, int stride, const void* sm, int64_t sm_pitch, int SC, const void* lg,
                        int64_t lg_pitch, int LC, int N, int Ds, int Hs, int Ws, const float* sm_scale,
                        const float* sm_shift, int sm_relu, const float* lg_scale, const float* lg_shift, int lg_relu,
                        float* dw, int sc_store, int lc_store, int accumulate, void* workspace, int64_t workspace_bytes,
                        hdf_stream stream) {
  WgradArgs w{};
  w.sm = sm, w.sm_pitch = sm_pitch, w.SC = SC;
  w.lg = lg, w.lg_pitch = lg_pitch, w.LC = LC;
  w.N = N;
  w.Ds = Ds, w.Hs = Hs, w.Ws = Ws;
  w.Dl = Ds == 1 ? 1 : stride * Ds, w.Hl = stride * Hs, w.Wl = stride * Ws;   // (depth 1: the 2-D operator)
  w.sm_scale = sm_scale, w.sm_shift = sm_shift, w.sm_relu = sm_relu;
  w.lg_scale = lg_scale, w.lg_shift = lg_shift, w.lg_relu = lg_relu;
  return hdf_launch_wgrad(dtype, stride, w, dw, sc_store, lc_store, accumulate, workspace, (size_t)workspace_bytes,
                          (hipStream_t)stream);
}
int hdf_op_in_finalize(const float* partials, int N, int tiles, int C, int CP, int64_t voxels, const float* gamma,
                       const float* beta, float eps, float* mean, float* rstd, float* scale, float* shift,
                       hdf_stream stream) {
  return hdf_launch_in_finalize(partials, N, tiles, C, CP, voxels, gamma, beta, eps, mean, rstd, scale, shift,
                                (hipStream_t)stream);
}
int64_t hdf_op_in_bwd_workspace_floats(int N, int C, int64_t voxels) {
  return (int64_t)N * hdf_in_bwd_blocks(voxels, C) * C * 2 + (int64_t)3 * N * C;
}
int hdf_op_in_bwd(int dtype, const void* da, int64_t da_pitch, const void* y, int64_t y_pitch, const float* scale,
                  const float* shift, const float* mean, const float* rstd, const float* gamma, void* dy,
                  int64_t dy_pitch, float* dgamma, float* dbeta, int N, int C, int64_t voxels, float* workspace,
                  hdf_stream stream) {
  HDF_CHECK_ARG(da && y && scale && shift && mean && rstd && dy && workspace, "in_bwd: null argument");
  const int blocks = hdf_in_bwd_blocks(voxels, C);
  float* partials = workspace;
  float* k1 = workspace + (int64_t)N * blocks * C * 2;
  float* ka = k1 + (int64_t)N * C;
  float* kb = ka + (int64_t)N * C;
  hipStream_t st = (hipStream_t)stream;
  HDF_TRY(hdf_launch_in_bwd_reduce(dtype, da, da_pitch, y, y_pitch, scale, shift, mean, rstd, partials, blocks, N, C,
                                   voxels, st));
  HDF_TRY(hdf_launch_in_bwd_finalize(partials, blocks, N, C, voxels, gamma, rstd, k1, ka, kb, dgamma, dbeta, st));
  return hdf_launch_in_bwd_apply(dtype, da, da_pitch, y, y_pitch, scale, shift, mean, rstd, k1, ka, kb, dy, dy_pitch, N,
                                 C, voxels, st);
}
int hdf_op_in_bwd_wgrad(int dtype, const void* da, int64_t da_pitch, const void* y, int64_t y_pitch, const float* scale,
                        const float* shift, const float* mean, const float* rstd, const float* gamma, void* dy,
                        int64_t dy_pitch, float* dgamma, float* dbeta, const void* x, int64_t x_pitch, int Cin,
                        const float* x_scale, const float* x_shift, int x_relu, int N, int Cout, int D, int H, int W,
                        float* dw, float* workspace, void* wgrad_workspace, int64_t wgrad_workspace_bytes,
                        hdf_stream stream) {
  HDF_CHECK_ARG(da && y && scale && shift && mean && rstd && dy && x && dw && workspace && wgrad_workspace,
                "in_bwd_wgrad: null argument");
  const int64_t voxels = (int64_t)D * H * W;
  const int blocks = hdf_in_bwd_blocks(voxels, Cout);
  float* partials = workspace;
  float* k1 = workspace + (int64_t)N * blocks * Cout * 2;
  float* ka = k1 + (int64_t)N * Cout;
  float* kb = ka + (int64_t)N * Cout;
  hipStream_t st = (hipStream_t)stream;
  WgradArgs w{};
  w.sm = da, w.sm_pitch = da_pitch, w.SC = Cout;
  w.lg = x, w.lg_pitch = x_pitch, w.LC = Cin;
  w.N = N;
  w.Ds = w.Dl = D, w.Hs = w.Hl = H, w.Ws = w.Wl = W;
  w.lg_scale = x_scale, w.lg_shift = x_shift, w.lg_relu = x_relu;
  w.ap_y = y, w.ap_y_pitch = y_pitch, w.ap_out = dy, w.ap_out_pitch = dy_pitch;
  w.ap_tab[0] = scale, w.ap_tab[1] = shift, w.ap_tab[2] = mean, w.ap_tab[3] = rstd;
  w.ap_tab[4] = k1, w.ap_tab[5] = ka, w.ap_tab[6] = kb;
  if (!hdf_wgrad_apply_takes(dtype, 1, w)) {
    hdf_set_error("in_bwd_wgrad: the fused kernel does not take this launch (dtype %d, %d -> %d channels, %dx%dx%d)", dtype,
                  Cin, Cout, D, H, W);
    return HDF_ERR_UNSUPPORTED;
  }
  HDF_TRY(hdf_launch_in_bwd_reduce(dtype, da, da_pitch, y, y_pitch, scale, shift, mean, rstd, partials, blocks, N, Cout,
                                   voxels, st));
  HDF_TRY(hdf_launch_in_bwd_finalize(partials, blocks, N, Cout, voxels, gamma, rstd, k1, ka, kb, dgamma, dbeta, st));
  return hdf_launch_wgrad(dtype, 1, w, dw, Cout, Cin, 0, wgrad_workspace, (size_t)wgrad_workspace_bytes, st);
}
int hdf_op_norm_relu_add(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift,
                         const void* skip, int64_t skip_pitch, void* out, int64_t out_pitch, int N, int C,
                         int64_t voxels, hdf_stream stream) {
  return hdf_launch_norm_relu_add(dtype, y, y_pitch, scale, shift, skip, skip_pitch, out, out_pitch, N, C, voxels,
                                  (hipStream_t)stream);
}
int hdf_op_maxpool_fwd(int dtype, const void* in, int64_t in_pitch, void* out, int64_t out_pitch, uint8_t* idx, int N,
                       int C, int Do, int Ho, int Wo, hdf_stream stream) {
  return hdf_launch_maxpool_fwd(dtype, in, in_pitch, out, out_pitch, idx, N, C, Do, Ho, Wo, (hipStream_t)stream);
}
int hdf_op_enc_tail(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift, const void* skip,
                    int64_t skip_pitch, void* ds, int64_t ds_pitch, void* pooled, int64_t pooled_pitch, uint8_t* idx,
                    int N, int C, int Do, int Ho, int Wo, hdf_stream stream) {
  HDF_CHECK_ARG(y && scale && shift && skip && ds && pooled && idx, "enc_tail: null argument");
  return hdf_launch_enc_tail(dtype, y, y_pitch, scale, shift, skip, skip_pitch, ds, ds_pitch, pooled, pooled_pitch, idx, N,
                             C, Do, Ho, Wo, (hipStream_t)stream);
}
int hdf_op_enc_tail_up(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift, const void* low,
                       int64_t low_pitch, const float* lscale, const float* lshift, void* ds, int64_t ds_pitch,
                       void* pooled, int64_t pooled_pitch, uint8_t* idx, int N, int C, int Do, int Ho, int Wo,
                       hdf_stream stream) {
  HDF_CHECK_ARG(y && scale && shift && low && lscale && lshift && ds && pooled && idx, "enc_tail_up: null argument");
  return hdf_launch_enc_tail_up(dtype, y, y_pitch, scale, shift, low, low_pitch, lscale, lshift, ds, ds_pitch, pooled,
                                pooled_pitch, idx, N, C, Do, Ho, Wo, (hipStream_t)stream);
}
int hdf_op_maxpool_bwd(int dtype, const void* dout, int64_t dout_pitch, const uint8_t* idx, void* din,
                       int64_t din_pitch, int N, int C, int Do, int Ho, int Wo, int accumulate, hdf_stream stream) {
  return hdf_launch_maxpool_bwd(dtype, dout, dout_pitch, idx, din, din_pitch, N, C, Do, Ho, Wo, accumulate,
                                (hipStream_t)stream);
}
int hdf_op_maxpool_bwd_in_rows(int C, int Do, int Ho, int Wo) {
  return hdf_maxpool_bwd_in_blocks((int64_t)Do * Ho * Wo, C);
}
int hdf_op_maxpool_bwd_in(int dtype, const void* dout, int64_t dout_pitch, const uint8_t* idx, void* din,
                          int64_t din_pitch, const void* y, int64_t y_pitch, const float* scale, const float* shift,
                          const float* mean, const float* rstd, float* partials, int N, int C, int Do, int Ho, int Wo,
                          hdf_stream stream) {
  HDF_CHECK_ARG(dout && idx && din && y && scale && shift && mean && rstd && partials, "maxpool_bwd_in: null argument");
  return hdf_launch_maxpool_bwd_in(dtype, dout, dout_pitch, idx, din, din_pitch, y, y_pitch, scale, shift, mean, rstd,
                                   partials, N, C, Do, Ho, Wo, (hipStream_t)stream);
}
int hdf_op_upsample_fwd(int dtype, const void* y, int64_t y_pitch, const float* scale, const float* shift, void* out,
                        int64_t out_pitch, int N, int C, int Di, int Hi, int Wi, hdf_stream stream) {
  return hdf_launch_upsample_fwd(dtype, y, y_pitch, scale, shift, out, out_pitch, N, C, Di, Hi, Wi,
                                 (hipStream_t)stream);
}
int hdf_op_upsample_bwd(int dtype, const void* dout, int64_t dout_pitch, void* din, int64_t din_pitch, int N, int C,
                        int Di, int Hi, int Wi, hdf_stream stream) {
  return hdf_launch_upsample_bwd(dtype, dout, dout_pitch, din, din_pitch, N, C, Di, Hi, Wi, (hipStream_t)stream);
}


// ---- transformer / head operator level (HDenseFormer.py:47-145,223-227) -------------------------------------------
namespace {
TfDims op_dims(int M, int B, int N, int DM, int64_t mstride, int training, uint64_t seed) {
  TfDims d;
  d.M = M, d.B = B, d.N = N, d.DM = DM, d.DMF = DM + 128;
  d.mstride = mstride;
  d.training = training ? 1 : 0;
  d.seed = (uint32_t)(seed & 0xffffffffu);
  d.thresh24 = 1u << 23;
  d.keep_scale = 2.0f;
  return d;
}
int op_dims_ok(int M, int B, int N, int DM) {
  HDF_CHECK_ARG(M >= 1 && B >= 1 && N >= 1, "transformer op: M=%d B=%d N=%d", M, B, N);
  HDF_CHECK_ARG(DM >= 32 && DM <= 256 && DM % 32 == 0, "transformer op: token dim %d unsupported (32..256, x32)", DM);
  return HDF_OK;
}
TfLayerP layer_ptrs(float* const* q) {
  return TfLayerP{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11], q[12]};
}
TfLayerSave save_ptrs(float* base, int64_t rows) {
  TfLayerSave s;
  s.h0 = base, s.qkv = base + rows * 32, s.ob = base + rows * 128, s.lse = base + rows * 160;
  s.h1 = base + rows * 168, s.h2 = base + rows * 200;
  return s;
}
}  // namespace

int hdf_op_attention_fwd(const float* qkv, int nseq, int N, float* ob, float* lse, hdf_stream stream) {
  HDF_CHECK_ARG(qkv && ob && lse && nseq >= 1, "attention_fwd: null argument");
  return tf_attention_fwd(N, nseq, qkv, ob, lse, (hipStream_t)stream);
}
int hdf_op_attention_bwd(const float* qkv, const float* ob, const float* lse, const float* d_ob, float* dqkv, int nseq,
                         int N, hdf_stream stream) {
  HDF_CHECK_ARG(qkv && ob && lse && d_ob && dqkv && nseq >= 1, "attention_bwd: null argument");
  return tf_attention_bwd(N, nseq, qkv, ob, lse, d_ob, dqkv, (hipStream_t)stream);
}
int hdf_op_attention_amp_bwd(int dtype, const float* qkv, const float* ob, const float* lse, const float* d_ob,
                             float* dqkv, int nseq, int N, hdf_stream stream) {
  HDF_CHECK_ARG(qkv && ob && lse && d_ob && dqkv && nseq >= 1, "attention_amp_bwd: null argument");
  HDF_CHECK_ARG(dtype == HDF_F32 || dtype == HDF_BF16 || dtype == HDF_F16, "attention_amp_bwd: dtype %d", dtype);
  return tf_attention_bwd(N, nseq, qkv, ob, lse, d_ob, dqkv, (hipStream_t)stream, dtype);
}
int hdf_op_patch_embed_fwd(const float* x, int M, int B, int D, int H, int W, int DM, const float* weight,
                           const float* bias, const float* pos, int64_t mstride, float* F, int training, uint64_t seed,
                           hdf_stream stream) {
  HDF_CHECK_ARG(x && weight && bias && pos && F, "patch_embed_fwd: null argument");
  HDF_CHECK_ARG(D % 16 == 0 && H % 16 == 0 && W % 16 == 0, "patch_embed_fwd: size (%d,%d,%d) not x16", D, H, W);
  const int N = (D / 16) * (H / 16) * (W / 16);
  HDF_TRY(op_dims_ok(M, B, N, DM));
  return tf_patch_embed_fwd(op_dims(M, B, N, DM, mstride, training, seed), x, D, H, W, weight, bias, pos, F,
                            (hipStream_t)stream);
}
int hdf_op_patch_embed_bwd(const float* x, int M, int B, int D, int H, int W, int DM, const float* dF, int64_t mstride,
                           float* dweight, float* dbias, float* dpos, float* scratch, int training, uint64_t seed,
                           hdf_stream stream) {
  HDF_CHECK_ARG(x && dF && dweight && dbias && dpos && scratch, "patch_embed_bwd: null argument");
  const int N = (D / 16) * (H / 16) * (W / 16);
  HDF_TRY(op_dims_ok(M, B, N, DM));
  return tf_patch_embed_bwd(op_dims(M, B, N, DM, mstride, training, seed), x, D, H, W, dF, dweight, dbias, dpos, scratch,
                            (hipStream_t)stream);
}
int hdf_op_dense_layer_fwd(int M, int B, int N, int DM, int block, int layer, const float* const* params13,
                           int64_t mstride, float* F, float* save, int training, uint64_t seed, hdf_stream stream) {
  HDF_CHECK_ARG(params13 && F && save && layer >= 0 && layer < 4, "dense_layer_fwd: bad argument");
  HDF_TRY(op_dims_ok(M, B, N, DM));
  return tf_layer_fwd(op_dims(M, B, N, DM, mstride, training, seed), block, layer,
                      layer_ptrs(const_cast<float* const*>(reinterpret_cast<const float* const*>(params13))), F,
                      save_ptrs(save, (int64_t)M * B * N), (hipStream_t)stream);
}
int hdf_op_dense_layer_bwd(int M, int B, int N, int DM, int block, int layer, const float* const* params13,
                           float* const* grads13, int64_t mstride, const float* F, float* dF, const float* save,
                           float* scratch, int training, uint64_t seed, hdf_stream stream) {
  HDF_CHECK_ARG(params13 && grads13 && F && dF && save && scratch && layer >= 0 && layer < 4,
                "dense_layer_bwd: bad argument");
  HDF_TRY(op_dims_ok(M, B, N, DM));
  return tf_layer_bwd(op_dims(M, B, N, DM, mstride, training, seed), block, layer,
                      layer_ptrs(const_cast<float* const*>(reinterpret_cast<const float* const*>(params13))),
                      layer_ptrs(grads13), F, dF, save_ptrs(const_cast<float*>(save), (int64_t)M * B * N), scratch,
                      (hipStream_t)stream);
}
int hdf_op_block_out_fwd(int M, int B, int N, int DM, int block, const float* const* params4, int64_t mstride,
                         const float* F, float* next_F, void* attnall, int dtype, int training, uint64_t seed,
                         hdf_stream stream) {
  HDF_CHECK_ARG(params4 && F && ((next_F != nullptr) != (attnall != nullptr)), "block_out_fwd: bad argument");
  HDF_TRY(op_dims_ok(M, B, N, DM));
  TfOutP o{const_cast<float*>(params4[0]), const_cast<float*>(params4[1]), const_cast<float*>(params4[2]),
           const_cast<float*>(params4[3])};
  return tf_block_out_fwd(op_dims(M, B, N, DM, mstride, training, seed), block, o, F, next_F, attnall, dtype,
                          (hipStream_t)stream);
}
int hdf_op_block_out_bwd(int M, int B, int N, int DM, int block, const float* const* params4, float* const* grads4,
                         int64_t mstride, const float* F, const float* dF_next, const void* d_attnall, int dtype,
                         float* dF, int training, uint64_t seed, hdf_stream stream) {
  HDF_CHECK_ARG(params4 && grads4 && F && dF && ((dF_next != nullptr) != (d_attnall != nullptr)),
                "block_out_bwd: bad argument");
  HDF_TRY(op_dims_ok(M, B, N, DM));
  TfOutP o{const_cast<float*>(params4[0]), const_cast<float*>(params4[1]), const_cast<float*>(params4[2]),
           const_cast<float*>(params4[3])};
  TfOutP g{grads4[0], grads4[1], grads4[2], grads4[3]};
  return tf_block_out_bwd(op_dims(M, B, N, DM, mstride, training, seed), block, o, g, F, dF_next, d_attnall, dtype, dF,
                          (hipStream_t)stream);
}
int hdf_op_head_fwd(int dtype, const void* in, int64_t in_pitch, const float* in_scale, const float* in_shift,
                    const float* weight, const float* bias, void* logits, int N, int C, int n_cls, int64_t voxels,
                    hdf_stream stream) {
  HDF_CHECK_ARG(in && weight && bias && logits, "head_fwd: null argument");
  return hdf_launch_head_fwd(dtype, in, in_pitch, in_scale, in_shift, weight, bias, logits, N, C, n_cls, voxels,
                             (hipStream_t)stream);
}
int hdf_op_head_bwd(int dtype, const void* dlogits, const void* in, int64_t in_pitch, const float* in_scale,
                    const float* in_shift, const float* weight, void* dx, int64_t dx_pitch, int accumulate_dx,
                    float* dweight, float* dbias, int N, int C, int n_cls, int64_t voxels, hdf_stream stream) {
  HDF_CHECK_ARG(dlogits && in && weight && dx && dweight && dbias, "head_bwd: null argument");
  return hdf_launch_head_bwd(dtype, dlogits, in, in_pitch, in_scale, in_shift, weight, dx, dx_pitch, accumulate_dx,
                             dweight, dbias, N, C, n_cls, voxels, (hipStream_t)stream);
}

}  // extern "C"
