// runtime.cpp -- static execution plan + C ABI (include/mvae_hip.h) of libmvae_hip.so.
//
// mvae_create() turns the MultiscaleVAE constructor arguments into a static plan: the parameter table
// (SURVEY.md appendix A order), a workspace layout for every saved activation, and the op sequence of
// mvae/multiscale_vae.py:73-288.  forward()/backward() walk that plan and enqueue HIP kernels; there is
// no autograd -- every backward step is written out against the forward it inverts.
#include "../../include/mvae_hip.h"
#include "kernels.h"
#include "prof.h"

#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

using namespace mvae;

namespace mvae {      // kernels_bf16.hip
bool launch16_pw(bool transposed, const void* in, const float* w, const float* bias, const float* gate, const void* residual,
                 void* out, int64_t M, int64_t rows_per_image, int K, int N, int act, hipStream_t s, bool out_f32 = false,
                 const float* pivot = nullptr, float* st1 = nullptr, float* st2 = nullptr, int nslots = 1, int64_t slot_stride = 0);
bool launch16_dual(const void* X, const float* W, const void* aux, const float* gate, const void* residual, void* Y,
                   float* dW, float* db, float* dot_out, int64_t M, int64_t rows_per_image, int C, GradSlots sl,
                   hipStream_t s, bool embed_mask = false);
bool launch16_taps(bool transposed, const void* in, const float* w, const float* bias, void* out, const ConvGeom& g,
                   hipStream_t s, const float* w2 = nullptr, const float* bias2 = nullptr, void* out2 = nullptr,
                   bool* chained = nullptr);   // w2 / out2: the following block's conv0 rides along (T-form 32 -> 64)
bool launch16_wgrad(const void* big, const void* small, float* dW, float* db, const ConvGeom& g, GradSlots sl, hipStream_t s,
                    float* db_big = nullptr, bool* db_big_done = nullptr);
}

namespace {

constexpr int kConvBaseFilters = 32;      // multiscale_vae.py:50
constexpr float kDropout = 0.1f;          // multiscale_vae.py:58
constexpr float kSeBnMomentum = 0.99f, kSeBnEps = 1e-3f;     // keras BatchNormalization defaults (layer_blocks.py:448)
constexpr float kDecBnMomentum = 0.999f, kDecBnEps = 1e-4f;  // multiscale_vae.py:420-421
constexpr int kGradSlots = 16;
constexpr int64_t kAlign = 64;            // floats; every tensor / buffer starts on a 256-byte line
constexpr int kChunk = 2048;               // optimiser work item: one block per chunk (small chunks = enough blocks in flight)
constexpr int kSlotMaxElems = 8192;       // tensors up to this size accumulate through the gradient slots

std::string g_create_error;

struct ParamInfo { std::string name; int64_t shape[4]; int ndim; int64_t offset, elems; int reg; };
struct StateInfo { std::string name; int64_t elems, offset; float momentum; int64_t per_image; };

struct MN {
  bool out_f32 = false;                     // bf16 scale: this block's output feeds the decoder BatchNorm and stays float32
  // set by decoder_forward on a decoder's LAST block (bf16 scale, training): the conv2 kernel that writes `out` also sums
  // (out - pivot) and (out - pivot)^2 per channel into the BatchNorm's slot copies (no statistics pass over `out`)
  const float* bn_piv = nullptr;
  float *bn_s1 = nullptr, *bn_s2 = nullptr;
  int bn_slots = 1;
  int64_t bn_stride = 0;
  bool bn_stats_done = false;
  int c = 0, H = 0, W = 0;
  int dg_slots = 1, dg_prefix = 0;          // slot copies of this block's gate gradient, and where they start in Scale::dg
  int64_t w0, b0, wd, bd, sw0, sb0, gam, bet, sw1, sb1, w2, b2;
  int64_t st_mean, st_var;
  float *t0 = nullptr, *t1 = nullptr, *out = nullptr;
  float *gap = nullptr, *s0 = nullptr, *xhat = nullptr, *s1 = nullptr, *ulin = nullptr, *g = nullptr, *invstd = nullptr;
  float* se_part = nullptr;                 // [se_max_blocks][2][c] per-block partial statistics (kernels_se.hip)
};
struct Block {
  bool has_conv = false;
  ConvGeom cg{};          // F-form coordinates (big = high-res side)
  int64_t cw = 0, cb = 0;
  float* cout = nullptr;  // conv / convT output
  float* wsplit = nullptr; // bf16 planes of the k x k weights (kernels_split.hip), refreshed by every forward pass
  MN mn;
};
struct Scale {
  bool bf = false;                          // this scale's wide tensors are bfloat16 (MVAE_ACT_BF16 and every shape covered)
  int dg_total = 0;                         // sum of the blocks' gate-gradient slot copies
  int nmn = 0;                              // MobileNetV3 blocks of this scale (encoder + decoder)
  int H, W, C, z, z_off;
  float *pcur = nullptr, *band = nullptr;
  int64_t cb_w, cb_b;
  float* e0 = nullptr;
  std::vector<Block> enc, dec;
  int fh, fw, fc;
  int64_t K;
  int64_t mu_w, mu_b, lv_w, lv_b, dd_w, dd_b;
  float *mu = nullptr, *lv = nullptr, *zs = nullptr, *d0 = nullptr;
  int dc;
  int64_t bn_g, bn_b, st_bn_mean, st_bn_var, out_w, out_b;
  float *bn_sum, *bn_sqdev, *bn_mean, *bn_invstd, *bn_scale, *bn_shift, *bn_sum_d, *bn_sum_dx;
  float* head_S = nullptr;                  // [head_slots()][2][dc] partial BatchNorm-backward sums (kernels_edge.hip)
  float *y = nullptr, *merged = nullptr, *dy = nullptr;
  int64_t scratch_elems = 0;
  float* scratch[4] = {nullptr, nullptr, nullptr, nullptr};
  bool scratch_used[4] = {false, false, false, false};
  // weight-gradient side stream of this scale: wgrad kernels are off the backward dependency chain, they run
  // concurrently with it; a scratch buffer they still read is guarded by ev_buf until it is recycled
  hipStream_t wstream = nullptr;
  hipEvent_t ev_prod = nullptr, ev_wjoin = nullptr, ev_buf[4] = {nullptr, nullptr, nullptr, nullptr};
  bool buf_pending[4] = {false, false, false, false};
  bool w_used = false;                      // launches on wstream since the last join
  int lru[4] = {0, 0, 0, 0}, tick = 0;
  int cmax = 0;
  float *dg, *dgap, *ds1, *dv, *dz, *dmu, *dlv, *dw_part;
  float* maskbuf = nullptr;                 // float32 scales: ReLU-mask words of t1, two per pixel (k_gemm_dual_s<3> -> k_mn_bwd_s)
  float* d_mid = nullptr;                   // gradient at the encoder output between the two backward phases
};

}  // namespace

struct mvae_handle {
  mvae_config cfg;
  std::string err;
  std::vector<ParamInfo> params;
  std::vector<StateInfo> states;
  std::map<std::string, std::pair<int64_t, int64_t>> tensors;   // name -> (workspace float offset, elems per image)
  std::map<std::string, int> tensor_dtype;                      // name -> MVAE_ACT_* (absent = float32)
  bool phase0_done = false;                                     // mvae_backward_phase(0) ran, phase 1 may follow
  bool kernel_gap = false;                                      // a bf16 launch found no kernel for its shape
  std::vector<Scale> scales;
  int64_t P = 0, S = 0, Z = 0, MET = 0;
  int64_t reduce_split = 0;                 // floats of the leading Dense-weight region of the gradient arena
  int64_t ws_floats = 0;
  std::vector<ChunkDesc> chunks;
  std::vector<ChunkDesc> slot_chunks;       // tensors of at most one chunk: the only ones that receive slot atomics
  ChunkDesc* d_slot_chunks = nullptr;
  int64_t off_slot_chunks = 0;
  std::vector<StateDesc> sdescs;
  // workspace offsets of the fixed tables / buffers
  int64_t off_chunks = 0, off_sdescs = 0, off_norms = 0, off_slots = 0;
  GradSlots gslots;                         // kernels.h: spread small-gradient atomics over kGradSlots arena copies
  // bound memory
  bool bound = false;
  int device = -1;
  float *dp = nullptr, *dr = nullptr, *da = nullptr, *ds = nullptr, *ws = nullptr;
  ChunkDesc* d_chunks = nullptr;
  StateDesc* d_sdescs = nullptr;
  float* d_norms = nullptr;
  float *xin = nullptr, *eps_buf = nullptr, *noise_buf = nullptr, *keep_buf = nullptr, *recon = nullptr,
        *losses = nullptr, *sgn = nullptr, *reg_tmp = nullptr;
  uint64_t* d_seed = nullptr;
  float* d_hp = nullptr;                    // kernels.h HP_*: loss factors, lr, clip, grad_scale (device-resident)
  // what the hyper-parameter block was last set to, and on which stream: a training loop passes the same values step after
  // step, and each update is an eager one-thread launch between two graph replays
  float hp_set[5] = {NAN, NAN, NAN, NAN, NAN};
  // the forward pass leaves seed + 1 in d_seed (k_seed_next, inside the graph): a caller that counts its seeds up step by step --
  // every training loop -- needs no eager launch to set it
  uint64_t seed_dev = 0;
  bool seed_dev_valid = false;
  hipStream_t seed_stream = nullptr;
  hipStream_t hp_stream[2] = {nullptr, nullptr};
  int64_t off_seed = 0, off_hp = 0, off_stamps = 0;
  uint64_t* d_stamps = nullptr;             // MVAE_STAMPS=1: device time stamps at the forks / joins of a step (diagnostic)
  bool stamps = false;
  // concurrency: scale 0 runs on the caller's stream, every other scale on its own side stream (fork/join with
  // events); each ABI call is captured into a hipGraph per argument signature and replayed.
  // wgrad_streams (MVAE_WGRAD_STREAMS): 0 (default) = weight gradients stay on their scale's chain.  2 = the k x k
  // convolution weight gradients of scale 0 leave the chain for its side stream (MFMA-bound kernels, 0.27 ms of the headline's
  // scale-0 chain, a first-level fork the graph capture handles): measured 5.27 against 5.24 ms -- the step is bound by what it
  // moves, not by that chain -- so it stays opt-in.  1 = every weight gradient of every scale:
  // correct in eager mode, but the ~450 extra event calls per step make the host the bottleneck there, and a fork from a
  // stream that itself joined the capture by a fork crashes hipStreamEndCapture (ROCm 7.2), so that mode never captures.
  bool merge_side = false;
  // side streams actually used: scale l runs on side[min(l, side_cap)].  A process has 4 hardware queues (GPU_MAX_HW_QUEUES), and a
  // forked graph's branches are dealt onto them in creation order: with the 7 scales of a 256 x 256 model as 7 branches, scales
  // 0 / 1 / 2 each queued behind one of the small scales' whole chains (in-graph stamps: scale 0 started 0.47 ms after the
  // forward's fork and 0.97 ms after the backward's).  MVAE_SIDE_STREAMS=3 makes exactly 4 branches (scales >= 3 one after the
  // other on the last side stream): measured SLOWER (20.5 against 20.2 ms) -- beside the big scales' kernels the small chains
  // take 2 - 5x as long.  Default: one stream per scale; the ISSUE ORDER (issue_order, set in mvae_create) decides who queues
  // behind whom.
  int side_cap = MVAE_MAX_LEVELS;
  int issue_order[MVAE_MAX_LEVELS] = {};   // the order in which the scales' chains are issued (= captured = dealt onto hardware queues)
  bool multi_stream = true, use_graphs = true;
  int wgrad_streams = 0;
  bool lsb_mask = true;                     // the depthwise backward takes the ReLU mask from the LSB of dt2 (MVAE_LSB_MASK=0: reads t1)
  bool det = false;                         // MVAE_DETERMINISTIC=1 at mvae_create: one slot per block, no split sums (kernels.h)
  int nslots = kGradSlots, stat_slots = kStatSlots;
  hipStream_t side[MVAE_MAX_LEVELS] = {};
  hipEvent_t ev_fork = nullptr, ev_join[MVAE_MAX_LEVELS] = {};
  std::map<std::string, hipGraphExec_t> graphs;
  // MVAE_GRAPH_SEGMENTS=1 (default 0): an ABI call is captured as LINEAR graphs -- what precedes the fork, one graph per scale's
  // chain, what follows the join -- replayed with eager fork / join events (run_captured below; measured, not faster).
  struct SegStep { int op; hipGraphExec_t exec; int stream; hipEvent_t ev; };   // op 0 launch, 1 record, 2 wait; stream 0 = caller's, l = side[l]
  std::map<std::string, std::vector<SegStep>> seg_graphs;
  std::vector<SegStep>* segcap = nullptr;   // the program being recorded (non-null only inside run_captured's capture)
  hipStream_t seg_main = nullptr, seg_open = nullptr;   // the caller's stream of that capture; the stream whose capture is open
  hipError_t seg_err = hipSuccess;
  bool graph_segments = false;
  int eager_fallbacks = 0;                 // calls that wanted a graph and ran eagerly (stream not capturable)
  std::string eager_reason;
  std::vector<hipEvent_t> ev_pool;      // one fresh event per cross-stream edge of a backward pass
  size_t ev_next = 0;
  // last forward
  int last_B = 0, last_train_B = 0;
  bool last_training = false;
  const float* last_x = nullptr;
  const float* last_eps = nullptr;
  // data-parallel exchange bound to RCCL directly (mvae_comm_init): one communicator per handle = per process = per GPU
  void* comm = nullptr;
  int comm_rank = 0, comm_nranks = 1;
};

namespace {

int fail(mvae_handle* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->err = buf; else g_create_error = buf;
  return code;
}

int64_t align_up(int64_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

void same_pad(int n, int k, int s, int* out, int* before) {
  *out = (n + s - 1) / s;
  int total = (*out - 1) * s + k - n;
  if (total < 0) total = 0;
  *before = total / 2;
}

// The Dense weights (mu / log_var / decoder Dense: 95 % of the gradient bytes of C256-nb) occupy the LEADING region of
// the parameter / gradient arena: their gradients are complete half-way through the backward pass (phase 0 of
// mvae_backward_phase), so a data-parallel caller can all-reduce [0, mvae_reduce_split) while phase 1 still runs.
constexpr int64_t kBigElems = 65536;
bool is_big_param(const std::string& name, int64_t elems) {
  auto ends = [&](const char* suf) { const size_t n = strlen(suf); return name.size() >= n && name.compare(name.size() - n, n, suf) == 0; };
  return elems >= kBigElems && (ends(".mu.w") || ends(".log_var.w") || ends(".dense.w"));
}

// ---- plan builder ---------------------------------------------------------------------------
struct Builder {
  mvae_handle* h;
  int64_t pcur = 0, scur = 0, wcur = 0;   // parameter / state / workspace cursors (floats)
  int64_t bcur = 0;                       // cursor of the leading "big" region (Dense weights), see build_plan
  int maxB;

  int64_t param(const std::string& name, std::initializer_list<int64_t> shape, int reg) {
    ParamInfo p;
    p.name = name;
    p.ndim = (int)shape.size();
    p.elems = 1;
    int i = 0;
    for (int k = 0; k < 4; ++k) p.shape[k] = 1;
    for (int64_t d : shape) { p.shape[i++] = d; p.elems *= d; }
    if (is_big_param(name, p.elems)) { p.offset = bcur; bcur = align_up(bcur + p.elems); }
    else { p.offset = pcur; pcur = align_up(pcur + p.elems); }
    p.reg = reg;
    h->params.push_back(p);
    return p.offset;
  }
  int64_t state(const std::string& name, int64_t elems, float momentum, int64_t per_image) {
    StateInfo s{name, elems, scur, momentum, per_image};
    scur = align_up(scur + elems);
    h->states.push_back(s);
    return s.offset;
  }
  // workspace buffer with `per_image` floats per image (batch-scaled) or `fixed` floats
  int64_t ws_alloc(int64_t floats) {
    int64_t o = wcur;
    wcur = align_up(wcur + floats);
    return o;
  }
  int64_t act(const std::string& name, int64_t per_image) {
    int64_t o = ws_alloc(per_image * maxB);
    if (!name.empty()) h->tensors[name] = {o, per_image};
    return o;
  }
  // a WIDE activation tensor: bfloat16 (half the floats) when the scale runs in bf16
  int64_t actw(const std::string& name, int64_t per_image, bool bf) {
    if (!bf) return act(name, per_image);
    int64_t o = ws_alloc((per_image * maxB + 1) / 2);
    if (!name.empty()) { h->tensors[name] = {o, per_image}; h->tensor_dtype[name] = MVAE_ACT_BF16; }
    return o;
  }
};

// offsets are stored as float* relative to a null base during planning and rebased at bind
inline float* as_ptr(int64_t off) { return reinterpret_cast<float*>(static_cast<intptr_t>(off * 4 + 4096)); }
inline float* rebase(float* p, float* base) {
  if (!p) return nullptr;
  intptr_t off = (reinterpret_cast<intptr_t>(p) - 4096) / 4;
  return base + off;
}

// Does every shape of the per-scale VAE at input size H x W have a bf16 kernel (kernels_bf16.hip + the storage-templated
// VALU kernels)?  A scale that does not (the 8x8 / 4x4 tops of a deep pyramid, odd channel counts) stays float32: the
// scales only meet in the 3-channel pyramid / merge tensors, which are float32 either way.
bool scale_bf16_ok(const mvae_config& c, int H, int W) {
  if (c.input_c != 3 && c.input_c != 1) return false;
  auto map_ok = [](int h, int w, int ch) { return (ch == 32 || ch == 64) && ((int64_t)h * w) % 32 == 0 && h % 4 == 0; };
  int ch = kConvBaseFilters, hh = H, ww = W;
  if (!map_ok(hh, ww, ch)) return false;
  auto walk = [&](int n, const int32_t* f, const int32_t* kh, const int32_t* kw, const int32_t* sh, const int32_t* sw, bool dec) {
    for (int i = 0; i < n; ++i) {
      if (sh[i] != 1 || sw[i] != 1 || f[i] != ch) {
        const bool pw = kh[i] == 1 && kw[i] == 1 && sh[i] == 1 && sw[i] == 1;
        const bool pair = (ch == 32 && f[i] == 64) || (ch == 64 && f[i] == 32);
        if (!pair) return false;
        // a k x k layer runs F-form in one direction and T-form in the other, plus the weight-gradient kernel: the
        // launcher predicates of all three (kernels_bf16.hip: launch16_taps / launch16_wgrad) must hold, or the scale
        // stays float32 -- kernel row count, taps per sub-pixel phase, 32-bit byte offsets into both tensors at max_batch
        const int bh = hh, bw = ww, bc = ch;             // the side this walk comes from
        if (dec) { hh *= sh[i]; ww *= sw[i]; }
        else { int o, p; same_pad(hh, kh[i], sh[i], &o, &p); hh = o; same_pad(ww, kw[i], sw[i], &o, &p); ww = o; }
        ch = f[i];
        if (!pw) {
          if (kw[i] != 5 || kh[i] > 8 || kh[i] * kw[i] > 25) return false;
          if (((kh[i] + sh[i] - 1) / sh[i]) * ((kw[i] + sw[i] - 1) / sw[i]) > 9) return false;
          const int64_t lim = 1LL << 31;
          if ((int64_t)c.max_batch * bh * bw * bc * 2 >= lim || (int64_t)c.max_batch * hh * ww * ch * 2 >= lim) return false;
        }
      }
      if (!map_ok(hh, ww, ch)) return false;
    }
    return true;
  };
  if (!walk(c.enc_n, c.enc_filters, c.enc_kh, c.enc_kw, c.enc_sh, c.enc_sw, false)) return false;
  const int64_t K = (int64_t)hh * ww * ch;
  if (K < 256 || (K % 256) != 0) return false;
  if (!walk(c.dec_n, c.dec_filters, c.dec_kh, c.dec_kw, c.dec_sh, c.dec_sw, true)) return false;
  return (ch & (ch - 1)) == 0 && ch >= 4 && ch <= 256;
}

void build_mn(Builder& b, MN& m, const std::string& p, int c, int H, int W, Scale& sc, bool out_f32 = false) {
  m.c = c; m.H = H; m.W = W;
  m.out_f32 = out_f32 && sc.bf;
  m.w0 = b.param(p + ".conv0.w", {1, 1, c, c}, MVAE_REG_L1);  m.b0 = b.param(p + ".conv0.b", {c}, 0);
  m.wd = b.param(p + ".dw.w", {3, 3, c, 1}, MVAE_REG_L1);     m.bd = b.param(p + ".dw.b", {c}, 0);
  m.sw0 = b.param(p + ".se.d0.w", {c, c}, MVAE_REG_L1);       m.sb0 = b.param(p + ".se.d0.b", {c}, 0);
  m.gam = b.param(p + ".se.bn.gamma", {c}, 0);                m.bet = b.param(p + ".se.bn.beta", {c}, 0);
  m.st_mean = b.state(p + ".se.bn.mean", c, kSeBnMomentum, 0);
  m.st_var = b.state(p + ".se.bn.var", c, kSeBnMomentum, 0);
  m.sw1 = b.param(p + ".se.d1.w", {c, c}, MVAE_REG_L1);       m.sb1 = b.param(p + ".se.d1.b", {c}, 0);
  m.w2 = b.param(p + ".conv2.w", {1, 1, c, c}, MVAE_REG_L1);  m.b2 = b.param(p + ".conv2.b", {c}, 0);
  int64_t hwc = (int64_t)H * W * c;
  m.t0 = as_ptr(b.actw(p + ".t0", hwc, sc.bf));
  m.t1 = as_ptr(b.actw(p + ".t1", hwc, sc.bf));
  m.out = as_ptr(b.actw(p + ".out", hwc, sc.bf && !m.out_f32));
  m.gap = as_ptr(b.act(p + ".gap", c));
  m.s0 = as_ptr(b.act(p + ".s0", c));
  m.xhat = as_ptr(b.act(p + ".xhat", c));
  m.s1 = as_ptr(b.act(p + ".s1", c));
  m.ulin = as_ptr(b.act(p + ".ulin", c));
  m.g = as_ptr(b.act(p + ".g", c));
  m.invstd = as_ptr(b.ws_alloc(c));
  m.se_part = as_ptr(b.ws_alloc((int64_t)se_max_blocks(b.maxB) * 2 * c));
  if (hwc > sc.scratch_elems) sc.scratch_elems = hwc;
  if (c > sc.cmax) sc.cmax = c;
  sc.nmn++;
  // gate-gradient slot copies: one per 1024 pixels of a feature map (power of two, at most 64) when the fused
  // squeeze-excite backward (which folds them) covers the shape
  int ds = 1;
  if ((c == 32 || c == 64) && b.maxB <= 4096)
    while (ds < 64 && (int64_t)H * W / ds > 1024) ds *= 2;
  m.dg_slots = ds;
  m.dg_prefix = sc.dg_total;
  sc.dg_total += ds;
}

int build_plan(mvae_handle* h) {
  const mvae_config& c = h->cfg;
  Builder b{h};
  b.maxB = c.max_batch;
  const int L = c.levels, C = c.input_c;
  {   // size of the leading Dense-weight region: K of every scale from the encoder's shape walk
    int64_t D = 0;
    int H0 = c.input_h, W0 = c.input_w;
    for (int s = 0; s < L; ++s) {
      int hh = H0, ww = W0, ch = kConvBaseFilters;
      for (int i = 0; i < c.enc_n; ++i)
        if (c.enc_sh[i] != 1 || c.enc_sw[i] != 1 || c.enc_filters[i] != ch) {
          int o, pd;
          same_pad(hh, c.enc_kh[i], c.enc_sh[i], &o, &pd); hh = o;
          same_pad(ww, c.enc_kw[i], c.enc_sw[i], &o, &pd); ww = o;
          ch = c.enc_filters[i];
        }
      const int64_t kz = (int64_t)hh * ww * ch * c.z_dims[s];
      if (kz >= kBigElems) D += 3 * align_up(kz);
      H0 /= 2; W0 /= 2;
    }
    b.pcur = D;
    h->reduce_split = D;
  }

  // fixed tables first (sizes known after the parameter pass; reserve generously afterwards) -- the
  // tables are appended at the end instead, see below.
  int H = c.input_h, W = c.input_w, zoff = 0;
  h->scales.resize(L);
  for (int s = 0; s < L; ++s) {
    Scale& sc = h->scales[s];
    sc.H = H; sc.W = W; sc.C = C; sc.z = c.z_dims[s]; sc.z_off = zoff;
    zoff += sc.z;
    char e[32], d[32];
    snprintf(e, sizeof(e), "enc%d", s);
    snprintf(d, sizeof(d), "dec%d", s);
    const std::string E(e), D(d);
    int64_t hwC = (int64_t)H * W * C;
    sc.pcur = as_ptr(b.act("pyr" + std::to_string(s), hwC));
    sc.band = (s == L - 1) ? sc.pcur : as_ptr(b.act("band" + std::to_string(s), hwC));
    if (s == L - 1) h->tensors["band" + std::to_string(s)] = h->tensors["pyr" + std::to_string(s)];
    sc.scratch_elems = hwC;
    // ---- encoder (multiscale_vae.py:319-385)
    sc.cb_w = b.param(E + ".conv_base.w", {3, 3, C, kConvBaseFilters}, MVAE_REG_L2);
    sc.cb_b = b.param(E + ".conv_base.b", {kConvBaseFilters}, 0);
    sc.bf = c.act_dtype == MVAE_ACT_BF16 && 2 * c.z_dims[s] <= 32 && scale_bf16_ok(c, H, W);
    sc.e0 = as_ptr(b.actw(E + ".conv_base", (int64_t)H * W * kConvBaseFilters, sc.bf));
    if ((int64_t)H * W * kConvBaseFilters > sc.scratch_elems) sc.scratch_elems = (int64_t)H * W * kConvBaseFilters;
    int ch = kConvBaseFilters, hh = H, ww = W;
    for (int i = 0; i < c.enc_n; ++i) {
      Block blk;
      int f = c.enc_filters[i], kh = c.enc_kh[i], kw = c.enc_kw[i], sh = c.enc_sh[i], sw = c.enc_sw[i];
      std::string bp = E + ".b" + std::to_string(i);
      if (sh != 1 || sw != 1 || f != ch) {                  // layer_blocks.py:946-949
        blk.has_conv = true;
        ConvGeom g{};
        g.IH = hh; g.IW = ww; g.CI = ch; g.CO = f; g.KH = kh; g.KW = kw; g.SH = sh; g.SW = sw;
        same_pad(hh, kh, sh, &g.OH, &g.PT);
        same_pad(ww, kw, sw, &g.OW, &g.PL);
        blk.cg = g;
        blk.cw = b.param(bp + ".conv.w", {kh, kw, ch, f}, MVAE_REG_L1);
        blk.cb = b.param(bp + ".conv.b", {f}, 0);
        hh = g.OH; ww = g.OW; ch = f;
        blk.cout = as_ptr(b.actw(bp + ".conv", (int64_t)hh * ww * ch, sc.bf));
        if (!sc.bf && split_conv_covers(g)) blk.wsplit = as_ptr(b.ws_alloc((split_planes_bytes(g) + 3) / 4));
      }
      build_mn(b, blk.mn, bp + ".mn", f, hh, ww, sc);
      sc.enc.push_back(blk);
    }
    sc.fh = hh; sc.fw = ww; sc.fc = ch;
    sc.K = (int64_t)hh * ww * ch;
    sc.mu_w = b.param(E + ".mu.w", {sc.K, sc.z}, MVAE_REG_L2);      sc.mu_b = b.param(E + ".mu.b", {sc.z}, 0);
    sc.lv_w = b.param(E + ".log_var.w", {sc.K, sc.z}, MVAE_REG_L2); sc.lv_b = b.param(E + ".log_var.b", {sc.z}, 0);
    sc.mu = as_ptr(b.act(E + ".mu", sc.z));
    sc.lv = as_ptr(b.act(E + ".log_var", sc.z));
    sc.zs = as_ptr(b.act(E + ".z", sc.z));
    // ---- decoder (multiscale_vae.py:389-433)
    sc.dd_w = b.param(D + ".dense.w", {sc.z, sc.K}, MVAE_REG_L2);   sc.dd_b = b.param(D + ".dense.b", {sc.K}, 0);
    sc.d0 = as_ptr(b.actw(D + ".dense", sc.K, sc.bf));
    for (int i = 0; i < c.dec_n; ++i) {
      Block blk;
      int f = c.dec_filters[i], kh = c.dec_kh[i], kw = c.dec_kw[i], sh = c.dec_sh[i], sw = c.dec_sw[i];
      std::string bp = D + ".b" + std::to_string(i);
      if (sh != 1 || sw != 1 || f != ch) {                  // layer_blocks.py:950-951, Conv2DTranspose
        blk.has_conv = true;
        ConvGeom g{};
        g.OH = hh; g.OW = ww; g.CO = ch;                    // small side = convT input
        g.IH = hh * sh; g.IW = ww * sw; g.CI = f;           // big side = convT output
        g.KH = kh; g.KW = kw; g.SH = sh; g.SW = sw;
        int oh, ow;
        same_pad(g.IH, kh, sh, &oh, &g.PT);
        same_pad(g.IW, kw, sw, &ow, &g.PL);
        blk.cg = g;
        blk.cw = b.param(bp + ".convT.w", {kh, kw, f, ch}, MVAE_REG_L1);
        blk.cb = b.param(bp + ".convT.b", {f}, 0);
        hh = g.IH; ww = g.IW; ch = f;
        blk.cout = as_ptr(b.actw(bp + ".convT", (int64_t)hh * ww * ch, sc.bf));
        if (!sc.bf && split_conv_covers(g)) blk.wsplit = as_ptr(b.ws_alloc((split_planes_bytes(g) + 3) / 4));
      }
      build_mn(b, blk.mn, bp + ".mn", f, hh, ww, sc, i == c.dec_n - 1);
      sc.dec.push_back(blk);
    }
    if (hh != H || ww != W)
      return fail(nullptr, MVAE_E_INVALID, "decoder of scale %d produces %dx%d but the scale is %dx%d", s, hh, ww, H, W);
    sc.dc = ch;
    sc.bn_g = b.param(D + ".bn.gamma", {ch}, 0);  sc.bn_b = b.param(D + ".bn.beta", {ch}, 0);
    sc.st_bn_mean = b.state(D + ".bn.mean", ch, kDecBnMomentum, 0);
    sc.st_bn_var = b.state(D + ".bn.var", ch, kDecBnMomentum, (int64_t)H * W);
    sc.out_w = b.param(D + ".out.w", {1, 1, ch, C}, MVAE_REG_L2);   sc.out_b = b.param(D + ".out.b", {C}, 0);
    float** small[] = {&sc.bn_mean, &sc.bn_invstd, &sc.bn_scale, &sc.bn_shift, &sc.bn_sum_d, &sc.bn_sum_dx};
    for (float** p : small) *p = as_ptr(b.ws_alloc(ch));
    {   // column-statistic slot copies: [stat_slots][ch] sums directly followed by [stat_slots][ch] squared deviations
      int64_t o = b.ws_alloc((int64_t)2 * h->stat_slots * ch);
      sc.bn_sum = as_ptr(o);
      sc.bn_sqdev = as_ptr(o + (int64_t)h->stat_slots * ch);
    }
    sc.head_S = as_ptr(b.ws_alloc((int64_t)head_slots() * 2 * ch));
    sc.y = as_ptr(b.act(D + ".y", hwC));
    sc.merged = (s == L - 1) ? sc.y : as_ptr(b.act("merged" + std::to_string(s), hwC));
    sc.dy = as_ptr(b.act("", hwC));
    for (int k = 0; k < 4; ++k) sc.scratch[k] = as_ptr(b.actw("", sc.scratch_elems, sc.bf));
    int64_t cm = sc.cmax;
    sc.dg = as_ptr(b.act("", cm * sc.dg_total)); sc.dgap = as_ptr(b.act("", cm));   // dg: [B, cmax] per slot copy
    sc.ds1 = as_ptr(b.act("", cm)); sc.dv = as_ptr(b.act("", cm));
    sc.dw_part = as_ptr(b.ws_alloc((int64_t)kDwMaxBlocks * 10 * cm));
    if (!sc.bf) sc.maskbuf = as_ptr(b.act("", 2 * (int64_t)sc.H * sc.W));
    sc.dz = as_ptr(b.act("", sc.z)); sc.dmu = as_ptr(b.act("", sc.z)); sc.dlv = as_ptr(b.act("", sc.z));
    H /= 2; W /= 2;
  }
  if (b.bcur != h->reduce_split) return fail(nullptr, MVAE_E_INVALID, "internal: Dense-weight region %lld != %lld",
                                             (long long)b.bcur, (long long)h->reduce_split);
  h->P = b.pcur; h->S = b.scur; h->Z = zoff;
  h->MET = align_up(4 + L);
  // global buffers
  int64_t hwC = (int64_t)c.input_h * c.input_w * C;
  h->xin = as_ptr(b.act("xin", hwC));
  h->eps_buf = as_ptr(b.act("eps", h->Z));
  h->noise_buf = as_ptr(b.act("noise", hwC));
  h->keep_buf = as_ptr(b.act("keep_mask", C));
  h->recon = as_ptr(b.act("recon", hwC));
  h->losses = as_ptr(b.act("losses", 3 + L));
  h->sgn = as_ptr(b.act("loss_sgn", 2 * C));
  h->reg_tmp = as_ptr(b.ws_alloc(kAlign));
  // optimiser tables
  for (size_t t = 0; t < h->params.size(); ++t) {
    const ParamInfo& p = h->params[t];
    const int32_t first = (int32_t)h->chunks.size();
    const int32_t count = (int32_t)((p.elems + kChunk - 1) / kChunk);
    for (int64_t o = 0; o < p.elems; o += kChunk) {
      ChunkDesc cd;
      cd.offset = p.offset + o;
      cd.len = (int32_t)((p.elems - o) < kChunk ? (p.elems - o) : kChunk);
      cd.tensor = (int32_t)t; cd.reg = p.reg; cd.pad = 0;
      cd.first = first; cd.count = count;
      h->chunks.push_back(cd);
    }
  }
  // deterministic mode: every gradient a slotted kernel produces goes through the slots (the 5x5 weights too); the big
  // Dense weights are written by kernels that own their outputs
  for (const ChunkDesc& cd : h->chunks)
    if (h->params[cd.tensor].elems <= kSlotMaxElems || (h->det && cd.offset >= h->reduce_split)) h->slot_chunks.push_back(cd);
  for (const StateInfo& s : h->states) {
    StateDesc sd;
    sd.offset = s.offset; sd.len = (int32_t)s.elems; sd.momentum = s.momentum;
    sd.per_image = (float)s.per_image;   // > 0: Bessel correction with n = B * per_image (fused 4-D BN path)
    sd.pad = 0;
    h->sdescs.push_back(sd);
  }
  h->off_chunks = b.ws_alloc((int64_t)(h->chunks.size() * sizeof(ChunkDesc) + 3) / 4);
  h->off_slot_chunks = b.ws_alloc((int64_t)(h->slot_chunks.size() * sizeof(ChunkDesc) + 3) / 4);
  h->off_sdescs = b.ws_alloc((int64_t)(h->sdescs.size() * sizeof(StateDesc) + 3) / 4);
  h->off_norms = b.ws_alloc(2 * (int64_t)h->chunks.size());      // one partial ||g||^2 per chunk, then one total per tensor
  h->off_seed = b.ws_alloc(kAlign);
  h->off_hp = b.ws_alloc(kAlign);
  h->off_stamps = b.ws_alloc(4 * kAlign);                    // 128 uint64 time stamps (MVAE_STAMPS=1, mvae_stamps)
  h->off_slots = b.ws_alloc((int64_t)h->nslots * h->P);
  h->ws_floats = b.wcur;
  return MVAE_OK;
}

void rebase_all(mvae_handle* h) {
  float* base = h->ws;
  auto rb = [&](float*& p) { p = rebase(p, base); };
  auto rb_mn = [&](MN& m) {
    rb(m.t0); rb(m.t1); rb(m.out); rb(m.gap); rb(m.s0); rb(m.xhat); rb(m.s1); rb(m.ulin); rb(m.g); rb(m.invstd); rb(m.se_part);
  };
  for (Scale& sc : h->scales) {
    bool alias_band = sc.band == sc.pcur, alias_m = sc.merged == sc.y;
    rb(sc.pcur);
    if (alias_band) sc.band = sc.pcur; else rb(sc.band);
    rb(sc.e0);
    for (Block& b : sc.enc) { rb(b.cout); rb(b.wsplit); rb_mn(b.mn); }
    for (Block& b : sc.dec) { rb(b.cout); rb(b.wsplit); rb_mn(b.mn); }
    rb(sc.mu); rb(sc.lv); rb(sc.zs); rb(sc.d0);
    rb(sc.bn_sum); rb(sc.bn_sqdev); rb(sc.bn_mean); rb(sc.bn_invstd); rb(sc.bn_scale); rb(sc.bn_shift);
    rb(sc.bn_sum_d); rb(sc.bn_sum_dx); rb(sc.head_S);
    rb(sc.y);
    if (alias_m) sc.merged = sc.y; else rb(sc.merged);
    rb(sc.dy);
    for (int k = 0; k < 4; ++k) rb(sc.scratch[k]);
    rb(sc.dw_part);
    if (sc.maskbuf) rb(sc.maskbuf);
    rb(sc.dg); rb(sc.dgap); rb(sc.ds1); rb(sc.dv); rb(sc.dz); rb(sc.dmu); rb(sc.dlv);
  }
  rb(h->xin);
  h->d_seed = reinterpret_cast<uint64_t*>(base + h->off_seed);
  h->d_hp = base + h->off_hp;
  h->d_stamps = reinterpret_cast<uint64_t*>(base + h->off_stamps);
  rb(h->eps_buf); rb(h->noise_buf); rb(h->keep_buf); rb(h->recon); rb(h->losses); rb(h->sgn); rb(h->reg_tmp);
  h->d_chunks = reinterpret_cast<ChunkDesc*>(base + h->off_chunks);
  h->d_sdescs = reinterpret_cast<StateDesc*>(base + h->off_sdescs);
  h->d_slot_chunks = reinterpret_cast<ChunkDesc*>(base + h->off_slot_chunks);
  h->d_norms = base + h->off_norms;
  h->gslots.base = base + h->off_slots;
  h->gslots.stride = h->P;
  h->gslots.n = h->nslots;
}

// ---- scratch pool (per scale, static order => stable pointers under graph capture) ----------
bool wgrad_async(mvae_handle* h, const Scale& sc, bool heavy) {
  if (!h->multi_stream || profiler().on || h->det) return false;
  return h->wgrad_streams == 1 || (h->wgrad_streams == 2 && heavy && &sc == &h->scales[0]);
}

// least-recently-released free buffer; if a weight-gradient kernel may still be reading it, the chain waits for it
float* acquire(mvae_handle* h, Scale& sc, hipStream_t chain) {
  int best = -1;
  for (int k = 0; k < 4; ++k)
    if (!sc.scratch_used[k] && (best < 0 || sc.lru[k] < sc.lru[best])) best = k;
  if (best < 0) return nullptr;
  sc.scratch_used[best] = true;
  if (sc.buf_pending[best]) {
    (void)hipStreamWaitEvent(chain, sc.ev_buf[best], 0);
    sc.buf_pending[best] = false;
  }
  return sc.scratch[best];
}
void release(Scale& sc, float* p) {
  for (int k = 0; k < 4; ++k)
    if (sc.scratch[k] == p) { sc.scratch_used[k] = false; sc.lru[k] = ++sc.tick; }
}
hipEvent_t fresh_event(mvae_handle* h) {
  if (h->ev_next == h->ev_pool.size()) {
    hipEvent_t e = nullptr;
    (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    h->ev_pool.push_back(e);
  }
  return h->ev_pool[h->ev_next++];
}
// stream for a weight-gradient launch whose inputs the chain has just produced (edge chain -> wstream)
hipStream_t wgrad_begin(mvae_handle* h, Scale& sc, hipStream_t chain, bool heavy = false) {
  if (!wgrad_async(h, sc, heavy)) return chain;
  sc.w_used = true;
  hipEvent_t e = fresh_event(h);
  (void)hipEventRecord(e, chain);
  (void)hipStreamWaitEvent(sc.wstream, e, 0);
  return sc.wstream;
}
// the launches just issued on `w` read pool buffer `p`: it must not be recycled before they finish
void wgrad_reads(mvae_handle* h, Scale& sc, const float* p, hipStream_t w) {
  if (w != sc.wstream) return;
  for (int k = 0; k < 4; ++k)
    if (sc.scratch[k] == p) {
      sc.ev_buf[k] = fresh_event(h);
      (void)hipEventRecord(sc.ev_buf[k], w);
      sc.buf_pending[k] = true;
    }
}
void wgrad_join(mvae_handle* h, Scale& sc, hipStream_t chain) {
  if (!sc.w_used) return;
  sc.w_used = false;
  hipEvent_t e = fresh_event(h);
  (void)hipEventRecord(e, sc.wstream);
  (void)hipStreamWaitEvent(chain, e, 0);
  for (int k = 0; k < 4; ++k) sc.buf_pending[k] = false;
}

ConvGeom geom1x1(int B, int H, int W, int ci, int co) {
  ConvGeom g{};
  g.B = B; g.IH = g.OH = H; g.IW = g.OW = W; g.CI = ci; g.CO = co;
  g.KH = g.KW = g.SH = g.SW = 1; g.PT = g.PL = 0;
  return g;
}

// k x k convolution of a float32 scale: split-bf16 kernel when the layer has weight planes, else the float32-MFMA / generic
// path.  `refresh`: split the weights first (the forward passes; the backward pass reuses the planes of its forward).
void conv_kxk(mvae_handle* h, Block& blk, bool transposed, bool refresh, const float* in, const float* bias, float* out,
              const ConvGeom& g, hipStream_t s) {
  const float* P = h->dp;
  PreOp none{nullptr, nullptr, nullptr};
  if (blk.wsplit) {
    const double nb = (double)g.B * g.IH * g.IW * g.CI, ns = (double)g.B * g.OH * g.OW * g.CO;
    ProfScope ps(transposed ? "k_conv_taps_s<T>" : "k_conv_taps_s<F>", 4.0 * (nb + ns + (double)g.KH * g.KW * g.CI * g.CO),
                 2.0 * ns * g.KH * g.KW * g.CI, s);
    if (refresh) launch_split_weights(P + blk.cw, blk.wsplit, g, s);
    if (launch_conv_taps_split(transposed, in, blk.wsplit, bias, out, g, s)) return;
  }
  if (transposed) launch_conv_t(in, P + blk.cw, bias, nullptr, out, g, s);
  else launch_conv_f(in, P + blk.cw, bias, nullptr, out, g, none, ACT_NONE, s);
}

// ---- MobileNetV3 block (layer_blocks.py:556-648 with squeeze_excite_block :418-462) ----------
// a bf16 launch must find a kernel: there is no silent fallback to another precision
inline void need16(mvae_handle* h, bool ok) { if (!ok) h->kernel_gap = true; }

// chain: the NEXT block of the same shape (no convolution in between), whose conv0 this block's conv2 launch also
// computes (64 channels); chain3 (bf16): the next Block when a 1x1 convolution 64 -> 32 sits between the two and the next
// block is 32 wide -- that convolution and the next conv0 then ride along too.  conv0_done: this block's conv0 was
// computed by the previous block's launch.  Returns 0 = nothing chained, 1 = the next conv0 is done, 2 = the next
// block's convolution and conv0 are done.
// 3 = the next block's conv0 AND its depthwise 3x3 + ReLU + GAP are done (k_mn_fwd_chain_s, kernels_fused_fwd.hip).
// chained_in: what the previous block's call returned.
int mn_forward(mvae_handle* h, MN& m, const float* x, int B, bool training, hipStream_t s, bool bf, MN* chain = nullptr,
               int chained_in = 0, Block* chain3 = nullptr, bool chain3_transposed = false) {
  const bool conv0_done = chained_in != 0;
  bool dw_first = false;                              // this call's FIRST-form launch did conv0 and the depthwise stage
  const float* P = h->dp;
  float* stats = h->dr + h->P;
  const int c = m.c;
  const int64_t M = (int64_t)B * m.H * m.W, HW = (int64_t)m.H * m.W;
  ConvGeom g = geom1x1(B, m.H, m.W, c, c);
  PreOp none{nullptr, nullptr, nullptr};
  if (bf) {
    if (!conv0_done) need16(h, launch16_pw(false, x, P + m.w0, P + m.b0, nullptr, nullptr, m.t0, M, HW, c, c, ACT_RELU, s));
  } else if (!conv0_done && mn_fwd_first_split_on() && mn_fwd_chain_split_kernel(B, m.H, m.W, c) && [&] {
               // conv0 and the depthwise stage in one pass: x -> t0, t1, gap (kernels_fused_fwd.hip, FIRST form)
               ProfScope ps("k_mn_fwd_first_s", 12.0 * B * m.H * m.W * c, (2.0 * c + 20.0) * B * m.H * m.W * c, s);
               return launch_mn_fwd_first_split(x, P + m.w0, P + m.b0, P + m.wd, P + m.bd, m.t0, m.t1, m.gap, B, m.H, m.W, c, s);
             }()) {
    dw_first = true;
  } else if (!conv0_done) {
    bool tiled0;
    {
      ProfScope ps("k_conv0_tile", 8.0 * B * m.H * m.W * c, 2.0 * B * m.H * m.W * c * c, s);
      tiled0 = launch_conv0_tile(x, P + m.w0, P + m.b0, nullptr, nullptr, m.t0, M, HW, c, s);
    }
    if (!tiled0) launch_conv_f(x, P + m.w0, P + m.b0, nullptr, m.t0, g, none, ACT_RELU, s);
  }
  const bool dw_done = chained_in == 3 || dw_first;
  bool fused_dw = dw_done;
  if (!dw_done) {
    ProfScope ps(dw_uses_img(false, false, B, m.H, m.W, c) ? "k_dw_fwd_img" : "k_dw_fwd_ring<true>", (bf ? 4.0 : 8.0) * B * m.H * m.W * c,
                 20.0 * B * m.H * m.W * c, s);
    fused_dw = launch_dw_fwd_gap(m.t0, P + m.wd, P + m.bd, m.t1, m.gap, B, m.H, m.W, c, s, bf);
  }
  if (bf) need16(h, fused_dw);
  if (!fused_dw) {
    launch_dw_fwd(m.t0, P + m.wd, P + m.bd, m.t1, B, m.H, m.W, c, s);
    launch_spatial_sum(m.t1, m.gap, B, (int64_t)m.H * m.W, c, 1.0f / (float)(m.H * m.W), s);
  }
  if (!launch_se_forward(m.gap, P + m.sw0, P + m.sb0, P + m.gam, P + m.bet, h->ds + m.st_mean, h->ds + m.st_var,
                         P + m.sw1, P + m.sb1, m.s0, m.xhat, m.invstd, m.ulin, m.g, stats + m.st_mean,
                         stats + m.st_var, m.se_part, B, c, kSeBnEps, training ? 1 : 0, s)) {
    launch_gemm_nn(m.gap, P + m.sw0, P + m.sb0, m.s0, nullptr, B, c, c, ACT_RELU, s);
    launch_bn1d_fwd(m.s0, P + m.gam, P + m.bet, h->ds + m.st_mean, h->ds + m.st_var, m.xhat, m.invstd, m.s1,
                    stats + m.st_mean, stats + m.st_var, B, c, kSeBnEps, training ? 1 : 0, s);
    launch_gemm_nn(m.s1, P + m.sw1, P + m.sb1, m.g, m.ulin, B, c, c, ACT_HSIG, s);
  }
  if (bf) {
    if (chain && !m.out_f32 && chain->c == c && chain->H == m.H && chain->W == m.W &&
        launch16_pw_chain(m.t1, P + m.w2, P + m.b2, m.g, x, m.out, nullptr, nullptr, nullptr, false, P + chain->w0,
                          P + chain->b0, chain->t0, M, HW, c, s))
      return 1;
    if (chain3 && !m.out_f32 && c == 64 && chain3->mn.c == 32 && chain3->mn.H == m.H && chain3->mn.W == m.W &&
        launch16_pw_chain(m.t1, P + m.w2, P + m.b2, m.g, x, m.out, P + chain3->cw, P + chain3->cb, chain3->cout,
                          chain3_transposed, P + chain3->mn.w0, P + chain3->mn.b0, chain3->mn.t0, M, HW, c, s))
      return 2;
    const bool st = m.out_f32 && m.bn_s1 != nullptr;
    need16(h, launch16_pw(false, m.t1, P + m.w2, P + m.b2, m.g, x, m.out, M, HW, c, c, ACT_NONE, s, m.out_f32, st ? m.bn_piv : nullptr,
                          st ? m.bn_s1 : nullptr, st ? m.bn_s2 : nullptr, m.bn_slots, m.bn_stride));
    m.bn_stats_done = st;
    return 0;
  }
  if (chain && chain->c == c && chain->H == m.H && chain->W == m.W && mn_fwd_chain_split_kernel(B, m.H, m.W, c)) {
    // conv2 of this block, conv0 and the depthwise stage of the next one in one pass (t0' is not read back)
    ProfScope ps("k_mn_fwd_chain_s", 20.0 * B * m.H * m.W * c, (4.0 * c + 20.0) * B * m.H * m.W * c, s);
    if (launch_mn_fwd_chain_split(m.t1, m.g, x, P + m.w2, P + m.b2, P + chain->w0, P + chain->b0, P + chain->wd, P + chain->bd,
                                  m.out, chain->t0, chain->t1, chain->gap, B, m.H, m.W, c, s))
      return 3;
  }
  if (chain && chain->c == c && chain->H == m.H && chain->W == m.W) {
    ProfScope ps("k_conv2_chain", 16.0 * B * m.H * m.W * c, 4.0 * B * m.H * m.W * c * c, s);
    if (launch_conv2_chain(m.t1, P + m.w2, P + m.b2, m.g, x, m.out, nullptr, nullptr, nullptr, false, P + chain->w0,
                           P + chain->b0, chain->t0, M, HW, c, s))
      return 1;
  }
  if (chain3 && c == 64 && chain3->mn.c == 32 && chain3->mn.H == m.H && chain3->mn.W == m.W) {
    ProfScope ps("k_conv2_chain3", (12.0 + 4.0) * B * m.H * m.W * c, (2.0 * c * c + 2.0 * c * 32 + 2.0 * 32 * 32) * B * m.H * m.W, s);
    if (launch_conv2_chain(m.t1, P + m.w2, P + m.b2, m.g, x, m.out, P + chain3->cw, P + chain3->cb, chain3->cout,
                           chain3_transposed, P + chain3->mn.w0, P + chain3->mn.b0, chain3->mn.t0, M, HW, c, s))
      return 2;
  }
  bool tiled2;
  {
    ProfScope ps("k_conv2_tile", 12.0 * B * m.H * m.W * c, 2.0 * B * m.H * m.W * c * c, s);
    tiled2 = launch_conv0_tile(m.t1, P + m.w2, P + m.b2, m.g, x, m.out, M, HW, c, s);
  }
  if (!tiled2) {
    PreOp gate{m.g, nullptr, nullptr};
    launch_conv_f(m.t1, P + m.w2, P + m.b2, x, m.out, g, gate, ACT_NONE, s);
  }
  return 0;
}

// returns the buffer holding d(loss)/d(block input); consumes (releases) `dout` when it is a pool buffer
float* mn_backward(mvae_handle* h, Scale& sc, MN& m, const float* x, float* dout, int B, hipStream_t s) {
  const float* P = h->dp;
  float* G = h->dr;
  const int c = m.c;
  const int64_t dg_stride = (int64_t)B * sc.cmax;
  float* dg = sc.dg + (int64_t)m.dg_prefix * dg_stride;  // zeroed once per scale at the start of its backward chain
  const int64_t HW = (int64_t)m.H * m.W;
  ConvGeom g = geom1x1(B, m.H, m.W, c, c);
  PreOp none{nullptr, nullptr, nullptr};
  float* bufB = acquire(h, sc, s);
  if (sc.bf) {
    // the same chain on bf16 storage: conv2 pair, squeeze-excite backward, depthwise backward (ReLU mask from t1), conv0 pair
    need16(h, launch16_dual(dout, P + m.w2, m.t1, m.g, nullptr, bufB, G + m.w2, G + m.b2, dg, (int64_t)B * HW, HW, c, h->gslots, s,
                            h->lsb_mask));
    need16(h, launch_se_backward(dg, m.ulin, m.xhat, m.invstd, P + m.gam, P + m.bet, m.s0, m.gap, P + m.sw1, P + m.sw0, sc.ds1,
                                 sc.dgap, G + m.sw1, G + m.sb1, G + m.gam, G + m.bet, G + m.sw0, G + m.sb0, m.se_part, B, c,
                                 h->gslots, m.dg_slots, dg_stride, s));
    float* bufC16 = acquire(h, sc, s);
    // depthwise backward and conv0's pair in one pass where the shape allows (dt0 is then never stored)
    if (h->lsb_mask && !dw_uses_img(true, true, B, m.H, m.W, c) &&
        launch16_dw_bwd_conv0(bufB, m.t0, P + m.wd, m.g, sc.dgap, P + m.w0, x, dout, bufC16, G + m.wd, G + m.bd, G + m.w0,
                              G + m.b0, h->gslots, B, m.H, m.W, c, s)) {
      release(sc, bufB);
      release(sc, dout);
      return bufC16;
    }
    {
      ProfScope ps(dw_uses_img(true, h->lsb_mask, B, m.H, m.W, c) ? "k_dw_bwd_img" : (h->lsb_mask ? "k_dw_bwd_ring<true>" : "k_dw_bwd_ring<false>"),
                   (h->lsb_mask ? 6.0 : 8.0) * B * HW * c, 40.0 * B * HW * c, s);
      need16(h, launch_dw_bwd_fused(bufB, m.t1, m.t0, P + m.wd, m.g, sc.dgap, bufC16, G + m.wd, G + m.bd, h->gslots,
                                    h->lsb_mask, B, m.H, m.W, c, s, true));
    }
    need16(h, launch16_dual(bufC16, P + m.w0, x, nullptr, dout, bufB, G + m.w0, G + m.b0, nullptr, (int64_t)B * HW, HW, c,
                            h->gslots, s));
    release(sc, bufC16);
    release(sc, dout);
    return bufB;
  }
  // fully fused form (kernels_fused.hip, k_mn_bwd_s): the conv2 pair leaves no dt2 behind, only the ReLU mask of t1 as bit
  // words; everything behind the squeeze-excite step is one pass that recomputes dt2 from dout
  const bool full_fuse = h->lsb_mask && sc.maskbuf && mn_bwd_split_kernel(B, m.H, m.W, c) &&
                         gemm_dual_split_kernel(true, (int64_t)B * HW, HW, c);
  bool dual2;
  if (full_fuse) {
    ProfScope ps("k_gemm_dual_s<3>", 8.0 * B * HW * c, 4.0 * B * HW * c * c, s);
    dual2 = launch_gemm_dual_stats(dout, P + m.w2, m.t1, m.g, reinterpret_cast<unsigned*>(sc.maskbuf), G + m.w2, G + m.b2, dg,
                                   (int64_t)B * HW, HW, c, h->gslots, m.dg_slots, dg_stride, 0, s);
  } else {
    const char* stag = gemm_dual_split_kernel(true, (int64_t)B * HW, HW, c);
    ProfScope ps(stag ? stag : (c == 64 ? "k_gemm_dual<64, 1>" : "k_gemm_dual<32, 1>"), 12.0 * B * HW * c, 4.0 * B * HW * c * c, s);   // rocprof kernel names
    // dt2 = dout . W2^T ; dW2 += (t1*g)^T dout ; db2 ; dg = sum_hw dt2 * t1     -- one pass over (dout, t1)
    dual2 = launch_gemm_dual_mfma(dout, P + m.w2, m.t1, m.g, nullptr, bufB, G + m.w2, G + m.b2, dg, (int64_t)B * HW, HW,
                                  c, h->gslots, m.dg_slots, dg_stride, s);
  }
  if (dual2 && h->det) {
    // deterministic mode: the gate gradient again as one pass per (image, channel) -- the dual kernel's own dot product is
    // added by several waves per cell.  (dt2 carries the ReLU mask in its LSB: at most one ulp per term.)
    launch_zero(dg, (int64_t)m.dg_slots * dg_stride, s);
    launch_spatial_dot(bufB, m.t1, dg, B, HW, c, s);
  }
  if (!dual2) {
    {
      hipStream_t w = wgrad_begin(h, sc, s);                                           // dout is ready on the chain
      PreOp gate{m.g, nullptr, nullptr};
      launch_conv_wgrad(m.t1, dout, G + m.w2, G + m.b2, g, gate, h->gslots, w);                   // dW2 = (t1*g)^T dout, db2
      wgrad_reads(h, sc, dout, w);
    }
    launch_conv_t_dot(dout, P + m.w2, bufB, m.t1, dg, g, s);        // dt2 = dout . W2^T ; dg = sum_hw dt2 * t1
  }
  // squeeze-excite backward: dg -> (dW1, db1, dgamma, dbeta, dW0, db0) and dgap
  if (!launch_se_backward(dg, m.ulin, m.xhat, m.invstd, P + m.gam, P + m.bet, m.s0, m.gap, P + m.sw1, P + m.sw0, sc.ds1,
                          sc.dgap, G + m.sw1, G + m.sb1, G + m.gam, G + m.bet, G + m.sw0, G + m.sb0, m.se_part, B, c,
                          h->gslots, m.dg_slots, dg_stride, s)) {
    {
      ProfScope ps("k_se_pair", 16.0 * B * c, 4.0 * B * c * c, s);
      // du = dg * hsig'(u):  dW1 += s1^T du, db1 += sum du  and  ds1 = du W1^T   (s1 = gamma*xhat + beta)
      launch_se_pair(m.xhat, dg, P + m.sw1, G + m.sw1, G + m.sb1, sc.ds1, B, c, c, P + m.gam, P + m.bet, m.ulin, s);
    }
    launch_bn1d_bwd(sc.ds1, m.xhat, m.invstd, P + m.gam, m.s0, sc.dv, G + m.gam, G + m.bet, B, c, s);
    {
      ProfScope ps("k_se_pair", 16.0 * B * c, 4.0 * B * c * c, s);
      // dW0 += gap^T dv, db0 += sum dv  and  dgap = dv W0^T
      launch_se_pair(m.gap, sc.dv, P + m.sw0, G + m.sw0, G + m.sb0, sc.dgap, B, c, c, nullptr, nullptr, nullptr, s);
    }
  }
  // through the gate multiply, the global average pool and the depthwise ReLU
  float* bufC = acquire(h, sc, s);
  if (full_fuse && dual2) {
    ProfScope ps("k_mn_bwd_s", 16.0 * B * HW * c, (40.0 + 6.0 * c) * B * HW * c, s);
    if (launch_mn_bwd_split(dout, reinterpret_cast<const unsigned*>(sc.maskbuf), m.t0, P + m.wd, m.g, sc.dgap, P + m.w2,
                            P + m.w0, x, bufC, G + m.wd, G + m.bd, G + m.w0, G + m.b0, h->gslots, B, m.H, m.W, c, s)) {
      release(sc, bufB);
      release(sc, dout);
      return bufC;
    }
    h->kernel_gap = true;                              // unreachable: the coverage tests above are the launchers' own
  }
  if (dual2 && h->lsb_mask && dw_bwd_conv0_split_kernel(B, m.H, m.W, c)) {
    // ... and conv0's backward pair in the same pass: dt0 is never stored (kernels_fused.hip)
    ProfScope ps("k_dw_bwd_conv0_s", 20.0 * B * HW * c, (40.0 + 4.0 * c) * B * HW * c, s);
    if (launch_dw_bwd_conv0_split(bufB, m.t0, P + m.wd, m.g, sc.dgap, P + m.w0, x, dout, bufC, G + m.wd, G + m.bd, G + m.w0,
                                  G + m.b0, h->gslots, B, m.H, m.W, c, s)) {
      release(sc, bufB);
      release(sc, dout);
      return bufC;
    }
  }
  bool fused_dw;
  {
    const bool lsb = dual2 && h->lsb_mask;
    ProfScope ps(dw_uses_img(true, lsb, B, m.H, m.W, c) ? "k_dw_bwd_img" : (lsb ? "k_dw_bwd_ring<true>" : "k_dw_bwd_ring<false>"),
                 (lsb ? 12.0 : 16.0) * B * HW * c,
                 40.0 * B * HW * c, s);            // 3 passes when t1 is not read (mask in the LSB of dt2)
    fused_dw = launch_dw_bwd_fused(bufB, m.t1, m.t0, P + m.wd, m.g, sc.dgap, bufC, G + m.wd, G + m.bd, h->gslots,
                                   dual2 && h->lsb_mask, B, m.H, m.W, c, s);
  }
  if (!fused_dw) {
    launch_mn_dt1pre(bufB, m.t1, m.g, sc.dgap, B, HW, c, 1.0f / (float)HW, s);
    launch_dw_wgrad(m.t0, bufB, G + m.wd, G + m.bd, B, m.H, m.W, c, s);
    launch_dw_bwd_data(bufB, P + m.wd, m.t0, bufC, B, m.H, m.W, c, s);                 // dt0pre
  }
  bool dual0;
  {
    const char* stag = gemm_dual_split_kernel(false, (int64_t)B * HW, HW, c);
    ProfScope ps(stag ? stag : (c == 64 ? "k_gemm_dual<64, 2>" : "k_gemm_dual<32, 2>"), 16.0 * B * HW * c, 4.0 * B * HW * c * c, s);
    // da = dt0pre . W0^T + dout ; dW0 += a^T dt0pre ; db0     -- one pass over (dt0pre, a, dout)
    dual0 = launch_gemm_dual_mfma(bufC, P + m.w0, x, nullptr, dout, bufB, G + m.w0, G + m.b0, nullptr, (int64_t)B * HW, HW,
                                  c, h->gslots, 1, 0, s);
  }
  if (!dual0) {
    {
      hipStream_t w = wgrad_begin(h, sc, s);                                           // dt0pre is ready on the chain
      launch_conv_wgrad(x, bufC, G + m.w0, G + m.b0, g, none, h->gslots, w);
      wgrad_reads(h, sc, bufC, w);
    }
    launch_conv_t(bufC, P + m.w0, nullptr, dout, bufB, g, s);                          // da = dt0pre.W0^T + dout
  }
  release(sc, bufC);
  release(sc, dout);
  return bufB;
}

void decoder_forward(mvae_handle* h, Scale& sc, int B, bool training, hipStream_t s) {
  const float* P = h->dp;
  float* stats = h->dr + h->P;
  bool fused_dd;
  {
    ProfScope ps("dense_expand", 4.0 * (B * (double)sc.K + sc.K * sc.z), 2.0 * B * sc.K * sc.z, s);
    fused_dd = launch_dense_expand(sc.zs, P + sc.dd_w, P + sc.dd_b, sc.d0, B, sc.z, (int)sc.K, s, sc.bf);
  }
  if (sc.bf) need16(h, fused_dd);
  if (!fused_dd) launch_gemm_nn(sc.zs, P + sc.dd_w, P + sc.dd_b, sc.d0, nullptr, B, sc.z, (int)sc.K, ACT_NONE, s);
  const float* x = sc.d0;
  int chained = 0;
  // batch statistics of the decoder BatchNorm (multiscale_vae.py:420-421), one pass about a pivot: on a bf16 scale the last
  // block's conv2 kernel accumulates them while it writes its output (pivot = the previous step's batch mean, zero at bind:
  // MVAE_BN_FUSED=0 switches that off); otherwise k_colstat4<2> reads the output once (pivot = its row 0; MVAE_BN_ONEPASS=0: the
  // two-pass form, sums then squared deviations).  Slot copies are zeroed here, before the producer runs.
  static const bool onepass = [] { const char* e = getenv("MVAE_BN_ONEPASS"); return e ? atoi(e) != 0 : true; }();
  static const bool bn_fused = [] { const char* e = getenv("MVAE_BN_FUSED"); return e ? atoi(e) != 0 : true; }();
  if (training) launch_zero(sc.bn_sum, (int64_t)2 * h->stat_slots * sc.dc, s);   // bn_sum and bn_sqdev are adjacent
  {
    MN& last = sc.dec.back().mn;
    const bool fuse = training && onepass && bn_fused && sc.bf && last.out_f32 && !h->det;
    last.bn_piv = fuse ? sc.bn_mean : nullptr;
    last.bn_s1 = fuse ? sc.bn_sum : nullptr;
    last.bn_s2 = fuse ? sc.bn_sqdev : nullptr;
    last.bn_slots = h->stat_slots;
    last.bn_stride = sc.dc;
    last.bn_stats_done = false;
  }
  for (Block& blk : sc.dec) {
    if (blk.has_conv && chained == 2) {
      x = blk.cout;                                   // computed by the previous block's conv2 launch
    } else if (blk.has_conv) {
      ConvGeom g = blk.cg; g.B = B;
      if (sc.bf) {
        if (g.KH * g.KW == 1) need16(h, launch16_pw(true, x, P + blk.cw, P + blk.cb, nullptr, nullptr, blk.cout,
                                                    (int64_t)B * g.IH * g.IW, (int64_t)g.IH * g.IW, g.CO, g.CI, ACT_NONE, s));
        else {
          bool c0 = false;                            // the block's conv0 from the chunks the convT has just stored
          const bool want = blk.mn.c == 64 && g.CI == 64;
          need16(h, launch16_taps(true, x, P + blk.cw, P + blk.cb, blk.cout, g, s, want ? P + blk.mn.w0 : nullptr,
                                  want ? P + blk.mn.b0 : nullptr, want ? blk.mn.t0 : nullptr, &c0));
          if (c0) chained = 1;
        }
      } else {
        conv_kxk(h, blk, true, true, x, P + blk.cb, blk.cout, g, s);
      }
      x = blk.cout;
    }
    Block* nb = &blk != &sc.dec.back() ? &blk + 1 : nullptr;
    MN* nextmn = (nb && !nb->has_conv) ? &nb->mn : nullptr;
    Block* next3 = (nb && nb->has_conv && nb->cg.KH * nb->cg.KW == 1 && nb->cg.SH == 1 && nb->cg.SW == 1 &&
                    nb->cg.CO == 64 && nb->cg.CI == 32) ? nb : nullptr;          // convT: CO = its input, CI = its output
    chained = mn_forward(h, blk.mn, x, B, training, s, sc.bf, nextmn, chained, next3, true);
    x = blk.mn.out;
  }
  const int64_t M = (int64_t)B * sc.H * sc.W;
  const float* pivot = nullptr;
  if (training && sc.dec.back().mn.bn_stats_done) {
    pivot = sc.bn_mean;                                                      // (the finalize kernel reads it before it stores the mean)
  } else if (training) {
    // (x = the last block's output: float32 storage in either mode, see MN::out_f32)
    if (onepass && launch_colstat_opt(2, x, nullptr, 0, 0.f, sc.bn_sum, h->stat_slots, sc.dc, M, sc.dc, s, false, sc.bn_sqdev)) {
      pivot = x;                                                             // row 0 of x: the kernel's pivot
    } else {
      const bool cs0 = launch_colstat_opt(0, x, nullptr, 0, 0.f, sc.bn_sum, h->stat_slots, sc.dc, M, sc.dc, s, false);
      if (sc.bf) need16(h, cs0);
      if (!cs0) launch_colsum(x, sc.bn_sum, M, sc.dc, s);
      if (!launch_colstat_opt(1, x, sc.bn_sum, h->stat_slots, 1.0f / (float)M, sc.bn_sqdev, h->stat_slots, sc.dc, M, sc.dc, s, false)) {
        launch_bn2d_mean(sc.bn_sum, nullptr, sc.bn_mean, M, sc.dc, 1, s);
        launch_colsqdev(x, sc.bn_mean, sc.bn_sqdev, M, sc.dc, s);
      }
    }
  }
  launch_bn2d_finalize(sc.bn_sum, sc.bn_sqdev, P + sc.bn_g, P + sc.bn_b, h->ds + sc.st_bn_mean, h->ds + sc.st_bn_var,
                       sc.bn_mean, sc.bn_invstd, sc.bn_scale, sc.bn_shift, stats + sc.st_bn_mean,
                       stats + sc.st_bn_var, M, sc.dc, kDecBnEps, training ? 1 : 0, h->stat_slots, s, pivot);
  ProfScope ps("head_fwd", 4.0 * M * (sc.dc + sc.C), 2.0 * M * sc.dc * sc.C, s);
  const bool hf = launch_head_fwd(x, sc.bn_scale, sc.bn_shift, P + sc.out_w, P + sc.out_b, sc.y, M, sc.dc, sc.C, s, false);
  if (sc.bf) need16(h, hf);
  if (!hf) {
    ConvGeom g = geom1x1(B, sc.H, sc.W, sc.dc, sc.C);
    PreOp bn{nullptr, sc.bn_scale, sc.bn_shift};
    launch_conv_f(x, P + sc.out_w, P + sc.out_b, nullptr, sc.y, g, bn, ACT_NONE, s);
  }
}

void merge_forward(mvae_handle* h, int B, float* recon, hipStream_t s) {
  const mvae_config& c = h->cfg;
  const int L = c.levels;
  for (int i = L - 2; i >= 0; --i) {
    Scale& sc = h->scales[i];
    launch_upsample_add(h->scales[i + 1].merged, sc.y, sc.merged, i == 0 ? recon : nullptr, B, sc.H, sc.W, sc.C,
                        c.min_value, c.max_value, s);
  }
}

// ---- fork / join of the per-scale chains over the side streams ------------------------------
// the instrumented pass runs the scales one after another (isolated kernel durations) unless MVAE_PROF_MULTI is set
static bool serial_scales(mvae_handle* h) {
  static const bool prof_multi = getenv("MVAE_PROF_MULTI") != nullptr;
  return !h->multi_stream || (profiler().on && !prof_multi);
}
hipStream_t scale_stream(mvae_handle* h, int scale, hipStream_t main) {
  if (serial_scales(h) || scale == 0) return main;
  if (h->merge_side) return h->side[1];
  return h->side[scale < h->side_cap ? scale : h->side_cap];
}
// TIMING DIAGNOSTIC ONLY (tools/chain_only.py): MVAE_DEBUG_ONLY_SCALE=k launches the kernels of scale k alone -- the
// results are then garbage; it prices one scale's chain without the others beside it.  The release library does not
// contain the switch: it exists only in a build made with MVAE_DEBUG_BUILD=1 (_build.py adds -DMVAE_DEBUG_BUILD), and
// mvae_debug_build() tells a caller (bench.py, the tests) which of the two it has loaded.
#ifdef MVAE_DEBUG_BUILD
bool debug_skip_scale(int si) {
  static const int only = [] { const char* e = getenv("MVAE_DEBUG_ONLY_SCALE"); return e ? atoi(e) : -1; }();
  return only >= 0 && si != only;
}
#else
constexpr bool debug_skip_scale(int) { return false; }
#endif
// Order in which the scales' chains are issued (and captured).  Default: the small scales first, scale 0 -- the long pole -- last
// on the caller's stream.  MVAE_SCALE_ORDER=1 issues scale 0 first: measured 4.90 against 4.84 ms (C32-nb) and 21.95 against
// 20.95 ms (C256-nb bf16) -- the big kernels then own the chip from the start and the latency-bound small chains run out
// alone at the end.  (A rocprofv3 kernel trace suggests the opposite -- scale 0's queue idle for 0.4 - 0.6 ms behind each fork --
// because the tracer serialises the dispatches of a graph replay on the host; the in-graph time stamps of MVAE_STAMPS=1 show
// the unperturbed timeline, tools/stamps.py.)
static bool scale0_first(int pass) {          // pass 1 = forward, 2 = backward; MVAE_SCALE_ORDER is a bit mask of the passes
  static const int v = [] { const char* e = getenv("MVAE_SCALE_ORDER"); return e ? atoi(e) : 0; }();
  return (v & pass) != 0;
}
// MVAE_STAMPS=1 (read at mvae_create): one-thread kernels that store the constant 100 MHz device clock at the forks, joins and
// chain ends of a step, INSIDE the replayed graphs -- an unperturbed timeline (rocprofv3's kernel trace serialises the
// dispatches of a graph replay on the host: under it the three scales' chains start 0.4 ms apart, without it they do not).
// ids: 0 forward start, 1 fork, 10+i / 20+i scale i forward begin / end, 2 joined, 3 forward end; 4 backward start, 5 fork,
// 30+i / 40+i scale i backward begin / end, 6 joined, 7 backward end; 8 / 9 apply start / end.
void stamp(mvae_handle* h, int id, hipStream_t s) {
  if (h->stamps && h->d_stamps && id >= 0 && id < 128) launch_stamp(h->d_stamps + id, s);
}
// ---- segmented capture (MVAE_GRAPH_SEGMENTS): see run_captured ----
static int seg_stream_index(mvae_handle* h, hipStream_t s) {
  if (s == h->seg_main) return 0;
  for (int l = 1; l < h->cfg.levels; ++l) if (h->side[l] == s) return l;
  return -1;
}
static void seg_begin(mvae_handle* h, hipStream_t s) {
  if (!h->segcap || h->seg_open || h->seg_err != hipSuccess) return;
  h->seg_err = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  if (h->seg_err == hipSuccess) h->seg_open = s;
}
static void seg_end(mvae_handle* h, hipStream_t s) {
  if (!h->segcap || h->seg_open != s) return;
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(s, &graph);
  h->seg_open = nullptr;
  size_t nodes = 0;
  if (e == hipSuccess) e = hipGraphGetNodes(graph, nullptr, &nodes);
  if (e == hipSuccess && nodes > 0) {
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e == hipSuccess) h->segcap->push_back({0, exec, seg_stream_index(h, s), nullptr});
  }
  if (graph) (void)hipGraphDestroy(graph);
  if (e != hipSuccess && h->seg_err == hipSuccess) h->seg_err = e;
}
// a scale's chain starts / ends on stream `ss` (between fork_scales and join_scales)
void chain_begin(mvae_handle* h, hipStream_t ss) {
  if (h->segcap && !serial_scales(h)) seg_begin(h, ss);
}
void chain_end(mvae_handle* h, hipStream_t ss) {
  if (h->segcap && !serial_scales(h)) seg_end(h, ss);
}
void fork_scales(mvae_handle* h, hipStream_t main) {
  if (serial_scales(h)) return;
  if (h->segcap) {                          // the graph of what came before ends here; the fork is replayed eagerly
    seg_end(h, main);
    h->segcap->push_back({1, nullptr, 0, h->ev_fork});
    for (int l = 1; l < h->cfg.levels && l <= h->side_cap; ++l) h->segcap->push_back({2, nullptr, l, h->ev_fork});
    return;
  }
  (void)hipEventRecord(h->ev_fork, main);
  for (int l = 1; l < h->cfg.levels && l <= h->side_cap; ++l) (void)hipStreamWaitEvent(h->side[l], h->ev_fork, 0);
}
void join_scales(mvae_handle* h, hipStream_t main) {
  if (serial_scales(h)) return;
  if (h->segcap) {
    for (int l = 1; l < h->cfg.levels && l <= h->side_cap; ++l) {
      h->segcap->push_back({1, nullptr, l, h->ev_join[l]});
      h->segcap->push_back({2, nullptr, 0, h->ev_join[l]});
    }
    seg_begin(h, main);                     // what follows the join: one more linear graph on the caller's stream
    return;
  }
  for (int l = 1; l < h->cfg.levels && l <= h->side_cap; ++l) {
    (void)hipEventRecord(h->ev_join[l], h->side[l]);
    (void)hipStreamWaitEvent(main, h->ev_join[l], 0);
  }
}

// ---- hipGraph cache: capture the launch sequence of one ABI call once per argument signature, then replay ----
// Two shapes of capture.  (a) default: ONE graph with the scales' chains as forked branches; hipGraphLaunch enqueues it node by
// node in creation order at ~2.3 us of host time each (tools/graph_launch_cost.hip: ~0.9 ms per pass for the ~380 nodes of a
// C32-nb forward).  (b) MVAE_GRAPH_SEGMENTS=1: linear graphs (before the fork | one per chain | after the join), which ROCm
// replays from pre-built AQL packets at 0.03 - 0.3 us per node, with the fork and the join as eager events between them.
// (b) cuts the host cost of a step from ~2 ms to ~0.3 ms and does NOT make the step faster (batch 128: 2.19 against 2.12 ms,
// batch 512: 5.02 against 4.91, medians of one box): a training loop never waits for the host -- it runs up to a full AQL ring
// ahead of the device, so the packets of a step are in their queues long before the device reaches them -- and the caller's
// stream and the side streams then have to sit on DIFFERENT hardware queues, which with the runtime's default of 4 queues per
// process they do not (scale 0 serialises behind scale 1: 3.0 ms at batch 128 unless GPU_MAX_HW_QUEUES=8), whereas the forked
// graph's branches are placed by the runtime.  (Host time measured around the ABI calls of a free-running loop is NOT the launch
// cost: once the ring is full every launch waits for the device, and the host appears to take exactly one device step per step.)
static void seg_destroy(std::vector<mvae_handle::SegStep>& prog) {
  for (auto& st : prog) if (st.op == 0 && st.exec) (void)hipGraphExecDestroy(st.exec);
  prog.clear();
}
static int run_segments(mvae_handle* h, const std::string& key, hipStream_t s, const std::function<void(hipStream_t)>& body) {
  auto it = h->seg_graphs.find(key);
  if (it == h->seg_graphs.end()) {
    std::vector<mvae_handle::SegStep> prog;
    h->segcap = &prog; h->seg_main = s; h->seg_open = nullptr; h->seg_err = hipSuccess;
    seg_begin(h, s);
    if (h->seg_err != hipSuccess) {           // e.g. the legacy default stream cannot be captured: run eagerly, and say so
      (void)hipGetLastError();
      h->segcap = nullptr;
      ++h->eager_fallbacks;
      h->eager_reason = hipGetErrorString(h->seg_err);
      body(s);
      return MVAE_OK;
    }
    body(s);
    if (h->seg_open) seg_end(h, h->seg_open);
    h->segcap = nullptr;
    if (h->kernel_gap) {                      // a launch sequence with a missing kernel is never cached (nor replayed)
      seg_destroy(prog);
      return fail(h, MVAE_E_INVALID, "%s: a bfloat16 launch found no kernel for its shape", key.c_str());
    }
    if (h->seg_err != hipSuccess) {
      seg_destroy(prog);
      (void)hipGetLastError();
      return fail(h, MVAE_E_HIP, "graph capture (%s): %s", key.c_str(), hipGetErrorString(h->seg_err));
    }
    it = h->seg_graphs.emplace(key, std::move(prog)).first;
  }
  for (const auto& st : it->second) {
    hipStream_t q = st.stream == 0 ? s : h->side[st.stream];
    hipError_t e = st.op == 0 ? hipGraphLaunch(st.exec, q) : st.op == 1 ? hipEventRecord(st.ev, q) : hipStreamWaitEvent(q, st.ev, 0);
    if (e != hipSuccess) return fail(h, MVAE_E_HIP, "graph replay (%s): %s", key.c_str(), hipGetErrorString(e));
  }
  return MVAE_OK;
}
int run_captured(mvae_handle* h, const std::string& key, hipStream_t s, const std::function<void(hipStream_t)>& body) {
  // a fork from a stream that itself joined the capture by a fork (the weight-gradient side streams) makes
  // hipStreamEndCapture segfault under ROCm 7.2: that mode never captures, whatever the environment says
  const bool eligible = h->use_graphs && h->wgrad_streams != 1 && !profiler().on && s != nullptr;
  if (!eligible) { body(s); return MVAE_OK; }
  if (h->graph_segments && h->wgrad_streams == 0) return run_segments(h, key, s, body);
  auto it = h->graphs.find(key);
  if (it == h->graphs.end()) {
    hipGraph_t graph = nullptr;
    const hipError_t eb = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (eb != hipSuccess) {
      (void)hipGetLastError();
      ++h->eager_fallbacks;           // e.g. the legacy default stream cannot be captured: run eagerly, and say so
      h->eager_reason = hipGetErrorString(eb);   // (mvae_graph_stats)
      body(s);
      return MVAE_OK;
    }
    body(s);
    hipError_t e = hipStreamEndCapture(s, &graph);
    if (h->kernel_gap) {                  // a launch sequence with a missing kernel is never cached (nor replayed)
      if (graph) (void)hipGraphDestroy(graph);
      return fail(h, MVAE_E_INVALID, "%s: a bfloat16 launch found no kernel for its shape", key.c_str());
    }
    hipGraphExec_t exec = nullptr;
    if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(h, MVAE_E_HIP, "graph capture (%s): %s", key.c_str(), hipGetErrorString(e));
    it = h->graphs.emplace(key, exec).first;
  }
  hipError_t e = hipGraphLaunch(it->second, s);
  if (e != hipSuccess) return fail(h, MVAE_E_HIP, "hipGraphLaunch: %s", hipGetErrorString(e));
  return MVAE_OK;
}
std::string fkey(const char* fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return buf;
}

int check_launch(mvae_handle* h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, MVAE_E_HIP, "%s: %s", what, hipGetErrorString(e));
  return MVAE_OK;
}

}  // namespace

// ==================================================================================================
// C ABI
// ==================================================================================================
extern "C" {

int mvae_abi_version(void) { return MVAE_ABI_VERSION; }

int mvae_deterministic(const mvae_handle* h) { return h ? (h->det ? 1 : 0) : MVAE_E_INVALID; }
int mvae_split_conv_status(void) { return split_conv_status(); }
int mvae_split_conv_erratum(void) { return split_conv_erratum_count(); }
int mvae_packed_f32_hazard(int32_t* beside_split, int32_t* beside_bf16) {
  if (beside_split) *beside_split = split_conv_erratum_count();
  if (beside_bf16) *beside_bf16 = k16_erratum_count();
  return MVAE_OK;
}
int mvae_stamps(const mvae_handle* h, uint64_t* out, int32_t n) {
  if (!h || !out || n <= 0 || n > 128) return MVAE_E_INVALID;
  if (!h->bound || !h->stamps) return MVAE_E_STATE;
  if (hipDeviceSynchronize() != hipSuccess) return MVAE_E_HIP;
  return hipMemcpy(out, h->d_stamps, sizeof(uint64_t) * n, hipMemcpyDeviceToHost) == hipSuccess ? MVAE_OK : MVAE_E_HIP;
}
int mvae_fused_launch_stats(int32_t* fwd, int32_t* bwd, int32_t* max_images_per_block) {
  int v[3];
  fused_launch_stats(v);
  if (fwd) *fwd = v[0];
  if (bwd) *bwd = v[1];
  if (max_images_per_block) *max_images_per_block = v[2];
  return MVAE_OK;
}

int mvae_debug_build(void) {
#ifdef MVAE_DEBUG_BUILD
  return 1;
#else
  return 0;
#endif
}

const char* mvae_last_error(const mvae_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mvae_create(const mvae_config* cfg, mvae_handle** out) {
  if (!cfg || !out) return fail(nullptr, MVAE_E_INVALID, "null argument");
  *out = nullptr;
  if (cfg->abi_version != MVAE_ABI_VERSION) return fail(nullptr, MVAE_E_INVALID, "ABI version mismatch");
  if (cfg->levels < 2 || cfg->levels > MVAE_MAX_LEVELS)
    return fail(nullptr, MVAE_E_INVALID, "len(z_dims) must be in [2, %d] (the reference merge model needs >= 2 levels)",
                MVAE_MAX_LEVELS);
  for (int i = 0; i < cfg->levels; ++i)
    if (cfg->z_dims[i] <= 0) return fail(nullptr, MVAE_E_INVALID, "z_dims elements should be > 0");
  if (cfg->input_h <= 0 || cfg->input_w <= 0 || cfg->input_c <= 0 || cfg->input_c > 8)
    return fail(nullptr, MVAE_E_INVALID, "input_dims must be positive with at most 8 channels");
  int div = 1 << (cfg->levels - 1);
  if (cfg->input_h % div || cfg->input_w % div)
    return fail(nullptr, MVAE_E_INVALID, "input height/width must be divisible by 2^(levels-1) = %d", div);
  if (cfg->enc_n <= 0 || cfg->enc_n > MVAE_MAX_BLOCKS || cfg->dec_n <= 0 || cfg->dec_n > MVAE_MAX_BLOCKS)
    return fail(nullptr, MVAE_E_INVALID, "encoder/decoder lists must have 1..%d entries", MVAE_MAX_BLOCKS);
  for (int i = 0; i < cfg->enc_n; ++i)
    if (cfg->enc_filters[i] <= 0 || cfg->enc_kh[i] <= 0 || cfg->enc_kw[i] <= 0 || cfg->enc_sh[i] <= 0 ||
        cfg->enc_sw[i] <= 0)
      return fail(nullptr, MVAE_E_INVALID, "Filters should be > 0 (encoder entry %d)", i);
  for (int i = 0; i < cfg->dec_n; ++i)
    if (cfg->dec_filters[i] <= 0 || cfg->dec_kh[i] <= 0 || cfg->dec_kw[i] <= 0 || cfg->dec_sh[i] <= 0 ||
        cfg->dec_sw[i] <= 0)
      return fail(nullptr, MVAE_E_INVALID, "Filters should be > 0 (decoder entry %d)", i);
  if (!(cfg->max_value > cfg->min_value)) return fail(nullptr, MVAE_E_INVALID, "max_value must exceed min_value");
  if (cfg->max_batch <= 0) return fail(nullptr, MVAE_E_INVALID, "max_batch must be > 0");
  if (cfg->act_dtype != MVAE_ACT_F32 && cfg->act_dtype != MVAE_ACT_BF16)
    return fail(nullptr, MVAE_E_INVALID, "act_dtype must be MVAE_ACT_F32 or MVAE_ACT_BF16");
  mvae_handle* h = new mvae_handle();
  h->cfg = *cfg;
  if (const char* v = getenv("MVAE_DETERMINISTIC")) h->det = atoi(v) != 0;
  if (const char* v = getenv("MVAE_STAMPS")) h->stamps = atoi(v) != 0;
  if (h->det) {
    if (cfg->act_dtype != MVAE_ACT_F32) { delete h; return fail(nullptr, MVAE_E_INVALID, "MVAE_DETERMINISTIC=1 supports float32 activations only"); }
    h->nslots = kDetSlots; h->stat_slots = kDetSlots;
  }
  set_det_mode(h->det);
  int rc = build_plan(h);
  if (rc == MVAE_OK && h->det && (int64_t)h->nslots * h->P * 4 > (24LL << 30)) {
    delete h;
    return fail(nullptr, MVAE_E_NOMEM, "MVAE_DETERMINISTIC=1: %d gradient-slot copies of this model exceed 24 GB", kDetSlots);
  }
  if (rc != MVAE_OK) { delete h; return rc; }
  *out = h;
  return MVAE_OK;
}

void mvae_destroy(mvae_handle* h) {
  if (!h) return;
  if (h->bound) {
    (void)hipDeviceSynchronize();
    for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
    for (auto& kv : h->seg_graphs) seg_destroy(kv.second);
    for (int l = 1; l < h->cfg.levels; ++l) {
      if (h->side[l]) (void)hipStreamDestroy(h->side[l]);
      if (h->ev_join[l]) (void)hipEventDestroy(h->ev_join[l]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (Scale& sc : h->scales) {
      if (sc.wstream) (void)hipStreamDestroy(sc.wstream);
    }
    for (hipEvent_t ev : h->ev_pool) (void)hipEventDestroy(ev);
  }
  (void)mvae_comm_destroy(h);
  delete h;
}

int64_t mvae_param_count(const mvae_handle* h) { return h ? (int64_t)h->params.size() : -1; }
int64_t mvae_param_elems(const mvae_handle* h) { return h ? h->P : -1; }
int mvae_param_info(const mvae_handle* h, int64_t i, char* name, int32_t name_cap, int64_t shape[4], int32_t* ndim,
                    int64_t* offset, int32_t* reg) {
  if (!h || i < 0 || i >= (int64_t)h->params.size()) return MVAE_E_INVALID;
  const ParamInfo& p = h->params[i];
  if (name && name_cap > 0) { strncpy(name, p.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (shape) for (int k = 0; k < 4; ++k) shape[k] = p.shape[k];
  if (ndim) *ndim = p.ndim;
  if (offset) *offset = p.offset;
  if (reg) *reg = p.reg;
  return MVAE_OK;
}
int64_t mvae_state_count(const mvae_handle* h) { return h ? (int64_t)h->states.size() : -1; }
int64_t mvae_state_elems(const mvae_handle* h) { return h ? h->S : -1; }
int mvae_state_info(const mvae_handle* h, int64_t i, char* name, int32_t name_cap, int64_t* elems, int64_t* offset) {
  if (!h || i < 0 || i >= (int64_t)h->states.size()) return MVAE_E_INVALID;
  const StateInfo& s = h->states[i];
  if (name && name_cap > 0) { strncpy(name, s.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (elems) *elems = s.elems;
  if (offset) *offset = s.offset;
  return MVAE_OK;
}
int64_t mvae_latent_dim(const mvae_handle* h) { return h ? h->Z : -1; }
int64_t mvae_reduce_elems(const mvae_handle* h) { return h ? h->P + h->S + h->MET : -1; }
int64_t mvae_metrics_offset(const mvae_handle* h) { return h ? h->P + h->S : -1; }
int64_t mvae_workspace_bytes(const mvae_handle* h) { return h ? h->ws_floats * 4 : -1; }

int mvae_bind(mvae_handle* h, int32_t device, float* params, float* reduce_arena, float* accum, float* state,
              void* workspace, int64_t workspace_bytes) {
  if (!h) return MVAE_E_INVALID;
  if (!params || !reduce_arena || !accum || !state || !workspace) return fail(h, MVAE_E_INVALID, "null device pointer");
  if (workspace_bytes < h->ws_floats * 4)
    return fail(h, MVAE_E_NOMEM, "workspace too small: %lld < %lld bytes", (long long)workspace_bytes,
                (long long)(h->ws_floats * 4));
  if (h->bound) return fail(h, MVAE_E_STATE, "handle already bound");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(h, MVAE_E_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
  h->device = device;
  h->dp = params; h->dr = reduce_arena; h->da = accum; h->ds = state; h->ws = static_cast<float*>(workspace);
  rebase_all(h);
  for (Scale& sc : h->scales)                                                 // the first step's pivot of the fused BatchNorm statistics
    if (sc.bn_mean) (void)hipMemset(sc.bn_mean, 0, sizeof(float) * sc.dc);
  h->gslots.gbase = h->dr;
  if (const char* v = getenv("MVAE_GRAD_SLOTS")) { if (!h->det) h->gslots.n = atoi(v) > 0 && atoi(v) <= kGradSlots ? atoi(v) : 0; }
  e = hipMemcpy(h->d_chunks, h->chunks.data(), h->chunks.size() * sizeof(ChunkDesc), hipMemcpyHostToDevice);
  if (e == hipSuccess && !h->slot_chunks.empty())
    e = hipMemcpy(h->d_slot_chunks, h->slot_chunks.data(), h->slot_chunks.size() * sizeof(ChunkDesc), hipMemcpyHostToDevice);
  if (e == hipSuccess && !h->sdescs.empty())
    e = hipMemcpy(h->d_sdescs, h->sdescs.data(), h->sdescs.size() * sizeof(StateDesc), hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail(h, MVAE_E_HIP, "table upload: %s", hipGetErrorString(e));
  // layer_blocks.gaussian_kernel((3,3),(2,2)) (layer_blocks.py:980-1002, multiscale_vae.py:56-57)
  double gk[9], sum = 0;
  for (int a = 0; a < 3; ++a)
    for (int b2 = 0; b2 < 3; ++b2) {
      double x = -2.0 + 2.0 * b2, y = -2.0 + 2.0 * a;
      gk[a * 3 + b2] = std::exp(-(x * x + y * y) / 2.0);
      sum += gk[a * 3 + b2];
    }
  float gf[9];
  for (int i = 0; i < 9; ++i) gf[i] = (float)(gk[i] / sum);
  set_gauss_constants(gf);
  if (const char* v = getenv("MVAE_GRAPHS")) h->use_graphs = atoi(v) != 0;
  if (const char* v = getenv("MVAE_STREAMS")) h->multi_stream = atoi(v) != 0;
  if (const char* v = getenv("MVAE_WGRAD_STREAMS")) h->wgrad_streams = atoi(v);
  if (const char* v = getenv("MVAE_GRAPH_SEGMENTS")) h->graph_segments = atoi(v) != 0;
  if (const char* v = getenv("MVAE_LSB_MASK")) h->lsb_mask = atoi(v) != 0;
  if (const char* v = getenv("MVAE_MERGE_SIDE")) h->merge_side = atoi(v) != 0;   // all scales > 0 on ONE side stream
  {   // issue order of the scales' chains: default the small scales first, scale 0 last; MVAE_ISSUE_ORDER="6,5,4,0,3,1,2" overrides
    const int L = h->cfg.levels;
    for (int i = 0; i < L; ++i) h->issue_order[i] = L - 1 - i;
    // 5 .. 7 scales: a replayed forked graph deals its branches onto the process's 4 hardware queues in creation order (branch i
    // -> queue i mod 4; in-graph stamps, tools/stamps.py with STAMPS_FREE_RUNNING=1), and a chain dealt onto an occupied queue starts
    // when the chain ahead of it has finished.  Small-first order put scale 0 -- the step's critical path -- behind scale 4's
    // whole chain: it started 0.47 ms after the forward's fork and 0.97 ms after the backward's.  Issued FOURTH it has the fourth
    // queue to itself (the fifth to seventh chains queue behind the three smallest ones): C256-nb bf16 20.08 - 20.22 -> 19.59 -
    // 19.73 ms (medians, interleaved; 3,2,1,0,6,5,4 / 5,4,3,0,6,1,2 / 4,5,6,0,1,2,3 measure the same).
    if (L >= 5 && L <= 7) {
      int n = 0;
      for (int k = 0; k < 3; ++k) h->issue_order[n++] = L - 1 - k;
      h->issue_order[n++] = 0;
      for (int sidx = L - 4; sidx >= 1; --sidx) h->issue_order[n++] = sidx;
    }
    if (const char* v = getenv("MVAE_ISSUE_ORDER")) {
      int tmp[MVAE_MAX_LEVELS], n = 0; bool seen[MVAE_MAX_LEVELS] = {}; bool ok = true;
      for (const char* p = v; *p && n < MVAE_MAX_LEVELS; ) {
        char* end = nullptr; const long k = strtol(p, &end, 10);
        if (end == p) break;
        if (k < 0 || k >= L || seen[k]) { ok = false; break; }
        seen[k] = true; tmp[n++] = (int)k; p = (*end == ',') ? end + 1 : end;
      }
      if (ok && n == L) for (int i = 0; i < L; ++i) h->issue_order[i] = tmp[i];
    }
  }
  if (const char* v = getenv("MVAE_SIDE_STREAMS")) { const int n = atoi(v); h->side_cap = n < 1 ? 1 : (n > MVAE_MAX_LEVELS - 1 ? MVAE_MAX_LEVELS - 1 : n); }
  if (h->wgrad_streams == 1) h->use_graphs = false;
  // scale 0 (on the caller's stream) is the long pole of every step and the other scales only fill the gaps it leaves,
  // so low-priority side streams look natural -- but they buy 0.4 % on the first handle of a process and cost up to 50 %
  // on every later one (measured, tools/chain_only.py: a second handle's side streams then share a hardware queue with
  // its main stream and the scales serialise).  Off unless MVAE_STREAM_PRIORITY=1.
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  const bool use_prio = getenv("MVAE_STREAM_PRIORITY") ? atoi(getenv("MVAE_STREAM_PRIORITY")) != 0 : false;
  for (int l = 1; l < h->cfg.levels && e == hipSuccess; ++l) {
    e = use_prio ? hipStreamCreateWithPriority(&h->side[l], hipStreamNonBlocking, prio_lo)
                 : hipStreamCreateWithFlags(&h->side[l], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_join[l], hipEventDisableTiming);
  }
  for (Scale& sc : h->scales) {
    if (e != hipSuccess) break;
    e = hipStreamCreateWithFlags(&sc.wstream, hipStreamNonBlocking);
  }
  // cross-stream edges of one backward pass: pre-created, so that no event is created while a capture is open
  for (int k = 0; h->wgrad_streams && k < (h->wgrad_streams == 1 ? 2048 : 64) && e == hipSuccess; ++k) {
    hipEvent_t ev = nullptr;
    e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) h->ev_pool.push_back(ev);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  {   // a plan with split-bf16 convolutions: once per process, check that this board runs them cleanly (kernels_split.hip)
    bool wants = false;
    for (Scale& sc : h->scales) {
      wants = wants || (!sc.bf && split_conv_status() != 0);      // float32 scales: the 1x1 backward pairs (k_gemm_dual_s)
      for (Block& b : sc.enc) wants = wants || b.wsplit;
      for (Block& b : sc.dec) wants = wants || b.wsplit;
    }
    if (wants && e == hipSuccess) (void)split_selftest();
  }
  if (e != hipSuccess) return fail(h, MVAE_E_HIP, "bind: %s", hipGetErrorString(e));
  h->bound = true;
  return MVAE_OK;
}

int mvae_forward(mvae_handle* h, const mvae_step_io* io, void* stream) {
  if (!h || !io) return MVAE_E_INVALID;
  if (!h->bound) return fail(h, MVAE_E_STATE, "mvae_bind has not been called");
  const mvae_config& c = h->cfg;
  const int B = io->batch, L = c.levels, C = c.input_c;
  if (B <= 0 || B > c.max_batch) return fail(h, MVAE_E_INVALID, "batch %d outside [1, %d]", B, c.max_batch);
  if (!io->x) return fail(h, MVAE_E_INVALID, "x is null");
  hipStream_t s0 = static_cast<hipStream_t>(stream);
  const bool training = io->training != 0;
  const float* P = h->dp;
  h->kernel_gap = false;                   // per call: an earlier failed call must not poison the handle
  set_det_mode(h->det);
  const int64_t hwC = (int64_t)c.input_h * c.input_w * C;
  float* metrics = h->dr + h->P + h->S;
  const mvae_step_io io_c = *io;
  // a replayable call: nothing injected, nothing copied out -> every pointer the kernels see is handle-owned
  const bool replayable = !io->eps && !io->noise && !io->keep_mask && !io->recon && !io->mu && !io->log_var &&
                          !io->z && !io->losses;
  const float* xsrc = io->x;
  if (replayable && h->use_graphs && !profiler().on && s0 != nullptr) {
    // (a caller that wrote the batch straight into the handle's input buffer -- mvae_tensor_lookup("xin"), Engine.input_buffer --
    // passes that pointer and saves this copy: an eager launch between two graph replays costs ~15 us of every step)
    if (io->x != h->xin) (void)hipMemcpyAsync(h->xin, io->x, sizeof(float) * hwC * B, hipMemcpyDeviceToDevice, s0);
    xsrc = h->xin;
  }
  if (!(h->seed_dev_valid && h->seed_dev == io->seed && h->seed_stream == s0)) launch_set_u64(h->d_seed, io->seed, s0);
  h->seed_dev_valid = false;                      // (valid again once this pass -- and its k_seed_next -- is enqueued)
  const float* eps = io->eps ? io->eps : h->eps_buf;

  auto body = [=](hipStream_t s) {
    stamp(h, 0, s);
    const mvae_step_io* io = &io_c;
    // ---- randomness: injected (parity) or Philox on the device (timed runs: all three draws and the zeroed metrics in one launch)
    const float *noise = nullptr, *keep = nullptr;
    if (training && !io->eps && !io->noise && !io->keep_mask) {
      launch_rng_step(h->eps_buf, (int64_t)B * h->Z, c.sample_std, h->noise_buf, hwC * B, h->keep_buf, (int64_t)B * C, kDropout,
                      metrics, (int64_t)(h->MET), h->d_seed, s);
      noise = h->noise_buf; keep = h->keep_buf;
    } else {
      launch_zero(metrics, (int64_t)(h->MET), s);
      if (!io->eps) launch_rng_normal(h->eps_buf, (int64_t)B * h->Z, c.sample_std, h->d_seed, 1u, s);
      if (training) {
        noise = io->noise; keep = io->keep_mask;
        if (!noise) { launch_rng_normal(h->noise_buf, hwC * B, 1.0f, h->d_seed, 2u, s); noise = h->noise_buf; }
        if (!keep) { launch_rng_keepmask(h->keep_buf, (int64_t)B * C, kDropout, h->d_seed, 3u, s); keep = h->keep_buf; }
      }
    }
    launch_seed_next(h->d_seed, s);                              // d_seed <- seed + 1 (after this pass's draws: same stream)
    // ---- input transform (multiscale_vae.py:129-160)
    launch_prep(xsrc, noise, keep, h->scales[0].pcur, B, c.input_h, c.input_w, C, c.min_value, c.max_value,
                1.0f / (c.max_value - c.min_value), 1.0f / (1.0f - kDropout), s);
    for (int l = 0; l + 1 < L; ++l) {
      Scale& sc = h->scales[l];
      launch_blur_split(sc.pcur, sc.band, h->scales[l + 1].pcur, B, sc.H, sc.W, C, s);
    }
    // ---- per-scale VAE: the scales are independent until the merge -> one stream each
    stamp(h, 1, s);
    fork_scales(h, s);
    PreOp none{nullptr, nullptr, nullptr};
    for (int ord = 0; ord < L; ++ord) {
      const int si = scale0_first(1) ? ord : h->issue_order[ord];
      if (debug_skip_scale(si)) continue;
      hipStream_t ss = scale_stream(h, si, s);
      profiler().cur_scale = si;
      Scale& sc = h->scales[si];
      chain_begin(h, ss);
      stamp(h, 10 + si, ss);
      ConvGeom g{};
      g.B = B; g.IH = g.OH = sc.H; g.IW = g.OW = sc.W; g.CI = C; g.CO = kConvBaseFilters;
      g.KH = g.KW = 3; g.SH = g.SW = 1; g.PT = g.PL = 1;
      {
        ProfScope ps("convbase_fwd", 4.0 * B * sc.H * sc.W * (C + kConvBaseFilters), 2.0 * B * sc.H * sc.W * 9 * C * kConvBaseFilters, ss);
        const bool cbf = launch_convbase_fwd(sc.band, P + sc.cb_w, P + sc.cb_b, sc.e0, B, sc.H, sc.W, C, kConvBaseFilters, ss, sc.bf);
        if (sc.bf) need16(h, cbf);
        if (!cbf) launch_conv_f(sc.band, P + sc.cb_w, P + sc.cb_b, nullptr, sc.e0, g, none, ACT_ELU, ss);
      }
      const float* x = sc.e0;
      int chained = 0;
      for (Block& blk : sc.enc) {
        if (blk.has_conv && chained == 2) {
          x = blk.cout;                               // computed by the previous block's conv2 launch
        } else if (blk.has_conv) {
          ConvGeom cg = blk.cg; cg.B = B;
          if (sc.bf) {
            if (cg.KH * cg.KW == 1) need16(h, launch16_pw(false, x, P + blk.cw, P + blk.cb, nullptr, nullptr, blk.cout,
                                                        (int64_t)B * cg.IH * cg.IW, (int64_t)cg.IH * cg.IW, cg.CI, cg.CO, ACT_NONE, ss));
            else need16(h, launch16_taps(false, x, P + blk.cw, P + blk.cb, blk.cout, cg, ss));
          } else {
            conv_kxk(h, blk, false, true, x, P + blk.cb, blk.cout, cg, ss);
          }
          x = blk.cout;
        }
        Block* nb = &blk != &sc.enc.back() ? &blk + 1 : nullptr;
        MN* nextmn = (nb && !nb->has_conv) ? &nb->mn : nullptr;
        Block* next3 = (nb && nb->has_conv && nb->cg.KH * nb->cg.KW == 1 && nb->cg.SH == 1 && nb->cg.SW == 1 &&
                        nb->cg.CI == 64 && nb->cg.CO == 32) ? nb : nullptr;
        chained = mn_forward(h, blk.mn, x, B, training, ss, sc.bf, nextmn, chained, next3, false);
        x = blk.mn.out;
      }
      bool fused_heads;
      {
        ProfScope ps("dense_mu_lv", 4.0 * (B * (double)sc.K + 2.0 * sc.K * sc.z), 4.0 * B * sc.K * sc.z, ss);
        fused_heads = launch_dense_mu_lv(x, P + sc.mu_w, P + sc.mu_b, P + sc.lv_w, P + sc.lv_b, sc.mu, sc.lv, B,
                                         (int)sc.K, sc.z, ss, sc.bf);
      }
      if (sc.bf) need16(h, fused_heads);
      if (!fused_heads) {
        launch_gemm_nn(x, P + sc.mu_w, P + sc.mu_b, sc.mu, nullptr, B, (int)sc.K, sc.z, ACT_NONE, ss);
        launch_gemm_nn(x, P + sc.lv_w, P + sc.lv_b, sc.lv, nullptr, B, (int)sc.K, sc.z, ACT_NONE, ss);
      }
      launch_sample_kl(sc.mu, sc.lv, eps, (int)h->Z, sc.z_off, sc.zs, h->losses, 3 + L, 3 + si, B, sc.z, ss);
      if (io->mu) launch_copy_cols(sc.mu, sc.z, 0, io->mu, (int)h->Z, sc.z_off, B, sc.z, ss);
      if (io->log_var) launch_copy_cols(sc.lv, sc.z, 0, io->log_var, (int)h->Z, sc.z_off, B, sc.z, ss);
      if (io->z) launch_copy_cols(sc.zs, sc.z, 0, io->z, (int)h->Z, sc.z_off, B, sc.z, ss);
      decoder_forward(h, sc, B, training, ss);
      stamp(h, 20 + si, ss);
      chain_end(h, ss);
    }
    join_scales(h, s);
    stamp(h, 2, s);
    merge_forward(h, B, h->recon, s);
    // (partial sums of large images go through scale 0's first gradient scratch buffer: free until the backward pass)
    launch_loss_fwd(xsrc, h->recon, h->losses, 3 + L, L, h->sgn, B, c.input_h, c.input_w, C, s, h->scales[0].scratch[0],
                    h->scales[0].scratch_elems * (h->scales[0].bf ? 1 : 2) / 2 * (int64_t)h->cfg.max_batch);
    launch_metrics(h->losses, 3 + L, B, metrics, s);
    stamp(h, 3, s);
    if (io->recon) (void)hipMemcpyAsync(io->recon, h->recon, sizeof(float) * hwC * B, hipMemcpyDeviceToDevice, s);
    if (io->losses)
      (void)hipMemcpyAsync(io->losses, h->losses, sizeof(float) * (3 + L) * B, hipMemcpyDeviceToDevice, s);
  };
  int rc = MVAE_OK;
  if (replayable && xsrc == h->xin) rc = run_captured(h, fkey("F:%d:%d", B, training ? 1 : 0), s0, body);
  else body(s0);
  if (rc != MVAE_OK) return rc;
  h->seed_dev = io->seed + 1; h->seed_dev_valid = true; h->seed_stream = s0;
  h->last_B = B; h->last_training = training; h->last_x = xsrc; h->last_eps = eps;
  if (training) h->last_train_B = B;
  if (h->kernel_gap) return fail(h, MVAE_E_INVALID, "mvae_forward: a bfloat16 launch found no kernel for its shape");
  return check_launch(h, "mvae_forward");
}

static int backward_impl(mvae_handle* h, int phase, float r_factor, float kl_factor, void* stream) {
  if (!h) return MVAE_E_INVALID;
  if (!h->bound || h->last_B <= 0 || !h->last_training)
    return fail(h, MVAE_E_STATE, "mvae_backward needs a preceding training-mode mvae_forward");
  if (phase == 2 && !h->phase0_done) return fail(h, MVAE_E_STATE, "mvae_backward_phase(1) needs phase 0 first");
  const mvae_config& c = h->cfg;
  const int B = h->last_B, L = c.levels, C = c.input_c;
  hipStream_t s0 = static_cast<hipStream_t>(stream);
  const float* P = h->dp;
  float* G = h->dr;
  h->kernel_gap = false;
  set_det_mode(h->det);
  // loss factors go through the device hyper-parameter block: the captured graph does not depend on their values
  {
    const float a = r_factor / (float)B, b = kl_factor / (float)B;
    if (!(h->hp_set[0] == a && h->hp_set[1] == b && h->hp_stream[0] == s0)) {
      launch_set_f3(h->d_hp + HP_RF_OVER_B, a, b, 0.f, 2, s0);
      h->hp_set[0] = a; h->hp_set[1] = b; h->hp_stream[0] = s0;
    }
  }
  // phase bit 1: loss, decoder halves, Dense gradients (everything that fills the leading arena region);
  // phase bit 2: encoder halves, conv_base, gradient-slot fold.  3 = the whole pass in one graph.
  auto body = [=](hipStream_t s) {
  PreOp none{nullptr, nullptr, nullptr};
  if (phase & 1) stamp(h, 4, s);
  if (phase & 1) {
  h->ev_next = 0;
  launch_zero(G, (int64_t)(h->P), s);
  if (h->gslots.n) launch_slot_zero(h->d_slot_chunks, (int)h->slot_chunks.size(), h->gslots.base, h->gslots.stride, h->gslots.n, s);
  // ---- loss -> clip/denormalise -> merge
  launch_loss_bwd(h->last_x, h->recon, h->scales[0].merged, h->sgn, h->scales[0].dy, B, c.input_h, c.input_w, C,
                  c.min_value, c.max_value, h->d_hp, s);
  for (int i = 0; i + 1 < L; ++i)
    launch_upsample_bwd(h->scales[i].dy, h->scales[i + 1].dy, B, h->scales[i + 1].H, h->scales[i + 1].W, C, s);
  }
  stamp(h, 5, s);
  fork_scales(h, s);
  hipStream_t s_main = s;
  // One scale's backward: `bits` & 1 = decoder half (ends with sc.d_mid), & 2 = encoder half.
  auto scale_half = [&](int si, int bits) {
    if (debug_skip_scale(si)) return;
    hipStream_t s = scale_stream(h, si, s_main);
    profiler().cur_scale = si;
    Scale& sc = h->scales[si];
    const int64_t M = (int64_t)B * sc.H * sc.W;
    float* d = nullptr;
    chain_begin(h, s);
    if (bits & 1) stamp(h, 30 + si, s);
    if (bits & 1) {
    for (int k = 0; k < 4; ++k) { sc.scratch_used[k] = false; sc.buf_pending[k] = false; }
    launch_zero(sc.dg, (int64_t)sc.dg_total * B * sc.cmax, s);   // squeeze-excite gate gradients (all slot copies)
    // ---- output conv + decoder BatchNorm
    const float* xbn = sc.dec.back().mn.out;
    ConvGeom go = geom1x1(B, sc.H, sc.W, sc.dc, C);
    PreOp bn{nullptr, sc.bn_scale, sc.bn_shift};
    d = acquire(h, sc, s);
    bool fused_head;
    {
      ProfScope ps("head_bwd", 4.0 * M * (3.0 * sc.dc + 2.0 * C), 6.0 * M * sc.dc * C, s);
      launch_zero(sc.head_S, (int64_t)head_slots() * 2 * sc.dc, s);
      fused_head = launch_head_bwd(xbn, sc.dy, P + sc.out_w, P + sc.bn_g, sc.bn_scale, sc.bn_shift, sc.bn_mean,
                                   sc.bn_invstd, sc.head_S, G + sc.out_w, G + sc.out_b, G + sc.bn_g, G + sc.bn_b, d, M,
                                   sc.dc, C, h->gslots, s, sc.bf);
    }
    if (sc.bf) need16(h, fused_head);
    if (!fused_head) {
      launch_zero(sc.bn_sum_d, (int64_t)(sc.dc), s);
      launch_zero(sc.bn_sum_dx, (int64_t)(sc.dc), s);
      launch_conv_wgrad(xbn, sc.dy, G + sc.out_w, G + sc.out_b, go, bn, h->gslots, s);
      launch_conv_t(sc.dy, P + sc.out_w, nullptr, nullptr, d, go, s);
      launch_bn_bwd_reduce(d, xbn, sc.bn_mean, sc.bn_invstd, sc.bn_sum_d, sc.bn_sum_dx, M, sc.dc, s);
      launch_bn2d_bwd_apply(d, xbn, sc.bn_mean, sc.bn_invstd, P + sc.bn_g, sc.bn_sum_d, sc.bn_sum_dx, G + sc.bn_g,
                            G + sc.bn_b, M, sc.dc, s);
    }
    // ---- decoder blocks, last to first
    for (int i = (int)sc.dec.size() - 1; i >= 0; --i) {
      Block& blk = sc.dec[i];
      const float* prev = i == 0 ? sc.d0 : sc.dec[i - 1].mn.out;
      const float* xin = blk.has_conv ? blk.cout : prev;
      d = mn_backward(h, sc, blk.mn, xin, d, B, s);
      if (blk.has_conv) {     // Conv2DTranspose: big = its output (d), small = its input (prev)
        ConvGeom g = blk.cg; g.B = B;
        if (sc.bf) {
          bool db_done = false;                       // the bias gradient (column sums of d) inside the weight-gradient kernel
          need16(h, launch16_wgrad(d, prev, G + blk.cw, nullptr, g, h->gslots, s, G + blk.cb, &db_done));
          if (!db_done)
            need16(h, launch_colstat_opt(0, d, nullptr, 0, 0.f, h->gslots.at(G + blk.cb), h->gslots.count(), h->gslots.stride,
                                         (int64_t)B * g.IH * g.IW, g.CI, s, true));
        } else {
          hipStream_t w = wgrad_begin(h, sc, s, g.KH * g.KW > 1);
          launch_conv_wgrad(d, prev, G + blk.cw, nullptr, g, none, h->gslots, w);
          // (the float32 5 x 5 weight-gradient kernel cannot take the bias gradient along as k16_wgrad_seg does: at 254 of 256
          // registers, four more accumulating adds in its tap loop made hipcc spill 36 - 95 registers)
          if (!launch_colstat_opt(0, d, nullptr, 0, 0.f, h->gslots.at(G + blk.cb), h->gslots.count(), h->gslots.stride,
                                  (int64_t)B * g.IH * g.IW, g.CI, w))
            launch_colsum(d, G + blk.cb, (int64_t)B * g.IH * g.IW, g.CI, w);
          wgrad_reads(h, sc, d, w);
        }
        float* n = acquire(h, sc, s);
        if (sc.bf) {
          if (g.KH * g.KW == 1) need16(h, launch16_pw(false, d, P + blk.cw, nullptr, nullptr, nullptr, n, (int64_t)B * g.IH * g.IW,
                                                      (int64_t)g.IH * g.IW, g.CI, g.CO, ACT_NONE, s));
          else need16(h, launch16_taps(false, d, P + blk.cw, nullptr, n, g, s));
        } else {
          conv_kxk(h, blk, false, false, d, nullptr, n, g, s);
        }
        release(sc, d);
        d = n;
      }
    }
    // ---- decoder Dense, sampling + KL, encoder Dense heads
    {
      hipStream_t w = wgrad_begin(h, sc, s);
      bool fused_ddw;
      {
        ProfScope ps("dense_wgrad_dec", 4.0 * (B * (double)sc.K + sc.K * sc.z), 2.0 * B * sc.K * sc.z, w);
        fused_ddw = launch_dense_wgrad_dec(sc.zs, d, G + sc.dd_w, G + sc.dd_b, B, sc.z, (int)sc.K, w, sc.bf);
      }
      if (sc.bf) need16(h, fused_ddw);
      if (!fused_ddw)
        launch_gemm_tn(sc.zs, d, G + sc.dd_w, G + sc.dd_b, B, sc.z, (int)sc.K, nullptr, nullptr, nullptr, w);
      wgrad_reads(h, sc, d, w);
    }
    bool fused_dz;
    {
      ProfScope ps("dense_dz", 4.0 * (B * (double)sc.K + sc.K * sc.z), 2.0 * B * sc.K * sc.z, s);
      fused_dz = launch_dense_dz(d, P + sc.dd_w, sc.dz, B, sc.z, (int)sc.K, s, sc.bf);
    }
    if (sc.bf) need16(h, fused_dz);
    if (!fused_dz) launch_gemm_nt(d, P + sc.dd_w, sc.dz, B, sc.z, (int)sc.K, nullptr, 0, s);
    launch_sample_kl_bwd(sc.dz, sc.mu, sc.lv, h->last_eps, (int)h->Z, sc.z_off, sc.dmu, sc.dlv,
                         h->d_hp, B, sc.z, s);
    const float* flat = sc.enc.back().mn.out;
    {   // dmu / dlv are per-scale buffers written once per backward: safe to read from the side stream
      hipStream_t w = wgrad_begin(h, sc, s);
      bool fused_w;
      {
        ProfScope ps("dense_wgrad", 4.0 * (B * (double)sc.K + 2.0 * sc.K * sc.z), 4.0 * B * sc.K * sc.z, w);
        fused_w = launch_dense_wgrad_mu_lv(flat, sc.dmu, sc.dlv, G + sc.mu_w, G + sc.lv_w, G + sc.mu_b, G + sc.lv_b, B,
                                           (int)sc.K, sc.z, w, sc.bf);
      }
      if (sc.bf) need16(h, fused_w);
      if (!fused_w) {
        launch_gemm_tn(flat, sc.dmu, G + sc.mu_w, G + sc.mu_b, B, (int)sc.K, sc.z, nullptr, nullptr, nullptr, w);
        launch_gemm_tn(flat, sc.dlv, G + sc.lv_w, G + sc.lv_b, B, (int)sc.K, sc.z, nullptr, nullptr, nullptr, w);
      }
    }
    release(sc, d);                              // the side stream may still read it: take another buffer
    d = acquire(h, sc, s);
    bool fused_df;
    {
      ProfScope ps("dense_dflat", 4.0 * (B * (double)sc.K + 2.0 * sc.K * sc.z), 4.0 * B * sc.K * sc.z, s);
      fused_df = launch_dense_dflat(sc.dmu, sc.dlv, P + sc.mu_w, P + sc.lv_w, d, B, (int)sc.K, sc.z, s, sc.bf);
    }
    if (sc.bf) need16(h, fused_df);
    if (!fused_df) {
      launch_gemm_nt(sc.dmu, P + sc.mu_w, d, B, (int)sc.K, sc.z, nullptr, 0, s);
      launch_gemm_nt(sc.dlv, P + sc.lv_w, d, B, (int)sc.K, sc.z, nullptr, 1, s);
    }
    if (!(bits & 2)) wgrad_join(h, sc, s);     // phase 0 ends here: its side-stream Dense gradients must be final
    sc.d_mid = d;
    }   // decoder half
    if (!(bits & 2)) { chain_end(h, s); return; }
    d = sc.d_mid;
    // ---- encoder blocks, last to first
    for (int i = (int)sc.enc.size() - 1; i >= 0; --i) {
      Block& blk = sc.enc[i];
      const float* prev = i == 0 ? sc.e0 : sc.enc[i - 1].mn.out;
      const float* xin = blk.has_conv ? blk.cout : prev;
      d = mn_backward(h, sc, blk.mn, xin, d, B, s);
      if (blk.has_conv) {     // Conv2D: big = its input (prev), small = its output (d)
        ConvGeom g = blk.cg; g.B = B;
        if (sc.bf) {
          need16(h, launch16_wgrad(prev, d, G + blk.cw, G + blk.cb, g, h->gslots, s));
        } else {
          hipStream_t w = wgrad_begin(h, sc, s, g.KH * g.KW > 1);
          launch_conv_wgrad(prev, d, G + blk.cw, G + blk.cb, g, none, h->gslots, w);
          wgrad_reads(h, sc, d, w);
        }
        float* n = acquire(h, sc, s);
        if (sc.bf) {
          if (g.KH * g.KW == 1) need16(h, launch16_pw(true, d, P + blk.cw, nullptr, nullptr, nullptr, n, (int64_t)B * g.IH * g.IW,
                                                      (int64_t)g.IH * g.IW, g.CO, g.CI, ACT_NONE, s));
          else need16(h, launch16_taps(true, d, P + blk.cw, nullptr, n, g, s));
        } else {
          conv_kxk(h, blk, true, false, d, nullptr, n, g, s);
        }
        release(sc, d);
        d = n;
      }
    }
    // ---- conv_base (ELU): weight gradients only, its input is data
    bool fused_base;
    {
      ProfScope ps("convbase_wgrad", 4.0 * M * (2.0 * kConvBaseFilters + C), 2.0 * M * 9 * C * kConvBaseFilters, s);
      fused_base = launch_convbase_wgrad(sc.band, d, sc.e0, G + sc.cb_w, G + sc.cb_b, B, sc.H, sc.W, C,
                                         kConvBaseFilters, h->gslots, s, sc.bf);
    }
    if (sc.bf) need16(h, fused_base);
    wgrad_join(h, sc, s);                       // every side-stream weight gradient of this scale is done
    if (!fused_base) {
      launch_elu_bwd(d, sc.e0, M * kConvBaseFilters, s);
      ConvGeom g{};
      g.B = B; g.IH = g.OH = sc.H; g.IW = g.OW = sc.W; g.CI = C; g.CO = kConvBaseFilters;
      g.KH = g.KW = 3; g.SH = g.SW = 1; g.PT = g.PL = 1;
      launch_conv_wgrad(sc.band, d, G + sc.cb_w, G + sc.cb_b, g, none, h->gslots, s);
    }
    release(sc, d);
    stamp(h, 40 + si, s);
    chain_end(h, s);
  };
  // (Holding the smaller scales back until scale 0 reaches its MFMA-bound 5x5 convolutions, so that their HBM-bound work
  // would fill those windows, was measured: 5.88 .. 6.05 ms against 5.92 -- their chains are latency-bound and only move
  // the contention.  They start at the fork.)
  for (int ord = 0; ord < L; ++ord) scale_half(scale0_first(2) ? ord : h->issue_order[ord], phase);
  join_scales(h, s_main);
  stamp(h, 6, s_main);
  profiler().cur_scale = -1;
  if ((phase & 2) && h->gslots.n)
    launch_slot_sum(h->d_slot_chunks, (int)h->slot_chunks.size(), G, h->gslots.base, h->gslots.stride, h->gslots.n, s_main);
  stamp(h, 7, s_main);
  };
  int rc = MVAE_OK;
  if (h->last_x == h->xin && h->last_eps == h->eps_buf)
    rc = run_captured(h, fkey("B%d:%d", phase, B), s0, body);
  else body(s0);
  if (rc != MVAE_OK) return rc;
  h->phase0_done = phase == 1;
  if (h->kernel_gap) return fail(h, MVAE_E_INVALID, "mvae_backward: a bfloat16 launch found no kernel for its shape");
  return check_launch(h, "mvae_backward");
}

int mvae_backward(mvae_handle* h, float r_factor, float kl_factor, void* stream) {
  return backward_impl(h, 3, r_factor, kl_factor, stream);
}

int mvae_backward_phase(mvae_handle* h, int32_t phase, float r_factor, float kl_factor, void* stream) {
  if (phase != 0 && phase != 1) return h ? fail(h, MVAE_E_INVALID, "phase must be 0 or 1") : MVAE_E_INVALID;
  return backward_impl(h, phase == 0 ? 1 : 2, r_factor, kl_factor, stream);
}

int64_t mvae_reduce_split(const mvae_handle* h) { return h ? h->reduce_split : -1; }

int mvae_graph_stats(const mvae_handle* h, int32_t* captured, int32_t* eager_fallbacks) {
  if (!h) return MVAE_E_INVALID;
  if (captured) *captured = (int32_t)(h->graphs.size() + h->seg_graphs.size());
  if (eager_fallbacks) *eager_fallbacks = h->eager_fallbacks;
  return MVAE_OK;
}

int mvae_apply_adagrad(mvae_handle* h, float lr, float clip_norm, float grad_scale, void* stream) {
  if (!h) return MVAE_E_INVALID;
  if (!h->bound) return fail(h, MVAE_E_STATE, "mvae_bind has not been called");
  hipStream_t s0 = static_cast<hipStream_t>(stream);
  set_det_mode(h->det);                     // (process-global launcher switch: every entry point sets it from ITS handle)
  const int Bt = h->last_train_B;
  const bool clip = clip_norm > 0.f;
  if (!(h->hp_set[2] == lr && h->hp_set[3] == clip_norm && h->hp_set[4] == grad_scale && h->hp_stream[1] == s0)) {
    launch_set_f3(h->d_hp + HP_LR, lr, clip_norm, grad_scale, 3, s0);   // HP_LR, HP_CLIP, HP_GRAD_SCALE are adjacent
    h->hp_set[2] = lr; h->hp_set[3] = clip_norm; h->hp_set[4] = grad_scale; h->hp_stream[1] = s0;
  }
  auto body = [=](hipStream_t s) {
    stamp(h, 8, s);
    launch_opt_prepare(h->dp, h->dr, h->d_chunks, (int)h->chunks.size(), h->d_norms, h->d_hp, s);
    launch_opt_apply(h->dp, h->dr, h->da, h->d_chunks, (int)h->chunks.size(), h->d_norms, h->d_hp, clip, s);
    // BN moving statistics from the (all-reduced, hence grad_scale) batch statistics
    launch_state_update(h->ds, h->dr + h->P, h->d_sdescs, (int)h->sdescs.size(), h->d_hp, Bt, s);
    stamp(h, 9, s);
  };
  int rc = run_captured(h, fkey("A:%d:%d", Bt, clip ? 1 : 0), s0, body);
  if (rc != MVAE_OK) return rc;
  return check_launch(h, "mvae_apply_adagrad");
}

int mvae_train_step(mvae_handle* h, const mvae_step_io* io, float r_factor, float kl_factor, float lr,
                    float clip_norm, void* stream) {
  if (!io || !io->training) return h ? fail(h, MVAE_E_INVALID, "mvae_train_step needs io->training = 1") : MVAE_E_INVALID;
  int rc = mvae_forward(h, io, stream);
  if (rc != MVAE_OK) return rc;
  rc = mvae_backward(h, r_factor, kl_factor, stream);
  if (rc != MVAE_OK) return rc;
  return mvae_apply_adagrad(h, lr, clip_norm, 1.0f, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Data-parallel exchange through RCCL, bound without a link-time dependency: librccl is looked up in the process first
// (a Python caller has torch's copy loaded; the two must not both initialise the same GPU's network state) and opened
// from the ROCm installation otherwise.  SURVEY 8(b) / 8(e): one ncclAllReduce(sum, float32) over the reduce arena
// [gradients | BatchNorm batch statistics | metrics] per step on the caller's stream, then the identical clipnorm + Adagrad
// on every rank with grad_scale = 1 / nranks.
extern "C++" {
namespace {

struct RcclId { char internal[MVAE_COMM_ID_BYTES]; };           // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
};
constexpr int kNcclFloat32 = 7, kNcclSum = 0;                   // rccl.h: ncclFloat32, ncclSum

RcclApi& rccl() {
  static RcclApi api;
  if (api.lib || !api.err.empty()) return api;
  const char* names[] = {getenv("MVAE_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {                                 // a copy already in the process wins (RTLD_NOLOAD)
    if (n && *n && (api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  }
  for (const char* n : names) {
    if (api.lib) break;
    if (n && *n) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  }
  if (!api.lib) {
    api.err = "librccl.so not found (set MVAE_RCCL_LIB)";
    return api;
  }
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.lib, "ncclAllReduce"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString) {
    api.err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
    api.lib = nullptr;
  }
  return api;
}

}  // namespace
}  // extern "C++"

int mvae_comm_unique_id(char id[MVAE_COMM_ID_BYTES]) {
  if (!id) return MVAE_E_INVALID;
  RcclApi& r = rccl();
  if (!r.lib) return fail(nullptr, MVAE_E_STATE, "%s", r.err.c_str());
  RcclId u;
  const int rc = r.GetUniqueId(&u);
  if (rc != 0) return fail(nullptr, MVAE_E_HIP, "ncclGetUniqueId: %s", r.GetErrorString(rc));
  memcpy(id, u.internal, MVAE_COMM_ID_BYTES);
  return MVAE_OK;
}

int mvae_comm_init(mvae_handle* h, const char id[MVAE_COMM_ID_BYTES], int32_t rank, int32_t nranks) {
  if (!h || !id) return MVAE_E_INVALID;
  if (!h->bound) return fail(h, MVAE_E_STATE, "mvae_bind has not been called (the communicator belongs to the bound device)");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(h, MVAE_E_INVALID, "rank %d outside [0, %d)", rank, nranks);
  if (h->comm) return fail(h, MVAE_E_STATE, "mvae_comm_init: this handle already has a communicator");
  RcclApi& r = rccl();
  if (!r.lib) return fail(h, MVAE_E_STATE, "%s", r.err.c_str());
  if (hipSetDevice(h->device) != hipSuccess) return fail(h, MVAE_E_HIP, "hipSetDevice(%d) failed", h->device);
  RcclId u;
  memcpy(u.internal, id, MVAE_COMM_ID_BYTES);
  void* comm = nullptr;
  const int rc = r.CommInitRank(&comm, nranks, u, rank);
  if (rc != 0 || !comm) return fail(h, MVAE_E_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, r.GetErrorString(rc));
  h->comm = comm;
  h->comm_rank = rank;
  h->comm_nranks = nranks;
  return MVAE_OK;
}

int mvae_comm_destroy(mvae_handle* h) {
  if (!h) return MVAE_E_INVALID;
  if (h->comm) {
    (void)hipDeviceSynchronize();
    (void)rccl().CommDestroy(h->comm);
    h->comm = nullptr;
    h->comm_nranks = 1;
    h->comm_rank = 0;
  }
  return MVAE_OK;
}

int mvae_comm_size(const mvae_handle* h) { return h ? (h->comm ? h->comm_nranks : 0) : MVAE_E_INVALID; }

int mvae_allreduce(mvae_handle* h, int64_t offset, int64_t count, void* stream) {
  if (!h) return MVAE_E_INVALID;
  if (!h->bound) return fail(h, MVAE_E_STATE, "mvae_bind has not been called");
  if (!h->comm) return fail(h, MVAE_E_STATE, "mvae_comm_init has not been called");
  const int64_t n = h->P + h->S + h->MET;
  if (count < 0) count = n - offset;
  if (offset < 0 || count < 0 || offset + count > n)
    return fail(h, MVAE_E_INVALID, "mvae_allreduce: [%lld, %lld) outside the reduce arena of %lld floats", (long long)offset,
                (long long)(offset + count), (long long)n);
  if (count == 0) return MVAE_OK;
  float* p = h->dr + offset;
  const int rc = rccl().AllReduce(p, p, (size_t)count, kNcclFloat32, kNcclSum, h->comm, static_cast<hipStream_t>(stream));
  if (rc != 0) return fail(h, MVAE_E_HIP, "ncclAllReduce: %s", rccl().GetErrorString(rc));
  return MVAE_OK;
}

int mvae_train_step_dp(mvae_handle* h, const mvae_step_io* io, float r_factor, float kl_factor, float lr, float clip_norm,
                       void* stream) {
  if (!io || !io->training) return h ? fail(h, MVAE_E_INVALID, "mvae_train_step_dp needs io->training = 1") : MVAE_E_INVALID;
  if (!h || !h->comm) return h ? fail(h, MVAE_E_STATE, "mvae_comm_init has not been called") : MVAE_E_INVALID;
  int rc = mvae_forward(h, io, stream);
  if (rc != MVAE_OK) return rc;
  rc = mvae_backward(h, r_factor, kl_factor, stream);
  if (rc != MVAE_OK) return rc;
  rc = mvae_allreduce(h, 0, -1, stream);
  if (rc != MVAE_OK) return rc;
  return mvae_apply_adagrad(h, lr, clip_norm, 1.0f / (float)h->comm_nranks, stream);
}

int mvae_reg_loss(mvae_handle* h, float* out_dev, void* stream) {
  if (!h || !out_dev) return MVAE_E_INVALID;
  if (!h->bound) return fail(h, MVAE_E_STATE, "mvae_bind has not been called");
  hipStream_t s = static_cast<hipStream_t>(stream);
  launch_zero(out_dev, 1, s);
  launch_reg_loss(h->dp, h->d_chunks, (int)h->chunks.size(), out_dev, s);
  return check_launch(h, "mvae_reg_loss");
}

int mvae_decode(mvae_handle* h, const float* z, int32_t batch, float* recon, void* stream) {
  if (!h || !z || !recon) return MVAE_E_INVALID;
  if (!h->bound) return fail(h, MVAE_E_STATE, "mvae_bind has not been called");
  const mvae_config& c = h->cfg;
  if (batch <= 0 || batch > c.max_batch) return fail(h, MVAE_E_INVALID, "batch %d outside [1, %d]", batch, c.max_batch);
  hipStream_t s = static_cast<hipStream_t>(stream);
  h->kernel_gap = false;
  set_det_mode(h->det);
  for (Scale& sc : h->scales) {
    launch_copy_cols(z, (int)h->Z, sc.z_off, sc.zs, sc.z, 0, batch, sc.z, s);
    decoder_forward(h, sc, batch, false, s);
  }
  merge_forward(h, batch, recon, s);
  h->last_B = 0;   // activations no longer belong to a training forward
  if (h->kernel_gap) return fail(h, MVAE_E_INVALID, "mvae_decode: a bfloat16 launch found no kernel for its shape");
  return check_launch(h, "mvae_decode");
}

int mvae_gather_rows(int32_t device, const float* src, const int64_t* idx, int64_t n, int64_t row_elems, float* dst,
                     void* stream) {
  if (!src || !idx || !dst || n < 0 || row_elems <= 0) return MVAE_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return MVAE_E_HIP;
  set_det_mode(false);                      // stateless entry point: never inherits a handle's deterministic mode
  launch_gather_rows(src, idx, dst, n, row_elems, static_cast<hipStream_t>(stream));
  return hipGetLastError() == hipSuccess ? MVAE_OK : MVAE_E_HIP;
}

// ---- stand-alone Laplacian pyramid (layer_blocks.py:23-185), stateless ----------------------
int mvae_laplacian_split(int32_t device, const float* x, int32_t batch, int32_t H, int32_t W, int32_t C, int32_t levels,
                         float min_value, float max_value, const float* gauss9, float* const* out, float* work,
                         void* stream) {
  if (!x || !gauss9 || !out || !work || batch <= 0 || C <= 0 || C > 8 || levels < 1 || levels > MVAE_MAX_LEVELS) return MVAE_E_INVALID;
  if (H <= 0 || W <= 0 || (H % (1 << (levels - 1))) || (W % (1 << (levels - 1)))) return MVAE_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return MVAE_E_HIP;
  set_det_mode(false);                      // stateless entry point: never inherits a handle's deterministic mode
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t n0 = (int64_t)batch * H * W * C;
  // level 0 reads the raw image and normalises on the fly; work holds the normalised image of levels 1 .. levels-2
  // (the last level's normalised image IS its output)
  if (levels == 1) {
    launch_prep(x, nullptr, nullptr, out[0], batch, H, W, C, min_value, max_value, 0.f, 1.f, s);
    return hipGetLastError() == hipSuccess ? MVAE_OK : MVAE_E_HIP;
  }
  const float* cur = x;
  int h = H, w = W;
  int64_t off = 0;
  for (int i = 0; i + 1 < levels; ++i) {
    const int64_t nd = (int64_t)batch * (h / 2) * (w / 2) * C;
    float* down = i + 2 == levels ? out[levels - 1] : work + off;
    if (!launch_lap_level(cur, out[i], down, batch, h, w, C, gauss9, i == 0, min_value, max_value, s)) return MVAE_E_INVALID;
    off += nd;
    cur = down;
    h /= 2; w /= 2;
  }
  (void)n0;
  return hipGetLastError() == hipSuccess ? MVAE_OK : MVAE_E_HIP;
}

int mvae_laplacian_merge(int32_t device, const float* const* in, int32_t batch, int32_t H, int32_t W, int32_t C,
                         int32_t levels, float min_value, float max_value, float* out, float* work, void* stream) {
  if (!in || !out || !work || batch <= 0 || C <= 0 || C > 8 || levels < 1 || levels > MVAE_MAX_LEVELS) return MVAE_E_INVALID;
  if (H <= 0 || W <= 0 || (H % (1 << (levels - 1))) || (W % (1 << (levels - 1)))) return MVAE_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return MVAE_E_HIP;
  set_det_mode(false);                      // stateless entry point: never inherits a handle's deterministic mode
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (levels == 1) {
    launch_denorm_clip(in[0], out, (int64_t)batch * H * W * C, min_value, max_value, s);
    return hipGetLastError() == hipSuccess ? MVAE_OK : MVAE_E_HIP;
  }
  // work: two buffers of the finest level's size, used alternately for the running sum
  const int64_t n0 = (int64_t)batch * H * W * C;
  const float* coarse = in[levels - 1];
  for (int i = levels - 2; i >= 0; --i) {
    const int h = H >> i, w = W >> i;
    float* dst = i == 0 ? out : work + (i & 1) * n0;
    if (!launch_lap_merge(coarse, in[i], dst, batch, h, w, C, i == 0, min_value, max_value, s)) return MVAE_E_INVALID;
    coarse = dst;
  }
  return hipGetLastError() == hipSuccess ? MVAE_OK : MVAE_E_HIP;
}

int mvae_laplacian_merge_mix(int32_t device, const float* const* in, int32_t batch, int32_t H, int32_t W, int32_t C,
                             int32_t levels, int32_t filters, const float* const* w3, const float* const* b3,
                             const float* const* w1, float min_value, float max_value, float* out, float* work,
                             void* stream) {
  if (!in || !out || !work || !w3 || !b3 || !w1 || batch <= 0 || C <= 0 || C > 8 || filters <= 0 || levels < 1 ||
      levels > MVAE_MAX_LEVELS)
    return MVAE_E_INVALID;
  if (H <= 0 || W <= 0 || (H % (1 << (levels - 1))) || (W % (1 << (levels - 1)))) return MVAE_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return MVAE_E_HIP;
  set_det_mode(false);                      // stateless entry point: never inherits a handle's deterministic mode
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t n0 = (int64_t)batch * H * W;
  if (levels == 1) {
    launch_denorm_clip(in[0], out, n0 * C, min_value, max_value, s);
    return hipGetLastError() == hipSuccess ? MVAE_OK : MVAE_E_HIP;
  }
  float* cat = work;                         // [B,H,W,2C]
  float* hid = cat + n0 * 2 * C;             // [B,H,W,filters]
  float* run[2] = {hid + n0 * filters, hid + n0 * filters + n0 * C};
  const float* coarse = in[levels - 1];
  for (int i = levels - 2; i >= 0; --i) {
    const int h = H >> i, w = W >> i;
    if (!launch_lap_concat(coarse, in[i], cat, batch, h, w, C, s)) return MVAE_E_INVALID;
    ConvGeom g3{};
    g3.B = batch; g3.IH = g3.OH = h; g3.IW = g3.OW = w; g3.CI = 2 * C; g3.CO = filters;
    g3.KH = g3.KW = 3; g3.SH = g3.SW = 1; g3.PT = g3.PL = 1;
    launch_conv_f_any(cat, w3[i], b3[i], nullptr, hid, g3, ACT_RELU, s);                       // mixing
    ConvGeom g1 = geom1x1(batch, h, w, filters, C);
    float* dst = run[i & 1];
    launch_conv_f_any(hid, w1[i], nullptr, in[i], dst, g1, ACT_TANH, s);                       // retargeting + Add
    coarse = dst;
  }
  launch_denorm_clip(coarse, out, n0 * C, min_value, max_value, s);
  return hipGetLastError() == hipSuccess ? MVAE_OK : MVAE_E_HIP;
}

// ---- stand-alone blocks of the reference's block library, stateless ------------------------------
static ConvGeom geom_same(int B, int H, int W, int ci, int co, int kh, int kw) {
  ConvGeom g{};
  g.B = B; g.IH = g.OH = H; g.IW = g.OW = W; g.CI = ci; g.CO = co; g.KH = kh; g.KW = kw; g.SH = g.SW = 1;
  g.PT = (kh - 1) / 2; g.PL = (kw - 1) / 2;                     // TF 'SAME' at stride 1: total k - 1, the smaller half first
  return g;
}

int mvae_mnv2_forward(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t F, const float* w0,
                      const float* b0, const float* wd, const float* bd, const float* w2, const float* b2, float* t0,
                      float* t1, float* u, float* y, void* stream) {
  if (!x || !w0 || !b0 || !wd || !bd || !w2 || !b2 || !t0 || !t1 || !u || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || F <= 0)
    return MVAE_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return MVAE_E_HIP;
  set_det_mode(false);                      // stateless entry point: never inherits a handle's deterministic mode
  hipStream_t s = static_cast<hipStream_t>(stream);
  PreOp none{nullptr, nullptr, nullptr};
  launch_conv_f(x, w0, b0, nullptr, t0, geom_same(B, H, W, C, F, 1, 1), none, ACT_NONE, s);      // conv0: 1x1, linear (:503-511)
  launch_dw_fwd(t0, wd, bd, t1, B, H, W, F, s);                                                  // conv1: depthwise 3x3, relu (:513-521)
  launch_conv_f(t1, w2, b2, nullptr, u, geom_same(B, H, W, F, C, 1, 1), none, ACT_RELU, s);      // conv2: 1x1, relu (:527-535)
  launch_add2(u, x, y, (int64_t)B * H * W * C, s);                                               // Add (:542-545)
  return hipGetLastError() == hipSuccess ? MVAE_OK : MVAE_E_HIP;
}

int mvae_mnv2_backward(int32_t device, const float* x, const float* t0, const float* t1, const float* u, const float* dy,
                       int32_t B, int32_t H, int32_t W, int32_t C, int32_t F, const float* w0, const float* wd,
                       const float* w2, float* dx, float* dw0, float* db0, float* dwd, float* dbd, float* dw2, float* db2,
                       float* work, void* stream) {
  if (!x || !t0 || !t1 || !u || !dy || !w0 || !wd || !w2 || !dx || !dw0 || !db0 || !dwd || !dbd || !dw2 || !db2 || !work ||
      B <= 0 || H <= 0 || W <= 0 || C <= 0 || F <= 0)
    return MVAE_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return MVAE_E_HIP;
  set_det_mode(false);                      // stateless entry point: never inherits a handle's deterministic mode
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t M = (int64_t)B * H * W;
  float *du = work, *d1 = work + M * C, *dt0 = d1 + M * F;
  PreOp none{nullptr, nullptr, nullptr};
  GradSlots direct;
  const ConvGeom g0 = geom_same(B, H, W, C, F, 1, 1), g2 = geom_same(B, H, W, F, C, 1, 1);
  launch_relu_bwd(dy, u, du, M * C, s);                                       // through conv2's relu
  launch_conv_wgrad(t1, du, dw2, db2, g2, none, direct, s);                   // dW2 += t1^T du, db2 += sum du
  launch_conv_t(du, w2, nullptr, nullptr, d1, g2, s);                         // dt1 = du . W2^T
  launch_relu_bwd(d1, t1, d1, M * F, s);                                      // through the depthwise relu
  launch_dw_wgrad(t0, d1, dwd, dbd, B, H, W, F, s);
  launch_dw_bwd_plain(d1, wd, dt0, B, H, W, F, s);                            // conv0 is linear: no mask behind the depthwise
  launch_conv_wgrad(x, dt0, dw0, db0, g0, none, direct, s);
  launch_conv_t(dt0, w0, nullptr, dy, dx, g0, s);                             // dx = dt0 . W0^T + dy (the skip connection)
  return hipGetLastError() == hipSuccess ? MVAE_OK : MVAE_E_HIP;
}

int mvae_resnet_forward(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t F, int32_t kh,
                        int32_t kw, int32_t relu, const float* w0, const float* b0, const float* w1, const float* b1,
                        const float* ws, const float* bs, float* x0, float* skip, float* y, void* stream) {
  if (!x || !w0 || !b0 || !w1 || !b1 || !x0 || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || F <= 0 || kh <= 0 || kw <= 0)
    return MVAE_E_INVALID;
  if (C != F && (!ws || !bs || !skip)) return MVAE_E_INVALID;              // a 1x1 skip convolution when the widths differ
  if (hipSetDevice(device) != hipSuccess) return MVAE_E_HIP;
  set_det_mode(false);                      // stateless entry point: never inherits a handle's deterministic mode
  hipStream_t s = static_cast<hipStream_t>(stream);
  PreOp none{nullptr, nullptr, nullptr};
  launch_conv_f(x, w0, b0, nullptr, x0, geom_same(B, H, W, C, F, kh, kw), none, relu ? ACT_RELU : ACT_NONE, s);   // conv0 (:830-838)
  const float* sk = x;                                                                                       // skip (:858-872)
  if (C != F) { launch_conv_f(x, ws, bs, nullptr, skip, geom_same(B, H, W, C, F, 1, 1), none, ACT_NONE, s); sk = skip; }
  launch_conv_f(x0, w1, b1, sk, y, geom_same(B, H, W, F, F, kh, kw), none, ACT_NONE, s);                     // conv1 + Add (:840-848, 874)
  if (relu) launch_relu_add(y, nullptr, true, (int64_t)B * H * W * F, s);                                    // Activation (:877-879)
  return hipGetLastError() == hipSuccess ? MVAE_OK : MVAE_E_HIP;
}

int mvae_resnet_backward(int32_t device, const float* x, const float* x0, const float* y, const float* dy, int32_t B, int32_t H,
                         int32_t W, int32_t C, int32_t F, int32_t kh, int32_t kw, int32_t relu, const float* w0,
                         const float* w1, const float* ws, float* dx, float* dw0, float* db0, float* dw1, float* db1,
                         float* dws, float* dbs, float* work, void* stream) {
  if (!x || !x0 || !y || !dy || !w0 || !w1 || !dx || !dw0 || !db0 || !dw1 || !db1 || !work || B <= 0 || H <= 0 || W <= 0 ||
      C <= 0 || F <= 0 || kh <= 0 || kw <= 0)
    return MVAE_E_INVALID;
  if (C != F && (!ws || !dws || !dbs)) return MVAE_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return MVAE_E_HIP;
  set_det_mode(false);                      // stateless entry point: never inherits a handle's deterministic mode
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t M = (int64_t)B * H * W;
  float *dpre = work, *d0 = work + M * F, *tmp = d0 + M * F;               // [M,F], [M,F], [M,C]
  PreOp none{nullptr, nullptr, nullptr};
  GradSlots direct;
  const ConvGeom g0 = geom_same(B, H, W, C, F, kh, kw), g1 = geom_same(B, H, W, F, F, kh, kw), gs = geom_same(B, H, W, C, F, 1, 1);
  if (relu) launch_relu_bwd(dy, y, dpre, M * F, s);
  else (void)hipMemcpyAsync(dpre, dy, sizeof(float) * M * F, hipMemcpyDeviceToDevice, s);
  launch_conv_wgrad(x0, dpre, dw1, db1, g1, none, direct, s);
  launch_conv_t(dpre, w1, nullptr, nullptr, d0, g1, s);                    // gradient at conv0's output
  if (relu) launch_relu_bwd(d0, x0, d0, M * F, s);
  launch_conv_wgrad(x, d0, dw0, db0, g0, none, direct, s);
  if (C == F) {
    launch_conv_t(d0, w0, nullptr, dpre, dx, g0, s);                       // identity skip: dx = d0 * W0^T + dpre
  } else {
    launch_conv_wgrad(x, dpre, dws, dbs, gs, none, direct, s);
    launch_conv_t(dpre, ws, nullptr, nullptr, tmp, gs, s);
    launch_conv_t(d0, w0, nullptr, tmp, dx, g0, s);
  }
  return hipGetLastError() == hipSuccess ? MVAE_OK : MVAE_E_HIP;
}

int mvae_profile_enable(int32_t on) {
  Profiler& p = profiler();
  p.on = on != 0;
  return MVAE_OK;
}

int64_t mvae_profile_report(char* buf, int64_t cap) {
  Profiler& p = profiler();
  if (hipDeviceSynchronize() != hipSuccess) return MVAE_E_HIP;
  struct Agg { int64_t n = 0; double ms = 0, bytes = 0, flops = 0; };
  std::vector<Agg> agg(p.tags.size());
  for (ProfRec& r : p.recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      Agg& a = agg[r.tag];
      a.n++; a.ms += ms; a.bytes += r.bytes; a.flops += r.flops;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  p.recs.clear();
  std::string out = "{";
  for (size_t i = 0; i < agg.size(); ++i) {
    if (!agg[i].n) continue;
    char line[256];
    snprintf(line, sizeof(line), "%s\"%s\": {\"count\": %lld, \"ms\": %.6f, \"bytes\": %.1f, \"flops\": %.1f}",
             out.size() > 1 ? ", " : "", p.tags[i].c_str(), (long long)agg[i].n, agg[i].ms, agg[i].bytes, agg[i].flops);
    out += line;
  }
  out += "}";
  if (buf && cap > 0) { strncpy(buf, out.c_str(), (size_t)cap - 1); buf[cap - 1] = 0; }
  return (int64_t)out.size();
}

int mvae_tensor_lookup(const mvae_handle* h, const char* name, float** ptr, int64_t* elems_per_image) {
  if (!h || !name) return MVAE_E_INVALID;
  auto it = h->tensors.find(name);
  if (it == h->tensors.end()) return MVAE_E_INVALID;
  if (ptr) *ptr = h->bound ? h->ws + it->second.first : nullptr;
  if (elems_per_image) *elems_per_image = it->second.second;
  return MVAE_OK;
}

int mvae_tensor_lookup2(const mvae_handle* h, const char* name, void** ptr, int64_t* elems_per_image, int32_t* dtype) {
  float* p = nullptr;
  const int rc = mvae_tensor_lookup(h, name, &p, elems_per_image);
  if (rc != MVAE_OK) return rc;
  if (ptr) *ptr = p;
  if (dtype) {
    auto it = h->tensor_dtype.find(name);
    *dtype = it == h->tensor_dtype.end() ? MVAE_ACT_F32 : it->second;
  }
  return MVAE_OK;
}

int mvae_scale_dtype(const mvae_handle* h, int32_t scale) {
  if (!h || scale < 0 || scale >= h->cfg.levels) return MVAE_E_INVALID;
  return h->scales[scale].bf ? MVAE_ACT_BF16 : MVAE_ACT_F32;
}

}  // extern "C"
