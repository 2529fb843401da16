// Network-level runtime for KeypointRegressionNet (+ RevGrad's domain classifier): layer graph, parameter/arena
// layout with the reference's state-dict names, activation workspace layout and the launch sequence of one
// forward and one backward pass.  Reference: park2019.py:101-165 (KRN), torchvision mobilenet_v2.features[:-1]
// (park2019.py:107-108), revgrad.py:58-96 (RevGrad), dann.py:68-100 (which passes call backward).
//
// Dataflow (training): every BatchNorm'd tensor lives in HBM only as the raw convolution output z plus
// per-channel batch sums; consumers normalise/activate on load.  Inverted-residual outputs with a skip connection,
// the tap after block 13 and the RouterV2 concat are the only materialised normalised tensors.  In backward every
// such tensor has g = dL/d(bn output) (activation mask applied) plus sum(g), sum(g*xhat); producers rebuild dz on load.
#include <string>
#include <vector>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include "common.h"

long long spb_wgrad_part_floats(int M, int K, int N);   // gemm_pw.hip

namespace {

struct PInfo { std::string name; int ndim; int shape[4]; long long off; long long numel; };
struct BNDef { int C; long long g_off, b_off, rm_off; int index; };
struct PWDef { int K, N; long long w_off, bias_off, wc_off, wct_off; };
struct DWDef { int C, stride; long long w_off; };
struct ActDef { int H, W, C, bn, act; float slope; };
struct MatDef { int H, W, C; };
struct Block { int t, cin, cout, stride, hid, Hin, Hout; bool res; PWDef E, P; DWDef D; int aE, aD, aP, matY;
               int matXe; };   // >= 0: slot for the expand convolution's operand round16(bn(block input)) when the expanded tensor is virtual

constexpr float kEps = 1e-5f;
constexpr float kMomentum = 0.1f;
constexpr int kIn = 224;
constexpr int kSplitBlock = 14;   // first inverted-residual block of the early gradient bucket (7x7 maps from here on)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct spb_krn {
  int nK = 0, J = 0, Jp = 0; bool dann = false;
  std::vector<PInfo> params, buffers;
  std::vector<std::string> bn_names;
  std::vector<BNDef> bns;
  std::vector<ActDef> acts;
  std::vector<MatDef> mats;
  Block blk[18];
  int aStem = -1; long long stem_w_off = 0;
  DWDef eD[4]; PWDef eP[4]; int aED[4], aEP[4];  // extras 0,1,3 are ConvDw; index 2 unused
  PWDef router; int aR = -1; int matCat = -1;
  long long head_w_off = 0, head_b_off = 0, head_wc_off = 0;
  PWDef dc0; long long dc3_w_off = 0, dc3_b_off = 0;
  long long n_params = 0, n_buffers = 0, wc_elems = 0;
  // data-parallel exchange: parameters [split_off, n_params) (blocks 14..17, extras, head, domain classifier: ~90 % of the
  // elements) are complete after a quarter of the backward pass; BatchNorm entries [split_bn, end) belong to them
  long long split_off = 0; int split_bn = 0;
  // bound state
  float* P = nullptr; float* G = nullptr; float* Bf = nullptr; long long* nbt = nullptr;
  char* wc = nullptr; spb_prep_entry_t* prep_d = nullptr; int n_prep = 0, n_prep_tiles = 0;
  std::vector<spb_prep_entry_t> prep;
  int dtype = -1;
  std::string prefix;
  bool det = false;   // reproducible mode (libspb_hip_det.so, spb_krn_set_det): exact accumulation + flush after every launch, one stream

  long long add_param(const std::string& name, std::vector<int> shape) {
    PInfo p; p.name = prefix + name; p.ndim = (int)shape.size(); p.numel = 1;
    for (int i = 0; i < 4; ++i) p.shape[i] = i < p.ndim ? shape[i] : 1;
    for (int d : shape) p.numel *= d;
    n_params = (long long)align_up((size_t)n_params, 4);  // keep every tensor 16-byte aligned in the arena
    p.off = n_params; n_params += p.numel;
    params.push_back(p);
    return p.off;
  }
  int add_bn(const std::string& name, int C) {
    BNDef b; b.C = C; b.index = (int)bns.size();
    b.g_off = add_param(name + ".weight", {C});
    b.b_off = add_param(name + ".bias", {C});
    PInfo rm; rm.name = prefix + name + ".running_mean"; rm.ndim = 1; rm.shape[0] = C; rm.shape[1] = rm.shape[2] = rm.shape[3] = 1;
    rm.numel = C; rm.off = n_buffers; n_buffers += C;
    PInfo rv = rm; rv.name = prefix + name + ".running_var"; rv.off = n_buffers; n_buffers += C;
    b.rm_off = rm.off;
    buffers.push_back(rm); buffers.push_back(rv);
    bn_names.push_back(prefix + name + ".num_batches_tracked");
    bns.push_back(b);
    return b.index;
  }
  int add_act(int H, int W, int C, int bn, int act, float slope = 0.f) {
    acts.push_back(ActDef{H, W, C, bn, act, slope});
    return (int)acts.size() - 1;
  }
  PWDef add_pw(const std::string& name, int K, int N, bool bias = false) {
    PWDef d; d.K = K; d.N = N; d.w_off = add_param(name + ".weight", {N, K, 1, 1});
    d.bias_off = bias ? add_param(name + ".bias", {N}) : -1;
    d.wc_off = (long long)align_up((size_t)wc_elems, 8); wc_elems = d.wc_off + (long long)N * K;
    d.wct_off = (long long)align_up((size_t)wc_elems, 8); wc_elems = d.wct_off + (long long)N * K;
    return d;
  }
  DWDef add_dw(const std::string& name, int C, int stride) {
    DWDef d; d.C = C; d.stride = stride; d.w_off = add_param(name + ".weight", {C, 1, 3, 3});
    return d;
  }
};

struct spb_krn_ctx {
  spb_krn* m = nullptr; int B = 0; char* ws = nullptr; size_t es = 2;
  std::vector<size_t> z_off, g_off, y_off;          // byte offsets in ws
  std::vector<size_t> sums_off, bsums_off;          // float offsets in the stats arena
  std::vector<int> R;
  size_t stats_off = 0, stats_floats = 0;
  size_t dcat_off = 0, dtap_off = 0, ddom_off = 0, dom1_off = 0, gdom_off = 0;
  size_t partial_off = 0, dout_off = 0, table_off = 0, dompool_off = 0;
  size_t junk_off = 0;              // FLOAT offset in the stats arena of 2 x 1280 floats nobody reads (batch sums of an evaluation-mode statistics
                                    // pass; inside the arena so that the reproducible build accumulates them exactly too: no counted miss)
  // weight-gradient partial sums (spb_red_job_t): one scratch slab per layer, keyed by the layer's weight offset in the arena
  struct PartSlab { long long key; size_t off; long long floats; };
  std::vector<PartSlab> parts;
  int S = 0;
  int last_training = 0;
  bool one_backward = false;      // the last forward was a fused-train-step forward (training & 16): exactly one backward follows it
  bool stats_clean = false;       // the batch-sum arena is all zeros (the last backward's bn_param_grads pass zeroed what it read): the next
                                  // training forward skips its memset
  bool pending_running = false;   // forward ran with training & 16: the running-statistics update rides with the next backward's side stream
  const float* x = nullptr;  // image of the last forward (needed by the stem weight gradient)
  // live per-launch timing (HIP events on the launch stream) with the algorithmic bytes of each launch
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;   // pairs
  std::vector<int> prof_cat;
  std::vector<double> prof_bytes, prof_flops;
  int prof_n = 0;
  // weight-gradient GEMMs run on a side stream: they only feed the optimizer, so they overlap the dgrad / depthwise
  // chain of the earlier layers (fork/join with events, capturable into a hipGraph)
  hipStream_t side = nullptr;
  bool side_on = true;
  std::vector<hipEvent_t> fork_ev;
  hipEvent_t join_ev = nullptr;
  hipEvent_t prep_ev = nullptr;     // weight copies refreshed on the side stream (spb_krn_forward, training & 4)
  hipEvent_t router_ev = nullptr;   // the RouterV2 branch (forward: 1x1 + reorg into the concat; backward: its input gradient) finished on the side stream
  hipEvent_t bucket_ev = nullptr;   // recorded when the gradients of [split_off, n_params) are final
  bool bucket_recorded = false;
  bool bucket_on = false;           // spb_krn_ctx_set_bucket: single-GPU runs skip the mid-backward join
  int n_fork = 0;
  // forks ordered by a device word instead of an event (see fork_gate_kernel): the word and the serial number of the last fork
  unsigned* fork_flag = nullptr;
  unsigned fork_serial = 0;
  unsigned* fork_poison = nullptr;  // host word (device-mapped) a gate raises when it gives up instead of trapping: checked at every entry
  bool det = false;                 // spb_krn_ctx_set_det: this context's batch-sum arena has an exact-accumulation shadow
  const float* loss_scale = nullptr;   // spb_krn_ctx_set_loss_scale: device scalar multiplied onto the upstream gradient (float16 recipe)
  ~spb_krn_ctx() {
    for (hipEvent_t e : fork_ev) hipEventDestroy(e);
    for (hipEvent_t e : prof_ev) hipEventDestroy(e);
    if (join_ev) hipEventDestroy(join_ev);
    if (bucket_ev) hipEventDestroy(bucket_ev);
    if (prep_ev) hipEventDestroy(prep_ev);
    if (router_ev) hipEventDestroy(router_ev);
    if (side) hipStreamDestroy(side);
    if (fork_flag) hipFree(fork_flag);
    spb_fork_poison_free(fork_poison);
  }
  bool poisoned() const { return fork_poison && *reinterpret_cast<volatile unsigned*>(fork_poison) != 0; }
};

enum ProfCat { PC_STEM_FWD = 0, PC_PW_FWD, PC_DW_FWD, PC_BN_APPLY, PC_HEAD_FWD, PC_BN_UPDATE, PC_HEAD_BWD, PC_PW_DGRAD,
               PC_PW_WGRAD, PC_DW_DGRAD, PC_DW_WGRAD, PC_BN_BWD_PREP, PC_STEM_WGRAD, PC_BN_PARAM_GRADS, PC_DOMAIN,
               PC_WEIGHT_PREP, PC_PW_BWD_FUSED, PC_PW_STATS, PC_COUNT };
static const char* kProfNames[PC_COUNT] = {"stem_fwd", "pw_gemm_fwd", "dw_fwd", "bn_apply", "head_fwd", "bn_running_update",
                                           "head_bwd", "pw_gemm_dgrad", "pw_wgrad", "dw_dgrad", "dw_wgrad", "bn_bwd_prep",
                                           "stem_wgrad", "bn_param_grads", "domain_head", "weight_prep", "pw_bwd_fused", "pw_stats"};

namespace {

void build_model(spb_krn* m, int nK, bool dann) {
  m->nK = nK; m->J = 2 * nK; m->Jp = (m->J + 15) / 16 * 16; m->dann = dann;
  m->prefix = dann ? "net." : "";
  // ---- torchvision mobilenet_v2.features[:-1]
  m->stem_w_off = m->add_param("base.0.0.weight", {32, 3, 3, 3});
  m->aStem = m->add_act(112, 112, 32, m->add_bn("base.0.1", 32), SPB_ACT_RELU6);
  const int cfg[7][4] = {{1, 16, 1, 1}, {6, 24, 2, 2}, {6, 32, 3, 2}, {6, 64, 4, 2}, {6, 96, 3, 1}, {6, 160, 3, 2}, {6, 320, 1, 1}};
  int cin = 32, H = 112, k = 1;
  for (int s = 0; s < 7; ++s)
    for (int i = 0; i < cfg[s][2]; ++i, ++k) {
      Block& b = m->blk[k];
      b.t = cfg[s][0]; b.cin = cin; b.cout = cfg[s][1]; b.stride = i == 0 ? cfg[s][3] : 1;
      b.hid = cin * b.t; b.Hin = H; b.Hout = (H - 1) / b.stride + 1;
      b.res = (b.stride == 1 && b.cin == b.cout);
      const std::string pre = "base." + std::to_string(k) + ".conv.";
      int idx = 0;
      b.aE = -1;
      if (b.t != 1) {
        b.E = m->add_pw(pre + "0.0", cin, b.hid);
        b.aE = m->add_act(H, H, b.hid, m->add_bn(pre + "0.1", b.hid), SPB_ACT_RELU6);
        idx = 1;
        if (k == kSplitBlock) { m->split_off = b.E.w_off; m->split_bn = m->acts[b.aE].bn; }
      }
      b.D = m->add_dw(pre + std::to_string(idx) + ".0", b.hid, b.stride);
      b.aD = m->add_act(b.Hout, b.Hout, b.hid, m->add_bn(pre + std::to_string(idx) + ".1", b.hid), SPB_ACT_RELU6);
      b.P = m->add_pw(pre + std::to_string(idx + 1), b.hid, b.cout);
      b.aP = m->add_act(b.Hout, b.Hout, b.cout, m->add_bn(pre + std::to_string(idx + 2), b.cout), SPB_ACT_NONE);
      b.matY = -1; b.matXe = -1;
      if (b.res) { m->mats.push_back(MatDef{b.Hout, b.Hout, b.cout}); b.matY = (int)m->mats.size() - 1; }
      // candidates for a virtual expanded tensor (Runner::virt) whose input is not a materialised block output: the statistics pass of the
      // expand convolution writes its operand here once, the recomputing kernels read it as it is (26 MB at bs=48 for blocks 2 and 3)
      if (b.t != 1 && b.cin <= 32 && H >= 28 && k >= 2 && !m->blk[k - 1].res) {
        m->mats.push_back(MatDef{H, H, cin}); b.matXe = (int)m->mats.size() - 1;
      }
      cin = b.cout; H = b.Hout;
    }
  // ---- extras (park2019.py:113-118): ConvDw(320,1024), ConvDw(1024,1024), RouterV2(96,64), ConvDw(1280,1024)
  const int ein[4] = {320, 1024, 0, 1280};
  for (int e = 0; e < 4; ++e) {
    const std::string pre = "extras." + std::to_string(e) + ".conv.";
    if (e == 2) {
      m->router = m->add_pw(pre + "0", 96, 64);
      m->aR = m->add_act(14, 14, 64, m->add_bn(pre + "1", 64), SPB_ACT_LEAKY, 0.2f);
      continue;
    }
    m->eD[e] = m->add_dw(pre + "0", ein[e], 1);
    m->aED[e] = m->add_act(7, 7, ein[e], m->add_bn(pre + "1", ein[e]), SPB_ACT_RELU);
    m->eP[e] = m->add_pw(pre + "3", ein[e], 1024);
    m->aEP[e] = m->add_act(7, 7, 1024, m->add_bn(pre + "4", 1024), SPB_ACT_RELU);
  }
  m->mats.push_back(MatDef{7, 7, 1280}); m->matCat = (int)m->mats.size() - 1;
  // ---- head (park2019.py:121)
  m->head_w_off = m->add_param("head.0.weight", {m->J, 1024, 7, 7});
  m->head_b_off = m->add_param("head.0.bias", {m->J});
  m->head_wc_off = (long long)align_up((size_t)m->wc_elems, 8);
  m->wc_elems = m->head_wc_off + (long long)m->Jp * 49 * 1024;
  // ---- RevGrad domain classifier (revgrad.py:75-80)
  if (dann) {
    m->prefix = "";
    m->dc0 = m->add_pw("domain_classifier.0", 320, 1280, true);
    m->dc3_w_off = m->add_param("domain_classifier.3.weight", {1, 1280, 1, 1});
    m->dc3_b_off = m->add_param("domain_classifier.3.bias", {1});
  }
  m->n_params = (long long)align_up((size_t)m->n_params, 4);
  // ---- weight-prep table: W copy, W^T copy per pointwise conv; permuted head weight
  int tile = 0;
  auto add_prep = [&](long long src, long long dst, int rows, int cols, int mode, int batch) {
    spb_prep_entry_t e; e.src_off = src; e.dst_off = dst; e.rows = rows; e.cols = cols; e.mode = mode;
    e.aux = 0; e.aux2 = 0; e.tile0 = tile;
    tile += batch * ((rows + 31) / 32) * ((cols + 31) / 32);
    m->prep.push_back(e);
  };
  auto add_pw_prep = [&](const PWDef& d) {
    add_prep(d.w_off, d.wc_off, d.N, d.K, 0, 1);
    add_prep(d.w_off, d.wct_off, d.N, d.K, 1, 1);
  };
  for (int kk = 1; kk <= 17; ++kk) { if (m->blk[kk].t != 1) add_pw_prep(m->blk[kk].E); add_pw_prep(m->blk[kk].P); }
  add_pw_prep(m->eP[0]); add_pw_prep(m->eP[1]); add_pw_prep(m->router); add_pw_prep(m->eP[3]);
  add_prep(m->head_w_off, m->head_wc_off, 1024, 49, 2, m->J);
  if (dann) add_pw_prep(m->dc0);
  m->n_prep = (int)m->prep.size(); m->n_prep_tiles = tile;
}

// ---------------------------------------------------------------------------------------------------------------
// An operand as a consumer sees it: tensor + its BatchNorm / activation.  join: the residual sum of an inverted-residual block that
// has not been materialised yet -- bn(ptr) + bn2(ptr2) -- which the next expand convolution forms while loading and writes to `mat`

// ---- reproducible mode (common.h, -DSPB_DET) --------------------------------------------------------------------------
// Host registry of the (float range -> shadow windows) regions; every translation unit of the twin library holds its own device copy
// of the table (no relocatable device code), uploaded through the setters the files registered at load time.
#ifdef SPB_DET
namespace {
std::vector<spb_det_tu_fn>& det_tus() { static std::vector<spb_det_tu_fn> v; return v; }
spb_det_table_t& det_table() { static spb_det_table_t t; return t; }
int det_upload() {
  if (hipDeviceSynchronize() != hipSuccess) return SPB_E_STATE;   // no kernel may be reading the old table
  for (spb_det_tu_fn f : det_tus()) (void)f(&det_table(), nullptr);   // (a file without float atomics has no table symbol: ignored)
  return 0;
}
__global__ void det_flush_kernel(float* __restrict__ f, long long* __restrict__ sh, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const longlong2 a = *reinterpret_cast<const longlong2*>(sh + 4 * i), b = *reinterpret_cast<const longlong2*>(sh + 4 * i + 2);
    if ((a.x | a.y | b.x | b.y) != 0) {
      const double v = (((double)b.y * 0x1p-120 + (double)b.x * 0x1p-80) + (double)a.y * 0x1p-40) + (double)a.x;
      f[i] += (float)v;
      *reinterpret_cast<longlong2*>(sh + 4 * i) = longlong2{0, 0};
      *reinterpret_cast<longlong2*>(sh + 4 * i + 2) = longlong2{0, 0};
    }
  }
}
// fold the shadow windows of the region that starts at `lo` into its float slots (and clear them), on `st`
int det_flush(const float* lo, hipStream_t st) {
  const spb_det_table_t& t = det_table();
  for (int i = 0; i < t.n; ++i)
    if (t.r[i].lo == lo) {
      const long long n = t.r[i].hi - t.r[i].lo;
      const int blocks = (int)std::min<long long>((n + 255) / 256, 2048);
      hipLaunchKernelGGL(det_flush_kernel, dim3(blocks), dim3(256), 0, st, const_cast<float*>(lo), t.r[i].shadow, n);
      return hipGetLastError() == hipSuccess ? 0 : SPB_E_STATE;
    }
  return SPB_E_STATE;
}
}  // namespace
extern "C" __attribute__((visibility("hidden"))) void spb_det_register_tu(spb_det_tu_fn f) { det_tus().push_back(f); }
extern "C" int spb_det_available(void) { return 1; }
extern "C" int spb_det_register(const float* lo, long long n_floats, long long* shadow) {
  if (!lo || n_floats <= 0 || !shadow) return SPB_E_ARG;
  spb_det_table_t& t = det_table();
  // a region that overlaps the new one is STALE: its memory went back to the caller's allocator and came out again (an engine that is
  // garbage but not collected yet -- its owner unregisters with spb_det_unregister_if, which then finds nothing).  Left in the table it
  // would catch the new owner's atomics first and accumulate them into a dead shadow: two "reproducible" runs of one process differed
  // (tests/test_surface_gpu.py::test_deterministic_module_generic_autograd_path, round 6)
  for (int i = 0; i < t.n;) {
    if (t.r[i].lo < lo + n_floats && lo < t.r[i].hi) { for (int j = i + 1; j < t.n; ++j) t.r[j - 1] = t.r[j]; t.n--; }
    else ++i;
  }
  if (t.n >= SPB_DET_MAX_REGIONS) return SPB_E_STATE;
  t.r[t.n].lo = lo; t.r[t.n].hi = lo + n_floats; t.r[t.n].shadow = shadow; t.n++;
  return det_upload();
}
extern "C" int spb_det_unregister(const float* lo) {
  spb_det_table_t& t = det_table();
  for (int i = 0; i < t.n; ++i)
    if (t.r[i].lo == lo) { for (int j = i + 1; j < t.n; ++j) t.r[j - 1] = t.r[j]; t.n--; return det_upload(); }
  return SPB_E_ARG;
}
// the same, but only if the region at `lo` still accumulates into `shadow` (the caller's own registration, not a later owner of the address)
extern "C" int spb_det_unregister_if(const float* lo, const long long* shadow) {
  spb_det_table_t& t = det_table();
  for (int i = 0; i < t.n; ++i)
    if (t.r[i].lo == lo && t.r[i].shadow == shadow) { for (int j = i + 1; j < t.n; ++j) t.r[j - 1] = t.r[j]; t.n--; return det_upload(); }
  return 0;
}
extern "C" int spb_det_flush(const float* lo, spb_stream_t stream) { return det_flush(lo, (hipStream_t)stream); }
extern "C" long long spb_det_misses(void) {   // float atomics that found no region since the last call (synchronises the device)
  if (hipDeviceSynchronize() != hipSuccess) return SPB_E_STATE;
  unsigned long long m = 0;
  for (spb_det_tu_fn f : det_tus()) (void)f(nullptr, &m);
  return (long long)m;
}
#else
extern "C" int spb_det_available(void) { return 0; }
extern "C" int spb_det_register(const float*, long long, long long*) { return SPB_E_UNSUPPORTED; }
extern "C" int spb_det_unregister(const float*) { return SPB_E_UNSUPPORTED; }
extern "C" int spb_det_unregister_if(const float*, const long long*) { return SPB_E_UNSUPPORTED; }
extern "C" int spb_det_flush(const float*, spb_stream_t) { return SPB_E_UNSUPPORTED; }
extern "C" long long spb_det_misses(void) { return SPB_E_UNSUPPORTED; }
#endif
// Switch a bound engine / a context to the reproducible mode.  The caller has registered the gradient arena (spb_krn_bind's grads) and
// the context's batch-sum arena (spb_krn_ctx_stats) with spb_det_register; the plan then keeps every launch on the one stream and folds
// the shadows after each launch.  Only libspb_hip_det.so accepts it.
extern "C" int spb_krn_set_det(spb_krn_t* m, int on) {
  if (!m) return SPB_E_ARG;
#ifdef SPB_DET
  m->det = on != 0; return 0;
#else
  return on ? SPB_E_UNSUPPORTED : 0;
#endif
}
extern "C" int spb_krn_ctx_set_det(spb_krn_ctx_t* c, int on) {
  if (!c) return SPB_E_ARG;
#ifdef SPB_DET
  c->det = on != 0; return 0;
#else
  return on ? SPB_E_UNSUPPORTED : 0;
#endif
}
// Introspection for the parity tests: where BatchNorm'd tensor `a` (raw convolution output z, NHWC, compute dtype) and its backward
// companion g live in the context's workspace, and which BatchNorm (index into spb_krn_bn_name) normalises it.
extern "C" int spb_krn_num_acts(const spb_krn_t* m) { return m ? (int)m->acts.size() : SPB_E_ARG; }
extern "C" int spb_krn_ctx_act_info(const spb_krn_ctx_t* c, int a, spb_act_info_t* out) {
  if (!c || !out || a < 0 || a >= (int)c->m->acts.size()) return SPB_E_ARG;
  const ActDef& d = c->m->acts[a];
  out->z_off = (long long)c->z_off[a]; out->g_off = (long long)c->g_off[a];
  out->H = d.H; out->W = d.W; out->C = d.C; out->bn_index = d.bn;
  return 0;
}
// the context's batch-sum arena (what spb_det_register shadows): pointer and length in floats
extern "C" int spb_krn_ctx_stats(spb_krn_ctx_t* c, float** ptr, long long* n_floats) {
  if (!c || !ptr || !n_floats) return SPB_E_ARG;
  *ptr = reinterpret_cast<float*>(c->ws + c->stats_off); *n_floats = (long long)c->stats_floats;
  return 0;
}

struct Src { const void* ptr; spb_bnref_t ref; const void* ptr2 = nullptr; spb_bnref_t ref2 = spb_bnref_t(); void* mat = nullptr; };

static int g_side_wgrad = 1;
// Pointwise weight gradients are queued and handed to the side stream right before the next depthwise backward kernel (one
// fork per inverted-residual block instead of two, and they then run beside a memory-bound kernel rather than beside the
// input-gradient GEMMs); at most g_wgrad_batch are held back.  Measured: fork per GEMM 3.52 ms, per 3 GEMMs 3.43 ms, at the
// depthwise kernels 3.37 ms per step.  spb_debug_set_wgrad_batch(n): n > 0 plain batches of n, n < 0 flush at depthwise, cap -n.
static int g_skip_side = 0;            // TIMING EXPERIMENT ONLY (spb_debug_set_launch_events(2|4)): 2 drops the pointwise, 4 the depthwise side-stream weight gradients
static int g_launch_events = 1;        // fork on the completion event of the preceding GEMM launch instead of an event record
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
// bits 3-4: fork mode (0 default = 2, 1 -> events, 2 -> flag stored by a one-wave kernel)
extern "C" int spb_debug_set_launch_events(int on);
#endif
// ---- forks without a packet on the launch queue: the weight gradients are handed to the side stream through a device word and a
// one-wave gate kernel instead of an event (the mechanism and its measurements: elemwise.hip, "stream forks without events")
static int g_fork_mode = 2;            // 0 events, 1 flag stored by a one-wave kernel, 2 flag stored at the entry of the next depthwise kernel
#ifdef SPB_TUNING
extern "C" int spb_debug_set_launch_events(int on) {
  g_launch_events = on & 1; g_skip_side = on & 6;
  const int fm = (on >> 3) & 3;
  g_fork_mode = fm == 0 ? 2 : fm - 1;
  return 0;
}
#endif
static int g_side_priority = 0;
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_side_priority(int on);
#endif
// Flags of the events that order the two streams of ONE device (fork / join / weight-prep).  Without hipEventDisableSystemFence the
// runtime gives the recording packet -- for a fork that is the input-gradient GEMM the event rides on (hipExtLaunchKernelGGL stop
// event) -- a SYSTEM-scope release: every XCD's L2 written back and invalidated for the host's benefit, a ~4.8 us bubble on the
// launch stream in front of each of the step's 22 depthwise backward kernels (profiles/r5_krn_chain.txt, `gap_us`).  Both sides of
// these events are queues of the same device: a device-scope release is all they need.  (bucket_ev, which a communication stream
// waits on before RCCL reads the gradients, keeps the system fence.)
static unsigned g_stream_event_flags = hipEventDisableTiming | hipEventDisableSystemFence;
#ifdef SPB_TUNING   // tuning build only: bit 0 = lowest stream priority for the side stream, bit 1 = system-fenced events (A/B)
extern "C" int spb_debug_set_side_priority(int on) {
  g_side_priority = on & 1;
  g_stream_event_flags = (on & 2) ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence);
  return 0;
}
#endif
static int g_wgrad_flush_at_dw = 1;
static int g_wgrad_min_flush = 1;      // flush at a depthwise kernel only with at least this many queued
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_wgrad_min_flush(int n);
#endif
static int g_wgrad_batch = 8;
// Depthwise layers on maps up to this many columns wide run their input gradient alone on the launch stream and send the weight
// gradient to the side stream with the pointwise ones: in the plane kernels (dwconv_plane.hip, maps up to 14x14) the weight
// gradient triples the instruction count (72 partials per lane, each reduced across lanes) on a path that is bound by VALU issue.
// Measured in the step (round 3): 0 -> 3.32 ms, 14 -> 3.22, 28 -> 3.23, 56 -> 3.22 (3.20 vs 3.19 after the later changes).
// (Round 2, row-unit kernels on every map: the split was slower, 3.33 vs 3.29 ms.)
static int g_dw_split_hw = 112;   // end of round 3 (with the resident-grid cap on the GEMMs): 56 -> 3.096 ms, 112 -> 3.084
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_dw_split(int hw) { g_dw_split_hw = hw; return 0; }
#endif
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_wgrad_batch(int n) {
  g_wgrad_flush_at_dw = n < 0;
  if (n < 0) n = -n;
  g_wgrad_batch = n < 1 ? 1 : n;
  return 0;
}
#endif
static long long g_replica_min_rows = 32768;   // BN-sum replicas (8) from this many rows up; spb_debug_set_replica_rows
static int g_replica_mid = 4;   // replicas for tensors with g_replica_mid_rows <= rows < g_replica_min_rows (the 14x14 maps at bs=48).  Round 2 (tiled
                                // kernels only): 8 replicas = producers -1 us, consumers +2 us, net zero -> 1.  Round 3: the row-slab GEMM (gemm_rs.hip)
                                // has 294 workgroups per channel address instead of 147 and same-address f32 atomics serialise at ~25 ns: with one
                                // replica its launches end 6 us after their last store (3.100 ms per step); 2 -> 3.061, 3 -> 3.037, 4 -> 3.027,
                                // 6 -> 3.029, 8 -> 3.033 (without the row-slab kernel the replica count changes nothing: 3.062)
static long long g_replica_mid_rows = 4096;
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_replica_rows(long long rows) {
  if (rows < -100) { g_replica_mid_rows = -rows; return 0; }   // below -100: the row count from which the mid-size replica count applies
  if (rows < 0) { g_replica_mid = (int)(-rows > SPB_MAX_REPLICAS ? SPB_MAX_REPLICAS : -rows); return 0; }   // negative: set the mid-size replica count instead
  g_replica_min_rows = rows;
  return 0;
}
#endif
static int g_flush_after_dw = 0;   // measured: 3.23 ms with the side-stream batch enqueued after the launch-stream kernel, 3.19 before it
// (sign convention of spb_debug_set_wgrad_min_flush: -1 -> before (default), -2 -> after)
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_wgrad_min_flush(int n) {
  if (n < 0) { g_flush_after_dw = n == -2; return 0; }
  g_wgrad_min_flush = n < 1 ? 1 : n;
  return 0;
}   // spb_debug_set_wgrad_min_flush(-1) restores flush-before-launch
#endif
static int g_wgrad_parts = 0;     // 1: pointwise weight gradients as partial sums + one reduce launch per batch instead of f32 atomics
                                  // (order-deterministic; measured 3.25 vs 3.18 ms per step: the slab traffic and the extra launches cost more
                                  // than the atomics they replace) -- spb_debug_set_wgrad_parts
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_wgrad_parts(int on) { g_wgrad_parts = on; return 0; }
#endif
static int g_domain_tail_rows = 1;   // row-parallel forward of the domain classifier's pooled tail (spb_debug_set_domain_tail_rows(0): the walking kernel)
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_domain_tail_rows(int on) { g_domain_tail_rows = on; return 0; }
#endif
static int g_join_fused = 1;      // residual adds folded into the next expand convolution (spb_debug_set_join_fused)
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_join_fused(int on) { g_join_fused = on; return 0; }
#endif
// Round 6: expand -> depthwise fusion with recompute.  For the inverted-residual blocks whose expand convolution has a short reduction on a
// large map (blocks 2-4: 16 -> 96 at 112x112, 24 -> 144 at 56x56 twice) the 6x-expanded tensor is VIRTUAL: the expand launch only
// accumulates its batch sums (spb_pwconv_gemm with Y == NULL), the depthwise forward / input-gradient / weight-gradient kernels and the
// fused pointwise backward rebuild the values they need on the matrix cores from the 16 / 24-channel block input (spb_dw_args_t::Xe,
// spb_pwbwd_args_t::Zn == NULL).  0.81 GB of the step's HBM traffic (four passes over 202 MB) become four passes over 34 MB, and the
// expanded activation is no longer rounded to 16 bits on the way (the 112x112 layers are where bf16 storage hurts the gradient most).
// 16-bit storage only; spb_debug_set_fuse_expand(min_width): 0 = off.
static int g_fuse_expand_min_hw = 56;
#ifdef SPB_TUNING
extern "C" int spb_debug_set_fuse_expand(int min_width) { g_fuse_expand_min_hw = min_width <= 0 ? (1 << 30) : min_width; return 0; }
#endif
// Round 6: the RouterV2 branch (park2019.py:60-80,116: 1x1 conv 96 -> 64 on block 13's output, reorg into the concat) hangs off the main
// chain -- forward it is needed only by extras[3], backward its input gradient only by block 14's expand convolution -- so its launches
// (forward: GEMM + the reorg bn_apply, 17 us; backward: bn_bwd_prep + the input-gradient GEMM, 20 us) run on the side stream beside the 7x7
// chain instead of inside it; one fork and one event wait each way.  spb_debug_set_router_side(0): in the chain, as before.
static int g_router_side = 1;
#ifdef SPB_TUNING
extern "C" int spb_debug_set_router_side(int on) { g_router_side = on != 0; return 0; }
#endif
static int g_fused_pw_bwd = 1;
static long long g_fused_pw_bwd_min_m = 100000;  // spb_debug_set_fused_pw_bwd(v > 1): fused kernel from v rows up.  The 28x28 layer
                                                 // (M = 37632, 144 -> 32) took 45 us fused for 26 MB; as GEMM + side-stream weight gradient the step is 12 us shorter
struct Runner {
  spb_krn_ctx* c; spb_krn* m; hipStream_t st; int dt; int err = 0;
  bool flag_forks = false;   // forks through the context's device word (fork_gate_kernel) instead of events
  Runner(spb_krn_ctx* c_, hipStream_t s) : c(c_), m(c_->m), st(s), dt(c_->m->dtype) {
    flag_forks = g_fork_mode != 0 && c->fork_flag && c->side && spb_fork_by_word(st, c->side);
  }
  void ok(int e) {
    if (e != 0 && err == 0) err = e;
#ifdef SPB_DET   // reproducible mode: fold the exact batch-sum shadows into the float slots the next launch reads
    if (c->det) { const int f = det_flush(stats(), st); if (f != 0 && err == 0) err = f; }
#endif
  }
  void det_flush_grads() {
#ifdef SPB_DET
    if (m->det && m->G) { const int f = det_flush(m->G, st); if (f != 0 && err == 0) err = f; }
#endif
  }
  // ---- live timing: tic(category, algorithmic bytes, flops) ... toc() around one launch
  void tic(int cat, double bytes, double flops = 0.0) {
    if (!c->prof_on) return;
    if ((size_t)(2 * c->prof_n + 2) > c->prof_ev.size()) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      c->prof_ev.push_back(a); c->prof_ev.push_back(b);
      c->prof_cat.push_back(0); c->prof_bytes.push_back(0); c->prof_flops.push_back(0);
    }
    c->prof_cat[c->prof_n] = cat; c->prof_bytes[c->prof_n] = bytes; c->prof_flops[c->prof_n] = flops;
    hipEventRecord(c->prof_ev[2 * c->prof_n], st);
  }
  void toc() {
    if (!c->prof_on) return;
    hipEventRecord(c->prof_ev[2 * c->prof_n + 1], st);
    c->prof_n++;
  }
  double es() const { return (double)c->es; }
  double elems(int a) const { return (double)M(a) * m->acts[a].C; }

  float* stats() const { return reinterpret_cast<float*>(c->ws + c->stats_off); }
  void* z(int a) const { return c->ws + c->z_off[a]; }
  void* g(int a) const { return c->ws + c->g_off[a]; }
  void* y(int mt) const { return c->ws + c->y_off[mt]; }
  float* sums(int a) const { return stats() + c->sums_off[a]; }
  float* bsums(int a) const { return stats() + c->bsums_off[a]; }
  int M(int a) const { return c->B * m->acts[a].H * m->acts[a].W; }
  const void* wc(long long off) const { return m->wc + (size_t)off * c->es; }
  // scratch slab of a layer's weight-gradient partial sums (keyed by its weight offset); floats = 0: none
  float* part_slab(long long key, long long* floats) const {
    for (const auto& p : c->parts)
      if (p.key == key) { *floats = p.floats; return reinterpret_cast<float*>(c->ws + p.off); }
    *floats = 0;
    return nullptr;
  }

  spb_bnref_t ref(int a, bool training) const {
    const ActDef& d = m->acts[a]; const BNDef& b = m->bns[d.bn];
    spb_bnref_t r; r.sums = sums(a); r.gamma = m->P + b.g_off; r.beta = m->P + b.b_off; r.bsums = bsums(a);
    r.inv_n = 1.f / (float)M(a); r.eps = kEps; r.slope = d.slope; r.C = d.C;
    r.R = training ? c->R[a] : 1; r.act = d.act; r.moments = training ? 0 : 1;
    return r;
  }
  static spb_bnref_t ident(int C) {
    spb_bnref_t r; std::memset(&r, 0, sizeof(r)); r.C = C; r.R = 1; r.inv_n = 1.f; r.eps = kEps; return r;
  }
  Src src_act(int a, bool tr) const { return Src{z(a), ref(a, tr)}; }
  Src src_mat(int mt) const { return Src{y(mt), ident(m->mats[mt].C)}; }
  Src block_out(int k, bool tr) const {  // output of inverted-residual block k as a consumer sees it
    const Block& b = m->blk[k];
    return b.res ? src_mat(b.matY) : src_act(b.aP, tr);
  }
  // the expanded tensor of block k exists only as its batch sums (see g_fuse_expand_min_hw): its consumers recompute it
  bool virt(int k) const {
    const Block& b = m->blk[k];
    return dt == SPB_BF16 && b.t != 1 && b.cin <= 32 && (b.cin & 7) == 0 && b.Hin >= g_fuse_expand_min_hw && b.Hin >= 28 &&
           g_fused_pw_bwd && (long long)c->B * b.Hin * b.Hin >= g_fused_pw_bwd_min_m;   // (its pointwise backward must be the recomputing fused kernel)
  }
  void set_xe(spb_dw_args_t& d, const PWDef& E, const Src& x) const {
    d.Xe = x.ptr; d.We = wc(E.wc_off); d.xe = x.ref; d.Ce = E.K;
  }

  // ---- forward pieces
  void pw_fwd(const PWDef& L, const Src& in, int aout, bool tr) {
    spb_gemm_args_t g; std::memset(&g, 0, sizeof(g));
    g.A = in.ptr; g.Bw = wc(L.wc_off); g.Y = z(aout); g.pro = in.ref; g.M = M(aout); g.K = L.K; g.N = L.N;
    g.pro_mode = 1; g.out_scale = 1.f;
    if (in.mat) { g.pro_mode = 3; g.A2 = in.ptr2; g.pro2 = in.ref2; g.Ymat = in.mat; }   // residual join of the previous block
    if (tr) { g.epi_mode = 1; g.osums = sums(aout); g.oR = c->R[aout]; } else { g.epi_mode = 0; g.oR = 1; }
    tic(PC_PW_FWD, ((double)g.M * ((in.mat ? 3 : 1) * L.K + L.N) + (double)L.K * L.N) * es(), 2.0 * g.M * L.K * L.N);
    ok(spb_pwconv_gemm(dt, &g, st));
    toc();
  }
  // what is left of a virtual block's expand convolution: its batch sums (training) and, if the operand is a residual join, the
  // materialised block output the join writes
  void pw_stats(const PWDef& L, const Src& in, int aout, bool tr, void* xmat) {
    spb_gemm_args_t g; std::memset(&g, 0, sizeof(g));
    g.A = in.ptr; g.Bw = wc(L.wc_off); g.Y = nullptr; g.pro = in.ref; g.M = M(aout); g.K = L.K; g.N = L.N;
    g.pro_mode = 1; g.out_scale = 1.f; g.epi_mode = 1; g.Ymat = xmat;     // xmat: the operand round16(bn(in)), for the recomputing kernels
    if (in.mat) { g.pro_mode = 3; g.A2 = in.ptr2; g.pro2 = in.ref2; g.Ymat = in.mat; }
    // evaluation: the sums are not wanted (the arena holds the running statistics): a scratch row behind the context's tables takes them
    g.osums = tr ? sums(aout) : stats() + c->junk_off; g.oR = tr ? c->R[aout] : 1;
    tic(PC_PW_STATS, ((double)g.M * ((in.mat ? 3 : 1) + (xmat ? 1 : 0)) * L.K + (double)L.K * L.N) * es(), 2.0 * g.M * L.K * L.N);
    ok(spb_pwconv_gemm(dt, &g, st));
    toc();
  }
  // the tensor the recomputing kernels of virtual block k read as the expand convolution's operand (identity BatchNorm on it)
  Src xe_src(int k) const {
    const Block& b = m->blk[k];
    return b.matXe >= 0 ? src_mat(b.matXe) : src_mat(m->blk[k - 1].matY);
  }
  // E / x: the expand convolution in front of the layer and ITS input, when the expanded tensor is virtual (`in` then only names the
  // BatchNorm + activation that sits on it)
  void dw_fwd(const DWDef& L, const Src& in, int Hin, int aout, bool tr, const PWDef* E = nullptr, const Src* x = nullptr) {
    spb_dw_args_t d; std::memset(&d, 0, sizeof(d));
    d.X = in.ptr; d.Wd = m->P + L.w_off; d.Y = z(aout); d.pro = in.ref; d.B = c->B; d.H = Hin; d.W = Hin; d.C = L.C;
    d.stride = L.stride; d.epi_mode = tr ? 1 : 0; d.osums = sums(aout); d.oR = c->R[aout];
    if (E) { set_xe(d, *E, *x); d.X = nullptr; }
    tic(PC_DW_FWD, ((double)c->B * Hin * Hin * (E ? E->K : L.C) + elems(aout)) * es(), 18.0 * elems(aout) + (E ? 2.0 * c->B * Hin * Hin * E->K * L.C : 0.0));
    ok(spb_dwconv_fwd(dt, &d, st));
    toc();
  }
  // ---- backward pieces.  `atgt` is the Act whose g / bsums the input gradient lands in (-1: plain output to `plain`)
  // before_dw: the next launch is a depthwise backward kernel, where the queued weight gradients fork off (flush_wgrads)
  void pw_bwd(const PWDef& L, const Src& in, int aout, int atgt, void* plain, const void* res, float plain_scale = 1.f,
              bool before_dw = false, bool virt_out = false) {
    // wide, shallow layers (the 112x112 / 56x56 maps): one fused pass over g and z for both gradients
    if (g_fused_pw_bwd && dt == SPB_BF16 && atgt >= 0 && M(aout) >= g_fused_pw_bwd_min_m) {
      spb_pwbwd_args_t f; std::memset(&f, 0, sizeof(f));
      f.G = this->g(aout); f.Zn = virt_out ? nullptr : z(aout); f.Wt = wc(L.wct_off); f.X = in.ptr; f.Zout = z(atgt); f.res = res;
      f.Y = this->g(atgt); f.dW = m->G + L.w_off; f.osums = bsums(atgt); f.pro_dz = ref(aout, true); f.pro_a = in.ref;
      f.epi = ref(atgt, true); f.M = M(aout); f.K = L.K; f.N = L.N; f.oR = c->R[atgt];
      const double mn = (double)f.M * L.N, mk = (double)f.M * L.K;
      tic(PC_PW_BWD_FUSED, ((L.K <= 32 ? 1 : 2) * mn + (2 + (f.X != f.Zout ? 1 : 0) + (res ? 1 : 0)) * mk + (double)L.K * L.N) * es() + 4.0 * L.K * L.N,
          4.0 * f.M * L.K * L.N);     // (K <= 32: z is recomputed, not read)
      const int e = spb_pwconv_bwd_fused(dt, &f, st);
      toc();
      // (ok(0): in the reproducible build this is also the point where the launch's exact batch sums are folded into the float slots the
      // NEXT launch reads.  Round 6: the early return skipped it, so the depthwise backward of blocks 1-3 rebuilt dz from backward sums
      // that were still zero -- the reproducible bf16 pass was a different (wrong) gradient: BatchNorm weights of the first layers 8-17x
      // too large, end-to-end cosine 0.981 where the float-atomic passes had 0.990.  f32 never takes this branch.)
      if (e == 0) { ok(0); return; }
      if (virt_out) { ok(e); return; }   // no other kernel can stand in: z does not exist
      if (e != SPB_E_UNSUPPORTED) { ok(e); return; }
      if (c->prof_on) c->prof_n--;   // no fused instance for this shape: drop the empty timing record
    }
    // weight gradient first: on the side stream it then runs beside this layer's input-gradient GEMM
    spb_wgrad_args_t w; std::memset(&w, 0, sizeof(w));
    w.G = this->g(aout); w.Zn = z(aout); w.X = in.ptr; w.dW = m->G + L.w_off; w.pro_dz = ref(aout, true);
    w.pro_a = in.ref; w.M = M(aout); w.K = L.K; w.N = L.N;
    if (g_wgrad_parts) w.part = part_slab(L.w_off, &w.part_cap);
    tic(PC_PW_WGRAD, ((double)w.M * (2 * L.N + L.K)) * es() + 4.0 * L.K * L.N, 2.0 * w.M * L.K * L.N);
    queue_wgrad(w);
    toc();
    spb_gemm_args_t g; std::memset(&g, 0, sizeof(g));
    g.A = this->g(aout); g.A2 = z(aout); g.Bw = wc(L.wct_off); g.pro = ref(aout, true);
    g.M = M(aout); g.K = L.N; g.N = L.K; g.pro_mode = 2; g.out_scale = plain_scale;
    if (atgt >= 0) {
      g.Y = this->g(atgt); g.Zout = z(atgt); g.epi = ref(atgt, true); g.osums = bsums(atgt); g.oR = c->R[atgt];
      g.res = res; g.epi_mode = 2;
    } else { g.Y = plain; g.epi_mode = 0; g.oR = 1; }
    {
      const double mn = (double)g.M * L.N, mk = (double)g.M * L.K;
      tic(PC_PW_DGRAD, (2 * mn + (atgt >= 0 ? 2 : 1) * mk + (res ? mk : 0) + (double)L.K * L.N) * es(), 2.0 * g.M * L.K * L.N);
    }
    // this launch's own completion event: what the queued weight gradients wait for (see flush_wgrads)
    launch_ev = nullptr;
    if (g_launch_events && !flag_forks && before_dw && (!pend.empty() || head_pending || g_dw_split_hw > 0) && side_usable()) {
      launch_ev = next_event(); g.stop_event = launch_ev;
    }
    ok(spb_pwconv_gemm(dt, &g, st));
    toc();
  }
  // Stream for work that only the optimizer consumes.  Forked from the launch stream at the current point (everything
  // the weight gradient reads -- g, z, bsums of the output, the forward activations -- is final before pw_bwd starts);
  // joined at the end of spb_krn_backward.  With the profiler on everything stays on the launch stream.
  hipEvent_t launch_ev = nullptr;   // completion event attached to the last input-gradient GEMM launch (or null)
  hipEvent_t next_event() {
    if ((size_t)c->n_fork >= c->fork_ev.size()) {
      hipEvent_t e; hipEventCreateWithFlags(&e, g_stream_event_flags);
      c->fork_ev.push_back(e);
    }
    return c->fork_ev[c->n_fork++];
  }
  bool side_usable() const { return !(c->prof_on || !c->side || !c->side_on || !g_side_wgrad || c->det || m->det); }
  // Launches of a branch that hangs off the chain (the RouterV2 branch): `body` runs with `st` switched to the side stream, forked from
  // the launch stream's current point; c->router_ev marks their end.  Returns false (and runs nothing) where the side stream cannot be
  // used: the caller then issues the same launches in the chain.
  template <typename F> bool branch_on_side(F body) {
    if (!g_router_side || !side_usable()) return false;
    hipStream_t keep = st;
    st = side_stream();
    body();
    hipEventRecord(c->router_ev, st);
    st = keep;
    return true;
  }
  void gate_side(unsigned serial) {   // the side stream waits until the context's fork word reaches `serial`
    spb_fork_gate(c->fork_flag, serial, c->side, spb_fork_poison_dev(c->fork_poison));
    forked = true;
  }
  hipStream_t side_stream() {
    if (!side_usable()) return st;
    if (flag_forks) {   // the word is stored by a one-wave kernel behind everything the launch stream holds so far
      const unsigned serial = ++c->fork_serial;
      spb_fork_store(c->fork_flag, serial, st);
      gate_side(serial);
      return c->side;
    }
    hipEvent_t e = next_event();
    hipEventRecord(e, st);
    hipStreamWaitEvent(c->side, e, 0);
    forked = true;
    return c->side;
  }
  // Weight-gradient GEMMs are handed to the side stream in batches: every fork costs the launch stream an event record, and
  // the kernel trace shows ~7.5 us of dispatch bubble on the launch stream for each (30 forks = 0.22 ms of a 3.5 ms step).
  // Their inputs (g, z, batch sums of the output tensor, the forward activations) are not overwritten during backward, so a
  // weight gradient may start any time after its layer's pw_bwd was reached.
  std::vector<spb_wgrad_args_t> pend;
  std::vector<spb_red_job_t> jobs;      // reduce jobs of the partial sums stored since the last run_jobs
  void run_jobs(hipStream_t s) {
    if (!jobs.empty()) ok(spb_partial_reduce(jobs.data(), (int)jobs.size(), s));
    jobs.clear();
  }
  void launch_wgrad(spb_wgrad_args_t w, hipStream_t s) {
    spb_red_job_t job; std::memset(&job, 0, sizeof(job));
    w.job_out = &job;
    ok(spb_pwconv_wgrad(dt, &w, s));
    if (job.nparts > 0) jobs.push_back(job);
  }
  void queue_wgrad(const spb_wgrad_args_t& w) {
    if (!side_usable()) { launch_wgrad(w, st); run_jobs(st); return; }
    pend.push_back(w);
    if ((int)pend.size() >= g_wgrad_batch) flush_wgrads();
  }
  bool head_pending = false;
  spb_head_bwd_args_t head_args;
  void queue_head_wgrad(const spb_head_bwd_args_t& h) {
    if (!side_usable()) { ok(spb_head_bwd(dt, &h, st)); return; }
    head_args = h; head_pending = true;
  }
  std::vector<spb_dw_args_t> pend_dw;   // depthwise weight gradients of the small maps (see dw_bwd)
  void flush_wgrads() {
    if (pend.empty() && pend_dw.empty() && !head_pending) { gated = false; return; }
    hipStream_t s;
    if (gated) { gated = false; s = c->side; }   // dw_bwd already put this batch's gate on the side stream
    else if (launch_ev) {   // everything the queued GEMMs read was final before that launch: wait for it, record nothing
      hipStreamWaitEvent(c->side, launch_ev, 0);
      launch_ev = nullptr; forked = true; s = c->side;
    } else s = side_stream();            // one event record for the whole batch
    if (c->pending_running) {   // deferred BatchNorm running-statistics update of the forward pass (spb_krn_forward, training & 16)
      ok(spb_bn_running_update(reinterpret_cast<const spb_bnupd_entry_t*>(c->ws + c->table_off), (int)m->bns.size(), stats(),
                               m->Bf, m->nbt, kMomentum, s));
      c->pending_running = false;
    }
    if (head_pending) { ok(spb_head_bwd(dt, &head_args, s)); head_pending = false; }
    if (!(g_skip_side & 4)) for (const spb_dw_args_t& d : pend_dw) ok(spb_dwconv_wgrad(dt, &d, s));
    pend_dw.clear();
    if (!(g_skip_side & 2)) for (const spb_wgrad_args_t& w : pend) launch_wgrad(w, s);
    pend.clear();
    run_jobs(s);   // one reduce launch for the batch
  }
  void join_side() {
    flush_wgrads();
    det_flush_grads();
    if (!forked) return;
    hipEventRecord(c->join_ev, c->side);
    hipStreamWaitEvent(st, c->join_ev, 0);
    forked = false;
  }
  bool forked = false;
  bool gated = false;
  void dw_bwd(const DWDef& L, const Src& in, int Hin, int aout, int atgt, void* plain, const void* res, const PWDef* E = nullptr,
              const Src* x = nullptr) {
    spb_dw_args_t d; std::memset(&d, 0, sizeof(d));
    d.X = this->g(aout); d.X2 = z(aout); d.Xin = in.ptr; d.Wd = m->P + L.w_off; d.dW = m->G + L.w_off;
    d.pro = ref(aout, true); d.pro_in = in.ref; d.B = c->B; d.H = Hin; d.W = Hin; d.C = L.C; d.stride = L.stride;
    d.Zout = in.ptr; d.epi = in.ref;  // the convolution's input and its BN/activation (== ref(atgt) when atgt >= 0)
    if (E) { set_xe(d, *E, *x); d.Xin = nullptr; d.Zout = nullptr; }   // virtual input: recomputed from the expand convolution's operand
    // On the small maps the weight gradient goes to the side stream with the pointwise ones (everything it reads -- g, z and the
    // batch sums of this layer's output, the forward input -- is final and never rewritten during backward) and the launch
    // stream runs the input gradient alone (spb_debug_set_dw_split).
    // (the instrumented pass measures the SAME two kernels, back to back on the one stream, not the fused one the product no longer runs)
    const bool split = (side_usable() || c->prof_on) && Hin <= g_dw_split_hw;
    spb_dw_args_t dwg = d;
    if (split) {
      if (!c->prof_on) pend_dw.push_back(d);
      d.dW = nullptr;
    }
    // one fused pass: input gradient (+ activation mask / BN sums of the input-side tensor) and weight gradient
    if (atgt >= 0) {
      d.Y = this->g(atgt); d.osums = bsums(atgt); d.oR = c->R[atgt]; d.res = res; d.epi_mode = 2;
    } else { d.Y = plain; d.epi_mode = 0; d.oR = 1; }
    const double nin = (double)c->B * Hin * Hin * L.C, nout = elems(aout);
    const double nzin = E ? (double)c->B * Hin * Hin * E->K : nin;       // elements read for the conv input (mask / a)
    const bool flush_due = g_wgrad_flush_at_dw && (int)(pend.size() + pend_dw.size()) >= g_wgrad_min_flush &&
                           (!pend.empty() || !pend_dw.empty() || head_pending);
    // fork through the device word: this kernel's first thread stores the serial (it runs behind a barrier bit: everything the queued
    // weight gradients read is complete by then), the gate and the batch follow it onto the side stream -- the kernel that stores the
    // word is always enqueued BEFORE the gate that waits for it (streams may share a hardware queue)
    const bool entry_fork = flag_forks && g_fork_mode == 2 && flush_due && side_usable();
    if (entry_fork) { d.entry_flag = c->fork_flag; d.entry_val = ++c->fork_serial; }
    if (!entry_fork && !g_flush_after_dw && flush_due) flush_wgrads();
    tic(PC_DW_DGRAD, (2 * nout + nin + nzin + (res ? nin : 0)) * es(), 36.0 * nout);
    const int dwe = spb_dwconv_dgrad(dt, &d, st);
    ok(dwe);
    toc();
    if (entry_fork) {
      // the entry returned an error before (or instead of) launching: nobody will store the serial -- publish it with the one-wave
      // kernel so that the gate passes and the error comes back as an error, not as a side stream spinning into its time-out
      if (dwe != 0) spb_fork_store(c->fork_flag, d.entry_val, st);
      gate_side(d.entry_val); gated = true; flush_wgrads();
    }
    if (split && c->prof_on) {
      tic(PC_DW_WGRAD, (2 * nout + nzin) * es() + 36.0 * L.C, 18.0 * nout);
      ok(spb_dwconv_wgrad(dt, &dwg, st));
      toc();
    }
    // the queued weight gradients run beside this memory-bound kernel.  They are handed to the side stream AFTER the launch
    // stream got its kernel: the host is only a bounded number of packets ahead of the GPU, and a burst of ~10 side-stream
    // launches in front of the next launch-stream kernel showed up as 30-50 us holes in the launch queue (round-3 trace)
    if (!entry_fork && g_flush_after_dw && flush_due) flush_wgrads();
    launch_ev = nullptr;   // the event belongs to the launch before the depthwise kernel: nothing queued later may fork on it
  }
};

__global__ void domain_tail_fwd_kernel(const void* D1, int dtype, const float* w3, const float* b3, float* pooled,
                                       float* logits, int HW, int C) {
  // logits[b] = b3 + sum_c w3[c] * mean_hw D1[b,hw,c]      (AvgPool2d(7) + Conv2d(1280,1,1), revgrad.py:78-79)
  __shared__ float red[4];
  const int b = blockIdx.x;
  float s = 0.f;
  for (int cidx = threadIdx.x; cidx < C; cidx += 256) {
    float a = 0.f;
    for (int p = 0; p < HW; ++p) {
      const size_t o = ((size_t)b * HW + p) * C + cidx;
      a += dtype == SPB_BF16 ? bf2f(reinterpret_cast<const bf16_t*>(D1)[o]) : reinterpret_cast<const float*>(D1)[o];
    }
    a /= (float)HW;
    pooled[(size_t)b * C + cidx] = a;
    s += a * w3[cidx];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) logits[b] = red[0] + red[1] + red[2] + red[3] + b3[0];
}

// Row-parallel form (round 3): grid (B, C / 256).  A workgroup owns 256 channels (32 lanes x 8-channel vectors) of one image and 8
// row lanes; the 7 rows of a lane are all in flight together, the row lanes meet in LDS, and the image's logit collects one f32 atomic
// per channel slab (logits are zeroed by the caller's memset; slab 0 adds the bias).  The kernel above walks 5 channels x 49 rows per
// thread in dependent steps on B workgroups (named in rounds 2 and 3 as the low-parallelism launch of the DANN forward).
template <typename T>
__global__ __launch_bounds__(256) void domain_tail_fwd_rows_kernel(const T* __restrict__ D1, const float* __restrict__ w3, const float* __restrict__ b3,
                                                                   float* __restrict__ pooled, float* __restrict__ logits, int HW, int C) {
  __shared__ float red[8][257];
  __shared__ float wred[4];
  const int b = blockIdx.x, cb = blockIdx.y * 256, t = threadIdx.x, v = t & 31, sub = t >> 5;
  const int c0 = cb + v * 8;
  const bool cok = c0 < C;
  const int cc = cok ? c0 : 0;
  constexpr int U = 7;                                   // rows per lane: 8 x 7 = 56 >= 49
  Raw8<T> raw[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int r = sub + 8 * u;
    raw[u] = ldraw<T>(D1 + ((size_t)b * HW + (r < HW ? r : HW - 1)) * C + cc);
  }
  float a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (sub + 8 * u < HW) {
      float d[8];
      cvt8(raw[u], d);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += d[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[sub][v * 8 + j] = a[j];
  __syncthreads();
  float s = 0.f;
  {
    const int c = cb + t;
    if (c < C) {
      float p = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) p += red[i][t];
      p /= (float)HW;
      pooled[(size_t)b * C + c] = p;
      s = p * w3[c];
    }
  }
  s = wave_sum(s);
  if ((t & 63) == 0) wred[t >> 6] = s;
  __syncthreads();
  if (t == 0) atomicAdd(logits + b, wred[0] + wred[1] + wred[2] + wred[3] + (blockIdx.y == 0 ? b3[0] : 0.f));
}

// Backward of AvgPool2d(7) + Conv2d(1280,1,1) behind the ReLU of domain_classifier.0 (revgrad.py:75-80):
//   dD1[b,hw,c] = dlogit[b]*w3[c]/HW * (D1 > 0)      dbias0[c] += sum_{b,hw} dD1      dw3[c] += sum_b dlogit[b]*pooled[b,c]
// grid (C/64, row ranges of 128): a workgroup owns 64 channels (8 lanes x 8-channel 16-byte vectors) x 32 row lanes with
// DTB_U rows of each lane in flight, so a launch is ONE memory round trip on some hundred workgroups.  (Round 1..2: one
// thread per channel on C/256 = 5 workgroups walking all B*49 rows -- 2352 dependent iterations, ~0.6-1.2 ms per launch on
// the critical path of both DANN passes.)  The element values are computed exactly as before; only the order of the
// per-channel bias sum changed (32-lane LDS reduction, then one f32 atomic per channel and row range).
constexpr int DTB_U = 4, DTB_ROWS = 32 * DTB_U;
template <typename T> __device__ __forceinline__ void dtb_store8(T* p, const float g[8]);
template <> __device__ __forceinline__ void dtb_store8<bf16_t>(bf16_t* p, const float g[8]) {
  uint4 u;
  u.x = pack_bf16x2(g[0], g[1]); u.y = pack_bf16x2(g[2], g[3]); u.z = pack_bf16x2(g[4], g[5]); u.w = pack_bf16x2(g[6], g[7]);
  *reinterpret_cast<uint4*>(p) = u;
}
template <> __device__ __forceinline__ void dtb_store8<float>(float* p, const float g[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(g[0], g[1], g[2], g[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(g[4], g[5], g[6], g[7]);
}

template <typename T>
__global__ __launch_bounds__(256) void domain_tail_bwd_kernel(const T* D1, T* Gd, const float* w3, const float* pooled,
                                                              const float* dlogit, float* dw3, float* db3, float* dbias0,
                                                              int B, int HW, int C) {
  __shared__ float red[32][65];
  const int t = threadIdx.x, v = t & 7, sub = t >> 3;
  const int cb = blockIdx.x * 64, c0 = cb + v * 8;
  const bool cok = c0 < C;                       // C % 8 == 0 (the 1x1 convolution above requires it)
  const int cc = cok ? c0 : 0;
  const int rows = B * HW;
  const int r0 = blockIdx.y * DTB_ROWS, r1 = min(rows, r0 + DTB_ROWS);
  const float hw = (float)HW;
  float w[8], gb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { w[j] = w3[cc + j]; gb[j] = 0.f; }
  Raw8<T> raw[DTB_U];
  float dl[DTB_U];
#pragma unroll
  for (int u = 0; u < DTB_U; ++u) {              // every load first, on clamped rows
    const int ru = r0 + sub + 32 * u;
    const int rc = ru < r1 ? ru : r1 - 1;
    raw[u] = ldraw<T>(D1 + (size_t)rc * C + cc);
    dl[u] = dlogit[rc / HW];
  }
#pragma unroll
  for (int u = 0; u < DTB_U; ++u) {
    const int ru = r0 + sub + 32 * u;
    if (ru < r1 && cok) {
      float d[8], g[8];
      cvt8(raw[u], d);
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = d[j] > 0.f ? dl[u] * w[j] / hw : 0.f;
      rnd8<T>(g);
      dtb_store8<T>(Gd + (size_t)ru * C + c0, g);
#pragma unroll
      for (int j = 0; j < 8; ++j) gb[j] += g[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[sub][v * 8 + j] = gb[j];
  __syncthreads();
  if (t < 64 && cb + t < C) {
    const int c = cb + t;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += red[i][t];
    atomicAdd(dbias0 + c, s);
    if (blockIdx.y == 0) {                       // single writer per channel
      float gw = 0.f;
      for (int b = 0; b < B; ++b) gw += dlogit[b] * pooled[(size_t)b * C + c];
      dw3[c] += gw;
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && t == 64) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dlogit[b];
    db3[0] += s;
  }
}

__global__ void domain_tail_bwd_ref_kernel(const void* D1, void* Gd, int dtype, const float* w3, const float* pooled,
                                       const float* dlogit, float* dw3, float* db3, float* dbias0, int B, int HW, int C) {
  // REFERENCE for the row-parallel kernel above (SPB_DOMAIN_TAIL_REF=1; the rounds-1..2 kernel: serial over rows, exact same element
  // arithmetic).  One block per channel slab of 256: dD1[b,hw,c] = dlogit[b]*w3[c]/HW * (D1>0);  dw3[c] += sum_b dlogit[b]*pooled[b,c]
  const int cidx = blockIdx.x * 256 + threadIdx.x;
  if (cidx < C) {
    float gw = 0.f, gb0 = 0.f;
    const float w = w3[cidx];
    for (int b = 0; b < B; ++b) {
      const float dl = dlogit[b];
      gw += dl * pooled[(size_t)b * C + cidx];
      const float up = dl * w / (float)HW;
      for (int p = 0; p < HW; ++p) {
        const size_t o = ((size_t)b * HW + p) * C + cidx;
        if (dtype == SPB_BF16) {
          const float d1 = bf2f(reinterpret_cast<const bf16_t*>(D1)[o]);
          const float gv = bf2f(f2bf(d1 > 0.f ? up : 0.f));
          reinterpret_cast<bf16_t*>(Gd)[o] = f2bf(gv);
          gb0 += gv;
        } else {
          const float d1 = reinterpret_cast<const float*>(D1)[o];
          const float gv = d1 > 0.f ? up : 0.f;
          reinterpret_cast<float*>(Gd)[o] = gv;
          gb0 += gv;
        }
      }
    }
    dw3[cidx] += gw;
    dbias0[cidx] += gb0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dlogit[b];
    db3[0] += s;
  }
}


__global__ void bce_logits_kernel(const float* logits, float label, int B, float* loss, float* dlogit, float gscale) {
  // binary_cross_entropy_with_logits(reduction='mean') against a constant label (dann.py:85-92)
  __shared__ float red[256];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float x = logits[b];
    s += fmaxf(x, 0.f) - x * label + log1pf(expf(-fabsf(x)));
    const float sig = 1.f / (1.f + expf(-x));
    if (dlogit) dlogit[b] = gscale * (sig - label) / (float)B;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0 && loss) loss[0] = red[0] / (float)B;
}

}  // namespace

// =================================================================================================================
extern "C" int spb_krn_create(int num_keypoints, int dann, spb_krn_t** out) {
  if (!out || num_keypoints <= 0 || num_keypoints > 16) return SPB_E_ARG;
  spb_krn* m = new spb_krn();
  build_model(m, num_keypoints, dann != 0);
  *out = m;
  return 0;
}
extern "C" void spb_krn_destroy(spb_krn_t* m) { delete m; }
extern "C" int spb_krn_num_params(const spb_krn_t* m) { return (int)m->params.size(); }
extern "C" int spb_krn_num_buffers(const spb_krn_t* m) { return (int)m->buffers.size(); }
extern "C" int spb_krn_num_bn(const spb_krn_t* m) { return (int)m->bns.size(); }
static int fill_info(const PInfo& p, spb_tensor_info_t* o) {
  std::memset(o, 0, sizeof(*o));
  std::snprintf(o->name, sizeof(o->name), "%s", p.name.c_str());
  o->offset = p.off; o->numel = p.numel; o->ndim = p.ndim;
  for (int i = 0; i < 4; ++i) o->shape[i] = p.shape[i];
  return 0;
}
extern "C" int spb_krn_param_info(const spb_krn_t* m, int i, spb_tensor_info_t* o) {
  if (!m || !o || i < 0 || i >= (int)m->params.size()) return SPB_E_ARG;
  return fill_info(m->params[i], o);
}
extern "C" int spb_krn_buffer_info(const spb_krn_t* m, int i, spb_tensor_info_t* o) {
  if (!m || !o || i < 0 || i >= (int)m->buffers.size()) return SPB_E_ARG;
  return fill_info(m->buffers[i], o);
}
extern "C" int spb_krn_bn_name(const spb_krn_t* m, int i, char* out96) {
  if (!m || !out96 || i < 0 || i >= (int)m->bn_names.size()) return SPB_E_ARG;
  std::snprintf(out96, 96, "%s", m->bn_names[i].c_str());
  return 0;
}
extern "C" long long spb_krn_param_numel(const spb_krn_t* m) { return m->n_params; }
extern "C" long long spb_krn_buffer_numel(const spb_krn_t* m) { return m->n_buffers; }
extern "C" long long spb_krn_wcompute_bytes(const spb_krn_t* m, int dtype) {
  return (long long)align_up((size_t)m->wc_elems * (dtype == SPB_BF16 ? 2 : 4), 256);
}
extern "C" long long spb_krn_tables_bytes(const spb_krn_t* m) {
  return (long long)align_up(m->prep.size() * sizeof(spb_prep_entry_t), 256);
}

extern "C" int spb_krn_bind(spb_krn_t* m, float* params, float* grads, float* buffers, long long* nbt, void* wcompute,
                            void* tables_dev, int dtype) {
  if (!m || !params || !buffers || !wcompute || !tables_dev) return SPB_E_ARG;
  if (dtype != SPB_F32 && dtype != SPB_BF16) return SPB_E_ARG;
  m->P = params; m->G = grads; m->Bf = buffers; m->nbt = nbt; m->wc = (char*)wcompute; m->dtype = dtype;
  m->prep_d = (spb_prep_entry_t*)tables_dev;
  hipError_t e = hipMemcpy(m->prep_d, m->prep.data(), m->prep.size() * sizeof(spb_prep_entry_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) return (int)e;
  e = hipMemset(wcompute, 0, (size_t)spb_krn_wcompute_bytes(m, dtype));  // padded head rows stay zero
  if (e != hipSuccess) return (int)e;
  return 0;
}

extern "C" long long spb_krn_weight_prep_bytes(const spb_krn_t* m) {  // algorithmic bytes of one prepare_weights call
  long long b = 0;
  for (const auto& e : m->prep) b += (long long)e.rows * e.cols * (e.mode == 2 ? m->J : 1) * (4 + (m->dtype == SPB_BF16 ? 2 : 4));
  return b;
}
extern "C" int spb_krn_prepare_weights(spb_krn_t* m, spb_stream_t stream) {
  if (!m || m->dtype < 0) return SPB_E_STATE;
  return spb_weight_prep(m->dtype, m->prep_d, m->n_prep, m->n_prep_tiles, m->P, m->wc, stream);
}

// ---- context layout -------------------------------------------------------------------------------------------
static void layout_ctx(const spb_krn* m, int B, int dtype, spb_krn_ctx* c, size_t* total) {
  const size_t es = dtype == SPB_BF16 ? 2 : 4;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const int nA = (int)m->acts.size();
  if (c) { c->es = es; c->z_off.resize(nA); c->g_off.resize(nA); c->sums_off.resize(nA); c->bsums_off.resize(nA); c->R.resize(nA); c->y_off.resize(m->mats.size()); }
  size_t table = take(m->bns.size() * sizeof(spb_bnupd_entry_t));
  size_t sf = 0;
  std::vector<size_t> so(nA), bo(nA); std::vector<int> Rv(nA);
  for (int a = 0; a < nA; ++a) {
    const ActDef& d = m->acts[a];
    const long long Mrows = (long long)B * d.H * d.W;
    Rv[a] = Mrows >= g_replica_min_rows ? 8 : (Mrows >= g_replica_mid_rows ? g_replica_mid : 1);
    so[a] = sf; sf += (size_t)Rv[a] * 2 * d.C;
    bo[a] = sf; sf += (size_t)Rv[a] * 2 * d.C;
  }
  const size_t junk = sf; sf += 2 * 1280;
  size_t stats = take(sf * sizeof(float));
  for (int a = 0; a < nA; ++a) {
    const ActDef& d = m->acts[a];
    const size_t bytes = (size_t)B * d.H * d.W * d.C * es;
    size_t zo = take(bytes), go = take(bytes);
    if (c) { c->z_off[a] = zo; c->g_off[a] = go; c->sums_off[a] = so[a]; c->bsums_off[a] = bo[a]; c->R[a] = Rv[a]; }
  }
  for (size_t i = 0; i < m->mats.size(); ++i) {
    size_t yo = take((size_t)B * m->mats[i].H * m->mats[i].W * m->mats[i].C * es);
    if (c) c->y_off[i] = yo;
  }
  const int S = 512;  // split-K waves of the head GEMM (128 workgroups; 256: 64 workgroups on 256 CUs)
  size_t dcat = take((size_t)B * 49 * 1280 * es);
  size_t dtap = take((size_t)B * 14 * 14 * 96 * es);
  size_t ddom = take((size_t)B * 49 * 320 * es);
  size_t dom1 = take((size_t)B * 49 * 1280 * es);
  size_t gdom = take((size_t)B * 49 * 1280 * es);
  size_t dompool = take((size_t)B * 1280 * sizeof(float));
  size_t partial = take((size_t)S * B * m->Jp * sizeof(float));
  size_t dout = take((size_t)B * m->J * sizeof(float));
  // scratch slabs of the weight-gradient partial sums: pointwise (row splits, or one part per workgroup of the fused
  // backward kernel on the 112x112 / 56x56 maps) and depthwise (one part per workgroup of a channel quad)
  std::vector<spb_krn_ctx::PartSlab> slabs;
  auto pw_slab = [&](const PWDef& d, int H) {
    const long long Mr = (long long)B * H * H;
    long long f = spb_wgrad_part_floats((int)Mr, d.K, d.N);
    if (Mr >= 32768 && 512LL * d.N * d.K > f) f = 512LL * d.N * d.K;
    if (f > 0) slabs.push_back(spb_krn_ctx::PartSlab{d.w_off, take((size_t)f * sizeof(float)), f});
  };
  auto dw_slab = [&](const DWDef& d) {
    const long long f = 520LL * d.C * 9;
    slabs.push_back(spb_krn_ctx::PartSlab{d.w_off, take((size_t)f * sizeof(float)), f});
  };
  for (int k = 1; k <= 17; ++k) {
    const Block& b = m->blk[k];
    if (b.t != 1) pw_slab(b.E, b.Hin);
    dw_slab(b.D);
    pw_slab(b.P, b.Hout);
  }
  for (int e = 0; e < 4; ++e) {
    if (e == 2) { pw_slab(m->router, 14); continue; }
    dw_slab(m->eD[e]); pw_slab(m->eP[e], 7);
  }
  if (m->dann) pw_slab(m->dc0, 7);
  if (c) {
    c->table_off = table; c->stats_off = stats; c->stats_floats = sf; c->dcat_off = dcat; c->dtap_off = dtap;
    c->ddom_off = ddom; c->dom1_off = dom1; c->gdom_off = gdom; c->dompool_off = dompool; c->partial_off = partial;
    c->junk_off = junk;
    c->dout_off = dout; c->S = S;
    c->parts = slabs;
  }
  *total = off;
}

extern "C" long long spb_krn_ctx_bytes(const spb_krn_t* m, int batch, int dtype) {
  if (!m || batch <= 0) return SPB_E_ARG;
  size_t total = 0;
  layout_ctx(m, batch, dtype, nullptr, &total);
  return (long long)total;
}

extern "C" int spb_krn_ctx_create(spb_krn_t* m, int batch, void* workspace, spb_krn_ctx_t** out) {
  if (!m || !workspace || !out || batch <= 0) return SPB_E_ARG;
  if (m->dtype < 0) return SPB_E_STATE;
  spb_krn_ctx* c = new spb_krn_ctx();
  c->m = m; c->B = batch; c->ws = (char*)workspace;
  size_t total = 0;
  layout_ctx(m, batch, m->dtype, c, &total);
  // BN maintenance table (stats offsets are per context)
  std::vector<spb_bnupd_entry_t> tab(m->bns.size());
  for (size_t a = 0; a < m->acts.size(); ++a) {
    const ActDef& d = m->acts[a];
    const BNDef& b = m->bns[d.bn];
    spb_bnupd_entry_t& e = tab[b.index];
    const double n = (double)batch * d.H * d.W;
    e.sums_off = (long long)c->sums_off[a]; e.bsums_off = (long long)c->bsums_off[a]; e.rm_off = b.rm_off;
    e.gamma_off = b.g_off; e.beta_off = b.b_off; e.C = d.C; e.R = c->R[a]; e.bn_index = b.index;
    e.inv_n = (float)(1.0 / n); e.unbias = n > 1 ? (float)(n / (n - 1.0)) : 1.f;
  }
  hipError_t e = hipMemcpy(c->ws + c->table_off, tab.data(), tab.size() * sizeof(spb_bnupd_entry_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) { delete c; return (int)e; }
  // the head's split-K partial workspace ends in the ticket word of its last-arriver reduction (spb_head_fwd): zero once, every call restores it
  e = hipMemset(c->ws + c->partial_off, 0, (size_t)c->S * batch * m->Jp * sizeof(float));
  if (e != hipSuccess) { delete c; return (int)e; }
  // side stream + events are created here, outside any stream capture
  {
    // side stream of the weight gradients.  g_side_priority (spb_debug_set_side_priority, A/B): 0 default priority, 1 the lowest the device
    // offers -- the launch stream's short, latency-bound kernels then get their workgroups placed ahead of the side stream's
    int least = 0, greatest = 0;
    hipError_t se = hipErrorUnknown;
    if (g_side_priority && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess)
      se = hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, least);
    if (se != hipSuccess && hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) c->side = nullptr;
  }
  // (the fill has to have executed before the first gate runs on the non-blocking side stream: see spb_fork_create in elemwise.hip)
  if (hipMalloc(&c->fork_flag, 256) != hipSuccess || hipMemset(c->fork_flag, 0, 256) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
    delete c; return SPB_E_STATE;
  }
  c->fork_poison = spb_fork_poison_alloc();   // (the fork self-test runs at the first pass, on the launch stream and this side stream: elemwise.hip)
  if (hipEventCreateWithFlags(&c->join_ev, g_stream_event_flags) != hipSuccess) { delete c; return SPB_E_STATE; }
  if (hipEventCreateWithFlags(&c->bucket_ev, hipEventDisableTiming) != hipSuccess) { delete c; return SPB_E_STATE; }
  if (hipEventCreateWithFlags(&c->prep_ev, g_stream_event_flags) != hipSuccess) { delete c; return SPB_E_STATE; }
  if (hipEventCreateWithFlags(&c->router_ev, g_stream_event_flags) != hipSuccess) { delete c; return SPB_E_STATE; }
  for (int i = 0; i < 64; ++i) {
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, g_stream_event_flags) != hipSuccess) break;
    c->fork_ev.push_back(ev);
  }
  *out = c;
  return 0;
}
extern "C" void spb_krn_ctx_destroy(spb_krn_ctx_t* c) { delete c; }
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_side_wgrad(int on) { g_side_wgrad = on; return 0; }
#endif
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_fused_pw_bwd(int on) { g_fused_pw_bwd = on != 0; if (on > 1) g_fused_pw_bwd_min_m = on; return 0; }
#endif
extern "C" int spb_krn_ctx_set_loss_scale(spb_krn_ctx_t* c, const float* scale) {
  if (!c) return SPB_E_ARG;
  c->loss_scale = scale;
  return 0;
}
extern "C" int spb_krn_ctx_set_side_stream(spb_krn_ctx_t* c, int on) {
  if (!c) return SPB_E_ARG;
  c->side_on = on != 0;
  return 0;
}

// ---- forward ---------------------------------------------------------------------------------------------------
extern "C" int spb_krn_forward(spb_krn_ctx_t* c, const float* x, const float* target, int training, float* pred,
                               float* scalars, float* domain_logits, spb_stream_t stream) {
  if (!c || !x || !pred) return SPB_E_ARG;
  if (target && !scalars) return SPB_E_ARG;
  if (c->poisoned()) return SPB_E_TIMEOUT;
  spb_krn* m = c->m;
  hipStream_t st = (hipStream_t)stream;
  Runner r(c, st);
  // training & 4: refresh the compute-dtype weight copies first (what spb_krn_prepare_weights does) -- on the side stream,
  // beside the stem and the first depthwise layer, which read the f32 parameters; joined before the first 1x1 convolution
  // training & 8: zero the bound gradient arena (optimizer.zero_grad()) on the side stream too -- nothing touches it before the
  // backward pass; training & 16: the BatchNorm running-statistics update is left to the next spb_krn_backward on this context,
  // which runs it on its side stream (no kernel of the step reads the running statistics)
  const bool prep = (training & 4) != 0, zero_g = (training & 8) != 0 && m->G != nullptr, defer_run = (training & 16) != 0;
  training &= 3;
  if (c->pending_running) {   // a deferred update that no backward pass picked up: apply it before the sums are overwritten
    r.ok(spb_bn_running_update(reinterpret_cast<const spb_bnupd_entry_t*>(c->ws + c->table_off), (int)m->bns.size(), r.stats(),
                               m->Bf, m->nbt, kMomentum, stream));
    c->pending_running = false;
  }
  bool prep_wait = false;
  if (prep || zero_g) {
    if (r.side_usable() && c->fork_ev.size() > 0) {
      if (r.flag_forks) (void)r.side_stream();   // (the previous step's optimizer is the producer: no kernel of this pass can carry the word)
      else {
        hipEventRecord(c->fork_ev[0], st);
        hipStreamWaitEvent(c->side, c->fork_ev[0], 0);
      }
      if (prep) r.ok(spb_weight_prep(m->dtype, m->prep_d, m->n_prep, m->n_prep_tiles, m->P, m->wc, c->side));
      if (zero_g && hipMemsetAsync(m->G, 0, (size_t)m->n_params * sizeof(float), c->side) != hipSuccess) r.ok(SPB_E_STATE);
      hipEventRecord(c->prep_ev, c->side);
      prep_wait = true;
    } else {
      if (prep) r.ok(spb_weight_prep(m->dtype, m->prep_d, m->n_prep, m->n_prep_tiles, m->P, m->wc, stream));
      if (zero_g && hipMemsetAsync(m->G, 0, (size_t)m->n_params * sizeof(float), st) != hipSuccess) r.ok(SPB_E_STATE);
    }
  }
  const bool tr = training != 0;
  const spb_bnupd_entry_t* tab = reinterpret_cast<const spb_bnupd_entry_t*>(c->ws + c->table_off);
  if (tr) {
    if (!c->stats_clean) {
      hipError_t e = hipMemsetAsync(r.stats(), 0, c->stats_floats * sizeof(float), st);
      if (e != hipSuccess) return (int)e;
    }
  } else {
    r.ok(spb_bn_load_running(tab, (int)m->bns.size(), r.stats(), m->Bf, stream));
  }
  c->last_training = training;
  c->one_backward = defer_run && tr;
  c->stats_clean = false;
  c->x = x;
  // stem
  r.tic(PC_STEM_FWD, (double)c->B * 3 * kIn * kIn * 4 + r.elems(m->aStem) * r.es(), 54.0 * r.elems(m->aStem));
  r.ok(spb_stem_fwd(m->dtype, x, m->P + m->stem_w_off, r.z(m->aStem), tr ? r.sums(m->aStem) : nullptr, c->R[m->aStem],
                    c->B, kIn, kIn, stream));
  r.toc();
  auto router_branch = [&]() {   // RouterV2: 1x1 (96 -> 64) + BN + LeakyReLU on block 13's output, reorg into channels [0, 256) of the concat
    r.pw_fwd(m->router, r.block_out(13, tr), m->aR, tr);
    spb_bnapply_args_t a; std::memset(&a, 0, sizeof(a));
    a.Z = r.z(m->aR); a.Y = r.y(m->matCat); a.bn = r.ref(m->aR, tr); a.bn_res = Runner::ident(64);
    a.B = c->B; a.H = 14; a.W = 14; a.C = 64; a.ldc = 1280; a.coff = 0; a.reorg = 2;
    r.tic(PC_BN_APPLY, 2.0 * r.elems(m->aR) * r.es());
    r.ok(spb_bn_apply(m->dtype, &a, (spb_stream_t)r.st));
    r.toc();
  };
  bool router_on_side = false;
  // inverted residual blocks
  Src cur = r.src_act(m->aStem, tr);
  for (int k = 1; k <= 17; ++k) {
    const Block& b = m->blk[k];
    if (b.t != 1 && r.virt(k)) {
      // virtual expanded tensor: the expand launch leaves only its batch sums (and the residual join it forms on the way), the depthwise
      // kernel recomputes the expanded activation from the block input
      // (three forms of block input: a raw tensor + its BatchNorm -> the pass writes the operand to the block's own slot; a pending residual
      // join -> the pass forms and writes the previous block's output, which is the operand; an already materialised previous output
      // (spb_debug_set_join_fused(0)) -> read as it is)
      r.pw_stats(b.E, cur, b.aE, tr, b.matXe >= 0 ? r.y(b.matXe) : nullptr);
      if (cur.mat) cur = r.block_out(k - 1, tr);
      const Src xs = r.xe_src(k);
      r.dw_fwd(b.D, r.src_act(b.aE, tr), b.Hin, b.aD, tr, &b.E, &xs);
    } else if (b.t != 1) {
      r.pw_fwd(b.E, cur, b.aE, tr);
      if (cur.mat) cur = r.block_out(k - 1, tr);   // the join is materialised now: later readers take the block's output tensor
      r.dw_fwd(b.D, r.src_act(b.aE, tr), b.Hin, b.aD, tr);
    } else {
      r.dw_fwd(b.D, cur, b.Hin, b.aD, tr);
    }
    if (prep_wait) { hipStreamWaitEvent(st, c->prep_ev, 0); prep_wait = false; }
    if (k == 14) {
      // block 13's output is final (its residual join was formed by this block's expand convolution): the RouterV2 branch starts
      // here on the side stream, beside blocks 14..17 and the first two ConvDw extras; joined in front of extras[3]
      router_on_side = r.branch_on_side(router_branch);
    }
    r.pw_fwd(b.P, r.src_act(b.aD, tr), b.aP, tr);
    if (b.res && g_join_fused && k < 17) {
      // y_k = bn(z_P) + y_{k-1}: formed by the next block's expand convolution while it loads its operand (pro_mode 3) and
      // written to the block's output tensor by that launch -- no bn_apply launch (10 of them per forward pass)
      Src j = r.src_act(b.aP, tr);
      j.ptr2 = cur.ptr; j.ref2 = cur.ref; j.mat = r.y(b.matY);
      cur = j;
      continue;
    }
    if (b.res) {
      spb_bnapply_args_t a; std::memset(&a, 0, sizeof(a));
      a.Z = r.z(b.aP); a.res = cur.ptr; a.Y = r.y(b.matY); a.bn = r.ref(b.aP, tr); a.bn_res = cur.ref;
      a.B = c->B; a.H = b.Hout; a.W = b.Hout; a.C = b.cout; a.ldc = b.cout; a.coff = 0; a.reorg = 0;
      r.tic(PC_BN_APPLY, 3.0 * r.elems(b.aP) * r.es());
      r.ok(spb_bn_apply(m->dtype, &a, stream));
      r.toc();
    }
    cur = r.block_out(k, tr);
  }
  const Src feat = cur;  // base[-1] output (RevGrad's hooked feature, revgrad.py:66-71)
  // extras
  r.dw_fwd(m->eD[0], feat, 7, m->aED[0], tr);
  r.pw_fwd(m->eP[0], r.src_act(m->aED[0], tr), m->aEP[0], tr);
  r.dw_fwd(m->eD[1], r.src_act(m->aEP[0], tr), 7, m->aED[1], tr);
  r.pw_fwd(m->eP[1], r.src_act(m->aED[1], tr), m->aEP[1], tr);
  if (!router_on_side) router_branch();
  {  // cat((reorg(router), x1), dim=1)  (park2019.py:74-80): x1 = extras[1] output into channels [256, 1280)
    spb_bnapply_args_t a; std::memset(&a, 0, sizeof(a));
    a.B = c->B; a.ldc = 1280; a.Y = r.y(m->matCat);
    a.Z = r.z(m->aEP[1]); a.bn = r.ref(m->aEP[1], tr); a.bn_res = Runner::ident(1024);
    a.H = 7; a.W = 7; a.C = 1024; a.coff = 256; a.reorg = 0;
    r.tic(PC_BN_APPLY, 2.0 * r.elems(m->aEP[1]) * r.es());
    r.ok(spb_bn_apply(m->dtype, &a, stream));
    r.toc();
  }
  if (router_on_side) hipStreamWaitEvent(st, c->router_ev, 0);   // the concat is complete when both halves are
  r.dw_fwd(m->eD[3], r.src_mat(m->matCat), 7, m->aED[3], tr);
  r.pw_fwd(m->eP[3], r.src_act(m->aED[3], tr), m->aEP[3], tr);
  {  // head + loss
    spb_head_args_t h; std::memset(&h, 0, sizeof(h));
    h.Z = r.z(m->aEP[3]); h.Wp = r.wc(m->head_wc_off); h.bias = m->P + m->head_b_off; h.target = target;
    h.partial = reinterpret_cast<float*>(c->ws + c->partial_off); h.pred = pred;
    h.dout = reinterpret_cast<float*>(c->ws + c->dout_off); h.scalars = scalars; h.pro = r.ref(m->aEP[3], tr);
    h.B = c->B; h.J = m->J; h.Jp = m->Jp; h.HW = 49; h.C = 1024; h.S = c->S;
    r.tic(PC_HEAD_FWD, (r.elems(m->aEP[3]) + (double)m->Jp * 49 * 1024) * r.es(), 2.0 * c->B * m->J * 49 * 1024);
    r.ok(spb_head_fwd(m->dtype, &h, stream));
    r.toc();
  }
  if (m->dann && domain_logits) {  // domain classifier on the (gradient-reversed) feature
    r.tic(PC_DOMAIN, ((double)c->B * 49 * (320 + 1280) + 320.0 * 1280) * r.es(), 2.0 * c->B * 49 * 320 * 1280);
    spb_gemm_args_t g; std::memset(&g, 0, sizeof(g));
    g.A = feat.ptr; g.Bw = r.wc(m->dc0.wc_off); g.Y = c->ws + c->dom1_off; g.bias = m->P + m->dc0.bias_off;
    g.pro = feat.ref; g.M = c->B * 49; g.K = 320; g.N = 1280; g.pro_mode = 1; g.epi_mode = 0; g.out_act = SPB_ACT_RELU;
    g.oR = 1; g.out_scale = 1.f;
    r.ok(spb_pwconv_gemm(m->dtype, &g, stream));
    if (g_domain_tail_rows && (49 <= 56) && !c->det) {      // (reproducible mode: the walking kernel -- no float atomics into the caller's logits)
      if (hipMemsetAsync(domain_logits, 0, (size_t)c->B * sizeof(float), st) != hipSuccess) r.ok(SPB_E_STATE);
      const dim3 tg((unsigned)c->B, (1280 + 255) / 256);
      if (m->dtype == SPB_BF16)
        hipLaunchKernelGGL(domain_tail_fwd_rows_kernel<bf16_t>, tg, dim3(256), 0, st, (const bf16_t*)(c->ws + c->dom1_off),
                           (const float*)(m->P + m->dc3_w_off), (const float*)(m->P + m->dc3_b_off),
                           reinterpret_cast<float*>(c->ws + c->dompool_off), domain_logits, 49, 1280);
      else
        hipLaunchKernelGGL(domain_tail_fwd_rows_kernel<float>, tg, dim3(256), 0, st, (const float*)(c->ws + c->dom1_off),
                           (const float*)(m->P + m->dc3_w_off), (const float*)(m->P + m->dc3_b_off),
                           reinterpret_cast<float*>(c->ws + c->dompool_off), domain_logits, 49, 1280);
    } else
    hipLaunchKernelGGL(domain_tail_fwd_kernel, dim3(c->B), dim3(256), 0, st, (const void*)(c->ws + c->dom1_off), m->dtype,
                       (const float*)(m->P + m->dc3_w_off), (const float*)(m->P + m->dc3_b_off),
                       reinterpret_cast<float*>(c->ws + c->dompool_off), domain_logits, 49, 1280);
    r.toc();
  }
  if (tr && defer_run && training != 2 && r.side_usable()) c->pending_running = true;
  else if (tr && training != 2) {   // training == 2: the caller applies the running-statistics update later (spb_krn_update_running)
    r.tic(PC_BN_UPDATE, (double)c->stats_floats * 2 + (double)m->n_buffers * 8);
    r.ok(spb_bn_running_update(tab, (int)m->bns.size(), r.stats(), m->Bf, m->nbt, kMomentum, stream));
    r.toc();
  }
  hipError_t le = hipGetLastError();
  if (le != hipSuccess && r.err == 0) r.err = (int)le;
  return r.err;
}

// BatchNorm running statistics + num_batches_tracked from the batch sums of this context's last training forward
// (momentum 0.1, unbiased variance).  For forwards enqueued with training == 2: two passes that run concurrently on two
// streams (DANN source / target) must still update the shared buffers in the reference's order.
extern "C" int spb_krn_update_running(spb_krn_ctx_t* c, spb_stream_t stream) {
  if (!c || !c->last_training) return SPB_E_STATE;
  spb_krn* m = c->m;
  c->pending_running = false;
  Runner r(c, (hipStream_t)stream);
  const spb_bnupd_entry_t* tab = reinterpret_cast<const spb_bnupd_entry_t*>(c->ws + c->table_off);
  return spb_bn_running_update(tab, (int)m->bns.size(), r.stats(), m->Bf, m->nbt, kMomentum, stream);
}

// Introspection (parity tests): write the raw output z of every VIRTUAL expand convolution of the context's last forward pass into its
// workspace slot (spb_krn_ctx_act_info), rounded to the storage type -- the tensors no kernel of the step reads or writes.  Batch sums
// and every other tensor are left alone.
extern "C" int spb_krn_ctx_materialize(spb_krn_ctx_t* c, spb_stream_t stream) {
  if (!c) return SPB_E_ARG;
  spb_krn* m = c->m;
  Runner r(c, (hipStream_t)stream);
  const bool tr = c->last_training != 0;
  for (int k = 2; k <= 17; ++k) {
    if (!r.virt(k)) continue;
    const Block& b = m->blk[k];
    const Src in = r.xe_src(k);                  // the operand the forward pass materialised
    spb_gemm_args_t g; std::memset(&g, 0, sizeof(g));
    g.A = in.ptr; g.Bw = r.wc(b.E.wc_off); g.Y = r.z(b.aE); g.pro = in.ref; g.M = r.M(b.aE); g.K = b.E.K; g.N = b.E.N;
    g.pro_mode = 1; g.out_scale = 1.f; g.epi_mode = 0; g.oR = 1;
    r.ok(spb_pwconv_gemm(m->dtype, &g, stream));
  }
  return r.err;
}
extern "C" int spb_krn_ctx_virtual(const spb_krn_ctx_t* c, int a) {   // 1: BatchNorm'd tensor `a` is virtual in this context (never stored)
  if (!c || a < 0 || a >= (int)c->m->acts.size()) return SPB_E_ARG;
  Runner r(const_cast<spb_krn_ctx_t*>(c), nullptr);
  for (int k = 2; k <= 17; ++k) if (c->m->blk[k].aE == a) return r.virt(k) ? 1 : 0;
  return 0;
}

// ---- backward --------------------------------------------------------------------------------------------------
extern "C" int spb_bce_logits(const float* logits, float label, int B, float* loss_out, float* dlogit_out, float gscale,
                              spb_stream_t stream) {
  if (!logits || B <= 0) return SPB_E_ARG;
  hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, label, B, loss_out, dlogit_out, gscale);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_krn_backward(spb_krn_ctx_t* c, float* grads, float gscale, int with_pose, const float* dlogit,
                                float alpha, spb_stream_t stream) {
  if (!c) return SPB_E_ARG;
  if (c->poisoned()) return SPB_E_TIMEOUT;
  spb_krn* m = c->m;
  float* const bound_G = m->G;
  if (grads) m->G = grads;
  struct Restore { spb_krn* m; float* g; ~Restore() { m->G = g; } } restore{m, bound_G};
  if (!m->G || !c->last_training) return SPB_E_STATE;
  if (c->stats_clean) return SPB_E_STATE;      // a second backward after a fused-train-step forward: the batch sums it needs were zeroed by the first
  // Fused train step (forward flag 16: exactly one backward follows): the last reader of the batch sums zeroes them for the next
  // forward.  Other callers may run several backward passes from one forward state (tests do), and a training == 2 context still owes
  // its running-statistics update (spb_krn_update_running reads the sums after both DANN passes): those keep the memset in forward.
  const bool zero_stats = c->one_backward && c->last_training == 1;
  if (!with_pose && !dlogit) return SPB_E_ARG;
  if (dlogit && !m->dann) return SPB_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  Runner r(c, st);
  c->n_fork = 0;
  c->bucket_recorded = false;
  const int dt = m->dtype;
  const int aF = m->blk[17].aP;  // feature = bn(z) of block 17's projection (no residual there)
  void* ddom = nullptr;
  bool router_on_side = false;
  if (dlogit) {  // domain classifier backward, then the gradient-reversal layer (-alpha) into the feature
    r.tic(PC_DOMAIN, ((double)c->B * 49 * (3 * 1280 + 2 * 320) + 2 * 320.0 * 1280) * r.es(), 4.0 * c->B * 49 * 320 * 1280);
    const dim3 tgrid((1280 + 63) / 64, (c->B * 49 + DTB_ROWS - 1) / DTB_ROWS);
    const char* tail_ref = std::getenv("SPB_DOMAIN_TAIL_REF");      // test rig: the serial reference kernel
    if (tail_ref && tail_ref[0] == '1')
      hipLaunchKernelGGL(domain_tail_bwd_ref_kernel, dim3((1280 + 255) / 256), dim3(256), 0, st,
                         (const void*)(c->ws + c->dom1_off), (void*)(c->ws + c->gdom_off), dt,
                         (const float*)(m->P + m->dc3_w_off), (const float*)(c->ws + c->dompool_off), dlogit,
                         m->G + m->dc3_w_off, m->G + m->dc3_b_off, m->G + m->dc0.bias_off, c->B, 49, 1280);
    else if (dt == SPB_BF16)
      hipLaunchKernelGGL(domain_tail_bwd_kernel<bf16_t>, tgrid, dim3(256), 0, st,
                         (const bf16_t*)(c->ws + c->dom1_off), (bf16_t*)(c->ws + c->gdom_off),
                         (const float*)(m->P + m->dc3_w_off), (const float*)(c->ws + c->dompool_off), dlogit,
                         m->G + m->dc3_w_off, m->G + m->dc3_b_off, m->G + m->dc0.bias_off, c->B, 49, 1280);
    else
      hipLaunchKernelGGL(domain_tail_bwd_kernel<float>, tgrid, dim3(256), 0, st,
                         (const float*)(c->ws + c->dom1_off), (float*)(c->ws + c->gdom_off),
                         (const float*)(m->P + m->dc3_w_off), (const float*)(c->ws + c->dompool_off), dlogit,
                         m->G + m->dc3_w_off, m->G + m->dc3_b_off, m->G + m->dc0.bias_off, c->B, 49, 1280);
    spb_gemm_args_t g; std::memset(&g, 0, sizeof(g));
    g.A = c->ws + c->gdom_off; g.Bw = r.wc(m->dc0.wct_off); g.Y = c->ws + c->ddom_off; g.pro = Runner::ident(1280);
    g.M = c->B * 49; g.K = 1280; g.N = 320; g.pro_mode = 2; g.epi_mode = 0; g.oR = 1; g.out_scale = -alpha;
    r.ok(spb_pwconv_gemm(dt, &g, stream));
    spb_wgrad_args_t w; std::memset(&w, 0, sizeof(w));
    w.G = c->ws + c->gdom_off; w.X = r.z(aF); w.dW = m->G + m->dc0.w_off; w.pro_dz = Runner::ident(1280);
    w.pro_a = r.ref(aF, true); w.M = c->B * 49; w.K = 320; w.N = 1280;
    r.ok(spb_pwconv_wgrad(dt, &w, stream));
    r.toc();
    ddom = c->ws + c->ddom_off;
  }
  if (with_pose) {
    {  // head
      spb_head_bwd_args_t h; std::memset(&h, 0, sizeof(h));
      h.Z = r.z(m->aEP[3]); h.Wp = r.wc(m->head_wc_off); h.dout = reinterpret_cast<float*>(c->ws + c->dout_off);
      h.G = r.g(m->aEP[3]); h.osums = r.bsums(m->aEP[3]); h.dW = m->G + m->head_w_off; h.dbias = m->G + m->head_b_off;
      h.pro = r.ref(m->aEP[3], true); h.gscale = gscale; h.gscale_dev = c->loss_scale; h.B = c->B; h.J = m->J; h.Jp = m->Jp; h.HW = 49; h.C = 1024;
      h.oR = c->R[m->aEP[3]];
      r.tic(PC_HEAD_BWD, (3.0 * r.elems(m->aEP[3]) + (double)m->Jp * 49 * 1024) * r.es() + 4.0 * m->J * 49 * 1024,
            4.0 * c->B * m->J * 49 * 1024);
      h.roles = 1;
      r.ok(spb_head_bwd(dt, &h, stream));
      h.roles = 2;   // weight + bias gradient: side stream (beside the input-gradient chain), with the first batch of queued
      r.queue_head_wgrad(h);   // pointwise weight gradients (its own fork would cost the launch stream another bubble)
      r.toc();
    }
    // extras[3] = ConvDw(1280,1024) on the concat
    r.pw_bwd(m->eP[3], r.src_act(m->aED[3], true), m->aEP[3], m->aED[3], nullptr, nullptr, 1.f, true);
    r.dw_bwd(m->eD[3], r.src_mat(m->matCat), 7, m->aED[3], -1, c->ws + c->dcat_off, nullptr);
    {  // split the concat gradient: channels [256,1280) -> extras[1] output, [0,256) -> un-reorg -> router output
      spb_bnbwd_args_t a; std::memset(&a, 0, sizeof(a));
      a.dY = c->ws + c->dcat_off; a.Z = r.z(m->aEP[1]); a.G = r.g(m->aEP[1]); a.osums = r.bsums(m->aEP[1]);
      a.bn = r.ref(m->aEP[1], true); a.B = c->B; a.H = 7; a.W = 7; a.C = 1024; a.ldc = 1280; a.coff = 256; a.reorg = 0;
      a.oR = c->R[m->aEP[1]];
      r.tic(PC_BN_BWD_PREP, 3.0 * r.elems(m->aEP[1]) * r.es());
      r.ok(spb_bn_bwd_prep(dt, &a, stream));
      r.toc();
    }
    // RouterV2 branch: un-reorg its share of the concat gradient, then the 1x1 convolution's backward.  Its input gradient (dtap) is
    // needed only when the chain reaches block 14's expand convolution (the skip from block 13's output): on the side stream, behind the
    // depthwise kernel that wrote the concat gradient (the weight gradient goes to the side stream's queue either way)
    auto router_branch = [&]() {
      spb_bnbwd_args_t a; std::memset(&a, 0, sizeof(a));
      a.dY = c->ws + c->dcat_off; a.Z = r.z(m->aR); a.G = r.g(m->aR); a.osums = r.bsums(m->aR); a.bn = r.ref(m->aR, true);
      a.B = c->B; a.H = 14; a.W = 14; a.C = 64; a.ldc = 1280; a.coff = 0; a.reorg = 2; a.oR = c->R[m->aR];
      r.tic(PC_BN_BWD_PREP, 3.0 * r.elems(m->aR) * r.es());
      r.ok(spb_bn_bwd_prep(dt, &a, (spb_stream_t)r.st));
      r.toc();
      r.pw_bwd(m->router, r.block_out(13, true), m->aR, -1, c->ws + c->dtap_off, nullptr);
    };
    router_on_side = r.branch_on_side(router_branch);
    if (!router_on_side) router_branch();
    r.pw_bwd(m->eP[1], r.src_act(m->aED[1], true), m->aEP[1], m->aED[1], nullptr, nullptr, 1.f, true);
    r.dw_bwd(m->eD[1], r.src_act(m->aEP[0], true), 7, m->aED[1], m->aEP[0], nullptr, nullptr);
    r.pw_bwd(m->eP[0], r.src_act(m->aED[0], true), m->aEP[0], m->aED[0], nullptr, nullptr, 1.f, true);
    r.dw_bwd(m->eD[0], r.block_out(17, true), 7, m->aED[0], aF, nullptr, ddom);
  } else {
    // target-domain pass of DANN: only the domain loss reaches the backbone (dann.py:89-92)
    spb_bnbwd_args_t a; std::memset(&a, 0, sizeof(a));
    a.dY = ddom; a.Z = r.z(aF); a.G = r.g(aF); a.osums = r.bsums(aF); a.bn = r.ref(aF, true);
    a.B = c->B; a.H = 7; a.W = 7; a.C = 320; a.ldc = 320; a.coff = 0; a.reorg = 0; a.oR = c->R[aF];
    r.tic(PC_BN_BWD_PREP, 3.0 * r.elems(aF) * r.es());
    r.ok(spb_bn_bwd_prep(dt, &a, stream));
    r.toc();
  }
  // inverted residual blocks, last to first
  for (int k = 17; k >= 1; --k) {
    const Block& b = m->blk[k];
    const Src in = k == 1 ? r.src_act(m->aStem, true) : r.block_out(k - 1, true);
    const int atgt = k == 1 ? m->aStem : m->blk[k - 1].aP;
    // gradient joining the block input besides this block's own path
    const void* res = nullptr;
    if (b.res) res = r.g(b.aP);                                  // skip connection: d y_{k-1} += d y_k
    else if (k == 14 && with_pose) {                             // RouterV2 branch taps block 13's output
      res = c->ws + c->dtap_off;
      if (router_on_side) { hipStreamWaitEvent(st, c->router_ev, 0); router_on_side = false; }   // its input gradient came from the side stream
    }
    r.pw_bwd(b.P, r.src_act(b.aD, true), b.aP, b.aD, nullptr, nullptr, 1.f, true);
    if (b.t != 1 && r.virt(k)) {
      const Src xs = r.xe_src(k);
      r.dw_bwd(b.D, r.src_act(b.aE, true), b.Hin, b.aD, b.aE, nullptr, nullptr, &b.E, &xs);
      r.pw_bwd(b.E, in, b.aE, atgt, nullptr, res, 1.f, false, true);
    } else if (b.t != 1) {
      r.dw_bwd(b.D, r.src_act(b.aE, true), b.Hin, b.aD, b.aE, nullptr, nullptr);
      r.pw_bwd(b.E, in, b.aE, atgt, nullptr, res);
    } else {
      r.dw_bwd(b.D, in, b.Hin, b.aD, atgt, nullptr, res);
    }
    if (k == kSplitBlock && c->bucket_on) {
      // early gradient bucket: everything from block 14 to the head (and the domain classifier) is final once the side
      // stream's weight gradients so far have landed and the BatchNorm affine gradients of those layers are written
      r.join_side();
      const spb_bnupd_entry_t* tab = reinterpret_cast<const spb_bnupd_entry_t*>(c->ws + c->table_off);
      r.tic(PC_BN_PARAM_GRADS, (double)c->stats_floats);
      if (zero_stats) r.ok(spb_bn_param_grads_zero(tab + m->split_bn, (int)m->bns.size() - m->split_bn, r.stats(), m->G, stream));
      else r.ok(spb_bn_param_grads(tab + m->split_bn, (int)m->bns.size() - m->split_bn, r.stats(), m->G, stream));
      r.toc();
      hipEventRecord(c->bucket_ev, st);
      c->bucket_recorded = true;
    }
  }
  // stem weight gradient (the image needs no gradient)
  {
    spb_bnref_t pro = r.ref(m->aStem, true);
    r.tic(PC_STEM_WGRAD, (double)c->B * 3 * kIn * kIn * 4 + 2.0 * r.elems(m->aStem) * r.es(), 54.0 * r.elems(m->aStem));
    r.ok(spb_stem_wgrad(dt, c->x, r.g(m->aStem), r.z(m->aStem), &pro, m->G + m->stem_w_off, c->B, kIn, kIn, stream));
    r.toc();
  }
  r.join_side();
  if (c->pending_running) {
    r.ok(spb_bn_running_update(reinterpret_cast<const spb_bnupd_entry_t*>(c->ws + c->table_off), (int)m->bns.size(), r.stats(),
                               m->Bf, m->nbt, kMomentum, stream));
    c->pending_running = false;
  }
  r.tic(PC_BN_PARAM_GRADS, (double)c->stats_floats * 2);
  // BatchNorm affine gradients: dgamma += sum(g*xhat), dbeta += sum(g)
  if (zero_stats)
    r.ok(spb_bn_param_grads_zero(reinterpret_cast<const spb_bnupd_entry_t*>(c->ws + c->table_off),
                                 c->bucket_on ? m->split_bn : (int)m->bns.size(), r.stats(), m->G, stream));
  else
    r.ok(spb_bn_param_grads(reinterpret_cast<const spb_bnupd_entry_t*>(c->ws + c->table_off),
                            c->bucket_on ? m->split_bn : (int)m->bns.size(), r.stats(), m->G, stream));
  r.toc();
  c->stats_clean = zero_stats && r.err == 0;    // every BatchNorm's slots were zeroed by the pass(es) above
  hipError_t le = hipGetLastError();
  if (le != hipSuccess && r.err == 0) r.err = (int)le;
  return r.err;
}

// ---- data-parallel gradient exchange, overlapped with backward ------------------------------------------------------
// Element offset in the parameter / gradient arena where the early bucket starts.
extern "C" long long spb_krn_bucket_split(const spb_krn_t* m) { return m ? m->split_off : -1; }
// Makes `comm_stream` wait until the last spb_krn_backward on this context has finished the gradients of
// [spb_krn_bucket_split, n_params): the caller then enqueues that bucket's all-reduce on comm_stream while the launch
// stream continues with blocks 13..1 and the stem.
extern "C" int spb_krn_ctx_set_bucket(spb_krn_ctx_t* c, int on) {
  if (!c) return SPB_E_ARG;
  c->bucket_on = on != 0; c->bucket_recorded = false;
  return 0;
}
extern "C" int spb_krn_ctx_wait_bucket(spb_krn_ctx_t* c, spb_stream_t comm_stream) {
  if (!c || !c->bucket_recorded) return SPB_E_STATE;
  hipError_t e = hipStreamWaitEvent((hipStream_t)comm_stream, c->bucket_ev, 0);
  return e == hipSuccess ? 0 : (int)e;
}

// ---- live per-launch timing ------------------------------------------------------------------------------------
extern "C" int spb_krn_prof_enable(spb_krn_ctx_t* c, int on) {
  if (!c) return SPB_E_ARG;
  c->prof_on = on != 0; c->prof_n = 0;
  return 0;
}
extern "C" int spb_krn_prof_num_categories(void) { return PC_COUNT; }
extern "C" const char* spb_krn_prof_category_name(int i) { return (i >= 0 && i < PC_COUNT) ? kProfNames[i] : ""; }
// Per-launch records since the last spb_krn_prof_read, in launch order: category, milliseconds, algorithmic bytes.  Returns the
// number of records (at most `max` are written); does not reset.
extern "C" int spb_krn_prof_launches(spb_krn_ctx_t* c, int max, int* cat, float* ms, double* bytes) {
  if (!c || !cat || !ms || !bytes) return SPB_E_ARG;
  for (int i = 0; i < c->prof_n && i < max; ++i) {
    if (hipEventSynchronize(c->prof_ev[2 * i + 1]) != hipSuccess) return SPB_E_STATE;
    float t = 0.f;
    if (hipEventElapsedTime(&t, c->prof_ev[2 * i], c->prof_ev[2 * i + 1]) != hipSuccess) return SPB_E_STATE;
    cat[i] = c->prof_cat[i]; ms[i] = t; bytes[i] = c->prof_bytes[i];
  }
  return c->prof_n;
}
// Sums the launches recorded since the last call per category (synchronises on the recorded events) and resets.
extern "C" int spb_krn_prof_read(spb_krn_ctx_t* c, int* launches, float* ms, double* bytes, double* flops) {
  if (!c || !launches || !ms || !bytes || !flops) return SPB_E_ARG;
  for (int i = 0; i < PC_COUNT; ++i) { launches[i] = 0; ms[i] = 0.f; bytes[i] = 0.0; flops[i] = 0.0; }
  for (int i = 0; i < c->prof_n; ++i) {
    hipError_t e = hipEventSynchronize(c->prof_ev[2 * i + 1]);
    if (e != hipSuccess) return (int)e;
    float t = 0.f;
    e = hipEventElapsedTime(&t, c->prof_ev[2 * i], c->prof_ev[2 * i + 1]);
    if (e != hipSuccess) return (int)e;
    const int k = c->prof_cat[i];
    launches[k] += 1; ms[k] += t; bytes[k] += c->prof_bytes[i]; flops[k] += c->prof_flops[i];
  }
  c->prof_n = 0;
  return 0;
}
