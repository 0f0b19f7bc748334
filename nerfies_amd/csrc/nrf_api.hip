// C-ABI layer of the MI355X nerfies hot path: handle, flat-parameter layout, workspace plan and the
// launch sequences that stand in for NerfModel.apply (models.py:289-375) and the gradient half of
// training.train_step (training.py:168-265).  See include/nerfies_amd.h for the contract.
#include "../../include/nerfies_amd.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <string>
#include <vector>

#include "nrf_internal.h"

using namespace nrf;

namespace {

thread_local char g_err[256] = "ok";

int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
int fail_hip(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), where);
  return NRF_E_HIP;
}

}  // namespace (reopened below)

const Knobs& nrf::knobs() {
  static const Knobs k = [] {
    Knobs v;
    v.trace_regions = getenv("NRF_TRACE_REGIONS") != nullptr;
#ifdef NRF_EXPERIMENT
    v.debug_occ = getenv("NRF_DEBUG_OCC") != nullptr;
    v.timeline = getenv("NRF_TIMELINE") != nullptr;
    v.dynamic_tiles = getenv("NRF_DYNAMIC_TILES") != nullptr;
    if (const char* e = getenv("NRF_GRID_MUL")) v.grid_mul = atoi(e) < 1 ? 1 : atoi(e);
    if (const char* e = getenv("NRF_WARP_GRID_MUL")) v.warp_grid_mul = atoi(e);
    if (const char* e = getenv("NRF_OLD_SHARE")) v.old_share = atof(e);
#endif
    return v;
  }();
  return k;
}

namespace {
constexpr size_t ALIGN_F = 64;   // workspace sub-buffers are aligned to 64 floats (256 B)
size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct LevelWs {   // float offsets from the workspace base, per level (0 = coarse, 1 = fine)
  size_t wpk, z, out4, rgb, depth, med, acc, weights, condterm;
  size_t alpha_ct = 0, dsig_ray = 0;   // use_alpha_condition: per-ray code term of the alpha head / per-ray sum of d raw sigma
  size_t bf_wpk = 0;   // bf16 weight stream of the NRF_FLAG_BF16 forward
  // bf16 training (NRF_FLAG_TRAIN | NRF_FLAG_BF16): dgrad weight stream, the two bf16 stashes (nrf_internal.h BfStash), bias slabs
  size_t bf_wpkT = 0, b_pe = 0, b_h = 0, b_bn = 0, b_rgbh = 0, b_bits = 0, b_dy = 0, b_dbn = 0, b_drgbh = 0, b_dsmall = 0;
  int b_ngroups = 0;
  // bf16 SE3 trunk (nrf_internal.h BfWarpStash) of this level's pass through the field
  size_t bw_in = 0, bw_h = 0, bw_bits = 0, bw_dy = 0, bw_dhead = 0;
  int bw_ngroups = 0;
  size_t st_pe, st_h, st_bn, st_rgbh, bits_trunk, bits_rgbh;
  size_t d_raw4, dy_trunk, dy_bn, dy_rgbh, dray, small_part, cond_grad;
  // SE3 warp field (per level: the field is evaluated on the coarse and on the fine samples)
  size_t wpoints, points_raw, d_points;
  size_t el_dw4, el_dv4;   // coarse level: dL/d(w, v) of the elastic regulariser through exp_se3's second derivatives
  size_t w_st_win, w_st_h, w_st_wv, w_bits, w_dy, w_dw4, w_dv4, w_small_part;
};

struct WsPlan {
  int B = -1;
  uint32_t flags = 0;
  uint64_t serial = 0;   // identity of this layout: a stash written under one plan must not be differentiated under another
  int S[4], rows[4], ntiles[4];
  size_t tables;        // byte region at the start: PackDesc[], WgradGroup[], ReduceDesc[]
  size_t pack_off_b, groups_off_b, reduce_off_b, segs_off_b, segbegin_off_b, emb_off_b;
  size_t bf_desc = 0;
  std::vector<RcPackDesc> bfpack;
  bool bf_stream_ok = true;   // the chunk tables emitted by build_plan add up to the stream lengths the kernels walk
  bool bfw = false;           // training plan: the SE3 trunk stashes / differentiates in bfloat16 (warp_bf16.hip)
  size_t bfw_wpk = 0, bfw_wpkT = 0;   // bf16 SE3 weight streams (forward: also in inference plans)
  size_t iparams = 0, igrad = 0;   // zero-padded parameter image / its gradient (models narrower than the kernels)
  std::vector<WgradSegment> segs;
  std::vector<int> seg_begin;
  int wgrad_nwg = 0;
  // the same tables for the bf16 wgrad kernel (NeRF MLP groups of a bf16 training plan; "tile" = 32-sample group)
  std::vector<WgradGroup> bgroups;
  std::vector<WgradSegment> bsegs;
  std::vector<int> bseg_begin;
  size_t bgroups_off_b = 0, bsegs_off_b = 0, bsegbegin_off_b = 0;
  int bwgrad_nwg = 0;
  size_t cond, mse, zero_rgb, slabs;
  size_t warp_wpk;      // packed SE3 trunk weights (shared by both levels)
  size_t bg_loss;       // [64] background-loss accumulator
  size_t bg_points = 0, bg_ids = 0;   // [bgN][3] noised points / [bgN] ids drawn by the library
  size_t el_sums;       // [5][rows_pad / 256] elastic_kernel's per-workgroup partial sums (loss, residual, det / div / curl J)
  size_t el_coef;       // [B][N_c] one-hot sample selector of elastic_reduce_method 'median'
  size_t wr_sums = 0;   // [64] warp_reg loss / residual accumulators (coarse: 0, 1; fine: 2, 3)
  size_t t_codes = 0, t_dcodes = 0, t_in = 0, t_h = 0, t_dpre = 0;   // TimeEncoder: codes [B][G], their gradient, stashes
  size_t counters;      // [64] ints: dynamic tile counters of the chain kernels, zeroed at the start of forward / backward
  size_t timeline;      // [2 levels][4 waves][64] uint64 debug stamps of workgroup 0 of the forward chain kernel
  size_t seg_clock;     // [nsegs] uint64 wall-clock ticks per wgrad segment (cost-model calibration)
  int nreduce_pass[4] = {0, 0, 0, 0};   // reduce descriptors by pass: pass 0 overwrites, passes 1 (fine level) and 2
                                     // (background batch), 3 (Jacobian tangents) add into leaves shared with earlier passes
  LevelWs L[4];          // 0 coarse, 1 fine, 2 background points (SE3 field only, training.py:117-135),
                         // 3 tangent pass of the coarse warp Jacobian (elastic regulariser, 3 x coarse tiles)
  int elastic = 0;       // plan built with the elastic regulariser's buffers
  bool bwd32 = false;    // training plan: the fp32 NeRF reverse chain runs on 32-row tiles (mlp_chain32.hip); decides the
                         // number of bias partials the reduce table sums
  int chain_rows_opt = 0;   // the handle's options the plan was built under
  int bf16_wgrad_merge = 0;
  int tg_tiles_per = 0;  // primal tiles one tangent pass covers (elastic: coarse level; Jacobian output: the larger level)
  int bgN = 0;           // number of background points the plan was built for
  size_t total_floats;
  std::vector<PackDesc> pack;
  std::vector<WgradGroup> groups;
  std::vector<ReduceDesc> reduce;
  int ntasks = 0;
};

}  // namespace

namespace {
struct ProfSlot { std::string name; double flops; hipEvent_t a = nullptr, b = nullptr; bool used = false; };
struct ProfAcc { std::string name; double ms = 0; int launches = 0; double flops = 0; };
struct Prof {
  bool on = false;
  std::vector<ProfSlot> slots;   // events recorded and not yet read
  size_t next = 0;
  std::vector<ProfAcc> acc;
  // NRF_TRACE_REGIONS=1 (debugging aid): name every region on stderr and synchronise the stream behind it, so that a device
  // fault is attributed to the kernel group that raised it
  static bool trace() { return knobs().trace_regions; }
  // Neither the trace's stream synchronise nor the profiler's event records are legal inside a stream capture (a
  // GraphedTrainStep / GraphedChunkRenderer capture with either switched on would be invalidated and surface as an unrelated
  // HIP error): both are skipped while `st` is capturing.
  static bool capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
  }
  bool open = false;   // begin() recorded an event that end() must close
  void begin(const char* name, double flops, hipStream_t st) {
    open = false;
    if (!trace() && !on) return;
    if (capturing(st)) return;
    if (trace()) { fprintf(stderr, "[nrf] %s ...", name); fflush(stderr); }
    if (!on) return;
    if (next == slots.size()) { slots.emplace_back(); (void)hipEventCreate(&slots.back().a); (void)hipEventCreate(&slots.back().b); }
    ProfSlot& s = slots[next];
    s.name = name; s.flops = flops; s.used = true;
    (void)hipEventRecord(s.a, st);
    open = true;
  }
  void end(hipStream_t st) {
    if (!trace() && !on) return;
    if (capturing(st)) return;
    if (trace()) { const hipError_t e = hipStreamSynchronize(st); fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e)); fflush(stderr); }
    if (!on || !open) return;
    (void)hipEventRecord(slots[next].b, st);
    ++next;
    open = false;
  }
  void drain() {
    for (size_t i = 0; i < next; ++i) {
      ProfSlot& s = slots[i];
      (void)hipEventSynchronize(s.b);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, s.a, s.b);
      ProfAcc* a = nullptr;
      for (auto& x : acc) if (x.name == s.name) a = &x;
      if (!a) { acc.emplace_back(); a = &acc.back(); a->name = s.name; }
      a->ms += ms; a->launches += 1; a->flops = s.flops;
    }
    next = 0;
  }
};
}  // namespace

struct nrf_handle_s {
  Prof prof;
  nrf_model_desc d;
  std::vector<nrf_tensor_info> layout;    // INTERNAL leaves (kernel widths); == xlayout unless `embed`
  int64_t nparams = 0;
  std::vector<nrf_tensor_info> xlayout;   // the caller's leaves (nrf_param_layout)
  int64_t xnparams = 0;
  std::vector<EmbedDesc> emb;             // external <-> internal element map, one per leaf
  bool embed = false;                     // trunk / rgb branch narrower than the kernels: run on a zero-padded image
  WarpParamOffsets xwpo;                  // warp leaves at their EXTERNAL offsets (nrf_warp_points reads the caller's buffer)
  MlpParamOffsets po[2];
  PackOffsets pk;
  int64_t app_off = -1, cam_off = -1;
  int P, PK, R, V, app_in_cond, nlevels;
  int A = 0;   // width of the alpha condition (use_appearance_metadata && use_alpha_condition: the appearance code)
  bool warp = false;
  WarpParamOffsets wpo;
  WarpPackOffsets wpk;
  int Fw = 0, G = 0, Win = 0, PKw = 0;
  bool time_enc = false;   // warp_metadata_encoder_type 'time': the codes come from modules.TimeEncoder instead of a GLO table
  int Ft = 0, Tin = 0;
  TimeParamOffsets tpo;
  int num_cus = 256;
  bool cu_queried = false;
  int chain_rows_opt = 0;   // NRF_OPT_CHAIN_TILE_ROWS: 0 automatic, 32, 64
  int bf16_wgrad_merge = 1; // NRF_OPT_BF16_WGRAD_MERGE: 1 (default) = skip-layer / bottleneck+alpha groups of the bf16 wgrad merged (operands
                            // streamed once: -10 % HBM fetch, +1..2 % step rate in the same-box A/B, profiles/r05_wgrad_bf16_merge_ab.md)
  WsPlan plan;
  // identity of the tables last uploaded to a workspace, and of the last stashed forward
  void* uploaded_ws = nullptr;
  int xdepth = TRUNK_DEPTH, xskip = SKIP_LAYER;   // the caller's trunk (<= 8 layers; its skip index or -1): nrf_create
  int emap[TRUNK_DEPTH];                          // internal trunk layer -> the caller's layer, or -1 (identity layer)
  int wxdepth = WARP_DEPTH, wxwidth = WARP_W;     // the caller's warp trunk (warp_kwargs trunk_depth / trunk_width)
  int uploaded_B = -1;
  uint32_t uploaded_flags = 0;
  int uploaded_bgN = 0;
  int uploaded_elastic = 0;
  void* stashed_ws = nullptr;
  uint64_t stashed_plan = 0;   // WsPlan::serial of the stashed forward
  int stashed_B = -1;
  bool stashed_warp = false;
  std::vector<PackDesc> wp_pack;   // pack table of nrf_warp_points (kept alive for the async upload)
  int64_t wp_pack_base = -1;
};

namespace {

// Appends a leaf to the internal layout (rows x cols = what the kernels index) and to the external one
// (xrows x xcols = what the model owns; defaults to the same).  External rows >= split sit `shift` rows lower inside.
// xname: the leaf's path in the caller's tree when it differs from the internal one; "" = internal only (no external
// leaf: stays zero in the padded image, its gradient is dropped).
void add_leaf(nrf_handle h, const std::string& name, int rows, int cols, int64_t* off_out, int xrows = -1, int xcols = -1,
              int split = -1, int64_t* xoff_out = nullptr, const char* xname = nullptr) {
  if (xrows < 0) xrows = rows;
  if (xcols < 0) xcols = cols;
  nrf_tensor_info t;
  memset(&t, 0, sizeof(t));
  snprintf(t.name, sizeof(t.name), "%s", name.c_str());
  t.offset = h->nparams;
  t.rows = rows;
  t.cols = cols;
  if (off_out) *off_out = t.offset;
  h->nparams += (int64_t)rows * cols;
  h->nparams = (int64_t)align_up((size_t)h->nparams, 4);   // keep every leaf 16-byte aligned
  h->layout.push_back(t);
  if (xname && !*xname) { h->embed = true; return; }
  nrf_tensor_info x = t;
  if (xname) { snprintf(x.name, sizeof(x.name), "%s", xname); h->embed = true; }
  x.offset = h->xnparams;
  x.rows = xrows;
  x.cols = xcols;
  if (xoff_out) *xoff_out = x.offset;
  h->xnparams += (int64_t)xrows * xcols;
  h->xnparams = (int64_t)align_up((size_t)h->xnparams, 4);
  h->xlayout.push_back(x);
  EmbedDesc e;
  e.ext_off = x.offset; e.int_off = t.offset; e.rows = xrows; e.ext_cols = xcols; e.int_cols = cols;
  e.split = split < 0 ? xrows : split; e.shift = rows - xrows; e.pad_ = 0;
  h->emb.push_back(e);
  if (xrows != rows || xcols != cols) h->embed = true;
}

void build_layout(nrf_handle h) {
  const nrf_model_desc& d = h->d;
  const int W = TRUNK_W, RW = RGB_W;                               // what the kernels index
  const int XW = d.nerf_trunk_width, XRW = d.nerf_rgb_branch_width;   // what the model owns (<= W, RW)
  for (int lv = 0; lv < h->nlevels; ++lv) {
    const std::string base = lv == 0 ? "nerf_mlps_coarse" : "nerf_mlps_fine";
    MlpParamOffsets& po = h->po[lv];
    for (int i = 0; i < TRUNK_DEPTH; ++i) {
      const int hid = i == 0 ? 0 : 1;                 // rows of the running activation, then (layer 0 / skip) the posenc rows
      const int e = h->emap[i];                       // the caller's layer that runs here, or -1: an identity layer
      const int pe = (i == 0 || i == d.nerf_skip_layer) ? h->P : 0;
      const int xpe = (e == 0 || (e >= 0 && e == h->xskip)) ? h->P : 0;   // a skip the caller's trunk never reaches: zero posenc rows inside
      const std::string kn = base + "/MLP_0/hidden_" + std::to_string(i) + "/kernel", bn = base + "/MLP_0/hidden_" + std::to_string(i) + "/bias";
      if (e >= 0) {
        const std::string xkn = base + "/MLP_0/hidden_" + std::to_string(e) + "/kernel", xbn = base + "/MLP_0/hidden_" + std::to_string(e) + "/bias";
        add_leaf(h, kn, hid * W + pe, W, &po.trunk_k[i], hid * XW + xpe, XW, hid * XW, nullptr, e != i ? xkn.c_str() : nullptr);
        add_leaf(h, bn, 1, W, &po.trunk_b[i], 1, XW, -1, nullptr, e != i ? xbn.c_str() : nullptr);
      } else {   // between / behind the caller's layers: internal-only identity (relu(h . I) = h for h >= 0; its gradient is dropped)
        add_leaf(h, kn, hid * W + pe, W, &po.trunk_k[i], -1, -1, -1, nullptr, "");
        add_leaf(h, bn, 1, W, &po.trunk_b[i], -1, -1, -1, nullptr, "");
        EmbedDesc e2;
        e2.ext_off = -1; e2.int_off = po.trunk_k[i]; e2.rows = XW; e2.ext_cols = 1; e2.int_cols = W; e2.split = XW; e2.shift = 0; e2.pad_ = 0;
        h->emb.push_back(e2);
      }
      if (xpe != pe) h->embed = true;
    }
    if (h->R == 0 && h->A == 0) {
      // no condition at all (use_viewdirs = False, no camera / appearance code): NerfMLP has NO bottleneck layer and the rgb branch
      // reads the trunk output (modules.py:149-164).  The kernels keep their layer list: the bottleneck becomes an internal-only
      // IDENTITY (x . I + 0 is exact in float32, and exact on the bf16 chain, whose h8 is already bf16), its gradient is dropped
      add_leaf(h, base + "/bottleneck/kernel", W, W, &po.bn_k, -1, -1, -1, nullptr, "");
      add_leaf(h, base + "/bottleneck/bias", 1, W, &po.bn_b, -1, -1, -1, nullptr, "");
      EmbedDesc e;
      e.ext_off = -1; e.int_off = po.bn_k; e.rows = XW; e.ext_cols = 1; e.int_cols = W; e.split = XW; e.shift = 0; e.pad_ = 0;
      h->emb.push_back(e);
    } else {
      add_leaf(h, base + "/bottleneck/kernel", W, W, &po.bn_k, XW, XW);
      add_leaf(h, base + "/bottleneck/bias", 1, W, &po.bn_b, 1, XW);
    }
    add_leaf(h, base + "/MLP_1/hidden_0/kernel", W + h->R, RW, &po.rgbh_k, XW + h->R, XRW, XW);
    add_leaf(h, base + "/MLP_1/hidden_0/bias", 1, RW, &po.rgbh_b, 1, XRW);
    add_leaf(h, base + "/MLP_1/logit/kernel", RW, 3, &po.logit_k, XRW, 3);
    add_leaf(h, base + "/MLP_1/logit/bias", 1, 3, &po.logit_b);
    add_leaf(h, base + "/MLP_2/logit/kernel", W + h->A, 1, &po.alpha_k, XW + h->A, 1, XW);   // [bottleneck | appearance code] (modules.py:152-157)
    add_leaf(h, base + "/MLP_2/logit/bias", 1, 1, &po.alpha_b);
  }
  if (h->warp) {   // warping.SE3Field (warping.py:202-320); flax names per SURVEY.md A.2
    WarpParamOffsets& w = h->wpo;
    WarpParamOffsets& x = h->xwpo;
    if (h->time_enc) {   // modules.TimeEncoder (modules.py:297-322): self.mlp = MLP(depth 6, width 64, skips (4,), output G)
      w.embed = x.embed = -1;
      for (int i = 0; i < TIME_DEPTH; ++i) {
        const int fin = i == 0 ? h->Tin : i == TIME_SKIP ? TIME_W + h->Tin : TIME_W;
        add_leaf(h, "warp_field/metadata_encoder/mlp/hidden_" + std::to_string(i) + "/kernel", fin, TIME_W, &h->tpo.k[i]);
        add_leaf(h, "warp_field/metadata_encoder/mlp/hidden_" + std::to_string(i) + "/bias", 1, TIME_W, &h->tpo.b[i]);
      }
      add_leaf(h, "warp_field/metadata_encoder/mlp/logit/kernel", TIME_W, d.num_warp_features, &h->tpo.lk);
      add_leaf(h, "warp_field/metadata_encoder/mlp/logit/bias", 1, d.num_warp_features, &h->tpo.lb);
    } else {
      add_leaf(h, "warp_field/metadata_encoder/embed/embedding", d.num_warp_embeddings, d.num_warp_features, &w.embed, -1, -1, -1,
               &x.embed);
    }
    // TranslationField (warping.py:62-199) = the same 6x128 trunk with ONE 3-channel output layer and x' = x + t:
    // exactly the SE3 field with a zero rotation head (theta = 0: R = I, p = v; the closed forms are series in
    // theta^2 there).  Its leaves 'warp_field/mlp/hidden_i' / 'mlp/logit' map onto trunk / branches_v; branches_w
    // exists only internally and stays zero.
    const bool tr = d.warp_field_type == NRF_WARP_TRANSLATION;
    // warp_kwargs trunk_depth / trunk_width (warping.py:225-227): a shallower / narrower trunk runs on the 6 x 128 kernels --
    // identity layers behind the caller's last one (every trunk layer ends in a ReLU: modules.py:41-50), zero padding to 128
    // columns, zero input rows in the skip layer when the caller's trunk (<= 4 layers) never reaches it
    const int XD = h->wxdepth, XWw = h->wxwidth;
    for (int i = 0; i < WARP_DEPTH; ++i) {
      const int hid = i == 0 ? 0 : 1;
      const int pe = (i == 0 || i == WARP_SKIP) ? h->Win : 0;
      const std::string nk = "warp_field/trunk/hidden_" + std::to_string(i) + "/kernel", nb = "warp_field/trunk/hidden_" + std::to_string(i) + "/bias";
      const std::string xk = "warp_field/mlp/hidden_" + std::to_string(i) + "/kernel", xb = "warp_field/mlp/hidden_" + std::to_string(i) + "/bias";
      if (i < XD) {
        add_leaf(h, nk, hid * WARP_W + pe, WARP_W, &w.trunk_k[i], hid * XWw + pe, XWw, hid * XWw, &x.trunk_k[i], tr ? xk.c_str() : nullptr);
        add_leaf(h, nb, 1, WARP_W, &w.trunk_b[i], 1, XWw, -1, &x.trunk_b[i], tr ? xb.c_str() : nullptr);
      } else {
        add_leaf(h, nk, hid * WARP_W + pe, WARP_W, &w.trunk_k[i], -1, -1, -1, nullptr, "");
        add_leaf(h, nb, 1, WARP_W, &w.trunk_b[i], -1, -1, -1, nullptr, "");
        EmbedDesc e;
        e.ext_off = -1; e.int_off = w.trunk_k[i]; e.rows = XWw; e.ext_cols = 1; e.int_cols = WARP_W; e.split = XWw; e.shift = 0; e.pad_ = 0;
        h->emb.push_back(e);
      }
    }
    add_leaf(h, "warp_field/branches_w/logit/kernel", WARP_W, 3, &w.w_k, XWw, 3, -1, &x.w_k, tr ? "" : nullptr);
    add_leaf(h, "warp_field/branches_w/logit/bias", 1, 3, &w.w_b, -1, -1, -1, &x.w_b, tr ? "" : nullptr);
    add_leaf(h, "warp_field/branches_v/logit/kernel", WARP_W, 3, &w.v_k, XWw, 3, -1, &x.v_k, tr ? "warp_field/mlp/logit/kernel" : nullptr);
    add_leaf(h, "warp_field/branches_v/logit/bias", 1, 3, &w.v_b, -1, -1, -1, &x.v_b, tr ? "warp_field/mlp/logit/bias" : nullptr);
  }
  if (d.use_appearance_metadata)
    add_leaf(h, "appearance_encoder/embed/embedding", d.num_appearance_embeddings, d.num_appearance_features, &h->app_off);
  if (d.use_camera_metadata)
    add_leaf(h, "camera_encoder/embed/embedding", d.num_camera_embeddings, d.num_camera_features, &h->cam_off);
}

void build_pack_offsets(nrf_handle h) {
  PackOffsets& pk = h->pk;
  int o = 0;
  auto take = [&](int n) { int r = o; o += n; return r; };
  pk.fwd_L[0] = take(h->PK * 256);
  for (int l = 1; l < TRUNK_DEPTH; ++l) pk.fwd_L[l] = take(256 * 256);
  pk.fwd_L4b = take(h->PK * 256);
  pk.fwd_bn = take(256 * 256);
  pk.fwd_rgbh = take(256 * 128);
  pk.bwd_rgbhT = take(128 * 256);
  pk.bwd_bnT = take(256 * 256);
  pk.bwd_LT[0] = 0;
  for (int l = 1; l < TRUNK_DEPTH; ++l) pk.bwd_LT[l] = take(256 * 256);
  pk.bwd_L0T = pk.bwd_L4bT = 0;
  if (h->warp) { pk.bwd_L0T = take(256 * 64); pk.bwd_L4bT = take(256 * 64); }
  pk.total = o + 4096;   // slack: the K loop prefetches two quads past a layer's last weights
  if (h->warp) {
    WarpPackOffsets& w = h->wpk;
    int ow = 0;
    auto takew = [&](int n) { int r = ow; ow += n; return r; };
    w.fwd_L[0] = takew(h->PKw * WARP_W);
    for (int l = 1; l < WARP_DEPTH; ++l) w.fwd_L[l] = takew(WARP_W * WARP_W);
    w.fwd_L4b = takew(h->PKw * WARP_W);
    w.bwd_LT[0] = 0;
    for (int l = 1; l < WARP_DEPTH; ++l) w.bwd_LT[l] = takew(WARP_W * WARP_W);
    w.total = ow + 4096;
  }
}

// Lays out the workspace for B rays and (re)builds the descriptor tables.
// dynamic tile counters (ints at ws + plan.counters)
enum { CT_WARP_FWD = 0, CT_MLP_FWD = 2, CT_TAN_FWD = 4, CT_MLP_BWD = 5, CT_WARP_BWD = 7, CT_TAN_BWD = 9, CT_BG_FWD = 10, CT_BG_BWD = 11 };
// Measured (r01): pulling tiles from a global counter is 4-6 % SLOWER than the static round-robin split for the
// chain kernels (fine forward 1.99 vs 1.87 ms) although it removes the tail where the younger workgroup of a CU
// runs alone -- so static is the default and NRF_DYNAMIC_TILES=1 keeps the other path testable.
// Uneven static tile split of the NeRF chain kernels (chain_common.h tile_iter): tiles the older workgroup of a CU takes out
// of the K = ceil(ntiles / CUs) of its CU, when the launch is exactly two workgroups per CU and K >= 4.  NRF_OLD_SHARE
// overrides the share (0 = even split).
int k_old_for(int ntiles, int grid, int num_cus, double dflt_share) {
  if (grid != 2 * num_cus) return 0;
  const int K = (ntiles + num_cus - 1) / num_cus;
  if (K < 4) return 0;
  const double share = knobs().old_share >= 0.0 ? knobs().old_share : dflt_share;
  if (share <= 0.0) return 0;
  int k = (int)floor(K * share + 0.5);
  return k < 1 ? 1 : (k > K - 1 ? K - 1 : k);
}

// workgroups per CU of the SE3 chain kernels' launches
int warp_grid_mul() {
  const int m = knobs().warp_grid_mul;
  return m < 1 ? 1 : (m > 4 ? 4 : m);
}

int* tile_counter_or_null(float* base, int idx) {
  return knobs().dynamic_tiles ? reinterpret_cast<int*>(base) + idx : nullptr;
}
constexpr int BG = 2;   // level index of the background-point batch
// automatic choice of the forward chain's tiling (chain32_for): 32-row tiles when the launch has fewer than this many 64-row
// tiles per CU (the 64-row grid of two workgroups per CU is then not filled)
constexpr int AUTO32_FWD_BELOW_TILES_PER_CU = 2;
constexpr int TG = 3;   // level index of the Jacobian tangent pass (3 x the coarse tiles)

// the flags a workspace layout depends on: TRAIN, WARP_JACOBIAN, and BF16 together with TRAIN (bf16 stash instead of fp32)
uint32_t plan_flags(uint32_t flags) {
  uint32_t f = flags & (NRF_FLAG_TRAIN | NRF_FLAG_WARP_JACOBIAN);
  if ((flags & NRF_FLAG_TRAIN) && (flags & NRF_FLAG_BF16)) f |= NRF_FLAG_BF16 | (flags & NRF_FLAG_WARP_F32);
  return f;
}

// Rows per workgroup tile of the float32 NeRF chain kernels for a launch over `ntiles` 64-row tiles: true = 32-row half tiles,
// four workgroups per CU (mlp_chain32.hip).  NRF_OPT_CHAIN_TILE_ROWS forces either.  Automatic = what the round-5 A/B measured
// (profiles/r05_chain32_ab.md): in steady state the 64-row kernels win by 3-5 % (forward 130 vs 123.5 TF, reverse 129 vs 125,
// eval forward 137 vs 132: every B operand float feeds one MFMA instead of two), but a launch that cannot fill the 64-row grid
// twice over -- fewer than two tiles per workgroup slot, e.g. one GPU's 128-ray share of a 1024-ray batch: 128 + 384 tiles for
// 512 slots -- runs 12-34 % faster on half tiles (coarse forward 0.160 -> 0.106 ms, fine 0.301 -> 0.264 ms).  The reverse
// chain never won (0.303 -> 0.327 ms at 512 tiles): its automatic choice stays 64.
bool chain32_for(const nrf_handle_s* h, int ntiles, bool reverse = false) {
  if (h->chain_rows_opt == 32) return true;
  if (h->chain_rows_opt == 64) return false;
  return !reverse && ntiles < AUTO32_FWD_BELOW_TILES_PER_CU * h->num_cus;
}

void build_plan(nrf_handle h, int B, uint32_t flags, int bgN = 0, int elastic = 0) {
  WsPlan& p = h->plan;
  flags = plan_flags(flags);
  if (p.B == B && p.flags == flags && p.bgN == bgN && p.elastic == elastic && p.chain_rows_opt == h->chain_rows_opt &&
      p.bf16_wgrad_merge == h->bf16_wgrad_merge) return;
  const nrf_model_desc& d = h->d;
  const bool train = flags & NRF_FLAG_TRAIN;
  const bool bft = train && (flags & NRF_FLAG_BF16);   // bf16 training: the NeRF MLPs stash / differentiate in bfloat16
  const bool jac = (flags & NRF_FLAG_WARP_JACOBIAN) && h->warp;   // tangent pass in an inference plan
  const bool bfw = bft && h->warp && !(flags & NRF_FLAG_WARP_F32);   // ... and so does the SE3 trunk (warp_bf16.hip)
  const bool wstash = (train && !bfw) || jac;                      // the fp32 warp kernels keep their input / sign-bit stash
  static std::atomic<uint64_t> next_serial{1};   // handles may be planned from several host threads
  p = WsPlan();
  p.serial = next_serial++;
  p.B = B;
  p.flags = flags;
  p.bgN = bgN;
  p.bfw = bfw;
  p.elastic = elastic;
  p.chain_rows_opt = h->chain_rows_opt;
  p.bf16_wgrad_merge = h->bf16_wgrad_merge;
  p.S[0] = d.num_coarse_samples;
  p.S[1] = d.num_coarse_samples + d.num_fine_samples;
  p.S[BG] = 1;
  p.S[TG] = 1;
  for (int lv = 0; lv < 3; ++lv) {
    p.rows[lv] = lv == BG ? bgN : B * p.S[lv];
    p.ntiles[lv] = (p.rows[lv] + TILE_ROWS - 1) / TILE_ROWS;
  }
  {   // the reverse chain's tiling is part of the plan (the reduce table sums one bias partial per workgroup of that launch);
      // the 32-row reverse kernel has no d-points path: models with a warp field keep the 64-row one
    int nt_mlp = 0;
    for (int q = 0; q < h->nlevels; ++q) nt_mlp += p.ntiles[q];
    p.bwd32 = train && !bft && !h->warp && chain32_for(h, nt_mlp, true);
  }
  p.tg_tiles_per = jac ? p.ntiles[h->nlevels - 1] : elastic ? p.ntiles[0] : 0;   // Jacobian output: levels run one after the other
  p.ntiles[TG] = 3 * p.tg_tiles_per;
  p.rows[TG] = p.ntiles[TG] * TILE_ROWS;
  const int G = h->num_cus;

  // ---- wgrad groups (training) ----
  struct GroupSpec { int lv; int xk; size_t* xoff; int xstride; int kvalid; int Kb; int yk; size_t* yoff; int ystride; int Nb;
                     int vec; int64_t dst; int dst_ld; int rows; int cols; int units; size_t xadd, yadd;
                     size_t* vecoff = nullptr; int accumulate = 0;
                     size_t* vecoff2 = nullptr; int64_t dst2 = -1; };   // second vector column set against the same X
  std::vector<GroupSpec> specs;
  const int Kb_pe = (h->PK + 31) / 32;          // posenc stash tiles hold whole 32-feature blocks
  const int PKS = Kb_pe * 32;
  // SE3 trunk + heads of level `lv` (coarse / fine samples, or the background-point batch)
  auto add_warp_groups = [&](int lv, int accu) {
    LevelWs& L = p.L[lv];
    const WarpParamOffsets& w = h->wpo;
    const size_t wl = (size_t)p.ntiles[lv] * FRAG_TILE_128;
    const int Kb_in = (h->PKw + 31) / 32;
    auto push = [&](GroupSpec g) { g.accumulate = accu; specs.push_back(g); };
    for (int l = 0; l < WARP_DEPTH; ++l) {
      if (l == 0) {
        push({lv, SRC_PLAIN, &L.w_st_win, Kb_in * 32 * TILE_ROWS, h->Win, Kb_in, SRC_FRAG128, &L.w_dy, FRAG_TILE_128, 4, 0,
              w.trunk_k[0], WARP_W, h->Win, WARP_W, Kb_in * 4, 0, 0});
      } else {
        push({lv, SRC_FRAG128, &L.w_st_h, FRAG_TILE_128, WARP_W, 4, SRC_FRAG128, &L.w_dy, FRAG_TILE_128, 4, 0,
              w.trunk_k[l], WARP_W, WARP_W, WARP_W, 16, (size_t)(l - 1) * wl, (size_t)l * wl});
        if (l == WARP_SKIP)
          push({lv, SRC_PLAIN, &L.w_st_win, Kb_in * 32 * TILE_ROWS, h->Win, Kb_in, SRC_FRAG128, &L.w_dy, FRAG_TILE_128, 4, 0,
                w.trunk_k[l] + (int64_t)WARP_W * WARP_W, WARP_W, h->Win, WARP_W, Kb_in * 4, 0, (size_t)l * wl});
      }
    }
    GroupSpec gw = {lv, SRC_FRAG128, &L.w_st_h, FRAG_TILE_128, WARP_W, 4, 0, nullptr, 0, 0, 3,
                    w.w_k, 3, WARP_W, 3, 6, (size_t)(WARP_DEPTH - 1) * wl, 0};
    gw.vecoff = &L.w_dw4;
    gw.vecoff2 = &L.w_dv4; gw.dst2 = w.v_k;   // both heads read h6: one pass over its stash
    push(gw);
  };
  // bf16 training: the NeRF MLP groups go to the bf16 wgrad kernel (X / dY = bf16 stash buffers of Kb / Nb blocks per
  // 32-sample group); bias = the group also owns the bias gradient (column sums of its dY)
  struct BSpec { int lv; size_t* xoff; size_t xadd; int Kb; size_t* yoff; size_t yadd; int Nb;
                 int64_t dst; int dst_ld, rows, cols, col0;          // weight leaf <- slab[0:rows][col0:col0+cols]
                 int64_t bias_dst; int bias_cols;                     // bias leaf <- column sums [0:bias_cols], or -1
                 int64_t bias2_dst; int bias2_col0;                   // a second bias leaf (alpha: column 3; SE3 v head: columns 3..5), or -1
                 int bias2_cols = 1; int accu = 0; int ngroups = 0;   // reduce pass the leaf is added in; groups (0: the MLP level's)
                 int64_t dst2 = -1; int col20 = 0;                    // a second weight leaf from the same slab (SE3 v head), or -1
                 int dst2_ld = 0, dst2_cols = 0;                      // ... of its own width (0: as the first leaf)
                 // an operand assembled from two stash buffers (WgradGroup x2_off / dy2_off): the last Kb2 / Nb2 blocks
                 size_t* x2off = nullptr; int Kb2 = 0, x2_blocks = 0; size_t* y2off = nullptr; int Nb2 = 0, y2_blocks = 0; };
  std::vector<BSpec> bspecs;
  if (bft) {
    for (int lv = 0; lv < h->nlevels; ++lv) {
      LevelWs& L = p.L[lv];
      const MlpParamOffsets& po = h->po[lv];
      L.b_ngroups = (p.rows[lv] + 255) / 256 * 8;
      const size_t layer = (size_t)L.b_ngroups * 8 * BF_BLOCK_DW;
      auto bpush = [&](size_t* xoff, size_t xadd, int Kb, size_t* yoff, size_t yadd, int Nb, int64_t dst, int dst_ld, int rows, int cols,
                       int col0, int64_t bias_dst, int bias_cols, int64_t bias2_dst = -1, int bias2_col0 = 0) {
        bspecs.push_back({lv, xoff, xadd, Kb, yoff, yadd, Nb, dst, dst_ld, rows, cols, col0, bias_dst, bias_cols, bias2_dst, bias2_col0});
      };
      for (int l = 0; l < TRUNK_DEPTH; ++l) {
        if (l == 0) {
          bpush(&L.b_pe, 0, 2, &L.b_dy, 0, 8, po.trunk_k[0], 256, h->P, 256, 0, po.trunk_b[0], 256);
        } else {
          if (l == d.nerf_skip_layer && h->bf16_wgrad_merge) {
            // (NRF_OPT_BF16_WGRAD_MERGE) the skip layer's kernel is [256 + P, 256]: rows 0..255 multiply h4, rows 256.. the posenc (modules.py:47-48).  ONE group,
            // X = [h4 (8 blocks) | posenc (2 blocks)] against dpre_4, so dpre_4 is streamed once (rounds 2-4: two groups, twice)
            bpush(&L.b_h, (size_t)(l - 1) * layer, 10, &L.b_dy, (size_t)l * layer, 8, po.trunk_k[l], 256, 256 + h->P, 256, 0, po.trunk_b[l], 256);
            bspecs.back().x2off = &L.b_pe; bspecs.back().Kb2 = 2; bspecs.back().x2_blocks = 2;
          } else {
            bpush(&L.b_h, (size_t)(l - 1) * layer, 8, &L.b_dy, (size_t)l * layer, 8, po.trunk_k[l], 256, 256, 256, 0, po.trunk_b[l], 256);
            if (l == d.nerf_skip_layer)   // merge off: the posenc rows of the skip layer as a group of their own (dpre_4 read twice)
              bpush(&L.b_pe, 0, 2, &L.b_dy, (size_t)l * layer, 8, po.trunk_k[l] + 256 * 256, 256, h->P, 256, 0, -1, 0);
          }
        }
      }
      const bool merge_alpha = h->bf16_wgrad_merge && h->A == 0;
      if (!merge_alpha) {
        bpush(&L.b_h, (size_t)7 * layer, 8, &L.b_dbn, 0, 8, po.bn_k, 256, 256, 256, 0, po.bn_b, 256);
      } else {
        // the bottleneck AND the alpha head read h8 (modules.py:149-157): ONE group, dY = [d bottleneck (8 blocks) | d raw (block 0 of
        // the small stash)], h8 streamed once; slab column 256 + 3 (d raw sigma) is the alpha kernel's gradient
        bpush(&L.b_h, (size_t)7 * layer, 8, &L.b_dbn, 0, 9, po.bn_k, 256, 256, 256, 0, po.bn_b, 256);
        BSpec& m = bspecs.back();
        m.y2off = &L.b_dsmall; m.Nb2 = 1; m.y2_blocks = 2;
        m.dst2 = po.alpha_k; m.col20 = 256 + 3; m.dst2_ld = 1; m.dst2_cols = 1;
      }
      bpush(&L.b_bn, 0, 8, &L.b_drgbh, 0, 4, po.rgbh_k, 128, 256, 128, 0, po.rgbh_b, 128);
      // narrow heads against the "small" dY block: columns 0..2 = d rgb logits (X = rgb hidden), column 3 = d raw sigma (X = h8)
      bpush(&L.b_rgbh, 0, 4, &L.b_dsmall, 0, 2, po.logit_k, 3, 128, 3, 0, po.logit_b, 3, po.alpha_b, 3);
      if (h->A > 0) bpush(&L.b_bn, 0, 8, &L.b_dsmall, 0, 2, po.alpha_k, 1, 256, 1, 3, -1, 0);   // use_alpha_condition: X = the bottleneck
      else if (!merge_alpha) bpush(&L.b_h, (size_t)7 * layer, 8, &L.b_dsmall, 0, 2, po.alpha_k, 1, 256, 1, 3, -1, 0);
      // (merged: the alpha head rides in the bottleneck's group above)
    }
  }
  // bf16 SE3 trunk: every pass through the field (coarse / fine samples, background points, the 3 tangents per coarse sample)
  // leaves its own X / dY stash; all of them add into the same leaves (reduce passes 0..3).  The tangent pass carries no bias.
  if (bfw) {
    const WarpParamOffsets& w = h->wpo;
    auto add_bf_warp = [&](int lv, int accu, bool tangent) {
      LevelWs& L = p.L[lv];
      const int rows = tangent ? p.rows[0] : p.rows[lv];
      L.bw_ngroups = (tangent ? 3 : 1) * ((rows + 255) / 256 * 8);
      const size_t layer = (size_t)L.bw_ngroups * 4 * BF_BLOCK_DW;
      auto wpush = [&](size_t* xoff, size_t xadd, int Kb, size_t* yoff, size_t yadd, int Nb, int64_t dst, int dst_ld, int rws, int cols,
                       int64_t bias_dst, int bias_cols) {
        BSpec b = {lv, xoff, xadd, Kb, yoff, yadd, Nb, dst, dst_ld, rws, cols, 0, tangent ? -1 : bias_dst, bias_cols, -1, 0};
        b.accu = accu; b.ngroups = L.bw_ngroups;
        bspecs.push_back(b);
      };
      for (int l = 0; l < WARP_DEPTH; ++l) {
        if (l == 0) {
          wpush(&L.bw_in, 0, 2, &L.bw_dy, 0, 4, w.trunk_k[0], WARP_W, h->Win, WARP_W, w.trunk_b[0], WARP_W);
        } else {
          wpush(&L.bw_h, (size_t)(l - 1) * layer, 4, &L.bw_dy, (size_t)l * layer, 4, w.trunk_k[l], WARP_W, WARP_W, WARP_W, w.trunk_b[l], WARP_W);
          if (l == WARP_SKIP)
            wpush(&L.bw_in, 0, 2, &L.bw_dy, (size_t)l * layer, 4, w.trunk_k[l] + (int64_t)WARP_W * WARP_W, WARP_W, h->Win, WARP_W, -1, 0);
        }
      }
      // both heads read h6 against the "small" dY block: columns 0..2 = dL/dw, 3..5 = dL/dv
      BSpec hd = {lv, &L.bw_h, (size_t)(WARP_DEPTH - 1) * layer, 4, &L.bw_dhead, 0, 2, w.w_k, 3, WARP_W, 3, 0,
                  tangent ? -1 : w.w_b, 3, tangent ? -1 : w.v_b, 3};
      hd.bias2_cols = 3; hd.accu = accu; hd.ngroups = L.bw_ngroups; hd.dst2 = w.v_k; hd.col20 = 3;
      bspecs.push_back(hd);
    };
    for (int lv = 0; lv < h->nlevels; ++lv) add_bf_warp(lv, lv > 0 ? 1 : 0, false);
    if (bgN > 0) add_bf_warp(BG, 2, false);
    if (elastic) add_bf_warp(TG, 3, true);
  }
  if (train) {
    for (int lv = 0; lv < h->nlevels; ++lv) {
      LevelWs& L = p.L[lv];
      const MlpParamOffsets& po = h->po[lv];
      const size_t layer = (size_t)p.ntiles[lv] * FRAG_TILE_256;
      for (int l = 0; l < TRUNK_DEPTH && !bft; ++l) {
        if (l == 0) {
          specs.push_back({lv, SRC_PLAIN, &L.st_pe, PKS * TILE_ROWS, h->P, Kb_pe, SRC_FRAG256, &L.dy_trunk, FRAG_TILE_256, 8, 0,
                           po.trunk_k[0], 256, h->P, 256, Kb_pe * 8, 0, 0});
        } else {
          specs.push_back({lv, SRC_FRAG256, &L.st_h, FRAG_TILE_256, 256, 8, SRC_FRAG256, &L.dy_trunk, FRAG_TILE_256, 8, 0,
                           po.trunk_k[l], 256, 256, 256, 64, (size_t)(l - 1) * layer, (size_t)l * layer});
          if (l == d.nerf_skip_layer)
            specs.push_back({lv, SRC_PLAIN, &L.st_pe, PKS * TILE_ROWS, h->P, Kb_pe, SRC_FRAG256, &L.dy_trunk, FRAG_TILE_256, 8, 0,
                             po.trunk_k[l] + 256 * 256, 256, h->P, 256, Kb_pe * 8, 0, (size_t)l * layer});
        }
      }
      if (!bft) {
      specs.push_back({lv, SRC_FRAG256, &L.st_h, FRAG_TILE_256, 256, 8, SRC_FRAG256, &L.dy_bn, FRAG_TILE_256, 8, 0,
                       po.bn_k, 256, 256, 256, 64, (size_t)7 * layer, 0});
      specs.push_back({lv, SRC_FRAG256, &L.st_bn, FRAG_TILE_256, 256, 8, SRC_FRAG128, &L.dy_rgbh, FRAG_TILE_128, 4, 0,
                       po.rgbh_k, 128, 256, 128, 32, 0, 0});
      // narrow heads on the VALU: alpha (X = h8, vec.w) and rgb logits (X = rgb hidden, vec.xyz)
      if (h->A > 0)   // use_alpha_condition: the alpha head reads the bottleneck
        specs.push_back({lv, SRC_FRAG256, &L.st_bn, FRAG_TILE_256, 256, 8, 0, nullptr, 0, 0, 1, po.alpha_k, 1, 256, 1, 12, 0, 0});
      else
        specs.push_back({lv, SRC_FRAG256, &L.st_h, FRAG_TILE_256, 256, 8, 0, nullptr, 0, 0, 1,
                         po.alpha_k, 1, 256, 1, 12, (size_t)7 * layer, 0});
      specs.push_back({lv, SRC_FRAG128, &L.st_rgbh, FRAG_TILE_128, 128, 4, 0, nullptr, 0, 0, 3,
                       po.logit_k, 3, 128, 3, 6, 0, 0});
      }
      if (h->warp && !bfw) add_warp_groups(lv, lv > 0 ? 1 : 0);   // the field is shared by both passes: level 1 accumulates
    }
    if (h->warp && !bfw && bgN > 0) add_warp_groups(BG, 2);
    if (h->warp && !bfw && elastic) add_warp_groups(TG, 3);   // tangent activations x tangent adjoints, same leaves
  }

  // ---- float layout ----
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o = align_up(o + n, ALIGN_F); return r; };
  // ---- stream-K partition of the wgrad work: equal cost per workgroup, one workgroup per CU ----
  // cost of one 64-row tile of a group, in units of a full 256x256 layer tile; the narrow groups
  // are staging/latency bound, so they are charged more than their MFMA share.
  auto env_cost = [](const char* name, double dflt) {   // calibration overrides (scripts/wgrad_calib.py): experiment builds only
#ifdef NRF_EXPERIMENT
    if (const char* e = getenv(name)) return atof(e);
#endif
    (void)name;
    return dflt;
  };
  // measured with scripts/wgrad_calib.py / wgrad_calib_vrig.py (per-segment wall clocks, least squares), relative to a
  // 256x256 tile; round 3 (asm LDS-DMA + 160 KiB ring: the narrow groups are no longer latency-bound): 8x8 = 14.8 us
  const double c_vec256 = env_cost("NRF_COST_VEC256", 0.105), c_vec128 = env_cost("NRF_COST_VEC128", 0.094),
               c_vec128x2 = env_cost("NRF_COST_VEC128X2", 0.150),   // SE3 heads: two vectors against one pass over h6
               c_pe = env_cost("NRF_COST_PE", 0.287),               // 2 x 8 blocks: posenc rows of the NeRF trunk
               c_rgbh = env_cost("NRF_COST_RGBH", 0.533),           // 8 x 4
               c_44 = env_cost("NRF_COST_44", 0.285),               // 4 x 4: SE3 trunk layers
               c_pe128 = env_cost("NRF_COST_PE128", 0.156),         // 2 x 4: SE3 trunk input rows
               c_seg = env_cost("NRF_COST_SEG", 0.5);   // fixed cost of opening a segment (pipeline fill + slab flush), in tiles
  auto tile_cost = [&](const GroupSpec& sp) -> double {
    if (sp.Nb == 0) return sp.Kb == 8 ? c_vec256 : sp.vecoff2 ? c_vec128x2 : c_vec128;   // vector columns only (VALU + HBM stream)
    if (sp.Nb == 8) return sp.Kb >= 5 ? 1.0 : c_pe;
    return sp.Kb >= 5 ? c_rgbh : sp.Kb >= 3 ? c_44 : c_pe128;
  };
  std::vector<int> nsplit(specs.size(), 0);
  if (!specs.empty()) {
    double total = 0;
    for (auto& sp : specs) total += tile_cost(sp) * p.ntiles[sp.lv];
    const int nwg = G;
    total += c_seg * (nwg + (double)specs.size());   // every workgroup and every group boundary opens a segment
    const double quota = total / nwg;
    p.seg_begin.assign(1, 0);
    int w = 0;
    double room = quota;
    for (size_t gi = 0; gi < specs.size(); ++gi) {
      const double c = tile_cost(specs[gi]);
      int t0 = 0;
      const int nt = p.ntiles[specs[gi].lv];
      while (t0 < nt) {
        int take_n = (int)floor((room - c_seg) / c + 1e-9);
        if (take_n <= 0 && w < nwg - 1) {            // this workgroup is full: move on
          p.seg_begin.push_back((int)p.segs.size());
          ++w; room += quota;
          continue;
        }
        if (take_n <= 0) take_n = nt - t0;           // last workgroup absorbs rounding leftovers
        if (w == nwg - 1) take_n = nt - t0;
        if (take_n > nt - t0) take_n = nt - t0;
        p.segs.push_back({(int)gi, t0, t0 + take_n, nsplit[gi]});
        nsplit[gi] += 1;
        t0 += take_n;
        room -= c_seg + take_n * c;
      }
    }
    while ((int)p.seg_begin.size() < nwg + 1) p.seg_begin.push_back((int)p.segs.size());
    p.wgrad_nwg = nwg;
  }
  // ---- the same stream-K cut for the bf16 groups: HBM-bound, cost = blocks streamed per 32-sample group ----
  std::vector<int> bnsplit(bspecs.size(), 0);
  if (!bspecs.empty()) {
    // a chunk costs (Kb + Nb) + a fixed term, in block units (2 KiB streamed).  Round 2 measured + 12 on config A (the per-chunk
    // barrier and HBM latency worth 24 KiB of streaming: wgrad 0.87 ms with a pure byte model, 0.61 ms with that one).  Round 6,
    // after the copies moved to per-segment SGPR tables (wgrad_bf16.hip): ALONE every shape streams 5.6-6.4 TB/s, i.e. cost ~ bytes
    // (scripts/micro/wgrad_bf16_bench.hip), but IN the mixed launch a byte-proportional model is 3-10 % slower than + 12, and the
    // narrow shapes (Kb + Nb <= 8: the SE3 trunk's 16 / 12 KiB chunks) are best charged + 8: swept on config D / vrig / A (bf16) at
    // narrow = 12 / 8 / 5 / 2: 1.19 / 1.13 / 1.18 / 1.38 ms, 0.92 / 0.84 / 0.90 / 1.03 ms, 0.456 / 0.460 / 0.495 / 0.618 ms
    // (profiles/r06_experiments.md section 3)
    const double bc_seg = env_cost("NRF_BCOST_SEG", 16.0);   // opening a segment (pipeline fill + 256 KiB slab flush), in block units
    const double bc_chunk = env_cost("NRF_BCOST_CHUNK", 12.0);   // per-chunk fixed cost (barrier + issue), in block units
    // the two merged shapes (10 x 8, 8 x 9: ten accumulator blocks per wave, five copies per wave and chunk) cost more per chunk
    // than their bytes: with a byte-proportional cost the kernel was 10 % SLOWER although it fetched 10 % less (the workgroups
    // inside the merged groups ran ~1.35 x their quota); swept on the GPU at +0 / 6 / 10 / 16 / 24 / 32 units: 0.555 / 0.508 /
    // 0.500 / 0.520 / 0.527 / 0.543 ms
    const double bc_merged = env_cost("NRF_BCOST_MERGED", 10.0);
    const double bc_chunk_narrow = env_cost("NRF_BCOST_CHUNK_NARROW", 8.0);   // ... of the 4 x 4 / 2 x 4 shapes (SE3 trunk: 16 / 12 KiB chunks)
    const double bc_quad = env_cost("NRF_BCOST_QUAD", 0.0);   // per accumulator block (Kb x Nb): the MFMA / operand-read side of a chunk
    auto bcost = [&](const BSpec& sp) {
      // Kb / Nb include the second source's blocks
      return (double)(sp.Kb + sp.Nb) + (sp.Kb + sp.Nb <= 8 ? bc_chunk_narrow : bc_chunk) + bc_quad * sp.Kb * sp.Nb + ((sp.Kb2 || sp.Nb2) ? bc_merged : 0.0);
    };
    double total = 0;
    auto bng = [&](const BSpec& sp) { return sp.ngroups ? sp.ngroups : p.L[sp.lv].b_ngroups; };
    for (auto& sp : bspecs) total += bcost(sp) * bng(sp);
    const int nwg = G;
    total += bc_seg * (nwg + (double)bspecs.size());
    const double quota = total / nwg;
    p.bseg_begin.assign(1, 0);
    int w = 0;
    double room = quota;
    for (size_t gi = 0; gi < bspecs.size(); ++gi) {
      const double c = bcost(bspecs[gi]);
      int t0 = 0;
      const int nt = bng(bspecs[gi]);
      while (t0 < nt) {
        int take_n = (int)floor((room - bc_seg) / c + 1e-9);
        if (take_n <= 0 && w < nwg - 1) {
          p.bseg_begin.push_back((int)p.bsegs.size());
          ++w; room += quota;
          continue;
        }
        if (take_n <= 0 || w == nwg - 1 || take_n > nt - t0) take_n = nt - t0;
        p.bsegs.push_back({(int)gi, t0, t0 + take_n, bnsplit[gi]});
        bnsplit[gi] += 1;
        t0 += take_n;
        room -= bc_seg + take_n * c;
      }
    }
    while ((int)p.bseg_begin.size() < nwg + 1) p.bseg_begin.push_back((int)p.bsegs.size());
    p.bwgrad_nwg = nwg;
  }
  p.ntasks = (int)p.segs.size();
  if (h->embed) {
    p.iparams = take((size_t)h->nparams);
    if (train) p.igrad = take((size_t)h->nparams);
  }
  p.bfpack.clear();
  if (!train || bft) {   // weight streams of the bf16 chains (mlp_bf16.hip): chunks (panels) in execution order
    for (int lv = 0; lv < h->nlevels; ++lv) {
      const MlpParamOffsets& po = h->po[lv];
      p.L[lv].bf_wpk = take((size_t)BF_FWD_STREAM_KB * 256);   // KiB -> floats
      size_t at = 0;   // floats from the level's stream base
      size_t base = p.L[lv].bf_wpk;
      int tr = 0;
      // One GEMM = nblocks / pb panels; a panel (chunk) = [row][block of the panel][lane] x 16 B, rows = [bias row,] then the
      // k-step rows of each input part (leaf, ld, row0, valid K, input blocks)
      struct Part { int64_t leaf; int ld, row0, krows, nin; };
      auto gemm = [&](int pb, int nblocks, int ncols, int64_t bias, std::initializer_list<Part> parts) {
        for (int pn = 0; pn < nblocks / pb; ++pn) {
          int row = 0;
          auto emit = [&](int kind, int64_t src, int ld, int row0, int krows, int nrows) {
            RcPackDesc e;
            memset(&e, 0, sizeof(e));
            e.src_off = src; e.dst_off = (long long)(base + at + (size_t)row * pb * 256); e.kind = kind; e.src_ld = ld; e.row0 = row0;
            e.krows = krows; e.ncols = ncols; e.ngroups = nrows; e.nout = pb; e.nout_panel = pb; e.o0 = 0; e.transposed = tr;
            e.oblk0 = pn * pb;
            p.bfpack.push_back(e);
            row += nrows;
          };
          if (bias >= 0) emit(1, bias, 0, 0, 0, 1);
          else if (bias == -2) emit(2, 0, 0, 0, 0, 1);   // a zero row where the kernel runs a bias-style k-step this model does not use
          for (const Part& q : parts) emit(0, q.leaf, q.ld, q.row0, q.krows, 2 * q.nin);
          at += (size_t)row * pb * 256;
        }
      };
      gemm(2, 8, TRUNK_W, po.trunk_b[0], {{po.trunk_k[0], TRUNK_W, 0, h->P, 2}});
      for (int l = 1; l < TRUNK_DEPTH; ++l) {
        if (l == SKIP_LAYER) gemm(2, 8, TRUNK_W, po.trunk_b[l], {{po.trunk_k[l], TRUNK_W, 0, TRUNK_W, 8}, {po.trunk_k[l], TRUNK_W, TRUNK_W, h->P, 2}});
        else gemm(2, 8, TRUNK_W, po.trunk_b[l], {{po.trunk_k[l], TRUNK_W, 0, TRUNK_W, 8}});
      }
      gemm(2, 8, TRUNK_W, po.bn_b, {{po.bn_k, TRUNK_W, 0, TRUNK_W, 8}});        // bottleneck
      gemm(1, 1, 1, po.alpha_b, {{po.alpha_k, 1, 0, TRUNK_W, 8}});               // alpha head: one block, column 0
      gemm(2, 4, RGB_W, -1, {{po.rgbh_k, RGB_W, 0, TRUNK_W, 8}});                // rgb hidden (bias: the fp32 per-ray term)
      gemm(1, 1, 3, po.logit_b, {{po.logit_k, 3, 0, RGB_W, 4}});                 // rgb logits: one block, columns 0..2
      p.bf_stream_ok = at == (size_t)BF_FWD_STREAM_KB * 256;
      if (bft) {
        // dgrad stream (nerf_mlp_bwd_bf16_kernel): A = W as stored, [m = the layer's input feature][k = its output feature];
        // ncols = valid M, Part.krows = valid K
        p.L[lv].bf_wpkT = take((size_t)(h->warp ? BF_BWD_STREAM_DPTS_KB : BF_BWD_STREAM_KB) * 256);
        base = p.L[lv].bf_wpkT; at = 0; tr = 1;
        gemm(4, 4, RGB_W, -1, {{po.logit_k, 3, 0, 3, 1}});                       // G1: one k-step (3 valid) + a zero one, 4 blocks
        // the alpha head's transpose is ONE bias-style row (w_alpha[0:256], B = d sigma) in the GEMM that produces the gradient of
        // its input: the trunk output (G3), or -- use_alpha_condition, modules.py:152-157 -- the bottleneck (G2); zeros in the other
        const int64_t arow = po.alpha_k;
        gemm(2, 8, TRUNK_W, h->A > 0 ? arow : -2, {{po.rgbh_k, RGB_W, 0, RGB_W, 4}});             // G2: rows 0..255 of [256+R, 128]
        gemm(2, 8, TRUNK_W, h->A > 0 ? -2 : arow, {{po.bn_k, TRUNK_W, 0, TRUNK_W, 8}});           // G3
        for (int l = TRUNK_DEPTH - 1; l >= 1; --l) gemm(2, 8, TRUNK_W, -1, {{po.trunk_k[l], TRUNK_W, 0, TRUNK_W, 8}});
        if (h->warp) {   // d posenc: W0 and the skip layer's posenc rows as A [m = posenc feature (P valid)][k = output feature]
          gemm(2, 2, h->P, -1, {{po.trunk_k[0], TRUNK_W, 0, TRUNK_W, 8}});
          gemm(2, 2, h->P, -1, {{po.trunk_k[d.nerf_skip_layer], TRUNK_W, TRUNK_W, TRUNK_W, 8}});
        }
        p.bf_stream_ok = p.bf_stream_ok && at == (size_t)(h->warp ? BF_BWD_STREAM_DPTS_KB : BF_BWD_STREAM_KB) * 256;
      }
    }
    if (h->warp) {   // bf16 SE3 trunk (warp_bf16.hip): forward stream (also for bf16 inference), reverse stream (training)
      const WarpParamOffsets& w = h->wpo;
      size_t at = 0, base = 0;
      int tr = 0;
      struct Part { int64_t leaf; int ld, row0, krows, nin; int64_t leaf2 = -1; int split = 0; };
      // as the NeRF gemm() above; bias2 / Part.leaf2: the second of two leaves side by side (heads w | v)
      auto gemm = [&](int pb, int nblocks, int ncols, int64_t bias, int64_t bias2, int bsplit, std::initializer_list<Part> parts) {
        for (int pn = 0; pn < nblocks / pb; ++pn) {
          int row = 0;
          auto emit = [&](int kind, int64_t src, int64_t src2, int split, int ld, int row0, int krows, int nrows) {
            RcPackDesc e;
            memset(&e, 0, sizeof(e));
            e.src_off = src; e.dst_off = (long long)(base + at + (size_t)row * pb * 256); e.kind = kind; e.src_ld = ld; e.row0 = row0;
            e.krows = krows; e.ncols = ncols; e.ngroups = nrows; e.nout = pb; e.nout_panel = pb; e.o0 = 0; e.transposed = tr;
            e.oblk0 = pn * pb; e.src_off2 = src2 >= 0 ? src2 : 0; e.split = src2 >= 0 ? split : 0;
            p.bfpack.push_back(e);
            row += nrows;
          };
          if (bias >= 0) emit(1, bias, bias2, bsplit, 0, 0, 0, 1);
          for (const Part& q : parts) emit(0, q.leaf, q.leaf2, q.split, q.ld, q.row0, q.krows, 2 * q.nin);
          at += (size_t)row * pb * 256;
        }
      };
      p.bfw_wpk = take((size_t)BFW_FWD_STREAM_KB * 256);
      base = p.bfw_wpk;
      gemm(2, 4, WARP_W, w.trunk_b[0], -1, 0, {{w.trunk_k[0], WARP_W, 0, h->Win, 2}});
      for (int l = 1; l < WARP_DEPTH; ++l) {
        if (l == WARP_SKIP) gemm(2, 4, WARP_W, w.trunk_b[l], -1, 0, {{w.trunk_k[l], WARP_W, 0, WARP_W, 4}, {w.trunk_k[l], WARP_W, WARP_W, h->Win, 2}});
        else gemm(2, 4, WARP_W, w.trunk_b[l], -1, 0, {{w.trunk_k[l], WARP_W, 0, WARP_W, 4}});
      }
      gemm(1, 1, 6, w.w_b, w.v_b, 3, {{w.w_k, 3, 0, WARP_W, 4, w.v_k, 3}});     // heads: columns 0..2 = w, 3..5 = v
      p.bf_stream_ok = p.bf_stream_ok && at == (size_t)BFW_FWD_STREAM_KB * 256;
      if (bfw) {
        // reverse stream: A = W as stored, [m = the layer's input feature][k = its output feature]
        p.bfw_wpkT = take((size_t)BFW_BWD_STREAM_KB * 256);
        base = p.bfw_wpkT; at = 0; tr = 1;
        gemm(4, 4, WARP_W, -1, -1, 0, {{w.w_k, 3, 0, 6, 1, w.v_k, 3}});         // heads^T: K = (w0..2, v0..2) of one k-step + a zero one
        for (int l = WARP_DEPTH - 1; l >= 1; --l) gemm(2, 4, WARP_W, -1, -1, 0, {{w.trunk_k[l], WARP_W, 0, WARP_W, 4}});
        gemm(2, 2, h->Win, -1, -1, 0, {{w.trunk_k[0], WARP_W, 0, WARP_W, 4}});                  // C0: d input through layer 0
        gemm(2, 2, h->Win, -1, -1, 0, {{w.trunk_k[WARP_SKIP], WARP_W, WARP_W, WARP_W, 4}});     // C4: ... through the skip rows
        p.bf_stream_ok = p.bf_stream_ok && at == (size_t)BFW_BWD_STREAM_KB * 256;
      }
    }
    p.bf_desc = take(p.bfpack.size() * sizeof(RcPackDesc) / 4 + 16);
  }

  auto alloc_warp = [&](LevelWs& L, size_t nt) {
    L.wpoints = take(nt * TILE_ROWS * 3);
    L.points_raw = take(nt * TILE_ROWS * 3);
    if (wstash) {
      L.w_st_win = take(nt * ((h->PKw + 31) / 32 * 32) * TILE_ROWS);
      L.w_st_h = take(nt * FRAG_TILE_128 * WARP_DEPTH);
      L.w_st_wv = take(nt * TILE_ROWS * 8);
      L.w_bits = take(nt * 4 * 64 * WARP_DEPTH);
    }
    if (train && !bfw) {
      L.d_points = take(nt * TILE_ROWS * 3);
      L.w_dy = take(nt * FRAG_TILE_128 * WARP_DEPTH);
      L.w_dw4 = take(nt * TILE_ROWS * 4);
      L.w_dv4 = take(nt * TILE_ROWS * 4);
      L.w_small_part = take((size_t)4 * G * WARP_SMALL_PART);
    }
    if (bfw) {   // bf16 trunk: fp32 rows only for what exp_se3 / the elastic kernel read and write; the rest is the bf16 stash
      const size_t ng = L.bw_ngroups;
      L.w_st_wv = take(nt * TILE_ROWS * 8);
      L.d_points = take(nt * TILE_ROWS * 3);
      L.w_dw4 = take(nt * TILE_ROWS * 4);
      L.w_dv4 = take(nt * TILE_ROWS * 4);
      L.bw_in = take(ng * 2 * BF_BLOCK_DW);
      L.bw_h = take(ng * 4 * BF_BLOCK_DW * WARP_DEPTH);
      L.bw_bits = take(ng * 64 * 2 * WARP_DEPTH);
      L.bw_dy = take(ng * 4 * BF_BLOCK_DW * WARP_DEPTH);
      L.bw_dhead = take(ng * 2 * BF_BLOCK_DW);
    }
  };
  p.cond = take((size_t)B * (h->R > 0 ? h->R : 1));
  p.mse = take((size_t)2 * B);   // [level][ray] squared error
  p.zero_rgb = take((size_t)B * 3);
  for (int lv = 0; lv < h->nlevels; ++lv) {
    LevelWs& L = p.L[lv];
    const size_t nt = p.ntiles[lv];
    L.wpk = take(h->pk.total);
    L.z = take((size_t)p.rows[lv]);
    L.out4 = take(nt * TILE_ROWS * 4);
    L.rgb = take((size_t)B * 3);
    L.depth = take(B);
    L.med = take(B);
    L.acc = take(B);
    L.weights = take((size_t)p.rows[lv]);
    L.condterm = take((size_t)B * RGB_W);
    if (h->A > 0) { L.alpha_ct = take(B); L.dsig_ray = take(B); }
    if (bft) {   // bf16 stashes (nrf_internal.h BfStash), dwords
      const size_t ng = L.b_ngroups;
      L.b_pe = take(ng * 2 * BF_BLOCK_DW);
      L.b_h = take(ng * 8 * BF_BLOCK_DW * TRUNK_DEPTH);
      L.b_bn = take(ng * 8 * BF_BLOCK_DW);
      L.b_rgbh = take(ng * 4 * BF_BLOCK_DW);
      L.b_bits = take(ng * 64 * 4 * (TRUNK_DEPTH + 1));
      L.b_dy = take(ng * 8 * BF_BLOCK_DW * TRUNK_DEPTH);
      L.b_dbn = take(ng * 8 * BF_BLOCK_DW);
      L.b_drgbh = take(ng * 4 * BF_BLOCK_DW);
      L.b_dsmall = take(ng * 2 * BF_BLOCK_DW);
      L.d_raw4 = take(nt * TILE_ROWS * 4);
      L.dray = take((size_t)B * RGB_W);
      L.small_part = take((size_t)4 * G * SMALL_PART);   // up to four workgroups per CU (32-row reverse chain)
      L.cond_grad = take((size_t)(h->R > 0 ? h->R : 1) * RGB_W);
    } else if (train) {
      L.st_pe = take(nt * PKS * TILE_ROWS);
      L.st_h = take(nt * FRAG_TILE_256 * TRUNK_DEPTH);
      L.st_bn = take(nt * FRAG_TILE_256);
      L.st_rgbh = take(nt * FRAG_TILE_128);
      L.bits_trunk = take(nt * 4 * 128 * TRUNK_DEPTH);
      L.bits_rgbh = take(nt * 4 * 64);
      L.d_raw4 = take(nt * TILE_ROWS * 4);
      L.dy_trunk = take(nt * FRAG_TILE_256 * TRUNK_DEPTH);
      L.dy_bn = take(nt * FRAG_TILE_256);
      L.dy_rgbh = take(nt * FRAG_TILE_128);
      L.dray = take((size_t)B * RGB_W);
      L.small_part = take((size_t)4 * G * SMALL_PART);   // up to four workgroups per CU (32-row reverse chain)
      L.cond_grad = take((size_t)(h->R > 0 ? h->R : 1) * RGB_W);
    }
    if (h->warp) alloc_warp(L, nt);
  }
  if (h->warp && bgN > 0) {
    alloc_warp(p.L[BG], p.ntiles[BG]);
    p.bg_loss = take(64);
    p.bg_points = take((size_t)bgN * 3);   // the library's own draw (nrf_background.warp_ids == NULL): noised points, ids
    p.bg_ids = take((size_t)bgN);
  }
  if (h->time_enc) {
    p.t_codes = take((size_t)B * h->G);
    if (train) {
      p.t_dcodes = take((size_t)B * h->G);
      p.t_in = take((size_t)B * TIME_MAX_IN);
      p.t_h = take((size_t)B * TIME_DEPTH * TIME_W);
      p.t_dpre = take((size_t)B * TIME_DEPTH * TIME_W);
    }
  }
  if (jac && !train) alloc_warp(p.L[TG], p.ntiles[TG]);
  if (h->warp && train) p.wr_sums = take(64);
  if (h->warp && elastic && train) {
    alloc_warp(p.L[TG], p.ntiles[TG]);
    p.L[0].el_dw4 = take((size_t)p.ntiles[0] * TILE_ROWS * 4);
    p.L[0].el_dv4 = take((size_t)p.ntiles[0] * TILE_ROWS * 4);
    p.el_sums = take((size_t)5 * (p.ntiles[0] * TILE_ROWS / 256 + 1));
    p.el_coef = take((size_t)p.rows[0]);
  }
  if (h->warp) p.warp_wpk = take(h->wpk.total);
  p.seg_clock = take(2 * (p.segs.size() + 1));
  p.counters = take(64);
  p.timeline = take(2 * (2 * 4 * 64 + 1024 + 4 * 4096));

  // ---- pack descriptors (both levels, forward and transposed streams); a bf16 TRAINING plan reads only the bf16 images of the
  //      NeRF MLPs (bfpack), so their fp32 fragment images are not rebuilt every step ----
  for (int lv = 0; lv < (bft ? 0 : h->nlevels); ++lv) {
    const MlpParamOffsets& po = h->po[lv];
    const int64_t base = (int64_t)p.L[lv].wpk;
    const PackOffsets& pk = h->pk;
    auto add = [&](int64_t src, int dst, int ld, int row0, int kvalid, int K, int ncb, int tr, int nwaves = 4,
                   int nvalid = 1 << 30) {
      PackDesc q;
      q.src_off = src; q.dst_off = base + dst; q.src_ld = ld; q.src_row0 = row0; q.kvalid = kvalid; q.K = K; q.ncb = ncb;
      q.transposed = tr; q.nwaves = nwaves; q.nvalid = nvalid;
      p.pack.push_back(q);
    };
    add(po.trunk_k[0], pk.fwd_L[0], 256, 0, h->P, h->PK, 2, 0);
    for (int l = 1; l < TRUNK_DEPTH; ++l) add(po.trunk_k[l], pk.fwd_L[l], 256, 0, 256, 256, 2, 0);
    add(po.trunk_k[d.nerf_skip_layer], pk.fwd_L4b, 256, 256, h->P, h->PK, 2, 0);
    add(po.bn_k, pk.fwd_bn, 256, 0, 256, 256, 2, 0);
    add(po.rgbh_k, pk.fwd_rgbh, 128, 0, 256, 256, 1, 0);
    add(po.rgbh_k, pk.bwd_rgbhT, 128, 0, 128, 128, 2, 1);
    add(po.bn_k, pk.bwd_bnT, 256, 0, 256, 256, 2, 1);
    for (int l = 1; l < TRUNK_DEPTH; ++l) add(po.trunk_k[l], pk.bwd_LT[l], 256, 0, 256, 256, 2, 1);
    if (h->warp) {   // d posenc streams: B[k][n] = W[row0 + n][k], n < P, one 64-column group
      add(po.trunk_k[0], pk.bwd_L0T, 256, 0, 256, 256, 2, 1, 1, h->P);
      add(po.trunk_k[d.nerf_skip_layer], pk.bwd_L4bT, 256, 256, 256, 256, 2, 1, 1, h->P);
    }
  }
  if (h->warp && !bfw) {   // fp32 fragment images of the SE3 trunk (a bf16-trunk training plan reads only its bf16 streams)
    const WarpParamOffsets& w = h->wpo;
    const WarpPackOffsets& wk = h->wpk;
    const int64_t base = (int64_t)p.warp_wpk;
    auto addw = [&](int64_t src, int dst, int row0, int kvalid, int K, int tr) {
      PackDesc q;
      q.src_off = src; q.dst_off = base + dst; q.src_ld = WARP_W; q.src_row0 = row0; q.kvalid = kvalid; q.K = K; q.ncb = 1;
      q.transposed = tr; q.nwaves = 4; q.nvalid = 1 << 30;
      p.pack.push_back(q);
    };
    addw(w.trunk_k[0], wk.fwd_L[0], 0, h->Win, h->PKw, 0);
    for (int l = 1; l < WARP_DEPTH; ++l) addw(w.trunk_k[l], wk.fwd_L[l], 0, WARP_W, WARP_W, 0);
    addw(w.trunk_k[WARP_SKIP], wk.fwd_L4b, WARP_W, h->Win, h->PKw, 0);
    for (int l = 1; l < WARP_DEPTH; ++l) addw(w.trunk_k[l], wk.bwd_LT[l], 0, WARP_W, WARP_W, 1);
  }

  // ---- wgrad groups + slabs + reduce descriptors ----
  std::vector<ReduceDesc> reduce2, reduce3, reduce4;   // accumulating descriptors (later launches)
  if (train) {
    int first = 0;
    for (size_t i = 0; i < specs.size(); ++i) {
      const GroupSpec& s = specs[i];
      WgradGroup g;
      memset(&g, 0, sizeof(g));
      g.x_off = (int64_t)(*s.xoff + s.xadd);
      g.x_kind = s.xk; g.x_tile_stride = s.xstride; g.x_kvalid = s.kvalid; g.Kb = s.Kb;
      g.dy_off = s.yoff ? (int64_t)(*s.yoff + s.yadd) : 0;
      g.dy_kind = s.yk; g.dy_tile_stride = s.ystride; g.Nb = s.Nb;
      g.ntiles = p.ntiles[s.lv];
      g.nsplit = nsplit[i];
      g.tiles_per = 0;
      g.first_task = first;
      first += g.nsplit;
      ReduceDesc r;
      memset(&r, 0, sizeof(r));
      r.dst_off = s.dst; r.dst_ld = s.dst_ld; r.rows = s.rows; r.cols = s.cols; r.accumulate = s.accumulate;
      if (s.vec) {
        g.vec_off = (int64_t)(s.vecoff ? *s.vecoff : p.L[s.lv].d_raw4);
        g.vslab_off = (int64_t)take((size_t)g.nsplit * 2 * g.Kb * 32 * 4);
        g.slab_off = 0;
        r.src_off = g.vslab_off + (s.vec == 1 ? 3 : 0);
        r.src_ld = 4; r.part_stride = (int64_t)g.Kb * 32 * 4; r.nparts = 2 * g.nsplit;
        g.vec2_off = -1;
        if (s.vecoff2) {
          g.vec2_off = (int64_t)*s.vecoff2;
          g.vslab2_off = (int64_t)take((size_t)g.nsplit * 2 * g.Kb * 32 * 4);
          ReduceDesc r2 = r;
          r2.dst_off = s.dst2; r2.src_off = g.vslab2_off;
          (r2.accumulate == 0 ? p.reduce : r2.accumulate == 1 ? reduce2 : r2.accumulate == 2 ? reduce3 : reduce4).push_back(r2);
        }
      } else {
        g.vec_off = -1; g.vslab_off = 0; g.vec2_off = -1;
        g.slab_off = (int64_t)take((size_t)g.nsplit * g.Kb * 32 * g.Nb * 32);
        r.src_off = g.slab_off; r.src_ld = g.Nb * 32; r.part_stride = (int64_t)g.Kb * 32 * g.Nb * 32; r.nparts = g.nsplit;
      }
      p.groups.push_back(g);
      (r.accumulate == 0 ? p.reduce : r.accumulate == 1 ? reduce2 : r.accumulate == 2 ? reduce3 : reduce4).push_back(r);
    }
    auto warp_bias_descs = [&](int lv, int grid, int accu) {
      const WarpParamOffsets& w = h->wpo;
      const LevelWs& L = p.L[lv];
      auto wsmall = [&](int64_t dst, int cols, int sp_off) {
        ReduceDesc r;
        memset(&r, 0, sizeof(r));
        r.dst_off = dst; r.dst_ld = cols; r.rows = 1; r.cols = cols; r.accumulate = accu;
        r.src_off = (int64_t)L.w_small_part + sp_off; r.src_ld = cols; r.part_stride = WARP_SMALL_PART; r.nparts = grid;
        (r.accumulate == 0 ? p.reduce : r.accumulate == 1 ? reduce2 : r.accumulate == 2 ? reduce3 : reduce4).push_back(r);
      };
      for (int l = 0; l < WARP_DEPTH; ++l) wsmall(w.trunk_b[l], WARP_W, l * WARP_W);
      wsmall(w.w_b, 3, 768);
      wsmall(w.v_b, 3, 771);
    };
    // bf16 groups: slab [Kb*32][Nb*32] per segment (+ a bias slab [Nb*32]); the leaf takes a column window of it
    for (size_t i = 0; i < bspecs.size(); ++i) {
      const BSpec& sp = bspecs[i];
      WgradGroup g;
      memset(&g, 0, sizeof(g));
      g.x_off = (int64_t)(*sp.xoff + sp.xadd); g.x_tile_stride = (sp.Kb - sp.Kb2) * BF_BLOCK_DW; g.Kb = sp.Kb; g.x_kvalid = sp.rows;
      g.dy_off = (int64_t)(*sp.yoff + sp.yadd); g.dy_tile_stride = (sp.Nb - sp.Nb2) * BF_BLOCK_DW; g.Nb = sp.Nb;
      g.Kb1 = sp.Kb - sp.Kb2; g.Nb1 = sp.Nb - sp.Nb2;
      g.x2_off = sp.x2off ? (int64_t)*sp.x2off : g.x_off; g.x2_tile_stride = sp.x2off ? sp.x2_blocks * BF_BLOCK_DW : 0;
      g.dy2_off = sp.y2off ? (int64_t)*sp.y2off : g.dy_off; g.dy2_tile_stride = sp.y2off ? sp.y2_blocks * BF_BLOCK_DW : 0;
      g.ntiles = sp.ngroups ? sp.ngroups : p.L[sp.lv].b_ngroups; g.nsplit = bnsplit[i]; g.vec_off = -1; g.vec2_off = -1;
      g.slab_off = (int64_t)take((size_t)g.nsplit * sp.Kb * 32 * sp.Nb * 32);
      g.vslab_off = sp.bias_dst >= 0 ? (int64_t)take((size_t)g.nsplit * sp.Nb * 32) : -1;
      p.bgroups.push_back(g);
      ReduceDesc r;
      memset(&r, 0, sizeof(r));
      r.dst_off = sp.dst; r.dst_ld = sp.dst_ld; r.rows = sp.rows; r.cols = sp.cols;
      r.src_off = g.slab_off + sp.col0; r.src_ld = sp.Nb * 32; r.part_stride = (int64_t)sp.Kb * 32 * sp.Nb * 32; r.nparts = g.nsplit;
      r.accumulate = sp.accu;
      auto rpush = [&](const ReduceDesc& q) {
        (q.accumulate == 0 ? p.reduce : q.accumulate == 1 ? reduce2 : q.accumulate == 2 ? reduce3 : reduce4).push_back(q);
      };
      rpush(r);
      if (sp.dst2 >= 0) {   // a second leaf out of the same slab (column window col20)
        ReduceDesc r2 = r;
        r2.dst_off = sp.dst2; r2.src_off = g.slab_off + sp.col20;
        if (sp.dst2_cols > 0) { r2.dst_ld = sp.dst2_ld; r2.cols = sp.dst2_cols; }
        rpush(r2);
      }
      auto bias = [&](int64_t dst, int cols, int col0) {
        ReduceDesc b;
        memset(&b, 0, sizeof(b));
        b.dst_off = dst; b.dst_ld = cols; b.rows = 1; b.cols = cols; b.accumulate = sp.accu;
        b.src_off = g.vslab_off + col0; b.src_ld = sp.Nb * 32; b.part_stride = sp.Nb * 32; b.nparts = g.nsplit;
        rpush(b);
      };
      if (sp.bias_dst >= 0) bias(sp.bias_dst, sp.bias_cols, 0);
      if (sp.bias2_dst >= 0) bias(sp.bias2_dst, sp.bias2_cols, sp.bias2_col0);
    }
    // bias gradients and per-ray condition rows
    for (int lv = 0; lv < h->nlevels; ++lv) {
      const MlpParamOffsets& po = h->po[lv];
      const LevelWs& L = p.L[lv];
      int nt_mlp = 0;
      for (int q = 0; q < h->nlevels; ++q) nt_mlp += p.ntiles[q];
      // ONE dgrad launch over the tiles of all levels: two workgroups per CU on 64-row tiles, four on 32-row half tiles
      const int grid = p.bwd32 ? (2 * nt_mlp < 4 * G ? 2 * nt_mlp : 4 * G) : (nt_mlp < 2 * G ? nt_mlp : 2 * G);
      auto small = [&](int64_t dst, int cols, int sp_off) {
        if (bft) return;   // the bf16 wgrad kernel sums the bias columns itself
        ReduceDesc r;
        memset(&r, 0, sizeof(r));
        r.dst_off = dst; r.dst_ld = cols; r.rows = 1; r.cols = cols;
        r.src_off = (int64_t)L.small_part + sp_off; r.src_ld = cols; r.part_stride = SMALL_PART; r.nparts = grid;
        p.reduce.push_back(r);
      };
      for (int l = 0; l < TRUNK_DEPTH; ++l) small(po.trunk_b[l], 256, l * 256);
      small(po.bn_b, 256, 2048);
      small(po.rgbh_b, 128, 2304);
      small(po.logit_b, 3, 2432);
      small(po.alpha_b, 1, 2435);
      if (h->R > 0) {
        ReduceDesc r;
        memset(&r, 0, sizeof(r));
        r.dst_off = po.rgbh_k + 256 * 128; r.dst_ld = 128; r.rows = h->R; r.cols = 128;
        r.src_off = (int64_t)L.cond_grad; r.src_ld = 128; r.part_stride = 0; r.nparts = 1;
        p.reduce.push_back(r);
      }
      if (h->warp && !bfw && lv == 0) {   // ONE SE3 dgrad launch (coarse + fine + background tiles), one set of bias partials
        const int nt_w = nt_mlp + (bgN > 0 ? p.ntiles[BG] : 0);
        warp_bias_descs(0, nt_w < warp_grid_mul() * G ? nt_w : warp_grid_mul() * G, 0);
      }
    }
  }
  p.nreduce_pass[0] = (int)p.reduce.size();
  p.nreduce_pass[1] = (int)reduce2.size();
  p.nreduce_pass[2] = (int)reduce3.size();
  p.nreduce_pass[3] = (int)reduce4.size();
  p.reduce.insert(p.reduce.end(), reduce2.begin(), reduce2.end());
  p.reduce.insert(p.reduce.end(), reduce3.begin(), reduce3.end());
  p.reduce.insert(p.reduce.end(), reduce4.begin(), reduce4.end());
  // ---- descriptor tables (bytes), sized from what was actually built (round 2 reserved 64 pack / 192 reduce
  //      descriptors without a check) ----
  p.pack_off_b = 0;
  p.groups_off_b = align_up((p.pack.size() + 1) * sizeof(PackDesc), 256);
  p.reduce_off_b = p.groups_off_b + align_up(specs.size() * sizeof(WgradGroup) + 256, 256);
  p.segs_off_b = p.reduce_off_b + align_up((p.reduce.size() + 1) * sizeof(ReduceDesc), 256);
  p.segbegin_off_b = p.segs_off_b + align_up(p.segs.size() * sizeof(WgradSegment) + 256, 256);
  p.emb_off_b = p.segbegin_off_b + align_up((p.seg_begin.size() + 1) * sizeof(int), 256);
  p.bgroups_off_b = p.emb_off_b + align_up((h->emb.size() + 1) * sizeof(EmbedDesc), 256);
  p.bsegs_off_b = p.bgroups_off_b + align_up(bspecs.size() * sizeof(WgradGroup) + 256, 256);
  p.bsegbegin_off_b = p.bsegs_off_b + align_up(p.bsegs.size() * sizeof(WgradSegment) + 256, 256);
  const size_t table_bytes = p.bsegbegin_off_b + align_up((p.bseg_begin.size() + 1) * sizeof(int), 256);
  p.tables = take(table_bytes / 4);
  p.total_floats = o;
}

int upload_tables(nrf_handle h, float* ws, hipStream_t stream) {
  WsPlan& p = h->plan;
  if (h->uploaded_ws == (void*)ws && h->uploaded_B == p.B && h->uploaded_flags == p.flags && h->uploaded_bgN == p.bgN && h->uploaded_elastic == p.elastic) return NRF_OK;
  char* base = reinterpret_cast<char*>(ws + p.tables);
  hipError_t e;
  if (!p.pack.empty()) {
    e = hipMemcpyAsync(base + p.pack_off_b, p.pack.data(), p.pack.size() * sizeof(PackDesc), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return fail_hip(e, "upload pack table");
  }
  if (!p.groups.empty()) {
    e = hipMemcpyAsync(base + p.groups_off_b, p.groups.data(), p.groups.size() * sizeof(WgradGroup), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return fail_hip(e, "upload wgrad table");
  }
  if (!p.segs.empty()) {
    e = hipMemcpyAsync(base + p.segs_off_b, p.segs.data(), p.segs.size() * sizeof(WgradSegment), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return fail_hip(e, "upload wgrad segments");
    e = hipMemcpyAsync(base + p.segbegin_off_b, p.seg_begin.data(), p.seg_begin.size() * sizeof(int), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return fail_hip(e, "upload wgrad segment index");
  }
  if (!p.bfpack.empty()) {
    e = hipMemcpyAsync(ws + p.bf_desc, p.bfpack.data(), p.bfpack.size() * sizeof(RcPackDesc), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return fail_hip(e, "upload bf16 pack table");
  }
  if (!p.bgroups.empty()) {
    e = hipMemcpyAsync(base + p.bgroups_off_b, p.bgroups.data(), p.bgroups.size() * sizeof(WgradGroup), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return fail_hip(e, "upload bf16 wgrad table");
    e = hipMemcpyAsync(base + p.bsegs_off_b, p.bsegs.data(), p.bsegs.size() * sizeof(WgradSegment), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return fail_hip(e, "upload bf16 wgrad segments");
    e = hipMemcpyAsync(base + p.bsegbegin_off_b, p.bseg_begin.data(), p.bseg_begin.size() * sizeof(int), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return fail_hip(e, "upload bf16 wgrad segment index");
  }
  if (h->embed) {
    e = hipMemcpyAsync(base + p.emb_off_b, h->emb.data(), h->emb.size() * sizeof(EmbedDesc), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return fail_hip(e, "upload embed table");
  }
  if (!p.reduce.empty()) {
    e = hipMemcpyAsync(base + p.reduce_off_b, p.reduce.data(), p.reduce.size() * sizeof(ReduceDesc), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return fail_hip(e, "upload reduce table");
  }
  h->uploaded_ws = ws;
  h->uploaded_B = p.B;
  h->uploaded_flags = p.flags;
  h->uploaded_bgN = p.bgN;
  h->uploaded_elastic = p.elastic;
  return NRF_OK;
}

void query_device(nrf_handle h) {
  if (h->cu_queried) return;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
    h->num_cus = cus;
  h->cu_queried = true;
}

int check_launch(const char* where) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, where);
  return NRF_OK;
}

#define CK(call)                      \
  do {                                \
    int rc_ = (call);                 \
    if (rc_ != NRF_OK) return rc_;    \
  } while (0)

int validate_rays(nrf_handle h, const nrf_rays* rays) {
  if (!rays || !rays->origins || !rays->directions) return fail(NRF_E_NULL, "rays / origins / directions is null");
  if (rays->num_rays <= 0) return fail(NRF_E_SHAPE, "num_rays must be positive");
  if (h->d.use_camera_metadata && !rays->camera_ids && !rays->camera_codes)
    return fail(NRF_E_NULL, "camera_ids (or camera_codes) required (use_camera_metadata)");
  if (h->app_in_cond && !rays->appearance_ids && !rays->appearance_codes) return fail(NRF_E_NULL, "appearance_ids (or appearance_codes) required");
  if (h->warp && !h->time_enc && !rays->warp_ids && !rays->warp_codes) return fail(NRF_E_NULL, "warp_ids (or warp_codes) required (use_warp)");
  if (h->warp && h->time_enc && !rays->time && !rays->warp_codes) return fail(NRF_E_NULL, "time (or warp_codes) required (warp_metadata_encoder_type 'time')");
  return NRF_OK;
}

BfStash bf_stash(const WsPlan& p, int lv, float* ws) {
  const LevelWs& L = p.L[lv];
  BfStash b;
  auto u = [&](size_t off) { return reinterpret_cast<uint32_t*>(ws + off); };
  b.pe = u(L.b_pe); b.h = u(L.b_h); b.bn = u(L.b_bn); b.rgbh = u(L.b_rgbh); b.bits = u(L.b_bits);
  b.dy = u(L.b_dy); b.dbn = u(L.b_dbn); b.drgbh = u(L.b_drgbh); b.dsmall = u(L.b_dsmall);
  b.ngroups = L.b_ngroups;
  return b;
}

BfWarpStash bfw_stash(const WsPlan& p, int lv, float* ws) {
  const LevelWs& L = p.L[lv];
  BfWarpStash b;
  auto u = [&](size_t off) { return reinterpret_cast<uint32_t*>(ws + off); };
  b.win = u(L.bw_in); b.h = u(L.bw_h); b.bits = u(L.bw_bits); b.dy = u(L.bw_dy); b.dhead = u(L.bw_dhead);
  b.ngroups = L.bw_ngroups;
  return b;
}

ChainFwdArgs fwd_args(nrf_handle h, int lv, const float* params, const nrf_rays* rays, float* ws, bool train, const nrf_rand* rnd,
                      const nrf_dynamic_scalars* dyn = nullptr) {
  const WsPlan& p = h->plan;
  const LevelWs& L = p.L[lv];
  ChainFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.params = params; a.po = h->po[lv]; a.wpk = ws + L.wpk; a.pk = h->pk;
  a.condterm = ws + L.condterm; a.zvals = ws + L.z; a.origins = rays->origins; a.directions = rays->directions;
  a.points = nullptr; a.out4 = reinterpret_cast<float4*>(ws + L.out4);
  a.S = p.S[lv]; a.B = p.B; a.rows = p.rows[lv]; a.ntiles = p.ntiles[lv];
  a.F = h->d.num_nerf_point_freqs; a.P = h->P; a.PK = h->PK; a.sigma_act = h->d.sigma_activation; a.skip = h->d.nerf_skip_layer;
  a.tile_counter = tile_counter_or_null(ws + p.counters, CT_MLP_FWD + lv);
  a.timeline = knobs().timeline ? reinterpret_cast<unsigned long long*>(ws + p.timeline) + lv * (256 + 512 + 4 * 2048) : nullptr;
  a.alpha_ct = h->A > 0 ? ws + L.alpha_ct : nullptr;
  if (h->d.noise_std > 0.f && h->d.use_stratified_sampling) {   // model_utils.noise_regularize (model_utils.py:266-282)
    a.noise_std = h->d.noise_std;
    a.noise = rnd ? (lv == 0 ? rnd->noise_coarse : rnd->noise_fine) : nullptr;
    a.noise_seed = rnd ? rnd->seed : 0; a.noise_offset = rnd ? rnd->offset : 0; a.noise_stream = 2u + (unsigned)lv;
    a.dyn = dyn;
  }
  if (train && (p.flags & NRF_FLAG_BF16)) {
    a.bst = bf_stash(p, lv, ws);
  } else if (train) {
    a.st_pe = ws + L.st_pe; a.st_h = ws + L.st_h; a.st_bn = ws + L.st_bn; a.st_rgbh = ws + L.st_rgbh;
    a.bits_trunk = reinterpret_cast<uint32_t*>(ws + L.bits_trunk);
    a.bits_rgbh = reinterpret_cast<uint32_t*>(ws + L.bits_rgbh);
  }
  return a;
}

int copy_out(float* dst, const float* src, size_t n, hipStream_t stream) {
  if (!dst) return NRF_OK;
  hipError_t e = hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, stream);
  return e == hipSuccess ? NRF_OK : fail_hip(e, "copy output");
}

// algorithmic flops per MLP row (2 flop / MAC, dense layers only, unpadded)
double fwd_flops_row(nrf_handle h) {
  const double P = h->P, R = h->R;
  return 2.0 * (P * 256 + 6 * 65536.0 + (256 + P) * 256 + 65536.0 + 256 + (256 + R) * 128 + 128 * 3);
}
double dgrad_flops_row(nrf_handle h, bool warp_on) {
  const double base = 2.0 * (128 * 3 + 256 * 128 + 65536.0 + 256 + 7 * 65536.0);
  return warp_on ? base + 2.0 * (2.0 * 256 * h->P) : base;   // + d posenc through layer 0 and the skip rows
}
// SE3 field per row: trunk + heads
double warp_fwd_flops_row(nrf_handle h) {
  const double Wi = h->Win;
  return 2.0 * (Wi * 128 + 3 * 16384.0 + (128 + Wi) * 128 + 16384.0 + 128 * 6);
}
double warp_dgrad_flops_row(nrf_handle h) { return 2.0 * (128 * 6 + 5 * 16384.0 + 2.0 * h->G * 128); }
double warp_fwd_flops_row_or0(nrf_handle h) { return h->warp ? warp_fwd_flops_row(h) : 0.0; }
double wgrad_flops_row(nrf_handle h) {
  const double P = h->P, R = h->R;
  return 2.0 * (2 * P * 256 + 7 * 65536.0 + 65536.0 + (256 + R) * 128 + 256 + 128 * 3);
}

WarpFwdArgs warp_fwd_args(nrf_handle h, int lv, const float* params, const nrf_rays* rays, const nrf_step_scalars* sc, float* ws, bool train) {
  const float alpha = sc->warp_alpha;
  const WsPlan& p = h->plan;
  const LevelWs& L = p.L[lv];
  WarpFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.params = params; a.po = h->wpo; a.wpk = ws + p.warp_wpk; a.pk = h->wpk;
  a.zvals = ws + L.z; a.origins = rays->origins; a.directions = rays->directions;
  // metadata_encoded (warping.py:378-381): the caller's per-ray codes stand in for the table, row = ray
  // the same for the TimeEncoder's per-ray output
  const bool per_ray = rays->warp_codes || h->time_enc;
  a.warp_ids = per_ray ? nullptr : rays->warp_ids;
  a.embed_table = rays->warp_codes ? rays->warp_codes : h->time_enc ? ws + p.t_codes : params + h->wpo.embed;
  a.points_out = ws + L.wpoints; a.points_raw = ws + L.points_raw;
  a.S = p.S[lv]; a.B = p.B; a.rows = p.rows[lv]; a.ntiles = p.ntiles[lv];
  a.F = h->Fw; a.G = h->G; a.Win = h->Win; a.PKw = h->PKw; a.alpha = alpha; a.dyn = sc->dynamic;
  a.tile_counter = tile_counter_or_null(ws + p.counters, CT_WARP_FWD + lv);
  if (train) {   // train: here "keep the stash" (training plan, or an inference plan that returns the Jacobian)
    a.st_win = ws + L.w_st_win; a.st_h = ws + L.w_st_h; a.st_wv = reinterpret_cast<float4*>(ws + L.w_st_wv);
    a.bits = reinterpret_cast<uint32_t*>(ws + L.w_bits);
  }
  return a;
}

// forward-mode pass of the warp Jacobian of level lv (warping.py:385-387): 3 tangent tiles per primal tile
void launch_tangent_fwd(nrf_handle h, int lv, const float* params, const nrf_rays* rays, const nrf_step_scalars* sc, float* ws, int gmul, hipStream_t stream) {
  const WsPlan& p = h->plan;
  const LevelWs& L = p.L[lv];
  const LevelWs& T = p.L[TG];
  WarpFwdArgs ta = warp_fwd_args(h, lv, params, rays, sc, ws, true);
  ta.nt_prim = p.ntiles[lv]; ta.prim_win = ws + L.w_st_win; ta.prim_bits = reinterpret_cast<const uint32_t*>(ws + L.w_bits);
  ta.ntiles = 3 * p.ntiles[lv]; ta.rows = ta.ntiles * TILE_ROWS;
  ta.st_win = ws + T.w_st_win; ta.st_h = ws + T.w_st_h; ta.st_wv = reinterpret_cast<float4*>(ws + T.w_st_wv);
  ta.bits = nullptr; ta.points_out = ws + T.wpoints; ta.points_raw = nullptr;
  ta.tile_counter = tile_counter_or_null(ws + p.counters, CT_TAN_FWD);
  (void)gmul;
  const int tgrid = ta.ntiles < warp_grid_mul() * h->num_cus ? ta.ntiles : warp_grid_mul() * h->num_cus;
  h->prof.begin("warp_tangent_fwd", 3.0 * warp_fwd_flops_row(h) * p.rows[lv], stream);
  if (p.bfw) {   // bf16 trunk: tangent groups = 3 x the primal groups, masks = the primal pass's bits
    ta.rows = p.rows[lv]; ta.rows_pad = p.ntiles[lv] * TILE_ROWS;
    ta.bwpk = ws + p.bfw_wpk; ta.bst = bfw_stash(p, TG, ws);
    ta.bprim_bits = reinterpret_cast<const uint32_t*>(ws + L.bw_bits); ta.bng_prim = L.bw_ngroups;
    launch_warp_fwd_bf16(ta, nullptr, true, h->num_cus, stream);
  } else {
    launch_warp_fwd(ta, nullptr, true, tgrid, stream);
  }
  h->prof.end(stream);
}

WarpFwdArgs bg_fwd_args(nrf_handle h, const float* params, const nrf_background* bg, const nrf_step_scalars* sc, float* ws);
void draw_background(nrf_handle h, const nrf_background* bg, const nrf_rand* rnd, const nrf_step_scalars* sc, float* ws, hipStream_t stream);

int forward_impl(nrf_handle h, const float* params_x, const nrf_rays* rays, const nrf_step_scalars* scalars, const nrf_rand* rnd,
                 const nrf_outputs* out, uint32_t flags, float* ws, size_t ws_bytes, hipStream_t stream, int bgN = 0,
                 int elastic = 0, const nrf_background* bg = nullptr) {
  CK(validate_rays(h, rays));
  if (!params_x || !ws) return fail(NRF_E_NULL, "params / workspace is null");
  query_device(h);
  const int B = rays->num_rays;
  build_plan(h, B, flags, bgN, elastic);
  WsPlan& p = h->plan;
  if (ws_bytes < p.total_floats * sizeof(float)) return fail(NRF_E_WORKSPACE, "workspace too small (see nrf_workspace_bytes)");
  if (!p.bf_stream_ok) return fail(NRF_E_STATE, "bf16 weight stream tables do not match the kernels' chunk sequence");
  const nrf_model_desc& d = h->d;
  const bool train = flags & NRF_FLAG_TRAIN;
  const bool warp_on = h->warp && !(flags & NRF_FLAG_NO_WARP);   // models.py:296 use_warp argument
  if (warp_on && !scalars) return fail(NRF_E_NULL, "nrf_step_scalars (warp_alpha) required with the warp field");
  if (h->warp && !warp_on && train) return fail(NRF_E_UNSUPPORTED, "NRF_FLAG_NO_WARP cannot be combined with NRF_FLAG_TRAIN");
  if (d.use_stratified_sampling && !rnd) return fail(NRF_E_NULL, "nrf_rand required with stratified sampling");
  const bool encoded = rays->warp_codes || rays->appearance_codes || rays->camera_codes;
  if (encoded && train) return fail(NRF_E_UNSUPPORTED, "pre-encoded metadata (metadata_encoded) is an inference input: no gradient flows to the codes");
  const bool jac = (flags & NRF_FLAG_WARP_JACOBIAN) != 0;
  if (jac && (!warp_on || train)) return fail(NRF_E_UNSUPPORTED, "NRF_FLAG_WARP_JACOBIAN needs the warp field and an inference call (training consumes the Jacobian through nrf_elastic)");
  if (!jac && out && (out->coarse.warp_jacobian || out->fine.warp_jacobian)) return fail(NRF_E_STATE, "warp_jacobian outputs need NRF_FLAG_WARP_JACOBIAN");
  CK(upload_tables(h, ws, stream));
  const char* tables = reinterpret_cast<const char*>(ws + p.tables);
  if (tile_counter_or_null(ws + p.counters, 0) &&   // NRF_DYNAMIC_TILES experiment only
      hipMemsetAsync(ws + p.counters, 0, 64 * sizeof(int), stream) != hipSuccess) return fail(NRF_E_HIP, "zero tile counters");
  const float* params = params_x;
  if (h->embed) {   // narrower model: run on its zero-padded image (nrf_internal.h EmbedDesc)
    if (hipMemsetAsync(ws + p.iparams, 0, (size_t)h->nparams * sizeof(float), stream) != hipSuccess) return fail(NRF_E_HIP, "zero padded params");
    launch_embed(reinterpret_cast<const EmbedDesc*>(tables + p.emb_off_b), (int)h->emb.size(), params_x, ws + p.iparams, true, stream);
    params = ws + p.iparams;
  }

  Prof& pf = h->prof;
  pf.begin("pack_prep_sample", 0, stream);
  if (!p.pack.empty()) launch_pack(reinterpret_cast<const PackDesc*>(tables + p.pack_off_b), (int)p.pack.size(), params, ws, stream);
  const bool bf16 = flags & NRF_FLAG_BF16;
  // the SE3 trunk follows the MLPs into bf16 unless the caller opts out (NRF_FLAG_WARP_F32) or asks for the Jacobian output
  // (inference tangent pass: fp32 kernels); a training plan has decided already (its stash layout depends on it)
  const bool bfw_on = warp_on && bf16 && (train ? p.bfw : !(flags & NRF_FLAG_WARP_F32) && !jac);
  if (bf16) launch_bf16_pack(reinterpret_cast<const RcPackDesc*>(ws + p.bf_desc), (int)p.bfpack.size(), params, ws, stream);
  const float* viewdirs = rays->viewdirs ? rays->viewdirs : rays->directions;   // models.py:326-329
  {
    RayPrepArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.params = params; ra.viewdirs = viewdirs;
    ra.app_ids = rays->appearance_codes ? nullptr : rays->appearance_ids; ra.app_codes = rays->appearance_codes;
    ra.cam_ids = rays->camera_codes ? nullptr : rays->camera_ids; ra.cam_codes = rays->camera_codes;
    ra.B = B; ra.Fv = d.num_nerf_viewdir_freqs; ra.use_viewdirs = d.use_viewdirs;
    ra.app_feat = h->app_in_cond ? d.num_appearance_features : 0; ra.app_off = h->app_off;
    ra.cam_feat = d.use_camera_metadata ? d.num_camera_features : 0; ra.cam_off = h->cam_off; ra.R = h->R;
    for (int lv = 0; lv < h->nlevels; ++lv) {
      ra.rgbh_k[lv] = h->po[lv].rgbh_k; ra.rgbh_b[lv] = h->po[lv].rgbh_b; ra.alpha_k[lv] = h->po[lv].alpha_k;
      ra.condterm[lv] = ws + p.L[lv].condterm;
      ra.alpha_ct[lv] = h->A > 0 ? ws + p.L[lv].alpha_ct : nullptr;
    }
    ra.cond = ws + p.cond;
    launch_ray_prep(ra, stream);
  }
  const nrf_dynamic_scalars* dyn = scalars ? scalars->dynamic : nullptr;
  launch_sample_coarse(rnd ? rnd->t_rand : nullptr, B, p.S[0], d.near_plane, d.far_plane, d.use_stratified_sampling,
                       d.use_linear_disparity, rnd ? rnd->seed : 0, rnd ? rnd->offset : 0, dyn, ws + p.L[0].z, stream);
  if (train && bg && p.bgN > 0 && warp_on) draw_background(h, bg, rnd, scalars, ws, stream);
  pf.end(stream);
  if (warp_on && h->time_enc && !rays->warp_codes) {   // modules.TimeEncoder once per ray (warping.py:311-313, models.py:252-254)
    TimeEncArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.params = params; ta.po = h->tpo; ta.time = rays->time; ta.B = B; ta.F = h->Ft; ta.Tin = h->Tin; ta.G = h->G;
    ta.alpha = scalars->time_alpha; ta.dyn = dyn; ta.codes = ws + p.t_codes;
    if (train) { ta.st_in = ws + p.t_in; ta.st_h = ws + p.t_h; }
    launch_time_encoder_fwd(ta, stream);
  }
  for (int lv = 0; lv < h->nlevels; ++lv) {
    const LevelWs& L = p.L[lv];
    if (lv == 1) {
      pf.begin("sample_pdf", 0, stream);
      launch_sample_fine(ws + p.L[0].z, ws + p.L[0].weights, B, d.num_coarse_samples, d.num_fine_samples,
                         d.use_stratified_sampling, rnd ? rnd->u : nullptr, rnd ? rnd->seed : 0, rnd ? rnd->offset : 0, dyn,
                         ws + L.z, stream);
      pf.end(stream);
    }
    ChainFwdArgs a = fwd_args(h, lv, params, rays, ws, train, rnd, dyn);
    const int gmul = knobs().grid_mul;
    const bool c32 = !bf16 && chain32_for(h, p.ntiles[lv]);   // 32-row half tiles, four workgroups per CU
    const int grid = c32 ? (2 * p.ntiles[lv] < 4 * h->num_cus ? 2 * p.ntiles[lv] : 4 * h->num_cus)
                         : (p.ntiles[lv] < gmul * h->num_cus ? p.ntiles[lv] : gmul * h->num_cus);   // two workgroups per CU
    if (warp_on) {
      // the background-point batch of the fused train step rides in the coarse launch (its 256 tiles under-fill the chip)
      const bool with_bg = lv == 0 && train && bg && p.bgN > 0;
      WarpFwdArgs bga;
      if (with_bg) bga = bg_fwd_args(h, params, bg, scalars, ws);
      const int wnt = p.ntiles[lv] + (with_bg ? p.ntiles[BG] : 0);
      const int wgrid = wnt < warp_grid_mul() * h->num_cus ? wnt : warp_grid_mul() * h->num_cus;
      pf.begin(lv == 0 ? "warp_fwd_coarse" : "warp_fwd_fine", warp_fwd_flops_row(h) * (p.rows[lv] + (with_bg ? p.bgN : 0)), stream);
      if (bfw_on) {   // SE3 trunk on bf16 operands (warp_bf16.hip); one workgroup per CU, 256 rows per iteration
        WarpFwdArgs wa = warp_fwd_args(h, lv, params, rays, scalars, ws, train);
        wa.bwpk = ws + p.bfw_wpk; wa.rows_pad = p.ntiles[lv] * TILE_ROWS;
        if (train) wa.bst = bfw_stash(p, lv, ws);
        if (with_bg) { bga.bwpk = wa.bwpk; bga.rows_pad = p.ntiles[BG] * TILE_ROWS; bga.bst = bfw_stash(p, BG, ws); }
        launch_warp_fwd_bf16(wa, with_bg ? &bga : nullptr, train, h->num_cus, stream);
      } else {
        launch_warp_fwd(warp_fwd_args(h, lv, params, rays, scalars, ws, train || jac), with_bg ? &bga : nullptr,
                        train || jac, wgrid, stream);
      }
      pf.end(stream);
      a.points = ws + L.wpoints;
      // forward-mode Jacobian of the warp: on the coarse samples for the elastic regulariser (models.py:345), per level
      // as an output (return_warp_jacobian, models.py:345-346, 367-368)
      float* jout = !out ? nullptr : lv == 0 ? out->coarse.warp_jacobian : out->fine.warp_jacobian;
      if ((lv == 0 && train && p.elastic) || (jac && jout)) launch_tangent_fwd(h, lv, params, rays, scalars, ws, gmul, stream);
      if (jac && jout) {
        JacobianArgs ja;
        memset(&ja, 0, sizeof(ja));   // x_rows = nullptr: the points come from the fp32 input stash
        ja.prim_win = ws + L.w_st_win; ja.prim_wv = reinterpret_cast<const float4*>(ws + L.w_st_wv);
        ja.tan_wv = reinterpret_cast<const float4*>(ws + p.L[TG].w_st_wv); ja.out = jout;
        ja.rows = p.rows[lv]; ja.rows_pad = p.ntiles[lv] * TILE_ROWS; ja.PKS = (h->PKw + 31) / 32 * 32;
        launch_jacobian(ja, stream);
      }
    }
    a.k_old = k_old_for(p.ntiles[lv], grid, h->num_cus, 0.0);
    pf.begin(lv == 0 ? "mlp_fwd_coarse" : "mlp_fwd_fine", fwd_flops_row(h) * p.rows[lv], stream);
    if (bf16) {   // one workgroup per CU (90 KiB of weight staging), 256 samples per workgroup iteration
      a.wpk = ws + L.bf_wpk;
      const int nit = (p.rows[lv] + 255) / 256;
      launch_chain_fwd_bf16(a, nit < h->num_cus ? nit : h->num_cus, stream);
    } else {
      if (c32) launch_chain_fwd32(a, train, grid, stream);
      else launch_chain_fwd(a, train, grid, stream);
    }
    pf.end(stream);
    pf.begin("composite_fwd", 0, stream);
    launch_composite_fwd(reinterpret_cast<const float4*>(ws + L.out4), ws + L.z, rays->directions, B, p.S[lv],
                         d.use_white_background, d.use_sample_at_infinity, ws + L.rgb, ws + L.depth, ws + L.med,
                         ws + L.acc, ws + L.weights, stream);
    pf.end(stream);
    if (out) {
      const nrf_level_out& lo = lv == 0 ? out->coarse : out->fine;
      CK(copy_out(lo.rgb, ws + L.rgb, (size_t)B * 3, stream));
      CK(copy_out(lo.depth, ws + L.depth, B, stream));
      CK(copy_out(lo.med_depth, ws + L.med, B, stream));
      CK(copy_out(lo.acc, ws + L.acc, B, stream));
      CK(copy_out(lo.weights, ws + L.weights, (size_t)p.rows[lv], stream));
      CK(copy_out(lo.z_vals, ws + L.z, (size_t)p.rows[lv], stream));
      if (lo.warped_points && !warp_on) return fail(NRF_E_UNSUPPORTED, "the warped_points output needs the warp field (models.py:266-267)");
      if (lo.points && !warp_on)   // models.py:247-248: `points` is returned whether or not the model warps
        launch_sample_points(rays->origins, rays->directions, ws + L.z, B, p.S[lv], lo.points, stream);
      else if (lo.points || lo.warped_points) {
        CK(copy_out(lo.points, ws + L.points_raw, (size_t)p.rows[lv] * 3, stream));
        CK(copy_out(lo.warped_points, ws + L.wpoints, (size_t)p.rows[lv] * 3, stream));
      }
    }
  }
  CK(check_launch("nrf_forward"));
  h->stashed_ws = train ? (void*)ws : nullptr;
  h->stashed_plan = train ? p.serial : 0;
  h->stashed_B = train ? B : -1;
  h->stashed_warp = warp_on;
  return NRF_OK;
}

// SE3 field on the (already noised) background points, one warp id per point (training.compute_background_loss,
// training.py:117-135): forward arguments of the BG level
// the points / ids the background level runs on: the caller's (already noised, ids given) or the library's own draw
const float* bg_points_of(const WsPlan& p, const nrf_background* bg, const float* ws) { return bg->warp_ids ? bg->points : ws + p.bg_points; }
const int32_t* bg_ids_of(const WsPlan& p, const nrf_background* bg, const float* ws) {
  return bg->warp_ids ? bg->warp_ids : reinterpret_cast<const int32_t*>(ws + p.bg_ids);
}

WarpFwdArgs bg_fwd_args(nrf_handle h, const float* params, const nrf_background* bg, const nrf_step_scalars* sc, float* ws) {
  const WsPlan& p = h->plan;
  const LevelWs& L = p.L[BG];
  WarpFwdArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.params = params; fa.po = h->wpo; fa.wpk = ws + p.warp_wpk; fa.pk = h->wpk;
  fa.points_in = bg_points_of(p, bg, ws); fa.point_ids = bg_ids_of(p, bg, ws); fa.points_out = ws + L.wpoints;
  fa.embed_table = params + h->wpo.embed;
  fa.S = 1; fa.B = p.bgN; fa.rows = p.bgN; fa.ntiles = p.ntiles[BG];
  fa.F = h->Fw; fa.G = h->G; fa.Win = h->Win; fa.PKw = h->PKw; fa.alpha = sc->warp_alpha; fa.dyn = sc->dynamic;
  fa.st_win = ws + L.w_st_win; fa.st_h = ws + L.w_st_h; fa.st_wv = reinterpret_cast<float4*>(ws + L.w_st_wv);
  fa.bits = reinterpret_cast<uint32_t*>(ws + L.w_bits);
  return fa;
}

// nrf_background.warp_ids == NULL: training.py:121-126 on the device (ids from id_choices, noise added), into the workspace
void draw_background(nrf_handle h, const nrf_background* bg, const nrf_rand* rnd, const nrf_step_scalars* sc, float* ws, hipStream_t stream) {
  const WsPlan& p = h->plan;
  if (bg->warp_ids) return;
  launch_background_draw(bg->points, p.bgN, bg->id_choices, bg->num_choices, bg->noise_std, rnd ? rnd->seed : 0, rnd ? rnd->offset : 0,
                         sc ? sc->dynamic : nullptr, ws + p.bg_points, reinterpret_cast<int32_t*>(ws + p.bg_ids), stream);
}

// d_rgb[lv] != nullptr: upstream gradient mode; else MSE-loss mode against `target`.
// Launch order (round 3): the reverse passes of the two levels are independent, so every kernel type runs ONCE
// over the tiles of all levels -- composite_bwd x levels, ONE NeRF-MLP dgrad launch (coarse + fine tiles), the regularisers'
// point gradients, ONE SE3 dgrad launch (coarse + fine + background tiles), the tangent pass, then wgrad / reduce.
int backward_impl(nrf_handle h, const float* params_x, const nrf_rays* rays, const float* const d_rgb[2], const float* target,
                  float* grad_x, float* stats, float* ws, hipStream_t stream, const nrf_background* bg = nullptr,
                  const nrf_step_scalars* scalars = nullptr, const nrf_elastic* el = nullptr, const nrf_warp_reg* wr = nullptr,
                  bool bg_forward_done = false) {
  WsPlan& p = h->plan;
  const nrf_model_desc& d = h->d;
  const int B = p.B;
  const bool warp_on = h->stashed_warp;
  const char* tables = reinterpret_cast<const char*>(ws + p.tables);
  // narrower model: the stashed forward left the padded parameter image in the workspace; gradients are formed
  // in the padded layout and copied out at the end
  const float* params = h->embed ? ws + p.iparams : params_x;
  // the gradient buffer is zero-filled and accumulated into with 16-byte accesses (zero_ranges_kernel, reduce passes)
  if ((reinterpret_cast<uintptr_t>(grad_x) & 15u) != 0) return fail(NRF_E_SHAPE, "grad_params must be 16-byte aligned");
  float* grad = h->embed ? ws + p.igrad : grad_x;
  const bool wr_on = wr && warp_on;
  const bool bg_on = bg && p.bgN > 0;
  const bool el_on = el && p.elastic && warp_on;
  const bool bft = p.flags & NRF_FLAG_BF16;
  {   // everything that is accumulated into, zeroed by one launch
    ZeroArgs z;
    memset(&z, 0, sizeof(z));
    z.add(grad, h->nparams);
    if (warp_on && h->time_enc) z.add(ws + p.t_dcodes, (long long)B * h->G);
    if (wr_on) z.add(ws + p.wr_sums, 64);
    if (bg_on) z.add(ws + p.bg_loss, 64);
    for (int lv = 0; lv < h->nlevels; ++lv) z.add(ws + p.L[lv].dray, (long long)B * RGB_W);
    if (p.bwd32 && !warp_on && !bft) {   // the 32-row reverse chain ADDS its bias column sums into the workgroups' slices
      int nt_all = 0;
      for (int lv = 0; lv < h->nlevels; ++lv) nt_all += p.ntiles[lv];
      const long long g32 = 2 * nt_all < 4 * h->num_cus ? 2 * nt_all : 4 * h->num_cus;
      for (int lv = 0; lv < h->nlevels; ++lv) z.add(ws + p.L[lv].small_part, g32 * SMALL_PART);
    }
    if (z.overflow) return fail(NRF_E_STATE, "zero_ranges table full: an accumulator would stay unzeroed");
    launch_zero_ranges(z, stream);
  }
  const int G2 = 2 * h->num_cus;   // chain kernels: two workgroups per CU
  h->prof.begin("composite_bwd", 0, stream);
  {
    CompositeBwdArgs ca[2];
    for (int lv = 0; lv < h->nlevels; ++lv) {
      const LevelWs& L = p.L[lv];
      CompositeBwdArgs& c = ca[lv];
      memset(&c, 0, sizeof(c));
      c.out4 = reinterpret_cast<const float4*>(ws + L.out4); c.z = ws + L.z; c.dirs = rays->directions;
      c.B = B; c.S = p.S[lv]; c.white_bkgd = d.use_white_background; c.sample_at_inf = d.use_sample_at_infinity;
      c.sigma_act = d.sigma_activation;
      c.rgb_out = ws + L.rgb; c.target = target; c.d_rgb = target ? nullptr : d_rgb[lv];
      c.loss_scale = 2.0f / (3.0f * (float)B);   // d/d rgb of mean over (B,3) (training.py:172)
      c.d_raw4 = reinterpret_cast<float4*>(ws + L.d_raw4); c.rows_pad = p.ntiles[lv] * TILE_ROWS;
      c.mse_ray = ws + p.mse + (size_t)lv * B; c.dsig_ray = h->A > 0 ? ws + L.dsig_ray : nullptr;
    }
    launch_composite_bwd(ca[0], h->nlevels > 1 ? &ca[1] : nullptr, stream);
  }
  h->prof.end(stream);
  double mlp_rows = 0;
  for (int lv = 0; lv < h->nlevels; ++lv) mlp_rows += p.rows[lv];
  if (bft) {   // bf16 dgrad chains (both levels, one launch): dpre of every layer into the bf16 dY stash, then the per-ray condition sums
    ChainBwdBf16Args ba[2];
    for (int lv = 0; lv < h->nlevels; ++lv) {
      const LevelWs& L = p.L[lv];
      ChainBwdBf16Args& b = ba[lv];
      memset(&b, 0, sizeof(b));
      b.wpk = ws + L.bf_wpkT; b.d_raw4 = reinterpret_cast<const float4*>(ws + L.d_raw4);
      b.S = p.S[lv]; b.B = B; b.rows = p.rows[lv]; b.st = bf_stash(p, lv, ws);
      if (warp_on) {
        b.points = ws + L.wpoints; b.d_points = ws + L.d_points; b.rows_pad = p.ntiles[lv] * TILE_ROWS;
        b.F = d.num_nerf_point_freqs; b.P = h->P;
      }
    }
    h->prof.begin("mlp_dgrad", dgrad_flops_row(h, warp_on) * mlp_rows, stream);
    launch_chain_bwd_bf16(ba[0], h->nlevels > 1 ? &ba[1] : nullptr, h->num_cus, stream);
    h->prof.end(stream);
    for (int lv = 0; lv < h->nlevels; ++lv) launch_dray_bf16(ba[lv].st.drgbh, B, p.S[lv], ws + p.L[lv].dray, stream);
  } else {
    ChainBwdArgs ca[2];
    int nt_all = 0;
    for (int lv = 0; lv < h->nlevels; ++lv) {
      const LevelWs& L = p.L[lv];
      ChainBwdArgs& a = ca[lv];
      memset(&a, 0, sizeof(a));
      a.params = params; a.po = h->po[lv]; a.wpk = ws + L.wpk; a.pk = h->pk;
      a.d_raw4 = reinterpret_cast<const float4*>(ws + L.d_raw4);
      a.S = p.S[lv]; a.B = B; a.rows = p.rows[lv]; a.ntiles = p.ntiles[lv];
      a.bits_trunk = reinterpret_cast<const uint32_t*>(ws + L.bits_trunk);
      a.bits_rgbh = reinterpret_cast<const uint32_t*>(ws + L.bits_rgbh);
      a.dy_trunk = ws + L.dy_trunk; a.dy_bn = ws + L.dy_bn; a.dy_rgbh = ws + L.dy_rgbh; a.dray = ws + L.dray;
      a.small_part = ws + L.small_part;
      if (warp_on) { a.d_points = ws + L.d_points; a.st_pe = ws + L.st_pe; }
      a.F = d.num_nerf_point_freqs; a.P = h->P; a.PK = h->PK; a.skip = d.nerf_skip_layer;
      a.alpha_on_bn = h->A > 0 ? 1 : 0;
      nt_all += p.ntiles[lv];
    }
    h->prof.begin("mlp_dgrad", dgrad_flops_row(h, warp_on) * mlp_rows, stream);
    if (p.bwd32 && !warp_on) {
      const int G4 = 4 * h->num_cus;
      launch_chain_bwd32(ca[0], h->nlevels > 1 ? &ca[1] : nullptr, 2 * nt_all < G4 ? 2 * nt_all : G4, stream);
    } else {
      if (p.bwd32) return fail(NRF_E_STATE, "plan built for the 32-row reverse chain but the stashed forward ran the warp field");
      launch_chain_bwd(ca[0], h->nlevels > 1 ? &ca[1] : nullptr, nt_all < G2 ? nt_all : G2, stream);
    }
    h->prof.end(stream);
    (void)nt_all;
  }
  if (el_on) {   // training.compute_elastic_loss on the coarse samples
    const LevelWs& L = p.L[0];
    const LevelWs& T = p.L[TG];
    ElasticArgs ea;
    memset(&ea, 0, sizeof(ea));
    ea.prim_win = ws + L.w_st_win; ea.prim_wv = reinterpret_cast<const float4*>(ws + L.w_st_wv);
    if (p.bfw) ea.x_rows = ws + L.points_raw;   // bf16 trunk: no fp32 input stash
    ea.tan_wv = reinterpret_cast<const float4*>(ws + T.w_st_wv); ea.coef = ws + L.weights;
    if (el->reduce_method == NRF_ELASTIC_MEDIAN) {   // training.py:182-188
      launch_median_coef(ws + L.weights, B, p.S[0], ws + p.el_coef, stream);
      ea.coef = ws + p.el_coef; ea.res_selected = 1;
    }
    ea.tan_dw4 = reinterpret_cast<float4*>(ws + T.w_dw4); ea.tan_dv4 = reinterpret_cast<float4*>(ws + T.w_dv4);
    ea.prim_dw4 = reinterpret_cast<float4*>(ws + L.el_dw4); ea.prim_dv4 = reinterpret_cast<float4*>(ws + L.el_dv4);
    ea.part = ws + p.el_sums;
    ea.rows = p.rows[0]; ea.rows_pad = p.ntiles[0] * TILE_ROWS; ea.PKS = (h->PKw + 31) / 32 * 32;
    ea.eps = el->eps; ea.alpha = el->loss_alpha; ea.scale = el->loss_scale; ea.gscale = el->loss_weight / (float)B;
    ea.inv_rays = 1.0f / (float)B; ea.dyn = scalars ? scalars->dynamic : nullptr;
    ea.loss_type = el->loss_type;
    h->prof.begin("elastic", 0, stream);
    launch_elastic(ea, stream);
    h->prof.end(stream);
  }
  if (wr_on)   // use_warp_reg_loss (training.py:199-212): + d loss / d warped point at the median-depth sample of each ray
    for (int lv = 0; lv < h->nlevels; ++lv) {
      const LevelWs& L = p.L[lv];
      launch_warp_reg(ws + L.weights, ws + L.points_raw, ws + L.wpoints, B, p.S[lv], wr->loss_alpha, wr->loss_scale,
                      wr->loss_weight / (float)B, ws + L.d_points, ws + p.wr_sums + 2 * lv, stream);
    }
  // ---- background regulariser (training.compute_background_loss, training.py:117-135): the SE3 field on the
  //      (already noised) background points with one warp id per point; general loss of |x' - x|^2 ----
  if (bg_on) {
    const LevelWs& L = p.L[BG];
    // the background batch's warp forward ran inside the coarse warp launch of the fused train step (the only caller that
    // passes `bg`: nrf_backward has no background argument)
    if (!bg_forward_done) return fail(NRF_E_STATE, "background regulariser without its forward pass");
    launch_background_loss(bg_points_of(p, bg, ws), ws + L.wpoints, p.bgN, p.ntiles[BG] * TILE_ROWS, bg->loss_alpha, bg->loss_scale,
                           bg->loss_weight, ws + L.d_points, ws + p.bg_loss, stream);
  }
  if (warp_on) {
    WarpBwdArgs wa[3];
    int nlev = 0, nt_all = 0;
    double rows_all = 0;
    auto common = [&](WarpBwdArgs& w, int lv) {
      const LevelWs& L = p.L[lv];
      memset(&w, 0, sizeof(w));
      w.params = params; w.po = h->wpo; w.wpk = ws + p.warp_wpk; w.pk = h->wpk;
      w.nt_prim = p.ntiles[lv];
      w.d_points = ws + L.d_points; w.st_win = ws + L.w_st_win;
      w.st_wv = reinterpret_cast<const float4*>(ws + L.w_st_wv);
      w.bits = reinterpret_cast<const uint32_t*>(ws + L.w_bits);
      w.S = p.S[lv]; w.rows = p.rows[lv]; w.ntiles = p.ntiles[lv];
      w.F = h->Fw; w.G = h->G; w.Win = h->Win; w.PKw = h->PKw;
      w.dy = ws + L.w_dy; w.d_w4 = reinterpret_cast<float4*>(ws + L.w_dw4); w.d_v4 = reinterpret_cast<float4*>(ws + L.w_dv4);
      w.small_part = ws + p.L[0].w_small_part;   // one set of bias partials for the whole launch
      nt_all += p.ntiles[lv]; rows_all += p.rows[lv];
    };
    for (int lv = 0; lv < h->nlevels; ++lv) {
      WarpBwdArgs& w = wa[nlev++];
      common(w, lv);
      w.B = B;
      w.warp_ids = h->time_enc ? nullptr : rays->warp_ids;   // TimeEncoder: the code gradient is per ray
      w.grad_embed = h->time_enc ? ws + p.t_dcodes : grad + h->wpo.embed;
      if (el_on && lv == 0) { w.extra_dw4 = reinterpret_cast<const float4*>(ws + p.L[0].el_dw4); w.extra_dv4 = reinterpret_cast<const float4*>(ws + p.L[0].el_dv4); }
    }
    if (bg_on) {
      WarpBwdArgs& w = wa[nlev++];
      common(w, BG);
      w.B = p.bgN; w.S = 1;
      w.point_ids = bg_ids_of(p, bg, ws);
      w.grad_embed = grad + h->wpo.embed;
    }
    const int GW = warp_grid_mul() * h->num_cus;
    if (p.bfw) {   // bf16 trunk: the reverse stream, this level's stash, the points as fp32 rows
      int q = 0;
      for (int lv = 0; lv < h->nlevels; ++lv, ++q) {
        wa[q].bwpk = ws + p.bfw_wpkT; wa[q].bst = bfw_stash(p, lv, ws); wa[q].x_rows = ws + p.L[lv].points_raw;
        wa[q].rows_pad = p.ntiles[lv] * TILE_ROWS;
      }
      if (bg_on) {
        wa[q].bwpk = ws + p.bfw_wpkT; wa[q].bst = bfw_stash(p, BG, ws); wa[q].x_rows = bg_points_of(p, bg, ws);
        wa[q].rows_pad = p.ntiles[BG] * TILE_ROWS;
      }
    }
    h->prof.begin("warp_dgrad", warp_dgrad_flops_row(h) * rows_all, stream);
    if (p.bfw) launch_warp_bwd_bf16(wa[0], nlev > 1 ? &wa[1] : nullptr, nlev > 2 ? &wa[2] : nullptr, h->num_cus, stream);
    else launch_warp_bwd(wa[0], nlev > 1 ? &wa[1] : nullptr, nlev > 2 ? &wa[2] : nullptr, nt_all < GW ? nt_all : GW, stream);
    h->prof.end(stream);
    if (el_on) {   // reverse of the tangent pass
      const LevelWs& T = p.L[TG];
      WarpBwdArgs ta = wa[0];
      ta.tangent = 1; ta.nt_prim = p.ntiles[0]; ta.ntiles = p.ntiles[TG]; ta.rows = p.rows[TG];
      ta.extra_dw4 = ta.extra_dv4 = nullptr;
      ta.d_points = nullptr; ta.st_win = nullptr; ta.st_wv = nullptr;
      ta.dy = ws + T.w_dy; ta.d_w4 = reinterpret_cast<float4*>(ws + T.w_dw4); ta.d_v4 = reinterpret_cast<float4*>(ws + T.w_dv4);
      ta.small_part = nullptr;
      const int tgrid = p.ntiles[TG] < GW ? p.ntiles[TG] : GW;
      h->prof.begin("warp_tangent_dgrad", 3.0 * warp_dgrad_flops_row(h) * p.rows[0], stream);
      if (p.bfw) {
        ta.rows = p.rows[0]; ta.rows_pad = p.ntiles[0] * TILE_ROWS;
        ta.bst = bfw_stash(p, TG, ws); ta.bprim_bits = reinterpret_cast<const uint32_t*>(ws + p.L[0].bw_bits);
        ta.bng_prim = p.L[0].bw_ngroups;
        launch_warp_bwd_bf16(ta, nullptr, nullptr, h->num_cus, stream);
      } else {
        launch_warp_bwd(ta, nullptr, nullptr, tgrid, stream);
      }
      h->prof.end(stream);
    }
  }
  h->prof.begin("cond_wgrad", 0, stream);
  launch_cond_wgrad(ws + p.cond, ws + p.L[0].dray, h->nlevels > 1 ? ws + p.L[1].dray : nullptr, B, h->R, ws + p.L[0].cond_grad,
                    h->nlevels > 1 ? ws + p.L[1].cond_grad : nullptr, stream);
  launch_cond_embed_grad(params, ws + p.L[0].dray, h->nlevels > 1 ? ws + p.L[1].dray : nullptr, rays->appearance_ids, rays->camera_ids,
                         B, h->V, h->app_in_cond ? d.num_appearance_features : 0, h->app_off,
                         d.use_camera_metadata ? d.num_camera_features : 0, h->cam_off, h->po[0].rgbh_k,
                         h->po[h->nlevels > 1 ? 1 : 0].rgbh_k, grad, stream);
  for (int lv = 0; lv < h->nlevels; ++lv) {
    const LevelWs& L = p.L[lv];
    if (h->A > 0)   // appearance-code rows of the alpha head and the codes' gradient through it (modules.py:152-157)
      launch_alpha_cond_grad(params, ws + p.cond, ws + L.dsig_ray, rays->appearance_ids, B, h->R, h->V, h->A, h->app_off,
                             h->po[lv].alpha_k, grad, stream);
  }
  h->prof.end(stream);
  if (warp_on && h->time_enc) {   // reverse of the TimeEncoder: d codes -> its six layers' weight gradients
    TimeEncArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.params = params; ta.po = h->tpo; ta.time = rays->time; ta.B = B; ta.F = h->Ft; ta.Tin = h->Tin; ta.G = h->G;
    ta.d_codes = ws + p.t_dcodes; ta.st_in = ws + p.t_in; ta.st_h = ws + p.t_h; ta.st_dpre = ws + p.t_dpre;
    launch_time_encoder_bwd(ta, stream);
    launch_time_encoder_wgrad(ta, grad, stream);
  }
  double wg_rows = mlp_rows;
  // the SE3 groups also run over the background rows and, with the elastic regulariser, over the three tangent rows per
  // coarse sample (warping.py:385-387 jacfwd): algorithmic work of the step, counted
  double warp_wg_rows = warp_on ? mlp_rows + (bg_on ? p.bgN : 0) + (el_on ? 3.0 * p.rows[0] : 0.0) : 0.0;
  if (!p.segs.empty()) {
    h->prof.begin("wgrad", (bft ? 0.0 : wgrad_flops_row(h)) * wg_rows + warp_fwd_flops_row_or0(h) * warp_wg_rows, stream);
    launch_wgrad(reinterpret_cast<const WgradGroup*>(tables + p.groups_off_b),
                 reinterpret_cast<const WgradSegment*>(tables + p.segs_off_b),
                 reinterpret_cast<const int*>(tables + p.segbegin_off_b), p.wgrad_nwg, ws,
                 reinterpret_cast<unsigned long long*>(ws + p.seg_clock), stream);
    h->prof.end(stream);
  }
  if (!p.bsegs.empty()) {
    h->prof.begin("wgrad_bf16", wgrad_flops_row(h) * wg_rows + (p.bfw ? warp_fwd_flops_row_or0(h) * warp_wg_rows : 0.0), stream);
    launch_wgrad_bf16(reinterpret_cast<const WgradGroup*>(tables + p.bgroups_off_b),
                      reinterpret_cast<const WgradSegment*>(tables + p.bsegs_off_b),
                      reinterpret_cast<const int*>(tables + p.bsegbegin_off_b), p.bwgrad_nwg, ws, stream);
    h->prof.end(stream);
  }
  h->prof.begin("grad_reduce", 0, stream);
  const ReduceDesc* rd = reinterpret_cast<const ReduceDesc*>(tables + p.reduce_off_b);
  for (int pass = 0, at = 0; pass < 4; at += p.nreduce_pass[pass], ++pass)   // later passes add into shared leaves (SE3 field)
    if (p.nreduce_pass[pass] > 0) launch_reduce(rd + at, p.nreduce_pass[pass], ws, grad, stream);
  if (h->embed) {
    hipError_t e = hipMemsetAsync(grad_x, 0, (size_t)h->xnparams * sizeof(float), stream);
    if (e != hipSuccess) return fail_hip(e, "zero grad");
    launch_embed(reinterpret_cast<const EmbedDesc*>(tables + p.emb_off_b), (int)h->emb.size(), grad, grad_x, false, stream);
  }
  if (stats) {
    StatsArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.mse_ray = ws + p.mse; sa.B = B; sa.nlevels = h->nlevels;
    if (bg_on) { sa.bg_sum = ws + p.bg_loss; sa.bgN = p.bgN; sa.bg_weight = bg->loss_weight; }
    if (el_on) {
      sa.el_part = ws + p.el_sums; sa.el_nwg = (p.ntiles[0] * TILE_ROWS + 255) / 256; sa.el_rows = el->reduce_method == NRF_ELASTIC_MEDIAN ? B : p.rows[0]; sa.el_jac_rows = p.rows[0];
      sa.el_weight = el->loss_weight;
    }
    if (wr_on) { sa.wr_sums = ws + p.wr_sums; sa.wr_weight = wr->loss_weight; }
    sa.stats = stats; sa.dyn = scalars ? scalars->dynamic : nullptr;
    launch_finish_stats(sa, stream);
  }
  h->prof.end(stream);
  return check_launch("nrf_backward");
}

}  // namespace

extern "C" {

int nrf_version(void) { return NRF_VERSION; }
const char* nrf_last_error(void) { return g_err; }

int nrf_create(const nrf_model_desc* desc, nrf_handle* out) {
  if (!desc || !out) return fail(NRF_E_NULL, "desc / out is null");
  const nrf_model_desc& d = *desc;
  // The chains run 8 trunk layers with ONE layer that reads [h, posenc] (modules.py:41-62 concatenates the inputs in front of layer i
  // for i in skips).  The caller's trunk (depth xd <= 8, skip xs) is laid out on them:
  //   * no skip reached (nerf_skips = () or xs >= xd): layers 0..xd-1, zero posenc rows in layer 4, identity layers behind;
  //   * xs <= 4 and xd - xs <= 4: layers 0..xs-1, IDENTITY layers xs..3 (relu(h . I) = h for h >= 0: exact, also on the bf16 chain),
  //     the caller's skip layer at 4, the rest behind it -- the kernels' own layout, every mode;
  //   * any other xs in 1..7: layers in place, the skip GEMM moves to layer xs (float32 chains only: a run-time layer index there,
  //     a compile-time position in the bf16 stream).
  const bool skip_reached = d.nerf_skip_layer >= 0 && d.nerf_skip_layer < d.nerf_trunk_depth;
  if (d.nerf_trunk_depth < 1 || d.nerf_trunk_depth > TRUNK_DEPTH)
    return fail(NRF_E_UNSUPPORTED, "nerf_trunk_depth must be in [1,8]");
  if (skip_reached && d.nerf_skip_layer == 0)   // layer 0 would read [posenc, posenc]: two blocks of one leaf on the same rows
    return fail(NRF_E_UNSUPPORTED, "nerf_skips = (0,) (the first layer reading its input twice) is not built");
  if (d.nerf_trunk_width < 1 || d.nerf_trunk_width > TRUNK_W)   // narrower trunks run zero-padded (test_vrig.gin: 128)
    return fail(NRF_E_UNSUPPORTED, "nerf_trunk_width must be in [1,256]");
  if (d.nerf_rgb_branch_depth != 1 || d.nerf_rgb_branch_width < 1 || d.nerf_rgb_branch_width > RGB_W)
    return fail(NRF_E_UNSUPPORTED, "rgb branch must be 1 layer of width <= 128");
  if (d.use_trunk_condition)   // models.py:203-204: never forwarded by construct_nerf, no preset; a silent no-op would change the layout
    return fail(NRF_E_UNSUPPORTED, "use_trunk_condition (trunk conditioning) is not built");
  if (d.use_alpha_condition && d.use_appearance_metadata && (d.num_appearance_features < 1 || d.num_appearance_features > 16))
    return fail(NRF_E_SHAPE, "num_appearance_features must be in [1,16]");
  if (!(d.noise_std >= 0.f)) return fail(NRF_E_SHAPE, "noise_std must be >= 0");
  if (d.use_warp && d.warp_metadata_encoder_type != NRF_META_GLO && d.warp_metadata_encoder_type != NRF_META_TIME)
    return fail(NRF_E_UNSUPPORTED, "warp_metadata_encoder_type must be glo or time ('blend' exists only for the TranslationField, no preset)");
  if (d.use_warp && d.warp_metadata_encoder_type == NRF_META_TIME && (d.num_time_encoder_freqs < 0 || d.num_time_encoder_freqs > 8))
    return fail(NRF_E_SHAPE, "num_time_encoder_freqs must be in [0,8]");
  if (d.use_warp) {
    if (d.warp_field_type != NRF_WARP_SE3 && d.warp_field_type != NRF_WARP_TRANSLATION) return fail(NRF_E_UNSUPPORTED, "warp_field_type");
    if (d.num_warp_freqs < 0 || d.num_warp_freqs > 8) return fail(NRF_E_SHAPE, "num_warp_freqs must be in [0,8]");
    if (d.num_warp_features < 1 || d.num_warp_features > 8) return fail(NRF_E_SHAPE, "num_warp_features must be in [1,8]");
    if (d.num_warp_embeddings < 1 && d.warp_metadata_encoder_type != NRF_META_TIME) return fail(NRF_E_SHAPE, "num_warp_embeddings must be positive");
    if (d.warp_trunk_depth < 0 || d.warp_trunk_depth > WARP_DEPTH) return fail(NRF_E_UNSUPPORTED, "warp_trunk_depth must be in [1,6] (0 = 6)");
    if (d.warp_trunk_width < 0 || d.warp_trunk_width > WARP_W) return fail(NRF_E_UNSUPPORTED, "warp_trunk_width must be in [1,128] (0 = 128)");
  }
  if (d.num_coarse_samples < 3 || d.num_coarse_samples > 256) return fail(NRF_E_SHAPE, "num_coarse_samples must be in [3,256]");
  if (d.num_fine_samples < 0 || d.num_coarse_samples + d.num_fine_samples > 512)
    return fail(NRF_E_SHAPE, "num_coarse_samples + num_fine_samples must be <= 512");
  if (d.num_nerf_point_freqs < 1 || d.num_nerf_point_freqs > 10) return fail(NRF_E_SHAPE, "num_nerf_point_freqs must be in [1,10]");
  if (d.num_nerf_viewdir_freqs < 0 || d.num_nerf_viewdir_freqs > 8) return fail(NRF_E_SHAPE, "num_nerf_viewdir_freqs must be in [0,8]");
  if (d.sigma_activation != NRF_ACT_RELU && d.sigma_activation != NRF_ACT_SOFTPLUS) return fail(NRF_E_UNSUPPORTED, "sigma_activation");
  nrf_handle h = new nrf_handle_s();
  h->d = d;
  {
    const int xd = d.nerf_trunk_depth, xs = skip_reached ? d.nerf_skip_layer : -1;
    h->xdepth = xd; h->xskip = xs;                                              // what the caller's tree holds
    int imap[TRUNK_DEPTH], K = SKIP_LAYER;
    for (int i = 0; i < TRUNK_DEPTH; ++i) imap[i] = i;
    if (xs >= 0 && xs <= SKIP_LAYER && xd - xs <= TRUNK_DEPTH - SKIP_LAYER) {
      for (int i = xs; i < xd; ++i) imap[i] = SKIP_LAYER + (i - xs);
    } else if (xs >= 0) {
      K = xs;
    }
    for (int i = 0; i < TRUNK_DEPTH; ++i) h->emap[i] = -1;
    for (int i = 0; i < xd; ++i) h->emap[imap[i]] = i;
    h->d.nerf_trunk_depth = TRUNK_DEPTH; h->d.nerf_skip_layer = K;                // what the kernels run
  }
  h->nlevels = d.num_fine_samples > 0 ? 2 : 1;
  h->P = 3 + 6 * d.num_nerf_point_freqs;
  h->PK = (h->P + 15) / 16 * 16;                  // K of the posenc GEMMs: whole 16-k quads of the MFMA loop
  h->V = d.use_viewdirs ? 3 + 6 * d.num_nerf_viewdir_freqs : 0;
  h->app_in_cond = (d.use_appearance_metadata && d.use_alpha_condition) ? 1 : 0;   // models.py:206
  h->A = h->app_in_cond ? d.num_appearance_features : 0;                              // models.py:204-205
  h->warp = d.use_warp != 0;
  if (h->warp) {
    h->time_enc = d.warp_metadata_encoder_type == NRF_META_TIME;
    h->Ft = d.num_time_encoder_freqs; h->Tin = 1 + 2 * h->Ft;   // AnnealedSinusoidalEncoder of the scalar time stamp
    h->Fw = d.num_warp_freqs; h->G = d.num_warp_features;
    h->wxdepth = d.warp_trunk_depth ? d.warp_trunk_depth : WARP_DEPTH;
    h->wxwidth = d.warp_trunk_width ? d.warp_trunk_width : WARP_W;
    h->Win = 3 + 6 * h->Fw + h->G;                 // [annealed posenc, GLO code] (warping.py:326-327)
    h->PKw = (h->Win + 15) / 16 * 16;
  }
  h->R = h->V + (h->app_in_cond ? d.num_appearance_features : 0) + (d.use_camera_metadata ? d.num_camera_features : 0);
  if (h->R > 64) { delete h; return fail(NRF_E_SHAPE, "rgb condition wider than 64"); }
  build_layout(h);
  build_pack_offsets(h);
  *out = h;
  return NRF_OK;
}

int nrf_destroy(nrf_handle h) {
  delete h;
  return NRF_OK;
}

int nrf_param_count(nrf_handle h, int64_t* n) {
  if (!h || !n) return fail(NRF_E_NULL, "null");
  *n = h->xnparams;
  return NRF_OK;
}

int nrf_param_layout(nrf_handle h, nrf_tensor_info* out, int32_t* n) {
  if (!h || !n) return fail(NRF_E_NULL, "null");
  if (out) {
    if (*n < (int32_t)h->xlayout.size()) return fail(NRF_E_SHAPE, "layout array too small");
    memcpy(out, h->xlayout.data(), h->xlayout.size() * sizeof(nrf_tensor_info));
  }
  *n = (int32_t)h->xlayout.size();
  return NRF_OK;
}

// The flag word of nrf_forward / nrf_workspace_bytes*: unknown bits and contradictory combinations are refused up front
// (they used to pass through: NRF_FLAG_WARP_F32 without NRF_FLAG_BF16 was silently ignored, and TRAIN | WARP_JACOBIAN sized a
// workspace for a plan no call can run).
static int check_flags(const nrf_handle_s* h, uint32_t flags) {
  const uint32_t known = NRF_FLAG_TRAIN | NRF_FLAG_NO_WARP | NRF_FLAG_BF16 | NRF_FLAG_WARP_JACOBIAN | NRF_FLAG_WARP_F32;
  if (flags & ~known) return fail(NRF_E_UNSUPPORTED, "unknown bits in flags");
  if ((flags & NRF_FLAG_WARP_F32) && !(flags & NRF_FLAG_BF16))
    return fail(NRF_E_UNSUPPORTED, "NRF_FLAG_WARP_F32 only qualifies NRF_FLAG_BF16 (the float32 mode runs the warp trunk in float32 anyway)");
  if ((flags & NRF_FLAG_TRAIN) && (flags & NRF_FLAG_WARP_JACOBIAN))
    return fail(NRF_E_UNSUPPORTED, "NRF_FLAG_WARP_JACOBIAN is an inference output (training consumes the Jacobian through nrf_elastic)");
  if ((flags & NRF_FLAG_TRAIN) && (flags & NRF_FLAG_NO_WARP))
    return fail(NRF_E_UNSUPPORTED, "NRF_FLAG_NO_WARP cannot be combined with NRF_FLAG_TRAIN");
  if ((flags & NRF_FLAG_BF16) && h->d.nerf_skip_layer != SKIP_LAYER)
    return fail(NRF_E_UNSUPPORTED, "NRF_FLAG_BF16: the bfloat16 chains run the skip at trunk layer 4; this model's nerf_skips cannot be laid out "
                                   "around it (needs skip <= 4 and depth - skip <= 4): use the float32 mode");
  return NRF_OK;
}

int nrf_workspace_bytes(nrf_handle h, int32_t num_rays, uint32_t flags, size_t* bytes) {
  if (!h || !bytes) return fail(NRF_E_NULL, "null");
  if (num_rays <= 0) return fail(NRF_E_SHAPE, "num_rays must be positive");
  CK(check_flags(h, flags));
  query_device(h);
  build_plan(h, num_rays, flags);
  *bytes = h->plan.total_floats * sizeof(float);
  return NRF_OK;
}

int nrf_forward(nrf_handle h, const float* params, const nrf_rays* rays, const nrf_step_scalars* scalars,
                const nrf_rand* rnd, const nrf_outputs* out, uint32_t flags, void* workspace, size_t workspace_bytes,
                void* stream) {
  if (!h) return fail(NRF_E_NULL, "handle is null");
  CK(check_flags(h, flags));
  return forward_impl(h, params, rays, scalars, rnd, out, flags, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}

int nrf_backward(nrf_handle h, const float* params, const nrf_rays* rays, const float* d_rgb_coarse,
                 const float* d_rgb_fine, float* grad_params, void* workspace, size_t workspace_bytes, void* stream) {
  if (!h || !params || !rays || !grad_params || !workspace) return fail(NRF_E_NULL, "null argument");
  if (h->stashed_ws != workspace || h->stashed_B != rays->num_rays)
    return fail(NRF_E_STATE, "nrf_backward needs a preceding nrf_forward(NRF_FLAG_TRAIN) on this workspace");
  if (h->stashed_plan != h->plan.serial)   // another call re-planned the handle (other num_rays / flags) since the stashed forward
    return fail(NRF_E_STATE, "nrf_backward: the workspace layout changed since the stashed nrf_forward (an intervening call with "
                             "another num_rays / flags); run nrf_forward(NRF_FLAG_TRAIN) again");
  if (workspace_bytes < h->plan.total_floats * sizeof(float)) return fail(NRF_E_WORKSPACE, "workspace too small");
  float* ws = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  const float* zero = ws + h->plan.zero_rgb;
  if (!d_rgb_coarse || (h->nlevels > 1 && !d_rgb_fine)) {
    hipError_t e = hipMemsetAsync(ws + h->plan.zero_rgb, 0, (size_t)rays->num_rays * 3 * sizeof(float), st);
    if (e != hipSuccess) return fail_hip(e, "zero d_rgb");
  }
  const float* dr[2] = {d_rgb_coarse ? d_rgb_coarse : zero, d_rgb_fine ? d_rgb_fine : zero};
  return backward_impl(h, params, rays, dr, nullptr, grad_params, nullptr, ws, st);
}

int nrf_train_step_loss_grad(nrf_handle h, const float* params, const nrf_rays* rays, const float* target_rgb,
                             const nrf_step_scalars* scalars, const nrf_rand* rnd, float* grad_params, float* stats,
                             void* workspace, size_t workspace_bytes, void* stream) {
  if (!h || !target_rgb || !grad_params) return fail(NRF_E_NULL, "null argument");
  CK(forward_impl(h, params, rays, scalars, rnd, nullptr, NRF_FLAG_TRAIN, (float*)workspace, workspace_bytes,
                  (hipStream_t)stream));
  const float* dr[2] = {nullptr, nullptr};
  return backward_impl(h, params, rays, dr, target_rgb, grad_params, stats, (float*)workspace, (hipStream_t)stream);
}

int nrf_train_step_loss_grad_ex(nrf_handle h, const float* params, const nrf_rays* rays, const float* target_rgb,
                                const nrf_step_scalars* scalars, const nrf_rand* rnd, const nrf_background* bg,
                                const nrf_elastic* el, const nrf_warp_reg* wr, uint32_t flags, float* grad_params, float* stats,
                                void* workspace, size_t workspace_bytes, void* stream) {
  if (!h || !target_rgb || !grad_params) return fail(NRF_E_NULL, "null argument");
  if (flags & ~(uint32_t)(NRF_FLAG_BF16 | NRF_FLAG_WARP_F32)) return fail(NRF_E_UNSUPPORTED, "nrf_train_step_loss_grad_ex flags: 0, NRF_FLAG_BF16 [| NRF_FLAG_WARP_F32]");
  CK(check_flags(h, NRF_FLAG_TRAIN | flags));   // the same word nrf_workspace_bytes_ex validated (WARP_F32 without BF16 is refused here too)
  int bgN = 0;
  if (el) {
    if (!h->warp) return fail(NRF_E_UNSUPPORTED, "the elastic regulariser needs the warp field");
    if (el->reduce_method != NRF_ELASTIC_WEIGHT && el->reduce_method != NRF_ELASTIC_MEDIAN)
      return fail(NRF_E_UNSUPPORTED, "unknown elastic reduce_method");
    if (el->loss_type < NRF_ELASTIC_LOG_SVALS || el->loss_type > NRF_ELASTIC_LOG_DET)
      return fail(NRF_E_UNSUPPORTED, "unknown elastic loss_type ('nr' is not built: the reference marks it as producing NaNs)");
    if (!scalars) return fail(NRF_E_NULL, "nrf_step_scalars required");
  }
  if (wr && !h->warp) return fail(NRF_E_UNSUPPORTED, "the warp_reg loss needs the warp field");
  if (bg && bg->num_points > 0) {
    if (!h->warp) return fail(NRF_E_UNSUPPORTED, "the background regulariser needs the warp field");
    if (h->time_enc) return fail(NRF_E_UNSUPPORTED, "the background regulariser draws warp IDS (training.py:121-123): not defined for the time encoder");
    if (!bg->points) return fail(NRF_E_NULL, "background points is null");
    if (!bg->warp_ids && (!bg->id_choices || bg->num_choices <= 0))
      return fail(NRF_E_NULL, "background: give warp_ids (points already noised) or id_choices (the library draws ids and noise)");
    if (!scalars) return fail(NRF_E_NULL, "nrf_step_scalars required");
    bgN = bg->num_points;
  }
  CK(forward_impl(h, params, rays, scalars, rnd, nullptr, NRF_FLAG_TRAIN | flags, (float*)workspace, workspace_bytes,
                  (hipStream_t)stream, bgN, el ? 1 : 0, bgN > 0 ? bg : nullptr));
  const float* dr[2] = {nullptr, nullptr};
  return backward_impl(h, params, rays, dr, target_rgb, grad_params, stats, (float*)workspace, (hipStream_t)stream,
                       bgN > 0 ? bg : nullptr, scalars, el, wr, /*bg_forward_done=*/bgN > 0);
}

int nrf_workspace_bytes_ex(nrf_handle h, int32_t num_rays, uint32_t flags, int32_t num_background_points,
                           int32_t use_elastic_loss, size_t* bytes) {
  if (!h || !bytes) return fail(NRF_E_NULL, "null");
  if (num_rays <= 0 || num_background_points < 0) return fail(NRF_E_SHAPE, "bad sizes");
  if ((num_background_points > 0 || use_elastic_loss) && !h->warp)
    return fail(NRF_E_UNSUPPORTED, "the background / elastic regularisers need the warp field");
  CK(check_flags(h, flags));
  query_device(h);
  const bool tr = flags & NRF_FLAG_TRAIN;
  build_plan(h, num_rays, flags, tr ? num_background_points : 0, tr && use_elastic_loss ? 1 : 0);
  *bytes = h->plan.total_floats * sizeof(float);
  return NRF_OK;
}

// Stand-alone SE3 field on arbitrary points (create_warp_field(num_batch_dims=1), models.py:165-184; the
// field training.compute_background_loss applies, training.py:127-130).  Mini workspace:
// [pack descriptors | packed trunk weights | padded output].
namespace {
struct WarpPointsPlan { size_t desc_f, wpk_f, out_f, ctr_f, emb_f, ip_f, total_f; int ntiles; };
WarpPointsPlan warp_points_plan(nrf_handle h, int n) {
  WarpPointsPlan q;
  q.ntiles = (n + TILE_ROWS - 1) / TILE_ROWS;
  size_t o = 0;
  auto take = [&](size_t f) { size_t r = o; o = align_up(o + f, ALIGN_F); return r; };
  q.desc_f = take(16 * sizeof(PackDesc) / 4);
  q.wpk_f = take(h->wpk.total);
  q.out_f = take((size_t)q.ntiles * TILE_ROWS * 3);
  q.ctr_f = take(16);
  q.emb_f = q.ip_f = 0;
  if (h->d.warp_field_type == NRF_WARP_TRANSLATION || h->wxdepth != WARP_DEPTH || h->wxwidth != WARP_W) {
    // no rotation head in the caller's tree, or a trunk shallower / narrower than the kernels': run on the padded image
    q.emb_f = take((h->emb.size() + 1) * sizeof(EmbedDesc) / 4);
    q.ip_f = take((size_t)h->nparams);
  }
  q.total_f = o;
  return q;
}
}  // namespace

int nrf_warp_points_workspace_bytes(nrf_handle h, int32_t num_points, size_t* bytes) {
  if (!h || !bytes) return fail(NRF_E_NULL, "null");
  if (!h->warp) return fail(NRF_E_UNSUPPORTED, "model has no warp field");
  if (num_points <= 0) return fail(NRF_E_SHAPE, "num_points must be positive");
  *bytes = warp_points_plan(h, num_points).total_f * sizeof(float);
  return NRF_OK;
}

int nrf_warp_points(nrf_handle h, const float* params, const float* points, const int32_t* warp_ids, int32_t num_points,
                    const nrf_step_scalars* scalars, float* warped, void* workspace, size_t workspace_bytes, void* stream) {
  if (!h || !params || !points || !warp_ids || !scalars || !warped || !workspace) return fail(NRF_E_NULL, "null argument");
  if (!h->warp) return fail(NRF_E_UNSUPPORTED, "model has no warp field");
  if (h->time_enc) return fail(NRF_E_UNSUPPORTED, "nrf_warp_points takes warp ids: not built for the time encoder");
  if (num_points <= 0) return fail(NRF_E_SHAPE, "num_points must be positive");
  query_device(h);
  const WarpPointsPlan q = warp_points_plan(h, num_points);
  if (workspace_bytes < q.total_f * sizeof(float)) return fail(NRF_E_WORKSPACE, "workspace too small (nrf_warp_points_workspace_bytes)");
  float* ws = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  const bool padded = q.ip_f != 0;
  if (padded) {
    hipError_t e0 = hipMemcpyAsync(ws + q.emb_f, h->emb.data(), h->emb.size() * sizeof(EmbedDesc), hipMemcpyHostToDevice, st);
    if (e0 != hipSuccess) return fail_hip(e0, "upload embed table");
    e0 = hipMemsetAsync(ws + q.ip_f, 0, (size_t)h->nparams * sizeof(float), st);
    if (e0 != hipSuccess) return fail_hip(e0, "zero padded params");
    launch_embed(reinterpret_cast<const EmbedDesc*>(ws + q.emb_f), (int)h->emb.size(), params, ws + q.ip_f, true, st);
    params = ws + q.ip_f;
  }
  const WarpParamOffsets& wo = padded ? h->wpo : h->xwpo;   // else the caller's buffer: external offsets
  if (h->wp_pack.empty() || h->wp_pack_base != (int64_t)q.wpk_f) {
    h->wp_pack.clear();
    const WarpParamOffsets& w = wo;
    const WarpPackOffsets& wk = h->wpk;
    auto addw = [&](int64_t src, int dst, int row0, int kvalid, int K) {
      PackDesc d;
      d.src_off = src; d.dst_off = (int64_t)q.wpk_f + dst; d.src_ld = WARP_W; d.src_row0 = row0; d.kvalid = kvalid; d.K = K;
      d.ncb = 1; d.transposed = 0; d.nwaves = 4; d.nvalid = 1 << 30;
      h->wp_pack.push_back(d);
    };
    addw(w.trunk_k[0], wk.fwd_L[0], 0, h->Win, h->PKw);
    for (int l = 1; l < WARP_DEPTH; ++l) addw(w.trunk_k[l], wk.fwd_L[l], 0, WARP_W, WARP_W);
    addw(w.trunk_k[WARP_SKIP], wk.fwd_L4b, WARP_W, h->Win, h->PKw);
    h->wp_pack_base = (int64_t)q.wpk_f;
  }
  hipError_t e = hipMemcpyAsync(ws + q.desc_f, h->wp_pack.data(), h->wp_pack.size() * sizeof(PackDesc), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return fail_hip(e, "upload warp pack table");
  launch_pack(reinterpret_cast<const PackDesc*>(ws + q.desc_f), (int)h->wp_pack.size(), params, ws, st);
  WarpFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.params = params; a.po = wo; a.wpk = ws + q.wpk_f; a.pk = h->wpk;
  a.points_in = points; a.point_ids = warp_ids; a.points_out = ws + q.out_f;
  a.embed_table = params + wo.embed;
  a.S = 1; a.B = num_points; a.rows = num_points; a.ntiles = q.ntiles;
  a.F = h->Fw; a.G = h->G; a.Win = h->Win; a.PKw = h->PKw; a.alpha = scalars->warp_alpha; a.dyn = scalars->dynamic;
  const int grid = q.ntiles < 2 * h->num_cus ? q.ntiles : 2 * h->num_cus;
  e = hipMemsetAsync(ws + q.ctr_f, 0, 16 * sizeof(int), st);
  if (e != hipSuccess) return fail_hip(e, "zero tile counter");
  a.tile_counter = tile_counter_or_null(ws + q.ctr_f, 0);
  launch_warp_fwd(a, nullptr, false, grid, st);
  e = hipMemcpyAsync(warped, ws + q.out_f, (size_t)num_points * 3 * sizeof(float), hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return fail_hip(e, "copy warped points");
  return check_launch("nrf_warp_points");
}

int nrf_set_option(nrf_handle h, int32_t option, int64_t value) {
  if (!h) return fail(NRF_E_NULL, "handle is null");
  if (option == NRF_OPT_CHAIN_TILE_ROWS) {
    if (value != 0 && value != 32 && value != 64) return fail(NRF_E_UNSUPPORTED, "NRF_OPT_CHAIN_TILE_ROWS: 0 (automatic), 32 or 64");
    if (h->chain_rows_opt != (int)value) {
      h->chain_rows_opt = (int)value;
      h->stashed_ws = nullptr;   // a stash written under the old plan is not differentiated under the new one
      h->uploaded_ws = nullptr;  // ... and the next call uploads the re-planned tables
    }
    return NRF_OK;
  }
  if (option == NRF_OPT_BF16_WGRAD_MERGE) {
    if (value != 0 && value != 1) return fail(NRF_E_UNSUPPORTED, "NRF_OPT_BF16_WGRAD_MERGE: 0 or 1");
    if (h->bf16_wgrad_merge != (int)value) {
      h->bf16_wgrad_merge = (int)value;
      h->stashed_ws = nullptr;
      h->uploaded_ws = nullptr;
    }
    return NRF_OK;
  }
  return fail(NRF_E_UNSUPPORTED, "unknown option");
}

int nrf_profile_enable(nrf_handle h, int32_t on) {
  if (!h) return fail(NRF_E_NULL, "handle is null");
  h->prof.drain();
  h->prof.acc.clear();
  h->prof.on = on != 0;
  return NRF_OK;
}

int nrf_profile_read(nrf_handle h, nrf_profile_entry* out, int32_t* n) {
  if (!h || !n) return fail(NRF_E_NULL, "null");
  h->prof.drain();
  const int32_t cnt = (int32_t)h->prof.acc.size();
  if (out) {
    if (*n < cnt) return fail(NRF_E_SHAPE, "profile array too small");
    for (int32_t i = 0; i < cnt; ++i) {
      memset(&out[i], 0, sizeof(out[i]));
      snprintf(out[i].name, sizeof(out[i].name), "%s", h->prof.acc[i].name.c_str());
      out[i].ms = h->prof.acc[i].ms; out[i].launches = h->prof.acc[i].launches; out[i].flops_per_launch = h->prof.acc[i].flops;
    }
    h->prof.acc.clear();
  }
  *n = cnt;
  return NRF_OK;
}

int nrf_debug_ws_offset(nrf_handle h, const char* name, int32_t level, int64_t* float_offset) {
  if (!h || !name || !float_offset) return fail(NRF_E_NULL, "null");
  if (level < 0 || level > 3 || h->plan.B < 0) return fail(NRF_E_STATE, "no workspace plan yet / bad level");
  const LevelWs& L = h->plan.L[level];
  const struct { const char* n; size_t v; } tab[] = {
      {"st_pe", L.st_pe}, {"st_h", L.st_h}, {"st_bn", L.st_bn}, {"st_rgbh", L.st_rgbh}, {"dy_trunk", L.dy_trunk},
      {"dy_bn", L.dy_bn}, {"dy_rgbh", L.dy_rgbh}, {"d_raw4", L.d_raw4}, {"z", L.z}, {"out4", L.out4},
      {"wpoints", L.wpoints}, {"d_points", L.d_points}, {"w_st_win", L.w_st_win}, {"w_st_h", L.w_st_h},
      {"w_st_wv", L.w_st_wv}, {"w_dy", L.w_dy}, {"w_dw4", L.w_dw4}, {"w_dv4", L.w_dv4},
      {"w_bits", L.w_bits}, {"bits_trunk", L.bits_trunk}, {"bits_rgbh", L.bits_rgbh},
      {"b_pe", L.b_pe}, {"b_h", L.b_h}, {"b_bn", L.b_bn}, {"b_rgbh", L.b_rgbh}, {"b_bits", L.b_bits}, {"b_dy", L.b_dy},
      {"b_dbn", L.b_dbn}, {"b_drgbh", L.b_drgbh}, {"b_dsmall", L.b_dsmall},
      {"bw_in", L.bw_in}, {"bw_h", L.bw_h}, {"bw_bits", L.bw_bits}, {"bw_dy", L.bw_dy}, {"bw_dhead", L.bw_dhead},
      {"points_raw", L.points_raw},
      {"bg_points", h->plan.bg_points}, {"bg_ids", h->plan.bg_ids},
      {"timeline", h->plan.timeline + (size_t)(level & 1) * 2 * (256 + 512 + 4 * 2048)}};
  for (const auto& t : tab)
    if (!strcmp(t.n, name)) { *float_offset = (int64_t)t.v; return NRF_OK; }
  return fail(NRF_E_SHAPE, "unknown workspace buffer name");
}

int nrf_debug_wgrad_segments(nrf_handle h, const void* workspace, double* out, int32_t* n) {
  if (!h || !n) return fail(NRF_E_NULL, "null");
  const WsPlan& p = h->plan;
  const int32_t cnt = (int32_t)p.segs.size();
  if (out) {
    if (*n < cnt || !workspace) return fail(NRF_E_SHAPE, "segment array too small / workspace null");
    std::vector<unsigned long long> clk(cnt);
    hipError_t e = hipMemcpy(clk.data(), (const float*)workspace + p.seg_clock, cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail_hip(e, "read segment clocks");
    int wg = 0;
    for (int32_t i = 0; i < cnt; ++i) {
      while (wg + 1 < (int)p.seg_begin.size() && p.seg_begin[wg + 1] <= i) ++wg;
      const WgradGroup& g = p.groups[p.segs[i].group];
      out[6 * i + 0] = wg; out[6 * i + 1] = p.segs[i].group; out[6 * i + 2] = g.Kb; out[6 * i + 3] = g.Nb;
      out[6 * i + 4] = p.segs[i].tile_end - p.segs[i].tile_begin; out[6 * i + 5] = (double)clk[i];
    }
  }
  *n = cnt;
  return NRF_OK;
}

int nrf_adam_step(float* params, float* m, float* v, const float* grad, int64_t n, double lr, double beta1, double beta2,
                  double eps, int64_t step, double grad_scale, void* stream) {
  if (!params || !m || !v || !grad) return fail(NRF_E_NULL, "null argument");
  if (n <= 0) return fail(NRF_E_SHAPE, "n must be positive");
  launch_adam(params, m, v, grad, n, lr, beta1, beta2, eps, step, grad_scale, (hipStream_t)stream);
  return check_launch("nrf_adam_step");
}

int nrf_dynamic_scalars_write(nrf_dynamic_scalars* device_dst, const nrf_dynamic_scalars* host_values, void* stream) {
  if (!device_dst || !host_values) return fail(NRF_E_NULL, "null argument");
  launch_dynamic_write(device_dst, *host_values, (hipStream_t)stream);
  return check_launch("nrf_dynamic_scalars_write");
}

int nrf_adam_step_dynamic(float* params, float* m, float* v, const float* grad, int64_t n, double beta1, double beta2, double eps,
                          const nrf_dynamic_scalars* dynamic, void* stream) {
  if (!params || !m || !v || !grad || !dynamic) return fail(NRF_E_NULL, "null argument");
  if (n <= 0) return fail(NRF_E_SHAPE, "n must be positive");
  launch_adam_dynamic(params, m, v, grad, n, beta1, beta2, eps, dynamic, (hipStream_t)stream);
  return check_launch("nrf_adam_step_dynamic");
}

int nrf_sample_along_rays(const float* origins, const float* directions, int32_t num_rays, int32_t num_samples,
                          float near_plane, float far_plane, int32_t stratified, int32_t linear_disparity,
                          const float* t_rand, uint64_t seed, uint64_t offset, float* z_vals, void* stream) {
  (void)origins; (void)directions;   // z_vals do not depend on the ray; points are formed inside the MLP kernel
  if (!z_vals) return fail(NRF_E_NULL, "z_vals is null");
  if (num_rays <= 0 || num_samples < 2) return fail(NRF_E_SHAPE, "bad shape");
  launch_sample_coarse(t_rand, num_rays, num_samples, near_plane, far_plane, stratified, linear_disparity, seed, offset,
                       nullptr, z_vals, (hipStream_t)stream);
  return check_launch("nrf_sample_along_rays");
}

int nrf_volumetric_rendering(const float* rgb_sigma, const float* z_vals, const float* directions, int32_t num_rays,
                             int32_t num_samples, int32_t white_background, int32_t sample_at_infinity,
                             const nrf_level_out* out, void* stream) {
  if (!rgb_sigma || !z_vals || !directions || !out) return fail(NRF_E_NULL, "null argument");
  if (num_rays <= 0 || num_samples < 1 || num_samples > 512) return fail(NRF_E_SHAPE, "num_samples must be in [1,512]");
  launch_composite_fwd(reinterpret_cast<const float4*>(rgb_sigma), z_vals, directions, num_rays, num_samples,
                       white_background, sample_at_infinity, out->rgb, out->depth, out->med_depth, out->acc, out->weights,
                       (hipStream_t)stream);
  return check_launch("nrf_volumetric_rendering");
}

int nrf_sample_pdf(const float* z_coarse, const float* weights_coarse, int32_t num_rays, int32_t num_coarse,
                   int32_t num_fine, int32_t stratified, const float* u, uint64_t seed, uint64_t offset, float* z_out,
                   void* stream) {
  if (!z_coarse || !weights_coarse || !z_out) return fail(NRF_E_NULL, "null argument");
  if (num_rays <= 0 || num_coarse < 3 || num_coarse > 256 || num_fine < 1 || num_coarse + num_fine > 512)
    return fail(NRF_E_SHAPE, "need 3 <= num_coarse <= 256 and num_coarse + num_fine <= 512");
  launch_sample_fine(z_coarse, weights_coarse, num_rays, num_coarse, num_fine, stratified, u, seed, offset, nullptr, z_out,
                     (hipStream_t)stream);
  return check_launch("nrf_sample_pdf");
}

namespace {
int camera_args(const nrf_camera* cam, CameraArgs* c) {
  if (!cam) return fail(NRF_E_NULL, "camera is null");
  if (!(cam->focal_length > 0.f) || !(cam->pixel_aspect_ratio > 0.f))
    return fail(NRF_E_SHAPE, "focal_length and pixel_aspect_ratio must be positive");
  if (cam->image_size[0] <= 0 || cam->image_size[1] <= 0) return fail(NRF_E_SHAPE, "image_size must be positive");
  for (int i = 0; i < 9; ++i) c->R[i] = cam->orientation[i];
  for (int i = 0; i < 3; ++i) c->pos[i] = cam->position[i];
  c->focal = cam->focal_length;
  c->cx = cam->principal_point[0];
  c->cy = cam->principal_point[1];
  c->skew = cam->skew;
  c->aspect = cam->pixel_aspect_ratio;
  c->k1 = cam->radial_distortion[0];
  c->k2 = cam->radial_distortion[1];
  c->k3 = cam->radial_distortion[2];
  c->p1 = cam->tangential_distortion[0];
  c->p2 = cam->tangential_distortion[1];
  c->width = cam->image_size[0];
  c->height = cam->image_size[1];
  // has_radial_distortion or has_tangential_distortion (camera.py:232): the solver is skipped, not run to a no-op
  c->distorted = (c->k1 != 0.f || c->k2 != 0.f || c->k3 != 0.f || c->p1 != 0.f || c->p2 != 0.f) ? 1 : 0;
  return NRF_OK;
}
bool aligned8(const void* p) { return ((uintptr_t)p & 7u) == 0; }
}  // namespace

int nrf_camera_pixels_to_rays(const nrf_camera* camera, const float* pixels, int64_t n, float* origins,
                              float* directions, float* pixels_out, void* stream) {
  CameraArgs c;
  CK(camera_args(camera, &c));
  if (!directions) return fail(NRF_E_NULL, "directions is null");
  if (n <= 0) return fail(NRF_E_SHAPE, "n must be positive");
  if (!pixels && n != (int64_t)c.width * c.height)
    return fail(NRF_E_SHAPE, "pixels == NULL renders the pixel centres: n must equal width*height");
  if (!aligned8(pixels) || !aligned8(pixels_out)) return fail(NRF_E_SHAPE, "pixel buffers must be 8-byte aligned");
  launch_camera_rays(c, pixels, nullptr, (long)n, origins, directions, pixels_out, (hipStream_t)stream);
  return check_launch("nrf_camera_pixels_to_rays");
}

int nrf_camera_pixels_to_points(const nrf_camera* camera, const float* pixels, const float* depth, int64_t n,
                                float* points, void* stream) {
  CameraArgs c;
  CK(camera_args(camera, &c));
  if (!pixels || !depth || !points) return fail(NRF_E_NULL, "pixels / depth / points is null");
  if (n <= 0) return fail(NRF_E_SHAPE, "n must be positive");
  if (!aligned8(pixels)) return fail(NRF_E_SHAPE, "pixel buffers must be 8-byte aligned");
  launch_camera_rays(c, pixels, depth, (long)n, nullptr, points, nullptr, (hipStream_t)stream);
  return check_launch("nrf_camera_pixels_to_points");
}

int nrf_camera_project(const nrf_camera* camera, const float* points, int64_t n, float* pixels, void* stream) {
  CameraArgs c;
  CK(camera_args(camera, &c));
  if (!points || !pixels) return fail(NRF_E_NULL, "points / pixels is null");
  if (n <= 0) return fail(NRF_E_SHAPE, "n must be positive");
  if (!aligned8(pixels)) return fail(NRF_E_SHAPE, "pixel buffers must be 8-byte aligned");
  launch_camera_project(c, points, (long)n, pixels, (hipStream_t)stream);
  return check_launch("nrf_camera_project");
}

}  // extern "C"
