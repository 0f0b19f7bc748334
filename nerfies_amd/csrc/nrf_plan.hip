// Plan side of the C-ABI layer: the flat-parameter layout a handle owns (external = the caller's flax tree, internal = what the kernels
// index), the weight-pack tables, and the workspace plan of one (model, num_rays, flags): sub-buffer offsets, the stream-K partitions of
// both wgrad kernels, the reduce / pack / embed descriptor tables and their upload.  See nrf_handle.h.
#include "nrf_handle.h"

using namespace nrf;
using namespace nrf::api;

namespace nrf {
namespace api {


// Appends a leaf to the internal layout (rows x cols = what the kernels index) and to the external one
// (xrows x xcols = what the model owns; defaults to the same).  External rows >= split sit `shift` rows lower inside.
// xname: the leaf's path in the caller's tree when it differs from the internal one; "" = internal only (no external
// leaf: stays zero in the padded image, its gradient is dropped).
static void add_leaf(nrf_handle h, const std::string& name, int rows, int cols, int64_t* off_out, int xrows = -1, int xcols = -1,
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
// automatic choice of the forward chain's tiling (chain32_for): 32-row tiles when the launch has fewer than this many 64-row
// tiles per CU (the 64-row grid of two workgroups per CU is then not filled)
constexpr int AUTO32_FWD_BELOW_TILES_PER_CU = 2;

// the flags a workspace layout depends on: TRAIN, WARP_JACOBIAN, and BF16 together with TRAIN (bf16 stash instead of fp32)
uint32_t plan_flags(uint32_t flags) {
  uint32_t f = flags & (NRF_FLAG_TRAIN | NRF_FLAG_WARP_JACOBIAN);
  if ((flags & NRF_FLAG_TRAIN) && (flags & NRF_FLAG_BF16)) f |= NRF_FLAG_BF16 | (flags & NRF_FLAG_WARP_F32);
  if (!(flags & NRF_FLAG_TRAIN) && (flags & NRF_FLAG_BF16X3)) f |= NRF_FLAG_BF16X3;   // its own (tripled) weight streams
  return f;
}

// Rows per workgroup tile of the float32 NeRF chain kernels for a launch over `ntiles` 64-row tiles: true = 32-row half tiles,
// four workgroups per CU (mlp_chain32.hip).  NRF_OPT_CHAIN_TILE_ROWS forces either.  Automatic = what the round-5 A/B measured
// (profiles/r05_chain32_ab.md): in steady state the 64-row kernels win by 3-5 % (forward 130 vs 123.5 TF, reverse 129 vs 125,
// eval forward 137 vs 132: every B operand float feeds one MFMA instead of two), but a launch that cannot fill the 64-row grid
// twice over -- fewer than two tiles per workgroup slot, e.g. one GPU's 128-ray share of a 1024-ray batch: 128 + 384 tiles for
// 512 slots -- runs 12-34 % faster on half tiles (coarse forward 0.160 -> 0.106 ms, fine 0.301 -> 0.264 ms).  The reverse
// chain never won (0.303 -> 0.327 ms at 512 tiles): its automatic choice stays 64.
bool chain32_for(const nrf_handle_s* h, int ntiles, bool reverse) {
  if (h->chain_rows_opt == 32) return true;
  if (h->chain_rows_opt == 64) return false;
  return !reverse && ntiles < AUTO32_FWD_BELOW_TILES_PER_CU * h->num_cus;
}

void build_plan(nrf_handle h, int B, uint32_t flags, int bgN, int elastic) {
  WsPlan& p = h->plan;
  flags = plan_flags(flags);
  if (p.B == B && p.flags == flags && p.bgN == bgN && p.elastic == elastic && p.chain_rows_opt == h->chain_rows_opt &&
      p.bf16_wgrad_merge == h->bf16_wgrad_merge) return;
  const nrf_model_desc& d = h->d;
  const bool train = flags & NRF_FLAG_TRAIN;
  const bool bft = train && (flags & NRF_FLAG_BF16);   // bf16 training: the NeRF MLPs stash / differentiate in bfloat16
  const bool x3 = !train && (flags & NRF_FLAG_BF16X3);  // split-bf16 inference chains (mlp_bf16x3.hip)
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
  // 256x256 tile; round 3 (asm LDS-DMA + 160 KiB ring: the narrow groups are no longer latency-bound): 8x8 = 14.8 us; round 6 (SGPR piece
  // tables, refill behind the MFMAs): re-fitted on config A and the vrig shape (gpurun_out/r6h: every narrow type ~6 % cheaper relative
  // to 8x8, a segment 0.4-0.5 tiles; the two-vector SE3 heads 0.155: the fit's Kb = 4, Nb = 0 row mixes them with the rgb logits)
  const double c_vec256 = env_cost("NRF_COST_VEC256", 0.106), c_vec128 = env_cost("NRF_COST_VEC128", 0.094),
               c_vec128x2 = env_cost("NRF_COST_VEC128X2", 0.155),   // SE3 heads: two vectors against one pass over h6
               c_pe = env_cost("NRF_COST_PE", 0.270),               // 2 x 8 blocks: posenc rows of the NeRF trunk
               c_rgbh = env_cost("NRF_COST_RGBH", 0.516),           // 8 x 4
               c_44 = env_cost("NRF_COST_44", 0.266),               // 4 x 4: SE3 trunk layers
               c_pe128 = env_cost("NRF_COST_PE128", 0.141),         // 2 x 4: SE3 trunk input rows
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
      // x3: the same GEMM sequence with every k-step row doubled (W_hi, W_lo); the kernel cuts a panel's rows into chunks itself
      const size_t fwd_kb = x3 ? BF_X3_STREAM_KB : BF_FWD_STREAM_KB;
      p.L[lv].bf_wpk = take(fwd_kb * 256);   // KiB -> floats
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
            e.oblk0 = pn * pb; e.x3 = x3 && !tr;
            p.bfpack.push_back(e);
            row += nrows;
          };
          if (bias >= 0) emit(1, bias, 0, 0, 0, 1);
          else if (bias == -2) emit(2, 0, 0, 0, 0, 1);   // a zero row where the kernel runs a bias-style k-step this model does not use
          for (const Part& q : parts) emit(0, q.leaf, q.ld, q.row0, q.krows, (x3 && !tr ? 2 : 1) * 2 * q.nin);
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
      p.bf_stream_ok = at == fwd_kb * 256;
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
    if (h->warp) {   // bf16 SE3 trunk (warp_bf16.hip): forward stream (also for bf16 inference), reverse stream (training); x3: its doubled rows (warp_bf16x3.hip)
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
            e.oblk0 = pn * pb; e.src_off2 = src2 >= 0 ? src2 : 0; e.split = src2 >= 0 ? split : 0; e.x3 = x3 && !tr;
            p.bfpack.push_back(e);
            row += nrows;
          };
          if (bias >= 0) emit(1, bias, bias2, bsplit, 0, 0, 0, 1);
          for (const Part& q : parts) emit(0, q.leaf, q.leaf2, q.split, q.ld, q.row0, q.krows, (x3 && !tr ? 2 : 1) * 2 * q.nin);
          at += (size_t)row * pb * 256;
        }
      };
      const size_t wfwd_kb = x3 ? BFW_X3_STREAM_KB : BFW_FWD_STREAM_KB;
      p.bfw_wpk = take(wfwd_kb * 256);
      base = p.bfw_wpk;
      gemm(2, 4, WARP_W, w.trunk_b[0], -1, 0, {{w.trunk_k[0], WARP_W, 0, h->Win, 2}});
      for (int l = 1; l < WARP_DEPTH; ++l) {
        if (l == WARP_SKIP) gemm(2, 4, WARP_W, w.trunk_b[l], -1, 0, {{w.trunk_k[l], WARP_W, 0, WARP_W, 4}, {w.trunk_k[l], WARP_W, WARP_W, h->Win, 2}});
        else gemm(2, 4, WARP_W, w.trunk_b[l], -1, 0, {{w.trunk_k[l], WARP_W, 0, WARP_W, 4}});
      }
      gemm(1, 1, 6, w.w_b, w.v_b, 3, {{w.w_k, 3, 0, WARP_W, 4, w.v_k, 3}});     // heads: columns 0..2 = w, 3..5 = v
      p.bf_stream_ok = p.bf_stream_ok && at == wfwd_kb * 256;
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
  {
    // Every pass of a destination in ONE launch: the descriptors that add into a leaf (the SE3 field's: fine level, tangent pass,
    // background batch) are chained behind the pass-0 descriptor of the same destination window; a workgroup column of reduce_kernel
    // walks the chain with one element -> thread mapping.  Heads first (the launch's grid), chained descriptors behind them.
    std::vector<ReduceDesc> all = p.reduce;
    all.insert(all.end(), reduce2.begin(), reduce2.end());
    all.insert(all.end(), reduce3.begin(), reduce3.end());
    all.insert(all.end(), reduce4.begin(), reduce4.end());
    auto wide_ok = [](const ReduceDesc& d) {
      return ((d.cols | d.src_ld | d.dst_ld) & 3) == 0 && (d.part_stride & 3) == 0 && (d.src_off & 3) == 0 && (d.dst_off & 3) == 0;
    };
    const int n = (int)all.size();
    std::vector<int> prev(n, -1), nxt(n, -1);
    for (int i = 0; i < n; ++i) {
      if (all[i].accumulate == 0) continue;
      for (int j = i - 1; j >= 0; --j)   // the latest earlier descriptor of the same destination window that is still a chain's tail
        if (nxt[j] < 0 && all[j].dst_off == all[i].dst_off && all[j].rows == all[i].rows && all[j].cols == all[i].cols &&
            all[j].dst_ld == all[i].dst_ld && all[j].accumulate < all[i].accumulate) { prev[i] = j; nxt[j] = i; break; }
    }
    std::vector<int> order, pos(n, -1);
    for (int i = 0; i < n; ++i) if (prev[i] < 0 && all[i].accumulate == 0) order.push_back(i);          // heads of pass 0
    const int nheads0 = (int)order.size();
    for (int i = 0; i < n; ++i) if (prev[i] >= 0) order.push_back(i);                                    // chained
    const int nchained_end = (int)order.size();
    for (int i = 0; i < n; ++i) if (prev[i] < 0 && all[i].accumulate != 0) order.push_back(i);          // no pass-0 partner: a second launch
    for (int k = 0; k < (int)order.size(); ++k) pos[order[k]] = k;
    p.reduce.clear();
    for (int k = 0; k < (int)order.size(); ++k) {
      ReduceDesc d = all[order[k]];
      d.next = nxt[order[k]] >= 0 ? pos[nxt[order[k]]] : -1;
      p.reduce.push_back(d);
    }
    for (int k = 0; k < (int)p.reduce.size(); ++k) {   // one mapping per chain, chosen at its head
      if (k >= nheads0 && k < nchained_end) continue;
      bool tall = p.reduce[k].rows == 1, wide = true, big = false;
      for (int q = k; q >= 0; q = p.reduce[q].next) { wide = wide && wide_ok(p.reduce[q]); big = big || p.reduce[q].nparts >= 64; }
      const int path = (tall && big) ? 2 : wide ? 1 : 0;
      for (int q = k; q >= 0; q = p.reduce[q].next) p.reduce[q].path = path;
    }
    p.nreduce_pass[0] = nheads0;
    p.nreduce_pass[1] = nchained_end - nheads0;            // reached through `next`, not launched
    p.nreduce_pass[2] = (int)order.size() - nchained_end;  // launched second (empty in every configuration built so far)
    p.nreduce_pass[3] = 0;
  }
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


}  // namespace api
}  // namespace nrf
