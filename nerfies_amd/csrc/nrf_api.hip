// C-ABI entry points of the MI355X nerfies hot path (include/nerfies_amd.h is the contract): handle life cycle, validation, the
// per-op and debug entry points.  The plan lives in nrf_plan.hip, the launch sequences in nrf_run.hip (nrf_handle.h).
#include "nrf_handle.h"

using namespace nrf;
using namespace nrf::api;

thread_local char nrf::api::g_err[256] = "ok";

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
  const uint32_t known = NRF_FLAG_TRAIN | NRF_FLAG_NO_WARP | NRF_FLAG_BF16 | NRF_FLAG_WARP_JACOBIAN | NRF_FLAG_WARP_F32 | NRF_FLAG_BF16X3;
  if (flags & ~known) return fail(NRF_E_UNSUPPORTED, "unknown bits in flags");
  if ((flags & NRF_FLAG_BF16X3) && (flags & (NRF_FLAG_TRAIN | NRF_FLAG_BF16)))
    return fail(NRF_E_UNSUPPORTED, "NRF_FLAG_BF16X3 is an inference mode of its own: not with NRF_FLAG_TRAIN (the training chains stash float32 or "
                                   "bfloat16 activations) and not with NRF_FLAG_BF16");
  if ((flags & NRF_FLAG_WARP_F32) && !(flags & (NRF_FLAG_BF16 | NRF_FLAG_BF16X3)))
    return fail(NRF_E_UNSUPPORTED, "NRF_FLAG_WARP_F32 only qualifies NRF_FLAG_BF16 / NRF_FLAG_BF16X3 (the float32 mode runs the warp trunk in float32 anyway)");
  if ((flags & NRF_FLAG_TRAIN) && (flags & NRF_FLAG_WARP_JACOBIAN))
    return fail(NRF_E_UNSUPPORTED, "NRF_FLAG_WARP_JACOBIAN is an inference output (training consumes the Jacobian through nrf_elastic)");
  if ((flags & NRF_FLAG_TRAIN) && (flags & NRF_FLAG_NO_WARP))
    return fail(NRF_E_UNSUPPORTED, "NRF_FLAG_NO_WARP cannot be combined with NRF_FLAG_TRAIN");
  if ((flags & (NRF_FLAG_BF16 | NRF_FLAG_BF16X3)) && h->d.nerf_skip_layer != SKIP_LAYER)
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
