// Launch sequences of the C-ABI layer: forward_impl stands in for NerfModel.apply (models.py:289-375), backward_impl for the gradient
// half of training.train_step (training.py:168-265) incl. the regularisers; both run on the plan nrf_plan.hip built.  See nrf_handle.h.
#include "nrf_handle.h"

using namespace nrf;
using namespace nrf::api;

namespace nrf {
namespace api {

int check_launch(const char* where) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, where);
  return NRF_OK;
}


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
                 const nrf_outputs* out, uint32_t flags, float* ws, size_t ws_bytes, hipStream_t stream, int bgN,
                 int elastic, const nrf_background* bg) {
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
  const bool x3 = (flags & NRF_FLAG_BF16X3) != 0;   // split-bf16 NeRF chains and SE3 trunk (inference; check_flags)
  // the SE3 trunk follows the MLPs into bf16 unless the caller opts out (NRF_FLAG_WARP_F32) or asks for the Jacobian output
  // (inference tangent pass: fp32 kernels); a training plan has decided already (its stash layout depends on it)
  const bool bfw_on = warp_on && bf16 && (train ? p.bfw : !(flags & NRF_FLAG_WARP_F32) && !jac);
  if (bf16 || x3) launch_bf16_pack(reinterpret_cast<const RcPackDesc*>(ws + p.bf_desc), (int)p.bfpack.size(), params, ws, stream);
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
    const bool c32 = !bf16 && !x3 && chain32_for(h, p.ntiles[lv]);   // 32-row half tiles, four workgroups per CU
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
      if (x3 && !jac && !(flags & NRF_FLAG_WARP_F32)) {   // SE3 trunk in split-bf16 arithmetic (warp_bf16x3.hip); the Jacobian output keeps the float32 kernels (their input stash)
        WarpFwdArgs wa = warp_fwd_args(h, lv, params, rays, scalars, ws, false);
        wa.bwpk = ws + p.bfw_wpk; wa.rows_pad = p.ntiles[lv] * TILE_ROWS;
        launch_warp_fwd_x3(wa, h->num_cus, stream);
      } else if (bfw_on) {   // SE3 trunk on bf16 operands (warp_bf16.hip); one workgroup per CU, 256 rows per iteration
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
      launch_chain_fwd_bf16(a, h->num_cus, stream);
    } else if (x3) {   // one four-wave workgroup per CU (150 KiB ring), 128 samples per workgroup iteration
      a.wpk = ws + L.bf_wpk;
      launch_chain_fwd_x3(a, h->num_cus, stream);
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
                  float* grad_x, float* stats, float* ws, hipStream_t stream, const nrf_background* bg,
                  const nrf_step_scalars* scalars, const nrf_elastic* el, const nrf_warp_reg* wr,
                  bool bg_forward_done) {
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
  // one launch: the grid's columns are the chain heads, the later passes into shared leaves (SE3 field) hang behind them (nrf_plan.hip)
  if (p.nreduce_pass[0] > 0) launch_reduce(rd, 0, p.nreduce_pass[0], ws, grad, stream);
  if (p.nreduce_pass[2] > 0) launch_reduce(rd, p.nreduce_pass[0] + p.nreduce_pass[1], p.nreduce_pass[2], ws, grad, stream);
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


}  // namespace api
}  // namespace nrf
