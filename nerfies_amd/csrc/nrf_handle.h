// Internal to the C-ABI layer (nrf_plan.hip / nrf_run.hip / nrf_api.hip): the handle, the workspace plan, the region profiler and the
// functions the three translation units call across.  Round 6: csrc/nrf_api.hip (2,400 lines: plan building, descriptor tables, launch
// sequences, every entry point, debug hooks) was cut along those lines, no behaviour change.
#pragma once
#include "../../include/nerfies_amd.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <string>
#include <vector>

#include "nrf_internal.h"

namespace nrf {
namespace api {

constexpr int BG = 2;   // level index of the background-point batch
constexpr int TG = 3;   // level index of the Jacobian tangent pass (3 x the coarse tiles)
// dynamic tile counters (ints at ws + plan.counters)
enum { CT_WARP_FWD = 0, CT_MLP_FWD = 2, CT_TAN_FWD = 4, CT_MLP_BWD = 5, CT_WARP_BWD = 7, CT_TAN_BWD = 9, CT_BG_FWD = 10, CT_BG_BWD = 11 };

extern thread_local char g_err[256];   // nrf_last_error (defined in nrf_api.hip)

inline int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
inline int fail_hip(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), where);
  return NRF_E_HIP;
}

#define CK(call)                      \
  do {                                \
    int rc_ = (call);                 \
    if (rc_ != NRF_OK) return rc_;    \
  } while (0)

constexpr size_t ALIGN_F = 64;   // workspace sub-buffers are aligned to 64 floats (256 B)
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

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

}  // namespace api
}  // namespace nrf

struct nrf_handle_s {
  nrf::api::Prof prof;
  nrf_model_desc d;
  std::vector<nrf_tensor_info> layout;    // INTERNAL leaves (kernel widths); == xlayout unless `embed`
  int64_t nparams = 0;
  std::vector<nrf_tensor_info> xlayout;   // the caller's leaves (nrf_param_layout)
  int64_t xnparams = 0;
  std::vector<nrf::EmbedDesc> emb;             // external <-> internal element map, one per leaf
  bool embed = false;                     // trunk / rgb branch narrower than the kernels: run on a zero-padded image
  nrf::WarpParamOffsets xwpo;                  // warp leaves at their EXTERNAL offsets (nrf_warp_points reads the caller's buffer)
  nrf::MlpParamOffsets po[2];
  nrf::PackOffsets pk;
  int64_t app_off = -1, cam_off = -1;
  int P, PK, R, V, app_in_cond, nlevels;
  int A = 0;   // width of the alpha condition (use_appearance_metadata && use_alpha_condition: the appearance code)
  bool warp = false;
  nrf::WarpParamOffsets wpo;
  nrf::WarpPackOffsets wpk;
  int Fw = 0, G = 0, Win = 0, PKw = 0;
  bool time_enc = false;   // warp_metadata_encoder_type 'time': the codes come from modules.TimeEncoder instead of a GLO table
  int Ft = 0, Tin = 0;
  nrf::TimeParamOffsets tpo;
  int num_cus = 256;
  bool cu_queried = false;
  int chain_rows_opt = 0;   // NRF_OPT_CHAIN_TILE_ROWS: 0 automatic, 32, 64
  int bf16_wgrad_merge = 1; // NRF_OPT_BF16_WGRAD_MERGE: 1 (default) = skip-layer / bottleneck+alpha groups of the bf16 wgrad merged (operands
                            // streamed once: -10 % HBM fetch, +1..2 % step rate in the same-box A/B, profiles/r05_wgrad_bf16_merge_ab.md)
  nrf::api::WsPlan plan;
  // identity of the tables last uploaded to a workspace, and of the last stashed forward
  void* uploaded_ws = nullptr;
  int xdepth = nrf::TRUNK_DEPTH, xskip = nrf::SKIP_LAYER;   // the caller's trunk (<= 8 layers; its skip index or -1): nrf_create
  int emap[nrf::TRUNK_DEPTH];                          // internal trunk layer -> the caller's layer, or -1 (identity layer)
  int wxdepth = nrf::WARP_DEPTH, wxwidth = nrf::WARP_W;     // the caller's warp trunk (warp_kwargs trunk_depth / trunk_width)
  int uploaded_B = -1;
  uint32_t uploaded_flags = 0;
  int uploaded_bgN = 0;
  int uploaded_elastic = 0;
  void* stashed_ws = nullptr;
  uint64_t stashed_plan = 0;   // WsPlan::serial of the stashed forward
  int stashed_B = -1;
  bool stashed_warp = false;
  std::vector<nrf::PackDesc> wp_pack;   // pack table of nrf_warp_points (kept alive for the async upload)
  int64_t wp_pack_base = -1;
};

namespace nrf {
namespace api {

// nrf_plan.hip: parameter layout, pack tables, the workspace plan of (model, num_rays, flags), its descriptor tables
void build_layout(nrf_handle h);
void build_pack_offsets(nrf_handle h);
int k_old_for(int ntiles, int grid, int num_cus, double dflt_share);
int warp_grid_mul();
int* tile_counter_or_null(float* base, int idx);
uint32_t plan_flags(uint32_t flags);
bool chain32_for(const nrf_handle_s* h, int ntiles, bool reverse = false);
void build_plan(nrf_handle h, int B, uint32_t flags, int bgN = 0, int elastic = 0);
int upload_tables(nrf_handle h, float* ws, hipStream_t stream);
void query_device(nrf_handle h);

// nrf_run.hip: the launch sequences of NerfModel.apply (forward_impl) and of the gradient half of train_step (backward_impl)
int check_launch(const char* where);
int forward_impl(nrf_handle h, const float* params_x, const nrf_rays* rays, const nrf_step_scalars* scalars, const nrf_rand* rnd,
                 const nrf_outputs* out, uint32_t flags, float* ws, size_t ws_bytes, hipStream_t stream, int bgN = 0,
                 int elastic = 0, const nrf_background* bg = nullptr);
int backward_impl(nrf_handle h, const float* params_x, const nrf_rays* rays, const float* const d_rgb[2], const float* target,
                  float* grad_x, float* stats, float* ws, hipStream_t stream, const nrf_background* bg = nullptr,
                  const nrf_step_scalars* scalars = nullptr, const nrf_elastic* el = nullptr, const nrf_warp_reg* wr = nullptr,
                  bool bg_forward_done = false);

}  // namespace api
}  // namespace nrf
