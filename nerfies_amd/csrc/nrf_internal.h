// Internal declarations shared by the HIP kernels and the C-ABI layer.
// gfx950 (MI355X) only: wave = 64 lanes, MFMA f32 32x32x2, 160 KiB LDS / CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nerfies_amd.h"

namespace nrf {

constexpr int TILE_ROWS = 64;    // rows (ray samples) per workgroup tile; two workgroups per CU
constexpr int TRUNK_W = 256;     // NeRF trunk width the MFMA chain is built for
constexpr int RGB_W = 128;       // rgb branch width
constexpr int TRUNK_DEPTH = 8;
constexpr int SKIP_LAYER = 4;
constexpr int ACT_FLOATS = TRUNK_W * TILE_ROWS;        // LDS activation tile
constexpr int FRAG_TILE_256 = TRUNK_W * TILE_ROWS;     // floats per stash tile, 256 wide
constexpr int FRAG_TILE_128 = RGB_W * TILE_ROWS;
constexpr int SMALL_PART = 3080;  // per-workgroup small-gradient partials (see mlp_chain.hip)

// SE3 warp field trunk (warping.py:224-231 defaults): 6 x 128, skip at 4
constexpr int WARP_W = 128;
constexpr int WARP_DEPTH = 6;
constexpr int WARP_SKIP = 4;
constexpr int WACT_FLOATS = WARP_W * TILE_ROWS;
constexpr int WARP_SMALL_PART = 784;   // db_trunk[6][128] | db_w[3] | db_v[3] | pad
constexpr int WARP_MAX_IN = 64;        // padded trunk input width (3 + 6 F_w + G <= 64)
// workgroups of the SE3 chain kernels resident per CU (= waves per SIMD the register allocation leaves room for): 48 KiB of
// LDS each, so three fit once the kernels stay within 168 VGPRs
#ifndef NRF_WARP_WAVES
#define NRF_WARP_WAVES 3
#endif

// Offsets (in floats) of one NeRF MLP's leaves inside the flat parameter buffer
// (canonical flax layout: kernel [in,out] row-major, then bias).
struct MlpParamOffsets {
  int64_t trunk_k[TRUNK_DEPTH];
  int64_t trunk_b[TRUNK_DEPTH];
  int64_t bn_k, bn_b;
  int64_t rgbh_k, rgbh_b;      // [256+R,128], [128]
  int64_t logit_k, logit_b;    // [128,3], [3]
  int64_t alpha_k, alpha_b;    // [256,1], [1]
};

// Offsets (floats) inside one MLP's packed-weight block (see pack kernel).
struct PackOffsets {
  int fwd_L[TRUNK_DEPTH];  // L0: K=PK ; others K=256
  int fwd_L4b;             // skip layer's posenc rows, K=PK
  int fwd_bn, fwd_rgbh;
  int bwd_rgbhT, bwd_bnT;
  int bwd_LT[TRUNK_DEPTH]; // [1..7] used (dX of layer l), [0] unused unless warp
  int bwd_L0T, bwd_L4bT;   // warp on: W0^T and the skip layer's posenc rows^T, 256 -> PK (64-column stream)
  int total;
};

// SE3 field leaves inside the flat parameter buffer
struct WarpParamOffsets {
  int64_t trunk_k[WARP_DEPTH];
  int64_t trunk_b[WARP_DEPTH];
  int64_t w_k, w_b, v_k, v_b;   // branches_w / branches_v logit: [128,3], [3]
  int64_t embed;                // metadata_encoder embedding [num_embeddings, G]
};

struct WarpPackOffsets {
  int fwd_L[WARP_DEPTH];   // L0: K=PKw ; others K=128 (one 32-column block per wave)
  int fwd_L4b;             // skip layer's input rows, K=PKw
  int bwd_LT[WARP_DEPTH];  // [1..5]: transposed 128x128
  int total;
};

// modules.TimeEncoder (modules.py:297-322): depth 6, width 64, skip at 4 -- the warp field's 'time' metadata encoder
constexpr int TIME_W = 64;
constexpr int TIME_DEPTH = 6;
constexpr int TIME_SKIP = 4;
constexpr int TIME_MAX_IN = 20;   // 1 + 2 * num_freqs <= 17, padded
struct TimeParamOffsets {
  int64_t k[TIME_DEPTH], b[TIME_DEPTH];   // hidden_i kernel [in,64] / bias
  int64_t lk, lb;                         // logit kernel [64,G] / bias [G]
};
struct TimeEncArgs {
  const float* params;
  TimeParamOffsets po;
  const float* time;       // [B] time stamps (metadata['time'])
  int B, F, Tin, G;        // rays, posenc freqs, 1 + 2F, code width
  float alpha;             // warp_extra['time_alpha']
  const nrf_dynamic_scalars* dyn;   // device-resident step scalars (overrides alpha) or nullptr
  float* codes;            // [B][G] out
  const float* d_codes;    // [B][G] (backward)
  float* st_in;            // [B][TIME_MAX_IN] encoder input (training stash) or nullptr
  float* st_h;             // [B][6][64] post-ReLU activations or nullptr
  float* st_dpre;          // [B][6][64] (backward)
};
void launch_time_encoder_fwd(const TimeEncArgs& a, hipStream_t stream);
void launch_time_encoder_bwd(const TimeEncArgs& a, hipStream_t stream);
void launch_time_encoder_wgrad(const TimeEncArgs& a, float* grad, hipStream_t stream);

// bf16 chains (mlp_bf16.hip): one descriptor fills rows of a weight stream
// stream lengths in KiB (mlp_bf16.hip: chunk tables FW_* / DG*; nrf_plan.hip build_plan emits the chunks)
constexpr int BF_FWD_STREAM_KB = 4 * 10 + 28 * 34 + 4 * 42 + 17 + 2 * 32 + 9;
constexpr int BF_BWD_STREAM_KB = 8 + 4 * 18 + 4 * 34 + 28 * 32;
constexpr int BF_BWD_STREAM_DPTS_KB = BF_BWD_STREAM_KB + 2 * 32;
// split-bf16 inference chain (mlp_bf16x3.hip): every k-step row twice (W_hi, W_lo)
constexpr int BF_X3_STREAM_KB = 4 * 18 + 28 * 66 + 4 * 82 + 33 + 4 * 32 + 17;
struct RcPackDesc {
  long long src_off, dst_off;   // params leaf / first row written, in floats from the workspace base
  int kind;                     // 0: k-step rows of weights, 1: the bias row, 2: a zero row
  int src_ld, row0, krows, ncols;   // leaf column count, first row, valid K and valid M (output index of the GEMM)
  int ngroups, nout, nout_panel, o0;   // rows written, output blocks written, blocks per row of the chunk, first block position in the row
  int transposed;               // 0: A[m][k] = leaf[row0 + k][m] (forward); 1: A[m][k] = leaf[row0 + m][k] (dgrad: W as is)
  int oblk0;                    // first output block of the GEMM this descriptor covers (column 32 (oblk0 + o) + m)
  // two leaves side by side (the SE3 heads w | v, [128, 3] each): split > 0 -> indices >= split of the OUTPUT columns (forward,
  // bias) or of K (transposed) read leaf src_off2 at index - split
  long long src_off2;
  int split;
  // split-bf16 stream (mlp_bf16x3.hip): kind 0 writes TWO rows per k-step -- bf16(W), bf16(W - bf16(W)) -- and
  // `ngroups` counts those rows; kind 1 writes the bias as the triple (hi, lo, lo2) in k-slots 0..2
  int x3;
};
void launch_bf16_pack(const RcPackDesc* descs, int ndesc, const float* params, float* ws, hipStream_t stream);

// ---- bf16 training stash (mlp_bf16.hip writes it, wgrad_bf16.hip reads it) ----
// Unit = one GROUP of 32 samples (one wave of the bf16 chain kernels) x one BLOCK of 32 features = 2 KiB:
//   [jp (2)][lane = n + 32 h (64)][8 bf16],  the 8 = features 32 b + 8 (2 jp + jj) + 4 h + i  in (jj, i) order
// -- the packed B operand of k-step (b, jp) of the transposed chain, stored as it lies in the registers (1 KiB coalesced
// per wave store).  A buffer of NB blocks is [group][b][jp][lane].
constexpr int BF_GROUP = 32;
constexpr int BF_BLOCK_DW = 512;   // dwords per (group, block)
struct BfStash {
  uint32_t* pe;      // [ngroups][2]   posenc (layer-0 / skip input)
  uint32_t* h;       // [8][ngroups][8]  h1..h8 (post-ReLU)
  uint32_t* bn;      // [ngroups][8]   bottleneck output (linear)
  uint32_t* rgbh;    // [ngroups][4]   rgb hidden (post-ReLU)
  uint32_t* bits;    // [9][ngroups][64 lanes][4 dwords]: ReLU derivative bits of trunk layers 0..7 and (index 8, two dwords
                     // used) the rgb hidden layer; dword w = blocks (2w, 2w + 1); accumulator registers (2q, 2q + 1) of block o
                     // are bit 15 - (8 (o & 1) + q) of the low / high half (bf16_chain.h bits_push); 1 = pre-activation > 0
  uint32_t* dy;      // [8][ngroups][8]  dpre_0..dpre_7
  uint32_t* dbn;     // [ngroups][8]
  uint32_t* drgbh;   // [ngroups][4]
  uint32_t* dsmall;  // [ngroups][2]   block 0 features 0..3 = d raw (r, g, b, sigma), everything else 0
  int ngroups;       // whole workgroup iterations: 8 * ceil(rows / 256)
};

// ---- bf16 SE3 trunk (warp_bf16.hip; NRF_FLAG_BF16 with the warp field) ----
// Weight streams in KiB (chunk tables WF_* / WB_* in warp_bf16.hip; build_plan emits the chunks):
//   forward:  L0 2 x 10, L1..L3 + L5 2 x 18 each, L4 (skip) 2 x 26, heads 9
//   reverse:  heads^T 8, L5..L1 2 x 16 each, code-gradient GEMMs C0 / C4 16 each (the tangent reverse stops before them)
constexpr int BFW_FWD_STREAM_KB = 2 * 10 + 8 * 18 + 2 * 26 + 9;
constexpr int BFW_X3_STREAM_KB = 2 * 18 + 8 * 34 + 2 * 50 + 17;   // split-bf16 inference trunk (warp_bf16x3.hip): rows (W_hi, W_lo)
constexpr int BFW_BWD_TAN_STREAM_KB = 8 + 10 * 16;
constexpr int BFW_BWD_STREAM_KB = BFW_BWD_TAN_STREAM_KB + 2 * 16;
// Stash of one pass over one level, units as BfStash (32-row group x 32-feature block = 2 KiB)
struct BfWarpStash {
  uint32_t* win;     // [ngroups][2]       trunk input: annealed posenc (or its tangent), code
  uint32_t* h;       // [6][ngroups][4]    h1..h6, post-ReLU (tangent pass: masked tangents)
  uint32_t* bits;    // [6][ngroups][64 lanes][2 dwords]: ReLU derivative bits (bf16_chain.h bits_push), dword = panel
  uint32_t* dy;      // [6][ngroups][4]    dpre_0..5
  uint32_t* dhead;   // [ngroups][2]       block 0 features 0..5 = (dL/dw, dL/dv), everything else 0
  int ngroups;       // 8 * ceil(rows / 256)
};

struct ChainBwdBf16Args {
  const float* wpk;          // dgrad weight stream (W as the A operand, K = the layer's output features)
  const float4* d_raw4;      // [rows_pad] dL/d(raw rgb, raw sigma); 0 on pad rows
  int S, B, rows;
  BfStash st;
  // warp on: gradient w.r.t. the (warped) sample points through both posenc inputs of the trunk (float32, for the warp kernels)
  const float* points;       // [rows][3] the points the forward encoded, or nullptr
  float* d_points;           // [rows_pad][3]
  int rows_pad, F, P;
};
void launch_chain_bwd_bf16(const ChainBwdBf16Args& a0, const ChainBwdBf16Args* a1, int max_grid, hipStream_t stream);   // a1: second level or nullptr
// per-ray sums of dpre_rgbh from its bf16 stash -> dray [B][128] (gradient of the rgb-condition columns)
void launch_dray_bf16(const uint32_t* drgbh, int B, int S, float* dray, hipStream_t stream);

struct ChainFwdArgs {
  const float* params;       // flat canonical parameters
  MlpParamOffsets po;
  const float* wpk;          // packed weights of this MLP
  PackOffsets pk;
  const float* condterm;     // [B][128]: rgb-branch per-ray term incl. bias
  const float* zvals;        // [B*S]
  const float* origins;      // [B][3]
  const float* directions;   // [B][3]
  const float* points;       // [rows][3] warped points, or nullptr -> o + z d
  float4* out4;              // [ntiles*128] (r,g,b,sigma) post-activation
  int S, B, rows, ntiles;
  int F, P, PK;              // point freqs, 3+6F, P rounded up to a multiple of 16
  int skip;                  // trunk layer that reads [h, posenc] (modules.py:47-48): 4 unless the model asks for another (float32 chains)
  int sigma_act;
  int* tile_counter;         // zeroed before the launch: dynamic tile hand-out (chain_common.h next_tile)
  int k_old;                 // > 0: uneven static split, tiles of the older workgroup of a CU (chain_common.h tile_iter)
  unsigned long long* timeline;   // debug: [4 waves][64] shader-clock stamps of workgroup 0 (or nullptr)
  // use_alpha_condition (modules.py:152-157): the alpha head reads [bottleneck, appearance code]; alpha_ct[ray] =
  // code . W_alpha[256:] (ray_prep), nullptr -> the head reads the trunk output
  const float* alpha_ct;
  // model_utils.noise_regularize (model_utils.py:266-282): raw density += noise_std * N(0,1); explicit normals [rows] or
  // Philox stream `noise_stream` of (seed, offset)
  const float* noise;
  float noise_std;
  unsigned long long noise_seed, noise_offset;
  unsigned noise_stream;
  const nrf_dynamic_scalars* dyn;   // device-resident step scalars (overrides the noise seed / offset) or nullptr
  // activation stash (training only)
  float* st_pe;              // [ntiles][PK][128]
  float* st_h;               // [8][ntiles][256*128]  h1..h8, fragment-native
  float* st_bn;              // [ntiles][256*128]
  float* st_rgbh;            // [ntiles][128*128]
  // ReLU sign bits for the dgrad pass, one bit per accumulator element, fragment-native
  uint32_t* bits_trunk;      // [8][ntiles][4 waves][64 lanes] x 4 dwords
  uint32_t* bits_rgbh;       // [ntiles][4 waves][64 lanes] x 2 dwords
  BfStash bst;               // bf16 training chain (NRF_FLAG_BF16 | NRF_FLAG_TRAIN); pointers null otherwise
};

struct ChainBwdArgs {
  const float* params;
  MlpParamOffsets po;
  const float* wpk;
  PackOffsets pk;
  const float4* d_raw4;      // [ntiles*128] dL/d(raw rgb, raw sigma); 0 on pad rows
  int S, B, rows, ntiles;
  const uint32_t* bits_trunk;
  const uint32_t* bits_rgbh;
  float* dy_trunk;           // [8][ntiles][256*128]  dpre_0..dpre_7
  float* dy_bn;              // [ntiles][256*128]
  float* dy_rgbh;            // [ntiles][128*128]
  float* dray;               // [B][128] += per-ray sums of dpre_rgbh (atomics)
  float* small_part;         // [gridDim.x][SMALL_PART]
  // warp on: gradient w.r.t. the (warped) sample points through both posenc inputs of the trunk
  float* d_points;           // [ntiles*128][3] or nullptr
  const float* st_pe;        // posenc stash of the forward pass
  int F, P, PK;
  int skip;                  // as ChainFwdArgs
  int* tile_counter;
  int k_old;                 // as ChainFwdArgs
  int alpha_on_bn;           // use_alpha_condition: d raw sigma enters at the bottleneck instead of the trunk output
};

// SE3Field forward (warping.py:322-353): x = o + z d (or explicit points) -> warped points.
struct WarpFwdArgs {
  const float* params;
  WarpParamOffsets po;
  const float* wpk;
  WarpPackOffsets pk;
  const float* zvals;        // [B*S]
  const float* origins;      // [B][3]
  const float* directions;   // [B][3]
  const int32_t* warp_ids;   // [B], or nullptr: row `ray` of the table (pre-encoded per-ray codes)
  const float* embed_table;  // [*][G] GLO table inside the parameters, or the per-ray codes (metadata_encoded / TimeEncoder)
  const float* points_in;    // [rows][3] explicit points (then ids are per point) or nullptr
  const int32_t* point_ids;  // [rows] with points_in
  float* points_out;         // [ntiles*128][3] warped points
  float* points_raw;         // optional [rows][3]: the unwarped sample points (return_points)
  int S, B, rows, ntiles;
  int F, G, Win, PKw;        // warp freqs, code width, 3+6F+G, Win rounded up to a multiple of 8
  float alpha;               // warp_extra['alpha']
  const nrf_dynamic_scalars* dyn;   // device-resident step scalars (overrides alpha) or nullptr
  float* st_win;             // [ntiles][PKw][128] trunk input (training stash)
  float* st_h;               // [6][ntiles][128*128] h1..h6, fragment-native
  float4* st_wv;             // [ntiles*128][2] raw head outputs (w, v)
  uint32_t* bits;            // [6][ntiles][4 waves][64 lanes] sign bits of the trunk pre-activations
  // tangent pass (prim_win != nullptr): tile tt = c * nt_prim + t, inputs / masks of primal tile t
  int nt_prim;
  const float* prim_win;
  const uint32_t* prim_bits;
  int* tile_counter;         // zeroed before the launch
  // bf16 trunk (warp_bf16.hip): this pass's stash (null pointers: inference), the stream, the primal pass's bits (tangent)
  const float* bwpk;
  BfWarpStash bst;
  const uint32_t* bprim_bits;
  int bng_prim;              // groups of the primal level (tangent pass: group = c * bng_prim + primal group)
  int rows_pad;              // rows of the per-row fp32 outputs (points_out, st_wv): ntiles * 64; tangent: per coordinate
};

struct WarpBwdArgs {
  const float* params;
  WarpParamOffsets po;
  const float* wpk;
  WarpPackOffsets pk;
  const float* d_points;     // [ntiles*128][3] dL/d warped point
  const float* st_win;
  const float4* st_wv;
  const uint32_t* bits;
  const int32_t* warp_ids;   // [B], or nullptr: row `ray` (per-ray codes of the TimeEncoder)
  const int32_t* point_ids;  // [rows] or nullptr
  int S, B, rows, ntiles;
  int F, G, Win, PKw;
  float* dy;                 // [6][ntiles][128*128] dpre_0..dpre_5
  float4* d_w4;              // [ntiles*128] (dw, 0)
  float4* d_v4;              // [ntiles*128] (dv, 0)
  float* grad_embed;         // embedding-table gradient, or the per-ray code gradient [B][G] (atomics)
  float* small_part;         // [gridDim.x][WARP_SMALL_PART]
  const float4* extra_dw4;   // primal pass: + dL/d(w, v) of the elastic regulariser, or nullptr
  const float4* extra_dv4;
  int* tile_counter;         // zeroed before the launch
  int tangent;               // reverse of the tangent pass: d_w4 / d_v4 are INPUTS, masks of primal tile tt % nt_prim
  int nt_prim;               // tiles of the primal level (== ntiles for the primal pass)
  // bf16 trunk (warp_bf16.hip)
  const float* bwpk;         // reverse stream
  BfWarpStash bst;           // dy / dhead written, bits read (tangent: the PRIMAL level's bits in bprim_bits)
  const uint32_t* bprim_bits;
  int bng_prim, rows_pad;
  const float* x_rows;       // [rows][3] the points the field was evaluated at (fp32)
};

// training.compute_elastic_loss on the coarse samples (see warp_chain.hip elastic_kernel)
struct ElasticArgs {
  const float* x_rows;       // [rows][3] sample points (bf16 trunk: there is no fp32 input stash), or nullptr -> prim_win
  const float* prim_win;     // primal trunk-input stash (x = features 0..2)
  const float4* prim_wv;     // primal raw head outputs (w, v) per row
  const float4* tan_wv;      // [3][rows_pad] x 2: (dw/dx_c, dv/dx_c)
  const float* coef;         // [rows] stop-gradient sample weights (elastic_reduce_method 'weight')
  float4* tan_dw4;           // out [3][rows_pad]: dL/d(dw/dx_c)
  float4* tan_dv4;
  float4* prim_dw4;          // out [rows_pad]: dL/dw, dL/dv through exp_se3's second derivatives
  float4* prim_dv4;
  float* part;               // out [5][workgroups]: per-workgroup sums of coef*rho, residual, det J, div J, |curl J| (no atomics)
  int rows, rows_pad, PKS;
  float eps, alpha, scale;
  float gscale;              // elastic_loss_weight / num_rays
  float inv_rays;            // 1 / num_rays
  const nrf_dynamic_scalars* dyn;   // device-resident step scalars: gscale = dyn->elastic_loss_weight * inv_rays
  int res_selected;          // 'median': the residual statistic only counts the selected sample of each ray
  int loss_type;             // NRF_ELASTIC_LOG_SVALS ...
};

// warp Jacobian as an output (return_warp_jacobian, models.py:264-265): J = I + d/dx [exp_se3(w, v) x - x]
struct JacobianArgs {
  const float* x_rows;       // as ElasticArgs
  const float* prim_win;
  const float4* prim_wv;
  const float4* tan_wv;
  float* out;                // [rows][3][3]
  int rows, rows_pad, PKS;
};

// One split-K slice of a weight-gradient GEMM  dW[k][n] = sum_rows X[row][k] dY[row][n].
enum { SRC_FRAG256 = 0, SRC_FRAG128 = 1, SRC_PLAIN = 2 };
struct WgradTask {
  const float* X;  int x_kind;  int x_tile_stride;  int x_kvalid;  int Kb;
  const float* dY; int dy_kind; int dy_tile_stride; int Nb;   // Nb == 0: vector columns only
  int tile_begin, tile_end;
  float* slab;     // [Kb*32][Nb*32]
  // optional narrow dY columns done on the VALU: vec[row] = float4 (d raw rgb, d raw sigma);
  // vslab[2][Kb*32][4] (two row halves) accumulates X^T vec.  vec2 / vslab2: a second vector against the same X
  // (SE3 heads: dL/dw and dL/dv both multiply h6) or nullptr.
  const float4* vec; float* vslab;
  const float4* vec2; float* vslab2;
};

// A layer's wgrad GEMM, cut into `nsplit` tasks of `tiles_per` tiles.  Offsets are in floats
// from the workspace base, so the table depends only on (model, num_rays).
struct WgradGroup {
  int64_t x_off, dy_off, slab_off;
  int64_t vec_off, vslab_off;     // vec_off < 0: no vector columns
  int64_t vec2_off, vslab2_off;   // vec2_off < 0: no second vector
  int x_kind, x_tile_stride, x_kvalid, Kb;
  int dy_kind, dy_tile_stride, Nb;
  int ntiles, nsplit, tiles_per, first_task;
  // bf16 kernel only: an operand assembled from TWO stash buffers, so that a buffer two weight matrices share is streamed once
  // (X = [h4 | posenc] against dpre_4; dY = [d bottleneck | d raw] against h8).  Blocks [0, Kb1) of X come from x_off, blocks
  // [Kb1, Kb) from x2_off (x2_tile_stride dwords per group); likewise Nb1 / dy2_off.  Kb1 == Kb, Nb1 == Nb: one source each.
  int64_t x2_off, dy2_off;
  int x2_tile_stride, dy2_tile_stride, Kb1, Nb1;
};

// Stream-K style partition of the wgrad work: workgroup w runs segments [seg_begin[w], seg_begin[w+1]).
struct WgradSegment { int group, tile_begin, tile_end, slab_idx; };

struct ReduceDesc {
  int64_t dst_off;    // floats from the flat gradient buffer
  int64_t src_off;    // floats from the workspace base
  int64_t part_stride;
  int dst_ld, rows, cols, src_ld, nparts;
  int accumulate;     // > 0: dst += (pass index: a later level adding into leaves shared with earlier passes)
  int next;           // the descriptor of the NEXT pass into the same destination, run by the same threads right behind this one
                      // (-1: none): every pass of a leaf in ONE launch, in pass order (round 6; rounds 1-5: one launch per pass)
  int path;           // element -> thread mapping, the same along a chain: 0 scalar, 1 four columns per thread, 2 one wave per column
};

struct PackDesc {
  int64_t src_off;    // canonical kernel [*, src_ld] row-major, floats from the flat params
  int64_t dst_off;    // floats from the workspace base
  int src_ld;
  int src_row0;       // first source row (forward) / unused (transposed)
  int kvalid;         // number of valid k
  int K;              // padded K (multiple of 4, or 8 when ncb==1)
  int ncb;            // 2 -> 64 columns per wave, 1 -> 32 columns per wave
  int transposed;     // B[k][n] = src[src_row0 + n][k]
  int nwaves;         // column groups (4 -> N = 256 / 128; 1 -> N = 64 / 32)
  int nvalid;         // valid n (columns beyond are zero)
};

// Environment knobs, read ONCE per process (first use), never on a launch path.  The product build honours only the debugging
// aid NRF_TRACE_REGIONS; the experiment knobs (tile hand-out, grid multipliers, timelines, occupancy print)
// exist only in builds compiled with -DNRF_EXPERIMENT (scripts/build_variant.py NAME -DNRF_EXPERIMENT ...).
struct Knobs {
  bool trace_regions = false;   // NRF_TRACE_REGIONS: name every kernel group on stderr and synchronise behind it
  bool debug_occ = false;       // NRF_DEBUG_OCC: print the forward kernel's occupancy once per launch
  bool timeline = false;        // NRF_TIMELINE: shader-clock stamps of workgroup 0 (needs -DNRF_TIMELINE_BUILD as well)
  bool dynamic_tiles = false;   // NRF_DYNAMIC_TILES: tiles from a global counter instead of the static split
  int grid_mul = 2;             // NRF_GRID_MUL: workgroups per CU of the fp32 NeRF chain launches
  int warp_grid_mul = NRF_WARP_WAVES;   // NRF_WARP_GRID_MUL: workgroups per CU of the SE3 chain launches
  double old_share = -1.0;      // NRF_OLD_SHARE: uneven static split (chain_common.h tile_iter); < 0: the caller's default
};
const Knobs& knobs();

// ---- launchers (all asynchronous on `stream`) ----
void launch_pack(const PackDesc* d_descs, int ndesc, const float* params, float* ws, hipStream_t stream);
void launch_chain_fwd(const ChainFwdArgs& a, bool stash, int grid, hipStream_t stream);
// a1 (optional): the fine level, run by the same launch (tiles [a0.ntiles, a0.ntiles + a1->ntiles))
void launch_chain_bwd(const ChainBwdArgs& a0, const ChainBwdArgs* a1, int grid, hipStream_t stream);
// the same chains on 32-row tiles, four workgroups per CU (mlp_chain32.hip); same arguments, same HBM images.  The reverse
// kernel has no d-points path yet: warp-on plans keep the 64-row reverse pass
void launch_chain_fwd32(const ChainFwdArgs& a, bool stash, int grid, hipStream_t stream);
void launch_chain_bwd32(const ChainBwdArgs& a0, const ChainBwdArgs* a1, int grid, hipStream_t stream);
// a1 (optional): a second level in the same launch (background points behind the coarse samples)
void launch_warp_fwd(const WarpFwdArgs& a, const WarpFwdArgs* a1, bool stash, int grid, hipStream_t stream);
// a1, a2 (optional): further levels in the same launch; bias partials of all levels go to a.small_part
void launch_warp_bwd(const WarpBwdArgs& a, const WarpBwdArgs* a1, const WarpBwdArgs* a2, int grid, hipStream_t stream);
void launch_elastic(const ElasticArgs& a, hipStream_t stream);
void launch_jacobian(const JacobianArgs& a, hipStream_t stream);
void launch_median_coef(const float* weights, int B, int S, float* coef, hipStream_t stream);
void launch_wgrad(const WgradGroup* d_groups, const WgradSegment* d_segs, const int* d_seg_begin, int nwg, float* ws,
                  unsigned long long* seg_clock, hipStream_t stream);
void launch_reduce(const ReduceDesc* d_table, int first, int ndesc, const float* ws, float* grad, hipStream_t stream);   // chain heads table[first .. first + ndesc)
// bf16 wgrad (wgrad_bf16.hip): the same group / segment tables, "tile" = one 32-sample group of the bf16 stash, Kb / Nb
// blocks per group for X / dY (x_tile_stride = Kb * 512, dy_tile_stride = Nb * 512 dwords); vslab_off >= 0: the
// group also sums dY over the rows (bias gradient) into vslab[slab_idx][Nb * 32]
void launch_wgrad_bf16(const WgradGroup* d_groups, const WgradSegment* d_segs, const int* d_seg_begin, int nwg, float* ws,
                       hipStream_t stream);

struct RayPrepArgs {
  const float* params;
  const float* viewdirs;
  const int32_t* app_ids;    // [B] or nullptr (then app_codes)
  const int32_t* cam_ids;
  const float* app_codes;    // [B][app_feat] pre-encoded (metadata_encoded) or nullptr
  const float* cam_codes;
  int B, Fv, use_viewdirs, app_feat, cam_feat, R;
  int64_t app_off, cam_off;
  int64_t rgbh_k[2], rgbh_b[2];   // per level (coarse, fine)
  int64_t alpha_k[2];             // use_alpha_condition: rows 256.. of MLP_2/logit/kernel hold the appearance-code weights
  float* cond;               // [B][R]
  float* condterm[2];        // [B][128]; [1] may be nullptr
  float* alpha_ct[2];        // [B] or nullptr
};
void launch_ray_prep(const RayPrepArgs& a, hipStream_t stream);
void launch_sample_coarse(const float* t_rand, int B, int N, float near_p, float far_p,
                          int stratified, int lindisp, uint64_t seed, uint64_t offset, const nrf_dynamic_scalars* dyn,
                          float* z, hipStream_t stream);
void launch_sample_points(const float* origins, const float* dirs, const float* z, int B, int S, float* out, hipStream_t stream);
void launch_composite_fwd(const float4* out4, const float* z, const float* dirs, int B, int S,
                          int white_bkgd, int sample_at_inf, float* rgb, float* depth,
                          float* med_depth, float* acc, float* weights, hipStream_t stream);
struct CompositeBwdArgs {
  const float4* out4; const float* z; const float* dirs;
  int B, S, white_bkgd, sample_at_inf, sigma_act;
  const float* rgb_out; const float* target; const float* d_rgb;   // target: MSE gradient; else d_rgb as given
  float loss_scale;
  float4* d_raw4; int rows_pad;
  float* mse_ray;      // [B] squared error per ray (or nullptr)
  float* dsig_ray;     // [B] sum of d sigma_raw over the ray (use_alpha_condition) or nullptr
};
struct CompositeBwdArgs2 { CompositeBwdArgs a[2]; };
void launch_composite_bwd(const CompositeBwdArgs& a0, const CompositeBwdArgs* a1, hipStream_t stream);   // a1: second level or nullptr
// use_alpha_condition: gradient of the appearance-code rows of the alpha head and of the codes through it
void launch_alpha_cond_grad(const float* params, const float* cond, const float* dsig_ray, const int32_t* app_ids, int B, int R,
                            int V, int app_feat, int64_t app_off, int64_t alpha_k, float* grad, hipStream_t stream);
// use_warp_reg_loss (training.py:199-212) of one level: adds d loss / d warped point into d_points, sums[0] += loss, sums[1] += residual
void launch_warp_reg(const float* weights, const float* points, const float* warped, int B, int S, float alpha, float scale,
                     float gscale, float* d_points, float* sums, hipStream_t stream);
void launch_sample_fine(const float* z_c, const float* w_c, int B, int Nc, int Nf, int stratified,
                        const float* u, uint64_t seed, uint64_t offset, const nrf_dynamic_scalars* dyn, float* z_out,
                        hipStream_t stream);
// training.compute_background_loss's draws (training.py:121-126) on the device: id = choices[floor(U n)], x += std N(0,1)
void launch_background_draw(const float* points, int N, const int32_t* choices, int nchoices, float noise_std, uint64_t seed,
                            uint64_t offset, const nrf_dynamic_scalars* dyn, float* out_points, int32_t* out_ids, hipStream_t stream);
void launch_cond_wgrad(const float* cond, const float* dray0, const float* dray1, int B, int R, float* dst0, float* dst1,
                       hipStream_t stream);   // dray1 / dst1: the second level (nullptr: one level)
void launch_cond_embed_grad(const float* params, const float* dray0, const float* dray1, const int32_t* app_ids,
                            const int32_t* cam_ids, int B, int V, int app_feat, int64_t app_off, int cam_feat, int64_t cam_off,
                            int64_t rgbh_k0, int64_t rgbh_k1, float* grad, hipStream_t stream);   // dray1: second level or nullptr
struct StatsArgs {
  const float* mse_ray; int B, nlevels;    // [nlevels][B] squared error per ray
  const float* bg_sum; int bgN; float bg_weight;
  const float* el_part; int el_nwg;   // elastic_kernel's per-workgroup partial sums [5][el_nwg]
  int el_rows, el_jac_rows; float el_weight;
  const float* wr_sums; float wr_weight;   // [4]: loss coarse, residual coarse, loss fine, residual fine
  float* stats;
  const nrf_dynamic_scalars* dyn;          // device-resident step scalars (overrides el_weight) or nullptr
};
void launch_finish_stats(const StatsArgs& a, hipStream_t stream);
void launch_background_loss(const float* points, const float* warped, int N, int rows_pad, float alpha, float scale,
                            float weight, float* d_points, float* loss_sum, hipStream_t stream);
// zero-fills up to 8 float ranges in one launch (16-byte aligned pointers)
struct ZeroArgs {
  float* p[12];
  long long n[12];
  int count;
  bool overflow;   // a thirteenth range was offered: the caller must fail (an accumulator would stay unzeroed)
  void add(float* ptr, long long nfloats) {
    if (nfloats <= 0) return;
    if (count >= 12) { overflow = true; return; }
    p[count] = ptr; n[count] = nfloats; ++count;
  }
};
void launch_zero_ranges(const ZeroArgs& a, hipStream_t stream);
void launch_adam(float* p, float* m, float* v, const float* g, int64_t n, double lr, double b1,
                 double b2, double eps, int64_t step, double gscale, hipStream_t stream);
void launch_dynamic_write(nrf_dynamic_scalars* dst, const nrf_dynamic_scalars& v, hipStream_t stream);
void launch_adam_dynamic(float* p, float* m, float* v, const float* g, int64_t n, double b1, double b2, double eps,
                         const nrf_dynamic_scalars* dyn, hipStream_t stream);

// Narrower models run on the 256-wide / 128-wide kernels by embedding: the caller's parameter leaves are copied into
// a zero-filled internal image with the kernels' widths (zero weights and biases for the extra units: relu(0) = 0 and
// zero outgoing weights make the padded network compute the same function), gradients are copied back out.
// element (r, c) of the external leaf <-> (r < split ? r : r + shift, c) of the internal leaf.  ext_off < 0: no external leaf --
// the internal one is the rows x rows identity (written on the way in, nothing on the way out).
struct EmbedDesc {
  long long ext_off, int_off;
  int rows, ext_cols, int_cols, split, shift, pad_;
};
void launch_chain_fwd_bf16(const struct ChainFwdArgs& a, int max_grid, hipStream_t stream);
// split-bf16 ("bf16x3", float32-emulating) inference chain: a.wpk = the x3 weight stream
void launch_chain_fwd_x3(const struct ChainFwdArgs& a, int max_grid, hipStream_t stream);
// bf16 SE3 trunk: forward of one or two levels (a1: e.g. the background batch) or the tangent pass (a.prim_... set);
// reverse of up to three levels, or of the tangent pass
void launch_warp_fwd_bf16(const WarpFwdArgs& a, const WarpFwdArgs* a1, bool stash, int max_grid, hipStream_t stream);
void launch_warp_bwd_bf16(const WarpBwdArgs& a, const WarpBwdArgs* a1, const WarpBwdArgs* a2, int max_grid, hipStream_t stream);
// split-bf16 (float32-emulating) SE3 trunk, inference forward of one level: a.bwpk = the x3 stream
void launch_warp_fwd_x3(const WarpFwdArgs& a, int max_grid, hipStream_t stream);
void launch_embed(const EmbedDesc* descs, int ndesc, const float* src, float* dst, bool to_internal, hipStream_t stream);

// camera.hip -- Camera.pixels_to_rays / pixels_to_points / project (nerfies/camera.py)
struct CameraArgs {
  float R[9], pos[3];
  float focal, cx, cy, skew, aspect;
  float k1, k2, k3, p1, p2;
  int width, height, distorted;
};
void launch_camera_rays(const CameraArgs& c, const float* pixels, const float* depth, long n, float* origins,
                        float* directions, float* pixels_out, hipStream_t stream);
void launch_camera_project(const CameraArgs& c, const float* points, long n, float* pixels, hipStream_t stream);

}  // namespace nrf
