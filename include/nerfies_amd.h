/* nerfies_amd.h -- C-ABI of the MI355X-native nerfies hot path.
 *
 * The reference (google/nerfies) is pure Python on JAX and has NO FFI / plugin
 * boundary: the seam this library replaces is the Python call
 *   NerfModel.apply            nerfies/models.py:289-375
 * together with its two callers
 *   training.train_step        nerfies/training.py:138-271
 *   evaluation.render_image    nerfies/evaluation.py:28-101
 * Each entry point below cites the reference function it stands in for.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host.
 *   - the caller owns every buffer, including the workspace; the library never
 *     allocates, frees or synchronises per call (hipGraph-capture safe).
 *   - all work is enqueued on the caller's hipStream_t (passed as void*).
 *   - every entry returns 0 on success or a negative NRF_E_* code; nothing
 *     throws or aborts.  nrf_last_error() returns the message of the last failing
 *     call MADE BY THE CALLING THREAD (thread-local storage; valid until that thread's
 *     next failing call).  Handles carry no error state, so the errors of several threads never mix.
 *   - a handle caches ONE workspace plan -- the layout for the (num_rays, flags, regularisers) of its most recent call -- and
 *     the descriptor tables it uploads into a workspace belong to that plan.  Calls on one handle must therefore be SERIALISED
 *     by the caller (one host thread at a time, launches of different (num_rays, flags) on different streams ordered by the
 *     caller); changing the batch size or the flags between calls is fine (the plan is rebuilt, the tables re-uploaded), and
 *     nrf_backward refuses a workspace whose forward was stashed under another plan (NRF_E_STATE).  For concurrent streams
 *     or threads create one handle per stream: nrf_create is host-only and cheap.
 *   - grad_params must be 16-byte aligned (it is cleared and accumulated into with 128-bit accesses; NRF_E_SHAPE otherwise).
 *   - fp32 everywhere (the reference computes in fp32); ids are int32.
 */
#ifndef NERFIES_AMD_H_
#define NERFIES_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRF_VERSION 600 /* 0.6.0: nrf_model_desc grows warp_trunk_depth / warp_trunk_width (SE3Field / TranslationField trunk_depth <= 6,
                           trunk_width <= 128 of ModelConfig.warp_kwargs); nerf_skip_layer accepts any single index 1..7 (float32
                           mode; the bfloat16 mode where the trunk can be laid out around the chains' layer 4); NRF_FLAG_BF16X3 (split-bf16,
                           float32-emulating inference chains).  Behaviour change
                           since 0.5.0: the flag word is validated on every entry point -- NRF_FLAG_TRAIN | NRF_FLAG_NO_WARP is refused
                           also for models without a warp field (it used to be a no-op there), NRF_FLAG_WARP_F32 without NRF_FLAG_BF16
                           is refused by nrf_train_step_loss_grad_ex as well.
                           0.5.0: nrf_set_option (NRF_OPT_CHAIN_TILE_ROWS: 32-row tiling of the fp32 chains).
                           0.4.0: NRF_FLAG_BF16 also runs the SE3 trunk in bfloat16 (NRF_FLAG_WARP_F32 opts out).
                           0.3.0: device-resident per-step scalars (a whole train step replays from one hipGraph), background ids /
                           noise drawn by the library, `points` output without the warp field.
                           0.2.0: alpha condition, pre-encoded metadata, warp Jacobian output, noise_std, warp_reg loss,
                           elastic loss types, time metadata encoder, stats[16], bf16 training */

enum {
  NRF_OK = 0,
  NRF_E_NULL = -1,        /* null pointer */
  NRF_E_SHAPE = -2,       /* bad size / shape */
  NRF_E_UNSUPPORTED = -3, /* configuration outside what the kernels cover */
  NRF_E_HIP = -4,         /* HIP runtime error (see nrf_last_error) */
  NRF_E_WORKSPACE = -5,   /* workspace too small */
  NRF_E_STATE = -6        /* call order (backward without a stashed forward) */
};

enum { NRF_ACT_RELU = 0, NRF_ACT_SOFTPLUS = 1 };
enum { NRF_WARP_SE3 = 0, NRF_WARP_TRANSLATION = 1 };   /* ModelConfig.warp_field_type (configs.py:99-100) */
enum { NRF_META_GLO = 0, NRF_META_TIME = 1 };           /* ModelConfig.warp_metadata_encoder_type (configs.py:101) */

/* Every NerfModel attribute that reaches the hot path (models.py:75-119), as
 * POD.  Defaults are those of configs.ModelConfig (configs.py:35-105). */
typedef struct nrf_model_desc {
  int32_t num_coarse_samples;      /* models.py:75  */
  int32_t num_fine_samples;        /* models.py:76  (0 = coarse only) */
  int32_t use_viewdirs;            /* models.py:77  */
  float near_plane;                /* models.py:78  */
  float far_plane;                 /* models.py:79  */
  int32_t nerf_trunk_depth;        /* 1..8; the kernels run 8 layers, a shallower trunk gets internal identity layers behind it */
  int32_t nerf_trunk_width;        /* <= 256; the kernels are 256 wide, narrower trunks run zero-padded (test_vrig.gin: 128) */
  int32_t nerf_rgb_branch_depth;   /* 1   */
  int32_t nerf_rgb_branch_width;   /* <= 128 (same) */
  int32_t nerf_skip_layer;         /* nerf_skips = (s,) -> s in 1..7 (modules.py:47-48: layer s reads [h, posenc]); -1 = none (or any index
                                      >= nerf_trunk_depth: never reached).  s <= 4 with nerf_trunk_depth - s <= 4 is laid out around the chains'
                                      own skip at layer 4 (identity layers in between: exact) and runs in every mode; any other s runs
                                      the float32 chains with the skip GEMM at layer s (NRF_FLAG_BF16 -> NRF_E_UNSUPPORTED) */
  int32_t use_stratified_sampling; /* models.py:88; eval.py:239 forces 0 */
  int32_t num_nerf_point_freqs;    /* models.py:89  */
  int32_t num_nerf_viewdir_freqs;  /* models.py:90  */
  int32_t sigma_activation;        /* NRF_ACT_*  (defaults.gin:66 softplus) */
  int32_t use_white_background;
  int32_t use_linear_disparity;
  int32_t use_sample_at_infinity;
  /* metadata / conditions (models.py:186-228) */
  int32_t use_appearance_metadata;
  int32_t num_appearance_embeddings; /* max(appearance_ids)+1, models.py:121 */
  int32_t num_appearance_features;
  int32_t use_camera_metadata;
  int32_t num_camera_embeddings;
  int32_t num_camera_features;
  int32_t use_alpha_condition; /* also gates appearance->rgb (models.py:206) */
  int32_t use_rgb_condition;   /* accepted, unused -- as in the reference */
  int32_t use_trunk_condition; /* never forwarded by construct_nerf */
  /* warp field (warping.py:202-389, SE3Field) */
  int32_t use_warp;
  int32_t num_warp_freqs;
  int32_t num_warp_embeddings;
  int32_t num_warp_features;
  int32_t warp_field_type;         /* NRF_WARP_SE3 (warping.py:202-389, every preset) or NRF_WARP_TRANSLATION
                                      (warping.py:62-199, the dataclass default): leaves warp_field/mlp/... */
  float noise_std;                 /* models.py:80, model_utils.noise_regularize (model_utils.py:266-282): N(0, noise_std) added to
                                      the raw density when > 0 and use_stratified_sampling; 0 = off (every preset) */
  int32_t warp_metadata_encoder_type; /* NRF_META_GLO (every preset) or NRF_META_TIME: modules.TimeEncoder on
                                      metadata['time'] (modules.py:297-322, warping.py:256-259, models.py:252-254) */
  int32_t num_time_encoder_freqs;  /* metadata_encoder_num_freqs (warping.py:234): 1 */
  /* ModelConfig.warp_kwargs (configs.py:105) that reach the trunk of SE3Field / TranslationField (warping.py:225-227, 80-82):
   * 0 = the field's default.  A shallower / narrower trunk runs on the 6 x 128 kernels with identity layers behind its last one and
   * zero padding (exact); `skips` stays (4,) -- a trunk of <= 4 layers never reaches it.  The other warp_kwargs (rotation /
   * pivot / translation branch depths, use_pivot, use_translation, activation, min / max_freq_log2) are not built: the Python
   * host refuses them unless they equal the field's defaults. */
  int32_t warp_trunk_depth;        /* 1..6, 0 -> 6   */
  int32_t warp_trunk_width;        /* 1..128, 0 -> 128 */
} nrf_model_desc;

typedef struct nrf_handle_s* nrf_handle;

/* One leaf of the parameter tree inside the single flat fp32 buffer.  `name`
 * is the flax path below params['model'], e.g.
 * "nerf_mlps_coarse/MLP_0/hidden_4/kernel" -- kernels are [in,out] row-major
 * exactly as flax nn.Dense stores them. */
typedef struct nrf_tensor_info {
  char name[96];
  int64_t offset; /* in floats from the start of the flat buffer */
  int32_t rows;   /* in  (embedding: num_embeddings) ; bias: 1 */
  int32_t cols;   /* out (embedding: features) */
} nrf_tensor_info;

/* rays_dict of NerfModel.__call__ (models.py:300-329). */
typedef struct nrf_rays {
  int32_t num_rays;
  const float* origins;          /* (B,3) */
  const float* directions;       /* (B,3) */
  const float* viewdirs;         /* (B,3) or NULL -> directions (models.py:326-329) */
  const int32_t* warp_ids;       /* (B,) metadata['warp'] or NULL */
  const int32_t* appearance_ids; /* (B,) or NULL */
  const int32_t* camera_ids;     /* (B,) or NULL */
  /* metadata_encoded=True (models.py:198-199, 210-211, 251; warping.py:378-381): the per-ray codes themselves instead
   * of ids into the embedding tables.  A non-NULL *_codes pointer replaces the matching *_ids.  Inference only. */
  const float* warp_codes;       /* (B, num_warp_features) */
  const float* appearance_codes; /* (B, num_appearance_features) */
  const float* camera_codes;     /* (B, num_camera_features) */
  const float* time;             /* (B,) metadata['time'] in [-1,1] (datasets/core.py:272-274); NRF_META_TIME only */
} nrf_rays;

/* The scalars that change from one optimisation step to the next (train.py:280-285 recomputes them from the schedules; the
 * rng key advances), in DEVICE memory.  A launch sequence captured into a hipGraph bakes every by-value argument in; with
 * nrf_step_scalars.dynamic set, the kernels read these values from the device instead, so the caller rewrites this struct
 * (one 64-byte host-to-device copy ahead of the replay) and replays the SAME graph for every step.  adam_c1 / adam_c2 are
 * the bias corrections 1 - beta^(step+1), formed in double on the host exactly as nrf_adam_step forms them. */
typedef struct nrf_dynamic_scalars {
  float warp_alpha;           /* replaces nrf_step_scalars.warp_alpha */
  float time_alpha;           /* replaces nrf_step_scalars.time_alpha */
  float elastic_loss_weight;  /* replaces nrf_elastic.loss_weight */
  float learning_rate;        /* nrf_adam_step_dynamic */
  float adam_c1;              /* nrf_adam_step_dynamic: 1 - beta1^(step+1) */
  float adam_c2;              /* nrf_adam_step_dynamic: 1 - beta2^(step+1) */
  float grad_scale;           /* nrf_adam_step_dynamic: 1 / world_size */
  float reserved0;
  uint64_t rng_seed;          /* replaces nrf_rand.seed (sampling streams, noise, the background draw) */
  uint64_t rng_offset;        /* replaces nrf_rand.offset */
  uint64_t reserved1[2];
} nrf_dynamic_scalars;        /* 64 bytes */

/* Writes *host_values into the device struct with a one-thread kernel on `stream` (the values travel as kernel arguments:
 * no host buffer has to outlive the call, nothing synchronises).  Call it ahead of every graph replay. */
int nrf_dynamic_scalars_write(nrf_dynamic_scalars* device_dst, const nrf_dynamic_scalars* host_values, void* stream);

/* warp_extra + the per-step scalars of training.ScalarParams (training.py:35-43). */
typedef struct nrf_step_scalars {
  float warp_alpha; /* warp_extra['alpha'] */
  float time_alpha; /* warp_extra['time_alpha']: annealing of the TimeEncoder's posenc (NRF_META_TIME) */
  const nrf_dynamic_scalars* dynamic; /* DEVICE pointer or NULL: when set, it overrides warp_alpha / time_alpha above,
                                         nrf_rand.seed / offset and nrf_elastic.loss_weight (graph-replayable step) */
} nrf_step_scalars;

/* Stand-in for the flax RNG streams 'coarse' / 'fine' (models.py:333,355).
 * Either explicit uniforms (parity runs) or an on-device Philox4x32-10 keyed by
 * (seed, offset) (throughput runs).  Ignored when use_stratified_sampling=0. */
typedef struct nrf_rand {
  const float* t_rand; /* (B,N_c) in [0,1) or NULL */
  const float* u;      /* (B,N_f) in [0,1) or NULL */
  uint64_t seed;
  uint64_t offset;
  /* noise_std > 0: explicit standard normals for model_utils.noise_regularize, (B,N_c) and (B,N_c+N_f), or NULL ->
   * Philox (streams 2, 3) + Box-Muller */
  const float* noise_coarse;
  const float* noise_fine;
} nrf_rand;

/* One level of the NerfModel output dict (models.py:278-287). Any pointer may
 * be NULL to skip that output. */
typedef struct nrf_level_out {
  float* rgb;       /* (B,3) */
  float* depth;     /* (B,)  */
  float* med_depth; /* (B,)  */
  float* acc;       /* (B,)  */
  float* weights;   /* (B,S) */
  float* z_vals;    /* (B,S)  (extra: the sample depths of this level) */
  float* points;        /* (B,S,3) sample points before the warp (return_points, models.py:247-248: with or without the warp) */
  float* warped_points; /* (B,S,3) after SE3Field (models.py:266-267); needs the warp field */
  float* warp_jacobian; /* (B,S,3,3) jax.jacfwd(SE3Field.warp) per sample, row-major d x'_i / d x_j (warping.py:385-387,
                           models.py:264-265); needs NRF_FLAG_WARP_JACOBIAN in flags and in nrf_workspace_bytes */
} nrf_level_out;

typedef struct nrf_outputs {
  nrf_level_out coarse;
  nrf_level_out fine;
} nrf_outputs;

/* flags for nrf_forward / nrf_workspace_bytes */
#define NRF_FLAG_TRAIN 1u   /* keep the activation stash nrf_backward needs */
#define NRF_FLAG_NO_WARP 2u /* NerfModel.__call__(use_warp=False) (models.py:296) */
#define NRF_FLAG_BF16 4u    /* NeRF-MLP operands in bfloat16 (fp32 accumulate, fp32 master weights / posenc / warp / composite /
                               loss / Adam); an opt-in mode with no reference counterpart (BASELINE config D) -- ~1e-2 on
                               rendered colour */
#define NRF_FLAG_WARP_JACOBIAN 8u /* return_warp_jacobian (models.py:297): forward-mode tangent pass of the warp per level */
#define NRF_FLAG_WARP_F32 16u     /* with NRF_FLAG_BF16 (or NRF_FLAG_BF16X3): keep SE3Field's 6 x 128 trunk on float32 operands (since 0.4.0 NRF_FLAG_BF16
                                     runs it on bfloat16 operands as well: annealed posenc, exp_se3, (w, v), the Jacobian algebra and
                                     the GLO table stay float32); a call that returns the warp Jacobian uses the float32 trunk anyway */
#define NRF_FLAG_BF16X3 32u       /* since 0.6.0, nrf_forward / nrf_workspace_bytes[_ex] only (inference): the NeRF MLPs
                                     (modules.py:26-62, 95-169) and SE3Field's trunk (warping.py:264-288) in split-bfloat16 arithmetic --
                                     every float32 operand as a bf16 pair hi + lo, a product as hi.hi + hi.lo + lo.hi on the bf16 matrix
                                     pipe, float32 accumulate: float32-EMULATING (an operand carries 16 mantissa bits; rendered colour
                                     within ~1e-6 of the float32 chains without the warp, ~1e-5 with it -- a warped point moves by ~1e-5
                                     and meets the 2^(F-1) posenc band --, far inside the 1e-3 parity gate), not bit-comparable with them.
                                     Posenc, exp_se3, sampling and compositing stay float32.  NRF_FLAG_WARP_F32 keeps the SE3 trunk on
                                     the float32 kernels (bit-identical warped points; a call that returns the warp Jacobian uses them
                                     anyway).  Not with NRF_FLAG_TRAIN or NRF_FLAG_BF16; same model limits as NRF_FLAG_BF16 */

int nrf_version(void);
const char* nrf_last_error(void);

/* models.construct_nerf (models.py:378-489) minus parameter init: validates the
 * configuration and fixes the parameter layout. */
int nrf_create(const nrf_model_desc* desc, nrf_handle* out);
int nrf_destroy(nrf_handle h);

int nrf_param_count(nrf_handle h, int64_t* n_floats);
/* out may be NULL to query *n only. */
int nrf_param_layout(nrf_handle h, nrf_tensor_info* out, int32_t* n);

int nrf_workspace_bytes(nrf_handle h, int32_t num_rays, uint32_t flags, size_t* bytes);

/* NerfModel.apply (models.py:289-375): sampling, [warp,] posenc, NeRF MLPs,
 * compositing, hierarchical resampling, fine pass. */
int nrf_forward(nrf_handle h, const float* params, const nrf_rays* rays,
                const nrf_step_scalars* scalars, const nrf_rand* rnd,
                const nrf_outputs* out, uint32_t flags, void* workspace,
                size_t workspace_bytes, void* stream);

/* Reverse pass of the nrf_forward(NRF_FLAG_TRAIN) call that last used this
 * workspace: the VJP jax.value_and_grad builds at training.py:264-265.
 * d_rgb_coarse / d_rgb_fine are dL/d(out[level]['rgb']) (B,3); either may be
 * NULL.  grad_params (flat, same layout as params) is OVERWRITTEN. */
int nrf_backward(nrf_handle h, const float* params, const nrf_rays* rays,
                 const float* d_rgb_coarse, const float* d_rgb_fine,
                 float* grad_params, void* workspace, size_t workspace_bytes,
                 void* stream);

/* training.train_step up to (excluding) pmean + Adam (training.py:168-265):
 * forward, loss = MSE_coarse + MSE_fine (training.py:172,261), backward.
 * target_rgb (B,3).  stats[NRF_NUM_STATS] (device): {mse_coarse, mse_fine, psnr_coarse,
 * psnr_fine, loss_total, 0...}.  grad_params is OVERWRITTEN. */
#define NRF_NUM_STATS 16
int nrf_train_step_loss_grad(nrf_handle h, const float* params, const nrf_rays* rays,
                             const float* target_rgb, const nrf_step_scalars* scalars,
                             const nrf_rand* rnd, float* grad_params, float* stats,
                             void* workspace, size_t workspace_bytes, void* stream);

/* Background-point regulariser of training.compute_background_loss (training.py:117-135, 248-259).  The
 * caller draws the warp id per point (random.choice over model.warp_ids) and adds the N(0, noise_std)
 * noise; the library warps the points with the shared SE3 field (create_warp_field(num_batch_dims=1),
 * models.py:165-184), evaluates  weight * mean(general_loss_with_squared_residual(|x'-x|^2, alpha, scale))
 * (utils.py:264-331) and ADDS its parameter gradient to grad_params. */
typedef struct nrf_background {
  int32_t num_points;
  const float* points;      /* (N,3) points: already noised when warp_ids is given, raw when the library draws */
  const int32_t* warp_ids;  /* (N,) or NULL: the library draws id = id_choices[floor(U * num_choices)] (random.choice over
                               model.warp_ids, training.py:121-123) and adds noise_std * N(0,1) to the points (:124-126) with
                               its Philox streams 4 and 5 of (nrf_rand.seed, offset) -- nothing left for the host to launch */
  float loss_weight;        /* scalar_params.background_loss_weight */
  float loss_alpha;         /* -2 (training.py:119) */
  float loss_scale;         /* 0.001 */
  const int32_t* id_choices;/* (num_choices,) model.warp_ids; used when warp_ids == NULL */
  int32_t num_choices;
  float noise_std;          /* scalar_params.background_noise_std; used when warp_ids == NULL */
} nrf_background;

/* Elastic regulariser of training.compute_elastic_loss (training.py:71-114, 177-197) on the COARSE samples
 * (models.py:345): J = jax.jacfwd(SE3Field.warp) per sample (warping.py:385-387), loss_type 'log_svals':
 * sum_k log(max(s_k(J), eps))^2 -> general_loss(alpha, scale) -> sum over samples with the stop-gradient
 * compositing weights (elastic_reduce_method 'weight'; 'median' instead keeps only the sample at
 * model_utils.compute_depth_index) -> mean over rays, times loss_weight.  The gradient
 * (reverse over the forward-mode Jacobian, incl. exp_se3's second derivatives) is added to grad_params. */
enum { NRF_ELASTIC_WEIGHT = 0, NRF_ELASTIC_MEDIAN = 1 };
/* elastic_loss_type (training.py:86-109): sq_residual = sum log(max(s,eps))^2 | sum (s-1)^2 | sum((J J^T - I)^2)/4 |
 * div(J)^2 | (det J - 1)^2 | log(max(det J, eps))^2.  'nr' (nearest rotation) is not built: the reference marks it as
 * producing NaNs (training.py:58). */
enum { NRF_ELASTIC_LOG_SVALS = 0, NRF_ELASTIC_SVALS = 1, NRF_ELASTIC_JTJ = 2, NRF_ELASTIC_DIV = 3, NRF_ELASTIC_DET = 4,
       NRF_ELASTIC_LOG_DET = 5 };
typedef struct nrf_elastic {
  float loss_weight;      /* scalar_params.elastic_loss_weight */
  int32_t reduce_method;  /* NRF_ELASTIC_WEIGHT / NRF_ELASTIC_MEDIAN */
  float eps;              /* 1e-6 */
  float loss_alpha;       /* -2 (training.py:112-113) */
  float loss_scale;       /* 0.03 */
  int32_t loss_type;      /* NRF_ELASTIC_LOG_SVALS ... */
} nrf_elastic;

/* use_warp_reg_loss (training.py:199-212), both levels: general_loss(|points - warped_points|^2 at the sample of
 * model_utils.compute_depth_index(stop_gradient(weights)), alpha, scale), mean over rays, times loss_weight. */
typedef struct nrf_warp_reg {
  float loss_weight;      /* scalar_params.warp_reg_loss_weight */
  float loss_alpha;       /* scalar_params.warp_reg_loss_alpha (-2) */
  float loss_scale;       /* scalar_params.warp_reg_loss_scale (0.001) */
} nrf_warp_reg;

/* nrf_train_step_loss_grad + the regularisers (bg, el, wr may each be NULL).  stats[5] = mean background loss
 * (unweighted, training.py:259), stats[6] = elastic loss (unweighted, training.py:195), stats[7] = mean elastic
 * residual (training.py:196), stats[8], stats[9] = warp_reg loss coarse / fine (training.py:210), stats[10], stats[11] =
 * mean warp_reg residual coarse / fine (:211), stats[12..14] = mean det / div / |curl| of the coarse warp Jacobian
 * (training.py:214-222, with el); stats[4] includes the weighted terms.  The workspace must come from
 * nrf_workspace_bytes_ex with the same num_background_points / use_elastic_loss.  flags: 0 or NRF_FLAG_BF16. */
int nrf_train_step_loss_grad_ex(nrf_handle h, const float* params, const nrf_rays* rays, const float* target_rgb,
                                const nrf_step_scalars* scalars, const nrf_rand* rnd, const nrf_background* bg,
                                const nrf_elastic* el, const nrf_warp_reg* wr, uint32_t flags, float* grad_params,
                                float* stats, void* workspace, size_t workspace_bytes, void* stream);
int nrf_workspace_bytes_ex(nrf_handle h, int32_t num_rays, uint32_t flags, int32_t num_background_points,
                           int32_t use_elastic_loss, size_t* bytes);

/* warping.SE3Field on arbitrary points with one warp id per point: model.create_warp_field(num_batch_dims=1)
 * .apply(points, metadata, warp_extra, False, False) (models.py:165-184, warping.py:355-389). */
int nrf_warp_points_workspace_bytes(nrf_handle h, int32_t num_points, size_t* bytes);
int nrf_warp_points(nrf_handle h, const float* params, const float* points, const int32_t* warp_ids, int32_t num_points,
                    const nrf_step_scalars* scalars, float* warped, void* workspace, size_t workspace_bytes, void* stream);

/* flax.optim.Adam.apply_gradient (training.py:268-269; flax 0.3.4 optim/adam.py):
 * g = grad*grad_scale (grad_scale = 1/world_size folds lax.pmean, training.py:266),
 * m,v updated in place, step = number of updates already applied.  Hyper-parameters are doubles
 * (flax keeps them as Python floats: 1-beta2 is formed in double, then rounded to fp32). */
int nrf_adam_step(float* params, float* m, float* v, const float* grad, int64_t n,
                  double lr, double beta1, double beta2, double eps, int64_t step,
                  double grad_scale, void* stream);
/* The same update with lr, the bias corrections and grad_scale read from DEVICE memory (nrf_dynamic_scalars): the launch
 * can sit in a captured graph and still follow the learning-rate schedule and the step count. */
int nrf_adam_step_dynamic(float* params, float* m, float* v, const float* grad, int64_t n, double beta1, double beta2,
                          double eps, const nrf_dynamic_scalars* dynamic, void* stream);

/* ---- measurement: per-kernel wall time from HIP events recorded on the launch stream ----
 * (the reference only has utils.TimeTracker host timers, utils.py:418-465, around an
 * asynchronously dispatched step; these are device-side).  Off by default; when on, entry points
 * are no longer hipGraph-capturable. */
typedef struct nrf_profile_entry {
  char name[32];           /* e.g. "mlp_fwd_fine", "wgrad" */
  double ms;               /* accumulated since the last read */
  int32_t launches;
  int32_t pad_;
  double flops_per_launch; /* ALGORITHMIC flops (2/MAC, dense layers, unpadded) */
} nrf_profile_entry;
int nrf_profile_enable(nrf_handle h, int32_t on);
/* Waits for the recorded events, returns and resets the accumulators.  out may be NULL to query *n. */
int nrf_profile_read(nrf_handle h, nrf_profile_entry* out, int32_t* n);

/* Calibration aid for the wgrad stream-K partition: for each segment of the last
 * nrf_backward / nrf_train_step_loss_grad on `workspace`, 6 doubles {workgroup, group, Kb, Nb,
 * tiles, wall-clock ticks (100 MHz)}.  Synchronous (hipMemcpy); out may be NULL to query *n. */
int nrf_debug_wgrad_segments(nrf_handle h, const void* workspace, double* out, int32_t* n);

/* Test aid: float offset inside the workspace of an internal buffer of `level` (0 coarse, 1 fine, 2 background
 * points, 3 Jacobian tangents) for the last planned (num_rays, flags): "st_pe", "st_h", "st_bn", "st_rgbh",
 * "dy_trunk", "dy_bn", "dy_rgbh", "d_raw4", "z", "out4", "wpoints", "d_points", "w_st_win", "w_st_h", "w_st_wv",
 * "w_dy", "w_dw4", "w_dv4", "bits_trunk", "bits_rgbh", "w_bits".
 * Stash tiles are [features][64 rows] in fragment order (csrc/chain_common.h frag_index); the ReLU sign bits are one
 * uint32 per (tile, wave, lane, column block): nibble q, bit e <-> tile row 4 * ((q&1) + 2*(lane>>5) + 4*(q>>1)) + e of
 * feature wave * 32 * NCB + 32 * cb + (lane & 31)  (NCB = 2 for the 256-wide trunk, 1 otherwise). */
int nrf_debug_ws_offset(nrf_handle h, const char* name, int32_t level, int64_t* float_offset);

/* Tuning options of a handle (no reference counterpart: XLA picks its own tilings).  Must be set before the first
 * nrf_workspace_bytes / nrf_forward call that uses the handle with a given batch size, or between steps (the next call
 * re-plans; a stashed forward cannot be differentiated across a change: NRF_E_STATE).  Unknown option / value:
 * NRF_E_UNSUPPORTED.
 *   NRF_OPT_CHAIN_TILE_ROWS  rows per workgroup tile of the float32 NeRF-MLP chain kernels (modules.py:95-169):
 *                            64 = two workgroups per CU (csrc/mlp_chain.hip), 32 = four per CU (csrc/mlp_chain32.hip),
 *                            0 = automatic (default: half tiles for forward launches that under-fill the 64-row grid).  The
 *                            forward results (and the stashes) are bit-identical under both tilings -- a ray's result does not
 *                            depend on the launch it rides in --, gradients agree to the order of float atomics; the workspace
 *                            layout does not depend on it. */
#define NRF_OPT_CHAIN_TILE_ROWS 1
/*   NRF_OPT_BF16_WGRAD_MERGE bf16 training mode: 1 (default) = the weight-gradient GEMMs of the skip layer (X = [h4 | posenc]) and of
 *                            the bottleneck + alpha head (dY = [d bottleneck | d raw]) run as ONE group each, so dpre_4 and h8 are
 *                            streamed once (HBM fetch of the kernel 1.16 x -> 1.03 x its algorithmic bytes; same-box A/B: the step
 *                            1-2 % faster, profiles/r05_wgrad_bf16_merge_ab.md); 0 = one group per weight matrix (rounds 2-4).  Same
 *                            results to float32 summation order; changes the workspace size (query nrf_workspace_bytes* after
 *                            setting it). */
#define NRF_OPT_BF16_WGRAD_MERGE 2
int nrf_set_option(nrf_handle h, int32_t option, int64_t value);

/* ---- individual operators (same device code the fused path runs), exposed so
 * parity tests can check each reference function in isolation. ---- */

/* model_utils.sample_along_rays (model_utils.py:36-73) -> z_vals (B,N). */
int nrf_sample_along_rays(const float* origins, const float* directions, int32_t num_rays,
                          int32_t num_samples, float near_plane, float far_plane,
                          int32_t stratified, int32_t linear_disparity,
                          const float* t_rand, uint64_t seed, uint64_t offset,
                          float* z_vals, void* stream);

/* model_utils.volumetric_rendering (+ compute_depth_map) (model_utils.py:76-136,
 * 218-263).  rgb_sigma is (B,S,4) = (r,g,b,sigma) post-activation. */
int nrf_volumetric_rendering(const float* rgb_sigma, const float* z_vals, const float* directions,
                             int32_t num_rays, int32_t num_samples, int32_t white_background,
                             int32_t sample_at_infinity, const nrf_level_out* out, void* stream);

/* model_utils.sample_pdf (model_utils.py:139-215) applied as models.py:353-357:
 * bins = midpoints of z_coarse, weights = weights_coarse[:,1:-1]; returns the
 * sorted union (B, N_c+N_f). */
int nrf_sample_pdf(const float* z_coarse, const float* weights_coarse, int32_t num_rays,
                   int32_t num_coarse, int32_t num_fine, int32_t stratified, const float* u,
                   uint64_t seed, uint64_t offset, float* z_out, void* stream);

/* ---- camera geometry (the eval / dataset side of the path) ----
 * nerfies/camera.py Camera: orientation is the world-to-camera rotation (row-major),
 * position the camera centre in world space; intrinsics as camera.py:110-140. */
typedef struct nrf_camera {
  float orientation[9];
  float position[3];
  float focal_length;
  float principal_point[2];
  float skew;
  float pixel_aspect_ratio;
  float radial_distortion[3];      /* k1 k2 k3 */
  float tangential_distortion[2];  /* p1 p2 */
  int32_t image_size[2];           /* width, height */
} nrf_camera;

/* Camera.pixels_to_rays (camera.py:244-269) incl. the fixed 10-iteration Newton
 * undistort (camera.py:26-105).  pixels: device [n,2] fp32, or NULL for the pixel
 * centres of the whole image in row-major [H,W] order (get_pixel_centers,
 * camera.py:317-321; then n must equal width*height) -- which makes this
 * datasets/core.py:50-75 camera_to_rays in one launch: directions [n,3], and,
 * when non-NULL, origins [n,3] (the tiled camera position) and pixels_out [n,2]. */
int nrf_camera_pixels_to_rays(const nrf_camera* camera, const float* pixels, int64_t n, float* origins,
                              float* directions, float* pixels_out, void* stream);

/* Camera.pixels_to_points (camera.py:271-277): points [n,3] at `depth` [n] measured
 * along the optical axis. */
int nrf_camera_pixels_to_points(const nrf_camera* camera, const float* pixels, const float* depth, int64_t n,
                                float* points, void* stream);

/* Camera.project (camera.py:283-315): world points [n,3] -> distorted pixel positions [n,2]. */
int nrf_camera_project(const nrf_camera* camera, const float* points, int64_t n, float* pixels, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFIES_AMD_H_ */
