/* caddy_hip.h -- C ABI of libcaddy_hip.so: the MI355X-native (gfx950) CADDY hot path.
 *
 * Drop-in boundary.  The reference is pure Python, so the "FFI" a maintainer binds is ctypes
 * (see INTEGRATION.md); every entry point takes raw device pointers + sizes and returns an int status
 * (0 = ok, negative = error, text via caddy_last_error()).  Nothing here allocates device memory: the caller hands over
 * one workspace (caddy_workspace_bytes) and the flat parameter / gradient buffers (caddy_param_floats /
 * caddy_trainable_floats floats), laid out per caddy_param_info_get -- tensors at the boundary are in the reference's
 * state_dict format (OIHW fp32, names of /root/reference model/main_model/model.py's state_dict).
 *
 * Each function cites the reference interface it replaces (paths relative to the reference repository).
 */
#ifndef CADDY_HIP_H
#define CADDY_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct caddy_ctx caddy_ctx;

/* Hyper-parameters read by the hot path (model/main_model/model.py:28-38, "Config keys" in SURVEY.md section 8a). */
typedef struct caddy_config {
    int variant;        /* 0: model.main_model.model, 1: model.reduced_model.model (config["model"]["architecture"]) */
    int batch;          /* B: clips per forward (per GPU) */
    int seq_len;        /* T: observations per clip */
    int height, width;  /* frame size, multiples of 16 */
    int stacking;       /* training.batching.observation_stacking */
    int actions;        /* data.actions_count */
    int action_dim;     /* model.action_network.action_space_dimension */
    int hidden;         /* model.dynamics_network.hidden_state_size (128 main / 64 reduced) */
    int use_gumbel, hard_gumbel, use_variations;   /* model.action_network.* */
    float centroid_alpha;                          /* model.centroid_estimator.alpha */
    int perceptual;     /* != 0: the context also holds the VGG19 perceptual loss (training/losses.py:379-491): the workspace grows by the
                           packed VGG19 weights and feature maps, caddy_load_vgg must be called before a caddy_loss_backward with
                           caddy_loss_cfg.perceptual != 0 */
    int ensemble;       /* model.action_network.ensamble_size (model/main_model/model.py:28,47): N action networks `action_network.{0..N-1}.*` in the parameter table; 0 or 1 = one.
                           The caller draws the member of a forward pass (model.py:152 / :358: random.choice) and names it with caddy_set_action_member.  At most CADDY_MAX_ENSEMBLE. */
} caddy_config;
#define CADDY_MAX_ENSEMBLE 8

typedef struct caddy_param_info {
    char name[128];   /* reference state_dict key */
    long offset;      /* float offset into the flat parameter buffer (and, for kind 0, the flat gradient buffer) */
    int ndim;
    int shape[4];
    int kind;         /* 0 trainable parameter, 1 BatchNorm running statistic, 2 non-trainable parameter (centroids) */
} caddy_param_info;

/* Noise drawn by the reference from torch's CPU generator, in its call order; device pointers. */
typedef struct caddy_noise {
    const float* eps_states;      /* (B*T, Da)      action_network.py:45 via :92  (first A call)            */
    const float* eps_dirs;        /* (B, T-1, Da)   action_network.py:45 via :108 (first A call)            */
    const float* gumbel_uniform;  /* (B*(T-1), K)   gumbel_softmax.py:33                                    */
    const float* eps_states_rec;  /* second A call (reconstructed states), model.py:277                     */
    const float* eps_dirs_rec;
} caddy_noise;

/* Loss weights of Trainer.compute_losses (training/trainer.py:494-500).  mi_ema: SmoothMutualInformationLoss state (K*K floats,
 * device) or NULL.  perceptual: loss_weights.perceptual_loss_lambda[_pretraining] (training/trainer.py:283,442) -- the VGG19 term of
 * ParallelPerceptualLoss (training/losses.py:379-491); non-zero requires caddy_config.perceptual and a preceding caddy_load_vgg.
 * perceptual_log: evaluate and report the perceptual losses even when the weight is 0 (the reference always logs them). */
typedef struct caddy_loss_cfg {
    double rec, states, entropy, dir_kl, mi, state_kl, hidden, mi_entropy_lambda;
    float* mi_ema;
    float mi_ema_alpha;
    int update_mi_ema;
    double perceptual;
    int perceptual_log;
    int diagnostics;    /* != 0: also evaluate the logging-only scalars of the reference's loss_info (trainer.py:475-491, :358-375) on the device, into
                           losses_host[CADDY_DIAG_0 ...]: no tensor leaves the GPU and no extra host synchronisation is needed for them */
    int no_sync;        /* != 0: losses_host (PINNED host memory) is filled by an asynchronous copy on the stream and the call returns without waiting for it -- the caller
                           reads it after its own event / stream synchronisation, typically once the NEXT step has been enqueued (reference: the .item() calls of
                           training/trainer.py:503-530 wait for the GPU every step) */
} caddy_loss_cfg;

/* losses_host slots.  CADDY_LOSS_PERCEPTUAL = avg_perceptual_loss, _TERM = loss_component_perceptual_loss (trainer.py:505,512);
 * CADDY_LOSS_PERC_R0 + 6 r = perceptual_loss_r{r}, + 1 + l = perceptual_loss_r{r}_l{l} -- l = 0 equals the resolution's total, exactly as the
 * reference logs it (in-place aliasing at training/losses.py:483-487, which also makes levels 1..4 count twice in the term). */
enum { CADDY_LOSS_TOTAL = 0, CADDY_LOSS_REC, CADDY_LOSS_STATES, CADDY_LOSS_ENTROPY, CADDY_LOSS_DIRKL, CADDY_LOSS_MI,
       CADDY_LOSS_STATEKL, CADDY_LOSS_HIDDEN, CADDY_LOSS_L1_R0, CADDY_LOSS_L1_R1, CADDY_LOSS_L1_R2,
       CADDY_LOSS_PERCEPTUAL = 11, CADDY_LOSS_PERCEPTUAL_TERM = 12,
       CADDY_LOSS_F16_SATURATED = 13,      /* 1.0: a split-f16 forward convolution (model or VGG19) staged |x| > 65504 since the last caddy_f16_saturated poll; it was clamped to the f16
                                              range -- call caddy_f16_saturated(ctx): the reporting layers move to a forward without a range limit */
       CADDY_LOSS_PERC_R0 = 16,
       /* caddy_loss_cfg.diagnostics: samples_entropy, action_distribution_entropy, states_magnitude, hidden_states_magnitude, action_directions_{mean,variance}_magnitude,
        * reconstructed_action_directions_{mean,variance}_magnitude, action_directions_reconstruction_error, reconstructed_action_directions_kl_loss,
        * centroids_mean_magnitude, average_centroids_distance, average_action_variations_norm_l2, action_variations_mean -- in this order */
       CADDY_DIAG_0 = 40, CADDY_DIAG_COUNT = 14, CADDY_LOSS_SLOTS = 56 };

/* Output ids of caddy_get_output: 0..19 = positions of the 20-tuple returned by Model.forward_full_model
 * (model/main_model/model.py:280-286); 100+r = r-th entry of the multi-resolution list (tuple position 1). */
enum { CADDY_OUT_MULTIRES0 = 100 };

const char* caddy_last_error(void);

/* --- parameters: replaces nn.Module.state_dict()/load_state_dict()/parameters() (training/trainer.py:36,80-122) --- */
int caddy_param_count(const caddy_config* cfg);
int caddy_param_info_get(const caddy_config* cfg, int index, caddy_param_info* out);
long caddy_param_floats(const caddy_config* cfg);       /* size of the flat parameter buffer (all kinds) */
long caddy_trainable_floats(const caddy_config* cfg);   /* kind-0 entries come first: [0, trainable) */

/* --- VGG19 weights of the perceptual loss: replaces torchvision.models.vgg19(pretrained=True).features inside Vgg19.__init__
 *     (model/layers/vgg.py:16-34).  The 13 convolutions the reference's slices evaluate (features.0 ... features.28; conv5_2..5_4 are
 *     loaded by torchvision but never run), names / shapes / float offsets per caddy_vgg_param_info_get (kind = 3), OIHW fp32.
 *     caddy_load_vgg packs them into the workspace (frozen: once, not per step); `vgg_flat` is not referenced afterwards. --- */
int caddy_vgg_param_count(void);
int caddy_vgg_param_info_get(int index, caddy_param_info* out);
long caddy_vgg_param_floats(void);
int caddy_load_vgg(caddy_ctx* ctx, const float* vgg_flat);
/* Arithmetic of the wide 3x3 convolutions (DESIGN.md "numerics"):  0 = exact fp32 (v_mfma_f32_32x32x2_f32);  16 = split f16 (hi + lo operands,
 * 3 products on the 16-bit matrix pipe, fp32 accumulate: fp32-class, forward default);  17 = split bf16 (3 products, 2^-16 with the full
 * fp32 exponent range: gradient default);  18 / 19 = single-product f16 / bf16 operands (VGG19 only).  caddy_set_precision: the model's
 * convolutions (forward: 0 | 16, backward: 0 | 17); caddy_set_vgg_precision: the perceptual loss network (forward 0 | 16 | 18, dgrad 0 | 17 | 19). */
int caddy_set_precision(caddy_ctx* ctx, int forward, int backward);
/* on (default): with caddy_config.perceptual and loaded VGG19 weights, a TRAINING-mode forward also runs the ground-truth branch of the
 * perceptual loss (resize + VGG19 features of the observations: independent of the model) on the driver's side stream, concurrently with
 * the model's forward pass; caddy_loss_backward then only runs the reconstruction branch.  off: everything inside caddy_loss_backward. */
int caddy_set_perceptual_prefetch(caddy_ctx* ctx, int on);
/* Evaluation (evaluation/evaluator.py:55,62,193-197: SequenceLossEvaluator(ParallelPerceptualLoss())): after a forward pass, the FULL-RESOLUTION perceptual distance per
 * reconstructed frame and VGG19 level, out_host[l * N + n] = mean |relu{l+1}_1(rec_n) - relu{l+1}_1(gt_n)|, n = b * Trec + t (Trec = T - 1, or T after forward_pretraining),
 * l = 0..4; 5 * N doubles, host memory; waits for the stream.  Position t of the reference's "perceptual_loss/pos_t" = sum_l mean_b out[l][b * Trec + t - off]. */
int caddy_perceptual_per_frame(caddy_ctx* ctx, double* out_host);
/* ... and the per-frame reconstruction / state losses of the same evaluator (evaluator.py:192,194, ObservationsLoss / StatesLoss per sequence position), from the loss-kernel
 * family: l1_host[b * Trec + t] = mean |frame - ground-truth observation[:3]| at full resolution, mse_host[b * T + t] = mean (reconstructed state - state)^2; host memory,
 * either may be NULL; waits for the stream. */
int caddy_sequence_losses_per_frame(caddy_ctx* ctx, double* l1_host, double* mse_host);
int caddy_set_vgg_precision(caddy_ctx* ctx, int forward, int dgrad);
/* on (default): caddy_start_inference folds every eval-mode BatchNorm of the roll-out path (E, R's non-recurrent blocks, D) into the packed
 * weights / bias of the convolution in front of it, and caddy_generate_next runs the folded graph (LeakyReLU and the residual add in the conv
 * epilogues, the ConvLSTM cells' BatchNorm as a second output of the gate kernel): ~35 fewer launches per frame.  off: one BatchNorm launch per
 * nn.BatchNorm2d as in the training graph.  Only caddy_generate_next is affected (model.py:570-607, eval mode). */
int caddy_set_rollout_fold(caddy_ctx* ctx, int on);
/* Bit-reproducible backward pass (default ON since round 5: two backward passes over the same forward give bit-identical gradients, like the reference's CPU path -- training/
 * trainer.py:575-587 on torch CPU kernels).  The forward pass is always bit-reproducible (fixed-order split-K, action indices!).  on = 0 selects the arrival-order form: the partial
 * sums of under-filled dgrads and of the weight-gradient pixel splits meet through fp32 atomics (run-to-run ~1e-5 relative on the flat gradient, amplified by BPTT through the
 * closed-loop steps); it is < 1 % faster (profiles/r05_experiments.md: slabs + fixed-order folds cost 0.6 / 0.9 ms of the 61.5 / 125.2 ms steps). */
int caddy_set_deterministic(caddy_ctx* ctx, int on);

/* --- context --- */
size_t caddy_workspace_bytes(const caddy_config* cfg);
caddy_ctx* caddy_ctx_create(const caddy_config* cfg, float* params, float* grads, void* workspace, size_t workspace_bytes);
void caddy_ctx_destroy(caddy_ctx* ctx);
int caddy_set_stream(caddy_ctx* ctx, void* hip_stream);
/* Bucketed gradient all-reduce (BASELINE north star: "all-reduce of gradients overlapped with the backward on a side HIP stream").
 * All weights are shared over time, so nothing is final before BPTT reaches t = 0; what IS final once the time loop's backward
 * is done are the dynamics_network (85 % of the bytes) and rendering_network ranges of the flat gradient buffer.  During
 * caddy_loss_backward of a caddy_forward_full graph the hook is called once per range, `stream` being the HIP stream on which those
 * gradients become valid (the caller enqueues its all-reduce behind it); the ranges must not be touched again until the caller has waited
 * for its collective.  The remaining ranges are valid when caddy_loss_backward returns (on the ctx stream), as before. */
typedef void (*caddy_grads_ready_hook)(float* grads, long offset, long count, void* stream, void* user);
int caddy_set_grads_ready_hook(caddy_ctx* ctx, caddy_grads_ready_hook hook, void* user);
/* Native data parallelism: the same three reductions issued from C into RCCL (ncclAllReduce over xGMI) on a communicator owned by the context -- replaces
 * nn.DataParallel's per-step replicate / gather / reduce-add (train.py:67-68).  One process per GPU:
 *   rank 0: caddy_dp_unique_id(id) -> the 256 bytes (TWO ncclUniqueIds) travel to every rank (any out-of-band channel, e.g. a torch.distributed broadcast) ->
 *   every rank: caddy_dp_init(ctx, id, world_size, rank, overlap) [overlap = 1: R / D gradient buckets behind the side stream during the backward] ->
 *   per step: caddy_forward_full, caddy_loss_backward, caddy_allreduce_grads (the rest + join), caddy_adam_step(..., grad_scale = 1 / world_size).
 * One communicator per stream: the first id's communicator carries every collective issued on the context's stream, the second one's (created when overlap = 1) the
 * gradient buckets on the side stream -- each communicator sees one totally ordered sequence of collectives on every rank.  caddy_dp_init blocks until every rank has
 * called it, at most CADDY_DP_INIT_TIMEOUT_S seconds (default 180): then it returns -3 with a message instead of hanging.
 * RCCL is resolved with dlopen at the first call (CADDY_RCCL_LIB, librccl.so of the process, /opt/rocm/lib): caddy_dp_available() says whether one was found. */
int caddy_dp_available(void);
int caddy_dp_unique_id(char* out256);
int caddy_dp_init(caddy_ctx* ctx, const char* id256, int world_size, int rank, int overlap);
int caddy_allreduce_grads(caddy_ctx* ctx);
long caddy_dp_bucket_floats(caddy_ctx* ctx);
int caddy_dp_shutdown(caddy_ctx* ctx);
/* Evaluation samplers (evaluation/action_sampler.py:14,63; evaluation/action_variation_sampler.py:14), consumed mid-forward exactly where
 * model/main_model/model.py:171-173,189-190 call them.  The hook runs stream-ordered on device pointers inside the workspace:
 *   stage 0 (if provides_samples):    write samples (n, K)    given log_probs (n, K)                      [n = batch * (seq_len - 1)]
 *   stage 1 (if provides_variations): write variations (n, Da) given sampled_dirs (n, Da) and the final samples (n, K)
 * hook == NULL clears it.  Affects caddy_forward_full / caddy_forward_pretraining until cleared. */
typedef void (*caddy_sampler_hook)(const float* log_probs, const float* sampled_dirs, float* samples, float* variations,
                                   int n, int K, int Da, int stage, void* user);
int caddy_set_sampler_hook(caddy_ctx* ctx, caddy_sampler_hook hook, void* user, int provides_samples, int provides_variations);
/* Data parallelism (one process per GPU): `hook(ptr, n, user)` must sum the n floats at device pointer `ptr` (inside the
 * workspace) over all ranks, in place, stream-ordered (RCCL all-reduce).  It is called for the centroid-EMA sums
 * (centroid_estimator.py:61-63) during caddy_forward_* and for the K x K joint matrix of the mutual-information loss
 * (losses.py:262) during caddy_loss_backward, which gives both the reference's global-batch semantics (the reference
 * computes them on GPU0 over the gathered batch under nn.DataParallel).  The parameter gradients themselves are reduced by
 * the caller with ONE all-reduce of the flat gradient buffer after caddy_loss_backward. */
int caddy_set_allreduce_hook(caddy_ctx* ctx, void (*hook)(float* device_ptr, int count, void* user), void* user, int world_size);

/* --- Model.forward(batch_tuple, ground_truth_observations_init, gumbel_temperature=...) in full-model mode:
 *     model/main_model/model.py:57-82 -> forward_full_model :84-286.  obs: (B,T,3S,H,W) fp32 device, reference layout.
 *     training != 0: train-mode BatchNorm + centroid EMA + backward tape; 0: eval mode (evaluation/evaluator.py:125).
 *     samples_in / variations_in (nullable): outputs of an evaluation action_sampler / action_variation_sampler. --- */
int caddy_forward_full(caddy_ctx* ctx, const float* obs, int gt_init, float tau, const caddy_noise* noise, int training,
                       const float* samples_in, const float* variations_in);
/* --- Model.forward(batch_tuple, pretraining=True, ...) -> forward_pretraining (model/main_model/model.py:290-468).
 *     Afterwards caddy_get_output ids follow THAT function's tuple (4 = reconstructed hidden states (B,T,Ch,h,w),
 *     5 = hidden states, 6 = selected actions, 7 = logits, 8 = samples, 9 = attention; 10..19 as in full mode; frames have T entries)
 *     and caddy_loss_backward adds the `hidden` term (HiddenStatesLoss, training/trainer.py:313). --- */
int caddy_forward_pretraining(caddy_ctx* ctx, const float* obs, float tau, const caddy_noise* noise, int training,
                              const float* samples_in, const float* variations_in);
int caddy_get_output(caddy_ctx* ctx, int id, void* dst);      /* copy one output, converted to the reference layout */
/* d(loss)/d(output) after caddy_loss_backward, same layout (what autograd holds in `.grad` of a retained output);
 * available for ids 0, 100-102, 2, 3, 4, 6, 8, 9, 10, 12, 15, 16, 18. */
int caddy_get_output_grad(caddy_ctx* ctx, int id, void* dst);

/* --- losses + loss.backward(): training/trainer.py:447-500,585 fused into one pass (L1 multi-resolution, VGG19 perceptual, states MSE,
 *     entropy, direction KL, (smooth) mutual information, action-state KL) followed by BPTT through D, R, E, A.
 *     Gradients land in the flat gradient buffer (reference layout).  losses_host: CADDY_LOSS_SLOTS doubles (host). --- */
int caddy_loss_backward(caddy_ctx* ctx, const caddy_loss_cfg* cfg, double* losses_host);

/* --- optimizer.step(): torch.optim.Adam with L2 weight decay (training/trainer.py:36,586); m, v: trainable floats --- */
int caddy_adam_step(caddy_ctx* ctx, float* m, float* v, float lr, float beta1, float beta2, float eps, float weight_decay,
                    int step, float grad_scale);
/* Ensemble of action networks (caddy_config.ensemble > 1; model/main_model/model.py:152,274 / :358,456: both A calls of a forward pass use the member drawn with random.choice).
 * caddy_set_action_member names the member of the NEXT forward pass.  The members that were not drawn receive no gradient (`.grad is None` in the reference): caddy_adam_step leaves
 * their moments and weights untouched, as torch.optim.Adam does, and applies the drawn member's range with ITS OWN step count `member_step` (torch keeps `step` per parameter: a member
 * drawn k times has been stepped k times); every other parameter uses `step`. */
int caddy_set_action_member(caddy_ctx* ctx, int member);
int caddy_adam_step_member(caddy_ctx* ctx, float* m, float* v, float lr, float beta1, float beta2, float eps, float weight_decay, int step, int member_step, float grad_scale);
/* The same step with torch.optim.Adam's per-parameter bookkeeping passed explicitly, for BOTH forms of optimizer.zero_grad() (training/trainer.py:584): member_step[k] (one per
 * ensemble member; NULL = only the drawn member, with `step`) and s2h_step (state_to_hidden_state_layer, model.py:41-43) are the bias-correction counts of those ranges; 0 skips a
 * range (its .grad is None: torch >= 2.0's set_to_none default for everything the last backward did not reach), > 0 steps it -- with the real gradient where the last pass produced
 * one (the drawn member; state_to_hidden_state_layer after caddy_forward_pretraining), with a ZERO gradient otherwise (torch < 2.0, the reference's pinned 1.4.0: zero_grad()
 * zero-fills, so a parameter that has had a gradient once keeps being decayed and its moments keep ageing).  The host mirror (trainer.py: training.zero_grad_semantics) keeps the counts. */
int caddy_adam_step_ex(caddy_ctx* ctx, float* m, float* v, float lr, float beta1, float beta2, float eps, float weight_decay, int step, const int* member_step, int s2h_step,
                       float grad_scale);

/* --- play.py roll-out: Model.start_inference (model.py:561-568) / Model.generate_next (model.py:570-607), eval mode.
 *     observation: (3S,H,W); variation: (Da) or NULL (= zeros, noise=False); frame_out: (3,H,W); obs_out: (3S,H,W) or NULL.
 *     observation, frame_out and obs_out must NOT overlap (obs_out = cat[frame, observation[:-3]] is written while observation is read):
 *     an overlapping call is rejected with -2. --- */
int caddy_start_inference(caddy_ctx* ctx);
/* Poll of the f16 range guards (one flag word per convolution layer, sticky on the device until this call reads and clears them; waits for the stream).  Return value: bit 0 -- a
 * split-f16 forward convolution of the model (model/layers/) or of VGG19 (model/layers/vgg.py:20-36) met an input beyond the f16 range (|x| > 65504) since the last poll and clamped
 * it; bit 1 -- a NaN was among them (the clamp made it finite: the loss call of that pass reported a NaN total, as the reference's fp32 arithmetic would have).  The layers that
 * reported -- and only those -- run without a range limit from the next forward pass on (exact fp32 for model layers, split bf16 for VGG19 layers), for the lifetime of the context;
 * caddy_fallback_layers returns how many have moved so far. */
int caddy_f16_saturated(caddy_ctx* ctx);
int caddy_fallback_layers(caddy_ctx* ctx);
int caddy_generate_next(caddy_ctx* ctx, const float* observation, int action, const float* variation, float* frame_out, float* obs_out);

/* --- live kernel timing: HIP events recorded on the launch stream around every conv launch between begin and end.
 *     out = CADDY_PROFILE_FAMILIES kernel families (csrc/common.h CK_*: the four k_conv_fwd tilings, k_conv_thin_out, k_conv_thin_in, three
 *             k_conv_wgrad tilings, k_conv_wgrad_small, k_wgrad_thin, k_conv_wgrad_tile, k_conv_narrow, k_conv_hx<128|64|32> on 4 waves, k_wgrad_hx, k_conv_hx<128> on 8 waves)
 *             x {launches, algorithmic FLOPs, total milliseconds, algorithmic bytes} (SURVEY 8d definitions) --- */
#define CADDY_PROFILE_FAMILIES 18
/* test aid: NaN-fill the not-zero-filled (first-touch) part of the gradient arena before every backward pass */
int caddy_debug_set_poison(caddy_ctx* ctx, int on);
/* test aid: caddy_loss_backward stops after the loss kernels, so caddy_get_output_grad returns the gradient of the DIRECT loss terms only
 * (what autograd holds in `.grad` of the stacked output tensors, which the D->E feedback does not read) */
int caddy_debug_set_seeds_only(caddy_ctx* ctx, int on);
int caddy_profile_begin(caddy_ctx* ctx);
int caddy_profile_end(caddy_ctx* ctx, double* out /* 4 * CADDY_PROFILE_FAMILIES doubles */);
/* per-launch records since caddy_profile_begin (call before caddy_profile_end): 7 doubles each
 * {kind 0 fwd / 1 dgrad / 2 wgrad / 3 VGG19 forward / 4 VGG19 dgrad, output pixels, K (padded input channels), Cout, kernel size, algorithmic FLOPs, ms} */
int caddy_profile_records(caddy_ctx* ctx, double* out, int max_records);
/* phase marks recorded on the ctx stream during profiled steps (forward: pack / E on the ground truth / A / teacher-forced steps / closed-loop steps / A on the
 * reconstructions; backward in reverse): names_out = max x 48 bytes, ms_out[i] = time since the previous mark.  Call before caddy_profile_end. */
int caddy_profile_phases(caddy_ctx* ctx, char* names_out, float* ms_out, int max);

/* --- introspection (debug / tests): the i-th intermediate activation (grad=0) or its gradient (grad=1) of the last
 *     forward, converted to (N,C,H,W) --- */
int caddy_debug_count(caddy_ctx* ctx);
/* BatchNorm fusion bookkeeping since creation: out3 = {train-mode BatchNorm calls, ... whose statistics came from the producing conv's epilogue,
 * ... whose normalised output was never materialised (applied by the consuming convolution)} */
int caddy_debug_fusion_counts(caddy_ctx* ctx, long* out3);
/* tests / A-B: which BatchNorm paths the driver may pick (all default to 1): the one-launch kernel for tiny maps, the lazily applied form, statistics from the conv epilogue */
int caddy_debug_set_bn_paths(caddy_ctx* ctx, int small, int lazy, int epilogue_stats);
/* tests / A-B runs: 0 keeps every VGG19 feature map of the perceptual loss (training/losses.py:379-491, model/layers/vgg.py:20-36) as an fp32 tensor; 1 (default) lets well-filled
 * layers exchange them pre-split for the 16-bit matrix pipe ("S16" tensors, csrc/common.h).  Same convolution results bit for bit; the feature L1 sees hi + lo (2^-22 relative). */
int caddy_debug_set_vgg_s16(caddy_ctx* ctx, int on);
/* tests / A-B runs (round 6): 0 keeps the gradient of every convolution output (the dY of model/layers/residual_block.py:49-68, same_block.py:34-47, up_block.py:31-45,
 * convolutional_lstm_cell.py:92-101) as an fp32 tensor; 1 (default) lets the point-wise backward kernel that produces it write it pre-split (S16-bf16) where the dgrad, the weight
 * gradient and the bias / broadcast-input sums that read it understand the format.  Same matrix operands bit for bit; the column sums see hi + lo (2^-17 relative).
 * caddy_debug_s16_grad_count: how many gradient tensors of the last caddy_loss_backward travelled pre-split. */
int caddy_debug_set_s16_grads(caddy_ctx* ctx, int on);
/* tests / A-B runs (round 6): the VGG19 perceptual loss (training/losses.py:379-491) of a training step is evaluated in `n` chunks of time steps, last steps first, on the side stream
 * beside the BPTT replay, which waits per time step for its chunk; 1 = one pass over all frames before the replay (rounds 2 - 5); 0 = the library's choice
 * (three chunks of the full-resolution level for steps of >= 4 M reconstructed pixels, four from 1 M, else one pass; CADDY_PERC_CHUNKS changes the three, a negative value forces its magnitude).  Takes effect at the next
 * caddy_forward_full.  Same loss; gradients equal up to the summation order of the per-chunk launches. */
int caddy_debug_set_perc_chunks(caddy_ctx* ctx, int n);
long caddy_debug_s16_grad_count(caddy_ctx* ctx);
int caddy_debug_set_pack_merged(caddy_ctx* ctx, int on);      /* tests: 0 = one (un)packing launch per layer and form instead of the job-table launch */
int caddy_debug_dims(caddy_ctx* ctx, int i, int* nhwc4);
int caddy_debug_get(caddy_ctx* ctx, int i, int grad, float* dst_nchw);

/* --- BatchNorm bookkeeping: number of train-mode calls of BN layer `i` since creation (num_batches_tracked) --- */
int caddy_bn_layer_count(caddy_ctx* ctx);
long caddy_bn_calls(caddy_ctx* ctx, int i, char* name_out128);

/* --- per-kernel entry points (unit-parity tests); argument structs are declared in csrc/common.h, pack.h --- */
struct ConvArgs; struct WgradArgs; struct PackDesc; struct TV;
int caddy_k_conv_fwd(const struct ConvArgs* a, void* stream);
int caddy_k_conv_wgrad(const struct WgradArgs* a, void* stream);
int caddy_k_conv_took_direct(void);      /* 1: this thread's last caddy_k_conv_fwd ran on the latency kernel (ConvArgs.direct_ok; model/main_model/model.py:570-607 batch-1 roll-out layers) */
/* BatchNorm fused with the convolutions around it (reference: the conv -> BatchNorm2d -> LeakyReLU chains of model/layers/residual_block.py:51-68,
 * same_block.py:34-47, up_block.py:31-45): per-tile partial sums from the producing conv's epilogue (ConvArgs.stats) -> finalisation without a pass over
 * the tensor; backward of a BatchNorm whose output was never materialised (ConvSrc.bn_scale / bn_shift applied by the consumer while staging) */
int caddy_k_conv_stats_tiles(void);
int caddy_k_bn_finalize_tiles(const float* part, int ntiles, int ldp, long count, const float* gamma, const float* beta, float* rmean, float* rvar, int C,
                              float* mean, float* invstd, float* scale, float* shift, void* stream);
int caddy_k_bn_bwd_lazy(const struct TV* dout, const struct TV* x, const float* mean, const float* invstd, const float* gamma, const float* scale, const float* shift, int act,
                        double* sums, double* scratch, const struct TV* dx, float* dgamma, float* dbeta, void* stream);
int caddy_k_conv_pick_bn(int cout);
/* split 16-bit operand form of a layer's weights for the 16-bit-MFMA convolution (conv_hx.hip): seg < 0 forward, else dgrad of that segment */
int caddy_k_hx_pick_bn(int cout);
int caddy_k_hx_force_big(int v);      /* tests: 1 / 0 force / forbid the 8-wave 16x16x128 tile variant, -1 automatic */
int caddy_k_hx_set_bg(int mask);   /* tests / A-B runs: which under-filled k_conv_hx tile variants read their weight fragments straight from global memory (conv_hx.hip, template
                                    * parameter BG; bit 0: 4x16x64, bit 1: 8x16x64, bit 2: 8x16x32 tiles; bit 3: also for workgroups that walk fewer than four 32-channel chunks); -1: the
                                    * CADDY_HX_BG environment variable, default 7 */
long caddy_k_hx_weight_bytes(const struct PackDesc* d, int seg, int rows_pad, int planes);
int caddy_k_pack_hx(const struct PackDesc* d, void* wq, int rows_pad, int seg, int precision, void* stream);
int caddy_k_pack_fwd(const struct PackDesc* d, float* wp, void* stream);
int caddy_k_pack_dgrad(const struct PackDesc* d, int seg, float* wpd, int Cd_pad, int Kd, void* stream);
int caddy_k_unpack_wgrad(const struct PackDesc* d, const float* dwp, void* stream);
int caddy_k_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step, float gscale, void* stream);
/* (pointwise kernels: caddy_k_copy, caddy_k_pool2[_bwd], caddy_k_up2[_bwd], caddy_k_stats, caddy_k_bn_finalize, caddy_k_bn_stats_finalize, caddy_k_bn_apply,
 *  caddy_k_bn_bwd_reduce, caddy_k_bn_bwd_apply, caddy_k_act_bwd_add, caddy_k_lstm_fwd, caddy_k_lstm_bwd, caddy_k_tanh_bwd,
 *  caddy_k_attn_mul[_bwd], caddy_k_gap[_bwd], caddy_k_colsum, caddy_k_spatial_sum, caddy_k_nchw_to_nhwc, caddy_k_nhwc_to_nchw,
 *  caddy_k_batch_sum -- see csrc/capi_kernels.cpp for the exact signatures) */

#ifdef __cplusplus
}
#endif
#endif
