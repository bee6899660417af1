#pragma once
#include "common.h"

#define AUX_LD 16   // per-(b,t) auxiliary input row of R: [action sample (K) | variation (Da) | zero pad]

// LOSS_PERC_R0 + 6 r: perceptual_loss_r{r}; + 1 + l: perceptual_loss_r{r}_l{l} (l = 0 aliases the total, as in the reference -- perceptual.hip)
enum { LOSS_TOTAL = 0, LOSS_REC, LOSS_STATES, LOSS_ENTROPY, LOSS_DIRKL, LOSS_MI, LOSS_STATEKL, LOSS_HIDDEN, LOSS_L1_R0, LOSS_L1_R1, LOSS_L1_R2,
       LOSS_PERCEPTUAL = 11, LOSS_PERCEPTUAL_TERM = 12, LOSS_F16_SATURATED = 13, LOSS_PERC_R0 = 16, LOSS_DIAG_0 = 40, LOSS_SLOTS = 56 };
int loss_report_flag(unsigned* flags /* [live n | sticky n] */, int n, double* slot, double* total, hipStream_t st);
int loss_diff_per_frame(const TV& a, int Ta, int a_off, const TV& b, int Tb, int C, int sq, double* acc, hipStream_t st);      // evaluation: per-frame sum |a - b| / (a - b)^2      // *slot = *flag != 0 (the f16 range guard of the split-f16 forward, common.h: ConvArgs.sat_flag)

struct LossWeights { double rec, states, entropy, dir_kl, mi, state_kl, hidden, mi_entropy_lambda, perceptual; };

// linear layers of the action network (boundary layout, straight views into the flat parameter / gradient buffers)
struct HeadParams {
    int F, Da, K;
    const float *Wm, *bm, *Wv, *bv, *Wf, *bf;
    float *dWm, *dbm, *dWv, *dbv, *dWf, *dbf;
};
// per-call buffers of one action-network evaluation over (B, T)
struct HeadBufs {
    const float* feat;          // (B*T, F) pooled features
    const float *eps_s, *eps_d, *unif;   // noise: (B*T,Da), (B*(T-1),Da), (B*(T-1),K)
    float *mu, *raw;            // (B*T, Da)
    float *sdist, *ssamp;       // (B*T, 2, Da), (B*T, Da)            action_states_distribution / sampled_action_states
    float *ddist, *dirs;        // (B*(T-1), 2, Da), (B*(T-1), Da)    action_directions_distribution / sampled_action_directions
    float *logits, *logp, *prob;  // (B*(T-1), K)
    float *ysoft, *samples, *variations, *aux, *cen_used;
    long long* selected;        // (B*(T-1)) int64 arg-max
    // gradients
    float *d_feat, *d_logits, *d_ddist, *d_sdist, *d_aux, *g_dmu, *g_dvar, *g_mu, *g_raw;
};
struct SampleCfg {
    int mode;                   // 0: softmax probabilities, 1: Gumbel-softmax, 2: externally supplied samples
    int hard, training, use_variations;
    float tau, alpha;
    float* centroids;           // (K, Da) estimated_centroids (updated in place in training mode)
    const float* samples_in;    // mode 2
    const float* variations_in; // optional external variations (evaluation action_variation_sampler)
};
typedef void (*allreduce_hook_t)(float* device_ptr, int count, void* user);   // in-place sum over ranks, enqueued on the caller's stream
// evaluation samplers (evaluation/action_sampler.py:14,63, action_variation_sampler.py:14; model.py:171-190), stream-ordered callbacks on
// device pointers.  stage 0: write samples (n,K) from log_probs (n,K);  stage 1: write variations (n,Da) from sampled_dirs (n,Da) and samples.
typedef void (*sampler_hook_t)(const float* log_probs, const float* sampled_dirs, float* samples, float* variations, int n, int K, int Da, int stage, void* user);
struct SamplerHooks { sampler_hook_t fn; void* user; int action, variation; float *samples_buf, *var_buf; };

struct SmallLossArgs {
    int K, Da, NS, NT;
    float* Pbuf; float mi_grad_scale;   // joint matrix scratch (K*K), world size
    const float *p, *q, *logp;  // softmax(logits), softmax(reconstructed logits), log_softmax(logits)   (NS, K)
    const float *ddist, *sdist, *sdist_r;
    float *d_logits, *d_logits_r, *d_ddist, *d_sdist_r;
    float* ema; float ema_alpha; int update_ema;   // SmoothMutualInformationLoss state (K,K) or null
    float mi_lamb, w_mi, w_entropy, w_dirkl, w_statekl;
    double* acc;
};

int head_rollout_in(const float* obs, float* o_nhwc, int HW, int C, int ld, float* aux, int action, const float* variation, int K, int Da, hipStream_t st);
int head_rollout_out(const float* f_nhwc, int fld, const float* obs, float* frame_out, float* obs_out, int HW, int C, hipStream_t st);
int head_softmax(const float* logits, float* prob, float* logp, int NS, int K, hipStream_t st);
int head_forward(const HeadBufs& h, const HeadParams& p, int B, int T, hipStream_t st);
int head_sample(const HeadBufs& h, const HeadParams& p, const SampleCfg& c, int NS, float* cen_sums, allreduce_hook_t hook, void* user, const SamplerHooks* sh, hipStream_t st);
int head_backward(const HeadBufs& h, const HeadParams& p, const SampleCfg& c, int B, int T, int first_call, hipStream_t st);
int loss_l1(const TV& gt, const TV& rec, const TV& drec, int f, int t_off, int Tobs, int Trec, float gscale, double* acc, float* gt_out /* nullable: resized ground truth (N,H,W,3) pitch 4 */, hipStream_t st);
int loss_mse(const TV& a, const TV& b, const TV& db, float gscale, double* acc, hipStream_t st);
int loss_small(const SmallLossArgs& a, allreduce_hook_t hook, void* user, hipStream_t st);
// logging-only scalars of the reference's loss_info (trainer.py:475-491) into acc[LOSS_DIAG_0 ...]; a = first A call, r = second (reconstructed states)
struct DiagArgs { const float *samples, *ddist, *rdist, *variations, *centroids; int NS, K, Da; double* acc; };
int loss_diagnostics(const DiagArgs& d, const TV& states, const TV& hidden, hipStream_t st);
struct VggLevels;
int loss_finalize(double* acc, const LossWeights& w, double n0, double n1, double n2, double nstates, double nhidden, const VggLevels* lv /* nullable */, hipStream_t st);
