// Host-side driver of the CADDY hot path: owns the packed weights, the activation / gradient arenas carved out of the
// caller's workspace, and a tape that replays the forward graph in reverse for BPTT.
#pragma once
#include <algorithm>
#include <array>
#include <deque>
#include <functional>
#include <string>
#include <vector>
#include "caddy_hip.h"
#include "common.h"
#include "head.h"
#include "pack.h"
#include "pointwise.h"
#include "perceptual.h"

// Format of a convolution output's GRADIENT (round 6), shared by every copy of its T4: `elig` is decided when the forward graph is built -- the gradient has one assigning writer and
// every reader (dgrad, weight gradient, bias / broadcast-input sums) understands S16-bf16 (TV::s16) --, `fmt` by that writer when the tape is replayed: a point-wise producer that
// knows the format (gv_w) writes the halves the matrix pipe will multiply, any other writer leaves fp32 and the readers convert as before.
struct GradFmt { bool elig; int fmt; };
struct T4 {            // activation + gradient views with identical geometry
    float* d; float* g;
    int N, H, W, C; long sn; int ld;
    bool nz = false;   // gradient lives in the NOT-zero-filled part of the arena: its single backward writer assigns (see Arena::alloc_top)
    bool nz2 = false;  // ... with TWO backward writers: a point-wise one that runs first in the reverse replay (residual add, bilinear up-sampling) and
                       // assigns, and a convolution dgrad that runs later and accumulates
    // lazily normalised tensor: `d` holds the RAW conv output x, the logical value is act(x * bn_scale[c] + bn_shift[c]) and is only ever formed inside the
    // consuming convolution's staging (ConvSrc.bn_*); `g` is the gradient w.r.t. the logical (normalised) value
    const float* bn_scale = nullptr; const float* bn_shift = nullptr; int bn_act = 0;
    int fmt = 0;       // 1: `d` is an S16 tensor (common.h: pre-split 16-bit halves, same geometry / footprint) -- VGG19 feature maps only (perceptual.hip)
    GradFmt* gs = nullptr;      // non-null: `g` may be written pre-split (see GradFmt); owned by caddy_ctx::gfmts, valid until the next forward pass
};
static inline TV dv(const T4& t) { return TV{t.d, t.N, t.H, t.W, t.C, t.sn, t.ld, 0}; }
static inline TV gv(const T4& t) { return TV{t.g, t.N, t.H, t.W, t.C, t.sn, t.ld, t.gs ? t.gs->fmt : 0}; }      // as its writer left it
// destination view for THE assigning writer of the whole gradient (`assigns`: it overwrites every element, no read-modify-write): pre-split when the consumer asked for it
static inline TV gv_w(const T4& t, bool assigns) {
    const int f = (t.gs && t.gs->elig && assigns) ? 1 : 0;
    if (t.gs) t.gs->fmt = f;
    return TV{t.g, t.N, t.H, t.W, t.C, t.sn, t.ld, f};
}

struct ParamEntry { std::string name; long offset; int ndim; int shape[4]; int kind; long numel; };

// Two-ended bump allocator.  Bottom-up: ordinary activations; their gradient mirror [0, off) is zero-filled before every backward pass
// because backward ops accumulate (+=) into it.  Top-down (alloc_top): tensors whose gradient has exactly ONE writer that overwrites it
// completely before anyone reads it (conv outputs consumed by a BatchNorm, pooled tensors, ConvLSTM gate pre-activations): no zero-fill
// and no read-modify-write for ~40 % of the gradient bytes.
struct Arena {
    char* base = nullptr; size_t cap = 0, off = 0, high = 0, top = 0, top_used = 0;
    void* alloc(size_t bytes) {
        size_t a = (off + 255) & ~(size_t)255;
        off = a + bytes;
        if (off > high) high = off;
        return base + a;
    }
    void* alloc_top(size_t bytes) {
        top = (top - bytes) & ~(size_t)255;
        if (cap - top > top_used) top_used = cap - top;
        return base + top;
    }
    void reset() { off = 0; top = cap & ~(size_t)255; }
    bool overflow() const { return off > top; }
};

struct BNL;
#define CADDY_N_FLAGS 128
#define CADDY_VGG_FLAG0 96
struct ConvL {
    PackDesc pd{};
    int flag_idx = 0;                // index of this layer's f16-range-guard word (caddy_ctx::sat_flag) / fallback switch
    float* wp = nullptr; float* dwp = nullptr;
    bool early_bucket = false, early_done = false;   // member of the R / D gradient buckets that are final after the time loop's backward
    float* wpd[CONV_MAX_SRC] = {nullptr, nullptr, nullptr};
    int cd_pad[CONV_MAX_SRC] = {0, 0, 0};
    int kd = 0;
    const float* bias = nullptr; float* dbias = nullptr;
    size_t wp_floats = 0;
    // split 16-bit operand forms for conv_hx.hip (3x3 layers with >= 32 channels on both sides): forward weights as f16 hi|lo, dgrad weights
    // (flipped / transposed) as bf16 hi|lo; re-split once per optimiser step by pack_all()
    void* wq = nullptr; void* wqd[CONV_MAX_SRC] = {nullptr, nullptr, nullptr};
    // roll-out: the eval-mode BatchNorm that follows this conv (with nothing but an average pool in between) is folded into the packed forward
    // weights (PackDesc.oscale) and into fold_bias = bias * scale + shift by pack_all(true)
    struct BNL* fold_bn = nullptr; float* fold_bias = nullptr;
    // off-chain dgrad of this layer (Seg::off_chain: d(h_{t-1}) of a ConvLSTM gate convolution on the decoder stream): event behind the last launch
    hipEvent_t off_ev = nullptr; bool off_pending = false;
};
struct BNL { std::string name; float *gamma, *beta, *dgamma, *dbeta, *rmean, *rvar; int C; long calls = 0;
             // D's BatchNorms are replayed on TWO streams in the backward (teacher-forced steps on the decoder stream beside the closed-loop steps on the main stream): the
             // decoder stream accumulates into private (dgamma_d, dbeta_d), folded into the flat gradient buffer after the join -- no read-modify-write race, fixed order
             float *dgamma_d = nullptr, *dbeta_d = nullptr;
             float* eval_stash = nullptr; bool eval_valid = false;
             // deferred running-statistics update (D's BatchNorms: their calls execute on two streams): every train-mode call leaves (mean, unbiased variance) in its
             // stash and the momentum updates are applied in CALL order by one kernel at the end of the forward
             bool deferred = false; std::vector<std::pair<const float*, const float*>> pend; };   // eval mode: (mean, invstd, scale, shift) from the running statistics, computed once per start_inference / eval forward
struct ResL { ConvL conv1, conv2, down; BNL bn1, bn2, bnd; bool has_down = false; int ds = 1; };
struct LstmL { ConvL gates; BNL bn; float *init_h, *init_c, *ginit_h, *ginit_c;   // boundary (C,h,w) params + grads
               T4 ih, ic;        // HWC copies (1,h,w,C) in the persistent arena (data + grad)
               T4 h, c;          // current state
               T4 ph, pc;        // persistent inference state (B,h,w,C)
               int C, Hs, Ws; };
struct Seg { T4 t; int bcast; bool need_grad; bool off_chain = false; };      // off_chain: the gradient of this input is not read before the previous time step's backward (h_{t-1} of a ConvLSTM): its dgrad leaves the BPTT chain (decoder stream)

struct HeadState { HeadBufs b{}; SampleCfg sc{}; T4 x65; T4 att; };

struct caddy_ctx {
    caddy_config cfg{};
    bool dry = false;
    bool fail = false;
    bool training = true;
    const struct LstmFuse* lstm_fuse = nullptr;      // lstm_step -> conv(): the gate convolution of a roll-out ConvLSTM cell may apply the cell update in its slab reduce
    bool recording = false;
    hipStream_t stream = nullptr;
    float* P = nullptr; float* G = nullptr;
    std::vector<ParamEntry> table;
    long n_floats = 0, n_train = 0;
    Arena persist, act;
    size_t grad_delta = 0;           // byte distance between an activation and its gradient
    std::vector<std::function<void()>> tape;
    // ---- the decoder of the teacher-forced time steps on its own stream ----
    // For t + 1 < gt_init nothing feeds D(t)'s frames back into the model (model.py:241-243): D(t) only depends on R(t), and in the backward pass D-bwd(t) only on
    // the loss seeds.  Those decoder calls (a third of D's work at gt_init = 6, T = 16) run on `dstream`, beside the serial R -> D -> E chain of the closed-loop
    // steps whose small grids leave most of the chip idle: forward forked after R(t), joined at the end of the forward; backward forked right after the loss
    // kernels, joined before R-bwd of the last teacher-forced step.  Their tape entries live in `tape2`; the stream-private scratch buffers are swapped with
    // the stream (enter_d / leave_d).  BatchNorm running statistics of D are applied in call order at the end of the forward (BNL::deferred), whatever the
    // execution order of the two streams was.
    std::vector<std::function<void()>> tape2;
    std::vector<std::function<void()>>* tp = &tape;      // tape the ops currently record into
    hipStream_t dstream = nullptr; bool use_dstream = true, in_d = false, d_forked = false; hipEvent_t d_done = nullptr;
    struct StreamRes { hipStream_t st; float* aux; float* split; double* red; } dsr{}, dsr_saved{};
    void enter_d(bool fork);
    void leave_d();
    void replay_tape2(bool concurrent);
    void fold_d_bn_grads();      // after the join of the decoder stream's backward: G += the private BatchNorm parameter gradients (BNL::dgamma_d)
    float* bn_dgamma(BNL* b) const { return in_d && b->dgamma_d ? b->dgamma_d : b->dgamma; }
    float* bn_dbeta(BNL* b) const { return in_d && b->dbeta_d ? b->dbeta_d : b->dbeta; }
    void end_forward();
    bool tape2_done = false;
    std::vector<T4> dbg;             // every alloc() of the current forward (debug introspection, caddy_debug_*)
    std::deque<GradFmt> gfmts;       // gradient formats of this forward's convolution outputs (T4::gs points here: a deque keeps the addresses stable)
    bool mask_from_x = true;         // BatchNorm backward of a single-input BatchNorm + LeakyReLU takes the slope from x * scale + shift instead of reading the output (CADDY_MASK_FROM_X=0: round-5 form)
    bool s16_grads = true;           // conv-output gradients pre-split by their producers where every reader understands it (CADDY_S16_GRADS=0: fp32 everywhere, the round-5 exchange)
    std::vector<ConvL*> convs;
    std::vector<BNL*> bns;
    // layers
    ConvL e_stem; BNL e_bn1; ResL e_res[6];
    ResL a_res[CADDY_MAX_ENSEMBLE][2]; HeadParams hp[CADDY_MAX_ENSEMBLE]{};      // action_network.{m}: the ensemble members (model.py:47); one is drawn per forward pass (model.py:152)
    int n_members = 1, member = 0;      // member of the current / next forward pass (caddy_set_action_member); both A calls of a pass use it
    long member_lo[CADDY_MAX_ENSEMBLE] = {0}, member_hi[CADDY_MAX_ENSEMBLE] = {0};      // trainable ranges of the members in the flat buffers (Adam skips the members that were not drawn)
    LstmL lstm[3]; ConvL r_c0, r_c1, r_c2; BNL r_bn0, r_bn1, r_bn2;
    ConvL d_up[3]; BNL d_norm[3]; ResL d_res[2]; ConvL d_final[3];
    ConvL s2h;
    float* centroids = nullptr;
    // per-forward state
    int gt_init = 0; float tau = 1.f;
    T4 obs, x65_gt, rec_x65, hidden, frames[3], attn_gt, rec_hidden;
    bool pretraining = false;         // last forward was forward_pretraining (model.py:290-468)
    HeadState head1, head2;
    float *q_prob = nullptr;
    double* loss_acc = nullptr;
    char* zero_pool = nullptr; size_t zero_pool_bytes = 0;      // see build_layers
    allreduce_hook_t hook = nullptr; void* hook_user = nullptr; int world = 1;
    // bucketed gradient all-reduce: R's and D's parameter ranges are final once the time loop's backward is done; they are
    // unpacked on the side stream and handed to the caller while A and E-on-ground-truth-frames still run their backward
    void (*grads_hook)(float* grads, long offset, long count, void* stream, void* user) = nullptr; void* grads_user = nullptr;
    long bucket_lo[2] = {0, 0}, bucket_hi[2] = {0, 0}; bool lstm_early_done = false;
    long s2h_lo = 0, s2h_hi = 0;     // state_to_hidden_state_layer: unused by forward_full_model -> its .grad is None there and torch's Adam skips it (no weight decay either)
    void early_gradient_buckets();   // data-parallel reductions of the K x K MI matrix / centroid sums
    // native data parallelism (dp_rccl.cpp): a communicator owned by the context; the hooks above then point at ncclAllReduce wrappers
    void* comm = nullptr; void* comm2 = nullptr;      // comm: everything on the context's stream; comm2: the gradient buckets on the side stream (one communicator per stream)
    int comm_world = 1; std::vector<std::pair<long, long>> comm_buckets; hipStream_t comm_bucket_stream = nullptr;
    SamplerHooks samplers{};         // evaluation action / variation samplers (caddy_set_sampler_hook)
    // caddy_set_deterministic: the backward pass is bit-reproducible -- split-K dgrads through slabs + fixed-order reduce instead of fp32 atomics, every pixel split of a
    // weight-gradient launch into its own copy of the packed layout + fixed-order reduce (WgradArgs.det_slab), single-workgroup bias sums
    bool deterministic = true;       // (round 5: the default -- the mode costs < 1 % of the step since its single-workgroup bias sums and serial folds are gone; 0 = arrival-order atomics)
    float* wgrad_det = nullptr; long wgrad_det_cap = 0;      // scratch of the deterministic weight gradients (the stream the weight gradients run on)
    hipStream_t wgrad_det_owner = nullptr; bool wgrad_det_owner_set = false;      // ... and the stream that uses it (a second stream takes it over behind an event: launch_conv_wgrad)
    long n_pool_fallback = 0;                                                      // pooled convolution launches that declined at run time and ran as conv + pool2 (conv())
    long n_wgrad_det_handover = 0;                                                 // such hand-overs since creation (tests)
    // f16 range guard of the split-f16 forward (ConvArgs.sat_flag), per LAYER (round 5): word i belongs to convs[i] (model) / CADDY_VGG_FLAG0 + i (VGG19 conv i); ORed by any staging
    // thread that met |x| > 65504 (| 2: a NaN); sticky on the device until caddy_f16_saturated() polls them, which also moves the reporting layers -- and only those -- onto a forward
    // without a range limit (layer_fallback: exact fp32 for model layers, split bf16 for VGG19; sticky for the context's lifetime)
    unsigned* sat_flag = nullptr;
    bool layer_fallback[CADDY_N_FLAGS] = {false};
    int n_fallback = 0;
    float* conv_split = nullptr; long conv_split_cap = 0;   // slabs of the deterministic forward split-K (main stream only)
    float* conv_aux = nullptr;       // CONV_AUX_BYTES scratch of the thin-channel conv kernels (main stream only)
    float* conv_aux2 = nullptr;      // ... of the VGG19 levels that run on the side stream (perceptual.hip)
    double* red_scratch = nullptr;   // per-block partial sums of the BatchNorm reductions (RED_MAX_BLOCKS x 2 x 1024 doubles)
    VggState vgg;                    // VGG19 perceptual loss (perceptual.hip); enabled by caddy_config.perceptual
    int vgg_precision = PREC_F16X3, vgg_precision_bwd = PREC_BF16X3;   // ConvArgs.precision of the VGG convolutions (forward / dgrad); PREC_FP32 = exact
    bool vgg_s16 = true;             // VGG19 feature maps / feature gradients of well-filled layers as S16 tensors (caddy_debug_set_vgg_s16; CADDY_VGG_S16=0)
    int prec_fwd = PREC_F16X3, prec_bwd = PREC_BF16X3;                 // ... of the model's wide 3x3 convolutions (caddy_set_precision; CADDY_PRECISION=exact)
    // ground-truth VGG19 branch overlapped with the forward pass on the side stream (perceptual.hip: vgg_gt_prefetch)
    T4 gt_img[3]{}; size_t gt_scratch_off = 0, gt_scratch_end = 0; bool gt_prefetched = false, perc_prefetch = true; hipEvent_t gt_done = nullptr;
    // Time-chunked perceptual pass (round 6; perceptual.hip, net.cpp: loss_backward).  The reconstructed frames are handed to VGG19 in chunks of time steps, LAST steps first, on the
    // side stream; chunk k covers t in [perc_t0[k + 1], perc_t0[k]) of the Trec reconstructed frames of every sample.  The BPTT replay on the main stream waits per time step for the
    // event of the chunk that holds that step's frames (perc_wait): the latency-bound BPTT chain of the late steps runs beside the throughput-bound VGG19 work of the early ones
    // instead of behind all of it.  perc_nch = 1: the one-pass form of rounds 2 - 5.  The ground-truth taps of a forward pass are laid out per chunk (gt_taps_c).
    static constexpr int PERC_MAX_CHUNKS = 8;
    int perc_chunks_cfg = 3;         // requested chunks of the full-resolution level (CADDY_PERC_CHUNKS, caddy_debug_set_perc_chunks; 1 = one pass) for steps of >= 4 M reconstructed pixels (four from 1 M)
    int perc_chunks_force = 0;       // > 0: caddy_debug_set_perc_chunks -- that many chunks whatever the size of the step
    int perc_nch = 1, perc_trec = 0; // chunk table of the current forward pass
    int perc_t0[PERC_MAX_CHUNKS + 1] = {};
    hipEvent_t perc_ev[PERC_MAX_CHUNKS] = {};
    bool perc_waited[PERC_MAX_CHUNKS] = {};
    bool perc_pipelined = false;     // this loss_backward runs the chunks on the side stream beside the tape replay
    T4 gt_taps_c[PERC_MAX_CHUNKS][3][5]{};
    // time range [t0, t0 + len) of resolution level r handled with chunk k (false: nothing).  The full-resolution level follows the chunk table; the half- and quarter-resolution
    // levels (a quarter of the work, launches that under-fill the chip at 120 frames) are cut ONCE, at the boundary in front of chunk nch / 2: the late half runs with the first
    // chunk (beside its full-resolution level), the early half in front of the full-resolution level of chunk nch / 2 -- whose event then covers it
    bool perc_range(int r, int k, int* t0, int* len) const {
        if (r == 0 || perc_nch == 1) { if (r != 0 && k != 0) return false; *t0 = r == 0 ? perc_t0[k + 1] : 0; *len = r == 0 ? perc_t0[k] - perc_t0[k + 1] : perc_trec; return true; }
        const int kb = perc_nch / 2;
        if (k == 0) { *t0 = perc_t0[kb]; *len = perc_trec - perc_t0[kb]; return true; }
        if (k == kb) { *t0 = 0; *len = perc_t0[kb]; return true; }
        return false;
    }
    void perc_plan(int Trec, bool chunked);
    void perc_wait(int t);
    size_t gt_lo = 0, gt_hi = 0;     // [gt_lo, gt_hi) of the activation arena: ground-truth VGG19 taps + scratch of vgg_gt_prefetch -- never back-propagated, so their gradient mirror is not zero-filled
    size_t fwd_off = 0;              // act.off at the end of the last forward: loss_backward allocates its VGG buffers past it and releases them
    int prof_kind_override = -1;     // profiling: record kind (3 = VGG forward, 4 = VGG dgrad) instead of 0 / 1
    // roll-out (generate_next) as ONE graph launch per frame: static input / output / action buffers, the per-frame kernel sequence captured on
    // an internal stream after start_inference and replayed afterwards (host cost of ~90 launches -> 1); eager fallback when capture fails
    float* inf_aux = nullptr;
    T4 roll_frame{};                 // full-resolution frame (NHWC, pitch 4) of the per-frame kernel sequence
    hipStream_t gstream = nullptr; hipGraph_t graph = nullptr; hipGraphExec_t graph_exec = nullptr;
    bool graph_valid = false, graph_failed = false, use_graph = true;
    hipEvent_t gev_in = nullptr, gev_out = nullptr;
    void drop_graph();
    // BatchNorm folding for the roll-out (weights are constant between start_inference calls): `fold` switches encode / dynamics / render to the
    // folded graph (conv' + LeakyReLU / residual epilogues, no BatchNorm launches), `packed_fold` says which form the packed weights hold
    bool use_fold = true, fold = false, packed_fold = false, rollout = false;
    void prepare_inference_weights();
    bool have_forward = false;
    bool seeds_only = false;         // caddy_debug_set_seeds_only: caddy_loss_backward stops after the loss kernels (tests of the loss gradient seeds)
    bool poison_nz = false;          // caddy_debug_set_poison: NaN-fill the first-touch gradient region before every backward (tests)
    int hs, ws;   // state resolution

    // ---- optional per-launch timing of the conv kernels (HIP events on the launch stream; bench.py roofline) ----
    struct ProfRec { hipEvent_t a, b; int fam; double flops; int P, K, Cout, KS, kind; double bytes; };   // kind: 0 fwd, 1 dgrad (accumulate), 2 wgrad
    bool prof = false;
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
    hipEvent_t ev() { if (ev_used == ev_pool.size()) { hipEvent_t e; hipEventCreate(&e); ev_pool.push_back(e); } return ev_pool[ev_used++]; }
    // phase marks on the main stream while profiling (caddy_profile_phases): where the wall time of a step goes
    struct PhaseMark { const char* name; hipEvent_t e; };
    std::vector<PhaseMark> phases;
    void mark(const char* name) { if (prof && !dry && !in_d) { hipEvent_t e = ev(); hipEventRecord(e, stream); phases.push_back({name, e}); } }
    int timed_conv_fwd(const ConvArgs& a, double flops);
    int timed_conv_wgrad(const WgradArgs& a, double flops);
    // weight gradients run on a side stream, off the BPTT critical path (dgrad chain); joined before unpack_all()
    hipStream_t side = nullptr; bool use_side = true; bool side_ready = false;
    std::vector<hipEvent_t> sev_pool; size_t sev_used = 0;
    hipEvent_t sev() { if (sev_used == sev_pool.size()) { hipEvent_t e; hipEventCreate(&e); sev_pool.push_back(e); } return sev_pool[sev_used++]; }
    hipStream_t wgrad_stream();
    // gradients that nothing in the BPTT chain reads -- those of the broadcast action / variation inputs of R's convolutions (consumed by the action network's
    // backward after the time loop) and the conv bias gradients (consumed by the optimiser) -- leave the critical path: onto the decoder stream (dstream), ordered
    // behind the compute stream per call, joined before the action network's backward / the gradient buckets / the final unpack
    bool a_dirty = false;
    void ensure_dstream();
    hipStream_t aux_grad_stream();
    void join_aux(hipStream_t onto);
    void ensure_side();
    // weight gradients of a layer are queued over consecutive BPTT time steps and launched as ONE time-batched kernel (WgradArgs.group_n)
    struct PendingW { WgradArgs first{}; int count = 0; long src_gs[CONV_MAX_SRC] = {0, 0, 0}; long dy_gs = 0; double flops = 0;
                      const float* last_src[CONV_MAX_SRC] = {nullptr, nullptr, nullptr}; const float* last_dy = nullptr;
                      long bn_gs[CONV_MAX_SRC] = {0, 0, 0}; const float* last_bn[CONV_MAX_SRC] = {nullptr, nullptr, nullptr}; };      // (scale, shift) tables of lazily normalised sources: one per time step
    std::vector<std::pair<ConvL*, PendingW>> pending;
    void queue_wgrad(ConvL* L, const WgradArgs& w, double flops);
    void flush_wgrad(PendingW& p);
    void flush_all_wgrad();

    // ---- helpers ----
    T4 alloc(int N, int H, int W, int C, int ld = 0);
    T4 alloc_nz(int N, int H, int W, int C, bool second_writer_is_conv = false);      // gradient not zero-filled: single assigning backward writer (or T4::nz2)
    float* falloc(size_t n);
    double* dalloc(size_t n);
    T4 conv(ConvL& L, const Seg* segs, int nseg, int act, const T4* into, bool nz_out = false, const T4* res = nullptr);
    // BatchNorm fusion (train mode).  (1) `want_stats`: set before conv() when a BatchNorm consumes the output directly: the conv epilogue leaves per-tile partial
    // sums behind (ConvArgs.stats) and bn_forward() finalises from them instead of re-reading the tensor.  (2) bn_act(..., lazy_for): the single consumer is a
    // convolution that applies scale / shift / LeakyReLU while staging its input, so the normalised tensor is never written (T4::bn_*).
    bool want_stats = false;
    int pool_fuse = 0;      // set by conv_pool() for the next conv(): 1 = average-pool the result, 2 = and apply LeakyReLU (cleared by conv() when its launch took the pooled epilogue)
    struct TileStats { const float* x = nullptr; float* part = nullptr; int tiles = 0, ldp = 0; };
    TileStats stats_ring[2]; int stats_next = 0;      // the two most recent producers (a residual block's conv2 and its 1x1 down-sampling conv feed one bn_act call)
    const TileStats* find_stats(const float* x) const { for (const TileStats& t : stats_ring) if (t.x == x && t.tiles > 0) return &t; return nullptr; }
    bool lazy_bn = true, epi_stats = true, bn_small = true;      // test switches (caddy_debug_set_bn_paths)
    long n_bn_lazy = 0, n_bn_tile_stats = 0, n_bn_calls = 0;      // since creation: BatchNorm calls applied lazily / finalised from conv-epilogue partial sums / all train-mode calls (caddy_debug_fusion_counts)
    bool lazy_ok(const ConvL& consumer, const T4& x) const;
    T4 pool2(const T4& x, bool act = false);
    T4 conv_pool(ConvL& L, const Seg* segs, int nseg, bool act);
    T4 up2(const T4& x);
    T4 bn_act(const T4& x, BNL& bn, const T4* x2, BNL* bn2, bool act, const T4* into, bool nz_out = false, bool nz2_out = false,
              const ConvL* lazy_for = nullptr);   // nz_out: the output feeds exactly one conv; nz2_out: T4::nz2; lazy_for: that one conv (same H, W) -- candidate for the lazily applied form
    T4 resblock(ResL& R, const T4& x, const T4* into, bool nz2_out = false);
    T4 encode(const T4& obs_in, bool input_grad, const T4* into);
    T4 lstm_step(int i, const T4& x, const T4& aux, const ConvL* next = nullptr);      // next: the convolution that consumes the cell's BatchNorm output
    T4 dynamics(const T4& state, const T4& aux, const T4* into);
    void render(const T4& hdn, int slot, int nslots);
    void action_net(const T4& x65, HeadState& hs_, const float* eps_s, const float* eps_d, const float* unif, bool first,
                    const float* samples_in, const float* variations_in);
    void copy_op(const T4& src, const T4& dst);
    void alloc_gt_images(T4* gi, int Trec);
    void pack_all(bool fold = false);
    void unpack_all();
    // one-launch (un)packing (pack.h: PackJob): job tables in the persistent arena, rebuilt when the precision / recording mode changes
    struct JobList { std::vector<PackJob> host; PackJob* dev = nullptr; int cap = 0, blocks = 0, key = -1; };
    JobList pack_jobs, unpack_jobs[3];      // unpack: 0 every layer, 1 the early-bucket layers (R / D), 2 the rest
    void add_job(JobList& jl, const PackDesc& d, void* buf, int kind, int seg, int p0, int p1, long total);
    void upload_jobs(JobList& jl, hipStream_t st);
    bool merged_pack = true;
    int fork_batch = 4;      // weight-gradient launches / auxiliary-gradient jobs handed to the other stream per fork (measured 1 / 4 / 8 / 16: profiles/r03_experiments.md)
    std::vector<std::pair<WgradArgs, double>> wgrad_jobs; std::vector<std::function<void()>> aux_jobs;
    void launch_wgrad_jobs(); int launch_conv_wgrad(const WgradArgs& a, double flops, hipStream_t st);
    void defer_aux(std::function<void()> job); void flush_aux(); void step_boundary();
    bool aux_enabled();
    void ck(int rc, const char* what);
};

void build_param_table(const caddy_config& c, std::vector<ParamEntry>& t, long* n_floats, long* n_train);
void set_error(const std::string& s);
bool caddy_serial_streams();      // CADDY_STREAMS=0 (net.cpp)
