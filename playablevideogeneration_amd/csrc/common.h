// Shared definitions for the CADDY hot-path HIP kernels (gfx950 / CDNA4).
//
// Data layout in HBM: every activation is NHWC fp32, addressed through a `TV` view
//   addr(n,y,x,c) = p + n*sn + (y*W + x)*ld + c          (ld = pixel stride, sn = sample stride, both in floats)
// so that channel-concatenation (reference: torch.cat along dim 1), time-slicing of (B,T,...) buffers
// (reference: tensor[:, t]) and channel-slicing (state = x[:, :-1]) are free views instead of copies.
// ld is always a multiple of 4 floats and base pointers are 16-byte aligned so float4 accesses are legal.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct TV {
    float* p;
    int N, H, W, C;
    long sn;
    int ld;
    // 1: an S16-bf16 tensor (see ConvArgs.in_s16: [hi x 32 | lo x 32] halves per 32-channel chunk, same geometry and footprint) -- round 6: the gradient of a convolution's
    // output, written ONCE in that form by its point-wise producer (BatchNorm / pooling / ConvLSTM-cell backward) and copied instead of converted by the dgrad and weight-gradient
    // launches that stage it.  Understood by the kernels that produce / read such gradients only (pointwise.hip: the functors with *_s16 in their comment); C a multiple of 32.
    // (occupies what was tail padding: the struct's size and the offsets of the other members are unchanged)
    int s16;
};

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// One input segment of a convolution.  A conv reads the channel-concatenation of up to CONV_MAX_SRC segments
// (reference: ConvLSTMCell.channelwise_concat, convolutional_lstm_cell.py:47-86).  bcast=1: the segment is a
// per-sample vector (N, C) broadcast over space (the action one-hot / variation inputs of R).
#define CONV_MAX_SRC 3
#define CONV_BK 16   // K-chunk: every segment is padded to a multiple of 16 channels in the packed weights
struct ConvSrc {
    const float* p;
    long sn;
    int ld;
    int C;      // logical channels
    int Cpad;   // round_up(C, CONV_BK)
    int bcast;
    // Lazily applied train-mode BatchNorm (+ LeakyReLU 0.2) of the PRODUCER of this segment: the consumer reads the raw conv output x and sees
    // act(x * bn_scale[c] + bn_shift[c]) -- the normalised tensor is never written to HBM (reference: conv -> BatchNorm2d -> LeakyReLU chains of
    // residual_block.py:51-68, same_block.py:34-47, up_block.py:31-45, conv_dynamics_network.py:41-45).  nullptr: plain input.  Zero padding is
    // applied AFTER the affine map (the reference pads the normalised tensor).  bn_gn > 0: per-sample-group parameters (time-batched calls whose
    // statistics stay per time step): sample n uses bn_scale + (n / bn_gn) * bn_gs.  Understood by k_conv_hx and k_wgrad_hx only.
    const float* bn_scale;
    const float* bn_shift;
    int bn_act;
    int bn_gn;
    long bn_gs;
};

struct ConvArgs {
    ConvSrc src[CONV_MAX_SRC];
    int nsrc;
    int N, H, W;        // output spatial size == (virtual) input spatial size (stride 1, pad KS/2)
    int KS;             // 1, 3 or 7
    const float* wp;    // packed weights [KS*KS][Cout_pad][Ktot]  (k contiguous)
    int Ktot;           // sum of Cpad
    int Cout, Cout_pad; // Cout_pad multiple of the N tile
    const float* bias;  // nullable, [Cout]
    int act;            // 0 none, 1 tanh, 2 ReLU, 3 LeakyReLU(0.2) (BatchNorm-folded inference convolutions)
    float* out;
    long out_sn;
    int out_ld;
    int accumulate;     // out += result
    int precision;      // 0: exact fp32 MFMA; 3: 3-way split bf16 (hi/mid/lo planes, 6 products, ~fp32 accuracy); 2: 2-way split (3 products)
    int splitk;         // set by the launcher: K range split across blockIdx.z (atomic accumulation; accumulate mode only)
    float* aux;         // optional device scratch (>= CONV_AUX_BYTES, private to the launching stream): compact weight table of the thin kernels
    float* split_scratch;   // optional: deterministic split-K of under-filled NON-accumulating launches (slabs + fixed-order reduce)
    long split_cap;         // capacity of split_scratch in floats
    long split_stride;      // set by the launcher: slab stride in floats (0 = atomics)
    // dgrad through a ReLU, fused into the epilogue (VGG19 perceptual loss, perceptual.hip): out = result * (mask > 0); with seed_ref also the
    // L1 feature-loss gradient seed of a tapped feature map: result += seed_w * sign(mask - seed_ref) before masking.  mask / seed_ref have
    // the geometry of `out` (out_sn, out_ld).
    const float* mask;
    const float* seed_ref;
    float seed_w;
    // split 16-bit operands (conv_hx.hip): weights pre-split by pack_hx for `precision` (PREC_*), rows padded to hx_pick_bn(Cout)
    const void* wq;
    int Kq;                 // set by the launcher: sum of the input segments padded to HX_KC channels
    float out_scale;        // set by the launcher: 1 / HX_WSCALE for split-f16 weights (stored pre-scaled by a power of two, see pack_hx)
    // residual input added before the activation (inference with folded BatchNorm: act(conv2'(a) + identity), residual_block.py:57-68);
    // geometry of `out` except for its own strides.  Supported by k_conv_fwd, k_conv_hx, k_conv_narrow and the split-K reduce.
    const float* res;
    long res_sn;
    int res_ld;
    int xcd_map;            // set by the launcher (conv_hx): workgroup -> (tile, channel block) order that keeps a pixel tile's channel blocks on one XCD
    // MaxPool2d(2, 2) of the activated output written by the conv epilogue (k_conv_hx only; the four pixels of a window sit in one lane's
    // accumulators): (N, H/2, W/2, Cout) with floor semantics on odd sizes.  skip_out: do not write the full-resolution output at all (a branch
    // that is never back-propagated: the ground-truth features of the VGG19 loss)
    float* pool_out;
    long pool_sn;
    int pool_ld;
    int skip_out;
    // Per-channel partial sums of the values this launch stores (k_conv_hx, EP = 0, splitk == 1 only): workgroup (pixel tile t, channel block) writes
    // stats[(t * stats_ld + c) * 2 + {0, 1}] = {sum v, sum v^2} over the valid pixels of its tile -- the BatchNorm statistics of the consumer without a
    // second pass over the conv output.  The launcher reports the number of tiles in g_last_conv_stats_tiles (0: this launch did not produce them).
    float* stats;
    int stats_ld;
    // Split-f16 forward only: |x| > 65504 does not fit the high half.  The staging clamps such a value (also an inf or a NaN) to the f16 range (instead of producing
    // inf - inf = NaN in the low half) and the launch ORs 1 into *sat_flag (nullable; | 2 when a NaN was among them: v_med3 turns it into a finite number, the reference would
    // propagate it).  The driver gives every layer its own word (round 5): caddy_f16_saturated() reads them, moves exactly the layers that reported onto a forward without a range
    // limit (exact fp32 for the model, split bf16 for VGG19) and the loss call reports CADDY_LOSS_F16_SATURATED / a NaN total.  Not reachable with BatchNorm-normalised activations;
    // it is the guard for externally supplied networks and inputs (VGG19 with real weights on un-normalised frames).
    unsigned* sat_flag;
    // Bit-reproducible mode (caddy_set_deterministic): the split-K of an under-filled ACCUMULATING launch (dgrad +=) goes through slabs of split_scratch and the
    // fixed-order reduce (which then adds the previous contents of `out`) instead of fp32 atomics in arrival order
    int deterministic;
    // Small assigning split-f16 launches may run on the latency kernel (conv_direct.hip: one launch, no slabs) instead of the tile kernel's split-K launch + slab reduce.
    // The driver sets it on the forward convolutions of inference-mode passes (roll-out, evaluation); kernel tests choose the path explicitly.
    int direct_ok;
    // Roll-out ConvLSTM cells: when the launch is split over K, its slab reduce applies the cell update itself (LstmFuse, host pointer; null otherwise) and sets
    // g_last_conv_lstm_fused -- the caller then skips its point-wise LSTM kernel.  Launch paths without a slab reduce ignore it.
    const struct LstmFuse* lstm;
    // "S16" tensors (round 5): an activation stored PRE-SPLIT for the 16-bit matrix pipe.  Same geometry and footprint as the fp32 NHWC tensor (4 bytes per element, C a
    // multiple of 32), but every pixel's 32-channel chunk holds [hi x 32 | lo x 32] 16-bit halves (hi = round16(x), lo = round16(x - hi); f16 for forward activations, bf16 for
    // gradients) -- exactly the LDS row k_conv_hx stages, so that a consumer copies it instead of converting every halo element in every output-channel block.  Understood by the
    // k_conv_hx instances conv_hx_s16_ok() selects and by the VGG19 point-wise kernels (perceptual.hip); the caller asks conv_hx_s16_ok() before it sets any of these.
    int in_s16;             // src[0] (the only segment, no lazy BatchNorm) is S16 of the launch's operand type
    int out_s16;            // `out` is written as S16 (no accumulate, no split-K)
    int pool_s16;           // `pool_out` is written as S16 (requires out_s16 semantics of the epilogue: set together with or without out_s16)
    int mask_s16, seed_s16; // `mask` / `seed_ref` are S16-f16 tensors
    // AvgPool2d(2) of the result inside the epilogue (inference with folded BatchNorm: conv -> avg_pool2d(2) -> affine -> LeakyReLU, residual_block.py:52-56 /
    // representation_network.py:39-41): `out` is the (N, H/2, W/2, Cout) map, bias / res / act are applied AFTER the pooling (H, W even).  Only the launches
    // conv_avgpool_ok() accepts (k_conv_narrow: a window's four pixels are two accumulators of one lane x a DPP neighbour); the caller asks first.
    int avgpool;
    // S16 OUTPUT range guard (SO launches, split f16): 1 = a value the epilogue had to clamp raises the word BEHIND *sat_flag -- the layer that will stage the tensor (VGG19 layer
    // i + 1: its fall-back makes this launch write fp32 again) -- instead of this layer's own word, whose fall-back would leave the oversized values where they are
    int sat_out_next;
};
int conv_avgpool_ok(const ConvArgs& a);      // (out / out_sn / out_ld need not be set yet)
// gates = [i | f | o | g] x C pre-activations (convolutional_lstm_cell.py:92-101): c' = sigm(f) c + sigm(i) tanh(g), h' = sigm(o) tanh(c'), hb = h' * scale + shift (the cell's
// eval-mode BatchNorm, conv_dynamics_network.py); all tensors NHWC with their own sample / pixel pitches
struct LstmFuse { const float* cprev; long cprev_sn; int cprev_ld; float* h; long h_sn; int h_ld; float* c; long c_sn; int c_ld; float* hb; long hb_sn; int hb_ld; const float* scale; const float* shift; int C; };
extern thread_local int g_last_conv_lstm_fused;
// split-f16 weights are stored multiplied by HX_WSCALE (exact: a power of two) and the accumulator is multiplied by 1 / HX_WSCALE in the
// epilogue: conv weights are O(1/sqrt(fan_in)) ~ 0.01-0.1, where the `lo` half (|lo| <= 2^-11 |w|) would fall into the f16 subnormal range and
// keep only ~8 of its 11 bits.  Scaled by 64 the weights are O(1) and carry 22 bits; |w| < 1023 stays finite.
#define HX_WSCALE 64.0f
// ConvArgs.precision.  0 = exact fp32 MFMA (k_conv_fwd; 2 / 3 = its in-loop split-bf16 variants, kept for A/B runs).  >= 16: conv_hx.hip --
// split f16 (hi + lo, 3 products: fp32-class accuracy, bounded operands = forward activations / weights), split bf16 (3 products: 2^-16, full
// exponent range = gradients), or single-product 16-bit operands.
enum { PREC_FP32 = 0, PREC_F16X3 = 16, PREC_BF16X3 = 17, PREC_F16X1 = 18, PREC_BF16X1 = 19 };
#define HX_KC 32
constexpr int CONV_AUX_BYTES = 128 * 1024;

// wgrad: dwp[tap][o][k] += sum_p dY[p][o] * A[p+tap][k]   (A = concatenation of the forward sources)
struct WgradArgs {
    ConvSrc src[CONV_MAX_SRC];
    int nsrc;
    int N, H, W;
    int KS;
    const float* dy;    // (N,H,W,Cout) view
    long dy_sn;
    int dy_ld;
    int Cout, Cout_pad;
    int Ktot;
    float* dwp;         // packed gradient, same layout as wp
    int slabs;          // split of the pixel reduction across blocks (atomicAdd when > 1)
    // time-batched launches: N = groups x group_n samples; sample n lives at p + (n / group_n) * gs + (n % group_n) * sn.  The weight
    // gradient of a layer is accumulated over several BPTT time steps in ONE launch, which amortises the atomic flush of the
    // persistent tile kernel (measured 21-27 % of a per-step launch).  group_n = 0: plain (N, sn) addressing.
    int group_n;
    long src_gs[CONV_MAX_SRC];
    long dy_gs;
    int precision;      // PREC_BF16X3: 3x3 layers with >= 32 channels on both sides run on the 16-bit matrix pipe (k_wgrad_hx); 0: exact fp32
    long src_bn_gs[CONV_MAX_SRC];   // time-batched launches: float distance between the (scale, shift) tables of consecutive groups of a lazily normalised source
    // Bit-reproducible mode (caddy_set_deterministic): every pixel split of the launch flushes its partial weight gradient into its OWN zero-filled copy of the packed
    // layout -- det_slab + split * det_stride -- and k_wgrad_det_reduce adds the copies to dwp in a fixed order; without it the splits meet in dwp through fp32 atomics
    // in arrival order.  det_slab: scratch of det_cap floats, private to the launching stream (nullptr: atomics); det_stride is set by the launcher.
    float* det_slab;
    long det_cap;
    long det_stride;
    int dy_s16;         // dy is an S16-bf16 tensor (TV::s16): k_wgrad_hx stages it with 16-byte copies; no other weight-gradient kernel understands it (Cout a multiple of 32)
};
// destination of a pixel split's flush (see WgradArgs.det_slab): exactly one workgroup adds to an element of a slab
#define WGRAD_DST(a_, split_) ((a_).det_slab ? (a_).det_slab + (long)(split_) * (a_).det_stride : (a_).dwp)
// launcher side: clamp the number of pixel splits to the scratch, zero-fill the slabs (returns the splits to use; <= 0: deterministic mode cannot run this launch)
long wgrad_det_begin(WgradArgs& a, long splits, hipStream_t st);
int wgrad_det_end(const WgradArgs& a, long splits, hipStream_t st);      // dwp[i] += sum over the splits, in order

// id of the kernel the last conv_*_launch on this host thread dispatched to (profiling; see CONV_KERNEL_NAMES in net.cpp)
enum { CK_FWD_128x128 = 0, CK_FWD_128x64, CK_FWD_64x64, CK_FWD_128x32, CK_THIN_OUT, CK_THIN_IN, CK_WGRAD_128, CK_WGRAD_64, CK_WGRAD_32, CK_WGRAD_SMALL, CK_WGRAD_THIN, CK_WGRAD_TILE, CK_NARROW,
       CK_HX_128, CK_HX_64, CK_HX_32, CK_WGRAD_HX, CK_HX_128_8W, CK_COUNT };      // CK_HX_128_8W: the 16x16x128 tile on 8 waves (VGG19, wide well-filled layers); CK_HX_128: 8x16x128 on 4 waves
extern thread_local int g_last_conv_kernel;
// rows a caller must provide in ConvArgs.stats: the smallest pixel tile (8 x 16) of the epilogue path, at least the 512 workgroups of the slab-reduce path
static inline long conv_stats_tiles_cap(int N, int H, int W) { long t = (long)N * cdiv(H, 8) * cdiv(W, 16); return t > 512 ? t : 512; }
extern thread_local int g_last_conv_stats_tiles;
extern thread_local int g_last_conv_direct;      // pixel tiles of the last conv_fwd_launch that wrote ConvArgs.stats (0 = none written)
bool conv_src_lazy_ok(const ConvArgs& a);             // will conv_fwd_launch run this launch on a kernel that applies ConvSrc.bn_* ?
bool wgrad_src_lazy_ok(const WgradArgs& a);           // ... conv_wgrad_launch ?
int conv_fwd_launch(const ConvArgs& a, hipStream_t st);
int conv_wgrad_launch(const WgradArgs& a, hipStream_t st);
int conv_pick_bn(int cout);
struct PackDesc;
int conv_hx_wgrad_try(const WgradArgs& a, hipStream_t st, bool dry = false);
int conv_hx_try(const ConvArgs& a, hipStream_t st, bool dry = false);      // dry: 1 = the launch would run here (nothing is launched)
bool conv_hx_s16_ok(int N, int H, int W, int Cout);       // ... on a tile variant that reads / writes S16 tensors (ConvArgs.in_s16 / out_s16)?  (the same two well-filled variants)
bool conv_hx_pool_ok(int N, int H, int W, int Cout);      // will conv_hx_try run this geometry on a tile variant with the fused max-pool epilogue?           // conv_hx.hip: 3x3 on the 16-bit MFMA with split operands (1 = handled)
int pack_hx(const PackDesc& d, void* wq, int rows_pad, int seg /* < 0: forward form, else dgrad form of that input segment */, int precision, hipStream_t st);
size_t hx_weight_bytes(const PackDesc& d, int seg, int rows_pad, int planes);
int hx_kq(const PackDesc& d, int seg);
int hx_pick_bn(int cout);
// Split 16-bit weight buffers ([tap][chunk][Cout_pad rows][planes x 32 channels], conv_hx.hip: pk_hx_elem): 16-byte piece index, inside its (tap, chunk) tile, of channels
// 8 g .. 8 g + 7 (g = 0 .. 3) of plane 0 of row `row`; plane 1 is 64 pieces further.  Fragment-major inside every 32-row block: [K half g >> 1][plane][lane = 32 (g & 1) + row & 31].
template <int NPL> __host__ __device__ __forceinline__ long hx_wq_piece(int row, int g) { return ((long)((row >> 5) * 2 + (g >> 1)) * NPL) * 64 + (g & 1) * 32 + (row & 31); }
extern int g_hx_big_override;
extern int g_hx_bg;      // conv_hx.hip: which under-filled tile variants take their weight fragments straight from global memory (bit mask; -1: CADDY_HX_BG or all)
int conv_split_reduce_pool_launch(const float* scr, long stride, int splits, int ldc, int N, int H, int W, int C, float* out, long out_sn, int out_ld, const float* bias, int act,
                                  const float* res, long res_sn, int res_ld, hipStream_t st);      // slab reduce + avg_pool2d(2) (+ bias / residual / activation at the pooled size)
int conv_split_reduce_lstm_launch(const float* scr, long stride, int splits, int ldc, int HW, long P, const float* bias, const LstmFuse& f, hipStream_t st);
int conv_split_reduce_launch(const float* scr, long stride, int splits, int ldc, int HW, long P, int C, float* out, long out_sn, int out_ld, const float* bias, int act,
                             const float* res, long res_sn, int res_ld, hipStream_t st, float* stats = nullptr, int stats_ld = 0, long stats_cap_tiles = 0, int accumulate = 0);
int conv_thin_fwd_try(const ConvArgs& a, hipStream_t st);     // conv_thin.hip: 1 = handled (thin-channel shape), 0 = not thin
int conv_head_fwd_try(const ConvArgs& a, hipStream_t st);     // conv_head.hip: 3-channel image heads (3x3 / 7x7) on the split-f16 matrix pipe (ConvArgs.precision == PREC_F16X3)
int conv_head_dgrad_try(const ConvArgs& a, hipStream_t st);    // conv_head.hip: dgrad of the 7x7 head (3 -> C channels) on the split-bf16 matrix pipe (ConvArgs.precision == PREC_BF16X3, no wq)
int conv_hx_avgpool_ok(const ConvArgs& a);
int conv1x1_lat_try(const ConvArgs& a, hipStream_t st, bool dry = false);   // conv_direct.hip: small 1x1 launches of inference passes (optionally average-pooled)
int conv_direct_try(const ConvArgs& a, hipStream_t st, bool dry = false);                 // conv_direct.hip: latency-bound 3x3 launches (batch-1 roll-out), called by conv_hx_try
int conv_stream_wgrad_try(const WgradArgs& a, hipStream_t st, bool dry = false);      // conv_stream.hip: HBM-bound 1x1 weight gradients, operands straight from global memory into the fp32 MFMA (1 = handled)
int conv_head_wgrad_try(const WgradArgs& a, hipStream_t st, bool dry = false);      // conv_stream.hip: weight gradient of the 7x7 FinalBlock head on the split-bf16 matrix pipe, taps on the M side (1 = handled)
extern thread_local int g_last_wgrad_grouped;      // 1: the kernel the last conv_*_wgrad_try picked understands time-batched arguments (WgradArgs.group_n)
int conv_narrow_wgrad_try(const WgradArgs& a, hipStream_t st, bool dry = false);
int conv_c4_wgrad_try(const WgradArgs& a, hipStream_t st, bool dry = false);   // dry: report the match without launching
int conv_c4_fwd_try(const ConvArgs& a, hipStream_t st);       // conv_narrow.hip: 3-channel (pitch 4) input, 3x3 / 7x7, on 16x16x4 MFMA
int conv_narrow_fwd_try(const ConvArgs& a, hipStream_t st);   // conv_narrow.hip: 1 = handled (3x3, 5..32 channels in, 5..32 out)
int conv_narrow_fwd_ok(const ConvArgs& a);                     // its shape test alone
int conv_thin_wgrad_try(const WgradArgs& a, hipStream_t st, bool dry = false);   // N-tile (32/64/128) the launcher will use for this Cout
