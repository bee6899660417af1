// Host launchers of the element-wise / reduction kernels (pointwise.hip) and of the packing / head kernels.
#pragma once
#include "common.h"

int pw_copy(const TV& s, const TV& d, int acc, hipStream_t st);
int pw_fill(const TV& d, float v, hipStream_t st);
int pw_pool2(const TV& in, const TV& out, hipStream_t st, int act = 0);      // act: LeakyReLU(0.2) after pooling (BatchNorm-folded roll-out)
int pw_pool2_bwd(const TV& dout, const TV& din, int assign, hipStream_t st);   // assign: din = ... (first and only writer) instead of din += ...
int pw_up2(const TV& in, const TV& out, hipStream_t st);
int pw_up2_bwd(const TV& dout, const TV& din, hipStream_t st, int assign = 0);      // assign: din = ... (first writer of a T4::nz2 gradient)
#define RED_MAX_BLOCKS 512   // per-block partial sums: scratch = RED_MAX_BLOCKS * 2 * C doubles (C <= 1024)
int pw_stats(const TV& x, double* sums, double* scratch, hipStream_t st);
int pw_bn_finalize(const double* sums, long count, const float* gamma, const float* beta, float* rmean, float* rvar, int C, int training,
                   float* mean, float* invstd, float* scale, float* shift, hipStream_t st);
int pw_bn_stats_finalize(const TV& x, double* sums, double* scratch, const float* gamma, const float* beta, float* rmean, float* rvar,
                         float* mean, float* invstd, float* scale, float* shift, hipStream_t st);
bool pw_bn_small_ok(const TV& x);
bool pw_bn_small_pays(const TV& x);
int pw_bn_small_fwd(const TV& x, const float* gamma, const float* beta, float* rmean, float* rvar, float* mean, float* invstd, float* scale, float* shift,
                    const TV* x2, int act, const TV& out, hipStream_t st);
int pw_bn_small_bwd(const TV& dout, const TV* outm, const TV& x, const float* mean, const float* invstd, const float* gamma, const TV& dx,
                    float* dgamma, float* dbeta, const TV* dres, int assign, hipStream_t st, int res_assign = 0)      /* res_assign: dres = ... (first writer of a T4::nz2 gradient) */;
int pw_bn_apply(const TV& x, const float* scale, const float* shift, const TV* x2, const float* scale2, const float* shift2, int act, const TV& out, hipStream_t st);
// lazy_scale / lazy_shift: the BatchNorm output was never materialised (ConvSrc.bn_*): the LeakyReLU slope comes from x * scale + shift instead of `outm`
int pw_bn_bwd_reduce(const TV& dout, const TV* outm, const TV& x, const float* mean, const float* invstd, double* sums, double* scratch, float* dgamma, float* dbeta, hipStream_t st,
                     const float* lazy_scale = nullptr, const float* lazy_shift = nullptr);
int pw_bn_bwd_apply(const TV& dout, const TV* outm, const TV& x, const float* mean, const float* invstd, const float* gamma, const double* sums,
                    const TV& dx, float* dgamma, float* dbeta, int assign /* dx = ... instead of += : dx has no other writer */, hipStream_t st,
                    const float* lazy_scale = nullptr, const float* lazy_shift = nullptr);
// train-mode statistics from the per-tile partial sums of the producing convolution's epilogue (ConvArgs.stats) + finalisation: no pass over the tensor
// rmean == nullptr in any of the train-mode finalisations above / below: deferred running statistics -- `rvar` then receives the unbiased variance of this call
// and pw_bn_ema applies the momentum updates of `n` calls in order
int pw_bn_ema(const float* const* means, const float* const* uvars, int n, int C, float* rmean, float* rvar, hipStream_t st);
int pw_bn_finalize_tiles(const float* part, int ntiles, int ldp, long count, const float* gamma, const float* beta, float* rmean, float* rvar, int C,
                         float* mean, float* invstd, float* scale, float* shift, hipStream_t st);
int pw_act_bwd_add(const TV& dout, const TV& outm, const TV& dres, hipStream_t st, int assign = 0);
int pw_lstm_fwd(const TV& gates, const TV& cprev, const TV& h, const TV& cn, hipStream_t st, const TV* hb = nullptr, const float* scale = nullptr, const float* shift = nullptr);
int pw_lstm_bwd(const TV& gates, const TV& cprev, const TV& cn, const TV& dh, const TV& dc, const TV& dgates, const TV& dcprev, hipStream_t st);
int pw_tanh_bwd(const TV& dy, const TV& y, const TV& dz, hipStream_t st);
int pw_attn_mul(const TV& x, const TV& out, const TV& att, hipStream_t st);
int pw_attn_mul_bwd(const TV& x, const TV& dout, const TV& datt, const TV& dx, hipStream_t st);
int pw_gap(const TV& x, float* out, hipStream_t st);
int pw_gap_bwd(const float* dout, const TV& dx, hipStream_t st);
int pw_colsum(const TV& x, float* out, hipStream_t st, bool det = false, double* scratch = nullptr);                     // det: single workgroup (bit-reproducible: no float atomics between workgroups)
int pw_spatial_sum(const TV& x, float* out, long out_sn, hipStream_t st, bool det = false);   // det: one workgroup per sample
int pw_nchw_to_nhwc(const float* src, long src_sn, const TV& d, hipStream_t st);
int pw_nhwc_to_nchw(const TV& s, float* dst, long dst_sn, int acc, hipStream_t st);
struct PackDesc;
int pw_bcast_input_grad(const TV& dz, const PackDesc& d, int seg, float* S /* N*Cout*9 scratch */, float* g, long g_sn,
                        float* dbias /* nullable: the conv's bias gradient falls out of the same sums */, hipStream_t st);
int pw_fold_bias(const float* bias /* nullable */, const float* scale, const float* shift, float* out, int C, hipStream_t st);
int pw_batch_sum(const float* src, long sn, long n_el, int N, float* dst, hipStream_t st);
// dst[i][k] += src[i][k] for up to VEC_ADD_MAX short vectors in ONE launch (private per-stream parameter-gradient accumulators -> the flat gradient buffer)
#define VEC_ADD_MAX 24
struct VecAddJobs { float* dst[VEC_ADD_MAX]; const float* src[VEC_ADD_MAX]; int n[VEC_ADD_MAX]; int count; };
int pw_vec_add(const VecAddJobs& j, hipStream_t st);
