// C-ABI entry points of the individual kernels (unit-parity tests call these through ctypes; SURVEY.md section 8b
// "plus per-kernel entry points").  Declared in include/caddy_hip.h.
#include "common.h"
#include "pointwise.h"
#include "pack.h"

#define ST(s) ((hipStream_t)(s))
extern "C" {
int caddy_k_conv_fwd(const ConvArgs* a, void* s) { return conv_fwd_launch(*a, ST(s)); }
int caddy_k_conv_took_direct(void) { return g_last_conv_direct; }      // 1: the last caddy_k_conv_fwd of this thread ran on the latency kernel (conv_direct.hip)
int caddy_k_conv_stats_tiles(void) { return g_last_conv_stats_tiles; }      // pixel tiles whose BatchNorm partial sums the last caddy_k_conv_fwd of this thread wrote (ConvArgs.stats)
int caddy_k_bn_finalize_tiles(const float* part, int ntiles, int ldp, long count, const float* gamma, const float* beta, float* rmean, float* rvar, int C,
                              float* mean, float* invstd, float* scale, float* shift, void* s) {
    return pw_bn_finalize_tiles(part, ntiles, ldp, count, gamma, beta, rmean, rvar, C, mean, invstd, scale, shift, ST(s));
}
int caddy_k_bn_bwd_lazy(const TV* dout, const TV* x, const float* mean, const float* invstd, const float* gamma, const float* scale, const float* shift, int act,
                        double* sums, double* scratch, const TV* dx, float* dgamma, float* dbeta, void* s) {      // backward of a lazily applied BatchNorm (no materialised output)
    int rc = pw_bn_bwd_reduce(*dout, nullptr, *x, mean, invstd, sums, scratch, dgamma, dbeta, ST(s), act ? scale : nullptr, shift);
    return rc ? rc : pw_bn_bwd_apply(*dout, nullptr, *x, mean, invstd, gamma, sums, *dx, nullptr, nullptr, 1, ST(s), act ? scale : nullptr, shift);
}
int caddy_k_conv_wgrad(const WgradArgs* a, void* s) { return conv_wgrad_launch(*a, ST(s)); }
int caddy_k_conv_pick_bn(int cout) { return conv_pick_bn(cout); }
int caddy_k_conv_avgpool_ok(const ConvArgs* a) { return conv_avgpool_ok(*a); }
int caddy_k_hx_pick_bn(int cout) { return hx_pick_bn(cout); }
int caddy_k_hx_force_big(int v) { g_hx_big_override = v; return 0; }
int caddy_k_hx_set_bg(int mask) { g_hx_bg = mask; return 0; }
long caddy_k_hx_weight_bytes(const PackDesc* d, int seg, int rows_pad, int planes) { return (long)hx_weight_bytes(*d, seg, rows_pad, planes); }
int caddy_k_pack_hx(const PackDesc* d, void* wq, int rows_pad, int seg, int precision, void* s) { return pack_hx(*d, wq, rows_pad, seg, precision, ST(s)); }
int caddy_k_pack_fwd(const PackDesc* d, float* wp, void* s) { return pack_fwd(*d, wp, ST(s)); }
int caddy_k_pack_dgrad(const PackDesc* d, int seg, float* wpd, int Cd_pad, int Kd, void* s) { return pack_dgrad(*d, seg, wpd, Cd_pad, Kd, ST(s)); }
int caddy_k_unpack_wgrad(const PackDesc* d, const float* dwp, void* s) { return unpack_wgrad(*d, dwp, ST(s)); }
int caddy_k_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step, float gscale, void* s) {
    return adam_launch(p, g, m, v, n, lr, b1, b2, eps, wd, step, gscale, ST(s));
}
int caddy_k_copy(const TV* a, const TV* b, int acc, void* s) { return pw_copy(*a, *b, acc, ST(s)); }
int caddy_k_pool2(const TV* in, const TV* out, void* s) { return pw_pool2(*in, *out, ST(s)); }
int caddy_k_pool2_bwd(const TV* dout, const TV* din, void* s) { return pw_pool2_bwd(*dout, *din, 0, ST(s)); }
int caddy_k_up2(const TV* in, const TV* out, void* s) { return pw_up2(*in, *out, ST(s)); }
int caddy_k_up2_bwd(const TV* dout, const TV* din, void* s) { return pw_up2_bwd(*dout, *din, ST(s)); }
int caddy_k_stats(const TV* x, double* sums, void* s) { return pw_stats(*x, sums, nullptr, ST(s)); }
int caddy_k_stats_partials(const TV* x, double* sums, double* scratch, void* s) { return pw_stats(*x, sums, scratch, ST(s)); }
int caddy_k_bn_finalize(const double* sums, long count, const float* gamma, const float* beta, float* rmean, float* rvar, int C, int training,
                        float* mean, float* invstd, float* scale, float* shift, void* s) {
    return pw_bn_finalize(sums, count, gamma, beta, rmean, rvar, C, training, mean, invstd, scale, shift, ST(s));
}
int caddy_k_bn_stats_finalize(const TV* x, double* sums, double* scratch, const float* gamma, const float* beta, float* rmean, float* rvar,
                              float* mean, float* invstd, float* scale, float* shift, void* s) {
    return pw_bn_stats_finalize(*x, sums, scratch, gamma, beta, rmean, rvar, mean, invstd, scale, shift, ST(s));
}
int caddy_k_bn_small_fwd(const TV* x, const float* gamma, const float* beta, float* rmean, float* rvar, float* mean, float* invstd, float* scale, float* shift,
                         const TV* x2, int act, const TV* out, void* s) {
    return pw_bn_small_fwd(*x, gamma, beta, rmean, rvar, mean, invstd, scale, shift, x2, act, *out, ST(s));
}
int caddy_k_bn_small_bwd(const TV* dout, const TV* outm, const TV* x, const float* mean, const float* invstd, const float* gamma, const TV* dx,
                         float* dgamma, float* dbeta, const TV* dres, void* s) {
    return pw_bn_small_bwd(*dout, outm, *x, mean, invstd, gamma, *dx, dgamma, dbeta, dres, 0, ST(s));
}
int caddy_k_bn_apply(const TV* x, const float* scale, const float* shift, const TV* x2, const float* scale2, const float* shift2, int act, const TV* out, void* s) {
    return pw_bn_apply(*x, scale, shift, x2, scale2, shift2, act, *out, ST(s));
}
int caddy_k_bn_bwd_reduce(const TV* dout, const TV* outm, const TV* x, const float* mean, const float* invstd, double* sums, void* s) {
    return pw_bn_bwd_reduce(*dout, outm, *x, mean, invstd, sums, nullptr, nullptr, nullptr, ST(s));
}
int caddy_k_bn_bwd_apply(const TV* dout, const TV* outm, const TV* x, const float* mean, const float* invstd, const float* gamma, const double* sums,
                         const TV* dx, float* dgamma, float* dbeta, void* s) {
    return pw_bn_bwd_apply(*dout, outm, *x, mean, invstd, gamma, sums, *dx, dgamma, dbeta, 0, ST(s));
}
// first-touch variants: the gradient buffer is ASSIGNED (it may hold garbage), see caddy_ctx::alloc_nz
int caddy_k_bn_bwd_apply_assign(const TV* dout, const TV* outm, const TV* x, const float* mean, const float* invstd, const float* gamma, const double* sums, const TV* dx, void* s) {
    return pw_bn_bwd_apply(*dout, outm, *x, mean, invstd, gamma, sums, *dx, nullptr, nullptr, 1, ST(s));
}
int caddy_k_bn_small_bwd_assign(const TV* dout, const TV* outm, const TV* x, const float* mean, const float* invstd, const float* gamma, const TV* dx,
                                float* dgamma, float* dbeta, const TV* dres, void* s) {
    return pw_bn_small_bwd(*dout, outm, *x, mean, invstd, gamma, *dx, dgamma, dbeta, dres, 1, ST(s));
}
int caddy_k_pool2_bwd_assign(const TV* dout, const TV* din, void* s) { return pw_pool2_bwd(*dout, *din, 1, ST(s)); }
int caddy_k_act_bwd_add(const TV* dout, const TV* outm, const TV* dres, void* s) { return pw_act_bwd_add(*dout, *outm, *dres, ST(s)); }
int caddy_k_lstm_fwd(const TV* gates, const TV* cprev, const TV* h, const TV* cn, void* s) { return pw_lstm_fwd(*gates, *cprev, *h, *cn, ST(s)); }
int caddy_k_lstm_bwd(const TV* gates, const TV* cprev, const TV* cn, const TV* dh, const TV* dc, const TV* dgates, const TV* dcprev, void* s) {
    return pw_lstm_bwd(*gates, *cprev, *cn, *dh, *dc, *dgates, *dcprev, ST(s));
}
int caddy_k_tanh_bwd(const TV* dy, const TV* y, const TV* dz, void* s) { return pw_tanh_bwd(*dy, *y, *dz, ST(s)); }
int caddy_k_attn_mul(const TV* x, const TV* out, const TV* att, void* s) { return pw_attn_mul(*x, *out, *att, ST(s)); }
int caddy_k_attn_mul_bwd(const TV* x, const TV* dout, const TV* datt, const TV* dx, void* s) { return pw_attn_mul_bwd(*x, *dout, *datt, *dx, ST(s)); }
int caddy_k_gap(const TV* x, float* out, void* s) { return pw_gap(*x, out, ST(s)); }
int caddy_k_gap_bwd(const float* dout, const TV* dx, void* s) { return pw_gap_bwd(dout, *dx, ST(s)); }
int caddy_k_colsum(const TV* x, float* out, void* s) { return pw_colsum(*x, out, ST(s)); }
int caddy_k_spatial_sum(const TV* x, float* out, long out_sn, void* s) { return pw_spatial_sum(*x, out, out_sn, ST(s)); }
int caddy_k_nchw_to_nhwc(const float* src, long src_sn, const TV* d, void* s) { return pw_nchw_to_nhwc(src, src_sn, *d, ST(s)); }
int caddy_k_nhwc_to_nchw(const TV* src, float* dst, long dst_sn, int acc, void* s) { return pw_nhwc_to_nchw(*src, dst, dst_sn, acc, ST(s)); }
int caddy_k_bcast_input_grad(const TV* dz, const PackDesc* d, int seg, float* S, float* g, long g_sn, float* dbias, void* s) { return pw_bcast_input_grad(*dz, *d, seg, S, g, g_sn, dbias, ST(s)); }
int caddy_k_batch_sum(const float* src, long sn, long n_el, int N, float* dst, void* s) { return pw_batch_sum(src, sn, n_el, N, dst, ST(s)); }
}
