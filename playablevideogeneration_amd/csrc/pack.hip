// Weight re-layout between the boundary format (reference state_dict: OIHW fp32, training/trainer.py:100) and the
// internal packed format of conv_mfma.hip ([tap][Cout_pad][Ktot], k contiguous, segments padded to 16 channels).
// Runs once per optimiser step (weights change every step); ~2 x 39 MB of traffic for BAIR-main: HBM-bound, negligible.
#include "common.h"
#include "pack.h"
#include "pack_elems.h"

namespace {
__global__ void k_pack_fwd(PackDesc d, float* wp) {
    long total = (long)d.KS * d.KS * d.Cout_pad * d.Ktot;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) pk_fwd_elem(d, wp, i);
}
__global__ void k_pack_dgrad(PackDesc d, int seg, float* wpd, int Cd_pad, int Kd) {
    long total = (long)d.KS * d.KS * Cd_pad * Kd;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) pk_dgrad_elem(d, seg, wpd, Cd_pad, Kd, i);
}
__global__ void k_unpack_wgrad(PackDesc d, const float* dwp) {
    long total = (long)d.KS * d.KS * d.Cout_pad * d.Ktot;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) pk_unpack_elem(d, dwp, i);
}
// fused Adam (torch.optim.Adam semantics, L2 weight decay folded into the gradient; training/trainer.py:36,584-587)
__global__ void k_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2s, float gscale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = (gscale != 0.f ? g[i] * gscale : 0.f) + wd * p[i];      // (gscale 0: a step with a ZERO gradient -- torch < 2.0's zero-filled .grad, caddy_adam_step_ex -- whatever the buffer holds)
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2s + eps);
    }
}
}  // namespace

static inline unsigned grid_for(long total) { long b = (total + 255) / 256; return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

int pack_fwd(const PackDesc& d, float* wp, hipStream_t st) {
    hipLaunchKernelGGL(k_pack_fwd, dim3(grid_for((long)d.KS * d.KS * d.Cout_pad * d.Ktot)), dim3(256), 0, st, d, wp);
    return 0;
}
int pack_dgrad(const PackDesc& d, int seg, float* wpd, int Cd_pad, int Kd, hipStream_t st) {
    hipLaunchKernelGGL(k_pack_dgrad, dim3(grid_for((long)d.KS * d.KS * Cd_pad * Kd)), dim3(256), 0, st, d, seg, wpd, Cd_pad, Kd);
    return 0;
}
int unpack_wgrad(const PackDesc& d, const float* dwp, hipStream_t st) {
    hipLaunchKernelGGL(k_unpack_wgrad, dim3(grid_for((long)d.KS * d.KS * d.Cout_pad * d.Ktot)), dim3(256), 0, st, d, dwp);
    return 0;
}
int adam_launch(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step, float gscale, hipStream_t st) {
    float bc1 = 1.f - powf(b1, (float)step), bc2s = sqrtf(1.f - powf(b2, (float)step));
    hipLaunchKernelGGL(k_adam, dim3(grid_for(n)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2s, gscale);
    return 0;
}
