// VGG19 perceptual loss, forward + input gradient, fused into caddy_loss_backward.
//
// Reference: training/losses.py:379-491 (ParallelPerceptualLoss -> UnmeanedPerceptualLoss), model/layers/vgg.py:8-56 (slices of
// torchvision's vgg19().features up to relu5_1), training/trainer.py:442-466,494-500 (per-resolution sum, sum_loss_components, the
// term in the total loss).  Per resolution r in {1, 1/2, 1/4}: VGG19 features of the bilinearly resized ground truth (no gradient) and of
// the reconstruction at relu1_1 .. relu5_1, per level mean |f_gt - f_rec|, and the gradient of the weighted sum w.r.t. the reconstruction
// (the VGG weights are frozen: dgrad only, no wgrad).  Inputs in [-1, 1] are fed as they are -- the reference applies no ImageNet
// normalisation.
//
// Reference quirk reproduced on purpose (pinned by tests/golden/perc_*.npz): losses.py:483-487 binds `total_loss` to the level-0 tensor
// and then accumulates the other levels into it IN PLACE, so the "level 0" entry handed to Trainer.sum_loss_components IS the total:
//   term_r = lambda * (l0 + 2 (l1 + l2 + l3 + l4)),   logged perceptual_loss_r{r}_l0 == perceptual_loss_r{r}.
//
// Convolutions run on the conv kernels of this library (13 x conv3x3 + bias + ReLU forward on 2N images per resolution; the dgrad of the
// reconstruction branch with the ReLU mask and the L1 seed of the tapped feature maps fused into the epilogue: ConvArgs.mask / seed_ref).
// The kernels here are the 2x2 max-pool (forward; backward fused with the ReLU mask of the layer below) and the feature L1.
#include "net.h"
#include "perceptual.h"
#include <cstdio>
#include <cstring>

#define RUN_CK(c, expr) do { if (!(c)->dry) (c)->ck((expr), #expr); } while (0)

namespace {

// torchvision vgg19().features: (index, Cin, Cout) of the 13 convolutions the reference evaluates; pool_before: MaxPool2d(2,2) on the input
struct VggSpec { int idx, cin, cout, pool_before, tap; };
const VggSpec VGG[VGG_NCONV] = {
    {0, 3, 64, 0, 0}, {2, 64, 64, 0, -1}, {5, 64, 128, 1, 1}, {7, 128, 128, 0, -1}, {10, 128, 256, 1, 2}, {12, 256, 256, 0, -1}, {14, 256, 256, 0, -1},
    {16, 256, 256, 0, -1}, {19, 256, 512, 1, 3}, {21, 512, 512, 0, -1}, {23, 512, 512, 0, -1}, {25, 512, 512, 0, -1}, {28, 512, 512, 1, 4}};

// MaxPool2d(2, 2) on dense NHWC maps; odd sizes floor like torch (the last row / column is not covered by any window)
__global__ __launch_bounds__(256) void k_maxpool2(const float* in, float* out, long n_out4, int Hi, int Wi, int C4) {
    const int Ho = Hi >> 1, Wo = Wi >> 1;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n_out4; i += (long)gridDim.x * 256) {
        int c = (int)(i % C4); long q = i / C4; int x = (int)(q % Wo); q /= Wo; int y = (int)(q % Ho); long n = q / Ho;
        const float4* p = reinterpret_cast<const float4*>(in) + ((n * Hi + 2 * y) * (long)Wi + 2 * x) * C4 + c;
        float4 a = p[0], b = p[C4], d = p[(long)Wi * C4], e = p[(long)Wi * C4 + C4];
        float4 m;
        m.x = fmaxf(fmaxf(a.x, b.x), fmaxf(d.x, e.x)); m.y = fmaxf(fmaxf(a.y, b.y), fmaxf(d.y, e.y));
        m.z = fmaxf(fmaxf(a.z, b.z), fmaxf(d.z, e.z)); m.w = fmaxf(fmaxf(a.w, b.w), fmaxf(d.w, e.w));
        reinterpret_cast<float4*>(out)[i] = m;
    }
}
// ---- S16 tensors (common.h): four consecutive channels at flat float index i4 (a multiple of 4) of a DENSE map (ld == C, C a multiple of 32) ----
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 s16_load4_f16(const float* p, long i4) {      // value = hi + lo
    const _Float16* h = reinterpret_cast<const _Float16*>(p + (i4 & ~31L)) + (i4 & 31);
    const f16x4_t hi = *reinterpret_cast<const f16x4_t*>(h), lo = *reinterpret_cast<const f16x4_t*>(h + 32);
    return make_float4((float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]);
}
__device__ __forceinline__ void s16_store4_bf16(float* p, long i4, float4 v) {      // the split of conv_hx.hip's loaders: hi = bf16(x), lo = bf16(x - hi)
    __bf16* h = reinterpret_cast<__bf16*>(p + (i4 & ~31L)) + (i4 & 31);
    bf16x4_t hi, lo;
    hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
    lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]); lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
    *reinterpret_cast<bf16x4_t*>(h) = hi;
    *reinterpret_cast<bf16x4_t*>(h + 32) = lo;
}
template <bool S> __device__ __forceinline__ float4 ld4(const float* p, long i) { return S ? s16_load4_f16(p, 4 * i) : reinterpret_cast<const float4*>(p)[i]; }
template <bool S> __device__ __forceinline__ void st4_grad(float* p, long i, float4 v) { if (S) s16_store4_bf16(p, 4 * i, v); else reinterpret_cast<float4*>(p)[i] = v; }

// gradient of [ReLU -> MaxPool2d(2,2)] in one pass: gz[pos] = g_pooled[window] if pos is the window's first maximum (torch's tie rule: strict >,
// scan order (0,0),(0,1),(1,0),(1,1)) and a[pos] > 0, else 0.  Every position of gz is ASSIGNED (no zero-fill needed), including the
// uncovered last row / column of odd-sized maps (zero).
__device__ __forceinline__ void route4(float a0, float a1, float a2, float a3, float g, float& z0, float& z1, float& z2, float& z3) {
    int k = 0; float m = a0;
    if (a1 > m) { m = a1; k = 1; }
    if (a2 > m) { m = a2; k = 2; }
    if (a3 > m) { m = a3; k = 3; }
    const float v = m > 0.f ? g : 0.f;
    z0 = k == 0 ? v : 0.f; z1 = k == 1 ? v : 0.f; z2 = k == 2 ? v : 0.f; z3 = k == 3 ? v : 0.f;
}
// AS: `a` (the forward activation) is an S16-f16 tensor; ZS: gz is written as S16-bf16 (the operand format of the dgrad that consumes it)
template <bool AS, bool ZS>
__global__ __launch_bounds__(256) void k_maxpool2_bwd_relu(const float* a, const float* gp, float* gz, long n_win4, int Hi, int Wi, int C4) {
    const int Ho = Hi >> 1, Wo = Wi >> 1, Hc = (Hi + 1) >> 1, Wc = (Wi + 1) >> 1;      // windows incl. the partial ones of odd sizes
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n_win4; i += (long)gridDim.x * 256) {
        int c = (int)(i % C4); long q = i / C4; int x = (int)(q % Wc); q /= Wc; int y = (int)(q % Hc); long n = q / Hc;
        const long base = ((n * Hi + 2 * y) * (long)Wi + 2 * x) * C4 + c;
        const long o1 = C4, o2 = (long)Wi * C4, o3 = o2 + C4;
        if (y >= Ho || x >= Wo) {      // partial window: no pooled output reads these positions
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            st4_grad<ZS>(gz, base, zero);
            if (2 * x + 1 < Wi) st4_grad<ZS>(gz, base + o1, zero);
            if (2 * y + 1 < Hi) { st4_grad<ZS>(gz, base + o2, zero); if (2 * x + 1 < Wi) st4_grad<ZS>(gz, base + o3, zero); }
            continue;
        }
        const float4 A = ld4<AS>(a, base), B = ld4<AS>(a, base + o1), D = ld4<AS>(a, base + o2), E = ld4<AS>(a, base + o3);
        const float4 G = reinterpret_cast<const float4*>(gp)[((n * Ho + y) * (long)Wo + x) * C4 + c];
        float4 zA, zB, zD, zE;
        route4(A.x, B.x, D.x, E.x, G.x, zA.x, zB.x, zD.x, zE.x);
        route4(A.y, B.y, D.y, E.y, G.y, zA.y, zB.y, zD.y, zE.y);
        route4(A.z, B.z, D.z, E.z, G.z, zA.z, zB.z, zD.z, zE.z);
        route4(A.w, B.w, D.w, E.w, G.w, zA.w, zB.w, zD.w, zE.w);
        st4_grad<ZS>(gz, base, zA); st4_grad<ZS>(gz, base + o1, zB); st4_grad<ZS>(gz, base + o2, zD); st4_grad<ZS>(gz, base + o3, zE);
    }
}
// sum |f_rec - f_gt| over a dense feature map (double atomics per block); optionally the masked L1 seed of the top level:
// gz = seed_w * sign(f_rec - f_gt) where f_rec > 0 (the ReLU that produced f_rec), else 0.  RS / GS: rec / gt are S16-f16 tensors; ZS: gz is written as S16-bf16
template <bool RS, bool GS, bool ZS>
__global__ __launch_bounds__(256) void k_feat_l1(const float* rec, const float* gt, long n4, float seed_w, float* gz, double* acc) {
    __shared__ double sh[4];
    double s = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 r = ld4<RS>(rec, i), g = ld4<GS>(gt, i);
        const float d[4] = {r.x - g.x, r.y - g.y, r.z - g.z, r.w - g.w};
        const float rv[4] = {r.x, r.y, r.z, r.w};
        float z[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            s += (double)fabsf(d[e]);
            z[e] = rv[e] > 0.f ? (d[e] > 0.f ? seed_w : (d[e] < 0.f ? -seed_w : 0.f)) : 0.f;
        }
        if (gz) st4_grad<ZS>(gz, i, make_float4(z[0], z[1], z[2], z[3]));
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, sh[0] + sh[1] + sh[2] + sh[3]);
}
// per IMAGE: acc[n] += sum |f_rec - f_gt| over image n's feature map (blockIdx.y = image); the evaluator's per-position perceptual loss (evaluation/evaluator.py:193)
template <bool RS, bool GS>
__global__ __launch_bounds__(256) void k_feat_l1_img(const float* rec, const float* gt, long n4_img, double* acc) {
    __shared__ double sh[4];
    const float* r4 = rec + 4 * (long)blockIdx.y * n4_img;
    const float* g4 = gt + 4 * (long)blockIdx.y * n4_img;
    double s = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4_img; i += (long)gridDim.x * 256) {
        const float4 r = ld4<RS>(r4, i), g = ld4<GS>(g4, i);
        s += (double)fabsf(r.x - g.x) + (double)fabsf(r.y - g.y) + (double)fabsf(r.z - g.z) + (double)fabsf(r.w - g.w);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc + blockIdx.y, sh[0] + sh[1] + sh[2] + sh[3]);
}
// ground-truth frames for the VGG19 branch: first 3 channels of observation t + t_off, bilinearly resized like F.interpolate(align_corners=False)
// does for exact factors (losses.py:450): f = 2 -> 2x2 mean, f = 4 -> mean of the central 2x2 of each 4x4 block (same arithmetic as k_loss_l1)
__global__ __launch_bounds__(256) void k_gt_resize(TV gt, float* out, int Ho, int Wo, long npix, int f, int t_off, int Tobs, int Trec) {
    for (long q = blockIdx.x * 256L + threadIdx.x; q < npix; q += (long)gridDim.x * 256) {
        const long fr = q / ((long)Ho * Wo); const int rem = (int)(q - fr * Ho * Wo); const int y = rem / Wo, x = rem - y * Wo;
        const long b = fr / Trec, t = fr - b * Trec;
        const float* g = gt.p + (b * Tobs + t + t_off) * gt.sn;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        float* ov = &o.x;
        for (int ch = 0; ch < 3; ch++) {
            if (f == 1) ov[ch] = g[((long)y * gt.W + x) * gt.ld + ch];
            else {
                const int y0 = f == 2 ? 2 * y : 4 * y + 1, x0 = f == 2 ? 2 * x : 4 * x + 1;
                const float* gp = g + ((long)y0 * gt.W + x0) * gt.ld + ch;
                ov[ch] = 0.5f * (0.5f * gp[0] + 0.5f * gp[gt.ld]) + 0.5f * (0.5f * gp[(long)gt.W * gt.ld] + 0.5f * gp[(long)(gt.W + 1) * gt.ld]);
            }
        }
        reinterpret_cast<float4*>(out)[q] = o;
    }
}
// frames of the time steps [t0, t0 + len) of every sample, (B, Tfull) sample-major <-> (B, len) sample-major; F4 = float4 per frame.  add = 0: chunk[q] = full[...] (gather);
// add = 1: full[...] += chunk[q] (the chunk's gradient back onto the seeds; each element has one writer per launch)
__global__ __launch_bounds__(256) void k_time_chunk(float4* full, float4* chunk, long F4, int Tfull, int t0, int len, long total4, int add) {
    for (long q = blockIdx.x * 256L + threadIdx.x; q < total4; q += (long)gridDim.x * 256) {
        const long fr = q / F4, i = q - fr * F4;
        const long b = fr / len, j = fr - b * len;
        const long s = ((b * Tfull + t0 + j) * F4) + i;
        if (add) { float4 a = full[s]; const float4 v = chunk[q]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; full[s] = a; }
        else chunk[q] = full[s];
    }
}
__global__ void k_copy_f(const float* src, float* dst, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}
inline unsigned grid_for(long items) { long b = (items + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

T4 valloc(caddy_ctx* c, int N, int H, int W, int C) {      // bottom-up allocation past the forward graph (released per resolution); gradient at the mirror
    const int ld = round_up(C, 4);
    float* d = (float*)c->act.alloc((size_t)N * H * W * ld * 4);
    return T4{d, (float*)((char*)d + c->grad_delta), N, H, W, C, (long)H * W * ld, ld, true};
}
// format dispatch of the point-wise kernels (S16 flags -> template instances)
void launch_feat_l1(hipStream_t st, const T4& rec, const T4& gt, float seed_w, float* gz, bool gz_s16, double* acc) {
    const long n4 = (long)rec.N * rec.H * rec.W * (rec.C / 4);
    const dim3 g(grid_for(n4)), b(256);
    const int k = (rec.fmt ? 4 : 0) | (gt.fmt ? 2 : 0) | ((gz && gz_s16) ? 1 : 0);
    const float* r = rec.d; const float* t = gt.d;
    switch (k) {
        case 0: hipLaunchKernelGGL((k_feat_l1<false, false, false>), g, b, 0, st, r, t, n4, seed_w, gz, acc); break;
        case 1: hipLaunchKernelGGL((k_feat_l1<false, false, true>), g, b, 0, st, r, t, n4, seed_w, gz, acc); break;
        case 2: hipLaunchKernelGGL((k_feat_l1<false, true, false>), g, b, 0, st, r, t, n4, seed_w, gz, acc); break;
        case 3: hipLaunchKernelGGL((k_feat_l1<false, true, true>), g, b, 0, st, r, t, n4, seed_w, gz, acc); break;
        case 4: hipLaunchKernelGGL((k_feat_l1<true, false, false>), g, b, 0, st, r, t, n4, seed_w, gz, acc); break;
        case 5: hipLaunchKernelGGL((k_feat_l1<true, false, true>), g, b, 0, st, r, t, n4, seed_w, gz, acc); break;
        case 6: hipLaunchKernelGGL((k_feat_l1<true, true, false>), g, b, 0, st, r, t, n4, seed_w, gz, acc); break;
        default: hipLaunchKernelGGL((k_feat_l1<true, true, true>), g, b, 0, st, r, t, n4, seed_w, gz, acc); break;
    }
}
void launch_feat_l1_img(hipStream_t st, const T4& rec, const T4& gt, double* acc) {
    const long n4 = (long)rec.H * rec.W * (rec.C / 4);
    const unsigned bx = (unsigned)(n4 / 1024 < 1 ? 1 : (n4 / 1024 > 64 ? 64 : n4 / 1024));
    const dim3 g(bx, rec.N), b(256);
    const float* r = rec.d; const float* t = gt.d;
    if (rec.fmt && gt.fmt) hipLaunchKernelGGL((k_feat_l1_img<true, true>), g, b, 0, st, r, t, n4, acc);
    else if (rec.fmt) hipLaunchKernelGGL((k_feat_l1_img<true, false>), g, b, 0, st, r, t, n4, acc);
    else if (gt.fmt) hipLaunchKernelGGL((k_feat_l1_img<false, true>), g, b, 0, st, r, t, n4, acc);
    else hipLaunchKernelGGL((k_feat_l1_img<false, false>), g, b, 0, st, r, t, n4, acc);
}
void launch_maxpool_bwd(hipStream_t st, const T4& pre, const float* gp, bool gz_s16) {
    const long n4 = (long)pre.N * ((pre.H + 1) / 2) * ((pre.W + 1) / 2) * (pre.C / 4);
    const dim3 g(grid_for(n4)), b(256);
    const float* a = pre.d;
    if (pre.fmt && gz_s16) hipLaunchKernelGGL((k_maxpool2_bwd_relu<true, true>), g, b, 0, st, a, gp, pre.g, n4, pre.H, pre.W, pre.C / 4);
    else if (pre.fmt) hipLaunchKernelGGL((k_maxpool2_bwd_relu<true, false>), g, b, 0, st, a, gp, pre.g, n4, pre.H, pre.W, pre.C / 4);
    else if (gz_s16) hipLaunchKernelGGL((k_maxpool2_bwd_relu<false, true>), g, b, 0, st, a, gp, pre.g, n4, pre.H, pre.W, pre.C / 4);
    else hipLaunchKernelGGL((k_maxpool2_bwd_relu<false, false>), g, b, 0, st, a, gp, pre.g, n4, pre.H, pre.W, pre.C / 4);
}
}  // namespace

int vgg_param_count() { return 2 * VGG_NCONV; }
long vgg_param_floats() {
    long n = 0;
    for (int i = 0; i < VGG_NCONV; i++) n += (long)VGG[i].cout * VGG[i].cin * 9 + round_up(VGG[i].cout, 4);
    return n;
}
int vgg_param_info(int index, caddy_param_info* out) {
    if (index < 0 || index >= 2 * VGG_NCONV) return -1;
    long off = 0;
    for (int i = 0; i < VGG_NCONV; i++) {
        const long nw = (long)VGG[i].cout * VGG[i].cin * 9;
        if (index == 2 * i || index == 2 * i + 1) {
            memset(out, 0, sizeof(*out));
            const bool w = index == 2 * i;
            snprintf(out->name, sizeof(out->name), "features.%d.%s", VGG[i].idx, w ? "weight" : "bias");
            out->offset = w ? off : off + nw; out->kind = 3;
            if (w) { out->ndim = 4; out->shape[0] = VGG[i].cout; out->shape[1] = VGG[i].cin; out->shape[2] = 3; out->shape[3] = 3; }
            else { out->ndim = 1; out->shape[0] = VGG[i].cout; out->shape[1] = out->shape[2] = out->shape[3] = 1; }
            return 0;
        }
        off += nw + round_up(VGG[i].cout, 4);
    }
    return -1;
}

// persistent packed weights (forward + dgrad form) and biases: carved out of the caller's workspace when caddy_config.perceptual != 0
void vgg_build(caddy_ctx* c) {
    VggState& V = c->vgg;
    V.enabled = true; V.loaded = false;
    for (int i = 0; i < VGG_NCONV; i++) {
        VggLayer& L = V.conv[i];
        PackDesc& d = L.pd;
        d = PackDesc{};
        d.nw = 1; d.Co_each = VGG[i].cout; d.Cin = VGG[i].cin; d.KS = 3; d.nseg = 1;
        d.seg_off[0] = 0; d.seg_C[0] = VGG[i].cin; d.seg_Cpad[0] = round_up(VGG[i].cin, CONV_BK);
        d.Cout = VGG[i].cout; d.Cout_pad = round_up(d.Cout, conv_pick_bn(d.Cout)); d.Ktot = d.seg_Cpad[0];
        L.wp = (float*)c->persist.alloc((size_t)9 * d.Cout_pad * d.Ktot * 4);
        L.kd = round_up(d.Cout, CONV_BK);
        L.cd_pad = round_up(VGG[i].cin, conv_pick_bn(VGG[i].cin));
        L.wpd = (float*)c->persist.alloc((size_t)9 * L.cd_pad * L.kd * 4);
        L.bias = (float*)c->persist.alloc((size_t)round_up(d.Cout, 4) * 4);
        for (int pl = 0; pl < 2; pl++) {
            L.wq[pl] = c->persist.alloc(hx_weight_bytes(d, -1, round_up(d.Cout, hx_pick_bn(d.Cout)), 2 - pl));
            L.wqd[pl] = VGG[i].cin >= 32 ? c->persist.alloc(hx_weight_bytes(d, 0, round_up(VGG[i].cin, hx_pick_bn(VGG[i].cin)), 2 - pl)) : nullptr;
        }
        L.wq[2] = c->persist.alloc(hx_weight_bytes(d, -1, round_up(d.Cout, hx_pick_bn(d.Cout)), 2));
    }
}

// caddy_load_vgg: `flat` = device buffer laid out per vgg_param_info (torchvision names features.{idx}.weight / .bias, OIHW fp32)
int vgg_load(caddy_ctx* c, const float* flat) {
    VggState& V = c->vgg;
    if (!V.enabled) { set_error("caddy_load_vgg: the context was created with caddy_config.perceptual = 0"); return -2; }
    bool dry = c->dry;
    long off = 0;
    for (int i = 0; i < VGG_NCONV; i++) {
        VggLayer& L = V.conv[i];
        const long nw = (long)VGG[i].cout * VGG[i].cin * 9;
        L.pd.w[0] = flat + off;
        RUN_CK(c, pack_fwd(L.pd, L.wp, c->stream));
        RUN_CK(c, pack_dgrad(L.pd, 0, L.wpd, L.cd_pad, L.kd, c->stream));
        for (int pl = 0; pl < 2; pl++) {
            RUN_CK(c, pack_hx(L.pd, L.wq[pl], round_up(L.pd.Cout, hx_pick_bn(L.pd.Cout)), -1, pl == 0 ? PREC_F16X3 : PREC_F16X1, c->stream));
            if (L.wqd[pl]) RUN_CK(c, pack_hx(L.pd, L.wqd[pl], round_up(VGG[i].cin, hx_pick_bn(VGG[i].cin)), 0, pl == 0 ? PREC_BF16X3 : PREC_BF16X1, c->stream));
        }
        RUN_CK(c, pack_hx(L.pd, L.wq[2], round_up(L.pd.Cout, hx_pick_bn(L.pd.Cout)), -1, PREC_BF16X3, c->stream));
        if (!dry) hipLaunchKernelGGL(k_copy_f, dim3(1), dim3(256), 0, c->stream, flat + off + nw, L.bias, (long)VGG[i].cout);
        L.pd.w[0] = nullptr;      // the caller's buffer is not referenced after this call
        off += nw + round_up(VGG[i].cout, 4);
    }
    V.loaded = true;
    return c->fail ? -1 : 0;
}

namespace {
struct Branch { T4 a[VGG_NCONV]; T4 p[VGG_NCONV]; };      // a[i]: ReLU output of conv i; p[i]: pooled input of conv i (pool_before)

int conv_call(caddy_ctx* c, const ConvArgs& a, double flops, int kind) {
    int save = c->prof_kind_override;
    c->prof_kind_override = kind;
    int rc = c->timed_conv_fwd(a, flops);
    c->prof_kind_override = save;
    return rc;
}
// does the context keep VGG19 feature maps / feature gradients of well-filled layers as S16 tensors?  (both passes on the split-operand kernels: an exact-fp32 pass would have to
// read them)
bool vgg_use_s16(const caddy_ctx* c) { return c->vgg_s16 && c->vgg_precision == PREC_F16X3 && c->vgg_precision_bwd == PREC_BF16X3; }
// is conv i's launch on an N x H x W input one of the k_conv_hx variants that read / write S16 tensors?
bool vgg_io_ok(const caddy_ctx* c, int i, int N, int H, int W) {
    return vgg_use_s16(c) && VGG[i].cin >= 32 && !c->layer_fallback[CADDY_VGG_FLAG0 + i] && conv_hx_s16_ok(N, H, W, VGG[i].cout);      // (a layer on the split-bf16 fallback exchanges fp32 tensors)
}

// 13 x (conv3x3 + bias + ReLU) with the 2x2 max-pools; taps[] (if given) receive the five tapped feature maps in place.
// Formats (round 5): a feature map is written PRE-SPLIT (S16-f16, T4::fmt) by the epilogue of its producer when that launch and the launch of the convolution that consumes it
// are both on the S16-capable tile variants (conv_hx_s16_ok: the well-filled ones) -- the consumer then copies operand rows instead of converting every halo element in every
// one of its output-channel blocks; the point-wise consumers (feature L1, max-pool backward, ReLU masks) read either format.
void vgg_forward(caddy_ctx* c, const T4& img, Branch& B, const T4* taps, bool keep_all) {
    bool dry = c->dry;
    VggState& V = c->vgg;
    T4 x = img;
    T4 pooled{};                                              // output of a max-pool fused into the previous conv's epilogue
    bool have_pooled = false;
    for (int i = 0; i < VGG_NCONV; i++) {
        VggLayer& L = V.conv[i];
        if (VGG[i].pool_before) {
            if (have_pooled) { B.p[i] = pooled; x = pooled; have_pooled = false; }
            else {
                T4 p = valloc(c, x.N, x.H / 2, x.W / 2, x.C);
                const long n4 = (long)p.N * p.H * p.W * (p.C / 4);
                if (x.fmt) { c->fail = true; set_error("internal: S16 feature map handed to the stand-alone max-pool"); }
                if (!dry) hipLaunchKernelGGL(k_maxpool2, dim3(grid_for(n4)), dim3(256), 0, c->stream, (const float*)x.d, p.d, n4, x.H, x.W, p.C / 4);
                B.p[i] = p; x = p;
            }
        }
        T4 out = (taps && VGG[i].tap >= 0) ? taps[VGG[i].tap] : valloc(c, x.N, x.H, x.W, VGG[i].cout);
        out.fmt = 0;
        ConvArgs a{};
        a.src[0] = ConvSrc{x.d, x.sn, x.ld, x.C, round_up(x.C, CONV_BK), 0};
        a.nsrc = 1; a.N = x.N; a.H = x.H; a.W = x.W; a.KS = 3; a.wp = L.wp; a.Ktot = L.pd.Ktot; a.Cout = L.pd.Cout; a.Cout_pad = L.pd.Cout_pad;
        a.bias = L.bias; a.act = 2; a.out = out.d; a.out_sn = out.sn; a.out_ld = out.ld;
        a.precision = c->vgg_precision == PREC_F16X1 ? PREC_F16X1 : (c->vgg_precision == PREC_FP32 ? PREC_FP32 : (c->vgg_precision == PREC_BF16X3 ? PREC_BF16X3 : PREC_F16X3));
        // f16 range guard: real VGG19 weights on un-normalised inputs are where a forward activation could leave the f16 range.  A layer that reported it (caddy_f16_saturated) runs
        // on split bf16 from then on -- 8 + 8 mantissa bits, the full fp32 exponent range -- and only that layer (round 4 moved the whole context to exact fp32: 2.2 x slower)
        if (a.precision == PREC_F16X3 && c->layer_fallback[CADDY_VGG_FLAG0 + i]) a.precision = PREC_BF16X3;
        if (a.precision != PREC_FP32) a.wq = L.wq[a.precision == PREC_F16X1 ? 1 : (a.precision == PREC_BF16X3 ? 2 : 0)];
        a.sat_flag = c->sat_flag + CADDY_VGG_FLAG0 + i;
        const bool io_here = vgg_io_ok(c, i, x.N, x.H, x.W);
        const bool pool_next = i + 1 < VGG_NCONV && VGG[i + 1].pool_before;
        a.in_s16 = x.fmt;
        if (x.fmt && !io_here) { c->fail = true; set_error("internal: S16 input for a VGG19 launch that cannot read it"); }
        // MaxPool2d(2, 2) in front of the next conv: written by THIS conv's epilogue on the split-operand kernel (the window's four pixels sit in
        // one lane) -- no separate pass over the full-resolution map; a branch that is never back-propagated (keep_all = false: the ground truth)
        // does not even store the full-resolution map of such a layer (it is no tap: the taps are the first convs AFTER a pool)
        if (pool_next && a.wq && a.precision == PREC_F16X3 && VGG[i].cin >= 32 && conv_hx_pool_ok(x.N, x.H, x.W, VGG[i].cout)) {
            pooled = valloc(c, x.N, x.H / 2, x.W / 2, VGG[i].cout);
            a.pool_out = pooled.d; a.pool_sn = pooled.sn; a.pool_ld = pooled.ld;
            a.skip_out = (!keep_all && VGG[i].tap < 0) ? 1 : 0;
            have_pooled = true;
            // the pooled map as S16 when its consumer (conv i + 1 on the pooled geometry) reads S16; the full-resolution map (read by the max-pool backward only) in the same format
            if (io_here && vgg_io_ok(c, i + 1, x.N, x.H / 2, x.W / 2)) { a.pool_s16 = 1; pooled.fmt = 1; if (!a.skip_out) { a.out_s16 = 1; out.fmt = 1; } }
        } else if (io_here && !pool_next) {
            // consumed by conv i + 1 on the same geometry (or, for relu5_1, by the feature L1 alone)
            if (i + 1 == VGG_NCONV || vgg_io_ok(c, i + 1, x.N, x.H, x.W)) { a.out_s16 = 1; out.fmt = 1; }
        }
        a.sat_out_next = ((a.out_s16 || a.pool_s16) && i + 1 < VGG_NCONV) ? 1 : 0;      // a clamped OUTPUT value is the next layer's range problem (flag words are consecutive per layer)
        if (!dry) c->ck(conv_call(c, a, 2.0 * x.N * x.H * x.W * 9.0 * VGG[i].cin * VGG[i].cout, 3), "vgg conv");
        B.a[i] = out; x = out;
    }
}
}  // namespace

// Ground-truth branch ahead of time: called from the forward pass right after the observations are in NHWC, it runs the resize + the 13
// convolutions of the ground-truth frames at the three resolutions on the driver's SIDE stream, concurrently with the model's forward pass
// (whose recurrent convolutions on 16x16 / 32x32 maps leave most of the chip idle).  The five tapped maps per resolution stay in the
// activation arena until caddy_loss_backward; the other feature maps go through one scratch region reused by the three resolutions.
void vgg_gt_prefetch(caddy_ctx* c, int Trec, int t_off) {
    bool dry = c->dry;
    const caddy_config& g = c->cfg;
    const int tc[5] = {64, 128, 256, 512, 512};
    // chunk table of this forward pass (caddy_ctx::perc_plan): the taps are laid out per chunk, whatever the loss call then does with them.  Pretraining (no BPTT chain to run
    // beside) and evaluation passes keep one chunk.
    c->perc_plan(Trec, c->training && !c->pretraining);
    c->gt_lo = (c->act.off + 255) & ~(size_t)255;
    T4 gimg[caddy_ctx::PERC_MAX_CHUNKS][3];
    int kmax = 0;
    // (caddy_ctx::perc_range: the full-resolution level follows the chunk table, the half- and quarter-resolution levels are cut once)
    for (int k = 0; k < c->perc_nch; k++) {
        if (c->perc_t0[k] - c->perc_t0[k + 1] > c->perc_t0[kmax] - c->perc_t0[kmax + 1]) kmax = k;
        for (int r = 0; r < 3; r++) {
            int t0, len;
            if (!c->perc_range(r, k, &t0, &len)) continue;
            const int N = g.batch * len;
            int h = g.height >> r, w = g.width >> r;
            gimg[k][r] = valloc(c, N, h, w, 3);                 // (ld 4)
            for (int l = 0; l < 5; l++) { c->gt_taps_c[k][r][l] = valloc(c, N, h, w, tc[l]); h /= 2; w /= 2; }
        }
    }
    for (int r = 0; r < 3; r++) c->gt_img[r] = gimg[0][r];
    {   // scratch for the non-tapped maps of the largest chunk at the largest resolution (allocation pattern of vgg_forward at r = 0)
        const size_t m0 = c->act.off;
        const bool was_dry = c->dry; c->dry = true;
        size_t end = m0;
        for (int r = 0; r < 2; r++) {      // (the un-chunked half-resolution level can need more than a short full-resolution chunk)
            Branch G{};
            c->act.off = m0;
            vgg_forward(c, gimg[r == 0 ? kmax : 0][r], G, c->gt_taps_c[r == 0 ? kmax : 0][r], false);
            if (c->act.off > end) end = c->act.off;
        }
        c->act.off = end;
        c->dry = was_dry;
        // keep the region allocated (the main stream goes on allocating past it); the side stream re-walks it for every chunk and resolution
        c->gt_scratch_off = m0; c->gt_scratch_end = c->act.off;
    }
    c->gt_hi = c->act.off;
    c->gt_prefetched = false;
    if (dry || !c->vgg.loaded || !c->perc_prefetch || !c->training) return;
    c->ensure_side();
    if (!c->use_side || !c->side) return;
    hipStream_t main_st = c->stream, side = c->wgrad_stream();      // ordered after everything enqueued so far (the NHWC observations)
    c->stream = side;
    for (int k = 0; k < c->perc_nch; k++) {                  // (the order the loss call consumes them in: last time steps first)
        for (int r = 0; r < 3; r++) {
            int t0, len;
            if (!c->perc_range(r, k, &t0, &len)) continue;
            const T4& gi = gimg[k][r];
            const long npix = (long)gi.N * gi.H * gi.W;
            hipLaunchKernelGGL(k_gt_resize, dim3(grid_for(npix)), dim3(256), 0, side, dv(c->obs), gi.d, gi.H, gi.W, npix, 1 << r, t_off + t0, g.seq_len, len);
            const size_t keep = c->act.off;
            c->act.off = c->gt_scratch_off;                    // (host-side bump pointer only: the region is private to the side stream)
            Branch G{};
            vgg_forward(c, gi, G, c->gt_taps_c[k][r], false);
            for (int i = 0; i < VGG_NCONV; i++) if (VGG[i].tap >= 0) c->gt_taps_c[k][r][VGG[i].tap].fmt = G.a[i].fmt;      // (a tapped map may be an S16 tensor)
            c->act.off = keep;
        }
    }
    c->stream = main_st;
    hipEventRecord(c->gt_done, side);
    c->gt_prefetched = true;
}

// Called from loss_backward after the L1 terms (which also wrote the resized ground-truth images gt_img[r]) and before the tape is replayed:
// accumulates d(perceptual term)/d(rec_r) into the gradients of c->frames[r] and the raw level sums into c->loss_acc.
// Round 6: chunk by chunk of time steps (caddy_ctx::perc_t0, last steps first).  One chunk: the form of rounds 2 - 5 (the half- and quarter-resolution levels on the side stream
// beside the full-resolution one, the dgrad of conv1_1 accumulating straight onto the L1 seeds).  Several chunks: a chunk's frames are gathered into a contiguous batch
// (k_time_chunk), its gradient is assigned to the gathered copy and added onto the seeds of its time steps; with caddy_ctx::perc_pipelined ALL of it runs on the side stream and
// an event per chunk tells the tape replay which time steps may start (perc_wait).
void vgg_perceptual(caddy_ctx* c, double lambda, const T4* gt_img, VggLevels* lv) {
    bool dry = c->dry;
    VggState& V = c->vgg;
    hipStream_t st = c->stream;
    const caddy_config& g = c->cfg;
    const bool pre = c->gt_prefetched && !dry;      // the ground-truth branch already ran beside the forward pass (vgg_gt_prefetch)
    const int nch = c->perc_nch, Trec = c->perc_trec, B = g.batch;
    const bool pipe = c->perc_pipelined && nch > 1 && !dry;
    // The half- and quarter-resolution levels (24 % of the work, most of it in under-filled launches on 8x8 ... 64x64 maps) run on the driver's side
    // stream BESIDE the full-resolution level (one-chunk form): separate memory (they are allocated first and stay live until the join), their own thin-kernel
    // scratch, the three levels only meet in the loss accumulators (atomics) and write disjoint gradient tensors (d rec_r).
    const bool no_par = caddy_serial_streams();
    const bool par = nch == 1 && !no_par && !dry && c->use_side && c->side != nullptr && !c->prof;
    hipStream_t side = (par || pipe) ? c->wgrad_stream() : st;      // (ordered after the L1 kernels that wrote the seeds / resized ground truth)
    if (pre) { hipStreamWaitEvent(st, c->gt_done, 0); if (pipe) hipStreamWaitEvent(side, c->gt_done, 0); }
    const size_t mark_all = c->act.off;
    size_t side_end = mark_all;                             // memory layout is the parallel one whether or not the levels really overlap (dry-run sizing)
    // Pipelined form: nothing runs beside the FIRST chunk (the tape replay waits for it), so that chunk keeps the level parallelism of the one-pass form -- its full-resolution
    // level on the main stream (in front of the replay's launches), its half- and quarter-resolution levels on the side stream; the later chunks run on the side stream alone,
    // beside the replay.  Memory: [mark_all, main_end) the first chunk's full-resolution level, everything the side stream touches above it.
    size_t side_base = mark_all;
    for (int k = 0; k < nch; k++) {
    const bool first_par = nch > 1 && k == 0 && (pipe || dry);      // (dry run: the same layout)
    for (int oi = 0; oi < 3; oi++) {
        const int r = first_par ? oi : (oi == 2 ? 0 : oi + 1);      // levels 1, 2, 0 -- the first chunk of the pipelined form 0 (main stream), 1, 2
        int t0, len;
        if (!c->perc_range(r, k, &t0, &len)) continue;      // (the half- and quarter-resolution levels are cut once: caddy_ctx::perc_range)
        const bool whole = len == Trec;                     // this level's batch is the frame tensor itself
        const bool on_side = (pipe && !(first_par && r == 0)) || (par && r != 0);
        c->stream = on_side ? side : st;
        hipStream_t st = c->stream;                         // (shadows the outer one for the point-wise launches below)
        float* const aux = on_side ? c->conv_aux2 : c->conv_aux;
        if (nch > 1) c->act.off = (first_par && r == 0) ? mark_all : side_base;      // chunks and levels that follow each other on one stream share their region
        else if (r == 0 && !no_par) c->act.off = side_end;      // above everything levels 1 / 2 may still be using
        const T4& full = c->frames[r];
        const size_t mark = c->act.off;
        const long F4 = (long)full.H * full.W * (full.ld / 4);
        // this chunk's reconstructed frames: the tensor itself (one chunk) or a gathered copy
        T4 rec = full;
        if (!whole) {
            rec = valloc(c, B * len, full.H, full.W, 3);
            if (full.ld != 4 || full.sn != F4 * 4) { c->fail = true; set_error("internal: frame tensor layout of the chunked perceptual pass"); }
            if (!dry) hipLaunchKernelGGL(k_time_chunk, dim3(grid_for((long)rec.N * F4)), dim3(256), 0, st, (float4*)full.d, (float4*)rec.d, F4, Trec, t0, len, (long)rec.N * F4, 0);
        }
        // ground-truth branch: keeps only the five tapped maps
        T4 taps[5];
        Branch G{}, R{};
        if (pre) { for (int l = 0; l < 5; l++) taps[l] = c->gt_taps_c[k][r][l]; }
        else {
            int h = rec.H, w = rec.W; const int tc[5] = {64, 128, 256, 512, 512};
            for (int l = 0; l < 5; l++) { taps[l] = valloc(c, rec.N, h, w, tc[l]); h /= 2; w /= 2; }      // MaxPool2d floors odd sizes
            T4 gi = gt_img[r];
            if (!whole) {
                gi = valloc(c, rec.N, rec.H, rec.W, 3);
                if (!dry) hipLaunchKernelGGL(k_time_chunk, dim3(grid_for((long)rec.N * F4)), dim3(256), 0, st, (float4*)gt_img[r].d, (float4*)gi.d, F4, Trec, t0, len, (long)rec.N * F4, 0);
            }
            const size_t mark2 = c->act.off;
            vgg_forward(c, gi, G, taps, false);
            for (int i = 0; i < VGG_NCONV; i++) if (VGG[i].tap >= 0) taps[VGG[i].tap].fmt = G.a[i].fmt;
            c->act.off = mark2;                  // stream order: the temporaries of the ground-truth branch are dead before anything below overwrites them
        }
        vgg_forward(c, rec, R, nullptr, true);
        // per-level sums and weights.  w_l = lambda * (l == 0 ? 1 : 2) / 3 / numel_l   (aliasing of level 0 with the total, see the header); numel_l counts ALL B x Trec frames
        float wl[5];
        int li[5];
        for (int i = 0; i < VGG_NCONV; i++) if (VGG[i].tap >= 0) li[VGG[i].tap] = i;
        // formats of the feature gradients gz_i = d/d(pre-ReLU output of conv i), stored at the gradient mirror of a[i]: S16-bf16 when the dgrad of conv i that consumes it runs on an
        // S16-capable launch AND its producer can write it -- the point-wise producers (feature L1 seed of relu5_1, max-pool backward) always can, the dgrad of conv i + 1 when it is
        // such a launch itself
        bool io_d[VGG_NCONV], gzs[VGG_NCONV];
        for (int i = 0; i < VGG_NCONV; i++) {
            const T4& in = VGG[i].pool_before ? R.p[i] : (i > 0 ? R.a[i - 1] : rec);
            io_d[i] = vgg_use_s16(c) && i > 0 && V.conv[i].wqd[0] != nullptr && conv_hx_s16_ok(in.N, in.H, in.W, VGG[i].cin);
        }
        for (int i = 0; i < VGG_NCONV; i++) gzs[i] = io_d[i] && (i + 1 == VGG_NCONV || VGG[i + 1].pool_before || io_d[i + 1]);
        for (int l = 0; l < 5; l++) {
            const T4& f = R.a[li[l]];
            const double numel = (double)B * Trec * f.H * f.W * f.C;
            lv->numel[r][l] = numel;
            wl[l] = (float)(lambda * (l == 0 ? 1.0 : 2.0) / 3.0 / numel);
            if (!dry) launch_feat_l1(st, f, taps[l], wl[l], l == 4 ? f.g : (float*)nullptr, gzs[li[4]], c->loss_acc + LOSS_PERC_R0 + 6 * r + 1 + l);
        }
        if (lambda != 0.0) {
            // backward of the reconstruction branch
            for (int i = VGG_NCONV - 1; i >= 0; i--) {
                VggLayer& L = V.conv[i];
                const T4& gz = R.a[i];
                const T4& in = VGG[i].pool_before ? R.p[i] : (i > 0 ? R.a[i - 1] : rec);
                ConvArgs d{};
                d.src[0] = ConvSrc{gz.g, gz.sn, gz.ld, L.pd.Cout, L.kd, 0};
                d.nsrc = 1; d.N = in.N; d.H = in.H; d.W = in.W; d.KS = 3; d.wp = L.wpd; d.Ktot = L.kd;
                d.Cout = VGG[i].cin; d.Cout_pad = L.cd_pad; d.bias = nullptr; d.act = 0; d.aux = aux;
                d.precision = c->vgg_precision_bwd == PREC_BF16X1 ? PREC_BF16X1 : (c->vgg_precision_bwd == PREC_FP32 ? PREC_FP32 : PREC_BF16X3);
                if (d.precision != PREC_FP32 && L.wqd[0]) d.wq = L.wqd[d.precision == PREC_BF16X1 ? 1 : 0];
                d.out = in.g; d.out_sn = in.sn; d.out_ld = in.ld;
                d.in_s16 = gzs[i] ? 1 : 0;
                if (i == 0) d.accumulate = whole ? 1 : 0;                     // whole batch: += into d(rec_r), next to the L1 seed; a chunk: assigned to the gathered copy, added below
                else if (!VGG[i].pool_before) {                               // direct input a[i-1]: ReLU mask (+ L1 seed when a[i-1] is tapped) in the epilogue; the output IS gz_{i-1}
                    d.mask = in.d; d.mask_s16 = in.fmt;
                    if (VGG[i - 1].tap >= 0) { const T4& tp = taps[VGG[i - 1].tap]; d.seed_ref = tp.d; d.seed_w = wl[VGG[i - 1].tap]; d.seed_s16 = tp.fmt; }
                    d.out_s16 = gzs[i - 1] ? 1 : 0;
                }
                if (!dry) c->ck(conv_call(c, d, 2.0 * in.N * in.H * in.W * 9.0 * VGG[i].cin * VGG[i].cout, 4), "vgg dgrad");
                if (VGG[i].pool_before) {                                     // pooled input: route through the max-pool and the ReLU of a[i-1] (fp32 gradient of the pooled map -> gz_{i-1})
                    if (!dry) launch_maxpool_bwd(st, R.a[i - 1], (const float*)in.g, gzs[i - 1]);
                }
            }
            if (!whole && !dry)      // the chunk's d(rec_r) onto the seeds of its time steps
                hipLaunchKernelGGL(k_time_chunk, dim3(grid_for((long)rec.N * F4)), dim3(256), 0, st, (float4*)full.g, (float4*)rec.g, F4, Trec, t0, len, (long)rec.N * F4, 1);
        }
        if (r != 0 && c->act.off > side_end) side_end = c->act.off;
        if (nch == 1) c->act.off = mark;                    // levels 1 and 2 share memory (same stream, in order)
        if (first_par && r == 0) side_base = c->act.off;
    }
    if (pipe) hipEventRecord(c->perc_ev[k], side);          // the seeds of the time steps [t0, t0 + len) are final (first chunk: its full-resolution level is ordered by the main stream itself)
    }
    c->stream = st;
    if (par) {                                               // join: the tape replay reads d(rec_1), d(rec_2)
        hipEvent_t e = c->sev();
        hipEventRecord(e, side);
        hipStreamWaitEvent(st, e, 0);
    }
    c->act.off = mark_all;
}

// Evaluation: the full-resolution perceptual distance PER RECONSTRUCTED FRAME and feature level -- out_host[l * N + n] = mean |f_l(rec_n) - f_l(gt_n)|, n = b * Trec + t --
// after a forward pass (any mode).  The reference evaluator applies ParallelPerceptualLoss to one sequence position at a time (evaluation/evaluator.py:55,62,193-197,
// training/losses.py:652-713): position t's value is sum_l mean_b out[l][b, t].  Same VGG19 kernels as the training loss; nothing is back-propagated.
int vgg_eval_per_frame(caddy_ctx* c, double* out_host) {
    const caddy_config& g = c->cfg;
    if (!g.perceptual || !c->vgg.loaded) { set_error("caddy_perceptual_per_frame needs a context created with caddy_config.perceptual = 1 and loaded VGG19 weights"); return -2; }
    if (!c->have_forward) { set_error("caddy_perceptual_per_frame: no forward results available"); return -2; }
    const int Trec = c->pretraining ? g.seq_len : g.seq_len - 1, t_off = c->pretraining ? 0 : 1;
    const T4& rec = c->frames[0];
    const int N = rec.N;
    hipStream_t st = c->stream;
    if (c->gt_prefetched) hipStreamWaitEvent(st, c->gt_done, 0);      // (a training-mode forward started the ground-truth branch on the side stream: its scratch lies below fwd_off, untouched here)
    c->act.off = c->fwd_off;
    const size_t mark = c->act.off;
    const int tc[5] = {64, 128, 256, 512, 512};
    T4 gi = valloc(c, N, g.height, g.width, 3);
    const long npix = (long)N * g.height * g.width;
    hipLaunchKernelGGL(k_gt_resize, dim3(grid_for(npix)), dim3(256), 0, st, dv(c->obs), gi.d, g.height, g.width, npix, 1, t_off, g.seq_len, Trec);
    T4 tg[5], tr[5];
    { int h = g.height, w = g.width; for (int l = 0; l < 5; l++) { tg[l] = valloc(c, N, h, w, tc[l]); tr[l] = valloc(c, N, h, w, tc[l]); h /= 2; w /= 2; } }
    double* acc = c->dalloc(5 * (size_t)N);
    hipMemsetAsync(acc, 0, sizeof(double) * 5 * (size_t)N, st);
    const size_t mark2 = c->act.off;
    { Branch G{}; vgg_forward(c, gi, G, tg, false); for (int i = 0; i < VGG_NCONV; i++) if (VGG[i].tap >= 0) tg[VGG[i].tap].fmt = G.a[i].fmt; }
    c->act.off = mark2;
    { Branch R{}; vgg_forward(c, rec, R, tr, false); for (int i = 0; i < VGG_NCONV; i++) if (VGG[i].tap >= 0) tr[VGG[i].tap].fmt = R.a[i].fmt; }
    if (c->act.overflow()) { c->act.off = mark; set_error("caddy_perceptual_per_frame: workspace too small"); return -1; }
    for (int l = 0; l < 5; l++) launch_feat_l1_img(st, tr[l], tg[l], acc + (size_t)l * N);
    hipMemcpyAsync(out_host, acc, sizeof(double) * 5 * (size_t)N, hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    for (int l = 0; l < 5; l++) { const double numel = (double)tr[l].H * tr[l].W * tr[l].C; for (int n = 0; n < N; n++) out_host[(size_t)l * N + n] /= numel; }
    c->act.off = mark;
    return c->fail ? -1 : 0;
}
