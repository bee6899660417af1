// HBM-bound element-wise / reduction kernels of the CADDY hot path, NHWC fp32 views.
// One work item = (pixel, group of 4 channels): float4 accesses, lanes run along the channel axis first so that a
// wave touches contiguous memory.  Per-channel reductions (BatchNorm statistics and their backward sums) are done in
// fp64 per thread -> LDS -> one fp64 atomicAdd per (block, channel), which keeps E[x^2]-E[x]^2 free of cancellation.
// Roofline for everything in this file: HBM (8 TB/s).
#include "common.h"
#include "pointwise.h"
#include "pack.h"

namespace {

// Index arithmetic.  A work item is (pixel q of the (N, H, W) view, channel quad); q -> (sample, pixel in sample) and item -> (q, quad) are divisions by
// run-time constants, and the hardware has no integer divider: hipcc expands every one of them into ~25 instructions (more for 64-bit operands), which made
// the index arithmetic ~half of the instructions these HBM / latency-bound kernels execute (two to four divisions per item).  The host picks a multiplier
// instead: q = (n * m) >> sh, exact for 0 <= n < 2^31 with m = ceil(2^(31+s) / d), s = ceil(log2 d) (error term n * (m d - 2^(31+s)) < 2^(31+s)).
struct FDiv { unsigned m; int sh; int d; };
__device__ __forceinline__ unsigned fdiv(unsigned n, const FDiv& f) { return (unsigned)(((unsigned long long)n * f.m) >> f.sh); }
inline FDiv make_fdiv(int d) {      // host
    if (d < 1) d = 1;
    int b = 0;
    while ((1L << b) < d) b++;
    FDiv f; f.m = (unsigned)((((unsigned long long)1 << (31 + b)) + (unsigned)d - 1) / (unsigned)d); f.sh = 31 + b; f.d = d;
    return f;
}
typedef FDiv PixDiv;      // divisor = H * W of a view: pixel index -> sample
__device__ __forceinline__ long tv_off(const TV& t, const PixDiv& hw, unsigned q) {
    const unsigned n = fdiv(q, hw);
    return (long)n * t.sn + (long)(q - n * (unsigned)hw.d) * t.ld;
}
// FULL: the caller guarantees that C is a multiple of 4 (every quad is complete): no tail test, no divergent paths -- a plain 16-byte access
template <bool FULL = false>
__device__ __forceinline__ float4 ld4(const float* p, int c, int C) {
    if (FULL) return *reinterpret_cast<const float4*>(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c + 4 <= C) v = *reinterpret_cast<const float4*>(p);
    else { if (c < C) v.x = p[0]; if (c + 1 < C) v.y = p[1]; if (c + 2 < C) v.z = p[2]; }
    return v;
}
template <bool FULL = false>
__device__ __forceinline__ void st4(float* p, int c, int C, float4 v) {
    if (FULL) { *reinterpret_cast<float4*>(p) = v; return; }
    if (c + 4 <= C) *reinterpret_cast<float4*>(p) = v;
    else { if (c < C) p[0] = v.x; if (c + 1 < C) p[1] = v.y; if (c + 2 < C) p[2] = v.z; }
}
// ---- S16-bf16 gradient tensors (TV::s16): the channels c .. c + 3 (c a multiple of 4) of the pixel row `row` ----
// split exactly as the loaders of k_conv_hx / k_wgrad_hx split a fp32 gradient (hi = bf16(v), lo = bf16(v - hi), round to nearest even): a consumer that copies these halves
// multiplies the same operands as one that converts the fp32 tensor
__device__ __forceinline__ void st4_s16(float* row, int c, float4 v) {
    bf16x4 hi, lo;
    hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
    lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]); lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
    __bf16* h = reinterpret_cast<__bf16*>(row + (c & ~31)) + (c & 31);
    *reinterpret_cast<bf16x4*>(h) = hi;
    *reinterpret_cast<bf16x4*>(h + 32) = lo;
}
__device__ __forceinline__ float4 ld4_s16(const float* row, int c) {      // hi + lo (16 mantissa bits: what the matrix pipe sees of the gradient)
    const __bf16* h = reinterpret_cast<const __bf16*>(row + (c & ~31)) + (c & 31);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(h), lo = *reinterpret_cast<const bf16x4*>(h + 32);
    return make_float4((float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]);
}
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 operator*(float a, float4 b) { return make_float4(a * b.x, a * b.y, a * b.z, a * b.w); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) { return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)); }
__device__ __forceinline__ float lrelu1(float v) { return v > 0.f ? v : 0.2f * v; }
__device__ __forceinline__ float4 lrelu4(float4 v) { return make_float4(lrelu1(v.x), lrelu1(v.y), lrelu1(v.z), lrelu1(v.w)); }
__device__ __forceinline__ float4 lmask4(float4 o) { return make_float4(o.x > 0.f ? 1.f : 0.2f, o.y > 0.f ? 1.f : 0.2f, o.z > 0.f ? 1.f : 0.2f, o.w > 0.f ? 1.f : 0.2f); }
__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }

template <class F, bool FULL>
__global__ __launch_bounds__(256) void k_map(unsigned items, FDiv C4, F f) {
    for (unsigned it = blockIdx.x * 256u + threadIdx.x; it < items; it += gridDim.x * 256u) {
        const unsigned q = fdiv(it, C4);
        f.template operator()<FULL>(q, (int)(it - q * (unsigned)C4.d) * 4);
    }
}
// full: every channel count the functor touches is a multiple of 4 (see ld4 / st4)
template <class F>
int run_map(long npix, int C, F f, hipStream_t st, bool full = false) {
    int C4 = (C + 3) / 4;
    long items = npix * C4;
    if (items <= 0) return 0;
    if (items >= (1L << 31) - 16384 * 256L) return -1;      // 32-bit item / pixel indices (the largest view of the path, 120 frames of 256 x 256 x 32 channels, has 2^26 items)
    long blocks = (items + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (full && (C & 3) == 0) hipLaunchKernelGGL((k_map<F, true>), dim3((unsigned)blocks), dim3(256), 0, st, (unsigned)items, make_fdiv(C4), f);
    else hipLaunchKernelGGL((k_map<F, false>), dim3((unsigned)blocks), dim3(256), 0, st, (unsigned)items, make_fdiv(C4), f);
    return 0;
}
static inline bool quads(int a, int b = 0, int c = 0, int d = 0) { return ((a | b | c | d) & 3) == 0; }

// ---- functors -------------------------------------------------------------------------------------------------
struct FCopy {
    TV s, d; PixDiv HW; int acc;
    template <bool FULL> __device__ void operator()(unsigned q, int c) const {
        float4 v = ld4<FULL>(s.p + tv_off(s, HW, q) + c, c, s.C);
        float* o = d.p + tv_off(d, HW, q) + c;
        if (acc) v = v + ld4<FULL>(o, c, d.C);
        st4<FULL>(o, c, d.C, v);
    }
};

struct FCopy1 {  // scalar variant for views whose base / pitch is not 16-byte aligned (channel-offset slices)
    TV s, d; PixDiv HW; int acc;
    template <bool FULL> __device__ void operator()(unsigned q, int) const {
        const float* i = s.p + tv_off(s, HW, q);
        float* o = d.p + tv_off(d, HW, q);
        for (int c = 0; c < d.C; c++) o[c] = acc ? o[c] + i[c] : i[c];
    }
};
struct FFill {
    TV d; PixDiv HW; float val;
    template <bool FULL> __device__ void operator()(unsigned q, int c) const { st4<FULL>(d.p + tv_off(d, HW, q) + c, c, d.C, make_float4(val, val, val, val)); }
};
struct FPool2 {  // q indexes OUTPUT pixels (F.avg_pool2d(x, 2): residual_block.py:56, same_block.py:40, representation_network.py:41)
    TV in, out; int act;      // act: LeakyReLU(0.2) on the pooled value (roll-out with the BatchNorm folded into the conv: conv' -> pool -> act)
    PixDiv HWo; FDiv Wo;
    template <bool FULL> __device__ void operator()(unsigned q, int c) const {
        const unsigned n = fdiv(q, HWo); const int rem = (int)(q - n * (unsigned)HWo.d); const int y = (int)fdiv(rem, Wo), x = rem - y * out.W;
        const float* b = in.p + (long)n * in.sn + ((long)(2 * y) * in.W + 2 * x) * in.ld + c;
        float4 v = ld4<FULL>(b, c, in.C) + ld4<FULL>(b + in.ld, c, in.C) + ld4<FULL>(b + (long)in.W * in.ld, c, in.C) + ld4<FULL>(b + (long)(in.W + 1) * in.ld, c, in.C);
        v = 0.25f * v;
        if (act) v = lrelu4(v);
        st4<FULL>(out.p + (long)n * out.sn + (long)rem * out.ld + c, c, out.C, v);
    }
};
struct FPool2Bwd {  // q indexes INPUT pixels; din (+)= dout/4; `assign`: first and only writer of din (no zero-fill, no read)
    TV dout, din; int assign; PixDiv HWi; FDiv Wi;
    template <bool FULL> __device__ void operator()(unsigned q, int c) const {
        const unsigned n = fdiv(q, HWi); const int rem = (int)(q - n * (unsigned)HWi.d); const int y = (int)fdiv(rem, Wi), x = rem - y * din.W;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((y >> 1) < dout.H && (x >> 1) < dout.W)          // odd sizes: the last row / column is not covered by any 2x2 window
            g = ld4<FULL>(dout.p + (long)n * dout.sn + ((long)(y >> 1) * dout.W + (x >> 1)) * dout.ld + c, c, dout.C);
        float* o = din.p + (long)n * din.sn + (long)rem * din.ld;
        if (din.s16) { st4_s16(o, c, 0.25f * g); return; }      // (din_s16: assigning writer of a conv output's gradient)
        st4<FULL>(o + c, c, din.C, assign ? 0.25f * g : ld4<FULL>(o + c, c, din.C) + 0.25f * g);
    }
};
struct FUp2 {  // bilinear x2, align_corners=False (up_block.py:35,43); q indexes OUTPUT pixels
    TV in, out; PixDiv HWo; FDiv Wo;
    template <bool FULL> __device__ void operator()(unsigned q, int c) const {
        const unsigned n = fdiv(q, HWo); const int rem = (int)(q - n * (unsigned)HWo.d); const int y = (int)fdiv(rem, Wo), x = rem - y * out.W;
        int iy = y >> 1, ix = x >> 1;
        int y0, y1, x0, x1; float wy1, wx1;
        if (y & 1) { y0 = iy; y1 = iy + 1 < in.H ? iy + 1 : in.H - 1; wy1 = 0.25f; } else { y0 = iy > 0 ? iy - 1 : 0; y1 = iy; wy1 = iy > 0 ? 0.75f : 0.f; }
        if (x & 1) { x0 = ix; x1 = ix + 1 < in.W ? ix + 1 : in.W - 1; wx1 = 0.25f; } else { x0 = ix > 0 ? ix - 1 : 0; x1 = ix; wx1 = ix > 0 ? 0.75f : 0.f; }
        float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
        const float* b = in.p + (long)n * in.sn + c;
        float4 v00 = ld4<FULL>(b + ((long)y0 * in.W + x0) * in.ld, c, in.C), v01 = ld4<FULL>(b + ((long)y0 * in.W + x1) * in.ld, c, in.C);
        float4 v10 = ld4<FULL>(b + ((long)y1 * in.W + x0) * in.ld, c, in.C), v11 = ld4<FULL>(b + ((long)y1 * in.W + x1) * in.ld, c, in.C);
        float4 r = wy0 * (wx0 * v00 + wx1 * v01) + wy1 * (wx0 * v10 + wx1 * v11);
        st4<FULL>(out.p + (long)n * out.sn + (long)rem * out.ld + c, c, out.C, r);
    }
};
struct FUp2Bwd {  // q indexes INPUT pixels: din (+)= sum_{a,b} w_a w_b dout[clamp(2i-1+a), clamp(2j-1+b)], w = {.25,.75,.75,.25}
    TV dout, din; int assign; PixDiv HWi; FDiv Wi;
    template <bool FULL> __device__ void operator()(unsigned q, int c) const {
        const unsigned n = fdiv(q, HWi); const int rem = (int)(q - n * (unsigned)HWi.d); const int i = (int)fdiv(rem, Wi), j = rem - i * din.W;
        const float w[4] = {0.25f, 0.75f, 0.75f, 0.25f};
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* b = dout.p + (long)n * dout.sn + c;
        for (int a = 0; a < 4; a++) {
            int r = 2 * i - 1 + a; r = r < 0 ? 0 : (r >= dout.H ? dout.H - 1 : r);
            for (int e = 0; e < 4; e++) {
                int s = 2 * j - 1 + e; s = s < 0 ? 0 : (s >= dout.W ? dout.W - 1 : s);
                acc = acc + (w[a] * w[e]) * ld4<FULL>(b + ((long)r * dout.W + s) * dout.ld, c, dout.C);
            }
        }
        float* o = din.p + (long)n * din.sn + (long)rem * din.ld + c;
        st4<FULL>(o, c, din.C, assign ? acc : ld4<FULL>(o, c, din.C) + acc);
    }
};
struct FBnApply {  // out = act(x*scale+shift + second), second = x2*scale2+shift2 | x2 | 0   (residual_block.py:57-68)
    TV x, x2, out; const float *scale, *shift, *scale2, *shift2; PixDiv HW; int has2; int act;
    template <bool FULL> __device__ void operator()(unsigned q, int c) const {
        // (one fused multiply-add per element, exactly what a consuming convolution computes when it applies the BatchNorm itself -- ConvSrc.bn_*: the two forms
        //  of a layer are bit-identical)
        float4 v = fma4(ld4<FULL>(x.p + tv_off(x, HW, q) + c, c, x.C), ld4<FULL>(scale + c, c, x.C), ld4<FULL>(shift + c, c, x.C));
        if (has2) {
            float4 r = ld4<FULL>(x2.p + tv_off(x2, HW, q) + c, c, x2.C);
            if (scale2) r = fma4(r, ld4<FULL>(scale2 + c, c, x.C), ld4<FULL>(shift2 + c, c, x.C));
            v = v + r;
        }
        if (act) v = lrelu4(v);
        st4<FULL>(out.p + tv_off(out, HW, q) + c, c, out.C, v);
    }
};
struct FBnBwdApply {  // dx += gamma*invstd*(dz - s1/M - xhat*s2/M), dz = dout*lrelu'(out)
    TV dout, outm, x, dx; const float *mean, *invstd, *gamma; const double* sums; PixDiv HW; int act; float invM; int assign;
    const float *scale, *shift;      // act without a materialised output (lazily applied BatchNorm, ConvSrc.bn_*): out = x * scale + shift is recomputed for the slope
    template <bool FULL> __device__ void operator()(unsigned q, int c) const {
        float4 dz = ld4<FULL>(dout.p + tv_off(dout, HW, q) + c, c, dout.C);
        float4 xv = ld4<FULL>(x.p + tv_off(x, HW, q) + c, c, x.C);
        if (act) dz = dz * lmask4(scale ? fma4(xv, ld4<FULL>(scale + c, c, x.C), ld4<FULL>(shift + c, c, x.C)) : ld4<FULL>(outm.p + tv_off(outm, HW, q) + c, c, outm.C));
        float4 mu = ld4<FULL>(mean + c, c, x.C), is = ld4<FULL>(invstd + c, c, x.C), ga = ld4<FULL>(gamma + c, c, x.C);
        float s1[4], s2[4];
        for (int e = 0; e < 4; e++) { bool ok = c + e < x.C; s1[e] = ok ? (float)(sums[2 * (c + e)] * invM) : 0.f; s2[e] = ok ? (float)(sums[2 * (c + e) + 1] * invM) : 0.f; }
        float4 xh = (xv - mu) * is;
        float4 g = ga * is * (dz - make_float4(s1[0], s1[1], s1[2], s1[3]) - xh * make_float4(s2[0], s2[1], s2[2], s2[3]));
        float* o = dx.p + tv_off(dx, HW, q);
        if (dx.s16) { st4_s16(o, c, g); return; }      // (dx_s16: assigning writer of a conv output's gradient)
        st4<FULL>(o + c, c, dx.C, assign ? g : ld4<FULL>(o + c, c, dx.C) + g);      // assign: dx has no other writer (conv output consumed by this BatchNorm only)
    }
};
struct FActBwdAdd {  // dres (+)= dout * lrelu'(out)
    TV dout, outm, dres; PixDiv HW; int assign;
    template <bool FULL> __device__ void operator()(unsigned q, int c) const {
        float4 dz = ld4<FULL>(dout.p + tv_off(dout, HW, q) + c, c, dout.C) * lmask4(ld4<FULL>(outm.p + tv_off(outm, HW, q) + c, c, outm.C));
        float* o = dres.p + tv_off(dres, HW, q) + c;
        st4<FULL>(o, c, dres.C, assign ? dz : ld4<FULL>(o, c, dres.C) + dz);
    }
};
struct FLstmFwd {  // gates (pre-activation, channel order [i|f|o|g] x C) -> post-activation in place; c' = f*c + i*g; h' = o*tanh(c')
    TV gates, cprev, h, cn; PixDiv HW;   // convolutional_lstm_cell.py:92-101
    TV hb; const float *scale, *shift;      // optional (roll-out): hb = h' * scale + shift, the eval-mode BatchNorm that follows the cell (conv_dynamics_network.py)
    template <bool FULL> __device__ void operator()(unsigned q, int c) const {
        int C = h.C;
        float* gp = gates.p + tv_off(gates, HW, q) + c;
        float4 gi = ld4<FULL>(gp, c, C), gf = ld4<FULL>(gp + C, c, C), go = ld4<FULL>(gp + 2 * C, c, C), gg = ld4<FULL>(gp + 3 * C, c, C);
        float4 cp = ld4<FULL>(cprev.p + tv_off(cprev, HW, q) + c, c, C);
        float4 i4 = make_float4(sigm(gi.x), sigm(gi.y), sigm(gi.z), sigm(gi.w));
        float4 f4 = make_float4(sigm(gf.x), sigm(gf.y), sigm(gf.z), sigm(gf.w));
        float4 o4 = make_float4(sigm(go.x), sigm(go.y), sigm(go.z), sigm(go.w));
        float4 g4 = make_float4(tanhf(gg.x), tanhf(gg.y), tanhf(gg.z), tanhf(gg.w));
        float4 cc = f4 * cp + i4 * g4;
        float4 hh = o4 * make_float4(tanhf(cc.x), tanhf(cc.y), tanhf(cc.z), tanhf(cc.w));
        st4<FULL>(gp, c, C, i4); st4<FULL>(gp + C, c, C, f4); st4<FULL>(gp + 2 * C, c, C, o4); st4<FULL>(gp + 3 * C, c, C, g4);
        st4<FULL>(cn.p + tv_off(cn, HW, q) + c, c, C, cc);
        st4<FULL>(h.p + tv_off(h, HW, q) + c, c, C, hh);
        if (scale) st4<FULL>(hb.p + tv_off(hb, HW, q) + c, c, C, hh * ld4<FULL>(scale + c, c, C) + ld4<FULL>(shift + c, c, C));
    }
};
struct FLstmBwd {
    TV gates, cprev, cn, dh, dc, dgates, dcprev; PixDiv HW;
    template <bool FULL> __device__ void operator()(unsigned q, int c) const {
        int C = dh.C;
        const float* gp = gates.p + tv_off(gates, HW, q) + c;
        float4 i4 = ld4<FULL>(gp, c, C), f4 = ld4<FULL>(gp + C, c, C), o4 = ld4<FULL>(gp + 2 * C, c, C), g4 = ld4<FULL>(gp + 3 * C, c, C);
        float4 cp = ld4<FULL>(cprev.p + tv_off(cprev, HW, q) + c, c, C), cc = ld4<FULL>(cn.p + tv_off(cn, HW, q) + c, c, C);
        float4 gh = ld4<FULL>(dh.p + tv_off(dh, HW, q) + c, c, C), gc = ld4<FULL>(dc.p + tv_off(dc, HW, q) + c, c, C);
        float4 tc = make_float4(tanhf(cc.x), tanhf(cc.y), tanhf(cc.z), tanhf(cc.w));
        float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
        float4 d_o = gh * tc;
        float4 dcc = gc + gh * o4 * (one - tc * tc);
        float4 d_i = dcc * g4, d_f = dcc * cp, d_g = dcc * i4;
        float* dg = dgates.p + tv_off(dgates, HW, q) + c;
        const float4 gi_ = d_i * i4 * (one - i4), gf_ = d_f * f4 * (one - f4), go_ = d_o * o4 * (one - o4), gg_ = d_g * (one - g4 * g4);      // (formed once: both stores see the same bits)
        if (dgates.s16) {      // (dgates_s16: the gate convolution's dY; 4 C channels, C a multiple of 32)
            float* row = dg - c;
            st4_s16(row, c, gi_); st4_s16(row, C + c, gf_); st4_s16(row, 2 * C + c, go_); st4_s16(row, 3 * C + c, gg_);
        } else {
            st4<FULL>(dg, c, C, gi_); st4<FULL>(dg + C, c, C, gf_); st4<FULL>(dg + 2 * C, c, C, go_); st4<FULL>(dg + 3 * C, c, C, gg_);
        }
        float* dp = dcprev.p + tv_off(dcprev, HW, q) + c;
        st4<FULL>(dp, c, C, ld4<FULL>(dp, c, C) + dcc * f4);
    }
};
struct FTanhBwd {
    TV dy, y, dz; PixDiv HW;
    template <bool FULL> __device__ void operator()(unsigned q, int c) const {
        float4 yv = ld4<FULL>(y.p + tv_off(y, HW, q) + c, c, y.C);
        float4 g = ld4<FULL>(dy.p + tv_off(dy, HW, q) + c, c, dy.C) * (make_float4(1.f, 1.f, 1.f, 1.f) - yv * yv);
        st4<FULL>(dz.p + tv_off(dz, HW, q) + c, c, dz.C, g);
    }
};
struct FAttnMul {  // attentive = state * sigmoid(x[..., C-1])  (representation_network.py:47-57, action_network.py:78)
    TV x, out, att; PixDiv HW;
    template <bool FULL> __device__ void operator()(unsigned q, int c) const {
        const float* xp = x.p + tv_off(x, HW, q);
        float a = sigm(xp[x.C - 1]);
        st4<FULL>(out.p + tv_off(out, HW, q) + c, c, out.C, a * ld4<FULL>(xp + c, c, out.C));
        if (c == 0 && att.p) att.p[tv_off(att, HW, q)] = a;
    }
};
// attention gate backward (action_network.py: x[:, :-1] * sigmoid(x[:, -1:])): 16 lanes per pixel, each a float4 of channels
// (coalesced), the per-pixel dot product d_att = sum_c dout[c] * x[c] reduced with width-16 shuffles.
struct AttnBwdArgs { TV x, dout, datt, dx; PixDiv HW; long npix; };
__global__ __launch_bounds__(256) void k_attn_mul_bwd(AttnBwdArgs a) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    const unsigned q = (unsigned)(gid >> 4);
    const int j = (int)(gid & 15);
    const bool ok = q < a.npix;
    const int Cs = a.x.C - 1;
    float dot = 0.f, att = 0.f;
    const float *xp = nullptr, *gp = nullptr; float* dp = nullptr;
    if (ok) {
        xp = a.x.p + tv_off(a.x, a.HW, q); gp = a.dout.p + tv_off(a.dout, a.HW, q); dp = a.dx.p + tv_off(a.dx, a.HW, q);
        att = sigm(xp[Cs]);
        for (int c = j * 4; c < Cs; c += 64) {
            float4 g = ld4(gp + c, c, Cs), xv = ld4(xp + c, c, Cs);
            dot += g.x * xv.x + g.y * xv.y + g.z * xv.z + g.w * xv.w;
            st4(dp + c, c, Cs, ld4(dp + c, c, Cs) + att * g);
        }
    }
    for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    if (ok && j == 0) {
        if (a.datt.p) dot += a.datt.p[tv_off(a.datt, a.HW, q)];
        dp[Cs] += dot * att * (1.f - att);
    }
}
struct FGapBwd {
    TV dx; const float* dout; PixDiv HW; float inv;
    template <bool FULL> __device__ void operator()(unsigned q, int c) const {
        const unsigned n = fdiv(q, HW);
        float* o = dx.p + tv_off(dx, HW, q) + c;
        st4<FULL>(o, c, dx.C, ld4<FULL>(o, c, dx.C) + inv * ld4<FULL>(dout + (long)n * dx.C + c, c, dx.C));
    }
};
struct FNchwToNhwc {  // per pixel; pad channels [C, ld) are zero-filled
    const float* src; long src_sn; TV d; PixDiv HW;
    template <bool FULL> __device__ void operator()(unsigned q, int) const {
        const long n = fdiv(q, HW); const long pix = q - n * HW.d;
        float* o = d.p + n * d.sn + pix * d.ld;
        for (int c = 0; c < d.C; c++) o[c] = src[n * src_sn + (long)c * HW.d + pix];
        for (int c = d.C; c < d.ld && c < ((d.C + 3) & ~3); c++) o[c] = 0.f;
    }
};
struct FNhwcToNchw {
    TV s; float* dst; long dst_sn; PixDiv HW; int acc;
    template <bool FULL> __device__ void operator()(unsigned q, int) const {
        const long n = fdiv(q, HW); const long pix = q - n * HW.d;
        const float* i = s.p + n * s.sn + pix * s.ld;
        for (int c = 0; c < s.C; c++) {
            float v = i[c];
            if (s.s16) { const __bf16* h = reinterpret_cast<const __bf16*>(i + (c & ~31)) + (c & 31); v = (float)h[0] + (float)h[32]; }      // (s_s16: debug read-out of a pre-split gradient)
            float* o = dst + n * dst_sn + (long)c * HW.d + pix; *o = acc ? *o + v : v;
        }
    }
};

// ---- per-channel reductions -------------------------------------------------------------------------------------
// MODE 0: sums[c][0..1] += (sum x, sum x^2)                         (BatchNorm batch statistics)
// MODE 1: sums[c][0..1] += (sum dz, sum dz*xhat), dz = dout*mask     (BatchNorm backward)
// MODE 2: outf[(n*out_sn) + c] += scale * sum x                      (per-sample spatial sum: GAP, broadcast-input grads)
// MODE 3: outf[c] += sum x                                           (bias gradient)
struct RedArgs {
    TV x, dout, outm; const float *mean, *invstd; double* sums; float* outf; long out_sn; float scale; int act; int pix_per_block;
    float *dgamma, *dbeta;   // MODE 1 + partials: fused parameter gradients
    double* partials;   // MODE 0/1: when set, block b writes its sums to partials[b][2C] (no atomics); k_sum_partials folds them
    const float *lz_scale, *lz_shift;   // MODE 1, act without a materialised output: the LeakyReLU slope is taken from x * scale + shift (lazily applied BatchNorm)
    PixDiv hw;                          // set by run_reduce: H * W of x
    int det;                            // MODE 2 / 3: one workgroup per output vector (the float atomics then have a single contributor each: bit-reproducible)
};
// ACT (MODE 1): 0 = no activation between the BatchNorm and the gradient, 1 = LeakyReLU slope from the materialised output `outm`, 2 = from x * scale + shift (lazily
// applied BatchNorm) -- a template parameter so that no branch stands between the loads of a trip: four pixels per trip, all their loads issued before the first use
// (one pixel per trip left every thread with two or three outstanding loads, i.e. latency-bound at a fraction of the HBM rate)
template <int MODE, bool FULL, int ACT>
__global__ __launch_bounds__(256) void k_reduce(RedArgs a) {
    __shared__ double sh[256 * 8];
    const int C = a.x.C, C4 = (C + 3) / 4;
    const int C4b = C4 < 256 ? C4 : 256;
    const int PT = 256 / C4b;
    const int tid = threadIdx.x, pt = tid / C4b, j = tid - pt * C4b;
    const int HW = a.x.H * a.x.W;
    // MODE 2 reduces within one sample: blockIdx.y = sample
    long qbeg, qend;
    if (MODE == 2) { qbeg = (long)blockIdx.y * HW + (long)blockIdx.x * a.pix_per_block; long e = qbeg + a.pix_per_block; long lim = (long)(blockIdx.y + 1) * HW; qend = e < lim ? e : lim; }
    else { long P = (long)a.x.N * HW; qbeg = (long)blockIdx.x * a.pix_per_block; long e = qbeg + a.pix_per_block; qend = e < P ? e : P; }
    double s[8];
    for (int e = 0; e < 8; e++) s[e] = 0.0;
    constexpr int U = 4;
    if (pt < PT && qbeg < qend) {
        const unsigned qlast = (unsigned)qend - 1;
        for (int jj = j; jj < C4; jj += C4b) {   // C4 <= 256 in practice -> single trip
            int c = jj * 4;
            float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu, lsc = mu, lsh = mu;
            if (MODE == 1) { mu = ld4<FULL>(a.mean + c, c, C); is = ld4<FULL>(a.invstd + c, c, C); }
            if (MODE == 1 && ACT == 2) { lsc = ld4<FULL>(a.lz_scale + c, c, C); lsh = ld4<FULL>(a.lz_shift + c, c, C); }
            for (unsigned q0 = (unsigned)qbeg + pt; q0 <= qlast; q0 += U * PT) {
                float4 xv[U], dz[U], om[U];
#pragma unroll
                for (int u = 0; u < U; u++) {      // loads only; rows past the end re-read the last pixel and are dropped below
                    const unsigned q = q0 + u * PT <= qlast ? q0 + u * PT : qlast;
                    if (MODE == 3 && a.x.s16) xv[u] = ld4_s16(a.x.p + tv_off(a.x, a.hw, q), c);      // (x_s16: bias gradient = column sums of a pre-split dY)
                    else xv[u] = ld4<FULL>(a.x.p + tv_off(a.x, a.hw, q) + c, c, C);
                    if (MODE == 1) dz[u] = ld4<FULL>(a.dout.p + tv_off(a.dout, a.hw, q) + c, c, C);
                    if (MODE == 1 && ACT == 1) om[u] = ld4<FULL>(a.outm.p + tv_off(a.outm, a.hw, q) + c, c, C);
                }
#pragma unroll
                for (int u = 0; u < U; u++) {      // (selects, not branches: a branch per row makes hipcc sink that row's loads behind it)
                    const float w = q0 + u * PT <= qlast ? 1.f : 0.f;
                    if (MODE == 0) {
                        const float4 v = w * xv[u];
                        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
                        s[4] += (double)v.x * v.x; s[5] += (double)v.y * v.y; s[6] += (double)v.z * v.z; s[7] += (double)v.w * v.w;
                    } else if (MODE == 1) {
                        float4 g = w * dz[u];
                        if (ACT == 1) g = g * lmask4(om[u]);
                        if (ACT == 2) g = g * lmask4(fma4(xv[u], lsc, lsh));
                        const float4 xh = (xv[u] - mu) * is;
                        s[0] += g.x; s[1] += g.y; s[2] += g.z; s[3] += g.w;
                        s[4] += (double)g.x * xh.x; s[5] += (double)g.y * xh.y; s[6] += (double)g.z * xh.z; s[7] += (double)g.w * xh.w;
                    } else {
                        const float4 v = w * xv[u];
                        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
                    }
                }
            }
        }
    }
    for (int e = 0; e < 8; e++) sh[tid * 8 + e] = s[e];
    __syncthreads();
    // fold the PT pixel rows of the block: a tree over `pt` (with 16 .. 64-channel tensors PT is 16 .. 64: one thread per channel quad walking them
    // serially was ~10 us of dependent LDS reads at the end of every launch -- more than the streaming part of E's / D's per-time-step maps)
    int top = 1;
    while (top < PT) top <<= 1;
    for (int half = top >> 1; half >= 1; half >>= 1) {
        if (pt < half && pt + half < PT)
            for (int e = 0; e < 8; e++) sh[tid * 8 + e] += sh[((pt + half) * C4b + j) * 8 + e];
        __syncthreads();
    }
    if (pt == 0 && j < C4) {
        for (int e = 0; e < 8; e++) s[e] = sh[j * 8 + e];
        int c = j * 4;
        for (int e = 0; e < 4; e++) {
            if (c + e >= C) break;
            if (MODE <= 1) {
                if (a.partials) { double* pp = a.partials + (long)blockIdx.x * 2 * C; pp[2 * (c + e)] = s[e]; pp[2 * (c + e) + 1] = s[4 + e]; }
                else { atomicAdd(&a.sums[2 * (c + e)], s[e]); atomicAdd(&a.sums[2 * (c + e) + 1], s[4 + e]); }
            }
            else if (MODE == 2) atomicAdd(&a.outf[(long)blockIdx.y * a.out_sn + c + e], (float)(s[e] * a.scale));
            else if (a.partials) a.partials[(long)blockIdx.x * C + c + e] = s[e];      // bit-reproducible bias gradient: per-block sums, folded in block order by k_sum_partials
            else atomicAdd(&a.outf[c + e], (float)s[e]);
        }
    }
}

// optional BatchNorm finalisation fused into the fold of the forward statistics (saves one launch per BatchNorm call)
struct BnFin { double count; const float *gamma, *beta; float *rmean, *rvar; int C; float momentum, eps; float *mean, *invstd, *scale, *shift; };

__device__ __forceinline__ void bn_finalize_channel(const BnFin& f, int c, double sum, double sq, int training) {
    float m, is;
    if (training) {
        double mu = sum / f.count;
        double var = sq / f.count - mu * mu;
        if (var < 0) var = 0;
        m = (float)mu;
        is = (float)(1.0 / sqrt(var + (double)f.eps));
        double unb = f.count > 1 ? var * f.count / (f.count - 1) : var;
        if (f.rmean) {
            f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * m;
            f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * (float)unb;
        } else if (f.rvar) f.rvar[c] = (float)unb;      // deferred running statistics (rmean == nullptr): leave the unbiased variance of THIS call for k_bn_ema
    } else {
        m = f.rmean[c];
        is = 1.f / sqrtf(f.rvar[c] + f.eps);
    }
    float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
    f.mean[c] = m; f.invstd[c] = is; f.scale[c] = g * is; f.shift[c] = b - m * g * is;
}

__global__ __launch_bounds__(256) void k_sum_partials(const double* partials, int nb, int n2c, double* sums, float* dgamma, float* dbeta, BnFin fin, float* addf = nullptr) {
    // sums[i] = sum_b partials[b][i] (assign: no memset needed); one wave per output index (4 per workgroup), lanes stride over blocks.
    // Optionally fused BatchNorm parameter gradients: dbeta[c] += sums[2c], dgamma[c] += sums[2c+1].
    __shared__ double sh[4];
    int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    double s = 0.0;
    if (i < n2c) for (int b = lane; b < nb; b += 64) s += partials[(long)b * n2c + i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (i < n2c && lane == 0) {
        if (addf) addf[i] += (float)s;      // (MODE 3: the bias gradient accumulates over the time steps, one writer per element and launch)
        else sums[i] = s;
        if (dgamma) { if (i & 1) dgamma[i >> 1] += (float)s; else dbeta[i >> 1] += (float)s; }
    }
    if (fin.mean) {                                   // workgroup b owns channels 2b, 2b+1 (indices 4b .. 4b+3)
        if (lane == 0) sh[threadIdx.x >> 6] = s;
        __syncthreads();
        int c = blockIdx.x * 2 + threadIdx.x;
        if (threadIdx.x < 2 && c < fin.C) bn_finalize_channel(fin, c, sh[2 * threadIdx.x], sh[2 * threadIdx.x + 1], 1);
    }
}

// BatchNorm statistics from the per-tile partial sums a convolution's epilogue wrote (ConvArgs.stats: part[(tile * ldp + c) * 2 + {0, 1}] = {sum, sum of squares}
// over the tile's valid pixels, fp32): one wave per channel, lanes stride over the tiles in a FIXED order, fp64 from here on (E[x^2] - E[x]^2), then the
// usual finalisation (running statistics, mean / invstd / scale / shift).  Replaces a full read of the conv output (k_reduce<0>) + k_sum_partials.
__global__ __launch_bounds__(256) void k_bn_finalize_tiles(const float* part, int ntiles, int ldp, BnFin fin) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= fin.C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int t = lane; t < ntiles; t += 64) { const float* p = part + ((long)t * ldp + c) * 2; s1 += (double)p[0]; s2 += (double)p[1]; }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) bn_finalize_channel(fin, c, s1, s2, 1);
}

template <int MODE>
int run_reduce(RedArgs a, hipStream_t st, const BnFin* fin = nullptr) {
    int HW = a.x.H * a.x.W;
    long P = (long)a.x.N * HW;
    if ((a.x.C + 3) / 4 > 256 || P >= (1L << 31) - 4096) return -1;
    a.hw = make_fdiv(HW);
    const bool full = (a.x.C & 3) == 0;      // (dout / outm share x's channel count)
    if (MODE == 2) {
        int ppb = (HW > 4096 && !a.det) ? 4096 : HW;
        a.pix_per_block = ppb;
        if (full) hipLaunchKernelGGL((k_reduce<MODE, true, 0>), dim3(cdiv(HW, ppb), a.x.N), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_reduce<MODE, false, 0>), dim3(cdiv(HW, ppb), a.x.N), dim3(256), 0, st, a);
    } else {
        // (MODE 3, bit-reproducible: per-block partial sums + fixed-order fold when the caller provides scratch; the former single-workgroup form took 460 us per launch on D's
        //  full-resolution maps -- 20 of the 29 ms the mode added to the serialised step)
        long maxb = ((MODE <= 1 || MODE == 3) && a.partials) ? RED_MAX_BLOCKS : ((MODE == 3 && a.det) ? 1 : 1024);
        long ppb = (P + maxb - 1) / maxb;
        int C4r = (a.x.C + 3) / 4; int ptr = 256 / (C4r < 256 ? C4r : 256);      // pixel rows handled in parallel by one block
        long minp = ptr * 4 > 16 ? ptr * 4 : 16;                                 // >= 4 pixels per thread
        if (ppb < minp) ppb = minp;
        a.pix_per_block = (int)ppb;
        int nb = cdiv(P, ppb);
        const int actm = MODE == 1 ? (a.act ? (a.lz_scale ? 2 : 1) : 0) : 0;
#define RED_LAUNCH(F_, A_) hipLaunchKernelGGL((k_reduce<MODE, F_, (MODE == 1 ? A_ : 0)>), dim3(nb), dim3(256), 0, st, a)
        if (full) { if (actm == 2) RED_LAUNCH(true, 2); else if (actm == 1) RED_LAUNCH(true, 1); else RED_LAUNCH(true, 0); }
        else { if (actm == 2) RED_LAUNCH(false, 2); else if (actm == 1) RED_LAUNCH(false, 1); else RED_LAUNCH(false, 0); }
#undef RED_LAUNCH
        BnFin nofin{}; nofin.mean = nullptr;
        if (MODE <= 1 && a.partials) hipLaunchKernelGGL(k_sum_partials, dim3(cdiv(2 * a.x.C, 4)), dim3(256), 0, st, (const double*)a.partials, nb, 2 * a.x.C, a.sums, a.dgamma, a.dbeta, fin ? *fin : nofin, (float*)nullptr);
        if (MODE == 3 && a.partials) hipLaunchKernelGGL(k_sum_partials, dim3(cdiv(a.x.C, 4)), dim3(256), 0, st, (const double*)a.partials, nb, a.x.C, (double*)nullptr, (float*)nullptr, (float*)nullptr, nofin, a.outf);
    }
    return 0;
}

// BN statistics -> (mean, invstd, scale, shift) + running-stat update (nn.BatchNorm2d train / eval semantics)
// ---- fused BatchNorm for small feature maps (R's 16x16 / 32x32 maps: <= 8192 pixels) ------------------------------------------
// The generic path is 4 launches forward (partial sums, fold+finalise, apply) and 3-4 backward, each 5-8 us of pure launch
// latency on these sizes.  Here ONE workgroup owns four channels and ALL pixels: every thread keeps its <= 32 pixels (float4) in
// registers, the block reduces in fp64 (shuffles + LDS), finalises, and applies from registers -- one launch, one read of x.
constexpr int BNS_PPT = 32;                  // pixels per thread -> up to 256 * 32 = 8192 pixels
struct BnSmallFwd { TV x, x2, out; int has2, act; BnFin fin; PixDiv hw; };
struct BnSmallBwd { TV dout, outm, x, dx, dres; int act, has_res; const float *mean, *invstd, *gamma; float *dgamma, *dbeta; int assign; int res_assign; PixDiv hw; };

__device__ __forceinline__ void block_reduce8(double* s, double* sh, int tid) {   // result valid for all threads in sh[0..7]
#pragma unroll
    for (int e = 0; e < 8; e++)
        for (int o = 32; o > 0; o >>= 1) s[e] += __shfl_xor(s[e], o);
    if ((tid & 63) == 0) for (int e = 0; e < 8; e++) sh[(tid >> 6) * 8 + e] = s[e];
    __syncthreads();
    if (tid < 8) sh[32 + tid] = sh[tid] + sh[8 + tid] + sh[16 + tid] + sh[24 + tid];
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_bn_small_fwd(BnSmallFwd a) {
    __shared__ double sh[40];
    __shared__ float ss[8];
    const int tid = threadIdx.x, c = blockIdx.x * 4, C = a.x.C;
    const PixDiv HW = a.hw;
    const unsigned P = (unsigned)a.x.N * (unsigned)HW.d;
    float4 v[BNS_PPT];
    double s[8];
#pragma unroll
    for (int e = 0; e < 8; e++) s[e] = 0.0;
#pragma unroll
    for (int i = 0; i < BNS_PPT; i++) {
        const unsigned q = tid + 256u * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < P) {
            v[i] = ld4(a.x.p + tv_off(a.x, HW, q) + c, c, C);
            s[0] += v[i].x; s[1] += v[i].y; s[2] += v[i].z; s[3] += v[i].w;
            s[4] += (double)v[i].x * v[i].x; s[5] += (double)v[i].y * v[i].y; s[6] += (double)v[i].z * v[i].z; s[7] += (double)v[i].w * v[i].w;
        }
    }
    block_reduce8(s, sh, tid);
    if (tid < 4 && c + tid < C) {
        bn_finalize_channel(a.fin, c + tid, sh[32 + tid], sh[36 + tid], 1);
        ss[tid] = a.fin.scale[c + tid]; ss[4 + tid] = a.fin.shift[c + tid];
    }
    __syncthreads();
    const float4 sc = make_float4(ss[0], ss[1], ss[2], ss[3]), sf = make_float4(ss[4], ss[5], ss[6], ss[7]);
#pragma unroll
    for (int i = 0; i < BNS_PPT; i++) {
        const unsigned q = tid + 256u * i;
        if (q < P) {
            float4 o = v[i] * sc + sf;
            if (a.has2) o = o + ld4(a.x2.p + tv_off(a.x2, HW, q) + c, c, C);
            if (a.act) o = lrelu4(o);
            st4(a.out.p + tv_off(a.out, HW, q) + c, c, C, o);
        }
    }
}

__global__ __launch_bounds__(256) void k_bn_small_bwd(BnSmallBwd a) {
    __shared__ double sh[40];
    const int tid = threadIdx.x, c = blockIdx.x * 4, C = a.x.C;
    const PixDiv HW = a.hw;
    const unsigned P = (unsigned)a.x.N * (unsigned)HW.d;
    const float4 mu = ld4(a.mean + c, c, C), is = ld4(a.invstd + c, c, C), ga = ld4(a.gamma + c, c, C);
    float4 dz[BNS_PPT];
    double s[8];
#pragma unroll
    for (int e = 0; e < 8; e++) s[e] = 0.0;
#pragma unroll
    for (int i = 0; i < BNS_PPT; i++) {
        const unsigned q = tid + 256u * i;
        dz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < P) {
            dz[i] = ld4(a.dout.p + tv_off(a.dout, HW, q) + c, c, C);
            if (a.act) dz[i] = dz[i] * lmask4(ld4(a.outm.p + tv_off(a.outm, HW, q) + c, c, C));
            float4 xh = (ld4(a.x.p + tv_off(a.x, HW, q) + c, c, C) - mu) * is;
            s[0] += dz[i].x; s[1] += dz[i].y; s[2] += dz[i].z; s[3] += dz[i].w;
            s[4] += (double)dz[i].x * xh.x; s[5] += (double)dz[i].y * xh.y; s[6] += (double)dz[i].z * xh.z; s[7] += (double)dz[i].w * xh.w;
        }
    }
    block_reduce8(s, sh, tid);
    if (tid < 4 && c + tid < C) { a.dbeta[c + tid] += (float)sh[32 + tid]; a.dgamma[c + tid] += (float)sh[36 + tid]; }
    const double invM = 1.0 / (double)P;
    const float4 s1 = make_float4((float)(sh[32] * invM), (float)(sh[33] * invM), (float)(sh[34] * invM), (float)(sh[35] * invM));
    const float4 s2 = make_float4((float)(sh[36] * invM), (float)(sh[37] * invM), (float)(sh[38] * invM), (float)(sh[39] * invM));
#pragma unroll
    for (int i = 0; i < BNS_PPT; i++) {
        const unsigned q = tid + 256u * i;
        if (q < P) {
            float4 xh = (ld4(a.x.p + tv_off(a.x, HW, q) + c, c, C) - mu) * is;
            float4 g = ga * is * (dz[i] - s1 - xh * s2);
            float* o = a.dx.p + tv_off(a.dx, HW, q);
            if (a.dx.s16) st4_s16(o, c, g);      // (dx_s16)
            else st4(o + c, c, C, a.assign ? g : ld4(o + c, c, C) + g);
            if (a.has_res) { float* r = a.dres.p + tv_off(a.dres, HW, q) + c; st4(r, c, C, a.res_assign ? dz[i] : ld4(r, c, C) + dz[i]); }
        }
    }
}

__global__ void k_bn_finalize(const double* sums, BnFin f, int training) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= f.C) return;
    bn_finalize_channel(f, c, training ? sums[2 * c] : 0.0, training ? sums[2 * c + 1] : 0.0, training);
}
// deferred momentum updates of one BatchNorm layer, applied in call order (same float arithmetic as bn_finalize_channel, so the result is bit-identical to
// updating inside every call): the calls of D's BatchNorms execute on two streams (caddy_ctx::dstream) and must not race on / reorder the running statistics
struct EmaArgs { const float* mean[32]; const float* uvar[32]; int n, C; float momentum; float *rmean, *rvar; };
__global__ void k_bn_ema(EmaArgs a) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.C) return;
    float m = a.rmean[c], v = a.rvar[c];
    for (int i = 0; i < a.n; i++) { m = (1.f - a.momentum) * m + a.momentum * a.mean[i][c]; v = (1.f - a.momentum) * v + a.momentum * a.uvar[i][c]; }
    a.rmean[c] = m; a.rvar[c] = v;
}
__global__ void k_bn_param_grad(const double* sums, int C, float* dgamma, float* dbeta) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbeta[c] += (float)sums[2 * c];
    dgamma[c] += (float)sums[2 * c + 1];
}
__global__ void k_batch_sum(const float* src, long sn, long n_el, int N, float* dst) {  // dst[i] += sum_n src[n*sn+i]
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n_el) return;
    float s = 0.f;
    for (int n = 0; n < N; n++) s += src[n * sn + i];
    dst[i] += s;
}

// ---- gradient of a spatially-broadcast conv input (the action / variation vectors of R, conv_dynamics_network.py:77-109) ----
// forward: y[p,o] += sum_tap W[o,c,tap] * a[c] * [p+off(tap) in bounds]   =>   da[c] = sum_{o,tap} W[o,c,tap] * S[o][tap],
// S[o][tap] = sum over the pixels whose tap stays in bounds = total - excluded border row/column + doubly excluded corner.
__global__ __launch_bounds__(256) void k_border_sums(TV dz, float* S) {   // S[n][c][9]; grid (N, ceil(C4/16)): 16 channel quads x 16 pixel groups
    __shared__ float4 sh[9][256];
    const int cq = threadIdx.x & 15, pg = threadIdx.x >> 4;
    const int n = blockIdx.x, c = (blockIdx.y * 16 + cq) * 4;
    const int H = dz.H, W = dz.W, HW = H * W;
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 T = z, R0 = z, RL = z, C0 = z, CL = z, K00 = z, K0L = z, KL0 = z, KLL = z;
    if (c < dz.C) {
        for (int p = pg; p < HW; p += 16) {
            int y = p / W, x = p - y * W;
            const float* row = dz.p + (long)n * dz.sn + (long)p * dz.ld;
            float4 v = dz.s16 ? ld4_s16(row, c) : ld4(row + c, c, dz.C);      // (dz_s16)
            T = T + v;
            if (y == 0) { R0 = R0 + v; if (x == 0) K00 = K00 + v; if (x == W - 1) K0L = K0L + v; }
            if (y == H - 1) { RL = RL + v; if (x == 0) KL0 = KL0 + v; if (x == W - 1) KLL = KLL + v; }
            if (x == 0) C0 = C0 + v;
            if (x == W - 1) CL = CL + v;
        }
    }
    sh[0][threadIdx.x] = T; sh[1][threadIdx.x] = R0; sh[2][threadIdx.x] = RL; sh[3][threadIdx.x] = C0; sh[4][threadIdx.x] = CL;
    sh[5][threadIdx.x] = K00; sh[6][threadIdx.x] = K0L; sh[7][threadIdx.x] = KL0; sh[8][threadIdx.x] = KLL;
    __syncthreads();
    if (threadIdx.x < 16 * 9 && (blockIdx.y * 16 + (threadIdx.x & 15)) * 4 < dz.C) {   // thread = (category k, channel quad)
        const int k = threadIdx.x >> 4, q = threadIdx.x & 15;
        float4 v = z;
        for (int g = 0; g < 16; g++) v = v + sh[k][g * 16 + q];
        sh[k][q] = v;                                   // slot (k, q) is only read by this thread
    }
    __syncthreads();
    if (pg == 0 && c < dz.C) {
        float4 v[9];
        for (int k = 0; k < 9; k++) v[k] = sh[k][cq];
        for (int ty = 0; ty < 3; ty++)
            for (int tx = 0; tx < 3; tx++) {
                float4 r = v[0];
                if (ty == 0) r = r - v[1];
                if (ty == 2) r = r - v[2];
                if (tx == 0) r = r - v[3];
                if (tx == 2) r = r - v[4];
                if (ty == 0 && tx == 0) r = r + v[5];
                if (ty == 0 && tx == 2) r = r + v[6];
                if (ty == 2 && tx == 0) r = r + v[7];
                if (ty == 2 && tx == 2) r = r + v[8];
                float rr[4] = {r.x, r.y, r.z, r.w};
                for (int e = 0; e < 4 && c + e < dz.C; e++) S[((long)n * dz.C + c + e) * 9 + ty * 3 + tx] = rr[e];
            }
    }
}
__global__ __launch_bounds__(256) void k_bcast_grad(PackDesc d, int seg, const float* S, float* g, long g_sn) {   // grid (N, seg_C)
    __shared__ float sh[256];
    const int n = blockIdx.x, c = blockIdx.y;
    float acc = 0.f;
    for (int i = threadIdx.x; i < d.Cout * 9; i += 256) {
        int o = i / 9, tap = i - o * 9;
        acc += d.w[o / d.Co_each][((long)(o % d.Co_each) * d.Cin + d.seg_off[seg] + c) * 9 + tap] * S[((long)n * d.Cout + o) * 9 + tap];
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) { if (threadIdx.x < s2) sh[threadIdx.x] += sh[threadIdx.x + s2]; __syncthreads(); }
    if (threadIdx.x == 0) g[n * g_sn + c] += sh[0];
}

// the centre tap's sum is the plain per-sample channel sum: the conv's bias gradient comes for free (dbias[c] += sum_n S[n][c][1,1])
__global__ void k_bias_from_sums(const float* S, int N, int C, float* dbias) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int n = 0; n < N; n++) s += S[((long)n * C + c) * 9 + 4];
    dbias[c] += s;
}

}  // namespace

int pw_bcast_input_grad(const TV& dz, const PackDesc& d, int seg, float* S, float* g, long g_sn, float* dbias, hipStream_t st) {
    if (d.KS != 3 || dz.C != d.Cout) return -1;
    hipLaunchKernelGGL(k_border_sums, dim3(dz.N, cdiv((dz.C + 3) / 4, 16)), dim3(256), 0, st, dz, S);
    hipLaunchKernelGGL(k_bcast_grad, dim3(dz.N, d.seg_C[seg]), dim3(256), 0, st, d, seg, (const float*)S, g, g_sn);
    if (dbias) hipLaunchKernelGGL(k_bias_from_sums, dim3(cdiv(dz.C, 256)), dim3(256), 0, st, (const float*)S, dz.N, dz.C, dbias);
    return 0;
}

int pw_copy(const TV& s, const TV& d, int acc, hipStream_t st) {
    bool aligned = !(((uintptr_t)s.p | (uintptr_t)d.p) & 15) && !((s.ld | d.ld) & 3) && !((s.sn | d.sn) & 3);
    if (!aligned) return run_map((long)d.N * d.H * d.W, 1, FCopy1{s, d, make_fdiv(d.H * d.W), acc}, st);
    return run_map((long)d.N * d.H * d.W, d.C, FCopy{s, d, make_fdiv(d.H * d.W), acc}, st, quads(s.C, d.C));
}
int pw_fill(const TV& d, float v, hipStream_t st) { return run_map((long)d.N * d.H * d.W, d.C, FFill{d, make_fdiv(d.H * d.W), v}, st, quads(d.C)); }
int pw_pool2(const TV& in, const TV& out, hipStream_t st, int act) { return run_map((long)out.N * out.H * out.W, out.C, FPool2{in, out, act, make_fdiv(out.H * out.W), make_fdiv(out.W)}, st, quads(in.C, out.C)); }
int pw_pool2_bwd(const TV& dout, const TV& din, int assign, hipStream_t st) { return run_map((long)din.N * din.H * din.W, din.C, FPool2Bwd{dout, din, assign, make_fdiv(din.H * din.W), make_fdiv(din.W)}, st, quads(dout.C, din.C)); }
int pw_up2(const TV& in, const TV& out, hipStream_t st) { return run_map((long)out.N * out.H * out.W, out.C, FUp2{in, out, make_fdiv(out.H * out.W), make_fdiv(out.W)}, st, quads(in.C, out.C)); }
int pw_up2_bwd(const TV& dout, const TV& din, hipStream_t st, int assign) { return run_map((long)din.N * din.H * din.W, din.C, FUp2Bwd{dout, din, assign, make_fdiv(din.H * din.W), make_fdiv(din.W)}, st, quads(dout.C, din.C)); }
int pw_stats(const TV& x, double* sums, double* scratch, hipStream_t st) { RedArgs a{}; a.x = x; a.sums = sums; a.partials = scratch; return run_reduce<0>(a, st); }
static BnFin make_fin(long count, const float* gamma, const float* beta, float* rmean, float* rvar, int C, float* mean, float* invstd, float* scale, float* shift) {
    BnFin f; f.count = (double)count; f.gamma = gamma; f.beta = beta; f.rmean = rmean; f.rvar = rvar; f.C = C; f.momentum = 0.1f; f.eps = 1e-5f;
    f.mean = mean; f.invstd = invstd; f.scale = scale; f.shift = shift;
    return f;
}
int pw_bn_finalize(const double* sums, long count, const float* gamma, const float* beta, float* rmean, float* rvar, int C, int training,
                   float* mean, float* invstd, float* scale, float* shift, hipStream_t st) {
    hipLaunchKernelGGL(k_bn_finalize, dim3(cdiv(C, 64)), dim3(64), 0, st, sums, make_fin(count, gamma, beta, rmean, rvar, C, mean, invstd, scale, shift), training);
    return 0;
}
// train-mode statistics + finalisation: k_reduce<0> (per-block partials) -> k_sum_partials with the finalisation fused
int pw_bn_stats_finalize(const TV& x, double* sums, double* scratch, const float* gamma, const float* beta, float* rmean, float* rvar,
                         float* mean, float* invstd, float* scale, float* shift, hipStream_t st) {
    if (!scratch) return -1;
    RedArgs a{}; a.x = x; a.sums = sums; a.partials = scratch;
    BnFin f = make_fin((long)x.N * x.H * x.W, gamma, beta, rmean, rvar, x.C, mean, invstd, scale, shift);
    return run_reduce<0>(a, st, &f);
}
int pw_bn_ema(const float* const* means, const float* const* uvars, int n, int C, float* rmean, float* rvar, hipStream_t st) {
    for (int i0 = 0; i0 < n; i0 += 32) {
        EmaArgs a{}; a.n = n - i0 < 32 ? n - i0 : 32; a.C = C; a.momentum = 0.1f; a.rmean = rmean; a.rvar = rvar;
        for (int i = 0; i < a.n; i++) { a.mean[i] = means[i0 + i]; a.uvar[i] = uvars[i0 + i]; }
        hipLaunchKernelGGL(k_bn_ema, dim3(cdiv(C, 64)), dim3(64), 0, st, a);
    }
    return 0;
}
int pw_bn_finalize_tiles(const float* part, int ntiles, int ldp, long count, const float* gamma, const float* beta, float* rmean, float* rvar, int C,
                         float* mean, float* invstd, float* scale, float* shift, hipStream_t st) {
    hipLaunchKernelGGL(k_bn_finalize_tiles, dim3(cdiv(C, 4)), dim3(256), 0, st, part, ntiles, ldp, make_fin(count, gamma, beta, rmean, rvar, C, mean, invstd, scale, shift));
    return 0;
}
// measured on the MI355X: one workgroup per 4 channels streams at ~13 GB/s (per-CU memory parallelism), so the one-launch path only
// wins below ~1024 pixels (Breakout's 13x10 maps, unit tests); BAIR's smallest map (16x16x8) stays on the multi-workgroup path
bool pw_bn_small_pays(const TV& x) { return (long)x.N * x.H * x.W <= 1024 && x.C >= 4; }
bool pw_bn_small_ok(const TV& x) { return (long)x.N * x.H * x.W <= 256L * BNS_PPT && x.C >= 4; }       // what the kernels can do
// train-mode BatchNorm forward in one launch: statistics, running-stat update, (mean, invstd, scale, shift) and out = act(bn(x) [+ x2])
int pw_bn_small_fwd(const TV& x, const float* gamma, const float* beta, float* rmean, float* rvar, float* mean, float* invstd, float* scale, float* shift,
                    const TV* x2, int act, const TV& out, hipStream_t st) {
    if (!pw_bn_small_ok(x)) return -1;
    BnSmallFwd a{x, x2 ? *x2 : x, out, x2 ? 1 : 0, act, make_fin((long)x.N * x.H * x.W, gamma, beta, rmean, rvar, x.C, mean, invstd, scale, shift), make_fdiv(x.H * x.W)};
    hipLaunchKernelGGL(k_bn_small_fwd, dim3(cdiv(x.C, 4)), dim3(256), 0, st, a);
    return 0;
}
// BatchNorm backward in one launch: dx += ..., dgamma/dbeta +=, and (optional) dres += dout * act'(out) for the residual input
int pw_bn_small_bwd(const TV& dout, const TV* outm, const TV& x, const float* mean, const float* invstd, const float* gamma, const TV& dx,
                    float* dgamma, float* dbeta, const TV* dres, int assign, hipStream_t st, int res_assign) {
    if (!pw_bn_small_ok(x)) return -1;
    BnSmallBwd a{dout, outm ? *outm : dout, x, dx, dres ? *dres : dx, outm ? 1 : 0, dres ? 1 : 0, mean, invstd, gamma, dgamma, dbeta, assign, res_assign, make_fdiv(x.H * x.W)};
    hipLaunchKernelGGL(k_bn_small_bwd, dim3(cdiv(x.C, 4)), dim3(256), 0, st, a);
    return 0;
}
int pw_bn_apply(const TV& x, const float* scale, const float* shift, const TV* x2, const float* scale2, const float* shift2, int act, const TV& out, hipStream_t st) {
    FBnApply f{x, x2 ? *x2 : x, out, scale, shift, scale2, shift2, make_fdiv(x.H * x.W), x2 ? 1 : 0, act};
    return run_map((long)x.N * x.H * x.W, x.C, f, st, quads(x.C, out.C, x2 ? x2->C : 0));
}
int pw_bn_bwd_reduce(const TV& dout, const TV* outm, const TV& x, const float* mean, const float* invstd, double* sums, double* scratch, float* dgamma, float* dbeta, hipStream_t st,
                     const float* lazy_scale, const float* lazy_shift) {
    RedArgs a{}; a.partials = scratch; if (scratch) { a.dgamma = dgamma; a.dbeta = dbeta; } a.x = x; a.dout = dout; a.outm = outm ? *outm : dout; a.act = (outm || lazy_scale) ? 1 : 0; a.mean = mean; a.invstd = invstd; a.sums = sums;
    a.lz_scale = lazy_scale; a.lz_shift = lazy_shift;
    return run_reduce<1>(a, st);
}
int pw_bn_bwd_apply(const TV& dout, const TV* outm, const TV& x, const float* mean, const float* invstd, const float* gamma, const double* sums,
                    const TV& dx, float* dgamma, float* dbeta, int assign, hipStream_t st, const float* lazy_scale, const float* lazy_shift) {
    long M = (long)x.N * x.H * x.W;
    FBnBwdApply f{dout, outm ? *outm : dout, x, dx, mean, invstd, gamma, sums, make_fdiv(x.H * x.W), (outm || lazy_scale) ? 1 : 0, (float)(1.0 / (double)M), assign, lazy_scale, lazy_shift};
    run_map(M, x.C, f, st, quads(dout.C, x.C, dx.C, outm ? outm->C : 0));
    if (dgamma) hipLaunchKernelGGL(k_bn_param_grad, dim3(cdiv(x.C, 64)), dim3(64), 0, st, sums, x.C, dgamma, dbeta);
    return 0;
}
int pw_act_bwd_add(const TV& dout, const TV& outm, const TV& dres, hipStream_t st, int assign) { return run_map((long)dres.N * dres.H * dres.W, dres.C, FActBwdAdd{dout, outm, dres, make_fdiv(dres.H * dres.W), assign}, st, quads(dout.C, outm.C, dres.C)); }
int pw_lstm_fwd(const TV& gates, const TV& cprev, const TV& h, const TV& cn, hipStream_t st, const TV* hb, const float* scale, const float* shift) {
    return run_map((long)h.N * h.H * h.W, h.C, FLstmFwd{gates, cprev, h, cn, make_fdiv(h.H * h.W), hb ? *hb : TV{}, hb ? scale : nullptr, hb ? shift : nullptr}, st, quads(h.C));
}
int pw_lstm_bwd(const TV& gates, const TV& cprev, const TV& cn, const TV& dh, const TV& dc, const TV& dgates, const TV& dcprev, hipStream_t st) {
    return run_map((long)dh.N * dh.H * dh.W, dh.C, FLstmBwd{gates, cprev, cn, dh, dc, dgates, dcprev, make_fdiv(dh.H * dh.W)}, st, quads(dh.C));
}
int pw_tanh_bwd(const TV& dy, const TV& y, const TV& dz, hipStream_t st) { return run_map((long)y.N * y.H * y.W, y.C, FTanhBwd{dy, y, dz, make_fdiv(y.H * y.W)}, st, quads(dy.C, y.C, dz.C)); }
int pw_attn_mul(const TV& x, const TV& out, const TV& att, hipStream_t st) { return run_map((long)x.N * x.H * x.W, out.C, FAttnMul{x, out, att, make_fdiv(x.H * x.W)}, st, quads(out.C)); }
int pw_attn_mul_bwd(const TV& x, const TV& dout, const TV& datt, const TV& dx, hipStream_t st) {
    long npix = (long)x.N * x.H * x.W;
    if (npix <= 0) return 0;
    if ((x.ld & 3) || (dout.ld & 3) || (dx.ld & 3) || (x.sn & 3) || (dout.sn & 3) || (dx.sn & 3)) return -1;
    hipLaunchKernelGGL(k_attn_mul_bwd, dim3((unsigned)cdiv(npix * 16, 256)), dim3(256), 0, st, AttnBwdArgs{x, dout, datt, dx, make_fdiv(x.H * x.W), npix});
    return 0;
}
int pw_gap(const TV& x, float* out, hipStream_t st) {
    hipMemsetAsync(out, 0, sizeof(float) * (size_t)x.N * x.C, st);
    RedArgs a{}; a.x = x; a.outf = out; a.out_sn = x.C; a.scale = 1.f / (float)(x.H * x.W);
    return run_reduce<2>(a, st);
}
int pw_gap_bwd(const float* dout, const TV& dx, hipStream_t st) { return run_map((long)dx.N * dx.H * dx.W, dx.C, FGapBwd{dx, dout, make_fdiv(dx.H * dx.W), 1.f / (float)(dx.H * dx.W)}, st, quads(dx.C)); }
int pw_colsum(const TV& x, float* out, hipStream_t st, bool det, double* scratch) {      // scratch (RED_MAX_BLOCKS x C doubles, private to the stream): the bit-reproducible form's partial sums
    RedArgs a{}; a.x = x; a.outf = out; a.det = det ? 1 : 0; a.partials = det ? scratch : nullptr; return run_reduce<3>(a, st);
}
int pw_spatial_sum(const TV& x, float* out, long out_sn, hipStream_t st, bool det) { RedArgs a{}; a.x = x; a.outf = out; a.out_sn = out_sn; a.scale = 1.f; a.det = det ? 1 : 0; return run_reduce<2>(a, st); }
int pw_nchw_to_nhwc(const float* src, long src_sn, const TV& d, hipStream_t st) { return run_map((long)d.N * d.H * d.W, 1, FNchwToNhwc{src, src_sn, d, make_fdiv(d.H * d.W)}, st); }
int pw_nhwc_to_nchw(const TV& s, float* dst, long dst_sn, int acc, hipStream_t st) { return run_map((long)s.N * s.H * s.W, 1, FNhwcToNchw{s, dst, dst_sn, make_fdiv(s.H * s.W), acc}, st); }
// bias of a conv with the following eval-mode BatchNorm folded in: out[o] = bias[o] * scale[o] + shift[o]
__global__ void k_fold_bias(const float* bias, const float* scale, const float* shift, float* out, int C) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o < C) out[o] = (bias ? bias[o] * scale[o] : 0.f) + shift[o];
}
int pw_fold_bias(const float* bias, const float* scale, const float* shift, float* out, int C, hipStream_t st) {
    hipLaunchKernelGGL(k_fold_bias, dim3(cdiv(C, 256)), dim3(256), 0, st, bias, scale, shift, out, C);
    return 0;
}
__global__ void k_vec_add(VecAddJobs j) {
    const int i = blockIdx.x;
    for (int k = threadIdx.x; k < j.n[i]; k += blockDim.x) j.dst[i][k] += j.src[i][k];
}
int pw_vec_add(const VecAddJobs& j, hipStream_t st) {
    if (j.count <= 0) return 0;
    hipLaunchKernelGGL(k_vec_add, dim3(j.count), dim3(256), 0, st, j);
    return 0;
}
int pw_batch_sum(const float* src, long sn, long n_el, int N, float* dst, hipStream_t st) {
    hipLaunchKernelGGL(k_batch_sum, dim3(cdiv(n_el, 256)), dim3(256), 0, st, src, sn, n_el, N, dst);
    return 0;
}
