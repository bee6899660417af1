// Thin-channel convolutions on the vector ALUs (fp32 FMA rate == fp32 MFMA rate on gfx950, so padding 3 OUTPUT channels to a
// 16/32-wide MFMA tile would waste 5-10x):
//   k_conv_thin_out : OUT <= 12 channels (FinalBlock 3-channel heads, final_block.py:9-29; dgrad of E's stem)
//   k_conv_thin_in  : IN  <= 12 channels -- only the observation_stacking > 1 case (12-channel stem); the 3-channel (pitch 4) input
//                     side runs on the 16x16x4 MFMA kernels of conv_narrow.hip (k_conv_c4 / k_wgrad_c4), tried first by the launchers
//   k_wgrad_thin    : weight gradients with one thin operand, same remark
// Same math, weight layout ([tap][OUT_pad][K]) and ConvArgs/WgradArgs contract as conv_mfma.hip, so the launchers there
// dispatch here purely on shape.  Output tile = 8 x 32 pixels per workgroup; the input halo tile is staged through LDS
// with a 20-float pixel pitch (conflict-free ds_read_b128); weights are wave-uniform: read with scalar loads from a compact
// per-launch table (ConvArgs.aux) -- the packed layout's 2-4 KB tap stride maps every row to the same scalar-cache set.
// These layers move ~1 byte per 10-50 FLOP: bound = HBM for the 3-channel side, VALU for the wide side.
#include "common.h"

namespace {
constexpr int TW = 32, TH = 8, CK = 16, PITCH = CK + 4;

__device__ __forceinline__ float dot4(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }

__device__ __forceinline__ float4 ldmask(const float* q, int c, int C) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c + 4 <= C) v = *reinterpret_cast<const float4*>(q);
    else { if (c < C) v.x = q[0]; if (c + 1 < C) v.y = q[1]; if (c + 2 < C) v.z = q[2]; }
    return v;
}

// stage channels [c0, c0+4*Q4) of the (TH+2R)x(TW+2R) halo tile of `src` (zero padded) into lds[pix][4*Q4+4]
template <int Q4 = 4>
__device__ __forceinline__ void stage_halo(float* lds, const float* base, int ld, int C, int c0, int H, int W, int y0, int x0, int R, int tid) {
    constexpr int PITCH = 4 * Q4 + 4;
    const int HWD = TW + 2 * R, HHT = TH + 2 * R;
    for (int idx = tid; idx < HHT * HWD * Q4; idx += 256) {
        int q = idx % Q4, pix = idx / Q4;
        int hy = pix / HWD, hx = pix - hy * HWD;
        int y = y0 - R + hy, x = x0 - R + hx;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        int c = c0 + q * 4;
        if (y >= 0 && y < H && x >= 0 && x < W && c < C) v = ldmask(base + ((long)y * W + x) * ld + c, c, C);
        *reinterpret_cast<float4*>(&lds[pix * PITCH + q * 4]) = v;
    }
}

// Weights are wave-uniform, but reading them with scalar loads thrashes the 16 KB scalar cache (49 taps x CO rows, 4 KB apart in
// the packed layout): they are staged per 16-channel chunk in LDS as [tap][quad][o] float4 and read back as broadcasts.
// compact weight tables (built per launch into ConvArgs.aux by k_thin_compact_*): contiguous, so the wave-uniform s_load stream of a
// workgroup stays inside the scalar cache -- the packed layout's 2-4 KB tap stride maps every row to the same cache set.
//   thin_out: T[chunk][tap][quad][o < CO]  float4        thin_in: T[group][tap][quad < IC4][o < 16]  float4
__global__ void k_thin_compact_out(ConvArgs a, int CO, int chunks) {
    int taps = a.KS * a.KS;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= chunks * taps * 4 * CO) return;
    int o = idx % CO, tq = idx / CO;
    int q = tq & 3, ct = tq >> 2;
    int tap = ct % taps, chunk = ct / taps;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o < a.Cout) w = *reinterpret_cast<const float4*>(a.wp + ((long)tap * a.Cout_pad + o) * a.Ktot + chunk * CK + q * 4);
    *reinterpret_cast<float4*>(a.aux + (long)idx * 4) = w;
}
__global__ void k_thin_compact_in(ConvArgs a, int IC4, int groups) {
    int taps = a.KS * a.KS;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= groups * taps * IC4 * 16) return;
    int o = idx & 15, tq = idx >> 4;
    int q = tq % IC4, gt = tq / IC4;
    int tap = gt % taps, og = gt / taps;
    *reinterpret_cast<float4*>(a.aux + (long)idx * 4) = *reinterpret_cast<const float4*>(a.wp + ((long)tap * a.Cout_pad + og * 16 + o) * a.Ktot + q * 4);
}

template <int CO, bool WS>
__global__ __launch_bounds__(256) void k_conv_thin_out(ConvArgs a) {
    __shared__ float lds[(TH + 6) * (TW + 6) * PITCH];
    __shared__ float wl[WS ? 4 : 49 * 4 * CO * 4];
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int n = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    const int R = a.KS >> 1, HWD = TW + 2 * R, taps = a.KS * a.KS;
    const ConvSrc s = a.src[0];
    const float* base = s.p + (long)n * s.sn;
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; o++) acc[o] = 0.f;
    for (int c0 = 0; c0 < s.C; c0 += CK) {
        stage_halo<4>(lds, base, s.ld, s.C, c0, a.H, a.W, y0, x0, R, tid);
        if (!WS) for (int idx = tid; idx < taps * 4 * CO; idx += 256) {
            int o = idx % CO, tq = idx / CO;
            int q = tq & 3, tap = tq >> 2;
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (o < a.Cout) w = *reinterpret_cast<const float4*>(a.wp + ((long)tap * a.Cout_pad + o) * a.Ktot + c0 + q * 4);
            *reinterpret_cast<float4*>(&wl[idx * 4]) = w;
        }
        __syncthreads();
        for (int tap = 0; tap < taps; tap++) {
            int dy = tap / a.KS, dx = tap - dy * a.KS;
            const float* xp = &lds[((ty + dy) * HWD + tx + dx) * PITCH];
            const float* __restrict__ wt = WS ? a.aux + ((long)(c0 / CK) * taps + tap) * (4 * CO * 4) : &wl[tap * 4 * CO * 4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float4 x4 = *reinterpret_cast<const float4*>(xp + q * 4);
#pragma unroll
                for (int o = 0; o < CO; o++) acc[o] += dot4(x4, *reinterpret_cast<const float4*>(wt + (q * CO + o) * 4));
            }
        }
        __syncthreads();
    }
    int y = y0 + ty, x = x0 + tx;
    if (y < a.H && x < a.W) {
        float* o_ = a.out + (long)n * a.out_sn + ((long)y * a.W + x) * a.out_ld;
#pragma unroll
        for (int o = 0; o < CO; o++) {
            if (o >= a.Cout) break;
            float v = acc[o] + (a.bias ? a.bias[o] : 0.f);
            if (a.act == 1) v = tanhf(v);
            if (a.accumulate) v += o_[o];
            o_[o] = v;
        }
    }
}

// Narrow input (IN <= 4*Q4 channels, K = Ktot <= 32): each thread computes one pixel x 16 output channels on the VALUs with no
// channel padding; blockIdx.z = n * groups + output group.  Q4 = 4 (pitch 20) also serves 7x7 heads' dgrad; Q4 = 8 (pitch 36) is 3x3/1x1 only.
template <int Q4, int MAXR, int WQ, bool WS>
__global__ __launch_bounds__(256) void k_conv_thin_in(ConvArgs a, int groups) {
    constexpr int PITCH = 4 * Q4 + 4;
    constexpr int MAXT = (2 * MAXR + 1) * (2 * MAXR + 1);
    __shared__ float lds[(TH + 2 * MAXR) * (TW + 2 * MAXR) * PITCH];
    __shared__ float wl[WS ? 4 : MAXT * WQ * 16 * 4];       // this output group's weights: [tap][quad][o] float4 (LDS broadcasts, see k_conv_thin_out)
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int n = blockIdx.z / groups, og = blockIdx.z - n * groups;
    const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    const int R = a.KS >> 1, HWD = TW + 2 * R, taps = a.KS * a.KS;
    const ConvSrc s = a.src[0];
    const int IC4 = (s.C + 3) >> 2;                // <= WQ
    stage_halo<Q4>(lds, s.p + (long)n * s.sn, s.ld, s.C, 0, a.H, a.W, y0, x0, R, tid);
    if (!WS) for (int idx = tid; idx < taps * IC4 * 16; idx += 256) {
        int o = idx & 15, tq = idx >> 4;
        int q = tq % IC4, tap = tq / IC4;
        *reinterpret_cast<float4*>(&wl[idx * 4]) = *reinterpret_cast<const float4*>(a.wp + ((long)tap * a.Cout_pad + og * 16 + o) * a.Ktot + q * 4);
    }
    __syncthreads();
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; o++) acc[o] = 0.f;
    for (int tap = 0; tap < taps; tap++) {
        int dy = tap / a.KS, dx = tap - dy * a.KS;
        const float* xp = &lds[((ty + dy) * HWD + tx + dx) * PITCH];
        const float* __restrict__ wt = WS ? a.aux + ((long)og * taps + tap) * (IC4 * 64) : &wl[tap * IC4 * 64];
        for (int q = 0; q < IC4; q++) {
            float4 x4 = *reinterpret_cast<const float4*>(xp + q * 4);
#pragma unroll
            for (int o = 0; o < 16; o++) acc[o] += dot4(x4, *reinterpret_cast<const float4*>(wt + (q * 16 + o) * 4));
        }
    }
    int y = y0 + ty, x = x0 + tx;
    if (y < a.H && x < a.W) {
        float* o_ = a.out + (long)n * a.out_sn + ((long)y * a.W + x) * a.out_ld + og * 16;
#pragma unroll
        for (int o4 = 0; o4 < 4; o4++) {
            int c = og * 16 + o4 * 4;
            if (c >= a.Cout) break;
            float4 v = make_float4(acc[o4 * 4], acc[o4 * 4 + 1], acc[o4 * 4 + 2], acc[o4 * 4 + 3]);
            if (a.bias) { v.x += a.bias[c]; if (c + 1 < a.Cout) v.y += a.bias[c + 1]; if (c + 2 < a.Cout) v.z += a.bias[c + 2]; if (c + 3 < a.Cout) v.w += a.bias[c + 3]; }
            float* p = o_ + o4 * 4;
            if (c + 4 <= a.Cout) {
                if (a.accumulate) { float4 e = *reinterpret_cast<float4*>(p); v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w; }
                *reinterpret_cast<float4*>(p) = v;
            } else {
                float vv[4] = {v.x, v.y, v.z, v.w};
                for (int e = 0; e < 4 && c + e < a.Cout; e++) p[e] = a.accumulate ? p[e] + vv[e] : vv[e];
            }
        }
    }
}

// G[tap][t][w] = sum_p thin[p][t] * wide[p + off(tap)][w], thin has TC <= 12 channels, wide is processed in 16-channel chunks.
//   swap = 0 (Cout thin): thin = dY, wide = X        -> dwp[(tap*Cout_pad + t)*Ktot + w]        += G[tap][t][w]
//   swap = 1 (Cin thin) : thin = X,  wide = dY       -> dwp[(flip(tap)*Cout_pad + w)*Ktot + t]  += G[tap][t][w]
// Workgroups stride over pixel tiles keeping their partial G in registers, then flush once with atomics.
struct ThinWgradArgs {
    const float* thin; long thin_sn; int thin_ld; int TC;
    const float* wide; long wide_sn; int wide_ld; int WC;
    int N, H, W, KS, swap, Cout_pad, Ktot;
    float* dwp;
    int tiles_x, tiles_y, chunks;
    float* det_slab; long det_stride;      // deterministic mode: every pixel split (blockIdx.x) flushes into its own copy (WgradArgs.det_slab)
};
template <int TCP>
__global__ __launch_bounds__(256) void k_wgrad_thin(ThinWgradArgs a) {
    __shared__ float wl[(TH + 6) * (TW + 6) * PITCH];
    __shared__ float tl[TH * TW * TCP];
    const int tid = threadIdx.x;
    const int taps = a.KS * a.KS, R = a.KS >> 1, HWD = TW + 2 * R;
    const int IT = taps * 4;                     // work items per chunk: (tap, quad of wide channels)
    const int PSn = IT >= 256 ? 1 : 256 / IT;    // pixel splits
    const int ps = tid / IT, item = tid - ps * IT;
    const bool active = ps < PSn;
    const int tap = item >> 2, wq = item & 3;
    const int dy = tap / a.KS, dx = tap - dy * a.KS;
    const int chunk = blockIdx.y;
    const long ntiles = (long)a.N * a.tiles_y * a.tiles_x;
    float acc[TCP][4];
#pragma unroll
    for (int t = 0; t < TCP; t++) { acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f; }
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int n = (int)(tile / (a.tiles_y * a.tiles_x));
        int rem = (int)(tile - (long)n * a.tiles_y * a.tiles_x);
        int y0 = (rem / a.tiles_x) * TH, x0 = (rem % a.tiles_x) * TW;
        stage_halo<4>(wl, a.wide + (long)n * a.wide_sn, a.wide_ld, a.WC, chunk * CK, a.H, a.W, y0, x0, R, tid);
        for (int idx = tid; idx < TH * TW; idx += 256) {
            int yy = y0 + idx / TW, xx = x0 + idx % TW;
            const float* tp = a.thin + (long)n * a.thin_sn + ((long)yy * a.W + xx) * a.thin_ld;
            bool ok = yy < a.H && xx < a.W;
#pragma unroll
            for (int t = 0; t < TCP; t++) tl[idx * TCP + t] = (ok && t < a.TC) ? tp[t] : 0.f;
        }
        __syncthreads();
        if (active) {
            for (int p = ps; p < TH * TW; p += PSn) {
                int py = p / TW, px = p - py * TW;
                float4 w4 = *reinterpret_cast<const float4*>(&wl[((py + dy) * HWD + px + dx) * PITCH + wq * 4]);
#pragma unroll
                for (int t = 0; t < TCP; t++) {
                    float tv = tl[p * TCP + t];
                    acc[t][0] = fmaf(tv, w4.x, acc[t][0]); acc[t][1] = fmaf(tv, w4.y, acc[t][1]);
                    acc[t][2] = fmaf(tv, w4.z, acc[t][2]); acc[t][3] = fmaf(tv, w4.w, acc[t][3]);
                }
            }
        }
        __syncthreads();
    }
    // fold the pixel-splits of this workgroup in LDS (ds_add_f32), then ONE global atomic per (tap, t, w) and workgroup
    float* red = wl;                                  // reuse the halo tile storage: IT * TCP * 4 floats
    for (int i = tid; i < IT * TCP * 4; i += 256) red[i] = 0.f;
    __syncthreads();
    for (int s = 0; s < PSn; s++) {      // the pixel splits take turns, in order: fixed summation order (ds_add_f32 from all of them at once summed in arrival order)
        if (active && ps == s) {
#pragma unroll
            for (int t = 0; t < TCP; t++)
#pragma unroll
                for (int e = 0; e < 4; e++) red[(item * TCP + t) * 4 + e] += acc[t][e];
        }
        __syncthreads();
    }
    float* const dwp = a.det_slab ? a.det_slab + (long)blockIdx.x * a.det_stride : a.dwp;
    for (int i = tid; i < IT * TCP * 4; i += 256) {
        int e = i & 3, t = (i >> 2) % TCP, it = (i >> 2) / TCP;
        int tp = it >> 2, w = chunk * CK + (it & 3) * 4 + e;
        if (t >= a.TC || w >= a.WC) continue;
        float* d = a.swap ? dwp + ((long)(taps - 1 - tp) * a.Cout_pad + w) * a.Ktot + t
                          : dwp + ((long)tp * a.Cout_pad + t) * a.Ktot + w;
        atomicAdd(d, red[i]);
    }
}
}  // namespace

// returns 1 if handled, 0 if the shape is not thin, <0 on error
int conv_thin_fwd_try(const ConvArgs& a, hipStream_t st) {
    if (a.nsrc != 1 || a.src[0].bcast || a.splitk > 1) return 0;
    dim3 grid(cdiv(a.W, TW), cdiv(a.H, TH), a.N);
    // OUT<=4: FinalBlock heads; OUT<=12 only with a narrow input (dgrad of the stem).  The 9-channel broadcast-input dgrad of R
    // (IN up to 1024 channels on 16x16 maps) stays on the MFMA kernel: one thread per pixel would leave the chip empty.
    // tiny grids (batch-1 roll-out: 16-64 workgroups, each walking all channel chunks serially) are latency-bound here: the MFMA kernel with
    // its deterministic split-K (bias / tanh applied by the reduce) fills the chip instead
    if (a.Cout <= 4 && a.KS == 3 && !a.accumulate && a.split_scratch && (long)grid.x * grid.y * grid.z <= 64 && a.src[0].C >= 32) return 0;
    if ((a.Cout <= 4 && a.src[0].C >= 16) || (a.Cout <= 12 && a.src[0].C >= 16 && a.src[0].C <= 32)) {
        const int CO = a.Cout <= 3 ? 3 : (a.Cout <= 4 ? 4 : 12), chunks = cdiv(a.src[0].C, CK);
        const int n4 = chunks * a.KS * a.KS * 4 * CO;
        if (a.aux && (long)n4 * 16 <= CONV_AUX_BYTES) {
            hipLaunchKernelGGL(k_thin_compact_out, dim3(cdiv(n4, 256)), dim3(256), 0, st, a, CO, chunks);
            if (CO == 3) hipLaunchKernelGGL((k_conv_thin_out<3, true>), grid, dim3(256), 0, st, a);
            else if (CO == 4) hipLaunchKernelGGL((k_conv_thin_out<4, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_conv_thin_out<12, true>), grid, dim3(256), 0, st, a);
        } else if (CO == 3) hipLaunchKernelGGL((k_conv_thin_out<3, false>), grid, dim3(256), 0, st, a);
        else if (CO == 4) hipLaunchKernelGGL((k_conv_thin_out<4, false>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_conv_thin_out<12, false>), grid, dim3(256), 0, st, a);
        g_last_conv_kernel = CK_THIN_OUT;
        return 1;
    }
    // IN <= 12 (stem, FinalBlock dgrad).  Wider inputs stay on the MFMA kernel: scalar v_fma_f32 peaks at half the packed/MFMA
    // fp32 rate, so the VALU path only wins when it avoids > 2x channel padding (measured: 32->64 @64x64: 482 us here vs 221 us on MFMA).
    if (a.src[0].C <= 12 && a.Ktot == 16 && a.act == 0) {
        if (conv_narrow_fwd_ok(a)) return 0;      // 5 .. 12 channels -> >= 16: the 16x16x4 matrix-instruction kernel of conv_narrow.hip (round 5: 22 + 5 -> 12 us for the stacked-frame stem at 256 x 256)
        int groups = cdiv(a.Cout, 16);
        if (groups * 16 > a.Cout_pad || (long)a.N * groups > 65535) return 0;
        grid.z = a.N * groups;
        const int IC4 = (a.src[0].C + 3) >> 2, n4 = groups * a.KS * a.KS * IC4 * 16;
        if (a.aux && (long)n4 * 16 <= CONV_AUX_BYTES) {
            hipLaunchKernelGGL(k_thin_compact_in, dim3(cdiv(n4, 256)), dim3(256), 0, st, a, IC4, groups);
            hipLaunchKernelGGL((k_conv_thin_in<4, 3, 3, true>), grid, dim3(256), 0, st, a, groups);
        } else if (a.src[0].C <= 4) hipLaunchKernelGGL((k_conv_thin_in<4, 3, 1, false>), grid, dim3(256), 0, st, a, groups);
        else hipLaunchKernelGGL((k_conv_thin_in<4, 3, 3, false>), grid, dim3(256), 0, st, a, groups);
        g_last_conv_kernel = CK_THIN_IN;
        return 1;
    }
    return 0;
}

int conv_thin_wgrad_try(const WgradArgs& w, hipStream_t st, bool dry) {
    if (w.nsrc != 1 || w.src[0].bcast) return 0;
    ThinWgradArgs a{};
    a.N = w.N; a.H = w.H; a.W = w.W; a.KS = w.KS; a.Cout_pad = w.Cout_pad; a.Ktot = w.Ktot; a.dwp = w.dwp;
    a.tiles_x = cdiv(w.W, TW); a.tiles_y = cdiv(w.H, TH);
    if ((w.Cout <= 4 && w.src[0].C >= 16) || (w.Cout <= 12 && w.src[0].C >= 16 && w.src[0].C <= 32)) {          // thin = dY, wide = X
        a.swap = 0; a.thin = w.dy; a.thin_sn = w.dy_sn; a.thin_ld = w.dy_ld; a.TC = w.Cout;
        a.wide = w.src[0].p; a.wide_sn = w.src[0].sn; a.wide_ld = w.src[0].ld; a.WC = w.src[0].C;
    } else if (w.src[0].C <= 12) {                    // thin = X, wide = dY
        a.swap = 1; a.thin = w.src[0].p; a.thin_sn = w.src[0].sn; a.thin_ld = w.src[0].ld; a.TC = w.src[0].C;
        a.wide = w.dy; a.wide_sn = w.dy_sn; a.wide_ld = w.dy_ld; a.WC = w.Cout;
    } else return 0;
    a.chunks = cdiv(a.WC, CK);
    long ntiles = (long)a.N * a.tiles_x * a.tiles_y;
    long gx = 512 / a.chunks; if (gx < 32) gx = 32; if (gx > ntiles) gx = ntiles;
    g_last_conv_kernel = CK_WGRAD_THIN;
    if (dry) return 1;
    WgradArgs b = w;
    if (b.det_slab) { gx = wgrad_det_begin(b, gx, st); if (gx <= 0) return -1; a.det_slab = b.det_slab; a.det_stride = b.det_stride; }
    dim3 grid((unsigned)gx, a.chunks);
    if (a.TC <= 4) hipLaunchKernelGGL((k_wgrad_thin<4>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_wgrad_thin<12>), grid, dim3(256), 0, st, a);
    if (b.det_slab) wgrad_det_end(b, gx, st);
    g_last_conv_kernel = CK_WGRAD_THIN;
    return 1;
}
