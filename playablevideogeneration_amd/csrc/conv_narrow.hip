// Narrow 3x3 convolutions (16 / 32 channels in AND out: E's first residual blocks on all B*T frames, their dgrads, the last
// decoder block) on v_mfma_f32_16x16x4_f32.
//
// The generic implicit-GEMM kernel (conv_mfma.hip) is a poor fit here: its 32-wide N tile is half padding for 16 output
// channels, and with only 16-32 input channels per tap it re-reads the input nine times through L2 (measured 302 us for a
// 16->16 layer over 128 frames of 128x128 whose HBM-bound time is ~70 us).  Here a workgroup stages the 10x34 halo of an 8x32
// pixel tile in LDS ONCE and all nine taps read from it; M = output channels (16 per MFMA, no padding), N = 16 pixels,
// K = 4 channels per MFMA.  One ds_read_b128 per lane (pixel = lane & 15, channels 4g..4g+3, g = lane >> 4) feeds four MFMAs
// (the j-th uses component j, i.e. the k-set {j, 4+j, 8+j, 12+j} on both operands); the weights of all taps live in registers
// for the lifetime of the (persistent) workgroup.  The D fragment (4 consecutive output channels of one pixel per lane) is a
// float4 NHWC store.  Exact fp32, same packed weight layout / ConvArgs contract as conv_mfma.hip; forward and dgrad.
// Roofline: HBM for 16->16 (AI ~ 18 FLOP/B), fp32 matrix rate for 32->32.
#include "common.h"
#include <cstdlib>

namespace {
constexpr int NTW = 32, NTH = 8, NHW = NTW + 2, NHH = NTH + 2;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 nld4(const float* q, int c, int C) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c + 4 <= C) v = *reinterpret_cast<const float4*>(q);
    else { if (c < C) v.x = q[0]; if (c + 1 < C) v.y = q[1]; if (c + 2 < C) v.z = q[2]; }
    return v;
}

// CI4 = input channels / 16 (1 or 2), NT = output channels / 16 (1 or 2)
template <int CI4, int NT>
__global__ __launch_bounds__(256) void k_conv_narrow(ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int CI = 16 * CI4, PITCH = CI + 4, Q = CI / 4;        // Q float4 per pixel
    constexpr int NLOAD = (NHH * NHW * Q + 255) / 256;               // halo float4 per thread (6 or 11)
    __shared__ float xs[NHH * NHW * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lp = lane & 15, g = lane >> 4;
    const ConvSrc s = a.src[0];
    const long ntiles = (long)a.N * tiles_x * tiles_y;

    // weights: W[tap][mt*16 + lp][ch*16 + 4g .. +3]
    float4 w[9][NT][CI4];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int m = 0; m < NT; m++)
#pragma unroll
            for (int ch = 0; ch < CI4; ch++)
                w[t][m][ch] = *reinterpret_cast<const float4*>(a.wp + ((long)t * a.Cout_pad + m * 16 + lp) * a.Ktot + ch * 16 + 4 * g);

    // halo loader: float4 index idx = tid + 256 i -> (pixel idx / Q, quad idx % Q); Q is a power of two
    float4 pre[NLOAD];
    auto gload = [&](long tile) {
        int n = (int)(tile / (tiles_x * tiles_y));
        int rem = (int)(tile - (long)n * tiles_x * tiles_y);
        int ty = rem / tiles_x;
        int y0 = ty * NTH, x0 = (rem - ty * tiles_x) * NTW;
        const float* base = s.p + (long)n * s.sn;
#pragma unroll
        for (int i = 0; i < NLOAD; i++) {
            int idx = tid + 256 * i, hp = idx / Q, c = (idx % Q) * 4;
            int hy = hp / NHW, hx = hp - hy * NHW;
            int y = y0 - 1 + hy, x = x0 - 1 + hx;
            pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (hp < NHH * NHW && y >= 0 && y < a.H && x >= 0 && x < a.W && c < s.C) pre[i] = nld4(base + ((long)y * a.W + x) * s.ld + c, c, s.C);
        }
    };

    long tile = blockIdx.x;
    if (tile < ntiles) gload(tile);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int i = 0; i < NLOAD; i++) {
            int idx = tid + 256 * i, hp = idx / Q;
            if (hp < NHH * NHW) *reinterpret_cast<float4*>(&xs[hp * PITCH + (idx % Q) * 4]) = pre[i];
        }
        __syncthreads();
        if (tile + gridDim.x < ntiles) gload(tile + gridDim.x);

        f32x4 acc[4][NT];
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int m = 0; m < NT; m++) acc[nt][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int dy = t / 3, dx = t - 3 * dy;
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {                       // n-tile = (row 2*wave + (nt >> 1), columns 16*(nt & 1) .. +15)
                const float* xp = &xs[((2 * wave + (nt >> 1) + dy) * NHW + 16 * (nt & 1) + lp + dx) * PITCH + 4 * g];
#pragma unroll
                for (int ch = 0; ch < CI4; ch++) {
                    const float4 xv = *reinterpret_cast<const float4*>(xp + 16 * ch);
#pragma unroll
                    for (int m = 0; m < NT; m++) {
                        acc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][m][ch].x, xv.x, acc[nt][m], 0, 0, 0);
                        acc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][m][ch].y, xv.y, acc[nt][m], 0, 0, 0);
                        acc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][m][ch].z, xv.z, acc[nt][m], 0, 0, 0);
                        acc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][m][ch].w, xv.w, acc[nt][m], 0, 0, 0);
                    }
                }
            }
        }
        // epilogue: lane holds output channels m*16 + 4g .. +3 of pixel (row, col)
        {
            int n = (int)(tile / (tiles_x * tiles_y));
            int rem = (int)(tile - (long)n * tiles_x * tiles_y);
            int ty = rem / tiles_x;
            int y0 = ty * NTH, x0 = (rem - ty * tiles_x) * NTW;
            // (bias, residual, LeakyReLU, accumulate) of channels c .. c+3 at pixel (y, x) of an OW-wide output map
            auto emit = [&](int y, int x, int OW, int c, float (&v)[4]) {
                float* o = a.out + (long)n * a.out_sn + ((long)y * OW + x) * a.out_ld;
                if (a.bias) for (int e = 0; e < 4; e++) if (c + e < a.Cout) v[e] += a.bias[c + e];
                if (a.res) {
                    const float* rp = a.res + (long)n * a.res_sn + ((long)y * OW + x) * a.res_ld + c;
                    for (int e = 0; e < 4; e++) if (c + e < a.Cout) v[e] += rp[e];
                }
                if (a.act == 3) for (int e = 0; e < 4; e++) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
                if (c + 4 <= a.Cout) {
                    float4 r = make_float4(v[0], v[1], v[2], v[3]);
                    if (a.accumulate) { float4 p = *reinterpret_cast<const float4*>(o + c); r.x += p.x; r.y += p.y; r.z += p.z; r.w += p.w; }
                    *reinterpret_cast<float4*>(o + c) = r;
                } else {
                    for (int e = 0; e < 4 && c + e < a.Cout; e++) o[c + e] = a.accumulate ? o[c + e] + v[e] : v[e];
                }
            };
            if (a.avgpool) {      // 2x2 average: the wave's two rows are accumulators nt and nt + 2 of one lane, the column neighbour is lane ^ 1 (every lane takes part in the exchange)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int y = y0 + 2 * wave, x = x0 + 16 * h + lp;
#pragma unroll
                    for (int m = 0; m < NT; m++) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float r = acc[h][m][e] + acc[h + 2][m][e];
                            const float nb = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0xB1, 0xF, 0xF, true));      // quad_perm [1, 0, 3, 2]
                            v[e] = 0.25f * (r + nb);
                        }
                        const int c = m * 16 + 4 * g;
                        if (!(lp & 1) && y < a.H && x < a.W && c < a.Cout) emit(y >> 1, x >> 1, a.W >> 1, c, v);
                    }
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    int y = y0 + 2 * wave + (nt >> 1), x = x0 + 16 * (nt & 1) + lp;
                    if (y >= a.H || x >= a.W) continue;
#pragma unroll
                    for (int m = 0; m < NT; m++) {
                        int c = m * 16 + 4 * g;
                        if (c >= a.Cout) continue;
                        float v[4] = {acc[nt][m][0], acc[nt][m][1], acc[nt][m][2], acc[nt][m][3]};
                        emit(y, x, a.W, c, v);
                    }
                }
            }
        }
        __syncthreads();
    }
}
}  // namespace

// returns 1 if handled, 0 if the shape does not qualify
// shape test of k_conv_narrow (the output pointer / pitches need not be set: conv_avgpool_ok asks before the output exists).  Round 5: 5 .. 12 input channels too (the
// observation_stacking > 1 stem: 12 -> 16; its K = 16 weight rows are zero beyond C and the loader zero-fills) -- 12 us instead of the vector-ALU kernel's 22 + 5 us at 256 x 256
static bool narrow_shape_ok(const ConvArgs& a) {
    if (a.nsrc != 1 || a.src[0].bcast || a.src[0].bn_scale || a.KS != 3 || (a.act != 0 && a.act != 3) || a.splitk > 1) return false;
    if (a.Ktot > 32 || a.Cout > 32 || a.src[0].C <= 4 || a.Cout <= 4) return false;
    if (a.src[0].C <= 12 && a.Cout < 16) return false;      // (thin input AND thin output: the stem's dgrad shapes stay on conv_thin.hip)
    if ((a.src[0].ld & 3) || (a.src[0].sn & 3) || a.Cout_pad < 16 * ((a.Cout + 15) / 16)) return false;
    if (a.mask || a.pool_out || a.skip_out || a.in_s16 || a.out_s16) return false;
    if ((long)a.N * a.H * a.W < 4096) return false;                    // tiny maps: the generic kernel's split-K paths do better
    if (a.avgpool && ((a.H | a.W) & 1)) return false;
    return true;
}
int conv_narrow_fwd_ok(const ConvArgs& a) { return narrow_shape_ok(a) ? 1 : 0; }
int conv_avgpool_ok(const ConvArgs& a) {
    if (a.wq && a.precision >= PREC_F16X3) return conv_hx_avgpool_ok(a);      // a split-operand layer: conv_fwd_launch tries k_conv_hx's launcher first (conv_direct.hip has the pooled epilogue)
    ConvArgs b = a; b.avgpool = 1; b.splitk = 1;
    if (a.KS == 1) return conv1x1_lat_try(b, nullptr, true);
    return narrow_shape_ok(b) ? 1 : 0;
}

int conv_narrow_fwd_try(const ConvArgs& a, hipStream_t st) {
    if (!narrow_shape_ok(a) || (a.out_ld & 3) || (a.out_sn & 3)) return 0;
    const int tx = cdiv(a.W, NTW), ty = cdiv(a.H, NTH);
    const long ntiles = (long)a.N * tx * ty;
    const int grid = (int)(ntiles < 1024 ? ntiles : 1024);
    const int ci4 = a.Ktot / 16, nt = (a.Cout + 15) / 16;
    if (ci4 == 1 && nt == 1) hipLaunchKernelGGL((k_conv_narrow<1, 1>), dim3(grid), dim3(256), 0, st, a, tx, ty);
    else if (ci4 == 1) hipLaunchKernelGGL((k_conv_narrow<1, 2>), dim3(grid), dim3(256), 0, st, a, tx, ty);
    else if (nt == 1) hipLaunchKernelGGL((k_conv_narrow<2, 1>), dim3(grid), dim3(256), 0, st, a, tx, ty);
    else hipLaunchKernelGGL((k_conv_narrow<2, 2>), dim3(grid), dim3(256), 0, st, a, tx, ty);
    g_last_conv_kernel = CK_NARROW;
    return 1;
}

// ------------------------------------------------------------------------------------------------------------------------------
// wgrad of the same layers when one side has only 16 channels (the 32x32x2 tile of k_conv_wgrad_small is 50-75 % padding there).
// M = 16 output channels, N = 16 input channels, K = 4 horizontally adjacent pixels per MFMA; operands are single floats read
// from [pixel][channel] LDS tiles whose pitch (16 floats for 16 channels, 48 for 32) puts the four k-lanes (pixels x .. x+3)
// on four disjoint 16-bank ranges.  A wave owns two rows of the 8x32 tile, keeps all nine taps' accumulators in registers over a
// persistent tile loop (one dY fragment + nine shifted X fragments per 9 MFMAs) and flushes once with fp32 atomics.
// ------------------------------------------------------------------------------------------------------------------------------
namespace {
template <int CI16, int CO16>
__global__ __launch_bounds__(256) void k_wgrad_narrow(WgradArgs a, int tiles_x, int tiles_y) {
    constexpr int PX = CI16 == 1 ? 16 : 48, PY = CO16 == 1 ? 16 : 48;
    constexpr int QX = 4 * CI16, QY = 4 * CO16;
    constexpr int NX = (NHH * NHW * QX + 255) / 256, NY = NTH * NTW * QY / 256;
    __shared__ float xs[NHH * NHW * PX];
    __shared__ float ys[NTH * NTW * PY];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lp = lane & 15, g = lane >> 4;
    const ConvSrc s = a.src[0];
    const long ntiles = (long)a.N * tiles_x * tiles_y;
    f32x4 acc[9][CO16][CI16];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int m = 0; m < CO16; m++)
#pragma unroll
            for (int n = 0; n < CI16; n++) acc[t][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 px[NX], py[NY];
    auto gload = [&](long tile) {
        int n = (int)(tile / (tiles_x * tiles_y));
        int rem = (int)(tile - (long)n * tiles_x * tiles_y);
        int ty = rem / tiles_x;
        int y0 = ty * NTH, x0 = (rem - ty * tiles_x) * NTW;
        const float* bx = s.p + (long)n * s.sn;
        const float* by = a.dy + (long)n * a.dy_sn;
#pragma unroll
        for (int i = 0; i < NX; i++) {
            int idx = tid + 256 * i, hp = idx / QX, c = (idx % QX) * 4;
            int hy = hp / NHW, hx = hp - hy * NHW;
            int y = y0 - 1 + hy, x = x0 - 1 + hx;
            px[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (hp < NHH * NHW && y >= 0 && y < a.H && x >= 0 && x < a.W && c < s.C) px[i] = nld4(bx + ((long)y * a.W + x) * s.ld + c, c, s.C);
        }
#pragma unroll
        for (int i = 0; i < NY; i++) {
            int idx = tid + 256 * i, p = idx / QY, c = (idx % QY) * 4;
            int y = y0 + p / NTW, x = x0 + (p & (NTW - 1));
            py[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y < a.H && x < a.W && c < a.Cout) py[i] = nld4(by + ((long)y * a.W + x) * a.dy_ld + c, c, a.Cout);
        }
    };

    long tile = blockIdx.x;
    if (tile < ntiles) gload(tile);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int i = 0; i < NX; i++) {
            int idx = tid + 256 * i, hp = idx / QX;
            if (hp < NHH * NHW) *reinterpret_cast<float4*>(&xs[hp * PX + (idx % QX) * 4]) = px[i];
        }
#pragma unroll
        for (int i = 0; i < NY; i++) {
            int idx = tid + 256 * i;
            *reinterpret_cast<float4*>(&ys[(idx / QY) * PY + (idx % QY) * 4]) = py[i];
        }
        __syncthreads();
        if (tile + gridDim.x < ntiles) gload(tile + gridDim.x);
#pragma unroll 2
        for (int grp = 0; grp < 16; grp++) {                 // 4-pixel groups of this wave's two rows
            const int row = 2 * wave + (grp >> 3), x0 = 4 * (grp & 7) + g;
            float fa[CO16];
#pragma unroll
            for (int m = 0; m < CO16; m++) fa[m] = ys[(row * NTW + x0) * PY + m * 16 + lp];
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int dy = t / 3, dx = t - 3 * dy;
                const float* xp = &xs[((row + dy) * NHW + x0 + dx) * PX + lp];
#pragma unroll
                for (int n = 0; n < CI16; n++) {
                    const float fb = xp[n * 16];
#pragma unroll
                    for (int m = 0; m < CO16; m++) acc[t][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m], fb, acc[t][m][n], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // fold the four waves' partial sums in LDS (ds_add_f32 on the now idle X tile), then one global atomic per weight and workgroup
    constexpr int NW = 9 * CO16 * 16 * CI16 * 16;
    static_assert(NW <= NHH * NHW * PX, "LDS reduction buffer must fit in the X tile");
    float* red = xs;
    for (int i = tid; i < NW; i += 256) red[i] = 0.f;
    __syncthreads();
    // (the waves take turns, in order: one wave's lanes hit distinct elements, so plain read-modify-write -- and a fixed summation order, where ds_add_f32 from four
    //  waves at once summed in arrival order)
    for (int wv = 0; wv < 4; wv++) {
        if (wave == wv) {
#pragma unroll
            for (int t = 0; t < 9; t++)
#pragma unroll
                for (int m = 0; m < CO16; m++)
#pragma unroll
                    for (int n = 0; n < CI16; n++)
#pragma unroll
                        for (int r = 0; r < 4; r++) red[((t * CO16 * 16) + m * 16 + 4 * g + r) * (CI16 * 16) + n * 16 + lp] += acc[t][m][n][r];
        }
        __syncthreads();
    }
    float* const dst = WGRAD_DST(a, blockIdx.x);
    for (int i = tid; i < NW; i += 256) {
        int k = i % (CI16 * 16), to = i / (CI16 * 16);
        int o = to % (CO16 * 16), t = to / (CO16 * 16);
        if (k < a.Ktot && o < a.Cout) atomicAdd(dst + ((long)t * a.Cout_pad + o) * a.Ktot + k, red[i]);
    }
}
}  // namespace

int conv_narrow_wgrad_try(const WgradArgs& a, hipStream_t st, bool dry) {
    if (a.nsrc != 1 || a.src[0].bcast || a.KS != 3) return 0;
    if (a.Ktot > 32 || a.Cout > 32 || a.src[0].C <= 12 || a.Cout <= 4) return 0;
    const int ci = a.Ktot / 16, co = (a.Cout + 15) / 16;
    if (ci == 2 && co == 2) return 0;                              // 32 x 32: no padding in the 32x32x2 kernel
    if ((a.src[0].ld & 3) || (a.src[0].sn & 3) || (a.dy_ld & 3) || (a.dy_sn & 3)) return 0;
    if ((long)a.N * a.H * a.W < 4096) return 0;
    const int tx = cdiv(a.W, NTW), ty = cdiv(a.H, NTH);
    const long ntiles = (long)a.N * tx * ty;
    long want = ntiles / 4;                                        // >= 4 tiles per workgroup amortise the flush
    const int grid = (int)(want < 64 ? (ntiles < 64 ? ntiles : 64) : (want < 512 ? want : 512));
    g_last_conv_kernel = CK_WGRAD_SMALL;
    if (dry) return 1;
    WgradArgs b = a;
    int gridn = grid;
    if (b.det_slab) { gridn = (int)wgrad_det_begin(b, grid, st); if (gridn <= 0) return -1; }
    if (ci == 1 && co == 1) hipLaunchKernelGGL((k_wgrad_narrow<1, 1>), dim3(gridn), dim3(256), 0, st, b, tx, ty);
    else if (ci == 1) hipLaunchKernelGGL((k_wgrad_narrow<1, 2>), dim3(gridn), dim3(256), 0, st, b, tx, ty);
    else hipLaunchKernelGGL((k_wgrad_narrow<2, 1>), dim3(gridn), dim3(256), 0, st, b, tx, ty);
    if (b.det_slab) wgrad_det_end(b, gridn, st);
    g_last_conv_kernel = CK_WGRAD_SMALL;
    return 1;
}

// ------------------------------------------------------------------------------------------------------------------------------
// 3-channel INPUT (pitch 4) convolutions: E's stem and the dgrad of the FinalBlocks (3x3 and 7x7).  v_mfma_f32_16x16x4_f32's K = 4
// is exactly one pixel's padded channel vector, so one MFMA per (tap, 16 output channels, 16 pixels) with NO channel padding beyond
// 3 -> 4: B operand = xs[pixel + tap][lane >> 4] straight from the staged halo tile (64 consecutive floats per wave read:
// conflict-free), A operand = the tap's weight column held in registers.  Replaces the vector-ALU k_conv_thin_in (200 us -> see
// DESIGN.md for the measured figure on the 7x7 3->32 dgrad at 256x256).
// ------------------------------------------------------------------------------------------------------------------------------
namespace {
template <int KS, int NT>
__global__ __launch_bounds__(256) void k_conv_c4(ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int R = KS / 2, HW_ = NTW + 2 * R, HH_ = NTH + 2 * R, TAPS = KS * KS;
    constexpr int NL = (HH_ * HW_ + 255) / 256;
    __shared__ float xs[HH_ * HW_ * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lp = lane & 15, g = lane >> 4;
    const ConvSrc s = a.src[0];
    const long ntiles = (long)a.N * tiles_x * tiles_y;
    const int og = blockIdx.y * NT * 16;                    // first output channel of this workgroup

    // A operands (weight column of lane (cout lp, channel g)) live in LDS as [tap][m][lane]: 64 consecutive floats per read, and only the
    // accumulators stay in registers -> 4+ waves per SIMD hide the staging / epilogue behind other waves' MFMAs (holding all 49 x NT
    // weights in registers forced 1 wave per SIMD)
    __shared__ float ws[TAPS * NT * 64];
    for (int i = tid; i < TAPS * NT * 64; i += 256) {
        int l = i & 63, tm = i >> 6;
        int m = tm % NT, t = tm / NT;
        ws[i] = a.wp[((long)t * a.Cout_pad + og + m * 16 + (l & 15)) * a.Ktot + (l >> 4)];
    }
    float4 pre[NL];
    auto gload = [&](long tile) {
        int n = (int)(tile / (tiles_x * tiles_y));
        int rem = (int)(tile - (long)n * tiles_x * tiles_y);
        int ty = rem / tiles_x;
        int y0 = ty * NTH, x0 = (rem - ty * tiles_x) * NTW;
        const float* base = s.p + (long)n * s.sn;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            int hp = tid + 256 * i;
            int hy = hp / HW_, hx = hp - hy * HW_;
            int y = y0 - R + hy, x = x0 - R + hx;
            pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (hp < HH_ * HW_ && y >= 0 && y < a.H && x >= 0 && x < a.W) pre[i] = nld4(base + ((long)y * a.W + x) * 4, 0, s.C);
        }
    };
    long tile = blockIdx.x;
    if (tile < ntiles) gload(tile);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            int hp = tid + 256 * i;
            if (hp < HH_ * HW_) *reinterpret_cast<float4*>(&xs[hp * 4]) = pre[i];
        }
        __syncthreads();
        if (tile + gridDim.x < ntiles) gload(tile + gridDim.x);
        f32x4 acc[4][NT];
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int m = 0; m < NT; m++) acc[nt][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int dy = 0; dy < KS; dy++) {
#pragma unroll
            for (int dx = 0; dx < KS; dx++) {
                float wa[NT];
#pragma unroll
                for (int m = 0; m < NT; m++) wa[m] = ws[((dy * KS + dx) * NT + m) * 64 + lane];
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    const float b = xs[((2 * wave + (nt >> 1) + dy) * HW_ + 16 * (nt & 1) + lp + dx) * 4 + g];
#pragma unroll
                    for (int m = 0; m < NT; m++) acc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m], b, acc[nt][m], 0, 0, 0);
                }
            }
        }
        {
            int n = (int)(tile / (tiles_x * tiles_y));
            int rem = (int)(tile - (long)n * tiles_x * tiles_y);
            int ty = rem / tiles_x;
            int y0 = ty * NTH, x0 = (rem - ty * tiles_x) * NTW;
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                int y = y0 + 2 * wave + (nt >> 1), x = x0 + 16 * (nt & 1) + lp;
                if (y >= a.H || x >= a.W) continue;
                float* o = a.out + (long)n * a.out_sn + ((long)y * a.W + x) * a.out_ld;
#pragma unroll
                for (int m = 0; m < NT; m++) {
                    int c = og + m * 16 + 4 * g;
                    if (c >= a.Cout) continue;
                    float v[4] = {acc[nt][m][0], acc[nt][m][1], acc[nt][m][2], acc[nt][m][3]};
                    if (a.bias) for (int e = 0; e < 4; e++) if (c + e < a.Cout) v[e] += a.bias[c + e];
                    if (a.res) {
                        const float* rp = a.res + (long)n * a.res_sn + ((long)y * a.W + x) * a.res_ld + c;
                        for (int e = 0; e < 4; e++) if (c + e < a.Cout) v[e] += rp[e];
                    }
                    if (a.act == 3) for (int e = 0; e < 4; e++) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
                    if (c + 4 <= a.Cout) {
                        float4 r = make_float4(v[0], v[1], v[2], v[3]);
                        if (a.accumulate) { float4 p = *reinterpret_cast<const float4*>(o + c); r.x += p.x; r.y += p.y; r.z += p.z; r.w += p.w; }
                        *reinterpret_cast<float4*>(o + c) = r;
                    } else {
                        for (int e = 0; e < 4 && c + e < a.Cout; e++) o[c + e] = a.accumulate ? o[c + e] + v[e] : v[e];
                    }
                }
            }
        }
        __syncthreads();
    }
}
}  // namespace

int conv_c4_fwd_try(const ConvArgs& a, hipStream_t st) {
    if (a.nsrc != 1 || a.src[0].bcast || a.act != 0 || a.splitk > 1 || (a.KS != 3 && a.KS != 7)) return 0;
    if (a.src[0].C > 4 || a.src[0].ld != 4 || (a.src[0].sn & 3) || a.Ktot != 16 || a.Cout < 8 || (a.out_ld & 3) || (a.out_sn & 3)) return 0;
    const int mt = (a.Cout + 15) / 16;                       // 16-channel output tiles
    if (a.Cout_pad < mt * 16) return 0;
    const int tx = cdiv(a.W, NTW), ty = cdiv(a.H, NTH);
    const long ntiles = (long)a.N * tx * ty;
    const int nt = a.KS == 7 ? (mt >= 2 ? 2 : 1) : (mt >= 4 ? 4 : (mt >= 2 ? 2 : 1));      // 16-channel tiles per workgroup (register budget)
    const int groups = cdiv(mt, nt);
    long gx = 1024 / groups; if (gx < 64) gx = 64; if (gx > ntiles) gx = ntiles;
    dim3 grid((unsigned)gx, groups);
#define C4_LAUNCH(KS_, NT_) hipLaunchKernelGGL((k_conv_c4<KS_, NT_>), grid, dim3(256), 0, st, a, tx, ty)
    if (a.KS == 7) { if (nt == 2) C4_LAUNCH(7, 2); else C4_LAUNCH(7, 1); }
    else if (nt == 4) C4_LAUNCH(3, 4);
    else if (nt == 2) C4_LAUNCH(3, 2);
    else C4_LAUNCH(3, 1);
#undef C4_LAUNCH
    g_last_conv_kernel = CK_THIN_IN;
    return 1;
}

// ------------------------------------------------------------------------------------------------------------------------------
// wgrad with a 3-channel (pitch 4) side: G[tap][t][w] = sum_p Wide[p][w] * Thin[p + sigma * off(tap)][t]
//   swap = 0 (FinalBlocks: thin = dY, wide = X, sigma = -1)  -> dwp[(tap * Cout_pad + t) * Ktot + w]
//   swap = 1 (E's stem:    thin = X,  wide = dY, sigma = +1) -> dwp[(tap * Cout_pad + w) * Ktot + t]
// 16x16x4 MFMA: M = 16 wide channels, N = 16 = (4 horizontal taps) x (4 thin channels), K = 4 horizontally adjacent pixels.  A 7x7
// layer is 7 x 2 such column groups: per 4-pixel group one wide fragment + 14 thin fragments (all single-float LDS reads, the thin
// ones mostly broadcasts) feed 14 MFMAs.  Waves own two rows of the 8x32 tile, accumulate over a persistent tile loop, fold in LDS and
// flush once with atomics.  blockIdx.y = 16-channel chunk of the wide tensor.  Replaces the vector-ALU k_wgrad_thin for C <= 4.
// ------------------------------------------------------------------------------------------------------------------------------
namespace {
template <int KS>
__global__ __launch_bounds__(256) void k_wgrad_c4(const float* thin, long thin_sn, int TC, const float* wide, long wide_sn, int wide_ld, int WC,
                                                   int N, int H, int W, int swap, int Cout_pad, int Ktot, float* dwp_all, int tiles_x, int tiles_y, float* det_slab, long det_stride) {
    float* const dwp = det_slab ? det_slab + (long)blockIdx.x * det_stride : dwp_all;      // deterministic mode: this pixel split's own copy (WgradArgs.det_slab)
    constexpr int R = KS / 2, HW_ = NTW + 2 * R, HH_ = NTH + 2 * R, DXG = (KS + 3) / 4, TAPS = KS * KS;
    constexpr int NLT = (HH_ * HW_ + 255) / 256, NLW = NTH * NTW * 4 / 256;
    constexpr int NG = KS * DXG * 16 * 16;                     // floats of the per-workgroup partial G (dy, dx group, column, row)
    __shared__ float th[HH_ * HW_ * 4];
    __shared__ float wd[(NTH * NTW * 16 > NG) ? NTH * NTW * 16 : NG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lp = lane & 15, g = lane >> 4;
    const int w0 = blockIdx.y * 16;
    const int sg = swap ? 1 : -1;
    const long ntiles = (long)N * tiles_x * tiles_y;
    f32x4 acc[KS][DXG];
#pragma unroll
    for (int i = 0; i < KS; i++)
#pragma unroll
        for (int j = 0; j < DXG; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this lane's column = (horizontal tap dxi, thin channel t); per column group the tap is dx = 4*dxg + dxi (clamped: extra columns are dropped)
    const int dxi = lp >> 2, tch = lp & 3;
    int hoff[DXG];
#pragma unroll
    for (int j = 0; j < DXG; j++) { int dx = 4 * j + dxi; if (dx > KS - 1) dx = KS - 1; hoff[j] = (R + sg * (dx - R)) * 4 + tch; }

    float4 pt[NLT], pw[NLW];
    auto gload = [&](long tile) {
        int n = (int)(tile / (tiles_x * tiles_y));
        int rem = (int)(tile - (long)n * tiles_x * tiles_y);
        int ty = rem / tiles_x;
        int y0 = ty * NTH, x0 = (rem - ty * tiles_x) * NTW;
#pragma unroll
        for (int i = 0; i < NLT; i++) {
            int hp = tid + 256 * i;
            int hy = hp / HW_, hx = hp - hy * HW_;
            int y = y0 - R + hy, x = x0 - R + hx;
            pt[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (hp < HH_ * HW_ && y >= 0 && y < H && x >= 0 && x < W) pt[i] = nld4(thin + (long)n * thin_sn + ((long)y * W + x) * 4, 0, TC);
        }
#pragma unroll
        for (int i = 0; i < NLW; i++) {
            int idx = tid + 256 * i, p = idx >> 2, c = w0 + (idx & 3) * 4;
            int y = y0 + p / NTW, x = x0 + (p & (NTW - 1));
            pw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y < H && x < W && c < WC) pw[i] = nld4(wide + (long)n * wide_sn + ((long)y * W + x) * wide_ld + c, c, WC);
        }
    };
    long tile = blockIdx.x;
    if (tile < ntiles) gload(tile);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int i = 0; i < NLT; i++) { int hp = tid + 256 * i; if (hp < HH_ * HW_) *reinterpret_cast<float4*>(&th[hp * 4]) = pt[i]; }
#pragma unroll
        for (int i = 0; i < NLW; i++) { int idx = tid + 256 * i; *reinterpret_cast<float4*>(&wd[(idx >> 2) * 16 + (idx & 3) * 4]) = pw[i]; }
        __syncthreads();
        if (tile + gridDim.x < ntiles) gload(tile + gridDim.x);
#pragma unroll 2
        for (int grp = 0; grp < 16; grp++) {
            const int row = 2 * wave + (grp >> 3), x0 = 4 * (grp & 7) + g;
            const float fa = wd[(row * NTW + x0) * 16 + lp];
#pragma unroll
            for (int i = 0; i < KS; i++) {
                const float* tp = &th[((row + R + sg * (i - R)) * HW_ + x0) * 4];
#pragma unroll
                for (int j = 0; j < DXG; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, tp[hoff[j]], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // fold the four waves in LDS, then one global atomic per element and workgroup
    float* red = wd;
    for (int i = tid; i < NG; i += 256) red[i] = 0.f;
    __syncthreads();
    for (int wv = 0; wv < 4; wv++) {      // waves in turn: fixed summation order (see k_wgrad_narrow)
        if (wave == wv) {
#pragma unroll
            for (int i = 0; i < KS; i++)
#pragma unroll
                for (int j = 0; j < DXG; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) red[((i * DXG + j) * 16 + lp) * 16 + 4 * g + r] += acc[i][j][r];
        }
        __syncthreads();
    }
    for (int e = tid; e < NG; e += 256) {
        int wr = e & 15, col = (e >> 4) & 15, ij = e >> 8;
        int j = ij % DXG, i = ij / DXG;
        int dx = 4 * j + (col >> 2), t = col & 3, wch = w0 + wr;
        if (dx >= KS || t >= TC || wch >= WC) continue;
        int tap = i * KS + dx;
        float* d = swap ? dwp + ((long)tap * Cout_pad + wch) * Ktot + t : dwp + ((long)tap * Cout_pad + t) * Ktot + wch;
        atomicAdd(d, red[e]);
    }
    (void)TAPS;
}
}  // namespace

int conv_c4_wgrad_try(const WgradArgs& w, hipStream_t st, bool dry) {
    if (w.nsrc != 1 || w.src[0].bcast || (w.KS != 3 && w.KS != 7)) return 0;
    const float *thin, *wide; long thin_sn, wide_sn; int TC, WC, wide_ld, swap;
    if (w.Cout <= 4 && w.dy_ld == 4 && w.src[0].C >= 16) {            // FinalBlocks: thin = dY
        swap = 0; thin = w.dy; thin_sn = w.dy_sn; TC = w.Cout; wide = w.src[0].p; wide_sn = w.src[0].sn; wide_ld = w.src[0].ld; WC = w.src[0].C;
    } else if (w.src[0].C <= 4 && w.src[0].ld == 4 && w.Cout >= 8) {   // stem / FinalBlock dgrad side: thin = X
        swap = 1; thin = w.src[0].p; thin_sn = w.src[0].sn; TC = w.src[0].C; wide = w.dy; wide_sn = w.dy_sn; wide_ld = w.dy_ld; WC = w.Cout;
    } else return 0;
    if ((thin_sn & 3) || (wide_sn & 3) || (wide_ld & 3)) return 0;
    const int tx = cdiv(w.W, NTW), ty = cdiv(w.H, NTH), chunks = cdiv(WC, 16);
    const long ntiles = (long)w.N * tx * ty;
    long want = ntiles / 4, cap = 512 / chunks > 32 ? 512 / chunks : 32;
    long gx = want < 32 ? (ntiles < 32 ? ntiles : 32) : (want < cap ? want : cap);
    g_last_conv_kernel = CK_WGRAD_THIN;
    if (dry) return 1;
    WgradArgs b = w;
    if (b.det_slab) { gx = wgrad_det_begin(b, gx, st); if (gx <= 0) return -1; }
    dim3 grid((unsigned)gx, chunks);
    if (w.KS == 7) hipLaunchKernelGGL((k_wgrad_c4<7>), grid, dim3(256), 0, st, thin, thin_sn, TC, wide, wide_sn, wide_ld, WC, w.N, w.H, w.W, swap, w.Cout_pad, w.Ktot, w.dwp, tx, ty, b.det_slab, b.det_stride);
    else hipLaunchKernelGGL((k_wgrad_c4<3>), grid, dim3(256), 0, st, thin, thin_sn, TC, wide, wide_sn, wide_ld, WC, w.N, w.H, w.W, swap, w.Cout_pad, w.Ktot, w.dwp, tx, ty, b.det_slab, b.det_stride);
    if (b.det_slab) wgrad_det_end(b, gx, st);
    g_last_conv_kernel = CK_WGRAD_THIN;
    return 1;
}
