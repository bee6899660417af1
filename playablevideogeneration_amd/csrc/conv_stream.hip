// HBM-bound weight gradients as streaming kernels (round 5).
//
// The 1x1 convolutions of the model are the identity paths of the down-sampling ResidualBlocks (model/layers/residual_block.py:38-46: conv1x1 -> avg-pool -> BatchNorm): 16 ... 64 input
// channels, 32 ... 128 output channels, up to 2.1 M pixels per launch (E on the 128 ground-truth frames).  dW[o][k] = sum_p dY[p][o] X[p][k] is a GEMM whose reduction side is the
// PIXELS and whose result is a few KB: the work is reading both operands once -- 50 us at the HBM peak for the largest launch -- while the generic implicit-GEMM weight-gradient kernel
// (k_conv_wgrad: im2col staging through LDS per tap, built for wide layers) spent 250 - 660 us on them at 2 - 3 TFLOP/s (profiles/r04_*_phases_and_layers.txt).
//
// Here both NHWC operands ARE the MFMA operands, straight from global memory, no LDS: v_mfma_f32_32x32x2_f32 wants A[i = lane & 31][k = lane >> 5] and B[k = lane >> 5][j = lane & 31];
// with i = output channel, j = input channel and k = pixel that is `dY[p + (lane >> 5)][o0 + (lane & 31)]` and `X[p + (lane >> 5)][k0 + (lane & 31)]` -- two coalesced 128-byte rows per
// operand load.  A wave owns every (32 x 32) block of the small result and an interleaved share of a workgroup's pixel slab; U pixel pairs are requested back to back before the
// first MFMA (the loads, not the 64-cycle MFMAs, are what has to be kept in flight); the four waves fold in LDS in wave order and the workgroups meet in dwp through fp32 atomics, or
// -- bit-reproducible mode -- in one slab per workgroup that wgrad_det_end folds in a fixed order.  Exact fp32 arithmetic (k-ordered fma chain of the fp32 MFMA).
#include "common.h"

namespace {

constexpr int SW_U = 8;      // pixel pairs in flight per wave and trip

// NOB / NKB: 32-channel blocks on the output / input side.  grid = (slabs per sample, samples); sample n of a time-batched launch lives in group n / group_n (WgradArgs.group_n).
template <int NOB, int NKB>
__global__ __launch_bounds__(256) void k_wgrad_1x1(WgradArgs a, int pix_per_slab) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    int n = blockIdx.y;
    const float* X = a.src[0].p;
    const float* Y = a.dy;
    if (a.group_n > 0) { const int grp = n / a.group_n; n -= grp * a.group_n; X += grp * a.src_gs[0]; Y += grp * a.dy_gs; }
    X += (long)n * a.src[0].sn; Y += (long)n * a.dy_sn;
    const int HW = a.H * a.W;
    const int p_beg = (int)blockIdx.x * pix_per_slab, p_end = p_beg + pix_per_slab < HW ? p_beg + pix_per_slab : HW;
    const int xld = a.src[0].ld, yld = a.dy_ld;
    bool oko[NOB], okk[NKB];
#pragma unroll
    for (int ob = 0; ob < NOB; ob++) oko[ob] = ob * 32 + col < a.Cout;
#pragma unroll
    for (int kb = 0; kb < NKB; kb++) okk[kb] = kb * 32 + col < a.src[0].C;
    f32x16 acc[NOB][NKB];
#pragma unroll
    for (int ob = 0; ob < NOB; ob++)
#pragma unroll
        for (int kb = 0; kb < NKB; kb++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[ob][kb][r] = 0.f;
    for (int p0 = p_beg + wave * 2 * SW_U; p0 < p_end; p0 += 4 * 2 * SW_U) {
        float ya[SW_U][NOB], xb[SW_U][NKB];
#pragma unroll
        for (int u = 0; u < SW_U; u++) {      // loads only (clamped addresses): everything requested before the first use
            const int p = p0 + 2 * u + half;
            const int pc = p < p_end ? p : p_beg;
#pragma unroll
            for (int ob = 0; ob < NOB; ob++) ya[u][ob] = Y[(long)pc * yld + (oko[ob] ? ob * 32 + col : 0)];
#pragma unroll
            for (int kb = 0; kb < NKB; kb++) xb[u][kb] = X[(long)pc * xld + (okk[kb] ? kb * 32 + col : 0)];
        }
#pragma unroll
        for (int u = 0; u < SW_U; u++) {
            const bool pv = p0 + 2 * u + half < p_end;
#pragma unroll
            for (int ob = 0; ob < NOB; ob++) {
                const float av = (pv && oko[ob]) ? ya[u][ob] : 0.f;
#pragma unroll
                for (int kb = 0; kb < NKB; kb++) acc[ob][kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, okk[kb] ? xb[u][kb] : 0.f, acc[ob][kb], 0, 0, 0);
            }
        }
    }
    // the four waves' partial results meet in LDS in wave order (fixed: the bit-reproducible mode needs it, and it is a quarter of the atomics otherwise); wave 0 flushes
    __shared__ float red[NOB * NKB * 16 * 64];
    for (int w = 1; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int ob = 0; ob < NOB; ob++)
#pragma unroll
                for (int kb = 0; kb < NKB; kb++)
#pragma unroll
                    for (int r = 0; r < 16; r++) red[((ob * NKB + kb) * 16 + r) * 64 + lane] = acc[ob][kb][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int ob = 0; ob < NOB; ob++)
#pragma unroll
                for (int kb = 0; kb < NKB; kb++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[ob][kb][r] += red[((ob * NKB + kb) * 16 + r) * 64 + lane];
        }
        __syncthreads();
    }
    if (wave != 0) return;
    // D fragment: column j = lane & 31 (input channel), row i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (output channel)
    float* const dst = WGRAD_DST(a, blockIdx.y * gridDim.x + blockIdx.x);
#pragma unroll
    for (int ob = 0; ob < NOB; ob++)
#pragma unroll
        for (int kb = 0; kb < NKB; kb++) {
            const int k = kb * 32 + col;
            if (k >= a.Ktot) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int o = ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (o < a.Cout) atomicAdd(dst + (long)o * a.Ktot + k, acc[ob][kb][r]);
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 7x7 FinalBlock head (C -> 3 channels, model/layers/final_block.py:9-29, rendering_network.py:41) on v_mfma_f32_16x16x32_bf16, split bf16.
//
//   dW[o][c][ty][tx] = sum_{y,x} dY[y][x][o] X[y + ty - 3][x + tx - 3][c] = sum_q dY[q_y - ty + 3][q_x - tx + 3][o] X[q][c]
//
// The second form puts the 49 taps on the M side: GEMM-M = (tap, o) -- 147 rows, ten 16-row blocks --, GEMM-N = the C <= 32 input channels (two 16-column halves), reduction = the
// pixels q, 32 per instruction (a row segment of the tile).  The first form's natural mapping (M = the 3 output channels padded to 16, one instruction per tap) issues 49 x 2 x 3
// instructions per 32 pixels with 13 of 16 rows empty; this one issues 10 x 2 x 3 -- the arithmetic all but disappears (7 us per 8 frames at 256 x 256) and the launch is bound by
// reading X once (k_wgrad_c4<7>: 770 us per five time steps at 32 TFLOP/s, 17 x its HBM bound; profiles/r05_*_phases_and_layers.txt).
//   * B fragment = X[32 pixels][16 channels], transposed on the way out of a [pixel][hi 32 | lo 32] LDS image by ds_read_b64_tr_b16 (as in k_wgrad_hx) -- unshifted, shared by all rows.
//   * A fragment = for row (tap, o) the EIGHT CONSECUTIVE pixels dY[q_y - ty + 3][q_x + (3 - tx) .. + 7][o]: a row shift is an address offset, but the column shift 3 - tx would make
//     the 16-byte read unaligned.  dY has only 3 channels, so the tile keeps SEVEN copies of it in LDS, copy d holding dY shifted by d = -3 .. 3 columns, channel-planar, hi / lo planes:
//     every A fragment is one aligned ds_read_b128 (48 KB for a 4 x 64 tile; the staging writes each dY value 14 times, 2-byte stores).
// A workgroup walks 4 x 64-pixel tiles persistently; its four waves take the tile's eight 32-pixel groups two each and keep all 10 x 2 accumulator blocks (80 registers); one LDS fold
// in wave order and one flush (atomics, or the workgroup's own slab in bit-reproducible mode) at the very end.
// ------------------------------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int H7_TR = 4, H7_TC = 64;                          // tile: rows x columns of X pixels
constexpr int H7_XP = 80;                                    // X image: halves per pixel (hi 32 | lo 32 | pad 16): 160 B -- four consecutive pixel rows of a transposing read on disjoint bank octets
constexpr int H7_YR = H7_TR + 6, H7_YP = H7_TC + 8;           // dY copies: rows (3 above / below the tile) x pitch in halves
constexpr int H7_YPLANE = H7_YR * H7_YP;                      // one (copy, channel, plane) image
constexpr int H7_MB = 10;                                    // 16-row blocks of the 147 (tap, o) rows

__device__ __forceinline__ bf16x8 h7_tr_frag(const __bf16* base, int off0, int off1) {
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off0));
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off1));
    return u.v;
}

__global__ __launch_bounds__(256) void k_wgrad_head7(WgradArgs a, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) __bf16 Xs[H7_TR * H7_TC * H7_XP];       // 40 KB (reused by the final fold)
    __shared__ __attribute__((aligned(16))) __bf16 Ys[7 * 3 * 2 * H7_YPLANE];       // 60 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.src[0].C;
    const int ntiles = a.N * tiles_x * tiles_y;
    // A fragment addresses: row i = lane & 15 of block mb is (tap, o) = (m / 3, m % 3), m = 16 mb + i (rows >= 147 repeat row 146: never flushed); k-octet = lane >> 4
    int aoff[H7_MB];
#pragma unroll
    for (int mb = 0; mb < H7_MB; mb++) {
        int m = mb * 16 + (lane & 15); m = m < 147 ? m : 146;
        const int tap = m / 3, o = m - 3 * tap, ty = tap / 7, tx = tap - 7 * ty;
        aoff[mb] = (((6 - tx) * 3 + o) * 2) * H7_YPLANE + (6 - ty) * H7_YP + 8 * (lane >> 4);      // copy d = 3 - tx is stored at index d + 3 = 6 - tx; image row of X row r: r - ty + 6
    }
    // B fragment: k-octet lane >> 4 -> pixels 8 (lane >> 4) + ((lane & 15) >> 2) (+ 4), column quad 4 (lane & 3)
    const int boff = ((lane >> 4) * 8 + ((lane & 15) >> 2)) * H7_XP + 4 * (lane & 3);
    f32x4 acc[H7_MB][2];
#pragma unroll
    for (int mb = 0; mb < H7_MB; mb++) { acc[mb][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mb][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int q = tid & 7;                                    // X staging: float4 column q of pixels (tid >> 3) + 32 i
    const int xld = a.src[0].ld;
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
        int n = tile / (tiles_x * tiles_y);
        const int rem = tile - n * tiles_x * tiles_y;
        const int tyi = rem / tiles_x, y0 = tyi * H7_TR, x0 = (rem - tyi * tiles_x) * H7_TC;
        const float* X = a.src[0].p;
        const float* Y = a.dy;
        if (a.group_n > 0) { const int grp = n / a.group_n; n -= grp * a.group_n; X += grp * a.src_gs[0]; Y += grp * a.dy_gs; }
        X += (long)n * a.src[0].sn; Y += (long)n * a.dy_sn;
        // ---- loads (X: 256 pixels x 8 float4; dY: 10 x 70 pixels x float4), then split + stores ----
        float4 rx[8], ry[3];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int p = (tid >> 3) + 32 * i, r = p >> 6, xc = p & 63;
            const bool ok = y0 + r < a.H && x0 + xc < a.W && 4 * q < C;
            rx[i] = ok ? *reinterpret_cast<const float4*>(X + ((long)(y0 + r) * a.W + x0 + xc) * xld + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int idx = tid + 256 * i, yy = idx / 70, xx = idx - 70 * yy;
            const int y = y0 - 3 + yy, x = x0 - 3 + xx;
            const bool ok = idx < H7_YR * 70 && y >= 0 && y < a.H && x >= 0 && x < a.W;
            ry[i] = ok ? *reinterpret_cast<const float4*>(Y + ((long)y * a.W + x) * a.dy_ld) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();                                      // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int p = (tid >> 3) + 32 * i;
            float v[4] = {rx[i].x, rx[i].y, rx[i].z, rx[i].w};
            bf16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; e++) { if (4 * q + e >= C) v[e] = 0.f; hi[e] = (__bf16)v[e]; lo[e] = (__bf16)(v[e] - (float)hi[e]); }
            *reinterpret_cast<bf16x4*>(&Xs[p * H7_XP + 4 * q]) = hi;
            *reinterpret_cast<bf16x4*>(&Xs[p * H7_XP + 32 + 4 * q]) = lo;
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int idx = tid + 256 * i, yy = idx / 70, xx = idx - 70 * yy;
            if (idx >= H7_YR * 70) continue;
            const float v[3] = {ry[i].x, ry[i].y, ry[i].z};
#pragma unroll
            for (int o = 0; o < 3; o++) {
                const float vo = o < a.Cout ? v[o] : 0.f;
                const __bf16 hi = (__bf16)vo, lo = (__bf16)(vo - (float)hi);
#pragma unroll
                for (int d = -3; d <= 3; d++) {               // copy d holds dY[y][x0 + xc + d] at column xc: this pixel (x = x0 - 3 + xx) lands at xc = xx - 3 - d
                    const int xc = xx - 3 - d;
                    if (xc >= 0 && xc < H7_TC) {
                        __bf16* dst = &Ys[(((d + 3) * 3 + o) * 2) * H7_YPLANE + yy * H7_YP + xc];
                        dst[0] = hi; dst[H7_YPLANE] = lo;
                    }
                }
            }
        }
        __syncthreads();
        // ---- this wave's two 32-pixel groups: g = 2 wave, 2 wave + 1 -> tile row g >> 1, column half g & 1 ----
#pragma unroll 1
        for (int gi = 0; gi < 2; gi++) {
            const int g = 2 * wave + gi, r = g >> 1, cg = g & 1;
            const __bf16* xb = Xs + (r * H7_TC + 32 * cg) * H7_XP + boff;
            bf16x8 bh[2], bl[2];
#pragma unroll
            for (int h = 0; h < 2; h++) { bh[h] = h7_tr_frag(xb, 16 * h, 16 * h + 4 * H7_XP); bl[h] = h7_tr_frag(xb, 32 + 16 * h, 32 + 16 * h + 4 * H7_XP); }
            const int ybase = r * H7_YP + 32 * cg;
#pragma unroll
            for (int mb = 0; mb < H7_MB; mb++) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Ys[aoff[mb] + ybase]);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(&Ys[aoff[mb] + ybase + H7_YPLANE]);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    f32x4 c = acc[mb][h];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[h], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[h], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[h], c, 0, 0, 0);
                    acc[mb][h] = c;
                }
            }
        }
    }
    // ---- fold the four waves in wave order (X image as scratch: 80 floats x 64 lanes = 20 KB), wave 0 flushes ----
    float* red = reinterpret_cast<float*>(Xs);
    for (int w = 1; w < 4; w++) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int mb = 0; mb < H7_MB; mb++)
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int r = 0; r < 4; r++) red[((mb * 2 + h) * 4 + r) * 64 + lane] = acc[mb][h][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int mb = 0; mb < H7_MB; mb++)
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[mb][h][r] += red[((mb * 2 + h) * 4 + r) * 64 + lane];
        }
    }
    if (wave != 0) return;
    // D fragment: column = lane & 15 (channel of the half), row = 4 (lane >> 4) + r
    float* const dst = WGRAD_DST(a, blockIdx.x);
#pragma unroll
    for (int mb = 0; mb < H7_MB; mb++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int c = 16 * h + (lane & 15);
            if (c >= a.Ktot) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = mb * 16 + 4 * (lane >> 4) + r;
                if (m >= 147) continue;
                const int tap = m / 3, o = m - 3 * tap;
                if (o < a.Cout) atomicAdd(dst + ((long)tap * a.Cout_pad + o) * a.Ktot + c, acc[mb][h][r]);
            }
        }
}

}  // namespace

thread_local int g_last_wgrad_grouped = 0;

// 1 = handled: weight gradient of the 7x7 head (<= 3 output channels, <= 32 input channels of one dense tensor) on the split-bf16 matrix pipe (WgradArgs.precision == PREC_BF16X3)
int conv_head_wgrad_try(const WgradArgs& a0, hipStream_t st, bool dry) {
    if (a0.KS != 7 || a0.precision != PREC_BF16X3 || a0.nsrc != 1 || a0.src[0].bcast || a0.src[0].bn_scale) return 0;
    if (a0.Cout < 1 || a0.Cout > 3 || a0.dy_ld < 4 || (a0.dy_ld & 3) || (a0.dy_sn & 3) || a0.src[0].C > 32 || (a0.src[0].C & 3) || (a0.src[0].ld & 3) || (a0.src[0].sn & 3)) return 0;
    if (a0.Ktot != round_up(a0.src[0].C, CONV_BK) || a0.H < 1 || a0.W < 8) return 0;
    g_last_conv_kernel = CK_WGRAD_THIN;
    g_last_wgrad_grouped = 1;
    if (dry) return 1;
    WgradArgs a = a0;
    const int tx = cdiv(a.W, H7_TC), ty = cdiv(a.H, H7_TR);
    const long ntiles = (long)a.N * tx * ty;
    if (ntiles >= (1L << 30)) return 0;
    long g = ntiles < 256 ? ntiles : 256;                     // one persistent workgroup per CU (100 KB of LDS)
    if (a.det_slab) { g = wgrad_det_begin(a, g, st); if (g <= 0) return -1; }
    hipLaunchKernelGGL(k_wgrad_head7, dim3((unsigned)g), dim3(256), 0, st, a, tx, ty);
    if (a.det_slab) wgrad_det_end(a, g, st);
    return 1;
}

// 1 = handled: 1x1 weight gradient of one dense input tensor, up to 128 output / 64 input channels
int conv_stream_wgrad_try(const WgradArgs& a0, hipStream_t st, bool dry) {
    g_last_wgrad_grouped = 0;
    if (a0.KS != 1 || a0.nsrc != 1 || a0.src[0].bcast || a0.src[0].bn_scale || a0.precision == PREC_BF16X1) return 0;
    if (a0.Cout > 128 || a0.src[0].C > 64 || a0.Ktot != round_up(a0.src[0].C, CONV_BK)) return 0;
    if (cdiv(a0.Cout, 32) * cdiv(a0.src[0].C, 32) > 6) return 0;      // (64 -> 128: eight result blocks per wave -- the flush atomics of 256 workgroups cost more than the generic kernel's LDS staging: 164 vs 115 us in situ)
    const long HW = (long)a0.H * a0.W;
    if (HW * (a0.src[0].ld > a0.dy_ld ? a0.src[0].ld : a0.dy_ld) >= (1L << 31) || a0.N > 65535) return 0;
    g_last_conv_kernel = CK_WGRAD_SMALL;
    g_last_wgrad_grouped = 1;      // (understands WgradArgs.group_n: one launch for a time-batched call)
    if (dry) return 1;
    WgradArgs a = a0;
    const int nob = cdiv(a.Cout, 32), nkb = cdiv(a.src[0].C, 32);
    // ~512 workgroups (two per CU keep ~64 KB of loads in flight per CU), each at least 256 pixels; fewer when every wave flushes many blocks (atomics into a few KB)
    long want = 512 / (nob * nkb > 2 ? 2 : 1);
    long per_sample = (want + a.N - 1) / a.N;
    if (per_sample < 1) per_sample = 1;
    long ppx = (HW + per_sample - 1) / per_sample;
    if (ppx < 256) ppx = 256;
    ppx = (ppx + 63) / 64 * 64;
    long slabs = (HW + ppx - 1) / ppx;
    if (a.det_slab) {      // one zero-filled copy of the packed layout per workgroup, folded in a fixed order
        const long fit = wgrad_det_begin(a, slabs * a.N, st);
        if (fit <= 0) return -1;
        if (fit < slabs * a.N) {      // fewer, longer slabs so that every workgroup has its own copy
            slabs = fit / a.N;
            if (slabs < 1) { a.det_slab = nullptr; slabs = 1; }      // (more samples than copies: cannot happen with the 64 MB scratch and these layer sizes; atomics then)
            ppx = ((HW + slabs - 1) / slabs + 63) / 64 * 64;
            slabs = (HW + ppx - 1) / ppx;
        }
    }
    dim3 grid((unsigned)slabs, (unsigned)a.N);
#define SW_LAUNCH(NOB_, NKB_) hipLaunchKernelGGL((k_wgrad_1x1<NOB_, NKB_>), grid, dim3(256), 0, st, a, (int)ppx)
    if (nkb == 1) { if (nob == 1) SW_LAUNCH(1, 1); else if (nob == 2) SW_LAUNCH(2, 1); else if (nob == 3) SW_LAUNCH(3, 1); else SW_LAUNCH(4, 1); }
    else { if (nob == 1) SW_LAUNCH(1, 2); else if (nob == 2) SW_LAUNCH(2, 2); else if (nob == 3) SW_LAUNCH(3, 2); else SW_LAUNCH(4, 2); }
#undef SW_LAUNCH
    if (a.det_slab) wgrad_det_end(a, slabs * a.N, st);
    return 1;
}
