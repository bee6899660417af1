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

}  // namespace

thread_local int g_last_wgrad_grouped = 0;

// 1 = handled: 1x1 weight gradient of one dense input tensor, up to 128 output / 64 input channels
int conv_stream_wgrad_try(const WgradArgs& a0, hipStream_t st, bool dry) {
    g_last_wgrad_grouped = 0;
    if (a0.KS != 1 || a0.nsrc != 1 || a0.src[0].bcast || a0.src[0].bn_scale || a0.precision == PREC_BF16X1) return 0;
    if (a0.Cout > 128 || a0.src[0].C > 64 || a0.Ktot != round_up(a0.src[0].C, CONV_BK)) return 0;
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
