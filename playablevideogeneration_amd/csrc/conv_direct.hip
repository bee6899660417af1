// Latency-bound 3x3 convolutions (batch-1 roll-out: model/main_model/model.py:570-607 -- E's residual blocks on a 32x32 map, R's small side branches, the 32-channel stages):
// one launch of a few hundred workgroups instead of the tile kernel's split-K launch + slab reduce.
//
// Why: such a launch is ~75 MFLOP; on k_conv_hx it is a chain of prologue -> 9..18 (tap, chunk) steps with a barrier each -> slab store, followed by k_split_reduce: 14.7 + 5 us
// per layer whatever the arithmetic (profiles/r04_rollout_kernels.txt: 18 + 17 of a frame's 66 launches).  Here a workgroup owns ONE 16-pixel x 16-channel output tile and its
// four waves split the K steps; every operand fragment goes from global memory / L2 straight into registers (no LDS staging, no barrier inside the loop, all loads of a wave
// in flight at once), the four partial tiles meet in LDS in a fixed order, and the epilogue (bias, residual, LeakyReLU / ReLU) is applied in place -- no slabs, no reduce launch.
// Arithmetic = conv_hx.hip's: v_mfma_f32_16x16x32_f16 on split operands (x = hi + lo, three products per fp32 product), weights from the same packed split-f16 buffer
// ([tap][chunk][Cout_pad][hi 32 | lo 32]: a row's 16-byte pieces ARE the A fragments), activations converted in registers.
//
// It only pays where the activations are small enough to be re-read by every 16-channel block (no reuse through LDS): the launcher takes launches of
// <= DIRECT_MAX_WORK workgroup-steps and leaves everything else to k_conv_hx.
#include "common.h"

namespace {
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr int DK = HX_KC;                                    // channels per K step (one instruction)
constexpr int DIRECT_MAX_CHUNKS = 64;
#define DR_F16_MAX 65504.f

struct ChunkRef { const float* base; int pl; int C; int c0; };      // source of one 32-channel chunk: sample base pointer, pixel pitch (0: broadcast vector), channels, first channel

__device__ __forceinline__ void split8(const float (&v)[8], h8& hi, h8& lo, unsigned& amax) {
#pragma unroll
    for (int e = 0; e < 8; e++) {
        amax = max(amax, __float_as_uint(v[e]) & 0x7fffffffu);
        const float t = __builtin_amdgcn_fmed3f(v[e], -DR_F16_MAX, DR_F16_MAX);      // f16 range guard (ConvArgs.sat_flag, see conv_hx.hip)
        hi[e] = (_Float16)t; lo[e] = (_Float16)(t - (float)hi[e]);
    }
}

// grid: x = 16-pixel groups (sample, row, column group), y = 16-channel output blocks.  NW waves split the K steps; U steps of a wave are in flight together.
template <int NW, int U>
__global__ __launch_bounds__(64 * NW) void k_conv_direct(ConvArgs a, int groups_x) {
    __shared__ ChunkRef tab[DIRECT_MAX_CHUNKS];
    __shared__ __attribute__((aligned(16))) float red[NW * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ConvArgs.avgpool: the 16 pixels are a 2 x 8 block (rows 2y, 2y + 1) whose 2x2 windows are averaged by the epilogue -> 4 pixels of row y of the (H/2, W/2) output
    const bool AP = a.avgpool != 0;
    const int rows = AP ? a.H >> 1 : a.H;
    int g = blockIdx.x;
    const int n = g / (groups_x * rows);
    g -= n * groups_x * rows;
    const int y = g / groups_x, x0 = (g - y * groups_x) * (AP ? 8 : 16);
    const int co0 = blockIdx.y * 16;
    const int nchunks = a.Kq / DK, nsteps = nchunks * 9;
    // chunk -> source table (segments are padded to whole chunks in the packed weights)
    if (tid < nchunks) {      // (nchunks <= DIRECT_MAX_CHUNKS <= the workgroup size)
        int s = 0, c = tid * DK;
        while (s + 1 < a.nsrc && c >= (a.src[s].C + DK - 1) / DK * DK) { c -= (a.src[s].C + DK - 1) / DK * DK; s++; }
        ChunkRef r; r.base = a.src[s].p + (long)n * a.src[s].sn; r.pl = a.src[s].bcast ? 0 : a.src[s].ld; r.C = a.src[s].C; r.c0 = c;
        tab[tid] = r;
    }
    __syncthreads();
    const int px = lane & 15, kg = lane >> 4;                 // B fragment: pixel column, k-group (8 channels); A fragment: output channel row px, same k-group
    const int py = AP ? 2 * y + (px >> 3) : y, pxx = x0 + (AP ? (px & 7) : px);      // this lane's input-resolution pixel
    const _Float16* wq = reinterpret_cast<const _Float16*>(a.wq) + hx_wq_piece<2>(co0 + px, kg) * 8;      // (fragment-major piece order of the packed weights, common.h)
    const long wstep = (long)a.Cout_pad * (2 * DK);           // halves between consecutive (tap, chunk) tiles
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned amax = 0u;
    for (int s0 = wave; s0 < nsteps; s0 += NW * U) {
        h8 wh[U], wl[U];
        float4 xa[U], xb[U];
        bool ok[U]; int cc[U], CC[U];
#pragma unroll
        for (int u = 0; u < U; u++) {                         // loads only (clamped addresses)
            int s = s0 + NW * u; s = s < nsteps ? s : nsteps - 1;
            const int chunk = s / 9, tap = s - 9 * chunk;
            const ChunkRef r = tab[chunk];
            const int yy = py + tap / 3 - 1, xx = pxx + tap % 3 - 1;
            const int c = r.c0 + kg * 8;
            ok[u] = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W; cc[u] = c; CC[u] = r.C;
            const float* p = r.base + (ok[u] ? ((long)yy * a.W + xx) * r.pl : 0L);
            xa[u] = *reinterpret_cast<const float4*>(p + (c < r.C ? c : 0));
            xb[u] = *reinterpret_cast<const float4*>(p + (c + 4 < r.C ? c + 4 : 0));
            const _Float16* w = wq + ((long)tap * nchunks + chunk) * wstep;
            wh[u] = *reinterpret_cast<const h8*>(w);
            wl[u] = *reinterpret_cast<const h8*>(w + 64 * 8);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (s0 + NW * u >= nsteps) break;                 // (wave-uniform)
            const float v4[8] = {xa[u].x, xa[u].y, xa[u].z, xa[u].w, xb[u].x, xb[u].y, xb[u].z, xb[u].w};
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = (ok[u] && cc[u] + e < CC[u]) ? v4[e] : 0.f;
            h8 xh, xl;
            split8(v, xh, xl, amax);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[u], xh, acc, 0, 0, 0);      // small terms first
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[u], xl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[u], xh, acc, 0, 0, 0);
        }
    }
    if (a.sat_flag != nullptr && amax > 0x477fe000u /* bits of 65504.f */) atomicOr(a.sat_flag, amax > 0x7f800000u ? 3u : 1u);      // bit 1: a NaN among them
    // ---- the waves' partial tiles: fixed order (bit-reproducible), then the epilogue on wave 0: lane holds channels co0 + 4 (lane >> 4) .. + 3 of pixel lane & 15 ----
    *reinterpret_cast<f32x4*>(&red[(wave * 64 + lane) * 4]) = acc;
    __syncthreads();
    if (wave != 0) return;
    float v[4];
    int xx = x0 + px, OW = a.W;
    if (AP) {      // lanes px < 4: pooled pixel x0 / 2 + px = the average of tile pixels {2 px, 2 px + 1} x {row 0, row 1}; each summed over the waves in order first
        if (px >= 4) return;
        const int l0 = (lane & 48) + 2 * px;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float q[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int l = l0 + (j & 1) + 8 * (j >> 1);
                float t = red[l * 4 + r];
#pragma unroll
                for (int w = 1; w < NW; w++) t += red[(w * 64 + l) * 4 + r];
                q[j] = t;
            }
            v[r] = 0.25f * ((q[0] + q[1]) + (q[2] + q[3]));
        }
        xx = (x0 >> 1) + px; OW = a.W >> 1;
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float t = red[lane * 4 + r];
#pragma unroll
            for (int w = 1; w < NW; w++) t += red[(w * 64 + lane) * 4 + r];
            v[r] = t;
        }
    }
    const int co = co0 + 4 * kg;
    if (xx >= OW || co >= a.Cout) return;
    const long pix = (long)y * OW + xx;
    float* o = a.out + (long)n * a.out_sn + pix * a.out_ld + co;
    const float* rp = a.res ? a.res + (long)n * a.res_sn + pix * a.res_ld + co : nullptr;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (co + r >= a.Cout) break;
        float t = v[r] * a.out_scale + (a.bias ? a.bias[co + r] : 0.f);
        if (rp) t += rp[r];
        if (a.act == 2) t = fmaxf(t, 0.f);
        else if (a.act == 3) t = t > 0.f ? t : 0.2f * t;
        v[r] = t;
    }
    if (co + 4 <= a.Cout) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    else for (int r = 0; r < 4 && co + r < a.Cout; r++) o[r] = v[r];
}

// 1x1 convolutions of a batch-1 frame (the residual blocks' down-sampling paths: 16 -> 32 @128x128, 32 -> 64 @64x64, 64 -> 65 @32x32; ~2 - 8 MFLOP): the implicit-GEMM kernel
// needs 8 - 9 us for them (+ 5 us for the avg-pool behind it).  Four lanes = one output pixel x four output channels, each lane a quarter of the input channels (exact fp32
// FMAs in channel order, the four partial sums meet through two DPP exchanges in a fixed order); with ConvArgs.avgpool the lanes average the 2x2 input window first (a 1x1
// convolution commutes with the pooling).  Weights: the packed fp32 rows wp[o][k] (BatchNorm scale folded in).
__global__ __launch_bounds__(256) void k_conv1x1_lat(ConvArgs a, int C4o, int kspan) {
    const bool AP = a.avgpool != 0;
    const int OH = AP ? a.H >> 1 : a.H, OW = AP ? a.W >> 1 : a.W;
    const long total = (long)a.N * OH * OW * C4o;
    const long tidx = blockIdx.x * 256L + threadIdx.x;
    const int kq = (int)(tidx & 3);
    const bool valid = (tidx >> 2) < total;
    const long idx = valid ? (tidx >> 2) : 0;       // (every lane of a quad takes part in the exchanges)
    const int q = (int)(idx % C4o);
    long pix = idx / C4o;
    const int n = (int)(pix / ((long)OH * OW));
    pix -= (long)n * OH * OW;
    const int y = (int)(pix / OW), x = (int)(pix - (long)y * OW);
    const ConvSrc s = a.src[0];
    const float* xp = s.p + (long)n * s.sn + ((long)(AP ? 2 * y : y) * a.W + (AP ? 2 * x : x)) * s.ld;
    const long dxo = s.ld, dyo = (long)a.W * s.ld;
    const int co = 4 * q;
    const float* w0 = a.wp + (long)co * a.Ktot;      // rows co .. co + 3 exist: Cout_pad is a multiple of 32
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int k1 = (kq + 1) * kspan < s.C ? (kq + 1) * kspan : s.C;
#pragma unroll 4
    for (int k = kq * kspan; k < k1; k += 4) {
        float4 v = *reinterpret_cast<const float4*>(xp + k);
        if (AP) {
            const float4 b = *reinterpret_cast<const float4*>(xp + dxo + k), c = *reinterpret_cast<const float4*>(xp + dyo + k), d = *reinterpret_cast<const float4*>(xp + dyo + dxo + k);
            v.x = 0.25f * ((v.x + b.x) + (c.x + d.x)); v.y = 0.25f * ((v.y + b.y) + (c.y + d.y)); v.z = 0.25f * ((v.z + b.z) + (c.z + d.z)); v.w = 0.25f * ((v.w + b.w) + (c.w + d.w));
        }
        if (k + 4 > s.C) { if (k + 1 >= s.C) v.y = 0.f; if (k + 2 >= s.C) v.z = 0.f; v.w = 0.f; }      // (channels beyond C are not ours to read as data: zero them)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float4 w = *reinterpret_cast<const float4*>(w0 + (long)r * a.Ktot + k);
            acc[r] = fmaf(v.x, w.x, acc[r]); acc[r] = fmaf(v.y, w.y, acc[r]); acc[r] = fmaf(v.z, w.z, acc[r]); acc[r] = fmaf(v.w, w.w, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {      // (k-quarter 0 + 1) + (2 + 3): the same order in every lane
        acc[r] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc[r]), 0xB1, 0xF, 0xF, true));      // quad_perm [1, 0, 3, 2]
        acc[r] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc[r]), 0x4E, 0xF, 0xF, true));      // quad_perm [2, 3, 0, 1]
    }
    if (!valid || kq != 0) return;
    const long opix = (long)y * OW + x;
    float* o = a.out + (long)n * a.out_sn + opix * a.out_ld + co;
    const float* rp = a.res ? a.res + (long)n * a.res_sn + opix * a.res_ld + co : nullptr;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (co + r >= a.Cout) break;
        float t = acc[r] + (a.bias ? a.bias[co + r] : 0.f);
        if (rp) t += rp[r];
        if (a.act == 3) t = t > 0.f ? t : 0.2f * t;
        acc[r] = t;
    }
    if (co + 4 <= a.Cout) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    else for (int r = 0; r < 4 && co + r < a.Cout; r++) o[r] = acc[r];
}
}  // namespace

// 1 = handled: small exact-fp32 1x1 launches of inference passes (ConvArgs.direct_ok); `dry`: shape test only (out need not be set)
int conv1x1_lat_try(const ConvArgs& a, hipStream_t st, bool dry) {
    if (a.KS != 1 || a.nsrc != 1 || !a.direct_ok || a.src[0].bcast || a.src[0].bn_scale || (a.act != 0 && a.act != 3) || a.accumulate || a.mask || a.pool_out || a.skip_out || a.stats) return 0;
    if (a.wq || a.precision != PREC_FP32 || a.splitk > 1 || a.in_s16 || a.out_s16 || !a.wp) return 0;
    if ((a.src[0].ld & 3) || (a.src[0].sn & 3) || ((uintptr_t)a.src[0].p & 15) || (a.Ktot & 3) || a.src[0].C > 128 || a.src[0].C > a.Ktot) return 0;
    if (a.avgpool && ((a.H | a.W) & 1)) return 0;
    const int C4o = cdiv(a.Cout, 4);
    if (4 * C4o > a.Cout_pad) return 0;
    const long total = (long)a.N * (a.avgpool ? a.H / 2 : a.H) * (a.avgpool ? a.W / 2 : a.W) * C4o;
    if (total > 65536) return 0;
    if (dry) return 1;
    if ((a.out_ld & 3) || (a.out_sn & 3) || ((uintptr_t)a.out & 15) || (a.res && ((a.res_ld & 3) || (a.res_sn & 3) || ((uintptr_t)a.res & 15)))) return 0;
    const int kspan = (cdiv(a.src[0].C, 4) + 3) / 4 * 4;      // input channels per lane of a quad (a multiple of 4)
    hipLaunchKernelGGL(k_conv1x1_lat, dim3((unsigned)cdiv(4 * total, 256)), dim3(256), 0, st, a, C4o, kspan);
    return 1;
}

// workgroup-steps up to which re-reading the activations per 16-channel block (no LDS reuse) is cheaper than the tile kernel's launch + slab reduce
#define DIRECT_MAX_WORK 24576

// 1 = handled.  Called by conv_hx_try for assigning split-f16 launches it would otherwise split over K (a.Kq, a.out_scale, a.Cout_pad set by the caller).
int conv_direct_try(const ConvArgs& a, hipStream_t st, bool dry) {
    if (a.KS != 3 || !a.wq || a.precision != PREC_F16X3 || a.accumulate || a.mask || a.pool_out || a.skip_out || a.stats || a.seed_ref) return 0;
    if (a.act != 0 && a.act != 2 && a.act != 3) return 0;
    if ((a.out_ld & 3) || (a.out_sn & 3) || (a.res && ((a.res_ld & 3) || (a.res_sn & 3)))) return 0;
    for (int s = 0; s < a.nsrc; s++) if (a.src[s].bn_scale || (a.src[s].ld & 3) || (a.src[s].sn & 3) || ((uintptr_t)a.src[s].p & 15)) return 0;
    if (((uintptr_t)a.out | (uintptr_t)a.res) & 15) return 0;      // (float4 accesses: a channel-offset view with c0 % 4 != 0 stays on the tile kernel)
    const int nchunks = a.Kq / DK;
    if (nchunks < 1 || nchunks > DIRECT_MAX_CHUNKS) return 0;
    if (a.avgpool && ((a.H | a.W) & 1)) return 0;
    const int gx = a.avgpool ? cdiv(a.W, 8) : cdiv(a.W, 16);
    const long groups = (long)a.N * (a.avgpool ? a.H / 2 : a.H) * gx, cob = cdiv(a.Cout, 16);
    if (groups * cob * nchunks * 9 > DIRECT_MAX_WORK) return 0;
    if ((long)cob * 16 > a.Cout_pad) return 0;
    if (dry) return 1;
    dim3 grid((unsigned)groups, (unsigned)cob);
    // (eight waves on the long reductions of R's 16x16 side branch -- 81 steps, 128 workgroups -- measured SLOWER: 13.7 us against ~8, roll-out 2068 -> 2014 frames/s)
    if (nchunks * 9 >= 36) hipLaunchKernelGGL((k_conv_direct<4, 3>), grid, dim3(256), 0, st, a, gx);
    else hipLaunchKernelGGL((k_conv_direct<4, 2>), grid, dim3(256), 0, st, a, gx);
    return 1;
}
