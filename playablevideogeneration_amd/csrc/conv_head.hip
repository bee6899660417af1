// Image heads of the rendering network on the 16-bit matrix pipe: nn.Conv2d(C -> 3, k = 3 | 7, padding = k / 2) + tanh of the FinalBlocks
// (model/layers/final_block.py:9-29, model/main_model/rendering_network.py:39-41,62-69), forward.
//
// Why a kernel of its own: 3 output channels.  The vector-ALU kernel these layers ran on (conv_thin.hip) reached 26 - 32 TFLOP/s on the 7x7 head (155 us per 8 frames of
// 256 x 256, profiles/r03_*_phases_and_layers.txt) although the layer only has to stream 67 MB -- it is bound by its 49 x 32 fp32 FMAs per output value.  Here the three
// output channels are the (padded) M = 16 rows of v_mfma_f32_16x16x32_f16, 16 pixels the N columns, 32 input channels the K of one instruction; operands are split as in
// conv_hx.hip (x = hi + lo in f16, three products per fp32 product, fp32 accumulation: fp32-class accuracy).  13 of the 16 rows are padding -- the instruction count per output
// value still drops from 1568 FMAs to 49 x 3 x (1/16 of a matrix instruction per pixel), and the kernel becomes LDS-read / staging bound instead of FMA bound.
//
// Layout.  A workgroup owns a TH x TW pixel tile.  Per 32-channel chunk the (TH + 2R) x (TW + 2R) halo of the fp32 input is staged once, split, into EIGHT LDS arrays
// [k-block 0..3][plane hi | lo][pixel][8 halves]: the B fragment of the instruction is (pixel = lane & 15, k-block = lane >> 4) -> one 16-byte read, and ds_read_b128 serves
// the lanes {0-3, 12-15, 20-27} together, i.e. pixels {0-3, 12-15} of k-block 0 and pixels {4-11} of k-block 1: with 16 bytes per pixel and the arrays a multiple of 256 B
// apart those 16 reads fall on the 16 distinct 16-byte slots of the bank row (a [pixel][32 channels] row-major image would put 7 of them on busy slots).  The weights of
// the chunk -- 3 rows x 32 channels per tap, converted from the packed fp32 layout by the workgroup itself, so no extra packed form is kept -- sit in LDS as
// [tap][row 0..3][hi 32 | lo 32] with row 3 all zero: lanes of the padding rows read row 3 (one broadcast address).  All taps of a chunk read the same staged image: no barrier
// inside the tap loop.
//
// Epilogue: rows 0..2 of the accumulator tile belong to lanes 0..15 (row = 4 (lane >> 4) + r, col = lane & 15): bias, tanh, three stores per pixel.
// Roofline: HBM (67 MB read + 6 MB written per 8 frames at 256 x 256 = 12 us); achieved is bounded by the staging (2.4 x halo over-fetch from L2) and the fragment reads.
#include "common.h"

namespace {
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int HD_KC = 32;                                   // channels per chunk = K of one instruction
#define HD_F16_MAX 65504.f

// NW waves per workgroup.  Tile choice = halo over-fetch vs occupancy: the 7x7 head on 8 x 16 tiles re-reads its input 2.4 x (14 x 22 halo pixels per 128 outputs) and was
// bound by exactly that traffic (138 us per 8 frames of 256 x 256); 16 x 32 tiles (22 x 38: 1.6 x) need 136 KB of LDS -- one workgroup per CU, so eight waves keep two per SIMD.
template <int KS, int TH, int TW, int NW>
__global__ __launch_bounds__(64 * NW) void k_conv_head(ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int NT = 64 * NW;
    constexpr int R = KS / 2, HH = TH + 2 * R, HW = TW + 2 * R, HPX = HH * HW, TAPS = KS * KS;
    constexpr int ARR = (HPX * 8 + 127) / 128 * 128;        // halves per array: a multiple of 256 B
    constexpr int WROW = 2 * HD_KC + 8;                      // weight row pitch in halves (hi 32 | lo 32 | pad)
    constexpr int NG = TH * TW / 16 / NW;                    // 16-pixel groups per wave
    constexpr int GPR = TW / 16;                             // groups per tile row
    static_assert(TW % 16 == 0 && (TH * TW) % (16 * NW) == 0, "tile");
    __shared__ __attribute__((aligned(16))) _Float16 Xs[8 * ARR];
    __shared__ __attribute__((aligned(16))) _Float16 Wl[TAPS * 4 * WROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tile = blockIdx.x;
    const int n = tile / (tiles_x * tiles_y);
    tile -= n * tiles_x * tiles_y;
    const int y0 = (tile / tiles_x) * TH, x0 = (tile % tiles_x) * TW;
    const ConvSrc s = a.src[0];
    const int nchunks = (s.C + HD_KC - 1) / HD_KC;

    f32x4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment addresses: A = weights (row = lane & 15 -> rows >= 3 read the zero row), B = pixels of this wave's groups
    const int kb = lane >> 4;
    const int arow = (lane & 15) < 3 ? (lane & 15) : 3;
    const int aoff = arow * WROW + kb * 8;
    int boff[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int gi = wave * NG + g;
        boff[g] = (2 * kb) * ARR + ((gi / GPR) * HW + (gi % GPR) * 16 + (lane & 15)) * 8;
    }
    const bool bn = s.bn_scale != nullptr;                   // lazily applied BatchNorm of the producer (ConvSrc.bn_*)
    const float slope = s.bn_act ? 0.2f : 1.f;
    unsigned amax = 0u;                                      // largest staged magnitude as a bit pattern (NaN > inf > finite): f16 range guard, see conv_hx.hip
    for (int chunk = 0; chunk < nchunks; chunk++) {
        if (chunk > 0) __syncthreads();                     // every wave is done with the previous chunk's image and weights
        // ---- weights of this chunk: packed fp32 [tap][Cout_pad][Ktot] -> split f16 (x 64: see HX_WSCALE), rows 0..2 + the zero row.  All loads of a thread are issued before the
        //      halo loads below and converted after them (round 5: a loop of load -> convert -> store cost a batch-1 frame's workgroups 7 dependent round trips) ----
        constexpr int NWI = TAPS * 4 * (HD_KC / 4), NWL = (NWI + NT - 1) / NT;
        float4 wreg[NWL];
#pragma unroll
        for (int j = 0; j < NWL; j++) {
            const int i = tid + NT * j;
            const int q_ = i % (HD_KC / 4), rr = (i / (HD_KC / 4)) & 3, tap = i / (HD_KC / 4) / 4;
            const int k = chunk * HD_KC + 4 * q_;
            const bool okw = i < NWI && rr < a.Cout && k < a.Ktot;
            wreg[j] = *reinterpret_cast<const float4*>(a.wp + (okw ? ((long)tap * a.Cout_pad + rr) * a.Ktot + k : 0L));
        }
        // ---- halo image of this chunk: thread = (pixel, channel quad q of 8); all loads of a thread are issued before the first conversion (clamped addresses, no branch
        //      around a load: a loop of load -> convert -> store leaves one load in flight per thread) ----
        constexpr int NL = (HPX * 8 + NT - 1) / NT, PP = NT / 8;      // float4 loads per thread, pixels per pass
        float4 ld[NL];
        const int q = tid & 7, c = chunk * HD_KC + 4 * q;
        const bool cok = c < s.C;
        const float* base = s.p + (long)n * s.sn + (cok ? c : 0);
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int p = (tid >> 3) + PP * i;
            const int hy = p / HW, hx = p - hy * HW;
            const int y = y0 - R + hy, x = x0 - R + hx;
            const bool ok = cok && p < HPX && y >= 0 && y < a.H && x >= 0 && x < a.W;
            ld[i] = *reinterpret_cast<const float4*>(base + (ok ? ((long)y * a.W + x) * s.ld : 0L));
        }
#pragma unroll
        for (int j = 0; j < NWL; j++) {
            const int i = tid + NT * j;
            if (NWL * NT > NWI && i >= NWI) continue;
            const int q_ = i % (HD_KC / 4), rr = (i / (HD_KC / 4)) & 3, tap = i / (HD_KC / 4) / 4;
            const bool okw = rr < a.Cout && chunk * HD_KC + 4 * q_ < a.Ktot;
            const float wv[4] = {okw ? wreg[j].x * HX_WSCALE : 0.f, okw ? wreg[j].y * HX_WSCALE : 0.f, okw ? wreg[j].z * HX_WSCALE : 0.f, okw ? wreg[j].w * HX_WSCALE : 0.f};
            h4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; e++) { hi[e] = (_Float16)wv[e]; lo[e] = (_Float16)(wv[e] - (float)hi[e]); }
            *reinterpret_cast<h4*>(&Wl[(tap * 4 + rr) * WROW + 4 * q_]) = hi;
            *reinterpret_cast<h4*>(&Wl[(tap * 4 + rr) * WROW + HD_KC + 4 * q_]) = lo;
        }
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bn && cok) { sc = *reinterpret_cast<const float4*>(s.bn_scale + c); sh = *reinterpret_cast<const float4*>(s.bn_shift + c); }
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int p = (tid >> 3) + PP * i;
            if (NL * PP > HPX && p >= HPX) continue;
            const int hy = p / HW, hx = p - hy * HW;
            const int y = y0 - R + hy, x = x0 - R + hx;
            const bool ok = cok && y >= 0 && y < a.H && x >= 0 && x < a.W;
            float v[4] = {ld[i].x, ld[i].y, ld[i].z, ld[i].w};
            if (bn) {      // act(x * scale + shift); the zero padding below applies to the NORMALISED tensor
                v[0] = fmaf(v[0], sc.x, sh.x); v[1] = fmaf(v[1], sc.y, sh.y); v[2] = fmaf(v[2], sc.z, sh.z); v[3] = fmaf(v[3], sc.w, sh.w);
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = v[e] > 0.f ? v[e] : slope * v[e];
            }
            h4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float t = (ok && c + e < s.C) ? v[e] : 0.f;
                amax = max(amax, __float_as_uint(t) & 0x7fffffffu);
                t = __builtin_amdgcn_fmed3f(t, -HD_F16_MAX, HD_F16_MAX);      // f16 range guard (ConvArgs.sat_flag)
                hi[e] = (_Float16)t; lo[e] = (_Float16)(t - (float)hi[e]);
            }
            _Float16* d = &Xs[(2 * (q >> 1)) * ARR + p * 8 + 4 * (q & 1)];
            *reinterpret_cast<h4*>(d) = hi;
            *reinterpret_cast<h4*>(d + ARR) = lo;
        }
        __syncthreads();
        // ---- 16 x 16 x 32 products: D[cout][pixel] += W[cout][k] X[k][pixel], three products per tap and group ----
#pragma unroll 7
        for (int tap = 0; tap < TAPS; tap++) {
            const int dy = tap / KS, dx = tap - dy * KS;
            const h8 wh = *reinterpret_cast<const h8*>(&Wl[tap * 4 * WROW + aoff]);
            const h8 wlo = *reinterpret_cast<const h8*>(&Wl[tap * 4 * WROW + aoff + HD_KC]);
            const int toff = (dy * HW + dx) * 8;
            h8 xh[NG], xl[NG];
#pragma unroll
            for (int g = 0; g < NG; g++) { xh[g] = *reinterpret_cast<const h8*>(&Xs[boff[g] + toff]); xl[g] = *reinterpret_cast<const h8*>(&Xs[boff[g] + toff + ARR]); }
            // (product-major order: consecutive matrix instructions write different accumulators -- three back-to-back products into one accumulator wait for each other)
#pragma unroll
            for (int g = 0; g < NG; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, xh[g], acc[g], 0, 0, 0);      // small terms first
#pragma unroll
            for (int g = 0; g < NG; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[g], acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[g], acc[g], 0, 0, 0);
        }
    }
    if (a.sat_flag != nullptr && amax > 0x477fe000u /* bits of 65504.f */) atomicOr(a.sat_flag, amax > 0x7f800000u ? 3u : 1u);      // bit 1: a NaN among them
    // ---- epilogue: lanes 0..15 hold rows (= output channels) 0..3 of their pixel column ----
    if (lane < 16) {
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int gi = wave * NG + g;
            const int y = y0 + gi / GPR, x = x0 + (gi % GPR) * 16 + lane;
            if (y >= a.H || x >= a.W) continue;
            float* o = a.out + (long)n * a.out_sn + ((long)y * a.W + x) * a.out_ld;
            for (int cc = 0; cc < a.Cout; cc++) {
                float v = acc[g][cc] * (1.0f / HX_WSCALE) + (a.bias ? a.bias[cc] : 0.f);
                if (a.act == 1) v = tanhf(v);
                o[cc] = v;
            }
        }
    }
}

// ---- 7x7 head forward with the taps on the N side (round 6; model/layers/final_block.py:9-29, rendering_network.py:41) ----
// k_conv_head<7> above puts the 3 output channels on the 16-row M side of v_mfma_f32_16x16x32_f16: 3 / 16 of every instruction is useful, 147 instructions per 16 pixels, and with
// one 136-KB workgroup per CU nothing overlaps its staging (93 us alone / 122 us in the step per 8 frames of 256 x 256, for a layer whose HBM time is 9 us -- on the closed-loop
// chain: D(t)'s frame feeds E(t + 1)).  Here one v_mfma_f32_32x32x16_f16 computes, for 32 consecutive HALO columns p of one image row,
//       P[(dx, co)][p] += sum_{c in 16 channels} W[dy][dx][co][c] * X[y + dy][p][c]              (M = (dx, co) = 21 of 32 rows, N = 32 halo columns, K = 16 channels of tap row dy)
// so an output row takes 7 tap rows x (C / 16) K-steps x 3 split products = 42 instructions per 26 output pixels (2.8 x fewer matrix cycles per pixel), and the result is the
// shift-add  out[y][x][co] = sum_dx P[(dx, co)][x + dx]  of the seven column partials, done through a wave-private LDS tile in the epilogue (the trick k_wgrad_head7 /
// k_head_dgrad7 use on their sides).  A workgroup owns 8 rows x 26 columns (halo 14 x 32 pixels: every halo column is a useful N column), a wave two output rows: the B fragment of
// a halo row is read once and used by both rows' tap rows.  The split weight fragments (7 tap rows x K-steps x hi / lo = 112 registers) stay in registers across the tiles of a
// persistent workgroup; LDS = 56 KB halo (the [k-block][plane][pixel][8 halves] arrays of k_conv_head, k-blocks skewed by 64 B so that the 8-byte staging stores of a wave cover all
// banks) + 21 KB shift-add tiles: two workgroups per CU, one stages while the other multiplies.  Workgroup -> tile order keeps vertically adjacent tiles (6 of 14 halo rows
// shared) on one XCD at the same time.  Same arithmetic class as k_conv_head: split f16 (weights x 64), three products, small terms first, fp32 accumulation.
template <int NS>      // K-steps of 16 channels: 1 (C <= 16: the reduced model's head) or 2
__global__ __launch_bounds__(256, 2) void k_conv_head7n(ConvArgs a, int tiles_x, int tiles_y, int chunk) {
    constexpr int TH = 8, TWO = 26, R = 3, HH = TH + 2 * R, HWD = 32, HPX = HH * HWD;
    constexpr int ARRB = HPX * 16;                 // bytes of one (k-block, plane) array: 16 B per pixel -- a multiple of 256
    constexpr int KSTR = 2 * ARRB + 64;            // k-block stride: hi array, lo array, 64-byte skew
    constexpr int SROW = 21 * 32;                  // floats of one output row's partial tile [m = (dx, co)][halo column]
    static_assert(ARRB % 256 == 0, "array pitch");
    __shared__ __attribute__((aligned(16))) unsigned char Xs[4 * KSTR];
    __shared__ __attribute__((aligned(16))) float Ss[4 * 2 * SROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const ConvSrc s = a.src[0];
    // ---- A fragments: W[m = 3 dx + co][k = 8 (lane >> 5) + j] of tap row dy, K-step ks: packed fp32 [tap][Cout_pad][Ktot] -> split f16 (x 64), once per workgroup ----
    h8 wh[7][NS], wl[7][NS];
    {
        const int m = lane & 31, kg = lane >> 5, dx = m / 3, co = m - 3 * dx;
        const bool mok = m < 21 && co < a.Cout;
        // (two batches of tap rows: all 28 float4 loads of a lane at once would hold 112 registers of fp32 weights beside the 112 of split fragments)
#pragma unroll
        for (int d0 = 0; d0 < 7; d0 += 4) {
            float4 w0[4][NS], w1[4][NS];
#pragma unroll
            for (int dd = 0; dd < 4; dd++)
#pragma unroll
                for (int ks = 0; ks < NS; ks++) {
                    const int dy = d0 + dd < 7 ? d0 + dd : 6;
                    const int k0 = 16 * ks + 8 * kg;
                    const bool ok = mok && k0 < a.Ktot;
                    const float* wp = a.wp + (ok ? ((long)(dy * 7 + dx) * a.Cout_pad + co) * a.Ktot + k0 : 0L);
                    w0[dd][ks] = *reinterpret_cast<const float4*>(wp);
                    w1[dd][ks] = *reinterpret_cast<const float4*>(wp + (ok ? 4 : 0));
                }
#pragma unroll
            for (int dd = 0; dd < 4; dd++)
#pragma unroll
                for (int ks = 0; ks < NS; ks++) {
                    if (d0 + dd >= 7) continue;
                    const bool ok = mok && 16 * ks + 8 * kg < a.Ktot;
                    const float wv[8] = {w0[dd][ks].x, w0[dd][ks].y, w0[dd][ks].z, w0[dd][ks].w, w1[dd][ks].x, w1[dd][ks].y, w1[dd][ks].z, w1[dd][ks].w};
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float v = ok ? wv[e] * HX_WSCALE : 0.f;
                        const _Float16 hi = (_Float16)v;
                        wh[d0 + dd][ks][e] = hi; wl[d0 + dd][ks][e] = (_Float16)(v - (float)hi);
                    }
                }
        }
    }
    const bool bn = s.bn_scale != nullptr;                   // lazily applied BatchNorm of the producer (ConvSrc.bn_*)
    const float slope = s.bn_act ? 0.2f : 1.f;
    const int q = tid & 7, hx = tid >> 3, c = 4 * q;         // staging: thread = (halo column, channel quad), one halo row per pass
    const bool cok = c < s.C && c < 16 * NS;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bn && cok) { sc = *reinterpret_cast<const float4*>(s.bn_scale + c); sh = *reinterpret_cast<const float4*>(s.bn_shift + c); }
    unsigned amax = 0u;                                      // f16 range guard (ConvArgs.sat_flag), see k_conv_head
    const int per_img = tiles_x * tiles_y;
    const long ntiles = (long)a.N * per_img;
    // workgroup -> tiles: XCD j (= blockIdx & 7: workgroups go round-robin over the 8 XCDs) owns the tiles [j chunk, (j + 1) chunk) in row-major order and its workgroups walk
    // them with a stride of gridDim / 8: at any time an XCD works on consecutive tile rows, whose halos overlap in its L2
    const int slots = gridDim.x >> 3;
    const long tend_ = (long)((blockIdx.x & 7) + 1) * chunk, tend = tend_ < ntiles ? tend_ : ntiles;
    for (long tile = (long)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3); tile < tend; tile += slots) {
        const int n = (int)(tile / per_img);
        const int rem = (int)(tile - (long)n * per_img);
        const int y0 = (rem / tiles_x) * TH, x0 = (rem % tiles_x) * TWO;
        // ---- halo: 14 rows x 32 columns x (4 NS) quads; all loads of a thread in flight before the first conversion ----
        float4 ld[HH];
        const int x = x0 - R + hx;
        const bool xok = cok && x >= 0 && x < a.W;
        const float* base = s.p + (long)n * s.sn + (cok ? c : 0);
#pragma unroll
        for (int i = 0; i < HH; i++) {
            const int y = y0 - R + i;
            const bool ok = xok && y >= 0 && y < a.H;
            ld[i] = *reinterpret_cast<const float4*>(base + (ok ? ((long)y * a.W + x) * s.ld : 0L));
        }
        if (q < 4 * NS) {
#pragma unroll
            for (int i = 0; i < HH; i++) {
                const int y = y0 - R + i;
                const bool ok = xok && y >= 0 && y < a.H;
                float v[4] = {ld[i].x, ld[i].y, ld[i].z, ld[i].w};
                if (bn) {      // act(x * scale + shift); the zero padding applies to the NORMALISED tensor
                    v[0] = fmaf(v[0], sc.x, sh.x); v[1] = fmaf(v[1], sc.y, sh.y); v[2] = fmaf(v[2], sc.z, sh.z); v[3] = fmaf(v[3], sc.w, sh.w);
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = v[e] > 0.f ? v[e] : slope * v[e];
                }
                h4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float t = (ok && c + e < s.C) ? v[e] : 0.f;
                    amax = max(amax, __float_as_uint(t) & 0x7fffffffu);
                    t = __builtin_amdgcn_fmed3f(t, -HD_F16_MAX, HD_F16_MAX);
                    hi[e] = (_Float16)t; lo[e] = (_Float16)(t - (float)hi[e]);
                }
                unsigned char* d = Xs + (q >> 1) * KSTR + (i * HWD + hx) * 16 + (q & 1) * 8;
                *reinterpret_cast<h4*>(d) = hi;
                *reinterpret_cast<h4*>(d + ARRB) = lo;
            }
        }
        __syncthreads();
        // ---- this wave's two output rows: halo rows 2 wave .. 2 wave + 7, each read once ----
        f32x16 accA, accB;
#pragma unroll
        for (int e = 0; e < 16; e++) { accA[e] = 0.f; accB[e] = 0.f; }
        const unsigned char* bb = Xs + (lane >> 5) * KSTR + ((2 * wave) * HWD + (lane & 31)) * 16;
#pragma unroll
        for (int rr = 0; rr < 8; rr++) {
#pragma unroll
            for (int ks = 0; ks < NS; ks++) {
                const h8 xh = *reinterpret_cast<const h8*>(bb + 2 * ks * KSTR + rr * HWD * 16);
                const h8 xl = *reinterpret_cast<const h8*>(bb + 2 * ks * KSTR + rr * HWD * 16 + ARRB);
                // (small terms first; consecutive instructions alternate between the two accumulators where both rows use this halo row)
                if (rr < 7) accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[rr][ks], xh, accA, 0, 0, 0);
                if (rr > 0) accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[rr - 1][ks], xh, accB, 0, 0, 0);
                if (rr < 7) accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[rr][ks], xl, accA, 0, 0, 0);
                if (rr > 0) accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[rr - 1][ks], xl, accB, 0, 0, 0);
                if (rr < 7) accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[rr][ks], xh, accA, 0, 0, 0);
                if (rr > 0) accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[rr - 1][ks], xh, accB, 0, 0, 0);
            }
        }
        // ---- shift-add of the seven column partials: D[m][p] (m = (e & 3) + 8 (e >> 2) + 4 (lane >> 5), p = lane & 31) -> LDS [row][m][p] -> out[x][co] = sum_dx D[3 dx + co][x + dx] ----
        float* S = Ss + wave * 2 * SROW;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int m = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            if (m < 21) { S[m * 32 + (lane & 31)] = accA[e]; S[SROW + m * 32 + (lane & 31)] = accB[e]; }
        }
        __syncthreads();      // (also: every wave is done with the halo image -- the next tile's staging may overwrite it)
        {
            const int xo = lane & 31, row = lane >> 5;
            const int y = y0 + 2 * wave + row, xg = x0 + xo;
            if (xo < TWO && y < a.H && xg < a.W) {
                const float* Sr = S + row * SROW + xo;
                float* o = a.out + (long)n * a.out_sn + ((long)y * a.W + xg) * a.out_ld;
                for (int cc = 0; cc < a.Cout; cc++) {
                    float v = 0.f;
#pragma unroll
                    for (int dx = 0; dx < 7; dx++) v += Sr[(3 * dx + cc) * 32 + dx];
                    v = v * (1.0f / HX_WSCALE) + (a.bias ? a.bias[cc] : 0.f);
                    if (a.act == 1) v = tanhf(v);
                    o[cc] = v;
                }
            }
        }
    }
    if (a.sat_flag != nullptr && amax > 0x477fe000u /* bits of 65504.f */) atomicOr(a.sat_flag, amax > 0x7f800000u ? 3u : 1u);
}

// ---- dgrad of the 7x7 head (3 -> C channels as a convolution of dY with the flipped weights; model/layers/final_block.py:9-29 backward) on v_mfma_f32_16x16x32_bf16 ----
// The 3-channel side is the REDUCTION here: K = 32 = one tap ROW of the 7x7 window = 4 tap pairs x 2 taps x (3 -> 4) channels (the eighth tap carries zero weights), so a
// 16-pixel x 16-channel block takes 7 instructions per product instead of 49 v_mfma_f32_16x16x4_f32 (k_conv_c4<7>: 85 us per 8 frames of 256 x 256, half of it matrix-pipe
// time).  B fragment = dY straight from the staged halo: lane (pixel p = lane & 15, pair j = lane >> 4) reads the 2 x 4 channels of the horizontally adjacent pixels
// p + 2j, p + 2j + 1 of row y + u - 3 -- 16 contiguous bytes of the [row][column][4 channels] bf16 image (implicit im2col, no expansion in LDS).  A fragments (weights of tap
// row u: [channel][pair, tap, ch]) are split once per workgroup and stay in registers (7 rows x NB blocks x hi / lo).  Operands split in bf16 as every gradient operand
// (conv_hx.hip); fp32 accumulation; the result is stored / accumulated as 4 consecutive channels per lane.  Persistent over 8 x 32-pixel tiles.
template <int NB>
__global__ __launch_bounds__(256) void k_head_dgrad7(ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int TH = 8, TW = 32, R = 3, HH = TH + 2 * R, HWD = TW + 2 * R + 2;      // 40 columns: 38 + the zero-weight eighth tap of the right-most pixel
    __shared__ __attribute__((aligned(16))) bf16x4 Xh[HH * HWD];
    __shared__ __attribute__((aligned(16))) bf16x4 Xl[HH * HWD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci = lane & 15, pj = lane >> 4;
    const ConvSrc s = a.src[0];
    // ---- weights: wp[tap][Cout_pad][Ktot] (dgrad-packed, taps already flipped) -> A fragments ----
    bf16x8 ah[7][NB], al[7][NB];
#pragma unroll
    for (int u = 0; u < 7; u++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            const int co = nb * 16 + ci;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int v = 2 * pj + (e >> 2), ch = e & 3;
                float w = 0.f;
                if (v < 7 && ch < s.C && co < a.Cout) w = a.wp[((long)(u * 7 + v) * a.Cout_pad + co) * a.Ktot + ch];
                const __bf16 h = (__bf16)w;
                ah[u][nb][e] = h; al[u][nb][e] = (__bf16)(w - (float)h);
            }
        }
    const long ntiles = (long)a.N * tiles_x * tiles_y;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = (int)(tile / (tiles_x * tiles_y));
        const int rem = (int)(tile - (long)n * tiles_x * tiles_y);
        const int y0 = (rem / tiles_x) * TH, x0 = (rem % tiles_x) * TW;
        const float* base = s.p + (long)n * s.sn;
        // ---- halo of dY: one float4 (3 channels + pad) per pixel -> split bf16 ----
        for (int p = tid; p < HH * HWD; p += 256) {
            const int hy = p / HWD, hx = p - hy * HWD;
            const int y = y0 - R + hy, x = x0 - R + hx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y >= 0 && y < a.H && x >= 0 && x < a.W) v = *reinterpret_cast<const float4*>(base + ((long)y * a.W + x) * s.ld);
            const float f[4] = {v.x, s.C > 1 ? v.y : 0.f, s.C > 2 ? v.z : 0.f, s.C > 3 ? v.w : 0.f};
            bf16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; e++) { hi[e] = (__bf16)f[e]; lo[e] = (__bf16)(f[e] - (float)hi[e]); }
            Xh[p] = hi; Xl[p] = lo;
        }
        __syncthreads();
#pragma unroll 1
        for (int g = 0; g < 4; g++) {                          // this wave's four 16-pixel groups: tile row 2 wave + (g >> 1), column half g & 1
            const int ry = 2 * wave + (g >> 1), cx = 16 * (g & 1);
            f32x4 acc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; nb++) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 7; u++) {
                const int o = (ry + u) * HWD + cx + ci + 2 * pj;
                union { bf16x4 q[2]; bf16x8 v; } bh, bl;
                bh.q[0] = Xh[o]; bh.q[1] = Xh[o + 1]; bl.q[0] = Xl[o]; bl.q[1] = Xl[o + 1];
#pragma unroll
                for (int nb = 0; nb < NB; nb++) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[u][nb], bh.v, acc[nb], 0, 0, 0);      // small terms first
#pragma unroll
                for (int nb = 0; nb < NB; nb++) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[u][nb], bl.v, acc[nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NB; nb++) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[u][nb], bh.v, acc[nb], 0, 0, 0);
            }
            const int y = y0 + ry, x = x0 + cx + ci;            // D: column = pixel lane & 15, rows = channels 4 (lane >> 4) .. + 3
            if (y < a.H && x < a.W) {
                float* o_ = a.out + (long)n * a.out_sn + ((long)y * a.W + x) * a.out_ld;
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    const int c = nb * 16 + 4 * pj;
                    if (c >= a.Cout) continue;
                    float v[4] = {acc[nb][0], acc[nb][1], acc[nb][2], acc[nb][3]};
                    if (c + 4 <= a.Cout) {
                        float4 r = make_float4(v[0], v[1], v[2], v[3]);
                        if (a.accumulate) { const float4 q = *reinterpret_cast<const float4*>(o_ + c); r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w; }
                        *reinterpret_cast<float4*>(o_ + c) = r;
                    } else {
                        for (int e = 0; e < 4 && c + e < a.Cout; e++) o_[c + e] = a.accumulate ? o_[c + e] + v[e] : v[e];
                    }
                }
            }
        }
        __syncthreads();
    }
}
}  // namespace

// 1 = handled: dgrad of the 7x7 FinalBlock head (a convolution of the 3-channel dY with the dgrad-packed weights) on the split-bf16 matrix pipe
// (precision == PREC_BF16X3 without split weights marks the layer as eligible: the context runs exact fp32 otherwise)
int conv_head_dgrad_try(const ConvArgs& a, hipStream_t st) {
    if (a.precision != PREC_BF16X3 || a.wq || a.KS != 7 || a.nsrc != 1 || a.src[0].bcast || a.src[0].bn_scale || a.src[0].C > 4 || a.src[0].ld != 4 || (a.src[0].sn & 3)) return 0;
    if (a.bias || a.res || a.mask || a.pool_out || a.skip_out || a.stats || a.act != 0 || a.Cout < 8 || a.Cout > 32 || (a.out_ld & 3) || (a.out_sn & 3)) return 0;
    const int nb = cdiv(a.Cout, 16);
    if (a.Cout_pad < nb * 16) return 0;
    const int tx = cdiv(a.W, 32), ty = cdiv(a.H, 8);
    const long ntiles = (long)a.N * tx * ty;
    const unsigned grid = (unsigned)(ntiles < 1024 ? ntiles : 1024);
    if (nb == 2) hipLaunchKernelGGL((k_head_dgrad7<2>), dim3(grid), dim3(256), 0, st, a, tx, ty);
    else hipLaunchKernelGGL((k_head_dgrad7<1>), dim3(grid), dim3(256), 0, st, a, tx, ty);
    g_last_conv_kernel = CK_THIN_IN;
    return 1;
}

// 1 = handled: forward of a FinalBlock head on the split-f16 matrix pipe (precision == PREC_F16X3 marks the layer as eligible: the context runs exact fp32 otherwise)
int conv_head_fwd_try(const ConvArgs& a, hipStream_t st) {
    if (a.precision != PREC_F16X3 || a.wq || a.nsrc != 1 || a.src[0].bcast || a.Cout > 3 || a.Cout < 1 || (a.KS != 3 && a.KS != 7)) return 0;
    if (a.accumulate || a.res || a.mask || a.pool_out || a.skip_out || a.stats || (a.act != 0 && a.act != 1)) return 0;
    if (a.src[0].C < 16 || (a.src[0].ld & 3) || (a.src[0].sn & 3) || (a.src[0].C & 3) || (a.Ktot & 3)) return 0;
    if (a.KS == 7 && a.src[0].C <= 32 && a.Cout_pad >= a.Cout && (a.Ktot & 15) == 0) {      // taps on the N side (round 6)
        const int tx = cdiv(a.W, 26), ty = cdiv(a.H, 8);
        const long ntiles = (long)a.N * tx * ty;
        const int chunk = (int)((ntiles + 7) / 8);
        const int slots = chunk < 64 ? chunk : 64;                    // <= 512 persistent workgroups, two per CU
        if (a.src[0].C <= 16) hipLaunchKernelGGL((k_conv_head7n<1>), dim3(8 * slots), dim3(256), 0, st, a, tx, ty, chunk);
        else hipLaunchKernelGGL((k_conv_head7n<2>), dim3(8 * slots), dim3(256), 0, st, a, tx, ty, chunk);
    } else if (a.KS == 7) {
        const int tx = cdiv(a.W, 32), ty = cdiv(a.H, 16);
        if ((long)a.N * tx * ty < 200) {      // a batch-1 frame: 128 workgroups of 16 x 32 pixels leave half the chip idle -- 8 x 32 tiles (2.0 x halo instead of 1.6 x, but twice the workgroups and a
                                              // chain half as long per workgroup: 22.4 -> 15 us per frame at 256 x 256)
            const int ty8 = cdiv(a.H, 8);
            hipLaunchKernelGGL((k_conv_head<7, 8, 32, 4>), dim3((unsigned)((long)a.N * tx * ty8)), dim3(256), 0, st, a, tx, ty8);
        } else
        hipLaunchKernelGGL((k_conv_head<7, 16, 32, 8>), dim3((unsigned)((long)a.N * tx * ty)), dim3(512), 0, st, a, tx, ty);
    } else {
        const int tx = cdiv(a.W, 32), ty = cdiv(a.H, 8);
        hipLaunchKernelGGL((k_conv_head<3, 8, 32, 4>), dim3((unsigned)((long)a.N * tx * ty)), dim3(256), 0, st, a, tx, ty);
    }
    g_last_conv_kernel = CK_THIN_OUT;
    return 1;
}
