// Weight gradients of the wide 3x3 layers on the CDNA4 16-bit matrix pipe with split bf16 operands (see conv_hx.hip for the arithmetic; SURVEY.md section 8a rows K1-K3, K5:
// the backward of nn.Conv2d in model/layers/*.py).  Own translation unit: the forward / dgrad tile kernels of conv_hx.hip compile for minutes.
#include "conv_hx_common.h"
#include <cstdlib>

namespace {

// ------------------------------------------------------------------------------------------------------------------------------------
// Weight gradient on the 16-bit matrix pipe:  dW[tap][o][k] += sum_pixels dY[p][o] * X[p + tap][k]   (3x3, the layers conv_hx runs forward).
//
// GEMM-M = output channels, GEMM-N = input channels, reduction = PIXELS -- so both MFMA operands need 8 consecutive pixels per lane for a
// fixed channel, the transpose of the NHWC layout.  gfx950's ds_read_b64_tr_b16 does that transpose on the way out of LDS: the tiles are
// staged as [pixel][channel] (a straight copy of the fp32 NHWC rows, split into bf16 hi | lo on the way) and every 16-lane group reads a
// [4 pixels][16 channels] block, each lane receiving its channel's four pixels (lane -> operand map measured on the MI355X, tools/probes/
// tr_probe.hip: lane s of a group supplies the address of row s >> 2, columns 4 (s & 3) .. +3; lane j receives column j of rows 0..3).
// Tap shifts are whole-pixel address offsets into the halo tile, so the nine taps share one staged X tile (as in k_conv_wgrad_tile).
// A workgroup owns a 64(o) x 64(k) weight tile for all nine taps -- four waves as 2 x 2, nine 32x32 accumulators each -- walks 4 x 16-pixel
// spatial tiles persistently with a register prefetch of the next tile, and flushes once with fp32 atomics.  Operands are split bf16
// (gradients have no lower magnitude bound), three products per fp32 product.  Row pitch 320 B: the four pixel rows x two channel halves of
// one tr-read cycle fall on disjoint bank octets.
// ------------------------------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int WG_TH = 4, WG_TW = 16, WG_KC = 64, WG_OC = 64;
constexpr int WG_HW = WG_TW + 2, WG_HH = WG_TH + 2;
constexpr int WG_PITCH = 2 * 64 + 32;                       // 16-bit elements per pixel row: hi 64 | lo 64 | pad

__device__ __forceinline__ SegRefH find_seg16(const ConvSrc* src, int nsrc, int k) {
    int s = 0;
    while (s + 1 < nsrc && k >= src[s].Cpad) { k -= src[s].Cpad; s++; }
    SegRefH r; r.p = src[s].p; r.sn = src[s].sn; r.ld = src[s].ld; r.C = src[s].C; r.bcast = src[s].bcast; r.c0 = k; r.idx = s;
    r.bn_scale = src[s].bn_scale; r.bn_shift = src[s].bn_shift; r.bn_act = src[s].bn_act; r.bn_gn = src[s].bn_gn; r.bn_gs = src[s].bn_gs;
    return r;
}

template <typename T>
__device__ __forceinline__ typename Vec<T>::v8 tr_frag(const T* base, int off0, int off1) {
    typedef typename Vec<T>::v8 v8;
    union { s16x4 s[2]; v8 v; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off0));
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off1));
    return u.v;
}

// OCC: workgroups per CU the register allocation is bounded for (2: 256 registers per lane -- 144 accumulators + everything else, ~2 spilled -- so that the
// LDS-store / barrier phase of one workgroup runs under the MFMA phase of the other; 1: the unconstrained allocation, one workgroup per CU)
// RSPLIT (layers of <= 32 output channels, round 4): the two wave rows split the tile's PIXEL rows instead of the output channels -- with the 64-channel block half empty
// the wm = 1 waves multiplied zeros (D's last UpBlock conv 64 -> 32 @256x256, the largest weight gradient of the step); their partial sums meet in the final atomics.
// YS16 (round 6): dY arrives PRE-SPLIT (WgradArgs.dy_s16: [hi 32 | lo 32] bf16 halves per 32-channel chunk, written once by the gradient's point-wise producer).  The 64-channel
// row segment of a pixel is the same 256 bytes at the same address as in the fp32 tensor: thread (pixel column p0, piece q) copies 16 bytes -- eight hi or eight lo halves of one
// chunk -- to their place in the [hi 64 | lo 64] LDS row; no conversion (was ~20 VALU per float4, repeated by every input-channel block of the layer), same operands bit for bit.
template <typename T, int OCC, bool RSPLIT, bool YS16>
__global__ __launch_bounds__(256, OCC) void k_wgrad_hx(WgradArgs a, int tiles_x, int tiles_y) {
    typedef typename Vec<T>::v8 v8;
    typedef typename Vec<T>::v4 v4;
    __shared__ __attribute__((aligned(16))) T Xh[WG_HH * WG_HW * WG_PITCH];
    __shared__ __attribute__((aligned(16))) T Yt[WG_TH * WG_TW * WG_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int k0 = blockIdx.x * WG_KC, o0 = blockIdx.y * WG_OC;
    const int ntiles = a.N * tiles_x * tiles_y;               // (< 2^31: the launcher checks)

    // loader roles (round 4: one halo ROW per pass -- row validity and row offset are scalars, the column part of the address is computed once per tile; the former
    // pixel = (tid >> 4) + 16 i map cost ~1000 instructions per tile in divisions, 64-bit offset products and branches around every load, against 108 MFMAs per wave):
    // float4 column q (0..15) fixed per thread; p0 = tid >> 4.  X halo: passes 0..5 -> pixel (row i, column p0), pass 6 -> the two right-most columns (16, 17) of row p0 >> 1
    // (p0 < 12).  dY: pass i -> pixel (row i, column p0).
    const int q = tid & 15, p0 = tid >> 4;
    const int kx = k0 + (q >> 2) * CONV_BK;
    const bool kok = kx < a.Ktot;
    const SegRefH sg = find_seg16(a.src, a.nsrc, kok ? kx : 0);
    const int cx = sg.c0 + (q & 3) * 4;                       // channel inside the segment
    const int yc = o0 + q * 4;
    const bool cok = kok && cx < sg.C, yok = yc < a.Cout;
    const bool xm1 = cx + 1 < sg.C, xm2 = cx + 2 < sg.C, xm3 = cx + 3 < sg.C;
    const bool ym1 = yc + 1 < a.Cout, ym2 = yc + 2 < a.Cout, ym3 = yc + 3 < a.Cout;
    const int hy6 = p0 >> 1, hx6 = WG_TW + (p0 & 1);          // pass 6
    const int yl16 = p0 * WG_PITCH + ((q >> 2) & 1) * 64 + (q >> 3) * 32 + (q & 3) * 8;      // YS16: piece q = (chunk q >> 3, hi | lo (q >> 2) & 1, eight halves q & 3) of the pixel's 256 bytes
    const int xl = p0 * WG_PITCH + 4 * q, xl6 = (hy6 * WG_HW + hx6) * WG_PITCH + 4 * q;      // LDS element offsets (pass i: + i * WG_HW * WG_PITCH; dY: + i * WG_TW * WG_PITCH)
    static_assert(WG_TW == 16 && WG_HW == 18 && 2 * WG_HH <= 16, "loader passes: 16 columns per row pass, the two right-most columns of all rows in one more");
    // tile-invariant address parts (32-bit element offsets inside one sample: H * W * ld < 2^31)
    const int xpl = sg.bcast ? 0 : sg.ld, xrow = a.W * xpl, yrow = a.W * a.dy_ld;
    const long sgs = a.group_n > 0 ? a.src_gs[sg.idx] : 0L, sbgs = a.group_n > 0 ? a.src_bn_gs[sg.idx] : 0L;
    const float inv_bn_gn = sg.bn_gn > 0 ? 1.f / (float)sg.bn_gn : 0.f;      // (n / bn_gn for n < 2^20: floor((n + 0.5) * inv))
    const bool xbn = sg.bn_scale != nullptr;
    const float xsl = sg.bn_act ? 0.2f : 1.f;
    float4 rx[WG_HH + 1], ry[WG_TH];
    float4 rxs, rxh;                                          // lazily applied BatchNorm of the X source (ConvSrc.bn_*): scale / shift of this thread's four channels, tile in flight
    long bnoff = 0;                                           // ... their offset in the (scale, shift) tables (set by WG_LOAD_X; the values are requested later, WG_LOAD_BN)
    int fy0 = 0, fx0 = 0;                                     // origin of the tile in flight (set by WG_LOAD, consumed by WG_STORE one iteration later)

    // loads only (clamped addresses): the zero-padding / tail selects are applied by WG_STORE one tile later, so that nothing waits for these
    // loads while the current tile's MFMAs run
#define WG_TILE_ORIGIN(tile_)                                                                                                      \
        int n_ = (tile_) / (tiles_x * tiles_y);                                                                                    \
        const int rem_ = (tile_) - n_ * tiles_x * tiles_y;                                                                         \
        const int ty_ = rem_ / tiles_x;                                                                                            \
        const int gy0_ = ty_ * WG_TH, gx0_ = (rem_ - ty_ * tiles_x) * WG_TW;
#define WG_LOAD_X(tile_)                                                                                                            \
    do {                                                                                                                            \
        WG_TILE_ORIGIN(tile_)                                                                                                      \
        fy0 = gy0_; fx0 = gx0_;                                                                                                    \
        const int xc_ = fx0 - 1 + p0;                           /* this thread's halo column */                                    \
        const bool xv_ = cok && xc_ >= 0 && xc_ < a.W;                                                                             \
        const int y6_ = fy0 - 1 + hy6, x6_ = fx0 - 1 + hx6;                                                                        \
        const bool v6_ = cok && p0 < 2 * WG_HH && y6_ >= 0 && y6_ < a.H && x6_ < a.W;                                              \
        const float* xp_ = sg.p;                                                                                                   \
        long bo_ = cok ? cx : 0;                                                                                                   \
        if (a.group_n > 0) { const int grp_ = n_ / a.group_n; n_ -= grp_ * a.group_n; xp_ += grp_ * sgs; bo_ += grp_ * sbgs; }     \
        bo_ += (long)(int)(((float)n_ + 0.5f) * inv_bn_gn) * sg.bn_gs;                                                             \
        bnoff = bo_;                                                                                                               \
        const float* xb_ = xp_ + (long)n_ * sg.sn + (cok ? cx : 0);                                                                \
        const int xo_ = (fy0 - 1) * xrow + (xv_ ? xc_ * xpl : 0);                                                                   \
        _Pragma("unroll") for (int i = 0; i < WG_HH; i++) {                                                                        \
            const int y_ = fy0 - 1 + i;                             /* scalar */                                                   \
            rx[i] = *reinterpret_cast<const float4*>(xb_ + (unsigned)((y_ >= 0 && y_ < a.H) ? xo_ + i * xrow : 0));               \
        }                                                                                                                          \
        rx[WG_HH] = *reinterpret_cast<const float4*>(xb_ + (unsigned)(v6_ ? y6_ * xrow + x6_ * xpl : 0));                         \
    } while (0)
    // (scale, shift) of the tile in flight: two L2-resident float4, clamped to a valid address without BatchNorm
#define WG_LOAD_BN()                                                                                                                \
    do {                                                                                                                            \
        rxs = *reinterpret_cast<const float4*>(xbn ? sg.bn_scale + bnoff : sg.p);                                                  \
        rxh = *reinterpret_cast<const float4*>(xbn ? sg.bn_shift + bnoff : sg.p);                                                  \
    } while (0)
    // the dY rows of the tile in flight: requested in the tail of the MFMA phase (the registers of the A fragments that are done serve them), consumed after the X rows
#define WG_LOAD_Y(tile_, i0_, i1_)                                                                                                            \
    do {                                                                                                                            \
        WG_TILE_ORIGIN(tile_)                                                                                                      \
        const int yx_ = gx0_ + p0;                              /* this thread's tile column */                                    \
        const bool yv_ = yok && yx_ < a.W;                                                                                         \
        const float* dyb_ = a.dy;                                                                                                  \
        if (a.group_n > 0) { const int grp_ = n_ / a.group_n; n_ -= grp_ * a.group_n; dyb_ += grp_ * a.dy_gs; }                    \
        const float* yb_ = dyb_ + (long)n_ * a.dy_sn;               /* scalar */                                                   \
        const int yo_ = gy0_ * yrow + (yok ? yc : 0) + (yv_ ? yx_ * a.dy_ld : 0);                                                  \
        _Pragma("unroll") for (int i = (i0_); i < (i1_); i++)                                                                      \
            ry[i] = *reinterpret_cast<const float4*>(yb_ + (unsigned)(gy0_ + i < a.H ? yo_ + i * yrow : (yok ? yc : 0)));          \
    } while (0)
    // value -> (hi, lo) halves of four channels, stored at dst_ / dst_ + 64.  bf16: one packed conversion per pair, the high halves re-expanded by shift / mask (16 VALU
    // per float4 incl. the four selects; the generic form converts every high half twice)
#define WG_SPLIT_STORE(dst_, v_)                                                                                                   \
    do {                                                                                                                            \
        v4 hi_, lo_;                                                                                                                \
        if (is_bf16<T>::value) {                                                                                                    \
            typedef T t2_ __attribute__((ext_vector_type(2)));                                                                      \
            t2_ h01_, h23_; h01_[0] = (T)(v_).x; h01_[1] = (T)(v_).y; h23_[0] = (T)(v_).z; h23_[1] = (T)(v_).w;                     \
            const unsigned u01_ = __builtin_bit_cast(unsigned, h01_), u23_ = __builtin_bit_cast(unsigned, h23_);                   \
            t2_ l01_, l23_;                                                                                                         \
            l01_[0] = (T)((v_).x - __builtin_bit_cast(float, u01_ << 16)); l01_[1] = (T)((v_).y - __builtin_bit_cast(float, u01_ & 0xffff0000u)); \
            l23_[0] = (T)((v_).z - __builtin_bit_cast(float, u23_ << 16)); l23_[1] = (T)((v_).w - __builtin_bit_cast(float, u23_ & 0xffff0000u)); \
            hi_[0] = h01_[0]; hi_[1] = h01_[1]; hi_[2] = h23_[0]; hi_[3] = h23_[1];                                                 \
            lo_[0] = l01_[0]; lo_[1] = l01_[1]; lo_[2] = l23_[0]; lo_[3] = l23_[1];                                                 \
        } else {                                                                                                                    \
            hi_[0] = (T)(v_).x; hi_[1] = (T)(v_).y; hi_[2] = (T)(v_).z; hi_[3] = (T)(v_).w;                                        \
            lo_[0] = (T)((v_).x - (float)hi_[0]); lo_[1] = (T)((v_).y - (float)hi_[1]);                                            \
            lo_[2] = (T)((v_).z - (float)hi_[2]); lo_[3] = (T)((v_).w - (float)hi_[3]);                                            \
        }                                                                                                                          \
        *reinterpret_cast<v4*>(dst_) = hi_;                                                                                        \
        *reinterpret_cast<v4*>((dst_) + 64) = lo_;                                                                                 \
    } while (0)
#define WG_X_ELEM(i_, ok_, dst_)                                                                                                   \
    do {                                                                                                                            \
        float4 v_ = rx[i_];                                                                                                        \
        if (xbn) {      /* act(x * scale + shift): what the forward conv consumed; zero padding applies to the normalised tensor */ \
            v_.x = fmaf(v_.x, rxs.x, rxh.x); v_.y = fmaf(v_.y, rxs.y, rxh.y); v_.z = fmaf(v_.z, rxs.z, rxh.z); v_.w = fmaf(v_.w, rxs.w, rxh.w); \
            v_.x = v_.x > 0.f ? v_.x : xsl * v_.x; v_.y = v_.y > 0.f ? v_.y : xsl * v_.y;                                          \
            v_.z = v_.z > 0.f ? v_.z : xsl * v_.z; v_.w = v_.w > 0.f ? v_.w : xsl * v_.w;                                          \
        }                                                                                                                          \
        v_.x = (ok_) ? v_.x : 0.f; v_.y = ((ok_) && xm1) ? v_.y : 0.f; v_.z = ((ok_) && xm2) ? v_.z : 0.f; v_.w = ((ok_) && xm3) ? v_.w : 0.f; \
        WG_SPLIT_STORE(dst_, v_);                                                                                                  \
    } while (0)
#define WG_STORE()                                                                                                                  \
    do {                                                                                                                            \
        const int xc_ = fx0 - 1 + p0, yx_ = fx0 + p0;                                                                              \
        const bool xv_ = cok && xc_ >= 0 && xc_ < a.W, yv_ = yok && yx_ < a.W;                                                     \
        const int y6_ = fy0 - 1 + hy6, x6_ = fx0 - 1 + hx6;                                                                        \
        const bool v6_ = cok && y6_ >= 0 && y6_ < a.H && x6_ < a.W;                                                                \
        _Pragma("unroll") for (int i = 0; i < WG_HH; i++) {                                                                        \
            const int y_ = fy0 - 1 + i;                                                                                            \
            const bool ok_ = xv_ && y_ >= 0 && y_ < a.H;                                                                           \
            WG_X_ELEM(i, ok_, &Xh[xl + i * (WG_HW * WG_PITCH)]);                                                                   \
        }                                                                                                                          \
        if (p0 < 2 * WG_HH) WG_X_ELEM(WG_HH, v6_, &Xh[xl6]);                                                                       \
        _Pragma("unroll") for (int i = 0; i < WG_TH; i++) {                                                                        \
            const bool ok_ = yv_ && fy0 + i < a.H;                                                                                 \
            float4 v_ = ry[i];                                                                                                     \
            if (YS16) {     /* (Cout a multiple of 32: a piece is inside or outside as a whole) */                                 \
                v_.x = ok_ ? v_.x : 0.f; v_.y = ok_ ? v_.y : 0.f; v_.z = ok_ ? v_.z : 0.f; v_.w = ok_ ? v_.w : 0.f;               \
                *reinterpret_cast<float4*>(&Yt[yl16 + i * (WG_TW * WG_PITCH)]) = v_;                                               \
                continue;                                                                                                          \
            }                                                                                                                      \
            v_.x = ok_ ? v_.x : 0.f; v_.y = (ok_ && ym1) ? v_.y : 0.f; v_.z = (ok_ && ym2) ? v_.z : 0.f; v_.w = (ok_ && ym3) ? v_.w : 0.f; \
            WG_SPLIT_STORE(&Yt[xl + i * (WG_TW * WG_PITCH)], v_);                                                                  \
        }                                                                                                                          \
    } while (0)

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    // tr-read lane roles: group g = (lane >> 4) & 1 -> channel half; lane & 15 -> (pixel row (lane & 15) >> 2, column quad lane & 3); lane >> 5 -> pixel octet
    const int prow = (lane >> 5) * 8 + ((lane & 15) >> 2);            // + 4 rr
    const int ccol = ((lane >> 4) & 1) * 16 + 4 * (lane & 3);
    const int yoff = prow * WG_PITCH + (RSPLIT ? 0 : wm * 32) + ccol;
    const int xoff = prow * WG_PITCH + wn * 32 + ccol;

    // MFMA phase, halo-row-major (round 6).  The former order -- tile row r, tap (dy, dx): four fragment reads, three dependent MFMAs, the next tap's reads into the SAME registers --
    // read every X fragment three times (halo row h serves (r, dy) = (h, 0), (h - 1, 1), (h - 2, 2)) and exposed one LDS round trip per tap: ~12 000 cycles per tile against
    // 3456 of matrix work.  Now the A fragments of the wave's NR tile rows are read once and kept, the X fragments of halo row h / column shift dx are read once (two register
    // sets: the next one is in flight while this one multiplies) and feed up to three taps -- independent accumulators back to back, no dependent pair adjacent: 88 instead
    // of 160 fragment reads per tile and wave.
    constexpr int NR = RSPLIT ? WG_TH / 2 : WG_TH, NH = NR + 2;
    const int rbase = RSPLIT ? wm * NR : 0;                    // first tile row of this wave (wave-uniform)
    const T* const ybase = Yt + rbase * (WG_TW * WG_PITCH) + yoff;
    const T* const xbase = Xh + rbase * (WG_HW * WG_PITCH) + xoff;

    int tile = (int)blockIdx.z;
    if (tile < ntiles) { WG_LOAD_X(tile); WG_LOAD_BN(); WG_LOAD_Y(tile, 0, WG_TH); }
    for (; tile < ntiles; tile += (int)gridDim.z) {
        WG_STORE();
        __syncthreads();
        // next tile (unconditional, no branch inside the MFMA sequence: past the end the current tile is requested again, unused)
        const int ntile = tile + (int)gridDim.z < ntiles ? tile + (int)gridDim.z : tile;
        WG_LOAD_X(ntile);
        v8 ah[NR], al[NR], bh[2], bl[2];
#pragma unroll
        for (int r = 0; r < NR; r++) { ah[r] = tr_frag<T>(ybase + r * (WG_TW * WG_PITCH), 0, 4 * WG_PITCH); al[r] = tr_frag<T>(ybase + r * (WG_TW * WG_PITCH), 64, 4 * WG_PITCH + 64); }
        bh[0] = tr_frag<T>(xbase, 0, 4 * WG_PITCH); bl[0] = tr_frag<T>(xbase, 64, 4 * WG_PITCH + 64);
#pragma unroll
        for (int idx = 0; idx < NH * 3; idx++) {
            const int h = idx / 3, dx = idx - 3 * h, cur = idx & 1;
            if (idx + 1 < NH * 3) {
                const int h1 = (idx + 1) / 3, dx1 = (idx + 1) - 3 * h1;
                const T* xb = xbase + (h1 * WG_HW + dx1) * WG_PITCH;
                bh[cur ^ 1] = tr_frag<T>(xb, 0, 4 * WG_PITCH); bl[cur ^ 1] = tr_frag<T>(xb, 64, 4 * WG_PITCH + 64);
            }
            // the rest of the next tile is requested as the A fragments retire (tile row r is last used by halo row r + 2): no register is held for it before
            if (idx == (NH - 3) * 3) WG_LOAD_BN();
            if (idx == (NH - 2) * 3) WG_LOAD_Y(ntile, 0, WG_TH / 2);
            if (idx == (NH - 1) * 3) WG_LOAD_Y(ntile, WG_TH / 2, WG_TH);
            // taps served by this fragment: dy = h - r for every tile row r of the wave with 0 <= dy <= 2; product-major so that consecutive MFMAs hit different accumulators
#pragma unroll
            for (int pr = 0; pr < 3; pr++)
#pragma unroll
                for (int dy = 0; dy < 3; dy++) {
                    const int r = h - dy;
                    if (r < 0 || r >= NR) continue;
                    f32x16 c = acc[dy * 3 + dx];
                    c = pr == 0 ? mfma16(al[r], bh[cur], c) : (pr == 1 ? mfma16(ah[r], bl[cur], c) : mfma16(ah[r], bh[cur], c));
                    acc[dy * 3 + dx] = c;
                }
        }
        __syncthreads();
    }
#undef WG_LOAD_X
#undef WG_LOAD_Y
#undef WG_LOAD_BN
#undef WG_TILE_ORIGIN
#undef WG_STORE
#undef WG_SPLIT_STORE
#undef WG_X_ELEM

    const int k = k0 + wn * 32 + (lane & 31);
    if (k < a.Ktot) {
#pragma unroll
        for (int t = 0; t < 9; t++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int o = o0 + (RSPLIT ? 0 : wm * 32) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (o < a.Cout) atomicAdd(WGRAD_DST(a, blockIdx.z) + ((long)t * a.Cout_pad + o) * a.Ktot + k, acc[t][r]);
            }
        }
    }
}

}  // namespace

// 1 = handled: 3x3 weight gradient with >= 32 channels on both sides on the 16-bit matrix pipe (split bf16 operands).  Same packed fp32
// gradient layout (dwp[tap][Cout_pad][Ktot], segments padded to 16) and (group, sample) time-batched addressing as k_conv_wgrad_tile.
static bool wgrad_hx_applies(const WgradArgs& a) {
    if (a.KS != 3 || a.precision != PREC_BF16X3 || a.Cout < 32 || a.Ktot < 32 || a.W < 8 || a.H < 2) return false;
    for (int s = 0; s < a.nsrc; s++) if ((a.src[s].ld & 3) || (a.src[s].sn & 3)) return false;
    if ((a.dy_ld & 3) || (a.dy_sn & 3)) return false;
    if (a.dy_s16 && (a.Cout & 31)) return false;
    return true;
}
bool wgrad_src_lazy_ok(const WgradArgs& a) { return wgrad_hx_applies(a); }      // k_wgrad_hx is the only weight-gradient kernel that applies ConvSrc.bn_*
int conv_hx_wgrad_try(const WgradArgs& a, hipStream_t st, bool dry) {
    if (!wgrad_hx_applies(a)) return 0;
    g_last_conv_kernel = CK_WGRAD_HX;
    if (dry) return 1;
    const int tx = cdiv(a.W, WG_TW), ty = cdiv(a.H, WG_TH);
    const long ntiles = (long)a.N * tx * ty;
    if (ntiles >= (1L << 30)) return 0;
    const int kt = cdiv(a.Ktot, WG_KC), ot = cdiv(a.Cout, WG_OC);
    // Register bound 2 (256 per lane) with still ONE persistent workgroup per CU: the side stream's workgroup then leaves half of every SIMD's register file
    // to the BPTT chain on the main stream, which shares the CU with it (measured, E/R/A/D step: unbounded 89.0 ms; bound 2 with 256 / 384 / 512 workgroups
    // 85.6 / 87.2 / 89.6 ms; serialised streams 94.5 -> 91.6 ms)
    long g = 256 / ((long)kt * ot);
    if (g < 1) g = 1;
    if (g > ntiles) g = ntiles;
    WgradArgs b = a;
    // bit-reproducible mode: one copy of the packed layout per pixel split.  With ONE split (kt x ot >= 129 tiles: the ConvLSTM gate layers) every element of dwp has exactly one
    // contributing lane per launch -- the plain flush is already order-free: no copies, no fold (the row-split layout of <= 32 output channels has two contributors per element)
    if (b.det_slab && g == 1 && a.Cout > 32) b.det_slab = nullptr;
    if (b.det_slab) { g = wgrad_det_begin(b, g, st); if (g <= 0) return -1; }
    if (a.dy_s16) {
        if (a.Cout <= 32) hipLaunchKernelGGL((k_wgrad_hx<__bf16, 2, true, true>), dim3(kt, ot, (unsigned)g), dim3(256), 0, st, b, tx, ty);
        else hipLaunchKernelGGL((k_wgrad_hx<__bf16, 2, false, true>), dim3(kt, ot, (unsigned)g), dim3(256), 0, st, b, tx, ty);
    } else if (a.Cout <= 32) hipLaunchKernelGGL((k_wgrad_hx<__bf16, 2, true, false>), dim3(kt, ot, (unsigned)g), dim3(256), 0, st, b, tx, ty);
    else hipLaunchKernelGGL((k_wgrad_hx<__bf16, 2, false, false>), dim3(kt, ot, (unsigned)g), dim3(256), 0, st, b, tx, ty);
    if (b.det_slab) wgrad_det_end(b, g, st);
    return 1;
}
