// 3x3 convolution on the CDNA4 16-bit matrix pipe with fp32-class accuracy: split operands (hi + lo halves), fp32 accumulation.
//
// Why: the CADDY step is FP32-compute-bound (conv arithmetic intensity ~122 FLOP/B, SURVEY.md section 7 hard part 1); the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32) peaks at 157 TFLOP/s, the 16-bit MFMA (v_mfma_f32_32x32x16_{f16,bf16}) at ~2.5 PFLOP/s.  A fp32 value x is split
// as x = hi + lo with hi = round16(x), lo = round16(x - hi); the product a*b is taken as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with fp32
// accumulation (3 MFMAs per fp32 product -> ~830 TFLOP/s effective peak).  f16 halves carry 11 + 11 mantissa bits (error ~2^-22, "fp32
// re-ordering class" -- measured indistinguishable from a fp32 summation-order change over 15 closed-loop steps);
// bf16 halves carry 8 + 8 bits with the full fp32 exponent range (error ~2^-16, used for the gradient operands whose magnitude is
// unbounded below).  NPROD = 1 keeps only hi*hi (plain 16-bit operands; selectable for the frozen VGG19 loss network).
//
// Structure (what the exact-fp32 k_conv_fwd could not do): operands arrive PRE-SPLIT where they are reused -- weights are split once per
// optimiser step by pack_hx (frozen VGG19 weights: once) -- and activations are split ONCE PER TILE, not once per tap: a workgroup owns a
// TH x TW pixel tile x BN output channels and, per 32-channel chunk, stages the (TH+2) x (TW+2) halo of the fp32 input through registers
// (float4 loads, cvt + sub, two 8-byte LDS stores per float4) -- all nine taps then read shifted rows of that one LDS image.  Compared with
// the per-tap im2col staging this cuts the global->LDS traffic of the activations 9x and puts the conversion at ~3 % of the MFMA time.
// Weight tiles ([tap][chunk][Cout][hi 32 | lo 32], 16 KB per 128 channels) stream through a 2-deep LDS ring with a register prefetch.
// LDS row pitch = row bytes + 16 -> the ds_read_b128 fragment reads of the 32x32x16 MFMA are bank-conflict-free for 16 consecutive rows.
//
// Replaces nn.Conv2d(k=3, padding=1) of every wide layer on the path (model/layers/*.py), its dgrad
// (same kernel on the flipped / transposed split weights) and the VGG19 convolutions of the perceptual loss (conv + bias + ReLU forward,
// dgrad with the fused ReLU mask / L1 seed epilogue).
#include "common.h"
#include "pack.h"
#include <cstdlib>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Vec;
template <> struct Vec<_Float16> { typedef f16x8 v8; typedef f16x4 v4; };
template <> struct Vec<__bf16> { typedef bf16x8 v8; typedef bf16x4 v4; };

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

constexpr int KC = HX_KC;      // channels per chunk

// T: _Float16 / __bf16.  NPL planes staged (2: hi + lo, 3 products; 1: hi only).  Tile TH x TW pixels x BN output channels, 4 waves as WM x WN.
template <typename T, int NPL, int TH, int TW, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void k_conv_hx(ConvArgs a, int tiles_x, int tiles_y) {
    typedef typename Vec<T>::v8 v8;
    typedef typename Vec<T>::v4 v4;
    constexpr int BM = TH * TW;
    constexpr int HW_ = TW + 2, HH_ = TH + 2, HPX = HW_ * HH_;
    constexpr int PITCH = NPL * KC + 8;                      // LDS row pitch in 16-bit elements: 72 (144 B) or 40 (80 B)
    constexpr int NA = (HPX + 31) / 32;                      // halo pixels per thread (8 threads x float4 cover one pixel's 32-channel chunk)
    constexpr int BROW16 = NPL * KC * 2 / 16;                // 16-byte pieces per weight row
    constexpr int NB = (BN * BROW16 + 255) / 256;
    constexpr int TMt = BM / WM / 32, TNt = BN / WN / 32;
    static_assert(WM * WN == 4 && BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "tile / wave layout");
    static_assert(TW == 16, "row <-> pixel map assumes 16-pixel tile rows");
    __shared__ __attribute__((aligned(16))) T As[HPX * PITCH];
    __shared__ __attribute__((aligned(16))) T Bs[2][BN * PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int tile = blockIdx.x;
    const int n = tile / (tiles_x * tiles_y);
    tile -= n * tiles_x * tiles_y;
    const int y0 = (tile / tiles_x) * TH, x0 = (tile % tiles_x) * TW;
    const int n0 = blockIdx.y * BN;
    const int nchunks = a.Kq / KC;

    // ---- A staging roles: float4 column q of halo pixels (tid >> 3) + 32 i ----
    const int q = tid & 7;
    int pixoff[NA];                                           // (y * W + x) of the halo pixel, -1: outside the image / beyond the halo
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int p = (tid >> 3) + 32 * i;
        const int hy = p / HW_, hx = p - hy * HW_;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        pixoff[i] = (p < HPX && y >= 0 && y < a.H && x >= 0 && x < a.W) ? y * a.W + x : -1;
    }
    // (staging steps are macros, not lambdas: a by-reference capture of the kernel-argument struct / register arrays forces them into
    //  scratch memory)
    float4 ra[NA];
#define HX_LOAD_A(chunk_)                                                                                                          \
    do {                                                                                                                           \
        int s_ = 0, c0_ = (chunk_) * KC;                                                                                          \
        while (s_ + 1 < a.nsrc && c0_ >= (a.src[s_].C + KC - 1) / KC * KC) { c0_ -= (a.src[s_].C + KC - 1) / KC * KC; s_++; }     \
        const ConvSrc sg_ = a.src[s_];                          /* wave-uniform */                                                \
        const int c_ = c0_ + 4 * q;                                                                                               \
        const bool cok_ = c_ < sg_.C;                            /* branch-free: clamped addresses + selects */                     \
        const float* base_ = sg_.p + (long)n * sg_.sn + (cok_ ? c_ : 0);                                                          \
        const int pl_ = sg_.bcast ? 0 : sg_.ld;                                                                                   \
        _Pragma("unroll") for (int i = 0; i < NA; i++) {                                                                          \
            const bool ok_ = cok_ && pixoff[i] >= 0;                                                                               \
            float4 v_ = *reinterpret_cast<const float4*>(base_ + (long)(ok_ ? pixoff[i] : 0) * pl_);                              \
            v_.x = ok_ ? v_.x : 0.f;                                                                                               \
            v_.y = (ok_ && c_ + 1 < sg_.C) ? v_.y : 0.f;                                                                           \
            v_.z = (ok_ && c_ + 2 < sg_.C) ? v_.z : 0.f;                                                                           \
            v_.w = (ok_ && c_ + 3 < sg_.C) ? v_.w : 0.f;                                                                           \
            ra[i] = v_;                                                                                                            \
        }                                                                                                                          \
    } while (0)
#define HX_STORE_A()                                                                                                               \
    do {                                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < NA; i++) {                                                                          \
            const int p_ = (tid >> 3) + 32 * i;                                                                                   \
            if (HPX % 32 == 0 || p_ < HPX) {                                                                                       \
                v4 hi_, lo_;                                                                                                       \
                hi_[0] = (T)ra[i].x; hi_[1] = (T)ra[i].y; hi_[2] = (T)ra[i].z; hi_[3] = (T)ra[i].w;                                \
                lo_[0] = (T)(ra[i].x - (float)hi_[0]); lo_[1] = (T)(ra[i].y - (float)hi_[1]);                                     \
                lo_[2] = (T)(ra[i].z - (float)hi_[2]); lo_[3] = (T)(ra[i].w - (float)hi_[3]);                                     \
                *reinterpret_cast<v4*>(&As[p_ * PITCH + 4 * q]) = hi_;                                                            \
                if (NPL == 2) *reinterpret_cast<v4*>(&As[p_ * PITCH + KC + 4 * q]) = lo_;                                         \
            }                                                                                                                      \
        }                                                                                                                          \
    } while (0)
    // ---- B staging: the (tap, chunk) tile of this workgroup is BN rows x (NPL * 64) bytes, contiguous in global memory ----
    const T* wq = reinterpret_cast<const T*>(a.wq);
    const long wrow = (long)a.Cout_pad * (NPL * KC);
    u32x4 rb[NB];
#define HX_LOAD_B(tap_, chunk_)                                                                                                    \
    do {                                                                                                                           \
        const u32x4* g_ = reinterpret_cast<const u32x4*>(wq + ((long)(tap_) * nchunks + (chunk_)) * wrow + (long)n0 * (NPL * KC)); \
        _Pragma("unroll") for (int i = 0; i < NB; i++) {                                                                          \
            const int idx_ = tid + 256 * i;                                                                                       \
            if ((BN * BROW16) % 256 == 0 || idx_ < BN * BROW16) rb[i] = g_[idx_];                                                                             \
        }                                                                                                                          \
    } while (0)
#define HX_STORE_B(buf_)                                                                                                           \
    do {                                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < NB; i++) {                                                                          \
            const int idx_ = tid + 256 * i;                                                                                       \
            if ((BN * BROW16) % 256 == 0 || idx_ < BN * BROW16) { const int row_ = idx_ / BROW16, c16_ = idx_ - row_ * BROW16;                                \
                                      *reinterpret_cast<u32x4*>(&Bs[buf_][row_ * PITCH + c16_ * 8]) = rb[i]; }                    \
        }                                                                                                                          \
    } while (0)

    // ---- fragment addresses ----
    int abase[TMt], bbase[TNt];
#pragma unroll
    for (int i = 0; i < TMt; i++) {
        const int m = wm * (BM / WM) + i * 32 + (lane & 31);
        abase[i] = ((m / TW) * HW_ + (m % TW)) * PITCH + (lane >> 5) * 8;
    }
#pragma unroll
    for (int j = 0; j < TNt; j++) bbase[j] = (wn * (BN / WN) + j * 32 + (lane & 31)) * PITCH + (lane >> 5) * 8;

    f32x16 acc[TMt][TNt];
#pragma unroll
    for (int i = 0; i < TMt; i++)
#pragma unroll
        for (int j = 0; j < TNt; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // split-K over channel chunks (blockIdx.z): partial sums combined with atomics (accumulating dgrads) or slabs (a.split_stride)
    const int cper = (nchunks + a.splitk - 1) / a.splitk;
    const int ch0 = blockIdx.z * cper, ch1 = ch0 + cper < nchunks ? ch0 + cper : nchunks;
    if (ch0 < ch1) { HX_LOAD_A(ch0); HX_LOAD_B(0, ch0); }
    int bbuf = 0;
    for (int chunk = ch0; chunk < ch1; chunk++) {
        __syncthreads();                                      // every wave is done reading As (previous chunk)
        HX_STORE_A();
        if (chunk + 1 < ch1) HX_LOAD_A(chunk + 1);
#pragma unroll 1
        for (int tap = 0; tap < 9; tap++) {
            HX_STORE_B(bbuf);
            __syncthreads();
            if (tap < 8) HX_LOAD_B(tap + 1, chunk);
            else if (chunk + 1 < ch1) HX_LOAD_B(0, chunk + 1);
            const int toff = ((tap / 3) * HW_ + tap % 3) * PITCH;
            const T* Bt = Bs[bbuf];
#pragma unroll
            for (int s = 0; s < KC / 16; s++) {
                v8 fa[TMt][NPL], fb[TNt][NPL];
#pragma unroll
                for (int i = 0; i < TMt; i++)
#pragma unroll
                    for (int pl = 0; pl < NPL; pl++) fa[i][pl] = *reinterpret_cast<const v8*>(&As[abase[i] + toff + pl * KC + s * 16]);
#pragma unroll
                for (int j = 0; j < TNt; j++)
#pragma unroll
                    for (int pl = 0; pl < NPL; pl++) fb[j][pl] = *reinterpret_cast<const v8*>(&Bt[bbase[j] + pl * KC + s * 16]);
#pragma unroll
                for (int i = 0; i < TMt; i++)
#pragma unroll
                    for (int j = 0; j < TNt; j++) {
                        if (NPL == 2) {                       // small terms first
                            acc[i][j] = mfma16(fa[i][NPL - 1], fb[j][0], acc[i][j]);
                            acc[i][j] = mfma16(fa[i][0], fb[j][NPL - 1], acc[i][j]);
                        }
                        acc[i][j] = mfma16(fa[i][0], fb[j][0], acc[i][j]);
                    }
            }
            bbuf ^= 1;
        }
    }

    // ---- epilogue: D fragment map col = lane & 31 (output channel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (pixel of the tile) ----
#pragma unroll
    for (int j = 0; j < TNt; j++) {
        const int col = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
        if (col >= a.Cout) continue;
        const float bv = (a.bias && blockIdx.z == 0) ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TMt; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int y = y0 + m / TW, x = x0 + m % TW;
                if (y >= a.H || x >= a.W) continue;
                const long off = (long)n * a.out_sn + ((long)y * a.W + x) * a.out_ld + col;
                float v = acc[i][j][r] + bv;
                if (a.splitk > 1) {
                    if (a.split_stride) a.out[blockIdx.z * a.split_stride + off] = v;      // slabs: bias / activation applied by the reduce
                    else atomicAdd(a.out + off, v);
                    continue;
                }
                if (a.act == 1) v = tanhf(v);
                else if (a.act == 2) v = fmaxf(v, 0.f);
                if (a.mask) {
                    const float mk = a.mask[off];
                    if (a.seed_ref) { const float d = mk - a.seed_ref[off]; v += d > 0.f ? a.seed_w : (d < 0.f ? -a.seed_w : 0.f); }
                    v = mk > 0.f ? v : 0.f;
                }
                if (a.accumulate) v += a.out[off];
                a.out[off] = v;
            }
        }
    }
}

#undef HX_LOAD_A
#undef HX_STORE_A
#undef HX_LOAD_B
#undef HX_STORE_B

// ---- weight packing: OIHW fp32 (reference state_dict layout) -> split 16-bit tiles [tap][chunk][Cout_pad][hi 32 | lo 32] ----
template <typename T, int NPL>
__global__ void k_pack_hx(PackDesc d, T* wq, int Cout_pad, int dgrad_seg) {
    // forward (dgrad_seg < 0): rows = output channels, k = concatenated input segments, each padded to KC.
    // dgrad of segment s: rows = input channels of s, k = output channels (padded to KC), taps flipped.
    const int taps = d.KS * d.KS;
    int Kq = 0;
    if (dgrad_seg < 0) { for (int s = 0; s < d.nseg; s++) Kq += (d.seg_C[s] + KC - 1) / KC * KC; }
    else Kq = (d.Cout + KC - 1) / KC * KC;
    const int nch = Kq / KC;
    const long total = (long)taps * nch * Cout_pad * KC;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % KC); long r = i / KC; const int row = (int)(r % Cout_pad); r /= Cout_pad; const int ch = (int)(r % nch); const int tap = (int)(r / nch);
        const int k = ch * KC + kk;
        float v = 0.f;
        if (dgrad_seg < 0) {
            int base = 0, cin = -1;
            for (int s = 0; s < d.nseg; s++) {
                const int pad = (d.seg_C[s] + KC - 1) / KC * KC;
                if (k < base + pad) { if (k - base < d.seg_C[s]) cin = d.seg_off[s] + k - base; break; }
                base += pad;
            }
            if (row < d.Cout && cin >= 0) v = d.w[row / d.Co_each][((long)(row % d.Co_each) * d.Cin + cin) * taps + tap];
        } else {
            if (row < d.seg_C[dgrad_seg] && k < d.Cout) v = d.w[k / d.Co_each][((long)(k % d.Co_each) * d.Cin + d.seg_off[dgrad_seg] + row) * taps + (taps - 1 - tap)];
        }
        const T hi = (T)v;
        T* o = wq + (((long)tap * nch + ch) * Cout_pad + row) * (NPL * KC) + kk;
        o[0] = hi;
        if (NPL == 2) o[KC] = (T)(v - (float)hi);
    }
}

}  // namespace

// bytes of the split weight buffer of a layer: forward form (seg < 0) or dgrad form of input segment `seg`
size_t hx_weight_bytes(const PackDesc& d, int seg, int rows_pad, int planes) {
    int Kq = 0;
    if (seg < 0) { for (int s = 0; s < d.nseg; s++) Kq += round_up(d.seg_C[s], HX_KC); }
    else Kq = round_up(d.Cout, HX_KC);
    return (size_t)d.KS * d.KS * Kq * rows_pad * planes * 2;
}
int hx_kq(const PackDesc& d, int seg) {
    int Kq = 0;
    if (seg < 0) { for (int s = 0; s < d.nseg; s++) Kq += round_up(d.seg_C[s], HX_KC); }
    else Kq = round_up(d.Cout, HX_KC);
    return Kq;
}
int pack_hx(const PackDesc& d, void* wq, int rows_pad, int seg, int precision, hipStream_t st) {
    const long total = (long)d.KS * d.KS * hx_kq(d, seg) * rows_pad;
    const unsigned grid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    switch (precision) {
        case PREC_F16X3: hipLaunchKernelGGL((k_pack_hx<_Float16, 2>), dim3(grid), dim3(256), 0, st, d, (_Float16*)wq, rows_pad, seg); break;
        case PREC_BF16X3: hipLaunchKernelGGL((k_pack_hx<__bf16, 2>), dim3(grid), dim3(256), 0, st, d, (__bf16*)wq, rows_pad, seg); break;
        case PREC_F16X1: hipLaunchKernelGGL((k_pack_hx<_Float16, 1>), dim3(grid), dim3(256), 0, st, d, (_Float16*)wq, rows_pad, seg); break;
        case PREC_BF16X1: hipLaunchKernelGGL((k_pack_hx<__bf16, 1>), dim3(grid), dim3(256), 0, st, d, (__bf16*)wq, rows_pad, seg); break;
        default: return -1;
    }
    return 0;
}
int hx_pick_bn(int cout) { return cout > 64 ? 128 : (cout > 32 ? 64 : 32); }

// 1 = handled.  Requirements: 3x3, split weights present (a.wq, packed for a.precision with rows padded to hx_pick_bn(Cout)).
int conv_hx_try(const ConvArgs& a0, hipStream_t st) {
    ConvArgs a = a0;
    if (a.KS != 3 || !a.wq || a.precision < PREC_F16X3 || a.precision > PREC_BF16X1) return 0;
    int kq = 0;
    for (int s = 0; s < a.nsrc; s++) { if ((a.src[s].ld & 3) || (a.src[s].sn & 3)) return -1; kq += round_up(a.src[s].C, HX_KC); }
    a.Kq = kq;
    const int bn = hx_pick_bn(a.Cout);
    a.Cout_pad = round_up(a.Cout, bn);
    if (a.mask && a.accumulate) return -1;
    const int nchunks = kq / HX_KC;
    // tiles: 8x16 pixels x 128 channels, or 16x16 x 64 / 32
    const int th = bn == 128 ? 8 : 16;
    const int tx = cdiv(a.W, 16), ty = cdiv(a.H, th);
    const long blocks = (long)a.N * tx * ty * (a.Cout_pad / bn);
    // under-filled launches: split the channel chunks across blockIdx.z.  Accumulating launches (dgrad +=) combine with fp32 atomics; assigning
    // launches use the caller's slab scratch + the fixed-order k_split_reduce (bit-reproducible forward), as k_conv_fwd does.
    a.splitk = 1; a.split_stride = 0;
    float* real_out = a.out; long real_sn = a.out_sn; int real_ld = a.out_ld; const float* real_bias = a.bias; const int real_act = a.act;
    const long P = (long)a.N * a.H * a.W;
    if (blocks < 200 && nchunks >= 4 && !a.mask) {
        int want = (int)((256 + blocks - 1) / blocks);
        if (want > nchunks / 2) want = nchunks / 2;
        if (want > 8) want = 8;
        if (want >= 2) {
            if (a.accumulate && a.act == 0 && !a.bias) a.splitk = want;
            else if (!a.accumulate && a.split_scratch) {
                const int ldc = round_up(a.Cout, 4);
                while (want >= 2 && (long)want * P * ldc > a.split_cap) want--;
                if (want >= 2) { a.splitk = want; a.split_stride = P * ldc; a.out = a.split_scratch; a.out_sn = (long)a.H * a.W * ldc; a.out_ld = ldc; a.bias = nullptr; a.act = 0; }
            }
        }
    }
    dim3 grid((unsigned)((long)a.N * tx * ty), a.Cout_pad / bn, a.splitk);
#define HX_LAUNCH(T_, NPL_)                                                                                                       \
    do {                                                                                                                          \
        if (bn == 128) hipLaunchKernelGGL((k_conv_hx<T_, NPL_, 8, 16, 128, 2, 2>), grid, dim3(256), 0, st, a, tx, ty);            \
        else if (bn == 64) hipLaunchKernelGGL((k_conv_hx<T_, NPL_, 16, 16, 64, 4, 1>), grid, dim3(256), 0, st, a, tx, ty);        \
        else hipLaunchKernelGGL((k_conv_hx<T_, NPL_, 16, 16, 32, 4, 1>), grid, dim3(256), 0, st, a, tx, ty);                      \
    } while (0)
    switch (a.precision) {
        case PREC_F16X3: HX_LAUNCH(_Float16, 2); break;
        case PREC_BF16X3: HX_LAUNCH(__bf16, 2); break;
        case PREC_F16X1: HX_LAUNCH(_Float16, 1); break;
        default: HX_LAUNCH(__bf16, 1); break;
    }
#undef HX_LAUNCH
    g_last_conv_kernel = bn == 128 ? CK_HX_128 : (bn == 64 ? CK_HX_64 : CK_HX_32);
    if (a.split_stride) conv_split_reduce_launch(a.split_scratch, a.split_stride, a.splitk, a.out_ld, a.H * a.W, P, a.Cout, real_out, real_sn, real_ld, real_bias, real_act, st);
    return 1;
}
