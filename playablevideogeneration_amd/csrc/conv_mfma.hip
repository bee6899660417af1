// Implicit-GEMM convolution on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Replaces every nn.Conv2d on the CADDY hot path: 3x3 / 1x1 / 7x7, stride 1,
// "same" zero padding, NHWC activations, the input being the channel-concatenation of up to three segments
// (ConvLSTM: [x, (a|v) broadcast, h_prev] -- convolutional_lstm_cell.py:88-95 -- without materialising the cat).
//   forward / dgrad :  out[p,o]  = sum_{tap,k} A[p+tap,k] * Wp[tap][o][k]          (k_conv_fwd; dgrad = same kernel on
//                                                                                    flipped/transposed packed weights)
//   wgrad           :  dWp[tap][o][k] += sum_p dY[p,o] * A[p+tap,k]                 (k_conv_wgrad)
//
// Tiling (wave64, 4 waves / workgroup): GEMM-M = output pixels, GEMM-N = output channels, GEMM-K = taps x channels in
// chunks of 16.  A (pixels x 16ch) and B (couts x 16ch) tiles are staged through LDS with a 20-float row pitch, which
// makes the ds_read_b128 fragment reads conflict-free (16 rows x 4 banks cover all 64 banks exactly once).
// Next tile's global loads are issued into registers before the MFMA block of the current tile (register double buffer).
// fp32 MFMA keeps the result bitwise a k-ordered fmaf chain, which is what holds the 1e-5 frame-MSE parity bound over
// 15 recurrent steps (hard part 1); roofline for this kernel = 157.3 TFLOP/s fp32 matrix peak.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int BK = CONV_BK;
constexpr int LDSK = BK + 4;

struct SegRef { const float* p; long sn; int ld; int C; int bcast; int c0; int idx; };

// locate channel `k` (index into the padded concatenation) -> segment + channel inside it
__device__ __forceinline__ SegRef find_seg(const ConvSrc* src, int nsrc, int k) {
    int s = 0;
    while (s + 1 < nsrc && k >= src[s].Cpad) { k -= src[s].Cpad; s++; }
    SegRef r; r.p = src[s].p; r.sn = src[s].sn; r.ld = src[s].ld; r.C = src[s].C; r.bcast = src[s].bcast; r.c0 = k; r.idx = s;
    return r;
}

__device__ __forceinline__ float4 load4_masked(const float* q, int c, int C) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c + 4 <= C) v = *reinterpret_cast<const float4*>(q);
    else {
        if (c < C) v.x = q[0];
        if (c + 1 < C) v.y = q[1];
        if (c + 2 < C) v.z = q[2];
    }
    return v;
}

// value of the (virtual, zero-padded) concatenated input at pixel (n,yy,xx), channels seg.c0+kq*4 .. +3
__device__ __forceinline__ float4 load_src4(const SegRef& sg, int kq, bool valid, int n, int yy, int xx, int W) {
    int c = sg.c0 + kq * 4;
    if (!valid || c >= sg.C) return make_float4(0.f, 0.f, 0.f, 0.f);
    const float* q = sg.p + (long)n * sg.sn + (sg.bcast ? 0L : (long)(yy * W + xx) * sg.ld) + c;
    return load4_masked(q, c, sg.C);
}

template <int A_PER, int B_PER>
struct TileRegs { float4 a[A_PER]; float4 b[B_PER]; };

// Incremental K iterator of the implicit GEMM: tap (outer) -> input segment -> 16-channel chunk (inner).  Everything that
// depends on the tap (shifted pixel, zero-padding validity) or on the segment (base pointer, pitch, channel count) is
// recomputed only when that level changes; the common step is "pointers += 16".  This keeps the per-K-step instruction
// count low -- the loop is issue-bound, not MFMA-bound, when a launch has only 1-4 waves per SIMD (R's small feature maps).
template <int A_PER, int B_PER>
struct ConvIter {
    const float* ap[A_PER];   // current A pointers (row i, channel c0 + kq*4 of the current segment)
    const float* bp[B_PER];   // current weight pointers
    bool aok[A_PER];          // shifted pixel inside the image (and row valid)
    int pixoff[A_PER];        // (y+dy)*W + (x+dx)
    int tap, seg, c0, segC, segCpad;
};

template <int A_PER, int B_PER>
__device__ __forceinline__ void iter_set_seg(ConvIter<A_PER, B_PER>& I, const ConvArgs& a, int kq, const int* pn) {
    const ConvSrc sg = a.src[I.seg];
    I.segC = sg.C; I.segCpad = sg.Cpad; I.c0 = 0;
#pragma unroll
    for (int i = 0; i < A_PER; i++)
        I.ap[i] = sg.p + (long)pn[i] * sg.sn + (sg.bcast ? 0L : (long)I.pixoff[i] * sg.ld) + kq * 4;
}
template <int A_PER, int B_PER, int BN>
__device__ __forceinline__ void iter_set_tap(ConvIter<A_PER, B_PER>& I, const ConvArgs& a, int tap, int R, int n0, int tid, int kq,
                                             const int* pn, const int* py, const int* px, const bool* pv) {
    I.tap = tap; I.seg = 0;
    int dy = tap / a.KS - R, dx = tap % a.KS - R;
#pragma unroll
    for (int i = 0; i < A_PER; i++) {
        int yy = py[i] + dy, xx = px[i] + dx;
        I.aok[i] = pv[i] && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        I.pixoff[i] = I.aok[i] ? yy * a.W + xx : 0;
    }
#pragma unroll
    for (int i = 0; i < B_PER; i++) {
        int r = (tid >> 2) + 64 * i;
        I.bp[i] = a.wp + ((long)tap * a.Cout_pad + n0 + (r < BN ? r : 0)) * a.Ktot + kq * 4;
    }
    iter_set_seg(I, a, kq, pn);
}
// load the current (tap, segment, chunk) tile into registers and advance
template <int A_PER, int B_PER, int BN>
__device__ __forceinline__ TileRegs<A_PER, B_PER> iter_load(ConvIter<A_PER, B_PER>& I, const ConvArgs& a, int R, int n0, int tid, int kq,
                                                            const int* pn, const int* py, const int* px, const bool* pv) {
    TileRegs<A_PER, B_PER> t;
    const int c = I.c0 + kq * 4;
#pragma unroll
    for (int i = 0; i < A_PER; i++) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (I.aok[i] && c < I.segC) {
            if (c + 4 <= I.segC) v = *reinterpret_cast<const float4*>(I.ap[i]);
            else v = load4_masked(I.ap[i], c, I.segC);
        }
        t.a[i] = v;
        I.ap[i] += BK;
    }
#pragma unroll
    for (int i = 0; i < B_PER; i++) {
        t.b[i] = *reinterpret_cast<const float4*>(I.bp[i]);
        I.bp[i] += BK;
    }
    I.c0 += BK;
    if (I.c0 >= I.segCpad) {                      // wave-uniform
        if (I.seg + 1 < a.nsrc) { I.seg++; iter_set_seg(I, a, kq, pn); }
        else if (I.tap + 1 < a.KS * a.KS) iter_set_tap<A_PER, B_PER, BN>(I, a, I.tap + 1, R, n0, tid, kq, pn, py, px, pv);
    }
    return t;
}

constexpr int LDSH = 24;

template <int NS>
__device__ __forceinline__ void split_store(unsigned short* base, int plane_stride, float4 v) {
    float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int pl = 0; pl < NS; pl++) {
        bf16x4 h;
#pragma unroll
        for (int e = 0; e < 4; e++) { h[e] = (__bf16)x[e]; x[e] -= (float)h[e]; }
        *reinterpret_cast<bf16x4*>(base + pl * plane_stride) = h;
    }
}

// NS = 0: exact fp32 (v_mfma_f32_32x32x2_f32).  NS = 2 / 3: split-bf16 planes (see above).
// K-loop: one step = CPS consecutive 16-channel chunks (CPS = 2 -> 32 channels per barrier, which amortises the barrier, the LDS
// round trip and the loop bookkeeping over twice the MFMA work); step `s` is staged registers -> LDS buffer (s & 1) ->
// fragments -> MFMA while the global loads of step s+1 are in flight (register double buffer).
template <int TM, int TN, int WM, int WN, int NS, int CPS>
__global__ __launch_bounds__(256) void k_conv_fwd(ConvArgs a) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int A_PER = (BM * 4 + 255) / 256;
    constexpr int B_PER = (BN * 4 + 255) / 256;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int PF = CPS * BK + 4;                                           // fp32 LDS row pitch (floats): 20 or 36, both conflict-free for ds_read_b128
    constexpr int PH = CPS * BK + 8;                                           // bf16 LDS row pitch (elements): 24 or 40
    constexpr int A_BYTES = NS == 0 ? BM * PF * 4 : NS * BM * PH * 2;          // one buffer
    constexpr int B_BYTES = NS == 0 ? BN * PF * 4 : NS * BN * PH * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];
    __shared__ long rowoff[BM];
    __shared__ long resoff[BM];      // residual input (a.res) offsets, same rows

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int HW = a.H * a.W;
    const long P = (long)a.N * HW;
    const long p0 = (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int R = a.KS >> 1;
    const int kq = tid & 3;

    int pn[A_PER], py[A_PER], px[A_PER];
    bool pv[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; i++) {
        int r = (tid >> 2) + 64 * i;
        long p = p0 + r;
        pv[i] = (r < BM) && p < P;
        long pp = pv[i] ? p : 0;
        pn[i] = (int)(pp / HW);
        int rem = (int)(pp - (long)pn[i] * HW);
        py[i] = rem / a.W;
        px[i] = rem - py[i] * a.W;
    }
    if (tid < BM) {
        long p = p0 + tid;
        long off = -1;
        long roff = 0;
        if (p < P) { long n = p / HW; off = n * a.out_sn + (p - n * HW) * a.out_ld; roff = n * a.res_sn + (p - n * HW) * a.res_ld; }
        rowoff[tid] = off;
        resoff[tid] = roff;
    }

    // split-K: blockIdx.z owns a contiguous range of TAPS; partial sums are combined with atomics
    const int nchunks = a.Ktot / BK;
    const int taps = a.KS * a.KS;
    const int tper = (taps + a.splitk - 1) / a.splitk;
    const int tap0 = blockIdx.z * tper;
    const int tap1 = (tap0 + tper < taps) ? tap0 + tper : taps;
    const int it0 = 0;
    const int it1 = tap0 < tap1 ? (tap1 - tap0) * nchunks : 0;
    ConvIter<A_PER, B_PER> I;
    if (it1 > 0) iter_set_tap<A_PER, B_PER, BN>(I, a, tap0, R, n0, tid, kq, pn, py, px, pv);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    TileRegs<A_PER, B_PER> t[CPS];
    const int nsteps = (it1 + CPS - 1) / CPS;
#pragma unroll
    for (int h = 0; h < CPS; h++) {
        if (h < it1) t[h] = iter_load<A_PER, B_PER, BN>(I, a, R, n0, tid, kq, pn, py, px, pv);
        else t[h] = TileRegs<A_PER, B_PER>{};
    }
    for (int st = 0; st < nsteps; st++) {
        unsigned char* abuf = smem + (st & 1) * (A_BYTES + B_BYTES);
        unsigned char* bbuf = abuf + A_BYTES;
        // registers -> LDS.  One barrier per step: it also orders the fragment reads of this buffer two steps ago.
#pragma unroll
        for (int h = 0; h < CPS; h++) {
#pragma unroll
            for (int i = 0; i < A_PER; i++) {
                int r = (tid >> 2) + 64 * i;
                if (r < BM) {
                    if (NS == 0) *reinterpret_cast<float4*>(abuf + (r * PF + h * BK + kq * 4) * 4) = t[h].a[i];
                    else split_store<NS == 0 ? 1 : NS>(reinterpret_cast<unsigned short*>(abuf) + r * PH + h * BK + kq * 4, BM * PH, t[h].a[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < B_PER; i++) {
                int r = (tid >> 2) + 64 * i;
                if (r < BN) {
                    if (NS == 0) *reinterpret_cast<float4*>(bbuf + (r * PF + h * BK + kq * 4) * 4) = t[h].b[i];
                    else split_store<NS == 0 ? 1 : NS>(reinterpret_cast<unsigned short*>(bbuf) + r * PH + h * BK + kq * 4, BN * PH, t[h].b[i]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < CPS; h++) {
            if ((st + 1) * CPS + h < it1) t[h] = iter_load<A_PER, B_PER, BN>(I, a, R, n0, tid, kq, pn, py, px, pv);
            else t[h] = TileRegs<A_PER, B_PER>{};       // K tail: zero tile
        }
        if (NS == 0) {
            const float* As = reinterpret_cast<const float*>(abuf);
            const float* Bs = reinterpret_cast<const float*>(bbuf);
#pragma unroll
            for (int kk = 0; kk < 2 * CPS; kk++) {
                float4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; i++)
                    fa[i] = *reinterpret_cast<const float4*>(&As[(wm * 32 * TM + i * 32 + (lane & 31)) * PF + kk * 8 + (lane >> 5) * 4]);
#pragma unroll
                for (int j = 0; j < TN; j++)
                    fb[j] = *reinterpret_cast<const float4*>(&Bs[(wn * 32 * TN + j * 32 + (lane & 31)) * PF + kk * 8 + (lane >> 5) * 4]);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                    }
            }
        } else {
            constexpr int NP = NS == 0 ? 1 : NS;
            const unsigned short* Ah = reinterpret_cast<const unsigned short*>(abuf);
            const unsigned short* Bh = reinterpret_cast<const unsigned short*>(bbuf);
#pragma unroll
            for (int h = 0; h < CPS; h++) {
                bf16x8 fa[TM][NP], fb[TN][NP];
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int pl = 0; pl < NP; pl++)
                        fa[i][pl] = *reinterpret_cast<const bf16x8*>(&Ah[pl * BM * PH + (wm * 32 * TM + i * 32 + (lane & 31)) * PH + h * BK + (lane >> 5) * 8]);
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int pl = 0; pl < NP; pl++)
                        fb[j][pl] = *reinterpret_cast<const bf16x8*>(&Bh[pl * BN * PH + (wn * 32 * TN + j * 32 + (lane & 31)) * PH + h * BK + (lane >> 5) * 8]);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) {
                        if (NP == 3) {   // smallest terms first
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][NP - 1], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][NP - 1], fb[j][0], acc[i][j], 0, 0, 0);
                        }
                        if (NP >= 2) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][NP >= 2 ? 1 : 0], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][NP >= 2 ? 1 : 0], fb[j][0], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
                    }
            }
        }
    }
    __syncthreads();   // rowoff visibility when the K range is empty

    // epilogue: D fragment map col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < TN; j++) {
        int col = n0 + wn * 32 * TN + j * 32 + (lane & 31);
        if (col >= a.Cout) continue;
        float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int row = wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                long off = rowoff[row];
                if (off < 0) continue;
                float v = acc[i][j][r] + bv;
                if (a.res) v += a.res[resoff[row] + col];
                if (a.act == 1) v = tanhf(v);
                else if (a.act == 2) v = fmaxf(v, 0.f);
                else if (a.act == 3) v = v > 0.f ? v : 0.2f * v;
                if (a.mask) {
                    const float m = a.mask[off + col];
                    if (a.seed_ref) { const float d = m - a.seed_ref[off + col]; v += d > 0.f ? a.seed_w : (d < 0.f ? -a.seed_w : 0.f); }
                    v = m > 0.f ? v : 0.f;
                }
                float* o = a.out + off + col;
                if (a.splitk > 1) { if (a.split_stride) o[blockIdx.z * a.split_stride] = v; else atomicAdd(o, v); }
                else { if (a.accumulate) v += *o; *o = v; }
            }
        }
    }
}

// fixed-order sum of the split-K slabs of a forward conv: out[p][c] = sum_z slab_z[p][c]   (deterministic, unlike atomics)
__global__ __launch_bounds__(256) void k_split_reduce(const float* scr, long stride, int splits, int ldc, int HW, long P, int C, float* out, long out_sn, int out_ld, const float* bias, int act,
                                                      const float* res, long res_sn, int res_ld, int accumulate) {
    const int C4 = ldc >> 2;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < P * C4; i += (long)gridDim.x * 256) {
        long p = i / C4; int c = (int)(i - p * C4) * 4;
        const float* q = scr + p * ldc + c;
        float4 v = *reinterpret_cast<const float4*>(q);
        for (int z = 1; z < splits; z++) { float4 w = *reinterpret_cast<const float4*>(q + z * stride); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        if (bias) { v.x += bias[c < C ? c : 0]; v.y += bias[c + 1 < C ? c + 1 : 0]; v.z += bias[c + 2 < C ? c + 2 : 0]; v.w += bias[c + 3 < C ? c + 3 : 0]; }
        long n = p / HW;
        if (res) {
            const float* rp = res + n * res_sn + (p - n * HW) * (long)res_ld + c;
            v.x += rp[0]; if (c + 1 < C) v.y += rp[1]; if (c + 2 < C) v.z += rp[2]; if (c + 3 < C) v.w += rp[3];
        }
        if (act == 1) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
        else if (act == 3) { v.x = v.x > 0.f ? v.x : 0.2f * v.x; v.y = v.y > 0.f ? v.y : 0.2f * v.y; v.z = v.z > 0.f ? v.z : 0.2f * v.z; v.w = v.w > 0.f ? v.w : 0.2f * v.w; }
        float* o = out + n * out_sn + (p - n * HW) * (long)out_ld + c;
        if (accumulate) { v.x += o[0]; if (c + 1 < C) v.y += o[1]; if (c + 2 < C) v.z += o[2]; if (c + 3 < C) v.w += o[3]; }      // deterministic split-K of a dgrad: out += sum of the slabs
        if (c + 4 <= C) *reinterpret_cast<float4*>(o) = v;
        else { if (c < C) o[0] = v.x; if (c + 1 < C) o[1] = v.y; if (c + 2 < C) o[2] = v.z; }
    }
}

// ... followed by avg_pool2d(2) (ConvArgs.avgpool of a K-split launch: the roll-out's conv -> pool -> affine -> LeakyReLU chains): one thread = one POOLED (pixel, channel quad);
// the four pixels of its window are each summed over the slabs in slab order, then averaged; bias / residual / activation at the pooled size
__global__ __launch_bounds__(256) void k_split_reduce_pool(const float* scr, long stride, int splits, int ldc, int N, int H, int W, int C, float* out, long out_sn, int out_ld,
                                                           const float* bias, int act, const float* res, long res_sn, int res_ld) {
    const int C4 = ldc >> 2, OH = H >> 1, OW = W >> 1;
    const long total = (long)N * OH * OW * C4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long p = i / C4; const int c = (int)(i - p * C4) * 4;
        const int n = (int)(p / ((long)OH * OW)); p -= (long)n * OH * OW;
        const int y = (int)(p / OW), x = (int)(p - (long)y * OW);
        float4 s4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float* q = scr + (((long)n * H + 2 * y + (j >> 1)) * W + 2 * x + (j & 1)) * ldc + c;
            float4 v = *reinterpret_cast<const float4*>(q);
            for (int z = 1; z < splits; z++) { const float4 w = *reinterpret_cast<const float4*>(q + z * stride); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
            s4[j] = v;
        }
        float4 v;
        v.x = 0.25f * ((s4[0].x + s4[1].x) + (s4[2].x + s4[3].x)); v.y = 0.25f * ((s4[0].y + s4[1].y) + (s4[2].y + s4[3].y));
        v.z = 0.25f * ((s4[0].z + s4[1].z) + (s4[2].z + s4[3].z)); v.w = 0.25f * ((s4[0].w + s4[1].w) + (s4[2].w + s4[3].w));
        if (bias) { v.x += bias[c < C ? c : 0]; v.y += bias[c + 1 < C ? c + 1 : 0]; v.z += bias[c + 2 < C ? c + 2 : 0]; v.w += bias[c + 3 < C ? c + 3 : 0]; }
        const long opix = (long)y * OW + x;
        if (res) {
            const float* rp = res + n * res_sn + opix * (long)res_ld + c;
            v.x += rp[0]; if (c + 1 < C) v.y += rp[1]; if (c + 2 < C) v.z += rp[2]; if (c + 3 < C) v.w += rp[3];
        }
        if (act == 3) { v.x = v.x > 0.f ? v.x : 0.2f * v.x; v.y = v.y > 0.f ? v.y : 0.2f * v.y; v.z = v.z > 0.f ? v.z : 0.2f * v.z; v.w = v.w > 0.f ? v.w : 0.2f * v.w; }
        float* o = out + n * out_sn + opix * (long)out_ld + c;
        if (c + 4 <= C) *reinterpret_cast<float4*>(o) = v;
        else { if (c < C) o[0] = v.x; if (c + 1 < C) o[1] = v.y; if (c + 2 < C) o[2] = v.z; }
    }
}

// the same for the gate convolution of a roll-out ConvLSTM cell (ConvArgs.lstm): the four gate quads of (pixel, channel quad) are summed over the slabs in the same fixed
// order, then the cell update of k_map<FLstmFwd> is applied in place -- same expressions (results within 2 ulp: fma contraction) -- and the gate tensor itself is never written (only a backward pass
// would read it).  One launch instead of k_split_reduce + the point-wise kernel.
__global__ __launch_bounds__(256) void k_split_reduce_lstm(const float* scr, long stride, int splits, int ldc, int HW, long P, const float* bias, LstmFuse f) {
    // four consecutive lanes own one (pixel, channel quad): lane k sums gate k's quad over the slabs (as many threads in flight as k_split_reduce has), a 4 x 4 transpose
    // through shuffles hands lane k the four gates of channel c + k, and every lane updates one channel
    const int C = f.C, C4 = C >> 2, lane = threadIdx.x & 63, k = lane & 3;
    const long items = P * C4;
    // (wave-uniform trip count: a wave owns 16 consecutive items per trip and every lane takes part in the shuffles; lanes past the end are clamped and do not store)
    for (long w0 = (blockIdx.x * 256L + (threadIdx.x & ~63)) >> 2; w0 < items; w0 += (long)gridDim.x * 64) {
        const long i0 = w0 + (lane >> 2);
        const long i = i0 < items ? i0 : items - 1;
        const long p = i / C4; const int c = (int)(i - p * C4) * 4;
        const float* q = scr + p * ldc + k * C + c;
        float4 v = *reinterpret_cast<const float4*>(q);
        for (int z = 1; z < splits; z++) { const float4 w = *reinterpret_cast<const float4*>(q + z * stride); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + k * C + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        // lane k holds gate k of channels c .. c + 3 -> lane k gets gates 0 .. 3 of channel c + k
        const int base = lane & ~3;
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float a0 = __shfl(v.x, base + j), a1 = __shfl(v.y, base + j), a2 = __shfl(v.z, base + j), a3 = __shfl(v.w, base + j);
            g[j] = k == 0 ? a0 : (k == 1 ? a1 : (k == 2 ? a2 : a3));
        }
        if (i0 < items) {
            const long n = p / HW, pp = p - n * HW;
            const int ch = c + k;
            const float cp = f.cprev[n * f.cprev_sn + pp * f.cprev_ld + ch];
            auto sg = [](float t) { return 1.f / (1.f + expf(-t)); };      // (pointwise.hip: sigm)
            const float cc = sg(g[1]) * cp + sg(g[0]) * tanhf(g[3]);
            const float hh = sg(g[2]) * tanhf(cc);
            f.c[n * f.c_sn + pp * f.c_ld + ch] = cc;
            f.h[n * f.h_sn + pp * f.h_ld + ch] = hh;
            if (f.hb) f.hb[n * f.hb_sn + pp * f.hb_ld + ch] = hh * f.scale[ch] + f.shift[ch];
        }
    }
}

// the same with the BatchNorm partial sums of the reduced tensor (ConvArgs.stats): a workgroup owns PPB consecutive pixels, thread = (pixel row tid / C4, channel quad
// tid % C4) with C4 | 256, so that every thread keeps one channel quad; the pixel rows are folded through LDS and workgroup b writes stats[(b * stats_ld + c) * 2 + {0, 1}]
// -- the layout k_bn_finalize_tiles reads, with "tiles" = workgroups.  (Split launches are exactly the under-filled ones -- R's 16x16 / 32x32 maps, E / A on one time
// step's frames -- where a separate statistics pass costs as much as the convolution.)
__global__ __launch_bounds__(256) void k_split_reduce_stats(const float* scr, long stride, int splits, int ldc, int HW, long P, int C, float* out, long out_sn, int out_ld,
                                                            int ppb, float* stats, int stats_ld) {
    __shared__ float sh[256 * 8];
    const int C4 = ldc >> 2, rows = 256 / C4;
    const int tid = threadIdx.x, pr = tid / C4, c = (tid - pr * C4) * 4;
    const long p0 = (long)blockIdx.x * ppb, p1 = p0 + ppb < P ? p0 + ppb : P;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (pr < rows)
        for (long p = p0 + pr; p < p1; p += rows) {
            const float* q = scr + p * ldc + c;
            float4 v = *reinterpret_cast<const float4*>(q);
            for (int z = 1; z < splits; z++) { float4 w = *reinterpret_cast<const float4*>(q + z * stride); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
            const long n = p / HW;
            float* o = out + n * out_sn + (p - n * HW) * (long)out_ld + c;
            if (c + 4 <= C) *reinterpret_cast<float4*>(o) = v;
            else { if (c < C) o[0] = v.x; if (c + 1 < C) o[1] = v.y; if (c + 2 < C) o[2] = v.z; }
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            s[4] = fmaf(v.x, v.x, s[4]); s[5] = fmaf(v.y, v.y, s[5]); s[6] = fmaf(v.z, v.z, s[6]); s[7] = fmaf(v.w, v.w, s[7]);
        }
    for (int e = 0; e < 8; e++) sh[tid * 8 + e] = s[e];
    __syncthreads();
    int top = 1;
    while (top < rows) top <<= 1;
    for (int half = top >> 1; half >= 1; half >>= 1) {
        if (pr < half && pr + half < rows)
            for (int e = 0; e < 8; e++) sh[tid * 8 + e] += sh[((pr + half) * C4 + (tid - pr * C4)) * 8 + e];
        __syncthreads();
    }
    if (pr == 0)
        for (int e = 0; e < 4; e++)
            if (c + e < C) { float* o = stats + ((long)blockIdx.x * stats_ld + c + e) * 2; o[0] = sh[tid * 8 + e]; o[1] = sh[tid * 8 + 4 + e]; }
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad.  GEMM-M = output channels o, GEMM-N = concatenated input channels k (one tap per workgroup), reduction over
// pixels in steps of 16; LDS tiles are [pixel][channel] so global->LDS is a straight float4 copy and the MFMA operand
// reads are stride-1 ds_read_b32 (conflict-free).  grid.z = taps x slabs (split of the pixel reduction).
// ------------------------------------------------------------------------------------------------------------------
constexpr int BP = 16;

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256) void k_conv_wgrad(WgradArgs a) {
    constexpr int BMo = 32 * TM * WM, BNk = 32 * TN * WN;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BNk == 128, "loader assumes a 128-wide k tile");
    constexpr int Y_PER = (BP * BMo / 4 + 255) / 256;
    __shared__ float Ys[2][BP * BMo];
    __shared__ float Xs[2][BP * BNk];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int HW = a.H * a.W;
    const long P = (long)a.N * HW;
    const int k0 = blockIdx.x * BNk, o0 = blockIdx.y * BMo;
    const int tap = blockIdx.z / a.slabs, slab = blockIdx.z - tap * a.slabs;
    const int R = a.KS >> 1;
    const int dy = tap / a.KS - R, dx = tap % a.KS - R;
    const long per = ((P + a.slabs - 1) / a.slabs + BP - 1) / BP * BP;
    const long ps = per * slab, pe = (ps + per < P) ? ps + per : P;

    // X loader: chunk (16 channels) is wave-uniform: chunks {wave, wave+4}; within: pixel = (lane>>2), q = lane&3
    SegRef sg[2];
    bool sgok[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        int k = k0 + (wave + 4 * i) * BK;
        sgok[i] = k < a.Ktot;
        sg[i] = find_seg(a.src, a.nsrc, sgok[i] ? k : 0);
    }
    const int xp = lane >> 2, xq = lane & 3;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    float4 rx[2], ry[Y_PER];
    auto gload = [&](long pbase) {
        // X tile
        long p = pbase + xp;
        bool pvld = p < pe;
        long pp = pvld ? p : 0;
        int n = (int)(pp / HW);
        int rem = (int)(pp - (long)n * HW);
        int y = rem / a.W, x = rem - y * a.W;
        int yy = y + dy, xx = x + dx;
        bool ok = pvld && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
#pragma unroll
        for (int i = 0; i < 2; i++) rx[i] = load_src4(sg[i], xq, ok && sgok[i], n, yy, xx, a.W);
        // dY tile: BP pixels x BMo channels
#pragma unroll
        for (int i = 0; i < Y_PER; i++) {
            int idx = tid + 256 * i;
            int pr = idx / (BMo / 4), q = idx - pr * (BMo / 4);
            ry[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pr < BP) {
                long p2 = pbase + pr;
                int c = o0 + q * 4;
                if (p2 < pe && c < a.Cout) {
                    long n2 = p2 / HW;
                    const float* qy = a.dy + n2 * a.dy_sn + (p2 - n2 * HW) * a.dy_ld + c;
                    ry[i] = load4_masked(qy, c, a.Cout);
                }
            }
        }
    };

    if (ps < pe) gload(ps);
    int buf = 0;
    for (long pb = ps; pb < pe; pb += BP, buf ^= 1) {   // LDS double buffer: one barrier per 16-pixel step
#pragma unroll
        for (int i = 0; i < 2; i++) *reinterpret_cast<float4*>(&Xs[buf][xp * BNk + (wave + 4 * i) * BK + xq * 4]) = rx[i];
#pragma unroll
        for (int i = 0; i < Y_PER; i++) {
            int idx = tid + 256 * i;
            int pr = idx / (BMo / 4), q = idx - pr * (BMo / 4);
            if (pr < BP) *reinterpret_cast<float4*>(&Ys[buf][pr * BMo + q * 4]) = ry[i];
        }
        __syncthreads();
        if (pb + BP < pe) gload(pb + BP);
#pragma unroll
        for (int s = 0; s < BP / 2; s++) {
            float fa[TM], fb[TN];
            int prow = 2 * s + (lane >> 5);
#pragma unroll
            for (int i = 0; i < TM; i++) fa[i] = Ys[buf][prow * BMo + wm * 32 * TM + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < TN; j++) fb[j] = Xs[buf][prow * BNk + wn * 32 * TN + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }

#pragma unroll
    for (int j = 0; j < TN; j++) {
        int k = k0 + wn * 32 * TN + j * 32 + (lane & 31);
        if (k >= a.Ktot) continue;
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int o = o0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (o >= a.Cout) continue;
                float* d = WGRAD_DST(a, slab) + ((long)tap * a.Cout_pad + o) * a.Ktot + k;
                if (a.slabs > 1) atomicAdd(d, acc[i][j][r]);
                else *d += acc[i][j][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad, tile-resident (3x3 layers with >= 33 output or > 64 input channels: R's ConvLSTM gates, the 64/128-channel
// residual blocks of E / A / D).  The one-tap-per-workgroup kernel above re-reads dY and X once per tap (9x) and pads K to
// 128; here a workgroup owns a 64(o) x 64(k) weight tile for ALL nine taps: a 4x16 pixel tile of dY and the 6x18 halo tile
// of X are staged in LDS once, every wave keeps nine 32x32 accumulators (one per tap, 144 registers) and walks the pixel
// pairs {(y,x),(y+2,x)} -- the two MFMA k-lanes -- reading one dY fragment and nine shifted X fragments per step.  Row
// pitches are padded by 16 floats so the two k-lanes (two rows apart) land in opposite halves of the LDS banks.
// Workgroups are persistent over spatial tiles (blockIdx.z-strided) with a register prefetch of the next tile, and flush
// their 9x64x64 partial sums once at the end with fp32 atomics.  OS = 2: 64 output channels (waves 2(o) x 2(k));
// OS = 1: 32 output channels (waves 2(k) x 2(pixel halves)).
// ------------------------------------------------------------------------------------------------------------------
constexpr int WT_W = 16, WT_H = 4, WT_KC = 64;
constexpr int WT_HW = WT_W + 2, WT_HH = WT_H + 2;
constexpr int WT_XROW = WT_HW * WT_KC + 16;
constexpr int WT_XLOADS = (WT_HH * WT_HW * (WT_KC / 4) + 255) / 256;   // 7 float4 per thread

template <int OS>
__global__ __launch_bounds__(256) void k_conv_wgrad_tile(WgradArgs a, int tiles_x, int tiles_y) {
    constexpr int OC = 32 * OS;
    constexpr int YROW = WT_W * OC + 16;
    constexpr int YLOADS = WT_H * WT_W * (OC / 4) / 256;               // 4 (OS = 2) or 2 (OS = 1)
    __shared__ float Xh[WT_HH * WT_XROW];
    __shared__ float Yt[WT_H * YROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1;
    const int wm = OS == 2 ? (wave >> 1) : 0;
    const int wp = OS == 2 ? 0 : (wave >> 1);
    const int k0 = blockIdx.x * WT_KC, o0 = blockIdx.y * OC;
    const long ntiles = (long)a.N * tiles_x * tiles_y;

    // loader roles: X -- float4 column q (0..15) is fixed per thread, halo pixel = (tid >> 4) + 16 i
    const int xq = tid & 15;
    const int kx = k0 + (xq >> 2) * BK;
    const bool kok = kx < a.Ktot;
    const SegRef sg = find_seg(a.src, a.nsrc, kok ? kx : 0);
    int xhy[WT_XLOADS], xhx[WT_XLOADS];
#pragma unroll
    for (int i = 0; i < WT_XLOADS; i++) {
        int pix = (tid >> 4) + 16 * i;
        xhy[i] = pix / WT_HW; xhx[i] = pix - xhy[i] * WT_HW;
    }
    // dY: float4 column yq (0..OC/4-1) fixed per thread
    const int yq = tid % (OC / 4), ypix0 = tid / (OC / 4);
    const int yc = o0 + yq * 4;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    float4 rx[WT_XLOADS], ry[YLOADS];
    auto gload = [&](long tile) {
        int n = (int)(tile / (tiles_x * tiles_y));
        int rem = (int)(tile - (long)n * tiles_x * tiles_y);
        int ty = rem / tiles_x;
        int y0 = ty * WT_H, x0 = (rem - ty * tiles_x) * WT_W;
        SegRef s2 = sg;
        const float* dyb = a.dy;
        if (a.group_n > 0) {                       // time-batched launch: (group, sample) addressing
            int grp = n / a.group_n;
            n -= grp * a.group_n;
            s2.p += grp * a.src_gs[sg.idx];
            dyb += grp * a.dy_gs;
        }
#pragma unroll
        for (int i = 0; i < WT_XLOADS; i++) {
            int y = y0 - 1 + xhy[i], x = x0 - 1 + xhx[i];
            bool ok = kok && xhy[i] < WT_HH && y >= 0 && y < a.H && x >= 0 && x < a.W;
            rx[i] = load_src4(s2, xq & 3, ok, n, y, x, a.W);
        }
#pragma unroll
        for (int i = 0; i < YLOADS; i++) {
            int pix = ypix0 + (256 / (OC / 4)) * i;
            int y = y0 + pix / WT_W, x = x0 + (pix & (WT_W - 1));
            ry[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y < a.H && x < a.W && yc < a.Cout) ry[i] = load4_masked(dyb + (long)n * a.dy_sn + ((long)y * a.W + x) * a.dy_ld + yc, yc, a.Cout);
        }
    };

    const int half = lane >> 5;
    const float* ybase = Yt + (2 * half) * YROW + wm * 32 + (lane & 31);
    const float* xbase = Xh + (2 * half) * WT_XROW + wn * 32 + (lane & 31);

    long tile = blockIdx.z;
    if (tile < ntiles) gload(tile);
    for (; tile < ntiles; tile += gridDim.z) {
#pragma unroll
        for (int i = 0; i < WT_XLOADS; i++)
            if (xhy[i] < WT_HH) *reinterpret_cast<float4*>(&Xh[xhy[i] * WT_XROW + xhx[i] * WT_KC + xq * 4]) = rx[i];
#pragma unroll
        for (int i = 0; i < YLOADS; i++) {
            int pix = ypix0 + (256 / (OC / 4)) * i;
            *reinterpret_cast<float4*>(&Yt[(pix / WT_W) * YROW + (pix & (WT_W - 1)) * OC + yq * 4]) = ry[i];
        }
        __syncthreads();
        if (tile + gridDim.z < ntiles) gload(tile + gridDim.z);
        // 32 steps = (2 row pairs) x (16 columns); OS = 1 splits the steps between the two wave pairs
        constexpr int NSTEP = OS == 2 ? 32 : 16;
#pragma unroll 4
        for (int s0 = 0; s0 < NSTEP; s0++) {
            const int s = s0 + wp * 16;
            const int yy = s >> 4, x = s & 15;
            const float fa = ybase[yy * YROW + x * OC];
            const float* xr = xbase + yy * WT_XROW + x * WT_KC;
#pragma unroll
            for (int dy = 0; dy < 3; dy++)
#pragma unroll
                for (int dx = 0; dx < 3; dx++)
                    acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, xr[dy * WT_XROW + dx * WT_KC], acc[dy * 3 + dx], 0, 0, 0);
        }
        __syncthreads();
    }

    const int k = k0 + wn * 32 + (lane & 31);
    if (k < a.Ktot) {
#pragma unroll
        for (int t = 0; t < 9; t++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int o = o0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (o < a.Cout) atomicAdd(WGRAD_DST(a, blockIdx.z) + ((long)t * a.Cout_pad + o) * a.Ktot + k, acc[t][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad for narrow layers (Cout <= 32 and K <= 64: E's 16/32-channel blocks, D's last UpBlock).  The generic kernel's
// 128-wide k tile would be mostly padding there.  Here a workgroup owns a 8x32 pixel tile: dY and the X halo tile are
// staged in LDS ONCE and reused by all taps; each wave owns the taps {w, w+4, w+8} and keeps one 32(o) x 32*KT(k)
// accumulator per tap in registers across many tiles (grid-stride), then flushes with atomics.
// ------------------------------------------------------------------------------------------------------------------
constexpr int STW = 32, STH = 8;
template <int KT>
__global__ __launch_bounds__(256) void k_conv_wgrad_small(WgradArgs a, int tiles_x, int tiles_y) {
    constexpr int KC = 32 * KT;                       // channels staged per pixel
    __shared__ float Xh[(STH + 2) * (STW + 2) * KC];  // halo tile [pixel][KC]   (3x3 or 1x1 only)
    __shared__ float Yt[STH * STW * 32];              // dY tile   [pixel][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int taps = a.KS * a.KS, R = a.KS >> 1, HWD = STW + 2 * R, HHT = STH + 2 * R;
    const ConvSrc s = a.src[0];
    const long ntiles = (long)a.N * tiles_x * tiles_y;
    f32x16 acc[3][KT];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int j = 0; j < KT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][j][r] = 0.f;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int n = (int)(tile / (tiles_x * tiles_y));
        int rem = (int)(tile - (long)n * tiles_x * tiles_y);
        int y0 = (rem / tiles_x) * STH, x0 = (rem % tiles_x) * STW;
        for (int idx = tid; idx < HHT * HWD * (KC / 4); idx += 256) {
            int q = idx % (KC / 4), pix = idx / (KC / 4);
            int hy = pix / HWD, hx = pix - hy * HWD;
            int y = y0 - R + hy, x = x0 - R + hx, c = q * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y >= 0 && y < a.H && x >= 0 && x < a.W && c < s.C) v = load4_masked(s.p + (long)n * s.sn + ((long)y * a.W + x) * s.ld + c, c, s.C);
            *reinterpret_cast<float4*>(&Xh[pix * KC + c]) = v;
        }
        for (int idx = tid; idx < STH * STW * 8; idx += 256) {
            int q = idx & 7, pix = idx >> 3;
            int y = y0 + pix / STW, x = x0 + pix % STW, c = q * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y < a.H && x < a.W && c < a.Cout) v = load4_masked(a.dy + (long)n * a.dy_sn + ((long)y * a.W + x) * a.dy_ld + c, c, a.Cout);
            *reinterpret_cast<float4*>(&Yt[pix * 32 + c]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 3; t++) {
            int tap = wave + 4 * t;
            if (tap < taps) {
                int dy = tap / a.KS, dx = tap - dy * a.KS;
                for (int p = 0; p < STH * STW; p += 2) {
                    int pp = p + (lane >> 5);                 // MFMA k index = pixel
                    int py = pp / STW, px = pp - py * STW;
                    float fa = Yt[pp * 32 + (lane & 31)];     // A[row = o][k = pixel]
                    const float* xr = &Xh[((py + dy) * HWD + px + dx) * KC + (lane & 31)];
#pragma unroll
                    for (int j = 0; j < KT; j++) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, xr[32 * j], acc[t][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 3; t++) {
        int tap = wave + 4 * t;
        if (tap >= taps) continue;
#pragma unroll
        for (int j = 0; j < KT; j++) {
            int k = j * 32 + (lane & 31);
            if (k >= a.Ktot) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int o = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (o < a.Cout) atomicAdd(WGRAD_DST(a, blockIdx.x) + ((long)tap * a.Cout_pad + o) * a.Ktot + k, acc[t][j][r]);
            }
        }
    }
}

}  // namespace

thread_local int g_last_conv_kernel = -1;

int conv_pick_bn(int cout) {
    int best = 128, bestpad = round_up(cout, 128);
    for (int bn : {64, 32}) {
        int pad = round_up(cout, bn);
        if (pad < bestpad) { best = bn; bestpad = pad; }
    }
    return best;
}

int conv_fwd_launch(const ConvArgs& a0, hipStream_t st) {
    ConvArgs a = a0;
    if (a.nsrc < 1 || a.nsrc > CONV_MAX_SRC || (a.KS != 1 && a.KS != 3 && a.KS != 7)) return -1;
    int kt = 0;
    for (int s = 0; s < a.nsrc; s++) {
        if (a.src[s].Cpad != round_up(a.src[s].C, BK) || (a.src[s].ld & 3) || (a.src[s].sn & 3)) return -1;
        kt += a.src[s].Cpad;
    }
    if (kt != a.Ktot || (a.out_ld < a.Cout)) return -1;
    int bn = conv_pick_bn(a.Cout);
    if (a.Cout_pad != round_up(a.Cout, bn)) return -1;
    a.splitk = 1;
    g_last_conv_stats_tiles = 0;
    if (a.avgpool) {      // pooled epilogue: the caller has asked conv_avgpool_ok() -- the latency kernel for split-operand layers, k_conv_narrow otherwise
        if (a.wq && a.precision >= PREC_F16X3) return conv_hx_try(a, st) == 1 ? 0 : -1;
        if (a.KS == 1) return conv1x1_lat_try(a, st) == 1 ? 0 : -1;
        return conv_narrow_fwd_try(a, st) == 1 ? 0 : -1;
    }
    if (a.KS == 1 && a.direct_ok && conv1x1_lat_try(a, st) == 1) { g_last_conv_kernel = CK_FWD_128x32; return 0; }      // tiny 1x1 launches of a roll-out frame
    { int rc = conv_hx_try(a, st); if (rc != 0) return rc < 0 ? rc : 0; }      // split 16-bit operands on the 16-bit matrix pipe (conv_hx.hip)
    if (a.in_s16 || a.out_s16 || a.pool_s16) return -1;                      // pre-split tensors are understood by k_conv_hx only: never hand their bytes to a kernel that reads fp32
    for (int s = 0; s < a.nsrc; s++) if (a.src[s].bn_scale) return -1;        // lazily normalised inputs are understood by k_conv_hx only: the caller must have materialised them
    if (a.pool_out || a.skip_out) return -1;      // fused max-pool / write-less epilogues exist in k_conv_hx only: the caller must not ask the other kernels for them
    const bool generic_only = a.act == 2 || a.mask != nullptr;      // ReLU / masked epilogues exist in k_conv_fwd only (VGG19 perceptual loss)
    if (a.seed_ref && !a.mask) return -1;
    const bool fold_epilogue = a.act == 3 || a.res != nullptr;      // LeakyReLU / residual epilogues (BatchNorm-folded roll-out): k_conv_fwd, k_conv_hx, k_conv_narrow
    if (!generic_only) {
        if (!fold_epilogue && conv_head_fwd_try(a, st) == 1) return 0;      // FinalBlock heads (-> 3 channels) on the 16-bit matrix pipe (conv_head.hip)
        if (!fold_epilogue && conv_head_dgrad_try(a, st) == 1) return 0;    // dgrad of the 7x7 FinalBlock head on the split-bf16 matrix pipe (conv_head.hip)
        if (!fold_epilogue && conv_c4_fwd_try(a, st) == 1) return 0;        // 3-channel input (stem, FinalBlock dgrad): 16x16x4 MFMA, K = one padded pixel
        if (!fold_epilogue && conv_thin_fwd_try(a, st) == 1) return 0;      // 3-channel heads / stem: vector-ALU kernels (conv_thin.hip)
        if (conv_narrow_fwd_try(a, st) == 1) return 0;    // 16/32-channel layers: halo-tile kernel on 16x16x4 MFMA (conv_narrow.hip)
    }
    long P = (long)a.N * a.H * a.W;
    // tile choice: 128-row tiles when they fill the chip (256 CUs x ~3 resident workgroups), otherwise 64x64 tiles; the
    // remaining deficit of accumulating launches (dgrad) is covered by splitting K across blockIdx.z (fp32 atomics).
    long blocks128 = (long)cdiv(P, 128) * (a.Cout_pad / bn);
    bool small = (bn >= 64) && blocks128 < 512;
    int niter = a.KS * a.KS * (a.Ktot / BK);
    a.splitk = 1;
    long blocks = small ? (long)cdiv(P, 64) * (a.Cout_pad / 64) : blocks128;
    bool det_accum = false;      // deterministic mode: the split of an accumulating launch goes through the slabs as well (k_split_reduce adds the old contents)
    if (a.accumulate && a.act == 0 && a.bias == nullptr && !a.mask && !a.res && blocks < 384 && niter >= 16 && a.KS == 3) {
        int want = (int)((512 + blocks - 1) / blocks);
        const int sk = want >= 5 ? 9 : (want >= 2 ? 3 : 1);      // whole taps per slice
        if (!a.deterministic) a.splitk = sk;
        else if (a.split_scratch && sk > 1 && (long)sk * P * round_up(a.Cout, 4) <= a.split_cap) det_accum = true;
    }
    // under-filled forward launches (batch-1 roll-out, R's 16x16 maps): split K over taps into slabs of a scratch buffer and sum them in
    // a fixed order afterwards -- keeps the forward pass bit-reproducible (action indices!) where atomics would not
    a.split_stride = 0;
    float* real_out = a.out; long real_sn = a.out_sn; int real_ld = a.out_ld; const float* real_bias = a.bias; const int real_act = a.act;
    const float* real_res = a.res;
    if ((det_accum || !a.accumulate) && a.split_scratch && !generic_only && a.KS == 3 && (det_accum || (niter >= 18 && blocks <= 256))) {   // (bias / tanh are applied by the reduce)
        int want = (int)((512 + blocks - 1) / blocks);
        int sk = want >= 5 ? 9 : (want >= 2 ? 3 : 1);
        int ldc = round_up(a.Cout, 4);
        if (sk > 1 && (long)sk * P * ldc <= a.split_cap) {
            a.splitk = sk; a.split_stride = P * ldc;
            a.out = a.split_scratch; a.out_sn = (long)a.H * a.W * ldc; a.out_ld = ldc; a.bias = nullptr; a.act = 0; a.res = nullptr; a.accumulate = 0;
        }
    }
    // (the exact-fp32 kernel: the in-loop split-bf16 planes / 32-channel steps of round 1 are no longer instantiated -- conv_hx.hip is the 16-bit path)
#define LAUNCH_CONV(TM_, TN_, WM_, WN_) hipLaunchKernelGGL((k_conv_fwd<TM_, TN_, WM_, WN_, 0, 1>), grid, dim3(256), 0, st, a)
    if (small) {
        dim3 grid(cdiv(P, 64), a.Cout_pad / 64, a.splitk);
        LAUNCH_CONV(1, 1, 2, 2);
        g_last_conv_kernel = CK_FWD_64x64;
    } else {
        dim3 grid(cdiv(P, 128), a.Cout_pad / bn, a.splitk);
        g_last_conv_kernel = bn == 128 ? CK_FWD_128x128 : (bn == 64 ? CK_FWD_128x64 : CK_FWD_128x32);
        if (bn == 128) LAUNCH_CONV(2, 2, 2, 2);
        else if (bn == 64) LAUNCH_CONV(2, 1, 2, 2);
        else hipLaunchKernelGGL((k_conv_fwd<1, 1, 4, 1, 0, 1>), grid, dim3(256), 0, st, a);
    }
#undef LAUNCH_CONV
    if (a.split_stride) conv_split_reduce_launch(a.split_scratch, a.split_stride, a.splitk, a.out_ld, a.H * a.W, P, a.Cout, real_out, real_sn, real_ld, real_bias, real_act, real_res, a.res_sn, a.res_ld, st,
                                                 nullptr, 0, 0, det_accum ? 1 : 0);
    return 0;
}
thread_local int g_last_conv_lstm_fused = 0;
int conv_split_reduce_lstm_launch(const float* scr, long stride, int splits, int ldc, int HW, long P, const float* bias, const LstmFuse& f, hipStream_t st) {
    const long thr = P * f.C;      // four lanes per (pixel, channel quad)
    hipLaunchKernelGGL(k_split_reduce_lstm, dim3((unsigned)(thr < 256L * 1024 ? cdiv(thr, 256) : 1024)), dim3(256), 0, st, scr, stride, splits, ldc, HW, P, bias, f);
    g_last_conv_lstm_fused = 1;
    return 0;
}
int conv_split_reduce_pool_launch(const float* scr, long stride, int splits, int ldc, int N, int H, int W, int C, float* out, long out_sn, int out_ld, const float* bias, int act,
                                  const float* res, long res_sn, int res_ld, hipStream_t st) {
    const long items = (long)N * (H / 2) * (W / 2) * (ldc >> 2);
    hipLaunchKernelGGL(k_split_reduce_pool, dim3((unsigned)(items < 256L * 1024 ? cdiv(items, 256) : 1024)), dim3(256), 0, st, scr, stride, splits, ldc, N, H, W, C, out, out_sn, out_ld, bias, act,
                       res, res_sn, res_ld);
    return 0;
}
int conv_split_reduce_launch(const float* scr, long stride, int splits, int ldc, int HW, long P, int C, float* out, long out_sn, int out_ld, const float* bias, int act,
                             const float* res, long res_sn, int res_ld, hipStream_t st, float* stats, int stats_ld, long stats_cap_tiles, int accumulate) {
    const int C4 = ldc >> 2;
    if (stats && !accumulate && !bias && !act && !res && C4 <= 256 && 256 % C4 == 0) {      // BatchNorm partial sums of the reduced tensor (conv -> BatchNorm chains carry no bias / activation)
        const int rows = 256 / C4;
        long ppb = rows * 4;                                                  // >= 4 pixels per thread ...
        while (cdiv(P, ppb) > 512) ppb *= 2;                                  // ... and at most 512 workgroups
        const long nb = cdiv(P, ppb);
        if (nb <= stats_cap_tiles) {
            hipLaunchKernelGGL(k_split_reduce_stats, dim3((unsigned)nb), dim3(256), 0, st, scr, stride, splits, ldc, HW, P, C, out, out_sn, out_ld, (int)ppb, stats, stats_ld);
            g_last_conv_stats_tiles = (int)nb;
            return 0;
        }
    }
    long items = P * (ldc >> 2);
    hipLaunchKernelGGL(k_split_reduce, dim3((unsigned)(items < 256L * 1024 ? cdiv(items, 256) : 1024)), dim3(256), 0, st, scr, stride, splits, ldc, HW, P, C, out, out_sn, out_ld, bias, act,
                       res, res_sn, res_ld, accumulate);
    return 0;
}

// ---- bit-reproducible weight gradients (WgradArgs.det_slab) ----
namespace {
// dwp[i] += sum over the pixel splits' copies, in a FIXED order: wave w of a workgroup sums the copies s = w, w + 4, ... of 64 float4 columns (four loads in flight), the four
// partial sums meet in LDS in wave order.  (The first form walked all copies with one dependent load per step from cdiv(stride, 1024) workgroups -- 18 for a 64 x 32 layer with its
// 256 copies: 50 us per launch, 9 ms of the serialised step.)
__global__ __launch_bounds__(256) void k_wgrad_det_reduce(float* dwp, const float* slab, int splits, long stride) {
    __shared__ float4 sh[3][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (long i0 = (long)blockIdx.x * 256; i0 < stride; i0 += (long)gridDim.x * 256) {      // (workgroup-uniform trip count: barriers inside)
        const long i = i0 + 4 * lane;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < stride) {
            int s = w;
            for (; s + 12 < splits; s += 16) {
                const float4 a0 = *reinterpret_cast<const float4*>(slab + (long)s * stride + i), a1 = *reinterpret_cast<const float4*>(slab + (long)(s + 4) * stride + i);
                const float4 a2 = *reinterpret_cast<const float4*>(slab + (long)(s + 8) * stride + i), a3 = *reinterpret_cast<const float4*>(slab + (long)(s + 12) * stride + i);
                v.x += a0.x; v.y += a0.y; v.z += a0.z; v.w += a0.w; v.x += a1.x; v.y += a1.y; v.z += a1.z; v.w += a1.w;
                v.x += a2.x; v.y += a2.y; v.z += a2.z; v.w += a2.w; v.x += a3.x; v.y += a3.y; v.z += a3.z; v.w += a3.w;
            }
            for (; s < splits; s += 4) { const float4 a0 = *reinterpret_cast<const float4*>(slab + (long)s * stride + i); v.x += a0.x; v.y += a0.y; v.z += a0.z; v.w += a0.w; }
        }
        if (w > 0) sh[w - 1][lane] = v;
        __syncthreads();
        if (w == 0 && i < stride) {
            for (int k = 0; k < 3; k++) { const float4 u = sh[k][lane]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
            float4 o = *reinterpret_cast<float4*>(dwp + i);
            o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
            *reinterpret_cast<float4*>(dwp + i) = o;
        }
        __syncthreads();
    }
}
}  // namespace
long wgrad_det_begin(WgradArgs& a, long splits, hipStream_t st) {
    a.det_stride = ((long)a.KS * a.KS * a.Cout_pad * a.Ktot + 3) / 4 * 4;
    if (a.det_stride <= 0 || a.det_stride > a.det_cap) return 0;
    const long fit = a.det_cap / a.det_stride;
    if (splits > fit) splits = fit;
    if (splits < 1) splits = 1;
    hipMemsetAsync(a.det_slab, 0, sizeof(float) * (size_t)splits * a.det_stride, st);
    return splits;
}
int wgrad_det_end(const WgradArgs& a, long splits, hipStream_t st) {
    const long blocks = cdiv(a.det_stride, 256);
    hipLaunchKernelGGL(k_wgrad_det_reduce, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, st, a.dwp, (const float*)a.det_slab, (int)splits, a.det_stride);
    return 0;
}

static int conv_wgrad_launch1(const WgradArgs& a0, hipStream_t st, bool dry);   // dry: only report the kernel (g_last_conv_kernel)
int conv_wgrad_launch(const WgradArgs& a0, hipStream_t st) {
    if (a0.group_n <= 0 || a0.N <= a0.group_n) return conv_wgrad_launch1(a0, st, false);
    if (conv_wgrad_launch1(a0, st, true) == 0 && (g_last_conv_kernel == CK_WGRAD_TILE || g_last_conv_kernel == CK_WGRAD_HX || g_last_wgrad_grouped)) return conv_wgrad_launch1(a0, st, false);
    // time-batched arguments on a kernel without (group, sample) addressing: one launch per group
    for (int g = 0; g * a0.group_n < a0.N; g++) {
        WgradArgs a = a0;
        a.N = a0.group_n; a.group_n = 0;
        for (int s = 0; s < a0.nsrc; s++) a.src[s].p = a0.src[s].p + g * a0.src_gs[s];
        a.dy = a0.dy + g * a0.dy_gs;
        int rc = conv_wgrad_launch1(a, st, false);
        if (rc) return rc;
    }
    return 0;
}
static int conv_wgrad_launch1(const WgradArgs& a0, hipStream_t st, bool dry) {
    WgradArgs a = a0;
    if (a.nsrc < 1 || a.nsrc > CONV_MAX_SRC) return -1;
    int bn = conv_pick_bn(a.Cout);
    if (a.Cout_pad != round_up(a.Cout, bn)) return -1;
    if (a.group_n > 0 && a.N <= a.group_n) a.group_n = 0;
    if (conv_hx_wgrad_try(a, st, dry) == 1) return 0;       // wide 3x3 layers: split bf16 on the 16-bit matrix pipe (conv_hx.hip)
    if (a.dy_s16) return -1;                                // a pre-split dY is understood by k_wgrad_hx only
    for (int s = 0; s < a.nsrc; s++) if (a.src[s].bn_scale) return -1;        // lazily normalised inputs: k_wgrad_hx only
    g_last_wgrad_grouped = 0;
    { int rc = conv_stream_wgrad_try(a, st, dry); if (rc != 0) return rc < 0 ? rc : 0; }      // 1x1 layers: streaming kernel, both operands straight into the fp32 MFMA (conv_stream.hip)
    { int rc = conv_head_wgrad_try(a, st, dry); if (rc != 0) return rc < 0 ? rc : 0; }        // 7x7 FinalBlock head: split bf16, taps on the M side (conv_stream.hip)
    if (conv_c4_wgrad_try(a, st, dry) == 1) return 0;       // 3-channel side: 16x16x4 MFMA (conv_narrow.hip)
    if (conv_thin_wgrad_try(a, st, dry) == 1) return 0;
    if (conv_narrow_wgrad_try(a, st, dry) == 1) return 0;   // 16-channel sides: 16x16x4 MFMA (conv_narrow.hip)
    long P = (long)a.N * a.H * a.W;
    int taps = a.KS * a.KS;
    if (a.nsrc == 1 && !a.src[0].bcast && a.Cout <= 32 && a.Ktot <= 64 && a.KS <= 3 && P >= 4096 && !(a.Ktot > 32 && a.KS == 3)) {   // narrow layers (K = 64, 3x3: the tile-resident kernel below)
        int tx = cdiv(a.W, STW), ty = cdiv(a.H, STH);
        long ntiles = (long)a.N * tx * ty;
        int grid = (int)(ntiles < 512 ? ntiles : 512);
        g_last_conv_kernel = CK_WGRAD_SMALL;
        if (dry) return 0;
        if (a.det_slab) { grid = (int)wgrad_det_begin(a, grid, st); if (grid <= 0) return -1; }
        if (a.Ktot <= 32) hipLaunchKernelGGL((k_conv_wgrad_small<1>), dim3(grid), dim3(256), 0, st, a, tx, ty);
        else hipLaunchKernelGGL((k_conv_wgrad_small<2>), dim3(grid), dim3(256), 0, st, a, tx, ty);
        if (a.det_slab) wgrad_det_end(a, grid, st);
        g_last_conv_kernel = CK_WGRAD_SMALL;
        return 0;
    }
    if (a.KS == 3 && a.W >= 8 && a.H >= 2) {
        int tx = cdiv(a.W, WT_W), ty = cdiv(a.H, WT_H);
        long ntiles = (long)a.N * tx * ty;
        int kt = cdiv(a.Ktot, WT_KC);
        bool o32 = a.Cout <= 32;
        int ot = o32 ? 1 : cdiv(a.Cout, 64);
        const int tile_blocks = 256;   // one persistent workgroup per CU: measured best inside the training step (the BPTT chain shares the chip)
        long g = tile_blocks / ((long)kt * ot);
        if (g < 1) g = 1;
        if (g > ntiles) g = ntiles;
        g_last_conv_kernel = CK_WGRAD_TILE;
        if (dry) return 0;
        if (a.det_slab) { g = wgrad_det_begin(a, g, st); if (g <= 0) return -1; }
        dim3 grid(kt, ot, (unsigned)g);
        if (o32) hipLaunchKernelGGL((k_conv_wgrad_tile<1>), grid, dim3(256), 0, st, a, tx, ty);
        else hipLaunchKernelGGL((k_conv_wgrad_tile<2>), grid, dim3(256), 0, st, a, tx, ty);
        if (a.det_slab) wgrad_det_end(a, g, st);
        g_last_conv_kernel = CK_WGRAD_TILE;
        return 0;
    }
    int ktiles = cdiv(a.Ktot, 128);
    int bmo = a.Cout_pad >= 128 ? 128 : (a.Cout_pad >= 64 ? 64 : 32);
    if (a.Cout_pad % bmo) bmo = 32;
    int otiles = a.Cout_pad / bmo;
    long blocks = (long)ktiles * otiles * taps;
    if (a.slabs <= 0) {
        long want = (1024 + blocks - 1) / blocks;
        long maxs = P / 512 > 0 ? P / 512 : 1;
        a.slabs = (int)(want < maxs ? want : maxs);
        if (a.slabs < 1) a.slabs = 1;
    }
    g_last_conv_kernel = bmo == 128 ? CK_WGRAD_128 : (bmo == 64 ? CK_WGRAD_64 : CK_WGRAD_32);
    if (dry) return 0;
    if (a.det_slab && a.slabs > 1) { a.slabs = (int)wgrad_det_begin(a, a.slabs, st); if (a.slabs <= 0) return -1; }
    const bool det_slabs = a.det_slab && a.slabs > 1;
    if (!det_slabs) a.det_slab = nullptr;      // one slab: the workgroup adds to dwp directly (plain read-modify-write, no atomics)
    dim3 grid(ktiles, otiles, taps * a.slabs);
    if (bmo == 128) hipLaunchKernelGGL((k_conv_wgrad<2, 2, 2, 2>), grid, dim3(256), 0, st, a);
    else if (bmo == 64) hipLaunchKernelGGL((k_conv_wgrad<1, 2, 2, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_conv_wgrad<1, 1, 1, 4>), grid, dim3(256), 0, st, a);
    if (det_slabs) wgrad_det_end(a, a.slabs, st);
    return 0;
}
