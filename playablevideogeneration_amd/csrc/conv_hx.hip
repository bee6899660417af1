// 3x3 convolution on the CDNA4 16-bit matrix pipe with fp32-class accuracy: split operands (hi + lo halves), fp32 accumulation.
//
// Why: the CADDY step is FP32-compute-bound (conv arithmetic intensity ~122 FLOP/B, SURVEY.md section 7 hard part 1); the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32) peaks at 157 TFLOP/s, the 16-bit MFMA (v_mfma_f32_32x32x16_{f16,bf16}) at ~2.5 PFLOP/s.  A fp32 value x is split
// as x = hi + lo with hi = round16(x), lo = round16(x - hi); the product a*b is taken as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with fp32
// accumulation (3 MFMAs per fp32 product -> ~830 TFLOP/s effective peak).  f16 halves carry 11 + 11 mantissa bits (error ~2^-22, "fp32
// re-ordering class" -- measured indistinguishable from a fp32 summation-order change over 15 closed-loop steps);
// bf16 halves carry 8 + 8 bits with the full fp32 exponent range (error ~2^-16, used for the gradient operands whose magnitude is
// unbounded below).  NPL = 1 keeps only hi*hi (plain 16-bit operands; a study variant for the frozen VGG19 loss network -- it fails the parity bounds).
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
// 2x2 max-pool written by the epilogue of the layer in front of it; dgrad with the fused ReLU mask / L1 seed epilogue).
#include "conv_hx_common.h"
#include "pack.h"
#include "pack_elems.h"
#include <cstdlib>

namespace {

// ---- S16 activation format (common.h): helpers of the SO epilogue / S16 mask reads ----
template <typename T> __device__ __forceinline__ unsigned s16_bits(T h) { unsigned short s; __builtin_memcpy(&s, &h, 2); return s; }
// value -> dword to store by this lane: the lanes (c, c + 1) of a channel pair exchange halves, the even lane stores [hi(c) | hi(c + 1)] into the chunk's hi block, the odd
// lane [lo(c - 1) | lo(c)] into its lo block -- one dword store per lane, as for fp32 output.  Must be executed by both lanes of a pair.
template <typename T> __device__ __forceinline__ unsigned s16_pair(float v, int lane) {
    const T hi = (T)v;
    const T lo = (T)(v - (float)hi);
    const unsigned u = s16_bits(hi) | (s16_bits(lo) << 16);
    const unsigned w = (unsigned)__builtin_amdgcn_mov_dpp((int)u, 0xB1 /* quad_perm [1, 0, 3, 2] */, 0xF, 0xF, true);
    return (lane & 1) ? ((w >> 16) | (u & 0xffff0000u)) : ((u & 0xffffu) | (w << 16));
}
// dword slot of channel `col` inside its pixel row for the store above: chunk base + (c >> 1), odd lanes 16 dwords further (the lo block)
__device__ __forceinline__ int s16_slot(int col) { return (col & ~31) | ((col & 31) >> 1) | ((col & 1) << 4); }
// reads of an S16-f16 tensor at (pixel row base in floats, channel): the high half alone (sign / zero test: hi == 0 implies lo == 0) or the value hi + lo
__device__ __forceinline__ float s16_hi(const float* row, int col) { return (float)reinterpret_cast<const _Float16*>(row + (col & ~31))[col & 31]; }
__device__ __forceinline__ float s16_val(const float* row, int col) {
    const _Float16* h = reinterpret_cast<const _Float16*>(row + (col & ~31));
    return (float)h[col & 31] + (float)h[32 + (col & 31)];
}

// ---- epilogue of k_conv_hx (round 5: rewritten).  D fragment map: col = lane & 31 (output channel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (pixel of the tile).
// The first form tested the launch-uniform modes (split-K, residual, accumulate, mask, tile clipping) per VALUE inside fully unrolled loops with 64-bit addresses: ~10 000
// instructions and ~1000 branches per instance, and the mask reads of the VGG19 dgrads were one dependent load + wait per value -- all of it exposed (one workgroup per CU: nothing
// overlaps an epilogue).  Now: wave-uniform base pointers + 32-bit byte offsets (one add per value), the row test on the scalar unit, column tests only in workgroups that straddle
// the image border (INTR = false), the activation as a slope select, everything a 32 x 32 block READS (mask / seed / residual / old value) requested back to back before the first
// use, and one straight-line instance per workgroup-uniform case: MODE 0 plain assigning store, 1 + ReLU mask / L1 seed (VGG19 dgrads), 2 everything (split-K atomics / slabs,
// residual, accumulate).  Measured: full step 141.9 -> 127.5 ms, E/R/A/D step 66.1 -> 63.1 ms (profiles/r05_experiments.md).
template <typename T, int TH, int TW, int BN, int WM, int WN, int EP, bool SO, bool INTR, int MODE, int TMt, int TNt>
__device__ __forceinline__ void hx_epilogue(f32x16 (&acc)[TMt][TNt], float (&st1)[TNt], float (&st2)[TNt], unsigned& amax_o, const ConvArgs& a,
                                            int n, int n0, int y0, int x0, int lane, int wave) {
    constexpr bool E_POOL = EP == 1 || EP == 3, E_MASK = EP == 2 || EP == 3;
    constexpr int BM = TH * TW, RW = BM / WM / TW;           // tile rows per wave
    const int wm_u = __builtin_amdgcn_readfirstlane(wave) / WN, wn = wave % WN;      // (wave-uniform copy: keeps the row arithmetic on the scalar unit)
    const int yw = y0 + wm_u * RW, xl = x0 + 4 * (lane >> 5);      // first image row of this wave's rows (uniform) / this lane's first column
    const int nx = a.W - xl;                                  // columns of this lane's row segment that lie inside the image
    const bool masked = MODE == 1 || (MODE == 2 && E_MASK && a.mask != nullptr);
    const bool split = MODE == 2 && a.splitk > 1, slabs = split && a.split_stride != 0, atomics = split && !slabs;
    const bool with_res = MODE == 2 && !split && a.res != nullptr, accum = MODE == 2 && !split && a.accumulate != 0;
    char* const op = reinterpret_cast<char*>(a.out + (long)n * a.out_sn + (slabs ? (long)blockIdx.z * a.split_stride : 0L));
    const char* const mp = masked ? reinterpret_cast<const char*>(a.mask + (long)n * a.out_sn) : nullptr;
    const char* const sp = (masked && a.seed_ref) ? reinterpret_cast<const char*>(a.seed_ref + (long)n * a.out_sn) : nullptr;
    const char* const rp = with_res ? reinterpret_cast<const char*>(a.res + (long)n * a.res_sn) : nullptr;
    const unsigned ldb = (unsigned)a.out_ld * 4u, wldb = (unsigned)a.W * ldb;
    const unsigned rldb = (unsigned)a.res_ld * 4u, rwldb = (unsigned)a.W * rldb;
    const unsigned pix0 = (unsigned)(yw * a.W + xl);          // this lane's first pixel (one sample stays below 4 GB: the launcher checks)
    const float osc = a.out_scale, slope = split ? 1.f : (a.act == 2 ? 0.f : (a.act == 3 ? 0.2f : 1.f)), sw = a.seed_w;
    const bool store_full = !E_POOL || !a.skip_out;
    const int ms16 = a.mask_s16, ss16 = a.seed_s16;
    // byte offset of value r of row block i relative to the lane's first pixel, for a pixel pitch of ldb_ / row pitch of wldb_
#define HX_EOFF(i_, r_, wldb_, ldb_) ((unsigned)((i_) * 2 + ((r_) >> 3)) * (wldb_) + (unsigned)(((r_) & 3) + 8 * (((r_) >> 2) & 1)) * (ldb_))
    // is value r of row block i inside the image?  (rows: wave-uniform; columns: only evaluated by border workgroups)
#define HX_EOK(i_, r_) (INTR || (yw + (i_) * 2 + ((r_) >> 3) < a.H && ((r_) & 3) + 8 * (((r_) >> 2) & 1) < nx))
#pragma unroll
    for (int j = 0; j < TNt; j++) {
        const int col = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
        if (col < a.Cout) {                                  // (lane-divergent only on the padded tail of a layer whose width is no multiple of 32; never with S16 tensors)
            const float bv = (a.bias && blockIdx.z == 0) ? a.bias[col] : 0.f;
            // channel part of the byte offsets (= a valid address by itself: pixel 0 of the sample, where clipped values send their loads) and the lane's first pixel added
            const unsigned f0 = (unsigned)col * 4u, h0 = (unsigned)(col & ~31) * 4u + (unsigned)(col & 31) * 2u;      // fp32 tensors / high half of channel col in an S16 tensor
            const unsigned m0 = ms16 ? h0 : f0, s0 = ss16 ? h0 : f0;
            const unsigned cb = pix0 * ldb + (SO ? (unsigned)s16_slot(col) * 4u : f0);
            const unsigned fb = pix0 * ldb + f0, mb = pix0 * ldb + m0, sb = pix0 * ldb + s0, rb = pix0 * rldb + f0;
#pragma unroll
            for (int i = 0; i < TMt; i++) {
                float mk[16], sd[16], ex[16];                 // mask / seed reference / (residual | old value): requested back to back, consumed below
                if (masked) {
                    if (!ms16) {
#pragma unroll
                        for (int r = 0; r < 16; r++) mk[r] = *reinterpret_cast<const float*>(mp + (HX_EOK(i, r) ? mb + HX_EOFF(i, r, wldb, ldb) : m0));
                    } else if (sp == nullptr) {               // sign / zero test only: the high half decides (hi == 0 implies lo == 0)
#pragma unroll
                        for (int r = 0; r < 16; r++) mk[r] = (float)*reinterpret_cast<const _Float16*>(mp + (HX_EOK(i, r) ? mb + HX_EOFF(i, r, wldb, ldb) : m0));
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; r++) { const _Float16* h_ = reinterpret_cast<const _Float16*>(mp + (HX_EOK(i, r) ? mb + HX_EOFF(i, r, wldb, ldb) : m0));
                                                       mk[r] = (float)h_[0] + (float)h_[32]; }
                    }
                    if (sp != nullptr) {
                        if (!ss16) {
#pragma unroll
                            for (int r = 0; r < 16; r++) sd[r] = *reinterpret_cast<const float*>(sp + (HX_EOK(i, r) ? sb + HX_EOFF(i, r, wldb, ldb) : s0));
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; r++) { const _Float16* h_ = reinterpret_cast<const _Float16*>(sp + (HX_EOK(i, r) ? sb + HX_EOFF(i, r, wldb, ldb) : s0));
                                                           sd[r] = (float)h_[0] + (float)h_[32]; }
                        }
                    }
                }
                if (with_res) {
#pragma unroll
                    for (int r = 0; r < 16; r++) ex[r] = *reinterpret_cast<const float*>(rp + (HX_EOK(i, r) ? rb + HX_EOFF(i, r, rwldb, rldb) : f0));
                } else if (accum) {
#pragma unroll
                    for (int r = 0; r < 16; r++) ex[r] = *reinterpret_cast<const float*>(op + (HX_EOK(i, r) ? fb + HX_EOFF(i, r, wldb, ldb) : f0));
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float v = acc[i][j][r] * osc + bv;
                    if (with_res) v += ex[r];
                    v = v > 0.f ? v : slope * v;
                    if (masked) {
                        if (sp != nullptr) { const float d_ = mk[r] - sd[r]; v += d_ > 0.f ? sw : (d_ < 0.f ? -sw : 0.f); }
                        v = mk[r] > 0.f ? v : 0.f;
                    }
                    if (accum) v += ex[r];
                    if (E_POOL) acc[i][j][r] = v;             // (kept for the fused max-pool below; clamped by the range guard when the launch writes S16 tensors)
                    if (HX_EOK(i, r)) {
                        if (EP == 0 && !split) { st1[j] += v; st2[j] = fmaf(v, v, st2[j]); }
                        const unsigned o_ = cb + HX_EOFF(i, r, wldb, ldb);
                        if (SO) {       // S16 output: range guard (f16) + split here, once per output element (both lanes of a channel pair are inside or outside together)
                            if (!is_bf16<T>::value) { amax_o = max(amax_o, __float_as_uint(v) & 0x7fffffffu); v = __builtin_amdgcn_fmed3f(v, -HX_F16_MAX, HX_F16_MAX); if (E_POOL) acc[i][j][r] = v; }
                            const unsigned w_ = s16_pair<T>(v, lane);
                            if (store_full) *reinterpret_cast<unsigned*>(op + o_) = w_;
                        } else if (atomics) atomicAdd(reinterpret_cast<float*>(op + o_), v);
                        else if (store_full) *reinterpret_cast<float*>(op + o_) = v;
                    }
                }
                if (E_POOL && a.pool_out) {      // 2x2 max of the activated values: window = accumulators {r, r + 1, r + 8, r + 9}, r in {0, 2, 4, 6} (rows 2i / 2i + 1 of the tile, columns x, x + 1)
                    char* const pp = reinterpret_cast<char*>(a.pool_out + (long)n * a.pool_sn);
                    const unsigned plb = (unsigned)a.pool_ld * 4u;
                    const int py = (yw + i * 2) >> 1;
                    const unsigned pb0 = (unsigned)(py * (a.W >> 1) + (xl >> 1)) * plb + ((SO && a.pool_s16) ? (unsigned)s16_slot(col) : (unsigned)col) * 4u;
#pragma unroll
                    for (int r = 0; r < 8; r += 2) {
                        const int pdx = ((r & 3) + 8 * ((r >> 2) & 1)) >> 1;
                        if (!INTR && (py >= (a.H >> 1) || (xl >> 1) + pdx >= (a.W >> 1))) continue;      // floor semantics: the last row / column of an odd map belongs to no window
                        const float mx = fmaxf(fmaxf(acc[i][j][r], acc[i][j][r + 1]), fmaxf(acc[i][j][r + 8], acc[i][j][r + 9]));
                        const unsigned po_ = pb0 + (unsigned)pdx * plb;
                        if (SO && a.pool_s16) *reinterpret_cast<unsigned*>(pp + po_) = s16_pair<T>(mx, lane);
                        else *reinterpret_cast<float*>(pp + po_) = mx;
                    }
                }
            }
        }
    }
#undef HX_EOFF
#undef HX_EOK
}

// T: _Float16 / __bf16.  NPL planes staged (2: hi + lo, 3 products; 1: hi only).  Tile TH x TW pixels x BN output channels, WM x WN waves
// (4 or 8), each owning a (BM / WM) x (BN / WN) sub-tile.  D = depth of the register ring of weight tiles: the tile of step s + D is
// requested while step s computes (D = 1: next step only; D = 3: ~1.9 us of latency tolerance at 8 waves -- the weights of a 512-channel
// layer (9.4 MB split) do not stay in the 4 MB L2 of an XCD, and one step of MFMA work (0.3 - 0.6 us) cannot hide that round trip).
// Measured on the MI355X (tools/gpu_hx_experiments.sh, VGG 512 -> 512 @32x32 x 60): of 864 us at D = 1 / 4 waves, 46 % was the weight-tile
// path and 32 % the activation staging (its global loads were also waited for by the in-order vmcnt of the next weight tile).
// EP: epilogue extras compiled in (template parameter, not run-time tests: a workgroup of an under-filled launch is a cold-start chain of prologue ->
// 9..18 steps -> epilogue, and the fully unrolled epilogue is half of the ~40 KB of code it has to fetch; with the VGG19-only paths in every
// instance the batch-1 roll-out ran 7 % slower).  0: bias / residual / ReLU / LeakyReLU / accumulate / split-K -- everything the model's layers need;
// 1: + fused 2x2 max-pool (ConvArgs.pool_out) and write-less mode (skip_out) of the VGG19 layers in front of a pool; 2: + ReLU mask / L1 seed of the
// VGG19 dgrad chain (ConvArgs.mask, seed_ref); 3: both (single-product study variants).  tanh (FinalBlocks) is not an hx epilogue at all.
// IO (round 5): operand formats at the kernel boundary.  bit 0 (PS): the input segment is PRE-SPLIT ("S16", common.h) -- the staging is a plain 16-byte copy per lane, no
// conversion, no range guard (the producer applied it).  bit 1 (SO): the epilogue writes `out` / `pool_out` as S16 of T -- split once per OUTPUT element instead of once
// per staged halo element in every output-channel block of every consumer (the 512-channel VGG19 layers staged and converted each halo tile four times).
// BG (round 6): the under-filled tile variants with the WEIGHT fragments straight from global memory / L2 into registers.  A (tap, chunk) step of a 4 x 16 or 8 x 16-pixel tile
// is 6 - 12 MFMAs per wave between two barriers (weight tile -> LDS -> barrier -> fragment reads), ~890 cycles measured against 190 - 380 of matrix work, and every wave read
// A and B fragments of 32 x 32 outputs each: 4 LDS fragment reads per 3 MFMAs.  Here the two waves of a row pair (wm = 2 wr + kh) multiply the SAME 2 x (BM / WM) pixel rows, each
// with one 16-channel half (kh) of every 32-channel step: the B fragment of (tap, chunk, kh) is one 16-byte load per lane and plane from the packed weights (a row's 16-byte
// pieces ARE the fragments; D-deep register ring, no LDS, no barrier), the A fragments of twice the rows are read for half the channels -- 2 fragment reads + 1 load per 3 MFMAs,
// no redundant weight traffic between the waves.  The halo image is double-buffered (the next chunk is stored while this one is multiplied): ONE barrier per 32-channel chunk
// instead of ten.  After the loop the two waves exchange the halves of their partial tiles through LDS (partial of kh = 0 + partial of kh = 1: a fixed order) and continue with
// the usual accumulator layout -- epilogues, BatchNorm partial sums, K split unchanged.
template <typename T, int NPL, int TH, int TW, int BN, int WM, int WN, int D, int EP = 0, int IO = 0, int BG = 0>
__global__ __launch_bounds__(64 * WM * WN) void k_conv_hx(ConvArgs a, int tiles_x, int tiles_y) {
    typedef typename Vec<T>::v8 v8;
    typedef typename Vec<T>::v4 v4;
    constexpr int NT = 64 * WM * WN;                         // threads
    constexpr bool E_POOL = EP == 1 || EP == 3, E_MASK = EP == 2 || EP == 3;
    constexpr bool PS = (IO & 1) != 0, SO = (IO & 2) != 0;
    static_assert(!(PS || SO) || NPL == 2, "S16 tensors carry both halves");
    constexpr int BM = TH * TW;
    constexpr int HW_ = TW + 2, HH_ = TH + 2, HPX = HW_ * HH_;
    constexpr int PITCH = NPL * KC + 8;                      // LDS row pitch in 16-bit elements: 72 (144 B) or 40 (80 B)
    // Halo image: pixel (hy, hx) at hy * AROW + hx * PITCH with the ROW pitch a multiple of 256 B.  A 32x32x16 A-fragment read (ds_read_b128) covers two tile rows of 16
    // pixels; the instruction is serviced in the lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32), i.e. 8 pixels of one tile row and 8 of the next.  With the odd
    // 16-byte-slot pitch (9 or 5) pixels i and j of one row never share a slot unless i = j (mod 16); a row pitch of k * 256 B makes the second row's pixels fall on the
    // slots of the SAME pixel numbers, i.e. on the eight slots the first row's eight pixels leave free: conflict-free.  (Dense rows -- HW_ * PITCH -- put two of the 16 pixels
    // of every group on a busy slot: SQ_LDS_BANK_CONFLICT was 35 % of SQ_LDS_IDX_ACTIVE on the 8 x 16 x 64 tile, profiles/r04_conv_pmc.md.)
    constexpr int AROW = (HW_ * PITCH + 127) / 128 * 128;
    constexpr int APP = NT / 8;                              // halo pixels staged per pass (8 threads x float4 cover one pixel's 32-channel chunk)
    constexpr int NA = (HPX + APP - 1) / APP;
    constexpr int BROW16 = NPL * KC * 2 / 16;                // 16-byte pieces per weight row
    constexpr int NB = (BN * BROW16 + NT - 1) / NT;
    constexpr int TMt = BM / WM / 32, TNt = BN / WN / 32;
    static_assert((WM * WN == 4 || WM * WN == 8) && BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "tile / wave layout");
    static_assert(TW == 16, "row <-> pixel map assumes 16-pixel tile rows");
    static_assert(D == 1 || D == 3, "9 taps per chunk: the ring depth must divide 9");
    constexpr bool BGM = BG != 0;
    constexpr int TMF = BGM ? 2 * TMt : TMt;                 // M tiles a wave multiplies (BG: the rows of its wave pair)
    constexpr int ASZ = HH_ * AROW;
    static_assert(!BGM || (NPL == 2 && (WM & 1) == 0 && D == 3 && EP == 0 && (IO & 2) == 0), "BG: split operands, wave row pairs, plain epilogue");
    static_assert(!BGM || 2 * ASZ * (int)sizeof(T) >= WM * WN * TMt * TNt * 16 * 64 * 4, "BG: the exchange of the partial tiles fits into the halo buffers");
    __shared__ __attribute__((aligned(16))) T As[(BGM ? 2 : 1) * ASZ];
    __shared__ __attribute__((aligned(16))) T Bs[2][BGM ? 8 : BN * PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // workgroup -> (pixel tile, output-channel block).  Workgroups go to the 8 XCDs round-robin in launch order (x fastest); with several
    // output-channel blocks (a.xcd_map) the ones of a pixel tile are put on the SAME XCD in consecutive slots, so that the tile's halo images are
    // fetched into that XCD's L2 once instead of once per channel block
    int tile = blockIdx.x, nblk = blockIdx.y;
    if (a.xcd_map && gridDim.y > 1) {
        const int nby = gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int grp = lin / (8 * nby), r = lin - grp * (8 * nby);
        const int m = (int)gridDim.x - grp * 8 < 8 ? (int)gridDim.x - grp * 8 : 8;      // tiles in this group (the last one may be short)
        tile = grp * 8 + r % m; nblk = r / m;
    }
    const int tile_lin = tile;                                // (sample, tile row, tile column) in launch-independent order: index of this tile's BatchNorm partial sums
    const int n = tile / (tiles_x * tiles_y);
    tile -= n * tiles_x * tiles_y;
    const int y0 = (tile / tiles_x) * TH, x0 = (tile % tiles_x) * TW;
    const int n0 = nblk * BN;
    const int nchunks = a.Kq / KC;

    // ---- A staging roles: float4 column q of halo pixels (tid >> 3) + 32 i ----
    // (pixel order inside every block of eight: 0 4 1 5 2 6 3 7 -- the 16 contiguous lanes that one ds_write_b64 group serves then hold pixels p and p + 4, whose 64-byte
    //  halves fall on disjoint bank halves (pixel pitch 36 dwords = 4 mod 32); neighbouring pixels p, p + 1 overlapped on 12 of their 16 banks)
    const int q = tid & 7;
    const int tp = ((tid >> 3) & ~7) | (((tid >> 3) & 1) << 2) | (((tid >> 3) >> 1) & 3);
    int pixoff[NA];                                           // (y * W + x) of the halo pixel, -1: outside the image / beyond the halo
    int aoff[NA];                                             // element offset of that pixel's row in the LDS image
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int p = tp + APP * i;
        const int hy = p / HW_, hx = p - hy * HW_;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        pixoff[i] = (p < HPX && y >= 0 && y < a.H && x >= 0 && x < a.W) ? y * a.W + x : -1;
        aoff[i] = hy * AROW + hx * PITCH + (PS ? 8 : 4) * q;      // (PS: lane q carries 16 bytes = 8 halves of the pixel's [hi 32 | lo 32] row)
    }
    // (staging steps are macros, not lambdas: a by-reference capture of the kernel-argument struct / register arrays forces them into
    //  scratch memory)
    float4 ra[NA];
    int asto = 0;                                             // element offset of the halo buffer HX_STORE_A writes (BG: the buffer of the NEXT chunk)
    unsigned amax = 0u;                                       // split-f16 only: largest |x| this thread staged, as a bit pattern -- NaN > inf > finite (saturation guard, ConvArgs.sat_flag)
    float4 rsc, rsh;                                          // lazily applied BatchNorm of the producer (ConvSrc.bn_*): scale / shift of this thread's four channels, chunk in flight
#define HX_SEG_OF(chunk_)                                                                                                          \
        int s_ = 0, c0_ = (chunk_) * KC;                                                                                          \
        while (s_ + 1 < a.nsrc && c0_ >= (a.src[s_].C + KC - 1) / KC * KC) { c0_ -= (a.src[s_].C + KC - 1) / KC * KC; s_++; }     \
        const ConvSrc sg_ = a.src[s_];                          /* wave-uniform */                                                \
        const int c_ = c0_ + 4 * q;                                                                                               \
        const bool cok_ = c_ < sg_.C;
    // loads only (clamped addresses, no use of the loaded values: the zero-padding / channel-tail selects happen in HX_STORE_A, one chunk
    // later, so that nothing waits for these loads while the taps of the current chunk run)
#define HX_LOAD_A(chunk_)                                                                                                          \
    do {                                                                                                                           \
        if (PS) {       /* pre-split input (one segment, C a multiple of 32): pixel row of the chunk = 128 contiguous bytes [hi 32 | lo 32] */ \
            const float* base_ = a.src[0].p + (long)n * a.src[0].sn + (chunk_) * KC + 4 * q;                                      \
            const int pl_ = a.src[0].ld;                                                                                           \
            _Pragma("unroll") for (int i = 0; i < NA; i++)                                                                        \
                ra[i] = *reinterpret_cast<const float4*>(base_ + (long)(pixoff[i] >= 0 ? pixoff[i] : 0) * pl_);                   \
            break;                                                                                                                 \
        }                                                                                                                          \
        HX_SEG_OF(chunk_)                                                                                                          \
        const float* base_ = sg_.p + (long)n * sg_.sn + (cok_ ? c_ : 0);                                                          \
        const int pl_ = sg_.bcast ? 0 : sg_.ld;                                                                                   \
        _Pragma("unroll") for (int i = 0; i < NA; i++)                                                                            \
            ra[i] = *reinterpret_cast<const float4*>(base_ + (long)((cok_ && pixoff[i] >= 0) ? pixoff[i] : 0) * pl_);             \
        /* (unconditional, clamped to a valid address when the segment carries no BatchNorm: no branch around loads) */           \
        const long bo_ = (sg_.bn_gn > 0 ? (long)(n / sg_.bn_gn) * sg_.bn_gs : 0L) + (cok_ ? c_ : 0);                              \
        rsc = *reinterpret_cast<const float4*>(sg_.bn_scale ? sg_.bn_scale + bo_ : sg_.p);                                        \
        rsh = *reinterpret_cast<const float4*>(sg_.bn_scale ? sg_.bn_shift + bo_ : sg_.p);                                        \
    } while (0)
#define HX_STORE_A(chunk_)                                                                                                          \
    do {                                                                                                                           \
        if (PS) {       /* 16-byte copy (zero padding outside the image) */                                                        \
            _Pragma("unroll") for (int i = 0; i < NA; i++) {                                                                      \
                const int p_ = tp + APP * i;                                                                                      \
                if (HPX % APP == 0 || p_ < HPX) {                                                                                  \
                    float4 v_ = ra[i];                                                                                             \
                    if (pixoff[i] < 0) v_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                       \
                    *reinterpret_cast<float4*>(&As[asto + aoff[i]]) = v_;                                                          \
                }                                                                                                                  \
            }                                                                                                                      \
            break;                                                                                                                 \
        }                                                                                                                          \
        HX_SEG_OF(chunk_)                                                                                                          \
        const bool m1_ = c_ + 1 < sg_.C, m2_ = c_ + 2 < sg_.C, m3_ = c_ + 3 < sg_.C;                                              \
        const bool bn_ = sg_.bn_scale != nullptr;               /* wave-uniform */                                                \
        const float sl_ = sg_.bn_act ? 0.2f : 1.f;                                                                                \
        _Pragma("unroll") for (int i = 0; i < NA; i++) {                                                                          \
            const int p_ = tp + APP * i;                                                                                          \
            if (HPX % APP == 0 || p_ < HPX) {                                                                                      \
                const bool ok_ = cok_ && pixoff[i] >= 0;                                                                           \
                float4 v_ = ra[i];                                                                                                 \
                if (bn_) {      /* act(x * scale + shift); the zero padding below applies to the NORMALISED tensor */             \
                    v_.x = fmaf(v_.x, rsc.x, rsh.x); v_.y = fmaf(v_.y, rsc.y, rsh.y); v_.z = fmaf(v_.z, rsc.z, rsh.z); v_.w = fmaf(v_.w, rsc.w, rsh.w); \
                    v_.x = v_.x > 0.f ? v_.x : sl_ * v_.x; v_.y = v_.y > 0.f ? v_.y : sl_ * v_.y;                                  \
                    v_.z = v_.z > 0.f ? v_.z : sl_ * v_.z; v_.w = v_.w > 0.f ? v_.w : sl_ * v_.w;                                  \
                }                                                                                                                  \
                float x0_ = ok_ ? v_.x : 0.f, x1_ = (ok_ && m1_) ? v_.y : 0.f;                                                     \
                float x2_ = (ok_ && m2_) ? v_.z : 0.f, x3_ = (ok_ && m3_) ? v_.w : 0.f;                                            \
                if (!is_bf16<T>::value) {      /* f16 range guard: 2.5 VALU per element (an unsigned max of the magnitude bits -- it orders NaN above inf above every */ \
                    /* finite value -- and one v_med3 clamp; a NaN is clamped too, but it raises the flag) */                     \
                    amax = max(max(amax, max(__float_as_uint(x0_) & 0x7fffffffu, __float_as_uint(x1_) & 0x7fffffffu)),             \
                               max(__float_as_uint(x2_) & 0x7fffffffu, __float_as_uint(x3_) & 0x7fffffffu));                       \
                    x0_ = __builtin_amdgcn_fmed3f(x0_, -HX_F16_MAX, HX_F16_MAX); x1_ = __builtin_amdgcn_fmed3f(x1_, -HX_F16_MAX, HX_F16_MAX); \
                    x2_ = __builtin_amdgcn_fmed3f(x2_, -HX_F16_MAX, HX_F16_MAX); x3_ = __builtin_amdgcn_fmed3f(x3_, -HX_F16_MAX, HX_F16_MAX); \
                }                                                                                                                  \
                v4 hi_, lo_;                                                                                                       \
                hi_[0] = (T)x0_; hi_[1] = (T)x1_; hi_[2] = (T)x2_; hi_[3] = (T)x3_;                                                \
                lo_[0] = (T)(x0_ - (float)hi_[0]); lo_[1] = (T)(x1_ - (float)hi_[1]);                                             \
                lo_[2] = (T)(x2_ - (float)hi_[2]); lo_[3] = (T)(x3_ - (float)hi_[3]);                                             \
                *reinterpret_cast<v4*>(&As[asto + aoff[i]]) = hi_;                                                                \
                if (NPL == 2) *reinterpret_cast<v4*>(&As[asto + aoff[i] + KC]) = lo_;                                             \
            }                                                                                                                      \
        }                                                                                                                          \
    } while (0)
    // ---- B staging: the (tap, chunk) tile of this workgroup is BN rows x (NPL * 64) bytes, contiguous in global memory ----
    const T* wq = reinterpret_cast<const T*>(a.wq);
    const long wrow = (long)a.Cout_pad * (NPL * KC);
    u32x4 rb[D][NB];
#define HX_LOAD_B(set_, tap_, chunk_)                                                                                              \
    do {                                                                                                                           \
        const u32x4* g_ = reinterpret_cast<const u32x4*>(wq + ((long)(tap_) * nchunks + (chunk_)) * wrow + (long)n0 * (NPL * KC)); \
        _Pragma("unroll") for (int i = 0; i < NB; i++) {                                                                          \
            const int idx_ = tid + NT * i;                                                                                        \
            if ((BN * BROW16) % NT == 0 || idx_ < BN * BROW16) rb[set_][i] = g_[idx_];                                            \
        }                                                                                                                          \
    } while (0)
#define HX_STORE_B(set_, buf_)                                                                                                     \
    do {                                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < NB; i++) {                                                                          \
            const int idx_ = tid + NT * i;                                                                                        \
            if ((BN * BROW16) % NT == 0 || idx_ < BN * BROW16) {      /* piece idx_ of the tile in fragment-major order (pk_hx_elem) -> its place in the row-major LDS tile */ \
                const int f_ = (idx_ >> 6) % (2 * NPL), l_ = idx_ & 63;                                                          \
                *reinterpret_cast<u32x4*>(&Bs[buf_][((idx_ / (128 * NPL)) * 32 + (l_ & 31)) * PITCH + (f_ % NPL) * KC + (f_ / NPL) * 16 + (l_ >> 5) * 8]) = rb[set_][i]; } \
        }                                                                                                                          \
    } while (0)
    // weight tile of the flattened step index st_ = (chunk - ch0) * 9 + tap, if it exists
    // (unconditional: past the end the last tile is requested again -- a branch around the loads makes hipcc fall back to vmcnt(0) waits)
#define HX_LOAD_B_STEP(set_, st_)                                                                                                  \
    do { int s__ = (st_); s__ = s__ < nsteps ? s__ : nsteps - 1; const int c__ = s__ / 9; HX_LOAD_B(set_, s__ - 9 * c__, ch0 + c__); } while (0)

    // ---- fragment addresses ----
    int abase[TMF], bbase[TNt];
#pragma unroll
    for (int i = 0; i < TMF; i++) {
        const int m = (BGM ? (wm & ~1) : wm) * (BM / WM) + i * 32 + (lane & 31);
        abase[i] = (m / TW) * AROW + (m % TW) * PITCH + (lane >> 5) * 8 + (BGM ? (wm & 1) * 16 : 0);
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
    const int nsteps = ch0 < ch1 ? (ch1 - ch0) * 9 : 0;
    if constexpr (!BGM) {
    if (ch0 < ch1) {
        HX_LOAD_A(ch0);
#pragma unroll
        for (int d = 0; d < D; d++) HX_LOAD_B_STEP(d, d);
    }
    if (nsteps == 0) { /* empty K slice of a split launch: nothing to add */ }
    int bbuf = 0, step = 0;
    for (int chunk = ch0; chunk < ch1; chunk++) {
        __syncthreads();                                      // every wave is done reading As (previous chunk)
        HX_STORE_A(chunk);
        // (D = 3: fully unrolled -- hipcc drains vmcnt to 0 at every loop back-edge, which would cut the ring back to depth 1 every third step)
#pragma unroll
        for (int tap0 = 0; tap0 < 9; tap0 += D) {
#pragma unroll
            for (int d = 0; d < D; d++, step++) {
                const int tap = tap0 + d;
                HX_STORE_B(d, bbuf);
                __syncthreads();
                HX_LOAD_B_STEP(d, step + D);                  // refill the register set just drained
                if (tap == (D == 1 ? 0 : 3)) HX_LOAD_A(chunk + 1 < ch1 ? chunk + 1 : chunk);      // next halo tile: consumed >= 6 steps later (last chunk: re-requested, unused)
                const int toff = (tap / 3) * AROW + (tap % 3) * PITCH;
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
                            if (NPL == 2) {                   // small terms first
                                acc[i][j] = mfma16(fa[i][NPL - 1], fb[j][0], acc[i][j]);
                                acc[i][j] = mfma16(fa[i][0], fb[j][NPL - 1], acc[i][j]);
                            }
                            acc[i][j] = mfma16(fa[i][0], fb[j][0], acc[i][j]);
                        }
                }
                bbuf ^= 1;
            }
        }
    }
    } else {
    // ---- BG: weight fragments straight into a nine-deep register ring, halo image double-buffered, one barrier per chunk ----
    f32x16 accf[TMF][TNt];
#pragma unroll
    for (int i = 0; i < TMF; i++)
#pragma unroll
        for (int j = 0; j < TNt; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) accf[i][j][r] = 0.f;
    // this lane's 16 bytes of the B fragment (32-row block (n0 + wn (BN / WN)) / 32 + j, K half kh = wm & 1, plane pl): 1 KB contiguous per fragment (pk_hx_elem's order)
    const T* wqb = wq + ((long)((((n0 + wn * (BN / WN)) >> 5) * 2 + (wm & 1)) * NPL) * 64 + lane) * 8;
    // ring depth = one chunk: the weights of R's gate convolutions (10 - 20 MB split) do not stay in an XCD's 4-MB L2, every workgroup of an XCD walks them in step, so each
    // (tap, chunk) tile is a miss for all of them at once -- ~1 us from the memory-side cache against 0.08 - 0.16 us of matrix work per tap.  (Depth 3: 70.0 us for the 528 -> 1024
    // gate convolution of a 16 x 16 map; see profiles/r06_experiments.md)
    constexpr int DB = 9;
    v8 fbr[DB][TNt][NPL];
#ifndef HX_BG_LOADA_TAP
#define HX_BG_LOADA_TAP 1      /* tap behind which the next chunk's halo tile is requested (alone: behind tap 3 76.7 us, tap 1 65.5, tap 0 67.2 for the 528 -> 1024 gate convolution; in situ no difference) */
#endif
#ifndef HX_EXP
#define HX_EXP 0      /* timing studies (tools/build_exp.sh; results are wrong with any bit set): 1 no halo conversion / store in the loop, 2 no weight loads, 4 no fragment reads, 8 no halo loads */
#endif
#define HX_LOAD_BG(set_, st_)                                                                                                      \
    do { int s__ = (st_); s__ = s__ < nsteps ? s__ : nsteps - 1; const int c__ = s__ / 9;                                         \
         const T* g_ = wqb + ((long)(s__ - 9 * c__) * nchunks + (ch0 + c__)) * wrow;                                               \
         _Pragma("unroll") for (int j = 0; j < TNt; j++)                                                                          \
             _Pragma("unroll") for (int pl = 0; pl < NPL; pl++) fbr[set_][j][pl] = *reinterpret_cast<const v8*>(g_ + (j * 2 * NPL + pl) * (64 * 8)); \
    } while (0)
    // A fragments of tap t_ (row blocks i0_ ... i0_ + HM - 1, both planes) into register set fs_
#define HX_READ_FA(fs_, t_, i0_)                                                                                                   \
    do { const int toff_ = ((t_) / 3) * AROW + ((t_) % 3) * PITCH;                                                               \
         _Pragma("unroll") for (int i = 0; i < HM; i++)                                                                           \
             _Pragma("unroll") for (int pl = 0; pl < NPL; pl++) fa[fs_][i][pl] = *reinterpret_cast<const v8*>(&Ac[abase[(i0_) + i] + toff_ + pl * KC]); \
    } while (0)
    if (ch0 < ch1) {
        HX_LOAD_A(ch0);
#pragma unroll
        for (int d = 0; d < DB; d++) HX_LOAD_BG(d, d);
        HX_STORE_A(ch0);                                      // (buffer 0)
    }
    __syncthreads();
    int cur = 0, step = 0;
    for (int chunk = ch0; chunk < ch1; chunk++) {
        const T* Ac = As + cur * ASZ;
        // A fragments ahead of their MFMAs (with one register set hipcc re-used eight registers for every fragment pair and each MFMA triple waited for its own LDS round trip):
        // TMF = 2: the fragments of tap t + 1 are requested before the MFMAs of tap t (two sets); TMF = 4: half a tap ahead -- the upper row blocks of tap t are requested before
        // the MFMAs of its lower row blocks, the lower row blocks of tap t + 1 before the MFMAs of the upper ones (two half sets: 32 registers instead of 64 keep the instance at
        // two waves per SIMD beside the nine-deep weight ring)
        constexpr int HM = TMF >= 4 ? TMF / 2 : TMF;          // row blocks per fragment set
        v8 fa[2][HM][NPL];
#define HX_MFMA_SET(fs_, i0_)                                                                                                     \
        do {      /* small terms first; product-major: consecutive MFMAs go to different accumulators (the order per accumulator is what matters for the result) */ \
            _Pragma("unroll") for (int i = 0; i < HM; i++)                                                                        \
                _Pragma("unroll") for (int j = 0; j < TNt; j++) accf[(i0_) + i][j] = mfma16(fa[fs_][i][1], fbr[d][j][0], accf[(i0_) + i][j]); \
            _Pragma("unroll") for (int i = 0; i < HM; i++)                                                                        \
                _Pragma("unroll") for (int j = 0; j < TNt; j++) accf[(i0_) + i][j] = mfma16(fa[fs_][i][0], fbr[d][j][1], accf[(i0_) + i][j]); \
            _Pragma("unroll") for (int i = 0; i < HM; i++)                                                                        \
                _Pragma("unroll") for (int j = 0; j < TNt; j++) accf[(i0_) + i][j] = mfma16(fa[fs_][i][0], fbr[d][j][0], accf[(i0_) + i][j]); \
        } while (0)
        HX_READ_FA(0, 0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; tap++, step++) {
            const int d = tap % DB;
            if (TMF >= 4) {
                if (!(HX_EXP & 4)) HX_READ_FA(1, tap, HM);
                __builtin_amdgcn_sched_barrier(0);
                HX_MFMA_SET(0, 0);
                if (tap + 1 < 9 && !(HX_EXP & 4)) HX_READ_FA(0, tap + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                HX_MFMA_SET(1, HM);
            } else {
                const int fs = tap & 1;
                if (tap + 1 < 9 && !(HX_EXP & 4)) HX_READ_FA(fs ^ 1, tap + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                HX_MFMA_SET((HX_EXP & 4) ? 0 : fs, 0);
            }
            if (!(HX_EXP & 2)) HX_LOAD_BG(d, step + DB);      // refill the register set just used
            if (tap == HX_BG_LOADA_TAP && !(HX_EXP & 8)) HX_LOAD_A(chunk + 1 < ch1 ? chunk + 1 : chunk);      // next halo tile (last chunk: re-requested, unused)
            // (without a barrier in the loop nothing stops hipcc's scheduler from sinking these loads down to their first use, a chunk later: load -> s_waitcnt vmcnt(0) -> MFMA,
            //  the ring gone.  Nothing may cross this point.)
            __builtin_amdgcn_sched_barrier(0);
        }
#undef HX_MFMA_SET
        if (chunk + 1 < ch1 && !(HX_EXP & 1)) { asto = (cur ^ 1) * ASZ; HX_STORE_A(chunk + 1); }      // into the other buffer: every wave passed the previous barrier, i.e. is done reading it
        __syncthreads();
        cur ^= 1;
    }
#undef HX_READ_FA
#undef HX_LOAD_BG
    // exchange: wave (wr, kh) keeps the M tiles kh TMt ... of its pair's rows and receives the partner's partial sums of them; sum = partial of kh 0 + partial of kh 1
    {
        float* xr = reinterpret_cast<float*>(As);             // (every wave is past the loop's last barrier: the halo buffers are dead)
        const bool kh = (wm & 1) != 0;
        const int pw = (wm ^ 1) * WN + wn;
#pragma unroll
        for (int i = 0; i < TMt; i++)
#pragma unroll
            for (int j = 0; j < TNt; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) xr[(((wave * TMt + i) * TNt + j) * 16 + r) * 64 + lane] = kh ? accf[i][j][r] : accf[TMt + i][j][r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TMt; i++)
#pragma unroll
            for (int j = 0; j < TNt; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float got = xr[(((pw * TMt + i) * TNt + j) * 16 + r) * 64 + lane];
                    const float own = kh ? accf[TMt + i][j][r] : accf[i][j][r];
                    acc[i][j][r] = kh ? got + own : own + got;
                }
        __syncthreads();                                      // (the BatchNorm partial sums below reuse the buffer)
    }
    }

    if (!is_bf16<T>::value && a.sat_flag != nullptr && amax > 0x477fe000u /* bits of 65504.f */) atomicOr(a.sat_flag, amax > 0x7f800000u ? 3u : 1u);      // bit 1: a NaN among them      // (rare: one atomic per saturating thread)
    unsigned amax_o = 0u;                                     // SO, split f16: the same guard on what this launch stores
    // ---- epilogue: D fragment map col = lane & 31 (output channel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (pixel of the tile) ----
    float st1[TNt], st2[TNt];                                 // per-channel sums of the stored values (ConvArgs.stats: BatchNorm statistics of the consumer)
#pragma unroll
    for (int j = 0; j < TNt; j++) { st1[j] = 0.f; st2[j] = 0.f; }
    // ---- epilogue (hx_epilogue above): the workgroup-uniform cases are separate straight-line instances ----
    {
        const bool interior = y0 + TH <= a.H && x0 + TW <= a.W;
        const bool general = a.splitk > 1 || a.res != nullptr || a.accumulate != 0;
        const bool masked = E_MASK && a.mask != nullptr;
#define HX_EPI(INTR_, MODE_) hx_epilogue<T, TH, TW, BN, WM, WN, EP, SO, INTR_, MODE_>(acc, st1, st2, amax_o, a, n, n0, y0, x0, lane, wave)
        if (interior) { if (general) HX_EPI(true, 2); else if (masked) HX_EPI(true, (E_MASK ? 1 : 0)); else HX_EPI(true, 0); }
        else { if (general) HX_EPI(false, 2); else if (masked) HX_EPI(false, (E_MASK ? 1 : 0)); else HX_EPI(false, 0); }
#undef HX_EPI
    }
    if (SO && !is_bf16<T>::value && a.sat_flag != nullptr && amax_o > 0x477fe000u) atomicOr(a.sat_flag + (a.sat_out_next ? 1 : 0), amax_o > 0x7f800000u ? 3u : 1u);
    // ---- BatchNorm partial sums of this tile: the two 32-lane halves of a wave hold different pixel rows of one channel, the WM waves of a column
    // block different rows too -> shuffle, then LDS (the staging tiles are dead), one plain store per (tile, channel): no atomics, fixed order ----
    if (EP == 0 && a.stats != nullptr) {                      // (grid-uniform; the launcher only passes it with splitk == 1)
#pragma unroll
        for (int j = 0; j < TNt; j++) { st1[j] += __shfl_xor(st1[j], 32); st2[j] += __shfl_xor(st2[j], 32); }
        float* red = reinterpret_cast<float*>(As);            // WM x BN x 2 floats <= 4 KB
        if (WM > 1) {
            __syncthreads();                                   // every wave is done with As / Bs
            if (lane < 32) {
#pragma unroll
                for (int j = 0; j < TNt; j++) {
                    const int cl = wn * (BN / WN) + j * 32 + lane;
                    red[(wm * BN + cl) * 2] = st1[j]; red[(wm * BN + cl) * 2 + 1] = st2[j];
                }
            }
            __syncthreads();
        }
        if (wm == 0 && lane < 32) {
#pragma unroll
            for (int j = 0; j < TNt; j++) {
                const int cl = wn * (BN / WN) + j * 32 + lane, col = n0 + cl;
                float s1 = st1[j], s2 = st2[j];
                if (WM > 1) {
                    s1 = red[cl * 2]; s2 = red[cl * 2 + 1];
#pragma unroll
                    for (int w = 1; w < WM; w++) { s1 += red[(w * BN + cl) * 2]; s2 += red[(w * BN + cl) * 2 + 1]; }
                }
                if (col < a.Cout) { float* o = a.stats + ((long)tile_lin * a.stats_ld + col) * 2; o[0] = s1; o[1] = s2; }
            }
        }
    }
}

#undef HX_SEG_OF
#undef HX_LOAD_A
#undef HX_STORE_A
#undef HX_LOAD_B
#undef HX_STORE_B
#undef HX_LOAD_B_STEP

// ---- weight packing: OIHW fp32 (reference state_dict layout) -> split 16-bit tiles [tap][chunk][Cout_pad][hi 32 | lo 32] ----
template <typename T, int NPL>
__device__ __forceinline__ void pk_hx_elem(const PackDesc& d, T* wq, int Cout_pad, int dgrad_seg, int nch, long i) {
    // forward (dgrad_seg < 0): rows = output channels, k = concatenated input segments, each padded to KC.
    // dgrad of segment s: rows = input channels of s, k = output channels (padded to KC), taps flipped.
    const int taps = d.KS * d.KS;
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
        if (row < d.Cout && cin >= 0) v = d.w[row / d.Co_each][((long)(row % d.Co_each) * d.Cin + cin) * taps + tap] * (d.oscale ? d.oscale[row] : 1.f);
    } else {
        if (row < d.seg_C[dgrad_seg] && k < d.Cout) v = d.w[k / d.Co_each][((long)(k % d.Co_each) * d.Cin + d.seg_off[dgrad_seg] + row) * taps + (taps - 1 - tap)];
    }
    if (sizeof(T) == 2 && !is_bf16<T>::value) v *= HX_WSCALE;      // f16 forms only (see HX_WSCALE)
    const T hi = (T)v;
    // FRAGMENT-MAJOR order inside every 32-row block of a (tap, chunk) tile (round 6; same bytes, permuted): [K half kh = kk >> 4][plane][lane = 32 (kk >> 3 & 1) + row & 31][8 halves]
    // -- the 1-KB run of (row block, kh, plane) IS the B operand of one v_mfma_f32_32x32x16 (lane l: row l & 31, eight channels (l >> 5) of the K half), so the kernels that take
    // their weight fragments straight from global memory (k_conv_hx<BG>) read it with one fully coalesced 16-byte load per lane; the LDS-staged kernels undo the permutation
    // when they store their 16-byte pieces (HX_STORE_B), conv_direct.hip addresses its 16 x 16 x 32 fragments through hx_wq_piece
    T* o = wq + ((long)tap * nch + ch) * Cout_pad * (NPL * KC) + hx_wq_piece<NPL>(row, kk >> 3) * 8 + (kk & 7);
    o[0] = hi;
    if (NPL == 2) o[64 * 8] = (T)(v - (float)hi);            // (plane 1: the next 64 pieces)
}
__device__ __forceinline__ int pk_hx_nch(const PackDesc& d, int dgrad_seg) {
    int Kq = 0;
    if (dgrad_seg < 0) { for (int s = 0; s < d.nseg; s++) Kq += (d.seg_C[s] + KC - 1) / KC * KC; }
    else Kq = (d.Cout + KC - 1) / KC * KC;
    return Kq / KC;
}
template <typename T, int NPL>
__global__ void k_pack_hx(PackDesc d, T* wq, int Cout_pad, int dgrad_seg) {
    const int nch = pk_hx_nch(d, dgrad_seg);
    const long total = (long)d.KS * d.KS * nch * Cout_pad * KC;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) pk_hx_elem<T, NPL>(d, wq, Cout_pad, dgrad_seg, nch, i);
}

// one launch over a job table (pack.h): workgroup -> job by binary search over the jobs' first blocks
__global__ __launch_bounds__(256) void k_pack_jobs(const PackJob* jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const PackJob& j = jobs[lo];
    const long first = ((long)blockIdx.x - j.block0) * 256 + threadIdx.x, stride = (long)j.nblocks * 256;
    switch (j.kind) {
        case PJ_FWD: for (long i = first; i < j.total; i += stride) pk_fwd_elem(j.d, (float*)j.buf, i); break;
        case PJ_DGRAD: for (long i = first; i < j.total; i += stride) pk_dgrad_elem(j.d, j.seg, (float*)j.buf, j.p0, j.p1, i); break;
        case PJ_UNPACK: for (long i = first; i < j.total; i += stride) pk_unpack_elem(j.d, (const float*)j.buf, i); break;
        case PJ_HX_FWD_F16: { const int nch = pk_hx_nch(j.d, -1); for (long i = first; i < j.total; i += stride) pk_hx_elem<_Float16, 2>(j.d, (_Float16*)j.buf, j.p0, -1, nch, i); break; }
        case PJ_HX_DGRAD_BF16: { const int nch = pk_hx_nch(j.d, j.seg); for (long i = first; i < j.total; i += stride) pk_hx_elem<__bf16, 2>(j.d, (__bf16*)j.buf, j.p0, j.seg, nch, i); break; }
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
int pack_jobs_launch(const PackJob* jobs_dev, int njobs, int total_blocks, hipStream_t st) {
    if (njobs <= 0 || total_blocks <= 0) return 0;
    hipLaunchKernelGGL(k_pack_jobs, dim3((unsigned)total_blocks), dim3(256), 0, st, jobs_dev, njobs);
    return 0;
}
int hx_pick_bn(int cout) { return cout > 64 ? 128 : (cout > 32 ? 64 : 32); }
int g_hx_big_override = -1;      // tests: force (1) / forbid (0) the 8-wave 16x16x128 tile variant regardless of the grid size
int g_hx_bg = -1;                // tests / A-B runs: >= 0 overrides CADDY_HX_BG (which under-filled tile variants take their weight fragments straight from global memory)

// does conv_hx_try run a launch of this geometry on one of the two tile variants that carry the fused max-pool epilogue (EP = 1)?
// (the same decisions as below, incl. the test override)
bool conv_hx_pool_ok(int N, int H, int W, int Cout) {
    const int bn = hx_pick_bn(Cout);
    const long wgs = (long)N * cdiv(W, 16) * cdiv(H, 16) * (round_up(Cout, bn) / bn);
    const int force_big = g_hx_big_override;
    if (bn == 128) return force_big >= 0 ? force_big == 1 : wgs >= 384;
    if (bn == 64) return force_big == 1 || wgs >= 384;
    return false;
}

bool conv_hx_s16_ok(int N, int H, int W, int Cout) { return (Cout & 31) == 0 && conv_hx_pool_ok(N, H, W, Cout); }

// will conv_fwd_launch hand this launch to k_conv_hx (the only forward kernel that applies ConvSrc.bn_* while staging its input)?
bool conv_src_lazy_ok(const ConvArgs& a) {
    if (a.KS != 3 || !a.wq || a.precision < PREC_F16X3 || a.precision > PREC_BF16X1 || a.act == 1) return false;
    for (int s = 0; s < a.nsrc; s++) if ((a.src[s].ld & 3) || (a.src[s].sn & 3)) return false;
    return true;
}
thread_local int g_last_conv_stats_tiles = 0;
thread_local int g_last_conv_direct = 0;

// would an average-pooled launch (ConvArgs.avgpool) of this split-f16 layer run here (latency kernel, or K-split tile launch + pooling slab reduce)?  `out` need not be set.
int conv_hx_avgpool_ok(const ConvArgs& a0) {
    ConvArgs a = a0;
    a.avgpool = 1;
    return conv_hx_try(a, nullptr, true) == 1 ? 1 : 0;
}

// 1 = handled.  Requirements: 3x3, split weights present (a.wq, packed for a.precision with rows padded to hx_pick_bn(Cout)).
int conv_hx_try(const ConvArgs& a0, hipStream_t st, bool dry) {
    ConvArgs a = a0;
    if (a.KS != 3 || !a.wq || a.precision < PREC_F16X3 || a.precision > PREC_BF16X1 || a.act == 1) return 0;      // (tanh: FinalBlocks, 3 output channels -- never an hx layer)
    int kq = 0;
    for (int s = 0; s < a.nsrc; s++) { if ((a.src[s].ld & 3) || (a.src[s].sn & 3)) return -1; kq += round_up(a.src[s].C, HX_KC); }
    a.Kq = kq;
    a.out_scale = (a.precision == PREC_F16X3 || a.precision == PREC_F16X1) ? 1.0f / HX_WSCALE : 1.0f;
    int bn = hx_pick_bn(a.Cout);
    a.Cout_pad = round_up(a.Cout, bn);      // (row padding of the packed weights: independent of the tile width chosen below)
    if (a.mask && a.accumulate) return -1;
    if (a.pool_out && (a.accumulate || a.mask)) return -1;
    if (a.avgpool) {      // average-pooled result (the caller asked conv_avgpool_ok): the latency kernel's pooled epilogue, or -- below -- a K-split launch whose slab reduce pools
        if ((a.H | a.W) & 1 || a.accumulate || a.mask || a.pool_out || a.skip_out || a.stats || a.lstm || a.precision != PREC_F16X3 || a.in_s16 || a.out_s16) return dry ? 0 : -1;
        if (a.direct_ok && conv_direct_try(a, st, dry) == 1) {
            if (!dry) { g_last_conv_kernel = CK_HX_32; g_last_conv_stats_tiles = 0; g_last_conv_direct = 1; }
            return 1;
        }
    }
    if ((long)a.H * a.W * a.out_ld >= (1L << 30) || (a.res && (long)a.H * a.W * a.res_ld >= (1L << 30))) return 0;      // (the epilogue addresses one sample with 32-bit byte offsets)
    const int nchunks = kq / HX_KC;
    // under-filled wide layers (R's gate / SameBlock convolutions on 16x16 .. 32x32 maps, A, D's first stage at batch 8): 64-channel tiles on the same
    // 128-row packed weights -> twice the workgroups, two co-resident per CU hiding each other's barrier / LDS latencies (one 4-wave workgroup per CU
    // runs its serial chain of (tap, chunk) steps at ~30 % of the MFMA rate).  Measured, E/R/A/D step: 79.5 -> 78.2 ms, flat for thresholds 256 / 512 / 1024.
    if (bn == 128 && g_hx_big_override < 0 && !a.pool_out && !a.skip_out && !a.mask &&
        (long)a.N * cdiv(a.W, 16) * cdiv(a.H, 8) * (a.Cout_pad / 128) <= 256) bn = 64;
    // tiles: 128-channel layers -> 16x16 pixels x 128 channels on 8 waves with the 3-deep weight-tile ring when that fills the chip, else
    // 8x16 pixels on 4 waves (R's small feature maps); 16x16 x 64 / 32 channels for the narrower layers
    const int force_big = g_hx_big_override;                   // tests: 0 never, 1 always (128-channel layers)
    bool big = bn == 128 && (long)a.N * cdiv(a.W, 16) * cdiv(a.H, 16) * (a.Cout_pad / bn) >= 384;
    if (force_big >= 0) big = bn == 128 && force_big == 1;
    // narrower layers: 16x16-pixel tiles, or 8x16 when those would leave CUs idle (E / A on one time step's frames, R's side branches)
    const bool small_tiles = bn < 128 && force_big != 1 && (long)a.N * cdiv(a.W, 16) * cdiv(a.H, 16) * (a.Cout_pad / bn) < 384;      // (tests force the well-filled variants with 1)
    // inference launches (ConvArgs.direct_ok) and gradient launches whose 8x16 x 64-channel grid is under-filled (< 200 workgroups: it would be split over K): 4x16-pixel tiles -- twice the
    // workgroups from the pixel side, half the K slices and slabs (D's 128x128 layers of a batch-1 roll-out frame run unsplit: 256 workgroups of 18 - 36 steps, no slab
    // reduce).  Measured, roll-out: only where 4x16 tiles reach 200 workgroups 2012 -> 2067 frames/s, for every under-filled launch 2144.
    const long blocks8 = (long)a.N * cdiv(a.W, 16) * cdiv(a.H, 8) * (a.Cout_pad / bn);
    // The same for the under-filled split-bf16 launches of a training step (dgrads of E / A / R's small maps: fewer K slices = fewer atomics / slabs): E/R/A/D step
    // 66.73 / 66.79 -> 66.37 / 66.46 ms.  (Not the training FORWARD: a different K slicing changes its summation order, and the parity bounds of the closed-loop forward are
    // calibrated on the 8x16 form.)
    const bool th4 = bn == 64 && small_tiles && !a.pool_out && !a.skip_out && !a.mask && !a.stats && blocks8 < 200 &&
                     ((a.direct_ok && a.precision == PREC_F16X3 && !a.accumulate) || a.precision == PREC_BF16X3);
    const int th = th4 ? 4 : ((bn == 128 && !big) || small_tiles) ? 8 : 16;
    const int tx = cdiv(a.W, 16), ty = cdiv(a.H, th);
    const long blocks = (long)a.N * tx * ty * (a.Cout_pad / bn);
    // under-filled launches: split the channel chunks across blockIdx.z.  Accumulating launches (dgrad +=) combine with fp32 atomics; assigning
    // launches use the caller's slab scratch + the fixed-order k_split_reduce (bit-reproducible forward), as k_conv_fwd does.
    // latency-bound assigning launches (batch-1 roll-out): one launch without slabs on conv_direct.hip
    g_last_conv_direct = 0;
    if (!dry && !a.avgpool && a.direct_ok && blocks < 200 && conv_direct_try(a, st) == 1) { g_last_conv_kernel = CK_HX_32; g_last_conv_stats_tiles = 0; g_last_conv_direct = 1; return 1; }
    a.splitk = 1; a.split_stride = 0;
    bool det_accum = false;
    float* real_out = a.out; long real_sn = a.out_sn; int real_ld = a.out_ld; const float* real_bias = a.bias; const int real_act = a.act;
    const float* real_res = a.res;
    const long P = (long)a.N * a.H * a.W;
    if (blocks < 200 && nchunks >= 2 && !a.mask && !a.pool_out) {
        int want = (int)((512 + blocks - 1) / blocks);         // two workgroups per CU (inference launches, 256 / 384 / 768 / 1024: roll-out within 0.5 %)
        if (want > nchunks) want = nchunks;                     // a slice is at least one chunk (= one staged halo tile) x nine taps
        if (want > 16) want = 16;
        if (want >= 2) {
            const bool acc_ok = a.accumulate && a.act == 0 && !a.bias && !a.res;
            if (acc_ok && !a.deterministic) a.splitk = want;      // fp32 atomics in arrival order
            else if ((acc_ok || !a.accumulate) && a.split_scratch) {      // slabs + fixed-order reduce (deterministic mode: accumulating launches too; the reduce adds the old contents)
                const int ldc = round_up(a.Cout, 4);
                while (want >= 2 && (long)want * P * ldc > a.split_cap) want--;
                if (want >= 2) { a.splitk = want; a.split_stride = P * ldc; a.out = a.split_scratch; a.out_sn = (long)a.H * a.W * ldc; a.out_ld = ldc; a.bias = nullptr; a.act = 0; a.res = nullptr;
                                 det_accum = a.accumulate != 0; a.accumulate = 0; }
            }
        }
    }
    a.xcd_map = 1;
    // BatchNorm partial sums: from the epilogue of EP = 0 instances when one workgroup holds the whole K range, from the slab reduce of a split launch otherwise
    float* const stats_req = a.stats; const int stats_req_ld = a.stats_ld;
    const long stats_cap = conv_stats_tiles_cap(a.N, a.H, a.W);                // tiles the caller sized the buffer for
    if (a.stats && (a.splitk != 1 || a.mask || a.pool_out || a.skip_out || a.precision == PREC_F16X1 || a.precision == PREC_BF16X1)) a.stats = nullptr;
    g_last_conv_stats_tiles = a.stats ? (int)((long)a.N * tx * ty) : 0;
    dim3 grid((unsigned)((long)a.N * tx * ty), a.Cout_pad / bn, a.splitk);
    // 4-wave variants: 3-deep register ring of weight tiles (with one step of prefetch the next tile has ~0.4 us to arrive from L2, less than its latency under load; measured,
    // E/R/A/D step: ring on launches of <= 256 / 512 / 1024 workgroups / always: 77.2 / 75.5 / 75.8 / 75.5 ms, batch-1 roll-out frame -3 %).  The single-product study
    // precisions (PREC_*X1, tools/bench_hx.py) keep one step of prefetch.
    // round 6: the under-filled variants with the weight fragments straight into registers (template parameter BG of k_conv_hx) -- split operands, plain epilogue, fp32 output.
    // g_hx_bg / CADDY_HX_BG: bit 0 the 4 x 16 x 64 tile, bit 1 8 x 16 x 64, bit 2 8 x 16 x 32 (tests and A/B runs; default all)
    static const int bg_env = []() { const char* e = getenv("CADDY_HX_BG"); return e ? atoi(e) : 7; }();
    // ... and only where a workgroup walks at least four 32-channel chunks: the nine-deep ring's prologue and the exchange of the partial tiles are pure overhead for the 1 - 2 chunks
    // of a K-split 64-channel layer (measured alone: E res 64 -> 64 @32 15.0 -> 15.9 us, R same 272 -> 128 @16 (8 K slices) 22.5 -> 23.6 us; 4+ chunks: -10 ... -25 %)
    const int bgm = (nchunks + a.splitk - 1) / a.splitk >= 4 || g_hx_bg >= 8 ? ((g_hx_bg >= 0 ? g_hx_bg : bg_env) & 7) : 0;
#define HX_LAUNCH(T_, NPL_, EP_, IO_)                                                                                             \
    do {                                                                                                                          \
        constexpr int D_ = NPL_ == 2 ? 3 : 1;                                                                                     \
        constexpr int BG_ = (NPL_ == 2 && EP_ == 0 && ((IO_) & 2) == 0) ? 1 : 0;                                                  \
        if (big) hipLaunchKernelGGL((k_conv_hx<T_, NPL_, 16, 16, 128, 4, 2, 3, EP_, IO_>), grid, dim3(512), 0, st, a, tx, ty);    \
        else if (bn == 128) hipLaunchKernelGGL((k_conv_hx<T_, NPL_, 8, 16, 128, 2, 2, D_, EP_, IO_>), grid, dim3(256), 0, st, a, tx, ty);   \
        else if (th4) { if (bgm & 1) hipLaunchKernelGGL((k_conv_hx<T_, 2, 4, 16, 64, 2, 2, 3, 0, IO_, 1>), grid, dim3(256), 0, st, a, tx, ty);       /* (plain epilogue only) */ \
                        else hipLaunchKernelGGL((k_conv_hx<T_, 2, 4, 16, 64, 2, 2, 3, 0, IO_>), grid, dim3(256), 0, st, a, tx, ty); }   \
        else if (bn == 64 && small_tiles) { if ((bgm & 2) && BG_) hipLaunchKernelGGL((k_conv_hx<T_, NPL_, 8, 16, 64, 2, 2, D_, EP_, IO_, BG_>), grid, dim3(256), 0, st, a, tx, ty);   \
                                            else hipLaunchKernelGGL((k_conv_hx<T_, NPL_, 8, 16, 64, 2, 2, D_, EP_, IO_>), grid, dim3(256), 0, st, a, tx, ty); }   \
        else if (bn == 64) hipLaunchKernelGGL((k_conv_hx<T_, NPL_, 16, 16, 64, 4, 1, D_, EP_, IO_>), grid, dim3(256), 0, st, a, tx, ty);    \
        else if (small_tiles) { if ((bgm & 4) && BG_) hipLaunchKernelGGL((k_conv_hx<T_, NPL_, 8, 16, 32, 4, 1, D_, EP_, IO_, BG_>), grid, dim3(256), 0, st, a, tx, ty); \
                                else hipLaunchKernelGGL((k_conv_hx<T_, NPL_, 8, 16, 32, 4, 1, D_, EP_, IO_>), grid, dim3(256), 0, st, a, tx, ty); } \
        else hipLaunchKernelGGL((k_conv_hx<T_, NPL_, 16, 16, 32, 4, 1, D_, EP_, IO_>), grid, dim3(256), 0, st, a, tx, ty);                  \
    } while (0)
    if (a.avgpool && !a.split_stride) return dry ? 0 : -1;      // (a whole-K tile launch has no pooled epilogue)
    if (dry) return 1;
    const bool vgg_bwd = a.mask != nullptr;      // ReLU mask / L1 seed epilogue (VGG19 dgrad chain): split-bf16 instances with EP = 2
    const int io = (a.in_s16 ? 1 : 0) | ((a.out_s16 || a.pool_s16) ? 2 : 0);
    // round 6: a PRE-SPLIT input alone (the model's gradient tensors, written as S16-bf16 by their point-wise producers) is a property of the loader only: every split-bf16 tile
    // variant has such an instance, with the full plain epilogue (accumulate, K split, slabs) -- handled by the general launch below
    const bool ps_grad = io == 1 && a.precision == PREC_BF16X3 && !vgg_bwd && !a.pool_out && !a.skip_out;
    if (ps_grad && (a.nsrc != 1 || (a.src[0].C & 31) || a.src[0].bcast || a.src[0].bn_scale)) return -1;
    if (io && !ps_grad) {      // S16 tensors at the boundary: the two well-filled tile variants only (conv_hx_s16_ok), whole-K workgroups, plain or VGG19 epilogues
        const bool wide = big || (bn == 64 && !small_tiles);
        if (!wide || a.splitk != 1 || a.accumulate || a.res || a.stats || (a.precision != PREC_F16X3 && a.precision != PREC_BF16X3)) return -1;
        if (a.in_s16 && (a.nsrc != 1 || (a.src[0].C & 31) || a.src[0].bcast || a.src[0].bn_scale)) return -1;
        if ((a.out_s16 || a.pool_s16) && (a.Cout & 31)) return -1;
        if (a.pool_out && !a.skip_out && (a.out_s16 != 0) != (a.pool_s16 != 0)) return -1;      // one epilogue format per launch
        if (a.precision == PREC_BF16X3 && (a.pool_out || a.skip_out)) return -1;
        if (a.precision == PREC_F16X3 && vgg_bwd) return -1;
#define HX_IO_LAUNCH(T_, EP_)                                                                                                       \
        do {                                                                                                                       \
            if (big) { if (io == 1) hipLaunchKernelGGL((k_conv_hx<T_, 2, 16, 16, 128, 4, 2, 3, EP_, 1>), grid, dim3(512), 0, st, a, tx, ty);      \
                       else if (io == 2) hipLaunchKernelGGL((k_conv_hx<T_, 2, 16, 16, 128, 4, 2, 3, EP_, 2>), grid, dim3(512), 0, st, a, tx, ty); \
                       else hipLaunchKernelGGL((k_conv_hx<T_, 2, 16, 16, 128, 4, 2, 3, EP_, 3>), grid, dim3(512), 0, st, a, tx, ty); }            \
            else { constexpr int D_ = EP_ == 1 ? 1 : 3;      /* (ring depths of the plain instances) */                                  \
                   if (io == 1) hipLaunchKernelGGL((k_conv_hx<T_, 2, 16, 16, 64, 4, 1, D_, EP_, 1>), grid, dim3(256), 0, st, a, tx, ty);          \
                   else if (io == 2) hipLaunchKernelGGL((k_conv_hx<T_, 2, 16, 16, 64, 4, 1, D_, EP_, 2>), grid, dim3(256), 0, st, a, tx, ty);     \
                   else hipLaunchKernelGGL((k_conv_hx<T_, 2, 16, 16, 64, 4, 1, D_, EP_, 3>), grid, dim3(256), 0, st, a, tx, ty); }                \
        } while (0)
        if (a.precision == PREC_F16X3) { if (a.pool_out || a.skip_out) HX_IO_LAUNCH(_Float16, 1); else HX_IO_LAUNCH(_Float16, 0); }
        else if (vgg_bwd) HX_IO_LAUNCH(__bf16, 2);
        else { if (io != 1) return -1;      // (dgrad of a layer behind a max-pool: its output is the fp32 gradient of the pooled map)
               if (big) hipLaunchKernelGGL((k_conv_hx<__bf16, 2, 16, 16, 128, 4, 2, 3, 0, 1>), grid, dim3(512), 0, st, a, tx, ty);
               else hipLaunchKernelGGL((k_conv_hx<__bf16, 2, 16, 16, 64, 4, 1, 3, 0, 1>), grid, dim3(256), 0, st, a, tx, ty); }
#undef HX_IO_LAUNCH
        g_last_conv_kernel = big ? CK_HX_128_8W : CK_HX_64;
        return 1;
    }
    if (a.mask_s16 || a.seed_s16) { if (!vgg_bwd) return -1; }
    if (a.pool_out || a.skip_out) {      // VGG19 layers in front of a max-pool: the two tile variants those layers run on (perceptual.hip asks only when this holds)
        if (a.precision != PREC_F16X3 || !(big || (bn == 64 && !small_tiles))) return -1;
        if (big) hipLaunchKernelGGL((k_conv_hx<_Float16, 2, 16, 16, 128, 4, 2, 3, 1>), grid, dim3(512), 0, st, a, tx, ty);
        else hipLaunchKernelGGL((k_conv_hx<_Float16, 2, 16, 16, 64, 4, 1, 1, 1>), grid, dim3(256), 0, st, a, tx, ty);
        g_last_conv_kernel = big ? CK_HX_128_8W : CK_HX_64;
        return 1;
    }
    switch (a.precision) {
        case PREC_F16X3:
            if (vgg_bwd) return -1;      // (masks exist on the gradient side only)
            HX_LAUNCH(_Float16, 2, 0, 0);
            break;
        case PREC_BF16X3:
            if (vgg_bwd) HX_LAUNCH(__bf16, 2, 2, 0); else if (ps_grad) HX_LAUNCH(__bf16, 2, 0, 1); else HX_LAUNCH(__bf16, 2, 0, 0);
            break;
        case PREC_F16X1: HX_LAUNCH(_Float16, 1, 3, 0); break;
        default: HX_LAUNCH(__bf16, 1, 3, 0); break;
    }
#undef HX_LAUNCH
    g_last_conv_kernel = bn == 128 ? (big ? CK_HX_128_8W : CK_HX_128) : (bn == 64 ? CK_HX_64 : CK_HX_32);
    if (a.split_stride && a.lstm && !det_accum && real_act == 0 && !real_res && !stats_req && a.Cout == 4 * a.lstm->C && (a.lstm->C & 3) == 0 && (a.out_ld & 3) == 0)
        conv_split_reduce_lstm_launch(a.split_scratch, a.split_stride, a.splitk, a.out_ld, a.H * a.W, P, real_bias, *a.lstm, st);      // roll-out ConvLSTM cell: the reduce applies the cell update
    else if (a.split_stride && a.avgpool)
        conv_split_reduce_pool_launch(a.split_scratch, a.split_stride, a.splitk, a.out_ld, a.N, a.H, a.W, a.Cout, real_out, real_sn, real_ld, real_bias, real_act, real_res, a.res_sn, a.res_ld, st);
    else if (a.split_stride) conv_split_reduce_launch(a.split_scratch, a.split_stride, a.splitk, a.out_ld, a.H * a.W, P, a.Cout, real_out, real_sn, real_ld, real_bias, real_act, real_res, a.res_sn, a.res_ld, st,
                                                 stats_req, stats_req_ld, stats_cap, det_accum ? 1 : 0);
    return 1;
}

