// Element-wise bodies of the weight (un)packing kernels, shared by the per-layer kernels (pack.hip) and the one-launch job kernel (conv_hx.hip).
#pragma once
#include "pack.h"

__device__ __forceinline__ bool pk_k_to_cin(const PackDesc& d, int k, int* cin) {
    int base = 0;
    for (int s = 0; s < d.nseg; s++) {
        if (k < base + d.seg_Cpad[s]) { int c = k - base; if (c >= d.seg_C[s]) return false; *cin = d.seg_off[s] + c; return true; }
        base += d.seg_Cpad[s];
    }
    return false;
}
__device__ __forceinline__ void pk_fwd_elem(const PackDesc& d, float* wp, long i) {
    int k = (int)(i % d.Ktot); long r = i / d.Ktot; int o = (int)(r % d.Cout_pad); int tap = (int)(r / d.Cout_pad);
    float v = 0.f; int cin;
    if (o < d.Cout && pk_k_to_cin(d, k, &cin)) v = d.w[o / d.Co_each][((long)(o % d.Co_each) * d.Cin + cin) * d.KS * d.KS + tap] * (d.oscale ? d.oscale[o] : 1.f);
    wp[i] = v;
}
// dgrad weights of segment `seg`: wpd[tap'][c][o] = W[o][seg_off+c][KS*KS-1-tap']   (flip both spatial axes)
__device__ __forceinline__ void pk_dgrad_elem(const PackDesc& d, int seg, float* wpd, int Cd_pad, int Kd, long i) {
    const int taps = d.KS * d.KS;
    int o = (int)(i % Kd); long r = i / Kd; int c = (int)(r % Cd_pad); int tap = (int)(r / Cd_pad);
    float v = 0.f;
    if (o < d.Cout && c < d.seg_C[seg]) v = d.w[o / d.Co_each][((long)(o % d.Co_each) * d.Cin + d.seg_off[seg] + c) * taps + (taps - 1 - tap)];
    wpd[i] = v;
}
__device__ __forceinline__ void pk_unpack_elem(const PackDesc& d, const float* dwp, long i) {
    const int taps = d.KS * d.KS;
    int k = (int)(i % d.Ktot); long r = i / d.Ktot; int o = (int)(r % d.Cout_pad); int tap = (int)(r / d.Cout_pad);
    int cin;
    if (o < d.Cout && pk_k_to_cin(d, k, &cin)) d.gw[o / d.Co_each][((long)(o % d.Co_each) * d.Cin + cin) * taps + tap] = dwp[i];
}
