#pragma once
#include "common.h"

// One convolution layer's weights at the boundary (reference layout) and its decomposition into input segments.
struct PackDesc {
    const float* w[4];   // up to 4 OIHW tensors stacked along O (ConvLSTM gates i,f,o,g: convolutional_lstm_cell.py:22-25)
    float* gw[4];        // their gradients (same layout)
    int nw, Co_each;
    int Cin;             // reference input channels (= sum of seg_C)
    int KS;
    int nseg, seg_off[CONV_MAX_SRC], seg_C[CONV_MAX_SRC], seg_Cpad[CONV_MAX_SRC];
    int Cout, Cout_pad, Ktot;
    const float* oscale;  // optional per-output-channel factor applied while packing the FORWARD forms (eval-mode BatchNorm folded into the conv for roll-outs)
};
int pack_fwd(const PackDesc& d, float* wp, hipStream_t st);
int pack_dgrad(const PackDesc& d, int seg, float* wpd, int Cd_pad, int Kd, hipStream_t st);
int unpack_wgrad(const PackDesc& d, const float* dwp, hipStream_t st);
int adam_launch(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step, float gscale, hipStream_t st);
