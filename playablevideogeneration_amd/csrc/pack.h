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

// One launch for a whole list of (un)packing jobs -- the per-step re-layout of every layer's weights into the kernels' forms (exact fp32 forward / dgrad forms,
// split f16 forward / split bf16 dgrad forms of conv_hx.hip) and the unpacking of the packed weight gradients: ~150 + ~40 launches of a few microseconds each
// at the head / tail of every training step otherwise (1.0 ms + 0.3 ms of the BAIR step, all launch latency).  The job table lives in device memory; a
// workgroup finds its job by binary search over the jobs' first-block prefix.
enum { PJ_FWD = 0, PJ_DGRAD = 1, PJ_HX_FWD_F16 = 2, PJ_HX_DGRAD_BF16 = 3, PJ_UNPACK = 4 };
struct PackJob {
    PackDesc d;
    void* buf;          // destination (packing) / packed gradient source (PJ_UNPACK)
    int kind, seg;      // seg: input segment of the dgrad forms
    int p0, p1;         // PJ_DGRAD: Cd_pad, Kd; PJ_HX_*: rows_pad
    long total;         // elements of the packed form
    int block0, nblocks;
};
int pack_jobs_launch(const PackJob* jobs_dev, int njobs, int total_blocks, hipStream_t st);      // conv_hx.hip
