// Small-tensor kernels of the hot path: action-network tail (action_network.py:86-118), Gumbel-softmax sampling
// (gumbel_softmax.py:25-72), centroid EMA + variations (centroid_estimator.py:38-94) and the fused loss terms with their
// gradient seeds (training/losses.py; training/trainer.py:447-500).  N = B*T samples (~128): latency-bound, not
// roofline-relevant; the L1 / MSE losses over frames and states are HBM-bound streaming reductions.
#include "common.h"
#include "head.h"
#include "perceptual.h"

namespace {

__device__ __forceinline__ double block_sum(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double t = 0;
    for (int i = 0; i < nw; i++) t += sh[i];
    return t;
}

// ---- A tail, part 1: mu / |raw| / sampled action states, one thread per (n, d) --------------------------------------
__global__ void k_head_fc(HeadBufs h, HeadParams p, int NBT) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NBT * p.Da) return;
    int n = i / p.Da, d = i - n * p.Da;
    const float* f = h.feat + (long)n * p.F;
    float m = p.bm[d], r = p.bv[d];
    for (int k = 0; k < p.F; k++) { m = fmaf(p.Wm[d * p.F + k], f[k], m); r = fmaf(p.Wv[d * p.F + k], f[k], r); }
    float var = fabsf(r);
    h.mu[i] = m; h.raw[i] = r;
    h.sdist[(long)n * 2 * p.Da + d] = m; h.sdist[(long)n * 2 * p.Da + p.Da + d] = var;
    h.ssamp[i] = h.eps_s[i] * sqrtf(var) + m;
}
// ---- part 2: direction distribution, sampled direction, logits; one thread per (b, t<T-1) ---------------------------
__global__ void k_head_dirs(HeadBufs h, HeadParams p, int B, int T) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= B * (T - 1)) return;
    int b = n / (T - 1), t = n - b * (T - 1);
    int s0 = (b * T + t) * p.Da, s1 = s0 + p.Da;
    float dv[8];
    for (int d = 0; d < p.Da; d++) {
        float dmu = h.mu[s1 + d] - h.mu[s0 + d];
        float dvar = fabsf(h.raw[s1 + d]) + fabsf(h.raw[s0 + d]);
        h.ddist[(long)n * 2 * p.Da + d] = dmu; h.ddist[(long)n * 2 * p.Da + p.Da + d] = dvar;
        dv[d] = h.eps_d[n * p.Da + d] * sqrtf(dvar) + dmu;
        h.dirs[n * p.Da + d] = dv[d];
    }
    for (int k = 0; k < p.K; k++) {
        float l = p.bf[k];
        for (int d = 0; d < p.Da; d++) l = fmaf(p.Wf[k * p.Da + d], dv[d], l);
        h.logits[n * p.K + k] = l;
    }
}
// ---- sampling, part 1: softmax / log-softmax and the centroid-EMA sums (centroid_estimator.py:61-63).  Single block. ----------
// cen_sums[k*Da+d] = sum_n p_nk * dmu_nd, cen_sums[K*Da + k] = sum_n p_nk.  Under data parallelism these sums are all-reduced
// between part 1 and part 2 so that every rank applies the global-batch estimate (collective 3).
__global__ void k_head_probs(HeadBufs h, HeadParams p, int NS, float* cen_sums) {
    const int K = p.K, Da = p.Da;
    for (int n = threadIdx.x; n < NS; n += blockDim.x) {
        const float* l = h.logits + n * K;
        float mx = l[0];
        for (int k = 1; k < K; k++) mx = fmaxf(mx, l[k]);
        float se = 0.f;
        for (int k = 0; k < K; k++) se += expf(l[k] - mx);
        float lse = mx + logf(se);
        for (int k = 0; k < K; k++) { h.logp[n * K + k] = l[k] - lse; h.prob[n * K + k] = expf(l[k] - lse); }
    }
    __syncthreads();
    if (threadIdx.x < K * Da) {
        int k = threadIdx.x / Da, d = threadIdx.x - k * Da;
        float num = 0.f;
        for (int n = 0; n < NS; n++) num += h.prob[n * K + k] * h.ddist[(long)n * 2 * Da + d];
        cen_sums[k * Da + d] = num;
    } else if (threadIdx.x < K * Da + K) {
        int k = threadIdx.x - K * Da;
        float den = 0.f;
        for (int n = 0; n < NS; n++) den += h.prob[n * K + k];
        cen_sums[K * Da + k] = den;
    }
}
// ---- sampling, part 2: centroid EMA (train), Gumbel sample, variations, arg-max.  Single block. ---------------------------------
__global__ void k_head_sample(HeadBufs h, HeadParams p, SampleCfg c, int NS, const float* cen_sums) {
    __shared__ float cen[16 * 8];
    const int K = p.K, Da = p.Da;
    if (threadIdx.x < K * Da) {
        int k = threadIdx.x / Da, d = threadIdx.x - k * Da;
        float cv = c.centroids[k * Da + d];
        if (c.training) {   // centroid_estimator.py:61-68 (means of the direction distribution, soft assignments)
            cv = cv * (1.f - c.alpha) + (cen_sums[k * Da + d] / cen_sums[K * Da + k]) * c.alpha;
            c.centroids[k * Da + d] = cv;
        }
        cen[k * Da + d] = cv;
        h.cen_used[k * Da + d] = cv;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < NS; n += blockDim.x) {
        float y[16];
        if (c.mode == 2) {            // externally supplied samples (evaluation action_sampler, model.py:173-174)
            for (int k = 0; k < K; k++) y[k] = c.samples_in[n * K + k];
        } else if (c.mode == 1) {     // Gumbel-softmax (gumbel_softmax.py:33-45)
            float z[16], mx = -1e30f;
            for (int k = 0; k < K; k++) {
                float g = -logf(-logf(h.unif[n * K + k] + 1e-20f) + 1e-20f);
                z[k] = (h.logp[n * K + k] + g) / c.tau;
                mx = fmaxf(mx, z[k]);
            }
            float se = 0.f;
            for (int k = 0; k < K; k++) { y[k] = expf(z[k] - mx); se += y[k]; }
            for (int k = 0; k < K; k++) y[k] /= se;
        } else {
            for (int k = 0; k < K; k++) y[k] = h.prob[n * K + k];
        }
        int am = 0;
        for (int k = 1; k < K; k++) if (y[k] > y[am]) am = k;
        for (int k = 0; k < K; k++) h.ysoft[n * K + k] = y[k];
        if (c.mode == 1 && c.hard) { for (int k = 0; k < K; k++) y[k] = (k == am) ? 1.f : 0.f; }   // straight-through value
        h.selected[n] = am;
        float* aux = h.aux + (long)n * AUX_LD;
        for (int k = 0; k < K; k++) { aux[k] = y[k]; h.samples[n * K + k] = y[k]; }
        for (int d = 0; d < Da; d++) {
            float v;
            if (c.variations_in) v = c.variations_in[n * Da + d];
            else {
                v = 0.f;
                for (int k = 0; k < K; k++) v += y[k] * (h.dirs[n * Da + d] - cen[k * Da + d]);
                if (!c.use_variations) v = v * 0.f;
            }
            aux[K + d] = v; h.variations[n * Da + d] = v;
        }
        for (int k = K + Da; k < AUX_LD; k++) aux[k] = 0.f;
    }
}

// ---- backward, phase 1: per (b, t<T-1): sampling -> logits -> direction sample -> (d_dmu, d_dvar) ------------------
__global__ void k_head_bwd1(HeadBufs h, HeadParams p, SampleCfg c, int NS, int first_call) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= NS) return;
    const int K = p.K, Da = p.Da;
    float dl[16], dd[8];
    for (int k = 0; k < K; k++) dl[k] = h.d_logits[n * K + k];
    for (int d = 0; d < Da; d++) dd[d] = 0.f;
    if (first_call && c.mode != 2) {
        const float* ga = h.d_aux + (long)n * AUX_LD;
        float da[16];
        for (int k = 0; k < K; k++) da[k] = ga[k];
        if (c.use_variations && !c.variations_in) {   // v = sum_k a_k (d - c_k)
            float sy = 0.f;
            for (int k = 0; k < K; k++) sy += h.samples[n * K + k];
            for (int d = 0; d < Da; d++) {
                float gv = ga[K + d];
                dd[d] += gv * sy;
                for (int k = 0; k < K; k++) da[k] += gv * (h.dirs[n * Da + d] - h.cen_used[k * Da + d]);
            }
        }
        const float* y = h.ysoft + n * K;
        float dot = 0.f;
        for (int k = 0; k < K; k++) dot += y[k] * da[k];
        if (c.mode == 1) {   // y = softmax((logp+g)/tau); logp = log_softmax(logits)
            float dz[16], sdz = 0.f;
            for (int k = 0; k < K; k++) { dz[k] = y[k] * (da[k] - dot) / c.tau; sdz += dz[k]; }
            for (int k = 0; k < K; k++) dl[k] += dz[k] - h.prob[n * K + k] * sdz;
        } else {             // samples = softmax(logits)
            for (int k = 0; k < K; k++) dl[k] += y[k] * (da[k] - dot);
        }
    }
    for (int k = 0; k < K; k++) {      // (the final_fc parameter gradients are summed over the samples in a fixed order by k_head_bwd3, from the d(logits) left here)
        h.d_logits[n * K + k] = dl[k];
        for (int d = 0; d < Da; d++) dd[d] += p.Wf[k * Da + d] * dl[k];
    }
    for (int d = 0; d < Da; d++) {
        float dvar = h.ddist[(long)n * 2 * Da + Da + d];
        h.g_dmu[n * Da + d] = dd[d] + h.d_ddist[(long)n * 2 * Da + d];
        h.g_dvar[n * Da + d] = dd[d] * h.eps_d[n * Da + d] * 0.5f / sqrtf(dvar) + h.d_ddist[(long)n * 2 * Da + Da + d];
    }
}
// ---- phase 2: per (b, t): fold pred/succ grads, abs, FC backward -> d_feat + FC weight grads ------------------------
// (round 5: one thread per (sample, feature) instead of per sample, one wave per parameter instead of one thread: the two launches were 86 + 92 us of serial loops on 2 - 4
//  workgroups, twice per step on the BPTT chain)
__global__ __launch_bounds__(256) void k_head_bwd2(HeadBufs h, HeadParams p, int B, int T) {
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= (long)B * T * p.F) return;
    const int n = (int)(idx / p.F), k = (int)(idx - (long)n * p.F);
    const int Da = p.Da;
    const int b = n / T, t = n - b * T;
    float acc = 0.f;
    for (int d = 0; d < Da; d++) {
        float gmu = h.d_sdist[(long)n * 2 * Da + d], gvar = h.d_sdist[(long)n * 2 * Da + Da + d];
        if (t >= 1) { int m = b * (T - 1) + t - 1; gmu += h.g_dmu[m * Da + d]; gvar += h.g_dvar[m * Da + d]; }
        if (t < T - 1) { int m = b * (T - 1) + t; gmu -= h.g_dmu[m * Da + d]; gvar += h.g_dvar[m * Da + d]; }
        const float raw = h.raw[n * Da + d];
        const float gr = gvar * (raw > 0.f ? 1.f : (raw < 0.f ? -1.f : 0.f));
        if (k == 0) { h.g_mu[n * Da + d] = gmu; h.g_raw[n * Da + d] = gr; }
        acc += p.Wm[d * p.F + k] * gmu + p.Wv[d * p.F + k] * gr;
    }
    h.d_feat[(long)n * p.F + k] = acc;
}
// FC parameter gradients: one wave per output, lane l sums the samples l, l + 64, ... in order, the 64 partial sums meet in a fixed butterfly (no atomics: bit-reproducible)
__device__ __forceinline__ double wave_sum_fixed(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// (fp64 partial sums: mean_fc.bias' gradient is a cancelling sum of +-1e-4 terms that ends near 3e-8 on the reference's golden cases -- in fp32 its value is the summation order)
__global__ __launch_bounds__(256) void k_head_bwd3(HeadBufs h, HeadParams p, int NBT, int NS) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= p.Da * p.F) return;           // (wave-uniform)
    const int d = i / p.F, k = i - d * p.F;
    double am = 0.0, av = 0.0;
    for (int n = lane; n < NBT; n += 64) { const double f = h.feat[(long)n * p.F + k]; am += (double)h.g_mu[n * p.Da + d] * f; av += (double)h.g_raw[n * p.Da + d] * f; }
    am = wave_sum_fixed(am); av = wave_sum_fixed(av);
    if (lane == 0) { p.dWm[i] += (float)am; p.dWv[i] += (float)av; }
    if (i < p.Da) {                        // mean_fc / variance_fc biases
        double bm = 0.0, bv = 0.0;
        for (int n = lane; n < NBT; n += 64) { bm += h.g_mu[n * p.Da + i]; bv += h.g_raw[n * p.Da + i]; }
        bm = wave_sum_fixed(bm); bv = wave_sum_fixed(bv);
        if (lane == 0) { p.dbm[i] += (float)bm; p.dbv[i] += (float)bv; }
    }
    if (i < p.K * p.Da) {                  // final_fc weight (K, Da) and bias from the d(logits) k_head_bwd1 finalised (K * Da <= 128 <= Da * F)
        const int kk = i / p.Da, dd = i - kk * p.Da;
        double w = 0.0, bsum = 0.0;
        for (int n = lane; n < NS; n += 64) { const double g = h.d_logits[n * p.K + kk]; w += g * (double)h.dirs[n * p.Da + dd]; bsum += g; }
        w = wave_sum_fixed(w); bsum = wave_sum_fixed(bsum);
        if (lane == 0) { p.dWf[i] += (float)w; if (dd == 0) p.dbf[kk] += (float)bsum; }
    }
}

// ---- losses ------------------------------------------------------------------------------------------------------------
// L1 between the resized ground-truth frame and a reconstruction (ObservationsLoss, losses.py:61-118): bilinear with
// align_corners=False at integer factors 1 / 2 / 4 == identity / 2x2 mean / mean of the central 2x2 of each 4x4 block.
__global__ __launch_bounds__(256) void k_loss_l1(TV gt, TV rec, TV drec, int f, int t_off, int Tobs, int Trec, float gscale, double* acc, float* gt_out) {
    __shared__ double sh[8];
    const int HW = rec.H * rec.W;
    const long npix = (long)rec.N * HW;
    double s = 0.0;
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < npix; q += (long)gridDim.x * blockDim.x) {
        long fr = q / HW; int rem = (int)(q - fr * HW); int y = rem / rec.W, x = rem - y * rec.W;
        long b = fr / Trec, t = fr - b * Trec;
        const float* g = gt.p + (b * Tobs + t + t_off) * gt.sn;
        const float* r = rec.p + fr * rec.sn + (long)rem * rec.ld;
        float* dr = drec.p + fr * drec.sn + (long)rem * drec.ld;
        for (int ch = 0; ch < 3; ch++) {
            float gv;
            if (f == 1) gv = g[((long)y * gt.W + x) * gt.ld + ch];
            else {
                int y0 = f == 2 ? 2 * y : 4 * y + 1, x0 = f == 2 ? 2 * x : 4 * x + 1;
                const float* gp = g + ((long)y0 * gt.W + x0) * gt.ld + ch;
                gv = 0.5f * (0.5f * gp[0] + 0.5f * gp[gt.ld]) + 0.5f * (0.5f * gp[(long)gt.W * gt.ld] + 0.5f * gp[(long)(gt.W + 1) * gt.ld]);
            }
            if (gt_out) gt_out[q * 4 + ch] = gv;       // the resized ground truth is also the input of the VGG19 ground-truth branch (perceptual.hip)
            float d = r[ch] - gv;
            s += fabsf(d);
            dr[ch] += gscale * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) atomicAdd(acc, s);
}
// MSE(a.detach(), b) over the first C channels (StatesLoss / HiddenStatesLoss, losses.py:14-53); db += gscale*2*(b-a)
__global__ __launch_bounds__(256) void k_loss_mse(TV a, TV b, TV db, float gscale, double* acc) {
    __shared__ double sh[8];
    const int HW = a.H * a.W, C = a.C;
    const long items = (long)a.N * HW * C;
    double s = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < items; i += (long)gridDim.x * blockDim.x) {
        long q = i / C; int c = (int)(i - q * C);
        long n = q / HW; long pix = q - n * HW;
        float d = b.p[n * b.sn + pix * b.ld + c] - a.p[n * a.sn + pix * a.ld + c];
        s += (double)d * d;
        db.p[n * db.sn + pix * db.ld + c] += gscale * 2.f * d;
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) atomicAdd(acc, s);
}
// joint matrix P = sum_n p_n q_n^T of the mutual-information loss (losses.py:243-262); all-reduced across ranks under data
// parallelism BEFORE symmetrise / normalise so the loss equals the reference's global-batch value (collective 2)
__global__ void k_joint_matrix(SmallLossArgs a) {
    const int K = a.K;
    if (threadIdx.x < K * K) {
        int i = threadIdx.x / K, j = threadIdx.x - i * K;
        float s = 0.f;
        for (int n = 0; n < a.NS; n++) s += a.p[n * K + i] * a.q[n * K + j];
        a.Pbuf[i * K + j] = s;
    }
}
// entropy + KL(dir || N(0,1)) + (smooth) mutual information + KL(general gaussian); single block; writes gradient seeds.
__global__ void k_loss_small(SmallLossArgs a) {
    __shared__ float Pm[16 * 16], Gm[16 * 16], rowv[16], colv[16];
    __shared__ double sh[8];
    __shared__ float Ssum;
    const int K = a.K, Da = a.Da, NS = a.NS, NT = a.NT;
    const float FEPS = 2.220446049250313e-16f;   // sys.float_info.epsilon (losses.py:270)
    if (threadIdx.x < K * K) { int i = threadIdx.x / K, j = threadIdx.x - i * K; Pm[i * 16 + j] = a.Pbuf[i * K + j]; }   // (globally reduced) joint matrix
    __syncthreads();
    if (threadIdx.x == 0) { float s = 0.f; for (int i = 0; i < K; i++) for (int j = 0; j < K; j++) s += Pm[i * 16 + j]; Ssum = s; }
    __syncthreads();
    float sym = 0.f, mp = 0.f;
    int mi = threadIdx.x / K, mj = threadIdx.x - mi * K;
    if (threadIdx.x < K * K) {
        sym = 0.5f * (Pm[mi * 16 + mj] + Pm[mj * 16 + mi]);
        mp = sym / Ssum;
        if (a.ema) { mp = a.ema[mi * K + mj] * (1.f - a.ema_alpha) + mp * a.ema_alpha; }
    }
    __syncthreads();
    if (threadIdx.x < K * K) { Gm[mi * 16 + mj] = mp; if (a.ema && a.update_ema) a.ema[mi * K + mj] = mp; }
    __syncthreads();
    if (threadIdx.x < K) { float r = 0.f, c = 0.f; for (int j = 0; j < K; j++) { r += Gm[threadIdx.x * 16 + j]; c += Gm[j * 16 + threadIdx.x]; } rowv[threadIdx.x] = r; colv[threadIdx.x] = c; }
    __syncthreads();
    double mi_term = 0.0;
    float Mc = 0.f, Rc = 0.f, Cc = 0.f;
    if (threadIdx.x < K * K) {
        Mc = mp < FEPS ? FEPS : mp; Rc = rowv[mi] < FEPS ? FEPS : rowv[mi]; Cc = colv[mj] < FEPS ? FEPS : colv[mj];
        mi_term = -(double)(Mc * (logf(Mc) - a.mi_lamb * logf(Rc) - a.mi_lamb * logf(Cc)));
    }
    double mi_loss = block_sum(mi_term, sh);
    __syncthreads();
    if (threadIdx.x < K * K) Pm[mi * 16 + mj] = Mc;   // clamped matrix
    __syncthreads();
    if (threadIdx.x < K * K) {
        float g = 0.f;
        if (mp >= FEPS) g += -(logf(Mc) - a.mi_lamb * logf(Rc) - a.mi_lamb * logf(Cc)) - 1.f;
        if (rowv[mi] >= FEPS) { float s = 0.f; for (int k = 0; k < K; k++) s += Pm[mi * 16 + k]; g += a.mi_lamb * s / Rc; }
        if (colv[mj] >= FEPS) { float s = 0.f; for (int k = 0; k < K; k++) s += Pm[k * 16 + mj]; g += a.mi_lamb * s / Cc; }
        Gm[mi * 16 + mj] = g * (a.ema ? a.ema_alpha : 1.f) * a.w_mi * a.mi_grad_scale;     // d total / d m (x world size: the gradient all-reduce averages)
    }
    __syncthreads();
    // m = sym / S : g_sym = g_m / S - sum(g_m * sym) / S^2   (sym recomputed from P kept in registers: sym)
    double gs = block_sum(threadIdx.x < K * K ? (double)Gm[mi * 16 + mj] * sym : 0.0, sh);
    __syncthreads();
    float gsym = 0.f;
    if (threadIdx.x < K * K) gsym = Gm[mi * 16 + mj] / Ssum - (float)gs / (Ssum * Ssum);
    __syncthreads();
    if (threadIdx.x < K * K) Pm[mi * 16 + mj] = gsym;
    __syncthreads();
    if (threadIdx.x < K * K) Gm[mi * 16 + mj] = 0.5f * (Pm[mi * 16 + mj] + Pm[mj * 16 + mi]);   // g_P
    __syncthreads();
    // per-sample terms
    double ent = 0.0, kl = 0.0;
    for (int n = threadIdx.x; n < NS; n += blockDim.x) {
        const float* pp = a.p + n * K; const float* qq = a.q + n * K; const float* lp = a.logp + n * K;
        float gp[16], gq[16], dotp = 0.f, dotq = 0.f, h = 0.f;
        for (int i = 0; i < K; i++) {
            float s1 = 0.f, s2 = 0.f;
            for (int j = 0; j < K; j++) { s1 += Gm[i * 16 + j] * qq[j]; s2 += Gm[j * 16 + i] * pp[j]; }
            gp[i] = s1; gq[i] = s2; dotp += pp[i] * s1; dotq += qq[i] * s2; h += pp[i] * lp[i];
        }
        ent += -(double)h;
        for (int i = 0; i < K; i++) {
            float ge = -(a.w_entropy / NS) * pp[i] * (lp[i] - h);
            a.d_logits[n * K + i] += pp[i] * (gp[i] - dotp) + ge;
            a.d_logits_r[n * K + i] += qq[i] * (gq[i] - dotq);
        }
        for (int d = 0; d < Da; d++) {
            float mu = a.ddist[(long)n * 2 * Da + d], var = a.ddist[(long)n * 2 * Da + Da + d];
            kl += (double)(1.f + logf(var) - mu * mu - var);
            a.d_ddist[(long)n * 2 * Da + d] += a.w_dirkl * mu / NS;
            a.d_ddist[(long)n * 2 * Da + Da + d] += a.w_dirkl * (-0.5f) * (1.f / var - 1.f) / NS;
        }
    }
    double skl = 0.0;
    for (int n = threadIdx.x; n < NT; n += blockDim.x) {
        for (int d = 0; d < Da; d++) {
            float mu = a.sdist_r[(long)n * 2 * Da + d], var = a.sdist_r[(long)n * 2 * Da + Da + d];
            float rmu = a.sdist[(long)n * 2 * Da + d], rvar = a.sdist[(long)n * 2 * Da + Da + d];
            float lv = logf(var), rlv = logf(rvar);
            float cv = fmaxf(var, 0.05f), crv = fmaxf(rvar, 0.05f);
            skl += (double)(rlv - lv - 1.f + cv / crv + (rmu - mu) * (rmu - mu) / crv);
            a.d_sdist_r[(long)n * 2 * Da + d] += a.w_statekl * (mu - rmu) / crv / NT;
        }
    }
    ent = block_sum(ent, sh); __syncthreads();
    kl = block_sum(kl, sh); __syncthreads();
    skl = block_sum(skl, sh);
    if (threadIdx.x == 0) {
        a.acc[LOSS_ENTROPY] = ent / NS;
        a.acc[LOSS_DIRKL] = -0.5 * kl / NS;
        a.acc[LOSS_MI] = mi_loss;
        a.acc[LOSS_STATEKL] = 0.5 * skl / NT;
    }
}
__global__ void k_loss_finalize(double* acc, LossWeights w, double n0, double n1, double n2, double nstates, double nhidden, VggLevels lv, int have_vgg) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double r0 = acc[LOSS_L1_R0] / n0, r1 = acc[LOSS_L1_R1] / n1, r2 = acc[LOSS_L1_R2] / n2;
    acc[LOSS_L1_R0] = r0; acc[LOSS_L1_R1] = r1; acc[LOSS_L1_R2] = r2;
    acc[LOSS_REC] = (r0 + r1 + r2) / 3.0;
    acc[LOSS_STATES] = acc[LOSS_STATES] / nstates;
    acc[LOSS_HIDDEN] = nhidden > 0 ? acc[LOSS_HIDDEN] / nhidden : 0.0;
    acc[LOSS_TOTAL] = w.rec * acc[LOSS_REC] + w.states * acc[LOSS_STATES] + w.entropy * acc[LOSS_ENTROPY] + w.dir_kl * acc[LOSS_DIRKL] +
                      w.mi * acc[LOSS_MI] + w.state_kl * acc[LOSS_STATEKL] + w.hidden * acc[LOSS_HIDDEN];
    if (have_vgg) {
        // trainer.py:447-466: float64 accumulators over the resolutions.  Level 0 handed to sum_loss_components IS the total (in-place
        // aliasing, losses.py:483-487), hence term_r = lambda * (total_r + l1 + l2 + l3 + l4) and the logged l0 equals the total.
        double avg = 0.0, term = 0.0;
        for (int r = 0; r < 3; r++) {
            double* s = acc + LOSS_PERC_R0 + 6 * r;
            double tot = 0.0, rest = 0.0;
            for (int l = 0; l < 5; l++) { s[1 + l] = s[1 + l] / lv.numel[r][l]; tot += s[1 + l]; if (l > 0) rest += s[1 + l]; }
            s[0] = tot; s[1] = tot;
            avg += tot; term += w.perceptual * (tot + rest);
        }
        acc[LOSS_PERCEPTUAL] = avg / 3.0; acc[LOSS_PERCEPTUAL_TERM] = term / 3.0;
        acc[LOSS_TOTAL] += term / 3.0;
    }
}
// f16 range guards of the forward pass (one word per layer, common.h: ConvArgs.sat_flag): slot = 1 if any layer clamped a value; a NaN among the clamped values makes the total NaN --
// v_med3 turned it into a finite operand, the reference would have propagated it into every loss
__global__ void k_report_flag(unsigned* flags, int n, double* slot, double* total) {
    // flags[0 .. n): the words the forward kernels of THIS step raised; flags[n .. 2n): everything raised since the last poll (caddy_f16_saturated).  The step's bits are reported
    // once and moved to the sticky half: a transient overflow marks one loss call, not every later one, and the poll still finds the layer.
    unsigned any = 0;
    for (int i = threadIdx.x; i < n; i += 64) { const unsigned f = flags[i]; any |= f; if (f) { flags[n + i] |= f; flags[i] = 0u; } }
    for (int o = 32; o > 0; o >>= 1) any |= __shfl_xor(any, o);
    if (threadIdx.x == 0) { *slot = (any & 1u) ? 1.0 : 0.0; if (any & 2u) *total = __builtin_nan(""); }
}
// Evaluation, per frame (evaluation/evaluator.py:192,194: SequenceLossEvaluator over ObservationsLoss / StatesLoss): acc[n] += sum over image n of |a - b| (sq = 0) or
// (a - b)^2 (sq = 1) over the first C channels; image n of `a` is frame (n / Tb) * Ta + n % Tb + a_off of the (B, Ta) sequence, image n of `b` is n.  blockIdx.y = n.
__global__ __launch_bounds__(256) void k_diff_per_frame(TV a, int Ta, int a_off, TV b, int Tb, int C, int sq, double* acc) {
    __shared__ double sh[4];
    const int n = blockIdx.y, HW = b.H * b.W;
    const float* ap = a.p + ((long)(n / Tb) * Ta + n % Tb + a_off) * a.sn;
    const float* bp = b.p + (long)n * b.sn;
    double s = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < (long)HW * C; i += (long)gridDim.x * 256) {
        const long pix = i / C; const int c = (int)(i - pix * C);
        const float d = ap[pix * a.ld + c] - bp[pix * b.ld + c];
        s += sq ? (double)d * d : (double)fabsf(d);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc + n, sh[0] + sh[1] + sh[2] + sh[3]);
}
__global__ void k_softmax_rows(const float* logits, float* prob, float* logp, int NS, int K) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= NS) return;
    const float* l = logits + n * K;
    float mx = l[0];
    for (int k = 1; k < K; k++) mx = fmaxf(mx, l[k]);
    float se = 0.f;
    for (int k = 0; k < K; k++) se += expf(l[k] - mx);
    float lse = mx + logf(se);
    for (int k = 0; k < K; k++) { if (logp) logp[n * K + k] = l[k] - lse; prob[n * K + k] = expf(l[k] - lse); }
}
}  // namespace

// Roll-out boundary kernels (model.py:570-607): ONE launch in front of the captured per-frame graph -- the caller's (3S, H, W) observation into the static NHWC
// buffer the graph reads, and the one-hot action + variation row -- and ONE behind it -- the (H, W, 3|pitch 4) frame into the caller's (3, H, W) frame and
// obs' = cat[frame, observation[:-3]] (model.py:605).  They replace three device-to-device copies, the two layout kernels, two copy kernels and k_set_aux.
__global__ __launch_bounds__(256) void k_rollout_in(const float* obs, float* o, int HW, int C, int ld, float* aux, int action, const float* variation, int K, int Da) {
    if (blockIdx.x == 0 && threadIdx.x < AUX_LD) {
        const int i = threadIdx.x;
        aux[i] = i == action ? 1.f : ((variation && i >= K && i < K + Da) ? variation[i - K] : 0.f);
    }
    for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
        float* q = o + p * ld;
        for (int c = 0; c < C; c++) q[c] = obs[(long)c * HW + p];
        for (int c = C; c < ld; c++) q[c] = 0.f;
    }
}
__global__ __launch_bounds__(256) void k_rollout_out(const float* f, int fld, const float* obs, float* frame_out, float* obs_out, int HW, int C) {
    for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
        const float* q = f + p * fld;
        const float r = q[0], g = q[1], b = q[2];
        frame_out[p] = r; frame_out[HW + p] = g; frame_out[2L * HW + p] = b;
        if (obs_out) {
            obs_out[p] = r; obs_out[HW + p] = g; obs_out[2L * HW + p] = b;
            for (int c = 3; c < C; c++) obs_out[(long)c * HW + p] = obs[(long)(c - 3) * HW + p];
        }
    }
}
int head_rollout_in(const float* obs, float* o_nhwc, int HW, int C, int ld, float* aux, int action, const float* variation, int K, int Da, hipStream_t st) {
    hipLaunchKernelGGL(k_rollout_in, dim3(cdiv(HW, 256)), dim3(256), 0, st, obs, o_nhwc, HW, C, ld, aux, action, variation, K, Da);
    return 0;
}
int head_rollout_out(const float* f_nhwc, int fld, const float* obs, float* frame_out, float* obs_out, int HW, int C, hipStream_t st) {
    hipLaunchKernelGGL(k_rollout_out, dim3(cdiv(HW, 256)), dim3(256), 0, st, f_nhwc, fld, obs, frame_out, obs_out, HW, C);
    return 0;
}
int head_softmax(const float* logits, float* prob, float* logp, int NS, int K, hipStream_t st) {
    hipLaunchKernelGGL(k_softmax_rows, dim3(cdiv(NS, 64)), dim3(64), 0, st, logits, prob, logp, NS, K);
    return 0;
}
int head_forward(const HeadBufs& h, const HeadParams& p, int B, int T, hipStream_t st) {
    hipLaunchKernelGGL(k_head_fc, dim3(cdiv((long)B * T * p.Da, 128)), dim3(128), 0, st, h, p, B * T);
    hipLaunchKernelGGL(k_head_dirs, dim3(cdiv((long)B * (T - 1), 64)), dim3(64), 0, st, h, p, B, T);
    return 0;
}
int head_sample(const HeadBufs& h, const HeadParams& p, const SampleCfg& c, int NS, float* cen_sums, allreduce_hook_t hook, void* user, const SamplerHooks* sh, hipStream_t st) {
    if (p.K > 16 || p.Da > 8 || p.K + p.Da > AUX_LD) return -1;
    hipLaunchKernelGGL(k_head_probs, dim3(1), dim3(256), 0, st, h, p, NS, cen_sums);
    if (hook && c.training) hook(cen_sums, p.K * p.Da + p.K, user);
    SampleCfg c2 = c;
    if (sh && sh->fn && sh->action) {        // model.py:171-173: an explicit action sampler replaces Gumbel / soft-max sampling
        sh->fn(h.logp, h.dirs, sh->samples_buf, nullptr, NS, p.K, p.Da, 0, sh->user);
        c2.mode = 2; c2.samples_in = sh->samples_buf;
    }
    hipLaunchKernelGGL(k_head_sample, dim3(1), dim3(256), 0, st, h, p, c2, NS, (const float*)cen_sums);
    if (sh && sh->fn && sh->variation) {     // model.py:189-190: the variation sampler sees the final samples; re-emit aux / variations with its result
        sh->fn(h.logp, h.dirs, h.samples, sh->var_buf, NS, p.K, p.Da, 1, sh->user);
        SampleCfg c3 = c2;
        c3.mode = 2; c3.samples_in = h.samples; c3.variations_in = sh->var_buf; c3.training = 0;     // centroids already updated above
        hipLaunchKernelGGL(k_head_sample, dim3(1), dim3(256), 0, st, h, p, c3, NS, (const float*)cen_sums);
    }
    return 0;
}
int head_backward(const HeadBufs& h, const HeadParams& p, const SampleCfg& c, int B, int T, int first_call, hipStream_t st) {
    hipLaunchKernelGGL(k_head_bwd1, dim3(cdiv((long)B * (T - 1), 64)), dim3(64), 0, st, h, p, c, B * (T - 1), first_call);
    hipLaunchKernelGGL(k_head_bwd2, dim3(cdiv((long)B * T * p.F, 256)), dim3(256), 0, st, h, p, B, T);
    hipLaunchKernelGGL(k_head_bwd3, dim3(cdiv((long)p.Da * p.F, 4)), dim3(256), 0, st, h, p, B * T, B * (T - 1));
    return 0;
}
int loss_l1(const TV& gt, const TV& rec, const TV& drec, int f, int t_off, int Tobs, int Trec, float gscale, double* acc, float* gt_out, hipStream_t st) {
    long npix = (long)rec.N * rec.H * rec.W;
    long blocks = (npix + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_loss_l1, dim3((unsigned)blocks), dim3(256), 0, st, gt, rec, drec, f, t_off, Tobs, Trec, gscale, acc, gt_out);
    return 0;
}
int loss_mse(const TV& a, const TV& b, const TV& db, float gscale, double* acc, hipStream_t st) {
    long items = (long)a.N * a.H * a.W * a.C;
    long blocks = (items + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_loss_mse, dim3((unsigned)blocks), dim3(256), 0, st, a, b, db, gscale, acc);
    return 0;
}
// sum |x| over a view -> acc (atomic, one per block)
__global__ __launch_bounds__(256) void k_abs_sum(TV a, double* acc) {
    __shared__ double sh[8];
    const int HW = a.H * a.W, C = a.C;
    const long items = (long)a.N * HW * C;
    double s = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < items; i += (long)gridDim.x * blockDim.x) {
        long q = i / C; int c = (int)(i - q * C);
        long n = q / HW; long pix = q - n * HW;
        s += (double)fabsf(a.p[n * a.sn + pix * a.ld + c]);
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) atomicAdd(acc, s);
}
// the small action tensors: one block, thread-strided rows, block sums in fp64.  (log p of EntropyProbabilityLoss carries no epsilon, as in the reference: losses.py:359-376)
__global__ __launch_bounds__(256) void k_diag_small(DiagArgs d, double n_states, double n_hidden) {
    __shared__ double sh[8];
    __shared__ double colsum[16];
    const int K = d.K, Da = d.Da, NS = d.NS, tid = threadIdx.x;
    double ent = 0, dm = 0, dv_ = 0, rm = 0, rv = 0, err = 0, kl = 0, vn = 0, vm = 0;
    if (tid < 16) colsum[tid] = 0.0;
    __syncthreads();
    double cs[16];
    for (int k = 0; k < K; k++) cs[k] = 0.0;
    for (int n = tid; n < NS; n += 256) {
        for (int k = 0; k < K; k++) { const float p = d.samples[n * K + k]; ent -= (double)(p * logf(p)); cs[k] += p; }
        double nrm = 0.0, klr = 0.0;
        for (int j = 0; j < Da; j++) {
            const float m = d.ddist[n * 2 * Da + j], v = d.ddist[n * 2 * Da + Da + j], m2 = d.rdist[n * 2 * Da + j], v2 = d.rdist[n * 2 * Da + Da + j];
            dm += fabsf(m); dv_ += fabsf(v); rm += fabsf(m2); rv += fabsf(v2); err += (double)(m2 - m) * (m2 - m);
            klr += 1.0 + (double)logf(v2) - (double)m2 * m2 - (double)v2;
            const float x = d.variations[n * Da + j];
            nrm += (double)x * x; vm += x;
        }
        kl += -0.5 * klr; vn += sqrt(nrm);
    }
    for (int k = 0; k < K; k++) atomicAdd(&colsum[k], cs[k]);
    double* o = d.acc + LOSS_DIAG_0;
    double t;
    t = block_sum(ent, sh); if (tid == 0) o[0] = t / NS; __syncthreads();
    t = block_sum(dm, sh); if (tid == 0) o[4] = t / ((double)NS * Da); __syncthreads();
    t = block_sum(dv_, sh); if (tid == 0) o[5] = t / ((double)NS * Da); __syncthreads();
    t = block_sum(rm, sh); if (tid == 0) o[6] = t / ((double)NS * Da); __syncthreads();
    t = block_sum(rv, sh); if (tid == 0) o[7] = t / ((double)NS * Da); __syncthreads();
    t = block_sum(err, sh); if (tid == 0) o[8] = t / ((double)NS * Da); __syncthreads();
    t = block_sum(kl, sh); if (tid == 0) o[9] = t / NS; __syncthreads();
    t = block_sum(vn, sh); if (tid == 0) o[12] = t / NS; __syncthreads();
    t = block_sum(vm, sh); if (tid == 0) o[13] = t / ((double)NS * Da); __syncthreads();
    if (tid == 0) {
        double e2 = 0.0;
        for (int k = 0; k < K; k++) { const float p = (float)(colsum[k] / NS); e2 -= (double)(p * logf(p)); }      // entropy of the batch-mean assignment (one row)
        o[1] = e2;
        o[2] = o[2] / n_states; o[3] = o[3] / n_hidden;                       // (sums of |x| accumulated by k_abs_sum before this kernel)
        double cm = 0.0, cd = 0.0;
        for (int i = 0; i < K; i++) {
            for (int j = 0; j < Da; j++) cm += fabsf(d.centroids[i * Da + j]);
            for (int i2 = 0; i2 < K; i2++) { double q = 0.0; for (int j = 0; j < Da; j++) { const double df = (double)d.centroids[i * Da + j] - d.centroids[i2 * Da + j]; q += df * df; } cd += sqrt(q); }
        }
        o[10] = cm / ((double)K * Da); o[11] = K > 1 ? cd / ((double)K * (K - 1)) : 0.0;
    }
}
int loss_diagnostics(const DiagArgs& d, const TV& states, const TV& hidden, hipStream_t st) {
    if (d.K > 16 || d.Da > 8) return -1;
    auto blocks = [](const TV& t) { long items = (long)t.N * t.H * t.W * t.C; long b = (items + 255) / 256; return (unsigned)(b > 1024 ? 1024 : (b < 1 ? 1 : b)); };
    hipLaunchKernelGGL(k_abs_sum, dim3(blocks(states)), dim3(256), 0, st, states, d.acc + LOSS_DIAG_0 + 2);
    hipLaunchKernelGGL(k_abs_sum, dim3(blocks(hidden)), dim3(256), 0, st, hidden, d.acc + LOSS_DIAG_0 + 3);
    hipLaunchKernelGGL(k_diag_small, dim3(1), dim3(256), 0, st, d, (double)states.N * states.H * states.W * states.C, (double)hidden.N * hidden.H * hidden.W * hidden.C);
    return 0;
}
int loss_small(const SmallLossArgs& a, allreduce_hook_t hook, void* user, hipStream_t st) {
    if (a.K > 16 || a.Da > 8) return -1;
    hipLaunchKernelGGL(k_joint_matrix, dim3(1), dim3(256), 0, st, a);
    if (hook) hook(a.Pbuf, a.K * a.K, user);
    hipLaunchKernelGGL(k_loss_small, dim3(1), dim3(256), 0, st, a);
    return 0;
}
int loss_finalize(double* acc, const LossWeights& w, double n0, double n1, double n2, double nstates, double nhidden, const VggLevels* lv, hipStream_t st) {
    VggLevels l{}; if (lv) l = *lv;
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(64), 0, st, acc, w, n0, n1, n2, nstates, nhidden, l, lv ? 1 : 0);
    return 0;
}
int loss_report_flag(unsigned* flags, int n, double* slot, double* total, hipStream_t st) {
    hipLaunchKernelGGL(k_report_flag, dim3(1), dim3(64), 0, st, flags, n, slot, total);
    return 0;
}
int loss_diff_per_frame(const TV& a, int Ta, int a_off, const TV& b, int Tb, int C, int sq, double* acc, hipStream_t st) {
    const long items = (long)b.H * b.W * C;
    const unsigned bx = (unsigned)(items / 2048 < 1 ? 1 : (items / 2048 > 64 ? 64 : items / 2048));
    hipLaunchKernelGGL(k_diff_per_frame, dim3(bx, b.N), dim3(256), 0, st, a, Ta, a_off, b, Tb, C, sq, acc);
    return 0;
}
