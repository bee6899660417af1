// Driver of the CADDY hot path: parameter table, layer construction, E / R / D / A graphs, tape-based BPTT.
// Reference behaviour: model/main_model/model.py (forward_full_model :84-286, generate_next :570-607) and the
// reduced variant (model/reduced_model/rendering_network.py:30-42).  All device work goes through the kernels of
// conv_mfma.hip / pointwise.hip / head.hip / pack.hip on one HIP stream; nothing here allocates device memory.
#include "net.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

static thread_local std::string g_err;
static int finish(caddy_ctx* c) {   // surface asynchronous launch errors of this API call
    if (!c->dry && c->act.overflow() && !c->fail) { c->fail = true; g_err = "activation arena overflow (workspace smaller than caddy_workspace_bytes)"; }
    if (!c->dry) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess && !c->fail) { c->fail = true; g_err = std::string("HIP error: ") + hipGetErrorString(e); }
    }
    return c->fail ? -1 : 0;
}
void set_error(const std::string& s) { g_err = s; }
extern "C" const char* caddy_last_error(void) { return g_err.c_str(); }

#define RUN(expr) do { if (!dry) ck((expr), #expr); } while (0)

void caddy_ctx::ck(int rc, const char* what) {
    if (rc != 0 && !fail) { fail = true; set_error(std::string("kernel launch failed: ") + what); }
}

// ---------------------------------------------------------------------------------------------------------------------
// parameter table (names = reference state_dict keys; see oracle/caddy_oracle.py param_table for the same list)
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct TB {
    std::vector<ParamEntry>& t;
    void add(const std::string& n, std::initializer_list<int> shp, int kind) {
        ParamEntry e; e.name = n; e.ndim = (int)shp.size(); e.kind = kind; e.numel = 1; e.offset = -1;
        int i = 0; for (int s : shp) { e.shape[i++] = s; e.numel *= s; }
        for (; i < 4; i++) e.shape[i] = 1;
        t.push_back(e);
    }
    void bn(const std::string& p, int c) { add(p + ".weight", {c}, 0); add(p + ".bias", {c}, 0); add(p + ".running_mean", {c}, 1); add(p + ".running_var", {c}, 1); }
    void res(const std::string& p, int cin, int cout, int ds) {
        add(p + ".conv1.weight", {cout, cin, 3, 3}, 0); bn(p + ".bn1", cout);
        add(p + ".conv2.weight", {cout, cout, 3, 3}, 0); bn(p + ".bn2", cout);
        if (ds != 1 || cin != cout) { add(p + ".downsample.0.weight", {cout, cin, 1, 1}, 0); bn(p + ".downsample.2", cout); }
    }
};
const int E_BLOCKS[6][3] = {{16, 16, 1}, {16, 32, 2}, {32, 32, 1}, {32, 64, 2}, {64, 64, 1}, {64, 65, 1}};  // representation_network.py:22-29
void dec_widths(const caddy_config& c, int w[4]) {
    if (c.variant == 0) { w[0] = 128; w[1] = 128; w[2] = 64; w[3] = 32; } else { w[0] = 64; w[1] = 64; w[2] = 32; w[3] = 16; }
}
}  // namespace

void build_param_table(const caddy_config& c, std::vector<ParamEntry>& t, long* n_floats, long* n_train) {
    t.clear();
    TB b{t};
    const int Cs = 64, aux = c.actions + c.action_dim, Ch = c.hidden, hs = c.height / 8, ws = c.width / 8;
    b.add("state_to_hidden_state_layer.0.weight", {Ch, Cs, 3, 3}, 0); b.add("state_to_hidden_state_layer.0.bias", {Ch}, 0);
    for (int m = 0; m < (c.ensemble > 1 ? c.ensemble : 1); m++) {      // nn.ModuleList of ensamble_size ActionNetworks (model.py:47)
        std::string p = "action_network." + std::to_string(m);
        b.res(p + ".residuals.0", Cs, 2 * Cs, 2); b.res(p + ".residuals.1", 2 * Cs, 2 * Cs, 1);
        b.add(p + ".mean_fc.weight", {c.action_dim, 2 * Cs}, 0); b.add(p + ".mean_fc.bias", {c.action_dim}, 0);
        b.add(p + ".variance_fc.weight", {c.action_dim, 2 * Cs}, 0); b.add(p + ".variance_fc.bias", {c.action_dim}, 0);
        b.add(p + ".final_fc.weight", {c.actions, c.action_dim}, 0); b.add(p + ".final_fc.bias", {c.actions}, 0);
    }
    const int lc[3][2] = {{Cs + aux, Ch}, {2 * Ch + aux, 2 * Ch}, {Ch + aux, Ch}};
    for (int i = 0; i < 3; i++) {
        std::string q = "dynamics_network.recurrent_layers_blocks." + std::to_string(i);
        int hh = i == 1 ? hs / 2 : hs, ww = i == 1 ? ws / 2 : ws, co = lc[i][1], ci = lc[i][0];
        b.add(q + ".0.initial_hidden_state", {co, hh, ww}, 0); b.add(q + ".0.initial_hidden_cell_state", {co, hh, ww}, 0);
        const char* gn[4] = {"input_gate", "forget_gate", "output_gate", "cell_gate"};
        for (int g = 0; g < 4; g++) b.add(q + ".0.cell." + gn[g] + ".weight", {co, ci + co, 3, 3}, 0);
        for (int g = 0; g < 4; g++) b.add(q + ".0.cell." + gn[g] + ".bias", {co}, 0);    // contiguous: one packed bias [i|f|o|g]
        b.bn(q + ".1", co);
    }
    std::string q = "dynamics_network.non_recurrent_blocks";
    b.add(q + ".0.conv1.weight", {2 * Ch, Ch + aux, 3, 3}, 0); b.bn(q + ".0.bn1", 2 * Ch);
    b.add(q + ".1.conv.weight", {Ch, 2 * Ch + aux, 3, 3}, 0); b.bn(q + ".1.norm", Ch);
    b.add(q + ".2.conv1.weight", {Ch, Ch + aux, 3, 3}, 0); b.bn(q + ".2.bn1", Ch);
    q = "representation_network";
    b.add(q + ".conv1.weight", {16, 3 * c.stacking, 3, 3}, 0); b.bn(q + ".bn1", 16);
    for (int i = 0; i < 6; i++) b.res(q + ".residuals." + std::to_string(i), E_BLOCKS[i][0], E_BLOCKS[i][1], E_BLOCKS[i][2]);
    int w[4]; dec_widths(c, w);
    q = "rendering_network";
    b.add(q + ".upsample_blocks.0.0.conv.weight", {w[1], w[0], 3, 3}, 0); b.bn(q + ".upsample_blocks.0.0.norm", w[1]);
    b.res(q + ".upsample_blocks.0.1", w[1], w[1], 1);
    b.add(q + ".upsample_blocks.1.0.conv.weight", {w[2], w[1], 3, 3}, 0); b.bn(q + ".upsample_blocks.1.0.norm", w[2]);
    b.res(q + ".upsample_blocks.1.1", w[2], w[2], 1);
    b.add(q + ".upsample_blocks.2.conv.weight", {w[3], w[2], 3, 3}, 0); b.bn(q + ".upsample_blocks.2.norm", w[3]);
    for (int i = 0; i < 3; i++) {
        int ks = i == 2 ? 7 : 3;
        b.add(q + ".final_blocks." + std::to_string(i) + ".conv.weight", {3, w[i + 1], ks, ks}, 0);
        b.add(q + ".final_blocks." + std::to_string(i) + ".conv.bias", {3}, 0);
    }
    b.add("centroid_estimator.estimated_centroids", {c.actions, c.action_dim}, 2);
    long off = 0;
    for (int kind : {0, 2, 1}) {
        for (auto& e : t) if (e.kind == kind) { e.offset = off; off += (e.numel + 3) / 4 * 4; }
        if (kind == 0) *n_train = off;
    }
    *n_floats = off;
}

// ---------------------------------------------------------------------------------------------------------------------
// construction
// ---------------------------------------------------------------------------------------------------------------------
namespace {
const ParamEntry& find(caddy_ctx* c, const std::string& n) {
    for (auto& e : c->table) if (e.name == n) return e;
    set_error("unknown parameter " + n); c->fail = true;
    return c->table[0];
}
float* PP(caddy_ctx* c, const std::string& n) { return c->P + find(c, n).offset; }
float* GP(caddy_ctx* c, const std::string& n) { return c->G + find(c, n).offset; }

void make_conv(caddy_ctx* c, ConvL& L, const std::vector<std::string>& wn, const std::string& bias, int KS, const std::vector<int>& segC) {
    const ParamEntry& e0 = find(c, wn[0]);
    PackDesc& d = L.pd;
    d.nw = (int)wn.size(); d.Co_each = e0.shape[0]; d.Cin = e0.shape[1]; d.KS = KS; d.nseg = (int)segC.size();
    int off = 0, kt = 0;
    for (int s = 0; s < d.nseg; s++) { d.seg_off[s] = off; d.seg_C[s] = segC[s]; d.seg_Cpad[s] = round_up(segC[s], CONV_BK); off += segC[s]; kt += d.seg_Cpad[s]; }
    if (off != d.Cin) { set_error("segment mismatch for " + wn[0]); c->fail = true; }
    for (int i = 0; i < d.nw; i++) { d.w[i] = PP(c, wn[i]); d.gw[i] = GP(c, wn[i]); }
    d.Cout = d.nw * d.Co_each; d.Cout_pad = round_up(d.Cout, conv_pick_bn(d.Cout)); d.Ktot = kt;
    L.wp_floats = (size_t)KS * KS * d.Cout_pad * d.Ktot;
    L.wp = (float*)c->persist.alloc(L.wp_floats * 4);
    L.dwp = nullptr;      // carved out of ONE contiguous pool at the end of build_layers (one memset per backward instead of one per layer)
    L.kd = round_up(d.Cout, CONV_BK);
    for (int s = 0; s < d.nseg; s++) {
        L.cd_pad[s] = round_up(segC[s], conv_pick_bn(segC[s]));
        L.wpd[s] = (float*)c->persist.alloc((size_t)KS * KS * L.cd_pad[s] * L.kd * 4);
    }
    if (!bias.empty()) { L.bias = PP(c, bias); L.dbias = GP(c, bias); }
    // split 16-bit operand forms (conv_hx.hip): 3x3 layers with >= 16 channels on both sides.  (Round 5: the 16-channel layers too -- a half-filled 32-channel chunk / 16 of 32 tile
    // columns on the 16-bit pipe still beat the exact-fp32 16x16x4 kernels of conv_narrow.hip, which ran at 28 - 40 % of a 16 x slower pipe: E/R/A/D step -0.6 ms, Breakout-160 -0.4 ms,
    // roll-out 2598 -> 2618 frames/s in one call.  Layers with fewer channels on a side -- the stems, the heads -- keep their own kernels.)
    if (KS == 3 && d.Cin >= 16 && d.Cout >= 16) L.wq = c->persist.alloc(hx_weight_bytes(d, -1, round_up(d.Cout, hx_pick_bn(d.Cout)), 2));
    for (int s = 0; s < d.nseg; s++)
        if (KS == 3 && segC[s] >= 16 && d.Cout >= 16) L.wqd[s] = c->persist.alloc(hx_weight_bytes(d, s, round_up(segC[s], hx_pick_bn(segC[s])), 2));
    L.flag_idx = (int)c->convs.size();
    if (L.flag_idx >= CADDY_VGG_FLAG0) { set_error("internal: too many convolution layers for the range-guard flag table"); c->fail = true; L.flag_idx = 0; }
    c->convs.push_back(&L);
}
void make_bn(caddy_ctx* c, BNL& b, const std::string& p) {
    const ParamEntry& e = find(c, p + ".weight");
    b.name = p; b.C = e.shape[0];
    b.gamma = PP(c, p + ".weight"); b.beta = PP(c, p + ".bias"); b.dgamma = GP(c, p + ".weight"); b.dbeta = GP(c, p + ".bias");
    b.rmean = PP(c, p + ".running_mean"); b.rvar = PP(c, p + ".running_var");
    b.eval_stash = (float*)c->persist.alloc(sizeof(float) * 4 * (size_t)round_up(b.C, 4));
    c->bns.push_back(&b);
}
void make_res(caddy_ctx* c, ResL& R, const std::string& p, int cin, int cout, int ds) {
    R.ds = ds;
    make_conv(c, R.conv1, {p + ".conv1.weight"}, "", 3, {cin}); make_bn(c, R.bn1, p + ".bn1");
    make_conv(c, R.conv2, {p + ".conv2.weight"}, "", 3, {cout}); make_bn(c, R.bn2, p + ".bn2");
    R.has_down = ds != 1 || cin != cout;
    if (R.has_down) { make_conv(c, R.down, {p + ".downsample.0.weight"}, "", 1, {cin}); make_bn(c, R.bnd, p + ".downsample.2"); }
}
T4 palloc(caddy_ctx* c, int N, int H, int W, int C) {   // persistent tensor with its own gradient buffer
    int ld = round_up(C, 4);
    size_t n = (size_t)N * H * W * ld * 4;
    T4 t; t.d = (float*)c->persist.alloc(n); t.g = (float*)c->persist.alloc(n);
    t.N = N; t.H = H; t.W = W; t.C = C; t.sn = (long)H * W * ld; t.ld = ld;
    return t;
}
void build_layers(caddy_ctx* c) {
    const caddy_config& g = c->cfg;
    const int Cs = 64, aux = g.actions + g.action_dim, Ch = g.hidden;
    c->hs = g.height / 8; c->ws = g.width / 8;
    make_conv(c, c->s2h, {"state_to_hidden_state_layer.0.weight"}, "state_to_hidden_state_layer.0.bias", 3, {Cs});
    c->n_members = g.ensemble > 1 ? g.ensemble : 1;
    for (int m = 0; m < c->n_members; m++) {
        std::string p = "action_network." + std::to_string(m);
        make_res(c, c->a_res[m][0], p + ".residuals.0", Cs, 2 * Cs, 2); make_res(c, c->a_res[m][1], p + ".residuals.1", 2 * Cs, 2 * Cs, 1);
        HeadParams& h = c->hp[m];
        h.F = 2 * Cs; h.Da = g.action_dim; h.K = g.actions;
        h.Wm = PP(c, p + ".mean_fc.weight"); h.bm = PP(c, p + ".mean_fc.bias"); h.Wv = PP(c, p + ".variance_fc.weight"); h.bv = PP(c, p + ".variance_fc.bias");
        h.Wf = PP(c, p + ".final_fc.weight"); h.bf = PP(c, p + ".final_fc.bias");
        h.dWm = GP(c, p + ".mean_fc.weight"); h.dbm = GP(c, p + ".mean_fc.bias"); h.dWv = GP(c, p + ".variance_fc.weight"); h.dbv = GP(c, p + ".variance_fc.bias");
        h.dWf = GP(c, p + ".final_fc.weight"); h.dbf = GP(c, p + ".final_fc.bias");
        long lo = -1, hi = -1;      // trainable range of this member (kind-0 entries are laid out in table order: one contiguous range per member)
        for (auto& e : c->table) if (e.kind == 0 && e.name.rfind(p + ".", 0) == 0) { if (lo < 0 || e.offset < lo) lo = e.offset; long en = e.offset + (e.numel + 3) / 4 * 4; if (en > hi) hi = en; }
        c->member_lo[m] = lo < 0 ? 0 : lo; c->member_hi[m] = lo < 0 ? 0 : hi;
    }
    const int lc[3][2] = {{Cs, Ch}, {2 * Ch, 2 * Ch}, {Ch, Ch}};
    for (int i = 0; i < 3; i++) {
        std::string q = "dynamics_network.recurrent_layers_blocks." + std::to_string(i);
        LstmL& L = c->lstm[i];
        L.C = lc[i][1]; L.Hs = i == 1 ? c->hs / 2 : c->hs; L.Ws = i == 1 ? c->ws / 2 : c->ws;
        make_conv(c, L.gates, {q + ".0.cell.input_gate.weight", q + ".0.cell.forget_gate.weight", q + ".0.cell.output_gate.weight", q + ".0.cell.cell_gate.weight"},
                  q + ".0.cell.input_gate.bias", 3, {lc[i][0], aux, L.C});
        make_bn(c, L.bn, q + ".1");
        L.init_h = PP(c, q + ".0.initial_hidden_state"); L.init_c = PP(c, q + ".0.initial_hidden_cell_state");
        L.ginit_h = GP(c, q + ".0.initial_hidden_state"); L.ginit_c = GP(c, q + ".0.initial_hidden_cell_state");
        L.ih = palloc(c, 1, L.Hs, L.Ws, L.C); L.ic = palloc(c, 1, L.Hs, L.Ws, L.C);
        L.ph = palloc(c, g.batch, L.Hs, L.Ws, L.C); L.pc = palloc(c, g.batch, L.Hs, L.Ws, L.C);
        L.h.d = nullptr; L.c.d = nullptr;
    }
    std::string q = "dynamics_network.non_recurrent_blocks";
    make_conv(c, c->r_c0, {q + ".0.conv1.weight"}, "", 3, {Ch, aux}); make_bn(c, c->r_bn0, q + ".0.bn1");
    make_conv(c, c->r_c1, {q + ".1.conv.weight"}, "", 3, {2 * Ch, aux}); make_bn(c, c->r_bn1, q + ".1.norm");
    make_conv(c, c->r_c2, {q + ".2.conv1.weight"}, "", 3, {Ch, aux}); make_bn(c, c->r_bn2, q + ".2.bn1");
    q = "representation_network";
    make_conv(c, c->e_stem, {q + ".conv1.weight"}, "", 3, {3 * g.stacking}); make_bn(c, c->e_bn1, q + ".bn1");
    for (int i = 0; i < 6; i++) make_res(c, c->e_res[i], q + ".residuals." + std::to_string(i), E_BLOCKS[i][0], E_BLOCKS[i][1], E_BLOCKS[i][2]);
    int w[4]; dec_widths(g, w);
    q = "rendering_network";
    for (int i = 0; i < 3; i++) {
        std::string u = q + ".upsample_blocks." + std::to_string(i) + (i < 2 ? ".0" : "");
        make_conv(c, c->d_up[i], {u + ".conv.weight"}, "", 3, {w[i]}); make_bn(c, c->d_norm[i], u + ".norm");
        if (i < 2) make_res(c, c->d_res[i], q + ".upsample_blocks." + std::to_string(i) + ".1", w[i + 1], w[i + 1], 1);
        std::string f = q + ".final_blocks." + std::to_string(i) + ".conv";
        make_conv(c, c->d_final[i], {f + ".weight"}, f + ".bias", i == 2 ? 7 : 3, {w[i + 1]});
    }
    {   // conv -> (avg-pool) -> BatchNorm pairs of the roll-out path (E, R's non-recurrent blocks, D): foldable in eval mode
        auto pair = [&](ConvL& L, BNL& b) { L.fold_bn = &b; L.fold_bias = (float*)c->persist.alloc(sizeof(float) * (size_t)round_up(L.pd.Cout, 4)); };
        auto pair_res = [&](ResL& R) { pair(R.conv1, R.bn1); pair(R.conv2, R.bn2); if (R.has_down) pair(R.down, R.bnd); };
        pair(c->e_stem, c->e_bn1);
        for (int i = 0; i < 6; i++) pair_res(c->e_res[i]);
        pair(c->r_c0, c->r_bn0); pair(c->r_c1, c->r_bn1); pair(c->r_c2, c->r_bn2);
        for (int i = 0; i < 3; i++) pair(c->d_up[i], c->d_norm[i]);
        for (int i = 0; i < 2; i++) pair_res(c->d_res[i]);
    }
    c->centroids = PP(c, "centroid_estimator.estimated_centroids");
    {   // gradient buckets: trainable parameters of dynamics_network (R) and rendering_network (D) are contiguous ranges of the flat buffer
        const char* pre[2] = {"dynamics_network.", "rendering_network."};
        for (int b = 0; b < 2; b++) {
            long lo = -1, hi = -1;
            for (auto& e : c->table) if (e.kind == 0 && e.name.rfind(pre[b], 0) == 0) { if (lo < 0 || e.offset < lo) lo = e.offset; long en = e.offset + (e.numel + 3) / 4 * 4; if (en > hi) hi = en; }
            c->bucket_lo[b] = lo < 0 ? 0 : lo; c->bucket_hi[b] = lo < 0 ? 0 : hi;
        }
        for (auto& e : c->table) if (e.kind == 0 && e.name.rfind("state_to_hidden_state_layer.", 0) == 0) {
            long en = e.offset + (e.numel + 3) / 4 * 4;
            if (c->s2h_hi == 0 || e.offset < c->s2h_lo) c->s2h_lo = e.offset;
            if (en > c->s2h_hi) c->s2h_hi = en;
        }
        for (int i = 0; i < 3; i++) c->lstm[i].gates.early_bucket = true;
        c->r_c0.early_bucket = c->r_c1.early_bucket = c->r_c2.early_bucket = true;
        for (int i = 0; i < 3; i++) { c->d_up[i].early_bucket = true; c->d_final[i].early_bucket = true; }
        for (int i = 0; i < 2; i++) { c->d_res[i].conv1.early_bucket = c->d_res[i].conv2.early_bucket = true; if (c->d_res[i].has_down) c->d_res[i].down.early_bucket = true; }
    }
    c->inf_aux = (float*)c->persist.alloc(AUX_LD * 4);      // roll-out: one-hot action + variation row read by the captured per-frame kernel sequence
    if (g.perceptual) vgg_build(c);
    for (int i = 0; i < 3; i++) c->d_norm[i].deferred = true;      // D's BatchNorms: their calls execute on two streams (see BNL)
    for (int i = 0; i < 2; i++) { c->d_res[i].bn1.deferred = c->d_res[i].bn2.deferred = true; if (c->d_res[i].has_down) c->d_res[i].bnd.deferred = true; }
    {   // zero pool: packed weight gradients of every layer + the (1,h,w,C) gradients of the learned ConvLSTM initial states + the loss accumulators,
        // contiguous so that loss_backward clears them with a single memset
        size_t bytes = 0;
        auto take = [&](size_t n) { size_t o = bytes; bytes += (n + 255) & ~(size_t)255; return o; };
        std::vector<size_t> offs;
        for (ConvL* L : c->convs) offs.push_back(take(L->wp_floats * 4));
        size_t lo[3][2];
        for (int i = 0; i < 3; i++) { lo[i][0] = take(c->lstm[i].ih.sn * 4); lo[i][1] = take(c->lstm[i].ic.sn * 4); }
        std::vector<std::pair<BNL*, size_t>> bnd;      // private parameter-gradient accumulators of the decoder stream: [dgamma | dbeta]
        for (BNL* b : c->bns) if (b->deferred) bnd.emplace_back(b, take(sizeof(float) * 2 * (size_t)round_up(b->C, 4)));
        const size_t acc_off = take(sizeof(double) * LOSS_SLOTS);
        char* pool = (char*)c->persist.alloc(bytes);
        c->zero_pool = pool; c->zero_pool_bytes = bytes;
        for (size_t i = 0; i < c->convs.size(); i++) c->convs[i]->dwp = (float*)(pool + offs[i]);
        for (int i = 0; i < 3; i++) { c->lstm[i].ih.g = (float*)(pool + lo[i][0]); c->lstm[i].ic.g = (float*)(pool + lo[i][1]); }
        c->loss_acc = (double*)(pool + acc_off);
        for (auto& kv : bnd) { kv.first->dgamma_d = (float*)(pool + kv.second); kv.first->dbeta_d = kv.first->dgamma_d + round_up(kv.first->C, 4); }
    }
    c->red_scratch = (double*)c->persist.alloc(sizeof(double) * RED_MAX_BLOCKS * 2 * 1024);
    c->sat_flag = (unsigned*)c->persist.alloc(sizeof(unsigned) * 2 * CADDY_N_FLAGS);      // [raised by the forward kernels since the last loss call | sticky until polled]
    c->wgrad_det_cap = 16L << 20;      // 64 MB: >= 3 copies of the largest packed weight gradient (ConvLSTM 1 gates, 9 x 1024 x 528), 256 of a 64 x 64 layer
    c->wgrad_det = (float*)c->persist.alloc(sizeof(float) * c->wgrad_det_cap);
    c->conv_aux = (float*)c->persist.alloc(CONV_AUX_BYTES);
    c->conv_aux2 = (float*)c->persist.alloc(CONV_AUX_BYTES);
    c->conv_split_cap = 9L * 4096 * 256;                      // 9 slabs x (<= 4096 pixels x 256 channels): only under-filled launches use it
    c->conv_split = (float*)c->persist.alloc(sizeof(float) * c->conv_split_cap);
    // device tables of the one-launch (un)packing jobs
    c->pack_jobs.cap = 8 * (int)c->convs.size() + 8; c->pack_jobs.dev = (PackJob*)c->persist.alloc(sizeof(PackJob) * c->pack_jobs.cap);
    for (int i = 0; i < 3; i++) { c->unpack_jobs[i].cap = (int)c->convs.size() + 8; c->unpack_jobs[i].dev = (PackJob*)c->persist.alloc(sizeof(PackJob) * c->unpack_jobs[i].cap); }
    // private scratch of the teacher-forced decoder stream (caddy_ctx::dstream)
    c->dsr.aux = (float*)c->persist.alloc(CONV_AUX_BYTES);
    c->dsr.split = (float*)c->persist.alloc(sizeof(float) * c->conv_split_cap);
    c->dsr.red = (double*)c->persist.alloc(sizeof(double) * RED_MAX_BLOCKS * 2 * 1024);
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// ops
// ---------------------------------------------------------------------------------------------------------------------
T4 caddy_ctx::alloc(int N, int H, int W, int C, int ld) {
    if (!ld) ld = round_up(C, 4);
    float* d = (float*)act.alloc((size_t)N * H * W * ld * 4);
    T4 t{d, (float*)((char*)d + grad_delta), N, H, W, C, (long)H * W * ld, ld};
    dbg.push_back(t);
    return t;
}
T4 caddy_ctx::alloc_nz(int N, int H, int W, int C, bool second_writer_is_conv) {
    int ld = round_up(C, 4);
    float* d = (float*)act.alloc_top((size_t)N * H * W * ld * 4);
    T4 t{d, (float*)((char*)d + grad_delta), N, H, W, C, (long)H * W * ld, ld, true};
    t.nz2 = second_writer_is_conv;
    dbg.push_back(t);
    return t;
}
float* caddy_ctx::falloc(size_t n) { return (float*)act.alloc(n * 4); }
double* caddy_ctx::dalloc(size_t n) { return (double*)act.alloc(n * 8); }
static inline float* tw(caddy_ctx* c, float* p) { return (float*)((char*)p + c->grad_delta); }
static inline T4 tslice(const T4& bt, int B, int T, int t) { T4 r = bt; r.d = bt.d + (long)t * bt.sn; r.g = bt.g + (long)t * bt.sn; r.N = B; r.sn = (long)T * bt.sn; return r; }
static inline T4 chan(const T4& t, int c0, int C) { T4 r = t; r.d += c0; r.g += c0; r.C = C; return r; }

static void fill_srcs(ConvSrc* dst, const Seg* segs, int nseg) {
    for (int s = 0; s < nseg; s++) {
        const T4& t = segs[s].t;
        dst[s] = ConvSrc{t.d, t.sn, t.ld, t.C, round_up(t.C, CONV_BK), segs[s].bcast, t.bn_scale, t.bn_shift, t.bn_act, 0, 0};
    }
}

int caddy_ctx::timed_conv_fwd(const ConvArgs& a, double flops) {
    if (!prof) return conv_fwd_launch(a, stream);
    int cin = 0; for (int s = 0; s < a.nsrc; s++) cin += a.src[s].bcast ? 0 : a.src[s].C;
    const double px = (double)a.N * a.H * a.W;   // algorithmic bytes: 4 * (|in| + |out| + |W|)
    ProfRec r{ev(), ev(), 0, flops, a.N * a.H * a.W, a.Ktot, a.Cout, a.KS, prof_kind_override >= 0 ? prof_kind_override : (a.accumulate ? 1 : 0),
              4.0 * (px * cin + px * a.Cout + (double)a.KS * a.KS * a.Ktot * a.Cout)};
    hipEventRecord(r.a, stream);
    int rc = conv_fwd_launch(a, stream);
    hipEventRecord(r.b, stream);
    r.fam = g_last_conv_kernel;
    prof_recs.push_back(r);
    return rc;
}
// CADDY_STREAMS=0: every kernel of the step on the caller's stream, in program order (serialised kernel breakdowns: tools/gpu_final_b.sh; also the reference point of the
// determinism tests).  One switch for the whole topology -- weight-gradient stream, decoder stream, auxiliary gradients, VGG19 levels: a partial combination (e.g. the decoder
// stream without the side stream) would put two accumulating writers of one buffer on different streams.
bool caddy_serial_streams() { static const bool serial = getenv("CADDY_STREAMS") && atoi(getenv("CADDY_STREAMS")) == 0; return serial; }
void caddy_ctx::ensure_side() {
    if (!use_side || side) return;
    if (caddy_serial_streams()) { use_side = false; return; }
    // lowest priority: weight-gradient workgroups fill the compute units the BPTT chain leaves idle (R's small feature maps,
    // point-wise kernels) instead of competing with it
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStreamCreateWithPriority(&side, hipStreamNonBlocking, least);
    if (!gt_done) hipEventCreateWithFlags(&gt_done, hipEventDisableTiming);
}
hipStream_t caddy_ctx::wgrad_stream() {   // order the side stream after everything already enqueued on the compute stream(s)
    if (!use_side || !side) return stream;
    hipEvent_t e = sev();
    hipEventRecord(e, stream);
    hipStreamWaitEvent(side, e, 0);
    if (d_forked) {      // the teacher-forced decoder branch is in flight on the other compute stream: a time-batched weight gradient may read what it produced
        hipStream_t other = in_d ? dsr_saved.st : dsr.st;
        hipEvent_t e2 = sev();
        hipEventRecord(e2, other);
        hipStreamWaitEvent(side, e2, 0);
    }
    return side;
}
// (The auxiliary work shares the decoder stream instead of owning one: the ROCm runtime multiplexes HIP streams onto 4 hardware queues by default, and with
//  a fifth stream in the process -- e.g. the input prefetcher of the training loop -- two of them land on one queue, where a packet waiting for an event of the main
//  stream stalls the packets of the other stream behind it: measured +44 ms per step.  The decoder stream is idle once the teacher-forced decoder backward, which is
//  enqueued first, has run.)
void caddy_ctx::ensure_dstream() {
    if (dry || dstream || d_done) return;      // once (the simulator build hands out null streams: everything then runs in order on the caller's stream)
    if (hipStreamCreateWithFlags(&dstream, hipStreamNonBlocking) != hipSuccess) dstream = nullptr;
    hipEventCreateWithFlags(&d_done, hipEventDisableTiming);
}
static bool aux_stream_off() { return caddy_serial_streams(); }
bool caddy_ctx::aux_enabled() { if (dry || aux_stream_off()) return false; ensure_dstream(); return dstream != nullptr; }
hipStream_t caddy_ctx::aux_grad_stream() {
    if (dry || aux_stream_off()) return stream;
    ensure_dstream();
    if (!dstream || in_d) return stream;       // (already on the decoder stream: its own ops are off the chain anyway)
    hipEvent_t e = sev();
    hipEventRecord(e, stream);
    hipStreamWaitEvent(dstream, e, 0);
    a_dirty = true;
    return dstream;
}
void caddy_ctx::defer_aux(std::function<void()> job) {
    if (dry || in_d) { job(); return; }      // (in_d: already off the chain, on the decoder stream)
    aux_jobs.push_back(std::move(job));
    if (fork_batch <= 1 || (int)aux_jobs.size() >= fork_batch) flush_aux();
}
void caddy_ctx::flush_aux() {
    if (aux_jobs.empty()) return;
    std::vector<std::function<void()>> jobs;
    jobs.swap(aux_jobs);
    hipStream_t as = aux_grad_stream();      // ONE fork
    if (as == stream) { for (auto& j : jobs) j(); return; }
    StreamRes keep{stream, conv_aux, conv_split, red_scratch};
    stream = as; conv_aux = dsr.aux; conv_split = dsr.split; red_scratch = dsr.red;
    for (auto& j : jobs) j();
    stream = keep.st; conv_aux = keep.aux; conv_split = keep.split; red_scratch = keep.red;
}
void caddy_ctx::step_boundary() { flush_aux(); launch_wgrad_jobs(); }
void caddy_ctx::join_aux(hipStream_t onto) {
    flush_aux();
    if (!dstream || !a_dirty || dry || onto == dstream) return;
    hipEvent_t e = sev();
    hipEventRecord(e, dstream);
    hipStreamWaitEvent(onto, e, 0);
    if (onto == stream && !in_d) a_dirty = false;
}
// switch the driver to the teacher-forced decoder stream (and its private scratch); fork: order it after everything enqueued on the main stream so far
void caddy_ctx::enter_d(bool fork) {
    ensure_dstream();
    tp = &tape2;
    if (dry || !dstream) return;
    if (fork) { hipEvent_t e = sev(); hipEventRecord(e, stream); hipStreamWaitEvent(dstream, e, 0); d_forked = true; }
    dsr_saved = StreamRes{stream, conv_aux, conv_split, red_scratch};
    dsr.st = dstream;
    stream = dsr.st; conv_aux = dsr.aux; conv_split = dsr.split; red_scratch = dsr.red;
    in_d = true;
}
void caddy_ctx::leave_d() {
    tp = &tape;
    if (!in_d) return;
    stream = dsr_saved.st; conv_aux = dsr_saved.aux; conv_split = dsr_saved.split; red_scratch = dsr_saved.red;
    in_d = false;
}
// backward of the teacher-forced decoder calls: concurrently on dstream (forked after the loss kernels), or inline at its place in the main tape
void caddy_ctx::replay_tape2(bool concurrent) {
    if (tape2.empty()) return;
    if (concurrent) enter_d(true);
    for (size_t i = tape2.size(); i-- > 0;) tape2[i]();
    if (concurrent) {
        flush_all_wgrad();      // while the decoder stream is current: the side stream is ordered behind what these chunks read
        if (in_d) hipEventRecord(d_done, stream);
        leave_d();
    }
}
// chunk table of the time-chunked perceptual pass: nch chunks over the Trec reconstructed frames, LAST time steps first (perc_t0[0] = Trec > perc_t0[1] > ... > perc_t0[nch] = 0);
// the late chunks are the larger ones (what is left behind the last chunk -- its steps' BPTT and the E / A tail -- has nothing to run beside)
void caddy_ctx::perc_plan(int Trec, bool chunked) {
    // chunking pays where the VGG19 launches of a chunk still fill the chip (profiles/r06_experiments.md section 9): three chunks for steps of >= 4 M reconstructed pixels, four
    // from 1 M, one pass below.  caddy_debug_set_perc_chunks / a negative CADDY_PERC_CHUNKS override the size test.
    // (measured with the first chunk level-parallel: BAIR 7.9 M pixels one pass / 2 / 3 / 4 chunks 124.3 / 120.7 / 120.2 / 121.7 ms; Breakout-160 2.1 M pixels 39.4 / 37.2 / 40.0 ms;
    //  Breakout-64 0.1 M pixels 12.3 / 14.9 / 18.5 ms)
    const long px = (long)cfg.batch * Trec * cfg.height * cfg.width;
    // (only the full-resolution level chunked, the final form: BAIR 2 / 3 / 4 / 5 / 6 chunks 121.6 / 120.5 / 119.7 / 120.7 / 121.0 ms; Breakout-160 39.2 / 38.3 / 37.2 / 36.6 ms for 1 / 2 / 3 / 4)
    // (the smaller levels cut once, the final form: BAIR one pass / 2 / 3 / 4 chunks 127.0 / 123.2 / 121.4 / 123.0 ms; Breakout-160 39.9 / 37.0 / 37.6 / 36.9 ms)
    const int by_size = px >= (4L << 20) ? perc_chunks_cfg : (px >= (1L << 20) ? (perc_chunks_cfg > 1 ? 4 : perc_chunks_cfg) : 1);
    int n = !chunked ? 1 : (perc_chunks_force > 0 ? perc_chunks_force : by_size);
    if (n > PERC_MAX_CHUNKS) n = PERC_MAX_CHUNKS;
    if (n > Trec) n = Trec;
    if (n < 1) n = 1;
    perc_nch = n; perc_trec = Trec;
    int t = Trec;
    for (int k = 0; k < n; k++) { perc_t0[k] = t; const int left = n - k; t -= (t + left - 1) / left; }
    perc_t0[n] = 0;
    if (chunked && n > 1) if (const char* e = getenv("CADDY_PERC_BOUNDS")) {      // A/B aid: explicit inner boundaries, descending ("10,5" = [10, Trec) [5, 10) [0, 5))
        int m = 0, b[PERC_MAX_CHUNKS];
        for (const char* q = e; *q && m < PERC_MAX_CHUNKS - 1;) { b[m++] = atoi(q); while (*q && *q != ',') q++; if (*q == ',') q++; }
        bool ok = m >= 1; int prev = Trec;
        for (int i = 0; i < m && ok; i++) { ok = b[i] > 0 && b[i] < prev; prev = b[i]; }
        if (ok) { perc_nch = m + 1; perc_t0[0] = Trec; for (int i = 0; i < m; i++) perc_t0[i + 1] = b[i]; perc_t0[m + 1] = 0; }
    }
}
// tape replay, in front of the backward of time step t: the seeds d(rec_r) of that step's frames must be final -- wait for the event of the chunk that holds them (and, with
// stacked observations, the frames the step's feedback encoder adds to: up to stacking - 1 steps earlier)
void caddy_ctx::perc_wait(int t) {
    if (!perc_pipelined || dry) return;
    t -= cfg.stacking - 1;
    if (t < 0) t = 0;
    int k = 0;
    while (k + 1 < perc_nch && t < perc_t0[k + 1]) k++;
    for (int j = 0; j <= k; j++) if (!perc_waited[j]) { hipStreamWaitEvent(stream, perc_ev[j], 0); perc_waited[j] = true; }
}
void caddy_ctx::fold_d_bn_grads() {
    VecAddJobs j{};
    for (BNL* b : bns) {
        if (!b->dgamma_d) continue;
        if (j.count + 2 > VEC_ADD_MAX) { RUN(pw_vec_add(j, stream)); j.count = 0; }
        j.dst[j.count] = b->dgamma; j.src[j.count] = b->dgamma_d; j.n[j.count++] = b->C;
        j.dst[j.count] = b->dbeta; j.src[j.count] = b->dbeta_d; j.n[j.count++] = b->C;
    }
    if (j.count) RUN(pw_vec_add(j, stream));
}
void caddy_ctx::flush_wgrad(PendingW& p) {
    if (p.count == 0) return;
    WgradArgs w = p.first;
    if (p.count > 1) {
        w.group_n = p.first.N; w.N = p.first.N * p.count;
        for (int s = 0; s < w.nsrc; s++) { w.src_gs[s] = p.src_gs[s]; w.src_bn_gs[s] = p.bn_gs[s]; }
        w.dy_gs = p.dy_gs;
    }
    double fl = p.flops;
    p.count = 0; p.flops = 0;
    RUN(timed_conv_wgrad(w, fl));
}
void caddy_ctx::flush_all_wgrad() { for (auto& kv : pending) flush_wgrad(kv.second); launch_wgrad_jobs(); }
void caddy_ctx::queue_wgrad(ConvL* L, const WgradArgs& w, double flops) {
    const int chunk = 5;      // time steps per launch
    // only the per-time-step calls (N == batch) repeat; the B*T-frame passes of E / A run once or twice: launch those immediately so that
    // they overlap with the rest of the backward instead of piling up behind flush_all_wgrad()
    if (chunk <= 1 || dry || w.N != cfg.batch) { RUN(timed_conv_wgrad(w, flops)); return; }
    PendingW* p = nullptr;
    for (auto& kv : pending) if (kv.first == L) { p = &kv.second; break; }
    if (!p) { pending.emplace_back(L, PendingW{}); p = &pending.back().second; }
    if (p->count > 0) {
        const WgradArgs& f = p->first;
        bool ok = f.N == w.N && f.H == w.H && f.W == w.W && f.nsrc == w.nsrc && f.dy_sn == w.dy_sn && f.dy_ld == w.dy_ld && f.dwp == w.dwp && f.dy_s16 == w.dy_s16;
        long ds[CONV_MAX_SRC] = {0, 0, 0}, db[CONV_MAX_SRC] = {0, 0, 0};
        for (int s = 0; ok && s < w.nsrc; s++) {
            ok = f.src[s].sn == w.src[s].sn && f.src[s].ld == w.src[s].ld && f.src[s].C == w.src[s].C && f.src[s].bcast == w.src[s].bcast;
            ds[s] = w.src[s].p - p->last_src[s];
            // lazily normalised source: every time step has its own (scale, shift) table -- same form, constant distance
            ok = ok && (f.src[s].bn_scale != nullptr) == (w.src[s].bn_scale != nullptr) && f.src[s].bn_act == w.src[s].bn_act;
            if (ok && w.src[s].bn_scale) {
                db[s] = w.src[s].bn_scale - p->last_bn[s];
                ok = (w.src[s].bn_shift - w.src[s].bn_scale) == (f.src[s].bn_shift - f.src[s].bn_scale) && (db[s] & 3) == 0;
            }
        }
        long dd = w.dy - p->last_dy;
        if (ok && p->count > 1) {                 // the stride between consecutive time steps must stay the same
            for (int s = 0; s < w.nsrc; s++) ok = ok && ds[s] == p->src_gs[s] && db[s] == p->bn_gs[s];
            ok = ok && dd == p->dy_gs;
        }
        if (!ok) flush_wgrad(*p);
        else if (p->count == 1) { for (int s = 0; s < w.nsrc; s++) { p->src_gs[s] = ds[s]; p->bn_gs[s] = db[s]; } p->dy_gs = dd; }
    }
    if (p->count == 0) p->first = w;
    p->count++; p->flops += flops;
    for (int s = 0; s < w.nsrc; s++) { p->last_src[s] = w.src[s].p; p->last_bn[s] = w.src[s].bn_scale; }
    p->last_dy = w.dy;
    if (p->count >= chunk) flush_wgrad(*p);
}
// Weight-gradient launches and auxiliary-gradient jobs are handed to the other streams in batches of fork_batch (and at every time-step boundary of the backward): a fork --
// hipEventRecord on the compute stream + hipStreamWaitEvent on the other one -- costs the RECORDING stream 5.3 us (tools/probes/event_cost.hip) and the BPTT chain forked
// ~460 times per step.  Measured, E/R/A/D step: 1 / 4 / 8 / 16 per fork -> 73.6 / 73.1 / 73.1-73.3 / 73.5 ms (larger batches start the deferred work too late).
int caddy_ctx::timed_conv_wgrad(const WgradArgs& a, double flops) {
    if (dry || fork_batch <= 1 || !use_side || !side) return launch_conv_wgrad(a, flops, wgrad_stream());
    wgrad_jobs.emplace_back(a, flops);
    if ((int)wgrad_jobs.size() >= fork_batch) launch_wgrad_jobs();
    return 0;
}
void caddy_ctx::launch_wgrad_jobs() {
    if (wgrad_jobs.empty()) return;
    hipStream_t s2 = wgrad_stream();      // ONE fork for all of them
    for (auto& j : wgrad_jobs) RUN(launch_conv_wgrad(j.first, j.second, s2));
    wgrad_jobs.clear();
}
int caddy_ctx::launch_conv_wgrad(const WgradArgs& a0, double flops, hipStream_t stream) {
    WgradArgs a = a0;
    if (deterministic) {
        // ONE scratch, zero-filled, filled and folded per launch: correct only while every weight-gradient launch of a backward pass goes to one stream (the side stream, or the caller's
        // under CADDY_STREAMS=0).  A second stream would silently corrupt gradients in exactly the mode that promises bit-reproducibility: refuse it loudly.
        // (the bit-reproducible mode is the default now, so this must not fail a backward -- e.g. when no side stream could be created while the decoder stream is
        //  active, or the caller moves the context to another stream between queued jobs): the scratch changes hands behind an event -- the new owner starts when everything the
        //  previous owner has in flight on it is done.
        if (wgrad_det_owner_set && wgrad_det_owner != stream) { hipEvent_t e = sev(); hipEventRecord(e, wgrad_det_owner); hipStreamWaitEvent(stream, e, 0); n_wgrad_det_handover++; }
        wgrad_det_owner = stream; wgrad_det_owner_set = true;
        a.det_slab = wgrad_det; a.det_cap = wgrad_det_cap;
    }
    if (!prof) return conv_wgrad_launch(a, stream);
    int cin = 0; for (int s = 0; s < a.nsrc; s++) cin += a.src[s].bcast ? 0 : a.src[s].C;
    const double px = (double)a.N * a.H * a.W;
    ProfRec r{ev(), ev(), 0, flops, a.N * a.H * a.W, a.Ktot, a.Cout, a.KS, 2, 4.0 * (px * cin + px * a.Cout + (double)a.KS * a.KS * a.Ktot * a.Cout)};
    hipEventRecord(r.a, stream);
    int rc = conv_wgrad_launch(a, stream);
    hipEventRecord(r.b, stream);
    r.fam = g_last_conv_kernel;
    prof_recs.push_back(r);
    return rc;
}

T4 caddy_ctx::conv(ConvL& L, const Seg* segs, int nseg, int actf, const T4* into, bool nz_out, const T4* res) {
    int N = 0, H = 0, W = 0;
    for (int s = 0; s < nseg; s++) if (!segs[s].bcast) { N = segs[s].t.N; H = segs[s].t.H; W = segs[s].t.W; break; }
    ConvArgs a{};
    fill_srcs(a.src, segs, nseg);
    a.nsrc = nseg; a.N = N; a.H = H; a.W = W; a.KS = L.pd.KS; a.wp = L.wp; a.Ktot = L.pd.Ktot; a.Cout = L.pd.Cout; a.Cout_pad = L.pd.Cout_pad;
    a.bias = (fold && L.fold_bn) ? L.fold_bias : L.bias; a.act = actf;
    const bool range_ok = !layer_fallback[L.flag_idx];      // (a layer that reported |x| > 65504 stays on the exact-fp32 forward: caddy_f16_saturated)
    if (L.wq && prec_fwd != PREC_FP32 && range_ok) { a.wq = L.wq; a.precision = PREC_F16X3; a.sat_flag = sat_flag + L.flag_idx; }      // (latency kernel: inference only -- training launches are 8 x larger or carry statistics epilogues, and their parity bounds were calibrated on the tile kernel's summation order)
    else if (L.pd.Cout <= 3 && L.pd.KS >= 3 && prec_fwd != PREC_FP32 && range_ok) { a.precision = PREC_F16X3; a.sat_flag = sat_flag + L.flag_idx; }      // FinalBlock heads: split f16 on conv_head.hip (weights split in the kernel)
    a.aux = conv_aux; a.split_scratch = conv_split; a.split_cap = conv_split_cap;
    a.direct_ok = training ? 0 : 1;      // latency kernels (conv_direct.hip): inference passes only -- a training launch is 8 x larger and its summation order is what the parity bounds were calibrated on
    bool pooled = false;
    if (pool_fuse) {      // conv_pool(): the 2x2 average (+ LeakyReLU) goes into the epilogue when the launch has one for it
        a.act = pool_fuse == 2 ? 3 : 0;
        if (!recording && !into && !res && conv_avgpool_ok(a)) { pooled = true; a.avgpool = 1; pool_fuse = 0; }
        else a.act = actf;
    }
    T4 out = into ? *into : (pooled ? alloc(N, H / 2, W / 2, L.pd.Cout) : (nz_out ? alloc_nz(N, H, W, L.pd.Cout) : alloc(N, H, W, L.pd.Cout)));
    a.out = out.d;
    if (recording && s16_grads && !into && !pooled && !res && out.nz && !out.nz2 && actf != 1) {
        // may d(out) travel pre-split (GradFmt)?  One assigning writer; the weight gradient on k_wgrad_hx and every dgrad on k_conv_hx (the launchers' own conditions), the
        // broadcast-input / bias sums on the S16-aware reductions
        bool elig = L.pd.KS == 3 && prec_bwd != PREC_FP32 && L.wq && (L.pd.Cout & 31) == 0 && L.pd.Cout >= 32 && L.pd.Ktot >= 32 && W >= 8 && H >= 2 && (out.ld & 3) == 0 && (out.sn & 3) == 0;
        for (int s = 0; s < nseg && elig; s++) {
            if ((segs[s].t.ld & 3) || (segs[s].t.sn & 3)) elig = false;
            if (segs[s].need_grad && !segs[s].bcast && !L.wqd[s]) elig = false;
        }
        if (elig) {
            gfmts.push_back(GradFmt{true, 0});
            out.gs = &gfmts.back();
            if (!dbg.empty() && dbg.back().d == out.d) dbg.back().gs = out.gs;
        }
    }
    if (res) { a.res = res->d; a.res_sn = res->sn; a.res_ld = res->ld; }
    a.out_sn = out.sn; a.out_ld = out.ld; a.accumulate = 0;
    g_last_conv_lstm_fused = 0;
    if (lstm_fuse) { a.lstm = lstm_fuse; lstm_fuse = nullptr; }      // (set by lstm_step for the gate convolution of a roll-out cell)
    for (int s = 0; s < nseg; s++)
        if (segs[s].t.bn_scale && !conv_src_lazy_ok(a)) { fail = true; set_error("internal: lazily normalised input handed to a convolution that cannot apply it"); }
    TileStats* ts_slot = nullptr;
    if (want_stats) {      // per-tile BatchNorm partial sums from the epilogue; sized for the smallest tile (8 x 16 pixels) whatever the launcher picks (dry-run safe)
        want_stats = false;
        if (epi_stats && training && !into && a.wq) {
            const int ldp = round_up(L.pd.Cout, 4);
            float* part = falloc((size_t)conv_stats_tiles_cap(N, H, W) * ldp * 2);
            a.stats = part; a.stats_ld = ldp;
            ts_slot = &stats_ring[stats_next]; stats_next ^= 1;
            *ts_slot = TileStats{out.d, part, 0, ldp};
        }
    }
    const double px_taps = 2.0 * N * H * W * L.pd.KS * L.pd.KS;     // algorithmic FLOPs = px_taps * Cin * Cout
    if (pooled) {
        //  conv_avgpool_ok() decided before `out` existed.  Should the launch decline after all (-1 from a launcher-side test that needs the real pointers, or a layer
        // that moved to the exact-fp32 fall-back between sizing and this pass), run conv + pool2 instead of failing the pass; the sizing run reserves the full-resolution map so
        // that the arena holds either layout.
        T4 full = dry ? alloc(N, H, W, L.pd.Cout) : T4{};
        (void)full;
        if (!dry && !fail) {
            const int rc = timed_conv_fwd(a, px_taps * L.pd.Cin * L.pd.Cout);
            if (rc != 0) {
                T4 f2 = alloc(N, H, W, L.pd.Cout);
                const int pact = a.act == 3 ? 1 : 0;
                a.avgpool = 0; a.act = 0; a.out = f2.d; a.out_sn = f2.sn; a.out_ld = f2.ld;
                RUN(timed_conv_fwd(a, px_taps * L.pd.Cin * L.pd.Cout));
                RUN(pw_pool2(dv(f2), dv(out), stream, pact));
                n_pool_fallback++;
            }
        }
    } else
    RUN(timed_conv_fwd(a, px_taps * L.pd.Cin * L.pd.Cout));
    if (ts_slot && !dry) ts_slot->tiles = g_last_conv_stats_tiles;
    if (recording) {
        T4 dz{}; if (actf == 1) dz = alloc(N, H, W, L.pd.Cout);
        Seg sg[CONV_MAX_SRC]; T4 tmp[CONV_MAX_SRC];
        for (int s = 0; s < nseg; s++) {   // broadcast inputs: scratch for the border-aware sums S[N][Cout][9] (3x3) or a dgrad temp otherwise
            sg[s] = segs[s];
            if (segs[s].bcast && segs[s].need_grad) tmp[s] = L.pd.KS == 3 ? alloc(N, 1, 1, L.pd.Cout * 9) : alloc(N, H, W, segs[s].t.C);
        }
        ConvL* Lp = &L;
        tp->push_back([=]() {
            TV dzv = gv(out);
            if (actf == 1) { RUN(pw_tanh_bwd(gv(out), dv(out), dv(dz), stream)); dzv = dv(dz); }
            WgradArgs w{};
            fill_srcs(w.src, sg, nseg);
            w.nsrc = nseg; w.N = N; w.H = H; w.W = W; w.KS = Lp->pd.KS; w.dy = dzv.p; w.dy_sn = dzv.sn; w.dy_ld = dzv.ld;
            w.Cout = Lp->pd.Cout; w.Cout_pad = Lp->pd.Cout_pad; w.Ktot = Lp->pd.Ktot; w.dwp = Lp->dwp; w.slabs = 0; w.dy_s16 = dzv.s16;
            w.precision = ((Lp->wq || (Lp->pd.Cout <= 3 && Lp->pd.KS == 7)) && prec_bwd != PREC_FP32) ? PREC_BF16X3 : PREC_FP32;      // (7x7 FinalBlock head: split bf16 on conv_stream.hip's k_wgrad_head7)
            queue_wgrad(Lp, w, px_taps * Lp->pd.Cin * Lp->pd.Cout);
            bool bias_done = !Lp->dbias;
            bool any_aux = !bias_done;
            for (int s = 0; s < nseg; s++) any_aux = any_aux || (sg[s].need_grad && sg[s].bcast && Lp->pd.KS == 3);
            if (any_aux) {      // off the BPTT chain: nothing before the action network's backward / the optimiser reads these (dz and the per-call scratch stay valid)
                const bool bias_first = bias_done;
                std::array<T4, CONV_MAX_SRC> tmpv{tmp[0], tmp[1], tmp[2]};
                std::array<Seg, CONV_MAX_SRC> sgv{sg[0], sg[1], sg[2]};
                defer_aux([=]() {
                    bool bd = bias_first;
                    for (int s = 0; s < nseg; s++) {      // broadcast inputs first: their border-aware sums of dY contain the bias gradient
                        if (!(sgv[s].need_grad && sgv[s].bcast && Lp->pd.KS == 3)) continue;
                        RUN(pw_bcast_input_grad(dzv, Lp->pd, s, tmpv[s].d, sgv[s].t.g, sgv[s].t.sn, bd ? nullptr : Lp->dbias, stream));
                        bd = true;
                    }
                    if (!bd) RUN(pw_colsum(dzv, Lp->dbias, stream, deterministic, red_scratch));
                });
            }
            for (int s = 0; s < nseg; s++) {
                if (!sg[s].need_grad) continue;
                if (sg[s].bcast && Lp->pd.KS == 3) continue;
                ConvArgs d{};
                d.src[0] = ConvSrc{dzv.p, dzv.sn, dzv.ld, Lp->pd.Cout, Lp->kd, 0};
                d.nsrc = 1; d.N = N; d.H = H; d.W = W; d.KS = Lp->pd.KS; d.wp = Lp->wpd[s]; d.Ktot = Lp->kd;
                d.Cout = sg[s].t.C; d.Cout_pad = Lp->cd_pad[s]; d.bias = nullptr; d.act = 0; d.aux = conv_aux; d.deterministic = deterministic ? 1 : 0; d.in_s16 = dzv.s16;
                if (Lp->wqd[s] && prec_bwd != PREC_FP32) { d.wq = Lp->wqd[s]; d.precision = PREC_BF16X3; }
                else if (Lp->pd.Cout <= 3 && Lp->pd.KS == 7 && prec_bwd != PREC_FP32) d.precision = PREC_BF16X3;      // 7x7 FinalBlock head: split bf16 on conv_head.hip (weights split in the kernel)
                const double dfl = px_taps * sg[s].t.C * Lp->pd.Cout;
                const int kind_save = prof_kind_override;
                if (prof_kind_override < 0) prof_kind_override = 1;      // profiling: a dgrad launch, whether it assigns or accumulates
                struct KindRestore { int& k; int v; ~KindRestore() { k = v; } } kind_restore{prof_kind_override, kind_save};
                if (!sg[s].bcast) {
                    // first-touch inputs (this conv is their only consumer): dgrad assigns -- plain stores, or the deterministic slab split-K when under-filled
                    const bool assign = sg[s].t.nz && !sg[s].t.nz2;      // nz2: a point-wise writer (residual add / up-sampling backward) assigned before this dgrad runs
                    d.out = sg[s].t.g; d.out_sn = sg[s].t.sn; d.out_ld = sg[s].t.ld; d.accumulate = assign ? 0 : 1;
                    // d(h_{t-1}) of a ConvLSTM's gate convolution: first read by the previous time step's cell backward, a whole R -> E -> D backward later.  The launch
                    // (under-filled: 16x16 / 32x32 maps) goes to the decoder stream, beside the chain; lstm_step's BatchNorm backward joins it.
                    if (sg[s].off_chain && !assign && !dry && !in_d && aux_enabled()) {
                        const int pk = prof_kind_override;
                        Lp->off_pending = true;      // (the consumer flushes the queued jobs before it waits for the event)
                        defer_aux([=]() mutable {      // (runs with stream / scratch of the auxiliary stream)
                            d.aux = conv_aux;
                            if (deterministic) { d.split_scratch = conv_split; d.split_cap = conv_split_cap; }
                            const int keepk = prof_kind_override; prof_kind_override = pk;
                            RUN(timed_conv_fwd(d, dfl));
                            prof_kind_override = keepk;
                            if (!Lp->off_ev) hipEventCreateWithFlags(&Lp->off_ev, hipEventDisableTiming);
                            hipEventRecord(Lp->off_ev, stream);
                        });
                    } else {
                        if (assign || deterministic) { d.split_scratch = conv_split; d.split_cap = conv_split_cap; }
                        RUN(timed_conv_fwd(d, dfl));
                    }
                }
                else {
                    d.out = tmp[s].d; d.out_sn = tmp[s].sn; d.out_ld = tmp[s].ld; d.accumulate = 0;
                    RUN(timed_conv_fwd(d, dfl));
                    RUN(pw_spatial_sum(dv(tmp[s]), sg[s].t.g, sg[s].t.sn, stream, deterministic));
                }
            }
        });
    }
    return out;
}

// conv -> avg_pool2d(2) [-> LeakyReLU] of the BatchNorm-folded inference graph as ONE launch where the kernel can (conv_avgpool_ok), conv + pool2 otherwise
T4 caddy_ctx::conv_pool(ConvL& L, const Seg* segs, int nseg, bool actf) {
    pool_fuse = actf ? 2 : 1;
    T4 o = conv(L, segs, nseg, 0, nullptr);
    if (pool_fuse) { pool_fuse = 0; o = pool2(o, actf); }
    return o;
}
T4 caddy_ctx::pool2(const T4& x, bool actf) {
    T4 o = x.nz ? alloc_nz(x.N, x.H / 2, x.W / 2, x.C) : alloc(x.N, x.H / 2, x.W / 2, x.C);   // conv -> pool -> BatchNorm chains stay first-touch
    RUN(pw_pool2(dv(x), dv(o), stream, actf ? 1 : 0));
    if (recording) tp->push_back([=]() { RUN(pw_pool2_bwd(gv(o), gv_w(x, x.nz), x.nz ? 1 : 0, stream)); });
    return o;
}
T4 caddy_ctx::up2(const T4& x) {
    T4 o = alloc_nz(x.N, x.H * 2, x.W * 2, x.C);      // every up-sampled map feeds exactly one conv (UpBlock / ConvLSTM gates): its dgrad assigns, no zero-fill, no read-modify-write
    RUN(pw_up2(dv(x), dv(o), stream));
    if (recording) tp->push_back([=]() { RUN(pw_up2_bwd(gv(o), gv(x), stream, x.nz2 ? 1 : 0)); });      // x.nz2: this is the first writer of d(x)
    return o;
}

struct BNStash { float *mean, *invstd, *scale, *shift, *uvar; double* sums; };
static BNStash bn_stash(caddy_ctx* c, BNL& bn) {
    BNStash s; float* f = c->falloc(5 * (size_t)round_up(bn.C, 4));
    int cp = round_up(bn.C, 4);
    c->dbg.push_back(T4{f, f, 1, 1, 1, 4 * cp, 4 * cp, 4 * cp});
    s.mean = f; s.invstd = f + cp; s.scale = f + 2 * cp; s.shift = f + 3 * cp; s.uvar = f + 4 * cp; s.sums = c->dalloc(2 * (size_t)bn.C);
    return s;
}
// running statistics: updated by the finalisation itself, or (BNL::deferred) left as (mean, unbiased variance) of this call for the in-order k_bn_ema at the end of the forward
static inline float* rm_of(BNL& bn) { return bn.deferred ? nullptr : bn.rmean; }
static inline float* rv_of(BNL& bn, const BNStash& s) { return bn.deferred ? s.uvar : bn.rvar; }
static inline void bn_called(caddy_ctx* c, BNL& bn, const BNStash& s) { if (!c->dry) { bn.calls++; if (bn.deferred) bn.pend.emplace_back(s.mean, s.uvar); } }
static BNStash bn_forward(caddy_ctx* c, const T4& x, BNL& bn) {
    BNStash s = bn_stash(c, bn);
    bool dry = c->dry;
    if (c->training) {
        const caddy_ctx::TileStats* ts = dry ? nullptr : c->find_stats(x.d);
        if (!dry) c->n_bn_calls++;
        if (ts) c->n_bn_tile_stats++;
        if (ts)      // the producing convolution left per-tile partial sums behind: no pass over x
            c->ck(pw_bn_finalize_tiles(ts->part, ts->tiles, ts->ldp, (long)x.N * x.H * x.W, bn.gamma, bn.beta, rm_of(bn), rv_of(bn, s), bn.C, s.mean, s.invstd, s.scale, s.shift, c->stream), "bn_finalize_tiles");
        else if (!dry) c->ck(pw_bn_stats_finalize(dv(x), s.sums, c->red_scratch, bn.gamma, bn.beta, rm_of(bn), rv_of(bn, s), s.mean, s.invstd, s.scale, s.shift, c->stream), "bn_stats_finalize");
        bn_called(c, bn, s);
        bn.eval_valid = false;
        return s;
    }
    // eval mode: the affine form depends only on the parameters -> once per roll-out (generate_next runs ~25 BatchNorms per frame)
    int cp = round_up(bn.C, 4);
    s.mean = bn.eval_stash; s.invstd = bn.eval_stash + cp; s.scale = bn.eval_stash + 2 * cp; s.shift = bn.eval_stash + 3 * cp;
    if (!dry && !bn.eval_valid) {
        c->ck(pw_bn_finalize(s.sums, (long)x.N * x.H * x.W, bn.gamma, bn.beta, bn.rmean, bn.rvar, bn.C, 0, s.mean, s.invstd, s.scale, s.shift, c->stream), "bn_finalize");
        bn.eval_valid = true;
    }
    return s;
}
// can `consumer` (a 3x3 convolution reading x's normalised form as one of its segments, same H x W) apply the BatchNorm while staging -- forward AND weight gradient?
bool caddy_ctx::lazy_ok(const ConvL& consumer, const T4& x) const {
    if (!lazy_bn || !consumer.wq || prec_fwd == PREC_FP32 || consumer.pd.KS != 3 || layer_fallback[consumer.flag_idx]) return false;
    if (recording) {      // its weight gradient must run on k_wgrad_hx (conv_hx_wgrad_try's conditions)
        if (prec_bwd == PREC_FP32 || consumer.pd.Cout < 32 || consumer.pd.Ktot < 32 || x.W < 8 || x.H < 2) return false;
    }
    return (x.ld & 3) == 0 && (x.sn & 3) == 0;
}

T4 caddy_ctx::bn_act(const T4& x, BNL& bn, const T4* x2, BNL* bn2, bool actf, const T4* into, bool nz_out, bool nz2_out, const ConvL* lazy_for) {
    const bool small = training && !bn2 && bn_small && pw_bn_small_pays(dv(x));      // one-launch path for R's small maps
    // lazily applied form: statistics + finalisation only; the single consumer (a k_conv_hx / k_wgrad_hx convolution) forms act(x * scale + shift) while
    // staging its input and the backward takes the LeakyReLU slope from the same expression -- the normalised tensor never exists in HBM
    if (lazy_for && !x2 && !into && !small && !fold && x.ld == round_up(x.C, 4) && x.sn == (long)x.H * x.W * x.ld && lazy_ok(*lazy_for, x)) {
        BNStash s1 = bn_forward(this, x, bn);
        if (!dry) n_bn_lazy++;
        T4 ag = alloc_nz(x.N, x.H, x.W, x.C);      // only its gradient half is used: d(loss) / d(normalised value), assigned by the consumer's dgrad
        T4 out = x;
        out.g = ag.g; out.nz = true; out.nz2 = false; out.gs = nullptr;      // (its own gradient buffer: written in fp32 by the consumer's dgrad)
        out.bn_scale = s1.scale; out.bn_shift = s1.shift; out.bn_act = actf ? 1 : 0;
        if (recording) {
            BNL* b1 = &bn;
            tp->push_back([=]() {
                const float* ls = actf ? s1.scale : nullptr;
                RUN(pw_bn_bwd_reduce(gv(out), nullptr, dv(x), s1.mean, s1.invstd, s1.sums, red_scratch, bn_dgamma(b1), bn_dbeta(b1), stream, ls, s1.shift));
                RUN(pw_bn_bwd_apply(gv(out), nullptr, dv(x), s1.mean, s1.invstd, b1->gamma, s1.sums, gv_w(x, x.nz), nullptr, nullptr, x.nz ? 1 : 0, stream, ls, s1.shift));
            });
        }
        return out;
    }
    T4 out = into ? *into : ((nz_out || nz2_out) ? alloc_nz(x.N, x.H, x.W, x.C, nz2_out) : alloc(x.N, x.H, x.W, x.C));
    TV x2v{}; if (x2) x2v = dv(*x2);
    BNStash s1{}, s2{};
    if (small) {
        s1 = bn_stash(this, bn);
        RUN(pw_bn_small_fwd(dv(x), bn.gamma, bn.beta, rm_of(bn), rv_of(bn, s1), s1.mean, s1.invstd, s1.scale, s1.shift, x2 ? &x2v : nullptr, actf ? 1 : 0, dv(out), stream));
        bn_called(this, bn, s1);
    } else {
        s1 = bn_forward(this, x, bn);
        if (x2 && bn2) s2 = bn_forward(this, *x2, *bn2);
        RUN(pw_bn_apply(dv(x), s1.scale, s1.shift, x2 ? &x2v : nullptr, bn2 ? s2.scale : nullptr, bn2 ? s2.shift : nullptr, actf ? 1 : 0, dv(out), stream));
    }
    if (recording) {
        T4 x2c{}; if (x2) x2c = *x2;
        bool has2 = x2 != nullptr; BNL* b1 = &bn; BNL* b2 = bn2;
        tp->push_back([=]() {
            TV om = dv(out);
            const TV* omp = actf ? &om : nullptr;
            // single-input BatchNorm + LeakyReLU: the slope decision of act(x * scale + shift) is recomputed from x, which both passes read anyway (the same fused multiply-add as
            // the forward's FBnApply: bit-identical decisions) -- the materialised output is not read a second and third time (round 6)
            const bool mask_x = actf && !has2 && !small && mask_from_x;
            const TV* omr = mask_x ? nullptr : omp;
            const float* lsx = mask_x ? s1.scale : nullptr;
            if (small) {
                TV dres{}; if (has2) dres = gv(x2c);
                RUN(pw_bn_small_bwd(gv(out), omp, dv(x), s1.mean, s1.invstd, b1->gamma, gv_w(x, x.nz), bn_dgamma(b1), bn_dbeta(b1), has2 ? &dres : nullptr, x.nz ? 1 : 0, stream, (has2 && x2c.nz2) ? 1 : 0));
                return;
            }
            RUN(pw_bn_bwd_reduce(gv(out), omr, dv(x), s1.mean, s1.invstd, s1.sums, red_scratch, bn_dgamma(b1), bn_dbeta(b1), stream, lsx, s1.shift));   // sums assigned; param grads fused
            RUN(pw_bn_bwd_apply(gv(out), omr, dv(x), s1.mean, s1.invstd, b1->gamma, s1.sums, gv_w(x, x.nz), nullptr, nullptr, x.nz ? 1 : 0, stream, lsx, s1.shift));
            if (has2 && b2) {
                RUN(pw_bn_bwd_reduce(gv(out), omp, dv(x2c), s2.mean, s2.invstd, s2.sums, red_scratch, bn_dgamma(b2), bn_dbeta(b2), stream));
                RUN(pw_bn_bwd_apply(gv(out), omp, dv(x2c), s2.mean, s2.invstd, b2->gamma, s2.sums, gv_w(x2c, x2c.nz), nullptr, nullptr, x2c.nz ? 1 : 0, stream));
            } else if (has2) {
                if (actf) RUN(pw_act_bwd_add(gv(out), dv(out), gv(x2c), stream, x2c.nz2 ? 1 : 0));      // identity path: first writer of d(x) when x is nz2
                else RUN(pw_copy(gv(out), gv(x2c), x2c.nz2 ? 0 : 1, stream));
            }
        });
    }
    return out;
}

// ResidualBlock (model/layers/residual_block.py:51-68)
T4 caddy_ctx::resblock(ResL& R, const T4& x, const T4* into, bool nz2_out) {
    Seg sx{x, 0, true};
    if (fold) {      // roll-out: BatchNorms folded into the convs -- conv1' (+ pool) + LeakyReLU, then act(conv2'(a) + identity) in conv2's epilogue
        T4 a1 = R.ds == 2 ? conv_pool(R.conv1, &sx, 1, true) : conv(R.conv1, &sx, 1, 3, nullptr);
        Seg sa{a1, 0, true};
        if (!R.has_down) return conv(R.conv2, &sa, 1, 3, into, false, &x);
        T4 idn = R.ds == 2 ? conv_pool(R.down, &sx, 1, false) : conv(R.down, &sx, 1, 0, nullptr);
        return conv(R.conv2, &sa, 1, 3, into, false, &idn);
    }
    want_stats = R.ds == 1;                                   // (a pooled map's statistics are not the conv output's)
    T4 c1 = conv(R.conv1, &sx, 1, 0, nullptr, true);        // conv -> (pool) -> BatchNorm: single assigning gradient writer
    if (R.ds == 2) c1 = pool2(c1);
    T4 a1 = bn_act(c1, R.bn1, nullptr, nullptr, true, nullptr, true, false, &R.conv2);      // consumed by conv2 only: never materialised when conv2 runs on k_conv_hx
    Seg sa{a1, 0, true};
    want_stats = true;
    T4 c2 = conv(R.conv2, &sa, 1, 0, nullptr, true);
    if (R.has_down) {
        want_stats = R.ds == 1;
        T4 idn = conv(R.down, &sx, 1, 0, nullptr, true);
        if (R.ds == 2) idn = pool2(idn);
        return bn_act(c2, R.bn2, &idn, &R.bnd, true, into, false, nz2_out);
    }
    return bn_act(c2, R.bn2, &x, nullptr, true, into, false, nz2_out);
}

// RepresentationNetwork.forward (model/main_model/representation_network.py:32-58); output keeps the 65th (attention) channel
T4 caddy_ctx::encode(const T4& obs_in, bool input_grad, const T4* into) {
    Seg so{obs_in, 0, input_grad};
    T4 x;
    if (fold) x = conv_pool(e_stem, &so, 1, true);
    else { x = conv(e_stem, &so, 1, 0, nullptr, true); x = pool2(x, false); }
    // (outputs consumed by a residual block WITHOUT down-sampling path -- identity add + one conv -- get first-touch gradients: T4::nz2)
    if (!fold) x = bn_act(x, e_bn1, nullptr, nullptr, true, nullptr, false, !e_res[0].has_down);
    for (int i = 0; i < 5; i++) x = resblock(e_res[i], x, nullptr, !e_res[i + 1].has_down);
    T4 dst = into ? *into : alloc(x.N, hs, ws, 65, 68);
    return resblock(e_res[5], x, &dst);
}

void caddy_ctx::copy_op(const T4& src, const T4& dst) {
    RUN(pw_copy(dv(src), dv(dst), 0, stream));
    if (recording) tp->push_back([=]() { RUN(pw_copy(gv(dst), gv(src), 1, stream)); });
}

// ConvLSTM step (convolutional_lstm.py:50-74, convolutional_lstm_cell.py:88-101) followed by its BatchNorm
T4 caddy_ctx::lstm_step(int i, const T4& x, const T4& aux, const ConvL* next) {
    LstmL& L = lstm[i];
    const int B = x.N;
    bool persistent = !training && !recording && L.h.d == L.ph.d && L.h.d != nullptr;
    const bool first_step = L.h.d == nullptr;      // h(t-1) = the learned initial state
    T4 hprev, cprev;
    if (L.h.d == nullptr) {   // lazily created from the learned initial state, repeated over the batch
        hprev = alloc(B, L.Hs, L.Ws, L.C); cprev = alloc(B, L.Hs, L.Ws, L.C);
        T4 ih = L.ih, ic = L.ic; ih.sn = 0; ic.sn = 0; ih.N = B; ic.N = B;
        RUN(pw_copy(dv(ih), dv(hprev), 0, stream)); RUN(pw_copy(dv(ic), dv(cprev), 0, stream));
        if (recording) {
            T4 ihg = L.ih, icg = L.ic;
            ConvL* Lg = &L.gates;
            tp->push_back([=]() {
                if (Lg->off_pending && !dry) { flush_aux(); hipStreamWaitEvent(stream, Lg->off_ev, 0); Lg->off_pending = false; }      // d(initial h) comes from the first step's gate-convolution dgrad on the decoder stream
                RUN(pw_batch_sum(hprev.g, hprev.sn, hprev.sn, B, ihg.g, stream));
                RUN(pw_batch_sum(cprev.g, cprev.sn, cprev.sn, B, icg.g, stream));
            });
        }
    } else { hprev = L.h; cprev = L.c; }
    Seg sg[3] = {{x, 0, true}, {aux, 1, true}, {hprev, 0, true}};
    sg[2].off_chain = !first_step;      // (the first step's d(h) is read right behind its cell backward, by the batch sum into the initial state's gradient: off the chain it was enqueued
                                        //  at that point and waited for -- 0.1 - 0.35 ms of idle main stream per ConvLSTM, profiles/r05_experiments.md)
    T4 hn, cn;
    if (persistent) { hn = L.ph; cn = L.pc; hn.N = B; cn.N = B; } else { hn = alloc(B, L.Hs, L.Ws, L.C); cn = alloc(B, L.Hs, L.Ws, L.C); }
    if (fold) {      // roll-out: the cell's eval-mode BatchNorm is a second output of the cell update; when the gate convolution is split over K its slab reduce applies
                     // the update itself (ConvArgs.lstm: one launch instead of k_split_reduce + k_map<FLstmFwd>, and the gate tensor is never written)
        T4 hb = alloc(B, L.Hs, L.Ws, L.C);
        const int cp = round_up(L.bn.C, 4);
        LstmFuse lf{cprev.d, cprev.sn, cprev.ld, hn.d, hn.sn, hn.ld, cn.d, cn.sn, cn.ld, hb.d, hb.sn, hb.ld, L.bn.eval_stash + 2 * cp, L.bn.eval_stash + 3 * cp, L.C};
        // (the persistent roll-out state is updated in place -- cn aliases cprev, hn the convolution's third input: every (pixel, channel quad) is read and written by one
        //  thread of the reduce, which starts after the convolution's last read)
        const bool fusable = (cprev.ld & 3) == 0 && (hn.ld & 3) == 0 && (cn.ld & 3) == 0 && (hb.ld & 3) == 0 &&
                             (cprev.sn & 3) == 0 && (hn.sn & 3) == 0 && (cn.sn & 3) == 0 && (hb.sn & 3) == 0;
        if (fusable) lstm_fuse = &lf;
        T4 gates = conv(L.gates, sg, 3, 0, nullptr, true);
        lstm_fuse = nullptr;
        if (!g_last_conv_lstm_fused || dry) {
            TV hbv = dv(hb);
            RUN(pw_lstm_fwd(dv(gates), dv(cprev), dv(hn), dv(cn), stream, &hbv, L.bn.eval_stash + 2 * cp, L.bn.eval_stash + 3 * cp));
        }
        L.h = hn; L.c = cn;
        return hb;
    }
    T4 gates = conv(L.gates, sg, 3, 0, nullptr, true);      // d(gates) is assigned by the LSTM point-wise backward
    RUN(pw_lstm_fwd(dv(gates), dv(cprev), dv(hn), dv(cn), stream));
    if (recording) tp->push_back([=]() { RUN(pw_lstm_bwd(dv(gates), dv(cprev), dv(cn), gv(hn), gv(cn), gv_w(gates, gates.nz), gv(cprev), stream)); });
    L.h = hn; L.c = cn;
    T4 hb = bn_act(hn, L.bn, nullptr, nullptr, false, nullptr, true, false, next);      // feeds exactly one conv (as its first segment, same resolution)
    if (recording) { ConvL* Lg = &L.gates;      // (reverse replay: first thing of this step's cell) d(h_t) also receives the NEXT step's gate-convolution dgrad, from the decoder stream
        tp->push_back([=]() { if (Lg->off_pending && !dry) { flush_aux(); hipStreamWaitEvent(stream, Lg->off_ev, 0); Lg->off_pending = false; } }); }
    return hb;
}

// ConvDynamicsNetwork.forward (model/main_model/conv_dynamics_network.py:111-133)
T4 caddy_ctx::dynamics(const T4& state, const T4& aux, const T4* into) {
    T4 x = lstm_step(0, state, aux, &r_c0);
    Seg s0[2] = {{x, 0, true}, {aux, 1, true}};
    if (fold) x = conv_pool(r_c0, s0, 2, true);      // (K-split launch: its slab reduce pools)
    else { x = conv(r_c0, s0, 2, 0, nullptr, true); x = pool2(x, false); }
    if (!fold) x = bn_act(x, r_bn0, nullptr, nullptr, true, nullptr, true, false, &lstm[1].gates);      // -> ConvLSTM 1 gates conv only
    x = lstm_step(1, x, aux, &r_c1);
    Seg s1[2] = {{x, 0, true}, {aux, 1, true}};
    want_stats = !fold;
    x = conv(r_c1, s1, 2, fold ? 3 : 0, nullptr, true);
    if (!fold) x = bn_act(x, r_bn1, nullptr, nullptr, true, nullptr, false, true);      // -> bilinear x2 only: its backward assigns
    x = up2(x);
    x = lstm_step(2, x, aux, &r_c2);
    Seg s2[2] = {{x, 0, true}, {aux, 1, true}};
    if (fold) return conv(r_c2, s2, 2, 3, into);
    want_stats = true;
    x = conv(r_c2, s2, 2, 0, nullptr, true);
    return bn_act(x, r_bn2, nullptr, nullptr, true, into);
}

// RenderingNetwork.forward (model/main_model/rendering_network.py:52-71): three upsampling stages, each with a tanh head
void caddy_ctx::render(const T4& hdn, int slot, int nslots) {
    T4 x = hdn;
    const int B = hdn.N;
    for (int i = 0; i < 3; i++) {
        T4 u = up2(x);
        Seg su{u, 0, true};
        want_stats = !fold;
        T4 c = conv(d_up[i], &su, 1, fold ? 3 : 0, nullptr, true);
        x = fold ? c : bn_act(c, d_norm[i], nullptr, nullptr, true, nullptr, i == 2, i < 2);      // last stage -> 7x7 FinalBlock only; others -> residual block (identity add + conv)
        if (i < 2) x = resblock(d_res[i], x, nullptr, true);      // -> next stage's up-sampling (assigns) + this stage's FinalBlock conv (accumulates)
        if (rollout && i < 2) continue;      // generate_next returns the full-resolution frame only (model.py:597-601): the two low-resolution heads are dead code there
        T4 dst = tslice(frames[2 - i], B, nslots, slot);
        Seg sx{x, 0, true};
        conv(d_final[i], &sx, 1, 1, &dst);
    }
}

// ActionNetwork.forward (model/main_model/action_network.py:62-118) + sampling (model.py:166-201) for the first call
void caddy_ctx::action_net(const T4& x65, HeadState& H, const float* eps_s, const float* eps_d, const float* unif, bool first,
                           const float* samples_in, const float* variations_in) {
    const int B = cfg.batch, T = cfg.seq_len, K = cfg.actions, Da = cfg.action_dim, NT = B * T, NS = B * (T - 1);
    H.x65 = x65;
    T4 st = alloc(NT, hs, ws, 64);
    H.att = alloc(NT, hs, ws, 1);
    RUN(pw_attn_mul(dv(x65), dv(st), dv(H.att), stream));
    if (recording) { T4 att = H.att; tp->push_back([=]() { TV none{}; (void)att; RUN(pw_attn_mul_bwd(dv(x65), gv(st), none, gv(x65), stream)); }); }
    ResL* const ar = a_res[member];                          // the member drawn for this forward pass (both A calls: model.py:152,274)
    const HeadParams hpv = hp[member];
    T4 r = resblock(ar[0], st, nullptr, !ar[1].has_down);
    r = resblock(ar[1], r, nullptr);
    HeadBufs& b = H.b;
    float* feat = falloc((size_t)NT * hpv.F);
    RUN(pw_gap(dv(r), feat, stream));
    b.feat = feat; b.d_feat = tw(this, feat);
    if (recording) { float* df = b.d_feat; tp->push_back([=]() { RUN(pw_gap_bwd(df, gv(r), stream)); }); }
    b.eps_s = eps_s; b.eps_d = eps_d; b.unif = unif;
    b.mu = falloc((size_t)NT * Da); b.raw = falloc((size_t)NT * Da);
    b.sdist = falloc((size_t)NT * 2 * Da); b.ssamp = falloc((size_t)NT * Da);
    b.ddist = falloc((size_t)NS * 2 * Da); b.dirs = falloc((size_t)NS * Da);
    b.logits = falloc((size_t)NS * K); b.logp = falloc((size_t)NS * K); b.prob = falloc((size_t)NS * K);
    b.ysoft = falloc((size_t)NS * K); b.samples = falloc((size_t)NS * K); b.variations = falloc((size_t)NS * Da);
    b.aux = falloc((size_t)NS * AUX_LD); b.cen_used = falloc(16 * 8);
    b.selected = (long long*)act.alloc(sizeof(long long) * NS);
    b.g_dmu = falloc((size_t)NS * Da); b.g_dvar = falloc((size_t)NS * Da);
    b.g_mu = falloc((size_t)NT * Da); b.g_raw = falloc((size_t)NT * Da);
    b.d_logits = tw(this, b.logits); b.d_ddist = tw(this, b.ddist); b.d_sdist = tw(this, b.sdist); b.d_aux = tw(this, b.aux);
    RUN(head_forward(b, hpv, B, T, stream));
    SampleCfg& sc = H.sc;
    sc = SampleCfg{};
    sc.mode = samples_in ? 2 : (cfg.use_gumbel ? 1 : 0); sc.hard = cfg.hard_gumbel; sc.training = training ? 1 : 0; sc.use_variations = cfg.use_variations;
    sc.tau = tau; sc.alpha = cfg.centroid_alpha; sc.centroids = centroids; sc.samples_in = samples_in; sc.variations_in = variations_in;
    if (first) {
        float* cs = falloc(16 * 8 + 16);
        SamplerHooks sh = samplers;
        sh.samples_buf = falloc((size_t)NS * K); sh.var_buf = falloc((size_t)NS * Da);       // always allocated: the arena layout must not depend on the hooks
        if (samples_in) sh.action = 0;
        if (variations_in) sh.variation = 0;
        RUN(head_sample(b, hpv, sc, NS, cs, hook, hook_user, sh.fn ? &sh : nullptr, stream));
        if (sh.fn && sh.action) { sc.mode = 2; sc.samples_in = sh.samples_buf; }             // what the backward pass must assume
        if (sh.fn && sh.variation) sc.variations_in = sh.var_buf;
    }
    if (recording) { HeadBufs bb = b; SampleCfg s2 = sc; tp->push_back([=]() { if (first) join_aux(stream);      // d(action / variation inputs) of every time step
                                                                               RUN(head_backward(bb, hpv, s2, B, T, first ? 1 : 0, stream)); }); }
}

void caddy_ctx::add_job(JobList& jl, const PackDesc& d, void* buf, int kind, int seg, int p0, int p1, long total) {
    PackJob j{};
    j.d = d; j.buf = buf; j.kind = kind; j.seg = seg; j.p0 = p0; j.p1 = p1; j.total = total;
    long nb = (total + 2047) / 2048;      // ~8 elements per thread
    j.nblocks = (int)(nb < 1 ? 1 : (nb > 256 ? 256 : nb));
    j.block0 = jl.blocks; jl.blocks += j.nblocks;
    jl.host.push_back(j);
}
void caddy_ctx::upload_jobs(JobList& jl, hipStream_t st) {
    if ((int)jl.host.size() > jl.cap) { fail = true; set_error("internal: pack job table overflow"); return; }
    hipMemcpyAsync(jl.dev, jl.host.data(), sizeof(PackJob) * jl.host.size(), hipMemcpyHostToDevice, st);
    hipStreamSynchronize(st);      // (once per mode: the host vector is pageable)
}
void caddy_ctx::pack_all(bool with_fold) {
    packed_fold = with_fold;
    if (!with_fold && merged_pack && !dry) {      // training / evaluation graphs: every layer's forms in ONE launch
        const int key = (prec_fwd != PREC_FP32 ? 1 : 0) | ((prec_bwd != PREC_FP32 && recording) ? 2 : 0);
        JobList& jl = pack_jobs;
        if (jl.key != key) {
            jl.host.clear(); jl.blocks = 0;
            for (ConvL* L : convs) {
                const PackDesc& d = L->pd;
                const long taps = (long)d.KS * d.KS;
                add_job(jl, d, L->wp, PJ_FWD, -1, 0, 0, taps * d.Cout_pad * d.Ktot);
                for (int s = 0; s < d.nseg; s++) add_job(jl, d, L->wpd[s], PJ_DGRAD, s, L->cd_pad[s], L->kd, taps * L->cd_pad[s] * L->kd);
                if (L->wq && (key & 1)) { const int rp = round_up(d.Cout, hx_pick_bn(d.Cout)); add_job(jl, d, L->wq, PJ_HX_FWD_F16, -1, rp, 0, taps * hx_kq(d, -1) * rp); }
                for (int s = 0; s < d.nseg; s++)
                    if (L->wqd[s] && (key & 2)) { const int rp = round_up(d.seg_C[s], hx_pick_bn(d.seg_C[s])); add_job(jl, d, L->wqd[s], PJ_HX_DGRAD_BF16, s, rp, 0, taps * hx_kq(d, s) * rp); }
            }
            upload_jobs(jl, stream);
            jl.key = key;
        }
        RUN(pack_jobs_launch(jl.dev, (int)jl.host.size(), jl.blocks, stream));
    } else
    for (ConvL* L : convs) {
        PackDesc fpd = L->pd;      // forward forms: optionally with the following eval-mode BatchNorm folded in (roll-out)
        if (with_fold && L->fold_bn) {
            const int cp = round_up(L->fold_bn->C, 4);
            fpd.oscale = L->fold_bn->eval_stash + 2 * cp;
            RUN(pw_fold_bias(L->bias, fpd.oscale, L->fold_bn->eval_stash + 3 * cp, L->fold_bias, L->pd.Cout, stream));
        }
        RUN(pack_fwd(fpd, L->wp, stream));
        if (!with_fold) for (int s = 0; s < L->pd.nseg; s++) RUN(pack_dgrad(L->pd, s, L->wpd[s], L->cd_pad[s], L->kd, stream));
        if (L->wq && prec_fwd != PREC_FP32) RUN(pack_hx(fpd, L->wq, round_up(L->pd.Cout, hx_pick_bn(L->pd.Cout)), -1, PREC_F16X3, stream));
        for (int s = 0; s < L->pd.nseg; s++)
            if (L->wqd[s] && prec_bwd != PREC_FP32 && recording) RUN(pack_hx(L->pd, L->wqd[s], round_up(L->pd.seg_C[s], hx_pick_bn(L->pd.seg_C[s])), s, PREC_BF16X3, stream));
    }
    for (int i = 0; i < 3; i++) {   // learned initial LSTM states: (C,h,w) -> (1,h,w,C)
        RUN(pw_nchw_to_nhwc(lstm[i].init_h, 0, dv(lstm[i].ih), stream));
        RUN(pw_nchw_to_nhwc(lstm[i].init_c, 0, dv(lstm[i].ic), stream));
    }
}
// packed weight gradients -> the flat gradient buffer (reference layout): which = 0 every layer, 1 the early-bucket layers, 2 the others
static void unpack_layers(caddy_ctx* c, int which, hipStream_t st) {
    bool dry = c->dry;
    if (c->merged_pack && !dry) {
        caddy_ctx::JobList& jl = c->unpack_jobs[which];
        if (jl.key != 1) {
            jl.host.clear(); jl.blocks = 0;
            for (ConvL* L : c->convs)
                if (which == 0 || (which == 1) == L->early_bucket) c->add_job(jl, L->pd, L->dwp, PJ_UNPACK, -1, 0, 0, (long)L->pd.KS * L->pd.KS * L->pd.Cout_pad * L->pd.Ktot);
            c->upload_jobs(jl, st);
            jl.key = 1;
        }
        c->ck(pack_jobs_launch(jl.dev, (int)jl.host.size(), jl.blocks, st), "unpack jobs");
        return;
    }
    for (ConvL* L : c->convs) if (which == 0 || (which == 1) == L->early_bucket) { if (!dry) c->ck(unpack_wgrad(L->pd, L->dwp, st), "unpack_wgrad"); }
}
void caddy_ctx::unpack_all() {
    bool early = false;
    for (ConvL* L : convs) { early = early || L->early_done; L->early_done = false; }
    unpack_layers(this, early ? 2 : 0, stream);
    for (int i = 0; i < 3 && !lstm_early_done; i++) {
        RUN(pw_nhwc_to_nchw(gv(lstm[i].ih), lstm[i].ginit_h, 0, 0, stream));
        RUN(pw_nhwc_to_nchw(gv(lstm[i].ic), lstm[i].ginit_c, 0, 0, stream));
    }
    lstm_early_done = false;
}
// called from the tape right after the time loop's backward (forward_full graphs only): everything that contributes to R's and D's
// parameter gradients has been enqueued.  Unpack those layers on the side stream (behind the queued wgrad chunks) and hand the two
// ranges to the caller, who starts their all-reduce while A and E-on-ground-truth-frames are still in their backward.
void caddy_ctx::early_gradient_buckets() {
    if (!grads_hook || dry) return;
    launch_wgrad_jobs();
    hipStream_t s2 = wgrad_stream();
    join_aux(s2);      // the conv bias gradients of R / D are part of the bucket ranges
    unpack_layers(this, 1, s2);
    for (ConvL* L : convs) if (L->early_bucket) L->early_done = true;
    for (int i = 0; i < 3; i++) {
        RUN(pw_nhwc_to_nchw(gv(lstm[i].ih), lstm[i].ginit_h, 0, 0, s2));
        RUN(pw_nhwc_to_nchw(gv(lstm[i].ic), lstm[i].ginit_c, 0, 0, s2));
    }
    lstm_early_done = true;
    for (int b = 0; b < 2; b++) if (bucket_hi[b] > bucket_lo[b]) grads_hook(G, bucket_lo[b], bucket_hi[b] - bucket_lo[b], (void*)s2, grads_user);
}

// ---------------------------------------------------------------------------------------------------------------------
// forward_full_model (model/main_model/model.py:84-286)
// ---------------------------------------------------------------------------------------------------------------------
static int forward_full(caddy_ctx* c, const float* obs, int gt_init, float tau, const caddy_noise* nz, int training,
                        const float* samples_in, const float* variations_in) {
    const caddy_config& g = c->cfg;
    const int B = g.batch, T = g.seq_len, H = g.height, W = g.width, S = g.stacking;
    bool dry = c->dry;
    if (gt_init <= 0) { set_error("To forward the full model specify a number of ground truth observations > 0"); return -2; }
    if (c->gt_prefetched && !dry) hipStreamWaitEvent(c->stream, c->gt_done, 0);      // a forward without a backward in between: the side stream may still read the old observations
    c->act.reset(); c->tape.clear(); c->tape2.clear(); c->tp = &c->tape; c->d_forked = false; for (BNL* b_ : c->bns) b_->pend.clear(); c->dbg.clear(); c->gfmts.clear(); c->gt_lo = c->gt_hi = 0; c->stats_ring[0] = c->stats_ring[1] = caddy_ctx::TileStats{};
    c->training = training != 0; c->recording = training != 0; c->gt_init = gt_init; c->tau = tau; c->pretraining = false;
    for (BNL* b : c->bns) b->eval_valid = false;
    for (int i = 0; i < 3; i++) { c->lstm[i].h.d = nullptr; c->lstm[i].c.d = nullptr; }
    c->mark("fwd:begin");
    c->pack_all();
    c->mark("fwd:packed");
    caddy_noise z{}; if (nz) z = *nz;
    // observations -> NHWC
    c->obs = c->alloc(B * T, H, W, 3 * S);
    if (!dry) c->ck(pw_nchw_to_nhwc(obs, (long)3 * S * H * W, dv(c->obs), c->stream), "obs layout");
    if (g.perceptual && training) vgg_gt_prefetch(c, T - 1, 1);      // VGG19 features of the ground-truth frames: side stream, beside the forward pass
    else c->gt_prefetched = false;
    c->x65_gt = c->encode(c->obs, false, nullptr);
    c->mark("fwd:E(gt)");
    c->action_net(c->x65_gt, c->head1, z.eps_states, z.eps_dirs, z.gumbel_uniform, true, samples_in, variations_in);
    c->mark("fwd:A1");
    c->rec_x65 = c->alloc(B * T, c->hs, c->ws, 65, 68);
    c->hidden = c->alloc(B * (T - 1), c->hs, c->ws, g.hidden);
    for (int r = 0; r < 3; r++) c->frames[r] = c->alloc(B * (T - 1), H >> r, W >> r, 3);
    for (int t = 0; t < gt_init && t < T; t++) c->copy_op(tslice(c->x65_gt, B, T, t), tslice(c->rec_x65, B, T, t));
    T4 aux_all{c->head1.b.aux, c->head1.b.d_aux, B * (T - 1), 1, 1, g.actions + g.action_dim, AUX_LD, AUX_LD};
    if (c->recording) c->tape.push_back([c]() { c->mark("bwd:time loop"); c->flush_all_wgrad(); c->early_gradient_buckets(); });      // runs AFTER the time loop's backward: the queued chunks (and the R / D gradient buckets) overlap with the A / E tail
    for (int t = 0; t < T - 1; t++) {
        if (c->recording) c->tape.push_back([c]() { c->step_boundary(); });      // (reverse replay: after this time step's backward)
        if (t == gt_init - 1) { c->mark("fwd:teacher-forced steps"); if (c->recording) c->tape.push_back([c]() { c->mark("bwd:closed-loop steps"); }); }
        T4 state = chan(tslice(c->rec_x65, B, T, t), 0, 64);
        T4 aux = tslice(aux_all, B, T - 1, t);
        T4 hslot = tslice(c->hidden, B, T - 1, t);
        T4 hdn = c->dynamics(state, aux, &hslot);
        const bool teacher_forced = t + 1 < gt_init;      // nothing feeds this step's frames back into the model (model.py:241-243)
        if (teacher_forced && c->use_dstream && c->recording) {
            c->enter_d(true);      // D(t) beside R(t + 1) ...: its tape entries go to tape2
            c->render(hdn, t, T - 1);
            c->leave_d();
            if (t + 2 >= gt_init || t + 2 >= T) c->tape.push_back([c]() {      // (reverse replay: runs right before R-bwd of the last teacher-forced step)
                if (c->d_forked && !c->dry) { hipStreamWaitEvent(c->stream, c->d_done, 0); c->d_forked = false; c->fold_d_bn_grads(); }
                else if (!c->tape2_done) { c->perc_wait(0); c->replay_tape2(false); }      // (pipelined perceptual pass: the teacher-forced decoder steps need every chunk's seeds)
            });
        } else c->render(hdn, t, T - 1);
        if (t + 1 >= gt_init) {   // feed the reconstruction back through E (model.py:249-258, compute_current_observation :499-543)
            int idx = t + 1, start = idx - S + 1;
            T4 fb;
            if (S == 1) fb = tslice(c->frames[0], B, T - 1, t);
            else {
                fb = c->alloc(B, H, W, 3 * S);
                int j = 0;
                for (int f = idx; f >= (start > gt_init ? start : gt_init); f--, j++)
                    c->copy_op(tslice(c->frames[0], B, T - 1, f - 1), chan(fb, 3 * j, 3));
                if (start < gt_init) {
                    int nch = (gt_init - start) * 3;
                    T4 src = chan(tslice(c->obs, B, T, gt_init - 1), 0, nch);
                    if (!dry) c->ck(pw_copy(dv(src), dv(chan(fb, 3 * j, nch)), 0, c->stream), "gt portion");
                }
            }
            T4 dst = tslice(c->rec_x65, B, T, t + 1);
            c->encode(fb, true, &dst);
        }
        if (c->recording && g.perceptual) c->tape.push_back([c, t]() { c->perc_wait(t); });      // (reverse replay: first thing of this time step's backward)
    }
    c->mark("fwd:closed-loop steps");
    if (c->recording) c->tape.push_back([c]() { c->mark("bwd:A2"); });
    c->action_net(c->rec_x65, c->head2, z.eps_states_rec, z.eps_dirs_rec, nullptr, false, nullptr, nullptr);
    c->q_prob = c->falloc((size_t)B * (T - 1) * g.actions + 256);
    c->fwd_off = c->act.off;
    if (dry && g.perceptual) { T4 gi[3]; VggLevels lv; c->alloc_gt_images(gi, T - 1); vgg_perceptual(c, 1.0, gi, &lv); c->act.off = c->fwd_off; }      // workspace sizing
    c->end_forward();
    c->mark("fwd:A2 + join");
    return finish(c);
}

// ---------------------------------------------------------------------------------------------------------------------
// forward_pretraining (model/main_model/model.py:290-468): D decodes conv3x3(states) for all T frames in one batch, R is
// unrolled on the GROUND-TRUTH states, E re-encodes the (re-stacked) reconstructions in one batch, A runs on both.
// ---------------------------------------------------------------------------------------------------------------------
static int forward_pretraining(caddy_ctx* c, const float* obs, float tau, const caddy_noise* nz, int training,
                               const float* samples_in, const float* variations_in) {
    const caddy_config& g = c->cfg;
    const int B = g.batch, T = g.seq_len, H = g.height, W = g.width, S = g.stacking;
    bool dry = c->dry;
    if (c->gt_prefetched && !dry) hipStreamWaitEvent(c->stream, c->gt_done, 0);
    c->act.reset(); c->tape.clear(); c->tape2.clear(); c->tp = &c->tape; c->d_forked = false; for (BNL* b_ : c->bns) b_->pend.clear(); c->dbg.clear(); c->gfmts.clear(); c->gt_lo = c->gt_hi = 0; c->stats_ring[0] = c->stats_ring[1] = caddy_ctx::TileStats{};
    c->training = training != 0; c->recording = training != 0; c->gt_init = 0; c->tau = tau; c->pretraining = true;
    for (BNL* b : c->bns) b->eval_valid = false;
    for (int i = 0; i < 3; i++) { c->lstm[i].h.d = nullptr; c->lstm[i].c.d = nullptr; }
    c->pack_all();
    caddy_noise z{}; if (nz) z = *nz;
    c->obs = c->alloc(B * T, H, W, 3 * S);
    if (!dry) c->ck(pw_nchw_to_nhwc(obs, (long)3 * S * H * W, dv(c->obs), c->stream), "obs layout");
    if (g.perceptual && training) vgg_gt_prefetch(c, T, 0);
    else c->gt_prefetched = false;
    c->x65_gt = c->encode(c->obs, false, nullptr);
    c->action_net(c->x65_gt, c->head1, z.eps_states, z.eps_dirs, z.gumbel_uniform, true, samples_in, variations_in);
    // state_to_hidden_state_layer (model.py:41-43,413) then D on all B*T frames at once
    Seg ss{chan(c->x65_gt, 0, 64), 0, true};
    c->rec_hidden = c->conv(c->s2h, &ss, 1, 0, nullptr);
    for (int r = 0; r < 3; r++) c->frames[r] = c->alloc(B * T, H >> r, W >> r, 3);
    c->render(c->rec_hidden, 0, 1);
    // R on ground-truth states
    c->hidden = c->alloc(B * (T - 1), c->hs, c->ws, g.hidden);
    T4 aux_all{c->head1.b.aux, c->head1.b.d_aux, B * (T - 1), 1, 1, g.actions + g.action_dim, AUX_LD, AUX_LD};
    if (c->recording) c->tape.push_back([c]() { c->flush_all_wgrad(); });
    for (int t = 0; t < T - 1; t++) {
        T4 hslot = tslice(c->hidden, B, T - 1, t);
        c->dynamics(chan(tslice(c->x65_gt, B, T, t), 0, 64), tslice(aux_all, B, T - 1, t), &hslot);
    }
    // compute_stacked_observations (model.py:470-486): channel group k of time t = reconstruction max(t-k, 0)
    T4 stacked;
    if (S == 1) stacked = c->frames[0];
    else {
        stacked = c->alloc(B * T, H, W, 3 * S);
        const T4& f = c->frames[0];
        for (int k = 0; k < S; k++) {
            int n = T - k;   // times k..T-1 <- reconstructions 0..T-1-k (one tall "image" of n frames per clip)
            if (n > 0) {
                T4 src{f.d, f.g, B, n * H, W, 3, (long)T * f.sn, f.ld};
                T4 dst{stacked.d + (long)k * stacked.sn + 3 * k, stacked.g + (long)k * stacked.sn + 3 * k, B, n * H, W, 3, (long)T * stacked.sn, stacked.ld};
                c->copy_op(src, dst);
            }
            for (int t = 0; t < k && t < T; t++) {   // repeated first reconstruction
                T4 src{f.d, f.g, B, H, W, 3, (long)T * f.sn, f.ld};
                T4 dst{stacked.d + (long)t * stacked.sn + 3 * k, stacked.g + (long)t * stacked.sn + 3 * k, B, H, W, 3, (long)T * stacked.sn, stacked.ld};
                c->copy_op(src, dst);
            }
        }
    }
    c->rec_x65 = c->alloc(B * T, c->hs, c->ws, 65, 68);
    c->encode(stacked, true, &c->rec_x65);
    c->action_net(c->rec_x65, c->head2, z.eps_states_rec, z.eps_dirs_rec, nullptr, false, nullptr, nullptr);
    c->q_prob = c->falloc((size_t)B * (T - 1) * g.actions + 256);
    c->fwd_off = c->act.off;
    if (dry && g.perceptual) { T4 gi[3]; VggLevels lv; c->alloc_gt_images(gi, T); vgg_perceptual(c, 1.0, gi, &lv); c->act.off = c->fwd_off; }
    c->end_forward();
    return finish(c);
}

// end of a forward graph: join the teacher-forced decoder stream, apply the deferred running-statistics updates in call order
void caddy_ctx::end_forward() {
    if (!dry && d_forked) { hipEventRecord(d_done, dstream); hipStreamWaitEvent(stream, d_done, 0); d_forked = false; }
    for (BNL* b : bns) {
        if (!b->deferred || b->pend.empty()) continue;
        std::vector<const float*> m, v;
        for (auto& pr : b->pend) { m.push_back(pr.first); v.push_back(pr.second); }
        if (!dry) ck(pw_bn_ema(m.data(), v.data(), (int)m.size(), b->C, b->rmean, b->rvar, stream), "bn_ema");
        b->pend.clear();
    }
    tape2_done = false;
    have_forward = true;
}

// resized ground-truth frames (N, H >> r, W >> r, 3) pitch 4: input of the VGG19 ground-truth branch, written by loss_l1
void caddy_ctx::alloc_gt_images(T4* gi, int Trec) {
    for (int r = 0; r < 3; r++) {
        const int N = cfg.batch * Trec, H = cfg.height >> r, W = cfg.width >> r;
        float* d = (float*)act.alloc((size_t)N * H * W * 4 * 4);
        gi[r] = T4{d, (float*)((char*)d + grad_delta), N, H, W, 3, (long)H * W * 4, 4, true};
    }
}

static int loss_backward(caddy_ctx* c, const caddy_loss_cfg* lc, double* losses_host) {
    if (!c->have_forward || !c->recording) { set_error("caddy_loss_backward needs a preceding training-mode caddy_forward_full"); return -2; }
    const caddy_config& g = c->cfg;
    const int B = g.batch, T = g.seq_len, K = g.actions, Da = g.action_dim;
    bool dry = c->dry;
    hipStream_t st = c->stream;
    c->mark("bwd:begin");
    const bool perc = c->cfg.perceptual && (lc->perceptual != 0.0 || lc->perceptual_log);
    if (lc->perceptual != 0.0 && !c->cfg.perceptual) { set_error("caddy_loss_cfg.perceptual != 0 needs a context created with caddy_config.perceptual = 1"); return -2; }
    if (perc && !c->vgg.loaded) { set_error("perceptual loss requested but no VGG19 weights were loaded (caddy_load_vgg)"); return -2; }
    c->act.off = c->fwd_off;                  // release what a previous loss_backward allocated past the forward graph
    if (!dry) {
        // gradient mirror of the forward's bottom-up activations (accumulating writers), minus the ground-truth VGG19 taps / scratch in its middle (several GB
        // at BAIR 256x256 that never receive a gradient).  (Zero-filling on the side stream during the forward pass was measured: no gain.)
        char* gm = (char*)c->act.base + c->grad_delta;
        if (c->gt_hi > c->gt_lo && c->gt_hi <= c->act.off) {
            hipMemsetAsync(gm, 0, c->gt_lo, st);
            if (c->act.off > c->gt_hi) hipMemsetAsync(gm + c->gt_hi, 0, c->act.off - c->gt_hi, st);
        } else hipMemsetAsync(gm, 0, c->act.off, st);
        // test aid (caddy_debug_set_poison): NaN-fill the first-touch gradient region so that a read-before-assign cannot go unnoticed
        if (c->poison_nz && c->act.top < c->act.cap) hipMemsetAsync((char*)c->act.base + c->grad_delta + c->act.top, 0xFF, c->act.cap - c->act.top, st);
            hipMemsetAsync(c->G, 0, sizeof(float) * c->n_train, st);
        hipMemsetAsync(c->zero_pool, 0, c->zero_pool_bytes, st);      // every layer's packed weight gradient, the ConvLSTM initial-state gradients, the loss accumulators
    }
    if (!dry) c->ensure_side();
    c->sev_used = 0;
    c->wgrad_det_owner_set = false;
    LossWeights w{lc->rec, lc->states, lc->entropy, lc->dir_kl, lc->mi, lc->state_kl, lc->hidden, lc->mi_entropy_lambda, lc->perceptual};
    double nr[3];
    const int Trec = c->pretraining ? T : T - 1;      // pretraining reconstructs all T frames (losses.py:83-87)
    T4 gt_img[3]{};
    if (perc) c->alloc_gt_images(gt_img, Trec);
    for (int r = 0; r < 3; r++) {
        const T4& f = c->frames[r];
        nr[r] = (double)f.N * 3 * f.H * f.W;
        if (!dry) c->ck(loss_l1(dv(c->obs), dv(f), gv(f), 1 << r, c->pretraining ? 0 : 1, T, Trec, (float)(w.rec / 3.0 / nr[r]), c->loss_acc + LOSS_L1_R0 + r, perc ? gt_img[r].d : nullptr, st), "loss_l1");
    }
    VggLevels lv{};
    // round 6: with several chunks (perc_plan) the VGG19 work runs on the side stream, chunk by chunk from the last time steps, BESIDE the tape replay below
    c->perc_pipelined = perc && lc->perceptual != 0.0 && c->perc_nch > 1 && !c->pretraining && !c->prof && !dry && !caddy_serial_streams() && c->use_side && c->side != nullptr;
    if (c->perc_pipelined) for (int k = 0; k < c->perc_nch; k++) { if (!c->perc_ev[k]) hipEventCreateWithFlags(&c->perc_ev[k], hipEventDisableTiming); c->perc_waited[k] = false; }
    if (perc) vgg_perceptual(c, lc->perceptual, gt_img, &lv);      // VGG19 features of both branches + d(term)/d(rec_r) added to the L1 seeds
    T4 sa = chan(c->x65_gt, 0, 64), sb = chan(c->rec_x65, 0, 64);
    double nst = (double)sa.N * 64 * sa.H * sa.W;
    if (!dry) c->ck(loss_mse(dv(sa), dv(sb), gv(sb), (float)(w.states / nst), c->loss_acc + LOSS_STATES, st), "loss_mse");
    double nhid = 0.0;
    if (c->pretraining && w.hidden != 0.0) {   // HiddenStatesLoss(hidden, rec_hidden[:, 1:].detach()) (trainer.py:313, losses.py:30-53)
        const T4& rh = c->rec_hidden; const T4& hd = c->hidden;
        TV a{rh.d + rh.sn, B, (T - 1) * rh.H, rh.W, rh.C, (long)T * rh.sn, rh.ld};
        TV b{hd.d, B, (T - 1) * hd.H, hd.W, hd.C, (long)(T - 1) * hd.sn, hd.ld};
        TV db{hd.g, B, (T - 1) * hd.H, hd.W, hd.C, (long)(T - 1) * hd.sn, hd.ld};
        nhid = (double)B * (T - 1) * hd.H * hd.W * hd.C;
        if (!dry) c->ck(loss_mse(a, b, db, (float)(w.hidden / nhid), c->loss_acc + LOSS_HIDDEN, st), "hidden loss");
    }
    if (!dry) c->ck(head_softmax(c->head2.b.logits, c->q_prob, nullptr, B * (T - 1), K, st), "softmax");
    SmallLossArgs a{};
    a.K = K; a.Da = Da; a.NS = B * (T - 1); a.NT = B * T;
    a.p = c->head1.b.prob; a.q = c->q_prob; a.logp = c->head1.b.logp;
    a.ddist = c->head1.b.ddist; a.sdist = c->head1.b.sdist; a.sdist_r = c->head2.b.sdist;
    a.d_logits = c->head1.b.d_logits; a.d_logits_r = c->head2.b.d_logits; a.d_ddist = c->head1.b.d_ddist; a.d_sdist_r = c->head2.b.d_sdist;
    a.ema = lc->mi_ema; a.ema_alpha = lc->mi_ema_alpha; a.update_ema = lc->update_mi_ema;
    a.mi_lamb = (float)w.mi_entropy_lambda; a.w_mi = (float)w.mi; a.w_entropy = (float)w.entropy; a.w_dirkl = (float)w.dir_kl; a.w_statekl = (float)w.state_kl;
    a.acc = c->loss_acc;
    a.Pbuf = c->q_prob + (size_t)B * (T - 1) * K;      // K*K floats allocated behind q_prob
    a.mi_grad_scale = (float)(c->hook ? c->world : 1);
    if (!dry) c->ck(loss_small(a, c->hook, c->hook_user, st), "loss_small");
    // (pipelined perceptual pass: its level sums arrive on the side stream -- the totals are formed behind the join at the end)
    auto join_perc = [&]() { if (c->perc_pipelined) { hipEvent_t e = c->sev(); hipEventRecord(e, c->side); hipStreamWaitEvent(st, e, 0); } };
    if (!dry && !c->perc_pipelined) c->ck(loss_finalize(c->loss_acc, w, nr[0], nr[1], nr[2], nst, nhid, perc ? &lv : nullptr, st), "loss_finalize");
    if (lc->diagnostics && !dry) {      // logging-only scalars (trainer.py:475-491): states = output 3, hidden states = output 4 (5 in pretraining): R's hidden states
        DiagArgs dg{c->head1.b.samples, c->head1.b.ddist, c->head2.b.ddist, c->head1.b.variations, c->centroids, B * (T - 1), K, Da, c->loss_acc};
        c->ck(loss_diagnostics(dg, dv(sa), dv(c->hidden), st), "loss_diagnostics");
    }
    if (c->seeds_only) {      // test aid (caddy_debug_set_seeds_only): stop after the loss kernels -- the gradient arena holds d(loss)/d(output) of the direct loss terms only
        if (!dry && c->perc_pipelined) { join_perc(); c->ck(loss_finalize(c->loss_acc, w, nr[0], nr[1], nr[2], nst, nhid, perc ? &lv : nullptr, st), "loss_finalize"); c->perc_pipelined = false; }
        if (!dry && losses_host) { hipMemcpyAsync(losses_host, c->loss_acc, sizeof(double) * LOSS_SLOTS, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }
        return finish(c);
    }
    c->mark("bwd:losses (+ VGG19)");
    // the decoder backward of the teacher-forced steps only needs the loss seeds: start it on its own stream, beside the serial BPTT chain
    c->tape2_done = false;
    // (pipelined perceptual pass: those steps' seeds are in the LAST chunk -- a decoder stream waiting for it would hold up the auxiliary jobs the BPTT chain forks onto that
    //  stream; the steps are replayed at their place in the main tape instead, behind perc_wait(0).  Starting them on the decoder stream once the last chunk has run was measured:
    //  no gain, profiles/r06_experiments.md section 9)
    if (!c->tape2.empty() && c->use_dstream && !dry && !c->perc_pipelined) { c->replay_tape2(true); c->tape2_done = true; }
    for (size_t i = c->tape.size(); i-- > 0;) c->tape[i]();
    c->flush_all_wgrad();
    c->join_aux(st);
    if (!dry && c->use_side && c->side) {      // join: the packed weight gradients must be complete before they are unpacked
        hipEvent_t e = c->sev();
        hipEventRecord(e, c->side);
        hipStreamWaitEvent(st, e, 0);
    }
    if (!dry && c->perc_pipelined) {      // (the join above covered the side stream: every chunk has run)
        c->ck(loss_finalize(c->loss_acc, w, nr[0], nr[1], nr[2], nst, nhid, perc ? &lv : nullptr, st), "loss_finalize");
        c->perc_pipelined = false;
    }
    c->mark("bwd:A1 + E(gt)");
    c->unpack_all();
    c->mark("bwd:join + unpack");
    if (!dry) c->ck(loss_report_flag(c->sat_flag, CADDY_N_FLAGS, c->loss_acc + LOSS_F16_SATURATED, c->loss_acc + LOSS_TOTAL, st), "saturation flag");      // after the VGG19 forward passes of this call, on whatever stream they ran; a NaN among the clamped values poisons the total (the reference would have propagated it)
    if (!dry && losses_host) { hipMemcpyAsync(losses_host, c->loss_acc, sizeof(double) * LOSS_SLOTS, hipMemcpyDeviceToHost, st); if (!lc->no_sync) hipStreamSynchronize(st); }
    return finish(c);
}

// Model.generate_next (model/main_model/model.py:570-607), batch 1, eval mode, persistent ConvLSTM state.
// The per-frame kernel sequence reads the NHWC observation (first allocation of the frame: the same address every frame) and inf_aux and writes the
// full-resolution frame (NHWC), so that it can be captured once and replayed as one graph launch per frame; one boundary kernel in front of it (the caller's
// observation -> NHWC, action -> one-hot) and one behind it (frame -> (3, H, W), obs' = cat[frame, observation[:-3]]) are all the data movement there is.
static void rollout_body(caddy_ctx* c, const T4& o) {
    const caddy_config& g = c->cfg;
    const int H = g.height, W = g.width, K = g.actions, Da = g.action_dim;
    c->tape.clear(); c->tape2.clear(); c->tp = &c->tape; c->training = false; c->recording = false; c->have_forward = false; c->stats_ring[0] = c->stats_ring[1] = caddy_ctx::TileStats{};
    c->fold = c->packed_fold; c->rollout = true;
    T4 x65 = c->encode(o, false, nullptr);
    T4 auxv{c->inf_aux, c->inf_aux, 1, 1, 1, K + Da, AUX_LD, AUX_LD};
    T4 hdn = c->dynamics(chan(x65, 0, 64), auxv, nullptr);
    for (int r = 0; r < 3; r++) c->frames[r] = c->alloc(1, H >> r, W >> r, 3);
    c->render(hdn, 0, 1);
    c->roll_frame = c->frames[0];      // (static: the arena is walked in the same order every frame; the boundary kernel behind the graph reads it)
    c->fold = false; c->rollout = false;
}
void caddy_ctx::drop_graph() {
    if (graph_exec) { hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    if (graph) { hipGraphDestroy(graph); graph = nullptr; }
    graph_valid = false;
}
static int generate_next(caddy_ctx* c, const float* observation, int action, const float* variation, float* frame_out, float* obs_out) {
    const caddy_config& g = c->cfg;
    const int H = g.height, W = g.width, S = g.stacking, K = g.actions, Da = g.action_dim;
    bool dry = c->dry;
    if (action < 0 || action >= K) { set_error("action out of range"); return -2; }
    static const int graph_env = getenv("CADDY_ROLLOUT_GRAPH") ? atoi(getenv("CADDY_ROLLOUT_GRAPH")) : 1;      // A/B aid: 0 eager launches, 2 the graph launched on the internal stream (events to / from the caller's stream every frame)
    static const bool graph_off = graph_env == 0;
    hipStream_t user = c->stream;
    if (c->use_fold && !c->packed_fold) { if (c->graph_exec) { hipStreamSynchronize(user); if (c->gstream) hipStreamSynchronize(c->gstream); } c->drop_graph(); c->prepare_inference_weights(); }      // a forward pass re-packed the plain weights since start_inference
    const bool try_graph = c->use_graph && !graph_off && !c->graph_failed && !dry;
    if (try_graph && !c->gstream) {      // internal stream: the frame's kernel sequence is CAPTURED there (the caller's stream may be the legacy default stream, which cannot be captured)
        if (hipStreamCreateWithFlags(&c->gstream, hipStreamNonBlocking) != hipSuccess) { c->graph_failed = true; c->gstream = nullptr; }
        else { hipEventCreateWithFlags(&c->gev_in, hipEventDisableTiming); hipEventCreateWithFlags(&c->gev_out, hipEventDisableTiming); }
    }
    const bool graphed = try_graph && !c->graph_failed;
    // The graph is LAUNCHED on the caller's stream, between the two boundary kernels: launching it on the internal stream costs an event record + cross-queue wait in
    // each direction per frame -- measured 682 vs 590 us per frame with identical kernels (CADDY_ROLLOUT_GRAPH=2 keeps that form for A/B runs).
    const bool on_internal = graphed && graph_env == 2;
    hipStream_t st = on_internal ? c->gstream : user;
    if (on_internal) { hipEventRecord(c->gev_in, user); hipStreamWaitEvent(st, c->gev_in, 0); }
    c->act.reset();
    T4 o = c->alloc(1, H, W, 3 * S);      // NHWC observation the per-frame kernel sequence reads: first allocation of the frame -> the same address every frame
    if (!dry) c->ck(head_rollout_in(observation, o.d, H * W, 3 * S, o.ld, c->inf_aux, action, variation, K, Da, st), "observation layout + one-hot action + variation");
    if (graphed) {
        if (!c->graph_valid) {      // first frame after start_inference: capture the kernel sequence (the capture itself executes nothing)
            c->stream = c->gstream;
            bool ok = hipStreamBeginCapture(c->gstream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                rollout_body(c, o);
                ok = hipStreamEndCapture(c->gstream, &c->graph) == hipSuccess && c->graph != nullptr && !c->fail;
                if (ok) ok = hipGraphInstantiate(&c->graph_exec, c->graph, nullptr, nullptr, 0) == hipSuccess;
            }
            c->stream = user;
            if (ok) c->graph_valid = true;
            else { c->drop_graph(); c->graph_failed = true; hipGetLastError(); c->fail = false; }
        }
        if (c->graph_valid) { if (hipGraphLaunch(c->graph_exec, st) != hipSuccess) { c->graph_valid = false; c->graph_failed = true; hipGetLastError(); } }
        if (!c->graph_valid) { c->stream = st; rollout_body(c, o); c->stream = user; }      // capture / launch failed: run this frame (and the following ones) eagerly
    } else rollout_body(c, o);
    if (!dry) c->ck(head_rollout_out(c->roll_frame.d, c->roll_frame.ld, observation, frame_out, obs_out, H * W, 3 * S, st), "frame + next observation");
    if (on_internal) { hipEventRecord(c->gev_out, st); hipStreamWaitEvent(user, c->gev_out, 0); }
    return c->fail ? -1 : 0;
}

// eval-mode affine form of every BatchNorm + the packed weights of the roll-out (BatchNorm folded into the preceding conv unless switched off):
// once per start_inference, and again if a forward pass re-packed the plain weights in between
void caddy_ctx::prepare_inference_weights() {
    for (BNL* b : bns) {
        b->eval_valid = false;
        if (!dry) {
            const int cp = round_up(b->C, 4);
            ck(pw_bn_finalize(nullptr, 1, b->gamma, b->beta, b->rmean, b->rvar, b->C, 0, b->eval_stash, b->eval_stash + cp, b->eval_stash + 2 * cp, b->eval_stash + 3 * cp, stream), "bn_finalize");
            b->eval_valid = true;
        }
    }
    pack_all(use_fold);
}
static int start_inference(caddy_ctx* c) {
    bool dry = c->dry;
    c->training = false; c->recording = false;
    if (c->graph_exec && !dry) { hipStreamSynchronize(c->stream); if (c->gstream) hipStreamSynchronize(c->gstream); }  // a graph launch of the previous roll-out may still be executing: never destroy its exec object under it
    c->drop_graph();                                   // weights / state may have changed: re-capture on the next frame
    c->prepare_inference_weights();
    for (int i = 0; i < 3; i++) {
        LstmL& L = c->lstm[i];
        T4 ih = L.ih, ic = L.ic; ih.sn = 0; ic.sn = 0; ih.N = 1; ic.N = 1;
        T4 ph = L.ph, pc = L.pc; ph.N = 1; pc.N = 1;
        if (!dry) { c->ck(pw_copy(dv(ih), dv(ph), 0, c->stream), "init h"); c->ck(pw_copy(dv(ic), dv(pc), 0, c->stream), "init c"); }
        L.h = L.ph; L.c = L.pc; L.h.N = 1; L.c.N = 1;
    }
    return c->fail ? -1 : 0;
}

static int get_output(caddy_ctx* c, int id, void* dst, bool grad) {
    if (!c->have_forward) { set_error("no forward results available"); return -2; }
    const caddy_config& g = c->cfg;
    const int B = g.batch, T = g.seq_len, K = g.actions, Da = g.action_dim;
    hipStream_t st = c->stream;
    auto nchw = [&](const T4& t) { return pw_nhwc_to_nchw(grad ? gv(t) : dv(t), (float*)dst, (long)t.C * t.H * t.W, 0, st); };
    auto raw = [&](const void* p, size_t bytes) { hipMemcpyAsync(dst, grad ? (const char*)p + c->grad_delta : (const char*)p, bytes, hipMemcpyDeviceToDevice, st); return 0; };
    if (grad && (id == 5 || id == 7 || id == 11 || id == 13 || id == 14 || id == 17 || id == 19)) { set_error("no gradient is kept for this output"); return -2; }
    const HeadBufs& a = c->head1.b; const HeadBufs& r = c->head2.b;
    const size_t NS = (size_t)B * (T - 1), NT = (size_t)B * T;
    if (c->pretraining) {   // tuple order of forward_pretraining (model.py:463-468): 4 = rec. hidden states, 5.. = full-model 4..8 shifted by one
        if (id == 4) return nchw(c->rec_hidden);
        if (id >= 5 && id <= 9) id -= 1;
    }
    switch (id) {
        case 0: case 100: return nchw(c->frames[0]);
        case 101: return nchw(c->frames[1]);
        case 102: return nchw(c->frames[2]);
        case 2: return nchw(chan(c->rec_x65, 0, 64));
        case 3: return nchw(chan(c->x65_gt, 0, 64));
        case 4: return nchw(c->hidden);
        case 5: return raw(a.selected, NS * sizeof(long long));
        case 6: return raw(a.logits, NS * K * 4);
        case 7: return raw(a.samples, NS * K * 4);
        case 8: return nchw(c->head1.att);
        case 9: {   // attention of the reconstructed states without the first (ground-truth) one
            T4 t = c->head2.att;
            TV v{(grad ? t.g : t.d) + t.sn, B, (T - 1) * t.H, t.W, 1, (long)T * t.sn, t.ld};
            return pw_nhwc_to_nchw(v, (float*)dst, (long)(T - 1) * t.H * t.W, 0, st);
        }
        case 10: return raw(a.ddist, NS * 2 * Da * 4);
        case 11: return raw(a.dirs, NS * Da * 4);
        case 12: return raw(a.sdist, NT * 2 * Da * 4);
        case 13: return raw(a.ssamp, NT * Da * 4);
        case 14: return raw(a.variations, NS * Da * 4);
        case 15: return raw(r.logits, NS * K * 4);
        case 16: return raw(r.ddist, NS * 2 * Da * 4);
        case 17: return raw(r.dirs, NS * Da * 4);
        case 18: return raw(r.sdist, NT * 2 * Da * 4);
        case 19: return raw(r.ssamp, NT * Da * 4);
    }
    set_error("unknown output id"); return -2;
}

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
static bool check_cfg(const caddy_config* g) {
    if (!g || g->batch < 1 || g->seq_len < 2 || g->height % 16 || g->width % 16 || g->height < 16 || g->width < 16 || g->stacking < 1 ||
        g->actions < 1 || g->actions > 16 || g->action_dim < 1 || g->action_dim > 8 || g->actions + g->action_dim > AUX_LD ||
        (g->variant != 0 && g->variant != 1) || g->hidden != (g->variant == 0 ? 128 : 64) || g->ensemble < 0 || g->ensemble > CADDY_MAX_ENSEMBLE) {
        set_error("invalid caddy_config (H, W multiples of 16; hidden 128 for main / 64 for reduced; K<=16; Da<=8; ensemble <= 8)");
        return false;
    }
    if (g->perceptual && (g->height < 64 || g->width < 64)) {      // the quarter-resolution image must survive four 2x2 max-pools (relu5_1)
        set_error("caddy_config.perceptual needs frames of at least 64x64 (VGG19 relu5_1 at a quarter of the resolution)");
        return false;
    }
    return true;
}
static caddy_ctx* make_ctx(const caddy_config* cfg, float* params, float* grads, void* ws, size_t act_cap) {
    caddy_ctx* c = new caddy_ctx();
    c->cfg = *cfg; c->dry = ws == nullptr; c->P = params; c->G = grads;
    build_param_table(*cfg, c->table, &c->n_floats, &c->n_train);
    c->persist.base = (char*)ws; c->persist.cap = (size_t)-1;
    build_layers(c);
    size_t pbytes = (c->persist.high + 4095) & ~(size_t)4095;
    c->act.base = (char*)ws + pbytes; c->act.cap = act_cap; c->grad_delta = act_cap;
    c->act.reset();
    return c;
}

extern "C" {
int caddy_param_count(const caddy_config* cfg) {
    if (!check_cfg(cfg)) return -1;
    std::vector<ParamEntry> t; long a, b; build_param_table(*cfg, t, &a, &b); return (int)t.size();
}
int caddy_param_info_get(const caddy_config* cfg, int i, caddy_param_info* out) {
    if (!check_cfg(cfg)) return -1;
    std::vector<ParamEntry> t; long a, b; build_param_table(*cfg, t, &a, &b);
    if (i < 0 || i >= (int)t.size()) { set_error("index"); return -1; }
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", t[i].name.c_str());
    out->offset = t[i].offset; out->ndim = t[i].ndim; out->kind = t[i].kind;
    for (int k = 0; k < 4; k++) out->shape[k] = t[i].shape[k];
    return 0;
}
long caddy_param_floats(const caddy_config* cfg) { if (!check_cfg(cfg)) return -1; std::vector<ParamEntry> t; long a, b; build_param_table(*cfg, t, &a, &b); return a; }
long caddy_trainable_floats(const caddy_config* cfg) { if (!check_cfg(cfg)) return -1; std::vector<ParamEntry> t; long a, b; build_param_table(*cfg, t, &a, &b); return b; }

static void dry_sizes(const caddy_config* cfg, size_t* persist, size_t* act) {
    caddy_ctx* c = make_ctx(cfg, nullptr, nullptr, nullptr, (size_t)1 << 50);
    forward_full(c, nullptr, 1, 1.f, nullptr, 1, nullptr, nullptr);
    forward_pretraining(c, nullptr, 1.f, nullptr, 1, nullptr, nullptr);     // Arena::high keeps the maximum of both graphs
    *persist = (c->persist.high + 4095) & ~(size_t)4095;
    *act = ((c->act.high + 4095) & ~(size_t)4095) + ((c->act.top_used + 4095) & ~(size_t)4095) + 4096;     // bottom-up part + top-down (first-touch) part
    delete c;
}
size_t caddy_workspace_bytes(const caddy_config* cfg) {
    if (!check_cfg(cfg)) return 0;
    size_t p, a; dry_sizes(cfg, &p, &a);
    return p + 2 * a + 4096;
}
caddy_ctx* caddy_ctx_create(const caddy_config* cfg, float* params, float* grads, void* workspace, size_t workspace_bytes) {
    if (!check_cfg(cfg)) return nullptr;
    if (!params || !grads || !workspace) { set_error("null buffer"); return nullptr; }
    size_t p, a; dry_sizes(cfg, &p, &a);
    if (workspace_bytes < p + 2 * a) { set_error("workspace too small (see caddy_workspace_bytes)"); return nullptr; }
    if (((uintptr_t)workspace & 255) || ((uintptr_t)params & 15) || ((uintptr_t)grads & 15)) { set_error("buffers must be 256-byte (workspace) / 16-byte (params, grads) aligned"); return nullptr; }
    caddy_ctx* c = make_ctx(cfg, params, grads, workspace, a);
    if (c->fail) { delete c; return nullptr; }
    if (caddy_serial_streams()) c->use_dstream = false;
    hipMemset(c->sat_flag, 0, sizeof(unsigned) * 2 * CADDY_N_FLAGS);      // (second half: sticky until polled, caddy_f16_saturated)
    if (const char* e = getenv("CADDY_DETERMINISTIC")) c->deterministic = atoi(e) != 0;      // profiling aid: the bit-reproducible backward without touching the caller (tools/gpu_serial_breakdown.sh)
    if (const char* e = getenv("CADDY_PERC_CHUNKS")) { const int n = atoi(e); if (n < 0) c->perc_chunks_force = -n; else c->perc_chunks_cfg = n; }      // (negative: that many chunks whatever the size of the step)      // A/B aid: 1 = the one-pass perceptual loss of rounds 2 - 5
    if (const char* e = getenv("CADDY_MASK_FROM_X")) c->mask_from_x = atoi(e) != 0;      // A/B aid: 0 = BatchNorm backward reads the materialised output for the LeakyReLU slope (round-5 form)
    if (const char* e = getenv("CADDY_S16_GRADS")) c->s16_grads = atoi(e) != 0;      // A/B aid: 0 = every model gradient as fp32 (round-5 form)
    if (const char* e = getenv("CADDY_VGG_S16")) c->vgg_s16 = atoi(e) != 0;      // A/B aid: 0 = every VGG19 feature map as fp32 (round-4 form)
    if (const char* e = getenv("CADDY_PRECISION")) {      // A/B + parity aid: "exact" = every convolution on the exact-fp32 MFMA path
        if (!strcmp(e, "exact") || !strcmp(e, "0")) { c->prec_fwd = c->prec_bwd = PREC_FP32; c->vgg_precision = c->vgg_precision_bwd = PREC_FP32; }
        else if (!strcmp(e, "fwd")) { c->prec_bwd = PREC_FP32; c->vgg_precision_bwd = PREC_FP32; }
    }
    return c;
}
extern "C" int caddy_dp_shutdown(caddy_ctx* c);
void caddy_ctx_destroy(caddy_ctx* c) {
    if (c && (c->comm || c->comm2)) caddy_dp_shutdown(c);
    if (c && c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); }
    if (c) for (int k = 0; k < caddy_ctx::PERC_MAX_CHUNKS; k++) if (c->perc_ev[k]) hipEventDestroy(c->perc_ev[k]);
    if (c && c->gstream) { if (c->graph_exec) hipStreamSynchronize(c->stream); hipStreamSynchronize(c->gstream); c->drop_graph(); hipStreamDestroy(c->gstream); }
    if (c) for (ConvL* L : c->convs) if (L->off_ev) hipEventDestroy(L->off_ev);
    if (c && c->dstream) { hipStreamSynchronize(c->dstream); hipStreamDestroy(c->dstream); if (c->d_done) hipEventDestroy(c->d_done); }
    delete c;
}
int caddy_set_stream(caddy_ctx* c, void* s) { c->stream = (hipStream_t)s; return 0; }
int caddy_debug_set_poison(caddy_ctx* c, int on) { c->poison_nz = on != 0; return 0; }
int caddy_debug_set_bn_paths(caddy_ctx* c, int small, int lazy, int epilogue_stats) { c->bn_small = small != 0; c->lazy_bn = lazy != 0; c->epi_stats = epilogue_stats != 0; return 0; }
int caddy_debug_set_vgg_s16(caddy_ctx* c, int on) { c->vgg_s16 = on != 0; return 0; }
int caddy_debug_set_s16_grads(caddy_ctx* c, int on) { c->s16_grads = on != 0; return 0; }
int caddy_debug_set_perc_chunks(caddy_ctx* c, int n) { c->perc_chunks_force = n; return 0; }
long caddy_debug_s16_grad_count(caddy_ctx* c) { long n = 0; for (const GradFmt& g : c->gfmts) n += g.fmt ? 1 : 0; return n; }
int caddy_debug_set_pack_merged(caddy_ctx* c, int on) { c->merged_pack = on != 0; c->pack_jobs.key = -1; for (auto& j : c->unpack_jobs) j.key = -1; return 0; }
int caddy_debug_set_seeds_only(caddy_ctx* c, int on) { c->seeds_only = on != 0; return 0; }
int caddy_set_grads_ready_hook(caddy_ctx* c, caddy_grads_ready_hook hook, void* user) { c->grads_hook = hook; c->grads_user = user; return 0; }
int caddy_set_sampler_hook(caddy_ctx* c, caddy_sampler_hook hook, void* user, int provides_samples, int provides_variations) {
    c->samplers = SamplerHooks{};
    if (hook && (provides_samples || provides_variations)) { c->samplers.fn = hook; c->samplers.user = user; c->samplers.action = provides_samples; c->samplers.variation = provides_variations; }
    return 0;
}
int caddy_set_allreduce_hook(caddy_ctx* c, void (*hook)(float*, int, void*), void* user, int world_size) {
    c->hook = hook; c->hook_user = user; c->world = world_size > 1 ? world_size : 1;
    return 0;
}
int caddy_forward_full(caddy_ctx* c, const float* obs, int gt_init, float tau, const caddy_noise* noise, int training, const float* samples_in, const float* variations_in) {
    c->fail = false;
    if (!obs || !noise) { set_error("null input"); return -2; }
    return forward_full(c, obs, gt_init, tau, noise, training, samples_in, variations_in);
}
int caddy_forward_pretraining(caddy_ctx* c, const float* obs, float tau, const caddy_noise* noise, int training, const float* samples_in, const float* variations_in) {
    c->fail = false;
    if (!obs || !noise) { set_error("null input"); return -2; }
    return forward_pretraining(c, obs, tau, noise, training, samples_in, variations_in);
}
int caddy_get_output(caddy_ctx* c, int id, void* dst) { return get_output(c, id, dst, false); }
int caddy_get_output_grad(caddy_ctx* c, int id, void* dst) { return get_output(c, id, dst, true); }
int caddy_loss_backward(caddy_ctx* c, const caddy_loss_cfg* cfg, double* losses_host) { c->fail = false; return loss_backward(c, cfg, losses_host); }
int caddy_set_action_member(caddy_ctx* c, int member) {
    if (member < 0 || member >= c->n_members) { set_error("caddy_set_action_member: member out of range"); return -2; }
    c->member = member; return 0;
}
int caddy_adam_step_ex(caddy_ctx* c, float* m, float* v, float lr, float b1, float b2, float eps, float wd, int step, const int* member_step, int s2h_step, float gscale) {
    // torch.optim.Adam keeps `step` per parameter and skips parameters whose .grad is None -- no moment update, no weight decay.  Which parameters those are depends on
    // optimizer.zero_grad(): set_to_none (torch >= 2.0 default) leaves every parameter that the last backward did not reach without a gradient; the zero-filling form of
    // torch < 2.0 (the reference pins pytorch 1.4.0, env.yml) keeps a ZERO gradient on every parameter that has had one before, so Adam goes on ageing its moments and applying
    // the weight decay.  The caller decides (trainer.py: training.zero_grad_semantics) and passes the bookkeeping: a count of 0 skips the range, a count > 0 steps it with that
    // bias-correction count -- with its real gradient if the last pass produced one (the drawn member; state_to_hidden_state_layer after forward_pretraining,
    // model.py:41-43,413), with g = 0 otherwise.
    struct Rng { long lo, hi; int count; float gs; };
    std::vector<Rng> special;
    if (c->s2h_hi > c->s2h_lo) special.push_back({c->s2h_lo, c->s2h_hi, s2h_step, c->pretraining ? gscale : 0.f});
    if (c->n_members > 1) for (int k = 0; k < c->n_members; k++) if (c->member_hi[k] > c->member_lo[k]) special.push_back({c->member_lo[k], c->member_hi[k], member_step ? member_step[k] : (k == c->member ? step : 0), k == c->member ? gscale : 0.f});
    std::sort(special.begin(), special.end(), [](const Rng& a, const Rng& b) { return a.lo < b.lo; });
    int rc = 0;
    long pos = 0;
    auto run = [&](long lo, long hi, int st, float gs) { if (!rc && hi > lo) rc = adam_launch(c->P + lo, c->G + lo, m + lo, v + lo, hi - lo, lr, b1, b2, eps, wd, st, gs, c->stream); };
    for (const Rng& r : special) { run(pos, r.lo, step, gscale); if (r.count > 0) run(r.lo, r.hi, r.count, r.gs); pos = r.hi; }
    run(pos, c->n_train, step, gscale);
    return rc;
}
int caddy_adam_step_member(caddy_ctx* c, float* m, float* v, float lr, float b1, float b2, float eps, float wd, int step, int member_step, float gscale) {
    // the set_to_none bookkeeping (torch >= 2.0): only the drawn member is stepped, with its own count; state_to_hidden_state_layer only after a pretraining pass
    int ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c->member >= 0 && c->member < 8) ms[c->member] = member_step;
    return caddy_adam_step_ex(c, m, v, lr, b1, b2, eps, wd, step, ms, c->pretraining ? step : 0, gscale);
}
int caddy_adam_step(caddy_ctx* c, float* m, float* v, float lr, float b1, float b2, float eps, float wd, int step, float gscale) {
    return caddy_adam_step_member(c, m, v, lr, b1, b2, eps, wd, step, step, gscale);
}
int caddy_vgg_param_count(void) { return vgg_param_count(); }
int caddy_vgg_param_info_get(int index, caddy_param_info* out) { return vgg_param_info(index, out); }
long caddy_vgg_param_floats(void) { return vgg_param_floats(); }
int caddy_load_vgg(caddy_ctx* c, const float* vgg_flat) {
    c->fail = false;
    if (!vgg_flat) { set_error("null input"); return -2; }
    return vgg_load(c, vgg_flat);
}
int caddy_set_perceptual_prefetch(caddy_ctx* c, int on) { c->perc_prefetch = on != 0; return 0; }
// evaluation: per-position reconstruction / state losses of the last forward (evaluation/evaluator.py:192,194).  l1_host: B * Trec means of |frame - ground truth| (full
// resolution, first 3 channels of the observation); mse_host: B * T means of (reconstructed state - state)^2.  Either may be NULL.
int caddy_sequence_losses_per_frame(caddy_ctx* c, double* l1_host, double* mse_host) {
    c->fail = false;
    if (!c->have_forward) { set_error("caddy_sequence_losses_per_frame: no forward results available"); return -2; }
    const caddy_config& g = c->cfg;
    const int B = g.batch, T = g.seq_len, Trec = c->pretraining ? T : T - 1, t_off = c->pretraining ? 0 : 1;
    hipStream_t st = c->stream;
    c->act.off = c->fwd_off;
    double* acc = c->dalloc((size_t)B * (Trec + T));
    if (c->act.overflow()) { c->act.off = c->fwd_off; set_error("caddy_sequence_losses_per_frame: workspace too small"); return -1; }
    hipMemsetAsync(acc, 0, sizeof(double) * (size_t)B * (Trec + T), st);
    const T4& f = c->frames[0];
    c->ck(loss_diff_per_frame(dv(c->obs), T, t_off, dv(f), Trec, 3, 0, acc, st), "per-frame L1");
    T4 sa = c->x65_gt, sb = c->rec_x65;
    c->ck(loss_diff_per_frame(dv(sa), T, 0, dv(sb), T, 64, 1, acc + (size_t)B * Trec, st), "per-frame state MSE");
    std::vector<double> host((size_t)B * (Trec + T));
    hipMemcpyAsync(host.data(), acc, sizeof(double) * host.size(), hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    c->act.off = c->fwd_off;
    const double nf = 3.0 * f.H * f.W, ns = 64.0 * sa.H * sa.W;
    if (l1_host) for (int i = 0; i < B * Trec; i++) l1_host[i] = host[i] / nf;
    if (mse_host) for (int i = 0; i < B * T; i++) mse_host[i] = host[(size_t)B * Trec + i] / ns;
    return finish(c);
}
int caddy_perceptual_per_frame(caddy_ctx* c, double* out_host) { c->fail = false; if (!out_host) { set_error("null output"); return -2; } return vgg_eval_per_frame(c, out_host); }
int caddy_set_rollout_fold(caddy_ctx* c, int on) {
    if (c->graph_exec && !c->dry) { hipStreamSynchronize(c->stream); if (c->gstream) hipStreamSynchronize(c->gstream); }      // a launch of the graph may still be executing: never destroy its exec object under it
    c->use_fold = on != 0; c->drop_graph(); c->packed_fold = false; return 0;
}
int caddy_set_vgg_precision(caddy_ctx* c, int forward, int dgrad) { c->vgg_precision = forward; c->vgg_precision_bwd = dgrad; return 0; }
int caddy_set_deterministic(caddy_ctx* c, int on) { c->deterministic = on != 0; return 0; }
int caddy_set_precision(caddy_ctx* c, int forward, int backward) {
    if ((forward != PREC_FP32 && forward != PREC_F16X3) || (backward != PREC_FP32 && backward != PREC_BF16X3)) { set_error("caddy_set_precision: forward 0 | 16, backward 0 | 17"); return -2; }
    c->prec_fwd = forward; c->prec_bwd = backward; return 0;
}
int caddy_start_inference(caddy_ctx* c) { c->fail = false; return start_inference(c); }
// Poll of the per-layer f16 range guards: waits for the stream, reads and clears the flag words.  Returns bit 0: a split-f16 forward convolution (model or VGG19) staged |x| > 65504
// since the last poll (it was clamped); bit 1: a NaN was among them.  The layers that reported -- and only those -- run without a range limit from the next forward on (exact fp32 /
// split bf16 for VGG19); caddy_fallback_layers counts them.
int caddy_f16_saturated(caddy_ctx* c) {
    if (c->dry) return 0;
    unsigned v[2 * CADDY_N_FLAGS];      // words raised since the last loss call + the sticky ones k_report_flag moved there
    if (c->side) hipStreamSynchronize(c->side);      // (VGG19 levels / ground-truth branch)
    hipMemcpyAsync(v, c->sat_flag, sizeof(v), hipMemcpyDeviceToHost, c->stream);
    hipStreamSynchronize(c->stream);
    unsigned any = 0;
    for (int i = 0; i < CADDY_N_FLAGS; i++) { v[i] |= v[CADDY_N_FLAGS + i]; if (v[i]) { any |= v[i]; if (!c->layer_fallback[i]) { c->layer_fallback[i] = true; c->n_fallback++; } } }
    if (any) {
        hipMemsetAsync(c->sat_flag, 0, sizeof(v), c->stream);
        if (c->graph_exec) { hipStreamSynchronize(c->stream); if (c->gstream) hipStreamSynchronize(c->gstream); }
        c->drop_graph();      // (a captured roll-out frame still holds the split-f16 launches of the layers that just moved)
    }
    return (int)(any & 3u);
}
int caddy_fallback_layers(caddy_ctx* c) { return c->n_fallback; }
int caddy_generate_next(caddy_ctx* c, const float* observation, int action, const float* variation, float* frame_out, float* obs_out) {
    c->fail = false;
    if (!observation || !frame_out) { set_error("null input"); return -2; }
    {   // the boundary kernel behind the frame reads `observation` while it writes obs_out = cat[frame, observation[:-3]] and frame_out: the buffers must not overlap
        const size_t ob = sizeof(float) * 3 * (size_t)c->cfg.stacking * c->cfg.height * c->cfg.width, fb = sizeof(float) * 3 * (size_t)c->cfg.height * c->cfg.width;
        auto overlap = [](const void* a, size_t na, const void* b, size_t nb) { return (const char*)a < (const char*)b + nb && (const char*)b < (const char*)a + na; };
        if ((obs_out && overlap(observation, ob, obs_out, ob)) || overlap(observation, ob, frame_out, fb) || (obs_out && overlap(obs_out, ob, frame_out, fb))) {
            set_error("caddy_generate_next: observation, frame_out and obs_out must not overlap (in-place update of the stacked observation is not supported)"); return -2; }
    }
    if (c->lstm[0].h.d != c->lstm[0].ph.d || c->lstm[0].h.d == nullptr) { set_error("call caddy_start_inference first"); return -2; }
    return generate_next(c, observation, action, variation, frame_out, obs_out);
}
int caddy_profile_begin(caddy_ctx* c) { c->prof = true; c->prof_recs.clear(); c->phases.clear(); c->ev_used = 0; return 0; }
// phase marks of the profiled steps: names_out receives `max` x 48-byte names, ms_out the time since the previous mark on the main stream; returns the count
int caddy_profile_phases(caddy_ctx* c, char* names_out, float* ms_out, int max) {
    hipStreamSynchronize(c->stream);
    int n = 0;
    for (size_t i = 0; i < c->phases.size() && n < max; i++, n++) {
        float ms = 0.f;
        if (i > 0) hipEventElapsedTime(&ms, c->phases[i - 1].e, c->phases[i].e);
        snprintf(names_out + 48 * n, 48, "%s", c->phases[i].name);
        ms_out[n] = ms;
    }
    return n;
}
int caddy_profile_records(caddy_ctx* c, double* out, int max_records) {   // per launch: {kind, P, K, Cout, KS, flops, ms}; returns count
    hipStreamSynchronize(c->stream);
    if (c->side) hipStreamSynchronize(c->side);
    int n = 0;
    for (auto& r : c->prof_recs) {
        if (n >= max_records) break;
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.a, r.b);
        double* o = out + 7 * n++;
        o[0] = r.kind; o[1] = r.P; o[2] = r.K; o[3] = r.Cout; o[4] = r.KS; o[5] = r.flops; o[6] = ms;
    }
    return n;
}
static_assert(CK_COUNT == CADDY_PROFILE_FAMILIES, "caddy_hip.h: CADDY_PROFILE_FAMILIES");
int caddy_profile_end(caddy_ctx* c, double* out18) {   // CK_COUNT kernel families x (launches, algorithmic FLOPs, milliseconds, algorithmic bytes)
    hipStreamSynchronize(c->stream);
    if (c->side) hipStreamSynchronize(c->side);
    for (int i = 0; i < 4 * CK_COUNT; i++) out18[i] = 0.0;
    for (auto& r : c->prof_recs) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.a, r.b);
        if (r.fam < 0 || r.fam >= CK_COUNT) continue;
        out18[r.fam * 4] += 1.0; out18[r.fam * 4 + 1] += r.flops; out18[r.fam * 4 + 2] += ms; out18[r.fam * 4 + 3] += r.bytes;
    }
    c->prof = false; c->prof_recs.clear();
    return 0;
}
int caddy_debug_fusion_counts(caddy_ctx* c, long* out3) { out3[0] = c->n_bn_calls; out3[1] = c->n_bn_tile_stats; out3[2] = c->n_bn_lazy; return 0; }
int caddy_debug_count(caddy_ctx* c) { return (int)c->dbg.size(); }
int caddy_debug_dims(caddy_ctx* c, int i, int* nhwc4) {
    if (i < 0 || i >= (int)c->dbg.size()) return -1;
    nhwc4[0] = c->dbg[i].N; nhwc4[1] = c->dbg[i].H; nhwc4[2] = c->dbg[i].W; nhwc4[3] = c->dbg[i].C; return 0;
}
int caddy_debug_get(caddy_ctx* c, int i, int grad, float* dst_nchw) {
    if (i < 0 || i >= (int)c->dbg.size()) return -1;
    const T4& t = c->dbg[i];
    return pw_nhwc_to_nchw(grad ? gv(t) : dv(t), dst_nchw, (long)t.C * t.H * t.W, 0, c->stream);
}
int caddy_bn_layer_count(caddy_ctx* c) { return (int)c->bns.size(); }
long caddy_bn_calls(caddy_ctx* c, int i, char* name_out128) {
    if (i < 0 || i >= (int)c->bns.size()) return -1;
    if (name_out128) snprintf(name_out128, 128, "%s", c->bns[i]->name.c_str());
    return c->bns[i]->calls;
}
}
