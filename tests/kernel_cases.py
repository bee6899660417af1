"""Per-kernel parity cases shared by the simulator tests (CPU, `-m "not gpu"`) and the GPU tests (`-m gpu`).

Every case takes (lib, dev): `lib` is a ctypes handle exposing the C-ABI kernel entry points, `dev` the torch device the
buffers live on ("cpu" for the simulator, "cuda" on the MI355X).  References are plain torch fp32 ops on CPU.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from playablevideogeneration_amd._lib import TV, ConvArgs, ConvSrc, PackDesc, WgradArgs, round_up, CONV_BK


def nhwc(x_nchw, ld=None, dev="cpu"):
    """NCHW cpu tensor -> (buffer (N,H,W,ld) on dev, with channels [C,ld) poisoned by NaN-free garbage)."""
    N, Cc, H, W = x_nchw.shape
    ld = ld or round_up(Cc, 4)
    buf = torch.full((N, H, W, ld), 7.5)
    buf[..., :Cc] = x_nchw.permute(0, 2, 3, 1)
    return buf.contiguous().to(dev)


def tv(buf, Cc, n_stride=None, ptr_off=0):
    N, H, W, ld = buf.shape
    t = TV(buf.data_ptr() + 4 * ptr_off, N, H, W, Cc, n_stride if n_stride is not None else H * W * ld, ld)
    t._keep = buf          # keep the storage alive for as long as the view object lives
    return t


def to_nchw(buf, Cc):
    return buf[..., :Cc].permute(0, 3, 1, 2).contiguous().cpu()


def stream(dev):
    return C.c_void_p(torch.cuda.current_stream().cuda_stream if dev != "cpu" else 0)


def sync(dev):
    if dev != "cpu":
        torch.cuda.synchronize()


def P(t):
    return C.c_void_p(t.data_ptr())


def make_pack(ws, segs, KS, lib):
    """ws: list of OIHW cpu tensors (stacked along O); segs: list of (ref channel offset, C)."""
    d = PackDesc()
    d.nw, d.Co_each, d.Cin, d.KS = len(ws), ws[0].shape[0], ws[0].shape[1], KS
    d.nseg = len(segs)
    kt = 0
    for i, (off, c) in enumerate(segs):
        d.seg_off[i], d.seg_C[i], d.seg_Cpad[i] = off, c, round_up(c, CONV_BK)
        kt += round_up(c, CONV_BK)
    d.Cout = d.nw * d.Co_each
    d.Cout_pad = round_up(d.Cout, lib.caddy_k_conv_pick_bn(d.Cout))
    d.Ktot = kt
    return d


def conv_case(lib, dev, *, N, H, W, segs, Cout, KS, nw=1, bias=False, act=0, seed=0, check_bwd=True, tol=2e-5, precision=0, use_aux=True, wgrad_precision=0, wgrad_tol=None,
              res=False, oscale=False, dgrad_precision=None, dgrad_tol=None, avgpool=False, direct_ok=False):
    """segs: list of (C, bcast).  Checks forward, dgrad (per spatial segment), wgrad against torch autograd.
    dgrad_precision = 17: ConvArgs.precision of the dgrad launches (split bf16: conv_head.hip's 7x7 head dgrad).
    avgpool: ConvArgs.avgpool -- the launch writes avg_pool2d(conv, 2) with bias / residual / activation applied after the pooling (forward only)."""
    g = torch.Generator().manual_seed(seed)
    Cin = sum(c for c, _ in segs)
    xs = []
    for c, bc in segs:
        xs.append(torch.randn(N, c, generator=g) if bc else torch.randn(N, c, H, W, generator=g))
    ws = [(torch.randn(Cout // nw, Cin, KS, KS, generator=g) / (Cin * KS * KS) ** 0.5) for _ in range(nw)]
    b = torch.randn(Cout, generator=g) if bias else None
    # reference
    xs_r = [x.clone().requires_grad_(True) for x in xs]
    ws_r = [w.clone().requires_grad_(True) for w in ws]
    full = torch.cat([x[:, :, None, None].expand(-1, -1, H, W) if bc else x for x, (c, bc) in zip(xs_r, segs)], dim=1)
    y_ref = F.conv2d(full, torch.cat(ws_r, 0), b, padding=KS // 2)
    osc = (torch.rand(Cout, generator=g) + 0.5) if oscale else None       # PackDesc.oscale: per-output-channel factor folded into the packed weights
    if oscale:
        y_ref = (y_ref - (b[None, :, None, None] if bias else 0)) * osc[None, :, None, None] + (b[None, :, None, None] if bias else 0)
    OH, OW = (H // 2, W // 2) if avgpool else (H, W)
    if avgpool:
        assert not check_bwd
        y_ref = F.avg_pool2d(y_ref, 2)
    r_in = torch.randn(N, Cout, OH, OW, generator=g) if res else None        # ConvArgs.res: residual input added before the activation
    if res:
        y_ref = y_ref + r_in
    if act == 1:
        y_ref = torch.tanh(y_ref)
    elif act == 3:
        y_ref = F.leaky_relu(y_ref, 0.2)
    dy = torch.randn(y_ref.shape, generator=g)
    # device buffers
    st = stream(dev)
    ws_d = [w.contiguous().to(dev) for w in ws]
    gws_d = [torch.zeros_like(w) for w in ws_d]
    offs, o = [], 0
    for c, _ in segs:
        offs.append((o, c)); o += c
    d = make_pack(ws, offs, KS, lib)
    for i in range(nw):
        d.w[i], d.gw[i] = ws_d[i].data_ptr(), gws_d[i].data_ptr()
    taps = KS * KS
    wp = torch.full((taps * d.Cout_pad * d.Ktot,), 3.0, device=dev)
    osc_d = osc.to(dev) if oscale else None
    d.oscale = osc_d.data_ptr() if oscale else None
    assert lib.caddy_k_pack_fwd(C.byref(d), P(wp), st) == 0
    a = ConvArgs()
    bufs = []
    for i, ((c, bc), x) in enumerate(zip(segs, xs)):
        if bc:
            bb = torch.full((N, 16), 7.5); bb[:, :c] = x; bb = bb.to(dev)
            a.src[i] = ConvSrc(bb.data_ptr(), 16, 16, c, round_up(c, CONV_BK), 1)
        else:
            bb = nhwc(x, dev=dev)
            a.src[i] = ConvSrc(bb.data_ptr(), H * W * bb.shape[3], bb.shape[3], c, round_up(c, CONV_BK), 0)
        bufs.append(bb)
    a.nsrc, a.N, a.H, a.W, a.KS = len(segs), N, H, W, KS
    a.wp, a.Ktot, a.Cout, a.Cout_pad = wp.data_ptr(), d.Ktot, Cout, d.Cout_pad
    b_d = b.to(dev) if bias else None
    a.bias = b_d.data_ptr() if bias else None
    a.act = act
    a.precision = precision
    aux = torch.zeros(128 * 1024 // 4, device=dev)     # CONV_AUX_BYTES scratch (compact weight tables of the thin kernels)
    a.aux = aux.data_ptr() if use_aux else None
    split = torch.zeros(9 * N * H * W * round_up(Cout, 4), device=dev)     # slabs of the deterministic forward split-K
    a.split_scratch, a.split_cap = (split.data_ptr(), split.numel()) if use_aux else (None, 0)
    out_ld = round_up(Cout, 4) + 4
    out = torch.full((N, OH, OW, out_ld), 9.0, device=dev)
    a.out, a.out_sn, a.out_ld, a.accumulate = out.data_ptr(), OH * OW * out_ld, out_ld, 0
    if res:
        r_d = nhwc(r_in, ld=round_up(Cout, 4) + 8, dev=dev)
        a.res, a.res_sn, a.res_ld = r_d.data_ptr(), OH * OW * r_d.shape[3], r_d.shape[3]
    a.direct_ok = 1 if direct_ok else 0          # inference launch: the latency kernels of conv_direct.hip may take it
    if avgpool:
        assert lib.caddy_k_conv_avgpool_ok(C.byref(a)) == 1
        a.avgpool = 1
    assert lib.caddy_k_conv_fwd(C.byref(a), st) == 0
    sync(dev)
    y = to_nchw(out, Cout)
    err = (y - y_ref.detach()).abs().max().item()
    assert err < tol, ("fwd", err)
    assert torch.all(out[..., Cout:] == 9.0)            # pad channels untouched
    if not check_bwd:
        return
    y_ref.backward(dy)
    # the kernels receive the gradient w.r.t. the pre-activation (tanh' is a separate element-wise kernel)
    dz = dy * (1 - y_ref.detach() ** 2) if act == 1 else dy
    dz_d = nhwc(dz, dev=dev)
    # wgrad
    dwp = torch.zeros_like(wp)
    wa = WgradArgs()
    for i in range(len(segs)):
        wa.src[i] = a.src[i]
    wa.nsrc, wa.N, wa.H, wa.W, wa.KS = a.nsrc, N, H, W, KS
    wa.dy, wa.dy_sn, wa.dy_ld = dz_d.data_ptr(), H * W * dz_d.shape[3], dz_d.shape[3]
    wa.Cout, wa.Cout_pad, wa.Ktot, wa.dwp, wa.slabs = Cout, d.Cout_pad, d.Ktot, dwp.data_ptr(), 0
    wa.precision = wgrad_precision          # 17: k_wgrad_hx (split bf16 on the 16-bit matrix pipe)
    assert lib.caddy_k_conv_wgrad(C.byref(wa), st) == 0
    assert lib.caddy_k_unpack_wgrad(C.byref(d), P(dwp), st) == 0
    sync(dev)
    for i in range(nw):
        e = (gws_d[i].cpu() - ws_r[i].grad).abs().max().item()
        assert e < (wgrad_tol or tol) * max(1.0, ws_r[i].grad.abs().max().item()) * 4, ("wgrad", i, e)
    # bit-reproducible mode (WgradArgs.det_slab: one zero-filled copy of the packed layout per pixel split + fixed-order reduce): same dW (up to the summation order), and two
    # launches give IDENTICAL bits -- whichever kernel the launcher picked for this shape
    det = torch.zeros(1 << 22, device=dev)
    dws = []
    for _ in range(2):
        dwp_d = torch.zeros_like(wp)
        wa.dwp, wa.det_slab, wa.det_cap = dwp_d.data_ptr(), det.data_ptr(), det.numel()
        assert lib.caddy_k_conv_wgrad(C.byref(wa), st) == 0
        sync(dev)
        dws.append(dwp_d)
    wa.dwp, wa.det_slab, wa.det_cap = dwp.data_ptr(), None, 0
    e = (dws[0] - dwp).abs().max().item()
    assert e < (wgrad_tol or tol) * max(1.0, dwp.abs().max().item()) * 4, ("deterministic wgrad vs atomics", e)
    assert torch.equal(dws[0], dws[1]), "deterministic wgrad not bit-reproducible"
    # time-batched addressing (WgradArgs.group_n): the same batch laid out as 2 groups with a padded group stride must give the same dW
    if N % 2 == 0:
        gn = N // 2
        wg = WgradArgs()
        keep = []
        for i, (c, bc) in enumerate(segs):
            sn = a.src[i].sn
            flat = bufs[i].reshape(-1)
            gs = gn * sn + 64
            buf = torch.full((2 * gs,), 5.0, device=dev)
            buf[:gn * sn] = flat[:gn * sn]; buf[gs:gs + gn * sn] = flat[gn * sn:2 * gn * sn]
            keep.append(buf)
            wg.src[i] = ConvSrc(buf.data_ptr(), sn, a.src[i].ld, a.src[i].C, a.src[i].Cpad, a.src[i].bcast)
            wg.src_gs[i] = gs
        dsn = H * W * dz_d.shape[3]
        dgs = gn * dsn + 128
        dbuf = torch.full((2 * dgs,), 5.0, device=dev)
        dflat = dz_d.reshape(-1)
        dbuf[:gn * dsn] = dflat[:gn * dsn]; dbuf[dgs:dgs + gn * dsn] = dflat[gn * dsn:]
        dwp2 = torch.zeros_like(wp)
        wg.nsrc, wg.N, wg.H, wg.W, wg.KS = a.nsrc, N, H, W, KS
        wg.dy, wg.dy_sn, wg.dy_ld, wg.dy_gs, wg.group_n = dbuf.data_ptr(), dsn, dz_d.shape[3], dgs, gn
        wg.Cout, wg.Cout_pad, wg.Ktot, wg.dwp, wg.slabs = Cout, d.Cout_pad, d.Ktot, dwp2.data_ptr(), 0
        assert lib.caddy_k_conv_wgrad(C.byref(wg), st) == 0
        sync(dev)
        e = (dwp2 - dwp).abs().max().item()
        assert e < tol * max(1.0, dwp.abs().max().item()) * 4, ("grouped wgrad", e)
    # dgrad per segment = conv of dz with flipped/transposed weights
    for si, ((c, bc), x_r) in enumerate(zip(segs, xs_r)):
        bn = lib.caddy_k_conv_pick_bn(c)
        cd_pad, kd = round_up(c, bn), round_up(Cout, CONV_BK)
        wpd = torch.full((taps * cd_pad * kd,), 3.0, device=dev)
        assert lib.caddy_k_pack_dgrad(C.byref(d), si, P(wpd), cd_pad, kd, st) == 0
        da = ConvArgs()
        da.src[0] = ConvSrc(dz_d.data_ptr(), H * W * dz_d.shape[3], dz_d.shape[3], Cout, kd, 0)
        da.nsrc, da.N, da.H, da.W, da.KS = 1, N, H, W, KS
        da.wp, da.Ktot, da.Cout, da.Cout_pad, da.bias, da.act = wpd.data_ptr(), kd, c, cd_pad, None, 0
        da.precision = precision if dgrad_precision is None else dgrad_precision
        da.aux = aux.data_ptr() if use_aux else None
        gx = torch.ones((N, H, W, round_up(c, 4)), device=dev)      # accumulate on top of ones
        da.out, da.out_sn, da.out_ld, da.accumulate = gx.data_ptr(), H * W * gx.shape[3], gx.shape[3], 1
        assert lib.caddy_k_conv_fwd(C.byref(da), st) == 0
        sync(dev)
        got = to_nchw(gx, c) - 1.0
        if bc:
            got = got.sum(dim=(2, 3))
        e = (got - x_r.grad).abs().max().item()
        assert e < (dgrad_tol or tol) * 8 * max(1.0, x_r.grad.abs().max().item()), ("dgrad", si, e)
        # deterministic split-K of the accumulating launch (ConvArgs.deterministic: slabs + fixed-order reduce that adds the old contents)
        scr = torch.zeros(9 * N * H * W * round_up(c, 4), device=dev)
        outs = []
        for _ in range(2):
            gx2 = torch.ones((N, H, W, round_up(c, 4)), device=dev)
            da.out, da.deterministic, da.split_scratch, da.split_cap = gx2.data_ptr(), 1, scr.data_ptr(), scr.numel()
            assert lib.caddy_k_conv_fwd(C.byref(da), st) == 0
            sync(dev)
            outs.append(gx2)
        da.deterministic, da.split_scratch, da.split_cap = 0, None, 0
        got2 = to_nchw(outs[0], c) - 1.0
        if bc:
            got2 = got2.sum(dim=(2, 3))
        e = (got2 - x_r.grad).abs().max().item()
        assert e < (dgrad_tol or tol) * 8 * max(1.0, x_r.grad.abs().max().item()), ("deterministic dgrad", si, e)
        assert torch.equal(outs[0], outs[1]), "deterministic dgrad not bit-reproducible"


# ------------------------------------------------------------------------------------------------------------------
# element-wise / reduction kernels
# ------------------------------------------------------------------------------------------------------------------
def _rand(g, *shape):
    return torch.randn(*shape, generator=g)


def pool_up_case(lib, dev, N=2, Cc=6, H=4, W=6, seed=0):
    g = torch.Generator().manual_seed(seed)
    st = stream(dev)
    x = _rand(g, N, Cc, 2 * H, 2 * W).requires_grad_(True)
    y_ref = F.avg_pool2d(x, 2)
    dy = _rand(g, *y_ref.shape)
    y_ref.backward(dy)
    xb, yb = nhwc(x.detach(), dev=dev), torch.zeros(N, H, W, round_up(Cc, 4), device=dev)
    assert lib.caddy_k_pool2(C.byref(tv(xb, Cc)), C.byref(tv(yb, Cc)), st) == 0
    gx = torch.ones_like(xb)
    assert lib.caddy_k_pool2_bwd(C.byref(tv(nhwc(dy, dev=dev), Cc)), C.byref(tv(gx, Cc)), st) == 0
    sync(dev)
    assert (to_nchw(yb, Cc) - y_ref.detach()).abs().max() < 1e-6
    assert (to_nchw(gx, Cc) - 1 - x.grad).abs().max() < 1e-6
    gx = torch.full_like(xb, float("nan"))                      # first-touch variant: the gradient buffer holds garbage and is assigned
    assert lib.caddy_k_pool2_bwd_assign(C.byref(tv(nhwc(dy, dev=dev), Cc)), C.byref(tv(gx, Cc)), st) == 0
    sync(dev)
    assert (to_nchw(gx, Cc) - x.grad).abs().max() < 1e-6
    # bilinear x2
    x = _rand(g, N, Cc, H, W).requires_grad_(True)
    y_ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    dy = _rand(g, *y_ref.shape)
    y_ref.backward(dy)
    xb, yb = nhwc(x.detach(), dev=dev), torch.zeros(N, 2 * H, 2 * W, round_up(Cc, 4), device=dev)
    assert lib.caddy_k_up2(C.byref(tv(xb, Cc)), C.byref(tv(yb, Cc)), st) == 0
    gx = torch.ones_like(xb)
    assert lib.caddy_k_up2_bwd(C.byref(tv(nhwc(dy, dev=dev), Cc)), C.byref(tv(gx, Cc)), st) == 0
    sync(dev)
    assert (to_nchw(yb, Cc) - y_ref.detach()).abs().max() < 1e-6
    assert (to_nchw(gx, Cc) - 1 - x.grad).abs().max() < 2e-6


def bn_case(lib, dev, N=3, Cc=10, H=5, W=4, second="bn", act=1, training=1, seed=0, fused=False):
    """out = lrelu(BN(x) + second) with second in {None, 'plain', 'bn'}; forward, running stats, backward.
    fused=True: the one-launch small-map kernels (k_bn_small_fwd / k_bn_small_bwd) instead of the generic multi-launch path."""
    g = torch.Generator().manual_seed(seed)
    st = stream(dev)
    x = (_rand(g, N, Cc, H, W) * 2 + 0.5).requires_grad_(True)
    x2 = _rand(g, N, Cc, H, W).requires_grad_(True)
    gam, bet = (1 + 0.1 * _rand(g, Cc)).requires_grad_(True), (0.1 * _rand(g, Cc)).requires_grad_(True)
    gam2, bet2 = (1 + 0.1 * _rand(g, Cc)).requires_grad_(True), (0.1 * _rand(g, Cc)).requires_grad_(True)
    rm, rv, rm2, rv2 = 0.1 * _rand(g, Cc), 1 + 0.2 * torch.rand(Cc, generator=g), 0.1 * _rand(g, Cc), 1 + 0.2 * torch.rand(Cc, generator=g)
    rm_r, rv_r, rm2_r, rv2_r = rm.clone(), rv.clone(), rm2.clone(), rv2.clone()
    y = F.batch_norm(x, rm_r, rv_r, gam, bet, bool(training), 0.1, 1e-5)
    if second == "bn":
        y = y + F.batch_norm(x2, rm2_r, rv2_r, gam2, bet2, bool(training), 0.1, 1e-5)
    elif second == "plain":
        y = y + x2
    if act:
        y = F.leaky_relu(y, 0.2)
    dy = _rand(g, *y.shape)
    y.backward(dy)
    M = N * H * W

    def run_bn(xt, gm, bt, rmean, rvar):
        xb = nhwc(xt.detach(), dev=dev)
        sums = torch.zeros(2 * Cc, dtype=torch.float64, device=dev)
        if training and seed % 2 == 0:
            assert lib.caddy_k_stats(C.byref(tv(xb, Cc)), P(sums), st) == 0
        o = [torch.zeros(Cc, device=dev) for _ in range(4)]
        gmd, btd, rmd, rvd = gm.detach().to(dev), bt.detach().to(dev), rmean.to(dev), rvar.to(dev)
        if training and seed % 2 == 1:          # the driver's path: per-block partials + fold with the finalisation fused (no atomics)
            scratch = torch.zeros(512 * 2 * Cc, dtype=torch.float64, device=dev)
            assert lib.caddy_k_bn_stats_finalize(C.byref(tv(xb, Cc)), P(sums), P(scratch), P(gmd), P(btd), P(rmd), P(rvd), *[P(t) for t in o], st) == 0
            return xb, o, gmd, rmd, rvd
        assert lib.caddy_k_bn_finalize(P(sums), C.c_long(M), P(gmd), P(btd), P(rmd), P(rvd), Cc, training, *[P(t) for t in o], st) == 0
        return xb, o, gmd, rmd, rvd

    if fused:
        assert training and second != "bn"
        xb = nhwc(x.detach(), dev=dev)
        mean, invstd, scale, shift = [torch.zeros(Cc, device=dev) for _ in range(4)]
        gmd, btd, rmd, rvd = gam.detach().to(dev), bet.detach().to(dev), rm.to(dev), rv.to(dev)
        x2b = nhwc(x2.detach(), dev=dev) if second == "plain" else None
        out = torch.zeros_like(xb)
        x2tv = tv(x2b, Cc) if x2b is not None else None
        assert lib.caddy_k_bn_small_fwd(C.byref(tv(xb, Cc)), P(gmd), P(btd), P(rmd), P(rvd), P(mean), P(invstd), P(scale), P(shift),
                                        C.byref(x2tv) if x2tv else None, act, C.byref(tv(out, Cc)), st) == 0
        sync(dev)
        assert (to_nchw(out, Cc) - y.detach()).abs().max() < 2e-6
        assert (rmd.cpu() - rm_r).abs().max() < 1e-6 and (rvd.cpu() - rv_r).abs().max() < 1e-6
        dyb = nhwc(dy, dev=dev)
        dx, dres = torch.ones_like(xb), torch.ones_like(xb)
        dg, db = torch.ones(Cc, device=dev), torch.ones(Cc, device=dev)
        dtv = tv(dres, Cc)
        assert lib.caddy_k_bn_small_bwd(C.byref(tv(dyb, Cc)), C.byref(tv(out, Cc)) if act else None, C.byref(tv(xb, Cc)), P(mean), P(invstd), P(gmd),
                                        C.byref(tv(dx, Cc)), P(dg), P(db), C.byref(dtv) if second == "plain" else None, st) == 0
        sync(dev)
        assert (to_nchw(dx, Cc) - 1 - x.grad).abs().max() < 5e-6 * max(1.0, x.grad.abs().max().item())
        assert (dg.cpu() - 1 - gam.grad).abs().max() < 2e-5 * max(1.0, gam.grad.abs().max().item())
        assert (db.cpu() - 1 - bet.grad).abs().max() < 2e-5 * max(1.0, bet.grad.abs().max().item())
        if second == "plain":
            assert (to_nchw(dres, Cc) - 1 - x2.grad).abs().max() < 2e-6
        dx2 = torch.full_like(xb, float("nan")); dres2 = torch.ones_like(xb); dt2 = tv(dres2, Cc)
        dg2, db2 = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
        assert lib.caddy_k_bn_small_bwd_assign(C.byref(tv(dyb, Cc)), C.byref(tv(out, Cc)) if act else None, C.byref(tv(xb, Cc)), P(mean), P(invstd), P(gmd),
                                               C.byref(tv(dx2, Cc)), P(dg2), P(db2), C.byref(dt2) if second == "plain" else None, st) == 0
        sync(dev)
        assert (to_nchw(dx2, Cc) - x.grad).abs().max() < 5e-6 * max(1.0, x.grad.abs().max().item())
        return
    xb, (mean, invstd, scale, shift), gmd, rmd, rvd = run_bn(x, gam, bet, rm, rv)
    x2b = None
    if second == "bn":
        x2b, (mean2, invstd2, scale2, shift2), gmd2, rmd2, rvd2 = run_bn(x2, gam2, bet2, rm2, rv2)
    elif second == "plain":
        x2b = nhwc(x2.detach(), dev=dev)
    out = torch.zeros_like(xb)
    x2tv = tv(x2b, Cc) if x2b is not None else None
    assert lib.caddy_k_bn_apply(C.byref(tv(xb, Cc)), P(scale), P(shift), C.byref(x2tv) if x2tv else None,
                                P(scale2) if second == "bn" else None, P(shift2) if second == "bn" else None, act, C.byref(tv(out, Cc)), st) == 0
    sync(dev)
    assert (to_nchw(out, Cc) - y.detach()).abs().max() < 2e-6
    if training:
        assert (rmd.cpu() - rm_r).abs().max() < 1e-6 and (rvd.cpu() - rv_r).abs().max() < 1e-6
    else:
        return
    # backward
    dyb = nhwc(dy, dev=dev)
    outm = C.byref(tv(out, Cc)) if act else None

    def bwd(xbuf, mean_, invstd_, gmd_, xref, gref, bref):
        sums = torch.zeros(2 * Cc, dtype=torch.float64, device=dev)
        assert lib.caddy_k_bn_bwd_reduce(C.byref(tv(dyb, Cc)), outm, C.byref(tv(xbuf, Cc)), P(mean_), P(invstd_), P(sums), st) == 0
        dx = torch.ones_like(xbuf)
        dg, db = torch.ones(Cc, device=dev), torch.ones(Cc, device=dev)
        assert lib.caddy_k_bn_bwd_apply(C.byref(tv(dyb, Cc)), outm, C.byref(tv(xbuf, Cc)), P(mean_), P(invstd_), P(gmd_), P(sums),
                                        C.byref(tv(dx, Cc)), P(dg), P(db), st) == 0
        sync(dev)
        assert (to_nchw(dx, Cc) - 1 - xref.grad).abs().max() < 5e-6 * max(1.0, xref.grad.abs().max().item())
        assert (dg.cpu() - 1 - gref.grad).abs().max() < 2e-5 * max(1.0, gref.grad.abs().max().item())
        assert (db.cpu() - 1 - bref.grad).abs().max() < 2e-5 * max(1.0, bref.grad.abs().max().item())
        dx2 = torch.full_like(xbuf, float("nan"))               # first-touch variant (dx assigned, not accumulated)
        assert lib.caddy_k_bn_bwd_apply_assign(C.byref(tv(dyb, Cc)), outm, C.byref(tv(xbuf, Cc)), P(mean_), P(invstd_), P(gmd_), P(sums), C.byref(tv(dx2, Cc)), st) == 0
        sync(dev)
        assert (to_nchw(dx2, Cc) - xref.grad).abs().max() < 5e-6 * max(1.0, xref.grad.abs().max().item())

    bwd(xb, mean, invstd, gmd, x, gam, bet)
    if second == "bn":
        bwd(x2b, mean2, invstd2, gmd2, x2, gam2, bet2)
    elif second == "plain":
        dres = torch.ones_like(x2b)
        if act:
            assert lib.caddy_k_act_bwd_add(C.byref(tv(dyb, Cc)), C.byref(tv(out, Cc)), C.byref(tv(dres, Cc)), st) == 0
        else:
            assert lib.caddy_k_copy(C.byref(tv(dyb, Cc)), C.byref(tv(dres, Cc)), 1, st) == 0
        sync(dev)
        assert (to_nchw(dres, Cc) - 1 - x2.grad).abs().max() < 2e-6


def lstm_case(lib, dev, N=2, Cc=12, H=3, W=5, seed=0):
    g = torch.Generator().manual_seed(seed)
    st = stream(dev)
    pre = _rand(g, N, 4 * Cc, H, W).requires_grad_(True)
    cp = _rand(g, N, Cc, H, W).requires_grad_(True)
    i, f, o, gg = pre[:, :Cc].sigmoid(), pre[:, Cc:2 * Cc].sigmoid(), pre[:, 2 * Cc:3 * Cc].sigmoid(), pre[:, 3 * Cc:].tanh()
    c = f * cp + i * gg
    h = o * torch.tanh(c)
    dh, dc = _rand(g, *h.shape), _rand(g, *c.shape)
    (h * dh).sum().backward(retain_graph=True)
    (c * dc).sum().backward()
    gates = nhwc(pre.detach(), dev=dev)
    cpb = nhwc(cp.detach(), dev=dev)
    hb, cb = torch.zeros_like(cpb), torch.zeros_like(cpb)
    assert lib.caddy_k_lstm_fwd(C.byref(tv(gates, 4 * Cc)), C.byref(tv(cpb, Cc)), C.byref(tv(hb, Cc)), C.byref(tv(cb, Cc)), st) == 0
    dgates, dcp = torch.zeros_like(gates), torch.ones_like(cpb)
    assert lib.caddy_k_lstm_bwd(C.byref(tv(gates, 4 * Cc)), C.byref(tv(cpb, Cc)), C.byref(tv(cb, Cc)), C.byref(tv(nhwc(dh, dev=dev), Cc)),
                                C.byref(tv(nhwc(dc, dev=dev), Cc)), C.byref(tv(dgates, 4 * Cc)), C.byref(tv(dcp, Cc)), st) == 0
    sync(dev)
    assert (to_nchw(hb, Cc) - h.detach()).abs().max() < 2e-6 and (to_nchw(cb, Cc) - c.detach()).abs().max() < 2e-6
    assert (to_nchw(dgates, 4 * Cc) - pre.grad).abs().max() < 5e-6
    assert (to_nchw(dcp, Cc) - 1 - cp.grad).abs().max() < 5e-6


def misc_case(lib, dev, N=3, H=4, W=5, seed=0):
    g = torch.Generator().manual_seed(seed)
    st = stream(dev)
    # attention gate
    x = _rand(g, N, 65, H, W).requires_grad_(True)
    att = torch.sigmoid(x[:, 64:])
    out = x[:, :64] * att
    dout, datt = _rand(g, *out.shape), _rand(g, *att.shape)
    (out * dout).sum().backward(retain_graph=True)
    (att * datt).sum().backward()
    xb = nhwc(x.detach(), ld=68, dev=dev)
    ob, ab = torch.zeros(N, H, W, 64, device=dev), torch.zeros(N, H, W, 4, device=dev)
    assert lib.caddy_k_attn_mul(C.byref(tv(xb, 65)), C.byref(tv(ob, 64)), C.byref(tv(ab, 1)), st) == 0
    dx = torch.ones_like(xb)
    assert lib.caddy_k_attn_mul_bwd(C.byref(tv(xb, 65)), C.byref(tv(nhwc(dout, dev=dev), 64)), C.byref(tv(nhwc(datt, dev=dev), 1)), C.byref(tv(dx, 65)), st) == 0
    sync(dev)
    assert (to_nchw(ob, 64) - out.detach()).abs().max() < 2e-6 and (to_nchw(ab, 1) - att.detach()).abs().max() < 2e-6
    assert (to_nchw(dx, 65) - 1 - x.grad).abs().max() < 1e-5
    # GAP fwd/bwd, colsum, spatial sum, tanh bwd
    x = _rand(g, N, 10, H, W)
    xb = nhwc(x, dev=dev)
    gap = torch.zeros(N, 10, device=dev)
    assert lib.caddy_k_gap(C.byref(tv(xb, 10)), P(gap), st) == 0
    dgap = _rand(g, N, 10).to(dev)
    dx = torch.ones_like(xb)
    assert lib.caddy_k_gap_bwd(P(dgap), C.byref(tv(dx, 10)), st) == 0
    cs = torch.ones(10, device=dev)
    assert lib.caddy_k_colsum(C.byref(tv(xb, 10)), P(cs), st) == 0
    ss = torch.ones(N, 16, device=dev)
    assert lib.caddy_k_spatial_sum(C.byref(tv(xb, 10)), P(ss), C.c_long(16), st) == 0
    y = torch.tanh(_rand(g, N, 3, H, W)); dy = _rand(g, N, 3, H, W)
    dz = torch.zeros(N, H, W, 4, device=dev)
    assert lib.caddy_k_tanh_bwd(C.byref(tv(nhwc(dy, dev=dev), 3)), C.byref(tv(nhwc(y, dev=dev), 3)), C.byref(tv(dz, 3)), st) == 0
    sync(dev)
    assert (gap.cpu() - x.mean(dim=(2, 3))).abs().max() < 2e-6
    assert (to_nchw(dx, 10) - 1 - (dgap.cpu() / (H * W))[:, :, None, None]).abs().max() < 2e-6
    assert (cs.cpu() - 1 - x.sum(dim=(0, 2, 3))).abs().max() < 2e-5
    assert (ss.cpu()[:, :10] - 1 - x.sum(dim=(2, 3))).abs().max() < 2e-5 and torch.all(ss.cpu()[:, 10:] == 1)
    assert (to_nchw(dz, 3) - dy * (1 - y * y)).abs().max() < 2e-6
    # layout conversion with strides (time-slice of a (B,T,C,H,W) tensor) + batch sum
    B, T, Cc = 2, 3, 6
    src = _rand(g, B, T, Cc, H, W).to(dev)
    d = torch.zeros(B, H, W, 8, device=dev)
    assert lib.caddy_k_nchw_to_nhwc(C.c_void_p(src.data_ptr() + 4 * Cc * H * W), C.c_long(T * Cc * H * W), C.byref(tv(d, Cc)), st) == 0
    back = torch.zeros(B, T, Cc, H, W, device=dev)
    assert lib.caddy_k_nhwc_to_nchw(C.byref(tv(d, Cc)), C.c_void_p(back.data_ptr() + 4 * 2 * Cc * H * W), C.c_long(T * Cc * H * W), 0, st) == 0
    bs = torch.ones(Cc * H * W, device=dev)
    assert lib.caddy_k_batch_sum(P(src), C.c_long(T * Cc * H * W), C.c_long(Cc * H * W), B, P(bs), st) == 0
    sync(dev)
    assert torch.equal(to_nchw(d, Cc), src[:, 1].cpu()) and torch.all(d[..., Cc:] == 0)
    assert torch.equal(back[:, 2].cpu(), src[:, 1].cpu())
    assert (bs.cpu() - 1 - src[:, 0].cpu().sum(0).flatten()).abs().max() < 1e-5
    # strided copy into a channel slice of a wider buffer
    wide = torch.zeros(B, H, W, 16, device=dev)
    s6 = nhwc(src[:, 0].cpu(), dev=dev)
    assert lib.caddy_k_copy(C.byref(tv(s6, Cc)), C.byref(tv(wide, Cc, ptr_off=4)), 0, st) == 0
    sync(dev)
    assert torch.equal(wide[..., 4:4 + Cc].cpu(), s6[..., :Cc].cpu()) and torch.all(wide[..., :4] == 0) and torch.all(wide[..., 4 + Cc:] == 0)


def adam_case(lib, dev, n=1000, seed=0):
    g = torch.Generator().manual_seed(seed)
    p = _rand(g, n); gr = _rand(g, n)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=4e-4, weight_decay=1e-6)
    pd, m, v = p.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step in (1, 2, 3):
        pr.grad = gr.clone() * step
        opt.step()
        gd = (gr * step).to(dev)
        assert lib.caddy_k_adam(P(pd), P(gd), P(m), P(v), C.c_long(n), C.c_float(4e-4), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8),
                                C.c_float(1e-6), step, C.c_float(1.0), stream(dev)) == 0
    sync(dev)
    assert (pd.cpu() - pr.detach()).abs().max() < 1e-6


def conv_fuzz(lib, dev, n, seed):
    import random
    rng = random.Random(seed)
    for i in range(n):
        ks = rng.choice([1, 3, 3, 3, 7])
        nseg = 1 if ks == 7 else rng.choice([1, 1, 2, 3])
        segs = []
        for s in range(nseg):
            c = rng.choice([1, 3, 4, 5, 12, 16, 17, 31, 32, 33, 48, 64, 65, 70])
            bc = 1 if (nseg > 1 and s == 1 and ks == 3) else 0
            segs.append((rng.choice([1, 5, 9, 12]) if bc else c, bc))             # broadcast inputs are <= 16-wide vectors (AUX_LD)
        if ks == 7:
            segs = [(rng.choice([3, 16, 32]), 0)]
        kw = dict(N=rng.choice([1, 2, 2, 4]), H=rng.choice([1, 2, 3, 5, 8, 9, 13, 16, 33]), W=rng.choice([1, 4, 7, 8, 10, 17, 32, 40]), segs=segs,
                  Cout=rng.choice([1, 3, 4, 9, 16, 24, 32, 33, 64, 72, 128, 130]) if ks != 7 else rng.choice([3, 16, 32]), KS=ks,
                  bias=rng.random() < 0.3, act=1 if rng.random() < 0.15 else 0, nw=4 if (nseg == 3 and rng.random() < 0.5) else 1)
        if kw["nw"] == 4 and kw["Cout"] % 4:
            kw["Cout"] = 64
        try:
            conv_case(lib, dev, **kw)
        except AssertionError as e:
            raise AssertionError((i, kw, e))


PREC_F16X3, PREC_BF16X3, PREC_F16X1, PREC_BF16X1 = 16, 17, 18, 19


def hx_conv_case(lib, dev, *, N, H, W, segs, Cout, precision=PREC_F16X3, bias=False, act=0, mask=False, seed_w=0.0, accumulate=False, dgrad_seg=None,
                 split=False, seed=0, tol=None, big=-1, res=False, oscale=False, pool=False, skip_out=False, direct=False, avgpool=False):
    """conv_hx.hip: 3x3 convolution on the 16-bit MFMA with split operands, through caddy_k_pack_hx + caddy_k_conv_fwd.
    direct = True: ConvArgs.direct_ok -- small assigning launches go to the latency kernel (conv_direct.hip); the case asserts that it was taken (the untouched pad columns of
    the output and, with `split`, the untouched slab scratch tell).
    avgpool = True (with direct): ConvArgs.avgpool -- the launch writes avg_pool2d(conv, 2) with bias / activation after the pooling (conv_direct.hip's 2 x 8-pixel groups).
    dgrad_seg = s: the dgrad form (input = dY with Cout channels, output = gradient of input segment s), reference = torch autograd.
    Reference: torch fp64 conv2d of the fp32 inputs (so that the split-f16 error itself is measured: tol ~ a few 1e-7 relative)."""
    lib.caddy_k_hx_weight_bytes.restype = C.c_long
    g = torch.Generator().manual_seed(seed)
    Cin = sum(c for c, _ in segs)
    xs = [torch.randn(N, c, generator=g) if bc else torch.randn(N, c, H, W, generator=g) for c, bc in segs]
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    st = stream(dev)
    w_d = w.contiguous().to(dev)
    offs, o = [], 0
    for c, _ in segs:
        offs.append((o, c)); o += c
    d = make_pack([w], offs, 3, lib)
    d.w[0] = w_d.data_ptr()
    planes = 2 if precision in (PREC_F16X3, PREC_BF16X3) else 1
    full = torch.cat([x[:, :, None, None].expand(-1, -1, H, W) if bc else x for x, (c, bc) in zip(xs, segs)], dim=1).double()
    a = ConvArgs()
    keep = []
    if dgrad_seg is None:
        rows_pad = round_up(Cout, lib.caddy_k_hx_pick_bn(Cout))
        wq = torch.zeros(lib.caddy_k_hx_weight_bytes(C.byref(d), -1, rows_pad, planes), dtype=torch.uint8, device=dev)
        osc = (torch.rand(Cout, generator=g) + 0.5) if oscale else None
        osc_d = osc.to(dev) if oscale else None
        d.oscale = osc_d.data_ptr() if oscale else None
        assert lib.caddy_k_pack_hx(C.byref(d), P(wq), rows_pad, -1, precision, st) == 0
        for i, ((c, bc), x) in enumerate(zip(segs, xs)):
            if bc:
                bb = torch.full((N, 16), 7.5); bb[:, :c] = x; bb = bb.to(dev)
                a.src[i] = ConvSrc(bb.data_ptr(), 16, 16, c, round_up(c, CONV_BK), 1)
            else:
                bb = nhwc(x, dev=dev)
                a.src[i] = ConvSrc(bb.data_ptr(), H * W * bb.shape[3], bb.shape[3], c, round_up(c, CONV_BK), 0)
            keep.append(bb)
        a.nsrc, out_c = len(segs), Cout
        ref = F.conv2d(full, (w * osc[:, None, None, None]).double() if oscale else w.double(), b.double() if bias else None, padding=1)
    else:
        c_s = segs[dgrad_seg][0]
        rows_pad = round_up(c_s, lib.caddy_k_hx_pick_bn(c_s))
        wq = torch.zeros(lib.caddy_k_hx_weight_bytes(C.byref(d), dgrad_seg, rows_pad, planes), dtype=torch.uint8, device=dev)
        assert lib.caddy_k_pack_hx(C.byref(d), P(wq), rows_pad, dgrad_seg, precision, st) == 0
        dy = torch.randn(N, Cout, H, W, generator=g) * 1e-6          # gradient-sized values: far below the f16 range, fine for split bf16
        bb = nhwc(dy, dev=dev)
        a.src[0] = ConvSrc(bb.data_ptr(), H * W * bb.shape[3], bb.shape[3], Cout, round_up(Cout, CONV_BK), 0)
        keep.append(bb)
        a.nsrc, out_c = 1, c_s
        fr = full.clone().requires_grad_(True)
        F.conv2d(fr, w.double(), None, padding=1).backward(dy.double())
        ref = fr.grad[:, offs[dgrad_seg][0]:offs[dgrad_seg][0] + c_s]
    a.N, a.H, a.W, a.KS = N, H, W, 3
    a.wp, a.Ktot, a.Cout, a.Cout_pad = None, sum(round_up(a.src[i].C, CONV_BK) for i in range(a.nsrc)), out_c, round_up(out_c, lib.caddy_k_conv_pick_bn(out_c))
    a.wq, a.precision = wq.data_ptr(), precision
    b_d = b.to(dev) if bias else None
    a.bias, a.act = (b_d.data_ptr() if bias else None), act
    OH, OW = H, W
    if avgpool:
        assert direct and dgrad_seg is None and not (mask or accumulate or pool)
        ref = F.avg_pool2d(ref, 2)
        OH, OW = H // 2, W // 2
    if res:
        r_in = torch.randn(N, out_c, OH, OW, generator=g)
        r_d = nhwc(r_in, ld=round_up(out_c, 4) + 8, dev=dev)
        a.res, a.res_sn, a.res_ld = r_d.data_ptr(), OH * OW * r_d.shape[3], r_d.shape[3]
        ref = ref + r_in.double()
    if act == 1:
        ref = torch.tanh(ref)
    elif act == 2:
        ref = torch.relu(ref)
    elif act == 3:
        ref = F.leaky_relu(ref, 0.2)
    out_ld = round_up(out_c, 4) + 4
    init = torch.randn(N, OH, OW, out_ld, generator=g)
    out = init.clone().to(dev)
    if mask:
        mk = torch.randn(N, H, W, out_ld, generator=g)
        sr = torch.randn(N, H, W, out_ld, generator=g)
        mk_d, sr_d = mk.to(dev), sr.to(dev)
        a.mask = mk_d.data_ptr()
        if seed_w:
            a.seed_ref, a.seed_w = sr_d.data_ptr(), seed_w
        m_, s_ = mk[..., :out_c].permute(0, 3, 1, 2).double(), sr[..., :out_c].permute(0, 3, 1, 2).double()
        if seed_w:
            ref = ref + seed_w * torch.sign(m_ - s_)
        ref = torch.where(m_ > 0, ref, torch.zeros_like(ref))
    if accumulate:
        a.accumulate = 1
        ref = ref + init[..., :out_c].permute(0, 3, 1, 2).double()
    if split:
        scr = torch.zeros(8 * N * H * W * round_up(out_c, 4), device=dev)
        a.split_scratch, a.split_cap = scr.data_ptr(), scr.numel()
    a.out, a.out_sn, a.out_ld = out.data_ptr(), OH * OW * out_ld, out_ld
    if direct:
        a.direct_ok = 1
        if avgpool:
            assert lib.caddy_k_conv_avgpool_ok(C.byref(a)) == 1
            a.avgpool = 1
        if not split:                      # the slab scratch stays untouched when the latency kernel takes the launch
            scr = torch.full((8 * N * H * W * round_up(out_c, 4),), 3.25, device=dev)
            a.split_scratch, a.split_cap = scr.data_ptr(), scr.numel()
    if pool:                               # MaxPool2d(2, 2) of the activated output written by the conv epilogue (floor on odd sizes)
        pld = round_up(out_c, 4) + 4
        pout = torch.full((N, H // 2, W // 2, pld), 5.5, device=dev)
        a.pool_out, a.pool_sn, a.pool_ld, a.skip_out = pout.data_ptr(), (H // 2) * (W // 2) * pld, pld, 1 if skip_out else 0
    lib.caddy_k_hx_force_big(big)          # 1: the 8-wave 16x16x128 variant with the 3-deep weight-tile ring even on a small grid
    try:
        assert lib.caddy_k_conv_fwd(C.byref(a), st) == 0
    finally:
        lib.caddy_k_hx_force_big(-1)
    sync(dev)
    y = to_nchw(out, out_c).double()
    scale = ref.abs().max().item()
    if tol is None:
        tol = {PREC_F16X3: 2e-6, PREC_BF16X3: 1e-4, PREC_F16X1: 4e-3, PREC_BF16X1: 3e-2}[precision]
    if pool:
        pref = F.max_pool2d(ref, 2, 2)
        perr = (to_nchw(pout, out_c).double() - pref).abs().max().item() / scale
        assert perr < tol, ("fused max-pool", perr, tol)
        assert torch.all(pout[..., out_c:].cpu() == 5.5)
    if skip_out:
        assert torch.equal(out.cpu(), init)                                 # the full-resolution output was not written at all
        return 0.0
    err = (y - ref).abs().max().item() / scale
    assert err < tol, ("hx conv", err, tol)
    assert torch.equal(out[..., out_c:].cpu(), init[..., out_c:])           # pad channels untouched
    if direct:
        assert lib.caddy_k_conv_took_direct() == (0 if direct == "tile4" else 1), ("latency kernel taken?", lib.caddy_k_conv_took_direct(), direct)
    return err


def lstm_fused_reduce_case(lib, dev, N=1, H=16, W=16, Cin=64, Cc=32, seed=0):
    """ConvArgs.lstm (round 4): the slab reduce of a K-split gate convolution applies the ConvLSTM cell update itself (roll-out frames: one launch instead of k_split_reduce +
    k_map<FLstmFwd>).  Same sums in the same order, same expressions (up to the compiler's fma contraction: <= 2 ulp): h' and c' agree with the two-kernel path to 1e-6,
    hb = h' * scale + shift, the gate tensor is not written, and the in-place form of the persistent roll-out state (c' over c) works."""
    from playablevideogeneration_amd._lib import LstmFuse
    lib.caddy_k_hx_weight_bytes.restype = C.c_long
    g = torch.Generator().manual_seed(seed)
    Cout = 4 * Cc
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    cp = torch.randn(N, Cc, H, W, generator=g)
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    st = stream(dev)
    w_d = w.contiguous().to(dev)
    d = make_pack([w], [(0, Cin)], 3, lib)
    d.w[0] = w_d.data_ptr()
    rows_pad = round_up(Cout, lib.caddy_k_hx_pick_bn(Cout))
    wq = torch.zeros(lib.caddy_k_hx_weight_bytes(C.byref(d), -1, rows_pad, 2), dtype=torch.uint8, device=dev)
    assert lib.caddy_k_pack_hx(C.byref(d), P(wq), rows_pad, -1, PREC_F16X3, st) == 0
    xb, b_d = nhwc(x, dev=dev), b.to(dev)
    scr = torch.zeros(16 * N * H * W * Cout, device=dev)

    def args(out):
        a = ConvArgs()
        a.src[0] = ConvSrc(xb.data_ptr(), H * W * xb.shape[3], xb.shape[3], Cin, round_up(Cin, CONV_BK), 0)
        a.nsrc, a.N, a.H, a.W, a.KS = 1, N, H, W, 3
        a.wp, a.Ktot, a.Cout, a.Cout_pad = None, round_up(Cin, CONV_BK), Cout, round_up(Cout, lib.caddy_k_conv_pick_bn(Cout))
        a.wq, a.precision, a.bias = wq.data_ptr(), PREC_F16X3, b_d.data_ptr()
        a.out, a.out_sn, a.out_ld = out.data_ptr(), H * W * Cout, Cout
        a.split_scratch, a.split_cap = scr.data_ptr(), scr.numel()
        return a
    gates = torch.full((N, H, W, Cout), 9.0, device=dev)
    a = args(gates)
    assert lib.caddy_k_conv_fwd(C.byref(a), st) == 0
    cpb = nhwc(cp, dev=dev)
    h1, c1 = torch.zeros_like(cpb), torch.zeros_like(cpb)
    assert lib.caddy_k_lstm_fwd(C.byref(tv(gates, Cout)), C.byref(tv(cpb, Cc)), C.byref(tv(h1, Cc)), C.byref(tv(c1, Cc)), st) == 0
    sync(dev)
    gates2 = torch.full((N, H, W, Cout), 9.0, device=dev)
    h2, c2, hb2 = torch.zeros_like(cpb), cpb.clone(), torch.zeros_like(cpb)      # c' written over c (the persistent roll-out state)
    sc_d, sh_d = sc.to(dev), sh.to(dev)
    lf = LstmFuse(c2.data_ptr(), H * W * Cc, Cc, h2.data_ptr(), H * W * Cc, Cc, c2.data_ptr(), H * W * Cc, Cc, hb2.data_ptr(), H * W * Cc, Cc, sc_d.data_ptr(), sh_d.data_ptr(), Cc)
    a = args(gates2)
    a.lstm = C.cast(C.pointer(lf), C.c_void_p)
    assert lib.caddy_k_conv_fwd(C.byref(a), st) == 0
    sync(dev)
    assert torch.all(gates2 == 9.0), "the fused reduce was not taken (the gate tensor was written)"
    assert (h1 - h2).abs().max().item() < 1e-6 and (c1 - c2).abs().max().item() < 1e-6, ((h1 - h2).abs().max().item(), (c1 - c2).abs().max().item())
    assert (hb2.cpu() - (h2.cpu() * sc + sh)).abs().max().item() < 1e-6


def hx_saturation_case(lib, dev, N=1, H=10, W=20, Cin=40, Cout=48, seed=0):
    """f16 range guard of the split-f16 forward (ConvArgs.sat_flag): inputs beyond +-65504 are clamped to the f16 range while they are staged (no inf - inf = NaN in the low
    half) and the launch sets the flag; in-range inputs leave the flag alone.  Reference: fp64 conv2d of the CLAMPED input."""
    lib.caddy_k_hx_weight_bytes.restype = C.c_long
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    st = stream(dev)
    w_d = w.contiguous().to(dev)
    d = make_pack([w], [(0, Cin)], 3, lib)
    d.w[0] = w_d.data_ptr()
    rows_pad = round_up(Cout, lib.caddy_k_hx_pick_bn(Cout))
    wq = torch.zeros(lib.caddy_k_hx_weight_bytes(C.byref(d), -1, rows_pad, 2), dtype=torch.uint8, device=dev)
    assert lib.caddy_k_pack_hx(C.byref(d), P(wq), rows_pad, -1, PREC_F16X3, st) == 0
    res = []
    for big_values in (False, True):
        x = torch.randn(N, Cin, H, W, generator=g)
        if big_values:
            x[0, 3, 2, 5], x[0, 7, 9, 19], x[0, 39, 0, 0] = 1.0e5, -2.5e5, 7.0e4
        flag = torch.zeros(4, dtype=torch.int32, device=dev)
        bb = nhwc(x, dev=dev)
        a = ConvArgs()
        a.src[0] = ConvSrc(bb.data_ptr(), H * W * bb.shape[3], bb.shape[3], Cin, round_up(Cin, CONV_BK), 0)
        a.nsrc, a.N, a.H, a.W, a.KS = 1, N, H, W, 3
        a.wp, a.Ktot, a.Cout, a.Cout_pad = None, round_up(Cin, CONV_BK), Cout, round_up(Cout, lib.caddy_k_conv_pick_bn(Cout))
        a.wq, a.precision, a.sat_flag = wq.data_ptr(), PREC_F16X3, flag.data_ptr()
        out = torch.zeros(N, H, W, round_up(Cout, 4), device=dev)
        a.out, a.out_sn, a.out_ld = out.data_ptr(), H * W * out.shape[3], out.shape[3]
        assert lib.caddy_k_conv_fwd(C.byref(a), st) == 0
        sync(dev)
        ref = F.conv2d(x.clamp(-65504.0, 65504.0).double(), w.double(), None, padding=1)
        y = to_nchw(out, Cout).double()
        assert torch.isfinite(y).all()
        err = (y - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-6, ("saturating conv", big_values, err)
        assert int(flag[0].item()) == (1 if big_values else 0), (big_values, flag.tolist())
        res.append(err)
    return res


def hx_lazy_bn_case(lib, dev, *, N, H, W, Cin, Cout, aux_c=0, act=1, seed=0, big=-1, groups=1, split=False):
    """The BatchNorm fusion of round 3 as one chain at kernel level (reference: conv -> BatchNorm2d(train) -> LeakyReLU(0.2) -> conv, e.g. conv1 / bn1 /
    conv2 of model/layers/residual_block.py:51-61), against torch autograd in fp64:
      x0 (raw output of a producing conv) --[statistics + finalisation]--> (mean, invstd, scale, shift)
      y = conv3x3(act(x0 * scale + shift) [, broadcast vector])     k_conv_hx with ConvSrc.bn_* (the normalised tensor is never written)
          + per-tile partial sums of y from the epilogue (ConvArgs.stats) -> caddy_k_bn_finalize_tiles == batch statistics of y
      dW = wgrad(dy, act(x0 * scale + shift))                        k_wgrad_hx with ConvSrc.bn_*
      d(x0) via the backward of the never-materialised BatchNorm      caddy_k_bn_bwd_lazy on the dgrad's output
    groups > 1: the batch is `groups` independent BatchNorm calls (time-batched launch, ConvSrc.bn_gn / bn_gs): statistics per group.
    split: give the launcher slab scratch -- an under-filled launch then splits K and the partial sums come from the slab reduce (k_split_reduce_stats)."""
    lib.caddy_k_hx_weight_bytes.restype = C.c_long
    g = torch.Generator().manual_seed(seed)
    st = stream(dev)
    assert N % groups == 0
    gn = N // groups
    x0 = torch.randn(N, Cin, H, W, generator=g) * 1.7 + 0.3
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    av = torch.randn(N, aux_c, generator=g) if aux_c else None
    Ct = Cin + aux_c
    w = torch.randn(Cout, Ct, 3, 3, generator=g) / (Ct * 9) ** 0.5
    dy = torch.randn(N, Cout, H, W, generator=g) * 1e-3
    # ---- reference (fp64 autograd), BatchNorm statistics per group ----
    xr, wr, gr, br = x0.double().requires_grad_(True), w.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    parts, means, invs = [], [], []
    for q in range(groups):
        xg = xr[q * gn:(q + 1) * gn]
        mu = xg.mean(dim=(0, 2, 3)); var = xg.var(dim=(0, 2, 3), unbiased=False)
        inv = 1.0 / torch.sqrt(var + 1e-5)
        bnv = (xg - mu[None, :, None, None]) * (inv * gr)[None, :, None, None] + br[None, :, None, None]
        parts.append(F.leaky_relu(bnv, 0.2) if act else bnv)
        means.append(mu.detach()); invs.append(inv.detach())
    a_ref = torch.cat(parts, 0)
    full = torch.cat([a_ref, av.double()[:, :, None, None].expand(-1, -1, H, W)], 1) if aux_c else a_ref
    y_ref = F.conv2d(full, wr, None, padding=1)
    (y_ref * dy.double()).sum().backward()
    # ---- device side ----
    x_d = nhwc(x0, dev=dev)
    cp = round_up(Cin, 4)
    tab = torch.zeros(groups, 4 * cp, device=dev)                   # per group: mean | invstd | scale | shift (the driver's BNStash layout)
    gam_d, bet_d = gamma.to(dev), beta.to(dev)
    rmean, rvar = torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev)
    sums = torch.zeros(2 * Cin, dtype=torch.float64, device=dev)
    scratch = torch.zeros(512 * 2 * 1024, dtype=torch.float64, device=dev)
    sn = H * W * x_d.shape[3]
    for q in range(groups):
        xt = TV(x_d.data_ptr() + 4 * q * gn * sn, gn, H, W, Cin, sn, x_d.shape[3])
        t0 = tab.data_ptr() + 4 * q * 4 * cp
        assert lib.caddy_k_bn_stats_finalize(C.byref(xt), P(sums), P(scratch), P(gam_d), P(bet_d), P(rmean), P(rvar),
                                             C.c_void_p(t0), C.c_void_p(t0 + 4 * cp), C.c_void_p(t0 + 8 * cp), C.c_void_p(t0 + 12 * cp), st) == 0
    sync(dev)
    for q in range(groups):
        assert (tab[q, :Cin].cpu().double() - means[q]).abs().max().item() < 1e-5 and (tab[q, cp:cp + Cin].cpu().double() / invs[q] - 1).abs().max().item() < 1e-5
    segs = [(0, Cin)] + ([(Cin, aux_c)] if aux_c else [])
    d = make_pack([w], segs, 3, lib)
    w_d = w.contiguous().to(dev)
    gw_d = torch.zeros_like(w_d)
    d.w[0], d.gw[0] = w_d.data_ptr(), gw_d.data_ptr()
    rows_pad = round_up(Cout, lib.caddy_k_hx_pick_bn(Cout))
    wq = torch.zeros(lib.caddy_k_hx_weight_bytes(C.byref(d), -1, rows_pad, 2), dtype=torch.uint8, device=dev)
    assert lib.caddy_k_pack_hx(C.byref(d), P(wq), rows_pad, -1, PREC_F16X3, st) == 0
    a = ConvArgs()
    lazy = ConvSrc(x_d.data_ptr(), sn, x_d.shape[3], Cin, round_up(Cin, CONV_BK), 0, tab.data_ptr() + 8 * cp, tab.data_ptr() + 12 * cp, act,
                   gn if groups > 1 else 0, 4 * cp if groups > 1 else 0)
    a.src[0] = lazy
    keep = [x_d]
    if aux_c:
        bb = torch.full((N, 16), 7.5); bb[:, :aux_c] = av; bb = bb.to(dev); keep.append(bb)
        a.src[1] = ConvSrc(bb.data_ptr(), 16, 16, aux_c, round_up(aux_c, CONV_BK), 1)
    a.nsrc, a.N, a.H, a.W, a.KS = len(segs), N, H, W, 3
    a.wp, a.Ktot, a.Cout, a.Cout_pad = None, d.Ktot, Cout, d.Cout_pad
    a.wq, a.precision = wq.data_ptr(), PREC_F16X3
    out_ld = round_up(Cout, 4)
    out = torch.full((N, H, W, out_ld), 9.0, device=dev)
    a.out, a.out_sn, a.out_ld = out.data_ptr(), H * W * out_ld, out_ld
    ldp = round_up(Cout, 4)
    max_tiles = max(512, N * ((H + 7) // 8) * ((W + 15) // 16))          # common.h: conv_stats_tiles_cap
    part = torch.full((max_tiles * ldp * 2,), float("nan"), device=dev)
    a.stats, a.stats_ld = part.data_ptr(), ldp
    if split:
        scr = torch.zeros(16 * N * H * W * round_up(Cout, 4), device=dev)
        a.split_scratch, a.split_cap = scr.data_ptr(), scr.numel()
    lib.caddy_k_hx_force_big(big)
    try:
        assert lib.caddy_k_conv_fwd(C.byref(a), st) == 0
    finally:
        lib.caddy_k_hx_force_big(-1)
    sync(dev)
    y = to_nchw(out, Cout).double()
    scale = y_ref.abs().max().item()
    err = (y - y_ref.detach()).abs().max().item() / scale
    assert err < 5e-6, ("conv on the lazily normalised input", err)
    tiles = lib.caddy_k_conv_stats_tiles()
    assert 0 < tiles <= max_tiles, tiles
    assert groups == 1 or tiles % groups == 0
    # statistics of y from the epilogue's partial sums == batch statistics of y, per group
    tpg = tiles // groups
    for q in range(groups):
        o4 = torch.zeros(4 * ldp, device=dev)
        rm2, rv2 = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
        g2, b2 = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        assert lib.caddy_k_bn_finalize_tiles(C.c_void_p(part.data_ptr() + 4 * q * tpg * ldp * 2), tpg, ldp, C.c_long(gn * H * W), P(g2), P(b2), P(rm2), P(rv2), Cout,
                                             C.c_void_p(o4.data_ptr()), C.c_void_p(o4.data_ptr() + 4 * ldp), C.c_void_p(o4.data_ptr() + 8 * ldp), C.c_void_p(o4.data_ptr() + 12 * ldp), st) == 0
        sync(dev)
        yq = y[q * gn:(q + 1) * gn]
        mu_y, var_y = yq.mean(dim=(0, 2, 3)), yq.var(dim=(0, 2, 3), unbiased=False)
        assert (o4[:Cout].cpu().double() - mu_y).abs().max().item() < 2e-6 * max(1.0, scale), ("mean from tiles", q)
        assert (o4[ldp:ldp + Cout].cpu().double() * torch.sqrt(var_y + 1e-5) - 1).abs().max().item() < 2e-5, ("invstd from tiles", q)
        cnt = gn * H * W
        assert (rv2.cpu().double() - (0.9 + 0.1 * var_y * cnt / max(1, cnt - 1))).abs().max().item() < 1e-5 * max(1.0, var_y.max().item()), "running_var"
    # ---- weight gradient on k_wgrad_hx with the lazily normalised source ----
    dz_d = nhwc(dy, dev=dev)
    dwp = torch.zeros(9 * d.Cout_pad * d.Ktot, device=dev)
    wa = WgradArgs()
    for i in range(a.nsrc):
        wa.src[i] = a.src[i]
    wa.nsrc, wa.N, wa.H, wa.W, wa.KS = a.nsrc, N, H, W, 3
    wa.dy, wa.dy_sn, wa.dy_ld = dz_d.data_ptr(), H * W * dz_d.shape[3], dz_d.shape[3]
    wa.Cout, wa.Cout_pad, wa.Ktot, wa.dwp, wa.slabs, wa.precision = Cout, d.Cout_pad, d.Ktot, dwp.data_ptr(), 0, PREC_BF16X3
    assert lib.caddy_k_conv_wgrad(C.byref(wa), st) == 0
    assert lib.caddy_k_unpack_wgrad(C.byref(d), P(dwp), st) == 0
    sync(dev)
    e = (gw_d.cpu().double() - wr.grad).abs().max().item() / wr.grad.abs().max().item()
    assert e < 2e-4, ("wgrad on the lazily normalised source", e)
    if groups > 1:      # the same through the time-batched addressing: groups laid out `groups` launches apart (src_gs / dy_gs / src_bn_gs)
        wg = WgradArgs()
        wg.src[0] = ConvSrc(x_d.data_ptr(), sn, x_d.shape[3], Cin, round_up(Cin, CONV_BK), 0, tab.data_ptr() + 8 * cp, tab.data_ptr() + 12 * cp, act, 0, 0)
        wg.src_gs[0], wg.src_bn_gs[0] = gn * sn, 4 * cp
        if aux_c:
            wg.src[1] = a.src[1]; wg.src_gs[1] = gn * 16
        dwp2 = torch.zeros_like(dwp)
        wg.nsrc, wg.N, wg.H, wg.W, wg.KS, wg.group_n = a.nsrc, N, H, W, 3, gn
        wg.dy, wg.dy_sn, wg.dy_ld, wg.dy_gs = dz_d.data_ptr(), H * W * dz_d.shape[3], dz_d.shape[3], gn * H * W * dz_d.shape[3]
        wg.Cout, wg.Cout_pad, wg.Ktot, wg.dwp, wg.slabs, wg.precision = Cout, d.Cout_pad, d.Ktot, dwp2.data_ptr(), 0, PREC_BF16X3
        assert lib.caddy_k_conv_wgrad(C.byref(wg), st) == 0
        sync(dev)
        e2 = (dwp2 - dwp).abs().max().item() / dwp.abs().max().item()
        assert e2 < 2e-4, ("time-batched wgrad with per-group BatchNorm tables", e2)
    # ---- backward of the BatchNorm that was never materialised: dgrad -> d(normalised), then reduce + apply with the slope from x0 * scale + shift ----
    rows_d = round_up(Cin, lib.caddy_k_hx_pick_bn(Cin))
    wqd = torch.zeros(lib.caddy_k_hx_weight_bytes(C.byref(d), 0, rows_d, 2), dtype=torch.uint8, device=dev)
    assert lib.caddy_k_pack_hx(C.byref(d), P(wqd), rows_d, 0, PREC_BF16X3, st) == 0
    da = ConvArgs()
    da.src[0] = ConvSrc(dz_d.data_ptr(), H * W * dz_d.shape[3], dz_d.shape[3], Cout, round_up(Cout, CONV_BK), 0)
    da.nsrc, da.N, da.H, da.W, da.KS = 1, N, H, W, 3
    da.wp, da.Ktot, da.Cout, da.Cout_pad = None, round_up(Cout, CONV_BK), Cin, round_up(Cin, lib.caddy_k_conv_pick_bn(Cin))
    da.wq, da.precision = wqd.data_ptr(), PREC_BF16X3
    ga = torch.full((N, H, W, cp), float("nan"), device=dev)          # first-touch gradient: assigned
    da.out, da.out_sn, da.out_ld, da.accumulate = ga.data_ptr(), H * W * cp, cp, 0
    assert lib.caddy_k_conv_fwd(C.byref(da), st) == 0
    gx = torch.full((N, H, W, cp), float("nan"), device=dev)
    dgam, dbet = torch.zeros(Cin, device=dev), torch.zeros(Cin, device=dev)
    for q in range(groups):
        o = 4 * q * gn * H * W * cp
        dt = TV(ga.data_ptr() + o, gn, H, W, Cin, H * W * cp, cp)
        xt = TV(x_d.data_ptr() + 4 * q * gn * sn, gn, H, W, Cin, sn, x_d.shape[3])
        gt = TV(gx.data_ptr() + o, gn, H, W, Cin, H * W * cp, cp)
        t0 = tab.data_ptr() + 4 * q * 4 * cp
        assert lib.caddy_k_bn_bwd_lazy(C.byref(dt), C.byref(xt), C.c_void_p(t0), C.c_void_p(t0 + 4 * cp), P(gam_d), C.c_void_p(t0 + 8 * cp), C.c_void_p(t0 + 12 * cp), act,
                                       P(sums), P(scratch), C.byref(gt), P(dgam), P(dbet), st) == 0
    sync(dev)
    gref = xr.grad
    e = (to_nchw(gx, Cin).double() - gref).abs().max().item() / gref.abs().max().item()
    assert e < 5e-4, ("d(x0) through the lazily applied BatchNorm", e)
    assert (dgam.cpu().double() - gr.grad).abs().max().item() < 5e-4 * gr.grad.abs().max().item() and (dbet.cpu().double() - br.grad).abs().max().item() < 5e-4 * br.grad.abs().max().item()
    return err


# ---- S16 tensors (csrc/common.h): activations stored pre-split for the 16-bit matrix pipe ----
def s16_decode(buf, Cc, dtype=torch.float16):
    """(N,H,W,ld) fp32-typed raw buffer holding an S16 tensor -> (hi, lo) as (N,H,W,C) tensors of `dtype` (cpu)."""
    raw = buf.cpu().contiguous().view(dtype)                                 # (N,H,W,2*ld) halves
    N, H, W, _ = raw.shape
    ch = raw[..., :2 * Cc].reshape(N, H, W, Cc // 32, 2, 32)
    return ch[..., 0, :].reshape(N, H, W, Cc), ch[..., 1, :].reshape(N, H, W, Cc)


def s16_encode(x_nhwc, dtype=torch.float16, dev="cpu"):
    """(N,H,W,C) fp32 cpu tensor -> raw fp32-typed (N,H,W,C) buffer holding the S16 tensor: per 32-channel chunk [hi x 32 | lo x 32], hi = round16(x), lo = round16(x - hi)."""
    N, H, W, Cc = x_nhwc.shape
    hi = x_nhwc.to(dtype)
    lo = (x_nhwc - hi.float()).to(dtype)
    ch = torch.stack([hi.reshape(N, H, W, Cc // 32, 32), lo.reshape(N, H, W, Cc // 32, 32)], dim=4)      # (N,H,W,chunks,2,32)
    return ch.reshape(N, H, W, 2 * Cc).contiguous().view(torch.float32).to(dev)


def hx_s16_chain_case(lib, dev, *, N, H, W, C0, C1, C2, pool=False, seed=0):
    """Two stacked 3x3 conv + bias + ReLU layers on k_conv_hx, once with an fp32 intermediate and once with the intermediate (and, with `pool`, its fused 2x2 max-pool) exchanged as
    an S16-f16 tensor (ConvArgs.out_s16 / pool_s16 -> in_s16): the second layer's output must be BIT-IDENTICAL, the S16 tensor must hold exactly round16(x) | round16(x - hi).
    Then the same for a dgrad pair in split bf16 with the ReLU-mask / L1-seed epilogue reading S16-f16 masks.  Well-filled tile variants forced (caddy_k_hx_force_big(1))."""
    lib.caddy_k_hx_weight_bytes.restype = C.c_long
    g = torch.Generator().manual_seed(seed)
    st = stream(dev)

    def layer(cin, cout, prec, dgrad=False):
        w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        w_d = w.contiguous().to(dev)
        d = make_pack([w], [(0, cin)], 3, lib)
        d.w[0] = w_d.data_ptr()
        rows = cin if dgrad else cout
        rows_pad = round_up(rows, lib.caddy_k_hx_pick_bn(rows))
        wq = torch.zeros(lib.caddy_k_hx_weight_bytes(C.byref(d), 0 if dgrad else -1, rows_pad, 2), dtype=torch.uint8, device=dev)
        assert lib.caddy_k_pack_hx(C.byref(d), P(wq), rows_pad, 0 if dgrad else -1, prec, st) == 0
        b = (torch.randn(cout, generator=g) * 0.1).to(dev)
        return dict(w=w, w_d=w_d, d=d, wq=wq, b=b, cin=cin, cout=cout)

    def run(L, x_buf, cin, cout_eff, Hh, Ww, prec, *, in_s16=0, out_s16=0, bias=True, act=2, pool_buf=None, pool_s16=0, mask=None, mask_s16=0, seedref=None, seed_s16=0, seed_w=0.0):
        a = ConvArgs()
        a.src[0] = ConvSrc(x_buf.data_ptr(), Hh * Ww * x_buf.shape[3], x_buf.shape[3], cin, round_up(cin, CONV_BK), 0)
        a.nsrc, a.N, a.H, a.W, a.KS = 1, N, Hh, Ww, 3
        a.wp, a.Ktot, a.Cout, a.Cout_pad = None, round_up(cin, CONV_BK), cout_eff, round_up(cout_eff, lib.caddy_k_conv_pick_bn(cout_eff))
        a.wq, a.precision = L["wq"].data_ptr(), prec
        a.bias, a.act = (L["b"].data_ptr() if bias else None), act
        out = torch.full((N, Hh, Ww, cout_eff), 9.25, device=dev)
        a.out, a.out_sn, a.out_ld = out.data_ptr(), Hh * Ww * cout_eff, cout_eff
        a.in_s16, a.out_s16, a.pool_s16 = in_s16, out_s16, pool_s16
        if pool_buf is not None:
            a.pool_out, a.pool_sn, a.pool_ld = pool_buf.data_ptr(), (Hh // 2) * (Ww // 2) * cout_eff, cout_eff
        if mask is not None:
            a.mask, a.mask_s16 = mask.data_ptr(), mask_s16
        if seedref is not None:
            a.seed_ref, a.seed_s16, a.seed_w = seedref.data_ptr(), seed_s16, seed_w
        lib.caddy_k_hx_force_big(1)
        try:
            rc = lib.caddy_k_conv_fwd(C.byref(a), st)
        finally:
            lib.caddy_k_hx_force_big(-1)
        assert rc == 0, rc
        sync(dev)
        return out

    # ---- forward pair, split f16 ----
    x = torch.randn(N, C0, H, W, generator=g)
    xb = nhwc(x, dev=dev)
    A, Bl = layer(C0, C1, PREC_F16X3), layer(C1, C2, PREC_F16X3)
    Hp, Wp = (H // 2, W // 2) if pool else (H, W)
    p32 = torch.full((N, Hp, Wp, C1), 1.5, device=dev) if pool else None
    p16 = torch.full((N, Hp, Wp, C1), 1.5, device=dev) if pool else None
    y32 = run(A, xb, C0, C1, H, W, PREC_F16X3, pool_buf=p32)
    y16 = run(A, xb, C0, C1, H, W, PREC_F16X3, out_s16=1, pool_buf=p16, pool_s16=1 if pool else 0)
    for f32, s16 in ((y32, y16),) + (((p32, p16),) if pool else ()):
        hi, lo = s16_decode(s16, C1)
        v = f32.cpu()
        assert torch.equal(hi, v.half()), "S16 high halves"
        assert torch.equal(lo, (v - v.half().float()).half()), "S16 low halves"
    z32 = run(Bl, p32 if pool else y32, C1, C2, Hp, Wp, PREC_F16X3)
    z16 = run(Bl, p16 if pool else y16, C1, C2, Hp, Wp, PREC_F16X3, in_s16=1)
    assert torch.equal(z32.cpu(), z16.cpu()), ("S16 input vs fp32 input", (z32.cpu() - z16.cpu()).abs().max().item())
    z16b = run(Bl, p16 if pool else y16, C1, C2, Hp, Wp, PREC_F16X3, in_s16=1, out_s16=1)      # both at once
    hi, lo = s16_decode(z16b, C2)
    assert torch.equal(hi, z32.cpu().half()) and torch.equal(lo, (z32.cpu() - z32.cpu().half().float()).half())
    ref = F.relu(F.conv2d(x.double(), A["w"].double(), A["b"].cpu().double(), padding=1))
    err = (to_nchw(y32, C1).double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err

    # ---- dgrad pair, split bf16: gz2 (C2 channels) -> [dgrad of layer B, masked by the forward activation of layer A, + L1 seed] -> gz1 (C1) -> [dgrad of layer A] -> (C0) ----
    DB, DA = layer(C1, C2, PREC_BF16X3, dgrad=True), layer(C0, C1, PREC_BF16X3, dgrad=True)
    gz2 = torch.randn(N, Hp, Wp, C2, generator=g) * 1e-4
    gz2_32 = gz2.to(dev)
    gz2_16 = s16_encode(gz2, torch.bfloat16, dev)
    act1 = (p32 if pool else y32)                              # ReLU output of layer A (mask: > 0), fp32 and S16 forms
    act1_s = (p16 if pool else y16)
    tap = torch.relu(torch.randn(N, Hp, Wp, C1, generator=g))
    tap32, tap16 = tap.to(dev), s16_encode(tap, torch.float16, dev)
    g1_32 = run(DB, gz2_32, C2, C1, Hp, Wp, PREC_BF16X3, bias=False, act=0, mask=act1, seedref=tap32, seed_w=1e-5)
    g1_m16 = run(DB, gz2_32, C2, C1, Hp, Wp, PREC_BF16X3, bias=False, act=0, mask=act1_s, mask_s16=1, seedref=tap16, seed_s16=1, seed_w=1e-5)      # S16 masks on the plain launch
    g1_16 = run(DB, gz2_16, C2, C1, Hp, Wp, PREC_BF16X3, bias=False, act=0, in_s16=1, out_s16=1, mask=act1_s, mask_s16=1, seedref=tap16, seed_s16=1, seed_w=1e-5)
    # (the S16 forms see hi + lo of the activations: identical signs unless |mask - tap| is below 2^-22 relative -- never for these random values)
    assert torch.equal(g1_32.cpu(), g1_m16.cpu()), (g1_32.cpu() - g1_m16.cpu()).abs().max().item()
    hi, lo = s16_decode(g1_16, C1, torch.bfloat16)
    v = g1_32.cpu()
    assert torch.equal(hi, v.bfloat16()) and torch.equal(lo, (v - v.bfloat16().float()).bfloat16()), "S16-bf16 gradient"
    if not pool:
        g0_32 = run(DA, g1_32, C1, C0, H, W, PREC_BF16X3, bias=False, act=0)
        g0_16 = run(DA, g1_16, C1, C0, H, W, PREC_BF16X3, bias=False, act=0, in_s16=1)
        assert torch.equal(g0_32.cpu(), g0_16.cpu())
    return err


# ---- round 6: the gradient of a convolution output exchanged pre-split (TV::s16 / WgradArgs.dy_s16 / ConvArgs.in_s16 with split bf16) ----
def s16_grad_case(lib, dev, *, N, H, W, Cin, Cout, seed=0, force_big=-1, producers=True):
    """dY of a 3x3 convolution Cin -> Cout as an S16-bf16 tensor.  Producers (the point-wise backward kernels that assign a conv output's gradient: BatchNorm backward, its one-launch
    form, average-pool backward, ConvLSTM cell backward) must write EXACTLY [bf16(v) | bf16(v - bf16(v))] of the value their fp32 form writes; readers must give BIT-IDENTICAL weight
    gradients (k_wgrad_hx, also time-batched) and dgrads (every k_conv_hx tile variant the shape selects, accumulating + deterministic K split) to those of the fp32 tensor, and the
    column / border sums the sums of hi + lo."""
    assert Cout % 32 == 0
    g = torch.Generator().manual_seed(seed)
    st = stream(dev)
    bf = torch.bfloat16
    lib.caddy_k_hx_weight_bytes.restype = C.c_long

    def tv16(buf, Cc):
        t = tv(buf, Cc)
        t.s16 = 1
        return t

    def check_s16(buf16, ref32, what):
        hi, lo = s16_decode(buf16, ref32.shape[3], bf)
        v = ref32.cpu()
        assert torch.equal(hi, v.to(bf)), what + ": high halves"
        assert torch.equal(lo, (v - v.to(bf).float()).to(bf)), what + ": low halves"

    # gradient magnitudes over many binades (bf16 halves: no lower bound)
    dy = torch.randn(N, Cout, H, W, generator=g) * torch.exp(torch.randn(N, Cout, 1, 1, generator=g) * 4.0 - 6.0)
    if producers:
        # (a) avg_pool2d(2) backward, assigning
        if H % 2 == 0 and W % 2 == 0:
            dpo = nhwc(torch.randn(N, Cout, H // 2, W // 2, generator=g), dev=dev)
            d32, d16 = torch.full((N, H, W, Cout), 3.0, device=dev), torch.full((N, H, W, Cout), 3.0, device=dev)
            assert lib.caddy_k_pool2_bwd_assign(C.byref(tv(dpo, Cout)), C.byref(tv(d32, Cout)), st) == 0
            assert lib.caddy_k_pool2_bwd_assign(C.byref(tv(dpo, Cout)), C.byref(tv16(d16, Cout)), st) == 0
            sync(dev)
            check_s16(d16, d32, "pool2_bwd")
        # (b) BatchNorm backward apply (assigning; LeakyReLU slope from the materialised output) and (c) its one-launch form
        x = nhwc(torch.randn(N, Cout, H, W, generator=g), dev=dev)
        om = nhwc(torch.randn(N, Cout, H, W, generator=g), dev=dev)
        dout = nhwc(dy, dev=dev)
        mean, invstd, gamma = (torch.randn(Cout, generator=g) * 0.1).to(dev), (torch.rand(Cout, generator=g) + 0.5).to(dev), (torch.rand(Cout, generator=g) + 0.5).to(dev)
        sums = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
        assert lib.caddy_k_bn_bwd_reduce(C.byref(tv(dout, Cout)), C.byref(tv(om, Cout)), C.byref(tv(x, Cout)), P(mean), P(invstd), P(sums), st) == 0
        d32, d16 = torch.full((N, H, W, Cout), 3.0, device=dev), torch.full((N, H, W, Cout), 3.0, device=dev)
        for dst in (tv(d32, Cout), tv16(d16, Cout)):
            assert lib.caddy_k_bn_bwd_apply_assign(C.byref(tv(dout, Cout)), C.byref(tv(om, Cout)), C.byref(tv(x, Cout)), P(mean), P(invstd), P(gamma), P(sums), C.byref(dst), st) == 0
        sync(dev)
        check_s16(d16, d32, "bn_bwd_apply")
        if N * H * W <= 8192:
            dg, db = torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev)
            d32, d16 = torch.full((N, H, W, Cout), 3.0, device=dev), torch.full((N, H, W, Cout), 3.0, device=dev)
            for dst in (tv(d32, Cout), tv16(d16, Cout)):
                assert lib.caddy_k_bn_small_bwd_assign(C.byref(tv(dout, Cout)), C.byref(tv(om, Cout)), C.byref(tv(x, Cout)), P(mean), P(invstd), P(gamma), C.byref(dst), P(dg), P(db), None, st) == 0
            sync(dev)
            check_s16(d16, d32, "bn_small_bwd")
        # (d) ConvLSTM cell backward: d(gates), 4 x Ch channels
        if Cout % 128 == 0:
            Ch = Cout // 4
            gates = nhwc(torch.rand(N, Cout, H, W, generator=g), dev=dev)
            cp, cn = nhwc(torch.randn(N, Ch, H, W, generator=g), dev=dev), nhwc(torch.randn(N, Ch, H, W, generator=g), dev=dev)
            dh, dc = nhwc(torch.randn(N, Ch, H, W, generator=g) * 1e-3, dev=dev), nhwc(torch.randn(N, Ch, H, W, generator=g) * 1e-3, dev=dev)
            d32, d16 = torch.full((N, H, W, Cout), 3.0, device=dev), torch.full((N, H, W, Cout), 3.0, device=dev)
            for dst in (tv(d32, Cout), tv16(d16, Cout)):
                dcp = torch.zeros(N, H, W, Ch, device=dev)
                assert lib.caddy_k_lstm_bwd(C.byref(tv(gates, Cout)), C.byref(tv(cp, Ch)), C.byref(tv(cn, Ch)), C.byref(tv(dh, Ch)), C.byref(tv(dc, Ch)), C.byref(dst), C.byref(tv(dcp, Ch)), st) == 0
            sync(dev)
            check_s16(d16, d32, "lstm_bwd")
    # ---- readers ----
    dz32 = nhwc(dy, dev=dev)                                               # (N,H,W,Cout), ld = Cout
    dz16 = s16_encode(dy.permute(0, 2, 3, 1).contiguous(), bf, dev)
    hi, lo = s16_decode(dz16, Cout, bf)
    dzv = (hi.float() + lo.float()).to(dev).contiguous()                   # what an S16 reader sees, as an fp32 tensor
    # debug read-out
    got = torch.zeros(N, Cout, H, W, device=dev)
    assert lib.caddy_k_nhwc_to_nchw(C.byref(tv16(dz16, Cout)), P(got), Cout * H * W, 0, st) == 0
    sync(dev)
    assert torch.equal(got.cpu(), dzv.cpu().permute(0, 3, 1, 2))
    # column sums (bias gradient): same kernel on hi + lo
    o32, o16 = torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev)
    assert lib.caddy_k_colsum(C.byref(tv(dzv, Cout)), P(o32), st) == 0 and lib.caddy_k_colsum(C.byref(tv16(dz16, Cout)), P(o16), st) == 0
    sync(dev)
    mag = dzv.abs().sum(dim=(0, 1, 2)).cpu()                                # (float atomics between workgroups: arrival order -- bound relative to sum |v|, the sums themselves cancel)
    assert ((o32.cpu() - o16.cpu()).abs() <= 1e-5 * mag + 1e-30).all(), "colsum"
    # the convolution itself
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    w_d = w.contiguous().to(dev)
    d = make_pack([w], [(0, Cin)], 3, lib)
    d.w[0] = w_d.data_ptr()
    x_in = nhwc(torch.randn(N, Cin, H, W, generator=g), dev=dev)
    # border sums -> gradient of a broadcast input + bias: S16 tensor vs the fp32 tensor of hi + lo, bit for bit
    wb = torch.randn(Cout, 8, 3, 3, generator=g)
    wb_d = wb.contiguous().to(dev)
    db_ = make_pack([wb], [(0, 8)], 3, lib)
    db_.w[0] = wb_d.data_ptr()
    res = []
    for t in (tv(dzv, Cout), tv16(dz16, Cout)):
        S, ga, dbias = torch.zeros(N * Cout * 9, device=dev), torch.zeros(N, 8, device=dev), torch.zeros(Cout, device=dev)
        assert lib.caddy_k_bcast_input_grad(C.byref(t), C.byref(db_), 0, P(S), P(ga), 8, P(dbias), st) == 0
        sync(dev)
        res.append((S.cpu(), ga.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), "border sums of an S16 gradient"
    lib.caddy_k_hx_force_big(force_big)
    try:
        # weight gradient (bit-reproducible mode), plain and time-batched
        wsz = 9 * d.Cout_pad * d.Ktot
        det = torch.zeros(1 << 22, device=dev)
        outs = []
        for buf, s16 in ((dz32, 0), (dz16, 1)):
            wa = WgradArgs()
            wa.src[0] = ConvSrc(x_in.data_ptr(), H * W * x_in.shape[3], x_in.shape[3], Cin, round_up(Cin, CONV_BK), 0)
            wa.nsrc, wa.N, wa.H, wa.W, wa.KS = 1, N, H, W, 3
            wa.dy, wa.dy_sn, wa.dy_ld, wa.dy_s16 = buf.data_ptr(), H * W * Cout, Cout, s16
            wa.Cout, wa.Cout_pad, wa.Ktot, wa.slabs, wa.precision = Cout, d.Cout_pad, d.Ktot, 0, PREC_BF16X3
            dwp = torch.zeros(wsz, device=dev)
            wa.dwp, wa.det_slab, wa.det_cap = dwp.data_ptr(), det.data_ptr(), det.numel()
            assert lib.caddy_k_conv_wgrad(C.byref(wa), st) == 0
            sync(dev)
            outs.append(dwp)
            if N % 2 == 0:
                wa.group_n, wa.dy_gs = N // 2, (N // 2) * H * W * Cout
                wa.src_gs[0] = (N // 2) * H * W * x_in.shape[3]
                dwp2 = torch.zeros(wsz, device=dev)
                wa.dwp = dwp2.data_ptr()
                assert lib.caddy_k_conv_wgrad(C.byref(wa), st) == 0
                sync(dev)
                assert torch.equal(dwp2, dwp), "time-batched weight gradient"
        assert torch.equal(outs[0], outs[1]), "weight gradient from a pre-split dY"
        assert outs[0].abs().max().item() > 0
        # dgrad: accumulating on top of ones, deterministic K split available
        bn = lib.caddy_k_hx_pick_bn(Cin)
        rows_pad = round_up(Cin, bn)
        wq = torch.zeros(lib.caddy_k_hx_weight_bytes(C.byref(d), 0, rows_pad, 2), dtype=torch.uint8, device=dev)
        assert lib.caddy_k_pack_hx(C.byref(d), P(wq), rows_pad, 0, PREC_BF16X3, st) == 0
        scr = torch.zeros(9 * N * H * W * round_up(Cin, 4), device=dev)
        for accumulate in (1, 0):
            outs = []
            for buf, s16 in ((dz32, 0), (dz16, 1)):
                da = ConvArgs()
                da.src[0] = ConvSrc(buf.data_ptr(), H * W * Cout, Cout, Cout, round_up(Cout, CONV_BK), 0)
                da.nsrc, da.N, da.H, da.W, da.KS = 1, N, H, W, 3
                da.wp, da.Ktot, da.Cout, da.Cout_pad = None, round_up(Cout, CONV_BK), Cin, round_up(Cin, lib.caddy_k_conv_pick_bn(Cin))
                da.wq, da.precision, da.in_s16 = wq.data_ptr(), PREC_BF16X3, s16
                gx = torch.ones((N, H, W, round_up(Cin, 4)), device=dev)
                da.out, da.out_sn, da.out_ld, da.accumulate = gx.data_ptr(), H * W * gx.shape[3], gx.shape[3], accumulate
                da.deterministic, da.split_scratch, da.split_cap = 1, scr.data_ptr(), scr.numel()
                assert lib.caddy_k_conv_fwd(C.byref(da), st) == 0
                sync(dev)
                outs.append(gx)
            assert torch.equal(outs[0], outs[1]), ("dgrad from a pre-split dY", accumulate)
            assert (outs[0] - 1.0).abs().max().item() > 0
    finally:
        lib.caddy_k_hx_force_big(-1)


def hx_register_weights_case(lib, dev, cases):
    """round 6 (conv_hx.hip, template parameter BG): the under-filled tile variants with the weight fragments straight from global memory -- every case runs with the variants
    forced on (caddy_k_hx_set_bg(15): bit 3 lifts the chunks-per-workgroup threshold) and forced off (0: LDS-staged weight tiles); hx_conv_case / s16_grad_case check each
    against the fp64 reference, so the two forms agree to the split-operand error although their summation orders differ (K halves per wave + one exchange vs whole K per wave)."""
    for mask in (15, 0):
        lib.caddy_k_hx_set_bg(mask)
        try:
            for kind, kw in cases:
                (hx_conv_case if kind == "conv" else s16_grad_case)(lib, dev, **kw)
        finally:
            lib.caddy_k_hx_set_bg(-1)
