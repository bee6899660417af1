"""HIP kernel sources executed on the host functional simulator (tests/emu) and compared with torch CPU ops."""
import pytest

from tests import kernel_cases as K
from tests.emu.loader import load_emu

pytestmark = pytest.mark.emu


@pytest.mark.parametrize("kw", [
    dict(N=2, H=6, W=5, segs=[(16, 0)], Cout=32, KS=3),
    dict(N=1, H=9, W=7, segs=[(5, 0)], Cout=3, KS=7, bias=True, act=1),
    dict(N=2, H=4, W=4, segs=[(20, 0), (9, 1), (24, 0)], Cout=64, KS=3, nw=4, bias=True),
    dict(N=3, H=8, W=8, segs=[(32, 0)], Cout=65, KS=1),
    dict(N=2, H=12, W=12, segs=[(8, 0), (4, 1)], Cout=136, KS=3),
    dict(N=2, H=10, W=36, segs=[(3, 0)], Cout=16, KS=3),                       # E stem (thin-in), tile overhang in x and y
    dict(N=1, H=9, W=33, segs=[(12, 0)], Cout=16, KS=3),                       # stem with observation_stacking = 4
    dict(N=2, H=8, W=40, segs=[(64, 0)], Cout=3, KS=3, bias=True, act=1),      # FinalBlock k3 (thin-out)
    dict(N=1, H=11, W=35, segs=[(32, 0)], Cout=3, KS=7, bias=True, act=1),     # FinalBlock k7
    dict(N=2, H=8, W=8, segs=[(48, 0)], Cout=9, KS=3),                         # shape of the broadcast-input dgrad (OUT = K + Da)
    dict(N=2, H=40, W=52, segs=[(16, 0)], Cout=16, KS=3),                      # narrow-layer wgrad kernel (K<=32), tile overhang
    dict(N=2, H=33, W=64, segs=[(64, 0)], Cout=32, KS=3),                      # narrow-layer wgrad kernel, KT=2
    dict(N=2, H=48, W=48, segs=[(16, 0)], Cout=32, KS=1),                      # 1x1 down-sample conv
    dict(N=2, H=20, W=36, segs=[(32, 0)], Cout=64, KS=3),                      # narrow-input VALU conv, 4 output groups
    dict(N=2, H=9, W=40, segs=[(24, 0)], Cout=16, KS=3, bias=True),
    dict(N=2, H=13, W=10, segs=[(64, 0), (9, 1), (40, 0)], Cout=72, KS=3),    # tile-resident wgrad: ragged tiles, 3 segments, 2 k-tiles
    dict(N=1, H=6, W=21, segs=[(80, 0)], Cout=24, KS=3),                       # tile-resident wgrad, 32-channel output variant
    dict(N=2, H=40, W=52, segs=[(16, 0)], Cout=16, KS=3),                      # narrow conv kernel <1,1> (16x16x4 MFMA), ragged tiles
    dict(N=2, H=42, W=50, segs=[(32, 0)], Cout=24, KS=3, bias=True),           # narrow conv <2,2>, channel tail, bias; dgrad runs <2,2> as 24 -> 32
    dict(N=1, H=64, W=64, segs=[(20, 0)], Cout=16, KS=3),                      # narrow conv <2,1>; dgrad <1,2>; wgrad <2,1>
    dict(N=1, H=65, W=66, segs=[(16, 0)], Cout=29, KS=3),                      # narrow wgrad <1,2>, ragged tiles, channel tail
    dict(N=2, H=48, W=52, segs=[(12, 0)], Cout=16, KS=3),                      # stacked-frame stem on k_conv_narrow<1,1> (round 5: 5..12 input channels, zero-filled K tail); dgrad / wgrad thin
    dict(N=1, H=16, W=40, segs=[(128, 0)], Cout=3, KS=3, bias=True, act=1),    # FinalBlock at 64x64 scale: dgrad = 3 -> 128 on k_conv_c4<3,4> x 2 groups
    dict(N=2, H=12, W=40, segs=[(3, 0)], Cout=16, KS=7),                       # 7x7 stem shape on k_conv_c4<7,1>
    dict(N=1, H=9, W=33, segs=[(3, 0)], Cout=40, KS=3, bias=True),             # k_conv_c4<3,2>, two output groups, channel tail
    dict(N=2, H=11, W=35, segs=[(32, 0)], Cout=3, KS=7, bias=True, act=1, precision=16),     # conv_head.hip <7>: ragged 8x16 tiles, tanh
    dict(N=1, H=17, W=40, segs=[(128, 0)], Cout=3, KS=3, bias=True, act=1, precision=16),   # conv_head.hip <3>: four 32-channel chunks, ragged 8x32 tiles
    dict(N=1, H=9, W=20, segs=[(16, 0)], Cout=3, KS=7, act=1, precision=16),                # conv_head.hip <7>: half-filled chunk (reduced model: 16 -> 3)
    dict(N=2, H=11, W=35, segs=[(32, 0)], Cout=3, KS=7, bias=True, act=1, precision=16, dgrad_precision=17, dgrad_tol=1e-4),      # ... + its dgrad on k_head_dgrad7<2> (split bf16, K = one tap row)
    dict(N=1, H=9, W=20, segs=[(16, 0)], Cout=3, KS=7, act=1, precision=16, dgrad_precision=17, dgrad_tol=1e-4),                  # k_head_dgrad7<1>
])
def test_conv(kw):
    K.conv_case(load_emu(), "cpu", **kw)


def test_thin_conv_without_aux_scratch():
    """thin kernels fall back to LDS-staged weights when the caller gives no scratch for the compact weight table"""
    K.conv_case(load_emu(), "cpu", N=1, H=9, W=33, segs=[(12, 0)], Cout=16, KS=3, use_aux=False)
    K.conv_case(load_emu(), "cpu", N=1, H=11, W=35, segs=[(32, 0)], Cout=3, KS=7, bias=True, act=1, use_aux=False)


@pytest.mark.parametrize("precision,tol", [(3, 3e-5), (2, 2e-3)])
def test_conv_split_bf16(precision, tol):
    """3-way split bf16 MFMA path must be fp32-class accurate; the 2-way split is ~2^-16."""
    K.conv_case(load_emu(), "cpu", N=2, H=8, W=9, segs=[(40, 0), (9, 1), (24, 0)], Cout=128, KS=3, nw=4, bias=True, precision=precision, tol=tol)
    K.conv_case(load_emu(), "cpu", N=1, H=12, W=12, segs=[(64, 0)], Cout=64, KS=3, precision=precision, tol=tol)


def test_pool_upsample():
    K.pool_up_case(load_emu(), "cpu")


@pytest.mark.parametrize("second,act,training", [("bn", 1, 1), ("plain", 1, 1), (None, 0, 1), (None, 1, 0), ("plain", 0, 1)])
def test_batchnorm(second, act, training):
    K.bn_case(load_emu(), "cpu", second=second, act=act, training=training)
    K.bn_case(load_emu(), "cpu", second=second, act=act, training=training, seed=1, N=2, H=9, W=8)


@pytest.mark.parametrize("second,act", [(None, 1), (None, 0), ("plain", 1), ("plain", 0)])
def test_batchnorm_fused_small(second, act):
    K.bn_case(load_emu(), "cpu", second=second, act=act, training=1, fused=True)
    K.bn_case(load_emu(), "cpu", second=second, act=act, training=1, fused=True, seed=1, N=2, H=33, W=31, Cc=13)     # > 8 pixels per thread, channel tail


def test_lstm_gates():
    K.lstm_case(load_emu(), "cpu")


def test_misc_pointwise():
    K.misc_case(load_emu(), "cpu")


def test_adam():
    K.adam_case(load_emu(), "cpu")


def test_conv_random_shapes():
    """seeded random geometry sweep over the conv launchers (forward, dgrad, wgrad, time-batched wgrad): tile overhangs, channel tails,
    1-pixel maps, every dispatch boundary between the MFMA / narrow / 3-channel / thin kernels (CONV_FUZZ_CASES=n for longer hunts)"""
    import os
    K.conv_fuzz(load_emu(), "cpu", int(os.environ.get("CONV_FUZZ_CASES", "100")), seed=1234)


def test_conv_hx_split_f16_forward_small():
    """conv_hx.hip on the simulator: partial tiles (10x20), three segments incl. a broadcast vector, bias, Cout not a multiple of the tile"""
    K.hx_conv_case(load_emu(), "cpu", N=1, H=10, W=20, segs=[(40, False), (9, True), (33, False)], Cout=72, bias=True)


def test_conv_hx_f16_range_guard_small():
    K.hx_saturation_case(load_emu(), "cpu")


def test_conv_hx_16_channel_layers_small():
    """round 5: the split-operand kernels also take the 3x3 layers with 16 channels on a side (E's first residual blocks, D's last stage: half-filled 32-channel chunk, 16 of 32
    tile columns): forward, dgrad to a 16-channel input, and the weight gradients on k_wgrad_hx"""
    lib = load_emu()
    K.hx_conv_case(lib, "cpu", N=1, H=10, W=20, segs=[(16, False)], Cout=16, bias=True)
    K.hx_conv_case(lib, "cpu", N=2, H=9, W=17, segs=[(16, False)], Cout=32, act=3, seed=1)
    K.hx_conv_case(lib, "cpu", N=1, H=10, W=20, segs=[(16, False)], Cout=32, precision=K.PREC_BF16X3, dgrad_seg=0, accumulate=True, seed=2)
    K.hx_conv_case(lib, "cpu", N=1, H=8, W=16, segs=[(32, False)], Cout=16, precision=K.PREC_BF16X3, dgrad_seg=0, seed=3)
    K.conv_case(lib, "cpu", N=2, H=7, W=18, segs=[(16, 0)], Cout=16, KS=3, wgrad_precision=17, wgrad_tol=1e-4, seed=4)
    K.conv_case(lib, "cpu", N=1, H=8, W=16, segs=[(16, 0)], Cout=32, KS=3, wgrad_precision=17, wgrad_tol=1e-4, seed=5)
    K.conv_case(lib, "cpu", N=1, H=6, W=20, segs=[(32, 0)], Cout=16, KS=3, wgrad_precision=17, wgrad_tol=1e-4, seed=6)


def test_conv_hx_narrow_output_tiles_small():
    """64- and 32-channel output tiles on 8x16-pixel tiles (under-filled launches) and their dgrad forms"""
    K.hx_conv_case(load_emu(), "cpu", N=1, H=10, W=20, segs=[(40, False)], Cout=48, bias=True)
    K.hx_conv_case(load_emu(), "cpu", N=1, H=9, W=17, segs=[(64, False)], Cout=32, act=2)
    K.hx_conv_case(load_emu(), "cpu", N=1, H=8, W=16, segs=[(40, False)], Cout=64, precision=K.PREC_BF16X3, dgrad_seg=0)


def test_conv_direct_latency_kernel_small():
    """conv_direct.hip (round 4): small assigning split-f16 launches in ONE launch -- a workgroup per 16-pixel x 16-channel tile, four waves splitting the K steps, fragments
    straight from global memory, fixed-order LDS fold, epilogue in place.  Ragged width (not a multiple of 16), channel tails on both sides, a broadcast segment, a single
    32-channel chunk (nine steps over four waves), bias / LeakyReLU / ReLU / residual epilogues, BatchNorm-folded weights (PackDesc.oscale)"""
    lib = load_emu()
    K.hx_conv_case(lib, "cpu", N=1, H=10, W=20, segs=[(40, False)], Cout=48, bias=True, direct=True)
    K.hx_conv_case(lib, "cpu", N=2, H=8, W=18, segs=[(64, False), (9, True)], Cout=65, bias=True, act=3, res=True, direct=True, seed=1)
    K.hx_conv_case(lib, "cpu", N=1, H=9, W=32, segs=[(32, False)], Cout=32, act=2, oscale=True, direct=True, seed=2)
    K.hx_conv_case(lib, "cpu", N=1, H=6, W=16, segs=[(64, False), (5, True), (64, False)], Cout=128, bias=True, act=3, direct=True, seed=3)
    K.hx_conv_case(lib, "cpu", N=1, H=4, W=16, segs=[(128, False), (9, True), (128, False)], Cout=40, bias=True, direct=True, seed=4)      # 81 steps
    K.hx_conv_case(lib, "cpu", N=1, H=8, W=16, segs=[(32, False)], Cout=64, bias=True, act=3, oscale=True, direct=True, avgpool=True, seed=5)      # round 5: avg_pool2d(2) in the epilogue
    K.hx_conv_case(lib, "cpu", N=2, H=6, W=26, segs=[(40, False)], Cout=20, bias=True, act=3, res=True, direct=True, avgpool=True, seed=6)         # ragged 8-pixel groups, tails
    K.hx_conv_case(lib, "cpu", N=1, H=16, W=48, segs=[(128, False), (9, True)], Cout=200, bias=True, act=3, direct="tile4", split=True, avgpool=True, seed=7)      # too large for the latency kernel: K-split tile launch, the slab reduce pools


def test_conv_hx_4x16_tiles_for_inference_small():
    """round 4: inference launches (ConvArgs.direct_ok) too large for the latency kernel and under-filled on 8x16 tiles run on 4x16-pixel tiles: with and without the K split
    into slabs, ragged rows, a channel tail"""
    lib = load_emu()
    K.hx_conv_case(lib, "cpu", N=1, H=14, W=64, segs=[(256, False)], Cout=128, bias=True, act=3, direct="tile4", split=True)
    K.hx_conv_case(lib, "cpu", N=1, H=18, W=60, segs=[(200, False), (9, True)], Cout=100, bias=True, res=True, direct="tile4", seed=1)


def test_lstm_cell_update_in_the_slab_reduce_small():
    K.lstm_fused_reduce_case(load_emu(), "cpu", N=1, H=8, W=16, Cin=64, Cc=32)


def test_conv_hx_8wave_pipelined_variant_small():
    """the 16x16x128 tile on 8 waves with the 3-deep weight-tile register ring: ragged 20x18 map, K tail (2 chunks + segment padding), Cout tail"""
    K.hx_conv_case(load_emu(), "cpu", N=1, H=20, W=18, segs=[(40, False), (5, True)], Cout=130, bias=True, act=2, big=1)


def test_conv_hx_fused_maxpool_epilogue_small():
    """MaxPool2d(2, 2) of the ReLU output written by the conv epilogue (VGG19 layers in front of a pool; the two tile variants that carry that
    epilogue, forced here): odd map sizes (floor), two channel blocks, and the write-less form of the ground-truth branch"""
    lib = load_emu()
    K.hx_conv_case(lib, "cpu", N=2, H=19, W=21, segs=[(40, False)], Cout=48, bias=True, act=2, pool=True, big=1)                # 16x16x64 tiles, odd H and W
    K.hx_conv_case(lib, "cpu", N=1, H=20, W=36, segs=[(64, False)], Cout=130, bias=True, act=2, pool=True, big=1)               # 8-wave 16x16x128, two channel blocks
    K.hx_conv_case(lib, "cpu", N=1, H=16, W=18, segs=[(64, False)], Cout=128, bias=True, act=2, pool=True, skip_out=True, big=1)
    K.hx_conv_case(lib, "cpu", N=1, H=34, W=16, segs=[(32, False)], Cout=64, act=2, pool=True, skip_out=True, big=1)


def test_conv_hx_dgrad_mask_seed_epilogue_small():
    """dgrad form on split bf16 with the fused ReLU mask + L1 seed epilogue (VGG19 perceptual loss) and the accumulate / split-K variants"""
    K.hx_conv_case(load_emu(), "cpu", N=1, H=8, W=16, segs=[(64, False)], Cout=32, precision=K.PREC_BF16X3, dgrad_seg=0, mask=True, seed_w=3e-7)
    K.hx_conv_case(load_emu(), "cpu", N=1, H=8, W=16, segs=[(160, False)], Cout=32, precision=K.PREC_BF16X3, dgrad_seg=0, accumulate=True)


def test_folded_inference_epilogues_small():
    """roll-out epilogues (eval-mode BatchNorm folded into the conv): PackDesc.oscale in both pack kernels, ConvArgs.res + LeakyReLU(0.2) (act 3) in
    k_conv_fwd, its split-K reduce, k_conv_narrow and k_conv_hx (direct + slab split-K)"""
    lib = load_emu()
    K.conv_case(lib, "cpu", N=1, H=9, W=12, segs=[(48, 0)], Cout=65, KS=3, bias=True, act=3, res=True, oscale=True, check_bwd=False)       # generic, under-filled -> slabs + reduce
    K.conv_case(lib, "cpu", N=2, H=40, W=52, segs=[(16, 0)], Cout=16, KS=3, bias=True, act=3, res=True, oscale=True, check_bwd=False)     # k_conv_narrow
    K.conv_case(lib, "cpu", N=2, H=24, W=24, segs=[(16, 0)], Cout=32, KS=1, bias=True, oscale=True, check_bwd=False)                      # 1x1 down-sample conv, folded, no activation
    K.conv_case(lib, "cpu", N=1, H=9, W=33, segs=[(12, 0)], Cout=16, KS=3, bias=True, oscale=True, check_bwd=False)                       # stacked-frame stem (thin-in kernel) keeps bias + scale
    # round 5: avg_pool2d(2) inside k_conv_narrow's epilogue (conv -> pool -> affine -> LeakyReLU of the folded stem / down-sampling residual blocks); ragged tiles, odd pooled width
    K.conv_case(lib, "cpu", N=2, H=40, W=52, segs=[(12, 0)], Cout=16, KS=3, bias=True, act=3, oscale=True, check_bwd=False, avgpool=True)
    K.conv_case(lib, "cpu", N=1, H=64, W=66, segs=[(16, 0)], Cout=32, KS=3, bias=True, act=3, res=True, oscale=True, check_bwd=False, avgpool=True)
    K.conv_case(lib, "cpu", N=1, H=66, W=64, segs=[(32, 0)], Cout=29, KS=3, check_bwd=False, avgpool=True)
    # latency 1x1 kernel of inference passes (k_conv1x1_lat): plain, average-pooled (the down-sampling paths), channel tails on both sides
    K.conv_case(lib, "cpu", N=1, H=12, W=20, segs=[(16, 0)], Cout=32, KS=1, bias=True, oscale=True, check_bwd=False, direct_ok=True, avgpool=True)
    K.conv_case(lib, "cpu", N=2, H=9, W=11, segs=[(64, 0)], Cout=65, KS=1, bias=True, oscale=True, check_bwd=False, direct_ok=True)
    K.conv_case(lib, "cpu", N=1, H=8, W=10, segs=[(30, 0)], Cout=19, KS=1, act=3, res=True, check_bwd=False, direct_ok=True, avgpool=True)
    K.hx_conv_case(lib, "cpu", N=1, H=10, W=20, segs=[(40, False)], Cout=48, bias=True, act=3, res=True, oscale=True)
    K.hx_conv_case(lib, "cpu", N=1, H=8, W=16, segs=[(96, False)], Cout=64, bias=True, act=3, res=True, oscale=True, split=True)


def test_batchnorm_fused_into_convolutions_small():
    """round 3: BatchNorm statistics from the producing conv's epilogue, BatchNorm + LeakyReLU applied by the consuming conv / weight gradient while
    staging (never materialised), backward with the slope recomputed from the raw tensor; ragged tiles, channel tails, a broadcast segment,
    the 8-wave tile variant, and per-group tables (time-batched launches)"""
    lib = load_emu()
    K.hx_lazy_bn_case(lib, "cpu", N=2, H=10, W=20, Cin=40, Cout=48, aux_c=9)
    K.hx_lazy_bn_case(lib, "cpu", N=1, H=9, W=17, Cin=64, Cout=33, act=0, seed=1)
    K.hx_lazy_bn_case(lib, "cpu", N=1, H=20, W=18, Cin=40, Cout=130, big=1, seed=2)
    K.hx_lazy_bn_case(lib, "cpu", N=4, H=8, W=16, Cin=32, Cout=64, groups=2, seed=3)
    K.hx_lazy_bn_case(lib, "cpu", N=2, H=8, W=16, Cin=96, Cout=64, split=True, seed=4)       # under-filled: split K, statistics from the slab reduce


def test_wgrad_hx_split_bf16_small():
    """k_wgrad_hx on the simulator (ds_read_b64_tr_b16 transposing fragment reads): ragged tiles, three segments incl. a broadcast vector,
    output-channel tail; the forward / dgrad of the same case run on the exact kernels"""
    K.conv_case(load_emu(), "cpu", N=2, H=13, W=10, segs=[(64, 0), (9, 1), (40, 0)], Cout=72, KS=3, wgrad_precision=17, wgrad_tol=1e-4)
    # <= 32 output channels: the two wave rows split the tile's pixel rows (round 4); ragged rows (H = 6, 7: the second half-tile partly / wholly outside)
    K.conv_case(load_emu(), "cpu", N=2, H=7, W=18, segs=[(40, 0)], Cout=32, KS=3, wgrad_precision=17, wgrad_tol=1e-4)
    K.conv_case(load_emu(), "cpu", N=1, H=6, W=16, segs=[(64, 0), (5, 1)], Cout=32, KS=3, wgrad_precision=17, wgrad_tol=1e-4, seed=3)


def test_conv_hx_s16_tensors_small():
    """round 5: activations exchanged pre-split between k_conv_hx launches (ConvArgs.out_s16 / pool_s16 -> in_s16): bit-identical to the fp32 exchange, forward and dgrad chain"""
    K.hx_s16_chain_case(load_emu(), "cpu", N=1, H=16, W=16, C0=64, C1=128, C2=64)
    K.hx_s16_chain_case(load_emu(), "cpu", N=1, H=18, W=20, C0=64, C1=64, C2=128, pool=True)


def test_streaming_weight_gradients_small():
    """round 5 (conv_stream.hip): the 7x7 FinalBlock head's weight gradient on the split-bf16 matrix pipe with the taps on the M side (k_wgrad_head7: ragged tiles, 16 / 32 input
    channels, time-batched groups, bit-reproducible slabs -- conv_case runs all of them) and the 1x1 identity-path weight gradients straight from global memory (k_wgrad_1x1)"""
    lib = load_emu()
    for kw in (dict(N=1, H=11, W=35, segs=[(32, 0)], Cout=3, KS=7, bias=True, act=1), dict(N=2, H=8, W=70, segs=[(16, 0)], Cout=3, KS=7, bias=True, act=1),
               dict(N=3, H=6, W=9, segs=[(32, 0)], Cout=3, KS=7)):
        K.conv_case(lib, "cpu", wgrad_precision=17, wgrad_tol=1e-4, dgrad_precision=17, dgrad_tol=1e-4, **kw)
    for kw in (dict(N=2, H=20, W=24, segs=[(64, 0)], Cout=128, KS=1), dict(N=3, H=17, W=9, segs=[(32, 0)], Cout=65, KS=1), dict(N=2, H=40, W=40, segs=[(16, 0)], Cout=32, KS=1)):
        K.conv_case(lib, "cpu", **kw)


def test_s16_gradients_small():
    """round 6: the gradient of a conv output exchanged pre-split (S16-bf16): producers write exactly the halves the loaders would form, weight gradients / dgrads (4 x 16, 8 x 16 and
    16 x 16-pixel tile variants, 32 / 64 / 128 output channels, K split) are bit-identical to those of the fp32 tensor, border / column sums see hi + lo"""
    lib = load_emu()
    K.s16_grad_case(lib, "cpu", N=2, H=8, W=16, Cin=64, Cout=128)                       # 4 x 16 tiles (under-filled split-bf16 launch), K split
    K.s16_grad_case(lib, "cpu", N=1, H=9, W=18, Cin=40, Cout=64, seed=1)                # ragged tiles, channel tail on the dgrad's output side
    K.s16_grad_case(lib, "cpu", N=2, H=6, W=16, Cin=32, Cout=32, seed=2)                # <= 32 output channels of the weight gradient: row-split waves
    K.s16_grad_case(lib, "cpu", N=1, H=16, W=16, Cin=128, Cout=64, seed=3, force_big=1, producers=False)      # 8-wave 16 x 16 x 128 tile


def test_conv_hx_register_weight_variants_small():
    """round 6: k_conv_hx<BG> -- 8 x 16 x 64 / 8 x 16 x 32 / 4 x 16 x 64-pixel tile variants with the weights straight into a register ring, K halves per wave pair, double-buffered
    halo: ragged tiles, three segments incl. a broadcast vector, channel tails, accumulating dgrads (deterministic K split and whole K), pre-split dY, an odd number of chunks"""
    K.hx_register_weights_case(load_emu(), "cpu", [
        ("conv", dict(N=1, H=10, W=20, segs=[(40, False), (9, True), (33, False)], Cout=48, bias=True)),                      # 8 x 16 x 64, 4 chunks, three segments
        ("conv", dict(N=1, H=9, W=17, segs=[(96, False)], Cout=32, act=3, seed=1)),                                             # 8 x 16 x 32 (four row-pair waves), 3 chunks
        ("conv", dict(N=1, H=8, W=16, segs=[(40, False)], Cout=64, precision=K.PREC_BF16X3, dgrad_seg=0, accumulate=True, seed=2)),   # split-bf16 dgrad on 4 x 16 tiles
        ("conv", dict(N=1, H=14, W=64, segs=[(256, False)], Cout=128, bias=True, act=3, direct="tile4", split=True, seed=3)),   # inference 4 x 16 tiles, K split into slabs
        ("s16", dict(N=2, H=8, W=16, Cin=64, Cout=160, seed=4, producers=False)),                                                # pre-split dY, 5 chunks
    ])
