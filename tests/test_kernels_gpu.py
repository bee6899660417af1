"""Per-kernel parity on the MI355X: the same cases as tests/test_kernels_emu.py, through the C-ABI of libcaddy_hip.so."""
import pytest
import torch

from tests import kernel_cases as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from playablevideogeneration_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _lib.load()


@pytest.mark.parametrize("kw", [
    dict(N=2, H=6, W=5, segs=[(16, 0)], Cout=32, KS=3),
    dict(N=1, H=9, W=7, segs=[(5, 0)], Cout=3, KS=7, bias=True, act=1),
    dict(N=2, H=4, W=4, segs=[(20, 0), (9, 1), (24, 0)], Cout=64, KS=3, nw=4, bias=True),
    dict(N=3, H=8, W=8, segs=[(32, 0)], Cout=65, KS=1),
    dict(N=2, H=12, W=12, segs=[(8, 0), (4, 1)], Cout=136, KS=3),
    dict(N=4, H=32, W=32, segs=[(64, 0), (9, 1), (128, 0)], Cout=512, KS=3, nw=4, bias=True, tol=5e-5),   # BAIR LSTM0 shape
    dict(N=2, H=64, W=64, segs=[(32, 0)], Cout=3, KS=7, bias=True, act=1, tol=5e-5),
    dict(N=2, H=48, W=40, segs=[(64, 0)], Cout=32, KS=3, tol=5e-5),
    dict(N=2, H=10, W=36, segs=[(3, 0)], Cout=16, KS=3),
    dict(N=1, H=9, W=33, segs=[(12, 0)], Cout=16, KS=3),
    dict(N=2, H=8, W=40, segs=[(64, 0)], Cout=3, KS=3, bias=True, act=1),
    dict(N=1, H=11, W=35, segs=[(32, 0)], Cout=3, KS=7, bias=True, act=1),
    dict(N=2, H=8, W=8, segs=[(48, 0)], Cout=9, KS=3),
    dict(N=4, H=64, W=64, segs=[(3, 0)], Cout=16, KS=3, tol=5e-5),
    dict(N=2, H=40, W=52, segs=[(16, 0)], Cout=16, KS=3),
    dict(N=2, H=33, W=64, segs=[(64, 0)], Cout=32, KS=3, tol=5e-5),
    dict(N=2, H=48, W=48, segs=[(16, 0)], Cout=32, KS=1),
    dict(N=2, H=20, W=36, segs=[(32, 0)], Cout=64, KS=3),
    dict(N=2, H=9, W=40, segs=[(24, 0)], Cout=16, KS=3, bias=True),
    dict(N=2, H=13, W=10, segs=[(64, 0), (9, 1), (40, 0)], Cout=72, KS=3),    # tile-resident wgrad: ragged tiles, 3 segments, 2 k-tiles
    dict(N=1, H=6, W=21, segs=[(80, 0)], Cout=24, KS=3),                       # tile-resident wgrad, 32-channel output variant
    dict(N=2, H=40, W=52, segs=[(16, 0)], Cout=16, KS=3),                      # narrow conv kernel <1,1> (16x16x4 MFMA), ragged tiles
    dict(N=2, H=42, W=50, segs=[(32, 0)], Cout=24, KS=3, bias=True),           # narrow conv <2,2>, channel tail, bias; dgrad runs <2,2> as 24 -> 32
    dict(N=2, H=64, W=80, segs=[(32, 0)], Cout=3, KS=7, bias=True, act=1, precision=16, tol=5e-5),      # conv_head.hip <7> (FinalBlock 7x7 on the 16 x 16 x 32 matrix instruction)
    dict(N=2, H=33, W=75, segs=[(64, 0)], Cout=3, KS=3, bias=True, act=1, precision=16, tol=5e-5),      # conv_head.hip <3>, two chunks, ragged tiles
    dict(N=1, H=17, W=40, segs=[(128, 0)], Cout=3, KS=3, bias=True, act=1, precision=16, tol=5e-5),     # conv_head.hip <3>, four chunks
    dict(N=1, H=9, W=20, segs=[(16, 0)], Cout=3, KS=7, act=1, precision=16),                            # conv_head.hip <7>, half-filled chunk
    dict(N=2, H=64, W=80, segs=[(32, 0)], Cout=3, KS=7, bias=True, act=1, precision=16, tol=5e-5, dgrad_precision=17, dgrad_tol=1e-4),      # ... + dgrad on k_head_dgrad7<2> (split bf16)
    dict(N=1, H=37, W=45, segs=[(16, 0)], Cout=3, KS=7, act=1, precision=16, dgrad_precision=17, dgrad_tol=1e-4),                            # k_head_dgrad7<1>, ragged tiles
    dict(N=8, H=128, W=160, segs=[(32, 0)], Cout=3, KS=7, bias=True, act=1, precision=16, tol=5e-5),     # conv_head.hip <7>, eight frames of a training step
    dict(N=1, H=64, W=64, segs=[(20, 0)], Cout=16, KS=3),                      # narrow conv <2,1>; dgrad <1,2>; wgrad <2,1>
    dict(N=1, H=65, W=66, segs=[(16, 0)], Cout=29, KS=3),                      # narrow wgrad <1,2>, ragged tiles, channel tail
    dict(N=1, H=16, W=40, segs=[(128, 0)], Cout=3, KS=3, bias=True, act=1),    # FinalBlock at 64x64 scale: dgrad = 3 -> 128 on k_conv_c4<3,4> x 2 groups
    dict(N=2, H=12, W=40, segs=[(3, 0)], Cout=16, KS=7),                       # 7x7 stem shape on k_conv_c4<7,1>
    dict(N=1, H=9, W=33, segs=[(3, 0)], Cout=40, KS=3, bias=True),             # k_conv_c4<3,2>, two output groups, channel tail
    dict(N=8, H=32, W=32, segs=[(128, 0), (9, 1), (128, 0)], Cout=512, KS=3, tol=1e-4),   # ConvLSTM gates at R's first resolution (persistent tiles)
])
def test_conv(lib, kw):
    K.conv_case(lib, "cuda", **kw)


def test_conv_random_shapes(lib):
    """the simulator's seeded geometry sweep, on the hardware (real MFMA fragment maps, atomics, LDS)"""
    K.conv_fuzz(lib, "cuda", 150, seed=4321)


def test_thin_conv_without_aux_scratch(lib):
    K.conv_case(lib, "cuda", N=1, H=9, W=33, segs=[(12, 0)], Cout=16, KS=3, use_aux=False)
    K.conv_case(lib, "cuda", N=1, H=11, W=35, segs=[(32, 0)], Cout=3, KS=7, bias=True, act=1, use_aux=False)


def test_pool_upsample(lib):
    K.pool_up_case(lib, "cuda")
    K.pool_up_case(lib, "cuda", N=3, Cc=64, H=16, W=24, seed=1)


@pytest.mark.parametrize("second,act,training", [("bn", 1, 1), ("plain", 1, 1), (None, 0, 1), (None, 1, 0), ("plain", 0, 1)])
def test_batchnorm(lib, second, act, training):
    K.bn_case(lib, "cuda", second=second, act=act, training=training)
    K.bn_case(lib, "cuda", N=4, Cc=65, H=32, W=32, second=second, act=act, training=training, seed=2)


@pytest.mark.parametrize("second,act", [(None, 1), (None, 0), ("plain", 1), ("plain", 0)])
def test_batchnorm_fused_small(lib, second, act):
    K.bn_case(lib, "cuda", second=second, act=act, training=1, fused=True)
    K.bn_case(lib, "cuda", second=second, act=act, training=1, fused=True, seed=1, N=2, H=33, W=31, Cc=13)     # > 8 pixels per thread, channel tail


def test_lstm_gates(lib):
    K.lstm_case(lib, "cuda")
    K.lstm_case(lib, "cuda", N=4, Cc=128, H=16, W=16, seed=3)


def test_misc_pointwise(lib):
    K.misc_case(lib, "cuda")


def test_adam(lib):
    K.adam_case(lib, "cuda", n=100003)


@pytest.mark.parametrize("kw", [
    dict(N=8, H=32, W=32, segs=[(64, False), (9, True), (128, False)], Cout=512, bias=True),            # R: ConvLSTM0 gates
    dict(N=2, H=16, W=16, segs=[(256, False), (9, True), (256, False)], Cout=1024, bias=True, split=True),   # ConvLSTM1 gates, slab split-K
    dict(N=3, H=26, W=20, segs=[(128, False), (4, True)], Cout=128),                                   # Breakout state resolution: ragged tiles
    dict(N=2, H=128, W=128, segs=[(64, False)], Cout=64),                                              # D: 16x16 x 64-channel tiles
    dict(N=1, H=256, W=256, segs=[(64, False)], Cout=32),                                              # D: last UpBlock conv
    dict(N=2, H=64, W=64, segs=[(128, False)], Cout=128, bias=True, act=2),                            # VGG19 conv + bias + ReLU
    dict(N=2, H=64, W=64, segs=[(128, False)], Cout=128, bias=True, act=2, precision=18),              # ... single-product f16
    dict(N=2, H=40, W=52, segs=[(33, False)], Cout=65),                                                # channel tails on both sides
    dict(N=60, H=32, W=32, segs=[(512, False)], Cout=512, bias=True, act=2),                           # VGG19 conv4_x: 8-wave variant, 16 chunks
    dict(N=3, H=40, W=52, segs=[(128, False), (9, True)], Cout=130, big=1),                            # 8-wave variant, ragged tiles, tails
    dict(N=8, H=64, W=64, segs=[(128, False)], Cout=128, big=1),
    dict(N=8, H=32, W=32, segs=[(64, False)], Cout=64),                                                # E / A on one time step: 8x16 x 64-channel tiles
    dict(N=8, H=64, W=64, segs=[(32, False)], Cout=32, bias=True),                                     # 8x16 x 32-channel tiles
    dict(N=8, H=32, W=32, segs=[(64, False)], Cout=65),
    dict(N=16, H=256, W=256, segs=[(64, False)], Cout=64, bias=True, act=2),                             # VGG19 conv1_2
    dict(N=8, H=250, W=256, segs=[(64, False)], Cout=32, bias=True),                                    # D's last UpBlock: 32 channels, ragged rows
])
def test_conv_hx_forward(lib, kw):
    K.hx_conv_case(lib, "cuda", **kw)


def test_conv_hx_16_channel_layers(lib):
    """round 5: 3x3 layers with 16 channels on a side on the split-operand kernels (E's first residual blocks on all B x T frames, D's last stage): forward, dgrad, weight gradients"""
    K.hx_conv_case(lib, "cuda", N=8, H=128, W=128, segs=[(16, False)], Cout=16, bias=True)
    K.hx_conv_case(lib, "cuda", N=8, H=128, W=128, segs=[(16, False)], Cout=32, act=3, seed=1)
    K.hx_conv_case(lib, "cuda", N=8, H=128, W=128, segs=[(16, False)], Cout=32, precision=K.PREC_BF16X3, dgrad_seg=0, accumulate=True, seed=2)
    K.hx_conv_case(lib, "cuda", N=2, H=66, W=70, segs=[(32, False)], Cout=16, precision=K.PREC_BF16X3, dgrad_seg=0, seed=3)
    K.conv_case(lib, "cuda", N=8, H=64, W=64, segs=[(16, 0)], Cout=16, KS=3, wgrad_precision=17, wgrad_tol=1e-4, seed=4)
    K.conv_case(lib, "cuda", N=4, H=64, W=64, segs=[(16, 0)], Cout=32, KS=3, wgrad_precision=17, wgrad_tol=1e-4, seed=5)
    K.conv_case(lib, "cuda", N=2, H=33, W=40, segs=[(32, 0)], Cout=16, KS=3, wgrad_precision=17, wgrad_tol=1e-4, seed=6)


@pytest.mark.parametrize("kw", [
    dict(N=1, H=32, W=32, segs=[(64, False)], Cout=64, bias=True, act=3, res=True),                    # E residual block conv of a roll-out frame (BatchNorm folded)
    dict(N=1, H=32, W=32, segs=[(64, False), (9, True)], Cout=65, bias=True),                          # channel tails + broadcast action input
    dict(N=1, H=64, W=64, segs=[(32, False)], Cout=64, bias=True, act=3),                              # one 32-channel chunk: nine steps over four waves
    dict(N=1, H=16, W=16, segs=[(128, False), (9, True), (128, False)], Cout=128, bias=True, act=3),   # R's side branch on the 16x16 map
    dict(N=2, H=20, W=26, segs=[(40, False)], Cout=48, act=2, oscale=True, seed=3),                    # ragged width, tails
    dict(N=1, H=64, W=64, segs=[(32, False)], Cout=64, bias=True, act=3, oscale=True, avgpool=True),   # round 5: E's second down-sampling block: conv -> avg-pool -> affine -> LeakyReLU in one launch
    dict(N=2, H=20, W=26, segs=[(40, False)], Cout=48, bias=True, act=3, res=True, avgpool=True, seed=5),   # pooled, ragged 8-pixel groups (pooled width 13), residual at the pooled size
])
def test_conv_direct_latency_kernel(lib, kw):
    """round 4: conv_direct.hip -- the one-launch form of small assigning split-f16 convolutions (batch-1 roll-out layers)"""
    K.hx_conv_case(lib, "cuda", direct=True, **kw)


def test_lstm_cell_update_in_the_slab_reduce(lib):
    """round 4: roll-out ConvLSTM cells -- the K-split gate convolution's slab reduce applies the cell update (ConvArgs.lstm) like k_split_reduce + k_map<FLstmFwd>"""
    K.lstm_fused_reduce_case(lib, "cuda", N=1, H=32, W=32, Cin=208, Cc=128)
    K.lstm_fused_reduce_case(lib, "cuda", N=1, H=16, W=16, Cin=528, Cc=256, seed=1)


@pytest.mark.parametrize("kw", [
    dict(N=1, H=128, W=128, segs=[(64, False)], Cout=64, bias=True, act=3),                  # D's 128x128 layers of a roll-out frame: 256 workgroups of 4x16 pixels, no K split
    dict(N=1, H=126, W=120, segs=[(128, False)], Cout=64, bias=True, act=3, res=True, seed=1),      # ragged rows / columns, 36 steps
    dict(N=1, H=64, W=64, segs=[(128, False)], Cout=128, bias=True, act=3, split=True, seed=2),      # still under-filled on 4x16 tiles: K split into slabs + fixed-order reduce
    dict(N=1, H=32, W=32, segs=[(128, False), (12, True)], Cout=256, bias=True, act=3, split=True, avgpool=True, oscale=True, seed=3),      # round 5: R's first block of a roll-out frame -- the slab reduce of the K-split launch also pools
    dict(N=1, H=18, W=52, segs=[(160, False)], Cout=200, bias=True, act=3, res=True, split=True, avgpool=True, seed=4),
])
def test_conv_hx_4x16_tiles_for_inference(lib, kw):
    """round 4: inference launches too large for the latency kernel whose 8x16 grid would be split over K run on 4x16-pixel tiles instead (no slab reduce)"""
    K.hx_conv_case(lib, "cuda", direct="tile4", **kw)


@pytest.mark.parametrize("kw", [
    dict(N=8, H=64, W=64, Cin=128, Cout=128),                      # D residual block: conv1 -> bn1 -> LeakyReLU -> conv2, 8x16x128 tiles
    dict(N=8, H=128, W=128, Cin=64, Cout=64, seed=1),              # 16x16x64 tiles
    dict(N=8, H=32, W=32, Cin=128, Cout=256, aux_c=9, act=0),      # ConvLSTM 0's BatchNorm (no activation) -> SameBlock conv with the broadcast action input
    dict(N=3, H=26, W=20, Cin=64, Cout=65, seed=2),                # Breakout state resolution: ragged tiles, channel tail
    dict(N=40, H=64, W=64, Cin=128, Cout=128, groups=5, seed=3),   # five time steps in one launch: per-step statistics, 8-wave tiles
    dict(N=16, H=32, W=32, Cin=64, Cout=64, groups=2, big=0, seed=4),
    dict(N=8, H=16, W=16, Cin=256, Cout=128, aux_c=9, split=True, seed=5),      # R's middle block on 16x16 maps: split K, statistics from the slab reduce
    dict(N=8, H=32, W=32, Cin=64, Cout=64, split=True, seed=6),                 # E on one time step's frames, 32x32
])
def test_batchnorm_fused_into_convolutions(lib, kw):
    K.hx_lazy_bn_case(lib, "cuda", **kw)


def test_conv_hx_f16_range_guard(lib):
    """activations beyond the f16 range (1e5-scale) in a split-f16 forward convolution: clamped while staged, finite output, flag word set"""
    K.hx_saturation_case(lib, "cuda")
    K.hx_saturation_case(lib, "cuda", N=2, H=40, W=52, Cin=64, Cout=128, seed=1)


def test_conv_hx_fused_maxpool_epilogue(lib):
    """MaxPool2d(2, 2) written by the conv epilogue at VGG19 shapes (conv1_2 / conv3_4 / conv4_4), incl. the write-less ground-truth form and an odd map"""
    K.hx_conv_case(lib, "cuda", N=4, H=256, W=256, segs=[(64, False)], Cout=64, bias=True, act=2, pool=True)
    K.hx_conv_case(lib, "cuda", N=12, H=64, W=64, segs=[(256, False)], Cout=256, bias=True, act=2, pool=True, skip_out=True)
    K.hx_conv_case(lib, "cuda", N=24, H=32, W=32, segs=[(512, False)], Cout=512, bias=True, act=2, pool=True)
    K.hx_conv_case(lib, "cuda", N=40, H=41, W=53, segs=[(128, False)], Cout=128, bias=True, act=2, pool=True)


def test_folded_inference_epilogues(lib):
    """roll-out epilogues (eval-mode BatchNorm folded into the conv): PackDesc.oscale, ConvArgs.res + LeakyReLU(0.2) at the batch-1 Tennis shapes"""
    K.conv_case(lib, "cuda", N=1, H=32, W=32, segs=[(64, 0)], Cout=65, KS=3, bias=True, act=3, res=True, oscale=True, check_bwd=False)        # E's last block: generic kernel, slabs + reduce
    K.conv_case(lib, "cuda", N=1, H=128, W=128, segs=[(16, 0)], Cout=16, KS=3, bias=True, act=3, res=True, oscale=True, check_bwd=False)     # k_conv_narrow
    K.conv_case(lib, "cuda", N=1, H=128, W=128, segs=[(16, 0)], Cout=32, KS=1, bias=True, oscale=True, check_bwd=False)                      # 1x1 down-sample
    K.conv_case(lib, "cuda", N=1, H=256, W=256, segs=[(12, 0)], Cout=16, KS=3, bias=True, oscale=True, check_bwd=False)                      # stacked-frame stem
    # round 5: avg_pool2d(2) + bias + LeakyReLU in k_conv_narrow's epilogue: the folded stem and the first down-sampling residual block of a roll-out frame
    K.conv_case(lib, "cuda", N=1, H=256, W=256, segs=[(12, 0)], Cout=16, KS=3, bias=True, act=3, oscale=True, check_bwd=False, avgpool=True)
    K.conv_case(lib, "cuda", N=1, H=128, W=128, segs=[(16, 0)], Cout=32, KS=3, bias=True, act=3, oscale=True, check_bwd=False, avgpool=True)
    K.conv_case(lib, "cuda", N=2, H=66, W=68, segs=[(32, 0)], Cout=29, KS=3, res=True, check_bwd=False, avgpool=True)
    K.conv_case(lib, "cuda", N=8, H=64, W=64, segs=[(12, 0)], Cout=16, KS=3)                                                                   # the stem's training launch (forward on k_conv_narrow, gradients thin)
    # the down-sampling paths' 1x1 convolutions on the latency kernel (k_conv1x1_lat), average-pooled where the block down-samples
    K.conv_case(lib, "cuda", N=1, H=128, W=128, segs=[(16, 0)], Cout=32, KS=1, bias=True, oscale=True, check_bwd=False, direct_ok=True, avgpool=True)
    K.conv_case(lib, "cuda", N=1, H=64, W=64, segs=[(32, 0)], Cout=64, KS=1, bias=True, oscale=True, check_bwd=False, direct_ok=True, avgpool=True)
    K.conv_case(lib, "cuda", N=1, H=32, W=32, segs=[(64, 0)], Cout=65, KS=1, bias=True, oscale=True, check_bwd=False, direct_ok=True)
    K.conv_case(lib, "cuda", N=2, H=18, W=22, segs=[(30, 0)], Cout=19, KS=1, act=3, res=True, check_bwd=False, direct_ok=True, avgpool=True)
    K.hx_conv_case(lib, "cuda", N=1, H=64, W=64, segs=[(128, False)], Cout=128, bias=True, act=3, res=True, oscale=True, split=True)         # D residual block, slab split-K
    K.hx_conv_case(lib, "cuda", N=1, H=256, W=256, segs=[(64, False)], Cout=32, bias=True, act=3, oscale=True)                               # D last UpBlock
    K.hx_conv_case(lib, "cuda", N=1, H=16, W=16, segs=[(256, False), (12, True)], Cout=128, bias=True, act=3, oscale=True, split=True)       # R's middle block


@pytest.mark.parametrize("kw", [
    dict(N=8, H=32, W=32, segs=[(64, False), (9, True), (128, False)], Cout=512, dgrad_seg=2, accumulate=True),    # dgrad to h_prev (+=), atomics split-K
    dict(N=8, H=32, W=32, segs=[(64, False), (9, True), (128, False)], Cout=512, dgrad_seg=0),
    dict(N=2, H=64, W=64, segs=[(256, False)], Cout=256, dgrad_seg=0, mask=True),                                  # VGG19 dgrad + ReLU mask
    dict(N=2, H=128, W=128, segs=[(64, False)], Cout=128, dgrad_seg=0, mask=True, seed_w=2e-7),                    # ... + L1 seed of a tapped map
    dict(N=2, H=64, W=64, segs=[(256, False)], Cout=256, dgrad_seg=0, mask=True, precision=19),
    dict(N=4, H=16, W=16, segs=[(256, False)], Cout=128, dgrad_seg=0, split=True),                                 # assigning dgrad, slab split-K
    dict(N=8, H=64, W=64, segs=[(64, False)], Cout=64, dgrad_seg=0, accumulate=True),                              # 256 workgroups of 8x16 pixels (smaller grids take the 4x16 tiles since round 4)
])
def test_conv_hx_dgrad(lib, kw):
    kw = dict(kw)
    kw.setdefault("precision", K.PREC_BF16X3)
    K.hx_conv_case(lib, "cuda", **kw)


@pytest.mark.parametrize("kw", [
    dict(N=8, H=32, W=32, segs=[(64, 0), (9, 1), (128, 0)], Cout=512, KS=3, nw=4, bias=True),      # ConvLSTM0 gates
    dict(N=3, H=26, W=20, segs=[(128, 0), (4, 1)], Cout=128, KS=3),                                 # Breakout state resolution
    dict(N=2, H=64, W=64, segs=[(128, 0)], Cout=128, KS=3),                                         # D residual block
    dict(N=2, H=128, W=128, segs=[(64, 0)], Cout=32, KS=3),                                         # D last UpBlock conv
    dict(N=2, H=40, W=52, segs=[(33, 0)], Cout=65, KS=3),                                           # channel tails
    dict(N=3, H=38, W=52, segs=[(32, 0)], Cout=32, KS=3),                                           # 32 -> 32 (row-split waves), ragged
])
def test_wgrad_hx(lib, kw):
    K.conv_case(lib, "cuda", wgrad_precision=17, wgrad_tol=1e-4, **kw)


@pytest.mark.parametrize("kw", [
    dict(N=1, H=16, W=16, C0=64, C1=128, C2=64),                 # 8-wave 16x16x128 producer, 16x16x64 consumer
    dict(N=2, H=32, W=48, C0=64, C1=64, C2=128, pool=True),      # fused 2x2 max-pool written as S16
    dict(N=2, H=40, W=36, C0=128, C1=256, C2=256),               # several output-channel blocks, ragged tiles
    dict(N=3, H=34, W=30, C0=64, C1=128, C2=128, pool=True),
])
def test_conv_hx_s16_tensors(lib, kw):
    """round 5: activations exchanged pre-split between k_conv_hx launches (ConvArgs.out_s16 / pool_s16 -> in_s16): bit-identical to the fp32 exchange, forward and dgrad chain"""
    K.hx_s16_chain_case(lib, "cuda", **kw)


@pytest.mark.parametrize("kw", [
    dict(N=8, H=32, W=32, Cin=128, Cout=512),                                  # R's gate dgrad: 4 x 16 tiles, 256 workgroups
    dict(N=8, H=16, W=16, Cin=256, Cout=1024, seed=1),                         # ... 16 x 16 maps: K split over slabs
    dict(N=8, H=64, W=64, Cin=128, Cout=128, seed=2),                          # D residual block: 8 x 16 x 128 / 64-channel tiles
    dict(N=8, H=128, W=128, Cin=64, Cout=64, seed=3),                          # 16 x 16 x 64, well-filled
    dict(N=4, H=128, W=128, Cin=64, Cout=32, seed=4),                          # row-split weight gradient (<= 32 output channels)
    dict(N=3, H=50, W=70, Cin=32, Cout=64, seed=5),                            # 32-channel dgrad tiles, ragged
    dict(N=8, H=32, W=32, Cin=32, Cout=32, seed=6),                            # 8 x 16 x 32
    dict(N=6, H=48, W=48, Cin=128, Cout=128, seed=7, force_big=1),             # 8-wave 16 x 16 x 128 tile
])
def test_pre_split_gradient_tensors(lib, kw):
    """round 6: conv-output gradients as S16-bf16 tensors -- point-wise producers write exactly the loaders' halves; k_wgrad_hx and every split-bf16 k_conv_hx tile variant give
    bit-identical results from them; border / column sums"""
    K.s16_grad_case(lib, "cuda", **kw)


@pytest.mark.parametrize("kw", [
    dict(N=2, H=64, W=64, segs=[(32, 0)], Cout=3, KS=7, bias=True, act=1, wgrad_precision=17, wgrad_tol=1e-4, dgrad_precision=17, dgrad_tol=1e-4),      # D's 7x7 head
    dict(N=5, H=44, W=100, segs=[(32, 0)], Cout=3, KS=7, wgrad_precision=17, wgrad_tol=1e-4, dgrad_precision=17, dgrad_tol=1e-4),                        # ragged tiles, many samples
    dict(N=2, H=32, W=48, segs=[(16, 0)], Cout=3, KS=7, bias=True, act=1, wgrad_precision=17, wgrad_tol=1e-4),                                           # reduced variant: 16 channels
    dict(N=16, H=32, W=32, segs=[(64, 0)], Cout=128, KS=1),                                                                                             # A's identity path
    dict(N=4, H=128, W=128, segs=[(16, 0)], Cout=32, KS=1),                                                                                             # E's first down-sampling block
    dict(N=6, H=20, W=20, segs=[(64, 0)], Cout=65, KS=1),                                                                                               # E's last block (65 channels), Breakout state map
])
def test_streaming_weight_gradients(lib, kw):
    """round 5 (conv_stream.hip): k_wgrad_head7 (7x7 head, split bf16, taps on the M side) and k_wgrad_1x1 (identity paths, operands straight from global memory)"""
    K.conv_case(lib, "cuda", **kw)


def test_conv_hx_register_weight_variants(lib):
    """round 6: k_conv_hx<BG> at the shapes of R's gate convolutions and their dgrads (and the simulator's small cases), forced on and off"""
    K.hx_register_weights_case(lib, "cuda", [
        ("conv", dict(N=1, H=10, W=20, segs=[(40, False), (9, True), (33, False)], Cout=48, bias=True)),
        ("conv", dict(N=1, H=9, W=17, segs=[(96, False)], Cout=32, act=3, seed=1)),
        ("conv", dict(N=8, H=32, W=32, segs=[(128, False), (16, True), (128, False)], Cout=512, bias=True, seed=2)),              # lstm0 gates: 8 x 16 x 64, 512 workgroups
        ("conv", dict(N=8, H=16, W=16, segs=[(256, False), (16, True), (256, False)], Cout=1024, bias=True, seed=3)),             # lstm1 gates
        ("conv", dict(N=8, H=32, W=32, segs=[(512, False)], Cout=128, precision=K.PREC_BF16X3, dgrad_seg=0, accumulate=True, seed=4)),   # gate dgrad: 4 x 16 tiles, 16 chunks
        ("conv", dict(N=1, H=14, W=64, segs=[(256, False)], Cout=128, bias=True, act=3, direct="tile4", split=True, seed=5)),
        ("s16", dict(N=8, H=32, W=32, Cin=128, Cout=512, seed=6, producers=False)),
        ("s16", dict(N=8, H=16, W=16, Cin=256, Cout=1024, seed=7, producers=False)),
    ])
