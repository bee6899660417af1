"""Perceptual-loss parity metrics for the arithmetic options of the VGG19 convolutions (run on the GPU box):
split f16 / split bf16 (default), single-product f16 / bf16, exact fp32.   python tools/vgg_precision_study.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from playablevideogeneration_amd import _lib  # noqa: E402
from tests import model_cases as M  # noqa: E402

lib = _lib.load()
for name in ("perc_main_s1",):
    for label, prec in (("exact fp32", (0, 0)), ("split f16 / split bf16", (16, 17)), ("split bf16 fwd / split bf16 dgrad", (17, 17)), ("f16 x1 fwd / split bf16 dgrad", (18, 17)), ("f16 x1 / bf16 x1", (18, 19))):
        try:
            eng, info = M.perceptual_case(name, lib, "cuda", prep=lambda e: e.set_vgg_precision(*prec))
            print(f"{name:22s} {label:32s} PASS  per-image median {['%.1e' % v for v in info['per_image_median']]} worst {['%.1e' % v for v in info['per_image_worst']]} "
                  f"vs golden {['%.1e' % v for v in info['vs_golden']]} ref-sens {['%.1e' % v for v in info['reference_sensitivity_1e5']]}", flush=True)
        except AssertionError as e:
            print(f"{name:22s} {label:32s} FAIL  {str(e)[:300]}", flush=True)
        torch.cuda.empty_cache()
