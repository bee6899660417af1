"""Build-container-only harness that imports the REAL reference (/root/reference) on CPU.

Never imported by the product, the GPU tests, smoke() or bench.py (the reference does not travel to the GPU
box).  Used by tools/gen_golden.py (to generate tests/golden/*.npz) and by tests/test_oracle.py
(skipped when /root/reference is absent).  Shims are the ones listed in SURVEY.md section 8c: .cuda() -> identity,
collections.Sequence alias, stub `wandb` and `torchvision` modules (VGG19 feature stack with the seeded weights of
oracle.caddy_oracle.make_vgg_params: the pretrained values are not available offline).
"""
import collections
import collections.abc
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get("CADDY_REFERENCE", "/root/reference")
VGG_SEED = 1234


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "model", "main_model"))


_installed = False


def install():
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not present")
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    if not hasattr(collections, "Sequence"):
        collections.Sequence = collections.abc.Sequence
    wandb = types.ModuleType("wandb")
    wandb.init = lambda *a, **k: None
    wandb.log = lambda *a, **k: None
    wandb.watch = lambda *a, **k: None
    wandb.Image = lambda *a, **k: None
    sys.modules.setdefault("wandb", wandb)
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")
        tvt = types.ModuleType("torchvision.transforms")

        def vgg19(pretrained=False):
            # torchvision's VGG-19 "E" feature stack; the pretrained weights cannot be downloaded here, so the 16 conv tensors are
            # filled from oracle.caddy_oracle.make_vgg_params (seeded; the tests re-derive the same values instead of storing them)
            cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
            layers, c = [], 3
            for v in cfg:
                if v == "M":
                    layers.append(nn.MaxPool2d(2, 2))
                else:
                    layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=True)]
                    c = v
            m = types.SimpleNamespace()
            m.features = nn.Sequential(*layers)
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from oracle import caddy_oracle as O
            V = O.make_vgg_params(VGG_SEED)
            with torch.no_grad():
                for k, t in m.features.state_dict().items():
                    t.copy_(V["features." + k])
            return m

        tvm.vgg19 = vgg19
        tv.models, tv.transforms = tvm, tvt
        sys.modules["torchvision"], sys.modules["torchvision.models"], sys.modules["torchvision.transforms"] = tv, tvm, tvt
    if REF not in sys.path:
        sys.path.insert(0, REF)
    _installed = True


def make_config(*, variant, actions, action_dim, hidden, stacking, state_res, hard_gumbel=False, use_gumbel=True,
                use_variations=True, alpha=0.1, mi_alpha=0.2, ensamble_size=1):
    """Minimal config dict with exactly the keys the hot path reads (SURVEY 8a 'Config keys')."""
    return {
        "data": {"actions_count": actions},
        "model": {
            "architecture": "model.main_model.model" if variant == "main" else "model.reduced_model.model",
            "representation_network": {"state_features": 64, "state_resolution": list(state_res)},
            "dynamics_network": {"hidden_state_size": hidden, "random_noise_size": 32},
            "action_network": {"ensamble_size": ensamble_size, "use_gumbel": use_gumbel, "hard_gumbel": hard_gumbel,
                               "gumbel_temperature": 1.0, "action_space_dimension": action_dim,
                               "use_variations": use_variations},
            "centroid_estimator": {"alpha": alpha},
        },
        "training": {"batching": {"observation_stacking": stacking}, "use_ground_truth_actions": False,
                     "pretraining_detach": False, "mutual_information_estimation_alpha": mi_alpha},
    }


def build_reference_model(config, params):
    """Instantiate the reference Model and overwrite its state_dict with `params` (name -> tensor)."""
    install()
    import importlib
    m = importlib.import_module(config["model"]["architecture"]).model(config)
    sd = m.state_dict()
    missing = set(sd) - set(params)
    extra = set(params) - set(sd)
    assert not missing and not extra, (sorted(missing)[:5], sorted(extra)[:5])
    m.load_state_dict({k: v.clone() for k, v in params.items()})
    return m
