"""Parameter initialisation with the reference's default scheme (torch defaults of nn.Conv2d / nn.Linear /
nn.BatchNorm2d, zeros for the learned ConvLSTM initial states -- convolutional_lstm.py:33-34 -- and N(0,1) centroids --
centroid_estimator.py:26-27), written directly into an Engine's flat parameter buffer."""
import math

import torch


def init_parameters(engine, seed: int = 0):
    """`engine`: anything with `.table` [(name, offset, shape, kind)] and `.view(entry)` (Engine or Model)."""
    g = torch.Generator().manual_seed(seed)
    fan = {}
    for name, off, shape, kind in engine.table:
        v = engine.view((name, off, shape, kind))
        leaf = name.rsplit(".", 1)[1]
        if leaf == "running_mean":
            v.zero_()
        elif leaf == "running_var":
            v.fill_(1.0)
        elif "initial_hidden" in name:
            v.zero_()
        elif name == "centroid_estimator.estimated_centroids":
            v.copy_(torch.randn(shape, generator=g))
        elif len(shape) >= 2:                       # conv / linear weight: kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in))
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            fan[name.rsplit(".", 1)[0]] = fan_in
            b = 1.0 / math.sqrt(fan_in)
            v.copy_((torch.rand(shape, generator=g) * 2 - 1) * b)
        else:
            prefix = name.rsplit(".", 1)[0]
            is_bn = any(t[0] == prefix + ".running_mean" for t in engine.table)
            if is_bn:
                v.fill_(1.0 if leaf == "weight" else 0.0)
            else:                                   # conv / linear bias
                b = 1.0 / math.sqrt(fan.get(prefix, 1))
                v.copy_((torch.rand(shape, generator=g) * 2 - 1) * b)


def random_vgg19_state(seed: int = 0):
    """torchvision vgg19().features state-dict SHAPES with random values (kaiming-normal weights, zero biases = torchvision's own
    `_initialize_weights`): what the synthetic benchmark loads, since the pretrained file cannot be downloaded on an air-gapped box."""
    g = torch.Generator().manual_seed(seed)
    sd, cin = {}, 3
    idx = 0
    for v in [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]:
        if v == "M":
            idx += 1
            continue
        sd[f"features.{idx}.weight"] = torch.randn((v, cin, 3, 3), generator=g) * math.sqrt(2.0 / (v * 9))      # fan_out mode
        sd[f"features.{idx}.bias"] = torch.zeros(v)
        cin = v
        idx += 2
    return sd
