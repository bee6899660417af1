"""`config["model"]["architecture"] = "playablevideogeneration_amd.reduced_model"` -- counterpart of model/reduced_model/model.py
(half-width decoder, model/reduced_model/rendering_network.py:30-42)."""
from .model import Model as _Main


class Model(_Main):
    VARIANT = "reduced"


def model(config):
    return Model(config)
