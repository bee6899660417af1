"""Counterpart of training/smooth_mi_trainer.py (SmoothMutualInformationLoss + `mi_estimator` in the checkpoint)."""
from .trainer import Trainer


class SmoothMITrainer(Trainer):
    SMOOTH_MI = True


def trainer(config, model, dataset, logger):
    return SmoothMITrainer(config, model, dataset, logger)
