"""Evaluation samplers with the reference's call signatures (evaluation/action_sampler.py, evaluation/action_variation_sampler.py),
device-agnostic (the reference hard-codes `.cuda()`).  Passed to `Model.__call__(..., action_sampler=..., action_variation_sampler=...)`;
the HIP driver calls them mid-forward through `caddy_set_sampler_hook`."""
from typing import Dict

import torch


class OneHotActionSampler:
    """one-hot of the most probable action (evaluation/action_sampler.py:6-34)"""

    def __call__(self, log_probabilities: torch.Tensor, ground_truth: torch.Tensor) -> torch.Tensor:
        out = torch.zeros_like(log_probabilities)
        out.scatter_(1, log_probabilities.argmax(dim=1, keepdim=True), 1.0)
        return out


class GroundTruthActionSampler:
    """one-hot of the ground-truth action mapped into the model's action space (evaluation/action_sampler.py:37-85)"""

    def __init__(self, ground_truth_to_actions_mapping: Dict):
        self.mapping_dict = ground_truth_to_actions_mapping

    def translate_ground_truth_indexes(self, ground_truth: torch.Tensor) -> torch.Tensor:
        out = ground_truth.clone()
        for gt_idx, idx in self.mapping_dict.items():
            out[ground_truth == gt_idx] = idx
        return out

    def __call__(self, log_probabilities: torch.Tensor, ground_truth: torch.Tensor) -> torch.Tensor:
        idx = self.translate_ground_truth_indexes(ground_truth).to(log_probabilities.device).long().reshape(-1, 1)
        out = torch.zeros_like(log_probabilities)
        out.scatter_(1, idx, 1.0)
        return out


class ZeroActionVariationSampler:
    """zero variation (evaluation/action_variation_sampler.py:6-29)"""

    def __call__(self, sampled_action_directions: torch.Tensor, action_samples: torch.Tensor) -> torch.Tensor:
        return sampled_action_directions * 0
