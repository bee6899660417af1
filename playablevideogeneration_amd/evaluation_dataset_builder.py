"""Counterpart of evaluation/evaluation_dataset_builder.py for the HIP path:
`config["evaluation_dataset"]["builder"] = "playablevideogeneration_amd.evaluation_dataset_builder"`, factory `builder(config, dataset, logger)`
(build_evaluation_dataset.py:38,55,57).

`build(model)` rolls the test set out exactly like the reference (evaluation_dataset_builder.py:37-81): full-model forward in eval mode with
`ground_truth_observations_init` ground-truth frames (4 in the reference configs), the arg-max one-hot action sampler and zero action
variations (evaluation/action_sampler.py:6-30, action_variation_sampler.py:6-25), temperature `gumbel_temperature_end`; the reconstruction is
padded with the first ground-truth frame, mapped from [-1, 1] to [0, 1] when its minimum is negative, and written as one folder per sequence in
the reference's dataset format (dataset/video.py:95-156): `NNNNN.png` frames + `actions.pkl`, `rewards.pkl`, `metadata.pkl`, `dones.pkl`.
Nothing but the forward runs on the GPU; the forward is the same `caddy_forward_full` the training step uses.
"""
import os
import pickle
from typing import List

import numpy as np
import torch

from .action_samplers import OneHotActionSampler, ZeroActionVariationSampler
from .batching import is_batch_element, single_batch_elements_collate_fn


class EvaluationVideo:
    """One generated sequence in the reference's on-disk layout (dataset/video.py: frames, actions, rewards, metadata, dones)."""

    def __init__(self, frames: np.ndarray, actions: List, rewards: List, metadata: List, dones: List):
        n = len(frames)
        if len(actions) != n or len(rewards) != n or len(metadata) != n or len(dones) != n:
            raise Exception("All arguments must have the same length")            # dataset/video.py:57-59
        self.frames, self.actions, self.rewards, self.metadata, self.dones = frames, actions, rewards, metadata, dones

    def save(self, path: str, extension: str = "png"):
        if os.path.isdir(path):
            raise Exception(f"A directory at '{path}' already exists")               # dataset/video.py:139-140
        os.makedirs(path)
        for name, obj in (("actions", self.actions), ("rewards", self.rewards), ("metadata", self.metadata), ("dones", self.dones)):
            with open(os.path.join(path, name + ".pkl"), "wb") as f:
                pickle.dump(obj, f)
        from PIL import Image
        for i, fr in enumerate(self.frames):
            Image.fromarray(fr).save(os.path.join(path, f"{i:05d}.{extension}"))


class EvaluationDatasetBuilder:
    def __init__(self, config, dataset, logger, logger_prefix="test"):
        self.config, self.logger, self.logger_prefix, self.dataset = config, logger, logger_prefix, dataset
        self.output_path = config["logging"]["evaluation_dataset_directory"]
        self.ground_truth_observations_init = config["evaluation_dataset"]["ground_truth_observations_init"]
        self.action_variation_sampler = ZeroActionVariationSampler()
        self.temperature = config["training"]["gumbel_temperature_end"]
        self.batch_size = config["evaluation"]["batching"]["batch_size"]

    def _batches(self):
        """DataLoader(dataset, batch_size, shuffle=False, collate_fn=single_batch_elements_collate_fn) for datasets of BatchElements
        (evaluation_dataset_builder.py:29); an iterable of ready batch tuples / Batch objects is used as it is."""
        ds = self.dataset
        if hasattr(ds, "__getitem__") and hasattr(ds, "__len__") and len(ds) > 0 and is_batch_element(ds[0]):
            from torch.utils.data import DataLoader
            nw = int(self.config["evaluation"]["batching"].get("num_workers", 0))
            return DataLoader(ds, batch_size=self.batch_size, shuffle=False, collate_fn=single_batch_elements_collate_fn, num_workers=nw)
        return ds

    @staticmethod
    def check_and_normalize_range(observations: torch.Tensor) -> torch.Tensor:
        """[-1, 1] -> [0, 1] when the minimum is negative (evaluation_dataset_builder.py:140-153)"""
        if torch.min(observations).item() < 0:
            observations = (observations + 1) / 2
        return observations

    def predictions_to_videos(self, images: np.ndarray, actions: np.ndarray, encoded_mus: np.ndarray) -> List[EvaluationVideo]:
        """(bs, T, H, W, C) images in [0, 1], (bs, T-1) inferred actions, (bs, T-1, Da) sampled action directions -> videos whose metadata
        carries the inferred / encoded action of every transition (evaluation_dataset_builder.py:83-123)"""
        images = (images * 255).astype(np.uint8)
        bs, T = images.shape[:2]
        if actions.shape[0] != bs:
            raise Exception(f"Images have batch size {bs} but actions have batch size {actions.shape[0]}")
        if actions.shape[1] != T - 1:
            raise Exception(f"Images have sequence length {T} but actions have sequence length {actions.shape[1]}")
        videos = []
        for b in range(bs):
            meta = [{"model": "ours", "inferred_action": a, "encoded_action": mu} for a, mu in zip(actions[b].tolist(), encoded_mus[b].tolist())]
            meta.append({"model": "ours"})                                           # no information for the last sample
            videos.append(EvaluationVideo(images[b], [0] * T, [0] * T, meta, [False] * T))
        return videos

    def build(self, model, write: bool = True) -> List[EvaluationVideo]:
        all_videos: List[EvaluationVideo] = []
        was_training = model.training
        model.eval()
        with torch.no_grad():
            for batch in self._batches():
                batch_tuple = batch.to_tuple() if hasattr(batch, "to_tuple") else batch
                results = model(batch_tuple, ground_truth_observations_init=self.ground_truth_observations_init, action_sampler=OneHotActionSampler(),
                                action_variation_sampler=self.action_variation_sampler, gumbel_temperature=self.temperature)
                rec, selected_actions, sampled_dirs = results[0], results[5], results[11]
                first = batch_tuple[0][:, 0:1, 0:3].to(rec.device, rec.dtype)           # pad with the first ground-truth frame
                rec = self.check_and_normalize_range(torch.cat([first, rec], dim=1))
                images = np.moveaxis(rec.cpu().numpy(), 2, -1)
                all_videos.extend(self.predictions_to_videos(images, selected_actions.cpu().numpy(), sampled_dirs.cpu().numpy()))
        model.train(was_training)
        if write:
            self.create_dataset(self.output_path, all_videos)
        return all_videos

    @staticmethod
    def create_dataset(path: str, videos: List[EvaluationVideo], extension: str = "png"):
        for idx, video in enumerate(videos):
            video.save(os.path.join(path, f"{idx:05d}"), extension)


def builder(config, dataset, logger):
    return EvaluationDatasetBuilder(config, dataset, logger)
