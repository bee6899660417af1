"""Reader side of the input-pipeline contract: the reference's on-disk video format and its sample grid, producing the
`BatchElement`s that `playablevideogeneration_amd.batching` collates for the trainer / evaluator / dataset builder.

    on-disk format   dataset/video.py:95-156        one directory per video: `NNNNN.<ext>` frames + actions.pkl, rewards.pkl, metadata.pkl, dones.pkl
    sample grid      dataset/video_dataset.py:92-149 observation i of a sample = frame `initial + i (skip + 1)`, stacks newest first, clamped at the start
    frame transform  dataset/transforms.py:13-30,90-107  crop -> bilinear resize to `target_input_size` -> [0, 255] -> [-1, 1]
    splits           dataset/dataset_splitter.py:11-46   "flat" (fractions of one sorted directory) / "splitted" (train / val / test sub-directories)

Nothing here touches the GPU: frames are decoded with PIL on the host (DataLoader workers), stacked by the collate function and moved by
`Batch.to_tuple()` / `DevicePrefetcher`.  `EvaluationVideo.save` (evaluation_dataset_builder.py) writes the same format this module reads.
"""
import glob
import os
import pickle
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

from .batching import BatchElement, accumulated_rewards, available_samples, normalize_frame, observation_indices

ANNOTATION_FILES = ("actions", "rewards", "metadata", "dones")


def _fill_defaults(name: str, seq: list, default, where: str) -> list:
    """a per-frame annotation list is either fully present or all None (dataset/video.py:49-87); None lists take the default value"""
    nones = sum(v is None for v in seq)
    if nones not in (0, len(seq)):
        raise Exception(f"Video dataset at {where} metadata error: both None and not None data are present ({name})")
    return [default if not isinstance(default, dict) else {} for _ in seq] if nones and len(seq) else seq


class VideoOnDisk:
    """One video directory.  Annotations live in memory; frames are decoded on demand (dataset/video.py:9-12,129-156)."""

    def __init__(self, path: str):
        if not os.path.isdir(path):
            raise Exception(f"Cannot load video: '{path}' is not a directory")
        self.frames_path = path
        ann = {}
        for name in ANNOTATION_FILES:
            with open(os.path.join(path, name + ".pkl"), "rb") as f:
                ann[name] = list(pickle.load(f))
        n = len(ann["actions"])
        if any(len(ann[k]) != n for k in ANNOTATION_FILES):
            raise Exception("Read data have inconsistent number of frames")
        self.actions = _fill_defaults("actions", ann["actions"], 0, path)
        self.rewards = _fill_defaults("rewards", ann["rewards"], 0.0, path)
        self.metadata = _fill_defaults("metadata", ann["metadata"], {}, path)
        self.dones = _fill_defaults("dones", ann["dones"], False, path)
        first = glob.glob(os.path.join(path, "00000.*"))
        if len(first) != 1:
            raise Exception("Could not find first video frame")
        self.extension = first[0].split(".")[-1]

    def get_frames_count(self) -> int:
        return len(self.actions)

    def get_frame_at(self, idx: int):
        """PIL image of frame idx; transparent images are flattened onto white like the reference does (dataset/video.py:158-175)"""
        if idx < 0 or idx >= len(self.actions):
            raise Exception(f"Index {idx} is out of range")
        from PIL import Image
        img = Image.open(os.path.join(self.frames_path, f"{idx:05d}.{self.extension}"))
        if img.mode in ("RGBA", "LA") or (img.mode == "P" and "transparency" in img.info):
            rgba = img.convert("RGBA")
            bg = Image.new("RGBA", rgba.size, (255, 255, 255, 255))
            bg.paste(rgba, mask=rgba.split()[-1])
            img = bg.convert("RGB")
        return img


def final_transform(config) -> Callable:
    """PIL image -> (3, H, W) fp32 tensor in [-1, 1]: crop `data.crop` ([left, upper, right, lower] or None), bilinear resize to
    `model.representation_network.target_input_size` ((width, height)) when the size differs, then (x / 255 - 0.5) / 0.5
    (dataset/transforms.py:13-30,90-107)."""
    crop = config["data"]["crop"]
    size = tuple(config["model"]["representation_network"]["target_input_size"])

    def transform(image):
        from PIL import Image
        if crop is not None:
            image = image.crop(crop)
        if image.size != size:
            image = image.resize(size, Image.BILINEAR)
        return normalize_frame(torch.from_numpy(np.asarray(image.convert("RGB"), dtype=np.uint8).copy()))
    return transform


class VideoDataset(Dataset):
    """Dataset of sampled sequences over a directory of videos (dataset/video_dataset.py:14-149).  `batching_config` is the reference's
    `training.batching` / `evaluation.batching` dict (observations_count, observation_stacking, skip_frames); `transform` maps a PIL frame to a
    (3, H, W) tensor; `allowed_videos` restricts the directory names (flat splits)."""

    def __init__(self, path: str, batching_config: Dict, transform: Callable, allowed_videos: Optional[Sequence[str]] = None):
        if not os.path.isdir(path):
            raise Exception(f"Dataset directory '{path}' is not a directory")
        self.batching_config = batching_config
        self.observations_stacking = batching_config["observation_stacking"]
        self.skip_frames = batching_config["skip_frames"]
        self.final_transform = transform
        names = sorted(os.listdir(path))
        allowed = set(names if allowed_videos is None else allowed_videos)
        self.all_videos = [VideoOnDisk(os.path.join(path, n)) for n in names if n in allowed and os.path.isdir(os.path.join(path, n))]
        self.observations_count = None
        self.set_observations_count(batching_config["observations_count"])

    def set_observations_count(self, observations_count: int):
        """the trainer grows the sequence length during training (training/trainer.py:139-152): re-derive the sample grid"""
        if self.observations_count != observations_count:
            self.observations_count = observations_count
            self.available_samples_list = [available_samples(v.get_frames_count(), observations_count, self.skip_frames) for v in self.all_videos]
            self.total_available_samples = sum(self.available_samples_list)

    def __len__(self):
        return self.total_available_samples

    def locate(self, index: int) -> Tuple[int, int]:
        """sample index -> (video index, first frame); samples are numbered video by video (video_dataset.py:114-127)"""
        if index < 0 or index >= self.total_available_samples:
            raise Exception(f"Requested sample at index {index} is out of range")
        for vi, n in enumerate(self.available_samples_list):
            if index < n:
                return vi, index
            index -= n
        raise AssertionError

    def __getitem__(self, index: int) -> BatchElement:
        vi, first = self.locate(index)
        video = self.all_videos[vi]
        obs_idx, stacks = observation_indices(first, self.observations_count, self.skip_frames, self.observations_stacking)
        cache: Dict[int, torch.Tensor] = {}

        def frame(i):                                         # consecutive stacks share frames: decode each once
            if i not in cache:
                cache[i] = self.final_transform(video.get_frame_at(i))
            return cache[i]
        observations = [[frame(i) for i in st] for st in stacks]
        return BatchElement(observations, [video.actions[i] for i in obs_idx], accumulated_rewards(video.rewards, obs_idx, self.skip_frames),
                            [video.dones[i] for i in obs_idx], video, first)


def generate_splits(config) -> Dict[str, Tuple[str, Dict, Optional[List[str]]]]:
    """{"train" | "validation" | "test": (path, batching config, allowed directory names | None)} (dataset/dataset_splitter.py:11-46)"""
    style = config["data"]["dataset_style"]
    root = config["data"]["data_root"]
    if style == "flat":
        names = sorted(os.listdir(root))
        fr = config["data"]["dataset_splits"]
        n_train, n_val = int(len(names) * fr[0]), int(len(names) * fr[1])
        return {"train": (root, config["training"]["batching"], names[:n_train]),
                "validation": (root, config["evaluation"]["batching"], names[n_train:n_train + n_val]),
                "test": (root, config["evaluation"]["batching"], names[n_train + n_val:])}
    if style == "splitted":
        return {"train": (os.path.join(root, "train"), config["training"]["batching"], None),
                "validation": (os.path.join(root, "val"), config["evaluation"]["batching"], None),
                "test": (os.path.join(root, "test"), config["evaluation"]["batching"], None)}
    raise Exception(f"Unknown dataset style '{style}'")


def build_datasets(config) -> Dict[str, VideoDataset]:
    """what train.py:42-51 does before it calls the trainer / evaluator factories"""
    tf = final_transform(config)
    return {k: VideoDataset(path, batching, tf, allowed) for k, (path, batching, allowed) in generate_splits(config).items()}
