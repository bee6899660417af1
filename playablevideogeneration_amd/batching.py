"""Input-pipeline contract of the hot path: what the reference's `dataset/` package hands to the trainer.

    sample indices      dataset/video_dataset.py:114-149   (frame skip, newest-first stacking clamped at the start of the video)
    BatchElement/Batch  dataset/batching.py:10-95          (`to_tuple()` -> (observations, actions, rewards, dones), `.size`)
    collate             dataset/batching.py:97-112         (stack of per-observation channel-concatenated frame stacks)
    normalisation       dataset/transforms.py:90-107       (ToTensor + Normalize(0.5, 0.5): uint8 [0, 255] -> fp32 [-1, 1])

The tensor contract lives here; the reader of the reference's on-disk format (PNG folders + pickles, dataset/video.py) and the sample
grid over videos is `playablevideogeneration_amd.video_dataset`.  `Batch.to_tuple()` moves
the tensors to the current HIP device like the reference moves them to CUDA (batching.py:67-87).
"""
from typing import List, Sequence, Tuple

import torch


def observation_indices(initial_frame: int, observations_count: int, skip_frames: int, observation_stacking: int) -> Tuple[List[int], List[List[int]]]:
    """Frame indices of one sample (video_dataset.py:131-137): observation i is frame `initial + i (skip + 1)`; its stack holds that frame
    and the `stacking - 1` preceding observations (newest first), clamped to the first frame the sampling grid can reach (`initial % (skip + 1)`)."""
    step = skip_frames + 1
    obs = [initial_frame + i * step for i in range(observations_count)]
    min_frame = initial_frame % step
    stacks = [[max(o - k * step, min_frame) for k in range(observation_stacking)] for o in obs]
    return obs, stacks


def available_samples(frames_count: int, observations_count: int, skip_frames: int) -> int:
    """video_dataset.py:92-106: samples that fit in a video of `frames_count` frames"""
    return frames_count - (observations_count + (observations_count - 1) * skip_frames) + 1


def accumulated_rewards(rewards: Sequence[float], obs_indices: Sequence[int], skip_frames: int) -> List[float]:
    """video_dataset.py:143: the reward of an observation includes the rewards of the frames skipped to reach it"""
    return [sum(rewards[max(i - skip_frames, 0):i + 1]) for i in obs_indices]


def normalize_frame(frame_uint8_hwc: torch.Tensor) -> torch.Tensor:
    """transforms.py:99-102 (ToTensor -> float -> Normalize(mean .5, std .5)): (H, W, 3) uint8 -> (3, H, W) fp32 in [-1, 1]"""
    x = frame_uint8_hwc.permute(2, 0, 1).to(torch.float32) / 255.0
    return (x - 0.5) / 0.5


class BatchElement:
    """One sampled sequence: `observations[i]` is the list of `observation_stacking` (3, H, W) tensors of observation i, newest first
    (dataset/batching.py:10-42; the frames arrive already transformed here)."""

    def __init__(self, observations, actions, rewards, dones, video=None, initial_frame_index: int = 0):
        self.observations_count = len(observations)
        self.observations_stacking = len(observations[0])
        if len(actions) != self.observations_count or len(rewards) != self.observations_count or len(dones) != self.observations_count:
            raise Exception("Missing elements in the current batch")
        self.observations, self.actions, self.rewards, self.dones = observations, actions, rewards, dones
        self.video, self.initial_frame_index = video, initial_frame_index


def is_batch_element(x) -> bool:
    return all(hasattr(x, a) for a in ("observations", "actions", "rewards", "dones")) and not torch.is_tensor(getattr(x, "observations"))


class Batch:
    """(bs, observations_count, 3 * stacking, H, W) observations + (bs, observations_count) actions / rewards / dones (batching.py:44-95)"""

    def __init__(self, observations: torch.Tensor, actions: torch.Tensor, rewards: torch.Tensor, dones: torch.Tensor, videos=None, initial_frames=None):
        self.size = actions.size(1)
        self.observations, self.actions, self.rewards, self.dones = observations, actions, rewards, dones
        self.video, self.initial_frames = videos, initial_frames

    def to_cuda(self):
        self.observations, self.actions = self.observations.cuda(non_blocking=True), self.actions.cuda(non_blocking=True)
        self.rewards, self.dones = self.rewards.cuda(non_blocking=True), self.dones.cuda(non_blocking=True)

    def to_tuple(self, cuda=True) -> Tuple:
        if cuda and torch.cuda.is_available():
            self.to_cuda()
        return self.observations, self.actions, self.rewards, self.dones

    def pin_memory(self):
        self.observations, self.actions = self.observations.pin_memory(), self.actions.pin_memory()
        self.rewards, self.dones = self.rewards.pin_memory(), self.dones.pin_memory()
        return self


def single_batch_elements_collate_fn(batch: List[BatchElement]) -> Batch:
    """batching.py:97-112: per observation the stack is concatenated along channels (newest frame first), observations are stacked along
    time, elements along the batch; actions are int32 like the reference's `dtype=torch.int`."""
    obs = torch.stack([torch.stack([torch.cat(list(stack)) for stack in el.observations], dim=0) for el in batch], dim=0)
    actions = torch.stack([torch.tensor(el.actions, dtype=torch.int) for el in batch], dim=0)
    rewards = torch.stack([torch.tensor(el.rewards) for el in batch], dim=0)
    dones = torch.stack([torch.tensor(el.dones) for el in batch], dim=0)
    return Batch(obs, actions, rewards, dones, [el.video for el in batch], [el.initial_frame_index for el in batch])


def multiple_batch_elements_collate_fn(batch: List[Tuple[BatchElement]]) -> List[Batch]:
    """batching.py:114-126: one Batch per position of the tuples"""
    n = len(batch[0])
    return [single_batch_elements_collate_fn([els[i] for els in batch]) for i in range(n)]
