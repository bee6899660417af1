"""Headless `train` / `play` / `interpolate` drivers (SURVEY.md section 8b "What calls it"): the loops of the reference's entry scripts without their
display / wandb / ffmpeg plumbing, on the plugin seam the reference itself uses (dotted-path factories from the YAML).

    python -m playablevideogeneration_amd.drivers train       --config cfg.yaml [--max-steps N]
    python -m playablevideogeneration_amd.drivers play        --config cfg.yaml --actions 1,3,3,2 [--out play_results] [--sample 0:0]
    python -m playablevideogeneration_amd.drivers interpolate --config cfg.yaml --first 1 --second 2 [--steps 6] [--frames 8]
    python -m playablevideogeneration_amd.drivers build-dataset --config cfg.yaml

    train        train.py:76-108      epochs of trainer.train_epoch, `latest` checkpoint after each, `checkpoint_<step>` every save_freq steps, evaluation with the inferred
                                      actions every eval_freq steps and -- when the data carries annotations -- with the ground-truth actions mapped through the Hungarian
                                      matching of the first evaluator
    play         play.py:115-207      start_inference + generate_next per action; the action list replaces the key presses (1-based as typed there, 0 ends the sequence);
                                      frames as <out>/<sequence>/<i>.png and play_metadata.pkl {"actions", "timestamps"} as the reference writes them
    interpolate  interpolate.py:102-158  one sequence per interpolation value in linspace(0, 1, steps + 1) through generate_next_interpolation
    build-dataset  build_evaluation_dataset.py:17-77  the `builder(config, dataset, logger)` factory of config["evaluation_dataset"]["builder"] on the TEST split: roll-outs with
                                      one-hot actions and zero variations, written in the on-disk video format under logging.evaluation_dataset_directory

The configuration defaults and directory layout are those of utils/configuration.py:31-110.  The model runs on the GPU through libcaddy_hip.so; there is no CPU path
(the `*_loop` functions take the model object so that the tests can drive them with the emulator build of the same kernels).
"""
import argparse
import importlib
import json
import os
import pickle
import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .action_samplers import GroundTruthActionSampler
from .video_dataset import build_datasets


def load_configuration(path: str, create_directories: bool = True) -> Dict:
    """YAML -> config dict with the derived keys and defaults of utils/configuration.py:31-94 (+ :102-110: the directory structure)"""
    import yaml
    with open(path) as f:
        config = yaml.safe_load(f)
    return finish_configuration(config, create_directories)


def finish_configuration(config: Dict, create_directories: bool = True) -> Dict:
    data, log, tr, ev = config["data"], config["logging"], config["training"], config["evaluation"]
    if not os.path.isdir(data["data_root"]):
        raise Exception(f"Data directory {data['data_root']} does not exist")
    out = log["output_directory"] = os.path.join(log["output_root"], log["run_name"])
    log["save_root_directory"] = os.path.join(log["save_root"], log["run_name"])
    for key, sub in (("output_images_directory", "images"), ("amt_sequences", "amt_sequences"), ("interpolated_sequences", "interpolated_sequences"),
                     ("evaluation_dataset_directory", "evaluation_dataset"), ("evaluation_images_directory", "evaluation_images")):
        log[key] = os.path.join(out, sub)
    if "dataset_splits" in data:
        data["dataset_style"] = "flat"
        if len(data["dataset_splits"]) != 3:
            raise Exception("Dataset splits must speficy exactly 3 elements")
        if sum(data["dataset_splits"]) != 1.0:
            raise Exception("Dataset splits must sum to 1.0")
    else:
        data["dataset_style"] = "splitted"
    data.setdefault("crop", None)
    data.setdefault("ground_truth_available", True)
    ev.setdefault("eval_freq", 0)
    ev.setdefault("max_evaluation_batches", None)
    tr.setdefault("use_motion_weights", False)
    tr.setdefault("motion_weights_bias", 0.0)
    tr.setdefault("action_direction_plotting_freq", 1000)
    tr.setdefault("action_mutual_information_entropy_lambda", 1.0)
    tr.setdefault("max_steps_per_epoch", 10000)
    if tr.get("use_ground_truth_actions", False) and not data["ground_truth_available"]:
        raise Exception("Requested to use ground truth data, but no annotations are present in the dataset")
    config["model"]["action_network"].setdefault("use_variations", True)
    if create_directories:
        for key in ("output_directory", "save_root_directory", "output_images_directory", "amt_sequences", "interpolated_sequences", "evaluation_dataset_directory",
                    "evaluation_images_directory"):
            os.makedirs(log[key], exist_ok=True)
    return config


class HeadlessLogger:
    """utils/logger.py without wandb: `print` to stdout and to <output_directory>/log.txt, scalars as JSON lines in <output_directory>/metrics.jsonl"""

    def __init__(self, config, echo: bool = True):
        self.dir = config["logging"]["output_directory"]
        os.makedirs(self.dir, exist_ok=True)
        self.echo = echo

    def print(self, *args, **kwargs):
        text = " ".join(str(a) for a in args)
        if self.echo:
            print(text, **kwargs)
        with open(os.path.join(self.dir, "log.txt"), "a") as f:
            f.write(text + "\n")

    def log(self, scalars: Dict, step: Optional[int] = None):
        with open(os.path.join(self.dir, "metrics.jsonl"), "a") as f:
            f.write(json.dumps(dict({k: (float(v) if isinstance(v, (int, float, np.floating)) else v) for k, v in scalars.items()}, step=step)) + "\n")

    def get_wandb(self):      # code written against `logger.get_wandb().log(...)` / `.watch(...)` keeps working
        return self

    def watch(self, *a, **k):
        pass


def _factory(dotted: str, name: str):
    return getattr(importlib.import_module(dotted), name)


def build_model(config):
    """train.py:37-40 / play.py:44-47: the `model(config)` factory of config["model"]["architecture"], moved to the GPU"""
    model = _factory(config["model"]["architecture"], "model")(config)
    return model.cuda()


def train_loop(config, model, datasets, logger, max_steps: Optional[int] = None) -> Dict:
    """train.py:53-108 on a built model and datasets.  -> {"steps", "epochs", "evaluations": [(step, prefix, results)]}"""
    trainer = _factory(config["training"]["trainer"], "trainer")(config, model, datasets["train"], logger)
    ev_path = config["evaluation"].get("evaluator", "playablevideogeneration_amd.evaluator")
    inferred = _factory(ev_path, "evaluator")(config, datasets["validation"], logger, action_sampler=None, logger_prefix="validation_inferred_actions")
    with_gt = _factory(ev_path, "evaluator")(config, datasets["validation"], logger, action_sampler=None, logger_prefix="validation_gt_actions")
    try:      # resume (train.py:61-65)
        trainer.load_checkpoint(model)
    except Exception as e:
        logger.print(e)
        logger.print("- Warning: training without loading saved checkpoint")
    limit = config["training"]["max_steps"] if max_steps is None else min(config["training"]["max_steps"], max_steps)
    last_save = last_eval = 0
    epochs, evaluations = 0, []
    while trainer.global_step < limit:
        model.train()
        before = trainer.global_step
        trainer.train_epoch(model)
        epochs += 1
        trainer.save_checkpoint(model)
        if trainer.global_step > last_save + config["training"]["save_freq"]:
            trainer.save_checkpoint(model, f"checkpoint_{trainer.global_step}")
            last_save = trainer.global_step
        model.eval()
        if trainer.global_step > last_eval + config["evaluation"]["eval_freq"]:
            res = inferred.evaluate(model, trainer.global_step)
            evaluations.append((trainer.global_step, inferred.logger_prefix, res))
            if hasattr(logger, "log"):
                logger.log({k: v for k, v in res.items() if isinstance(v, (int, float))}, trainer.global_step)
            if config["data"]["ground_truth_available"]:      # ground-truth actions translated into the model's action space (train.py:96-104)
                with_gt.set_action_sampler(GroundTruthActionSampler(inferred.get_best_action_mappings()))
                res = with_gt.evaluate(model, trainer.global_step)
                evaluations.append((trainer.global_step, with_gt.logger_prefix, res))
                if hasattr(logger, "log"):
                    logger.log({k: v for k, v in res.items() if isinstance(v, (int, float))}, trainer.global_step)
            last_eval = trainer.global_step
        if trainer.global_step == before:
            raise Exception("train_epoch performed no step: the training split yields no batch (batch_size larger than the number of samples with drop_last?)")
    return {"steps": trainer.global_step, "epochs": epochs, "evaluations": evaluations, "trainer": trainer}


def frame_to_uint8(frame: torch.Tensor) -> np.ndarray:
    """(3, H, W) in [-1, 1] -> (H, W, 3) uint8 as play.py:143 does it (truncating cast)"""
    return (((frame + 1) / 2).permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8)


def _first_observations(dataset, batch_size: int) -> torch.Tensor:
    """the first validation batch's observations (play.py:101-103, 118): (B, T, 3S, H, W)"""
    from torch.utils.data import DataLoader
    from .batching import single_batch_elements_collate_fn
    for batch in DataLoader(dataset, batch_size=batch_size, shuffle=False, collate_fn=single_batch_elements_collate_fn):
        return batch.to_tuple(cuda=False)[0]
    raise Exception("the validation split is empty")


def play_loop(model, start_observation: torch.Tensor, actions: Sequence[int], out_dir: Optional[str] = None, sequence_idx: int = 0) -> Dict:
    """play.py:115-207 for ONE sequence: `actions` are what the user would type (1 .. actions_count; 0 = stop, implied at the end of the list).
    -> {"frames": uint8 (n + 1, H, W, 3), "actions": [...], "timestamps": [...]}; frames / metadata written under <out_dir>/<sequence_idx>/ when out_dir is given."""
    model.eval()
    seq_dir = None
    if out_dir is not None:
        seq_dir = os.path.join(out_dir, str(sequence_idx))
        os.makedirs(seq_dir, exist_ok=True)
    frames, stamps, done = [], [], []
    with torch.no_grad():
        obs = start_observation
        frame = obs[:3]
        model.start_inference()
        t0 = None
        for i in range(len(actions) + 1):
            img = frame_to_uint8(frame)
            if t0 is None:
                t0 = time.time()
                stamps.append(0)
            else:
                stamps.append(time.time() - t0)
            frames.append(img)
            if seq_dir is not None:
                from PIL import Image
                Image.fromarray(img).save(os.path.join(seq_dir, f"{i}.png"))
            if i == len(actions) or actions[i] == 0:
                break
            a = int(actions[i]) - 1
            if a < 0 or a >= model.module.dims["actions"]:
                raise Exception(f"action {actions[i]} outside [1, {model.module.dims['actions']}]")
            done.append(a + 1)
            frame, obs = model.generate_next(obs, a)
    meta = {"actions": done, "timestamps": stamps}
    if seq_dir is not None:
        with open(os.path.join(seq_dir, "play_metadata.pkl"), "wb") as f:
            pickle.dump(meta, f)
    return {"frames": np.stack(frames, axis=0), **meta}


def interpolate_loop(model, start_observation: torch.Tensor, first_action: int, second_action: int, steps: int, frames_count: int, out_dir: Optional[str] = None) -> List[np.ndarray]:
    """interpolate.py:102-158: for each value in linspace(0, 1, steps + 1) one sequence of `frames_count` generate_next_interpolation calls from the same start"""
    model.eval()
    sequences = []
    with torch.no_grad():
        for si, alpha in enumerate(np.linspace(0.0, 1.0, steps + 1).tolist()):
            model.start_inference()
            obs = start_observation
            frame = obs[:3]
            frames = []
            for i in range(frames_count + 1):
                img = frame_to_uint8(frame)
                frames.append(img)
                if out_dir is not None:
                    from PIL import Image
                    os.makedirs(os.path.join(out_dir, str(si)), exist_ok=True)
                    Image.fromarray(img).save(os.path.join(out_dir, str(si), f"{i}.png"))
                if i == frames_count:
                    break
                frame, obs = model.generate_next_interpolation(obs, first_action, second_action, alpha)
            sequences.append(np.stack(frames, axis=0))
    return sequences


def build_dataset_loop(config, model, datasets, logger) -> int:
    """build_evaluation_dataset.py:56-77 on a built model: -> number of videos written"""
    path = config.get("evaluation_dataset", {}).get("builder", "playablevideogeneration_amd.evaluation_dataset_builder")
    b = _factory(path, "builder")(config, datasets["test"], logger)
    model.eval()
    return len(b.build(model))


def _load_for_inference(config, logger, required=True):
    """play.py:44-70: model, datasets, checkpoint (mandatory there: "Cannot play without loading checkpoint")"""
    model = build_model(config)
    datasets = build_datasets(config)
    trainer = _factory(config["training"]["trainer"], "trainer")(config, model, datasets["train"], logger)
    try:
        trainer.load_checkpoint(model)
    except Exception as e:
        logger.print(e)
        if required:
            logger.print("Cannot play without loading checkpoint")
            raise SystemExit(1)
    return model, datasets


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("train"); p.add_argument("--config", required=True); p.add_argument("--max-steps", type=int, default=None)
    p = sub.add_parser("play"); p.add_argument("--config", required=True); p.add_argument("--actions", required=True, help="comma-separated, 1-based as typed in play.py")
    p.add_argument("--out", default="play_results"); p.add_argument("--sample", default="0:0", help="batch_index:observation_index of the first validation batch")
    p = sub.add_parser("interpolate"); p.add_argument("--config", required=True); p.add_argument("--first", type=int, required=True); p.add_argument("--second", type=int, required=True)
    p.add_argument("--steps", type=int, default=6); p.add_argument("--frames", type=int, default=8); p.add_argument("--out", default=None)
    p = sub.add_parser("build-dataset"); p.add_argument("--config", required=True)
    args = ap.parse_args(argv)
    config = load_configuration(args.config)
    logger = HeadlessLogger(config)
    if args.cmd == "train":
        model = build_model(config)
        res = train_loop(config, model, build_datasets(config), logger, args.max_steps)
        logger.print(f"- finished at step {res['steps']} after {res['epochs']} epoch(s)")
        return 0
    if args.cmd == "build-dataset":      # (build_evaluation_dataset.py goes on without a checkpoint: its `raise` is commented out)
        model, datasets = _load_for_inference(config, logger, required=False)
        n = build_dataset_loop(config, model, datasets, logger)
        logger.print(f"- {n} videos written to {config['logging']['evaluation_dataset_directory']}")
        return 0
    model, datasets = _load_for_inference(config, logger)
    obs = _first_observations(datasets["validation"], config["evaluation"]["batching"]["batch_size"])
    if args.cmd == "play":
        b, o = (int(x) for x in args.sample.split(":"))
        res = play_loop(model, obs[b, o].cuda(), [int(a) for a in args.actions.split(",") if a != ""], args.out)
        logger.print(f"- {len(res['frames'])} frames written to {os.path.join(args.out, '0')}")
        return 0
    out = args.out or config["logging"]["interpolated_sequences"]
    interpolate_loop(model, obs[0, 0].cuda(), args.first, args.second, args.steps, args.frames, out)
    logger.print(f"- {args.steps + 1} sequences written to {out}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
