"""Headless train / play / interpolate drivers (playablevideogeneration_amd/drivers.py; SURVEY.md section 8b "What calls it": train.py:76-108, play.py:115-207,
interpolate.py:102-158) on a tiny on-disk dataset in the reference's video format, with the emulator build of the kernels behind the plugin model."""
import os
import pickle

import numpy as np
import pytest
import torch
import yaml

from playablevideogeneration_amd import drivers as D
from playablevideogeneration_amd import video_dataset as VD
from playablevideogeneration_amd.evaluation_dataset_builder import EvaluationVideo
from tests.test_host_api_emu import _config, _make_model

H = W = 32


def _dataset(root, videos=6, frames=8, seed=0):
    rng = np.random.RandomState(seed)
    for vi in range(videos):
        fr = rng.randint(0, 256, size=(frames, H, W, 3)).astype(np.uint8)
        EvaluationVideo(fr, [int(rng.randint(0, 3)) for _ in range(frames)], [0.0] * frames, [{} for _ in range(frames)], [False] * frames).save(os.path.join(root, f"{vi:05d}"))


def _yaml_config(tmp_path):
    root = str(tmp_path / "data")
    os.makedirs(root)
    _dataset(root)
    cfg = _config()
    cfg["data"].update({"data_root": root, "dataset_splits": [0.5, 0.25, 0.25]})
    cfg["evaluation_dataset"] = {"builder": "playablevideogeneration_amd.evaluation_dataset_builder", "ground_truth_observations_init": 2}
    cfg["model"]["representation_network"]["target_input_size"] = [W, H]
    cfg["logging"] = {"output_root": str(tmp_path / "out"), "save_root": str(tmp_path / "ckpt"), "run_name": "run0"}
    cfg["training"]["batching"].update({"batch_size": 2, "skip_frames": 0, "num_workers": 0, "observations_count": 4, "observations_count_start": 4})
    cfg["training"].update({"max_steps": 1000, "save_freq": 1, "max_steps_per_epoch": 1})
    cfg["evaluation"] = {"evaluator": "playablevideogeneration_amd.evaluator", "eval_freq": 0, "max_evaluation_batches": 1,
                         "batching": {"batch_size": 2, "observations_count": 4, "observation_stacking": 1, "skip_frames": 0}}
    path = str(tmp_path / "cfg.yaml")
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    return path


def test_configuration_defaults_and_directories(tmp_path):
    cfg = D.load_configuration(_yaml_config(tmp_path))
    assert cfg["data"]["dataset_style"] == "flat" and cfg["data"]["crop"] is None and cfg["data"]["ground_truth_available"] is True      # configuration.py:47-73
    assert cfg["training"]["use_motion_weights"] is False and cfg["model"]["action_network"]["use_variations"] is True
    assert cfg["logging"]["save_root_directory"].endswith(os.path.join("ckpt", "run0")) and os.path.isdir(cfg["logging"]["save_root_directory"])
    for key in ("output_images_directory", "amt_sequences", "interpolated_sequences", "evaluation_dataset_directory", "evaluation_images_directory"):
        assert os.path.isdir(cfg["logging"][key])
    bad = dict(cfg, data=dict(cfg["data"], dataset_splits=[0.5, 0.6, 0.0]))
    with pytest.raises(Exception, match="sum to 1.0"):
        D.finish_configuration(bad, create_directories=False)
    with pytest.raises(Exception, match="does not exist"):
        D.finish_configuration(dict(cfg, data=dict(cfg["data"], data_root=str(tmp_path / "nowhere"))), create_directories=False)


def test_train_then_play_then_interpolate(tmp_path):
    cfg = D.load_configuration(_yaml_config(tmp_path))
    logger = D.HeadlessLogger(cfg, echo=False)
    datasets = VD.build_datasets(cfg)
    assert len(datasets["train"]) > 0 and len(datasets["validation"]) > 0
    m = _make_model(cfg)
    torch.manual_seed(0)
    res = D.train_loop(cfg, m, datasets, logger, max_steps=2)
    # train.py:76-108: `latest` after every epoch, a numbered checkpoint once save_freq steps have passed, both evaluators once annotations exist
    assert res["steps"] >= 2 and res["epochs"] >= 1
    save = cfg["logging"]["save_root_directory"]
    assert os.path.isfile(os.path.join(save, "latest.pth.tar")) and any(n.startswith("checkpoint_") for n in os.listdir(save))
    prefixes = [p for _, p, _ in res["evaluations"]]
    assert "validation_inferred_actions" in prefixes and "validation_gt_actions" in prefixes
    step, _, ev = res["evaluations"][0]
    assert 0.0 <= ev["validation_inferred_actions/actions_accuracy"] <= 1.0 and np.isfinite(ev["validation_inferred_actions/observations_loss/avg"])
    text = open(os.path.join(cfg["logging"]["output_directory"], "log.txt")).read()
    assert "step: 1 " in text and "== Evaluation" in text and os.path.isfile(os.path.join(cfg["logging"]["output_directory"], "metrics.jsonl"))
    # resume: a second run picks the checkpoint up and continues from its step (train.py:61-65)
    m2 = _make_model(cfg)
    res2 = D.train_loop(cfg, m2, datasets, logger, max_steps=res["steps"] + 1)
    # (an epoch ends AFTER step max_steps_per_epoch + 1, trainer.py:560: "performed > max" -- 2 steps per epoch here; max_steps is tested between epochs)
    assert res2["steps"] == res["steps"] + 2 and res2["trainer"].opt_steps == res2["steps"]
    # play: the action list replaces the key presses; frames are what generate_next returns, written as <out>/0/<i>.png + play_metadata.pkl
    obs = D._first_observations(datasets["validation"], 2)
    start = obs[1, 0]
    out = str(tmp_path / "play_results")
    played = D.play_loop(m2, start, [1, 3, 2, 0, 1], out)
    assert played["actions"] == [1, 3, 2] and played["frames"].shape == (4, H, W, 3) and len(played["timestamps"]) == 4
    m2.start_inference()
    o, want = start, [D.frame_to_uint8(start[:3])]
    with torch.no_grad():
        for a in (0, 2, 1):
            f, o = m2.generate_next(o, a)
            want.append(D.frame_to_uint8(f))
    assert np.array_equal(played["frames"], np.stack(want))
    from PIL import Image
    for i in range(4):
        assert np.array_equal(np.asarray(Image.open(os.path.join(out, "0", f"{i}.png"))), played["frames"][i])
    meta = pickle.load(open(os.path.join(out, "0", "play_metadata.pkl"), "rb"))
    assert meta["actions"] == [1, 3, 2] and len(meta["timestamps"]) == 4
    with pytest.raises(Exception, match="outside"):
        D.play_loop(m2, start, [9])
    # interpolate: steps + 1 sequences of frames_count + 1 frames; alpha = 0 acts exactly like the first action with zero variation
    seqs = D.interpolate_loop(m2, start, 0, 1, steps=2, frames_count=2)
    assert len(seqs) == 3 and all(s.shape == (3, H, W, 3) for s in seqs)
    m2.start_inference()
    with torch.no_grad():
        f, _ = m2.generate_next(start, 0)
    assert np.array_equal(seqs[0][1], D.frame_to_uint8(f))


def test_build_evaluation_dataset_loop(tmp_path):
    """build_evaluation_dataset.py:56-77: roll-outs of the TEST split (one-hot actions, zero variations, gt_init from the config) written in the on-disk video format
    that VideoDataset reads back"""
    cfg = D.load_configuration(_yaml_config(tmp_path))
    logger = D.HeadlessLogger(cfg, echo=False)
    datasets = VD.build_datasets(cfg)
    m = _make_model(cfg)
    n = D.build_dataset_loop(cfg, m, datasets, logger)
    assert n == len(datasets["test"]) and n > 0
    out = cfg["logging"]["evaluation_dataset_directory"]
    assert sorted(os.listdir(out)) == [f"{i:05d}" for i in range(n)]
    back = VD.VideoDataset(out, cfg["evaluation"]["batching"], VD.final_transform(cfg))
    assert len(back) == n and back.all_videos[0].get_frames_count() == cfg["evaluation"]["batching"]["observations_count"]
    assert back.all_videos[0].metadata[0]["model"] == "ours" and "inferred_action" in back.all_videos[0].metadata[0]
