"""Frame-quality metrics of the paper's evaluation protocol that need no third-party network: per-observation MSE and
PSNR between a reference and a generated sequence, the contracts of `evaluation/metrics/mse.py:13-24` and `evaluation/metrics/psnr.py:11-31`.
Inputs are (bs, observations_count, channels, height, width) tensors in the same value range; results are (bs, observations_count).
(FID / FVD / LPIPS / detector-based metrics depend on pretrained networks and stay out of scope.)"""
import torch


def mse(reference_observations: torch.Tensor, generated_observations: torch.Tensor) -> torch.Tensor:
    return torch.mean((reference_observations - generated_observations).pow(2), dim=[2, 3, 4])


def psnr(reference_observations: torch.Tensor, generated_observations: torch.Tensor, value_range: float = 1.0) -> torch.Tensor:
    """-10 log10(MSE of the range-normalised frames + 1e-8): the reference's stabilising constant caps the score at 80 dB"""
    err = torch.mean(((reference_observations - generated_observations) / value_range) ** 2, dim=[2, 3, 4])
    return -10.0 * torch.log10(err + 1e-8)


def rollout_quality(model, batch_tuple, ground_truth_observations_init: int = 1, gumbel_temperature: float = 1.0) -> dict:
    """MSE / PSNR of an eval-mode roll-out against its ground truth, frames mapped from [-1, 1] to [0, 1] like the evaluation dataset builder
    does (evaluation_dataset_builder.py:140-153): the end-to-end quality number of the paper's protocol on the HIP path."""
    was_training = model.training
    model.eval()
    with torch.no_grad():
        rec = model(batch_tuple, ground_truth_observations_init=ground_truth_observations_init, gumbel_temperature=gumbel_temperature)[0]
    model.train(was_training)
    gt = batch_tuple[0][:, 1:, 0:3].to(rec.device, rec.dtype)
    a, b = (gt + 1) / 2, (rec + 1) / 2
    m, p = mse(a, b), psnr(a, b)
    return {"mse": m.mean().item(), "psnr": p.mean().item(), "mse_per_position": m.mean(0).tolist(), "psnr_per_position": p.mean(0).tolist()}
