"""Hot-path hyper-parameters of the reference YAMLs (configs/01_bair.yaml, 02_breakout.yaml, 03_tennis.yaml) combined with
the geometries BASELINE.json names; only the keys the path reads (SURVEY.md section 8a "Config keys")."""

BAIR = dict(variant="main", actions=7, action_dim=2, hidden=128, stacking=1)        # configs/01_bair.yaml:18,38,56,117
BREAKOUT = dict(variant="reduced", actions=3, action_dim=1, hidden=64, stacking=1)  # configs/02_breakout.yaml
TENNIS = dict(variant="main", actions=7, action_dim=5, hidden=128, stacking=4)      # configs/03_tennis.yaml

# training/trainer.py:494-500 weights of configs/01_bair.yaml:122-156 (perceptual_loss_lambda: 1.0 there)
LOSS_WEIGHTS = dict(rec=1.0, states=0.2, entropy=0.0, dir_kl=1e-4, mi=0.15, state_kl=0.0, mi_entropy=1.0, perceptual=1.0)

WORKLOADS = {
    # BASELINE.json configs[1]: the configuration the headline metric is quoted on
    "bair256_t16_b8": dict(BAIR, batch=8, seq_len=16, height=256, width=256, gt_init=6, tau=0.4),
    # configs[0]: plumbing case
    "breakout64_t8_b4": dict(BREAKOUT, batch=4, seq_len=8, height=64, width=64, gt_init=6, tau=0.85),
    # configs[4] (one DP shard)
    "breakout160_t9_b8": dict(BREAKOUT, batch=8, seq_len=9, height=160, width=160, gt_init=6, tau=0.4),
}
