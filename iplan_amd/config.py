"""Default ``args`` namespace for the hot path.

The reference builds a flat ``SimpleNamespace`` from ``config/default.yaml`` <- ``config/envs/*.yaml``
<- ``config/algs/ippo.yaml`` (first writer wins, main.py:59-69) and every class on the path reads
its hyper-parameters by attribute (SURVEY.md §8b lists the consumed attributes).  This module
restates those defaults so tests, ``bench.py`` and examples can build the same namespace without
the YAML files; a real deployment keeps passing the reference's own ``args`` object.
"""
from types import SimpleNamespace

_BASE = dict(
    # config/default.yaml
    runner="ippo", mac="dcntrl", env="highway", batch_size_run=32, t_max=2000000, use_cuda=True,
    gamma=0.99, batch_size=255, buffer_size=256, lr=0.0005, critic_lr=0.0005, optim_eps=1e-5,
    agent="ippo", critic="ippo", rnn_hidden_dim=64, mlp_hidden_dim=64, obs_agent_id=True,
    obs_last_action=True, log_prefix="ippo_GAT_behavior_stable_H_", learner_log_interval=20000,
    max_history_len=10,
    Behavior_enable=True, Behavior_warmup=20000, encoder_rnn_dim=32, num_encoder_layer=1,
    latent_dim=8, decoder_rnn_dim=64, num_decoder_layer=1, lr_behavior=0.0001, decoder_dropout=0.1,
    soft_update_enable=True, soft_update_coef=0.1, behavior_variation_penalty=0,
    thres_small_variation=0.005, behavior_fully_connected=False,
    GAT_enable=True, GAT_use_behavior=True, GAT_warmup=20000, GAT_hidden_dim=32, attention_dim=32,
    teacher_forcing_ratio=0, pred_batch_size=64, lr_predict=0.00002, pred_dropout=0.1, pred_length=5,
    use_max_grad_norm=True, max_grad_norm=10.0,
    # config/algs/ippo.yaml
    weight_decay=0, ppo_epoch=15, use_clipped_value_loss=True, use_linear_lr_decay=False,
    clip_param=0.2, num_mini_batch=1, data_chunk_length=10, value_loss_coef=0.5, entropy_coef=0.01,
    use_gae=True, gae_lambda=0.95, use_huber_loss=True, huber_delta=10.0, gain=0.01,
    use_orthogonal=True, use_policy_active_masks=True, use_value_active_masks=True,
    use_recurrent_policy=True, recurrent_N=1, use_ReLU=True, stacked_frames=1, layer_N=1,
    use_feature_normalization=True, use_popart=True, action_selector="epsilon_greedy",
    epsilon_start=1.0, epsilon_finish=0.05, epsilon_anneal_time=50000, agent_output_type=None,
    # config/envs/highway.yaml
    n_actions=5, obs_shape_single=5, n_agents=5, n_other_vehicles=50, episode_limit=90,
)

_ENVS = {
    # Highway (configs 2-4 of BASELINE.json): N = 50 + 5 entities, d = 5
    "highway": dict(env="highway", n_agents=5, n_actions=5, obs_shape_single=5, max_vehicle_num=55,
                    episode_limit=90, obs_shape=25, state_shape=25),
    # MPE easy (config 1): 3 agents + 3 landmarks, d = 4, GAT / Behaviour off
    "mpe_easy": dict(env="MPE", n_agents=3, n_actions=5, obs_shape_single=4, max_vehicle_num=6,
                     episode_limit=50, obs_shape=24, state_shape=24, GAT_enable=False,
                     Behavior_enable=False, GAT_use_behavior=False),
}


def default_args(env="highway", **overrides):
    """Return a SimpleNamespace carrying every attribute the hot path consumes."""
    cfg = dict(_BASE)
    cfg.update(_ENVS[env])
    cfg.update(overrides)
    return SimpleNamespace(**cfg)
