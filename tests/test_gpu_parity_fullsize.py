"""GPU parity of what bench.py times, at the sizes it is timed (VERDICT r1 "next" #1):

* the device-resident rollout (SyntheticLoop._rollout_body: write_back / out= launches, two streams) against the oracle
  stepped the same way -- BASELINE config 3 (32 envs x 5 agents x 55 entities) and config 2 (16 envs, Behaviour off);
* each learner at config 3 size against the oracle (fp64 ground truth, fp32 oracle beside it), ALL FIVE agents of the fused
  launches (round 4; agent 0 alone in the round-3 tests that stay): Behavior_policy.learn at E = 32 over the whole episode,
  IPPOLearner.train on 255 x 90 = 22 950 rows (one epoch every agent, two epochs agents 0 and 4), Prediction_policy.learn at
  64 x 55; and the rollout body over the FULL 90-step episode;
* config 2 (F = 2045) and config 5 (N = 64, D = 128) forward + backward;
* the reference-shaped per-agent PPO methods (a13) on the GPU.
Worst errors are appended to gpurun_out/parity_errors.json (copied to profiles/ as the tolerance evidence)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _log(name, worst):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_errors.json")
    data = {}
    if os.path.exists(path):
        with open(path) as f:
            data = json.load(f)
    data[name] = worst
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print(name, worst)


def _args(**kw):
    from iplan_amd.config import default_args
    return default_args("highway", use_cuda=True, **kw)


def test_rollout_body_config3_vs_oracle():
    from tests.rollout_oracle import check_rollout_body
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    _log("rollout_body_cfg3_E32_T4", check_rollout_body(_args(episode_limit=4, batch_size_run=32), 32, "cuda", seed=21))


def test_rollout_body_config3_full_episode_vs_oracle():
    """VERDICT r3 "next" #4: the WHOLE episode the benchmark times -- config 3, 32 envs x 5 agents x 55 entities, T = 90 vector
    steps, gumbel noise and the action race injected -- every field of every step against the oracle: the error growth of the
    tau = 0.01 gate and of the carried recurrent states over 90 dependent steps (actions bit-equal, states <= 1e-5)"""
    from tests.rollout_oracle import check_rollout_body
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    _log("rollout_body_cfg3_E32_T90", check_rollout_body(_args(batch_size_run=32), 32, "cuda", seed=31))


def test_rollout_body_config2_vs_oracle():
    from tests.rollout_oracle import check_rollout_body
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    _log("rollout_body_cfg2_E16_T3", check_rollout_body(_args(episode_limit=3, batch_size_run=16, Behavior_enable=False), 16, "cuda", seed=22))


def test_behavior_learn_config3_one_agent_vs_oracle():
    from tests.oracle_checks import check_behavior_learn_vs_oracle
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    _log("behavior_learn_cfg3_E32_agent0", check_behavior_learn_vs_oracle(_args(batch_size_run=32), 32, "cuda", seed=23, agents=(0,)))


def test_behavior_learn_config3_all_agents_vs_oracle():
    """the other four nets of the SAME launch (agents 1 .. 4: the tiles behind agent 0's in every arena, record and partial
    buffer, the last one at the arena's end) at the full config-3 size, another seed"""
    from tests.oracle_checks import check_behavior_learn_vs_oracle
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    _log("behavior_learn_cfg3_E32_agents1to4", check_behavior_learn_vs_oracle(_args(batch_size_run=32), 32, "cuda", seed=33, agents=(1, 2, 3, 4)))


def test_ppo_train_config3_one_agent_vs_oracle():
    """255 x 90 = 22 950 rows x F = 2485: one PPO epoch (gradients to 1e-5 of the fp64 oracle), then two epochs (the second
    runs on the first one's Adam-updated weights: post-train parameters asserted against the fp64 trajectory, the second
    epoch's gradients asserted against the fp64 oracle evaluated at the learner's own pre-step parameters)"""
    from tests.oracle_checks import check_ppo_train_vs_oracle
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    _log("ppo_train_cfg3_22950rows_agent0_1epoch", check_ppo_train_vs_oracle(_args(ppo_epoch=1), "cuda", seed=24, agents=(0,)))
    _log("ppo_train_cfg3_22950rows_agent0_2epochs", check_ppo_train_vs_oracle(_args(ppo_epoch=2), "cuda", seed=24, agents=(0,)))


def test_ppo_train_config3_all_agents_vs_oracle():
    """all five agents of the one fused launch at the full config-3 size (22 950 rows x F = 2485): one epoch on every agent,
    two epochs on the agents at the arena's ends"""
    from tests.oracle_checks import check_ppo_train_vs_oracle
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    _log("ppo_train_cfg3_22950rows_agents1to4_1epoch", check_ppo_train_vs_oracle(_args(ppo_epoch=1), "cuda", seed=34, agents=(1, 2, 3, 4)))
    _log("ppo_train_cfg3_22950rows_agents0and4_2epochs", check_ppo_train_vs_oracle(_args(ppo_epoch=2), "cuda", seed=35, agents=(0, 4)))


def test_ppo_train_config3_15_epochs_vs_oracle():
    """the optimiser-step count the benchmark times -- ppo_epoch = 15 (config/algs/ippo.yaml:6,
    learners/ippo_learner.py:286-303) at the full config-3 / config-4 size (22 950 rows x F = 2485), the agents at both arena ends.
    Gradients at the learner's own parameters in front of optimiser steps 8, 15 and one randomly drawn other step vs the fp64 oracle (<= max(1e-5, 1.5 e32)), one fp64
    Adam step from each of those states (moments at t = 8 and t = 15) <= 1e-5, hints <= 8 per step, and EVERY one of the 15 Adam
    updates of both agents replayed in fp64 from the step's own state and gradients (<= 1e-6)."""
    from tests.oracle_checks import check_ppo_train_vs_oracle
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    # ... and ONE MORE step from the thirteen others (VERDICT r5 "missing" 4: the gradient kernels of steps 2-7 and 9-14 ran
    # unchecked), drawn from the hash of the kernel sources: another step with every build, the same step for every run of one build
    # (a per-run draw would make the driver's -x suite a lottery on a comparison whose bound is statistical).  IPLAN_PPO_PROBE_STEP
    # (1-based) pins it; scripts/gpu_r6_final.sh runs steps 2 ... 14 once each on the round's final build (profiles/r06*_ppo_all_steps.json).
    import bench
    others = [k for k in range(14) if k != 7]
    extra = int(os.environ["IPLAN_PPO_PROBE_STEP"]) - 1 if os.environ.get("IPLAN_PPO_PROBE_STEP") else others[int(bench.csrc_sha16(), 16) % len(others)]
    w = check_ppo_train_vs_oracle(_args(ppo_epoch=15), "cuda", seed=54, agents=(0, 4), mid_probes=tuple(sorted({7, extra})), adam_replay=True)
    assert w["adam_replay_updates"] == 15 * 2 * 2, w
    w["random_mid_probe_step_1_based"] = extra + 1
    _log("ppo_train_cfg3_22950rows_agents0and4_15epochs_probes_at_8_and_15", w)


def test_config4_learners_on_one_gpu_vs_oracle():
    """BASELINE config 4 on ONE GPU (what the bench line's ``config4_n1`` times): 256 envs per rollout.
    Behavior_policy.learn switches to 128-env chunks there (nova/stable_behavior_policy.py: two launches' gradients accumulate under
    the all-env window normalisers) -- agent 3 replayed by the fp64 oracle in env shares of 32 (exact partition of the loss:
    tests/test_oracle_golden.py::test_behavior_learn_loss_env_shares_add_up); IPPOLearner.train on 255 x 90 rows of the
    256-episode buffer that ONE insert of a 256-env rollout fills (one epoch, agent 3)."""
    from tests.oracle_checks import check_behavior_learn_vs_oracle, check_ppo_train_vs_oracle
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    a = _args(batch_size_run=256)
    _log("behavior_learn_cfg4_E256_two_chunks_agent3",
         check_behavior_learn_vs_oracle(a, 256, "cuda", seed=55, agents=(3,), oracle_env_chunk=32, fp32_oracle=False))
    _log("ppo_train_cfg4_E256_one_insert_22950rows_agent3_1epoch",
         check_ppo_train_vs_oracle(_args(batch_size_run=256, ppo_epoch=1), "cuda", seed=56, agents=(3,)))


def test_prediction_learn_config3_vs_oracle():
    from tests.oracle_checks import check_prediction_learn_vs_oracle
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    _log("prediction_learn_cfg3_S64_N55", check_prediction_learn_vs_oracle(_args(batch_size_run=32), 32, "cuda", seed=25))     # all five agents


def test_config2_learners_vs_oracle():
    """config 2: IPPO-GAT, Behaviour off (F = 55 x 37 + 10 = 2045), 16 envs"""
    from tests.oracle_checks import check_ppo_train_vs_oracle, check_prediction_learn_vs_oracle
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    a = _args(Behavior_enable=False, batch_size_run=16, buffer_size=16, batch_size=15, ppo_epoch=1)
    _log("ppo_train_cfg2_F2045_agent1", check_ppo_train_vs_oracle(a, "cuda", seed=26, agents=(1,)))
    _log("prediction_learn_cfg2", check_prediction_learn_vs_oracle(a, 16, "cuda", seed=27, agents=(2,)))


def test_config5_gat_and_behaviour_vs_oracle():
    """config 5: 64 entities x 63 neighbours x obs_dim 128 -- GAT forward + backward; behaviour encoder / decoder learning at N = 64"""
    from tests.oracle_checks import check_gat_fwd_bwd_vs_oracle, check_behavior_learn_vs_oracle
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    _log("gat_fwd_bwd_cfg5_N64_D128_B4", check_gat_fwd_bwd_vs_oracle(B=4, N=64, D=128, device="cuda", seed=28))
    _log("behavior_learn_cfg5_N64", check_behavior_learn_vs_oracle(_args(max_vehicle_num=64, n_agents=2, episode_limit=30, batch_size_run=4),
                                                                   4, "cuda", seed=29))


@pytest.mark.parametrize("tag", ["ippo_train_mpe", "ippo_train"])
def test_ippo_reference_shaped_methods_gpu(golden, tag):
    """a13: get_value_ippo / eval_action_ippo / compute_returns / generate_data / ppo_update / _build_inputs_ippo on the GPU"""
    from tests.oracle_checks import check_ippo_reference_shaped_methods
    _log("reference_shaped_methods_" + tag, dict(post=check_ippo_reference_shaped_methods(golden(tag), "cuda")))


def test_ippo_train_unfilled_trailing_steps_gpu(golden):
    from tests.test_emu_learners import check_ippo_train_vs_oracle, unfilled_tail
    _log("ippo_train_unfilled_tail", dict(post=check_ippo_train_vs_oracle(golden("ippo_train"), "cuda", mutate=unfilled_tail)))


def test_lifted_restrictions_vs_oracle_gpu():
    """configurations the reference accepts beyond its shipped YAMLs: num_mini_batch > 1 (randperm minibatches),
    behavior_variation_penalty != 0 (stability term differentiated), weight_decay != 0"""
    from tests.oracle_checks import check_ppo_train_vs_oracle, check_behavior_learn_vs_oracle
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    a = _args(max_vehicle_num=9, n_agents=2, episode_limit=12, batch_size_run=4, buffer_size=6, batch_size=5, ppo_epoch=2, num_mini_batch=3)
    _log("ppo_minibatches_3x2", check_ppo_train_vs_oracle(a, "cuda", seed=41))
    b = _args(max_vehicle_num=9, n_agents=2, episode_limit=20, batch_size_run=4, behavior_variation_penalty=0.3, thres_small_variation=0.05)
    _log("behavior_learn_penalty_0.3", check_behavior_learn_vs_oracle(b, 4, "cuda", seed=42))


def test_deferred_decoder_update_gpu():
    """Behavior_policy.learn(defer_decoder=True), the form the benchmark loop uses: decoder weight gradients / all-reduce /
    clip / Adam on a side stream beside whatever follows.  Same gradients and parameters as the oracle, and two consecutive
    deferred calls end where two in-line calls end (parameters and Adam step counts)."""
    from tests.oracle_checks import check_behavior_learn_vs_oracle, check_deferred_equals_inline
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    b = _args(max_vehicle_num=9, n_agents=2, episode_limit=20, batch_size_run=4)
    _log("behavior_learn_deferred_decoder", check_behavior_learn_vs_oracle(b, 4, "cuda", seed=43, learn_kwargs=dict(defer_decoder=True)))
    check_deferred_equals_inline(b, 4, "cuda")


def test_ppo_loss_switches_vs_oracle():
    """the PPO loss switches off their shipped values (MSE, no value clipping, plain means, no GAE) on the real kernels"""
    from tests.oracle_checks import check_ppo_train_vs_oracle
    from tests.test_emu_learners import _small
    a = _small(ppo_epoch=2, use_huber_loss=False, use_clipped_value_loss=False, use_value_active_masks=False,
               use_policy_active_masks=False, use_gae=False)
    # Same bound as every learner check (tests/oracle_checks.py: max(1e-5, 1.5 x e32) plus the data-derived conditioning term of the
    # two row-sum tensors -- at seed 41 the critic's v_out.bias / rnn.norm.bias gradients are means of v - return that cancel to
    # 1 / 300 over the 27 rows); seed 42 is an ordinary draw of the same case.  Until round 5 this test carried a hand-set "6 x e32".
    w = check_ppo_train_vs_oracle(a, "cuda", seed=41)
    assert w["value_grad_row_sum_cond"] > 50, w
    _log("ppo_loss_switches_all_off", w)
    w2 = check_ppo_train_vs_oracle(a, "cuda", seed=42)
    assert w2["grad"] < 1e-5, w2
    _log("ppo_loss_switches_all_off_seed42", w2)


def test_behavior_learn_decoder_forward_first_form_gpu(monkeypatch):
    """IPLAN_DEC_FWD_V1=1 (the fp32-MFMA form of the decoder forward; the split-bf16 second form is the default and is what every
    other behaviour test here runs) at config-3 size (one agent, 32 envs, the whole episode) and at a ragged size (partial /
    missing tiles in the last workgroup of a net) vs the fp64 oracle"""
    from tests.oracle_checks import check_behavior_learn_vs_oracle
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    monkeypatch.setenv("IPLAN_DEC_FWD_V1", "1")
    _log("behavior_learn_cfg3_E32_agent0_dec_fwd_v1", check_behavior_learn_vs_oracle(_args(batch_size_run=32), 32, "cuda", seed=23, agents=(0,)))
    b = _args(max_vehicle_num=9, n_agents=2, episode_limit=20, batch_size_run=4)
    _log("behavior_learn_ragged_dec_fwd_v1", check_behavior_learn_vs_oracle(b, 4, "cuda", seed=43))


def test_fc1_split_vs_fp32_gpu():
    """the split-bf16 fc1 kernels of the PPO epochs at the full feature width (55 entities, F = 2485: 157 k-tiles, an odd
    count) against the fp32 MFMA kernels and an fp64 evaluation -- forward pre-activation and the fc1 / feature_norm gradients"""
    from tests.test_emu_kernels import check_fc1_split_vs_fp32
    check_fc1_split_vs_fp32("cuda", rows_per_ep=37, n_eps=29, seed=5, N=55, log=lambda w: _log("fc1_split_vs_fp32", w))


def test_behavior_learn_encoder_fp32_form_gpu(monkeypatch):
    """IPLAN_ENC_FP32=1 (the fp32-MFMA form of the behaviour encoder's forward and BPTT; the split-bf16 form is the default since
    round 4 and is what every other behaviour test here runs) at config-3 size and at a ragged size vs the fp64 oracle"""
    from tests.oracle_checks import check_behavior_learn_vs_oracle
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    monkeypatch.setenv("IPLAN_ENC_FP32", "1")
    _log("behavior_learn_cfg3_E32_agent2_enc_fp32", check_behavior_learn_vs_oracle(_args(batch_size_run=32), 32, "cuda", seed=23, agents=(2,)))
    b = _args(max_vehicle_num=9, n_agents=2, episode_limit=20, batch_size_run=4)
    _log("behavior_learn_ragged_enc_fp32", check_behavior_learn_vs_oracle(b, 4, "cuda", seed=43))


def test_other_runtime_dims_vs_oracle_gpu():
    """d = 7, Z = 5, L = 4, 4 actions, 3 agents, prediction horizon 3 (every run-time dimension off its shipped value) on the GPU"""
    from tests.test_emu_learners import check_other_runtime_dims
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    for k, w in check_other_runtime_dims("cuda").items():
        _log("other_runtime_dims_" + k, w)


def test_tanh_activation_vs_oracle_gpu(golden):
    """args.use_ReLU off (utils/mappo_utils/mlp.py:10): the actor/critic kernels' second activation -- the reference-recorded
    fixture (tests/golden/ippo_train_tanh.pt), IPPOLearner.train at the full config-3 size (one epoch, agent 2, split-bf16 fc1
    path) and the rollout body (actions bit-equal) against the oracle"""
    from tests.oracle_checks import check_ppo_train_vs_oracle
    from tests.rollout_oracle import check_rollout_body
    from tests.test_emu_learners import check_ippo_train
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    check_ippo_train(golden("ippo_train_tanh"), "cuda")
    _log("tanh_ppo_train_cfg3_22950rows_agent2_1epoch", check_ppo_train_vs_oracle(_args(ppo_epoch=1, use_ReLU=False), "cuda", seed=44, agents=(2,)))
    _log("tanh_rollout_body_cfg3_E32_T4", check_rollout_body(_args(episode_limit=4, batch_size_run=32, use_ReLU=False), 32, "cuda", seed=45))


@pytest.mark.parametrize("kw", ["1", "2"])
def test_rollout_body_config3_other_ksplit_wg(kw, monkeypatch):
    """the fused vector step with 1 / 2 workgroups per action-selection unit (the shapes larger env counts pick) at config 3 width"""
    from tests.rollout_oracle import check_rollout_body
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    monkeypatch.setenv("IPLAN_AC_KSPLIT_WG", kw)
    _log("rollout_body_cfg3_E32_T3_ksplit_wg" + kw, check_rollout_body(_args(episode_limit=3, batch_size_run=32), 32, "cuda", seed=46))


def test_rollout_body_config4_width_vs_oracle():
    """256 envs on one GPU (BASELINE config 4, N = 1: 1 280 scene + 560 encoder + 160 actor/critic workgroups per fused launch, the
    action selection's K split over one workgroup per unit): two vector steps against the oracle and the two-launch form"""
    from tests.rollout_oracle import check_rollout_body
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    _log("rollout_body_cfg4_E256_T2", check_rollout_body(_args(episode_limit=2, batch_size_run=256), 256, "cuda", seed=47))
