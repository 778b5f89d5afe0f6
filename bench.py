"""bench.py -- env-steps/sec of the iPLAN hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[2]: Heterogeneous-Highway chaotic, full iPLAN (Behaviour + GAT +
soft update), 5 agents, 32 parallel envs per GPU (configs[3] = the same sharded over 8 GPUs, weak
scaling), synthetic observation tensors already resident in HBM.
One "step" = one training cycle of the reference loop (run_ippo.py:261-332): buffer_size/E rollouts
of E envs x 90 steps (per vector step: select_actions_ippo, GAT_latent_update, latent_update,
episode-buffer writes), after every rollout Behavior_policy.learn + Prediction_policy.learn +
insert_episode_batch, and one IPPOLearner.train (15 PPO epochs) when the 256-episode buffer fills.
value = env transitions processed by all ranks / max-over-ranks wall time.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from iplan_amd.config import default_args  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_16x16x4_f32


def gat_algorithmic_flops(n_nets, B, N, D, H=32, A=32):
    """V(2DH + 24H^2 + 6HA + 12A^2) + P(12H^2 + 8H + 4A) per net-forward."""
    V, P = B * N, B * N * (N - 1)
    return n_nets * (V * (2 * D * H + 24 * H * H + 6 * H * A + 12 * A * A) + P * (12 * H * H + 8 * H + 4 * A))


def cpu_baseline(args, E, budget_s=20.0):
    """The oracle (a CPU port of the reference arithmetic) timed on this host's cores on a bounded
    sample of the same workload.  Reported beside the GPU number, never mixed into it."""
    from oracle import iplan_oracle as O
    from iplan_amd import synth
    # intra-op threads: the reference's hot path is thousands of tiny ATen ops; beyond ~16 threads the
    # fork/join overhead dominates (with 256 threads on the GPU box's 2 x 64-core EPYC one vector step
    # took minutes), so the baseline uses min(host cores, 16) threads and says so in `cores`.
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    nA, N, d, Z, A, L = args.n_agents, args.max_vehicle_num, args.obs_shape_single, args.latent_dim, args.attention_dim, args.max_history_len
    from iplan_amd.harness import SyntheticLoop
    cargs = default_args("highway", use_cuda=False)
    # parameters: random init of the same architectures (state_dict layout of the reference)
    from iplan_amd.nova.GAT_Net import GAT_Net
    from iplan_amd.nova.behavior_net import EncoderRNN
    from iplan_amd.modules.agents.ippo_actor import R_Actor
    from iplan_amd.modules.critics.ippo_critic import R_Critic
    sd = lambda m: {k: v.detach() for k, v in m.state_dict().items()}  # noqa: E731
    gat = [sd(GAT_Net(d + Z, cargs)) for _ in range(nA)]
    enc = [sd(EncoderRNN(d, 32, Z, 1)) for _ in range(nA)]
    F = N * (d + A + Z) + args.n_actions + nA
    act = [sd(R_Actor(F, cargs)) for _ in range(nA)]
    cri = [sd(R_Critic(F, cargs)) for _ in range(nA)]
    hist, window = synth.rollout_step_inputs(cargs, E, 0)
    hist, window = torch.as_tensor(hist, dtype=torch.float32), torch.as_tensor(window, dtype=torch.float32)
    att = torch.zeros(E, nA, N, A)
    lat = torch.full((E, nA, N, Z), 1.0 / Z)
    eh = torch.zeros(E, 1, nA, N, 32)
    ha = torch.zeros(E, nA, 64)

    def vector_step():
        nonlocal att, lat, eh
        with torch.no_grad():
            new_att = []
            for i in range(nA):
                noise = O.gumbel_noise_like_reference(E * N * (N - 1))
                new_att.append(O.gat_forward(gat[i], torch.cat([hist[:, i], lat[:, i]], -1), att[:, i].reshape(E * N, A), noise).reshape(E, N, A))
            att = torch.stack(new_att, 1)
            lat, eh = O.latent_update(enc, window, eh, lat, cargs.soft_update_coef)
            x = O.build_inputs_rollout(hist, att, lat, torch.zeros(E, nA, args.n_actions), nA)
            for i in range(nA):
                O.actor_logits(act[i], x[:, i], ha[:, i])
                O.critic_value(cri[i], x[:, i], ha[:, i])
    t0 = time.time()
    vector_step()                                   # warm-up (first call pays one-time init)
    warm = time.time() - t0
    budget_s = max(2.0, min(budget_s, 60.0 - warm))
    t0 = time.time()
    n = 0
    while n < 1 or (time.time() - t0 < budget_s and n < 200):
        vector_step()
        n += 1
    dt = time.time() - t0
    return dict(value=n * E / dt, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{n} rollout vector steps (E={E}, 5 agents x 55 entities: GAT_latent_update + latent_update + "
                       f"select_actions) of the oracle in {dt:.1f}s; learners not included in this sample")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=32, help="parallel envs per GPU (config 3: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    opt = ap.parse_args()
    if os.environ.get("IPLAN_BENCH_WATCHDOG"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["IPLAN_BENCH_WATCHDOG"]), repeat=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    args = default_args("highway", use_cuda=True, batch_size_run=opt.envs)
    E = opt.envs
    from iplan_amd.harness import SyntheticLoop
    loop = SyntheticLoop(args, E, seed=1234 + rank, device=dev)
    rollouts_per_step = max(1, args.buffer_size // (E * world)) if False else max(1, args.buffer_size // E)

    gat_ms = []

    def one_step(timed):
        for _ in range(rollouts_per_step):
            loop.rollout()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(opt.warmup):
        one_step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(opt.steps):
        one_step(True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    env_steps = opt.steps * rollouts_per_step * E * args.episode_limit * world

    # dominant kernel: fused GAT forward -- live HIP-event timing on the launch stream
    from iplan_amd import ops
    from iplan_amd.nova.GAT_Net import gumbel_noise
    nA, N, d, Z, A = args.n_agents, args.max_vehicle_num, args.obs_shape_single, args.latent_dim, args.attention_dim
    hist = loop.obs_sets[0]["hist"][0].permute(1, 0, 2, 3)
    lat = torch.softmax(torch.randn(nA, E, N, Z, device=dev), -1)
    hid = torch.randn(nA, E, N, A, device=dev) * 0.1
    noise = gumbel_noise((nA, E, N, N - 1, 2), dev)
    out = torch.empty(nA, E, N, A, device=dev)
    for _ in range(3):
        ops.gat_forward(loop.prediction.gat_arena, hist, lat, hid, noise, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    e0.record()
    for _ in range(iters):
        ops.gat_forward(loop.prediction.gat_arena, hist, lat, hid, noise, out=out)
    e1.record()
    torch.cuda.synchronize()
    gat_s = e0.elapsed_time(e1) / iters / 1e3
    flops = gat_algorithmic_flops(nA, E, N, d + Z)
    achieved = flops / gat_s / 1e12

    if rank == 0:
        line = {
            "metric": "env-steps/sec (whole node), Hetero-Highway chaotic 5-agent",
            "value": env_steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": opt.steps,
            "warmup": opt.warmup, "ms_per_step": dt / opt.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Highway chaotic full iPLAN, 5 agents x 55 entities, "
                                   f"{E} envs/GPU x 90 steps; ROLLOUT INFERENCE ONLY in this build "
                                   "(learners not yet in the timed region)",
                       "envs_per_gpu": E, "rollouts_per_step": rollouts_per_step},
            "roofline": {"kernel": "gat_fwd_kernel", "bound": "mfma", "achieved": achieved,
                         "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                         "traffic": None, "us_per_launch": gat_s * 1e6, "algorithmic_gflop_per_launch": flops / 1e9},
        }
        if not opt.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, E)
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
