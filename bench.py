"""bench.py -- env-steps/sec of the iPLAN hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[2]: Heterogeneous-Highway chaotic, full iPLAN (Behaviour + GAT +
soft update), 5 agents, 32 parallel envs per GPU (configs[3] = the same sharded over the GPUs of a
node: weak scaling, 32 envs per rank, PPO / prediction / behaviour gradients all-reduced over RCCL),
synthetic observation tensors already resident in HBM (the simulators cannot be
installed offline; env.step and observation_wrapper are outside the hot path).

One "step" = one full training cycle of the reference loop (run_ippo.py:261-285): buffer_size / E
rollouts of E envs x 90 steps (per vector step: select_actions_ippo, GAT_latent_update,
latent_update, episode-buffer writes), after every rollout insert_episode_batch +
Behavior_policy.learn + Prediction_policy.learn, and IPPOLearner.train (15 PPO epochs over 255 x 90
rows per agent) when the 256-episode buffer fills.  Nothing is skipped inside the timed region.
value = env transitions processed by all ranks / max-over-ranks wall time.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from iplan_amd.config import default_args  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_16x16x4_f32, dense)
BF16_MFMA_PEAK_TFLOPS = 2500.0       # same guide: v_mfma_f32_16x16x32_bf16, dense (the split-bf16 kernels issue 6 of them per fp32 product)
HBM_PEAK_GBS = 8000.0                # HBM3E spec (same guide; ~6.3 TB/s is what a float4 copy achieves)
# kernel sources that no launch of the benchmark cycle comes from (dead code in the reference, ablations, the real runner's history
# wrapper, the opt-in P2P collective): editing them does not invalidate a counter / kernel-statistics series of the cycle
NOT_IN_THE_CYCLE = ("seq2seq.hip", "mlp3.hip", "obs_history.hip", "p2p.hip")


def csrc_sha16():
    """Hash of the kernel sources the benchmark cycle's kernels are built from: every file of iplan_amd/csrc except
    NOT_IN_THE_CYCLE, plus the C header (a test-only or dead-code kernel must not invalidate the evidence)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "iplan_amd", "csrc", "*"))) + [os.path.join(ROOT, "include", "iplan_hip.h")]
    for f in files:
        if os.path.isfile(f) and not f.endswith((".o", ".so")) and os.path.basename(f) not in NOT_IN_THE_CYCLE:
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def roofline_coverage(rows, pmc_src_file):
    """Share of a cycle's KERNEL TIME the roofline rows account for, from the rocprofv3 kernel statistics committed with the
    counter summary of this build (profiles/<series>_full_cycle_kernel_stats.csv, the same command under --kernel-trace --stats).
    Every row names the profiler's kernel(s) it stands for (``profile_kernels``: substrings of the demangled name).  A row list
    that leaves more than 10 % of the kernel time unexplained is an error (the worst kernel was missing)."""
    import csv
    if pmc_src_file is None:
        return {"covered_frac": None, "note": "no kernel statistics of this build's kernel sources under profiles/"}
    f = pmc_src_file.replace("_pmc_summary.json", "_full_cycle_kernel_stats.csv")
    if not os.path.exists(f):
        return {"covered_frac": None, "note": f"{os.path.basename(f)} not found"}
    stats = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in stats)
    pats = [p for row in rows for p in row.get("profile_kernels", ())]
    cov = sum(float(r["TotalDurationNs"]) for r in stats if any(p in r["Name"] for p in pats))
    missing = sorted(((float(r["TotalDurationNs"]) / tot, r["Name"].split("(")[0][-60:]) for r in stats if not any(p in r["Name"] for p in pats)),
                     reverse=True)[:4]
    return {"covered_frac": cov / tot, "source": os.path.basename(f), "largest_uncovered": [{"frac": round(a, 4), "kernel": b} for a, b in missing]}


def load_pmc_traffic(envs_per_gpu):
    """``roofline.traffic`` comes from ONE place: the newest profiles/*_pmc_summary.json (scripts/gpu_pmc_all.sh +
    scripts/pmc_to_bench_json.py: rocprofv3 --pmc passes of the bench's own pieces, FETCH_SIZE and WRITE_SIZE in separate
    passes, HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE per the gfx950 correction of MI355X_MICROARCH.md).  Counters cannot be
    collected from inside bench.py, so the file is a committed measurement -- of a particular build: it records the hash of
    the kernel sources it was taken on, and a file whose hash is not this checkout's is REFUSED (traffic = null) rather than
    quoted for kernels it never saw.  -> ({bench key: bytes per launch}, source note)"""
    import glob
    best, sha = None, csrc_sha16()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        # the summary measured on THIS build's kernel sources wins, whatever its name sorts as; failing that the
        # lexicographically last one is named in the refusal below
        if best is None or d.get("csrc_sha16") == sha or best[1].get("csrc_sha16") != sha:
            best = (f, d)
    global PMC_FILE
    PMC_FILE = None
    if best is None:
        return {}, "no profiles/*_pmc_summary.json"
    f, d = best
    if d.get("csrc_sha16") != sha:
        return {}, f"{os.path.basename(f)} was measured on kernel sources {d.get('csrc_sha16')}, this build is {sha}: refused"
    if d.get("envs_per_gpu") != envs_per_gpu:
        return {}, f"{os.path.basename(f)} was measured at {d.get('envs_per_gpu')} envs per GPU"
    PMC_FILE = f
    return {k: v["bytes"] for k, v in d["per_launch"].items()}, f"{os.path.basename(f)} (series {d.get('series')}, kernel sources {d.get('csrc_sha16')})"


PMC_FILE = None                              # the counter summary load_pmc_traffic accepted (its series' kernel statistics: roofline_coverage)


def gat_algorithmic_flops(n_nets, B, N, D, H=32, A=32):
    """V(2DH + 24H^2 + 6HA + 12A^2) + P(12H^2 + 8H + 4A) per net-forward."""
    V, P = B * N, B * N * (N - 1)
    return n_nets * (V * (2 * D * H + 24 * H * H + 6 * H * A + 12 * A * A) + P * (12 * H * H + 8 * H + 4 * A))


def cpu_baseline(args, E):
    """The oracle (CPU port of the reference arithmetic, kind "port") timed on this host's cores on a BOUNDED sample of the
    same workload (oracle/cpu_baseline.py: a rollout vector step at full width plus one Behaviour / Prediction / PPO learner
    pass on a reduced number of envs / rows; warm-up + best of 3; scaled to seconds per env-step and summed).  Reported beside
    the GPU number, never mixed into it.  profiles/r06_cpu_baseline_anchor.json ties the oracle's speed to the REAL reference
    classes timed on identical inputs in the build container (the reference cannot travel to the GPU box)."""
    from oracle.cpu_baseline import measure
    # intra-op threads: the path is thousands of small ATen ops; beyond ~16 threads fork/join overhead dominates (with all
    # 256 hardware threads of the GPU box one vector step took minutes)
    cores = min(os.cpu_count() or 1, 16)
    # Behaviour learn (the dominant CPU leg) at the FULL env count of one rollout: its per-env cost keeps falling with the
    # batch (anchor file: behaviour_leg_linearity), so a small-batch sample would understate the CPU path
    m = measure("oracle", E, cores, Eb=min(E, 32), reps=2)       # (the behaviour leg: all agents, timed once)
    return dict(value=m["value"], unit="env-steps/s", cores=cores, kind="port",
                sample=m["sample"] + "; oracle-vs-reference speed on identical inputs (this round's oracle, build container): profiles/r06_cpu_baseline_anchor.json")


def config5_object():
    """BASELINE.json configs[4] ("Synthetic 64 agents x 63 neighbours x obs_dim 128, GAT+behavior forward/backward only, rocprof
    roofline run"): iplan_amd/config5.py's four measurements at B = 32 and B = 256 env rows (the same function scripts/cfg5_bench.py
    runs under rocprofv3 for profiles/*_cfg5_*), fewer repetitions.  Fractions are of the fp32 MFMA peak, algorithmic FLOPs by
    SURVEY.md section 8d's convention (backward = 2x forward)."""
    from iplan_amd.config5 import measure
    rows = measure((32, 256), ("gat", "beh"), reps=0.6)
    return {"what": "BASELINE configs[4]: 5 nets x B env rows x 64 entities x 63 neighbours x obs_dim 128; GAT forward (rollout form), GAT "
                    "forward + backward + weight gradients (training form), behaviour encoder / decoder forward and forward + BPTT + weight "
                    "gradients (90-step episode at B = 32, 30-step at B = 256); HIP events around each call, outside the timed region",
            "peak_tflops": FP32_MFMA_PEAK_TFLOPS,
            "rows": [{"piece": r["piece"], "B": r["B"], "ms": round(r["ms"], 4), "algorithmic_gflop": round(r["gflop"], 2),
                      "tflops": round(r["tflops"], 2), "frac": round(r["frac"], 4)} for r in rows]}


def strong_loop(total_envs, shard_world, rank, world, emu_world, dev, seed):
    """config 4: the env rows of ONE ``total_envs``-env run sharded over ``shard_world`` ranks; a rank stores its own episodes of
    the global buffer, and the reference's "first batch_size = buffer_size - 1 episodes" drops the last episode of the last rank
    (iplan_amd.parallel.shard_args: the same helper a run_ippo.py user calls)"""
    from iplan_amd.harness import SyntheticLoop
    from iplan_amd.parallel import shard_args
    assert total_envs % shard_world == 0, "--total-envs must be divisible by the number of GPUs"
    base = default_args("highway")
    drop = base.buffer_size - base.batch_size               # 1: train() uses the first buffer_size - 1 episodes
    union = default_args("highway", use_cuda=True, batch_size_run=total_envs, buffer_size=total_envs, batch_size=total_envs - drop)
    args = shard_args(union, shard_world, rank, "strong")   # (--emulate-rank-of W: rank 0 of W, never the rank that drops)
    E = args.batch_size_run
    return SyntheticLoop(args, E, seed=seed, device=dev), args, E, drop


def config4_n1_object(opt, dev, steps=3, warmup=1):
    """BASELINE.json configs[3] on ONE GPU (= `bench.py --scaling strong` at N = 1): 256 envs, one global 256-episode buffer,
    step = 1 rollout + the three learners (train() on 255 x 90 rows x 5 agents).  The denominator of the strong-scaling curve."""
    loop, args, E, _ = strong_loop(opt.total_envs, 1, 0, 1, 0, dev, 4321)
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(warmup):
            loop.cycle()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loop.cycle()
        loop.finish()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    del loop
    torch.cuda.empty_cache()
    return {"what": f"BASELINE configs[3] on one GPU ({E} envs, one global {E}-episode buffer; 1 rollout + Behavior_policy.learn in two "
                    "128-env chunks + Prediction_policy.learn + IPPOLearner.train per step), run after the timed region",
            "steps": steps, "warmup": warmup, "ms_per_step": dt * 1e3, "value": E * args.episode_limit / dt, "unit": "env-steps/s"}


def projection_split(loop, args, E, dev, dist, step_s, emu_world, opt):
    """``--emulate-rank-of W``: where the rank step goes, and what it projects to.  In the cycle (events on the main stream, the
    side learners run beside): the rollout, then everything up to the join before the next rollout.  Alone (nothing else on the GPU):
    the three learners and the step's collectives (the same count and sizes, back to back on the 1-rank RCCL group: launch cost
    only).  ``projected_speedup`` = config 4's N = 1 step on THIS box / this rank's step."""
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    reps = 4

    def alone(fn, pre=None):
        tot = 0.0
        for it in range(reps + 1):
            if pre is not None:
                pre()
            torch.cuda.synchronize()
            e0, e1 = ev(), ev()
            e0.record()
            with contextlib.redirect_stdout(io.StringIO()):
                r = fn()
            if callable(r):
                r()
            e1.record()
            torch.cuda.synchronize()
            if it:
                tot += e0.elapsed_time(e1)
        return tot / reps
    with contextlib.redirect_stdout(io.StringIO()):
        loop.phase_events = []
        for _ in range(reps):
            loop.cycle()
        loop.finish()
        torch.cuda.synchronize()
        ph = loop.phase_events
        loop.phase_events = None
    roll = sum(a.elapsed_time(b) for a, b, _ in ph) / len(ph)
    rest = sum(b.elapsed_time(c) for _, b, c in ph) / len(ph)
    batch = loop.rollout()
    t_roll = alone(lambda: loop.rollout())

    def beh():
        loop.behavior.learn(batch, 0, defer_decoder=True)
        loop.behavior.join_decoder()
    t_beh = alone(beh)
    t_pred = alone(lambda: loop.prediction.learn(batch, 0))

    def fill():
        loop.learner.store.clear()
        loop.learner.insert_episode_batch(batch)
    t_ppo = alone(lambda: loop.learner.train(0), pre=fill)
    # the step's collectives: behaviour 2 arenas (+ the [nA, J] window sums), prediction 2 (+ [nA]), PPO 2 x 15 (+ 3 scalars vectors)
    arenas = [loop.behavior.enc_arena, loop.behavior.dec_arena, loop.prediction.gat_arena, loop.prediction.dec_arena]
    ppo = [loop.mac.actor_arena, loop.mac.critic_arena]
    small = torch.zeros(args.n_agents, device=dev)
    dp = loop.learner.dp

    def colls():
        for a in arenas:
            dp.all_reduce_grads(a)
        for _ in range(args.ppo_epoch):
            dp.all_reduce_grads(*ppo)
        for _ in range(5):
            dp.all_reduce_sum(small)
    t_coll = alone(colls)
    n1 = config4_n1_object(opt, dev)
    return {"rank_step_ms": step_s * 1e3,
            "in_cycle_ms": {"rollout": roll, "learn_phase_until_next_rollout": rest,
                            "note": "HIP events on the main stream; Behavior_policy.learn on the main stream with Prediction_policy.learn and "
                                    "IPPOLearner.train beside it on two side streams, the deferred decoder update beside the next rollout"},
            "alone_ms": {"rollout": t_roll, "behaviour_learn": t_beh, "prediction_learn": t_pred, "ppo_train": t_ppo,
                         "collectives_launch_only": t_coll,
                         "note": f"each piece alone on the GPU; collectives = the step's {len(arenas) + 2 * args.ppo_epoch} gradient-arena and 5 "
                                 "normaliser all-reduces issued back to back on the 1-rank RCCL group"},
            "config4_n1_same_box": n1,
            "projected_speedup": n1["ms_per_step"] / (step_s * 1e3),
            "target": "north_star: >= 6x strong scaling at 8 GPUs (rank step <= config4_n1 step / 6)"}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_argv(n, port, argv):
    """the command line ``bench.py --gpus n`` re-executes itself under: one rank per GPU of ONE node, rendezvous on 127.0.0.1
    (the container's hostname may not resolve) -- the same line the driver uses for N > 1 (README.md quick start)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(opt, argv):
    """``python bench.py --gpus N`` WITHOUT a launcher around it (no WORLD_SIZE in the environment): start the N ranks here.
    N = 1 goes the same way, so the spawn path is the one the driver's single-GPU line exercises every round.  The ranks'
    stdout / stderr pass through (rank 0 prints the one JSON line), the exit code is the launcher's (non-zero if any rank died).
    More ranks than visible GPUs is an error that says so -- never a silent smaller run."""
    import subprocess
    cmd = spawn_argv(opt.gpus, free_port(), argv)
    if opt.print_launch:
        print(json.dumps({"launch_argv": cmd}))
        return 0
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if opt.gpus > have:
        print(f"bench.py: --gpus {opt.gpus} asks for {opt.gpus} ranks, one per GPU, but this process sees {have} GPU(s) "
              f"(torch.cuda.device_count(); HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES', '<unset>')}): not running a smaller job "
              "under the name of a larger one", file=sys.stderr)
        return 2
    env = dict(os.environ, IPLAN_BENCH_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's cross-process buffer sharing needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(1, opt.gpus))))   # (torchrun would set 1: the cpu_baseline leg wants the cores)
    if opt.gpus > 1:
        return subprocess.run(cmd, env=env).returncode
    # N = 1: the rank's stdout is held back until it has ended well -- if the launcher itself fails on this box (no free port, a broken
    # elastic agent ...) the single-GPU line is still measured, in this process, and says so (launcher.spawn_failed); a failure of the
    # BENCHMARK shows up in both forms and is not masked
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode == 0 and lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
        return 0
    print(f"bench.py: the spawned rank ended with rc = {r.returncode} and {len(lines)} JSON line(s); measuring in this process instead",
          file=sys.stderr)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks = GPUs of this node.  Under a launcher (WORLD_SIZE set: the driver's torch.distributed.run line) it must equal "
                         "WORLD_SIZE; without one, bench.py starts the N ranks itself (launch_ranks)")
    ap.add_argument("--print-launch", action="store_true", help="print the torch.distributed.run command line --gpus N would start, and exit")
    ap.add_argument("--in-process", action="store_true",
                    help="single rank in THIS process, no launcher and no process group (profilers that attach to one process: "
                         "scripts/gpu_r6_final.sh runs rocprofv3 around this form; the line says launcher.spawned = false)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=32, help="parallel envs per GPU in weak-scaling mode (config 3: 32)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): 32 envs per GPU, every rank its own 256-episode PPO buffer.  strong = BASELINE config 4: "
                         "--total-envs (256) envs in total, sharded over the GPUs, ONE global 256-episode buffer (train() after every "
                         "rollout, the first 255 episodes trained on), all gradients all-reduced")
    ap.add_argument("--total-envs", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-rank-of", type=int, default=0, metavar="W",
                    help="diagnostic, 1 GPU, with --scaling strong: run ONE rank's share of config 4 on W GPUs (total-envs / W envs, its "
                         "slice of the global buffer, the union's normalisers, every collective launched on a 1-rank RCCL group) and "
                         "report the PROJECTED W-GPU strong-scaling figure next to this GPU's own number")
    ap.add_argument("--rollout-only", action="store_true", help="diagnostic: time rollout inference without the learners")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the two objects appended AFTER the timed region at N = 1 (outside `value`): `config5` (BASELINE configs[4], "
                         "iplan_amd/config5.py at B = 32 / 256) and `config4_n1` (configs[3]'s 256 envs + global buffer on this one GPU)")
    opt = ap.parse_args()
    if os.environ.get("IPLAN_BENCH_WATCHDOG"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["IPLAN_BENCH_WATCHDOG"]), repeat=True)

    # Who starts the ranks.  Under a launcher (the driver's torch.distributed.run line, or our own launch_ranks) WORLD_SIZE is set
    # and must agree with --gpus; without one this process IS the launcher.  The single-process diagnostics (--in-process,
    # --emulate-rank-of) stay here.
    launched = "WORLD_SIZE" in os.environ
    single_process = opt.in_process or opt.emulate_rank_of > 1
    spawn_failed = False
    if opt.print_launch or (not launched and not single_process):
        rc = launch_ranks(opt, [a for a in sys.argv[1:] if a != "--print-launch"])
        if rc is not None:
            sys.exit(rc)
        spawn_failed = True                                 # (N = 1 only: fall through to the in-process form)
    if launched and os.environ.get("IPLAN_BENCH_SPAWNED") and os.environ.get("IPLAN_BENCH_TEST_FAIL_IN_RANK"):
        sys.exit(3)                                         # (tests/test_bench_launcher.py: a launcher that dies -> the N = 1 fallback)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if opt.gpus != world:
        print(f"bench.py: --gpus {opt.gpus} but " + (f"the launcher started WORLD_SIZE = {world} rank(s)" if launched else
              "--in-process / --emulate-rank-of run ONE rank in this process") + ": refusing to print a line under the wrong n_gpus",
              file=sys.stderr)
        sys.exit(2)
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if local_rank >= have:
        print(f"bench.py: rank {rank} (LOCAL_RANK {local_rank}) has no GPU of its own: {have} visible", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    rccl = None
    emu_world = opt.emulate_rank_of if (opt.emulate_rank_of > 1 and world == 1 and opt.scaling == "strong") else 0
    def init_rccl(**kw):
        """one rank per GPU over RCCL ("nccl" IS RCCL on ROCm); the first collective builds the communicator outside the timed region.
        RCCL prints a version banner on fd 1 when it initialises: stdout carries ONE JSON line, so fd 1 points at stderr meanwhile."""
        import torch.distributed as dist
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev, **kw)
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            try:                                            # the banner sits in C stdio's buffer (fd 1 is a pipe / file: fully buffered) and
                import ctypes                               # would come out at exit, BEHIND the JSON line: flush it while fd 1 is stderr
                ctypes.CDLL(None).fflush(None)
            except Exception:                               # noqa: BLE001
                pass
            os.dup2(keep, 1)
            os.close(keep)
        info = {"rccl_ranks": dist.get_world_size(), "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()),
                "all_reduce_of_ones": float(probe.item()), "backend": dist.get_backend()}
        assert info["all_reduce_of_ones"] == dist.get_world_size(), info
        return dist, info

    # N = 1 under a launcher: the process group is created AFTER the timed region (below) -- its communicator's streams would
    # otherwise take hardware queues ahead of the training cycle's own streams and change which of those share a queue
    # (4 hardware queues by default; measured: 272.7 ms per cycle with the group alive against 263-264 ms without,
    # profiles/r06_notes.md) -- the single-GPU line then times exactly what `--in-process` times.  IPLAN_BENCH_PG_EARLY=1: A/B.
    pg_late = launched and world == 1 and not os.environ.get("IPLAN_BENCH_PG_EARLY")
    if launched and not pg_late:
        dist, rccl = init_rccl()
    elif emu_world:
        dist, _ = init_rccl(init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1)

    strong = opt.scaling == "strong"
    shard_world = emu_world or world                             # how many ranks the global run is split over
    from iplan_amd.harness import SyntheticLoop
    if strong:
        loop, args, E, drop = strong_loop(opt.total_envs, shard_world, rank, world, emu_world, dev, 1234 + rank)
    else:
        E = opt.envs
        args = default_args("highway", use_cuda=True, batch_size_run=E)
        loop = SyntheticLoop(args, E, seed=1234 + rank, device=dev)
    if world > 1 or emu_world:
        from iplan_amd.parallel import DataParallel
        DataParallel(dist.group.WORLD).attach(loop)               # (strong mode: the union's row counts ride on shard_args' namespace)
        if strong:
            assert loop.learner.dp_global_rows == (args.buffer_size * shard_world - drop) * args.episode_limit
            assert loop.learner.dp_global_count == args.buffer_size * shard_world * args.episode_limit
    rollouts_per_step = max(1, args.buffer_size // E)

    def one_step():
        with contextlib.redirect_stdout(io.StringIO()):            # the reference prints "TRAINING IPPO"
            for _ in range(rollouts_per_step):
                if opt.rollout_only:
                    loop.rollout()
                else:
                    loop.cycle()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(opt.warmup):
        one_step()
    barrier()
    # roofline timing, live in the timed region: HIP events on the launch stream right around the launches of the kernels that
    # make up > 5 % of a cycle (ops.KernelTimers).  EVERY gat_fwd launch is bracketed: a rollout's first launches queue behind
    # the decoder update the previous learn() deferred (2.8 ms for the unlucky one), and a 1-in-8 sample whose phase drifts
    # against the 91 launches of a rollout over-weights them (275 us sampled vs 183 us in the rocprofv3 statistics)
    from iplan_amd import ops
    ops.TIMERS = ops.KernelTimers(every={"ac_fwd_kernel:rollout": 8})
    t0 = time.perf_counter()
    for _ in range(opt.steps):
        one_step()
    if hasattr(loop, "finish"):
        loop.finish()                                       # the last cycle's staged loss read-backs (host run-ahead, harness.cycle)
    barrier()
    dt = time.perf_counter() - t0
    timed = ops.TIMERS.summary()
    from iplan_amd.streams import probe_mode
    queue_probe = {"mode": probe_mode(), "streams_replaced_after_first_cycle": getattr(loop, "queue_repairs", None)}
    scene_clocks = ops.TIMERS.clocks.get("gat_scenes_clocks", [])
    if scene_clocks:                                        # span of the scenes inside the sampled fused launches (100 MHz stamps)
        spans = [(c.view(-1, 5)[:, 4].max() - c.view(-1, 5)[:, 0].min()).item() / 100.0 for c in scene_clocks]
        timed["gat_scenes_span_us"] = (len(spans), sum(spans) / len(spans), 0.0)
    ops.TIMERS = None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    env_steps = opt.steps * rollouts_per_step * E * args.episode_limit * world

    if rank == 0:
        mode = (f"config 4 (strong scaling): {opt.total_envs} envs in total = {E} envs/GPU x {world} GPU(s), one global "
                f"{opt.total_envs}-episode buffer; step = 1 rollout + insert + Behavior_policy.learn + Prediction_policy.learn + "
                "IPPOLearner.train (15 epochs over the first 255 episodes x 90 rows x 5 agents), gradients all-reduced"
                if strong else
                f"{E} envs/GPU x 90 steps; step = {rollouts_per_step} rollouts, each followed by insert + Behavior_policy.learn + "
                "Prediction_policy.learn, then IPPOLearner.train (15 epochs x 255 x 90 rows x 5 agents)")
        rl = rooflines(args, E, timed, opt, rollouts_per_step)
        line = {
            "metric": "env-steps/sec (whole node), Hetero-Highway chaotic 5-agent",
            "value": env_steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": opt.steps,
            "warmup": opt.warmup, "ms_per_step": dt / opt.steps * 1e3, "higher_is_better": True,
            "scaling": opt.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Highway chaotic full iPLAN (Behaviour + GAT + soft update), 5 agents x 55 entities, " + mode
                                   + (" [ROLLOUT ONLY diagnostic]" if opt.rollout_only else ""),
                       "envs_per_gpu": E, "rollouts_per_step": rollouts_per_step,
                       "env_steps_per_step": rollouts_per_step * E * args.episode_limit * world},
            "roofline": rl[0], "roofline_others": rl[1:],
            # how the ranks came to be: spawned_by_bench = this line's ranks were started by bench.py's own launch_ranks()
            "launcher": {"launched": launched, "spawned_by_bench": bool(os.environ.get("IPLAN_BENCH_SPAWNED")), "world_size": world,
                         **({"spawn_failed": True} if spawn_failed else {}), "hardware_queue_probe": queue_probe,
                         **(rccl or {"rccl_ranks": None, "note": "single process, no process group (--in-process / --emulate-rank-of)"})},
        }
        if emu_world:
            per_rank = env_steps / dt
            line["projection"] = {
                "what": f"PROJECTED {emu_world}-GPU strong-scaling figure of config 4, not a measurement: this GPU ran rank 0's share "
                        f"({E} of {opt.total_envs} envs, its {args.buffer_size} episodes of the global buffer, loss normalisers of the union, "
                        "every gradient / normaliser collective launched on a 1-rank RCCL group, i.e. at its launch cost without any "
                        "xGMI transfer or straggler wait); projected value = total envs x 90 / this rank's step time",
                "emulated_world": emu_world, "projected_value": per_rank * emu_world, "unit": "env-steps/s",
                "not_included": "xGMI transfer and rank skew of the 2 + 2 + 30 small all-reduces per step (0.3-4 MB each)"}
        if emu_world and not opt.no_extras:
            line["projection"].update(projection_split(loop, args, E, dev, dist, dt / opt.steps, emu_world, opt))
        if world == 1 and not emu_world and not strong and not opt.rollout_only and not opt.no_extras:
            # BASELINE.json's other GPU configs on the same box, same build, AFTER the timed region (never part of `value`)
            del loop
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            line["config5"] = config5_object()
            line["config4_n1"] = config4_n1_object(opt, dev)
        if pg_late:
            torch.cuda.synchronize()
            try:
                dist, rccl = init_rccl()
                rccl["note"] = "1-rank group created after the timed region (see bench.py: pg_late)"
                line["launcher"].update(rccl)
            except Exception as e:                               # noqa: BLE001 -- the measurement is done: a group that cannot be
                dist = None                                      # created must not cost the single-GPU line
                line["launcher"]["rccl_error"] = f"{type(e).__name__}: {str(e)[:200]}"
        if not opt.no_cpu_baseline and world == 1:               # (N = 1 only: at N > 1 the other ranks would sit in RCCL's teardown meanwhile)
            line["cpu_baseline"] = cpu_baseline(args, E)

        def clean(o):                                            # a kernel that was not launched (diagnostic modes): null, not NaN
            if isinstance(o, float) and o != o:
                return None
            if isinstance(o, dict):
                return {k: clean(v) for k, v in o.items()}
            if isinstance(o, list):
                return [clean(v) for v in o]
            return o
        print(json.dumps(clean(line)))
    if dist is not None and dist.is_initialized():
        dist.destroy_process_group()


def rooflines(args, E, timed, opt, rollouts_per_step):
    """Roofline entries of every kernel above 5 % of a cycle's kernel time (profiles/history/r02*_full_cycle_kernel_stats.csv):
    ALGORITHMIC work per launch (SURVEY.md §8d convention) / the in-situ mean launch duration.  ``traffic`` =
    HBM bytes per launch from profiles/<series>_pmc_summary.json (load_pmc_traffic: null when that file is not of this build)."""
    nA, N, d, Z = args.n_agents, args.max_vehicle_num, args.obs_shape_single, args.latent_dim
    Hd, R, Lw, M, T = args.decoder_rnn_dim, args.encoder_rnn_dim, args.max_history_len, args.rnn_hidden_dim, args.episode_limit
    J = T - 1 - Lw
    V = E * N
    F = N * (d + args.attention_dim + Z) + args.n_actions + nA
    nan = float("nan")

    def entry(kernel, key, bound, work, unit_scale, peak, unit, note, traffic_key=None, work_from_timer=False, prof=(), work_scale=1.0):
        n, sec, w = timed.get(key, (0, nan, 0.0))
        if work_from_timer:
            work = w * work_scale
        ach = work / sec / unit_scale if n else nan
        tr = traffic_of(traffic_key or kernel)
        # no row may imply more than the HBM peak (a counter file matched to the wrong launches did): the PMC figure is of
        # another run of the same kernels, so allow for box-to-box spread in the duration (8 %), nothing more
        rejected = None
        if tr is not None and n and sec > 0 and tr / sec / 1e9 > 1.08 * HBM_PEAK_GBS:
            rejected = f"PMC figure {tr} B over the measured {sec * 1e6:.1f} us would be {tr / sec / 1e12:.2f} TB/s > the HBM peak: not reported"
            tr = None
        return {"kernel": kernel, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                "traffic": tr, **({"traffic_rejected": rejected} if rejected else {}), "us_per_launch": sec * 1e6 if n else nan,
                "traffic_source": pmc_src,
                "launches_timed": n, ("algorithmic_gbyte_per_launch" if bound == "hbm" else "algorithmic_gflop_per_launch"): work / 1e9,
                "profile_kernels": list(prof), "note": note}

    pmc, pmc_src = load_pmc_traffic(E)

    def traffic_of(key):
        key = key.replace("fwd2_kernel", "fwd_kernel").replace("bwd2_kernel", "bwd_kernel")      # (the PMC summary's keys name the kernel family)
        t = pmc.get(key)
        if t is not None and key == "iplan_wgrad:beh_dec":
            t = t // pieces(key)
        return t

    def pieces(key):
        n = timed.get(key, (0, nan, 0.0))[0]
        learns = max(1, opt.steps * rollouts_per_step * (1 if E <= 128 else -(-E // 128)))
        return max(1, round(n / learns))

    f_dec = nA * V * J * Lw * (2 * (d + Z) * Hd + 12 * Hd * Hd + 2 * Hd * d)            # decoder, all windows, one pass
    f_enc = nA * V * J * (Lw * (2 * d * R + 12 * R * R) + 2 * R * Z)                     # encoder, all windows, one pass
    if E > 128:                                                                         # env-chunked behaviour learning
        f_dec, f_enc = f_dec * 128 / E, f_enc * 128 / E
    rows = args.batch_size * T
    kpad32 = -(-(-(-N * d // 16) + -(-N * args.attention_dim // 16) + -(-N * Z // 16) + -(-(args.n_actions + nA) // 16)) // 2) * 32
    f_fc1 = nA * 2.0 * rows * F * 2 * M                                                  # fc1 of actor + critic: 2 F M FLOP per row and net
    b_fc1 = nA * (4.0 * rows * kpad32 + 2 * 4.0 * rows * M)                              # fragments once + z1 of both nets
    b_tail = nA * 2 * 4.0 * rows * (M + M + 648)                                         # z1 + h in, the activation record out
    fused_ac = not os.environ.get("IPLAN_NO_FUSE_AC") and not os.environ.get("IPLAN_NO_FUSE_ENC")      # harness._rollout_body
    # actor + critic forward of one vector step (rows (2 F M + 14 M^2 + 2 M n_out) per net)
    f_ac = (nA * E * (2 * (2.0 * F * M + 14.0 * M * M) + 2.0 * M * (args.n_actions + 1))) if fused_ac else 0.0
    S_pred = args.pred_batch_size
    f_gat_learn = gat_algorithmic_flops(nA, S_pred, N, d + Z)                            # one GAT forward at Prediction_policy.learn's shape
    out = [
        entry("gat_enc_ac_fwd_kernel" if fused_ac else "gat_enc_fwd_kernel", "gat_enc_ac_fwd_kernel" if fused_ac else "gat_fwd_kernel", "mfma",
              gat_algorithmic_flops(nA, E, N, d + Z) + nA * V * (Lw * (2 * d * R + 12 * R * R) + 2 * R * Z) + f_ac, 1e12, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s",
              f"the rollout's vector step as one launch: GAT_latent_update (5 nets x {E} envs x 55 entities = 160 scene workgroups, 6.41 GFLOP) "
              f"+ the behaviour encoder's latent_update behind them (1.1 GFLOP)"
              + (f" + the NEXT step's select_actions_ippo as the grid's last workgroups ({f_ac / 1e9:.2f} GFLOP; it waits for the scenes, so the "
                 "launch lasts GAT + the action selection's tail)" if fused_ac else "")
              + (f", {rollouts_per_step * (T - 1)} launches per step (an episode's first action selection, its initial GAT update and its "
                 "last latent updates are launches of their own)" if fused_ac else
                 f", {rollouts_per_step * (T + 1)} launches per step (the episode-initial GAT update of each rollout runs alone)")
              + "; fp32 results: "
              "the 54-step bi-GRU recurrence (84 % of the algorithmic FLOPs) is issued as 6 bf16 piece products per fp32 product "
              "on the bf16 matrix cores (fp32-exact split, DESIGN.md section 4), the rest as fp32 MFMA; peak = the fp32 MFMA / vector peak",
              prof=("gat_enc_ac_fwd_kernel", "gat_enc_fwd_kernel")),
        entry("beh_dec_bwd_kernel" if os.environ.get("IPLAN_DEC_BWD_V1") else "beh_dec_bwd2_kernel", "beh_dec_bwd_kernel", "mfma", f_dec / pieces("beh_dec_bwd_kernel"), 1e12, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s",
              f"decoder BPTT of Behavior_policy.learn (backward-data pass = 1x the forward FLOPs) in {pieces('beh_dec_bwd_kernel')} "
              "window-range launches per learn(), beside the weight-gradient contraction and the encoder BPTT of the previous range",
              prof=("beh_dec_bwd2_kernel", "beh_dec_bwd_kernel", "beh_dec_thin_grad_kernel")),
        entry("beh_dec_fwd_kernel" if os.environ.get("IPLAN_DEC_FWD_V1") else "beh_dec_fwd2_kernel", "beh_dec_fwd_kernel", "mfma", f_dec / pieces("beh_dec_fwd_kernel"), 1e12, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s",
              f"decoder forward of Behavior_policy.learn in {pieces('beh_dec_fwd_kernel')} window-range launches, beside the encoder forward",
              prof=("beh_dec_fwd2_kernel", "beh_dec_fwd_kernel")),
        entry("beh_enc_bwd_kernel", "beh_enc_bwd_kernel", "mfma", 2 * f_enc / pieces("beh_enc_bwd_kernel"), 1e12, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s",
              "encoder BPTT with in-kernel weight gradients (2x the forward FLOPs), side stream, beside the decoder BPTT: the two share the chip, "
              "so the in-cycle duration is not the kernel's own (alone: scripts/microbench.py)", prof=("beh_enc_bwd_kernel", "beh_enc_grad_kernel")),
        entry("beh_enc_fwd_kernel", "beh_enc_fwd_kernel", "mfma", f_enc / pieces("beh_enc_fwd_kernel"), 1e12, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s",
              f"encoder forward of Behavior_policy.learn ({J} windows x {Lw} GRU32 steps per chain, hidden state carried; SURVEY 8d: "
              f"{f_enc / nA / J / 1e6:.1f} MFLOP per net and window) in {pieces('beh_enc_fwd_kernel')} window-range launches on a side stream, "
              "beside the decoder forward (same remark as the encoder BPTT)", prof=("beh_enc_fwd_kernel",)),
        entry("wgrad_pair_bf16_kernel (+reduce)", "iplan_wgrad:beh_dec", "mfma", 0.0, 1e12, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s",
              "the deferred iplan_wgrad call of Behavior_policy.learn: the decoder GRU's W_ih and W_hh gradients over all (row, step) records as "
              "ONE paired launch (the [dr dz] columns both need are fetched once) + reduction, side stream, beside the next rollout's first "
              "steps.  Arithmetic intensity 2 x 192 x 64 x 2 FLOP per 1 536 B = 32 FLOP/B, above the fp32 ridge (19.7): the binding roof is the "
              "matrix pipe -- this row is against the fp32 MFMA peak (as every fp32-result row); the kernel issues its products in split-bf16 "
              "form, and what it is short of is ISSUE time: 1 395 VALU instructions (the bf16 splitting of 12 operand tiles) beside 288 MFMAs "
              "per 32-row block and wave.  Next row: the same launches against the HBM roof",
              traffic_key="iplan_wgrad:beh_dec", work_from_timer=True, work_scale=2.0 * 192 * 64 * 2 / 1536.0,
              prof=("wgrad_pair_bf16_kernel", "wgrad_reduce_kernel")),
        entry("wgrad_pair_bf16_kernel (+reduce) (HBM roof)", "iplan_wgrad:beh_dec", "hbm", 0.0, 1e9, HBM_PEAK_GBS, "GB/s",
              "same launches against the HBM roof: algorithmic bytes = every operand column of every row once, 4 (256 + 64 + 64) bytes per row "
              "and step", traffic_key="iplan_wgrad:beh_dec", work_from_timer=True),
        entry("ac_fc1_split_fwd_kernel", "ac_fc1_split_fwd", "mfma", f_fc1, 1e12, BF16_MFMA_PEAK_TFLOPS / 6.0, "TFLOP/s",
              f"fc1 of the actor AND the critic of one PPO epoch ({rows} rows x F = {F} x 128 outputs, 5 agents) over the packed normalised "
              "features, fp32-exact split-bf16: 6 bf16 piece products per fp32 product, so peak = the dense bf16 MFMA peak / 6 in "
              "fp32-equivalent FLOPs (the timed span includes the two small weight-piece kernels in front of the contraction)",
              prof=("ac_fc1_split_fwd_kernel", "ac_fc1_wprep_kernel", "ac_fc1_wsplit", "ac_fc1_wbeta")),
        entry("ac_fc1_split_fwd_kernel (HBM roof)", "ac_fc1_split_fwd", "hbm", b_fc1, 1e9, HBM_PEAK_GBS, "GB/s",
              "same launches against the HBM roof: the fp32 feature fragments read once for both nets + the pre-activations written",
              traffic_key="ac_fc1_split_fwd"),
        entry("ac_fc1_split_wgrad_kernel", "ac_fc1_split_wgrad", "mfma", f_fc1, 1e12, BF16_MFMA_PEAK_TFLOPS / 6.0, "TFLOP/s",
              "fc1 weight gradient G = dz1^T xhat of both nets of one PPO epoch, same split-bf16 form (K = rows)",
              prof=("ac_fc1_split_wgrad_kernel", "ac_fc1_finalize_kernel")),
        entry("ac_fwd_kernel<PPO tail>", "ac_fwd_kernel:train", "hbm", b_tail, 1e9, HBM_PEAK_GBS, "GB/s",
              f"the 64-wide tail of the actor + critic forward of one PPO epoch from the stored fc1 pre-activations ({rows} rows, 5 agents): "
              "reads z1 and the stored GRU state, writes the 648-float activation record per row and net",
              traffic_key="ac_fwd_kernel:train", prof=("ac_fwd_kernel<2, true",)),
        entry("ac_bwd_tail_kernel", "ac_bwd_tail_kernel", "hbm", 0.0, 1e9, HBM_PEAK_GBS, "GB/s",
              "the 64-wide tail of the actor + critic backward of one PPO epoch: reads 8 of the record's 10 rows per net and row + the hidden "
              "state, writes the 400-float row gradients the weight-gradient contractions read, weights (transposed) staged in LDS",
              traffic_key="ac_bwd_tail_kernel", work_from_timer=True, prof=("ac_bwd_tail_kernel",)),
        entry("gat_fwd_kernel", "gat_fwd_kernel", "mfma", 0.0, 1e12, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s",
              f"GAT launches outside the fused vector step: the saving forward of Prediction_policy.learn ({S_pred} sampled rows x 5 nets, "
              f"{f_gat_learn / 1e9:.2f} GFLOP), each rollout's episode-initial update and its last step's latent updates ({E} envs, 6.41 GFLOP); "
              "work = the mean over the timed launches of each launch's own algorithmic FLOPs.  The "
              "in-cycle duration is NOT the kernel's: the episode-initial launch queues behind the deferred decoder contraction that holds "
              "every SIMD at the rollout's head (DESIGN.md section 9), and the events / the profiler's span include that wait; alone the saving "
              "forward takes 0.11 ms", work_from_timer=True, prof=("gat_fwd_kernel<",)),
        entry("gat_bwd_kernel", "gat_bwd_kernel", "mfma", 2 * f_gat_learn, 1e12, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s",
              f"GAT backward of Prediction_policy.learn ({S_pred} sampled rows x 5 nets; backward = 2x the forward FLOPs: pair-GRU BPTT with "
              "in-kernel dW_hh, attention and node-projection backward), side stream beside behaviour learning",
              prof=("gat_bwd_kernel", "gat_whh_grad_kernel")),
    ]
    n_sc, sc_us, _ = timed.get("gat_scenes_span_us", (0, nan, 0.0))
    if n_sc:
        f_scenes = gat_algorithmic_flops(nA, E, N, d + Z) + nA * V * (Lw * (2 * d * R + 12 * R * R) + 2 * R * Z)
        out[0]["scenes"] = {"what": "the GAT scenes + encoder tiles INSIDE the launch: first scene's entry stamp to the last scene's end stamp "
                                    "(wall_clock64, 100 MHz, thread 0 of every scene workgroup), mean over the sampled launches; the rest of the launch "
                                    "is the next step's action selection, a latency chain behind the last scene (DESIGN.md section 4)",
                            "launches_sampled": n_sc, "span_us": sc_us, "algorithmic_gflop": f_scenes / 1e9,
                            "achieved": f_scenes / (sc_us * 1e-6) / 1e12, "frac": f_scenes / (sc_us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
    cov = roofline_coverage(out, PMC_FILE)
    out[0]["rows_cover_kernel_time"] = cov
    if cov["covered_frac"] is not None and cov["covered_frac"] < 0.90 and not (opt.rollout_only or os.environ.get("IPLAN_BENCH_NO_COVERAGE_CHECK")):
        raise SystemExit(f"bench.py: the roofline rows explain only {cov['covered_frac']:.1%} of the cycle's kernel time ({cov['source']}); "
                         f"largest kernels without a row: {cov['largest_uncovered']}")
    return out


if __name__ == "__main__":
    main()
