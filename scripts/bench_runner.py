"""Host-in-the-loop rate of the device-resident ParallelRunner (iplan_amd/runners/ippo_parallel_runner.py) at
BASELINE config 3 with a stub vector env in place of the simulator: per step ONE device->host copy of the [E, nA] actions,
env.step + the vectorised id->slot history wrapper on the host, one batch of pinned host->device copies.  Reported BESIDE
bench.py's HBM-resident `value` (never as it): it contains host work the hot path does not own.

    python scripts/bench_runner.py [--episodes 4]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from iplan_amd import synth  # noqa: E402
from iplan_amd.config import default_args  # noqa: E402
from iplan_amd.controllers.dcntrl_controller import DcntrlMAC  # noqa: E402
from iplan_amd.nova.prediction_policy import Prediction_policy  # noqa: E402
from iplan_amd.nova.stable_behavior_policy import Behavior_policy  # noqa: E402
from iplan_amd.runners.ippo_parallel_runner import ParallelRunner  # noqa: E402


class Log:
    def log_stat(self, *a, **k):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=4)
    ap.add_argument("--envs", type=int, default=32)
    opt = ap.parse_args()
    E = opt.envs
    torch.set_num_threads(min(8, os.cpu_count() or 1))     # host work is small-array numpy / torch ops: no 256-thread fan-out
    args = default_args("highway", use_cuda=True, batch_size_run=E, n_obs_vehicles=15, device="cuda", animation_enable=False)
    args.obs_shape = args.obs_shape_single * args.n_obs_vehicles
    args.state_shape = args.obs_shape_single * args.max_vehicle_num
    scheme = synth.make_scheme(args)
    mac = DcntrlMAC(scheme, {"agents": args.n_agents}, args)
    beh, pred = Behavior_policy(args, Log()), Prediction_policy(args, Log())
    scheme_in = dict(scheme)
    scheme_in.pop("actions_onehot")
    scheme_in.pop("filled")

    class OneHot:
        def infer_output_info(self, vshape_in, dtype_in):
            return (args.n_actions,), torch.float32
    env = synth.StubHighwayVecEnv(E, args.n_agents, args.n_obs_vehicles, args.obs_shape_single, args.max_vehicle_num,
                                  args.episode_limit, seed=0, n_ids=args.max_vehicle_num - args.n_agents - 1)
    runner = ParallelRunner(args, env, Log())
    runner.setup(scheme_in, {"agents": args.n_agents}, {"actions": ("actions_onehot", [OneHot()])}, mac, beh, pred)
    runner.run()                                                       # warm-up
    torch.cuda.synchronize()
    runner.host_seconds = 0.0
    t0 = time.perf_counter()
    for _ in range(opt.episodes):
        runner.run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = opt.episodes * E * args.episode_limit
    print(f"device-resident ParallelRunner, stub env, E={E}: {dt / opt.episodes * 1e3:.1f} ms per {args.episode_limit}-step episode = "
          f"{dt / opt.episodes / args.episode_limit * 1e3:.3f} ms per vector step -> {steps / dt:.0f} env-steps/s rollout-only with the host in the "
          f"loop; host share (stub env.step + history wrapper) {100 * runner.host_seconds / dt:.0f} % of the wall time")


if __name__ == "__main__":
    main()
