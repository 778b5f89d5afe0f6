"""Per-tensor PPO gradient error table (VERDICT r2 "next" #2): for every (net, tensor) of the chosen cases

    kernel        |g_kernel - g_fp64| / max|g_fp64|, both at the parameters the learner held before its LAST optimiser step
    fp32_oracle   the same for the oracle in fp32 (= the reference's arithmetic) at that same parameter point
    trajectory    g_kernel against the fp64 oracle's OWN multi-step trajectory (what round 2 logged as "grad")
    post          post-train parameters against the fp64 trajectory

Runs on the GPU library (default; IPLAN_HIP_LIB selects an A/B build, e.g. one compiled with -DIPLAN_EXACT_GATES) or, with
--emu, on the host-emulated build of the same kernel sources.  Test infrastructure: imports the oracle as the checker.

    python scripts/ppo_grad_error_table.py [--emu] [--cases small2,mb3x2,switches,cfg3_1,cfg3_2] [--json out.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def cases(dev):
    from iplan_amd.config import default_args
    from tests.test_emu_learners import _small
    cuda = dev == "cuda"
    A = lambda **kw: default_args("highway", use_cuda=cuda, **kw)  # noqa: E731
    return {
        "small2": (lambda: _small(ppo_epoch=2), dict(seed=6)),
        "mb3x2": (lambda: A(max_vehicle_num=9, n_agents=2, episode_limit=12, batch_size_run=4, buffer_size=6, batch_size=5, ppo_epoch=2,
                            num_mini_batch=3), dict(seed=41)),
        "switches": (lambda: _small(ppo_epoch=2, use_huber_loss=False, use_clipped_value_loss=False, use_value_active_masks=False,
                                    use_policy_active_masks=False, use_gae=False), dict(seed=41)),
        "cfg3_1": (lambda: A(ppo_epoch=1), dict(seed=24, agents=(0,))),
        "cfg3_2": (lambda: A(ppo_epoch=2), dict(seed=24, agents=(0,))),
        "cfg3_3": (lambda: A(ppo_epoch=3), dict(seed=24, agents=(0,))),
        # which dimension makes a case ill-conditioned: entities (feature width) or rows?
        "n55_rows60": (lambda: A(episode_limit=12, batch_size_run=4, buffer_size=6, batch_size=5, ppo_epoch=2), dict(seed=24, agents=(0,))),
        "n55_rows2250": (lambda: A(buffer_size=26, batch_size=25, ppo_epoch=2), dict(seed=24, agents=(0,))),
        "n9_rows22950": (lambda: A(max_vehicle_num=9, n_agents=2, ppo_epoch=2), dict(seed=24, agents=(0,))),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("--cases", default="small2,mb3x2,switches")
    ap.add_argument("--json", default=None)
    ap.add_argument("--top", type=int, default=12)
    a = ap.parse_args()
    dev = "cpu" if a.emu else "cuda"
    if a.emu:
        from iplan_amd import _lib as L
        from tests.emu.emu_lib import get_emu_lib
        L.use_library_for_tests(get_emu_lib())
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    from tests.oracle_checks import check_ppo_train_vs_oracle
    out = {"device": dev, "lib": os.environ.get("IPLAN_HIP_LIB", "default"), "cases": {}}
    cs = cases(dev)
    for name in a.cases.split(","):
        mk, kw = cs[name]
        table = []
        try:
            worst = check_ppo_train_vs_oracle(mk(), dev, table=table, assert_grads=False, **kw)
            failed = None
        except AssertionError as e:                      # keep the rows collected so far: the table is the point
            worst, failed = None, str(e)[:300]
        table.sort(key=lambda r: -r["kernel"])
        print(f"== {name}  ({len(table)} tensors)  worst={worst}  failed={failed}")
        print(f"{'agent':>5} {'net':>6} {'tensor':<34} {'max|g|':>10} {'kernel':>10} {'fp32 orac':>10} {'ratio':>6} {'trajectory':>10} {'post':>9}")
        for r in table[:a.top]:
            f32 = r["fp32_oracle"] or 0.0
            print(f"{r['agent']:>5} {r['net']:>6} {r['tensor']:<34} {r['gmax']:>10.3e} {r['kernel']:>10.2e} {f32:>10.2e} "
                  f"{(r['kernel'] / f32 if f32 else float('nan')):>6.2f} {r['kernel_vs_trajectory']:>10.2e} {r['post']:>9.2e}")
        out["cases"][name] = dict(worst=worst, failed=failed, rows=table)
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
