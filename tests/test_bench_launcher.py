"""bench.py's rank launcher (VERDICT r5 "missing" #1): ``python bench.py --gpus N`` without a launcher around it starts the N
ranks itself under torch.distributed.run, refuses to run fewer ranks than asked for, and refuses a --gpus that disagrees with
the launcher's WORLD_SIZE.  No GPU needed: these paths end before the first CUDA call."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def test_print_launch_is_the_torchrun_line():
    r = _run(["--gpus", "8", "--steps", "20", "--warmup", "5", "--print-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    argv = json.loads(r.stdout.strip().splitlines()[-1])["launch_argv"]
    assert argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(argv[argv.index("--master-port") + 1]) < 65536
    i = argv.index(BENCH)
    assert argv[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]        # every other argument is handed on


def test_more_ranks_than_gpus_is_an_error_not_a_smaller_run():
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run(["--gpus", str(have + 1)])
    assert r.returncode != 0
    assert f"sees {have} GPU(s)" in r.stderr and f"--gpus {have + 1}" in r.stderr
    assert "{" not in r.stdout, "no JSON line may be printed"


def test_gpus_must_agree_with_the_launchers_world_size():
    r = _run(["--gpus", "2"], WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE = 4" in r.stderr and "{" not in r.stdout


def test_in_process_form_is_one_rank_only():
    r = _run(["--gpus", "2", "--in-process"])
    assert r.returncode != 0 and "ONE rank" in r.stderr


@pytest.mark.gpu
def test_gpus_1_goes_through_the_spawn_path_and_reports_rccl():
    """the driver's single-GPU command: the line must come from a rank started by bench.py's own launcher, on a 1-rank RCCL group"""
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    la = line["launcher"]
    assert line["n_gpus"] == 1 and la["launched"] and la["spawned_by_bench"] and la["rccl_ranks"] == 1 and la["all_reduce_of_ones"] == 1.0
    assert la["rccl_version"].count(".") == 2
    assert len([ln for ln in r.stdout.splitlines() if ln.strip()]) == 1, "stdout carries exactly ONE line (RCCL's banner goes to stderr)"


@pytest.mark.gpu
def test_a_failing_launcher_does_not_cost_the_single_gpu_line():
    """if the spawned rank dies (here: on purpose) the N = 1 line is measured in the launching process and says so; still ONE line"""
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], IPLAN_BENCH_TEST_FAIL_IN_RANK="1")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(out) == 1
    la = json.loads(out[0])["launcher"]
    assert la["spawn_failed"] and not la["launched"] and la["rccl_ranks"] is None
    assert "measuring in this process instead" in r.stderr
