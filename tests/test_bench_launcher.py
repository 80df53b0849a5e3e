"""`python bench.py --gpus N` from the bare command: the parent starts N ranks itself (the reference's
benchmarks are started by one `mpirun -n N` line, benchmarks/run_benchmarks.py:27-31,181), relays rank 0's line and exits with
the ranks' status -- non-zero when one of them fails.  `--launch-check` replaces the model by one all-reduce over gloo, so the
launcher path runs end to end without a GPU; the full two-rank run on one GPU is the `gpu` test below."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_bare_command_starts_its_own_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0's line only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["sum_of_rank_ids_plus_one"] == 3 and rec["master"].startswith("127.0.0.1:")


def test_a_failing_rank_fails_the_job():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_env(RH_BENCH_TEST_FAIL_RANK="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "rank 1 exited with status 3" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]   # no line from a job that lost a rank


def test_launched_by_torch_distributed_run_the_script_does_not_launch_again():
    # the driver's N > 1 form: the ranks find WORLD_SIZE set and are ranks
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", BENCH, "--gpus", "2", "--launch-check"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_from_the_bare_command():
    """The whole bench on two ranks sharing device 0 (RH_BENCH_SINGLE_DEVICE=1: RCCL refuses two ranks on one device, the summary word
    goes over gloo) started by the bare command."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "6", "--warmup", "2", "--size", "256", "250", "--no-cpu-baseline"],
                       env=_env(RH_BENCH_SINGLE_DEVICE="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["steps"] == 6 and rec["value"] > 0
    assert rec["config"]["cells_per_gpu"] == 64000


@pytest.mark.gpu
def test_one_rank_communicator_is_counted():
    """RH_BENCH_FORCE_DIST=1: the multi-GPU stepping (rh_run_steps_dist) on a one-rank RCCL communicator; the line records ncclCommCount."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "6", "--warmup", "2", "--size", "256", "250", "--no-cpu-baseline"],
                       env=_env(RH_BENCH_FORCE_DIST="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["config"]["n_ranks_in_comm"] == 1 and "rh_run_steps_dist" in rec["config"]["stepping"]
