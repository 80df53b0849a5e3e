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


def test_global_size_is_refused_when_the_ranks_do_not_divide_it():
    """--global-size: the reference's decomposition rule (roger/distributed.py:121-138), applied before any rank is started."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--model", "oned", "--global-size", "3200", "3125", "--launch-check"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "do not divide domain evenly in x-direction" in r.stderr


@pytest.mark.gpu
def test_strong_scaling_command_rehearsed_with_two_ranks_on_one_gpu():
    """BASELINE configs[3] as ONE command: `python bench.py --gpus N --model oned --global-size NX NY` gives each rank NX / N x NY columns
    (num_proc = (N, 1)) and reports strong scaling; rehearsed here at a small global size with two ranks sharing device 0
    (RH_BENCH_SINGLE_DEVICE=1).  On an 8-GPU node: python bench.py --gpus 8 --model oned --global-size 3200 3125."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--model", "oned", "--global-size", "512", "250", "--steps", "6", "--warmup", "2",
                        "--no-cpu-baseline"], env=_env(RH_BENCH_SINGLE_DEVICE="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 2 and rec["steps"] == 6
    assert rec["config"]["global_cells"] == 512 * 250 and rec["config"]["cells_per_gpu"] == 256 * 250
    assert abs(rec["value"] - 512 * 250 * 6 / (rec["ms_per_step"] * 6 / 1e3)) < 1e-6 * rec["value"]


def test_a_terminated_launcher_takes_its_ranks_with_it():
    """SIGTERM to the parent (a driver's timeout): the ranks it started are stopped, none is left behind."""
    import signal
    import time

    import threading

    env = _env(RH_BENCH_TEST_HANG="1")
    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    pids = []

    def reader():   # (a thread: a launcher that never prints must fail the test, not hang it)
        for line in p.stdout:
            if line.startswith("rank-pid "):
                pids.append(int(line.split()[1]))
                if len(pids) == 2:
                    return

    th = threading.Thread(target=reader, daemon=True)
    th.start()
    th.join(timeout=400)   # (the ranks announce themselves after `import torch`, which can take minutes in a fresh container)
    if len(pids) != 2:
        p.kill()
    assert len(pids) == 2
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=60) == 128 + signal.SIGTERM
    time.sleep(0.5)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = True
        except ProcessLookupError:
            alive = False
        assert not alive, f"rank process {pid} survived its launcher"


@pytest.mark.gpu
@pytest.mark.parametrize("stepping,on_device", [("script", True), ("hooks", False)])
def test_bench_as_a_setup_script(stepping, on_device):
    """`bench.py --stepping script`: the benchmark as a RogerSetup script with its OWN set_forcing / set_parameters / after_timestep (the
    reference's bodies, benchmarks/SVAT_benchmark.py:105-110, 152-181) is recognised hook by hook and run() stays on the device;
    `--stepping hooks`: a hook that does something of its own keeps the loop on the host behind one native call per step."""
    r = subprocess.run([sys.executable, BENCH, "--stepping", stepping, "--size", "80", "53", "--days", "4", "--warmup-days", "1"],
                       env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    hooks = rec["config"]["hooks_on_device"]
    assert all(hooks[h] for h in ("set_forcing", "set_parameters", "after_timestep"))
    assert hooks["read_data"] is on_device
    assert rec["steps"] >= 4 and rec["value"] > 0
