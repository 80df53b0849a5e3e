"""GPU: the library's code for MORE THAN ONE RANK, on the device, with two ranks ("multi-rank device code has never run
with more than one rank").  RCCL does not place two ranks on one device and a build box has one GPU, so the two ranks are two threads of a
child process, each with its own context, and the NCCL entry points the library resolves at run time come from tests/loopback_nccl.cpp
(RH_RCCL_LIB): ncclAllReduce, grouped ncclSend / ncclRecv with per-pair FIFO matching, as synchronous exchanges between the threads.  What runs
is the product's rh_comm_init, rh_run_steps_dist, route_exchange (its neighbour ranks, buffer offsets, counts and order) and the
predicate-word expansion / compression kernels with nranks = 2; what is checked is that two halves of a golden domain equal the single
domain bit for bit and the reference's golden run.  The real RCCL with one rank: tests/test_hip_comm.py; the protocol over gloo on the CPU:
tests/test_distributed_gloo.py."""
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def loopback(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc is needed to build the loopback communicator")
    so = tmp_path_factory.mktemp("loopback") / "libloopback_nccl.so"
    subprocess.run([HIPCC, "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(HERE, "loopback_nccl.cpp"), "-o", str(so)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return str(so)


def _child(loopback, scenario):
    env = dict(os.environ, RH_RCCL_LIB=loopback)
    r = subprocess.run([sys.executable, os.path.join(HERE, "loopback_ranks_child.py"), scenario], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"{scenario}:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    print(r.stdout.strip())
    return r.stdout


def test_ranks_all_reduce_the_predicate_words(loopback):
    out = _child(loopback, "allreduce")
    assert "2 ranks == single domain == golden" in out and "4 ranks == single domain == golden" in out


def test_ranks_exchange_the_edge_columns_of_the_routing(loopback):
    out = _child(loopback, "routing")
    assert out.count("2 ranks == single domain") >= 2 and "4 ranks == single domain == golden" in out and "3 ranks == single domain" in out


def test_two_ranks_routing_entry_points(loopback):
    assert "2 ranks == single domain" in _child(loopback, "routing_by_routine")
