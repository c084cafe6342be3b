"""`python bench.py --gpus N` as the driver types it - no launcher around it.  bench.py starts its own N ranks
under torch.distributed.run, rank 0 prints the ONE JSON line, the exit code is the ranks'.  Here on gloo ranks with the
device code on the CPU behind the C++ sharded host (SS_BENCH_SELFTEST=1: tests/bench_selftest.py), since this container has no GPU; and the
refusals: fewer devices than ranks, a group whose size is not --gpus."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(argv, env_extra, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.fixture(scope="module")
def emulated_library():
    """the default `--sharded-host cpp` self-test runs the C++ host over the device code on the CPU (tests/hipemu: ~15 s when built)"""
    out = subprocess.run(["bash", os.path.join(ROOT, "tests", "hipemu", "build.sh")], capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    from tests import mini_air_host
    mini_air_host.load()                                 # built once here, not by N ranks at the same time
    return out.stdout.strip().splitlines()[-1]


@pytest.mark.parametrize("gpus", [2, 4, 8])
def test_bench_launches_its_own_ranks(gpus, emulated_library):
    """the path of `bench.py --gpus N` - the C++ sharded host, one PROCESS per rank, group self check before the warm-up"""
    host = "cpp"
    out = run_bench(["--gpus", str(gpus), "--steps", "2", "--warmup", "1"],
                    {"SS_BENCH_SELFTEST": "1", "OMP_NUM_THREADS": "1", "SS_TEST_HIPEMU_LIB": emulated_library})
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout                   # the contract: ONE line on stdout, whatever the ranks and libraries print
    line = json.loads(lines[0])
    assert line["n_gpus"] == gpus and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "strong"
    assert line["selftest"] is True and line["proof_is_the_single_device_proof"] is True
    assert line["config"]["sharded_host"] == host
    if host == "cpp":
        assert line["config"]["comm_self_check"]["ok"] is True and line["config"]["comm_self_check"]["exchange_gbps"] > 0
    assert line["value"] > 0 and abs(line["ms_per_step"] - 1e3 * line["value"]) < 1e-6


def test_the_retired_python_driver_is_refused():
    """one distribution in the tree: `--sharded-host python` names nothing any more"""
    out = run_bench(["--gpus", "2", "--sharded-host", "python"], {"SS_BENCH_SELFTEST": "1"})
    assert out.returncode != 0 and "retired" in out.stderr and out.stdout.strip() == ""


def test_bench_refuses_more_ranks_than_devices():
    """no GPU here: `--gpus 2` must exit non-zero with a message, not hang in a rendezvous or die on an assert"""
    out = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert out.returncode == 2 and "this node has 0 GPU(s)" in out.stderr and out.stdout.strip() == ""


def test_bench_refuses_a_group_of_another_size():
    out = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "inside a group of 1 ranks" in out.stderr
