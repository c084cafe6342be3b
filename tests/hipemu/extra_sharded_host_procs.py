"""The C++ host's sharded prover (sandstorm_amd/host/sharded.cpp) on 2, 4 and 8 ranks that are PROCESSES - the path `bench.py --gpus N`
takes by default - over the device code on the CPU: every process under torch.distributed.run loads the emulated library, runs its
own coin in lock step, deals its columns and meets the others in the driver's CallbackTransport over gloo
(tests/dist_cpp_host_worker.py).  The bytes must be the single-device proofs.  Spread base columns included: the mini AIR's two
columns on 4 and 8 ranks and the recursive layout's seventh on 2 are each ONE transform over the ranks (sharded.cpp `nb_owned`).
Run by tests/test_device_code_on_host.py."""
import os
import subprocess
import sys

import pytest

from tests.sharded_host_cases import GOLD, mini_case, single_device_mini


ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

pytestmark = pytest.mark.gpu


def run_processes(world, case, tmp_path, repeat=1, timeout=1500, **env_extra):
    """-> rank 0's proof bytes of `case` proved by `world` processes (tests/dist_cpp_host_worker.py under torch.distributed.run)"""
    out_path = os.path.join(str(tmp_path), "proof_%d.bin" % world)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", **env_extra)
    env.pop("HIPEMU_THREADS", None)                          # the worker sizes the emulator's pool for its number of ranks
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_cpp_host_worker.py"), case, out_path, str(repeat)]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0 and "SHARDED_PROOF_WRITTEN" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    with open(out_path, "rb") as f:
        return f.read()


@pytest.fixture(autouse=True)
def spread_small_fri_layers(monkeypatch):
    """the ranks fold FRI layers above 2^21 values together: make these small proofs do it too (sharded.cpp)"""
    monkeypatch.setenv("SSH_FRI_SPREAD_MIN_LOG", "6")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_processes_write_the_single_device_proof(world, tmp_path):
    with open(os.path.join(GOLD, "mini_proof_eth_log9.bin"), "rb") as f:
        want = f.read()
    assert run_processes(world, "mini:9:4", tmp_path, repeat=2 if world == 2 else 1) == want      # (a group outlives a proof)


@pytest.mark.parametrize("world", [2, 8])
def test_processes_friendly_tree_and_cairo_coin(world, tmp_path):
    """FriendlyMerkleTree<7> over 2^10 leaves: the Blake2s / Pedersen boundary above and below the ranks' sub-tree roots, the host's
    Pedersen merges of the top levels and the Cairo coin's Pedersen chains in every process"""
    from sandstorm_amd import backend as be
    _, case = mini_case(9, 4, "cairo", 7)
    ctx = be.Context(0)
    want = single_device_mini(ctx, case)
    ctx.close()
    assert run_processes(world, "mini-cairo:9:4:7", tmp_path) == want


@pytest.mark.parametrize("world,case", [(2, "recursive:14"), (4, "recursive:14"), (4, "recursive:14:blocks")])
def test_processes_real_recursive_air_cairo_claim(world, case, tmp_path):
    """the reference's example under the CLI's claim for it: tests/golden/array_sum_recursive_cairo.proof (written by the single-device
    C++ host on the MI355X).  2 ranks: base column 6 is left over and spread; 4 ranks: every base column on its owner.  blocks: the
    extension trace's scans divided over the processes (one all-gather of the blocks' totals over gloo)"""
    with open(os.path.join(GOLD, "array_sum_recursive_cairo.proof"), "rb") as f:
        want = f.read()
    assert run_processes(world, case, tmp_path) == want


@pytest.mark.parametrize("world", [2, 8])
def test_group_self_check_passes_and_catches_a_flipped_byte(world, tmp_path):
    """hostlib.group_self_check (host/sharded.cpp transport_self_check) - what `bench.py --gpus N` runs over its group before the warm-up:
    every ordered pair of ranks exchanges two messages of different sizes whose bytes name (source, destination, message); one flipped
    byte in what rank 1 receives is refused by EVERY rank (the verdicts are gathered: nobody is left in a collective), naming rank 1"""
    assert run_processes(world, "selfcheck", tmp_path).decode().split("\n") == ["PASSED"] * world
    got = run_processes(world, "selfcheck:corrupt", tmp_path).decode().split("\n")
    assert [g.split(":")[0] for g in got] == ["REFUSED rank %d" % r for r in range(world)], got
    assert all("rank 1 got a wrong byte in message" in g for g in got), got
