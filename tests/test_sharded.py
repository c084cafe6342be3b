"""One proof over several ranks: sandstorm_amd/sharded_prover.py on 1, 2 and 4 gloo ranks, every rank
running the real driver - column-sharded LDE, point-to-point re-shard into row blocks with the wrap-around halo,
row-block constraint evaluation, digest exchange into leaf-block sub-trees, root all-gather, composition and DEEP
gathers to rank 0, sharded openings - with the CPU oracle standing in for the HIP kernels only.  The proof must be the
single-device proof, byte for byte: compared with the files under tests/golden/ that the MI355X wrote (C++ / Python host
over the HIP kernels, one device)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_sharded(world, case, tmp_path, timeout=600, threads=1):
    out_file = str(tmp_path / "proof.bin")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_prove_worker.py"), case, out_file]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0 and "SHARDED_PROOF_WRITTEN" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    with open(out_file, "rb") as f:
        return f.read()


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.parametrize("name,case", [("mini_proof_eth_log5.bin", "mini:5:4"), ("mini_proof_eth_log9.bin", "mini:9:4"),
                                       ("mini_proof_eth_log5_nolayers.bin", "mini:5:32")])
def test_sharded_proof_is_the_single_device_proof(world, name, case, tmp_path):
    with open(os.path.join(GOLD, name), "rb") as f:
        want = f.read()
    assert run_sharded(world, case, tmp_path) == want


def test_sharded_proof_of_the_reference_example(tmp_path):
    """the reference's example run with the real recursive AIR (133 mask cells, row offsets up to 2058: a 4116-row halo
    that wraps around the domain on the last rank), 2 ranks: the 80 KB the MI355X wrote, byte for byte"""
    with open(os.path.join(GOLD, "array_sum_recursive_eth.proof"), "rb") as f:
        want = f.read()
    assert run_sharded(2, "example", tmp_path, timeout=900, threads=4) == want
