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


@pytest.mark.parametrize("world", [1, 2, 4, 8])
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


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_friendly_tree_proof_is_the_single_device_proof(world, tmp_path, oracle):
    """CairoVerifierClaim's parts - FriendlyMerkleTree + the Cairo coin - over the ranks: sub-trees with N - log2 R
    Pedersen layers, the top levels merged on the hosts, the `MixedMerkleDigest` tags of every path entry carried through
    the sharded openings into the wire format.  N = 7 puts the Blake2s / Pedersen boundary INSIDE the 2^10-leaf trees (and,
    on 8 ranks, 4 levels above the sub-tree roots).  The bytes must be the single-device prover's, and verify."""
    from oracle.cpu_context import CpuContext
    from sandstorm_amd import backend as be, verifier, wire
    from sandstorm_amd.coin import canonical
    from sandstorm_amd.prover import Prover
    from tests import dist_prove_worker as w, mini_air
    from tests.test_verifier import mini_verifier_air
    n_friendly = 7
    n, cols, claim, opt, _, seed, leaf_hash = w.mini(9, 4, "cairo", n_friendly)
    ctx = CpuContext()
    c0 = [int(v) for v in oracle.from_mont(cols[0])]
    base = be.Matrix.from_host(ctx, [cols[0], cols[1]])
    ref = Prover(ctx, claim, opt).prove(seed, base, lambda ch: be.Matrix.from_host(
        ctx, [oracle.to_mont(mini_air.extension_trace(c0, canonical(ch[0])))]))
    want = wire.serialize(wire.from_proof(ref, leaf_hash))
    parsed = wire.parse(want, be.TREE_FRIENDLY)
    assert {t for o in parsed.base_openings for t in o.tags} == {0, 1}          # both digest variants on the wire
    got = run_sharded(world, "mini:9:4:cairo:%d" % n_friendly, tmp_path, timeout=900)
    assert got == want
    verifier.verify(got, mini_verifier_air(), be.TREE_FRIENDLY, be.COIN_CAIRO, seed, required_security_bits=20, n_friendly_layers=n_friendly)
