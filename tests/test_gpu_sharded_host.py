"""ONE proof over several ranks by the C++ host (sandstorm_amd/host/sharded.cpp: column-sharded LDE, re-shard into row blocks with
the wrap-around halo, row-block constraints / DEEP / hashing, leaf-block sub-trees with the top levels merged on the hosts,
composition and DEEP gathers, FRI on rank 0, sharded openings) with the REAL kernels.  The GPU box has one GPU: the ranks are
threads of this process, each with its own context on that GPU, meeting in the LocalTransport (device-to-device copies between
the contexts); on a multi-GPU node the same driver runs one process per GPU over RCCL (ss_comm_*: grouped ncclSend / ncclRecv),
which this box exercises with a group of one.  The bytes must be the single-device proofs'."""
import os

import pytest

from tests.sharded_host_cases import GOLD, mini_case, recursive_case, run_ranks, single_device_mini

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def spread_small_fri_layers(monkeypatch):
    """the ranks fold FRI layers above 2^21 values together: make these small proofs do it too (sharded.cpp)"""
    monkeypatch.setenv("SSH_FRI_SPREAD_MIN_LOG", "6")


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.parametrize("name,log_n,max_remainder", [("mini_proof_eth_log9.bin", 9, 4), ("mini_proof_eth_log5_nolayers.bin", 5, 32)])
def test_cpp_sharded_prover_writes_the_single_device_proofs(world, name, log_n, max_remainder):
    make, _ = mini_case(log_n, max_remainder)
    with open(os.path.join(GOLD, name), "rb") as f:
        want = f.read()
    assert run_ranks(world, make(world)) == want


@pytest.mark.parametrize("world", [2, 8])
def test_cpp_sharded_prover_friendly_tree_and_cairo_coin(world):
    """FriendlyMerkleTree<7> over 2^10 leaves: the Blake2s / Pedersen boundary inside the trees, above and below the ranks' sub-tree
    roots; `MixedMerkleDigest` tags through the sharded openings"""
    from sandstorm_amd import backend as be
    make, case = mini_case(9, 4, "cairo", 7)
    ctx = be.Context(0)
    want = single_device_mini(ctx, case)
    ctx.close()
    assert run_ranks(world, make(world)) == want


@pytest.mark.parametrize("world", [1, 4])
def test_cpp_sharded_prover_real_recursive_air_cairo_claim(world):
    """the reference's example under the CLI's claim for it: tests/golden/array_sum_recursive_cairo.proof (the single-device C++
    host's), wrap-around halo of 4116 rows on the last rank"""
    make, _ = recursive_case(14)
    with open(os.path.join(GOLD, "array_sum_recursive_cairo.proof"), "rb") as f:
        want = f.read()
    assert run_ranks(world, make(world)) == want


def test_cpp_sharded_prover_at_2p16_steps_two_ranks():
    """BASELINE configs[1]'s size: equal to the single-device proof of the same statement (hostlib.prove)"""
    from sandstorm_amd import backend as be, hostlib
    make, (tree, nf, coin, opt, host, log_n, pi, seed) = recursive_case(16)
    ctx = be.Context(0)
    air = hostlib.RecursiveHostAir(ctx, pi, log_n, 1)
    base = [ctx.column(c) for c in host]
    keep = []

    def ext(challenges):
        keep.append(hostlib.build_extension_columns(ctx, "recursive", [base[3], base[4], base[5], base[1], base[2]], 1 << log_n, challenges))
        return keep[-1].cols
    want = hostlib.prove(ctx, air, tree, nf, coin, seed, base, log_n, ext, opt, wire=True)
    for m in keep:
        m.close()
    air.close()
    del base
    ctx.close()
    assert run_ranks(2, make(2)) == want


def test_rccl_transport_with_a_group_of_one():
    """the RCCL path (ss_comm_create from a unique id, grouped send / receive, all-gather) as far as one GPU can take it: a group of
    one - communicator set-up, the own-rank copies, the host all-gathers through device buffers; the communicator is made once and
    proves twice (an RCCL unique id serves one ncclCommInitRank per rank: ADVICE r3)"""
    make, _ = mini_case(9, 4)
    with open(os.path.join(GOLD, "mini_proof_eth_log9.bin"), "rb") as f:
        want = f.read()
    assert run_ranks(1, make(1), group="rccl", repeat=2) == want
