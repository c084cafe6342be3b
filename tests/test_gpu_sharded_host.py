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


@pytest.mark.parametrize("world,blocks", [(1, False), (4, False), (1, True), (2, True), (8, True)])
def test_cpp_sharded_prover_real_recursive_air_cairo_claim(world, blocks):
    """the reference's example under the CLI's claim for it: tests/golden/array_sum_recursive_cairo.proof (the single-device C++
    host's), wrap-around halo of 4116 rows on the last rank.  blocks: the extension trace's scans divided over the ranks (ABI 12:
    hostlib.build_extension_blocks, one all-gather of the blocks' totals) - the same bytes"""
    make, _ = recursive_case(14)
    with open(os.path.join(GOLD, "array_sum_recursive_cairo.proof"), "rb") as f:
        want = f.read()
    assert run_ranks(world, make(world, blocks=blocks)) == want


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


@pytest.mark.parametrize("world,case,gold", [(2, "mini:9:4", "mini_proof_eth_log9.bin"), (4, "mini:9:4", "mini_proof_eth_log9.bin"),
                                             (2, "recursive:14", "array_sum_recursive_cairo.proof"),
                                             (4, "recursive:14:blocks", "array_sum_recursive_cairo.proof")])
def test_cpp_sharded_prover_with_ranks_as_processes(world, case, gold, tmp_path):
    """the ranks as PROCESSES under torch.distributed.run - what `bench.py --gpus N` starts - sharing this box's GPU: every process with
    its own context, coin and columns, the group self check first, the exchanges through the driver's CallbackTransport over gloo
    (staged through the host; RCCL refuses two ranks on one device).  The single-device proofs, byte for byte; with 2 ranks the
    recursive layout's seventh base column is left over and spread.  (The CPU suite runs the same on the emulated device code on
    2 / 4 / 8 processes: tests/hipemu/extra_sharded_host_procs.py.)"""
    from tests.hipemu.extra_sharded_host_procs import run_processes
    with open(os.path.join(GOLD, gold), "rb") as f:
        want = f.read()
    assert run_processes(world, case, tmp_path, timeout=900) == want


def test_group_self_check_over_rccl_and_over_threads():
    """hostlib.group_self_check (what bench.py runs over its RCCL group before the warm-up) on the transports one GPU offers: the RCCL
    group of one rank (own-rank copies, all-gathers through device buffers) and four thread-ranks in a LocalGroup"""
    import threading
    from sandstorm_amd import backend as be, hostlib
    if os.environ.get("SS_TEST_HIPEMU") != "1":              # (the emulated device has no RCCL)
        ctx = be.Context(0)
        grp = hostlib.RcclGroup(ctx, hostlib.rccl_unique_id(), 0, 1)
        assert hostlib.group_self_check(ctx, 0, 1, grp, 1 << 20) == 0.0      # nobody to exchange with: no rate
        grp.close()
        ctx.close()
    world, group, rates, errs = 4, hostlib.LocalGroup(4), [None] * 4, []

    def body(rank):
        c = be.Context(0)
        try:
            rates[rank] = hostlib.group_self_check(c, rank, world, group, 1 << 22)
        except BaseException as e:          # noqa: BLE001
            errs.append(e)
        finally:
            c.close()
    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    group.close()
    assert not errs, errs
    assert all(r > 0 for r in rates), rates
