"""The C++ host's sharded prover (sandstorm_amd/host/sharded.cpp) over the device code on the CPU: 1, 2, 4 and 8 ranks as threads
of one process, each with its own (emulated) context, meeting in the LocalTransport.  The bytes must be the single-device
proofs: the MI355X-made fixtures and, for the friendly-tree flavour, the single-device C++ prover run here.
Run by tests/test_device_code_on_host.py with HIPEMU_THREADS=1 (the emulator's worker pool serves one launching thread; the
ranks are the parallelism here)."""
import os

import pytest

from tests.sharded_host_cases import GOLD, mini_case, recursive_case, run_ranks, single_device_mini

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def spread_small_fri_layers(monkeypatch):
    """the ranks fold FRI layers above 2^21 values together: make these small proofs do it too (sharded.cpp)"""
    monkeypatch.setenv("SSH_FRI_SPREAD_MIN_LOG", "6")


@pytest.mark.parametrize("world,name,log_n,max_remainder", [(w, "mini_proof_eth_log9.bin", 9, 4) for w in (1, 2, 4, 8)]
                         + [(w, "mini_proof_eth_log5_nolayers.bin", 5, 32) for w in (1, 2, 4)])     # a transform over R ranks has >= R^2 points
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


@pytest.mark.parametrize("world,blocks", [(4, False), (8, False), (2, True), (8, True)])
def test_cpp_sharded_prover_real_recursive_air_cairo_claim(world, blocks):
    """the reference's example under the CLI's claim for it, 4 and 8 ranks: tests/golden/array_sum_recursive_cairo.proof (written by
    the single-device C++ host on the MI355X), wrap-around halo of 4116 rows included; three extension columns, the composition
    and DEEP's extension each ONE transform over the ranks, two FRI layers folded by the ranks.  blocks: the extension trace's scans
    divided over the ranks too (build_extension_blocks: every rank its rows, one all-gather of the blocks' totals) - the same bytes"""
    make, _ = recursive_case(14)
    with open(os.path.join(GOLD, "array_sum_recursive_cairo.proof"), "rb") as f:
        want = f.read()
    assert run_ranks(world, make(world, blocks=blocks)) == want


def test_too_few_rows_for_the_ranks_is_an_error():
    """a transform of fewer than R^2 points does not spread over R ranks: a message, not a wrong root"""
    make, _ = mini_case(5, 32)
    with pytest.raises(Exception, match="R\\^2|ranks"):
        run_ranks(8, make(8))
