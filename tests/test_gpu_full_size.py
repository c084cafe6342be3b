"""BASELINE.json's full-size configurations, checked ("configs not exercised by a `-m gpu` test").

configs[2] - starknet layout, 2^20 steps, one GPU - and configs[3]'s shape - 2^22 steps - as REAL statements: the
reference's array-sum run re-declared for the starknet layout and padded with its final state (the program ends in
`jmp rel 0`, so any power of two of steps is a valid run; `tests/test_layout_starknet.py::starknet_example`).  The C++
host builds the 2^24- (2^26-) row base trace, every stage of the proof is a HIP kernel behind the C ABI at the size
`bench.py` times, and the proof must satisfy both verifiers - the C++ one and the Python one, whose AIR evaluation at
the out-of-domain point is the independent restatement (layouts/starknet.py) pinned by the reference's own proofs.
The oracle cannot run these sizes in seconds; what holds the kernels here is the size-independent property that a
proof of a true statement verifies (one wrong LDE value, leaf, quotient point, DEEP term or fold breaks it), plus the
byte-identity of the sharded driver's proof with the single-device one at 2^20 steps."""
import os
import time

import pytest

from tests.test_layout_starknet import starknet_example

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _statement(log_steps):
    from sandstorm_amd import backend as be, binary, hostlib, public_input
    from sandstorm_amd.layouts import starknet as sk
    t0 = time.time()
    states, memory, spi = starknet_example(log_steps)
    cols = hostlib.starknet_base_trace(binary.write_register_states(states), binary.write_memory(memory), spi)
    del states, memory
    log_n = log_steps + 4
    assert cols[0].shape[0] == 1 << log_n
    t1 = time.time()
    ctx = be.Context(0)
    base = be.Matrix.from_host(ctx, cols)
    del cols
    air = hostlib.StarknetHostAir(ctx, spi, log_n)
    seed = public_input.public_coin_seed(spi, be.COIN_SOLIDITY)
    print("2^%d steps: base trace on the host %.1f s, to the device %.1f s" % (log_steps, t1 - t0, time.time() - t1))
    return ctx, base, air, seed, spi, sk, log_n


def _prove_and_verify(log_steps, python_verifier):
    from sandstorm_amd import backend as be, hostlib, verifier, wire
    from sandstorm_amd.prover import ProofOptions
    ctx, base, air, seed, spi, sk, log_n = _statement(log_steps)
    keep = []

    def build_extension(challenges):               # check=True: the memory / range-check / diluted products must close
        m = hostlib.build_extension_columns(ctx, "starknet", [base.cols[sk.COL_NPC], base.cols[sk.COL_MEMORY], base.cols[sk.COL_RANGE_CHECK]],
                                            1 << log_n, challenges)
        keep.append(m)
        return m.cols
    try:
        opt = ProofOptions()                       # cli/src/main.rs:51-60 defaults: 65 queries, blowup 2, 16 grinding bits
        t0 = time.time()
        raw = hostlib.prove(ctx, air, be.TREE_KECCAK_M20, 0, be.COIN_SOLIDITY, seed, base.cols, log_n, build_extension, opt, wire=True)
        t_prove = time.time() - t0
        parsed = wire.parse(raw)
        assert parsed.trace_len == 1 << log_n and len(parsed.ood_trace) == 269 and len(parsed.ood_composition) == 2
        assert len(parsed.fri_layers) == verifier.fri_layer_count(1 << log_n, 8, 16)[0]
        positions = hostlib.verify(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw, expected_options=opt)
        assert len(positions) == len(parsed.base_openings) and len(set(positions)) == len(positions)
        if python_verifier:
            assert verifier.verify(raw, sk.verifier_air(spi), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, expected_options=opt) == positions
        # and a flipped bit anywhere is caught
        bad = bytearray(raw)
        bad[len(bad) // 2] ^= 1
        with pytest.raises(Exception):
            hostlib.verify(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, bytes(bad), expected_options=opt)
        print("2^%d steps: proof of %d bytes in %.2f s (first call: plans and tables included), %d queries verified"
              % (log_steps, len(raw), t_prove, len(positions)))
        return raw
    finally:
        for m in keep:
            m.close()
        air.close()
        del base
        ctx.close()


@pytest.fixture(scope="module")
def proof_2p20():
    return _prove_and_verify(20, python_verifier=True)


def test_starknet_2p20_steps_real_statement_proves_and_verifies(proof_2p20):
    """BASELINE configs[2] at full size: 2^24 rows x 10 columns, the layout's real 195-constraint AIR"""
    assert len(proof_2p20) > 100_000


def test_cpp_sharded_host_two_ranks_at_2p20_steps_real_starknet_air(proof_2p20):
    """the C++ host's sharded prover (host/sharded.cpp) at BASELINE configs[2]'s size on two ranks - threads of this process, each
    with its own context on this box's GPU: column-owned base LDE and re-shard with the 66 316-row halo, the extension column, the
    composition (one 2^25-point inverse, two extensions) and DEEP's extension each ONE transform over the ranks, FRI layers 0 and
    1 (2^25 and 2^22 values) folded and committed by both ranks - the single-device proof, byte for byte.  The extension column comes as
    row blocks: each rank scans its 2^23 rows, one all-gather of the blocks' totals (hostlib.build_extension_blocks, ABI 12)"""
    from tests.sharded_host_cases import run_ranks, starknet_case
    make, _ = starknet_case(20)
    assert run_ranks(2, make(2, blocks=True)) == proof_2p20


def test_cpp_sharded_host_two_PROCESSES_at_2p20_steps_real_starknet_air(proof_2p20, tmp_path):
    """the same as processes - what `bench.py --gpus N` starts: two ranks under torch.distributed.run sharing this box's GPU, each with
    its own context, coin and columns, the group self check first, the exchanges through the driver's CallbackTransport over gloo
    (tests/dist_cpp_host_worker.py): BASELINE configs[2]'s statement, the layout of configs[3], the single-device proof byte for byte"""
    from tests.hipemu.extra_sharded_host_procs import run_processes
    assert run_processes(2, "starknet:20", tmp_path, timeout=1500) == proof_2p20


@pytest.fixture(scope="module")
def proof_2p22():
    import torch
    if torch.cuda.mem_get_info(0)[1] < 200 << 30:
        pytest.skip("needs an MI355X-sized device")
    return _prove_and_verify(22, python_verifier=False)


def test_starknet_2p22_steps_real_statement_proves_and_verifies(proof_2p22):
    """BASELINE configs[3]'s statement size (2^26 rows x 10 columns: 21 GB of base trace, 43 GB of LDE) on ONE device -
    the capacity check behind DESIGN's "sized for 288 GB"; on the driver's 8-GPU node the same statement size is what
    `bench.py --gpus 8 --workload starknet_2p22` shards"""
    assert len(proof_2p22) > 100_000


def test_cpp_sharded_host_two_ranks_at_2p22_steps(proof_2p22):
    """BASELINE configs[3] in its stated form as far as one GPU goes: the 2^22-step statement SHARDED - two ranks (threads, own
    contexts, this box's GPU, which holds both ranks' 2^27-row working sets; on the driver's node: one process per GPU over RCCL,
    the same driver) - writes the single-device proof of the same statement, byte for byte"""
    from tests.sharded_host_cases import run_ranks, starknet_case
    make, _ = starknet_case(22)
    assert run_ranks(2, make(2)) == proof_2p22
