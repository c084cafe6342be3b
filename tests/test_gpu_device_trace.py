"""The base trace made ON the device (csrc/trace.hip behind ss_trace_*, driven by host/device_trace.hpp) against the C++ host
generator (host/trace_{recursive,starknet}.cpp), cell for cell - which tests/test_layout_{recursive,starknet}.py hold, cell for cell,
to the Python restatement that the reference's own proof openings pin (layouts/src/{recursive,starknet}/trace.rs).

Runs on the MI355X (`-m gpu`) and, in the CPU suite, on the host build of the device code (tests/test_device_code_on_host.py)."""
import gzip
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
EMULATED = os.environ.get("SS_TEST_HIPEMU") == "1"
EX = os.path.join(ROOT, "tests", "golden", "example")


@pytest.fixture(scope="module")
def ctx():
    from sandstorm_amd import backend as be
    c = be.Context(0)
    yield c
    c.close()


def example_files():
    from sandstorm_amd import public_input
    with open(os.path.join(EX, "trace.bin"), "rb") as f:
        trace_bin = f.read()
    with open(os.path.join(EX, "memory.bin"), "rb") as f:
        memory_bin = f.read()
    pi = public_input.AirPublicInput.from_json(os.path.join(ROOT, "tests", "golden", "air_public_input_array_sum.json"))
    return trace_bin, memory_bin, pi


def device_columns(ctx, layout, trace_bin, memory_bin, pi, priv):
    from sandstorm_amd import hostlib
    n = 16 * (len(trace_bin) // 24)
    cols = hostlib.device_base_trace(ctx, layout, trace_bin, memory_bin, pi, priv)
    out = [c.download(np.uint64, (n, 4)) for c in cols]
    for c in cols:
        c.free()
    return out


def assert_same_columns(got, want):
    assert len(got) == len(want)
    for c, (g, w) in enumerate(zip(got, want)):
        if not np.array_equal(g, w):
            rows = np.nonzero((g != w).any(axis=1))[0]
            raise AssertionError("column %d differs in %d rows, first %s (offsets in a cycle: %s)" % (c, len(rows), rows[:8], sorted(set(int(r) % 16 for r in rows[:64]))))


def recursive_private():
    rng = random.Random(21)
    top = (1 << 251) | (1 << 196) | (1 << 192)
    return {"pedersen": [(0, rng.getrandbits(250), rng.getrandbits(250)), (1, top, (1 << 251) | (1 << 196)), (5, 0, 5)],
            "bitwise": [(i, rng.getrandbits(251), rng.getrandbits(251)) for i in range(9)],
            "range_check": [(i, sum(rng.randrange(32700, 32800) << (16 * k) for k in range(8))) for i in range(6)]}


@pytest.mark.parametrize("real_instances", [False, True])
def test_recursive_columns_of_the_example_run(ctx, real_instances):
    """the reference's example run (2^14 steps, a real program: calls, jumps, conditional jumps with non-zero dst) - every cell of
    the 7 columns; then with real Pedersen / bitwise / range-check instances (templates per distinct instance)"""
    from sandstorm_amd import hostlib
    trace_bin, memory_bin, pi = example_files()
    priv = recursive_private() if real_instances else None
    want = hostlib.recursive_base_trace(trace_bin, memory_bin, pi, priv)
    got = device_columns(ctx, "recursive", trace_bin, memory_bin, pi, priv)
    assert_same_columns(got, want)


def padded_statement(layout, log_steps):
    from sandstorm_amd import binary, examples
    states, memory, pi = (examples.starknet_example if layout == "starknet" else examples.recursive_example)(log_steps)
    return binary.write_register_states(states), binary.write_memory(memory), pi


@pytest.mark.parametrize("layout,log_steps", [("recursive", 15), ("starknet", 17)])
def test_columns_of_the_bench_statements(ctx, layout, log_steps):
    """the statements bench.py's files -> proof leg proves (the example run re-declared and padded with its final state: most cycles
    idle in `jmp rel 0`, every builtin instance is the dummy one) at the smallest size the layout's pools fit"""
    from sandstorm_amd import hostlib
    trace_bin, memory_bin, pi = padded_statement(layout, log_steps)
    gen = hostlib.starknet_base_trace if layout == "starknet" else hostlib.recursive_base_trace
    want = gen(trace_bin, memory_bin, pi)
    got = device_columns(ctx, layout, trace_bin, memory_bin, pi, None)
    assert_same_columns(got, want)


def test_starknet_columns_of_the_references_bootloader_run_with_every_builtin(ctx):
    """the reference's own starknet-layout run (example/bootloader: 2^17 steps, two real Pedersen instances) with real range-check,
    ECDSA, bitwise, EC-op and Poseidon instances on top: a template per distinct instance, the dummies' beside"""
    from sandstorm_amd import hostlib
    from test_layout_starknet import real_instances, bootloader_run
    g = os.path.join(ROOT, "tests", "golden")
    with gzip.open(os.path.join(g, "bootloader", "trace.bin.gz")) as f:
        trace_bin = f.read()
    with gzip.open(os.path.join(g, "bootloader", "memory.bin.gz")) as f:
        memory_bin = f.read()
    _, _, pi, priv = bootloader_run()
    both = dict(real_instances(), pedersen=priv["pedersen"])
    want = hostlib.starknet_base_trace(trace_bin, memory_bin, pi, both)
    got = device_columns(ctx, "starknet", trace_bin, memory_bin, pi, both)
    assert_same_columns(got, want)


def test_input_errors_are_the_generators_refusals(ctx):
    """what the host generator refuses, the device path refuses (status bits of the kernels -> the same messages' substance)"""
    from sandstorm_amd import hostlib
    from sandstorm_amd._lib import SandstormHipError
    trace_bin, memory_bin, pi = example_files()
    # a cell the run reads is missing from memory.bin: drop the record of the first instruction
    pc0 = int.from_bytes(trace_bin[16:24], "little")
    recs = [memory_bin[o:o + 40] for o in range(0, len(memory_bin), 40)]
    missing = b"".join(r for r in recs if int.from_bytes(r[:8], "little") != pc0)
    with pytest.raises(SandstormHipError, match="does not hold"):
        device_columns(ctx, "recursive", trace_bin, missing, pi, None)
    # not an instruction: the word at pc with bit 63 set
    broken = b"".join((r[:8] + (int.from_bytes(r[8:16], "little") | (1 << 63)).to_bytes(8, "little") + r[16:]) if int.from_bytes(r[:8], "little") == pc0 else r for r in recs)
    with pytest.raises(SandstormHipError, match="not an instruction"):
        device_columns(ctx, "recursive", trace_bin, broken, pi, None)
    # memory that is not single-valued: the public input declares another value for an address the run reads
    import copy
    bad = copy.deepcopy(pi)
    k = next(i for i, e in enumerate(bad.public_memory) if e[0] > 1)
    bad.public_memory[k] = (bad.public_memory[k][0], bad.public_memory[k][1] + 1)
    with pytest.raises(SandstormHipError, match="continuous and single-valued"):
        device_columns(ctx, "recursive", trace_bin, memory_bin, bad, None)
    with pytest.raises(SandstormHipError, match="continuous and single-valued"):
        hostlib.recursive_base_trace(trace_bin, memory_bin, bad)
    # an instance index beyond the trace, an instance given twice, more range-check instances than slots
    with pytest.raises(SandstormHipError, match="beyond the trace"):
        device_columns(ctx, "recursive", trace_bin, memory_bin, pi, {"pedersen": [(1 << 20, 1, 2)]})
    with pytest.raises(SandstormHipError, match="given twice"):
        hostlib.recursive_base_trace(trace_bin, memory_bin, pi, {"bitwise": [(3, 1, 2), (3, 4, 5)]})
    with pytest.raises(SandstormHipError, match="more range-check instances"):
        hostlib.recursive_base_trace(trace_bin, memory_bin, pi, {"range_check": [(i, 5) for i in range((1 << 14) // 8 + 1)]})
    # ... and the context is as good as before
    want = hostlib.recursive_base_trace(trace_bin, memory_bin, pi)
    assert_same_columns(device_columns(ctx, "recursive", trace_bin, memory_bin, pi, None), want)


@pytest.mark.skipif(EMULATED, reason="the bench's size: hardware only")
@pytest.mark.parametrize("layout", ["recursive", "starknet"])
def test_columns_at_2p20_steps(ctx, layout):
    """BASELINE configs[2]'s size: 2^24 rows per column, every cell against the host generator"""
    from sandstorm_amd import hostlib
    trace_bin, memory_bin, pi = padded_statement(layout, 20)
    gen = hostlib.starknet_base_trace if layout == "starknet" else hostlib.recursive_base_trace
    n = 16 << 20
    cols = hostlib.device_base_trace(ctx, layout, trace_bin, memory_bin, pi, None)
    want = gen(trace_bin, memory_bin, pi)
    for c, col in enumerate(cols):
        got = col.download(np.uint64, (n, 4))
        assert np.array_equal(got, want[c]), "column %d" % c
        col.free()


@pytest.mark.skipif(EMULATED, reason="BASELINE configs[3]'s size: hardware only")
def test_starknet_columns_at_2p22_steps(ctx):
    """BASELINE configs[3]'s statement size (starknet layout, 2^22 steps): 2^26 rows per column, 19 GB of columns from 101 MB of files -
    every cell against the host generator (32-bit pool addresses, the u32 prefix sums and the 2^25-entry count arrays at their largest)"""
    from sandstorm_amd import hostlib
    trace_bin, memory_bin, pi = padded_statement("starknet", 22)
    n = 16 << 22
    cols = hostlib.device_base_trace(ctx, "starknet", trace_bin, memory_bin, pi, None)
    want = hostlib.starknet_base_trace(trace_bin, memory_bin, pi)
    for c, col in enumerate(cols):
        got = col.download(np.uint64, (n, 4))
        assert np.array_equal(got, want[c]), "column %d" % c
        col.free()
        want[c] = None
