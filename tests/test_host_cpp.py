"""No GPU: the C++ host library (libsandstorm_host.so) — its Fiat-Shamir coin against the
reference KATs and the oracle coin."""
import numpy as np
import pytest

from tests.util import random_column


def test_cpp_coin_kats_and_oracle(oracle, golden):
    from sandstorm_amd import hostlib
    g = golden("coins.json")
    c = hostlib.HostCoin(0, bytes(32))
    for want in g["solidity_zero_seed_draws"]:
        assert int(oracle.from_mont(c.draw())) == int(want)
    k = g["cairo_reseed"]
    c = hostlib.HostCoin(1, bytes.fromhex(k["seed"]))
    c.reseed_bytes(int(k["element"]).to_bytes(32, "big"))
    assert c.state[0].hex() == k["digest"]
    for kind in (0, 1):
        a, b = hostlib.HostCoin(kind, bytes(range(32))), oracle.Coin(kind, bytes(range(32)))
        felts = random_column(7, 3)
        a.reseed_felts(felts); b.reseed_felts(felts)
        assert a.state[0] == b.digest
        assert np.array_equal(a.draw(), b.draw())
        a.reseed_felt_vector(felts); b.reseed_felt_vector(felts)
        a.reseed_int(777); b.reseed_int(777)
        assert a.draw_queries(9, 1 << 12) == b.draw_queries(9, 1 << 12)
        assert a.state == (b.digest, b.counter)


def test_small_integers_to_montgomery_form(tmp_path):
    """felt_from_u64's short path (host/coin.cpp: v 2^256 mod p = p - (544 v 2^192 + 32 v) below 2^49, what the trace generators
    call by the million) equals the general Montgomery product (tests/cpp/felt_from_u64_test.cpp, linked against the host library)"""
    import os
    import subprocess
    from sandstorm_amd import hostlib
    hostlib.load()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build = os.path.join(root, "sandstorm_amd", "_build")
    exe = str(tmp_path / "felt_from_u64")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tests", "cpp", "felt_from_u64_test.cpp"),
                           "-L" + build, "-lsandstorm_host", "-Wl,-rpath," + build])
    out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, LD_LIBRARY_PATH=build + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", "")))
    assert out.returncode == 0 and "FELT_FROM_U64_OK" in out.stdout, out.stdout + out.stderr


def test_trace_generators_write_every_cell_of_the_callers_columns():
    """the C++ base-trace generators fill the caller's (pinned) columns in place and only pre-zero the columns that have cells no
    section writes (host/trace_recursive.cpp, trace_starknet.cpp): from poisoned columns they must leave the trace a fresh call
    returns - the padded example of both layouts, and the reference's bootloader run with real instances of every builtin"""
    from sandstorm_amd import binary, examples, hostlib
    from tests.test_layout_starknet import bootloader_run, real_instances
    cases = []
    for layout, log_steps in (("recursive", 14), ("recursive", 15), ("starknet", 17)):
        states, memory, pi = (examples.recursive_example if layout == "recursive" else examples.starknet_example)(log_steps)
        cases.append((layout, binary.write_register_states(states), binary.write_memory(memory), pi, None))
    states, memory, pi, priv = bootloader_run()
    more = dict(real_instances())
    more["pedersen"] = priv["pedersen"] + more["pedersen"]
    cases.append(("starknet", binary.write_register_states(states), binary.write_memory(memory), pi, more))
    for layout, trace_bin, memory_bin, pi, priv in cases:
        gen = hostlib.recursive_base_trace if layout == "recursive" else hostlib.starknet_base_trace
        want = gen(trace_bin, memory_bin, pi, priv)
        out = [np.full(c.shape, 0x7777777777777777, dtype=np.uint64) for c in want]
        gen(trace_bin, memory_bin, pi, priv, out=out)
        for c, (a, b) in enumerate(zip(out, want)):
            assert np.array_equal(a, b), (layout, c)


def test_the_files_entry_takes_the_private_input_as_the_column_entries_do():
    """host_capi.cpp make_trace_job - what ssh_prove_files and ssh_base_trace_cb build their generator call from - reads the builtin
    instances of air-private-input.json out of six flat arrays: with real instances of every builtin its columns are the ones
    ssh_recursive_base_trace / ssh_starknet_base_trace write (those are held against the Python mirror in tests/test_layout_*.py),
    both layouts, and an instance list the generator refuses comes back as its message"""
    import random
    from sandstorm_amd import binary, examples, hostlib
    from sandstorm_amd._lib import SandstormHipError
    from tests.test_layout_starknet import bootloader_run, real_instances
    states, memory, pi = examples.recursive_example(14)
    rng = random.Random(3)
    rec_priv = {"pedersen": [(0, rng.getrandbits(250), rng.getrandbits(250)), (5, 0, 5), (7, (1 << 251) | (1 << 196) | (1 << 192), 1)],
                "bitwise": [(i, rng.getrandbits(251), rng.getrandbits(251)) for i in range(4)],
                "range_check": [(i, sum(rng.randrange(32700, 32800) << (16 * k) for k in range(8))) for i in range(3)]}
    cases = [("recursive", binary.write_register_states(states), binary.write_memory(memory), pi, rec_priv)]
    states, memory, pi, priv = bootloader_run()
    more = dict(real_instances())
    more["pedersen"] = priv["pedersen"] + more["pedersen"]
    cases.append(("starknet", binary.write_register_states(states), binary.write_memory(memory), pi, more))
    for layout, trace_bin, memory_bin, pi, private in cases:
        plain = hostlib.recursive_base_trace if layout == "recursive" else hostlib.starknet_base_trace
        want = plain(trace_bin, memory_bin, pi, private)
        got, _ = hostlib.base_trace_with_callback(layout, trace_bin, memory_bin, pi, private)
        assert len(got) == len(want)
        for c, (a, b) in enumerate(zip(got, want)):
            assert np.array_equal(a, b), (layout, c)
        dummies = plain(trace_bin, memory_bin, pi, {"pedersen": private["pedersen"][:2]} if layout == "starknet" else None)
        assert any(not np.array_equal(a, b) for a, b in zip(want, dummies))         # (the instances do reach the cells)
    with pytest.raises(SandstormHipError, match="signature is invalid"):
        bad = dict(more)
        bad["ecdsa"] = [(1,) + tuple(more["ecdsa"][0][1:4]) + (more["ecdsa"][0][4] + 1,)]
        hostlib.base_trace_with_callback("starknet", trace_bin, memory_bin, pi, bad)
