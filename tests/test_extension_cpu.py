"""Extension-trace scans, CPU side:
  * the HIP kernels' lane bodies and level driver (sandstorm_amd/csrc/ext_scan.h, host+device code) run on the host
    against the reference's sequential loops;
  * the oracle restatement (oracle/ext.c) against the big-integer definition of the same loops
    (layouts/src/recursive/trace.rs:699-814), including ark-ff batch_inversion's zero rule;
  * oracle build_extension_columns: a genuine permutation closes to one (the reference's own assert)."""
import os
import subprocess

import numpy as np
import pytest

from tests.util import P, random_column

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scan_lane_bodies_on_host(tmp_path):
    exe = str(tmp_path / "ext_scan_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "ext_scan_test.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr


def _inv0(x):
    return pow(x, P - 2, P) if x else 0


@pytest.mark.parametrize("count", [1, 2, 7, 64, 65, 200])
def test_oracle_permutation_product_definition(oracle, count):
    a = oracle.from_mont(random_column(2 * count, 1))
    b = oracle.from_mont(random_column(2 * count, 2))
    z, alpha = 0x1234567 ** 7 % P, 0x7654321 ** 9 % P
    am, bm = oracle.to_mont(list(a)), oracle.to_mont(list(b))
    out = np.zeros((2 * count, 4), dtype=np.uint64)
    out[:] = oracle.to_mont([99])[0]                      # cells the product does not own must stay untouched
    last = oracle.permutation_product((am, 2, 0, 1), (bm, 2, 0, 1), count, oracle.to_mont([z])[0], oracle.to_mont([alpha])[0],
                                      out, 2, 1)
    got = oracle.from_mont(out)
    nacc = dacc = 1
    for k in range(count):
        nacc = nacc * (z - (alpha * int(a[2 * k + 1]) + int(a[2 * k]))) % P
        dacc = dacc * (z - (alpha * int(b[2 * k + 1]) + int(b[2 * k]))) % P
        assert got[2 * k + 1] == nacc * _inv0(dacc) % P and got[2 * k] == 99
    assert oracle.from_mont(last[None])[0] == got[2 * count - 1]


def test_oracle_permutation_product_zero_denominator(oracle):
    """ark-ff batch_inversion leaves a zero product zero: everything from the zero term on is n * 0 = 0"""
    count, z = 20, 31337
    x = [int(v) for v in oracle.from_mont(random_column(4 * count, 3))]
    x[4 * 9 + 2] = z                                       # ordered value of item 9 equals z
    xm = oracle.to_mont(x)
    out = np.zeros((count, 4), dtype=np.uint64)
    oracle.permutation_product((xm, 4, 0, -1), (xm, 4, 2, -1), count, oracle.to_mont([z])[0], np.zeros(4, dtype=np.uint64), out)
    got = oracle.from_mont(out)
    nacc = dacc = 1
    for k in range(count):
        nacc = nacc * (z - x[4 * k]) % P
        dacc = dacc * (z - x[4 * k + 2]) % P
        assert got[k] == nacc * _inv0(dacc) % P
        assert (got[k] == 0) == (k >= 9)


@pytest.mark.parametrize("count", [1, 2, 65, 130])
def test_oracle_diluted_aggregate_definition(oracle, count):
    x = [int(v) for v in oracle.from_mont(random_column(8 * count, 4))]
    z, alpha = 3 ** 100 % P, 5 ** 90 % P
    out = np.zeros((8 * count, 4), dtype=np.uint64)
    oracle.diluted_aggregate(oracle.to_mont(x), 8, 5, count, oracle.to_mont([z])[0], oracle.to_mont([alpha])[0], out, 8, 3)
    got = oracle.from_mont(out)
    acc = 1
    assert got[3] == 1
    for i in range(1, count):
        u = (x[8 * i + 5] - x[8 * (i - 1) + 5]) % P
        acc = (acc * (1 + z * u) + alpha * u * u) % P
        assert got[8 * i + 3] == acc
    assert all(got[j] == 0 for j in range(8 * count) if j % 8 != 3)


def permuted_trace(oracle, layout, n, seed=0):
    """auxiliary columns whose range-check / diluted-check / memory multisets really are permutations
    (canonical ints -> Montgomery), so the reference's `is_one` asserts hold"""
    rng = np.random.default_rng(seed)

    def small(k, bits):
        return [int(v) for v in rng.integers(0, 1 << bits, size=k)]
    npc = small(n, 40)
    pairs = sorted((npc[2 * i], npc[2 * i + 1]) for i in range(n // 2))
    memory = [v for pr in pairs for v in pr]
    rc = small(n, 16)
    if layout == "recursive":
        ordered = sorted(rc[4 * i] for i in range(n // 4))
        for i, v in enumerate(ordered):
            rc[4 * i + 2] = v
        du = small(n, 30)
        cols = {"npc": npc, "memory": memory, "range_check": rc, "diluted_unordered": du, "diluted_ordered": sorted(du)}
    else:
        # range check cells: offsets 0 (OffDst) / 2 (Ordered) of every 4; diluted cells: offsets 1 (Unordered) / 5 (Ordered) of every 8
        ordered = sorted(rc[4 * i] for i in range(n // 4))
        for i, v in enumerate(ordered):
            rc[4 * i + 2] = v
        dord = sorted(rc[8 * i + 1] for i in range(n // 8))
        for i, v in enumerate(dord):
            rc[8 * i + 5] = v
        cols = {"npc": npc, "memory": memory, "range_check": rc}
    return {k: oracle.to_mont(v) for k, v in cols.items()}


def challenges(oracle):
    return [oracle.to_mont([pow(7, 11 + 3 * i, P)])[0] for i in range(6)]


@pytest.mark.parametrize("layout", ["recursive", "starknet"])
def test_oracle_extension_columns_close_to_one(oracle, layout):
    n = 256
    cols = permuted_trace(oracle, layout, n)
    out, lasts = oracle.build_extension_columns(layout, cols, challenges(oracle), n)
    assert [int(v) for v in oracle.from_mont(np.stack(lasts))] == [1, 1, 1]
    assert len(out) == (3 if layout == "recursive" else 1) and all(c.shape == (n, 4) for c in out)
    if layout == "starknet":
        perm = oracle.from_mont(out[0])
        assert perm[3] == 1                                # DilutedCheck::Aggregate initial value (starknet/trace.rs:1082)
        # memory owns every even row, range check rows 1 (mod 4), diluted check row 7 and aggregate row 3 (mod 8)
        assert all(perm[i] != 0 for i in range(8))
