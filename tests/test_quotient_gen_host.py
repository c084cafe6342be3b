"""The generated constraint kernels, checked without a GPU: the body of csrc/quotient_gen_<layout>.inc (what
tools/gen_quotient.py writes and the device kernel includes) compiled for the host (tests/cpp/quotient_gen_host_test.cpp)
and run over a whole evaluation domain of random columns, tables and constants, against the oracle's constraint VM on
the program it was generated from - bit for bit; also in the row-block form of the sharded prover, and with a "grid"
that does not divide the domain (the rotating prefetch registers cross the loop edge at every point).  A compiled program
is several kernels (tools/gen_quotient.py split_program): the parts run one after the other, the first storing and the others
adding into the output, as the device launches them."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from tests.test_gpu_real_quotient import _rand
from tests.test_layout_recursive import load_run
from tests.test_layout_starknet import CHALLENGES, P, starknet_example

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp", "quotient_gen_host_test.cpp")


def part_bodies(layout):
    """the bodies of the layout's compiled program, in launch order: csrc/quotient_gen_<layout>_p<j>.inc"""
    import glob
    import re
    csrc = os.path.join(ROOT, "sandstorm_amd", "csrc")
    found = {int(re.search(r"_p(\d+)\.inc$", f).group(1)): f for f in glob.glob(os.path.join(csrc, "quotient_gen_%s_p*.inc" % layout))}
    assert sorted(found) == list(range(len(found))) and found, found
    return [found[j] for j in range(len(found))]


def build(layout, tmp):
    exe = os.path.join(tmp, "qg_host_%s" % layout)
    parts = part_bodies(layout)
    with open(os.path.join(tmp, "qg_parts.h"), "w") as f:
        for j, inc in enumerate(parts):
            f.write("static void run_lane_p%d(HostArgs &a, uint64_t lane, uint64_t lanes) {\n    QG_LANE_PRELUDE\n#include \"%s\"\n}\n" % (j, inc))
        f.write("static const part_fn PARTS[] = {%s};\n" % ", ".join("run_lane_p%d" % j for j in range(len(parts))))
    scaled = os.path.join(ROOT, "sandstorm_amd", "csrc", "quotient_gen_%s_scaled.inc" % layout)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fopenmp", "-I", tmp, "-DQG_PARTS_H=\"qg_parts.h\"", "-DQG_SCALED_H=\"%s\"" % scaled,
                           "-o", exe, CPP])
    return exe


def program(oracle, layout, log_n):
    from sandstorm_amd import hostlib
    if layout == "starknet":
        from sandstorm_amd.layouts import starknet as lay
        _, _, pi = starknet_example(11)
        cpp = hostlib.StarknetHostAir(None, pi, log_n)
    else:
        from sandstorm_amd.layouts import recursive as lay
        _, _, pi = load_run()
        cpp = hostlib.RecursiveHostAir(None, pi, log_n)
    n = 1 << log_n
    code, consts, n_slots, specs = cpp.dump(n, [oracle.to_mont([c])[0] for c in CHALLENGES], oracle.to_mont([pow(5, 77, P)])[0])
    cpp.close()
    return lay, code, consts, n_slots, specs


def run_host(exe, tmp, cols, tab, desc, consts, npoints, row0, trace_mask, lb, lanes, offset, w):
    path_in, path_out = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
    with open(path_in, "wb") as f:
        f.write(struct.pack("<10Q", len(cols), len(cols[0]), len(tab), len(desc) // 2, len(consts), npoints, row0, trace_mask, lb, lanes))
        for c in cols:
            f.write(np.ascontiguousarray(c).tobytes())
        f.write(np.ascontiguousarray(tab).tobytes())
        td = []
        for k in range(0, len(desc), 2):
            td += [desc[k], (1 << desc[k + 1]) - 1]
        f.write(np.asarray(td, dtype=np.uint32).tobytes())
        f.write(np.ascontiguousarray(consts).tobytes())
        f.write(np.ascontiguousarray(offset).tobytes())
        f.write(np.ascontiguousarray(w).tobytes())
    subprocess.check_call([exe, path_in, path_out], env=dict(os.environ, OMP_NUM_THREADS=str(min(8, os.cpu_count() or 1))))
    return np.fromfile(path_out, dtype=np.uint64).reshape(npoints, 4)


@pytest.mark.parametrize("layout,log_n", [("recursive", 14), ("starknet", 16)])
def test_generated_kernel_body_on_the_host(oracle, layout, log_n, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_quotient
    tmp = str(tmp_path)
    lay, code, consts, n_slots, specs = program(oracle, layout, log_n)
    with open(os.path.join(ROOT, "sandstorm_amd", "csrc", "quotient_gen_%s.hip" % layout)) as f:
        assert ("0x%016x" % gen_quotient.code_hash(code)) in f.read(), "the committed kernel was generated from another program"
    exe = build(layout, tmp)
    n, N = 1 << log_n, 2 << log_n
    tables = lay.Tables(n)
    rng = np.random.default_rng(23)
    tabs, desc, off = [], [], 0
    for spec in specs:
        t = _rand(rng, tables.length(spec))
        desc += [off, len(t).bit_length() - 1]
        off += len(t)
        tabs.append(t)
    tab = np.concatenate(tabs)
    lde = [_rand(rng, N) for _ in range(10)]
    g = oracle.to_mont([3])[0]
    w = oracle.to_mont([pow(3, (P - 1) // N, P)])[0]
    want = oracle.eval_program(code, consts, tab, desc, n_slots, lde, log_n, 1, g)
    # whole domain; 96 lanes do not divide it evenly and make every lane loop many times over the loop edge
    got = run_host(exe, tmp, lde, tab, desc, consts, N, 0, N - 1, 1, 96, g, w)
    assert np.array_equal(got, want)
    # row-block form (ss_eval_quotient_rows): the third quarter of the domain with the rows behind it
    B = N // 4
    halo = max((int(c) & 0xffffff) for c in code[1::2][(code[0::2] >> 12) & 0xf == 3]) << 1
    idx = (2 * B + np.arange(B + halo)) % N
    got = run_host(exe, tmp, [c[idx] for c in lde], tab, desc, consts, B, 2 * B, 0xffffffff, 1, 64, g, w)
    assert np.array_equal(got, want[2 * B:3 * B])


def _extreme(rng, count):
    """canonical 256-bit images at the edges of the lazy limb forms: 0, 1, p - 1, p - 2, 2^251 - 1 (every limb below the top
    one full), single full 28-bit limbs, and a few random values in between (4 x u64 little endian, like _rand)"""
    pool = [0, 1, 2, P - 1, P - 2, (1 << 251) - 1, (1 << 251), (1 << 251) + (17 << 192), (P - 1) // 2, (1 << 224) - 1,
            ((1 << 28) - 1) << 28, ((1 << 28) - 1) << 168, ((1 << 28) - 1) << 196, sum(((1 << 28) - 1) << (56 * k) for k in range(4))]
    pool += [int(rng.integers(0, 1 << 62)) * int(rng.integers(0, 1 << 62)) * int(rng.integers(0, 1 << 62)) % P for _ in range(6)]
    table = np.array([[(v >> (64 * k)) & ((1 << 64) - 1) for k in range(4)] for v in pool], dtype=np.uint64)
    return table[rng.integers(0, len(pool), size=count)]


@pytest.mark.parametrize("layout,log_n", [("recursive", 14), ("starknet", 16)])
def test_generated_kernel_body_on_values_at_the_limb_forms_edges(oracle, layout, log_n, tmp_path):
    """the generator places lazy-form operations by BOUNDS (sums of two values subtracted unreduced, products of lazy factors, a
    constraint's own wide sum with its L 2^256 term, negated factors): random columns hardly ever reach those bounds - columns,
    tables and constants drawn from the values at the limb forms' edges do"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_quotient
    tmp = str(tmp_path)
    lay, code, consts, n_slots, specs = program(oracle, layout, log_n)
    with open(os.path.join(ROOT, "sandstorm_amd", "csrc", "quotient_gen_%s.hip" % layout)) as f:
        assert ("0x%016x" % gen_quotient.code_hash(code)) in f.read(), "the committed kernel was generated from another program"
    exe = build(layout, tmp)
    n, N = 1 << log_n, 2 << log_n
    tables = lay.Tables(n)
    rng = np.random.default_rng(29)
    tabs, desc, off = [], [], 0
    for spec in specs:
        t = _extreme(rng, tables.length(spec))
        desc += [off, len(t).bit_length() - 1]
        off += len(t)
        tabs.append(t)
    tab = np.concatenate(tabs)
    lde = [_extreme(rng, N) for _ in range(10)]
    consts = np.array(consts, dtype=np.uint64).reshape(-1, 4).copy()
    edge = _extreme(rng, len(consts))
    pick = rng.random(len(consts)) < 0.5                      # half of the constants at the edges too
    consts[pick] = edge[pick]
    g = oracle.to_mont([3])[0]
    w = oracle.to_mont([pow(3, (P - 1) // N, P)])[0]
    want = oracle.eval_program(code, consts, tab, desc, n_slots, lde, log_n, 1, g)
    got = run_host(exe, tmp, lde, tab, desc, consts, N, 0, N - 1, 1, 96, g, w)
    assert np.array_equal(got, want)


def test_committed_kernels_are_what_the_generator_writes(tmp_path):
    """the library recognises a compiled program by the hash of its code words only: a change to tools/gen_quotient.py (or to its
    per-part tuning table) that was not followed by a regeneration would ship stale kernels silently"""
    import filecmp
    import glob
    out = str(tmp_path)
    env = {k: v for k, v in os.environ.items() if not k.startswith("QG_")}
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_quotient.py")], env=dict(env, QG_OUT_DIR=out), stdout=subprocess.DEVNULL)
    made = sorted(os.path.basename(f) for f in glob.glob(os.path.join(out, "quotient_gen_*")))
    assert len(made) >= 2 * 7 + 2 + 2, made                    # parts (.hip + .inc), kernel tables, source lists
    csrc = os.path.join(ROOT, "sandstorm_amd", "csrc")
    stale = [f for f in made if not filecmp.cmp(os.path.join(out, f), os.path.join(csrc, f), shallow=False)]
    assert not stale, "regenerate with `python tools/gen_quotient.py`: %s" % stale
