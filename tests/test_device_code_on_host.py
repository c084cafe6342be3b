"""The library's DEVICE CODE on the CPU.  tests/hipemu/ compiles sandstorm_amd/csrc/*.hip - every kernel, every launch, the C ABI's
host side - for the host over a stand-in for <hip/hip_runtime.h> (workgroups one after the other, lanes as fibers that yield at
barriers), into tests/hipemu/_build/libsandstorm_hipemu.so.  Selected `gpu` parity tests then run in a process of their own with
that library in place of the product's (tests/conftest.py, SS_TEST_HIPEMU=1) - the same tests, oracle and tolerances (bit-exact) the
MI355X runs, minus the sizes that only make sense on the hardware.

TEST INFRASTRUCTURE: this is not a fallback.  Nothing under sandstorm_amd/ can load the emulation; the product loads
sandstorm_amd/_build/libsandstorm_hip.so and fails without an MI355X.  What this buys is that index arithmetic, launch geometry,
lazy-bound bookkeeping and generated kernels are checked on every CPU run, and that kernel work can be verified before GPU time is
spent on measuring it.  What it cannot see: anything the gfx950 compiler or hardware does (spills, LDS limits, wave-level timing)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")


@pytest.fixture(scope="module")
def emulated_library():
    if not (os.path.exists(CLANG) or shutil.which(CLANG)):
        pytest.skip("no clang++ to build the host emulation with (%s)" % CLANG)
    out = subprocess.run(["bash", os.path.join(ROOT, "tests", "hipemu", "build.sh")], capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return out.stdout.strip().splitlines()[-1]


def heavy():
    """the selections that take minutes without a handful of cores (workgroups are spread over OS threads): skipped on small machines"""
    if (os.cpu_count() or 1) < 4 and os.environ.get("SS_TEST_HIPEMU_ALL") != "1":
        pytest.skip("fewer than 4 cores: set SS_TEST_HIPEMU_ALL=1 to run this selection anyway")


def run_gpu_tests_on_host(lib, args, timeout=1500, **more_env):
    # HIPEMU_ORDER=shuffle: between two barriers the lanes of a workgroup run in an order that changes with every pass and workgroup
    # (any order is a legal schedule; code that is missing a barrier passes in one and fails in another)
    env = dict(os.environ, SS_TEST_HIPEMU="1", SS_TEST_HIPEMU_LIB=lib, HIPEMU_ORDER=os.environ.get("HIPEMU_ORDER", "shuffle"), **more_env)
    out = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=timeout)
    tail = out.stdout[-3000:] + out.stderr[-2000:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and " failed" not in out.stdout and " error" not in out.stdout, tail
    return out.stdout


def test_every_kernel_against_the_oracle(emulated_library):
    """tests/test_gpu_parity.py: transforms, row hashing, Keccak / Blake2s / Pedersen trees and openings, FRI folds, DEEP, the constraint
    VM, proof of work - the 252-bit path's parity tests, all but the two that only exist for their size"""
    # (the trees of this file are small: the 32-lanes-per-hash Pedersen kernel serves their levels as it does on the device)
    out = run_gpu_tests_on_host(emulated_library, ["tests/test_gpu_parity.py", "-k", "not large_impulse"], SS_PED_SMALL_MAX="1024")
    assert "207 passed" in out, out[-500:]


def test_sweeps_over_the_sizes_the_gpu_suite_samples(emulated_library):
    """tests/hipemu/extra_fp252_sweeps.py: every transform size, extension factor, fold factor x layer length, row width x hash,
    tree size x tree kind, the small out-of-domain / DEEP sizes - cheap here, so all of them; tests/hipemu/extra_boundaries.py: DEEP masks
    that sit on the kernels' bookkeeping boundaries (15 ... 130 cells in one column, wrapped offsets, doubled cells), forty more random
    constraint programs"""
    out = run_gpu_tests_on_host(emulated_library, ["tests/hipemu/extra_fp252_sweeps.py", "tests/hipemu/extra_boundaries.py"])
    assert "70 passed" in out, out[-500:]                     # 61 sweeps + DEEP masks on the kernels' bookkeeping boundaries (both fields) + 40 random programs


def test_the_64_bit_field(emulated_library):
    """tests/test_goldilocks.py below the benchmark sizes, every transform size (tests/hipemu/extra_gl64_sizes.py), and a whole proof
    of the plain layout equal to the one the MI355X wrote, and one under the SHA-256 claim (tests/hipemu/extra_gl64_proof.py)"""
    # (the whole proof: the Python host over the C ABI with CPU tensors as device buffers writes the MI355X-made fixture)
    out = run_gpu_tests_on_host(emulated_library, ["tests/test_goldilocks.py", "tests/hipemu/extra_gl64_sizes.py", "tests/hipemu/extra_gl64_proof.py",
                                                   "-k", "not benchmark_size"])
    assert "61 passed" in out, out[-500:]                     # 34 + 17 sizes + folds, row shapes (SHA-256's padding edges too), running products + 2 proofs + the C++ host's two


def test_extension_columns_and_compiled_constraint_kernels(emulated_library):
    """tests/test_gpu_extension.py (the scans behind Trace::build_extension_columns) and tests/test_gpu_real_quotient.py (the generated
    starknet / recursive kernels against the interpreter and the oracle over whole domains)"""
    heavy()
    out = run_gpu_tests_on_host(emulated_library, ["tests/test_gpu_extension.py", "tests/test_gpu_real_quotient.py", "-k", "not back_to_back"])
    assert "53 passed" in out, out[-500:]                     # 48 (12 of them the scans over row blocks) + 5 (the sixth queues 2^20-point evaluations back to back: hardware only)


def test_the_base_trace_made_by_the_device_code(emulated_library):
    """tests/test_gpu_device_trace.py: csrc/trace.hip (one lane per Cairo cycle, builtin templates, the pools and the ordered memory by
    histogram + prefix sum + binary search) against the host generator, cell for cell - the reference's example run with and without
    real builtin instances, the bench's statements of both layouts, the reference's bootloader run with every builtin, the input's
    errors - and the 2^14-step files -> proof call through it (the committed proof's bytes)"""
    out = run_gpu_tests_on_host(emulated_library, ["tests/test_gpu_device_trace.py"])
    assert "6 passed, 3 skipped" in out, out[-500:]
    heavy()
    out = run_gpu_tests_on_host(emulated_library, ["tests/test_gpu_recursive_claim.py", "-k", "device_generator and 14"], timeout=2400)
    assert "1 passed" in out, out[-500:]


def test_whole_proofs(emulated_library):
    """tests/test_gpu_prove.py: prove -> serialise -> verify on the mini AIR, both hosts, every tree and coin"""
    heavy()
    run_gpu_tests_on_host(emulated_library, ["tests/test_gpu_prove.py"])


def test_row_block_entry_points(emulated_library):
    """tests/test_gpu_row_blocks.py: the row-block entry points against the whole-domain ones (halo, refusal, DEEP blocks + extension); ONE
    proof over several ranks is test_cpp_sharded_prover_over_ranks_as_threads / _as_processes below"""
    heavy()
    out = run_gpu_tests_on_host(emulated_library, ["tests/test_gpu_row_blocks.py"])
    assert "1 passed" in out, out[-500:]


def test_entry_points_refuse_what_they_cannot_serve(emulated_library):
    """tests/hipemu/extra_bad_arguments.py: every entry point with a NULL context, with everything else zero / NULL, with huge sizes and
    NULL data - an error status each time, no crash (each call in a forked child), nothing launched"""
    run_gpu_tests_on_host(emulated_library, ["tests/hipemu/extra_bad_arguments.py"])


def test_the_smoke_invocation(emulated_library):
    """__graft_entry__.smoke() - what the driver runs on the MI355X before the bench - over the host build: its checks are right"""
    code = ("import sys; sys.path.insert(0, %r); from sandstorm_amd import _lib; _lib.LIB_PATH = %r; import __graft_entry__ as g; g.smoke()"
            % (ROOT, emulated_library))
    env = dict(os.environ, SS_PED_WINDOW="16", SS_PED_SMALL_MAX="128")          # the emulated device's sizes (tests/conftest.py)
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "smoke ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_cpp_sharded_prover_over_ranks_as_threads(emulated_library):
    """tests/hipemu/extra_sharded_host.py: the C++ host's sharded prover (sandstorm_amd/host/sharded.cpp) on 1, 2, 4 and 8 ranks - threads
    of one process, each with its own emulated context, meeting in the LocalTransport: the MI355X-made single-device proofs byte for
    byte (mini AIR; the reference's example with the real recursive AIR under the CairoVerifierClaim), friendly trees with the
    Blake2s / Pedersen boundary inside"""
    heavy()
    os.environ["HIPEMU_THREADS"] = "1"           # the emulator's worker pool serves one launching thread: the ranks are the parallelism
    try:
        out = run_gpu_tests_on_host(emulated_library, ["tests/hipemu/extra_sharded_host.py"])
    finally:
        del os.environ["HIPEMU_THREADS"]
    assert "14 passed" in out, out[-500:]                     # 7 mini + 2 friendly + 4 real AIR (2 with the extension trace as row blocks) + the too-few-rows error


def test_cpp_sharded_prover_over_ranks_as_processes(emulated_library):
    """tests/hipemu/extra_sharded_host_procs.py: the same driver with the ranks as 2, 4 and 8 PROCESSES under torch.distributed.run - what
    `bench.py --gpus N` starts -, every process with its own emulated device, coin and columns, meeting in the driver's
    CallbackTransport over gloo: the single-device proofs byte for byte (mini AIR; friendly trees + Cairo coin; the real recursive
    AIR with a spread base column)"""
    heavy()
    out = run_gpu_tests_on_host(emulated_library, ["tests/hipemu/extra_sharded_host_procs.py"])
    assert "10 passed" in out, out[-500:]                     # 3 mini + 2 friendly + 3 real AIR (one with the extension trace as row blocks) + the group self check on 2 and 8
