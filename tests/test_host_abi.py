"""No GPU: the C-ABI library loads and exports every symbol include/sandstorm_hip.h
declares; argument validation and the no-device error path work; nothing under
sandstorm_amd/ reaches for the oracle."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "sandstorm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from sandstorm_amd import _lib
    lib = _lib.load()
    declared = header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), name
    assert set(declared) == set(_lib.SIGNATURES), set(declared) ^ set(_lib.SIGNATURES)
    from sandstorm_amd import _lib
    assert lib.ss_abi_version() == _lib.header_abi_version() == 12


def test_boundary_document_and_bindings_follow_the_header():
    """INTEGRATION.md's `extern "C"` block is generated from include/sandstorm_hip.h (tools/gen_integration.py) and must not be
    stale; the ctypes table of sandstorm_amd/_lib.py has every entry point with the header's argument count, a pointer where the
    header has a pointer and an integer of the header's width elsewhere (a hand-written block had ss_bitrev_permute32
    in place and ss_comm_all_gather's arguments in another order)"""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_integration as gi
    from sandstorm_amd import _lib
    with open(gi.DOC) as f:
        doc = f.read()
    assert gi.render(doc) == doc, "INTEGRATION.md section 1 is stale: run python tools/gen_integration.py"
    protos = gi.prototypes()
    assert len(protos) == len(_lib.SIGNATURES) >= 60
    width = {"u64": 8, "u32": 4, "u8": 1, "c_int": 4, "usize": C.sizeof(C.c_size_t), "i64": 8, "f64": 8}
    for name, ret, params in protos:
        restype, argtypes = _lib.SIGNATURES[name]
        assert len(argtypes) == len(params), name
        for (ctype, pname), at in zip(params, argtypes):
            rt = gi.rust_type(ctype).split(" /*")[0]
            if rt.startswith("*"):
                assert at in (C.c_void_p, C.c_char_p) or hasattr(at, "contents") or issubclass(at, C._Pointer), (name, pname)
            else:
                assert C.sizeof(at) == width[rt] and not issubclass(at, C._Pointer) and at not in (C.c_void_p, C.c_char_p), (name, pname, rt)
    # the two the hand-written block had wrong
    by_name = {n: [p for _, p in ps] for n, _, ps in protos}
    assert by_name["ss_bitrev_permute32"] == ["ctx", "d_src", "log_n", "d_dst"]
    assert by_name["ss_comm_all_gather"] == ["comm", "d_send", "bytes", "d_recv"]


def test_no_device_fails_loudly():
    """In a GPU-less process ss_ctx_create must fail with a message, never fall back."""
    code = ("import sandstorm_amd; from sandstorm_amd.backend import Context\n"
            "try:\n    Context(0); print('CREATED')\nexcept sandstorm_amd.SandstormHipError as e:\n    print('ERR', e)\n")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert "ERR" in out.stdout and "CREATED" not in out.stdout, out.stdout + out.stderr


def test_host_pedersen_matches_golden(golden, oracle):
    """ss_pedersen_hash_host (the coin's host hash, product code) against the reference KATs."""
    from sandstorm_amd import backend as be
    g = golden("pedersen.json")
    for case in g["hash_examples"] + g["extra"]:
        a, b = be.felt(int(case["a"])), be.felt(int(case["b"]))
        got = be.pedersen_hash_host(a, b)
        assert int(oracle.from_mont(got)) == int(case["hash"])


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sandstorm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_py" not in text and "liboracle" not in text and "from oracle" not in text, f
                assert not re.search(r"#include\s+\"[^\"]*oracle", text), f


def test_host_coins_match_oracle(oracle, golden):
    """sandstorm_amd.coin (host product code) replays the reference coin KATs and agrees with the oracle coin."""
    from sandstorm_amd import backend as be
    from sandstorm_amd.coin import PublicCoin, canonical
    g = golden("coins.json")
    c = PublicCoin(be.COIN_SOLIDITY, bytes(32))
    for want in g["solidity_zero_seed_draws"]:
        assert canonical(c.draw()) == int(want)
    k = g["cairo_reseed"]
    c = PublicCoin(be.COIN_CAIRO, bytes.fromhex(k["seed"]))
    c.reseed_with_bytes(int(k["element"]).to_bytes(32, "big"))
    assert c.digest.hex() == k["digest"]
    # longer mixed transcript against the oracle implementation
    import numpy as np
    from tests.util import random_column
    for kind in (0, 1):
        a, b = PublicCoin(kind, bytes(range(32))), oracle.Coin(kind, bytes(range(32)))
        felts = random_column(5, 3)
        a.reseed_with_field_elements(list(felts)); b.reseed_felts(felts)
        assert a.digest == b.digest
        assert np.array_equal(a.draw(), b.draw())
        a.reseed_with_field_element_vector(list(felts)); b.reseed_felt_vector(felts)
        a.reseed_with_int(12345); b.reseed_int(12345)
        assert a.draw_queries(9, 1 << 12) == b.draw_queries(9, 1 << 12)
        assert a.digest == b.digest and a.counter == b.counter


def test_host_library_document_names_what_the_library_exports():
    """INTEGRATION.md section 4 against libsandstorm_host.so (the table still listed an entry point that had left the library):
    every `ssh_*` the document names is exported, every exported `ssh_*` entry point of host_capi.cpp is either named there or one of
    the test / diagnostic hooks listed here, and the bindings check the host ABI version at load"""
    import ctypes as C
    from sandstorm_amd import hostlib
    lib = hostlib.load()
    assert lib.ssh_abi_version() == hostlib.HOST_ABI_VERSION == 4
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        doc = f.read()
    named = set(re.findall(r"\b(ssh_[a-z0-9_]+)\b", doc)) - {"ssh_air", "ssh_matrix", "ssh_coin"}          # handle types
    for name in sorted(named):
        C.cast(getattr(lib, name), C.c_void_p)              # AttributeError: the document names something the library lacks
    src = open(os.path.join(ROOT, "sandstorm_amd", "host", "host_capi.cpp")).read()
    defined = set(re.findall(r"^[a-z_0-9 \*]*?\b(ssh_[a-z0-9_]+)\(", src, flags=re.M))
    hooks = {"ssh_last_error", "ssh_free", "ssh_air_destroy", "ssh_air_columns", "ssh_air_dump", "ssh_air_program", "ssh_air_prepare_program", "ssh_air_mask",
             "ssh_air_num_challenges", "ssh_coin_new", "ssh_coin_free", "ssh_coin_op", "ssh_matrix_num_cols", "ssh_matrix_col",
             "ssh_matrix_destroy", "ssh_local_group_create", "ssh_local_group_destroy", "ssh_rccl_group_destroy", "ssh_prove",
             "ssh_prove_wire_with_nonce", "ssh_build_extension_columns", "ssh_public_coin_seed"}
    assert len(defined) >= 25
    assert defined - named - hooks == set(), defined - named - hooks
