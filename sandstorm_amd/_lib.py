"""ctypes loader for libsandstorm_hip.so (the C ABI of include/sandstorm_hip.h).

There is no CPU fallback: if the shared library has not been built, or no
gfx950 device is usable, every entry point raises.  The oracle under oracle/
is test infrastructure and is never imported from this package.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libsandstorm_hip.so")

HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sandstorm_hip.h")

SS_OK = 0


def header_abi_version():
    """SS_ABI_VERSION of include/sandstorm_hip.h (the one definition of the boundary's version)."""
    import re
    with open(HEADER_PATH) as f:
        m = re.search(r"#define\s+SS_ABI_VERSION\s+(\d+)u?", f.read())
    if not m:
        raise SandstormHipError("SS_ABI_VERSION not found in %s" % HEADER_PATH)
    return int(m.group(1))


class SandstormHipError(RuntimeError):
    pass


class AirProgram(C.Structure):
    """ss_air_program"""
    _fields_ = [("code", C.POINTER(C.c_uint32)), ("n_instr", C.c_uint32),
                ("consts", C.POINTER(C.c_uint64)), ("n_consts", C.c_uint32),
                ("d_tables", C.c_void_p), ("table_desc", C.POINTER(C.c_uint32)),
                ("n_tables", C.c_uint32), ("n_slots", C.c_uint32)]


class GatherJob(C.Structure):
    """ss_gather_job"""
    _fields_ = [("d_cols", C.POINTER(C.c_void_p)), ("ncols", C.c_uint32), ("entry_bytes", C.c_uint32), ("idx", C.POINTER(C.c_uint64)),
                ("nidx", C.c_uint32), ("out", C.c_void_p)]


class PermOperand(C.Structure):
    """ss_perm_operand"""
    _fields_ = [("d_data", C.c_void_p), ("stride", C.c_uint64), ("addr_offset", C.c_uint64), ("value_offset", C.c_int64)]


_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)
_vpp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); every symbol include/sandstorm_hip.h declares
SIGNATURES = {
    "ss_last_error": (C.c_char_p, []),
    "ss_abi_version": (C.c_uint32, []),
    "ss_ctx_create": (C.c_int, [C.c_int, _vpp]),
    "ss_ctx_destroy": (None, [C.c_void_p]),
    "ss_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ss_ctx_sync": (C.c_int, [C.c_void_p]),
    "ss_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, _vpp]),
    "ss_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ss_ctx_trim": (C.c_int, [C.c_void_p]),
    "ss_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ss_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ss_dev_zero": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ss_dev_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ss_dev_copy_2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]),
    "ss_bitrev_permute32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "ss_comm_unique_id": (C.c_int, [C.c_char_p]),
    "ss_comm_create": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, _vpp]),
    "ss_comm_destroy": (None, [C.c_void_p]),
    "ss_comm_exchange": (C.c_int, [C.c_void_p, C.c_uint32, _u32p, _vpp, _u64p, C.c_uint32, _u32p, _vpp, _u64p]),
    "ss_comm_all_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "ss_permutation_product": (C.c_int, [C.c_void_p, C.POINTER(PermOperand), C.POINTER(PermOperand), C.c_uint64, _u64p, _u64p,
                                         C.c_void_p, C.c_uint64, C.c_uint64, _u64p]),
    "ss_diluted_aggregate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, _u64p, _u64p,
                                       C.c_void_p, C.c_uint64, C.c_uint64]),
    "ss_scale_strided": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, _u64p]),
    "ss_diluted_aggregate_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, _u64p, _u64p,
                                             C.c_void_p, _u64p]),
    "ss_affine_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, _u64p, C.c_void_p, C.c_uint64, C.c_uint64]),
    "ss_ntt_fp252": (C.c_int, [C.c_void_p, _vpp, C.c_uint32, C.c_uint32, C.c_int, _u64p, C.c_int, C.c_int]),
    "ss_lde_fp252": (C.c_int, [C.c_void_p, _vpp, C.c_uint32, C.c_uint32, C.c_uint32, _u64p, _vpp, _vpp]),
    "ss_evaluate_fp252": (C.c_int, [C.c_void_p, _vpp, C.c_uint32, C.c_uint32, C.c_uint32, _u64p, _vpp]),
    "ss_ntt_shard_fp252": (C.c_int, [C.c_void_p, _vpp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _u64p, C.c_int, C.c_uint32, _vpp]),
    "ss_hash_rows": (C.c_int, [C.c_void_p, C.c_int, _vpp, C.c_uint32, C.c_uint64, C.c_void_p]),
    "ss_hash_rows_ex": (C.c_int, [C.c_void_p, C.c_int, _vpp, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p]),
    "ss_merkle_build": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_uint64,
                                  C.c_void_p, C.c_void_p, _u8p]),
    "ss_merkle_build_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_uint64, C.c_int,
                                  C.c_void_p, C.c_void_p, _u8p]),
    "ss_merkle_open": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, _u64p, C.c_uint32,
                                 C.c_void_p, C.c_void_p]),
    "ss_gather_rows": (C.c_int, [C.c_void_p, _vpp, C.c_uint32, _u64p, C.c_uint32, C.c_void_p]),
    "ss_gather_batch": (C.c_int, [C.c_void_p, C.POINTER(GatherJob), C.c_uint32]),
    # the base trace on the device (ABI 11): structs as opaque pointers (the C++ host fills them: host/device_trace.hpp)
    "ss_trace_memory_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    "ss_trace_cpu_cells": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, _u64p, C.c_uint64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ss_trace_builtin": (C.c_int, [C.c_void_p, _vpp, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint64,
                                   C.c_uint64, C.c_uint64, C.c_void_p]),
    "ss_trace_rc_pool": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "ss_trace_rc_builtin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ss_trace_ordered_runs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]),
    "ss_trace_patch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]),
    "ss_trace_ordered_memory": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, _u64p,
                                          C.c_uint32, C.c_void_p]),
    "ss_trace_status": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ss_inverse_table": (C.c_int, [C.c_void_p, C.c_uint32, _u64p, _u64p, C.c_void_p]),
    "ss_eval_quotient": (C.c_int, [C.c_void_p, C.POINTER(AirProgram), _vpp, C.c_uint32, C.c_uint32,
                                   C.c_uint32, _u64p, C.c_void_p]),
    "ss_eval_quotient_rows": (C.c_int, [C.c_void_p, C.POINTER(AirProgram), _vpp, C.c_uint32, C.c_uint32,
                                        C.c_uint32, _u64p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "ss_deep_compose_rows": (C.c_int, [C.c_void_p, _vpp, C.c_uint32, _vpp, C.c_uint32, C.c_uint32, C.c_uint32,
                                       _u64p, _u32p, _u32p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, _u64p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "ss_deep_extend": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, _u64p, C.c_void_p]),
    "ss_ood_eval": (C.c_int, [C.c_void_p, _vpp, C.c_uint32, C.c_uint32, _u32p, _u32p, C.c_uint32, _u64p,
                              C.c_void_p]),
    "ss_poly_eval": (C.c_int, [C.c_void_p, _vpp, C.c_uint32, C.c_uint32, _u64p, C.c_void_p]),
    "ss_deep_prepare": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, _u64p, _u64p]),
    "ss_deep_compose": (C.c_int, [C.c_void_p, _vpp, C.c_uint32, _vpp, C.c_uint32, C.c_uint32, C.c_uint32,
                                  _u64p, _u32p, _u32p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, _u64p, C.c_void_p]),
    "ss_ntt_gl64": (C.c_int, [C.c_void_p, _vpp, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, C.c_int, C.c_int]),
    "ss_lde_gl64": (C.c_int, [C.c_void_p, _vpp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, _vpp, _vpp]),
    "ss_fri_fold_gl64x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, _u64p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "ss_running_product_gl64x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, _u64p, _u64p, _vpp,
                                            C.c_uint64, C.c_uint64, _u64p]),
    "ss_hash_rows_gl64": (C.c_int, [C.c_void_p, C.c_int, _vpp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p]),
    "ss_gather_rows_gl64": (C.c_int, [C.c_void_p, _vpp, C.c_uint32, C.c_uint32, C.c_uint64, _u64p, C.c_uint32, _u64p]),
    "ss_eval_quotient_gl64x3": (C.c_int, [C.c_void_p, C.c_void_p, _vpp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p]),
    "ss_ood_eval_gl64x3": (C.c_int, [C.c_void_p, _vpp, C.c_uint32, C.c_uint32, _u32p, _u32p, C.c_uint32, _u64p, _u64p]),
    "ss_deep_compose_gl64x3": (C.c_int, [C.c_void_p, _vpp, C.c_uint32, _vpp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, _u32p, _u32p,
                                         C.c_uint32, _u64p, _u64p, _u64p, _u64p, _u64p, _u64p, C.c_void_p]),
    "ss_fri_fold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, _u64p, _u64p, C.c_void_p]),
    "ss_fri_fold_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, _u64p, _u64p, C.c_uint32, C.c_void_p]),
    "ss_fri_fold_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, _u64p, _u64p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]),
    "ss_pow_grind": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_uint32, _u64p]),
    "ss_pedersen_hash": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "ss_pedersen_hash_host": (C.c_int, [_u64p, _u64p, _u64p]),
    "ss_keccak256_host": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p]),
    "ss_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "ss_profile_reset": (C.c_int, [C.c_void_p]),
    "ss_profile_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), _u64p]),
    "ss_upload_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, _u64p]),
    "ss_wait_upload": (C.c_int, [C.c_void_p, C.c_uint64]),
    "ss_profile_read_clock": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ss_fp252_mul_bench": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
}

_lib = None


def load():
    """dlopen the HIP library and bind every declared symbol (no compute)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SandstormHipError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    # If torch is part of this process its bundled HIP runtime must be the one that
    # is loaded (same SONAME libamdhip64.so.7); import it first when it is wanted.
    if os.environ.get("SANDSTORM_WITH_TORCH") == "1" and "torch" not in sys.modules:
        import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    want = header_abi_version()
    if lib.ss_abi_version() != want:
        raise SandstormHipError("%s has ABI version %d, include/sandstorm_hip.h declares %d: rebuild"
                                % (LIB_PATH, lib.ss_abi_version(), want))
    _lib = lib
    return lib


def check(status):
    if status != SS_OK:
        msg = load().ss_last_error()
        raise SandstormHipError("status %d: %s" % (status, msg.decode() if msg else "?"))
