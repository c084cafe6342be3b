"""Trace::build_extension_columns on the device ("next" row X1).

The reference fills the extension trace with sequential host loops between the base-trace commitment and the
extension-trace commitment (layouts/src/recursive/trace.rs:699-814, layouts/src/starknet/trace.rs:997-1100,
"TODO: multithread").  Here the auxiliary columns stay in HBM and the three running permutation products and the
diluted-check aggregate are device scans (ss_permutation_product / ss_diluted_aggregate, csrc/ext.hip), so the
challenges are the only thing that crosses PCIe between the two phases.

`TraceColumns` names the columns the reference's trace object keeps next to the base matrix; the constants are
the layouts' (`layouts/src/{recursive,starknet}/mod.rs`, `air.rs` enums).
"""
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import backend as be
from .coin import canonical

MEMORY_STEP = 2                 # recursive/mod.rs:18, starknet/mod.rs:16
RANGE_CHECK_STEP = 4            # recursive/mod.rs:19, starknet/mod.rs:17
DILUTED_CHECK_STEP = {"recursive": 1, "starknet": 8}        # recursive/mod.rs:20, starknet/mod.rs:18
RC_OFF_DST, RC_ORDERED = 0, 2   # enum RangeCheck (recursive/air.rs:1637-1639, starknet/air.rs:3159-3161)
# challenge indices (air.rs: MemoryPermutation, RangeCheckPermutation, DilutedCheckPermutation, DilutedCheckAggregation)
MEM_Z, MEM_A, RC_Z, DC_Z, AGG_Z, AGG_A = 0, 1, 2, 3, 4, 5


class PermutationCheckError(ValueError):
    """the reference's `assert!((numerator_acc / denominator_acc).is_one())` (trace.rs:734, 757)"""


@dataclass
class TraceColumns:
    """device columns (DeviceBuffer / torch tensor / address) of Montgomery felts"""
    npc: object                             # program-order accesses [a, v, a, v, ...]   (npc_column)
    memory: object                          # address-order accesses [a', v', ...]       (memory_column)
    range_check: object                     # range_check_column
    trace_len: int
    diluted_unordered: Optional[object] = None      # recursive only (diluted_check_unordered_column)
    diluted_ordered: Optional[object] = None        # recursive only (diluted_check_ordered_column)


def _is_one(limbs):
    return canonical(limbs) == 1


def build_extension_columns(layout, ctx, cols: TraceColumns, challenges, check=True, out=None):
    """-> backend.Matrix of the extension columns, resident in HBM (`out`: a Matrix of caller-owned columns to fill).
    recursive: [diluted_check_aggregate (col 7), diluted_check_permutation (col 8), mem_and_rc_permutation (col 9)]
    starknet:  [permutation_column (col 9)]
    check: raise PermutationCheckError where the reference asserts that a product closes to one."""
    n = cols.trace_len
    z_mem, a_mem, z_rc, z_dc = challenges[MEM_Z], challenges[MEM_A], challenges[RC_Z], challenges[DC_Z]
    z_agg, a_agg = challenges[AGG_Z], challenges[AGG_A]
    n_mem, n_rc = n // MEMORY_STEP, n // RANGE_CHECK_STEP
    step = DILUTED_CHECK_STEP[layout]
    if layout == "recursive":
        out = out or be.Matrix.empty(ctx, 3, n)
        agg, dperm, mem_rc = out.cols
        for c in out.cols:
            ctx.zero(c)
        # Permutation::col_and_shift: Memory (9, 0), RangeCheck (9, 1), DilutedCheck (8, 0)  (recursive/air.rs:1705-1711)
        ctx.permutation_product((cols.npc, MEMORY_STEP, 0, 1), (cols.memory, MEMORY_STEP, 0, 1), n_mem, z_mem, a_mem,
                                mem_rc, MEMORY_STEP, 0, want_last=False)
        last_rc = ctx.permutation_product((cols.range_check, RANGE_CHECK_STEP, RC_OFF_DST, -1),
                                          (cols.range_check, RANGE_CHECK_STEP, RC_ORDERED, -1), n_rc, z_rc, None,
                                          mem_rc, RANGE_CHECK_STEP, 1)
        last_dc = ctx.permutation_product((cols.diluted_unordered, 1, 0, -1), (cols.diluted_ordered, 1, 0, -1), n, z_dc, None, dperm)
        ctx.diluted_aggregate(cols.diluted_ordered, 1, 0, n, z_agg, a_agg, agg)
    elif layout == "starknet":
        out = out or be.Matrix.empty(ctx, 1, n)
        perm = out.cols[0]
        ctx.zero(perm)
        # enum Permutation {Memory = 0, RangeCheck = 1, DilutedCheck = 7}, DilutedCheck {Unordered = 1, Ordered = 5,
        # Aggregate = 3} (starknet/air.rs:3137-3141, 3220-3225): all four live in the one permutation column
        ctx.permutation_product((cols.npc, MEMORY_STEP, 0, 1), (cols.memory, MEMORY_STEP, 0, 1), n_mem, z_mem, a_mem,
                                perm, MEMORY_STEP, 0, want_last=False)
        last_rc = ctx.permutation_product((cols.range_check, RANGE_CHECK_STEP, RC_OFF_DST, -1),
                                          (cols.range_check, RANGE_CHECK_STEP, RC_ORDERED, -1), n_rc, z_rc, None,
                                          perm, RANGE_CHECK_STEP, 1)
        last_dc = ctx.permutation_product((cols.range_check, step, 1, -1), (cols.range_check, step, 5, -1), n // step, z_dc, None,
                                          perm, step, 7)
        ctx.diluted_aggregate(cols.range_check, step, 5, n // step, z_agg, a_agg, perm, step, 3)
    else:
        raise ValueError("unknown layout %r" % layout)
    if check:
        if not _is_one(last_rc):
            raise PermutationCheckError("range-check permutation product does not close to one")
        if not _is_one(last_dc):
            raise PermutationCheckError("diluted-check permutation product does not close to one")
    return out
