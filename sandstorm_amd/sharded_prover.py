"""One proof over several GPUs of a node - one process per GPU, torch.distributed for the exchanges
("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference has nothing to match here: its parallelism is rayon loops over rows and columns inside one process
(crypto/src/merkle/utils.rs:30-32, layouts/src/starknet/trace.rs:182).  The proof is the single-device proof, byte for
byte (tests/test_sharded.py runs this driver on 2 and 4 gloo ranks against prover.Prover).

Distribution of one proof's data over R ranks (R a power of two), per stage of prover.Prover.prove:

  stage                               distribution                          exchange that leads to it
  ----------------------------------  ------------------------------------  ----------------------------------------------
  LDE of trace columns (N1/N2)        by COLUMN: column c on rank c % R     none - the trace generator fills owned columns
  row hashing (H1), constraint        by ROW BLOCK: rank r holds LDE rows   point-to-point re-shard: every column owner
  evaluation (Q1), DEEP (D1)          [r N/R, (r+1) N/R) of every column,   sends each rank its block (+ the halo, which
                                      plus the `halo` rows behind them      wraps around the domain); R (R-1) messages per
                                      that the constraints reach            matrix, all links busy at once (no ring)
  Merkle sub-trees (H3/H4)            by LEAF BLOCK: rank r owns leaves     row digests (32 B per row) go from the rank
                                      [r N/R, (r+1) N/R) = a contiguous     that hashed the row to the rank that owns its
                                      sub-tree; leaf i is row bitrev(i)     leaf; then an all-gather of R sub-tree roots,
                                                                            the log2 R top levels on every host
  composition polynomial (Q2)         one vector: rank 0 interpolates,      gather of the R row blocks of the composition
                                      its 2 column LDEs on ranks 0 and 1    evaluations to rank 0, one column to rank 1
  DEEP polynomial, FRI (F1), PoW      rank 0: the DEEP polynomial has       gather of the n / R sub-coset values per rank
                                      degree < n; every rank composes it    (half a column in total) to rank 0, which
                                      on its part of the trace-size         interpolates, re-expands and runs FRI as the
                                      sub-coset (ss_deep_compose_rows)      single-device prover does
  query openings                      rows on their row-block rank, paths   broadcast of the positions, gather of the
                                      on their leaf-block rank              opened rows / paths to rank 0

Streams: the tensors the ranks exchange are torch tensors and the kernels run behind the C ABI, so both must be on ONE HIP
stream - make a torch.cuda.Stream current and hand its handle to backend.Context (torch's default stream has handle 0, which
ss_ctx_set_stream takes to mean "the context's own stream").

Nothing is a sum over ranks: there is no all-reduce.  The Fiat-Shamir coin runs on every rank in lock step up to the
out-of-domain evaluations (every rank sees the same roots and values); from DEEP on only rank 0 holds the transcript.
"""
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import numpy as np

from . import backend as be
from . import sharding
from .coin import PublicCoin, canonical
from .prover import (Claim, Conventions, Proof, ProofOptions, _log2, _pow_limbs, bitrev, fri_commit_phase, fri_open,
                     proof_of_work)


def _torch():
    import torch
    return torch


class Comm:
    """the process group, reduced to what the prover needs"""

    def __init__(self, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device                        # torch device of the exchanged tensors
        self.stage_through_host = dist.get_backend() == "gloo" and device is not None and _torch().device(device).type == "cuda"
        if self.world > 1 and dist.get_backend() == "nccl":
            # RCCL builds its communicators at the first collective: outside any timed region
            dist.all_reduce(_torch().zeros(1, device=device))

    def exchange(self, sends, recvs):
        """sends / recvs: [(peer, tensor)] in a fixed, globally agreed order per pair of ranks.  ONE all-to-all with uneven
        splits that EVERY rank enters, whatever it has to send or receive (a rank with nothing for a peer contributes a
        zero-length split): over RCCL an all-to-all is a grouped send/recv per pair - on xGMI every pair of GPUs has its own
        link, so the R (R-1) transfers run concurrently, nothing is a ring - and a collective all ranks enter cannot
        dead-lock the way one-sided point-to-point batches can (rank 0 only receiving while the others only send)."""
        torch, dist, R = _torch(), self.dist, self.world
        if R == 1:
            assert not sends and not recvs
            return

        def words(t):                               # every exchanged buffer is a whole number of 8-byte words (felts, digests)
            assert t.is_contiguous() and (t.numel() * t.element_size()) % 8 == 0
            return t.view(torch.int64).reshape(-1) if t.dtype != torch.int64 else t.reshape(-1)
        out_parts, in_views = [[] for _ in range(R)], [[] for _ in range(R)]
        for p, t in sends:
            out_parts[p].append(words(t))
        for p, t in recvs:
            in_views[p].append(words(t))
        in_splits = [sum(v.numel() for v in out_parts[p]) for p in range(R)]        # what this rank puts in, per destination
        out_splits = [sum(v.numel() for v in in_views[p]) for p in range(R)]        # what it gets out, per source
        dev = "cpu" if self.stage_through_host else self.device     # gloo moves host memory only (ranks sharing one GPU in the tests)
        flat_in = [v.to(dev) for p in range(R) for v in out_parts[p]]
        src = torch.cat(flat_in) if flat_in else torch.zeros(0, dtype=torch.int64, device=dev)
        dst = torch.zeros(sum(out_splits), dtype=torch.int64, device=dev)
        dist.all_to_all_single(dst, src, out_splits, in_splits)
        o = 0
        for p in range(R):
            for v in in_views[p]:
                v.copy_(dst[o:o + v.numel()])
                o += v.numel()

    def all_to_all_equal(self, dst, src):
        """equal splits: chunk p of `src` (first dimension cut in R) goes to rank p, chunk s of `dst` comes from rank s"""
        if self.stage_through_host:
            h = _torch().zeros_like(src, device="cpu")
            self.dist.all_to_all_single(h, src.cpu())
            dst.copy_(h)
        else:
            self.dist.all_to_all_single(dst, src)

    def all_gather_object(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def gather_object(self, obj, dst=0):
        """(an all-gather underneath: gather_object is not available on every backend, and the objects are small)"""
        out = self.all_gather_object(obj)
        return out if self.rank == dst else None

    def broadcast_object(self, obj, src=0):
        box = [obj]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]


def _dbg(comm, what, t):
    """SS_SHARD_DEBUG=1: fingerprint of an intermediate on every rank (to diff a run against another backend's)"""
    import os
    if os.environ.get("SS_SHARD_DEBUG"):
        import hashlib
        b = t if isinstance(t, (bytes, bytearray)) else np.ascontiguousarray(t.cpu().numpy()).tobytes()
        print("[shard dbg] rank %d %-28s %s" % (comm.rank, what, hashlib.sha256(b).hexdigest()[:16]), flush=True)


@dataclass
class _Commitment:
    """a matrix committed over R ranks: this rank's leaf block and sub-tree, and the replicated top levels"""
    leaves: object                  # tensor [B, 32] u8 (row digests) or [B, 4] i64 (single column: the elements)
    leaf_kind: int
    nodes: object                   # sub-tree heap [2B, 32]
    tags: object
    top: list                       # top levels, root first: [[(digest, tag)] per level] down to the R sub-tree roots
    root: bytes
    root_tag: int


class ShardedProver:
    """prover.Prover.prove for one proof over the ranks of a process group"""

    def __init__(self, ctx, claim: Claim, comm: Comm, options: ProofOptions = None, conventions: Conventions = None):
        self.ctx, self.claim, self.comm = ctx, claim, comm
        self.options = options or ProofOptions()
        self.conv = conventions or Conventions()
        self.pow_nonce = None
        R = comm.world
        assert R & (R - 1) == 0, "the row blocks need a power-of-two number of ranks"

    # ---- buffers: torch tensors (what torch.distributed moves); the C ABI takes their data_ptr
    def felts(self, rows):
        """a column of `rows` field elements.  Every caller hands it to a kernel or an exchange that writes all of it: no fill (a zero
        fill of the ~21 GB a 2^20-step proof allocates is ~7 ms of the GPU's time per proof)"""
        return _torch().empty((rows, 4), dtype=_torch().int64, device=self.comm.device)

    def owner(self, col):
        return col % self.comm.world

    # ---- column blocks -> row blocks (+ halo)
    def to_row_blocks(self, owned: Dict[int, object], ncols, first_col, N, halo):
        """owned: {column index: tensor [N, 4]} of the columns this rank extended (indices first_col .. first_col + ncols).
        -> [tensor [B + halo, 4]] for ALL ncols columns: LDE rows (r B + k) mod N, k < B + halo, of each."""
        torch, comm = _torch(), self.comm
        R, r = comm.world, comm.rank
        B = N // R

        def block_of(col_tensor, p):
            lo, hi = p * B, p * B + B + halo
            if hi <= N:
                return col_tensor[lo:hi]
            return torch.cat([col_tensor[lo:], col_tensor[:hi - N]])
        out, sends, recvs = {}, [], []
        for c in range(first_col, first_col + ncols):
            o = self.owner(c)
            if o == r:
                for p in range(R):
                    blk = block_of(owned[c], p).contiguous()
                    if p == r:
                        # not a view: the whole column is released after the re-shard (a group of one keeps the column itself)
                        out[c] = blk.clone() if R > 1 else blk
                    else:
                        sends.append((p, blk))
            else:
                out[c] = self.felts(B + halo)
                recvs.append((o, out[c]))
        comm.exchange(sends, recvs)
        for c in range(first_col, first_col + ncols):
            _dbg(comm, "row block of column %d" % c, out[c])
        return [out[c] for c in range(first_col, first_col + ncols)]

    # ---- commitment of a row-block matrix
    def commit(self, blocks, N, order) -> _Commitment:
        torch, comm, ctx = _torch(), self.comm, self.ctx
        R, r = comm.world, comm.rank
        B, log_N, log_R = N // R, _log2(N), _log2(comm.world)
        Tree = self.claim.tree
        single = len(blocks) == 1
        log_B = log_N - log_R
        # Row i = r B + k of the matrix is leaf bitrev(i) (or i).  Bit-reversed: bitrev_{log N}(i) = bitrev_{log B}(k) << log R
        # | bitrev_{log R}(r), so with this rank's digests stored at their LOCAL bit-reversed slot bitrev_{log B}(k) - which is
        # what the row-hash kernel's scatter does for free - the B / R slots of chunk p are exactly the leaves rank p owns:
        # one equal-split all-to-all (32 B per row), after which the chunk from rank s is the stride-R comb at offset
        # bitrev_{log R}(s) of the leaf block.  No index tensors, no gather on either side.
        if single:                                  # raw-element leaves (merkle/mod.rs:113-117)
            mine = blocks[0][:B]
            if order == be.BITREV:                  # (a torch index gather of 2^25 elements is 2.4 ms; the library's permutation 0.5)
                src, mine = mine.contiguous(), self.felts(B)
                ctx.bitrev_permute32(src, log_B, mine)
        else:
            mine = torch.empty((B, 32), dtype=torch.uint8, device=comm.device)
            ctx.hash_rows(Tree.row_hash, blocks, B, mine, order)            # the blocks' halo is not hashed
        _dbg(comm, "row digests / leaves (local order)", mine)
        if R == 1 or order != be.BITREV:            # natural order: a rank's rows are its leaves
            leaves = mine
        else:
            got = torch.empty_like(mine)
            comm.all_to_all_equal(got, mine.contiguous())
            W = mine.shape[1]
            leaves = torch.empty_like(mine)
            comb = leaves.view(B // R, R, W)
            for src in range(R):
                comb[:, bitrev(src, log_R)] = got[src * (B // R):(src + 1) * (B // R)]
        _dbg(comm, "leaf block", leaves)
        # this rank's sub-tree: its root sits at depth log2 R of the whole tree
        nodes = torch.zeros((2 * B, 32), dtype=torch.uint8, device=comm.device)
        tags = torch.zeros(2 * B, dtype=torch.uint8, device=comm.device) if Tree.tree_kind == be.TREE_FRIENDLY else None
        leaf_kind = be.LEAF_FELT if single else be.LEAF_DIGEST
        sub_root, sub_tag = ctx.merkle_build(Tree.tree_kind, sharding.subtree_friendly_layers(Tree.n_friendly, R), leaf_kind,
                                             leaves, B, nodes, tags, be.NATURAL)
        _dbg(comm, "sub-tree root", sub_root)
        roots = comm.all_gather_object((sub_root, sub_tag))
        # the top log2 R levels on the host (<= 7 hashes)
        top = [list(roots)]
        depth = log_R - 1
        while len(top[0]) > 1:
            lvl = top[0]
            if single and Tree.tree_kind == be.TREE_FRIENDLY:
                nf = 1 << 30                        # a single-column friendly tree is Pedersen at every level (mod.rs:113-117)
            else:
                nf = Tree.n_friendly
            top.insert(0, [sharding.merge_nodes(Tree.tree_kind, nf, depth, lvl[2 * q], lvl[2 * q + 1]) for q in range(len(lvl) // 2)])
            depth -= 1
        return _Commitment(leaves, leaf_kind, nodes, tags, top, top[0][0][0], top[0][0][1])

    def open(self, com: _Commitment, blocks, N, positions, order):
        """-> on rank 0: (rows [nq, ncols, 4], paths [nq, log N, 32], leaf digests [nq, 32] or None, MixedMerkleDigest tags
        [nq, log N] of the path entries (FriendlyMerkleTree; None otherwise)); None elsewhere"""
        comm, ctx = self.comm, self.ctx
        R, r = comm.world, comm.rank
        B, log_N, log_R = N // R, _log2(N), _log2(comm.world)
        log_B = log_N - log_R
        nat = [bitrev(p, log_N) for p in positions] if order == be.BITREV else list(positions)
        my_rows = [(q, i % B) for q, i in enumerate(nat) if i // B == r]
        my_leaves = [(q, p % B) for q, p in enumerate(positions) if p // B == r]
        part = {"rows": {}, "paths": {}, "digests": {}, "tags": {}}
        if my_rows:
            got = ctx.gather_rows(blocks, [k for _, k in my_rows])
            for (q, _), row in zip(my_rows, got):
                part["rows"][q] = row
        if my_leaves:
            paths, ptags = ctx.merkle_open(com.nodes, com.tags, B, [k for _, k in my_leaves])
            lv = None
            if com.leaf_kind == be.LEAF_DIGEST:         # only the opened leaves cross to the host
                idx = _torch().tensor([k for _, k in my_leaves], dtype=_torch().int64, device=com.leaves.device)
                lv = com.leaves[idx].cpu().numpy()
            for t, ((q, k), path) in enumerate(zip(my_leaves, paths)):
                part["paths"][q] = path
                part["tags"][q] = ptags[t]
                if lv is not None:
                    part["digests"][q] = lv[t].copy()
        parts = comm.gather_object(part, 0)
        if r != 0:
            return None
        nq = len(positions)
        rows = np.zeros((nq, len(blocks), 4), dtype=np.uint64)
        paths = np.zeros((nq, log_N, 32), dtype=np.uint8)
        digests = np.zeros((nq, 32), dtype=np.uint8) if com.leaf_kind == be.LEAF_DIGEST else None
        tags = np.zeros((nq, log_N), dtype=np.uint8) if com.tags is not None else None
        for prt in parts:
            for q, row in prt["rows"].items():
                rows[q] = row
            for q, path in prt["paths"].items():
                paths[q, :log_B] = path
            for q, d in prt["digests"].items():
                digests[q] = d
            if tags is not None:
                for q, tg in prt["tags"].items():
                    tags[q, :log_B] = tg
        for q, p in enumerate(positions):           # the top levels: siblings of the sub-tree root's ancestors
            node = p // B
            for lvl in range(log_R):
                sib = com.top[log_R - lvl][node ^ 1]
                paths[q, log_B + lvl] = np.frombuffer(sib[0], dtype=np.uint8)
                if tags is not None:
                    tags[q, log_B + lvl] = sib[1]
                node >>= 1
        return rows, paths, digests, tags

    # ---- the proof
    def prove(self, coin_seed: bytes, my_base: Dict[int, object], build_extension: Callable[[List[np.ndarray]], Dict[int, object]],
              n: int) -> Optional[Proof]:
        """my_base: {column index: tensor [n, 4]} of the base columns this rank owns (column c on rank c % R);
        build_extension(challenges) -> {column index: tensor [n, 4]} of the extension columns this rank owns (global column
        numbers, base columns first).  -> the Proof on rank 0, None elsewhere."""
        torch, comm, ctx = _torch(), self.comm, self.ctx
        opt, conv, air = self.options, self.conv, self.claim.air
        R, r = comm.world, comm.rank
        log_n, lb = _log2(n), _log2(opt.lde_blowup_factor)
        log_N, N = log_n + lb, n << lb
        B = N // R
        assert N % R == 0 and (n // R) >= 1
        g = be.felt(conv.lde_offset)
        order = be.BITREV if conv.bitrev_commit else be.NATURAL
        nb, ne = air.num_base_columns, air.num_extension_columns
        # rows behind a block that its constraints reach (wrap-around included); never more than the rest of the domain
        halo = max((o for _, o in air.mask), default=0) << lb
        assert R == 1 or halo <= N - B, "the constraints reach further than the rest of the domain: fewer ranks for a trace this short"
        halo = min(halo, N - B)
        coin = PublicCoin(self.claim.coin_kind, coin_seed)
        proof = Proof(opt, n, tree_kind=self.claim.tree.tree_kind)
        import os
        import time
        timing = os.environ.get("SS_SHARD_TIMING") and r == 0
        t_last = [time.perf_counter()]

        def mark(stage):
            """SS_SHARD_TIMING=1: wall clock per stage on rank 0 (synchronises the stream: diagnosis only)"""
            if timing:
                ctx.sync()
                now = time.perf_counter()
                print("[shard timing] %-34s %9.3f ms" % (stage, 1e3 * (now - t_last[0])), flush=True)
                t_last[0] = now

        def extend(owned):
            """LDE of this rank's columns -> ({col: evaluations [N, 4]}, {col: bit-reversed coefficients [n, 4]})"""
            cols = sorted(owned)
            ev = {c: self.felts(N) for c in cols}
            co = {c: self.felts(n) for c in cols}
            if cols:
                ctx.lde([owned[c] for c in cols], log_n, lb, g, [ev[c] for c in cols], [co[c] for c in cols])
            return ev, co

        # 2. base trace
        assert sorted(my_base) == [c for c in range(nb) if self.owner(c) == r], "column c lives on rank c % R"
        base_ev, coeffs = extend(my_base)
        mark("base lde")
        base_blocks = self.to_row_blocks(base_ev, nb, 0, N, halo)
        del base_ev
        mark("base re-shard")
        base_com = self.commit(base_blocks, N, order)
        mark("base commit")
        proof.base_root = base_com.root
        coin.reseed_with_digest(proof.base_root)
        # 3-4. challenges -> extension trace
        challenges = [coin.draw() for _ in range(air.num_challenges)]
        proof.challenges = challenges
        blocks, ext_blocks, ext_com = list(base_blocks), [], None
        if ne:
            my_ext = build_extension(challenges)
            assert sorted(my_ext) == [c for c in range(nb, nb + ne) if self.owner(c) == r]
            ext_ev, ext_co = extend(my_ext)
            coeffs.update(ext_co)
            ext_blocks = self.to_row_blocks(ext_ev, ne, nb, N, halo)
            del ext_ev
            mark("extension build + lde + re-shard")
            ext_com = self.commit(ext_blocks, N, order)
            mark("extension commit")
            proof.extension_root = ext_com.root
            coin.reseed_with_digest(proof.extension_root)
            blocks += ext_blocks
        # 5. the composition constraint on this rank's rows; interpolation on rank 0; the two column LDEs on ranks 0 and 1 % R
        comp_coeff = coin.draw()
        proof.composition_coeff = comp_coeff
        program, tables, table_desc = air.build_program(n, challenges, comp_coeff)
        d_tables = tables if tables is None or hasattr(tables, "ptr") else (ctx.column(tables) if len(tables) else None)
        mark("program")
        q_block = self.felts(B)
        if R == 1:
            ctx.eval_quotient(program, d_tables, table_desc, blocks, log_n, lb, g, q_block)
        else:
            ctx.eval_quotient_rows(program, d_tables, table_desc, blocks, log_n, lb, g, r * B, B, B + halo, q_block)
        mark("quotient")
        ncomp = conv.composition_columns
        assert ncomp == 1 << lb == 2, "composition split implemented for blowup 2"
        comp_owned, comp_co = {}, {}
        # gather of the R row blocks to rank 0 (every rank enters the exchange), the inverse transform there, then column
        # k's coefficients to rank k % R (again one exchange that every rank enters)
        comp_evals = None
        if R == 1:
            comp_evals = q_block
        elif r == 0:
            comp_evals = self.felts(N)
            comp_evals[:B] = q_block
            comm.exchange([], [(p, comp_evals[p * B:(p + 1) * B]) for p in range(1, R)])
        else:
            comm.exchange([(0, q_block)], [])
        if r == 0:
            ctx.ntt([comp_evals], log_N, be.INVERSE, g, be.NATURAL, be.BITREV)   # H0 | H1, each bit-reversed: the split is free
        sends, recvs = [], []
        for k in range(ncomp):
            o = self.owner(k)
            if r == 0:
                half = comp_evals[k * n:(k + 1) * n]
                if o == 0:
                    comp_co[k] = half
                else:
                    sends.append((o, half))
            elif o == r:
                comp_co[k] = self.felts(n)
                recvs.append((0, comp_co[k]))
        comm.exchange(sends, recvs)
        for k, co in comp_co.items():
            comp_owned[k] = self.felts(N)
            ctx.evaluate([co], log_n, lb, g, [comp_owned[k]])
        mark("composition gather + ntt + lde")
        comp_blocks = self.to_row_blocks(comp_owned, ncomp, 0, N, 0)
        del comp_owned
        comp_com = self.commit(comp_blocks, N, order)
        mark("composition re-shard + commit")
        proof.composition_root = comp_com.root
        coin.reseed_with_digest(proof.composition_root)
        # 6. out-of-domain point: every column owner evaluates its cells, everybody learns all of them
        z = coin.draw()
        proof.z = z
        mine = [(j, c, o) for j, (c, o) in enumerate(air.mask) if self.owner(c) == r]
        part = {}
        if mine:
            cols = sorted({c for _, c, _ in mine})
            vals = ctx.ood_eval([coeffs[c] for c in cols], log_n, [cols.index(c) for _, c, _ in mine], [o for _, _, o in mine], z)
            part = {j: vals[t] for t, (j, _, _) in enumerate(mine)}
        zc = be.felt(pow(canonical(z), ncomp, be.P))
        part_c = {k: ctx.poly_eval([co], log_n, zc)[0] for k, co in comp_co.items()}
        gathered = comm.all_gather_object((part, part_c))
        ood_t, ood_c = {}, {}
        for pt, pc in gathered:
            ood_t.update(pt)
            ood_c.update(pc)
        proof.ood_trace = np.stack([ood_t[j] for j in range(len(air.mask))])
        proof.ood_composition = np.stack([ood_c[k] for k in range(ncomp)])
        coin.reseed_with_field_elements(list(proof.ood_trace) + list(proof.ood_composition))
        mark("ood")
        # 7. DEEP composition on this rank's part of the trace-size sub-coset; rank 0 interpolates and re-expands
        deep_alpha = coin.draw()
        proof.deep_alpha = deep_alpha
        dcoef = _pow_limbs(deep_alpha, len(air.mask) + ncomp)
        mask_col, mask_off = [c for c, _ in air.mask], [o for _, o in air.mask]
        cnt = n // R
        sub_block = self.felts(cnt)
        ctx.deep_compose_rows(blocks, comp_blocks, log_n, lb, g, mask_col, mask_off, proof.ood_trace, dcoef[:len(air.mask)],
                              proof.ood_composition, dcoef[len(air.mask):], z, r * cnt, cnt, sub_block)
        if r != 0:
            comm.exchange([(0, sub_block)], [])
            positions = comm.broadcast_object(None, 0)
        else:
            if R == 1:
                sub = sub_block
            else:
                sub = self.felts(n)
                sub[:cnt] = sub_block
                comm.exchange([], [(p, sub[p * cnt:(p + 1) * cnt]) for p in range(1, R)])
            mark("deep rows + gather")
            deep = _TensorBuffer(ctx, self.felts(N))
            ctx.deep_extend(sub, log_n, lb, g, deep)
            mark("deep extend")
            # 8-9. FRI, proof of work, query positions: on rank 0, as the single-device prover does them
            layers = fri_commit_phase(ctx, self.claim.tree, conv, opt, coin, proof, deep, log_N, n)
            proof.pow_nonce = proof_of_work(ctx, self.claim.coin_kind, coin, opt, self.pow_nonce)
            coin.reseed_with_int(proof.pow_nonce)
            positions = coin.draw_queries(opt.num_queries, N)
            proof.query_positions = positions
            mark("fri + pow")
            comm.broadcast_object(positions, 0)
        # openings of the three trace commitments: rows from the row-block ranks, paths from the leaf-block ranks
        opened = [self.open(base_com, base_blocks, N, positions, order),
                  self.open(ext_com, ext_blocks, N, positions, order) if ne else None,
                  self.open(comp_com, comp_blocks, N, positions, order)]
        if r != 0:
            return None
        proof.base_rows, proof.base_paths, proof.base_leaf_digests, proof.base_path_tags = opened[0]
        if ne:
            proof.extension_rows, proof.extension_paths, proof.extension_leaf_digests, proof.extension_path_tags = opened[1]
        proof.composition_rows, proof.composition_paths, proof.composition_leaf_digests, proof.composition_path_tags = opened[2]
        proof.root_tags = [base_com.root_tag, ext_com.root_tag if ne else 0, comp_com.root_tag]
        fri_open(ctx, conv, opt, proof, layers, positions)
        mark("openings")
        return proof


class _TensorBuffer:
    """a torch tensor where the single-device code expects a backend.DeviceBuffer (ptr / nbytes / ctx / download)"""

    def __init__(self, ctx, t):
        self.ctx, self.t = ctx, t
        self.ptr, self.nbytes = t.data_ptr(), t.numel() * t.element_size()

    def data_ptr(self):
        return self.ptr

    def download(self, dtype, shape):
        return np.ascontiguousarray(self.t.cpu().numpy()).view(dtype).reshape(shape).copy()
