"""AirPublicInput -> public-coin seed.

`sandstorm-cli prove` seeds the Fiat-Shamir coin with a hash of the public input laid out the way StarkWare's
verifiers expect it: `CairoAuxInput::public_input_elements` (src/input.rs:10-150) and `CairoPublicCoin::
from_public_input` (src/lib.rs:145-167).  `AirPublicInput` is the `air-public-input.json` that `cairo-run` writes
(binary/src/lib.rs:296-340).  Host-only; the Pedersen page hash uses the library's host Pedersen.
"""
import json
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from . import backend as be
from .coin import blake2s256, canonical, keccak256

P = be.P
# Layout::sharp_code (binary/src/lib.rs:92-102): the layout name as a big-endian integer
SHARP_CODE = {"starknet": int.from_bytes(b"starknet", "big"), "recursive": int.from_bytes(b"recursive", "big")}
SEGMENTS = ("program", "execution", "output", "pedersen", "range_check", "ecdsa", "bitwise", "ec_op", "poseidon")


@dataclass
class AirPublicInput:
    """binary/src/lib.rs:308-318"""
    layout: str
    rc_min: int
    rc_max: int
    n_steps: int
    memory_segments: Dict[str, Optional[Tuple[int, int]]]           # name -> (begin_addr, stop_ptr) or None
    public_memory: List[Tuple[int, int]] = field(default_factory=list)   # (address, value)

    @classmethod
    def from_json(cls, src):
        j = json.load(open(src)) if isinstance(src, str) else src
        segs = {}
        for name in SEGMENTS:
            s = j["memory_segments"].get(name)
            segs[name] = (int(s["begin_addr"]), int(s["stop_ptr"])) if s else None
        if segs["program"] is None or segs["execution"] is None:
            raise ValueError("program and execution segments are mandatory")
        mem = [(int(e["address"]), int(e["value"], 16) % P) for e in j["public_memory"]]
        return cls(j["layout"], int(j["rc_min"]), int(j["rc_max"]), int(j["n_steps"]), segs, mem)

    def public_memory_padding(self):
        """binary/src/lib.rs:337-339: the entry at address 1"""
        for e in self.public_memory:
            if e[0] == 1:
                return e
        raise ValueError("public memory has no entry at address 1")


def _segment(pi, name, which):
    s = pi.memory_segments.get(name)
    if s is None:                                   # `Option::unwrap` on a missing segment (src/input.rs:47)
        raise ValueError("the %s layout needs the %s segment" % (pi.layout, name))
    return s[which]


def base_values(pi):
    """src/input.rs:10-48"""
    if pi.n_steps <= 0:
        raise ValueError("n_steps must be positive")
    vals = [pi.n_steps.bit_length() - 1, pi.rc_min, pi.rc_max, SHARP_CODE[pi.layout]]
    for name in ("program", "execution", "output", "pedersen", "range_check"):
        vals += [_segment(pi, name, 0), _segment(pi, name, 1)]
    return vals


def layout_specific_values(pi):
    """src/input.rs:50-116"""
    pad_addr, pad_value = pi.public_memory_padding()
    if pi.layout == "starknet":
        vals = []
        for name in ("ecdsa", "bitwise", "ec_op", "poseidon"):
            vals += [_segment(pi, name, 0), _segment(pi, name, 1)]
    elif pi.layout == "recursive":
        vals = [_segment(pi, "bitwise", 0), _segment(pi, "bitwise", 1)]
    else:
        raise NotImplementedError("layout %r (the reference: unimplemented!())" % pi.layout)
    return vals + [pad_addr, pad_value, 1]           # one public memory page


def _page_hash(pi, coin_kind):
    """H::hash_elements over address, value, address, value, ... as 32 big-endian bytes
    (CanonicalKeccak256HashFn: crypto/src/hash/keccak.rs:124-134; PedersenHashFn: hash/pedersen.rs:67-76)"""
    if coin_kind == be.COIN_SOLIDITY:
        return keccak256(b"".join(int(v).to_bytes(32, "big") for e in pi.public_memory for v in e))
    cur = be.felt(0)
    n = 0
    for e in pi.public_memory:
        for v in e:
            cur = be.pedersen_hash_host(cur, be.felt(v))
            n += 1
    return canonical(be.pedersen_hash_host(cur, be.felt(n))).to_bytes(32, "big")


def memory_page_values(pi, coin_kind):
    """src/input.rs:118-139: [page size, page hash] of the main page"""
    return [len(pi.public_memory), int.from_bytes(_page_hash(pi, coin_kind), "big")]


def public_input_elements(pi, coin_kind):
    """src/input.rs:141-149 (256-bit integers)"""
    return base_values(pi) + layout_specific_values(pi) + memory_page_values(pi, coin_kind)


def public_coin_seed(pi, coin_kind) -> bytes:
    """CairoPublicCoin::from_public_input (src/lib.rs:145-167): the coin's initial digest"""
    seed = b"".join(int(v).to_bytes(32, "big") for v in public_input_elements(pi, coin_kind))
    return keccak256(seed) if coin_kind == be.COIN_SOLIDITY else blake2s256(seed)
