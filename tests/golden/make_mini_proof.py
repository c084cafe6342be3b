"""Generates tests/golden/mini_proof_eth_log{5,9}.bin: proofs of the mini AIR (tests/mini_air.py) produced by the C++
host on an MI355X and serialised in the reference's wire format (ssh_prove_wire).  The CPU suite verifies them with
sandstorm_amd/verifier.py (tests/test_verifier.py), so the wire format, the conventions and the verifier are
exercised without a GPU.  Run on the GPU box:  python tests/golden/make_mini_proof.py  (writes gpurun_out/)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle_py as oracle                      # noqa: E402  (only its to_mont helper)
from sandstorm_amd import backend as be, hostlib            # noqa: E402
from sandstorm_amd.coin import canonical                    # noqa: E402
from sandstorm_amd.prover import ProofOptions               # noqa: E402
from tests import mini_air                                  # noqa: E402

out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
ctx = be.Context(0)
seed = bytes(range(32))
opt = ProofOptions(num_queries=12, grinding_factor=8, fri_max_remainder_coeffs=4)
meta = {"seed_hex": seed.hex(), "options": [opt.num_queries, opt.lde_blowup_factor, opt.grinding_factor, opt.fri_folding_factor,
                                            opt.fri_max_remainder_coeffs],
        "claim": "EthVerifierClaim flavour: LeafVariantMerkleTree<MaskedKeccak256HashFn<20>>, SolidityVerifierPublicCoin", "files": {}}
# the third proof has NO FRI layer (the trace fits the remainder bound): the DEEP evaluations go straight into the remainder
for log_n, opt, name in ((5, opt, "mini_proof_eth_log5.bin"), (9, opt, "mini_proof_eth_log9.bin"),
                         (5, ProofOptions(num_queries=12, grinding_factor=8, fri_max_remainder_coeffs=32), "mini_proof_eth_log5_nolayers.bin")):
    n = 1 << log_n
    c0, c1 = mini_air.base_trace(n)
    base = be.Matrix.from_host(ctx, [oracle.to_mont(c0), oracle.to_mont(c1)])
    keep = []

    def build_extension(challenges):
        e0 = mini_air.extension_trace(c0, canonical(challenges[0]))
        m = be.Matrix.from_host(ctx, [oracle.to_mont(e0)])
        keep.append(m)
        return m.cols
    air = hostlib.HostAir(ctx, hostlib.AIR_MINI, log_n)
    raw = hostlib.prove(ctx, air, be.TREE_KECCAK_M20, 0, be.COIN_SOLIDITY, seed, base.cols, log_n, build_extension, opt, wire=True)
    air.close()
    with open(os.path.join(out_dir, name), "wb") as f:
        f.write(raw)
    meta["files"][name] = {"trace_len": n, "bytes": len(raw)}
    print(name, len(raw))
with open(os.path.join(out_dir, "mini_proof_meta.json"), "w") as f:
    json.dump(meta, f, indent=1)
