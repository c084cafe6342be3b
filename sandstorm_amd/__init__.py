"""sandstorm_amd — MI355X (gfx950) backend for the proving hot path of
andrewmilson/sandstorm: Fp252 NTT/LDE, row/leaf/node hashing, Merkle trees
(Keccak, Blake2s, Pedersen), constraint-quotient evaluation, DEEP, coset FRI.

The product is the C-ABI shared library built from csrc/ (include/sandstorm_hip.h);
this package is the thin host-side mirror of the reference's interface used by
tests and bench.py.  Importing it never touches the oracle under oracle/.
"""
from ._lib import SandstormHipError, load  # noqa: F401

__all__ = ["SandstormHipError", "load"]
