"""No GPU: the C++ host library (libsandstorm_host.so) — its Fiat-Shamir coin against the
reference KATs and the oracle coin."""
import numpy as np

from tests.util import random_column


def test_cpp_coin_kats_and_oracle(oracle, golden):
    from sandstorm_amd import hostlib
    g = golden("coins.json")
    c = hostlib.HostCoin(0, bytes(32))
    for want in g["solidity_zero_seed_draws"]:
        assert int(oracle.from_mont(c.draw())) == int(want)
    k = g["cairo_reseed"]
    c = hostlib.HostCoin(1, bytes.fromhex(k["seed"]))
    c.reseed_bytes(int(k["element"]).to_bytes(32, "big"))
    assert c.state[0].hex() == k["digest"]
    for kind in (0, 1):
        a, b = hostlib.HostCoin(kind, bytes(range(32))), oracle.Coin(kind, bytes(range(32)))
        felts = random_column(7, 3)
        a.reseed_felts(felts); b.reseed_felts(felts)
        assert a.state[0] == b.digest
        assert np.array_equal(a.draw(), b.draw())
        a.reseed_felt_vector(felts); b.reseed_felt_vector(felts)
        a.reseed_int(777); b.reseed_int(777)
        assert a.draw_queries(9, 1 << 12) == b.draw_queries(9, 1 << 12)
        assert a.state == (b.digest, b.counter)
