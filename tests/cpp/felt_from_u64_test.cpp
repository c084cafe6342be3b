// Host check of felt_from_u64's short path (sandstorm_amd/host/coin.cpp: v 2^256 mod p = p - (544 v 2^192 + 32 v) for v < 2^49)
// against the general Montgomery product with 2^512 mod p, at the boundaries and on two million values.
#include <cstdint>
#include <cstdio>
#include "../../sandstorm_amd/host/coin.hpp"
using namespace ssh;
int main() {
    const Felt R2 = {0xfffffd737e000401ull, 0x00000001330fffffull, 0xffffffffff6f8000ull, 0x07ffd4ab5e008810ull};
    uint64_t st = 12345;
    int bad = 0;
    auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (int i = 0; i < 2000000; ++i) {
        const uint64_t v = i < 1000 ? (uint64_t)i : i % 3 == 0 ? next() >> (15 + next() % 49) : i % 3 == 1 ? ((1ull << 49) - 1 - (next() % 1000)) + (next() % 2000) : next();
        if (felt_from_u64(v) != felt_mul(Felt{v, 0, 0, 0}, R2)) ++bad;
    }
    printf(bad ? "FELT_FROM_U64_FAIL %d\n" : "FELT_FROM_U64_OK\n", bad);
    return bad != 0;
}
