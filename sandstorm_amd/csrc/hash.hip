// hash.hip — row / node hashing and proof-of-work grinding for gfx950.
//
// Replaces crypto/src/merkle/utils.rs:19-46 (hash_rows), the HashFn impls of
// crypto/src/hash/keccak.rs:13-98 and blake2s.rs:10-100 as used by the Merkle
// tree configs (crypto/src/merkle/mod.rs:419-437, mixed.rs:106-125), and the
// grinding loops of crypto/src/public_coin/solidity.rs:120-141 / cairo.rs:133-154.
//
// One lane = one row / one node: the matrix is column-major, so lane i reads
// element i of every column — consecutive lanes, consecutive 32-byte elements,
// fully coalesced — and the whole sponge state lives in VGPRs.  The message of a
// row is the concatenation of each element's Montgomery limbs as 32 big-endian
// bytes (keccak.rs:50-58), i.e. limbs in reverse order, each byte-swapped.
#include <hip/hip_runtime.h>
#include <cstring>
#include "fp252.h"
#include "kernels.h"

namespace ss {

// ------------------------------------------------------------------ Keccak
__constant__ uint64_t KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
    0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

// Keccak-f[1600] on 32-bit halves, written for the gfx950 VALU: a 64-bit rotation is two full-rate
// v_alignbit_b32 (the compiler's 64-bit shift pairs are quarter rate), chi's a ^ (~b & c) and theta's
// 3-way xors are single v_bitop3_b32 instructions.  ~190 VALU per round instead of ~290.
struct K64 { uint32_t lo, hi; };
__device__ __forceinline__ K64 k_split(uint64_t x) { K64 r; r.lo = (uint32_t)x; r.hi = (uint32_t)(x >> 32); return r; }
__device__ __forceinline__ uint64_t k_join(const K64 &x) { return ((uint64_t)x.hi << 32) | x.lo; }
template <int N>
__device__ __forceinline__ K64 k_rotl(const K64 &x) {
    K64 r;
    if (N == 0) { r = x; }
    else if (N == 32) { r.lo = x.hi; r.hi = x.lo; }
    else if (N < 32) {
        r.hi = __builtin_amdgcn_alignbit(x.hi, x.lo, 32 - N);       // (hi:lo) >> (32 - N), low word
        r.lo = __builtin_amdgcn_alignbit(x.lo, x.hi, 32 - N);
    } else {
        r.hi = __builtin_amdgcn_alignbit(x.lo, x.hi, 64 - N);
        r.lo = __builtin_amdgcn_alignbit(x.hi, x.lo, 64 - N);
    }
    return r;
}
__device__ __forceinline__ uint32_t k_xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t k_chi(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xD2); }   // a ^ (~b & c)
__device__ __forceinline__ K64 k_xor5(const K64 &a, const K64 &b, const K64 &c, const K64 &d, const K64 &e) {
    K64 r; r.lo = k_xor3(k_xor3(a.lo, b.lo, c.lo), d.lo, e.lo); r.hi = k_xor3(k_xor3(a.hi, b.hi, c.hi), d.hi, e.hi); return r;
}
__device__ __forceinline__ K64 k_xor(const K64 &a, const K64 &b) { K64 r; r.lo = a.lo ^ b.lo; r.hi = a.hi ^ b.hi; return r; }
__device__ __forceinline__ K64 k_chi64(const K64 &a, const K64 &b, const K64 &c) { K64 r; r.lo = k_chi(a.lo, b.lo, c.lo); r.hi = k_chi(a.hi, b.hi, c.hi); return r; }
template <int N>
__device__ __forceinline__ K64 k_xrot(const K64 &a, const K64 &d) { return k_rotl<N>(k_xor(a, d)); }

__device__ __forceinline__ void keccak_f1600(uint64_t st[25]) {
    K64 s[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) s[i] = k_split(st[i]);
#pragma unroll 1
    for (int round = 0; round < 24; ++round) {
        const K64 c0 = k_xor5(s[0], s[5], s[10], s[15], s[20]), c1 = k_xor5(s[1], s[6], s[11], s[16], s[21]);
        const K64 c2 = k_xor5(s[2], s[7], s[12], s[17], s[22]), c3 = k_xor5(s[3], s[8], s[13], s[18], s[23]);
        const K64 c4 = k_xor5(s[4], s[9], s[14], s[19], s[24]);
        const K64 d0 = k_xor(c4, k_rotl<1>(c1)), d1 = k_xor(c0, k_rotl<1>(c2)), d2 = k_xor(c1, k_rotl<1>(c3));
        const K64 d3 = k_xor(c2, k_rotl<1>(c4)), d4 = k_xor(c3, k_rotl<1>(c0));
        // theta + rho + pi into b
        const K64 b0 = k_xor(s[0], d0);
        const K64 b10 = k_xrot<1>(s[1], d1), b20 = k_xrot<62>(s[2], d2), b5 = k_xrot<28>(s[3], d3);
        const K64 b15 = k_xrot<27>(s[4], d4), b16 = k_xrot<36>(s[5], d0), b1 = k_xrot<44>(s[6], d1);
        const K64 b11 = k_xrot<6>(s[7], d2), b21 = k_xrot<55>(s[8], d3), b6 = k_xrot<20>(s[9], d4);
        const K64 b7 = k_xrot<3>(s[10], d0), b17 = k_xrot<10>(s[11], d1), b2 = k_xrot<43>(s[12], d2);
        const K64 b12 = k_xrot<25>(s[13], d3), b22 = k_xrot<39>(s[14], d4), b23 = k_xrot<41>(s[15], d0);
        const K64 b8 = k_xrot<45>(s[16], d1), b18 = k_xrot<15>(s[17], d2), b3 = k_xrot<21>(s[18], d3);
        const K64 b13 = k_xrot<8>(s[19], d4), b14 = k_xrot<18>(s[20], d0), b24 = k_xrot<2>(s[21], d1);
        const K64 b9 = k_xrot<61>(s[22], d2), b19 = k_xrot<56>(s[23], d3), b4 = k_xrot<14>(s[24], d4);
        // chi
        s[0] = k_chi64(b0, b1, b2); s[1] = k_chi64(b1, b2, b3); s[2] = k_chi64(b2, b3, b4); s[3] = k_chi64(b3, b4, b0); s[4] = k_chi64(b4, b0, b1);
        s[5] = k_chi64(b5, b6, b7); s[6] = k_chi64(b6, b7, b8); s[7] = k_chi64(b7, b8, b9); s[8] = k_chi64(b8, b9, b5); s[9] = k_chi64(b9, b5, b6);
        s[10] = k_chi64(b10, b11, b12); s[11] = k_chi64(b11, b12, b13); s[12] = k_chi64(b12, b13, b14); s[13] = k_chi64(b13, b14, b10); s[14] = k_chi64(b14, b10, b11);
        s[15] = k_chi64(b15, b16, b17); s[16] = k_chi64(b16, b17, b18); s[17] = k_chi64(b17, b18, b19); s[18] = k_chi64(b18, b19, b15); s[19] = k_chi64(b19, b15, b16);
        s[20] = k_chi64(b20, b21, b22); s[21] = k_chi64(b21, b22, b23); s[22] = k_chi64(b22, b23, b24); s[23] = k_chi64(b23, b24, b20); s[24] = k_chi64(b24, b20, b21);
        const uint64_t rc = KECCAK_RC[round];
        s[0].lo ^= (uint32_t)rc; s[0].hi ^= (uint32_t)(rc >> 32);
    }
#pragma unroll
    for (int i = 0; i < 25; ++i) st[i] = k_join(s[i]);
}

__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
__device__ __forceinline__ uint64_t bswap64(uint64_t x) { return __builtin_bswap64(x); }

// message lane k (0..3) of one element: LE u64 of its big-endian byte image
__device__ __forceinline__ uint64_t felt_msg_lane(const uint64_t *elem, int k) {
    return bswap64(elem[3 - k]);
}

__device__ __forceinline__ void keccak_store_digest(const uint64_t s[25], uint8_t *out, bool mask20) {
    uint64_t o2 = s[2], o3 = s[3];
    if (mask20) { o2 &= 0xffffffffull; o3 = 0; }      // keep the first 20 bytes (hash/mod.rs:5-13)
    ulonglong2 *q = reinterpret_cast<ulonglong2 *>(out);
    q[0] = make_ulonglong2(s[0], s[1]);
    q[1] = make_ulonglong2(o2, o3);
}

// Absorb `nelem` felts fetched through `fetch(c)` (pointer to the 4 u64 limbs).
template <typename Fetch>
__device__ __forceinline__ void keccak_absorb_felts(uint64_t s[25], uint32_t nelem, Fetch fetch) {
    const uint32_t data_lanes = 4 * nelem;            // pad byte 0x01 sits in lane `data_lanes`
    const uint32_t nblocks = data_lanes / 17 + 1;
    for (uint32_t blk = 0; blk < nblocks; ++blk) {
#pragma unroll
        for (int pos = 0; pos < 17; ++pos) {
            const uint32_t lane = blk * 17 + pos;
            if (lane < data_lanes) s[pos] ^= felt_msg_lane(fetch(lane >> 2), lane & 3);
            else if (lane == data_lanes) s[pos] ^= 0x01ull;
        }
        if (blk == nblocks - 1) s[16] ^= 0x8000000000000000ull;
        keccak_f1600(s);
    }
}

// brev_bits != 0: digest i is that of matrix row bitrev(i) over brev_bits bits - the commitment order of the
// reference's proofs (index i of a committed vector is the point offset * w^bitrev(i); tests/golden/
// make_proof_golden.py), straight out of the natural-order LDE.  The lane keeps its NATURAL row (coalesced reads
// of every column) and scatters the one 32-byte digest to slot bitrev(row): bit reversal is an involution.
__device__ __forceinline__ uint64_t digest_slot(uint64_t row, uint32_t brev_bits) {
    return brev_bits ? (__brevll(row) >> (64u - brev_bits)) : row;
}

__global__ __launch_bounds__(256) void keccak_rows_kernel(ConstColPtrs cols, uint32_t ncols, uint64_t nrows,
                                                          uint8_t *__restrict__ out, int mask20, uint32_t brev_bits) {
    for (uint64_t row = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; row < nrows;
         row += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t s[25];
#pragma unroll
        for (int i = 0; i < 25; ++i) s[i] = 0;
        keccak_absorb_felts(s, ncols, [&](uint32_t c) {
            return reinterpret_cast<const uint64_t *>(cols.p[c]) + 4 * row;
        });
        keccak_store_digest(s, out + 32 * digest_slot(row, brev_bits), mask20 != 0);
    }
}

// Rows of 8-byte elements (the 64-bit field, X4): the message of row i is the little-endian bytes of its elements,
// segment by segment - element e of segment s is segs.p[s][i * seg_len + e].  A trace matrix is nseg columns of seg_len 1;
// the row of a FRI layer over Fq3 (interleaved [len][3], row j = {evals[j + k rows]}) is nseg = fold segments of seg_len 3.
__global__ __launch_bounds__(256) void keccak_rows_u64_kernel(ConstColPtrs segs, uint32_t nseg, uint32_t seg_len, uint64_t nrows,
                                                              uint8_t *__restrict__ out) {
    const uint32_t data_lanes = nseg * seg_len, nblocks = data_lanes / 17 + 1;
    for (uint64_t row = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; row < nrows; row += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t s[25];
#pragma unroll
        for (int i = 0; i < 25; ++i) s[i] = 0;
        uint32_t seg = 0, e = 0;
        for (uint32_t blk = 0; blk < nblocks; ++blk) {
#pragma unroll
            for (int pos = 0; pos < 17; ++pos) {
                const uint32_t lane = blk * 17 + pos;
                if (lane < data_lanes) {
                    const uint64_t *sp = reinterpret_cast<const uint64_t *>(segs.p[0]);
#pragma unroll
                    for (int c = 1; c < MAX_COLS; ++c) if (seg == (uint32_t)c) sp = reinterpret_cast<const uint64_t *>(segs.p[c]);
                    s[pos] ^= sp[row * seg_len + e];
                    if (++e == seg_len) { e = 0; ++seg; }
                } else if (lane == data_lanes) s[pos] ^= 0x01ull;
            }
            if (blk == nblocks - 1) s[16] ^= 0x8000000000000000ull;
            keccak_f1600(s);
        }
        keccak_store_digest(s, out + 32 * row, false);
    }
}
// out[(j * nseg + s) * seg_len + e] = segs.p[s][idx[j] * seg_len + e]: the opened rows of such a matrix
__global__ void gather_rows_u64_kernel(ConstColPtrs segs, uint32_t nseg, uint32_t seg_len, const uint64_t *__restrict__ idx, uint32_t nidx,
                                       uint64_t *__restrict__ out) {
    const uint32_t per = nseg * seg_len, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nidx * per) return;
    const uint32_t j = t / per, r = t - j * per, s = r / seg_len, e = r - s * seg_len;
    out[t] = reinterpret_cast<const uint64_t *>(segs.p[s])[idx[j] * seg_len + e];
}

// leaf level of UnhashedLeafConfig: H::hash_elements([l0, l1]) (merkle/mod.rs:426-428)
__global__ __launch_bounds__(256) void keccak_felt_pairs_kernel(const uint64_t *__restrict__ felts, uint64_t count,
                                                                uint8_t *__restrict__ out, int mask20) {
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < count;
         k += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t s[25];
#pragma unroll
        for (int i = 0; i < 25; ++i) s[i] = 0;
        keccak_absorb_felts(s, 2, [&](uint32_t c) { return felts + 4 * (2 * k + c); });
        keccak_store_digest(s, out + 32 * k, mask20 != 0);
    }
}

// inner nodes: H::merge(n0, n1) = H(n0 || n1) (keccak.rs:27-32)
__global__ __launch_bounds__(256) void keccak_pairs_kernel(const uint8_t *__restrict__ in, uint64_t count,
                                                           uint8_t *__restrict__ out, int mask20) {
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < count;
         k += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t s[25];
        const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(in + 64 * k);
#pragma unroll
        for (int i = 0; i < 4; ++i) { ulonglong2 v = q[i]; s[2 * i] = v.x; s[2 * i + 1] = v.y; }
        s[8] = 0x01ull;
#pragma unroll
        for (int i = 9; i < 25; ++i) s[i] = 0;
        s[16] = 0x8000000000000000ull;
        keccak_f1600(s);
        keccak_store_digest(s, out + 32 * k, mask20 != 0);
    }
}

// ----------------------------------------------------------------- Blake2s
__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return __builtin_rotateright32(x, n); }

#define B2S_G(a, b, c, d, x, y)                       \
    a = a + b + (x); d = rotr32(d ^ a, 16);           \
    c = c + d;       b = rotr32(b ^ c, 12);           \
    a = a + b + (y); d = rotr32(d ^ a, 8);            \
    c = c + d;       b = rotr32(b ^ c, 7);

#define B2S_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    B2S_G(v0, v4, v8, v12, m[s0], m[s1]);   B2S_G(v1, v5, v9, v13, m[s2], m[s3]);       \
    B2S_G(v2, v6, v10, v14, m[s4], m[s5]);  B2S_G(v3, v7, v11, v15, m[s6], m[s7]);      \
    B2S_G(v0, v5, v10, v15, m[s8], m[s9]);  B2S_G(v1, v6, v11, v12, m[s10], m[s11]);    \
    B2S_G(v2, v7, v8, v13, m[s12], m[s13]); B2S_G(v3, v4, v9, v14, m[s14], m[s15]);

__device__ __forceinline__ void blake2s_compress(uint32_t h[8], const uint32_t m[16], uint32_t t, bool last) {
    uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    uint32_t v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au;
    uint32_t v12 = 0x510E527Fu ^ t, v13 = 0x9B05688Cu, v14 = last ? ~0x1F83D9ABu : 0x1F83D9ABu, v15 = 0x5BE0CD19u;
    B2S_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    B2S_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    B2S_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    B2S_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    B2S_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    B2S_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    B2S_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    B2S_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    B2S_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    B2S_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    h[0] ^= v0 ^ v8;  h[1] ^= v1 ^ v9;  h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11;
    h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}

__device__ __forceinline__ void blake2s_init(uint32_t h[8]) {
    h[0] = 0x6A09E667u ^ 0x01010020u; h[1] = 0xBB67AE85u; h[2] = 0x3C6EF372u; h[3] = 0xA54FF53Au;
    h[4] = 0x510E527Fu; h[5] = 0x9B05688Cu; h[6] = 0x1F83D9ABu; h[7] = 0x5BE0CD19u;
}

__device__ __forceinline__ void blake2s_store_digest(const uint32_t h[8], uint8_t *out, bool mask20) {
    uint4 *q = reinterpret_cast<uint4 *>(out);
    if (mask20) {  // keep the LAST 20 bytes (hash/mod.rs:15-23)
        q[0] = make_uint4(0, 0, 0, h[3]);
    } else {
        q[0] = make_uint4(h[0], h[1], h[2], h[3]);
    }
    q[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

// message words of one element: w-th LE u32 of its big-endian image
__device__ __forceinline__ void felt_msg_words(const Fp &e, uint32_t *m) {
#pragma unroll
    for (int w = 0; w < 8; ++w) m[w] = bswap32(e.v[7 - w]);
}
__device__ __forceinline__ Fp load_fp(const void *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fp r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}

template <typename Fetch>
__device__ __forceinline__ void blake2s_hash_felts(uint32_t h[8], uint32_t nelem, Fetch fetch) {
    blake2s_init(h);
    const uint32_t nblocks = nelem == 0 ? 1 : (nelem + 1) / 2;
    for (uint32_t blk = 0; blk < nblocks; ++blk) {
        uint32_t m[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = 0;
        if (2 * blk < nelem) felt_msg_words(fetch(2 * blk), m);
        if (2 * blk + 1 < nelem) felt_msg_words(fetch(2 * blk + 1), m + 8);
        const bool last = blk == nblocks - 1;
        const uint32_t t = last ? 32 * nelem : 64 * (blk + 1);
        blake2s_compress(h, m, t, last);
    }
}

__global__ __launch_bounds__(256) void blake2s_rows_kernel(ConstColPtrs cols, uint32_t ncols, uint64_t nrows,
                                                           uint8_t *__restrict__ out, int mask20, uint32_t brev_bits) {
    for (uint64_t row = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; row < nrows;
         row += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t h[8];
        blake2s_hash_felts(h, ncols, [&](uint32_t c) {
            return load_fp(reinterpret_cast<const uint8_t *>(cols.p[c]) + 32 * row);
        });
        blake2s_store_digest(h, out + 32 * digest_slot(row, brev_bits), mask20 != 0);
    }
}

__global__ __launch_bounds__(256) void blake2s_felt_pairs_kernel(const Fp *__restrict__ felts, uint64_t count,
                                                                 uint8_t *__restrict__ out, int mask20) {
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < count;
         k += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t h[8];
        blake2s_hash_felts(h, 2, [&](uint32_t c) { return load_fp(felts + 2 * k + c); });
        blake2s_store_digest(h, out + 32 * k, mask20 != 0);
    }
}

__global__ __launch_bounds__(256) void blake2s_pairs_kernel(const uint8_t *__restrict__ in, uint64_t count,
                                                            uint8_t *__restrict__ out, int mask20) {
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < count;
         k += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t h[8], m[16];
        const uint4 *q = reinterpret_cast<const uint4 *>(in + 64 * k);
#pragma unroll
        for (int i = 0; i < 4; ++i) { uint4 v = q[i]; m[4 * i] = v.x; m[4 * i + 1] = v.y; m[4 * i + 2] = v.z; m[4 * i + 3] = v.w; }
        blake2s_init(h);
        blake2s_compress(h, m, 64, true);
        blake2s_store_digest(h, out + 32 * k, mask20 != 0);
    }
}

__global__ __launch_bounds__(256) void felts_to_be_kernel(const Fp *__restrict__ felts, uint64_t count,
                                                          uint8_t *__restrict__ out) {
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < count;
         k += (uint64_t)gridDim.x * blockDim.x) {
        Fp e = load_fp(felts + k);
        uint4 *q = reinterpret_cast<uint4 *>(out + 32 * k);
        q[0] = make_uint4(bswap32(e.v[7]), bswap32(e.v[6]), bswap32(e.v[5]), bswap32(e.v[4]));
        q[1] = make_uint4(bswap32(e.v[3]), bswap32(e.v[2]), bswap32(e.v[1]), bswap32(e.v[0]));
    }
}

// ------------------------------------------------------------ PoW grinding
// prefix = H(be64(0x0123456789ABCDED) || digest || bits)  (41 bytes)
// (solidity.rs:121-125 / cairo.rs:134-138), then for each nonce
// H(prefix || be64(nonce)) must have >= bits leading zero bits; the smallest
// valid nonce of the window wins (atomicMin).
struct Digest32 { uint64_t w[4]; };

__global__ void pow_prefix_kernel(int coin_kind, Digest32 digest, uint32_t bits, uint64_t *__restrict__ prefix) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint64_t magic = 0x0123456789ABCDEDull;
    if (coin_kind == 0) {
        uint64_t s[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) s[k] = 0;
        s[0] = bswap64(magic);
        s[1] = digest.w[0]; s[2] = digest.w[1]; s[3] = digest.w[2]; s[4] = digest.w[3];
        s[5] = (uint64_t)(bits & 0xffu) | (0x01ull << 8);
        s[16] = 0x8000000000000000ull;
        keccak_f1600(s);
        prefix[0] = s[0]; prefix[1] = s[1]; prefix[2] = s[2]; prefix[3] = s[3];
    } else {
        uint32_t h[8], m[16];
        m[0] = bswap32((uint32_t)(magic >> 32)); m[1] = bswap32((uint32_t)magic);
#pragma unroll
        for (int k = 0; k < 4; ++k) { m[2 + 2 * k] = (uint32_t)digest.w[k]; m[3 + 2 * k] = (uint32_t)(digest.w[k] >> 32); }
        m[10] = bits & 0xffu;
#pragma unroll
        for (int k = 11; k < 16; ++k) m[k] = 0;
        blake2s_init(h);
        blake2s_compress(h, m, 41, true);
#pragma unroll
        for (int k = 0; k < 4; ++k) prefix[k] = (uint64_t)h[2 * k] | ((uint64_t)h[2 * k + 1] << 32);
    }
}

__global__ __launch_bounds__(256) void pow_grind_kernel(int coin_kind, const uint64_t *__restrict__ prefix,
                                                        uint32_t bits, uint64_t start, uint64_t count,
                                                        unsigned long long *__restrict__ best) {
    const uint64_t p0 = prefix[0], p1 = prefix[1], p2 = prefix[2], p3 = prefix[3];
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t nonce = start + i;
        uint64_t lead;  // first 8 digest bytes as a big-endian integer
        if (coin_kind == 0) {
            uint64_t s[25];
#pragma unroll
            for (int k = 0; k < 25; ++k) s[k] = 0;
            s[0] = p0; s[1] = p1; s[2] = p2; s[3] = p3;
            s[4] = bswap64(nonce);
            s[5] = 0x01ull;
            s[16] = 0x8000000000000000ull;
            keccak_f1600(s);
            lead = bswap64(s[0]);
        } else {
            uint32_t h[8], m[16];
            m[0] = (uint32_t)p0; m[1] = (uint32_t)(p0 >> 32); m[2] = (uint32_t)p1; m[3] = (uint32_t)(p1 >> 32);
            m[4] = (uint32_t)p2; m[5] = (uint32_t)(p2 >> 32); m[6] = (uint32_t)p3; m[7] = (uint32_t)(p3 >> 32);
            m[8] = bswap32((uint32_t)(nonce >> 32));
            m[9] = bswap32((uint32_t)nonce);
#pragma unroll
            for (int k = 10; k < 16; ++k) m[k] = 0;
            blake2s_init(h);
            blake2s_compress(h, m, 40, true);
            lead = ((uint64_t)bswap32(h[0]) << 32) | bswap32(h[1]);
        }
        const bool ok = bits == 0 || (lead >> (64 - bits)) == 0;
        if (ok) atomicMin(best, (unsigned long long)nonce);
    }
}

// generic gathers: out[i] = in[idx[i]] (32-byte digests / 1-byte tags)
__global__ void gather32_kernel(const uint8_t *__restrict__ in, const uint64_t *__restrict__ idx, uint64_t n,
                                uint8_t *__restrict__ out) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 *q = reinterpret_cast<const uint4 *>(in + 32 * idx[i]);
    uint4 *o = reinterpret_cast<uint4 *>(out + 32 * i);
    o[0] = q[0]; o[1] = q[1];
}
// rows of a column set: out[(q * row_cols + c0 + c)] = cols[c][idx[q]] for c < ncols - the opened rows in the order the proof lists them
// (row after row), one launch for the chunk's columns
__global__ void gather32_cols_kernel(ConstColPtrs cols, uint32_t ncols, uint32_t c0, uint32_t row_cols, const uint64_t *__restrict__ idx, uint64_t nidx,
                                     uint8_t *__restrict__ out) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= nidx * ncols) return;
    const uint64_t q = i / ncols;
    const uint32_t c = (uint32_t)(i - q * ncols);
    const uint4 *src = reinterpret_cast<const uint4 *>((const uint8_t *)cols.p[c] + 32 * idx[q]);
    uint4 *o = reinterpret_cast<uint4 *>(out + 32 * (q * row_cols + c0 + c));
    o[0] = src[0]; o[1] = src[1];
}
__global__ void gather8_kernel(const uint8_t *__restrict__ in, const uint64_t *__restrict__ idx, uint64_t n,
                               uint8_t *__restrict__ out) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}

// -------------------------------------------------------------- launchers
// ---------------------------------------------------------------- SHA-256 (FIPS 180-4)
// The reference instantiates its 64-bit-field claim with ministark's Sha256HashFn trees (cli/src/main.rs:105,119: un-vendored;
// the hash itself is public: FIPS 180-4, pinned by its vectors in tests/test_goldilocks.py).  Rows of 8-byte elements as their
// little-endian bytes, like the other two hashes of ss_hash_rows_gl64; a node is SHA-256(left || right).
__device__ __forceinline__ uint32_t sha_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }      // -> v_alignbit_b32
__device__ __constant__ uint32_t SHA256_K[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
    0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
    0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
    0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
__device__ __forceinline__ void sha256_init(uint32_t h[8]) {
    h[0] = 0x6a09e667u; h[1] = 0xbb67ae85u; h[2] = 0x3c6ef372u; h[3] = 0xa54ff53au; h[4] = 0x510e527fu; h[5] = 0x9b05688cu; h[6] = 0x1f83d9abu; h[7] = 0x5be0cd19u;
}
// one 64-byte block: w[0..15] = its big-endian words (clobbered: the schedule runs in place)
__device__ __forceinline__ void sha256_compress(uint32_t h[8], uint32_t w[16]) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int t = 0; t < 64; ++t) {
        if (t >= 16) {
            const uint32_t w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
            const uint32_t s0 = sha_rotr(w15, 7) ^ sha_rotr(w15, 18) ^ (w15 >> 3), s1 = sha_rotr(w2, 17) ^ sha_rotr(w2, 19) ^ (w2 >> 10);
            w[t & 15] = w[t & 15] + s0 + w[(t + 9) & 15] + s1;
        }
        const uint32_t S1 = sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25), ch = (e & f) ^ (~e & g);
        const uint32_t t1 = hh + S1 + ch + SHA256_K[t] + w[t & 15];
        const uint32_t S0 = sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22), maj = (a & b) ^ (a & c) ^ (b & c);
        const uint32_t t2 = S0 + maj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
__device__ __forceinline__ void sha256_store_digest(const uint32_t h[8], uint8_t *out) {
    uint4 *q = reinterpret_cast<uint4 *>(out);
    q[0] = make_uint4(bswap32(h[0]), bswap32(h[1]), bswap32(h[2]), bswap32(h[3]));
    q[1] = make_uint4(bswap32(h[4]), bswap32(h[5]), bswap32(h[6]), bswap32(h[7]));
}
// rows of 8-byte elements (ss_hash_rows_gl64's shape): 8 elements per block, then the padding (0x80, zeros, the bit length)
__global__ __launch_bounds__(256) void sha256_rows_u64_kernel(ConstColPtrs segs, uint32_t nseg, uint32_t seg_len, uint64_t nrows,
                                                              uint8_t *__restrict__ out) {
    const uint32_t nelem = nseg * seg_len;
    const uint32_t nblocks = (8 * nelem + 9 + 63) / 64;                  // message + 0x80 + 8 length bytes
    for (uint64_t row = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; row < nrows; row += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t h[8];
        sha256_init(h);
        uint32_t seg = 0, e = 0;
        for (uint32_t blk = 0; blk < nblocks; ++blk) {
            uint32_t w[16];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t idx = 8 * blk + i;
                uint64_t v = 0;
                if (idx < nelem) {
                    const uint64_t *sp = reinterpret_cast<const uint64_t *>(segs.p[0]);
#pragma unroll
                    for (int c = 1; c < MAX_COLS; ++c) if (seg == (uint32_t)c) sp = reinterpret_cast<const uint64_t *>(segs.p[c]);
                    v = sp[row * seg_len + e];
                    if (++e == seg_len) { e = 0; ++seg; }
                } else if (idx == nelem) v = 0x80ull;                     // the byte after the message (little-endian image of the element slot)
                // the element's little-endian bytes b0..b7 are message bytes: big-endian words (b0 b1 b2 b3), (b4 b5 b6 b7)
                w[2 * i] = bswap32((uint32_t)v); w[2 * i + 1] = bswap32((uint32_t)(v >> 32));
            }
            if (blk == nblocks - 1) { w[14] = 0; w[15] = 64u * nelem; }  // bit length (< 2^32: rows of at most 16 x 64 elements)
            sha256_compress(h, w);
        }
        sha256_store_digest(h, out + 32 * row);
    }
}
// node = SHA-256(left || right): one data block and the constant padding block of a 64-byte message
__global__ __launch_bounds__(256) void sha256_pairs_kernel(const uint8_t *__restrict__ in, uint64_t count, uint8_t *__restrict__ out) {
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < count; k += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t h[8], w[16];
        const uint4 *q = reinterpret_cast<const uint4 *>(in + 64 * k);
#pragma unroll
        for (int i = 0; i < 4; ++i) { uint4 v = q[i]; w[4 * i] = bswap32(v.x); w[4 * i + 1] = bswap32(v.y); w[4 * i + 2] = bswap32(v.z); w[4 * i + 3] = bswap32(v.w); }
        sha256_init(h);
        sha256_compress(h, w);
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = 0;
        w[0] = 0x80000000u; w[15] = 512u;
        sha256_compress(h, w);
        sha256_store_digest(h, out + 32 * k);
    }
}

static inline uint32_t grid_for(uint64_t n, uint32_t block, uint32_t cap) {
    uint64_t g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    return (uint32_t)g;
}

// the same rows under Blake2s-256 (8 elements per 64-byte block)
__global__ __launch_bounds__(256) void blake2s_rows_u64_kernel(ConstColPtrs segs, uint32_t nseg, uint32_t seg_len, uint64_t nrows,
                                                               uint8_t *__restrict__ out) {
    const uint32_t nelem = nseg * seg_len, nblocks = nelem == 0 ? 1 : (nelem + 7) / 8;
    for (uint64_t row = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; row < nrows; row += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t h[8];
        blake2s_init(h);
        uint32_t seg = 0, e = 0;
        for (uint32_t blk = 0; blk < nblocks; ++blk) {
            uint32_t m[16];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint64_t v = 0;
                if (8 * blk + i < nelem) {
                    const uint64_t *sp = reinterpret_cast<const uint64_t *>(segs.p[0]);
#pragma unroll
                    for (int c = 1; c < MAX_COLS; ++c) if (seg == (uint32_t)c) sp = reinterpret_cast<const uint64_t *>(segs.p[c]);
                    v = sp[row * seg_len + e];
                    if (++e == seg_len) { e = 0; ++seg; }
                }
                m[2 * i] = (uint32_t)v; m[2 * i + 1] = (uint32_t)(v >> 32);
            }
            const bool last = blk == nblocks - 1;
            blake2s_compress(h, m, last ? 8 * nelem : 64 * (blk + 1), last);
        }
        blake2s_store_digest(h, out + 32 * row, false);
    }
}

hipError_t launch_hash_rows_u64(hipStream_t st, int kind, const ConstColPtrs &segs, uint32_t nseg, uint32_t seg_len, uint64_t nrows, uint8_t *digests) {
    if (kind == 0) hipLaunchKernelGGL(keccak_rows_u64_kernel, dim3(grid_for(nrows, 256, 1u << 20)), dim3(256), 0, st, segs, nseg, seg_len, nrows, digests);
    else if (kind == 2) hipLaunchKernelGGL(sha256_rows_u64_kernel, dim3(grid_for(nrows, 256, 1u << 20)), dim3(256), 0, st, segs, nseg, seg_len, nrows, digests);
    else hipLaunchKernelGGL(blake2s_rows_u64_kernel, dim3(grid_for(nrows, 256, 1u << 20)), dim3(256), 0, st, segs, nseg, seg_len, nrows, digests);
    return hipGetLastError();
}
hipError_t launch_gather_rows_u64(hipStream_t st, const ConstColPtrs &segs, uint32_t nseg, uint32_t seg_len, const uint64_t *d_idx, uint32_t nidx,
                                  uint64_t *d_out) {
    const uint32_t total = nidx * nseg * seg_len;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_u64_kernel, dim3((total + 255) / 256), dim3(256), 0, st, segs, nseg, seg_len, d_idx, nidx, d_out);
    return hipGetLastError();
}
hipError_t launch_hash_rows(hipStream_t st, int kind, const ConstColPtrs &cols, uint32_t ncols,
                            uint64_t nrows, uint32_t brev_bits, uint8_t *digests) {
    const uint32_t grid = grid_for(nrows, 256, 1u << 20);
    if (kind == 0 || kind == 1)
        hipLaunchKernelGGL(keccak_rows_kernel, dim3(grid), dim3(256), 0, st, cols, ncols, nrows, digests, kind == 1, brev_bits);
    else
        hipLaunchKernelGGL(blake2s_rows_kernel, dim3(grid), dim3(256), 0, st, cols, ncols, nrows, digests, kind == 3, brev_bits);
    return hipGetLastError();
}
// dst[i] = src[bitrev(i)] (32-byte elements): the bit-reversed image of a single-column matrix whose raw
// elements are the leaves of its tree
__global__ __launch_bounds__(256) void bitrev_copy_kernel(const Fp *__restrict__ src, uint32_t log_n, Fp *__restrict__ dst) {
    const uint64_t n = 1ull << log_n;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t j = log_n ? (__brevll(i) >> (64u - log_n)) : 0;
        const uint4 *q = reinterpret_cast<const uint4 *>(src + j);
        uint4 *d = reinterpret_cast<uint4 *>(dst + i);
        d[0] = q[0]; d[1] = q[1];
    }
}
hipError_t launch_bitrev_copy(hipStream_t st, const Fp *src, uint32_t log_n, Fp *dst) {
    hipLaunchKernelGGL(bitrev_copy_kernel, dim3(grid_for(1ull << log_n, 256, 1u << 20)), dim3(256), 0, st, src, log_n, dst);
    return hipGetLastError();
}
hipError_t launch_hash_pairs(hipStream_t st, int kind, const uint8_t *in, uint64_t count, uint8_t *out) {
    const uint32_t grid = grid_for(count, 256, 1u << 20);
    if (kind == 4)                              // SS_HASH_SHA256 (the header's enum is not included here)
        hipLaunchKernelGGL(sha256_pairs_kernel, dim3(grid), dim3(256), 0, st, in, count, out);
    else if (kind == 0 || kind == 1)
        hipLaunchKernelGGL(keccak_pairs_kernel, dim3(grid), dim3(256), 0, st, in, count, out, kind == 1);
    else
        hipLaunchKernelGGL(blake2s_pairs_kernel, dim3(grid), dim3(256), 0, st, in, count, out, kind == 3);
    return hipGetLastError();
}
hipError_t launch_hash_felt_pairs(hipStream_t st, int kind, const Fp *felts, uint64_t count, uint8_t *out) {
    const uint32_t grid = grid_for(count, 256, 1u << 20);
    if (kind == 0 || kind == 1)
        hipLaunchKernelGGL(keccak_felt_pairs_kernel, dim3(grid), dim3(256), 0, st,
                           reinterpret_cast<const uint64_t *>(felts), count, out, kind == 1);
    else
        hipLaunchKernelGGL(blake2s_felt_pairs_kernel, dim3(grid), dim3(256), 0, st, felts, count, out, kind == 3);
    return hipGetLastError();
}
hipError_t launch_felts_to_be(hipStream_t st, const Fp *felts, uint64_t count, uint8_t *out) {
    hipLaunchKernelGGL(felts_to_be_kernel, dim3(grid_for(count, 256, 1u << 20)), dim3(256), 0, st, felts, count, out);
    return hipGetLastError();
}
hipError_t launch_pow_prefix(hipStream_t st, int coin_kind, const uint8_t digest[32], uint32_t bits,
                             uint64_t *d_prefix) {
    Digest32 d;
    memcpy(d.w, digest, 32);
    hipLaunchKernelGGL(pow_prefix_kernel, dim3(1), dim3(64), 0, st, coin_kind, d, bits, d_prefix);
    return hipGetLastError();
}
hipError_t launch_pow_grind(hipStream_t st, int coin_kind, const uint64_t *d_prefix, uint32_t bits,
                            uint64_t start, uint64_t count, unsigned long long *d_best) {
    hipLaunchKernelGGL(pow_grind_kernel, dim3(grid_for(count, 256, 1u << 16)), dim3(256), 0, st, coin_kind,
                       d_prefix, bits, start, count, d_best);
    return hipGetLastError();
}
hipError_t launch_gather32(hipStream_t st, const uint8_t *in, const uint64_t *d_idx, uint64_t n, uint8_t *out) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gather32_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, in, d_idx, n, out);
    return hipGetLastError();
}
hipError_t launch_gather32_cols(hipStream_t st, const void *const *d_cols, uint32_t ncols, uint32_t c0, uint32_t row_cols, const uint64_t *d_idx,
                                uint64_t nidx, uint8_t *out) {
    ConstColPtrs cp;
    for (uint32_t c = 0; c < ncols; ++c) cp.p[c] = d_cols[c];
    const uint64_t total = nidx * ncols;
    hipLaunchKernelGGL(gather32_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, cp, ncols, c0, row_cols, d_idx, nidx, out);
    return hipGetLastError();
}
hipError_t launch_gather8(hipStream_t st, const uint8_t *in, const uint64_t *d_idx, uint64_t n, uint8_t *out) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gather8_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, in, d_idx, n, out);
    return hipGetLastError();
}

}  // namespace ss
