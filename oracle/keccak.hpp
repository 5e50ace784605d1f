// ORACLE (test infrastructure). Keccak-256 (original Keccak padding 0x01, NOT SHA3's 0x06),
// as used by plonkish_backend::util::hash::Keccak256 (= sha3::Keccak256) at
// /root/reference/bfv-gkr/src/transcript.rs:11,117,141. Pinned by the KAT Keccak256("") =
// c5d24601...5d85a470 in tests/test_oracle_kats.py.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>

namespace orc_keccak {  // field-independent: shared by both field back ends

static inline uint64_t rotl64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

static inline void keccak_f1600(uint64_t st[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39,
                                41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int round = 0; round < 24; round++) {
        uint64_t C[5], D[5], B[25];
        for (int x = 0; x < 5; x++) C[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
        for (int x = 0; x < 5; x++) D[x] = C[(x + 4) % 5] ^ rotl64(C[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) st[i] ^= D[i % 5];
        // rho + pi: B[y, 2x+3y] = rot(A[x,y])
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) B[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(st[x + 5 * y], ROT[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++)
                st[x + 5 * y] = B[x + 5 * y] ^ ((~B[(x + 1) % 5 + 5 * y]) & B[(x + 2) % 5 + 5 * y]);
        st[0] ^= RC[round];
    }
}

// one-shot Keccak-256 of a byte string
static inline void keccak256(const uint8_t* in, size_t len, uint8_t out[32]) {
    uint64_t st[25];
    memset(st, 0, sizeof(st));
    const size_t rate = 136;
    uint8_t block[136];
    while (len >= rate) {
        for (size_t i = 0; i < rate / 8; i++) {
            uint64_t w;
            memcpy(&w, in + 8 * i, 8);
            st[i] ^= w;  // little-endian host assumed (x86-64)
        }
        keccak_f1600(st);
        in += rate;
        len -= rate;
    }
    memset(block, 0, rate);
    if (len) memcpy(block, in, len);
    block[len] ^= 0x01;
    block[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; i++) {
        uint64_t w;
        memcpy(&w, block + 8 * i, 8);
        st[i] ^= w;
    }
    keccak_f1600(st);
    memcpy(out, st, 32);
}

}  // namespace orc_keccak
