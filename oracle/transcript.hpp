// ORACLE (test infrastructure). Proof byte stream + challenge generation.
// Follows /root/reference/bfv-gkr/src/transcript.rs:
//   * squeeze_challenge (F): hash = state.finalize_fixed_reset(); state.update(hash);
//     fe_mod_from_le_bytes(hash)                                   (:198-203)
//     -> since write_felt only appends to the stream (:183-189) and common_felt is a
//        no-op (:156), the challenge stream is the fixed chain H1 = Keccak256(""),
//        H_{j+1} = Keccak256(H_j), c_j = LE(H_j) mod p.
//   * squeeze_challenge (E) = from_bases(DEGREE consecutive base challenges)   (:149-154)
//   * write_felt: canonical repr, byte-reversed to big-endian                  (:183-189)
//   * write_felt_ext: bases in order                                           (:191-195)
//   * read_felt: 8 bytes BE -> from_repr_vartime (reject non-canonical)        (:162-170)
// `fe_mod_from_le_bytes` lives in plonkish_backend (not vendored): published behaviour
// restated = little-endian integer reduced mod p. KATs: SURVEY.md §8(c) item 5.
#pragma once
#include <vector>
#include <stdexcept>
#include "gl.hpp"
#include "keccak.hpp"

namespace orc {

static inline uint64_t fe_mod_from_le_bytes32(const uint8_t h[32]) {
    uint64_t l[4];
    memcpy(l, h, 32);
    uint64_t r = f_from_u64(l[3]);
    for (int i = 2; i >= 0; i--) r = f_add(f_mul(r, GL_EPS), f_from_u64(l[i]));
    return r;
}

struct ChallengeChain {
    uint8_t h[32];
    bool started = false;
    uint64_t next_f() {
        if (!started) { keccak256(nullptr, 0, h); started = true; }
        else { uint8_t t[32]; keccak256(h, 32, t); memcpy(h, t, 32); }
        return fe_mod_from_le_bytes32(h);
    }
    E next_e() { uint64_t a = next_f(); uint64_t b = next_f(); return E{a, b}; }
};

struct TranscriptW {
    ChallengeChain ch;
    std::vector<uint8_t> stream;
    E squeeze() { return ch.next_e(); }
    std::vector<E> squeeze_n(size_t n) { std::vector<E> v(n); for (auto& x : v) x = squeeze(); return v; }
    void write_f(uint64_t a) { for (int i = 7; i >= 0; i--) stream.push_back((uint8_t)(a >> (8 * i))); }
    void write_e(E a) { write_f(a.c0); write_f(a.c1); }
    void write_es(const std::vector<E>& v) { for (auto& x : v) write_e(x); }
};

struct TranscriptR {
    ChallengeChain ch;
    const uint8_t* p; size_t len; size_t pos = 0;
    TranscriptR(const uint8_t* p_, size_t l) : p(p_), len(l) {}
    E squeeze() { return ch.next_e(); }
    std::vector<E> squeeze_n(size_t n) { std::vector<E> v(n); for (auto& x : v) x = squeeze(); return v; }
    uint64_t read_f() {
        if (pos + 8 > len) throw std::runtime_error("transcript: unexpected end of proof");
        uint64_t a = 0;
        for (int i = 0; i < 8; i++) a = (a << 8) | p[pos + i];
        pos += 8;
        if (a >= GL_P) throw std::runtime_error("transcript: invalid field element");
        return a;
    }
    E read_e() { uint64_t a = read_f(); uint64_t b = read_f(); return E{a, b}; }
    std::vector<E> read_es(size_t n) { std::vector<E> v(n); for (auto& x : v) x = read_e(); return v; }
};

}  // namespace orc
