// ORACLE (test infrastructure). Proof byte stream + challenge generation.
// Follows /root/reference/bfv-gkr/src/transcript.rs:
//   * squeeze_challenge (F): hash = state.finalize_fixed_reset(); state.update(hash);
//     fe_mod_from_le_bytes(hash)                                   (:198-203)
//     -> since the gkr-trait write_felt only appends to the stream (:183-189) and common_felt is a
//        no-op (:156), the challenge stream is the fixed chain H1 = Keccak256(""),
//        H_{j+1} = Keccak256(H_j), c_j = LE(H_j) mod p.
//   * squeeze_challenge (E) = from_bases(DEGREE consecutive base challenges)   (:149-154)
//   * write_felt: canonical repr, byte-reversed to big-endian                  (:183-189)
//   * write_felt_ext: bases in order                                           (:191-195)
//   * read_felt: F_BYTES bytes BE -> from_repr_vartime (reject non-canonical)  (:162-170)
// `fe_mod_from_le_bytes` lives in plonkish_backend (not vendored): published behaviour
// restated = little-endian integer reduced mod p. KATs: SURVEY.md §8(c) item 5.
//
// ProtocolMode (both default off = the reference as it is):
//   absorb       write_felt / read_felt also hash the element the way the in-tree plonkish-trait writer of the SAME struct
//                does: common_field_element -> state.update(fe.to_repr()) before the bytes go to the stream
//                (transcript.rs:205-208, 224-233). Challenges then depend on every prover message.
//   ext_memcheck the memory-checking challenges gamma, tau stay in E instead of being truncated to base limb 0
//                (lasso/src/memory_checking/prover.rs:36-39; README.md:108 "Known issues"): hash tables live in E.
#pragma once
#include <vector>
#include <stdexcept>
#include "field.hpp"
#include "keccak.hpp"

namespace ORC_NS {

struct ProtocolMode {
    bool absorb = false;
    bool ext_memcheck = false;
};

// Keccak state as the byte string absorbed since the last squeeze (H::update appends; finalize_fixed_reset hashes and clears)
struct FsState {
    std::vector<uint8_t> pending;
    void update(const uint8_t* p, size_t n) { pending.insert(pending.end(), p, p + n); }
    F squeeze_f() {  // transcript.rs:198-203
        uint8_t h[32];
        ::orc_keccak::keccak256(pending.data(), pending.size(), h);
        pending.assign(h, h + 32);
        return f_from_hash_le(h);
    }
    E squeeze_e() {  // transcript.rs:149-154
        F b[E_DEGREE];
        for (size_t i = 0; i < E_DEGREE; i++) b[i] = squeeze_f();
        return e_from_bases(b);
    }
    void absorb_f(F a) { uint8_t r[F_BYTES]; f_repr_le(a, r); update(r, F_BYTES); }
};

struct TranscriptW {
    ProtocolMode mode;
    FsState st;
    std::vector<uint8_t> stream;
    TranscriptW() {}
    explicit TranscriptW(ProtocolMode m) : mode(m) {}
    E squeeze() { return st.squeeze_e(); }
    std::vector<E> squeeze_n(size_t n) { std::vector<E> v(n); for (auto& x : v) x = squeeze(); return v; }
    void write_f(F a) {
        if (mode.absorb) st.absorb_f(a);
        size_t at = stream.size();
        stream.resize(at + F_BYTES);
        f_write_be(a, stream.data() + at);
    }
    void write_e(E a) { F b[E_DEGREE]; e_as_bases(a, b); for (size_t i = 0; i < E_DEGREE; i++) write_f(b[i]); }
    void write_es(const std::vector<E>& v) { for (auto& x : v) write_e(x); }
};

struct TranscriptR {
    ProtocolMode mode;
    FsState st;
    const uint8_t* p; size_t len; size_t pos = 0;
    TranscriptR(const uint8_t* p_, size_t l, ProtocolMode m = ProtocolMode()) : mode(m), p(p_), len(l) {}
    E squeeze() { return st.squeeze_e(); }
    std::vector<E> squeeze_n(size_t n) { std::vector<E> v(n); for (auto& x : v) x = squeeze(); return v; }
    F read_f() {
        if (pos + F_BYTES > len) throw std::runtime_error("transcript: unexpected end of proof");
        F a;
        if (!f_read_be(p + pos, a)) throw std::runtime_error("transcript: invalid field element");
        pos += F_BYTES;
        if (mode.absorb) st.absorb_f(a);
        return a;
    }
    E read_e() { F b[E_DEGREE]; for (size_t i = 0; i < E_DEGREE; i++) b[i] = read_f(); return e_from_bases(b); }
    std::vector<E> read_es(size_t n) { std::vector<E> v(n); for (auto& x : v) x = read_e(); return v; }
};

}  // namespace ORC_NS
