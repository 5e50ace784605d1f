#include "host.hpp"
#include <algorithm>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <queue>
#include <sstream>
#include <functional>
#include <omp.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace hg {

// ------------------------------------------------------------------------------------------------
// Keccak-256 (rate 136, pad 0x01..0x80) — sha3::Keccak256 as used by transcript.rs:117,141
static const u64 KECCAK_RC[24] = {
    0x1ULL, 0x8082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x808bULL, 0x80000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x8aULL, 0x88ULL, 0x80008009ULL, 0x8000000aULL,
    0x8000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x80000001ULL, 0x8000000080008008ULL};
// rho rotation and pi destination of lane i = x + 5y, walked along the pi cycle starting at lane 1
static const int KECCAK_PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
static const int KECCAK_ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};

static inline u64 rol(u64 x, int s) { return (x << s) | (x >> (64 - s)); }

static void keccak_permute(u64 a[25]) {
    for (int rnd = 0; rnd < 24; rnd++) {
        u64 bc[5];
        for (int i = 0; i < 5; i++) bc[i] = a[i] ^ a[i + 5] ^ a[i + 10] ^ a[i + 15] ^ a[i + 20];
        for (int i = 0; i < 5; i++) {
            u64 t = bc[(i + 4) % 5] ^ rol(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) a[j + i] ^= t;
        }
        u64 t = a[1];
        for (int i = 0; i < 24; i++) {
            int j = KECCAK_PILN[i];
            u64 b = a[j];
            a[j] = rol(t, KECCAK_ROTC[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = a[j + i];
            for (int i = 0; i < 5; i++) a[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        a[0] ^= KECCAK_RC[rnd];
    }
}

void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
    u64 st[25] = {0};
    uint8_t* sb = reinterpret_cast<uint8_t*>(st);  // little-endian host
    const size_t rate = 136;
    size_t fill = 0;
    for (size_t i = 0; i < len; i++) {
        sb[fill++] ^= data[i];
        if (fill == rate) { keccak_permute(st); fill = 0; }
    }
    sb[fill] ^= 0x01;
    sb[rate - 1] ^= 0x80;
    keccak_permute(st);
    memcpy(out, sb, 32);
}

// fe_mod_from_le_bytes (transcript.rs:202): 256-bit little-endian integer mod p
u64 felt_from_hash(const uint8_t h[32]) {
    u64 limb[4];
    memcpy(limb, h, 32);
    u64 acc = 0;
    for (int i = 3; i >= 0; i--) acc = gl_add(gl_mul(acc, GL_EPS), gl_from_u64(limb[i]));  // * 2^64 = * (2^32-1)
    return acc;
}

const u64* challenge_chain(size_t n_base) {
    static std::mutex mu;
    static std::vector<u64>* cache = nullptr;  // leaked on purpose: pointers stay valid for the process lifetime
    static uint8_t last[32];
    static std::vector<std::vector<u64>*> old;
    std::lock_guard<std::mutex> lk(mu);
    if (!cache) cache = new std::vector<u64>();
    if (cache->size() < n_base) {
        size_t want = std::max<size_t>(n_base, cache->size() * 2 + 4096);
        auto* grown = new std::vector<u64>(*cache);
        grown->reserve(want);
        while (grown->size() < want) {
            uint8_t nh[32];
            if (grown->empty()) keccak256(nullptr, 0, nh);
            else keccak256(last, 32, nh);
            memcpy(last, nh, 32);
            grown->push_back(felt_from_hash(nh));
        }
        old.push_back(cache);  // keep earlier buffers alive for readers holding a pointer
        cache = grown;
    }
    return cache->data();
}

// ------------------------------------------------------------------------------------------------
static int ilog2u(u64 x) { return 63 - __builtin_clzll(x); }

Params::Params(const hg_params& p) : raw(p) {
    if (p.n < 4 || (p.n & (p.n - 1))) throw Error("n must be a power of two");
    if (p.k < 1 || p.k > HG_MAX_K || (p.k & (p.k - 1))) throw Error("k must be a power of two <= 16");
    n_log2 = ilog2u(p.n);
    L = n_log2 + 1;
    k = (int)p.k;
    log2k = ilog2u(p.k);
}

static const hg_params BUILTIN[] = {
#include "params_table.inc"
};

bool params_builtin(uint32_t n, uint32_t k, hg_params* out) {
    for (const hg_params& b : BUILTIN)
        if (b.n == n && b.k == k) { *out = b; return true; }
    return false;
}


// The constants emitter of scripts/circuit_sk.py (:80 k0i, :249 k1_bound, :296-297 r2i_bound, :334-337 r1i_bound, :422-439 the
// emitted Rust constants). Python's `/` is true division to a double: int((q - 1) / 2) is the 53-bit rounding of (q-1)/2, which
// is why the shipped R2 bounds are not exactly (q-1)/2; the R1 bound divides an exact big integer by q the same way.
static u64 py_int_of_quotient(unsigned __int128 num, u64 den) {  // int(num / den) with num / den rounded to the nearest double
    const unsigned __int128 q = num / den, rem = num % den;
    if (q >= ((unsigned __int128)1 << 53)) {  // the quotient itself does not fit a double's mantissa: round it to 53 bits (ties to even)
        int bits = 0;
        for (unsigned __int128 t = q; t; t >>= 1) bits++;
        const int drop = bits - 53;
        unsigned __int128 m = q >> drop;
        const unsigned __int128 lost = q & (((unsigned __int128)1 << drop) - 1), half = (unsigned __int128)1 << (drop - 1);
        if (lost > half || (lost == half && (rem || (m & 1)))) m++;
        return (u64)(m << drop);
    }
    // small quotient: the fraction rem / den only matters if it rounds the double up to the next integer
    int bits = 0;
    for (unsigned __int128 t = q; t; t >>= 1) bits++;
    const int frac_bits = 53 - bits;  // bits of the fraction a double keeps
    // rounds up to q + 1 iff 1 - rem/den <= 2^-(frac_bits + 1)
    const unsigned __int128 gap = den - rem;
    if (frac_bits < 126 && (gap << (frac_bits + 1)) <= (unsigned __int128)den && rem != 0) return (u64)q + 1;
    return (u64)q;
}
static u64 inv_mod(u64 a, u64 m) {  // a^-1 mod m (extended Euclid), gcd(a, m) = 1
    __int128 t = 0, nt = 1, r = m, nr = a % m;
    while (nr != 0) {
        __int128 qq = r / nr;
        __int128 tmp = t - qq * nt; t = nt; nt = tmp;
        tmp = r - qq * nr; r = nr; nr = tmp;
    }
    if (r != 1) throw Error("hg_params_derive: t is not invertible modulo a q_i");
    if (t < 0) t += m;
    return (u64)t;
}
void params_derive(uint32_t n, uint32_t k, const u64* qis, u64 t, hg_params* out) {
    if (n < 2 || (n & (n - 1))) throw Error("hg_params_derive: n must be a power of two");
    if (k < 1 || k > HG_MAX_K || (k & (k - 1))) throw Error("hg_params_derive: k must be 1, 2, 4, 8 or 16");
    if (t < 3 || !(t & 1)) throw Error("hg_params_derive: t must be odd");
    memset(out, 0, sizeof(*out));
    out->n = n; out->k = k;
    out->s_bound = 1;                                    // :225
    out->e_bound = 19;                                   // :236 int(discrete_gaussian.z_upper), sigma = 3.2
    out->k1_bound = py_int_of_quotient(t - 1, 2);        // :249
    for (uint32_t i = 0; i < k; i++) {
        const u64 q = qis[i];
        if (q < 3 || !(q & 1) || q >= (1ULL << 62)) throw Error("hg_params_derive: q_i must be odd and below 2^62");
        const u64 k0 = inv_mod(q - (t % q), q);          // :80 pow(-t, -1, q)
        const u64 half = py_int_of_quotient(q - 1, 2);   // :296 int((qis[i] - 1) / 2)
        const unsigned __int128 num = (unsigned __int128)half * (n + 2) + 19 + (unsigned __int128)out->k1_bound * k0;  // :334
        out->qis[i] = q; out->k0is[i] = k0;
        out->r2_bounds[i] = half;
        out->r1_bounds[i] = py_int_of_quotient(num, q);  // :336 int(.. / qis[i])
    }
}

static bool env_token(const char* var, const char* token) {
    const char* e = getenv(var);
    if (!e || !*e) return false;
    const size_t n = strlen(token);
    for (const char* p = e; (p = strstr(p, token)) != nullptr; p += n)
        if ((p == e || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
    return false;
}
bool hg_debug(const char* token) { return env_token("HG_DEBUG", token); }
bool hg_times(const char* token) { return env_token("HG_TIMES", token); }
const char* hg_proof_map_path() { const char* e = getenv("HG_PROOF_MAP"); return e && *e ? e : nullptr; }
bool hg_env_on(const char* name) { const char* e = getenv(name); return e && e[0] == '1'; }

// ------------------------------------------------------------------------------------------------
// OpenMP inside a container: omp_get_max_threads() reports the machine's cores (256 on the MI355X boxes) while the cgroup may grant
// far fewer CPUs (16 there: /sys/fs/cgroup/cpu.max = "1600000 100000"), and libomp's idle workers spin for 200 ms after every
// parallel region. Together they exhaust the quota and the kernel THROTTLES the whole process for the rest of the 100 ms period -
// measured as 30-40 ms stalls of the prover's host thread (the sequential prover's mailbox made them visible; cpu.stat
// nr_throttled counted them). So, once per process: no more OpenMP threads than the quota grants, and idle workers sleep at once.
extern "C" void kmp_set_blocktime(int) __attribute__((weak));
int cgroup_cpu_quota() {   // CPUs granted by the cgroup (v2 cpu.max, v1 cfs quota), 0 = unlimited / unknown
    auto read2 = [](const char* path, long long* a, long long* b) -> int {
        FILE* f = fopen(path, "r");
        if (!f) return 0;
        char buf[64] = {0};
        int n = 0;
        if (fgets(buf, sizeof(buf), f)) {
            if (strncmp(buf, "max", 3) == 0) n = -1;
            else n = sscanf(buf, "%lld %lld", a, b);
        }
        fclose(f);
        return n;
    };
    long long q = 0, per = 0;
    int n = read2("/sys/fs/cgroup/cpu.max", &q, &per);
    if (n == 2 && q > 0 && per > 0) return (int)((q + per - 1) / per);
    if (n == 0) {
        long long q1 = 0, p1 = 0, dummy = 0;
        if (read2("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", &q1, &dummy) >= 1 && read2("/sys/fs/cgroup/cpu/cpu.cfs_period_us", &p1, &dummy) >= 1 && q1 > 0 && p1 > 0)
            return (int)((q1 + p1 - 1) / p1);
    }
    return 0;
}
// threads an OpenMP region of this library may use: the runtime's default, capped by the cgroup quota (unless OMP_NUM_THREADS says
// otherwise). A clause on the library's own regions - the process-wide default of a host application is not touched.
int hg_omp_threads() {
    static const int n = [] {
        int t = omp_get_max_threads();
        if (!getenv("OMP_NUM_THREADS")) { const int q = cgroup_cpu_quota(); if (q > 0 && q < t) t = q; }
        return t < 1 ? 1 : t;
    }();
    return n;
}
static const int g_openmp_guard = [] {
    // (idle workers that spin burn the quota for every thread of the process: the one process-wide setting this library makes)
    if (!getenv("KMP_BLOCKTIME") && !getenv("OMP_WAIT_POLICY") && kmp_set_blocktime) kmp_set_blocktime(0);
    return 0;
}();

// ------------------------------------------------------------------------------------------------
// JSON witness: {"s":[".."],"e":[..],"k1":[..],"r2is":[[..]],"r1is":[[..]],"ais":[[..]],"ct0is":[[..]]}
namespace {
// One flat array of quoted decimals whose elements are parsed later, in parallel (parse_array_jobs)
struct ArrayJob { size_t begin, end; int field, z; };   // bytes [begin, end) between the brackets; field: 0 s, 1 e, 2 k1, 3 r2is, 4 r1is, 5 ais, 6 ct0is

struct JsonCursor {
    const char* s;
    size_t n;
    size_t i = 0;
    std::vector<ArrayJob> jobs;
    std::vector<size_t> closers;   // positions of every ']' of the text, ascending (found in parallel by the constructor)
    JsonCursor(const char* p, size_t len) : s(p), n(len) {
        const int nt = std::max(1, std::min<int>(hg_omp_threads(), (int)std::min<size_t>(32, len / ((size_t)1 << 20) + 1)));
        std::vector<std::vector<size_t>> part(nt);
#pragma omp parallel for schedule(static, 1) num_threads(nt)
        for (int t = 0; t < nt; t++) {
            size_t b = len * (size_t)t / nt;
            const size_t e = len * (size_t)(t + 1) / nt;
            while (b < e) {
                const void* q = memchr(p + b, ']', e - b);
                if (!q) break;
                part[t].push_back((size_t)((const char*)q - p));
                b = part[t].back() + 1;
            }
        }
        for (auto& v : part) closers.insert(closers.end(), v.begin(), v.end());
    }
    size_t next_closer(size_t from) const {
        auto it = std::lower_bound(closers.begin(), closers.end(), from);
        if (it == closers.end()) throw Error("witness json: unterminated array");
        return *it;
    }
    void ws() { while (i < n && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) i++; }
    bool eat(char c) { ws(); if (i < n && s[i] == c) { i++; return true; } return false; }
    void need(char c) { if (!eat(c)) throw Error(std::string("witness json: expected '") + c + "' at offset " + std::to_string(i)); }
    std::string str() {
        need('"');
        size_t b = i;
        const void* q = memchr(s + i, '"', n - i);
        if (!q) throw Error("witness json: unterminated string");
        i = (size_t)((const char*)q - s);
        return std::string(s + b, i++ - b);
    }
    bool bn254 = false;  // coefficients are bn256::Fr elements holding small signed integers (see witness_from_json_bn254)
    u64 felt_bn254(const std::string& d) {
        // decimal -> 256-bit integer v < r; z = v (v < 2^62) or -(r - v) (r - v < 2^62); returned in the Goldilocks form of z
        static const u64 RL[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
        u64 v[4] = {0, 0, 0, 0};
        for (char c : d) {
            if (c < '0' || c > '9') throw Error("witness json: non-decimal coefficient");
            unsigned __int128 carry = (unsigned)(c - '0');
            for (int i = 0; i < 4; i++) {
                unsigned __int128 t = (unsigned __int128)v[i] * 10 + carry;
                v[i] = (u64)t;
                carry = t >> 64;
            }
            if (carry) throw Error("witness json: coefficient out of field range");
        }
        bool lt = false;
        for (int i = 3; i >= 0; i--) if (v[i] != RL[i]) { lt = v[i] < RL[i]; break; }
        if (!lt) throw Error("witness json: coefficient out of field range");
        if (!(v[1] | v[2] | v[3]) && v[0] < (1ULL << 62)) return v[0];
        u64 m[4];
        unsigned __int128 borrow = 0;
        for (int i = 0; i < 4; i++) {
            unsigned __int128 t = (unsigned __int128)RL[i] - v[i] - (u64)borrow;
            m[i] = (u64)t;
            borrow = (t >> 64) & 1;
        }
        if ((m[1] | m[2] | m[3]) || m[0] >= (1ULL << 62) || m[0] == 0)
            throw Error("witness json: bn254 coefficient is not a signed integer below 2^62 in magnitude (unsupported witness)");
        return GL_P - m[0];
    }
    u64 felt_digits(const char* d, size_t len) {  // F::from_str_vartime on a decimal string (poly.rs:13-16)
        if (len == 0) throw Error("witness json: empty coefficient");
        if (bn254) return felt_bn254(std::string(d, len));
        if (len <= 19) {   // below 10^19 < 2^64: plain 64-bit accumulation
            u64 v = 0;
            for (size_t q = 0; q < len; q++) {
                const unsigned c = (unsigned char)d[q] - '0';
                if (c > 9) throw Error("witness json: non-decimal coefficient");
                v = v * 10 + c;
            }
            if (v >= GL_P) throw Error("witness json: coefficient out of field range");
            return v;
        }
        unsigned __int128 v = 0;
        for (size_t q = 0; q < len; q++) {
            const char c = d[q];
            if (c < '0' || c > '9') throw Error("witness json: non-decimal coefficient");
            v = v * 10 + (unsigned)(c - '0');
            if (v >= ((unsigned __int128)1 << 64)) throw Error("witness json: coefficient out of field range");
        }
        if ((u64)v >= GL_P) throw Error("witness json: coefficient out of field range");
        return (u64)v;
    }
    // A flat array: only its extent is found here (a decimal string cannot contain ']', so the next ']' closes it; anything else
    // between the brackets is rejected by the element parser); the elements are parsed by parse_array_jobs.
    void felts(int field, int z = 0) {
        need('[');
        const size_t end = next_closer(i);
        jobs.push_back(ArrayJob{i, end, field, z});
        i = end + 1;
    }
    int felts2(int field) {   // -> number of inner arrays
        need('[');
        if (eat(']')) return 0;
        int z = 0;
        do felts(field, z++); while (eat(','));
        need(']');
        return z;
    }
    // Parses the elements of bytes [b, e): ws* ( '"' digits '"' ws* (',' | end) )*. Strict: anything else is an error.
    void parse_piece(size_t b, size_t e, std::vector<u64>& out) {
        size_t q = b;
        auto skip = [&] { while (q < e && (s[q] == ' ' || s[q] == '\n' || s[q] == '\t' || s[q] == '\r')) q++; };
        skip();
        while (q < e) {
            if (s[q] != '"') throw Error("witness json: expected '\"' at offset " + std::to_string(q));
            const size_t d0 = ++q;
            while (q < e && s[q] != '"') q++;
            if (q >= e) throw Error("witness json: unterminated string");
            out.push_back(felt_digits(s + d0, q - d0));
            q++;
            skip();
            if (q < e) {
                if (s[q] != ',') throw Error("witness json: expected ',' at offset " + std::to_string(q));
                q++;
                skip();
                if (q >= e) throw Error("witness json: trailing ',' in an array");
            }
        }
    }
    // All deferred arrays, in parallel: every array is cut into pieces at element boundaries (the byte after a ','), every piece is
    // parsed by one thread into its own vector, the pieces are concatenated per array. The 53 MB of decimal strings of an
    // n=32768 k=16 witness (2.7 M coefficients) take a few ms on a many-core host instead of > 100 ms.
    // place(job, count) -> where the job's `count` elements go (checks the count; the destination is already zero-filled)
    void parse_array_jobs(const std::function<u64*(const ArrayJob&, size_t)>& place) {
        struct Piece { size_t job, b, e; std::vector<u64> v; };
        std::vector<Piece> pieces;
        const size_t target = (size_t)256 << 10;   // bytes per piece
        for (size_t j = 0; j < jobs.size(); j++) {
            size_t b = jobs[j].begin;
            const size_t e = jobs[j].end;
            while (b < e) {
                size_t cut = e;
                if (e - b > target + target / 2) {
                    const void* c = memchr(s + b + target, ',', e - (b + target));
                    if (c) cut = (size_t)((const char*)c - s) + 1;   // the piece keeps its closing ','; the next starts at an element
                }
                pieces.push_back(Piece{j, b, cut, {}});
                b = cut;
            }
        }
        std::string err;
        [[maybe_unused]] const int nt = std::max(1, std::min<int>(hg_omp_threads(), std::min<int>(64, (int)pieces.size())));   // (the device pass of hipcc ignores the pragma)
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
        for (size_t q = 0; q < pieces.size(); q++) {
            Piece& P = pieces[q];
            try {
                P.v.reserve((P.e - P.b) / 16 + 8);
                size_t e = P.e;
                const bool last = e == jobs[P.job].end;
                if (!last) e--;   // drop the separator that ends this piece (parse_piece would call it a trailing ',')
                parse_piece(P.b, e, P.v);
            } catch (const std::exception& ex) {
#pragma omp critical
                if (err.empty()) err = ex.what();
            }
        }
        if (!err.empty()) throw Error(err);
        std::vector<size_t> first(jobs.size() + 1, pieces.size());
        for (size_t q = pieces.size(); q-- > 0;) first[pieces[q].job] = q;
        for (size_t j = jobs.size(); j-- > 0;) if (first[j] == pieces.size()) first[j] = first[j + 1];
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
        for (size_t j = 0; j < jobs.size(); j++) {
            size_t total = 0;
            for (size_t q = first[j]; q < pieces.size() && pieces[q].job == j; q++) total += pieces[q].v.size();
            try {
                u64* d = place(jobs[j], total);
                size_t at = 0;
                for (size_t q = first[j]; d && q < pieces.size() && pieces[q].job == j; q++) {
                    memcpy(d + at, pieces[q].v.data(), pieces[q].v.size() * 8);
                    at += pieces[q].v.size();
                }
            } catch (const std::exception& ex) {
#pragma omp critical
                if (err.empty()) err = ex.what();
            }
        }
        if (!err.empty()) throw Error(err);
        jobs.clear();
    }
};

// Poly::new_padded (poly.rs:20-28)
void put_padded(const std::vector<u64>& c, size_t size, u64* dst) {
    if (c.size() > size) throw Error("witness: polynomial longer than its table");
    std::fill(dst, dst + size, 0);
    std::copy(c.begin(), c.end(), dst);
}
// Poly::new_shifted (poly.rs:30-44): left-pad to `size`, then resize to next_power_of_two(size)
std::vector<u64> shifted(const std::vector<u64>& c, size_t size) {
    size_t pad = size > c.size() ? size - c.size() : 0;
    std::vector<u64> v(pad, 0);
    v.insert(v.end(), c.begin(), c.end());
    size_t np = 1;
    while (np < size) np <<= 1;
    v.resize(np, 0);
    return v;
}

struct RawArgs {  // BfvSkEncryptArgs (sk_encryption_circuit.rs:64-73), coefficients highest degree first
    std::vector<u64> s, e, k1;
    std::vector<std::vector<u64>> r2is, r1is, ais, ct0is;
};

Witness layout_inputs(const Params& p, const RawArgs& a) {  // get_inputs (sk_encryption_circuit.rs:365-415)
    const size_t SZ = p.SZ(), PZ = p.PZ();
    const size_t k = (size_t)p.k;
    if (a.ct0is.size() < k || a.ais.size() < k || a.r1is.size() < k || a.r2is.size() < k) throw Error("witness: fewer CRT components than k");
    Witness w;
    w.s.resize(SZ);
    put_padded(a.s, SZ, w.s.data());
    w.e = shifted(a.e, SZ - 1);
    w.k1 = shifted(a.k1, SZ - 1);
    if (w.e.size() != SZ || w.k1.size() != SZ) throw Error("witness: e/k1 length");
    w.ais.resize(k * SZ);
    w.r1is.resize(k * SZ);
    w.r2is.assign(k * PZ, 0);
    w.ct0is.resize(k * SZ);
    for (size_t z = 0; z < k; z++) if (a.r2is[z].size() + 1 != PZ) throw Error("witness: r2i must have n-1 coefficients");
    std::string err;
#pragma omp parallel for schedule(static, 1) num_threads((int)std::min<size_t>(k, (size_t)hg_omp_threads()))
    for (size_t z = 0; z < k; z++) {
        try {
            put_padded(a.ais[z], SZ, &w.ais[z * SZ]);
            put_padded(a.r1is[z], SZ, &w.r1is[z * SZ]);
            std::copy(a.r2is[z].begin(), a.r2is[z].end(), &w.r2is[z * PZ]);  // + one trailing zero (:402-405)
            std::vector<u64> ct = shifted(a.ct0is[z], SZ);                    // :393
            if (ct.size() != SZ) throw Error("witness: ct0i length");
            std::copy(ct.begin() + 1, ct.end(), &w.ct0is[z * SZ]);            // [1..] then push 0 (:394-395)
            w.ct0is[z * SZ + SZ - 1] = 0;
        } catch (const std::exception& ex) {   // (an exception must not leave the parallel region)
#pragma omp critical
            if (err.empty()) err = ex.what();
        }
    }
    if (!err.empty()) throw Error(err);
    return w;
}
}  // namespace

static Witness witness_from_json_impl(const Params& p, const std::string& path, bool bn254);
Witness witness_from_json(const Params& p, const std::string& path) { return witness_from_json_impl(p, path, false); }
// The bn254 fixtures hold the same witness with negatives assigned as r - |z| (scripts/utils.py:4-18). Every coefficient of a
// valid witness is a small signed integer (range-checked, or centred modulo q_i < 2^60), so the loader recovers z and keeps
// it in the Goldilocks form (p - |z|); the BN254 prover lifts z into Fr on the device (bn254_gkr.inc: k_bn_lift_signed).
Witness witness_from_json_bn254(const Params& p, const std::string& path) { return witness_from_json_impl(p, path, true); }
static Witness witness_from_json_impl(const Params& p, const std::string& path, bool bn254) {
    // the file is mapped, not copied; the structure is walked once (only bracket searches), the coefficients are parsed in parallel
    struct Mapped {
        int fd = -1; void* p = MAP_FAILED; size_t n = 0;
        ~Mapped() { if (p != MAP_FAILED) munmap(p, n); if (fd >= 0) close(fd); }
    } m;
    m.fd = open(path.c_str(), O_RDONLY);
    if (m.fd < 0) throw Error("witness json: cannot open " + path);
    struct stat stt;
    if (fstat(m.fd, &stt) != 0 || stt.st_size <= 0) throw Error("witness json: cannot read " + path);
    m.n = (size_t)stt.st_size;
    m.p = mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE | MAP_POPULATE, m.fd, 0);
    if (m.p == MAP_FAILED) throw Error("witness json: cannot map " + path);
    const bool tm = hg_times("json");
    const double t0 = omp_get_wtime();
    JsonCursor c(static_cast<const char*>(m.p), m.n);
    c.bn254 = bn254;
    c.need('{');
    int seen = 0, inner[7] = {1, 1, 1, 0, 0, 0, 0};
    do {
        std::string key = c.str();
        c.need(':');
        if (key == "s") c.felts(0);
        else if (key == "e") c.felts(1);
        else if (key == "k1") c.felts(2);
        else if (key == "r2is") inner[3] = c.felts2(3);
        else if (key == "r1is") inner[4] = c.felts2(4);
        else if (key == "ais") inner[5] = c.felts2(5);
        else if (key == "ct0is") inner[6] = c.felts2(6);
        else throw Error("witness json: unknown key " + key);
        seen++;
    } while (c.eat(','));
    c.need('}');
    if (seen != 7) throw Error("witness json: expected 7 fields");
    const double t1 = omp_get_wtime();
    // get_inputs (sk_encryption_circuit.rs:365-415) applied while the coefficients are placed: the layouts of layout_inputs()
    // (Poly::new_padded / new_shifted, poly.rs:20-44) as destination offsets, so the 22 MB of tables are written once
    const size_t SZ = p.SZ(), PZ = p.PZ(), k = (size_t)p.k;
    for (int f = 3; f < 7; f++) if ((size_t)inner[f] < k) throw Error("witness: fewer CRT components than k");
    Witness w;
    w.s.resize(SZ); w.e.resize(SZ); w.k1.resize(SZ);
    w.ais.resize(k * SZ); w.r1is.resize(k * SZ); w.r2is.resize(k * PZ); w.ct0is.resize(k * SZ);
    // (a coefficient count that does not fit its table is an error here; Poly::new_shifted's resize would truncate it silently)
    c.parse_array_jobs([&](const ArrayJob& J, size_t cnt) -> u64* {
        const size_t z = (size_t)J.z;
        if (J.field >= 3 && z >= k) return nullptr;   // components beyond k are parsed (validated) and ignored, as get_inputs ignores them
        switch (J.field) {
            case 0: if (cnt > SZ) throw Error("witness: polynomial longer than its table"); return w.s.data();                       // new_padded
            case 1: case 2: {                                                                                                        // new_shifted(.., 2^L - 1)
                if (cnt > SZ - 1) throw Error("witness: e/k1 length");
                return (J.field == 1 ? w.e.data() : w.k1.data()) + (SZ - 1 - cnt);
            }
            case 3: if (cnt + 1 != PZ) throw Error("witness: r2i must have n-1 coefficients"); return &w.r2is[z * PZ];             // + one trailing zero (:402-405)
            case 4: if (cnt > SZ) throw Error("witness: polynomial longer than its table"); return &w.r1is[z * SZ];
            case 5: if (cnt > SZ) throw Error("witness: polynomial longer than its table"); return &w.ais[z * SZ];
            default: {                                                                                                               // shifted to 2^L, first dropped, 0 pushed (:393-395)
                if (cnt + 1 > SZ) throw Error("witness: ct0i length");
                return &w.ct0is[z * SZ] + (SZ - cnt - 1);
            }
        }
    });
    const double t2 = omp_get_wtime();
    if (tm) fprintf(stderr, "[hg] json: structure %.2f ms, coefficients + layout %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
    return w;
}

// ------------------------------------------------------------------------------------------------
// Synthetic witness: the math of scripts/circuit_sk.py:18-140 on seeded integer-only samplers.
//   s uniform in {-1,0,1}; e centred binomial (eta = 20, variance 10 ~ sigma 3.2) truncated to +-19;
//   a_i uniform in [-(q_i-1)/2, (q_i-1)/2]; k1 uniform in [-(t-1)/2, (t-1)/2] ([q m]_t is a bijection of m);
//   ct0i_hat = a_i s + e + k0_i k1 over Z;  ct0i = ct0i_hat mod (X^n+1, q_i) centred;
//   r2i = ((ct0i - ct0i_hat) mod q_i) / (X^n+1);  r1i = (ct0i - ct0i_hat - r2i (X^n+1)) / q_i;
//   negatives are assigned as p - |z| (utils.py:4-18).
namespace {
struct SplitMix64 {
    u64 x;
    u64 next() {
        u64 z = (x += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    u64 below(u64 m) {  // uniform in [0, m) by rejection
        u64 lim = ~0ULL - (~0ULL % m);
        u64 v;
        do v = next(); while (v >= lim);
        return v % m;
    }
};
typedef __int128 i128;
inline u64 assign(i128 z) { return z >= 0 ? (u64)z : GL_P - (u64)(-z); }
inline i128 centre_mod(i128 z, i128 q) {
    i128 r = z % q;
    if (r < 0) r += q;
    if (r > (q - 1) / 2) r -= q;
    return r;
}
}  // namespace

Witness witness_synthetic(const Params& p, u64 seed) {
    const size_t n = p.raw.n, k = (size_t)p.k;
    const int64_t t = 65537;
    for (int attempt = 0; attempt < 8; attempt++) {
        SplitMix64 rng{seed + 0x1000003ULL * (u64)attempt};
        std::vector<int64_t> s(n), e(n), k1(n);  // ascending degree
        for (auto& x : s) x = (int64_t)rng.below(3) - 1;
        for (auto& x : e) {
            int64_t v;
            do {
                u64 bits = rng.next();
                v = (int64_t)__builtin_popcountll(bits & 0xFFFFFULL) - (int64_t)__builtin_popcountll((bits >> 20) & 0xFFFFFULL);
            } while (v > 19 || v < -19);
            x = v;
        }
        for (auto& x : k1) x = (int64_t)rng.below((u64)t) - (t - 1) / 2;
        std::vector<std::vector<int64_t>> a(k, std::vector<int64_t>(n));
        for (size_t i = 0; i < k; i++) {
            u64 q = p.raw.qis[i];
            for (auto& x : a[i]) x = (int64_t)rng.below(q) - (int64_t)((q - 1) / 2);
        }
        RawArgs out;
        out.r2is.resize(k); out.r1is.resize(k); out.ais.resize(k); out.ct0is.resize(k);
        bool ok = true;
#pragma omp parallel for schedule(dynamic, 1) num_threads(hg_omp_threads())
        for (long long ii = 0; ii < (long long)k; ii++) {
            size_t i = (size_t)ii;
            const i128 q = (i128)p.raw.qis[i];
            const i128 k0 = (i128)p.raw.k0is[i];
            // a_i * s over Z with a split into 32-bit halves so the inner loops stay in int64
            std::vector<int64_t> alo(n), ahi(n), acc_lo(2 * n, 0), acc_hi(2 * n, 0);
            for (size_t m = 0; m < n; m++) { alo[m] = (int64_t)((u64)a[i][m] & 0xFFFFFFFFULL); ahi[m] = a[i][m] >> 32; }
            for (size_t j = 0; j < n; j++) {
                if (s[j] == 0) continue;
                int64_t* dl = &acc_lo[j];
                int64_t* dh = &acc_hi[j];
                if (s[j] > 0) for (size_t m = 0; m < n; m++) { dl[m] += alo[m]; dh[m] += ahi[m]; }
                else for (size_t m = 0; m < n; m++) { dl[m] -= alo[m]; dh[m] -= ahi[m]; }
            }
            std::vector<i128> h(2 * n, 0);  // ct0i_hat, ascending, degree <= 2n-2
            for (size_t j = 0; j < 2 * n - 1; j++) h[j] = (i128)acc_hi[j] * ((i128)1 << 32) + (i128)acc_lo[j];   // (a product, not a shift: acc_hi may be negative)
            for (size_t j = 0; j < n; j++) h[j] += (i128)e[j] + k0 * (i128)k1[j];
            std::vector<i128> ct0(n), r2(n - 1), r1(2 * n - 1);
            for (size_t j = 0; j < n; j++) ct0[j] = centre_mod(h[j] - h[j + n], q);
            for (size_t j = 0; j + 1 < n; j++) r2[j] = centre_mod(-h[j + n], q);
            bool good = true;
            for (size_t j = 0; j < 2 * n - 1; j++) {
                i128 d = (j < n ? ct0[j] : 0) - h[j];
                size_t jj = j < n ? j : j - n;
                if (jj + 1 < n) d -= r2[jj];
                if (d % q != 0) good = false;
                r1[j] = d / q;
            }
            const i128 b1 = (i128)p.raw.r1_bounds[i], b2 = (i128)p.raw.r2_bounds[i], b2c = (i128)p.raw.r2_bounds[0];
            for (auto v : r1) if (v > b1 || v < -b1) good = false;
            for (auto v : r2) if (v > b2 || v < -b2 || v > b2c || v < -b2c) good = false;
            auto rev = [&](const std::vector<i128>& v) {
                std::vector<u64> o(v.size());
                for (size_t j = 0; j < v.size(); j++) o[v.size() - 1 - j] = assign(v[j]);
                return o;
            };
            out.ct0is[i] = rev(ct0);
            out.r2is[i] = rev(r2);
            out.r1is[i] = rev(r1);
            std::vector<i128> ai(n);
            for (size_t j = 0; j < n; j++) ai[j] = a[i][j];
            out.ais[i] = rev(ai);
            if (!good) {
#pragma omp critical
                ok = false;
            }
        }
        if (!ok) continue;  // a coefficient left its range-check bound (probability ~2^-40): resample
        auto rev64 = [&](const std::vector<int64_t>& v) {
            std::vector<u64> o(v.size());
            for (size_t j = 0; j < v.size(); j++) o[v.size() - 1 - j] = assign((i128)v[j]);
            return o;
        };
        out.s = rev64(s); out.e = rev64(e); out.k1 = rev64(k1);
        return layout_inputs(p, out);
    }
    throw Error("synthetic witness: could not satisfy the range bounds");
}

// ------------------------------------------------------------------------------------------------
// Lasso preprocessing
int LassoPlan::lookup_index(u64 bound) const {
    for (size_t i = 0; i < lookups.size(); i++) if (lookups[i].bound == bound) return (int)i;
    throw Error("lasso: unknown lookup bound");
}

std::string LassoPlan::layout_text() const {
    std::string s;
    for (int m = 0; m < alpha; m++) { if (m) s += ","; s += mems[m].subtable_id + "@" + std::to_string(mems[m].dim); }
    s += "|";
    for (size_t l = 0; l < lookups.size(); l++) {
        if (l) s += ";";
        s += lookups[l].id + ":" + std::to_string(lookups[l].total_bits) + ":";
        for (size_t i = 0; i < lookups[l].mems.size(); i++) { if (i) s += "/"; s += std::to_string(lookups[l].mems[i]); }
    }
    return s;
}

namespace {
// RangeLookup decomposition of one bound (range.rs:207-250): which subtable serves which dimension
struct RangeShape {
    int full_dims;      // dimensions 0..full_dims-1 use the full 16-bit limb table
    bool has_rem;       // one BoundSubtable at dimension `rem_dim`
    int rem_dim;
    u64 cutoff;         // BoundSubtable cutoff (range.rs:59-61)
    int total_bits;     // sum(chunk_bits) (range.rs:234-250)
};
RangeShape range_shape(u64 bound) {
    const u64 M = 1u << LassoPlan::LOGM;
    RangeShape r{};
    int bits = ilog2u(bound);
    int limbs = bits / LassoPlan::LOGM;
    u64 cutoff = (1ULL << (bits % LassoPlan::LOGM)) + bound % M;
    if (bound % M == 0) { r.full_dims = limbs; r.has_rem = false; }
    else if (bound < M) { r.full_dims = 0; r.has_rem = true; r.rem_dim = 0; }
    else { r.full_dims = limbs; r.has_rem = true; r.rem_dim = limbs; }
    r.cutoff = cutoff;
    r.total_bits = limbs * LassoPlan::LOGM + (bound % M != 0 ? ilog2u(cutoff) : 0);
    return r;
}
}  // namespace

LassoPlan lasso_preprocess(const Params& p) {
    LassoPlan lp;
    // setup(): S, E, K1, R1[..k], R2[..k] as RangeLookup(2b+1) (sk_encryption_circuit.rs:327-341), keyed by id string
    std::map<std::string, u64> by_id;
    auto add = [&](u64 b) { by_id["range_" + std::to_string(2 * b + 1)] = 2 * b + 1; };
    add(p.raw.s_bound); add(p.raw.e_bound); add(p.raw.k1_bound);
    for (int i = 0; i < p.k; i++) add(p.raw.r1_bounds[i]);
    for (int i = 0; i < p.k; i++) add(p.raw.r2_bounds[i]);
    // subtables in first-seen order over lookups in key order; per subtable the union of its dimensions
    std::vector<u64> st_dims_mask;
    auto subtable_slot = [&](const std::string& id, u64 bound) {
        for (size_t i = 0; i < lp.subtable_ids.size(); i++) if (lp.subtable_ids[i] == id) return (int)i;
        lp.subtable_ids.push_back(id);
        lp.subtable_bound.push_back(bound);
        st_dims_mask.push_back(0);
        return (int)lp.subtable_ids.size() - 1;
    };
    struct Use { int subtable; u64 dims_mask; };
    std::vector<std::vector<Use>> uses;
    for (auto& kv : by_id) {
        RangeShape sh = range_shape(kv.second);
        std::vector<Use> u;
        if (sh.full_dims > 0 || !sh.has_rem) {
            int si = subtable_slot("full", 0);
            u64 mask = (1ULL << sh.full_dims) - 1;
            st_dims_mask[si] |= mask;
            u.push_back({si, mask});
        }
        if (sh.has_rem) {
            int si = subtable_slot("bound_" + std::to_string(kv.second), kv.second);
            st_dims_mask[si] |= 1ULL << sh.rem_dim;
            u.push_back({si, 1ULL << sh.rem_dim});
        }
        uses.push_back(u);
        lp.lookups.push_back({kv.second, kv.first, sh.total_bits, {}});
    }
    // memories: contiguous per subtable, ascending dimension (lasso.rs:574-586)
    std::vector<std::vector<int>> st_mems(lp.subtable_ids.size());
    for (size_t si = 0; si < lp.subtable_ids.size(); si++)
        for (int d = 0; d < 64; d++)
            if (st_dims_mask[si] >> d & 1) {
                u64 cutoff = lp.subtable_bound[si] ? range_shape(lp.subtable_bound[si]).cutoff : (1u << LassoPlan::LOGM);
                if (cutoff > (1u << LassoPlan::LOGM)) cutoff = 1u << LassoPlan::LOGM;
                st_mems[si].push_back((int)lp.mems.size());
                lp.mems.push_back({(int)si, d, cutoff, lp.subtable_ids[si]});
            }
    lp.alpha = (int)lp.mems.size();
    for (size_t l = 0; l < lp.lookups.size(); l++)
        for (auto& u : uses[l])
            for (int m : st_mems[u.subtable])
                if (u.dims_mask >> lp.mems[m].dim & 1) lp.lookups[l].mems.push_back(m);
    for (auto& m : lp.mems) if (m.dim >= LassoPlan::C) throw Error("lasso: dimension index exceeds C");
    // node rows (sk_encryption_circuit.rs:182-204)
    const int P = p.n_log2, L = p.L;
    const int r2i_l = p.k == 1 ? L : P;
    lp.seg_shift = P;
    auto push_rows = [&](u64 bound, int log2rows) {
        uint8_t id = (uint8_t)lp.lookup_index(2 * bound + 1);
        lp.seg_lookup.insert(lp.seg_lookup.end(), (size_t)1 << (log2rows - P), id);
    };
    for (int i = 0; i < p.k; i++) push_rows(p.raw.r1_bounds[i], L);
    for (int i = 0; i < p.k; i++) push_rows(p.raw.r2_bounds[i], r2i_l);
    push_rows(p.raw.s_bound, L); push_rows(p.raw.e_bound, L); push_rows(p.raw.k1_bound, L);
    lp.rows = lp.seg_lookup.size() << P;
    lp.nu = 0;
    while (((size_t)1 << lp.nu) < lp.rows) lp.nu++;
    // memory-checking chunks
    std::map<int, std::vector<int>> cm;
    for (int m = 0; m < lp.alpha; m++) cm[lp.mems[m].dim].push_back(m);
    for (auto& kv : cm) {
        lp.chunks.push_back(kv);
        for (int m : kv.second) { lp.gkr_order.push_back(m); lp.gkr_chunk.push_back(kv.first); }
        if (kv.first >= lp.alpha) throw Error("lasso: chunk index has no counter memory");
    }
    return lp;
}

// ------------------------------------------------------------------------------------------------
// Circuit wiring (BfvEncryptBlock::configure)
namespace {
struct Gates {  // VanillaNode::new(input_arity, log2_sub_input_size, gates, num_reps)
    HNode n;
    u32 g = 0;
    Gates(int arity, int log2_sub_in, int reps) {
        n.kind = NK_VANILLA; n.arity = arity; n.log2_sub_in = log2_sub_in; n.log2_reps = ilog2u((u64)reps);
    }
    void relay(u32 i, u32 j, u64 scale = 1, u64 add = 0) {  // relay / relay_mul_const / relay_add_const (:525-531)
        if (add) n.w0.push_back({g, add});
        n.lin.push_back({g, i, j, scale});
        g++;
    }
    void zero() { g++; }  // VanillaGate::constant(F::ZERO)
    void mul(u32 i0, u32 j0, u32 i1, u32 j1) { n.mul.push_back({g, i0, j0, i1, j1, 1}); g++; }
    void begin_sum() {}
    void sum_term(u32 i, u32 j) { n.lin.push_back({g, i, j, 1}); }
    void end_sum() { g++; }
    HNode done() {
        n.num_gates = g;
        n.log2_sub_out = 0;
        while ((1u << n.log2_sub_out) < g) n.log2_sub_out++;
        n.left_use.assign(n.arity, 0);
        n.right_use.assign(n.arity, 0);
        for (auto& t : n.lin) n.left_use[t.in] = 1;
        for (auto& t : n.mul) { n.left_use[t.i0] = 1; n.right_use[t.i1] = 1; }
        return n;
    }
};
}  // namespace

HCircuit build_circuit(const Params& p, const LassoPlan& lp) {
    HCircuit c;
    auto insert = [&](HNode n) { c.nodes.push_back(std::move(n)); return (int)c.nodes.size() - 1; };
    auto connect = [&](int from, int to) { c.nodes[to].preds.push_back(from); c.nodes[from].succs.push_back(to); };
    auto input = [&](int log2, int reps) { HNode n; n.kind = NK_INPUT; n.log2_size = log2 + ilog2u((u64)reps); return insert(n); };
    auto fft = [&](int log2, bool inv) { HNode n; n.kind = NK_FFT; n.log2_size = log2; n.inverse = inv; return insert(n); };
    const int P = p.n_log2, L = p.L, k = p.k;
    const u32 SZ = (u32)1 << L;
    int s = input(L, 1), e = input(L, 1), k1 = input(L, 1);  // :358-360
    int es, k1kis;
    { Gates g(1, L, 1); for (int i = 0; i < k; i++) for (u32 j = 0; j < SZ; j++) g.relay(0, j); es = insert(g.done()); }                          // :97-103
    { Gates g(1, L, 1); for (int i = 0; i < k; i++) for (u32 j = 0; j < SZ; j++) g.relay(0, j, gl_from_u64(p.raw.k0is[i])); k1kis = insert(g.done()); }  // :105-115
    connect(e, es); connect(k1, k1kis);
    std::vector<int> ais, r1is;
    for (int i = 0; i < k; i++) ais.push_back(input(L, 1));   // :122-124
    for (int i = 0; i < k; i++) r1is.push_back(input(L, 1));  // :126-128
    int r1iqis;
    { Gates g(k, L, 1); for (int i = 0; i < k; i++) for (u32 j = 0; j < SZ; j++) g.relay((u32)i, j, gl_from_u64(p.raw.qis[i])); r1iqis = insert(g.done()); }  // :130-141
    for (int i = 0; i < k; i++) connect(r1is[i], r1iqis);
    int r2is = input(P, k);  // :147
    const int r2l = P + p.log2k;
    std::vector<int> chunks;
    for (u64 st = 0; st < (1ULL << r2l); st += SZ) {  // :149-161
        Gates g(1, r2l, 1);
        u64 en = std::min<u64>(st + SZ, 1ULL << r2l);
        for (u64 j = st; j < en; j++) g.relay(0, (u32)j);
        for (u64 j = en - st; j < SZ; j++) g.zero();
        int node = insert(g.done());
        connect(r2is, node);
        chunks.push_back(node);
    }
    {   // lasso_inputs_batched :163-181
        std::vector<u64> bounds;
        for (int i = 0; i < k; i++) bounds.push_back(p.raw.r1_bounds[i]);
        for (size_t i = 0; i < chunks.size(); i++) bounds.push_back(p.raw.r2_bounds[0]);
        bounds.push_back(p.raw.s_bound); bounds.push_back(p.raw.e_bound); bounds.push_back(p.raw.k1_bound);
        Gates g((int)bounds.size(), L, 1);
        for (size_t i = 0; i < bounds.size(); i++) for (u32 j = 0; j < SZ; j++) g.relay((u32)i, j, 1, gl_from_u64(bounds[i]));
        c.lasso_in_id = insert(g.done());
    }
    { HNode n; n.kind = NK_LASSO; c.lasso_id = insert(n); }  // :182-210 (rows/lookups live in the LassoPlan)
    if ((int)c.nodes[c.lasso_in_id].log2_out() != lp.nu) throw Error("circuit: lasso input size != 2^nu");
    for (int i = 0; i < k; i++) connect(r1is[i], c.lasso_in_id);
    for (int ch : chunks) connect(ch, c.lasso_in_id);
    connect(s, c.lasso_in_id); connect(e, c.lasso_in_id); connect(k1, c.lasso_in_id);
    connect(c.lasso_in_id, c.lasso_id);
    int s_eval = fft(L, false);  // :224-225
    connect(s, s_eval);
    int s_eval_copy;
    { Gates g(1, L, 1); for (u32 j = 0; j < SZ; j++) g.relay(0, j); s_eval_copy = insert(g.done()); }  // :227-235
    connect(s_eval, s_eval_copy);
    int sai_par;
    { Gates g(k, L, 1); for (int i = 0; i < k; i++) for (u32 j = 0; j < SZ; j++) g.relay((u32)i, j); sai_par = insert(g.done()); }  // :237-243
    for (int i = 0; i < k; i++) {  // :245-260
        int ai_eval = fft(L, false);
        int sai_eval;
        { Gates g(2, L, 1); for (u32 j = 0; j < SZ; j++) g.mul(0, j, 1, j); sai_eval = insert(g.done()); }
        int sai = fft(L, true);
        connect(ais[i], ai_eval);
        connect(s_eval_copy, sai_eval); connect(ai_eval, sai_eval);
        connect(sai_eval, sai);
        connect(sai, sai_par);
    }
    int cyclo;
    {   // r2i_cyclo :262-278
        u32 r2sz = ((u32)1 << P) - 1;
        Gates g(1, P, k);
        for (u32 j = 0; j < r2sz; j++) g.relay(0, j);
        g.zero();
        for (u32 j = 0; j < r2sz; j++) g.relay(0, j);
        g.zero();
        cyclo = insert(g.done());
    }
    {   // sum :280-285
        Gates g(5, L, k);
        for (u32 j = 0; j < SZ; j++) { for (u32 i = 0; i < 5; i++) g.sum_term(i, j); g.end_sum(); }
        c.sum_id = insert(g.done());
    }
    connect(r2is, cyclo);
    connect(sai_par, c.sum_id); connect(es, c.sum_id); connect(k1kis, c.sum_id); connect(r1iqis, c.sum_id); connect(cyclo, c.sum_id);
    // node order: Kahn, smallest id first
    std::vector<int> indeg(c.nodes.size());
    for (size_t i = 0; i < c.nodes.size(); i++) indeg[i] = (int)c.nodes[i].preds.size();
    std::priority_queue<int, std::vector<int>, std::greater<int>> ready;
    for (size_t i = 0; i < c.nodes.size(); i++) if (!indeg[i]) ready.push((int)i);
    while (!ready.empty()) {
        int u = ready.top(); ready.pop();
        c.topo.push_back(u);
        for (int v : c.nodes[u].succs) if (--indeg[v] == 0) ready.push(v);
    }
    if (c.topo.size() != c.nodes.size()) throw Error("circuit: cycle");
    for (size_t i = 0; i < c.nodes.size(); i++) if (c.nodes[i].kind == NK_INPUT) c.input_ids.push_back((int)i);
    return c;
}

// ------------------------------------------------------------------------------------------------
u64 root_of_unity(int log2n) {
    u64 w = gl_pow(7, (GL_P - 1) >> 32);
    for (int i = log2n; i < 32; i++) w = gl_mul(w, w);
    return w;
}

void ntt_host(u64* a, int log2n, bool inverse) {
    // decimation in frequency (natural in, bit-reversed out) followed by the bit-reversal permutation
    const size_t N = (size_t)1 << log2n;
    u64 w = root_of_unity(log2n);
    if (inverse) w = gl_inv(w);
    std::vector<u64> tw(N / 2);
    tw[0] = 1;
    for (size_t i = 1; i < N / 2; i++) tw[i] = gl_mul(tw[i - 1], w);
    for (int s = log2n - 1; s >= 0; s--) {
        size_t h = (size_t)1 << s, step = N >> (s + 1);
        for (size_t base = 0; base < N; base += 2 * h)
            for (size_t j = 0; j < h; j++) {
                u64 x = a[base + j], y = a[base + j + h];
                a[base + j] = gl_add(x, y);
                a[base + j + h] = gl_mul(gl_sub(x, y), tw[j * step]);
            }
    }
    for (size_t i = 0; i < N; i++) {
        size_t r = 0;
        for (int b = 0; b < log2n; b++) r |= ((i >> b) & 1) << (log2n - 1 - b);
        if (r > i) std::swap(a[i], a[r]);
    }
    if (inverse) {
        u64 ninv = gl_inv(gl_from_u64(N));
        for (size_t i = 0; i < N; i++) a[i] = gl_mul(a[i], ninv);
    }
}

std::vector<std::vector<u64>> circuit_evaluate(const HCircuit& c, const Params& p, const Witness& w) {
    std::vector<std::vector<u64>> v(c.nodes.size());
    const size_t SZ = p.SZ();
    {   // inputs in NodeId order: s, e, k1, ais.., r1is.., r2is (chain_par! :408)
        size_t idx = 0;
        auto put = [&](const u64* src, size_t len) {
            int id = c.input_ids.at(idx++);
            if (len != ((size_t)1 << c.nodes[id].log2_size)) throw Error("circuit: input size mismatch");
            v[id].assign(src, src + len);
        };
        put(w.s.data(), SZ); put(w.e.data(), SZ); put(w.k1.data(), SZ);
        for (int i = 0; i < p.k; i++) put(&w.ais[i * SZ], SZ);
        for (int i = 0; i < p.k; i++) put(&w.r1is[i * SZ], SZ);
        put(w.r2is.data(), w.r2is.size());
        if (idx != c.input_ids.size()) throw Error("circuit: input count mismatch");
    }
    // level-synchronous evaluation so independent nodes (the 2k FFT chains) run in parallel
    std::vector<int> level(c.nodes.size(), 0);
    int maxl = 0;
    for (int id : c.topo) { for (int pr : c.nodes[id].preds) level[id] = std::max(level[id], level[pr] + 1); maxl = std::max(maxl, level[id]); }
    for (int l = 1; l <= maxl; l++) {
        std::vector<int> ids;
        for (int id : c.topo) if (level[id] == l) ids.push_back(id);
#pragma omp parallel for schedule(dynamic, 1) num_threads(hg_omp_threads())
        for (long long q = 0; q < (long long)ids.size(); q++) {
            int id = ids[q];
            const HNode& n = c.nodes[id];
            if (n.kind == NK_FFT) {
                v[id] = v[n.preds[0]];
                ntt_host(v[id].data(), n.log2_size, n.inverse);
            } else if (n.kind == NK_LASSO) {
                v[id].assign(1, 0);  // LassoNode::evaluate returns [0] (lasso.rs:53-55)
            } else if (n.kind == NK_VANILLA) {
                const size_t G = (size_t)1 << n.log2_sub_out, S = (size_t)1 << n.log2_sub_in, R = (size_t)1 << n.log2_reps;
                std::vector<u64>& o = v[id];
                o.assign(G * R, 0);
                for (size_t rep = 0; rep < R; rep++) {
                    u64* dst = o.data() + rep * G;
                    for (auto& t : n.w0) dst[t.gate] = gl_add(dst[t.gate], t.c);
                    for (auto& t : n.lin) {
                        u64 x = v[n.preds[t.in]][rep * S + t.j];
                        dst[t.gate] = gl_add(dst[t.gate], t.c == 1 ? x : gl_mul(t.c, x));
                    }
                    for (auto& t : n.mul)
                        dst[t.gate] = gl_add(dst[t.gate], gl_mul(t.c, gl_mul(v[n.preds[t.i0]][rep * S + t.j0], v[n.preds[t.i1]][rep * S + t.j1])));
                }
            }
        }
    }
    return v;
}

}  // namespace hg
