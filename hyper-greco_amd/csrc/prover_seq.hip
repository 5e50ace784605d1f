// Round-by-round prover for the protocol modes of SURVEY.md 8(f) f-4 (hg_prove_mode):
//   bit 0  absorbing transcript: write_felt also hashes the element (the in-tree plonkish-trait writer's rule,
//          transcript.rs:205-208, 224-233), so every challenge depends on every earlier prover message;
//   bit 1  extension-field memory checking: gamma, tau stay in E instead of being truncated to base limb 0
//          (lasso/src/memory_checking/prover.rs:36-39; README.md:108 "Known issues").
// Mode 0 (the reference as it is) never comes here: prover.hip enqueues that whole proof behind one synchronisation, because
// its challenges are known up front. With an absorbing transcript they are not, so this prover walks the protocol in
// transcript order and synchronises once per sum-check round:
//   pass 1  the round kernel with a placeholder challenge: its hypercube sums do not depend on the challenge;
//   host    round polynomial -> transcript (absorbed) -> squeeze r -> one 16-byte upload into the device challenge table;
//   pass 2  the same launch again: now the fold it writes is the real one.
// Every kernel is the product kernel of the fast path (kernels.hip), launched for one sum-check at a time; the only extra
// kernels are the Ext2 forms of the hash / product-tree steps that mode bit 1 needs. This is a measured mode, not a tuned
// one: DESIGN.md gives its time next to the headline and says what a tuned version would change (device-side Keccak, the
// fold of round i fused with the sums of round i+1).
#include <algorithm>
#include <chrono>
#include <cstring>
#include "prover.hpp"

namespace hg {
namespace {

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

const u64 INV2 = gl_inv(2), INV3 = gl_inv(3), INV6 = gl_inv(6);
void interpolate(const E2* ev, int d, E2* c) {  // evaluations at 0..d -> coefficients low..high (d = 2 or 3)
    E2 d1 = e2_sub(ev[1], ev[0]);
    E2 d2 = e2_add(e2_sub(ev[2], e2_dbl(ev[1])), ev[0]);
    if (d == 2) { c[0] = ev[0]; c[2] = e2_mul_f(d2, INV2); c[1] = e2_sub(d1, c[2]); return; }
    E2 d3 = e2_sub(e2_sub(ev[3], ev[0]), e2_mul_f(e2_sub(ev[2], ev[1]), 3));
    c[0] = ev[0];
    c[3] = e2_mul_f(d3, INV6);
    c[2] = e2_mul_f(e2_sub(d2, d3), INV2);
    c[1] = e2_add(e2_sub(d1, e2_mul_f(d2, INV2)), e2_mul_f(d3, INV3));
}
E2 horner(const E2* c, int d, E2 x) {
    E2 r = c[d];
    for (int i = d - 1; i >= 0; i--) r = e2_add(e2_mul(r, x), c[i]);
    return r;
}

// ---- Ext2 forms of the memory-checking steps (mode bit 1) -----------------------------------------------------------
// h = a + v gamma + t gamma^2 - tau with gamma, tau in E (a, v, t small integers)
__device__ __forceinline__ E2 hash_e(u64 a, u64 v, u64 t, E2 g, E2 g2, E2 tau) {
    return e2_sub(e2_add(e2_add_f(e2_mul_f(g, v), a), e2_mul_f(g2, t)), tau);
}
__global__ void k_hash_rw_e2(size_t n, const u64* __restrict__ dim, const u64* __restrict__ ts, const u64* __restrict__ ep, E2 g, E2 g2, E2 tau,
                             E2* __restrict__ rd, E2* __restrict__ wr) {
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        const E2 a = hash_e(dim[j], ep[j], ts[j], g, g2, tau);
        rd[j] = a;
        wr[j] = e2_add(a, g2);  // t + 1
    }
}
__global__ void k_hash_if_e2(u32 cutoff, const u64* __restrict__ fc, E2 g, E2 g2, E2 tau, E2* __restrict__ init, E2* __restrict__ fin) {
    const u32 a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= 65536) return;
    const u64 tv = a < cutoff ? (u64)a : 0;
    const E2 h0 = hash_e(a, tv, 0, g, g2, tau);
    init[a] = h0;
    fin[a] = e2_add(h0, e2_mul_f(g2, gl_from_u64(fc[a])));
}
// product tree level on Ext2 rows: out[b][i] = in[b][i] * in[b][i + h]
__global__ void k_prod_level_e2(const E2* __restrict__ in, size_t in_len, E2* __restrict__ out) {
    const size_t h = in_len >> 1;
    const E2* src = in + (size_t)blockIdx.y * in_len;
    E2* dst = out + (size_t)blockIdx.y * h;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < h; i += (size_t)gridDim.x * blockDim.x) dst[i] = e2_mul(src[i], src[i + h]);
}
__global__ void k_gp_top_e2(const E2* __restrict__ top, int nb, E2* __restrict__ roots, E2* __restrict__ evals) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const E2 l = top[2 * b], r = top[2 * b + 1];
    roots[b] = e2_mul(l, r);
    evals[2 * b] = l;
    evals[2 * b + 1] = r;
}

// ---- the transcript mailbox ------------------------------------------------------------------------------------------------
// With an absorbing transcript a round's challenge is a hash of the round's message, so the device must hand the round sums to the
// transcript and get the challenge back before it can fold. A stream synchronisation plus a challenge upload costs ~30 us per
// round; here the host never synchronises inside a sum-check: the round kernels write their sums into the host-mapped result
// buffer as always, a one-thread kernel (k_mail) then posts a sequence number into pinned host memory and SPINS on the host's
// answer, the host thread (which has meanwhile enqueued the next launches) spins on the posted number, runs the transcript step
// (interpolation, absorb, two Keccak permutations: ~2 us) and posts the challenge, which k_mail copies into the device
// challenge table - and, after a first round, multiplies into the job's first-round weights. One PCIe round trip, no launch gap:
// the kernels behind k_mail are already in the queue. The wait is bounded (a host that died cannot hang the device).
struct Mail {
    unsigned long long gpu_seq;   // written by the device: "results up to message gpu_seq are in the result buffer"
    unsigned long long cpu_seq;   // written by the host: "the answer to message cpu_seq is in chal[]"
    unsigned long long timeouts;  // device-side waits that gave up
    unsigned long long pad;
    E2 chal[4];
    unsigned long long dbg[8];    // -DHG_SEQ_STAMPS: device clock (100 MHz) differences, see k_sq_round
    // round sums of the fused round kernels: one 16-byte store per base-field word, the word and the message number side by side.
    // A slot is written by ONE store instruction of one lane and arrives as one write: the host that reads `tag == seq` reads the
    // value that came with it. No store has to be waited for before a separate "posted" word (1.4-2.4 us per round).
    struct Slot { unsigned long long v, tag; } slot[32];
};
__global__ void k_mail(Mail* m, unsigned long long seq, E2* chain_dst, dev::StJob* patch, int npw) {   // one wave
    __shared__ E2 s_r;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&m->gpu_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const long long t0 = wall_clock64();   // 100 MHz
        while (__hip_atomic_load(&m->cpu_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            if (wall_clock64() - t0 > 500000000ll) { __hip_atomic_fetch_add(&m->timeouts, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }   // 5 s
            __builtin_amdgcn_s_sleep(8);
        }
        E2 r;
        r.c0 = __hip_atomic_load(&m->chal[0].c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        r.c1 = __hip_atomic_load(&m->chal[0].c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        chain_dst[0] = r;
        s_r = r;
    }
    __syncthreads();
    if (patch && (int)threadIdx.x < npw) patch->pwr[threadIdx.x] = e2_mul(patch->pw[threadIdx.x], s_r);   // first-round fold weights pw[i] * r_0
}
// posts a sequence number only (the host waits for results, the device for nothing)
__global__ void k_post(Mail* m, unsigned long long seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(&m->gpu_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- round kernels of the sequential prover ---------------------------------------------------------------------------------------
// One launch per sum-check round, whatever the transcript: the kernel of round rd FOLDS the previous round's tables with r_(rd-1)
// on the fly (that challenge is in the device table by now), writes the folded tables once, accumulates round rd's sums on them,
// and its last-arriving workgroup hands the sums to the host transcript through the mailbox and waits for r_rd. Each table entry
// is read once and written once per round, as in the fast path; what the fast path cannot do here is run rounds of different
// sum-checks side by side or two rounds in one pass - every round's challenge depends on the round before.
//   tables: T_0 = the input (u64 or E2), T_rd[j] = T_(rd-1)[2j] + r_(rd-1) (T_(rd-1)[2j+1] - T_(rd-1)[2j]) (convention C3), natural order;
//   weights (gamma^b of a grand product's left tables, M^i of the collation tables) are multiplied in by the first fold, so that
//   later rounds need none; round 0 applies them to the products instead. Final left evaluations therefore carry their weight
//   (the host divides it out, as on the fast path).
constexpr int SQ_MAX_BLOCKS = 1024;
struct SqJob {
    int kind;               // 0 collation, 1 grand product, 2 sum of pair products
    int ntab, nvars, nv;    // nv = sums per round (t = 0, 2 [, 3])
    const void* in;         // kinds 0 / 1: table t at in + t * in_stride elements
    size_t in_stride;
    const void* tab[2 * dev::PS_MAX_PAIRS];   // kind 2: a_i = tab[2i] (input element type), b_i = tab[2i+1] (E2)
    E2* fin[2 * dev::PS_MAX_PAIRS];           // kind 2: where table t's final evaluation goes
    E2* final_out;          // kinds 0 / 1: ntab final evaluations
    E2* buf[2];             // T_rd (rd >= 1) = buf[rd & 1] + t * 2^(nvars - rd)
    size_t r_off, sums_slot;
    Mail* mail;
    unsigned long long seq0;   // round rd's message is seq0 + rd
    int dev_rounds;            // > 0: only rounds 0 .. dev_rounds-1 run on the device (the host finishes the sum-check, run_sq): the last
    int pad_;                  //      of them posts its sums and does not wait for the challenge
    E2 pw[dev::PW_MAX];
};
// what a round launch is told besides the job (kernel arguments: nothing here waits for a load of the job descriptor)
struct SqArgs {
    int rd, jb_log2;
    E2* chain_prev;                  // where r_(rd-1) goes in the device's challenge table (rd > 0)
    unsigned long long* ready;       // device word: the number of the last message whose answer is in the table
    Mail* mail;
    unsigned long long seq;          // this round's message number; the round before has seq - 1
    E2* partials;
    unsigned* ticket;
    const E2* src;                   // rd >= 2: T_(rd-1), table t at src + t * 2^(nvars-rd+1)
    E2* dst;                         // rd >= 1: T_rd
    int kind, ntab, nvars, last;     // (copies of the job's; last: the host takes over after this round, run_sq)
    int rank, world;                 // sharded form (prove_resident_mode_sharded): this rank evaluates the sums of the tiles t = rank (mod world)
};
template <typename T> __device__ __forceinline__ E2 sq_ld(const T* p, size_t i);
template <> __device__ __forceinline__ E2 sq_ld<u64>(const u64* p, size_t i) { return e2(p[i], 0); }
template <> __device__ __forceinline__ E2 sq_ld<E2>(const E2* p, size_t i) { return p[i]; }
__device__ __forceinline__ E2 sq_fold(E2 x, E2 y, E2 r) { return e2_add(x, e2_mul(r, e2_sub(y, x))); }
__device__ __forceinline__ E2 sq_fold_base(u64 x, u64 y, E2 r) { return e2_add_f(e2_mul_f(r, gl_sub(y, x)), x); }

// (X, Y) = entries 2j, 2j+1 of table `t` in round rd; rd >= 1: folded on the fly from the four entries 4j .. 4j+3 of the round before
// and stored. `wmul`: multiply by w (the first fold of a weighted table).
template <typename TIN>
__device__ __forceinline__ void sq_pair(const SqJob& J, const SqArgs& A, int t, size_t j, E2 r_prev, bool wmul, E2 w, E2& X, E2& Y) {
    const int rd = A.rd;
    const size_t len_prev = (size_t)1 << (A.nvars - rd + 1), len = len_prev >> 1;
    if (rd == 0) {
        const TIN* p = A.kind == 2 ? static_cast<const TIN*>(J.tab[t]) : static_cast<const TIN*>(J.in) + (size_t)t * J.in_stride;
        X = sq_ld<TIN>(p, 2 * j); Y = sq_ld<TIN>(p, 2 * j + 1);
        return;
    }
    if (rd == 1) {
        const TIN* p = A.kind == 2 ? static_cast<const TIN*>(J.tab[t]) : static_cast<const TIN*>(J.in) + (size_t)t * J.in_stride;
        if constexpr (sizeof(TIN) == 8) {
            X = sq_fold_base(p[4 * j], p[4 * j + 1], r_prev); Y = sq_fold_base(p[4 * j + 2], p[4 * j + 3], r_prev);
        } else {
            X = sq_fold(p[4 * j], p[4 * j + 1], r_prev); Y = sq_fold(p[4 * j + 2], p[4 * j + 3], r_prev);
        }
        if (wmul) { X = e2_mul(X, w); Y = e2_mul(Y, w); }
    } else {
        const E2* p = A.src + (size_t)t * len_prev;
        X = sq_fold(p[4 * j], p[4 * j + 1], r_prev); Y = sq_fold(p[4 * j + 2], p[4 * j + 3], r_prev);
    }
    E2* o = A.dst + (size_t)t * len;
    o[2 * j] = X; o[2 * j + 1] = Y;
}
// kind 2's b tables are E2 whatever the a tables are
__device__ __forceinline__ void sq_pair_b(const SqJob& J, const SqArgs& A, int t, size_t j, E2 r_prev, E2& X, E2& Y) {
    const int rd = A.rd;
    const size_t len_prev = (size_t)1 << (A.nvars - rd + 1), len = len_prev >> 1;
    if (rd == 0) { const E2* p = static_cast<const E2*>(J.tab[t]); X = p[2 * j]; Y = p[2 * j + 1]; return; }
    const E2* p = rd == 1 ? static_cast<const E2*>(J.tab[t]) : A.src + (size_t)t * len_prev;
    X = sq_fold(p[4 * j], p[4 * j + 1], r_prev); Y = sq_fold(p[4 * j + 2], p[4 * j + 3], r_prev);
    E2* o = A.dst + (size_t)t * len;
    o[2 * j] = X; o[2 * j + 1] = Y;
}

__device__ __forceinline__ E2 sq_shfl_xor(E2 v, int mask) {
    E2 o;
    o.c0 = (u64)__shfl_xor((unsigned long long)v.c0, mask, 64);
    o.c1 = (u64)__shfl_xor((unsigned long long)v.c1, mask, 64);
    return o;
}
// sum over the lanes of a wave whose index differs in the bits >= lo_bit (butterfly: every lane ends with the sum of its class)
template <int NV> __device__ __forceinline__ void sq_wave_sum(E2* acc, int lo_bit) {
    for (int m = 32; m >= (1 << lo_bit); m >>= 1) {
#pragma unroll
        for (int t = 0; t < NV; t++) acc[t] = e2_add(acc[t], sq_shfl_xor(acc[t], m));
    }
}
template <int NV> __device__ __forceinline__ void sq_block_sum(E2* acc, E2 (*sm)[256]) {   // -> thread 0 holds the sums
    // wave butterflies, one LDS hop across the four waves (a tree of eight barrier steps cost 2-3 us of a 10 us round)
    sq_wave_sum<NV>(acc, 0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < NV; t++) sm[t][wave] = acc[t];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int t = 0; t < NV; t++) acc[t] = e2_add(e2_add(sm[t][0], sm[t][1]), e2_add(sm[t][2], sm[t][3]));
    }
    __syncthreads();
}
__device__ __forceinline__ void sq_store_slot(Mail::Slot* s, u64 v, unsigned long long tag) {
#if defined(HG_STRICT_TICKETS) || (defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__))
    __hip_atomic_store(&s->v, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&s->tag, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 w = {(unsigned)v, (unsigned)(v >> 32), (unsigned)tag, (unsigned)(tag >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(s), "v"(w) : "memory");   // one write-through store to system scope
#endif
}
// the last-arriving workgroup's thread 0: the sums go to the host, tagged with the message number; nothing is waited for - the NEXT
// launch's first workgroup picks the answer up (sq_wait_challenge), so that its launch and its first loads overlap the round trip
// (one more slot carries the XOR of the words: the host takes a message only when it adds up - a slot that were ever delivered in two
// halves would read as "not there yet", not as a wrong sum)
__device__ __forceinline__ void sq_post_sums(const SqArgs& A, const E2* total, int nv) {
    u64 x = 0;
    for (int t = 0; t < nv; t++) {
        sq_store_slot(&A.mail->slot[2 * t], total[t].c0, A.seq);
        sq_store_slot(&A.mail->slot[2 * t + 1], total[t].c1, A.seq);
        x ^= total[t].c0 ^ total[t].c1;
    }
    sq_store_slot(&A.mail->slot[2 * nv], x, A.seq);
}
// r of the message `seq_prev`: workgroup 0's thread 0 takes it from the mailbox (waiting for the host if need be), writes it into the
// device's challenge table and raises `ready`; the other workgroups wait for `ready`. Returns r to every thread of the workgroup.
__device__ __forceinline__ E2 sq_wait_challenge(Mail* m, unsigned long long seq_prev, E2* chain_slot, unsigned long long* ready, E2* s_r) {
    if (threadIdx.x == 0) {
        E2 r;
        const long long t0 = wall_clock64();
        if (blockIdx.x == 0) {
            // pairs with the host's release store of cpu_seq. On gfx942 / gfx950 the in-order loads plus the control dependency order
            // the chal[] loads behind it; the memory-model form (any other target, HG_STRICT_TICKETS) acquires, as k_mail does
#if defined(HG_STRICT_TICKETS) || (defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__))
            while (__hip_atomic_load(&m->cpu_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq_prev) {
#else
            while (__hip_atomic_load(&m->cpu_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq_prev) {
#endif
                if (wall_clock64() - t0 > 500000000ll) { __hip_atomic_fetch_add(&m->timeouts, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }   // 5 s
                __builtin_amdgcn_s_sleep(2);
            }
            r.c0 = __hip_atomic_load(&m->chal[0].c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            r.c1 = __hip_atomic_load(&m->chal[0].c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&chain_slot->c0, r.c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&chain_slot->c1, r.c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if defined(HG_STRICT_TICKETS) || (defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__))
            __hip_atomic_store(ready, seq_prev, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this thread's two write-through stores are complete)
            __hip_atomic_store(ready, seq_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        } else {
#if defined(HG_STRICT_TICKETS) || (defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__))
            while (__hip_atomic_load(ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq_prev) {
#else
            while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq_prev) {
#endif
                if (wall_clock64() - t0 > 600000000ll) break;
                __builtin_amdgcn_s_sleep(8);
            }
            r.c0 = __hip_atomic_load(&chain_slot->c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r.c1 = __hip_atomic_load(&chain_slot->c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *s_r = r;
    }
    __syncthreads();
    return *s_r;
}

// the hypercube sums of round rd over this workgroup's share of the pair indices (tiles tile0, tile0 + step, ...), folding the round
// before on the fly (see above); acc[t] += ... for t = 0, 2 [, 3]
template <int KIND, typename TIN>
__device__ __forceinline__ void sq_round_sums(const SqJob& J, const SqArgs& A, E2 r_prev, size_t tile0, size_t tile_step, E2* acc, E2 (*sm)[256]) {
    constexpr int NV = KIND == 1 ? 3 : 2;
    const int rd = A.rd, jb_log2 = A.jb_log2;
    const int hl = A.nvars - 1 - rd;                 // log2 of this round's pair count
    const size_t h = (size_t)1 << hl;
    // 2^jb_log2 of the 256 threads run along the pair index j, the other 256 / 2^jb_log2 groups split the tables (host: sq_plan)
    const int JB = 1 << jb_log2, G = 256 >> jb_log2;
    const int jj = threadIdx.x & (JB - 1), g = threadIdx.x >> jb_log2;
    const size_t ntiles = h >> jb_log2;
    const int units = KIND == 0 ? A.ntab : A.ntab / 2;
    for (size_t tile = tile0; tile < ntiles; tile += tile_step) {
        const size_t j = (tile << jb_log2) + jj;
        // sharded form: every rank folds every tile (it needs the whole table of the next round), the hypercube sums of a tile are
        // evaluated by ONE rank; the ranks' partial sums meet in the round's all-reduce on the host side (answer_round)
        const bool own = A.world <= 1 || (int)(tile % (size_t)A.world) == A.rank;
        if (!own) {
            if (rd == 0) continue;   // nothing to fold yet
            for (int u = g; u < units; u += G) {
                const E2 w = (KIND == 2 || rd > 1) ? e2_one() : J.pw[u];
                E2 X, Y;
                if (KIND == 0) sq_pair<TIN>(J, A, u, j, r_prev, rd == 1 && u > 0, w, X, Y);
                else if (KIND == 1) { sq_pair<TIN>(J, A, 2 * u, j, r_prev, rd == 1 && u > 0, w, X, Y); sq_pair<TIN>(J, A, 2 * u + 1, j, r_prev, false, w, X, Y); }
                else { sq_pair<TIN>(J, A, 2 * u, j, r_prev, false, w, X, Y); sq_pair_b(J, A, 2 * u + 1, j, r_prev, X, Y); }
            }
            continue;
        }
        E2 s[NV], q[NV];
#pragma unroll
        for (int t = 0; t < NV; t++) { s[t] = e2_zero(); q[t] = e2_zero(); }
        for (int u = g; u < units; u += G) {
            const E2 w = (KIND == 2 || rd > 1) ? e2_one() : J.pw[u];   // (weights: round 0's products, round 1's first fold)
            if (KIND == 0) {
                E2 X, Y;
                sq_pair<TIN>(J, A, u, j, r_prev, rd == 1 && u > 0, w, X, Y);
                const E2 v2 = e2_sub(e2_dbl(Y), X);
                if (u == 0) { q[0] = X; q[1] = v2; }
                if (rd == 0) { s[0] = e2_add(s[0], e2_mul_f(X, w.c0)); s[1] = e2_add(s[1], e2_mul_f(v2, w.c0)); }   // M^i is a base-field constant
                else { s[0] = e2_add(s[0], X); s[1] = e2_add(s[1], v2); }
            } else if (KIND == 1) {
                E2 lx, ly, rx, ry;
                sq_pair<TIN>(J, A, 2 * u, j, r_prev, rd == 1 && u > 0, w, lx, ly);
                sq_pair<TIN>(J, A, 2 * u + 1, j, r_prev, false, w, rx, ry);
                const E2 dl = e2_sub(ly, lx), dr = e2_sub(ry, rx);
                const E2 l2 = e2_add(ly, dl), r2 = e2_add(ry, dr), l3 = e2_add(l2, dl), r3 = e2_add(r2, dr);
                if (u == 0) { q[0] = lx; q[1] = l2; q[NV - 1] = l3; }
                E2 p0 = e2_mul(lx, rx), p2 = e2_mul(l2, r2), p3 = e2_mul(l3, r3);
                if (rd == 0 && u > 0) { p0 = e2_mul(p0, w); p2 = e2_mul(p2, w); p3 = e2_mul(p3, w); }
                s[0] = e2_add(s[0], p0); s[1] = e2_add(s[1], p2); s[NV - 1] = e2_add(s[NV - 1], p3);
            } else {
                E2 ax, ay, bx, by;
                sq_pair<TIN>(J, A, 2 * u, j, r_prev, false, w, ax, ay);
                sq_pair_b(J, A, 2 * u + 1, j, r_prev, bx, by);
                const E2 a2 = e2_sub(e2_dbl(ay), ax), b2 = e2_sub(e2_dbl(by), bx);
                s[0] = e2_add(s[0], e2_mul(ax, bx)); s[1] = e2_add(s[1], e2_mul(a2, b2));
            }
        }
        if (KIND == 2) {
#pragma unroll
            for (int t = 0; t < NV; t++) acc[t] = e2_add(acc[t], s[t]);
        } else {
            // the groups' shares of sum_i meet BEFORE the multiplication by p_0 (g = p_0 * sum_i ...); group 0 holds p_0's values
            if (G > 1) {
                // thread index = g * JB + jj: inside a wave the groups sit in the lane bits above log2(JB) (butterfly), across the
                // waves one LDS hop; group 0 (which holds p_0's values) ends up with the sum over all groups
                if (jb_log2 < 6) sq_wave_sum<NV>(s, jb_log2);
                const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
                const bool rep = jb_log2 < 6 ? lane < JB : true;   // one lane per (wave, jj) carries the wave's share
                if (rep) {
#pragma unroll
                    for (int t = 0; t < NV; t++) sm[t][jb_log2 < 6 ? wave * 64 + lane : threadIdx.x] = s[t];
                }
                __syncthreads();
                if (g == 0) {
                    if (jb_log2 < 6) {   // g == 0: wave 0, lanes jj < JB: add the other waves' lanes jj
#pragma unroll
                        for (int t = 0; t < NV; t++) s[t] = e2_add(e2_add(sm[t][jj], sm[t][64 + jj]), e2_add(sm[t][128 + jj], sm[t][192 + jj]));
                    } else {             // groups are whole waves (JB = 64, 128): add the other groups' threads jj
                        for (int gg = 1; gg < G; gg++) {
#pragma unroll
                            for (int t = 0; t < NV; t++) s[t] = e2_add(s[t], sm[t][gg * JB + jj]);
                        }
                    }
                }
                __syncthreads();
            }
            if (g == 0) {
#pragma unroll
                for (int t = 0; t < NV; t++) acc[t] = e2_add(acc[t], e2_mul(q[t], s[t]));
            }
        }
    }
}

// A short round is a chain of dependent latencies (measured with -DHG_SEQ_STAMPS when the kernel still waited for its own answer:
// kernel start -> sums 9-12 us with the job descriptor, the challenge and the table entries loaded one behind the other, sums ->
// posted 1.4-2.4 us, posted -> answered 3.4 us). Now: addresses come with the kernel arguments, the entries of the first tile are
// touched BEFORE the wait for the challenge of the round before (which overlaps the launch as well), the sums leave in tagged
// stores that nothing waits for. (Passing the whole 2.2 KB job by value was measured earlier: slower, 13.6 us against 12.2.)
template <int KIND, typename TIN>
__global__ __launch_bounds__(256) void k_sq_round(const SqJob* __restrict__ jp, const SqArgs A) {
    constexpr int NV = KIND == 1 ? 3 : 2;
    __shared__ E2 sm[NV][256];
    __shared__ unsigned s_last;
    __shared__ E2 s_r;
    const SqJob& J = *jp;
    const int rd = A.rd;
#ifdef HG_SEQ_STAMPS
    const long long t_start = wall_clock64();
#endif
    E2 r_prev = e2_zero();
    if (rd > 0) {
        if (rd >= 2) {   // warm this workgroup's first tile while the answer is on its way (the loads are repeated below, from the caches)
            const int hl = A.nvars - 1 - rd, JB = 1 << A.jb_log2, G = 256 >> A.jb_log2;
            const int jj = threadIdx.x & (JB - 1), g = threadIdx.x >> A.jb_log2;
            const size_t ntiles = ((size_t)1 << hl) >> A.jb_log2, len_prev = (size_t)1 << (A.nvars - rd + 1);
            if (blockIdx.x < ntiles) {
                const size_t j = ((size_t)blockIdx.x << A.jb_log2) + jj;
                u64 touch = 0;
                const int step = KIND == 0 ? 1 : 2;
                for (int t = g * step; t < A.ntab; t += G * step) {
                    touch ^= A.src[(size_t)t * len_prev + 4 * j].c0;
                    if (step == 2) touch ^= A.src[(size_t)(t + 1) * len_prev + 4 * j].c0;
                }
                asm volatile("" ::"v"(touch));
            }
        }
        r_prev = sq_wait_challenge(A.mail, A.seq - 1, A.chain_prev, A.ready, &s_r);
    }
#ifdef HG_SEQ_STAMPS
    const long long t_r = wall_clock64();
#endif
    E2 acc[NV];
#pragma unroll
    for (int t = 0; t < NV; t++) acc[t] = e2_zero();
    sq_round_sums<KIND, TIN>(J, A, r_prev, blockIdx.x, gridDim.x, acc, sm);
    sq_block_sum<NV>(acc, sm);
    const int nblocks = (int)gridDim.x;
    if (nblocks > 1) {
        if (threadIdx.x == 0) {
            for (int t = 0; t < NV; t++) {
                __hip_atomic_store(&A.partials[(size_t)blockIdx.x * NV + t].c0, acc[t].c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&A.partials[(size_t)blockIdx.x * NV + t].c1, acc[t].c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // (the partials were written through with agent-scope atomic stores by THIS thread: their completion - vmcnt - is all the
            // ticket must be ordered after; a formal release would write back every dirty line of the L2, i.e. the folded tables this
            // kernel streams out. gfx942 / gfx950 only: kernels.hip, finish_partials)
#if defined(HG_STRICT_TICKETS) || (defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__))
            const unsigned tk = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned tk = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            s_last = tk == (unsigned)nblocks - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (!s_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int t = 0; t < NV; t++) acc[t] = e2_zero();
        for (int b = threadIdx.x; b < nblocks; b += 256)
            for (int t = 0; t < NV; t++) {
                E2 v;
                v.c0 = __hip_atomic_load(&A.partials[(size_t)b * NV + t].c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v.c1 = __hip_atomic_load(&A.partials[(size_t)b * NV + t].c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc[t] = e2_add(acc[t], v);
            }
        sq_block_sum<NV>(acc, sm);
        if (threadIdx.x == 0) __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
#ifdef HG_SEQ_STAMPS
        const long long t_sum = wall_clock64();
#endif
        sq_post_sums(A, acc, NV);
#ifdef HG_SEQ_STAMPS
        Mail* m = A.mail;   // (statistics of the posting workgroup: start -> challenge in hand, -> sums, -> stores issued)
        m->dbg[0] += (unsigned long long)(t_r - t_start); m->dbg[1] += (unsigned long long)(t_sum - t_r); m->dbg[2] += (unsigned long long)(wall_clock64() - t_sum); m->dbg[3] += 1;
#endif
    }
}
// thread mapping of a round with 2^hl pair indices and `units` table units (pairs / tables): as many groups along the units as
// they can use (one unit per thread when the round is small: a workgroup that walks 50 pairs per thread is a 50 us latency
// chain), fewer when the round has enough pair indices to fill the chip anyway (long coalesced runs along j); host and device
static inline void sq_plan(int hl, int units, int nblocks_max, int* jb_log2, int* nblk) {
    int g_log2 = 0;
    while ((1 << g_log2) < units && g_log2 < 8) g_log2++;
    while (g_log2 > 0 && hl + g_log2 > 18) g_log2--;
    int jb = 8 - g_log2;
    if (jb > hl) jb = hl;
    *jb_log2 = jb;
    const size_t ntiles = ((size_t)1 << hl) >> jb;
    *nblk = (int)(ntiles < (size_t)nblocks_max ? ntiles : (size_t)nblocks_max);
}
// after the last round: table t's final evaluation = fold of its last two entries with the last challenge (weights: see above)
template <typename TIN>
__global__ void k_sq_final(const SqJob* __restrict__ jp, const SqArgs A) {   // A: seq = the last round's + 1, chain_prev = the last challenge's place
    __shared__ E2 s_r;
    const SqJob& J = *jp;
    const E2 r = sq_wait_challenge(A.mail, A.seq - 1, A.chain_prev, A.ready, &s_r);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= J.ntab) return;
    const int n = J.nvars;
    E2 v;
    if (n == 1) {
        if (J.kind == 2 && (t & 1)) { const E2* p = static_cast<const E2*>(J.tab[t]); v = sq_fold(p[0], p[1], r); }
        else {
            const TIN* p = J.kind == 2 ? static_cast<const TIN*>(J.tab[t]) : static_cast<const TIN*>(J.in) + (size_t)t * J.in_stride;
            v = sq_fold(sq_ld<TIN>(p, 0), sq_ld<TIN>(p, 1), r);
        }
        if (J.kind == 0 && t > 0) v = e2_mul(v, J.pw[t]);
        if (J.kind == 1 && !(t & 1) && t > 0) v = e2_mul(v, J.pw[t / 2]);
    } else {
        const E2* p = J.buf[(n - 1) & 1] + (size_t)t * 2;
        v = sq_fold(p[0], p[1], r);
    }
    if (J.kind == 2) *J.fin[t] = v; else J.final_out[t] = v;
}

// hands the tables of the last device round (T_(R-1): ntab runs of 2^(nvars-R+1) entries, contiguous) to the host: copies them into
// pinned host memory and posts `seq`. One workgroup; runs behind round R-1's kernel, i.e. after every workgroup of it has stored its
// share of the tables (kernel boundary).
__global__ __launch_bounds__(256) void k_sq_export(const E2* __restrict__ src, unsigned n, E2* dst_host, Mail* m, unsigned long long seq) {
    // 16-byte stores, consecutive lanes on consecutive entries: the host memory is uncached on the device side, the stores leave as
    // full write bursts (8-byte system-scope atomic stores went out one by one: 250 us for 32 KB)
    for (unsigned i = threadIdx.x; i < n; i += 256) dst_host[i] = src[i];
    // plain stores may sit in this XCD's L2: every thread releases them to system scope (write-back + wait; the L2 holds little
    // else right behind a kernel boundary), the barrier collects the threads, then the post
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&m->gpu_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// what the host computed for the device after a host tail, in one launch (kernel arguments carry the values: no copy engine in
// the stream): the challenges the device has not seen, and final evaluations whose destination is device memory
struct SqInstall {
    E2* chain_dst;
    int nchain, nres;
    E2 chain[24];
    E2* res_dst[2 * dev::PS_MAX_PAIRS];
    E2 res[2 * dev::PS_MAX_PAIRS];
};
__global__ void k_sq_install(SqInstall a) {
    const int i = threadIdx.x;
    if (i < a.nchain) a.chain_dst[i] = a.chain[i];
    if (i < a.nres) *a.res_dst[i] = a.res[i];
}

struct Claim {  // evaluation claim whose point is a run of this proof's challenge table
    size_t point_off;
    int len;
    E2 value;
};

struct SeqProver {
    hg_ctx* ctx;
    const hg_pk* pk;
    int mode;
    hipStream_t st;
    FsTranscript tr;
    std::vector<E2> chain;  // challenges squeezed so far (host copy)
    E2* d_chain = nullptr;  // the same in HBM: kernels take challenge positions, exactly like the fast path
    size_t chain_cap = 0;
    size_t res_used = 0;
    size_t n_sync = 0;   // stream synchronisations
    size_t n_drained = 0;
    size_t n_mail = 0;   // mailbox round trips (no synchronisation: the device spins on the host's answer)
    Mail* mail = nullptr;               // pinned host memory, mapped into the device
    static constexpr size_t MAIL_BYTES = 4096, HOST_TAIL_CAP = 8192;
    E2* h_tail = nullptr;               // (same allocation)
    size_t host_tail_entries = 0;       // a sum-check's last rounds run on the host once all its tables together have at most this many entries (HG_SEQ_HOST_TAIL, 0 = never)
    size_t n_host_rounds = 0;
    unsigned long long mail_seq = 0;
    bool use_mail = true;
    std::vector<const u64*> d_vals;
    std::vector<std::vector<Claim>> claims;
    // sharded form: one all-reduce per sum-check round (the form that stays available for an absorbing transcript).
    // `reduce` adds n canonical Goldilocks words lane-wise mod p over the ranks, in place, and returns 0.
    int shard_rank = 0, shard_world = 1;
    int (*reduce)(void*, uint64_t*, size_t) = nullptr;
    void* reduce_user = nullptr;
    size_t n_reduce = 0;
    void reduce_sums(E2* sums, int nv) {
        if (shard_world <= 1) return;
        uint64_t w[6];
        for (int t = 0; t < nv; t++) { w[2 * t] = sums[t].c0; w[2 * t + 1] = sums[t].c1; }
        if (reduce(reduce_user, w, (size_t)2 * nv) != 0) throw Error("sharded prove: the round's all-reduce failed");
        for (int t = 0; t < nv; t++) {
            if (w[2 * t] >= GL_P || w[2 * t + 1] >= GL_P) throw Error("sharded prove: the all-reduce returned a non-canonical word");
            sums[t] = e2(w[2 * t], w[2 * t + 1]);
        }
        n_reduce++;
    }

    // ---- node ownership (round 6; SURVEY 8(e): "one CRT modulus per GPU, allreduce per round") -------------------------------------------
    // own_mode: the values object is a rank's share (hg_witness_gen_shard): it holds the Lasso node's input and the inputs of the node
    // reductions this rank owns (shard_plan: node chains dealt by modulus, as in mode 0). A Vanilla / FFT node's reduction then runs on
    // its owner alone - every tile of every round kernel, the rounds it finishes on the host - and the other ranks ("ghosts") launch
    // nothing for it: they take part in the same all-reduces with zeros, absorb the same messages and squeeze the same challenges. The
    // Lasso node keeps the tile-split form (every rank holds its input).
    bool own_mode = false;
    ShardPlan plan;
    int cur_owner = -1;   // the owner of the node reduction being walked; -1: every rank works (tile split)
    bool ghost() const { return cur_owner >= 0 && cur_owner != shard_rank; }
    bool solo() const { return cur_owner >= 0 && cur_owner == shard_rank; }
    struct Owned {        // scope of one node reduction
        SeqProver* P;
        Owned(SeqProver* p, int owner) : P(p) { P->cur_owner = P->own_mode ? owner : -1; }
        ~Owned() { P->cur_owner = -1; }
    };
    // values that exist on one rank only (`have`) -> every rank: one all-reduce in which the others add zeros
    void share(E2* v, size_t n, bool have) {
        if (shard_world <= 1 || n == 0) return;
        std::vector<uint64_t> w(2 * n, 0);
        if (have) for (size_t i = 0; i < n; i++) { w[2 * i] = v[i].c0; w[2 * i + 1] = v[i].c1; }
        if (reduce(reduce_user, w.data(), w.size()) != 0) throw Error("sharded prove: an all-reduce failed");
        for (size_t i = 0; i < n; i++) {
            if (w[2 * i] >= GL_P || w[2 * i + 1] >= GL_P) throw Error("sharded prove: the all-reduce returned a non-canonical word");
            v[i] = e2(w[2 * i], w[2 * i + 1]);
        }
        n_reduce++;
    }
    void share_slot_list(const std::vector<size_t>& slots) {   // result slots of the current node: the owner's values to every rank
        if (cur_owner < 0) return;
        std::vector<E2> t(slots.size());
        for (size_t i = 0; i < slots.size(); i++) t[i] = ctx->h_res[slots[i]];
        share(t.data(), t.size(), solo());
        for (size_t i = 0; i < slots.size(); i++) ctx->h_res[slots[i]] = t[i];
    }
    // challenges a ghost squeezed for a sum-check it did not run, into its own device table (later nodes it owns are opened at them)
    void install_chain(size_t first, size_t cnt) {
        while (cnt) {
            const size_t take = std::min<size_t>(cnt, 24);
            SqInstall in;
            memset(&in, 0, sizeof(in));
            in.chain_dst = d_chain + first;
            in.nchain = (int)take;
            for (size_t i = 0; i < take; i++) in.chain[i] = chain[first + i];
            k_sq_install<<<1, 64, 0, st>>>(in);
            first += take; cnt -= take;
        }
    }

    SeqProver(hg_ctx* c, const hg_pk* k, int m) : ctx(c), pk(k), mode(m), st(c->stream) {
        tr.absorb = (m & 1) != 0;
        chain_cap = 1 << 15;
        d_chain = ctx->alloc_n<E2>(chain_cap);
        hip_check(hipMemsetAsync(d_chain, 0, chain_cap * sizeof(E2), st), "clear challenge table");
        for (E2* pbuf : {ctx->d_partials, ctx->d_partials2})
            hip_check(hipMemsetAsync(reinterpret_cast<char*>(pbuf) + dev::PARTIALS_E2 * sizeof(E2), 0, dev::PARTIALS_TICKETS * sizeof(unsigned), st), "clear reduction tickets");
        static const bool no_mail = hg_env_on("HG_SEQ_NO_MAIL");
        use_mail = !no_mail && ctx->d_res == ctx->h_res;   // (the sums must land in host memory without a copy)
        if (use_mail) {
            // the mailbox and, behind it, the buffer through which the last tables of a sum-check travel to the host (host_tail)
            // (kept by the context: pinning memory costs more than a dozen rounds)
            const size_t need = MAIL_BYTES + HOST_TAIL_CAP * sizeof(E2);
            if (!ctx->h_mailbox || ctx->mailbox_bytes < need) {
                if (ctx->h_mailbox) { (void)hipHostFree(ctx->h_mailbox); ctx->h_mailbox = nullptr; }
                hip_check(hipHostMalloc(&ctx->h_mailbox, need, hipHostMallocDefault), "hipHostMalloc(mailbox)");
                ctx->mailbox_bytes = need;
            }
            mail = static_cast<Mail*>(ctx->h_mailbox);
            memset(mail, 0, sizeof(Mail));
            h_tail = reinterpret_cast<E2*>(reinterpret_cast<char*>(mail) + MAIL_BYTES);
            const char* e = getenv("HG_SEQ_HOST_TAIL");
            host_tail_entries = e && *e ? (size_t)atol(e) : 1024;   // (c3, mode 3: 39.4 ms without, 38.8 at 512, 37.8 at 1024, 40.5 at 2048)
            if (host_tail_entries > HOST_TAIL_CAP) host_tail_entries = HOST_TAIL_CAP;
        }
    }
    ~SeqProver() {
        if (mail) {   // whatever still waits on the device is released before the mailbox goes away
            __atomic_store_n(&mail->cpu_seq, ~0ull, __ATOMIC_RELEASE);
            (void)hipStreamSynchronize(st);
        }
    }
    // host side of the mailbox: wait until the device has posted `seq` (results of everything enqueued before are in h_res)
    double t_wait_rounds = 0, t_wait_results = 0, t_enqueue_rounds = 0;   // HG_SEQ_TIMES=1: where the host's time goes
    double t_kind[3][2] = {{0, 0}, {0, 0}, {0, 0}};   // ... waiting for round sums, by sum-check kind and single- / multi-workgroup round
    size_t n_kind[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    void mail_wait(unsigned long long seq) {
        const double t0 = now_ms();
        unsigned spins = 0;
        while (__atomic_load_n(&mail->gpu_seq, __ATOMIC_ACQUIRE) < seq) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFF) == 0) {
                hip_check(hipGetLastError(), "round (mailbox)");
                if (hipStreamQuery(st) == hipSuccess && __atomic_load_n(&mail->gpu_seq, __ATOMIC_ACQUIRE) < seq) throw Error("mailbox: the stream drained without posting");
                if (now_ms() - t0 > 10000.0) throw Error("mailbox: no answer from the device within 10 s");
            }
        }
        n_mail++;
        t_wait_rounds += now_ms() - t0;
        if (slow_log && now_ms() - t0 > 3.0) fprintf(stderr, "[hg] slow: waited %.2f ms for message %llu (%s)\n", now_ms() - t0, seq, where);
    }
    const char* where = "";
    bool slow_log = hg_times("seq");
    struct Slow {   // reports a host-side step that took more than 3 ms (HG_SEQ_TIMES=1)
        SeqProver* P; const char* what; double t0;
        Slow(SeqProver* p, const char* w) : P(p), what(w), t0(now_ms()) {}
        ~Slow() { if (P->slow_log && now_ms() - t0 > 3.0) fprintf(stderr, "[hg] slow: %s took %.2f ms\n", what, now_ms() - t0); }
    };
    // "everything enqueued so far is done and its results are in h_res": a mailbox post when available, else a synchronisation
    void wait_results() {
        if (!use_mail) { sync(); return; }
        const unsigned long long seq = ++mail_seq;
        k_post<<<1, 64, 0, st>>>(mail, seq);
        const double t0 = now_ms(), r0 = t_wait_rounds;
        mail_wait(seq);
        // the stream is drained here: let the runtime retire its completed commands now, a few at a time (this mode never
        // synchronises; left alone the runtime reaped thousands of them at once, somewhere inside a launch: stalls of 30-40 ms)
        static const int sync_every = [] { const char* e = getenv("HG_SEQ_SYNC_EVERY"); return e && *e ? atoi(e) : 4; }();
        if (sync_every > 0 && (++n_drained % sync_every) == 0) { hip_check(hipStreamSynchronize(st), "drain"); n_sync++; }
        else (void)hipStreamQuery(st);
        t_wait_rounds = r0;
        t_wait_results += now_ms() - t0;
    }
    // the final evaluations of the sum-check just run: already in the result buffer when the host finished it (host_tail)
    bool sq_results_on_host = false;
    void wait_sumcheck_results() {
        if (sq_results_on_host) { sq_results_on_host = false; return; }
        wait_results();
    }
    bool ext_mc() const { return (mode & 2) != 0; }
    // HG_SEQ_CLASSIC=1: the fast path's round kernels run twice per round (sums, then the fold once the challenge is known)
    static bool classic() { static const bool c = hg_env_on("HG_SEQ_CLASSIC"); return c; }
    E2* d_res() { return ctx->d_res; }
    const E2* h_res() { return ctx->h_res; }
    size_t slot(size_t n) {
        if (res_used + n > ctx->res_cap) throw Error("result buffer exhausted");
        size_t s = res_used;
        res_used += n;
        return s;
    }
    size_t epos() const { return chain.size(); }
    void sync() {
        hip_check(hipStreamSynchronize(st), "round synchronisation");
        n_sync++;
    }
    // squeeze_challenge (transcript.rs:146-157): the value goes into the host chain and into the device table
    E2 squeeze(bool upload_now = true) {
        if (chain.size() >= chain_cap) throw Error("challenge table exhausted");
        const E2 r = tr.squeeze();
        chain.push_back(r);
        if (upload_now) {   // (a round's challenge travels through the mailbox instead: mail_round)
            E2* h = static_cast<E2*>(stage(&r, sizeof(E2)));
            hip_check(hipMemcpyAsync(d_chain + chain.size() - 1, h, sizeof(E2), hipMemcpyHostToDevice, st), "upload challenge");
        }
        return r;
    }
    void* stage(const void* src, size_t bytes) {
        size_t need = (bytes + 63) & ~(size_t)63;
        if (ctx->stage_used + need > ctx->stage_cap) {  // a proof in this mode uploads thousands of tiny descriptors: recycle
            wait_results();
            ctx->stage_used = 0;
        }
        void* p = ctx->h_stage + ctx->stage_used;
        ctx->stage_used += need;
        memcpy(p, src, bytes);
        return p;
    }
    template <typename T> T* upload(const T* src, size_t n) {
        T* d = ctx->alloc_n<T>(n ? n : 1);
        if (n) hip_check(hipMemcpyAsync(d, stage(src, n * sizeof(T)), n * sizeof(T), hipMemcpyHostToDevice, st), "upload descriptor");
        return d;
    }
    void write_slots(size_t s, size_t n) { for (size_t i = 0; i < n; i++) tr.write_e(h_res()[s + i]); }

    // transcript side of one round: d+1 coefficients, eval(1) derived from the running claim (C1), then the challenge
    E2 round_message(const E2* sums, int deg, E2& claim, bool upload_now = true) {
        E2 ev[4], c[4];
        ev[0] = sums[0];
        ev[1] = e2_sub(claim, sums[0]);
        ev[2] = sums[1];
        if (deg == 3) ev[3] = sums[2];
        interpolate(ev, deg, c);
        for (int k = 0; k <= deg; k++) tr.write_e(c[k]);
        const E2 r = squeeze(upload_now);
        claim = horner(c, deg, r);
        return r;
    }
    // One round through the mailbox. Called right after the round's first pass has been enqueued: enqueues k_mail (post, wait, copy the
    // challenge into the device table, patch the first-round weights of `patch`), lets the caller enqueue the second pass BEHIND it,
    // then waits for the sums, runs the transcript step on the host and answers.
    // The two halves of a round trip. A whole sum-check is ENQUEUED first (per round: first pass, k_mail, second pass - no launch
    // parameter depends on a challenge, the kernels read it from the device table), then the host answers the rounds one by one:
    // the device never waits for a launch, only for the answers.
    unsigned long long mail_enqueue(size_t chain_pos, dev::StJob* patch, int npw) {
        const unsigned long long seq = ++mail_seq;
        k_mail<<<1, 64, 0, st>>>(mail, seq, d_chain + chain_pos, patch, npw);
        return seq;
    }
    double t_answer = 0;
    E2 mail_answer(unsigned long long seq, const E2* sums, int deg, E2& claim) {
        mail_wait(seq);
        const double t0 = slow_log ? now_ms() : 0;
        const E2 r = round_message(sums, deg, claim, false);
        mail->chal[0] = r;
        __atomic_store_n(&mail->cpu_seq, seq, __ATOMIC_RELEASE);
        if (slow_log) t_answer += now_ms() - t0;
        return r;
    }

    // ---- one sum-check through the fused round kernels and the mailbox ---------------------------------------------------------
    unsigned* d_ticket = nullptr;
    SqArgs round_args(const SqJob& J, int rd, int jb_log2) {
        SqArgs A;
        memset(&A, 0, sizeof(A));
        A.rd = rd; A.jb_log2 = jb_log2;
        A.chain_prev = rd > 0 ? d_chain + J.r_off + rd - 1 : nullptr;
        A.ready = reinterpret_cast<unsigned long long*>(d_ticket + 2);
        A.mail = mail;
        A.seq = J.seq0 + (unsigned long long)rd;
        A.partials = ctx->d_partials;
        A.ticket = d_ticket;
        A.src = rd >= 2 ? J.buf[(rd - 1) & 1] : nullptr;
        A.dst = rd >= 1 ? J.buf[rd & 1] : nullptr;
        A.kind = J.kind; A.ntab = J.ntab; A.nvars = J.nvars;
        A.last = J.dev_rounds && rd == J.dev_rounds - 1;
        A.rank = cur_owner >= 0 ? 0 : shard_rank; A.world = cur_owner >= 0 ? 1 : shard_world;   // (a node's owner evaluates every tile itself)
        return A;
    }
    template <int KIND> void launch_round(bool in_base, const SqJob& J, const SqJob* d_job, int rd, int jb_log2, int grid) {
        const SqArgs A = round_args(J, rd, jb_log2);
        if (in_base) k_sq_round<KIND, u64><<<grid, 256, 0, st>>>(d_job, A);
        else k_sq_round<KIND, E2><<<grid, 256, 0, st>>>(d_job, A);
    }
    // host side of a fused round: the sums arrive in the mailbox's tagged slots
    bool slots_posted(unsigned long long seq, int nslots) const {   // nslots words + their XOR
        for (int k = nslots; k >= 0; k--)
            if (__atomic_load_n(&mail->slot[k].tag, __ATOMIC_ACQUIRE) != seq) return false;
        unsigned long long x = 0;
        for (int k = 0; k < nslots; k++) x ^= __atomic_load_n(&mail->slot[k].v, __ATOMIC_RELAXED);
        return x == __atomic_load_n(&mail->slot[nslots].v, __ATOMIC_RELAXED);
    }
    E2 answer_round(unsigned long long seq, int nv, int deg, E2& claim) {
        const double t0 = now_ms();
        unsigned spins = 0;
        while (!slots_posted(seq, 2 * nv)) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFF) == 0) {
                hip_check(hipGetLastError(), "round (mailbox)");
                if (hipStreamQuery(st) == hipSuccess && !slots_posted(seq, 2 * nv)) throw Error("mailbox: the stream drained without posting");
                if (now_ms() - t0 > 10000.0) throw Error("mailbox: no answer from the device within 10 s");
            }
        }
        n_mail++;
        const double t1 = now_ms();
        t_wait_rounds += t1 - t0;
        if (slow_log && t1 - t0 > 3.0) fprintf(stderr, "[hg] slow: waited %.2f ms for message %llu (%s)\n", t1 - t0, seq, where);
        E2 sums[3];
        for (int t = 0; t < nv; t++) sums[t] = e2(mail->slot[2 * t].v, mail->slot[2 * t + 1].v);
        reduce_sums(sums, nv);   // (sharded form: this rank's tiles only so far)
        const E2 r = round_message(sums, deg, claim, false);
        mail->chal[0] = r;
        __atomic_store_n(&mail->cpu_seq, seq, __ATOMIC_RELEASE);
        if (slow_log) t_answer += now_ms() - t1;
        return r;
    }
    // The last rounds of a sum-check on the host. Once the tables are small a round is nothing but latency on the device (kernel
    // start, cold loads, the mailbox trip: 11-20 us); a host core does the same round in the time of its arithmetic alone. The
    // device runs rounds 0 .. R-1, k_sq_export hands T_(R-1) over, and from there: fold with r_(R-1), sums, transcript, fold, ...
    // exactly the kernels' definitions (T_rd[j] = fold of T_(rd-1)[2j], [2j+1]; weights were multiplied in by the first fold, R >= 2).
    int host_tail_start(const SqJob& J) const {
        if (!host_tail_entries || !h_tail) return 0;
        for (int R = 2; R < J.nvars; R++)
            if (((size_t)J.ntab << (J.nvars - R + 1)) <= host_tail_entries && J.nvars - R + 1 <= 24) return R;
        return 0;
    }
    std::vector<E2> tail_a, tail_b, tail_sums;   // (tail_sums: the host rounds' sums of a node this rank owns, for the ghosts)
    // host arithmetic of the tail: products accumulated as 128-bit integers (+ carry), one reduction per extension-field coefficient
    struct Acc { unsigned __int128 v = 0; u64 hi = 0; void add(u64 a, u64 b) { const unsigned __int128 p = (unsigned __int128)a * b; v += p; hi += v < p; } };
    static u64 acc_reduce(const Acc& a) {   // v + hi 2^128 mod p; 2^128 = -2^32 (mod p)
        u64 r = gl_reduce128((u64)a.v, (u64)(a.v >> 64));
        if (a.hi) r = gl_sub(r, gl_mul(a.hi % GL_P, (u64)1 << 32));
        return r;
    }
    struct MulBy {   // x -> r * x for a fixed r
        u64 r0, r1, r1_7;
        explicit MulBy(E2 r) : r0(r.c0), r1(r.c1), r1_7(gl_mul7_lazy(r.c1)) {}
        E2 operator()(E2 d) const {
            Acc c0, c1;
            c0.add(r0, d.c0); c0.add(r1_7, d.c1);
            c1.add(r0, d.c1); c1.add(r1, d.c0);
            return e2(acc_reduce(c0), acc_reduce(c1));
        }
    };
    struct DotAcc {  // sum of products of extension-field elements, reduced once
        Acc c0, c1;
        void add(E2 a, E2 b) { c0.add(a.c0, b.c0); c0.add(gl_mul7_lazy(a.c1), b.c1); c1.add(a.c0, b.c1); c1.add(a.c1, b.c0); }
        E2 value() const { return e2(acc_reduce(c0), acc_reduce(c1)); }
    };
    SqInstall inst;
    void put_result(E2* where, E2 v) {   // a final evaluation: the result buffer is host memory (use_mail), anything else is device memory
        const E2* lo = ctx->h_res;
        if (where >= lo && where < lo + ctx->res_cap) { *where = v; return; }
        if (inst.nres >= 2 * dev::PS_MAX_PAIRS) throw Error("host tail: too many device-side results");
        inst.res_dst[inst.nres] = where; inst.res[inst.nres] = v; inst.nres++;
    }
    void host_tail(const SqJob& J, int R, E2& claim) {
        const int ntab = J.ntab, nvars = J.nvars, deg = J.kind == 1 ? 3 : 2;
        size_t len = (size_t)1 << (nvars - R + 1);                 // entries per table of T_(R-1)
        const E2* prev = h_tail;
        std::vector<E2>* bufs[2] = {&tail_a, &tail_b};
        E2 r = chain[J.r_off + R - 1];
        auto fold = [](E2 x, E2 y, E2 rr) { return e2_add(x, e2_mul(rr, e2_sub(y, x))); };
        for (int rd = R; rd < nvars; rd++) {
            const size_t half = len >> 1;
            std::vector<E2>& cur = *bufs[rd & 1];
            cur.resize((size_t)ntab * half);
            const MulBy times_r(r);
            for (int t = 0; t < ntab; t++) {
                const E2* pp = prev + (size_t)t * len;
                E2* cc = cur.data() + (size_t)t * half;
                for (size_t i = 0; i < half; i++) cc[i] = e2_add(pp[2 * i], times_r(e2_sub(pp[2 * i + 1], pp[2 * i])));
            }
            len = half;
            prev = cur.data();
            const size_t pairs = len >> 1;
            E2 sums[3] = {e2_zero(), e2_zero(), e2_zero()};
            if (J.kind == 2) {
                DotAcc d0, d2;
                for (int u = 0; u < ntab / 2; u++) {
                    const E2 *A = prev + (size_t)(2 * u) * len, *B = prev + (size_t)(2 * u + 1) * len;
                    for (size_t j = 0; j < pairs; j++) {
                        const E2 ax = A[2 * j], ay = A[2 * j + 1], bx = B[2 * j], by = B[2 * j + 1];
                        d0.add(ax, bx);
                        d2.add(e2_sub(e2_dbl(ay), ax), e2_sub(e2_dbl(by), bx));
                    }
                }
                sums[0] = d0.value(); sums[1] = d2.value();
            } else if (J.kind == 0) {   // sum_j p_0(j) * sum_u p_u(j) at t = 0 and t = 2 (the tables carry their weights already)
                for (size_t j = 0; j < pairs; j++) {
                    E2 s0 = e2_zero(), s2 = e2_zero(), q0 = e2_zero(), q2 = e2_zero();
                    for (int u = 0; u < ntab; u++) {
                        const E2 X = prev[(size_t)u * len + 2 * j], Y = prev[(size_t)u * len + 2 * j + 1];
                        const E2 v2 = e2_sub(e2_dbl(Y), X);
                        if (u == 0) { q0 = X; q2 = v2; }
                        s0 = e2_add(s0, X); s2 = e2_add(s2, v2);
                    }
                    sums[0] = e2_add(sums[0], e2_mul(q0, s0));
                    sums[1] = e2_add(sums[1], e2_mul(q2, s2));
                }
            } else {                    // sum_j l_0(j) * sum_u l_u(j) r_u(j) at t = 0, 2, 3
                for (size_t j = 0; j < pairs; j++) {
                    E2 s0 = e2_zero(), s2 = e2_zero(), s3 = e2_zero(), q0 = e2_zero(), q2 = e2_zero(), q3 = e2_zero();
                    for (int u = 0; u < ntab / 2; u++) {
                        const E2 *Lp = prev + (size_t)(2 * u) * len + 2 * j, *Rp = prev + (size_t)(2 * u + 1) * len + 2 * j;
                        const E2 lx = Lp[0], ly = Lp[1], rx = Rp[0], ry = Rp[1];
                        const E2 dl = e2_sub(ly, lx), dr = e2_sub(ry, rx);
                        const E2 l2 = e2_add(ly, dl), r2 = e2_add(ry, dr), l3 = e2_add(l2, dl), r3 = e2_add(r2, dr);
                        if (u == 0) { q0 = lx; q2 = l2; q3 = l3; }
                        s0 = e2_add(s0, e2_mul(lx, rx)); s2 = e2_add(s2, e2_mul(l2, r2)); s3 = e2_add(s3, e2_mul(l3, r3));
                    }
                    sums[0] = e2_add(sums[0], e2_mul(q0, s0));
                    sums[1] = e2_add(sums[1], e2_mul(q2, s2));
                    sums[2] = e2_add(sums[2], e2_mul(q3, s3));
                }
            }
            if (solo()) tail_sums.insert(tail_sums.end(), sums, sums + J.nv);
            r = round_message(sums, deg, claim, false);
            n_host_rounds++;
        }
        inst.nres = 0;
        for (int t = 0; t < ntab; t++) {   // k_sq_final: the last two entries folded with the last challenge
            const E2 v = fold(prev[(size_t)t * 2], prev[(size_t)t * 2 + 1], r);
            put_result(J.kind == 2 ? J.fin[t] : J.final_out + t, v);
        }
        // the challenges the device has not seen (r_(R-1) onwards) go into its table behind everything enqueued so far
        const size_t first = J.r_off + R - 1, cnt = (size_t)(nvars - R + 1);
        if (cnt > 24) throw Error("host tail: more rounds than the install launch carries");
        inst.chain_dst = d_chain + first;
        inst.nchain = (int)cnt;
        for (size_t i = 0; i < cnt; i++) inst.chain[i] = chain[first + i];
        k_sq_install<<<1, 64, 0, st>>>(inst);
    }

    // J: kind, ntab, nvars, tables, final destinations, pw filled in by the caller. Enqueues every round and the final fold, then
    // answers the rounds in order. Returns the chain position of the point.
    size_t run_sq(SqJob& J, bool in_base, E2& claim) {
        if (!d_ticket) {
            d_ticket = ctx->alloc_n<unsigned>(64);
            hip_check(hipMemsetAsync(d_ticket, 0, 64 * sizeof(unsigned), st), "clear ticket");
        }
        const int nvars = J.nvars;
        J.nv = J.kind == 1 ? 3 : 2;
        const size_t point_off = epos();
        J.r_off = point_off;
        if (ghost()) {
            // another rank runs this sum-check: the same all-reduces with zeros (one per device round, one for the rounds its host
            // finishes), the same transcript steps; the challenges go into this rank's device table afterwards
            const int R = host_tail_start(J), ndev = R ? R : nvars, deg = J.kind == 1 ? 3 : 2;
            for (int rd = 0; rd < ndev; rd++) {
                E2 sums[3] = {e2_zero(), e2_zero(), e2_zero()};
                reduce_sums(sums, J.nv);
                round_message(sums, deg, claim, false);
            }
            if (R) {
                std::vector<E2> ts((size_t)(nvars - R) * J.nv, e2_zero());
                share(ts.data(), ts.size(), false);
                for (int rd = R; rd < nvars; rd++) round_message(ts.data() + (size_t)(rd - R) * J.nv, deg, claim, false);
            }
            install_chain(point_off, (size_t)nvars);
            sq_results_on_host = true;   // (nothing of it is on this rank's device: the final evaluations arrive through share_slot_list)
            return point_off;
        }
        J.sums_slot = slot((size_t)nvars * J.nv);
        const size_t N = (size_t)1 << nvars;
        J.buf[1] = ctx->alloc_n<E2>((size_t)J.ntab * std::max<size_t>(N / 2, 1));
        J.buf[0] = ctx->alloc_n<E2>((size_t)J.ntab * std::max<size_t>(N / 4, 1));
        J.mail = mail;
        J.seq0 = mail_seq + 1;
        const int R = host_tail_start(J);          // 0: every round on the device
        J.dev_rounds = R;
        const int ndev = R ? R : nvars;            // rounds on the device; one more launch behind them (export or final fold)
        mail_seq += (unsigned long long)ndev + (R ? 1 : 0);
        const double te0 = now_ms();
        where = J.kind == 0 ? "collation round" : J.kind == 1 ? "grand-product round" : "pair-product round";
        const SqJob* d_job = nullptr;
        { Slow sl(this, "run_sq set-up (allocations, job upload)"); d_job = upload(&J, 1); }
        // the launches run a few rounds ahead of the answers: the device never waits for a launch, and the first round's answer does
        // not wait for the host to have enqueued the whole sum-check
        const int LOOKAHEAD = 6;   // (how far the launches may run ahead of the answers; the first answer waits for two launches only)
        int enq = 0;
        auto enqueue_round = [&] {
            Slow sl(this, "enqueueing a round");
            const int rd = enq;
            enq++;
            if (rd == ndev) {
                if (R) k_sq_export<<<1, 256, 0, st>>>(J.buf[(R - 1) & 1], (unsigned)((size_t)J.ntab << (nvars - R + 1)), h_tail, mail, J.seq0 + (unsigned long long)R);
                else if (in_base) k_sq_final<u64><<<(J.ntab + 63) / 64, 64, 0, st>>>(d_job, round_args(J, nvars, 0));
                else k_sq_final<E2><<<(J.ntab + 63) / 64, 64, 0, st>>>(d_job, round_args(J, nvars, 0));
                return;
            }
            int jb = 0, grid = 1;
            sq_plan(nvars - 1 - rd, J.kind == 0 ? J.ntab : J.ntab / 2, SQ_MAX_BLOCKS, &jb, &grid);
            if (J.kind == 0) launch_round<0>(in_base, J, d_job, rd, jb, grid);
            else if (J.kind == 1) launch_round<1>(in_base, J, d_job, rd, jb, grid);
            else launch_round<2>(in_base, J, d_job, rd, jb, grid);
        };
        // Sharded form: a round is launched only once the challenge of the round before has been posted, so that NO kernel ever waits
        // on the device. A rank's host answers a round only after the all-reduce, i.e. after every other rank's round kernel has run;
        // ranks that share a device (or a hardware queue of it: ranks as threads of one process) would otherwise wait for a kernel
        // that cannot start behind their own waiting one. One launch latency per round: this form is bound by the all-reduce anyway.
        const bool lockstep = shard_world > 1;
        while (enq < std::min(ndev + 1, lockstep ? 1 : 2)) enqueue_round();
        t_enqueue_rounds += now_ms() - te0;
        for (int rd = 0; rd < ndev; rd++) {
            // answering is what the device waits for: a launch is squeezed in first only while the round's sums are not there yet
            const unsigned long long seq = J.seq0 + (unsigned long long)rd;
            while (enq <= ndev && (lockstep ? enq <= rd : (enq < rd + 2 * LOOKAHEAD && (enq < rd + 2 || !slots_posted(seq, 2 * J.nv))))) {
                const double t1 = now_ms(); enqueue_round(); t_enqueue_rounds += now_ms() - t1;
            }
            const double w0 = t_wait_rounds;
            answer_round(seq, J.nv, J.kind == 1 ? 3 : 2, claim);
            const int big = nvars - 1 - rd > 8 ? 1 : 0;   // (statistics only)
            t_kind[J.kind][big] += t_wait_rounds - w0; n_kind[J.kind][big]++;
        }
        while (enq <= ndev) enqueue_round();
        if (R) {
            const double w0 = now_ms(), r0 = t_wait_rounds;
            mail_wait(J.seq0 + (unsigned long long)R);   // the tables are in h_tail
            t_wait_rounds = r0;
            t_wait_export += now_ms() - w0;
            const double h0 = now_ms();
            tail_sums.clear();
            host_tail(J, R, claim);
            t_host_tail += now_ms() - h0;
            if (solo()) share(tail_sums.data(), tail_sums.size(), true);   // the ghosts absorb the same messages
        }
        sq_results_on_host = R != 0;
        return point_off;
    }
    double t_wait_export = 0, t_host_tail = 0;

    // ---- prove_sum_check, stride layout (collation / grand-product shapes) ------------------------------------------------
    // tables: ntab rows at `in + t * in_stride` (u64 if base else E2). final evaluations land in d_res[evals_slot ..).
    size_t sumcheck_stride(int kind, const void* in, bool base, size_t in_stride, int ntab, int nvars, const dev::Powers& pw, E2& claim, size_t evals_slot) {
        if (shard_world > 1 && !(use_mail && !classic())) throw Error("sharded prove: needs the mailbox round kernels (HG_SEQ_NO_MAIL / HG_SEQ_CLASSIC unset)");
        if (use_mail && !classic()) {
            SqJob Q;
            memset(&Q, 0, sizeof(Q));
            Q.kind = kind == dev::SC_GRANDPROD ? 1 : 0; Q.ntab = ntab; Q.nvars = nvars;
            Q.in = in; Q.in_stride = in_stride;
            Q.final_out = d_res() + evals_slot;
            memcpy(Q.pw, pw.v, sizeof(Q.pw));
            return run_sq(Q, base, claim);
        }
        const int nv = kind == dev::SC_GRANDPROD ? 3 : 2, deg = nv;
        const size_t point_off = epos();
        const size_t sums_slot = slot((size_t)nvars * nv);
        const size_t N = (size_t)1 << nvars;
        dev::StJob J;
        memset(&J, 0, sizeof(J));
        J.in = in; J.in_stride = in_stride;
        J.buf[0] = ctx->alloc_n<E2>((size_t)ntab * std::max<size_t>(N / 2, 1));
        J.buf[1] = ctx->alloc_n<E2>((size_t)ntab * std::max<size_t>(N / 4, 1));
        J.final_out = d_res() + evals_slot;
        J.kind = kind; J.ntab = ntab; J.nvars = nvars; J.base = base ? 1 : 0;
        J.r_off = point_off; J.sums_slot = sums_slot;
        memcpy(J.pw, pw.v, sizeof(J.pw));
        dev::StJob* d_job = upload(&J, 1);
        // every round's item and grid are known up front (sizes only): one upload for the whole sum-check
        std::vector<dev::StItem> items(nvars);
        std::vector<int> grids(nvars);
        {
            const void* cur_in = in;
            size_t cur_stride = in_stride;
            for (int rd = 0; rd < nvars; rd++) {
                const int h = nvars - 1 - rd;
                dev::StItem& it = items[rd];
                memset(&it, 0, sizeof(it));
                it.job = 0; it.h_log2 = h; it.in = cur_in; it.in_stride = cur_stride;
                it.out = rd == nvars - 1 ? J.final_out : (cur_in == (const void*)J.buf[0] ? J.buf[1] : J.buf[0]);
                grids[rd] = dev::st_plan_blocks(&it, 1, false);
                cur_in = it.out; cur_stride = (size_t)1 << h;
            }
        }
        dev::StItem* d_items = upload(items.data(), items.size());
        if (use_mail) {
            int npw = 0;
            for (int i = 0; i < dev::PW_MAX; i++) if (pw.v[i].c0 | pw.v[i].c1) npw = i + 1;
            std::vector<unsigned long long> seqs(nvars);
            for (int rd = 0; rd < nvars; rd++) {
                // the first-round kernels store the weighted fold pw[i] * (x + r d) as pw[i] x + pwr[i] d with pwr = pw * r_0: r_0 is
                // not known in the first pass (pwr = 0: sums unaffected); k_mail fills pwr in before the second pass
                dev::st_step(st, kind, base && rd == 0, d_job, d_items + rd, 1, grids[rd], d_chain, ctx->d_partials, d_res());
                seqs[rd] = mail_enqueue(point_off + rd, rd == 0 ? d_job : nullptr, rd == 0 ? npw : 0);
                dev::st_step(st, kind, base && rd == 0, d_job, d_items + rd, 1, grids[rd], d_chain, ctx->d_partials, d_res());  // the real fold
            }
            for (int rd = 0; rd < nvars; rd++) mail_answer(seqs[rd], h_res() + sums_slot + (size_t)rd * nv, deg, claim);
            return point_off;
        }
        for (int rd = 0; rd < nvars; rd++) {
            dev::StItem* d_it = d_items + rd;
            const int grid = grids[rd];
            dev::st_step(st, kind, base && rd == 0, d_job, d_it, 1, grid, d_chain, ctx->d_partials, d_res());
            auto pass2 = [&] { dev::st_step(st, kind, base && rd == 0, d_job, d_it, 1, grid, d_chain, ctx->d_partials, d_res()); };  // the real fold
            {
                sync();
                const E2 r = round_message(h_res() + sums_slot + (size_t)rd * nv, deg, claim);
                if (rd == 0) {
                    for (int i = 0; i < dev::PW_MAX; i++) J.pwr[i] = e2_mul(pw.v[i], r);
                    hip_check(hipMemcpyAsync(d_job, stage(&J, sizeof(J)), sizeof(J), hipMemcpyHostToDevice, st), "upload job");
                }
                pass2();
            }
        }
        return point_off;
    }

    // ---- prove_sum_check, sum of pair products (Libra / zkCNN reductions) --------------------------------------------------
    size_t sumcheck_prodsum(const std::vector<const u64*>& a, const std::vector<const E2*>& b, int nvars, const std::vector<E2*>& fin_a,
                            const std::vector<E2*>& fin_b, E2& claim) {
        if (shard_world > 1 && !(use_mail && !classic())) throw Error("sharded prove: needs the mailbox round kernels (HG_SEQ_NO_MAIL / HG_SEQ_CLASSIC unset)");
        if (use_mail && !classic()) {
            SqJob Q;
            memset(&Q, 0, sizeof(Q));
            const int np = (int)a.size();
            if (np > dev::PS_MAX_PAIRS) throw Error("prodsum: too many table pairs");
            Q.kind = 2; Q.ntab = 2 * np; Q.nvars = nvars;
            for (int i = 0; i < np; i++) { Q.tab[2 * i] = a[i]; Q.tab[2 * i + 1] = b[i]; Q.fin[2 * i] = fin_a[i]; Q.fin[2 * i + 1] = fin_b[i]; }
            return run_sq(Q, true, claim);
        }
        const size_t point_off = epos();
        const size_t sums_slot = slot((size_t)nvars * 2);
        const int np = (int)a.size();
        if (np > dev::PS_MAX_PAIRS) throw Error("prodsum: too many table pairs");
        const size_t N = (size_t)1 << nvars;
        dev::PsJob J;
        memset(&J, 0, sizeof(J));
        J.npairs = np; J.nvars = nvars; J.r_off = point_off; J.sums_slot = sums_slot; J.tail_rd = nvars; J.tail_buf = -1;
        for (int q = 0; q < 2; q++) {
            J.bufa[q] = ctx->alloc_n<E2>((size_t)np * std::max<size_t>(N >> (q + 1), 1));
            J.bufb[q] = ctx->alloc_n<E2>((size_t)np * std::max<size_t>(N >> (q + 1), 1));
        }
        for (int i = 0; i < np; i++) { J.a[i] = a[i]; J.b[i] = b[i]; J.fin_a[i] = fin_a[i]; J.fin_b[i] = fin_b[i]; }
        dev::PsJob* d_job = upload(&J, 1);
        std::vector<dev::PsItem> items(nvars);
        std::vector<int> grids(nvars);
        {
            int cur = -1;
            for (int rd = 0; rd < nvars; rd++) {
                dev::PsItem& it = items[rd];
                memset(&it, 0, sizeof(it));
                it.job = 0; it.rd = rd; it.in_buf = cur;
                it.out_buf = rd == nvars - 1 ? -1 : (cur == 0 ? 1 : 0);
                grids[rd] = dev::ps_plan_blocks(&it, 1, &J, false);
                cur = it.out_buf;
            }
        }
        dev::PsItem* d_items = upload(items.data(), items.size());
        if (use_mail) {
            std::vector<unsigned long long> seqs(nvars);
            for (int rd = 0; rd < nvars; rd++) {
                dev::ps_round(st, false, d_job, d_items + rd, 1, grids[rd], d_chain, ctx->d_partials, d_res());
                seqs[rd] = mail_enqueue(point_off + rd, nullptr, 0);
                dev::ps_round(st, false, d_job, d_items + rd, 1, grids[rd], d_chain, ctx->d_partials, d_res());
            }
            for (int rd = 0; rd < nvars; rd++) mail_answer(seqs[rd], h_res() + sums_slot + 2 * (size_t)rd, 2, claim);
            return point_off;
        }
        for (int rd = 0; rd < nvars; rd++) {
            dev::PsItem* d_it = d_items + rd;
            const int grid = grids[rd];
            dev::ps_round(st, false, d_job, d_it, 1, grid, d_chain, ctx->d_partials, d_res());
            sync();
            round_message(h_res() + sums_slot + 2 * (size_t)rd, 2, claim);
            dev::ps_round(st, false, d_job, d_it, 1, grid, d_chain, ctx->d_partials, d_res());
        }
        return point_off;
    }

    // ---- small device helpers ---------------------------------------------------------------------------------------------------
    void eq_table(E2* out, int n, const dev::ClaimSet& cs) {
        const size_t N = (size_t)1 << n;
        std::vector<dev::EqJob> jobs;
        E2* tmp = cs.n > 1 ? ctx->alloc_n<E2>((size_t)cs.n * N) : nullptr;
        for (int a = 0; a < cs.n; a++) {
            dev::EqJob J;
            memset(&J, 0, sizeof(J));
            J.n = n; J.out = cs.n > 1 ? tmp + (size_t)a * N : out;
            J.cs.n = 1; J.cs.unit_alpha = cs.unit_alpha; J.cs.alpha_off = cs.alpha_off + a; J.cs.point_off[0] = cs.point_off[a];
            jobs.push_back(J);
        }
        dev::eq_jobs(st, upload(jobs.data(), jobs.size()), (int)jobs.size(), n, d_chain);
        if (cs.n > 1) dev::sum_tables(st, out, tmp, cs.n, N);
    }
    void eq_single(E2* out, int n, size_t point_off) {
        dev::ClaimSet cs;
        memset(&cs, 0, sizeof(cs));
        cs.n = 1; cs.unit_alpha = 1; cs.point_off[0] = point_off;
        eq_table(out, n, cs);
    }
    // out[t] = sum_j eq[j] tabs[t][j]
    void dots(const E2* eq, const std::vector<const u64*>& tabs, size_t n, size_t out_slot) {
        for (size_t o = 0; o < tabs.size(); o += 8) {
            const int cnt = (int)std::min<size_t>(8, tabs.size() - o);
            const u64* t8[8] = {nullptr};
            for (int t = 0; t < cnt; t++) t8[t] = tabs[o + t];
            dev::dot_eq(st, eq, t8, cnt, n, ctx->d_partials, d_res() + out_slot + o);
        }
    }

    // ---- prove_grand_product (prover.rs:183-266) on nb rows of `len` entries (u64, or E2 when `ext`) ---------------------
    size_t grand_product(const void* H, bool ext, size_t len, int nb) {
        int nv = 0;
        while (((size_t)1 << nv) < len) nv++;
        const size_t el = ext ? sizeof(E2) : sizeof(u64);
        std::vector<const void*> lev(nv, nullptr);
        lev[0] = H;
        for (int k = 1; k < nv; k++) {  // Layer::bottom / Layer::up
            const size_t in_len = len >> (k - 1);
            void* out = ctx->alloc((size_t)nb * (len >> k) * el);
            if (ext) k_prod_level_e2<<<dim3((unsigned)std::min<size_t>((in_len / 2 + 255) / 256, 256), (unsigned)nb), 256, 0, st>>>(static_cast<const E2*>(lev[k - 1]), in_len, static_cast<E2*>(out));
            else dev::prod_level(st, static_cast<const u64*>(lev[k - 1]), in_len, static_cast<u64*>(out), nb);
            lev[k] = out;
        }
        const size_t roots = slot(nb), ev0 = slot(2 * (size_t)nb);
        if (ext) k_gp_top_e2<<<(nb + 63) / 64, 64, 0, st>>>(static_cast<const E2*>(lev[nv - 1]), nb, d_res() + roots, d_res() + ev0);
        else dev::gp_top(st, static_cast<const u64*>(lev[nv - 1]), nb, d_res() + roots, d_res() + ev0);
        wait_results();
        std::vector<E2> cl(nb);
        for (int b = 0; b < nb; b++) { cl[b] = h_res()[roots + b]; tr.write_e(cl[b]); }  // root products (prover.rs:197-221)
        write_slots(ev0, 2 * (size_t)nb);                                                // layer 0: v_l, v_r (prover.rs:257)
        size_t point_off = epos();
        auto layer_down = [&](size_t evals_slot) {  // prover.rs:259, 288-294
            const E2 mu = squeeze();
            const E2* ev = h_res() + evals_slot;
            for (int b = 0; b < nb; b++) cl[b] = e2_add(ev[2 * b], e2_mul(mu, e2_sub(ev[2 * b + 1], ev[2 * b])));
        };
        layer_down(ev0);
        for (int n = 1; n < nv; n++) {
            const int k = nv - 1 - n;
            const size_t h = (size_t)1 << n;
            const E2 gamma = squeeze();  // prover.rs:238
            dev::Powers pw;
            memset(&pw, 0, sizeof(pw));
            if (nb > dev::PW_MAX) throw Error("grand product: too many batched tables");
            E2 g = e2_one(), claim = e2_zero();
            for (int b = 0; b < nb; b++) { pw.v[b] = g; claim = e2_add(claim, e2_mul(cl[b], g)); g = e2_mul(g, gamma); }  // prover.rs:281-286
            const size_t evals = slot(2 * (size_t)nb);
            point_off = sumcheck_stride(dev::SC_GRANDPROD, lev[k], !ext, h, 2 * nb, n, pw, claim, evals);
            wait_sumcheck_results();
            // the kernels leave the final LEFT evaluation of pair b multiplied by pw[b] = gamma^b
            if (nb >= 2) {
                if (pw.v[1].c0 == 0 && pw.v[1].c1 == 0) throw Error("grand product: zero batching weight");
                const E2 ginv = e2_inv(pw.v[1]);
                E2 w = ginv;
                for (int b = 1; b < nb; b++) { ctx->h_res[evals + 2 * b] = e2_mul(ctx->h_res[evals + 2 * b], w); w = e2_mul(w, ginv); }
            }
            write_slots(evals, 2 * (size_t)nb);  // prover.rs:257
            layer_down(evals);
        }
        return point_off;
    }

    // ---- LassoNode::prove_claim_reduction (lasso.rs:57-114) ------------------------------------------------------------------
    Claim lasso_node(const u64* d_input) {
        Slow sl_all(this, "lasso node (all of it)");
        const LassoPlan& lp = pk->lasso;
        const dev::LassoDev& L = pk->lasso_dev;
        const int nu = lp.nu, A = lp.alpha;
        const size_t N = (size_t)1 << nu, M = 65536;
        u64* dims = ctx->alloc_n<u64>(4 * N);
        u64* ep = ctx->alloc_n<u64>((size_t)A * N);
        dev::lasso_split(st, L, d_input, dims, ep, dev::ep_rows_all(A));  // polynomialize (lasso.rs:157-250)
        const size_t r_off = epos();
        for (int i = 0; i < nu; i++) squeeze();       // lasso.rs:85
        E2* eq = ctx->alloc_n<E2>(N);
        eq_single(eq, nu, r_off);
        const size_t claim_slot = slot(1);
        {
            int grid = dev::lasso_claim(st, L, eq, ep, dev::ep_rows_all(A), ctx->d_partials);
            dev::reduce_partials(st, ctx->d_partials, grid, 1, d_res() + claim_slot);
        }
        wait_results();
        const E2 claimed = h_res()[claim_slot];
        tr.write_e(claimed);  // lasso.rs:100-107
        {   // collation sum-check (lasso.rs:271-279), result dropped (:97)
            dev::Powers pw;
            memset(&pw, 0, sizeof(pw));
            if (A > dev::PW_MAX) throw Error("lasso: too many memories");
            u64 c = 1;
            for (int i = 0; i < A; i++) { pw.v[i] = e2(c, 0); c = gl_mul(c, M); }
            E2 claim = claimed;
            sumcheck_stride(dev::SC_COLLATION, ep, true, N, A, nu, pw, claim, slot(A));
        }
        const E2 gamma_e = squeeze(), tau_e = squeeze();  // lasso.rs:99
        // counters of the chunk-indexed memories (lasso.rs:317-319)
        std::map<int, u64*> read_ts, final_cts;
        {
            size_t tb = dev::lasso_counter_temp_bytes(N);
            void* temp = ctx->alloc(tb);
            u32* keys = ctx->alloc_n<u32>(N); u32* keys2 = ctx->alloc_n<u32>(N);
            u32* rows = ctx->alloc_n<u32>(N); u32* rows2 = ctx->alloc_n<u32>(N);
            u32* starts = ctx->alloc_n<u32>(65537);
            for (auto& chk : lp.chunks) {
                int c = chk.first;
                if (c < 0 || c >= 4) continue;
                read_ts[c] = ctx->alloc_n<u64>(N);
                final_cts[c] = ctx->alloc_n<u64>(M);
                dev::lasso_counters(st, L, c, dims, read_ts[c], final_cts[c], temp, tb, keys, keys2, rows, rows2, starts);
            }
        }
        // MemoryCheckingProver::new (prover.rs:35-89): reads | writes of every memory in memory-GKR order, then inits | finals
        const int G = (int)lp.gkr_order.size();
        if (G > 32) throw Error("lasso: more than 32 memories");
        size_t xoff, yoff;
        if (!ext_mc()) {
            const u64 gamma = gamma_e.c0, tau = tau_e.c0;  // prover.rs:38-39
            u64* H1 = ctx->alloc_n<u64>((size_t)2 * G * N);
            u64* H2 = ctx->alloc_n<u64>((size_t)2 * G * M);
            for (int i = 0; i < G; i++) {
                dev::HashRwArgs ha;
                memset(&ha, 0, sizeof(ha));
                ha.ep[0] = ep + (size_t)lp.gkr_order[i] * N; ha.rd[0] = H1 + (size_t)i * N; ha.wr[0] = H1 + (size_t)(G + i) * N;
                const int c = lp.gkr_chunk[i];
                dev::lasso_hash_rw(st, N, dims + (size_t)c * N, read_ts[c], ha, 1, gamma, tau);
            }
            dev::HashIfArgs hi;
            memset(&hi, 0, sizeof(hi));
            for (int i = 0; i < G; i++) { hi.cutoff[i] = (u32)lp.mems[lp.gkr_order[i]].cutoff; hi.fc[i] = final_cts[lp.gkr_chunk[i]]; hi.row_init[i] = i; hi.row_fin[i] = G + i; }
            dev::lasso_hash_if(st, hi, G, gamma, tau, H2);
            xoff = grand_product(H1, false, N, 2 * G);   // prover.rs:161-165
            yoff = grand_product(H2, false, M, 2 * G);   // prover.rs:167-171
        } else {
            const E2 g2 = e2_mul(gamma_e, gamma_e);
            E2* H1 = ctx->alloc_n<E2>((size_t)2 * G * N);
            E2* H2 = ctx->alloc_n<E2>((size_t)2 * G * M);
            for (int i = 0; i < G; i++) {
                const int c = lp.gkr_chunk[i];
                k_hash_rw_e2<<<1024, 256, 0, st>>>(N, dims + (size_t)c * N, read_ts[c], ep + (size_t)lp.gkr_order[i] * N, gamma_e, g2, tau_e,
                                                   H1 + (size_t)i * N, H1 + (size_t)(G + i) * N);
                k_hash_if_e2<<<256, 256, 0, st>>>((u32)lp.mems[lp.gkr_order[i]].cutoff, final_cts[c], gamma_e, g2, tau_e, H2 + (size_t)i * M, H2 + (size_t)(G + i) * M);
            }
            xoff = grand_product(H1, true, N, 2 * G);
            yoff = grand_product(H2, true, M, 2 * G);
        }
        // openings (prover.rs:173-178, mod.rs:80-93)
        E2* eqx = eq;
        E2* eqy = ctx->alloc_n<E2>(M);
        eq_single(eqx, nu, xoff);
        eq_single(eqy, 16, yoff);
        std::vector<std::pair<size_t, size_t>> wires;  // (slot, count) in wire order
        for (auto& chk : lp.chunks) {
            const int c = chk.first;
            const size_t base = slot(3 + chk.second.size());
            std::vector<const u64*> xs = {dims + (size_t)c * N, read_ts[c]};
            const size_t tmp = slot(2 + chk.second.size());
            for (int m : chk.second) xs.push_back(ep + (size_t)m * N);
            dots(eqx, xs, N, tmp);
            dots(eqy, {final_cts[c]}, M, base + 2);
            wires.push_back({base, tmp});
        }
        wait_results();
        size_t q = 0;
        for (auto& chk : lp.chunks) {  // dim(x), read_ts(x), final_cts(y), then E_m(x)
            const size_t base = wires[q].first, tmp = wires[q].second;
            q++;
            tr.write_e(h_res()[tmp]); tr.write_e(h_res()[tmp + 1]); tr.write_e(h_res()[base + 2]);
            for (size_t i = 0; i < chk.second.size(); i++) tr.write_e(h_res()[tmp + 2 + i]);
        }
        return Claim{r_off, nu, claimed};  // lasso.rs:97,113
    }

    // ---- prove_gkr (sk_encryption_circuit.rs:455-457), conventions G1-G4 of the fast path ---------------------------------
    dev::ClaimSet claim_set(const std::vector<Claim>& cl, std::vector<E2>* alphas) {
        dev::ClaimSet cs;
        memset(&cs, 0, sizeof(cs));
        if (cl.empty()) throw Error("gkr: node without claim");
        if (cl.size() > (size_t)dev::MAX_CLAIMS) throw Error("gkr: too many claims on one node");
        cs.n = (int)cl.size();
        cs.unit_alpha = cl.size() == 1;
        cs.alpha_off = epos();
        alphas->clear();
        if (cl.size() > 1) for (size_t a = 0; a < cl.size(); a++) alphas->push_back(squeeze());
        else alphas->push_back(e2_one());
        for (size_t a = 0; a < cl.size(); a++) cs.point_off[a] = cl[a].point_off;
        return cs;
    }
    static E2 combined(const std::vector<Claim>& cl, const std::vector<E2>& alphas) {
        E2 s = e2_zero();
        for (size_t a = 0; a < cl.size(); a++) s = e2_add(s, e2_mul(cl[a].value, alphas[a]));
        return s;
    }

    void vanilla_node(int id) {
        Slow sl_all(this, "a vanilla node (all of it)");
        const HNode& n = pk->circuit.nodes[id];
        const hg_pk::NodeDev& nd = pk->node_dev[id];
        const int nin = n.log2_sub_in + n.log2_reps;
        const size_t SR = (size_t)1 << nin;
        Owned scope(this, own_mode ? plan.node_owner[id] : -1);
        const bool gh = ghost();   // another rank's node: no device work here, its results arrive through the group
        if (own_mode && hg_debug("shard")) fprintf(stderr, "[hg] rank %d: vanilla node %d, owner %d\n", shard_rank, id, cur_owner);
        std::vector<E2> alphas;
        dev::ClaimSet cs = claim_set(claims[id], &alphas);
        for (auto& c : claims[id]) if (c.len != n.log2_out()) throw Error("gkr: claim arity mismatch");
        E2 claim = combined(claims[id], alphas);
        E2* eqc = gh ? nullptr : ctx->alloc_n<E2>((size_t)1 << n.log2_out());
        if (!gh) eq_table(eqc, n.log2_out(), cs);
        if (nd.nconst) {  // claim -= sum_g eqc[g] w0_g
            const size_t s = slot(1);
            if (!gh) {
                int grid = dev::vanilla_const_sum(st, nd.const_gate, nd.const_coef, nd.nconst, eqc, n.log2_sub_out, n.log2_reps, ctx->d_partials);
                dev::reduce_partials(st, ctx->d_partials, grid, 1, d_res() + s);
                wait_results();
            }
            share_slot_list({s});
            claim = e2_sub(claim, h_res()[s]);
        }
        std::vector<int> li, ri;
        for (int i = 0; i < n.arity; i++) { if (n.left_use[i]) li.push_back(i); if (n.right_use[i]) ri.push_back(i); }
        if (n.arity > dev::PS_MAX_PAIRS) throw Error("vanilla: arity too large");
        dev::GatherT gt;
        memset(&gt, 0, sizeof(gt));
        for (int i = 0; i < n.arity; i++) gt.in_vals[i] = d_vals[n.preds[i]];
        std::vector<const u64*> a;
        std::vector<const E2*> b;
        std::vector<E2*> fa, fb;
        const size_t u_base = slot(n.arity);
        E2* scratch = gh ? nullptr : ctx->alloc_n<E2>(n.arity);
        std::vector<dev::GatherJob> gj;
        for (int i : li) {
            E2* T = gh ? nullptr : ctx->alloc_n<E2>(SR);
            gt.lin = nd.lin[i];
            gt.mul = nd.mulL[i];
            gj.push_back(dev::GatherJob{gt, eqc, n.log2_sub_in, n.log2_sub_out, n.log2_reps, T});
            a.push_back(d_vals[n.preds[i]]);
            b.push_back(T);
            fa.push_back(d_res() + u_base + i);
            fb.push_back(scratch + i);
        }
        if (!gj.empty() && !gh) dev::gather_jobs(st, upload(gj.data(), gj.size()), (int)gj.size(), SR);
        const size_t rx_off = sumcheck_prodsum(a, b, nin, fa, fb, claim);  // Libra phase 1
        wait_sumcheck_results();
        {
            std::vector<size_t> sl;
            for (int i : li) sl.push_back(u_base + i);
            share_slot_list(sl);
        }
        for (int i : li) {
            const E2 v = h_res()[u_base + i];
            tr.write_e(v);
            claims[n.preds[i]].push_back(Claim{rx_off, nin, v});
        }
        if (n.mul.empty()) return;
        if (!n.lin.empty()) throw Error("vanilla: nodes mixing linear and mul gates are not on this path");
        // phase 2: sum_y sum_i in_i(y) B_i(y); the claim carries over (no linear part)
        E2* eqx = gh ? nullptr : ctx->alloc_n<E2>(SR);
        if (!gh) eq_single(eqx, nin, rx_off);
        std::vector<const u64*> a2;
        std::vector<const E2*> b2;
        std::vector<E2*> fa2, fb2;
        const size_t w_base = slot(n.arity);
        std::vector<dev::GatherBJob> bj;
        for (int i : ri) {
            E2* B = gh ? nullptr : ctx->alloc_n<E2>(SR);
            bj.push_back(dev::GatherBJob{nd.mulR[i], eqc, eqx, d_res() + u_base, n.log2_sub_in, n.log2_sub_out, n.log2_reps, B});
            a2.push_back(d_vals[n.preds[i]]);
            b2.push_back(B);
            fa2.push_back(d_res() + w_base + i);
            fb2.push_back(scratch + i);
        }
        if (!gh) dev::gather_B_jobs(st, upload(bj.data(), bj.size()), (int)bj.size(), SR);
        const size_t ry_off = sumcheck_prodsum(a2, b2, nin, fa2, fb2, claim);
        wait_sumcheck_results();
        {
            std::vector<size_t> sl;
            for (int i : ri) sl.push_back(w_base + i);
            share_slot_list(sl);
        }
        for (int i : ri) {
            const E2 v = h_res()[w_base + i];
            tr.write_e(v);
            claims[n.preds[i]].push_back(Claim{ry_off, nin, v});
        }
    }

    void fft_node(int id) {
        Slow sl_all(this, "an fft node (all of it)");
        const HNode& n = pk->circuit.nodes[id];
        const int L = n.log2_size;
        const size_t N = (size_t)1 << L;
        Owned scope(this, own_mode ? plan.node_owner[id] : -1);
        const bool gh = ghost();
        if (own_mode && hg_debug("shard")) fprintf(stderr, "[hg] rank %d: fft node %d, owner %d\n", shard_rank, id, cur_owner);
        std::vector<E2> alphas;
        dev::ClaimSet cs = claim_set(claims[id], &alphas);
        E2 claim = combined(claims[id], alphas);
        E2* F = gh ? nullptr : ctx->alloc_n<E2>(N);
        if (!gh) {
            const u64* W = (n.inverse ? pk->w_inv : pk->w_fwd).at(L);
            dev::FftJob fj{F, W, n.inverse ? gl_inv(gl_from_u64(N)) : 1, L, cs};
            E2* tab = ctx->alloc_n<E2>((size_t)cs.n * (N >> 4) + 1);
            dev::fft_jobs(st, upload(&fj, 1), 1, L, cs.n, d_chain, tab);
        }
        const size_t u = slot(1);
        E2* scratch = gh ? nullptr : ctx->alloc_n<E2>(1);
        const size_t off = sumcheck_prodsum({d_vals[n.preds[0]]}, {F}, L, {d_res() + u}, {scratch}, claim);
        wait_sumcheck_results();
        share_slot_list({u});
        const E2 v = h_res()[u];
        tr.write_e(v);
        claims[n.preds[0]].push_back(Claim{off, L, v});
    }

    void gkr(const Claim& sum_claim) {
        const HCircuit& c = pk->circuit;
        claims.assign(c.nodes.size(), {});
        claims[c.lasso_id].push_back(Claim{epos(), 0, e2_zero()});  // EvalClaim::new(vec![], E::ZERO) (:450)
        claims[c.sum_id].push_back(sum_claim);
        for (size_t q = c.topo.size(); q-- > 0;) {
            const int id = c.topo[q];
            const HNode& n = c.nodes[id];
            switch (n.kind) {
                case NK_INPUT: break;
                case NK_VANILLA: vanilla_node(id); break;
                case NK_FFT: fft_node(id); break;
                case NK_LASSO: claims[n.preds[0]].push_back(lasso_node(d_vals[n.preds[0]])); break;
            }
        }
    }
};

}  // namespace

// BfvEncrypt::prove on resident node values in a non-default protocol mode; gpu_ms = wall time of the whole walk (the device
// is synchronised every round, so there is no separate device span)
ProveResult prove_resident_mode(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int mode) {
    return prove_resident_mode_sharded(ctx, pk, v, mode, 0, 1, nullptr, nullptr);
}
// The round-by-round prover on `world` ranks. With the whole witness on every rank: rank r evaluates the hypercube sums of the tiles
// t = r (mod world) of every round kernel and folds everything; one all-reduce (`reduce`) per round completes the sums, after which
// every rank's transcript absorbs the same message and squeezes the same challenge. Rounds finished on the host (host_tail) and the
// scalar steps between the sum-checks are replicated. With a rank's share of the witness (hg_witness_gen_shard): node ownership, see
// SeqProver::own_mode. Every rank returns the same proof, byte for byte the single-rank one.
ProveResult prove_resident_mode_sharded(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int mode, int rank, int world,
                                        int (*reduce)(void*, uint64_t*, size_t), void* user) {
    if (world < 1 || rank < 0 || rank >= world) throw Error("sharded prove: rank out of range");
    if (world > 1 && !reduce) throw Error("sharded prove: no all-reduce given");
    if (mode == 0) {
        if (world > 1) throw Error("sharded prove: mode 0 shards through hg_prove_sharded / hg_prove_shard_begin (one all-reduce per proof)");
        return prove_resident(ctx, pk, v);
    }
    if (mode < 0 || mode > 3) throw Error("hg_prove_mode: unknown mode bits");
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->arena_reset();
    const double t0 = now_ms();
    SeqProver P(ctx, pk, mode);
    P.shard_rank = rank; P.shard_world = world; P.reduce = reduce; P.reduce_user = user;
    P.d_vals = v->d_vals;
    if (v->shard_rank >= 0) {   // a rank's share of the witness (hg_witness_gen_shard): node reductions run on their owners only
        if (v->shard_rank != rank || v->shard_world != world) throw Error("sharded prove: these values hold the tables of rank " + std::to_string(v->shard_rank) + " of " + std::to_string(v->shard_world));
        P.own_mode = world > 1;
        P.plan = shard_plan(pk, rank, world);
    }
    const Params& p = pk->params;
    // "eval output" (sk_encryption_circuit.rs:444-448)
    const int ov = p.ct0is_log2();
    const size_t point_off = P.epos();
    for (int i = 0; i < ov; i++) P.squeeze();
    const size_t vslot = P.slot(1);
    const bool out_mine = !P.own_mode || P.plan.own_out_claim == rank;   // (own_mode: ct0is is resident on the owner of the output claim only)
    if (out_mine) {
        E2* eq = ctx->alloc_n<E2>((size_t)1 << ov);
        P.eq_single(eq, ov, point_off);
        P.dots(eq, {v->d_ct0is}, (size_t)1 << ov, vslot);
        P.wait_results();
    }
    if (P.own_mode) P.share(ctx->h_res + vslot, 1, out_mine);
    P.gkr(Claim{point_off, ov, ctx->h_res[vslot]});
    hip_check(hipGetLastError(), "prove (mode)");
    ProveResult res;
    res.prove_ms = now_ms() - t0;
    res.gpu_ms = res.prove_ms;
    res.sync_ms = (double)P.n_sync;  // number of synchronisations (reported through hg_timings::sync_ms in this mode)
    res.enqueue_ms = (double)P.n_mail;  // ... and of mailbox round trips (hg_timings::enqueue_ms in this mode)
    res.replay_ms = (double)P.n_reduce; // ... and of all-reduces (sharded form; hg_timings::replay_ms in this mode)
    if (hg_times("seq"))
        fprintf(stderr, "[hg] mode %d: %.2f ms; host waited %.2f ms for round sums (%zu round trips), %.2f ms for other results, spent %.2f ms enqueueing rounds\n", mode,
                res.prove_ms, P.t_wait_rounds, P.n_mail, P.t_wait_results, P.t_enqueue_rounds);
    if (hg_times("seq")) fprintf(stderr, "[hg]   host transcript steps between 'sums seen' and 'challenge posted': %.2f ms in total\n", P.t_answer);
    if (hg_times("seq")) fprintf(stderr, "[hg]   rounds finished on the host: %zu in %.2f ms, after waiting %.2f ms for their tables\n", P.n_host_rounds, P.t_host_tail, P.t_wait_export);
#ifdef HG_SEQ_STAMPS
    if (P.mail && P.mail->dbg[3])
        fprintf(stderr, "[hg]   device clocks per round, posting workgroup (us): kernel start -> challenge in hand %.2f, -> sums %.2f, -> tagged stores issued %.2f (%llu rounds)\n",
                P.mail->dbg[0] / 100.0 / P.mail->dbg[3], P.mail->dbg[1] / 100.0 / P.mail->dbg[3], P.mail->dbg[2] / 100.0 / P.mail->dbg[3], P.mail->dbg[3]);
#endif
    if (hg_times("seq"))
        for (int kd = 0; kd < 3; kd++)
            fprintf(stderr, "[hg]   kind %d: %zu one-workgroup rounds %.2f ms, %zu larger rounds %.2f ms\n", kd, P.n_kind[kd][0], P.t_kind[kd][0], P.n_kind[kd][1], P.t_kind[kd][1]);
    if (P.mail && P.mail->timeouts) throw Error("mailbox: a device-side wait timed out");

    res.proof = std::move(P.tr.bytes);
    return res;
}

}  // namespace hg
