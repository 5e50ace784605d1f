// gfx950 (MI355X / CDNA4) kernels of the GKR prover. Integer modular work in 64-bit lanes: no MFMA.
// Conventions: 256-thread workgroups (4 wave64), grid-stride loops over at most SC_MAX_BLOCKS workgroups per job,
// 16-byte coalesced loads of adjacent table entries (a sum-check pair (T[2j], T[2j+1]) is one 16-B or two
// contiguous 16-B accesses per lane), deferred-reduction arithmetic for the dot products (gl_wide.hpp), wave-level
// DPP reductions then one LDS hop per workgroup; the workgroup that arrives last sums the per-workgroup partials.
#include <hip/hip_runtime.h>
#include <cstring>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <type_traits>
#include "kernels.hpp"
#include "gl_wide.hpp"

namespace hg {
namespace dev {

constexpr int TPB = 256;

// ------------------------------------------------------------------------------------------------
// Wave-level modular sum with DPP moves (VALU rate) instead of ds_bpermute: row_shr 1/2/4/8 inside the rows of 16,
// then row_bcast15 / row_bcast31 across rows; lane 63 ends up with the total, which v_readlane broadcasts.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
    u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, CTRL, ROW_MASK, 0xF, false);
    u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), CTRL, ROW_MASK, 0xF, false);
    return ((u64)hi << 32) | lo;   // lanes without a source (or outside ROW_MASK) read 0: the identity of gl_add
}
__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
    v = gl_add(v, dpp_u64<0x111, 0xF>(v));  // row_shr:1
    v = gl_add(v, dpp_u64<0x112, 0xF>(v));  // row_shr:2
    v = gl_add(v, dpp_u64<0x114, 0xF>(v));  // row_shr:4
    v = gl_add(v, dpp_u64<0x118, 0xF>(v));  // row_shr:8
    v = gl_add(v, dpp_u64<0x142, 0xA>(v));  // row_bcast:15 into rows 1, 3
    v = gl_add(v, dpp_u64<0x143, 0xC>(v));  // row_bcast:31 into rows 2, 3
    u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, 63), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), 63);
    return ((u64)hi << 32) | lo;
}
// total over the wave, in every lane
__device__ __forceinline__ E2 wave_sum(E2 v) { return e2(wave_sum_u64(v.c0), wave_sum_u64(v.c1)); }
// sums `v` over the workgroup; result valid in thread 0. `sm` holds TPB/64 E2 slots.
__device__ __forceinline__ E2 block_sum(E2 v, E2* sm) {
    v = wave_sum(v);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TPB / 64; w++) v = e2_add(v, sm[w]);
    }
    return v;
}

// launch shapes (the measured best on MI355X)
// (measured at n=32768 k=16: 32768 / 131072 / 262144 threads 2.05 / 2.04 / 2.14 against 1.97 ms; 1024 / 256 workgroups 1.99 / 2.11 against 1.97)
static size_t st_min_threads() { return 65536; }   // (with the pipelined round bodies of round 5: 32768 / 131072 / 262144 = 1.764 / 1.721 / 1.829 against 1.695 ms)
#ifndef HG_TEST_ST_MAX_BLOCKS
#define HG_TEST_ST_MAX_BLOCKS 512   // (a test build with 48: workgroups whose tiles straddle segment pairs - k_gp_first_hash_slot restages - and long tile loops)
#endif
static int st_max_blocks() { return HG_TEST_ST_MAX_BLOCKS; }   // (384 / 768 / 1024 in round 5: 1.894 / 1.785 / 1.812 against 1.797 ms; with the pipelined round bodies 1.729 / 1.704 / 1.678 against 1.695)

static inline int grid_for(size_t work_items) {
    size_t b = (work_items + TPB - 1) / TPB;
    if (b < 1) b = 1;
    return (int)(b > (size_t)SC_MAX_BLOCKS ? SC_MAX_BLOCKS : b);
}

// ------------------------------------------------------------------------------------------------
// Sum-check round, stride layout. KIND 0: g = p0 * sum_i M^i p_i (deg 2). KIND 1: g = p0 * sum_i gam^i p_2i p_2i+1 (deg 3).
template <typename T> struct Val;
template <> struct Val<u64> {
    static __device__ __forceinline__ u64 add(u64 a, u64 b) { return gl_add(a, b); }
    static __device__ __forceinline__ u64 sub(u64 a, u64 b) { return gl_sub(a, b); }
    static __device__ __forceinline__ u64 mul(u64 a, u64 b) { return gl_mul(a, b); }
    static __device__ __forceinline__ E2 scale(E2 c, u64 a) { return e2_mul_f(c, a); }        // c (E) * a
    static __device__ __forceinline__ E2 fold(u64 x, u64 d, E2 r) { return e2_add_f(e2_mul_f(r, d), x); }
    static __device__ __forceinline__ E2 lift(u64 a) { return e2(a, 0); }
    static __device__ __forceinline__ u64 zero() { return 0; }
};
template <> struct Val<E2> {
    static __device__ __forceinline__ E2 add(E2 a, E2 b) { return e2_add(a, b); }
    static __device__ __forceinline__ E2 sub(E2 a, E2 b) { return e2_sub(a, b); }
    static __device__ __forceinline__ E2 mul(E2 a, E2 b) { return e2_mul(a, b); }
    static __device__ __forceinline__ E2 scale(E2 c, E2 a) { return e2_mul(c, a); }
    static __device__ __forceinline__ E2 fold(E2 x, E2 d, E2 r) { return e2_add(x, e2_mul(r, d)); }
    static __device__ __forceinline__ E2 lift(E2 a) { return a; }
    static __device__ __forceinline__ E2 zero() { return e2_zero(); }
};

// Loads / stores of table entries go through the GLOBAL address space: a pointer read from a job descriptor is generic to hipcc, and a
// generic access is a flat_load / flat_store - counted by vmcnt AND lgkmcnt and returned out of order, so every wait behind one is
// `vmcnt(0) lgkmcnt(0)`, and every wait for a scalar load or an LDS read drains the table loads in flight as well. (No table these
// helpers touch lives in LDS except in k_st_tail, whose rounds run LDS -> LDS through sc_round_body: GIO = false keeps the generic form.)
typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
#define HG_GLOBAL(T, p) (reinterpret_cast<__attribute__((address_space(1))) T*>((unsigned long long)(p)))
template <bool GIO = true>
__device__ __forceinline__ u64x2_t load16(const void* p) {
    if constexpr (GIO) return *HG_GLOBAL(const u64x2_t, p);
    else return *reinterpret_cast<const u64x2_t*>(p);
}
template <bool GIO = true>
__device__ __forceinline__ void store16(void* p, u64 a, u64 b) {
    u64x2_t w; w.x = a; w.y = b;
    if constexpr (GIO) *HG_GLOBAL(u64x2_t, p) = w;
    else *reinterpret_cast<u64x2_t*>(p) = w;
}
template <typename T, bool GIO = true>
__device__ __forceinline__ void load_pair(const T* p, T& x, T& y) {
    if constexpr (std::is_same<T, u64>::value) {
        const u64x2_t v = load16<GIO>(p);
        x = v.x; y = v.y;
    } else {
        const u64x2_t a = load16<GIO>(p), b = load16<GIO>(p + 1);
        x = e2(a.x, a.y); y = e2(b.x, b.y);
    }
}
// Folded (E2) tables are stored DE-INTERLEAVED: logical entry i of a table of length 2h lives at (i & 1) * h + (i >> 1),
// so the pair (T[2j], T[2j+1]) the next round needs is (buf[j], buf[h + j]): both loads are 16 B per lane and
// contiguous across the wave (a 32-B lane stride would halve the useful bytes per load instruction).
// Round-0 inputs (level rows, node tables, bookkeeping tables) are in natural order.
template <typename T, bool NATURAL, bool GIO = true>
__device__ __forceinline__ void load_xy(const T* tab, size_t j, size_t half, T& x, T& y) {
    if constexpr (NATURAL || std::is_same<T, u64>::value) load_pair<T, GIO>(tab + 2 * j, x, y);
    else {
        const u64x2_t a = load16<GIO>(tab + j), b = load16<GIO>(tab + half + j);
        x = e2(a.x, a.y); y = e2(b.x, b.y);
    }
}
// position of logical entry j in a de-interleaved table of length `len`
__device__ __forceinline__ size_t dpos(size_t j, size_t len) { return (j & 1) * (len >> 1) + (j >> 1); }
template <bool GIO = true>
__device__ __forceinline__ void store_e2(E2* p, E2 v) { store16<GIO>(p, v.c0, v.c1); }
// streaming store for tables far larger than the caches (the first-round folds of the big layers): measured +5 % on that kernel
template <bool GIO = true>
__device__ __forceinline__ void store_e2_nt(E2* p, E2 v) {
    u64x2_t w; w.x = v.c0; w.y = v.c1;
    if constexpr (GIO) __builtin_nontemporal_store(w, HG_GLOBAL(u64x2_t, p));
    else *reinterpret_cast<u64x2_t*>(p) = w;
}
__device__ __forceinline__ void store_u64x2(u64* p, u64 a, u64 b) { store16<true>(p, a, b); }
__device__ __forceinline__ E2 gload_e2(const E2* p) {
    const u64x2_t v = *HG_GLOBAL(const u64x2_t, p);
    return e2(v.x, v.y);
}
__device__ __forceinline__ u64 gload_u64(const u64* p) { return *HG_GLOBAL(const u64, p); }
__device__ __forceinline__ void gstore_e2(E2* p, E2 v) { store_e2(p, v); }

// ---- first grand-product round of one table pair on base-field values ---------------------------------------------
// Accumulates gamma^i * (P0, P1, Pinf) with P0 = xl xr, P1 = yl yr, Pinf = (yl - xl)(yr - xr) (each base product reduced once,
// its two weighted copies unreduced), stores the folded left table multiplied by gamma^i and the folded right table, and -
// when `nxt` is given - the next product-tree level: P0 and P1 ARE its entries 2j, 2j+1 (Layer::up, prover.rs:332-354).
struct GpFirstAcc { WAcc a0, b0, a1, b1, ai, bi; };
__device__ __forceinline__ GpFirstAcc gp_first_acc_zero() {
    GpFirstAcc A;
    A.a0 = wacc_zero(); A.b0 = wacc_zero(); A.a1 = wacc_zero(); A.b1 = wacc_zero(); A.ai = wacc_zero(); A.bi = wacc_zero();
    return A;
}
template <bool WANT_Q, bool GIO = true>   // WANT_Q: the products of the two positions are always computed and handed back (slot form: the caller stores them)
__device__ __forceinline__ void gp_first_pair(GpFirstAcc& A, u64 xl, u64 yl, u64 xr, u64 yr, E2 gm, E2 gr, E2 r, bool summed,
                                              E2* __restrict__ out_l, E2* __restrict__ out_r, u64* __restrict__ nxt, u64& q0, u64& q1) {
    const u64 dl = gl_sub(yl, xl), dr = gl_sub(yr, xr);
    if (WANT_Q || summed || nxt) {
        // (reading v_l v_r from the tree level above instead of multiplying was measured slower: the
        // first round is bound by its 8-byte-element traffic, not by these products)
        q0 = gl_mul(xl, xr); q1 = gl_mul(yl, yr);
        if (nxt) store_u64x2(nxt, q0, q1);
        if (summed) {
            const u64 qi = gl_mul(dl, dr);
            wmac2(A.a0, gm.c0, q0, A.b0, gm.c1, q0);
            wmac2(A.a1, gm.c0, q1, A.b1, gm.c1, q1);
            wmac2(A.ai, gm.c0, qi, A.bi, gm.c1, qi);
        }
    }
    // gamma^i * (xl + r dl) = gamma^i xl + (gamma^i r) dl
    const WAcc f0 = wacc_pair_init(0, gm.c0, xl, gr.c0, dl), f1 = wacc_pair_init(0, gm.c1, xl, gr.c1, dl);
    store_e2_nt<GIO>(out_l, e2(wreduce(f0), wreduce(f1)));
    const WAcc h0 = wacc_mul_init(xr, r.c0, dr), h1 = wacc_mul_init(0, r.c1, dr);
    store_e2_nt<GIO>(out_r, e2(wreduce(h0), wreduce(h1)));
}
// product-tree entries (q0, q1) of positions 2j, 2j + 1 to every row named by `mask` (GpHashSrc::emit_rd, StJob::emit_mask)
__device__ __forceinline__ void emit_rows(u64* __restrict__ next_level, size_t row_stride, size_t at, u64 mask, ulonglong2 q) {
    while (mask) {
        const int t = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        store_u64x2(next_level + (size_t)t * row_stride + at, q.x, q.y);
    }
}
__device__ __forceinline__ void gp_first_acc_reduce(const GpFirstAcc& A, E2& s0, E2& s2, E2& s3) {
    s0 = e2(wreduce(A.a0), wreduce(A.b0)); s2 = e2(wreduce(A.a1), wreduce(A.b1)); s3 = e2(wreduce(A.ai), wreduce(A.bi));
}

// ---- first grand-product round on base-field rows: the whole round of one workgroup ---------------------------------------------
// (sc_round_body's FIRST / u64 / grand-product case.) The workgroup's (tile, pair) items form ONE stream: the four values, the two
// weights and the emit mask of item t+1 are requested before item t is computed on, all through the global address space, so that
// the waits are counted (`vmcnt(n)`) and nothing of item t waits for a store of item t-1. The first form of this loop asked for the
// values, waited, asked for the weights (generic loads), waited, computed, stored, asked for the emit mask and waited again, draining
// its own stores: three memory round trips in series per item, T = T_mem + T_valu (0.38 / 0.40 of the two roofs in round 5).
// (The weights as scalar loads - the pair index is uniform over a wave when 64 or more threads run along j - were tried: hipcc
// waits for a scalar load where it is issued, and that latency is then exposed once per item.)
template <bool SLOT, bool GIO>
__device__ __forceinline__ void gp_first_round_body(const u64* __restrict__ in, size_t in_stride, E2* __restrict__ out, size_t out_stride, int ntab, size_t half, E2 r,
                                                    const E2* __restrict__ pw, const E2* __restrict__ pwr, int jb_log2, E2* __restrict__ red, E2* acc,
                                                    size_t first_tile, size_t tile_step, bool p0_only, u64* __restrict__ next_level, const StJob* __restrict__ mirror) {
    const int BD = blockDim.x, tid = threadIdx.x;
    const int G = BD >> jb_log2;
    const int jj = tid & ((1 << jb_log2) - 1), g = tid >> jb_log2;
    const size_t ntiles = half >> jb_log2;
    const int nb = ntab >> 1;
    // (descriptor fields read once: behind the loop's stores hipcc reloads them every iteration)
    const E2* __restrict__ slotw = SLOT ? mirror->slotw : nullptr;
    // (no next level: the mask is loaded all the same - from the weights, any readable words - and ignored; see the loop)
    const u64* __restrict__ emask = !SLOT ? nullptr : next_level ? mirror->emit_mask : reinterpret_cast<const u64*>(mirror->slotw);
    const int slot_ng = SLOT ? mirror->slot_ng : 0, slot_shift = SLOT ? mirror->slot_shift : 0;
    struct Item { u64 xl, yl, xr, yr; E2 gm, gr; u64 em; };
    auto any_item = [] {   // "no value yet" without an instruction (and without a dependence on the loads of the item in flight)
        const u64 z = 0, u = __builtin_nondeterministic_value(z);
        return Item{u, u, u, u, e2(u, u), e2(u, u), u};
    };
    auto fetch = [&](size_t tile, int i, Item& it) {
        const size_t j = (tile << jb_log2) + jj;
        load_xy<u64, true, GIO>(in + (size_t)(2 * i) * in_stride, j, half, it.xl, it.yl);
        load_xy<u64, true, GIO>(in + (size_t)(2 * i + 1) * in_stride, j, half, it.xr, it.yr);
        it.em = 0;
        if constexpr (SLOT) {
            // (slot form: the group is uniform over the tile - a segment holds at least 512 positions)
            const size_t at = (size_t)i * slot_ng + (((tile << jb_log2) * 2) >> slot_shift);
            it.gm = gload_e2(slotw + 2 * at); it.gr = gload_e2(slotw + 2 * at + 1);
            it.em = gload_u64(emask + at);   // (always: see below)
        } else if constexpr (GIO) { it.gm = gload_e2(pw + i); it.gr = gload_e2(pwr + i); }
        else { it.gm = pw[i]; it.gr = pwr[i]; }   // (k_st_tail: the job descriptor is in LDS)
    };
    // Every iteration issues the SAME loads whatever the item (the last one asks for the thread's first item again): the waits
    // hipcc inserts are the minimum over all paths to them, and one path without the next item's loads makes every wait for the
    // current item's values a wait for the next item's as well.
    const bool active = g < nb && first_tile < ntiles;
    Item cur = any_item();
    if (active) fetch(first_tile, g, cur);
    for (size_t tile = first_tile; tile < ntiles; tile += tile_step) {
        const size_t j = (tile << jb_log2) + jj;
        const size_t jo = dpos(j, half);
        GpFirstAcc A = gp_first_acc_zero();
        u64 p0 = 0, p2 = 0, p3 = 0;
        if (active) for (int i = g; i < nb; i += G) {
            const bool more_i = i + G < nb, more_t = tile + tile_step < ntiles;
            Item nxt;
            fetch(more_i || !more_t ? tile : tile + tile_step, more_i ? i + G : g, nxt);
            if (i == 0) { const u64 dl = gl_sub(cur.yl, cur.xl); p0 = cur.xl; p2 = gl_add(cur.yl, dl); p3 = gl_add(p2, dl); }
            u64 q0, q1;
            if constexpr (SLOT) {
                gp_first_pair<true, GIO>(A, cur.xl, cur.yl, cur.xr, cur.yr, cur.gm, cur.gr, r, !(p0_only && i == 0), out + (size_t)(2 * i) * out_stride + jo,
                                         out + (size_t)(2 * i + 1) * out_stride + jo, nullptr, q0, q1);
                if (next_level) emit_rows(next_level, in_stride, 2 * j, cur.em, make_ulonglong2(q0, q1));
            } else {
                gp_first_pair<false, GIO>(A, cur.xl, cur.yl, cur.xr, cur.yr, cur.gm, cur.gr, r, !(p0_only && i == 0), out + (size_t)(2 * i) * out_stride + jo,
                                          out + (size_t)(2 * i + 1) * out_stride + jo, next_level ? next_level + (size_t)i * in_stride + 2 * j : nullptr, q0, q1);
            }
            cur = nxt;
        }
        E2 s0, s2, s3;
        gp_first_acc_reduce(A, s0, s2, s3);
        if (G > 1) {
            red[tid] = s0; red[BD + tid] = s2; red[2 * BD + tid] = s3;
            __syncthreads();
            if (g == 0) {
                const int gmax = G < nb ? G : nb;
                for (int gg = 1; gg < gmax; gg++) {
                    int o = (gg << jb_log2) + jj;
                    s0 = e2_add(s0, red[o]); s2 = e2_add(s2, red[BD + o]); s3 = e2_add(s3, red[2 * BD + o]);
                }
            }
            __syncthreads();
        }
        if (g == 0) {
            // (s0, s2, s3) hold (sum P0, sum P1, sum Pinf): q(2) = 2 P1 - P0 + 2 Pinf, q(3) = 3 P1 - 2 P0 + 6 Pinf
            const E2 P1x2 = e2_dbl(s2), Pix2 = e2_dbl(s3);
            const E2 q2 = e2_add(e2_sub(P1x2, s0), Pix2);
            const E2 q3 = e2_add(e2_sub(e2_add(P1x2, s2), e2_dbl(s0)), e2_add(e2_dbl(Pix2), Pix2));
            acc[0] = e2_add(acc[0], e2_mul_f(s0, p0));
            acc[1] = e2_add(acc[1], e2_mul_f(q2, p2));
            acc[2] = e2_add(acc[2], e2_mul_f(q3, p3));
        }
    }
}

struct GpItemE2 { E2 xl, yl, xr, yr; };
__device__ __forceinline__ GpItemE2 gp_item_any() {   // "no value yet" without an instruction
    const u64 z = 0, u = __builtin_nondeterministic_value(z);
    return GpItemE2{e2(u, u), e2(u, u), e2(u, u), e2(u, u)};
}
// ---- one sum-check round as a device function ---------------------------------------------------
// Thread mapping inside a workgroup of BD threads: JB = 2^jb_log2 threads along the pair index j
// (coalesced) times G = BD/JB groups along the table index i. Large rounds use JB = BD (one thread per
// pair, serial loop over tables); small rounds trade j-parallelism for i-parallelism so that the
// serial per-thread loop over the tables (a dependent chain of HBM/L2 latencies) shrinks to
// ceil(nb / G) iterations. Partial sums over the groups are combined through LDS (`red`: NV*BD E2)
// BEFORE the multiplication by p_0, which is what the quirky g = p_0 * (sum_i ..) needs.
// Grand-product shape: in the FIRST round of a job the folded LEFT table of pair i is stored multiplied by
// gamma^i (pw[i]); later rounds then need no per-pair scaling at all (the weight rides along in the table),
// which removes 3 of the 8 extension multiplications per (pair, j). The host divides the final left
// evaluations by gamma^i again before they reach the transcript.
// MODE (grand product, later rounds on Ext2 tables only; round 6): 0 = the round; 1 = its folds only (the next round needs nothing else: the
// challenges are known before the proof), 2 = its sums only - a small round is one workgroup's dependent chain, and the chain of a
// fold-only launch is a third of the whole round's; the sums of every split round then run in one launch beside the rounds that follow
template <int KIND, typename T, bool FIRST, bool SLOT = false, bool GIO = true, int MODE = 0>   // SLOT: first round of a slot-form job (StJob::slotw), passed as `mirror`; GIO = false: tables behind generic pointers (k_st_tail: LDS)
__device__ __forceinline__ void sc_round_body(const T* __restrict__ in, size_t in_stride, E2* __restrict__ out,
                                              size_t out_stride, int ntab, size_t half, E2 r, const E2* __restrict__ pw, const E2* __restrict__ pwr,
                                              int jb_log2, E2* __restrict__ red, E2* acc, size_t first_tile, size_t tile_step,
                                              bool p0_only = false, u64* __restrict__ next_level = nullptr, const StJob* __restrict__ mirror = nullptr) {
    using V = Val<T>;
    if constexpr (KIND == SC_GRANDPROD && FIRST && std::is_same<T, u64>::value) {
        gp_first_round_body<SLOT, GIO>(in, in_stride, out, out_stride, ntab, half, r, pw, pwr, jb_log2, red, acc, first_tile, tile_step, p0_only, next_level, mirror);
        return;
    }
    static_assert(MODE == 0 || (KIND == SC_GRANDPROD && !FIRST && std::is_same<T, E2>::value), "split rounds: grand product, Ext2 tables");
    const int BD = blockDim.x, tid = threadIdx.x;
    const int G = BD >> jb_log2;
    const int jj = tid & ((1 << jb_log2) - 1), g = tid >> jb_log2;
    const size_t ntiles = half >> jb_log2;  // host guarantees 2^jb_log2 <= half
    [[maybe_unused]] GpItemE2 gp_cur = gp_item_any();   // (grand product, later rounds: the item in flight across the tile loop)
    for (size_t tile = first_tile; tile < ntiles; tile += tile_step) {
        const size_t j = (tile << jb_log2) + jj;
        const size_t jo = dpos(j, half);  // this round's output (length `half`) is written de-interleaved
        if constexpr (KIND == SC_GRANDPROD) {
            E2 s0 = e2_zero(), s2 = e2_zero(), s3 = e2_zero();
            T p0 = V::zero(), p2 = V::zero(), p3 = V::zero();
            const int nb = ntab >> 1;
            // a(t) b(t) = P0 + t (P1 - P0 - Pinf) + t^2 Pinf with P0 = a(0)b(0), P1 = a(1)b(1), Pinf = (a1-a0)(b1-b0):
            // the three dot products over the table pairs stay unreduced in column accumulators (gl_wide.hpp)
            // and are reduced once per j after the loop.
            if constexpr (!FIRST && std::is_same<T, E2>::value) {
                WE2 w0 = we2_zero(), w1 = we2_zero(), wi = we2_zero();
                const FoldR fr = fold_r(r);
                // software pipeline over the workgroup's whole (tile, pair) stream: the four loads of the NEXT item are requested before
                // this one is computed on, into registers of their own (a copy of the current item's would wait for them and for the
                // stores behind them first), and every iteration requests the same loads (the last item asks for the thread's first
                // again): hipcc's waits are the minimum over all paths to them (gp_first_round_body).
                auto fetch = [&](size_t tl, int i, GpItemE2& it) {
                    const size_t jn = (tl << jb_log2) + jj;
                    load_xy<E2, false, GIO>(in + (size_t)(2 * i) * in_stride, jn, half, it.xl, it.yl);
                    load_xy<E2, false, GIO>(in + (size_t)(2 * i + 1) * in_stride, jn, half, it.xr, it.yr);
                };
                if (g < nb && tile == first_tile) fetch(tile, g, gp_cur);
                if (g < nb) for (int i = g; i < nb; i += G) {
                    const bool more_i = i + G < nb, more_t = tile + tile_step < ntiles;
                    GpItemE2 nxt;
                    fetch(more_i || !more_t ? tile : tile + tile_step, more_i ? i + G : g, nxt);
                    const E2 xl = gp_cur.xl, yl = gp_cur.yl, xr = gp_cur.xr, yr = gp_cur.yr;
                    E2 dl = e2_sub(yl, xl), dr = e2_sub(yr, xr);
                    if (i == 0) { p0 = xl; p2 = e2_add(yl, dl); p3 = e2_add(p2, dl); }
                    if (MODE != 1 && !(p0_only && i == 0)) {  // a p0-only pair 0 belongs to another rank's share of the batch
                        we2_mac(w0, xl, xr);
                        we2_mac(w1, yl, yr);
                        we2_mac(wi, dl, dr);
                    }
                    if (MODE != 2) {
                        store_e2<GIO>(out + (size_t)(2 * i) * out_stride + jo, e2_fold_wide(xl, dl, fr));
                        store_e2<GIO>(out + (size_t)(2 * i + 1) * out_stride + jo, e2_fold_wide(xr, dr, fr));
                    }
                    gp_cur = nxt;
                }
                if (MODE != 1) { s0 = we2_reduce(w0); s2 = we2_reduce(w1); s3 = we2_reduce(wi); }
                if (mirror && g == 0) {  // the linear table S of a mirrored job (StJob::mirror): K1 S(t) + K2 joins P0 and P1
                    E2 x, y;
                    load_xy<E2, false, GIO>(in + (size_t)(2 * nb) * in_stride, j, half, x, y);
                    if (MODE != 1) {
                        s0 = e2_add(s0, e2_add(e2_mul(mirror->mk1, x), mirror->mk2));
                        s2 = e2_add(s2, e2_add(e2_mul(mirror->mk1, y), mirror->mk2));
                    }
                    if (MODE != 2) store_e2<GIO>(out + (size_t)(2 * nb) * out_stride + jo, e2_fold_wide(x, e2_sub(y, x), fr));
                }
            } else {
            for (int i = g; i < nb; i += G) {
                T xl, yl, xr, yr;
                load_xy<T, FIRST, GIO>(in + (size_t)(2 * i) * in_stride, j, half, xl, yl);
                load_xy<T, FIRST, GIO>(in + (size_t)(2 * i + 1) * in_stride, j, half, xr, yr);
                T dl = V::sub(yl, xl), dr = V::sub(yr, xr);
                if (i == 0) { p0 = xl; p2 = V::add(yl, dl); p3 = V::add(p2, dl); }
                const bool summed = !(p0_only && i == 0);
                E2 gm = pw[i];
                if (summed) {
                    s0 = e2_add(s0, V::scale(gm, V::mul(xl, xr)));
                    s2 = e2_add(s2, V::scale(gm, V::mul(yl, yr)));
                    s3 = e2_add(s3, V::scale(gm, V::mul(dl, dr)));
                }
                // gamma^i * (xl + r dl) = gamma^i xl + (gamma^i r) dl
                store_e2<GIO>(out + (size_t)(2 * i) * out_stride + jo, e2_add(V::scale(gm, xl), V::scale(pwr[i], dl)));
                store_e2<GIO>(out + (size_t)(2 * i + 1) * out_stride + jo, V::fold(xr, dr, r));
            }
            }
            if constexpr (MODE == 1) continue;   // (folds only: no sums)
            if (G > 1) {
                red[tid] = s0; red[BD + tid] = s2; red[2 * BD + tid] = s3;
                __syncthreads();
                if (g == 0) {
                    const int gmax = G < nb ? G : nb;
                    for (int gg = 1; gg < gmax; gg++) {
                        int o = (gg << jb_log2) + jj;
                        s0 = e2_add(s0, red[o]); s2 = e2_add(s2, red[BD + o]); s3 = e2_add(s3, red[2 * BD + o]);
                    }
                }
                __syncthreads();
            }
            if (g == 0) {
                // (s0, s2, s3) hold (sum P0, sum P1, sum Pinf): q(2) = 2 P1 - P0 + 2 Pinf, q(3) = 3 P1 - 2 P0 + 6 Pinf
                E2 P0 = s0, P1 = s2, Pi = s3;
                E2 P1x2 = e2_dbl(P1), Pix2 = e2_dbl(Pi);
                E2 q2 = e2_add(e2_sub(P1x2, P0), Pix2);
                E2 q3 = e2_add(e2_sub(e2_add(P1x2, P1), e2_dbl(P0)), e2_add(e2_dbl(Pix2), Pix2));
                acc[0] = e2_add(acc[0], V::scale(P0, p0));
                acc[1] = e2_add(acc[1], V::scale(q2, p2));
                acc[2] = e2_add(acc[2], V::scale(q3, p3));
            }
        } else {
            T s0 = V::zero(), s2 = V::zero(), p0 = V::zero(), p2 = V::zero();
            // Collation shape: like the grand product, the FIRST round stores table i multiplied by its weight M^i
            // (pw[i], a base-field constant; pwr[i] = M^i r), so later rounds only add: s(t) = sum_i table_i(t).
            if constexpr (FIRST && std::is_same<T, u64>::value) {
                WAcc w0 = wacc_zero(), w2 = wacc_zero();
                for (int i = g; i < ntab; i += G) {
                    u64 x, y;
                    load_xy<u64, true, GIO>(in + (size_t)i * in_stride, j, half, x, y);
                    u64 d = gl_sub(y, x);
                    u64 v2 = gl_add(y, d);
                    if (i == 0) { p0 = x; p2 = v2; }
                    const u64 m = pw[i].c0;
                    const E2 mr = pwr[i];
                    if (!(p0_only && i == 0)) wmac2(w0, m, x, w2, m, v2);  // (a p0-only table 0 belongs to another rank's share)
                    const WAcc f0 = wacc_pair_init(0, m, x, mr.c0, d), f1 = wacc_mul_init(0, mr.c1, d);
                    store_e2<GIO>(out + (size_t)i * out_stride + jo, e2(wreduce(f0), wreduce(f1)));
                }
                s0 = wreduce(w0); s2 = wreduce(w2);
            } else if constexpr (!FIRST && std::is_same<T, E2>::value) {
                const FoldR fr = fold_r(r);
                for (int i = g; i < ntab; i += G) {
                    E2 x, y;
                    load_xy<E2, false, GIO>(in + (size_t)i * in_stride, j, half, x, y);
                    E2 d = e2_sub(y, x);
                    E2 v2 = e2_add(y, d);
                    if (i == 0) { p0 = x; p2 = v2; }
                    if (!(p0_only && i == 0)) { s0 = e2_add(s0, x); s2 = e2_add(s2, v2); }
                    store_e2<GIO>(out + (size_t)i * out_stride + jo, e2_fold_wide(x, d, fr));
                }
            } else {
            for (int i = g; i < ntab; i += G) {
                T x, y;
                load_xy<T, FIRST, GIO>(in + (size_t)i * in_stride, j, half, x, y);
                T d = V::sub(y, x);
                T v2 = V::add(y, d);
                if (i == 0) { p0 = x; p2 = v2; }
                const bool summed = !(p0_only && i == 0);
                if constexpr (FIRST) {
                    u64 m = pw[i].c0;
                    if constexpr (std::is_same<T, u64>::value) {
                        if (summed) { s0 = gl_add(s0, gl_mul(m, x)); s2 = gl_add(s2, gl_mul(m, v2)); }
                        store_e2<GIO>(out + (size_t)i * out_stride + jo, e2_add_f(e2_mul_f(pwr[i], d), gl_mul(m, x)));
                    } else {
                        if (summed) { s0 = e2_add(s0, e2_mul_f(x, m)); s2 = e2_add(s2, e2_mul_f(v2, m)); }
                        store_e2<GIO>(out + (size_t)i * out_stride + jo, e2_mul_f(V::fold(x, d, r), m));
                    }
                } else {
                    if (summed) { s0 = V::add(s0, x); s2 = V::add(s2, v2); }
                    store_e2<GIO>(out + (size_t)i * out_stride + jo, V::fold(x, d, r));
                }
            }
            }
            E2 t0 = V::lift(s0), t2 = V::lift(s2);
            if (G > 1) {
                red[tid] = t0; red[BD + tid] = t2;
                __syncthreads();
                if (g == 0) {
                    const int gmax = G < ntab ? G : ntab;
                    for (int gg = 1; gg < gmax; gg++) {
                        int o = (gg << jb_log2) + jj;
                        t0 = e2_add(t0, red[o]); t2 = e2_add(t2, red[BD + o]);
                    }
                }
                __syncthreads();
            }
            if (g == 0) {
                acc[0] = e2_add(acc[0], e2_mul(V::lift(p0), t0));
                acc[1] = e2_add(acc[1], e2_mul(V::lift(p2), t2));
            }
        }
    }
}

// sums `v` over a workgroup of any size (multiple of 64, <= 1024); result valid in thread 0
__device__ __forceinline__ E2 block_sum_n(E2 v, E2* sm /* >= 16 */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) for (int w = 1; w < nw; w++) v = e2_add(v, sm[w]);
    return v;
}

constexpr int SM_SLOTS = 96;      // block-sum scratch at the start of dynamic LDS: 16 waves x up to 6 values
extern __shared__ E2 dyn_lds[];  // [SM_SLOTS block-sum slots][kernel-specific: NV * BD reduction slots, chunk tables, ...]
// sums N values over the workgroup with one pair of barriers; results valid in thread 0
template <int N>
__device__ __forceinline__ void block_sum_multi(E2 (&v)[N], E2* sm /* >= 16 * N */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; k++) sm[wave * N + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < nw; w++) {
#pragma unroll
            for (int k = 0; k < N; k++) v[k] = e2_add(v[k], sm[w * N + k]);
        }
}

// ---- per-workgroup partial sums, finished by the workgroup that arrives last --------------------------------------
// Partials cross workgroups (and XCDs, whose L2s are not coherent with each other) as relaxed agent-scope atomics:
// write-through stores, L2-bypassing loads, no cache-wide write-back or invalidate.
__device__ __forceinline__ void part_store(E2* p, E2 v) {
    __hip_atomic_store(&p->c0, v.c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&p->c1, v.c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ E2 part_load(const E2* p) {
    return e2(__hip_atomic_load(&p->c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&p->c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ unsigned* tickets_of(E2* partials) { return reinterpret_cast<unsigned*>(partials + PARTIALS_E2); }
// Called by the whole workgroup after thread 0 part_store()d its `per` values at p[blockIdx.x * per ..]. In the
// workgroup that arrives last (block-uniform), sums the nblocks partials of each value and stores them to out[0..per).
struct NoPost { __device__ __forceinline__ void operator()(E2*) const {} };
// `post` (thread 0 of the last workgroup) may turn the `per` <= 6 summed values into what is stored (k_ps_step2<true>: the sums are
// inner products the round's four values are combinations of - formed once per job here instead of once per workgroup)
template <typename Post = NoPost>
__device__ __forceinline__ void finish_partials(E2* p, int per, unsigned* ticket, E2* __restrict__ out, E2* sm /* >= 16 */,
                                                int nblocks = -1, Post post = Post()) {
    __shared__ unsigned s_last;
    if (nblocks < 0) nblocks = (int)gridDim.x;
    if (threadIdx.x == 0) {
#if !defined(HG_STRICT_TICKETS) && defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "the vmcnt-ordered ticket below is only valid on gfx942 / gfx950 (stores tracked by vmcnt, sc1 stores written through); build other architectures with -DHG_STRICT_TICKETS"
#endif
#ifdef HG_STRICT_TICKETS
        // C++-memory-model form: release on the ticket (pairs with the last arriver's acquire). On gfx950 an agent-scope
        // release is `buffer_wbl2 sc1` - a write-back of every dirty line of this XCD's L2, i.e. of the folded tables this very
        // kernel is streaming out - once per workgroup: measured +0.8 ms on a 5.2 ms prove (profiles/r02_a_tickets_ab.txt).
        unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
        // Default form. The partials are written by THIS thread with agent-scope atomic stores (part_store: `global_store ...
        // sc1`, written through to the agent coherence point, never left dirty in the L2), so the only thing the ticket must
        // be ordered after is the completion of those stores: `s_waitcnt vmcnt(0)` - the same vmcnt-tracked completion the
        // compiler's own release sequence waits on after its write-back. The cache-wide write-back a formal release adds is
        // only needed for plain (non-atomic) stores, and no plain store of this kernel is read through the ticket.
        // The compiler barrier keeps the stores, the wait and the ticket RMW in program order.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        s_last = (t == (unsigned)nblocks - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    // acquire side (once per launch, cheap: `buffer_inv sc1`): the last workgroup must not see stale lines of the partials
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    constexpr int PER_MAX = 6;
    for (int v0 = 0; v0 < per; v0 += PER_MAX) {
        E2 a[PER_MAX];
#pragma unroll
        for (int v = 0; v < PER_MAX; v++) a[v] = e2_zero();
        for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
            E2 t[PER_MAX];
#pragma unroll
            for (int v = 0; v < PER_MAX; v++) if (v0 + v < per) t[v] = part_load(p + (size_t)b * per + v0 + v);  // all loads in flight together
#pragma unroll
            for (int v = 0; v < PER_MAX; v++) if (v0 + v < per) a[v] = e2_add(a[v], t[v]);
        }
#pragma unroll
        for (int v = 0; v < PER_MAX; v++) if (v0 + v < per) a[v] = block_sum_n(a[v], sm);
        if (threadIdx.x == 0) {
            if constexpr (!std::is_same<Post, NoPost>::value) post(a);
#pragma unroll
            for (int v = 0; v < PER_MAX; v++) if (v0 + v < per) out[v0 + v] = a[v];
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
}

// ---- stride-layout sum-check jobs, batched over grid.y ---------------------------------------------------
// Round rd of job J reads tables of length 2h (h = 2^(nvars-1-rd)) at stride 2h (round 0: J.in / J.in_stride)
// and writes the folded tables at stride h into the ping-pong buffers (last round: J.final_out, stride 1).
__device__ __forceinline__ void st_io(const StJob& J, int rd, const void*& in, size_t& in_stride, E2*& out) {
    const size_t h = (size_t)1 << (J.nvars - 1 - rd);
    if (rd == 0) { in = J.in; in_stride = J.in_stride; }
    else { in = J.buf[(rd - 1) & 1]; in_stride = 2 * h; }
    out = rd == J.nvars - 1 ? J.final_out : J.buf[rd & 1];
}

// The items of a launch share a 1-D grid: item y owns workgroups [blk0, blk0 + nblk).
// (a launch holds at most 64 items: ONE 64-lane load of the items' first workgroups and a ballot, instead of a binary search whose six
// dependent loads stood at the head of every workgroup of every round launch)
__device__ __forceinline__ int find_item(const StItem* __restrict__ items, int nitems, int blk) {
    const int l = threadIdx.x & 63;
    const int b0 = l < nitems ? items[l].blk0 : 0x7fffffff;
    const unsigned long long m = __ballot(b0 <= blk);
    return __builtin_amdgcn_readfirstlane(__popcll(m) - 1);
}
// one step: every item runs its job's round with half = 2^item.h_log2
template <int KIND, typename T, bool SLOT = false, int MODE = 0>   // SLOT: the first round of ONE slot-form job (StJob::slotw); MODE: sc_round_body
__global__ __launch_bounds__(256) void k_st_step(const StJob* __restrict__ jobs, const StItem* __restrict__ items, int nitems,
                                                 const E2* __restrict__ chal, E2* __restrict__ partials, E2* __restrict__ res) {
    constexpr int NV = KIND == SC_GRANDPROD ? 3 : 2;
    const int y = find_item(items, nitems, blockIdx.x);
    const StItem& I = items[y];
    const StJob& J = jobs[I.job];
    const int h_log2 = I.h_log2, jb_log2 = I.jb_log2;
    const int rd = J.nvars - 1 - h_log2;
    const size_t half = (size_t)1 << h_log2;
    const int nblocks = I.nblk, bx = (int)blockIdx.x - I.blk0;
    const void* in = I.in; const size_t in_stride = I.in_stride; E2* out = I.out;
    E2* sm = dyn_lds;
    E2* red = dyn_lds + SM_SLOTS;
    E2 acc[NV];
#pragma unroll
    for (int t = 0; t < NV; t++) acc[t] = e2_zero();
    if (rd == 0) sc_round_body<KIND, T, true, SLOT>(reinterpret_cast<const T*>(in), in_stride, out, half, J.ntab, half, chal[J.r_off + rd], J.pw, J.pwr, jb_log2, red, acc, bx, nblocks, J.p0_only != 0, J.next_level,
                                                    SLOT ? &J : nullptr);
    else sc_round_body<KIND, T, false, false, true, MODE>(reinterpret_cast<const T*>(in), in_stride, out, half, J.ntab, half, chal[J.r_off + rd], J.pw, J.pwr, jb_log2, red, acc, bx, nblocks, J.p0_only != 0, nullptr,
                                                          J.mirror ? &J : nullptr);
    if constexpr (MODE == 1) return;
    E2* part = partials + (size_t)y * SC_MAX_BLOCKS * NV;
    block_sum_multi<NV>(acc, sm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int t = 0; t < NV; t++) {
            if (nblocks == 1) res[J.sums_slot + (size_t)rd * NV + t] = acc[t];
            else part_store(part + (size_t)bx * NV + t, acc[t]);
        }
    }
    if (nblocks > 1) finish_partials(part, NV, tickets_of(partials) + y * 32, res + J.sums_slot + (size_t)rd * NV, sm, nblocks);
}
// ---- first round of grand product #1's top layer straight from the Lasso integer tables ----------------------------------
// The level-0 rows (the 2 alpha multiset-hash tables of 2^nu entries, 838 MB at n=32768 k=16) are never materialised: thread j
// recomputes h = dim + E gamma + ts gamma^2 - tau at the four indices 2j, 2j+1, N/2 + 2j, N/2 + 2j+1 it needs (v_l, v_r are the
// two halves of a row), once per MEMORY for its read pair and its write pair (write hash = read hash + gamma^2), with the
// dim / ts part shared by the memories of a chunk. E, dim, ts are small integers (< 2^16, < 2^16, < 2^32): gl_mul_small.
// It also emits product-tree level 1 (J.next_level), so no separate hash or level-1 pass exists.
// SLOT: slot form (GpHashSrc::slot_of; always mirrored)
template <bool MIRROR, bool RECOMP, bool SLOT = false>   // MIRROR: write rows = read rows + gamma^2: not stored, not multiplied (StJob::mirror); RECOMP: E from the limbs
#ifndef HG_HASH_SLOT_WAVES
#define HG_HASH_SLOT_WAVES 3
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SLOT ? HG_HASH_SLOT_WAVES : 3, SLOT ? HG_HASH_SLOT_WAVES : 3))) void k_gp_first_hash(const StJob* __restrict__ job, const StItem* __restrict__ item,
                                                       const E2* __restrict__ chal, E2* __restrict__ partials, E2* __restrict__ res) {
    const StJob& J = *job;
    const StItem& I = *item;
    const GpHashSrc& H = *J.hash_src;
    const int h_log2 = I.h_log2;
    const size_t half = (size_t)1 << h_log2;      // pair indices j of this round; rows have 4 * half entries (v_l | v_r)
    const size_t hN = half << 1;                  // N / 2: offset of v_r inside a row
    const size_t ntiles = half >> 8;
    const int nblocks = I.nblk, bx = (int)blockIdx.x;
    E2* __restrict__ out = I.out;
    const E2 r = chal[J.r_off];
    const bool p0_only = J.p0_only != 0;
    [[maybe_unused]] constexpr bool mirror = MIRROR;
    const u64 gamma = H.gamma, gamma2 = H.gamma2, tau = H.tau;
    E2 acc[3] = {e2_zero(), e2_zero(), e2_zero()};
    for (size_t tile = bx; tile < ntiles; tile += nblocks) {
        const size_t j = (tile << 8) + threadIdx.x;
        const size_t jo = dpos(j, half);
        GpFirstAcc A = gp_first_acc_zero();
        E2 Sx = e2_zero(), Sy = e2_zero();  // mirror: S at 0 and 1, S = sum_i w_i (l_i + r_i) (reduced sums: four column accumulators would cost a wave of occupancy)
        u64 p0 = 0, p2 = 0, p3 = 0;
        int cur_chunk = -1;
        u64 c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        u32 a01 = 0, a23 = 0;          // RECOMP: the chunk's 16-bit limbs at 2j, 2j+1 | N/2+2j, N/2+2j+1
        u32 uses_lo = 0, uses_hi = 0;  // RECOMP: memories the two rows' lookups use (alpha <= 32)
        if constexpr (RECOMP) {
            if (2 * j < H.rows) uses_lo = (u32)H.lookup_uses[H.seg_lookup[(2 * j) >> H.seg_shift]];
            if (hN + 2 * j < H.rows) uses_hi = (u32)H.lookup_uses[H.seg_lookup[(hN + 2 * j) >> H.seg_shift]];
        }
        [[maybe_unused]] const int sp = SLOT ? (int)((tile << 9) >> H.seg_shift) : 0;   // slot form: this tile's segment pair (uniform: 512 positions of one segment)
        // slot form: only the memories that represent a joint class in this segment pair (in ascending order, so still chunk by chunk)
        for (int m = 0; m < (SLOT ? H.nslots : H.nmem); m++) {
            [[maybe_unused]] int slot_v = m;
            int mi = m;
            if constexpr (SLOT) {
                mi = H.rep[(size_t)m * H.npairs + sp];
                if (mi == 255) {   // fewer classes here than the job has table pairs: the rest is zero
                    store_e2_nt(out + (size_t)(2 * m) * half + jo, e2_zero());
                    store_e2_nt(out + (size_t)(2 * m + 1) * half + jo, e2_zero());
                    continue;
                }
            }
            const GpHashMem M = H.mems[mi];
            if (M.chunk != cur_chunk) {  // uniform: memories are listed chunk by chunk
                cur_chunk = M.chunk;
                const u64* __restrict__ dim = H.dim[cur_chunk];
                const u64* __restrict__ ts = H.ts[cur_chunk];
                const ulonglong2 dl = *reinterpret_cast<const ulonglong2*>(dim + 2 * j), dh = *reinterpret_cast<const ulonglong2*>(dim + hN + 2 * j);
                const ulonglong2 tl = *reinterpret_cast<const ulonglong2*>(ts + 2 * j), th = *reinterpret_cast<const ulonglong2*>(ts + hN + 2 * j);
                c0 = gl_sub(gl_add(dl.x, gl_mul_small(gamma2, (u32)tl.x)), tau); c1 = gl_sub(gl_add(dl.y, gl_mul_small(gamma2, (u32)tl.y)), tau);
                c2 = gl_sub(gl_add(dh.x, gl_mul_small(gamma2, (u32)th.x)), tau); c3 = gl_sub(gl_add(dh.y, gl_mul_small(gamma2, (u32)th.y)), tau);
                if constexpr (RECOMP) { a01 = (u32)dl.x | ((u32)dl.y << 16); a23 = (u32)dh.x | ((u32)dh.y << 16); }
            }
            // (prefetching the next memory's E loads was measured slower here: 594 vs 560 us)
            u32 e0, e1, e2v, e3;
            if constexpr (RECOMP) {   // T_s[a] = a below the cutoff, 0 above; 0 for rows whose lookup does not use the memory
                const u32 cut = M.cutoff;
                const bool ul = (uses_lo >> M.mem) & 1, uh = (uses_hi >> M.mem) & 1;
                e0 = a01 & 0xFFFF; e1 = a01 >> 16; e2v = a23 & 0xFFFF; e3 = a23 >> 16;
                e0 = (ul && e0 < cut) ? e0 : 0; e1 = (ul && e1 < cut) ? e1 : 0;
                e2v = (uh && e2v < cut) ? e2v : 0; e3 = (uh && e3 < cut) ? e3 : 0;
            } else {
                const ulonglong2 el = *reinterpret_cast<const ulonglong2*>(M.ep + 2 * j), eh = *reinterpret_cast<const ulonglong2*>(M.ep + hN + 2 * j);
                e0 = (u32)el.x; e1 = (u32)el.y; e2v = (u32)eh.x; e3 = (u32)eh.y;
            }
            u64 xl = gl_add(c0, gl_mul_small(gamma, e0)), yl = gl_add(c1, gl_mul_small(gamma, e1));
            u64 xr = gl_add(c2, gl_mul_small(gamma, e2v)), yr = gl_add(c3, gl_mul_small(gamma, e3));
            if constexpr (SLOT) {
                // the class's table pair, weighted with the class weight; tree level 1 of the read and of the write row (+ gamma^2:
                // t + 1) go to every row that equals this class's here
                const int i = M.rd_row;
                if (i == 0) { const u64 d = gl_sub(yl, xl); p0 = xl; p2 = gl_add(yl, d); p3 = gl_add(p2, d); }
                const size_t at = (size_t)slot_v * H.npairs + sp;
                const E2 gm = H.slotw[2 * at], gr = H.slotw[2 * at + 1];
                u64 q0, q1;
                const bool summed = !(p0_only && slot_v == 0);   // (sharded: class 0 = pair 0, held for p_0 only)
                gp_first_pair<true>(A, xl, yl, xr, yr, gm, gr, r, summed, out + (size_t)(2 * slot_v) * half + jo, out + (size_t)(2 * slot_v + 1) * half + jo, nullptr, q0, q1);
                if (summed) {
                    const u64 hx = gl_add(xl, xr), hy = gl_add(yl, yr);
                    Sx = e2_add(Sx, e2_mul_f(gm, hx));
                    Sy = e2_add(Sy, e2_mul_f(gm, hy));
                }
                if (J.next_level) {
                    emit_rows(J.next_level, hN, 2 * j, H.emit_rd[at], make_ulonglong2(q0, q1));
                    xl = gl_add(xl, gamma2); yl = gl_add(yl, gamma2); xr = gl_add(xr, gamma2); yr = gl_add(yr, gamma2);
                    emit_rows(J.next_level, hN, 2 * j, H.emit_wr[at], make_ulonglong2(gl_mul(xl, xr), gl_mul(yl, yr)));
                }
            } else {
            if (M.rd_row >= 0) {
                const int i = M.rd_row;
                if (i == 0) { const u64 d = gl_sub(yl, xl); p0 = xl; p2 = gl_add(yl, d); p3 = gl_add(p2, d); }
                u64 q0, q1;
                gp_first_pair<false>(A, xl, yl, xr, yr, J.pw[i], J.pwr[i], r, !(p0_only && i == 0), out + (size_t)(2 * i) * half + jo,
                                     out + (size_t)(2 * i + 1) * half + jo, J.next_level ? J.next_level + (size_t)i * hN + 2 * j : nullptr, q0, q1);
                if constexpr (MIRROR) if (!(p0_only && i == 0)) {
                    const E2 gm = J.pw[i];
                    const u64 hx = gl_add(xl, xr), hy = gl_add(yl, yr);
                    Sx = e2_add(Sx, e2_mul_f(gm, hx));
                    Sy = e2_add(Sy, e2_mul_f(gm, hy));
                }
            }
            if (M.wr_row >= 0) {
                const int i = M.wr_row;
                xl = gl_add(xl, gamma2); yl = gl_add(yl, gamma2); xr = gl_add(xr, gamma2); yr = gl_add(yr, gamma2);  // t + 1
                if constexpr (!MIRROR) {
                    u64 q0, q1;
                    gp_first_pair<false>(A, xl, yl, xr, yr, J.pw[i], J.pwr[i], r, true, out + (size_t)(2 * i) * half + jo,
                                         out + (size_t)(2 * i + 1) * half + jo, J.next_level ? J.next_level + (size_t)i * hN + 2 * j : nullptr, q0, q1);
                } else if (J.next_level)   // only the tree needs the write row: its level-1 entries
                    *reinterpret_cast<ulonglong2*>(J.next_level + (size_t)i * hN + 2 * j) = make_ulonglong2(gl_mul(xl, xr), gl_mul(yl, yr));
            }
            }
        }
        E2 s0, s2, s3;
        gp_first_acc_reduce(A, s0, s2, s3);
        if constexpr (MIRROR) {
            s0 = e2_add(s0, e2_add(e2_mul(J.mk1, Sx), J.mk2));
            s2 = e2_add(s2, e2_add(e2_mul(J.mk1, Sy), J.mk2));
            store_e2_nt(out + (size_t)(J.ntab - 1) * half + jo, e2_add(Sx, e2_mul(r, e2_sub(Sy, Sx))));
        }
        // (s0, s2, s3) = (sum P0, sum P1, sum Pinf): q(2) = 2 P1 - P0 + 2 Pinf, q(3) = 3 P1 - 2 P0 + 6 Pinf, times p(0), p(2), p(3)
        const E2 P1x2 = e2_dbl(s2), Pix2 = e2_dbl(s3);
        const E2 q2 = e2_add(e2_sub(P1x2, s0), Pix2);
        const E2 q3 = e2_add(e2_sub(e2_add(P1x2, s2), e2_dbl(s0)), e2_add(e2_dbl(Pix2), Pix2));
        acc[0] = e2_add(acc[0], e2_mul_f(s0, p0));
        acc[1] = e2_add(acc[1], e2_mul_f(q2, p2));
        acc[2] = e2_add(acc[2], e2_mul_f(q3, p3));
    }
    E2* sm = dyn_lds;
    block_sum_multi<3>(acc, sm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int t = 0; t < 3; t++) {
            if (nblocks == 1) res[J.sums_slot + t] = acc[t];
            else part_store(partials + (size_t)bx * 3 + t, acc[t]);
        }
    }
    if (nblocks > 1) finish_partials(partials, 3, tickets_of(partials), res + J.sums_slot, sm, nblocks);
}
// ---- the same first round, slot form with recomputed E (the form the large parameter sets run), restructured -----------------------
// k_gp_first_hash<true, true, true> asks memory six times in series per (tile, slot) - the slot's representative, that memory's
// descriptor, the chunk's dim / ts values, the class weights, the two emit masks (each wait behind a store draining the stores too) -
// and its 512 workgroups take tiles 512 apart. Here a workgroup takes CONSECUTIVE tiles: they lie in one segment pair (a segment
// holds 2^(seg_shift - 9) >= 1 tiles, 32 at n = 32768), so everything that depends on (slot, segment pair) only - representative,
// memory descriptor, weights, masks, the lookups' memory sets - is staged in LDS once per workgroup (again when the segment pair
// changes), and the dim / ts values of the NEXT chunk (of this tile or the next) are requested when the current chunk's are taken.
struct HashSlotD {
    E2 gm, gr;            // class weight, times r_0
    u64 erd, ewr;         // emit masks (read row, write row)
    int chunk, pad0;
    int rd_row, mem;
    u32 cutoff;
    int valid;            // 0: no such class in this segment pair (zero tables)
    int pad[2];
};
static_assert(sizeof(HashSlotD) == 80, "HashSlotD layout");
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gp_first_hash_slot(const StJob* __restrict__ job, const StItem* __restrict__ item, const E2* __restrict__ chal,
                                                           E2* __restrict__ partials, E2* __restrict__ res) {
    const StJob& J = *job;
    const StItem& I = *item;
    const GpHashSrc& H = *J.hash_src;
    const size_t half = (size_t)1 << I.h_log2;
    const size_t hN = half << 1;
    const size_t ntiles = half >> 8;
    const int nblocks = I.nblk, bx = (int)blockIdx.x, tid = threadIdx.x;
    E2* __restrict__ out = I.out;
    const E2 r = chal[J.r_off];
    const bool p0_only = J.p0_only != 0;
    const u64 gamma = H.gamma, gamma2 = H.gamma2, tau = H.tau;
    const int nslots = H.nslots, npairs = H.npairs, seg_shift = H.seg_shift;
    u64* __restrict__ next_level = J.next_level;
    const int ntab = J.ntab;
    const E2 mk1 = J.mk1, mk2 = J.mk2;
    const size_t per = (ntiles + nblocks - 1) / nblocks;
    const size_t t_lo = (size_t)bx * per, t_hi = t_lo + per < ntiles ? t_lo + per : ntiles;
    __shared__ HashSlotD D[64];
    __shared__ const u64* s_dim[4];
    __shared__ const u64* s_ts[4];
    __shared__ int s_ng, s_glo[65], s_gchunk[64];   // runs of slots whose classes share a chunk (the memories are listed chunk by chunk): [s_glo[k], s_glo[k+1])
    __shared__ u32 s_uses[2];
    int staged_sp = -1;
    auto stage = [&](int sp, size_t tile) {   // (all threads; uniform arguments)
        __syncthreads();
        if (tid < nslots) {
            HashSlotD d;
            const size_t at = (size_t)tid * npairs + sp;
            const int mi = H.rep[at];
            d.gm = gload_e2(H.slotw + 2 * at); d.gr = gload_e2(H.slotw + 2 * at + 1);
            d.erd = next_level ? gload_u64(H.emit_rd + at) : 0; d.ewr = next_level ? gload_u64(H.emit_wr + at) : 0;
            d.valid = mi != 255;
            d.chunk = -1; d.pad0 = 0; d.rd_row = 0; d.mem = 0; d.cutoff = 0; d.pad[0] = d.pad[1] = 0;
            if (d.valid) { const GpHashMem M = H.mems[mi]; d.chunk = M.chunk; d.rd_row = M.rd_row; d.mem = M.mem; d.cutoff = M.cutoff; }
            D[tid] = d;
        }
        if (tid < 4) { s_dim[tid] = H.dim[tid]; s_ts[tid] = H.ts[tid]; }
        if (tid == 64) {   // the memories the lookups of this tile's two row segments use (alpha <= 32)
            const size_t p = tile << 9;
            s_uses[0] = p < H.rows ? (u32)H.lookup_uses[H.seg_lookup[p >> seg_shift]] : 0u;
            s_uses[1] = hN + p < H.rows ? (u32)H.lookup_uses[H.seg_lookup[(hN + p) >> seg_shift]] : 0u;
        }
        __syncthreads();
        if (tid == 0) {
            int ng = 0, cur = -1;
            for (int m = 0; m < nslots; m++) {
                if (!D[m].valid || D[m].chunk == cur) continue;
                cur = D[m].chunk;
                s_glo[ng] = ng == 0 ? 0 : m; s_gchunk[ng] = cur; ng++;
            }
            if (ng == 0) { s_glo[0] = 0; s_gchunk[0] = 0; ng = 1; }   // (no class at all here: zero tables; the values asked for are ignored)
            s_glo[ng] = nslots;
            s_ng = ng;
        }
        __syncthreads();
        staged_sp = sp;
    };
    struct Raw { u64x2_t dl, dh, tl, th; };
    auto request = [&](size_t tile, int chunk, Raw& w) {
        const size_t j = (tile << 8) + tid;
        const u64* __restrict__ dim = s_dim[chunk];
        const u64* __restrict__ ts = s_ts[chunk];
        w.dl = load16<true>(dim + 2 * j); w.dh = load16<true>(dim + hN + 2 * j);
        w.tl = load16<true>(ts + 2 * j); w.th = load16<true>(ts + hN + 2 * j);
    };
    E2 acc[3] = {e2_zero(), e2_zero(), e2_zero()};
    Raw nxt;
    if (t_lo < t_hi) {
        stage((int)((t_lo << 9) >> seg_shift), t_lo);
        request(t_lo, s_gchunk[0], nxt);
    }
    for (size_t tile = t_lo; tile < t_hi; tile++) {
        const int sp = (int)((tile << 9) >> seg_shift);
        if (sp != staged_sp) {   // (rare: a workgroup's tiles straddle two segment pairs) - the values asked for may be another chunk's
            stage(sp, tile);
            request(tile, s_gchunk[0], nxt);
        }
        const size_t j = (tile << 8) + tid;
        const size_t jo = dpos(j, half);
        const u32 uses_lo = 2 * j < H.rows ? s_uses[0] : 0u, uses_hi = hN + 2 * j < H.rows ? s_uses[1] : 0u;   // (row segment sp and sp + npairs)
        GpFirstAcc A = gp_first_acc_zero();
        E2 Sx = e2_zero(), Sy = e2_zero();
        u64 p0 = 0, p2 = 0, p3 = 0;
        const int ng = s_ng;
        for (int k = 0; k < ng; k++) {
        // take the values asked for, ask for the next run's: this tile's, else the next tile's first (the last run of the workgroup asks
        // for its own again: every run issues the same loads). The values are consumed BEFORE the request and the slots of the run
        // do not touch them: the loop over the slots carries no copy of values in flight.
        const u64 c0 = gl_sub(gl_add(nxt.dl.x, gl_mul_small(gamma2, (u32)nxt.tl.x)), tau), c1 = gl_sub(gl_add(nxt.dl.y, gl_mul_small(gamma2, (u32)nxt.tl.y)), tau);
        const u64 c2 = gl_sub(gl_add(nxt.dh.x, gl_mul_small(gamma2, (u32)nxt.th.x)), tau), c3 = gl_sub(gl_add(nxt.dh.y, gl_mul_small(gamma2, (u32)nxt.th.y)), tau);
        const u32 a01 = (u32)nxt.dl.x | ((u32)nxt.dl.y << 16), a23 = (u32)nxt.dh.x | ((u32)nxt.dh.y << 16);   // the chunk's 16-bit limbs at 2j, 2j+1 | N/2+2j, N/2+2j+1
        {
            const bool more_k = k + 1 < ng, more_t = tile + 1 < t_hi;
            request(more_k || !more_t ? tile : tile + 1, s_gchunk[more_k ? k + 1 : (more_t ? 0 : k)], nxt);
        }
        const int m_hi = s_glo[k + 1];
        for (int m = s_glo[k]; m < m_hi; m++) {
            const int valid = __builtin_amdgcn_readfirstlane(D[m].valid);
            if (!valid) {
                store_e2_nt(out + (size_t)(2 * m) * half + jo, e2_zero());
                store_e2_nt(out + (size_t)(2 * m + 1) * half + jo, e2_zero());
                continue;
            }
            const u32 cut = (u32)__builtin_amdgcn_readfirstlane((int)D[m].cutoff);
            const int mem = __builtin_amdgcn_readfirstlane(D[m].mem), rd_row = __builtin_amdgcn_readfirstlane(D[m].rd_row);
            const bool ul = (uses_lo >> mem) & 1, uh = (uses_hi >> mem) & 1;
            u32 e0 = a01 & 0xFFFF, e1 = a01 >> 16, e2v = a23 & 0xFFFF, e3 = a23 >> 16;
            e0 = (ul && e0 < cut) ? e0 : 0; e1 = (ul && e1 < cut) ? e1 : 0;
            e2v = (uh && e2v < cut) ? e2v : 0; e3 = (uh && e3 < cut) ? e3 : 0;
            u64 xl = gl_add(c0, gl_mul_small(gamma, e0)), yl = gl_add(c1, gl_mul_small(gamma, e1));
            u64 xr = gl_add(c2, gl_mul_small(gamma, e2v)), yr = gl_add(c3, gl_mul_small(gamma, e3));
            if (rd_row == 0) { const u64 d = gl_sub(yl, xl); p0 = xl; p2 = gl_add(yl, d); p3 = gl_add(p2, d); }
            const E2 gm = D[m].gm, gr = D[m].gr;
            u64 q0, q1;
            const bool summed = !(p0_only && m == 0);   // (sharded: class 0 = pair 0, held for p_0 only)
            gp_first_pair<true>(A, xl, yl, xr, yr, gm, gr, r, summed, out + (size_t)(2 * m) * half + jo, out + (size_t)(2 * m + 1) * half + jo, nullptr, q0, q1);
            if (summed) {
                const u64 hx = gl_add(xl, xr), hy = gl_add(yl, yr);
                Sx = e2_add(Sx, e2_mul_f(gm, hx));
                Sy = e2_add(Sy, e2_mul_f(gm, hy));
            }
            if (next_level) {
                const u64 erd = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(D[m].erd >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)D[m].erd);
                const u64 ewr = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(D[m].ewr >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)D[m].ewr);
                emit_rows(next_level, hN, 2 * j, erd, make_ulonglong2(q0, q1));
                xl = gl_add(xl, gamma2); yl = gl_add(yl, gamma2); xr = gl_add(xr, gamma2); yr = gl_add(yr, gamma2);
                emit_rows(next_level, hN, 2 * j, ewr, make_ulonglong2(gl_mul(xl, xr), gl_mul(yl, yr)));
            }
        }
        }
        E2 s0, s2, s3;
        gp_first_acc_reduce(A, s0, s2, s3);
        s0 = e2_add(s0, e2_add(e2_mul(mk1, Sx), mk2));
        s2 = e2_add(s2, e2_add(e2_mul(mk1, Sy), mk2));
        store_e2_nt(out + (size_t)(ntab - 1) * half + jo, e2_add(Sx, e2_mul(r, e2_sub(Sy, Sx))));
        const E2 P1x2 = e2_dbl(s2), Pix2 = e2_dbl(s3);
        const E2 q2 = e2_add(e2_sub(P1x2, s0), Pix2);
        const E2 q3 = e2_add(e2_sub(e2_add(P1x2, s2), e2_dbl(s0)), e2_add(e2_dbl(Pix2), Pix2));
        acc[0] = e2_add(acc[0], e2_mul_f(s0, p0));
        acc[1] = e2_add(acc[1], e2_mul_f(q2, p2));
        acc[2] = e2_add(acc[2], e2_mul_f(q3, p3));
    }
    E2* sm = dyn_lds;
    block_sum_multi<3>(acc, sm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int t = 0; t < 3; t++) {
            if (nblocks == 1) res[J.sums_slot + t] = acc[t];
            else part_store(partials + (size_t)bx * 3 + t, acc[t]);
        }
    }
    if (nblocks > 1) finish_partials(partials, 3, tickets_of(partials), res + J.sums_slot, sm, nblocks);
}
static inline size_t sc_lds_bytes(int nv, int bd);
void st_first_hash(hipStream_t st, const StJob* job, const StItem* item, int grid, bool mirror, bool recomp, const E2* chal, E2* partials, E2* res, bool slot) {
    const size_t lds = sc_lds_bytes(0, 256);
    if (slot) {
        if (!mirror) throw std::runtime_error("st_first_hash: the slot form is a mirrored job");
        if (recomp) k_gp_first_hash_slot<<<grid, 256, lds, st>>>(job, item, chal, partials, res);
        else k_gp_first_hash<true, false, true><<<grid, 256, lds, st>>>(job, item, chal, partials, res);
    } else if (recomp) {
        if (mirror) k_gp_first_hash<true, true><<<grid, 256, lds, st>>>(job, item, chal, partials, res);
        else k_gp_first_hash<false, true><<<grid, 256, lds, st>>>(job, item, chal, partials, res);
    } else {
        if (mirror) k_gp_first_hash<true, false><<<grid, 256, lds, st>>>(job, item, chal, partials, res);
        else k_gp_first_hash<false, false><<<grid, 256, lds, st>>>(job, item, chal, partials, res);
    }
}

// ---- two grand-product rounds in one pass ------------------------------------------------------------------
// Thread j (one per pair index of round t, jb = 8) does round t as in sc_round_body; the folded values T'[j] stay in
// registers. Round t+1 pairs (T'[2j'], T'[2j'+1]) live in the two lanes 2j', 2j'+1 of a wave: they swap their
// folded values (DPP quad_perm) and split the work: the even lane accumulates P0 = x'l x'r (its own values) and
// folds the left table, the odd lane accumulates P1 = y'l y'r (its own values) and folds the right table; Pinf =
// d'l d'r is split by Ext2 coordinate (the sign of d' cancels in the product). Round-t sums use two column
// accumulators per Ext2 sum (7 a1 pre-multiplied), so the register count stays that of the single-round kernel.
// HBM traffic per (pair, j): 64 B read + 16 B written instead of (64 + 32) + (32 + 16).
struct W2 { WAcc c0, c1; };
__device__ __forceinline__ W2 w2_zero() { W2 w; w.c0 = wacc_zero(); w.c1 = wacc_zero(); return w; }
__device__ __forceinline__ void w2_mac(W2& w, E2 a, E2 b) {
    const u64 a7 = gl_mul7_lazy(a.c1);  // any 64-bit residue will do as a multiplicand
    wmac_pair(w.c0, a.c0, b.c0, a7, b.c1);
    wmac_pair(w.c1, a.c0, b.c1, a.c1, b.c0);
}
__device__ __forceinline__ E2 w2_reduce(const W2& w) { return e2(wreduce(w.c0), wreduce(w.c1)); }
__device__ __forceinline__ u64 swap_lane_u64(u64 v) {
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    lo = (u32)__builtin_amdgcn_mov_dpp((int)lo, 0xB1, 0xF, 0xF, true);  // quad_perm:[1,0,3,2]
    hi = (u32)__builtin_amdgcn_mov_dpp((int)hi, 0xB1, 0xF, 0xF, true);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ E2 swap_lane(E2 v) { return e2(swap_lane_u64(v.c0), swap_lane_u64(v.c1)); }
__device__ __forceinline__ E2 gp_combine(E2 P0, E2 P1, E2 Pi, E2 p0, E2 p2, E2 p3, E2& a0, E2& a1, E2& a2) {
    // q(2) = 2 P1 - P0 + 2 Pinf, q(3) = 3 P1 - 2 P0 + 6 Pinf
    E2 P1x2 = e2_dbl(P1), Pix2 = e2_dbl(Pi);
    E2 q2 = e2_add(e2_sub(P1x2, P0), Pix2);
    E2 q3 = e2_add(e2_sub(e2_add(P1x2, P1), e2_dbl(P0)), e2_add(e2_dbl(Pix2), Pix2));
    a0 = e2_add(a0, e2_mul(P0, p0));
    a1 = e2_add(a1, e2_mul(q2, p2));
    a2 = e2_add(a2, e2_mul(q3, p3));
    return q2;
}
__global__ __launch_bounds__(256) void k_st_step2(const StJob* __restrict__ jobs, const StItem* __restrict__ items, int nitems,
                                                  const E2* __restrict__ chal, E2* __restrict__ partials, E2* __restrict__ res) {
    const int y = find_item(items, nitems, blockIdx.x);
    const StItem& I = items[y];
    const StJob& J = jobs[I.job];
    const int h_log2 = I.h_log2;
    const int rd = J.nvars - 1 - h_log2;
    const size_t half = (size_t)1 << h_log2, half2 = half >> 1;
    const size_t ntiles = half >> 8;
    const int nblocks = I.nblk, bx = (int)blockIdx.x - I.blk0;
    const E2* __restrict__ in = reinterpret_cast<const E2*>(I.in);
    const size_t in_stride = I.in_stride;
    E2* __restrict__ out = I.out;
    const int nb = J.ntab >> 1;
    const bool p0_only = J.p0_only != 0;
    const int tid = threadIdx.x;
    const bool odd = tid & 1;
    const FoldR fa = fold_r(chal[J.r_off + rd]);
    const E2 rb = chal[J.r_off + rd + 1];
    // the odd lane folds y' + (1 - r)(x' - y') = x' + r (y' - x') from its own value y'
    const FoldR fb = fold_r(odd ? e2_sub(e2_one(), rb) : rb);
    E2 acc[6];
#pragma unroll
    for (int t = 0; t < 6; t++) acc[t] = e2_zero();
    GpItemE2 cur = gp_item_any();
    if ((size_t)bx < ntiles) {
        load_xy<E2, false>(in, ((size_t)bx << 8) + tid, half, cur.xl, cur.yl);
        load_xy<E2, false>(in + in_stride, ((size_t)bx << 8) + tid, half, cur.xr, cur.yr);
    }
    for (size_t tile = bx; tile < ntiles; tile += nblocks) {
        const size_t j = (tile << 8) + tid;
        const size_t j2 = j >> 1;
        const size_t jo2 = dpos(j2, half2);
        // round t: three accumulators per Ext2 sum (a0 b0 | a1 b1 | cross terms), the factor 7 of X^2 = 7 applied once per j at
        // the reduction instead of once per operand; round t+1 (vm, vi): two accumulators with 7 a1 pre-multiplied (registers)
        WE2 w0 = we2_zero(), w1 = we2_zero(), wi = we2_zero();
        W2 vm = w2_zero();
        WAcc vi = wacc_zero();
        E2 p0 = e2_zero(), p2 = e2_zero(), p3 = e2_zero(), q0 = e2_zero(), q2 = e2_zero(), q3 = e2_zero();
        if (J.mirror) {
            // The linear table S of a mirrored job (StJob::mirror), both rounds, BEFORE the pair loop: K1 S + K2 joins P0 and P1 of
            // round t and, on the folded S, of round t+1 (even lane: P0', odd lane: P1') - as the starting values of the column
            // accumulators (their low word is a plain residue), so nothing stays live across the loop.
            E2 x, y;
            load_xy<E2, false>(in + (size_t)(2 * nb) * in_stride, j, half, x, y);
            const E2 ms = e2_fold_wide(x, e2_sub(y, x), fa);
            const E2 es = e2_sub(swap_lane(ms), ms);
            const E2 c0 = e2_add(e2_mul(J.mk1, x), J.mk2), c1 = e2_add(e2_mul(J.mk1, y), J.mk2), cm = e2_add(e2_mul(J.mk1, ms), J.mk2);
            w0.A.L = c0.c0; w0.C.L = c0.c1;
            w1.A.L = c1.c0; w1.C.L = c1.c1;
            vm.c0.L = cm.c0; vm.c1.L = cm.c1;
            const E2 fs = e2_fold_wide(ms, es, fb);
            if (!odd) store_e2(out + (size_t)(2 * nb) * half2 + jo2, fs);
        }
        // software pipeline over the workgroup's whole (tile, pair) stream (two waves per SIMD cannot hide an HBM round trip behind
        // ~800 instructions otherwise): the next item's four loads go out before this one is computed on, into registers of their
        // own, and every iteration issues them (the last item asks for the thread's first again) - see gp_first_round_body
        for (int i = 0; i < nb; i++) {
            const bool more_i = i + 1 < nb, more_t = tile + nblocks < ntiles;
            GpItemE2 nxt;
            {
                const size_t jn = ((more_i || !more_t ? tile : tile + nblocks) << 8) + tid;
                const int in_ = more_i ? i + 1 : 0;
                load_xy<E2, false>(in + (size_t)(2 * in_) * in_stride, jn, half, nxt.xl, nxt.yl);
                load_xy<E2, false>(in + (size_t)(2 * in_ + 1) * in_stride, jn, half, nxt.xr, nxt.yr);
            }
            const E2 xl = cur.xl, yl = cur.yl, xr = cur.xr, yr = cur.yr;
            const E2 dl = e2_sub(yl, xl), dr = e2_sub(yr, xr);
            const bool summed = !(p0_only && i == 0);
            if (i == 0) { p0 = xl; p2 = e2_add(yl, dl); p3 = e2_add(p2, dl); }
            if (summed) { we2_mac(w0, xl, xr); we2_mac(w1, yl, yr); we2_mac(wi, dl, dr); }
            const E2 ml = e2_fold_wide(xl, dl, fa), mr = e2_fold_wide(xr, dr, fa);  // T'[j] of the left / right table
            const E2 ol = swap_lane(ml), orr = swap_lane(mr);                        // the neighbour's
            const E2 el = e2_sub(ol, ml), er = e2_sub(orr, mr);                      // +-(y' - x')
            if (i == 0) {  // p' of round t+1 at 0, 2, 3 from x' = even lane's, y' = odd lane's left value
                const E2 x = odd ? ol : ml, y = odd ? ml : ol;
                const E2 d = e2_sub(y, x);
                q0 = x; q2 = e2_add(y, d); q3 = e2_add(q2, d);
            }
            if (summed) {
                w2_mac(vm, ml, mr);  // even lane: P0 term, odd lane: P1 term
                // Pinf = el * er: the even lane takes coordinate 0 (el0 er0 + 7 el1 er1), the odd lane coordinate 1
                const u64 b = odd ? er.c1 : er.c0, d = odd ? er.c0 : er.c1;
                const u64 c = odd ? el.c1 : gl_mul7_lazy(el.c1);
                wmac_pair(vi, el.c0, b, c, d);
            }
            const E2 fx = odd ? mr : ml, fd = odd ? er : el;
            store_e2(out + (size_t)(2 * i + (odd ? 1 : 0)) * half2 + jo2, e2_fold_wide(fx, fd, fb));
            cur = nxt;
        }
        gp_combine(we2_reduce(w0), we2_reduce(w1), we2_reduce(wi), p0, p2, p3, acc[0], acc[1], acc[2]);
        const E2 mine = w2_reduce(vm), other = swap_lane(mine);
        const u64 ip = wreduce(vi), iq = swap_lane_u64(ip);
        if (!odd) gp_combine(mine, other, e2(ip, iq), q0, q2, q3, acc[3], acc[4], acc[5]);
    }
    E2* sm = dyn_lds;
    E2* part = partials + (size_t)y * SC_MAX_BLOCKS * 6;
    block_sum_multi<6>(acc, sm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int t = 0; t < 6; t++) {
            if (nblocks == 1) res[J.sums_slot + (size_t)rd * 3 + t] = acc[t];
            else part_store(part + (size_t)bx * 6 + t, acc[t]);
        }
    }
    if (nblocks > 1) finish_partials(part, 6, tickets_of(partials) + y * 32, res + J.sums_slot + (size_t)rd * 3, sm, nblocks);
}
// Collation shape, two rounds per pass (folded Ext2 tables, weights already in the tables): the round sums are plain
// sums over the tables; the lane pair (2j', 2j'+1) shares round t+1 as in k_st_step2 - each lane adds up its own folded
// values (even lane: x' = T'[2j'], odd lane: y' = T'[2j'+1]) and the folds of round t+1 are split by table parity
// (tables are taken two at a time: the even lane folds table i, the odd lane table i+1).
__global__ __launch_bounds__(256) void k_col_step2(const StJob* __restrict__ jobs, const StItem* __restrict__ items, int nitems,
                                                   const E2* __restrict__ chal, E2* __restrict__ partials, E2* __restrict__ res) {
    const int y = find_item(items, nitems, blockIdx.x);
    const StItem& I = items[y];
    const StJob& J = jobs[I.job];
    const int h_log2 = I.h_log2;
    const int rd = J.nvars - 1 - h_log2;
    const size_t half = (size_t)1 << h_log2, half2 = half >> 1;
    const size_t ntiles = half >> 8;
    const int nblocks = I.nblk, bx = (int)blockIdx.x - I.blk0;
    const E2* __restrict__ in = reinterpret_cast<const E2*>(I.in);
    const size_t in_stride = I.in_stride;
    E2* __restrict__ out = I.out;
    const int ntab = J.ntab;
    const bool p0_only = J.p0_only != 0;
    const int tid = threadIdx.x;
    const bool odd = tid & 1;
    const FoldR fa = fold_r(chal[J.r_off + rd]);
    const E2 rb = chal[J.r_off + rd + 1];
    const FoldR fb = fold_r(odd ? e2_sub(e2_one(), rb) : rb);  // odd lane: y' + (1 - r)(x' - y')
    E2 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = e2_zero();
    for (size_t tile = bx; tile < ntiles; tile += nblocks) {
        const size_t j = (tile << 8) + tid;
        const size_t jo2 = dpos(j >> 1, half2);
        E2 s0 = e2_zero(), s2 = e2_zero(), own = e2_zero();
        E2 p0 = e2_zero(), p2 = e2_zero(), q0 = e2_zero(), q2 = e2_zero();
        for (int i = 0; i < ntab; i += 2) {
            const bool two = i + 1 < ntab;
            E2 xa, ya, xb = e2_zero(), yb = e2_zero();
            load_xy<E2, false>(in + (size_t)i * in_stride, j, half, xa, ya);
            if (two) load_xy<E2, false>(in + (size_t)(i + 1) * in_stride, j, half, xb, yb);
            const E2 da = e2_sub(ya, xa), db = e2_sub(yb, xb);
            const E2 va = e2_add(ya, da), vb = e2_add(yb, db);
            if (i == 0) { p0 = xa; p2 = va; }
            if (p0_only && i == 0) { s0 = e2_add(s0, xb); s2 = e2_add(s2, vb); }  // table 0 only supplies p_0 (another rank sums it)
            else { s0 = e2_add(s0, e2_add(xa, xb)); s2 = e2_add(s2, e2_add(va, vb)); }
            const E2 ma = e2_fold_wide(xa, da, fa), mb = two ? e2_fold_wide(xb, db, fa) : e2_zero();  // T'[j] of tables i, i+1
            const E2 oa = swap_lane(ma), ob = swap_lane(mb);
            if (i == 0) {
                const E2 x = odd ? oa : ma, yv = odd ? ma : oa;
                q0 = x; q2 = e2_add(yv, e2_sub(yv, x));
            }
            own = e2_add(own, (p0_only && i == 0) ? mb : e2_add(ma, mb));
            const E2 fx = odd ? mb : ma, fd = odd ? e2_sub(ob, mb) : e2_sub(oa, ma);
            if (!odd || two) store_e2(out + (size_t)(i + (odd ? 1 : 0)) * half2 + jo2, e2_fold_wide(fx, fd, fb));
        }
        acc[0] = e2_add(acc[0], e2_mul(p0, s0));
        acc[1] = e2_add(acc[1], e2_mul(p2, s2));
        const E2 other = swap_lane(own);
        if (!odd) {  // s0' = sum x', s2' = sum (2 y' - x')
            acc[2] = e2_add(acc[2], e2_mul(q0, own));
            acc[3] = e2_add(acc[3], e2_mul(q2, e2_sub(e2_dbl(other), own)));
        }
    }
    E2* sm = dyn_lds;
    E2* part = partials + (size_t)y * SC_MAX_BLOCKS * 4;
    block_sum_multi<4>(acc, sm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (nblocks == 1) res[J.sums_slot + (size_t)rd * 2 + t] = acc[t];
            else part_store(part + (size_t)bx * 4 + t, acc[t]);
        }
    }
    if (nblocks > 1) finish_partials(part, 4, tickets_of(partials) + y * 32, res + J.sums_slot + (size_t)rd * 2, sm, nblocks);
}

static inline size_t sc_lds_bytes(int nv, int bd) { return (SM_SLOTS + (size_t)nv * bd) * sizeof(E2); }

// ---- LDS-resident tail: ALL remaining rounds of a job in one workgroup ------------------------------------------------------
// The item's first round (half = 2^h0 pair indices) reads the job's tables from HBM and folds them into LDS; every later round
// runs LDS -> LDS with the thread mapping of the big rounds (sc_round_body: 2^jb threads along j, the rest along the tables), the
// last one writes the ntab scalars. How many rounds fit depends on the job's table count (st_tail_h: 12 for the two collation
// tables, 7 for the mirrored top layer, 6 for a full grand-product layer, 9-10 for the few tables of a sharded rank), so a small
// job spends its whole launch-bound second half in ONE launch. Dynamic LDS: [SM_SLOTS][NV * ST_TAIL_THREADS reduction slots][ntab 2^h0][ntab 2^(h0-1)].
constexpr size_t ST_TAIL_LDS_BYTES = 120 * 1024;   // for the two table regions
constexpr int ST_TAIL_THREADS = 256;               // (512: 92.8 against 84 us for the grand-product tail - the per-round chain is the same and the sums cross more waves)
int st_tail_h(int ntab, int nvars) {
    int h = 0;
    while (h + 1 <= nvars - 1 && (size_t)ntab * (((size_t)1 << (h + 1)) + ((size_t)1 << h)) * sizeof(E2) <= ST_TAIL_LDS_BYTES) h++;
    return h;
}
template <int KIND>
__global__ __launch_bounds__(ST_TAIL_THREADS) void k_st_tail(const StJob* __restrict__ jobs, const StItem* __restrict__ items, const E2* __restrict__ chal,
                                                 E2* __restrict__ res) {
    constexpr int NV = KIND == SC_GRANDPROD ? 3 : 2;
    // The job and item descriptors are copied to LDS first: every round reads a dozen of their fields behind a barrier. (A round of
    // this kernel takes 6.2-6.8 us whatever its size - in-kernel clock reads; that is the dependent chain of extension-field
    // multiplications per round (fold factors, pair products, the p_0 scaling, the wave reductions), not memory: the same figure
    // with the descriptors in global memory and with the round sums stored to the host-mapped result buffer every round.)
    __shared__ StJob Jl;
    __shared__ StItem Il;
    {
        const StItem* gi = items + blockIdx.x;
        const unsigned* src = reinterpret_cast<const unsigned*>(jobs + gi->job);
        unsigned* dst = reinterpret_cast<unsigned*>(&Jl);
        for (unsigned k = threadIdx.x; k < sizeof(StJob) / 4; k += blockDim.x) dst[k] = src[k];
        if (threadIdx.x < sizeof(StItem) / 4) reinterpret_cast<unsigned*>(&Il)[threadIdx.x] = reinterpret_cast<const unsigned*>(gi)[threadIdx.x];
    }
    __syncthreads();
    const StItem& I = Il;
    if (threadIdx.x == 0 && Il.ntab > 0) Jl.ntab = Il.ntab;   // (a slot-form job regrouped into its per-memory tables ahead of this launch)
    __syncthreads();
    const StJob& J = Jl;
    E2* sm = dyn_lds;
    E2* red = dyn_lds + SM_SLOTS;
    E2* tab[2];
    const int h0 = J.nvars - 1 - I.rd;
    tab[0] = red + NV * ST_TAIL_THREADS;
    tab[1] = tab[0] + ((size_t)J.ntab << h0);
    const bool p0_only = J.p0_only != 0;
    const StJob* mirror = (KIND == SC_GRANDPROD && J.mirror) ? &J : nullptr;
    const void* in = I.in;
    size_t in_stride = I.in_stride;
    // The round sums go to the result buffer - host memory across PCIe - ONCE, at the end, not a store per round ahead of the
    // round's barrier. The challenges are fetched once as well.
    __shared__ E2 keep[3 * 32], rch[32];
    if ((int)threadIdx.x < J.nvars - I.rd) rch[threadIdx.x] = chal[J.r_off + I.rd + threadIdx.x];
    __syncthreads();
#pragma unroll 1
    for (int rd = I.rd; rd < J.nvars; rd++) {
        const int h_log2 = J.nvars - 1 - rd;
        const size_t half = (size_t)1 << h_log2;
        const int jb_log2 = h_log2 < 8 ? h_log2 : 8;
        const bool last = rd == J.nvars - 1;
        E2* out = last ? J.final_out : tab[(rd - I.rd) & 1];
        E2 acc[NV];
#pragma unroll
        for (int t = 0; t < NV; t++) acc[t] = e2_zero();
        const E2 r = rch[rd - I.rd];
        if (rd == 0 && J.base) sc_round_body<KIND, u64, true, false, false>(reinterpret_cast<const u64*>(in), in_stride, out, half, J.ntab, half, r, J.pw, J.pwr, jb_log2, red, acc, 0, 1, p0_only);
        else if (rd == 0) sc_round_body<KIND, E2, true, false, false>(reinterpret_cast<const E2*>(in), in_stride, out, half, J.ntab, half, r, J.pw, J.pwr, jb_log2, red, acc, 0, 1, p0_only);
        else sc_round_body<KIND, E2, false, false, false>(reinterpret_cast<const E2*>(in), in_stride, out, half, J.ntab, half, r, J.pw, J.pwr, jb_log2, red, acc, 0, 1, p0_only, nullptr, mirror);
        block_sum_multi<NV>(acc, sm);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int t = 0; t < NV; t++) keep[(rd - I.rd) * NV + t] = acc[t];
        }
        __syncthreads();   // the folded tables of this round are complete in LDS
        in = out; in_stride = half;
    }
    if ((int)threadIdx.x < (J.nvars - I.rd) * NV) res[J.sums_slot + (size_t)I.rd * NV + threadIdx.x] = keep[threadIdx.x];
}
__global__ void k_gp_slot_regroup(const E2* __restrict__ in, E2* __restrict__ out, const uint8_t* __restrict__ slot_of, const E2* __restrict__ ratio,
                                  int nrows, int nslots, int npairs, int len_log2, int sh, int has_s) {
    const size_t len = (size_t)1 << len_log2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ((size_t)nrows + has_s) << len_log2) return;
    const int b = (int)(idx >> len_log2);
    const size_t p = idx & (len - 1);   // storage position inside a table; the tables are de-interleaved
    if (b == nrows) { out[((size_t)(2 * nrows) << len_log2) + p] = in[((size_t)(2 * nslots) << len_log2) + p]; return; }   // S
    const size_t t = p < len / 2 ? 2 * p : 2 * (p - len / 2) + 1;   // the logical entry at that position
    const int sp = (int)(t >> sh);
    const int v = slot_of[(size_t)b * npairs + sp];
    const E2 l = in[((size_t)(2 * v) << len_log2) + p], r = in[((size_t)(2 * v + 1) << len_log2) + p];
    out[((size_t)(2 * b) << len_log2) + p] = b == 0 ? l : e2_mul(ratio[(size_t)b * npairs + sp], l);
    out[((size_t)(2 * b + 1) << len_log2) + p] = r;
}
// the same for several jobs in one launch (grid.y = job): the four slot-form layers' regroups ahead of the tail were four launches on the
// critical chain of the prove
__global__ void k_gp_slot_regroup_jobs(const SlotRegroupJob* __restrict__ jobs) {
    const SlotRegroupJob& J = jobs[blockIdx.y];
    const size_t len = (size_t)1 << J.len_log2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ((size_t)J.nrows + J.has_s) << J.len_log2) return;
    const int b = (int)(idx >> J.len_log2);
    const size_t p = idx & (len - 1);   // storage position inside a table; the tables are de-interleaved
    if (b == J.nrows) { J.out[((size_t)(2 * J.nrows) << J.len_log2) + p] = J.in[((size_t)(2 * J.nslots) << J.len_log2) + p]; return; }   // S
    const size_t t = p < len / 2 ? 2 * p : 2 * (p - len / 2) + 1;   // the logical entry at that position
    const int sp = (int)(t >> J.sh);
    const int v = J.slot_of[(size_t)b * J.npairs + sp];
    const E2 l = J.in[((size_t)(2 * v) << J.len_log2) + p], r = J.in[((size_t)(2 * v + 1) << J.len_log2) + p];
    J.out[((size_t)(2 * b) << J.len_log2) + p] = b == 0 ? l : e2_mul(J.ratio[(size_t)b * J.npairs + sp], l);
    J.out[((size_t)(2 * b + 1) << J.len_log2) + p] = r;
}
void gp_slot_regroup_jobs(hipStream_t st, const SlotRegroupJob* jobs, int njobs, size_t max_entries) {
    if (njobs) k_gp_slot_regroup_jobs<<<dim3((unsigned)((max_entries + 255) / 256), (unsigned)njobs), 256, 0, st>>>(jobs);
}
SlotRegroupJob gp_slot_regroup_job(const E2* in, E2* out, const uint8_t* slot_of, const E2* ratio, int nrows, int nslots, int npairs, int len_log2, bool has_s) {
    int np_log2 = 0;
    while ((1 << np_log2) < npairs) np_log2++;
    SlotRegroupJob J;
    J.in = in; J.out = out; J.slot_of = slot_of; J.ratio = ratio; J.nrows = nrows; J.nslots = nslots; J.npairs = npairs; J.len_log2 = len_log2; J.sh = len_log2 - np_log2; J.has_s = has_s ? 1 : 0;
    return J;
}
void gp_slot_regroup(hipStream_t st, const E2* in, E2* out, const uint8_t* slot_of, const E2* ratio, int nrows, int nslots, int npairs, int len_log2, bool has_s) {
    int np_log2 = 0;
    while ((1 << np_log2) < npairs) np_log2++;
    const size_t n = ((size_t)nrows + (has_s ? 1 : 0)) << len_log2;
    k_gp_slot_regroup<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(in, out, slot_of, ratio, nrows, nslots, npairs, len_log2, len_log2 - np_log2, has_s ? 1 : 0);
}
void st_tail(hipStream_t st, int kind, const StJob* jobs, const StItem* items, int nitems, size_t table_bytes, const E2* chal, E2* res) {
    const size_t lds = (SM_SLOTS + 3 * ST_TAIL_THREADS) * sizeof(E2) + table_bytes;
    static const hipError_t a1 = hipFuncSetAttribute((const void*)k_st_tail<SC_GRANDPROD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((SM_SLOTS + 3 * ST_TAIL_THREADS) * sizeof(E2) + ST_TAIL_LDS_BYTES));
    static const hipError_t a2 = hipFuncSetAttribute((const void*)k_st_tail<SC_COLLATION>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((SM_SLOTS + 3 * ST_TAIL_THREADS) * sizeof(E2) + ST_TAIL_LDS_BYTES));
    (void)a1; (void)a2;
    if (kind == SC_GRANDPROD) k_st_tail<SC_GRANDPROD><<<nitems, ST_TAIL_THREADS, lds, st>>>(jobs, items, chal, res);
    else k_st_tail<SC_COLLATION><<<nitems, ST_TAIL_THREADS, lds, st>>>(jobs, items, chal, res);
}

int st_plan_blocks(StItem* items, int nitems, bool rounds2) {
    // total work of the launch decides how finely the small items are split along the tables (jb < 8)
    size_t total = 0;
    for (int q = 0; q < nitems; q++) total += (size_t)1 << items[q].h_log2;
    int blk = 0;
    for (int q = 0; q < nitems; q++) {
        StItem& I = items[q];
        int jb = 8;
        if (!rounds2) while (jb > 2 && (total << (8 - jb)) < st_min_threads()) jb--;
        if (jb > I.h_log2) jb = I.h_log2;
        const size_t ntiles = ((size_t)1 << I.h_log2) >> jb;
        I.jb_log2 = jb;
        I.blk0 = blk;
        I.nblk = (int)std::min<size_t>(ntiles, (size_t)st_max_blocks());
        blk += I.nblk;
    }
    return blk;
}
void st_step(hipStream_t st, int kind, bool base, const StJob* jobs, const StItem* items, int nitems, int grid, const E2* chal,
             E2* partials, E2* res, bool slot, int mode) {
    const int nv = kind == SC_GRANDPROD ? 3 : 2;
    const size_t lds = sc_lds_bytes(nv, 256);
    if (mode) {   // a split round (sc_round_body): later rounds of grand-product jobs only
        if (kind != SC_GRANDPROD || base || slot) throw std::runtime_error("st_step: split rounds are grand-product rounds on Ext2 tables");
        if (mode == 1) k_st_step<SC_GRANDPROD, E2, false, 1><<<grid, 256, lds, st>>>(jobs, items, nitems, chal, partials, res);
        else k_st_step<SC_GRANDPROD, E2, false, 2><<<grid, 256, lds, st>>>(jobs, items, nitems, chal, partials, res);
        return;
    }
    if (slot) {
        if (kind != SC_GRANDPROD || !base || nitems != 1) throw std::runtime_error("st_step: a slot-form first round is launched alone");
        k_st_step<SC_GRANDPROD, u64, true><<<grid, 256, lds, st>>>(jobs, items, nitems, chal, partials, res);
    } else if (kind == SC_GRANDPROD) {
        if (base) k_st_step<SC_GRANDPROD, u64><<<grid, 256, lds, st>>>(jobs, items, nitems, chal, partials, res);
        else k_st_step<SC_GRANDPROD, E2><<<grid, 256, lds, st>>>(jobs, items, nitems, chal, partials, res);
    } else {
        if (base) k_st_step<SC_COLLATION, u64><<<grid, 256, lds, st>>>(jobs, items, nitems, chal, partials, res);
        else k_st_step<SC_COLLATION, E2><<<grid, 256, lds, st>>>(jobs, items, nitems, chal, partials, res);
    }
}
void st_step2(hipStream_t st, int kind, const StJob* jobs, const StItem* items, int nitems, int grid, const E2* chal, E2* partials, E2* res) {
    if (kind == SC_GRANDPROD) k_st_step2<<<grid, 256, sc_lds_bytes(0, 256), st>>>(jobs, items, nitems, chal, partials, res);
    else k_col_step2<<<grid, 256, sc_lds_bytes(0, 256), st>>>(jobs, items, nitems, chal, partials, res);
}
// ---- PRODSUM: g = sum_i a_i * b_i (Libra / zkCNN reductions), batched over independent instances -----
// Round rd of job J: inputs are a[i]/b[i] (rd = 0; a in the base field) or the ping-pong buffers; table i of
// a buffer sits at buf + i * (current length).
__device__ __forceinline__ void ps_io(const PsJob& J, int rd, int rounds, int in_buf, int out_buf, int i, const void*& a, const E2*& b, E2*& oa, E2*& ob) {
    const size_t N = (size_t)1 << J.nvars;
    if (in_buf < 0) { a = J.a[i]; b = J.b[i]; }
    else { size_t len = N >> rd; a = J.bufa[in_buf] + (size_t)i * len; b = J.bufb[in_buf] + (size_t)i * len; }
    if (out_buf < 0) { oa = J.fin_a[i]; ob = J.fin_b[i]; }
    else { size_t len = N >> (rd + rounds); oa = J.bufa[out_buf] + (size_t)i * len; ob = J.bufb[out_buf] + (size_t)i * len; }
}

template <typename TA>
__device__ __forceinline__ void ps_round_body(const PsJob& J, int rd, int in_buf, int out_buf, size_t half, E2 r, int jb_log2, E2& a0, E2& a2,
                                              size_t first_tile, size_t tile_step) {
    using V = Val<TA>;
    const int BD = blockDim.x, tid = threadIdx.x;
    const int G = BD >> jb_log2;
    const int jj = tid & ((1 << jb_log2) - 1), g = tid >> jb_log2;
    const size_t ntiles = half >> jb_log2;
    // g = sum_i a_i b_i has no per-j factor, so the two evaluation sums stay unreduced (gl_wide.hpp) over the
    // whole grid-stride loop of this thread and are reduced once at the end.
    WE2 w0 = we2_zero(), w2 = we2_zero();
    const FoldR fr = fold_r(r);
    bool touched = false;
    for (size_t tile = first_tile; tile < ntiles; tile += tile_step) {
        const size_t j = (tile << jb_log2) + jj;
        const size_t jo = dpos(j, half);
        for (int i = g; i < J.npairs; i += G) {
            touched = true;
            const void* pa; const E2* pb; E2* oa; E2* ob;
            ps_io(J, rd, 1, in_buf, out_buf, i, pa, pb, oa, ob);
            TA xa, ya;
            E2 xb, yb;
            if (in_buf < 0) {
                load_pair<TA>(reinterpret_cast<const TA*>(pa) + 2 * j, xa, ya);
                load_pair<E2>(pb + 2 * j, xb, yb);
            } else {
                load_xy<TA, false>(reinterpret_cast<const TA*>(pa), j, half, xa, ya);
                load_xy<E2, false>(pb, j, half, xb, yb);
            }
            TA da = V::sub(ya, xa);
            E2 db = e2_sub(yb, xb);
            E2 vb = e2_add(yb, db);
            TA va = V::add(ya, da);
            if constexpr (std::is_same<TA, u64>::value) {
                wmac2(w0.A, xb.c0, xa, w0.C, xb.c1, xa);
                wmac2(w2.A, vb.c0, va, w2.C, vb.c1, va);
                const WAcc f0 = wacc_mul_init(xa, fr.r0, da), f1 = wacc_mul_init(0, fr.r1, da);
                store_e2(oa + jo, e2(wreduce(f0), wreduce(f1)));
            } else {
                we2_mac(w0, xb, xa);
                we2_mac(w2, vb, va);
                store_e2(oa + jo, e2_fold_wide(xa, da, fr));
            }
            store_e2(ob + jo, e2_fold_wide(xb, db, fr));
        }
    }
    if (touched) {  // (threads without work skip the six reductions: most of a small round's workgroup)
        a0 = e2_add(a0, we2_reduce(w0));
        a2 = e2_add(a2, we2_reduce(w2));
    }
}

// one round of every item's job; items share a 1-D grid (item y owns workgroups [blk0, blk0 + nblk))
__device__ __forceinline__ int find_ps_item(const PsItem* __restrict__ items, int nitems, int blk) {   // (see find_item)
    const int l = threadIdx.x & 63;
    const int b0 = l < nitems ? items[l].blk0 : 0x7fffffff;
    const unsigned long long m = __ballot(b0 <= blk);
    return __builtin_amdgcn_readfirstlane(__popcll(m) - 1);
}
__global__ __launch_bounds__(256) void k_ps_one(const PsJob* __restrict__ jobs, const PsItem* __restrict__ items, int nitems,
                                                const E2* __restrict__ chal, E2* __restrict__ partials, E2* __restrict__ res) {
    const int y = find_ps_item(items, nitems, blockIdx.x);
    const PsItem& I = items[y];
    const PsJob& J = jobs[I.job];
    const int rd = I.rd;
    const int nblocks = I.nblk, bx = (int)blockIdx.x - I.blk0;
    E2* sm = dyn_lds;
    const size_t half = (size_t)1 << (J.nvars - 1 - rd);
    E2 r = chal[J.r_off + rd];
    E2 a0 = e2_zero(), a2 = e2_zero();
    if (I.in_buf < 0) ps_round_body<u64>(J, rd, I.in_buf, I.out_buf, half, r, I.jb_log2, a0, a2, bx, nblocks);
    else ps_round_body<E2>(J, rd, I.in_buf, I.out_buf, half, r, I.jb_log2, a0, a2, bx, nblocks);
    E2 sv[2] = {a0, a2};
    block_sum_multi<2>(sv, sm);
    const E2 s0 = sv[0], s2 = sv[1];
    E2* part = partials + (size_t)y * SC_MAX_BLOCKS * 4;
    if (threadIdx.x == 0) {
        if (nblocks == 1) { res[J.sums_slot + 2 * rd] = s0; res[J.sums_slot + 2 * rd + 1] = s2; }
        else { part_store(part + (size_t)bx * 2, s0); part_store(part + (size_t)bx * 2 + 1, s2); }
    }
    if (nblocks > 1) finish_partials(part, 2, tickets_of(partials) + y * 32, res + J.sums_slot + 2 * rd, sm, nblocks);
}
// two consecutive rounds per pass: thread j runs round t; the lane pair (2j', 2j'+1) then shares round t+1 on the folded
// values it holds in registers (even lane: a'b' at x' -> P0 and the fold of a; odd lane: at y' -> P1 and the fold of b;
// Pinf = (a'_y - a'_x)(b'_y - b'_x) split by Ext2 coordinate). s0' = P0, s2' = 2 P1 - P0 + 2 Pinf.
template <typename TA>
__device__ __forceinline__ void ps_step2_body(const PsJob& J, const PsItem& I, size_t half, E2 ra, E2 rb, E2* acc /*[4]*/, E2* sm) {
    using V = Val<TA>;
    const int tid = threadIdx.x;
    const bool odd = tid & 1;
    const size_t half2 = half >> 1, ntiles = half >> 8;
    const int nblocks = I.nblk, bx = (int)blockIdx.x - I.blk0;
    const FoldR fa = fold_r(ra);
    const FoldR fb = fold_r(odd ? e2_sub(e2_one(), rb) : rb);
    WE2 w0 = we2_zero(), w2 = we2_zero();
    W2 vm = w2_zero();
    WAcc vi = wacc_zero();
    // the (tile, pair) items of the workgroup as one stream, the next item's four loads requested before the current one is computed
    // on, by every iteration alike (the last asks for the first again): gp_first_round_body
    struct Item { TA xa, ya; E2 xb, yb; };
    const int np = J.npairs;
    auto fetch = [&](size_t tl, int i, Item& it) {
        const size_t jn = (tl << 8) + tid;
        const void* pa; const E2* pb; E2* oa; E2* ob;
        ps_io(J, I.rd, 2, I.in_buf, I.out_buf, i, pa, pb, oa, ob);
        if (I.in_buf < 0) {
            load_pair<TA>(reinterpret_cast<const TA*>(pa) + 2 * jn, it.xa, it.ya);
            load_pair<E2>(pb + 2 * jn, it.xb, it.yb);
        } else {
            load_xy<TA, false>(reinterpret_cast<const TA*>(pa), jn, half, it.xa, it.ya);
            load_xy<E2, false>(pb, jn, half, it.xb, it.yb);
        }
    };
    Item cur;
    if ((size_t)bx < ntiles) fetch(bx, 0, cur);
    for (size_t tile = bx; tile < ntiles; tile += nblocks) {
        const size_t j = (tile << 8) + tid;
        const size_t jo2 = dpos(j >> 1, half2);
        for (int i = 0; i < np; i++) {
            const bool more_i = i + 1 < np, more_t = tile + nblocks < ntiles;
            Item nxt;
            fetch(more_i || !more_t ? tile : tile + nblocks, more_i ? i + 1 : 0, nxt);
            const void* pa; const E2* pb; E2* oa; E2* ob;
            ps_io(J, I.rd, 2, I.in_buf, I.out_buf, i, pa, pb, oa, ob);
            const TA xa = cur.xa, ya = cur.ya;
            const E2 xb = cur.xb, yb = cur.yb;
            TA da = V::sub(ya, xa);
            E2 db = e2_sub(yb, xb);
            E2 vb = e2_add(yb, db);
            TA va = V::add(ya, da);
            E2 ma;  // folded a
            if constexpr (std::is_same<TA, u64>::value) {
                wmac2(w0.A, xb.c0, xa, w0.C, xb.c1, xa);
                wmac2(w2.A, vb.c0, va, w2.C, vb.c1, va);
                const WAcc f0 = wacc_mul_init(xa, fa.r0, da), f1 = wacc_mul_init(0, fa.r1, da);
                ma = e2(wreduce(f0), wreduce(f1));
            } else {
                we2_mac(w0, xb, xa);
                we2_mac(w2, vb, va);
                ma = e2_fold_wide(xa, da, fa);
            }
            const E2 mb = e2_fold_wide(xb, db, fa);
            const E2 oa_ = swap_lane(ma), ob_ = swap_lane(mb);
            const E2 ea = e2_sub(oa_, ma), eb = e2_sub(ob_, mb);  // +-(y' - x'); the sign cancels in the product
            w2_mac(vm, ma, mb);
            const u64 b = odd ? eb.c1 : eb.c0, d = odd ? eb.c0 : eb.c1;
            const u64 c = odd ? ea.c1 : gl_mul7_lazy(ea.c1);
            wmac_pair(vi, ea.c0, b, c, d);
            const E2 fx = odd ? mb : ma, fd = odd ? eb : ea;
            store_e2((odd ? ob : oa) + jo2, e2_fold_wide(fx, fd, fb));
            cur = nxt;
        }
    }
    acc[0] = we2_reduce(w0);
    acc[1] = we2_reduce(w2);
    const E2 mine = w2_reduce(vm);
    const u64 ip = wreduce(vi);
    E2 v[5] = {acc[0], acc[1], odd ? e2_zero() : mine, odd ? mine : e2_zero(), odd ? e2(0, ip) : e2(ip, 0)};
    block_sum_multi<5>(v, sm);                                          // results valid in thread 0
    acc[0] = v[0]; acc[1] = v[1];
    acc[2] = v[2];
    acc[3] = e2_add(e2_sub(e2_dbl(v[3]), v[2]), e2_dbl(v[4]));
}
// The same two rounds of an eq-factored job (PsJob::eq_n): no b table, and everything a round needs is LINEAR in the tables.
// Thread j'' owns the four entries 4j'' .. 4j''+3 of every table (p = b0 + 2 b1: b0 is folded with r_t, b1 with r_(t+1)):
//  * a_i is only folded: a''_i[j''] = sum_p c_p a_i[4j''+p], c = (1-r_t, r_t) x (1-r_(t+1), r_(t+1)) - one four-term dot product
//    and one reduction per output entry instead of three folds;
//  * with A = sum_i kappa_i a_i (built once by k_ps_eq_A ahead of round 0; a_0 itself when eq_single) and U_p = sum_j'' eq(z'_(t+2..);
//    j'') A[4j''+p], both rounds' sums are combinations of the four U_p:  S_0 = (1-z) U_0 + z U_2, S_1 = (1-z) U_1 + z U_3 with
//    z = z'_(t+1) (round t), S'_0 = (1-r_t) U_0 + r_t U_1, S'_1 = (1-r_t) U_2 + r_t U_3 (round t+1). The eq factor of j'' is
//    lo[tid] * SUF_(t+10)[tile] (PsEqPoint): the second is uniform over the tile (zero outside a windowed table's block: nothing
//    is added there), the first multiplies the thread's four sums once, after the loop.
// Folded tables are written 4-WAY de-interleaved (entry e of a table of length len at (e & 3) len/4 + (e >> 2)) so that the next
// pass's four loads are 16 B per lane and contiguous across the wave; the pass ahead of the tail writes the tail's layout.
__device__ __forceinline__ void w2_mac_pre(W2& w, u64 a0, u64 a1, u64 a17, E2 b) {   // w += (a0, a1) * b, a17 = 7 a1 (any residue)
    wmac_pair(w.c0, a0, b.c0, a17, b.c1);
    wmac_pair(w.c1, a0, b.c1, a1, b.c0);
}
struct Fold4 { E2 c[4]; u64 c17[4]; };   // the four-term double fold's constants (c17 = 7 c.c1 as any residue)
__device__ __forceinline__ Fold4 fold4(const E2* __restrict__ c) {   // c = the pass's (1-ra)(1-rb), ra (1-rb), (1-ra) rb, ra rb (PsJob::eq_scal)
    Fold4 f;
#pragma unroll
    for (int p = 0; p < 4; p++) { f.c[p] = c[p]; f.c17[p] = gl_mul7_lazy(f.c[p].c1); }
    return f;
}
__device__ __forceinline__ E2 fold4_apply(const Fold4& f, const u64 (&v)[4]) {
    WAcc a = wacc_mul_init(0, f.c[0].c0, v[0]), b = wacc_mul_init(0, f.c[0].c1, v[0]);
#pragma unroll
    for (int p = 1; p < 4; p++) wmac2(a, f.c[p].c0, v[p], b, f.c[p].c1, v[p]);
    return e2(wreduce(a), wreduce(b));
}
__device__ __forceinline__ E2 fold4_apply(const Fold4& f, const E2 (&v)[4]) {
    WAcc a = wacc_pair_init(0, f.c[0].c0, v[0].c0, f.c17[0], v[0].c1), b = wacc_pair_init(0, f.c[0].c0, v[0].c1, f.c[0].c1, v[0].c0);
#pragma unroll
    for (int p = 1; p < 4; p++) {
        wmac_pair(a, f.c[p].c0, v[p].c0, f.c17[p], v[p].c1);
        wmac_pair(b, f.c[p].c0, v[p].c1, f.c[p].c1, v[p].c0);
    }
    return e2(wreduce(a), wreduce(b));
}
// the four entries of thread jq: natural order (round-0 inputs) or 4-way de-interleaved (tables this kernel wrote), q = len / 4
template <typename T>
__device__ __forceinline__ void load4(const T* tab, bool natural, size_t jq, size_t q, T (&v)[4]) {
    if (natural) { load_pair<T>(tab + 4 * jq, v[0], v[1]); load_pair<T>(tab + 4 * jq + 2, v[2], v[3]); }
    else {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if constexpr (std::is_same<T, u64>::value) v[p] = gload_u64(tab + (size_t)p * q + jq);
            else v[p] = gload_e2(tab + (size_t)p * q + jq);
        }
    }
}
constexpr int PS_EQ_OUT_TAIL = 1;   // PsItem::pad of an eq-factored item: the pass writes the tail's (2-way) layout
// The workgroups of an item form I.jb_log2 groups of I.nblk / groups: group g folds the tables [g gs, (g+1) gs) (a job of 27 tables
// would otherwise be a serial chain of 27 dependent load -> fold -> store steps per tile), group 0 also owns A and the sums.
template <typename TA>
__device__ __forceinline__ bool ps_eq_step2_body(const PsJob& J, const PsItem& I, size_t half, E2* acc /*[4]*/, E2* sm) {
    constexpr bool BASE = std::is_same<TA, u64>::value;
    const int tid = threadIdx.x;
    const size_t q = half >> 1, ntiles = q >> 8;   // q output entries per table; the host plans such a pass at half >= 2^9 only
    const int ngroups = I.jb_log2, nblocks = I.nblk / ngroups;
    const int grp = ((int)blockIdx.x - I.blk0) / nblocks, bx = ((int)blockIdx.x - I.blk0) % nblocks;
    const int rd = I.rd, nm = J.eq_n;
    const int gs = (nm + ngroups - 1) / ngroups, m0 = grp * gs, m1 = m0 + gs < nm ? m0 + gs : nm;
    const bool single = J.eq_single != 0, natural = I.in_buf < 0, out_tail = (I.pad & PS_EQ_OUT_TAIL) != 0;
    const bool sums = grp == 0;
    const Fold4 f4 = fold4(J.eq_scal + 2 * J.nvars + 1 + nm + J.nvars + 4 * (rd >> 1));
    const E2* __restrict__ hi = J.eq_suf + (((size_t)1 << (J.nvars - (rd + 10))) - 1);   // SUF_(rd+10): one entry per tile
    const E2* __restrict__ Ain = single ? nullptr : (natural ? J.eqA0 : J.bufA[I.in_buf]);
    const E2 l8 = gload_e2(J.eq_lo + (size_t)(rd + 1) * 384 + tid);   // eq(z'_(rd+2..rd+9); tid): the thread's factor, applied after the loop
    W2 U[4];
#pragma unroll
    for (int p = 0; p < 4; p++) U[p] = w2_zero();
    auto member_tab = [&](int i) { return natural ? reinterpret_cast<const TA*>(J.a[i]) : reinterpret_cast<const TA*>(J.bufa[I.in_buf]) + (size_t)i * 4 * q; };
    // the loads of the next (tile, table) are in flight while the current one is computed on: requested into registers of their own
    // before the current values are touched, by every iteration alike (the last asks for the first again) - gp_first_round_body
    TA cv[4];
    E2 nhv = e2_zero();
    if ((size_t)bx < ntiles && m0 < m1) { load4<TA>(member_tab(m0), natural, ((size_t)bx << 8) + tid, q, cv); nhv = gload_e2(hi + bx); }
    for (size_t tile = bx; tile < ntiles; tile += nblocks) {
        const size_t jq = (tile << 8) + tid;
        const size_t jo = out_tail ? dpos(jq, q) : (jq & 3) * (q >> 2) + (jq >> 2);
        const E2 hv = nhv;
        const bool more = tile + nblocks < ntiles;
        if (more) nhv = gload_e2(hi + tile + nblocks);
        const bool live = (hv.c0 | hv.c1) != 0;   // (uniform)
        const u64 h17 = gl_mul7_lazy(hv.c1);
        E2 Av[4];
        if (!single && sums) load4<E2>(Ain, false, jq, q, Av);
        for (int i = m0; i < m1; i++) {
            TA nv[4];
            {
                const bool more_i = i + 1 < m1;
                load4<TA>(member_tab(more_i ? i + 1 : m0), natural, more_i || !more ? jq : ((tile + nblocks) << 8) + tid, q, nv);
            }
            const TA (&v)[4] = cv;
            if (single && live) {
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    if constexpr (BASE) wmac2(U[p].c0, hv.c0, v[p], U[p].c1, hv.c1, v[p]);
                    else w2_mac_pre(U[p], hv.c0, hv.c1, h17, v[p]);
                }
            }
            gstore_e2(J.bufa[I.out_buf] + (size_t)i * q + jo, fold4_apply(f4, v));
#pragma unroll
            for (int p = 0; p < 4; p++) cv[p] = nv[p];
        }
        if (!single && sums) {
            if (live) {
#pragma unroll
                for (int p = 0; p < 4; p++) w2_mac_pre(U[p], hv.c0, hv.c1, h17, Av[p]);
            }
            gstore_e2(J.bufA[I.out_buf] + jo, fold4_apply(f4, Av));
        }
    }
    if (!sums) return false;
    E2 v[4];
#pragma unroll
    for (int p = 0; p < 4; p++) v[p] = e2_mul(l8, w2_reduce(U[p]));
    block_sum_multi<4>(v, sm);   // this workgroup's share of U_0 .. U_3, valid in thread 0
#pragma unroll
    for (int p = 0; p < 4; p++) acc[p] = v[p];
    return true;
}
// U_0 .. U_3 -> the four values of the two rounds (see ps_eq_step2_body), once per job
struct PsEqPost {
    const PsJob* J; int rd; E2 ra;
    __device__ __forceinline__ void operator()(E2* v) const {
        const E2* __restrict__ pre = J->eq_scal + 2 * rd;
        const E2 z = J->eq_scal[2 * J->nvars + 1 + J->eq_n + rd + 1];   // z'_(rd+1)
        const E2 S0 = e2_add(v[0], e2_mul(z, e2_sub(v[2], v[0]))), S1 = e2_add(v[1], e2_mul(z, e2_sub(v[3], v[1])));
        const E2 T0 = e2_add(v[0], e2_mul(ra, e2_sub(v[1], v[0]))), T1 = e2_add(v[2], e2_mul(ra, e2_sub(v[3], v[2])));
        v[0] = e2_mul(pre[0], S0);
        v[1] = e2_mul(pre[1], e2_sub(e2_dbl(S1), S0));
        v[2] = e2_mul(pre[2], T0);
        v[3] = e2_mul(pre[3], e2_sub(e2_dbl(T1), T0));
    }
};
// A = sum_i kappa_i a_i of the eq-factored jobs with several tables (base-field inputs in natural order), written 4-way
// de-interleaved for the first pass; one thread per four entries, job = jobs[ids[blockIdx.y]]
__global__ __launch_bounds__(256) void k_ps_eq_A(const PsJob* __restrict__ jobs, const int* __restrict__ ids) {
    const PsJob& J = jobs[ids[blockIdx.y]];
    const size_t q = ((size_t)1 << J.nvars) >> 2;
    const E2* __restrict__ kap = J.eq_scal + 2 * J.nvars + 1;
    for (size_t jq = (size_t)blockIdx.x * blockDim.x + threadIdx.x; jq < q; jq += (size_t)gridDim.x * blockDim.x) {
        WAcc w[4][2];
#pragma unroll
        for (int p = 0; p < 4; p++) { w[p][0] = wacc_zero(); w[p][1] = wacc_zero(); }
        u64 nv[4];
        load4<u64>(reinterpret_cast<const u64*>(J.a[0]), true, jq, q, nv);
        for (int i = 0; i < J.eq_n; i++) {
            u64 v[4];
#pragma unroll
            for (int p = 0; p < 4; p++) v[p] = nv[p];
            if (i + 1 < J.eq_n) load4<u64>(reinterpret_cast<const u64*>(J.a[i + 1]), true, jq, q, nv);
            const E2 k = kap[i];
#pragma unroll
            for (int p = 0; p < 4; p++) wmac2(w[p][0], k.c0, v[p], w[p][1], k.c1, v[p]);
        }
#pragma unroll
        for (int p = 0; p < 4; p++) gstore_e2(J.eqA0 + (size_t)p * q + jq, e2(wreduce(w[p][0]), wreduce(w[p][1])));
    }
}
void ps_eq_A(hipStream_t st, const PsJob* jobs, const int* ids, int nids, size_t max_quads) {
    if (nids <= 0) return;
    const size_t bx = std::min<size_t>((max_quads + 255) / 256, 1024);
    k_ps_eq_A<<<dim3((unsigned)std::max<size_t>(bx, 1), (unsigned)nids), 256, 0, st>>>(jobs, ids);
}
template <bool EQ>   // EQ: the items are eq-factored jobs (PsJob::eq_n)
__global__ __launch_bounds__(256) void k_ps_step2(const PsJob* __restrict__ jobs, const PsItem* __restrict__ items, int nitems,
                                                  const E2* __restrict__ chal, E2* __restrict__ partials, E2* __restrict__ res) {
    const int y = find_ps_item(items, nitems, blockIdx.x);
    const PsItem& I = items[y];
    const PsJob& J = jobs[I.job];
    const int rd = I.rd;
    const int nblocks = I.nblk, bx = (int)blockIdx.x - I.blk0;
    E2* sm = dyn_lds;
    const size_t half = (size_t)1 << (J.nvars - 1 - rd);
    E2 acc[4];
    if constexpr (EQ) {   // (a kernel of its own: both bodies in one took the other jobs' rounds from 166 to 206 registers, three waves to two)
        const bool sums = I.in_buf < 0 ? ps_eq_step2_body<u64>(J, I, half, acc, sm) : ps_eq_step2_body<E2>(J, I, half, acc, sm);
        if (!sums) return;   // (uniform: a workgroup that only folds tables)
        const int nb = I.nblk / I.jb_log2;   // the workgroups of group 0
        const PsEqPost post{&J, rd, chal[J.r_off + rd]};
        E2* part = partials + (size_t)y * SC_MAX_BLOCKS * 4;
        if (nb == 1) {
            if (threadIdx.x == 0) { post(acc); for (int t = 0; t < 4; t++) res[J.sums_slot + 2 * rd + t] = acc[t]; }
            return;
        }
        if (threadIdx.x == 0) for (int t = 0; t < 4; t++) part_store(part + (size_t)bx * 4 + t, acc[t]);
        finish_partials(part, 4, tickets_of(partials) + y * 32, res + J.sums_slot + 2 * rd, sm, nb, post);
        return;
    } else if (I.in_buf < 0) ps_step2_body<u64>(J, I, half, chal[J.r_off + rd], chal[J.r_off + rd + 1], acc, sm);
    else ps_step2_body<E2>(J, I, half, chal[J.r_off + rd], chal[J.r_off + rd + 1], acc, sm);
    const E2 s0 = acc[0], s2 = acc[1];  // summed over the workgroup by ps_step2_body (thread 0)
    E2* part = partials + (size_t)y * SC_MAX_BLOCKS * 4;
    if (threadIdx.x == 0) {
        E2* out = nblocks == 1 ? res + J.sums_slot + 2 * rd : nullptr;
        const E2 v[4] = {s0, s2, acc[2], acc[3]};
        for (int t = 0; t < 4; t++) { if (out) out[t] = v[t]; else part_store(part + (size_t)bx * 4 + t, v[t]); }
    }
    if (nblocks > 1) finish_partials(part, 4, tickets_of(partials) + y * 32, res + J.sums_slot + 2 * rd, sm, nblocks);
}
// rounds [tail_rd, nvars) of every job, one workgroup per job. A round here is a chain of dependent instructions and barriers,
// not work (measured: 7 us per round with 1024 threads, block sums through thread 0 and spilled registers; 147 us per launch), so:
//  * the first round reads its (pair, j) items from HBM four at a time (all loads of a batch in flight before the first product)
//    and writes the folds to LDS; every later round reads and writes the SAME LDS region (all reads, barrier, all writes,
//    barrier: the folds of a round fit its thread's registers), tables de-interleaved as in HBM so both reads are conflict-free;
//  * per-wave sums by DPP, then ONE wave adds the per-wave values (no serial loop in thread 0), overlapping the other waves' writes;
//  * once a round has at most 64 items, wave 0 finishes the job alone: no barrier, no LDS hop for the sums;
//  * an eq-factored job (PsJob::eq_n) never stores its b tables: the first round forms b = (kappa_i P) SUF_tail_rd on the fly.
constexpr int PS_TAIL_THREADS = 512;
constexpr size_t PS_TAIL_ITEMS_MAX = 4096;                 // (pair, j) items of a job's first tail round (host: flush_prodsum)
constexpr size_t PS_TAIL_LDS_E2 = 2 * PS_TAIL_ITEMS_MAX;   // their folds: the a tables, then the b tables
template <typename TA>
__device__ __forceinline__ void ps_tail_item(TA xa, TA ya, E2 xb, E2 yb, const FoldR& fr, WE2& w0, WE2& w2, E2& fa, E2& fb) {
    using V = Val<TA>;
    const TA da = V::sub(ya, xa);
    const E2 db = e2_sub(yb, xb);
    const E2 vb = e2_add(yb, db);
    const TA va = V::add(ya, da);
    if constexpr (std::is_same<TA, u64>::value) {
        wmac2(w0.A, xb.c0, xa, w0.C, xb.c1, xa);
        wmac2(w2.A, vb.c0, va, w2.C, vb.c1, va);
        const WAcc f0 = wacc_mul_init(xa, fr.r0, da), f1 = wacc_mul_init(0, fr.r1, da);
        fa = e2(wreduce(f0), wreduce(f1));
    } else {
        we2_mac(w0, xb, xa);
        we2_mac(w2, vb, va);
        fa = e2_fold_wide(xa, da, fr);
    }
    fb = e2_fold_wide(xb, db, fr);
}
// first tail round: items from HBM (or, the b side of an eq-factored job, from the point's suffix table), folds to T (fin_a / fin_b
// when it is also the last round)
template <typename TA>
__device__ __forceinline__ void ps_tail_first(const PsJob& J, const E2* __restrict__ kq, E2 r, E2* __restrict__ T, WE2& w0, WE2& w2, bool& touched) {
    const int BD = blockDim.x, tid = threadIdx.x;
    const int rd = J.tail_rd, hl = J.nvars - 1 - rd;
    const size_t half = (size_t)1 << hl, len = half << 1, items = (size_t)J.npairs * half;
    const bool natural = J.tail_buf < 0, last = hl == 0;
    const FoldR fr = fold_r(r);
    const E2* __restrict__ suf = J.eq_n ? J.eq_suf + (len - 1) : nullptr;   // SUF_tail_rd
    for (size_t t0 = tid; t0 < items; t0 += (size_t)4 * BD) {
        TA xa[4], ya[4];
        E2 xb[4], yb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const size_t t = t0 + (size_t)k * BD;
            if (t < items) {
                const int i = (int)(t >> hl);
                const size_t j = t & (half - 1);
                if (natural) {
                    load_pair<TA>(reinterpret_cast<const TA*>(J.a[i]) + 2 * j, xa[k], ya[k]);
                    load_pair<E2>(J.b[i] + 2 * j, xb[k], yb[k]);
                } else {
                    load_xy<TA, false>(reinterpret_cast<const TA*>(J.bufa[J.tail_buf]) + (size_t)i * len, j, half, xa[k], ya[k]);
                    if (suf) load_pair<E2>(suf + 2 * j, xb[k], yb[k]);
                    else load_xy<E2, false>(J.bufb[J.tail_buf] + (size_t)i * len, j, half, xb[k], yb[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const size_t t = t0 + (size_t)k * BD;
            if (t < items) {
                const int i = (int)(t >> hl);
                const size_t j = t & (half - 1);
                if (suf) { xb[k] = e2_mul(kq[i], xb[k]); yb[k] = e2_mul(kq[i], yb[k]); }
                E2 fa, fb;
                ps_tail_item<TA>(xa[k], ya[k], xb[k], yb[k], fr, w0, w2, fa, fb);
                touched = true;
                if (last) { J.fin_a[i][0] = fa; J.fin_b[i][0] = fb; }
                else { const size_t o = (size_t)i * half + dpos(j, half); T[o] = fa; T[items + o] = fb; }
            }
        }
    }
}
__global__ __launch_bounds__(PS_TAIL_THREADS) void k_ps_tail(const PsJob* __restrict__ jobs, const E2* __restrict__ chal, E2* __restrict__ res) {
    __shared__ PsJob Jl;   // (descriptor in LDS: see k_st_tail)
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(jobs + blockIdx.x);
        for (unsigned k = threadIdx.x; k < sizeof(PsJob) / 4; k += blockDim.x) reinterpret_cast<unsigned*>(&Jl)[k] = src[k];
    }
    __syncthreads();
    const PsJob& J = Jl;
    E2* T = dyn_lds;
    const int BD = blockDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = BD >> 6;
    const int R = J.nvars - J.tail_rd;   // rounds of this tail
    // round sums to the (host-memory) result buffer once, at the end; challenges fetched once
    __shared__ E2 keep[2 * 32], rch[32], red[2][2 * 16], kq[PS_MAX_PAIRS];
    if (tid < R) rch[tid] = chal[J.r_off + J.tail_rd + tid];
    if (tid < J.eq_n) kq[tid] = e2_mul(J.eq_scal[2 * J.nvars + 1 + tid], J.eq_scal[2 * J.nvars]);   // kappa_i P_tail_rd
    __syncthreads();
    int hl = J.nvars - 1 - J.tail_rd;
    size_t half = (size_t)1 << hl, items = (size_t)J.npairs * half;
    // a round's per-wave sums -> red[round parity][wave]; wave 0 adds them (after the barrier that follows) into keep[]
    auto wave_part = [&](int r, WE2& w0, WE2& w2, bool touched) {
        E2 s0 = e2_zero(), s2 = e2_zero();
        if (touched) { s0 = we2_reduce(w0); s2 = we2_reduce(w2); }
        s0 = wave_sum(s0); s2 = wave_sum(s2);
        if (lane == 0) { red[r & 1][2 * wave] = s0; red[r & 1][2 * wave + 1] = s2; }
    };
    auto wave0_total = [&](int r) {
        E2 s0 = lane < nw ? red[r & 1][2 * lane] : e2_zero(), s2 = lane < nw ? red[r & 1][2 * lane + 1] : e2_zero();
        s0 = wave_sum(s0); s2 = wave_sum(s2);
        if (lane == 0) { keep[2 * r] = s0; keep[2 * r + 1] = s2; }
    };
    {
        WE2 w0 = we2_zero(), w2 = we2_zero();
        bool touched = false;
        if (J.tail_buf < 0) ps_tail_first<u64>(J, kq, rch[0], T, w0, w2, touched);
        else ps_tail_first<E2>(J, kq, rch[0], T, w0, w2, touched);
        wave_part(0, w0, w2, touched);
        __syncthreads();
        if (wave == 0) wave0_total(0);
    }
    int r = 1;
    // rounds on the LDS tables, whole workgroup: at most PS_TAIL_ITEMS_MAX / 2 / PS_TAIL_THREADS = 4 items per thread
    for (; r < R && (items >> 1) > 64; r++) {
        const size_t in_items = items, in_half = half;
        hl--; half >>= 1; items >>= 1;
        const FoldR fr = fold_r(rch[r]);
        WE2 w0 = we2_zero(), w2 = we2_zero();
        bool touched = false;
        E2 fa[4], fb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const size_t t = (size_t)tid + (size_t)k * BD;
            if (t < items) {
                const size_t i = t >> hl, j = t & (half - 1);
                const E2* pa = T + i * in_half;
                const E2* pb = T + in_items + i * in_half;
                ps_tail_item<E2>(pa[j], pa[half + j], pb[j], pb[half + j], fr, w0, w2, fa[k], fb[k]);
                touched = true;
            }
        }
        wave_part(r, w0, w2, touched);
        __syncthreads();   // every read of this round's tables is done; the per-wave sums are in red[]
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const size_t t = (size_t)tid + (size_t)k * BD;
            if (t < items) {
                const size_t i = t >> hl, j = t & (half - 1);
                const size_t o = i * half + dpos(j, half);
                T[o] = fa[k]; T[items + o] = fb[k];
            }
        }
        if (wave == 0) wave0_total(r);
        __syncthreads();
    }
    if (wave != 0) return;
    // at most 64 items per round from here on (128 in the table pair read first): wave 0 alone, in program order
    for (; r < R; r++) {
        const size_t in_items = items, in_half = half;
        hl--; half >>= 1; items >>= 1;
        const bool last = hl == 0 && r == R - 1;
        const FoldR fr = fold_r(rch[r]);
        WE2 w0 = we2_zero(), w2 = we2_zero();
        E2 fa = e2_zero(), fb = e2_zero();
        const size_t t = lane;
        const size_t i = t >> hl, j = t & (half - 1);
        const bool mine = t < items;
        if (mine) {
            const E2* pa = T + i * in_half;
            const E2* pb = T + in_items + i * in_half;
            ps_tail_item<E2>(pa[j], pa[half + j], pb[j], pb[half + j], fr, w0, w2, fa, fb);
        }
        E2 s0 = mine ? we2_reduce(w0) : e2_zero(), s2 = mine ? we2_reduce(w2) : e2_zero();
        s0 = wave_sum(s0); s2 = wave_sum(s2);
        if (lane == 0) { keep[2 * r] = s0; keep[2 * r + 1] = s2; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the wave's LDS reads above stay ahead of the writes below)
        if (mine) {
            if (last) { J.fin_a[i][0] = fa; J.fin_b[i][0] = fb; }
            else { const size_t o = i * half + dpos(j, half); T[o] = fa; T[items + o] = fb; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (lane < 2 * R) res[J.sums_slot + 2 * (size_t)J.tail_rd + lane] = keep[lane];
}

// Factor tables of the eq-factored jobs' points (PsEqPoint). Every table is a product of at most two small factor tables kept in
// LDS: TOP = eq over the top 7 coordinates, MID_k = eq over coordinates k .. nvars-8 (the stored suffix tables reach at most 6 + 7
// bits at the sizes of this circuit; PS_EQ_MAX_VARS bounds the general case), so no entry costs more than ~10 dependent products
// and nothing waits on a table written by the same launch. grid = points x PS_EQ_PARTS, each part writes a slice of the outputs.
constexpr int PS_EQ_PARTS = 32, PS_EQ_TOPB = 7, PS_EQ_MID_MAX = 1024;
__device__ __forceinline__ E2 eq_bits(const E2* z, int first, int bits, unsigned v) {   // eq(z_(first .. first+bits-1); v)
    E2 acc = e2_one();
    for (int b = 0; b < bits; b++) {
        const E2 zz = z[first + b];
        acc = e2_mul(acc, (v >> b) & 1 ? zz : e2_sub(e2_one(), zz));
    }
    return acc;
}
__global__ __launch_bounds__(256) void k_ps_eq_prep(const PsEqPoint* __restrict__ pts, const E2* __restrict__ chal) {
    const PsEqPoint P = pts[blockIdx.x / PS_EQ_PARTS];
    const int part = blockIdx.x % PS_EQ_PARTS, tid = threadIdx.x, n = P.nvars;
    __shared__ E2 z[32], top[1 << PS_EQ_TOPB], mid[2 * PS_EQ_MID_MAX];
    if (tid < n) z[tid] = tid < P.w ? chal[P.point_off + tid] : e2((P.hib >> (tid - P.w)) & 1u, 0);
    __syncthreads();
    // TOP over coordinates n-tb .. n-1; MID_k over k .. n-tb-1 for k >= kmin, 2^(n-tb-k) entries at mid + 2^(n-tb-k) - 1
    const int tb = n < PS_EQ_TOPB ? n : PS_EQ_TOPB, nt = n - tb;
    for (int t = tid; t < (1 << tb); t += blockDim.x) top[t] = eq_bits(z, nt, tb, (unsigned)t);
    const int kmid = P.kmin < nt ? P.kmin : nt;
    const size_t mid_total = ((size_t)2 << (nt - kmid)) - 1;
    for (size_t q = tid; q < mid_total; q += blockDim.x) {
        const int lg = 63 - __clzll((long long)(q + 1));   // table of 2^lg entries, entry q + 1 - 2^lg
        mid[q] = eq_bits(z, nt - lg, lg, (unsigned)(q + 1 - ((size_t)1 << lg)));
    }
    __syncthreads();
    // suffix tables: SUF_k[x] = MID_k[x & (2^(nt-k) - 1)] * TOP[x >> (nt - k)] for k <= nt, a sub-cube of TOP above
    const size_t suf_total = ((size_t)2 << (n - P.kmin)) - 1;
    for (size_t q = (size_t)part * blockDim.x + tid; q < suf_total; q += (size_t)PS_EQ_PARTS * blockDim.x) {
        const int lg = 63 - __clzll((long long)(q + 1));   // = n - k
        const size_t x = q + 1 - ((size_t)1 << lg);
        const int k = n - lg;
        E2 v;
        if (k >= nt) v = eq_bits(z, k, lg, (unsigned)x);
        else {
            const int mb = nt - k;
            v = e2_mul(mid[((size_t)1 << mb) - 1 + (x & (((size_t)1 << mb) - 1))], top[x >> mb]);
        }
        gstore_e2(P.suf + q, v);
    }
    // per-round thread factors
    const size_t lo_total = n > 8 ? (size_t)(n - 8) * 384 : 0;
    for (size_t q = (size_t)part * blockDim.x + tid; q < lo_total; q += (size_t)PS_EQ_PARTS * blockDim.x) {
        const int rd = (int)(q / 384), t = (int)(q % 384);
        gstore_e2(P.lo + q, t < 256 ? eq_bits(z, rd + 1, 8, (unsigned)t) : eq_bits(z, rd + 2, 7, (unsigned)(t - 256)));
    }
}
void ps_eq_prep(hipStream_t st, const PsEqPoint* pts, int npts, const E2* chal) {
    if (npts > 0) k_ps_eq_prep<<<npts * PS_EQ_PARTS, 256, 0, st>>>(pts, chal);
}

int ps_plan_blocks(PsItem* items, int nitems, const PsJob* host_jobs, bool rounds2) {
    if (nitems > 0 && host_jobs[items[0].job].eq_n) {   // eq-factored items (a launch holds one kind): a tile = 256 threads x 4 entries
        size_t all_tiles = 0;
        for (int q = 0; q < nitems; q++) all_tiles += ((size_t)1 << (host_jobs[items[q].job].nvars - 1 - items[q].rd)) >> 9;
        // a workgroup pays ~1000 instructions per thread once (four reductions, four products, the block sums) against 100-150 per
        // table and tile: several tiles each when the launch has them, the tables of a many-table job dealt to groups of workgroups
        constexpr size_t target_blocks = 1024, serial_big = 16, serial_small = 6;   // (2048 / 512 workgroups: 91 / 65 us for the first launch against 68)
        const size_t per = std::min<size_t>(std::max<size_t>(all_tiles / target_blocks, 1), 16);
        const size_t budget = all_tiles >= 1024 ? serial_big : serial_small;   // (table, tile) steps per thread
        int blk = 0;
        for (int q = 0; q < nitems; q++) {
            PsItem& I = items[q];
            const PsJob& J = host_jobs[I.job];
            const size_t ntiles = ((size_t)1 << (J.nvars - 1 - I.rd)) >> 9;
            const size_t nb = std::min<size_t>(std::max<size_t>((ntiles + per - 1) / per, 1), (size_t)st_max_blocks());
            const size_t tiles_each = (ntiles + nb - 1) / nb;
            size_t groups = std::min<size_t>(std::max<size_t>((tiles_each * J.eq_n + budget - 1) / budget, 1), (size_t)J.eq_n);
            I.jb_log2 = (int)groups; I.blk0 = blk;   // (jb_log2 of an eq-factored item = its number of table groups)
            I.nblk = (int)(nb * groups);
            blk += I.nblk;
        }
        return blk;
    }
    size_t total = 0;
    for (int q = 0; q < nitems; q++) total += ((size_t)1 << (host_jobs[items[q].job].nvars - 1 - items[q].rd)) * host_jobs[items[q].job].npairs;
    int jb0 = 8;
    if (!rounds2) while (jb0 > 2 && (total << (8 - jb0)) < st_min_threads()) jb0--;
    // a workgroup pays ~10 reductions and 4-5 block sums once, whatever it processed: give it several tiles when the
    // launch has enough of them (the items have only 1-3 table pairs each, unlike the grand-product jobs)
    size_t all_tiles = 0;
    for (int q = 0; q < nitems; q++) all_tiles += ((size_t)1 << (host_jobs[items[q].job].nvars - 1 - items[q].rd)) >> jb0;
    constexpr size_t target_blocks = 1024;   // (4096 until round 5: 102 -> 55 us for the first launch of a prove in isolation, GPU time 1.875 -> 1.820 ms; 2048 / 512: 1.842 / 1.862)
    size_t per = all_tiles / target_blocks;
    if (per < 1) per = 1;
    if (per > 32) per = 32;
    int blk = 0;
    for (int q = 0; q < nitems; q++) {
        PsItem& I = items[q];
        const int hl = host_jobs[I.job].nvars - 1 - I.rd;
        int jb = jb0;
        if (jb > hl) jb = hl;
        const size_t ntiles = ((size_t)1 << hl) >> jb;
        I.jb_log2 = jb; I.blk0 = blk;
        I.nblk = (int)std::min<size_t>((ntiles + per - 1) / per, (size_t)st_max_blocks());
        blk += I.nblk;
    }
    return blk;
}
void ps_round(hipStream_t st, bool rounds2, const PsJob* jobs, const PsItem* items, int nitems, int grid, const E2* chal, E2* partials, E2* res, bool eq) {
    if (eq) {
        if (!rounds2) throw std::runtime_error("ps_round: eq-factored jobs run in fused pairs of rounds");
        k_ps_step2<true><<<grid, 256, SM_SLOTS * sizeof(E2), st>>>(jobs, items, nitems, chal, partials, res);
    } else if (rounds2) k_ps_step2<false><<<grid, 256, SM_SLOTS * sizeof(E2), st>>>(jobs, items, nitems, chal, partials, res);
    else k_ps_one<<<grid, 256, SM_SLOTS * sizeof(E2), st>>>(jobs, items, nitems, chal, partials, res);
}
size_t ps_tail_items_max() { return PS_TAIL_ITEMS_MAX; }
void ps_tail(hipStream_t st, const PsJob* jobs, int njobs, const E2* chal, E2* res) {
    const size_t lds = PS_TAIL_LDS_E2 * sizeof(E2);
    static const hipError_t attr = hipFuncSetAttribute((const void*)k_ps_tail, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)attr;
    k_ps_tail<<<njobs, PS_TAIL_THREADS, lds, st>>>(jobs, chal, res);
}

// debugging aid (HG_STAMP=1): device wall clock (100 MHz) at a point of a stream, also inside a replayed launch graph
__global__ void k_stamp(unsigned long long* t) { *t = wall_clock64(); }
void stamp(hipStream_t st, unsigned long long* slot) { k_stamp<<<1, 1, 0, st>>>(slot); }

// one launch instead of up to three memset nodes at the head of a prove (reduction tickets of both streams, the result-buffer prefix)
__global__ void k_clear_words(ClearSet c) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (int r = 0; r < 4; r++)
        for (size_t k = i; k < c.n[r]; k += step) c.p[r][k] = 0;
}
void clear_words(hipStream_t st, const ClearSet& c) {
    size_t most = std::max(std::max(c.n[0], c.n[1]), std::max(c.n[2], c.n[3]));
    if (most) k_clear_words<<<(unsigned)std::min<size_t>((most + 255) / 256, 256), 256, 0, st>>>(c);
}

__global__ void k_scatter_e2(const ScatterEnt* __restrict__ ents, size_t n, E2* __restrict__ dst_base) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst_base[ents[i].dst] = ents[i].src[0];
}
void scatter_e2(hipStream_t st, const ScatterEnt* ents, size_t n, E2* dst_base) {
    if (n) k_scatter_e2<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(ents, n, dst_base);
}

__global__ void k_set_e2(E2* dst, E2 v) {
    __hip_atomic_store(&dst->c1, (unsigned long long)v.c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&dst->c0, (unsigned long long)v.c0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void set_e2(hipStream_t st, E2* dst, E2 v) { k_set_e2<<<1, 1, 0, st>>>(dst, v); }

__global__ __launch_bounds__(TPB) void k_reduce_partials(const E2* __restrict__ partials, int nblocks, int nv, E2* __restrict__ out) {
    __shared__ E2 sm[TPB / 64];
    for (int v = 0; v < nv; v++) {
        E2 a = e2_zero();
        for (int b = threadIdx.x; b < nblocks; b += TPB) a = e2_add(a, partials[(size_t)b * nv + v]);
        a = block_sum(a, sm);
        if (threadIdx.x == 0) out[v] = a;
    }
}
void reduce_partials(hipStream_t st, const E2* partials, int nblocks, int nv, E2* out) {
    k_reduce_partials<<<1, TPB, 0, st>>>(partials, nblocks, nv, out);
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ E2 eq_bits(const E2* __restrict__ pt, int nbits, size_t idx) {
    E2 p = e2_one();
    for (int i = 0; i < nbits; i++) {
        E2 ri = pt[i];
        p = e2_mul(p, (idx >> i) & 1 ? ri : e2_sub(e2_one(), ri));
    }
    return p;
}
// high-index slice per workgroup: enough workgroups to fill the chip (>= 256 where the table allows), at least 4
// outputs per thread so the per-workgroup A/B setup amortises
__host__ __device__ inline size_t eq_k_for(int hi) {
    size_t all = (size_t)1 << hi;
    size_t k = all >> 8;
    if (k < 4) k = 4;
    if (k > 256) k = 256;
    if (k > all) k = all;
    return k;
}
// eq tables as an outer product: eq(r, idx) = A[idx & (2^lo - 1)] * B[idx >> lo]. A workgroup builds the 256-entry
// low table and a 256-entry slice of the high table in LDS (n multiplications per thread in total), then
// emits 65,536 entries with ONE extension multiplication each (instead of n each). Multi-claim form:
// out = sum_a alpha_a eq(r_a, .), alpha folded into B, accumulated claim by claim by the owning thread.
__global__ __launch_bounds__(256) void k_eq_jobs(const EqJob* __restrict__ jobs, const E2* __restrict__ chal) {
    const EqJob& J = jobs[blockIdx.y];
    const int n = J.n;
    const int lo = n < 8 ? n : 8;
    const int hi = n - lo;
    const size_t K = eq_k_for(hi);                         // high indices per workgroup
    const size_t nblk = ((size_t)1 << hi) / K;
    if (blockIdx.x >= nblk) return;
    __shared__ E2 A[256], B[256];
    const int t = threadIdx.x;
    const size_t hi_base = (size_t)blockIdx.x * K;
    const E2* base = J.point_dev ? J.point_dev : chal;
    for (int a = 0; a < J.cs.n; a++) {
        const E2* pt = base + J.cs.point_off[a];
        __syncthreads();
        if (t < (1 << lo)) A[t] = eq_bits(pt, lo, (size_t)t);
        if ((size_t)t < K) {
            E2 bv = eq_bits(pt + lo, hi, hi_base + t);
            B[t] = J.cs.unit_alpha ? bv : e2_mul(bv, chal[J.cs.alpha_off + a]);
        }
        __syncthreads();
        if (t < (1 << lo)) {
            E2 av = A[t];
            for (size_t kk = 0; kk < K; kk++) {
                size_t idx = ((hi_base + kk) << lo) | (size_t)t;
                E2 v = e2_mul(av, B[kk]);
                if (a > 0) v = e2_add(v, J.out[idx]);
                store_e2(J.out + idx, v);
            }
        }
    }
}
// out[i] = sum_t tabs[t * n + i]  (multi-claim eq tables are built one claim per job, then summed)
__global__ __launch_bounds__(TPB) void k_sum_tables(E2* __restrict__ out, const E2* __restrict__ tabs, int ntabs, size_t n) {
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * TPB) {
        E2 acc = tabs[i];
        for (int t = 1; t < ntabs; t++) acc = e2_add(acc, tabs[(size_t)t * n + i]);
        store_e2(out + i, acc);
    }
}
void sum_tables(hipStream_t st, E2* out, const E2* tabs, int ntabs, size_t n) {
    k_sum_tables<<<(unsigned)std::min<size_t>((n + TPB - 1) / TPB, 2048), TPB, 0, st>>>(out, tabs, ntabs, n);
}
// ---- two-launch form -----
__device__ __forceinline__ E2 eq_bits_tree(const E2* __restrict__ pt, int nbits, size_t idx) {  // the same product, balanced (independent factors: ILP)
    E2 f[16];
#pragma unroll
    for (int i = 0; i < 16; i++) f[i] = i < nbits ? ((idx >> i) & 1 ? pt[i] : e2_sub(e2_one(), pt[i])) : e2_one();
    if (nbits <= 4) return e2_mul(e2_mul(f[0], f[1]), e2_mul(f[2], f[3]));
    if (nbits <= 8) return e2_mul(e2_mul(e2_mul(f[0], f[1]), e2_mul(f[2], f[3])), e2_mul(e2_mul(f[4], f[5]), e2_mul(f[6], f[7])));
    E2 p = e2_mul(e2_mul(e2_mul(f[0], f[1]), e2_mul(f[2], f[3])), e2_mul(e2_mul(f[4], f[5]), e2_mul(f[6], f[7])));
    E2 q = e2_mul(e2_mul(e2_mul(f[8], f[9]), e2_mul(f[10], f[11])), e2_mul(e2_mul(f[12], f[13]), e2_mul(f[14], f[15])));
    return e2_mul(p, q);
}
// one workgroup per (job, claim, part): part p writes B[256 p .. 256 p + 255] (and A when p == 0)
__global__ __launch_bounds__(256) void k_eq_prep(const EqJob* __restrict__ jobs, int njobs, const E2* __restrict__ chal) {
    int jl = 0, jh = njobs - 1;
    while (jl < jh) {
        int mid = (jl + jh + 1) >> 1;
        if (jobs[mid].pblk0 <= (int)blockIdx.x) jl = mid; else jh = mid - 1;
    }
    const EqJob& J = jobs[jl];
    const int n = J.n, lo = n < 8 ? n : 8, hi = n - lo;
    const size_t nB = (size_t)1 << hi;
    const int nparts = (int)((nB + 255) / 256);
    const int local = (int)blockIdx.x - J.pblk0;
    const int a = local / nparts, part = local % nparts;
    const E2* pt = (J.point_dev ? J.point_dev : chal) + J.cs.point_off[a];
    E2* A = J.ab + (size_t)a * eq_ab_entries(n);
    E2* B = A + 256;
    const int t = threadIdx.x;
    if (part == 0 && t < (1 << lo)) store_e2(A + t, eq_bits_tree(pt, lo, (size_t)t));
    const size_t h = (size_t)part * 256 + t;
    if (h < nB) {
        E2 bv = eq_bits_tree(pt + lo, hi, h);
        if (!J.cs.unit_alpha) bv = e2_mul(bv, chal[J.cs.alpha_off + a]);
        store_e2(B + h, bv);
    }
}
constexpr int EQ_ROWS = 8;   // table rows (of 256 outputs) per workgroup of the fill launch, at most
__global__ __launch_bounds__(256) void k_eq_fill(const EqJob* __restrict__ jobs, int njobs) {
    int jl = 0, jh = njobs - 1;
    while (jl < jh) {
        int mid = (jl + jh + 1) >> 1;
        if (jobs[mid].blk0 <= (int)blockIdx.x) jl = mid; else jh = mid - 1;
    }
    const EqJob& J = jobs[jl];
    const int n = J.n, lo = n < 8 ? n : 8, hi = n - lo;
    const int t = threadIdx.x;
    if (t >= (1 << lo)) return;
    const size_t nrows = (size_t)1 << hi;
    const size_t r0 = (size_t)((int)blockIdx.x - J.blk0) * J.rows;
    const int cnt = (int)(r0 + J.rows < nrows ? J.rows : nrows - r0);
    const size_t stride = eq_ab_entries(n);
    const E2* ab = J.ab;
    E2* out = J.out + (r0 << lo) + t;
    // all row factors of a claim are fetched before the first product (independent loads, one wait); the products of all claims
    // of an output accumulate unreduced (gl_wide.hpp): two column accumulators, two reductions per output
    WAcc w0[EQ_ROWS], w1[EQ_ROWS];
#pragma unroll
    for (int k = 0; k < EQ_ROWS; k++) { w0[k] = wacc_zero(); w1[k] = wacc_zero(); }
    for (int a = 0; a < J.cs.n; a++) {
        const FoldR fa = fold_r(gload_e2(ab + a * stride + t));
        E2 bv[EQ_ROWS];
#pragma unroll
        for (int k = 0; k < EQ_ROWS; k++) bv[k] = gload_e2(ab + a * stride + 256 + r0 + (k < cnt ? k : 0));
#pragma unroll
        for (int k = 0; k < EQ_ROWS; k++) {
            wmac_pair(w0[k], fa.r0, bv[k].c0, fa.r17, bv[k].c1);
            wmac_pair(w1[k], fa.r0, bv[k].c1, fa.r1, bv[k].c0);
        }
    }
#pragma unroll
    for (int k = 0; k < EQ_ROWS; k++)
        if (k < cnt) gstore_e2(out + ((size_t)k << lo), e2(wreduce(w0[k]), wreduce(w1[k])));
}
EqAbGrid eq_ab_plan(EqJob* host_jobs, int njobs) {
    // rows per workgroup: 8 (2048 outputs) when that still gives the launch a few thousand workgroups, fewer for small batches
    size_t rows_total = 0;
    for (int q = 0; q < njobs; q++) rows_total += (size_t)1 << (host_jobs[q].n > 8 ? host_jobs[q].n - 8 : 0);
    int rows = EQ_ROWS;
    while (rows > 1 && rows_total / rows < 2048) rows >>= 1;
    int blk = 0, pblk = 0;
    for (int q = 0; q < njobs; q++) {
        const size_t nrows = (size_t)1 << (host_jobs[q].n > 8 ? host_jobs[q].n - 8 : 0);
        host_jobs[q].blk0 = blk;
        host_jobs[q].rows = rows;
        host_jobs[q].pblk0 = pblk;
        blk += (int)((nrows + rows - 1) / rows);
        pblk += host_jobs[q].cs.n * (int)((nrows + 255) / 256);
    }
    return EqAbGrid{pblk, blk};
}
void eq_jobs_ab(hipStream_t st, const EqJob* jobs, int njobs, EqAbGrid grid, const E2* chal) {
    k_eq_prep<<<grid.prep, 256, 0, st>>>(jobs, njobs, chal);
    k_eq_fill<<<grid.fill, 256, 0, st>>>(jobs, njobs);
}
void eq_jobs(hipStream_t st, const EqJob* jobs, int njobs, int max_n, const E2* chal) {
    int hi = max_n > 8 ? max_n - 8 : 0;
    size_t nblk = ((size_t)1 << hi) / eq_k_for(hi);
    k_eq_jobs<<<dim3((unsigned)nblk, njobs), 256, 0, st>>>(jobs, chal);
}

// ------------------------------------------------------------------------------------------------
// Lasso
EpRows ep_rows_all(int alpha) {
    EpRows r;
    for (int m = 0; m < 32; m++) r.row[m] = m < alpha ? (signed char)m : (signed char)-1;
    return r;
}
__global__ __launch_bounds__(TPB) void k_lasso_split(LassoDev L, const u64* __restrict__ input, u64* __restrict__ dims,
                                                     u64* __restrict__ e_polys, EpRows R, ColPow P, u64* __restrict__ col) {
    // The per-lookup / per-memory maps go to LDS first: indexed by values that come out of the data, they would otherwise be a chain
    // of dependent loads from the kernel-argument segment per row (45 us for 50 MB; the kernel sits at the head of the main stream).
    __shared__ u64 s_mask[32], s_uses[32], s_pv[32];
    __shared__ u32 s_cut[32];
    __shared__ int s_nm[32], s_mems[32][4], s_dim[32], s_row[32];
    if (threadIdx.x < 32) {
        const int t = threadIdx.x;
        s_mask[t] = L.lookup_mask[t]; s_uses[t] = L.lookup_uses[t]; s_pv[t] = P.v[t]; s_cut[t] = L.mem_cutoff[t];
        s_nm[t] = L.lookup_nmems[t]; s_dim[t] = L.mem_dim[t]; s_row[t] = R.row[t];
        for (int i = 0; i < 4; i++) s_mems[t][i] = L.lookup_mems[t][i];
    }
    __syncthreads();
    const size_t N = (size_t)1 << L.nu;
    for (size_t j = (size_t)blockIdx.x * TPB + threadIdx.x; j < N; j += (size_t)gridDim.x * TPB) {
        u32 idx[4] = {0, 0, 0, 0};
        u64 uses = 0;
        int l = -1;
        if (j < L.rows) {
            l = L.seg_lookup[j >> L.seg_shift];
            u64 v = input[j] & s_mask[l];  // truncate to sum(chunk_bits) (lasso.rs:388-389)
            idx[0] = (u32)(v & 0xFFFF); idx[1] = (u32)((v >> 16) & 0xFFFF);
            idx[2] = (u32)((v >> 32) & 0xFFFF); idx[3] = (u32)((v >> 48) & 0xFFFF);
            uses = s_uses[l];
        }
        if (dims) {
#pragma unroll
            for (int c = 0; c < 4; c++) dims[(size_t)c * N + j] = idx[c];
        }
        // C[j] = sum_m colpow[m] E_m[j]: at most four non-zero terms, those of the memories of the row's lookup (whether or not
        // their tables are materialised)
        if (col) {
            u64 cv = 0;
            if (l >= 0)
                for (int i = 0; i < s_nm[l]; i++) {
                    const int m = s_mems[l][i];
                    const u32 a = idx[s_dim[m]];
                    if (s_pv[m] && ((uses >> m) & 1) && a && a < s_cut[m]) cv = gl_add(cv, gl_mul_small(s_pv[m], a));
                }
            col[j] = cv;
        }
        for (int m = 0; m < L.alpha; m++) {
            if (s_row[m] < 0) continue;
            u32 a = idx[s_dim[m]];
            u64 ev = ((uses >> m) & 1) && a < s_cut[m] ? (u64)a : 0;  // T_s[a] (range.rs:15-17, 58-72)
            e_polys[(size_t)s_row[m] * N + j] = ev;
        }
    }
}
// the four 16-bit limbs only: the counter sorts need nothing else, so they can start while the E tables are still being written
__global__ __launch_bounds__(TPB) void k_lasso_dims(LassoDev L, const u64* __restrict__ input, u64* __restrict__ dims) {
    const size_t N = (size_t)1 << L.nu;
    for (size_t j = (size_t)blockIdx.x * TPB + threadIdx.x; j < N; j += (size_t)gridDim.x * TPB) {
        u64 v = 0;
        if (j < L.rows) v = input[j] & L.lookup_mask[L.seg_lookup[j >> L.seg_shift]];
#pragma unroll
        for (int c = 0; c < 4; c++) dims[(size_t)c * N + j] = (v >> (16 * c)) & 0xFFFF;
    }
}
void lasso_dims(hipStream_t st, const LassoDev& L, const u64* input, u64* dims) {
    size_t N = (size_t)1 << L.nu;
    k_lasso_dims<<<(unsigned)std::min<size_t>((N + TPB - 1) / TPB, 4096), TPB, 0, st>>>(L, input, dims);
}
void lasso_split(hipStream_t st, const LassoDev& L, const u64* input, u64* dims, u64* e_polys, const EpRows& rows, const ColPow* colpow, u64* col) {
    size_t N = (size_t)1 << L.nu;
    ColPow P;
    memset(&P, 0, sizeof(P));
    if (colpow) P = *colpow;
    k_lasso_split<<<(unsigned)std::min<size_t>((N + TPB - 1) / TPB, 4096), TPB, 0, st>>>(L, input, dims, e_polys, rows, P, colpow ? col : nullptr);
}

__global__ __launch_bounds__(TPB) void k_counter_starts(const u32* __restrict__ ks, size_t n, u32* __restrict__ starts) {
    for (size_t p = (size_t)blockIdx.x * TPB + threadIdx.x; p < n; p += (size_t)gridDim.x * TPB) {
        u32 key = ks[p];
        if (p == 0 || ks[p - 1] != key) starts[key] = (u32)p;
    }
}
__global__ __launch_bounds__(TPB) void k_counter_ranks(const u32* __restrict__ ks, const u32* __restrict__ rs, size_t n,
                                                       const u32* __restrict__ starts, u64* __restrict__ read_ts,
                                                       u64* __restrict__ final_cts) {
    for (size_t p = (size_t)blockIdx.x * TPB + threadIdx.x; p < n; p += (size_t)gridDim.x * TPB) {
        u32 key = ks[p];
        u32 rank = (u32)p - starts[key];
        read_ts[rs[p]] = rank;                                   // = number of earlier rows on the same address
        if (p + 1 == n || ks[p + 1] != key) final_cts[key] = (u64)rank + 1;
    }
}
// ---- stable radix sort of (key, value) pairs on keys of at most 18 bits -------------------------------------------------------------
// What the counters need (polynomialize, lasso.rs:170-204, is a sequential scan per memory: read_cts[j] = how many EARLIER rows hit
// the same address): sort the (address, row) pairs by address, rows of one address staying in row order; a row's rank is then its
// position in the address's run. Addresses are 16-bit limbs (18 bits with the chunk index), so two passes over 9-bit digits do it.
// One pass = three launches:
//   k_cs_hist     every WAVE owns a tile of 1024 consecutive pairs and counts its digits (LDS atomics) -> hist[digit][tile]
//                 (the first pass also produces the pairs themselves from the limb tables: Gen)
//   k_cs_scan     one workgroup per digit: exclusive scan of its row of hist (the tiles in order) in place, the digit's total
//   k_cs_scatter  every wave re-reads its tile 64 pairs at a time, in order: a pair's destination is
//                 base(digit) + scanned hist[digit][tile] + (pairs of the same digit seen earlier in this tile); the lanes of one step
//                 rank themselves with nine ballots (the mask of lanes holding the same digit, its population count below the lane).
// Stable by construction: tiles in order (the scan), steps in order, lanes in order. Replaces rocprim::radix_sort_pairs (14 launches
// for the 18-bit keys of all four chunks, 160 us) with 6 launches.
constexpr int CS_BITS = 9, CS_BINS = 1 << CS_BITS, CS_WTILE = 1024, CS_WAVES = 4;
struct CsGenNone { __device__ __forceinline__ void operator()(size_t, u32&, u32&) const {} };
// Gen(q, key, val): produces pair q (first pass) - or nothing (CsGenNone: the pairs are read from keys_in / vals_in)
template <bool GEN, typename Gen>
__global__ __launch_bounds__(64 * CS_WAVES) void k_cs_hist(const u32* __restrict__ keys_in, u32* __restrict__ keys_gen, u32* __restrict__ vals_gen, size_t n, int sh,
                                                          u32* __restrict__ hist, size_t nwt, Gen gen) {
    __shared__ u32 cnt[CS_WAVES][CS_BINS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t wt = (size_t)blockIdx.x * CS_WAVES + wave;
    for (int d = lane; d < CS_BINS; d += 64) cnt[wave][d] = 0;
    __builtin_amdgcn_wave_barrier();
    if (wt < nwt) {
        const size_t i0 = wt * CS_WTILE;
        for (int r = 0; r < CS_WTILE / 64; r++) {
            const size_t i = i0 + (size_t)r * 64 + lane;
            if (i < n) {
                u32 key, val = 0;
                if (GEN) { gen(i, key, val); keys_gen[i] = key; vals_gen[i] = val; }
                else key = keys_in[i];
                atomicAdd(&cnt[wave][(key >> sh) & (CS_BINS - 1)], 1u);
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int d = lane; d < CS_BINS; d += 64) hist[(size_t)d * nwt + wt] = cnt[wave][d];
    }
}
__global__ __launch_bounds__(256) void k_cs_scan(u32* __restrict__ hist, size_t nwt, u32* __restrict__ totals) {
    __shared__ u32 part[256];
    u32* row = hist + (size_t)blockIdx.x * nwt;
    const size_t per = (nwt + 255) / 256, b = (size_t)threadIdx.x * per, e = b + per < nwt ? b + per : nwt;
    u32 s = 0;
    for (size_t i = b; i < e; i++) s += row[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {   // inclusive scan of the 256 chunk sums
        const u32 v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    u32 run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (size_t i = b; i < e; i++) { const u32 v = row[i]; row[i] = run; run += v; }
    if (threadIdx.x == 255) totals[blockIdx.x] = part[255];
}
__global__ __launch_bounds__(64 * CS_WAVES) void k_cs_scatter(const u32* __restrict__ keys_in, const u32* __restrict__ vals_in, u32* __restrict__ keys_out,
                                                             u32* __restrict__ vals_out, size_t n, int sh, const u32* __restrict__ hist,
                                                             const u32* __restrict__ totals, size_t nwt) {
    __shared__ u32 base[CS_BINS];
    __shared__ u32 off[CS_WAVES][CS_BINS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // base[d] = pairs with a smaller digit: exclusive scan of the 512 totals (two per thread)
    {
        const u32 a = totals[2 * threadIdx.x], b = totals[2 * threadIdx.x + 1];
        base[threadIdx.x] = a + b;   // (slots 0..255 used as scan scratch first)
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const u32 v = (int)threadIdx.x >= o ? base[threadIdx.x - o] : 0;
            __syncthreads();
            base[threadIdx.x] += v;
            __syncthreads();
        }
        const u32 excl = threadIdx.x ? base[threadIdx.x - 1] : 0;
        __syncthreads();
        base[2 * threadIdx.x] = excl;
        base[2 * threadIdx.x + 1] = excl + a;
        __syncthreads();
    }
    const size_t wt = (size_t)blockIdx.x * CS_WAVES + wave;
    if (wt >= nwt) return;
    for (int d = lane; d < CS_BINS; d += 64) off[wave][d] = base[d] + hist[(size_t)d * nwt + wt];
    __builtin_amdgcn_wave_barrier();
    const size_t i0 = wt * CS_WTILE;
    const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    for (int r = 0; r < CS_WTILE / 64; r++) {
        const size_t i = i0 + (size_t)r * 64 + lane;
        const bool valid = i < n;
        const u32 key = valid ? keys_in[i] : 0, val = valid ? vals_in[i] : 0;
        const u32 d = (key >> sh) & (CS_BINS - 1);
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < CS_BITS; bit++) {
            const bool on = (d >> bit) & 1;
            const unsigned long long bal = __ballot(valid && on);
            m &= on ? bal : ~bal;
        }
        if (valid) {
            const u32 rank = (u32)__popcll(m & lt);
            const u32 pos = off[wave][d] + rank;
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __builtin_amdgcn_wave_barrier();   // every lane has read off[] before the leaders move it
        if (valid && (m & lt) == 0) off[wave][d] += (u32)__popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
}
static size_t cs_tiles(size_t n) { return (n + CS_WTILE - 1) / CS_WTILE; }
static size_t cs_temp_bytes(size_t n) { return (cs_tiles(std::max<size_t>(n, 1)) * CS_BINS + CS_BINS) * sizeof(u32); }
// sorts n pairs by the low `bits` (<= 18) bits of the key, stably. First pass: pairs generated by `gen` into (keys, vals); result in
// (keys, vals) again - (keys2, vals2) is the ping-pong buffer.
template <typename Gen>
static void cs_sort_pairs(hipStream_t st, void* temp, size_t temp_bytes, u32* keys, u32* keys2, u32* vals, u32* vals2, size_t n, Gen gen) {
    if (n == 0) return;
    if (temp_bytes < cs_temp_bytes(n)) throw std::runtime_error("counter sort: scratch too small");
    const size_t nwt = cs_tiles(n);
    u32* hist = static_cast<u32*>(temp);
    u32* totals = hist + nwt * CS_BINS;
    const unsigned grid = (unsigned)((nwt + CS_WAVES - 1) / CS_WAVES);
    k_cs_hist<true, Gen><<<grid, 64 * CS_WAVES, 0, st>>>(nullptr, keys, vals, n, 0, hist, nwt, gen);
    k_cs_scan<<<CS_BINS, 256, 0, st>>>(hist, nwt, totals);
    k_cs_scatter<<<grid, 64 * CS_WAVES, 0, st>>>(keys, vals, keys2, vals2, n, 0, hist, totals, nwt);
    k_cs_hist<false, CsGenNone><<<grid, 64 * CS_WAVES, 0, st>>>(keys2, nullptr, nullptr, n, CS_BITS, hist, nwt, CsGenNone());
    k_cs_scan<<<CS_BINS, 256, 0, st>>>(hist, nwt, totals);
    k_cs_scatter<<<grid, 64 * CS_WAVES, 0, st>>>(keys2, vals2, keys, vals, n, CS_BITS, hist, totals, nwt);
}

static void check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
size_t lasso_counter_temp_bytes(size_t n) { return cs_temp_bytes(n); }
struct CsGenChunk {   // pair q of one counter memory: (address, row) of the q-th row that touches it (segment list in LassoDev)
    LassoDev L; int m; const u64* dim;
    __device__ __forceinline__ void operator()(size_t q, u32& key, u32& val) const {
        const size_t smask = ((size_t)1 << L.seg_shift) - 1;
        const size_t row = ((size_t)L.cnt_segs[m][q >> L.seg_shift] << L.seg_shift) | (q & smask);
        key = (u32)dim[row];
        val = (u32)row;
    }
};
void lasso_counters(hipStream_t st, const LassoDev& L, int m, const u64* dims, u64* read_ts, u64* final_cts, void* temp,
                    size_t temp_bytes, u32* keys, u32* keys_sorted, u32* rows_in, u32* rows_sorted, u32* starts) {
    const size_t N = (size_t)1 << L.nu;
    check_hip(hipMemsetAsync(read_ts, 0, N * sizeof(u64), st), "clear read_ts");
    check_hip(hipMemsetAsync(final_cts, 0, 65536 * sizeof(u64), st), "clear final_cts");
    const size_t cnt = (size_t)L.cnt_nsegs[m] << L.seg_shift;  // rows whose lookup type uses memory m (lasso.rs:181-183)
    if (cnt == 0) return;
    int grid = grid_for(cnt);
    // stable sort on the 16-bit address keeps the rows of one address in row order (the pairs are produced by the first pass)
    cs_sort_pairs(st, temp, temp_bytes, keys, keys_sorted, rows_in, rows_sorted, cnt, CsGenChunk{L, m, dims + (size_t)L.mem_dim[m] * N});
    k_counter_starts<<<grid, TPB, 0, st>>>(keys, cnt, starts);
    k_counter_ranks<<<grid, TPB, 0, st>>>(keys, rows_in, cnt, starts, read_ts, final_cts);
}

// ---- all counter memories in one sort -------------------------------------------------------------------------------------
struct CounterPlan { int nchunks; int chunk[4]; size_t off[5]; };  // rows of chunk[q] occupy positions [off[q], off[q+1]) before the sort
static CounterPlan counter_plan(const LassoDev& L, unsigned chunk_mask) {
    CounterPlan P;
    memset(&P, 0, sizeof(P));
    for (int c = 0; c < 4; c++)
        if ((chunk_mask >> c) & 1) {
            P.chunk[P.nchunks] = c;
            P.off[P.nchunks + 1] = P.off[P.nchunks] + ((size_t)L.cnt_nsegs[c] << L.seg_shift);
            P.nchunks++;
        }
    return P;
}
size_t lasso_counters_all_elems(const LassoDev& L, unsigned chunk_mask) { CounterPlan P = counter_plan(L, chunk_mask); return P.off[P.nchunks]; }
size_t lasso_counters_all_temp_bytes(size_t n_elems) { return cs_temp_bytes(n_elems); }
struct CsGenAll {   // pair q of the concatenated chunks: ((chunk, address), row); rows ascend inside a chunk and the sort is stable
    LassoDev L; CounterPlan P; const u64* dims;
    __device__ __forceinline__ void operator()(size_t q, u32& key, u32& val) const {
        const size_t N = (size_t)1 << L.nu;
        const size_t smask = ((size_t)1 << L.seg_shift) - 1;
        int s = 0;
        while (s + 1 < P.nchunks && q >= P.off[s + 1]) s++;
        const int c = P.chunk[s];
        const size_t local = q - P.off[s];
        const size_t row = ((size_t)L.cnt_segs[c][local >> L.seg_shift] << L.seg_shift) | (local & smask);
        key = ((u32)c << 16) | (u32)dims[(size_t)L.mem_dim[c] * N + row];
        val = (u32)row;
    }
};
__global__ __launch_bounds__(TPB) void k_counter_ranks_all(const u32* __restrict__ ks, const u32* __restrict__ vs, size_t n, const u32* __restrict__ starts,
                                                           CounterOut out) {
    for (size_t p = (size_t)blockIdx.x * TPB + threadIdx.x; p < n; p += (size_t)gridDim.x * TPB) {
        const u32 key = ks[p];
        const u32 rank = (u32)p - starts[key];
        const int c = (int)(key >> 16);
        out.read_ts[c][vs[p]] = rank;                                                     // = number of earlier rows on the same address
        if (p + 1 == n || ks[p + 1] != key) out.final_cts[c][key & 0xFFFF] = (u64)rank + 1;
    }
}
__global__ __launch_bounds__(TPB) void k_counter_clear(CounterOut out, CounterPlan P, size_t N) {
    const size_t per = (N + 65536) / 2, total = per * P.nchunks;   // 16-byte units per chunk: read_ts (N entries) then final_cts (2^16)
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int c = P.chunk[i / per];
        const size_t j = (i % per) * 2;
        u64* dst = j < N ? out.read_ts[c] + j : out.final_cts[c] + (j - N);
        *reinterpret_cast<ulonglong2*>(dst) = make_ulonglong2(0, 0);
    }
}
void lasso_counters_all(hipStream_t st, const LassoDev& L, unsigned chunk_mask, const u64* dims, const CounterOut& out, void* temp, size_t temp_bytes,
                        u32* keys, u32* keys_sorted, u32* vals, u32* vals_sorted, u32* starts) {
    const size_t N = (size_t)1 << L.nu;
    const CounterPlan P = counter_plan(L, chunk_mask);
    // rows outside a chunk's counted segments and addresses never touched keep counter 0: one clearing launch for all chunks
    // (eight hipMemsetAsync calls cost 5-14 us each on the stream and, as nodes of a captured graph, ~60 us each)
    if (P.nchunks > 0) k_counter_clear<<<grid_for((size_t)P.nchunks * (N + 65536) / 2), TPB, 0, st>>>(out, P, N);
    const size_t total = P.off[P.nchunks];
    if (total == 0) return;
    const int grid = grid_for(total);
    // stable sort on (chunk, address) keeps the rows of one address of one chunk in row order
    cs_sort_pairs(st, temp, temp_bytes, keys, keys_sorted, vals, vals_sorted, total, CsGenAll{L, P, dims});
    k_counter_starts<<<grid, TPB, 0, st>>>(keys, total, starts);
    k_counter_ranks_all<<<grid, TPB, 0, st>>>(keys, vals, total, starts, out);
}

__global__ __launch_bounds__(TPB) void k_lasso_claim(LassoDev L, const E2* __restrict__ eq, const u64* __restrict__ e_polys, EpRows R,
                                                     E2* __restrict__ partials) {
    __shared__ E2 sm[TPB / 64];
    const size_t N = (size_t)1 << L.nu;
    E2 acc = e2_zero();
    for (size_t k = (size_t)blockIdx.x * TPB + threadIdx.x; k < L.rows; k += (size_t)gridDim.x * TPB) {
        int l = L.seg_lookup[k >> L.seg_shift];
        u64 comb = 0;  // combine_lookups (range.rs:184-195): sum_i M^i * operand_i
        for (int i = 0; i < L.lookup_nmems[l]; i++) {
            const int row = R.row[L.lookup_mems[l][i]];
            if (row >= 0) comb = gl_add(comb, gl_mul(L.mpow[i], e_polys[(size_t)row * N + k]));
        }
        E2 e = eq[k];
        acc = e2_add(acc, e2_mul_f(e, comb));
    }
    E2 s = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
// the same with the E values recomputed from the node input (no E tables): own = bit mask of the memories whose terms this rank sums
__global__ __launch_bounds__(TPB) void k_lasso_claim_in(LassoDev L, const E2* __restrict__ eq, const u64* __restrict__ input, u32 own,
                                                        E2* __restrict__ partials) {
    __shared__ E2 sm[TPB / 64];
    E2 acc = e2_zero();
    for (size_t k = (size_t)blockIdx.x * TPB + threadIdx.x; k < L.rows; k += (size_t)gridDim.x * TPB) {
        const int l = L.seg_lookup[k >> L.seg_shift];
        const u64 v = input[k] & L.lookup_mask[l];
        u64 comb = 0;  // combine_lookups (range.rs:184-195): sum_i M^i * operand_i
        for (int i = 0; i < L.lookup_nmems[l]; i++) {
            const int m = L.lookup_mems[l][i];
            if (!((own >> m) & 1) || !((L.lookup_uses[l] >> m) & 1)) continue;
            const u32 a = (u32)(v >> (16 * L.mem_dim[m])) & 0xFFFF;
            if (a < L.mem_cutoff[m]) comb = gl_add(comb, gl_mul_small(L.mpow[i], a));
        }
        acc = e2_add(acc, e2_mul_f(eq[k], comb));
    }
    E2 s = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
int lasso_claim_in(hipStream_t st, const LassoDev& L, const E2* eq, const u64* input, u32 own, E2* partials) {
    int grid = grid_for(L.rows);
    k_lasso_claim_in<<<grid, TPB, 0, st>>>(L, eq, input, own, partials);
    return grid;
}
int lasso_claim(hipStream_t st, const LassoDev& L, const E2* eq, const u64* e_polys, const EpRows& rows, E2* partials) {
    int grid = grid_for(L.rows);
    k_lasso_claim<<<grid, TPB, 0, st>>>(L, eq, e_polys, rows, partials);
    return grid;
}

// read / write multiset hashes of one memory and, in the same pass, the first product-tree level of both tables
// (row j and row j + n/2 are hashed by the same thread), so the tree never re-reads the 2^nu-row hash tables
// Read / write multiset hashes of up to HASH_RW_MAX memories that share one chunk (same dim and read_ts columns):
// h(a,v,t) = a + v*gamma + t*gamma^2 - tau (prover.rs:44); the address/counter part is computed once per row and
// reused for every memory. Thread j handles rows 2j, 2j+1 and their partners in the upper half (16-byte accesses), and
// also emits the first product-tree level rd1[j] = rd[j] rd[j + n/2] (Layer::bottom + Layer::up, prover.rs:310-354).
__global__ __launch_bounds__(TPB) void k_hash_rw(size_t n, const u64* __restrict__ dim, const u64* __restrict__ ts, HashRwArgs args, int nmem,
                                                 u64 gamma, u64 gamma2, u64 tau) {
    const size_t h = n >> 1;
    typedef ulonglong2 V2;
    for (size_t j = ((size_t)blockIdx.x * TPB + threadIdx.x) * 2; j < h; j += (size_t)gridDim.x * TPB * 2) {
        const V2 dl = *reinterpret_cast<const V2*>(dim + j), dh = *reinterpret_cast<const V2*>(dim + j + h);
        const V2 tl = *reinterpret_cast<const V2*>(ts + j), th = *reinterpret_cast<const V2*>(ts + j + h);
        const u64 c0 = gl_sub(gl_add(dl.x, gl_mul(tl.x, gamma2)), tau), c1 = gl_sub(gl_add(dl.y, gl_mul(tl.y, gamma2)), tau);
        const u64 c2 = gl_sub(gl_add(dh.x, gl_mul(th.x, gamma2)), tau), c3 = gl_sub(gl_add(dh.y, gl_mul(th.y, gamma2)), tau);
        for (int m = 0; m < nmem; m++) {
            const u64* __restrict__ ep = args.ep[m];
            const V2 el = *reinterpret_cast<const V2*>(ep + j), eh = *reinterpret_cast<const V2*>(ep + j + h);
            const u64 a0 = gl_add(c0, gl_mul(el.x, gamma)), a1 = gl_add(c1, gl_mul(el.y, gamma));
            const u64 a2 = gl_add(c2, gl_mul(eh.x, gamma)), a3 = gl_add(c3, gl_mul(eh.y, gamma));
            const u64 b0 = gl_add(a0, gamma2), b1 = gl_add(a1, gamma2), b2 = gl_add(a2, gamma2), b3 = gl_add(a3, gamma2);  // write hash: t + 1
            *reinterpret_cast<V2*>(args.rd[m] + j) = make_ulonglong2(a0, a1);
            *reinterpret_cast<V2*>(args.rd[m] + j + h) = make_ulonglong2(a2, a3);
            *reinterpret_cast<V2*>(args.wr[m] + j) = make_ulonglong2(b0, b1);
            *reinterpret_cast<V2*>(args.wr[m] + j + h) = make_ulonglong2(b2, b3);
            if (args.rd1[m]) {
                *reinterpret_cast<V2*>(args.rd1[m] + j) = make_ulonglong2(gl_mul(a0, a2), gl_mul(a1, a3));
                *reinterpret_cast<V2*>(args.wr1[m] + j) = make_ulonglong2(gl_mul(b0, b2), gl_mul(b1, b3));
            }
        }
    }
}
void lasso_hash_rw(hipStream_t st, size_t n, const u64* dim, const u64* read_ts, const HashRwArgs& args, int nmem, u64 gamma, u64 tau) {
    k_hash_rw<<<grid_for(n >> 2) * 4, TPB, 0, st>>>(n, dim, read_ts, args, nmem, gamma, gl_mul(gamma, gamma), tau);
}
__global__ __launch_bounds__(TPB) void k_hash_if(HashIfArgs args, u64 gamma, u64 gamma2, u64 tau, u64* __restrict__ H2, int G) {
    const int i = blockIdx.y;  // memory
    u32 a = blockIdx.x * TPB + threadIdx.x;
    if (a >= 65536) return;
    u64 tv = a < args.cutoff[i] ? (u64)a : 0;
    u64 h0 = gl_sub(gl_add((u64)a, gl_mul(tv, gamma)), tau);
    if (args.row_init[i] >= 0) H2[(size_t)args.row_init[i] * 65536 + a] = h0;
    if (args.row_fin[i] >= 0) H2[(size_t)args.row_fin[i] * 65536 + a] = gl_add(h0, gl_mul(gl_from_u64(args.fc[i][a]), gamma2));
}
void lasso_hash_if(hipStream_t st, const HashIfArgs& args, int G, u64 gamma, u64 tau, u64* H2) {
    k_hash_if<<<dim3(65536 / TPB, G), TPB, 0, st>>>(args, gamma, gl_mul(gamma, gamma), tau, H2, G);
}

__global__ __launch_bounds__(TPB) void k_prod_level(const u64* __restrict__ in, size_t in_len, u64* __restrict__ out) {
    const size_t h = in_len >> 1;
    const u64* src = in + (size_t)blockIdx.y * in_len;
    u64* dst = out + (size_t)blockIdx.y * h;
    if (h & 1) {
        for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < h; i += (size_t)gridDim.x * TPB) dst[i] = gl_mul(src[i], src[i + h]);
        return;
    }
    // two outputs per thread: 16-byte loads and stores
    for (size_t i = ((size_t)blockIdx.x * TPB + threadIdx.x) * 2; i < h; i += (size_t)gridDim.x * TPB * 2) {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(src + i), b = *reinterpret_cast<const ulonglong2*>(src + i + h);
        *reinterpret_cast<ulonglong2*>(dst + i) = make_ulonglong2(gl_mul(a.x, b.x), gl_mul(a.y, b.y));
    }
}
void prod_level(hipStream_t st, const u64* in, size_t in_len, u64* out, int nb) {
    size_t h = in_len >> 1;
    dim3 grid((unsigned)std::min<size_t>((h / 2 + TPB - 1) / TPB + 1, 256), (unsigned)nb);
    k_prod_level<<<grid, TPB, 0, st>>>(in, in_len, out);
}
// three levels per launch: thread j of a row holds the eight entries j + m * in_len/8 and writes 4 + 2 + 1 products (the mid-size
// levels are launch-bound: six of them between the levels the sum-check first rounds emit and the single-workgroup tail)
__global__ __launch_bounds__(TPB) void k_prod_level3(const u64* __restrict__ in, size_t in_len, u64* __restrict__ o1, u64* __restrict__ o2, u64* __restrict__ o3) {
    const size_t e = in_len >> 3;
    const u64* src = in + (size_t)blockIdx.y * in_len;
    u64* d1 = o1 + (size_t)blockIdx.y * (4 * e);
    u64* d2 = o2 + (size_t)blockIdx.y * (2 * e);
    u64* d3 = o3 + (size_t)blockIdx.y * e;
    for (size_t j = (size_t)blockIdx.x * TPB + threadIdx.x; j < e; j += (size_t)gridDim.x * TPB) {
        u64 x[8];
#pragma unroll
        for (int m = 0; m < 8; m++) x[m] = src[j + m * e];
        u64 y[4];
#pragma unroll
        for (int m = 0; m < 4; m++) { y[m] = gl_mul(x[m], x[m + 4]); d1[j + m * e] = y[m]; }
        const u64 z0 = gl_mul(y[0], y[2]), z1 = gl_mul(y[1], y[3]);
        d2[j] = z0; d2[j + e] = z1;
        d3[j] = gl_mul(z0, z1);
    }
}
void prod_level3(hipStream_t st, const u64* in, size_t in_len, u64* o1, u64* o2, u64* o3, int nb) {
    const size_t e = in_len >> 3;
    dim3 grid((unsigned)std::min<size_t>((e + TPB - 1) / TPB, 256), (unsigned)nb);
    k_prod_level3<<<grid, TPB, 0, st>>>(in, in_len, o1, o2, o3);
}
// the small levels of the tree in one launch: workgroup b holds row b of a level of at most PROD_TAIL_LEN entries in LDS and
// writes every level above it (levels[k] = row-major nb x (in_len >> (k+1)))
__global__ __launch_bounds__(TPB) void k_prod_tail(const u64* __restrict__ in, int in_len, ProdTailOut outs, int nlevels) {
    __shared__ u64 row[PROD_TAIL_LEN];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < in_len; i += TPB) row[i] = in[(size_t)b * in_len + i];
    __syncthreads();
    int len = in_len;
    for (int k = 0; k < nlevels; k++) {
        const int h = len >> 1;
        u64 v[PROD_TAIL_LEN / 2 / TPB + 1];
        int c = 0;
        for (int i = threadIdx.x; i < h; i += TPB) v[c++] = gl_mul(row[i], row[i + h]);
        __syncthreads();
        c = 0;
        for (int i = threadIdx.x; i < h; i += TPB) { row[i] = v[c]; outs.p[k][(size_t)b * h + i] = v[c]; c++; }
        __syncthreads();
        len = h;
    }
}
void prod_tail(hipStream_t st, const u64* in, int in_len, const ProdTailOut& outs, int nlevels, int nb) {
    k_prod_tail<<<nb, TPB, 0, st>>>(in, in_len, outs, nlevels);
}
__global__ void k_gp_top(const u64* __restrict__ top, int nb, E2* __restrict__ roots, E2* __restrict__ evals) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    u64 l = top[2 * b], r = top[2 * b + 1];
    roots[b] = e2(gl_mul(l, r), 0);
    evals[2 * b] = e2(l, 0);
    evals[2 * b + 1] = e2(r, 0);
}
void gp_top(hipStream_t st, const u64* top, int nb, E2* roots, E2* evals) {
    k_gp_top<<<(nb + 63) / 64, 64, 0, st>>>(top, nb, roots, evals);
}

// out[t] = sum_j eq[j] * tabs[t][j] for up to DOT_MAX base-field tables sharing one eq table, in ONE launch: grid.y = group of 8
// tables (a thread keeps 8 accumulators; eq is re-read per group, from L2 / MALL after the first). The reduction launch has one
// workgroup per table and writes result t to out[slot[t]].
// VIRT: tables with tabs.t[t] == nullptr are E tables that are not materialised - E_m[j] is recomputed from the node input (DotVirt:
// memory tabs.emem[t]); a group's eight tables then cost one 8-byte load per entry instead of eight.
template <bool VIRT>
__global__ __launch_bounds__(TPB) void k_dot_eq(const E2* __restrict__ eq, DotTabs tabs, int ntab_all, size_t n, E2* __restrict__ partials, DotVirt V) {
    __shared__ E2 sm[TPB / 64];
    const int t0 = blockIdx.y * 8;
    const int ntab = min(8, ntab_all - t0);
    E2 acc[8];
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] = e2_zero();
    if constexpr (VIRT) {
        for (size_t j = (size_t)blockIdx.x * TPB + threadIdx.x; j < n; j += (size_t)gridDim.x * TPB) {
            const E2 e = eq[j];
            u64 v = 0, uses = 0;
            if (j < V.rows) { const int l = V.seg_lookup[j >> V.seg_shift]; v = V.input[j] & V.lookup_mask[l]; uses = V.lookup_uses[l]; }
#pragma unroll
            for (int t = 0; t < 8; t++)
                if (t < ntab) {
                    const u64* tab = tabs.t[t0 + t];
                    if (tab) acc[t] = e2_add(acc[t], e2_mul_f(e, tab[j]));
                    else {
                        const int m = tabs.emem[t0 + t];
                        const u32 a = (u32)(v >> (16 * V.mem_dim[m])) & 0xFFFF;
                        if (((uses >> m) & 1) && a && a < V.mem_cutoff[m]) acc[t] = e2_add(acc[t], e2(gl_mul_small(e.c0, a), gl_mul_small(e.c1, a)));
                    }
                }
        }
    } else
    if ((n & 1) == 0) {  // two entries per thread: 16-byte table loads
        for (size_t j = ((size_t)blockIdx.x * TPB + threadIdx.x) * 2; j < n; j += (size_t)gridDim.x * TPB * 2) {
            const E2 e0 = eq[j], e1 = eq[j + 1];
#pragma unroll
            for (int t = 0; t < 8; t++)
                if (t < ntab) {
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(tabs.t[t0 + t] + j);
                    acc[t] = e2_add(acc[t], e2_add(e2_mul_f(e0, v.x), e2_mul_f(e1, v.y)));
                }
        }
    } else {
        for (size_t j = (size_t)blockIdx.x * TPB + threadIdx.x; j < n; j += (size_t)gridDim.x * TPB) {
            E2 e = eq[j];
#pragma unroll
            for (int t = 0; t < 8; t++)
                if (t < ntab) acc[t] = e2_add(acc[t], e2_mul_f(e, tabs.t[t0 + t][j]));
        }
    }
#pragma unroll
    for (int t = 0; t < 8; t++)
        if (t < ntab) {
            E2 s = block_sum(acc[t], sm);
            if (threadIdx.x == 0) partials[(size_t)blockIdx.x * ntab_all + t0 + t] = s;
        }
}
__global__ __launch_bounds__(TPB) void k_dot_reduce(const E2* __restrict__ partials, int nblocks, int nv, DotTabs tabs, E2* __restrict__ out) {
    __shared__ E2 sm[TPB / 64];
    const int v = blockIdx.x;
    E2 a = e2_zero();
    for (int b = threadIdx.x; b < nblocks; b += TPB) a = e2_add(a, partials[(size_t)b * nv + v]);
    a = block_sum(a, sm);
    if (threadIdx.x == 0) out[tabs.slot[v]] = a;
}
void dot_eq_many(hipStream_t st, const E2* eq, const DotTabs& tabs, int ntab, size_t n, E2* partials, E2* out, const DotVirt* virt) {
    if (ntab <= 0) return;
    if (ntab > DOT_MAX) throw std::runtime_error("dot_eq_many: too many tables");
    // (measured slower here: column accumulators - 8 independent 8-byte streams per thread need the occupancy more -
    // and the last-arriving-workgroup reduction - 8 values x 1024 partials)
    const int gx = grid_for((n + 1) / 2);
    DotVirt V;
    memset(&V, 0, sizeof(V));
    if (virt) { V = *virt; k_dot_eq<true><<<dim3(gx, (ntab + 7) / 8), TPB, 0, st>>>(eq, tabs, ntab, n, partials, V); }
    else k_dot_eq<false><<<dim3(gx, (ntab + 7) / 8), TPB, 0, st>>>(eq, tabs, ntab, n, partials, V);
    k_dot_reduce<<<ntab, TPB, 0, st>>>(partials, gx, ntab, tabs, out);
}
struct OpenPlan { int nmat, nvirt; short mat[DOT_MAX], virt[DOT_MAX]; };
constexpr int OPEN_ACT = 8;
__global__ __launch_bounds__(TPB) void k_open_x(const E2* __restrict__ eq, DotTabs tabs, OpenPlan P, int ntab_all, size_t n, size_t chunk,
                                                E2* __restrict__ partials, DotVirt V) {
    __shared__ E2 sm[TPB / 64];
    __shared__ int s_act[OPEN_ACT];
    __shared__ int s_nact, s_l;
    const size_t row0 = (size_t)blockIdx.x * chunk, row1 = row0 + chunk < n ? row0 + chunk : n;
    const int ngm = (P.nmat + 7) / 8;
    E2 acc[8];
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] = e2_zero();
    if ((int)blockIdx.y < ngm) {   // a group of materialised tables
        const int t0 = blockIdx.y * 8;
        const int nt = P.nmat - t0 < 8 ? P.nmat - t0 : 8;
        for (size_t j = row0 + threadIdx.x; j < row1; j += TPB) {
            const E2 e = eq[j];
#pragma unroll
            for (int t = 0; t < 8; t++)
                if (t < nt) acc[t] = e2_add(acc[t], e2_mul_f(e, tabs.t[P.mat[t0 + t]][j]));
        }
#pragma unroll
        for (int t = 0; t < 8; t++)
            if (t < nt) {
                const E2 s = block_sum(acc[t], sm);
                if (threadIdx.x == 0) partials[(size_t)blockIdx.x * ntab_all + P.mat[t0 + t]] = s;
            }
        return;
    }
    // the recomputed E tables: the rows of this workgroup belong to one lookup
    if (threadIdx.x == 0) {
        int nact = 0, l = 0;
        if (row0 < V.rows) {
            l = V.seg_lookup[row0 >> V.seg_shift];
            const u64 uses = V.lookup_uses[l];
            for (int m = 0; m < 32; m++)
                if (((uses >> m) & 1) && nact < OPEN_ACT) s_act[nact++] = m;
        }
        s_nact = nact; s_l = l;
    }
    __syncthreads();
    const int nact = s_nact;
    const u64 mask = V.lookup_mask[s_l];
    int sh[OPEN_ACT];
    u32 cut[OPEN_ACT];
#pragma unroll
    for (int k = 0; k < OPEN_ACT; k++) { const int m = k < nact ? s_act[k] : 0; sh[k] = 16 * V.mem_dim[m]; cut[k] = k < nact ? V.mem_cutoff[m] : 0u; }
    const size_t rend = row1 < V.rows ? row1 : V.rows;
    for (size_t j = row0 + threadIdx.x; j < rend; j += TPB) {
        const E2 e = eq[j];
        const u64 v = V.input[j] & mask;
#pragma unroll
        for (int k = 0; k < OPEN_ACT; k++) {
            const u32 a = (u32)(v >> sh[k]) & 0xFFFF;
            if (a && a < cut[k]) acc[k] = e2_add(acc[k], e2(gl_mul_small(e.c0, a), gl_mul_small(e.c1, a)));
        }
    }
    E2 sum[OPEN_ACT];
#pragma unroll
    for (int k = 0; k < OPEN_ACT; k++) sum[k] = k < nact ? block_sum(acc[k], sm) : e2_zero();   // (nact is uniform: the barriers inside match)
    if (threadIdx.x == 0)
        for (int i = 0; i < P.nvirt; i++) {
            const int vi = P.virt[i], m = tabs.emem[vi];
            E2 val = e2_zero();
#pragma unroll
            for (int k = 0; k < OPEN_ACT; k++) if (k < nact && s_act[k] == m) val = sum[k];
            partials[(size_t)blockIdx.x * ntab_all + vi] = val;
        }
}
bool open_x(hipStream_t st, const E2* eq, const DotTabs& tabs, int ntab, size_t n, E2* partials, E2* out, const DotVirt& virt) {
    if (ntab <= 0 || ntab > DOT_MAX) return false;   // (false: the caller takes dot_eq_many's groups of eight)
    const int gx = grid_for((n + 1) / 2);
    if (n % (size_t)gx) return false;
    const size_t chunk = n / (size_t)gx;
    if ((chunk & (chunk - 1)) || chunk > ((size_t)1 << virt.seg_shift)) return false;
    for (int l = 0; l < 32; l++) if (__builtin_popcountll(virt.lookup_uses[l]) > OPEN_ACT) return false;
    OpenPlan P;
    memset(&P, 0, sizeof(P));
    for (int t = 0; t < ntab; t++) {
        if (tabs.t[t]) P.mat[P.nmat++] = (short)t; else P.virt[P.nvirt++] = (short)t;
    }
    const int ngm = (P.nmat + 7) / 8;
    k_open_x<<<dim3(gx, ngm + (P.nvirt ? 1 : 0)), TPB, 0, st>>>(eq, tabs, P, ntab, n, chunk, partials, virt);
    k_dot_reduce<<<ntab, TPB, 0, st>>>(partials, gx, ntab, tabs, out);
    return true;
}
void dot_eq(hipStream_t st, const E2* eq, const u64* const tabs[8], int ntab, size_t n, E2* partials, E2* out) {
    DotTabs d;
    memset(&d, 0, sizeof(d));
    for (int t = 0; t < ntab; t++) { d.t[t] = tabs[t]; d.slot[t] = t; }
    dot_eq_many(st, eq, d, ntab, n, partials, out);
}

// ------------------------------------------------------------------------------------------------
// Vanilla / FFT node bookkeeping
__global__ __launch_bounds__(TPB) void k_gather_B(CsrMul m, const E2* __restrict__ eqc, const E2* __restrict__ eqx,
                                                  const E2* __restrict__ u, int log2_S, int log2_G, int log2_R, E2* __restrict__ B) {
    const size_t total = (size_t)1 << (log2_S + log2_R);
    const size_t smask = ((size_t)1 << log2_S) - 1;
    for (size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x; idx < total; idx += (size_t)gridDim.x * TPB) {
        size_t y = idx & smask, rep = idx >> log2_S;
        E2 acc = e2_zero();
        for (u32 e = m.ptr[y]; e < m.ptr[y + 1]; e++) {
            E2 q = eqc[(rep << log2_G) + m.gate[e]];
            u64 c = m.coef[e];
            if (c != 1) q = e2_mul_f(q, c);
            q = e2_mul(q, eqx[(rep << log2_S) + m.other_j[e]]);
            acc = e2_add(acc, u ? e2_mul(q, u[m.other_in[e]]) : q);   // (u null: the factor is applied by the caller)
        }
        store_e2(B + idx, acc);
    }
}
void vanilla_gather_B(hipStream_t st, const CsrMul& mulR, const E2* eqc, const E2* eqx, const E2* u, int log2_S, int log2_G, int log2_R, E2* B) {
    size_t total = (size_t)1 << (log2_S + log2_R);
    k_gather_B<<<grid_for(total) * 4, TPB, 0, st>>>(mulR, eqc, eqx, u, log2_S, log2_G, log2_R, B);
}
// the same for a batch of (node, right input) jobs, grid.y = job
__global__ __launch_bounds__(TPB) void k_gather_B_jobs(const GatherBJob* __restrict__ jobs) {
    const GatherBJob& J = jobs[blockIdx.y];
    const CsrMul& m = J.m;
    const size_t total = (size_t)1 << (J.log2_S + J.log2_R);
    const size_t smask = ((size_t)1 << J.log2_S) - 1;
    for (size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x; idx < total; idx += (size_t)gridDim.x * TPB) {
        size_t y = idx & smask, rep = idx >> J.log2_S;
        E2 acc = e2_zero();
        for (u32 e = m.ptr[y]; e < m.ptr[y + 1]; e++) {
            E2 q = J.eqc[(rep << J.log2_G) + m.gate[e]];
            u64 c = m.coef[e];
            if (c != 1) q = e2_mul_f(q, c);
            q = e2_mul(q, J.eqx[(rep << J.log2_S) + m.other_j[e]]);
            acc = e2_add(acc, J.u ? e2_mul(q, J.u[m.other_in[e]]) : q);   // (u null: the factor is applied by the host's transcript replay)
        }
        store_e2(J.B + idx, acc);
    }
}
void gather_B_jobs(hipStream_t st, const GatherBJob* jobs, int njobs, size_t max_total) {
    k_gather_B_jobs<<<dim3(grid_for(max_total) * 4, njobs), TPB, 0, st>>>(jobs);
}
__global__ __launch_bounds__(TPB) void k_const_sum(const u32* __restrict__ gate, const u64* __restrict__ coef, size_t nterms,
                                                   const E2* __restrict__ eqc, int log2_G, int log2_R, E2* __restrict__ partials) {
    __shared__ E2 sm[TPB / 64];
    const size_t total = nterms << log2_R;
    E2 acc = e2_zero();
    for (size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x; idx < total; idx += (size_t)gridDim.x * TPB) {
        size_t t = idx % nterms, rep = idx / nterms;
        acc = e2_add(acc, e2_mul_f(eqc[(rep << log2_G) + gate[t]], coef[t]));
    }
    E2 s = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
int vanilla_const_sum(hipStream_t st, const u32* gate, const u64* coef, size_t nterms, const E2* eqc, int log2_G, int log2_R, E2* partials) {
    int grid = grid_for(nterms << log2_R);
    k_const_sum<<<grid, TPB, 0, st>>>(gate, coef, nterms, eqc, log2_G, log2_R, partials);
    return grid;
}

// zkCNN DFT-row tables of every FFT node (grid.y = node): F(x) = scale * sum_a alpha_a prod_b f_{a,b}(x),
// f_{a,b}(x) = 1 + r_{a,b} (w^(2^b x) - 1). The factor b only depends on x mod 2^(L-b), so the factors b >= FFT_SPLIT
// are a table over x mod 2^(L - FFT_SPLIT) (k_fft_tab, 2^(L-4) entries per claim) and every output needs just the 4
// low-b factors and one table lookup: 5 instead of 16 Ext2 multiplications per output at L = 16.
constexpr int FFT_SPLIT = 4;
__global__ __launch_bounds__(TPB) void k_fft_tab(const FftJob* __restrict__ jobs, const E2* __restrict__ chal, E2* __restrict__ tab, size_t tab_stride,
                                                 int max_claims) {
    const FftJob& J = jobs[blockIdx.y / max_claims];
    const int a = blockIdx.y % max_claims;
    const int L = J.L;
    if (a >= J.cs.n || L <= FFT_SPLIT) return;
    const size_t N = (size_t)1 << L, n = N >> FFT_SPLIT;
    const u64* __restrict__ W = J.W;
    const E2* r = chal + J.cs.point_off[a];
    E2* out = tab + (size_t)blockIdx.y * tab_stride;
    for (size_t y = (size_t)blockIdx.x * TPB + threadIdx.x; y < n; y += (size_t)gridDim.x * TPB) {
        E2 p = J.cs.unit_alpha ? e2(J.scale, 0) : e2_mul_f(chal[J.cs.alpha_off + a], J.scale);
        for (int b = FFT_SPLIT; b < L; b++) {
            u64 wx = W[(y << b) & (N - 1)];
            p = e2_mul(p, e2_add_f(e2_mul_f(r[b], gl_sub(wx, 1)), 1));
        }
        store_e2(out + y, p);
    }
}
__global__ __launch_bounds__(TPB) void k_fft_jobs(const FftJob* __restrict__ jobs, const E2* __restrict__ chal, const E2* __restrict__ tab,
                                                  size_t tab_stride, int max_claims) {
    const FftJob& J = jobs[blockIdx.y];
    const int L = J.L;
    const size_t N = (size_t)1 << L;
    const u64* __restrict__ W = J.W;
    const bool split = L > FFT_SPLIT;
    const int nb = split ? FFT_SPLIT : L;
    for (size_t x = (size_t)blockIdx.x * TPB + threadIdx.x; x < N; x += (size_t)gridDim.x * TPB) {
        E2 acc = e2_zero();
        for (int a = 0; a < J.cs.n; a++) {
            const E2* r = chal + J.cs.point_off[a];
            E2 p = split ? tab[((size_t)blockIdx.y * max_claims + a) * tab_stride + (x & ((N >> FFT_SPLIT) - 1))]
                         : (J.cs.unit_alpha ? e2(J.scale, 0) : e2_mul_f(chal[J.cs.alpha_off + a], J.scale));
            for (int b = 0; b < nb; b++) {
                u64 wx = W[(x << b) & (N - 1)];
                E2 f = e2_add_f(e2_mul_f(r[b], gl_sub(wx, 1)), 1);  // 1 - r_b + r_b w^(2^b x)
                p = e2_mul(p, f);
            }
            acc = e2_add(acc, p);
        }
        store_e2(J.out + x, acc);
    }
}
void fft_jobs(hipStream_t st, const FftJob* jobs, int njobs, int max_L, int max_claims, const E2* chal, E2* tab) {
    size_t N = (size_t)1 << max_L;
    const size_t tab_stride = max_L > FFT_SPLIT ? N >> FFT_SPLIT : 1;
    if (max_L > FFT_SPLIT)
        k_fft_tab<<<dim3((unsigned)std::min<size_t>((tab_stride + TPB - 1) / TPB, 64), njobs * max_claims), TPB, 0, st>>>(jobs, chal, tab, tab_stride, max_claims);
    k_fft_jobs<<<dim3((unsigned)std::min<size_t>((N + TPB - 1) / TPB, 1024), njobs), TPB, 0, st>>>(jobs, chal, tab, tab_stride, max_claims);
}
// Libra bookkeeping tables of many (node, input) pairs in one launch (grid.y = job)
__global__ __launch_bounds__(TPB) void k_gather_jobs(const GatherJob* __restrict__ jobs) {
    const GatherJob& J = jobs[blockIdx.y];
    const GatherT& g = J.g;
    const int log2_S = J.log2_S, log2_G = J.log2_G;
    const size_t total = (size_t)1 << (log2_S + J.log2_R);
    const size_t smask = ((size_t)1 << log2_S) - 1;
    const E2* __restrict__ eqc = J.eqc;
    for (size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x; idx < total; idx += (size_t)gridDim.x * TPB) {
        size_t x = idx & smask, rep = idx >> log2_S;
        const E2* eq_rep = eqc + (rep << log2_G);
        E2 acc = e2_zero();
        if (g.lin.ptr) {
            for (u32 e = g.lin.ptr[x]; e < g.lin.ptr[x + 1]; e++) {
                E2 q = eq_rep[g.lin.gate[e]];
                u64 c = g.lin.coef[e];
                acc = e2_add(acc, c == 1 ? q : e2_mul_f(q, c));
            }
        }
        if (g.mul.ptr) {
            for (u32 e = g.mul.ptr[x]; e < g.mul.ptr[x + 1]; e++) {
                E2 q = eq_rep[g.mul.gate[e]];
                u64 other = g.in_vals[g.mul.other_in[e]][(rep << log2_S) + g.mul.other_j[e]];
                u64 c = g.mul.coef[e];
                acc = e2_add(acc, e2_mul_f(q, c == 1 ? other : gl_mul(c, other)));
            }
        }
        store_e2(J.T + idx, acc);
    }
}
// run-length form (see kernels.hpp): 1-D grid, job q owns workgroups [blk0_q, blk0_(q+1)), 4 outputs per thread
constexpr int GSEG_PER_THREAD = 4;
constexpr int GSEG_MAX = 64;
__global__ __launch_bounds__(TPB) void k_gather_seg_jobs(const GatherSegJob* __restrict__ jobs, int njobs) {
    int jl = 0, jh = njobs - 1;
    while (jl < jh) {
        int mid = (jl + jh + 1) >> 1;
        if (jobs[mid].blk0 <= (int)blockIdx.x) jl = mid; else jh = mid - 1;
    }
    const GatherSegJob& J = jobs[jl];
    __shared__ GatherSeg sg[GSEG_MAX];   // the segment list, read once per workgroup
    const int nseg = J.nseg;
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(J.segs);
        for (unsigned k = threadIdx.x; k < nseg * (sizeof(GatherSeg) / 4); k += TPB) reinterpret_cast<unsigned*>(sg)[k] = src[k];
    }
    __syncthreads();
    const int log2_S = J.log2_S, log2_G = J.log2_G;
    const size_t total = (size_t)1 << (log2_S + J.log2_R);
    const size_t smask = ((size_t)1 << log2_S) - 1;
    const E2* eqc = J.eqc;
    const size_t base = (size_t)((int)blockIdx.x - J.blk0) * (TPB * GSEG_PER_THREAD) + threadIdx.x;
    size_t x[GSEG_PER_THREAD], rep[GSEG_PER_THREAD];
    E2 acc[GSEG_PER_THREAD];
#pragma unroll
    for (int k = 0; k < GSEG_PER_THREAD; k++) {
        const size_t idx = base + (size_t)k * TPB;
        x[k] = idx & smask; rep[k] = idx >> log2_S;
        if (idx >= total) x[k] = ~(size_t)0;   // in no segment
        acc[k] = e2_zero();
    }
    for (int s = 0; s < nseg; s++) {
        const GatherSeg g = sg[s];
        bool in[GSEG_PER_THREAD];
        E2 q[GSEG_PER_THREAD];
        u64 other[GSEG_PER_THREAD];
        // every load of the step first (positions outside the segment read entry 0 of their table and are dropped)
#pragma unroll
        for (int k = 0; k < GSEG_PER_THREAD; k++) {
            in[k] = x[k] >= g.lo && x[k] < g.hi;
            q[k] = gload_e2(eqc + (in[k] ? (rep[k] << log2_G) + (size_t)((long long)x[k] + g.goff) : 0));
        }
        if (g.other_in >= 0) {
            const u64* ov = J.in_vals[g.other_in];
#pragma unroll
            for (int k = 0; k < GSEG_PER_THREAD; k++) other[k] = gload_u64(ov + (in[k] ? (rep[k] << log2_S) + (size_t)((long long)x[k] + g.joff) : 0));
        }
#pragma unroll
        for (int k = 0; k < GSEG_PER_THREAD; k++) {
            u64 c = g.coef;
            E2 v = q[k];
            if (g.other_in >= 0) v = e2_mul_f(v, c == 1 ? other[k] : gl_mul(c, other[k]));
            else if (c != 1) v = e2_mul_f(v, c);
            if (in[k]) acc[k] = e2_add(acc[k], v);
        }
    }
#pragma unroll
    for (int k = 0; k < GSEG_PER_THREAD; k++) {
        const size_t idx = base + (size_t)k * TPB;
        if (idx < total) gstore_e2(J.T + idx, acc[k]);
    }
}
int gather_seg_plan(GatherSegJob* host_jobs, int njobs) {
    int blk = 0;
    for (int q = 0; q < njobs; q++) {
        const size_t total = (size_t)1 << (host_jobs[q].log2_S + host_jobs[q].log2_R);
        host_jobs[q].blk0 = blk;
        blk += (int)((total + TPB * GSEG_PER_THREAD - 1) / (TPB * GSEG_PER_THREAD));
    }
    return blk;
}
void gather_seg_jobs(hipStream_t st, const GatherSegJob* jobs, int njobs, int grid) {
    k_gather_seg_jobs<<<grid, TPB, 0, st>>>(jobs, njobs);
}
void gather_jobs(hipStream_t st, const GatherJob* jobs, int njobs, size_t max_total) {
    k_gather_jobs<<<dim3((unsigned)std::min<size_t>((max_total + TPB - 1) / TPB, 1024), njobs), TPB, 0, st>>>(jobs);
}
__global__ void k_powers_table(u64* __restrict__ W, u64 w, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 r = 1, b = w;
    for (size_t e = i; e; e >>= 1) { if (e & 1) r = gl_mul(r, b); b = gl_mul(b, b); }
    W[i] = r;
}
void powers_table(hipStream_t st, u64* W, u64 w, size_t n) {
    k_powers_table<<<(unsigned)((n + TPB - 1) / TPB), TPB, 0, st>>>(W, w, n);
}

// ------------------------------------------------------------------------------------------------
// Vanilla node evaluation: out[rep][g] = w0_g + sum c * in_i[rep][j] + sum c * in_i0[rep][j0] * in_i1[rep][j1]
__global__ __launch_bounds__(TPB) void k_gate_eval(EvalNode N) {
    const size_t total = (size_t)1 << (N.log2_G + N.log2_R);
    const size_t gmask = ((size_t)1 << N.log2_G) - 1;
    for (size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x; idx < total; idx += (size_t)gridDim.x * TPB) {
        const size_t g = idx & gmask, rep = idx >> N.log2_G;
        u64 acc = 0;
        if (g < N.num_gates) {
            if (N.w0) acc = N.w0[g];
            if (N.lptr) {
                for (u32 e = N.lptr[g]; e < N.lptr[g + 1]; e++) {
                    u64 x = N.in[N.lin_in[e]][(rep << N.log2_S) + N.lin_j[e]];
                    u64 c = N.lcoef[e];
                    acc = gl_add(acc, c == 1 ? x : gl_mul(c, x));
                }
            }
            if (N.mptr) {
                for (u32 e = N.mptr[g]; e < N.mptr[g + 1]; e++) {
                    u64 x = N.in[N.mi0[e]][(rep << N.log2_S) + N.mj0[e]];
                    u64 y = N.in[N.mi1[e]][(rep << N.log2_S) + N.mj1[e]];
                    u64 c = N.mcoef[e];
                    u64 pr = gl_mul(x, y);
                    acc = gl_add(acc, c == 1 ? pr : gl_mul(c, pr));
                }
            }
        }
        N.out[idx] = acc;
    }
}
void gate_eval(hipStream_t st, const EvalNode& n) {
    size_t total = (size_t)1 << (n.log2_G + n.log2_R);
    k_gate_eval<<<(unsigned)std::min<size_t>((total + TPB - 1) / TPB, 4096), TPB, 0, st>>>(n);
}

// ------------------------------------------------------------------------------------------------
// NTT: decimation-in-frequency stages in HBM, then bit reversal (+ scaling for the inverse)
__global__ __launch_bounds__(TPB) void k_ntt_stage(u64* __restrict__ data, int log2n, int s, size_t batch, const u64* __restrict__ W) {
    const size_t halfN = (size_t)1 << (log2n - 1);
    const size_t total = batch * halfN;
    const size_t h = (size_t)1 << s;
    for (size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x; idx < total; idx += (size_t)gridDim.x * TPB) {
        size_t bi = idx >> (log2n - 1), k = idx & (halfN - 1);
        size_t j = k & (h - 1), base = (k >> s) << (s + 1);
        u64* a = data + (bi << log2n) + base + j;
        u64 x = a[0], y = a[h];
        a[0] = gl_add(x, y);
        a[h] = gl_mul(gl_sub(x, y), W[j << (log2n - 1 - s)]);
    }
}
__global__ __launch_bounds__(TPB) void k_ntt_bitrev(u64* __restrict__ data, int log2n, size_t batch, u64 scale) {
    const size_t N = (size_t)1 << log2n;
    const size_t total = batch * N;
    for (size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x; idx < total; idx += (size_t)gridDim.x * TPB) {
        size_t bi = idx >> log2n, i = idx & (N - 1);
        size_t r = (size_t)(__brevll((unsigned long long)i) >> (64 - log2n));
        u64* a = data + (bi << log2n);
        if (r > i) {
            u64 x = a[i], y = a[r];
            if (scale != 1) { x = gl_mul(x, scale); y = gl_mul(y, scale); }
            a[i] = y; a[r] = x;
        } else if (r == i && scale != 1) {
            a[i] = gl_mul(a[i], scale);
        }
    }
}
// ---- four-step NTT, both steps LDS-resident -------------------------------------------------------------
// N = N1*N2 (N1 = 2^n1, N2 = 2^n2 <= 256). Input index i = i1*N2 + i2, output index k = k1 + N1*k2:
//   X[k1 + N1 k2] = sum_i2 w_N2^(i2 k2) * [ w^(i2 k1) * sum_i1 x[i1 N2 + i2] w_N1^(i1 k1) ]
// Step A: a workgroup takes 16 adjacent columns i2 (128-B coalesced row segments), runs the N1-point transforms
// down the columns in LDS (radix-2 decimation in frequency, bit reversal absorbed in the store index), applies
// the twiddle w^(i2 k1) and stores Y[k1*N2 + i2]. Step B: a workgroup takes 16 adjacent rows k1, transposes them
// into LDS (row stride 17: conflict-free), runs the N2-point transforms and stores 16 adjacent outputs per k2.
constexpr int NTT_TILE = 16;
__device__ __forceinline__ u32 brev_bits(u32 x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

template <int STRIDE>
__device__ __forceinline__ void lds_ntt_dif(u64* tile, int m, const u64* __restrict__ W, size_t wstep_log2) {
    // 2^m-point DIF along the slow index of tile[pos*STRIDE + lane16]; twiddle w_M^j = W[j << wstep_log2]
    const int M = 1 << m;
    for (int s = m - 1; s >= 0; s--) {
        const int h = 1 << s;
        for (int q = threadIdx.x; q < (M / 2) * NTT_TILE; q += blockDim.x) {
            const int c = q & (NTT_TILE - 1), p = q >> 4;
            const int j = p & (h - 1), a = ((p >> s) << (s + 1)) + j;
            u64 x = tile[a * STRIDE + c], y = tile[(a + h) * STRIDE + c];
            tile[a * STRIDE + c] = gl_add(x, y);
            u64 d = gl_sub(x, y);
            tile[(a + h) * STRIDE + c] = j ? gl_mul(d, W[((size_t)j << (m - 1 - s)) << wstep_log2]) : d;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_ntt4_cols(const u64* __restrict__ in, u64* __restrict__ out, int n, int n1,
                                                   const u64* __restrict__ W) {
    extern __shared__ u64 ntile[];
    const int n2 = n - n1;
    const size_t N = (size_t)1 << n, N2 = (size_t)1 << n2;
    const int N1 = 1 << n1;
    const size_t i2_0 = (size_t)blockIdx.x * NTT_TILE;
    const u64* x = in + (size_t)blockIdx.y * N;
    u64* y = out + (size_t)blockIdx.y * N;
    for (int idx = threadIdx.x; idx < N1 * NTT_TILE; idx += blockDim.x) {
        int i1 = idx >> 4, c = idx & (NTT_TILE - 1);
        ntile[idx] = x[(size_t)i1 * N2 + i2_0 + c];
    }
    __syncthreads();
    lds_ntt_dif<NTT_TILE>(ntile, n1, W, n2);  // w_N1 = w^(N2)
    for (int idx = threadIdx.x; idx < N1 * NTT_TILE; idx += blockDim.x) {
        int pos = idx >> 4, c = idx & (NTT_TILE - 1);
        u32 k1 = brev_bits((u32)pos, n1);
        size_t i2 = i2_0 + c;
        u64 v = ntile[idx];
        size_t e = (size_t)k1 * i2;  // < N
        y[(size_t)k1 * N2 + i2] = e ? gl_mul(v, W[e]) : v;
    }
}

__global__ __launch_bounds__(256) void k_ntt4_rows(const u64* __restrict__ in, u64* __restrict__ out, int n, int n1,
                                                   const u64* __restrict__ W, u64 scale) {
    extern __shared__ u64 ntile[];
    constexpr int ST = NTT_TILE + 1;
    const int n2 = n - n1;
    const size_t N = (size_t)1 << n, N1 = (size_t)1 << n1;
    const int N2 = 1 << n2;
    const size_t k1_0 = (size_t)blockIdx.x * NTT_TILE;
    const u64* y = in + (size_t)blockIdx.y * N;
    u64* X = out + (size_t)blockIdx.y * N;
    for (int idx = threadIdx.x; idx < N2 * NTT_TILE; idx += blockDim.x) {
        int r = idx >> n2, i2 = idx & (N2 - 1);
        ntile[i2 * ST + r] = y[(k1_0 + r) * (size_t)N2 + i2];
    }
    __syncthreads();
    lds_ntt_dif<ST>(ntile, n2, W, n1);  // w_N2 = w^(N1)
    for (int idx = threadIdx.x; idx < N2 * NTT_TILE; idx += blockDim.x) {
        int pos = idx >> 4, r = idx & (NTT_TILE - 1);
        u32 k2 = brev_bits((u32)pos, n2);
        u64 v = ntile[pos * ST + r];
        X[k1_0 + r + N1 * (size_t)k2] = scale == 1 ? v : gl_mul(v, scale);
    }
}

// W must hold w^i for i < N (four-step) — the radix-2 fallback only reads i < N/2.
void ntt_batch(hipStream_t st, u64* data, int log2n, size_t batch, const u64* W, u64 scale, u64* scratch) {
    if (log2n >= 8 && log2n <= 16 && scratch) {
        const int n1 = log2n / 2, n2 = log2n - n1;
        const size_t lds = (size_t)(1 << (n1 > n2 ? n1 : n2)) * (NTT_TILE + 1) * sizeof(u64);
        k_ntt4_cols<<<dim3(1u << (n2 - 4), (unsigned)batch), 256, lds, st>>>(data, scratch, log2n, n1, W);
        k_ntt4_rows<<<dim3(1u << (n1 - 4), (unsigned)batch), 256, lds, st>>>(scratch, data, log2n, n1, W, scale);
        return;
    }
    size_t total = batch << (log2n - 1);
    int grid = (int)std::min<size_t>((total + TPB - 1) / TPB, 4096);
    for (int s = log2n - 1; s >= 0; s--) k_ntt_stage<<<grid, TPB, 0, st>>>(data, log2n, s, batch, W);
    k_ntt_bitrev<<<grid * 2, TPB, 0, st>>>(data, log2n, batch, scale);
}

}  // namespace dev
}  // namespace hg
