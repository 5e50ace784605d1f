// BfvEncrypt::prove over bn256::Fr (BASELINE config 5: the reference's `bn254` test family, F = E = Fr): Fr arithmetic on gfx950, the
// Keccak challenge chain over Fr, gkr::sum_check::prove_sum_check for the three shapes of the path, prove_grand_product, the whole
// Lasso node, MLE evaluation and NTT; bn254_gkr.inc (included at the end) adds witness generation, the Libra / zkCNN node
// reductions and the prove_gkr walk. Structure per sum-check as in the Goldilocks prover: all rounds are enqueued with the
// (message-independent) challenges, one copy back, then the transcript is replayed on the host.
// [REF bfv-gkr/src/transcript.rs:146-157,183-189,198-203 (challenges, 32-byte big-endian felts);
//  lasso/src/lasso.rs:457-475 (collation g), lasso/src/memory_checking/prover.rs:268-279 (grand-product g);
//  sk_encryption_circuit.rs:417-460, 614-626 (prove; Fr, Fr)]
#include <hip/hip_runtime.h>
#include <mutex>
#include <thread>
#include <exception>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>
#include <algorithm>
#include <chrono>
#include <tuple>
#include <functional>
#include <memory>
#include <map>
#include "bn254_field.hpp"
#include "bn254_wide.hpp"
#include "bn254_lazy.hpp"
#include "bn254_mfma.hpp"
#include "host.hpp"
#include "prover.hpp"
#include "kernels.hpp"

namespace hg {
namespace bn {

// ---- challenges: c_j = LE(Keccak^j("")) mod r; E = F, so one base challenge per squeeze -----------------------
static Fr fr_from_le32_mod(const uint8_t h[32]) {
    // 256-bit little-endian integer mod r: at most 5 subtractions of r (2^256 / r < 6)
    Fr v;
    memcpy(v.l, h, 32);
    for (;;) {
        bool ge = fr_geq_p(v);
        if (!ge) break;
        v = fr_sub_p(v);
    }
    return v;
}
// The chain does not depend on anything (transcript.rs:146-157: nothing is absorbed): computed once per process and extended on
// demand - a prove squeezes ~9000 challenges, i.e. ~9000 Keccak permutations that used to run at the start of every prove
// while the GPU waited.
static std::mutex g_bn_chain_mu;
static std::vector<Fr> g_bn_chain;
static uint8_t g_bn_chain_h[32];
// copies challenges [from, from + n) into out
static void bn_chain_copy(size_t from, size_t n, Fr* out) {
    std::lock_guard<std::mutex> lk(g_bn_chain_mu);
    if (g_bn_chain.empty()) keccak256(nullptr, 0, g_bn_chain_h);
    if (g_bn_chain.size() < from + n) {
        const size_t want = std::max(from + n, g_bn_chain.size() + 4096);
        g_bn_chain.reserve(want);
        while (g_bn_chain.size() < want) {
            g_bn_chain.push_back(fr_from_le32_mod(g_bn_chain_h));
            uint8_t nx[32];
            keccak256(g_bn_chain_h, 32, nx);
            memcpy(g_bn_chain_h, nx, 32);
        }
    }
    for (size_t i = 0; i < n; i++) out[i] = g_bn_chain[from + i];
}
std::vector<Fr> challenge_chain_bn254(size_t n) {
    std::vector<Fr> out(n);
    if (n) bn_chain_copy(0, n, out.data());
    return out;
}

void challenges_bn254_raw(size_t n, uint64_t* out4) {
    std::vector<Fr> c = challenge_chain_bn254(n);
    for (size_t i = 0; i < n; i++) memcpy(out4 + 4 * i, c[i].l, 32);
}

// ---- kernels ---------------------------------------------------------------------------------------------------
constexpr int BN_TPB = 256;
enum { BN_COLLATION = 0, BN_GRANDPROD = 1, BN_PRODSUM = 2 };

__global__ void k_bn_to_mont(Fr* __restrict__ t, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = fr_to_mont(t[i]);
}
__global__ void k_bn_from_mont(Fr* __restrict__ t, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = fr_from_mont(t[i]);
}
// ops 5 .. 9: the loose arithmetic of bn254_lazy.hpp on RAW 256-bit operands (no conversion: the test feeds the edge values 0, p - 1, p,
// 2p - 1, 2^256 - 1 as they are); results normalised with lz_canon so that the test can compare residues:
//   5  lz_mul(a, b) = a b R^-1 mod p                      (any operands)
//   6  lz_fold(a, b, fold_consts(r)) = a + r b R^-1 mod p, r = 2^200 + 12345 (raw); a < 2p, b any
//   7  lz_add(a, b), 8  lz_subr(a, b)                     (a, b < 2p)
//   9  three products through one lz_reduce: (a b + a a + b b) R^-1 mod p with the subtraction operand lz_sub(a, b) as a factor of a
//      fourth: + lz_sub(a, b) * b                          (a, b < 2p)
//  10  mf_fold (bn254_mfma.hpp): a + r (b - a) R^-1 mod p on the matrix cores, r as in 6 (a, b < 2^256: any bytes)
__global__ void k_bn_lazy_op(int op, size_t n, const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* __restrict__ out, FoldK fk, const MfA* __restrict__ mfa) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (op >= 10) {   // whole waves: a lane beyond n works on entry n - 1 and stores nothing
        const int lane = threadIdx.x & 63;
        Fr r;
        {
            const MfLane K = mf_load(mfa, lane);
            const size_t i0 = i - lane, e0 = i0 + (lane & 31) < n ? i0 + (lane & 31) : n - 1, e1 = i0 + 32 + (lane & 31) < n ? i0 + 32 + (lane & 31) : n - 1;
            const int h16 = 16 * (lane >> 5);
            r = mf_fold(K, mf_gload16(reinterpret_cast<const char*>(&a[e0]) + h16), mf_gload16(reinterpret_cast<const char*>(&b[e0]) + h16),
                        mf_gload16(reinterpret_cast<const char*>(&a[e1]) + h16), mf_gload16(reinterpret_cast<const char*>(&b[e1]) + h16));
        }
        if (i < n) out[i] = lz_canon(r);
        return;
    }
    if (i >= n) return;
    const Fr x = a[i], y = b[i];
    Fr r;
    if (op == 5) r = lz_mul(x, y);
    else if (op == 6) {
        const LzK KK = lz_load_k(fk.k);
        r = lz_fold(x, y, KK.k);
    } else if (op == 7) r = lz_add(x, y);
    else if (op == 8) r = lz_subr(x, y);
    else {
        WCol w = wcol_zero();
        wcol_mac(w, x, y); wcol_mac(w, x, x); wcol_mac(w, y, y); wcol_mac(w, lz_sub(x, y), y);
        r = lz_reduce(w);
    }
    out[i] = lz_canon(r);
}
// op 0 add, 1 sub, 2 mul, 3 / 4 the column-accumulator forms (canonical in / out): the field KAT entry point
__global__ void k_bn_binop(int op, size_t n, const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = fr_to_mont(a[i]), y = fr_to_mont(b[i]);
    Fr r;
    if (op == 3) r = fr_mul_wide(x, y);                       // column-accumulator product (bn254_wide.hpp)
    else if (op == 4) {                                       // a b + a a + b b through one deferred reduction
        WCol w = wcol_zero();
        wcol_mac(w, x, y); wcol_mac(w, x, x); wcol_mac(w, y, y);
        r = wcol_reduce(w);
    } else r = op == 0 ? fr_add(x, y) : (op == 1 ? fr_sub(x, y) : fr_mul(x, y));
    out[i] = fr_from_mont(r);
}

__device__ __forceinline__ Fr block_sum_fr(Fr v, Fr* sm) {
    const int t = threadIdx.x;
    sm[t] = v;
    __syncthreads();
    for (int s = BN_TPB / 2; s > 0; s >>= 1) {
        if (t < s) sm[t] = fr_add(sm[t], sm[t + s]);
        __syncthreads();
    }
    Fr r = sm[0];
    __syncthreads();
    return r;
}

// One round. Tables in Montgomery form, natural order: table t at in + t * 2 * half, pair (T[2j], T[2j+1]) adjacent
// (one 64-byte access per lane). Writes the folded tables (out + t * half) and the per-workgroup sums of
// g(0), g(2)[, g(3)] to partials[blockIdx.x * nv + v].
//   kind 0: g = p_0 * sum_i pw_i p_i;  kind 1: g = p_0 * sum_i pw_i p_2i p_2i+1;  kind 2: g = sum_i p_2i p_2i+1
template <int KIND>
__global__ __launch_bounds__(BN_TPB) void k_bn_round(const Fr* __restrict__ in, Fr* __restrict__ out, int ntab, size_t half, Fr r,
                                                     const Fr* __restrict__ pw, Fr* __restrict__ partials) {
    // gridDim.y = P groups share the tables / pairs round-robin (see k_bn_gp_round_jobs); P = 1 is the plain one-thread-per-j form
    constexpr int NV = KIND == BN_GRANDPROD ? 3 : 2;
    __shared__ Fr sm[BN_TPB];
    Fr acc[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) acc[v] = fr_zero();
    const int P = gridDim.y, pi = blockIdx.y;
    for (size_t j = (size_t)blockIdx.x * BN_TPB + threadIdx.x; j < half; j += (size_t)gridDim.x * BN_TPB) {
        Fr s0 = fr_zero(), s2 = fr_zero(), s3 = fr_zero();
        Fr p0, p2, p3;  // table 0 at 0, 2, 3
        {
            const Fr x = in[2 * j], y = in[2 * j + 1];
            const Fr d = fr_sub(y, x);
            p0 = x; p2 = fr_add(y, d); p3 = fr_add(p2, d);
        }
        if (KIND == BN_COLLATION) {
            WCol c0 = wcol_zero(), c2 = wcol_zero();   // the two weighted sums stay unreduced over the tables
            for (int i = pi; i < ntab; i += P) {
                const Fr x = in[(size_t)i * 2 * half + 2 * j], y = in[(size_t)i * 2 * half + 2 * j + 1];
                const Fr d = fr_sub(y, x);
                const Fr v2 = fr_add(y, d);
                const Fr w = pw[i];
                wcol_mac(c0, w, x);
                wcol_mac(c2, w, v2);
                out[(size_t)i * half + j] = fr_add(x, fr_mul_wide(r, d));
            }
            acc[0] = fr_add(acc[0], fr_mul_wide(p0, wcol_reduce(c0)));
            acc[1] = fr_add(acc[1], fr_mul_wide(p2, wcol_reduce(c2)));
        } else {
            const int nb = ntab >> 1;
            for (int i = pi; i < nb; i += P) {
                const Fr xl = in[(size_t)(2 * i) * 2 * half + 2 * j], yl = in[(size_t)(2 * i) * 2 * half + 2 * j + 1];
                const Fr xr = in[(size_t)(2 * i + 1) * 2 * half + 2 * j], yr = in[(size_t)(2 * i + 1) * 2 * half + 2 * j + 1];
                const Fr dl = fr_sub(yl, xl), dr = fr_sub(yr, xr);
                const Fr l2 = fr_add(yl, dl), r2 = fr_add(yr, dr);
                if (KIND == BN_GRANDPROD) {
                    const Fr l3 = fr_add(l2, dl), r3 = fr_add(r2, dr);
                    const Fr w = pw[i];
                    s0 = fr_add(s0, fr_mul(w, fr_mul(xl, xr)));
                    s2 = fr_add(s2, fr_mul(w, fr_mul(l2, r2)));
                    s3 = fr_add(s3, fr_mul(w, fr_mul(l3, r3)));
                } else {
                    s0 = fr_add(s0, fr_mul(xl, xr));
                    s2 = fr_add(s2, fr_mul(l2, r2));
                }
                out[(size_t)(2 * i) * half + j] = fr_add(xl, fr_mul(r, dl));
                out[(size_t)(2 * i + 1) * half + j] = fr_add(xr, fr_mul(r, dr));
            }
            if (KIND == BN_GRANDPROD) {
                acc[0] = fr_add(acc[0], fr_mul(p0, s0));
                acc[1] = fr_add(acc[1], fr_mul(p2, s2));
                acc[NV - 1] = fr_add(acc[NV - 1], fr_mul(p3, s3));
            } else {
                acc[0] = fr_add(acc[0], s0);
                acc[1] = fr_add(acc[1], s2);
            }
        }
    }
    const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
    for (int v = 0; v < NV; v++) {
        Fr s = block_sum_fr(acc[v], sm);
        if (threadIdx.x == 0) partials[blk * NV + v] = s;
    }
}
// Grand-product round of prove_grand_product, tuned: (a) the LEFT table of pair i enters the first round already multiplied by
// its weight gamma^i (k_bn_weight_rows: two products per (pair, j), once), so no round needs weights and the folded left
// tables stay weighted (the host divides the final left evaluations by gamma^i again); (b) the dot products over the pairs are
// accumulated unreduced in column accumulators (bn254_wide.hpp) and Montgomery-reduced once per pair index j; (c) two lanes share
// a pair index, one per table side (see the kernel): every table entry is loaded once, two accumulators per lane, two waves per
// SIMD; (d) in small rounds the pairs are dealt round-robin to gy thread groups so that a round is not one long serial chain per
// thread (Montgomery reduction is linear: every group reduces its own partial sums and multiplies by p_v itself).
// One launch is round k of every layer that still has one (blockIdx.y = layer; a layer uses gx * gy of the gridDim.x workgroups):
// the layers of a grand product only share the product tree
// left table of pair i at l_base + i * l_stride, right table at r_base + i * r_stride (first round: the pre-weighted left halves and
// the right halves of the level rows; later rounds: the interleaved folded tables)
// mirror (top layer of the Lasso read / write product, see StJob::mirror in kernels.hpp): the job holds the read pairs only; s_in is
// the linear table S = sum_i w_i (l_i + r_i) in natural order ([2j], [2j+1]), folded into s_out; K1 S(t) + K2 joins P0 and P1.
struct GpJobDev { const Fr* l_base; const Fr* r_base; Fr* out; Fr* part; Fr r; unsigned long long half, l_stride, r_stride; int nb, gx, gy, mirror;
                  const Fr* s_in; Fr* s_out; Fr k1, k2;
                  FoldK fk;      // fold_consts(r): the mirrored layer's S table folds through lz_fold (bn254_lazy.hpp)
                  int wg0, pad0; // first workgroup of the job in its round's launch
                  Fr* acc;       // per-thread running sums [gx gy BN_TPB][3] of a job whose threads take more than one pair index (half > gx BN_GP_J), else null
                  MfA* mf; };    // the round's folds x + r (y - x) as an int8 matrix product (bn254_mfma.hpp); filled by k_bn_mf_consts ahead of the rounds
__global__ __launch_bounds__(64) void k_bn_mf_consts_one(const Fr* __restrict__ r, MfA* __restrict__ out) { mf_consts_column(*r, threadIdx.x, out); }
// the MfA of every job of a descriptor array: one workgroup of 64 threads per job
template <typename JOB>
__global__ __launch_bounds__(64) void k_bn_mf_consts(const JOB* __restrict__ jobs) { mf_consts_column(jobs[blockIdx.x].r, threadIdx.x, jobs[blockIdx.x].mf); }
__device__ __forceinline__ Fr fr_swap_lane(const Fr& v) {
    Fr o;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u32 lo = (u32)v.l[i], hi = (u32)(v.l[i] >> 32);
        lo = (u32)__builtin_amdgcn_mov_dpp((int)lo, 0xB1, 0xF, 0xF, true);  // quad_perm:[1,0,3,2]
        hi = (u32)__builtin_amdgcn_mov_dpp((int)hi, 0xB1, 0xF, 0xF, true);
        o.l[i] = ((u64)hi << 32) | lo;
    }
    return o;
}
__device__ __forceinline__ Fr fr_sel(bool c, const Fr& a, const Fr& b) { return fr_make(c ? a.l[0] : b.l[0], c ? a.l[1] : b.l[1], c ? a.l[2] : b.l[2], c ? a.l[3] : b.l[3]); }
// One lane per pair index j, three column accumulators (P0 = sum xl xr, P1 = sum yl yr, Pinf = sum dl dr over the job's pairs; x, y =
// T[2j], T[2j+1], d = y - x), all arithmetic in the branch-free loose form of bn254_lazy.hpp:
//   g(0) = p0 P0, g(2) = p2 (2 P1 - P0 + 2 Pinf), g(3) = p3 (3 P1 - 2 P0 + 6 Pinf)   (p_v = table 0 at v).
// Per (pair, j): four 32-byte loads, three multiply-accumulates, two folds x + r d with the round's precomputed constants, two
// 32-byte stores - one straight-line block of about 900 VALU instructions with three independent accumulation chains. (Round 3 dealt
// a pair index to a lane PAIR with DPP operand swaps and used the canonical, branching field operations: 1700 instructions and 311
// conditional branches per kernel, 0.59 of the issue rate.) The folded tables and the mirrored layer's S table are written LOOSE
// (any representative below 2p): their readers are this kernel, the tail kernel (normalises on load) and fr_from_mont.
constexpr int BN_GP_J = BN_TPB;   // pair indices per workgroup
// Launch: ONE grid dimension over the workgroups of every job of the round (jobmap[blockIdx.x] = job, GpJobDev::wg0 = the job's first
// workgroup). Until round 6 the grid was (largest job's workgroups) x (jobs) and a workgroup beyond its job's count returned at once:
// 134 000 workgroups per prove of which ~15 000 had work - every empty one still claimed 64 KiB of LDS and 256 registers per lane
// for the microsecond its descriptor load took.
__global__ __launch_bounds__(BN_TPB) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_bn_gp_round_jobs(const GpJobDev* __restrict__ jobs, const unsigned short* __restrict__ jobmap) {
    const GpJobDev& J = jobs[jobmap[blockIdx.x]];
    const int tile = (int)blockIdx.x - J.wg0;
    // per wave: the eight 16-byte pieces (xl, yl, xr, yr) of every lane's NEXT TWO (pair, j) items, written by LDS-DMA while the
    // current one is computed on. Two waves per SIMD at 256 VGPRs cannot hide a load -> wait -> compute chain otherwise (measured: the
    // loads alone, one item in flight per wave, take 1.47 ms of a 2.3 ms launch - 64 KiB in flight per CU against a loaded memory
    // latency of 3-5 us is 3.4 TB/s; fetching a wave's contiguous 4 KiB run as coalesced 1 KiB pieces instead was SLOWER, 1.93 ms: the
    // 64-byte-strided reads back from LDS conflict 16 ways). Buffer t mod 2 holds item t; it is refilled with item t + 2 as soon as item
    // t is in registers. The block sums at the end reuse the same bytes.
    __shared__ lz_u32x4 stage[2][BN_TPB / 64][8][64];
    static_assert(sizeof(stage) >= sizeof(Fr) * BN_TPB, "the block sums reuse the staging buffer");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 lds0 = __builtin_amdgcn_readfirstlane((u32)(uintptr_t)(__attribute__((address_space(3))) void*)(&stage[0][wave][0][0]));
    constexpr u32 BUF = (u32)sizeof(stage) / 2;
    const Fr* __restrict__ lb = J.l_base;
    const Fr* __restrict__ rb = J.r_base;
    const size_t ls = J.l_stride, rs = J.r_stride;
    Fr* __restrict__ out = J.out;
    const size_t half = J.half;
    const int nb = J.nb, P = J.gy, bx = tile % J.gx, pi = tile / J.gx;
    const MfLane MK = mf_load(J.mf, lane);   // the round's fold as a matrix product: this lane's rows of A and its C operand
    const size_t jstep = (size_t)J.gx * BN_GP_J;
    auto prefetch = [&](int i, size_t j, u32 buf) {   // lanes beyond the table issue nothing
        if (j < half) {
            const char* gl = reinterpret_cast<const char*>(&lb[(size_t)i * ls + 2 * j]);
            const char* gr = reinterpret_cast<const char*>(&rb[(size_t)i * rs + 2 * j]);
#pragma unroll
            for (int k = 0; k < 4; k++) lz_glds16(gl + 16 * k, lds0 + buf * BUF + 1024u * k);
#pragma unroll
            for (int k = 0; k < 4; k++) lz_glds16(gr + 16 * k, lds0 + buf * BUF + 1024u * (4 + k));
        }
    };
    // the thread's running sums of g(0), g(2), g(3) over its pair indices live in HBM between two of them (J.acc; loose): 24 registers
    // that the matrix-core folds' accumulators need. A thread with one pair index (most jobs) never touches the buffer.
    Fr* __restrict__ accp = J.acc ? J.acc + ((size_t)(pi * J.gx + bx) * BN_TPB + threadIdx.x) * 3 : nullptr;
    Fr acc0 = fr_zero(), acc2 = fr_zero(), acc3 = fr_zero();   // loose
    // The wave's items, in order: t = 0 .. n_items - 1 <-> (jw, i) with i running fastest. All 64 lanes stay in the loops (the counts are
    // uniform over the wave); a lane beyond the table (tables shorter than a wave) is masked where it loads, stores or sums.
    const size_t jw0 = (size_t)bx * BN_GP_J + 64 * wave;
    const u32 npi = (u32)((nb - pi + P - 1) / P);
    const u32 n_items = jw0 < half ? (u32)((half - jw0 + jstep - 1) / jstep) * npi : 0u;
    u32 issued = 0, t = 0;
    int pf_i = pi;
    size_t pf_jw = jw0;
    auto issue_next = [&] {   // item `issued` into buffer issued & 1
        prefetch(pf_i, pf_jw + lane, issued & 1);
        issued++;
        pf_i += P;
        if (pf_i >= nb) { pf_i = pi; pf_jw += jstep; }
    };
    if (issued < n_items) issue_next();
    if (issued < n_items) issue_next();
    for (size_t jw = jw0; jw < half; jw += jstep) {
        const size_t j = jw + lane;
        const bool valid = j < half;
        WCol c0 = wcol_zero(), c1 = wcol_zero(), ci = wcol_zero();
        for (int i = pi; i < nb; i += P, t++) {
            // Item t has landed when nothing older than the eight DMAs of item t + 1 is outstanding (the counter retires in order).
            // NOTHING in this loop may make hipcc wait for memory on its own: a spilled register's reload is followed by a
            // vmcnt(0) that also drains the DMAs in flight (the first form of this loop held the four operands in registers next to
            // the three accumulators, spilled, and ran load -> wait -> compute in sequence: 2.3-2.5 ms for the first launch). So the
            // staged item is the operand store: x and y are read from LDS when a step needs them and dropped again.
            if (issued > t + 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else lz_wait_vm0();
            const lz_u32x4* mine = &stage[t & 1][wave][0][lane];
            // the folds on the matrix cores (bn254_mfma.hpp): the operand of lane (n, h) is piece h of the staged x / y of elements n and
            // 32 + n - the same bytes, read once more in another order
            const mf_v4i* mfp = reinterpret_cast<const mf_v4i*>(&stage[t & 1][wave][lane >> 5][lane & 31]);
            {
                const Fr fl = mf_fold(MK, mfp[0], mfp[128], mfp[32], mfp[160]);
                if (valid) lz_gstore_nt(&out[(size_t)(2 * i) * half + j], fl);
            }
            asm volatile("" ::: "memory");
            {
                const Fr fr_ = mf_fold(MK, mfp[256], mfp[384], mfp[288], mfp[416]);
                if (valid) lz_gstore_nt(&out[(size_t)(2 * i + 1) * half + j], fr_);
            }
            asm volatile("" ::: "memory");
            {
                const Fr xl = lz_from_x4(mine[0], mine[64]), yl = lz_from_x4(mine[128], mine[192]);
                const Fr dl = lz_sub(yl, xl);
                const Fr xr = lz_from_x4(mine[256], mine[320]), yr = lz_from_x4(mine[384], mine[448]);
                const Fr dr = lz_sub(yr, xr);
                wcol_mac(ci, dl, dr);
            }
            asm volatile("" ::: "memory");
            {
                const Fr xl = lz_from_x4(mine[0], mine[64]), xr = lz_from_x4(mine[256], mine[320]);
                wcol_mac(c0, xl, xr);
            }
            asm volatile("" ::: "memory");
            {
                const Fr yl = lz_from_x4(mine[128], mine[192]), yr = lz_from_x4(mine[384], mine[448]);
                wcol_mac(c1, yl, yr);
            }
            lz_wait_lgkm0();  // the item is consumed: its buffer is free for item t + 2
            if (issued < n_items) issue_next();
        }
        const size_t jc = valid ? j : 0;   // (masked lanes read entry 0 and contribute nothing)
        Fr p0, p2, p3;  // table 0 (= left table of pair 0, weight gamma^0 = 1) at 0, 2, 3 (loaded here: not live across the pair loop)
        {
            const Fr x = lz_gload(&lb[2 * jc]), y = lz_gload(&lb[2 * jc + 1]);
            const Fr d = lz_subr(y, x);
            p0 = x; p2 = lz_add(y, d); p3 = lz_add(p2, d);
        }
        Fr P0 = lz_reduce(c0), P1 = lz_reduce(c1);
        const Fr Pi = lz_reduce(ci);
        if (J.mirror && pi == 0) {   // (uniform over the workgroup) P0 += K1 S(0) + K2, P1 += K1 S(1) + K2
            const Fr sx = lz_gload(&J.s_in[2 * jc]), sy = lz_gload(&J.s_in[2 * jc + 1]);
            P0 = lz_add(P0, lz_add(lz_mul(J.k1, sx), J.k2));
            P1 = lz_add(P1, lz_add(lz_mul(J.k1, sy), J.k2));
            const Fr sf = lz_fold(sx, lz_sub(sy, sx), J.fk.k);
            if (valid) lz_gstore(&J.s_out[j], sf);
        }
        const Fr P1x2 = lz_add(P1, P1), Pix2 = lz_add(Pi, Pi);
        const Fr q2 = lz_add(lz_subr(P1x2, P0), Pix2);
        const Fr q3 = lz_add(lz_subr(lz_add(P1x2, P1), lz_add(P0, P0)), lz_add(lz_add(Pix2, Pix2), Pix2));
        const Fr t0 = lz_mul(p0, P0), t2 = lz_mul(p2, q2), t3 = lz_mul(p3, q3);
        acc0 = valid ? t0 : fr_zero();
        acc2 = valid ? t2 : fr_zero();
        acc3 = valid ? t3 : fr_zero();
        if (jw != jw0) {   // (uniform over the wave; implies accp)
            acc0 = lz_add(acc0, lz_gload(&accp[0]));
            acc2 = lz_add(acc2, lz_gload(&accp[1]));
            acc3 = lz_add(acc3, lz_gload(&accp[2]));
        }
        if (jw + jstep < half) {
            lz_gstore(&accp[0], acc0);
            lz_gstore(&accp[1], acc2);
            lz_gstore(&accp[2], acc3);
        }
    }
    lz_wait_vm0();
    __syncthreads();   // every wave is done with its staging bytes
    Fr* sm = reinterpret_cast<Fr*>(&stage[0][0][0]);
    const size_t blk = (size_t)pi * J.gx + bx;
    Fr s = block_sum_fr(lz_canon(acc0), sm);
    if (threadIdx.x == 0) J.part[blk * 3 + 0] = s;
    s = block_sum_fr(lz_canon(acc2), sm);
    if (threadIdx.x == 0) J.part[blk * 3 + 1] = s;
    s = block_sum_fr(lz_canon(acc3), sm);
    if (threadIdx.x == 0) J.part[blk * 3 + 2] = s;
}
// pw[n][b] = g_n^b for every layer n of a grand product in one launch (blockIdx.y = layer)
struct GammaSet { Fr g[32]; };
__global__ void k_bn_gamma_powers(Fr* __restrict__ pw, GammaSet gs, int nb) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    Fr r = fr_one_mont(), base = gs.g[blockIdx.y];
    for (int e = b; e; e >>= 1) { if (e & 1) r = fr_mul_wide(r, base); base = fr_mul_wide(base, base); }
    pw[(size_t)blockIdx.y * nb + b] = r;
}
// out[b][i] = pw[b] * rows[b][i], i < h: the left halves of a level's rows with the layer's weights gamma^b folded in
__global__ void k_bn_weight_rows(const Fr* __restrict__ rows, size_t row_len, const Fr* __restrict__ pw, Fr* __restrict__ out, size_t h, int nb) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= h * (size_t)nb) return;
    const size_t b = idx >> (__ffsll((long long)h) - 1), i = idx & (h - 1);   // (h is a power of two)
    const Fr x = rows[b * row_len + i];
    out[idx] = b == 0 ? x : lz_mul(pw[b], x);
}
// mirrored top layer: the weighted left halves of the READ rows (as k_bn_weight_rows) and, in the same pass, the linear table
// S[i] = sum_b pw[b] (l_b[i] + r_b[i]) over the read rows b < nb (r_b = the right half of row b, in place at rows + h)
__global__ __launch_bounds__(256) void k_bn_weight_rows_sum(const Fr* __restrict__ rows, size_t row_len, const Fr* __restrict__ pw, const FoldK* __restrict__ fk,
                                                          Fr* __restrict__ out, Fr* __restrict__ S, size_t h, int nb) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h) return;
    Fr acc = fr_zero();   // loose throughout (bn254_lazy.hpp)
    WCol c = wcol_zero();
    for (int b = 0; b < nb; b++) {
        const Fr x = lz_gload(&rows[(size_t)b * row_len + i]);
        const Fr lw = b == 0 ? x : lz_fold(fr_zero(), x, fk[b].k);   // (b is the loop index: uniform over the wave)
        lz_gstore(&out[(size_t)b * h + i], lw);
        acc = lz_add(acc, lw);
        wcol_mac(c, pw[b], lz_gload(&rows[(size_t)b * row_len + h + i]));
    }
    lz_gstore(&S[i], lz_add(acc, lz_reduce(c)));
}
// Slot form of the mirrored top layer (grand_product_core): row v of `rows` holds, in every segment pair, the values of one CLASS of
// memories with identical rows there; W[v * npairs + sp] = the sum of the class members' weights, fk = fold_consts of it. Slot 0
// is memory 0 alone (weight one: the p_0 factor of the round polynomial is its plain left table).
__global__ __launch_bounds__(256) void k_bn_weight_slots_sum(const Fr* __restrict__ rows, size_t row_len, const Fr* __restrict__ W, const FoldK* __restrict__ fk,
                                                            Fr* __restrict__ out, Fr* __restrict__ S, size_t h, int V, int npairs, int seg_shift) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h) return;
    const int sp = __builtin_amdgcn_readfirstlane((int)(i >> seg_shift));   // (a wave's 64 consecutive positions lie in one segment)
    Fr acc = fr_zero();   // loose throughout (bn254_lazy.hpp)
    WCol c = wcol_zero();
    for (int v = 0; v < V; v++) {
        const Fr x = lz_gload(&rows[(size_t)v * row_len + i]);
        const Fr lw = v == 0 ? x : lz_fold(fr_zero(), x, fk[(size_t)v * npairs + sp].k);
        lz_gstore(&out[(size_t)v * h + i], lw);
        acc = lz_add(acc, lw);
        wcol_mac(c, W[(size_t)v * npairs + sp], lz_gload(&rows[(size_t)v * row_len + h + i]));
    }
    lz_gstore(&S[i], lz_add(acc, lz_reduce(c)));
}
// After seg_shift rounds a slot table is one entry per segment pair: the per-memory tables of the remaining rounds are gathered from
// them (left tables: weight of the memory over the weight of its class, `ratio`; right tables as they are). in: table t at in + t * npairs
// (left of slot v: t = 2v, right: 2v + 1), out: the same layout over the G2 memories.
__global__ void k_bn_gp_regroup(const Fr* __restrict__ in, Fr* __restrict__ out, const unsigned char* __restrict__ slot_of, const Fr* __restrict__ ratio, int G2, int npairs) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= G2 * npairs) return;
    const int b = idx / npairs, sp = idx % npairs;
    const int v = slot_of[idx];
    const Fr l = in[(size_t)(2 * v) * npairs + sp], r = in[(size_t)(2 * v + 1) * npairs + sp];
    out[(size_t)(2 * b) * npairs + sp] = b == 0 ? l : lz_mul(ratio[idx], l);
    out[(size_t)(2 * b + 1) * npairs + sp] = r;
}
// launch shape of a round over `half` pair indices and `nitems` independent items (pairs / tables): grid.x workgroups along j,
// grid.y groups of items; large rounds keep one thread per j, small ones spread the items
struct RoundGrid { int gx, gy; int blocks() const { return gx * gy; } };
static RoundGrid round_grid(size_t half, int nitems) {
    RoundGrid g;
    // pair indices per thread in the long rounds: a workgroup's fixed cost (first loads, block sums, partial store) is paid once per four
    // (round 6, config-5 prove with 1 / 2 / 4 / 8: 8.6-9.0 / 8.7 / 8.5-8.6 / 8.8-8.9 ms)
    const size_t want = half >= ((size_t)BN_TPB << 4) ? 4 : 1;
    g.gx = (int)std::min<size_t>((half + BN_TPB * want - 1) / (BN_TPB * want), (size_t)1024);
    const size_t threads = (size_t)g.gx * BN_TPB;
    g.gy = (int)std::max<size_t>(1, std::min<size_t>((size_t)nitems, (size_t)131072 / threads));
    return g;
}
// grand-product rounds: one lane per pair index (k_bn_gp_round_jobs), the pairs dealt to gy groups in small rounds
// `launch_wgs`: workgroups the whole launch (round rd of every layer of the products sharing it) has when no job deals its pairs out
// (gy = 1). Dealing a job's pairs to gy groups shortens its serial chain but every group pays the per-j epilogue (three reductions,
// three products: ~5300 instructions against ~1050 per (pair, j)) - worth it only when the launch would otherwise leave the device
// empty. Round 3 decided per job (any job below 2^17 pair indices was dealt out, also inside launches that fill the device many
// times over): 22 % of the kernel's instructions went into those epilogues.
static RoundGrid round_grid_gp(size_t half, int nb, size_t launch_wgs) {
    RoundGrid g;
    g.gx = (int)std::min<size_t>((half + BN_GP_J - 1) / BN_GP_J, (size_t)1024);
    constexpr size_t target = 2048;   // (256 .. 4096 swept in round 4: 13.4 / 13.4 / 13.2 / 13.1 / 13.5 ms)
    const size_t want = launch_wgs ? (target + launch_wgs - 1) / launch_wgs : (size_t)nb;
    g.gy = (int)std::max<size_t>(1, std::min<size_t>((size_t)nb, want));
    return g;
}
constexpr int BN_PART_STRIDE = 2048;  // per-round slots (workgroups) in a partials buffer; round_grid never exceeds 1024 + 512
__global__ __launch_bounds__(BN_TPB) void k_bn_reduce(const Fr* __restrict__ partials, int nblocks, int nv, Fr* __restrict__ out) {
    __shared__ Fr sm[BN_TPB];
    for (int v = 0; v < nv; v++) {
        Fr a = fr_zero();
        for (int b = threadIdx.x; b < nblocks; b += BN_TPB) a = fr_add(a, partials[(size_t)b * nv + v]);
        a = block_sum_fr(a, sm);
        if (threadIdx.x == 0) out[v] = fr_from_mont(a);  // canonical for the host
    }
}

// all rounds of one sum-check in one launch: workgroup rd sums counts.n[rd] per-workgroup partials of round rd
struct RoundCounts { int n[32]; };
__global__ __launch_bounds__(BN_TPB) void k_bn_reduce_rounds(const Fr* __restrict__ partials, RoundCounts counts, int nv, Fr* __restrict__ out) {
    __shared__ Fr sm[BN_TPB];
    const int rd = blockIdx.x;
    const Fr* src = partials + (size_t)rd * BN_PART_STRIDE * nv;
    for (int v = 0; v < nv; v++) {
        Fr a = fr_zero();
        for (int b = threadIdx.x; b < counts.n[rd]; b += BN_TPB) a = fr_add(a, src[(size_t)b * nv + v]);
        a = block_sum_fr(a, sm);
        if (threadIdx.x == 0) out[rd * nv + v] = fr_from_mont(a);  // canonical for the host
    }
}

// the same for many sum-checks: grid (32, jobs)
struct RedJobDev { const Fr* part; Fr* out; int n[32]; int nrounds, pad; };
__global__ __launch_bounds__(BN_TPB) void k_bn_reduce_jobs(const RedJobDev* __restrict__ jobs, int nv) {
    __shared__ Fr sm[BN_TPB];
    const RedJobDev& J = jobs[blockIdx.y];
    const int rd = blockIdx.x;
    if (rd >= J.nrounds) return;
    const Fr* src = J.part + (size_t)rd * BN_PART_STRIDE * nv;
    for (int v = 0; v < nv; v++) {
        Fr a = fr_zero();
        for (int b = threadIdx.x; b < J.n[rd]; b += BN_TPB) a = fr_add(a, src[(size_t)b * nv + v]);
        a = block_sum_fr(a, sm);
        if (threadIdx.x == 0) J.out[rd * nv + v] = fr_from_mont(a);
    }
}

// The last rounds of a sum-check (table length <= 2 * BN_TAIL_HALF) in ONE single-workgroup launch instead of one small launch per
// round: tables in HBM (L2-resident at this size), work items (pair index, pair) dealt to the threads, round sums written in
// canonical form. KIND 2: g = sum_i a_i b_i; KIND 1: g = p_0 * sum_i l_i r_i with the weights already in the left tables
// (rounds after the first of a grand-product layer, see k_bn_gp_round_jobs).
constexpr int BN_TAIL_HALF = 16, BN_TAIL_ROUNDS = 8;  // 16: at most two work items per thread in the first tail round
struct TailR { Fr r[BN_TAIL_ROUNDS]; };
template <int KIND>
__device__ __forceinline__ void bn_tail_body(const Fr* __restrict__ in, Fr* __restrict__ buf, int npairs, int half0, int nrounds, const TailR& rs,
                                             Fr* __restrict__ sums_out, Fr* __restrict__ fin_out) {
    // blockDim.x = NV * BN_TPB: thread group v evaluates the round polynomial at its own point (0, 2[, 3]) and groups 0 / 1 write the
    // left / right folds, so the per-item dependent chain is one or two products instead of eight
    constexpr int NV = KIND == BN_GRANDPROD ? 3 : 2;
    __shared__ Fr sm[NV][BN_TPB];
    const int v = threadIdx.x / BN_TPB, t = threadIdx.x % BN_TPB;
    const int ntab = 2 * npairs;
    const Fr* cur = in;                         // table q at cur + q * 2 * half
    Fr* nxt = buf;                              // table q at nxt + q * half
    for (int rd = 0; rd < nrounds; rd++) {
        const int half = half0 >> rd;
        const Fr r = rs.r[rd];
        Fr acc = fr_zero();
        for (int idx = t; idx < half * npairs; idx += BN_TPB) {
            const int j = idx % half, i = idx / half;
            // loose arithmetic throughout (bn254_lazy.hpp: the round kernels leave loose values, every product below takes them as they
            // are; until round 6 this body normalised on load and multiplied canonically - compare-and-branch code on a dependent chain)
            const Fr xa = cur[(size_t)(2 * i) * 2 * half + 2 * j], ya = cur[(size_t)(2 * i) * 2 * half + 2 * j + 1];
            const Fr xb = cur[(size_t)(2 * i + 1) * 2 * half + 2 * j], yb = cur[(size_t)(2 * i + 1) * 2 * half + 2 * j + 1];
            const Fr da = lz_subr(ya, xa), db = lz_subr(yb, xb);
            Fr av, bv;  // the pair at this group's evaluation point
            if (v == 0) { av = xa; bv = xb; }
            else if (v == 1) { av = lz_add(ya, da); bv = lz_add(yb, db); }
            else { av = lz_add(lz_add(ya, da), da); bv = lz_add(lz_add(yb, db), db); }
            Fr term = lz_mul(av, bv);
            if (KIND == BN_GRANDPROD) {
                const Fr x0 = cur[2 * j], y0 = cur[2 * j + 1];
                const Fr d0 = lz_subr(y0, x0);
                const Fr pv = v == 0 ? x0 : (v == 1 ? lz_add(y0, d0) : lz_add(lz_add(y0, d0), d0));
                term = lz_mul(pv, term);
            }
            acc = lz_add(acc, term);
            if (v == 0) nxt[(size_t)(2 * i) * half + j] = lz_add(xa, lz_mul(r, da));
            else if (v == 1) nxt[(size_t)(2 * i + 1) * half + j] = lz_add(xb, lz_mul(r, db));
        }
        sm[v][t] = acc;
        __syncthreads();
        int s0 = BN_TPB / 2;   // threads at or above half * npairs hold zero: the tree starts at the first level that has a non-zero partner
        while (s0 > 1 && s0 >= half * npairs) s0 >>= 1;
        for (int s = s0; s > 0; s >>= 1) {
            if (t < s) sm[v][t] = lz_add(sm[v][t], sm[v][t + s]);
            __syncthreads();
        }
        if (t == 0) sums_out[rd * NV + v] = fr_from_mont(sm[v][0]);
        __syncthreads();                        // sums consumed; the folded tables are complete (and visible to the workgroup)
        cur = nxt;
        nxt = nxt == buf ? buf + (size_t)ntab * half0 : buf;
    }
    for (int q = threadIdx.x; q < ntab; q += blockDim.x) fin_out[q] = fr_from_mont(cur[q]);
}

// the tails of many sum-checks at once: one workgroup per job
struct TailJobDev { const Fr* in; Fr* buf; Fr* sums_out; Fr* fin_out; TailR rs; int npairs, half0, nrounds, pad; };
template <int KIND>
__global__ __launch_bounds__(BN_TPB * (KIND == BN_GRANDPROD ? 3 : 2)) void k_bn_tail_jobs(const TailJobDev* __restrict__ jobs) {
    const TailJobDev& J = jobs[blockIdx.x];
    bn_tail_body<KIND>(J.in, J.buf, J.npairs, J.half0, J.nrounds, J.rs, J.sums_out, J.fin_out);
}

// ---- PRODSUM rounds (Libra / zkCNN node reductions, bn254_gkr.inc; the collation sum-check of the Lasso node) ----------------
// The second sum is g(2) ITSELF, not a coefficient from which the host could rebuild it with the running claim: the reference's prover
// takes eval(1) from the claim and computes eval(2) (convention C1), and the claim is not the sum of g for the collation sum-check of
// the Lasso node (Expression::poly(0) stands where an eq table would, lasso.rs:457-475) nor for any node of an INVALID witness - the
// transcript must be the reference's there too (test_bn254_invalid_witness_rejected_by_both_verifiers).
struct PsJobDev { const Fr* t[2 * dev::PS_MAX_PAIRS]; Fr* out; Fr* part; Fr r; unsigned long long half; int npairs, gx, gy, pad; unsigned long long wlo, whi; FoldK fk; MfA* mf; int wg0, pad0; };   // fk = fold_consts(r); pad bit 0: every pair has the same b table, bit 1: write its fold once per pair; bit 2: the (single) b table is zero outside the pair indices [wlo, whi), bit 3: write zeros there
// One round of g = sum_i a_i b_i for many independent sum-checks (blockIdx.y = job). Two workgroup sets per tile (v = 0: g(0) = sum xa xb
// and the folds of the a tables, v = 1: g(2) = sum (2 ya - xa)(2 yb - xb) and the folds of the b tables; ids 8 (2 q + v) + xcd keep a
// tile's two workgroups on one XCD, adjacent in dispatch order, so the second reads the tables from L2): one column accumulator, one
// multiply-accumulate and one fold per lane and (pair, j) keeps the kernel at ~120 VGPRs = four waves per SIMD, which is what hides the
// load latency of these mostly small, single-pair jobs (one lane doing both halves needs 259 registers, one wave per SIMD: 1.3x
// slower). Loose arithmetic throughout (bn254_lazy.hpp); the folded tables are loose.
// MF: the folds run on the matrix cores (bn254_mfma.hpp: a wave folds its 64 entries together; needs whole waves inside the table and
// inside / outside a windowed table's window, i.e. half and the window bounds multiples of 64 - every launched round of the large
// parameter sets); otherwise lane by lane through lz_fold.
template <bool MF>
__device__ __forceinline__ void ps_round_body(const PsJobDev& J, int tile, int v, Fr* sm) {
    Fr acc = fr_zero();   // loose
    const int bx = tile % J.gx, pi = tile / J.gx, P = J.gy, npairs = J.npairs;
    const size_t half = J.half;
    Fr* __restrict__ out = J.out;
    const u32* __restrict__ K = J.fk.k;
    const bool shared = J.pad & 1, replicate = J.pad & 2;
    const int lane = threadIdx.x & 63;
    MfLane MK;
    if (MF) MK = mf_load(J.mf, lane);
    for (size_t j = (size_t)bx * BN_TPB + threadIdx.x; j < half; j += (size_t)J.gx * BN_TPB) {
        const size_t jw = j - lane;   // the wave's first entry
        WCol s = wcol_zero();
        if (shared) {
            // every pair has the SAME b table (the node's eq table: one unit relay per position and input): sum_i a_i b = (sum_i a_i) b,
            // one product per evaluation point and ONE fold of b instead of npairs (exact arithmetic: the same field elements)
            const Fr* tb = J.t[1];
            const Fr xb = lz_gload(&tb[2 * j]), yb = lz_gload(&tb[2 * j + 1]);
            Fr sa = fr_zero();
            for (int i = pi; i < npairs; i += P) {
                const Fr* ta = J.t[2 * i];
                if (v == 0) {
                    const Fr xa = lz_gload(&ta[2 * j]);
                    sa = lz_add(sa, xa);
                    Fr f;
                    if (MF) f = mf_fold_global(MK, ta, jw, half, lane);
                    else { const Fr ya = lz_gload(&ta[2 * j + 1]); f = lz_fold(xa, lz_sub(ya, xa), K); }
                    lz_gstore(&out[(size_t)(2 * i) * half + j], f);
                } else {
                    const Fr xa = lz_gload(&ta[2 * j]), ya = lz_gload(&ta[2 * j + 1]);
                    sa = lz_add(sa, lz_add(ya, lz_subr(ya, xa)));
                }
            }
            if (v == 0) wcol_mac(s, sa, xb);
            else {
                const Fr db = lz_sub(yb, xb);
                wcol_mac(s, sa, lz_add(yb, lz_cond_sub_2p(db)));
                if (pi == 0) {
                    const Fr fb = MF ? mf_fold_global(MK, tb, jw, half, lane) : lz_fold(xb, db, K);
                    lz_gstore(&out[half + j], fb);
                    if (replicate)   // the last shared-launch round: the tail workgroup reads one b table per pair
                        for (int i = 1; i < npairs; i++) lz_gstore(&out[(size_t)(2 * i + 1) * half + j], fb);
                }
            }
            acc = lz_add(acc, lz_reduce(s));
            continue;
        }
        if ((J.pad & 4) && (j < J.wlo || j >= J.whi)) {   // (MF: uniform over the wave)
            // the b table is zero here (a chunk node's table covers one 2^L slice of the 2^(L+3) positions): no product, the fold of b
            // is zero and is not even written - the next round does not look outside its (halved) window either, except behind the
            // last shared-launch round, whose successor (the tail workgroup) reads whole tables
            if (v == 0) {
                const Fr* ta = J.t[0];
                Fr f;
                if (MF) f = mf_fold_global(MK, ta, jw, half, lane);
                else { const Fr xa = lz_gload(&ta[2 * j]), ya = lz_gload(&ta[2 * j + 1]); f = lz_fold(xa, lz_sub(ya, xa), K); }
                lz_gstore(&out[j], f);
            } else if (J.pad & 8) lz_gstore(&out[half + j], fr_zero());
            continue;
        }
        for (int i = pi; i < npairs; i += P) {
            const Fr* ta = J.t[2 * i];
            const Fr* tb = J.t[2 * i + 1];
            if (v == 0) {   // (uniform over the workgroup)
                const Fr xa = lz_gload(&ta[2 * j]), xb = lz_gload(&tb[2 * j]);
                wcol_mac(s, xa, xb);
                Fr f;
                if (MF) f = mf_fold_global(MK, ta, jw, half, lane);
                else { const Fr ya = lz_gload(&ta[2 * j + 1]); f = lz_fold(xa, lz_sub(ya, xa), K); }
                lz_gstore(&out[(size_t)(2 * i) * half + j], f);
            } else {
                const Fr xa = lz_gload(&ta[2 * j]), ya = lz_gload(&ta[2 * j + 1]);
                const Fr xb = lz_gload(&tb[2 * j]), yb = lz_gload(&tb[2 * j + 1]);
                const Fr db = lz_sub(yb, xb);
                wcol_mac(s, lz_add(ya, lz_subr(ya, xa)), lz_add(yb, lz_cond_sub_2p(db)));   // (2 ya - xa)(2 yb - xb)
                const Fr f = MF ? mf_fold_global(MK, tb, jw, half, lane) : lz_fold(xb, db, K);
                lz_gstore(&out[(size_t)(2 * i + 1) * half + j], f);
            }
        }
        acc = lz_add(acc, lz_reduce(s));
    }
    const size_t blk = (size_t)pi * J.gx + bx;
    const Fr r = block_sum_fr(lz_canon(acc), sm);
    if (threadIdx.x == 0) J.part[blk * 2 + v] = r;
}
// Launch: one grid dimension over every job's workgroups (jobmap[blockIdx.x >> 4] = job; a job owns a multiple of 16 consecutive
// workgroups from PsJobDev::wg0 on, so that the pairing by XCD below survives); jobmap null: one job, the grid is its own. Until
// round 6 the grid was (largest job) x (jobs): 570 000 workgroups per prove, three quarters of them without work.
__global__ __launch_bounds__(BN_TPB) void k_bn_ps_round_jobs(const PsJobDev* __restrict__ jobs, const unsigned short* __restrict__ jobmap) {
    const PsJobDev& J = jobs[jobmap ? jobmap[blockIdx.x >> 4] : 0];
    const int local = (int)blockIdx.x - J.wg0;
    const int xcd = local & 7, slot = local >> 3, v = slot & 1, tile = (slot >> 1) * 8 + xcd;
    if (tile >= J.gx * J.gy) return;
    __shared__ Fr sm[BN_TPB];
    const bool mf = J.mf && (J.half & 63) == 0 && (!(J.pad & 4) || ((J.wlo | J.whi) & 63) == 0);
    if (mf) ps_round_body<true>(J, tile, v, sm);
    else ps_round_body<false>(J, tile, v, sm);
}

// ---- host arithmetic for the transcript replay (Montgomery form) ---------------------------------------------
static Fr fr_pow(Fr b, const u64 e[4]) {
    Fr r = fr_one_mont();
    for (int w = 3; w >= 0; w--)
        for (int bit = 63; bit >= 0; bit--) {
            r = fr_mul(r, r);
            if ((e[w] >> bit) & 1) r = fr_mul(r, b);
        }
    return r;
}
static Fr fr_inv(Fr a) {
    const u64 e[4] = {FR_P0 - 2, FR_P1, FR_P2, FR_P3};
    return fr_pow(a, e);
}
static Fr fr_small(u64 v) { return fr_to_mont(fr_make(v, 0, 0, 0)); }

static double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void hipc(hipError_t e, const char* what) {
    if (e != hipSuccess) throw Error(std::string(what) + ": " + hipGetErrorString(e));
}

// Scalar results (round sums, folded values, openings) are written by the kernels straight into the context's result buffer,
// which is host-mapped pinned memory: no copy-back calls, the host reads them after a synchronisation.
struct ResRef { Fr* dev; const Fr* host; };
static ResRef res_slots(hg_ctx* ctx, size_t n) {
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(Fr);
    if (ctx->bn_res_used + bytes > ctx->res_cap * sizeof(E2)) throw Error("bn254: result buffer exhausted");
    ResRef r;
    r.dev = reinterpret_cast<Fr*>(reinterpret_cast<char*>(ctx->d_res) + ctx->bn_res_used);
    r.host = reinterpret_cast<const Fr*>(reinterpret_cast<const char*>(ctx->h_res) + ctx->bn_res_used);
    ctx->bn_res_used += bytes;
    return r;
}
static void res_sync(hg_ctx* ctx, hipStream_t st, const char* what) {
    if (ctx->d_res != ctx->h_res && ctx->bn_res_used)
        hipc(hipMemcpyAsync(ctx->h_res, ctx->d_res, ctx->bn_res_used, hipMemcpyDeviceToHost, st), "copy results");
    hipc(hipStreamSynchronize(st), what);
    hipc(hipGetLastError(), what);
}
// Host -> device descriptor arrays: staged in the context's pinned buffer and in its device mirror AT THE SAME OFFSET; bn_flush copies
// everything staged since the last flush in one transfer on the stream whose launches read it (call it after staging, before the
// first launch of a phase). Round 3 issued one hipMemcpyAsync per array: ~100 copy kernels per prove, and a copy from pageable
// memory would make the host wait for the stream.
static void* bn_stage_bytes(hg_ctx* ctx, const void* src, size_t bytes) {
    const size_t need = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    if (!ctx->h_stage || ctx->stage_used + need > ctx->stage_cap) throw Error("bn254: descriptor staging buffer exhausted");
    if (!ctx->bn_dstage) hipc(hipMalloc((void**)&ctx->bn_dstage, ctx->stage_cap), "hipMalloc(descriptor mirror)");
    if (ctx->bn_flushed > ctx->stage_used) ctx->bn_flushed = ctx->stage_used;   // (the staging buffer was reset by another user)
    memcpy(ctx->h_stage + ctx->stage_used, src, bytes);
    void* d = ctx->bn_dstage + ctx->stage_used;
    ctx->stage_used += need;
    return d;
}
template <typename T> static T* bn_stage(hg_ctx* ctx, const T* src, size_t n) { return static_cast<T*>(bn_stage_bytes(ctx, src, n * sizeof(T))); }
static void bn_flush(hg_ctx* ctx, hipStream_t st) {
    if (ctx->bn_flushed > ctx->stage_used) ctx->bn_flushed = ctx->stage_used;
    if (ctx->stage_used == ctx->bn_flushed) return;
    hipc(hipMemcpyAsync(ctx->bn_dstage + ctx->bn_flushed, ctx->h_stage + ctx->bn_flushed, ctx->stage_used - ctx->bn_flushed, hipMemcpyHostToDevice, st), "upload descriptors");
    ctx->bn_flushed = ctx->stage_used;
}
__global__ void k_bn_copy_from_mont(const Fr* __restrict__ src, Fr* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = fr_from_mont(src[i]);
}
static void build_eq_dev(hipStream_t st, Fr* eq, const Fr* pt_canon, int n, Fr* scratch);  // bn254_gkr.inc
static size_t eq_scratch_len(int n);
__global__ void k_bn_powers(Fr* __restrict__ W, Fr w, size_t n);                  // W[i] = w^i (below)

// prove_sum_check on caller tables (all in the base field; E = F). Conventions as for Goldilocks:
// a round message is the d+1 coefficients of the round polynomial, eval(1) = claim - eval(0), lowest variable first.
void sumcheck_bn254(hg_ctx* ctx, int kind, size_t nv, size_t ntab, const u64* const* tables, const u64* pw4, size_t npw, const u64* claim4,
                    size_t chain_skip, u64* msgs, u64* point, u64* evals, u64* sums) {
    if (kind < 0 || kind > 2) throw Error("hg_sumcheck_bn254: kind must be 0, 1 or 2");
    if (ntab == 0 || (kind != BN_COLLATION && (ntab & 1))) throw Error("hg_sumcheck_bn254: bad table count");
    const size_t need_pw = kind == BN_COLLATION ? ntab : (kind == BN_GRANDPROD ? ntab / 2 : 0);
    if (npw < need_pw) throw Error("hg_sumcheck_bn254: too few weights");
    hipc(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t st = ctx->stream;
    const size_t N = (size_t)1 << nv;
    const int d = kind == BN_GRANDPROD ? 3 : 2, NV = d;
    Fr *buf0 = nullptr, *buf1 = nullptr, *d_pw = nullptr, *d_part = nullptr, *d_sums = nullptr;
    const int max_blocks = 1024;
    hipc(hipMalloc((void**)&buf0, ntab * N * sizeof(Fr)), "hipMalloc");
    hipc(hipMalloc((void**)&buf1, ntab * std::max<size_t>(N / 2, 1) * sizeof(Fr)), "hipMalloc");
    hipc(hipMalloc((void**)&d_pw, std::max<size_t>(need_pw, 1) * sizeof(Fr)), "hipMalloc");
    hipc(hipMalloc((void**)&d_part, (size_t)max_blocks * 3 * sizeof(Fr)), "hipMalloc");
    hipc(hipMalloc((void**)&d_sums, std::max<size_t>(nv, 1) * 3 * sizeof(Fr)), "hipMalloc");
    std::vector<Fr> h_sums(nv * NV), h_final(ntab);
    try {
        for (size_t t = 0; t < ntab; t++)
            hipc(hipMemcpyAsync(buf0 + t * N, tables[t], N * sizeof(Fr), hipMemcpyHostToDevice, st), "upload table");
        k_bn_to_mont<<<(unsigned)((ntab * N + 255) / 256), 256, 0, st>>>(buf0, ntab * N);
        if (need_pw) {
            hipc(hipMemcpyAsync(d_pw, pw4, need_pw * sizeof(Fr), hipMemcpyHostToDevice, st), "upload weights");
            k_bn_to_mont<<<(unsigned)((need_pw + 255) / 256), 256, 0, st>>>(d_pw, need_pw);
        }
        const std::vector<Fr> chain = challenge_chain_bn254(chain_skip + nv);
        Fr* cur = buf0;
        Fr* nxt = buf1;
        for (size_t rd = 0; rd < nv; rd++) {
            const size_t half = N >> (rd + 1);
            const Fr r = fr_to_mont(chain[chain_skip + rd]);
            const int grid = (int)std::min<size_t>((half + BN_TPB - 1) / BN_TPB, (size_t)max_blocks);
            if (kind == BN_COLLATION) k_bn_round<BN_COLLATION><<<grid, BN_TPB, 0, st>>>(cur, nxt, (int)ntab, half, r, d_pw, d_part);
            else if (kind == BN_GRANDPROD) k_bn_round<BN_GRANDPROD><<<grid, BN_TPB, 0, st>>>(cur, nxt, (int)ntab, half, r, d_pw, d_part);
            else k_bn_round<BN_PRODSUM><<<grid, BN_TPB, 0, st>>>(cur, nxt, (int)ntab, half, r, d_pw, d_part);
            k_bn_reduce<<<1, BN_TPB, 0, st>>>(d_part, grid, NV, d_sums + rd * NV);
            std::swap(cur, nxt);
        }
        k_bn_from_mont<<<(unsigned)((ntab + 255) / 256), 256, 0, st>>>(cur, ntab);  // the folded scalars (table t at cur + t)
        if (nv) hipc(hipMemcpyAsync(h_sums.data(), d_sums, nv * NV * sizeof(Fr), hipMemcpyDeviceToHost, st), "copy sums");
        hipc(hipMemcpyAsync(h_final.data(), cur, ntab * sizeof(Fr), hipMemcpyDeviceToHost, st), "copy evals");
        hipc(hipStreamSynchronize(st), "sumcheck_bn254: sync");
        hipc(hipGetLastError(), "sumcheck_bn254: launch");
        // transcript replay
        const Fr inv2 = fr_inv(fr_small(2)), inv3 = fr_inv(fr_small(3)), inv6 = fr_inv(fr_small(6)), three = fr_small(3);
        Fr claim = fr_to_mont(fr_make(claim4[0], claim4[1], claim4[2], claim4[3]));
        for (size_t rd = 0; rd < nv; rd++) {
            Fr ev[4], c[4];
            ev[0] = fr_to_mont(h_sums[rd * NV]);
            ev[1] = fr_sub(claim, ev[0]);
            ev[2] = fr_to_mont(h_sums[rd * NV + 1]);
            if (d == 3) ev[3] = fr_to_mont(h_sums[rd * NV + 2]);
            const Fr d1 = fr_sub(ev[1], ev[0]);
            const Fr d2 = fr_add(fr_sub(ev[2], fr_dbl(ev[1])), ev[0]);
            if (d == 2) {
                c[0] = ev[0];
                c[2] = fr_mul(d2, inv2);
                c[1] = fr_sub(d1, c[2]);
            } else {
                const Fr d3 = fr_sub(fr_sub(ev[3], ev[0]), fr_mul(fr_sub(ev[2], ev[1]), three));
                c[0] = ev[0];
                c[3] = fr_mul(d3, inv6);
                c[2] = fr_mul(fr_sub(d2, d3), inv2);
                c[1] = fr_add(fr_sub(d1, fr_mul(d2, inv2)), fr_mul(d3, inv3));
            }
            const Fr rr = fr_to_mont(chain[chain_skip + rd]);
            Fr h = c[d];
            for (int i = d - 1; i >= 0; i--) h = fr_add(fr_mul(h, rr), c[i]);
            claim = h;
            for (int k = 0; k <= d; k++) { Fr o = fr_from_mont(c[k]); memcpy(msgs + (rd * (d + 1) + k) * 4, o.l, 32); }
            memcpy(point + rd * 4, chain[chain_skip + rd].l, 32);
            for (int v = 0; v < NV; v++) memcpy(sums + (rd * NV + v) * 4, h_sums[rd * NV + v].l, 32);
        }
        for (size_t t = 0; t < ntab; t++) memcpy(evals + t * 4, h_final[t].l, 32);
    } catch (...) {
        (void)hipFree(buf0); (void)hipFree(buf1); (void)hipFree(d_pw); (void)hipFree(d_part); (void)hipFree(d_sums);
        throw;
    }
    (void)hipFree(buf0); (void)hipFree(buf1); (void)hipFree(d_pw); (void)hipFree(d_part); (void)hipFree(d_sums);
}

void field_op_bn254(hg_ctx* ctx, int op, size_t n, const u64* a, const u64* b, u64* out) {
    if (op < 0 || op > 10) throw Error("hg_bn254_field_op: op must be 0 (add), 1 (sub), 2 (mul), 3 (wide mul), 4 (wide a b + a a + b b), 5 .. 9 (loose forms, raw operands) or 10 (the matrix-core fold)");
    hipc(hipSetDevice(ctx->device), "hipSetDevice");
    Fr *da = nullptr, *db = nullptr, *dc = nullptr;
    hipc(hipMalloc((void**)&da, n * sizeof(Fr)), "hipMalloc");
    hipc(hipMalloc((void**)&db, n * sizeof(Fr)), "hipMalloc");
    hipc(hipMalloc((void**)&dc, n * sizeof(Fr)), "hipMalloc");
    hipError_t e1 = hipMemcpy(da, a, n * sizeof(Fr), hipMemcpyHostToDevice), e2 = hipMemcpy(db, b, n * sizeof(Fr), hipMemcpyHostToDevice);
    if (e1 == hipSuccess && e2 == hipSuccess) {
        if (op >= 5) {
            FoldK fk;
            Fr rr = fr_make(12345, 0, 0, 1ULL << 8);   // r = 2^200 + 12345 as a raw residue (fold_consts: K_i = r 2^(32 i) R^-1 mod p)
            fold_consts(rr, &fk);
            MfA* mfa = nullptr;
            hipc(hipMalloc((void**)&mfa, sizeof(MfA) + sizeof(Fr)), "hipMalloc");
            Fr* d_r = reinterpret_cast<Fr*>(mfa + 1);
            hipc(hipMemcpyAsync(d_r, &rr, sizeof(Fr), hipMemcpyHostToDevice, ctx->stream), "upload");
            k_bn_mf_consts_one<<<1, 64, 0, ctx->stream>>>(d_r, mfa);
            k_bn_lazy_op<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(op, n, da, db, dc, fk, mfa);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(mfa);
        } else
        k_bn_binop<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(op, n, da, db, dc);
        e1 = hipStreamSynchronize(ctx->stream);
        if (e1 == hipSuccess) e1 = hipMemcpy(out, dc, n * sizeof(Fr), hipMemcpyDeviceToHost);
    }
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc);
    hipc(e1, "hg_bn254_field_op");
    hipc(e2, "hg_bn254_field_op");
}

// ---- prove_grand_product over Fr [REF lasso/src/memory_checking/prover.rs:183-266, 268-294, 297-355] ----------------------
// lw (optional, with pw): the left half of every output row times the row's weight pw[b] - what the sum-check layer that reads this
// level wants as its left tables (k_bn_gp_round_jobs), written here instead of by a second pass over the level (k_bn_weight_rows)
__global__ void k_bn_prod_level(const Fr* __restrict__ in, size_t in_len, Fr* __restrict__ out, int nb, const Fr* __restrict__ pw = nullptr,
                                Fr* __restrict__ lw = nullptr) {
    const size_t h = in_len >> 1, total = h * nb;   // (h is a power of two: shift and mask, not a 64-bit division per thread)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t b = i >> (__ffsll((long long)h) - 1), j = i & (h - 1);
    const Fr v = lz_mul(in[b * in_len + j], in[b * in_len + j + h]);  // Layer::bottom / Layer::up: v_l * v_r on the MSB split (loose in, loose out)
    out[b * h + j] = v;
    if (lw && j < (h >> 1)) lw[b * (h >> 1) + j] = b == 0 ? v : lz_mul(pw[b], v);
}
// the same with the row index on grid.y (uniform per workgroup), so that the weight of row b is a LAUNCH-WIDE constant of the
// workgroup: lw = pw[b] * v runs through fr_fold_const with the row's precomputed constants fk[b] (k_bn_fold_consts)
// slot_of (optional): `in` holds slot rows - row b's values at entry j are those of slot row slot_of[b * ng + (j >> seg_shift)]
__global__ void k_bn_prod_level_rows(const Fr* __restrict__ in, size_t in_len, Fr* __restrict__ out, const FoldK* __restrict__ fk, Fr* __restrict__ lw,
                                     const unsigned char* __restrict__ slot_of = nullptr, int ng = 0, int seg_shift = 0) {
    const size_t h = in_len >> 1, b = blockIdx.y;
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= h) return;
    const size_t rb = slot_of ? slot_of[b * (size_t)ng + (j >> seg_shift)] : b;
    const Fr v = lz_mul(lz_gload(&in[rb * in_len + j]), lz_gload(&in[rb * in_len + j + h]));
    lz_gstore_nt(&out[b * h + j], v);
    if (j < (h >> 1)) lz_gstore(&lw[b * (h >> 1) + j], b == 0 ? v : lz_fold(fr_zero(), v, fk[b].k));
}
// fk[e] = fold_consts(pw[e]) for a run of weights (one thread per (weight, limb row))
__global__ void k_bn_fold_consts(const Fr* __restrict__ pw, FoldK* __restrict__ fk, size_t count) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * 8) return;
    const size_t e = t >> 3;
    const int i = (int)(t & 7);
    Fr x = fr_make(0, 0, 0, 0);
    x.l[i >> 1] = 1ULL << (32 * (i & 1));
    const Fr ki = fr_mul_wide(pw[e], x);
    for (int q = 0; q < 4; q++) { fk[e].k[8 * i + 2 * q] = (u32)ki.l[q]; fk[e].k[8 * i + 2 * q + 1] = (u32)(ki.l[q] >> 32); }
}
// level 1 of a mirrored product: `in` holds the nb/2 READ rows only; row b >= nb/2 of the output is the product of the read row
// b - nb/2 shifted by c (the write rows are never materialised)
__global__ void k_bn_prod_level_mirror(const Fr* __restrict__ in, size_t in_len, Fr* __restrict__ out, int nb, Fr c, Fr c2, FoldK kc, const FoldK* __restrict__ fk = nullptr,
                                       Fr* __restrict__ lw = nullptr, const unsigned char* __restrict__ slot_of = nullptr, int npairs = 0, int seg_shift = 0) {
    // one thread per entry of a READ row (row index on grid.y): its product x y is the read row's level-1 entry, and the write row's
    // entry is (x + c)(y + c) = x y + c (x + y) + c^2 - the multiplication by the launch-wide c goes through fr_fold_const (kc =
    // fold_consts(c)), and so do the rows' weights (fk[b], fk[b + nb/2])
    const size_t h = in_len >> 1, half = (size_t)nb / 2, b = blockIdx.y;
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= h) return;
    const size_t rb = slot_of ? slot_of[b * (size_t)npairs + (j >> seg_shift)] : b;   // slot form: the row that holds memory b's values in this segment pair
    const Fr x = lz_gload(&in[rb * in_len + j]), y = lz_gload(&in[rb * in_len + j + h]);   // loose (k_bn_hash_rw)
    const Fr v = lz_mul(x, y);
    const Fr vw = lz_add(v, lz_fold(c2, lz_add(x, y), kc.k));
    lz_gstore(&out[b * h + j], v);
    lz_gstore(&out[(b + half) * h + j], vw);
    if (lw && j < (h >> 1)) {
        lz_gstore(&lw[b * (h >> 1) + j], b == 0 ? v : lz_fold(fr_zero(), v, fk[b].k));
        lz_gstore(&lw[(b + half) * (h >> 1) + j], lz_fold(fr_zero(), vw, fk[b + half].k));
    }
}
// Level 1 of a mirrored product whose NEXT layer runs in slot form (GpSlots::deep[0]): one thread per entry of a level-1 SLOT row (slot on
// grid.y). rep1 names the row (b < nb/2: read row of memory b, else the write row of memory b - nb/2) whose product the slot row holds
// in the entry's segment group; the level-0 values come from the slot rows of the top layer through slot_of0 as in
// k_bn_prod_level_mirror. lw: the next layer's left halves with the class weights (fkW[slot * ng1 + group]) folded in.
__global__ void k_bn_prod_level_mirror_slots(const Fr* __restrict__ in, size_t in_len, Fr* __restrict__ out, int nb, Fr c2, FoldK kc, const FoldK* __restrict__ fkW,
                                             Fr* __restrict__ lw, const unsigned char* __restrict__ slot_of0, int npairs, const unsigned char* __restrict__ rep1, int ng1,
                                             int seg_shift) {
    const size_t h = in_len >> 1, half = (size_t)nb / 2, v1 = blockIdx.y;
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= h) return;
    const int seg = __builtin_amdgcn_readfirstlane((int)(j >> seg_shift));   // (a wave's 64 consecutive entries lie in one segment)
    const int grp = seg % ng1;
    const int b = rep1[v1 * (size_t)ng1 + grp];
    Fr v = fr_zero();
    if (b != 255) {
        const size_t mem = (size_t)b % half;
        const size_t rb = slot_of0[mem * (size_t)npairs + seg];
        const Fr x = lz_gload(&in[rb * in_len + j]), y = lz_gload(&in[rb * in_len + j + h]);
        v = lz_mul(x, y);
        if ((size_t)b >= half) v = lz_add(v, lz_fold(c2, lz_add(x, y), kc.k));   // (x + c)(y + c)
    }
    lz_gstore_nt(&out[v1 * h + j], v);
    if (j < (h >> 1)) lz_gstore(&lw[v1 * (h >> 1) + j], (v1 == 0 || b == 255) ? v : lz_fold(fr_zero(), v, fkW[v1 * (size_t)ng1 + grp].k));
}
// Level 0 of the mirrored top layer in slot form, ONE pass (round 6): what k_bn_weight_slots_sum (the layer's weighted left halves and
// its S table) and k_bn_prod_level_mirror_slots (level 1 in the next layer's slot rows) each read for themselves - the first all V0
// slot rows once (605 MB at c3), the second a level-0 row once per level-1 row that is made from it (1.3 GB for 22 rows out of 9). One
// thread per position j walks the V0 rows: x = row[j], y = row[j + h] are loaded once, the products x y and (x + c)(y + c) go to every
// level-1 slot row whose representative in j's segment group is a read / write row held by this level-0 row (a uniform scan of rep1),
// the left half and W y join the layer's own tables. Same values as the two kernels (exact arithmetic).
__global__ __launch_bounds__(256) void k_bn_level0_slots_fused(const Fr* __restrict__ in, size_t in_len, Fr* __restrict__ out1, int nb, Fr c2, FoldK kc, const FoldK* __restrict__ fkW1,
                                                              Fr* __restrict__ lw1, const unsigned char* __restrict__ slot_of0, int npairs, const unsigned char* __restrict__ rep1,
                                                              int ng1, int V1, int seg_shift, const Fr* __restrict__ W0, const FoldK* __restrict__ fk0, Fr* __restrict__ lw0,
                                                              Fr* __restrict__ S, int V0) {
    const size_t h = in_len >> 1, half = (size_t)nb / 2;
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // (launched with h / 256 workgroups exactly: h >= 2^seg_shift >= 256)
    // (h is a multiple of the workgroup's 256 positions and they lie in one segment: nobody returned above, the maps below are the workgroup's)
    const int seg = __builtin_amdgcn_readfirstlane((int)(j >> seg_shift));
    const int grp = seg % ng1;
    const bool low = j < (h >> 1);   // (uniform over the wave: h / 2 is a multiple of 64)
    // which level-1 rows take their entry from level-0 row v in this segment: lists per v, built once per workgroup (scanning rep1 /
    // slot_of0 from every (v, v1) was ~200 dependent scalar loads per thread)
    __shared__ unsigned char s_cnt[64], s_list[64][64], s_wr[64][64], s_empty[64];
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        int ne = 0;
        for (int v1 = 0; v1 < V1; v1++) {
            const int b = rep1[v1 * (size_t)ng1 + grp];
            if (b == 255) { s_empty[ne++] = (unsigned char)v1; continue; }
            const int v = slot_of0[((size_t)b % half) * (size_t)npairs + seg];
            const int k = s_cnt[v]++;
            s_list[v][k] = (unsigned char)v1;
            s_wr[v][k] = (size_t)b >= half ? 1 : 0;
        }
        s_cnt[63] = (unsigned char)ne;   // (V0 <= 32: slot 63 is free)
    }
    __syncthreads();
    Fr acc = fr_zero();   // loose throughout (bn254_lazy.hpp)
    WCol c = wcol_zero();
    for (int v = 0; v < V0; v++) {
        const Fr x = lz_gload(&in[(size_t)v * in_len + j]), y = lz_gload(&in[(size_t)v * in_len + j + h]);
        const Fr l0 = v == 0 ? x : lz_fold(fr_zero(), x, fk0[(size_t)v * npairs + seg].k);
        lz_gstore(&lw0[(size_t)v * h + j], l0);
        acc = lz_add(acc, l0);
        wcol_mac(c, W0[(size_t)v * npairs + seg], y);
        const Fr pr = lz_mul(x, y);
        const Fr pw = lz_add(pr, lz_fold(c2, lz_add(x, y), kc.k));   // (x + c)(y + c)
        const int cnt = __builtin_amdgcn_readfirstlane((int)s_cnt[v]);
        for (int k = 0; k < cnt; k++) {
            const int v1 = __builtin_amdgcn_readfirstlane((int)s_list[v][k]);
            const bool wr = __builtin_amdgcn_readfirstlane((int)s_wr[v][k]) != 0;
            const Fr val = wr ? pw : pr;
            lz_gstore_nt(&out1[v1 * h + j], val);
            if (low) lz_gstore(&lw1[v1 * (h >> 1) + j], v1 == 0 ? val : lz_fold(fr_zero(), val, fkW1[v1 * (size_t)ng1 + grp].k));
        }
    }
    {   // slot rows without a member in this segment group
        const int ne = __builtin_amdgcn_readfirstlane((int)s_cnt[63]);
        for (int k = 0; k < ne; k++) {
            const int v1 = __builtin_amdgcn_readfirstlane((int)s_empty[k]);
            lz_gstore_nt(&out1[v1 * h + j], fr_zero());
            if (low) lz_gstore(&lw1[v1 * (h >> 1) + j], fr_zero());
        }
    }
    lz_gstore(&S[j], lz_add(acc, lz_reduce(c)));
}

// A deeper level in slot form from the slot rows of the level above: entry j of slot row v = the product entry of row b = rep_out[v][group
// of j], whose values in the level above are those of slot row slot_of_in[b][group of j there]; lw as in k_bn_prod_level_mirror_slots.
__global__ void k_bn_prod_level_slots(const Fr* __restrict__ in, size_t in_len, Fr* __restrict__ out, const FoldK* __restrict__ fkW, Fr* __restrict__ lw,
                                      const unsigned char* __restrict__ slot_of_in, int ng_in, const unsigned char* __restrict__ rep_out, int ng_out, int seg_shift) {
    const size_t h = in_len >> 1, v = blockIdx.y;
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= h) return;
    const int gin = __builtin_amdgcn_readfirstlane((int)(j >> seg_shift));   // group in the level above (0 .. ng_in - 1: j < h = half a row)
    const int grp = gin % ng_out;
    const int b = rep_out[v * (size_t)ng_out + grp];
    Fr x = fr_zero();
    if (b != 255) {
        const size_t rb = slot_of_in[(size_t)b * ng_in + gin];
        x = lz_mul(lz_gload(&in[rb * in_len + j]), lz_gload(&in[rb * in_len + j + h]));
    }
    lz_gstore_nt(&out[v * h + j], x);
    if (j < (h >> 1)) lz_gstore(&lw[v * (h >> 1) + j], (v == 0 || b == 255) ? x : lz_fold(fr_zero(), x, fkW[v * (size_t)ng_out + grp].k));
}
// The per-row tables of a slot-form layer ahead of its tail: tables of `len` >= ng entries each (entry j belongs to group j >> sh), table
// 2v / 2v + 1 = left / right of slot v at in + t * len; out: the same layout over the nrows rows (left: times ratio[row][group]).
__global__ void k_bn_gp_regroup_tab(const Fr* __restrict__ in, Fr* __restrict__ out, const unsigned char* __restrict__ slot_of, const Fr* __restrict__ ratio, int nrows, int ng,
                                    int len, int sh) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nrows * len) return;
    const int b = idx / len, j = idx % len, g = j >> sh;
    const int v = slot_of[b * ng + g];
    const Fr l = in[(size_t)(2 * v) * len + j], r = in[(size_t)(2 * v + 1) * len + j];
    out[(size_t)(2 * b) * len + j] = b == 0 ? l : lz_mul(ratio[b * ng + g], l);
    out[(size_t)(2 * b + 1) * len + j] = r;
}
// The short end of the product tree in ONE launch, a workgroup per row: every level whose rows are shorter than 256 entries (a level
// reads the one before it, of the same row: the workgroup barrier orders them), then the row's root product and the canonical copies
// of the root and of the two-entry top level for the host - what took a launch per level plus three more per grand product. (Until
// round 6 one workgroup of 1024 threads walked all rows: ~19 000 dependent-chain products on one CU, 93-110 us per grand product;
// 28-33 us now.)
struct ProdTailLevels { const Fr* in; size_t in_len; int nb, nlev; Fr* lev[12]; Fr* lw[12]; const Fr* pw[12]; Fr* top_out; Fr* roots_out; };
__global__ __launch_bounds__(128) void k_bn_prod_tail_levels(ProdTailLevels A) {
    const Fr* in = A.in;
    size_t in_len = A.in_len;
    const size_t b = blockIdx.x;
    for (int q = 0; q < A.nlev; q++) {
        const size_t h = in_len >> 1;
        for (size_t j = threadIdx.x; j < h; j += blockDim.x) {
            const Fr v = lz_mul(in[b * in_len + j], in[b * in_len + j + h]);
            A.lev[q][b * h + j] = v;
            if (A.lw[q] && j < (h >> 1)) A.lw[q][b * (h >> 1) + j] = b == 0 ? v : lz_mul(A.pw[q][b], v);
        }
        __syncthreads();
        in = A.lev[q];
        in_len = h;
    }
    if (threadIdx.x == 0) {   // in: rows of length 2
        const Fr l = in[2 * b], r = in[2 * b + 1];
        A.top_out[2 * b] = fr_from_mont(l);
        A.top_out[2 * b + 1] = fr_from_mont(r);
        A.roots_out[b] = fr_from_mont(lz_mul(l, r));
    }
}
static void write_be32(std::vector<uint8_t>& out, const Fr& canonical) {  // transcript.rs:183-189: repr, byte-reversed
    const size_t at = out.size();
    out.resize(at + 32);
    for (int w = 0; w < 4; w++) {
        const u64 be = __builtin_bswap64(canonical.l[3 - w]);
        memcpy(out.data() + at + 8 * w, &be, 8);
    }
}
// interpolation of one round from g(0), g(2)[, g(3)] and the running claim (conventions C1: d+1 coefficients,
// eval(1) = claim - eval(0)); returns the next claim; all values in Montgomery form
static Fr replay_round(const Fr* sums_canonical, int d, Fr claim, Fr r, Fr* c /* d+1 */) {
    static const Fr inv2 = fr_inv(fr_small(2)), inv3 = fr_inv(fr_small(3)), inv6 = fr_inv(fr_small(6)), three = fr_small(3);
    Fr ev[4];
    ev[0] = fr_to_mont(sums_canonical[0]);
    ev[1] = fr_sub(claim, ev[0]);
    ev[2] = fr_to_mont(sums_canonical[1]);
    if (d == 3) ev[3] = fr_to_mont(sums_canonical[2]);
    const Fr d1 = fr_sub(ev[1], ev[0]);
    const Fr d2 = fr_add(fr_sub(ev[2], fr_dbl(ev[1])), ev[0]);
    if (d == 2) {
        c[0] = ev[0];
        c[2] = fr_mul(d2, inv2);
        c[1] = fr_sub(d1, c[2]);
    } else {
        const Fr d3 = fr_sub(fr_sub(ev[3], ev[0]), fr_mul(fr_sub(ev[2], ev[1]), three));
        c[0] = ev[0];
        c[3] = fr_mul(d3, inv6);
        c[2] = fr_mul(fr_sub(d2, d3), inv2);
        c[1] = fr_add(fr_sub(d1, fr_mul(d2, inv2)), fr_mul(d3, inv3));
    }
    Fr h = c[d];
    for (int i = d - 1; i >= 0; i--) h = fr_add(fr_mul(h, r), c[i]);
    return h;
}

static int ps_nmain(int nv) {   // rounds done by the shared launches; the rest (short tables) by the job's tail workgroup
    const size_t N = (size_t)1 << nv;
    for (int rd = 1; rd < nv; rd++)
        if ((N >> (rd + 1)) <= (size_t)BN_TAIL_HALF && nv - rd <= BN_TAIL_ROUNDS) return rd;
    throw Error("bn254 prodsum: table too short for the tail split");
}
// nb tables of `len` = 2^nv elements each (host, canonical). The transcript is entered after `chain_skip` challenges.
// proof: root products, then per layer the sum-check rounds (4 coefficients each) and the 2 nb evaluations, as 32-byte
// big-endian elements. claims_out: nb final claims, point_out: nv coordinates.
static size_t gp_challenges(int nv) { size_t need = 1; for (int n = 1; n < nv; n++) need += 2 + n; return need; }
// core: level 0 already on the device in Montgomery form (nb rows of len), or uploaded from `tables` when d_lev0 is null
// The round launches of one or several grand products: round rd of every layer of every product in ONE launch (the products are
// as independent of each other as the layers of one product are). grand_product_core fills a set; gp_launch_set runs it.
struct GpLaunchSet {
    std::vector<std::vector<GpJobDev>> by_rd;
    std::vector<RedJobDev> reds;
    std::vector<TailJobDev> tails;
    std::vector<std::function<void()>> posts;   // after the rounds: final values of layers without a tail
    std::vector<std::pair<size_t, std::function<void()>>> pre;   // (round, launch) ahead of that round's launch (the slot form's regroup)
    std::vector<std::function<void()>> pre_tail;                 // ahead of the tail launch (the regroup of a slot-form layer with a tail)
    std::vector<size_t> wgs;                    // per round: workgroups of the jobs queued so far at gy = 1 (round_grid_gp)
    void merge(GpLaunchSet& o) {
        if (wgs.size() < o.wgs.size()) wgs.resize(o.wgs.size(), 0);
        for (size_t rd = 0; rd < o.wgs.size(); rd++) wgs[rd] += o.wgs[rd];
        if (by_rd.size() < o.by_rd.size()) by_rd.resize(o.by_rd.size());
        for (size_t rd = 0; rd < o.by_rd.size(); rd++) by_rd[rd].insert(by_rd[rd].end(), o.by_rd[rd].begin(), o.by_rd[rd].end());
        reds.insert(reds.end(), o.reds.begin(), o.reds.end());
        tails.insert(tails.end(), o.tails.begin(), o.tails.end());
        posts.insert(posts.end(), o.posts.begin(), o.posts.end());
        pre.insert(pre.end(), o.pre.begin(), o.pre.end());
        pre_tail.insert(pre_tail.end(), o.pre_tail.begin(), o.pre_tail.end());
    }
};
static void gp_launch_set(hg_ctx* ctx, hipStream_t st, GpLaunchSet& S) {
    std::vector<GpJobDev> descs;
    std::vector<size_t> off, map_off;
    std::vector<int> total_blocks;
    std::vector<unsigned short> jobmap;   // per round: workgroup -> job of the round
    for (auto& v : S.by_rd) {
        off.push_back(descs.size());
        map_off.push_back(jobmap.size());
        int wg = 0;
        for (size_t q = 0; q < v.size(); q++) {
            v[q].wg0 = wg;
            wg += v[q].gx * v[q].gy;
            jobmap.insert(jobmap.end(), (size_t)v[q].gx * v[q].gy, (unsigned short)q);
        }
        total_blocks.push_back(wg);
        descs.insert(descs.end(), v.begin(), v.end());
    }
    if (descs.empty()) return;
    {   // the folds' matrices, one per (layer, round): built on the device from the descriptors' challenges, one launch for the whole set
        MfA* mfa = static_cast<MfA*>(ctx->alloc(descs.size() * sizeof(MfA)));
        for (size_t i = 0; i < descs.size(); i++) descs[i].mf = mfa + i;
    }
    const GpJobDev* d_descs = bn_stage(ctx, descs.data(), descs.size());
    const unsigned short* d_map = bn_stage(ctx, jobmap.data(), jobmap.size());
    const RedJobDev* d_reds = S.reds.empty() ? nullptr : bn_stage(ctx, S.reds.data(), S.reds.size());
    const TailJobDev* d_tails = S.tails.empty() ? nullptr : bn_stage(ctx, S.tails.data(), S.tails.size());
    bn_flush(ctx, st);
    k_bn_mf_consts<GpJobDev><<<(unsigned)descs.size(), 64, 0, st>>>(d_descs);
    for (size_t rd = 0; rd < S.by_rd.size(); rd++) {
        for (auto& pf : S.pre) if (pf.first == rd) pf.second();
        if (!S.by_rd[rd].empty()) k_bn_gp_round_jobs<<<(unsigned)total_blocks[rd], BN_TPB, 0, st>>>(d_descs + off[rd], d_map + map_off[rd]);
    }
    if (!S.reds.empty()) k_bn_reduce_jobs<<<dim3(32, (unsigned)S.reds.size()), BN_TPB, 0, st>>>(d_reds, 3);
    for (auto& f : S.pre_tail) f();
    if (!S.tails.empty()) k_bn_tail_jobs<BN_GRANDPROD><<<(unsigned)S.tails.size(), 3 * BN_TPB, 0, st>>>(d_tails);
    for (auto& f : S.posts) f();
}
// Slot form of the mirrored top layer (the Lasso read / write product). The hash of memory m at row j is h(dim_c[j], E_m[j], ts_c[j]):
// chunk value and counter belong to the memory's CHUNK position, E_m[j] is zero on every row whose lookup does not use m, and rows come
// in segments of 2^seg_shift per lookup - inside a segment all unused memories of a chunk position have the same row. The top layer
// multiplies the two HALVES of a row (prover.rs:308-313), i.e. segment s with segment s + npairs: memories that are in the same class
// in BOTH segments ("joint class") contribute w_b l r with the same l r. `slot_of[b][sp]` numbers the joint classes of segment pair
// sp (slot 0 = memory 0 alone: its plain left table is the p_0 factor), `rep[v][sp]` names a member. Level 0 is then V slot rows
// instead of nb / 2 memory rows, the layer's first seg_shift rounds run on V pairs whose left tables carry the class weights
// sum_{b in class} gamma^b; after them a table is one entry per segment pair and the per-memory tables are gathered back
// (k_bn_gp_regroup) for the remaining rounds. Same field elements as the memory form (exact arithmetic).
struct GpSlots {
    int V = 0, npairs = 0, seg_shift = 0, G2 = 0;
    std::vector<unsigned char> slot_of;   // [G2][npairs]
    std::vector<unsigned char> rep;       // [V][npairs]
    const unsigned char* d_slot_of = nullptr;
    const unsigned char* d_rep = nullptr;
    // The layers below the top one in slot form as well (HG_BN_SLOT_DEPTH = number of slot-form layers, 1 = the top one only): the first multiplies FOUR segments npairs / 2 apart,
    // its rows are the nb read and write rows; V classes per group of four (slot 0 = row 0 alone). Level 1 of the tree is then V
    // slot rows (k_bn_prod_level_mirror_slots: entry j of slot row v = the product entry of row rep[v][group of j]), whose left
    // halves carry the class weights; the layer's shared rounds run on V pairs and k_bn_gp_regroup_tab gathers the per-row tables for its tail.
    // deep[q]: layer nv - 2 - q (2^(q+2) segments npairs >> (q+1) apart); level q + 1 of the tree is its V slot rows
    // (k_bn_prod_level_slots from the slot rows of the level above), the first level below the last slot-form layer is read back per row.
    struct Deep {
        int V = 0, ng = 0;
        std::vector<unsigned char> slot_of;  // [nb][ng]
        std::vector<unsigned char> rep;      // [V][ng] (255: no such class in that group)
        const unsigned char* d_slot_of = nullptr;
        const unsigned char* d_rep = nullptr;
    };
    std::vector<Deep> deep;
};
// mirror_c (Montgomery, optional): rows nb/2 .. nb-1 of level 0 are rows 0 .. nb/2-1 plus this constant (the Lasso write hashes are
// the read hashes + gamma^2): the top layer then runs on the read rows only (GpJobDev::mirror) and d_lev0 HOLDS ONLY THOSE nb/2
// ROWS - level 1 is computed from them (k_bn_prod_level_mirror). Needs len >= 4 (a level 1 and a sum-check layer on level 0).
static void grand_product_core(hg_ctx* ctx, size_t nb, size_t len, const u64* const* tables, const Fr* d_lev0, size_t chain_skip,
                               std::vector<uint8_t>& proof, std::vector<Fr>& claims_canon, std::vector<Fr>& point_canon, const Fr* mirror_c = nullptr,
                               std::function<void()>* defer = nullptr, GpLaunchSet* set = nullptr, const GpSlots* slots = nullptr) {
    if (set && !defer) throw Error("grand_product_core: a shared launch set needs the deferred form");
    if (slots && (!mirror_c || !d_lev0 || slots->G2 != (int)(nb / 2) || ((len / 2) >> slots->seg_shift) != (size_t)slots->npairs)) throw Error("grand_product_core: slot plan does not fit");
    if (nb == 0 || len < 2 || (len & (len - 1))) throw Error("hg_grand_product_bn254: need nb >= 1 tables of a power-of-two length >= 2");
    hipc(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t st = ctx->stream;
    int nv = 0;
    while (((size_t)1 << nv) < len) nv++;
    const size_t ntab = 2 * nb;
    // challenges in protocol order: mu_0, then per layer n >= 1: gamma, n round challenges, mu
    size_t need = 1;
    for (int n = 1; n < nv; n++) need += 2 + n;
    const std::vector<Fr> chain = challenge_chain_bn254(chain_skip + need);
    // buffers come from the context's bump arena and are handed back (rewind) when the copy-back has completed
    const std::vector<size_t> arena_mark = ctx->arena_mark();
    auto dalloc = [&](size_t n_fr) { return static_cast<Fr*>(ctx->alloc(std::max<size_t>(n_fr, 1) * sizeof(Fr))); };
    struct LayerRec { size_t gamma_at, r_at, mu_at; Fr* d_sums; Fr* d_final; const Fr* sums; const Fr* fin; };
    std::vector<LayerRec> layers(nv);
    const Fr *h_top = nullptr, *h_roots = nullptr;
    try {
        std::vector<const Fr*> lev(nv, nullptr);
        if (d_lev0) lev[0] = d_lev0;
        else {
            Fr* l0 = dalloc(nb * len);
            for (size_t b = 0; b < nb; b++) hipc(hipMemcpyAsync(l0 + b * len, tables[b], len * sizeof(Fr), hipMemcpyHostToDevice, st), "upload table");
            k_bn_to_mont<<<(unsigned)((nb * len + 255) / 256), 256, 0, st>>>(l0, nb * len);
            lev[0] = l0;
        }
        size_t pos = chain_skip;
        layers[0].mu_at = pos++;
        // plan every layer (buffers, result slots, launch shapes), then run the rounds of all layers round-synchronised
        struct LayerPlan { Fr *buf0, *buf1, *part, *tbuf, *d_pw, *lw, *S, *sbuf0, *sbuf1; int nmain; bool mirror; };
        const size_t G2 = nb / 2;
        if (mirror_c && (nv < 2 || (nb & 1) || !d_lev0)) throw Error("hg_grand_product_bn254: mirrored rows need an even batch, two layers and device rows");
        std::vector<LayerPlan> plan(nv);
        const Fr* slot_ratio = nullptr;   // slot form of the top layer: gamma^b / class weight per (memory, segment pair)
        Fr* slot_regroup = nullptr;       // ... and the per-memory tables gathered after its first seg_shift rounds
        int max_main = 0;
        if (nv > 32) throw Error("hg_grand_product_bn254: more than 32 layers");
        Fr* pw_all = dalloc((size_t)nv * nb);
        GammaSet gammas;
        for (auto& g : gammas.g) g = fr_zero();
        for (int n = 1; n < nv; n++) {
            LayerRec& L = layers[n];
            L.gamma_at = pos++; L.r_at = pos; pos += n; L.mu_at = pos++;
            if (n > 32) throw Error("hg_grand_product_bn254: more than 32 rounds");
            const size_t h = (size_t)1 << n;  // table length of this layer's sum-check (n variables)
            LayerPlan& P = plan[n];
            P.d_pw = pw_all + (size_t)n * nb;                                                                       // gamma_n^b, b < nb
            gammas.g[n] = fr_to_mont(chain[L.gamma_at]);
            P.lw = dalloc(nb * h);                                                                                  // weighted left halves
            P.buf0 = dalloc(ntab * (h / 2));
            P.buf1 = dalloc(ntab * std::max<size_t>(h / 4, 1));
            P.part = dalloc((size_t)n * BN_PART_STRIDE * 3);
            P.tbuf = dalloc(2 * ntab * (size_t)BN_TAIL_HALF);
            const ResRef rs_sums = res_slots(ctx, (size_t)n * 3), rs_fin = res_slots(ctx, ntab);
            L.d_sums = rs_sums.dev; L.sums = rs_sums.host;
            L.d_final = rs_fin.dev; L.fin = rs_fin.host;
            P.nmain = n;   // rounds done by the shared launches; the rest (short tables) by the tail workgroup of the layer
            P.mirror = mirror_c && n == nv - 1;   // (every round of a mirrored layer runs in the shared launches: the tail kernel knows no S table)
            P.S = P.sbuf0 = P.sbuf1 = nullptr;
            if (P.mirror) { P.S = dalloc(h); P.sbuf0 = dalloc(h / 2); P.sbuf1 = dalloc(std::max<size_t>(h / 4, 1)); }
            for (int rd = 1; rd < n && !P.mirror; rd++)
                if ((h >> (rd + 1)) <= (size_t)BN_TAIL_HALF && n - rd <= BN_TAIL_ROUNDS) { P.nmain = rd; break; }
            max_main = std::max(max_main, P.nmain);
        }
        FoldK* fk_all = static_cast<FoldK*>(ctx->alloc((size_t)nv * nb * sizeof(FoldK)));   // fold_consts of every weight gamma_n^b
        if (nv > 1) {
            k_bn_gamma_powers<<<dim3((unsigned)((nb + 255) / 256), nv), 256, 0, st>>>(pw_all, gammas, (int)nb);
            k_bn_fold_consts<<<(unsigned)(((size_t)nv * nb * 8 + 255) / 256), 256, 0, st>>>(pw_all, fk_all, (size_t)nv * nb);
        }
        // slot form of the layers below the top one (GpSlots::deep): class weights, their fold constants, gamma^b / W per (row, group)
        int D1 = 0;   // deep layers in slot form: layer nv - 2 - q for q < D1 (each needs a tail that starts while a table still has an entry per group)
        while (slots && D1 < (int)slots->deep.size() && nv - 2 - D1 >= 1 && plan[nv - 2 - D1].nmain < nv - 2 - D1 && plan[nv - 2 - D1].nmain <= slots->seg_shift &&
               (len >> (D1 + 2)) >= 256) D1++;   // (levels 1 .. D1 + 1 go through the long-row kernels)
        std::vector<const FoldK*> fkW(D1, nullptr);
        std::vector<const Fr*> deep_ratio(D1, nullptr);
        for (int q = 0; q < D1; q++) {
            const GpSlots::Deep& dp = slots->deep[q];
            const int n = nv - 2 - q, V1 = dp.V, NG = dp.ng;
            const Fr g = fr_to_mont(chain[layers[n].gamma_at]);
            std::vector<Fr> pwh(nb), W((size_t)V1 * NG, fr_zero()), ratio((size_t)nb * NG, fr_zero());
            { Fr w = fr_one_mont(); for (size_t b = 0; b < nb; b++) { pwh[b] = w; w = fr_mul(w, g); } }
            for (size_t b = 0; b < nb; b++)
                for (int u = 0; u < NG; u++) { Fr& x = W[(size_t)dp.slot_of[b * NG + u] * NG + u]; x = fr_add(x, pwh[b]); }
            std::vector<size_t> idx;
            for (size_t u = 0; u < W.size(); u++) if (W[u].l[0] | W[u].l[1] | W[u].l[2] | W[u].l[3]) idx.push_back(u);
            std::vector<Fr> pre(idx.size() + 1, fr_one_mont());
            for (size_t u = 0; u < idx.size(); u++) pre[u + 1] = fr_mul(pre[u], W[idx[u]]);
            Fr inv = fr_inv(pre[idx.size()]);
            std::vector<Fr> Winv(W.size(), fr_zero());
            for (size_t u = idx.size(); u-- > 0;) { Winv[idx[u]] = fr_mul(inv, pre[u]); inv = fr_mul(inv, W[idx[u]]); }
            for (size_t b = 0; b < nb; b++)
                for (int u = 0; u < NG; u++) {
                    const size_t w = (size_t)dp.slot_of[b * NG + u] * NG + u;
                    if (!(W[w].l[0] | W[w].l[1] | W[w].l[2] | W[w].l[3])) throw Error("hg_grand_product_bn254: degenerate batching challenge (a class weight is zero)");
                    ratio[b * NG + u] = fr_mul(pwh[b], Winv[w]);
                }
            const Fr* dW = bn_stage(ctx, W.data(), W.size());
            deep_ratio[q] = bn_stage(ctx, ratio.data(), ratio.size());
            bn_flush(ctx, st);
            FoldK* fk = static_cast<FoldK*>(ctx->alloc(W.size() * sizeof(FoldK)));
            k_bn_fold_consts<<<(unsigned)((W.size() * 8 + 255) / 256), 256, 0, st>>>(dW, fk, W.size());
            fkW[q] = fk;
        }
        // the mirrored top layer's class weights (slot form): W[v][sp] = sum of gamma^b over the members, their fold constants, and
        // gamma^b / W for the regroup - ahead of the product tree, whose first pass shares the layer's level-0 reads (k_bn_level0_slots_fused)
        const Fr* top_dW = nullptr;
        FoldK* top_fkW = nullptr;
        if (nv > 1 && plan[nv - 1].mirror && slots) {
            const int n = nv - 1;
            const int V = slots->V, NP = slots->npairs;
            const Fr g = fr_to_mont(chain[layers[n].gamma_at]);
            std::vector<Fr> pwh(G2), W((size_t)V * NP, fr_zero()), ratio((size_t)G2 * NP, fr_zero());
            { Fr w = fr_one_mont(); for (size_t b = 0; b < G2; b++) { pwh[b] = w; w = fr_mul(w, g); } }
            for (size_t b = 0; b < G2; b++)
                for (int sp = 0; sp < NP; sp++) { Fr& x = W[(size_t)slots->slot_of[b * NP + sp] * NP + sp]; x = fr_add(x, pwh[b]); }
            {   // one inversion for all class weights (prefix products), empty slots (weight zero) left out
                std::vector<size_t> idx;
                for (size_t q = 0; q < W.size(); q++) if (W[q].l[0] | W[q].l[1] | W[q].l[2] | W[q].l[3]) idx.push_back(q);
                std::vector<Fr> pre(idx.size() + 1, fr_one_mont());
                for (size_t q = 0; q < idx.size(); q++) pre[q + 1] = fr_mul(pre[q], W[idx[q]]);
                Fr inv = fr_inv(pre[idx.size()]);
                std::vector<Fr> Winv(W.size(), fr_zero());
                for (size_t q = idx.size(); q-- > 0;) { Winv[idx[q]] = fr_mul(inv, pre[q]); inv = fr_mul(inv, W[idx[q]]); }
                for (size_t b = 0; b < G2; b++)
                    for (int sp = 0; sp < NP; sp++) {
                        const size_t w = (size_t)slots->slot_of[b * NP + sp] * NP + sp;
                        if (!(W[w].l[0] | W[w].l[1] | W[w].l[2] | W[w].l[3])) throw Error("hg_grand_product_bn254: degenerate batching challenge (a class weight is zero)");
                        ratio[b * NP + sp] = fr_mul(pwh[b], Winv[w]);
                    }
            }
            top_dW = bn_stage(ctx, W.data(), W.size());
            slot_ratio = bn_stage(ctx, ratio.data(), ratio.size());
            bn_flush(ctx, st);
            top_fkW = static_cast<FoldK*>(ctx->alloc(W.size() * sizeof(FoldK)));
            k_bn_fold_consts<<<(unsigned)((W.size() * 8 + 255) / 256), 256, 0, st>>>(top_dW, top_fkW, W.size());
            slot_regroup = dalloc(2 * G2 * (size_t)slots->npairs);
        }
        static const bool no_fused0 = hg_env_on("HG_BN_NO_FUSED0");
        const bool fused0 = top_dW && mirror_c && D1 >= 1 && !no_fused0 && slots->seg_shift >= 8 && ((len >> 1) & 255) == 0 && slots->V <= 32 && slots->deep[0].V <= 64;   // level 0 read once for the layer's own tables and for level 1
        // the product tree; level k (rows of length len >> k) is read by layer n = nv - 1 - k, whose weighted left halves are written
        // in the same pass (level 0, the input, gets its own pass below)
        ProdTailLevels tl;
        memset(&tl, 0, sizeof(tl));
        tl.nb = (int)nb;
        for (int k = 1; k < nv; k++) {
            Fr* lk = dalloc(nb * (len >> k));
            const int n = nv - 1 - k;
            const Fr* pw_n = n >= 1 ? plan[n].d_pw : nullptr;
            Fr* lw_n = n >= 1 ? plan[n].lw : nullptr;
            if (k == 1 && mirror_c) {   // level 0: read rows only
                FoldK kc;
                fold_consts(*mirror_c, &kc);
                const size_t hh = len >> 1;
                if (fused0)
                    k_bn_level0_slots_fused<<<(unsigned)((hh + 255) / 256), 256, 0, st>>>(lev[0], len, lk, (int)nb, fr_mul(*mirror_c, *mirror_c), kc, fkW[0], lw_n, slots->d_slot_of, slots->npairs,
                                                                                      slots->deep[0].d_rep, slots->deep[0].ng, slots->deep[0].V, slots->seg_shift, top_dW, top_fkW,
                                                                                      plan[nv - 1].lw, plan[nv - 1].S, slots->V);
                else if (D1 >= 1) k_bn_prod_level_mirror_slots<<<dim3((unsigned)((hh + 255) / 256), (unsigned)slots->deep[0].V), 256, 0, st>>>(lev[0], len, lk, (int)nb, fr_mul(*mirror_c, *mirror_c), kc, fkW[0], lw_n,
                                                                                                                      slots->d_slot_of, slots->npairs, slots->deep[0].d_rep, slots->deep[0].ng, slots->seg_shift);
                else
                k_bn_prod_level_mirror<<<dim3((unsigned)((hh + 255) / 256), (unsigned)(nb / 2)), 256, 0, st>>>(lev[0], len, lk, (int)nb, *mirror_c, fr_mul(*mirror_c, *mirror_c), kc,
                                                                                                      n >= 1 ? fk_all + (size_t)n * nb : nullptr, lw_n,
                                                                                                      slots ? slots->d_slot_of : nullptr, slots ? slots->npairs : 0, slots ? slots->seg_shift : 0);
            }
            else if (lw_n && (len >> k) >= 256) {   // long rows: the row index on grid.y, the row's weight as a launch-wide constant
                const size_t hh = len >> k;
                if (k >= 2 && k - 1 < D1)          // slot rows of the level above -> this level's slot rows (layer nv - 1 - k = deep[k - 1])
                    k_bn_prod_level_slots<<<dim3((unsigned)((hh + 255) / 256), (unsigned)slots->deep[k - 1].V), 256, 0, st>>>(lev[k - 1], len >> (k - 1), lk, fkW[k - 1], lw_n, slots->deep[k - 2].d_slot_of,
                                                                                                                      slots->deep[k - 2].ng, slots->deep[k - 1].d_rep, slots->deep[k - 1].ng, slots->seg_shift);
                else if (k >= 2 && k - 2 < D1)     // the first level below the slot-form layers: per row, read through the map of the layer above
                    k_bn_prod_level_rows<<<dim3((unsigned)((hh + 255) / 256), (unsigned)nb), 256, 0, st>>>(lev[k - 1], len >> (k - 1), lk, fk_all + (size_t)n * nb, lw_n,
                                                                                                   slots->deep[k - 2].d_slot_of, slots->deep[k - 2].ng, slots->seg_shift);
                else
                k_bn_prod_level_rows<<<dim3((unsigned)((hh + 255) / 256), (unsigned)nb), 256, 0, st>>>(lev[k - 1], len >> (k - 1), lk, fk_all + (size_t)n * nb, lw_n);
            }
            else {   // short rows: queued for the single-workgroup launch below
                if (tl.nlev == 0) { tl.in = lev[k - 1]; tl.in_len = len >> (k - 1); }
                if (tl.nlev >= 12) throw Error("hg_grand_product_bn254: more than 12 short tree levels");
                tl.lev[tl.nlev] = lk; tl.lw[tl.nlev] = lw_n; tl.pw[tl.nlev] = pw_n; tl.nlev++;
            }
            lev[k] = lk;
        }
        // short levels, roots and top evaluations (level nv-1 has rows of length 2)
        if (tl.nlev == 0) { tl.in = lev[nv - 1]; tl.in_len = 2; }
        const ResRef top = res_slots(ctx, 2 * nb), roots = res_slots(ctx, nb);
        tl.top_out = top.dev; tl.roots_out = roots.dev;
        k_bn_prod_tail_levels<<<(unsigned)nb, 128, 0, st>>>(tl);
        h_top = top.host; h_roots = roots.host;
        if (nv > 1) {   // the top layer reads level 0
            const int n = nv - 1;
            const size_t h = (size_t)1 << n;
            if (plan[n].mirror && slots) {
                if (!fused0) k_bn_weight_slots_sum<<<(unsigned)((h + 255) / 256), 256, 0, st>>>(lev[0], 2 * h, top_dW, top_fkW, plan[n].lw, plan[n].S, h, slots->V, slots->npairs, slots->seg_shift);
            }
            else if (plan[n].mirror) k_bn_weight_rows_sum<<<(unsigned)((h + 255) / 256), 256, 0, st>>>(lev[0], 2 * h, plan[n].d_pw, fk_all + (size_t)n * nb, plan[n].lw, plan[n].S, h, (int)G2);
            else k_bn_weight_rows<<<(unsigned)((nb * h + 255) / 256), 256, 0, st>>>(lev[0], 2 * h, plan[n].d_pw, plan[n].lw, h, (int)nb);
        }
        GpLaunchSet own;
        own.by_rd.resize(max_main);
        std::vector<RedJobDev>& reds = own.reds;
        std::vector<TailJobDev>& tails = own.tails;
        std::vector<int> red_index(nv, -1);
        for (int n = 1; n < nv; n++) { red_index[n] = (int)reds.size(); RedJobDev r; memset(&r, 0, sizeof(r)); reds.push_back(r); }
        own.wgs.assign(max_main, 0);
        for (int rd = 0; rd < max_main; rd++)
            for (int n = 1; n < nv; n++)
                if (rd < plan[n].nmain) own.wgs[rd] += std::min<size_t>(((((size_t)1 << n) >> (rd + 1)) + BN_GP_J - 1) / BN_GP_J, (size_t)1024);
        for (int rd = 0; rd < max_main; rd++) {
            const size_t launch_wgs = own.wgs[rd] + (set && (size_t)rd < set->wgs.size() ? set->wgs[rd] : 0);
            for (int n = 1; n < nv; n++) {
                const LayerPlan& P = plan[n];
                if (rd >= P.nmain) continue;
                const size_t h = (size_t)1 << n, half = h >> (rd + 1);
                GpJobDev d;
                memset(&d, 0, sizeof(d));
                const bool slot_layer = slots && P.mirror;
                if (rd == 0) {   // rows [v_l | v_r] of length 2h: weighted left halves (compact), right halves in place
                    d.l_base = P.lw; d.l_stride = h;
                    d.r_base = lev[nv - 1 - n] + h; d.r_stride = 2 * h;
                } else {         // folded tables: table t at inb + t * 2 * half
                    const Fr* inb = (rd & 1) ? P.buf0 : P.buf1;
                    if (slot_layer && rd == slots->seg_shift) {   // the slot tables are one entry per segment pair now: back to one pair per memory
                        const Fr* prev = inb;
                        Fr* rg = slot_regroup;
                        const unsigned char* so = slots->d_slot_of;
                        const Fr* ra = slot_ratio;
                        const int g2 = (int)G2, np = slots->npairs;
                        own.pre.push_back({(size_t)rd, [prev, rg, so, ra, g2, np, st] { k_bn_gp_regroup<<<(unsigned)((g2 * np + 255) / 256), 256, 0, st>>>(prev, rg, so, ra, g2, np); }});
                        inb = slot_regroup;
                    }
                    d.l_base = inb; d.r_base = inb + 2 * half; d.l_stride = d.r_stride = 4 * half;
                }
                d.out = (rd & 1) ? P.buf1 : P.buf0;
                d.part = P.part + (size_t)rd * BN_PART_STRIDE * 3;
                d.r = fr_to_mont(chain[layers[n].r_at + rd]);
                fold_consts(d.r, &d.fk);
                d.half = half;
                d.nb = (nv - 2 - n >= 0 && nv - 2 - n < D1) ? slots->deep[nv - 2 - n].V : (int)nb;
                if (P.mirror) {
                    // sum_b w_b l_b r_b over reads and writes = (1 + kappa) [sum_reads w l r + K1 S + K2], kappa = gamma^(nb/2), c = *mirror_c:
                    // K1 = kappa c / (1 + kappa), K2 = kappa c^2 sum_{b < nb/2} gamma^b / (1 + kappa); the host applies 1 + kappa
                    const Fr g = fr_to_mont(chain[layers[n].gamma_at]);
                    Fr kappa = fr_one_mont(), lam = fr_zero();
                    for (size_t b = 0; b < G2; b++) { lam = fr_add(lam, kappa); kappa = fr_mul(kappa, g); }
                    const Fr onek = fr_add(fr_one_mont(), kappa);
                    if (!(onek.l[0] | onek.l[1] | onek.l[2] | onek.l[3])) throw Error("hg_grand_product_bn254: degenerate batching challenge");
                    const Fr kp = fr_mul(kappa, fr_inv(onek));
                    d.nb = slot_layer && rd < slots->seg_shift ? slots->V : (int)G2;
                    d.mirror = 1;
                    d.k1 = fr_mul(kp, *mirror_c);
                    d.k2 = fr_mul(fr_mul(kp, fr_mul(*mirror_c, *mirror_c)), lam);
                    d.s_in = rd == 0 ? P.S : ((rd & 1) ? P.sbuf0 : P.sbuf1);
                    d.s_out = (rd & 1) ? P.sbuf1 : P.sbuf0;
                }
                const RoundGrid g = round_grid_gp(half, d.nb, launch_wgs);
                d.gx = g.gx; d.gy = g.gy;
                if (half > (size_t)g.gx * BN_GP_J) d.acc = dalloc((size_t)g.blocks() * BN_TPB * 3);   // (the largest layers' first rounds only)
                reds[red_index[n]].n[rd] = g.blocks();
                own.by_rd[rd].push_back(d);
            }
        }
        for (int n = 1; n < nv; n++) {
            const LayerPlan& P = plan[n];
            const LayerRec& L = layers[n];
            RedJobDev& r = reds[red_index[n]];
            r.part = P.part; r.out = L.d_sums; r.nrounds = P.nmain;
            const Fr* last = ((P.nmain - 1) & 1) ? P.buf1 : P.buf0;   // output of the last shared round (not const: a slot-form layer regroups it)
            if (P.nmain < n) {
                TailJobDev t;
                memset(&t, 0, sizeof(t));
                if (nv - 2 - n >= 0 && nv - 2 - n < D1) {   // back to one pair per row ahead of the tail
                    const GpSlots::Deep& dp = slots->deep[nv - 2 - n];
                    const int tlen = (int)(((size_t)1 << n) >> P.nmain);   // table length after the shared rounds (>= ng entries)
                    int sh = 0;
                    while ((dp.ng << sh) < tlen) sh++;
                    if ((dp.ng << sh) != tlen) throw Error("grand_product_core: slot tables shorter than the segment groups");
                    Fr* rg = dalloc(ntab * (size_t)tlen);
                    const Fr* prev = last;
                    const unsigned char* so = dp.d_slot_of;
                    const Fr* ra = deep_ratio[nv - 2 - n];
                    const int nr = (int)nb, ng = dp.ng;
                    own.pre_tail.push_back([prev, rg, so, ra, nr, ng, tlen, sh, st] { k_bn_gp_regroup_tab<<<(unsigned)((nr * tlen + 255) / 256), 256, 0, st>>>(prev, rg, so, ra, nr, ng, tlen, sh); });
                    last = rg;
                }
                t.in = last; t.buf = P.tbuf; t.sums_out = L.d_sums + (size_t)P.nmain * 3; t.fin_out = L.d_final;
                for (int q = 0; q < BN_TAIL_ROUNDS; q++) t.rs.r[q] = q < n - P.nmain ? fr_to_mont(chain[L.r_at + P.nmain + q]) : fr_zero();
                t.npairs = (int)nb; t.half0 = (int)(((size_t)1 << n) >> (P.nmain + 1)); t.nrounds = n - P.nmain;
                tails.push_back(t);
            }
        }
        for (int n = 1; n < nv; n++)   // layers without a tail (n = 1): the folded values are the last shared round's output
            if (plan[n].nmain == n) {   // (a mirrored layer leaves the read rows' values only)
                const size_t cnt = plan[n].mirror ? 2 * G2 : ntab;
                const Fr* src = ((plan[n].nmain - 1) & 1) ? plan[n].buf1 : plan[n].buf0;
                Fr* dst = layers[n].d_final;
                own.posts.push_back([src, dst, cnt, st] { k_bn_copy_from_mont<<<(unsigned)((cnt + 255) / 256), 256, 0, st>>>(src, dst, cnt); });
            }
        if (set) set->merge(own);   // launched by the caller together with the other product's rounds (gp_launch_set)
        else gp_launch_set(ctx, st, own);
        // `defer`: the caller waits later (it has more to enqueue that does not depend on this grand product's results) and runs
        // the transcript replay then; the buffers stay until the caller rewinds the arena
        if (!defer) res_sync(ctx, st, "grand_product_bn254: sync");
    } catch (...) {
        ctx->arena_rewind(arena_mark);
        throw;
    }
    if (!defer) ctx->arena_rewind(arena_mark);
    // the final point is challenges only: the last layer's round challenges, then its mu
    point_canon.clear();
    if (nv > 1) for (int rd = 0; rd < nv - 1; rd++) point_canon.push_back(chain[layers[nv - 1].r_at + rd]);
    point_canon.push_back(chain[layers[nv - 1].mu_at]);
    const bool has_mirror = mirror_c != nullptr;
    const Fr mirror_val = mirror_c ? *mirror_c : fr_zero();
    // 1 / gamma_n per layer: a 254-bit exponentiation each, challenges only - computed now (the GPU is busy), not in the replay
    std::vector<Fr> ginvs(nv, fr_one_mont());
    for (int n = 1; n < nv; n++) ginvs[n] = fr_inv(fr_to_mont(chain[layers[n].gamma_at]));
    // transcript replay (reads the host-mapped result slots: valid after the synchronisation)
    auto replay = [layers, chain, ginvs, h_top, h_roots, nb, nv, has_mirror, mirror_val, &proof, &claims_canon] {
    const Fr* mirror_c = has_mirror ? &mirror_val : nullptr;
    proof.clear();
    std::vector<Fr> claims(nb), x;
    for (size_t b = 0; b < nb; b++) { write_be32(proof, h_roots[b]); claims[b] = fr_to_mont(h_roots[b]); }
    for (int n = 0; n < nv; n++) {
        const LayerRec& L = layers[n];
        std::vector<Fr> evals(2 * nb);  // Montgomery
        if (n == 0) {
            x.clear();
            for (size_t i = 0; i < 2 * nb; i++) evals[i] = fr_to_mont(h_top[i]);
        } else {
            const Fr g = fr_to_mont(chain[L.gamma_at]);
            Fr claim = fr_zero(), w = fr_one_mont();
            for (size_t b = 0; b < nb; b++) { claim = fr_add(claim, fr_mul(claims[b], w)); w = fr_mul(w, g); }  // prover.rs:281-286
            x.clear();
            const bool mirrored = mirror_c && n == nv - 1;
            Fr onek = fr_one_mont();
            if (mirrored) { Fr kappa = fr_one_mont(); for (size_t b = 0; b < nb / 2; b++) kappa = fr_mul(kappa, g); onek = fr_add(onek, kappa); }
            for (int rd = 0; rd < n; rd++) {
                Fr c[4];
                const Fr r = fr_to_mont(chain[L.r_at + rd]);
                Fr sums[3];
                for (int t = 0; t < 3; t++) sums[t] = mirrored ? fr_from_mont(fr_mul(fr_to_mont(L.sums[(size_t)rd * 3 + t]), onek)) : L.sums[(size_t)rd * 3 + t];
                claim = replay_round(sums, 3, claim, r, c);
                for (int k = 0; k < 4; k++) write_be32(proof, fr_from_mont(c[k]));
                x.push_back(chain[L.r_at + rd]);
            }
            const size_t held = mirrored ? nb / 2 : nb;   // rows whose evaluations the kernels produced
            for (size_t i = 0; i < 2 * held; i++) evals[i] = fr_to_mont(L.fin[i]);
            // the kernels leave the left evaluation of pair b multiplied by gamma^b (k_bn_gp_round_jobs)
            const Fr ginv = ginvs[n];
            Fr u = ginv;
            for (size_t b = 1; b < held; b++) { evals[2 * b] = fr_mul(evals[2 * b], u); u = fr_mul(u, ginv); }
            if (mirrored)   // folding is affine with coefficients summing to one: row + c stays row + c
                for (size_t b = 0; b < held; b++) { evals[2 * (held + b)] = fr_add(evals[2 * b], *mirror_c); evals[2 * (held + b) + 1] = fr_add(evals[2 * b + 1], *mirror_c); }
        }
        for (size_t i = 0; i < 2 * nb; i++) write_be32(proof, fr_from_mont(evals[i]));  // prover.rs:257
        const Fr mu = fr_to_mont(chain[L.mu_at]);                                          // prover.rs:259
        for (size_t b = 0; b < nb; b++) claims[b] = fr_add(evals[2 * b], fr_mul(mu, fr_sub(evals[2 * b + 1], evals[2 * b])));  // :288-294
        x.push_back(chain[L.mu_at]);
    }
    claims_canon.resize(nb);
    for (size_t b = 0; b < nb; b++) claims_canon[b] = fr_from_mont(claims[b]);
    };
    if (defer) *defer = replay;
    else replay();
}
void grand_product_bn254(hg_ctx* ctx, size_t nb, size_t len, const u64* const* tables, size_t chain_skip, std::vector<uint8_t>& proof,
                         u64* claims_out, u64* point_out) {
    std::vector<Fr> claims, x;
    grand_product_core(ctx, nb, len, tables, nullptr, chain_skip, proof, claims, x);
    for (size_t b = 0; b < nb; b++) memcpy(claims_out + 4 * b, claims[b].l, 32);
    for (size_t i = 0; i < x.size(); i++) memcpy(point_out + 4 * i, x[i].l, 32);
}

// ---- MLE evaluation and NTT over Fr (the other primitives of the path, A13/A14) -------------------------------------
// t'[j] = t[2j] + r (t[2j+1] - t[2j]): binds the lowest variable (fix_var order of the path)
__global__ void k_bn_fold(const Fr* __restrict__ in, Fr* __restrict__ out, size_t half, Fr r) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= half) return;
    const Fr x = in[2 * j], y = in[2 * j + 1];
    out[j] = fr_add(x, fr_mul(r, fr_sub(y, x)));
}
// m variables bound per launch (2 <= m <= 10): workgroup g folds the 2^m consecutive entries in[g 2^m ..] into out[g] - an MLE
// evaluation over 20 variables in two launches instead of twenty. to_slot: the single result is written canonical (result slot).
struct FoldPoint { Fr r[10]; };
__global__ __launch_bounds__(256) void k_bn_fold_multi(const Fr* __restrict__ in, Fr* __restrict__ out, int m, FoldPoint P, int to_slot) {
    __shared__ Fr sm[256];
    const int t = threadIdx.x;
    const int le = m > 8 ? m - 8 : 0, E = 1 << le;   // entries per thread, folded in registers first
    const int cnt0 = 1 << (m - le);                  // values entering the LDS tree
    const Fr* base = in + ((size_t)blockIdx.x << m);
    if (t < cnt0) {
        Fr v[4];
        for (int e = 0; e < E; e++) v[e] = base[(size_t)t * E + e];
        for (int l = 0; l < le; l++)
            for (int e = 0; e < (E >> (l + 1)); e++) v[e] = lz_add(v[2 * e], lz_mul(P.r[l], lz_sub(v[2 * e + 1], v[2 * e])));
        sm[t] = v[0];
    }
    int cnt = cnt0;
    for (int l = le; l < m; l++) {
        __syncthreads();
        Fr x = fr_zero(), y = fr_zero();
        if (t < cnt / 2) { x = sm[2 * t]; y = sm[2 * t + 1]; }
        __syncthreads();
        if (t < cnt / 2) sm[t] = lz_add(x, lz_mul(P.r[l], lz_sub(y, x)));
        cnt >>= 1;
    }
    if (t == 0) out[blockIdx.x] = to_slot ? fr_from_mont(sm[0]) : lz_canon(sm[0]);
}
// W[i] = w^i (Montgomery), i < n
__global__ void k_bn_powers(Fr* __restrict__ W, Fr w, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr r = fr_one_mont(), b = w;
    for (size_t e = i; e; e >>= 1) { if (e & 1) r = fr_mul(r, b); b = fr_mul(b, b); }
    W[i] = r;
}
__global__ void k_bn_bitrev(const Fr* __restrict__ in, Fr* __restrict__ out, int log2n, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t n = (size_t)1 << log2n, b = i >> log2n, k = i & (n - 1);
    size_t rv = 0;
    for (int q = 0; q < log2n; q++) rv |= ((k >> q) & 1) << (log2n - 1 - q);
    out[(b << log2n) + rv] = in[i];
}
// one radix-2 decimation-in-time stage (input bit-reversed): butterflies of span 2^s
__global__ void k_bn_ntt_stage(Fr* __restrict__ a, const Fr* __restrict__ W, int log2n, int s, size_t total_half) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_half) return;
    const size_t half_n = (size_t)1 << (log2n - 1);
    const size_t b = i / half_n, t = i % half_n;
    const size_t span = (size_t)1 << s, grp = t >> s, pos = t & (span - 1);
    const size_t i0 = (b << log2n) + (grp << (s + 1)) + pos, i1 = i0 + span;
    const Fr w = W[pos << (log2n - 1 - s)];
    const Fr u = a[i0], v = fr_mul(a[i1], w);
    a[i0] = fr_add(u, v);
    a[i1] = fr_sub(u, v);
}
// ---- four-step NTT with LDS-resident sub-transforms (2^8 <= N <= 2^16): N = N1 N2, the counterpart of k_ntt4_cols / k_ntt4_rows ----
//   X[k1 + N1 k2] = sum_{n2} w_N^(n2 k1) [ sum_{n1} x[N2 n1 + n2] w_N1^(n1 k1) ] w_N2^(n2 k2)
// columns kernel: BN_NTT_TILE consecutive n2 per workgroup, the N1-point transforms in LDS (decimation in frequency, result bit-
// reversed in the slow index), the twiddle w_N^(n2 k1) on the way out; rows kernel: BN_NTT_TILE consecutive k1, N2-point transforms,
// the 1/N of the inverse transform and the canonical form on the way out. Two launches and two passes over HBM per batch instead of
// a bit-reversal, log2 N radix-2 stage launches, a scaling pass and a copy. Loose arithmetic inside (bn254_lazy.hpp).
constexpr int BN_NTT_TILE = 4;
template <int STRIDE>
__device__ __forceinline__ void bn_lds_ntt_dif(Fr* tile, int m, const Fr* __restrict__ W, int wstep_log2) {
    // 2^m-point transform along the slow index of tile[pos * STRIDE + c]; twiddle w_M^j = W[j << wstep_log2]
    const int M = 1 << m;
    for (int s = m - 1; s >= 0; s--) {
        const int h = 1 << s;
        for (int q = threadIdx.x; q < (M / 2) * BN_NTT_TILE; q += blockDim.x) {
            const int c = q & (BN_NTT_TILE - 1), p = q / BN_NTT_TILE;
            const int j = p & (h - 1), a = ((p >> s) << (s + 1)) + j;
            const Fr x = tile[a * STRIDE + c], y = tile[(a + h) * STRIDE + c];
            tile[a * STRIDE + c] = lz_add(x, y);
            tile[(a + h) * STRIDE + c] = j ? lz_mul(lz_sub(x, y), W[((size_t)j << (m - 1 - s)) << wstep_log2]) : lz_subr(x, y);
        }
        __syncthreads();
    }
}
__device__ __forceinline__ u32 bn_brev_bits(u32 x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }
struct NttSrc { const Fr* p[64]; };   // per transform of the batch: where its input lies (null: in place, at in + b N)
__global__ __launch_bounds__(256) void k_bn_ntt4_cols(const Fr* __restrict__ in, Fr* __restrict__ out, int n, int n1, const Fr* __restrict__ W, NttSrc srcs) {
    extern __shared__ Fr bn_ntile[];
    const int n2 = n - n1;
    const size_t N = (size_t)1 << n, N2 = (size_t)1 << n2;
    const int N1 = 1 << n1;
    const size_t i2_0 = (size_t)blockIdx.x * BN_NTT_TILE;
    const Fr* x = blockIdx.y < 64 && srcs.p[blockIdx.y] ? srcs.p[blockIdx.y] : in + (size_t)blockIdx.y * N;
    Fr* y = out + (size_t)blockIdx.y * N;
    for (int idx = threadIdx.x; idx < N1 * BN_NTT_TILE; idx += blockDim.x) {
        const int i1 = idx / BN_NTT_TILE, c = idx & (BN_NTT_TILE - 1);
        bn_ntile[idx] = x[(size_t)i1 * N2 + i2_0 + c];
    }
    __syncthreads();
    bn_lds_ntt_dif<BN_NTT_TILE>(bn_ntile, n1, W, n2);   // w_N1 = w^(N2)
    for (int idx = threadIdx.x; idx < N1 * BN_NTT_TILE; idx += blockDim.x) {
        const int pos = idx / BN_NTT_TILE, c = idx & (BN_NTT_TILE - 1);
        const u32 k1 = bn_brev_bits((u32)pos, n1);
        const size_t i2 = i2_0 + c, e = (size_t)k1 * i2;   // e < N
        const Fr v = bn_ntile[idx];
        y[(size_t)k1 * N2 + i2] = e ? lz_mul(v, W[e]) : v;
    }
}
__global__ __launch_bounds__(256) void k_bn_ntt4_rows(const Fr* __restrict__ in, Fr* __restrict__ out, int n, int n1, const Fr* __restrict__ W, Fr scale, int scaled) {
    extern __shared__ Fr bn_ntile[];
    constexpr int ST = BN_NTT_TILE + 1;
    const int n2 = n - n1;
    const size_t N = (size_t)1 << n, N1 = (size_t)1 << n1;
    const int N2 = 1 << n2;
    const size_t k1_0 = (size_t)blockIdx.x * BN_NTT_TILE;
    const Fr* y = in + (size_t)blockIdx.y * N;
    Fr* X = out + (size_t)blockIdx.y * N;
    for (int idx = threadIdx.x; idx < N2 * BN_NTT_TILE; idx += blockDim.x) {
        const int r = idx >> n2, i2 = idx & (N2 - 1);
        bn_ntile[i2 * ST + r] = y[(k1_0 + r) * (size_t)N2 + i2];
    }
    __syncthreads();
    bn_lds_ntt_dif<ST>(bn_ntile, n2, W, n1);   // w_N2 = w^(N1)
    for (int idx = threadIdx.x; idx < N2 * BN_NTT_TILE; idx += blockDim.x) {
        const int pos = idx / BN_NTT_TILE, r = idx & (BN_NTT_TILE - 1);
        const u32 k2 = bn_brev_bits((u32)pos, n2);
        const Fr v = bn_ntile[pos * ST + r];
        X[k1_0 + r + N1 * (size_t)k2] = lz_canon(scaled ? lz_mul(v, scale) : v);   // the node tables are canonical (k_bn_gate_eval adds them)
    }
}
// in place on `a` through `tmp` (same size); W[i] = w^i for i < 2^log2n; returns false when the size is outside the four-step range
static bool ntt4_dev(hipStream_t st, Fr* a, Fr* tmp, const Fr* W, int log2n, bool scaled, Fr scale, size_t batch, const NttSrc* srcs = nullptr) {
    if (log2n < 8 || log2n > 16 || (srcs && batch > 64)) return false;
    NttSrc none;
    memset(&none, 0, sizeof(none));
    const int n1 = log2n / 2, n2 = log2n - n1;
    const size_t lds = (size_t)(1 << (n1 > n2 ? n1 : n2)) * (BN_NTT_TILE + 1) * sizeof(Fr);
    k_bn_ntt4_cols<<<dim3(1u << n2 >> 2, (unsigned)batch), 256, lds, st>>>(a, tmp, log2n, n1, W, srcs ? *srcs : none);
    k_bn_ntt4_rows<<<dim3(1u << n1 >> 2, (unsigned)batch), 256, lds, st>>>(tmp, a, log2n, n1, W, scale, scaled ? 1 : 0);
    return true;
}
__global__ void k_bn_scale(Fr* __restrict__ a, Fr c, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = fr_mul(a[i], c);
}

// 2^28-th root of unity of bn256::Fr = 7^((r-1)/2^28) (halo2curves ROOT_OF_UNITY, S = 28; checked: order exactly 2^28)
static Fr fr_root_of_unity(int log2n) {
    if (log2n > 28) throw Error("bn254: two-adicity is 28");
    Fr w = fr_to_mont(fr_make(0xd34f1ed960c37c9cULL, 0x3215cf6dd39329c8ULL, 0x98865ea93dd31f74ULL, 0x03ddb9f5166d18b7ULL));
    for (int i = log2n; i < 28; i++) w = fr_mul(w, w);
    return w;
}

// = BoxMultilinearPoly::evaluate over Fr [REF memory_checking/mod.rs:80-93, sk_encryption_circuit.rs:446]
void mle_eval_bn254(hg_ctx* ctx, const u64* table4, size_t nv, const u64* point4, u64* out4) {
    hipc(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t st = ctx->stream;
    const size_t N = (size_t)1 << nv;
    Fr *a = nullptr, *b = nullptr;
    hipc(hipMalloc((void**)&a, N * sizeof(Fr)), "hipMalloc");
    hipc(hipMalloc((void**)&b, std::max<size_t>(N / 2, 1) * sizeof(Fr)), "hipMalloc");
    hipError_t e = hipMemcpyAsync(a, table4, N * sizeof(Fr), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        k_bn_to_mont<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(a, N);
        Fr *cur = a, *nxt = b;
        for (size_t v = 0; v < nv; v++) {
            const size_t half = N >> (v + 1);
            const Fr r = fr_to_mont(fr_make(point4[4 * v], point4[4 * v + 1], point4[4 * v + 2], point4[4 * v + 3]));
            k_bn_fold<<<(unsigned)((half + 255) / 256), 256, 0, st>>>(cur, nxt, half, r);
            std::swap(cur, nxt);
        }
        k_bn_from_mont<<<1, 64, 0, st>>>(cur, 1);
        e = hipMemcpyAsync(out4, cur, sizeof(Fr), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    (void)hipFree(a); (void)hipFree(b);
    hipc(e, "hg_mle_eval_bn254");
}

// = FftNode evaluate over Fr: out[k] = sum_j in[j] w^(jk), w the 2^log2n-th root of unity (inverse: w^-1 and 1/n);
// natural order in and out [REF sk_encryption_circuit.rs:224,249,251]
void ntt_bn254(hg_ctx* ctx, const u64* in4, int log2n, bool inverse, size_t batch, u64* out4) {
    hipc(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t st = ctx->stream;
    const size_t n = (size_t)1 << log2n, total = n * batch;
    Fr w = fr_root_of_unity(log2n);
    if (inverse) w = fr_inv(w);
    Fr *a = nullptr, *b = nullptr, *W = nullptr;
    hipc(hipMalloc((void**)&a, total * sizeof(Fr)), "hipMalloc");
    hipc(hipMalloc((void**)&b, total * sizeof(Fr)), "hipMalloc");
    hipc(hipMalloc((void**)&W, n * sizeof(Fr)), "hipMalloc");
    hipError_t e = hipMemcpyAsync(a, in4, total * sizeof(Fr), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        k_bn_to_mont<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(a, total);
        k_bn_powers<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(W, w, n);
        const Fr ninv = inverse ? fr_inv(fr_to_mont(fr_make((u64)n, 0, 0, 0))) : fr_one_mont();
        if (ntt4_dev(st, a, b, W, log2n, inverse, ninv, batch)) std::swap(a, b);   // (result in `a`: the code below reads `b`)
        else {
        k_bn_bitrev<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(a, b, log2n, total);
        const size_t th = total / 2;
        for (int s = 0; s < log2n; s++) k_bn_ntt_stage<<<(unsigned)((th + 255) / 256), 256, 0, st>>>(b, W, log2n, s, th);
        if (inverse) k_bn_scale<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(b, ninv, total);
        }
        k_bn_from_mont<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(b, total);
        e = hipMemcpyAsync(out4, b, total * sizeof(Fr), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(W);
    hipc(e, "hg_ntt_bn254");
}

// (defined below, behind bn254_gkr.inc: it uses that file's eq-table job arrays)
static void lasso_prove_bn254_impl(hg_ctx* ctx, const hg_pk* pk, const u64* in4, const Fr* d_in_mont, size_t chain_skip, std::vector<uint8_t>& proof,
                                   u64* claim_out, const std::function<void()>* mid = nullptr, int mid_at = 0);
#include "bn254_gkr.inc"

// ---- LassoNode::prove_claim_reduction over Fr [REF lasso/src/lasso.rs:57-114] -------------------------------------------
// The limb split and the counters are integer work on the low limb (fe_to_bits_le truncates to at most 63 bits,
// lasso.rs:381-414, 654-669): the Goldilocks kernels (lasso_split, lasso_counters) are reused as they are; everything
// that involves challenges runs over Fr.
__global__ void k_bn_from_u64(const u64* __restrict__ in, Fr* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fr_to_mont(fr_make(in[i], 0, 0, 0));
}
// eq table by doubling: after step i the first 2^(i+1) entries hold eq(r_0..r_i, .)
__global__ void k_bn_eq_step(Fr* __restrict__ eq, size_t cur, Fr r) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cur) return;
    const Fr hi = fr_mul(eq[j], r);
    eq[j + cur] = hi;
    eq[j] = fr_sub(eq[j], hi);
}
// Integer tables meet field constants through "double Montgomery" constants: for a plain integer x and K2 = K R^2 mod r (the raw
// limbs of fr_to_mont(K~)), the Montgomery reduction of x * K2 is (x K) R, i.e. x K in Montgomery form - one short multiply-
// accumulate (wcol_mac_u64) per table entry and one reduction per hash / row instead of a conversion and a full product each.
struct MPow { Fr v[5]; };  // (M^i) R^2
// sum_k eq[k] * sum_i M^i E_{mems(lookup(k))[i]}[k] (lasso.rs:422-454, range.rs:184-195) -> per-workgroup partials
__global__ __launch_bounds__(BN_TPB) void k_bn_lasso_claim(dev::LassoDev L, const Fr* __restrict__ eq, const u64* __restrict__ e_polys, MPow mp2,
                                                           Fr* __restrict__ partials) {
    __shared__ Fr sm[BN_TPB];
    const size_t N = (size_t)1 << L.nu;
    WCol acc = wcol_zero();   // sum of eq~ * comb~ over this thread's rows (a handful)
    for (size_t k = (size_t)blockIdx.x * BN_TPB + threadIdx.x; k < L.rows; k += (size_t)gridDim.x * BN_TPB) {
        const int l = L.seg_lookup[k >> L.seg_shift];
        WCol c = wcol_zero();
        for (int i = 0; i < L.lookup_nmems[l]; i++) wcol_mac_u64(c, e_polys[(size_t)L.lookup_mems[l][i] * N + k], mp2.v[i]);
        wcol_mac(acc, eq[k], wcol_reduce(c));
    }
    Fr s = block_sum_fr(wcol_reduce(acc), sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
// The two tables of the collation sum-check (lasso.rs:271-279): g = E_0 * sum_m M^m E_m only needs E_0 and the weighted SUM of
// the E tables (folding is linear, the final evaluations are dropped, lasso.rs:97). out[0][k] = E_0[k], out[1][k] = sum_{m >= 0}
// M^m E_m[k] (the full weighted sum: g = out[0] * out[1] is then ONE product pair of the PRODSUM round kernel), both in Montgomery
// form; at most four memories are non-zero in a row.
struct BnColPow { Fr v[32]; };  // (M^m) R^2 (raw limbs of fr_to_mont(M^m)), see k_bn_lasso_claim
__global__ __launch_bounds__(BN_TPB) void k_bn_collation_tabs(dev::LassoDev L, const u64* __restrict__ e_polys, BnColPow P, Fr* __restrict__ out) {
    const size_t N = (size_t)1 << L.nu;
    for (size_t k = (size_t)blockIdx.x * BN_TPB + threadIdx.x; k < N; k += (size_t)gridDim.x * BN_TPB) {
        WCol c0 = wcol_zero(), c1 = wcol_zero();
        if (k < L.rows) {
            const u64 uses = L.lookup_uses[L.seg_lookup[k >> L.seg_shift]];
            if (uses & 1) { wcol_mac_u64(c0, e_polys[k], P.v[0]); wcol_mac_u64(c1, e_polys[k], P.v[0]); }
            for (int m = 1; m < L.alpha; m++)
                if ((uses >> m) & 1) wcol_mac_u64(c1, e_polys[(size_t)m * N + k], P.v[m]);
        }
        out[k] = wcol_reduce(c0);
        out[N + k] = wcol_reduce(c1);
    }
}
// sum_k eq[k] * t[k] for a table of small integers
__global__ __launch_bounds__(BN_TPB) void k_bn_dot_u64(const Fr* __restrict__ eq, const u64* __restrict__ t, size_t n, Fr* __restrict__ partials) {
    __shared__ Fr sm[BN_TPB];
    WCol acc = wcol_zero();
    for (size_t k = (size_t)blockIdx.x * BN_TPB + threadIdx.x; k < n; k += (size_t)gridDim.x * BN_TPB) wcol_mac_u64(acc, t[k], eq[k]);
    // reduce(sum t eq~) = sum t eq as a plain residue: back to Montgomery form once per thread
    Fr s = block_sum_fr(fr_to_mont(wcol_reduce(acc)), sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
// the same for up to 4 integer tables sharing one eq table per thread (grid.y = group of 4): eq, 32 bytes per entry against 8 per
// table, is read once per group instead of once per table; one column accumulator per table (more would cost the occupancy).
// partials[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + t]
constexpr int BN_DOT_GROUP = 4;
struct DotU64Tabs { const u64* t[64]; };
__global__ __launch_bounds__(BN_TPB) void k_bn_dot_u64_multi(const Fr* __restrict__ eq, DotU64Tabs tabs, int ntab_all, size_t n, Fr* __restrict__ partials) {
    __shared__ Fr sm[BN_TPB];
    const int t0 = blockIdx.y * BN_DOT_GROUP, ntab = min(BN_DOT_GROUP, ntab_all - t0);
    WCol acc[BN_DOT_GROUP];
#pragma unroll
    for (int t = 0; t < BN_DOT_GROUP; t++) acc[t] = wcol_zero();
    for (size_t k = (size_t)blockIdx.x * BN_TPB + threadIdx.x; k < n; k += (size_t)gridDim.x * BN_TPB) {
        const Fr e = eq[k];
#pragma unroll
        for (int t = 0; t < BN_DOT_GROUP; t++)
            if (t < ntab) wcol_mac_u64(acc[t], tabs.t[t0 + t][k], e);
    }
#pragma unroll
    for (int t = 0; t < BN_DOT_GROUP; t++)
        if (t < ntab) {
            Fr s = block_sum_fr(fr_to_mont(wcol_reduce(acc[t])), sm);   // plain residue -> Montgomery form once per thread (as k_bn_dot_u64)
            if (threadIdx.x == 0) partials[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * BN_DOT_GROUP + t] = s;
        }
}
// out_slot[t] <- sum over the workgroups of table t (canonical, for the host)
struct DotU64Out { Fr* out[64]; };
__global__ __launch_bounds__(BN_TPB) void k_bn_dot_u64_reduce(const Fr* __restrict__ partials, int nblocks, DotU64Out outs) {
    __shared__ Fr sm[BN_TPB];
    const int t = blockIdx.x, grp = t / BN_DOT_GROUP, q = t % BN_DOT_GROUP;
    Fr a = fr_zero();
    for (int b = threadIdx.x; b < nblocks; b += BN_TPB) a = fr_add(a, partials[((size_t)grp * nblocks + b) * BN_DOT_GROUP + q]);
    a = block_sum_fr(a, sm);
    if (threadIdx.x == 0) *outs.out[t] = fr_from_mont(a);
}
// The openings of the E tables at x: a workgroup walks `chunk` consecutive rows, all of one lookup segment, where only the (at most
// four) memories that lookup uses have non-zero entries - one multiply-accumulate per row and USED memory, eq (32 bytes per entry)
// read once for all of them, instead of one per row and memory in groups of four tables (k_bn_dot_u64_multi: 25 tables = 7 passes
// over eq). partials[bx * 32 + m]; memories the segment does not use get zero.
struct BnOpenE { const u64* ep; size_t N, rows; const uint8_t* seg_lookup; int seg_shift, nE; int nmems[32]; int mems[32][4]; int mem[32]; Fr* out[32]; };
__global__ __launch_bounds__(BN_TPB) void k_bn_open_e(const Fr* __restrict__ eq, BnOpenE O, size_t chunk, Fr* __restrict__ partials) {
    __shared__ Fr sm[BN_TPB];
    const size_t row0 = (size_t)blockIdx.x * chunk;
    const size_t row1 = row0 + chunk < O.N ? row0 + chunk : O.N;
    int nact = 0, act[4] = {0, 0, 0, 0};
    if (row0 < O.rows) {
        const int l = O.seg_lookup[row0 >> O.seg_shift];
        nact = O.nmems[l];
        for (int k = 0; k < 4; k++) act[k] = O.mems[l][k];
    }
    if (threadIdx.x < 32) {   // memories this segment does not touch
        bool used = false;
        for (int k = 0; k < 4; k++) used = used || (k < nact && act[k] == (int)threadIdx.x);
        if (!used) partials[(size_t)blockIdx.x * 32 + threadIdx.x] = fr_zero();
    }
    if (!nact) return;
    WCol acc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) acc[k] = wcol_zero();
    const size_t rend = row1 < O.rows ? row1 : O.rows;
    for (size_t j = row0 + threadIdx.x; j < rend; j += BN_TPB) {
        const Fr e = eq[j];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k < nact) wcol_mac_u64(acc[k], O.ep[(size_t)act[k] * O.N + j], e);
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (k < nact) {   // (nact is uniform over the workgroup)
            const Fr s = block_sum_fr(fr_to_mont(wcol_reduce(acc[k])), sm);
            if (threadIdx.x == 0) partials[(size_t)blockIdx.x * 32 + act[k]] = s;
        }
}
__global__ __launch_bounds__(BN_TPB) void k_bn_open_e_reduce(const Fr* __restrict__ partials, int nblocks, BnOpenE O) {
    __shared__ Fr sm[BN_TPB];
    const int m = O.mem[blockIdx.x];
    Fr a = fr_zero();
    for (int b = threadIdx.x; b < nblocks; b += BN_TPB) a = fr_add(a, partials[(size_t)b * 32 + m]);
    a = block_sum_fr(a, sm);
    if (threadIdx.x == 0) *O.out[blockIdx.x] = fr_from_mont(a);
}
// h(a,v,t) = a + v gamma + t gamma^2 - tau (prover.rs:44) for the reads (t) and writes (t + 1) of one memory
struct HashK { Fr one2, gamma2x, gammasq2x, gammasq, tau;   // R^2, gamma R^2, gamma^2 R^2 (raw), gamma^2 and tau (Montgomery)
               u32 kc[32]; };                              // limbs of R, gamma R, gamma^2 R, p - tau R: fr_lin3_const (bn254_wide.hpp)
static void hashk_consts(HashK& K, const Fr& gamma_mont, const Fr& gamma2_mont, const Fr& tau_mont) {
    const Fr c[4] = {fr_one_mont(), gamma_mont, gamma2_mont, fr_sub(fr_zero(), tau_mont)};
    for (int q = 0; q < 4; q++) for (int i = 0; i < 4; i++) { K.kc[8 * q + 2 * i] = (u32)c[q].l[i]; K.kc[8 * q + 2 * i + 1] = (u32)(c[q].l[i] >> 32); }
}
// all memories of the node in one launch (blockIdx.y = position in the memory-GKR order): the read (and write) hash rows, and for
// blockIdx.x beyond the rows' blocks the init / final rows of the same memory (k_bn_hash_if's work)
struct HashMem { const u64* dim; const u64* ep; const u64* ts; const u64* fc; Fr* rd; Fr* wr; Fr* init; Fr* fin; u32 cutoff, pad; };
struct HashMems { HashMem m[32]; };
__device__ __forceinline__ void bn_hash_if_entry(u32 a, u32 cutoff, const u64* __restrict__ fc, const HashK& K, Fr* __restrict__ init, Fr* __restrict__ fin);
// Slot form (GpSlotsDev, see grand_product_core): blockIdx.y = slot, the row written is slot_rows + slot * n and its entries in row
// segment s come from the memory that represents the slot there (rep); the init / final rows are then a launch of their own
// (row_blocks = 0: every workgroup is an init / final one).
struct HashSlots { const unsigned char* rep; Fr* slot_rows; int npairs, seg_shift; };
__global__ __launch_bounds__(256) void k_bn_hash_all(size_t n, HashMems M, HashK K, size_t row_blocks, HashSlots SL) {
    const size_t nblk = row_blocks;
    if (blockIdx.x >= nblk) {   // (uniform) the 2^16-entry init / final rows
        const HashMem& mi = M.m[blockIdx.y];
        bn_hash_if_entry((u32)((blockIdx.x - nblk) * 256 + threadIdx.x), mi.cutoff, mi.fc, K, mi.init, mi.fin);
        return;
    }
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    int pos = blockIdx.y;
    Fr* rd_row = nullptr;
    if (SL.rep) {   // (uniform over the workgroup: 256 consecutive rows lie in one segment)
        const size_t sp = (j >> SL.seg_shift) % (size_t)SL.npairs;
        pos = SL.rep[(size_t)blockIdx.y * SL.npairs + sp];
        rd_row = SL.slot_rows + (size_t)blockIdx.y * n;
    }
    const HashMem& m = M.m[pos];
    if (!rd_row) rd_row = m.rd;
    const u64 a = m.dim[j], v = m.ep[j], t = m.ts[j];
    Fr h;   // loose (bn254_lazy.hpp): read by the product-tree and round kernels
    if (((a | v | t) >> 32) == 0) h = lz_lin3((u32)a, (u32)v, (u32)t, K.kc);   // addresses, limb values, counters: always
    else {
        WCol w = wcol_zero();
        wcol_mac_u64(w, a, K.one2);
        wcol_mac_u64(w, v, K.gamma2x);
        wcol_mac_u64(w, t, K.gammasq2x);
        h = fr_sub(wcol_reduce(w), K.tau);
    }
    lz_gstore(&rd_row[j], h);
    if (m.wr && !SL.rep) lz_gstore(&m.wr[j], lz_add(h, K.gammasq));   // (null: the write rows are not materialised, see grand_product_core's mirror_c)
}
__global__ void k_bn_hash_rw(size_t n, const u64* __restrict__ dim, const u64* __restrict__ ep, const u64* __restrict__ ts, HashK K,
                             Fr* __restrict__ rd, Fr* __restrict__ wr) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const u64 a = dim[j], v = ep[j], t = ts[j];
    Fr h;   // loose (bn254_lazy.hpp): read by the product-tree and round kernels
    if (((a | v | t) >> 32) == 0) h = lz_lin3((u32)a, (u32)v, (u32)t, K.kc);   // addresses, limb values, counters: always
    else {
        WCol w = wcol_zero();
        wcol_mac_u64(w, a, K.one2);
        wcol_mac_u64(w, v, K.gamma2x);
        wcol_mac_u64(w, t, K.gammasq2x);
        h = fr_sub(wcol_reduce(w), K.tau);
    }
    rd[j] = h;
    if (wr) wr[j] = lz_add(h, K.gammasq);   // (null: the write rows are not materialised, see grand_product_core's mirror_c)
}
__device__ __forceinline__ void bn_hash_if_entry(u32 a, u32 cutoff, const u64* __restrict__ fc, const HashK& K, Fr* __restrict__ init, Fr* __restrict__ fin) {
    if (a >= 65536) return;
    const u64 f = fc[a];
    const u32 tv = a < cutoff ? a : 0u;
    init[a] = fr_lin3_const(a, tv, 0u, K.kc);
    if ((f >> 32) == 0) fin[a] = fr_lin3_const(a, tv, (u32)f, K.kc);
    else {
        WCol w2 = wcol_zero();
        wcol_mac_u64(w2, a, K.one2);
        if (a < cutoff) wcol_mac_u64(w2, a, K.gamma2x);
        wcol_mac_u64(w2, f, K.gammasq2x);
        fin[a] = fr_sub(wcol_reduce(w2), K.tau);
    }
}
__global__ void k_bn_hash_if(u32 cutoff, const u64* __restrict__ fc, HashK K, Fr* __restrict__ init, Fr* __restrict__ fin) {
    bn_hash_if_entry(blockIdx.x * blockDim.x + threadIdx.x, cutoff, fc, K, init, fin);
}

// low limbs of a Montgomery-form table; *bad is set when an element does not fit one limb
__global__ void k_bn_low_limb(const Fr* __restrict__ in, u64* __restrict__ out, size_t n, int* __restrict__ bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr v = fr_from_mont(in[i]);
    out[i] = v.l[0];
    if (v.l[1] | v.l[2] | v.l[3]) atomicOr(bad, 1);
}
// in4: host table (4 canonical limbs per element), or null with d_in_mont: the node input as it lies in HBM (Montgomery form)
// the flag of k_bn_low_limb into a result slot (plain store into host-mapped memory; no blocking copy of a device word)
__global__ void k_bn_flag_out(const int* __restrict__ flag, Fr* __restrict__ slot) { *slot = fr_make((u64)*flag, 0, 0, 0); }
// mid (optional): called once while the node's launches are being enqueued, at stage `mid_at` (1: behind the limb split and the
// counters, 2: behind the hashes and before the grand products, 3: behind everything) - the caller enqueues independent work on
// another stream there, so that the device has both chains queued from the start of the prove (bn254_gkr.inc: BnProver::run)
static void lasso_prove_bn254_impl(hg_ctx* ctx, const hg_pk* pk, const u64* in4, const Fr* d_in_mont, size_t chain_skip, std::vector<uint8_t>& proof,
                                   u64* claim_out, const std::function<void()>* mid, int mid_at) {
    if (!pk->ctx) throw Error("hg_lasso_prove_bn254: host-only prover key");
    const LassoPlan& lp = pk->lasso;
    const dev::LassoDev& L = pk->lasso_dev;
    const int nu = lp.nu, A = lp.alpha;
    const size_t N = (size_t)1 << nu, M = 65536;
    hipc(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t st = ctx->stream;
    // the node input must be small non-negative integers (range-shifted values): only the low limb takes part in the split
    std::vector<u64> low;
    if (in4) {
        low.resize(N);
        for (size_t j = 0; j < N; j++) {
            if (in4[4 * j + 1] | in4[4 * j + 2] | in4[4 * j + 3]) throw Error("hg_lasso_prove_bn254: input " + std::to_string(j) + " is not below 2^64 (not a range-shifted value)");
            low[j] = in4[4 * j];
        }
    }
    const Fr* h_bad = nullptr;
    const size_t r_at = chain_skip, col_at = r_at + nu, gamma_at = col_at + nu, tau_at = gamma_at + 1, gp1_at = tau_at + 1,
                 gp2_at = gp1_at + gp_challenges(nu), total = gp2_at + gp_challenges(16);
    const std::vector<Fr> chain = challenge_chain_bn254(total);
    const std::vector<size_t> arena_mark = ctx->arena_mark();
    auto dalloc_b = [&](size_t bytes) { return ctx->alloc(std::max<size_t>(bytes, 16)); };
    auto dalloc = [&](size_t n_fr) { return (Fr*)dalloc_b(n_fr * sizeof(Fr)); };
    auto grid1 = [](size_t n) { return (unsigned)((n + 255) / 256); };
    std::vector<uint8_t> gp1_bytes, gp2_bytes;
    std::vector<Fr> x, y, tmp_claims, tmp_claims2, h_col((size_t)nu * 2), opens;
    std::function<void()> replay_gp1, replay_gp2;
    Fr h_claimed;
    int col_nmain = 0;
    try {
        // polynomialize (lasso.rs:157-250): integer kernels of the Goldilocks path
        u64* d_in = (u64*)dalloc_b(N * 8);
        int* d_bad = (int*)dalloc_b(sizeof(int));
        if (in4) hipc(hipMemcpyAsync(d_in, low.data(), N * 8, hipMemcpyHostToDevice, st), "upload input");
        else {
            hipc(hipMemsetAsync(d_bad, 0, sizeof(int), st), "clear flag");
            k_bn_low_limb<<<grid1(N), 256, 0, st>>>(d_in_mont, d_in, N, d_bad);
            const ResRef fl = res_slots(ctx, 1);
            k_bn_flag_out<<<1, 1, 0, st>>>(d_bad, fl.dev);
            h_bad = fl.host;
        }
        u64* dims = (u64*)dalloc_b(4 * N * 8);
        u64* ep = (u64*)dalloc_b((size_t)A * N * 8);
        dev::lasso_split(st, L, d_in, dims, ep, dev::ep_rows_all(L.alpha));
        std::map<int, u64*> read_ts, final_cts;
        {   // all counter chunks in one stable sort over (chunk, address) keys (as the Goldilocks prover: kernels.hip)
            unsigned mask = 0;
            dev::CounterOut co;
            memset(&co, 0, sizeof(co));
            for (auto& chk : lp.chunks) {
                const int c = chk.first;
                if (c < 0 || c >= 4) throw Error("hg_lasso_prove_bn254: chunk index out of range");
                mask |= 1u << c;
                read_ts[c] = (u64*)dalloc_b(N * 8);
                final_cts[c] = (u64*)dalloc_b(M * 8);
                co.read_ts[c] = read_ts[c]; co.final_cts[c] = final_cts[c];
            }
            const size_t elems = std::max<size_t>(dev::lasso_counters_all_elems(L, mask), 1);
            const size_t tb = dev::lasso_counters_all_temp_bytes(elems);
            void* temp = dalloc_b(tb);
            u32* keys = (u32*)dalloc_b(elems * 4); u32* keys2 = (u32*)dalloc_b(elems * 4);
            u32* rows = (u32*)dalloc_b(elems * 4); u32* rows2 = (u32*)dalloc_b(elems * 4);
            u32* starts = (u32*)dalloc_b((4 * 65536 + 1) * 4);
            dev::lasso_counters_all(st, L, mask, dims, co, temp, tb, keys, keys2, rows, rows2, starts);
        }
        if (mid && mid_at == 1) (*mid)();
        Fr* d_part = dalloc(1024 * 3);
        const ResRef r_claimed = res_slots(ctx, 1), r_col = res_slots(ctx, (size_t)nu * 2);
        // The three eq tables of the node - at r (claimed sum), at the grand products' final points x and y (openings) - are challenges
        // only (a grand product's point is the run of its last layer's round challenges and mu): built now, in three launches for all
        // of them (heads, then two levels of outer products) instead of thirteen spread over the node.
        Fr* eq = dalloc(N);
        Fr* eqx = dalloc(N);
        Fr* eqy = dalloc(M);
        auto final_point_at = [](size_t gp_at, int nv) { size_t pos = gp_at + 1; for (int n = 1; n < nv - 1; n++) pos += 2 + n; return nv > 1 ? pos + 1 : gp_at; };
        {
            DevPool eq_pool(ctx);
            std::vector<std::shared_ptr<void>> keep;
            EqPlan plan;
            plan.add(eq, &chain[r_at], nu, dalloc(eq_scratch_len(nu)));
            plan.add(eqx, &chain[final_point_at(gp1_at, nu)], nu, dalloc(eq_scratch_len(nu)));
            plan.add(eqy, &chain[final_point_at(gp2_at, 16)], 16, dalloc(eq_scratch_len(16)));
            eq_plan_flush(st, eq_pool, plan, keep);
            eq_pool.marks.clear();   // (the arena is rewound by this function, not by the pool)
        }
        // The claimed sum and the collation sum-check (0.9 ms of mostly launch-bound rounds nothing but the host reads) go to the SECOND
        // stream, behind the node reductions already enqueued there (round 6; HG_BN_COL_MAIN=1: on the main stream as before): the hashes,
        // the product trees and the grand products - 4.7 ms of HBM-bound launches - then start right behind the counters instead of
        // behind those rounds, and the launch-bound work of the other stream runs under them. The main stream waits for the second one
        // before the synchronisation at the end of the node.
        static const bool col_main = hg_env_on("HG_BN_COL_MAIN");
        hipStream_t st_col = (col_main || !ctx->stream2 || ctx->stream2 == st) ? st : ctx->stream2;
        if (st_col != st) {
            if (!ctx->bn_ev[0]) for (auto& e : ctx->bn_ev) hipc(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
            bn_flush(ctx, st);   // (descriptors staged so far belong to launches of the main stream)
            hipc(hipEventRecord(ctx->bn_ev[0], st), "event record");   // limb split, E tables, eq tables: what the claim and the collation tables read
            hipc(hipStreamWaitEvent(st_col, ctx->bn_ev[0], 0), "stream wait");
        }
        // r, claimed sum (lasso.rs:85, 264-269)
        MPow mp;
        {
            const Fr m = fr_small(M);
            Fr pwr = fr_one_mont();
            for (int i = 0; i < 5; i++) { mp.v[i] = fr_to_mont(pwr); pwr = fr_mul(pwr, m); }   // (M^i) R^2: see k_bn_lasso_claim
        }
        {
            const int grid = (int)std::min<size_t>((L.rows + BN_TPB - 1) / BN_TPB, 1024);
            k_bn_lasso_claim<<<grid, BN_TPB, 0, st_col>>>(L, eq, ep, mp, d_part);
            k_bn_reduce<<<1, BN_TPB, 0, st_col>>>(d_part, grid, 1, r_claimed.dev);
        }
        // collation sum-check (lasso.rs:271-279): g = poly(0) * sum_i M^i poly(i) - on two tables, E_0 and the weighted sum of the others
        {
            Fr* tabs = dalloc(2 * N);
            BnColPow cp;
            {
                const Fr m = fr_small(M);
                Fr pwr = fr_one_mont();
                for (int i = 0; i < 32; i++) { cp.v[i] = i < A ? fr_to_mont(pwr) : fr_zero(); pwr = fr_mul(pwr, m); }
            }
            k_bn_collation_tabs<<<(unsigned)std::min<size_t>((N + BN_TPB - 1) / BN_TPB, 4096), BN_TPB, 0, st_col>>>(L, ep, cp, tabs);
            // g = tabs[0] * tabs[1]: a PRODSUM sum-check with one pair (k_bn_ps_round_jobs for the long rounds, one tail workgroup for
            // the rest); the round slots hold (g(0), g(2))
            if (nu > 32) throw Error("hg_lasso_prove_bn254: more than 32 rounds");
            col_nmain = ps_nmain(nu);
            Fr* buf0 = dalloc(N);
            Fr* buf1 = dalloc(std::max<size_t>(N / 2, 1));
            Fr* part = dalloc((size_t)nu * BN_PART_STRIDE * 2);
            Fr* tbuf = dalloc(2 * 2 * (size_t)BN_TAIL_HALF);
            Fr* fin_dummy = dalloc(2);
            std::vector<PsJobDev> descs(col_nmain);
            RedJobDev red;
            memset(&red, 0, sizeof(red));
            std::vector<int> blocks(col_nmain);
            for (int rd = 0; rd < col_nmain; rd++) {
                const size_t half = N >> (rd + 1);
                PsJobDev& d = descs[rd];
                memset(&d, 0, sizeof(d));
                const Fr* inb = (rd & 1) ? buf0 : buf1;
                d.t[0] = rd == 0 ? tabs : inb;
                d.t[1] = rd == 0 ? tabs + N : inb + 2 * half;
                d.out = (rd & 1) ? buf1 : buf0;
                d.part = part + (size_t)rd * BN_PART_STRIDE * 2;
                d.r = fr_to_mont(chain[col_at + rd]);
                fold_consts(d.r, &d.fk);
                d.half = half; d.npairs = 1;
                const RoundGrid g = round_grid(half, 1);
                d.gx = g.gx; d.gy = g.gy;
                red.n[rd] = blocks[rd] = g.blocks();
            }
            red.part = part; red.out = r_col.dev; red.nrounds = col_nmain;
            TailJobDev tl;
            memset(&tl, 0, sizeof(tl));
            tl.in = ((col_nmain - 1) & 1) ? buf1 : buf0;
            tl.buf = tbuf; tl.sums_out = r_col.dev + (size_t)col_nmain * 2; tl.fin_out = fin_dummy;
            for (int q = 0; q < BN_TAIL_ROUNDS; q++) tl.rs.r[q] = q < nu - col_nmain ? fr_to_mont(chain[col_at + col_nmain + q]) : fr_zero();
            tl.npairs = 1; tl.half0 = (int)(N >> (col_nmain + 1)); tl.nrounds = nu - col_nmain;
            {   // the folds' matrices (bn254_mfma.hpp), one per round
                MfA* mfa = static_cast<MfA*>(dalloc_b(descs.size() * sizeof(MfA)));
                for (size_t i = 0; i < descs.size(); i++) descs[i].mf = mfa + i;
            }
            const PsJobDev* d_descs = bn_stage(ctx, descs.data(), descs.size());
            const RedJobDev* d_red = bn_stage(ctx, &red, 1);
            const TailJobDev* d_tl = bn_stage(ctx, &tl, 1);
            bn_flush(ctx, st_col);
            if (!descs.empty()) k_bn_mf_consts<PsJobDev><<<(unsigned)descs.size(), 64, 0, st_col>>>(d_descs);
            for (int rd = 0; rd < col_nmain; rd++) k_bn_ps_round_jobs<<<dim3(2 * ((blocks[rd] + 7) / 8 * 8), 1, 1), BN_TPB, 0, st_col>>>(d_descs + rd, nullptr);
            k_bn_reduce_jobs<<<dim3(32, 1), BN_TPB, 0, st_col>>>(d_red, 2);
            k_bn_tail_jobs<BN_PRODSUM><<<1, 2 * BN_TPB, 0, st_col>>>(d_tl);
            if (st_col != st) hipc(hipEventRecord(ctx->bn_ev[1], st_col), "event record");
        }
        // MemoryCheckingProver::new (prover.rs:35-89): gamma, tau are the challenges themselves (E = F)
        const Fr gamma = fr_to_mont(chain[gamma_at]), tau = fr_to_mont(chain[tau_at]), gamma2 = fr_mul(gamma, gamma);
        HashK HK;
        HK.one2 = fr_r2(); HK.gamma2x = fr_to_mont(gamma); HK.gammasq2x = fr_to_mont(gamma2); HK.gammasq = gamma2; HK.tau = tau;
        hashk_consts(HK, gamma, gamma2, tau);
        const int G = (int)lp.gkr_order.size();
        constexpr bool use_mirror = true;
        const bool mirror = use_mirror && nu >= 2;   // write hash = read hash + gamma^2: only the read rows exist
        // Slot form of the read rows (GpSlots above): joint classes of the memories per segment pair (s, s + npairs)
        GpSlots slots;
        // HG_BN_SLOT_DEPTH = number of slot-form layers (default 4; 0 = every memory's own rows)
        static const int depth = [] { const char* e = getenv("HG_BN_SLOT_DEPTH"); return e && *e ? atoi(e) : 4; }();
        const bool no_slots = depth <= 0;
        if (mirror && !no_slots && G <= 32 && L.seg_shift >= 1 && nu - 1 > L.seg_shift && ((N / 2) >> L.seg_shift) >= 1 && ((N / 2) >> L.seg_shift) <= 64) {
            const int NP = (int)((N / 2) >> L.seg_shift);
            auto cls = [&](int i, int s) -> int {   // class of GKR position i in row segment s: itself when its memory is looked up there, else its chunk position
                const size_t row0 = (size_t)s << L.seg_shift;
                if (row0 < L.rows) {
                    const int l = lp.seg_lookup[s];
                    if ((L.lookup_uses[l] >> lp.gkr_order[i]) & 1) return 1000 + i;
                }
                return lp.gkr_chunk[i];
            };
            slots.npairs = NP; slots.seg_shift = L.seg_shift; slots.G2 = G;
            slots.slot_of.assign((size_t)G * NP, 0);
            std::vector<std::vector<int>> reps(NP);
            int V = 0;
            for (int sp = 0; sp < NP; sp++) {
                std::vector<std::pair<int, int>> keys;   // slot -> key
                for (int i = 0; i < G; i++) {
                    const std::pair<int, int> key = i == 0 ? std::make_pair(-1, -1) : std::make_pair(cls(i, sp), cls(i, sp + NP));
                    int v = -1;
                    for (size_t q = 0; q < keys.size(); q++) if (keys[q] == key) v = (int)q;
                    if (v < 0) { v = (int)keys.size(); keys.push_back(key); reps[sp].push_back(i); }
                    slots.slot_of[(size_t)i * NP + sp] = (unsigned char)v;
                }
                V = std::max(V, (int)keys.size());
            }
            slots.V = V;
            slots.rep.assign((size_t)V * NP, 0);
            for (int sp = 0; sp < NP; sp++) for (size_t v = 0; v < reps[sp].size(); v++) slots.rep[v * NP + sp] = (unsigned char)reps[sp][v];
            if (V < G) {   // (nothing to gain otherwise)
                slots.d_slot_of = bn_stage(ctx, slots.slot_of.data(), slots.slot_of.size());
                slots.d_rep = bn_stage(ctx, slots.rep.data(), slots.rep.size());
                // the layers below: layer q + 1 multiplies 2^(q+2) segments NP >> (q+1) apart; read and write rows apart, row 0 alone
                for (int q = 0; q + 2 <= depth && (NP >> (q + 1)) >= 2 && 2 * G <= 254; q++) {
                    GpSlots::Deep dp;
                    const int NG = NP >> (q + 1), cnt = 4 << q;
                    dp.ng = NG;
                    dp.slot_of.assign((size_t)2 * G * NG, 0);
                    std::vector<std::vector<int>> reps1(NG);
                    for (int u = 0; u < NG; u++) {
                        std::vector<std::vector<int>> keys;
                        for (int b = 0; b < 2 * G; b++) {
                            std::vector<int> key;
                            if (b == 0) key.push_back(-1);
                            else { key.push_back(b >= G ? 1 : 0); for (int t = 0; t < cnt; t++) key.push_back(cls(b % G, u + t * NG)); }
                            int v = -1;
                            for (size_t w = 0; w < keys.size(); w++) if (keys[w] == key) v = (int)w;
                            if (v < 0) { v = (int)keys.size(); keys.push_back(key); reps1[u].push_back(b); }
                            dp.slot_of[(size_t)b * NG + u] = (unsigned char)v;
                        }
                        dp.V = std::max(dp.V, (int)keys.size());
                    }
                    if (dp.V >= 2 * G) break;
                    dp.rep.assign((size_t)dp.V * NG, 255);
                    for (int u = 0; u < NG; u++) for (size_t v = 0; v < reps1[u].size(); v++) dp.rep[v * NG + u] = (unsigned char)reps1[u][v];
                    dp.d_slot_of = bn_stage(ctx, dp.slot_of.data(), dp.slot_of.size());
                    dp.d_rep = bn_stage(ctx, dp.rep.data(), dp.rep.size());
                    slots.deep.push_back(std::move(dp));
                }
                bn_flush(ctx, st);
            } else slots.V = 0;
        }
        const bool use_slots = slots.V > 0;
        if (hg_times("bn")) fprintf(stderr, "[hg bn]   read rows: %d slot rows for %d memories, %d segment pairs of 2^%d rows; %d slot-form layers below\n", use_slots ? slots.V : G, G, slots.npairs, slots.seg_shift, (int)slots.deep.size());
        Fr* H1 = dalloc((size_t)(use_slots ? slots.V : (mirror ? G : 2 * G)) * N);
        Fr* H2 = dalloc((size_t)2 * G * M);
        {   // every memory's read (write) hash rows and init / final rows in one launch
            if (G > 32) throw Error("hg_lasso_prove_bn254: more than 32 memories");
            HashMems HM;
            memset(&HM, 0, sizeof(HM));
            for (int i = 0; i < G; i++) {
                const int m = lp.gkr_order[i], c = lp.gkr_chunk[i];
                HashMem& h = HM.m[i];
                h.dim = dims + (size_t)c * N; h.ep = ep + (size_t)m * N; h.ts = read_ts[c]; h.fc = final_cts[c];
                h.rd = use_slots ? nullptr : H1 + (size_t)i * N; h.wr = mirror ? nullptr : H1 + (size_t)(G + i) * N;
                h.init = H2 + (size_t)i * M; h.fin = H2 + (size_t)(G + i) * M; h.cutoff = (u32)lp.mems[m].cutoff;
            }
            HashSlots HS;
            memset(&HS, 0, sizeof(HS));
            if (use_slots) {
                HS.rep = slots.d_rep; HS.slot_rows = H1; HS.npairs = slots.npairs; HS.seg_shift = slots.seg_shift;
                k_bn_hash_all<<<dim3(grid1(N), (unsigned)slots.V), 256, 0, st>>>(N, HM, HK, (size_t)grid1(N), HS);   // the slot rows
                HashSlots none;
                memset(&none, 0, sizeof(none));
                k_bn_hash_all<<<dim3(65536 / 256, (unsigned)G), 256, 0, st>>>(N, HM, HK, 0, none);                  // init / final rows per memory
            } else k_bn_hash_all<<<dim3(grid1(N) + 65536 / 256, (unsigned)G), 256, 0, st>>>(N, HM, HK, (size_t)grid1(N), HS);
        }
        // the write hashes are the read hashes + gamma^2 (k_bn_hash_rw): the top layer runs on the read rows only
        // both grand products and the openings are enqueued back to back (nothing here depends on a result read by the host: the
        // points are challenges); ONE wait at the end, then the two transcript replays
        if (mid && mid_at == 2) (*mid)();
        GpLaunchSet gp_set;   // the rounds of BOTH grand products share their launches: the small one (2^16 rows) hides inside the big one's
        grand_product_core(ctx, 2 * G, N, nullptr, H1, gp1_at, gp1_bytes, tmp_claims, x, mirror ? &gamma2 : nullptr, &replay_gp1, &gp_set, use_slots ? &slots : nullptr);  // reads then writes (prover.rs:161-165)
        grand_product_core(ctx, 2 * G, M, nullptr, H2, gp2_at, gp2_bytes, tmp_claims2, y, nullptr, &replay_gp2, &gp_set);  // inits then finals (prover.rs:167-171)
        gp_launch_set(ctx, st, gp_set);
        // openings (prover.rs:173-178, mod.rs:80-93)
        // (the points the grand products report are the runs the eq tables were built from)
        if (x.size() != (size_t)nu || y.size() != 16) throw Error("hg_lasso_prove_bn254: unexpected grand-product point length");
        for (int i = 0; i < nu; i++) if (!fr_eq(x[i], chain[final_point_at(gp1_at, nu) + i])) throw Error("hg_lasso_prove_bn254: grand product #1's point is not the expected challenge run");
        for (int i = 0; i < 16; i++) if (!fr_eq(y[i], chain[final_point_at(gp2_at, 16) + i])) throw Error("hg_lasso_prove_bn254: grand product #2's point is not the expected challenge run");
        // every opening at x in one launch, every opening at y in another (k_bn_dot_u64_multi); order on the wire per chunk: dim(x),
        // read_ts(x), final_cts(y), then E_m(x)
        std::vector<const Fr*> open_at;
        {
            DotU64Tabs tx, ty;
            DotU64Out ox, oy;
            memset(&tx, 0, sizeof(tx)); memset(&ty, 0, sizeof(ty)); memset(&ox, 0, sizeof(ox)); memset(&oy, 0, sizeof(oy));
            int nx = 0, ny = 0;
            auto add = [&](DotU64Tabs& T, DotU64Out& O, int& cnt, const u64* t) {
                if (cnt >= 64) throw Error("hg_lasso_prove_bn254: too many openings");
                const ResRef o = res_slots(ctx, 1);
                T.t[cnt] = t; O.out[cnt] = o.dev; cnt++;
                open_at.push_back(o.host);
            };
            // the E tables at x go through k_bn_open_e when the launch shape fits (a workgroup's rows inside one lookup segment)
            const int egx = (int)std::min<size_t>((N + BN_TPB - 1) / BN_TPB, 1024);
            const size_t echunk = N / (size_t)egx;
            bool e_fast = N % (size_t)egx == 0 && (echunk & (echunk - 1)) == 0 && echunk <= ((size_t)1 << L.seg_shift) && L.alpha <= 32;
            for (int l = 0; l < L.num_lookups && e_fast; l++) if (L.lookup_nmems[l] > 4) e_fast = false;
            BnOpenE OE;
            memset(&OE, 0, sizeof(OE));
            for (auto& chk : lp.chunks) {
                const int c = chk.first;
                add(tx, ox, nx, dims + (size_t)c * N);
                add(tx, ox, nx, read_ts[c]);
                add(ty, oy, ny, final_cts[c]);
                for (int m : chk.second) {
                    if (!e_fast) { add(tx, ox, nx, ep + (size_t)m * N); continue; }
                    const ResRef o = res_slots(ctx, 1);
                    OE.mem[OE.nE] = m; OE.out[OE.nE] = o.dev; OE.nE++;
                    open_at.push_back(o.host);
                }
            }
            auto run = [&](const Fr* e, const DotU64Tabs& T, const DotU64Out& O, int cnt, size_t n) {
                if (!cnt) return;
                const int gx = (int)std::min<size_t>((n + BN_TPB - 1) / BN_TPB, 512), gy = (cnt + BN_DOT_GROUP - 1) / BN_DOT_GROUP;
                Fr* part = dalloc((size_t)gx * gy * BN_DOT_GROUP);
                k_bn_dot_u64_multi<<<dim3(gx, gy), BN_TPB, 0, st>>>(e, T, cnt, n, part);
                k_bn_dot_u64_reduce<<<cnt, BN_TPB, 0, st>>>(part, gx, O);
            };
            run(eqx, tx, ox, nx, N);
            if (e_fast && OE.nE) {
                OE.ep = ep; OE.N = N; OE.rows = L.rows; OE.seg_lookup = L.seg_lookup; OE.seg_shift = L.seg_shift;
                for (int l = 0; l < 32; l++) { OE.nmems[l] = l < L.num_lookups ? L.lookup_nmems[l] : 0; for (int k = 0; k < 4; k++) OE.mems[l][k] = L.lookup_mems[l][k]; }
                Fr* part = dalloc((size_t)egx * 32);
                k_bn_open_e<<<egx, BN_TPB, 0, st>>>(eqx, OE, echunk, part);
                k_bn_open_e_reduce<<<OE.nE, BN_TPB, 0, st>>>(part, egx, OE);
            }
            run(eqy, ty, oy, ny, M);
        }
        if (mid && mid_at >= 3) (*mid)();
        if (st_col != st) hipc(hipStreamWaitEvent(st, ctx->bn_ev[1], 0), "stream wait");   // the claimed sum and the collation rounds' sums
        const bool times = hg_times("bn");   // read at every call (host.hpp)
        const double t_enq = wall_ms();
        res_sync(ctx, st, "lasso_prove_bn254: sync");
        if (times) fprintf(stderr, "[hg bn]   lasso node: enqueued, waited %.3f ms for its stream\n", wall_ms() - t_enq);
        if (h_bad && h_bad->l[0]) throw Error("hg_lasso_prove_bn254: the node input holds a value that is not below 2^64 (not a range-shifted value)");
        {   // the two replays are independent (own byte buffers, own claim vectors): ~0.25 and ~0.2 ms of host field arithmetic
            std::exception_ptr err;
            std::thread other([&] { try { replay_gp2(); } catch (...) { err = std::current_exception(); } });
            try { replay_gp1(); } catch (...) { other.join(); throw; }
            other.join();
            if (err) std::rethrow_exception(err);
        }
        h_claimed = *r_claimed.host;
        for (size_t i = 0; i < h_col.size(); i++) h_col[i] = r_col.host[i];
        for (const Fr* q : open_at) opens.push_back(*q);
    } catch (...) {
        ctx->arena_rewind(arena_mark);
        throw;
    }
    ctx->arena_rewind(arena_mark);
    // transcript (lasso.rs:57-114)
    proof.clear();
    write_be32(proof, h_claimed);                                        // :269
    Fr claim = fr_to_mont(h_claimed);
    for (int rd = 0; rd < nu; rd++) {                                   // collation rounds; the result is dropped (:97)
        Fr c[4];
        claim = replay_round(&h_col[(size_t)rd * 2], 2, claim, fr_to_mont(chain[col_at + rd]), c);
        for (int k = 0; k < 3; k++) write_be32(proof, fr_from_mont(c[k]));
    }
    proof.insert(proof.end(), gp1_bytes.begin(), gp1_bytes.end());
    proof.insert(proof.end(), gp2_bytes.begin(), gp2_bytes.end());
    for (const Fr& v : opens) write_be32(proof, v);
    for (int i = 0; i < nu; i++) memcpy(claim_out + 4 * i, chain[r_at + i].l, 32);
    memcpy(claim_out + 4 * nu, h_claimed.l, 32);
}

void lasso_prove_bn254(hg_ctx* ctx, const hg_pk* pk, const u64* in4, size_t chain_skip, std::vector<uint8_t>& proof, u64* claim_out) {
    if (!in4) throw Error("hg_lasso_prove_bn254: null input");
    lasso_prove_bn254_impl(ctx, pk, in4, nullptr, chain_skip, proof, claim_out);
}


}  // namespace bn
}  // namespace hg
