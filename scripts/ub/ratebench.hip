// Issue-rate micro-benchmark for the gfx950 VALU instructions the Goldilocks kernels are made of.
// Each kernel runs ITER iterations of 32 independent copies of one instruction (8 register sets, 4 each);
// the report is wave-instructions per cycle per SIMD relative to v_mov_b32 (= 1 issue slot per 4 cycles / wave64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2000
#define REP4(x) x x x x
#define K(name, body, decl, sink)                                                          \
    __global__ __launch_bounds__(256) void name(uint64_t* out, uint32_t seed) {            \
        decl;                                                                              \
        for (int it = 0; it < ITER; it++) { REP4(body) }                                   \
        sink;                                                                              \
    }
#define DECL32 uint32_t a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = threadIdx.x
#define SINK32 out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7
#define DECL64 uint64_t a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = threadIdx.x
#define SINK64 out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7
#define EIGHT(fmt, cons) \
    asm volatile(fmt : "+v"(a0) : cons); asm volatile(fmt : "+v"(a1) : cons); asm volatile(fmt : "+v"(a2) : cons); asm volatile(fmt : "+v"(a3) : cons); \
    asm volatile(fmt : "+v"(a4) : cons); asm volatile(fmt : "+v"(a5) : cons); asm volatile(fmt : "+v"(a6) : cons); asm volatile(fmt : "+v"(a7) : cons);
#define EIGHTC(fmt, cons, clob) \
    asm volatile(fmt : "+v"(a0) : cons : clob); asm volatile(fmt : "+v"(a1) : cons : clob); asm volatile(fmt : "+v"(a2) : cons : clob); asm volatile(fmt : "+v"(a3) : cons : clob); \
    asm volatile(fmt : "+v"(a4) : cons : clob); asm volatile(fmt : "+v"(a5) : cons : clob); asm volatile(fmt : "+v"(a6) : cons : clob); asm volatile(fmt : "+v"(a7) : cons : clob);

#define CLOB_S "s20", "s21"
K(k_mov, EIGHT("v_mov_b32 %0, %1", "v"(b)), DECL32, SINK32)
K(k_add32, EIGHT("v_add_u32 %0, %0, %1", "v"(b)), DECL32, SINK32)
K(k_addco, EIGHTC("v_add_co_u32 %0, vcc, %0, %1", "v"(b), "vcc"), DECL32, SINK32)
K(k_addc, EIGHTC("v_addc_co_u32 %0, vcc, %0, %1, vcc", "v"(b), "vcc"), DECL32, SINK32)
K(k_cndmask, EIGHTC("v_cndmask_b32 %0, %0, %1, vcc", "v"(b), "vcc"), DECL32, SINK32)
K(k_add3, EIGHT("v_add3_u32 %0, %0, %1, %1", "v"(b)), DECL32, SINK32)
K(k_mullo, EIGHT("v_mul_lo_u32 %0, %0, %1", "v"(b)), DECL32, SINK32)
K(k_mulhi, EIGHT("v_mul_hi_u32 %0, %0, %1", "v"(b)), DECL32, SINK32)
K(k_mad24, EIGHT("v_mad_u32_u24 %0, %0, %1, %1", "v"(b)), DECL32, SINK32)
K(k_mulhi24, EIGHT("v_mul_hi_u32_u24 %0, %0, %1", "v"(b)), DECL32, SINK32)
K(k_lshladd64, EIGHT("v_lshl_add_u64 %0, %0, 0, %1", "v"(b)), DECL64, SINK64)
K(k_mov64, EIGHT("v_mov_b64 %0, %1", "v"(b)), DECL64, SINK64)
K(k_cmp64, EIGHTC("v_cmp_lt_u64 vcc, %0, %1", "v"(b), "vcc"), DECL64, SINK64)
K(k_cmp32, EIGHTC("v_cmp_lt_u32 vcc, %0, %1", "v"(b), "vcc"), DECL32, SINK32)
K(k_mad64, EIGHTC("v_mad_u64_u32 %0, s[20:21], %1, %1, %0", "v"((uint32_t)b), CLOB_S), DECL64, SINK64)
K(k_mad64z, EIGHTC("v_mad_u64_u32 %0, s[20:21], %1, %1, 0", "v"((uint32_t)b), CLOB_S), DECL64, SINK64)
K(k_lshl64, EIGHT("v_lshlrev_b64 %0, 3, %0", "v"(b)), DECL64, SINK64)
K(k_fma64, EIGHT("v_fma_f64 %0, %0, %1, %1", "v"(b)), DECL64, SINK64)
K(k_pkadd, EIGHT("v_pk_add_u16 %0, %0, %1", "v"(b)), DECL32, SINK32)
K(k_alignbit, EIGHT("v_alignbit_b32 %0, %0, %1, 7", "v"(b)), DECL32, SINK32)
K(k_subrev64pair, EIGHTC("v_sub_co_u32 %0, vcc, %0, %1\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_subb_co_u32 %0, vcc, %0, %1, vcc", "v"(b), "vcc"), DECL32, SINK32)
// carry written to an SGPR pair and consumed by the very next instruction (hazard probe: correctness + time)
K(k_carry_b2b, EIGHTC("v_add_co_u32 %0, s[20:21], %0, %1\n v_addc_co_u32 %0, s[20:21], %0, %1, s[20:21]", "v"(b), CLOB_S), DECL32, SINK32)

K(k_cnd_e64s, EIGHTC("v_cndmask_b32_e64 %0, %0, %1, s[20:21]", "v"(b), CLOB_S), DECL32, SINK32)
K(k_cnd_const, EIGHTC("v_cndmask_b32_e64 %0, 0, -1, vcc", "v"(b), "vcc"), DECL32, SINK32)
K(k_cmp_cnd, EIGHTC("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc", "v"(b), "vcc"), DECL32, SINK32)
K(k_cmp_x_cnd, EIGHTC("v_cmp_lt_u32 vcc, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc", "v"(b), "vcc"), DECL32, SINK32)
K(k_and, EIGHT("v_and_b32 %0, %0, %1", "v"(b)), DECL32, SINK32)
K(k_lshr, EIGHT("v_lshrrev_b32 %0, 3, %0", "v"(b)), DECL32, SINK32)
K(k_sub32, EIGHT("v_sub_u32 %0, %0, %1", "v"(b)), DECL32, SINK32)
K(k_min, EIGHT("v_min_u32 %0, %0, %1", "v"(b)), DECL32, SINK32)
K(k_addco64, EIGHTC("v_add_co_u32_e64 %0, s[20:21], %0, %1", "v"(b), CLOB_S), DECL32, SINK32)
K(k_mov_imm, EIGHT("v_mov_b32 %0, 0", "v"(b)), DECL32, SINK32)
K(k_xor, EIGHT("v_xor_b32 %0, %0, %1", "v"(b)), DECL32, SINK32)
K(k_lshl_or, EIGHT("v_lshl_or_b32 %0, %0, 3, %1", "v"(b)), DECL32, SINK32)
K(k_bfe, EIGHT("v_bfe_u32 %0, %0, 3, 5", "v"(b)), DECL32, SINK32)
K(k_subb, EIGHTC("v_subb_co_u32 %0, vcc, %0, %1, vcc", "v"(b), "vcc"), DECL32, SINK32)
K(k_addc64, EIGHTC("v_addc_co_u32_e64 %0, s[20:21], %0, %1, s[20:21]", "v"(b), CLOB_S), DECL32, SINK32)
struct Ent { const char* name; void (*fn)(uint64_t*, uint32_t); int per_iter; };
int main() {
    uint64_t* d; hipMalloc(&d, 2048 * 256 * 8);
    Ent es[] = {{"v_mov_b32", k_mov, 32}, {"v_add_u32", k_add32, 32}, {"v_add_co_u32", k_addco, 32}, {"v_addc_co_u32", k_addc, 32},
                {"v_cndmask_b32", k_cndmask, 32}, {"v_add3_u32", k_add3, 32}, {"v_mul_lo_u32", k_mullo, 32}, {"v_mul_hi_u32", k_mulhi, 32},
                {"v_mad_u32_u24", k_mad24, 32}, {"v_mul_hi_u32_u24", k_mulhi24, 32}, {"v_lshl_add_u64", k_lshladd64, 32},
                {"v_mov_b64", k_mov64, 32}, {"v_cmp_lt_u64", k_cmp64, 32}, {"v_cmp_lt_u32", k_cmp32, 32}, {"v_mad_u64_u32 acc", k_mad64, 32},
                {"v_mad_u64_u32 +0", k_mad64z, 32}, {"v_lshlrev_b64", k_lshl64, 32}, {"v_fma_f64", k_fma64, 32}, {"v_pk_add_u16", k_pkadd, 32},
                {"v_alignbit_b32", k_alignbit, 32}, {"v_cndmask_b32_e64 sgpr mask", k_cnd_e64s, 32}, {"v_cndmask_b32_e64 0,-1,vcc", k_cnd_const, 32},
                {"v_cmp_lt_u32; v_cndmask (2 instr)", k_cmp_cnd, 64}, {"v_cmp; add; add; v_cndmask (4 instr)", k_cmp_x_cnd, 128},
                {"v_and_b32", k_and, 32}, {"v_lshrrev_b32", k_lshr, 32}, {"v_sub_u32", k_sub32, 32}, {"v_min_u32", k_min, 32},
                {"v_add_co_u32_e64 sgpr", k_addco64, 32}, {"v_mov_b32 imm", k_mov_imm, 32}, {"v_xor_b32", k_xor, 32}, {"v_lshl_or_b32", k_lshl_or, 32},
                {"v_bfe_u32", k_bfe, 32}, {"v_subb_co_u32", k_subb, 32}, {"v_addc_co_u32_e64 sgpr", k_addc64, 32},
                {"sub_co,mov,mov,subb (4 instr)", k_subrev64pair, 128}, {"add_co->addc back-to-back (2 instr)", k_carry_b2b, 64}};
    const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    double base = 0;
    for (auto& e : es) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0); e.fn<<<blocks, 256>>>(d, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        double winstr = (double)blocks * 4 * ITER * e.per_iter;  // wave-instructions
        double per_simd_per_s = winstr / (256.0 * 4) / (ms * 1e-3);
        if (base == 0) base = per_simd_per_s;
        printf("%-40s %8.3f ms  %7.3f G wave-instr/s/SIMD  cost vs v_mov_b32: %.2f\n", e.name, ms, per_simd_per_s / 1e9, base / per_simd_per_s);
    }
    return 0;
}
