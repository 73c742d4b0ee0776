/*
 *  valu_peak.hip - calibrates the INTEGER-VALU ceiling the edit-distance kernels are priced against.
 *
 *  The scoring kernels are integer VALU (+ LDS gather) bound, not HBM bound, so the roofline that
 *  says something about kernel quality is "wave-instructions per second per opcode".  This program measures it on the
 *  box it runs on, per opcode the kernels actually use, with 8 independent dependency chains per lane and 8 waves/SIMD.
 *
 *      hipcc --offload-arch=gfx950 -O2 scripts/valu_peak.hip -o scripts/bin/valu_peak && scripts/bin/valu_peak
 *
 *  Prints one JSON object: {"op": lane-ops per second in units of 1e12, ...}.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHAINS 8
#define INNER 64

#define BENCH_KERNEL(NAME, ASM)                                                                                        \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed, int iterations) {                        \
        uint32_t x[CHAINS];                                                                                            \
        uint32_t a = seed ^ threadIdx.x, b = seed * 2654435761u + blockIdx.x;                                          \
        for (int c = 0; c < CHAINS; ++c) x[c] = a * (c + 1) + b;                                                       \
        for (int i = 0; i < iterations; ++i) {                                                                         \
            _Pragma("unroll") for (int k = 0; k < INNER; ++k) {                                                        \
                _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) asm volatile(ASM : "+v"(x[c]) : "v"(a), "v"(b));    \
            }                                                                                                          \
        }                                                                                                              \
        uint32_t sum = 0;                                                                                              \
        for (int c = 0; c < CHAINS; ++c) sum ^= x[c];                                                                  \
        if (sum == 0x12345678u) out[0] = sum;                                                                          \
    }

BENCH_KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
BENCH_KERNEL(k_or_b32, "v_or_b32 %0, %0, %1")
BENCH_KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
BENCH_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 31")
BENCH_KERNEL(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
BENCH_KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
BENCH_KERNEL(k_max_i32, "v_max_i32 %0, %0, %1")
BENCH_KERNEL(k_max3_i32, "v_max3_i32 %0, %0, %1, %2")
BENCH_KERNEL(k_bfe_i32, "v_bfe_i32 %0, %0, 8, 8")
BENCH_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
BENCH_KERNEL(k_pk_add_i16, "v_pk_add_i16 %0, %0, %1")
BENCH_KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
BENCH_KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
BENCH_KERNEL(k_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
BENCH_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
BENCH_KERNEL(k_sdwa_add, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
BENCH_KERNEL(k_pk_add_i16_opsel, "v_pk_add_i16 %0, %0, %1 op_sel_hi:[1,0]")
BENCH_KERNEL(k_and_b32, "v_and_b32 %0, %0, %1")
BENCH_KERNEL(k_xor_b32, "v_xor_b32 %0, %0, %1")
BENCH_KERNEL(k_sub_u32, "v_sub_u32 %0, %0, %1")
BENCH_KERNEL(k_lshlrev, "v_lshlrev_b32 %0, 1, %0")
BENCH_KERNEL(k_lshrrev, "v_lshrrev_b32 %0, 1, %0")
BENCH_KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
BENCH_KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
BENCH_KERNEL(k_add_co, "v_add_co_u32 %0, vcc, %0, %1")
BENCH_KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
BENCH_KERNEL(k_sub_clamp, "v_sub_u32 %0, %0, %1 clamp")
BENCH_KERNEL(k_max_u32, "v_max_u32 %0, %0, %1")
BENCH_KERNEL(k_min_i32, "v_min_i32 %0, %0, %1")
BENCH_KERNEL(k_med3_i32, "v_med3_i32 %0, %0, %1, %2")
BENCH_KERNEL(k_pk_sub_u16_clamp, "v_pk_sub_u16 %0, %0, %1 clamp")
BENCH_KERNEL(k_pk_max_u16, "v_pk_max_u16 %0, %0, %1")
BENCH_KERNEL(k_mov, "v_mov_b32 %0, %1")
BENCH_KERNEL(k_add_sdwa_sext, "v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2")
BENCH_KERNEL(k_cndmask_e64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
BENCH_KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
BENCH_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
BENCH_KERNEL(k_add_f32, "v_add_f32 %0, %0, %1")
BENCH_KERNEL(k_max_f32, "v_max_f32 %0, %0, %1")
BENCH_KERNEL(k_max3_f32, "v_max3_f32 %0, %0, %1, %2")
BENCH_KERNEL(k_pk_max_f16, "v_pk_max_f16 %0, %0, %1")
BENCH_KERNEL(k_pk_add_f16, "v_pk_add_f16 %0, %0, %1")

/* LDS gather: one ds_read_b128 per step from a 4 KiB table of 16-byte rows, row picked by a per-lane pseudo-random byte
 * (the Myers kernel's Peq access pattern) or by a uniform row (conflict-free broadcast). */
template <int random_rows_>
__global__ __launch_bounds__(256) void k_lds_b128(uint32_t *out, uint32_t seed, int iterations) {
    __shared__ __attribute__((aligned(16))) uint4 table[256];
    table[threadIdx.x] = make_uint4(threadIdx.x, seed, blockIdx.x, 7);
    __syncthreads();
    uint32_t state = seed ^ (threadIdx.x * 2654435761u);
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int i = 0; i < iterations; ++i) {
#pragma unroll
        for (int k = 0; k < INNER; ++k) {
            state = state * 1664525u + 1013904223u;
            uint32_t row = random_rows_ ? (state >> 24) % 95u + 32u : (uint32_t)(k & 255);
            uint4 const v = table[row];
            acc.x ^= v.x, acc.y ^= v.y, acc.z ^= v.z, acc.w ^= v.w;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc.x;
}


/* The Myers column update of csrc/hip/lev_myers.hip on synthetic match masks held in registers: the pure-VALU rate of
 * the real instruction mix (bitop3 / or / and / addc carry chain / alignbit), no LDS, no memory. */
template <int words_>
__device__ __forceinline__ void myers_column_probe(uint32_t (&vp)[words_], uint32_t (&vn)[words_], uint32_t const (&eq)[words_],
                                                   uint32_t &hp_history, uint32_t &hn_history) {
    uint32_t carry = 0, hp_below = 0, hn_below = 0;
#pragma unroll
    for (int w = 0; w < words_; ++w) {
        uint32_t const xv = eq[w] | vn[w];
        uint32_t carry_out;
        uint32_t const sum = __builtin_addc(eq[w] & vp[w], vp[w], carry, &carry_out);
        carry = carry_out;
        uint32_t const d0 = (sum ^ vp[w]) | eq[w];
        uint32_t const hp = vn[w] | ~(d0 | vp[w]);
        uint32_t const hn = vp[w] & d0;
        uint32_t const hp_shifted = w == 0 ? ((hp << 1) | 1u) : __builtin_amdgcn_alignbit(hp, hp_below, 31);
        uint32_t const hn_shifted = w == 0 ? (hn << 1) : __builtin_amdgcn_alignbit(hn, hn_below, 31);
        hp_below = hp, hn_below = hn;
        vp[w] = hn_shifted | ~(xv | hp_shifted);
        vn[w] = hp_shifted & xv;
    }
    (void)hp_history, (void)hn_history; /* the kernel no longer tracks the score per column (popcount at text end) */
}

/* The same column in BLOCK form (Myers 1999 for long patterns; the reference's serial.hpp:2182-2204): no carry between the
 * words of the add - a -1 entering a word from below acts like a match in its first row - so the add is a plain v_add_u32
 * and the shifts take their low bit from the word below as a 0 / 1 value.  Measured beside the full-width form to decide
 * which one the kernels should use. */
template <int words_, bool add_shift_>
__device__ __forceinline__ void myers_column_block_probe(uint32_t (&vp)[words_], uint32_t (&vn)[words_], uint32_t const (&eq)[words_]) {
    uint32_t hp_in = 1, hn_in = 0;
#pragma unroll
    for (int w = 0; w < words_; ++w) {
        uint32_t const xv = eq[w] | vn[w];
        uint32_t const eq_in = eq[w] | hn_in;
        uint32_t const sum = (eq_in & vp[w]) + vp[w];
        uint32_t const d0 = (sum ^ vp[w]) | eq_in;
        uint32_t const hp = vn[w] | ~(d0 | vp[w]);
        uint32_t const hn = vp[w] & d0;
        uint32_t const hp_shifted = add_shift_ ? ((hp + hp) | hp_in) : ((hp << 1) | hp_in);
        uint32_t const hn_shifted = add_shift_ ? ((hn + hn) | hn_in) : ((hn << 1) | hn_in);
        hp_in = hp >> 31, hn_in = hn >> 31;
        vp[w] = hn_shifted | ~(xv | hp_shifted);
        vn[w] = hp_shifted & xv;
    }
}

template <int words_, bool add_shift_>
__global__ __launch_bounds__(256) void k_myers_block(uint32_t *out, uint32_t seed, int iterations) {
    uint32_t vp[words_], vn[words_], eq[4][words_];
    for (int w = 0; w < words_; ++w) {
        vp[w] = ~0u, vn[w] = 0;
        for (int k = 0; k < 4; ++k) eq[k][w] = (seed * (w + 3) + threadIdx.x * 2654435761u) >> (k * 3 + (blockIdx.x & 3));
    }
    for (int i = 0; i < iterations; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) myers_column_block_probe<words_, add_shift_>(vp, vn, eq[k & 3]);
    }
    uint32_t sum = 0;
    for (int w = 0; w < words_; ++w) sum ^= vp[w] ^ vn[w];
    if (sum == 0x12345678u) out[0] = sum;
}

template <int words_>
__global__ __launch_bounds__(256) void k_myers_pure(uint32_t *out, uint32_t seed, int iterations) {
    uint32_t vp[words_], vn[words_], eq[4][words_];
    for (int w = 0; w < words_; ++w) {
        vp[w] = ~0u, vn[w] = 0;
        for (int k = 0; k < 4; ++k) eq[k][w] = (seed * (w + 3) + threadIdx.x * 2654435761u) >> (k * 3 + (blockIdx.x & 3));
    }
    uint32_t hp_history = 0, hn_history = 0;
    for (int i = 0; i < iterations; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) myers_column_probe<words_>(vp, vn, eq[k & 3], hp_history, hn_history);
    }
    uint32_t sum = hp_history ^ hn_history;
    for (int w = 0; w < words_; ++w) sum ^= vp[w] ^ vn[w];
    if (sum == 0x12345678u) out[0] = sum;
}

/* The tiny-token launch's column (hip/myers_tiny.hip: `tiny_column`): two 16-row patterns to a register, R registers a lane, one
 * column of a text per step - on register-resident masks, no LDS, no barriers: the issue ceiling of that launch's columns. */
typedef unsigned short probe_pk_u16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tiny_column_probe(uint32_t &vp, uint32_t &vn, uint32_t eq) {
    uint32_t const xv = eq | vn;
    probe_pk_u16 const sum16 = __builtin_bit_cast(probe_pk_u16, eq & vp) + __builtin_bit_cast(probe_pk_u16, vp);
    uint32_t const sum = __builtin_bit_cast(uint32_t, sum16);
    uint32_t const d0 = (sum ^ vp) | eq;
    uint32_t const hp = vn | ~(d0 | vp);
    uint32_t const hn = vp & d0;
    probe_pk_u16 const hp16 = __builtin_bit_cast(probe_pk_u16, hp) << (probe_pk_u16)(1), hn16 = __builtin_bit_cast(probe_pk_u16, hn) << (probe_pk_u16)(1);
    uint32_t const hp_shifted = __builtin_bit_cast(uint32_t, hp16) | 0x00010001u, hn_shifted = __builtin_bit_cast(uint32_t, hn16);
    vp = hn_shifted | ~(xv | hp_shifted);
    vn = hp_shifted & xv;
}
template <int registers_>
__global__ __launch_bounds__(256, 4) void k_tiny_pure(uint32_t *out, uint32_t seed, int iterations) {
    uint32_t vp[registers_], vn[registers_], eq[4][registers_];
    for (int d = 0; d < registers_; ++d) {
        vp[d] = ~0u, vn[d] = 0;
        for (int k = 0; k < 4; ++k) eq[k][d] = (seed * (d + 3) + threadIdx.x * 2654435761u) >> (k * 3 + (blockIdx.x & 3));
    }
    for (int i = 0; i < iterations; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int d = 0; d < registers_; ++d) tiny_column_probe(vp[d], vn[d], eq[k][d]);
    }
    uint32_t sum = 0;
    for (int d = 0; d < registers_; ++d) sum ^= vp[d] ^ vn[d];
    if (sum == 0x12345678u) out[0] = sum;
}

template <typename kernel_t>
static double time_kernel(kernel_t kernel, uint32_t *out, int iterations, int blocks) {
    hipEvent_t start, stop;
    hipEventCreate(&start), hipEventCreate(&stop);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 12345u, 4); /* warm up */
    hipDeviceSynchronize();
    hipEventRecord(start, 0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 12345u, iterations);
    hipEventRecord(stop, 0);
    hipEventSynchronize(stop);
    float ms = 0;
    hipEventElapsedTime(&ms, start, stop);
    hipEventDestroy(start), hipEventDestroy(stop);
    return ms * 1e-3;
}

int main() {
    hipDeviceProp_t props;
    if (hipGetDeviceProperties(&props, 0) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
    int const cus = props.multiProcessorCount;
    int const blocks = cus * 8; /* 8 workgroups x 4 waves = 32 waves per CU = 8 per SIMD */
    int const iterations = 400;
    uint32_t *out;
    hipMalloc(&out, 64);
    double const lane_ops = (double)blocks * 256 * iterations * INNER * CHAINS;
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"clock_mhz\": %d", props.gcnArchName, cus, props.clockRate / 1000);
#define REPORT(NAME, KERNEL) printf(", \"%s\": %.2f", NAME, lane_ops / time_kernel(KERNEL, out, iterations, blocks) / 1e12);
    REPORT("v_add_u32", k_add_u32)
    REPORT("v_or_b32", k_or_b32)
    REPORT("v_bitop3_b32", k_bitop3)
    REPORT("v_alignbit_b32", k_alignbit)
    REPORT("v_addc_co_u32", k_addc)
    REPORT("v_add3_u32", k_add3)
    REPORT("v_max_i32", k_max_i32)
    REPORT("v_max3_i32", k_max3_i32)
    REPORT("v_bfe_i32", k_bfe_i32)
    REPORT("v_perm_b32", k_perm)
    REPORT("v_pk_add_i16", k_pk_add_i16)
    REPORT("v_pk_max_i16", k_pk_max_i16)
    REPORT("v_pk_add_u16", k_pk_add_u16)
    REPORT("v_pk_min_u16", k_pk_min_u16)
    REPORT("v_cndmask_b32", k_cndmask)
    REPORT("v_add_u32_sdwa", k_sdwa_add)
    REPORT("v_pk_add_i16_opsel", k_pk_add_i16_opsel)
    REPORT("v_and_b32", k_and_b32)
    REPORT("v_xor_b32", k_xor_b32)
    REPORT("v_sub_u32", k_sub_u32)
    REPORT("v_lshlrev_b32", k_lshlrev)
    REPORT("v_lshrrev_b32", k_lshrrev)
    REPORT("v_lshl_or_b32", k_lshl_or)
    REPORT("v_and_or_b32", k_and_or)
    REPORT("v_add_co_u32", k_add_co)
    REPORT("v_bcnt_u32_b32", k_bcnt)
    REPORT("v_sub_u32_clamp", k_sub_clamp)
    REPORT("v_max_u32", k_max_u32)
    REPORT("v_min_i32", k_min_i32)
    REPORT("v_med3_i32", k_med3_i32)
    REPORT("v_pk_sub_u16_clamp", k_pk_sub_u16_clamp)
    REPORT("v_pk_max_u16", k_pk_max_u16)
    REPORT("v_mov_b32", k_mov)
    REPORT("v_add_u32_sdwa_sext", k_add_sdwa_sext)
    REPORT("v_cndmask_b32_e64_sgpr", k_cndmask_e64)
    REPORT("v_mad_u32_u24", k_mad_u32_u24)
    REPORT("v_mul_lo_u32", k_mul_lo)
    REPORT("v_add_f32", k_add_f32)
    REPORT("v_max_f32", k_max_f32)
    REPORT("v_max3_f32", k_max3_f32)
    REPORT("v_pk_max_f16", k_pk_max_f16)
    REPORT("v_pk_add_f16", k_pk_add_f16)
    {   /* word-steps per second: one word-step = 32 DP cells of one lane */
        int const columns = 1600; /* iterations x 16 */
        double const base = (double)blocks * 256 * columns;
        printf(", \"myers_pure_W4_Tcells\": %.2f", base * 128 / time_kernel(k_myers_pure<4>, out, columns / 16, blocks) / 1e12);
        printf(", \"myers_pure_W5_Tcells\": %.2f", base * 160 / time_kernel(k_myers_pure<5>, out, columns / 16, blocks) / 1e12);
        printf(", \"myers_pure_W8_Tcells\": %.2f", base * 256 / time_kernel(k_myers_pure<8>, out, columns / 16, blocks) / 1e12);
        printf(", \"myers_pure_W1_Tcells\": %.2f", base * 32 / time_kernel(k_myers_pure<1>, out, columns / 16, blocks) / 1e12);
        printf(", \"myers_block_W4_Tcells\": %.2f", base * 128 / time_kernel((k_myers_block<4, false>), out, columns / 16, blocks) / 1e12);
        printf(", \"myers_block_addshift_W4_Tcells\": %.2f", base * 128 / time_kernel((k_myers_block<4, true>), out, columns / 16, blocks) / 1e12);
        printf(", \"myers_block_W8_Tcells\": %.2f", base * 256 / time_kernel((k_myers_block<8, false>), out, columns / 16, blocks) / 1e12);
        printf(", \"myers_block_addshift_W8_Tcells\": %.2f", base * 256 / time_kernel((k_myers_block<8, true>), out, columns / 16, blocks) / 1e12);
    }
    {   /* pair-columns per second: one step of a lane = one text column under 2 R patterns (hip/myers_tiny.hip) */
        int const columns = 1600, tiny_blocks = cus * 4; /* four workgroups a CU, as that launch runs */
        double const base = (double)tiny_blocks * 256 * columns;
        printf(", \"tiny_pure_R16_Tpair_columns\": %.3f", base * 32 / time_kernel(k_tiny_pure<16>, out, columns / 4, tiny_blocks) / 1e12);
        printf(", \"tiny_pure_R8_Tpair_columns\": %.3f", base * 16 / time_kernel(k_tiny_pure<8>, out, columns / 4, tiny_blocks) / 1e12);
    }
    double const lds_reads = (double)blocks * 256 * iterations * INNER;
    printf(", \"ds_read_b128_random95_Tlane_reads\": %.3f", lds_reads / time_kernel(k_lds_b128<1>, out, iterations, blocks) / 1e12);
    printf(", \"ds_read_b128_uniform_Tlane_reads\": %.3f", lds_reads / time_kernel(k_lds_b128<0>, out, iterations, blocks) / 1e12);
    printf(", \"unit\": \"1e12 lane-ops/s (a packed op counts once per lane)\"}\n");
    hipFree(out);
    return 0;
}
