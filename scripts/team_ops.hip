/*
 *  team_ops.hip - what the team tier (csrc/hip/weighted_teams.hip) may assume about gfx950's packed maxima.
 *
 *  1. RATES of the candidate instructions, alone and in the kernel's own mix (per register of an affine local step: four
 *     v_add_u32 and six two-input / four and a half three-input maxima), 8 wavefronts per SIMD, 8 chains per lane.
 *  2. EXACTNESS of `v_pk_maximum3_f16` / `v_pk_max_f16` as INTEGER maxima: positive normal halves (bit patterns 0x0400 ...
 *     0x7BFF) order like their patterns, so on cells kept inside that range a floating-point maximum is an unsigned one.
 *     Checked on 2^26 random triples plus the edges, and - informational - outside the range (denormals, infinities, NaN).
 *
 *      hipcc --offload-arch=gfx950 -O2 scripts/team_ops.hip -o scripts/bin/team_ops && scripts/bin/team_ops
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHAINS 8
#define INNER 64

#define BENCH_KERNEL(NAME, ASM)                                                                                        \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed, int iterations) {                        \
        uint32_t x[CHAINS];                                                                                            \
        uint32_t a = (seed ^ threadIdx.x) & 0x3FFF3FFFu | 0x04000400u, b = (seed * 2654435761u + blockIdx.x) & 0x3FFF3FFFu | 0x04000400u; \
        for (int c = 0; c < CHAINS; ++c) x[c] = (a * (c + 1) + b) & 0x3FFF3FFFu | 0x04000400u;                         \
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
BENCH_KERNEL(k_pk_max_u16, "v_pk_max_u16 %0, %0, %1")
BENCH_KERNEL(k_pk_max_f16, "v_pk_max_f16 %0, %0, %1")
BENCH_KERNEL(k_pk_maximum3_f16, "v_pk_maximum3_f16 %0, %0, %1, %2")
BENCH_KERNEL(k_pk_minimum3_f16, "v_pk_minimum3_f16 %0, %0, %1, %2")
BENCH_KERNEL(k_maximum3_f32, "v_maximum3_f32 %0, %0, %1, %2")
BENCH_KERNEL(k_max3_i32, "v_max3_i32 %0, %0, %1, %2")
BENCH_KERNEL(k_max3_u16, "v_max3_u16 %0, %0, %1, %2")
BENCH_KERNEL(k_dpp_mov, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
BENCH_KERNEL(k_add_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
/* the step's mixes, per register: 10 instructions = 4 + 6, and 8.5 -> 17 per two registers = 8 + 9 */
BENCH_KERNEL(k_mix_two_input,
             "v_add_u32 %0, %0, %1\n v_pk_max_u16 %0, %0, %2\n v_pk_max_u16 %0, %0, %1\n v_pk_max_u16 %0, %0, %2\n v_add_u32 %0, %0, %1\n"
             "v_add_u32 %0, %0, %2\n v_pk_max_u16 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_pk_max_u16 %0, %0, %2\n v_pk_max_u16 %0, %0, %1")
BENCH_KERNEL(k_mix_three_input,
             "v_add_u32 %0, %0, %1\n v_pk_maximum3_f16 %0, %0, %2, %1\n v_pk_max_f16 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
             "v_add_u32 %0, %0, %2\n v_pk_max_f16 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_pk_max_f16 %0, %0, %2\n"
             "v_add_u32 %0, %0, %1\n v_pk_maximum3_f16 %0, %0, %2, %1\n v_pk_max_f16 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
             "v_add_u32 %0, %0, %2\n v_pk_max_f16 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_pk_max_f16 %0, %0, %2\n v_pk_maximum3_f16 %0, %0, %2, %1")

__device__ __forceinline__ uint32_t lcg(uint32_t &state) { return state = state * 1664525u + 1013904223u; }
__device__ __forceinline__ uint32_t half_max3(uint32_t a, uint32_t b, uint32_t c) { uint32_t m = a > b ? a : b; return m > c ? m : c; }

/* mode 0: every half inside [0x0400, 0x7BFF]; 1: anywhere in [0, 0x7FFF] (denormals, infinity, NaN patterns included) */
__global__ void k_exactness(unsigned long long *wrong, uint32_t seed, int rounds, int mode) {
    uint32_t state = seed ^ (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    unsigned long long bad3 = 0, bad2 = 0;
    for (int i = 0; i < rounds; ++i) {
        uint32_t v[3];
        for (int k = 0; k < 3; ++k) {
            uint32_t const r = lcg(state) ^ (lcg(state) >> 16);
            uint32_t low = r & 0x7FFFu, high = (r >> 16) & 0x7FFFu;
            if (mode == 0) low = 0x0400u + low % (0x7C00u - 0x0400u), high = 0x0400u + high % (0x7C00u - 0x0400u);
            if ((i & 15) == k) high = low;           /* equal halves */
            if ((i & 31) == 7 + k) low = low | 1u;     /* neighbours */
            v[k] = low | high << 16;
        }
        if ((i & 63) == 0) v[1] = v[0];
        if ((i & 127) == 1) v[0] = 0x04000400u, v[2] = 0x7BFF7BFFu;
        uint32_t got3, got2;
        asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(got3) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
        asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(got2) : "v"(v[0]), "v"(v[1]));
        uint32_t const want3 = half_max3(v[0] & 0xFFFFu, v[1] & 0xFFFFu, v[2] & 0xFFFFu) | half_max3(v[0] >> 16, v[1] >> 16, v[2] >> 16) << 16;
        uint32_t const want2 = half_max3(v[0] & 0xFFFFu, v[1] & 0xFFFFu, 0) | half_max3(v[0] >> 16, v[1] >> 16, 0) << 16;
        bad3 += got3 != want3, bad2 += got2 != want2;
    }
    if (bad3) atomicAdd(&wrong[0], bad3);
    if (bad2) atomicAdd(&wrong[1], bad2);
}

template <typename kernel_t>
static double time_kernel(kernel_t kernel, uint32_t *out, int iterations, int blocks) {
    hipEvent_t start, stop;
    hipEventCreate(&start), hipEventCreate(&stop);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 12345u, 4);
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
    int const blocks = cus * 8, iterations = 200;
    uint32_t *out;
    hipMalloc(&out, 64);
    double const lane_ops = (double)blocks * 256 * iterations * INNER * CHAINS;
    printf("{\"device\": \"%s\", \"compute_units\": %d", props.gcnArchName, cus);
#define REPORT(NAME, KERNEL, PER) printf(", \"%s\": %.2f", NAME, lane_ops * PER / time_kernel(KERNEL, out, iterations, blocks) / 1e12);
    REPORT("v_add_u32", k_add_u32, 1)
    REPORT("v_pk_max_u16", k_pk_max_u16, 1)
    REPORT("v_pk_max_f16", k_pk_max_f16, 1)
    REPORT("v_pk_maximum3_f16", k_pk_maximum3_f16, 1)
    REPORT("v_pk_minimum3_f16", k_pk_minimum3_f16, 1)
    REPORT("v_maximum3_f32", k_maximum3_f32, 1)
    REPORT("v_max3_i32", k_max3_i32, 1)
    REPORT("v_max3_u16", k_max3_u16, 1)
    REPORT("v_mov_b32_dpp_row_shr", k_dpp_mov, 1)
    REPORT("v_add_u32_dpp_row_shr", k_add_dpp, 1)
    {   /* registers (= pairs of cells) of an affine local step per second, and the cells per second that means */
        double const two = lane_ops / time_kernel(k_mix_two_input, out, iterations, blocks) / 1e12;
        double const three = lane_ops * 2 / time_kernel(k_mix_three_input, out, iterations, blocks) / 1e12;
        printf(", \"mix_4add_6max_Tregisters\": %.2f, \"mix_4add_4.5max3_Tregisters\": %.2f, \"ceiling_Tcells\": [%.2f, %.2f]", two, three, 2 * two, 2 * three);
    }
    unsigned long long *wrong;
    hipMalloc(&wrong, 16);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(wrong, 0, 16);
        hipLaunchKernelGGL(k_exactness, dim3(1024), dim3(256), 0, 0, wrong, 99u + mode, 256, mode);
        unsigned long long host[2] = {0, 0};
        hipMemcpy(host, wrong, 16, hipMemcpyDeviceToHost);
        printf(", \"%s\": {\"triples\": %llu, \"maximum3_differs\": %llu, \"max_differs\": %llu}", mode ? "any_pattern_below_0x8000" : "normal_halves_0x0400_0x7BFF",
               1024ull * 256 * 256, host[0], host[1]);
    }
    printf(", \"unit\": \"1e12 lane-ops/s\"}\n");
    hipFree(out), hipFree(wrong);
    return 0;
}
