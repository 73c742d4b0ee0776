/*
 *  wave_latency.hip - what does ONE wavefront pay per instruction?  The systolic tier (hip/systolic.hip) often runs one
 *  or two wavefronts per SIMD and its per-column recurrence is a dependent chain, so its ceiling is set by issue LATENCY,
 *  not by the throughput numbers of valu_peak.hip.  This measures, in SIMD cycles per instruction at the clock the
 *  device reports: dependent chains of the opcodes of that kernel at 1, 2, 4 and 8 independent chains per lane, for 1, 2
 *  and 4 wavefronts on one SIMD.
 *
 *      hipcc --offload-arch=gfx950 -O2 scripts/wave_latency.hip -o scripts/bin/wave_latency && scripts/bin/wave_latency
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define INNER 256

template <int chains_, int op_>
__global__ __launch_bounds__(1024) void chain_kernel(uint32_t *out, uint32_t seed, int iterations) {
    uint32_t x[chains_];
    uint32_t a = seed ^ threadIdx.x, b = seed * 2654435761u + blockIdx.x;
    for (int c = 0; c < chains_; ++c) x[c] = a * (c + 1) + b;
    for (int i = 0; i < iterations; ++i) {
#pragma unroll
        for (int k = 0; k < INNER / chains_; ++k) {
#pragma unroll
            for (int c = 0; c < chains_; ++c) {
                if constexpr (op_ == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
                if constexpr (op_ == 1) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                if constexpr (op_ == 2)
                    asm volatile("v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2"
                                 : "+v"(x[c]) : "v"(a));
                if constexpr (op_ == 3) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[c]));
                if constexpr (op_ == 4) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[c]));
                if constexpr (op_ == 5) { // the systolic hand-over: readlane -> SGPR -> v_mov -> DPP
                    uint32_t s;
                    asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(x[c]));
                    uint32_t y;
                    asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "s"(s));
                    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(y) : "v"(x[c]));
                    x[c] = y;
                }
                if constexpr (op_ == 6) { // one DP cell, linear gaps: sdwa add, max3, add (dependent on the previous cell)
                    uint32_t t;
                    asm volatile("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
                                 : "=v"(t) : "v"(b), "v"(a));
                    asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(t));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
                }
            }
        }
    }
    uint32_t sum = 0;
    for (int c = 0; c < chains_; ++c) sum ^= x[c];
    if (sum == 0x12345678u) out[0] = sum;
}

template <typename kernel_t>
static double cycles_per_instruction(kernel_t kernel, uint32_t *out, int waves, double clock_hz, int instructions_per_inner) {
    int const iterations = 2000;
    hipEvent_t start, stop;
    (void)hipEventCreate(&start), (void)hipEventCreate(&stop);
    // ONE workgroup of 4 x `waves` wavefronts: a workgroup lives on one CU and its wavefronts are dealt over the 4 SIMDs
    hipLaunchKernelGGL(kernel, dim3(1), dim3(256 * waves), 0, 0, out, 1u, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(start, 0);
    hipLaunchKernelGGL(kernel, dim3(1), dim3(256 * waves), 0, 0, out, 1u, iterations);
    (void)hipEventRecord(stop, 0);
    (void)hipEventSynchronize(stop);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, start, stop);
    double const per_wave = (double)iterations * INNER * instructions_per_inner; // instructions one wavefront issued
    return ms * 1e-3 * clock_hz / (per_wave * waves);                           // SIMD cycles per instruction
}

int main() {
    uint32_t *out;
    (void)hipMalloc((void **)&out, 64);
    hipDeviceProp_t props;
    (void)hipGetDeviceProperties(&props, 0);
    double const clock_hz = props.clockRate * 1e3;
    printf("{\"device\": \"%s\", \"clock_mhz\": %d, \"unit\": \"SIMD cycles per wave-instruction (lower is better); waves = wavefronts per SIMD\"",
           props.gcnArchName, props.clockRate / 1000);
#define ROW(NAME, OP, PER)                                                                                             \
    for (int waves = 1; waves <= 4; waves *= 2) {                                                                      \
        printf(",\n \"%s chains=1 waves=%d\": %.2f", NAME, waves, cycles_per_instruction(chain_kernel<1, OP>, out, waves, clock_hz, PER)); \
        printf(", \"%s chains=2 waves=%d\": %.2f", NAME, waves, cycles_per_instruction(chain_kernel<2, OP>, out, waves, clock_hz, PER));   \
        printf(", \"%s chains=4 waves=%d\": %.2f", NAME, waves, cycles_per_instruction(chain_kernel<4, OP>, out, waves, clock_hz, PER));   \
        printf(", \"%s chains=8 waves=%d\": %.2f", NAME, waves, cycles_per_instruction(chain_kernel<8, OP>, out, waves, clock_hz, PER));   \
    }
    ROW("v_add_u32", 0, 1)
    ROW("v_max3_i32", 1, 1)
    ROW("v_add_u32_sdwa", 2, 1)
    ROW("v_mov_dpp wave_shr", 3, 1)
    ROW("v_mov_dpp row_shr", 4, 1)
    ROW("readlane+mov+dpp", 5, 3)
    ROW("dp cell (sdwa,max3,add)", 6, 3)
    printf("}\n");
    return 0;
}
