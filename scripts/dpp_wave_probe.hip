// dpp_wave_probe.hip - does gfx950 execute `v_mov_b32_dpp wave_shr:1` (a GFX8 / GFX9 DPP control: the whole wavefront shifts by one
// lane, across the four rows of sixteen) the way the ISA documents it, and at what rate?  The team tier (hip/weighted_teams.hip) hands
// strips from lane to lane with `row_shr:1` inside a row of sixteen; a team of 32 or 64 lanes needs the hand-over to cross rows.
//   hipcc --offload-arch=gfx950 -O3 dpp_wave_probe.hip -o bin/dpp_wave_probe && bin/dpp_wave_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int control_>
__device__ __forceinline__ unsigned shifted(unsigned head, unsigned value) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)head, (int)value, control_, 0xF, 0xF, false);
}

__global__ void check_kernel(unsigned *out) {
    unsigned const lane = threadIdx.x, value = 1000u + lane, head = 7u;
    out[lane] = shifted<0x138>(head, value);       // wave_shr:1
    out[64 + lane] = shifted<0x111>(head, value);  // row_shr:1
    out[128 + lane] = shifted<0x130>(head, value); // wave_shl:1
}

template <int control_>
__global__ __launch_bounds__(256) void rate_kernel(unsigned *out, unsigned rounds) {
    unsigned a = threadIdx.x, b = threadIdx.x * 3u, c = threadIdx.x * 5u, d = threadIdx.x * 7u;
    for (unsigned r = 0; r < rounds; ++r) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            a = shifted<control_>(a, b) + 1u, b = shifted<control_>(b, c) + 1u;
            c = shifted<control_>(c, d) + 1u, d = shifted<control_>(d, a) + 1u;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}

template <int control_>
static double rate(unsigned *device, char const *name) {
    unsigned const rounds = 4096, blocks = 256 * 8;
    hipEvent_t begin, end;
    hipEventCreate(&begin), hipEventCreate(&end);
    hipLaunchKernelGGL(rate_kernel<control_>, dim3(blocks), dim3(256), 0, 0, device, 16u);
    hipEventRecord(begin, 0);
    hipLaunchKernelGGL(rate_kernel<control_>, dim3(blocks), dim3(256), 0, 0, device, rounds);
    hipEventRecord(end, 0);
    hipEventSynchronize(end);
    float ms = 0;
    hipEventElapsedTime(&ms, begin, end);
    double const moves = (double)blocks * 256 * rounds * 64; // DPP moves (each followed by an addition)
    std::printf("{\"control\": \"%s\", \"ms\": %.3f, \"T_lane_moves_per_s\": %.2f}\n", name, ms, moves / (ms * 1e-3) / 1e12);
    return ms;
}

int main() {
    unsigned *device = nullptr;
    hipMalloc(&device, 256 * 8 * 256 * sizeof(unsigned));
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, device);
    std::vector<unsigned> host(192);
    hipMemcpy(host.data(), device, 192 * sizeof(unsigned), hipMemcpyDeviceToHost);
    int wrong_wave = 0, wrong_row = 0, wrong_left = 0;
    for (unsigned lane = 0; lane < 64; ++lane) {
        wrong_wave += host[lane] != (lane ? 1000u + lane - 1 : 7u);
        wrong_row += host[64 + lane] != (lane % 16 ? 1000u + lane - 1 : 7u);
        wrong_left += host[128 + lane] != (lane < 63 ? 1000u + lane + 1 : 7u);
    }
    std::printf("{\"wave_shr_1_wrong_lanes\": %d, \"row_shr_1_wrong_lanes\": %d, \"wave_shl_1_wrong_lanes\": %d, \"lane16\": %u, \"lane32\": %u, \"lane48\": %u, \"lane0\": %u}\n",
                wrong_wave, wrong_row, wrong_left, host[16], host[32], host[48], host[0]);
    rate<0x111>(device, "row_shr:1");
    rate<0x138>(device, "wave_shr:1");
    return wrong_wave != 0;
}
