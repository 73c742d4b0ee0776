/*
 *  occupancy_census.hip - how many 256-thread workgroups does an MI355X CU really run at once, and how evenly does a
 *  ~2048-workgroup grid (one launch of the Myers kernel on the 1024 x 1024 config) spread over the 256 CUs?
 *
 *  Every workgroup records {XCC id, HW id, start, end} (s_memrealtime, 100 MHz) around a fixed VALU loop; the host
 *  then reconstructs per-CU concurrency.  Resource shape mirrors levenshtein_myers_kernel<4,4>: <= 64 VGPRs, 4 KiB LDS.
 *
 *      hipcc --offload-arch=gfx950 -O2 scripts/occupancy_census.hip -o scripts/bin/occupancy_census
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct record_t {
    uint32_t xcc, hw_id;
    uint64_t start, end;
};

template <int chains_>
__global__ __launch_bounds__(1024) void census(record_t *records, uint32_t *sink, int iterations) {
    extern __shared__ uint32_t lds[];
    uint64_t const start = __builtin_readcyclecounter();
    uint64_t const start_real = wall_clock64();
    lds[threadIdx.x & 255] = threadIdx.x;
    __syncthreads();
    uint32_t x[chains_];
    for (int c = 0; c < chains_; ++c) x[c] = threadIdx.x * (c + 3) + blockIdx.x;
    for (int i = 0; i < iterations; ++i) {
#pragma unroll
        for (int k = 0; k < 256 / chains_; ++k) {
#pragma unroll
            for (int c = 0; c < chains_; ++c) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(x[c]) : "v"(x[(c + 1) % chains_]));
        }
    }
    uint32_t sum = 0;
    for (int c = 0; c < chains_; ++c) sum ^= x[c];
    if (sum == 0x12345u) sink[0] = sum + lds[(threadIdx.x + 1) & 255];
    uint64_t const end_real = wall_clock64();
    (void)start;
    if ((threadIdx.x & 63) == 0) { /* one record per WAVE */
        uint32_t xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        records[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = {xcc, hw, start_real, end_real};
    }
}

int main(int argc, char **argv) {
    int const blocks = argc > 1 ? atoi(argv[1]) : 2028;
    int const lds_bytes = argc > 2 ? atoi(argv[2]) : 4096;
    int const iterations = argc > 3 ? atoi(argv[3]) : 30;
    record_t *records;
    uint32_t *sink;
    int const threads = argc > 5 ? atoi(argv[5]) : 256;
    hipMalloc(&records, sizeof(record_t) * blocks * (threads / 64));
    hipMalloc(&sink, 64);
    int api_blocks = 0;
    int const fat = argc > 4 ? atoi(argv[4]) : 0; /* 1: ~60 live VGPRs like the Myers kernel */
    auto kernel = fat ? census<56> : census<8>;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&api_blocks, kernel, threads, lds_bytes);
    for (int repeat = 0; repeat < 2; ++repeat) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, 0, records, sink, iterations);
        hipDeviceSynchronize();
    }
    int const waves = blocks * (threads / 64);
    record_t *host = (record_t *)malloc(sizeof(record_t) * waves);
    hipMemcpy(host, records, sizeof(record_t) * waves, hipMemcpyDeviceToHost);
    uint64_t first = ~0ull, last = 0, latest_start = 0;
    for (int w = 0; w < waves; ++w) {
        if (host[w].start < first) first = host[w].start;
        if (host[w].end > last) last = host[w].end;
        if (host[w].start > latest_start) latest_start = host[w].start;
    }
    /* SIMD key out of XCC_ID and HW_ID (gfx9 layout): simd [5:4], cu [11:8], sh [12], se [15:13]. */
    enum { keys = 8 * 8 * 2 * 16 * 4 };
    static int resident_now[keys], resident_max[keys], seen[keys];
    /* sweep: sort events by time (start = +1, end = -1); simple O(n log n) via qsort on encoded events */
    typedef struct { uint64_t time; int key, delta; } event_t;
    event_t *events = (event_t *)malloc(sizeof(event_t) * 2 * waves);
    for (int w = 0; w < waves; ++w) {
        uint32_t const hw = host[w].hw_id;
        int const key = (int)((((host[w].xcc & 7) * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15)) * 4 + (int)((hw >> 4) & 3);
        events[2 * w] = {host[w].start, key, +1};
        events[2 * w + 1] = {host[w].end, key, -1};
        seen[key] = 1;
    }
    qsort(events, 2 * waves, sizeof(event_t), [](void const *a, void const *b) -> int {
        event_t const *x = (event_t const *)a, *y = (event_t const *)b;
        if (x->time != y->time) return x->time < y->time ? -1 : 1;
        return x->delta - y->delta; /* ends before starts at equal time */
    });
    for (int e = 0; e < 2 * waves; ++e) {
        resident_now[events[e].key] += events[e].delta;
        if (resident_now[events[e].key] > resident_max[events[e].key]) resident_max[events[e].key] = resident_now[events[e].key];
    }
    int simds_seen = 0, histogram[33] = {0};
    for (int k = 0; k < keys; ++k)
        if (seen[k]) simds_seen++, histogram[resident_max[k] > 32 ? 32 : resident_max[k]]++;
    double mean_life = 0;
    for (int w = 0; w < waves; ++w) mean_life += (double)(host[w].end - host[w].start);
    mean_life /= waves;
    printf("{\"fat_vgprs\": %d, \"blocks\": %d, \"threads\": %d, \"lds_bytes\": %d, \"api_blocks_per_cu\": %d, \"simds_seen\": %d, "
           "\"launch_ticks\": %llu, \"latest_start_ticks\": %llu, \"mean_wave_life_ticks\": %.0f, \"max_resident_waves_per_simd_histogram\": {",
           fat, blocks, threads, lds_bytes, api_blocks, simds_seen, (unsigned long long)(last - first),
           (unsigned long long)(latest_start - first), mean_life);
    int printed = 0;
    for (int n = 0; n <= 32; ++n)
        if (histogram[n]) printf("%s\"%d\": %d", printed++ ? ", " : "", n, histogram[n]);
    printf("}, \"tick\": \"wall_clock64 (100 MHz)\"}\n");
    return 0;
}
