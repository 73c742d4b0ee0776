// launch_latency.hip - what one synchronous C-ABI call costs on this box BESIDES its kernels: the fixed prices of the HIP
// runtime that bound `host_overhead_ms_per_step` of bench.py from below.  Prints one JSON object of microseconds.
//   hipcc --offload-arch=gfx950 -O2 scripts/launch_latency.hip -o scripts/bin/launch_latency
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

__global__ void empty_kernel() {}
__global__ void spin_kernel(long long ticks) { // ~ticks x 10 ns of device time (100 MHz constant clock)
    long long const start = wall_clock64();
    while (wall_clock64() - start < ticks) {}
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename body_t>
static double median_us(int repeats, body_t body) {
    std::vector<double> samples;
    for (int i = 0; i < repeats; ++i) {
        double const start = now_us();
        body();
        samples.push_back(now_us() - start);
    }
    std::sort(samples.begin(), samples.end());
    return samples[samples.size() / 2];
}

int main() {
    hipStream_t stream;
    hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    hipEvent_t start, stop;
    hipEventCreate(&start), hipEventCreate(&stop);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, stream);
    hipStreamSynchronize(stream);
    int const repeats = 300;
    long long const spin = 20000; // 200 us, config 2's kernel

    double const sync_idle = median_us(repeats, [&] { hipStreamSynchronize(stream); });
    double const one_empty = median_us(repeats, [&] {
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, stream);
        hipStreamSynchronize(stream);
    });
    double const two_empty = median_us(repeats, [&] {
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, stream);
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, stream);
        hipStreamSynchronize(stream);
    });
    double const one_with_events = median_us(repeats, [&] {
        hipEventRecord(start, stream);
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, stream);
        hipEventRecord(stop, stream);
        hipStreamSynchronize(stream);
    });
    double const spin_plain = median_us(repeats, [&] {
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, stream, spin);
        hipStreamSynchronize(stream);
    });
    double const spin_with_events = median_us(repeats, [&] {
        hipEventRecord(start, stream);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, stream, spin);
        hipEventRecord(stop, stream);
        hipStreamSynchronize(stream);
    });
    float event_ms = 0;
    hipEventElapsedTime(&event_ms, start, stop);
    double const planner_like = median_us(repeats, [&] { // a small kernel in front of the long one, events around the long one
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(1024), 0, stream);
        hipEventRecord(start, stream);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, stream, spin);
        hipEventRecord(stop, stream);
        hipStreamSynchronize(stream);
    });
    double const elapsed_query = median_us(repeats, [&] { hipEventElapsedTime(&event_ms, start, stop); });
    void *probe = nullptr;
    hipMalloc(&probe, 4096);
    hipPointerAttribute_t attributes;
    double const pointer_query = median_us(repeats, [&] { hipPointerGetAttributes(&attributes, probe); });
    char pinned_host[64];
    void *pinned = nullptr;
    hipHostMalloc(&pinned, 4096, hipHostMallocDefault);
    double const small_d2h = median_us(repeats, [&] {
        hipMemcpyAsync(pinned, probe, 4096, hipMemcpyDeviceToHost, stream);
        hipStreamSynchronize(stream);
    });
    (void)pinned_host;
    printf("{\"sync_idle_us\": %.2f, \"launch_empty_sync_us\": %.2f, \"launch_two_empty_sync_us\": %.2f, \"launch_empty_with_events_sync_us\": %.2f, "
           "\"spin200us_sync_us\": %.2f, \"spin200us_with_events_sync_us\": %.2f, \"spin200us_event_ms\": %.4f, "
           "\"small_kernel_then_spin_with_events_us\": %.2f, \"event_elapsed_query_us\": %.2f, \"pointer_attributes_us\": %.2f, "
           "\"d2h_4k_sync_us\": %.2f}\n",
           sync_idle, one_empty, two_empty, one_with_events, spin_plain, spin_with_events, event_ms, planner_like, elapsed_query,
           pointer_query, small_d2h);
    return 0;
}
