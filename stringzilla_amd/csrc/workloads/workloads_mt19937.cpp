// csrc/workloads/workloads_mt19937.cpp - the synthetic batches of SURVEY.md section 8(d) from std::mt19937_64, so that a C++
// program can reproduce them without this repository's Python: `workloads.config(n, generator="mt19937_64")` calls in here.
//
// The mapping from the engine to strings is spelled out (the standard leaves uniform_int_distribution to the implementation):
//   one engine per tape, seeded with  seed = 1000 x config + side  (side: 0 queries, 1 candidates);
//   lengths first:  length[i] = low + engine() % (high - low + 1),  i = 0 .. count - 1;
//   then the bytes: byte[j]  = alphabet[engine() % alphabet_size],  j over the whole tape in order.
// Built into stringzilla_amd/lib/libszs_workloads_mt19937.so beside the scoring library (csrc/Makefile); the scoring library
// itself neither links nor loads it - only stringzilla_amd/workloads.py does, for bench.py and the tests.
#include <cstdint>
#include <random>

extern "C" {

/** Fills `offsets[count + 1]` (u32) and returns the tape's byte size; call again with `data` to fill the bytes. */
uint64_t szs_workload_mt19937_64(uint64_t seed, uint32_t count, uint32_t low, uint32_t high, uint8_t const *alphabet,
                                 uint32_t alphabet_size, uint32_t *offsets, uint8_t *data) {
    std::mt19937_64 engine(seed);
    uint64_t total = 0;
    offsets[0] = 0;
    for (uint32_t i = 0; i < count; ++i) {
        total += low + engine() % ((uint64_t)high - low + 1);
        offsets[i + 1] = (uint32_t)total;
    }
    if (data)
        for (uint64_t j = 0; j < total; ++j) data[j] = alphabet[engine() % alphabet_size];
    return total;
}
}
