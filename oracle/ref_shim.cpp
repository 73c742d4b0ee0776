/**
 *  oracle/ref_shim.cpp - C-ABI shim over the REAL reference engines, compiled from the sources where they lie
 *  under /root/reference (never copied into this repo).  Output goes to oracle/_ref/libszs_ref.so only.
 *
 *  TEST INFRASTRUCTURE ONLY: used by tests/ to pin oracle/sz_oracle.c against the reference itself, and by
 *  bench.py's `cpu_baseline` leg (kind "reference").  The product library never links or loads this file.
 *
 *  What it instantiates (reference file:line):
 *    - levenshtein_serial_t / affine_levenshtein_serial_t            include/stringzillas/similarities/serial.hpp:677,685
 *    - needleman_wunsch_serial_t / affine_needleman_wunsch_serial_t  serial.hpp:679,687
 *    - smith_waterman_serial_t / affine_smith_waterman_serial_t      serial.hpp:681,689
 *    - the *_icelake_t and *_haswell_t siblings                      serial.hpp:696-725 (bodies in icelake.hpp / haswell.hpp)
 *    - error_costs_32x32_t::blosum62() / nuc44()                     serial.hpp:221-287
 *
 *    - floating_rolling_hashers<sz_cap_serial_k, 64> and basic_rolling_hashers<floating_rolling_hasher<f64_t>, u32_t>
 *                                                                    include/stringzillas/fingerprints/serial.hpp:1119,646
 *      composed exactly as the reference's C shim composes them    c/stringzillas/fingerprints.cuh:49-176
 *
 *  ForkUnion (the reference's thread pool) is an absent submodule, so rows are sharded over std::thread here,
 *  one engine instance per thread (the reference's engines are not re-entrant: serial.hpp:3694-3745).
 */
#include <cstdint>
#include <cstring>
#include <string_view>
#include <thread>
#include <vector>

#include <stringzillas/fingerprints.hpp>
#include <stringzillas/similarities.hpp>

namespace szs = ashvardanian::stringzillas;

namespace {

using views_t = std::vector<std::string_view>;

views_t views_from_tape(char const *data, uint64_t const *offsets, size_t count) {
    views_t views(count);
    for (size_t i = 0; i != count; ++i) views[i] = std::string_view(data + offsets[i], offsets[i + 1] - offsets[i]);
    return views;
}

enum tier_t { tier_serial = 0, tier_haswell = 1, tier_icelake = 2 };

int best_tier() {
#if defined(__x86_64__)
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") &&
        __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vbmi") && __builtin_cpu_supports("bmi2"))
        return tier_icelake;
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma") && __builtin_cpu_supports("bmi2"))
        return tier_haswell;
#endif
    return tier_serial;
}

/** Runs `make_engine()(queries[rows], candidates, results)` over `threads` contiguous row blocks balanced by bytes. */
template <typename value_t, typename make_engine_t>
int run_cross(make_engine_t make_engine, views_t const &queries, views_t const *candidates, value_t *results,
              size_t stride, int threads) {
    if (!candidates) { // symmetric: the reference computes the lower triangle and mirrors it (serial.hpp:3169-3182)
        auto engine = make_engine();
        szs::strided_rows<value_t> rows {results, queries.size(), queries.size(), stride};
        return (int)engine(queries, rows);
    }
    size_t const count = queries.size();
    if (threads <= 1 || count < 2) {
        auto engine = make_engine();
        szs::strided_rows<value_t> rows {results, count, candidates->size(), stride};
        return (int)engine(queries, *candidates, rows);
    }
    size_t total_bytes = 0;
    for (auto const &q : queries) total_bytes += q.size() + 1;
    std::vector<size_t> cuts {0};
    size_t running = 0, next_cut = 1;
    for (size_t i = 0; i != count; ++i) {
        running += queries[i].size() + 1;
        while (next_cut < (size_t)threads && running * threads >= total_bytes * next_cut) cuts.push_back(i + 1), ++next_cut;
    }
    while (cuts.size() < (size_t)threads + 1) cuts.push_back(count);
    cuts.back() = count;
    std::vector<int> statuses(threads, 0);
    std::vector<std::thread> pool;
    for (int t = 0; t != threads; ++t) {
        pool.emplace_back([&, t] {
            size_t const first = cuts[t], last = cuts[t + 1];
            if (first >= last) return;
            views_t block(queries.begin() + first, queries.begin() + last);
            auto engine = make_engine();
            szs::strided_rows<value_t> rows {results + first * stride, last - first, candidates->size(), stride};
            statuses[t] = (int)engine(block, *candidates, rows);
        });
    }
    for (auto &worker : pool) worker.join();
    for (int status : statuses)
        if (status) return status;
    return 0;
}

template <typename value_t, typename serial_t, typename haswell_t, typename icelake_t, typename... args_t>
int dispatch_tier(int tier, views_t const &queries, views_t const *candidates, value_t *results, size_t stride,
                  int threads, args_t const &...args) {
    if (tier > best_tier()) tier = best_tier();
#if SZ_USE_ICELAKE
    if (tier == tier_icelake)
        return run_cross<value_t>([&] { return icelake_t {args...}; }, queries, candidates, results, stride, threads);
#endif
#if SZ_USE_HASWELL
    if (tier >= tier_haswell)
        return run_cross<value_t>([&] { return haswell_t {args...}; }, queries, candidates, results, stride, threads);
#endif
    return run_cross<value_t>([&] { return serial_t {args...}; }, queries, candidates, results, stride, threads);
}

szs::error_costs_32x32_t costs_from(uint8_t const *byte_to_class, int8_t const *class_costs) {
    szs::error_costs_32x32_t costs;
    std::memcpy(costs.byte_to_class, byte_to_class, 256);
    std::memcpy(costs.class_substitution_costs, class_costs, 32 * 32);
    return costs;
}

} // namespace

/** One block of texts through the 64-dimension slices of one tier's hashers (szs_ref_fingerprints_tiered below). */
template <sz_capability_t capability_>
static int fingerprints_sliced(size_t dimensions, size_t alphabet_size, size_t const *window_widths, size_t window_widths_count, uint64_t seed,
                               views_t const &texts, size_t first, size_t last, uint32_t *min_hashes, uint32_t *min_counts) {
    constexpr size_t slice = 64;
    using hashers_t = szs::floating_rolling_hashers<capability_, slice>;
    using byte_span_t = ashvardanian::stringzilla::span<ashvardanian::stringzilla::byte_t const>;
    for (size_t i = 0; i != dimensions / slice; ++i) {
        hashers_t hashers;
        if ((int)hashers.try_seed(window_widths[i % window_widths_count], alphabet_size, i * slice, seed) != 0) return -1;
        for (size_t t = first; t != last; ++t) {
            byte_span_t text {reinterpret_cast<ashvardanian::stringzilla::byte_t const *>(texts[t].data()), texts[t].size()};
            hashers.fingerprint(text, typename hashers_t::min_hashes_span_t {min_hashes + t * dimensions + i * slice},
                                typename hashers_t::min_counts_span_t {min_counts + t * dimensions + i * slice});
        }
    }
    return 0;
}


extern "C" {

/** Highest SIMD tier this host CPU can run: 0 serial, 1 Haswell (AVX2), 2 Ice Lake (AVX-512 VBMI). */
int szs_ref_best_tier(void) { return best_tier(); }

/**
 *  Cross-product (or symmetric, when `c_data == NULL`) Levenshtein distances through the reference engines.
 *  Offsets are 64-bit tapes with count+1 entries.  `tier`: 0 serial, 1 Haswell, 2 Ice Lake (clamped to the host).
 */
int szs_ref_levenshtein(int tier, int threads, int8_t match, int8_t mismatch, int8_t open, int8_t extend,
                        char const *q_data, uint64_t const *q_offsets, size_t q_count, //
                        char const *c_data, uint64_t const *c_offsets, size_t c_count, //
                        size_t *results, size_t stride) {
    views_t queries = views_from_tape(q_data, q_offsets, q_count);
    views_t candidates_storage;
    views_t const *candidates = nullptr;
    if (c_data || c_offsets) candidates_storage = views_from_tape(c_data, c_offsets, c_count), candidates = &candidates_storage;
    szs::uniform_substitution_costs_t subs {match, mismatch};
    if (open == extend) // mirrors the linear/affine fork of c/stringzillas/levenshtein.cuh:117
        return dispatch_tier<size_t, szs::levenshtein_serial_t, szs::levenshtein_haswell_t, szs::levenshtein_icelake_t>(
            tier, queries, candidates, results, stride, threads, subs, szs::linear_gap_costs_t {open});
    return dispatch_tier<size_t, szs::affine_levenshtein_serial_t, szs::affine_levenshtein_haswell_t,
                         szs::affine_levenshtein_icelake_t>(tier, queries, candidates, results, stride, threads, subs,
                                                            szs::affine_gap_costs_t {open, extend});
}

/**
 *  Codepoint-level Levenshtein through the reference's UTF-8 engines (serial.hpp:678,685; SIMD siblings exist for
 *  linear gaps only, mirroring the ladder of c/stringzillas/levenshtein.cuh:266-312).
 */
int szs_ref_levenshtein_utf8(int tier, int threads, int8_t match, int8_t mismatch, int8_t open, int8_t extend,
                             char const *q_data, uint64_t const *q_offsets, size_t q_count, //
                             char const *c_data, uint64_t const *c_offsets, size_t c_count, //
                             size_t *results, size_t stride) {
    views_t queries = views_from_tape(q_data, q_offsets, q_count);
    views_t candidates_storage;
    views_t const *candidates = nullptr;
    if (c_data || c_offsets) candidates_storage = views_from_tape(c_data, c_offsets, c_count), candidates = &candidates_storage;
    szs::uniform_substitution_costs_t subs {match, mismatch};
    if (open == extend)
        return dispatch_tier<size_t, szs::levenshtein_utf8_serial_t, szs::levenshtein_utf8_haswell_t,
                             szs::levenshtein_utf8_icelake_t>(tier, queries, candidates, results, stride, threads, subs,
                                                              szs::linear_gap_costs_t {open});
    return run_cross<size_t>([&] { return szs::affine_levenshtein_utf8_serial_t {subs, szs::affine_gap_costs_t {open, extend}}; },
                             queries, candidates, results, stride, threads);
}

int szs_ref_needleman_wunsch(int tier, int threads, uint8_t const *byte_to_class, int8_t const *class_costs,
                             int8_t open, int8_t extend,                                      //
                             char const *q_data, uint64_t const *q_offsets, size_t q_count, //
                             char const *c_data, uint64_t const *c_offsets, size_t c_count, //
                             ptrdiff_t *results, size_t stride) {
    views_t queries = views_from_tape(q_data, q_offsets, q_count);
    views_t candidates_storage;
    views_t const *candidates = nullptr;
    if (c_data || c_offsets) candidates_storage = views_from_tape(c_data, c_offsets, c_count), candidates = &candidates_storage;
    szs::error_costs_32x32_t subs = costs_from(byte_to_class, class_costs);
    if (open == extend) // c/stringzillas/needleman_wunsch.cuh:99-118
        return dispatch_tier<ptrdiff_t, szs::needleman_wunsch_serial_t, szs::needleman_wunsch_haswell_t,
                             szs::needleman_wunsch_icelake_t>(tier, queries, candidates, results, stride, threads, subs,
                                                              szs::linear_gap_costs_t {open});
    return dispatch_tier<ptrdiff_t, szs::affine_needleman_wunsch_serial_t, szs::affine_needleman_wunsch_haswell_t,
                         szs::affine_needleman_wunsch_icelake_t>(tier, queries, candidates, results, stride, threads,
                                                                 subs, szs::affine_gap_costs_t {open, extend});
}

int szs_ref_smith_waterman(int tier, int threads, uint8_t const *byte_to_class, int8_t const *class_costs,
                           int8_t open, int8_t extend,                                      //
                           char const *q_data, uint64_t const *q_offsets, size_t q_count, //
                           char const *c_data, uint64_t const *c_offsets, size_t c_count, //
                           ptrdiff_t *results, size_t stride) {
    views_t queries = views_from_tape(q_data, q_offsets, q_count);
    views_t candidates_storage;
    views_t const *candidates = nullptr;
    if (c_data || c_offsets) candidates_storage = views_from_tape(c_data, c_offsets, c_count), candidates = &candidates_storage;
    szs::error_costs_32x32_t subs = costs_from(byte_to_class, class_costs);
    if (open == extend)
        return dispatch_tier<ptrdiff_t, szs::smith_waterman_serial_t, szs::smith_waterman_haswell_t,
                             szs::smith_waterman_icelake_t>(tier, queries, candidates, results, stride, threads, subs,
                                                            szs::linear_gap_costs_t {open});
    return dispatch_tier<ptrdiff_t, szs::affine_smith_waterman_serial_t, szs::affine_smith_waterman_haswell_t,
                         szs::affine_smith_waterman_icelake_t>(tier, queries, candidates, results, stride, threads, subs,
                                                               szs::affine_gap_costs_t {open, extend});
}

/** Exports the reference's own BLOSUM62 (which=0) or NUC.4.4 (which=1) compact tables (serial.hpp:221-287). */
void szs_ref_substitution_table(int which, uint8_t *byte_to_class, int8_t *class_costs) {
    szs::error_costs_32x32_t costs = which == 0 ? szs::error_costs_32x32_t::blosum62() : szs::error_costs_32x32_t::nuc44();
    std::memcpy(byte_to_class, costs.byte_to_class, 256);
    std::memcpy(class_costs, costs.class_substitution_costs, 32 * 32);
}

/**
 *  MinHash fingerprints + Count-Min counts of `count` texts (u64 tape), through the reference's own serial engines, with
 *  the same choice of engine the reference's `szs_fingerprints_init` makes (c/stringzillas/fingerprints.cuh:49-176):
 *  slices of 64 dimensions sharing one window width when `dimensions` is a whole multiple of 64 x widths, else one
 *  per-dimension hasher with interleaved widths.  Rows are `dimensions` consecutive u32.
 */
int szs_ref_fingerprints(size_t dimensions, size_t alphabet_size, size_t const *window_widths, size_t window_widths_count,
                         uint64_t seed, char const *data, uint64_t const *offsets, size_t count, uint32_t *min_hashes,
                         uint32_t *min_counts) {
    constexpr size_t slice = 64;
    size_t const default_widths[] = {3, 4, 5, 7, 9, 11, 15, 31};
    if (!window_widths || !window_widths_count) window_widths = default_widths, window_widths_count = 8;
    if (!alphabet_size) alphabet_size = 256;
    views_t const texts = views_from_tape(data, offsets, count);
    size_t const per_width_min = dimensions / window_widths_count;
    size_t const per_width_max = (dimensions + window_widths_count - 1) / window_widths_count;
    bool const sliced = per_width_min == per_width_max && per_width_min % slice == 0;
    using byte_span_t = ashvardanian::stringzilla::span<ashvardanian::stringzilla::byte_t const>;
    if (sliced) {
        using hashers_t = szs::floating_rolling_hashers<sz_cap_serial_k, slice>;
        for (size_t i = 0; i != dimensions / slice; ++i) {
            hashers_t hashers;
            if ((int)hashers.try_seed(window_widths[i % window_widths_count], alphabet_size, i * slice, seed) != 0) return -1;
            for (size_t t = 0; t != count; ++t) {
                byte_span_t text {reinterpret_cast<ashvardanian::stringzilla::byte_t const *>(texts[t].data()), texts[t].size()};
                hashers.fingerprint(text, typename hashers_t::min_hashes_span_t {min_hashes + t * dimensions + i * slice},
                                    typename hashers_t::min_counts_span_t {min_counts + t * dimensions + i * slice});
            }
        }
        return 1;
    }
    szs::basic_rolling_hashers<szs::floating_rolling_hasher<double>, uint32_t> hashers;
    for (size_t d = 0; d != dimensions; ++d)
        if ((int)hashers.try_extend(window_widths[d % window_widths_count], 1, alphabet_size, seed) != 0) return -1;
    for (size_t t = 0; t != count; ++t) {
        byte_span_t text {reinterpret_cast<ashvardanian::stringzilla::byte_t const *>(texts[t].data()), texts[t].size()};
        ashvardanian::stringzilla::span<uint32_t> hashes {min_hashes + t * dimensions, dimensions};
        ashvardanian::stringzilla::span<uint32_t> counts {min_counts + t * dimensions, dimensions};
        if ((int)hashers.try_fingerprint(text, hashes, counts) != 0) return -1;
    }
    return 2;
}

/**
 *  The same fingerprints through the reference's SIMD engines - the CPU baseline bench.py times beside `szs_fingerprints_u32tape`:
 *  `floating_rolling_hashers<sz_cap_skylake_k / sz_cap_haswell_k / sz_cap_serial_k, 64>` (fingerprints/skylake.hpp:49,
 *  haswell.hpp:42, serial.hpp:1120), slices of 64 dimensions sharing one window width as the reference's C shim composes them
 *  (c/stringzillas/fingerprints.cuh:49-176), the texts dealt over `threads` std::threads in contiguous blocks (each thread seeds its
 *  own hashers: they hold state).  `dimensions` must be a whole multiple of 64 x widths.  Returns the tier that ran, -1 on failure.
 */
int szs_ref_fingerprints_tiered(int tier, int threads, size_t dimensions, size_t alphabet_size, size_t const *window_widths,
                                size_t window_widths_count, uint64_t seed, char const *data, uint64_t const *offsets, size_t count,
                                uint32_t *min_hashes, uint32_t *min_counts) {
    size_t const default_widths[] = {3, 4, 5, 7, 9, 11, 15, 31};
    if (!window_widths || !window_widths_count) window_widths = default_widths, window_widths_count = 8;
    if (!alphabet_size) alphabet_size = 256;
    if (dimensions % (64 * window_widths_count)) return -1;
    int const best = best_tier();
    if (tier > best) tier = best;
    views_t const texts = views_from_tape(data, offsets, count);
    if (threads < 1) threads = 1;
    if ((size_t)threads > count) threads = (int)(count ? count : 1);
    std::vector<int> outcomes((size_t)threads, 0);
    auto run = [&](int thread) {
        size_t const first = count * (size_t)thread / (size_t)threads, last = count * ((size_t)thread + 1) / (size_t)threads;
#if SZ_USE_SKYLAKE
        if (tier >= tier_icelake) {
            outcomes[thread] = fingerprints_sliced<sz_cap_skylake_k>(dimensions, alphabet_size, window_widths, window_widths_count, seed, texts, first, last, min_hashes, min_counts);
            return;
        }
#endif
#if SZ_USE_HASWELL
        if (tier >= tier_haswell) {
            outcomes[thread] = fingerprints_sliced<sz_cap_haswell_k>(dimensions, alphabet_size, window_widths, window_widths_count, seed, texts, first, last, min_hashes, min_counts);
            return;
        }
#endif
        outcomes[thread] = fingerprints_sliced<sz_cap_serial_k>(dimensions, alphabet_size, window_widths, window_widths_count, seed, texts, first, last, min_hashes, min_counts);
    };
    std::vector<std::thread> pool;
    for (int thread = 1; thread < threads; ++thread) pool.emplace_back(run, thread);
    run(0);
    for (auto &worker : pool) worker.join();
    for (int outcome : outcomes)
        if (outcome) return -1;
    return tier;
}

} // extern "C"
