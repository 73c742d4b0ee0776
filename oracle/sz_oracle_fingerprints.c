/*
 *  sz_oracle_fingerprints.c - CPU restatement of the reference's rolling MinHash / Count-Min fingerprints.
 *  TEST INFRASTRUCTURE: linked into oracle/libsz_oracle.so; only tests/, smoke() and bench cpu-baseline legs may use it.
 *
 *  What it restates (all /root/reference/include/stringzillas/fingerprints/serial.hpp unless noted):
 *    - splitmix64                                                       :44-50
 *    - per-dimension parameters from `seed + dim`: multiplier in [256, 640), modulo = 4503599626977 - (mix % 2^20),
 *      inverse modulo, -(multiplier^(width-1) mod modulo)              :495-534,  :510-512 (the fmod loop)
 *    - the rolling update of the 64-dimension engine: remove tail, Barrett-reduce, add head, Barrett-reduce;
 *      minimum over the full 52-bit state, count of its occurrences    :1234-1265, Barrett :1328-1338
 *    - export: low 32 bits of the minimum; 0xFFFFFFFF and count 0 when the text is shorter than the window
 *                                                                       :1188-1192, :1268-1277
 *    - which width a dimension gets: `widths[(dim / 64) % count]` when `dimensions` is a whole multiple of 64 x widths,
 *      else `widths[dim % count]`; default widths {3,4,5,7,9,11,15,31}, alphabet 0 -> 256
 *                                                                       c/stringzillas/fingerprints.cuh:31-62,128-176
 *  Every intermediate is an integer below 2^52 held exactly in a double, so the only inexact operation is the product
 *  `x * inverse_modulo` inside the Barrett step, whose rounding the two fix-ups absorb: the results are the canonical
 *  residues whatever the order of the exact operations - which is why the reference's two engine variants (and the GPU
 *  kernel) agree bit for bit.  Pinned against the reference itself in tests/test_oracle.py.
 */
#include "sz_oracle.h"

#include <math.h>
#include <stdlib.h>

static uint64_t szo_splitmix64(uint64_t state) {
    state += 0x9E3779B97F4A7C15ull;
    uint64_t z = state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void szo_fingerprint_parameters(size_t dimensions, size_t const *window_widths, size_t window_widths_count, uint64_t seed,
                                size_t *widths, double *multipliers, double *modulos, double *inverse_modulos,
                                double *negative_discarding_multipliers) {
    static size_t const default_widths[] = {3, 4, 5, 7, 9, 11, 15, 31};
    if (!window_widths || !window_widths_count) window_widths = default_widths, window_widths_count = 8;
    size_t const per_width_min = dimensions / window_widths_count;
    size_t const per_width_max = (dimensions + window_widths_count - 1) / window_widths_count;
    int const sliced = per_width_min == per_width_max && per_width_min % 64 == 0;
    for (size_t dim = 0; dim < dimensions; ++dim) {
        size_t const width = window_widths[(sliced ? dim / 64 : dim) % window_widths_count];
        double const multiplier = (double)(256ull + szo_splitmix64(seed + dim) % 384ull);
        double const modulo = (double)(4503599626977ull - szo_splitmix64(szo_splitmix64(seed + dim)) % (1ull << 20));
        double power = 1.0; /* multiplier^(width - 1) mod modulo */
        for (size_t i = 0; i + 1 < width; ++i) power = fmod(power * multiplier, modulo);
        widths[dim] = width, multipliers[dim] = multiplier, modulos[dim] = modulo, inverse_modulos[dim] = 1.0 / modulo;
        negative_discarding_multipliers[dim] = -power;
    }
}

static double szo_barrett(double x, double modulo, double inverse_modulo) {
    double const q = floor(x * inverse_modulo);
    double result = x - q * modulo;
    if (result < 0.0) result += modulo;
    if (result >= modulo) result -= modulo;
    return result;
}

void szo_fingerprints_cross(char const *data, uint64_t const *offsets, size_t count, size_t dimensions, size_t alphabet_size,
                            size_t const *window_widths, size_t window_widths_count, uint64_t seed, uint32_t *min_hashes,
                            uint32_t *min_counts) {
    (void)alphabet_size; /* the reference derives nothing from it for the f64 hasher (serial.hpp:524-533) */
    size_t *widths = malloc(dimensions * sizeof(size_t));
    double *parameters = malloc(dimensions * 4 * sizeof(double));
    double *multipliers = parameters, *modulos = parameters + dimensions, *inverses = parameters + 2 * dimensions,
           *discarding = parameters + 3 * dimensions;
    szo_fingerprint_parameters(dimensions, window_widths, window_widths_count, seed, widths, multipliers, modulos, inverses,
                               discarding);
    for (size_t t = 0; t < count; ++t) {
        unsigned char const *text = (unsigned char const *)data + offsets[t];
        size_t const length = offsets[t + 1] - offsets[t];
        for (size_t dim = 0; dim < dimensions; ++dim) {
            size_t const width = widths[dim];
            uint32_t *hash_out = min_hashes + t * dimensions + dim, *count_out = min_counts + t * dimensions + dim;
            if (length < width) {
                *hash_out = 0xFFFFFFFFu, *count_out = 0;
                continue;
            }
            double state = 0;
            for (size_t i = 0; i < width; ++i) state = szo_barrett(state * multipliers[dim] + (text[i] + 1.0), modulos[dim], inverses[dim]);
            double minimum = state;
            uint32_t occurrences = 1;
            for (size_t i = width; i < length; ++i) {
                state = szo_barrett(discarding[dim] * (text[i - width] + 1.0) + state, modulos[dim], inverses[dim]);
                state = szo_barrett(state * multipliers[dim] + (text[i] + 1.0), modulos[dim], inverses[dim]);
                if (state < minimum) minimum = state, occurrences = 1;
                else if (state == minimum) ++occurrences;
            }
            *hash_out = (uint32_t)((uint64_t)minimum & 0xFFFFFFFFull), *count_out = occurrences;
        }
    }
    free(widths), free(parameters);
}
