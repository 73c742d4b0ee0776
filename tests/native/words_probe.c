/*
 *  words_probe.c - the tiny-token regime (hip/myers_tiny.hip) driven from plain C, no Python, no torch: batches of word-like
 *  tokens in DEVICE memory, a results matrix with padding columns, every cell checked against the CPU oracle
 *  (oracle/sz_oracle.c), kernel and wall times of a stream of fresh batches.  Test infrastructure: a GPU visit exercises and
 *  times the path in seconds (tests/test_gpu_round5.py runs it; scripts/measure_words.sh times it on the repository's prose).
 *
 *      words_probe SOURCE Q C [REPEATS] [PADDING]
 *  SOURCE = file:PATH            words of that file (split at white space), drawn at random
 *         | mix:PERMILLE:LONGEST tokens of 0 ... 16 bytes, PERMILLE in a thousand of 17 ... LONGEST bytes instead
 *  Two batches of the same counts take turns (a fresh batch every call); SZS_ROCM_TINY picks the path as usual.
 *  PROBE_WIDE=1: 64-bit offsets.  PROBE_NO_ORACLE=1: times only.  PROBE_UTF8=1: the codepoint engine (`mix` tokens then hold
 *  two- and three-byte runes and a few bytes that are not UTF-8; lengths are in BYTES).  PROBE_SYMMETRIC=1: the queries against
 *  themselves (C is ignored).
 */
#define _POSIX_C_SOURCE 200809L
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "../../include/stringzillas/stringzillas.h"
#include "../../include/stringzillas/stringzillas_rocm.h"
#include "../../oracle/sz_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rng(void) {
    rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17;
    return rng_state;
}
static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static char const *stage = "start";
static void on_alarm(int sig) {
    (void)sig;
    fprintf(stderr, "\nHUNG in stage: %s\n", stage);
    _exit(3);
}

typedef struct {
    char *data;
    uint64_t *offsets64;
    uint32_t *offsets32;
    size_t count, bytes, longest, beyond_16;
    char *device_data;
    void *device_offsets;
} tape_t;

/* the words of a file: [start, end) pairs */
static char *corpus = NULL;
static size_t *word_starts = NULL, *word_lengths = NULL, words_found = 0;
static void load_corpus(char const *path) {
    FILE *file = fopen(path, "rb");
    if (!file) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(file, 0, SEEK_END);
    long const size = ftell(file);
    fseek(file, 0, SEEK_SET);
    corpus = malloc((size_t)size + 1);
    if (fread(corpus, 1, (size_t)size, file) != (size_t)size) { fprintf(stderr, "short read\n"); exit(2); }
    fclose(file);
    word_starts = malloc(((size_t)size / 2 + 1) * sizeof(size_t)), word_lengths = malloc(((size_t)size / 2 + 1) * sizeof(size_t));
    for (long i = 0; i < size;) {
        while (i < size && (corpus[i] == ' ' || corpus[i] == '\n' || corpus[i] == '\t' || corpus[i] == '\r')) ++i;
        long const start = i;
        while (i < size && !(corpus[i] == ' ' || corpus[i] == '\n' || corpus[i] == '\t' || corpus[i] == '\r')) ++i;
        if (i > start) word_starts[words_found] = (size_t)start, word_lengths[words_found] = (size_t)(i - start), ++words_found;
    }
}

static char const letters[] = "etaoinshrdlucmfwypvbgkqjxz\xC0\xC1\xC2\xC3\xFF\x80";
static int utf8_tokens = 0;
/* (PROBE_UTF8: whole sequences - e-acute, the euro sign, a Cyrillic letter - among ASCII, and now and then a lone byte of one) */
static char const *const runes[] = {"e", "t", "a", "o", "i", "n", "s", "h", "r", "\xC3\xA9", "\xE2\x82\xAC", "\xD0\xB6", "\xC3\xBC", "\x80", "\xE2"};

static tape_t make_tape(char const *source, size_t count, int wide) {
    tape_t tape;
    memset(&tape, 0, sizeof(tape));
    tape.count = count;
    tape.offsets64 = calloc(count + 1, 8), tape.offsets32 = calloc(count + 1, 4);
    size_t *picks = malloc(count * sizeof(size_t)), *lengths = malloc(count * sizeof(size_t));
    unsigned permille = 0, longest = 16;
    int const from_file = !strncmp(source, "file:", 5);
    if (!from_file && sscanf(source, "mix:%u:%u", &permille, &longest) != 2) { fprintf(stderr, "bad source %s\n", source); exit(2); }
    size_t total = 0;
    for (size_t i = 0; i < count; ++i) {
        if (from_file) picks[i] = rng() % words_found, lengths[i] = word_lengths[picks[i]];
        else if (rng() % 1000 < permille && longest > 16) lengths[i] = (rng() % 8 == 0) ? longest : 17 + rng() % (longest - 16);
        else lengths[i] = (rng() % 6 == 0) ? (rng() % 2 ? 0 : 16) : rng() % 17;
        total += lengths[i], tape.offsets64[i + 1] = total, tape.offsets32[i + 1] = (uint32_t)total;
        tape.longest = lengths[i] > tape.longest ? lengths[i] : tape.longest, tape.beyond_16 += lengths[i] > 16;
    }
    tape.bytes = total;
    tape.data = malloc(total + 1);
    for (size_t i = 0; i < count; ++i)
        for (size_t j = 0; j < lengths[i]; ++j)
            tape.data[tape.offsets64[i] + j] = from_file ? corpus[word_starts[picks[i]] + j] : letters[rng() % (sizeof(letters) - 1)];
    if (utf8_tokens && !from_file)
        for (size_t i = 0; i < count; ++i) /* whole sequences as far as the token's bytes go (what is cut short at its end is part of the test) */
            for (size_t j = 0; j < lengths[i];) {
                char const *const rune = runes[rng() % (rng() % 4 ? 9 : sizeof(runes) / sizeof(runes[0]))];
                for (size_t k = 0; rune[k] && j < lengths[i]; ++k, ++j) tape.data[tape.offsets64[i] + j] = rune[k];
            }
    hipMalloc((void **)&tape.device_data, total + 1), hipMalloc(&tape.device_offsets, (count + 1) * 8);
    hipMemcpy(tape.device_data, tape.data, total, hipMemcpyHostToDevice);
    if (wide) hipMemcpy(tape.device_offsets, tape.offsets64, (count + 1) * 8, hipMemcpyHostToDevice);
    else hipMemcpy(tape.device_offsets, tape.offsets32, (count + 1) * 4, hipMemcpyHostToDevice);
    free(picks), free(lengths);
    return tape;
}

int main(int argc, char **argv) {
    if (argc < 4) return fprintf(stderr, "usage: %s file:PATH|mix:PERMILLE:LONGEST Q C [REPEATS] [PADDING]\n", argv[0]), 2;
    char const *source = argv[1];
    size_t const q_count = strtoul(argv[2], 0, 10), c_count = getenv("PROBE_SYMMETRIC") ? q_count : strtoul(argv[3], 0, 10);
    int const repeats = argc > 4 ? atoi(argv[4]) : 4;
    size_t const padding = argc > 5 ? strtoul(argv[5], 0, 10) : 0, stride = c_count + padding;
    int const wide = getenv("PROBE_WIDE") != NULL, no_oracle = getenv("PROBE_NO_ORACLE") != NULL, symmetric = getenv("PROBE_SYMMETRIC") != NULL;
    utf8_tokens = getenv("PROBE_UTF8") != NULL;
    if (getenv("PROBE_SEED")) rng_state ^= strtoull(getenv("PROBE_SEED"), 0, 10) * 0x2545F4914F6CDD1Dull;
    signal(SIGALRM, on_alarm);
    unsigned const patience = getenv("PROBE_ALARM") ? (unsigned)atoi(getenv("PROBE_ALARM")) : 60;
    alarm(patience);
    if (!strncmp(source, "file:", 5)) load_corpus(source + 5);

    tape_t batches[2][2];
    for (int b = 0; b < 2; ++b) batches[b][0] = make_tape(source, q_count, wide), batches[b][1] = symmetric ? batches[b][0] : make_tape(source, c_count, wide);

    char const *error = NULL;
    szs_device_scope_t scope = NULL;
    sz_status_t status = szs_device_scope_init_gpu_device(0, &scope, &error);
    if (status) return fprintf(stderr, "scope: %d %s\n", status, error ? error : ""), 1;
    sz_capability_t caps;
    szs_device_scope_get_capabilities(scope, &caps, &error);
    void *engine = NULL;
    status = utf8_tokens ? szs_levenshtein_distances_utf8_init(0, 1, 1, 1, NULL, caps, &engine, &error)
                         : szs_levenshtein_distances_init(0, 1, 1, 1, NULL, caps, &engine, &error);
    if (status) return fprintf(stderr, "init: %d %s\n", status, error ? error : ""), 1;

    size_t const cells_count = q_count * stride;
    uint64_t *expected[2] = {malloc(q_count * c_count * 8), malloc(q_count * c_count * 8)}, *got = malloc(cells_count * 8), *device_results = NULL;
    hipMalloc((void **)&device_results, cells_count * 8);
    stage = "oracle";
    alarm(600);
    for (int b = 0; b < 2 && !no_oracle; ++b)
        (utf8_tokens ? szo_levenshtein_utf8_cross : szo_levenshtein_cross)(batches[b][0].data, batches[b][0].offsets64, q_count, batches[b][1].data,
                                                                             batches[b][1].offsets64, c_count, 0, 1, 1, 1, expected[b], c_count);

    int failures = 0;
    double best_kernel = 1e30, best_wall = 1e30;
    szs_rocm_call_profile_t profile;
    memset(&profile, 0, sizeof(profile));
    for (int run = 0; run < repeats; ++run) {
        int const which = run & 1;
        tape_t const *q = &batches[which][0], *c = &batches[which][1];
        stage = "engine call";
        alarm(patience);
        hipMemset(device_results, 0xEE, cells_count * 8);
        hipDeviceSynchronize();
        double const started = now_ms();
        if (wide) {
            sz_sequence_u64tape_t q_tape = {q->device_data, (sz_u64_t const *)q->device_offsets, q_count}, c_tape = {c->device_data, (sz_u64_t const *)c->device_offsets, c_count};
            status = (utf8_tokens ? szs_levenshtein_distances_utf8_u64tape : szs_levenshtein_distances_u64tape)(engine, scope, &q_tape, symmetric ? NULL : &c_tape,
                                                                                                               (sz_size_t *)device_results, stride, &error);
        }
        else {
            sz_sequence_u32tape_t q_tape = {q->device_data, (sz_u32_t const *)q->device_offsets, q_count}, c_tape = {c->device_data, (sz_u32_t const *)c->device_offsets, c_count};
            status = (utf8_tokens ? szs_levenshtein_distances_utf8_u32tape : szs_levenshtein_distances_u32tape)(engine, scope, &q_tape, symmetric ? NULL : &c_tape,
                                                                                                               (sz_size_t *)device_results, stride, &error);
        }
        double const elapsed = now_ms() - started;
        if (status) { printf("run %d: status %d %s\n", run, status, error ? error : ""); ++failures; continue; }
        szs_rocm_last_call_profile(engine, &profile);
        if (run >= 2 || repeats <= 2) {
            best_kernel = profile.kernel_milliseconds < best_kernel ? profile.kernel_milliseconds : best_kernel;
            best_wall = elapsed < best_wall ? elapsed : best_wall;
        }
        size_t bad = 0, touched_padding = 0;
        if (!no_oracle) {
            hipMemcpy(got, device_results, cells_count * 8, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < q_count; ++i) {
                for (size_t j = 0; j < c_count; ++j) {
                    uint64_t const want = expected[which][i * c_count + j], have = got[i * stride + j];
                    if (have != want && bad++ < 6)
                        printf("  [run %d q%zu(len %llu) c%zu(len %llu): got %lld want %lld]\n", run, i, (unsigned long long)(q->offsets64[i + 1] - q->offsets64[i]), j,
                               (unsigned long long)(c->offsets64[j + 1] - c->offsets64[j]), (long long)have, (long long)want);
                }
                for (size_t j = c_count; j < stride; ++j) touched_padding += got[i * stride + j] != 0xEEEEEEEEEEEEEEEEull;
            }
        }
        printf("run %d: planner %u launches %u %.1f us wall %.1f us kernel, %zu bad cells, %zu padding cells touched\n", run, profile.planner, profile.launches,
               elapsed * 1e3, profile.kernel_milliseconds * 1e3, bad, touched_padding);
        fflush(stdout);
        failures += bad != 0 || touched_padding != 0;
    }
    printf("{\"source\": \"%s\", \"queries\": %zu, \"candidates\": %zu, \"longest\": [%zu, %zu], \"beyond_16\": [%zu, %zu], \"cells\": %llu, \"planner\": %u, "
           "\"launches\": %u, \"kernel_us\": %.1f, \"wall_us\": %.1f, \"kernel_gcups\": %.1f, \"wall_gcups\": %.1f, \"results_gb_s\": %.1f, \"checked\": %s, \"failures\": %d}\n",
           source, q_count, c_count, batches[0][0].longest, batches[0][1].longest, batches[0][0].beyond_16, batches[0][1].beyond_16, (unsigned long long)profile.cells,
           profile.planner, profile.launches, best_kernel * 1e3, best_wall * 1e3, profile.cells / best_kernel * 1e-6, profile.cells / best_wall * 1e-6,
           (double)q_count * c_count * 8 / best_kernel * 1e-6, no_oracle ? "false" : "true", failures);
    return failures ? 1 : 0;
}
