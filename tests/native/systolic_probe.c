/*
 *  systolic_probe.c - a torch-free parity probe: drives the C-ABI of libstringzillas_rocm_shared.so from plain C on
 *  seeded random batches and checks every cell against the CPU oracle (oracle/sz_oracle.c).  Test infrastructure: it
 *  exists so that a GPU visit can exercise a kernel tier in seconds (no Python, no 1-2 minute `import torch`).
 *
 *      systolic_probe FAMILY Q C LEN_LO LEN_HI [REPEATS] [OPEN EXTEND]      FAMILY = lev | levw | nw | sw
 *  The tier / orientation are chosen by the library; pin them with SZS_ROCM_TIER / SZS_ROCM_SWAP as usual.
 *  PROBE_ALTERNATE=1: two batches of the same shape take turns (a stream of FRESH batches: the calls that plan themselves inside
 *  their launch, or that score tiny tokens straight from the tapes - the same tapes again would take the re-use path).
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
    size_t count;
    char *device_data;
    uint32_t *device_offsets;
} tape_t;

static tape_t make_tape(size_t count, size_t lo, size_t hi, char const *alphabet) {
    tape_t tape;
    memset(&tape, 0, sizeof(tape));
    tape.count = count;
    tape.offsets64 = calloc(count + 1, 8), tape.offsets32 = calloc(count + 1, 4);
    size_t const letters = strlen(alphabet);
    size_t total = 0;
    for (size_t i = 0; i < count; ++i) total += lo + rng() % (hi - lo + 1), tape.offsets64[i + 1] = total, tape.offsets32[i + 1] = (uint32_t)total;
    tape.data = malloc(total + 1);
    for (size_t i = 0; i < total; ++i) tape.data[i] = alphabet[rng() % letters];
    hipMalloc((void **)&tape.device_data, total + 1), hipMalloc((void **)&tape.device_offsets, (count + 1) * 4);
    hipMemcpy(tape.device_data, tape.data, total, hipMemcpyHostToDevice);
    hipMemcpy(tape.device_offsets, tape.offsets32, (count + 1) * 4, hipMemcpyHostToDevice);
    return tape;
}

int main(int argc, char **argv) {
    if (argc < 6) return fprintf(stderr, "usage: %s lev|levw|nw|sw Q C LEN_LO LEN_HI [REPEATS] [OPEN EXTEND]\n", argv[0]), 2;
    char const *family = argv[1];
    size_t const q_count = strtoul(argv[2], 0, 10), c_count = strtoul(argv[3], 0, 10);
    size_t const lo = strtoul(argv[4], 0, 10), hi = strtoul(argv[5], 0, 10);
    int const repeats = argc > 6 ? atoi(argv[6]) : 3;
    int const is_lev = family[0] == 'l';
    int8_t open = argc > 8 ? (int8_t)atoi(argv[7]) : (is_lev ? (family[3] ? 3 : 1) : -4);
    int8_t extend = argc > 8 ? (int8_t)atoi(argv[8]) : (is_lev ? (family[3] ? 3 : 1) : (family[0] == 's' ? -1 : -4));
    int8_t const match = is_lev && family[3] ? 1 : 0, mismatch = is_lev && family[3] ? 3 : 1;
    signal(SIGALRM, on_alarm);
    unsigned const patience = getenv("PROBE_ALARM") ? (unsigned)atoi(getenv("PROBE_ALARM")) : 40;
    alarm(patience);

    uint8_t byte_to_class[256];
    int8_t class_costs[32 * 32];
    if (family[0] == 'n') szo_blosum62(byte_to_class, class_costs);
    else szo_nuc44(byte_to_class, class_costs);
    char const *alphabet = family[0] == 'n' ? "ARNDCQEGHILKMFPSTWYV" : "ACGT";
    int const alternate = getenv("PROBE_ALTERNATE") != NULL;
    tape_t const batches[2][2] = {{make_tape(q_count, lo, hi, alphabet), make_tape(c_count, lo, hi, alphabet)},
                                  {make_tape(alternate ? q_count : 1, lo, hi, alphabet), make_tape(alternate ? c_count : 1, lo, hi, alphabet)}};
    tape_t const queries = batches[0][0], candidates = batches[0][1];

    char const *error = NULL;
    szs_device_scope_t scope = NULL;
    sz_status_t status = szs_device_scope_init_gpu_device(0, &scope, &error);
    if (status) return fprintf(stderr, "scope: %d %s\n", status, error ? error : ""), 1;
    sz_capability_t caps;
    szs_device_scope_get_capabilities(scope, &caps, &error);
    void *engine = NULL;
    if (is_lev) status = szs_levenshtein_distances_init(match, mismatch, open, extend, NULL, caps, &engine, &error);
    else if (family[0] == 'n') status = szs_needleman_wunsch_scores_init(byte_to_class, class_costs, open, extend, NULL, caps, &engine, &error);
    else status = szs_smith_waterman_scores_init(byte_to_class, class_costs, open, extend, NULL, caps, &engine, &error);
    if (status) return fprintf(stderr, "init: %d %s\n", status, error ? error : ""), 1;

    size_t const cells_count = q_count * c_count;
    int64_t *expected = malloc(cells_count * 8), *expected_other = malloc(cells_count * 8), *got = malloc(cells_count * 8), *device_results = NULL;
    hipMalloc((void **)&device_results, cells_count * 8);
    stage = "oracle";
    int const no_oracle = getenv("PROBE_NO_ORACLE") != NULL; /* timing runs on batches the CPU checker would take minutes for */
    if (no_oracle) memset(expected, 0, cells_count * 8);
    else if (is_lev) szo_levenshtein_cross(queries.data, queries.offsets64, q_count, candidates.data, candidates.offsets64, c_count, match, mismatch, open, extend, (uint64_t *)expected, c_count);
    else if (family[0] == 'n') szo_needleman_wunsch_cross(queries.data, queries.offsets64, q_count, candidates.data, candidates.offsets64, c_count, byte_to_class, class_costs, open, extend, expected, c_count);
    else szo_smith_waterman_cross(queries.data, queries.offsets64, q_count, candidates.data, candidates.offsets64, c_count, byte_to_class, class_costs, open, extend, expected, c_count);

    if (alternate && !no_oracle) { /* the second batch's matrix */
        tape_t const *q = &batches[1][0], *c = &batches[1][1];
        if (is_lev) szo_levenshtein_cross(q->data, q->offsets64, q_count, c->data, c->offsets64, c_count, match, mismatch, open, extend, (uint64_t *)expected_other, c_count);
        else if (family[0] == 'n') szo_needleman_wunsch_cross(q->data, q->offsets64, q_count, c->data, c->offsets64, c_count, byte_to_class, class_costs, open, extend, expected_other, c_count);
        else szo_smith_waterman_cross(q->data, q->offsets64, q_count, c->data, c->offsets64, c_count, byte_to_class, class_costs, open, extend, expected_other, c_count);
    }
    int64_t *const expected_first = expected;
    int failures = 0;
    for (int run = 0; run < repeats; ++run) {
        int const which = alternate ? run & 1 : 0;
        sz_sequence_u32tape_t q_tape = {batches[which][0].device_data, batches[which][0].device_offsets, q_count};
        sz_sequence_u32tape_t c_tape = {batches[which][1].device_data, batches[which][1].device_offsets, c_count};
        expected = which ? expected_other : expected_first;
        stage = "engine call";
        alarm(patience);
        hipMemset(device_results, 0xEE, cells_count * 8);
        double const started = now_ms();
        if (is_lev) status = szs_levenshtein_distances_u32tape(engine, scope, &q_tape, &c_tape, (sz_size_t *)device_results, c_count, &error);
        else if (family[0] == 'n') status = szs_needleman_wunsch_scores_u32tape(engine, scope, &q_tape, &c_tape, (sz_ssize_t *)device_results, c_count, &error);
        else status = szs_smith_waterman_scores_u32tape(engine, scope, &q_tape, &c_tape, (sz_ssize_t *)device_results, c_count, &error);
        double const elapsed = now_ms() - started;
        if (status) { printf("run %d: status %d %s\n", run, status, error ? error : ""); ++failures; continue; }
        hipMemcpy(got, device_results, cells_count * 8, hipMemcpyDeviceToHost);
        szs_rocm_call_profile_t profile;
        szs_rocm_last_call_profile(engine, &profile);
        size_t bad = 0;
        for (size_t i = 0; i < cells_count && !no_oracle; ++i) bad += got[i] != expected[i];
        printf("%s %zux%zu len[%zu,%zu] gaps %d/%d run %d: tier %u swapped %u planner %u launches %u %.3f ms wall %.3f ms kernel, %zu bad cells", family, q_count,
               c_count, lo, hi, open, extend, run, profile.tier, profile.transposed, profile.planner, profile.launches, elapsed, profile.kernel_milliseconds, bad);
        for (size_t i = 0, shown = 0; i < cells_count && shown < 4 && !no_oracle; ++i)
            if (got[i] != expected[i])
                printf(" [q%zu(len %llu) c%zu(len %llu): got %lld want %lld]", i / c_count,
                       (unsigned long long)(batches[which][0].offsets64[i / c_count + 1] - batches[which][0].offsets64[i / c_count]), i % c_count,
                       (unsigned long long)(batches[which][1].offsets64[i % c_count + 1] - batches[which][1].offsets64[i % c_count]),
                       (long long)got[i], (long long)expected[i]), ++shown;
        printf("\n");
        fflush(stdout);
        failures += bad != 0;
    }
    return failures ? 1 : 0;
}
