/*
 *  node_probe.c - the multi-GPU C entry (`szs_rocm_node_*`, csrc/host/node.c) driven from plain C, no Python, no torch:
 *  seeded ragged batches in HOST memory (plain malloc), every cell checked against the CPU oracle (oracle/sz_oracle.c).
 *  Test infrastructure: tests/test_gpu_round2.py runs it, and it is what a C caller of the node API looks like.
 *
 *      node_probe FAMILY Q C LEN_LO LEN_HI GPU [GPU ...]       FAMILY = lev | nw | sw      (a GPU may be named twice)
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/stringzillas/stringzillas.h"
#include "../../include/stringzillas/stringzillas_rocm.h"
#include "../../oracle/sz_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rng(void) {
    rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17;
    return rng_state;
}

typedef struct {
    char *data;
    uint32_t *offsets;
    uint64_t *wide;
    size_t count;
} tape_t;

static tape_t make_tape(size_t count, size_t lo, size_t hi, char const *alphabet) {
    tape_t tape;
    tape.count = count, tape.offsets = calloc(count + 1, 4), tape.wide = calloc(count + 1, 8);
    size_t total = 0;
    for (size_t i = 0; i < count; ++i) total += lo + rng() % (hi - lo + 1), tape.offsets[i + 1] = (uint32_t)total, tape.wide[i + 1] = total;
    tape.data = malloc(total + 1);
    size_t const letters = strlen(alphabet);
    for (size_t i = 0; i < total; ++i) tape.data[i] = alphabet[rng() % letters];
    return tape;
}

int main(int argc, char **argv) {
    if (argc < 7) return fprintf(stderr, "usage: node_probe lev|nw|sw[-sym] Q C LEN_LO LEN_HI GPU [GPU ...]   (-sym: the queries against themselves)\n"), 2;
    char family[8] = {0};
    strncpy(family, argv[1], 7);
    int const symmetric = strstr(argv[1], "-sym") != NULL; /* `candidates == NULL`: the lower triangle in bands, mirrored (node.c) */
    if (strchr(family, '-')) *strchr(family, '-') = 0;
    size_t const q = strtoul(argv[2], NULL, 10), c = symmetric ? q : strtoul(argv[3], NULL, 10), lo = strtoul(argv[4], NULL, 10), hi = strtoul(argv[5], NULL, 10);
    sz_size_t gpus[SZS_ROCM_NODE_MOST_GPUS];
    size_t gpu_count = 0;
    for (int i = 6; i < argc && gpu_count < SZS_ROCM_NODE_MOST_GPUS; ++i) gpus[gpu_count++] = strtoul(argv[i], NULL, 10);

    tape_t queries = make_tape(q, lo, hi, "ACGT"), candidates = make_tape(c, lo, hi, "ACGT");
    uint8_t byte_to_class[256];
    int8_t class_costs[32 * 32];
    szo_nuc44(byte_to_class, class_costs);

    char const *error = NULL;
    szs_rocm_node_t node = NULL;
    szs_rocm_node_engine_t engine = NULL;
    sz_status_t status = szs_rocm_node_init(gpus, gpu_count, &node, &error);
    if (status != sz_success_k) return fprintf(stderr, "node_init: %d %s\n", status, error ? error : ""), 1;
    if (!strcmp(family, "lev")) status = szs_rocm_node_levenshtein_distances_init(node, 0, 1, 1, 1, &engine, &error);
    else if (!strcmp(family, "nw")) status = szs_rocm_node_needleman_wunsch_scores_init(node, byte_to_class, class_costs, -4, -1, &engine, &error);
    else status = szs_rocm_node_smith_waterman_scores_init(node, byte_to_class, class_costs, -4, -1, &engine, &error);
    if (status != sz_success_k) return fprintf(stderr, "engine_init: %d %s\n", status, error ? error : ""), 1;

    size_t const stride = c + 3; /* padded rows: the padding must stay untouched */
    int64_t *results = malloc(q * stride * sizeof(int64_t));
    for (size_t i = 0; i < q * stride; ++i) results[i] = -777;
    sz_sequence_u32tape_t const q_tape = {queries.data, queries.offsets, q}, c_tape = {candidates.data, candidates.offsets, c};
    szs_rocm_node_stats_t stats;
    for (int round = 0; round < 2; ++round) { /* the second call reuses replicas, blocks and speculates its launches */
        status = szs_rocm_node_scores_u32tape(engine, &q_tape, symmetric ? NULL : &c_tape, results, stride, &stats, &error);
        if (status != sz_success_k) return fprintf(stderr, "scores: %d %s\n", status, error ? error : ""), 1;
    }

    size_t mismatches = 0;
    for (size_t i = 0; i < q; ++i) {
        for (size_t j = 0; j < c; ++j) {
            tape_t const *other = symmetric ? &queries : &candidates;
            char const *a = queries.data + queries.offsets[i], *b = other->data + other->offsets[j];
            size_t const la = queries.offsets[i + 1] - queries.offsets[i], lb = other->offsets[j + 1] - other->offsets[j];
            int64_t const expected = !strcmp(family, "lev")  ? (int64_t)szo_levenshtein(a, la, b, lb, 0, 1, 1, 1)
                                     : !strcmp(family, "nw") ? szo_needleman_wunsch(a, la, b, lb, byte_to_class, class_costs, -4, -1)
                                                             : szo_smith_waterman(a, la, b, lb, byte_to_class, class_costs, -4, -1);
            if (results[i * stride + j] != expected && mismatches++ < 5)
                fprintf(stderr, "cell (%zu, %zu): got %lld, expected %lld\n", i, j, (long long)results[i * stride + j], (long long)expected);
        }
        for (size_t j = c; j < stride; ++j) mismatches += results[i * stride + j] != -777;
    }
    uint64_t cells = 0;
    printf("{\"family\": \"%s\", \"gpus\": %zu, \"wall_ms\": %.3f, \"mismatches\": %zu, \"per_gpu\": [", family, (size_t)stats.gpus,
           stats.wall_milliseconds, mismatches);
    for (size_t g = 0; g < stats.gpus; ++g) {
        printf("%s{\"rows\": %u, \"busy_ms\": %.3f, \"kernel_ms\": %.3f}", g ? ", " : "", stats.rows[g], stats.busy_milliseconds[g],
               stats.kernel_milliseconds[g]);
        cells += stats.cells[g];
    }
    printf("], \"cells\": %llu}\n", (unsigned long long)cells);
    szs_rocm_node_engine_free(engine);
    szs_rocm_node_free(node);
    return mismatches ? 1 : 0;
}
