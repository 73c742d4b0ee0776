/*
 *  tuning.c - the library's few tuning / testing knobs, kept OUT of the call path.
 *
 *  Round 1 read `SZS_ROCM_*` environment variables with getenv() inside every engine call: a stray variable silently
 *  changed kernel selection and getenv() raced with setenv() in threaded hosts.  Now the environment is read ONCE, when
 *  the library is loaded, into a table of plain ints; afterwards the table only changes through the additive entry
 *  point `szs_rocm_tuning_set` (include/stringzillas/stringzillas_rocm.h), which tests and probes call explicitly.
 *  Readers use relaxed atomic loads: a knob is a whole int and any interleaving of set / call is a valid setting.
 *
 *  No knob changes a RESULT - every tier, orientation and cell width computes the same scores (tests pin that); they
 *  choose which of the equivalent kernels runs.
 */
#include "szs_internal.h"

#include <stdlib.h>
#include <string.h>

static int knobs[szs_knob_count_k];

static struct {
    char const *name, *environment;
} const knob_names[szs_knob_count_k] = {
    [szs_knob_tier_k] = {"tier", "SZS_ROCM_TIER"},
    [szs_knob_swap_k] = {"swap", "SZS_ROCM_SWAP"},
    [szs_knob_packed_k] = {"packed", "SZS_ROCM_PACKED"},
    [szs_knob_rune_ids_k] = {"rune_ids", "SZS_ROCM_RUNE_IDS"},
    [szs_knob_chain_waves_k] = {"chain_waves", "SZS_ROCM_CHAIN_WAVES"},
    [szs_knob_trace_k] = {"trace", "SZS_ROCM_TRACE"},
    [szs_knob_cells_k] = {"cells", "SZS_ROCM_CELLS"},
    [szs_knob_planner_k] = {"planner", "SZS_ROCM_PLANNER"},
    [szs_knob_speculate_k] = {"speculate", "SZS_ROCM_SPECULATE"},
    [szs_knob_cpu_requests_k] = {"cpu_requests", "SZS_ROCM_CPU_REQUESTS"},
    [szs_knob_streams_k] = {"streams", "SZS_ROCM_STREAMS"},
    [szs_knob_reuse_k] = {"reuse", "SZS_ROCM_REUSE"},
    [szs_knob_split_k] = {"split", "SZS_ROCM_SPLIT"},
    [szs_knob_alphabet_k] = {"alphabet", "SZS_ROCM_ALPHABET"},
    [szs_knob_merge_k] = {"merge", "SZS_ROCM_MERGE"},
    [szs_knob_team_k] = {"team", "SZS_ROCM_TEAM"},
    [szs_knob_queues_k] = {"queues", "SZS_ROCM_QUEUES"},
    [szs_knob_roctx_k] = {"roctx", "SZS_ROCM_ROCTX"},
    [szs_knob_queue_k] = {"queue", "SZS_ROCM_QUEUE"},
    [szs_knob_queue_words_k] = {"queue_words", "SZS_ROCM_QUEUE_WORDS"},
    [szs_knob_queue_rounds_k] = {"queue_rounds", "SZS_ROCM_QUEUE_ROUNDS"},
    [szs_knob_queue_priority_k] = {"queue_priority", "SZS_ROCM_QUEUE_PRIORITY"},
    [szs_knob_fused_k] = {"fused", "SZS_ROCM_FUSED"},
    [szs_knob_tiny_k] = {"tiny", "SZS_ROCM_TINY"},
};

/** Text -> value.  -1 always means "automatic".  Tier names: lanes 0, systolic 1, chain 2; planner: host 0, device 1. */
static int parse_knob(int knob, char const *text) {
    if (!text || !text[0] || !strcmp(text, "auto")) return -1;
    if (knob == szs_knob_tier_k) {
        if (text[0] == 'l') return SZS_TIER_LANES;
        if (text[0] == 's') return SZS_TIER_SYSTOLIC;
        if (text[0] == 'c') return SZS_TIER_MYERS_CHAIN;
    }
    if (knob == szs_knob_planner_k) {
        if (text[0] == 'h') return 0;
        if (text[0] == 'd') return 1;
    }
    if (knob == szs_knob_cpu_requests_k) {
        if (text[0] == 'g') return 1; /* "gpu" */
        if (text[0] == 's') return 0; /* "strict" */
    }
    return atoi(text);
}

__attribute__((constructor)) static void szs_tuning_load(void) {
    for (int knob = 0; knob < szs_knob_count_k; ++knob)
        knobs[knob] = parse_knob(knob, getenv(knob_names[knob].environment));
    if (knobs[szs_knob_trace_k] < 0) knobs[szs_knob_trace_k] = 0;
    /*  The per-width launches of one call fan out over streams (dispatch.c: enqueue), and the HIP runtime multiplexes a process's
     *  streams onto GPU_MAX_HW_QUEUES hardware queues - 4 unless the APPLICATION says otherwise before HIP initialises.  The
     *  library only READS that variable, once, here (round 2 exported it from this constructor: a write to the environment
     *  of a process that may already have threads, and a change of queue allocation for every other HIP user in it).  The
     *  fan-out is sized to the queues the process really has: more streams than queues would share queues in an order the
     *  library does not control.  `queues` knob / SZS_ROCM_QUEUES: the queue count to assume; automatic = GPU_MAX_HW_QUEUES, or 4. */
    if (knobs[szs_knob_queues_k] < 0) {
        char const *const exported = getenv("GPU_MAX_HW_QUEUES");
        long const queues = exported ? strtol(exported, NULL, 10) : 0;
        knobs[szs_knob_queues_k] = queues > 0 ? (int)(queues < 64 ? queues : 64) : 4; /* anything unparsable or absurd: the runtime's default */
    }
}

int szs_tuning_get(int knob) { return __atomic_load_n(&knobs[knob], __ATOMIC_RELAXED); }

sz_status_t szs_rocm_tuning_set(char const *knob, char const *value) {
    if (!knob) return sz_status_unknown_k;
    for (int k = 0; k < szs_knob_count_k; ++k)
        if (!strcmp(knob, knob_names[k].name) || !strcmp(knob, knob_names[k].environment)) {
            int parsed = parse_knob(k, value);
            if (k == szs_knob_trace_k && parsed < 0) parsed = 0;
            if (k == szs_knob_queues_k && parsed <= 0) parsed = 4; /* the runtime's default */
            if (k == szs_knob_queues_k && parsed > 64) parsed = 64;
            __atomic_store_n(&knobs[k], parsed, __ATOMIC_RELAXED);
            return sz_success_k;
        }
    return sz_status_unknown_k;
}

/** The compiled instances of the team tier (hip/weighted_teams.hip), for tests and tuning scripts: 0 past the last one. */
sz_u32_t szs_rocm_team_shape(sz_size_t index) { return index < 0xFFFFFFFFu ? szs_hip_weighted_team_shape((unsigned)index) : 0; }
