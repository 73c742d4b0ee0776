/*
 *  szs_internal.h - private declarations of the C host side of libstringzillas_rocm_shared.so.
 *
 *  The host is plain C11.  It owns: status/message plumbing, device scopes, engine objects, input
 *  normalisation (tapes / callback sequences -> string refs), the planner (length-sorting candidates, grouping
 *  queries per kernel variant) and the launch sequence.  It reaches the GPU only through the HIP runtime C API
 *  and through the `extern "C"` launchers of csrc/hip/kernels.h.  There is no CPU scoring path in this library.
 */
#ifndef SZS_INTERNAL_H_
#define SZS_INTERNAL_H_

#include <stddef.h>
#include <stdint.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "../../../include/stringzillas/stringzillas.h"
#include "../../../include/stringzillas/stringzillas_rocm.h"
#include "../hip/kernels.h"

#define SZS_VERSION_MAJOR 5 /* tracks the reference ABI: /root/reference/include/stringzilla/stringzilla.h:79-81 */
#define SZS_VERSION_MINOR 1
#define SZS_VERSION_PATCH 2

/* ---- status plumbing (reference: c/stringzillas/stringzillas.cuh:207-257) ---------------------------------------- */

/** Stores the static message for `status` (or `override_message` verbatim) and returns `status`. */
sz_status_t szs_report(sz_status_t status, char const **error_message, char const *override_message);
/** Maps a HIP failure to `sz_status_unknown_k` (or `sz_bad_alloc_k` for OOM) with the HIP error *name* as message. */
sz_status_t szs_report_hip(hipError_t error, char const **error_message);

/* ---- device scopes ----------------------------------------------------------------------------------------------- */

typedef enum { szs_scope_default_k = 0, szs_scope_cpu_k = 1, szs_scope_gpu_k = 2 } szs_scope_kind_t;

typedef struct szs_scope_s {
    szs_scope_kind_t kind;
    size_t cpu_cores;
    int gpu_device;
    hipStream_t stream; /* created lazily on first use; non-blocking */
} szs_scope_s;

/** Resolves a scope handle to (device ordinal, stream) for a GPU engine; CPU scopes are a mismatch. */
sz_status_t szs_scope_bind_gpu(szs_scope_s *scope, int *device, hipStream_t *stream, char const **error_message);

/* ---- grow-only buffers ------------------------------------------------------------------------------------------- */

typedef enum { szs_memory_host_k = 0, szs_memory_pinned_k = 1, szs_memory_device_k = 2 } szs_memory_kind_t;

typedef struct szs_buffer_t {
    void *pointer;
    size_t capacity;
    szs_memory_kind_t kind;
    int device; /* for pinned / device memory */
} szs_buffer_t;

/** Ensures `buffer` holds at least `bytes`; contents are NOT preserved. */
sz_status_t szs_buffer_reserve(szs_buffer_t *buffer, szs_memory_kind_t kind, int device, size_t bytes,
                               char const **error_message);
void szs_buffer_release(szs_buffer_t *buffer);

/* ---- engines ----------------------------------------------------------------------------------------------------- */

typedef enum {
    szs_family_levenshtein_k = 0,
    szs_family_levenshtein_utf8_k = 1,
    szs_family_needleman_wunsch_k = 2,
    szs_family_smith_waterman_k = 3
} szs_family_t;

typedef struct szs_engine_s {
    uint32_t magic;
    szs_family_t family;
    /* cost model as given at init */
    int8_t match, mismatch, open, extend;
    uint8_t byte_to_class[256];
    int8_t class_costs[32 * 32];
    int is_linear;    /* open == extend (levenshtein.cuh:117) */
    int is_unit_cost; /* match 0, mismatch 1, gap 1: bit-parallel kernel (serial.hpp:118-120) */
    unsigned magnitude; /* max |cost| over substitutions and gaps, for the reach rule (serial.hpp:135-162) */

    /* grow-only scratch, bound to the device of the last call */
    int device;
    szs_buffer_t host_lengths;   /* host: addresses + lengths of both sides */
    szs_buffer_t host_scratch;   /* host: the planner's sort keys and counting bins */
    szs_buffer_t pinned_staging; /* pinned: offsets downloads, string-ref uploads */
    szs_buffer_t device_refs;    /* device: string refs of both sides */
    szs_buffer_t device_results; /* device: dense results when the caller's matrix is not device-accessible */
    szs_buffer_t device_boundary;/* device: strip boundaries of the weighted kernels */
    szs_buffer_t device_model;   /* device: szs_cost_model_t */
    szs_buffer_t device_systolic;/* device: control block of the systolic tier (epoch-tagged words, zeroed once) */
    uint32_t systolic_epoch;     /* launches that have used `device_systolic` since it was zeroed */
    szs_buffer_t device_tape;    /* device: packed copy of strings living in plain host memory (`cpu_requests = gpu` only) */
    szs_buffer_t pinned_tape;    /* pinned: the host side of that copy */
    szs_buffer_t device_runes;   /* device: UTF-32 transcription of both sides (codepoint-level engine) */
    szs_buffer_t device_narrow;  /* device: tiny tokens of a codepoint call as byte strings of rune ids (hip/utf8.hip: utf8_narrow_kernel) */
    uint64_t runes_needed;       /* runes the last device-planned codepoint call needed in that buffer (~ the bytes of its batch) */
    uint64_t cells_before;       /* cells of the previous call of this engine (the call profile is cleared when a call begins) */
    szs_buffer_t device_transcode; /* device: raw refs, rune starts, rune counts and the multibyte flag of that pass */
    szs_buffer_t device_alphabet;  /* device: the hash table that renumbers a batch's runes (hip/utf8.hip) */
    szs_buffer_t pinned_transcode; /* pinned: the host's side of the same */
    int model_uploaded_device;
    int model_uploaded_transposed; /* the uploaded class table is the transpose (sides swapped by the planner) */
    szs_cost_model_t host_model;   /* what was uploaded: lives as long as the engine, so the upload needs no wait */
    /* non-unit Levenshtein engines over byte tapes: the dense alphabet of the CURRENT call's batch (hip/weighted_teams.hip:
     * byte_presence_kernel), 0 classes when the call's inputs are not tapes the device can scan */
    uint32_t uniform_classes;
    uint8_t uniform_byte_to_class[256];
    szs_buffer_t device_presence;  /* device: the 256 presence bits */
    szs_buffer_t device_queue;     /* device: the ticket counter of hip/myers_queue.hip - zeroed when allocated, never again */
    szs_buffer_t device_queue_trace; /* device: per-workgroup begin / end ticks of that launch (`trace` knob only) */
    void *queue_zeroed;            /* the allocation of `device_queue` that was zeroed: another pointer means a fresh buffer */
    uint32_t queue_tickets;        /* the counter's value when the next launch begins (every launch says what it takes) */
    uint32_t queue_unfit_sequence; /* what the current call's queue launch writes to pinned memory if a query fits none of its tables; 0: no such launch */
    int queue_refused;             /* this call is being scored again without the queue */
    int last_queued;               /* enqueue() issued the persistent launch for the call being finished (the call profile says so) */
    szs_buffer_t device_fused;     /* device: the two `ready` words of the short launch that plans itself (kernels.h: szs_fused_plan_t) */
    void *fused_zeroed;            /* the allocation of `device_fused` that was zeroed */
    int fused_gave_up;             /* a launch that plans itself ran out of polls on this engine: it is not tried again */
    int tiny_valid;                /* the previous call of these counts was scored by the tiny-token kernel (hip/myers_tiny.hip): go straight there */
    uint32_t tiny_q_count, tiny_c_count;
    int tiny_runes_valid;          /* the same for the codepoint engine: narrowed to byte strings of rune ids, then that kernel (round 6) */
    uint32_t tiny_runes_q_count, tiny_runes_c_count;
    void *narrow_zeroed;           /* the narrow buffer whose head - the table of claimed runes, the totals - holds what the last call left */
    int tiny_refused;              /* ... or was REFUSED by it (dense in long strings): calls of these counts skip the summary-driven attempt */
    hipEvent_t event_start, event_stop;
    int events_device;
    /* launches of different bit-vector widths fan out over these and fill each other's tails (dispatch.c: enqueue) */
#ifndef SZS_AUX_STREAMS
#define SZS_AUX_STREAMS 7 /* eight streams for up to nine width groups: the short launch - thousands of workgroups that live
                             microseconds - goes LAST on the stream of the lightest long one and fills the others' tails; on a
                             stream of its own it takes the slots first (config 5: 9.90 ms on nine streams, 9.60 on eight) */
#endif
    unsigned last_streams;       /* streams the launches of the last call were dealt over (call profile) */
    hipStream_t aux_streams[SZS_AUX_STREAMS];
    hipEvent_t aux_done[SZS_AUX_STREAMS], fork_event;
    int aux_device; /* device the auxiliary streams live on, -1: none yet */

    /* device-side planning (hip/planner.hip) */
    szs_buffer_t device_plan_refs; /* device: ascending + descending refs of both sides */
    szs_buffer_t pinned_summary;   /* pinned: the planner's szs_plan_summary_t */
    szs_buffer_t pinned_squares;   /* pinned: a symmetric tiny-token call's sums of squared lengths, one per block of 256 strings */
    uint32_t plan_sequence;        /* echoed by the planner: tells this call's summary from a stale one */
    struct szs_decision_t *remembered; /* the launch shape of the previous device-planned call (speculation), or NULL */

    szs_rocm_call_profile_t last_profile;
} szs_engine_s;

#define SZS_ENGINE_MAGIC 0x535A5345u

/* ---- inputs ------------------------------------------------------------------------------------------------------ */

typedef enum { szs_input_u32tape_k = 0, szs_input_u64tape_k = 1, szs_input_sequence_k = 2 } szs_input_kind_t;

typedef struct szs_input_t {
    szs_input_kind_t kind;
    size_t count;
    char const *data;          /* tapes */
    void const *offsets;       /* tapes: count + 1 entries of 4 or 8 bytes */
    sz_sequence_t const *sequence;
} szs_input_t;

/* ---- input normalisation shared by the similarity and fingerprint calls (dispatch.c) ----------------------------- */

typedef struct {
    int host_readable;
    int device_accessible;
    int device_resident; /* hipMalloc'ed: writes from a kernel stay in HBM instead of crossing the host link */
} szs_pointer_traits_t;

szs_pointer_traits_t szs_classify_pointer(void const *pointer);
/** Enqueues the download of device-only tape offsets into `pinned_staging + staging_offset`; see dispatch.c. */
sz_status_t szs_prefetch_offsets(void *pinned_staging, hipStream_t stream, szs_input_t const *input, size_t staging_offset,
                                 void const **host_offsets, int *pending, char const **error_message);
/** Absolute addresses and 32-bit lengths of every string of one side; vets device accessibility like the reference. */
sz_status_t szs_gather_strings(szs_input_t const *input, void const *offsets, uint64_t *addresses, uint32_t *lengths,
                               uint64_t *total_bytes, int *needs_staging /* NULL: host-only strings are an error */,
                               char const **error_message);

/* ---- fingerprint engines (fingerprint_engines.c) --------------------------------------------------------------------------- */

typedef struct szs_fingerprints_s szs_fingerprints_s;
sz_status_t szs_fingerprints_create(sz_size_t dimensions, sz_size_t alphabet_size, sz_size_t const *window_widths,
                                    sz_size_t window_widths_count, sz_u64_t seed, sz_capability_t capabilities,
                                    szs_fingerprints_t *engine, char const **error_message);
sz_status_t szs_fingerprints_call(szs_fingerprints_s *engine, szs_scope_s *scope, szs_input_t const *texts,
                                  sz_u32_t *min_hashes, sz_size_t min_hashes_stride, sz_u32_t *min_counts,
                                  sz_size_t min_counts_stride, char const **error_message);
void szs_fingerprints_destroy(szs_fingerprints_s *engine);

/* ---- planner (plan.c) - pure host logic, unit-tested without a GPU through the szs_rocm_plan_* exports ------------ */

typedef struct szs_plan_group_t {
    unsigned variant;     /* Myers: rounded word count; weighted: 0 */
    uint32_t first, count;/* slice of the planned query array */
} szs_plan_group_t;

#define SZS_PLAN_MAX_GROUPS 24

typedef struct szs_plan_t {
    uint32_t longest_query, longest_candidate;
    uint64_t cells; /* sum over live pairs of len(q) * len(c): the GCUPS numerator (bench/similarities.cuh:344-366) */
    unsigned groups_count;
    szs_plan_group_t groups[SZS_PLAN_MAX_GROUPS];
    /* the lengths at SZS_PLAN_RANK_SAMPLES + 1 ascending ranks of the kernels' query side [0] and candidate side [1]
     * (hip/kernels.h: szs_plan_summary_t::rank_lengths), when `has_ranks`: what the queue of hip/myers_queue.hip is ordered by */
    int has_ranks;
    uint32_t rank_lengths[2][SZS_PLAN_RANK_SAMPLES + 1];
} szs_plan_t;

/**
 *  What to launch for one call.  A pure function of the engine's cost model and of the statistics of the two sides (their
 *  counts, longest strings, sums, band counts and strings per launch variant) - the host planner computes those from its
 *  length arrays, the device planner delivers them in its summary.  Every field stays VALID for any batch with the same
 *  counts, the same strings per launch variant on the kernels' query side and no longer longest strings, which is exactly
 *  what hip/planner.hip verifies before it lets speculated launches score anything.
 */
typedef struct szs_decision_t {
    int valid;
    int symmetric, runes;
    uint32_t alphabet;         /* runes: 0, or the size of the batch's renumbered alphabet (the UTF-32 arrays then hold ids) */
    uint32_t q_count, c_count; /* the caller's sides */
    int tier, transposed, layout;
    int use_myers, banded, maximise;
    int objective, narrow, packed, packed_local, wide_cells;
    int team_objective;        /* the team tier's objective: 0 global, 1 local, 2 distance (uniform-cost Levenshtein) */
    int team_wide;             /* the team tier's cell order: 0 half-float patterns (three-input maxima), 1 unsigned (hip/team_core.hpp) */
    unsigned team;             /* 0, or the shape of the team tier that scores the call (hip/kernels.h: lanes * 10000 + registers * 100 + waves) */
    uint32_t classes;
    uint32_t kq_count, kc_count; /* kernel roles */
    uint64_t kq_symbols;         /* symbols of the kernels' query side (with longest_query: how skewed the lengths are) */
    uint32_t longest[2];         /* the caller's sides: queries, candidates */
    szs_plan_t plan;             /* kernel roles: groups of the query side, longest strings, cells */
    size_t systolic_control_bytes, systolic_parked_bytes;
    uint32_t variant_counts[SZS_PLAN_VARIANTS]; /* of the kernels' query side */
    int use_queue;             /* the bit-parallel width groups of the call are ONE persistent launch (hip/myers_queue.hip) */
    szs_queue_plan_t queue;    /* its tiles, in the order the workgroups draw them */
    /* what the refs on the device were planned FROM (device-planned calls): the key of the guarded re-use, dispatch.c */
    int refs_current;          /* engine->device_plan_refs holds the complete plan of exactly these tapes */
    void const *key_data[2], *key_offsets[2];
    int key_wide[2];
    szs_plan_summary_t summary; /* of that plan: the call profile is filled from it */
} szs_decision_t;

/**
 *  Fills `candidate_refs` with the candidates sorted by ascending length (stable), and `query_refs` grouped by kernel
 *  variant, longest first.  `myers` = widest bit-vector (in 32-bit words) a bit-parallel kernel exists for - 0 for the
 *  weighted engines, SZS_MYERS_MAX_WORDS for bytes and codepoints; longer queries get variant 0
 *  (scored by the weighted kernel).
 *  Lengths and addresses are parallel arrays.  Scratch: `keys` holds max(q, c) uint32_t, `scratch`
 *  szs_plan_scratch_bytes(max(q, c), longest string) bytes - the planner itself never allocates.
 */
size_t szs_plan_scratch_bytes(uint32_t count, uint32_t longest);
void szs_plan_build(unsigned myers, int symmetric, uint64_t const *query_addresses, uint32_t const *query_lengths,
                    uint32_t queries_count, uint64_t const *candidate_addresses, uint32_t const *candidate_lengths,
                    uint32_t candidates_count, szs_string_ref_t *query_refs, szs_string_ref_t *candidate_refs,
                    uint32_t *keys, void *scratch, szs_plan_t *plan);
/** Statistics of one side (what the tier model reads) and, optionally, its strings per launch variant. */
void szs_side_stats(uint32_t const *lengths, uint32_t count, unsigned myers, szs_side_stats_t *stats, uint32_t *variant_counts);
/** Launch groups of queries sorted longest first, from their per-variant counts (hip/kernels.h: SZS_PLAN_VARIANTS). */
void szs_plan_groups(uint32_t const *variant_counts, szs_plan_t *plan);

/** The launch of one width group of the bit-parallel kernels (plan.c): the kernel's words (>= the group's variant) and the
 *  lanes per pair (0: one lane per pair), and the order the groups' launches leave the host in. */
typedef struct szs_launch_shape_t {
    unsigned words, lanes;
} szs_launch_shape_t;
szs_launch_shape_t szs_plan_myers_shape(int knob, unsigned variant, uint64_t workgroups_unsplit, int runes);
void szs_plan_launch_order(szs_plan_t const *plan, int use_myers, int runes, uint64_t candidate_blocks, int split_knob,
                           szs_launch_shape_t *shapes, unsigned *order);

/**
 *  The work queue of the ONE persistent launch that scores every bit-parallel width group of a unit-cost byte call
 *  (hip/myers_queue.hip): tiles of (a slice of the queries) x (a column of the candidates), each with its shape - lanes per
 *  pair, words per lane, candidates per work item - sorted by the time one of their items holds a workgroup, longest first.
 *  From the plan's groups (variant 0, the strip kernel's, is left out), its rank samples and its cells; plan.c has the model.
 *  Strings are bytes, or - for a codepoint batch the device renumbered (hip/utf8.hip) - ids of an alphabet of `alphabet` runes.
 */
void szs_plan_queue(szs_plan_t const *plan, uint32_t queries_count, uint32_t candidates_count,
                    uint32_t alphabet /* 0: bytes; A: codepoints renumbered 1 ... A */, size_t table_bytes /* LDS of a workgroup's tables */,
                    szs_queue_plan_t *queue /* items_total 0: nothing to queue, or (codepoints) an alphabet too rich for its tables */);

#define SZS_TIER_LANES 0    /* one pair per lane: lev_myers.hip, weighted.hip */
#define SZS_TIER_SYSTOLIC 1 /* one pair per chain of wavefronts: systolic.hip */
#define SZS_TIER_MYERS_CHAIN 2 /* the same chain with the bit-parallel recurrence: myers_chain.hip (unit-cost bytes) */

/**
 *  Estimated SIMD cycles of a call in its better tier (plan.c), for one orientation of the cross-product: the caller
 *  evaluates both (queries as the workgroup / band side, or candidates) and may swap the sides, which every scorer
 *  here permits - gap costs apply to both strings alike, and a swapped class table is its transpose.
 *  `bit_parallel_limit`: longest query (symbols) the Myers kernels take, 0 for weighted engines; `uniform`:
 *  Levenshtein-family costs.  The `tier` knob (host/tuning.c) forces the tier (testing aid).
 */
double szs_plan_estimate(unsigned bit_parallel_limit, int bit_parallel_chain, int affine, int uniform, int team_capable, int symmetric,
                         szs_side_stats_t const *queries, szs_side_stats_t const *candidates, unsigned band_rows, int *tier);

/**
 *  Lanes per (pair of queries, candidate) for the team tier of the 16-bit class-table scorers (hip/weighted_teams.hip): 16,
 *  4, or 0 for the one-pair-per-lane kernel - from the measured sweep profiles/r03/team_sweep_*.jsonl.  `team_capable` above:
 *  the call may take that tier at all (class table, every DP value within 16 bits).
 */
unsigned szs_plan_team_lanes(int affine, szs_side_stats_t const *queries, szs_side_stats_t const *candidates);

/**
 *  The decision itself: evaluates szs_plan_estimate for both orientations and reports the tier to run and whether the
 *  sides are swapped (never for symmetric calls).  The `swap` knob forces the orientation (testing aid).
 */
void szs_plan_orient(unsigned bit_parallel_limit, int bit_parallel_chain, int affine, int uniform, int team_capable, int symmetric,
                     szs_side_stats_t const *queries, szs_side_stats_t const *candidates, unsigned band_rows, int *tier,
                     int *transposed);

/* ---- the call (dispatch.c) --------------------------------------------------------------------------------------- */

sz_status_t szs_engine_cross(szs_engine_s *engine, szs_scope_s *scope, szs_input_t const *queries,
                             szs_input_t const *candidates /* NULL: symmetric */, void *results,
                             size_t results_row_stride, char const **error_message);

#endif /* SZS_INTERNAL_H_ */
