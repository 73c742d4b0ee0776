/*
 *  node.c - one cross-product over the N GPUs of a host, driven from C: `szs_rocm_node_*` (stringzillas_rocm.h).
 *
 *  The reference has no multi-GPU path: one engine call = one device (stringzillas.h:137), and its C-ABI allows the
 *  obvious composition - N scopes and N engines driven by N host threads.  This file IS that
 *  composition, packaged behind additive symbols so that a C / Rust / Go caller gets the 8-GPU run BASELINE.json asks
 *  for without writing the plumbing:
 *
 *    deal       query ROWS are dealt to the GPUs by longest-processing-time on len(query) (szs_rocm_shard_rows): cells are
 *               independent, every GPU writes a disjoint set of result rows, ragged batches (config 5) stay balanced;
 *    replicate  each GPU receives both tapes' bytes once (peer-to-peer over xGMI when the source lives on another GPU the node
 *               enabled peer access to, staged through pinned memory otherwise, over the host link when it lives in host
 *               memory; skipped when the bytes are already resident on that GPU) - a few MB against seconds of scoring;
 *    score      one host thread per GPU calls the ordinary single-GPU engine (dispatch.c) on `its rows x all candidates`:
 *               its rows as a callback sequence over the replica, the candidates as a tape over the replica, results into
 *               a dense block in that GPU's HBM.  No collective, no cross-GPU traffic while scoring;
 *    place      every result row goes from the block to its place in the caller's matrix (`hipMemcpyAsync` per run of
 *               consecutive rows; the matrix may live in host, pinned, unified or any GPU's memory).
 *
 *  A SYMMETRIC call (candidates NULL) shards the LOWER TRIANGLE instead (round 4; round 3 scored the full square, twice the
 *  single-GPU engines' work): contiguous bands of rows of equal weight - row i weighs len_i x sum_{j <= i} len_j
 *  (szs_rocm_shard_triangle) - each band a rectangle (its rows x everything before it) plus the triangle of
 *  its own rows, two ordinary engine calls; the cells above the diagonal are mirrored once every band has landed (hip/mirror.hip).
 *
 *  Peer access is arranged, not assumed (round 4): szs_rocm_node_init asks `hipDeviceCanAccessPeer` for every ordered pair of the
 *  node's GPUs and enables what it can; a replica then crosses xGMI in one `hipMemcpyPeerAsync`, and a pair without peer
 *  access is staged through pinned host memory.  Both are counted in szs_rocm_node_stats_t.
 *
 *  The per-GPU engines are the same objects `szs_*_init` creates: same kernels, same planner, bit-identical scores.
 */
#include "szs_internal.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define SZS_NODE_MAGIC 0x535A4E44u
#define SZS_NODE_ENGINE_MAGIC 0x535A4E45u

void szs_engine_release(szs_engine_s *engine); /* dispatch.c */

static double node_now_milliseconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

typedef struct szs_node_s {
    uint32_t magic;
    uint32_t references; /* the handle the caller holds + one per engine of the node: an engine may outlive szs_rocm_node_free */
    uint32_t released;   /* szs_rocm_node_free was called: the handle is no longer the caller's to use */
    size_t count;
    int devices[SZS_ROCM_NODE_MOST_GPUS];
    szs_scope_s *scopes[SZS_ROCM_NODE_MOST_GPUS];
    /* peer[i][j]: GPU i of the node reads GPU j's memory directly (hipDeviceEnablePeerAccess succeeded): a replica then comes over
     * xGMI in one copy; without it the bytes are staged through pinned host memory */
    uint8_t peer[SZS_ROCM_NODE_MOST_GPUS][SZS_ROCM_NODE_MOST_GPUS];
    uint32_t peer_pairs;
} szs_node_s;

typedef struct {
    uint64_t *addresses;
    uint32_t *lengths;
} node_rows_t;

typedef struct szs_node_engine_s {
    uint32_t magic;
    szs_node_s *node;
    void *engines[SZS_ROCM_NODE_MOST_GPUS]; /* ordinary single-GPU engines, one per GPU */
    /* per-GPU replicas and blocks, grow-only */
    szs_buffer_t query_bytes[SZS_ROCM_NODE_MOST_GPUS], candidate_bytes[SZS_ROCM_NODE_MOST_GPUS], blocks[SZS_ROCM_NODE_MOST_GPUS];
    szs_buffer_t staging[SZS_ROCM_NODE_MOST_GPUS]; /* pinned: replicas from a GPU this one has no peer access to */
    /* host scratch of one call */
    szs_buffer_t offsets_copy, shard_of_row, weights, row_lists, row_addresses, row_lengths;
} szs_node_engine_s;

/* ---- nodes ---------------------------------------------------------------------------------------------------------------- */

sz_status_t szs_rocm_node_init(sz_size_t const *gpu_devices, sz_size_t count, szs_rocm_node_t *out, char const **error_message) {
    if (!out) return szs_report(sz_status_unknown_k, error_message, "Node must not be null");
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        (void)hipGetLastError();
        return szs_report(sz_missing_gpu_k, error_message, NULL);
    }
    if (!count || !gpu_devices) count = (sz_size_t)visible, gpu_devices = NULL;
    if (count > SZS_ROCM_NODE_MOST_GPUS) return szs_report(sz_unexpected_dimensions_k, error_message, "Too many GPUs for one node");
    szs_node_s *node = (szs_node_s *)calloc(1, sizeof(szs_node_s));
    if (!node) return szs_report(sz_bad_alloc_k, error_message, NULL);
    node->magic = SZS_NODE_MAGIC, node->count = count, node->references = 1;
    for (size_t i = 0; i < count; ++i) {
        size_t const ordinal = gpu_devices ? gpu_devices[i] : i;
        szs_device_scope_t scope = NULL;
        sz_status_t const status = szs_device_scope_init_gpu_device(ordinal, &scope, error_message);
        if (status != sz_success_k) {
            szs_rocm_node_free(node);
            return status;
        }
        node->devices[i] = (int)ordinal, node->scopes[i] = (szs_scope_s *)scope; /* the same GPU may appear twice (testing) */
    }
    /* Peer access, once per ordered pair of distinct GPUs: the replicas of a call then travel GPU to GPU over xGMI.  A pair the
     * runtime refuses is not an error - its copies are staged through pinned host memory (node_replicate) - it is REPORTED
     * (szs_rocm_node_stats_t.peer_pairs, peer_copies, staged_copies). */
    int previous = 0;
    (void)hipGetDevice(&previous);
    for (size_t i = 0; i < count; ++i)
        for (size_t j = 0; j < count; ++j) {
            if (node->devices[i] == node->devices[j]) continue;
            int reachable = 0;
            if (hipDeviceCanAccessPeer(&reachable, node->devices[i], node->devices[j]) != hipSuccess || !reachable) {
                (void)hipGetLastError();
                continue;
            }
            hipError_t error = hipSetDevice(node->devices[i]);
            if (error == hipSuccess) error = hipDeviceEnablePeerAccess(node->devices[j], 0);
            if (error == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError(), error = hipSuccess;
            if (error == hipSuccess) node->peer[i][j] = 1, node->peer_pairs++;
            else (void)hipGetLastError();
        }
    (void)hipSetDevice(previous);
    *out = node;
    return szs_report(sz_success_k, error_message, NULL);
}

sz_size_t szs_rocm_node_size(szs_rocm_node_t handle) {
    szs_node_s const *node = (szs_node_s const *)handle;
    return node && node->magic == SZS_NODE_MAGIC ? node->count : 0;
}

/** Drops one reference; the scopes and the node itself go with the last one. */
static void node_unreference(szs_node_s *node) {
    if (__atomic_sub_fetch(&node->references, 1u, __ATOMIC_ACQ_REL)) return;
    for (size_t i = 0; i < node->count; ++i)
        if (node->scopes[i]) szs_device_scope_free(node->scopes[i]);
    node->magic = 0;
    free(node);
}

void szs_rocm_node_free(szs_rocm_node_t handle) {
    szs_node_s *node = (szs_node_s *)handle;
    if (!node || node->magic != SZS_NODE_MAGIC) return;
    /* A second free is recognised (and ignored) only while engines created from the node still keep it alive; once the last
     * reference is gone the memory has been returned and the handle must not be used again - as with every other handle here. */
    if (__atomic_exchange_n(&node->released, 1u, __ATOMIC_ACQ_REL)) return;
    node_unreference(node); /* engines created from the node keep it alive until they are freed themselves */
}

/* ---- engines of a node ---------------------------------------------------------------------------------------------------- */

static sz_status_t node_engine_new(szs_rocm_node_t handle, szs_node_engine_s **created, szs_rocm_node_engine_t *out, char const **error_message) {
    szs_node_s *node = (szs_node_s *)handle;
    if (!node || node->magic != SZS_NODE_MAGIC || __atomic_load_n(&node->released, __ATOMIC_ACQUIRE))
        return szs_report(sz_status_unknown_k, error_message, "Node must be initialized");
    if (!out) return szs_report(sz_status_unknown_k, error_message, "Engine must not be null");
    if (*out) return szs_report(sz_status_unknown_k, error_message, "Engine must be uninitialized");
    szs_node_engine_s *engine = (szs_node_engine_s *)calloc(1, sizeof(szs_node_engine_s));
    if (!engine) return szs_report(sz_bad_alloc_k, error_message, NULL);
    engine->magic = SZS_NODE_ENGINE_MAGIC, engine->node = node;
    __atomic_add_fetch(&node->references, 1u, __ATOMIC_ACQ_REL);
    *created = engine, *out = engine;
    return sz_success_k;
}

void szs_rocm_node_engine_free(szs_rocm_node_engine_t handle) {
    szs_node_engine_s *engine = (szs_node_engine_s *)handle;
    if (!engine || engine->magic != SZS_NODE_ENGINE_MAGIC) return;
    int previous = 0;
    (void)hipGetDevice(&previous);
    for (size_t i = 0; i < engine->node->count; ++i) {
        (void)hipSetDevice(engine->node->devices[i]);
        szs_buffer_release(&engine->query_bytes[i]);
        szs_buffer_release(&engine->candidate_bytes[i]);
        szs_buffer_release(&engine->blocks[i]);
        szs_buffer_release(&engine->staging[i]);
        if (engine->engines[i]) szs_levenshtein_distances_free(engine->engines[i]); /* every family frees the same object */
    }
    (void)hipSetDevice(previous);
    szs_buffer_release(&engine->offsets_copy), szs_buffer_release(&engine->shard_of_row), szs_buffer_release(&engine->weights);
    szs_buffer_release(&engine->row_lists), szs_buffer_release(&engine->row_addresses), szs_buffer_release(&engine->row_lengths);
    node_unreference(engine->node);
    engine->magic = 0;
    free(engine);
}

#define SZS_NODE_INIT_BODY(CALL)                                                                                       \
    szs_node_engine_s *engine = NULL;                                                                                  \
    sz_status_t status = node_engine_new(node, &engine, out, error_message);                                           \
    if (status != sz_success_k) return status;                                                                         \
    for (size_t i = 0; i < engine->node->count && status == sz_success_k; ++i) status = CALL;                          \
    if (status != sz_success_k) szs_rocm_node_engine_free(engine), *out = NULL;                                        \
    return status;

sz_status_t szs_rocm_node_levenshtein_distances_init(szs_rocm_node_t node, sz_error_cost_t match, sz_error_cost_t mismatch,
                                                     sz_error_cost_t open, sz_error_cost_t extend, szs_rocm_node_engine_t *out,
                                                     char const **error_message) {
    SZS_NODE_INIT_BODY(szs_levenshtein_distances_init(match, mismatch, open, extend, NULL, sz_cap_cuda_k, &engine->engines[i], error_message))
}
sz_status_t szs_rocm_node_levenshtein_distances_utf8_init(szs_rocm_node_t node, sz_error_cost_t match, sz_error_cost_t mismatch,
                                                          sz_error_cost_t open, sz_error_cost_t extend, szs_rocm_node_engine_t *out,
                                                          char const **error_message) {
    SZS_NODE_INIT_BODY(szs_levenshtein_distances_utf8_init(match, mismatch, open, extend, NULL, sz_cap_cuda_k, &engine->engines[i], error_message))
}
sz_status_t szs_rocm_node_needleman_wunsch_scores_init(szs_rocm_node_t node, sz_u8_t const *byte_to_class,
                                                       sz_error_cost_t const *class_substitution_costs, sz_error_cost_t open,
                                                       sz_error_cost_t extend, szs_rocm_node_engine_t *out, char const **error_message) {
    SZS_NODE_INIT_BODY(szs_needleman_wunsch_scores_init(byte_to_class, class_substitution_costs, open, extend, NULL, sz_cap_cuda_k,
                                                        &engine->engines[i], error_message))
}
sz_status_t szs_rocm_node_smith_waterman_scores_init(szs_rocm_node_t node, sz_u8_t const *byte_to_class,
                                                     sz_error_cost_t const *class_substitution_costs, sz_error_cost_t open,
                                                     sz_error_cost_t extend, szs_rocm_node_engine_t *out, char const **error_message) {
    SZS_NODE_INIT_BODY(szs_smith_waterman_scores_init(byte_to_class, class_substitution_costs, open, extend, NULL, sz_cap_cuda_k,
                                                      &engine->engines[i], error_message))
}

/* ---- one call --------------------------------------------------------------------------------------------------------------- */

static sz_cptr_t node_row_start(void const *handle, sz_sorted_idx_t index) {
    return (sz_cptr_t)(uintptr_t)((node_rows_t const *)handle)->addresses[index];
}
static sz_size_t node_row_length(void const *handle, sz_sorted_idx_t index) { return ((node_rows_t const *)handle)->lengths[index]; }

/** Host-readable view of a tape's offsets: the caller's pointer, or a download into `landing`. */
static sz_status_t node_offsets(void const *offsets, size_t bytes, void *landing, void const **readable, char const **error_message) {
    if (!offsets) return szs_report(sz_status_unknown_k, error_message, "Tape offsets must not be null");
    if (szs_classify_pointer(offsets).host_readable) return *readable = offsets, sz_success_k;
    hipError_t const error = hipMemcpy(landing, offsets, bytes, hipMemcpyDeviceToHost);
    if (error != hipSuccess) return szs_report_hip(error, error_message);
    *readable = landing;
    return sz_success_k;
}

static uint64_t offset_at(void const *offsets, int wide, size_t index) {
    return wide ? ((uint64_t const *)offsets)[index] : (uint64_t)((uint32_t const *)offsets)[index];
}

typedef struct {
    szs_node_engine_s *engine;
    size_t shard;
    /* inputs */
    char const *query_data, *candidate_data;
    uint64_t query_first, query_bytes, candidate_first, candidate_bytes; /* byte ranges of the tapes that hold strings */
    void const *query_offsets, *candidate_offsets; /* host-readable; the candidates' are rebased to start at zero */
    int wide;
    size_t queries_count, candidates_count;
    uint32_t const *rows; /* global query rows of this shard, ascending */
    size_t rows_count;
    int symmetric;        /* the shard is the band of rows [band_first, band_first + rows_count) of the lower triangle */
    size_t band_first;
    uint32_t peer_copies, staged_copies;
    uint64_t *row_addresses;
    uint32_t *row_lengths;
    void *results;
    size_t results_row_stride;
    /* outputs */
    sz_status_t status;
    char const *message;
    double busy_milliseconds, kernel_milliseconds;
    uint64_t cells;
} node_task_t;

/** Bytes [first, first + bytes) of a tape on the GPU of this task: in place when they already live there, else replicated
 *  once per call.  `*base` addresses byte `first` of the tape - offsets are REBASED by the caller to start at zero, so that
 *  every address handed to the engine lies inside a real allocation. */
static sz_status_t node_replicate(szs_node_engine_s *engine, size_t shard, szs_buffer_t *replica, int device, hipStream_t stream,
                                  char const *data, uint64_t first, uint64_t bytes, char const **base, uint32_t *peer_copies,
                                  uint32_t *staged_copies, char const **error_message) {
    *base = data + first;
    if (!bytes) return sz_success_k;
    szs_node_s const *node = engine->node;
    hipPointerAttribute_t attributes;
    memset(&attributes, 0, sizeof(attributes));
    int source_device = -1; /* the GPU the bytes live on, -1: host memory */
    if (hipPointerGetAttributes(&attributes, data + first) == hipSuccess) {
        if ((attributes.type == hipMemoryTypeDevice && attributes.device == device) || attributes.type == hipMemoryTypeManaged)
            return sz_success_k; /* resident on this GPU, or migrating on demand: nothing to copy */
        if (attributes.type == hipMemoryTypeDevice) source_device = attributes.device;
    }
    else (void)hipGetLastError();
    sz_status_t status = szs_buffer_reserve(replica, szs_memory_device_k, device, bytes, error_message);
    if (status != sz_success_k) return status;
    /* From another GPU: straight over xGMI when szs_rocm_node_init enabled peer access for the pair; else through pinned host
     * memory, explicitly - what the runtime would do behind `hipMemcpyDefault` anyway, but counted. */
    int direct = source_device < 0;
    for (size_t j = 0; j < node->count && !direct; ++j) direct = node->devices[j] == source_device && node->peer[shard][j];
    hipError_t error;
    if (direct) {
        error = source_device < 0 ? hipMemcpyAsync(replica->pointer, data + first, bytes, hipMemcpyDefault, stream)
                                  : hipMemcpyPeerAsync(replica->pointer, device, data + first, source_device, bytes, stream);
        if (source_device < 0) ++*staged_copies; /* from host memory: over the host link, by definition */
        else ++*peer_copies;
    }
    else {
        status = szs_buffer_reserve(&engine->staging[shard], szs_memory_pinned_k, device, bytes, error_message);
        if (status != sz_success_k) return status;
        error = hipMemcpy(engine->staging[shard].pointer, data + first, bytes, hipMemcpyDeviceToHost);
        if (error == hipSuccess) error = hipMemcpyAsync(replica->pointer, engine->staging[shard].pointer, bytes, hipMemcpyHostToDevice, stream);
        ++*staged_copies;
    }
    if (error != hipSuccess) return szs_report_hip(error, error_message);
    *base = (char const *)replica->pointer;
    return sz_success_k;
}

/**
 *  One band of a SYMMETRIC call: rows [a, b) of the lower triangle = (rows [a, b) x strings [0, a)), a rectangle, + the triangle
 *  of the rows themselves - two ordinary calls of this GPU's engine on slices of ONE tape (the rebased offsets of the call),
 *  results into one block [b - a][b] of this GPU's HBM, then one 2-D copy to the caller's rows.  The cells above the diagonal
 *  that belong to other bands are mirrored after every band has landed (node_cross).  Scores what the single-GPU engines score:
 *  the lower triangle, once (serial.hpp:3169-3182).
 */
static void node_band(node_task_t *task, int device, hipStream_t stream, char const *base) {
    szs_node_engine_s *engine = task->engine;
    size_t const shard = task->shard, first = task->band_first, rows = task->rows_count, end = first + rows;
    task->status = szs_buffer_reserve(&engine->blocks[shard], szs_memory_device_k, device, rows * end * sizeof(uint64_t), &task->message);
    if (task->status != sz_success_k) {
        (void)hipStreamSynchronize(stream);
        return;
    }
    size_t const offset_size = task->wide ? 8 : 4;
    szs_input_kind_t const kind = task->wide ? szs_input_u64tape_k : szs_input_u32tape_k;
    /* `query_offsets` were rebased to the replica by node_cross: slices of them are tapes over `base` */
    szs_input_t const band = {kind, rows, base, (char const *)task->query_offsets + first * offset_size, NULL};
    szs_input_t const before = {kind, first, base, task->query_offsets, NULL};
    szs_engine_s *const single = (szs_engine_s *)engine->engines[shard];
    uint64_t *const block = (uint64_t *)engine->blocks[shard].pointer;
    if (first) { /* the rectangle: columns [0, first) of the block */
        task->status = szs_engine_cross(single, engine->node->scopes[shard], &band, &before, block, end, &task->message);
        if (task->status != sz_success_k) return;
        task->kernel_milliseconds += single->last_profile.kernel_milliseconds, task->cells += single->last_profile.cells;
    }
    /* the triangle: columns [first, end) of the block, both halves of it (the engine mirrors inside its own square) */
    task->status = szs_engine_cross(single, engine->node->scopes[shard], &band, NULL, block + first, end, &task->message);
    if (task->status != sz_success_k) return;
    task->kernel_milliseconds += single->last_profile.kernel_milliseconds, task->cells += single->last_profile.cells;
    hipError_t error = hipMemcpy2DAsync((uint64_t *)task->results + first * task->results_row_stride, task->results_row_stride * sizeof(uint64_t),
                                        block, end * sizeof(uint64_t), end * sizeof(uint64_t), rows, hipMemcpyDefault, stream);
    hipError_t const drained = hipStreamSynchronize(stream);
    if (error == hipSuccess) error = drained;
    if (error != hipSuccess) task->status = szs_report_hip(error, &task->message);
}

static void *node_worker(void *argument) {
    node_task_t *task = (node_task_t *)argument;
    szs_node_engine_s *engine = task->engine;
    szs_node_s *node = engine->node;
    size_t const shard = task->shard;
    double const started = node_now_milliseconds();
    task->status = sz_success_k, task->message = NULL;
    if (!task->rows_count) return NULL;

    int device = 0;
    hipStream_t stream = NULL;
    task->status = szs_scope_bind_gpu(node->scopes[shard], &device, &stream, &task->message); /* makes the GPU current on this thread */
    if (task->status != sz_success_k) return NULL;

    char const *query_base = NULL, *candidate_base = NULL;
    task->status = node_replicate(engine, shard, &engine->query_bytes[shard], device, stream, task->query_data, task->query_first,
                                  task->query_bytes, &query_base, &task->peer_copies, &task->staged_copies, &task->message);
    if (task->symmetric) { /* one tape plays both roles: see node_band */
        if (task->status == sz_success_k) node_band(task, device, stream, query_base);
        else (void)hipStreamSynchronize(stream);
        task->busy_milliseconds = node_now_milliseconds() - started;
        return NULL;
    }
    if (task->status == sz_success_k)
        task->status = node_replicate(engine, shard, &engine->candidate_bytes[shard], device, stream, task->candidate_data, task->candidate_first,
                                      task->candidate_bytes, &candidate_base, &task->peer_copies, &task->staged_copies, &task->message);
    if (task->status == sz_success_k)
        task->status = szs_buffer_reserve(&engine->blocks[shard], szs_memory_device_k, device,
                                          task->rows_count * task->candidates_count * sizeof(uint64_t), &task->message);
    if (task->status != sz_success_k) {
        (void)hipStreamSynchronize(stream);
        return NULL;
    }

    /* this shard's rows as a callback sequence over the replica; all candidates as a tape over the replica */
    for (size_t r = 0; r < task->rows_count; ++r) {
        uint64_t const from = offset_at(task->query_offsets, task->wide, task->rows[r]);
        task->row_addresses[r] = (uint64_t)(uintptr_t)(query_base + (from - task->query_first));
        task->row_lengths[r] = (uint32_t)(offset_at(task->query_offsets, task->wide, (size_t)task->rows[r] + 1) - from);
    }
    node_rows_t const view = {task->row_addresses, task->row_lengths};
    sz_sequence_t const sequence = {&view, task->rows_count, node_row_start, node_row_length};
    szs_input_t const query_input = {szs_input_sequence_k, task->rows_count, NULL, NULL, &sequence};
    szs_input_t const candidate_input = {task->wide ? szs_input_u64tape_k : szs_input_u32tape_k, task->candidates_count, candidate_base,
                                         task->candidate_offsets, NULL};
    szs_engine_s *const single = (szs_engine_s *)engine->engines[shard];
    task->status = szs_engine_cross(single, node->scopes[shard], &query_input, &candidate_input, engine->blocks[shard].pointer,
                                    task->candidates_count, &task->message);
    if (task->status != sz_success_k) return NULL;
    task->kernel_milliseconds = single->last_profile.kernel_milliseconds, task->cells = single->last_profile.cells;

    /* rows to their places: one 2-D copy per run of consecutive global rows */
    uint64_t const *block = (uint64_t const *)engine->blocks[shard].pointer;
    size_t const row_bytes = task->candidates_count * sizeof(uint64_t);
    hipError_t error = hipSuccess;
    for (size_t r = 0; r < task->rows_count && error == hipSuccess;) {
        size_t run = 1;
        while (r + run < task->rows_count && task->rows[r + run] == task->rows[r] + run) ++run;
        error = hipMemcpy2DAsync((uint64_t *)task->results + (size_t)task->rows[r] * task->results_row_stride,
                                 task->results_row_stride * sizeof(uint64_t), block + r * task->candidates_count, row_bytes, row_bytes, run,
                                 hipMemcpyDefault, stream);
        r += run;
    }
    hipError_t const drained = hipStreamSynchronize(stream);
    if (error == hipSuccess) error = drained;
    if (error != hipSuccess) task->status = szs_report_hip(error, &task->message);
    task->busy_milliseconds = node_now_milliseconds() - started;
    return NULL;
}

/** Fills the cells above the diagonal from the ones below: in place on the GPU that owns the matrix when that is one of the node's
 *  (or the matrix is unified / pinned memory any GPU reaches); staged through the node's first GPU when the matrix lives on a GPU
 *  the node does not drive (peer access is only arranged among the node's own GPUs: a kernel there would fault); on the host, tile
 *  by tile, when no GPU reaches it. */
static sz_status_t node_mirror(szs_node_s *node, void *results, size_t side, size_t stride, char const **error_message) {
    szs_pointer_traits_t const traits = szs_classify_pointer(results);
    if (!traits.device_accessible) {
        uint64_t *const matrix = (uint64_t *)results;
        enum { tile = 32 }; /* a strided column walk over the whole triangle missed the cache on every cell */
        for (size_t i0 = 0; i0 < side; i0 += tile)
            for (size_t j0 = 0; j0 <= i0; j0 += tile)
                for (size_t i = i0; i < side && i < i0 + tile; ++i)
                    for (size_t j = j0; j < i && j < j0 + tile; ++j) matrix[j * stride + i] = matrix[i * stride + j];
        return sz_success_k;
    }
    size_t shard = 0; /* the GPU that owns the matrix, when it is one of the node's; else the first */
    int foreign = 0;  /* device memory of a GPU that is not one of the node's */
    hipPointerAttribute_t attributes;
    memset(&attributes, 0, sizeof(attributes));
    if (hipPointerGetAttributes(&attributes, results) == hipSuccess && attributes.type == hipMemoryTypeDevice) {
        foreign = 1;
        for (size_t s = 0; s < node->count; ++s)
            if (node->devices[s] == attributes.device) { shard = s, foreign = 0; break; }
    }
    else (void)hipGetLastError();
    int device = 0;
    hipStream_t stream = NULL;
    sz_status_t const status = szs_scope_bind_gpu(node->scopes[shard], &device, &stream, error_message);
    if (status != sz_success_k) return status;
    hipError_t error = hipSuccess;
    if (foreign) { /* copy in (the runtime routes device-to-device copies itself), mirror here, copy back */
        uint64_t *staged = NULL;
        size_t const row_bytes = side * sizeof(uint64_t);
        error = hipMalloc((void **)&staged, side * row_bytes);
        if (error == hipSuccess) error = hipMemcpy2DAsync(staged, row_bytes, results, stride * sizeof(uint64_t), row_bytes, side, hipMemcpyDefault, stream);
        if (error == hipSuccess) error = (hipError_t)szs_hip_mirror_lower(staged, (uint32_t)side, side, stream);
        if (error == hipSuccess) error = hipMemcpy2DAsync(results, stride * sizeof(uint64_t), staged, row_bytes, row_bytes, side, hipMemcpyDefault, stream);
        hipError_t const drained = hipStreamSynchronize(stream);
        if (error == hipSuccess) error = drained;
        if (staged) (void)hipFree(staged);
        return error == hipSuccess ? sz_success_k : szs_report_hip(error, error_message);
    }
    error = (hipError_t)szs_hip_mirror_lower((uint64_t *)results, (uint32_t)side, stride, stream);
    hipError_t const drained = hipStreamSynchronize(stream);
    if (error == hipSuccess) error = drained;
    return error == hipSuccess ? sz_success_k : szs_report_hip(error, error_message);
}

static sz_status_t node_cross(szs_node_engine_s *engine, int wide, int symmetric, char const *query_data, void const *query_offsets_raw,
                              size_t queries_count, char const *candidate_data, void const *candidate_offsets_raw, size_t candidates_count,
                              void *results, size_t results_row_stride, szs_rocm_node_stats_t *stats, char const **error_message) {
    if (!engine || engine->magic != SZS_NODE_ENGINE_MAGIC) return szs_report(sz_status_unknown_k, error_message, "Engine must be initialized");
    double const started = node_now_milliseconds();
    szs_node_s *node = engine->node;
    if (stats) memset(stats, 0, sizeof(*stats)), stats->gpus = node->count, stats->peer_pairs = node->peer_pairs, stats->symmetric = symmetric != 0;
    if (!queries_count || !candidates_count) return szs_report(sz_success_k, error_message, NULL);
    if (queries_count > 0xFFFFFFFFull || candidates_count > 0xFFFFFFFFull) return szs_report(sz_overflow_risk_k, error_message, NULL);
    if (!results) return szs_report(sz_status_unknown_k, error_message, "Results must not be null");
    if (results_row_stride < candidates_count) return szs_report(sz_unexpected_dimensions_k, error_message, NULL);

    /* ---- offsets where the host can read them */
    size_t const offset_size = wide ? 8 : 4;
    size_t const q_offsets_bytes = (queries_count + 1) * offset_size, c_offsets_bytes = (candidates_count + 1) * offset_size;
    sz_status_t status = szs_buffer_reserve(&engine->offsets_copy, szs_memory_host_k, 0, q_offsets_bytes + c_offsets_bytes + 16, error_message);
    if (status != sz_success_k) return status;
    void const *query_offsets = NULL, *candidate_offsets = NULL;
    status = node_offsets(query_offsets_raw, q_offsets_bytes, engine->offsets_copy.pointer, &query_offsets, error_message);
    if (status != sz_success_k) return status;
    void *const candidate_landing = (char *)engine->offsets_copy.pointer + ((q_offsets_bytes + 7) & ~(size_t)7);
    status = node_offsets(candidate_offsets_raw, c_offsets_bytes, candidate_landing, &candidate_offsets, error_message);
    if (status != sz_success_k) return status;
    for (size_t i = 0; i < candidates_count; ++i) /* before anything is derived from them: a descending tape is a caller's error, not an allocation failure */
        if (offset_at(candidate_offsets, wide, i + 1) < offset_at(candidate_offsets, wide, i) ||
            offset_at(candidate_offsets, wide, i + 1) - offset_at(candidate_offsets, wide, i) > 0xFFFFFFFFull)
            return szs_report(sz_unexpected_dimensions_k, error_message, "Tape offsets must ascend");
    /* the candidates' offsets, rebased to start at zero, in OUR buffer (the caller's array is never written) */
    uint64_t const candidate_first = offset_at(candidate_offsets, wide, 0);
    uint64_t const candidate_bytes = offset_at(candidate_offsets, wide, candidates_count) - candidate_first;
    if (candidate_offsets != candidate_landing) memmove(candidate_landing, candidate_offsets, c_offsets_bytes);
    for (size_t i = 0; i <= candidates_count; ++i) {
        if (wide) ((uint64_t *)candidate_landing)[i] -= candidate_first;
        else ((uint32_t *)candidate_landing)[i] -= (uint32_t)candidate_first;
    }
    candidate_offsets = candidate_landing;
    for (size_t i = 0; i < queries_count; ++i)
        if (offset_at(query_offsets, wide, i + 1) < offset_at(query_offsets, wide, i) ||
            offset_at(query_offsets, wide, i + 1) - offset_at(query_offsets, wide, i) > 0xFFFFFFFFull)
            return szs_report(sz_unexpected_dimensions_k, error_message, "Tape offsets must ascend");

    size_t const shards = node->count;
    size_t band_first[SZS_ROCM_NODE_MOST_GPUS + 1] = {0};
    sz_u64_t band_weights[SZS_ROCM_NODE_MOST_GPUS] = {0};
    if (symmetric) {
        /* ---- one tape in both roles: the LOWER TRIANGLE in contiguous bands of rows of equal weight (szs_rocm_shard_triangle),
         * scored once - the full square would be twice the work of the single-GPU engines (serial.hpp:3169-3182).  The band
         * calls read slices of the query offsets, rebased to the replica: in OUR buffer, the caller's array is never written. */
        uint64_t const first = offset_at(query_offsets, wide, 0);
        void *const landing = engine->offsets_copy.pointer;
        if (query_offsets != landing) memmove(landing, query_offsets, q_offsets_bytes);
        for (size_t i = 0; i <= queries_count; ++i) {
            if (wide) ((uint64_t *)landing)[i] -= first;
            else ((uint32_t *)landing)[i] -= (uint32_t)first;
        }
        sz_status_t dealt = szs_buffer_reserve(&engine->weights, szs_memory_host_k, 0, queries_count * sizeof(sz_size_t), error_message);
        if (dealt != sz_success_k) return dealt;
        sz_size_t *const lengths = (sz_size_t *)engine->weights.pointer;
        for (size_t i = 0; i < queries_count; ++i) lengths[i] = (sz_size_t)(offset_at(landing, wide, i + 1) - offset_at(landing, wide, i));
        dealt = szs_rocm_shard_triangle(lengths, queries_count, shards, band_first, band_weights);
        if (dealt != sz_success_k) return szs_report(dealt, error_message, NULL);
        node_task_t tasks[SZS_ROCM_NODE_MOST_GPUS];
        pthread_t threads[SZS_ROCM_NODE_MOST_GPUS];
        int started_threads[SZS_ROCM_NODE_MOST_GPUS] = {0};
        memset(tasks, 0, sizeof(tasks));
        for (size_t s = 0; s < shards; ++s) {
            node_task_t *task = &tasks[s];
            task->engine = engine, task->shard = s, task->wide = wide, task->symmetric = 1;
            task->query_data = query_data, task->query_offsets = landing, task->query_first = first;
            task->query_bytes = offset_at(landing, wide, queries_count);
            task->queries_count = task->candidates_count = queries_count;
            task->band_first = band_first[s], task->rows_count = band_first[s + 1] - band_first[s];
            task->results = results, task->results_row_stride = results_row_stride;
            if (shards == 1) node_worker(task);
            else if (pthread_create(&threads[s], NULL, node_worker, task) == 0) started_threads[s] = 1;
            else node_worker(task);
        }
        for (size_t s = 0; s < shards; ++s)
            if (started_threads[s]) pthread_join(threads[s], NULL);
        status = sz_success_k;
        for (size_t s = 0; s < shards; ++s) {
            if (tasks[s].status != sz_success_k && status == sz_success_k) {
                status = tasks[s].status;
                if (error_message) *error_message = tasks[s].message;
            }
            if (stats) {
                stats->busy_milliseconds[s] = tasks[s].busy_milliseconds, stats->kernel_milliseconds[s] = tasks[s].kernel_milliseconds;
                stats->cells[s] = tasks[s].cells, stats->rows[s] = (sz_u32_t)tasks[s].rows_count, stats->row_weights[s] = band_weights[s];
                stats->peer_copies[s] = tasks[s].peer_copies, stats->staged_copies[s] = tasks[s].staged_copies;
            }
        }
        /* every band has landed: the cells above the diagonal that belong to other bands' rows */
        if (status == sz_success_k && shards > 1) status = node_mirror(node, results, queries_count, results_row_stride, error_message);
        if (stats) stats->wall_milliseconds = node_now_milliseconds() - started;
        if (status == sz_success_k && error_message) *error_message = NULL;
        return status;
    }

    /* ---- deal the rows: LPT on len(query) + 1 */
    status = szs_buffer_reserve(&engine->weights, szs_memory_host_k, 0, queries_count * sizeof(sz_size_t), error_message);
    if (status == sz_success_k) status = szs_buffer_reserve(&engine->shard_of_row, szs_memory_host_k, 0, queries_count * sizeof(uint32_t), error_message);
    if (status == sz_success_k) status = szs_buffer_reserve(&engine->row_lists, szs_memory_host_k, 0, queries_count * sizeof(uint32_t), error_message);
    if (status == sz_success_k) status = szs_buffer_reserve(&engine->row_addresses, szs_memory_host_k, 0, queries_count * sizeof(uint64_t), error_message);
    if (status == sz_success_k) status = szs_buffer_reserve(&engine->row_lengths, szs_memory_host_k, 0, queries_count * sizeof(uint32_t), error_message);
    if (status != sz_success_k) return status;
    sz_size_t *weights = (sz_size_t *)engine->weights.pointer;
    uint32_t *shard_of_row = (uint32_t *)engine->shard_of_row.pointer, *row_lists = (uint32_t *)engine->row_lists.pointer;
    for (size_t i = 0; i < queries_count; ++i) weights[i] = (sz_size_t)(offset_at(query_offsets, wide, i + 1) - offset_at(query_offsets, wide, i)) + 1;
    sz_u64_t loads[SZS_ROCM_NODE_MOST_GPUS];
    status = szs_rocm_shard_rows(weights, queries_count, shards, shard_of_row, loads);
    if (status != sz_success_k) return szs_report(status, error_message, NULL);
    size_t counts[SZS_ROCM_NODE_MOST_GPUS] = {0}, firsts[SZS_ROCM_NODE_MOST_GPUS] = {0};
    for (size_t i = 0; i < queries_count; ++i) counts[shard_of_row[i]]++;
    for (size_t s = 1; s < shards; ++s) firsts[s] = firsts[s - 1] + counts[s - 1];
    size_t cursors[SZS_ROCM_NODE_MOST_GPUS];
    memcpy(cursors, firsts, sizeof(cursors));
    for (size_t i = 0; i < queries_count; ++i) row_lists[cursors[shard_of_row[i]]++] = (uint32_t)i; /* ascending within a shard */

    /* ---- one host thread per GPU */
    node_task_t tasks[SZS_ROCM_NODE_MOST_GPUS];
    pthread_t threads[SZS_ROCM_NODE_MOST_GPUS];
    int started_threads[SZS_ROCM_NODE_MOST_GPUS] = {0};
    memset(tasks, 0, sizeof(tasks));
    for (size_t s = 0; s < shards; ++s) {
        node_task_t *task = &tasks[s];
        task->engine = engine, task->shard = s, task->wide = wide;
        task->query_data = query_data, task->candidate_data = candidate_data;
        task->query_offsets = query_offsets, task->candidate_offsets = candidate_offsets;
        task->query_first = offset_at(query_offsets, wide, 0);
        task->query_bytes = offset_at(query_offsets, wide, queries_count) - task->query_first;
        task->candidate_first = candidate_first, task->candidate_bytes = candidate_bytes;
        task->queries_count = queries_count, task->candidates_count = candidates_count;
        task->rows = row_lists + firsts[s], task->rows_count = counts[s];
        task->row_addresses = (uint64_t *)engine->row_addresses.pointer + firsts[s];
        task->row_lengths = (uint32_t *)engine->row_lengths.pointer + firsts[s];
        task->results = results, task->results_row_stride = results_row_stride;
        if (shards == 1) node_worker(task);
        else if (pthread_create(&threads[s], NULL, node_worker, task) == 0) started_threads[s] = 1;
        else node_worker(task); /* no thread to be had: this shard runs on the calling thread, after the others were started */
    }
    for (size_t s = 0; s < shards; ++s)
        if (started_threads[s]) pthread_join(threads[s], NULL);

    status = sz_success_k;
    for (size_t s = 0; s < shards; ++s) {
        if (tasks[s].status != sz_success_k && status == sz_success_k) {
            status = tasks[s].status;
            if (error_message) *error_message = tasks[s].message;
        }
        if (stats) {
            stats->busy_milliseconds[s] = tasks[s].busy_milliseconds, stats->kernel_milliseconds[s] = tasks[s].kernel_milliseconds;
            stats->cells[s] = tasks[s].cells, stats->rows[s] = (sz_u32_t)counts[s], stats->row_weights[s] = loads[s];
            stats->peer_copies[s] = tasks[s].peer_copies, stats->staged_copies[s] = tasks[s].staged_copies;
        }
    }
    if (stats) stats->wall_milliseconds = node_now_milliseconds() - started;
    if (status == sz_success_k && error_message) *error_message = NULL;
    return status;
}

sz_status_t szs_rocm_node_scores_u32tape(szs_rocm_node_engine_t engine, sz_sequence_u32tape_t const *queries,
                                         sz_sequence_u32tape_t const *candidates, void *results, sz_size_t results_row_stride,
                                         szs_rocm_node_stats_t *stats, char const **error_message) {
    if (!queries) return szs_report(sz_status_unknown_k, error_message, "Queries must not be null");
    int const symmetric = candidates == NULL; /* self-similarity: the lower triangle in bands of rows, mirrored */
    if (symmetric) candidates = queries;
    return node_cross((szs_node_engine_s *)engine, 0, symmetric, queries->data, queries->offsets, queries->count, candidates->data,
                      candidates->offsets, candidates->count, results, results_row_stride, stats, error_message);
}

sz_status_t szs_rocm_node_scores_u64tape(szs_rocm_node_engine_t engine, sz_sequence_u64tape_t const *queries,
                                         sz_sequence_u64tape_t const *candidates, void *results, sz_size_t results_row_stride,
                                         szs_rocm_node_stats_t *stats, char const **error_message) {
    if (!queries) return szs_report(sz_status_unknown_k, error_message, "Queries must not be null");
    int const symmetric = candidates == NULL;
    if (symmetric) candidates = queries;
    return node_cross((szs_node_engine_s *)engine, 1, symmetric, queries->data, queries->offsets, queries->count, candidates->data,
                      candidates->offsets, candidates->count, results, results_row_stride, stats, error_message);
}
