/*
 *  plan.c - the host planner: pure C, no HIP, unit-tested on CPU through szs_rocm_plan_probe / szs_rocm_shard_rows.
 *
 *  The reference routes work per *cell*: it materialises a 112-byte task per (query, candidate) pair on the device,
 *  counting-sorts the tasks by size tier and scatters results afterwards (cuda.cuh:1652-1711,1887-1957,2082,2146) -
 *  112 MB of bookkeeping traffic per million pairs.  A cross-product does not need that: tiering the ROWS and the
 *  COLUMNS tiers every cell.  So the plan is O(Q + C):
 *    - candidates are sorted by length, so each 64-lane wavefront (one candidate per lane) walks near-equal texts;
 *    - queries are grouped by kernel variant (for the bit-parallel kernel: the 32-bit word count of their bit-vector,
 *      rounded up to an instantiated kernel), one launch per group;
 *    - nothing per cell is ever stored: kernels write results[query.index * stride + candidate.index] directly.
 */
#include "szs_internal.h"

#include <stdlib.h>
#include <string.h>

static int compare_u64(void const *a, void const *b) {
    uint64_t const x = *(uint64_t const *)a, y = *(uint64_t const *)b;
    return x < y ? -1 : x > y;
}

size_t szs_plan_scratch_bytes(uint32_t count, uint32_t longest) {
    /* valid for BOTH sides of a call, whichever holds the longest string: a side whose own longest is shorter may still
     * take the counting sort */
    size_t const counting = ((size_t)(longest < (1u << 16) ? longest : (1u << 16) - 1) + 2) * sizeof(uint32_t);
    size_t const keyed = (size_t)count * sizeof(uint64_t);
    return (counting > keyed ? counting : keyed) + 16;
}

/** Stable ascending sort of indices [0, count) by `lengths`; `order` receives the permutation.  No allocation: `scratch`
 *  holds szs_plan_scratch_bytes(count, longest) bytes (the engine's grow-only host buffer). */
static void sort_by_length(uint32_t const *lengths, uint32_t count, uint32_t longest, uint32_t *order, void *scratch) {
    if (count == 0) return;
    if (longest < (1u << 16)) { /* counting sort: O(count + longest), the common case */
        uint32_t *bins = (uint32_t *)scratch;
        memset(bins, 0, ((size_t)longest + 2) * sizeof(uint32_t));
        for (uint32_t i = 0; i < count; ++i) bins[lengths[i] + 1]++;
        for (uint32_t l = 0; l <= longest; ++l) bins[l + 1] += bins[l];
        for (uint32_t i = 0; i < count; ++i) order[bins[lengths[i]]++] = i;
        return;
    }
    uint64_t *keys = (uint64_t *)scratch;
    for (uint32_t i = 0; i < count; ++i) keys[i] = ((uint64_t)lengths[i] << 32) | i; /* index breaks ties: stable */
    qsort(keys, count, sizeof(uint64_t), compare_u64);
    for (uint32_t i = 0; i < count; ++i) order[i] = (uint32_t)keys[i];
}

static unsigned const variant_steps[SZS_PLAN_VARIANTS] = {0, SZS_MYERS_SHORT_WORDS, 10, 12, 16, 20, 24, 32, 48, 64};

/** Slot of a string in `variant_counts` (hip/kernels.h): 0 = no bit-parallel width, 1 = the mixed short launch, 2..9 = the
 *  long widths in ascending order. */
static unsigned variant_slot(uint32_t length, unsigned widest) {
    unsigned const words = length ? (length + 31) / 32 : 1;
    if (!widest || words > widest) return 0;
    for (unsigned slot = 1; slot < SZS_PLAN_VARIANTS; ++slot)
        if (words <= variant_steps[slot]) return slot;
    return 0;
}

void szs_side_stats(uint32_t const *lengths, uint32_t count, unsigned myers, szs_side_stats_t *stats, uint32_t *variant_counts) {
    memset(stats, 0, sizeof(*stats));
    if (variant_counts) memset(variant_counts, 0, SZS_PLAN_VARIANTS * sizeof(uint32_t));
    stats->count = count;
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t const length = lengths[i];
        if (length > stats->longest) stats->longest = length;
        stats->symbols += length;
        stats->bands_systolic += length ? (length + SZS_SYSTOLIC_BAND_ROWS - 1) / SZS_SYSTOLIC_BAND_ROWS : 1;
        stats->bands_chain += length ? (length + SZS_MYERS_CHAIN_BAND_ROWS - 1) / SZS_MYERS_CHAIN_BAND_ROWS : 1;
        if (variant_counts) variant_counts[variant_slot(length, myers)]++;
    }
}

void szs_plan_groups(uint32_t const *variant_counts, szs_plan_t *plan) {
    /* Queries arrive LONGEST FIRST: the strings no bit-parallel width takes (slot 0), then the long widths from the
     * widest down, last the one mixed-width launch of all short queries. */
    plan->groups_count = 0;
    uint32_t first = 0;
    for (unsigned step = 0; step < SZS_PLAN_VARIANTS; ++step) {
        unsigned const slot = step == 0 ? 0 : SZS_PLAN_VARIANTS - step;
        if (!variant_counts[slot]) continue;
        szs_plan_group_t *group = &plan->groups[plan->groups_count++];
        group->variant = variant_steps[slot], group->first = first, group->count = variant_counts[slot];
        first += variant_counts[slot];
    }
}

void szs_plan_build(unsigned myers, int symmetric, uint64_t const *query_addresses, uint32_t const *query_lengths,
                    uint32_t queries_count, uint64_t const *candidate_addresses, uint32_t const *candidate_lengths,
                    uint32_t candidates_count, szs_string_ref_t *query_refs, szs_string_ref_t *candidate_refs,
                    uint32_t *keys, void *scratch, szs_plan_t *plan) {
    memset(plan, 0, sizeof(*plan));
    szs_side_stats_t query_stats, candidate_stats;
    uint32_t variant_counts[SZS_PLAN_VARIANTS];
    szs_side_stats(query_lengths, queries_count, myers, &query_stats, variant_counts);
    szs_side_stats(candidate_lengths, candidates_count, 0, &candidate_stats, NULL);
    plan->longest_query = query_stats.longest, plan->longest_candidate = candidate_stats.longest;
    if (symmetric) { /* lower triangle incl. diagonal: sum_i len_i * sum_{j <= i} len_j */
        uint64_t prefix = 0, cells = 0;
        for (uint32_t i = 0; i < queries_count; ++i) prefix += query_lengths[i], cells += (uint64_t)query_lengths[i] * prefix;
        plan->cells = cells;
    }
    else { plan->cells = query_stats.symbols * candidate_stats.symbols; }

    /* Candidates: ascending length. */
    sort_by_length(candidate_lengths, candidates_count, plan->longest_candidate, keys, scratch);
    for (uint32_t slot = 0; slot < candidates_count; ++slot) {
        uint32_t const c = keys[slot];
        candidate_refs[slot].address = candidate_addresses[c];
        candidate_refs[slot].length = candidate_lengths[c];
        candidate_refs[slot].index = c;
    }

    /* Queries: LONGEST FIRST.  The launch variant is monotone in the length, so descending order makes every variant a
     * contiguous slice - first the queries too long for the bit-parallel kernels (variant 0, weighted kernel), then the
     * long widths from the widest down, last the one mixed-width launch of all short queries - and within a launch the
     * heaviest workgroups are handed out first, so a launch drains on its lightest work.  The weighted engines use
     * the same order for their single group (variant 0): their kernels pull work items heaviest first. */
    sort_by_length(query_lengths, queries_count, plan->longest_query, keys, scratch);
    for (uint32_t slot = 0; slot < queries_count; ++slot) {
        uint32_t const q = keys[queries_count - 1 - slot]; /* ascending order read backwards */
        query_refs[slot].address = query_addresses[q];
        query_refs[slot].length = query_lengths[q];
        query_refs[slot].index = q;
    }
    szs_plan_groups(variant_counts, plan);
    /* the lengths at 33 ranks of each side, as the device planner reports them (hip/planner.hip): the queue of the one-launch
     * bit-parallel kernel is ordered by them (szs_plan_queue) */
    plan->has_ranks = 1;
    for (unsigned k = 0; k <= SZS_PLAN_RANK_SAMPLES; ++k) {
        plan->rank_lengths[0][k] = queries_count ? query_refs[queries_count - 1 - (uint32_t)((uint64_t)k * (queries_count - 1) / SZS_PLAN_RANK_SAMPLES)].length : 0;
        plan->rank_lengths[1][k] = candidates_count ? candidate_refs[(uint32_t)((uint64_t)k * (candidates_count - 1) / SZS_PLAN_RANK_SAMPLES)].length : 0;
    }
}

/* ---- the work queue of the one-launch bit-parallel kernel (hip/myers_queue.hip) ------------------------------------------- */

/*
 *  The model.  A workgroup of the persistent kernel is eight wavefronts behind one query's match masks; one of its work items
 *  is a few queries against S candidates.  A lane that holds `w` words of a pattern takes ~`SZS_QUEUE_WORD_COLUMN_NS` per text
 *  column and word on a busy device (10.5 instructions, four wavefronts taking turns on a SIMD: profiles/r04), so
 *      one wave block ~ w x (longest candidate of its column) x that,
 *  and nothing shortens it but more lanes per pair.  The call as a whole takes ~cells / SZS_QUEUE_CELLS_PER_NS.  A wave block
 *  must stay under 0.6 of that even when it starts first, or the call ends waiting for it: that bounds the words per lane,
 *  column by column (16, 12, 8, 4: a whole-device batch keeps 16 everywhere; one GPU's eighth of a Zipf batch goes down to 4 -
 *  sixteen lanes for a 2048-byte query - against its longest candidates only).  An
 *  item is several rounds while it stays under a sixteenth of the call (the masks are built once per item, and eight
 *  wavefronts that draw several rounds of (candidate block, query) pairs balance each other: more is better while the queue's
 *  end stays fine-grained) - first as many QUERIES as the 64-word table of a workgroup takes side by side, then candidates.  Tiles are sorted by rounds x words per lane x longest candidate: longest-processing-time-first list
 *  scheduling of the workgroup slots.
 */
#define SZS_QUEUE_WORD_COLUMN_NS 90.0
#define SZS_QUEUE_CELLS_PER_NS 9.0e4 /* 90 TCUPS */
#define SZS_QUEUE_MOST_SLICES 12u
#define SZS_QUEUE_MOST_COLUMNS 8u
#define SZS_QUEUE_MOST_ROUNDS 8u
#define SZS_QUEUE_LANE_OVERHEAD 14.0 /* instructions per column a lane of a team spends on the hand-over, beside 10.5 per word */

typedef struct {
    uint32_t first, count, bound; /* queries [first, first + count) of the descending array, none longer than `bound` symbols */
    unsigned words_per_lane, lanes;
} queue_slice_t;

static unsigned words_of(uint32_t length) { return length ? (length + 31u) / 32u : 1u; }

/** What bounds a tile's tables (hip/myers_queue.hip): bytes have 256 rows and a 64 KB table per workgroup; codepoints of a
 *  renumbered batch have alphabet + 1 rows and 72 KB, and tables that do not fit as rows fit as pointers + a pool. */
typedef struct {
    int runes;
    uint32_t rows;         /* symbols + 1 */
    uint32_t arena_dwords; /* the workgroup's LDS for tables */
} queue_tables_t;

static uint32_t queue_pointer_bytes(unsigned words_per_lane) { return words_per_lane <= 4 ? 2u : words_per_lane <= 8 ? 4u : 8u; }

/** Dwords of LDS the table of a query of `length` symbols takes in a shape, as the kernel computes them: 0 = not at all. */
static uint32_t queue_table_dwords(queue_tables_t const *tables, uint32_t length, unsigned words_per_lane, unsigned lanes, int *sparse) {
    unsigned const needed = words_of(length);
    *sparse = 0;
    if (!tables->runes) { /* rows of 1, 2 or 4 words: [chunk][256][4] */
        unsigned const body = lanes > 1 ? words_per_lane * lanes : needed <= 8 ? needed : needed <= 10 ? 10u : needed <= 12 ? 12u : needed <= 16 ? 16u : 20u;
        uint64_t const dwords = body >= 3 ? 256ull * ((body + 3u) & ~3u) : 256ull * body;
        return dwords <= tables->arena_dwords ? (uint32_t)dwords : 0u; /* (a 12 x 6 = 72-word team does not fit the 64 KB arena: not a shape) */
    }
    unsigned const body = lanes > 1 ? words_per_lane * lanes : needed <= 8 ? needed : needed <= 12 ? 12u : 16u;
    uint64_t const direct = (uint64_t)tables->rows * ((body + 3u) & ~3u);
    if (direct <= tables->arena_dwords) return (uint32_t)direct;
    if (lanes < 2) return 0; /* one lane per pair reads rows */
    uint64_t const pointers = ((uint64_t)tables->rows * lanes * queue_pointer_bytes(words_per_lane) + 15u) / 16u * 4u;
    uint64_t const pooled = pointers + ((uint64_t)length + 1u) * 4u;
    *sparse = 1;
    return pooled <= tables->arena_dwords ? (uint32_t)pooled : 0u;
}

/**
 *  Lanes per pair and words per lane for patterns of up to `bound` symbols when a lane may hold `most` (4 / 8 / 12 / 16) words
 *  of them, one lane at most `most_alone`, a lane of a team at most `team_most`.  Returns 0 when no shape's table fits.
 */
static int queue_shape(queue_tables_t const *tables, uint32_t bound, unsigned most_alone, unsigned team_most, unsigned *words_per_lane,
                       unsigned *lanes, int *sparse) {
    unsigned const words = words_of(bound);
    if (words <= most_alone && queue_table_dwords(tables, bound, 0, 1, sparse)) { /* one lane per pair, at each query's own width */
        *words_per_lane = 0, *lanes = 1;
        return 1;
    }
    double best = -1;
    for (unsigned attempt = 0; attempt < 2 && best < 0; ++attempt) /* second round: any width, when the budget's shapes have no table */
        for (unsigned w = 4; w <= (attempt ? 16u : team_most); w += 4) {
            unsigned const l = (words + w - 1) / w;
            if (l > 16u) continue;
            unsigned const team = l < 2u ? 2u : l;
            int as_pool = 0;
            if (!queue_table_dwords(tables, bound, w, team, &as_pool)) continue;
            double const width_used = (double)words / (double)(w * team);
            double const lanes_used = (double)(team * (64u / team)) / 64.0;
            double const issue_used = 10.5 * w / (10.5 * w + SZS_QUEUE_LANE_OVERHEAD + (as_pool ? 2.0 * (w / 4) : 0.0));
            double const efficiency = width_used * lanes_used * issue_used;
            if (efficiency > best) best = efficiency, *words_per_lane = w, *lanes = team, *sparse = as_pool;
        }
    return best >= 0;
}

void szs_plan_queue(szs_plan_t const *plan, uint32_t queries_count, uint32_t candidates_count, uint32_t alphabet, size_t table_bytes,
                    szs_queue_plan_t *queue) {
    memset(queue, 0, sizeof(*queue));
    if (!queries_count || !candidates_count) return;
    queue_tables_t const tables = {alphabet != 0, alphabet ? alphabet + 1u : 256u, (uint32_t)(table_bytes / 4)};
    uint32_t const longest_candidate = plan->longest_candidate ? plan->longest_candidate : 1u;
    double const call_ns = (double)plan->cells / SZS_QUEUE_CELLS_PER_NS;
    /* what one wave block may take of the call: the words one lane holds x the longest candidate of the tile's column, at the
     * per-word-column time of a busy device, must stay under it (the `queue_words` knob pins the words instead) */
    double const round_ns = call_ns * 0.6;
    int const words_knob = szs_tuning_get(szs_knob_queue_words_k);

    /* ---- slices of the queries: every width group of the plan, cut further where the sampled lengths fall by a quarter */
    queue_slice_t slices[SZS_QUEUE_MOST_SLICES];
    unsigned slices_count = 0, groups_left = 0;
    for (unsigned g = 0; g < plan->groups_count; ++g) groups_left += plan->groups[g].variant != 0;
    for (unsigned g = 0; g < plan->groups_count && slices_count < SZS_QUEUE_MOST_SLICES; ++g) {
        szs_plan_group_t const *group = &plan->groups[g];
        if (!group->variant || !group->count) continue;
        --groups_left;
        uint32_t bound = group->variant * 32u;
        if (bound > plan->longest_query) bound = plan->longest_query;
        queue_slice_t *slice = &slices[slices_count++];
        slice->first = group->first, slice->count = group->count, slice->bound = bound;
        if (!plan->has_ranks) continue;
        /* samples in descending order of length = ascending position in the descending query array */
        for (int k = (int)SZS_PLAN_RANK_SAMPLES; k >= 0 && slices_count + groups_left < SZS_QUEUE_MOST_SLICES; --k) {
            uint32_t const position = queries_count - 1 - (uint32_t)((uint64_t)k * (queries_count - 1) / SZS_PLAN_RANK_SAMPLES);
            if (position <= slice->first || position >= group->first + group->count) continue;
            uint32_t const sampled = plan->rank_lengths[0][k];
            if (words_of(sampled) * 4u > words_of(slice->bound) * 3u) continue; /* not yet a quarter narrower */
            queue_slice_t *next = &slices[slices_count++];
            next->first = position, next->count = slice->first + slice->count - position, next->bound = sampled;
            slice->count = position - slice->first;
            slice = next;
        }
    }
    if (!slices_count) return;

    /* ---- columns of the candidates: cut where the sampled lengths have fallen to 0.6 of the column's longest, so that the
     * items of a tile are alike (equal counts would put 700 ... 1900-byte texts of a Zipf batch into one column) */
    unsigned columns_most = SZS_QUEUE_MOST_TILES / slices_count;
    if (columns_most > SZS_QUEUE_MOST_COLUMNS) columns_most = SZS_QUEUE_MOST_COLUMNS;
    uint32_t column_begin[SZS_QUEUE_MOST_COLUMNS], column_end[SZS_QUEUE_MOST_COLUMNS], column_longest[SZS_QUEUE_MOST_COLUMNS];
    unsigned columns = 0;
    {
        uint32_t end = candidates_count, top = longest_candidate;
        if (plan->has_ranks && candidates_count > 1)
            for (int k = (int)SZS_PLAN_RANK_SAMPLES - 1; k >= 0 && columns + 1 < columns_most; --k) {
                uint32_t const rank = (uint32_t)((uint64_t)k * (candidates_count - 1) / SZS_PLAN_RANK_SAMPLES), sampled = plan->rank_lengths[1][k];
                if (rank + 1 >= end || (uint64_t)sampled * 10u > (uint64_t)top * 6u) continue;
                /* the cut goes DOWN to a whole number of wavefronts (64 candidates) below the end of the array: wave blocks are
                 * counted from a column's end, and a cut anywhere else leaves every query a half-empty wave block per column -
                 * a twelfth of config 5's wave blocks, and of its instructions (profiles/r04) */
                uint32_t const cut = candidates_count - (candidates_count - (rank + 1) + 63u) / 64u * 64u < candidates_count
                                         ? candidates_count - (candidates_count - (rank + 1) + 63u) / 64u * 64u : 0u;
                if (!cut || cut >= end) continue;
                column_begin[columns] = cut, column_end[columns] = end, column_longest[columns] = top ? top : 1u, ++columns;
                end = cut, top = sampled; /* every candidate of a rank up to `rank` (so: below `cut`) is no longer than the sample there */
            }
        column_begin[columns] = 0, column_end[columns] = end, column_longest[columns] = top ? top : 1u, ++columns;
    }

    /* ---- tiles, their shapes and their keys.  An item is G queries x S candidates = G x (S / pairs per wave block) wave blocks
     * for the workgroup's eight wavefronts: as many queries as the 64-word table takes side by side, then as many candidate
     * blocks as keep the item under its share of the call. */
    int const rounds_knob = szs_tuning_get(szs_knob_queue_rounds_k);
    double const item_ns = call_ns / 16.0; /* what one item may hold a workgroup for: the queue's granularity at its end */
    double keys[SZS_QUEUE_MOST_TILES];
    unsigned tiles = 0;
    uint64_t chain_most = 0; /* words one lane holds x the longest candidate it meets: the launch's longest chain of dependent steps */
    for (unsigned column = 0; column < columns; ++column) {
        uint32_t const begin = column_begin[column], end = column_end[column], longest = column_longest[column];
        /* the most words one lane may hold against THIS column: the largest of 16 / 12 / 8 / 4 whose wave block fits.  The shape
         * follows the column - an eighth of a Zipf batch spreads a 2048-byte query over sixteen lanes against its 2000-byte
         * candidates only, and keeps four lanes of sixteen words (a third fewer instructions) against the short ones */
        double const fitting = round_ns / (SZS_QUEUE_WORD_COLUMN_NS * longest);
        unsigned most = fitting >= 16 ? 16u : fitting >= 12 ? 12u : fitting >= 8 ? 8u : 4u;
        /* teams stop at twelve words per lane: the sixteen-word team body fills its 128 registers and config 5 runs 1.7 % slower with
         * it (9.69 against 9.52 ms, profiles/r04); ONE lane still takes up to twenty words when the budget allows sixteen */
        unsigned team_most = most < 12u ? most : 12u;
        if (words_knob == 4 || words_knob == 8 || words_knob == 12 || words_knob == 16) most = team_most = (unsigned)words_knob;
        for (unsigned i = 0; i < slices_count; ++i) {
            queue_slice_t *slice = &slices[i];
            int sparse = 0;
            /* one lane per pair: up to 20 words of bytes (16 of codepoints) when the budget allows sixteen, else what it allows */
            if (!queue_shape(&tables, slice->bound, most == 16u ? (tables.runes ? 16u : 20u) : most, team_most, &slice->words_per_lane, &slice->lanes,
                             &sparse)) {
                memset(queue, 0, sizeof(*queue)); /* codepoints whose alphabet leaves this slice no table: not a call for this kernel */
                return;
            }
            szs_queue_tile_t *tile = &queue->tiles[tiles];
            unsigned const bound_words = words_of(slice->bound);
            unsigned const lane_words = slice->lanes > 1 ? slice->words_per_lane : bound_words;
            /* as many queries side by side as have room for their symbols (64 words of pattern between them, the kernel's
             * indexing) and for their tables (an equal share of the workgroup's LDS each) */
            uint32_t const table_dwords = queue_table_dwords(&tables, slice->bound, slice->words_per_lane, slice->lanes, &sparse);
            unsigned const pattern_words = slice->lanes > 1 ? slice->words_per_lane * slice->lanes : bound_words;
            unsigned side_by_side = 1;
            for (unsigned g = 16; g > 1; --g)
                if (((64u / g) & ~3u) >= pattern_words && ((tables.arena_dwords / g) & ~3u) >= table_dwords) { side_by_side = g; break; }
            uint32_t const pairs_per_wave = 64u / slice->lanes;
            double const one_round_ns = SZS_QUEUE_WORD_COLUMN_NS * lane_words * longest; /* eight wave blocks, one per wavefront */
            unsigned rounds = one_round_ns >= item_ns ? 1u : (unsigned)(item_ns / one_round_ns);
            if (rounds > SZS_QUEUE_MOST_ROUNDS) rounds = SZS_QUEUE_MOST_ROUNDS;
            if (rounds_knob > 0) rounds = (unsigned)rounds_knob;
            unsigned const wave_blocks = 8u * rounds;
            if (side_by_side > wave_blocks) side_by_side = wave_blocks;
            if (side_by_side > slice->count) side_by_side = slice->count;
            uint64_t per_item = (uint64_t)(wave_blocks / side_by_side ? wave_blocks / side_by_side : 1u) * pairs_per_wave;
            if (per_item > end - begin) per_item = end - begin;
            tile->query_first = slice->first, tile->query_count = slice->count;
            tile->candidate_first = begin, tile->candidate_end = end;
            tile->candidates_per_item = (uint32_t)per_item;
            tile->words_per_lane = (uint8_t)slice->words_per_lane, tile->lanes = (uint8_t)slice->lanes;
            tile->queries_per_item = (uint8_t)side_by_side, tile->flags = sparse ? SZS_QUEUE_TILE_SPARSE : 0;
            uint64_t const blocks_of_item = side_by_side * ((per_item + pairs_per_wave - 1) / pairs_per_wave);
            keys[tiles] = (double)((blocks_of_item + 7u) / 8u) * lane_words * longest;
            if ((uint64_t)lane_words * longest > chain_most) chain_most = (uint64_t)lane_words * longest;
            ++tiles;
        }
    }
#define SZS_QUEUE_ITEMS_OF(TILE)                                                                                                          \
    ((uint64_t)(((TILE)->query_count + (TILE)->queries_per_item - 1) / (TILE)->queries_per_item) *                                        \
     (((TILE)->candidate_end - (TILE)->candidate_first + (TILE)->candidates_per_item - 1) / (TILE)->candidates_per_item))
    /* a queue of more than 2^31 items: coarser items (the kernel addresses them with 32 bits) */
    for (;;) {
        uint64_t items = 0;
        for (unsigned t = 0; t < tiles; ++t) items += SZS_QUEUE_ITEMS_OF(&queue->tiles[t]);
        if (items < (1ull << 31)) break;
        unsigned coarsened = 0;
        for (unsigned t = 0; t < tiles; ++t) {
            szs_queue_tile_t *tile = &queue->tiles[t];
            uint32_t const span = tile->candidate_end - tile->candidate_first;
            if (tile->candidates_per_item >= span) continue; /* an item already spans the column */
            tile->candidates_per_item = tile->candidates_per_item * 2u < span ? tile->candidates_per_item * 2u : span;
            keys[t] *= 2, ++coarsened;
        }
        if (!coarsened) { /* 2^31 items even with whole columns per item (2^31 groups of queries): no queue - the call takes the
                             per-width launches, which cut enormous cross-products along the query axis themselves */
            memset(queue, 0, sizeof(*queue));
            return;
        }
    }
    /* ---- longest first (stable: equal keys keep widest-slice-first, lightest column first), then the tiles' places in the queue */
    for (unsigned t = 1; t < tiles; ++t) {
        szs_queue_tile_t const moved = queue->tiles[t];
        double const key = keys[t];
        unsigned at = t;
        for (; at > 0 && keys[at - 1] < key; --at) queue->tiles[at] = queue->tiles[at - 1], keys[at] = keys[at - 1];
        queue->tiles[at] = moved, keys[at] = key;
    }
    uint32_t items = 0;
    for (unsigned t = 0; t < tiles; ++t) {
        szs_queue_tile_t *tile = &queue->tiles[t];
        tile->first_item = items;
        items += (uint32_t)SZS_QUEUE_ITEMS_OF(tile);
    }
#undef SZS_QUEUE_ITEMS_OF
    queue->tiles_count = tiles, queue->items_total = items;
    /* Longest chain first on the SIMD (hip/myers_queue.hip): -11 % on an eighth of config 5, -9 % on a half, -0.7 % on the whole; on
     * codepoints -4 % on an eighth, 0 on a half and +2 % on the whole batch (5.40 -> 5.51 ms, profiles/r04/queue_priority_5u.txt):
     * automatic for byte calls and for codepoint calls of up to ~3 ms; the `queue_priority` knob pins it. */
    int const priority_knob = szs_tuning_get(szs_knob_queue_priority_k);
    int const prioritised = priority_knob >= 0 ? priority_knob != 0 : !(tables.runes && call_ns > 3.0e6);
    queue->chain_most = !prioritised ? 0u : chain_most < 0xFFFFFFFFull ? (uint32_t)chain_most : 0xFFFFFFFFu;
}

/* ---- tier and orientation choice ------------------------------------------------------------------------------------ */

/**
 *  Estimated SIMD cycles of a call and the tier that achieves them: lanes tier (lev_myers.hip / weighted.hip: one pair
 *  per lane, one query per workgroup) or systolic tier (systolic.hip: one pair per chain of wavefronts).  A throughput
 *  model good to a factor of two - which is all the decisions need, the alternatives are an order of magnitude apart
 *  wherever they matter:
 *    lanes    : a wavefront scores one query against up to 64 candidates, its lanes in lock step, `lane_rate` cells per
 *               lane-cycle; at most 1024 wavefronts advance at once; the call lasts at least as long as its largest pair;
 *    systolic : a wavefront-step scores 64 x R x K cells in `step_cycles`; a (pair, band) ticket takes len(candidate) / K
 *               + 63 steps; at most 1024 wavefronts advance at once; the call lasts at least as long as the band chain of
 *               its largest pair (each band trails its predecessor by ~95 steps).
 *  Constants from the measured gfx950 rates (profiles/r01/valu_peak.json): fast VALU ~2.5 cycles, slow ~4.2.
 *  Inputs are the per-side statistics only (hip/kernels.h: szs_side_stats_t), so that the device planner's summary feeds
 *  the same model as the host planner's length arrays.
 */
unsigned szs_plan_team_lanes(int affine, szs_side_stats_t const *queries, szs_side_stats_t const *candidates) {
    if (!queries->count || !candidates->count) return 0;
    double const mean_query = (double)queries->symbols / queries->count, mean_candidate = (double)candidates->symbols / candidates->count;
    double const items = (double)((queries->count + 1) / 2) * candidates->count; /* (pair of queries, candidate) */
    /* profiles/r03/team_sweep_v2.jsonl (query rows 24 ... 512 x candidate columns 128 / 512, every compiled shape):
     * four lanes beat one pair per lane from 24 rows (linear: +8 % there, +50 % at 48) / 40 rows (affine) up; sixteen lanes pay
     * their fifteen fill and drain steps back only over candidates of a few hundred columns, from ~190 query rows. */
    if (mean_query < (affine ? 40 : 24)) return 0; /* a few rows per lane: the step's fixed cost takes over */
    /* Round 6: the WHOLE wavefront as one team (64 lanes x 32 rows = 2048 query rows per pass; hip/weighted_teams.hip, `wave_shr:1`) is
     * compiled and selectable (`team` knob = 643202) but never chosen here: config 4's 4 KB queries are then two or three passes instead
     * of eight or nine and the call moves 12.6 GB instead of 55.5 (5.9 x its algorithmic bytes instead of 26 x) - and takes 595 ms
     * instead of 517 (7.4 TCUPS against 8.5; a share of an eighth of the rows: 78.7 against 78.4 ms).  The parked rows are 107 GB/s
     * of an 8 TB/s memory: traffic this path can afford, time it cannot (profiles/r06/team_wave_wide_cfg4.txt). */
    if (mean_query >= 192 && mean_candidate >= 256) return 16;
    if (items * 4 / 64 < 4096 && mean_query >= 128) return 16; /* four lanes per item would leave SIMDs short of wavefronts */
    return 4;
}

double szs_plan_estimate(unsigned bit_parallel_limit, int bit_parallel_chain, int affine, int uniform, int team_capable, int symmetric,
                         szs_side_stats_t const *queries, szs_side_stats_t const *candidates, unsigned band_rows, int *tier) {
    if (bit_parallel_chain) band_rows = SZS_MYERS_CHAIN_BAND_ROWS; /* hip/myers_chain.hip: 64 lanes x 32 rows x 8 columns */
    *tier = SZS_TIER_LANES;
    uint32_t const queries_count = queries->count, candidates_count = candidates->count;
    uint32_t const longest_query = queries->longest, longest_candidate = candidates->longest;
    if (!queries_count || !candidates_count) return 0;
    double const simds = 1024.0;
    double const mean_candidate = (double)candidates->symbols / candidates_count;
    double const scale = symmetric ? 0.5 : 1.0; /* half the matrix is scored */

    double const query_symbols = (double)queries->symbols;
    double const bands_total = (double)(bit_parallel_chain ? queries->bands_chain : queries->bands_systolic);

    /* lanes: cells per lane-cycle; the bit-parallel kernels only exist up to `bit_parallel_limit` symbols per query.
     * A wavefront that has its SIMD to itself issues a dependent instruction every ~8 cycles instead of every ~4
     * (scripts/wave_latency.hip), so an underfilled device runs each lane at about half its saturated rate. */
    int const bit_parallel = bit_parallel_limit && longest_query <= bit_parallel_limit;
    double const waves_per_query = (candidates_count + 63) / 64;
    double lane_waves = queries_count * waves_per_query * scale;
    /* The long bit-parallel widths (24 ... 64 words) spread a pair over 2 or 4 lanes (plan.c: szs_plan_myers_shape): that many
     * times the wavefronts, each pair that many times shorter - 128 x 128 x 1000 B runs at 29 TCUPS on this tier, 15 on the
     * chain (profiles/r02/shapes.jsonl).  Bytes only: a side of codepoints reaches here with `bit_parallel_limit` = 2048 too,
     * and its split kernels follow the same rule. */
    double split = 1.0;
    if (bit_parallel && longest_query > 640 && longest_query <= 2048 && szs_tuning_get(szs_knob_split_k) != 0) {
        int const pinned = szs_tuning_get(szs_knob_split_k);
        double const workgroups = (double)queries_count * ((candidates_count + 255) / 256);
        split = pinned == 2 || pinned == 4 ? pinned : workgroups < 1024 ? 4.0 : 2.0;
        lane_waves *= split;
    }
    /* The team tier of the 16-bit class-table scorers (hip/weighted_teams.hip): `team` lanes per (pair of queries, candidate),
     * 0.052 (affine) / 0.12 (linear) cells per lane-cycle on a full device against 0.034 / 0.079 for one pair per lane
     * (configs 4 and 3: 8.2 and 19.1 TCUPS against 6.0 and 13.2, profiles/r03), the largest pair `team` / 2 times shorter. */
    unsigned const team = team_capable && !bit_parallel_limit && szs_tuning_get(szs_knob_team_k) != 0
                              ? (szs_tuning_get(szs_knob_team_k) > 0 ? (unsigned)szs_tuning_get(szs_knob_team_k) / 10000u
                                                                    : szs_plan_team_lanes(affine, queries, candidates))
                              : 0;
    if (team) lane_waves = (double)((queries_count + 1) / 2) * candidates_count * team / 64.0 * scale, split = team / 2.0;
    if (lane_waves < 1) lane_waves = 1;
    double const fill = lane_waves >= 2 * simds ? 1.0 : lane_waves <= simds ? 0.5 : lane_waves / (2 * simds);
    double const lane_rate = (bit_parallel ? 0.85 : team ? (affine ? 0.052 : 0.12) : 1.0 / ((affine ? 7.0 : 3.0) * 4.2)) * fill;
    double lanes_cycles = query_symbols * mean_candidate * (team ? candidates_count / 64.0 : waves_per_query) * scale / lane_rate /
                          (lane_waves < simds ? lane_waves : simds);
    /* A workgroup is four wavefronts; with fewer than 193 candidates some of them have no pair at all, and the live ones
     * of neighbouring workgroups do not spread evenly over the SIMDs (4096 x 1 x 128 B measured 2x the model). */
    double const live_waves = team || waves_per_query >= 4 ? 4 : waves_per_query;
    lanes_cycles *= 1.0 + 0.5 * (1.0 - live_waves / 4.0);
    /* A team workgroup is one pair of queries x 256 / team candidates: with fewer candidates than that its other teams idle
     * (32768 x 8 scored 1.3 TCUPS in the caller's orientation, 7.0 on its side: profiles/r03/shapes.jsonl). */
    if (team && candidates_count < (team > 16 ? 512u : 256u) / team) lanes_cycles *= ((team > 16 ? 512.0 : 256.0) / team) / candidates_count;
    double const largest_pair = (double)longest_query * longest_candidate / lane_rate / split;
    if (largest_pair > lanes_cycles) lanes_cycles = largest_pair;

    /* systolic: measured on MI355X (profiles/r01/shapes_v6.jsonl): a wavefront-step of 64 lanes x 8 rows x 4 columns
     * takes ~1400 cycles with linear gaps and ~2700 with affine gaps, class-table and uniform costs alike, whether the
     * wavefront is alone on its SIMD or shares it (LDS holds two profiles per SIMD). */
    double const columns_per_step = bit_parallel_chain ? 8.0 : 4.0;
    double const step_cycles = bit_parallel_chain ? 1800.0 : affine ? 2700.0 : 1400.0;
    (void)uniform;
    double tickets = bands_total * candidates_count * scale;
    if (tickets < 1) tickets = 1;
    /* Wavefronts that advance at once: one per SIMD.  The bit-parallel chain keeps 64 KB of match masks per workgroup in
     * LDS - two workgroups per CU, each scoring up to 16 candidates against one query band - and its step is a single
     * dependency chain, so wavefronts sharing a SIMD slow each other down far less than proportionally.  Measured
     * (profiles/r01/chain_waves_v1.txt), w resident wavefronts advance like 2300 w / (w + 3200) lone ones: 2560 like
     * ~1000, 8192 like ~1650, 32768 like ~2050, half a million like ~2300. */
    double slots = simds;
    if (bit_parallel_chain) {
        double const resident = 512.0 * (candidates_count < 16 ? candidates_count : 16);
        double const advancing = tickets < resident ? tickets : resident, lone = advancing < 512.0 ? advancing : 512.0;
        slots = 2300.0 * tickets / (tickets + 3200.0);
        if (slots > advancing) slots = advancing;
        if (slots < lone) slots = lone;
    }
    double systolic_cycles = tickets * (mean_candidate / columns_per_step + 63.0) * step_cycles / (tickets < slots ? tickets : slots);
    double const longest_chain = (double)((longest_query + band_rows - 1) / band_rows);
    double const chain_cycles =
        (longest_candidate / columns_per_step + 63.0 + 95.0 * (longest_chain > 1 ? longest_chain - 1 : 0)) * step_cycles;
    if (chain_cycles > systolic_cycles) systolic_cycles = chain_cycles;

    int const forced = szs_tuning_get(szs_knob_tier_k); /* testing aid (host/tuning.c): lanes | systolic | chain */
    if (forced == SZS_TIER_LANES) return lanes_cycles;
    int const chained = bit_parallel_chain ? SZS_TIER_MYERS_CHAIN : SZS_TIER_SYSTOLIC;
    if (forced == SZS_TIER_SYSTOLIC || forced == SZS_TIER_MYERS_CHAIN) return *tier = chained, systolic_cycles;
    if (!band_rows || systolic_cycles >= 0.8 * lanes_cycles) return lanes_cycles; /* ties go to the simpler tier */
    *tier = chained;
    return systolic_cycles;
}

void szs_plan_orient(unsigned bit_parallel_limit, int bit_parallel_chain, int affine, int uniform, int team_capable, int symmetric,
                     szs_side_stats_t const *queries, szs_side_stats_t const *candidates, unsigned band_rows, int *tier,
                     int *transposed) {
    int swapped_tier = SZS_TIER_LANES;
    double const cycles = szs_plan_estimate(bit_parallel_limit, bit_parallel_chain, affine, uniform, team_capable, symmetric, queries,
                                            candidates, band_rows, tier);
    *transposed = 0;
    if (symmetric) return; /* nothing to swap */
    double const swapped_cycles = szs_plan_estimate(bit_parallel_limit, bit_parallel_chain, affine, uniform, team_capable, 0, candidates,
                                                    queries, band_rows, &swapped_tier);
    int const forced = szs_tuning_get(szs_knob_swap_k); /* testing aid: 0 | 1 */
    *transposed = forced >= 0 ? forced == 1 : swapped_cycles < 0.6 * cycles;
    if (*transposed) *tier = swapped_tier;
}

/* ---- exported probes ------------------------------------------------------------------------------------------------ */

sz_status_t szs_rocm_plan_probe(int unit_cost, int symmetric, sz_u32_t const *query_lengths, sz_size_t queries_count,
                                sz_u32_t const *candidate_lengths, sz_size_t candidates_count,
                                sz_u32_t *candidate_order, sz_u32_t *query_order, sz_u32_t *query_variant,
                                sz_u64_t *cells) {
    if (queries_count > 0xFFFFFFFFu || candidates_count > 0xFFFFFFFFu) return sz_overflow_risk_k;
    uint32_t const q = (uint32_t)queries_count, c = (uint32_t)candidates_count;
    size_t const most = q > c ? q : c;
    uint64_t *addresses = (uint64_t *)calloc(most + 1, sizeof(uint64_t));
    szs_string_ref_t *query_refs = (szs_string_ref_t *)calloc((size_t)q + 1, sizeof(szs_string_ref_t));
    szs_string_ref_t *candidate_refs = (szs_string_ref_t *)calloc((size_t)c + 1, sizeof(szs_string_ref_t));
    uint32_t *keys = (uint32_t *)calloc(most + 1, sizeof(uint32_t));
    uint32_t longest = 0;
    for (uint32_t i = 0; i < q; ++i) longest = query_lengths[i] > longest ? query_lengths[i] : longest;
    for (uint32_t i = 0; i < c; ++i) longest = candidate_lengths[i] > longest ? candidate_lengths[i] : longest;
    void *scratch = malloc(szs_plan_scratch_bytes((uint32_t)most, longest));
    if (!addresses || !query_refs || !candidate_refs || !keys || !scratch) {
        free(addresses), free(query_refs), free(candidate_refs), free(keys), free(scratch);
        return sz_bad_alloc_k;
    }
    szs_plan_t plan;
    szs_plan_build(unit_cost ? SZS_MYERS_MAX_WORDS : 0, symmetric, addresses, query_lengths, q, addresses, candidate_lengths, c, query_refs,
                   candidate_refs, keys, scratch, &plan);
    if (candidate_order)
        for (uint32_t i = 0; i < c; ++i) candidate_order[i] = candidate_refs[i].index;
    if (query_order)
        for (uint32_t i = 0; i < q; ++i) query_order[i] = query_refs[i].index;
    if (query_variant)
        for (unsigned g = 0; g < plan.groups_count; ++g)
            for (uint32_t i = 0; i < plan.groups[g].count; ++i)
                query_variant[plan.groups[g].first + i] = plan.groups[g].variant;
    if (cells) *cells = plan.cells;
    free(addresses), free(query_refs), free(candidate_refs), free(keys), free(scratch);
    return sz_success_k;
}

/** The launch of one width group of the bit-parallel kernels: `words` of the kernel that takes it (>= the group's variant:
 *  patterns are right-aligned over phantom low rows, so a wider kernel scores a narrower query exactly) and `lanes` per pair
 *  (0: one lane per pair, the long kernels).
 *
 *  24 ... 64 words always spread a pair over lanes (hip/lev_myers.hip: levenshtein_myers_split_kernel): 2 when the launch
 *  fills the device anyway, 4 under 1024 workgroups, 8 under 256 - its longest pairs ARE its duration.  16 and 20 words join
 *  them only in a launch of fewer than 256 workgroups (round 3): an eighth of config 5 on each of eight GPUs is nine launches
 *  of ~200 workgroups, and the 20-word one - 2048 columns x 20 words x 10.5 instructions in ONE lane, 1.4 ms however idle
 *  the chip - ended the call at 2.0 ms for 1.2 ms of work (profiles/r03/timeline_cfg5_eighth_v1.txt).  Ten words per lane
 *  are not whole 16-byte Peq chunks: 20 words run as 24 over two lanes.  The `split` knob pins 0 / 2 / 4 / 8. */
szs_launch_shape_t szs_plan_myers_shape(int knob, unsigned variant, uint64_t workgroups_unsplit, int runes) {
    szs_launch_shape_t shape = {variant, 0};
    if (knob == 0 || variant < 16 || variant == SZS_MYERS_SHORT_WORDS) return shape;
    int const pinned = knob == 2 || knob == 4 || knob == 8;
    unsigned lanes = pinned ? (unsigned)knob : workgroups_unsplit < 256 ? 8u : workgroups_unsplit < 1024 ? 4u : 2u;
    if (variant < 24) {
        if (!pinned && workgroups_unsplit >= 256) return shape;
        if (runes) return shape; /* codepoints: measured slower (real-text lines 50.9 against 55.7 T cells/s, an eighth of config 5u 1.88 / 1.85 ms) */
        if (variant == 20) shape.words = 24;
        else lanes = pinned && lanes >= 4 ? 4u : 2u; /* eight words per lane are as short as the other launches' pairs */
    }
    if (shape.words == 24 && lanes >= 4) lanes = 2;
    if (shape.words == 48 && lanes == 8) lanes = 4; /* six words per lane are not whole chunks either */
    if (runes) {
        if (lanes == 8) lanes = 4; /* the rune kernels are instantiated for two and four lanes */
        /* 48 and 64 words always over four lanes: their rune table leaves room for one workgroup per CU, and only the split
         * kernel puts more than one wavefront per SIMD behind it (lev_myers.hip) */
        if (!pinned && variant >= 48) lanes = 4;
    }
    shape.lanes = lanes;
    return shape;
}

/** Every width group's launch shape, and the order the launches go out in: LONGEST PAIR FIRST.  What a launch cannot go
 *  under is its longest pair - columns x words PER LANE, one dependent instruction after the other - and the first launch
 *  submitted takes every free wavefront slot: the widest group (64 words over eight lanes: many workgroups, short pairs)
 *  used to go first and the 20-word launch - one lane per pair, the longest pairs of all - got its first wavefront 0.4 ms
 *  into a 2 ms call.  Groups that need the workspace (variant 0) keep their place at the front, in order; the short launch
 *  goes last, behind the lightest long one on its stream (dispatch.c: enqueue). */
void szs_plan_launch_order(szs_plan_t const *plan, int use_myers, int runes, uint64_t candidate_blocks, int split_knob,
                           szs_launch_shape_t *shapes, unsigned *order) {
    unsigned urgency[SZS_PLAN_MAX_GROUPS];
    for (unsigned g = 0; g < plan->groups_count; ++g) {
        szs_plan_group_t const *group = &plan->groups[g];
        shapes[g] = szs_plan_myers_shape(use_myers ? split_knob : 0, group->variant, (uint64_t)group->count * candidate_blocks, runes);
        urgency[g] = group->variant == 0 ? ~0u : group->variant == SZS_MYERS_SHORT_WORDS ? 0u : shapes[g].words / (shapes[g].lanes ? shapes[g].lanes : 1u);
        unsigned at = g;
        for (; at > 0 && urgency[order[at - 1]] < urgency[g]; --at) order[at] = order[at - 1]; /* stable: ties stay widest first */
        order[at] = g;
    }
}

sz_status_t szs_rocm_queue_probe(int symmetric, sz_u32_t alphabet, sz_u32_t const *query_lengths, sz_size_t queries_count,
                                 sz_u32_t const *candidate_lengths, sz_size_t candidates_count, sz_u32_t *tiles, sz_size_t capacity,
                                 sz_size_t *tiles_count, sz_u64_t *items_total) {
    if (queries_count > 0xFFFFFFFFu || candidates_count > 0xFFFFFFFFu || !tiles_count || !items_total) return sz_overflow_risk_k;
    uint32_t const q = (uint32_t)queries_count, c = (uint32_t)candidates_count;
    size_t const most = q > c ? q : c;
    uint64_t *addresses = (uint64_t *)calloc(most + 1, sizeof(uint64_t));
    szs_string_ref_t *query_refs = (szs_string_ref_t *)calloc((size_t)q + 1, sizeof(szs_string_ref_t));
    szs_string_ref_t *candidate_refs = (szs_string_ref_t *)calloc((size_t)c + 1, sizeof(szs_string_ref_t));
    uint32_t *keys = (uint32_t *)calloc(most + 1, sizeof(uint32_t));
    uint32_t longest = 0;
    for (uint32_t i = 0; i < q; ++i) longest = query_lengths[i] > longest ? query_lengths[i] : longest;
    for (uint32_t i = 0; i < c; ++i) longest = candidate_lengths[i] > longest ? candidate_lengths[i] : longest;
    void *scratch = malloc(szs_plan_scratch_bytes((uint32_t)most, longest));
    szs_queue_plan_t *queue = (szs_queue_plan_t *)calloc(1, sizeof(szs_queue_plan_t));
    if (!addresses || !query_refs || !candidate_refs || !keys || !scratch || !queue) {
        free(addresses), free(query_refs), free(candidate_refs), free(keys), free(scratch), free(queue);
        return sz_bad_alloc_k;
    }
    szs_plan_t plan;
    szs_plan_build(SZS_MYERS_MAX_WORDS, symmetric, addresses, query_lengths, q, addresses, candidate_lengths, c, query_refs, candidate_refs, keys,
                   scratch, &plan);
    szs_plan_queue(&plan, q, c, alphabet, alphabet ? 72u << 10 : 64u << 10, queue);
    *tiles_count = queue->tiles_count, *items_total = queue->items_total;
    for (unsigned t = 0; t < queue->tiles_count && t < capacity && tiles; ++t) {
        szs_queue_tile_t const *tile = &queue->tiles[t];
        uint32_t const row[10] = {tile->first_item, tile->query_first, tile->query_count, tile->candidate_first, tile->candidate_end,
                                  tile->candidates_per_item, tile->words_per_lane, tile->lanes, tile->queries_per_item, tile->flags};
        memcpy(tiles + 10 * (size_t)t, row, sizeof(row));
    }
    free(addresses), free(query_refs), free(candidate_refs), free(keys), free(scratch), free(queue);
    return sz_success_k;
}

sz_status_t szs_rocm_launch_order_probe(int runes, sz_u32_t const *query_lengths, sz_size_t queries_count, sz_size_t candidates_count,
                                        sz_u32_t *variants, sz_u32_t *words, sz_u32_t *lanes, sz_size_t capacity, sz_size_t *launches) {
    if (queries_count > 0xFFFFFFFFu || candidates_count > 0xFFFFFFFFu || !launches) return sz_overflow_risk_k;
    szs_side_stats_t stats;
    uint32_t variant_counts[SZS_PLAN_VARIANTS];
    szs_side_stats(query_lengths, (uint32_t)queries_count, SZS_MYERS_MAX_WORDS, &stats, variant_counts);
    szs_plan_t plan;
    memset(&plan, 0, sizeof(plan));
    szs_plan_groups(variant_counts, &plan);
    szs_launch_shape_t shapes[SZS_PLAN_MAX_GROUPS];
    unsigned order[SZS_PLAN_MAX_GROUPS];
    uint64_t const candidate_blocks = ((uint64_t)candidates_count + SZS_CANDIDATES_PER_WORKGROUP - 1) / SZS_CANDIDATES_PER_WORKGROUP;
    szs_plan_launch_order(&plan, 1, runes, candidate_blocks, szs_tuning_get(szs_knob_split_k), shapes, order);
    *launches = plan.groups_count;
    for (unsigned turn = 0; turn < plan.groups_count && turn < capacity; ++turn) {
        if (variants) variants[turn] = plan.groups[order[turn]].variant;
        if (words) words[turn] = shapes[order[turn]].words;
        if (lanes) lanes[turn] = shapes[order[turn]].lanes;
    }
    return sz_success_k;
}

sz_status_t szs_rocm_team_orientation_probe(int affine, int symmetric, sz_u32_t const *query_lengths, sz_size_t queries_count,
                                            sz_u32_t const *candidate_lengths, sz_size_t candidates_count, int *tier, int *transposed,
                                            sz_u32_t *lanes) {
    if (queries_count > 0xFFFFFFFFu || candidates_count > 0xFFFFFFFFu || !tier || !transposed || !lanes) return sz_overflow_risk_k;
    szs_side_stats_t query_stats, candidate_stats;
    szs_side_stats(query_lengths, (uint32_t)queries_count, 0, &query_stats, NULL);
    szs_side_stats(symmetric ? query_lengths : candidate_lengths, (uint32_t)(symmetric ? queries_count : candidates_count), 0,
                   &candidate_stats, NULL);
    szs_plan_orient(0, 0, affine, 0, 1, symmetric, &query_stats, &candidate_stats, SZS_SYSTOLIC_BAND_ROWS, tier, transposed);
    *lanes = *tier == SZS_TIER_LANES ? szs_plan_team_lanes(affine, *transposed ? &candidate_stats : &query_stats,
                                                           *transposed ? &query_stats : &candidate_stats)
                                     : 0;
    return sz_success_k;
}

sz_status_t szs_rocm_orientation_probe(int unit_cost, int affine, int uniform, int symmetric,
                                       sz_u32_t const *query_lengths, sz_size_t queries_count,
                                       sz_u32_t const *candidate_lengths, sz_size_t candidates_count, int *tier,
                                       int *transposed) {
    if (queries_count > 0xFFFFFFFFu || candidates_count > 0xFFFFFFFFu || !tier || !transposed) return sz_overflow_risk_k;
    szs_side_stats_t query_stats, candidate_stats;
    szs_side_stats(query_lengths, (uint32_t)queries_count, 0, &query_stats, NULL);
    szs_side_stats(symmetric ? query_lengths : candidate_lengths, (uint32_t)(symmetric ? queries_count : candidates_count), 0,
                   &candidate_stats, NULL);
    szs_plan_orient(unit_cost ? 0xFFFFFFFFu : 0, unit_cost, affine, uniform, 0, symmetric, &query_stats, &candidate_stats, /* as dispatch.c does for bytes */
                    SZS_SYSTOLIC_BAND_ROWS, tier, transposed);
    return sz_success_k;
}

typedef struct {
    uint64_t weight;
    uint32_t row;
} weighted_row_t;

static int compare_rows_descending(void const *a, void const *b) {
    weighted_row_t const *x = (weighted_row_t const *)a, *y = (weighted_row_t const *)b;
    if (x->weight != y->weight) return x->weight > y->weight ? -1 : 1;
    return x->row < y->row ? -1 : x->row > y->row;
}

sz_status_t szs_rocm_shard_triangle(sz_size_t const *lengths, sz_size_t rows, sz_size_t shards, sz_size_t *band_first, sz_u64_t *band_weights) {
    /* Row i of the lower triangle meets the strings 0 ... i: weight (len_i + 1) x sum_{j <= i} (len_j + 1) - cells, plus one per
     * string so that empty strings still count as pairs.  Bands are CONTIGUOUS (band g = rows [first[g], first[g + 1])): a band
     * is then a rectangle - its rows against everything before it - and a square triangle of its own, two ordinary engine calls.
     * The cuts go where the running weight crosses g / shards of the total; a row is at most 2 / rows of it. */
    if (!shards || !band_first || (rows && !lengths)) return sz_unexpected_dimensions_k;
    long double total = 0, prefix = 0;
    for (size_t i = 0; i < rows; ++i) prefix += (long double)lengths[i] + 1, total += ((long double)lengths[i] + 1) * prefix;
    size_t band = 0;
    long double running = 0, before = 0;
    prefix = 0;
    band_first[0] = 0;
    for (size_t i = 0; i < rows; ++i) {
        prefix += (long double)lengths[i] + 1;
        long double const weight = ((long double)lengths[i] + 1) * prefix;
        /* row i opens the next band when the bands so far hold their share (the row goes to whichever side leaves them closer) */
        while (band + 1 < shards && running + weight / 2 >= total * (long double)(band + 1) / (long double)shards) {
            if (band_weights) band_weights[band] = (sz_u64_t)(running - before);
            before = running, band_first[++band] = i;
        }
        running += weight;
    }
    if (band_weights) band_weights[band] = (sz_u64_t)(running - before);
    while (band + 1 < shards) { /* fewer rows than shards: the remaining bands are empty */
        band_first[++band] = rows;
        if (band_weights) band_weights[band] = 0;
    }
    band_first[shards] = rows;
    return sz_success_k;
}

sz_status_t szs_rocm_shard_rows(sz_size_t const *row_weights, sz_size_t rows, sz_size_t shards, sz_u32_t *shard_of_row,
                                sz_u64_t *shard_loads) {
    if (!shards || !shard_of_row || (rows && !row_weights) || rows > 0xFFFFFFFFu) return sz_unexpected_dimensions_k;
    weighted_row_t *sorted = (weighted_row_t *)malloc((rows + 1) * sizeof(weighted_row_t));
    uint64_t *loads = (uint64_t *)calloc(shards, sizeof(uint64_t));
    if (!sorted || !loads) {
        free(sorted), free(loads);
        return sz_bad_alloc_k;
    }
    for (size_t i = 0; i < rows; ++i) sorted[i].weight = row_weights[i], sorted[i].row = (uint32_t)i;
    qsort(sorted, rows, sizeof(weighted_row_t), compare_rows_descending);
    for (size_t i = 0; i < rows; ++i) { /* heaviest first, always onto the lightest shard */
        size_t lightest = 0;
        for (size_t s = 1; s < shards; ++s)
            if (loads[s] < loads[lightest]) lightest = s;
        loads[lightest] += sorted[i].weight;
        shard_of_row[sorted[i].row] = (uint32_t)lightest;
    }
    if (shard_loads) memcpy(shard_loads, loads, shards * sizeof(uint64_t));
    free(sorted), free(loads);
    return sz_success_k;
}
