/*
 *  team_core.hpp - the arithmetic of the TEAM tier of the weighted scorers (hip/weighted_teams.hip), kept apart from the
 *  kernel so that the very same code is compiled twice: by hipcc into the gfx950 kernel, and by g++ into
 *  tests/native/team_model.cpp, a lane-by-lane model of the kernel's data flow that is checked against the oracle on the
 *  CPU (no GPU needed to find an off-by-one in a border, a carry between the halves of a register or a skewed hand-over).
 *
 *  The recurrences are the reference's tile_scorer ones (/root/reference/include/stringzillas/similarities/serial.hpp:
 *  778-876 global linear, 890-988 local linear, 1002-1137 global affine, 1151-1278 local affine) and must give the same
 *  scores bit for bit.  What is specific to this tier:
 *
 *  TWO QUERIES PER REGISTER.  A 32-bit register holds the same DP row index of TWO queries - the low half belongs to query
 *  2p, the high half to query 2p + 1 of the (longest first) query list - scored against the SAME candidate symbol.  Both
 *  halves are therefore always at the same column and need the same cost row: one LDS read of a profile keyed by ONE class
 *  delivers the pair (weighted_packed.hip pairs two columns of one query instead and needs a profile keyed by a PAIR of
 *  classes: 40 KB per strip for BLOSUM62 where this one takes 3 KB - which is what makes room for sixteen strips at once).
 *
 *  BIASED UNSIGNED CELLS, FULL-RATE ADDITIONS.  A cell is stored as `true value + bias` in an unsigned 16-bit half.  With
 *  every half provably inside [0, 65535] before and after an addition, ONE 32-bit `v_add_u32` adds a (signed) increment to
 *  both halves at once: the increment pair is stored as `(high << 16) + sign_extended(low)`, i.e. the borrow that a negative
 *  low increment takes from the high half is paid back in advance.  gfx950 issues v_add_u32 at 59 T lane-operations/s and
 *  every maximum, packed or not, integer or float, at 35 (profiles/r03/team_ops.json), so the additions of the recurrence
 *  cost 0.6 of what they cost weighted_packed.hip's v_pk_add_i16.
 *
 *  THREE-INPUT MAXIMA ON HALF-FLOAT PATTERNS.  gfx950 has no three-input integer maximum on packed halves, but it has
 *  `v_pk_maximum3_f16`, at the rate of the two-input ones - and positive NORMAL half floats (bit patterns 0x0400 ... 0x7BFF)
 *  order exactly like their patterns read as unsigned integers.  Cells kept inside that range are therefore maximised three
 *  at a time by a floating-point instruction that never sees a float (profiles/r03/team_ops.json: 0 differences from the
 *  integer maximum on 2^26 triples).  The recurrence's maxima per register (= per two cells) drop from 6 to 4.5 (local
 *  affine), 4 to 3 (global affine), 2 to 1 (global linear).  `narrow` order, the default:
 *      global (Needleman-Wunsch): bias 16384; the reach rule (serial.hpp:135-162) must stay below 15000.
 *      local  (Smith-Waterman, gap costs <= 0): bias 2048; H >= 0 and the gap tracks >= open + 2 extend, so nothing sinks
 *              below 1792; scores must stay below 29000.
 *  `wide` order - plain v_pk_max_u16, two inputs, every pattern valid - covers what lies between those bounds and 16 bits:
 *      global: bias 32768, reach below 32000; local: bias 1024, scores below 62000.
 *  Garbage is harmless in either order: a half can only leave its range in rows BELOW the last row of the longer query
 *  (padded rows, global borders running on); what a low half then carries into the high half lands in a row that is padded
 *  for the shorter query too; carries never run downwards (high to low), the maxima treat the halves separately, and a NaN
 *  pattern only ever propagates down and to the right, like everything else in the matrix.
 *
 *  TWO TRACKS INSTEAD OF THREE.  weighted_packed.hip keeps H, H + open and E + extend per row; here a row keeps H and
 *  `e` = the horizontal-gap value ENTERING the next column, max(H + open, E + extend), which is formed from values the step
 *  has at hand anyway (`H + open` also feeds the vertical track): the same number of operations, a third fewer registers.
 *  Linear gaps keep ONE track, `g = H + gap`; the `- gap` that the diagonal needs is folded into the cost profile.
 */
#ifndef SZS_TEAM_CORE_HPP_
#define SZS_TEAM_CORE_HPP_

#include <stdint.h>

#if defined(__HIPCC__)
#define SZS_HD __host__ __device__ __forceinline__
#else
#define SZS_HD inline
#endif

namespace szs_team {

using u32 = uint32_t;
using i32 = int32_t;

/** An INCREMENT (or a value) for both halves of a register: adding it with one 32-bit addition adds `low` to the low half
 *  and `high` to the high half, as long as the low half stays inside [0, 65535]. */
SZS_HD u32 pair_of(i32 low, i32 high) { return ((u32)high << 16) + (u32)low; }
SZS_HD u32 both(i32 value) { return pair_of(value, value); }
SZS_HD i32 low_of(u32 pair) { return (i32)(pair & 0xFFFFu); }
SZS_HD i32 high_of(u32 pair) { return (i32)(pair >> 16); }

#if !defined(__HIP_DEVICE_COMPILE__)
/** Host model of the IEEE-754 `maximum` on one half-float PATTERN: a NaN in, the canonical NaN out; negative patterns order
 *  downwards.  The kernel's claims about its value ranges are only as good as a model that breaks when they are wrong. */
inline u32 model_half_maximum(u32 a, u32 b) {
    auto const is_nan = [](u32 x) { return (x & 0x7FFFu) > 0x7C00u; };
    if (is_nan(a) || is_nan(b)) return 0x7E00u;
    auto const key = [](u32 x) { return (x & 0x8000u) ? -(i32)(x & 0x7FFFu) : (i32)(x & 0x7FFFu); };
    return key(a) >= key(b) ? a : b;
}
#endif

/** The two orders a kernel instance can keep its cells in (see the header). */
template <bool wide_>
struct team_order {
    /** Per-half maximum of two registers: v_pk_max_u16 / v_pk_max_f16. */
    SZS_HD static u32 max2(u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (wide_) {
            typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));
            return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(pk_u16, a), __builtin_bit_cast(pk_u16, b)));
        }
        // IEEE-754 `maximum` through the compiler's own builtin, NOT inline assembly: a packed result needs a wait state
        // before its first reader on gfx950, and only instructions the hazard recogniser can see get one.
        typedef _Float16 pk_f16 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(u32, __builtin_elementwise_maximum(__builtin_bit_cast(pk_f16, a), __builtin_bit_cast(pk_f16, b)));
#else
        if (wide_) {
            u32 const low = (a & 0xFFFFu) > (b & 0xFFFFu) ? (a & 0xFFFFu) : (b & 0xFFFFu);
            u32 const high = (a >> 16) > (b >> 16) ? (a >> 16) : (b >> 16);
            return (high << 16) | low;
        }
        return model_half_maximum(a & 0xFFFFu, b & 0xFFFFu) | model_half_maximum(a >> 16, b >> 16) << 16;
#endif
    }
    /** Per-half maximum of three registers: ONE v_pk_maximum3_f16 in the narrow order, two v_pk_max_u16 in the wide one. */
    SZS_HD static u32 max3(u32 a, u32 b, u32 c) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (!wide_) { // the nested maxima fold into one v_pk_maximum3_f16
            typedef _Float16 pk_f16 __attribute__((ext_vector_type(2)));
            pk_f16 const first = __builtin_elementwise_maximum(__builtin_bit_cast(pk_f16, a), __builtin_bit_cast(pk_f16, b));
            return __builtin_bit_cast(u32, __builtin_elementwise_maximum(first, __builtin_bit_cast(pk_f16, c)));
        }
#endif
        return max2(max2(a, b), c);
    }
};

/** The scoring constants of one launch in the chosen representation.
 *  `distance_`: a Levenshtein engine - global alignment over NEGATED non-negative costs, so every value lies in [-reach, 0]
 *  and the bias sits at the top of the range: twice the reach of a Needleman-Wunsch engine in the same bits. */
template <bool local_, bool affine_, bool wide_, bool distance_ = false>
struct team_costs_t {
    static_assert(!(local_ && distance_), "a distance is a global objective");
    static constexpr bool local = local_, affine = affine_, wide = wide_, distance = distance_;
    static constexpr i32 bias = wide_ ? (distance_ ? 65024 : local_ ? 1024 : 32768) : (distance_ ? 31232 : local_ ? 2048 : 16384);
    using order = team_order<wide_>;

    i32 open, extend; // signed, ADDED; linear gaps: open == extend == the gap cost
    u32 open_pair, extend_pair, zero_pair;

    SZS_HD team_costs_t(i32 gap_open, i32 gap_extend)
        : open(gap_open), extend(affine_ ? gap_extend : gap_open), open_pair(both(gap_open)),
          extend_pair(both(affine_ ? gap_extend : gap_open)), zero_pair(both(bias)) {}

    /** H(k, 0) = H(0, k): the all-gap border (serial.hpp:821-823,1045-1047); local: 0. */
    SZS_HD i32 border(u32 k) const {
        if (local_) return 0;
        if (affine_) return k ? open + extend * (i32)(k - 1) : 0;
        return open * (i32)k;
    }
    /** The gap track ENTERING the first real column / row next to a border cell of value `edge`: the reference seeds the
     *  track with the finite "discard" value edge + open + extend (serial.hpp:1049-1056, local: 1195-1201) and the first
     *  step forms max(edge + open, seed + extend). */
    SZS_HD i32 entering(i32 edge) const {
        i32 const fresh = edge + open, carried = edge + open + 2 * extend;
        return fresh > carried ? fresh : carried;
    }
    SZS_HD u32 stored(i32 truth) const { return (u32)(truth + bias) & 0xFFFFu; }
    SZS_HD i32 truth(u32 half) const { return (i32)half - bias; }

    /** One profile entry: the costs of one candidate class against the same row of the two queries (0 for a padded row).
     *  Linear gaps: the diagonal is kept as H + gap, so the entry carries `cost - gap`. */
    SZS_HD u32 profile_entry(i32 cost_low, i32 cost_high) const {
        return affine_ ? pair_of(cost_low, cost_high) : pair_of(cost_low - open, cost_high - open);
    }
};

/** Largest worst-case magnitude a call may have for an instance: the reach of serial.hpp:135-162 (global, distance), or
 *  (shorter side + 3) x largest cost (local).  The host checks it (dispatch.c).  `objective`: 0 global, 1 local, 2 distance. */
SZS_HD u32 team_reach_limit(int objective, bool wide) {
    return wide ? (objective == 2 ? 64000u : objective == 1 ? 62000u : 32000u) : (objective == 2 ? 30000u : objective == 1 ? 29000u : 15000u);
}

/**
 *  The rows of one strip at one column, per lane.
 *  affine: h[r] = H(row r, column), e[r] = max(H + open, E + extend) = the horizontal-gap track entering the NEXT column.
 *  linear: h[r] = H(row r, column) + gap; `e` is unused.
 */
template <bool affine_, int R>
struct team_rows_t {
    u32 h[R];
    u32 e[affine_ ? R : 1];
};

/** What travels from a strip to the strip below it, per column: H of the bottom row (linear: + gap) and, affine, the
 *  vertical-gap track entering the row below. */
struct team_edge_t {
    u32 h, f;
};

/** Column 0 of the strip whose first DP row (1-based) is `first_row + 1`, for a pair of queries alike. */
template <typename costs_t, int R>
SZS_HD void team_seed(costs_t const &k, u32 first_row, team_rows_t<costs_t::affine, R> &rows, u32 &diagonal) {
    for (int r = 0; r < R; ++r) {
        i32 const edge = k.border(first_row + (u32)r + 1);
        if (costs_t::affine) rows.h[r] = both(k.stored(edge)), rows.e[costs_t::affine ? r : 0] = both(k.stored(k.entering(edge)));
        else rows.h[r] = both(k.stored(edge + k.open));
    }
    diagonal = both(k.stored(k.border(first_row) + (costs_t::affine ? 0 : k.open)));
}

/** What the strip below DP row 0 - the border row - hands down at column `j` (1-based): prefilled into the parked rows. */
template <typename costs_t>
SZS_HD team_edge_t team_border_edge(costs_t const &k, u32 j) {
    i32 const edge = k.border(j);
    team_edge_t out;
    out.h = both(k.stored(edge + (costs_t::affine ? 0 : k.open)));
    out.f = costs_t::affine ? both(k.stored(k.entering(edge))) : 0u;
    return out;
}

/** Pins a value: no instruction, but the compiler may neither re-associate a maximum through it nor sink its operands.
 *  What hangs OFF the row-to-row chain must be formed off it; left alone, hipcc folds the clamp behind the maximum with
 *  the vertical track and the chain grows by one slow operation and a wait state per row. */
SZS_HD void settle(u32 &value) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(value));
#else
    (void)value;
#endif
}

/**
 *  One step of one lane: the R rows of its strip advance by one column - begin(), rows in order, end().
 *  `above`: what the strip above handed down at this column; `diagonal`: its `h` of the previous column (replaced).
 *  `cost`: the profile entry of this column's class for the row at hand.  `best`: running maxima of H (local only).
 *
 *  Operations per register = per TWO cells (additions v_add_u32 + maxima):
 *                      narrow order                         wide order
 *      global linear   2 + 1   max3(sub, left, up)          2 + 2
 *      local  linear   2 + 2.5 max3(sub, left, 0); up; best 2 + 4
 *      global affine   4 + 3   max3(e, sub, f); e'; f'      4 + 4
 *      local  affine   4 + 4.5 max3(e, sub, 0); f; e'; f'   4 + 6
 *  (`best` takes two rows per maximum in the narrow order).  The chain from one row to the next is one maximum and one
 *  addition (linear) or two maxima and one addition (affine); everything else hangs off it.
 */
template <typename costs_t, int R>
struct team_step_t {
    static constexpr bool local_ = costs_t::local, affine_ = costs_t::affine, wide_ = costs_t::wide;
    using order = typename costs_t::order;
    u32 entering; // the substitution branch of the row about to be scored: H of the row above at the previous column + its cost
    u32 up;       // affine: F entering the row, linear: H + gap of the row above
    u32 pending;  // local, narrow order: the cell of the even row, waiting for its odd neighbour
    u32 bottom;   // what the last row scored hands down as `h`

    SZS_HD void begin(team_edge_t const &above, u32 &diagonal, u32 first_cost) {
        entering = diagonal + first_cost, diagonal = above.h;
        up = affine_ ? above.f : above.h;
        pending = 0, bottom = 0;
    }
    /** Row `r`: serial.hpp:1091-1102, 1238-1239 (affine), 846-848, 957-965 (linear).  `next_cost` is the cost of row r + 1
     *  (anything for the last row): the NEXT row's substitution branch is formed here, from this row's H of the previous
     *  column, before that register is overwritten - kept for the next row instead (round 3's first version), every row cost
     *  a `v_mov_b32`: 42 of the 146 VALU instructions of a linear step. */
    SZS_HD void row(costs_t const &k, team_rows_t<affine_, R> &rows, int r, u32 next_cost, u32 (&best)[4]) {
        u32 const substituted = entering; // linear: (H(row - 1, column - 1) + gap) + (cost - gap)
        if (r + 1 < R) entering = rows.h[r] + next_cost;
        u32 const across = affine_ ? rows.e[affine_ ? r : 0] : rows.h[r]; // the horizontal branch: E, or left + gap
        u32 cell;
        if (local_) {
            u32 inner = order::max3(across, substituted, k.zero_pair); // only the substitution branch needs the clamp; the tracks lose anyway
            settle(inner);
            cell = order::max2(inner, up);
            if (wide_) best[r % 4] = order::max2(best[r % 4], cell);
            else if (r % 2) best[r / 2 % 4] = order::max3(best[r / 2 % 4], pending, cell);
            else pending = cell;
        }
        else if (wide_) {
            u32 inner = order::max2(across, substituted);
            settle(inner);
            cell = order::max2(inner, up);
        }
        else cell = order::max3(across, substituted, up);
        u32 const opened = cell + k.open_pair;
        if (affine_) {
            rows.h[r] = bottom = cell;
            rows.e[affine_ ? r : 0] = order::max2(opened, rows.e[affine_ ? r : 0] + k.extend_pair);
            up = order::max2(opened, up + k.extend_pair);
        }
        else rows.h[r] = up = bottom = opened;
    }
    /** After the rows that the pass holds - all R of them, or fewer in the last pass (team_pass_registers). */
    SZS_HD team_edge_t end() const {
        team_edge_t below;
        below.h = bottom, below.f = affine_ ? up : 0u;
        return below;
    }
};

/** The first `registers` rows of a strip (a multiple of four, at most R). */
template <typename costs_t, int R>
SZS_HD team_edge_t team_advance(costs_t const &k, team_rows_t<costs_t::affine, R> &rows, u32 const *costs, team_edge_t above,
                                u32 &diagonal, u32 (&best)[4], u32 registers) {
    team_step_t<costs_t, R> step;
    step.begin(above, diagonal, costs[0]);
    for (u32 r = 0; r < registers; ++r) step.row(k, rows, (int)r, r + 1 < registers ? costs[r + 1] : 0u, best);
    return step.end();
}

/* ---- passes ------------------------------------------------------------------------------------------------------------
 *  A team walks the longer query of its pair in passes of L x R rows, lane k on the rows [first + k R, first + (k + 1) R).
 *  The LAST pass rarely has L x R rows left: it deals what is left in chunks of four registers or (global scores under linear
 *  gaps, round 6) in pairs of them, the same number to every lane, so that a pair wastes fewer than 4 L or 2 L padded rows instead of half
 *  a pass on average (config 3's 512-row proteins on sixteen lanes x sixteen registers: a quarter of all rows). */

/** Passes over the candidate that a pair of queries needs: the LONGER one decides. */
template <int L, int R>
SZS_HD u32 team_passes(u32 longer_query) { return (longer_query + (u32)L * R - 1) / ((u32)L * R); }

/** The registers that a short last pass is rounded up to: a pair for global scores under linear gaps, a chunk of four otherwise. */
template <bool local_, bool affine_>
SZS_HD constexpr int team_granule() { return local_ || affine_ ? 4 : 2; }

/** Registers (rows per lane) of pass `pass`: R, or in the last pass the rows left over L lanes, rounded up to `G` of them. */
template <int L, int R, int G>
SZS_HD u32 team_pass_registers(u32 longer_query, u32 pass) {
    static_assert(G == 2 || G == 4, "a profile row is read four registers at a time: whole chunks, or whole chunks and half a one");
    u32 const left = longer_query - pass * (u32)L * R;
    // (round 6: whole PAIRS of registers under linear gaps - a pair of queries wastes fewer than 2 L padded rows in its last pass;
    // config 3's 385 ... 640-row proteins on sixteen lanes padded 32 rows on average, 7 % of a 450-row pass, now 16.  The affine
    // and the local kernels keep whole chunks: one copy of the main loop per count of pairs took the affine four-lane shape from
    // 4 spilled registers to 39 and the local linear ones from 17 to 32, and config 4 gained 0.7 % where config 3 gained 2.2 %)
    return left >= (u32)L * R ? (u32)R : (left + (u32)G * L - 1) / ((u32)G * L) * (u32)G;
}

/** Where the last DP row of a query of `length` > 0 lives when its pair's longer query has `longer_query` rows. */
template <int L, int R, int G>
SZS_HD void team_last_row(u32 length, u32 longer_query, u32 &pass, u32 &lane, u32 &reg) {
    u32 const row = length - 1;
    pass = row / ((u32)L * R);
    u32 const registers = team_pass_registers<L, R, G>(longer_query, pass), within = row - pass * (u32)L * R;
    lane = within / registers, reg = within % registers;
}

/* ---- the cost profile in LDS ------------------------------------------------------------------------------------------
 *  One row of R entries (R x 4 bytes) per (strip, class); a lane reads its row with R / 4 `ds_read_b128` at immediate offsets.
 *  `ds_read_b128` serves a wavefront in four groups of 16 lanes and a group is conflict-free when its lanes' 16-byte
 *  addresses differ modulo 256 (identical addresses broadcast).  The lanes of a team read DIFFERENT strips and teams (mostly)
 *  different classes:
 *    L = 16  a lane group holds each strip exactly once, so (address / 16) mod 16 is made a function of the strip alone: the
 *            strips are dealt over `blocks` regions whose sizes are 16 modulo 256, and inside a region the rows of `slots`
 *            strips of one class share a block of 256 bytes.  No conflicts whatever the classes (PMC, config 4: 0.0 %).
 *    L > 16  (round 6: teams of 32 or 64 lanes, handed over by `wave_shr:1`) every lane group still holds sixteen DIFFERENT strips
 *            whose numbers are distinct modulo 16 (the groups are {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32): strip k
 *            lies where strip k mod 16 of the sixteen-lane layout lies, in quarter k / 16 - the quarters a multiple of 256 bytes apart.
 *    L < 16  a lane group holds 16 / L teams with a class each.  Rows of R x 4 + 16 bytes, classes L rows + 16 bytes apart:
 *            the lanes of one team never collide, two teams collide on a lane or two for some class differences - the first
 *            layout (classes 256 bytes apart) had every team of a group on the SAME banks: 72 % of the LDS cycles of a
 *            four-lane launch were conflicts, and the launch was LDS-bound (profiles/r03/pmc_configs.json, cfg7). */
template <int L, int R>
struct team_profile_layout {
    static constexpr u32 row_bytes = (u32)R * 4;
    static constexpr bool whole_row = L >= 16; // every strip (modulo 16) once per lane group
    static constexpr u32 inner = L >= 16 ? 16u : (u32)L; // strips that share one sixteen-lane layout
    static constexpr u32 slots = !whole_row ? 1 : row_bytes >= 256 ? 1 : (256 / row_bytes < inner ? 256 / row_bytes : inner); // strips per class block
    static constexpr u32 blocks = whole_row ? inner / slots : 1;
    static constexpr u32 strip_bytes = row_bytes + 16; // L < 16
    static constexpr u32 class_bytes = !whole_row ? (u32)L * strip_bytes + 16 : slots > 1 ? 256 : row_bytes;
    SZS_HD static u32 region_bytes(u32 classes) { return classes * class_bytes + 16; }
    /** L > 16: the bytes of sixteen strips, rounded up to the 256 bytes the banks repeat after. */
    SZS_HD static u32 quarter_bytes(u32 classes) { return (blocks * region_bytes(classes) + 255u) / 256u * 256u; }
    SZS_HD static u32 total_bytes(u32 classes) { return L > 16 ? (u32)(L / 16) * quarter_bytes(classes) : blocks * region_bytes(classes); }
    /** Byte offset of the row of strip `k`, class 0; the row of class c lies c x class_bytes further. */
    SZS_HD static u32 strip_base(u32 k, u32 classes) {
        if (L > 16) return (k / 16) * quarter_bytes(classes) + ((k % 16) % blocks) * region_bytes(classes) + ((k % 16) / blocks) * row_bytes;
        return whole_row ? (k % blocks) * region_bytes(classes) + (k / blocks) * row_bytes : k * strip_bytes;
    }
};

} // namespace szs_team

#endif /* SZS_TEAM_CORE_HPP_ */
