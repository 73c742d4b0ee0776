/*
 *  team_model.cpp - a lane-by-lane CPU model of the TEAM tier's data flow (stringzilla_amd/csrc/hip/weighted_teams.hip).
 *
 *  TEST INFRASTRUCTURE.  It compiles the kernel's own arithmetic (csrc/hip/team_core.hpp: representation, seeds, border
 *  edges, the step, the profile entries) with g++ and drives it exactly the way the kernel does - L lanes per (pair of
 *  queries, candidate), lane k one column behind lane k - 1, hand-over from lane to lane, strip groups parked between passes,
 *  fill / unpredicated / drain phases decided per wavefront - so that tests/test_team_model.py can compare every engine
 *  family with the oracle WITHOUT a GPU.  What it cannot see is what only exists on the device: DPP encodings, LDS
 *  addressing, the text stream.  Those are covered by the `-m gpu` parity tests.
 */
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../stringzilla_amd/csrc/hip/team_core.hpp"

using namespace szs_team;

namespace {

struct text_t {
    uint8_t const *bytes;
    uint32_t length;
};

template <bool local_, bool affine_, bool wide_, bool distance_, int L, int R>
void score_item(text_t q_low, text_t q_high, bool has_high, std::vector<text_t> const &block, uint8_t const *byte_to_class,
                int8_t const *class_costs, int open, int extend, int64_t *out_low, int64_t *out_high) {
    constexpr uint32_t teams_per_wave = 64 / L;
    using costs_t = team_costs_t<local_, affine_, wide_, distance_>;
    costs_t const k(open, extend);
    uint32_t const teams = (uint32_t)block.size();
    uint32_t const longer = q_low.length; // the queries arrive longest first
    for (uint32_t team = 0; team < teams; ++team) {
        // an empty side never enters the loop (serial.hpp:1366-1373, 1594-1605, 3077-3080)
        if (!q_low.length) out_low[team] = k.border(block[team].length);
        if (has_high && !q_high.length) out_high[team] = k.border(block[team].length);
    }
    if (!longer) return;

    uint32_t longest_text = 0;
    for (auto const &text : block) longest_text = std::max(longest_text, text.length);
    // parked rows [column][team]: poisoned - nothing may read an entry before a tail lane has written it
    std::vector<team_edge_t> parked((size_t)(longest_text + 1) * teams, team_edge_t {0xDEADBEEFu, 0xDEADBEEFu});

    uint32_t const passes = team_passes<L, R>(longer);
    struct lane_t {
        team_rows_t<affine_, R> rows;
        uint32_t diagonal;
        uint32_t best[4];
        team_edge_t out;
        uint32_t out_class;
    };
    std::vector<lane_t> lanes((size_t)teams * L);
    for (auto &lane : lanes)
        for (auto &best : lane.best) best = k.zero_pair;

    for (uint32_t pass = 0; pass < passes; ++pass) {
        uint32_t const first_row = pass * L * R;
        uint32_t const registers = team_pass_registers<L, R, team_granule<local_, affine_>()>(longer, pass); // rows per lane in this pass
        // ---- the profile of the L strips: [strip][class][register]
        std::vector<uint32_t> profile((size_t)L * 33 * R);
        for (uint32_t strip = 0; strip < (uint32_t)L; ++strip)
            for (uint32_t symbol_class = 0; symbol_class < 33; ++symbol_class)
                for (uint32_t r = 0; r < (uint32_t)R; ++r) {
                    uint32_t const row = r < registers ? first_row + strip * registers + r : 0xFFFFFFFFu;
                    int const low = row < q_low.length && symbol_class < 32 ? class_costs[byte_to_class[q_low.bytes[row]] * 32 + symbol_class] : 0;
                    int const high = has_high && row < q_high.length && symbol_class < 32 ? class_costs[byte_to_class[q_high.bytes[row]] * 32 + symbol_class] : 0;
                    profile[((size_t)strip * 33 + symbol_class) * R + r] = k.profile_entry(low, high);
                }
        for (uint32_t team = 0; team < teams; ++team)
            for (uint32_t lane = 0; lane < (uint32_t)L; ++lane) {
                lane_t &state = lanes[(size_t)team * L + lane];
                team_seed<costs_t, R>(k, first_row + lane * registers, state.rows, state.diagonal);
            }

        for (uint32_t wave_first = 0; wave_first < teams; wave_first += teams_per_wave) {
            uint32_t const wave_teams = std::min(teams_per_wave, teams - wave_first);
            uint32_t shortest = 0xFFFFFFFFu, longest = 0;
            for (uint32_t w = 0; w < wave_teams; ++w)
                shortest = std::min(shortest, block[wave_first + w].length), longest = std::max(longest, block[wave_first + w].length);
            if (!longest) continue;
            uint32_t const fill = (uint32_t)((L - 1 + 3) / 4 * 4);
            auto step = [&](uint32_t t, bool predicated) {
                // hand-over first (every lane, active or not), then the step
                for (uint32_t w = 0; w < wave_teams; ++w) {
                    uint32_t const team = wave_first + w;
                    text_t const &text = block[team];
                    team_edge_t in[L];
                    uint32_t in_class[L];
                    for (int lane = L - 1; lane >= 1; --lane)
                        in[lane] = lanes[(size_t)team * L + lane - 1].out, in_class[lane] = lanes[(size_t)team * L + lane - 1].out_class;
                    uint32_t const head_column = t + 1;
                    if (head_column <= text.length) {
                        // the first pass COMPUTES what DP row 0 hands down (as the kernel does since round 6); later passes read what the
                        // tail lanes of the previous pass parked
                        in[0] = pass == 0 ? team_border_edge(k, head_column) : parked[(size_t)head_column * teams + team];
                        in_class[0] = byte_to_class[text.bytes[head_column - 1]];
                    }
                    else in[0] = team_edge_t {0xDEADBEEFu, 0xDEADBEEFu}, in_class[0] = 0;
                    for (uint32_t lane = 0; lane < (uint32_t)L; ++lane) {
                        lane_t &state = lanes[(size_t)team * L + lane];
                        int64_t const column = (int64_t)t - lane + 1;
                        bool const active = column >= 1 && column <= (int64_t)text.length;
                        if (!predicated && !active) __builtin_trap(); // the unpredicated loop must only see active lanes
                        if (!active) continue; // as in the kernel's predicated step: a lane without a column keeps what it last produced
                        uint32_t const *costs = &profile[((size_t)lane * 33 + in_class[lane]) * R];
                        state.out = team_advance<costs_t, R>(k, state.rows, costs, in[lane], state.diagonal, state.best, registers);
                        state.out_class = in_class[lane];
                        if (lane == (uint32_t)L - 1 && pass + 1 < passes) parked[(size_t)column * teams + team] = state.out;
                    }
                }
            };
            uint32_t t = 0;
            for (; t < fill; ++t) step(t, true);
            for (; t + 4 <= shortest; t += 4)
                for (uint32_t s = 0; s < 4; ++s) step(t + s, false);
            for (; t < longest + L - 1; ++t) step(t, true);
        }

        // ---- results that are complete after this pass
        for (int half = 0; half < (has_high ? 2 : 1); ++half) {
            text_t const &query = half ? q_high : q_low;
            int64_t *const out = half ? out_high : out_low;
            if (!query.length) continue;
            uint32_t last_pass, last_lane, last_reg;
            team_last_row<L, R, team_granule<local_, affine_>()>(query.length, longer, last_pass, last_lane, last_reg);
            if (local_ ? pass + 1 != passes : pass != last_pass) continue;
            for (uint32_t team = 0; team < teams; ++team) {
                if (local_) {
                    uint32_t best = 0;
                    for (uint32_t lane = 0; lane < (uint32_t)L; ++lane)
                        for (uint32_t b : lanes[(size_t)team * L + lane].best) best = std::max(best, half ? (uint32_t)high_of(b) : (uint32_t)low_of(b));
                    out[team] = k.truth(best);
                }
                else {
                    uint32_t const cell = lanes[(size_t)team * L + last_lane].rows.h[last_reg];
                    out[team] = k.truth(half ? (uint32_t)high_of(cell) : (uint32_t)low_of(cell)) - (affine_ ? 0 : k.open);
                }
            }
        }
    }
}

template <bool local_, bool affine_, bool wide_, bool distance_, int L, int R>
void cross(text_t const *queries, uint32_t queries_count, text_t const *candidates, uint32_t candidates_count,
           uint8_t const *byte_to_class, int8_t const *class_costs, int open, int extend, int64_t *results, uint64_t stride) {
    // the kernel's roles: queries longest first, candidates by ascending length, 256 / L candidates per block
    std::vector<uint32_t> q_order(queries_count), c_order(candidates_count);
    for (uint32_t i = 0; i < queries_count; ++i) q_order[i] = i;
    for (uint32_t i = 0; i < candidates_count; ++i) c_order[i] = i;
    std::stable_sort(q_order.begin(), q_order.end(), [&](uint32_t a, uint32_t b) { return queries[a].length > queries[b].length; });
    std::stable_sort(c_order.begin(), c_order.end(), [&](uint32_t a, uint32_t b) { return candidates[a].length < candidates[b].length; });
    uint32_t const per_block = (L > 16 ? 512 : 256) / L; // teams of more than sixteen lanes: workgroups of 512 threads
    for (uint32_t pair = 0; pair * 2 < queries_count; ++pair) {
        uint32_t const low = q_order[2 * pair];
        bool const has_high = 2 * pair + 1 < queries_count;
        uint32_t const high = has_high ? q_order[2 * pair + 1] : low;
        for (uint32_t first = 0; first < candidates_count; first += per_block) {
            uint32_t const count = std::min(per_block, candidates_count - first);
            std::vector<text_t> block(count);
            for (uint32_t i = 0; i < count; ++i) block[i] = candidates[c_order[first + i]];
            std::vector<int64_t> out_low(count), out_high(count);
            score_item<local_, affine_, wide_, distance_, L, R>(queries[low], queries[high], has_high, block, byte_to_class, class_costs, open, extend,
                                              out_low.data(), out_high.data());
            for (uint32_t i = 0; i < count; ++i) {
                results[(uint64_t)low * stride + c_order[first + i]] = out_low[i];
                if (has_high) results[(uint64_t)high * stride + c_order[first + i]] = out_high[i];
            }
        }
    }
}

} // namespace

#define TEAM_SHAPES(CALL) CALL(16, 32) CALL(16, 16) CALL(16, 24) CALL(8, 32) CALL(4, 32) CALL(4, 16) CALL(4, 8) CALL(2, 16) CALL(1, 32) CALL(1, 4) CALL(64, 32) CALL(32, 16) CALL(64, 4)

/** Tapes with count + 1 64-bit offsets; results[q * stride + c].  `wide`: cells ordered as unsigned integers (two-input maxima)
 *  instead of as half-float patterns.  `local` = 2: a DISTANCE engine - the caller passes negated costs (a 256-class identity map
 *  and a table of -match / -mismatch stand in for uniform costs) and negates the scores back.  Returns 0, or -1 for a shape that
 *  is not instantiated. */
extern "C" int team_model_cross(int local, int affine, int wide, int lanes, int registers, char const *q_data, uint64_t const *q_offsets,
                                uint32_t q_count, char const *c_data, uint64_t const *c_offsets, uint32_t c_count,
                                uint8_t const *byte_to_class, int8_t const *class_costs, int open, int extend, int64_t *results,
                                uint64_t stride) {
    std::vector<text_t> queries(q_count), candidates(c_count);
    for (uint32_t i = 0; i < q_count; ++i) queries[i] = {(uint8_t const *)q_data + q_offsets[i], (uint32_t)(q_offsets[i + 1] - q_offsets[i])};
    for (uint32_t i = 0; i < c_count; ++i) candidates[i] = {(uint8_t const *)c_data + c_offsets[i], (uint32_t)(c_offsets[i + 1] - c_offsets[i])};
#define TEAM_ARGUMENTS queries.data(), q_count, candidates.data(), c_count, byte_to_class, class_costs, open, extend, results, stride
#define TEAM_ORDER(WIDE, L, R)                                                                                                   \
    {                                                                                                                            \
        if (local == 2 && affine) cross<false, true, WIDE, true, L, R>(TEAM_ARGUMENTS);                                          \
        else if (local == 2) cross<false, false, WIDE, true, L, R>(TEAM_ARGUMENTS);                                              \
        else if (local && affine) cross<true, true, WIDE, false, L, R>(TEAM_ARGUMENTS);                                          \
        else if (local) cross<true, false, WIDE, false, L, R>(TEAM_ARGUMENTS);                                                   \
        else if (affine) cross<false, true, WIDE, false, L, R>(TEAM_ARGUMENTS);                                                  \
        else cross<false, false, WIDE, false, L, R>(TEAM_ARGUMENTS);                                                             \
        return 0;                                                                                                                \
    }
#define TEAM_CALL(L, R)                                                                                                          \
    if (lanes == L && registers == R) {                                                                                          \
        if (wide) TEAM_ORDER(true, L, R) else TEAM_ORDER(false, L, R)                                                            \
    }
    TEAM_SHAPES(TEAM_CALL)
#undef TEAM_CALL
    return -1;
}
