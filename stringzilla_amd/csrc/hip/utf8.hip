/*
 *  utf8.hip - transcoding UTF-8 tapes to UTF-32 on the device, the front end of the codepoint-level Levenshtein engine.
 *
 *  Replaces, for the ROCm build, the reference's per-pair transcoding
 *      levenshtein_distance_utf8::operator()   /root/reference/include/stringzillas/similarities/serial.hpp:2825-2837
 *      decode_utf8_rune / rune-offset index    .../similarities/cuda.cuh:3162-3175,3145-3703
 *  with ONE pass per call: a cross-product scores every string against many others, so each string is decoded once into
 *  a UTF-32 scratch tape and the scoring kernels stream fixed-width runes (one aligned dword per DP column) instead of
 *  re-decoding variable-width sequences Q x C times.
 *
 *  Value contract = `sz_rune_decode_unchecked` (include/stringzilla/utf8_runes/serial.h:111-124): the sequence length
 *  comes from the lead byte alone, continuation bytes contribute their low six bits unvalidated, a stray continuation
 *  byte is a one-byte rune, nothing is ever rejected.  Bytes missing from a sequence truncated by the end of the string
 *  read as zero (the reference over-reads there).
 *
 *  Layout: string i's runes start at `rune_starts[i]` (the host passes the prefix sum of BYTE lengths: a string never
 *  has more runes than bytes, so slots cannot collide and no device-side scan is needed).
 */
#include "device_common.hpp"

namespace szs_hip {

/**
 *  One string per WAVEFRONT, 64 bytes per step, one byte per lane - and still the sequential contract, bit for bit.
 *
 *  `sz_rune_decode_unchecked` takes the length of a sequence from its lead byte alone, so which bytes ARE leads is a chain:
 *  position p is a lead iff some lead q < p has q + length(byte q) == p.  One thread per string followed that chain byte by
 *  byte - a dependent global-memory round trip per rune in round 1 (0.9 ms for config 5's 2 KB strings, and it would have
 *  been half a second for a 1 MB document).  Here the chain is the ORBIT of the chunk's first lead under
 *  `next(p) = p + length(byte p)`, and an orbit is computed by pointer doubling: after step k the marked set holds
 *  next^i(start) for every i < 2^(k+1), and next^(2^(k+1)) = next^(2^k) o next^(2^k) - six steps for 64 positions, each one
 *  LDS scatter / gather and one lane permute.  Chunks of plain ASCII (no byte >= 0x80, no sequence hanging in from the
 *  chunk before) skip all of it: every byte is a lead.  A lead's rune index is a popcount of the lead mask below it, so the
 *  runes of a chunk are written in one coalesced burst.  The next chunk's bytes are in flight while this one is decoded.
 */
constexpr int transcode_waves_k = 16; // wavefronts (strings in flight) per workgroup
constexpr u32 transcode_depth_k = 8; // 64-byte chunks of a string in flight
constexpr u32 alphabet_slots_k = SZS_ALPHABET_SLOTS, alphabet_empty_k = ~0u; // the renumbering pass's table (below): [keys][ids][control]

/** Where string i lies and where its runes go - from refs and host-made starts (the host-planned path) ... */
struct transcode_refs_t {
    szs_string_ref_t const *strings;
    u64 const *rune_starts;
    __device__ __forceinline__ void locate(u32 i, u8 const *&bytes, u32 &length, u64 &start) const {
        bytes = reinterpret_cast<u8 const *>(strings[i].address), length = strings[i].length, start = rune_starts[i];
    }
};

/**
 *  ... or straight from a TAPE the host has never read (the device-planned path): string i's runes start at
 *      side_base + align4(offset[i] - offset[0]) + 8 i
 *  - a string never has more runes than bytes and its slot is its byte span rounded up to four plus eight, so slots cannot
 *  collide, every array starts on a 16-byte boundary and owns its storage up to the next one (what the codepoint kernels'
 *  four-rune loads need), and no scan is needed.  `side_base` of the candidates is the span of the queries, worked out here
 *  from the queries' own offsets.  The starts are WRITTEN (`starts_out`) for the renumbering pass and the planner.  A string
 *  whose offsets descend, or whose slot would pass `capacity` runes (the buffer was sized by an earlier call), is skipped:
 *  the planner reports the former, `*needed` (runes) lets the host grow the buffer and come back for the latter.
 */
struct transcode_tape_t {
    u8 const *data;
    void const *offsets, *before_offsets; // `before`: the tape whose runes come first in the buffer (NULL: none)
    u32 count, wide, before_count, before_wide;
    u64 capacity;
    u64 *starts_out, *needed;
    __device__ __forceinline__ static u64 at(void const *offsets, u32 wide, u64 index) {
        return wide ? static_cast<u64 const *>(offsets)[index] : (u64) static_cast<u32 const *>(offsets)[index];
    }
    __device__ __forceinline__ static u64 span(void const *offsets, u32 wide, u32 count) {
        u64 const first = at(offsets, wide, 0), last = at(offsets, wide, count);
        return last >= first ? ((last - first + 3) & ~(u64)3) + 8ull * count + 4 : 0; // runes
    }
    __device__ __forceinline__ void locate(u32 i, u8 const *&bytes, u32 &length, u64 &start) const {
        u64 const first = at(offsets, wide, 0), from = at(offsets, wide, i), to = at(offsets, wide, (u64)i + 1);
        u64 const side_base = before_offsets ? span(before_offsets, before_wide, before_count) : 0;
        bool const sound = from >= first && to >= from && to - from <= 0xFFFFFFFFull;
        start = side_base + (sound ? ((from - first + 3) & ~(u64)3) + 8ull * i : 0);
        length = sound ? (u32)(to - from) : 0u;
        if (start + (((u64)length + 3) & ~(u64)3) > capacity) length = 0; // the host finds `needed` above `capacity` and returns
        bytes = data + from;
        starts_out[i] = start;
        if (i == 0 && needed) *needed = side_base + span(offsets, wide, count);
    }
};

/** Both tapes of a call in ONE launch (round 3: every launch ahead of the planner is ~5 us of a short call): strings
 *  0 ... first.count - 1 are the first tape's, the rest the second's; the counts land in one array, side by side. */
struct transcode_tapes_t {
    transcode_tape_t first, second;
    __device__ __forceinline__ void locate(u32 i, u8 const *&bytes, u32 &length, u64 &start) const {
        if (i < first.count) first.locate(i, bytes, length, start);
        else second.locate(i - first.count, bytes, length, start);
    }
};

template <typename source_t>
__global__ __launch_bounds__(64 * transcode_waves_k) void utf8_transcode_kernel(source_t source, u32 count, u32 *__restrict__ runes,
                                                                                u32 *__restrict__ rune_counts,
                                                                                u32 *__restrict__ any_multibyte,
                                                                                u32 *__restrict__ alphabet_workspace) {
    // ---- the renumbering pass that follows wants an empty table: emptied here, by everybody, instead of by two fills of the
    //      stream ahead of it (each an operation of its own in front of the planner, ~5 us of a short call)
    if (alphabet_workspace) {
        for (u32 slot = blockIdx.x * blockDim.x + threadIdx.x; slot < alphabet_slots_k; slot += gridDim.x * blockDim.x)
            alphabet_workspace[slot] = alphabet_empty_k;
        if (blockIdx.x == 0 && threadIdx.x < 2) alphabet_workspace[2 * alphabet_slots_k + threadIdx.x] = 0; // `control`
    }
    __shared__ u8 reached[transcode_waves_k][64];
    u32 const lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u8 volatile *const mine = reached[wave];
    u64 const below = (1ull << lane) - 1; // lanes before this one
    for (u32 i = blockIdx.x * transcode_waves_k + wave; i < count; i += gridDim.x * transcode_waves_k) {
        u8 const *bytes;
        u32 length;
        u64 start;
        source.locate(i, bytes, length, start);
        u32 *const out = runes + start;
        u32 produced = 0, hanging = 0; // `hanging`: bytes at the head of the chunk that belong to the previous chunk's last rune
        bool multibyte = false;
        // A string is a chain of 64-byte chunks, each waiting for its bytes: with the next chunk alone in flight a 2048-byte line
        // was 32 memory round trips one after the other (59 us for an eighth of config 5u, whatever the other 3,500 strings did).
        // Eight chunks are in flight now: a chunk's bytes are asked for eight steps before they are decoded.
        u32 window[transcode_depth_k];
#pragma unroll
        for (u32 j = 0; j < transcode_depth_k; ++j) window[j] = 64u * j + lane < length ? bytes[64u * j + lane] : 0u;
        for (u32 round_base = 0; round_base < length; round_base += 64u * transcode_depth_k)
#pragma unroll
        for (u32 j = 0; j < transcode_depth_k; ++j) {
            u32 const base = round_base + 64u * j;
            if (base >= length) break; // (wavefront-uniform)
            u32 const byte = window[j];
            u32 const refill = base + 64u * transcode_depth_k + lane;
            window[j] = refill < length ? bytes[refill] : 0u;
            u32 const after = base + 64 + lane;
            u32 const ahead = window[(j + 1) % transcode_depth_k]; // the chunk after this one (j = depth - 1: refilled at j = 0)
            bool const valid = base + lane < length;
            u64 const valid_mask = __ballot(valid);
            u32 const sequence = 1u + (byte >= 0xC0u) + (byte >= 0xE0u) + (byte >= 0xF0u);
            u64 const high_mask = __ballot(valid && byte >= 0x80u);
            multibyte |= high_mask != 0;

            // ---- well-formed text (round 4): when every byte that is not a continuation byte is followed by exactly the
            //      continuation bytes its own value announces - and the chunk's head by exactly the ones the previous chunk left
            //      hanging - the chain of jumps below visits the non-continuation bytes and nothing else: two ballots instead
            //      of six rounds through LDS.  (A 2048-byte line of prose was 32 chunks x 6 rounds, one after another: the
            //      transcoding of an eighth of config 5u took 73 us, all of it that one chain.)  Anything else - stray
            //      continuation bytes, tails cut short, a lead at the end of the string - takes the chain, as before.
            u64 const continuing = __ballot(valid && (byte & 0xC0u) == 0x80u);
            u64 const continuing_ahead = __ballot(after < length && (ahead & 0xC0u) == 0x80u);
            u32 const following = (u32)(((continuing >> lane) >> 1) | (continuing_ahead << (63u - lane))) & 0xFu; // bytes lane + 1 ... + 4
            bool const announced = (following & ((1u << sequence) - 1u)) == (1u << (sequence - 1u)) - 1u;
            bool const lead = valid && (byte & 0xC0u) != 0x80u;
            bool const head_as_left = hanging < 4u && (continuing & ((2ull << hanging) - 1ull)) == (1ull << hanging) - 1ull;
            bool const well_formed = head_as_left && !__ballot(lead && !announced);

            u64 leads;
            if (!high_mask && !hanging) leads = valid_mask; // plain ASCII: every byte is a lead
            else if (well_formed) leads = valid_mask & ~continuing;
            else {
                u32 jump = valid ? lane + sequence : 64u; // next^(2^k) of this position, 64 = beyond the chunk
                jump = jump > 64u ? 64u : jump;
                u64 marked = hanging < 64u ? 1ull << hanging : 0ull;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    mine[lane] = 0;
                    __builtin_amdgcn_wave_barrier();
                    if (((marked >> lane) & 1ull) && jump < 64u) mine[jump] = 1;
                    __builtin_amdgcn_wave_barrier();
                    marked |= __ballot(mine[lane] != 0);
                    __builtin_amdgcn_wave_barrier();
                    u32 const onward = (u32)__shfl((int)jump, (int)(jump & 63u), 64); // next^(2^k) of where this lane lands
                    jump = jump < 64u ? onward : 64u;
                }
                leads = marked & valid_mask;
            }

            // ---- decode the leads: tail bytes come from the lanes above, the last three lanes' from the chunk ahead
            u32 tail[3];
#pragma unroll
            for (u32 k = 1; k < 4; ++k) {
                u32 const here = (u32)__shfl((int)byte, (int)((lane + k) & 63u), 64);
                u32 const there = (u32)__shfl((int)ahead, (int)((lane + k) & 63u), 64);
                u32 const source = lane + k < 64u ? here : there;
                tail[k - 1] = k < sequence && base + lane + k < length ? source & 0x3Fu : 0u;
            }
            u32 rune = byte;
            if (sequence == 2) rune = (byte & 0x1Fu) << 6 | tail[0];
            if (sequence == 3) rune = (byte & 0x0Fu) << 12 | tail[0] << 6 | tail[1];
            if (sequence == 4) rune = (byte & 0x07u) << 18 | tail[0] << 12 | tail[1] << 6 | tail[2];
            if ((leads >> lane) & 1ull) out[produced + (u32)__popcll(leads & below)] = rune;
            produced += (u32)__popcll(leads);

            // ---- what hangs over into the next chunk: the last lead's sequence may end beyond this one
            if (leads) {
                u32 const last = 63u - (u32)__clzll(leads);
                u32 const end = last + (u32)__shfl((int)sequence, (int)last, 64);
                hanging = end > 64u ? end - 64u : 0u;
            }
            else hanging = hanging >= 64u ? hanging - 64u : 0u; // (a chunk without leads: cannot happen, sequences are <= 4 bytes)
        }
        if (lane == 0) {
            rune_counts[i] = produced;
            // (the flag is ONE address: an atomic from every wavefront of a 6,000-string batch is 6,000 read-modify-writes in a row at
            // the memory side, ~13 ns each - the kernel lasted 83 us however short its strings were.  Once it is set, nobody writes.)
            if (multibyte && !__atomic_load_n(any_multibyte, __ATOMIC_RELAXED)) atomicOr(any_multibyte, 1u);
        }
    }
}


/* ---- a batch-wide DENSE ALPHABET ---------------------------------------------------------------------------------------------
 *
 *  The codepoint kernels key their match masks by rune; a rune has 21 bits, so every column of every pair starts with a hash
 *  probe into the query's rune table (lev_myers.hip) - a divergent loop around a dependent LDS read, which the compiler cannot
 *  hoist or overlap (config 5u: 0.62 VALU lane-operations per cell and 12x the scalar instructions of the byte kernels, which
 *  index a table with the byte).  Only EQUALITY of runes matters to the distance, so the runes of the whole batch are renumbered
 *  1 ... A once per call - a hash table in global memory, one claim per distinct rune - and the UTF-32 arrays rewritten in
 *  place.  With A <= SZS_ALPHABET_MOST the kernels then look a symbol up in a direct table (`local[id]`, one LDS read, no loop);
 *  a richer batch keeps its runes (the second pass does nothing) and the kernels keep probing.
 */
constexpr u32 alphabet_seen_lines_k = 4096, alphabet_batch_k = 8; // ... and the 64-rune chunks of a string a wavefront asks for at once

__device__ __forceinline__ u32 alphabet_slot(u32 rune) { return (rune * 2654435761u) >> (32 - __builtin_ctz(alphabet_slots_k)); }

/** control[0]: distinct runes claimed so far (the alphabet's size), control[1]: 1 when the table ran out of room.
 *
 *  Text repeats its runes, and a claim is an atomic on a table in global memory: 8192 wavefronts that all meet ' ' in their
 *  first sixty-four runes are 100,000 compare-and-swaps on ONE address, serialised at the memory side - 0.62 ms for 4096 + 4096
 *  lines of prose (profiles/r03: the claim kernel was longer than any scoring launch of that call).  So a workgroup claims in
 *  LDS first: a direct-mapped table of the runes somebody in THIS workgroup has taken charge of; only the lane that installs a
 *  rune there (or finds its line taken by another rune) goes to the global table.  Few, large, persistent workgroups - one per
 *  CU at most, sixteen wavefronts each - so that "once per workgroup" is a few hundred global claims per distinct rune, not
 *  thousands. */
constexpr u32 alphabet_claim_threads_k = 1024;

__global__ __launch_bounds__(alphabet_claim_threads_k) void alphabet_claim_kernel(u32 count, u64 const *__restrict__ rune_starts,
                                                                                  u32 const *__restrict__ rune_counts, u32 const *__restrict__ runes,
                                                                                  u32 const *__restrict__ any_multibyte, u32 *__restrict__ keys,
                                                                                  u32 *__restrict__ ids, u32 *__restrict__ control, u32 most) {
    if (!*any_multibyte) return; // an ASCII batch goes to the byte engines
    __shared__ u32 taken[alphabet_seen_lines_k];
    for (u32 line = threadIdx.x; line < alphabet_seen_lines_k; line += blockDim.x) taken[line] = alphabet_empty_k;
    __syncthreads();
    u32 const lane = threadIdx.x % 64u, waves = gridDim.x * (blockDim.x / 64u);
    for (u32 i = blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u; i < count; i += waves) {
        u32 const *const text = runes + rune_starts[i];
        u32 const length = rune_counts[i];
        // (eight 64-rune chunks asked for at once: one wavefront walks a 1,500-rune line in three memory round trips, not 24)
        for (u32 first = 0; first < length; first += 64u * alphabet_batch_k) {
            u32 batch[alphabet_batch_k];
#pragma unroll
            for (u32 b = 0; b < alphabet_batch_k; ++b) batch[b] = first + 64u * b + lane < length ? text[first + 64u * b + lane] : alphabet_empty_k;
            // (asked once per batch, with the batch's runes, not once per new rune behind its own round trip)
            bool const given_up = __atomic_load_n(&control[0], __ATOMIC_RELAXED) > most || __atomic_load_n(&control[1], __ATOMIC_RELAXED);
#pragma unroll
            for (u32 b = 0; b < alphabet_batch_k; ++b) {
                u32 const rune = batch[b]; // (a decoded sequence has 21 bits: alphabet_empty_k marks the lanes past the end)
                if (rune == alphabet_empty_k) continue;
                u32 slot = alphabet_slot(rune);
                u32 const line = slot & (alphabet_seen_lines_k - 1);
                if (taken[line] == rune || atomicCAS(&taken[line], alphabet_empty_k, rune) == rune) continue;
                // Once the alphabet has outgrown `most` (or the table) nothing will be renamed: stop claiming.  A batch of
                // high-entropy bytes decoded unchecked would otherwise fill all 65536 slots and every later rune would probe the
                // whole table (seconds on a large batch, for a result that is thrown away).  The counter only grows, and the
                // renaming pass reads the same words and skips.
                if (given_up) return;
                for (u32 probes = 0;; ++probes) {
                    u32 key = keys[slot]; // a slot is written once: a stale line can only read as empty, and then the swap decides
                    if (key == alphabet_empty_k) key = atomicCAS(&keys[slot], alphabet_empty_k, rune);
                    if (key == rune) break;
                    if (key == alphabet_empty_k) { // this thread claimed the slot: the rune's id is the next one
                        u32 const id = atomicAdd(&control[0], 1u) + 1;
                        ids[slot] = id;
                        break;
                    }
                    if (probes >= alphabet_slots_k) { control[1] = 1; break; } // table full: the batch keeps its runes
                    // (a crowded table: somebody has given up by now, or soon will - asked every sixteen probes)
                    if ((probes & 15u) == 15u && (__atomic_load_n(&control[0], __ATOMIC_RELAXED) > most || __atomic_load_n(&control[1], __ATOMIC_RELAXED))) return;
                    slot = (slot + 1) & (alphabet_slots_k - 1);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void alphabet_rename_kernel(u32 count, u64 const *__restrict__ rune_starts, u32 const *__restrict__ rune_counts,
                                                              u32 *__restrict__ runes, u32 const *__restrict__ any_multibyte,
                                                              u32 const *__restrict__ keys, u32 const *__restrict__ ids,
                                                              u32 const *__restrict__ control, u32 most, u32 *__restrict__ alphabet_out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) alphabet_out[0] = control[0], alphabet_out[1] = control[1]; // distinct runes, overflow
    if (!*any_multibyte || control[1] || control[0] > most) return;
    u32 const lane = threadIdx.x % 64u, waves = gridDim.x * (blockDim.x / 64u);
    for (u32 i = blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u; i < count; i += waves) {
        u32 *const text = runes + rune_starts[i];
        u32 const length = rune_counts[i];
        for (u32 first = 0; first < length; first += 64u * alphabet_batch_k) {
            u32 batch[alphabet_batch_k];
#pragma unroll
            for (u32 b = 0; b < alphabet_batch_k; ++b) batch[b] = first + 64u * b + lane < length ? text[first + 64u * b + lane] : alphabet_empty_k;
#pragma unroll
            for (u32 b = 0; b < alphabet_batch_k; ++b) {
                u32 const rune = batch[b];
                if (rune == alphabet_empty_k) continue;
                u32 slot = alphabet_slot(rune);
                while (keys[slot] != rune) slot = (slot + 1) & (alphabet_slots_k - 1); // every rune of the batch was claimed
                text[first + 64u * b + lane] = ids[slot];
            }
        }
    }
}


/* ---- tiny tokens of a codepoint call as strings of BYTES (round 6) ----------------------------------------------------------------
 *
 *  Words of text are ~6 runes: the general front end above (transcode, claim, rename, plan: five dependent launches and a wait,
 *  ~120 us) costs more than scoring 4096 x 4096 of them (85 us in hip/myers_tiny.hip).  That kernel keys its masks by BYTE; only
 *  equality of symbols matters; and a batch of words holds a hundred distinct runes, not a million.  So: one pass, one thread per
 *  string (a word is a handful of sequential steps - the chain of lead bytes is walked as it stands), every rune becomes a byte:
 *  ASCII itself, anything else 128 + the slot it claims in a table of 128 runes in device memory (open addressing, compare-and-
 *  swap; the table outlives the call and every workgroup works from a copy in LDS, so a stream of batches claims its runes
 *  once).  A word is walked by its own thread, a longer string by its whole wavefront (below).  The strings land where their
 *  bytes lay - string i at `side base + offset[i] - offset[0]`, runes <= bytes - so no scan is needed; what says where and how
 *  long is one word per string (kernels.h: szs_hip_utf8_narrow).
 */
constexpr u32 narrow_slots_k = SZS_NARROW_SLOTS, narrow_most_runes_k = SZS_TINY_LONGEST, narrow_most_bytes_k = 4u * SZS_TINY_LONGEST;

__global__ __launch_bounds__(64) void utf8_narrow_kernel(szs_tape_t queries, szs_tape_t candidates, u8 *__restrict__ narrow, u64 capacity,
                                                          u64 *__restrict__ entries, u32 *__restrict__ table, u64 *__restrict__ totals,
                                                          u32 *unfit, u32 unfit_sequence) {
    // The table of claimed runes LIVES ON from call to call (the host zeroes it once, and again after a batch that overflowed it): a
    // stream of batches in one language claims its hundred runes in the first call and none after.  Every workgroup starts from a copy
    // of it in LDS; only a rune that copy does not hold goes to device memory (one compare-and-swap a probe).
    __shared__ u32 keys[narrow_slots_k];
    u32 const lane = threadIdx.x & 63u;
    u64 const strings = (u64)queries.count + candidates.count, i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    auto at = [](szs_tape_t const &tape, u64 index) { return transcode_tape_t::at(tape.offsets, tape.wide, index); };
    // where the sides lie in the narrow buffer: the candidates behind the queries' span (every thread works it out: two cached loads)
    u64 const q_first = at(queries, 0), q_last = at(queries, queries.count), c_first = at(candidates, 0), c_last = at(candidates, candidates.count);
    for (u32 slot = threadIdx.x; slot < narrow_slots_k; slot += blockDim.x) keys[slot] = __hip_atomic_load(&table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads(); // (the copy's loads travel beside the offsets')
    u64 const c_base = q_last >= q_first ? ((q_last - q_first + 15) & ~(u64)15) + 16 : 0;
    if (q_last < q_first || c_last < c_first || c_base + (c_last - c_first) + 16 > capacity) { // (+ 16: the byte kernel reads whole dwords)
        if (threadIdx.x == 0) *unfit = unfit_sequence;
        if (i < strings) entries[i] = 0; // (the scoring launch is already behind this one: it finds empty strings, not what the memory held)
        return;
    }
    bool const mine = i < strings, of_candidates = mine && i >= queries.count;
    szs_tape_t const &tape = of_candidates ? candidates : queries;
    u64 const index = of_candidates ? i - queries.count : i;
    u64 address = 0, place = 0;
    u32 length = 0, produced = 0;
    bool fits = true;
    if (mine) {
        u64 const first = of_candidates ? c_first : q_first, last = of_candidates ? c_last : q_last;
        u64 const from = at(tape, index), to = at(tape, index + 1);
        bool const sound = from >= first && to >= from && to <= last && to - from <= narrow_most_bytes_k;
        place = (of_candidates ? c_base : 0) + (from - first), address = tape.base + from;
        length = sound ? (u32)(to - from) : 0u, fits = sound;
    }
    u8 *const out = narrow + place;

    /** A rune's byte: itself below 0x80, else 128 + its slot in the table (claimed now if nobody has); `fits` falls when the table is full. */
    auto id_of = [&](u32 rune, bool &fits) -> u32 {
        if (rune < 0x80u) return rune;
        u32 slot = ((rune * 2654435761u) >> 16) & (narrow_slots_k - 1), probes = 0;
#pragma unroll 1
        for (; probes < narrow_slots_k; ++probes) {
            u32 key = keys[slot];
            if (!key) { // a slot is written once, from zero: the copy can only be behind by saying "empty", and then the swap decides
                key = atomicCAS(&table[slot], 0u, rune);
                keys[slot] = key ? key : rune;
            }
            if (!key || key == rune) break;
            slot = (slot + 1) & (narrow_slots_k - 1);
        }
        if (probes == narrow_slots_k) fits = false; // more distinct runes than the table holds: not a batch for this way
        return 128u + slot;
    };
    auto rune_of = [](u32 byte, u32 sequence, u32 const (&tail)[3]) -> u32 {
        u32 rune = byte;
        if (sequence == 2) rune = (byte & 0x1Fu) << 6 | tail[0];
        if (sequence == 3) rune = (byte & 0x0Fu) << 12 | tail[0] << 6 | tail[1];
        if (sequence == 4) rune = (byte & 0x07u) << 18 | tail[0] << 12 | tail[1] << 6 | tail[2];
        return rune;
    };

    /**
     *  This thread's string, by itself: the chain of lead bytes walked as it stands - a WORD's way (one round per rune, four ASCII
     *  bytes a round; a round is ~50 dependent instructions of one wavefront, a quarter of a microsecond), and the way of the rare
     *  string that is not well-formed UTF-8.  Sixteen bytes at a time in a window of five registers: the sequence at the window's head is
     *  decoded, the window shifted down by its length (`v_alignbyte_b32` takes the count from a register).
     */
    auto walk = [&]() {
        text_stream_t const text(address, length);
        u32 raw[6];
#pragma unroll
        for (u32 d = 0; d < 6; ++d) raw[d] = text.raw(d);
        u32 position = 0; // of the window's head in the string
        produced = 0;
#pragma unroll 1
        for (u32 base = 0; base < length && fits; base += 16) {
            u32 w[5]; // bytes [base, base + 20): a lead at the chunk's end finds its tail here
#pragma unroll
            for (u32 d = 0; d < 5; ++d) w[d] = text.splice(raw[d], raw[d + 1]);
            raw[0] = raw[4], raw[1] = raw[5];
#pragma unroll
            for (u32 d = 2; d < 6; ++d) raw[d] = text.raw(base / 4 + 4 + d);
            auto shift = [&](u32 bytes) { // the window down by 0 ... 4 bytes
                if (bytes == 4) w[0] = w[1], w[1] = w[2], w[2] = w[3], w[3] = w[4], w[4] = 0;
                else {
#pragma unroll
                    for (u32 d = 0; d < 4; ++d) w[d] = __builtin_amdgcn_alignbyte(w[d + 1], w[d], bytes);
                    w[4] >>= 8 * bytes;
                }
            };
            shift(position - base); // what the last sequence of the chunk before took of this one
            u32 const end = base + 16 < length ? base + 16 : length;
#pragma unroll 1
            while (position < end && fits) {
                if (!(w[0] & 0x80808080u) && position + 4 <= end) { // four ASCII bytes at the head: four runes that are their own ids
#pragma unroll
                    for (u32 k = 0; k < 4; ++k) out[produced + k] = (u8)(w[0] >> (8 * k));
                    produced += 4, position += 4;
                    shift(4);
                    continue;
                }
                u32 const byte = w[0] & 0xFFu;
                u32 const sequence = 1u + (byte >= 0xC0u) + (byte >= 0xE0u) + (byte >= 0xF0u);
                u32 tail[3]; // (bytes missing from a sequence cut short by the end of the string read as zero: the transcoder's contract)
#pragma unroll
                for (u32 k = 1; k < 4; ++k) tail[k - 1] = k < sequence && position + k < length ? (w[0] >> (8 * k)) & 0x3Fu : 0u;
                out[produced] = (u8)id_of(rune_of(byte, sequence, tail), fits); // (produced <= position: inside the string's own bytes)
                ++produced, position += sequence;
                shift(sequence);
            }
        }
    };

    constexpr u32 alone_k = 16; // bytes of the longest string a thread walks by itself
    bool const is_long = length > alone_k;
    if (length && !is_long) walk();
    // ---- the longer strings (a few per cent of a text's tokens, and every wavefront holds one): by the WHOLE wavefront, one string
    //      after the other, sixty-four bytes a step - walked by their own threads they held the wavefront for a round per rune (a
    //      60-byte line of box-drawing characters: 40 rounds, 19 us of a pass whose other strings took 2).  In well-formed text the
    //      leads are the bytes that are not continuation bytes (hip/utf8.hip: utf8_transcode_kernel makes the same observation):
    //      every such byte followed by exactly the continuation bytes its value announces, the string's first byte among them.
    //      A string that is anything else goes back to its own thread.
    bool alone_after_all = false;
    u64 const below = (1ull << lane) - 1;
    for (u64 pending = __ballot(is_long); pending; pending &= pending - 1) {
        int const owner = (int)__builtin_ctzll(pending);
        u64 const its_address = (u64)(u32)__shfl((int)(u32)address, owner, 64) | (u64)(u32)__shfl((int)(u32)(address >> 32), owner, 64) << 32;
        u64 const its_place = (u64)(u32)__shfl((int)(u32)place, owner, 64) | (u64)(u32)__shfl((int)(u32)(place >> 32), owner, 64) << 32;
        u32 const its_length = (u32)__builtin_amdgcn_readfirstlane(__shfl((int)length, owner, 64));
        u8 const *const bytes = reinterpret_cast<u8 const *>(its_address);
        u8 *const its_out = narrow + its_place;
        u32 done = 0;
        bool well_formed = true, all_fit = true;
        u32 byte = lane < its_length ? bytes[lane] : 0u;
#pragma unroll 1
        for (u32 base = 0; base < its_length; base += 64) {
            u32 const ahead = base + 64 + lane < its_length ? bytes[base + 64 + lane] : 0u; // (a byte past the end reads as no continuation)
            bool const valid = base + lane < its_length;
            u64 const continuing = __ballot(valid && (byte & 0xC0u) == 0x80u), continuing_ahead = __ballot((ahead & 0xC0u) == 0x80u);
            u32 const sequence = 1u + (byte >= 0xC0u) + (byte >= 0xE0u) + (byte >= 0xF0u);
            u32 const following = (u32)(((continuing >> lane) >> 1) | (continuing_ahead << (63u - lane))) & 0xFu; // bytes lane + 1 ... + 4
            bool const lead = valid && (byte & 0xC0u) != 0x80u;
            bool const announced = (following & ((1u << sequence) - 1u)) == (1u << (sequence - 1u)) - 1u;
            if (__ballot(lead && !announced) || (base == 0 && (continuing & 1ull))) {
                well_formed = false;
                break;
            }
            u32 tail[3];
#pragma unroll
            for (u32 k = 1; k < 4; ++k) {
                u32 const here = (u32)__shfl((int)byte, (int)((lane + k) & 63u), 64), there = (u32)__shfl((int)ahead, (int)((lane + k) & 63u), 64);
                tail[k - 1] = k < sequence ? (lane + k < 64u ? here : there) & 0x3Fu : 0u;
            }
            u64 const leads = __ballot(lead);
            if (lead) {
                bool lane_fits = true;
                its_out[done + (u32)__popcll(leads & below)] = (u8)id_of(rune_of(byte, sequence, tail), lane_fits);
                if (!lane_fits) all_fit = false;
            }
            done += (u32)__popcll(leads);
            byte = ahead;
        }
        bool const somebody_did_not_fit = __ballot(!all_fit) != 0;
        if ((int)lane == owner) {
            if (!well_formed) alone_after_all = true;
            else produced = done, fits = fits && !somebody_did_not_fit;
        }
    }
    if (alone_after_all) walk();
    if (mine) {
        if (!fits || produced > narrow_most_runes_k) *unfit = unfit_sequence, produced = 0;
        entries[i] = place | (u64)produced << 56;
    }
    // the sides' totals of runes (the call's cells are their product): one atomic per wavefront and side
    for (u32 side = 0; side < 2; ++side) {
        u32 sum = mine && (side != 0) == of_candidates ? produced : 0u;
#pragma unroll
        for (int offset = 32; offset; offset >>= 1) sum += (u32)__shfl_xor((int)sum, offset, 64);
        if ((threadIdx.x & 63u) == 0 && sum) atomicAdd(reinterpret_cast<unsigned long long *>(&totals[side]), (unsigned long long)sum);
    }
}

} // namespace szs_hip

extern "C" int szs_hip_utf8_transcode(szs_string_ref_t const *strings, uint32_t count, uint64_t const *rune_starts,
                                      uint32_t *runes, uint32_t *rune_counts, uint32_t *any_multibyte, void *stream) {
    using namespace szs_hip;
    if (!count) return 0;
    u32 const blocks = (count + transcode_waves_k - 1) / transcode_waves_k;
    transcode_refs_t const source = {strings, rune_starts};
    hipLaunchKernelGGL(utf8_transcode_kernel<transcode_refs_t>, dim3(blocks < 65536u ? blocks : 65536u), dim3(64 * transcode_waves_k), 0,
                       static_cast<hipStream_t>(stream), source, count, runes, rune_counts, any_multibyte, static_cast<u32 *>(nullptr));
    return (int)hipGetLastError();
}

extern "C" int szs_hip_utf8_transcode_tape(void const *data, void const *offsets, uint32_t count, int wide, void const *before_offsets,
                                           uint32_t before_count, int before_wide, uint64_t capacity, uint32_t *runes,
                                           uint64_t *rune_starts, uint32_t *rune_counts, uint32_t *any_multibyte, uint64_t *needed,
                                           void *alphabet_workspace, void *stream) {
    using namespace szs_hip;
    if (!count) return 0; /* (nothing to renumber either) */
    u32 const blocks = (count + transcode_waves_k - 1) / transcode_waves_k;
    transcode_tape_t const source = {static_cast<u8 const *>(data), offsets, before_offsets, count, (u32)(wide != 0), before_count,
                                     (u32)(before_wide != 0), capacity, rune_starts, needed};
    hipLaunchKernelGGL(utf8_transcode_kernel<transcode_tape_t>, dim3(blocks < 65536u ? blocks : 65536u), dim3(64 * transcode_waves_k), 0,
                       static_cast<hipStream_t>(stream), source, count, runes, rune_counts, any_multibyte, static_cast<u32 *>(alphabet_workspace));
    return (int)hipGetLastError();
}

extern "C" int szs_hip_utf8_transcode_tapes(void const *first_data, void const *first_offsets, uint32_t first_count, int first_wide,
                                            void const *second_data, void const *second_offsets, uint32_t second_count, int second_wide,
                                            uint64_t capacity, uint32_t *runes, uint64_t *rune_starts, uint32_t *rune_counts,
                                            uint32_t *any_multibyte, uint64_t *needed, void *alphabet_workspace, void *stream) {
    using namespace szs_hip;
    if (!second_count)
        return szs_hip_utf8_transcode_tape(first_data, first_offsets, first_count, first_wide, nullptr, 0, 0, capacity, runes, rune_starts,
                                           rune_counts, any_multibyte, needed, alphabet_workspace, stream);
    if (!first_count)
        return szs_hip_utf8_transcode_tape(second_data, second_offsets, second_count, second_wide, first_offsets, 0, first_wide, capacity, runes,
                                           rune_starts, rune_counts, any_multibyte, needed, alphabet_workspace, stream);
    u64 const count = (u64)first_count + second_count;
    if (count > 0xFFFFFFFFull) return (int)hipErrorInvalidValue;
    u64 const blocks = (count + transcode_waves_k - 1) / transcode_waves_k;
    transcode_tapes_t const source = {
        {static_cast<u8 const *>(first_data), first_offsets, nullptr, first_count, (u32)(first_wide != 0), 0u, 0u, capacity, rune_starts, nullptr},
        {static_cast<u8 const *>(second_data), second_offsets, first_offsets, second_count, (u32)(second_wide != 0), first_count,
         (u32)(first_wide != 0), capacity, rune_starts + first_count, needed}};
    hipLaunchKernelGGL(utf8_transcode_kernel<transcode_tapes_t>, dim3(blocks < 65536u ? (u32)blocks : 65536u), dim3(64 * transcode_waves_k), 0,
                       static_cast<hipStream_t>(stream), source, (u32)count, runes, rune_counts, any_multibyte, static_cast<u32 *>(alphabet_workspace));
    return (int)hipGetLastError();
}

extern "C" size_t szs_hip_alphabet_workspace_bytes(void) { return (size_t)SZS_ALPHABET_SLOTS * 2 * sizeof(uint32_t) + 2 * sizeof(uint32_t); }

extern "C" int szs_hip_alphabet_rename(uint32_t count, uint64_t const *rune_starts, uint32_t const *rune_counts, uint32_t *runes,
                                       uint32_t const *any_multibyte, void *workspace, int workspace_is_empty, uint32_t most,
                                       uint32_t *alphabet_out, void *stream) {
    using namespace szs_hip;
    if (!count) return 0;
    hipStream_t const s = static_cast<hipStream_t>(stream);
    u32 *const keys = static_cast<u32 *>(workspace), *const ids = keys + alphabet_slots_k, *const control = ids + alphabet_slots_k;
    if (!workspace_is_empty) { /* (the transcoding launch ahead of this one was not asked to empty it) */
        hipError_t error = hipMemsetAsync(keys, 0xFF, (size_t)alphabet_slots_k * sizeof(u32), s);
        if (error == hipSuccess) error = hipMemsetAsync(control, 0, 2 * sizeof(u32), s);
        if (error != hipSuccess) return (int)error;
    }
    u32 const blocks = (count + 3) / 4 < 2048u ? (count + 3) / 4 : 2048u;
    u32 const claim_waves = alphabet_claim_threads_k / 64u, claim_blocks = (count + claim_waves - 1) / claim_waves;
    hipLaunchKernelGGL(alphabet_claim_kernel, dim3(claim_blocks < 256u ? claim_blocks : 256u), dim3(alphabet_claim_threads_k), 0, s, count, rune_starts,
                       rune_counts, runes, any_multibyte, keys, ids, control, most);
    hipLaunchKernelGGL(alphabet_rename_kernel, dim3(blocks), dim3(256), 0, s, count, rune_starts, rune_counts, runes, any_multibyte, keys, ids,
                       control, most, alphabet_out);
    return (int)hipGetLastError();
}

extern "C" int szs_hip_utf8_narrow(szs_tape_t const *queries, szs_tape_t const *candidates, void *narrow, uint64_t capacity, uint64_t *entries,
                                   void *workspace, uint32_t *unfit, uint32_t unfit_sequence, void *stream) {
    using namespace szs_hip;
    u64 const strings = (u64)queries->count + candidates->count;
    if (!strings) return 0;
    u64 const blocks = (strings + 63) / 64; // (small workgroups: 8192 words are 128 of them - the pass is latency, spread it out)
    if (blocks > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    u32 *const table = static_cast<u32 *>(workspace);
    hipLaunchKernelGGL(utf8_narrow_kernel, dim3((u32)blocks), dim3(64), 0, static_cast<hipStream_t>(stream), *queries, *candidates,
                       static_cast<u8 *>(narrow), capacity, entries, table, reinterpret_cast<u64 *>(table + narrow_slots_k), unfit, unfit_sequence);
    return (int)hipGetLastError();
}

