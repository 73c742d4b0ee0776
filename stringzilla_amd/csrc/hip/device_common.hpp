/*
 *  device_common.hpp - helpers shared by the gfx950 kernels: wave64 reductions and the per-lane text stream.
 *  gfx950 only: wavefront = 64 lanes, no other target is supported or dispatched.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace szs_hip {

using u8 = uint8_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i32 = int32_t;
using i64 = int64_t;

constexpr u32 wave_size_k = 64;

/* ---- host side: occupancy / attribute caches are kept PER DEVICE ORDINAL (the C-ABI allows one scope per GPU inside one
 *      process, each driven by its own host thread).  A slot holds an idempotent value, written and read with relaxed
 *      atomics: two threads that race on the first use compute the same number. */
constexpr int device_slots_k = 64;
inline int device_slot() {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) (void)hipGetLastError(), device = 0;
    return device < 0 || device >= device_slots_k ? 0 : device;
}
inline int cached(int const *slot) { return __atomic_load_n(slot, __ATOMIC_RELAXED); }
inline void remember(int *slot, int value) { __atomic_store_n(slot, value, __ATOMIC_RELAXED); }

/** Maximum of `value` over the 64 lanes of the wavefront (butterfly over ds_swizzle / DPP via __shfl_xor). */
__device__ __forceinline__ u32 wave_max_u32(u32 value) {
#pragma unroll
    for (int offset = 32; offset >= 1; offset >>= 1) {
        u32 const other = (u32)__shfl_xor((int)value, offset, 64);
        value = other > value ? other : value;
    }
    return value;
}

/**
 *  Streams the bytes of one string per lane, four at a time, out of global memory.
 *
 *  Only naturally aligned dwords that contain at least one byte of the string are ever loaded: such a dword
 *  lies in the same 4 KiB page as that byte, so the stream never touches memory the caller does not own, even
 *  though consecutive lanes read unrelated, arbitrarily aligned strings.  `v_alignbyte_b32` splices two
 *  neighbouring dwords into the next four text bytes.
 *
 *  The strings of a 64-candidate block are neighbours on the caller's tape, every lane walks its own string
 *  front to back, and all 256 candidates of a workgroup (~32 KiB at 128 B each) are re-read for each query from
 *  L2, never from HBM: the algorithmic HBM traffic is the tapes once plus the results.
 */
struct text_stream_t {
    u32 const *aligned_base; // string address rounded down to 4 bytes
    u32 byte_shift;          // string address & 3
    u32 valid_dwords;        // dwords [0, valid_dwords) contain string bytes

    __device__ __forceinline__ text_stream_t(u64 address, u32 length) {
        aligned_base = reinterpret_cast<u32 const *>(address & ~(u64)3);
        byte_shift = (u32)(address & 3);
        valid_dwords = length ? (byte_shift + length + 3) / 4 : 0;
    }

    __device__ __forceinline__ u32 raw(u32 dword_index) const {
        return dword_index < valid_dwords ? aligned_base[dword_index] : 0u;
    }

    /** Branch-free variant for hot loops: reads past the last valid dword re-read that dword (the bytes are never
     *  consumed).  Requires `valid_dwords >= 1`, i.e. a non-empty string. */
    __device__ __forceinline__ u32 raw_clamped(u32 dword_index) const {
        return aligned_base[dword_index < valid_dwords ? dword_index : valid_dwords - 1];
    }

    /** Text bytes [4k, 4k+4) given raw dwords k and k+1. */
    __device__ __forceinline__ u32 splice(u32 raw_low, u32 raw_high) const {
        return __builtin_amdgcn_alignbyte(raw_high, raw_low, byte_shift);
    }
};

} // namespace szs_hip
