/*
 *  mirror.hip - the upper triangle of a symmetric result matrix from its lower one, on the device.
 *
 *  A symmetric (self-similarity) call scores cell (i, j <= i) once and writes both [i][j] and [j][i]
 *  (/root/reference/include/stringzillas/similarities/serial.hpp:3169-3182).  On ONE GPU the scoring kernels write the mirror
 *  cell themselves.  Over several GPUs (host/node.c, stringzilla_amd/sharded.py) every GPU scores a band of ROWS of the lower
 *  triangle - rows [a, b) against candidates [0, b) - and the cells above the diagonal that belong to other GPUs' rows are
 *  filled here, once, after every band has landed: 32 x 32 tiles through LDS, coalesced on both sides.
 */
#include "device_common.hpp"

namespace szs_hip {

__global__ __launch_bounds__(256) void mirror_lower_kernel(u64 *__restrict__ matrix, u32 side, u64 row_stride) {
    __shared__ u64 tile[32][33];
    // tile (row_tile, column_tile) of the LOWER triangle, column_tile <= row_tile: blockIdx.x enumerates them row by row
    u32 row_tile = 0, remaining = blockIdx.x;
    while (remaining > row_tile) remaining -= row_tile + 1, ++row_tile; // (a few hundred steps at most; one thread's worth of scalar work)
    u32 const column_tile = remaining;
    u32 const x = threadIdx.x & 31u, y = threadIdx.x >> 5; // 8 rows of 32 columns per pass
    for (u32 pass = 0; pass < 4; ++pass) {
        u32 const row = row_tile * 32u + y + pass * 8u, column = column_tile * 32u + x;
        if (row < side && column < side) tile[y + pass * 8u][x] = matrix[(u64)row * row_stride + column];
    }
    __syncthreads();
    for (u32 pass = 0; pass < 4; ++pass) {
        // transposed: the tile's column index becomes the row of the upper triangle
        u32 const row = column_tile * 32u + y + pass * 8u, column = row_tile * 32u + x;
        if (row < side && column < side && column > row) matrix[(u64)row * row_stride + column] = tile[x][y + pass * 8u];
    }
}

} // namespace szs_hip

extern "C" int szs_hip_mirror_lower(uint64_t *matrix, uint32_t side, uint64_t row_stride, void *stream) {
    using namespace szs_hip;
    if (side < 2) return 0;
    u64 const tiles = ((u64)side + 31) / 32;
    u64 const blocks = tiles * (tiles + 1) / 2;
    if (blocks > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(mirror_lower_kernel, dim3((u32)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), matrix, side, row_stride);
    return (int)hipGetLastError();
}
