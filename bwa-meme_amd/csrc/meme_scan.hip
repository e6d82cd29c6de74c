// Exclusive prefix sum over 64-bit counts on the ctx's stream: per-tile sums, one block scans the tile sums, per-tile apply.
// Used wherever a stage sizes its output from per-read counts (chains, extension jobs, alignment records).
#include "meme_common.h"

namespace {

constexpr int SB = 256, SI = 8, ST = SB * SI;

__device__ __forceinline__ i64 blk_scan_excl(i64 v, i64* total) {
    __shared__ i64 wsum[SB / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    i64 x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const i64 y = __shfl_up(x, d); if (lane >= d) x += y; }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    i64 off = 0, tot = 0;
    for (int w = 0; w < SB / 64; ++w) { if (w < wid) off += wsum[w]; tot += wsum[w]; }
    __syncthreads();
    *total = tot;
    return off + x - v;
}

__global__ void __launch_bounds__(SB) k_scan_tile_sums(const i64* __restrict__ in, i64 n, i64* __restrict__ tiles) {
    const i64 base = (i64)blockIdx.x * ST + threadIdx.x * SI;
    i64 a = 0;
    for (int k = 0; k < SI; ++k) if (base + k < n) a += in[base + k];
    i64 t;
    blk_scan_excl(a, &t);
    if (threadIdx.x == 0) tiles[blockIdx.x] = t;
}
__global__ void __launch_bounds__(SB) k_scan_tile_offsets(i64* __restrict__ tiles, i64 ntiles, i64* __restrict__ total) {
    i64 carry = 0;
    for (i64 b = 0; b < ntiles; b += SB) {
        const i64 i = b + threadIdx.x;
        const i64 a = i < ntiles ? tiles[i] : 0;
        i64 t;
        const i64 e = blk_scan_excl(a, &t);
        if (i < ntiles) tiles[i] = carry + e;
        carry += t;
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(SB) k_scan_apply(const i64* __restrict__ in, i64 n, const i64* __restrict__ tiles, i64* __restrict__ out) {
    const i64 base = (i64)blockIdx.x * ST + threadIdx.x * SI;
    i64 v[SI], a = 0;
    for (int k = 0; k < SI; ++k) { v[k] = base + k < n ? in[base + k] : 0; a += v[k]; }
    i64 t;
    i64 e = blk_scan_excl(a, &t) + tiles[blockIdx.x];
    for (int k = 0; k < SI; ++k) { if (base + k < n) out[base + k] = e; e += v[k]; }
}

}  // namespace

// out[i] = in[0] + ... + in[i-1] for i in [0, n]; out has n + 1 elements (out[n] = the total, also left in *d_total when given).
// in and out may be the same array only if n + 1 elements are allocated.  Asynchronous on ctx->stream.
int meme_scan_exclusive(meme_ctx* ctx, const i64* d_in, i64* d_out, i64 n) {
    if (n < 0) return MEME_E_ARG;
    const i64 ntiles = (n + ST - 1) / ST > 0 ? (n + ST - 1) / ST : 1;
    int rc;
    if ((rc = meme_buf_reserve(ctx, ctx->scan_tmp, (size_t)(ntiles + 2) * sizeof(i64)))) return rc;
    i64* tiles = (i64*)ctx->scan_tmp.p;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)ntiles), dim3(SB), 0, ctx->stream, d_in, n, tiles);
    hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(SB), 0, ctx->stream, tiles, ntiles, d_out + n);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)ntiles), dim3(SB), 0, ctx->stream, d_in, n, (const i64*)tiles, d_out);
    HIP_TRY(hipGetLastError());
    return MEME_OK;
}
