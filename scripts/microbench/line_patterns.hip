// How should a lane (or a small group of lanes) fetch ONE random 128-byte line of 16 eight-byte keys?
// Candidate access patterns of the round-2 seeding kernel, measured as random lines per second out of a `gb`-GB
// array, `UNROLL` independent lines in flight per owner, occupancy optionally capped through dynamic LDS.
//   line_patterns <gb> [iters] [lds_kb_per_block]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

enum Pattern { LANE_8x16 = 0, COOP8_8x16 = 1, LANE_4x16_HALF = 2, PAIR_4x16 = 3, QUAD_4x8 = 4, QUAD_2x16 = 5, LANE_1x16 = 6 };

template <int P, int UNROLL>
__global__ void __launch_bounds__(256) k_lines(const ulonglong2* __restrict__ a, uint64_t n_lines, int iters, uint64_t* out) {
    extern __shared__ unsigned char dummy[];
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint64_t acc = 0;
    for (int it = 0; it < iters; it += UNROLL) {
        if constexpr (P == LANE_8x16 || P == LANE_4x16_HALF || P == LANE_1x16) {
            constexpr int NL = P == LANE_8x16 ? 8 : (P == LANE_4x16_HALF ? 4 : 1);
            ulonglong2 v[UNROLL][NL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const uint64_t r = mix(tid * 1315423911ull + (uint64_t)(it + u)) % n_lines;
#pragma unroll
                for (int e = 0; e < NL; ++e) v[u][e] = a[r * 8 + e];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                for (int e = 0; e < NL; ++e) acc ^= v[u][e].x + v[u][e].y;
        } else if constexpr (P == COOP8_8x16) {
            // every lane owns a line; instruction k fetches the lines of lanes 8k..8k+7, eight lanes per line
            ulonglong2 v[UNROLL][8];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const uint64_t r = mix(tid * 1315423911ull + (uint64_t)(it + u)) % n_lines;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint64_t rk = __shfl(r, 8 * k + (lane >> 3));
                    v[u][k] = a[rk * 8 + (lane & 7)];
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc ^= v[u][e].x + v[u][e].y;
        } else if constexpr (P == PAIR_4x16) {
            ulonglong2 v[UNROLL][4];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const uint64_t r = mix((tid >> 1) * 1315423911ull + (uint64_t)(it + u)) % n_lines;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u][e] = a[r * 8 + 2 * e + (lane & 1)];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc ^= v[u][e].x + v[u][e].y;
        } else if constexpr (P == QUAD_4x8) {
            const uint64_t* a8 = reinterpret_cast<const uint64_t*>(a);
            uint64_t v[UNROLL][4];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const uint64_t r = mix((tid >> 2) * 1315423911ull + (uint64_t)(it + u)) % n_lines;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u][e] = a8[r * 16 + 4 * e + (lane & 3)];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc ^= v[u][e];
        } else {   // QUAD_2x16
            ulonglong2 v[UNROLL][2];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const uint64_t r = mix((tid >> 2) * 1315423911ull + (uint64_t)(it + u)) % n_lines;
#pragma unroll
                for (int e = 0; e < 2; ++e) v[u][e] = a[r * 8 + 4 * e + (lane & 3)];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                for (int e = 0; e < 2; ++e) acc ^= v[u][e].x + v[u][e].y;
        }
    }
    if (acc == 0x1234567) out[0] = acc + dummy[0];
}

template <int P, int UNROLL>
void run(const char* name, int owners_per_wave, const ulonglong2* a, uint64_t bytes, int iters, uint64_t* out, int lds_kb) {
    const uint64_t n_lines = bytes / 128;
    const int blocks = 256 * 32;
    const size_t lds = (size_t)lds_kb * 1024;
    if (lds > 64 * 1024) CHECK(hipFuncSetAttribute((const void*)k_lines<P, UNROLL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_lines<P, UNROLL>), dim3(blocks), dim3(256), lds, 0, a, n_lines, UNROLL * 2, out);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_lines<P, UNROLL>), dim3(blocks), dim3(256), lds, 0, a, n_lines, iters, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double lines = (double)blocks * 4 * owners_per_wave * iters * (P == LANE_4x16_HALF ? 1.0 : 1.0);
    printf("[lines] %-34s unroll %d lds %3d KB: %6.2f G lines/s (%.2f ms)\n", name, UNROLL, lds_kb, lines / ms / 1e6, ms);
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 16;
    const int iters = argc > 2 ? atoi(argv[2]) : 64;
    const int lds_kb = argc > 3 ? atoi(argv[3]) : 0;
    const uint64_t bytes = (uint64_t)(gb * 1e9) & ~0xfffull;
    ulonglong2* a; uint64_t* out;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&out, 8));
    CHECK(hipMemset(a, 1, bytes));
    printf("[lines] array %.1f GB, %d iterations, %d KB LDS per 256-thread block\n", bytes / 1e9, iters, lds_kb);
    run<LANE_1x16, 4>("lane: 1 x 16 B (gather baseline)", 64, a, bytes, iters, out, lds_kb);
    run<LANE_8x16, 1>("lane owns line: 8 x dwordx4", 64, a, bytes, iters, out, lds_kb);
    run<LANE_8x16, 2>("lane owns line: 8 x dwordx4", 64, a, bytes, iters, out, lds_kb);
    run<COOP8_8x16, 1>("8 lanes per line, 8 instr/64 lines", 64, a, bytes, iters, out, lds_kb);
    run<COOP8_8x16, 2>("8 lanes per line, 8 instr/64 lines", 64, a, bytes, iters, out, lds_kb);
    run<LANE_4x16_HALF, 2>("lane owns half line: 4 x dwordx4", 64, a, bytes, iters, out, lds_kb);
    run<PAIR_4x16, 2>("2 lanes per line: 4 x dwordx4", 32, a, bytes, iters, out, lds_kb);
    run<PAIR_4x16, 4>("2 lanes per line: 4 x dwordx4", 32, a, bytes, iters, out, lds_kb);
    run<QUAD_4x8, 4>("4 lanes per line: 4 x dwordx2", 16, a, bytes, iters, out, lds_kb);
    run<QUAD_2x16, 4>("4 lanes per line: 2 x dwordx4", 16, a, bytes, iters, out, lds_kb);
    return 0;
}
