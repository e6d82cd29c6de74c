// What does a memory request cost on MI355X, as a function of how lanes share it and where it hits?
// Random records out of an array of `mb` MB: LPR lanes per record, E loads of B bytes per lane (contiguous record of
// LPR*E*B bytes, lane t loads bytes [(e*LPR + t)*B, +B)), 4 independent records in flight per owner.
//   request_model <mb> [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

template <int B> struct Word;
template <> struct Word<4> { typedef uint32_t T; };
template <> struct Word<8> { typedef uint64_t T; };
template <> struct Word<16> { typedef ulonglong2 T; };
__device__ __forceinline__ uint64_t fold(uint32_t v) { return v; }
__device__ __forceinline__ uint64_t fold(uint64_t v) { return v; }
__device__ __forceinline__ uint64_t fold(ulonglong2 v) { return v.x + v.y; }

template <int LPR, int E, int B>
__global__ void __launch_bounds__(256) k_req(const void* __restrict__ av, uint64_t n_rec, int iters, uint64_t* out) {
    typedef typename Word<B>::T W;
    const W* a = reinterpret_cast<const W*>(av);
    const uint64_t gid = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / LPR;
    const int t = threadIdx.x % LPR;
    uint64_t acc = 0;
    for (int it = 0; it < iters; it += 4) {
        W v[4][E];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t r = mix(gid * 1315423911ull + (uint64_t)(it + u)) % n_rec;
#pragma unroll
            for (int e = 0; e < E; ++e) v[u][e] = a[r * (LPR * E) + e * LPR + t];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) acc ^= fold(v[u][e]);
    }
    if (acc == 0x1234567) out[0] = acc;
}

template <int LPR, int E, int B>
void run(const void* a, uint64_t bytes, int iters, uint64_t* out) {
    const uint64_t rec = (uint64_t)LPR * E * B, n_rec = bytes / rec;
    const int blocks = 256 * 32;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_req<LPR, E, B>), dim3(blocks), dim3(256), 0, 0, a, n_rec, 8, out);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_req<LPR, E, B>), dim3(blocks), dim3(256), 0, 0, a, n_rec, iters, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double recs = (double)blocks * 256 / LPR * iters;
    printf("[req] %8.0f MB  record %4llu B = %2d lanes x %d x %2d B: %7.2f G records/s  %7.2f G lane-instr-requests/s  (%.2f ms)\n",
           bytes / 1e6, (unsigned long long)rec, LPR, E, B, recs / ms / 1e6, recs * E / ms / 1e6, ms);
}

int main(int argc, char** argv) {
    const double mb = argc > 1 ? atof(argv[1]) : 1024;
    const int iters = argc > 2 ? atoi(argv[2]) : 128;
    const uint64_t bytes = (uint64_t)(mb * 1e6) & ~0xfffull;
    void* a; uint64_t* out;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&out, 8));
    CHECK(hipMemset(a, 1, bytes));
    run<1, 1, 4>(a, bytes, iters, out);
    run<1, 1, 8>(a, bytes, iters, out);
    run<1, 1, 16>(a, bytes, iters, out);
    run<1, 2, 8>(a, bytes, iters, out);     // 16-B record as two dwordx2 of one lane
    run<1, 2, 16>(a, bytes, iters, out);    // 32-B record, one lane, two dwordx4
    run<2, 1, 16>(a, bytes, iters, out);    // 32-B record, two lanes
    run<4, 1, 8>(a, bytes, iters, out);     // 32-B record, four lanes
    run<4, 1, 16>(a, bytes, iters, out);    // 64 B, four lanes
    run<4, 2, 16>(a, bytes, iters, out);    // 128 B, four lanes x 2
    run<4, 4, 8>(a, bytes, iters, out);     // 128 B, four lanes x 4 dwordx2
    run<8, 1, 16>(a, bytes, iters, out);    // 128 B, eight lanes
    run<16, 1, 8>(a, bytes, iters, out);    // 128 B, sixteen lanes
    return 0;
}
