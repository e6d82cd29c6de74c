// Random-gather roofline of the GPU for the access pattern of the seeding kernel: independent random reads of
// `bytes` contiguous bytes (16 B per lane, `bytes/16` lanes side by side) out of a `gb`-GB array, no dependent chain,
// as many in flight as the hardware takes.  Prints requests/s and GB/s.   gather_roofline <gb> [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// LANES consecutive lanes read one contiguous record of LANES*16*E bytes (E loads of 16 B per lane, LANES*16 B apart)
template <int LANES, int E>
__global__ void __launch_bounds__(256) k_gather(const ulonglong2* __restrict__ a, uint64_t n_rec, int iters, uint64_t* out) {
    const uint64_t gid = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / LANES;
    const int t = threadIdx.x % LANES;
    uint64_t acc = 0;
    for (int it = 0; it < iters; it += 4) {
        ulonglong2 v[4][E];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t r = mix(gid * 1315423911ull + (uint64_t)(it + u)) % n_rec;
#pragma unroll
            for (int e = 0; e < E; ++e) v[u][e] = a[r * (LANES * E) + e * LANES + t];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) acc ^= v[u][e].x + v[u][e].y;
    }
    if (acc == 0x1234567) out[0] = acc;
}

template <int LANES, int E>
void run(const ulonglong2* a, uint64_t bytes, int iters, uint64_t* out) {
    const uint64_t rec = (uint64_t)LANES * E * 16, n_rec = bytes / rec;
    const int blocks = 256 * 32;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_gather<LANES, E>), dim3(blocks), dim3(256), 0, 0, a, n_rec, 8, out);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather<LANES, E>), dim3(blocks), dim3(256), 0, 0, a, n_rec, iters, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double reqs = (double)blocks * 256 / LANES * iters;
    printf("[gather] record %4llu B (%2d lanes x %d x 16 B): %.2f G records/s, %.2f TB/s requested (%.1f ms)\n",
           (unsigned long long)rec, LANES, E, reqs / ms / 1e6, reqs * rec / ms / 1e9, ms);
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 16;
    const int iters = argc > 2 ? atoi(argv[2]) : 256;
    const uint64_t bytes = (uint64_t)(gb * 1e9) & ~0xfffull;
    ulonglong2* a; uint64_t* out;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&out, 8));
    CHECK(hipMemset(a, 1, bytes));
    printf("[gather] array %.1f GB\n", bytes / 1e9);
    run<1, 1>(a, bytes, iters, out);      // 16 B
    run<4, 1>(a, bytes, iters, out);      // 64 B  (one sector; the model record / one reference word pair)
    run<4, 2>(a, bytes, iters, out);      // 128 B
    run<4, 3>(a, bytes, iters, out);      // 192 B  (the kernel's 12-slot window)
    run<8, 4>(a, bytes, iters, out);      // 512 B
    run<64, 4>(a, bytes, iters, out);     // 4 KB
    return 0;
}
