// Dependent-load latency of random 8-byte reads as a function of the array size (TLB reach, page walks) and of how many
// wavefronts issue them at once:   chase_latency <mb> [waves_per_cu] [steps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__global__ void __launch_bounds__(64) k_chase(const uint64_t* __restrict__ a, uint64_t n, int steps, uint64_t* out, unsigned long long* cyc) {
    uint64_t x = mix((uint64_t)blockIdx.x * 64 + threadIdx.x + 1) % n;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < steps; ++i) x = (a[x] + mix(x + i)) % n;      // a[] holds zeros: the address depends on the loaded value
    const unsigned long long t1 = wall_clock64();
    if (x == 0x123456789) out[0] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main(int argc, char** argv) {
    const double mb = argc > 1 ? atof(argv[1]) : 1024;
    const int wpc = argc > 2 ? atoi(argv[2]) : 1;
    const int steps = argc > 3 ? atoi(argv[3]) : 2000;
    const uint64_t n = (uint64_t)(mb * 1e6 / 8);
    uint64_t* a; uint64_t* out; unsigned long long* cyc;
    CHECK(hipMalloc(&a, n * 8)); CHECK(hipMalloc(&out, 8)); CHECK(hipMalloc(&cyc, 8));
    CHECK(hipMemset(a, 0, n * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_chase, dim3(256 * wpc), dim3(64), 0, 0, a, n, 16, out, cyc);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_chase, dim3(256 * wpc), dim3(64), 0, 0, a, n, steps, out, cyc);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("[chase] %8.0f MB, %2d waves/CU (64 independent chains each): %.0f ns per dependent step (%.1f G loads/s chip-wide)\n",
           mb, wpc, ms * 1e6 / steps, 256.0 * wpc * 64 * steps / ms / 1e6);
    return 0;
}
