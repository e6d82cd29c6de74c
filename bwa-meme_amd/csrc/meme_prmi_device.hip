// P-RMI training on the device (index building).
//
// The reference trains its "partial 3-layer recursive model index" with an offline Rust tool (RMI/rmi_lib/src/train/
// two_layer.rs) over the 8-byte keys of all suffixes; the aligner only reads the two parameter files
// (src/LearnedIndex_seeding.cpp:74-122, 186-210).  Our host trainer (bwa-meme_amd/host/meme_prmi.cpp) writes files of that
// layout; on 256 threads it needs 70 s for the 6.2 G suffixes of a human-sized genome.  Leaves are independent and the
// sorted keys are already in HBM (the staged entry array), so the same model -- bit for bit, tests/test_gpu_prmi.py -- is a
// handful of streaming kernels here:
//
//   k_prmi_bounds     first suffix-array slot of every leaf (leaf m serves the keys with key >> (64 - bits) == m)
//   k_prmi_partials   leaves with more than `partial_threshold` keys get round(count / 20) third-layer records
//   (rocPRIM scan)     where each leaf's third-layer records start
//   k_prmi_leaves     one lane per leaf: the leaf's line and error bounds, or its routing line into the third layer
//   k_prmi_third      one lane per third-layer record: its key range (the routing line is monotone), line and bounds
//
// Record semantics as in the host trainer: a line anchored on the neighbouring keys outside the segment, errors measured
// over the segment including runs of equal keys, +2 slack; the model is a search hint -- results never depend
// on it, only the number of windows a search needs.
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include "meme_common.h"

// The records have to equal the host trainer's bit for bit: no implicit fused multiply-adds (the host build pins the same,
// -ffp-contract=off); the one FMA of the model is the explicit fma() of the prediction, as in the aligner.
#pragma clang fp contract(off)

namespace {

__device__ inline double lin(double icpt, double slope, double x) { return fma(slope, x, icpt); }
__device__ inline i64 route_clamp(double v, double bound) {
  if (v < 0.0) return 0;
  return v > bound ? (i64)bound : (i64)v;
}

struct Rec { double icpt, slope; u64 err; };

__device__ inline void store24(uint8_t* out, i64 idx, const Rec& r) {
  u64* p = (u64*)(out + idx * 24);
  p[0] = (u64)__double_as_longlong(r.icpt);
  p[1] = (u64)__double_as_longlong(r.slope);
  p[2] = r.err;
}

__device__ inline Rec constant_record(i64 idx) {
  Rec r;
  r.icpt = (double)idx; r.slope = 0.0; r.err = ((u64)2 << 32) | 2u;
  return r;
}

// keys[s, e) non-empty (host: fit_segment)
__device__ Rec fit_segment(const SaEnt* __restrict__ ent, i64 n, i64 s, i64 e, u64 dom_lo, u64 dom_hi) {
  Rec r;
  u64 a_lo = dom_lo, a_hi = dom_hi;
  if (s > 0) { const u64 k = ent[s - 1].key; if (k >= dom_lo) a_lo = k; }
  if (e < n) { const u64 k = ent[e].key; if (k <= dom_hi) a_hi = k; }
  const double x0 = (double)a_lo, x1 = (double)a_hi, y0 = (double)s, y1 = (double)e;
  double slope = 0.0, icpt = y0;
  if (x1 > x0) {
    slope = (y1 - y0) / (x1 - x0);
    if (slope * 1.8446744073709552e19 * 1.1102230246251565e-16 > 0.25) { slope = 0.0; icpt = 0.5 * (y0 + y1); }
    else icpt = y0 - slope * x0;
  } else icpt = 0.5 * (y0 + y1);
  r.icpt = icpt; r.slope = slope;
  double lo_err = 1.0, hi_err = 1.0;
  double pred_prev = lin(icpt, slope, (double)a_lo);
  i64 i = s;
  u64 ki = ent[s].key;
  while (i < e) {
    i64 j = i;
    u64 kn = 0;
    while (j + 1 < e && (kn = ent[j + 1].key) == ki) ++j;
    const double p = lin(icpt, slope, (double)ki);
    lo_err = fmax(lo_err, p - (double)i);
    hi_err = fmax(hi_err, (double)(j + 1) - pred_prev);
    pred_prev = p;
    i = j + 1;
    ki = kn;
  }
  const double p_hi = lin(icpt, slope, (double)a_hi);
  lo_err = fmax(lo_err, p_hi - (double)e);
  hi_err = fmax(hi_err, (double)e - pred_prev);
  u64 lo = (u64)ceil(lo_err) + 2, hi = (u64)ceil(hi_err) + 2;
  if (lo > 0x3fffffffu) lo = 0x3fffffffu;
  if (hi > 0x7fffffffu) hi = 0x7fffffffu;
  r.err = (lo << 32) | hi;
  return r;
}

__global__ void __launch_bounds__(256) k_prmi_bounds(const SaEnt* __restrict__ ent, i64 n, int shift, i64 nleaf, i64* __restrict__ start) {
  for (i64 m = (i64)blockIdx.x * blockDim.x + threadIdx.x; m <= nleaf; m += (i64)gridDim.x * blockDim.x) {
    if (m == nleaf) { start[m] = n; continue; }
    const u64 want = shift >= 64 ? 0 : ((u64)m << shift);
    i64 lo = 0, hi = n;                                         // first slot whose key >= want
    while (lo < hi) {
      const i64 mid = lo + ((hi - lo) >> 1);
      if (ent[mid].key < want) lo = mid + 1; else hi = mid;
    }
    start[m] = lo;
  }
}

__global__ void __launch_bounds__(256) k_prmi_partials(const i64* __restrict__ start, i64 nleaf, int threshold, i64* __restrict__ np) {
  for (i64 m = (i64)blockIdx.x * blockDim.x + threadIdx.x; m < nleaf; m += (i64)gridDim.x * blockDim.x) {
    const i64 c = start[m + 1] - start[m];
    np[m] = c > threshold ? (i64)llround((double)c / 20.0) : 0;
  }
}

__device__ inline Rec routing_line(const SaEnt* __restrict__ ent, i64 s, i64 e, i64 P, u64 dom_lo, u64 dom_hi) {
  Rec leaf;
  const double x0 = (double)ent[s].key, x1 = (double)ent[e - 1].key;
  if (x1 > x0) {
    leaf.slope = (double)P / (x1 - x0) * (1.0 - 1e-9);
    if (leaf.slope * 1.8446744073709552e19 * 1.1102230246251565e-16 > 0.25) {
      leaf.slope = (double)P / ((double)dom_hi - (double)dom_lo);
      leaf.icpt = -leaf.slope * (double)dom_lo;
    } else leaf.icpt = -leaf.slope * x0;
  } else { leaf.slope = 0.0; leaf.icpt = 0.0; }
  return leaf;
}

__global__ void __launch_bounds__(256) k_prmi_leaves(const SaEnt* __restrict__ ent, i64 n, int shift, i64 nleaf,
                                                     const i64* __restrict__ start, const i64* __restrict__ np,
                                                     const i64* __restrict__ pstart, uint8_t* __restrict__ l2) {
  for (i64 m = (i64)blockIdx.x * blockDim.x + threadIdx.x; m < nleaf; m += (i64)gridDim.x * blockDim.x) {
    const i64 s = start[m], e = start[m + 1];
    const u64 dom_lo = shift >= 64 ? 0 : ((u64)m << shift);
    const u64 dom_hi = shift >= 64 ? ~(u64)0 : dom_lo + (((u64)1 << shift) - 1);
    Rec r;
    if (e == s) r = constant_record(s);
    else if (np[m] == 0) r = fit_segment(ent, n, s, e, dom_lo, dom_hi);
    else {
      r = routing_line(ent, s, e, np[m], dom_lo, dom_hi);
      r.err = ((u64)1 << 63) | ((u64)pstart[m] << 32) | (u64)np[m];
    }
    store24(l2, m, r);
  }
}

// One lane per third-layer record g: its leaf is the last one whose first record is <= g (leaves without records have
// pstart[m] == pstart[m+1] and are skipped by taking the LAST such leaf that owns records).
__global__ void __launch_bounds__(256) k_prmi_third(const SaEnt* __restrict__ ent, i64 n, int shift, i64 nleaf,
                                                    const i64* __restrict__ start, const i64* __restrict__ np,
                                                    const i64* __restrict__ pstart, i64 total, uint8_t* __restrict__ l1) {
  for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (i64)gridDim.x * blockDim.x) {
    i64 lo = 0, hi = nleaf;                                     // first leaf with pstart[m + 1] > g
    while (lo < hi) {
      const i64 mid = lo + ((hi - lo) >> 1);
      if (pstart[mid + 1] <= g) lo = mid + 1; else hi = mid;
    }
    const i64 m = lo;
    const i64 P = np[m], j = g - pstart[m];
    const i64 s = start[m], e = start[m + 1];
    const u64 dom_lo = shift >= 64 ? 0 : ((u64)m << shift);
    const u64 dom_hi = shift >= 64 ? ~(u64)0 : dom_lo + (((u64)1 << shift) - 1);
    const Rec leaf = routing_line(ent, s, e, P, dom_lo, dom_hi);
    // keys routed to record j: route(key) == j; route is monotone non-decreasing along the sorted keys
    auto first_above = [&](i64 jj) {                            // first slot in [s, e) with route(key) > jj
      i64 a = s, b = e;
      while (a < b) {
        const i64 mid = a + ((b - a) >> 1);
        if (route_clamp(lin(leaf.icpt, leaf.slope, (double)ent[mid].key), (double)P - 1.0) <= jj) a = mid + 1; else b = mid;
      }
      return a;
    };
    const i64 ps = j == 0 ? s : first_above(j - 1);
    const i64 pe = first_above(j);
    const Rec r = pe == ps ? constant_record(ps) : fit_segment(ent, n, ps, pe, dom_lo, dom_hi);
    store24(l1, g, r);
  }
}

unsigned blocks_for(i64 items) {
  i64 b = (items + 255) / 256;
  const i64 cap = 256 * 64;
  return (unsigned)(b < cap ? (b < 1 ? 1 : b) : cap);
}

}  // namespace

extern "C" int meme_prmi_train_device(meme_ctx* ctx, const void* d_sa_ent, int64_t n, int bits, int partial_threshold,
                                      void* d_l2_24, void* d_l1_24, int64_t l1_capacity, int64_t* l1_records) {
    if (!ctx || !d_sa_ent || !d_l2_24 || !l1_records || n < 1 || bits < 1 || bits > 30) { meme_set_error("meme_prmi_train_device: bad argument"); return MEME_E_ARG; }
    if (partial_threshold <= 0) partial_threshold = 1000;
    HIP_TRY(hipSetDevice(ctx->device));
    const SaEnt* ent = (const SaEnt*)d_sa_ent;
    const i64 nleaf = (i64)1 << bits;
    const int shift = 64 - bits;
    i64 *d_start = nullptr, *d_np = nullptr, *d_pstart = nullptr;
    void* d_scan = nullptr;
    auto release = [&]() { (void)hipFree(d_start); (void)hipFree(d_np); (void)hipFree(d_pstart); (void)hipFree(d_scan); };
    auto fail = [&](int code) { (void)hipStreamSynchronize(ctx->stream); release(); return code; };
    if (hipMalloc(&d_start, (size_t)(nleaf + 1) * 8) != hipSuccess || hipMalloc(&d_np, (size_t)(nleaf + 1) * 8) != hipSuccess ||
        hipMalloc(&d_pstart, (size_t)(nleaf + 1) * 8) != hipSuccess) { meme_set_error("meme_prmi_train_device: out of device memory"); return fail(MEME_E_HIP); }
    hipLaunchKernelGGL(k_prmi_bounds, dim3(blocks_for(nleaf + 1)), dim3(256), 0, ctx->stream, ent, (i64)n, shift, nleaf, d_start);
    hipLaunchKernelGGL(k_prmi_partials, dim3(blocks_for(nleaf)), dim3(256), 0, ctx->stream, d_start, nleaf, partial_threshold, d_np);
    if (hipMemsetAsync(d_np + nleaf, 0, 8, ctx->stream) != hipSuccess) return fail(MEME_E_HIP);
    size_t scan_bytes = 0;
    if (rocprim::exclusive_scan(nullptr, scan_bytes, d_np, d_pstart, 0, (size_t)(nleaf + 1), rocprim::plus<>(), ctx->stream) != hipSuccess) return fail(MEME_E_HIP);
    if (hipMalloc(&d_scan, scan_bytes ? scan_bytes : 8) != hipSuccess) return fail(MEME_E_HIP);
    if (rocprim::exclusive_scan(d_scan, scan_bytes, d_np, d_pstart, 0, (size_t)(nleaf + 1), rocprim::plus<>(), ctx->stream) != hipSuccess) return fail(MEME_E_HIP);
    i64 total = 0;
    if (hipMemcpyAsync(&total, d_pstart + nleaf, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return fail(MEME_E_HIP);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(MEME_E_HIP);
    *l1_records = total;
    if (total >= ((i64)1 << 31)) { meme_set_error("meme_prmi_train_device: %lld third-layer records do not fit the record format", (long long)total); return fail(MEME_E_ARG); }
    if (total > 0 && (!d_l1_24 || l1_capacity < total)) {
        // the caller learns the size from the first call and comes back with room for the records
        meme_set_error("meme_prmi_train_device: %lld third-layer records, room for %lld", (long long)total, (long long)l1_capacity);
        return fail(MEME_E_CAPACITY);
    }
    hipLaunchKernelGGL(k_prmi_leaves, dim3(blocks_for(nleaf)), dim3(256), 0, ctx->stream, ent, (i64)n, shift, nleaf, d_start, d_np, d_pstart, (uint8_t*)d_l2_24);
    if (total > 0)
        hipLaunchKernelGGL(k_prmi_third, dim3(blocks_for(total)), dim3(256), 0, ctx->stream, ent, (i64)n, shift, nleaf, d_start, d_np, d_pstart, total, (uint8_t*)d_l1_24);
    if (hipGetLastError() != hipSuccess) return fail(MEME_E_HIP);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { meme_set_error("meme_prmi_train_device: kernel failed"); return fail(MEME_E_HIP); }
    release();
    return MEME_OK;
}
