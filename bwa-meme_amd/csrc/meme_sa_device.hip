// Suffix-array construction on the MI355X: the fwd+rc text of a GRCh38-sized genome has 6.2 G suffixes and
// the host builders (the reference's sais, our host prefix-doubling sorter) take minutes on 256 threads; on the device it is
// a radix sort by the first 32 bases (one u64 key per suffix, straight from the 2-bit text) followed by prefix doubling on
// the few groups that are still tied (repeats), HBM-bound throughout.
//
//   1. 2-bit text (k_pack_text, meme_ctx.hip), histogram of the first four bases;
//   2. the 256 four-base buckets are sorted one group of buckets at a time (bounded workspace): gather (key, position),
//      rocPRIM radix sort (rocprim::radix_sort_pairs, called directly), emit SA, group heads and ranks (rank = suffix-array position of the group's first member);
//   3. while tied groups remain: sort them by (rank of the suffix, rank of the suffix h bases further on), h = 32, 64, ...;
//      a suffix that ends before h more bases sorts first, shorter before longer (the '$' of a classic suffix array).
// Order convention = the reference's (src/Learnedindex.cpp:157-229, 242, 456-548; host/meme_sa.cpp): the text is followed by
// k = max(longest A run, longest T run) + 1 bases T and then the end sentinel; the suffix array of that padded text is built
// and the k entries that point into the padding are dropped.  The result equals the reference's `.pos_packed` order
// (checked against the host builder by tests/test_gpu_sa.py).
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/functional.hpp>

#include "meme_common.h"

namespace {

// 2-bit image of the padded text: bases of the text, then T up to N = n + k, then A (0, the smallest code) -- so the 32-base
// key of a suffix that reaches the sentinel is never larger than the key of a continuation of it; such ties are broken by the
// "ends first sorts first" rule of the refinement rounds
__global__ void __launch_bounds__(256) k_sa_pack(const uint8_t* __restrict__ text, i64 n, i64 N, u64* __restrict__ pac, i64 words) {
    for (i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (i64)gridDim.x * blockDim.x) {
        const i64 b0 = w << 5;
        u64 v = 0;
        for (int r = 0; r < 32; ++r) {
            const i64 p = b0 + r;
            const u64 c = p < n ? (u64)(text[p] & 3) : (p < N ? 3ull : 0ull);
            v = (v << 2) | c;
        }
        pac[w] = v;
    }
}

// longest run of A and of T in the text (a thread that sits on the first base of a run walks it)
__global__ void __launch_bounds__(256) k_sa_runs(const uint8_t* __restrict__ text, i64 n, unsigned long long* __restrict__ best) {
    unsigned long long loc = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const uint8_t c = text[i];
        if ((c == 0 || c == 3) && (i == 0 || text[i - 1] != c)) {
            i64 j = i + 1;
            while (j < n && text[j] == c) ++j;
            if ((unsigned long long)(j - i) > loc) loc = (unsigned long long)(j - i);
        }
    }
    if (loc) atomicMax(best, loc);
}

__device__ __forceinline__ u64 sa_key(const u64* __restrict__ pac, i64 N, i64 i) { (void)N; return extract32(pac, i); }

// suffix-array entries that point into the text (the padding's own suffixes are dropped): slot j of the padded array moves
// down by the number of padding entries before it (at most k of them, their slots sorted in `pad_slots`)
__global__ void __launch_bounds__(256) k_sa_drop_padding(const u64* __restrict__ sa_pad, i64 N, i64 n, const u64* __restrict__ pad_slots,
                                                          int n_pad, u64* __restrict__ out) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += (i64)gridDim.x * blockDim.x) {
        const u64 i = sa_pad[j];
        if ((i64)i >= n) continue;
        int lo = 0, hi = n_pad;                               // number of padding slots below j
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((i64)pad_slots[mid] < j) lo = mid + 1; else hi = mid; }
        out[j - lo] = i;
    }
}

__global__ void __launch_bounds__(256) k_sa_flag_padding(const u64* __restrict__ sa_pad, i64 N, i64 n, unsigned char* __restrict__ flag) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += (i64)gridDim.x * blockDim.x) flag[j] = (i64)sa_pad[j] >= n ? 1 : 0;
}

__global__ void __launch_bounds__(256) k_sa_hist(const u64* __restrict__ pac, i64 n, unsigned long long* __restrict__ hist) {
    __shared__ unsigned int lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x)
        atomicAdd(&lh[(unsigned)(sa_key(pac, n, i) >> 56)], 1u);
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

// (key, position) of every suffix whose first four bases fall into buckets [b0, b1); order does not matter (sorted next).
// A workgroup takes tiles of 4 096 consecutive suffixes: every lane notes which of its 16 suffixes qualify, one scan over
// the workgroup and ONE global atomic per tile reserve the output range (a reservation per wavefront round -- 16 M atomics on
// one address per pass over a 1 G-suffix text -- made this kernel 85 % of the whole construction).
constexpr int GATHER_ITEMS = 16;
__global__ void __launch_bounds__(256) k_sa_gather(const u64* __restrict__ pac, i64 n, unsigned b0, unsigned b1,
                                                    unsigned long long* __restrict__ counter, u64* __restrict__ keys,
                                                    u64* __restrict__ vals) {
    __shared__ unsigned int wsum[4];
    __shared__ unsigned long long tile_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 tile = (i64)256 * GATHER_ITEMS;
    const i64 ntiles = (n + tile - 1) / tile;
    for (i64 tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const i64 i0 = tl * tile + threadIdx.x;
        unsigned mask = 0;
#pragma unroll
        for (int r = 0; r < GATHER_ITEMS; ++r) {
            const i64 i = i0 + (i64)256 * r;
            if (i < n) {
                const unsigned b = (unsigned)(sa_key(pac, n, i) >> 56);
                if (b >= b0 && b < b1) mask |= 1u << r;
            }
        }
        const unsigned cnt = (unsigned)__popc(mask);
        unsigned incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned v = (unsigned)__shfl_up((int)incl, d); if (lane >= d) incl += v; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wave) woff += wsum[w]; total += wsum[w]; }
        if (threadIdx.x == 0) tile_base = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
        __syncthreads();
        unsigned long long p = tile_base + woff + (incl - cnt);
        for (unsigned m = mask; m; m &= m - 1) {
            const i64 i = i0 + (i64)256 * (__ffs((int)m) - 1);
            keys[p] = sa_key(pac, n, i);
            vals[p] = (u64)i;
            ++p;
        }
        __syncthreads();                                       // wsum / tile_base are reused by the next tile
    }
}

// heads of the groups of equal keys in a sorted run; v[j] = j at a head, 0 elsewhere (max-scan gives the group start)
__global__ void __launch_bounds__(256) k_sa_heads(const u64* __restrict__ k1, const u64* __restrict__ k2, i64 m,
                                                   unsigned char* __restrict__ head, i64* __restrict__ v) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (i64)gridDim.x * blockDim.x) {
        const bool h = j == 0 || k1[j] != k1[j - 1] || (k2 && k2[j] != k2[j - 1]);
        head[j] = h ? 1 : 0;
        v[j] = h ? j : 0;
    }
}

// first pass: SA positions off + j; ranks = position of the group's first member; tied = member of a group of two or more
__global__ void __launch_bounds__(256) k_sa_emit(const u64* __restrict__ vals, const unsigned char* __restrict__ head,
                                                  const i64* __restrict__ gstart, i64 m, i64 off, u64* __restrict__ sa,
                                                  u64* __restrict__ rank, unsigned char* __restrict__ tied) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (i64)gridDim.x * blockDim.x) {
        const u64 i = vals[j];
        sa[off + j] = i;
        rank[i] = (u64)(off + gstart[j]);
        tied[j] = (head[j] && (j + 1 == m || head[j + 1])) ? 0 : 1;
    }
}

// refinement round, step 1: keys of the tied suffixes -- their current rank and the rank of the suffix h bases on
__global__ void __launch_bounds__(256) k_sa_prep(const u64* __restrict__ upos, i64 m, const u64* __restrict__ sa,
                                                  const u64* __restrict__ rank, i64 n, i64 h, u64* __restrict__ prim,
                                                  u64* __restrict__ sec, u64* __restrict__ val, unsigned* __restrict__ idx) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += (i64)gridDim.x * blockDim.x) {
        const u64 i = sa[upos[k]];
        prim[k] = rank[i];
        const i64 j = (i64)i + h;
        // a suffix that ends within the next h bases sorts before every continuation, the shorter the earlier
        sec[k] = (u64)((j < n ? (i64)rank[j] : n - j - 1) + h);
        val[k] = i;
        idx[k] = (unsigned)k;
    }
}

template <class T>
__global__ void __launch_bounds__(256) k_gather_by(const T* __restrict__ in, const unsigned* __restrict__ perm, i64 m, T* __restrict__ out) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += (i64)gridDim.x * blockDim.x) out[k] = in[perm[k]];
}

// refinement round, last step: new order into the suffix array, new ranks, which suffixes are still tied
__global__ void __launch_bounds__(256) k_sa_apply(const u64* __restrict__ upos, const u64* __restrict__ val_s,
                                                   const unsigned char* __restrict__ head, const i64* __restrict__ gstart, i64 m,
                                                   u64* __restrict__ sa, u64* __restrict__ rank, unsigned char* __restrict__ tied) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += (i64)gridDim.x * blockDim.x) {
        const u64 i = val_s[k];
        sa[upos[k]] = i;
        rank[i] = upos[gstart[k]];
        tied[k] = (head[k] && (k + 1 == m || head[k + 1])) ? 0 : 1;
    }
}

__global__ void __launch_bounds__(256) k_iota_off(u64* __restrict__ out, i64 m, i64 off) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += (i64)gridDim.x * blockDim.x) out[k] = (u64)(off + k);
}

unsigned grid_for(i64 items) {
    i64 b = (items + 255) / 256;
    const i64 cap = 256 * 32;
    return (unsigned)(b < cap ? (b < 1 ? 1 : b) : cap);
}

struct Scratch {          // device allocations of one build, released together
    std::vector<void*> ptrs;
    ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T>
    int get(T** p, size_t count) {
        void* q = nullptr;
        if (hipMalloc(&q, count * sizeof(T) + 256) != hipSuccess) { meme_set_error("meme_sa_build_device: out of HBM (%.1f GB request)", count * sizeof(T) / 1e9); return MEME_E_HIP; }
        ptrs.push_back(q);
        *p = (T*)q;
        return MEME_OK;
    }
    void drop(void* p) {
        for (auto it = ptrs.begin(); it != ptrs.end(); ++it) if (*it == p) { (void)hipFree(p); ptrs.erase(it); return; }
    }
};

}  // namespace

// d_text0123: sa_num bytes (forward strand then its reverse complement, codes 0..3); d_sa: sa_num u64 out.
extern "C" int meme_sa_build_device(meme_ctx* ctx, const uint8_t* d_text0123, int64_t n, uint64_t* d_sa) {
    if (!ctx || !d_text0123 || !d_sa || n < 64) { meme_set_error("meme_sa_build_device: bad argument"); return MEME_E_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    Scratch S;
    int rc;
    unsigned long long* d_cnt = nullptr;
    if ((rc = S.get(&d_cnt, 264))) return rc;
    HIP_TRY(hipMemsetAsync(d_cnt, 0, 264 * sizeof(unsigned long long), st));
    // ---- padded text: k = longest A / T run + 1 bases T behind the text (the reference's convention), N = n + k suffixes
    unsigned long long* d_run = d_cnt + 260;
    hipLaunchKernelGGL(k_sa_runs, dim3(grid_for(n)), dim3(256), 0, st, d_text0123, (i64)n, d_run);
    unsigned long long h_run = 0;
    HIP_TRY(hipMemcpyAsync(&h_run, d_run, sizeof(h_run), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const i64 k_pad = (i64)h_run + 1, N = n + k_pad;
    const i64 words = ((N + 31) >> 5) + 2;
    u64* pac = nullptr;
    u64* rank = nullptr;
    u64* sa = nullptr;                                   // suffix array of the padded text
    if ((rc = S.get(&pac, (size_t)words))) return rc;
    if ((rc = S.get(&rank, (size_t)N))) return rc;
    if ((rc = S.get(&sa, (size_t)N))) return rc;
    hipLaunchKernelGGL(k_sa_pack, dim3(grid_for(words)), dim3(256), 0, st, d_text0123, (i64)n, N, pac, words);
    hipLaunchKernelGGL(k_sa_hist, dim3(grid_for(N)), dim3(256), 0, st, (const u64*)pac, N, d_cnt);
    unsigned long long h_hist[256];
    HIP_TRY(hipMemcpyAsync(h_hist, d_cnt, sizeof(h_hist), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // ---- groups of consecutive buckets, each at most `chunk` suffixes (a single bucket may exceed it) ----------------------
    i64 chunk = (i64)1 << 28, biggest = 0;
    std::vector<std::pair<unsigned, unsigned>> groups;
    {
        unsigned b = 0;
        while (b < 256) {
            unsigned e = b;
            i64 tot = 0;
            while (e < 256 && (e == b || tot + (i64)h_hist[e] <= chunk)) { tot += (i64)h_hist[e]; ++e; }
            if (tot > 0) groups.push_back({b, e});
            if (tot > biggest) biggest = tot;
            b = e;
        }
    }
    if (biggest >= ((i64)1 << 31)) { meme_set_error("meme_sa_build_device: %lld suffixes share their first four bases (degenerate text)", (long long)biggest); return MEME_E_ARG; }
    u64 *ka = nullptr, *kb = nullptr, *va = nullptr, *vb = nullptr;
    unsigned char *head = nullptr, *tied = nullptr;
    i64* gst = nullptr;
    i64* vv = nullptr;
    if ((rc = S.get(&ka, (size_t)biggest)) || (rc = S.get(&kb, (size_t)biggest)) || (rc = S.get(&va, (size_t)biggest)) ||
        (rc = S.get(&vb, (size_t)biggest)) || (rc = S.get(&head, (size_t)biggest)) || (rc = S.get(&tied, (size_t)biggest)) ||
        (rc = S.get(&gst, (size_t)biggest)) || (rc = S.get(&vv, (size_t)biggest))) return rc;
    size_t tmp_bytes = 0, t1 = 0, t2 = 0, t3 = 0;
    {
        rocprim::double_buffer<u64> dk(ka, kb), dv(va, vb);
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, t1, dk, dv, (i64)biggest, 0, 64, st));
        HIP_TRY(rocprim::inclusive_scan(nullptr, t2, vv, gst, (size_t)((i64)biggest), rocprim::maximum<i64>(), st));
        HIP_TRY(rocprim::select(nullptr, t3, (u64*)nullptr, tied, (u64*)nullptr, (unsigned long long*)nullptr, (i64)biggest, st));
        tmp_bytes = t1 > t2 ? t1 : t2;
        if (t3 > tmp_bytes) tmp_bytes = t3;
    }
    unsigned char* tmp = nullptr;
    if ((rc = S.get(&tmp, tmp_bytes))) return rc;
    // tied suffix-array positions, collected group by group (ascending)
    std::vector<std::pair<u64*, i64>> tied_parts;
    i64 n_tied = 0, off = 0;
    unsigned long long* d_nsel = d_cnt + 256;
    for (auto& g : groups) {
        i64 m = 0;
        for (unsigned b = g.first; b < g.second; ++b) m += (i64)h_hist[b];
        HIP_TRY(hipMemsetAsync(d_nsel, 0, sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_sa_gather, dim3(grid_for((N + GATHER_ITEMS - 1) / GATHER_ITEMS)), dim3(256), 0, st, (const u64*)pac, N, g.first, g.second, d_nsel, ka, va);
        rocprim::double_buffer<u64> dk(ka, kb), dv(va, vb);
        HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, dk, dv, m, 0, 64, st));
        hipLaunchKernelGGL(k_sa_heads, dim3(grid_for(m)), dim3(256), 0, st, (const u64*)dk.current(), (const u64*)nullptr, m, head, vv);
        HIP_TRY(rocprim::inclusive_scan(tmp, tmp_bytes, vv, gst, (size_t)(m), rocprim::maximum<i64>(), st));
        hipLaunchKernelGGL(k_sa_emit, dim3(grid_for(m)), dim3(256), 0, st, (const u64*)dv.current(), (const unsigned char*)head,
                           (const i64*)gst, m, off, sa, rank, tied);
        // positions of the tied slots of this run
        u64* iota = dk.alternate();                       // (the key buffers are free again)
        hipLaunchKernelGGL(k_iota_off, dim3(grid_for(m)), dim3(256), 0, st, iota, m, off);
        u64* sel = dv.alternate();
        HIP_TRY(rocprim::select(tmp, tmp_bytes, iota, tied, sel, d_nsel, m, st));
        unsigned long long h_sel = 0;
        HIP_TRY(hipMemcpyAsync(&h_sel, d_nsel, sizeof(h_sel), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h_sel) {
            u64* part = nullptr;
            if ((rc = S.get(&part, (size_t)h_sel))) return rc;
            HIP_TRY(hipMemcpyAsync(part, sel, (size_t)h_sel * 8, hipMemcpyDeviceToDevice, st));
            tied_parts.push_back({part, (i64)h_sel});
            n_tied += (i64)h_sel;
        }
        off += m;
    }
    HIP_TRY(hipStreamSynchronize(st));
    for (void* p : {(void*)ka, (void*)kb, (void*)va, (void*)vb, (void*)head, (void*)tied, (void*)gst, (void*)vv, (void*)tmp}) S.drop(p);
    if (off != N) { meme_set_error("meme_sa_build_device: internal count mismatch"); return MEME_E_STATE; }
    // ---- prefix doubling on the tied suffixes ---------------------------------------------------------------------------------
    if (n_tied >= ((i64)1 << 32)) { meme_set_error("meme_sa_build_device: %lld suffixes tie on 32 bases (more than 2^32): text too repetitive for this builder", (long long)n_tied); return MEME_E_ARG; }
    u64* upos = nullptr;
    if (n_tied) {
        if ((rc = S.get(&upos, (size_t)n_tied))) return rc;
        i64 p = 0;
        for (auto& t : tied_parts) { HIP_TRY(hipMemcpyAsync(upos + p, t.first, (size_t)t.second * 8, hipMemcpyDeviceToDevice, st)); p += t.second; }
        HIP_TRY(hipStreamSynchronize(st));
        for (auto& t : tied_parts) S.drop(t.first);
    }
    i64 m = n_tied;
    for (i64 h = 32; m > 0; h *= 2) {
        if (h > 4 * N) { meme_set_error("meme_sa_build_device: refinement did not converge"); return MEME_E_STATE; }
        Scratch R;
        u64 *prim = nullptr, *sec = nullptr, *val = nullptr, *k2 = nullptr, *prim_s = nullptr, *sec_s = nullptr, *val_s = nullptr, *upos2 = nullptr;
        unsigned *idx = nullptr, *idx2 = nullptr, *idx3 = nullptr;
        unsigned char *hd = nullptr, *td = nullptr;
        i64 *gs = nullptr, *v2 = nullptr;
        if ((rc = R.get(&prim, (size_t)m)) || (rc = R.get(&sec, (size_t)m)) || (rc = R.get(&val, (size_t)m)) || (rc = R.get(&k2, (size_t)m)) ||
            (rc = R.get(&prim_s, (size_t)m)) || (rc = R.get(&sec_s, (size_t)m)) || (rc = R.get(&val_s, (size_t)m)) || (rc = R.get(&upos2, (size_t)m)) ||
            (rc = R.get(&idx, (size_t)m)) || (rc = R.get(&idx2, (size_t)m)) || (rc = R.get(&idx3, (size_t)m)) || (rc = R.get(&hd, (size_t)m)) ||
            (rc = R.get(&td, (size_t)m)) || (rc = R.get(&gs, (size_t)m)) || (rc = R.get(&v2, (size_t)m))) return rc;
        size_t r1 = 0, r2 = 0, r3 = 0;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, r1, (const u64*)sec, k2, (const unsigned*)idx, idx2, m, 0, 64, st));
        HIP_TRY(rocprim::inclusive_scan(nullptr, r2, v2, gs, (size_t)(m), rocprim::maximum<i64>(), st));
        HIP_TRY(rocprim::select(nullptr, r3, upos, td, upos2, (unsigned long long*)nullptr, m, st));
        size_t rb = r1 > r2 ? r1 : r2;
        if (r3 > rb) rb = r3;
        unsigned char* rtmp = nullptr;
        if ((rc = R.get(&rtmp, rb))) return rc;
        const unsigned g = grid_for(m);
        hipLaunchKernelGGL(k_sa_prep, dim3(g), dim3(256), 0, st, (const u64*)upos, m, (const u64*)sa, (const u64*)rank, N, h, prim, sec, val, idx);
        // stable LSD: by the second rank, then by the first
        HIP_TRY(rocprim::radix_sort_pairs(rtmp, rb, (const u64*)sec, k2, (const unsigned*)idx, idx2, m, 0, 44, st));
        hipLaunchKernelGGL(k_gather_by<u64>, dim3(g), dim3(256), 0, st, (const u64*)prim, (const unsigned*)idx2, m, prim_s);
        HIP_TRY(rocprim::radix_sort_pairs(rtmp, rb, (const u64*)prim_s, k2, (const unsigned*)idx2, idx3, m, 0, 44, st));
        hipLaunchKernelGGL(k_gather_by<u64>, dim3(g), dim3(256), 0, st, (const u64*)prim, (const unsigned*)idx3, m, prim_s);
        hipLaunchKernelGGL(k_gather_by<u64>, dim3(g), dim3(256), 0, st, (const u64*)sec, (const unsigned*)idx3, m, sec_s);
        hipLaunchKernelGGL(k_gather_by<u64>, dim3(g), dim3(256), 0, st, (const u64*)val, (const unsigned*)idx3, m, val_s);
        hipLaunchKernelGGL(k_sa_heads, dim3(g), dim3(256), 0, st, (const u64*)prim_s, (const u64*)sec_s, m, hd, v2);
        HIP_TRY(rocprim::inclusive_scan(rtmp, rb, v2, gs, (size_t)(m), rocprim::maximum<i64>(), st));
        hipLaunchKernelGGL(k_sa_apply, dim3(g), dim3(256), 0, st, (const u64*)upos, (const u64*)val_s, (const unsigned char*)hd, (const i64*)gs, m,
                           sa, rank, td);
        HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), st));
        HIP_TRY(rocprim::select(rtmp, rb, upos, td, upos2, d_cnt, m, st));
        unsigned long long h_sel = 0;
        HIP_TRY(hipMemcpyAsync(&h_sel, d_cnt, sizeof(h_sel), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h_sel) HIP_TRY(hipMemcpyAsync(upos, upos2, (size_t)h_sel * 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        m = (i64)h_sel;
    }
    // ---- drop the k entries that point into the padding -------------------------------------------------------------------------
    S.drop(rank);
    {
        unsigned char* flag = nullptr;
        u64 *iota = nullptr, *pad_slots = nullptr;
        unsigned char* stmp = nullptr;
        // the padding entries sit where suffixes start with a long T run: flag + select over the whole array, in pieces
        if ((rc = S.get(&pad_slots, (size_t)k_pad + 8))) return rc;
        const i64 piece = (i64)1 << 28;
        if ((rc = S.get(&flag, (size_t)piece)) || (rc = S.get(&iota, (size_t)piece))) return rc;
        size_t sb = 0;
        HIP_TRY(rocprim::select(nullptr, sb, iota, flag, pad_slots, d_cnt, piece, st));
        if ((rc = S.get(&stmp, sb))) return rc;
        i64 found = 0;
        for (i64 o = 0; o < N; o += piece) {
            const i64 mlen = N - o < piece ? N - o : piece;
            hipLaunchKernelGGL(k_sa_flag_padding, dim3(grid_for(mlen)), dim3(256), 0, st, (const u64*)(sa + o), mlen, (i64)n, flag);
            hipLaunchKernelGGL(k_iota_off, dim3(grid_for(mlen)), dim3(256), 0, st, iota, mlen, o);
            HIP_TRY(rocprim::select(stmp, sb, iota, flag, pad_slots + found, d_cnt, mlen, st));
            unsigned long long h_sel = 0;
            HIP_TRY(hipMemcpyAsync(&h_sel, d_cnt, sizeof(h_sel), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            found += (i64)h_sel;
            if (found > k_pad) { meme_set_error("meme_sa_build_device: internal padding count mismatch"); return MEME_E_STATE; }
        }
        if (found != k_pad) { meme_set_error("meme_sa_build_device: internal padding count mismatch (%lld of %lld)", (long long)found, (long long)k_pad); return MEME_E_STATE; }
        hipLaunchKernelGGL(k_sa_drop_padding, dim3(grid_for(N)), dim3(256), 0, st, (const u64*)sa, N, (i64)n, (const u64*)pad_slots, (int)k_pad, (u64*)d_sa);
    }
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return MEME_OK;
}
