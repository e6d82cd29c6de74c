// Context management and index staging for the MI355X backend.
//
// Replaces, on the device, the index part of memoryAllocLearned() (reference src/fastmap.cpp:422-617):
// the 2-bit fwd+rc text image and the widening of the 5-byte suffix-array file into probe-ready
// entries with their 64-bit keys are produced by two streaming HIP kernels instead of an OpenMP loop.
#include <fcntl.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <functional>
#include <thread>

#include "meme_common.h"

static thread_local char g_err[512] = "";

void meme_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* meme_last_error(void) { return g_err; }

int meme_buf_reserve(meme_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return MEME_OK;
    if (b.p) { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    HIP_TRY(hipMalloc(&b.p, want));
    b.cap = want;
    return MEME_OK;
}

int meme_side_stream(meme_ctx* ctx, int i) {
    if (ctx->stream_side[i]) return MEME_OK;
    int lo = 0, hi = 0;
    if (ctx->chain_side_priority && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
        HIP_TRY(hipStreamCreateWithPriority(&ctx->stream_side[i], hipStreamNonBlocking, hi));
    else
        HIP_TRY(hipStreamCreateWithFlags(&ctx->stream_side[i], hipStreamNonBlocking));
    return MEME_OK;
}

int meme_hostbuf_reserve(meme_ctx* ctx, meme_ctx::HostBuf& b, size_t bytes) {
    if (bytes <= b.cap) return MEME_OK;
    if (b.p) { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipHostFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(hipHostMalloc(&b.p, want, hipHostMallocDefault));
    b.cap = want;
    return MEME_OK;
}

// pinned host memory for the caller's staging buffers: copies from / to it are true asynchronous DMA transfers
extern "C" void* meme_host_alloc(int64_t bytes) {
    void* p = nullptr;
    if (bytes <= 0) return nullptr;
    if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) { meme_set_error("hipHostMalloc(%lld) failed", (long long)bytes); return nullptr; }
    return p;
}
extern "C" void meme_host_free(void* p) { if (p) (void)hipHostFree(p); }

extern "C" int meme_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The chaining stage runs four kernels side by side (the lane tier and three sizes of the wavefront tier, each on a stream of its own);
// the HIP runtime multiplexes a process's streams onto 4 hardware queues per device by default, and two kernels on one queue run one
// after the other (measured: 7.9 ms per 2 M reads with 4 queues, 7.4 ms with 8).  GPU_MAX_HW_QUEUES=8 is read when the runtime initialises.
// (Round 3 set the variable from a library constructor; that changed the queue configuration of every HIP user in the host process and
// raced getenv() in other threads.  Now the binding sets it in its own start-up code, before the first HIP call, and bench.py in its
// environment; the library only documents it.)

extern "C" meme_ctx* meme_ctx_create(int device) {
    int n = meme_device_count();
    if (n <= 0) { meme_set_error("no HIP device visible: the MI355X backend has no CPU fallback"); return nullptr; }
    if (device < 0 || device >= n) { meme_set_error("device %d out of range (0..%d)", device, n - 1); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { meme_set_error("hipSetDevice(%d) failed", device); return nullptr; }
    meme_ctx* ctx = new meme_ctx();
    ctx->device = device;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->n_cus = prop.multiProcessorCount; }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        meme_set_error("hipStreamCreate failed");
        delete ctx;
        return nullptr;
    }
    for (auto& e : ctx->ev) {
        if (hipEventCreate(&e) != hipSuccess) { meme_set_error("hipEventCreate failed"); delete ctx; return nullptr; }
    }
    return ctx;
}

static void free_buf(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

extern "C" void meme_ctx_destroy(meme_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf* bufs[] = {&ctx->reads, &ctx->read_off, &ctx->slots[0], &ctx->slots[1], &ctx->slots[2], &ctx->ovf[0],
                      &ctx->ovf[1], &ctx->slot_cnt, &ctx->slot_hits, &ctx->slot_loc, &ctx->smem_off, &ctx->hit_off,
                      &ctx->smems, &ctx->hits, &ctx->scan_tmp, &ctx->counters, &ctx->pairs, &ctx->refb, &ctx->qerb,
                      &ctx->packed, &ctx->bsw_order, &ctx->bsw_ws, &ctx->pend, &ctx->blk};
    for (DevBuf* b : bufs) free_buf(*b);
    for (DevBuf& b : ctx->chain) free_buf(b);
    for (DevBuf& b : ctx->ext) free_buf(b);
    for (DevBuf& b : ctx->gcig) free_buf(b);
    for (DevBuf& b : ctx->sam) free_buf(b);
    for (DevBuf& b : ctx->kswv) free_buf(b);
    for (DevBuf& b : ctx->mate) free_buf(b);
    for (meme_ctx::HostBuf& h : ctx->h_chain) if (h.p) (void)hipHostFree(h.p);
    for (meme_ctx::HostBuf& h : ctx->h_ext) if (h.p) (void)hipHostFree(h.p);
    for (meme_ctx::HostBuf& h : ctx->h_gcig) if (h.p) (void)hipHostFree(h.p);
    for (meme_ctx::HostBuf& h : ctx->h_sam) if (h.p) (void)hipHostFree(h.p);
    for (meme_ctx::HostBuf& h : ctx->h_mate) if (h.p) (void)hipHostFree(h.p);
    if (ctx->h_kswv.p) (void)hipHostFree(ctx->h_kswv.p);
    if (ctx->owns_index) for (auto& o : ctx->owned) (void)hipFree(o.first);
    if (ctx->plcp_aux) (void)hipFree(ctx->plcp_aux);
    for (meme_ctx::HostBuf* h : {&ctx->h_smems, &ctx->h_hits, &ctx->h_smem_off, &ctx->h_hit_off, &ctx->h_misc})
        if (h->p) (void)hipHostFree(h->p);
    for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->ev_chain) if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->ev_ext) if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->ev_gcig) if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->ev_sam) if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->ev_kswv) if (e) (void)hipEventDestroy(e);
    if (ctx->ev_aux) (void)hipEventDestroy(ctx->ev_aux);
    for (auto& e : ctx->ev_emit) if (e) (void)hipEventDestroy(e);
    if (ctx->stream_emit) { (void)hipStreamSynchronize(ctx->stream_emit); (void)hipStreamDestroy(ctx->stream_emit); }
    for (auto& e : ctx->ev_side) if (e) (void)hipEventDestroy(e);
    for (auto& st : ctx->stream_side) if (st) (void)hipStreamDestroy(st);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int meme_ctx_sync(meme_ctx* ctx) {
    if (!ctx) return MEME_E_ARG;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MEME_OK;
}

extern "C" void* meme_ctx_stream(meme_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int meme_get_timings(meme_ctx* ctx, meme_timings* out) {
    if (!ctx || !out) return MEME_E_ARG;
    *out = ctx->tm;
    return MEME_OK;
}

extern "C" int meme_set_tuning(meme_ctx* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return MEME_E_ARG;
    if (!strcmp(key, "seed_blocks")) ctx->seed_blocks = value;
    else if (!strcmp(key, "smem_cap")) ctx->smem_cap = value < 8 ? 8 : value;
    else if (!strcmp(key, "seed_defer")) ctx->seed_defer = value;
    else if (!strcmp(key, "max_batch")) ctx->max_batch = value;
    else if (!strcmp(key, "sam_max_batch")) ctx->sam_max_batch = value;
    else if (!strcmp(key, "bsw_circ")) ctx->bsw_circ = value;
    else if (!strcmp(key, "ext_split")) ctx->ext_split = value;
    else if (!strcmp(key, "ext_census")) ctx->ext_census = value;
    else if (!strcmp(key, "gcig_zcap")) ctx->gcig_zcap = value;
    else if (!strcmp(key, "gcig_groups")) ctx->gcig_groups = value;
    else if (!strcmp(key, "ext_live_only")) ctx->ext_live_only = value;
    else if (!strcmp(key, "ext_rounds")) ctx->ext_rounds = value < 0 ? 0 : value;
    else if (!strcmp(key, "seed_early_tier")) ctx->seed_early_tier = value;
    else if (!strcmp(key, "bsw_blocks")) ctx->bsw_blocks = value;
    else if (!strcmp(key, "bsw_lane_min_pairs")) ctx->bsw_lane_min_pairs = value;
    else if (!strcmp(key, "chain_wave_tiers")) ctx->chain_wave_tiers = value;
    else if (!strcmp(key, "chain_side_priority")) ctx->chain_side_priority = value;
    else if (!strcmp(key, "chain_lane_hits")) ctx->chain_lane_hits = value;
    else if (!strcmp(key, "chain_light_hits")) ctx->chain_light_hits = value;
    else if (!strcmp(key, "group_lanes")) {
        if (value != 1 && value != 2 && value != 4 && value != 8 && value != 16 && value != 32) { meme_set_error("group_lanes must be 1, 2, 4, 8, 16 or 32"); return MEME_E_ARG; }
        ctx->group_lanes = value;
    } else if (!strcmp(key, "seed_blocks_per_cu")) ctx->seed_blocks_per_cu = value < 1 ? 1 : value;
    else { meme_set_error("unknown tuning key %s", key); return MEME_E_ARG; }
    return MEME_OK;
}

static unsigned stage_blocks(i64 items);
extern "C" int meme_index_share(meme_ctx* ctx, meme_ctx* owner);
// ---- staging kernels ---------------------------------------------------------------------------------
extern "C" int64_t meme_index_pac64_words(int64_t sa_num) { return ((sa_num + 31) >> 5) + 8; }
extern "C" int64_t meme_index_pos5_bytes(int64_t sa_num) { return sa_num * 5 + 16; }

// one thread per output word: 32 text bytes -> one u64, first base in the top bits; T past the end
__global__ void __launch_bounds__(256) k_pack_text(const uint8_t* __restrict__ text, i64 n, u64* __restrict__ pac, i64 words) {
  // grid-stride: a launch cannot carry more than 2^32 work-items, a GRCh38-sized index has 6.4 G slots
  for (i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (i64)gridDim.x * blockDim.x) {
    i64 base = w << 5;
    u64 v = 0;
    if (base + 32 <= n) {
        const uint4* p = reinterpret_cast<const uint4*>(text + base);   // 32-byte aligned chunks
        uint4 a = p[0], b = p[1];
        uint32_t q[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t x = q[k];   // little-endian: byte 0 is the earliest base
            v = (v << 8) | ((u64)(x & 3) << 6) | ((u64)((x >> 8) & 3) << 4) | ((u64)((x >> 16) & 3) << 2) |
                (u64)((x >> 24) & 3);
        }
    } else {
        for (int r = 0; r < 32; ++r) {
            i64 p = base + r;
            u64 c = p < n ? (u64)(text[p] & 3) : 3ull;
            v = (v << 2) | c;
        }
    }
    pac[w] = v;
  }
}

// probe-ready entries {64-bit key, text position} from the 5-byte position image (or a u64 suffix array): replaces the
// OpenMP expansion loop of src/fastmap.cpp:549-613 (13-byte entries + inverse suffix array on the host)
__global__ void __launch_bounds__(256) k_build_entries(const uint8_t* __restrict__ pos_packed, const u64* __restrict__ sa_u64,
                                                        i64 n, const u64* __restrict__ pac, SaEnt* __restrict__ ent) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    const u64 pos = sa_u64 ? sa_u64[i] : load_pos5(pos_packed, i);
    SaEnt e;
    e.key = extract32(pac, (i64)pos);   // the pad words make this "T-filled past the text end"
    e.pos = pos;
    ent[i] = e;
  }
}

// 8-byte suffix array -> the 5-byte on-disk image (hosts that hold the array as u64, e.g. bench.py's builder; the image is
// what a multi-GPU start-up broadcasts: 5 instead of 8 bytes per suffix)
__global__ void __launch_bounds__(256) k_pos5_from_sa(const u64* __restrict__ sa, i64 n, uint8_t* __restrict__ pos5) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    const u64 pos = sa[i];
    uint8_t* p = pos5 + i * 5;
    const uint32_t hi = (uint32_t)(pos >> 8);
    p[0] = (uint8_t)hi; p[1] = (uint8_t)(hi >> 8); p[2] = (uint8_t)(hi >> 16); p[3] = (uint8_t)(hi >> 24); p[4] = (uint8_t)(pos & 0xff);
  }
}

// 24-byte on-disk P-RMI records -> 32-byte records (a lookup never straddles a 128-byte line)
__global__ void __launch_bounds__(256) k_rmi32(const RmiRec* __restrict__ in, i64 n, Rmi32* __restrict__ out) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    const RmiRec r = in[i];
    Rmi32 o;
    o.icpt = r.icpt; o.slope = r.slope; o.err = r.err; o.pad = 0;
    out[i] = o;
  }
}

// Longest common prefix (capped at 255, and by the shorter suffix) of the suffixes in two adjacent suffix-array slots: the keys settle
// it below 32 bases, the 2-bit text beyond.
__device__ __forceinline__ int pair_lcp(const SaEnt a, const SaEnt b, i64 n, const u64* __restrict__ pac) {
    i64 lim = n - (i64)(a.pos > b.pos ? a.pos : b.pos);
    if (lim > 255) lim = 255;
    const u64 x = a.key ^ b.key;
    int l = x ? (__clzll((long long)x) >> 1) : 32;
    if (!x) {
        for (int m = 1; l < lim; ++m) {
            const u64 y = extract32(pac, (i64)a.pos + 32 * m) ^ extract32(pac, (i64)b.pos + 32 * m);
            if (y) { l += __clzll((long long)y) >> 1; break; }
            l += 32;
        }
    }
    return l < (int)lim ? l : (int)lim;
}

// plcp[text position of slot i] = min(255, max(lcp(slot i-1, slot i), lcp(slot i, slot i+1))): one thread per slot, the lcp with the
// next slot computed once and handed to the neighbour thread through LDS.  The byte stores scatter over the text (one-off, at staging).
__global__ void __launch_bounds__(256) k_build_plcp(const SaEnt* __restrict__ ent, i64 n, const u64* __restrict__ pac, uint8_t* __restrict__ plcp) {
    __shared__ int sl[257];
    for (i64 b0 = (i64)blockIdx.x * 256; b0 < n; b0 += (i64)gridDim.x * 256) {
        const i64 i = b0 + threadIdx.x;
        SaEnt e; e.key = 0; e.pos = 0;
        int ln = 0;
        if (i < n) {
            e = ent[i];
            if (i + 1 < n) ln = pair_lcp(e, ent[i + 1], n, pac);
        }
        sl[threadIdx.x + 1] = ln;
        if (threadIdx.x == 0) sl[0] = b0 > 0 ? pair_lcp(ent[b0 - 1], e, n, pac) : 0;
        __syncthreads();
        if (i < n) {
            const int a = sl[threadIdx.x], b = sl[threadIdx.x + 1];
            plcp[e.pos] = (uint8_t)(a > b ? a : b);
        }
        __syncthreads();
    }
}

extern "C" int meme_stage_build_plcp(meme_ctx* ctx, const void* d_sa_ent, int64_t n, const void* d_pac64, void* d_plcp) {
    if (!ctx || !d_sa_ent || !d_pac64 || !d_plcp || n <= 0) return MEME_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_build_plcp, dim3(stage_blocks(n)), dim3(256), 0, ctx->stream, (const SaEnt*)d_sa_ent, (i64)n, (const u64*)d_pac64,
                       (uint8_t*)d_plcp);
    HIP_TRY(hipGetLastError());
    return MEME_OK;
}

extern "C" int meme_stage_pack_text(meme_ctx* ctx, const uint8_t* d_text, int64_t n, void* d_pac64) {
    if (!ctx || !d_text || !d_pac64 || n <= 0) return MEME_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    i64 words = meme_index_pac64_words(n);
    hipLaunchKernelGGL(k_pack_text, dim3(stage_blocks(words)), dim3(256), 0, ctx->stream, d_text, n,
                       (u64*)d_pac64, words);
    HIP_TRY(hipGetLastError());
    return MEME_OK;
}

extern "C" int meme_stage_build_entries(meme_ctx* ctx, const uint8_t* d_pos_packed, int64_t n, const void* d_pac64,
                                        void* d_sa_ent) {
    if (!ctx || !d_pos_packed || !d_pac64 || !d_sa_ent || n <= 0) return MEME_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_build_entries, dim3(stage_blocks(n)), dim3(256), 0, ctx->stream, d_pos_packed,
                       (const u64*)nullptr, (i64)n, (const u64*)d_pac64, (SaEnt*)d_sa_ent);
    HIP_TRY(hipGetLastError());
    return MEME_OK;
}

extern "C" int meme_stage_entries_from_sa(meme_ctx* ctx, const uint64_t* d_sa, int64_t n, const void* d_pac64,
                                          void* d_sa_ent) {
    if (!ctx || !d_sa || !d_pac64 || !d_sa_ent || n <= 0) return MEME_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_build_entries, dim3(stage_blocks(n)), dim3(256), 0, ctx->stream,
                       (const uint8_t*)nullptr, (const u64*)d_sa, (i64)n, (const u64*)d_pac64, (SaEnt*)d_sa_ent);
    HIP_TRY(hipGetLastError());
    return MEME_OK;
}

extern "C" int meme_stage_pos5_from_sa(meme_ctx* ctx, const uint64_t* d_sa, int64_t n, void* d_pos5) {
    if (!ctx || !d_sa || !d_pos5 || n <= 0) return MEME_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_pos5_from_sa, dim3(stage_blocks(n)), dim3(256), 0, ctx->stream, (const u64*)d_sa, (i64)n, (uint8_t*)d_pos5);
    HIP_TRY(hipGetLastError());
    return MEME_OK;
}

extern "C" int meme_stage_rmi32(meme_ctx* ctx, const void* d_rmi24, int64_t records, void* d_rmi32) {
    if (!ctx || !d_rmi32 || records < 0 || (records > 0 && !d_rmi24)) return MEME_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (records == 0) return MEME_OK;
    hipLaunchKernelGGL(k_rmi32, dim3(stage_blocks(records)), dim3(256), 0, ctx->stream, (const RmiRec*)d_rmi24, (i64)records, (Rmi32*)d_rmi32);
    HIP_TRY(hipGetLastError());
    return MEME_OK;
}

// ---- index objects ------------------------------------------------------------------------------------
static unsigned stage_blocks(i64 items) {
    i64 b = (items + 255) / 256;
    const i64 cap = 256 * 64;           // 64 workgroups per CU, grid-stride beyond that
    return (unsigned)(b < cap ? (b < 1 ? 1 : b) : cap);
}

static int set_rmi(meme_ctx* ctx, i64 l2_records, i64 l1_records) {
    if (l2_records <= 0 || (l2_records & (l2_records - 1)) != 0) {
        // learned_index_load() requires num_model to be a power of two (src/LearnedIndex_seeding.cpp:113-119)
        meme_set_error("L2 parameter table must hold a power-of-two number of 24-byte records (got %lld)",
                       (long long)l2_records);
        return MEME_E_IO;
    }
    int bits = 0;
    while (((i64)1 << bits) < l2_records) ++bits;
    ctx->idx.shift = 64 - bits;
    ctx->idx.n_l2 = l2_records;
    ctx->idx.n_l1 = l1_records;
    return MEME_OK;
}

static void drop_index(meme_ctx* ctx) {
    if (ctx->owns_index) for (auto& o : ctx->owned) (void)hipFree(o.first);
    ctx->owned.clear();
    ctx->owns_index = false;
    if (ctx->plcp_aux) { (void)hipFree(ctx->plcp_aux); ctx->plcp_aux = nullptr; }
    ctx->idx = DevIndex();
}

static int own_alloc(meme_ctx* ctx, void** p, size_t bytes) {
    HIP_TRY(hipMalloc(p, bytes));
    ctx->owned.push_back({*p, bytes});
    ctx->owns_index = true;
    return MEME_OK;
}

// One loader for both sources.  `fetch(which, d_dst, bytes)` brings input `which` (0 text bytes, 1 position image,
// 2 second-layer records, 3 partial-layer records) to the device buffer; the staging kernels then run on ctx->stream.
typedef std::function<int(int, void*, size_t)> index_fetch_fn;
// Optional second source interface: input `which` arrives in pieces of whole `unit`-byte records; `consume(d_piece, first, count, st)`
// is called per piece on the stream that carried it (the piece's device buffer may be reused once that stream has passed it).
typedef std::function<void(const void*, i64, i64, hipStream_t)> index_piece_fn;
typedef std::function<int(int, int, const index_piece_fn&)> index_stream_fn;
static int index_build_from(meme_ctx* ctx, int64_t n, int64_t l1_bytes, int64_t l2_bytes, const index_fetch_fn& fetch,
                            const index_stream_fn& stream = index_stream_fn()) {
    if (n < 64 || l2_bytes < 24 || l2_bytes % 24 || l1_bytes % 24) {
        meme_set_error("index: bad sizes (sa_num must be >= 64, parameter files multiples of 24 B)");
        return MEME_E_ARG;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    drop_index(ctx);
    int rc = set_rmi(ctx, l2_bytes / 24, l1_bytes / 24);
    if (rc) return rc;
    const bool trace = getenv("MEME_LOAD_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[meme] index build: %s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    const i64 words = meme_index_pac64_words(n);
    const i64 n_l2 = l2_bytes / 24, n_l1 = l1_bytes / 24;
    void *d_ent = nullptr, *d_pac = nullptr, *d_l2 = nullptr, *d_l1 = nullptr, *d_tmp = nullptr;
    if ((rc = own_alloc(ctx, &d_ent, (size_t)n * sizeof(SaEnt)))) return rc;
    if ((rc = own_alloc(ctx, &d_pac, (size_t)words * 8))) return rc;
    if ((rc = own_alloc(ctx, &d_l2, (size_t)n_l2 * 32))) return rc;
    if ((rc = own_alloc(ctx, &d_l1, (size_t)(n_l1 > 0 ? n_l1 : 1) * 32))) return rc;
    // staging buffer: the 5-byte position image (the largest input), reused for the text bytes and the parameter records
    // (a streaming source needs it for the text bytes only: the 31 GB position image and the 24-byte records pass through small rings)
    size_t tmp_bytes = stream ? (size_t)n + 64 : (size_t)meme_index_pos5_bytes(n);
    if (!stream && (size_t)l2_bytes > tmp_bytes) tmp_bytes = (size_t)l2_bytes;
    if (!stream && (size_t)l1_bytes > tmp_bytes) tmp_bytes = (size_t)l1_bytes;
    HIP_TRY(hipMalloc(&d_tmp, tmp_bytes));
    lap("device allocations");
    auto fail = [&](int code) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(d_tmp); return code; };
    if ((rc = fetch(0, d_tmp, (size_t)n))) return fail(rc);
    if ((rc = meme_stage_pack_text(ctx, (const uint8_t*)d_tmp, n, d_pac))) return fail(rc);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(MEME_E_HIP);
    lap("text + pack kernel");
    if (stream) {
        // every piece of the position image becomes entries as soon as it is on the device; the same for the model records
        if ((rc = stream(1, 5, [&](const void* d_piece, i64 first, i64 count, hipStream_t st) {
                hipLaunchKernelGGL(k_build_entries, dim3(stage_blocks(count)), dim3(256), 0, st, (const uint8_t*)d_piece, (const u64*)nullptr, count,
                                   (const u64*)d_pac, (SaEnt*)d_ent + first);
            }))) return fail(rc);
        lap("position image -> entries (streamed)");
        if ((rc = stream(2, 24, [&](const void* d_piece, i64 first, i64 count, hipStream_t st) {
                hipLaunchKernelGGL(k_rmi32, dim3(stage_blocks(count)), dim3(256), 0, st, (const RmiRec*)d_piece, count, (Rmi32*)d_l2 + first);
            }))) return fail(rc);
        if (n_l1 > 0 && (rc = stream(3, 24, [&](const void* d_piece, i64 first, i64 count, hipStream_t st) {
                hipLaunchKernelGGL(k_rmi32, dim3(stage_blocks(count)), dim3(256), 0, st, (const RmiRec*)d_piece, count, (Rmi32*)d_l1 + first);
            }))) return fail(rc);
        if (hipGetLastError() != hipSuccess) return fail(MEME_E_HIP);
    } else {
        if ((rc = fetch(1, d_tmp, (size_t)n * 5))) return fail(rc);
        if (hipMemsetAsync((uint8_t*)d_tmp + (size_t)n * 5, 0, 16, ctx->stream) != hipSuccess) return fail(MEME_E_HIP);
        if ((rc = meme_stage_build_entries(ctx, (const uint8_t*)d_tmp, n, d_pac, d_ent))) return fail(rc);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(MEME_E_HIP);
        lap("position image + entry kernel");
        if ((rc = fetch(2, d_tmp, (size_t)l2_bytes))) return fail(rc);
        if ((rc = meme_stage_rmi32(ctx, d_tmp, n_l2, d_l2))) return fail(rc);
        if (n_l1 > 0) {
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(MEME_E_HIP);
            if ((rc = fetch(3, d_tmp, (size_t)l1_bytes))) return fail(rc);
            if ((rc = meme_stage_rmi32(ctx, d_tmp, n_l1, d_l1))) return fail(rc);
        }
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    lap("model records");
    HIP_TRY(hipFree(d_tmp));
    lap("free of the staging buffer");
    void* d_plcp = nullptr;
    if ((rc = own_alloc(ctx, &d_plcp, (size_t)n + 256))) return rc;
    if ((rc = meme_stage_build_plcp(ctx, d_ent, n, d_pac, d_plcp))) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    lap("plcp table");
    ctx->idx.plcp = (const uint8_t*)d_plcp;
    ctx->idx.n = n;
    ctx->idx.sa = (const SaEnt*)d_ent;
    ctx->idx.pac = (const u64*)d_pac;
    ctx->idx.l2 = (const Rmi32*)d_l2;
    ctx->idx.l1 = (const Rmi32*)d_l1;
    return MEME_OK;
}

extern "C" int meme_index_load_host(meme_ctx* ctx, const uint8_t* pos_packed, int64_t n, const uint8_t* text,
                                    const void* l1, int64_t l1_bytes, const void* l2, int64_t l2_bytes) {
    if (!ctx || !pos_packed || !text || !l2 || (l1_bytes > 0 && !l1)) { meme_set_error("meme_index_load_host: null argument"); return MEME_E_ARG; }
    const void* src[4] = {text, pos_packed, l2, l1};
    return index_build_from(ctx, n, l1_bytes, l2_bytes, [&](int which, void* d_dst, size_t bytes) {
        if (hipMemcpyAsync(d_dst, src[which], bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return (int)MEME_E_HIP;
        return hipStreamSynchronize(ctx->stream) == hipSuccess ? (int)MEME_OK : (int)MEME_E_HIP;
    });
}

// A file straight to device memory: reader threads, each with two pinned pieces (LoadBuffers) and a stream of its own -- piece k of the
// file is read by thread k mod T while the thread's previous piece is still on its way over PCIe.  (A 31 GB position image
// read into a zero-filled std::vector by one thread and copied from pageable memory took 20 s; this takes what the
// slower of page cache and PCIe allows.)
static long long file_size(const std::string& path) {
    struct stat st;
    return stat(path.c_str(), &st) == 0 ? (long long)st.st_size : -1;
}

// pinned pieces of the reader threads, allocated once per index load (pinning memory costs ~0.4 s per GB: a fresh set per file
// was a third of the load time)
struct LoadBuffers {
    static constexpr size_t PIECE = (size_t)16 << 20;
    int T = 0;
    std::vector<uint8_t*> buf;            // 2 per reader (pinned host)
    std::vector<uint8_t*> dbuf;           // 2 per reader (device): pieces that a kernel consumes where they land
    int init(int readers) {
        T = readers;
        buf.assign((size_t)2 * T, nullptr);
        dbuf.assign((size_t)2 * T, nullptr);
        for (auto& b : buf) if (hipHostMalloc((void**)&b, PIECE, hipHostMallocDefault) != hipSuccess) return MEME_E_HIP;
        for (auto& b : dbuf) if (hipMalloc((void**)&b, PIECE + 64) != hipSuccess) return MEME_E_HIP;
        return MEME_OK;
    }
    ~LoadBuffers() { for (auto b : buf) if (b) (void)hipHostFree(b); for (auto b : dbuf) if (b) (void)hipFree(b); }
};

// d_dst != nullptr: the file's bytes to d_dst.  consume != nullptr: pieces of whole `unit`-byte records to the reader's device ring
// slot, each handed to consume(d_piece, first_record, records, stream) on the reader's stream.
static int file_to_device(meme_ctx* ctx, LoadBuffers& LB, const std::string& path, void* d_dst, size_t bytes, int unit = 1,
                          const index_piece_fn* consume = nullptr) {
    if (bytes == 0) return MEME_OK;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) { meme_set_error("cannot open %s", path.c_str()); return MEME_E_IO; }
    const size_t piece = LoadBuffers::PIECE / (size_t)unit * (size_t)unit;
    const size_t n_pieces = (bytes + piece - 1) / piece;
    const int T = (int)(n_pieces < (size_t)LB.T ? n_pieces : (size_t)LB.T);
    std::atomic<int> err{MEME_OK};
    auto reader = [&](int t) {
        if (hipSetDevice(ctx->device) != hipSuccess) { err = MEME_E_HIP; return; }
        hipStream_t st = nullptr;
        uint8_t* buf[2] = {LB.buf[(size_t)2 * t], LB.buf[(size_t)2 * t + 1]};
        hipEvent_t ev[2] = {nullptr, nullptr};
        bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
        for (int b = 0; b < 2 && ok; ++b) ok = hipEventCreateWithFlags(&ev[b], hipEventDisableTiming) == hipSuccess;
        int turn = 0;
        for (size_t k = (size_t)t; ok && k < n_pieces && err == MEME_OK; k += (size_t)T, turn ^= 1) {
            const size_t off = k * piece, len = bytes - off < piece ? bytes - off : piece;
            if (hipEventSynchronize(ev[turn]) != hipSuccess) { ok = false; break; }       // the buffer's previous copy has left
            size_t got = 0;
            while (got < len) {
                const ssize_t r = pread(fd, buf[turn] + got, len - got, (off_t)(off + got));
                if (r <= 0) break;
                got += (size_t)r;
            }
            if (got != len) { err = MEME_E_IO; break; }
            if (consume) {
                uint8_t* slot = LB.dbuf[(size_t)2 * t + turn];
                ok = hipMemcpyAsync(slot, buf[turn], len, hipMemcpyHostToDevice, st) == hipSuccess;
                if (ok) (*consume)(slot, (i64)(off / (size_t)unit), (i64)(len / (size_t)unit), st);
                ok = ok && hipEventRecord(ev[turn], st) == hipSuccess;      // (after the kernel: it covers the device slot too)
            } else
                ok = hipMemcpyAsync((uint8_t*)d_dst + off, buf[turn], len, hipMemcpyHostToDevice, st) == hipSuccess &&
                     hipEventRecord(ev[turn], st) == hipSuccess;
        }
        if (st && hipStreamSynchronize(st) != hipSuccess) ok = false;
        for (int b = 0; b < 2; ++b) if (ev[b]) (void)hipEventDestroy(ev[b]);
        if (st) (void)hipStreamDestroy(st);
        if (!ok && err == MEME_OK) err = MEME_E_HIP;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(reader, t);
    reader(0);
    for (auto& x : th) x.join();
    close(fd);
    if (err == MEME_E_IO) meme_set_error("short read from %s", path.c_str());
    else if (err != MEME_OK) meme_set_error("HIP error while loading %s: %s", path.c_str(), hipGetErrorString(hipGetLastError()));
    return err;
}

extern "C" int meme_index_load_files(meme_ctx* ctx, const char* prefix) {
    if (!ctx || !prefix) return MEME_E_ARG;
    const std::string p(prefix);
    // same file names memoryAllocLearned() opens (src/fastmap.cpp:425-470, 1496-1525)
    const std::string name[4] = {p + ".0123", p + ".pos_packed", p + ".suffixarray_uint64_L2_PARAMETERS", p + ".suffixarray_uint64_L1_PARAMETERS"};
    long long size[4];
    for (int i = 0; i < 4; ++i)
        if ((size[i] = file_size(name[i])) < 0) { meme_set_error("cannot read %s", name[i].c_str()); return MEME_E_IO; }
    if (size[1] % 5 || size[1] / 5 != size[0]) {
        meme_set_error("%s: .pos_packed (%lld B) and .0123 (%lld B) disagree on the suffix count", prefix, size[1], size[0]);
        return MEME_E_IO;
    }
    const bool trace = getenv("MEME_LOAD_TRACE") != nullptr;
    LoadBuffers LB;
    HIP_TRY(hipSetDevice(ctx->device));
    if (LB.init(getenv("MEME_LOAD_THREADS") && atoi(getenv("MEME_LOAD_THREADS")) > 0 ? atoi(getenv("MEME_LOAD_THREADS")) : 8)) { meme_set_error("pinned staging for the index load could not be allocated"); return MEME_E_HIP; }
    return index_build_from(ctx, size[0], size[3], size[2], [&](int which, void* d_dst, size_t bytes) {
        if ((long long)bytes != size[which]) { meme_set_error("%s changed size while loading", name[which].c_str()); return (int)MEME_E_IO; }
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = file_to_device(ctx, LB, name[which], d_dst, bytes);
        if (trace) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            fprintf(stderr, "[meme] %s: %.2f GB to the device in %.2f s (%.1f GB/s)\n", name[which].c_str(), bytes / 1e9, dt, bytes / 1e9 / dt);
        }
        return rc;
    }, [&](int which, int unit, const index_piece_fn& consume) {
        if (size[which] % unit) { meme_set_error("%s is not a whole number of %d-byte records", name[which].c_str(), unit); return (int)MEME_E_IO; }
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = file_to_device(ctx, LB, name[which], nullptr, (size_t)size[which], unit, &consume);
        if (trace) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            fprintf(stderr, "[meme] %s: %.2f GB through the device rings in %.2f s (%.1f GB/s)\n", name[which].c_str(), size[which] / 1e9, dt, size[which] / 1e9 / dt);
        }
        return rc;
    });
}

extern "C" int meme_index_attach(meme_ctx* ctx, const meme_index_arrays* a) {
    if (!ctx || !a || !a->d_sa_ent || !a->d_pac64 || !a->d_l2 || a->sa_num < 64) return MEME_E_ARG;
    drop_index(ctx);
    int rc = set_rmi(ctx, a->l2_records, a->l1_records);
    if (rc) return rc;
    ctx->idx.n = a->sa_num;
    ctx->idx.sa = (const SaEnt*)a->d_sa_ent;
    ctx->idx.pac = (const u64*)a->d_pac64;
    ctx->idx.l2 = (const Rmi32*)a->d_l2;
    ctx->idx.l1 = (const Rmi32*)a->d_l1;
    // the plcp table is derived data: built here into memory of the ctx (the caller's arrays stay the caller's)
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMalloc(&ctx->plcp_aux, (size_t)a->sa_num + 256));
    if ((rc = meme_stage_build_plcp(ctx, a->d_sa_ent, a->sa_num, a->d_pac64, ctx->plcp_aux))) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->idx.plcp = (const uint8_t*)ctx->plcp_aux;
    return MEME_OK;
}

extern "C" int meme_index_describe(meme_ctx* ctx, meme_index_arrays* out) {
    if (!ctx || !out) return MEME_E_ARG;
    if (!ctx->idx.sa) { meme_set_error("no index loaded"); return MEME_E_STATE; }
    out->sa_num = ctx->idx.n;
    out->d_sa_ent = (void*)ctx->idx.sa;
    out->d_pac64 = (void*)ctx->idx.pac;
    out->d_l2 = (void*)ctx->idx.l2;
    out->l2_records = ctx->idx.n_l2;
    out->d_l1 = (void*)ctx->idx.l1;
    out->l1_records = ctx->idx.n_l1;
    return MEME_OK;
}

// Multi-GPU start-up: copy a staged index to a ctx on another device, device to device over xGMI (no host round trip,
// no re-staging).  Equivalent of the RCCL broadcast of bench.py for hosts that drive all GPUs from one process (the
// reference's kt_for threads).
extern "C" int meme_index_replicate(meme_ctx* dst, meme_ctx* src) {
    if (!dst || !src || dst == src) return MEME_E_ARG;
    if (!src->idx.sa) { meme_set_error("meme_index_replicate: source has no index"); return MEME_E_STATE; }
    if (dst->device == src->device) return meme_index_share(dst, src);
    if (!src->owns_index) { meme_set_error("meme_index_replicate: the source ctx must own its index (load_host / load_files)"); return MEME_E_STATE; }
    HIP_TRY(hipSetDevice(dst->device));
    drop_index(dst);
    int can = 0;
    (void)hipDeviceCanAccessPeer(&can, dst->device, src->device);
    if (can) { hipError_t e = hipDeviceEnablePeerAccess(src->device, 0); (void)e; (void)hipGetLastError(); }
    DevIndex I = src->idx;
    for (auto& o : src->owned) {
        void* d = nullptr;
        int rc = own_alloc(dst, &d, o.second);
        if (rc) return rc;
        HIP_TRY(hipMemcpyPeerAsync(d, dst->device, o.first, src->device, o.second, dst->stream));
        if ((const void*)I.sa == o.first) I.sa = (const SaEnt*)d;
        if ((const void*)I.pac == o.first) I.pac = (const u64*)d;
        if ((const void*)I.l2 == o.first) I.l2 = (const Rmi32*)d;
        if ((const void*)I.l1 == o.first) I.l1 = (const Rmi32*)d;
        if ((const void*)I.plcp == o.first) I.plcp = (const uint8_t*)d;
    }
    HIP_TRY(hipStreamSynchronize(dst->stream));
    dst->idx = I;
    return MEME_OK;
}

extern "C" int meme_index_share(meme_ctx* ctx, meme_ctx* owner) {
    if (!ctx || !owner || ctx->device != owner->device) return MEME_E_ARG;
    if (!owner->idx.sa) { meme_set_error("owner has no index"); return MEME_E_STATE; }
    drop_index(ctx);
    ctx->idx = owner->idx;
    return MEME_OK;
}
