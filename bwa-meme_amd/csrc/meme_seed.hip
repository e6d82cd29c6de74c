// Learned-index seeding on MI355X: host side of the C ABI (launch logic, offsets scan, hit gather).
// The search kernel itself lives in meme_seed_kernel.h.
//
// Replaces the per-read loop body of mem_kernel1_core_Learned() (reference src/bwamem.cpp:1249-1394); the
// outputs are the mem_tl records and hit positions its unchanged consumer mem_chain_Learned() reads
// (src/bwamem.cpp:1122-1204): hits of one SMEM in ascending suffix-array order.
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <utility>

#include "meme_seed_kernel.h"

namespace {

using namespace seedk;

// ---- offsets: two-level exclusive scan over reads -------------------------------------------------------
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ __forceinline__ i64 block_scan_excl(i64 v, i64* total) {
    __shared__ i64 wsum[SCAN_BLOCK / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    i64 x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        i64 y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    i64 off = 0, tot = 0;
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        if (w < wid) off += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return off + x - v;
}

// pass 1: per-tile sums of (smem count, hit count)
__global__ void __launch_bounds__(SCAN_BLOCK) k_tile_sums(const int* __restrict__ cnt, const i64* __restrict__ hits, i64 n,
                                                          i64* __restrict__ tile_sums) {
    i64 base = (i64)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    i64 a = 0, b = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        i64 i = base + k;
        if (i < n) {
            int c = cnt[i];
            a += c < 0 ? 0 : c;
            b += hits[i];
        }
    }
    i64 ta, tb;
    block_scan_excl(a, &ta);
    block_scan_excl(b, &tb);
    if (threadIdx.x == 0) { tile_sums[2 * blockIdx.x] = ta; tile_sums[2 * blockIdx.x + 1] = tb; }
}

// pass 2: one block scans the tile sums in place (exclusive); totals to out[0..1]
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tiles(i64* __restrict__ tile_sums, i64 ntiles, i64* __restrict__ totals) {
    i64 ca = 0, cb = 0;
    for (i64 base = 0; base < ntiles; base += SCAN_BLOCK) {
        i64 i = base + threadIdx.x;
        i64 a = i < ntiles ? tile_sums[2 * i] : 0, b = i < ntiles ? tile_sums[2 * i + 1] : 0;
        i64 ta, tb;
        i64 ea = block_scan_excl(a, &ta);
        i64 eb = block_scan_excl(b, &tb);
        if (i < ntiles) { tile_sums[2 * i] = ca + ea; tile_sums[2 * i + 1] = cb + eb; }
        ca += ta;
        cb += tb;
    }
    if (threadIdx.x == 0) { totals[0] = ca; totals[1] = cb; }
}

// pass 3: per-read offsets
__global__ void __launch_bounds__(SCAN_BLOCK) k_offsets(const int* __restrict__ cnt, const i64* __restrict__ hits, i64 n,
                                                        const i64* __restrict__ tile_sums, i64* __restrict__ smem_off,
                                                        i64* __restrict__ hit_off) {
    i64 base = (i64)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    i64 va[SCAN_ITEMS], vb[SCAN_ITEMS], a = 0, b = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        i64 i = base + k;
        va[k] = vb[k] = 0;
        if (i < n) {
            int c = cnt[i];
            va[k] = c < 0 ? 0 : c;
            vb[k] = hits[i];
        }
        a += va[k];
        b += vb[k];
    }
    i64 ta, tb;
    i64 ea = block_scan_excl(a, &ta) + tile_sums[2 * blockIdx.x];
    i64 eb = block_scan_excl(b, &tb) + tile_sums[2 * blockIdx.x + 1];
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        i64 i = base + k;
        if (i < n) { smem_off[i] = ea; hit_off[i] = eb; }
        ea += va[k];
        eb += vb[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_BLOCK - 1) { smem_off[n] = ea; hit_off[n] = eb; }
}

// ---- compaction + hit gather: one 16-lane group per read --------------------------------------------------
struct TierTable {
    const SlotRec* base[N_TIERS];
    int cap[N_TIERS];
};

__global__ void __launch_bounds__(BLOCK) k_gather(const SaEnt* __restrict__ sa, TierTable tiers, const i64* __restrict__ slot_loc,
                                                   const int* __restrict__ cnt, i64 n, int hits_per_smem,
                                                   const i64* __restrict__ smem_off, const i64* __restrict__ hit_off,
                                                   meme_mem_tl* __restrict__ smems, u64* __restrict__ hits) {
    // one 16-lane group per read, one lane per SMEM: the slot reads, the first-hit lookups and the mem_tl stores of up
    // to 16 SMEMs go out together; hit lists longer than a few entries are copied cooperatively afterwards
    constexpr int GL = 16;
    const int lane = threadIdx.x & 63;
    const int t = lane & (GL - 1);
    const int gbase = lane - t;
    i64 gid = ((i64)blockIdx.x * BLOCK + threadIdx.x) / GL;
    const i64 ngroups = (i64)gridDim.x * BLOCK / GL;
    for (i64 r = gid; r < n; r += ngroups) {
        const int c = cnt[r];
        if (c <= 0) continue;
        const i64 loc = slot_loc[r];
        const int tier = (int)(loc >> 40);
        const SlotRec* sl = tiers.base[tier] + (loc & ((1ll << 40) - 1)) * tiers.cap[tier];
        meme_mem_tl* out = smems + smem_off[r];
        u64* hout = hits + hit_off[r];
        i64 hb0 = 0;                                   // hits before this chunk of SMEMs
        for (int k0 = 0; k0 < c; k0 += GL) {
            const int k = k0 + t;
            const bool act = k < c;
            SlotRec s;
            s.start = s.end = 0; s.sa_start = 0; s.count = 0;
            if (act) s = sl[k];
            i64 h = s.count;
            if (hits_per_smem > 0 && h > hits_per_smem) h = hits_per_smem;
            // exclusive prefix of h over the group's lanes
            i64 incl = h;
#pragma unroll
            for (int d = 1; d < GL; d <<= 1) {
                i64 y = __shfl_up(incl, d, GL);
                if (t >= d) incl += y;
            }
            const i64 hb = hb0 + incl - h;
            u64 first = 0;
            if (act) {
                // an SMEM with one occurrence usually carries the position itself (the search had it in registers)
                first = (s.sa_start & SLOT_POS) ? (u64)(s.sa_start & SLOT_VAL) : sa[s.sa_start].pos;
                meme_mem_tl m;
                m.start = s.start;
                m.end = s.end;
                m.hitbeg = (int32_t)hb;
                m.hitcount = s.count > (i64)INT_MAX ? INT_MAX : (int32_t)s.count;
                m.cache_refpos = first;
                out[k] = m;
                if (h > 0) hout[hb] = first;
                for (i64 i = 1; i < h && i < 4; ++i) hout[hb + i] = sa[s.sa_start + i].pos;
            }
            // long hit lists: all lanes of the group copy them together
            u64 longmask = (__ballot(act && h > 4) >> gbase) & 0xffffull;
            while (longmask) {
                const int src = __ffsll((long long)longmask) - 1;
                longmask &= longmask - 1;
                const i64 lh = __shfl(h, src, GL), lhb = __shfl(hb, src, GL), lsa = __shfl(s.sa_start, src, GL);
                for (i64 i = 4 + t; i < lh; i += GL) hout[lhb + i] = sa[lsa + i].pos;
            }
            hb0 += __shfl(incl, GL - 1, GL);
        }
    }
}

template <int G>
int launch_k_seed(hipStream_t stream, const SeedArgs& A, size_t lds, i64 blocks) {
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_seed<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_seed<G>, dim3((unsigned)blocks), dim3(BLOCK), lds, stream, A);
    HIP_TRY(hipGetLastError());
    return MEME_OK;
}

int launch_seed(meme_ctx* ctx, const uint8_t* d_reads, const i64* d_read_off, i64 nreads, i64 max_len, i64 total_bytes,
                const meme_seed_opt* opt, meme_seed_result* out) {
    unsigned long long h_counters[16];
    int rc;
    ctx->last_seed_reads = 0;          // whatever batch meme_chain_last_batch_host could have chained is being overwritten
    ctx->sam_text_reads = 0;           // ... and the names / qualities staged for it belong to the previous batch
    if ((rc = meme_buf_reserve(ctx, ctx->slot_cnt, (size_t)nreads * sizeof(int)))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->slot_hits, (size_t)nreads * sizeof(i64)))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->slot_loc, (size_t)nreads * sizeof(i64)))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->counters, 2 * 16 * sizeof(unsigned long long)))) return rc;     // two sets: an overflow tier may run beside the re-seeding kernels
    const int dev_cus = ctx->n_cus;
    // ---- pack the reads: 2 bits/base, both strands, N masks (k_pack_reads) ---------------------------------
    // the reference exits on reads longer than LEARNED_MAX_READ_LEN (src/bwamem.cpp:1259-1262): fail loudly, never seed part of a batch
    if (max_len > MAX_READ_LEN) {
        meme_set_error("read of %lld bases exceeds the learned-index limit of %d (LEARNED_MAX_READ_LEN)", (long long)max_len, MAX_READ_LEN);
        return MEME_E_ARG;
    }
    if (max_len < 1) max_len = 1;
    const i64 stage_len = max_len;                                // longest read as staged by the packing kernel
    PackGeom geo;
    geo.W = (int)((max_len + 31) / 32) + 2;
    geo.MW = (int)((max_len + 63) / 64);
    geo.stride = 2 * geo.W + 2 * geo.MW + 1;
    if ((rc = meme_buf_reserve(ctx, ctx->packed, (size_t)nreads * geo.stride * 8))) return rc;
    HIP_TRY(hipEventRecord(ctx->ev[6], ctx->stream));
    {
        int rb = (int)((48 * 1024) / stage_len);                    // reads per workgroup: <= 48 KB of staged bytes
        if (rb > 32) rb = 32;
        if (rb < 1) rb = 1;
        i64 pblocks = (nreads + rb - 1) / rb;
        if (pblocks > (i64)dev_cus * 16) pblocks = (i64)dev_cus * 16;   // grid-stride beyond that
        size_t plds = ((size_t)rb * (size_t)stage_len + 16 + 48 + 3) & ~(size_t)3;   // + the packer's 9-dword reads past a read's last word
        hipLaunchKernelGGL(k_pack_reads, dim3((unsigned)pblocks), dim3(256), plds, ctx->stream, d_reads,
                           d_read_off, nreads, total_bytes, geo, rb, (u64*)ctx->packed.p);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(ctx->ev[7], ctx->stream));
    float ms_total = 0.f, ms_reseed = 0.f;
    i64 launches = 0, searches = 0, windows = 0, lane_searches = 0;
    TierTable tiers;
    for (int t = 0; t < N_TIERS; ++t) { tiers.base[t] = nullptr; tiers.cap[t] = 0; }
    // One k_seed launch of a tier: `n_todo` reads (tier 0: the batch; overflow tiers: the reads named in `pending`), SMEM slots in
    // ctx->slots[tier], the reads that overflow THIS tier appended to ctx->ovf[tier & 1], counters in set `cset`.
    auto launch_tier = [&](int tier, i64 n_todo, const i64* pending, int cset, hipStream_t stream, bool defer) -> int {
        // overflow tiers hold a 512-entry SMEM ring per read in LDS: run them 32 lanes per read (8 reads per block)
        int G = tier == 0 ? (int)ctx->group_lanes : 32;
        const int cap = tier == 0 ? (int)ctx->smem_cap : TIER_CAP[tier];
        const int lcap = cap < TIER_LCAP[tier] ? cap : TIER_LCAP[tier];
        DevBuf& sb = ctx->slots[tier];
        DevBuf& ob = ctx->ovf[tier & 1];
        const size_t need = (size_t)n_todo * cap * sizeof(SlotRec);
        if (tier > 0 && need > sb.cap) {
            // overflow tiers re-run the reads that emitted more SMEMs than their slots hold, with 32 x more slots each:
            // refuse instead of exhausting the HBM the index lives in
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b / 2) {
                meme_set_error("%lld reads emitted more than %d SMEMs each: re-running them needs %.1f GB of slots, more than half of the "
                               "free HBM (%.1f GB); seed this batch in smaller pieces", (long long)n_todo, tiers.cap[tier - 1],
                               need / 1e9, free_b / 1e9);
                return MEME_E_CAPACITY;
            }
        }
        int rc2;
        if ((rc2 = meme_buf_reserve(ctx, sb, need))) return rc2;
        if ((rc2 = meme_buf_reserve(ctx, ob, (size_t)n_todo * sizeof(i64)))) return rc2;
        unsigned long long* counters = (unsigned long long*)ctx->counters.p + 16 * cset;
        HIP_TRY(hipMemsetAsync(counters, 0, 16 * sizeof(unsigned long long), stream));
        SeedArgs A;
        A.I = ctx->idx;
        A.packed = (const u64*)ctx->packed.p;
        A.read_off = d_read_off;
        A.nreads = n_todo;
        A.geo = geo;
        A.opt = *opt;
        A.slots = (SlotRec*)sb.p;
        A.slot_cnt = (int*)ctx->slot_cnt.p;
        A.slot_hits = (i64*)ctx->slot_hits.p;
        A.slot_loc = (i64*)ctx->slot_loc.p;
        A.pending = pending;
        A.ovf_list = (i64*)ob.p;
        A.cap = cap;
        A.lcap = lcap;
        A.tier = tier;
        A.counters = counters;
        A.defer = defer ? 1 : 0;
        tiers.base[tier] = (const SlotRec*)sb.p;
        tiers.cap[tier] = cap;
        while (G < 32 && seed_lds_bytes(G, geo, lcap) > (size_t)160 * 1024) G *= 2;   // long reads: fewer reads per workgroup
        const int groups = BLOCK / G;
        size_t lds = seed_lds_bytes(G, geo, lcap);
        i64 want = (n_todo + groups - 1) / groups;
        i64 blocks = ctx->seed_blocks > 0 ? ctx->seed_blocks : (i64)dev_cus * ctx->seed_blocks_per_cu;
        if (blocks > want) blocks = want;
        if (blocks < 1) blocks = 1;
        if (tier == 0) HIP_TRY(hipEventRecord(ctx->ev[0], stream));
        switch (G) {
        case 1: return launch_k_seed<1>(stream, A, lds, blocks);
        case 2: return launch_k_seed<2>(stream, A, lds, blocks);
        case 4: return launch_k_seed<4>(stream, A, lds, blocks);
        case 8: return launch_k_seed<8>(stream, A, lds, blocks);
        case 16: return launch_k_seed<16>(stream, A, lds, blocks);
        case 32: return launch_k_seed<32>(stream, A, lds, blocks);
        default: meme_set_error("group_lanes must be 1, 2, 4, 8, 16 or 32"); return MEME_E_ARG;
        }
    };
    auto tally = [&](const unsigned long long* h) {
        ++launches;
        searches += (i64)h[1];
        windows += (i64)h[3];
#ifdef SEED_PROF
        {
            double tot = 0; for (int k = 0; k < 8; ++k) tot += (double)h[4 + k];
            fprintf(stderr, "[seed prof]:");
            const char* nm[6] = {"control", "request+rmi", "window+compare", "resolve", "level", "apply"};
            for (int k = 0; k < 6; ++k) fprintf(stderr, " %s %.1f%%", nm[k], 100.0 * (double)h[4 + k] / (tot > 0 ? tot : 1));
            fprintf(stderr, "\n");
        }
#endif
    };
    // ---- tier 0: the whole batch.  It leaves the re-seeding regions of unique SMEMs to k_reseed (a walk on the plcp table, one lane per
    // read) and the batches of searches behind it; the overflow tiers search everything themselves.
    const bool defer = ctx->seed_defer != 0 && ctx->idx.plcp != nullptr && opt->rounds >= 2;
    if (defer && (rc = meme_buf_reserve(ctx, ctx->pend, (size_t)nreads * sizeof(i64)))) return rc;
    if (defer && (rc = meme_buf_reserve(ctx, ctx->blk, (size_t)nreads * BLK_PER_READ * 2 * sizeof(BlkRec)))) return rc;
    if ((rc = launch_tier(0, nreads, nullptr, 0, ctx->stream, defer))) return rc;
    i64 n_early = -1;                  // reads of the tier-1 launch that ran beside the re-seeding kernels (-1: none did)
    { const int src = meme_side_stream(ctx, 0); if (src) return src; }
    if (!ctx->ev_side[0]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_side[0], hipEventDisableTiming));
    if (!ctx->ev_aux) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_aux, hipEventDisableTiming));
    if (defer) {
        const int cap = (int)ctx->smem_cap;
        HIP_TRY(hipEventRecord(ctx->ev[4], ctx->stream));
        ReseedArgs R;
        R.I = ctx->idx; R.packed = (const u64*)ctx->packed.p; R.geo = geo; R.nreads = nreads; R.opt = *opt;
        R.slots = (SlotRec*)ctx->slots[0].p; R.cap = cap; R.slot_cnt = (int*)ctx->slot_cnt.p; R.slot_hits = (i64*)ctx->slot_hits.p;
        R.ovf_list = (i64*)ctx->ovf[0].p; R.counters = (unsigned long long*)ctx->counters.p; R.pend_list = (i64*)ctx->pend.p;
        R.blk = (BlkRec*)ctx->blk.p; R.blk_out = R.blk + nreads * BLK_PER_READ; R.blk_cap = nreads * BLK_PER_READ;
        R.blk_ctr = 14; R.blk_out_ctr = 15;
        i64 rblocks = (nreads + 255) / 256;
        if (rblocks > (i64)dev_cus * 32) rblocks = (i64)dev_cus * 32;
        hipLaunchKernelGGL(k_reseed, dim3((unsigned)rblocks), dim3(256), 0, ctx->stream, R);
        HIP_TRY(hipEventRecord(ctx->ev_aux, ctx->stream));
        // the batch searches (list sizes stay on the device: fixed grids, grid-stride loops), then the blocked regions' passes
        const unsigned sblocks = (unsigned)(rblocks < (i64)dev_cus * 4 ? rblocks : (i64)dev_cus * 4);
        // the pending intervals on a stream of their own, beside the blocked regions' rounds (both are latency-bound lane-per-item
        // kernels; they touch different slots of the reads they share)
        if (!ctx->stream_emit) { HIP_TRY(hipStreamCreateWithFlags(&ctx->stream_emit, hipStreamNonBlocking)); HIP_TRY(hipEventCreateWithFlags(&ctx->ev_emit[0], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&ctx->ev_emit[1], hipEventDisableTiming)); }
        HIP_TRY(hipEventRecord(ctx->ev_emit[0], ctx->stream));
        HIP_TRY(hipStreamWaitEvent(ctx->stream_emit, ctx->ev_emit[0], 0));
        hipLaunchKernelGGL(k_reseed_emit, dim3(sblocks), dim3(256), 0, ctx->stream_emit, R);
        HIP_TRY(hipEventRecord(ctx->ev_emit[1], ctx->stream_emit));
        constexpr int ROUNDS = 4;                           // (named configuration: 3 rounds leave 2.2 ms of one-by-one searches to the last pass)
        for (int round = 0; round < ROUNDS; ++round) {     // blocked regions ping-pong between two lists; the last pass searches for itself
            hipLaunchKernelGGL(k_reseed_search, dim3(sblocks), dim3(256), 0, ctx->stream, R);
            if (round < ROUNDS - 1) {
                hipLaunchKernelGGL(k_reseed_resume<false>, dim3(sblocks), dim3(256), 0, ctx->stream, R);
                HIP_TRY(hipMemsetAsync((unsigned long long*)ctx->counters.p + R.blk_ctr, 0, sizeof(unsigned long long), ctx->stream));
                std::swap(R.blk, R.blk_out); std::swap(R.blk_ctr, R.blk_out_ctr);
            } else hipLaunchKernelGGL(k_reseed_resume<true>, dim3(sblocks), dim3(256), 0, ctx->stream, R);
        }
        HIP_TRY(hipGetLastError());
        // The reads that overflowed their slots in k_seed or in k_reseed's pass are known once k_reseed has finished -- the only re-seeding
        // kernel that looks at every read; the ones behind it work from lists of reads that did NOT overflow.  Their tier-1 launch (a few
        // homopolymer reads of thousands of SMEMs each: 2.5 ms at the named configuration however few they are) runs beside the batches
        // of searches instead of behind them.  Reads that overflow later (appended by a resume pass: rare) get a second launch below.
        if (ctx->seed_early_tier != 0) {
            unsigned long long h_ovf = 0;
            HIP_TRY(hipStreamWaitEvent(ctx->stream_side[0], ctx->ev_aux, 0));
            HIP_TRY(hipMemcpyAsync(&h_ovf, (unsigned long long*)ctx->counters.p + 2, 8, hipMemcpyDeviceToHost, ctx->stream_side[0]));
            HIP_TRY(hipStreamSynchronize(ctx->stream_side[0]));
            if (h_ovf > 0) {
                n_early = (i64)h_ovf;
                if ((rc = launch_tier(1, n_early, (const i64*)ctx->ovf[0].p, 1, ctx->stream_side[0], false))) return rc;
                HIP_TRY(hipEventRecord(ctx->ev_side[0], ctx->stream_side[0]));
                HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_side[0], 0));
            }
        }
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_emit[1], 0));
    }
    HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    HIP_TRY(hipMemcpyAsync(h_counters, ctx->counters.p, sizeof(h_counters), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        ms_total += ms;
        if (defer) {
            float a = 0.f;
            HIP_TRY(hipEventElapsedTime(&a, ctx->ev[4], ctx->ev[1]));
            ms_reseed += a;
            lane_searches += (i64)h_counters[12];
#ifdef RESEED_PROF
            fprintf(stderr, "[reseed prof] (wave-time in 10 ns ticks, executions) stage windows %llu / %llu, table walk %llu / %llu, lane search %llu / %llu\n", h_counters[4], h_counters[5],
                    h_counters[6], h_counters[7], h_counters[8], h_counters[9]);
#endif
            if (getenv("MEME_SEED_TRACE")) fprintf(stderr, "[meme] seed tier 0: search kernel + re-seeding kernels %.2f ms, of which behind k_seed %.2f ms (%lld lane searches; %lld reads of the "
                                                   "overflow tier beside them)\n", ms, a, (long long)h_counters[12], (long long)(n_early < 0 ? 0 : n_early));
        }
    }
    tally(h_counters);
    // ---- overflow tiers: reads that produced more SMEMs than their slots hold (pathological repeats) are re-run alone
    i64 n_todo = (i64)h_counters[2];
    const i64* pending = (const i64*)ctx->ovf[0].p;
    for (int tier = 1; n_todo > 0; ++tier) {
        if (tier >= N_TIERS) {
            meme_set_error("a read produced more than %d SMEMs", TIER_CAP[N_TIERS - 1]);
            return MEME_E_CAPACITY;
        }
        const int cset = tier & 1;
        if (tier == 1 && n_early == n_todo) {
            // the launch beside the re-seeding kernels took them all: only its counters are left to read
            HIP_TRY(hipMemcpyAsync(h_counters, (unsigned long long*)ctx->counters.p + 16 * cset, sizeof(h_counters), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        } else {
            // (tier 1 after an early launch that did not see every overflowed read: all of them again -- rare, and simple)
            HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
            if ((rc = launch_tier(tier, n_todo, pending, cset, ctx->stream, false))) return rc;
            HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
            HIP_TRY(hipMemcpyAsync(h_counters, (unsigned long long*)ctx->counters.p + 16 * cset, sizeof(h_counters), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
            ms_total += ms;
        }
        tally(h_counters);
        n_todo = (i64)h_counters[2];
        pending = (const i64*)ctx->ovf[tier & 1].p;
    }
    {
        float pms = 0.f;
        HIP_TRY(hipEventElapsedTime(&pms, ctx->ev[6], ctx->ev[7]));
        ctx->tm.seed_pack_ms = pms;
    }
    ctx->tm.seed_kernel_ms = ms_total;
    ctx->tm.seed_launches = launches;
    ctx->tm.seed_windows = windows;
    ctx->tm.seed_reseed_ms = ms_reseed;
    ctx->tm.seed_lane_searches = lane_searches;
    // offsets
    i64 ntiles = (nreads + SCAN_TILE - 1) / SCAN_TILE;
    if ((rc = meme_buf_reserve(ctx, ctx->scan_tmp, (size_t)(2 * ntiles + 2) * sizeof(i64)))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->smem_off, (size_t)(nreads + 1) * sizeof(i64)))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->hit_off, (size_t)(nreads + 1) * sizeof(i64)))) return rc;
    i64* tiles = (i64*)ctx->scan_tmp.p;
    i64* totals = tiles + 2 * ntiles;
    HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    hipLaunchKernelGGL(k_tile_sums, dim3((unsigned)ntiles), dim3(SCAN_BLOCK), 0, ctx->stream, (const int*)ctx->slot_cnt.p,
                       (const i64*)ctx->slot_hits.p, nreads, tiles);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(SCAN_BLOCK), 0, ctx->stream, tiles, ntiles, totals);
    hipLaunchKernelGGL(k_offsets, dim3((unsigned)ntiles), dim3(SCAN_BLOCK), 0, ctx->stream, (const int*)ctx->slot_cnt.p,
                       (const i64*)ctx->slot_hits.p, nreads, (const i64*)tiles, (i64*)ctx->smem_off.p,
                       (i64*)ctx->hit_off.p);
    HIP_TRY(hipGetLastError());
    i64 h_tot[2];
    HIP_TRY(hipMemcpyAsync(h_tot, totals, sizeof(h_tot), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if ((rc = meme_buf_reserve(ctx, ctx->smems, (size_t)(h_tot[0] + 1) * sizeof(meme_mem_tl)))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->hits, (size_t)(h_tot[1] + 1) * sizeof(u64)))) return rc;
    i64 gblocks = (nreads + 15) / 16;
    if (gblocks > (i64)dev_cus * 8) gblocks = (i64)dev_cus * 8;
    if (gblocks < 1) gblocks = 1;
    hipLaunchKernelGGL(k_gather, dim3((unsigned)gblocks), dim3(BLOCK), 0, ctx->stream, ctx->idx.sa, tiers,
                       (const i64*)ctx->slot_loc.p, (const int*)ctx->slot_cnt.p, nreads, opt->hits_per_smem,
                       (const i64*)ctx->smem_off.p, (const i64*)ctx->hit_off.p, (meme_mem_tl*)ctx->smems.p,
                       (u64*)ctx->hits.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float gms = 0.f;
    HIP_TRY(hipEventElapsedTime(&gms, ctx->ev[2], ctx->ev[3]));
    ctx->tm.seed_gather_ms = gms;
    out->d_smems = (const meme_mem_tl*)ctx->smems.p;
    out->d_smem_off = (const i64*)ctx->smem_off.p;
    out->d_hits = (const u64*)ctx->hits.p;
    out->d_hit_off = (const i64*)ctx->hit_off.p;
    out->total_smems = h_tot[0];
    out->total_hits = h_tot[1];
    out->searches = searches;
    return MEME_OK;
}

// longest read of the batch (sizes the packed layout and the LDS tile)
__global__ void __launch_bounds__(256) k_max_len(const i64* __restrict__ off, i64 n, int* out) {
    __shared__ int wmax[4];
    int v = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        int l = (int)(off[i + 1] - off[i]);
        v = v > l ? v : l;
    }
    for (int d = 32; d >= 1; d >>= 1) { int y = __shfl_xor(v, d); v = v > y ? v : y; }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) v = v > wmax[k] ? v : wmax[k];
        atomicMax(out, v);
    }
}

int device_max_len(meme_ctx* ctx, const i64* d_read_off, i64 nreads, i64* max_len) {
    int rc;
    if ((rc = meme_buf_reserve(ctx, ctx->scan_tmp, 64))) return rc;
    HIP_TRY(hipMemsetAsync(ctx->scan_tmp.p, 0, 4, ctx->stream));
    i64 mblocks = (nreads + 255) / 256;
    if (mblocks > 1024) mblocks = 1024;
    hipLaunchKernelGGL(k_max_len, dim3((unsigned)mblocks), dim3(256), 0, ctx->stream, d_read_off, nreads,
                       (int*)ctx->scan_tmp.p);
    int v = 0;
    HIP_TRY(hipMemcpyAsync(&v, ctx->scan_tmp.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *max_len = v;
    return MEME_OK;
}

int check_opt(const meme_seed_opt* opt) {
    if (!opt || opt->min_seed_len < 1 || opt->rounds < 1 || opt->rounds > 3 || opt->hits_per_smem < 0) {
        meme_set_error("bad meme_seed_opt");
        return MEME_E_ARG;
    }
    return MEME_OK;
}

}  // namespace

extern "C" int meme_seed_batch_device(meme_ctx* ctx, const uint8_t* d_reads, const int64_t* d_read_off, int64_t nreads,
                                      int64_t total_bases, const meme_seed_opt* opt, meme_seed_result* out) {
    if (!ctx || !d_reads || !d_read_off || !out || nreads < 0) return MEME_E_ARG;
    int rc = check_opt(opt);
    if (rc) return rc;
    if (!ctx->idx.sa) { meme_set_error("meme_seed_batch: no index loaded"); return MEME_E_STATE; }
    HIP_TRY(hipSetDevice(ctx->device));
    if (nreads == 0) { memset(out, 0, sizeof(*out)); return MEME_OK; }
    i64 max_len = 0;
    if ((rc = device_max_len(ctx, (const i64*)d_read_off, nreads, &max_len))) return rc;
    return launch_seed(ctx, d_reads, (const i64*)d_read_off, nreads, max_len, total_bases, opt, out);
}

extern "C" int meme_seed_batch(meme_ctx* ctx, const uint8_t* reads, const int64_t* read_off, int64_t nreads,
                               const meme_seed_opt* opt, meme_mem_tl* smems, int64_t smem_capacity, int64_t* smem_off,
                               uint64_t* hits, int64_t hit_capacity, int64_t* hit_off, int64_t* total_smems,
                               int64_t* total_hits) {
    if (!ctx || !reads || !read_off || !smems || !smem_off || !hits || !hit_off || nreads < 0) return MEME_E_ARG;
    int rc = check_opt(opt);
    if (rc) return rc;
    if (!ctx->idx.sa) { meme_set_error("meme_seed_batch: no index loaded"); return MEME_E_STATE; }
    HIP_TRY(hipSetDevice(ctx->device));
    if (nreads == 0) { smem_off[0] = 0; hit_off[0] = 0; if (total_smems) *total_smems = 0; if (total_hits) *total_hits = 0; return MEME_OK; }
    const i64 bases = read_off[nreads] - read_off[0];
    if (read_off[0] != 0) { meme_set_error("read_off[0] must be 0"); return MEME_E_ARG; }
    if ((rc = meme_buf_reserve(ctx, ctx->reads, (size_t)bases + 16))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->read_off, (size_t)(nreads + 1) * sizeof(i64)))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->reads.p, reads, (size_t)bases, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->read_off.p, read_off, (size_t)(nreads + 1) * sizeof(i64), hipMemcpyHostToDevice, ctx->stream));
    meme_seed_result res;
    i64 max_len = 0;
    for (i64 i = 0; i < nreads; ++i) max_len = read_off[i + 1] - read_off[i] > max_len ? read_off[i + 1] - read_off[i] : max_len;
    rc = launch_seed(ctx, (const uint8_t*)ctx->reads.p, (const i64*)ctx->read_off.p, nreads, max_len, bases, opt, &res);
    if (rc) return rc;
    if (total_smems) *total_smems = res.total_smems;
    if (total_hits) *total_hits = res.total_hits;
    HIP_TRY(hipMemcpyAsync(smem_off, res.d_smem_off, (size_t)(nreads + 1) * sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(hit_off, res.d_hit_off, (size_t)(nreads + 1) * sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    if (res.total_smems > smem_capacity || res.total_hits > hit_capacity) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        meme_set_error("output buffers too small: need %lld SMEMs and %lld hits", (long long)res.total_smems,
                       (long long)res.total_hits);
        return MEME_E_CAPACITY;
    }
    HIP_TRY(hipMemcpyAsync(smems, res.d_smems, (size_t)res.total_smems * sizeof(meme_mem_tl), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(hits, res.d_hits, (size_t)res.total_hits * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MEME_OK;
}

// Host-pointer variant whose results land in pinned buffers owned by the ctx (valid until the next seeding call on it):
// no capacity negotiation, and the device-to-host copies are plain DMA transfers.  `reads` / `read_off` may be pageable or
// pinned (meme_host_alloc).  This is the call a chunk-level binding issues once per -K chunk.
// Workspaces and pinned result buffers of a meme_seed_batch_host() + meme_chain_last_batch_host() call of this size, ahead of
// time: pinned host memory takes ~0.4 s per GB to allocate, which a caller can hide behind its index load.  Sizes are estimates
// (12 SMEMs, 24 hits, 3 chains, 6 chained seeds per read); whatever turns out larger grows on first use as before.
extern "C" int meme_seed_reserve(meme_ctx* ctx, int64_t nreads, int64_t total_bases) {
    if (!ctx || nreads < 0 || total_bases < 0) return MEME_E_ARG;
    if (nreads == 0) return MEME_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)nreads;
    const i64 len = total_bases / nreads + 32;
    const size_t stride = (size_t)(2 * ((len + 31) / 32 + 2) + 2 * ((len + 63) / 64) + 1);
    int rc;
    struct { DevBuf* b; size_t bytes; } dev[] = {
        {&ctx->reads, (size_t)total_bases + 16}, {&ctx->read_off, (n + 1) * 8}, {&ctx->slot_cnt, n * 4}, {&ctx->slot_hits, n * 8},
        {&ctx->slot_loc, n * 8}, {&ctx->counters, 2 * 16 * 8}, {&ctx->pend, n * 8}, {&ctx->blk, n * BLK_PER_READ * 2 * sizeof(BlkRec)}, {&ctx->packed, n * stride * 8}, {&ctx->slots[0], n * (size_t)ctx->smem_cap * sizeof(SlotRec)},
        {&ctx->smem_off, (n + 1) * 8}, {&ctx->hit_off, (n + 1) * 8}, {&ctx->smems, n * 12 * sizeof(meme_mem_tl)}, {&ctx->hits, n * 24 * 8},
        {&ctx->chain[0], n * 16 * 32}, {&ctx->chain[1], n * 16 * 8 * 16}, {&ctx->chain[2], n * 24}, {&ctx->chain[3], n * 4}, {&ctx->chain[8], n * 8},
        {&ctx->chain[5], (n + 1) * 32 + n * 5 + 64}, {&ctx->chain[6], n * 3 * sizeof(meme_chain)}, {&ctx->chain[7], n * 6 * sizeof(meme_chain_seed)}};
    for (auto& d : dev) if ((rc = meme_buf_reserve(ctx, *d.b, d.bytes))) return rc;
    struct { meme_ctx::HostBuf* b; size_t bytes; } host[] = {
        {&ctx->h_smem_off, (n + 1) * 8}, {&ctx->h_hit_off, (n + 1) * 8}, {&ctx->h_smems, n * 12 * sizeof(meme_mem_tl)}, {&ctx->h_hits, n * 24 * 8},
        {&ctx->h_chain[0], (n + 1) * 8}, {&ctx->h_chain[1], n * 3 * sizeof(meme_chain)}, {&ctx->h_chain[2], (n + 1) * 8},
        {&ctx->h_chain[3], n * 6 * sizeof(meme_chain_seed)}, {&ctx->h_chain[4], n * 4}, {&ctx->h_chain[5], n * 4}, {&ctx->h_chain[6], n}};
    for (auto& h : host) if ((rc = meme_hostbuf_reserve(ctx, *h.b, h.bytes))) return rc;
    // the stages behind seeding (meme_extend_last_batch_host, meme_global_batch_host) at their usual sizes for short reads: 4 alignment
    // records and 4 extension jobs of ~250 sequence bytes per read, one global alignment per 3 reads -- the first chunks of a run
    // otherwise pay for growing these (pinned memory: ~0.4 s per GB)
    struct { DevBuf* b; size_t bytes; } dev2[] = {
        {&ctx->ext[0], n * 3 * 16}, {&ctx->ext[1], n * 4 * sizeof(meme_alnreg)}, {&ctx->ext[2], n * 4 * 4}, {&ctx->ext[3], (n + 1) * 48},
        {&ctx->ext[4], n * 2 * sizeof(meme_seqpair)}, {&ctx->ext[5], n * 2 * sizeof(meme_seqpair)}, {&ctx->ext[6], n * 4 * sizeof(meme_seqpair)},
        {&ctx->ext[7], n * 4 * 250}};
    for (auto& d : dev2) if ((rc = meme_buf_reserve(ctx, *d.b, d.bytes))) return rc;
    if ((rc = meme_hostbuf_reserve(ctx, ctx->h_ext[0], (n + 1) * 8)) || (rc = meme_hostbuf_reserve(ctx, ctx->h_ext[1], n * 4 * sizeof(meme_alnreg))) ||
        (rc = meme_hostbuf_reserve(ctx, ctx->h_gcig[0], n / 2 * sizeof(meme_gres))) || (rc = meme_hostbuf_reserve(ctx, ctx->h_gcig[1], n * 4))) return rc;
    return MEME_OK;
}

// FASTQ letters (or codes) -> base codes in place, as mem_kernel1_core_Learned does it on the host (src/bwamem.cpp:1277-1279:
// c < 4 ? c : nst_nt4_table[c]); 16 bytes per lane (the buffer is padded to a multiple of 16)
__global__ void __launch_bounds__(256) k_ascii_to_codes(uint4* __restrict__ p, i64 nvec) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (i64)gridDim.x * blockDim.x) {
        uint4 v = p[i];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t c = (w[k] >> (8 * b)) & 0xffu, u = c & 0xdfu;        // u: upper case
                const uint32_t code = c < 4 ? c : (u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : 4u);
                o |= code << (8 * b);
            }
            w[k] = o;
        }
        v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
        p[i] = v;
    }
}

// reads from the host into HBM + the seeding kernels; `out` (may be null) receives the results in the ctx's pinned buffers
static int seed_host_reads(meme_ctx* ctx, const uint8_t* reads, const int64_t* read_off, int64_t nreads, const meme_seed_opt* opt, const char* who,
                           meme_seed_host_result* out, int64_t* totals, bool ascii = false) {
    if (!ctx || !reads || !read_off || nreads < 0) return MEME_E_ARG;
    int rc = check_opt(opt);
    if (rc) return rc;
    if (!ctx->idx.sa) { meme_set_error("%s: no index loaded", who); return MEME_E_STATE; }
    HIP_TRY(hipSetDevice(ctx->device));
    if (out) {
        memset(out, 0, sizeof(*out));
        if ((rc = meme_hostbuf_reserve(ctx, ctx->h_smem_off, (size_t)(nreads + 1) * sizeof(i64)))) return rc;
        if ((rc = meme_hostbuf_reserve(ctx, ctx->h_hit_off, (size_t)(nreads + 1) * sizeof(i64)))) return rc;
        out->smem_off = (const int64_t*)ctx->h_smem_off.p;
        out->hit_off = (const int64_t*)ctx->h_hit_off.p;
    }
    if (totals) totals[0] = totals[1] = 0;
    ctx->last_seed_reads = 0;
    if (nreads == 0) { if (out) { ((i64*)ctx->h_smem_off.p)[0] = 0; ((i64*)ctx->h_hit_off.p)[0] = 0; } return MEME_OK; }
    if (read_off[0] != 0) { meme_set_error("read_off[0] must be 0"); return MEME_E_ARG; }
    const i64 bases = read_off[nreads];
    if ((rc = meme_buf_reserve(ctx, ctx->reads, (size_t)bases + 16))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->read_off, (size_t)(nreads + 1) * sizeof(i64)))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->reads.p, reads, (size_t)bases, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->read_off.p, read_off, (size_t)(nreads + 1) * sizeof(i64), hipMemcpyHostToDevice, ctx->stream));
    if (ascii) {
        const i64 nvec = (bases + 15) / 16;
        i64 cb = (nvec + 255) / 256;
        if (cb > (i64)ctx->n_cus * 16) cb = (i64)ctx->n_cus * 16;
        hipLaunchKernelGGL(k_ascii_to_codes, dim3((unsigned)(cb < 1 ? 1 : cb)), dim3(256), 0, ctx->stream, (uint4*)ctx->reads.p, nvec);
        HIP_TRY(hipGetLastError());
    }
    i64 max_len = 0;
    for (i64 i = 0; i < nreads; ++i) max_len = read_off[i + 1] - read_off[i] > max_len ? read_off[i + 1] - read_off[i] : max_len;
    meme_seed_result res;
    rc = launch_seed(ctx, (const uint8_t*)ctx->reads.p, (const i64*)ctx->read_off.p, nreads, max_len, bases, opt, &res);
    if (rc) return rc;
    if (out) {
        if ((rc = meme_hostbuf_reserve(ctx, ctx->h_smems, (size_t)(res.total_smems + 1) * sizeof(meme_mem_tl)))) return rc;
        if ((rc = meme_hostbuf_reserve(ctx, ctx->h_hits, (size_t)(res.total_hits + 1) * sizeof(u64)))) return rc;
        HIP_TRY(hipMemcpyAsync(ctx->h_smem_off.p, res.d_smem_off, (size_t)(nreads + 1) * sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->h_hit_off.p, res.d_hit_off, (size_t)(nreads + 1) * sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->h_smems.p, res.d_smems, (size_t)res.total_smems * sizeof(meme_mem_tl), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->h_hits.p, res.d_hits, (size_t)res.total_hits * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
        out->smems = (const meme_mem_tl*)ctx->h_smems.p;
        out->hits = (const uint64_t*)ctx->h_hits.p;
        out->total_smems = res.total_smems;
        out->total_hits = res.total_hits;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));             // (the caller's read buffer is free again)
    if (totals) { totals[0] = res.total_smems; totals[1] = res.total_hits; }
    ctx->last_seed_reads = nreads;
    ctx->reads_resident = true;
    ctx->last_seed_max_len = max_len;
    return MEME_OK;
}

extern "C" int meme_seed_batch_host(meme_ctx* ctx, const uint8_t* reads, const int64_t* read_off, int64_t nreads,
                                    const meme_seed_opt* opt, meme_seed_host_result* out) {
    if (!out) return MEME_E_ARG;
    return seed_host_reads(ctx, reads, read_off, nreads, opt, "meme_seed_batch_host", out, nullptr);
}

extern "C" int meme_seed_batch_resident(meme_ctx* ctx, const uint8_t* reads, const int64_t* read_off, int64_t nreads,
                                        const meme_seed_opt* opt, int64_t* total_smems, int64_t* total_hits) {
    int64_t tot[2] = {0, 0};
    const int rc = seed_host_reads(ctx, reads, read_off, nreads, opt, "meme_seed_batch_resident", nullptr, tot);
    if (total_smems) *total_smems = tot[0];
    if (total_hits) *total_hits = tot[1];
    return rc;
}

// Same, for reads as they stand in the FASTQ file (letters; bytes below 4 are taken as codes): the conversion of
// mem_kernel1_core_Learned (src/bwamem.cpp:1277-1279) runs on the device, so a caller only has to gather the bytes.
extern "C" int meme_seed_batch_resident_ascii(meme_ctx* ctx, const uint8_t* reads, const int64_t* read_off, int64_t nreads,
                                              const meme_seed_opt* opt, int64_t* total_smems, int64_t* total_hits) {
    int64_t tot[2] = {0, 0};
    const int rc = seed_host_reads(ctx, reads, read_off, nreads, opt, "meme_seed_batch_resident_ascii", nullptr, tot, true);
    if (total_smems) *total_smems = tot[0];
    if (total_hits) *total_hits = tot[1];
    return rc;
}
