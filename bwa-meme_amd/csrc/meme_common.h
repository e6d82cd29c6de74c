// Shared declarations of the HIP backend (gfx950 / MI355X only -- no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <utility>
#include <vector>

#include "meme_hip.h"

typedef uint64_t u64;
typedef int64_t i64;

// ---- HBM layout of the index ---------------------------------------------------------------------
// One suffix-array slot: the first 32 bases of the suffix as an integer whose unsigned order is the
// lexicographic order (first base in bits 63..62, T-filled past the text end), plus the text position.
// 16-byte aligned: one probe = one dwordx4 load = one 64-B HBM sector.
struct __attribute__((aligned(16))) SaEnt {
    u64 key;
    u64 pos;
};

struct RmiRec {   // on-disk P-RMI record (reference src/LearnedIndex_seeding.cpp:197-206)
    double icpt;
    double slope;
    u64 err;
};

struct Rmi32 {    // the same record padded to 32 bytes in HBM: a lookup never straddles a 128-byte line
    double icpt;
    double slope;
    u64 err;
    u64 pad;
};

struct DevIndex {
    i64 n = 0;                 // sa_num = 2 * l_pac
    const SaEnt* sa = nullptr;
    const u64* pac = nullptr;  // 2-bit text, base i in bits (62 - 2*(i&31)) of word i>>5
    const Rmi32* l2 = nullptr;
    const Rmi32* l1 = nullptr;
    i64 n_l2 = 0, n_l1 = 0;
    int shift = 64;            // key >> shift = leaf index
    // plcp[u] = min(255, longest common prefix of the suffix at TEXT position u with either of its suffix-array neighbours): the depth
    // down to which that suffix is not alone in its suffix-array interval.  Derived from sa + pac when an index is staged
    // (k_build_plcp); what the re-seeding verifier (k_reseed) walks instead of searching.  1 byte per suffix.
    const uint8_t* plcp = nullptr;
};

struct DevBuf {   // growable device workspace
    void* p = nullptr;
    size_t cap = 0;
};

struct meme_ctx {
    int device = 0;
    int n_cus = 256;                   // compute units of the device (cached: hipGetDeviceProperties is slow)
    hipStream_t stream = nullptr;
    DevIndex idx;
    bool owns_index = false;
    std::vector<std::pair<void*, size_t>> owned;   // device allocations of the index (pointer, bytes)
    void* plcp_aux = nullptr;                      // the plcp table of an attached index (meme_index_attach: the arrays are the caller's, this is ours)
    // workspaces
    DevBuf reads, read_off, slots[3], ovf[2], slot_cnt, slot_hits, slot_loc, smem_off, hit_off, smems, hits,
           scan_tmp, counters, pend, blk, pairs, refb, qerb, packed, bsw_order, bsw_ws, chain[14], ext[19], gcig[11], kswv[7], sam[9], mate[8];
    // pinned host staging owned by the ctx (results of meme_seed_batch_host, inputs of meme_bsw_batch)
    struct HostBuf { void* p = nullptr; size_t cap = 0; } h_smems, h_hits, h_smem_off, h_hit_off, h_misc, h_chain[7], h_ext[2], h_gcig[3], h_kswv, h_sam[2], h_mate[4];
    i64 last_seed_max_len = 0;         // longest read of that batch
    i64 last_seed_reads = 0;           // reads of the batch whose seeds are in smems / hits (input of meme_chain_last_batch_host)
    bool reads_resident = false;       // ctx->reads holds the bases of that batch (false after meme_chain_batch_host: seeds brought by the caller)
    // tuning
    i64 seed_blocks = 0;               // 0 = auto
    i64 smem_cap = 128;                // per-read SMEM slots in the search kernel's scratch (tier 0; 3 KB per read.  With 64 a handful of
                                       // the benchmark's 10 M reads overflowed and their sequential re-run cost every step 1.9 ms)
    i64 group_lanes = 4;               // lanes per read in the search kernel (4, 8, 16, 32)
    i64 seed_blocks_per_cu = 5;
    i64 max_batch = 0;                 // > 0: the batch calls behind seeding (extension, global alignment) refuse more reads / jobs than this with
                                       // MEME_E_CAPACITY, as they do when their scratch would not fit: a caller's memory bound, and how the tests reach that path
    i64 ext_split = 1;                 // 1: the extension stage's read-walking kernels run eight lanes per read for reads with at most 8 chained seeds, a wavefront per read for the rest; 0: a wavefront per read
    i64 bsw_circ = 1;                  // 1: lane-per-pair banded SW of queries longer than 2w + 2 columns keeps its columns in a ring (k_bsw_lane_circ); 0: a word per query column
    i64 sam_max_batch = 0;             // > 0: meme_sam_format_batch_host refuses more record slots than this with MEME_E_CAPACITY (the caller then formats in pieces)
    i64 seed_early_tier = 1;           // 1: the overflow tier of the reads known to have overflowed after k_reseed runs beside the re-seeding batches
    i64 ext_live_only = 0;             // 1: meme_extend_last_batch_host hands over the surviving records only (qe > qb: what src/bwamem.cpp:1680-1693 keeps)
    i64 ext_rounds = 1;                // with ext_live_only: rounds of one seed per read before everything still ahead is extended at once (0: the reference's batch, then compaction)
    i64 gcig_groups = 1;               // 1: CIGAR jobs with bands of at most 16 / 32 / 64 columns run 4 / 2 / 1 to a wavefront as one chunk per row (k_gcig_grp); 0: a wavefront each, 64-column chunks; 2: also bands of 65-128 columns as one chunk, two columns per lane (measured slower: off by default)
    i64 gcig_zcap = -1;                // >= 0: bytes of LDS per CIGAR job for its backtrack matrix / window (default: 8192 where a typical matrix of the batch fits, else 2048)
    i64 ext_census = 0;                // 1: the extension stage counts its exact-prefix jobs (a measurement, profiles/r05_bsw.md)
    i64 seed_defer = 1;                // 1: re-seeding regions of unique SMEMs are verified on the plcp table (k_reseed) instead of searched
    i64 chain_light_hits = 32;         // reads with more hits to walk skip the lane-per-read tier: LDS tier at once, beside it
    i64 chain_lane_hits = 256;         // hits per read the lane-per-read chaining tier walks; reads with more go to the wavefront tiers at once
    i64 chain_side_priority = 0;       // 1: the side streams of the routed chaining tiers are created with the highest stream priority (set before the first chaining call)
    i64 chain_wave_tiers = 1;          // 0: the chaining stage skips the LDS tier (everything beyond the lane tier through the B-tree tier; tests)
    i64 bsw_blocks = 0;
    i64 bsw_lane_min_pairs = 32768;   // batches at least this big use the lane-per-pair kernel (throughput); smaller ones the
                                       // lanes-per-pair kernel (latency: a lone pair takes ~6 ms on one lane, ~0.3 ms on 64)
    // timings
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_chain[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_ext[2] = {nullptr, nullptr};
    hipEvent_t ev_gcig[2] = {nullptr, nullptr};
    hipEvent_t ev_kswv[2] = {nullptr, nullptr};
    hipEvent_t ev_sam[2] = {nullptr, nullptr};
    i64 sam_text_reads = 0;            // reads of the batch whose names / qualities meme_sam_stage_text staged (0: none)
    bool sam_has_quals = false;
    hipStream_t stream_side[3] = {nullptr, nullptr, nullptr};   // the routed chaining tiers run beside the lane-per-read tier
    hipEvent_t ev_side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_aux = nullptr;
    hipStream_t stream_emit = nullptr;                          // k_reseed_emit runs beside the blocked regions' rounds
    hipEvent_t ev_emit[2] = {nullptr, nullptr};
    i64 chain_reads = 0, chain_tier2_reads = 0, chain_tier3_reads = 0;   // of the last meme_chain_run(): reads chained, of which by the wavefront-per-read tier
    meme_timings tm = {};
};

void meme_set_error(const char* fmt, ...);
int meme_buf_reserve(meme_ctx* ctx, DevBuf& b, size_t bytes);
// side stream i of the ctx (the routed chaining tiers, the early overflow tier of seeding), created on first use -- ONE place, so that the tuning
// "chain_side_priority" applies to all three whichever stage touches a stream first (advisor, round 5)
int meme_side_stream(meme_ctx* ctx, int i);
int meme_hostbuf_reserve(meme_ctx* ctx, meme_ctx::HostBuf& b, size_t bytes);
// exclusive prefix sum of n 64-bit counts into out[0..n] (out[n] = total), asynchronous on ctx->stream (meme_scan.hip)
int meme_scan_exclusive(meme_ctx* ctx, const i64* d_in, i64* d_out, i64 n);
// the banded-SW kernels on device-resident pairs, no host synchronisation (meme_bsw.hip); host_maxq = an upper bound of the query lengths or -1
int meme_bsw_launch(meme_ctx* ctx, meme_seqpair* d_pairs, const uint8_t* d_ref, const uint8_t* d_qer, int npairs, int w, const meme_bsw_opt* opt,
                    int host_maxq);
// local alignment scores of the jobs mem_flt_chained_seeds poses (meme_kswv.hip): window [rb, rb + tlen) of the text against the tlen x qlen
// bases at reads[qoff]; sc[seed] = score.  The job count is read on the device; max_jobs sizes the launch.
constexpr int MEME_SEEDSW_MAX = 200;    // MEM_SHORT_LEN, src/bwamem.cpp:250: windows are shorter
struct meme_seedsw_job { i64 rb; i64 qoff; int seed; short tlen, qlen; };
int meme_seedsw_launch(meme_ctx* ctx, const meme_seedsw_job* d_jobs, const unsigned long long* d_njobs, i64 max_jobs, int* d_sc, const meme_ext_opt* o);
// the mate-rescue kernels on jobs whose sequences are already in ctx->kswv[2] / [3] and whose records are in ctx->kswv[0] (meme_kswv.hip; the jobs also on the host,
// for the sort into LDS classes); staged = false: meme_kswv_batch_host's own path (sequences and jobs come from the host)
int meme_kswv_run(meme_ctx* ctx, const meme_kswv_job* jobs, int64_t njobs, const uint8_t* ref, int64_t ref_bytes, const uint8_t* qer, int64_t qer_bytes, const meme_bsw_opt* opt,
                  bool staged, meme_kswv_host_result* out);
// chains of the batch just seeded, left in HBM (meme_chain.hip); totals[0..1] = chains, chained seeds
int meme_chain_run(meme_ctx* ctx, const meme_contig* contigs, int32_t n_contigs, const meme_chain_opt* opt, i64* totals);

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            meme_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MEME_E_HIP;                                                                 \
        }                                                                                      \
    } while (0)

// ---- device helpers --------------------------------------------------------------------------------
// text position of suffix-array slot i from the 5-byte image: the two aligned dwords around the record
__device__ __forceinline__ u64 load_pos5(const uint8_t* pos5, i64 i) {
    const i64 bo = i * 5;
    const uint32_t* pd = reinterpret_cast<const uint32_t*>(pos5 + (bo & ~3ll));
    const u64 two = (u64)pd[0] | ((u64)pd[1] << 32);
    const u64 v5 = (two >> (8 * (int)(bo & 3))) & 0xffffffffffull;
    return ((v5 & 0xffffffffull) << 8) | (v5 >> 32);
}
// 32 bases starting at base offset s from a big-endian-within-word 2-bit array
__device__ __forceinline__ u64 extract32(const u64* w, i64 s) {
    i64 k = s >> 5;
    int sh = (int)(s & 31) * 2;
    u64 a = w[k], b = w[k + 1];
    return sh ? (a << sh) | (b >> (64 - sh)) : a;
}
