// The reference's PER-READ seeding API on the MI355X backend (second row; reference src/LearnedIndex_seeding.h:207-297).
//
// The aligner itself enters the backend once per -K chunk (meme_dropin.cpp); this file is for the OTHER callers of the reference's
// seeding interface -- test/Learned_seeding_big_read.cpp:247-280 (the harness BASELINE configs[1] names) and anything else written
// against
//     bool learned_index_load(L0path, L1path, L2path, suffix_array_num)                               src/LearnedIndex_seeding.h:290
//     void learned_index_cleanup()                                                                     :292
//     void Learned_getSMEMsAllPosOneThread(iaux, raux, smems, hits, hasN, split_len, split_width)      :265
//     void Learned_getSMEMsAllPosOneThread_step1only(...)                                              :257
//     void Learned_bwtSeedStrategyAllPosOneThread(iaux, raux, smems, hits, hasN)                       :241
//     void Learned_bwtSeedStrategyAllPosOneThread_mem_tradeoff(...)                                    :247
// with the reference's conventions: the caller owns the `smems` / `hits` kvecs (thread-private), the callee APPENDS (kv_push), hitbeg indexes
// the caller's hit array, the functions are re-entrant per thread and the index is process-global and read-only.  Written against the
// reference's own headers and linked, like the aligner's binding, in front of libbwa_pic.so: the definitions below win symbol resolution
// (oracle/Makefile.ref target learned_seeding_dropin = the reference's harness source, unmodified, + this file).
//
// A caller of this interface consumes a read's seeds right after the call, so a call is a batch of ONE read -- all latency (a few hundred
// microseconds against the reference's ~20 on a CPU core); the interface exists for drop-in completeness and for running the reference's own
// harness against the device, not for throughput: throughput callers use the batch surface (include/meme_hip.h).  Every thread gets a ctx of
// its own on first use (one stream, its own workspaces) that shares the one index in HBM.
//
// What a call computes: rounds 1 + 2 (`_step1only`: round 1) of the read in raux->unpacked_queue_buf for Learned_getSMEMs*, and for
// Learned_bwtSeedStrategy* the third round's seeds alone -- the device call has no "round 3 only" form, so the shim asks for rounds 1-3 and
// appends the multiset difference against rounds 1-2 of the same read (cached per thread from the call before; equal (start, end) keys carry
// equal hit lists, so the difference is well defined).  The in/out fields of raux that only steer the reference's own search (pivot,
// max_l_seq, max_refpos, cache_*) are left as they are: max_l_seq == 0 makes a caller choose the plain variant of the third round, and
// both variants land here anyway.  The single-search helpers (mem_search, right_smem_search, *_tradeoff, :207-236) are not interposed: they
// are the internals of the functions above, and callers that use them directly keep the reference's CPU code.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "bwa.h"                    // reference headers (-I$(REF)/src)
#include "FMI_search.h"
#include "LearnedIndex_seeding.h"

#include "meme_hip.h"               // our C ABI (-Iinclude)
#include "profiling.h"

// The reference's profiling counters are globals of its main.cpp (src/main.cpp:42), which a harness does not link; with the reference's objects as a
// shared library every one of them is referenced, so the executable that carries this file provides the storage (zero-initialised, as there).
uint64_t proc_freq, tprof[LIM_R][LIM_C], prof[LIM_R];

namespace {

std::mutex g_mu;
meme_ctx* g_owner = nullptr;                // holds the index
std::vector<meme_ctx*> g_all;               // every ctx made (destroyed by learned_index_cleanup)

[[noreturn]] void die(const char* what) {
    fprintf(stderr, "[meme-perread] %s: %s\n", what, meme_last_error());
    exit(1);
}

struct PerThread {
    meme_ctx* ctx = nullptr;
    // rounds 1 + 2 of the read the last Learned_getSMEMsAllPosOneThread call of this thread saw
    const uint8_t* seq = nullptr; int l_seq = -1; int min_seed_len = 0, split_len = 0, split_width = 0;
    std::vector<uint8_t> bases;
    std::vector<meme_mem_tl> smems; std::vector<uint64_t> hits;
};
thread_local PerThread tl;

meme_ctx* my_ctx() {
    if (tl.ctx) return tl.ctx;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_owner) { fprintf(stderr, "[meme-perread] a seeding call before learned_index_load()\n"); exit(1); }
    meme_ctx* c = meme_ctx_create(0);
    if (!c || meme_index_share(c, g_owner)) die("a thread's ctx");
    g_all.push_back(c);
    return tl.ctx = c;
}

// one read through the device: `rounds` rounds, results in the ctx's pinned buffers
void seed_one(const Learned_read_aux_t* raux, int min_seed_len, int split_len, int split_width, int max_mem_intv, int rounds, meme_seed_host_result* out) {
    meme_seed_opt so;
    so.min_seed_len = min_seed_len; so.split_len = split_len; so.split_width = split_width; so.max_mem_intv = max_mem_intv; so.rounds = rounds; so.hits_per_smem = 0;
    const int64_t off[2] = {0, raux->l_seq};
    if (meme_seed_batch_host(my_ctx(), raux->unpacked_queue_buf, off, 1, &so, out)) die("meme_seed_batch_host");
}

// the callee appends: mem_tl records with hitbeg pointing into the caller's hit array (src/LearnedIndex_seeding.cpp:2639-2657)
void append(mem_tlv* smems, u64v* hits, const meme_mem_tl& m, const uint64_t* h) {
    mem_tl t;
    t.start = m.start; t.end = m.end; t.hitbeg = (int)hits->n; t.hitcount = m.hitcount; t.cache_refpos = m.cache_refpos;
    for (int k = 0; k < m.hitcount; ++k) kv_push(uint64_t, *hits, h[k]);
    kv_push(mem_tl, *smems, t);
}

void rounds12(Learned_read_aux_t* raux, mem_tlv* smems, u64v* hits, int split_len, int split_width, int rounds) {
    meme_seed_host_result R;
    seed_one(raux, raux->min_seed_len, split_len, split_width, 0, rounds, &R);
    for (int64_t i = 0; i < R.total_smems; ++i) append(smems, hits, R.smems[i], R.hits + R.smems[i].hitbeg);
    // kept for the third round's call on the same read
    tl.seq = raux->unpacked_queue_buf; tl.l_seq = raux->l_seq; tl.min_seed_len = raux->min_seed_len; tl.split_len = split_len; tl.split_width = split_width;
    tl.bases.assign(raux->unpacked_queue_buf, raux->unpacked_queue_buf + raux->l_seq);
    tl.smems.assign(R.smems, R.smems + R.total_smems);
    tl.hits.assign(R.hits, R.hits + R.total_hits);
    if (rounds == 1) tl.l_seq = -1;                        // (a first round alone is not what the third round's difference is taken against)
}

void round3(Learned_read_aux_t* raux, mem_tlv* smems, u64v* hits) {
    // the caller has set raux->min_seed_len = min_seed_len + 1 and raux->min_intv_limit = max_mem_intv for this round (src/bwamem.cpp:1379-1392)
    const int min_seed_len = raux->min_seed_len - 1;
    const bool cached = tl.l_seq == raux->l_seq && tl.min_seed_len == min_seed_len && (int)tl.bases.size() == raux->l_seq &&
                        memcmp(tl.bases.data(), raux->unpacked_queue_buf, (size_t)raux->l_seq) == 0;
    int split_len = tl.split_len, split_width = tl.split_width;
    if (!cached) {                                          // the third round alone: rounds 1 + 2 with the aligner's defaults (src/bwamem.cpp:126-162, 1348) first
        split_len = (int)(min_seed_len * 1.5 + .499); split_width = 10;
        meme_seed_host_result R2;
        seed_one(raux, min_seed_len, split_len, split_width, 0, 2, &R2);
        tl.smems.assign(R2.smems, R2.smems + R2.total_smems);
    }
    std::vector<uint64_t> seen;                              // (start, end) of the rounds before, a multiset
    seen.reserve(tl.smems.size());
    for (const meme_mem_tl& m : tl.smems) seen.push_back((uint64_t)(uint32_t)m.start << 32 | (uint32_t)m.end);
    std::sort(seen.begin(), seen.end());
    meme_seed_host_result R;
    seed_one(raux, min_seed_len, split_len, split_width, raux->min_intv_limit, 3, &R);
    std::vector<int64_t> order((size_t)R.total_smems);
    for (int64_t i = 0; i < R.total_smems; ++i) order[(size_t)i] = i;
    auto key = [&](int64_t i) { return (uint64_t)(uint32_t)R.smems[i].start << 32 | (uint32_t)R.smems[i].end; };
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return key(a) < key(b); });
    size_t s = 0;
    for (size_t k = 0; k < order.size();) {
        size_t e = k;
        while (e < order.size() && key(order[e]) == key(order[k])) ++e;
        while (s < seen.size() && seen[s] < key(order[k])) ++s;
        size_t before = 0;
        while (s + before < seen.size() && seen[s + before] == key(order[k])) ++before;
        for (size_t j = k + before; j < e; ++j) append(smems, hits, R.smems[order[j]], R.hits + R.smems[order[j]].hitbeg);   // what the third round added
        k = e;
    }
    tl.l_seq = -1;
}

}  // namespace

bool learned_index_load(char const* dataPath, char const* dataPath2, char const* dataPath3, double suffix_array_num) {
    (void)dataPath;                                         // (the L0 file is ignored by the reference as well, src/LearnedIndex_seeding.cpp:74-122)
    static const char suffix[] = ".suffixarray_uint64_L2_PARAMETERS";
    std::string p = dataPath3 ? dataPath3 : "";
    if (p.size() <= sizeof(suffix) - 1 || p.compare(p.size() - (sizeof(suffix) - 1), sizeof(suffix) - 1, suffix) != 0) {
        fprintf(stderr, "[meme-perread] learned_index_load: %s is not <prefix>%s\n", p.c_str(), suffix);
        return false;
    }
    (void)dataPath2;
    const std::string prefix = p.substr(0, p.size() - (sizeof(suffix) - 1));
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_owner) return true;
    meme_ctx* c = meme_ctx_create(0);
    if (!c) { fprintf(stderr, "[meme-perread] meme_ctx_create: %s\n", meme_last_error()); return false; }
    if (meme_index_load_files(c, prefix.c_str())) { fprintf(stderr, "[meme-perread] meme_index_load_files(%s): %s\n", prefix.c_str(), meme_last_error()); meme_ctx_destroy(c); return false; }   // (false: the caller exits, as with the reference)
    meme_index_arrays ia;
    if (meme_index_describe(c, &ia) || (double)ia.sa_num != suffix_array_num) {
        fprintf(stderr, "[meme-perread] the index at %s has %lld suffixes, the caller expects %.0f\n", prefix.c_str(), (long long)ia.sa_num, suffix_array_num);
        meme_ctx_destroy(c);
        return false;
    }
    g_owner = c;
    g_all.push_back(c);
    return true;
}

void learned_index_cleanup() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = g_all.size(); i-- > 0;) meme_ctx_destroy(g_all[i]);       // (the owner last)
    g_all.clear();
    g_owner = nullptr;
}

void Learned_getSMEMsAllPosOneThread(Learned_index_aux_t* iaux, Learned_read_aux_t* raux, mem_tlv* smems, u64v* hits, bool hasN, int split_len, int split_width) {
    (void)iaux; (void)hasN;
    rounds12(raux, smems, hits, split_len, split_width, 2);
}
void Learned_getSMEMsAllPosOneThread_step1only(Learned_index_aux_t* iaux, Learned_read_aux_t* raux, mem_tlv* smems, u64v* hits, bool hasN, int split_len, int split_width) {
    (void)iaux; (void)hasN;
    rounds12(raux, smems, hits, split_len, split_width, 1);
}
void Learned_bwtSeedStrategyAllPosOneThread(Learned_index_aux_t* iaux, Learned_read_aux_t* raux, mem_tlv* smems, u64v* hits, bool hasN) {
    (void)iaux; (void)hasN;
    round3(raux, smems, hits);
}
void Learned_bwtSeedStrategyAllPosOneThread_mem_tradeoff(Learned_index_aux_t* iaux, Learned_read_aux_t* raux, mem_tlv* smems, u64v* hits, bool hasN) {
    (void)iaux; (void)hasN;
    round3(raux, smems, hits);
}
