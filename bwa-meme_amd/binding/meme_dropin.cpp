// Reference-side binding of the MI355X backend (what INTEGRATION.md describes), written against the *reference's own
// headers* and linked into the reference aligner (oracle/Makefile.ref target bwa-meme_dropin).  The reference objects
// are built position-independent into libbwa_pic.so; the definitions below live in the executable and therefore win
// symbol resolution (ELF interposition) -- no reference source is modified or copied, the calls below go to functions
// the reference exports.  A maintainer integrating the backend would put the same code behind an #ifdef at the four
// places named here:
//
//   memoryAllocLearned()            src/fastmap.cpp:351-641   worker buffers as before, but the index goes to HBM
//                                   (meme_index_load_files + meme_index_replicate per extra GPU) instead of being
//                                   expanded on the host (13-byte suffix-array entries + ISA, ~100 s / ~120 GB at GRCh38)
//   mem_process_seqs()              src/bwamem.cpp:1920-1972  ONE meme_seed_batch_host() per -K chunk (split over the
//                                   visible GPUs) before kt_for(worker_bwt); the reference's own body then runs unchanged
//   mem_kernel1_core_Learned()      src/bwamem.cpp:1230-1413  per 512-read batch: takes the chunk's precomputed SMEMs and
//                                   hits, then the reference's ks_introsort / mem_chain_Learned / mem_chain_flt /
//                                   mem_flt_chained_seeds
//   BandedPairWiseSW::getScores8 / getScores16 / scalarBandedSWAWrapper   src/bandedSWA.cpp:242-260,1970-,2664-
//                                   -> meme_bsw_batch(); the concurrent calls of the kt_for workers are combined into one
//                                   backend call per GPU (group commit), staged through pinned buffers
#include <dlfcn.h>
#include <sched.h>
#include <time.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "fastmap.h"             // reference headers (-I$(REF)/src): ktp_aux_t, worker_t, mem_opt_t, bseq1_t ...
#include "bandedSWA.h"
#include "ksort.h"

#include "meme_hip.h"            // our C ABI (-Iinclude)

// reference functions used unchanged
void mem_chain_Learned(const mem_opt_t* opt, const bntseq_t* bns, int len, mem_tlv* smems, mem_chain_v* chain,
                       int seqid, u64v* hits, mem_seed_t* seedBuf, int64_t seedBufSize, int64_t& seedBufCount, int tid);
int mem_chain_flt(const mem_opt_t* opt, int n_chn_, mem_chain_t* a_, int tid);
void mem_flt_chained_seeds(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, bseq1_t* seq_, int n_chn,
                           mem_chain_t* a);

#define dropin_smem_lt(a, b) ((a).start == (b).start ? (a).end < (b).end : (a).start < (b).start)
KSORT_INIT(meme_dropin_smem, mem_tl, dropin_smem_lt)

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
[[noreturn]] void die(const char* what) {
    fprintf(stderr, "[meme-dropin] %s: %s\n", what, meme_last_error());
    exit(1);
}
bool verbose() { static const bool v = getenv("MEME_DROPIN_VERBOSE") != nullptr; return v; }

// ---- devices -------------------------------------------------------------------------------------------------
struct Device {
    meme_ctx* seed = nullptr;     // owns (device 0) or holds a replica of the index
    meme_ctx* bsw = nullptr;
};
std::vector<Device> g_dev;
std::mutex g_mu;
std::atomic<double> g_t_seed{0}, g_t_bsw_gather{0}, g_t_bsw_call{0}, g_t_bsw_kernel{0};
std::atomic<int64_t> g_n_bsw_calls{0}, g_n_bsw_pairs{0}, g_n_seed_reads{0};

void init_devices(const char* prefix) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_dev.empty()) return;
    int n = meme_device_count();
    if (n <= 0) die("no HIP device");
    if (getenv("MEME_DROPIN_DEVICES")) { int want = atoi(getenv("MEME_DROPIN_DEVICES")); if (want >= 1 && want < n) n = want; }
    g_dev.resize((size_t)n);
    const double t0 = now_s();
    for (int d = 0; d < n; ++d) {
        if (!(g_dev[(size_t)d].seed = meme_ctx_create(d))) die("meme_ctx_create");
        if (!(g_dev[(size_t)d].bsw = meme_ctx_create(d))) die("meme_ctx_create");
        // small combined calls keep the lanes-per-pair kernels; combined calls of the whole thread team are big enough
        // for the lane-per-pair kernel much earlier than a lone caller's
        if (getenv("MEME_DROPIN_BSW_LANE_MIN")) meme_set_tuning(g_dev[(size_t)d].bsw, "bsw_lane_min_pairs", atoll(getenv("MEME_DROPIN_BSW_LANE_MIN")));
    }
    if (meme_index_load_files(g_dev[0].seed, prefix)) die("meme_index_load_files");
    const double t1 = now_s();
    std::vector<std::thread> th;
    for (int d = 1; d < n; ++d)                                 // device-to-device over xGMI, all replicas at once
        th.emplace_back([d] { if (meme_index_replicate(g_dev[(size_t)d].seed, g_dev[0].seed)) die("meme_index_replicate"); });
    for (auto& t : th) t.join();
    fprintf(stderr, "[meme-dropin] index staged in HBM in %.2f s, replicated to %d more GPU(s) in %.2f s\n", t1 - t0, n - 1,
            now_s() - t1);
}

// ---- memoryAllocLearned (src/fastmap.cpp:351-641) ----------------------------------------------------------------
// Worker buffers exactly as the reference sizes them (they are indexed by the kt_for thread id all over
// mem_chain2aln_across_reads_V2 and freed by process(), src/fastmap.cpp:1098-1110); the host-side index expansion is gone.
uint8_t bitrev8(uint8_t b) {
    b = (uint8_t)(((b & 0xF0) >> 4) | ((b & 0x0F) << 4));
    b = (uint8_t)(((b & 0xCC) >> 2) | ((b & 0x33) << 2));
    return (uint8_t)(((b & 0xAA) >> 1) | ((b & 0x55) << 1));
}

}  // namespace

void memoryAllocLearned(ktp_aux_t* aux, worker_t& w, int32_t nreads, int32_t nthreads, char* idx_prefix) {
    const double t0 = now_s();
    const int64_t memSize = nreads;
    w.regs = (mem_alnreg_v*)calloc((size_t)memSize, sizeof(mem_alnreg_v));
    w.chain_ar = (mem_chain_v*)malloc((size_t)memSize * sizeof(mem_chain_v));
    w.seedBuf = (mem_seed_t*)calloc(sizeof(mem_seed_t), (size_t)memSize * AVG_SEEDS_PER_READ);
    if (!w.regs || !w.chain_ar || !w.seedBuf) { fprintf(stderr, "[meme-dropin] out of memory\n"); exit(1); }
    w.seedBufSize = BATCH_SIZE * AVG_SEEDS_PER_READ;
    const int64_t wsize = BATCH_SIZE * SEEDS_PER_READ;
    for (int l = 0; l < nthreads; ++l) {
        w.mmc.seqBufLeftRef[l * CACHE_LINE] = (uint8_t*)_mm_malloc((size_t)wsize * MAX_SEQ_LEN_REF + MAX_LINE_LEN, 64);
        w.mmc.seqBufLeftQer[l * CACHE_LINE] = (uint8_t*)_mm_malloc((size_t)wsize * MAX_SEQ_LEN_QER + MAX_LINE_LEN, 64);
        w.mmc.seqBufRightRef[l * CACHE_LINE] = (uint8_t*)_mm_malloc((size_t)wsize * MAX_SEQ_LEN_REF + MAX_LINE_LEN, 64);
        w.mmc.seqBufRightQer[l * CACHE_LINE] = (uint8_t*)_mm_malloc((size_t)wsize * MAX_SEQ_LEN_QER + MAX_LINE_LEN, 64);
        w.mmc.wsize_buf_ref[l * CACHE_LINE] = wsize * MAX_SEQ_LEN_REF;
        w.mmc.wsize_buf_qer[l * CACHE_LINE] = wsize * MAX_SEQ_LEN_QER;
        w.mmc.seqPairArrayAux[l] = (SeqPair*)malloc((size_t)(wsize + MAX_LINE_LEN) * sizeof(SeqPair));
        w.mmc.seqPairArrayLeft128[l] = (SeqPair*)malloc((size_t)(wsize + MAX_LINE_LEN) * sizeof(SeqPair));
        w.mmc.seqPairArrayRight128[l] = (SeqPair*)malloc((size_t)(wsize + MAX_LINE_LEN) * sizeof(SeqPair));
        w.mmc.wsize[l] = wsize;
        w.mmc.lim[l] = (int32_t*)_mm_malloc((BATCH_SIZE + 32) * sizeof(int32_t), 64);
        if (!w.mmc.seqBufLeftRef[l * CACHE_LINE] || !w.mmc.seqBufLeftQer[l * CACHE_LINE] || !w.mmc.seqBufRightRef[l * CACHE_LINE] ||
            !w.mmc.seqBufRightQer[l * CACHE_LINE] || !w.mmc.seqPairArrayAux[l] || !w.mmc.seqPairArrayLeft128[l] ||
            !w.mmc.seqPairArrayRight128[l] || !w.mmc.lim[l]) { fprintf(stderr, "[meme-dropin] out of memory\n"); exit(1); }
    }
    // forward + reverse-complement 2-bit text in the byte order the reference's seeding code uses (src/fastmap.cpp:441-457).
    // The reference hands this array -- not idx->pac -- to mem_flt_chained_seeds (src/bwamem.cpp:1407, 1768), so the
    // binding has to provide the very same bytes for the SAM output to be identical.
    const int64_t l_pac = aux->fmi->idx->bns->l_pac;
    const int64_t ll_pac = (l_pac * 2 + 3) / 4 * 4;
    w.rc_pac = (uint8_t*)malloc((size_t)(ll_pac / 4));
    if (!w.rc_pac) { fprintf(stderr, "[meme-dropin] out of memory\n"); exit(1); }
    const uint8_t* pac = aux->fmi->idx->pac;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < ll_pac / 4; ++k) {
        uint8_t b = 0;
        for (int j = 0; j < 4; ++j) {
            const int64_t p = 4 * k + j;
            int c = 0;
            if (p < l_pac) c = pac[p >> 2] >> ((~p & 3) << 1) & 3;
            else if (p < 2 * l_pac) { const int64_t q = 2 * l_pac - 1 - p; c = 3 - (pac[q >> 2] >> ((~q & 3) << 1) & 3); }
            b = (uint8_t)(b | (c << ((~j & 3) << 1)));
        }
        w.rc_pac[k] = bitrev8(b);
    }
    w.sa_position = nullptr;                                   // the suffix array lives in HBM
    w.ref2sa = nullptr;
    w.smemBufSize = MAX_LINE_LEN * sizeof(mem_tlv);
    w.l_smems = (mem_tlv*)malloc((size_t)nthreads * w.smemBufSize);
    w.hitBufSize = MAX_LINE_LEN * sizeof(u64v);
    w.hits_ar = (u64v*)malloc((size_t)nthreads * w.hitBufSize);
    if (!w.l_smems || !w.hits_ar) { fprintf(stderr, "[meme-dropin] out of memory\n"); exit(1); }
    for (int i = 0; i < nthreads; ++i) {
        kv_init_base(mem_tl, w.l_smems[i * MAX_LINE_LEN], BATCH_MUL * READ_LEN);
        kv_init_base(uint64_t, w.hits_ar[i * MAX_LINE_LEN], 65536);
    }
    w.useErt = 0;
    w.useLearned = 1;
    const double t1 = now_s();
    const char* prefix = getenv("MEME_INDEX_PREFIX") ? getenv("MEME_INDEX_PREFIX") : idx_prefix;
    init_devices(prefix);
    fprintf(stderr, "[meme-dropin] worker buffers + fwd/rc text %.2f s, HBM index %.2f s (no host-side index expansion)\n",
            t1 - t0, now_s() - t1);
}

// ---- chunk-level seeding ----------------------------------------------------------------------------------------------
namespace {

struct ChunkPart {                     // the slice of a chunk one GPU seeded
    int64_t first = 0, count = 0;
    meme_seed_host_result res;
    uint8_t* flat = nullptr; int64_t flat_cap = 0;       // pinned staging (grow-only)
    int64_t* off = nullptr; int64_t off_cap = 0;
};
struct Chunk {
    const bseq1_t* seqs = nullptr;
    int64_t n = 0;
    std::vector<ChunkPart> part;
} g_chunk;

meme_seed_opt seed_opt_of(const mem_opt_t* opt) {
    meme_seed_opt so;
    so.min_seed_len = opt->min_seed_len;
    so.split_len = (int)(opt->min_seed_len * opt->split_factor + .499);   // src/bwamem.cpp:1348
    so.split_width = opt->split_width;
    so.max_mem_intv = opt->max_mem_intv;
    so.rounds = 3;
    so.hits_per_smem = 0;
    return so;
}

void seed_part(int d, const mem_opt_t* opt, bseq1_t* seqs, ChunkPart& P) {
    if (P.count + 1 > P.off_cap) { meme_host_free(P.off); P.off_cap = P.count + P.count / 4 + 64; if (!(P.off = (int64_t*)meme_host_alloc(P.off_cap * 8))) die("meme_host_alloc"); }
    int64_t bytes = 0;
    for (int64_t i = 0; i < P.count; ++i) { P.off[i] = bytes; bytes += seqs[P.first + i].l_seq; }
    P.off[P.count] = bytes;
    if (bytes + 16 > P.flat_cap) { meme_host_free(P.flat); P.flat_cap = bytes + bytes / 4 + 4096; if (!(P.flat = (uint8_t*)meme_host_alloc(P.flat_cap))) die("meme_host_alloc"); }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P.count; ++i) {
        // base codes in place, as the reference leaves them for the later stages (src/bwamem.cpp:1277-1279)
        bseq1_t& s = seqs[P.first + i];
        uint8_t* dst = P.flat + P.off[i];
        for (int k = 0; k < s.l_seq; ++k) { const char c = s.seq[k]; s.seq[k] = c < 4 ? c : (char)nst_nt4_table[(int)c]; dst[k] = (uint8_t)s.seq[k]; }
    }
    const meme_seed_opt so = seed_opt_of(opt);
    if (meme_seed_batch_host(g_dev[(size_t)d].seed, P.flat, P.off, P.count, &so, &P.res)) die("meme_seed_batch_host");
}

void seed_chunk(const mem_opt_t* opt, bseq1_t* seqs, int64_t n) {
    const double t0 = now_s();
    const int nd = (int)g_dev.size();
    g_chunk.seqs = seqs;
    g_chunk.n = n;
    if (g_chunk.part.size() != (size_t)nd) g_chunk.part.resize((size_t)nd);
    // consecutive 512-read batches of the chunk go to consecutive GPUs: contiguous ranges, batch-aligned
    const int64_t nb = (n + BATCH_SIZE - 1) / BATCH_SIZE;
    std::vector<std::thread> th;
    for (int d = 0; d < nd; ++d) {
        ChunkPart& P = g_chunk.part[(size_t)d];
        const int64_t b0 = nb * d / nd, b1 = nb * (d + 1) / nd;
        P.first = b0 * BATCH_SIZE;
        P.count = (b1 * BATCH_SIZE < n ? b1 * BATCH_SIZE : n) - P.first;
        if (P.count < 0) P.count = 0;
        if (d + 1 < nd) th.emplace_back(seed_part, d, opt, seqs, std::ref(P));
        else seed_part(d, opt, seqs, P);
    }
    for (auto& t : th) t.join();
    g_t_seed = g_t_seed + (now_s() - t0);
    g_n_seed_reads += n;
    if (verbose()) fprintf(stderr, "[meme-dropin] chunk of %lld reads seeded on %d GPU(s) in %.3f s\n", (long long)n, nd, now_s() - t0);
}

int g_team = 1;                        // kt_for worker threads of the run (opt->n_threads)

typedef void (*process_fn)(mem_opt_t*, int64_t, int, bseq1_t*, const mem_pestat_t*, worker_t&);

}  // namespace

void mem_process_seqs(mem_opt_t* opt, int64_t n_processed, int n, bseq1_t* seqs, const mem_pestat_t* pes0, worker_t& w) {
    static process_fn next = nullptr;
    if (!next) {
        next = (process_fn)dlsym(RTLD_NEXT, "_Z16mem_process_seqsP9mem_opt_tliP7bseq1_tPK12mem_pestat_tR8worker_t");
        if (!next) { fprintf(stderr, "[meme-dropin] the reference's mem_process_seqs was not found: %s\n", dlerror()); exit(1); }
    }
    g_team = opt->n_threads > 0 ? opt->n_threads : 1;
    if (w.useLearned) seed_chunk(opt, seqs, n);
    next(opt, n_processed, n, seqs, pes0, w);
    g_chunk.seqs = nullptr;
    if (verbose())
        fprintf(stderr, "[meme-dropin] totals: seeding %.3f s for %lld reads; bsw %lld calls, %lld pairs "
                "(copy-in thread-seconds %.3f, backend calls %.3f s of which kernels %.3f s)\n",
                (double)g_t_seed, (long long)g_n_seed_reads, (long long)g_n_bsw_calls, (long long)g_n_bsw_pairs,
                (double)g_t_bsw_gather, (double)g_t_bsw_call, (double)g_t_bsw_kernel);
}

int mem_kernel1_core_Learned(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, bseq1_t* seq_, int nseq,
                             mem_chain_v* chain_ar, mem_seed_t* seedBuf, int64_t seedBufSize, uint8_t* sa_pos,
                             uint8_t* ref2sa, uint8_t* ref_string, mem_tlv* smems, u64v* hits, int tid) {
    (void)sa_pos; (void)ref2sa; (void)ref_string;
    static_assert(sizeof(meme_mem_tl) == sizeof(mem_tl), "mem_tl layout");
    const int64_t g0 = seq_ - g_chunk.seqs;                   // this batch's position in the chunk seeded above
    if (!g_chunk.seqs || g0 < 0 || g0 + nseq > g_chunk.n) { fprintf(stderr, "[meme-dropin] batch outside the seeded chunk\n"); exit(1); }
    int64_t seedBufCount = 0;
    for (int l = 0; l < nseq; ++l) {
        const int64_t g = g0 + l;
        const ChunkPart* P = nullptr;
        for (const ChunkPart& c : g_chunk.part) if (g >= c.first && g < c.first + c.count) { P = &c; break; }
        const int64_t r = g - P->first;
        const int64_t s0 = P->res.smem_off[r], ns = P->res.smem_off[r + 1] - s0;
        const int64_t h0 = P->res.hit_off[r], nh = P->res.hit_off[r + 1] - h0;
        smems->n = 0;
        hits->n = 0;
        if ((int64_t)smems->m < ns) kv_resize(mem_tl, *smems, (size_t)ns);
        if ((int64_t)hits->m < nh) kv_resize(uint64_t, *hits, (size_t)nh);
        if (ns) memcpy(smems->a, P->res.smems + s0, (size_t)ns * sizeof(mem_tl));
        if (nh) memcpy(hits->a, P->res.hits + h0, (size_t)nh * sizeof(uint64_t));
        smems->n = (size_t)ns;
        hits->n = (size_t)nh;
        ks_introsort(meme_dropin_smem, smems->n, smems->a);            // src/bwamem.cpp:1397
        kv_init(chain_ar[l]);
        mem_chain_Learned(opt, bns, seq_[l].l_seq, smems, &chain_ar[l], l, hits, seedBuf, seedBufSize, seedBufCount, tid);
        mem_chain_v* chn = &chain_ar[l];
        chn->n = mem_chain_flt(opt, chn->n, chn->a, tid);
        mem_flt_chained_seeds(opt, bns, pac, seq_, chn->n, chn->a);
    }
    return 1;
}

// ---- banded SW: the three entry points of the reference class forward to the HIP batch call -------------------------------
namespace {

struct BswReq {
    SeqPair* pairs; const uint8_t* ref; const uint8_t* qer; int n; int w; meme_bsw_opt o; int64_t rb, qb;
    int64_t pn = 0, pr = 0, pq = 0;      // where this request sits in the staging buffers
    bool done = false;
};

// Group commit with double-buffered pinned staging.  A worker reserves room for its request in the open staging buffer
// (a short critical section), copies its pairs and sequences in by itself (all workers copy in parallel) and waits; the
// first waiter that finds no call in flight becomes the leader: it closes the buffer, lets new arrivals fill the other
// one, issues ONE backend call for everything in it and publishes the batch's epoch; the owners then copy their own
// results out.  While a call is in flight the next batch assembles itself.  Waiting is spin + yield on atomics (the
// workers have nothing else to do, and a condition-variable broadcast to 256 threads costs more than a backend call).
// One combiner per GPU.
struct Staging {
    meme_seqpair* pairs = nullptr; uint8_t* ref = nullptr; uint8_t* qer = nullptr;     // pinned, fixed capacity
    int64_t n = 0, rb = 0, qb = 0;                     // reserved so far          (under Combiner::m)
    int w = 0; meme_bsw_opt o; bool has_key = false;   // band / penalties of the batch
    bool closed = false;                               // no more reservations: being executed or drained
    uint64_t epoch = 1;                                // number of the batch being assembled
    std::atomic<int> nreq{0}, copying{0}, reading{0};
    std::atomic<uint64_t> done_epoch{0};
};

inline void backoff(unsigned& spins) {
    ++spins;
    if (spins < 64) { __builtin_ia32_pause(); return; }
    if ((spins & 15) != 0) { for (int k = 0; k < 16; ++k) __builtin_ia32_pause(); return; }
    if (spins < 4096) sched_yield();
    else { struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }
}

struct Combiner {
    static constexpr int64_t CAP_PAIRS = 1 << 20, CAP_REF = 384ll << 20, CAP_QER = 192ll << 20;
    std::mutex m, ctx_mu;
    Staging st[2];
    int open = 0;                                       // under m
    std::atomic<bool> busy{false};
    int device = 0;

    void init() {
        for (Staging& S : st) {
            S.pairs = (meme_seqpair*)meme_host_alloc(CAP_PAIRS * (int64_t)sizeof(meme_seqpair));
            S.ref = (uint8_t*)meme_host_alloc(CAP_REF + 64);
            S.qer = (uint8_t*)meme_host_alloc(CAP_QER + 64);
            if (!S.pairs || !S.ref || !S.qer) die("meme_host_alloc");
        }
    }
    static bool same_key(const Staging& S, const BswReq* r) { return S.w == r->w && !memcmp(&S.o, &r->o, sizeof(meme_bsw_opt)); }
    static bool fits(const Staging& S, const BswReq* r) {
        return S.n + r->n <= CAP_PAIRS && S.rb + r->rb <= CAP_REF && S.qb + r->qb <= CAP_QER;
    }

    // become the leader if nobody is, and run the batch that is being assembled.  `mine` / `my_epoch`: the caller's own
    // request, if it has one -- a thread whose batch has just been completed must first take its results out (the leader of
    // the next batch waits for exactly that before it can reuse the buffer), so it does not lead.
    void try_lead(int expected, const Staging* mine = nullptr, uint64_t my_epoch = 0) {
        bool f = false;
        if (!busy.compare_exchange_strong(f, true, std::memory_order_acquire)) return;
        if (mine && mine->done_epoch.load(std::memory_order_acquire) >= my_epoch) { busy.store(false, std::memory_order_release); return; }
        Staging* S;
        {
            std::lock_guard<std::mutex> lk(m);
            S = &st[open];
            if (S->closed || S->nreq.load() == 0) { busy.store(false, std::memory_order_release); return; }
        }
        // a moment for the rest of the team to join
        const double t0 = now_s();
        unsigned sp = 0;
        while (S->nreq.load(std::memory_order_relaxed) < expected && now_s() - t0 < 60e-6) backoff(sp);
        { std::lock_guard<std::mutex> lk(m); S->closed = true; }
        Staging* other = &st[S == &st[0] ? 1 : 0];
        for (sp = 0;;) {                                 // the other buffer is free once its previous owners have drained it
            {
                std::lock_guard<std::mutex> lk(m);
                if (!other->closed) { open = S == &st[0] ? 1 : 0; break; }
            }
            backoff(sp);
        }
        for (sp = 0; S->copying.load(std::memory_order_acquire) > 0;) backoff(sp);
        const double t1 = now_s();
        {
            std::lock_guard<std::mutex> cl(ctx_mu);
            if (meme_bsw_batch(g_dev[(size_t)device].bsw, S->pairs, S->ref, S->rb, S->qer, S->qb, (int)S->n, S->w, &S->o)) die("meme_bsw_batch");
            if (verbose()) { meme_timings tm; if (!meme_get_timings(g_dev[(size_t)device].bsw, &tm)) g_t_bsw_kernel = g_t_bsw_kernel + tm.bsw_kernel_ms * 1e-3; }
        }
        g_t_bsw_call = g_t_bsw_call + (now_s() - t1);
        g_n_bsw_calls += 1;
        g_n_bsw_pairs += S->n;
        S->reading.store(S->nreq.load(), std::memory_order_relaxed);
        S->done_epoch.store(S->epoch, std::memory_order_release);
        busy.store(false, std::memory_order_release);
    }

    void submit(BswReq* r, int expected) {
        if (r->n > CAP_PAIRS || r->rb > CAP_REF || r->qb > CAP_QER) {      // a request bigger than the staging area: on its own
            std::lock_guard<std::mutex> cl(ctx_mu);
            if (meme_bsw_batch(g_dev[(size_t)device].bsw, (meme_seqpair*)r->pairs, r->ref, r->rb, r->qer, r->qb, r->n, r->w, &r->o)) die("meme_bsw_batch");
            return;
        }
        Staging* S = nullptr;
        uint64_t my_epoch = 0;
        for (unsigned sp = 0;;) {
            {
                std::lock_guard<std::mutex> lk(m);
                S = &st[open];
                if (!S->closed && (!S->has_key || same_key(*S, r)) && fits(*S, r)) {
                    if (!S->has_key) { S->w = r->w; S->o = r->o; S->has_key = true; }
                    r->pn = S->n; r->pr = S->rb; r->pq = S->qb;
                    S->n += r->n; S->rb += r->rb; S->qb += r->qb;
                    S->copying.fetch_add(1, std::memory_order_relaxed);
                    S->nreq.fetch_add(1, std::memory_order_relaxed);
                    my_epoch = S->epoch;
                    break;
                }
            }
            try_lead(0);                                 // flush what blocks the way (other band / penalties, or full)
            backoff(sp);
        }
        const double t0 = now_s();
        memcpy(S->ref + r->pr, r->ref, (size_t)r->rb);
        memcpy(S->qer + r->pq, r->qer, (size_t)r->qb);
        for (int i = 0; i < r->n; ++i) {
            meme_seqpair p;
            memcpy(&p, &r->pairs[i], sizeof(p));
            p.idr += (int32_t)r->pr; p.idq += (int32_t)r->pq;
            S->pairs[r->pn + i] = p;
        }
        g_t_bsw_gather = g_t_bsw_gather + (now_s() - t0);
        S->copying.fetch_sub(1, std::memory_order_release);
        for (unsigned sp = 0; S->done_epoch.load(std::memory_order_acquire) < my_epoch;) {
            if (!busy.load(std::memory_order_relaxed)) try_lead(expected, S, my_epoch);
            backoff(sp);
        }
        for (int i = 0; i < r->n; ++i) {
            const meme_seqpair& g = S->pairs[r->pn + i];
            SeqPair& p = r->pairs[i];
            p.score = g.score; p.tle = g.tle; p.gtle = g.gtle; p.qle = g.qle; p.gscore = g.gscore; p.max_off = g.max_off;
        }
        if (S->reading.fetch_sub(1, std::memory_order_acq_rel) == 1) {      // last owner out: the buffer can be filled again
            std::lock_guard<std::mutex> lk(m);
            S->n = S->rb = S->qb = 0; S->has_key = false; S->nreq.store(0); S->epoch += 1; S->closed = false;
        }
    }
};

Combiner* g_comb = nullptr;
std::once_flag g_comb_once;
std::atomic<int> g_thread_seq{0};

void bsw_forward(const int8_t* mat, int o_del, int e_del, int o_ins, int e_ins, int zdrop, int end_bonus, SeqPair* pairs,
                 uint8_t* ref, uint8_t* qer, int n, int w) {
    if (n <= 0) return;
    static_assert(sizeof(meme_seqpair) == sizeof(SeqPair), "SeqPair layout");
    if (g_dev.empty()) { fprintf(stderr, "[meme-dropin] banded SW called before the devices were set up\n"); exit(1); }
    std::call_once(g_comb_once, [] {
        g_comb = new Combiner[g_dev.size()];
        for (size_t d = 0; d < g_dev.size(); ++d) { g_comb[d].device = (int)d; g_comb[d].init(); }
    });
    BswReq rq;
    rq.pairs = pairs; rq.ref = ref; rq.qer = qer; rq.n = n; rq.w = w;
    memset(&rq.o, 0, sizeof(rq.o));
    rq.o.o_del = o_del; rq.o.e_del = e_del; rq.o.o_ins = o_ins; rq.o.e_ins = e_ins; rq.o.zdrop = zdrop; rq.o.end_bonus = end_bonus;
    rq.o.a = mat[0]; rq.o.b = -mat[1];                        // mat = bwa_fill_scmat(a, b)
    rq.rb = rq.qb = 0;
    for (int i = 0; i < n; ++i) {
        if ((int64_t)pairs[i].idr + pairs[i].len1 > rq.rb) rq.rb = (int64_t)pairs[i].idr + pairs[i].len1;
        if ((int64_t)pairs[i].idq + pairs[i].len2 > rq.qb) rq.qb = (int64_t)pairs[i].idq + pairs[i].len2;
    }
    // the kt_for thread id is not passed down to this level: number the calling threads as they show up
    static thread_local int my = g_thread_seq++;
    const int nd = (int)g_dev.size();
    const int per = (g_team + nd - 1) / nd;
    g_comb[my % nd].submit(&rq, per);
}
}  // namespace

void BandedPairWiseSW::scalarBandedSWAWrapper(SeqPair* p, uint8_t* r, uint8_t* q, int n, int nthreads, int32_t w) {
    (void)nthreads;
    bsw_forward(mat, o_del, e_del, o_ins, e_ins, zdrop, end_bonus, p, r, q, n, w);
}
void BandedPairWiseSW::getScores16(SeqPair* p, uint8_t* r, uint8_t* q, int32_t n, uint16_t nthreads, int32_t w) {
    (void)nthreads;
    bsw_forward(mat, o_del, e_del, o_ins, e_ins, zdrop, end_bonus, p, r, q, n, w);
}
void BandedPairWiseSW::getScores8(SeqPair* p, uint8_t* r, uint8_t* q, int32_t n, uint16_t nthreads, int32_t w) {
    (void)nthreads;
    bsw_forward(mat, o_del, e_del, o_ins, e_ins, zdrop, end_bonus, p, r, q, n, w);
}
