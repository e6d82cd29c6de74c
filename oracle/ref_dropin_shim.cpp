// TEST INFRASTRUCTURE -- the reference-side binding of the MI355X backend, exactly what INTEGRATION.md
// describes, written against the *reference's own headers* and linked into a copy of the reference aligner
// (oracle/Makefile.ref target bwa-meme_dropin) so that the end-to-end test can diff SAM files:
//
//   mem_kernel1_core_Learned()                      (reference src/bwamem.cpp:1230-1413)
//       per-read seeding loop  ->  meme_seed_batch(); everything after seeding (ks_introsort, mem_chain_Learned,
//       mem_chain_flt, mem_flt_chained_seeds) is the reference's own code.
//   BandedPairWiseSW::getScores8 / getScores16 / scalarBandedSWAWrapper   (src/bandedSWA.cpp:242-260,1970-,2664-)
//       ->  meme_bsw_batch()
// The reference calls both from n_threads workers with 512 reads / a few thousand pairs at a time; a GPU call of
// that size is all latency.  The binding therefore *combines* concurrent calls: the first worker to arrive becomes
// the leader, waits a moment for the others, issues ONE backend call for everybody and hands the slices back
// (Combiner below).  MEME_DROPIN_COMBINE=0 keeps one backend call per reference call.
//
// The reference objects are built position-independent into libbwa_pic.so; these definitions live in the
// executable and therefore win symbol resolution (ELF interposition) -- no reference source is modified or
// copied.  Contains no reference code: the calls below go to functions the reference exports.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <tuple>
#include <vector>

#include "bwamem.h"              // reference headers (-I$(REF)/src)
#include "LearnedIndex_seeding.h"
#include "bandedSWA.h"
#include "ksort.h"

#include "meme_hip.h"            // our C ABI (-Iinclude)

// reference functions used unchanged
void mem_chain_Learned(const mem_opt_t* opt, const bntseq_t* bns, int len, mem_tlv* smems, mem_chain_v* chain,
                       int seqid, u64v* hits, mem_seed_t* seedBuf, int64_t seedBufSize, int64_t& seedBufCount, int tid);
int mem_chain_flt(const mem_opt_t* opt, int n_chn_, mem_chain_t* a_, int tid);
void mem_flt_chained_seeds(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, bseq1_t* seq_, int n_chn,
                           mem_chain_t* a);
extern uint64_t tprof[LIM_R][LIM_C];

#define dropin_smem_lt(a, b) ((a).start == (b).start ? (a).end < (b).end : (a).start < (b).start)
KSORT_INIT(meme_dropin_smem, mem_tl, dropin_smem_lt)

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#define TRACE(...) do { if (getenv("MEME_DROPIN_TRACE")) { fprintf(stderr, "[meme-dropin] " __VA_ARGS__); fputc('\n', stderr); } } while (0)

namespace {

void segv_handler(int sig) {
    void* bt[64];
    int n = backtrace(bt, 64);
    fprintf(stderr, "[meme-dropin] signal %d, backtrace:\n", sig);
    backtrace_symbols_fd(bt, n, 2);
    _exit(139);
}

std::mutex g_mu;
meme_ctx* g_ctx[1024];
meme_ctx* g_owner = nullptr;
int g_nthreads = 1;

meme_ctx* new_ctx_locked(bool with_index) {
    meme_ctx* c = meme_ctx_create(0);
    if (!c) { fprintf(stderr, "[meme-dropin] %s\n", meme_last_error()); exit(1); }
    if (!with_index) return c;
    if (!g_owner) {
        const char* prefix = getenv("MEME_INDEX_PREFIX");
        if (!prefix) { fprintf(stderr, "[meme-dropin] set MEME_INDEX_PREFIX to the index prefix\n"); exit(1); }
        if (meme_index_load_files(c, prefix)) { fprintf(stderr, "[meme-dropin] %s\n", meme_last_error()); exit(1); }
        g_owner = c;
    } else if (meme_index_share(c, g_owner)) { fprintf(stderr, "[meme-dropin] %s\n", meme_last_error()); exit(1); }
    return c;
}

meme_ctx* ctx_for(int tid) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (tid < 0 || tid >= 1024) { fprintf(stderr, "[meme-dropin] bad tid %d\n", tid); exit(1); }
    if (!g_ctx[tid]) g_ctx[tid] = new_ctx_locked(true);
    return g_ctx[tid];
}

bool combine_enabled() {
    static const bool on = !(getenv("MEME_DROPIN_COMBINE") && atoi(getenv("MEME_DROPIN_COMBINE")) == 0);
    return on;
}

// Combines concurrent requests of the reference's worker threads into one backend call.
template <class Req>
struct Combiner {
    std::mutex m;
    std::condition_variable cv;
    std::vector<Req*> queue;
    bool busy = false;

    template <class Exec>
    void submit(Req* r, int expected, Exec&& exec) {
        std::unique_lock<std::mutex> lk(m);
        queue.push_back(r);
        cv.notify_all();
        for (;;) {
            if (r->done) return;
            if (busy) { cv.wait(lk); continue; }
            busy = true;                                   // leader: wait a moment for the other workers
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
            while ((int)queue.size() < expected && cv.wait_until(lk, deadline) != std::cv_status::timeout) {}
            std::vector<Req*> batch;
            batch.swap(queue);
            lk.unlock();
            exec(batch);
            lk.lock();
            for (Req* b : batch) b->done = true;
            busy = false;
            cv.notify_all();
        }
    }
};

struct SeedReq {
    const uint8_t* flat; const int64_t* off; int nseq;
    std::vector<meme_mem_tl> sm; std::vector<uint64_t> ht; std::vector<int64_t> smo, hto;
    bool done = false;
};

void seed_call(meme_ctx* ctx, const meme_seed_opt& so, const uint8_t* flat, const int64_t* off, int64_t nseq,
               std::vector<meme_mem_tl>& sm, std::vector<uint64_t>& ht, std::vector<int64_t>& smo, std::vector<int64_t>& hto) {
    // output buffers only ever grow (callers keep them across calls): a fresh 1024-hits-per-read buffer for a combined
    // call of 32 k reads would be 268 MB of page faults per call
    if (sm.size() < (size_t)nseq * 16 + 1024) sm.resize((size_t)nseq * 16 + 1024);
    if (ht.size() < (size_t)nseq * 32 + 65536) ht.resize((size_t)nseq * 32 + 65536);
    smo.resize((size_t)nseq + 1); hto.resize((size_t)nseq + 1);
    int64_t ts = 0, th = 0;
    for (;;) {
        int rc = meme_seed_batch(ctx, flat, off, nseq, &so, sm.data(), (int64_t)sm.size(), smo.data(),
                                 ht.data(), (int64_t)ht.size(), hto.data(), &ts, &th);
        if (rc == MEME_E_CAPACITY) { sm.resize((size_t)(ts + ts / 4) + 1); ht.resize((size_t)(th + th / 4) + 1); continue; }
        if (rc) { fprintf(stderr, "[meme-dropin] meme_seed_batch: %s\n", meme_last_error()); exit(1); }
        break;
    }
}

Combiner<SeedReq> g_seed_comb;
meme_seed_opt g_seed_opt;

void seed_exec(std::vector<SeedReq*>& batch) {
    static meme_ctx* ctx = nullptr;                        // only the leader runs here, one at a time
    if (!ctx) { std::lock_guard<std::mutex> lk(g_mu); ctx = new_ctx_locked(true); }
    int64_t nseq = 0, bytes = 0;
    for (SeedReq* r : batch) { nseq += r->nseq; bytes += r->off[r->nseq]; }
    std::vector<uint8_t> flat((size_t)bytes + 1);
    std::vector<int64_t> off((size_t)nseq + 1, 0);
    int64_t q = 0, b = 0;
    for (SeedReq* r : batch) {
        memcpy(&flat[(size_t)b], r->flat, (size_t)r->off[r->nseq]);
        for (int l = 0; l < r->nseq; ++l) off[(size_t)(q + l + 1)] = b + r->off[l + 1];
        q += r->nseq; b += r->off[r->nseq];
    }
    static std::vector<meme_mem_tl> sm;                      // leader-only, reused across calls
    static std::vector<uint64_t> ht;
    static std::vector<int64_t> smo, hto;
    seed_call(ctx, g_seed_opt, flat.data(), off.data(), nseq, sm, ht, smo, hto);
    q = 0;
    for (SeedReq* r : batch) {                              // hand every worker its slice (hitbeg is per read already)
        const int64_t s0 = smo[(size_t)q], s1 = smo[(size_t)(q + r->nseq)], h0 = hto[(size_t)q], h1 = hto[(size_t)(q + r->nseq)];
        r->sm.assign(sm.begin() + s0, sm.begin() + s1);
        r->ht.assign(ht.begin() + h0, ht.begin() + h1);
        r->smo.resize((size_t)r->nseq + 1); r->hto.resize((size_t)r->nseq + 1);
        for (int l = 0; l <= r->nseq; ++l) { r->smo[(size_t)l] = smo[(size_t)(q + l)] - s0; r->hto[(size_t)l] = hto[(size_t)(q + l)] - h0; }
        q += r->nseq;
    }
}

}  // namespace

int mem_kernel1_core_Learned(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, bseq1_t* seq_, int nseq,
                             mem_chain_v* chain_ar, mem_seed_t* seedBuf, int64_t seedBufSize, uint8_t* sa_pos,
                             uint8_t* ref2sa, uint8_t* ref_string, mem_tlv* smems, u64v* hits, int tid) {
    (void)sa_pos; (void)ref2sa; (void)ref_string;
    if (getenv("MEME_DROPIN_TRACE")) signal(SIGSEGV, segv_handler);
    TRACE("seed batch tid=%d nseq=%d", tid, nseq);
    meme_ctx* ctx = combine_enabled() ? nullptr : ctx_for(tid);
    int64_t seedBufCount = 0;
    // base codes in place, as the reference leaves them for the later stages (src/bwamem.cpp:1277-1279)
    std::vector<int64_t> off((size_t)nseq + 1, 0);
    for (int l = 0; l < nseq; ++l) {
        char* s = seq_[l].seq;
        for (int i = 0; i < seq_[l].l_seq; ++i) s[i] = s[i] < 4 ? s[i] : nst_nt4_table[(int)s[i]];
        off[(size_t)l + 1] = off[(size_t)l] + seq_[l].l_seq;
    }
    std::vector<uint8_t> flat((size_t)off[(size_t)nseq] + 1);
    for (int l = 0; l < nseq; ++l) memcpy(&flat[(size_t)off[(size_t)l]], seq_[l].seq, (size_t)seq_[l].l_seq);
    meme_seed_opt so;
    so.min_seed_len = opt->min_seed_len;
    so.split_len = (int)(opt->min_seed_len * opt->split_factor + .499);   // src/bwamem.cpp:1348
    so.split_width = opt->split_width;
    so.max_mem_intv = opt->max_mem_intv;
    so.rounds = 3;
    so.hits_per_smem = 0;
    SeedReq rq;
    rq.flat = flat.data(); rq.off = off.data(); rq.nseq = nseq;
    if (combine_enabled()) {
        g_nthreads = opt->n_threads > 0 ? opt->n_threads : 1;
        g_seed_opt = so;                                    // identical for every worker of a run
        g_seed_comb.submit(&rq, g_nthreads, seed_exec);
    } else seed_call(ctx, so, flat.data(), off.data(), nseq, rq.sm, rq.ht, rq.smo, rq.hto);   // (fresh buffers per call: 512 reads)
    std::vector<meme_mem_tl>& sm = rq.sm;
    std::vector<uint64_t>& ht = rq.ht;
    std::vector<int64_t>&smo = rq.smo, &hto = rq.hto;
    const int64_t ts = (int64_t)sm.size(), th = (int64_t)ht.size();
    TRACE("seeded: %lld smems %lld hits", (long long)ts, (long long)th);
    static_assert(sizeof(meme_mem_tl) == sizeof(mem_tl), "mem_tl layout");
    for (int l = 0; l < nseq; ++l) {
        const int64_t ns = smo[(size_t)l + 1] - smo[(size_t)l], nh = hto[(size_t)l + 1] - hto[(size_t)l];
        smems->n = 0;
        hits->n = 0;
        if ((int64_t)smems->m < ns) kv_resize(mem_tl, *smems, (size_t)ns);
        if ((int64_t)hits->m < nh) kv_resize(uint64_t, *hits, (size_t)nh);
        if (ns) memcpy(smems->a, sm.data() + smo[(size_t)l], (size_t)ns * sizeof(mem_tl));
        if (nh) memcpy(hits->a, ht.data() + hto[(size_t)l], (size_t)nh * sizeof(uint64_t));
        smems->n = (size_t)ns;
        hits->n = (size_t)nh;
        ks_introsort(meme_dropin_smem, smems->n, smems->a);            // src/bwamem.cpp:1397
        kv_init(chain_ar[l]);
        mem_chain_Learned(opt, bns, seq_[l].l_seq, smems, &chain_ar[l], l, hits, seedBuf, seedBufSize, seedBufCount, tid);
        mem_chain_v* chn = &chain_ar[l];
        chn->n = mem_chain_flt(opt, chn->n, chn->a, tid);
        mem_flt_chained_seeds(opt, bns, pac, seq_, chn->n, chn->a);
    }
    TRACE("chained tid=%d", tid);
    return 1;
}

// ---- banded SW: the three entry points of the reference class forward to the HIP batch call ---------------
namespace {
struct BswReq {
    SeqPair* pairs; uint8_t* ref; uint8_t* qer; int n; int w; meme_bsw_opt o; int64_t rb, qb;
    bool done = false;
};
Combiner<BswReq> g_bsw_comb;

void bsw_call(meme_ctx* ctx, meme_seqpair* pairs, const uint8_t* ref, int64_t rb, const uint8_t* qer, int64_t qb, int n, int w,
              const meme_bsw_opt& o) {
    if (meme_bsw_batch(ctx, pairs, ref, rb, qer, qb, n, w, &o)) {
        fprintf(stderr, "[meme-dropin] meme_bsw_batch: %s\n", meme_last_error());
        exit(1);
    }
}

void bsw_exec(std::vector<BswReq*>& batch) {
    static meme_ctx* ctx = nullptr;
    if (!ctx) { std::lock_guard<std::mutex> lk(g_mu); ctx = new_ctx_locked(false); }
    std::vector<char> used(batch.size(), 0);
    for (size_t k0 = 0; k0 < batch.size(); ++k0) {
        if (used[k0]) continue;
        // requests with the same band and penalties share one call
        std::vector<size_t> grp;
        for (size_t k = k0; k < batch.size(); ++k)
            if (!used[k] && batch[k]->w == batch[k0]->w && !memcmp(&batch[k]->o, &batch[k0]->o, sizeof(meme_bsw_opt))) { grp.push_back(k); used[k] = 1; }
        int64_t n = 0, rb = 0, qb = 0;
        for (size_t k : grp) { n += batch[k]->n; rb += batch[k]->rb; qb += batch[k]->qb; }
        if (rb >= INT32_MAX || qb >= INT32_MAX || n >= INT32_MAX) {      // SeqPair offsets are 32-bit: fall back to one call each
            for (size_t k : grp) bsw_call(ctx, (meme_seqpair*)batch[k]->pairs, batch[k]->ref, batch[k]->rb, batch[k]->qer, batch[k]->qb, batch[k]->n, batch[k]->w, batch[k]->o);
            continue;
        }
        std::vector<meme_seqpair> all((size_t)n);
        std::vector<uint8_t> ref((size_t)rb + 1), qer((size_t)qb + 1);
        int64_t pn = 0, pr = 0, pq = 0;
        for (size_t k : grp) {
            BswReq* r = batch[k];
            memcpy(&ref[(size_t)pr], r->ref, (size_t)r->rb);
            memcpy(&qer[(size_t)pq], r->qer, (size_t)r->qb);
            memcpy(&all[(size_t)pn], r->pairs, (size_t)r->n * sizeof(meme_seqpair));
            for (int i = 0; i < r->n; ++i) { all[(size_t)(pn + i)].idr += (int32_t)pr; all[(size_t)(pn + i)].idq += (int32_t)pq; }
            pn += r->n; pr += r->rb; pq += r->qb;
        }
        bsw_call(ctx, all.data(), ref.data(), rb, qer.data(), qb, (int)n, batch[k0]->w, batch[k0]->o);
        pn = 0;
        for (size_t k : grp) {
            BswReq* r = batch[k];
            for (int i = 0; i < r->n; ++i) {
                const meme_seqpair& g = all[(size_t)(pn + i)];
                SeqPair& p = r->pairs[i];
                p.score = g.score; p.tle = g.tle; p.gtle = g.gtle; p.qle = g.qle; p.gscore = g.gscore; p.max_off = g.max_off;
            }
            pn += r->n;
        }
    }
}

void bsw_forward(const int8_t* mat, int o_del, int e_del, int o_ins, int e_ins, int zdrop, int end_bonus,
                 SeqPair* pairs, uint8_t* ref, uint8_t* qer, int n, int w) {
    TRACE("bsw n=%d w=%d", n, w);
    if (n <= 0) return;
    static_assert(sizeof(meme_seqpair) == sizeof(SeqPair), "SeqPair layout");
    BswReq rq;
    rq.pairs = pairs; rq.ref = ref; rq.qer = qer; rq.n = n; rq.w = w;
    memset(&rq.o, 0, sizeof(rq.o));
    rq.o.o_del = o_del; rq.o.e_del = e_del; rq.o.o_ins = o_ins; rq.o.e_ins = e_ins; rq.o.zdrop = zdrop; rq.o.end_bonus = end_bonus;
    rq.o.a = mat[0]; rq.o.b = -mat[1];
    rq.rb = rq.qb = 0;
    for (int i = 0; i < n; ++i) {
        if ((int64_t)pairs[i].idr + pairs[i].len1 > rq.rb) rq.rb = (int64_t)pairs[i].idr + pairs[i].len1;
        if ((int64_t)pairs[i].idq + pairs[i].len2 > rq.qb) rq.qb = (int64_t)pairs[i].idq + pairs[i].len2;
    }
    if (combine_enabled()) { g_bsw_comb.submit(&rq, g_nthreads, bsw_exec); return; }
    // tid is not passed down to this level: one private context per calling thread
    static thread_local meme_ctx* ctx = nullptr;
    if (!ctx) { std::lock_guard<std::mutex> lk(g_mu); ctx = new_ctx_locked(false); }
    bsw_call(ctx, (meme_seqpair*)pairs, ref, rq.rb, qer, rq.qb, n, w, rq.o);
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
