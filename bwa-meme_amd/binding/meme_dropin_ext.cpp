// Part of the reference-side binding of the MI355X backend (see meme_dropin.h / meme_dropin.cpp).
#include "meme_dropin.h"

using namespace dropin;

// ---- banded SW: the three entry points of the reference class forward to the HIP batch call -------------------------------
namespace dropin {

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
}  // namespace dropin

// ---- seed extension: mem_chain2aln_across_reads_V2 (src/bwamem.cpp:2573-3497) ----------------------------------------------------------
// The reference's function takes one 512-read batch: it creates a left and a right extension job per chained seed, runs them class by
// class (8-bit / 16-bit / scalar lanes, two band widths each -- up to twelve BandedPairWiseSW calls of a few hundred pairs) and finally
// purges the alignments of seeds that an earlier alignment of the read already covers.  With the backend bound all of that -- and the
// seed filter before it, mem_flt_chained_seeds -- has run on the device for the whole -K chunk before the first batch gets here
// (meme_extend_last_batch_host in seed_part): a batch's call only takes its reads' alignment records.  MEME_DROPIN_EXT=0 keeps the
// reference's function (its BandedPairWiseSW calls then go through the combiner above): the cross-check.
namespace dropin {
typedef void (*chain2aln_fn)(const mem_opt_t*, const bntseq_t*, const uint8_t*, bseq1_t*, int, mem_chain_v*, mem_alnreg_v*, mem_cache*, uint8_t*, int);
}  // namespace dropin

void mem_chain2aln_across_reads_V2(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, bseq1_t* seq_, int nseq,
                                   mem_chain_v* chain_ar, mem_alnreg_v* av_v, mem_cache* mmc, uint8_t* ref_string, int tid) {
    if (ext_mode() == 0 || !g_chunk.seqs) {
        static const chain2aln_fn next = (chain2aln_fn)ref_sym(R_CHAIN2ALN_V2);
        next(opt, bns, pac, seq_, nseq, chain_ar, av_v, mmc, ref_string, tid);
        return;
    }
    const int64_t g0 = seq_ - g_chunk.seqs;
    if (g0 < 0 || g0 + nseq > g_chunk.n) { fprintf(stderr, "[meme-dropin] batch outside the chunk\n"); exit(1); }
    if (g_ext_on_device) {                                 // the records the device made of this batch's reads (:2633: the reference owns them from here on)
        static_assert(sizeof(meme_alnreg) == sizeof(mem_alnreg_t), "mem_alnreg_t layout");
        static_assert(offsetof(meme_alnreg, c) == offsetof(mem_alnreg_t, c) && offsetof(meme_alnreg, score) == offsetof(mem_alnreg_t, score) &&
                      offsetof(meme_alnreg, seedlen0) == offsetof(mem_alnreg_t, seedlen0) && offsetof(meme_alnreg, frac_rep) == offsetof(mem_alnreg_t, frac_rep) &&
                      offsetof(meme_alnreg, hash) == offsetof(mem_alnreg_t, hash) && offsetof(meme_alnreg, flg) == offsetof(mem_alnreg_t, flg), "mem_alnreg_t layout");
        for (int l = 0; l < nseq; ++l) {
            const int64_t g = g0 + l;
            const ChunkPart* P = nullptr;
            for (const ChunkPart& c : g_chunk.part) if (g >= c.first && g < c.first + c.count) { P = &c; break; }
            if (!P || !P->has_ext) { fprintf(stderr, "[meme-dropin] no alignment records for a read of the chunk\n"); exit(1); }
            const int64_t r = g - P->first, b = P->ext.reg_off[r], m = P->ext.reg_off[r + 1] - b;
            mem_alnreg_t* a = (mem_alnreg_t*)calloc((size_t)m, sizeof(mem_alnreg_t));
            if (m) {
                if (!a) { fprintf(stderr, "[meme-dropin] out of memory\n"); exit(1); }
                memcpy(a, P->ext.regs + b, (size_t)m * sizeof(mem_alnreg_t));
                for (int64_t i = 0; i < m; ++i) a[i].c = nullptr;                  // (held the chain's index; dead after the stage)
            }
            av_v[l].n = (size_t)m; av_v[l].m = (size_t)m; av_v[l].a = a;
        }
        return;
    }
    fprintf(stderr, "[meme-dropin] mem_chain2aln_across_reads_V2: no device records for this run\n");
    exit(1);
}

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
