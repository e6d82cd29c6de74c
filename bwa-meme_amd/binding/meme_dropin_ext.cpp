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

// ---- chunk-wide seed extension: mem_chain2aln_across_reads_V2 (src/bwamem.cpp:2573-3497) ------------------------------
// The reference's function takes one 512-read batch: it creates a left and a right extension job per chained seed, runs them
// class by class (8-bit / 16-bit / scalar lanes, two band widths each -- up to twelve BandedPairWiseSW calls of a few hundred
// pairs) and finally purges the alignments of seeds that an earlier alignment of the read already covers.  The chains of the
// whole -K chunk exist when the first batch gets here (kt_for(worker_bwt) has finished, src/bwamem.cpp:1941-1945), so the
// binding computes the same function for ALL reads of the chunk, stage by stage: jobs of a slab of reads built by a thread
// team straight into pinned staging, ONE backend call per direction and band width, results folded back by the team.  A
// batch's call then only takes its reads' alignment arrays.  MEME_DROPIN_EXT=0 keeps the reference's function (its
// BandedPairWiseSW calls then go through the combiner above).
namespace dropin {

struct Team {                                   // persistent helper threads (the kt_for workers are parked on g_ext.mu meanwhile)
    struct Job {                                // one run(): a late waker that still holds an old Job finds it exhausted
        std::function<void(int64_t)> fn;
        int64_t n = 0;
        std::atomic<int64_t> next{0}, done{0};
    };
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_go;
    std::shared_ptr<Job> job;
    uint64_t gen = 0;
    static void work(Job& j) {
        for (int64_t i; (i = j.next.fetch_add(1, std::memory_order_relaxed)) < j.n;) { j.fn(i); j.done.fetch_add(1, std::memory_order_release); }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            std::shared_ptr<Job> j;
            { std::unique_lock<std::mutex> lk(m); cv_go.wait(lk, [&] { return gen != seen; }); seen = gen; j = job; }
            work(*j);
        }
    }
    void ensure(int nt) { while ((int)th.size() < nt) { th.emplace_back([this] { loop(); }); th.back().detach(); } }
    // returns when every item is done -- not when every helper has woken up: one descheduled thread must not hold up a stage
    void run(int64_t items, const std::function<void(int64_t)>& f) {
        if (items <= 0) return;
        auto j = std::make_shared<Job>();
        j->fn = f; j->n = items;
        { std::lock_guard<std::mutex> lk(m); job = j; ++gen; }
        cv_go.notify_all();
        work(*j);
        for (unsigned sp = 0; j->done.load(std::memory_order_acquire) < items;) backoff(sp);
    }
};

struct ExtStage {                               // pinned staging of one direction's jobs
    SeqPair* pairs = nullptr; uint8_t* ref = nullptr; uint8_t* qer = nullptr;
    int64_t cap_n = 0, cap_r = 0, cap_q = 0;
    int64_t n = 0, rb = 0, qb = 0;              // reserved; beyond the capacity = the slab is rebuilt with larger buffers
    struct Cut { int64_t n, rb, qb; };
    std::vector<Cut> cuts;                      // block boundaries: where a stage may be split over several GPUs
    std::mutex mu;
    void reset() { n = rb = qb = 0; cuts.clear(); }
    bool over() const { return n > cap_n || rb > cap_r || qb > cap_q; }
    void fit(int64_t want_n, int64_t want_r, int64_t want_q) {
        if (want_n > cap_n) { meme_host_free(pairs); cap_n = want_n + want_n / 4 + 1024; if (!(pairs = (SeqPair*)meme_host_alloc(cap_n * (int64_t)sizeof(SeqPair)))) die("meme_host_alloc"); }
        if (want_r > cap_r) { meme_host_free(ref); cap_r = want_r + want_r / 4 + 4096; if (!(ref = (uint8_t*)meme_host_alloc(cap_r + 64))) die("meme_host_alloc"); }
        if (want_q > cap_q) { meme_host_free(qer); cap_q = want_q + want_q / 4 + 4096; if (!(qer = (uint8_t*)meme_host_alloc(cap_q + 64))) die("meme_host_alloc"); }
        if (cap_r >= (1ll << 31) || cap_q >= (1ll << 31)) { fprintf(stderr, "[meme-dropin] extension staging beyond 2 GiB: lower MEME_DROPIN_EXT_SLAB\n"); exit(1); }
    }
    // room for dn pairs / dr + dq sequence bytes; false when the capacity is exceeded (the counters keep counting)
    bool reserve(int64_t dn, int64_t dr, int64_t dq, int64_t& pn, int64_t& pr, int64_t& pq) {
        std::lock_guard<std::mutex> lk(mu);
        pn = n; pr = rb; pq = qb;
        n += dn; rb += dr; qb += dq;
        cuts.push_back({n, rb, qb});
        return !over();
    }
};

struct ExtScratch {                             // a helper thread's jobs of one block of reads, offsets relative to the block
    std::vector<SeqPair> L, R;
    std::vector<uint8_t> Lr, Lq, Rr, Rq;
    std::vector<uint64_t> srt;
    std::vector<SeqPair> again;
    void clear() { L.clear(); R.clear(); Lr.clear(); Lq.clear(); Rr.clear(); Rq.clear(); }
};

constexpr int EXT_BLOCK = 128;                  // reads per work item
constexpr int EXT_BAND_TRIES = 2;               // MAX_BAND_TRY, src/bwamem.cpp:62

struct Ext {
    std::mutex mu;
    uint64_t gen = 0;                           // chunk whose alignments `av` holds
    std::vector<mem_alnreg_v> av;               // per read of the chunk; arrays pass to the reference batch by batch
    std::vector<std::vector<uint32_t>> order;   // per block: seed indices in extension order, chain after chain (srtgg)
    std::vector<int64_t> order_off;             // per read: where its seeds start in its block's `order`
    // alignment records of the chunk in ONE buffer kept across chunks (reg_off[g] = first record of read g).  A batch copies
    // its reads' records into calloc'ed arrays of its own when it takes them: the reference frees them one by one, and small
    // allocations made by the helper threads would grow 255 fresh malloc arenas page by page (2.7 s for a first chunk of 2 M reads).
    mem_alnreg_t* regs = nullptr;
    int64_t regs_cap = 0;
    std::vector<int64_t> reg_off;
    ExtStage L, R, X[2];
    std::vector<SeqPair> retry;                 // jobs of the stage just folded that need the next band width
    std::mutex retry_mu;
    Team team;
    double t_build = 0, t_call = 0, t_fold = 0, t_purge = 0, t_total = 0;
    int64_t n_calls = 0, n_pairs = 0, n_rebuilt = 0, n_retried = 0;
} *g_ext = nullptr;

inline int ext_max_gap(const mem_opt_t* opt, int qlen) {          // cal_max_gap, src/bwamem.cpp:85-95
    const int l_del = (int)((double)(qlen * opt->a - opt->o_del) / opt->e_del + 1.);
    const int l_ins = (int)((double)(qlen * opt->a - opt->o_ins) / opt->e_ins + 1.);
    int l = l_del > l_ins ? l_del : l_ins;
    l = l > 1 ? l : 1;
    return l < opt->w << 1 ? l : opt->w << 1;
}

inline void ext_seedcov(mem_alnreg_t* a) {                        // seeds of the chain fully inside the alignment (:2907-2917)
    if (a->rb == H0_ || a->qb == H0_ || a->qe == H0_ || a->re == H0_) return;
    int cov = 0;
    for (int i = 0; i < a->c->n; ++i) {
        const mem_seed_t* t = &a->c->seeds[i];
        if (t->qbeg >= a->qb && t->qbeg + t->len <= a->qe && t->rbeg >= a->rb && t->rbeg + t->len <= a->re) cov += t->len;
    }
    a->seedcov = cov;
}

// jobs of the reads [g_first, g_last) (src/bwamem.cpp:2612-2934)
void ext_build_block(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, const bseq1_t* seqs, mem_chain_v* chain_ar,
                     uint8_t* ref_string, int64_t slab0, int64_t blk, int64_t g_first, int64_t g_last) {
    static thread_local ExtScratch sc;
    sc.clear();
    Ext& E = *g_ext;
    std::vector<uint32_t>& order = E.order[(size_t)blk];
    order.clear();
    const int64_t l_pac = bns->l_pac;
    for (int64_t g = g_first; g < g_last; ++g) {
        const uint8_t* query = (const uint8_t*)seqs[g].seq;
        const int l_query = seqs[g].l_seq;
        mem_chain_v* chn = &chain_ar[g];
        mem_alnreg_v* av = &E.av[(size_t)g];
        av->m = (size_t)(E.reg_off[(size_t)g + 1] - E.reg_off[(size_t)g]);      // one record per chained seed
        av->n = 0;
        av->a = E.regs + E.reg_off[(size_t)g];
        if (av->m) memset(av->a, 0, av->m * sizeof(mem_alnreg_t));
        E.order_off[(size_t)g] = (int64_t)order.size();
        for (size_t j = 0; j < chn->n; ++j) {
            mem_chain_t* c = &chn->a[j];
            if (c->n == 0) continue;
            int64_t rmax0 = l_pac << 1, rmax1 = 0;                // the widest reference span any seed of the chain may reach
            for (int i = 0; i < c->n; ++i) {
                const mem_seed_t* t = &c->seeds[i];
                const int64_t b = t->rbeg - (t->qbeg + ext_max_gap(opt, t->qbeg));
                const int tail = l_query - t->qbeg - t->len;
                const int64_t e = t->rbeg + t->len + (tail + ext_max_gap(opt, tail));
                if (b < rmax0) rmax0 = b;
                if (e > rmax1) rmax1 = e;
            }
            if (rmax0 < 0) rmax0 = 0;
            if (rmax1 > l_pac << 1) rmax1 = l_pac << 1;
            if (rmax0 < l_pac && l_pac < rmax1) { if (c->seeds[0].rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac; }
            int rid = 0;
            const uint8_t* rseq = bns_fetch_seq_v2(bns, pac, &rmax0, c->seeds[0].rbeg, &rmax1, &rid, ref_string, nullptr);
            if (!rseq || rid != c->rid) { fprintf(stderr, "[meme-dropin] chain outside its reference sequence\n"); exit(1); }
            sc.srt.resize((size_t)c->n);
            for (int i = 0; i < c->n; ++i) sc.srt[(size_t)i] = (uint64_t)c->seeds[i].score << 32 | (uint32_t)i;
            if (c->n > 1) std::sort(sc.srt.begin(), sc.srt.end());             // keys are unique: any sort gives ks_introsort_64's order
            for (int i = 0; i < c->n; ++i) order.push_back((uint32_t)sc.srt[(size_t)i]);
            for (int k = c->n - 1; k >= 0; --k) {                              // best seed first
                mem_seed_t* s = &c->seeds[(uint32_t)sc.srt[(size_t)k]];
                mem_alnreg_t* a = &av->a[av->n++];                             // zeroed by calloc
                s->aln = (int)av->n - 1;
                a->w = opt->w;
                a->score = a->truesc = -1;
                a->rid = c->rid;
                a->frac_rep = c->frac_rep;
                a->seedlen0 = s->len;
                a->c = c;
                a->rb = a->qb = a->re = a->qe = H0_;
                if (s->qbeg) {                                                 // left of the seed: both sequences reversed
                    SeqPair sp;
                    memset(&sp, 0, sizeof(sp));
                    sp.h0 = s->len * opt->a;
                    sp.seqid = (int32_t)(g - slab0);
                    sp.regid = (int32_t)av->n - 1;
                    sp.len2 = s->qbeg;
                    sp.len1 = (int32_t)(s->rbeg - rmax0);
                    sp.idq = (int32_t)sc.Lq.size();
                    sp.idr = (int32_t)sc.Lr.size();
                    sc.Lq.resize(sc.Lq.size() + (size_t)sp.len2);
                    sc.Lr.resize(sc.Lr.size() + (size_t)sp.len1);
                    uint8_t* qs = sc.Lq.data() + sp.idq;
                    uint8_t* rs = sc.Lr.data() + sp.idr;
                    for (int i = 0; i < sp.len2; ++i) qs[i] = query[s->qbeg - 1 - i];
                    for (int i = 0; i < sp.len1; ++i) rs[i] = rseq[sp.len1 - 1 - i];
                    sc.L.push_back(sp);
                    a->qb = s->qbeg;
                    a->rb = s->rbeg;
                } else {
                    a->score = a->truesc = s->len * opt->a;
                    a->qb = 0;
                    a->rb = s->rbeg;
                }
                if (s->qbeg + s->len != l_query) {                             // right of the seed
                    const int qe = s->qbeg + s->len;
                    const int64_t re = s->rbeg + s->len - rmax0;
                    SeqPair sp;
                    memset(&sp, 0, sizeof(sp));
                    sp.h0 = H0_;                                               // the left extension's score, known after stage 1
                    sp.seqid = (int32_t)(g - slab0);
                    sp.regid = (int32_t)av->n - 1;
                    sp.len2 = l_query - qe;
                    sp.len1 = (int32_t)(rmax1 - rmax0 - re);
                    sp.idq = (int32_t)sc.Rq.size();
                    sp.idr = (int32_t)sc.Rr.size();
                    sc.Rq.insert(sc.Rq.end(), query + qe, query + qe + sp.len2);
                    sc.Rr.insert(sc.Rr.end(), rseq + re, rseq + re + sp.len1);
                    sc.R.push_back(sp);
                    a->qe = qe;
                    a->re = rmax0 + re;
                } else {
                    a->qe = l_query;
                    a->re = s->rbeg + s->len;
                    ext_seedcov(a);
                }
            }
        }
    }
    // hand the block's jobs to the slab's staging buffers
    struct Side { ExtStage* S; std::vector<SeqPair>* P; std::vector<uint8_t>* r; std::vector<uint8_t>* q; } side[2] = {
        {&E.L, &sc.L, &sc.Lr, &sc.Lq}, {&E.R, &sc.R, &sc.Rr, &sc.Rq}};
    for (Side& d : side) {
        int64_t pn, pr, pq;
        if (!d.S->reserve((int64_t)d.P->size(), (int64_t)d.r->size(), (int64_t)d.q->size(), pn, pr, pq)) continue;
        if (!d.r->empty()) memcpy(d.S->ref + pr, d.r->data(), d.r->size());
        if (!d.q->empty()) memcpy(d.S->qer + pq, d.q->data(), d.q->size());
        SeqPair* dst = d.S->pairs + pn;
        for (size_t i = 0; i < d.P->size(); ++i) { SeqPair sp = (*d.P)[i]; sp.idr += (int32_t)pr; sp.idq += (int32_t)pq; dst[i] = sp; }
    }
}

// one band width of one direction on the GPU(s): every job of the stage in one backend call per device
void ext_run_stage(ExtStage& S, int w, const meme_bsw_opt& o) {
    Ext& E = *g_ext;
    if (S.n == 0) return;
    const double t0 = now_s();
    static const int want_parts = getenv("MEME_DROPIN_EXT_SPLIT") ? atoi(getenv("MEME_DROPIN_EXT_SPLIT")) : 0;
    const int nd = (int)g_dev.size();
    int parts = want_parts > 0 ? want_parts : nd;
    if ((want_parts <= 0 && S.n < 65536 * (int64_t)parts) || S.cuts.size() < (size_t)parts) parts = 1;
    std::vector<ExtStage::Cut> at((size_t)parts + 1);
    at[0] = {0, 0, 0};
    at[(size_t)parts] = {S.n, S.rb, S.qb};
    for (int p = 1; p < parts; ++p) {                           // the block boundary closest to an even share of the pairs
        const int64_t want = S.n * p / parts;
        size_t lo = 0, hi = S.cuts.size() - 1;
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (S.cuts[mid].n < want) lo = mid + 1; else hi = mid; }
        at[(size_t)p] = S.cuts[lo];
    }
    static std::mutex* dev_mu = new std::mutex[64];
    auto one = [&](int p) {
        const ExtStage::Cut a = at[(size_t)p], b = at[(size_t)p + 1];
        const int64_t n = b.n - a.n;
        if (n <= 0) return;
        SeqPair* P = S.pairs + a.n;
        if (a.rb || a.qb) for (int64_t i = 0; i < n; ++i) { P[i].idr -= (int32_t)a.rb; P[i].idq -= (int32_t)a.qb; }
        {
            std::lock_guard<std::mutex> lk(dev_mu[p % nd]);
            if (meme_bsw_batch(g_dev[(size_t)(p % nd)].bsw, (meme_seqpair*)P, S.ref + a.rb, b.rb - a.rb, S.qer + a.qb, b.qb - a.qb, (int32_t)n, w, &o))
                die("meme_bsw_batch");
            if (verbose()) { meme_timings tm; if (!meme_get_timings(g_dev[(size_t)(p % nd)].bsw, &tm)) g_t_bsw_kernel = g_t_bsw_kernel + tm.bsw_kernel_ms * 1e-3; }
        }
        if (a.rb || a.qb) for (int64_t i = 0; i < n; ++i) { P[i].idr += (int32_t)a.rb; P[i].idq += (int32_t)a.qb; }
    };
    if (parts == 1) one(0);
    else {
        std::vector<std::thread> th;
        for (int p = 1; p < parts; ++p) th.emplace_back(one, p);
        one(0);
        for (auto& t : th) t.join();
    }
    E.t_call += now_s() - t0;
    g_t_bsw_call = g_t_bsw_call + (now_s() - t0);
    E.n_calls += parts;
    E.n_pairs += S.n;
    g_n_bsw_calls += parts;
    g_n_bsw_pairs += S.n;
}

// fold the results of one stage back into the alignments (src/bwamem.cpp:2985-3018 and its five siblings); jobs whose
// band was too narrow are collected in E.retry for the next band width
void ext_fold(const mem_opt_t* opt, const bseq1_t* seqs, int64_t slab0, ExtStage& S, bool left, int w, int attempt) {
    Ext& E = *g_ext;
    const double t0 = now_s();
    E.retry.clear();
    const int64_t step = 4096;
    const std::function<void(int64_t)> fold = [&](int64_t item) {
        static thread_local ExtScratch sc;
        sc.again.clear();
        const int64_t i0 = item * step, i1 = i0 + step < S.n ? i0 + step : S.n;
        for (int64_t i = i0; i < i1; ++i) {
            const SeqPair& sp = S.pairs[i];
            const int64_t g = slab0 + sp.seqid;
            mem_alnreg_t* a = &E.av[(size_t)g].a[sp.regid];
            const int prev = a->score;
            a->score = sp.score;
            if (a->score == prev || sp.max_off < (w >> 1) + (w >> 2) || attempt + 1 == EXT_BAND_TRIES) {
                if (left) {
                    if (sp.gscore <= 0 || sp.gscore <= a->score - opt->pen_clip5) { a->qb -= sp.qle; a->rb -= sp.tle; a->truesc = a->score; }
                    else { a->qb = 0; a->rb -= sp.gtle; a->truesc = sp.gscore; }
                } else {
                    if (sp.gscore <= 0 || sp.gscore <= a->score - opt->pen_clip3) { a->qe += sp.qle; a->re += sp.tle; a->truesc += a->score - sp.h0; }
                    else { a->qe = seqs[g].l_seq; a->re += sp.gtle; a->truesc += sp.gscore - sp.h0; }
                }
                a->w = a->w > w ? a->w : w;
                ext_seedcov(a);
            } else sc.again.push_back(sp);
        }
        if (!sc.again.empty()) {
            std::lock_guard<std::mutex> lk(E.retry_mu);
            E.retry.insert(E.retry.end(), sc.again.begin(), sc.again.end());
        }
    };
    E.team.run((S.n + step - 1) / step, fold);
    E.t_fold += now_s() - t0;
}

// the jobs in E.retry, with their sequences (still staged in `from`), as a stage of their own
void ext_stage_retry(const ExtStage& from, ExtStage& to) {
    Ext& E = *g_ext;
    to.reset();
    int64_t dr = 0, dq = 0;
    for (const SeqPair& sp : E.retry) { dr += sp.len1; dq += sp.len2; }
    to.fit((int64_t)E.retry.size(), dr, dq);
    for (SeqPair sp : E.retry) {
        memcpy(to.ref + to.rb, from.ref + sp.idr, (size_t)sp.len1);
        memcpy(to.qer + to.qb, from.qer + sp.idq, (size_t)sp.len2);
        sp.idr = (int32_t)to.rb; sp.idq = (int32_t)to.qb;
        to.rb += sp.len1; to.qb += sp.len2;
        to.pairs[to.n++] = sp;
    }
}

// alignments of seeds that an earlier (better-seeded) alignment of the read already explains are purged, in the order the
// one-read-at-a-time aligner would have met them (src/bwamem.cpp:3402-3491)
void ext_purge_read(const mem_opt_t* opt, const bseq1_t* seqs, mem_chain_v* chain_ar, int64_t g, uint32_t* order) {
    mem_alnreg_v* av = &g_ext->av[(size_t)g];
    mem_chain_v* chn = &chain_ar[g];
    const int l_query = seqs[g].l_seq;
    int kept = 0;
    for (size_t j = 0; j < chn->n; ++j) {
        mem_chain_t* c = &chn->a[j];
        uint32_t* ord = order;
        order += c->n;
        for (int k = c->n - 1; k >= 0; --k) {
            const mem_seed_t* s = &c->seeds[ord[k]];
            int v = 0;
            for (size_t i = 0; i < av->n && v < kept; ++i) {
                const mem_alnreg_t* p = &av->a[i];
                if (p->qb == -1 && p->qe == -1) continue;
                if (s->rbeg < p->rb || s->rbeg + s->len > p->re || s->qbeg < p->qb || s->qbeg + s->len > p->qe) { ++v; continue; }
                if (s->len - p->seedlen0 > .1 * l_query) { ++v; continue; }
                int qd = s->qbeg - p->qb;                                      // ahead of the seed
                int64_t rd = s->rbeg - p->rb;
                int max_gap = ext_max_gap(opt, qd < rd ? qd : (int)rd);
                int band = max_gap < p->w ? max_gap : p->w;
                if (qd - rd < band && rd - qd < band) break;
                qd = p->qe - (s->qbeg + s->len);                               // behind it
                rd = p->re - (s->rbeg + s->len);
                max_gap = ext_max_gap(opt, qd < rd ? qd : (int)rd);
                band = max_gap < p->w ? max_gap : p->w;
                if (qd - rd < band && rd - qd < band) break;
                ++v;
            }
            if (v < kept) {                                                    // (almost) contained -- unless a long overlapping seed says otherwise
                int u;
                for (u = k + 1; u < c->n; ++u) {
                    if (ord[u] == UINT32_MAX) continue;
                    const mem_seed_t* t = &c->seeds[ord[u]];
                    if (t->len < s->len * .95) continue;
                    if (s->qbeg <= t->qbeg && s->qbeg + s->len - t->qbeg >= s->len >> 2 && t->qbeg - s->qbeg != t->rbeg - s->rbeg) break;
                    if (t->qbeg <= s->qbeg && t->qbeg + t->len - s->qbeg >= s->len >> 2 && s->qbeg - t->qbeg != s->rbeg - t->rbeg) break;
                }
                if (u == c->n) {
                    mem_alnreg_t* ar = &av->a[s->aln];
                    ar->qb = ar->qe = -1;
                    ord[k] = UINT32_MAX;
                    continue;
                }
            }
            ++kept;
        }
    }
}

// helper threads of the extension stage: the aligner's thread count minus the caller (MEME_DROPIN_TEAM overrides)
int team_helpers(int threads) {
    static const int forced = getenv("MEME_DROPIN_TEAM") ? atoi(getenv("MEME_DROPIN_TEAM")) : -1;
    if (forced >= 0) return forced;
    return threads > 1 ? threads - 1 : 0;
}

int64_t ext_slab_reads() {
    static const int64_t v = getenv("MEME_DROPIN_EXT_SLAB") && atoll(getenv("MEME_DROPIN_EXT_SLAB")) > 0 ? atoll(getenv("MEME_DROPIN_EXT_SLAB")) : 262144;
    return v;
}
bool ext_enabled() { return ext_mode() != 0; }
// first guess of a slab's staging (3 jobs per read and direction; a job's target is the query side plus the gap allowance);
// a slab that needs more is rebuilt once with the exact sizes
void ext_size_for(int64_t reads, int64_t read_len) {
    Ext& E = *g_ext;
    static const bool undersize = getenv("MEME_DROPIN_EXT_UNDERSIZE") != nullptr;      // tests: force the rebuild path
    if (undersize) reads = reads / 16 + 1;
    for (ExtStage* S : {&E.L, &E.R}) S->fit(reads * 3, reads * 3 * (read_len + 64), reads * 2 * read_len);
}

void ext_chunk(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, const bseq1_t* seqs, int64_t n, mem_chain_v* chain_ar,
               uint8_t* ref_string) {
    Ext& E = *g_ext;
    const double t_begin = now_s();
    E.team.ensure(team_helpers(g_team));
    E.av.assign((size_t)n, mem_alnreg_v());
    for (mem_alnreg_v& v : E.av) memset(&v, 0, sizeof(v));
    E.order_off.assign((size_t)n, 0);
    E.reg_off.assign((size_t)n + 1, 0);
    {
        const int64_t step = 4096;
        const std::function<void(int64_t)> count = [&](int64_t item) {
            const int64_t g1 = (item + 1) * step < n ? (item + 1) * step : n;
            for (int64_t g = item * step; g < g1; ++g) {
                int64_t m = 0;
                for (size_t j = 0; j < chain_ar[g].n; ++j) m += chain_ar[g].a[j].n;
                E.reg_off[(size_t)g + 1] = m;
            }
        };
        E.team.run((n + step - 1) / step, count);
        for (int64_t g = 0; g < n; ++g) E.reg_off[(size_t)g + 1] += E.reg_off[(size_t)g];
        const int64_t total = E.reg_off[(size_t)n];
        if (total > E.regs_cap) {
            free(E.regs);
            E.regs_cap = total + total / 8 + 1024;
            if (!(E.regs = (mem_alnreg_t*)malloc((size_t)E.regs_cap * sizeof(mem_alnreg_t)))) { fprintf(stderr, "[meme-dropin] out of memory\n"); exit(1); }
        }
    }
    const int64_t slab_reads = ext_slab_reads();
    meme_bsw_opt ol, orr;
    memset(&ol, 0, sizeof(ol));
    ol.o_del = opt->o_del; ol.e_del = opt->e_del; ol.o_ins = opt->o_ins; ol.e_ins = opt->e_ins; ol.zdrop = opt->zdrop;
    ol.a = opt->a; ol.b = opt->b;
    orr = ol;
    ol.end_bonus = opt->pen_clip5;                                             // bswLeft / bswRight, src/bwamem.cpp:2953-2959
    orr.end_bonus = opt->pen_clip3;
    for (int64_t slab0 = 0; slab0 < n; slab0 += slab_reads) {
        const int64_t ns = n - slab0 < slab_reads ? n - slab0 : slab_reads;
        const int64_t nblk = (ns + EXT_BLOCK - 1) / EXT_BLOCK;
        if ((int64_t)E.order.size() < nblk) E.order.resize((size_t)nblk);
        ext_size_for(ns, n > 0 ? (int64_t)seqs[0].l_seq : READ_LEN);
        const std::function<void(int64_t)> build = [&](int64_t b) {
            const int64_t g0 = slab0 + b * EXT_BLOCK;
            ext_build_block(opt, bns, pac, seqs, chain_ar, ref_string, slab0, b, g0, g0 + EXT_BLOCK < slab0 + ns ? g0 + EXT_BLOCK : slab0 + ns);
        };
        double t0 = now_s();
        for (;;) {
            E.L.reset(); E.R.reset();
            E.team.run(nblk, build);
            if (!E.L.over() && !E.R.over()) break;
            E.L.fit(E.L.n, E.L.rb, E.L.qb);                                    // now the sizes are known: rebuild the slab
            E.R.fit(E.R.n, E.R.rb, E.R.qb);
            ++E.n_rebuilt;
        }
        E.t_build += now_s() - t0;
        for (int dir = 0; dir < 2; ++dir) {
            ExtStage* S = dir == 0 ? &E.L : &E.R;
            if (dir == 1) {                                                    // h0 of the right extension = score after the left one (:3371-3376)
                t0 = now_s();
                const int64_t step = 16384;
                const std::function<void(int64_t)> seth0 = [&](int64_t item) {
                    const int64_t i1 = (item + 1) * step < S->n ? (item + 1) * step : S->n;
                    for (int64_t i = item * step; i < i1; ++i) { SeqPair& sp = S->pairs[i]; sp.h0 = E.av[(size_t)(slab0 + sp.seqid)].a[sp.regid].score; }
                };
                E.team.run((S->n + step - 1) / step, seth0);
                E.t_fold += now_s() - t0;
            }
            for (int attempt = 0; attempt < EXT_BAND_TRIES && S->n > 0; ++attempt) {
                const int w = opt->w << attempt;
                ext_run_stage(*S, w, dir == 0 ? ol : orr);
                ext_fold(opt, seqs, slab0, *S, dir == 0, w, attempt);
                if (E.retry.empty()) break;
                E.n_retried += (int64_t)E.retry.size();
                ExtStage* again = &E.X[attempt & 1];
                ext_stage_retry(*S, *again);
                S = again;
            }
        }
        t0 = now_s();
        const std::function<void(int64_t)> purge = [&](int64_t b) {
            const int64_t g0 = slab0 + b * EXT_BLOCK, g1 = g0 + EXT_BLOCK < slab0 + ns ? g0 + EXT_BLOCK : slab0 + ns;
            for (int64_t g = g0; g < g1; ++g) ext_purge_read(opt, seqs, chain_ar, g, E.order[(size_t)b].data() + E.order_off[(size_t)g]);
        };
        E.team.run(nblk, purge);
        E.t_purge += now_s() - t0;
    }
    E.t_total += now_s() - t_begin;
}

void ext_report() {
    if (!g_ext) return;
    const Ext& E = *g_ext;
    static double last[5] = {0, 0, 0, 0, 0};
    fprintf(stderr, "[meme-dropin] extension: this chunk %.3f s (jobs built %.3f, backend calls %.3f, folded %.3f, purged %.3f); totals %.3f s, "
            "%lld backend calls with %lld pairs (%lld of them again with the doubled band), %lld slab rebuilds\n", E.t_total - last[0],
            E.t_build - last[1], E.t_call - last[2], E.t_fold - last[3], E.t_purge - last[4], E.t_total, (long long)E.n_calls,
            (long long)E.n_pairs, (long long)E.n_retried, (long long)E.n_rebuilt);
    last[0] = E.t_total; last[1] = E.t_build; last[2] = E.t_call; last[3] = E.t_fold; last[4] = E.t_purge;
}

void ext_prepare(int64_t chunk_reads, int threads) {
    if (ext_mode() == 0 || g_ext) return;
    g_ext = new Ext;
    g_ext->team.ensure(team_helpers(threads));
    // pinned memory needs a HIP context; device 0's is created here if init_devices() has not got there yet
    ext_size_for(chunk_reads < ext_slab_reads() ? chunk_reads : ext_slab_reads(), READ_LEN);
}

typedef void (*chain2aln_fn)(const mem_opt_t*, const bntseq_t*, const uint8_t*, bseq1_t*, int, mem_chain_v*, mem_alnreg_v*, mem_cache*, uint8_t*, int);
}  // namespace dropin

void mem_chain2aln_across_reads_V2(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, bseq1_t* seq_, int nseq,
                                   mem_chain_v* chain_ar, mem_alnreg_v* av_v, mem_cache* mmc, uint8_t* ref_string, int tid) {
    if (ext_mode() == 0 || !g_chunk.seqs) {
        static chain2aln_fn next = (chain2aln_fn)dlsym(RTLD_NEXT, "_Z29mem_chain2aln_across_reads_V2PK9mem_opt_tPK8bntseq_tPKhP7bseq1_tiP11mem_chain_vP12mem_alnreg_vP9mem_cachePhi");
        if (!next) { fprintf(stderr, "[meme-dropin] the reference's mem_chain2aln_across_reads_V2 was not found\n"); exit(1); }
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
    if (chain_ar - g0 != g_chunk_chain_ar) { fprintf(stderr, "[meme-dropin] the batch's chains are not a slice of the chunk's chain array\n"); exit(1); }
    {
        static std::mutex prep_mu;                              // (device mode skips the host stage's set-up; -W brings the run back here)
        std::lock_guard<std::mutex> lk(prep_mu);
        if (!g_ext) ext_prepare(g_chunk.n, g_team);
    }
    {
        std::lock_guard<std::mutex> lk(g_ext->mu);                             // the first batch to arrive extends the whole chunk
        if (g_ext->gen != g_chunk_gen) {
            ext_chunk(opt, bns, pac, g_chunk.seqs, g_chunk.n, chain_ar - g0, ref_string);
            g_ext->gen = g_chunk_gen;
        }
    }
    for (int l = 0; l < nseq; ++l) {                       // this batch's alignment arrays, owned by the reference from here on (:2633)
        const mem_alnreg_v& src = g_ext->av[(size_t)(g0 + l)];
        mem_alnreg_t* a = (mem_alnreg_t*)calloc(src.m, sizeof(mem_alnreg_t));
        if (src.n) { if (!a) { fprintf(stderr, "[meme-dropin] out of memory\n"); exit(1); } memcpy(a, src.a, src.n * sizeof(mem_alnreg_t)); }
        av_v[l].n = src.n; av_v[l].m = src.m; av_v[l].a = a;
    }
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
