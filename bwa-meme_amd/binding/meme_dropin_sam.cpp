// Part of the reference-side binding of the MI355X backend (see meme_dropin.h / meme_dropin.cpp).
#include "meme_dropin.h"
#include "meme_dropin_prof.h"
#include <string>

using namespace dropin;

// ---- mate rescue of the SAM phase on the device -----------------------------------------------------------------------------
// worker_sam (src/bwamem.cpp:1827-1902, AVX-512 build) handles a batch of read pairs in three steps: mem_sam_pe_batch_pre poses the
// Smith-Waterman jobs of mate rescue (a mate against the window its partner's alignment points at), mem_sam_pe_batch runs them through
// the kswv kernels, mem_sam_pe_batch_post turns the results into alignment records.  At the SAM phase's quiescent point (the interposed
// third kt_for call, as for the CIGAR stage) the binding runs the reference's own mem_sam_pe_batch_pre for every batch of the chunk into
// buffers of its own -- the step reads the alignment records and writes only to the mem_cache it is given --, sends the jobs of the whole
// chunk to the GPU(s) in one meme_kswv_batch_host call each, and keeps per batch the kswr_t records and the job index array (`gar`) the
// first step left for the third.  The kt_for call then runs sam_worker_dev instead of worker_sam: the third step of worker_sam's
// paired-end branch as written there (src/bwamem.cpp:1879-1900: mem_sam_pe_batch_post per pair, which also writes the SAM text, then the
// pair's alignment arrays are freed), fed from the table.
// OFF unless MEME_DROPIN_MATESW=1: measured on 2 M pairs of 150-bp reads (-t 64, 185 406 jobs) the stage costs 0.2-0.4 s of the SAM phase's
// 1.3 s instead of saving the ~0.1 s the host's 64 threads spend in the kswv kernels -- the jobs are few (one per ~22 reads; a 250-bp / 5 %
// run poses 5 000 in all), their kernel time is small next to the pre-pass that has to pose them ahead of worker_sam (0.12 s, of which
// kernels 0.03-0.05 s), and the third step then meets the alignment records cold.  SAM output is identical either way (tests).
#include "kswv.h"
namespace dropin {
std::atomic<double> g_t_matesw{0};
std::atomic<int64_t> g_n_matesw{0};
// On by default since round 4 (MEME_DROPIN_MATESW=0: the reference's own batches).  Round 3 measured the stage as a loss on a host that
// was allocator-bound; with the allocator settled and the host's threads the scarce resource (the reference's kswv batches hold ~17 jobs
// each: 45 thread-microseconds per job, 6 thread-seconds per 4 M reads) it wins: profiles/r04_mate_rescue_ab.md.
bool matesw_on_device() { static const bool v = !(getenv("MEME_DROPIN_MATESW") && atoi(getenv("MEME_DROPIN_MATESW")) == 0); return v; }
// One job per lane needs thousands of jobs to fill the GPU (a chunk of 666 k reads of 150 bases poses 22-31 k: 8 ms of kernel); below this
// many jobs per chunk (250-base reads with 5 % errors pose under a thousand) the reference's own batch runs.  MEME_DROPIN_MATESW_MIN overrides.
int64_t matesw_min_jobs() { static const int64_t v = getenv("MEME_DROPIN_MATESW_MIN") ? atoll(getenv("MEME_DROPIN_MATESW_MIN")) : 8192; return v; }
double g_mate_jobs_per_read = -1;                // of the last chunk whose jobs were posed
struct MateTable {
    std::vector<int64_t> off;                    // first record of every worker batch (+ the total)
    std::vector<kswr_t> aln;                     // records, batch after batch, in the order the jobs were posed (= regid)
    std::vector<std::vector<int32_t>> gar;       // per batch: job index (or -1) of every (alignment, orientation) mem_matesw_batch_pre looked at
    uint64_t gen = 0;                            // chunk the table belongs to
    int64_t lo = 0, hi = 0;                      // the reads it covers: [lo, hi) of the chunk, lo a multiple of BATCH_SIZE (round 6: a chunk's SAM phase runs in two halves)
};
MateTable g_mate_tab[2];
struct MateShared {                              // (the halves' pre-passes never run at the same time)
    double t_prepass = 0, t_kernel_ms = 0;
    int64_t n_jobs = 0;
    mem_cache* cache = nullptr;                  // the host pre-pass's own buffers, one slot per helper thread
    int slots = 0;
} g_mate;
struct HalfRange { int64_t lo = 0, hi = 0; } g_half[2];          // of the chunk being processed; [1] empty when the chunk is not split
std::atomic<int64_t> g_mate_hits{0}, g_mate_miss{0};
// worker_sam's paired-end branch after its first two steps (src/bwamem.cpp:1879-1900), the results of those coming from the table
void sam_worker_dev(void* data, long seqid, long batch_size, int tid) {
    worker_t* w = (worker_t*)data;
    const MateTable& T = g_mate_tab[(g_half[1].hi > g_half[1].lo && seqid >= g_half[1].lo) ? 1 : 0];      // (g_half is fixed before the phase's threads start)
    const size_t b = (size_t)((seqid - T.lo) / BATCH_SIZE);
    const std::vector<int32_t>& gar = T.gar[b];
    if (!gar.empty()) memcpy(w->mmc.seqPairArrayAux[tid], gar.data(), gar.size() * sizeof(int32_t));     // where mem_sam_pe_batch_post reads it
    kswr_t* myaln = const_cast<kswr_t*>(T.aln.data()) + T.off[b];
    int32_t gcnt = 0;
    int pos = (int)(seqid >> 1);
    for (long i = seqid; i < seqid + batch_size; i += 2) {
        mem_sam_pe_batch_post(w->opt, w->fmi->idx->bns, w->fmi->idx->pac, w->pes, (uint64_t)((w->n_processed >> 1) + pos++), &w->seqs[i], &w->regs[i], &myaln, &w->mmc,
                              gcnt, tid);
        free(w->regs[i].a);
        free(w->regs[i + 1].a);
    }
    g_mate_hits.fetch_add(T.off[b + 1] - T.off[b], std::memory_order_relaxed);
}
int cig_threads();

void mate_cache_init(int slots) {
    // worst case of one batch: 256 pairs x 2 ends x max_matesw (50) alignments x 4 orientations (mem_matesw_batch_pre asserts room before it grows)
    const int64_t cap = (int64_t)BATCH_SIZE / 2 * 2 * 50 * 4 + 1024;
    mem_cache* C = (mem_cache*)calloc(1, sizeof(mem_cache));
    if (!C) die("calloc");
    for (int t = 0; t < slots; ++t) {
        C->seqPairArrayAux[t] = (SeqPair*)malloc((size_t)(cap + MAX_LINE_LEN) * sizeof(SeqPair));
        C->seqPairArrayLeft128[t] = (SeqPair*)malloc((size_t)(cap + MAX_LINE_LEN) * sizeof(SeqPair));
        C->seqPairArrayRight128[t] = (SeqPair*)malloc((size_t)(cap + MAX_LINE_LEN) * sizeof(SeqPair));
        C->wsize[t] = cap;
        const int64_t rcap = 8 << 20, qcap = 2 << 20;
        C->wsize_buf_ref[t * CACHE_LINE] = rcap; C->wsize_buf_qer[t * CACHE_LINE] = qcap;
        C->seqBufLeftRef[t * CACHE_LINE] = (uint8_t*)_mm_malloc((size_t)rcap, 64); C->seqBufRightRef[t * CACHE_LINE] = (uint8_t*)_mm_malloc((size_t)rcap, 64);
        C->seqBufLeftQer[t * CACHE_LINE] = (uint8_t*)_mm_malloc((size_t)qcap, 64); C->seqBufRightQer[t * CACHE_LINE] = (uint8_t*)_mm_malloc((size_t)qcap, 64);
        if (!C->seqPairArrayAux[t] || !C->seqPairArrayLeft128[t] || !C->seqPairArrayRight128[t] || !C->seqBufLeftRef[t * CACHE_LINE] ||
            !C->seqBufRightRef[t * CACHE_LINE] || !C->seqBufLeftQer[t * CACHE_LINE] || !C->seqBufRightQer[t * CACHE_LINE]) die("mate-rescue buffers");
    }
    g_mate.cache = C; g_mate.slots = slots;
}

// ---- the posing step on the device too (round 6): meme_matesw_batch_host poses the chunk's jobs from the records' (rb, rid, score) -- gathered here in
// one parallel walk -- and the bases in HBM, runs them, and returns what the third step needs per worker batch: gar and the kswr_t records.  The host
// keeps mem_sam_pe_batch_post only.  MEME_DROPIN_MATE_POSE=0: the reference's mem_sam_pe_batch_pre per batch on the host, sequences shipped (rounds 3-5).
// MEME_DROPIN_MATE_CHECK=1 (tests): both, compared entry by entry -- a difference is fatal.
bool mate_pose_on_device() { static const bool v = !(getenv("MEME_DROPIN_MATE_POSE") && atoi(getenv("MEME_DROPIN_MATE_POSE")) == 0); return v; }
bool mate_check() { static const bool v = getenv("MEME_DROPIN_MATE_CHECK") != nullptr; return v; }
struct MateStage { meme_mate_reg* regs = nullptr; int64_t regs_cap = 0; int64_t* off = nullptr; int64_t off_cap = 0; };      // pinned, grow-only
MateStage g_mate_stage;
std::atomic<int64_t> g_mate_posed_dev{0};
double g_mate_pose_ms = 0;

// ---- the chunk's alignment records, once (round 6) ----------------------------------------------------------------------------------
// Between worker_aln and worker_sam three loops of the binding read the alignment records of every read of the chunk -- the insert-size
// pre-filter (mem_pestat's four tests), the CIGAR stage's candidates (+ its posing and entry loops), the mate-rescue stage's gather --
// and each record is its own heap block: 0.8-1.7 CPU-seconds per 8 M reads EACH, nearly all of it cache misses (profiles/r06_host_cpu.md).
// The first of them to run walks the heap once and leaves the fields all three need back to back (48 bytes per record, read order); the
// others read that.  Nothing writes a record between worker_aln's end and worker_sam's start (src/bwamem.cpp:2003-2036), which is the
// span the digest is used in; it is tagged with the chunk's number and the array it was taken from.
struct RecDigest { int64_t rb, re; int32_t qb, qe, rid, score, truesc, w, secondary, sec_score; };
struct ChunkDigest {
    uint64_t gen = 0; const mem_alnreg_v* regs = nullptr; int64_t n = 0;
    std::vector<int64_t> off;                           // read g's records: [off[g], off[g + 1])
    RecDigest* rec = nullptr; int64_t cap = 0;
} g_digest;
// (When mem_pestat did not run for the chunk -- insert sizes given with -I, fewer than 64 pairs, MEME_DROPIN_PESTAT=0 -- the CIGAR and the mate-rescue pre-pass of a
// half arrive here side by side: the first builds, the other waits.  A later call for the same chunk returns what is there; the next chunk's build starts only after
// this chunk's worker_sam, i.e. after every reader.)
std::mutex g_digest_mu;
const ChunkDigest& chunk_digest(const mem_alnreg_v* regs, int64_t n) {
    std::lock_guard<std::mutex> lk(g_digest_mu);
    ChunkDigest& D = g_digest;
    if (D.gen == g_chunk_gen && D.regs == regs && D.n == n) return D;
    TeamLabel lbl("records: one walk over the heap");
    const int nt = cig_threads();
    D.off.resize((size_t)n + 1);
    int64_t* off = D.off.data();
    off[0] = 0;
    team_for(n, nt, [&](int64_t g0, int64_t g1, int) { for (int64_t g = g0; g < g1; ++g) off[g + 1] = (int64_t)regs[g].n; });
    for (int64_t g = 0; g < n; ++g) off[g + 1] += off[g];
    if (off[n] + 1 > D.cap) { free(D.rec); D.cap = off[n] + off[n] / 4 + 4096; if (!(D.rec = (RecDigest*)malloc((size_t)D.cap * sizeof(RecDigest)))) { fprintf(stderr, "[meme-dropin] out of memory\n"); exit(1); } }
    RecDigest* rec = D.rec;
    team_for(n, nt, [&](int64_t g0, int64_t g1, int) {
        for (int64_t g = g0; g < g1; ++g) {
            const mem_alnreg_v& av = regs[g];
            RecDigest* o = rec + off[g];
            for (size_t k = 0; k < av.n; ++k) {
                const mem_alnreg_t& p = av.a[k];
                o[k].rb = p.rb; o[k].re = p.re; o[k].qb = p.qb; o[k].qe = p.qe; o[k].rid = p.rid; o[k].score = p.score; o[k].truesc = p.truesc; o[k].w = p.w; o[k].secondary = p.secondary;
            }
            for (size_t k = 0; k < av.n; ++k) o[k].sec_score = o[k].secondary >= 0 && o[k].secondary < (int)av.n ? o[o[k].secondary].score : 0;
        }
    });
    D.gen = g_chunk_gen; D.regs = regs; D.n = n;
    return D;
}

// 1: the table is filled, 0: too few jobs (worker_sam runs as it is), -1: the device stage is not available for this chunk (the host poses)
int matesw_prepass_device(int h) {
    const double t0 = now_s();
    worker_t* w = g_worker;
    const mem_opt_t* opt = g_opt;
    const int64_t lo = g_half[h].lo, n = g_half[h].hi - lo;     // the half's reads are [lo, lo + n) of the chunk; index g below is relative to lo
    const int nd = (int)g_dev.size();
    if ((n & 1) || (lo % BATCH_SIZE) != 0) return -1;
    for (int d = 0; d < nd; ++d) { const ChunkPart& P = g_chunk.part[(size_t)d]; if (P.count > 0 && (!P.reads_on_ctx || (P.first % BATCH_SIZE) != 0)) return -1; }
    MateStage& G = g_mate_stage;
    if (n + 1 > G.off_cap) { meme_host_free(G.off); G.off_cap = n + n / 4 + 64; if (!(G.off = (int64_t*)meme_host_alloc(G.off_cap * 8))) die("meme_host_alloc"); }
    const int nt = cig_threads();
    const ChunkDigest& D = chunk_digest(w->regs, g_chunk.n);
TeamLabel tl_9237("mate: gather records");
    const int64_t d0 = D.off[(size_t)lo];
    team_for(n + 1, nt, [&](int64_t g0, int64_t g1, int) { for (int64_t g = g0; g < g1; ++g) G.off[g] = D.off[(size_t)(lo + g)] - d0; });
    const int64_t nrec = G.off[n];
    if (nrec + 1 > G.regs_cap) { meme_host_free(G.regs); G.regs_cap = nrec + nrec / 4 + 4096; if (!(G.regs = (meme_mate_reg*)meme_host_alloc(G.regs_cap * (int64_t)sizeof(meme_mate_reg)))) die("meme_host_alloc"); }
    team_for(nrec, nt, [&](int64_t k0, int64_t k1, int) {
        const RecDigest* src = D.rec + d0;
        for (int64_t k = k0; k < k1; ++k) { G.regs[k].rb = src[k].rb; G.regs[k].rid = src[k].rid; G.regs[k].score = src[k].score; }
    });
    meme_pestat pes[4];
    for (int r = 0; r < 4; ++r) { pes[r].low = w->pes[r].low; pes[r].high = w->pes[r].high; pes[r].failed = w->pes[r].failed; pes[r].pad = 0; }
    meme_mate_opt mo;
    mo.a = opt->a; mo.b = opt->b; mo.o_del = opt->o_del; mo.e_del = opt->e_del; mo.o_ins = opt->o_ins; mo.e_ins = opt->e_ins;
    mo.pen_unpaired = opt->pen_unpaired; mo.max_matesw = opt->max_matesw; mo.min_seed_len = opt->min_seed_len; mo.batch_reads = BATCH_SIZE;
    const int64_t nb = (n + BATCH_SIZE - 1) / BATCH_SIZE;
    struct PartRes { std::vector<int32_t> gar; std::vector<int64_t> gar_off, job_off; std::vector<kswr_t> aln; double pose_ms = 0, kernel_ms = 0; int rc = 0; };
    std::vector<PartRes> PR((size_t)nd);
    // the half's reads on device slice d: [s0, s1) of the chunk
    auto part_range = [&](int d, int64_t& s0, int64_t& s1) {
        const ChunkPart& P = g_chunk.part[(size_t)d];
        s0 = P.first > lo ? P.first : lo;
        s1 = P.first + P.count < lo + n ? P.first + P.count : lo + n;
    };
    auto run_part = [&](int d) {
        const ChunkPart& P = g_chunk.part[(size_t)d];
        PartRes& R = PR[(size_t)d];
        int64_t s0, s1;
        part_range(d, s0, s1);
        if (s1 <= s0) return;
        const int64_t cnt = s1 - s0;
        std::vector<int64_t> off((size_t)cnt + 1);
        const int64_t base = G.off[s0 - lo];
        for (int64_t i = 0; i <= cnt; ++i) off[(size_t)i] = G.off[s0 - lo + i] - base;
        meme_mate_host_result H;
        // (the GPU's second ctx executes; the bases are read where the slice's seeding ctx holds them -- that ctx is busy with the CIGAR stage meanwhile)
        R.rc = meme_matesw_batch_host(g_dev[(size_t)d].bsw, P.ctx, G.regs + base, off.data(), s0 - P.first, cnt, pes, g_contigs.data(), (int32_t)g_contigs.size(), g_bns->l_pac, &mo, &H);
        if (R.rc) return;
        static_assert(sizeof(kswr_t) == sizeof(meme_kswr), "kswr_t layout");
        R.gar.assign(H.gar, H.gar + H.n_gar);
        R.gar_off.assign(H.gar_off, H.gar_off + H.nbatches + 1);
        R.job_off.assign(H.job_off, H.job_off + H.nbatches + 1);
        R.aln.resize((size_t)H.njobs);
        if (H.njobs) memcpy(R.aln.data(), H.res, (size_t)H.njobs * sizeof(kswr_t));
        R.pose_ms = H.pose_ms; R.kernel_ms = H.kernel_ms;
        if (verify_on() && g_dev[(size_t)d].vfy_bsw) {           // MEME_DROPIN_VERIFY: the same once more on a ctx of its own (the slot's verify ctx belongs to the CIGAR pre-pass running beside this one)
            meme_mate_host_result V;
            if (meme_matesw_batch_host(g_dev[(size_t)d].vfy_bsw, P.ctx, G.regs + base, off.data(), s0 - P.first, cnt, pes, g_contigs.data(), (int32_t)g_contigs.size(), g_bns->l_pac, &mo, &V)) die("MEME_DROPIN_VERIFY: the second run of the mate-rescue stage");
            if (V.njobs != H.njobs || V.n_gar != (int64_t)R.gar.size() || (V.n_gar && memcmp(V.gar, R.gar.data(), (size_t)V.n_gar * 4) != 0)) verify_fail("mate rescue (jobs posed)", -1, "");
            for (int64_t k = 0; k < V.njobs; ++k) if (memcmp(&V.res[k], &R.aln[(size_t)k], sizeof(kswr_t)) != 0) verify_fail("mate-rescue Smith-Waterman", k, "");
            verify_note(g_chunk.seq, "mate-rescue", d, verify_hash(R.aln.data(), R.aln.size() * sizeof(kswr_t), verify_hash(R.gar.data(), R.gar.size() * 4)), V.njobs);
        }
    };
    std::vector<std::thread> th;
    for (int d = 1; d < nd; ++d) th.emplace_back(run_part, d);
    run_part(0);
    for (auto& x : th) x.join();
    for (int d = 0; d < nd; ++d)
        if (PR[(size_t)d].rc) {
            static std::atomic<int> said{0};
            if (said.fetch_add(1) < 2) fprintf(stderr, "[meme-dropin] mate rescue: the device stage is not available for this chunk (%s): the reference's posing function on the host\n", meme_last_error());
            return -1;
        }
    MateTable& T = g_mate_tab[h];
    T.lo = lo; T.hi = lo + n;
    T.off.assign((size_t)nb + 1, 0);
    T.gar.assign((size_t)nb, std::vector<int32_t>());
    int64_t total = 0;
    for (int d = 0; d < nd; ++d) total += (int64_t)PR[(size_t)d].aln.size();
    g_mate_jobs_per_read = n > 0 ? (double)total / (double)n : 0;
    if (total * (g_chunk.n / (n > 0 ? n : 1)) < matesw_min_jobs() && !mate_check()) { g_mate.t_prepass += now_s() - t0; T.gen = 0; return 0; }
    T.aln.resize((size_t)total);
    int64_t jbase = 0;
    double km = 0, pm = 0;
    for (int d = 0; d < nd; ++d) {
        const PartRes& R = PR[(size_t)d];
        int64_t s0, s1;
        part_range(d, s0, s1);
        if (s1 <= s0) continue;
        const int64_t b0 = (s0 - lo) / BATCH_SIZE, nbl = (int64_t)R.gar_off.size() - 1;
        for (int64_t b = 0; b < nbl; ++b) {
            T.off[(size_t)(b0 + b)] = jbase + R.job_off[(size_t)b];
            T.gar[(size_t)(b0 + b)].assign(R.gar.begin() + R.gar_off[(size_t)b], R.gar.begin() + R.gar_off[(size_t)b + 1]);
        }
        if (!R.aln.empty()) memcpy(&T.aln[(size_t)jbase], R.aln.data(), R.aln.size() * sizeof(kswr_t));
        jbase += (int64_t)R.aln.size();
        km = km > R.kernel_ms ? km : R.kernel_ms; pm = pm > R.pose_ms ? pm : R.pose_ms;
    }
    T.off[(size_t)nb] = total;
    g_mate.t_kernel_ms += km; g_mate_pose_ms += pm; g_mate.n_jobs += total; g_mate_posed_dev += total;
    g_mate.t_prepass += now_s() - t0;
    T.gen = g_chunk_gen;
    return 1;
}

bool matesw_prepass_host(int h);
bool matesw_prepass(int h) {                     // false: too few jobs for the device, worker_sam runs as it is (for this half of the chunk)
    if (mate_pose_on_device()) {
        const int r = matesw_prepass_device(h);
        if (r >= 0 && !mate_check()) return r == 1;
        if (r >= 0) {                            // MEME_DROPIN_MATE_CHECK: the reference's own posing function over the same records, then its jobs on the device as before
            MateTable dev;
            dev.off = g_mate_tab[h].off; dev.aln = g_mate_tab[h].aln; dev.gar = g_mate_tab[h].gar;
            const int64_t dev_jobs = (int64_t)dev.aln.size();
            g_mate.n_jobs -= dev_jobs;           // (counted again below)
            const bool ok = matesw_prepass_host(h);
            const MateTable& H = g_mate_tab[h];
            if (!ok) { if (dev_jobs >= matesw_min_jobs()) { fprintf(stderr, "[meme-dropin] MATE_CHECK: the host poses fewer jobs than the threshold, the device %lld\n", (long long)dev_jobs); exit(1); } return false; }
            bool same = dev.off == H.off && dev.gar.size() == H.gar.size() && dev.aln.size() == H.aln.size();
            // (the host path pre-fills its job index buffer with -1 in this mode: the reference leaves entries it does not pose unwritten, the device writes -1)
            for (size_t b = 0; same && b < dev.gar.size(); ++b) same = dev.gar[b] == H.gar[b];
            for (size_t k = 0; same && k < dev.aln.size(); ++k) same = memcmp(&dev.aln[k], &H.aln[k], sizeof(kswr_t)) == 0;
            if (!same) { fprintf(stderr, "[meme-dropin] MATE_CHECK: mate rescue posed on the device differs from the reference's mem_sam_pe_batch_pre (jobs %zu vs %zu)\n", dev.aln.size(), H.aln.size()); exit(1); }
            return true;
        }
    }
    return matesw_prepass_host(h);
}

bool matesw_prepass_host(int h) {
    const double t0 = now_s();
    worker_t* w = g_worker;
    const mem_opt_t* opt = g_opt;
    const int64_t lo = g_half[h].lo, n = g_half[h].hi - lo;     // the half's reads: [lo, lo + n) of the chunk
    const int64_t nb = (n + BATCH_SIZE - 1) / BATCH_SIZE;
    const int nt = cig_threads();
    if (!g_mate.cache) mate_cache_init(nt);
    struct BatchJobs { std::vector<meme_kswv_job> jobs; std::vector<uint8_t> ref, qer; };
    std::vector<BatchJobs> B((size_t)nb);
    MateTable& T = g_mate_tab[h];
    T.lo = lo; T.hi = lo + n;
    T.gar.assign((size_t)nb, std::vector<int32_t>());
    std::atomic<int64_t> next_b{0};
TeamLabel tl_46641("mate: host posing (mem_sam_pe_batch_pre)");
    team_run(g_mate.slots, [&](int t) {                             // (one buffer slot per share; batches handed out one by one)
    for (int64_t b = next_b.fetch_add(1); b < nb; b = next_b.fetch_add(1)) {
        const int64_t st = lo + b * BATCH_SIZE, ed = lo + ((b + 1) * BATCH_SIZE < n ? (b + 1) * BATCH_SIZE : n);
        int64_t pcnt = 0;
        int32_t gcnt = 0, maxRef = 0, maxQer = 0;
        int64_t pos = st >> 1;
        if (mate_check()) memset(g_mate.cache->seqPairArrayAux[t], 0xff, (size_t)BATCH_SIZE * 50 * 4 * sizeof(int32_t));     // (entries the step does not write compare as -1)
        for (int64_t i = st; i + 1 < ed; i += 2)                  // worker_sam's loop (src/bwamem.cpp:1855-1866)
            mem_sam_pe_batch_pre(opt, w->fmi->idx->bns, w->fmi->idx->pac, w->pes, (uint64_t)((w->n_processed >> 1) + pos++), &w->seqs[i], &w->regs[i], g_mate.cache,
                                 pcnt, gcnt, maxRef, maxQer, t);
        BatchJobs& J = B[(size_t)b];
        T.gar[(size_t)b].assign((const int32_t*)g_mate.cache->seqPairArrayAux[t], (const int32_t*)g_mate.cache->seqPairArrayAux[t] + gcnt);
        if (pcnt == 0) continue;
        const SeqPair* sp = g_mate.cache->seqPairArrayLeft128[t];
        const int64_t rbytes = (int64_t)sp[pcnt - 1].idr + sp[pcnt - 1].len1, qbytes = (int64_t)sp[pcnt - 1].idq + sp[pcnt - 1].len2;
        J.ref.assign(g_mate.cache->seqBufLeftRef[t * CACHE_LINE], g_mate.cache->seqBufLeftRef[t * CACHE_LINE] + rbytes);
        J.qer.assign(g_mate.cache->seqBufLeftQer[t * CACHE_LINE], g_mate.cache->seqBufLeftQer[t * CACHE_LINE] + qbytes);
        J.jobs.resize((size_t)pcnt);
        for (int64_t k = 0; k < pcnt; ++k) { meme_kswv_job& j = J.jobs[(size_t)k]; j.idr = sp[k].idr; j.idq = sp[k].idq; j.len1 = sp[k].len1; j.len2 = sp[k].len2; j.xtra = sp[k].h0; j.pad = 0; }
    }
    });
    T.off.assign((size_t)nb + 1, 0);
    for (int64_t b = 0; b < nb; ++b) T.off[(size_t)b + 1] = T.off[(size_t)b] + (int64_t)B[(size_t)b].jobs.size();
    const int64_t total = T.off[(size_t)nb];
    g_mate_jobs_per_read = n > 0 ? (double)total / (double)n : 0;
    if (total * (g_chunk.n / (n > 0 ? n : 1)) < matesw_min_jobs()) { g_mate.t_prepass += now_s() - t0; T.gen = 0; return false; }
    T.aln.resize((size_t)total);
    // the chunk's batches in contiguous runs over the GPUs, one call each
    const int nd = (int)g_dev.size();
    meme_bsw_opt bo;
    memset(&bo, 0, sizeof(bo));
    bo.o_del = opt->o_del; bo.e_del = opt->e_del; bo.o_ins = opt->o_ins; bo.e_ins = opt->e_ins; bo.a = opt->a; bo.b = opt->b;
    std::vector<double> kms((size_t)nd, 0.0);
    auto run_part = [&](int d) {
        const int64_t b0 = nb * d / nd, b1 = nb * (d + 1) / nd;
        const int64_t j0 = T.off[(size_t)b0], nj = T.off[(size_t)b1] - j0;
        if (nj == 0) return;
        std::vector<meme_kswv_job> jobs((size_t)nj);
        int64_t rtot = 0, qtot = 0;
        for (int64_t b = b0; b < b1; ++b) { rtot += (int64_t)B[(size_t)b].ref.size(); qtot += (int64_t)B[(size_t)b].qer.size(); }
        std::vector<uint8_t> ref((size_t)rtot + 1), qer((size_t)qtot + 1);
        int64_t ro = 0, qo = 0, k = 0;
        for (int64_t b = b0; b < b1; ++b) {
            const BatchJobs& J = B[(size_t)b];
            if (!J.ref.empty()) memcpy(ref.data() + ro, J.ref.data(), J.ref.size());
            if (!J.qer.empty()) memcpy(qer.data() + qo, J.qer.data(), J.qer.size());
            for (const meme_kswv_job& j : J.jobs) { meme_kswv_job x = j; x.idr += ro; x.idq += qo; jobs[(size_t)k++] = x; }
            ro += (int64_t)J.ref.size(); qo += (int64_t)J.qer.size();
        }
        meme_kswv_host_result R;
        if (meme_kswv_batch_host(g_dev[(size_t)d].bsw, jobs.data(), nj, ref.data(), rtot, qer.data(), qtot, &bo, &R)) die("meme_kswv_batch_host");
        static_assert(sizeof(kswr_t) == sizeof(meme_kswr) && offsetof(kswr_t, score) == 0 && offsetof(kswr_t, te) == 4 && offsetof(kswr_t, qe) == 8 &&
                      offsetof(kswr_t, score2) == 12 && offsetof(kswr_t, te2) == 16 && offsetof(kswr_t, tb) == 20 && offsetof(kswr_t, qb) == 24, "kswr_t layout");
        memcpy(&T.aln[(size_t)j0], R.res, (size_t)nj * sizeof(kswr_t));
        if (verify_on() && g_dev[(size_t)d].vfy_bsw) {                 // MEME_DROPIN_VERIFY: the same jobs on another ctx of the GPU
            meme_kswv_host_result V;
            if (meme_kswv_batch_host(g_dev[(size_t)d].vfy_bsw, jobs.data(), nj, ref.data(), rtot, qer.data(), qtot, &bo, &V)) die("MEME_DROPIN_VERIFY: the second run of the mate-rescue stage");
            for (int64_t k = 0; k < nj; ++k) if (memcmp(&T.aln[(size_t)(j0 + k)], &V.res[k], sizeof(kswr_t)) != 0) verify_fail("mate-rescue Smith-Waterman", k, "");
            verify_note(g_chunk.seq, "mate-rescue", d, verify_hash(V.res, (size_t)nj * sizeof(kswr_t)), nj);
        }
        kms[(size_t)d] = R.kernel_ms;
    };
    std::vector<std::thread> th;
    for (int d = 1; d < nd; ++d) th.emplace_back(run_part, d);
    run_part(0);
    for (auto& x : th) x.join();
    double km = 0;
    for (double v : kms) km = km > v ? km : v;
    g_mate.t_kernel_ms += km; g_mate.n_jobs += total; g_mate.t_prepass += now_s() - t0;
    T.gen = g_chunk_gen;
    return true;
}
}  // namespace dropin

// (with the stage off: the reference's batch, timed)
typedef int (*sam_pe_batch_fn)(const mem_opt_t*, mem_cache*, int64_t&, int64_t&, kswr_t*, int32_t, int32_t, int);
int mem_sam_pe_batch(const mem_opt_t* opt, mem_cache* mmc, int64_t& pcnt, int64_t& pcnt8, kswr_t* aln, int32_t maxRefLen, int32_t maxQerLen, int tid) {
    static const sam_pe_batch_fn next = (sam_pe_batch_fn)ref_sym(R_MEM_SAM_PE_BATCH);
    g_mate_miss.fetch_add(pcnt, std::memory_order_relaxed);
    const double t0 = now_s();
    const int64_t n = pcnt;
    const int rc = next(opt, mmc, pcnt, pcnt8, aln, maxRefLen, maxQerLen, tid);
    g_t_matesw = g_t_matesw + (now_s() - t0);
    g_n_matesw += n;
    return rc;
}
void meme_dropin_report_mate() {
    if (!matesw_on_device()) return;
    fprintf(stderr, "[meme-dropin] mate rescue on the device: %lld Smith-Waterman jobs posed so far (kernels %.3f s, whole pre-pass %.3f s); jobs whose results worker_sam's third step took from the table "
            "%lld, run by the reference's kernels %lld; jobs posed by the device's own posing step %lld (posing kernels %.3f s)\n", (long long)g_mate.n_jobs, g_mate.t_kernel_ms * 1e-3, g_mate.t_prepass,
            (long long)g_mate_hits.load(), (long long)g_mate_miss.load(), (long long)g_mate_posed_dev.load(), g_mate_pose_ms * 1e-3);
}
void meme_dropin_report_matesw() {
    fprintf(stderr, "[meme-dropin] mate rescue by the reference's kernels (chunks below the job threshold): %.3f thread-seconds for %lld pairs\n", (double)g_t_matesw, (long long)g_n_matesw);
}

// ---- CIGAR generation of the SAM phase on the device ---------------------------------------------------------------------
// mem_reg2aln (src/bwamem.cpp:2314-2380) calls bwa_gen_cigar2 (src/bwa.cpp:274-362) up to three times per alignment written out: that
// function unpacks the target from the 2-bit reference (bns_get_seq: a malloc and a base-by-base loop), reverses both sequences on the
// reverse strand, runs ksw_global2 (src/ksw.cpp:560-670: banded global alignment with traceback, 42 % of the SAM phase's thread time on 250-bp
// reads with 5 % errors) unless the alignment is gap-free, and walks the CIGAR once more for NM and the MD string.  Between the two kt_for
// phases -- worker_aln has joined, worker_sam has not started: the kt_for call that carries worker_sam is interposed -- the binding poses
// those very calls for EVERY alignment record of the chunk (the records are complete and nobody touches them; mem_reg2aln's band loop is
// followed round by round), runs them on the GPU(s) as one batch per round (meme_gen_cigar_batch_host: the text and the reads are already
// in HBM) and keeps score, CIGAR, NM and MD; bwa_gen_cigar2 itself is interposed and answers from that table after an exact comparison of
// its arguments and of the query bases.  Calls the table does not hold (alignments made later by mate rescue, calls from other places)
// go to the reference's function.  MEME_DROPIN_CIGAR=0 switches the stage off.  Round 4 hooked ksw_global2 only: the target unpacking,
// the reversals, two sequence hashes per call and the NM / MD loop stayed on the host.
namespace dropin {

struct CigEntry { int64_t rb; int64_t blob; int32_t g, qb, qlen, tlen, w_, score, n_cigar, nm, md_len, pad; };     // blob: the entry's n_cigar operations, then its MD string + NUL
struct CigTable {
    std::mutex mu;
    std::atomic<uint64_t> gen{0};                        // chunk the table belongs to; published (release) when the table is complete: the other half's workers may be looking
    std::vector<CigEntry> e;
    std::vector<char> blob;                              // per entry what the hook returns: operations + MD, back to back
    std::vector<uint32_t> slot;                          // open addressing: entry + 1, 0 = empty; size a power of two
    uint64_t mask = 0;
    double t_prepass = 0, t_kernel_ms = 0;
    int64_t n_jobs = 0;
} g_cig_tab[2];                                      // one table per half of the chunk (round 6); the totals are kept in [0]
#define g_cig g_cig_tab[0]
std::atomic<int64_t> g_cig_hits{0}, g_cig_miss{0};
// (the hook's counters per thread, added to the totals when a worker thread ends: two shared atomics touched four million times per
// 4 M reads by 64 threads cost more than the look-ups they count)
struct CigTally { int64_t hits = 0, miss = 0; ~CigTally() { if (hits) g_cig_hits += hits; if (miss) g_cig_miss += miss; } };
thread_local CigTally tl_cig;
bool cigar_on_device() { static const bool v = !(getenv("MEME_DROPIN_CIGAR") && atoi(getenv("MEME_DROPIN_CIGAR")) == 0); return v; }

inline uint64_t mix64(uint64_t h, uint64_t v) { h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); return h * 0xff51afd7ed558ccdull; }
inline uint64_t cig_key(int64_t rb, int qlen, int tlen, int w_) { const uint64_t h = mix64(mix64((uint64_t)rb, (uint64_t)(uint32_t)qlen << 32 | (uint32_t)tlen), (uint64_t)w_); return h ^ (h >> 29); }
inline int infer_bw_(int l1, int l2, int score, int a, int q, int r) {        // infer_bw, src/bwamem.cpp:2151-2158
    if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
    int w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
    if (w < abs(l1 - l2)) w = abs(l1 - l2);
    return w;
}

// helper threads of the pre-pass's host loops: a few dozen are enough, and an OpenMP team of 256 would still be spinning when worker_sam starts
int cig_threads() { static const int m = (int)std::thread::hardware_concurrency(); return m < 1 ? 1 : (m < 32 ? m : 32); }

void cig_prepass(int h) {
    const double t0 = now_s();
    double cpu0; { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); cpu0 = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
    CigTable& T = g_cig_tab[h];
    T.e.clear(); T.blob.clear();
    const mem_opt_t* opt = g_opt;
    const int64_t h_lo = g_half[h].lo, n = g_half[h].hi - h_lo, l_pac = g_bns->l_pac;      // the half's reads: [h_lo, h_lo + n) of the chunk
    // per alignment record: where mem_reg2aln's loop stands (band argument of the next call, score of the last one)
    // (the record's own fields ride along: the posing and entry loops below do not go back to the heap)
    struct Cand { int64_t g, rb; int32_t reg, w2, last_sc, tries, qb, qlen, tlen, truesc; };
    std::vector<Cand> cand;
    {
        const int nt = cig_threads();
        std::vector<std::vector<Cand>> part((size_t)nt);
        const ChunkDigest& D = chunk_digest(g_worker->regs, g_chunk.n);
TeamLabel tl_67525("cigar: candidates");
        team_for(n, nt, [&](int64_t g_lo, int64_t g_hi, int t) {
            std::vector<Cand>& mine = part[(size_t)t];
            for (int64_t g = h_lo + g_lo; g < h_lo + g_hi; ++g) {
                const RecDigest* av = D.rec + D.off[(size_t)g];
                const int avn = (int)(D.off[(size_t)g + 1] - D.off[(size_t)g]);
                for (int i = 0; i < avn; ++i) {
                    const RecDigest& p = av[i];
                    if (p.rb < 0 || p.re < 0 || p.score < opt->T) continue;
                    if (p.secondary >= 0 && p.secondary < avn && p.score < p.sec_score * opt->XA_drop_ratio) continue;
                    // what bwa_gen_cigar2 rejects, or would unpack from beyond the text, is left to it (src/bwa.cpp:285-287)
                    if (p.qe <= p.qb || p.rb >= p.re || (p.rb < l_pac && p.re > l_pac) || p.re > 2 * l_pac) continue;
                    const int tmp = infer_bw_(p.qe - p.qb, (int)(p.re - p.rb), p.truesc, opt->a, opt->o_del, opt->e_del);
                    int w2 = infer_bw_(p.qe - p.qb, (int)(p.re - p.rb), p.truesc, opt->a, opt->o_ins, opt->e_ins);
                    w2 = w2 > tmp ? w2 : tmp;
                    if (w2 > opt->w) w2 = w2 < p.w ? w2 : p.w;
                    mine.push_back({g, p.rb, (int32_t)i, w2, -(1 << 30), 0, p.qb, p.qe - p.qb, (int32_t)(p.re - p.rb), p.truesc});
                }
            }
        });
        size_t tot = 0;
        for (auto& v : part) tot += v.size();
        cand.reserve(tot);
        for (auto& v : part) cand.insert(cand.end(), v.begin(), v.end());      // (static schedule: still in read order)
    }
    const double t_cand1 = now_s();
    const int nd = (int)g_dev.size();
    meme_bsw_opt bo;
    memset(&bo, 0, sizeof(bo));
    bo.o_del = opt->o_del; bo.e_del = opt->e_del; bo.o_ins = opt->o_ins; bo.e_ins = opt->e_ins; bo.a = opt->a; bo.b = opt->b;
    double t_pose = 0, t_call = 0, t_take = 0;
    bool warned = false;
    for (int round = 0; round < 3 && !cand.empty(); ++round) {
        // this round's calls, per device part (candidates are in read order: a part's candidates are contiguous).  Posed in parallel
        // (the alignment records are scattered over the heap), compacted in order.
        double tp = now_s();
        const int64_t nc = (int64_t)cand.size();
        std::vector<meme_cjob> posed((size_t)nc);
        std::vector<int8_t> dev_of((size_t)nc);
TeamLabel tl_22918("cigar: pose");
        team_for(nc, cig_threads(), [&](int64_t c_lo, int64_t c_hi, int) {
        for (int64_t c = c_lo; c < c_hi; ++c) {
            Cand& C = cand[(size_t)c];
            C.w2 = C.w2 < opt->w << 2 ? C.w2 : opt->w << 2;                         // (:2342)
            dev_of[(size_t)c] = -1;
            int d = 0;
            while (d + 1 < nd && C.g >= g_chunk.part[(size_t)d].first + g_chunk.part[(size_t)d].count) ++d;
            if (!g_chunk.part[(size_t)d].reads_on_ctx) { C.tries = 99; continue; }       // (a part seeded in pieces: its reads are not all on the ctx; the reference's function computes these)
            meme_cjob& J = posed[(size_t)c];
            J.rb = C.rb; J.read = (int32_t)(C.g - g_chunk.part[(size_t)d].first); J.qb = C.qb; J.qlen = C.qlen; J.tlen = C.tlen; J.w_ = C.w2; J.pad = 0;
            dev_of[(size_t)c] = (int8_t)d;
        }
        });
        std::vector<std::vector<meme_cjob>> jobs((size_t)nd);
        std::vector<std::vector<uint32_t>> who((size_t)nd);
        for (int64_t c = 0; c < nc; ++c) {
            const int d = dev_of[(size_t)c];
            if (d < 0) continue;
            jobs[(size_t)d].push_back(posed[(size_t)c]);
            who[(size_t)d].push_back((uint32_t)c);
        }
        t_pose += now_s() - tp; tp = now_s();
        // One call per device; a call that the backend refuses for want of memory (MEME_E_CAPACITY: the backtrack matrices of the batch
        // beside the resident index) is repeated in halves, and whatever cannot be computed at all is simply left out of the table:
        // the hook then runs the reference's function for those alignments.  The stage is an optimisation, never a reason to stop.
        struct Part { std::vector<meme_cres> res; std::vector<char> blob; int64_t done = 0; double kernel_ms = 0; };      // res[k].cigar_off: the job's place in blob
        std::vector<Part> part((size_t)nd);
        std::vector<std::thread> th;
        auto run = [&](int d) {
            const std::vector<meme_cjob>& Jv = jobs[(size_t)d];
            Part& P = part[(size_t)d];
            const int64_t nj = (int64_t)Jv.size();
            P.res.reserve((size_t)nj);
            int64_t piece = nj;
            while (P.done < nj) {
                const int64_t m = piece < nj - P.done ? piece : nj - P.done;
                meme_cres_host R;
                const int rc = meme_gen_cigar_batch_host(g_chunk.part[(size_t)d].ctx, Jv.data() + P.done, m, &bo, &R);
                if (rc == MEME_E_CAPACITY && m > 4096) { piece = m / 2; continue; }
                if (rc != MEME_OK) {
                    static std::mutex warn_mu;
                    std::lock_guard<std::mutex> lk(warn_mu);
                    if (!warned) fprintf(stderr, "[meme-dropin] CIGAR stage: %lld of this chunk's alignments stay with the host (%s)\n", (long long)(nj - P.done), meme_last_error());
                    warned = true;
                    break;
                }
                if (verify_on() && g_chunk.part[(size_t)d].vfy) {          // MEME_DROPIN_VERIFY: the same calls on the ctx that holds the same reads
                    const uint64_t h = verify_hash(R.md, (size_t)R.md_bytes, verify_hash(R.cigars, (size_t)R.total_ops * 4, verify_hash(R.res, (size_t)m * sizeof(meme_cres))));
                    meme_cres_host V;
                    if (meme_gen_cigar_batch_host(g_chunk.part[(size_t)d].vfy, Jv.data() + P.done, m, &bo, &V)) die("MEME_DROPIN_VERIFY: the second run of the CIGAR stage");
                    if (V.total_ops != R.total_ops || V.md_bytes != R.md_bytes) verify_fail("CIGAR", -1, "totals of operations / MD bytes");
                    if (memcmp(V.res, R.res, (size_t)m * sizeof(meme_cres)) != 0)
                        for (int64_t k = 0; k < m; ++k) if (memcmp(&V.res[k], &R.res[k], sizeof(meme_cres)) != 0) verify_fail("CIGAR (score / NM / lengths of a call)", P.done + k, g_chunk.seqs[g_chunk.part[(size_t)d].first + Jv[(size_t)(P.done + k)].read].name);
                    if (memcmp(V.cigars, R.cigars, (size_t)R.total_ops * 4) != 0 || memcmp(V.md, R.md, (size_t)R.md_bytes) != 0)
                        for (int64_t k = 0; k < m; ++k)
                            if (memcmp(V.cigars + R.res[k].cigar_off, R.cigars + R.res[k].cigar_off, (size_t)R.res[k].n_cigar * 4) != 0 || memcmp(V.md + R.res[k].md_off, R.md + R.res[k].md_off, (size_t)R.res[k].md_len + 1) != 0)
                                verify_fail("CIGAR (operations / MD string of a call)", P.done + k, g_chunk.seqs[g_chunk.part[(size_t)d].first + Jv[(size_t)(P.done + k)].read].name);
                    char st[48]; snprintf(st, sizeof(st), "cigar-round-%d%s", round, h ? "-half2" : "");
                    verify_note(g_chunk.seq, st, d, h, m);
                }
                // operations and MD string of every job back to back: the device packs both in job order, so 4 x cigar_off + md_off is a packing too
                const int64_t b0 = (int64_t)P.blob.size();
                P.blob.resize((size_t)(b0 + 4 * R.total_ops + R.md_bytes));
                const size_t r0 = P.res.size();
                P.res.resize(r0 + (size_t)m);
                char* const bl = P.blob.data() + b0;
TeamLabel tl_72393("cigar: take results");
                team_for(m, cig_threads() / (nd > 1 ? 2 : 1), [&](int64_t k_lo, int64_t k_hi, int) {
                for (int64_t k = k_lo; k < k_hi; ++k) {
                    meme_cres g = R.res[k];
                    const int64_t o = 4 * g.cigar_off + g.md_off;
                    memcpy(bl + o, R.cigars + g.cigar_off, (size_t)g.n_cigar * 4);
                    memcpy(bl + o + (int64_t)g.n_cigar * 4, R.md + g.md_off, (size_t)g.md_len + 1);
                    g.cigar_off = b0 + o;
                    P.res[r0 + (size_t)k] = g;
                }
                });
                P.kernel_ms += R.kernel_ms;
                P.done += m;
            }
        };
        for (int d = 1; d < nd; ++d) th.emplace_back(run, d);
        run(0);
        for (auto& t : th) t.join();
        t_call += now_s() - tp; tp = now_s();
        std::vector<Cand> next;
        for (int d = 0; d < nd; ++d) {
            const Part& R = part[(size_t)d];
            if (R.done == 0) continue;
            g_cig.t_kernel_ms += R.kernel_ms;
            g_cig.n_jobs += R.done;
            const size_t e0 = T.e.size(), b0 = T.blob.size();
            T.blob.insert(T.blob.end(), R.blob.begin(), R.blob.end());
            T.e.resize(e0 + (size_t)R.done);
            std::vector<uint8_t> again((size_t)R.done);
TeamLabel tl_12033("cigar: entries");
            team_for(R.done, cig_threads(), [&](int64_t k_lo, int64_t k_hi, int) {
            for (int64_t k = k_lo; k < k_hi; ++k) {
                const meme_cjob& J = jobs[(size_t)d][(size_t)k];
                const meme_cres& X = R.res[(size_t)k];
                CigEntry& E = T.e[e0 + (size_t)k];
                Cand& C = cand[who[(size_t)d][(size_t)k]];
                E.g = (int32_t)C.g; E.rb = J.rb; E.qb = J.qb; E.qlen = J.qlen; E.tlen = J.tlen; E.w_ = J.w_; E.pad = 0;
                E.score = X.score; E.n_cigar = X.n_cigar; E.nm = X.nm; E.md_len = X.md_len; E.blob = (int64_t)b0 + X.cigar_off;
                // mem_reg2aln's loop (:2340-2347): again with the doubled band while the global score stays below the local one
                again[(size_t)k] = 0;
                const int score = E.score;
                if (score == C.last_sc || C.w2 == opt->w << 2) continue;        // (inside the share's loop: the next alignment)
                C.last_sc = score;
                const int prev = C.w2;
                C.w2 <<= 1;
                // (a doubled 0 is the same call again: its answer is in the table already, and it ends the loop -- score == last_sc)
                if (++C.tries < 3 && score < C.truesc - opt->a && (C.w2 < opt->w << 2 ? C.w2 : opt->w << 2) != prev) again[(size_t)k] = 1;
            }
            });
            for (int64_t k = 0; k < R.done; ++k) if (again[(size_t)k]) next.push_back(cand[who[(size_t)d][(size_t)k]]);
        }
        cand.swap(next);
        t_take += now_s() - tp;
    }
    const double t_idx0 = now_s();
    // the table: open addressing over (rb, lengths, w_); the hook verifies the query bases of what it finds
    {
        uint64_t cap = 64;
        while (cap < 2 * T.e.size() + 16) cap <<= 1;
        T.slot.assign((size_t)cap, 0u);
        T.mask = cap - 1;
        uint32_t* sl = T.slot.data();
TeamLabel tl_2052("cigar: hash table");
        team_for((int64_t)T.e.size(), cig_threads(), [&](int64_t k_lo, int64_t k_hi, int) {
        for (int64_t k = k_lo; k < k_hi; ++k) {
            const CigEntry& E = T.e[(size_t)k];
            uint64_t h = cig_key(E.rb, E.qlen, E.tlen, E.w_) & T.mask;
            for (;;) {
                uint32_t empty = 0;
                if (__atomic_compare_exchange_n(&sl[h], &empty, (uint32_t)k + 1, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
                h = (h + 1) & T.mask;
            }
        }
        });
    }
    g_cig.t_prepass += now_s() - t0;
    if (verbose()) {
        timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts);
        fprintf(stderr, "[meme-dropin] CIGAR pre-pass of this chunk %.3f s (process CPU %.2f s): candidates %.3f, jobs posed %.3f, backend calls %.3f, results taken %.3f, table %.3f\n", now_s() - t0,
                (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec - cpu0, t_cand1 - t0, t_pose, t_call, t_take, now_s() - t_idx0);
    }
}

typedef uint32_t* (*gen_cigar2_fn)(const int8_t*, int, int, int, int, int, int64_t, const uint8_t*, int, uint8_t*, int64_t, int64_t, int*, int*, int*);
}  // namespace dropin

extern "C" uint32_t* bwa_gen_cigar2(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, int64_t l_pac, const uint8_t* pac, int l_query,
                                    uint8_t* query, int64_t rb, int64_t re, int* score, int* n_cigar, int* NM) {
    static const gen_cigar2_fn next = (gen_cigar2_fn)ref_sym(R_BWA_GEN_CIGAR2);
    PROF_SCOPE(P_GEN_CIGAR2_HOOK);
    const mem_opt_t* opt = g_opt;
    if (!cigar_on_device() || !score || !n_cigar || !NM || !g_chunk.seqs || !g_worker || !opt || g_dev.empty() || mat != opt->mat || o_del != opt->o_del ||
        e_del != opt->e_del || o_ins != opt->o_ins || e_ins != opt->e_ins || !g_bns || l_pac != g_bns->l_pac || l_query <= 0 || rb >= re || re - rb > 0x7fffffff)
        return next(mat, o_del, e_del, o_ins, e_ins, w_, l_pac, pac, l_query, query, rb, re, score, n_cigar, NM);
    const int tlen = (int)(re - rb);
    const uint64_t key = cig_key(rb, l_query, tlen, w_);
    for (int half = 0; half < 2; ++half) {               // (a read's calls are in its own half's table; the other one is looked at for what it may hold, once it is published)
    const CigTable& T = g_cig_tab[half];
    if (T.gen.load(std::memory_order_acquire) != g_chunk_gen) continue;
    for (uint64_t h = key & T.mask;; h = (h + 1) & T.mask) {
        const uint32_t s = T.slot[(size_t)h];
        if (!s) break;
        const CigEntry& E = T.e[s - 1];
        if (E.rb != rb || E.qlen != l_query || E.tlen != tlen || E.w_ != w_) continue;
        if (memcmp(g_chunk.seqs[E.g].seq + E.qb, query, (size_t)l_query) != 0) continue;        // (the read's bases are codes by now, as the caller's copy is)
        // the block the reference's function returns: the operations, the MD string right behind them (src/bwa.cpp:324, 352-354)
        const size_t bytes = (size_t)E.n_cigar * 4 + (size_t)E.md_len + 1;
        uint32_t* cg = (uint32_t*)malloc(bytes);
        if (!cg) { fprintf(stderr, "[meme-dropin] out of memory\n"); exit(1); }
        memcpy(cg, T.blob.data() + E.blob, bytes);
        *score = E.score; *n_cigar = E.n_cigar; *NM = E.nm;
        ++tl_cig.hits;
        return cg;
    }
    }
    ++tl_cig.miss;
    return next(mat, o_del, e_del, o_ins, e_ins, w_, l_pac, pac, l_query, query, rb, re, score, n_cigar, NM);
}

void meme_dropin_report_cigar() {
    if (!cigar_on_device()) return;
    fprintf(stderr, "[meme-dropin] CIGAR stage on the device: %lld alignments posed so far (kernels %.3f s, whole pre-pass %.3f s); "
            "bwa_gen_cigar2 calls answered from the table %lld, computed by the reference's function %lld (alignments made by mate rescue)\n",
            (long long)g_cig.n_jobs, g_cig.t_kernel_ms * 1e-3, g_cig.t_prepass, (long long)g_cig_hits.load(), (long long)g_cig_miss.load());
}

// ---- SAM text on the device ----------------------------------------------------------------------------------------------
// mem_aln2sam (src/bwamem.cpp:2174-2312) is what was left of the SAM phase on the host after pairing: 28 % of worker_sam's third step
// (profiles/r05_sam_phase_split.md) -- kputw / kputc by the dozen, SEQ and QUAL copied base by base -- and behind it an output step that
// handles one string per read.  The function is interposed: for a record that needs nothing but itself, its mate's record and the read
// (the only record of its read, no comment, no pa / XR tag: all but the chimeric reads) the hook notes a descriptor -- the numbers, the
// final CIGAR with its MD string, the mate's CIGAR, the XA string -- in the calling thread's arena and leaves a one-byte marker in the
// kstring (the caller makes s->sam of it as of any text).  When worker_sam has joined, the chunk's descriptors go to the GPU(s) in one
// meme_sam_format_batch_host call each (names, qualities and bases are in HBM since the seeding call) and the text comes back as one
// arena per device, in read order; the output step writes it from there.  Everything else goes to the reference's function as before.
// MEME_DROPIN_SAM=0: off.  MEME_DROPIN_SAM_CHECK=1 (tests): every such record is ALSO formatted by the reference's function and compared.
namespace dropin {
std::atomic<bool> g_sam_dev_failed{false};            // a device call of the stage failed for good: later chunks keep the reference's mem_aln2sam
bool sam_on_device() {
    // (only the environment is cached: ext_mode() may still be overridden by ext_mode_decide() when the first call comes early -- advisor, round 5)
    static const bool env_on = !(getenv("MEME_DROPIN_SAM") && atoi(getenv("MEME_DROPIN_SAM")) == 0);
    return env_on && fast_out() && ext_mode() == 2 && !g_sam_dev_failed.load(std::memory_order_relaxed);
}
bool sam_check() { static const bool v = getenv("MEME_DROPIN_SAM_CHECK") != nullptr; return v; }
// What the worker threads of one chunk noted: arenas handed out per thread and chunk, owned by the chunk's slot (the threads end with
// their kt_for call; the arenas stay until the output step has formatted them).
struct SamArena { std::vector<meme_sam_rec> recs; std::vector<uint8_t> blob; std::vector<std::pair<int64_t, std::string>> check; };
struct SamSlot {
    std::mutex mu;
    std::vector<std::unique_ptr<SamArena>> arenas;   // in use by the chunk being processed / waiting for the output step
    size_t used = 0;
    const bseq1_t* seqs = nullptr;                   // the chunk whose records these are (set when its worker_sam has joined)
    int64_t chunk_seq = -1, n = 0;
    Chunk* chunk = nullptr;
    int softclip = 0;
    std::string rg;
    // pinned staging of the device calls (grow-only), one set per device slice
    struct Stage { meme_sam_rec* recs = nullptr; int64_t recs_cap = 0; uint8_t* blob = nullptr; int64_t blob_cap = 0;
                   std::vector<char> own_text; std::vector<int64_t> own_off; };      // own_*: a slice formatted in pieces / by the reference's function (the device refused it whole)
    const mem_opt_t* opt = nullptr;
    std::vector<Stage> stage;
    SamText text;
} g_sam_slot[2];
thread_local SamArena* tl_sam_arena = nullptr;
thread_local uint64_t tl_sam_gen = 0;
SamArena& sam_arena() {                               // the calling thread's arena for the chunk being processed
    if (tl_sam_arena && tl_sam_gen == g_chunk_gen) return *tl_sam_arena;
    SamSlot& S = g_sam_slot[g_cur_chunk_seq & 1];
    std::lock_guard<std::mutex> lk(S.mu);
    if (S.used == S.arenas.size()) S.arenas.emplace_back(new SamArena);
    tl_sam_arena = S.arenas[S.used++].get();
    tl_sam_gen = g_chunk_gen;
    return *tl_sam_arena;
}
std::atomic<int64_t> g_sam_dev_recs{0}, g_sam_ref_recs{0}, g_sam_ref_fallback{0};
double g_sam_kernel_ms = 0, g_sam_stage_s = 0;
struct SamTally { int64_t dev = 0, ref = 0; ~SamTally() { if (dev) g_sam_dev_recs += dev; if (ref) g_sam_ref_recs += ref; } };
thread_local SamTally tl_sam_tally;
std::vector<char> g_contig_names; std::vector<int32_t> g_contig_name_off;
typedef void (*aln2sam_fn)(const mem_opt_t*, const bntseq_t*, kstring_t*, bseq1_t*, int, const mem_aln_t*, int, const mem_aln_t*);

// worker_sam of the chunk has joined: what its threads noted now belongs to the chunk's output step
void sam_chunk_closed() {
    SamSlot& S = g_sam_slot[g_cur_chunk_seq & 1];
    std::lock_guard<std::mutex> lk(S.mu);
    S.seqs = nullptr;
    bool any = false;
    for (size_t a = 0; a < S.used; ++a) any = any || !S.arenas[a]->recs.empty();
    if (!any) { S.used = 0; return; }
    S.seqs = g_chunk.seqs; S.chunk_seq = g_cur_chunk_seq; S.n = g_chunk.n; S.chunk = g_cur_chunk;
    S.softclip = (g_opt->flag & MEM_F_SOFTCLIP) ? 1 : 0; S.rg = bwa_rg_id; S.opt = g_opt;
    if (g_contig_name_off.empty()) {
        g_contig_name_off.push_back(0);
        for (int i = 0; i < g_bns->n_seqs; ++i) { const char* nm = g_bns->anns[i].name; g_contig_names.insert(g_contig_names.end(), nm, nm + strlen(nm)); g_contig_name_off.push_back((int32_t)g_contig_names.size()); }
        g_contig_names.push_back(0);
    }
}
bool sam_release_deferred(int64_t chunk_seq) {
    SamSlot& S = g_sam_slot[chunk_seq & 1];
    std::lock_guard<std::mutex> lk(S.mu);
    return S.seqs != nullptr && S.chunk_seq == chunk_seq;
}

// The output step's part (kt_pipeline step 2, meme_dropin_io.cpp): the chunk's noted records through the device.  One record slot per read
// of a device slice, in read order (a read without a noted record keeps read = -1: no text), assembled in pinned memory by the helper team;
// the text stays in the ctxs' pinned result buffers until sam_output_done().
SamText* sam_format_for_output(const bseq1_t* seqs) {
    SamSlot* Sp = nullptr;
    for (SamSlot& X : g_sam_slot) { std::lock_guard<std::mutex> lk(X.mu); if (X.seqs && X.seqs == seqs) Sp = &X; }
    if (!Sp) return nullptr;
    SamSlot& S = *Sp;
    const double t0 = now_s();
    Chunk& C = *S.chunk;
    const int nd = (int)C.part.size();
    if ((int)S.stage.size() < nd) S.stage.resize((size_t)nd);
    // where every arena's blob goes in the concatenation
    std::vector<int64_t> base(S.used + 1, 0);
    for (size_t a = 0; a < S.used; ++a) base[a + 1] = base[a] + (((int64_t)S.arenas[a]->blob.size() + 3) & ~(int64_t)3);
    const int64_t blob_bytes = base[S.used];
    for (int d = 0; d < nd; ++d) {
        SamSlot::Stage& G = S.stage[(size_t)d];
        const int64_t cnt = C.part[(size_t)d].count;
        if (cnt > G.recs_cap) { meme_host_free(G.recs); G.recs_cap = cnt + cnt / 4 + 64; if (!(G.recs = (meme_sam_rec*)meme_host_alloc(G.recs_cap * (int64_t)sizeof(meme_sam_rec)))) die("meme_host_alloc"); }
        if (d == 0 && blob_bytes + 64 > G.blob_cap) { meme_host_free(G.blob); G.blob_cap = blob_bytes + blob_bytes / 4 + 4096; if (!(G.blob = (uint8_t*)meme_host_alloc(G.blob_cap))) die("meme_host_alloc"); }
TeamLabel tl_59587("sam text: assemble descriptors");
        team_for(cnt, cig_threads(), [&](int64_t i0, int64_t i1, int) { for (int64_t i = i0; i < i1; ++i) G.recs[i].read = -1; });
    }
    uint8_t* const blob = S.stage[0].blob;               // (one blob for all slices: a slice's call ships it whole -- tens of MB)
    team_for((int64_t)S.used, cig_threads(), [&](int64_t a0, int64_t a1, int) {
        for (int64_t a = a0; a < a1; ++a) {
            const SamArena& A = *S.arenas[(size_t)a];
            if (!A.blob.empty()) memcpy(blob + base[(size_t)a], A.blob.data(), A.blob.size());
            for (meme_sam_rec r : A.recs) {
                const int64_t g = r.read;
                int d = 0;
                while (d + 1 < nd && g >= C.part[(size_t)d].first + C.part[(size_t)d].count) ++d;
                if (r.n_cigar > 0) r.cigar_off += base[(size_t)a];
                if (r.m_n_cigar > 0) r.m_cigar_off += base[(size_t)a];
                if (r.xa_off >= 0) r.xa_off += base[(size_t)a];
                r.read = (int32_t)(g - C.part[(size_t)d].first);
                S.stage[(size_t)d].recs[r.read] = r;
            }
        }
    });
    S.text.part.assign((size_t)nd, SamPart());
    std::vector<double> kms((size_t)nd, 0.0);
    // A slice the device cannot format whole is formatted in pieces (MEME_E_CAPACITY: halves, as the CIGAR stage does), and whatever the device
    // cannot format at all by the REFERENCE's own mem_aln2sam from the noted descriptors -- the hook left only a marker in s->sam, so the text must
    // come from somewhere; the stage is an optimisation, never a reason to stop after all the alignment work is done (advisor, round 5).
    static const aln2sam_fn ref_aln2sam = (aln2sam_fn)ref_sym(R_MEM_ALN2SAM);
    auto by_reference = [&](int d, int64_t k0, int64_t k1, std::vector<char>& text, std::vector<int64_t>& off) {
        const ChunkPart& P = C.part[(size_t)d];
        const meme_sam_rec* recs = S.stage[(size_t)d].recs;
        std::vector<std::string> tmp((size_t)(k1 - k0));
        team_for(k1 - k0, cig_threads(), [&](int64_t i0, int64_t i1, int) {
            std::vector<uint32_t> cg, mcg;
            for (int64_t i = i0; i < i1; ++i) {
                const meme_sam_rec& r = recs[k0 + i];
                if (r.read < 0) continue;
                mem_aln_t a, m;
                memset(&a, 0, sizeof(a)); memset(&m, 0, sizeof(m));
                a.pos = r.pos; a.rid = r.rid; a.flag = r.flag; a.is_rev = (uint32_t)r.is_rev; a.is_alt = (uint32_t)r.is_alt; a.mapq = (uint32_t)r.mapq; a.NM = (uint32_t)r.NM;
                a.n_cigar = r.n_cigar; a.score = r.score; a.sub = r.sub; a.alt_sc = 0;
                if (r.n_cigar > 0) {                                  // (blob offsets need not be aligned: the operations are copied, the MD string behind them with them)
                    const uint8_t* b = blob + r.cigar_off;
                    const size_t bytes = (size_t)r.n_cigar * 4 + strlen((const char*)b + (size_t)r.n_cigar * 4) + 1;
                    cg.resize((bytes + 3) / 4); memcpy(cg.data(), b, bytes); a.cigar = cg.data();
                }
                a.XA = r.xa_off >= 0 ? (char*)(blob + r.xa_off) : nullptr;
                if (r.has_mate) {
                    m.pos = r.m_pos; m.rid = r.m_rid; m.is_rev = (uint32_t)r.m_is_rev; m.is_alt = (uint32_t)r.m_is_alt; m.n_cigar = r.m_n_cigar;
                    if (r.m_n_cigar > 0) { mcg.resize((size_t)r.m_n_cigar); memcpy(mcg.data(), blob + r.m_cigar_off, (size_t)r.m_n_cigar * 4); m.cigar = mcg.data(); }
                }
                kstring_t t = {0, 0, 0};
                ref_aln2sam(S.opt, g_bns, &t, const_cast<bseq1_t*>(&seqs[P.first + k0 + i]), 1, &a, r.which, r.has_mate ? &m : nullptr);
                tmp[(size_t)i].assign(t.s, t.l);
                free(t.s);
            }
        });
        for (int64_t i = 0; i < k1 - k0; ++i) { text.insert(text.end(), tmp[(size_t)i].begin(), tmp[(size_t)i].end()); off[(size_t)(k0 + i + 1)] = (int64_t)text.size(); }
    };
    auto run = [&](int d) {
        const ChunkPart& P = C.part[(size_t)d];
        if (P.count == 0 || !P.sam_staged) return;
        SamSlot::Stage& G = S.stage[(size_t)d];
        SamPart& T = S.text.part[(size_t)d];
        T.first = P.first; T.count = P.count;
        auto call = [&](int64_t k0, int64_t m, meme_sam_host_result* R) {
            return meme_sam_format_batch_host(P.ctx, G.recs + k0, m, blob, blob_bytes, g_contig_names.data(), g_contig_name_off.data(), g_bns->n_seqs, S.softclip, S.rg.c_str(), R);
        };
        meme_sam_host_result R;
        int rc = g_sam_dev_failed.load() ? MEME_E_STATE : call(0, P.count, &R);
        if (rc == MEME_OK) {
        if (verify_on() && P.vfy) {                                    // MEME_DROPIN_VERIFY: the same records on the ctx that holds the same reads, names and qualities
                meme_sam_host_result V;
                if (meme_sam_format_batch_host(P.vfy, S.stage[(size_t)d].recs, P.count, blob, blob_bytes, g_contig_names.data(), g_contig_name_off.data(), g_bns->n_seqs, S.softclip, S.rg.c_str(), &V))
                    die("MEME_DROPIN_VERIFY: the second run of the SAM text stage");
                if (V.text_bytes != R.text_bytes) verify_fail("SAM text", -1, "total bytes");
                for (int64_t k = 0; k < P.count; ++k)
                    if (V.text_off[k + 1] != R.text_off[k + 1] || memcmp(V.text + R.text_off[k], R.text + R.text_off[k], (size_t)(R.text_off[k + 1] - R.text_off[k])) != 0) verify_fail("SAM text", k, seqs[P.first + k].name);
                verify_note(C.seq, "sam-text", d, verify_hash(R.text, (size_t)R.text_bytes), P.count);
            }
            T.text = R.text; T.text_off = R.text_off;
            kms[(size_t)d] = R.kernel_ms;
            return;
        }
        {
            static std::atomic<int> said{0};
            if (said.fetch_add(1) < 2) fprintf(stderr, "[meme-dropin] SAM text stage: the device does not take a slice of %lld records whole (%s): in pieces, the rest by the reference's mem_aln2sam\n", (long long)P.count, meme_last_error());
        }
        G.own_text.clear(); G.own_off.assign((size_t)P.count + 1, 0);
        int64_t done = 0, piece = (P.count + 1) / 2;
        while (done < P.count) {
            const int64_t m = piece < P.count - done ? piece : P.count - done;
            int prc = rc == MEME_E_CAPACITY ? call(done, m, &R) : rc;
            if (prc == MEME_E_CAPACITY && m > 1024) { piece = (m + 1) / 2; continue; }
            if (prc == MEME_OK) {
                const int64_t base = (int64_t)G.own_text.size();
                G.own_text.insert(G.own_text.end(), R.text, R.text + R.text_bytes);
                for (int64_t i = 0; i < m; ++i) G.own_off[(size_t)(done + i + 1)] = base + R.text_off[i + 1];
                kms[(size_t)d] += R.kernel_ms;
                done += m;
                continue;
            }
            if (prc != MEME_E_CAPACITY) g_sam_dev_failed = true;      // (not a question of size: the chunks that follow are not staged for the device at all)
            g_sam_ref_fallback += P.count - done;
            by_reference(d, done, P.count, G.own_text, G.own_off);
            done = P.count;
        }
        T.text = G.own_text.data(); T.text_off = G.own_off.data();
    };
    std::vector<std::thread> th;
    for (int d = 1; d < nd; ++d) th.emplace_back(run, d);
    run(0);
    for (auto& x : th) x.join();
    for (size_t a = 0; a < S.used; ++a)
        for (const auto& c : S.arenas[a]->check) {
            const int64_t g = c.first;
            int d = 0;
            while (d + 1 < nd && g >= C.part[(size_t)d].first + C.part[(size_t)d].count) ++d;
            const SamPart& T = S.text.part[(size_t)d];
            const char* t = T.text ? T.text + T.text_off[g - T.first] : nullptr;
            const int64_t l = T.text ? T.text_off[g - T.first + 1] - T.text_off[g - T.first] : 0;
            if (!t || (size_t)l != c.second.size() || memcmp(t, c.second.data(), c.second.size()) != 0) {
                fprintf(stderr, "[meme-dropin] SAM text of read %s differs between the device and the reference's mem_aln2sam:\n  device   : %.*s  reference: %s", seqs[g].name, (int)l, t ? t : "",
                        c.second.c_str());
                exit(1);
            }
        }
    double km = 0;
    for (double v : kms) km = km > v ? km : v;
    g_sam_kernel_ms += km; g_sam_stage_s += now_s() - t0;
    return &S.text;
}
void sam_output_done(const bseq1_t* seqs) {
    for (SamSlot& S : g_sam_slot) {
        int64_t seq = -1;
        {
            std::lock_guard<std::mutex> lk(S.mu);
            if (!S.seqs || S.seqs != seqs) continue;
            for (size_t a = 0; a < S.used; ++a) { S.arenas[a]->recs.clear(); S.arenas[a]->blob.clear(); S.arenas[a]->check.clear(); }
            S.used = 0; S.seqs = nullptr; seq = S.chunk_seq;
        }
        if (seq >= 0) prefetch_processed(seq);           // the slot's reads, seeds and staged text may be overwritten now
    }
}
}  // namespace dropin

void mem_aln2sam(const mem_opt_t* opt, const bntseq_t* bns, kstring_t* str, bseq1_t* s, int n, const mem_aln_t* list, int which, const mem_aln_t* m) {
    static const aln2sam_fn next = (aln2sam_fn)ref_sym(R_MEM_ALN2SAM);
    PROF_SCOPE(P_ALN2SAM);
    const mem_aln_t& p = list[which];
    const int64_t g = (g_chunk.seqs && g_cur_chunk_seq >= 0) ? s - g_chunk.seqs : -1;
    bool dev = sam_on_device() && n == 1 && which == 0 && g >= 0 && g < g_chunk.n && opt == g_opt && bns == g_bns && !s->comment && !(opt->flag & MEM_F_REF_HDR) && p.alt_sc <= 0 &&
               p.rid < bns->n_seqs && (!m || m->rid < bns->n_seqs) && p.n_cigar >= 0 && (!m || m->n_cigar >= 0) && (p.n_cigar == 0 || p.cigar) && (!m || m->n_cigar == 0 || m->cigar);
    if (dev) {
        int d = 0;
        const int nd = (int)g_chunk.part.size();
        while (d + 1 < nd && g >= g_chunk.part[(size_t)d].first + g_chunk.part[(size_t)d].count) ++d;
        dev = g_chunk.part[(size_t)d].sam_staged;
    }
    if (!dev) { ++tl_sam_tally.ref; next(opt, bns, str, s, n, list, which, m); return; }
    SamArena& A = sam_arena();
    meme_sam_rec r;
    memset(&r, 0, sizeof(r));
    r.read = (int32_t)g; r.flag = p.flag; r.rid = p.rid; r.pos = p.pos; r.is_rev = p.is_rev; r.is_alt = p.is_alt; r.mapq = p.mapq; r.NM = p.NM; r.score = p.score; r.sub = p.sub;
    r.n_cigar = p.n_cigar; r.which = which; r.xa_off = -1;
    if (p.n_cigar > 0) {                                   // the operations, the MD string right behind them (as mem_reg2aln leaves the block)
        const char* md = (const char*)(p.cigar + p.n_cigar);
        const size_t bytes = (size_t)p.n_cigar * 4 + strlen(md) + 1;
        r.cigar_off = (int64_t)A.blob.size();
        A.blob.insert(A.blob.end(), (const uint8_t*)p.cigar, (const uint8_t*)p.cigar + bytes);
    }
    if (m) {
        r.has_mate = 1; r.m_rid = m->rid; r.m_pos = m->pos; r.m_is_rev = m->is_rev; r.m_is_alt = m->is_alt; r.m_n_cigar = m->n_cigar;
        if (m->n_cigar > 0) { r.m_cigar_off = (int64_t)A.blob.size(); A.blob.insert(A.blob.end(), (const uint8_t*)m->cigar, (const uint8_t*)m->cigar + (size_t)m->n_cigar * 4); }
    }
    if (p.XA) { r.xa_off = (int64_t)A.blob.size(); A.blob.insert(A.blob.end(), (const uint8_t*)p.XA, (const uint8_t*)p.XA + strlen(p.XA) + 1); }
    A.recs.push_back(r);
    ++tl_sam_tally.dev;
    if (sam_check()) {
        kstring_t t = {0, 0, 0};
        next(opt, bns, &t, s, n, list, which, m);
        A.check.emplace_back(g, std::string(t.s, t.l));
        free(t.s);
    }
    kputc('\x01', str);                                    // the caller makes s->sam of the kstring: one byte that says "the text is the device's"
}
void meme_dropin_report_sam() {
    if (!sam_on_device()) return;
    fprintf(stderr, "[meme-dropin] SAM text on the device: %lld records formatted there so far (kernels %.3f s, whole stage %.3f s), %lld by the reference's mem_aln2sam "
            "(reads with several records, comments, pa tags); record slots the device refused and the reference's function formatted from the descriptors: %lld\n", (long long)g_sam_dev_recs.load(),
            g_sam_kernel_ms * 1e-3, g_sam_stage_s, (long long)g_sam_ref_recs.load(), (long long)g_sam_ref_fallback.load());
}

// ---- insert-size statistics: mem_pestat (src/bwamem_pair.cpp:81-148) ------------------------------------------------------------------
// Between worker_aln and worker_sam the reference walks the alignment records of every pair of the chunk in ONE thread -- 667 k scattered
// heap records, a cache miss each -- collects the insert sizes of the uniquely aligned pairs and sorts them: 0.03-0.045 s of a chunk's
// 0.19 s with every other thread idle.  The binding decides in parallel which pairs the function would take (its four `continue` tests,
// :96-99, with cal_sub :67-79) and hands the reference's own function a compact stand-in array: per such pair two one-record vectors with
// the pair's rid / rb (one record: cal_sub returns the floor and the score is set above it, so every stand-in passes the tests the
// pre-pass has already applied), ordered by orientation and insert size so that the function's sort meets sorted input.  What it then
// computes -- the `is <= max_ins` test, the sort, percentiles, mean, standard deviation, bounds, its messages -- is its own code on the
// same multiset of insert sizes.  MEME_DROPIN_PESTAT=0: the reference's walk.
namespace dropin {
bool pestat_fast() { static const bool v = !(getenv("MEME_DROPIN_PESTAT") && atoi(getenv("MEME_DROPIN_PESTAT")) == 0); return v; }
inline int pestat_cal_sub(const mem_opt_t* opt, const RecDigest* a, int n) {          // cal_sub, :67-79
    int j;
    for (j = 1; j < n; ++j) {
        const int b_max = a[j].qb > a[0].qb ? a[j].qb : a[0].qb;
        const int e_min = a[j].qe < a[0].qe ? a[j].qe : a[0].qe;
        if (e_min > b_max) {
            const int min_l = a[j].qe - a[j].qb < a[0].qe - a[0].qb ? a[j].qe - a[j].qb : a[0].qe - a[0].qb;
            if (e_min - b_max >= min_l * opt->mask_level) break;
        }
    }
    return j < n ? a[j].score : opt->min_seed_len * opt->a;
}
typedef void (*pestat_fn)(const mem_opt_t*, int64_t, int, const mem_alnreg_v*, mem_pestat_t*);
}  // namespace dropin
void mem_pestat(const mem_opt_t* opt, int64_t l_pac, int n, const mem_alnreg_v* regs, mem_pestat_t pes[4]) {
    static const pestat_fn next = (pestat_fn)ref_sym(R_MEM_PESTAT);
    const int np = n >> 1;
    if (!pestat_fast() || np < 64) { next(opt, l_pac, n, regs, pes); return; }
    struct Key { uint64_t k; int32_t i; };                       // (orientation << 60 | insert size, pair): the order of the stand-ins
    const int nt = cig_threads();
    std::vector<std::vector<Key>> part((size_t)nt);
    const ChunkDigest& D = chunk_digest(regs, n);
TeamLabel tl_54637("pestat: pre-filter + stand-ins");
    team_for(np, nt, [&](int64_t i_lo, int64_t i_hi, int t) {
        std::vector<Key>& mine = part[(size_t)t];
        for (int i = (int)i_lo; i < (int)i_hi; ++i) {
            const int64_t o0 = D.off[(size_t)i << 1], o1 = D.off[((size_t)i << 1) + 1], o2 = D.off[((size_t)i << 1) + 2];
            const RecDigest* r0 = D.rec + o0;
            const RecDigest* r1 = D.rec + o1;
            const int n0 = (int)(o1 - o0), n1 = (int)(o2 - o1);
            if (n0 == 0 || n1 == 0) continue;                                             // :96
            if (pestat_cal_sub(opt, r0, n0) > 0.8 * r0[0].score) continue;                // :97 (MIN_RATIO, :49)
            if (pestat_cal_sub(opt, r1, n1) > 0.8 * r1[0].score) continue;                // :98
            if (r0[0].rid != r1[0].rid) continue;                                         // :99
            // (only the order below depends on these two: mem_infer_dir, :58-65)
            const int64_t b1 = r0[0].rb, b2 = r1[0].rb;
            const int s1 = b1 >= l_pac, s2 = b2 >= l_pac;
            const int64_t p2 = s1 == s2 ? b2 : (l_pac << 1) - 1 - b2;
            const uint64_t dist = (uint64_t)(p2 > b1 ? p2 - b1 : b1 - p2);
            const int dir = (s1 == s2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
            mine.push_back({(uint64_t)dir << 60 | (dist & ((1ull << 60) - 1)), i});
        }
    });
    std::vector<Key> keys;
    { size_t tot = 0; for (auto& v : part) tot += v.size(); keys.reserve(tot); for (auto& v : part) keys.insert(keys.end(), v.begin(), v.end()); }
    // (pairs with equal keys are interchangeable stand-ins for the function -- same orientation, same insert size --, so the order among them does not matter)
    team_sort(keys, nt, [](const Key& x, const Key& y) { return x.k < y.k; });
    const int64_t m = (int64_t)keys.size();
    mem_alnreg_t* a = (mem_alnreg_t*)malloc((size_t)(2 * m + 1) * sizeof(mem_alnreg_t));
    mem_alnreg_v* v = (mem_alnreg_v*)malloc((size_t)(2 * m + 1) * sizeof(mem_alnreg_v));
    if (!a || !v) { fprintf(stderr, "[meme-dropin] out of memory\n"); exit(1); }
    team_for(m, nt, [&](int64_t k_lo, int64_t k_hi, int) {
    for (int64_t k = k_lo; k < k_hi; ++k)
        for (int e = 0; e < 2; ++e) {
            const RecDigest& src = D.rec[D.off[(size_t)(keys[(size_t)k].i << 1 | e)]];
            mem_alnreg_t& d = a[2 * k + e];
            d.rb = src.rb; d.rid = src.rid; d.score = 1 << 28;
            v[2 * k + e].n = v[2 * k + e].m = 1; v[2 * k + e].a = &d;
        }
    });
    next(opt, l_pac, (int)(2 * m), v, pes);
    free(a); free(v);
}

// kt_for (src/kthread.cpp:79-114) is called three times per chunk by mem_process_seqs: worker_bwt, worker_aln, worker_sam.  Before the
// third call every alignment record of the chunk exists and no worker thread is running: the CIGAR stage's quiescent point.
namespace dropin { std::atomic<int> g_ktfor_calls{0}; std::atomic<int>& ktfor_calls() { return g_ktfor_calls; } }
typedef void (*kt_for_fn)(void (*)(void*, long, long, int), void*, int);
namespace dropin {
// the second half of a chunk's worker_sam phase through the reference's kt_for, which counts its batches from read 0
struct HalfShim { void (*func)(void*, long, long, int) = nullptr; long off = 0; } g_half_shim;
void sam_half_shim(void* data, long seqid, long batch_size, int tid) { g_half_shim.func(data, seqid + g_half_shim.off, batch_size, tid); }
}
void kt_for(void (*func)(void*, long, long, int), void* data, int n) {
    static const kt_for_fn next = (kt_for_fn)ref_sym(R_KT_FOR);
    // (verbose runs: where a chunk's time inside mem_process_seqs goes -- the three worker phases and what lies between them)
    // (wall seconds and, in brackets, CPU seconds of the whole process -- helper threads of the binding included: what a host with a CPU quota is short of)
    static double t_ph[8], c_ph[8];
    const int call = (g_chunk.seqs && data == (void*)g_worker) ? g_ktfor_calls.load() : -1;
    auto cpu_s = [] { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    struct Phase { int c; double *t, *u; double (*cpu)(); ~Phase() {
        if (c < 0 || c > 2) return;
        t[2 * c + 1] = now_s(); u[2 * c + 1] = cpu();
        if (c == 2 && verbose())
            fprintf(stderr, "[meme-dropin] phases of this chunk, wall (process CPU) s: worker_bwt %.3f (%.2f), worker_aln %.3f (%.2f), mem_pestat %.3f (%.2f), CIGAR + mate-rescue pre-passes and worker_sam %.3f (%.2f)\n",
                    t[1] - t[0], u[1] - u[0], t[3] - t[2], u[3] - u[2], t[4] - t[3], u[4] - u[3], t[5] - t[4], u[5] - u[4]);
    } } phase{call, t_ph, c_ph, +cpu_s};
    if (call >= 0 && call <= 2) { t_ph[2 * call] = now_s(); c_ph[2 * call] = cpu_s(); }
    // worker_sam is the third worker function a chunk's mem_process_seqs hands to kt_for; the first chunk tells which pointer that is, later
    // calls are recognised by it (another kt_for call somewhere does not shift the count)
    static void (*fn_sam)(void*, long, long, int) = nullptr;
    bool is_sam = false;
    if (g_chunk.seqs && data == (void*)g_worker) {
        const int c = g_ktfor_calls.fetch_add(1);
        if (!fn_sam && c == 2) fn_sam = func;
        is_sam = fn_sam != nullptr && func == fn_sam;
    }
    if (is_sam) {
        bool mate = false;
#if __AVX512BW__          // (only this build of the reference batches mate rescue: src/bwamem.cpp:1838)
        mate = matesw_on_device() && (g_opt->flag & MEM_F_PE) && !(g_opt->flag & MEM_F_NO_RESCUE) && !g_dev.empty();
#endif
        // (a chunk that will not reach the job threshold -- by the previous chunk's jobs per read -- is not posed at all)
        if (mate && g_mate_jobs_per_read >= 0 && g_mate_jobs_per_read * (double)g_chunk.n < (double)matesw_min_jobs()) mate = false;
        // The two pre-passes side by side (they read the same alignment records, write tables of their own and use different ctxs): while
        // one waits for its kernels the other poses its jobs -- host time that nobody used.  MEME_DROPIN_PREPASS_OVERLAP=0: one after the other.
        static const bool overlap = !(getenv("MEME_DROPIN_PREPASS_OVERLAP") && atoi(getenv("MEME_DROPIN_PREPASS_OVERLAP")) == 0);
        // Round 6: the phase in two halves.  The pre-passes are a chain of waits for the GPU (a chunk's 24 M CIGAR calls and 46 k mate-rescue jobs:
        // 40-50 ms of which the kernels are 20) in front of worker_sam, which is all host work: with the chunk split at a worker-batch boundary
        // the second half's pre-passes run BESIDE the first half's worker_sam -- tables per half, the hooks look at both -- and only the first
        // half's pre-passes stay on the chunk's critical path.  worker_sam itself runs through the reference's kt_for as before, once per half
        // (the second call through a shim that adds the half's first read).  MEME_DROPIN_HALVES=0: the whole chunk at once.
        static const bool halves_on = !(getenv("MEME_DROPIN_HALVES") && atoi(getenv("MEME_DROPIN_HALVES")) == 0);
        const int64_t nb = ((int64_t)n + BATCH_SIZE - 1) / BATCH_SIZE;
        const bool paired = (g_opt->flag & MEM_F_PE) != 0;
        const int64_t n1 = (halves_on && nb >= 8 && (cigar_on_device() || mate)) ? (nb / 2) * BATCH_SIZE : (int64_t)n;
        g_half[0].lo = 0; g_half[0].hi = n1; g_half[1].lo = n1; g_half[1].hi = n;
        (void)paired;
        auto prepass = [&](int h) -> bool {
            bool mate_ok = false;
            std::thread mate_th;
            if (mate && overlap && cigar_on_device()) mate_th = std::thread([&mate_ok, h] { mate_ok = matesw_prepass(h); });
            if (cigar_on_device()) {
                std::lock_guard<std::mutex> lk(g_cig_tab[h].mu);
                cig_prepass(h);
                g_cig_tab[h].gen.store(g_chunk_gen, std::memory_order_release);
            }
            if (mate_th.joinable()) mate_th.join();
            else if (mate) mate_ok = matesw_prepass(h);
            return mate_ok;
        };
        const bool mate0 = prepass(0);
        if (n1 == (int64_t)n) next(mate0 ? sam_worker_dev : func, data, n);
        else {
            bool mate1 = false;
            std::thread second([&] { mate1 = prepass(1); });
            next(mate0 ? sam_worker_dev : func, data, (int)n1);
            second.join();
            g_half_shim.func = mate1 ? sam_worker_dev : func;
            g_half_shim.off = (long)n1;
            next(sam_half_shim, data, (int)(n - n1));
        }
        if (sam_on_device() && g_cur_chunk_seq >= 0) sam_chunk_closed();      // the worker threads have joined: their descriptors are the output step's now
        return;
    }
    next(func, data, n);
}
