// Part of the reference-side binding of the MI355X backend (see meme_dropin.h / meme_dropin.cpp).
#include "meme_dropin.h"

using namespace dropin;

// ---- FASTQ input (first step): the two mate files parsed by two threads, ahead of the pipeline ---------------------
// bseq_read_orig() (src/bwa.cpp:184-230) parses both files of a paired run with one thread, read by read; with the backend bound
// that parser is the longest stage of the aligner's three-stage pipeline (0.9 s per 100 M-base chunk against 0.5-0.7 s of compute).
// The records come from the same kseq_read() calls on the same streams, in the same order -- only that each stream has a thread
// of its own that keeps a bounded queue filled (records travel in batches of 4 096), and the pipeline's step 0 takes what is ready.
// MEME_DROPIN_IO=0 switches it off (the reference's reader).
#include <deque>
#include <map>
#include <sys/uio.h>
#include <unistd.h>
#include "profiling.h"
namespace dropin {

// ---- record strings without a malloc each (round 4) ---------------------------------------------------------------------------------
// The reference's output step (kt_pipeline step 2, src/fastmap.cpp:843-862) writes a chunk's SAM text record by record and frees five
// strings per read in ONE thread: 1.67 s per 4 M reads, three times the compute stages of the bound aligner -- what a run's wall time was
// made of.  The binding interposes that step (below).  Name, bases and qualities of a record then never needed to be strings of their own:
// they stay where the parser thread put them, in the batch's text arena, and the arenas of a chunk are released together when the chunk
// has been written.  (Comments stay malloc'ed strings: step 0 frees them one by one unless -C is given.)
bool fast_out() { static const bool v = !(getenv("MEME_DROPIN_OUT") && atoi(getenv("MEME_DROPIN_OUT")) == 0); return v; }
struct ArenaRef { std::vector<std::shared_ptr<void>> batches; };
std::mutex g_arena_mu;
std::map<const bseq1_t*, ArenaRef> g_arenas;         // chunk (its record array) -> the text arenas its records point into

struct ReadQueue {
    // Records travel in batches: one lock + one wake-up per BATCH records (per-record locking cost more than the parsing it was meant to
    // hide).  The parser thread keeps a batch's text in ONE arena; the strings the reference frees one by one (free(seqs[i].name) ... in
    // its output step) are allocated by the caller of bseq_read_orig, as in the reference -- strings allocated by the parser threads
    // would be freed into those threads' malloc arenas while they allocate from them (measured: the SAM-writing step 3x slower).
    static constexpr int BATCH = 4096;
    struct Rec { uint32_t name, name_l, comment, comment_l, seq, seq_l, qual, qual_l; };     // offsets into the arena; comment / qual: *_l == UINT32_MAX when absent
    struct Batch { std::vector<char> text; std::vector<Rec> recs; int64_t bases = 0; };
    std::mutex m;
    std::condition_variable cv_put, cv_get;
    std::deque<Batch> q;
    int64_t bases = 0;                                          // parsed and not yet taken
    bool eof = false;
    kseq_t* ks = nullptr;
    std::thread th;
    int64_t LIMIT = 100000000;                                 // bases parsed ahead per stream (set to the chunk size on the first call)
    std::shared_ptr<Batch> cur;                                 // the consumer's current batch (shared: a batch may straddle two chunks)
    size_t cur_i = 0;
    static uint32_t put(std::vector<char>& t, const char* p, size_t l) { const uint32_t o = (uint32_t)t.size(); t.insert(t.end(), p, p + l); t.push_back(0); return o; }
    void run() {
        Batch b;
        b.recs.reserve(BATCH);
        for (;;) {
            const bool got = kseq_read(ks) >= 0;
            if (got) {                                           // trim_readno, src/bwa.cpp:66-70
                if (ks->name.l > 2 && ks->name.s[ks->name.l - 2] == '/' && isdigit((unsigned char)ks->name.s[ks->name.l - 1])) { ks->name.l -= 2; ks->name.s[ks->name.l] = 0; }
                Rec r;
                r.name_l = (uint32_t)strlen(ks->name.s); r.name = put(b.text, ks->name.s, r.name_l);          // (strdup: up to the first NUL)
                if (ks->comment.l) { r.comment_l = (uint32_t)strlen(ks->comment.s); r.comment = put(b.text, ks->comment.s, r.comment_l); } else { r.comment = 0; r.comment_l = UINT32_MAX; }
                r.seq_l = (uint32_t)strlen(ks->seq.s); r.seq = put(b.text, ks->seq.s, r.seq_l);
                if (ks->qual.l) { r.qual_l = (uint32_t)strlen(ks->qual.s); r.qual = put(b.text, ks->qual.s, r.qual_l); } else { r.qual = 0; r.qual_l = UINT32_MAX; }
                b.recs.push_back(r);
                b.bases += r.seq_l < (uint32_t)ERT_MAX_READ_LEN ? r.seq_l : (uint32_t)ERT_MAX_READ_LEN;
            }
            if (!got || (int)b.recs.size() == BATCH) {
                std::unique_lock<std::mutex> lk(m);
                if (!b.recs.empty()) {
                    cv_put.wait(lk, [&] { return bases < LIMIT; });
                    bases += b.bases;
                    q.push_back(std::move(b));
                    b = Batch(); b.recs.reserve(BATCH);
                }
                if (!got) eof = true;
                cv_get.notify_all();
                if (!got) return;
            }
        }
    }
    static char* dup(const char* p, uint32_t l) { char* s = (char*)malloc((size_t)l + 1); if (!s) { fprintf(stderr, "[meme-dropin] out of memory\n"); exit(1); } memcpy(s, p, (size_t)l + 1); return s; }
    // `arena` != null: name / seq / qual point into the batch's text (kept alive through *arena); else they are strings of their own
    bool pop(bseq1_t& out, ArenaRef* arena) {                    // false: the stream is exhausted.  kseq2bseq1, src/bwa.cpp:82-89
        if (!cur || cur_i == cur->recs.size()) {
            std::unique_lock<std::mutex> lk(m);
            cv_get.wait(lk, [&] { return !q.empty() || eof; });
            if (q.empty()) return false;
            cur = std::make_shared<Batch>(std::move(q.front()));
            q.pop_front();
            cur_i = 0;
            bases -= cur->bases;
            cv_put.notify_one();
        }
        const Rec& r = cur->recs[cur_i++];
        char* t = cur->text.data();
        memset(&out, 0, sizeof(out));
        out.comment = r.comment_l == UINT32_MAX ? 0 : dup(t + r.comment, r.comment_l);
        if (arena) {
            if (arena->batches.empty() || arena->batches.back().get() != (void*)cur.get()) arena->batches.push_back(std::static_pointer_cast<void>(cur));
            out.name = t + r.name; out.seq = t + r.seq; out.qual = r.qual_l == UINT32_MAX ? 0 : t + r.qual;
        } else {
            out.name = dup(t + r.name, r.name_l);
            out.seq = dup(t + r.seq, r.seq_l);
            out.qual = r.qual_l == UINT32_MAX ? 0 : dup(t + r.qual, r.qual_l);
        }
        out.l_seq = (int)(r.seq_l < (uint32_t)ERT_MAX_READ_LEN ? r.seq_l : (uint32_t)ERT_MAX_READ_LEN);   // strnlen_s(s->seq, ERT_MAX_READ_LEN)
        return true;
    }
};
ReadQueue* g_rq[2] = {nullptr, nullptr};
void* g_rq_ks[2] = {nullptr, nullptr};
typedef bseq1_t* (*bseq_read_fn)(int64_t, int*, void*, void*, int64_t*);

}  // namespace dropin

namespace dropin {
// One chunk as bseq_read_orig (src/bwa.cpp:184-230) forms it: whole pairs until `chunk_size` bases are reached.
bseq1_t* assemble_chunk(int64_t chunk_size, int64_t* n_, int64_t* size_) {
    int64_t size = 0, m = 0, n = 0;
    bseq1_t* seqs = 0;
    bseq1_t a, b;
    ArenaRef arena_store, *arena = fast_out() ? &arena_store : nullptr;
    while (g_rq[0]->pop(a, arena)) {
        if (g_rq[1] && !g_rq[1]->pop(b, arena)) {                   // the 2nd file has fewer reads (:190-193)
            fprintf(stderr, "[W::%s] the 2nd file has fewer sequences.\n", "bseq_read_orig");
            break;
        }
        if (n + 1 >= m) { m = m ? m << 1 : 256; seqs = (bseq1_t*)realloc(seqs, (size_t)m * sizeof(bseq1_t)); }
        a.id = (int)n; seqs[n] = a; size += seqs[n++].l_seq;
        if (g_rq[1]) { b.id = (int)n; seqs[n] = b; size += seqs[n++].l_seq; }
        if (size >= chunk_size && (n & 1) == 0) break;
    }
    if (size == 0) {                                                // test if the 2nd file is finished (:223-226)
        if (g_rq[1] && g_rq[1]->pop(b, nullptr)) { fprintf(stderr, "[W::%s] the 1st file has fewer sequences.\n", "bseq_read_orig"); free(b.name); free(b.comment); free(b.seq); free(b.qual); }
        for (int k = 0; k < 2; ++k)                                 // end of the input: the parsers finish before the caller destroys the streams
            if (g_rq[k] && g_rq[k]->th.joinable()) {
                while (g_rq[k]->pop(b, nullptr)) { free(b.name); free(b.comment); free(b.seq); free(b.qual); }
                g_rq[k]->th.join();
            }
    }
    *n_ = n; *size_ = size;
    if (arena && seqs) { std::lock_guard<std::mutex> lk(g_arena_mu); g_arenas[seqs] = std::move(arena_store); }
    if (n > 0) prefetch_submit(seqs, n);                            // the chunk's device stages may start as soon as its slot is free
    return seqs;
}

// The NEXT chunk is formed as soon as the current one has been handed out (the pipeline asks for it only after it has written the chunk
// before the current one): its device stages then have the whole of the current chunk's host phases to run beside, and the pipeline's
// read step finds the record array ready.  The chunk size of a run is one number (aux->task_size, src/fastmap.cpp:743).
struct Ahead {
    std::mutex m; std::condition_variable cv;
    std::thread th;
    int64_t chunk_size = 0;
    bool ready = false;
    bseq1_t* seqs = nullptr; int64_t n = 0, size = 0;
};
Ahead& g_ahead = *new Ahead;        // (on the heap for good: a run that stops early must not meet a joinable thread's destructor)
}  // namespace dropin

extern "C" bseq1_t* bseq_read_orig(int64_t chunk_size, int* n_, void* ks1_, void* ks2_, int64_t* s) {
    static const bool on = !(getenv("MEME_DROPIN_IO") && atoi(getenv("MEME_DROPIN_IO")) == 0);
    static const bseq_read_fn next = (bseq_read_fn)ref_sym(R_BSEQ_READ_ORIG);
    // only the run's read files (the first streams seen); any other caller gets the reference's function
    if (on && !g_rq[0] && ks1_) {
        for (int k = 0; k < 2; ++k) {
            void* ks = k ? ks2_ : ks1_;
            if (!ks) continue;
            g_rq_ks[k] = ks;
            g_rq[k] = new ReadQueue;
            g_rq[k]->ks = (kseq_t*)ks;
            g_rq[k]->LIMIT = chunk_size > 1000000 ? chunk_size : 1000000;
            g_rq[k]->th = std::thread([k] { pthread_setname_np(pthread_self(), "meme-parser"); g_rq[k]->run(); note_thread_cpu("meme-parser"); });
        }
        g_ahead.chunk_size = chunk_size;
        g_ahead.th = std::thread([] {
            pthread_setname_np(pthread_self(), "meme-ahead");
            Ahead& A = g_ahead;
            for (;;) {
                int64_t n = 0, size = 0;
                bseq1_t* seqs = assemble_chunk(A.chunk_size, &n, &size);
                std::unique_lock<std::mutex> lk(A.m);
                A.seqs = seqs; A.n = n; A.size = size; A.ready = true;
                A.cv.notify_all();
                if (size == 0) { note_thread_cpu("meme-ahead"); return; }      // (the end of the input: handed out once, then the thread is joined)
                A.cv.wait(lk, [&] { return !A.ready; });
            }
        });
    }
    if (!on || ks1_ != g_rq_ks[0] || ks2_ != g_rq_ks[1]) {
        if (!next) { fprintf(stderr, "[meme-dropin] the reference's bseq_read_orig was not found\n"); exit(1); }
        return next(chunk_size, n_, ks1_, ks2_, s);
    }
    Ahead& A = g_ahead;
    if (chunk_size != A.chunk_size) { fprintf(stderr, "[meme-dropin] bseq_read_orig: the chunk size changed during the run (%lld, was %lld)\n", (long long)chunk_size, (long long)A.chunk_size); exit(1); }
    bseq1_t* seqs;
    {
        std::unique_lock<std::mutex> lk(A.m);
        if (!A.th.joinable() && !A.ready) { *n_ = 0; *s = 0; return 0; }      // (asked again after the end of the input)
        A.cv.wait(lk, [&] { return A.ready; });
        seqs = A.seqs; *n_ = (int)A.n; *s = A.size;
        A.ready = false;
        A.cv.notify_all();
    }
    if (*s == 0 && A.th.joinable()) A.th.join();
    return seqs;
}

// ---- the output step ---------------------------------------------------------------------------------------------------------------
// kt_pipeline(shared, 2, data) (src/fastmap.cpp:843-862): the chunk's SAM records to the output stream in read order, then the chunk's
// memory back.  Here: the records' lengths by a few helper threads, the text straight from the records to the file descriptor with
// writev() (no copy through stdio's buffer), the SAM strings freed by the helper threads (they come from the arenas of all the worker
// threads), name / bases / qualities released with their text arenas.  Steps 0 and 1 are the reference's.  MEME_DROPIN_OUT=0: all of it.
typedef ktp_data_t* (*ktp_step_fn)(void*, int, void*, mem_opt_t*, worker_t&);
ktp_data_t* kt_pipeline(void* shared, int step, void* data, mem_opt_t* opt, worker_t& w) {
    static const ktp_step_fn next = (ktp_step_fn)ref_sym(R_KT_PIPELINE);
    if (step != 2 || !fast_out()) return next(shared, step, data, opt, w);
    ktp_aux_t* aux = (ktp_aux_t*)shared;
    ktp_data_t* ret = (ktp_data_t*)data;
    // (tests: MEME_DROPIN_TEST_OUTPUT_DELAY_MS holds this step back before it touches the pipeline's read counter, so that the next chunk's PROCESS step reads
    // the counter first -- the order the reference's own race allows and a loaded host produces now and then; see mem_process_seqs in meme_dropin.cpp)
    static const int delay_ms = getenv("MEME_DROPIN_TEST_OUTPUT_DELAY_MS") ? atoi(getenv("MEME_DROPIN_TEST_OUTPUT_DELAY_MS")) : 0;
    if (delay_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
    aux->n_processed += ret->n_seqs;
    const uint64_t tim = __rdtsc();
    const int n = ret->n_seqs;
    bseq1_t* seqs = ret->seqs;
    ArenaRef arena;
    bool in_arena = false;
    {
        std::lock_guard<std::mutex> lk(g_arena_mu);
        auto it = g_arenas.find(seqs);
        if (it != g_arenas.end()) { arena = std::move(it->second); g_arenas.erase(it); in_arena = true; }
    }
    const int nt = cig_threads() < 8 ? cig_threads() : 8;
    const double t0 = now_s();
    std::vector<struct iovec> iov((size_t)n);
    size_t n_iov = 0;
    // a read whose record the device formatted (meme_dropin_sam.cpp) has its text in the chunk's arena, in read order: neighbours are adjacent
    // there and leave as one piece; everybody else's s->sam is the text
    SamText* dev_text = sam_format_for_output(seqs);
    TeamLabel lbl_("output step: lengths + frees");
    team_for(n, nt, [&](int64_t i0, int64_t i1, int) {
        size_t d = 0;
        for (int64_t i = i0; i < i1; ++i) {
            const char* t = nullptr;
            int64_t l = 0;
            if (dev_text) {
                while (d + 1 < dev_text->part.size() && i >= dev_text->part[d].first + dev_text->part[d].count) ++d;
                const SamPart& P = dev_text->part[d];
                if (P.text && i >= P.first && i < P.first + P.count) { l = P.text_off[i - P.first + 1] - P.text_off[i - P.first]; t = P.text + P.text_off[i - P.first]; }
            }
            if (l > 0) { iov[(size_t)i].iov_base = (void*)t; iov[(size_t)i].iov_len = (size_t)l; }
            else {
                // (a read the mem_aln2sam hook handled carries the one-byte marker instead of text: it must have come back from the device stage --
                // a marker reaching the output would be a 0x01 byte in the SAM stream; advisor, round 5)
                if (seqs[i].sam && seqs[i].sam[0] == '\x01') { fprintf(stderr, "[meme-dropin] read %s: its SAM record was left to the device's text stage, which returned nothing for it\n", seqs[i].name); exit(1); }
                iov[(size_t)i].iov_base = seqs[i].sam; iov[(size_t)i].iov_len = seqs[i].sam ? strlen(seqs[i].sam) : 0;
            }
        }
    });
    for (int i = 0; i < n; ++i) {
        if (!iov[(size_t)i].iov_len) continue;
        if (n_iov && (char*)iov[n_iov - 1].iov_base + iov[n_iov - 1].iov_len == (char*)iov[(size_t)i].iov_base) iov[n_iov - 1].iov_len += iov[(size_t)i].iov_len;
        else iov[n_iov++] = iov[(size_t)i];
    }
    const double t1 = now_s();
    fflush(aux->fp);                                                // (whatever stdio still holds goes first)
    const int fd = fileno(aux->fp);
    for (size_t k = 0; k < n_iov;) {
        const int cnt = (int)(n_iov - k < 1024 ? n_iov - k : 1024);
        ssize_t wr = writev(fd, &iov[k], cnt);
        if (wr < 0) { if (errno == EINTR) continue; perror("[meme-dropin] writing the SAM output"); exit(1); }
        while (wr > 0 && k < n_iov) {                               // (a short write: go on inside the record it stopped in)
            if ((size_t)wr >= iov[k].iov_len) { wr -= (ssize_t)iov[k].iov_len; ++k; }
            else { iov[k].iov_base = (char*)iov[k].iov_base + wr; iov[k].iov_len -= (size_t)wr; wr = 0; }
        }
    }
    const double t2 = now_s();
    sam_output_done(seqs);                                          // (the device's text has left: the chunk's slot is free for the next chunk's device stages)
    team_for(n, nt, [&](int64_t i0, int64_t i1, int) {
        for (int64_t i = i0; i < i1; ++i) {
            free(seqs[i].sam); free(seqs[i].comment);
            if (!in_arena) { free(seqs[i].name); free(seqs[i].seq); free(seqs[i].qual); }
        }
    });
    arena.batches.clear();
    free(seqs);
    free(ret);
    if (verbose()) fprintf(stderr, "[meme-dropin] output step of %d reads: lengths %.3f s, writev %.3f s, frees %.3f s\n", n, t1 - t0, t2 - t1, now_s() - t2);
    tprof[SAM_IO][0] += __rdtsc() - tim;
    return 0;
}
