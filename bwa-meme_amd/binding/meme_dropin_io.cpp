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
namespace dropin {

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
    Batch cur;                                                  // the consumer's current batch
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
    bool pop(bseq1_t& out) {                                     // false: the stream is exhausted.  kseq2bseq1, src/bwa.cpp:82-89
        if (cur_i == cur.recs.size()) {
            std::unique_lock<std::mutex> lk(m);
            cv_get.wait(lk, [&] { return !q.empty() || eof; });
            if (q.empty()) return false;
            cur = std::move(q.front());
            q.pop_front();
            cur_i = 0;
            bases -= cur.bases;
            cv_put.notify_one();
        }
        const Rec& r = cur.recs[cur_i++];
        const char* t = cur.text.data();
        memset(&out, 0, sizeof(out));
        out.name = dup(t + r.name, r.name_l);
        out.comment = r.comment_l == UINT32_MAX ? 0 : dup(t + r.comment, r.comment_l);
        out.seq = dup(t + r.seq, r.seq_l);
        out.qual = r.qual_l == UINT32_MAX ? 0 : dup(t + r.qual, r.qual_l);
        out.l_seq = (int)(r.seq_l < (uint32_t)ERT_MAX_READ_LEN ? r.seq_l : (uint32_t)ERT_MAX_READ_LEN);   // strnlen_s(s->seq, ERT_MAX_READ_LEN)
        return true;
    }
};
ReadQueue* g_rq[2] = {nullptr, nullptr};
void* g_rq_ks[2] = {nullptr, nullptr};
typedef bseq1_t* (*bseq_read_fn)(int64_t, int*, void*, void*, int64_t*);

}  // namespace dropin

extern "C" bseq1_t* bseq_read_orig(int64_t chunk_size, int* n_, void* ks1_, void* ks2_, int64_t* s) {
    static const bool on = !(getenv("MEME_DROPIN_IO") && atoi(getenv("MEME_DROPIN_IO")) == 0);
    static bseq_read_fn next = (bseq_read_fn)dlsym(RTLD_NEXT, "bseq_read_orig");
    // only the run's read files (the first streams seen); any other caller gets the reference's function
    if (on && !g_rq[0] && ks1_) {
        for (int k = 0; k < 2; ++k) {
            void* ks = k ? ks2_ : ks1_;
            if (!ks) continue;
            g_rq_ks[k] = ks;
            g_rq[k] = new ReadQueue;
            g_rq[k]->ks = (kseq_t*)ks;
            g_rq[k]->LIMIT = chunk_size > 1000000 ? chunk_size : 1000000;
            g_rq[k]->th = std::thread([k] { g_rq[k]->run(); });
        }
    }
    if (!on || ks1_ != g_rq_ks[0] || ks2_ != g_rq_ks[1]) {
        if (!next) { fprintf(stderr, "[meme-dropin] the reference's bseq_read_orig was not found\n"); exit(1); }
        return next(chunk_size, n_, ks1_, ks2_, s);
    }
    int64_t size = 0, m = 0, n = 0;
    bseq1_t* seqs = 0;
    bseq1_t a, b;
    while (g_rq[0]->pop(a)) {
        if (g_rq[1] && !g_rq[1]->pop(b)) {                          // the 2nd file has fewer reads (:190-193)
            fprintf(stderr, "[W::%s] the 2nd file has fewer sequences.\n", __func__);
            break;
        }
        if (n + 1 >= m) { m = m ? m << 1 : 256; seqs = (bseq1_t*)realloc(seqs, (size_t)m * sizeof(bseq1_t)); }
        a.id = (int)n; seqs[n] = a; size += seqs[n++].l_seq;
        if (g_rq[1]) { b.id = (int)n; seqs[n] = b; size += seqs[n++].l_seq; }
        if (size >= chunk_size && (n & 1) == 0) break;
    }
    if (size == 0) {                                                // test if the 2nd file is finished (:223-226)
        if (g_rq[1] && g_rq[1]->pop(b)) fprintf(stderr, "[W::%s] the 1st file has fewer sequences.\n", __func__);
        for (int k = 0; k < 2; ++k)                                 // end of the input: the parsers finish before the caller destroys the streams
            if (g_rq[k] && g_rq[k]->th.joinable()) {
                while (g_rq[k]->pop(b)) { free(b.name); free(b.comment); free(b.seq); free(b.qual); }
                g_rq[k]->th.join();
            }
    }
    *n_ = (int)n;
    *s = size;
    if (n > 0) prefetch_submit(seqs, n);                            // the chunk's device stages start now, beside the previous chunk's SAM phase
    return seqs;
}
