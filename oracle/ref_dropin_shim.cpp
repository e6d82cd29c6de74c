// TEST INFRASTRUCTURE -- the reference-side binding of the MI355X backend, exactly what INTEGRATION.md
// describes, written against the *reference's own headers* and linked into a copy of the reference aligner
// (oracle/Makefile.ref target bwa-meme_dropin) so that the end-to-end test can diff SAM files:
//
//   mem_kernel1_core_Learned()                      (reference src/bwamem.cpp:1230-1413)
//       per-read seeding loop  ->  one meme_seed_batch() call per 512-read batch; everything after seeding
//       (ks_introsort, mem_chain_Learned, mem_chain_flt, mem_flt_chained_seeds) is the reference's own code.
//   BandedPairWiseSW::getScores8 / getScores16 / scalarBandedSWAWrapper   (src/bandedSWA.cpp:242-260,1970-,2664-)
//       ->  meme_bsw_batch()
//
// The reference objects are built position-independent into libbwa_pic.so; these definitions live in the
// executable and therefore win symbol resolution (ELF interposition) -- no reference source is modified or
// copied.  Contains no reference code: the calls below go to functions the reference exports.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
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

meme_ctx* ctx_for(int tid) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (tid < 0 || tid >= 1024) { fprintf(stderr, "[meme-dropin] bad tid %d\n", tid); exit(1); }
    if (g_ctx[tid]) return g_ctx[tid];
    meme_ctx* c = meme_ctx_create(0);
    if (!c) { fprintf(stderr, "[meme-dropin] %s\n", meme_last_error()); exit(1); }
    if (!g_owner) {
        const char* prefix = getenv("MEME_INDEX_PREFIX");
        if (!prefix) { fprintf(stderr, "[meme-dropin] set MEME_INDEX_PREFIX to the index prefix\n"); exit(1); }
        if (meme_index_load_files(c, prefix)) { fprintf(stderr, "[meme-dropin] %s\n", meme_last_error()); exit(1); }
        g_owner = c;
    } else if (meme_index_share(c, g_owner)) { fprintf(stderr, "[meme-dropin] %s\n", meme_last_error()); exit(1); }
    g_ctx[tid] = c;
    return c;
}

}  // namespace

int mem_kernel1_core_Learned(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, bseq1_t* seq_, int nseq,
                             mem_chain_v* chain_ar, mem_seed_t* seedBuf, int64_t seedBufSize, uint8_t* sa_pos,
                             uint8_t* ref2sa, uint8_t* ref_string, mem_tlv* smems, u64v* hits, int tid) {
    (void)sa_pos; (void)ref2sa; (void)ref_string;
    if (getenv("MEME_DROPIN_TRACE")) signal(SIGSEGV, segv_handler);
    TRACE("seed batch tid=%d nseq=%d", tid, nseq);
    meme_ctx* ctx = ctx_for(tid);
    TRACE("ctx ok");
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
    std::vector<meme_mem_tl> sm((size_t)nseq * 64 + 1024);
    std::vector<uint64_t> ht((size_t)nseq * 1024 + 65536);
    std::vector<int64_t> smo((size_t)nseq + 1), hto((size_t)nseq + 1);
    int64_t ts = 0, th = 0;
    for (;;) {
        int rc = meme_seed_batch(ctx, flat.data(), off.data(), nseq, &so, sm.data(), (int64_t)sm.size(), smo.data(),
                                 ht.data(), (int64_t)ht.size(), hto.data(), &ts, &th);
        if (rc == MEME_E_CAPACITY) { sm.resize((size_t)ts + 1); ht.resize((size_t)th + 1); continue; }
        if (rc) { fprintf(stderr, "[meme-dropin] meme_seed_batch: %s\n", meme_last_error()); exit(1); }
        break;
    }
    TRACE("seeded: %lld smems %lld hits", (long long)ts, (long long)th);
    static_assert(sizeof(meme_mem_tl) == sizeof(mem_tl), "mem_tl layout");
    for (int l = 0; l < nseq; ++l) {
        const int64_t ns = smo[(size_t)l + 1] - smo[(size_t)l], nh = hto[(size_t)l + 1] - hto[(size_t)l];
        smems->n = 0;
        hits->n = 0;
        if ((int64_t)smems->m < ns) kv_resize(mem_tl, *smems, (size_t)ns);
        if ((int64_t)hits->m < nh) kv_resize(uint64_t, *hits, (size_t)nh);
        memcpy(smems->a, &sm[(size_t)smo[(size_t)l]], (size_t)ns * sizeof(mem_tl));
        memcpy(hits->a, &ht[(size_t)hto[(size_t)l]], (size_t)nh * sizeof(uint64_t));
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
void bsw_forward(const int8_t* mat, int o_del, int e_del, int o_ins, int e_ins, int zdrop, int end_bonus,
                 SeqPair* pairs, uint8_t* ref, uint8_t* qer, int n, int w) {
    TRACE("bsw n=%d w=%d", n, w);
    if (n <= 0) return;
    // tid is not passed down to this level; extension runs on the ctx of slot 1023-... use a private pool
    static thread_local meme_ctx* ctx = nullptr;
    if (!ctx) {
        ctx = meme_ctx_create(0);
        if (!ctx) { fprintf(stderr, "[meme-dropin] %s\n", meme_last_error()); exit(1); }
    }
    meme_bsw_opt o;
    o.o_del = o_del; o.e_del = e_del; o.o_ins = o_ins; o.e_ins = e_ins; o.zdrop = zdrop; o.end_bonus = end_bonus;
    o.a = mat[0]; o.b = -mat[1];
    int64_t rb = 0, qb = 0;
    for (int i = 0; i < n; ++i) {
        if ((int64_t)pairs[i].idr + pairs[i].len1 > rb) rb = (int64_t)pairs[i].idr + pairs[i].len1;
        if ((int64_t)pairs[i].idq + pairs[i].len2 > qb) qb = (int64_t)pairs[i].idq + pairs[i].len2;
    }
    static_assert(sizeof(meme_seqpair) == sizeof(SeqPair), "SeqPair layout");
    if (meme_bsw_batch(ctx, (meme_seqpair*)pairs, ref, rb, qer, qb, n, w, &o)) {
        fprintf(stderr, "[meme-dropin] meme_bsw_batch: %s\n", meme_last_error());
        exit(1);
    }
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
