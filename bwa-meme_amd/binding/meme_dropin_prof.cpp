// MEASUREMENT ONLY -- linked into oracle/_ref/bwa-meme_dropin_prof, never into bwa-meme_dropin.  Where the host's thread time goes in the SAM
// phase (worker_sam, src/bwamem.cpp:1827-1902) once seeding, chaining, extension, the CIGAR kernel and mate rescue are on the device: every
// function of that phase that the reference exports is interposed by a timer (TSC ticks per thread, flushed when a worker thread ends, one
// report when the process exits).  Times are INCLUSIVE: mem_sam_pe_batch_post contains everything below it, mem_reg2sam contains
// mem_reg2aln + mem_aln2sam, mem_reg2aln contains bwa_gen_cigar2 (which the binding answers from its table; its own counter is in
// meme_dropin_sam.cpp, compiled with the same switch).  The report (profiles/r05_sam_phase_split.md) decides what moves to the device next.
#define MEME_DROPIN_PROF 1
#include "meme_dropin_prof.h"

#include "meme_dropin.h"
#include "kswv.h"

using namespace dropin_prof;

namespace {

const char* const NAMES[P_N] = {"mem_sam_pe_batch_post (whole third step)", "mem_sam_pe_batch_pre (job posing, the binding's pre-pass)", "mem_matesw_batch_post",
                                "mem_matesw_batch_post_mate_sort", "mem_mark_primary_se", "mem_pair", "mem_gen_alt", "mem_reg2aln", "mem_reg2sam", "mem_sort_dedup_patch",
                                "mem_sort_dedup_patch_mate_sort", "mem_approx_mapq_se", "bwa_gen_cigar2 (the binding's hook)", "mem_aln2sam"};
std::atomic<uint64_t> g_ticks[P_N], g_calls[P_N];
uint64_t g_tsc0 = 0;
double g_wall0 = 0;
double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__attribute__((constructor)) void prof_start() { g_tsc0 = __rdtsc(); g_wall0 = wall(); }
__attribute__((destructor)) void prof_report() {
    const double hz = (double)(__rdtsc() - g_tsc0) / (wall() - g_wall0);
    { Local& l = local(); flush(l.t, l.n); for (int i = 0; i < P_N; ++i) l.n[i] = l.t[i] = 0; }
    fprintf(stderr, "[meme-dropin-prof] SAM phase, inclusive thread-seconds (TSC at %.2f GHz; a thread that is descheduled keeps counting):\n", hz * 1e-9);
    for (int i = 0; i < P_N; ++i)
        fprintf(stderr, "[meme-dropin-prof]   %-62s %9.3f s  %12llu calls\n", NAMES[i], (double)g_ticks[i].load() / hz, (unsigned long long)g_calls[i].load());
}

template <typename F> F next_of(const char* sym) {
    F f = (F)dlsym(RTLD_NEXT, sym);
    if (!f) { fprintf(stderr, "[meme-dropin-prof] %s not found in the reference library\n", sym); exit(1); }
    return f;
}

}  // namespace

void dropin_prof::flush(const uint64_t* t, const uint64_t* n) { for (int i = 0; i < P_N; ++i) if (n[i]) { g_ticks[i] += t[i]; g_calls[i] += n[i]; } }

int mem_sam_pe_batch_post(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, const mem_pestat_t pes[4], uint64_t id, bseq1_t s[2], mem_alnreg_v a[2], kswr_t** myaln,
                          mem_cache* mmc, int32_t& gcnt, int tid) {
    typedef int (*fn)(const mem_opt_t*, const bntseq_t*, const uint8_t*, const mem_pestat_t*, uint64_t, bseq1_t*, mem_alnreg_v*, kswr_t**, mem_cache*, int32_t&, int);
    static fn next = next_of<fn>("_Z21mem_sam_pe_batch_postPK9mem_opt_tPK8bntseq_tPKhPK12mem_pestat_tmP7bseq1_tP12mem_alnreg_vPP6kswr_tP9mem_cacheRii");
    PROF_SCOPE(P_POST);
    return next(opt, bns, pac, pes, id, s, a, myaln, mmc, gcnt, tid);
}
int mem_sam_pe_batch_pre(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, const mem_pestat_t pes[4], uint64_t id, bseq1_t s[2], mem_alnreg_v a[2], mem_cache* mmc,
                         int64_t& pcnt, int32_t& gcnt, int32_t& maxRefLen, int32_t& maxQerLen, int tid) {
    typedef int (*fn)(const mem_opt_t*, const bntseq_t*, const uint8_t*, const mem_pestat_t*, uint64_t, bseq1_t*, mem_alnreg_v*, mem_cache*, int64_t&, int32_t&, int32_t&, int32_t&, int);
    static fn next = next_of<fn>("_Z20mem_sam_pe_batch_prePK9mem_opt_tPK8bntseq_tPKhPK12mem_pestat_tmP7bseq1_tP12mem_alnreg_vP9mem_cacheRlRiSH_SH_i");
    PROF_SCOPE(P_PRE);
    return next(opt, bns, pac, pes, id, s, a, mmc, pcnt, gcnt, maxRefLen, maxQerLen, tid);
}
int mem_matesw_batch_post(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, const mem_pestat_t pes[4], const mem_alnreg_t* a, int l_ms, const uint8_t* ms, mem_alnreg_v* ma,
                          kswr_t** myaln, int32_t gcnt, int32_t* gar, mem_cache* mmc) {
    typedef int (*fn)(const mem_opt_t*, const bntseq_t*, const uint8_t*, const mem_pestat_t*, const mem_alnreg_t*, int, const uint8_t*, mem_alnreg_v*, kswr_t**, int32_t, int32_t*, mem_cache*);
    static fn next = next_of<fn>("_Z21mem_matesw_batch_postPK9mem_opt_tPK8bntseq_tPKhPK12mem_pestat_tPK12mem_alnreg_tiS6_P12mem_alnreg_vPP6kswr_tiPiP9mem_cache");
    PROF_SCOPE(P_MATESW_POST);
    return next(opt, bns, pac, pes, a, l_ms, ms, ma, myaln, gcnt, gar, mmc);
}
int mem_matesw_batch_post_mate_sort(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, const mem_pestat_t pes[4], const mem_alnreg_t* a, int l_ms, const uint8_t* ms,
                                    mem_alnreg_v* ma, kswr_t** myaln, int32_t gcnt, int32_t* gar, mem_cache* mmc) {
    typedef int (*fn)(const mem_opt_t*, const bntseq_t*, const uint8_t*, const mem_pestat_t*, const mem_alnreg_t*, int, const uint8_t*, mem_alnreg_v*, kswr_t**, int32_t, int32_t*, mem_cache*);
    static fn next = next_of<fn>("_Z31mem_matesw_batch_post_mate_sortPK9mem_opt_tPK8bntseq_tPKhPK12mem_pestat_tPK12mem_alnreg_tiS6_P12mem_alnreg_vPP6kswr_tiPiP9mem_cache");
    PROF_SCOPE(P_MATESW_POST_MS);
    return next(opt, bns, pac, pes, a, l_ms, ms, ma, myaln, gcnt, gar, mmc);
}
int mem_mark_primary_se(const mem_opt_t* opt, int n, mem_alnreg_t* a, int64_t id) {
    typedef int (*fn)(const mem_opt_t*, int, mem_alnreg_t*, int64_t);
    static fn next = next_of<fn>("_Z19mem_mark_primary_sePK9mem_opt_tiP12mem_alnreg_tl");
    PROF_SCOPE(P_MARK_PRIMARY);
    return next(opt, n, a, id);
}
int mem_pair(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, const mem_pestat_t pes[4], bseq1_t s[2], mem_alnreg_v a[2], int id, int* sub, int* n_sub, int z[2], int n_pri[2]) {
    typedef int (*fn)(const mem_opt_t*, const bntseq_t*, const uint8_t*, const mem_pestat_t*, bseq1_t*, mem_alnreg_v*, int, int*, int*, int*, int*);
    static fn next = next_of<fn>("_Z8mem_pairPK9mem_opt_tPK8bntseq_tPKhPK12mem_pestat_tP7bseq1_tP12mem_alnreg_viPiSE_SE_SE_");
    PROF_SCOPE(P_PAIR);
    return next(opt, bns, pac, pes, s, a, id, sub, n_sub, z, n_pri);
}
char** mem_gen_alt(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, const mem_alnreg_v* a, int l_query, const char* query) {
    typedef char** (*fn)(const mem_opt_t*, const bntseq_t*, const uint8_t*, const mem_alnreg_v*, int, const char*);
    static fn next = next_of<fn>("_Z11mem_gen_altPK9mem_opt_tPK8bntseq_tPKhPK12mem_alnreg_viPKc");
    PROF_SCOPE(P_GEN_ALT);
    return next(opt, bns, pac, a, l_query, query);
}
mem_aln_t mem_reg2aln(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, int l_query, const char* query, const mem_alnreg_t* ar) {
    typedef mem_aln_t (*fn)(const mem_opt_t*, const bntseq_t*, const uint8_t*, int, const char*, const mem_alnreg_t*);
    static fn next = next_of<fn>("_Z11mem_reg2alnPK9mem_opt_tPK8bntseq_tPKhiPKcPK12mem_alnreg_t");
    PROF_SCOPE(P_REG2ALN);
    return next(opt, bns, pac, l_query, query, ar);
}
void mem_reg2sam(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, bseq1_t* s, mem_alnreg_v* a, int extra_flag, const mem_aln_t* m) {
    typedef void (*fn)(const mem_opt_t*, const bntseq_t*, const uint8_t*, bseq1_t*, mem_alnreg_v*, int, const mem_aln_t*);
    static fn next = next_of<fn>("_Z11mem_reg2samPK9mem_opt_tPK8bntseq_tPKhP7bseq1_tP12mem_alnreg_viPK9mem_aln_t");
    PROF_SCOPE(P_REG2SAM);
    next(opt, bns, pac, s, a, extra_flag, m);
}
int mem_sort_dedup_patch(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, uint8_t* query, int n, mem_alnreg_t* a) {
    typedef int (*fn)(const mem_opt_t*, const bntseq_t*, const uint8_t*, uint8_t*, int, mem_alnreg_t*);
    static fn next = next_of<fn>("_Z20mem_sort_dedup_patchPK9mem_opt_tPK8bntseq_tPKhPhiP12mem_alnreg_t");
    PROF_SCOPE(P_SORT_DEDUP);
    return next(opt, bns, pac, query, n, a);
}
int mem_sort_dedup_patch_mate_sort(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, uint8_t* query, int n, mem_alnreg_t* a, bool* useMateSort) {
    typedef int (*fn)(const mem_opt_t*, const bntseq_t*, const uint8_t*, uint8_t*, int, mem_alnreg_t*, bool*);
    static fn next = next_of<fn>("_Z30mem_sort_dedup_patch_mate_sortPK9mem_opt_tPK8bntseq_tPKhPhiP12mem_alnreg_tPb");
    PROF_SCOPE(P_SORT_DEDUP_MS);
    return next(opt, bns, pac, query, n, a, useMateSort);
}
int mem_approx_mapq_se(const mem_opt_t* opt, const mem_alnreg_t* a) {
    typedef int (*fn)(const mem_opt_t*, const mem_alnreg_t*);
    static fn next = next_of<fn>("_Z18mem_approx_mapq_sePK9mem_opt_tPK12mem_alnreg_t");
    PROF_SCOPE(P_APPROX_MAPQ);
    return next(opt, a);
}
