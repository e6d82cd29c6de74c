// Reference-side binding of the MI355X backend (what INTEGRATION.md describes), written against the *reference's own
// headers* and linked into the reference aligner (oracle/Makefile.ref target bwa-meme_dropin).  The reference objects
// are built position-independent into libbwa_pic.so; the definitions below live in the executable and therefore win
// symbol resolution (ELF interposition) -- no reference source is modified or copied, the calls below go to functions
// the reference exports.  A maintainer integrating the backend would put the same code behind an #ifdef at the five
// places named here:
//
//   memoryAllocLearned()            src/fastmap.cpp:351-641   worker buffers as before, but the index files stream to HBM
//                                   (meme_index_load_files + meme_index_replicate per extra GPU) instead of being
//                                   expanded on the host (13-byte suffix-array entries + ISA, ~200 s / ~120 GB at GRCh38)
//   mem_process_seqs()              src/bwamem.cpp:1920-1972  ONE meme_seed_batch_host() per -K chunk (split over the
//                                   visible GPUs) followed by meme_chain_last_batch_host() (mem_chain_Learned +
//                                   mem_chain_flt on the device) before kt_for(worker_bwt); then the reference's own body
//   mem_kernel1_core_Learned()      src/bwamem.cpp:1230-1413  per 512-read batch: takes the chunk's chains; the reference's
//                                   ks_introsort / mem_chain_Learned / mem_chain_flt only for reads the device flagged;
//                                   mem_flt_chained_seeds as before
//   mem_chain2aln_across_reads_V2() src/bwamem.cpp:2573-3497  the extension jobs of the WHOLE chunk, stage by stage: one
//                                   meme_bsw_batch() per direction and band width (the first batch to arrive does it)
//   BandedPairWiseSW::getScores8 / getScores16 / scalarBandedSWAWrapper   src/bandedSWA.cpp:242-260,1970-,2664-
//                                   -> meme_bsw_batch(); concurrent calls of the kt_for workers combined into one backend
//                                   call per GPU (group commit) -- the path when the chunk-wide stage is switched off
#include <dlfcn.h>
#include <sched.h>
#include <time.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
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
// (reached through an accessor: the early-start thread below may run before this file's dynamic initialisers)
std::vector<Device>& device_slots() { static std::vector<Device>* v = new std::vector<Device>(); return *v; }
#define g_dev (device_slots())
std::mutex g_mu;
std::atomic<double> g_t_seed{0}, g_t_bsw_gather{0}, g_t_bsw_call{0}, g_t_bsw_kernel{0};
std::atomic<int64_t> g_n_bsw_calls{0}, g_n_bsw_pairs{0}, g_n_seed_reads{0};

void init_devices(const char* prefix, int64_t chunk_reads) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_dev.empty()) return;
    int n = meme_device_count();
    if (n <= 0) die("no HIP device");
    if (getenv("MEME_DROPIN_DEVICES")) { int want = atoi(getenv("MEME_DROPIN_DEVICES")); if (want >= 1 && want < n) n = want; }
    // MEME_DROPIN_VIRTUAL=k (tests): k device slots cycling over the real devices, so that the multi-GPU paths -- a chunk's
    // reads split over the slots, replicas of the index, one extension call per slot -- also run on a one-GPU box
    const int n_real = n;
    if (getenv("MEME_DROPIN_VIRTUAL") && atoi(getenv("MEME_DROPIN_VIRTUAL")) > n) n = atoi(getenv("MEME_DROPIN_VIRTUAL"));
    g_dev.resize((size_t)n);
    const double t0 = now_s();
    for (int d = 0; d < n; ++d) {
        if (!(g_dev[(size_t)d].seed = meme_ctx_create(d % n_real))) die("meme_ctx_create");
        if (!(g_dev[(size_t)d].bsw = meme_ctx_create(d % n_real))) die("meme_ctx_create");
        // small combined calls keep the lanes-per-pair kernels; combined calls of the whole thread team are big enough
        // for the lane-per-pair kernel much earlier than a lone caller's
        if (getenv("MEME_DROPIN_BSW_LANE_MIN")) meme_set_tuning(g_dev[(size_t)d].bsw, "bsw_lane_min_pairs", atoll(getenv("MEME_DROPIN_BSW_LANE_MIN")));
    }
    // while the index streams in: the seeding / chaining buffers of a chunk on every device slot (pinned memory is slow to allocate)
    std::thread reserve([n, chunk_reads] {
        int64_t per = chunk_reads / n + BATCH_SIZE;
        if (per > (1 << 20)) per = 1 << 20;                  // beyond a million reads per device the buffers grow on first use
        for (int d = 0; d < n; ++d) if (meme_seed_reserve(g_dev[(size_t)d].seed, per, per * READ_LEN)) die("meme_seed_reserve");
    });
    if (meme_index_load_files(g_dev[0].seed, prefix)) die("meme_index_load_files");
    reserve.join();
    const double t1 = now_s();
    std::vector<std::thread> th;
    for (int d = 1; d < n; ++d)                                 // device-to-device over xGMI, all replicas at once
        th.emplace_back([d] { if (meme_index_replicate(g_dev[(size_t)d].seed, g_dev[0].seed)) die("meme_index_replicate"); });
    for (auto& t : th) t.join();
    fprintf(stderr, "[meme-dropin] index staged in HBM in %.2f s, replicated to %d more GPU(s) in %.2f s\n", t1 - t0, n - 1,
            now_s() - t1);
}

// ---- early start -------------------------------------------------------------------------------------------------------------
// When the index prefix comes through the environment (MEME_INDEX_PREFIX) and the process is `... mem ... -7 ...`, the index starts
// streaming to HBM when the binding is loaded -- while the aligner still parses its arguments and reads its own copy of the reference
// (bns, pac, the 6 GB .0123 text: 1.5 s at GRCh38 size) -- instead of when memoryAllocLearned() is reached.  MEME_DROPIN_EARLY=0: off.
std::thread* g_early = nullptr;
const char* g_early_prefix = nullptr;
__attribute__((constructor)) void meme_dropin_early_start() {
    const char* p = getenv("MEME_INDEX_PREFIX");
    if (!p || !*p || (getenv("MEME_DROPIN_EARLY") && atoi(getenv("MEME_DROPIN_EARLY")) == 0)) return;
    FILE* f = fopen("/proc/self/cmdline", "rb");
    if (!f) return;
    char buf[8192];
    const size_t len = fread(buf, 1, sizeof(buf) - 1, f);
    fclose(f);
    buf[len] = 0;
    bool is_mem = false, learned = false;
    int k = 0;
    for (size_t i = 0; i < len; i += strlen(buf + i) + 1, ++k) {
        if (k == 1 && !strcmp(buf + i, "mem")) is_mem = true;
        if (k > 1 && !strcmp(buf + i, "-7")) learned = true;
    }
    if (!is_mem || !learned) return;
    g_early_prefix = p;
    g_early = new std::thread([p] { init_devices(p, 1 << 20); });
}

// ---- memoryAllocLearned (src/fastmap.cpp:351-641) ----------------------------------------------------------------
// Worker buffers exactly as the reference sizes them (they are indexed by the kt_for thread id all over
// mem_chain2aln_across_reads_V2 and freed by process(), src/fastmap.cpp:1098-1110); the host-side index expansion is gone.
void ext_prepare(int64_t chunk_reads, int threads);       // defined with the extension stage below
int ext_mode();

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
    if (strcmp(prefix, idx_prefix) != 0)
        fprintf(stderr, "[meme-dropin] note: the HBM index comes from MEME_INDEX_PREFIX=%s, bns / pac / .0123 from %s (checked below: same number of suffixes)\n", prefix, idx_prefix);
    std::thread prep([nreads, nthreads] { if (ext_mode() == 1) ext_prepare((int64_t)nreads, (int)nthreads); });   // host stage: pinned staging + helper threads, while the index loads
    if (g_early_prefix && strcmp(g_early_prefix, prefix) != 0) { fprintf(stderr, "[meme-dropin] MEME_INDEX_PREFIX changed after start-up\n"); exit(1); }
    init_devices(prefix, (int64_t)nreads);                                       // (returns at once when the early load below has done it)
    if (g_early) { g_early->join(); delete g_early; g_early = nullptr; }
    prep.join();
    {   // the suffix array in HBM must describe the genome whose bns / pac the aligner loaded
        meme_index_arrays ia;
        if (meme_index_describe(g_dev[0].seed, &ia)) die("meme_index_describe");
        if (ia.sa_num != 2 * l_pac) {
            fprintf(stderr, "[meme-dropin] the index staged from %s has %lld suffixes, the reference sequence of %s needs %lld: wrong MEME_INDEX_PREFIX?\n", prefix,
                    (long long)ia.sa_num, idx_prefix, (long long)(2 * l_pac));
            exit(1);
        }
    }
    fprintf(stderr, "[meme-dropin] worker buffers + fwd/rc text %.2f s, HBM index %.2f s (no host-side index expansion)\n",
            t1 - t0, now_s() - t1);
}

// ---- chunk-level seeding ----------------------------------------------------------------------------------------------
namespace {

struct ChunkPart {                     // the slice of a chunk one GPU seeded
    int64_t first = 0, count = 0;
    meme_seed_host_result res;
    meme_chain_host_result chains;                       // valid when g_chain_on_device (host extension stage)
    meme_ext_host_result ext;                            // valid in device-extension mode: alignment records of the part's reads
    bool has_ext = false;
    uint8_t* flat = nullptr; int64_t flat_cap = 0;       // pinned staging (grow-only)
    int64_t* off = nullptr; int64_t off_cap = 0;
};
struct Chunk {
    const bseq1_t* seqs = nullptr;
    int64_t n = 0;
    std::vector<ChunkPart> part;
} g_chunk;

const bntseq_t* g_bns = nullptr;               // of the run (set by mem_process_seqs)
std::vector<meme_contig> g_contigs;
// MEME_DROPIN_EXT: "device" (default) = chaining AND seed extension on the GPU, the host receives alignment records;
// "host" = chains from the device, extension jobs built / folded / purged by the binding's thread team (round 2's arrangement, and
// what runs when -W min_chain_weight makes mem_flt_chained_seeds more than a no-op); "0" = the reference's own per-batch function.
int ext_mode() {
    static const int v = [] {
        const char* e = getenv("MEME_DROPIN_EXT");
        if (!e || !strcmp(e, "device") || !strcmp(e, "2")) return 2;
        if (!strcmp(e, "0")) return 0;
        return 1;
    }();
    return v;
}
bool g_ext_on_device = false;           // decided per run in mem_process_seqs (needs opt)
std::atomic<double> g_t_ext_dev{0}, g_t_ext_chain_ms{0}, g_t_ext_ms{0}, g_t_ext_bsw_ms{0};
std::atomic<int64_t> g_n_ext_pairs{0}, g_n_ext_retried{0}, g_n_ext_regs{0}, g_n_ext_tier2{0};
bool chain_on_device() { static const bool v = !(getenv("MEME_DROPIN_CHAIN") && atoi(getenv("MEME_DROPIN_CHAIN")) == 0); return v; }
bool chain_check() { static const bool v = getenv("MEME_DROPIN_CHAIN_CHECK") != nullptr; return v; }
// MEME_DROPIN_CHAIN_DUMP=<file> (fixture generation, tests/golden/make_chain_golden.py): every read's seeds and the chains the
// REFERENCE's host functions make of them, as text
FILE* chain_dump() { static FILE* f = getenv("MEME_DROPIN_CHAIN_DUMP") ? fopen(getenv("MEME_DROPIN_CHAIN_DUMP"), "w") : nullptr; return f; }
std::mutex g_dump_mu;
std::atomic<int64_t> g_n_chain_fallback{0}, g_n_chain_reads{0};

int cig_threads();
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
    // (a few dozen helper threads: an OpenMP team of all 256 host threads takes longer to start than the loop runs, and keeps spinning
    // into the worker phases that follow)
#pragma omp parallel for schedule(static) num_threads(cig_threads())
    for (int64_t i = 0; i < P.count; ++i) {
        // base codes in place, as the reference leaves them for the later stages (src/bwamem.cpp:1277-1279)
        bseq1_t& s = seqs[P.first + i];
        uint8_t* dst = P.flat + P.off[i];
        for (int k = 0; k < s.l_seq; ++k) { const char c = s.seq[k]; s.seq[k] = c < 4 ? c : (char)nst_nt4_table[(int)c]; dst[k] = (uint8_t)s.seq[k]; }
    }
    const meme_seed_opt so = seed_opt_of(opt);
    memset(&P.chains, 0, sizeof(P.chains));
    P.has_ext = false;
    if (g_ext_on_device) {                               // seeds stay in HBM (nothing on the host reads them)
        memset(&P.res, 0, sizeof(P.res));
        if (meme_seed_batch_resident(g_dev[(size_t)d].seed, P.flat, P.off, P.count, &so, nullptr, nullptr)) die("meme_seed_batch_resident");
    } else if (meme_seed_batch_host(g_dev[(size_t)d].seed, P.flat, P.off, P.count, &so, &P.res)) die("meme_seed_batch_host");
    if (g_ext_on_device && P.count > 0) {                // chains + extension where the seeds lie: only alignment records come back
        meme_chain_opt co;
        co.w = opt->w; co.max_chain_gap = opt->max_chain_gap; co.max_occ = opt->max_occ; co.min_seed_len = opt->min_seed_len;
        co.min_chain_weight = opt->min_chain_weight; co.max_chain_extend = opt->max_chain_extend;
        co.mask_level = opt->mask_level; co.drop_ratio = opt->drop_ratio; co.l_pac = g_bns->l_pac;
        meme_ext_opt eo;
        eo.a = opt->a; eo.b = opt->b; eo.o_del = opt->o_del; eo.e_del = opt->e_del; eo.o_ins = opt->o_ins; eo.e_ins = opt->e_ins;
        eo.pen_clip5 = opt->pen_clip5; eo.pen_clip3 = opt->pen_clip3; eo.w = opt->w; eo.zdrop = opt->zdrop;
        const double t0 = now_s();
        if (meme_extend_last_batch_host(g_dev[(size_t)d].seed, g_contigs.data(), (int32_t)g_contigs.size(), &co, &eo, &P.ext)) die("meme_extend_last_batch_host");
        g_t_ext_dev = g_t_ext_dev + (now_s() - t0);
        g_t_ext_chain_ms = g_t_ext_chain_ms + P.ext.chain_ms; g_t_ext_ms = g_t_ext_ms + P.ext.ext_ms; g_t_ext_bsw_ms = g_t_ext_bsw_ms + P.ext.bsw_ms;
        g_n_ext_pairs += P.ext.n_pairs; g_n_ext_retried += P.ext.n_retried; g_n_ext_regs += P.ext.total_regs; g_n_ext_tier2 += P.ext.n_tier2;
        P.has_ext = true;
        return;
    }
    if (chain_on_device() && P.count > 0) {              // mem_chain_Learned + mem_chain_flt while the seeds are still in HBM
        meme_chain_opt co;
        co.w = opt->w; co.max_chain_gap = opt->max_chain_gap; co.max_occ = opt->max_occ; co.min_seed_len = opt->min_seed_len;
        co.min_chain_weight = opt->min_chain_weight; co.max_chain_extend = opt->max_chain_extend;
        co.mask_level = opt->mask_level; co.drop_ratio = opt->drop_ratio; co.l_pac = g_bns->l_pac;
        if (meme_chain_last_batch_host(g_dev[(size_t)d].seed, g_contigs.data(), (int32_t)g_contigs.size(), &co, &P.chains)) die("meme_chain_last_batch_host");
    }
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
worker_t* g_worker = nullptr;             // of the chunk being processed (alignment records, ref_string: the CIGAR stage reads them)
const mem_opt_t* g_opt = nullptr;
std::atomic<int>& ktfor_calls();
mem_chain_v* g_chunk_chain_ar = nullptr;   // w.chain_ar of the chunk being processed: every batch's chain_ar is a slice of it
uint64_t g_chunk_gen = 0;               // counts the chunks seeded

typedef void (*process_fn)(mem_opt_t*, int64_t, int, bseq1_t*, const mem_pestat_t*, worker_t&);
void ext_report();
}
void meme_dropin_report_matesw();
void meme_dropin_report_cigar();
void meme_dropin_report_mate();
namespace {

}  // namespace

void mem_process_seqs(mem_opt_t* opt, int64_t n_processed, int n, bseq1_t* seqs, const mem_pestat_t* pes0, worker_t& w) {
    static process_fn next = nullptr;
    if (!next) {
        next = (process_fn)dlsym(RTLD_NEXT, "_Z16mem_process_seqsP9mem_opt_tliP7bseq1_tPK12mem_pestat_tR8worker_t");
        if (!next) { fprintf(stderr, "[meme-dropin] the reference's mem_process_seqs was not found: %s\n", dlerror()); exit(1); }
    }
    g_team = opt->n_threads > 0 ? opt->n_threads : 1;
    if (w.useLearned && !g_bns) {
        g_bns = w.fmi->idx->bns;
        for (int i = 0; i < g_bns->n_seqs; ++i) g_contigs.push_back({g_bns->anns[i].offset, g_bns->anns[i].len, g_bns->anns[i].is_alt});
    }
    if (w.useLearned) {
        // mem_flt_chained_seeds (src/bwamem.cpp:565-598) sits between chaining and extension; it is a no-op unless
        // 1.1 * min_chain_weight <= 0.05 * read length (never with the default min_chain_weight 0 and reads of at most 500 bases)
        g_ext_on_device = ext_mode() == 2 && opt->min_chain_weight == 0;
        g_chunk_chain_ar = w.chain_ar;
        g_worker = &w;
        g_opt = opt;
        ktfor_calls() = 0;
        seed_chunk(opt, seqs, n);
        ++g_chunk_gen;
    }
    next(opt, n_processed, n, seqs, pes0, w);
    g_chunk.seqs = nullptr;
    if (verbose())
        fprintf(stderr, "[meme-dropin] totals: seeding %.3f s for %lld reads; bsw %lld calls, %lld pairs "
                "(copy-in thread-seconds %.3f, backend calls %.3f s of which kernels %.3f s)\n",
                (double)g_t_seed, (long long)g_n_seed_reads, (long long)g_n_bsw_calls, (long long)g_n_bsw_pairs,
                (double)g_t_bsw_gather, (double)g_t_bsw_call, (double)g_t_bsw_kernel);
    if (verbose() && g_ext_on_device)
        fprintf(stderr, "[meme-dropin] chaining + extension on the device: %.3f s in the backend calls so far (HIP events: chaining %.3f s, extension stage %.3f s of "
                "which banded SW %.3f s); %lld alignment records, %lld extension jobs (%lld of them again with the doubled band), %lld reads chained by the "
                "wavefront-per-read tier, 0 reads chained on the host\n", (double)g_t_ext_dev, (double)g_t_ext_chain_ms * 1e-3, (double)g_t_ext_ms * 1e-3,
                (double)g_t_ext_bsw_ms * 1e-3, (long long)g_n_ext_regs, (long long)g_n_ext_pairs, (long long)g_n_ext_retried, (long long)g_n_ext_tier2);
    if (verbose() && !g_ext_on_device && chain_on_device())
        fprintf(stderr, "[meme-dropin] chaining on the device: %lld of %lld reads so far were chained on the host instead\n",
                (long long)g_n_chain_fallback, (long long)g_n_chain_reads);
    if (verbose() && !g_ext_on_device) ext_report();
    if (verbose() && getenv("MEME_DROPIN_PROFILE_SAM")) meme_dropin_report_matesw();
    if (verbose()) meme_dropin_report_cigar();
    if (verbose()) meme_dropin_report_mate();
}

namespace {

// one read's chains on the host, with the reference's own functions (src/bwamem.cpp:1396-1407)
void host_chain_read(const mem_opt_t* opt, const bntseq_t* bns, const bseq1_t& rd, const ChunkPart& P, int64_t r, int seqid, mem_tlv* smems,
                     u64v* hits, mem_chain_v* chain, mem_seed_t* seedBuf, int64_t seedBufSize, int64_t& seedBufCount, int tid) {
    const int64_t s0 = P.res.smem_off[r], ns = P.res.smem_off[r + 1] - s0;
    const int64_t h0 = P.res.hit_off[r], nh = P.res.hit_off[r + 1] - h0;
    smems->n = 0;
    hits->n = 0;
    if ((int64_t)smems->m < ns) kv_resize(mem_tl, *smems, (size_t)ns);
    if ((int64_t)hits->m < nh) kv_resize(uint64_t, *hits, (size_t)nh);
    if (ns) memcpy(smems->a, P.res.smems + s0, (size_t)ns * sizeof(mem_tl));
    if (nh) memcpy(hits->a, P.res.hits + h0, (size_t)nh * sizeof(uint64_t));
    smems->n = (size_t)ns;
    hits->n = (size_t)nh;
    ks_introsort(meme_dropin_smem, smems->n, smems->a);            // src/bwamem.cpp:1397
    kv_init(*chain);
    mem_chain_Learned(opt, bns, rd.l_seq, smems, chain, seqid, hits, seedBuf, seedBufSize, seedBufCount, tid);
    chain->n = mem_chain_flt(opt, chain->n, chain->a, tid);
}

// the device's chains of one read as the reference's structures: chain array sized like kv_resize(kb_size(tree)) leaves it,
// single-seed chains in the batch's seed slab, longer ones in arrays of their own (the reference frees those, :1667-1676)
void device_chain_read(const mem_opt_t* opt, const bseq1_t& rd, const ChunkPart& P, int64_t r, int seqid, mem_chain_v* chain, mem_seed_t* seedBuf,
                       int64_t seedBufSize, int64_t& seedBufCount) {
    kv_init(*chain);
    if (rd.l_seq < opt->min_seed_len) return;                      // (:1138)
    const meme_chain_host_result& C = P.chains;
    chain->m = (size_t)C.tree_size[r];
    chain->a = (mem_chain_t*)malloc(sizeof(mem_chain_t) * (chain->m ? chain->m : 1));
    const int64_t c0 = C.chain_off[r], nc = C.chain_off[r + 1] - c0;
    const meme_chain_seed* sd = C.seeds + C.seed_off[r];
    for (int64_t k = 0; k < nc; ++k) {
        const meme_chain& m = C.chains[c0 + k];
        mem_chain_t c;
        memset(&c, 0, sizeof(c));
        c.seqid = seqid; c.n = m.n_seeds; c.first = m.first; c.rid = m.rid; c.w = (uint32_t)m.w; c.kept = (uint32_t)m.kept;
        c.is_alt = (uint32_t)m.is_alt; c.frac_rep = C.frac_rep[r]; c.pos = m.pos;
        c.m = SEEDS_PER_CHAIN;
        while (c.m < c.n) c.m <<= 1;                               // how test_and_merge grows a chain (:474-489)
        if (c.m == SEEDS_PER_CHAIN && seedBufCount + c.m <= seedBufSize) { c.seeds = seedBuf + seedBufCount; seedBufCount += c.m; memset((void*)c.seeds, 0, c.m * sizeof(mem_seed_t)); }
        else { if (c.m == SEEDS_PER_CHAIN) c.m += 1; c.seeds = (mem_seed_t*)calloc((size_t)c.m, sizeof(mem_seed_t)); }
        for (int j = 0; j < c.n; ++j) {
            const meme_chain_seed& s = sd[m.seed_beg + j];
            c.seeds[j].rbeg = s.rbeg; c.seeds[j].qbeg = s.qbeg; c.seeds[j].len = s.len; c.seeds[j].score = s.len;
        }
        chain->a[chain->n++] = c;
    }
}

void free_chains(mem_chain_v* chain) {
    for (size_t i = 0; i < chain->n; ++i) if (chain->a[i].m > SEEDS_PER_CHAIN) free(chain->a[i].seeds);
    free(chain->a);
}

void dump_chains(int64_t g, const bseq1_t& rd, const ChunkPart& P, int64_t r, const mem_chain_v* host) {
    const int64_t s0 = P.res.smem_off[r], ns = P.res.smem_off[r + 1] - s0;
    const int64_t h0 = P.res.hit_off[r], nh = P.res.hit_off[r + 1] - h0;
    std::lock_guard<std::mutex> lk(g_dump_mu);
    FILE* f = chain_dump();
    uint32_t fr = 0;
    if (host->n) memcpy(&fr, &host->a[0].frac_rep, 4);
    fprintf(f, "R %lld %d %lld %lld %zu %zu %u\n", (long long)g, rd.l_seq, (long long)ns, (long long)nh, host->m, host->n, fr);
    for (int64_t i = 0; i < ns; ++i) { const meme_mem_tl& m = P.res.smems[s0 + i]; fprintf(f, "S %d %d %d %d\n", m.start, m.end, m.hitbeg, m.hitcount); }
    fprintf(f, "H");
    for (int64_t i = 0; i < nh; ++i) fprintf(f, " %llu", (unsigned long long)P.res.hits[h0 + i]);
    fprintf(f, "\n");
    for (size_t i = 0; i < host->n; ++i) {
        const mem_chain_t& c = host->a[i];
        fprintf(f, "C %lld %d %d %d %d %d %d :", (long long)c.pos, c.rid, c.n, (int)c.w, (int)c.kept, c.first, (int)c.is_alt);
        for (int j = 0; j < c.n; ++j) fprintf(f, " %lld %d %d", (long long)c.seeds[j].rbeg, c.seeds[j].qbeg, c.seeds[j].len);
        fprintf(f, "\n");
    }
}

// MEME_DROPIN_CHAIN_CHECK: every read chained both ways, any difference is fatal
void compare_chains(const bseq1_t& rd, const mem_chain_v* dev, const mem_chain_v* host) {
    bool same = dev->n == host->n;
    for (size_t i = 0; same && i < dev->n; ++i) {
        const mem_chain_t &a = dev->a[i], &b = host->a[i];
        same = a.n == b.n && a.first == b.first && a.rid == b.rid && a.w == b.w && a.kept == b.kept && a.is_alt == b.is_alt && a.pos == b.pos &&
               a.seqid == b.seqid && !memcmp(&a.frac_rep, &b.frac_rep, sizeof(float));
        for (int j = 0; same && j < a.n; ++j)
            same = a.seeds[j].rbeg == b.seeds[j].rbeg && a.seeds[j].qbeg == b.seeds[j].qbeg && a.seeds[j].len == b.seeds[j].len && a.seeds[j].score == b.seeds[j].score;
    }
    if (same) return;
    fprintf(stderr, "[meme-dropin] chains of read %s differ between the device and the host (%zu vs %zu chains)\n", rd.name, dev->n, host->n);
    for (int side = 0; side < 2; ++side) {
        const mem_chain_v* v = side ? host : dev;
        for (size_t i = 0; i < v->n; ++i) {
            const mem_chain_t& c = v->a[i];
            fprintf(stderr, "  %s chain %zu: pos %lld rid %d n %d w %d kept %d first %d frac_rep %.6f seeds", side ? "host  " : "device", i, (long long)c.pos, c.rid, c.n, (int)c.w,
                    (int)c.kept, c.first, c.frac_rep);
            for (int j = 0; j < c.n; ++j) fprintf(stderr, " (%lld,%d,%d)", (long long)c.seeds[j].rbeg, c.seeds[j].qbeg, c.seeds[j].len);
            fprintf(stderr, "\n");
        }
    }
    exit(1);
}

}  // namespace

int mem_kernel1_core_Learned(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, bseq1_t* seq_, int nseq,
                             mem_chain_v* chain_ar, mem_seed_t* seedBuf, int64_t seedBufSize, uint8_t* sa_pos,
                             uint8_t* ref2sa, uint8_t* ref_string, mem_tlv* smems, u64v* hits, int tid) {
    (void)sa_pos; (void)ref2sa; (void)ref_string;
    static_assert(sizeof(meme_mem_tl) == sizeof(mem_tl), "mem_tl layout");
    const int64_t g0 = seq_ - g_chunk.seqs;                   // this batch's position in the chunk seeded above
    if (!g_chunk.seqs || g0 < 0 || g0 + nseq > g_chunk.n) { fprintf(stderr, "[meme-dropin] batch outside the seeded chunk\n"); exit(1); }
    if (g_ext_on_device) {                                       // the chains stay in HBM: worker_aln takes the alignment records
        for (int l = 0; l < nseq; ++l) kv_init(chain_ar[l]);
        return 1;
    }
    int64_t seedBufCount = 0, n_fb = 0;
    static thread_local mem_seed_t* check_buf = nullptr;
    for (int l = 0; l < nseq; ++l) {
        const int64_t g = g0 + l;
        const ChunkPart* P = nullptr;
        for (const ChunkPart& c : g_chunk.part) if (g >= c.first && g < c.first + c.count) { P = &c; break; }
        const int64_t r = g - P->first;
        mem_chain_v* chn = &chain_ar[l];
        if (P->chains.nreads == P->count && !P->chains.fallback[r]) {
            device_chain_read(opt, seq_[l], *P, r, l, chn, seedBuf, seedBufSize, seedBufCount);
            if (chain_check() || chain_dump()) {
                const int64_t check_slots = 4096;
                if (!check_buf) check_buf = (mem_seed_t*)calloc((size_t)check_slots + 8, sizeof(mem_seed_t));
                mem_chain_v ref;
                int64_t cnt = 0;
                host_chain_read(opt, bns, seq_[l], *P, r, l, smems, hits, &ref, check_buf, check_slots, cnt, tid);
                if (chain_check()) compare_chains(seq_[l], chn, &ref);
                if (chain_dump()) dump_chains(g, seq_[l], *P, r, &ref);
                free_chains(&ref);
            }
        } else {
            host_chain_read(opt, bns, seq_[l], *P, r, l, smems, hits, chn, seedBuf, seedBufSize, seedBufCount, tid);
            ++n_fb;
            if (chain_dump()) dump_chains(g, seq_[l], *P, r, chn);
        }
        mem_flt_chained_seeds(opt, bns, pac, seq_, chn->n, chn->a);
    }
    g_n_chain_fallback += n_fb;
    g_n_chain_reads += nseq;
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

// ---- chunk-wide seed extension: mem_chain2aln_across_reads_V2 (src/bwamem.cpp:2573-3497) ------------------------------
// The reference's function takes one 512-read batch: it creates a left and a right extension job per chained seed, runs them
// class by class (8-bit / 16-bit / scalar lanes, two band widths each -- up to twelve BandedPairWiseSW calls of a few hundred
// pairs) and finally purges the alignments of seeds that an earlier alignment of the read already covers.  The chains of the
// whole -K chunk exist when the first batch gets here (kt_for(worker_bwt) has finished, src/bwamem.cpp:1941-1945), so the
// binding computes the same function for ALL reads of the chunk, stage by stage: jobs of a slab of reads built by a thread
// team straight into pinned staging, ONE backend call per direction and band width, results folded back by the team.  A
// batch's call then only takes its reads' alignment arrays.  MEME_DROPIN_EXT=0 keeps the reference's function (its
// BandedPairWiseSW calls then go through the combiner above).
namespace {

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
}  // namespace

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
#include <omp.h>
#include "kswv.h"
namespace {
std::atomic<double> g_t_matesw{0};
std::atomic<int64_t> g_n_matesw{0};
bool matesw_on_device() { static const bool v = getenv("MEME_DROPIN_MATESW") && atoi(getenv("MEME_DROPIN_MATESW")) != 0; return v; }
// One job per lane needs tens of thousands of jobs to fill the GPU (a chunk of 666 k reads poses ~31 k: 8 ms of kernel for what the host's
// 64 threads do in about as long); below this many jobs per chunk the reference's own batch runs (the bigger chunks bwa-meme reads by
// default -- 10 M bases x threads -- pose ~50 jobs per 1 000 reads: 200 k per chunk at -t 64).  MEME_DROPIN_MATESW_MIN overrides.
int64_t matesw_min_jobs() { static const int64_t v = getenv("MEME_DROPIN_MATESW_MIN") ? atoll(getenv("MEME_DROPIN_MATESW_MIN")) : 65536; return v; }
double g_mate_jobs_per_read = -1;                // of the last chunk whose jobs were posed
struct MateTable {
    std::vector<int64_t> off;                    // first record of every worker batch (+ the total)
    std::vector<kswr_t> aln;                     // records, batch after batch, in the order the jobs were posed (= regid)
    std::vector<std::vector<int32_t>> gar;       // per batch: job index (or -1) of every (alignment, orientation) mem_matesw_batch_pre looked at
    uint64_t gen = 0;                            // chunk the table belongs to
    double t_prepass = 0, t_kernel_ms = 0;
    int64_t n_jobs = 0;
    mem_cache* cache = nullptr;                  // the pre-pass's own buffers, one slot per helper thread
    int slots = 0;
} g_mate;
std::atomic<int64_t> g_mate_hits{0}, g_mate_miss{0};
// worker_sam's paired-end branch after its first two steps (src/bwamem.cpp:1879-1900), the results of those coming from the table
void sam_worker_dev(void* data, long seqid, long batch_size, int tid) {
    worker_t* w = (worker_t*)data;
    const MateTable& T = g_mate;
    const size_t b = (size_t)(seqid / BATCH_SIZE);
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

bool matesw_prepass() {                          // false: too few jobs for the device, worker_sam runs as it is
    const double t0 = now_s();
    worker_t* w = g_worker;
    const mem_opt_t* opt = g_opt;
    const int64_t n = g_chunk.n;
    const int64_t nb = (n + BATCH_SIZE - 1) / BATCH_SIZE;
    const int nt = cig_threads();
    if (!g_mate.cache) mate_cache_init(nt);
    struct BatchJobs { std::vector<meme_kswv_job> jobs; std::vector<uint8_t> ref, qer; };
    std::vector<BatchJobs> B((size_t)nb);
    g_mate.gar.assign((size_t)nb, std::vector<int32_t>());
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_mate.slots)
    for (int64_t b = 0; b < nb; ++b) {
        const int t = omp_get_thread_num();
        const int64_t st = b * BATCH_SIZE, ed = (b + 1) * BATCH_SIZE < n ? (b + 1) * BATCH_SIZE : n;
        int64_t pcnt = 0;
        int32_t gcnt = 0, maxRef = 0, maxQer = 0;
        int64_t pos = st >> 1;
        for (int64_t i = st; i + 1 < ed; i += 2)                  // worker_sam's loop (src/bwamem.cpp:1855-1866)
            mem_sam_pe_batch_pre(opt, w->fmi->idx->bns, w->fmi->idx->pac, w->pes, (uint64_t)((w->n_processed >> 1) + pos++), &w->seqs[i], &w->regs[i], g_mate.cache,
                                 pcnt, gcnt, maxRef, maxQer, t);
        BatchJobs& J = B[(size_t)b];
        g_mate.gar[(size_t)b].assign((const int32_t*)g_mate.cache->seqPairArrayAux[t], (const int32_t*)g_mate.cache->seqPairArrayAux[t] + gcnt);
        if (pcnt == 0) continue;
        const SeqPair* sp = g_mate.cache->seqPairArrayLeft128[t];
        const int64_t rbytes = (int64_t)sp[pcnt - 1].idr + sp[pcnt - 1].len1, qbytes = (int64_t)sp[pcnt - 1].idq + sp[pcnt - 1].len2;
        J.ref.assign(g_mate.cache->seqBufLeftRef[t * CACHE_LINE], g_mate.cache->seqBufLeftRef[t * CACHE_LINE] + rbytes);
        J.qer.assign(g_mate.cache->seqBufLeftQer[t * CACHE_LINE], g_mate.cache->seqBufLeftQer[t * CACHE_LINE] + qbytes);
        J.jobs.resize((size_t)pcnt);
        for (int64_t k = 0; k < pcnt; ++k) { meme_kswv_job& j = J.jobs[(size_t)k]; j.idr = sp[k].idr; j.idq = sp[k].idq; j.len1 = sp[k].len1; j.len2 = sp[k].len2; j.xtra = sp[k].h0; j.pad = 0; }
    }
    MateTable& T = g_mate;
    T.off.assign((size_t)nb + 1, 0);
    for (int64_t b = 0; b < nb; ++b) T.off[(size_t)b + 1] = T.off[(size_t)b] + (int64_t)B[(size_t)b].jobs.size();
    const int64_t total = T.off[(size_t)nb];
    g_mate_jobs_per_read = n > 0 ? (double)total / (double)n : 0;
    if (total < matesw_min_jobs()) { T.t_prepass += now_s() - t0; T.gen = 0; return false; }
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
        kms[(size_t)d] = R.kernel_ms;
    };
    std::vector<std::thread> th;
    for (int d = 1; d < nd; ++d) th.emplace_back(run_part, d);
    run_part(0);
    for (auto& x : th) x.join();
    double km = 0;
    for (double v : kms) km = km > v ? km : v;
    T.t_kernel_ms += km; T.n_jobs += total; T.t_prepass += now_s() - t0;
    T.gen = g_chunk_gen;
    return true;
}
}  // namespace

// (with the stage off: the reference's batch, timed)
typedef int (*sam_pe_batch_fn)(const mem_opt_t*, mem_cache*, int64_t&, int64_t&, kswr_t*, int32_t, int32_t, int);
int mem_sam_pe_batch(const mem_opt_t* opt, mem_cache* mmc, int64_t& pcnt, int64_t& pcnt8, kswr_t* aln, int32_t maxRefLen, int32_t maxQerLen, int tid) {
    static sam_pe_batch_fn next = (sam_pe_batch_fn)dlsym(RTLD_NEXT, "_Z16mem_sam_pe_batchPK9mem_opt_tP9mem_cacheRlS4_P6kswr_tiii");
    if (!next) { fprintf(stderr, "[meme-dropin] the reference's mem_sam_pe_batch was not found\n"); exit(1); }
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
            "%lld, run by the reference's kernels %lld\n", (long long)g_mate.n_jobs, g_mate.t_kernel_ms * 1e-3, g_mate.t_prepass, (long long)g_mate_hits.load(),
            (long long)g_mate_miss.load());
}
namespace {
std::atomic<double> g_t_cigar{0}, g_t_sam{0}; std::atomic<int64_t> g_n_cigar{0}, g_n_sam{0};
// per-record timing with shared counters costs a 256-thread run a third of its compute time: only on request
bool profile_sam() { static const bool v = getenv("MEME_DROPIN_PROFILE_SAM") != nullptr; return v; }
}
// (measurement only, MEME_DROPIN_PROFILE_SAM=1) the two other candidates of the SAM phase: CIGAR generation and SAM formatting
typedef uint32_t* (*gen_cigar2_fn)(const int8_t*, int, int, int, int, int, int64_t, const uint8_t*, int, uint8_t*, int64_t, int64_t, int*, int*, int*);
extern "C" uint32_t* bwa_gen_cigar2(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, int64_t l_pac, const uint8_t* pac, int l_query,
                                    uint8_t* query, int64_t rb, int64_t re, int* score, int* n_cigar, int* NM) {
    static gen_cigar2_fn next = (gen_cigar2_fn)dlsym(RTLD_NEXT, "bwa_gen_cigar2");
    if (!profile_sam()) return next(mat, o_del, e_del, o_ins, e_ins, w_, l_pac, pac, l_query, query, rb, re, score, n_cigar, NM);
    const double t0 = now_s();
    uint32_t* r = next(mat, o_del, e_del, o_ins, e_ins, w_, l_pac, pac, l_query, query, rb, re, score, n_cigar, NM);
    g_t_cigar = g_t_cigar + (now_s() - t0);
    g_n_cigar += 1;
    return r;
}
typedef void (*aln2sam_fn)(const mem_opt_t*, const bntseq_t*, kstring_t*, bseq1_t*, int, const mem_aln_t*, int, const mem_aln_t*);
void mem_aln2sam(const mem_opt_t* opt, const bntseq_t* bns, kstring_t* str, bseq1_t* s, int n, const mem_aln_t* list, int which, const mem_aln_t* m) {
    static aln2sam_fn next = (aln2sam_fn)dlsym(RTLD_NEXT, "_Z11mem_aln2samPK9mem_opt_tPK8bntseq_tP11__kstring_tP7bseq1_tiPK9mem_aln_tiSB_");
    if (!profile_sam()) { next(opt, bns, str, s, n, list, which, m); return; }
    const double t0 = now_s();
    next(opt, bns, str, s, n, list, which, m);
    g_t_sam = g_t_sam + (now_s() - t0);
    g_n_sam += 1;
}
void meme_dropin_report_matesw() {
    fprintf(stderr, "[meme-dropin] SAM phase on the host, thread-seconds so far: mate-rescue SW (kswv) %.3f for %lld pairs; CIGAR generation (bwa_gen_cigar2) %.3f "
            "for %lld alignments; SAM formatting (mem_aln2sam) %.3f for %lld records\n", (double)g_t_matesw, (long long)g_n_matesw, (double)g_t_cigar,
            (long long)g_n_cigar, (double)g_t_sam, (long long)g_n_sam);
}

// ---- CIGAR generation of the SAM phase on the device ---------------------------------------------------------------------
// mem_reg2aln (src/bwamem.cpp:2314-2380) calls bwa_gen_cigar2 (src/bwa.cpp:274-362) up to three times per alignment written out, and
// that runs ksw_global2 (src/ksw.cpp:560-670): banded global alignment with traceback, 42 % of the SAM phase's thread time on 250-bp
// reads with 5 % errors.  Between the two kt_for phases -- worker_aln has joined, worker_sam has not started: the third kt_for call of
// mem_process_seqs (src/bwamem.cpp:1941-1965) is interposed -- the binding poses the same alignments for EVERY alignment record of the
// chunk (the records are complete and nobody touches them; same band arithmetic as mem_reg2aln / bwa_gen_cigar2), runs them
// on the GPU(s) as one batch per band attempt (meme_global_batch_host) and keeps score + CIGAR; ksw_global2 calls are then answered
// from that table after an exact comparison of both sequences.  Calls the table does not hold (alignments made later by mate rescue,
// calls without traceback from mem_patch_reg) go to the reference's function.  MEME_DROPIN_CIGAR=0 switches the stage off.
#include <omp.h>
#include <parallel/algorithm>
namespace {

struct CigEntry { int64_t g; int64_t rb; int32_t qb, qlen, tlen, w, rev, score, n_cigar; int64_t ops; };
struct CigTable {
    std::mutex mu;
    uint64_t gen = 0;
    std::vector<CigEntry> e;
    std::vector<uint32_t> ops;
    std::vector<std::pair<uint64_t, uint32_t>> idx;      // (key, entry), sorted
    double t_prepass = 0, t_kernel_ms = 0;
    int64_t n_jobs = 0;
} g_cig;
std::atomic<int64_t> g_cig_hits{0}, g_cig_miss{0};
bool cigar_on_device() { static const bool v = !(getenv("MEME_DROPIN_CIGAR") && atoi(getenv("MEME_DROPIN_CIGAR")) == 0); return v; }

inline uint64_t mix64(uint64_t h, uint64_t v) { h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); return h * 0xff51afd7ed558ccdull; }
inline uint64_t hash_bytes(const uint8_t* p, int n, bool rev) {
    uint64_t h = 1469598103934665603ull;
    if (!rev) for (int i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
    else for (int i = n - 1; i >= 0; --i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}
inline uint64_t cig_key(int qlen, int tlen, int w, uint64_t hq, uint64_t ht) {
    return mix64(mix64(mix64(mix64((uint64_t)qlen, (uint64_t)tlen), (uint64_t)w), hq), ht);
}
inline int infer_bw_(int l1, int l2, int score, int a, int q, int r) {        // infer_bw, src/bwamem.cpp:2151-2158
    if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
    int w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
    if (w < abs(l1 - l2)) w = abs(l1 - l2);
    return w;
}
// the band bwa_gen_cigar2 hands to ksw_global2 for a call with w_ (src/bwa.cpp:306-316); false: no DP (rejected, or the gap-free shortcut)
inline bool gen_cigar_band(const mem_opt_t* opt, int64_t l_pac, int l_query, int64_t rb, int64_t re, int w_, int* w_out) {
    if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return false;
    const int64_t rlen = re - rb;
    if (l_query == rlen && w_ == 0) return false;
    int max_ins = (int)((double)(((l_query + 1) >> 1) * opt->mat[0] - opt->o_ins) / opt->e_ins + 1.);
    int max_del = (int)((double)(((l_query + 1) >> 1) * opt->mat[0] - opt->o_del) / opt->e_del + 1.);
    int max_gap = max_ins > max_del ? max_ins : max_del;
    max_gap = max_gap > 1 ? max_gap : 1;
    int w = (max_gap + abs((int)rlen - l_query) + 1) >> 1;
    w = w < w_ ? w : w_;
    const int min_w = abs((int)rlen - l_query) + 3;
    w = w > min_w ? w : min_w;
    *w_out = w;
    return true;
}

// helper threads of the pre-pass's host loops: a few dozen are enough, and an OpenMP team of 256 would still be spinning when worker_sam starts
int cig_threads() { const int m = omp_get_max_threads(); return m < 32 ? m : 32; }

void cig_prepass() {
    const double t0 = now_s();
    CigTable& T = g_cig;
    T.e.clear(); T.ops.clear(); T.idx.clear();
    const mem_opt_t* opt = g_opt;
    const int64_t n = g_chunk.n, l_pac = g_bns->l_pac;
    // per alignment record: where mem_reg2aln's loop stands (band of the next call, score of the last one)
    struct Cand { int64_t g; int32_t reg, w2, last_sc, tries; };
    std::vector<Cand> cand;
    {
        const int nt = cig_threads();
        std::vector<std::vector<Cand>> part((size_t)nt);
#pragma omp parallel num_threads(nt)
        {
            std::vector<Cand>& mine = part[(size_t)omp_get_thread_num()];
#pragma omp for schedule(static)
            for (int64_t g = 0; g < n; ++g) {
                const mem_alnreg_v& av = g_worker->regs[g];
                for (size_t i = 0; i < av.n; ++i) {
                    const mem_alnreg_t& p = av.a[i];
                    if (p.rb < 0 || p.re < 0 || p.score < opt->T) continue;
                    if (p.secondary >= 0 && p.secondary < (int)av.n && p.score < av.a[p.secondary].score * opt->XA_drop_ratio) continue;
                    const int tmp = infer_bw_(p.qe - p.qb, (int)(p.re - p.rb), p.truesc, opt->a, opt->o_del, opt->e_del);
                    int w2 = infer_bw_(p.qe - p.qb, (int)(p.re - p.rb), p.truesc, opt->a, opt->o_ins, opt->e_ins);
                    w2 = w2 > tmp ? w2 : tmp;
                    if (w2 > opt->w) w2 = w2 < p.w ? w2 : p.w;
                    mine.push_back({g, (int32_t)i, w2, -(1 << 30), 0});
                }
            }
        }
        size_t tot = 0;
        for (auto& v : part) tot += v.size();
        cand.reserve(tot);
        for (auto& v : part) cand.insert(cand.end(), v.begin(), v.end());      // (static schedule: still in read order)
    }
    const int nd = (int)g_dev.size();
    meme_bsw_opt bo;
    memset(&bo, 0, sizeof(bo));
    bo.o_del = opt->o_del; bo.e_del = opt->e_del; bo.o_ins = opt->o_ins; bo.e_ins = opt->e_ins; bo.a = opt->a; bo.b = opt->b;
    for (int round = 0; round < 3 && !cand.empty(); ++round) {
        // this round's calls, per device part (candidates are in read order: a part's candidates are contiguous)
        std::vector<std::vector<meme_gjob>> jobs((size_t)nd);
        std::vector<std::vector<uint32_t>> who((size_t)nd);
        for (size_t c = 0; c < cand.size(); ++c) {
            Cand& C = cand[c];
            const mem_alnreg_t& p = g_worker->regs[C.g].a[C.reg];
            C.w2 = C.w2 < opt->w << 2 ? C.w2 : opt->w << 2;
            int w = 0;
            if (!gen_cigar_band(opt, l_pac, p.qe - p.qb, p.rb, p.re, C.w2, &w)) { C.tries = 99; continue; }
            int d = 0;
            while (d + 1 < nd && C.g >= g_chunk.part[(size_t)d].first + g_chunk.part[(size_t)d].count) ++d;
            meme_gjob J;
            J.rb = p.rb; J.read = (int32_t)(C.g - g_chunk.part[(size_t)d].first); J.qb = p.qb; J.qlen = p.qe - p.qb; J.tlen = (int32_t)(p.re - p.rb); J.w = w;
            J.rev = p.rb >= l_pac ? 1 : 0;
            jobs[(size_t)d].push_back(J);
            who[(size_t)d].push_back((uint32_t)c);
        }
        std::vector<meme_gres_host> res((size_t)nd);
        std::vector<std::thread> th;
        auto run = [&](int d) {
            memset(&res[(size_t)d], 0, sizeof(meme_gres_host));
            if (jobs[(size_t)d].empty()) return;
            if (meme_global_batch_host(g_dev[(size_t)d].seed, jobs[(size_t)d].data(), (int64_t)jobs[(size_t)d].size(), &bo, &res[(size_t)d])) die("meme_global_batch_host");
        };
        for (int d = 1; d < nd; ++d) th.emplace_back(run, d);
        run(0);
        for (auto& t : th) t.join();
        std::vector<Cand> next;
        for (int d = 0; d < nd; ++d) {
            const meme_gres_host& R = res[(size_t)d];
            if (R.njobs == 0) continue;
            T.t_kernel_ms += R.kernel_ms;
            T.n_jobs += R.njobs;
            const size_t e0 = T.e.size(), o0 = T.ops.size();
            T.ops.insert(T.ops.end(), R.cigars, R.cigars + R.total_ops);      // the device packs the operations in job order
            T.e.resize(e0 + (size_t)R.njobs);
#pragma omp parallel for schedule(static) num_threads(cig_threads())
            for (int64_t k = 0; k < R.njobs; ++k) {
                const meme_gjob& J = jobs[(size_t)d][(size_t)k];
                CigEntry& E = T.e[e0 + (size_t)k];
                E.g = cand[who[(size_t)d][(size_t)k]].g; E.rb = J.rb; E.qb = J.qb; E.qlen = J.qlen; E.tlen = J.tlen; E.w = J.w; E.rev = J.rev;
                E.score = R.res[k].score; E.n_cigar = R.res[k].n_cigar; E.ops = (int64_t)o0 + R.res[k].cigar_off;
            }
            for (int64_t k = 0; k < R.njobs; ++k) {
                // mem_reg2aln's loop (:2340-2347): again with the doubled band while the global score stays below the local one
                Cand& C = cand[who[(size_t)d][(size_t)k]];
                const mem_alnreg_t& p = g_worker->regs[C.g].a[C.reg];
                const int score = R.res[k].score;
                if (score == C.last_sc || C.w2 == opt->w << 2) continue;
                C.last_sc = score;
                C.w2 <<= 1;
                if (++C.tries < 3 && score < p.truesc - opt->a) next.push_back(C);
            }
        }
        cand.swap(next);
    }
    // index: sequences hashed the way the hook will see them (both reversed on the reverse strand); sorted by key, looked up by bisection
    const uint8_t* ref = g_worker->ref_string;
    T.idx.resize(T.e.size());
#pragma omp parallel for schedule(static) num_threads(cig_threads())
    for (int64_t k = 0; k < (int64_t)T.e.size(); ++k) {
        const CigEntry& E = T.e[(size_t)k];
        const uint8_t* q = (const uint8_t*)g_chunk.seqs[E.g].seq + E.qb;
        T.idx[(size_t)k] = {cig_key(E.qlen, E.tlen, E.w, hash_bytes(q, E.qlen, E.rev), hash_bytes(ref + E.rb, E.tlen, E.rev)), (uint32_t)k};
    }
    __gnu_parallel::sort(T.idx.begin(), T.idx.end(), __gnu_parallel::default_parallel_tag((unsigned)cig_threads()));
    T.t_prepass += now_s() - t0;
}

typedef int (*ksw_global2_fn)(int, const uint8_t*, int, const uint8_t*, int, const int8_t*, int, int, int, int, int, int*, uint32_t**);
}  // namespace

extern "C" int ksw_global2(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat, int o_del, int e_del, int o_ins,
                           int e_ins, int w, int* n_cigar_, uint32_t** cigar_) {
    static ksw_global2_fn next = (ksw_global2_fn)dlsym(RTLD_NEXT, "ksw_global2");
    if (!next) { fprintf(stderr, "[meme-dropin] the reference's ksw_global2 was not found\n"); exit(1); }
    const mem_opt_t* opt = g_opt;
    if (!cigar_on_device() || !n_cigar_ || !cigar_ || !g_chunk.seqs || !g_worker || !opt || g_dev.empty() || m != 5 || mat != opt->mat || o_del != opt->o_del ||
        e_del != opt->e_del || o_ins != opt->o_ins || e_ins != opt->e_ins)
        return next(qlen, query, tlen, target, m, mat, o_del, e_del, o_ins, e_ins, w, n_cigar_, cigar_);
    CigTable& T = g_cig;
    if (T.gen != g_chunk_gen) return next(qlen, query, tlen, target, m, mat, o_del, e_del, o_ins, e_ins, w, n_cigar_, cigar_);   // (no table for this chunk)
    const uint64_t key = cig_key(qlen, tlen, w, hash_bytes(query, qlen, false), hash_bytes(target, tlen, false));
    const uint8_t* ref = g_worker->ref_string;
    for (auto it = std::lower_bound(T.idx.begin(), T.idx.end(), std::make_pair(key, (uint32_t)0)); it != T.idx.end() && it->first == key; ++it) {
        const CigEntry& E = T.e[it->second];
        if (E.qlen != qlen || E.tlen != tlen || E.w != w) continue;
        const uint8_t* q = (const uint8_t*)g_chunk.seqs[E.g].seq + E.qb;
        const uint8_t* t = ref + E.rb;
        bool same = true;
        if (!E.rev) same = !memcmp(q, query, (size_t)qlen) && !memcmp(t, target, (size_t)tlen);
        else {
            for (int i = 0; same && i < qlen; ++i) same = q[qlen - 1 - i] == query[i];
            for (int i = 0; same && i < tlen; ++i) same = t[tlen - 1 - i] == target[i];
        }
        if (!same) continue;
        uint32_t* cg = (uint32_t*)malloc((size_t)(E.n_cigar > 0 ? E.n_cigar : 1) * 4);   // the caller owns (and grows) it, as with the reference's
        if (!cg) { fprintf(stderr, "[meme-dropin] out of memory\n"); exit(1); }
        memcpy(cg, T.ops.data() + E.ops, (size_t)E.n_cigar * 4);
        *cigar_ = cg;
        *n_cigar_ = E.n_cigar;
        g_cig_hits.fetch_add(1, std::memory_order_relaxed);
        return E.score;
    }
    g_cig_miss.fetch_add(1, std::memory_order_relaxed);
    return next(qlen, query, tlen, target, m, mat, o_del, e_del, o_ins, e_ins, w, n_cigar_, cigar_);
}

void meme_dropin_report_cigar() {
    if (!cigar_on_device()) return;
    fprintf(stderr, "[meme-dropin] CIGAR stage on the device: %lld global alignments with traceback posed so far (kernels %.3f s, whole pre-pass %.3f s); "
            "ksw_global2 calls answered from the table %lld, computed by the reference's function %lld (alignments made by mate rescue, calls without traceback)\n",
            (long long)g_cig.n_jobs, g_cig.t_kernel_ms * 1e-3, g_cig.t_prepass, (long long)g_cig_hits.load(), (long long)g_cig_miss.load());
}

// kt_for (src/kthread.cpp:79-114) is called three times per chunk by mem_process_seqs: worker_bwt, worker_aln, worker_sam.  Before the
// third call every alignment record of the chunk exists and no worker thread is running: the CIGAR stage's quiescent point.
namespace { std::atomic<int> g_ktfor_calls{0}; std::atomic<int>& ktfor_calls() { return g_ktfor_calls; } }
typedef void (*kt_for_fn)(void (*)(void*, long, long, int), void*, int);
void kt_for(void (*func)(void*, long, long, int), void* data, int n) {
    static kt_for_fn next = (kt_for_fn)dlsym(RTLD_NEXT, "_Z6kt_forPFvPvlliES_i");
    if (!next) { fprintf(stderr, "[meme-dropin] the reference's kt_for was not found\n"); exit(1); }
    if (g_chunk.seqs && data == (void*)g_worker && g_ktfor_calls.fetch_add(1) == 2) {
        bool mate = false;
#if __AVX512BW__          // (only this build of the reference batches mate rescue: src/bwamem.cpp:1838)
        mate = matesw_on_device() && (g_opt->flag & MEM_F_PE) && !(g_opt->flag & MEM_F_NO_RESCUE) && !g_dev.empty();
#endif
        // (a chunk that will not reach the job threshold -- by the previous chunk's jobs per read -- is not posed at all)
        if (mate && g_mate_jobs_per_read >= 0 && g_mate_jobs_per_read * (double)g_chunk.n < (double)matesw_min_jobs()) mate = false;
        if (cigar_on_device()) {
            std::lock_guard<std::mutex> lk(g_cig.mu);
            cig_prepass();
            g_cig.gen = g_chunk_gen;
        }
        if (mate && matesw_prepass()) {
            next(sam_worker_dev, data, n);
            return;
        }
    }
    next(func, data, n);
}

// ---- FASTQ input (first step): the two mate files parsed by two threads, ahead of the pipeline ---------------------
// bseq_read_orig() (src/bwa.cpp:184-230) parses both files of a paired run with one thread, read by read; with the backend bound
// that parser is the longest stage of the aligner's three-stage pipeline (0.9 s per 100 M-base chunk against 0.5-0.7 s of compute).
// The records come from the same kseq_read() calls on the same streams, in the same order -- only that each stream has a thread
// of its own that keeps a bounded queue filled (records travel in batches of 4 096), and the pipeline's step 0 takes what is ready.
// MEME_DROPIN_IO=0 switches it off (the reference's reader).
#include <deque>
namespace {

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

}  // namespace

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
    return seqs;
}
