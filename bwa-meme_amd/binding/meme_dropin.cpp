// Reference-side binding of the MI355X backend (what INTEGRATION.md describes), written against the *reference's own
// headers* and linked into the reference aligner (oracle/Makefile.ref target bwa-meme_dropin).  The reference objects
// are built position-independent into libbwa_pic.so; the definitions below live in the executable and therefore win
// symbol resolution (ELF interposition) -- no reference source is modified or copied, the calls below go to functions
// the reference exports.  The binding is four translation units around meme_dropin.h (this file: devices, index, the chunk-level
// calls, mem_kernel1_core_Learned; meme_dropin_ext.cpp: the BandedPairWiseSW entry points and the hand-over of the device's alignment
// records; meme_dropin_sam.cpp: the SAM phase's tables; meme_dropin_io.cpp: the FASTQ reader and the output step);
// reference_Makefile.patch adds a `dropin` target to the reference's own Makefile.  A maintainer integrating the backend would put the
// same code behind an #ifdef at the places named here:
//
//   memoryAllocLearned()            src/fastmap.cpp:351-641   worker buffers as before, but the index files stream to HBM
//                                   (meme_index_load_files + meme_index_replicate per extra GPU) instead of being
//                                   expanded on the host (13-byte suffix-array entries + ISA, ~200 s / ~120 GB at GRCh38)
//   mem_process_seqs()              src/bwamem.cpp:1920-1972  per -K chunk (split over the visible GPUs), before kt_for(worker_bwt):
//                                   meme_seed_batch_resident_ascii() + meme_extend_last_batch_host() -- seeding, mem_chain_Learned,
//                                   mem_chain_flt, mem_flt_chained_seeds and mem_chain2aln_across_reads_V2 on the device; the next
//                                   chunk's run ahead beside this chunk's SAM phase; then the reference's own body
//   mem_kernel1_core_Learned()      src/bwamem.cpp:1230-1413  per 512-read batch: nothing left to do (MEME_DROPIN_EXT=0, the
//                                   cross-check: takes the chunk's chains from the device, mem_flt_chained_seeds as before)
//   mem_chain2aln_across_reads_V2() src/bwamem.cpp:2573-3497  per batch: takes its reads' alignment records
//   BandedPairWiseSW::getScores8 / getScores16 / scalarBandedSWAWrapper   src/bandedSWA.cpp:242-260,1970-,2664-
//                                   -> meme_bsw_batch(); concurrent calls of the kt_for workers combined into one backend
//                                   call per GPU (group commit) -- the path of MEME_DROPIN_EXT=0
#include "meme_dropin.h"
#include <malloc.h>
#include "meme_letters.h"
#include <dirent.h>
#include <unistd.h>
#include <map>
#include <string>

#define dropin_smem_lt(a, b) ((a).start == (b).start ? (a).end < (b).end : (a).start < (b).start)
KSORT_INIT(meme_dropin_smem, mem_tl, dropin_smem_lt)

using namespace dropin;

namespace dropin {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
[[noreturn]] void die(const char* what) {
    fprintf(stderr, "[meme-dropin] %s: %s\n", what, meme_last_error());
    exit(1);
}
bool verbose() { static const bool v = getenv("MEME_DROPIN_VERBOSE") != nullptr; return v; }

// ---- MEME_DROPIN_VERIFY (see meme_dropin.h) -------------------------------------------------------------------------------------
bool verify_on() { static const bool v = getenv("MEME_DROPIN_VERIFY") && atoi(getenv("MEME_DROPIN_VERIFY")) != 0; return v; }
uint64_t verify_hash(const void* p, size_t bytes, uint64_t h) {
    const uint8_t* b = (const uint8_t*)p;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) { uint64_t v; memcpy(&v, b + i, 8); h = (h ^ v) * 0xff51afd7ed558ccdull; h ^= h >> 32; }
    if (i < bytes) { uint64_t v = 0; memcpy(&v, b + i, bytes - i); h = (h ^ v) * 0xff51afd7ed558ccdull; h ^= h >> 32; }
    return h;
}
void verify_note(int64_t chunk_seq, const char* stage, int dev, uint64_t hash, int64_t items) {
    fprintf(stderr, "[meme-dropin] verify chunk %lld %s dev %d: %lld items, hash %016llx, identical on two ctxs\n", (long long)chunk_seq, stage, dev, (long long)items, (unsigned long long)hash);
}
[[noreturn]] void verify_fail(const char* stage, int64_t item, const char* what) {
    fprintf(stderr, "[meme-dropin] VERIFY FAILED: the %s stage gave different results on two ctxs of one GPU for the same input (first difference: item %lld%s%s)\n", stage, (long long)item,
            what && *what ? ", " : "", what ? what : "");
    exit(3);
}

// ---- the reference's functions behind the interposed ones (see meme_dropin.h) ---------------------------------------------------------------
namespace {
struct RefEntry { const char* what; const char* symbol; void* p; };
RefEntry g_ref[R_N_SYMS] = {
    {"mem_process_seqs (src/bwamem.cpp:1920)", "_Z16mem_process_seqsP9mem_opt_tliP7bseq1_tPK12mem_pestat_tR8worker_t", nullptr},
    {"mem_chain2aln_across_reads_V2 (src/bwamem.cpp:2573)", "_Z29mem_chain2aln_across_reads_V2PK9mem_opt_tPK8bntseq_tPKhP7bseq1_tiP11mem_chain_vP12mem_alnreg_vP9mem_cachePhi", nullptr},
    {"bseq_read_orig (src/bwa.cpp:184)", "bseq_read_orig", nullptr},
    {"kt_pipeline, the step function (src/fastmap.cpp:843)", "_Z11kt_pipelinePviS_P9mem_opt_tR8worker_t", nullptr},
    {"mem_sam_pe_batch (src/bwamem_pair.cpp:719)", "_Z16mem_sam_pe_batchPK9mem_opt_tP9mem_cacheRlS4_P6kswr_tiii", nullptr},
    {"bwa_gen_cigar2 (src/bwa.cpp:274)", "bwa_gen_cigar2", nullptr},
    {"mem_aln2sam (src/bwamem.cpp:2174)", "_Z11mem_aln2samPK9mem_opt_tPK8bntseq_tP11__kstring_tP7bseq1_tiPK9mem_aln_tiSB_", nullptr},
    {"mem_pestat (src/bwamem_pair.cpp:81)", "_Z10mem_pestatPK9mem_opt_tliPK12mem_alnreg_vP12mem_pestat_t", nullptr},
    {"kt_for (src/kthread.cpp:79)", "_Z6kt_forPFvPvlliES_i", nullptr},
};
__attribute__((constructor)) void resolve_reference_symbols() {
    int missing = 0;
    for (RefEntry& e : g_ref) if (!(e.p = dlsym(RTLD_NEXT, e.symbol))) ++missing;
    if (!missing) return;
    fprintf(stderr, "[meme-dropin] the reference library does not export %d of the %d functions the binding stands in front of:\n", missing, (int)R_N_SYMS);
    for (const RefEntry& e : g_ref) if (!e.p) fprintf(stderr, "[meme-dropin]   %s as %s\n", e.what, e.symbol);
    fprintf(stderr, "[meme-dropin] (check with: nm -D libbwa_pic.so | grep -E 'mem_process_seqs|kt_for|...'; C++ names are mangled from the signatures in the reference's headers)\n");
    exit(1);
}
}  // namespace
void* ref_sym(RefSym which) { return g_ref[which].p; }

// ---- the helper team (see meme_dropin.h) ----------------------------------------------------------------------------------------
namespace {
struct TeamJob { const std::function<void(int)>* f; int nt; std::atomic<int> next{0}, done{0}; const char* label = nullptr; };
std::mutex g_label_mu;
std::map<std::string, double>& label_cpu() { static auto* m = new std::map<std::string, double>(); return *m; }
inline double thread_cpu_s() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
inline void run_share(TeamJob* J, int t) {
    if (!verbose()) { (*J->f)(t); return; }
    const double c0 = thread_cpu_s();
    (*J->f)(t);
    const double c = thread_cpu_s() - c0;
    std::lock_guard<std::mutex> lk(g_label_mu);
    label_cpu()[J->label ? J->label : "(unlabelled)"] += c;
}
struct Team {
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::deque<TeamJob*> q;
    int threads = 0;
    void worker() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv_work.wait(lk, [&] { return !q.empty(); });
            TeamJob* J = q.front();
            const int t = J->next.fetch_add(1);
            if (t >= J->nt) { if (!q.empty() && q.front() == J) q.pop_front(); continue; }
            if (t + 1 == J->nt && !q.empty() && q.front() == J) q.pop_front();      // the last share is taken: nobody else needs to look at this job
            lk.unlock();
            run_share(J, t);
            lk.lock();
            if (J->done.fetch_add(1) + 1 == J->nt) cv_done.notify_all();
        }
    }
};
Team& team() { static Team* T = new Team; return *T; }                          // (never destroyed: its threads sleep until the process ends)
}  // namespace
thread_local const char* tl_team_label = nullptr;
void team_run(int nt, const std::function<void(int)>& f) {
    if (nt <= 1) { if (nt == 1) f(0); return; }
    Team& T = team();
    TeamJob J;
    J.f = &f; J.nt = nt; J.label = tl_team_label;
    {
        std::lock_guard<std::mutex> lk(T.m);
        const int want = nt - 1 < 48 ? nt - 1 : 48;
        while (T.threads < want) { std::thread([&T] { pthread_setname_np(pthread_self(), "meme-team"); T.worker(); }).detach(); ++T.threads; }
        T.q.push_back(&J);
    }
    T.cv_work.notify_all();
    for (;;) {                                                       // the submitter takes shares of its own job
        const int t = J.next.fetch_add(1);
        if (t >= nt) break;
        run_share(&J, t);
        J.done.fetch_add(1);
    }
    std::unique_lock<std::mutex> lk(T.m);
    for (auto it = T.q.begin(); it != T.q.end(); ++it) if (*it == &J) { T.q.erase(it); break; }    // (all shares are out: no helper may pick the job up any more)
    T.cv_done.wait(lk, [&] { return J.done.load() >= nt; });
}

// ---- devices -------------------------------------------------------------------------------------------------
std::vector<Device>& device_slots() { static std::vector<Device>* v = new std::vector<Device>(); return *v; }
std::mutex g_mu;
std::atomic<double> g_t_seed{0}, g_t_seed_call{0}, g_t_bsw_gather{0}, g_t_bsw_call{0}, g_t_bsw_kernel{0};
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
    // A thread that waits for the GPU sleeps instead of spinning (the HIP runtime's default for a lightly threaded process is to spin):
    // the aligner's worker threads want every core the host gives the process -- and under a CPU quota (the boxes this was measured on
    // give a process 16 CPUs' worth of a 256-thread machine) a spinning waiter is paid for with the workers' time.  The flag belongs to
    // the device's primary context and has to be set before the first ctx creates it.  MEME_DROPIN_BLOCKING_SYNC=0: the runtime's default.
    if (!(getenv("MEME_DROPIN_BLOCKING_SYNC") && atoi(getenv("MEME_DROPIN_BLOCKING_SYNC")) == 0)) {
        typedef int (*set_dev_fn)(int);
        typedef int (*set_flags_fn)(unsigned);
        const set_dev_fn set_dev = (set_dev_fn)dlsym(RTLD_DEFAULT, "hipSetDevice");
        const set_flags_fn set_flags = (set_flags_fn)dlsym(RTLD_DEFAULT, "hipSetDeviceFlags");
        if (set_dev && set_flags)
            for (int d = 0; d < n_real; ++d) if (set_dev(d) == 0) (void)set_flags(0x4u /* hipDeviceScheduleBlockingSync */);
    }
    for (int d = 0; d < n; ++d) {
        if (!(g_dev[(size_t)d].seed = meme_ctx_create(d % n_real))) die("meme_ctx_create");
        if (!(g_dev[(size_t)d].bsw = meme_ctx_create(d % n_real))) die("meme_ctx_create");
        // small combined calls keep the lanes-per-pair kernels; combined calls of the whole thread team are big enough
        // for the lane-per-pair kernel much earlier than a lone caller's
        if (getenv("MEME_DROPIN_BSW_LANE_MIN")) meme_set_tuning(g_dev[(size_t)d].bsw, "bsw_lane_min_pairs", atoll(getenv("MEME_DROPIN_BSW_LANE_MIN")));
        if (getenv("MEME_DROPIN_MAX_BATCH")) meme_set_tuning(g_dev[(size_t)d].seed, "max_batch", atoll(getenv("MEME_DROPIN_MAX_BATCH")));   // a memory bound (and the tests' way to the split-and-retry paths)
        // the second slot's ctx (chunks alternate between two, see "the next chunk ahead of its turn"): created and given its buffers
        // now, while the index loads -- its first chunk otherwise pays 0.25 s of allocations in the middle of the run
        if (getenv("MEME_DROPIN_SAM_MAX_BATCH")) meme_set_tuning(g_dev[(size_t)d].seed, "sam_max_batch", atoll(getenv("MEME_DROPIN_SAM_MAX_BATCH")));   // (tests: the SAM text stage in pieces / by the reference's function)
        if (verify_on() && !(g_dev[(size_t)d].vfy_bsw = meme_ctx_create(d % n_real))) die("meme_ctx_create");
        if (verify_on())
            for (int k = 0; k < 2; ++k) {
                if (!(g_dev[(size_t)d].vfy[k] = meme_ctx_create(d % n_real))) die("meme_ctx_create");
                if (getenv("MEME_DROPIN_MAX_BATCH")) meme_set_tuning(g_dev[(size_t)d].vfy[k], "max_batch", atoll(getenv("MEME_DROPIN_MAX_BATCH")));
            }
        if (prefetch_on() && ext_mode() == 2) {
            if (!(g_dev[(size_t)d].seed2 = meme_ctx_create(d % n_real))) die("meme_ctx_create");
            if (getenv("MEME_DROPIN_MAX_BATCH")) meme_set_tuning(g_dev[(size_t)d].seed2, "max_batch", atoll(getenv("MEME_DROPIN_MAX_BATCH")));
            if (getenv("MEME_DROPIN_SAM_MAX_BATCH")) meme_set_tuning(g_dev[(size_t)d].seed2, "sam_max_batch", atoll(getenv("MEME_DROPIN_SAM_MAX_BATCH")));
        }
    }
    // while the index streams in: the seeding / chaining buffers of a chunk on every device slot (pinned memory is slow to allocate)
    std::thread reserve([n, chunk_reads] {
        int64_t per = chunk_reads / n + BATCH_SIZE;
        if (per > (1 << 20)) per = 1 << 20;                  // beyond a million reads per device the buffers grow on first use
        for (int d = 0; d < n; ++d) if (meme_seed_reserve(g_dev[(size_t)d].seed, per, per * READ_LEN)) die("meme_seed_reserve");
        for (int d = 0; d < n; ++d) if (g_dev[(size_t)d].seed2 && meme_seed_reserve(g_dev[(size_t)d].seed2, per, per * READ_LEN)) die("meme_seed_reserve");
    });
    if (meme_index_load_files(g_dev[0].seed, prefix)) die("meme_index_load_files");
    reserve.join();
    const double t1 = now_s();
    std::vector<std::thread> th;
    for (int d = 1; d < n; ++d)                                 // device-to-device over xGMI, all replicas at once
        th.emplace_back([d] { if (meme_index_replicate(g_dev[(size_t)d].seed, g_dev[0].seed)) die("meme_index_replicate"); });
    for (auto& t : th) t.join();
    for (int d = 0; d < n; ++d) if (g_dev[(size_t)d].seed2 && meme_index_share(g_dev[(size_t)d].seed2, g_dev[(size_t)d].seed)) die("meme_index_share");
    for (int d = 0; d < n; ++d) for (int k = 0; k < 2; ++k) if (g_dev[(size_t)d].vfy[k] && meme_index_share(g_dev[(size_t)d].vfy[k], g_dev[(size_t)d].seed)) die("meme_index_share");
    for (int d = 0; d < n; ++d) if (meme_index_share(g_dev[(size_t)d].bsw, g_dev[(size_t)d].seed)) die("meme_index_share");
    for (int d = 0; d < n; ++d) if (g_dev[(size_t)d].vfy_bsw && meme_index_share(g_dev[(size_t)d].vfy_bsw, g_dev[(size_t)d].seed)) die("meme_index_share");      // (mate rescue poses its jobs from the text: round 6)
    fprintf(stderr, "[meme-dropin] index staged in HBM in %.2f s, replicated to %d more GPU(s) in %.2f s\n", t1 - t0, n - 1,
            now_s() - t1);
}

// ---- early start -------------------------------------------------------------------------------------------------------------
// When the index prefix comes through the environment (MEME_INDEX_PREFIX) and the process is `... mem ... -7 ...`, the index starts
// streaming to HBM when the binding is loaded -- while the aligner still parses its arguments and reads its own copy of the reference
// (bns, pac, the 6 GB .0123 text: 1.5 s at GRCh38 size) -- instead of when memoryAllocLearned() is reached.  MEME_DROPIN_EARLY=0: off.
std::thread* g_early = nullptr;
const char* g_early_prefix = nullptr;
// MEME_DROPIN_VERBOSE: where the process's CPU seconds went, by thread role, when the run ends (what a host with a CPU quota is short of): the threads that
// are still alive -- the binding's helper team (every host loop of the binding), its prefetcher (the device stages' submission, HIP runtime included), its
// FASTQ parsers, the aligner's two pipeline threads (the pre-passes' serial parts, the output step) -- and, as the remainder, the aligner's kt_for workers
// (they end with their phase: the reference's own worker_aln / worker_sam code and the hooks called from it).
std::mutex g_role_mu;
std::map<std::string, std::pair<double, int>>& role_cpu() { static auto* m = new std::map<std::string, std::pair<double, int>>(); return *m; }
void note_thread_cpu(const char* role) {
    timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    std::lock_guard<std::mutex> lk(g_role_mu);
    auto& e = role_cpu()[role];
    e.first += (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; e.second += 1;
}
void report_thread_cpu() {
    if (!verbose()) return;
    const double tick = (double)sysconf(_SC_CLK_TCK);
    std::map<std::string, std::pair<double, int>> by;
    double live = 0;
    char path[256], buf[1024];
    for (int pass = 0; pass < 1; ++pass) {
        DIR* d = opendir("/proc/self/task");
        if (!d) return;
        while (dirent* e = readdir(d)) {
            if (e->d_name[0] == '.') continue;
            snprintf(path, sizeof(path), "/proc/self/task/%s/stat", e->d_name);
            FILE* f = fopen(path, "r");
            if (!f) continue;
            const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
            fclose(f);
            buf[n] = 0;
            char* l = strchr(buf, '('); char* r = strrchr(buf, ')');
            if (!l || !r) continue;
            std::string name(l + 1, r);
            unsigned long ut = 0, st = 0;
            // fields after the name: state ppid pgrp session tty tpgid flags minflt cminflt majflt cmajflt utime stime
            if (sscanf(r + 2, "%*c %*d %*d %*d %*d %*d %*u %*u %*u %*u %*u %lu %lu", &ut, &st) != 2) continue;
            const double c = (double)(ut + st) / tick;
            by[name].first += c; by[name].second += 1; live += c;
        }
        closedir(d);
    }
    timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts);
    const double total = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    fprintf(stderr, "[meme-dropin] process CPU %.2f s by thread role (threads alive at the end; the remainder = threads that ended, i.e. the aligner's kt_for workers):", total);
    for (const auto& kv : by) fprintf(stderr, " %s x%d %.2f s;", kv.first.c_str(), kv.second.second, kv.second.first);
    double noted = 0;
    { std::lock_guard<std::mutex> lk(g_role_mu); for (const auto& kv : role_cpu()) { fprintf(stderr, " %s x%d (ended) %.2f s;", kv.first.c_str(), kv.second.second, kv.second.first); noted += kv.second.first; } }
    fprintf(stderr, " other ended threads (kt_for workers) %.2f s\n", total - live - noted);
    { std::lock_guard<std::mutex> lk(g_label_mu); fprintf(stderr, "[meme-dropin] the helper team's CPU by host loop (submitters' own shares included):"); for (const auto& kv : label_cpu()) fprintf(stderr, " %s %.2f s;", kv.first.c_str(), kv.second); fprintf(stderr, "\n"); }
}

__attribute__((constructor)) void meme_dropin_early_start() {
    atexit(report_thread_cpu);
    // The chaining stage runs four kernels side by side on streams of their own; the HIP runtime multiplexes a process's streams onto 4
    // hardware queues unless told otherwise when it initialises (7.9 -> 7.4 ms per 2 M reads with 8).  This is the aligner's own start-up
    // code, before its first HIP call and before it has threads: the place for it (the backend library itself no longer touches the environment).
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    // The SAM phase is allocator-bound (the reference's default build links mimalloc for that reason; this build, like the reference binary
    // it is compared with, runs on glibc malloc): keep freed memory in the arenas instead of handing it back to the kernel between chunks
    // and grow the heaps in large steps.  MEME_DROPIN_MALLOPT=0: glibc's defaults.
    if (!(getenv("MEME_DROPIN_MALLOPT") && atoi(getenv("MEME_DROPIN_MALLOPT")) == 0)) {
        mallopt(M_TRIM_THRESHOLD, 1 << 30);
        mallopt(M_TOP_PAD, 64 << 20);
        mallopt(M_MMAP_THRESHOLD, 32 << 20);
    }
    // The output step writes one SAM record per fputs() into stdio's 4 KB buffer: 320 000 write() calls per 4 M reads.  A buffer of
    // 16 MB on the aligner's output stream (stdout unless -o names a file) before anything is written to it.  MEME_DROPIN_OUTBUF=0: off.
    if (!(getenv("MEME_DROPIN_OUTBUF") && atoi(getenv("MEME_DROPIN_OUTBUF")) == 0)) setvbuf(stdout, nullptr, _IOFBF, (size_t)16 << 20);
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

}  // namespace dropin

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
    // binding has to provide the very same bytes for the SAM output to be identical.  Only the cross-check mode calls that function on the
    // host (MEME_DROPIN_EXT=0); with the seed filter on the device nothing reads the array and it is not built.  At GRCh38 size that is 0.38 s
    // less host work in this function (0.49 -> 0.11 s) but no wall time: the function then waits that much longer for the index stream to
    // HBM (0.63 -> 1.08 s), which is the critical path of the start-up (profiles/r04_e2e_dropin_c/_d.stderr).
    ext_mode_decide(aux->opt);
    const int64_t l_pac = aux->fmi->idx->bns->l_pac;
    const int64_t ll_pac = ext_mode() == 0 ? (l_pac * 2 + 3) / 4 * 4 : 0;
    w.rc_pac = ll_pac ? (uint8_t*)malloc((size_t)(ll_pac / 4)) : nullptr;                    // (process() frees it, src/fastmap.cpp:1109)
    if (ll_pac && !w.rc_pac) { fprintf(stderr, "[meme-dropin] out of memory\n"); exit(1); }
    const uint8_t* pac = aux->fmi->idx->pac;
    team_for(ll_pac / 4, cig_threads(), [&](int64_t k0, int64_t k1, int) {
    for (int64_t k = k0; k < k1; ++k) {
        uint8_t b = 0;
        for (int j = 0; j < 4; ++j) {
            const int64_t p = 4 * k + j;
            int c = 0;
            if (p < l_pac) c = pac[p >> 2] >> ((~p & 3) << 1) & 3;
            else if (p < 2 * l_pac) { const int64_t q = 2 * l_pac - 1 - p; c = 3 - (pac[q >> 2] >> ((~q & 3) << 1) & 3); }
            b = (uint8_t)(b | (c << (j << 1)));                 // base j of the byte in bits 2j, 2j + 1: what the reference's "BitReverseTable256"
        }                                                       // (src/LearnedIndex_seeding.h:129-137: it reverses the 2-bit groups) makes of _set_pac's byte
        w.rc_pac[k] = b;
    }
    });
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
    (void)ext_mode();                                                            // (a bad MEME_DROPIN_EXT stops the run before the index loads)
    if (g_early_prefix && strcmp(g_early_prefix, prefix) != 0) { fprintf(stderr, "[meme-dropin] MEME_INDEX_PREFIX changed after start-up\n"); exit(1); }
    init_devices(prefix, (int64_t)nreads);                                       // (returns at once when the early load below has done it)
    if (g_early) { g_early->join(); delete g_early; g_early = nullptr; }
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
namespace dropin {

Chunk g_chunks[2];
Chunk* g_cur_chunk = &g_chunks[0];
int64_t g_cur_chunk_seq = -1;

const bntseq_t* g_bns = nullptr;               // of the run (set by mem_process_seqs)
std::vector<meme_contig> g_contigs;
// MEME_DROPIN_EXT: "device" (default) = chaining, the seed filter (mem_flt_chained_seeds) AND seed extension on the GPU, the host receives
// alignment records; "0" = the reference's own per-batch functions on chains brought back from the device (the cross-check).
int g_ext_mode_override = -1;           // set once by ext_mode_decide(), before the first chunk
// The device's seed filter (mem_flt_chained_seeds under -W) packs scores in 12 bits and penalties in a signed byte (csrc/meme_kswv.hip:
// 199 x match < 4096, match / mismatch <= 127).  A run outside those limits keeps the reference's own per-batch functions on chains
// brought back from the device (the MEME_DROPIN_EXT=0 arrangement) instead of stopping in the middle of its first chunk.
void ext_mode_decide(const mem_opt_t* opt) {
    if (ext_mode() != 2 || opt->min_chain_weight <= 0) return;
    if ((int64_t)199 * opt->a < 4096 && opt->a <= 127 && opt->b <= 127) return;
    fprintf(stderr, "[meme-dropin] -W with match score %d / mismatch penalty %d is beyond the device seed filter's limits: the extension stage of this run "
            "uses the reference's per-batch functions (as MEME_DROPIN_EXT=0)\n", opt->a, opt->b);
    g_ext_mode_override = 0;
}
int ext_mode() {
    if (g_ext_mode_override >= 0) return g_ext_mode_override;
    static const int v = [] {
        const char* e = getenv("MEME_DROPIN_EXT");
        if (!e || !strcmp(e, "device") || !strcmp(e, "2")) return 2;
        if (!strcmp(e, "0")) return 0;
        fprintf(stderr, "[meme-dropin] MEME_DROPIN_EXT=%s: the values are device and 0 (the host-side extension stage is gone: the device runs the seed filter too)\n", e);
        exit(1);
    }();
    return v;
}
std::atomic<bool> g_ext_on_device{false};           // decided per run in mem_process_seqs (needs opt)
std::atomic<double> g_t_ext_dev{0}, g_t_ext_chain_ms{0}, g_t_ext_ms{0}, g_t_ext_bsw_ms{0};
std::atomic<int64_t> g_n_ext_pairs{0}, g_n_ext_retried{0}, g_n_ext_regs{0}, g_n_ext_live{0}, g_n_ext_tier2{0}, g_n_flt_jobs{0}, g_n_flt_dropped{0};
bool chain_on_device() { static const bool v = !(getenv("MEME_DROPIN_CHAIN") && atoi(getenv("MEME_DROPIN_CHAIN")) == 0); return v; }
bool chain_check() { static const bool v = getenv("MEME_DROPIN_CHAIN_CHECK") != nullptr; return v; }
// MEME_DROPIN_CHAIN_DUMP=<file> (fixture generation, tests/golden/make_chain_golden.py): every read's seeds and the chains the
// REFERENCE's host functions make of them, as text
FILE* chain_dump() { static FILE* f = getenv("MEME_DROPIN_CHAIN_DUMP") ? fopen(getenv("MEME_DROPIN_CHAIN_DUMP"), "w") : nullptr; return f; }
std::mutex g_dump_mu;
std::atomic<int64_t> g_n_chain_fallback{0}, g_n_chain_reads{0};

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
    meme_ctx* const ctx = P.ctx;
    if (P.count + 1 > P.off_cap) { meme_host_free(P.off); P.off_cap = P.count + P.count / 4 + 64; if (!(P.off = (int64_t*)meme_host_alloc(P.off_cap * 8))) die("meme_host_alloc"); }
    int64_t bytes = 0;
    for (int64_t i = 0; i < P.count; ++i) { P.off[i] = bytes; bytes += seqs[P.first + i].l_seq; }
    P.off[P.count] = bytes;
    if (bytes + 16 > P.flat_cap) { meme_host_free(P.flat); P.flat_cap = bytes + bytes / 4 + 4096; if (!(P.flat = (uint8_t*)meme_host_alloc(P.flat_cap))) die("meme_host_alloc"); }
    // Base codes in place, as the reference leaves them for the later stages (src/bwamem.cpp:1277-1279: mem_sort_dedup_patch in
    // worker_aln is the first to read them).  With the extension stage on the device the GPU does not wait for that: the letters are
    // gathered as they are and converted on the device (meme_seed_batch_resident_ascii), and the in-place conversion runs on a helper
    // team beside the backend calls.
    // (a few dozen helper threads: an OpenMP team of all 256 host threads takes longer to start than the loop runs, and keeps spinning
    // into the worker phases that follow)
    const bool raw = g_ext_on_device;
    TeamLabel lbl_("seed: gather reads");
    team_for(P.count, cig_threads(), [&](int64_t i0, int64_t i1, int) {
        for (int64_t i = i0; i < i1; ++i) {
            bseq1_t& s = seqs[P.first + i];
            uint8_t* dst = P.flat + P.off[i];
            if (raw) memcpy(dst, s.seq, (size_t)s.l_seq);
            else { letters_to_codes(s.seq, s.l_seq); memcpy(dst, s.seq, (size_t)s.l_seq); }
        }
    });
    std::thread codes;
    if (raw) codes = std::thread([&P, seqs] {
        TeamLabel l2("seed: letters to codes in place");
        team_for(P.count, cig_threads() / 2 > 0 ? cig_threads() / 2 : 1, [&](int64_t i0, int64_t i1, int) {
            for (int64_t i = i0; i < i1; ++i) {
                bseq1_t& s = seqs[P.first + i];
                letters_to_codes(s.seq, s.l_seq);
            }
        });
    });
    struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join_codes{codes};
    const meme_seed_opt so = seed_opt_of(opt);
    memset(&P.chains, 0, sizeof(P.chains));
    P.has_ext = false;
    const double ts0 = now_s();
    P.reads_on_ctx = true;
    if (!g_ext_on_device) {
        if (meme_seed_batch_host(ctx, P.flat, P.off, P.count, &so, &P.res)) die("meme_seed_batch_host");
        g_t_seed_call = g_t_seed_call + (now_s() - ts0);
    }
    if (g_ext_on_device && P.count > 0) {                // seeds stay in HBM; chains + extension where the seeds lie: only alignment records come back
        memset(&P.res, 0, sizeof(P.res));
        meme_chain_opt co;
        co.w = opt->w; co.max_chain_gap = opt->max_chain_gap; co.max_occ = opt->max_occ; co.min_seed_len = opt->min_seed_len;
        co.min_chain_weight = opt->min_chain_weight; co.max_chain_extend = opt->max_chain_extend;
        co.mask_level = opt->mask_level; co.drop_ratio = opt->drop_ratio; co.l_pac = g_bns->l_pac;
        meme_ext_opt eo;
        eo.a = opt->a; eo.b = opt->b; eo.o_del = opt->o_del; eo.e_del = opt->e_del; eo.o_ins = opt->o_ins; eo.e_ins = opt->e_ins;
        eo.pen_clip5 = opt->pen_clip5; eo.pen_clip3 = opt->pen_clip3; eo.w = opt->w; eo.zdrop = opt->zdrop;
        // The whole part in one go -- or, when the backend refuses it for want of memory (MEME_E_CAPACITY: a stage's scratch beside the
        // resident index), in halves, quarters, ...: reads are independent, the pieces' records are merged on the host.  Only a piece
        // of one 512-read batch that still does not fit stops the run.
        auto run_piece = [&](int64_t first, int64_t count, meme_ext_host_result* R) -> int {
            std::vector<int64_t> off;
            const int64_t* poff = P.off + first;
            if (first > 0) { off.resize((size_t)count + 1); for (int64_t i = 0; i <= count; ++i) off[(size_t)i] = P.off[first + i] - P.off[first]; poff = off.data(); }
            const double tc0 = now_s();
            int rc = meme_seed_batch_resident_ascii(ctx, P.flat + P.off[first], poff, count, &so, nullptr, nullptr);
            g_t_seed_call = g_t_seed_call + (now_s() - tc0);
            // (only the records mem_kernel2_core keeps cross to the host, src/bwamem.cpp:1680-1693; MEME_DROPIN_EXT_LIVE=0: all of them, as the reference's stage leaves them)
            static const int64_t live_only = getenv("MEME_DROPIN_EXT_LIVE") ? atoll(getenv("MEME_DROPIN_EXT_LIVE")) : 1;
            if (rc == MEME_OK) rc = meme_set_tuning(ctx, "ext_live_only", live_only);
            if (rc == MEME_OK) rc = meme_extend_last_batch_host(ctx, g_contigs.data(), (int32_t)g_contigs.size(), &co, &eo, R);
            return rc;
        };
        const double t0 = now_s();
        int rc = run_piece(0, P.count, &P.ext);
        if (rc == MEME_E_CAPACITY) {
            // (said twice, not per chunk: the first time with the backend's reason, and once more when it turns out to be the run's normal state --
            // a card with too little free HBM beside the index and the second slot's ctx runs every chunk this way, CIGAR stage on the host included)
            static std::atomic<int> n_pieces{0};
            const int seen = ++n_pieces;
            if (seen == 1)
                fprintf(stderr, "[meme-dropin] the backend cannot take %lld reads at once (%s): in pieces\n", (long long)P.count, meme_last_error());
            else if (seen == 4)
                fprintf(stderr, "[meme-dropin] chunks keep running in pieces (%d so far): the device has too little free memory beside the index for a whole -K chunk; "
                                "a smaller -K, or MEME_DROPIN_PREFETCH=0 (one seeding ctx per GPU instead of two), avoids the split\n", seen);
            P.own_regs.clear(); P.own_reg_off.assign(1, 0);
            meme_ext_host_result tot;
            memset(&tot, 0, sizeof(tot));
            int64_t piece = (P.count / 2 + BATCH_SIZE - 1) / BATCH_SIZE * BATCH_SIZE, done = 0;
            while (done < P.count) {
                const int64_t m = piece < P.count - done ? piece : P.count - done;
                meme_ext_host_result R;
                rc = run_piece(done, m, &R);
                if (rc == MEME_E_CAPACITY && m > BATCH_SIZE) { piece = (m / 2 + BATCH_SIZE - 1) / BATCH_SIZE * BATCH_SIZE; continue; }
                if (rc) die("chunk-level device stages");
                const int64_t base = (int64_t)P.own_regs.size();
                P.own_regs.insert(P.own_regs.end(), R.regs, R.regs + R.total_regs);
                for (int64_t i = 1; i <= m; ++i) P.own_reg_off.push_back(base + R.reg_off[i]);
                tot.total_regs += R.total_regs; tot.total_seeds += R.total_seeds; tot.total_chains += R.total_chains; tot.n_pairs += R.n_pairs; tot.n_retried += R.n_retried; tot.n_bsw_calls += R.n_bsw_calls;
                tot.n_tier2 += R.n_tier2; tot.chain_ms += R.chain_ms; tot.ext_ms += R.ext_ms; tot.bsw_ms += R.bsw_ms;
                tot.n_flt_jobs += R.n_flt_jobs; tot.n_flt_dropped += R.n_flt_dropped;
                done += m;
            }
            tot.nreads = P.count; tot.regs = P.own_regs.data(); tot.reg_off = P.own_reg_off.data();
            P.ext = tot;
            P.reads_on_ctx = false;                       // (the CIGAR stage names reads of the batch resident on the ctx: not for this part)
        } else if (rc) die("chunk-level device stages (meme_seed_batch_resident_ascii / meme_extend_last_batch_host)");
        // MEME_DROPIN_VERIFY: the same stages once more on the slot's verify ctx, the records compared byte for byte
        P.vfy = nullptr;
        if (verify_on() && P.reads_on_ctx && g_dev[(size_t)d].vfy[P.ctx == g_dev[(size_t)d].seed2 ? 1 : 0]) {
            meme_ctx* const v = g_dev[(size_t)d].vfy[P.ctx == g_dev[(size_t)d].seed2 ? 1 : 0];
            static const int64_t live_only = getenv("MEME_DROPIN_EXT_LIVE") ? atoll(getenv("MEME_DROPIN_EXT_LIVE")) : 1;
            meme_ext_host_result V;
            int vr = meme_seed_batch_resident_ascii(v, P.flat, P.off, P.count, &so, nullptr, nullptr);
            if (vr == MEME_OK) vr = meme_set_tuning(v, "ext_live_only", live_only);
            if (vr == MEME_OK) vr = meme_extend_last_batch_host(v, g_contigs.data(), (int32_t)g_contigs.size(), &co, &eo, &V);
            if (vr) die("MEME_DROPIN_VERIFY: the second run of the chunk-level device stages");
            const meme_ext_host_result& A = P.ext;
            if (A.total_regs != V.total_regs || A.total_seeds != V.total_seeds || A.total_chains != V.total_chains || A.n_pairs != V.n_pairs || A.n_retried != V.n_retried || A.n_tier2 != V.n_tier2) {
                char msg[256];
                snprintf(msg, sizeof(msg), "totals: records %lld / %lld, chained seeds %lld / %lld, chains %lld / %lld, extension jobs %lld / %lld, retried %lld / %lld", (long long)A.total_regs, (long long)V.total_regs,
                         (long long)A.total_seeds, (long long)V.total_seeds, (long long)A.total_chains, (long long)V.total_chains, (long long)A.n_pairs, (long long)V.n_pairs, (long long)A.n_retried, (long long)V.n_retried);
                verify_fail("seeding + chaining + extension", -1, msg);
            }
            for (int64_t i = 0; i <= P.count; ++i) if (A.reg_off[i] != V.reg_off[i]) verify_fail("seeding + chaining + extension", i, seqs[P.first + (i < P.count ? i : P.count - 1)].name);
            if (memcmp(A.regs, V.regs, (size_t)A.total_regs * sizeof(meme_alnreg)) != 0)
                for (int64_t i = 0; i < P.count; ++i)
                    if (memcmp(A.regs + A.reg_off[i], V.regs + A.reg_off[i], (size_t)(A.reg_off[i + 1] - A.reg_off[i]) * sizeof(meme_alnreg)) != 0) verify_fail("seeding + chaining + extension (alignment records of a read)", i, seqs[P.first + i].name);
            uint64_t h = verify_hash(A.reg_off, (size_t)(P.count + 1) * 8);
            h = verify_hash(A.regs, (size_t)A.total_regs * sizeof(meme_alnreg), h);
            verify_note(P.chunk_seq, "ext-records", d, h, A.total_regs);
            P.vfy = v;
        }
        // names and qualities of the slice beside its bases, for the SAM text kernel (the whole slice on the ctx, every read with qualities or none)
        P.sam_staged = false;
        if (sam_on_device() && P.reads_on_ctx) {
            int64_t nb = 0, with_q = 0;
            if (P.count + 1 > P.name_off_cap) { meme_host_free(P.name_off); P.name_off_cap = P.count + P.count / 4 + 64; if (!(P.name_off = (int64_t*)meme_host_alloc(P.name_off_cap * 8))) die("meme_host_alloc"); }
            for (int64_t i = 0; i < P.count; ++i) { P.name_off[i] = nb; nb += (int64_t)strlen(seqs[P.first + i].name); with_q += seqs[P.first + i].qual != nullptr; }
            P.name_off[P.count] = nb;
            if (with_q == 0 || with_q == P.count) {
                if (nb + 16 > P.names_cap) { meme_host_free(P.names); P.names_cap = nb + nb / 4 + 4096; if (!(P.names = (char*)meme_host_alloc(P.names_cap))) die("meme_host_alloc"); }
                if (with_q && bytes + 16 > P.quals_cap) { meme_host_free(P.quals); P.quals_cap = bytes + bytes / 4 + 4096; if (!(P.quals = (char*)meme_host_alloc(P.quals_cap))) die("meme_host_alloc"); }
                TeamLabel l3("seed: names + qualities for the SAM text");
                team_for(P.count, cig_threads(), [&](int64_t i0, int64_t i1, int) {
                    for (int64_t i = i0; i < i1; ++i) {
                        const bseq1_t& s = seqs[P.first + i];
                        memcpy(P.names + P.name_off[i], s.name, (size_t)(P.name_off[i + 1] - P.name_off[i]));
                        if (with_q) memcpy(P.quals + P.off[i], s.qual, (size_t)s.l_seq);
                    }
                });
                if (meme_sam_stage_text(ctx, P.names, P.name_off, with_q ? P.quals : nullptr)) die("meme_sam_stage_text");
                if (P.vfy && meme_sam_stage_text(P.vfy, P.names, P.name_off, with_q ? P.quals : nullptr)) die("meme_sam_stage_text");
                P.sam_staged = true;
            }
        }
        g_t_ext_dev = g_t_ext_dev + (now_s() - t0);
        g_t_ext_chain_ms = g_t_ext_chain_ms + P.ext.chain_ms; g_t_ext_ms = g_t_ext_ms + P.ext.ext_ms; g_t_ext_bsw_ms = g_t_ext_bsw_ms + P.ext.bsw_ms;
        g_n_ext_pairs += P.ext.n_pairs; g_n_ext_retried += P.ext.n_retried; g_n_ext_regs += P.ext.total_seeds; g_n_ext_live += P.ext.total_regs; g_n_ext_tier2 += P.ext.n_tier2;
        g_n_flt_jobs += P.ext.n_flt_jobs; g_n_flt_dropped += P.ext.n_flt_dropped;
        P.has_ext = true;
        return;
    }
    if (chain_on_device() && P.count > 0) {              // mem_chain_Learned + mem_chain_flt while the seeds are still in HBM
        meme_chain_opt co;
        co.w = opt->w; co.max_chain_gap = opt->max_chain_gap; co.max_occ = opt->max_occ; co.min_seed_len = opt->min_seed_len;
        co.min_chain_weight = opt->min_chain_weight; co.max_chain_extend = opt->max_chain_extend;
        co.mask_level = opt->mask_level; co.drop_ratio = opt->drop_ratio; co.l_pac = g_bns->l_pac;
        if (meme_chain_last_batch_host(ctx, g_contigs.data(), (int32_t)g_contigs.size(), &co, &P.chains)) die("meme_chain_last_batch_host");
    }
}

void seed_chunk(const mem_opt_t* opt, bseq1_t* seqs, int64_t n, int slot) {
    const double t0 = now_s();
    const int nd = (int)g_dev.size();
    Chunk& C = g_chunks[slot];
    C.seqs = seqs;
    C.n = n;
    if (C.part.size() != (size_t)nd) C.part.resize((size_t)nd);
    if (slot == 1)
        for (int d = 0; d < nd; ++d)
            if (!g_dev[(size_t)d].seed2) {                        // the second slot's ctxs: same index, buffers of their own
                Device& D = g_dev[(size_t)d];
                meme_index_arrays ia;
                if (meme_index_describe(D.seed, &ia)) die("meme_index_describe");
                int dev_id = 0;
                { int nreal = meme_device_count(); dev_id = nreal > 0 ? d % nreal : 0; }
                if (!(D.seed2 = meme_ctx_create(dev_id)) || meme_index_share(D.seed2, D.seed)) die("second seeding ctx");
                if (getenv("MEME_DROPIN_MAX_BATCH")) meme_set_tuning(D.seed2, "max_batch", atoll(getenv("MEME_DROPIN_MAX_BATCH")));
                if (getenv("MEME_DROPIN_SAM_MAX_BATCH")) meme_set_tuning(D.seed2, "sam_max_batch", atoll(getenv("MEME_DROPIN_SAM_MAX_BATCH")));
            }
    // consecutive 512-read batches of the chunk go to consecutive GPUs: contiguous ranges, batch-aligned
    const int64_t nb = (n + BATCH_SIZE - 1) / BATCH_SIZE;
    std::vector<std::thread> th;
    for (int d = 0; d < nd; ++d) {
        ChunkPart& P = C.part[(size_t)d];
        P.ctx = slot ? g_dev[(size_t)d].seed2 : g_dev[(size_t)d].seed;
        P.chunk_seq = C.seq; P.dev = d;
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
    if (verbose()) {
        timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
        fprintf(stderr, "[meme-dropin] chunk of %lld reads seeded on %d GPU(s) in %.3f s (this thread's CPU so far %.2f s)\n", (long long)n, nd, now_s() - t0, (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec);
    }
}

// ---- the next chunk ahead of its turn -------------------------------------------------------------------------------------------
// The aligner's pipeline (kt_pipeline, two threads for read / process / write) processes one chunk at a time, and the GPUs are idle for
// most of a chunk's SAM phase.  The binding's FASTQ reader assembles chunk k+1 as soon as chunk k has been handed out and submits it
// here; a helper thread takes the submitted chunks, in order, through the device stages (gather, seeding, chaining, seed filter,
// extension) on the other slot's ctxs, and mem_process_seqs finds its chunk's alignment records waiting.  Chunk s lives in slot s & 1:
// its device stages start when chunk s - 2 has left mem_process_seqs (whose CIGAR stage still names reads resident on that slot's ctxs),
// i.e. they run beside the whole of chunk s - 1's host phases.  The helper and its state live until the process ends (no destructor
// runs against a waiting thread).  MEME_DROPIN_PREFETCH=0: off.
struct Prefetcher {
    std::mutex m;
    std::condition_variable cv;
    struct Job { bseq1_t* seqs; int64_t n, seq; int state; };   // state: 1 queued, 2 running, 3 done
    std::deque<Job> jobs;                                       // in chunk order
    int64_t done[2] = {-1, -1};                                 // per slot: highest chunk number whose reads, seeds and staged text nobody needs any more
    mem_opt_t opt;                                              // the run's options BY VALUE (mem -p hands mem_process_seqs a stack-local copy, src/fastmap.cpp:790-828)
    bool has_opt = false;
    bool started = false;
    bool disabled = false;                                      // the chunks mem_process_seqs receives are not the arrays the reader handed out (mem -p): nothing runs ahead
};
Prefetcher* g_pf = new Prefetcher;
// Chunks are numbered where they are read (the binding's FASTQ reader calls prefetch_submit for every chunk, in order): chunk s lives in
// slot s & 1, whoever seeds it.  Chunks from any other reader are not numbered: they take the slots in turn, and nothing runs ahead.
std::mutex g_seq_mu;
int64_t g_seq_next = 0;
struct SeqTag { const bseq1_t* seqs; int64_t seq; };
SeqTag g_seq_tags[8] = {};
int g_next_slot = 0;                    // slot of the next un-numbered chunk
std::atomic<double> g_t_prefetched{0};
bool prefetch_on() { static const bool v = !(getenv("MEME_DROPIN_PREFETCH") && atoi(getenv("MEME_DROPIN_PREFETCH")) == 0); return v; }

void prefetch_submit(bseq1_t* seqs, int64_t n) {
    int64_t seq;
    {
        std::lock_guard<std::mutex> lk(g_seq_mu);
        seq = g_seq_next++; g_seq_tags[seq & 7] = {seqs, seq};
    }
    Prefetcher& F = *g_pf;
    std::lock_guard<std::mutex> lk(F.m);
    if (F.disabled) return;
    F.jobs.push_back({seqs, n, seq, 1});
    if (!F.started && prefetch_on()) {
        F.started = true;
        std::thread([] {
            pthread_setname_np(pthread_self(), "meme-prefetch");
            Prefetcher& F = *g_pf;
            for (;;) {
                bseq1_t* seqs = nullptr; int64_t n = 0, seq = 0;
                {
                    // the first queued chunk, once the run's options are known (the first mem_process_seqs call) and its slot is free
                    std::unique_lock<std::mutex> lk(F.m);
                    F.cv.wait(lk, [&] {
                        if (!F.has_opt || F.disabled || !g_ext_on_device || g_dev.empty()) return false;
                        for (Prefetcher::Job& J : F.jobs)
                            if (J.state == 1) { if (F.done[J.seq & 1] < J.seq - 2) return false; J.state = 2; seqs = J.seqs; n = J.n; seq = J.seq; return true; }
                        return false;
                    });
                }
                const double t0 = now_s();
                g_chunks[seq & 1].seq = seq;
                seed_chunk(&F.opt, seqs, n, (int)(seq & 1));
                g_t_prefetched = g_t_prefetched + (now_s() - t0);
                {
                    std::lock_guard<std::mutex> lk(F.m);
                    for (Prefetcher::Job& J : F.jobs) if (J.seq == seq) J.state = 3;
                }
                F.cv.notify_all();
            }
        }).detach();
    }
    F.cv.notify_all();
}

// true: chunk `seqs` has been through the device stages already (waits for the helper when it is at it); false: it is the caller's to seed
bool prefetch_take(bseq1_t* seqs, int64_t n, int* slot) {
    Prefetcher& F = *g_pf;
    std::unique_lock<std::mutex> lk(F.m);
    auto find = [&] { for (size_t i = 0; i < F.jobs.size(); ++i) if (F.jobs[i].seqs == seqs && F.jobs[i].n == n) return (int)i; return -1; };
    int i = find();
    if (i < 0) return false;
    if (F.jobs[(size_t)i].state == 1) { F.jobs.erase(F.jobs.begin() + i); return false; }
    F.cv.wait(lk, [&] { i = find(); return i >= 0 && F.jobs[(size_t)i].state == 3; });
    *slot = (int)(F.jobs[(size_t)i].seq & 1);
    F.jobs.erase(F.jobs.begin() + i);
    return true;
}
// chunk number `seq` has left mem_process_seqs -- and, when the device writes its SAM text, the output step has taken that text: its slot
// may be seeded again
void prefetch_processed(int64_t seq) {
    Prefetcher& F = *g_pf;
    { std::lock_guard<std::mutex> lk(F.m); if (seq > F.done[seq & 1]) F.done[seq & 1] = seq; }
    F.cv.notify_all();
}

// mem_process_seqs was handed an array the reader never gave out (mem -p: every chunk is split into two arrays of its own, each
// processed with a stack-local copy of the options, src/fastmap.cpp:790-828), or smart pairing is on: the helper must never touch
// such a run's chunks.  Queued chunks are dropped, a chunk the helper is at is waited for (its slot is then simply overwritten by
// the caller's own seed_chunk), later submissions are ignored.
void prefetch_disable() {
    Prefetcher& F = *g_pf;
    std::unique_lock<std::mutex> lk(F.m);
    if (F.disabled) return;
    F.disabled = true;
    F.cv.wait(lk, [&] { for (const Prefetcher::Job& J : F.jobs) if (J.state == 2) return false; return true; });
    F.jobs.clear();
}

int g_team = 1;                        // kt_for worker threads of the run (opt->n_threads)
worker_t* g_worker = nullptr;             // of the chunk being processed (alignment records, ref_string: the CIGAR stage reads them)
const mem_opt_t* g_opt = nullptr;
mem_chain_v* g_chunk_chain_ar = nullptr;   // w.chain_ar of the chunk being processed: every batch's chain_ar is a slice of it
uint64_t g_chunk_gen = 0;               // counts the chunks seeded

typedef void (*process_fn)(mem_opt_t*, int64_t, int, bseq1_t*, const mem_pestat_t*, worker_t&);

}  // namespace dropin

void mem_process_seqs(mem_opt_t* opt, int64_t n_processed, int n, bseq1_t* seqs, const mem_pestat_t* pes0, worker_t& w) {
    static const process_fn next = (process_fn)ref_sym(R_MEM_PROCESS_SEQS);
    const double t_enter = now_s();
    int64_t chunk_seq = -1;                                      // the chunk's number when it came from the binding's reader
    g_team = opt->n_threads > 0 ? opt->n_threads : 1;
    if (w.useLearned && !g_bns) {
        g_bns = w.fmi->idx->bns;
        for (int i = 0; i < g_bns->n_seqs; ++i) g_contigs.push_back({g_bns->anns[i].offset, g_bns->anns[i].len, g_bns->anns[i].is_alt});
    }
    if (w.useLearned) {
        g_ext_on_device = ext_mode() == 2;
        g_chunk_chain_ar = w.chain_ar;
        g_worker = &w;
        g_opt = opt;
        ktfor_calls() = 0;
        int slot = -1;
        {
            std::lock_guard<std::mutex> lk(g_seq_mu);
            for (const SeqTag& t : g_seq_tags) if (t.seqs == seqs && t.seq >= g_seq_next - 8 && g_seq_next > 0) { slot = (int)(t.seq & 1); chunk_seq = t.seq; }
            if (slot >= 0) for (SeqTag& t : g_seq_tags) if (t.seqs == seqs) t.seqs = nullptr;      // (the array is freed and its address may come back)
        }
        // Chunks run ahead of their turn only when mem_process_seqs receives exactly the arrays the reader handed out, with the run's
        // own options: not under mem -p (MEM_F_SMARTPE), and not for an array without a tag.
        if (slot < 0 || (opt->flag & MEM_F_SMARTPE)) { prefetch_disable(); slot = -1; chunk_seq = -1; }
        else {
            { std::lock_guard<std::mutex> lk(g_pf->m); if (!g_pf->has_opt) { g_pf->opt = *opt; g_pf->has_opt = true; } }
            g_pf->cv.notify_all();                               // (chunks submitted before the options were known may go ahead now)
        }
        if (slot < 0) { slot = g_next_slot; g_next_slot ^= 1; g_chunks[slot].seq = -1; seed_chunk(opt, seqs, n, slot); }      // not from our reader: nothing is ahead
        else if (!prefetch_take(seqs, n, &slot)) { g_chunks[slot].seq = chunk_seq; seed_chunk(opt, seqs, n, slot); }
        g_cur_chunk = &g_chunks[slot];
        g_cur_chunk_seq = chunk_seq;
        ++g_chunk_gen;
    }
    const double t_body = now_s();
    // ---- the read counter the chunk is processed with (round 6: the cause of the one differing SAM md5 of round 5, found by ThreadSanitizer) --------
    // The reference's pipeline adds a chunk's reads to aux->n_processed in its OUTPUT step (src/fastmap.cpp:845) and reads the counter in the
    // PROCESS step of the next chunk (:805, :818, :832) -- two threads, no ordering between them.  The value becomes w.n_processed and from there the
    // `id` of every read (src/bwamem.cpp:1844-1909), which seeds hash_64(id + i) -- the tie-breaker between equally good alignments
    // (mem_mark_primary_se :2010) and equally good pairs (mem_pair, src/bwamem_pair.cpp:412).  Nearly always the output step's thread gets there
    // first (it continues straight on, the other thread has to wake up), so the counter holds all earlier chunks; when it does not -- a thread
    // descheduled at the wrong moment, e.g. 64 workers under a 16-CPU quota -- the chunk is processed with the previous chunk's ids and reads with tied
    // placements come out differently: same number of lines, another md5.  The calls of this function ARE ordered (the pipeline runs one PROCESS
    // step at a time), so the binding counts for itself: every call gets the number of reads all earlier calls were given, which is what the
    // reference computes whenever its two threads run in their usual order (-p: the two calls of a chunk get S and S + n_sep[0], as there).
    // MEME_DROPIN_NPROC=ref: the argument as the pipeline read it.
    static int64_t s_before = 0;
    static int64_t n_stale = 0;
    static const bool nproc_ref = getenv("MEME_DROPIN_NPROC") && !strcmp(getenv("MEME_DROPIN_NPROC"), "ref");
    const int64_t n_det = s_before;
    s_before += n;
    if (n_det != n_processed) {
        ++n_stale;
        if (n_stale <= 3 || verbose())
            fprintf(stderr, "[meme-dropin] note: the pipeline's read counter says %lld where %lld reads have been handed to mem_process_seqs (the reference's unordered update, src/fastmap.cpp:832 / 845; "
                    "%lld chunk(s) so far): %s\n", (long long)n_processed, (long long)n_det, (long long)n_stale, nproc_ref ? "used as given (MEME_DROPIN_NPROC=ref)" : "the ordered count is used");
    }
    next(opt, nproc_ref ? n_processed : n_det, n, seqs, pes0, w);
    g_chunk.seqs = nullptr;
    if (chunk_seq >= 0 && !sam_release_deferred(chunk_seq)) prefetch_processed(chunk_seq);      // (a chunk with SAM text to format is released by the output step)
    if (verbose()) fprintf(stderr, "[meme-dropin] mem_process_seqs of this chunk: %.3f s until its device records were there, %.3f s in the reference's body\n", t_body - t_enter, now_s() - t_body);
    if (verbose() && (double)g_t_prefetched > 0) fprintf(stderr, "[meme-dropin] device stages run ahead of their chunk's turn (beside the previous chunk's SAM phase): %.3f s so far\n", (double)g_t_prefetched);
    if (verbose())
        fprintf(stderr, "[meme-dropin] totals: chunk-level device stages (gather + seeding + chaining + extension) %.3f s for %lld reads, of which the seeding calls %.3f s; bsw %lld calls, %lld pairs "
                "(copy-in thread-seconds %.3f, backend calls %.3f s of which kernels %.3f s)\n",
                (double)g_t_seed, (long long)g_n_seed_reads, (double)g_t_seed_call, (long long)g_n_bsw_calls, (long long)g_n_bsw_pairs,
                (double)g_t_bsw_gather, (double)g_t_bsw_call, (double)g_t_bsw_kernel);
    if (verbose() && g_ext_on_device)
        fprintf(stderr, "[meme-dropin] chaining + extension on the device: %.3f s in the backend calls so far (HIP events: chaining %.3f s, extension stage %.3f s of "
                "which banded SW %.3f s); %lld alignment records, %lld extension jobs (%lld of them again with the doubled band), %lld reads chained by the "
                "wavefront-per-read tier, 0 reads chained on the host; %lld records handed to the host\n", (double)g_t_ext_dev, (double)g_t_ext_chain_ms * 1e-3, (double)g_t_ext_ms * 1e-3,
                (double)g_t_ext_bsw_ms * 1e-3, (long long)g_n_ext_regs, (long long)g_n_ext_pairs, (long long)g_n_ext_retried, (long long)g_n_ext_tier2, (long long)g_n_ext_live);
    if (verbose() && g_ext_on_device && g_n_flt_jobs > 0)
        fprintf(stderr, "[meme-dropin] seed filter (mem_flt_chained_seeds) on the device: %lld alignments, %lld chained seeds removed\n", (long long)g_n_flt_jobs, (long long)g_n_flt_dropped);
    if (verbose() && !g_ext_on_device && chain_on_device())
        fprintf(stderr, "[meme-dropin] chaining on the device: %lld of %lld reads so far were chained on the host instead\n",
                (long long)g_n_chain_fallback, (long long)g_n_chain_reads);
    if (verbose()) meme_dropin_report_matesw();
    if (verbose()) meme_dropin_report_cigar();
    if (verbose()) meme_dropin_report_mate();
    if (verbose()) meme_dropin_report_sam();
}

namespace dropin {

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

}  // namespace dropin

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
