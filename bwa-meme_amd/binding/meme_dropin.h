// Shared declarations of the binding's translation units (bwa-meme_amd/binding/meme_dropin*.cpp).  Written against the *reference's own
// headers*; see meme_dropin.cpp for what the binding is and how it is linked.
//   meme_dropin.cpp       devices, index to HBM (memoryAllocLearned), one seeding / chaining / extension call per chunk (mem_process_seqs),
//                         mem_kernel1_core_Learned
//   meme_dropin_ext.cpp   the extension stage on the host side (MEME_DROPIN_EXT=host), mem_chain2aln_across_reads_V2, the three
//                         BandedPairWiseSW entry points with the group-commit combiner
//   meme_dropin_sam.cpp   the SAM phase: kt_for interposer, CIGAR table (ksw_global2), opt-in mate-rescue table (worker_sam's third step)
//   meme_dropin_io.cpp    FASTQ reader (bseq_read_orig)
#ifndef MEME_DROPIN_H
#define MEME_DROPIN_H
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
#include <deque>
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

namespace dropin {

// The reference's own functions behind the ones the binding interposes: ONE table (meme_dropin.cpp), every entry looked up when the binding
// is loaded, one message naming everything that is missing -- a reference built with another compiler or other signatures stops at start-up,
// not in the middle of a run.  (C++ functions are found by their mangled names: the Itanium ABI spelling of the signatures in the reference's headers.)
enum RefSym { R_MEM_PROCESS_SEQS, R_CHAIN2ALN_V2, R_BSEQ_READ_ORIG, R_KT_PIPELINE, R_MEM_SAM_PE_BATCH, R_BWA_GEN_CIGAR2, R_MEM_ALN2SAM, R_MEM_PESTAT, R_KT_FOR, R_N_SYMS };
void* ref_sym(RefSym which);

double now_s();
void note_thread_cpu(const char* role);          // (verbose accounting: a binding thread that ends says how much CPU it used)
[[noreturn]] void die(const char* what);
bool verbose();

struct Device {
    meme_ctx* seed = nullptr;     // owns (device 0) or holds a replica of the index
    meme_ctx* seed2 = nullptr;    // shares seed's index: chunks alternate between the two, so that the next chunk's device stages can run
                                  // while this chunk's reads, seeds and alignment records are still in use (created with the first prefetch)
    meme_ctx* bsw = nullptr;      // second ctx of the GPU: BandedPairWiseSW calls (host extension stage), mate rescue
    meme_ctx* vfy_bsw = nullptr;  // MEME_DROPIN_VERIFY: the second run of mate rescue (its pre-pass runs beside the CIGAR pre-pass, which verifies on vfy[slot]: a ctx is one thread's)
    meme_ctx* vfy[2] = {nullptr, nullptr};   // MEME_DROPIN_VERIFY: per chunk slot a ctx with buffers (and a history) of its own on which every device stage runs once more
};
// MEME_DROPIN_VERIFY=1 (round 6, the determinism question): every device stage of every chunk -- seeding + chaining + extension records, the CIGAR
// table, the mate-rescue table, the SAM text -- runs a second time on another ctx of the same GPU (same index, other workspaces with another
// history: what a kernel reads without having written it differs between the two) and the two outputs are compared byte for byte in the aligner;
// a difference stops the run with the first differing item.  One line per chunk carries a 64-bit hash of each stage's output, so that two RUNS
// can be compared stage by stage as well.
bool verify_on();
uint64_t verify_hash(const void* p, size_t bytes, uint64_t h = 0x9e3779b97f4a7c15ull);
void verify_note(int64_t chunk_seq, const char* stage, int dev, uint64_t hash, int64_t items);   // one line on stderr: "[meme-dropin] verify chunk <seq> <stage> dev <d>: <items> items, hash <h>, identical on two ctxs"
[[noreturn]] void verify_fail(const char* stage, int64_t item, const char* what);
// (reached through an accessor: the early-start thread may run before the dynamic initialisers of meme_dropin.cpp)
std::vector<Device>& device_slots();
#define g_dev (dropin::device_slots())
extern std::atomic<double> g_t_seed, g_t_bsw_gather, g_t_bsw_call, g_t_bsw_kernel;
extern std::atomic<int64_t> g_n_bsw_calls, g_n_bsw_pairs, g_n_seed_reads;

struct ChunkPart {                     // the slice of a chunk one GPU seeded
    int64_t first = 0, count = 0;
    meme_ctx* ctx = nullptr;                             // the ctx that holds the slice's reads (the CIGAR stage names them)
    meme_ctx* vfy = nullptr;                             // MEME_DROPIN_VERIFY: the ctx the slice's stages ran on once more (holds the same reads, seeds and staged text)
    int64_t chunk_seq = -1; int dev = 0;                 // for the verify lines
    meme_seed_host_result res;
    meme_chain_host_result chains;                       // valid when g_chain_on_device (host extension stage)
    meme_ext_host_result ext;                            // valid in device-extension mode: alignment records of the part's reads
    bool has_ext = false;
    // a part the backend could only take in pieces (MEME_E_CAPACITY): the pieces' records, merged; the ctx then holds the LAST piece's reads only
    std::vector<meme_alnreg> own_regs; std::vector<int64_t> own_reg_off; bool reads_on_ctx = true;
    uint8_t* flat = nullptr; int64_t flat_cap = 0;       // pinned staging (grow-only)
    int64_t* off = nullptr; int64_t off_cap = 0;
    // SAM text on the device: names and qualities of the slice staged beside its bases (meme_sam_stage_text)
    bool sam_staged = false;
    char* names = nullptr; int64_t names_cap = 0; int64_t* name_off = nullptr; int64_t name_off_cap = 0; char* quals = nullptr; int64_t quals_cap = 0;
};
struct Chunk {
    const bseq1_t* seqs = nullptr;
    int64_t n = 0;
    int64_t seq = -1;                                    // the chunk's number when it came from the binding's reader (verify lines name it)
    std::vector<ChunkPart> part;
};
// SAM text on the device (meme_dropin_sam.cpp): the worker threads of a chunk note descriptors, the OUTPUT step of that chunk -- the pipeline's
// other thread, beside the next chunk's processing -- sends them through the device and writes the text from the ctxs' pinned result buffers.
struct SamPart { const char* text = nullptr; const int64_t* text_off = nullptr; int64_t first = 0, count = 0; };     // reads [first, first + count): text of read g at text_off[g - first]
struct SamText { std::vector<SamPart> part; };
SamText* sam_format_for_output(const bseq1_t* seqs);  // formats the chunk's noted records (null: none); the text is valid until sam_output_done
void sam_output_done(const bseq1_t* seqs);            // the chunk's text has been written: its slot may take the next chunk
bool sam_release_deferred(int64_t chunk_seq);         // true: the chunk has records noted, the output step will release its slot
void prefetch_processed(int64_t seq);
bool sam_on_device();                                  // MEME_DROPIN_SAM (default on; needs the binding's reader and output step)
extern int64_t g_cur_chunk_seq;                        // number of the chunk being processed when it came from the binding's reader, else -1
extern Chunk* g_cur_chunk;                     // the chunk mem_process_seqs is working on (one of two slots)
#define g_chunk (*dropin::g_cur_chunk)
// the device stages of a chunk ahead of its turn (started when the FASTQ reader hands the chunk out; meme_dropin.cpp)
void prefetch_submit(bseq1_t* seqs, int64_t n);      // (called by the binding's FASTQ reader: chunks get their sequence number there)
extern const bntseq_t* g_bns;                  // of the run (set by mem_process_seqs)
extern std::vector<meme_contig> g_contigs;
int ext_mode();                                // MEME_DROPIN_EXT: 2 device (default), 0 the reference's per-batch functions (the cross-check)
void ext_mode_decide(const mem_opt_t* opt);    // options beyond the device seed filter's limits: mode 0 for the run
bool prefetch_on();                            // MEME_DROPIN_PREFETCH: chunks go through the device stages ahead of their turn
extern std::atomic<bool> g_ext_on_device;           // (written by every mem_process_seqs call with the same value, read by the helper that runs chunks ahead)
extern int g_team;                             // kt_for worker threads of the run (opt->n_threads)
extern worker_t* g_worker;                     // of the chunk being processed
extern const mem_opt_t* g_opt;
std::atomic<int>& ktfor_calls();               // kt_for calls of the chunk so far (the third one is worker_sam)
extern mem_chain_v* g_chunk_chain_ar;          // w.chain_ar of the chunk being processed
extern uint64_t g_chunk_gen;                   // counts the chunks seeded
int cig_threads();                             // helper threads of the binding's own host loops
bool fast_out();                               // MEME_DROPIN_OUT: the binding's output step (writev from the records)

// The binding's own host loops (gathering reads, posing jobs, taking results, the output step) run on a team of helper threads that SLEEP
// between jobs.  They were OpenMP regions until round 5: libgomp's idle threads spin before they sleep, and under the CPU quota of the
// boxes this was measured on (16 CPUs' worth for a process) a spinning helper is paid for with the aligner's worker threads' time --
// a chunk's mem_pestat phase cost 0.3-0.4 CPU-seconds for 0.03 s of work.  Any thread may submit; the submitter works along.
//   team_run(nt, f)             f(t) for t in [0, nt), on up to nt threads at once; returns when all are done
//   team_for(n, nt, f)          f(lo, hi, t): [0, n) in nt contiguous ranges
void team_run(int nt, const std::function<void(int)>& f);
// (verbose accounting: the CPU seconds of the team's shares are summed per label; a submitter names what follows with `TeamLabel l("...")`)
extern thread_local const char* tl_team_label;
struct TeamLabel { const char* prev; explicit TeamLabel(const char* l) : prev(tl_team_label) { tl_team_label = l; } ~TeamLabel() { tl_team_label = prev; } };
template <class F> inline void team_for(int64_t n, int nt, F&& f) {
    if (nt > n) nt = n > 0 ? (int)n : 1;
    if (nt <= 1) { f((int64_t)0, n, 0); return; }
    team_run(nt, [&](int t) { f(n * t / nt, n * (t + 1) / nt, t); });
}
// ascending sort of v by `less` on nt threads: runs sorted side by side, merged pairwise
template <class T, class Less> inline void team_sort(std::vector<T>& v, int nt, Less less) {
    const int64_t n = (int64_t)v.size();
    if (n < 65536 || nt < 2) { std::sort(v.begin(), v.end(), less); return; }
    int runs = 1;
    while (runs * 2 <= nt && runs < 16) runs *= 2;
    std::vector<int64_t> cut((size_t)runs + 1);
    for (int r = 0; r <= runs; ++r) cut[(size_t)r] = n * r / runs;
    team_run(runs, [&](int r) { std::sort(v.begin() + cut[(size_t)r], v.begin() + cut[(size_t)r + 1], less); });
    for (int w = 1; w < runs; w *= 2)
        team_run(runs / (2 * w), [&](int k) { std::inplace_merge(v.begin() + cut[(size_t)(2 * w * k)], v.begin() + cut[(size_t)(2 * w * k + w)], v.begin() + cut[(size_t)(2 * w * k + 2 * w)], less); });
}

}  // namespace dropin

void meme_dropin_report_matesw();
void meme_dropin_report_cigar();
void meme_dropin_report_mate();
void meme_dropin_report_sam();
#endif
