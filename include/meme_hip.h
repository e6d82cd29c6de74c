/* meme_hip.h -- C ABI of the MI355X (gfx950) backend for BWA-MEME's learned-index seeding and
 * banded Smith-Waterman seed extension.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  Each entry point
 * names the reference interface it replaces (paths relative to the BWA-MEME source tree):
 *
 *   meme_index_load_host / _files   <- learned_index_load()            src/LearnedIndex_seeding.h:290
 *                                      + index part of memoryAllocLearned()  src/fastmap.cpp:422-617
 *   meme_seed_batch                 <- per-read loop of mem_kernel1_core_Learned()  src/bwamem.cpp:1249-1394
 *                                      = Learned_getSMEMsAllPosOneThread()   src/LearnedIndex_seeding.h:265
 *                                      + Learned_bwtSeedStrategyAllPosOneThread[_mem_tradeoff]() :241,:247
 *                                      (outputs: mem_tl records + hit positions, the inputs of the unchanged
 *                                       host consumer mem_chain_Learned(), src/bwamem.cpp:1122-1204)
 *   meme_chain_last_batch_host      <- mem_chain_Learned() + mem_chain_flt()          src/bwamem.cpp:1122-1204, 599-717
 *   meme_extend_last_batch_host     <- mem_chain2aln_across_reads_V2()              src/bwamem.cpp:2573-3497 (behind the chaining stage)
 *   meme_global_batch_host          <- ksw_global2() under bwa_gen_cigar2()         src/ksw.cpp:560-670, src/bwa.cpp:274-362
 *   meme_gen_cigar_batch_host       <- bwa_gen_cigar2() whole: CIGAR + NM + MD       src/bwa.cpp:274-362
 *   meme_sam_format_batch_host      <- mem_aln2sam() for a chunk's plain records     src/bwamem.cpp:2174-2312
 *   meme_matesw_batch_host          <- mem_sam_pe_batch_pre() + mem_sam_pe_batch()   src/bwamem_pair.cpp:660-716, 1060-1223, 719-818
 *   meme_bsw_batch                  <- BandedPairWiseSW::getScores8 / getScores16 /
 *                                      scalarBandedSWAWrapper                 src/bandedSWA.h:118-135,257-297
 *
 * Threading: a meme_ctx owns one HIP device, one stream and its workspaces; calls on one ctx are
 * serialised by the caller (one ctx per kt_for worker `tid`, or one submitter per phase).  Several
 * ctxs may share one index through meme_index_share().
 *
 * Errors: every function returns 0 on success or a negative MEME_E_* code; meme_last_error() gives
 * the message.  There is no CPU fallback: without a usable HIP device every call fails loudly.
 */
#ifndef MEME_HIP_H
#define MEME_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MEME_OK 0
#define MEME_E_HIP (-1)       /* HIP runtime error / no device */
#define MEME_E_ARG (-2)       /* bad argument */
#define MEME_E_IO (-3)        /* index file missing / malformed */
#define MEME_E_CAPACITY (-4)  /* caller-provided output buffer too small (sizes reported back) */
#define MEME_E_STATE (-5)     /* index not loaded etc. */

typedef struct meme_ctx meme_ctx;

/* = mem_tl, src/LearnedIndex_seeding.h:121-127 (24 bytes) */
typedef struct {
    int32_t start, end;       /* SMEM = read[start,end) */
    int32_t hitbeg, hitcount; /* hits[hitbeg .. hitbeg+hitcount) relative to the read's first hit */
    uint64_t cache_refpos;    /* text position of the first hit */
} meme_mem_tl;

/* = SeqPair, src/bandedSWA.h:90-99 (56 bytes) */
typedef struct {
    int32_t idr, idq, id;
    int32_t len1, len2;
    int32_t h0;
    int32_t seqid, regid;
    int32_t score, tle, gtle, qle;
    int32_t gscore, max_off;
} meme_seqpair;

/* seeding options = the mem_opt_t fields the path reads (src/bwamem.cpp:126-162, 1348, 1358-1394) */
typedef struct {
    int32_t min_seed_len;   /* opt->min_seed_len, 19 */
    int32_t split_len;      /* (int)(min_seed_len * split_factor + .499), 28 */
    int32_t split_width;    /* opt->split_width, 10 */
    int32_t max_mem_intv;   /* opt->max_mem_intv, 20; 0 disables the third round */
    int32_t rounds;         /* 1: first round only (= ..._step1only), 2: + re-seeding, 3: + third round */
    int32_t hits_per_smem;  /* 0: materialise every hit (reference behaviour); k>0: only the first k per SMEM */
} meme_seed_opt;

/* BandedPairWiseSW constructor arguments (src/bandedSWA.h:118-122); mat = bwa_fill_scmat(a, b) */
typedef struct {
    int32_t o_del, e_del, o_ins, e_ins, zdrop, end_bonus, a, b;
} meme_bsw_opt;

/* ---- context ------------------------------------------------------------------------------------ */
int meme_device_count(void);
meme_ctx* meme_ctx_create(int device);
void meme_ctx_destroy(meme_ctx* ctx);
const char* meme_last_error(void);
int meme_ctx_sync(meme_ctx* ctx);
void* meme_ctx_stream(meme_ctx* ctx);          /* the hipStream_t every kernel of this ctx is launched on */

/* ---- index staging ------------------------------------------------------------------------------
 * Inputs are the reference's on-disk images: pos_packed = 5 B per SA slot,
 * text0123 = 1 B per base fwd+rc, L1/L2 = 24-B P-RMI records.  The HBM layout is private:
 *   sa_ent[n]   16 B {u64 key (32 bases, first base in the top bits, T-filled), u64 text position}
 *   pac64[]     2-bit text, 32 bases per u64, first base in the top bits
 *   l2[], l1[]  P-RMI records padded to 32 bytes (a lookup never straddles a 128-byte line)
 *   plcp[n]     1 B per text position: LCP of that suffix with its nearer suffix-array neighbour, capped at 255 (derived on the device)
 * Keys are generated on the device (replaces the OpenMP loop of src/fastmap.cpp:549-613; no inverse suffix array). */
int meme_index_load_host(meme_ctx* ctx, const uint8_t* pos_packed, int64_t sa_num,
                         const uint8_t* text0123, const void* l1_params, int64_t l1_bytes,
                         const void* l2_params, int64_t l2_bytes);
int meme_index_load_files(meme_ctx* ctx, const char* prefix);
/* Device-resident variant for multi-GPU start-up with one process per GPU: the caller owns the arrays (e.g. the raw
 * images received through an RCCL broadcast and staged with the meme_stage_* calls) and they must outlive the ctx. */
typedef struct {
    int64_t sa_num;
    void* d_sa_ent;            /* sa_num * 16 B */
    void* d_pac64;             /* meme_index_pac64_words(sa_num) * 8 B */
    void* d_l2; int64_t l2_records;   /* 32-byte records (meme_stage_rmi32) */
    void* d_l1; int64_t l1_records;
} meme_index_arrays;
int64_t meme_index_pac64_words(int64_t sa_num);
int64_t meme_index_pos5_bytes(int64_t sa_num);                      /* 5 * sa_num + padding */
int meme_index_attach(meme_ctx* ctx, const meme_index_arrays* arrays);
int meme_index_describe(meme_ctx* ctx, meme_index_arrays* out);     /* device pointers of a loaded index */
int meme_index_share(meme_ctx* ctx, meme_ctx* owner);               /* second ctx on the same device */
/* Multi-GPU: copy src's staged index to dst (a ctx on another device) device-to-device over xGMI.  The counterpart,
 * for a host that drives all GPUs from one process (the reference's kt_for threads, src/kthread.cpp:79-114), of the
 * RCCL broadcast a one-process-per-GPU launcher uses.  Same device: behaves like meme_index_share. */
int meme_index_replicate(meme_ctx* dst, meme_ctx* src);
/* Pinned host memory for staging buffers handed to the batch calls (their copies then run as asynchronous DMA). */
void* meme_host_alloc(int64_t bytes);
void meme_host_free(void* p);
/* staging kernels usable on caller-owned device buffers (used by the multi-GPU path of bench.py) */
int meme_stage_pack_text(meme_ctx* ctx, const uint8_t* d_text0123, int64_t sa_num, void* d_pac64);
int meme_stage_pos5_from_sa(meme_ctx* ctx, const uint64_t* d_sa, int64_t sa_num, void* d_pos5);   /* u64 SA -> 5-byte image */
int meme_stage_build_entries(meme_ctx* ctx, const uint8_t* d_pos_packed, int64_t sa_num,
                             const void* d_pac64, void* d_sa_ent);
int meme_stage_entries_from_sa(meme_ctx* ctx, const uint64_t* d_sa, int64_t sa_num, const void* d_pac64,
                               void* d_sa_ent);
int meme_stage_rmi32(meme_ctx* ctx, const void* d_rmi24, int64_t records, void* d_rmi32);
/* Derived table of the staged index, built by every load / attach call (1 byte per suffix, by TEXT position): how many bases the suffix
 * shares with the closer of its two suffix-array neighbours, capped at 255.  It answers "how long a prefix of this text position occurs
 * at least twice" without a search, which is what the re-seeding round (src/LearnedIndex_seeding.cpp:923-947, 1897-2126) asks of the
 * middle of every unique SMEM. */
int meme_stage_build_plcp(meme_ctx* ctx, const void* d_sa_ent, int64_t sa_num, const void* d_pac64, void* d_plcp /* sa_num + 64 B */);

/* ---- suffix-array construction on the device (index building) -----------------------------------------
 * d_text0123: sa_num bytes, the forward strand followed by its reverse complement (codes 0..3, the reference's .0123 image);
 * d_sa: sa_num u64 out, the suffix array in the order `bwa-meme index` writes to .pos_packed (a suffix that ends sorts
 * before its continuations).  Radix sort by the first 32 bases + prefix doubling on the tied groups.  Workspace, allocated and
 * released inside the call: rank + suffix arrays of the padded text (16 bytes per suffix), ~13 GB for the sort of one 2^28-suffix
 * group, and ~90 bytes per suffix that still ties after 32 bases -- over 110 GB beside d_sa and the text at GRCh38 size. */
int meme_sa_build_device(meme_ctx* ctx, const uint8_t* d_text0123, int64_t sa_num, uint64_t* d_sa);

/* ---- P-RMI training on the device (index building) --------------------------------------------------------------------
 * Trains the model the aligner loads from <prefix>.suffixarray_uint64_L2_PARAMETERS / _L1_PARAMETERS (reference
 * src/LearnedIndex_seeding.cpp:74-122; trained offline by RMI/rmi_lib/src/train/two_layer.rs) from the staged entry array
 * (sorted 32-base keys): 2^bits second-layer records into d_l2_24 and the partial third layer (leaves with more than
 * partial_threshold keys, default 1000) into d_l1_24, both in the 24-byte file layout.  *l1_records receives the number
 * of third-layer records; when it exceeds l1_capacity nothing is written and MEME_E_CAPACITY is returned -- call once
 * with l1_capacity 0 to learn the size.  Same records, bit for bit, as the host trainer (bwa-meme_amd/host/meme_prmi.cpp). */
int meme_prmi_train_device(meme_ctx* ctx, const void* d_sa_ent, int64_t sa_num, int bits, int partial_threshold,
                           void* d_l2_24, void* d_l1_24, int64_t l1_capacity, int64_t* l1_records);

/* ---- seeding ------------------------------------------------------------------------------------
 * reads: concatenated base codes 0..3, >=4 = ambiguous (what mem_kernel1_core_Learned leaves in
 * bseq1_t.seq, src/bwamem.cpp:1277-1279); read_off[nreads+1].  A read longer than 500 bases
 * (LEARNED_MAX_READ_LEN; the reference exits, src/bwamem.cpp:1259-1262) fails the whole call with MEME_E_ARG.
 * Outputs, per read r: smems[smem_off[r] .. smem_off[r+1]) -- the reference's SMEMs of the read in an unspecified order (the caller
 * sorts them by (start, end), src/bwamem.cpp:1397, and records with equal keys carry equal hit lists, so the order is not observable;
 * the first round's come first, the re-seeding round's are appended by the kernels that find them) -- and
 * hits[hit_off[r] .. hit_off[r+1]) in ascending SA order per SMEM.          */
int meme_seed_batch(meme_ctx* ctx, const uint8_t* reads, const int64_t* read_off, int64_t nreads,
                    const meme_seed_opt* opt,
                    meme_mem_tl* smems, int64_t smem_capacity, int64_t* smem_off,
                    uint64_t* hits, int64_t hit_capacity, int64_t* hit_off,
                    int64_t* total_smems, int64_t* total_hits);

/* Same, results in pinned host buffers owned by the ctx (valid until its next seeding call): the form a chunk-level
 * binding uses -- one call per -K chunk of mem_process_seqs (src/bwamem.cpp:1920-1972), no capacity negotiation. */
typedef struct {
    const meme_mem_tl* smems;
    const int64_t* smem_off;     /* nreads+1 */
    const uint64_t* hits;
    const int64_t* hit_off;      /* nreads+1 */
    int64_t total_smems, total_hits;
} meme_seed_host_result;
int meme_seed_batch_host(meme_ctx* ctx, const uint8_t* reads, const int64_t* read_off, int64_t nreads,
                         const meme_seed_opt* opt, meme_seed_host_result* out);

/* Same, but SMEMs and hits stay in HBM for the calls that work on "the batch just seeded" (meme_chain_last_batch_host,
 * meme_extend_last_batch_host): what a binding uses that only wants chains or alignment records back.  Only the totals return. */
int meme_seed_batch_resident(meme_ctx* ctx, const uint8_t* reads, const int64_t* read_off, int64_t nreads,
                             const meme_seed_opt* opt, int64_t* total_smems, int64_t* total_hits);

/* Same, for reads as they stand in the FASTQ records (letters; bytes below 4 are taken as base codes): the conversion of
 * mem_kernel1_core_Learned (src/bwamem.cpp:1277-1279, c < 4 ? c : nst_nt4_table[c]) runs on the device. */
int meme_seed_batch_resident_ascii(meme_ctx* ctx, const uint8_t* reads, const int64_t* read_off, int64_t nreads,
                                   const meme_seed_opt* opt, int64_t* total_smems, int64_t* total_hits);

/* Optional: allocate the workspaces and pinned result buffers of a meme_seed_batch_host() (+ meme_chain_last_batch_host()) call of
 * this size ahead of time, e.g. on a helper thread while the index loads (pinned memory is slow to allocate). */
int meme_seed_reserve(meme_ctx* ctx, int64_t nreads, int64_t total_bases);

/* ---- chaining of the batch just seeded -------------------------------------------------------------------------------------
 * mem_chain_Learned() + mem_chain_flt() (reference src/bwamem.cpp:1122-1204, 599-717) for every read of the batch the last
 * meme_seed_batch_host() call on this ctx has seeded -- the SMEMs and hits are still in HBM.  Per read: the chains that survive
 * the filter, in the filter's output order, each with its seeds in chain order (seed score = seed length, as mem_chain_Learned
 * sets it).  tree_size[r] = number of chains before the filter (what the reference sizes chain_ar[r] with); frac_rep[r] as
 * mem_chain_Learned computes it.  Every read is chained on the device (reads with many chains, long chains, hundreds of hits to walk
 * or chains at equal positions by a wavefront-per-read tier that keeps the reference's B-tree, src/kbtree.h): fallback[r] is always 0
 * and n_fallback == 0 (the fields remain for ABI stability).  Results live in pinned buffers of the ctx until
 * the next call.  meme_contig = the fields of bntann1_t the stage needs (src/bntseq.h). */
typedef struct { int64_t offset; int32_t len; int32_t is_alt; } meme_contig;
typedef struct {
    int32_t w, max_chain_gap, max_occ, min_seed_len, min_chain_weight, max_chain_extend;   /* mem_opt_t fields of the same names */
    float mask_level, drop_ratio;
    int64_t l_pac;
} meme_chain_opt;
typedef struct { int64_t pos; int32_t rid, n_seeds, w, first; int16_t kept, is_alt; int32_t seed_beg /* in the read's seeds */; int32_t pad; } meme_chain;
typedef struct { int64_t rbeg; int32_t qbeg, len; } meme_chain_seed;
typedef struct {
    int64_t nreads;
    const int64_t* chain_off;        /* nreads+1 */
    const meme_chain* chains;
    const int64_t* seed_off;         /* nreads+1 */
    const meme_chain_seed* seeds;
    const int32_t* tree_size;
    const float* frac_rep;
    const uint8_t* fallback;
    int64_t total_chains, total_seeds, n_fallback;
    int64_t n_tier2;                 /* reads chained by the wavefront-per-read tier (repeats, equal positions) */
} meme_chain_host_result;
int meme_chain_last_batch_host(meme_ctx* ctx, const meme_contig* contigs, int32_t n_contigs, const meme_chain_opt* opt,
                               meme_chain_host_result* out);
/* The same for seeds the caller brings (host arrays laid out as meme_seed_batch_host returns them; read_len[r] = length of read r). */
int meme_chain_batch_host(meme_ctx* ctx, const meme_mem_tl* smems, const int64_t* smem_off, const uint64_t* hits, const int64_t* hit_off,
                          const int32_t* read_len, int64_t nreads, const meme_contig* contigs, int32_t n_contigs, const meme_chain_opt* opt,
                          meme_chain_host_result* out);

/* ---- seed extension of the batch just seeded ----------------------------------------------------------------------------------
 * mem_chain2aln_across_reads_V2() (reference src/bwamem.cpp:2573-3497) for every read of the batch the last meme_seed_batch_host() call
 * on this ctx has seeded, behind the chaining stage above and without leaving HBM in between: per chained seed one alignment record in
 * the reference's extension order (chain after chain, best seed of a chain first), left and right banded extensions with the band
 * doubled once where the reference doubles it (MAX_BAND_TRY 2), records of seeds an earlier alignment already covers marked qb = qe = -1
 * (:3389-3485).  regs[reg_off[r] .. reg_off[r+1]) = what the reference leaves in av_v[r] (n = m = the number of chained seeds).
 * meme_alnreg = mem_alnreg_t (src/bwamem.h:143-165, 112 bytes); its `c` (a chain pointer in the reference, dead after the stage)
 * holds the chain's index in the batch.  Between the two stages runs mem_flt_chained_seeds (src/bwamem.cpp:565-598) -- a no-op for reads
 * of at most 500 bases unless chain_opt->min_chain_weight (-W) is set; then, for reads of 22 x W bases and more, chained seeds whose
 * neighbourhood aligns below the bar (mem_seed_sw, :494-520) leave their chains and the others are extended in the order of those scores.
 * The alignment reads its windows as the aligner's call does: from the byte-reversed fwd + rc text through the plain _get_pac
 * (src/bwamem.cpp:1770, src/fastmap.cpp:440-457, src/bntseq.cpp:515-539) -- the bases of every aligned group of four in reverse order. */
typedef struct {
    int64_t rb, re; int32_t qb, qe; int32_t rid; int32_t pad0; uint64_t c;
    int32_t score, truesc, sub, alt_sc, csub, sub_n, w, seedcov, secondary, secondary_all, seedlen0;
    int32_t n_comp_is_alt;        /* n_comp:30, is_alt:2 */
    float frac_rep; int32_t pad1; uint64_t hash; int32_t flg; int32_t pad2;
} meme_alnreg;
typedef struct { int32_t a, b, o_del, e_del, o_ins, e_ins, pen_clip5, pen_clip3, w, zdrop; } meme_ext_opt;   /* mem_opt_t fields of the same names */
typedef struct {
    int64_t nreads;
    const int64_t* reg_off;          /* nreads+1 */
    const meme_alnreg* regs;
    int64_t total_regs, total_chains;   /* records handed over (= reg_off[nreads]), chains */
    int64_t n_pairs, n_retried, n_bsw_calls;   /* extension jobs run, of which with the doubled band; backend launches */
    int64_t n_tier2;                 /* reads chained by the wavefront-per-read tier */
    float chain_ms, ext_ms, bsw_ms;  /* HIP-event times: chaining kernels; the extension stage (incl. its host round trips); of which banded SW */
    int64_t n_flt_jobs, n_flt_dropped;   /* mem_flt_chained_seeds: alignments run (mem_seed_sw), chained seeds removed (0 / 0 where it is a no-op) */
    int64_t n_exact_prefix;          /* measurement (tuning "ext_census" = 1, else -1): first-attempt jobs whose query equals the first qlen target bases */
    int64_t census_band_cells;       /* ... the DP cells of their band-limited matrices (no trimming, no z-drop: an upper bound of the cells evaluated) */
    int64_t census_class[9];         /* ... jobs per LDS size class of the lane-per-pair kernel (query <= 30, 62, 94, 126, 158, 222, 318, 600 bases), [8]: longer queries */
    int64_t total_seeds;             /* chained seeds behind the seed filter = records made on the device (== total_regs unless tuning "ext_live_only" is set) */
    int64_t n_ext_seeds;             /* of them, seeds whose extension jobs ran (== total_seeds unless the stage ran in rounds, tuning "ext_rounds") */
} meme_ext_host_result;
int meme_extend_last_batch_host(meme_ctx* ctx, const meme_contig* contigs, int32_t n_contigs, const meme_chain_opt* chain_opt,
                                const meme_ext_opt* ext_opt, meme_ext_host_result* out);

/* Same with inputs and outputs resident in HBM (pointers valid until the next call on this ctx).
 * d_reads must be 4-byte aligned (any hipMalloc'ed pointer is); total_bases = read_off[nreads] = bytes in d_reads. */
typedef struct {
    const meme_mem_tl* d_smems;
    const int64_t* d_smem_off;   /* nreads+1 */
    const uint64_t* d_hits;
    const int64_t* d_hit_off;    /* nreads+1 */
    int64_t total_smems, total_hits;
    int64_t searches;            /* locate operations issued (deterministic per data set) */
} meme_seed_result;
int meme_seed_batch_device(meme_ctx* ctx, const uint8_t* d_reads, const int64_t* d_read_off,
                           int64_t nreads, int64_t total_bases, const meme_seed_opt* opt,
                           meme_seed_result* out);

/* ---- banded Smith-Waterman extension --------------------------------------------------------------
 * Semantics of scalarBandedSWA == ksw_extend2 for every pair (the int8 / int16 / scalar classes of the
 * reference compute the same function); results are written in place into
 * pairs[i].{score,tle,gtle,qle,gscore,max_off}.                                                      */
int meme_bsw_batch(meme_ctx* ctx, meme_seqpair* pairs, const uint8_t* ref_buf, int64_t ref_bytes,
                   const uint8_t* qer_buf, int64_t qer_bytes, int32_t npairs, int32_t w,
                   const meme_bsw_opt* opt);
int meme_bsw_batch_device(meme_ctx* ctx, meme_seqpair* d_pairs, const uint8_t* d_ref, const uint8_t* d_qer,
                          int32_t npairs, int32_t w, const meme_bsw_opt* opt);

/* ---- banded global alignment with traceback: the CIGAR kernel of the SAM phase -----------------------------------------------------
 * ksw_global2() (reference src/ksw.cpp:560-670) as bwa_gen_cigar2() (src/bwa.cpp:274-362) calls it from mem_reg2aln()
 * (src/bwamem.cpp:2314-2380): score and CIGAR of the global alignment of a read's query span [qb, qb+qlen) against the text span
 * [rb, rb+tlen) of the fwd+rc reference within band w.  Sequences are not shipped: `read` indexes the batch the last
 * meme_seed_batch_host() call staged on this ctx, the text is the 2-bit image in HBM; rev != 0 (alignments on the reverse strand,
 * rb >= l_pac) reverses both sequences as bwa_gen_cigar2 does.  CIGAR operations in the BAM encoding (len << 4 | op, op 0 M, 1 I, 2 D),
 * ties broken as the reference breaks them (M over E over F), so gaps sit where the reference puts them.  A job's band must reach the
 * last cell of its matrix, w >= |tlen - qlen| (every band bwa_gen_cigar2 computes does), and its query span must lie inside the read:
 * anything else fails the call with MEME_E_ARG.  MEME_E_CAPACITY (the batch's backtrack matrices do not fit beside the index): submit
 * fewer jobs per call. */
typedef struct { int64_t rb; int32_t read, qb, qlen, tlen, w, rev; } meme_gjob;
typedef struct { int32_t score, n_cigar; int64_t cigar_off; /* first operation in meme_gres_host::cigars */ } meme_gres;
typedef struct { int64_t njobs; const meme_gres* res; const uint32_t* cigars; int64_t total_ops; float kernel_ms; } meme_gres_host;
int meme_global_batch_host(meme_ctx* ctx, const meme_gjob* jobs, int64_t njobs, const meme_bsw_opt* opt /* o_del e_del o_ins e_ins a b */,
                           meme_gres_host* out);

/* ---- bwa_gen_cigar2 whole: CIGAR + NM + MD of a batch of its calls -----------------------------------------------------------------
 * What bwa_gen_cigar2() (reference src/bwa.cpp:274-362) returns for a call that it does not reject: the query is qlen bases of read
 * `read` of the batch resident on the ctx starting at qb (the `&query[qb]` mem_reg2aln passes, src/bwamem.cpp:2314-2380), the target the
 * span [rb, rb + tlen) of the fwd+rc text (what bns_get_seq unpacks, :286), w_ the band argument as mem_reg2aln passes it.  As in the
 * reference: both sequences reversed when rb >= l_pac (:288-293), the gap-free shortcut for equal lengths and w_ == 0 (:295-304: one M
 * operation, the score summed over the scoring matrix), else the band of :306-316 and ksw_global2 with traceback; then NM and the MD string
 * (:322-355, "ACGTN" / "TGCAN" by strand, a deletion at either end of the CIGAR left out).  Results in job order: operations packed in
 * `cigars`, MD strings NUL-terminated in `md`.  The hook a binding writes for bwa_gen_cigar2 copies n_cigar operations followed by
 * md_len + 1 bytes into one malloc'd block -- the layout the reference's caller expects (cigar, then MD behind it).  Jobs that the
 * reference's function rejects (empty spans, a target bridging the strands) fail the call with MEME_E_ARG; MEME_E_CAPACITY as above. */
typedef struct { int64_t rb; int32_t read, qb, qlen, tlen, w_, pad; } meme_cjob;
typedef struct { int32_t score, n_cigar, nm, md_len; int64_t cigar_off, md_off; } meme_cres;
typedef struct { int64_t njobs; const meme_cres* res; const uint32_t* cigars; int64_t total_ops; const char* md; int64_t md_bytes; float kernel_ms; } meme_cres_host;
int meme_gen_cigar_batch_host(meme_ctx* ctx, const meme_cjob* jobs, int64_t njobs, const meme_bsw_opt* opt /* o_del e_del o_ins e_ins a b */,
                              meme_cres_host* out);

/* ---- SAM text: mem_aln2sam() for the plain records of a chunk --------------------------------------------------------------------------
 * What mem_aln2sam() (reference src/bwamem.cpp:2174-2312, with add_cigar :2161-2172 and get_rlen :2402-2410) appends to its kstring for
 * one alignment record of a read, for the records that need nothing but the record itself, its mate's record and the read: the only
 * record of its read (n == 1: no SA tag), no read comment (-C), no pa tag (alt_sc == 0), no XR tag (-V).  Flags, coordinates swapped in
 * from the mate for an unmapped end, hard / soft clipping by `which`, template length from both CIGARs, SEQ / QUAL reversed on the reverse
 * strand, NM, MD, MC, AS, XS, RG, XA as in the reference, byte for byte.  The read's bases are those of the batch resident on the ctx
 * (codes: a base prints as "ACGTN"[code]); names and qualities are staged beside them once per batch (meme_sam_stage_text).
 * A record names its variable-length parts by offsets into `blob`: its n_cigar operations with the MD string right behind them (as
 * bwa_gen_cigar2 / mem_reg2aln leave them), the mate's operations, the XA string.  Results: the records' texts back to back in record
 * order (text_off[k] .. text_off[k+1]), in pinned memory owned by the ctx, valid until its next call of this function. */
typedef struct {
    int64_t pos, m_pos;          /* mem_aln_t::pos of the record and of its mate */
    int64_t cigar_off;           /* blob: n_cigar x uint32, then the NUL-terminated MD string (only read when n_cigar > 0) */
    int64_t m_cigar_off;         /* blob: the mate's m_n_cigar x uint32 */
    int64_t xa_off;              /* blob: the NUL-terminated XA string, or -1 */
    int32_t read;                /* read of the batch resident on the ctx; < 0: an empty slot (its text has length 0), so that a caller can submit one slot per read */
    int32_t flag;                /* mem_aln_t::flag as handed to mem_aln2sam */
    int32_t rid, is_rev, is_alt, mapq, NM, score, sub, n_cigar;
    int32_t has_mate, m_rid, m_is_rev, m_is_alt, m_n_cigar;
    int32_t which;               /* index of the record among its read's (clipping style) */
} meme_sam_rec;
typedef struct { int64_t nrecs; const int64_t* text_off; const char* text; int64_t text_bytes; float kernel_ms; } meme_sam_host_result;
/* names (name_off[r] .. name_off[r+1], no terminator) and qualities (same offsets as the reads' bases; NULL: no qualities, '*') of the
 * batch the last seeding call left on the ctx */
int meme_sam_stage_text(meme_ctx* ctx, const char* names, const int64_t* name_off /* nreads + 1 */, const char* quals);
int meme_sam_format_batch_host(meme_ctx* ctx, const meme_sam_rec* recs, int64_t nrecs, const uint8_t* blob, int64_t blob_bytes,
                               const char* contig_names, const int32_t* contig_name_off /* n_contigs + 1 */, int32_t n_contigs,
                               int32_t softclip /* opt->flag & MEM_F_SOFTCLIP */, const char* rg_id /* bwa_rg_id, may be empty */,
                               meme_sam_host_result* out);

/* ---- mate-rescue Smith-Waterman: the other DP kernel of the SAM phase ---------------------------------------------------------------
 * What mem_sam_pe_batch() (reference src/bwamem_pair.cpp:719-818) computes with kswv::getScores8 / getScores16 (src/kswv.cpp) for the
 * SeqPair jobs mem_matesw_batch_pre() (src/bwamem_pair.cpp:1060-1223) posed: local alignment of a mate (query, len2 bases at qer + idq)
 * inside a window of the reference (target, len1 bases at ref + idr), bases 0..3, 4 = N.  xtra = SeqPair.h0 = KSW_X* flags | threshold
 * (src/ksw.h:31-34): KSW_XBYTE selects the int8 arithmetic of getScores8 (sort_classify, src/bwamem.cpp:1798-1825), KSW_XSUBO the
 * second-best score, KSW_XSTART the second pass for the start.  Results are kswr_t records (src/ksw.h:44-50), field for field what the
 * AVX-512 build of the reference produces (its stripe padding, tie rules and kept-row-maximum rule included).  opt: a, b, o_del, e_del,
 * o_ins, e_ins (match a, mismatch -b, N -1).  Limits: len2 <= 512, len2 * a < 4096, len1 <= 32767 (the reference's own: reads of at most 500 bases, int16 lanes). */
typedef struct { int64_t idr, idq; int32_t len1, len2, xtra, pad; } meme_kswv_job;
typedef struct { int32_t score, te, qe, score2, te2, tb, qb; } meme_kswr;
typedef struct { int64_t njobs; const meme_kswr* res; /* pinned, owned by the ctx, valid until its next kswv call */ float kernel_ms; } meme_kswv_host_result;
int meme_kswv_batch_host(meme_ctx* ctx, const meme_kswv_job* jobs, int64_t njobs, const uint8_t* ref, int64_t ref_bytes, const uint8_t* qer, int64_t qer_bytes,
                         const meme_bsw_opt* opt, meme_kswv_host_result* out);

/* ---- mate rescue whole: the posing step in front of the Smith-Waterman kernel -------------------------------------------------------------
 * What worker_sam's first two steps compute for the read pairs of a chunk (reference src/bwamem.cpp:1855-1877): mem_sam_pe_batch_pre
 * (src/bwamem_pair.cpp:660-716) with mem_matesw_batch_pre (:1060-1223) poses, per worker batch of `batch_reads` reads, the Smith-Waterman jobs
 * of mate rescue -- for each end's alignment records within pen_unpaired of its best (at most max_matesw), the orientations that pes[] allows
 * and no record of the mate explains, a window of the reference clamped to the strand and sequence of its midpoint -- and mem_sam_pe_batch
 * runs them (meme_kswv_batch_host above).  The reads are [first_read, first_read + nreads) of the batch resident on the ctx (both even; pair p = reads
 * 2p, 2p + 1; worker batches are counted from first_read), the text is the index's; regs[reg_off[r] .. reg_off[r+1]) = the fields of read r's mem_alnreg_t records the step reads, in the
 * records' order (best score first, as mem_sort_dedup_patch leaves them).  Results, in pinned memory of the ctx until its next call:
 *   gar[gar_off[b] .. gar_off[b+1])   worker batch b's job index array as mem_matesw_batch_pre fills it: four entries per (end, record) looked at,
 *                                     the job's index AMONG THE BATCH'S JOBS or -1 (not posed: mem_sam_pe_batch_post computes it itself)
 *   res[job_off[b] .. job_off[b+1])   kswr_t records of the batch's jobs in posing order (what mem_sam_pe_batch leaves in `aln`)
 *   jobs[]                            the jobs (len1 window, len2 query, xtra) for inspection; idr / idq index device buffers
 * MEME_E_STATE when the ctx does not hold the batch.  opt: a, b, o_del, e_del, o_ins, e_ins = mem_opt_t fields of the same names. */
typedef struct { int64_t rb; int32_t rid, score; } meme_mate_reg;
typedef struct { int32_t low, high, failed, pad; } meme_pestat;      /* mem_pestat_t::low, high, failed (src/bwamem.h:175-179) */
typedef struct { int32_t a, b, o_del, e_del, o_ins, e_ins, pen_unpaired, max_matesw, min_seed_len, batch_reads; } meme_mate_opt;
typedef struct {
    int64_t nreads, nbatches, njobs, n_gar;
    const int32_t* gar; const int64_t* gar_off; const int64_t* job_off;     /* nbatches + 1 offsets each */
    const meme_kswv_job* jobs; const meme_kswr* res;
    float pose_ms, kernel_ms;        /* HIP events: the posing kernels (+ scans, sequence unpacking); the Smith-Waterman launches */
} meme_mate_host_result;
int meme_matesw_batch_host(meme_ctx* ctx, meme_ctx* reads_of /* NULL: ctx itself; else another ctx of the same GPU whose resident batch is read (not written) */,
                           const meme_mate_reg* regs, const int64_t* reg_off /* nreads + 1 */, int64_t first_read, int64_t nreads, const meme_pestat* pes /* 4 */,
                           const meme_contig* contigs, int32_t n_contigs, int64_t l_pac, const meme_mate_opt* opt, meme_mate_host_result* out);

/* ---- measurement ---------------------------------------------------------------------------------
 * HIP-event timings of the kernels of the last *_device call, measured on the ctx's stream.          */
typedef struct {
    float seed_kernel_ms;      /* SA-search kernel (rounds 1-3 state machine) */
    float seed_gather_ms;      /* offsets scan + SMEM compaction + hit gather */
    float bsw_kernel_ms;
    int64_t seed_launches, bsw_launches;
    float seed_pack_ms;        /* read packing kernel */
    int64_t seed_windows;      /* suffix-array windows loaded by the SA-search kernel */
    float chain_kernel_ms;     /* chaining kernels of the last meme_chain_last_batch_host call (both passes + packing) */
    float chain_pass2_ms;      /* everything after the lane-per-read tier has finished (the routed LDS tiers run beside it), B-tree tier included */
    float chain_tier3_ms;      /* of which the B-tree tier (reads with chains at equal positions or more than 256 chains) */
    int64_t chain_tier2_reads, chain_tier3_reads;   /* reads beyond the lane-per-read tier; of which through the B-tree tier */
    float seed_reseed_ms;      /* of seed_kernel_ms: the re-seeding kernel (k_reseed: unique SMEMs' regions walked on the plcp table) */
    int64_t seed_lane_searches;   /* searches k_reseed did itself, one lane each (what the table cannot answer) */
    int64_t gcig_class_jobs[6];   /* the last CIGAR call's jobs by kernel: 16-lane groups, 32-lane groups, a wavefront each in 64-column chunks, the gap-free shortcut, 64-lane "groups", two columns per lane */
} meme_timings;
int meme_get_timings(meme_ctx* ctx, meme_timings* out);
int meme_set_tuning(meme_ctx* ctx, const char* key, int64_t value);   /* "group_lanes", "seed_blocks_per_cu", "seed_blocks", "smem_cap", "bsw_blocks", "bsw_lane_min_pairs", "chain_wave_tiers", "chain_lane_hits", "chain_light_hits", "seed_defer", "seed_early_tier" (0: overflow tiers strictly behind the re-seeding kernels), "ext_live_only" (1: meme_extend_last_batch_host hands over only the records mem_kernel2_core keeps, src/bwamem.cpp:1680-1693 -- qe > qb, in order -- instead of one per chained seed with the purged ones marked; the stage then also runs in rounds -- "ext_rounds" (default 1; 0: off): a read's seeds are taken in extension order, tested against the read's surviving alignments first and extended only if they survive, one seed per read and round, after that many rounds everything still ahead at once: the same surviving records, without the banded SW of seeds the purge drops), "gcig_groups" (1, default: CIGAR jobs whose band has at most 16 / 32 / 64 columns run 4 / 2 / 1 to a wavefront with one chunk per row; 0: one wavefront per job in 64-column chunks; 2: bands of 65-128 columns as one chunk too, two columns per lane -- measured slower, kept for the record), "gcig_zcap" (>= 0: bytes of LDS a CIGAR job keeps for its backtrack matrix or the window it is walked back through; default: chosen per call, 8192 where a typical matrix of the batch fits, else 2048; results do not depend on it), "chain_side_priority" (1: the routed chaining tiers' streams are created with the highest stream priority -- set before the first chaining call; measured slower, profiles/r05_chain_priority.md; default 0), "ext_census" (1: meme_extend_last_batch_host also counts the extension jobs whose query is a prefix of its target), "max_batch" (> 0: meme_extend_last_batch_host / meme_global_batch_host refuse larger batches with MEME_E_CAPACITY), "ext_split" (default 1: the extension stage's kernels that walk a read's chains run eight lanes per read for reads with at most 8 chained seeds and a wavefront per read for the others; 0: a wavefront per read throughout; same records), "bsw_circ" (default 1: the lane-per-pair banded-SW kernel keeps the columns of queries longer than its band in a ring of 2w + 2 columns -- same results, more wavefronts per CU; 0: one LDS word per query column as before), "sam_max_batch" (> 0: meme_sam_format_batch_host refuses more record slots than this with MEME_E_CAPACITY; the caller formats in pieces) */

#ifdef __cplusplus
}
#endif
#endif
