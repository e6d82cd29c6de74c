/* TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the BWA-MEME learned-index seeding
 * path and of the banded Smith-Waterman seed extension.  Nothing under oracle/ is linked into,
 * imported by or executed from the product path (bwa-meme_amd/, include/); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this oracle against seed dumps and
 * SeqPair dumps produced by the compiled reference (oracle/_ref, built from /root/reference by
 * oracle/Makefile.ref; generator script tests/golden/make_golden.py), and tests/test_ref_live.py
 * re-checks it against the reference binaries whenever oracle/_ref is present.
 */
#ifndef MEME_ORACLE_H
#define MEME_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* = mem_tl, reference src/LearnedIndex_seeding.h:121-127 */
typedef struct {
    int32_t start, end;      /* [start,end) in the read */
    int32_t hitbeg, hitcount;
    uint64_t cache_refpos;   /* text position of the first hit */
} orc_mem_tl;

/* the index as plain arrays: fwd+rc text, 1 byte per base (".0123"), and the suffix array */
typedef struct {
    const uint8_t* text;
    const uint64_t* sa;
    int64_t n;               /* = 2*l_pac */
} orc_index;

typedef struct {
    int32_t min_seed_len;    /* opt->min_seed_len (19) */
    int32_t split_len;       /* (int)(min_seed_len*split_factor+.499) (28) */
    int32_t split_width;     /* opt->split_width (10) */
    int32_t max_mem_intv;    /* opt->max_mem_intv (20); 0 disables round 3 */
    int32_t steps;           /* 1: round 1 only, 2: +re-seeding, 3: +round 3 (test harness "steps") */
} orc_seed_params;

/* work counters: the deterministic per-dataset figures of SURVEY.md section 8(d) */
typedef struct {
    int64_t searches;        /* locate operations (right/left, any round) */
    int64_t level_steps;     /* interval (count) evaluations */
    int64_t smems, hits;
} orc_counters;

/* Seeds one read (codes 0..3, >=4 = N).  Returns 0, or -1 if a capacity was exceeded. */
int orc_seed_read(const orc_index* idx, const uint8_t* read, int32_t len, const orc_seed_params* p,
                  orc_mem_tl* smems, int32_t smem_cap, int32_t* n_smems,
                  uint64_t* hits, int64_t hit_cap, int64_t* n_hits, orc_counters* ctr);

/* Batch driver (OpenMP over reads).  reads: concatenated codes, read_off[nreads+1].
 * Outputs are per-read slots: smems[r*smem_cap ..], n_smems[r]; hits are written through
 * hit_off (computed by a first counting pass inside): caller passes hits capacity. */
int orc_seed_batch(const orc_index* idx, const uint8_t* reads, const int64_t* read_off, int64_t nreads,
                   const orc_seed_params* p, orc_mem_tl* smems, int32_t smem_cap, int32_t* n_smems,
                   uint64_t* hits, int64_t hit_cap_per_read, int64_t* n_hits, orc_counters* ctr,
                   int threads);

/* mem_aln2sam for the only record of a read (reference src/bwamem.cpp:2174-2312); orc_sam_rec = meme_sam_rec of include/meme_hip.h */
typedef struct {
    int64_t pos, m_pos, cigar_off, m_cigar_off, xa_off;
    int32_t read, flag, rid, is_rev, is_alt, mapq, NM, score, sub, n_cigar, has_mate, m_rid, m_is_rev, m_is_alt, m_n_cigar, which;
} orc_sam_rec;
int64_t orc_aln2sam(const orc_sam_rec* r, const uint8_t* blob, const char* name, int l_name, const uint8_t* seq, int l_seq, const char* qual,
                    const char* contig_names, const int32_t* contig_name_off, int softclip, const char* rg_id, char* out);

/* measurement only: census of the searches a diagonal + plcp shortcut could answer (see meme_oracle.c); plcp[u] per TEXT position, out = 4 x 6 counters */
void orc_diag_census_enable(const uint8_t* plcp);
void orc_diag_census_get(long long* out);
void orc_build_plcp(const uint8_t* text, const uint64_t* sa, int64_t n, uint8_t* plcp);

/* single primitives, exposed for unit tests */
int orc_compare(const orc_index* idx, uint64_t sa_slot, const uint8_t* q, int64_t valid_len,
                uint32_t* match_len, int* exact);
/* locate + level search: returns match_len L = max l <= maxLCP with f(l) >= min_intv;
 * *start,*count = SA interval of suffixes sharing >= L bases with q */
uint32_t orc_search(const orc_index* idx, const uint8_t* q, int64_t valid_len, int32_t min_intv,
                    int64_t* start, int64_t* count, orc_counters* ctr);

/* ---- banded Smith-Waterman extension ---------------------------------------------------------- */
typedef struct {
    int32_t o_del, e_del, o_ins, e_ins, zdrop, end_bonus, a, b;
} orc_bsw_params;

/* = SeqPair, reference src/bandedSWA.h:90-99 (56 bytes) */
typedef struct {
    int32_t idr, idq, id;
    int32_t len1, len2;
    int32_t h0;
    int32_t seqid, regid;
    int32_t score, tle, gtle, qle;
    int32_t gscore, max_off;
} orc_seqpair;

int orc_bsw_extend(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int w, int h0,
                   const orc_bsw_params* p, int* qle, int* tle, int* gtle, int* gscore, int* max_off,
                   int64_t* cells);
void orc_bsw_batch(orc_seqpair* pairs, const uint8_t* ref, const uint8_t* qer, int32_t n, int32_t w,
                   const orc_bsw_params* p, int threads, int64_t* cells);

/* ---- chaining (mem_chain_Learned + mem_chain_flt, reference src/bwamem.cpp:1122-1204, 599-717) ---------------------------- */
typedef struct {
    int32_t w, max_chain_gap, max_occ, min_seed_len, min_chain_weight, max_chain_extend;
    float mask_level, drop_ratio;
    int64_t l_pac;
} orc_chain_opt;
typedef struct { int64_t pos; int32_t rid, n_seeds, w, first, kept, is_alt; int32_t seed_beg; } orc_chain;
typedef struct { int64_t rbeg; int32_t qbeg, len; } orc_cseed;
/* Chains of one read from its SMEMs (any order) and hits: the chains that survive the filter, in the filter's output order, into
 * out[chain_cap] / seeds_out[seed_cap].  Returns their number (chains at equal positions are ordered as the reference's
 * B-tree, src/kbtree.h, orders them: the tree is restated); -2 when a capacity is too small.  *tree_size = chains before
 * the filter, *frac_rep as mem_chain_Learned computes it. */
int orc_chain_read(const orc_mem_tl* smems, int n_smems, const uint64_t* hits, int len, const int64_t* contig_off,
                   const uint8_t* contig_alt, int n_contigs, const orc_chain_opt* o, orc_chain* out, int chain_cap,
                   orc_cseed* seeds_out, int seed_cap, int* tree_size, float* frac_rep);

/* ---- seed extension (mem_chain2aln_across_reads_V2, reference src/bwamem.cpp:2573-3497) ------------------------------------------- */
typedef struct { int32_t a, b, o_del, e_del, o_ins, e_ins, pen_clip5, pen_clip3, w, zdrop; } orc_ext_opt;
typedef struct {                 /* the fields of mem_alnreg_t the stage sets (src/bwamem.h:143-165) */
    int64_t rb, re; int32_t qb, qe, rid, score, truesc, sub, alt_sc, csub, sub_n, w, seedcov, secondary, secondary_all, seedlen0, n_comp, is_alt;
    float frac_rep; int32_t pad;
} orc_alnreg;
/* One read: its chains (as orc_chain_read returns them; seed_beg indexes `seeds`) -> one record per chained seed in extension order
 * (chain after chain, best seed of a chain first), purged records marked qb = qe = -1.  text = fwd+rc codes, 1 byte per base.
 * Returns the number of records; *n_jobs / *n_retried count the banded-SW calls and those with the doubled band. */
int orc_extend_read(const uint8_t* read, int l_query, const orc_chain* chains, int n_chains, const orc_cseed* seeds, float frac_rep,
                    const uint8_t* text, int64_t l_pac, const int64_t* contig_off, const int32_t* contig_len, const orc_ext_opt* o,
                    orc_alnreg* out, int64_t* n_jobs, int64_t* n_retried);
int orc_extend_batch(const uint8_t* reads, const int64_t* read_off, int64_t nreads, const int64_t* chain_off, const orc_chain* chains,
                     const int64_t* seed_off, const orc_cseed* seeds, const float* frac_rep, const uint8_t* text, int64_t l_pac,
                     const int64_t* contig_off, const int32_t* contig_len, const orc_ext_opt* o, orc_alnreg* out, int threads, int64_t* stats);

/* the same with the seeds' scores as mem_flt_chained_seeds leaves them (NULL: score = length) */
int orc_extend_read_scored(const uint8_t* read, int l_query, const orc_chain* chains, int n_chains, const orc_cseed* seeds, const int32_t* seed_score,
                           float frac_rep, const uint8_t* text, int64_t l_pac, const int64_t* contig_off, const int32_t* contig_len, const orc_ext_opt* o,
                           orc_alnreg* out, int64_t* n_jobs, int64_t* n_retried);
int orc_extend_batch_scored(const uint8_t* reads, const int64_t* read_off, int64_t nreads, const int64_t* chain_off, const orc_chain* chains,
                            const int64_t* seed_off, const orc_cseed* seeds, const int32_t* seed_score, const float* frac_rep, const uint8_t* text, int64_t l_pac,
                            const int64_t* contig_off, const int32_t* contig_len, const orc_ext_opt* o, orc_alnreg* out, int threads, int64_t* stats);

/* ---- mem_flt_chained_seeds (reference src/bwamem.cpp:565-598) and mem_seed_sw (:494-520, ksw_i16 src/ksw.cpp:236-320) ----------------
 * orc_seed_sw: -1 (no alignment needed) or the local alignment score of the seed's neighbourhood.  orc_flt_chained_seeds: one read --
 * seeds that fail leave their chains: the read's seeds are packed to the front chain after chain, seed_beg / n_seeds follow, score[] =
 * mem_seed_t::score afterwards; returns the number of seeds that stay. */
int orc_seed_sw(const uint8_t* read, int l_query, const orc_cseed* s, const uint8_t* text, int64_t l_pac, const int64_t* contig_off,
                const int32_t* contig_len, int n_contigs, const orc_ext_opt* o);
int orc_flt_chained_seeds(const uint8_t* read, int l_query, orc_chain* chains, int n_chains, orc_cseed* seeds, int32_t* score, const uint8_t* text,
                          int64_t l_pac, const int64_t* contig_off, const int32_t* contig_len, int n_contigs, const orc_ext_opt* o, int min_chain_weight,
                          int64_t* n_sw);

/* every read of a batch (flat layout of orc_extend_batch; seeds / score indexed by the OLD seed_off, kept[r] = seeds of read r that stay) */
int orc_flt_batch(const uint8_t* reads, const int64_t* read_off, int64_t nreads, const int64_t* chain_off, orc_chain* chains, const int64_t* seed_off,
                  orc_cseed* seeds, int32_t* score, const uint8_t* text, int64_t l_pac, const int64_t* contig_off, const int32_t* contig_len, int n_contigs,
                  const orc_ext_opt* o, int min_chain_weight, int64_t* kept, int threads, int64_t* n_sw);

/* ---- banded global alignment with traceback (ksw_global2, reference src/ksw.cpp:560-670) ------------------------------------------ */
int orc_ksw_global2(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int a, int b, int o_del, int e_del, int o_ins, int e_ins, int w,
                    int* n_cigar, uint32_t* cigar /* capacity qlen + tlen + 2 */);
/* bwa_gen_cigar2 whole (reference src/bwa.cpp:274-362): CIGAR + NM + MD; see meme_oracle.c */
int orc_gen_cigar2(const uint8_t* text, int64_t l_pac, int a, int b, int o_del, int e_del, int o_ins, int e_ins, int w_, int l_query, const uint8_t* query,
                   int64_t rb, int64_t re, int* score, int* n_cigar, uint32_t* cigar, int* NM, char* md);

/* ---- mate-rescue Smith-Waterman of the SAM phase (kswv::getScores8 / getScores16 as driven by mem_sam_pe_batch; reference
 * src/kswv.cpp:372-712, 934-1199, src/bwamem_pair.cpp:719-818): forward pass, second-best score, reverse pass for the start ---------- */
typedef struct { int64_t idr, idq; int32_t len1 /* target */, len2 /* query */, xtra /* SeqPair.h0: KSW_X* flags | threshold */, pad; } orc_kswv_job;
typedef struct { int32_t score, te, qe, score2, te2, tb, qb; } orc_kswr;          /* kswr_t, src/ksw.h:44-50 */
void orc_kswv_pair(const uint8_t* target, int tlen, const uint8_t* query, int qlen, int xtra, int a, int b, int o_del, int e_del, int o_ins, int e_ins,
                   orc_kswr* r, int64_t* cells);
int orc_kswv_batch(const orc_kswv_job* jobs, int64_t n, const uint8_t* ref, const uint8_t* qer, int a, int b, int o_del, int e_del, int o_ins, int e_ins,
                   orc_kswr* out, int threads, int64_t* cells);


/* ---- mate rescue, the posing step: mem_sam_pe_batch_pre (reference src/bwamem_pair.cpp:660-716) with mem_matesw_batch_pre (:1060-1223) --------
 * For the read pairs of one worker batch (reads [first, first + count), count even; pair = two consecutive reads): which Smith-Waterman jobs
 * worker_sam's first step poses -- per end i of a pair its alignment records with score >= (best score - pen_unpaired), at most max_matesw of them,
 * each against the windows of the four orientations that the insert-size statistics allow and no record of the mate already explains
 * (mem_infer_dir :58-65, skip[] :1082-1097), the window clamped to the strand and the reference sequence of its midpoint (bns_fetch_seq,
 * src/bntseq.cpp:541-570), kept when it lies on the record's sequence and holds at least min_seed_len bases (:1134).
 * regs[reg_off[r] .. reg_off[r+1]): the fields of read r's mem_alnreg_t records the step reads.  Outputs in the step's own order:
 *   gar[4 * q + o]   for the q-th (end, record) the step looks at: index of the job of orientation o among the BATCH's jobs, or -1
 *   jobs[k]          len1 (window), len2 (mate length), xtra (:1135), rb (window start in fwd+rc coordinates), read (the mate's index), is_rev
 * Returns the number of jobs; *n_gar receives the number of gar entries (4 per (end, record)).  text = fwd+rc bases (1 byte each). */
typedef struct { int64_t rb; int32_t rid, score; } orc_mate_reg;
typedef struct { int32_t low, high, failed, pad; } orc_pestat;
typedef struct { int32_t a, pen_unpaired, max_matesw, min_seed_len; } orc_mate_opt;
typedef struct { int64_t rb; int32_t read, len1, len2, xtra, is_rev, pad; } orc_mate_job;
int64_t orc_matesw_pose(const orc_mate_reg* regs, const int64_t* reg_off, int64_t first, int64_t count, const int32_t* read_len, const orc_pestat* pes /* 4 */,
                        int64_t l_pac, const int64_t* contig_off, const int32_t* contig_len, int n_contigs, const orc_mate_opt* opt,
                        int32_t* gar, int64_t gar_cap, int64_t* n_gar, orc_mate_job* jobs, int64_t job_cap);
/* the sequences of one posed job as the step stores them: `ref` = text[rb, rb + len1), `qer` = the mate (reversed and complemented when is_rev) */
void orc_matesw_job_seqs(const orc_mate_job* j, const uint8_t* text, const uint8_t* reads, const int64_t* read_off, uint8_t* ref, uint8_t* qer);

#ifdef __cplusplus
}
#endif
#endif
