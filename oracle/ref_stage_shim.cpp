// TEST INFRASTRUCTURE ONLY (oracle/): a C shim of OURS over the *reference's* own functions of the stages between the two
// kernels, so that the test-suite and the fixture generators can call them through ctypes on arbitrary inputs:
//
//   ref_chain_read      mem_chain_Learned() + mem_chain_flt()              reference src/bwamem.cpp:1122-1204, 599-717
//                       (klib B-tree src/kbtree.h, ks_introsort src/ksort.h included -- whatever they do with equal keys)
//   ref_extend_reads    mem_chain2aln_across_reads_V2()                    reference src/bwamem.cpp:2573-3497
//                       with the reference's own BandedPairWiseSW kernels
//   ref_flt_chained_seeds  mem_flt_chained_seeds() with mem_seed_sw()      reference src/bwamem.cpp:565-598, 494-520 (ksw_align2, src/ksw.cpp)
//   ref_kswv_batch      sort_classify() + mem_sam_pe_batch()                reference src/bwamem.cpp:1798-1825, src/bwamem_pair.cpp:719-818
//                       (the AVX-512 mate-rescue kernels kswv::getScores8 / getScores16, src/kswv.cpp)
//   ref_aln2sam         mem_aln2sam()                                      reference src/bwamem.cpp:2174-2312
//   ref_ksw_global2 / ref_gen_cigar2   ksw_global2(), bwa_gen_cigar2() whole (CIGAR + NM + MD)   reference src/ksw.cpp:560-670, src/bwa.cpp:274-362
//
// Linked against oracle/_ref/libbwa_pic.so (the reference's objects, built where the sources lie by oracle/Makefile.ref) into
// oracle/_ref/libstage_ref.so.  Contains no reference code: structures are filled through the reference's headers.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "bwamem.h"      // reference headers, found via -I$(REF)/src at build time
#include "bwa.h"
#include "LearnedIndex_seeding.h"
#include "ksort.h"
#include "ksw.h"

// globals the reference objects expect from the rest of the binary (src/main.cpp)
uint64_t proc_freq = 1, tprof[LIM_R][LIM_C];

void mem_chain_Learned(const mem_opt_t* opt, const bntseq_t* bns, int len, mem_tlv* smems, mem_chain_v* chain, int seqid, u64v* hits,
                       mem_seed_t* seedBuf, int64_t seedBufSize, int64_t& seedBufCount, int tid);
int mem_chain_flt(const mem_opt_t* opt, int n_chn_, mem_chain_t* a_, int tid);
int64_t sort_classify(mem_cache* mmc, int64_t pcnt, int tid);
int mem_sam_pe_batch_pre(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, const mem_pestat_t pes[4], uint64_t id, bseq1_t s[2], mem_alnreg_v a[2], mem_cache* mmc,
                         int64_t& pcnt, int32_t& gcnt, int32_t& maxRefLen, int32_t& maxQerLen, int tid);
void mem_flt_chained_seeds(const mem_opt_t* opt, const bntseq_t* bns, const uint8_t* pac, bseq1_t* seq_, int n_chn, mem_chain_t* a);

#define shim_smem_lt(a, b) ((a).start == (b).start ? (a).end < (b).end : (a).start < (b).start)
KSORT_INIT(shim_smem, mem_tl, shim_smem_lt)

namespace {

struct Bns {
    bntseq_t b;
    bntann1_t* anns;
    Bns(const int64_t* off, const int32_t* len, const uint8_t* alt, int n, int64_t l_pac) {
        memset(&b, 0, sizeof(b));
        anns = (bntann1_t*)calloc((size_t)n, sizeof(bntann1_t));
        for (int i = 0; i < n; ++i) { anns[i].offset = off[i]; anns[i].len = len[i]; anns[i].is_alt = alt ? alt[i] : 0; anns[i].name = (char*)"c"; anns[i].anno = (char*)""; }
        b.l_pac = l_pac; b.n_seqs = n; b.anns = anns;
    }
    ~Bns() { free(anns); }
};

}  // namespace

extern "C" {

struct shim_chain_opt {   // = orc_chain_opt / meme_chain_opt
    int32_t w, max_chain_gap, max_occ, min_seed_len, min_chain_weight, max_chain_extend;
    float mask_level, drop_ratio;
    int64_t l_pac;
};
struct shim_chain { int64_t pos; int32_t rid, n_seeds, w, first, kept, is_alt; int32_t seed_beg; };
struct shim_cseed { int64_t rbeg; int32_t qbeg, len; };

static void fill_opt(mem_opt_t* opt, const shim_chain_opt* o) {
    opt->w = o->w; opt->max_chain_gap = o->max_chain_gap; opt->max_occ = o->max_occ; opt->min_seed_len = o->min_seed_len;
    opt->min_chain_weight = o->min_chain_weight; opt->max_chain_extend = o->max_chain_extend; opt->mask_level = o->mask_level;
    opt->drop_ratio = o->drop_ratio;
}

// One read: SMEMs in any order (sorted here the way mem_kernel1_core_Learned sorts them, src/bwamem.cpp:1397) + hits -> the chains that
// survive the filter.  Returns their number or -2 when a capacity is too small; *tree_size = chains before the filter.
int ref_chain_read(const mem_tl* smems_in, int n_smems, const uint64_t* hits_in, int64_t n_hits, int len, const int64_t* contig_off,
                   const int32_t* contig_len, const uint8_t* contig_alt, int n_contigs, const shim_chain_opt* o, shim_chain* out, int chain_cap,
                   shim_cseed* seeds_out, int seed_cap, int* tree_size, uint32_t* frac_rep_bits) {
    mem_opt_t* opt = mem_opt_init();
    fill_opt(opt, o);
    Bns bns(contig_off, contig_len, contig_alt, n_contigs, o->l_pac);
    mem_tlv smems; u64v hits;
    kv_init(smems); kv_init(hits);
    kv_resize(mem_tl, smems, (size_t)(n_smems + 1));
    kv_resize(uint64_t, hits, (size_t)(n_hits + 1));
    memcpy(smems.a, smems_in, sizeof(mem_tl) * (size_t)n_smems); smems.n = (size_t)n_smems;
    memcpy(hits.a, hits_in, sizeof(uint64_t) * (size_t)n_hits); hits.n = (size_t)n_hits;
    ks_introsort(shim_smem, smems.n, smems.a);
    mem_chain_v chn;
    kv_init(chn);
    const int64_t slab = 1 << 16;
    mem_seed_t* seedBuf = (mem_seed_t*)calloc((size_t)slab, sizeof(mem_seed_t));
    int64_t seedBufCount = 0;
    mem_chain_Learned(opt, &bns.b, len, &smems, &chn, 0, &hits, seedBuf, slab, seedBufCount, 0);
    *tree_size = (int)chn.m;
    *frac_rep_bits = 0;
    if (chn.n) memcpy(frac_rep_bits, &chn.a[0].frac_rep, 4);
    const size_t n_before = chn.n;
    int n = chn.n ? mem_chain_flt(opt, (int)chn.n, chn.a, 0) : 0;      // (frees the seeds of the chains it drops)
    int rc = n, ns = 0;
    for (int k = 0; k < n && rc >= 0; ++k) {
        const mem_chain_t& c = chn.a[k];
        if (k >= chain_cap || ns + c.n > seed_cap) { rc = -2; break; }
        out[k].pos = c.pos; out[k].rid = c.rid; out[k].n_seeds = c.n; out[k].w = (int32_t)c.w; out[k].first = c.first; out[k].kept = (int32_t)c.kept;
        out[k].is_alt = (int32_t)c.is_alt; out[k].seed_beg = ns;
        for (int j = 0; j < c.n; ++j) { seeds_out[ns].rbeg = c.seeds[j].rbeg; seeds_out[ns].qbeg = c.seeds[j].qbeg; seeds_out[ns].len = c.seeds[j].len; ++ns; }
    }
    (void)n_before;
    for (int k = 0; k < n; ++k) if (chn.a[k].m > SEEDS_PER_CHAIN) free(chn.a[k].seeds);
    free(chn.a); free(seedBuf); free(smems.a); free(hits.a); free(opt);
    return rc;
}


// ---- mem_chain2aln_across_reads_V2 on chains the caller brings --------------------------------------------------------------------
// reads: concatenated codes 0..4 with read_off[nreads+1]; chains / seeds flat with chain_off / seed_off per read (seed_beg relative to the
// read's first seed, as ref_chain_read and the device return them); text0123: the fwd+rc text, 1 byte per base (= the reference's
// ref_string).  Output: per read the records the reference leaves in av_v (reg_off[r] = seed_off[r]: one record per chained seed).
struct shim_ext_opt { int32_t a, b, o_del, e_del, o_ins, e_ins, pen_clip5, pen_clip3, w, zdrop; };
struct shim_alnreg {        // the fields of mem_alnreg_t the stage sets
    int64_t rb, re; int32_t qb, qe, rid, score, truesc, sub, alt_sc, csub, sub_n, w, seedcov, secondary, secondary_all, seedlen0, n_comp, is_alt;
    float frac_rep; int32_t pad;
};

int ref_extend_reads_scored(const uint8_t* reads, const int64_t* read_off, int64_t nreads, const int64_t* chain_off, const shim_chain* chains,
                            const int64_t* seed_off, const shim_cseed* seeds, const int32_t* seed_score, const uint32_t* frac_rep_bits, uint8_t* text0123,
                            const int64_t* contig_off, const int32_t* contig_len, const uint8_t* contig_alt, int n_contigs, int64_t l_pac,
                            const shim_ext_opt* eo, shim_alnreg* out);
int ref_extend_reads(const uint8_t* reads, const int64_t* read_off, int64_t nreads, const int64_t* chain_off, const shim_chain* chains,
                     const int64_t* seed_off, const shim_cseed* seeds, const uint32_t* frac_rep_bits, uint8_t* text0123,
                     const int64_t* contig_off, const int32_t* contig_len, const uint8_t* contig_alt, int n_contigs, int64_t l_pac,
                     const shim_ext_opt* eo, shim_alnreg* out) {
    return ref_extend_reads_scored(reads, read_off, nreads, chain_off, chains, seed_off, seeds, nullptr, frac_rep_bits, text0123, contig_off, contig_len, contig_alt,
                                   n_contigs, l_pac, eo, out);
}

// mem_flt_chained_seeds on chains the caller brings (layout as above).  In place: a read's surviving seeds are packed to the front of its
// range chain after chain, seed_beg / n_seeds of its chains follow, score[] = mem_seed_t::score afterwards, kept[r] = seeds that stay.
int ref_flt_chained_seeds(const uint8_t* reads, const int64_t* read_off, int64_t nreads, const int64_t* chain_off, shim_chain* chains,
                          const int64_t* seed_off, shim_cseed* seeds, int32_t* score, const uint8_t* text0123, const int64_t* contig_off,
                          const int32_t* contig_len, const uint8_t* contig_alt, int n_contigs, int64_t l_pac, const shim_ext_opt* eo, int min_chain_weight,
                          int64_t* kept) {
    mem_opt_t* opt = mem_opt_init();
    opt->a = eo->a; opt->b = eo->b; opt->o_del = eo->o_del; opt->e_del = eo->e_del; opt->o_ins = eo->o_ins; opt->e_ins = eo->e_ins;
    opt->min_chain_weight = min_chain_weight;
    bwa_fill_scmat(opt->a, opt->b, opt->mat);
    Bns bns(contig_off, contig_len, contig_alt, n_contigs, l_pac);
    // what the aligner hands the function as `pac` (src/bwamem.cpp:1770): worker_t::rc_pac -- forward + reverse complement, 2 bits per base,
    // every byte through the reference's own BitReverseTable256 (src/LearnedIndex_seeding.h:129-137; the recipe of src/fastmap.cpp:440-457,
    // followed here on the caller's text)
    std::vector<uint8_t> pac((size_t)((2 * l_pac + 3) / 4 + 2), 0);
    for (int64_t i = 0; i < 2 * l_pac; ++i) pac[(size_t)(i >> 2)] |= (uint8_t)((text0123[i] & 3) << ((~i & 3) << 1));
    for (uint8_t& b : pac) b = BitReverseTable256[b];
    for (int64_t r = 0; r < nreads; ++r) {
        bseq1_t seq;
        memset(&seq, 0, sizeof(seq));
        seq.l_seq = (int)(read_off[r + 1] - read_off[r]);
        seq.seq = (char*)(reads + read_off[r]);
        const int64_t c0 = chain_off[r], nc = chain_off[r + 1] - c0;
        std::vector<mem_chain_t> a((size_t)(nc ? nc : 1));
        for (int64_t k = 0; k < nc; ++k) {
            const shim_chain& sc = chains[c0 + k];
            mem_chain_t& c = a[(size_t)k];
            memset(&c, 0, sizeof(c));
            c.seqid = 0; c.n = c.m = sc.n_seeds; c.rid = sc.rid; c.pos = sc.pos;
            c.seeds = (mem_seed_t*)calloc((size_t)(sc.n_seeds ? sc.n_seeds : 1), sizeof(mem_seed_t));
            for (int j = 0; j < sc.n_seeds; ++j) {
                const shim_cseed& sd = seeds[seed_off[r] + sc.seed_beg + j];
                c.seeds[j].rbeg = sd.rbeg; c.seeds[j].qbeg = sd.qbeg; c.seeds[j].len = sd.len; c.seeds[j].score = sd.len;
            }
        }
        mem_flt_chained_seeds(opt, &bns.b, pac.data(), &seq, (int)nc, a.data());
        int64_t n = 0;
        for (int64_t k = 0; k < nc; ++k) {
            const mem_chain_t& c = a[(size_t)k];
            chains[c0 + k].seed_beg = (int32_t)n; chains[c0 + k].n_seeds = c.n;
            for (int j = 0; j < c.n; ++j, ++n) {
                shim_cseed& sd = seeds[seed_off[r] + n];
                sd.rbeg = c.seeds[j].rbeg; sd.qbeg = c.seeds[j].qbeg; sd.len = c.seeds[j].len;
                score[seed_off[r] + n] = c.seeds[j].score;
            }
            free(c.seeds);
        }
        kept[r] = n;
    }
    free(opt);
    return 0;
}

int ref_extend_reads_scored(const uint8_t* reads, const int64_t* read_off, int64_t nreads, const int64_t* chain_off, const shim_chain* chains,
                            const int64_t* seed_off, const shim_cseed* seeds, const int32_t* seed_score, const uint32_t* frac_rep_bits, uint8_t* text0123,
                            const int64_t* contig_off, const int32_t* contig_len, const uint8_t* contig_alt, int n_contigs, int64_t l_pac,
                            const shim_ext_opt* eo, shim_alnreg* out) {
    mem_opt_t* opt = mem_opt_init();
    opt->a = eo->a; opt->b = eo->b; opt->o_del = eo->o_del; opt->e_del = eo->e_del; opt->o_ins = eo->o_ins; opt->e_ins = eo->e_ins;
    opt->pen_clip5 = eo->pen_clip5; opt->pen_clip3 = eo->pen_clip3; opt->w = eo->w; opt->zdrop = eo->zdrop;
    bwa_fill_scmat(opt->a, opt->b, opt->mat);
    Bns bns(contig_off, contig_len, contig_alt, n_contigs, l_pac);
    // worker buffers as memoryAllocLearned sizes them for one thread (src/fastmap.cpp:351-420)
    static mem_cache mmc;
    static bool init = false;
    if (!init) {
        init = true;
        const int64_t wsize = BATCH_SIZE * SEEDS_PER_READ;
        mmc.seqBufLeftRef[0] = (uint8_t*)_mm_malloc((size_t)wsize * MAX_SEQ_LEN_REF + MAX_LINE_LEN, 64);
        mmc.seqBufLeftQer[0] = (uint8_t*)_mm_malloc((size_t)wsize * MAX_SEQ_LEN_QER + MAX_LINE_LEN, 64);
        mmc.seqBufRightRef[0] = (uint8_t*)_mm_malloc((size_t)wsize * MAX_SEQ_LEN_REF + MAX_LINE_LEN, 64);
        mmc.seqBufRightQer[0] = (uint8_t*)_mm_malloc((size_t)wsize * MAX_SEQ_LEN_QER + MAX_LINE_LEN, 64);
        mmc.wsize_buf_ref[0] = wsize * MAX_SEQ_LEN_REF;
        mmc.wsize_buf_qer[0] = wsize * MAX_SEQ_LEN_QER;
        mmc.seqPairArrayAux[0] = (SeqPair*)malloc((size_t)(wsize + MAX_LINE_LEN) * sizeof(SeqPair));
        mmc.seqPairArrayLeft128[0] = (SeqPair*)malloc((size_t)(wsize + MAX_LINE_LEN) * sizeof(SeqPair));
        mmc.seqPairArrayRight128[0] = (SeqPair*)malloc((size_t)(wsize + MAX_LINE_LEN) * sizeof(SeqPair));
        mmc.wsize[0] = wsize;
        mmc.lim[0] = (int32_t*)_mm_malloc((BATCH_SIZE + 32) * sizeof(int32_t), 64);
    }
    for (int64_t g0 = 0; g0 < nreads; g0 += BATCH_SIZE) {
        const int nseq = (int)(nreads - g0 < BATCH_SIZE ? nreads - g0 : BATCH_SIZE);
        std::vector<bseq1_t> seq((size_t)nseq);
        std::vector<mem_chain_v> chn((size_t)nseq);
        std::vector<mem_alnreg_v> av((size_t)nseq);
        for (int l = 0; l < nseq; ++l) {
            const int64_t r = g0 + l;
            memset(&seq[l], 0, sizeof(bseq1_t));
            seq[l].l_seq = (int)(read_off[r + 1] - read_off[r]);
            seq[l].seq = (char*)(reads + read_off[r]);
            kv_init(chn[l]);
            memset(&av[l], 0, sizeof(mem_alnreg_v));
            const int64_t c0 = chain_off[r], nc = chain_off[r + 1] - c0;
            chn[l].n = chn[l].m = (size_t)nc;
            chn[l].a = (mem_chain_t*)calloc((size_t)(nc ? nc : 1), sizeof(mem_chain_t));
            for (int64_t k = 0; k < nc; ++k) {
                const shim_chain& sc = chains[c0 + k];
                mem_chain_t& c = chn[l].a[k];
                c.seqid = l; c.n = c.m = sc.n_seeds; c.first = sc.first; c.rid = sc.rid; c.w = (uint32_t)sc.w; c.kept = (uint32_t)sc.kept;
                c.is_alt = (uint32_t)sc.is_alt; memcpy(&c.frac_rep, &frac_rep_bits[r], 4); c.pos = sc.pos;
                c.seeds = (mem_seed_t*)calloc((size_t)sc.n_seeds, sizeof(mem_seed_t));
                for (int j = 0; j < sc.n_seeds; ++j) {
                    const shim_cseed& sd = seeds[seed_off[r] + sc.seed_beg + j];
                    c.seeds[j].rbeg = sd.rbeg; c.seeds[j].qbeg = sd.qbeg; c.seeds[j].len = sd.len;
                    c.seeds[j].score = seed_score ? seed_score[seed_off[r] + sc.seed_beg + j] : sd.len;
                }
            }
        }
        mem_chain2aln_across_reads_V2(opt, &bns.b, text0123 /* pac: unused when ref_string is given */, seq.data(), nseq, chn.data(), av.data(), &mmc, text0123, 0);
        for (int l = 0; l < nseq; ++l) {
            const int64_t r = g0 + l;
            if ((int64_t)av[l].n != seed_off[r + 1] - seed_off[r]) return -1;
            for (size_t i = 0; i < av[l].n; ++i) {
                const mem_alnreg_t& a = av[l].a[i];
                shim_alnreg& o = out[seed_off[r] + (int64_t)i];
                o.rb = a.rb; o.re = a.re; o.qb = a.qb; o.qe = a.qe; o.rid = a.rid; o.score = a.score; o.truesc = a.truesc; o.sub = a.sub; o.alt_sc = a.alt_sc;
                o.csub = a.csub; o.sub_n = a.sub_n; o.w = a.w; o.seedcov = a.seedcov; o.secondary = a.secondary; o.secondary_all = a.secondary_all;
                o.seedlen0 = a.seedlen0; o.n_comp = a.n_comp; o.is_alt = a.is_alt; o.frac_rep = a.frac_rep; o.pad = 0;
            }
            free(av[l].a);
            for (size_t k = 0; k < chn[l].n; ++k) free(chn[l].a[k].seeds);
            free(chn[l].a);
        }
    }
    free(opt);
    return 0;
}


// ---- ksw_global2 (src/ksw.cpp:560-670) and bwa_gen_cigar2 (src/bwa.cpp:274-362) ------------------------------------------------------
// cigar receives at most cap operations; returns the score, *n_cigar the number of operations (or -1 when cap is too small)
int ref_ksw_global2(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int a, int b, int o_del, int e_del, int o_ins, int e_ins, int w,
                    int* n_cigar, uint32_t* cigar, int cap) {
    int8_t mat[25];
    bwa_fill_scmat(a, b, mat);
    uint32_t* cg = nullptr;
    int n = 0;
    const int score = ksw_global2(qlen, query, tlen, target, 5, mat, o_del, e_del, o_ins, e_ins, w, &n, &cg);
    if (n > cap) n = -1;
    for (int i = 0; i < n; ++i) cigar[i] = cg[i];
    free(cg);
    *n_cigar = n;
    return score;
}

// bwa_gen_cigar2 whole (src/bwa.cpp:274-362): `fwd` = the forward strand as codes 0..3 (packed here into the reference's pac array, 4 bases per
// byte, first base in the top bits -- _set_pac, src/bntseq.cpp), query codes 0..4.  out[0..3] = score, n_cigar, NM, length of the MD string;
// cigar / md receive the operations and the string (NUL included).  Returns 0, or -1 when the function returned no CIGAR (a rejected call).
int ref_gen_cigar2(const uint8_t* fwd, int64_t l_pac, int a, int b, int o_del, int e_del, int o_ins, int e_ins, int w_, int l_query, const uint8_t* query, int64_t rb,
                   int64_t re, int32_t* out, uint32_t* cigar, int cap, char* md, int md_cap) {
    static const uint8_t* pac_of = nullptr;
    static std::vector<uint8_t> pac;
    if (pac_of != fwd || (int64_t)pac.size() != l_pac / 4 + 1) {
        pac.assign((size_t)(l_pac / 4 + 1), 0);
        for (int64_t i = 0; i < l_pac; ++i) pac[(size_t)(i >> 2)] |= (uint8_t)((fwd[i] & 3) << ((~i & 3) << 1));
        pac_of = fwd;
    }
    int8_t mat[25];
    bwa_fill_scmat(a, b, mat);
    std::vector<uint8_t> q(query, query + l_query);
    int score = 0, n_cigar = 0, NM = 0;
    uint32_t* cg = bwa_gen_cigar2(mat, o_del, e_del, o_ins, e_ins, w_, l_pac, pac.data(), l_query, q.data(), rb, re, &score, &n_cigar, &NM);
    if (memcmp(q.data(), query, (size_t)l_query) != 0) { free(cg); return -2; }      // (the function reverses the query in place and must put it back)
    if (!cg) { out[0] = score; out[1] = n_cigar; out[2] = NM; out[3] = -1; return -1; }
    const char* m = (const char*)(cg + n_cigar);
    const int l = (int)strlen(m);
    out[0] = score; out[1] = n_cigar; out[2] = NM; out[3] = l;
    if (n_cigar > cap || l + 1 > md_cap) { free(cg); return -3; }
    memcpy(cigar, cg, (size_t)n_cigar * 4);
    memcpy(md, m, (size_t)l + 1);
    free(cg);
    return 0;
}

// mem_aln2sam (src/bwamem.cpp:2174-2312) for one record of a read that has `n_list` records (here 1) and, optionally, a mate's record.
// rec = the fields of meme_sam_rec / orc_sam_rec (16 x int32 after 5 x int64), blob = cigars / MD / XA as the record names them.
// seq: codes.  out receives the text; returns its length or -1 when cap is too small.
struct shim_sam_rec {
    int64_t pos, m_pos, cigar_off, m_cigar_off, xa_off;
    int32_t read, flag, rid, is_rev, is_alt, mapq, NM, score, sub, n_cigar, has_mate, m_rid, m_is_rev, m_is_alt, m_n_cigar, which;
};
int64_t ref_aln2sam(const shim_sam_rec* r, const uint8_t* blob, const char* name, const uint8_t* seq, int l_seq, const char* qual, const char* contig_names,
                    const int32_t* contig_name_off, int n_contigs, int softclip, const char* rg_id, char* out, int64_t cap) {
    mem_opt_t* opt = mem_opt_init();
    if (softclip) opt->flag |= MEM_F_SOFTCLIP;
    bntseq_t bns;
    memset(&bns, 0, sizeof(bns));
    bns.n_seqs = n_contigs;
    std::vector<bntann1_t> anns((size_t)n_contigs);
    std::vector<std::string> names((size_t)n_contigs);
    for (int i = 0; i < n_contigs; ++i) {
        names[(size_t)i].assign(contig_names + contig_name_off[i], contig_names + contig_name_off[i + 1]);
        memset(&anns[(size_t)i], 0, sizeof(bntann1_t));
        anns[(size_t)i].name = (char*)names[(size_t)i].c_str();
    }
    bns.anns = anns.data();
    strncpy(bwa_rg_id, rg_id ? rg_id : "", 255);
    std::string nm(name), sq((const char*)seq, (size_t)l_seq), ql(qual ? qual : "");
    bseq1_t s;
    memset(&s, 0, sizeof(s));
    s.name = (char*)nm.c_str(); s.seq = (char*)sq.data(); s.qual = qual ? (char*)ql.c_str() : nullptr; s.l_seq = l_seq;
    mem_aln_t p, m;
    memset(&p, 0, sizeof(p)); memset(&m, 0, sizeof(m));
    // (the operations and, behind them, the MD string: one block, as mem_reg2aln leaves it)
    std::vector<uint32_t> pc((size_t)(r->n_cigar > 0 ? r->n_cigar : 0) + 64), mc((size_t)(r->m_n_cigar > 0 ? r->m_n_cigar : 0) + 1);
    if (r->n_cigar > 0) {
        memcpy(pc.data(), blob + r->cigar_off, (size_t)r->n_cigar * 4);
        const char* md = (const char*)(blob + r->cigar_off + 4 * (int64_t)r->n_cigar);
        pc.resize((size_t)r->n_cigar + strlen(md) / 4 + 2);
        strcpy((char*)(pc.data() + r->n_cigar), md);
    }
    if (r->m_n_cigar > 0) memcpy(mc.data(), blob + r->m_cigar_off, (size_t)r->m_n_cigar * 4);
    p.pos = r->pos; p.rid = r->rid; p.flag = r->flag; p.is_rev = r->is_rev; p.is_alt = r->is_alt; p.mapq = r->mapq; p.NM = r->NM; p.n_cigar = r->n_cigar;
    p.cigar = r->n_cigar > 0 ? pc.data() : nullptr; p.XA = r->xa_off >= 0 ? (char*)(blob + r->xa_off) : nullptr; p.score = r->score; p.sub = r->sub; p.alt_sc = 0;
    m.pos = r->m_pos; m.rid = r->m_rid; m.is_rev = r->m_is_rev; m.is_alt = r->m_is_alt; m.n_cigar = r->m_n_cigar; m.cigar = r->m_n_cigar > 0 ? mc.data() : nullptr;
    // `which` > 0: the record is the which-th of its read's list; the others are secondary (0x100), so that no SA tag is written (the device's scope)
    std::vector<mem_aln_t> list((size_t)r->which + 1);
    for (auto& x : list) { memset(&x, 0, sizeof(x)); x.flag = 0x100; x.rid = 0; }
    list[(size_t)r->which] = p;
    kstring_t str = {0, 0, 0};
    mem_aln2sam(opt, &bns, &str, &s, r->which + 1, list.data(), r->which, r->has_mate ? &m : nullptr);
    int64_t n = (int64_t)str.l;
    if (n > cap) n = -1; else memcpy(out, str.s, (size_t)n);
    free(str.s); free(opt);
    bwa_rg_id[0] = 0;
    return n;
}


// ---- the mate-rescue Smith-Waterman batch exactly as worker_sam runs it (src/bwamem.cpp:1871-1877) --------------------------------------
// jobs: idr / idq offsets into ref / qer, len1 (target), len2 (query), xtra (SeqPair.h0).  out[i] = kswr_t of job i.
struct shim_kswv_job { int64_t idr, idq; int32_t len1, len2, xtra, pad; };
int ref_kswv_batch(const shim_kswv_job* jobs, int64_t n, const uint8_t* ref, int64_t ref_bytes, const uint8_t* qer, int64_t qer_bytes, int a, int b, int o_del,
                   int e_del, int o_ins, int e_ins, int32_t* out /* n x 7 */) {
#if __AVX512BW__
    mem_opt_t* opt = mem_opt_init();
    opt->a = a; opt->b = b; opt->o_del = o_del; opt->e_del = e_del; opt->o_ins = o_ins; opt->e_ins = e_ins;
    bwa_fill_scmat(a, b, opt->mat);
    mem_cache* mmc = (mem_cache*)calloc(1, sizeof(mem_cache));
    const size_t cap = (size_t)n + 2 * MAX_LINE_LEN + 128;
    mmc->seqPairArrayLeft128[0] = (SeqPair*)calloc(cap, sizeof(SeqPair));
    mmc->seqPairArrayRight128[0] = (SeqPair*)calloc(cap, sizeof(SeqPair));
    mmc->seqPairArrayAux[0] = (SeqPair*)calloc(cap, sizeof(SeqPair));
    uint8_t* r = (uint8_t*)_mm_malloc((size_t)ref_bytes + 4096, 64);
    uint8_t* q = (uint8_t*)_mm_malloc((size_t)qer_bytes + 4096, 64);
    memcpy(r, ref, (size_t)ref_bytes); memset(r + ref_bytes, 0, 4096);     // (the batch reverses prefixes in place)
    memcpy(q, qer, (size_t)qer_bytes); memset(q + qer_bytes, 0, 4096);
    mmc->seqBufLeftRef[0] = r; mmc->seqBufLeftQer[0] = q;
    mmc->wsize[0] = (int64_t)cap; mmc->wsize_buf_ref[0] = ref_bytes + 4096; mmc->wsize_buf_qer[0] = qer_bytes + 4096;
    int32_t maxRef = 0, maxQer = 0;
    SeqPair* sp = mmc->seqPairArrayLeft128[0];
    for (int64_t i = 0; i < n; ++i) {
        SeqPair p;
        memset(&p, 0xff, sizeof(p));
        p.idr = (int32_t)jobs[i].idr; p.idq = (int32_t)jobs[i].idq; p.len1 = jobs[i].len1; p.len2 = jobs[i].len2; p.h0 = jobs[i].xtra; p.regid = (int)i;
        sp[i] = p;
        maxRef = maxRef > p.len1 ? maxRef : p.len1; maxQer = maxQer > p.len2 ? maxQer : p.len2;
    }
    int64_t pcnt = n;
    int64_t pcnt8 = sort_classify(mmc, pcnt, 0);
    kswr_t* aln = (kswr_t*)_mm_malloc((size_t)(pcnt + SIMD_WIDTH8) * sizeof(kswr_t), 64);
    mem_sam_pe_batch(opt, mmc, pcnt, pcnt8, aln, maxRef, maxQer, 0);
    for (int64_t i = 0; i < n; ++i) {
        int32_t* o = out + 7 * i;
        o[0] = aln[i].score; o[1] = aln[i].te; o[2] = aln[i].qe; o[3] = aln[i].score2; o[4] = aln[i].te2; o[5] = aln[i].tb; o[6] = aln[i].qb;
    }
    _mm_free(aln); _mm_free(r); _mm_free(q);
    free(mmc->seqPairArrayLeft128[0]); free(mmc->seqPairArrayRight128[0]); free(mmc->seqPairArrayAux[0]);
    free(mmc); free(opt);
    return 0;
#else
    (void)jobs; (void)n; (void)ref; (void)ref_bytes; (void)qer; (void)qer_bytes; (void)a; (void)b; (void)o_del; (void)e_del; (void)o_ins; (void)e_ins; (void)out;
    return -1;          // the batched kernels only exist in the AVX-512 build (src/bwamem.cpp:1838: other builds run mem_sam_pe / ksw_align2)
#endif
}


// ---- the posing step of mate rescue exactly as worker_sam runs it for one worker batch (src/bwamem.cpp:1855-1866): mem_sam_pe_batch_pre
// (src/bwamem_pair.cpp:660-716) -> mem_matesw_batch_pre (:1060-1223) per pair, into a mem_cache of the shim's own.  regs: the fields of the
// reads' alignment records the step reads.  Outputs: gar (the step's job index array, -1 where it wrote nothing), the SeqPair jobs it posed
// (idr, idq, len1, len2, h0 as xtra) and the two sequence buffers.
struct shim_mate_reg { int64_t rb; int32_t rid, score; };
int64_t ref_matesw_pose(const uint8_t* fwd, int64_t l_pac, const int64_t* contig_off, const int32_t* contig_len, int n_contigs, const uint8_t* reads,
                        const int64_t* read_off, int64_t first, int64_t count, const shim_mate_reg* regs, const int64_t* reg_off, const int32_t* pes_lhf /* 4 x {low, high, failed} */,
                        int a, int pen_unpaired, int max_matesw, int min_seed_len, int32_t* gar, int64_t gar_cap, int64_t* n_gar, shim_kswv_job* jobs, int64_t job_cap,
                        uint8_t* ref_out, int64_t ref_cap, int64_t* ref_bytes, uint8_t* qer_out, int64_t qer_cap, int64_t* qer_bytes) {
    Bns bns(contig_off, contig_len, nullptr, n_contigs, l_pac);
    std::vector<uint8_t> pac((size_t)(l_pac / 4 + 1), 0);
    for (int64_t i = 0; i < l_pac; ++i) pac[(size_t)(i >> 2)] |= (uint8_t)((fwd[i] & 3) << ((~i & 3) << 1));
    mem_opt_t* opt = mem_opt_init();
    opt->a = a; opt->pen_unpaired = pen_unpaired; opt->max_matesw = max_matesw; opt->min_seed_len = min_seed_len;
    bwa_fill_scmat(opt->a, opt->b, opt->mat);
    mem_pestat_t pes[4];
    memset(pes, 0, sizeof(pes));
    for (int r = 0; r < 4; ++r) { pes[r].low = pes_lhf[3 * r]; pes[r].high = pes_lhf[3 * r + 1]; pes[r].failed = pes_lhf[3 * r + 2]; }
    mem_cache* mmc = (mem_cache*)calloc(1, sizeof(mem_cache));
    const size_t cap = (size_t)job_cap + 2 * MAX_LINE_LEN + 128;
    mmc->seqPairArrayLeft128[0] = (SeqPair*)calloc(cap, sizeof(SeqPair));
    mmc->seqPairArrayRight128[0] = (SeqPair*)calloc(cap, sizeof(SeqPair));
    const size_t gar_pairs = ((size_t)gar_cap * 4 + sizeof(SeqPair) - 1) / sizeof(SeqPair) + cap;
    mmc->seqPairArrayAux[0] = (SeqPair*)malloc(gar_pairs * sizeof(SeqPair));
    memset(mmc->seqPairArrayAux[0], 0xff, gar_pairs * sizeof(SeqPair));                  // (-1 wherever the step does not write)
    uint8_t* r = (uint8_t*)_mm_malloc((size_t)ref_cap + 4096, 64);
    uint8_t* q = (uint8_t*)_mm_malloc((size_t)qer_cap + 4096, 64);
    mmc->seqBufLeftRef[0] = r; mmc->seqBufLeftQer[0] = q;
    mmc->seqBufRightRef[0] = (uint8_t*)_mm_malloc(64, 64); mmc->seqBufRightQer[0] = (uint8_t*)_mm_malloc(64, 64);
    mmc->wsize[0] = (int64_t)job_cap; mmc->wsize_buf_ref[0] = ref_cap; mmc->wsize_buf_qer[0] = qer_cap;     // (sized by the caller so that the step never reallocates)
    int64_t pcnt = 0;
    int32_t gcnt = 0, maxRef = 0, maxQer = 0;
    for (int64_t p = first; p + 1 < first + count; p += 2) {
        bseq1_t s[2];
        mem_alnreg_v av[2];
        memset(s, 0, sizeof(s));
        for (int i = 0; i < 2; ++i) {
            const int64_t rd = p + i;
            s[i].seq = (char*)(reads + read_off[rd]);
            s[i].l_seq = (int)(read_off[rd + 1] - read_off[rd]);
            const int64_t n = reg_off[rd + 1] - reg_off[rd];
            kv_init(av[i]);
            av[i].n = av[i].m = (size_t)n;
            av[i].a = (mem_alnreg_t*)calloc((size_t)(n ? n : 1), sizeof(mem_alnreg_t));
            for (int64_t k = 0; k < n; ++k) { av[i].a[k].rb = regs[reg_off[rd] + k].rb; av[i].a[k].rid = regs[reg_off[rd] + k].rid; av[i].a[k].score = regs[reg_off[rd] + k].score; }
        }
        mem_sam_pe_batch_pre(opt, &bns.b, pac.data(), pes, (uint64_t)(p >> 1), s, av, mmc, pcnt, gcnt, maxRef, maxQer, 0);
        free(av[0].a); free(av[1].a);
        if (pcnt > job_cap || gcnt > gar_cap) return -1;
    }
    const int32_t* g = (const int32_t*)mmc->seqPairArrayAux[0];
    for (int32_t k = 0; k < gcnt; ++k) gar[k] = g[k];
    *n_gar = gcnt;
    const SeqPair* sp = mmc->seqPairArrayLeft128[0];
    int64_t rb = 0, qb = 0;
    for (int64_t k = 0; k < pcnt; ++k) {
        jobs[k].idr = sp[k].idr; jobs[k].idq = sp[k].idq; jobs[k].len1 = sp[k].len1; jobs[k].len2 = sp[k].len2; jobs[k].xtra = sp[k].h0; jobs[k].pad = 0;
        rb = sp[k].idr + sp[k].len1; qb = sp[k].idq + sp[k].len2;
    }
    memcpy(ref_out, mmc->seqBufLeftRef[0], (size_t)rb); memcpy(qer_out, mmc->seqBufLeftQer[0], (size_t)qb);
    *ref_bytes = rb; *qer_bytes = qb;
    _mm_free(mmc->seqBufLeftRef[0]); _mm_free(mmc->seqBufLeftQer[0]); _mm_free(mmc->seqBufRightRef[0]); _mm_free(mmc->seqBufRightQer[0]);
    free(mmc->seqPairArrayLeft128[0]); free(mmc->seqPairArrayRight128[0]); free(mmc->seqPairArrayAux[0]);
    free(mmc); free(opt);
    return pcnt;
}

}  // extern "C"
