// Seed extension of a whole batch on the device: mem_chain2aln_across_reads_V2 (reference src/bwamem.cpp:2573-3497) behind the chaining
// kernels -- seeds, chains, extension jobs and their sequences never leave HBM; the host receives mem_alnreg_t records.
//
//   k_flt_pose / k_seedsw / k_flt_count / k_flt_move   mem_flt_chained_seeds (:565-598), only for reads it is not a no-op for (long
//                      reads, -W): seeds whose neighbourhood aligns poorly leave their chains before anything is extended
//   k_ext_jobs<false>  per read: the reference span of every chain (:2648-2690 + bns_fetch_seq_v2, src/bntseq.cpp:479-512), the
//                      number of left / right extension jobs and of sequence bytes                                  -> scans
//   k_ext_jobs<true>   per read: one alignment record per chained seed in extension order (best seed of a chain first, :2692-2702),
//                      the left (:2722-2781) and right (:2783-2850) jobs as SeqPair records, their sequences copied from the 2-bit
//                      text and the read (left jobs reversed), the per-chain seed order the purge step walks
//   meme_bsw           k_bsw_lane / k_bsw<LP> on the jobs of one direction and band width (meme_bsw.hip)
//   k_ext_fold<left>   results into the records (:2985-3018 and siblings); jobs whose band was too narrow to a retry list (MAX_BAND_TRY 2)
//   k_ext_h0           start score of the right extension = score after the left one (:3371-3376)
//   k_ext_purge        alignments of seeds an earlier alignment of the read already covers are dropped (:3389-3485)
//
// One wavefront per read in the kernels that walk a read's chains (a read has 1 to a few hundred chained seeds: lanes take seeds, the
// wavefront reduces / scans across them, and a repeat-rich read cannot hold 63 others back); one lane per job in the others.
#include <math.h>
#include <string.h>

#include "meme_common.h"

namespace {

constexpr int H0 = -99;                    // H0_, src/macro.h:44
constexpr int EXT_BAND_TRIES = 2;          // MAX_BAND_TRY, src/bwamem.cpp:62

struct ExtArgs {
    const uint8_t* reads; const i64* read_off; i64 g0, ns;            // reads [g0, g0 + ns) of the batch
    const i64* chain_off; const meme_chain* chains; const i64* seed_off; const meme_chain_seed* seeds; const float* frac_rep;
    const int* seed_score;            // per chained seed, or null: score = length (what chaining leaves, src/bwamem.cpp:1163)
    const u64* pac; i64 l_pac;
    const i64* contig_off; const int* contig_len;
    meme_ext_opt o;
    i64* rmax;                        // [2 * chain]: reference span of a chain
    meme_alnreg* regs; int* order;    // per chained seed of the batch
    i64 *cntL, *cntR, *cntB;          // per read (from g0): left jobs, right jobs, sequence bytes (k_ext_jobs<false>)
    const i64 *offL, *offR, *offB;    // their exclusive scans over the batch (pointing at read g0)
    i64 job0L, job0R, byte0;          // offsets of this slab's first job / byte (subtracted: SeqPair offsets are 32-bit)
    meme_seqpair* L; meme_seqpair* R; uint8_t* seq;
    // extension in rounds (tuning "ext_rounds"): mode 0 = every chained seed at once (the reference's batch, :2573-3388); 1 = records and the extension
    // order only, no jobs; 2 = jobs of the seeds k_ext_advance selected (sel: per chained seed; act: per read, any selected)
    int mode; const uint8_t* sel; const uint8_t* act; int4* state;
    // round 6: the kernels that walk a read's chains exist in two widths -- a wavefront per read (G = 64) for the reads in `list` (more than
    // EXT_LIGHT chained seeds), and EIGHT LANES per read (G = 8: eight reads per wavefront) for the others, which are the many: a 150-bp read
    // has four chained seeds on average and a wavefront's life is a chain of dependent loads, not work.  list == nullptr: every read of the slab.
    const i64* list; i64 nlist; int light;           // light: G = 8 launch -- reads with more than EXT_LIGHT seeds are skipped (they are in the other launch's list)
};
constexpr int EXT_LIGHT = 8;

__device__ __forceinline__ int cal_max_gap(const meme_ext_opt& o, int qlen) {       // src/bwamem.cpp:85-95
    const int l_del = (int)((double)(qlen * o.a - o.o_del) / o.e_del + 1.);
    const int l_ins = (int)((double)(qlen * o.a - o.o_ins) / o.e_ins + 1.);
    int l = l_del > l_ins ? l_del : l_ins;
    l = l > 1 ? l : 1;
    return l < o.w << 1 ? l : o.w << 1;
}
__device__ __forceinline__ int text_base(const u64* pac, i64 p) { return (int)(pac[p >> 5] >> (62 - 2 * (int)(p & 31))) & 3; }
__device__ __forceinline__ i64 wave_min(i64 v) { for (int d = 32; d >= 1; d >>= 1) { const i64 y = __shfl_xor(v, d); v = y < v ? y : v; } return v; }
__device__ __forceinline__ i64 wave_max(i64 v) { for (int d = 32; d >= 1; d >>= 1) { const i64 y = __shfl_xor(v, d); v = y > v ? y : v; } return v; }
__device__ __forceinline__ int pad4(int x) { return (x + 3) & ~3; }
// sub-wavefront groups of G lanes (G = 64: the wavefront): ballots, broadcasts and reductions that stay inside the group
template <int G> __device__ __forceinline__ u64 gballot(bool p, int gbase) { const u64 b = __ballot(p); return G == 64 ? b : (b >> gbase) & (((u64)1 << (G & 63)) - 1); }
template <int G> __device__ __forceinline__ i64 group_min(i64 v) { for (int d = G / 2; d >= 1; d >>= 1) { const i64 y = __shfl_xor(v, d); v = y < v ? y : v; } return v; }
template <int G> __device__ __forceinline__ i64 group_max(i64 v) { for (int d = G / 2; d >= 1; d >>= 1) { const i64 y = __shfl_xor(v, d); v = y > v ? y : v; } return v; }

// seeds of the chain fully inside the alignment (src/bwamem.cpp:2907-2917 and after every fold)
__device__ inline void seedcov(meme_alnreg* a, const meme_chain_seed* sd, int n) {
    if (a->rb == H0 || a->qb == H0 || a->qe == H0 || a->re == H0) return;
    int cov = 0;
    for (int i = 0; i < n; ++i) {
        const meme_chain_seed t = sd[i];
        if (t.qbeg >= a->qb && t.qbeg + t.len <= a->qe && t.rbeg >= a->rb && t.rbeg + t.len <= a->re) cov += t.len;
    }
    a->seedcov = cov;
}

template <bool WRITE, int G>
__global__ void __launch_bounds__(256) k_ext_jobs(ExtArgs A) {
    const i64 it = (i64)blockIdx.x * (256 / G) + threadIdx.x / G;  // a group of G lanes per read (G = 64: a wavefront each, four to a workgroup -- 2 M one-wave workgroups take 4 ms to dispatch)
    if (it >= (A.list ? A.nlist : A.ns)) return;
    const i64 r = A.list ? A.list[it] : A.g0 + it;
    const i64 rl = r - A.g0;                          // read of the slab
    const int lane = threadIdx.x & (G - 1);           // lane of the group
    const int gbase = (threadIdx.x & 63) & ~(G - 1);  // the group's first lane in its wavefront
    const i64 c0 = A.chain_off[r];
    const int nc = (int)(A.chain_off[r + 1] - c0);
    const i64 s0 = A.seed_off[r];
    const int S = (int)(A.seed_off[r + 1] - s0);
    if (A.light && S > EXT_LIGHT) return;             // (the wavefront-per-read launch has it)
    const int l_query = (int)(A.read_off[r + 1] - A.read_off[r]);
    const meme_ext_opt& o = A.o;
    if (A.mode == 2 && !A.act[r]) {                  // nothing of this read in the round
        if (!WRITE && lane == 0) { A.cntL[rl] = 0; A.cntR[rl] = 0; A.cntB[rl] = 0; }
        return;
    }
    if (WRITE ? A.mode == 1 : A.mode == 0) {
        // ---- reference span of every chain: the widest any of its seeds may reach (:2648-2690)
        for (int c = 0; c < nc; ++c) {
            const meme_chain ch = A.chains[c0 + c];
            const meme_chain_seed* sd = A.seeds + s0 + ch.seed_beg;
            i64 b_min = A.l_pac << 1, e_max = 0;
            for (int i = lane; i < ch.n_seeds; i += G) {
                const meme_chain_seed t = sd[i];
                const i64 b = t.rbeg - (t.qbeg + cal_max_gap(o, t.qbeg));
                const int tail = l_query - t.qbeg - t.len;
                const i64 e = t.rbeg + t.len + (tail + cal_max_gap(o, tail));
                b_min = b < b_min ? b : b_min;
                e_max = e > e_max ? e : e_max;
            }
            i64 rmax0 = group_min<G>(b_min), rmax1 = group_max<G>(e_max);
            if (rmax0 < 0) rmax0 = 0;
            if (rmax1 > A.l_pac << 1) rmax1 = A.l_pac << 1;
            const i64 mid = sd[0].rbeg;
            if (rmax0 < A.l_pac && A.l_pac < rmax1) { if (mid < A.l_pac) rmax1 = A.l_pac; else rmax0 = A.l_pac; }
            // bns_fetch_seq_v2: the span stays inside the chain's reference sequence (on the strand of its first seed)
            i64 far_beg = A.contig_off[ch.rid], far_end = far_beg + A.contig_len[ch.rid];
            if (mid >= A.l_pac) { const i64 t = far_beg; far_beg = (A.l_pac << 1) - far_end; far_end = (A.l_pac << 1) - t; }
            rmax0 = rmax0 > far_beg ? rmax0 : far_beg;
            rmax1 = rmax1 < far_end ? rmax1 : far_end;
            A.rmax[2 * (c0 + c)] = rmax0;
            A.rmax[2 * (c0 + c) + 1] = rmax1;
        }
    }
    // ---- one lane per chained seed: its place in the extension order, its two jobs
    i64 nL = 0, nR = 0, nB = 0;                       // running totals of the read (uniform)
    const bool posing = WRITE && A.mode != 1;
    const i64 jobL0 = posing ? A.offL[rl] - A.job0L : 0, jobR0 = posing ? A.offR[rl] - A.job0R : 0;
    const i64 byte0 = posing ? A.offB[rl] - A.byte0 : 0;
    for (int jb = 0; jb < S; jb += G) {
        const int j = jb + lane;
        const bool valid = j < S;
        int c = 0;
        if (valid) {                                  // the chain of seed j: chains are packed in order, seed_beg ascending
            int lo = 0, hi = nc - 1;
            while (lo < hi) { const int m = (lo + hi + 1) >> 1; if (A.chains[c0 + m].seed_beg <= j) lo = m; else hi = m - 1; }
            c = lo;
        }
        const meme_chain ch = A.chains[c0 + (valid ? c : 0)];
        const meme_chain_seed* sd = A.seeds + s0 + ch.seed_beg;
        const int il = valid ? j - ch.seed_beg : 0;
        const meme_chain_seed t = sd[il];
        const i64 rmax0 = A.rmax[2 * (c0 + c)], rmax1 = A.rmax[2 * (c0 + c) + 1];
        const bool pick = valid && A.mode != 1 && (A.mode == 0 || A.sel[s0 + j]);              // its jobs are posed by this launch
        const bool sideL = valid && t.qbeg > 0, sideR = valid && t.qbeg + t.len != l_query;   // the seed has a left / right flank at all
        const bool hasL = pick && sideL, hasR = pick && sideR;
        const int qe = t.qbeg + t.len;
        const int l2L = t.qbeg, l1L = (int)(t.rbeg - rmax0);
        const int l2R = l_query - qe, l1R = (int)(rmax1 - (t.rbeg + t.len));
        const int bytesL = hasL ? pad4(l1L) + pad4(l2L) : 0, bytesR = hasR ? pad4(l1R) + pad4(l2R) : 0;
        const u64 mL = gballot<G>(hasL, gbase), mR = gballot<G>(hasR, gbase), below = ((u64)1 << lane) - 1;
        // exclusive scan of the bytes over the lanes
        int bsum = bytesL + bytesR, bex;
        {
            int x = bsum;
            for (int d = 1; d < G; d <<= 1) { const int y = __shfl_up(x, d); if (lane >= d) x += y; }
            bex = x - bsum;
            bsum = __shfl(x, gbase + G - 1);
        }
        if (WRITE && valid && (A.mode != 2 || pick)) {
            // rank among the chain's seeds by (score, index): ks_introsort_64 on score << 32 | index, keys unique (:2692-2699)
            int rank = 0;
            if (A.seed_score) {
                const int* ss = A.seed_score + s0 + ch.seed_beg;
                const int mine = ss[il];
                for (int k = 0; k < ch.n_seeds; ++k) { const int lk = ss[k]; rank += (lk < mine || (lk == mine && k < il)) ? 1 : 0; }
            } else
                for (int k = 0; k < ch.n_seeds; ++k) { const int lk = sd[k].len; rank += (lk < t.len || (lk == t.len && k < il)) ? 1 : 0; }
            if (A.mode != 2) A.order[s0 + ch.seed_beg + rank] = il;
            const i64 reg = s0 + ch.seed_beg + (ch.n_seeds - 1 - rank);        // best seed first
            meme_alnreg a;
            memset(&a, 0, sizeof(a));
            a.w = o.w; a.score = a.truesc = -1; a.rid = ch.rid; a.frac_rep = A.frac_rep[r]; a.seedlen0 = t.len; a.c = (u64)(c0 + c);
            a.rb = a.re = H0; a.qb = a.qe = H0;
            if (sideL) { a.qb = t.qbeg; a.rb = t.rbeg; } else { a.score = a.truesc = t.len * o.a; a.qb = 0; a.rb = t.rbeg; }
            if (sideR) { a.qe = qe; a.re = t.rbeg + t.len; } else { a.qe = l_query; a.re = t.rbeg + t.len; seedcov(&a, sd, ch.n_seeds); }
            if (A.mode != 2) A.regs[reg] = a;
            if (hasL) {
                meme_seqpair sp;
                memset(&sp, 0, sizeof(sp));
                sp.h0 = t.len * o.a; sp.seqid = (int32_t)r; sp.regid = (int32_t)reg; sp.len1 = l1L; sp.len2 = l2L;
                sp.idr = (int32_t)(byte0 + nB + bex); sp.idq = sp.idr + pad4(l1L);
                A.L[jobL0 + nL + __popcll(mL & below)] = sp;
            }
            if (hasR) {
                meme_seqpair sp;
                memset(&sp, 0, sizeof(sp));
                sp.h0 = H0; sp.seqid = (int32_t)r; sp.regid = (int32_t)reg; sp.len1 = l1R; sp.len2 = l2R;
                sp.idr = (int32_t)(byte0 + nB + bex + bytesL); sp.idq = sp.idr + pad4(l1R);
                A.R[jobR0 + nR + __popcll(mR & below)] = sp;
            }
        }
        if (WRITE) {
            // the sequences, job after job, 64 lanes x 4 bases per step: left jobs reversed (both sequences run away from the seed)
            const uint8_t* rd = A.reads + A.read_off[r];
            for (int side = 0; side < 2; ++side) {
                u64 m = side ? mR : mL;
                while (m) {
                    const int jl = __builtin_ctzll(m);
                    m &= m - 1;
                    const i64 rbeg = __shfl(t.rbeg, gbase + jl);
                    const int qb = __shfl(t.qbeg, gbase + jl), ln = __shfl(t.len, gbase + jl);
                    const int l1 = __shfl(side ? l1R : l1L, gbase + jl), l2 = __shfl(side ? l2R : l2L, gbase + jl);
                    const i64 dst = byte0 + nB + __shfl(bex, gbase + jl) + (side ? __shfl(bytesL, gbase + jl) : 0);
                    uint8_t* dr = A.seq + dst;
                    uint8_t* dq = dr + pad4(l1);
                    for (int i = lane * 4; i < l1; i += 4 * G) {
                        unsigned v = 0;
                        for (int b = 0; b < 4; ++b) {
                            const int k = i + b;
                            const int code = k < l1 ? text_base(A.pac, side ? rbeg + ln + k : rbeg - 1 - k) : 0;
                            v |= (unsigned)code << (8 * b);
                        }
                        *reinterpret_cast<unsigned*>(dr + i) = v;
                    }
                    for (int i = lane * 4; i < l2; i += 4 * G) {
                        unsigned v = 0;
                        for (int b = 0; b < 4; ++b) {
                            const int k = i + b;
                            const int code = k < l2 ? rd[side ? qb + ln + k : qb - 1 - k] : 0;
                            v |= (unsigned)code << (8 * b);
                        }
                        *reinterpret_cast<unsigned*>(dq + i) = v;
                    }
                }
            }
        }
        nL += __popcll(mL); nR += __popcll(mR); nB += bsum;
    }
    if (!WRITE && lane == 0) { A.cntL[rl] = nL; A.cntR[rl] = nR; A.cntB[rl] = nB; }
    if (WRITE && A.mode == 1 && lane == 0) A.state[r] = make_int4(0, nc ? A.chains[c0].n_seeds - 1 : 0, 0, 0);   // k_ext_advance starts at the best seed of the first chain
}

// ---- mem_flt_chained_seeds (src/bwamem.cpp:565-598) ---------------------------------------------------------------------------------
struct FltArgs {
    const i64* read_off; i64 n;
    const i64* chain_off; meme_chain* chains; const i64* seed_off; const meme_chain_seed* seeds;
    i64 l_pac; const i64* contig_off; const int* contig_len; int n_contigs;
    const int* hsp;                  // per read length: min_HSP_score (:582), -1 where the filter does not run (:583)
    int a;
    int* sc;                         // per chained seed: FLT_OFF, -1 (kept without alignment, :502 / :515) or mem_seed_sw's score
    meme_seedsw_job* jobs; unsigned long long* n_jobs;
    i64* cnt;                        // per read: seeds that stay
    const i64* off2;                 // its exclusive scan = the new seed_off
    meme_chain_seed* seeds2; int* score2;
};
constexpr int FLT_OFF = INT32_MIN;
constexpr int SEEDSW_EXT = 50;      // MEM_SHORT_EXT, src/bwamem.cpp:249

// the alignments mem_seed_sw would run (src/bwamem.cpp:494-520, bns_fetch_seq src/bntseq.cpp:541-570): one lane per chained seed
__global__ void __launch_bounds__(64) k_flt_pose(FltArgs A) {
    const i64 r = blockIdx.x;
    if (r >= A.n) return;
    const int lane = threadIdx.x;
    const i64 s0 = A.seed_off[r];
    const int S = (int)(A.seed_off[r + 1] - s0);
    const i64 q0 = A.read_off[r];
    const int l_query = (int)(A.read_off[r + 1] - q0);
    const int hsp = A.hsp[l_query];
    for (int jb = 0; jb < S; jb += 64) {
        const int j = jb + lane;
        bool job = false;
        meme_seedsw_job J;
        if (j < S) {
            int v = FLT_OFF;
            if (hsp >= 0) {
                v = -1;
                const meme_chain_seed t = A.seeds[s0 + j];
                if (t.len < MEME_SEEDSW_MAX) {
                    int qb = t.qbeg, qe = t.qbeg + t.len;
                    i64 rb = t.rbeg, re = t.rbeg + t.len;
                    const i64 mid = (rb + re) >> 1;
                    qb -= SEEDSW_EXT; qb = qb > 0 ? qb : 0;
                    qe += SEEDSW_EXT; qe = qe < l_query ? qe : l_query;
                    rb -= SEEDSW_EXT; rb = rb > 0 ? rb : 0;
                    re += SEEDSW_EXT; re = re < A.l_pac << 1 ? re : A.l_pac << 1;
                    if (rb < A.l_pac && A.l_pac < re) { if (mid < A.l_pac) re = A.l_pac; else rb = A.l_pac; }
                    if (!(qe - qb >= MEME_SEEDSW_MAX || re - rb >= MEME_SEEDSW_MAX)) {
                        // the window stays inside the reference sequence of the seed's midpoint, on its strand
                        const bool rev = mid >= A.l_pac;
                        const i64 fpos = rev ? (A.l_pac << 1) - 1 - mid : mid;
                        int lo = 0, hi = A.n_contigs - 1;
                        while (lo < hi) { const int m = (lo + hi + 1) >> 1; if (A.contig_off[m] <= fpos) lo = m; else hi = m - 1; }
                        i64 far_beg = A.contig_off[lo], far_end = far_beg + A.contig_len[lo];
                        if (rev) { const i64 x = far_beg; far_beg = (A.l_pac << 1) - far_end; far_end = (A.l_pac << 1) - x; }
                        rb = rb > far_beg ? rb : far_beg;
                        re = re < far_end ? re : far_end;
                        J.rb = rb; J.qoff = q0 + qb; J.seed = (int)(s0 + j); J.tlen = (short)(re - rb); J.qlen = (short)(qe - qb);
                        job = true;
                    }
                }
            }
            A.sc[s0 + j] = v;
        }
        const u64 m = __ballot(job);
        if (m) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(A.n_jobs, (unsigned long long)__popcll(m));
            base = __shfl(base, 0);
            if (job) A.jobs[base + __popcll(m & (((u64)1 << lane) - 1))] = J;
        }
    }
}

__device__ __forceinline__ bool flt_keeps(int v, int hsp) { return v == FLT_OFF || v < 0 || v >= hsp; }      // (:589)

// seeds that stay: per read (-> scan -> the new seed_off), per chain (n_seeds; seed_beg = the kept seeds of the read's earlier chains)
template <bool MOVE>
__global__ void __launch_bounds__(64) k_flt_apply(FltArgs A) {
    const i64 r = blockIdx.x;
    if (r >= A.n) return;
    const int lane = threadIdx.x;
    const i64 c0 = A.chain_off[r];
    const int nc = (int)(A.chain_off[r + 1] - c0);
    const i64 s0 = A.seed_off[r];
    const int hsp = A.hsp[(int)(A.read_off[r + 1] - A.read_off[r])];
    const i64 d0 = MOVE ? A.off2[r] : 0;
    int kept = 0;                                     // of the read so far (uniform)
    for (int c = 0; c < nc; ++c) {
        const meme_chain ch = A.chains[c0 + c];
        int kc = 0;
        for (int jb = 0; jb < ch.n_seeds; jb += 64) {
            const int j = jb + lane;
            const i64 g = s0 + ch.seed_beg + (j < ch.n_seeds ? j : 0);
            const int v = A.sc[g];
            const bool keep = j < ch.n_seeds && flt_keeps(v, hsp);
            const u64 m = __ballot(keep);
            if (MOVE && keep) {
                const meme_chain_seed t = A.seeds[g];
                const i64 d = d0 + kept + kc + __popcll(m & (((u64)1 << lane) - 1));
                A.seeds2[d] = t;
                A.score2[d] = v == FLT_OFF ? t.len : (v < 0 ? t.len * A.a : v);       // (:591)
            }
            kc += __popcll(m);
        }
        if (MOVE && lane == 0) { A.chains[c0 + c].seed_beg = kept; A.chains[c0 + c].n_seeds = kc; }
        kept += kc;
    }
    if (!MOVE && lane == 0) A.cnt[r] = kept;
}

struct FoldArgs {
    meme_seqpair* pairs; i64 n;
    meme_alnreg* regs; const meme_chain* chains; const i64* seed_off; const meme_chain_seed* seeds; const i64* read_off;
    meme_ext_opt o; int w, last;
    meme_seqpair* retry; unsigned long long* n_retry;
};

// results of one stage into the alignment records (src/bwamem.cpp:2985-3018 left, :3253-3290 right, and their siblings)
template <bool LEFT>
__global__ void __launch_bounds__(256) k_ext_fold(FoldArgs F) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < F.n; i += (i64)gridDim.x * blockDim.x) {
        const meme_seqpair sp = F.pairs[i];
        meme_alnreg* a = &F.regs[sp.regid];
        const int prev = a->score;
        a->score = sp.score;
        if (a->score == prev || sp.max_off < (F.w >> 1) + (F.w >> 2) || F.last) {
            if (LEFT) {
                if (sp.gscore <= 0 || sp.gscore <= a->score - F.o.pen_clip5) { a->qb -= sp.qle; a->rb -= sp.tle; a->truesc = a->score; }
                else { a->qb = 0; a->rb -= sp.gtle; a->truesc = sp.gscore; }
            } else {
                if (sp.gscore <= 0 || sp.gscore <= a->score - F.o.pen_clip3) { a->qe += sp.qle; a->re += sp.tle; a->truesc += a->score - sp.h0; }
                else { a->qe = (int)(F.read_off[sp.seqid + 1] - F.read_off[sp.seqid]); a->re += sp.gtle; a->truesc += sp.gscore - sp.h0; }
            }
            a->w = a->w > F.w ? a->w : F.w;
            const meme_chain ch = F.chains[a->c];
            seedcov(a, F.seeds + F.seed_off[sp.seqid] + ch.seed_beg, ch.n_seeds);
        } else F.retry[atomicAdd(F.n_retry, 1ull)] = sp;
    }
}

__global__ void __launch_bounds__(256) k_ext_h0(meme_seqpair* __restrict__ pairs, i64 n, const meme_alnreg* __restrict__ regs) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) pairs[i].h0 = regs[pairs[i].regid].score;
}

struct PurgeArgs {
    const i64* read_off; i64 nreads;
    const i64* chain_off; const meme_chain* chains; const i64* seed_off; const meme_chain_seed* seeds;
    meme_alnreg* regs; int* order;
    meme_ext_opt o;
};

// (:3389-3485) in the order the one-read-at-a-time aligner would have met the seeds: chain after chain, best seed first
__global__ void __launch_bounds__(256) k_ext_purge(PurgeArgs P) {
    const i64 r = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= P.nreads) return;
    const int lane = threadIdx.x & 63;
    const i64 c0 = P.chain_off[r];
    const int nc = (int)(P.chain_off[r + 1] - c0);
    const i64 s0 = P.seed_off[r];
    const int S = (int)(P.seed_off[r + 1] - s0);
    if (S < 2) return;                                 // a lone alignment has nothing before it
    const int l_query = (int)(P.read_off[r + 1] - P.read_off[r]);
    meme_alnreg* av = P.regs + s0;
    int cur = 0;                                       // record of the seed at hand = number of seeds met so far
    for (int c = 0; c < nc; ++c) {
        const meme_chain ch = P.chains[c0 + c];
        const meme_chain_seed* sd = P.seeds + s0 + ch.seed_beg;
        int* ord = P.order + s0 + ch.seed_beg;
        for (int k = ch.n_seeds - 1; k >= 0; --k, ++cur) {
            const meme_chain_seed s = sd[ord[k]];
            // an earlier, surviving alignment that contains the seed and has it within its band on either side?
            bool found = false;
            for (int ib = 0; ib < cur && !found; ib += 64) {
                const int i = ib + lane;
                bool hit = false;
                if (i < cur) {
                    const meme_alnreg* p = &av[i];
                    const i64 prb = p->rb, pre = p->re;
                    const int pqb = p->qb, pqe = p->qe;
                    if (!(pqb == -1 && pqe == -1) && !(s.rbeg < prb || s.rbeg + s.len > pre || s.qbeg < pqb || s.qbeg + s.len > pqe) &&
                        !(s.len - p->seedlen0 > .1 * l_query)) {
                        const int pw = p->w;
                        int qd = s.qbeg - pqb;
                        i64 rd = s.rbeg - prb;
                        int max_gap = cal_max_gap(P.o, qd < rd ? qd : (int)rd);
                        int band = max_gap < pw ? max_gap : pw;
                        if (qd - rd < band && rd - qd < band) hit = true;
                        else {
                            qd = pqe - (s.qbeg + s.len);
                            rd = pre - (s.rbeg + s.len);
                            max_gap = cal_max_gap(P.o, qd < rd ? qd : (int)rd);
                            band = max_gap < pw ? max_gap : pw;
                            if (qd - rd < band && rd - qd < band) hit = true;
                        }
                    }
                }
                found = __ballot(hit) != 0;
            }
            if (!found) continue;
            // (almost) contained -- unless a long, overlapping, better seed of the chain on another diagonal says otherwise
            bool other = false;
            for (int ub = k + 1; ub < ch.n_seeds && !other; ub += 64) {
                const int u = ub + lane;
                bool hit = false;
                if (u < ch.n_seeds && ord[u] >= 0) {
                    const meme_chain_seed t = sd[ord[u]];
                    if (!(t.len < s.len * .95)) {
                        if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) hit = true;
                        else if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) hit = true;
                    }
                }
                other = __ballot(hit) != 0;
            }
            if (!other) { av[cur].qb = -1; av[cur].qe = -1; ord[k] = -1; }
        }
    }
}

// ---- extension in rounds ------------------------------------------------------------------------------------------------------------------
// The reference extends every chained seed of a batch and then drops, read by read and in extension order, the alignments of seeds that an
// earlier SURVIVING alignment of the read covers (:3389-3485); mem_kernel2_core deletes the dropped records at once (src/bwamem.cpp:1680-1693).
// A dropped seed's extension is therefore never looked at, and whether a seed is dropped depends on the survivors before it only: the same
// records come out when a read's seeds are taken in that order, tested first, and extended only if they survive.  k_ext_advance walks a read
// from where it stands: dropped seeds are marked, the first survivor is selected for this round's jobs (ADV_ONE); in the last round every seed
// from the next survivor on is selected at once (ADV_REST: repeat-rich reads have tens of survivors, one round each would be launches without work) and
// ADV_FINISH walks to the end over alignments that then all exist -- which is the reference's own pass from that point on.
enum { ADV_ONE = 0, ADV_REST = 1, ADV_FINISH = 2 };
struct AdvArgs {
    PurgeArgs P; int4* state; uint8_t* sel; uint8_t* act; i64* cntS; int mode;     // cntS: seeds selected per read (summed by a scan: an atomic per read on one address costs 12 ns each)
    const i64* rmax; i64 *cntL, *cntR, *cntB;        // the round's plan as k_ext_jobs<false> would count it: left / right jobs and sequence bytes of the read's selected seeds
    const i64* list; i64 nlist; int light;           // as in ExtArgs: the wavefront-per-read launch walks `list`, the eight-lanes-per-read launch everything with at most EXT_LIGHT seeds
};
// jobs and bytes of one selected seed (the arithmetic of k_ext_jobs)
__device__ __forceinline__ void adv_count(const meme_chain_seed& t, i64 rmax0, i64 rmax1, int l_query, i64& nL, i64& nR, i64& nB) {
    const int qe = t.qbeg + t.len;
    if (t.qbeg > 0) { ++nL; nB += pad4((int)(t.rbeg - rmax0)) + pad4(t.qbeg); }
    if (qe != l_query) { ++nR; nB += pad4((int)(rmax1 - (t.rbeg + t.len))) + pad4(l_query - qe); }
}

template <int G>
__global__ void __launch_bounds__(256) k_ext_advance(AdvArgs V) {
    const PurgeArgs& P = V.P;
    const i64 it = (i64)blockIdx.x * (256 / G) + threadIdx.x / G;
    if (it >= (V.list ? V.nlist : P.nreads)) return;
    const i64 r = V.list ? V.list[it] : it;
    const int lane = threadIdx.x & (G - 1);
    const int gbase = (threadIdx.x & 63) & ~(G - 1);
    const i64 c0 = P.chain_off[r];
    const int nc = (int)(P.chain_off[r + 1] - c0);
    const i64 s0 = P.seed_off[r];
    if (V.light && P.seed_off[r + 1] - s0 > EXT_LIGHT) return;
    int4 st = V.state[r];                              // x chain, y rank of the next seed in it (-1: the chain is done), z seeds met so far
    if (V.mode != ADV_FINISH && lane == 0) { V.act[r] = 0; V.cntL[r] = 0; V.cntR[r] = 0; V.cntB[r] = 0; V.cntS[r] = 0; }
    if (st.x >= nc) return;
    const int l_query = (int)(P.read_off[r + 1] - P.read_off[r]);
    meme_alnreg* av = P.regs + s0;
    int cur = st.z;
    for (int c = st.x; c < nc; ++c) {
        const meme_chain ch = P.chains[c0 + c];
        const meme_chain_seed* sd = P.seeds + s0 + ch.seed_beg;
        int* ord = P.order + s0 + ch.seed_beg;
        for (int k = c == st.x ? st.y : ch.n_seeds - 1; k >= 0; --k, ++cur) {
            const meme_chain_seed s = sd[ord[k]];
            bool found = false;
            for (int ib = 0; ib < cur && !found; ib += G) {
                const int i = ib + lane;
                bool hit = false;
                if (i < cur) {
                    const meme_alnreg* p = &av[i];
                    const i64 prb = p->rb, pre = p->re;
                    const int pqb = p->qb, pqe = p->qe;
                    if (!(pqb == -1 && pqe == -1) && !(s.rbeg < prb || s.rbeg + s.len > pre || s.qbeg < pqb || s.qbeg + s.len > pqe) &&
                        !(s.len - p->seedlen0 > .1 * l_query)) {
                        const int pw = p->w;
                        int qd = s.qbeg - pqb;
                        i64 rd = s.rbeg - prb;
                        int max_gap = cal_max_gap(P.o, qd < rd ? qd : (int)rd);
                        int band = max_gap < pw ? max_gap : pw;
                        if (qd - rd < band && rd - qd < band) hit = true;
                        else {
                            qd = pqe - (s.qbeg + s.len);
                            rd = pre - (s.rbeg + s.len);
                            max_gap = cal_max_gap(P.o, qd < rd ? qd : (int)rd);
                            band = max_gap < pw ? max_gap : pw;
                            if (qd - rd < band && rd - qd < band) hit = true;
                        }
                    }
                }
                found = gballot<G>(hit, gbase) != 0;
            }
            bool drop = false;
            if (found) {
                bool other = false;
                for (int ub = k + 1; ub < ch.n_seeds && !other; ub += G) {
                    const int u = ub + lane;
                    bool hit = false;
                    if (u < ch.n_seeds && ord[u] >= 0) {
                        const meme_chain_seed t = sd[ord[u]];
                        if (!(t.len < s.len * .95)) {
                            if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) hit = true;
                            else if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) hit = true;
                        }
                    }
                    other = gballot<G>(hit, gbase) != 0;
                }
                drop = !other;
            }
            if (drop) { av[cur].qb = -1; av[cur].qe = -1; ord[k] = -1; continue; }
            if (V.mode == ADV_FINISH) continue;        // a survivor whose alignment exists
            if (V.mode == ADV_REST) {                  // this survivor and everything behind it go into the round; ADV_FINISH walks on from here
                unsigned long long cnt = 0;
                i64 nL = 0, nR = 0, nB = 0;
                for (int c2 = c; c2 < nc; ++c2) {
                    const meme_chain ch2 = P.chains[c0 + c2];
                    const int* ord2 = P.order + s0 + ch2.seed_beg;
                    const int k1 = c2 == c ? k : ch2.n_seeds - 1;
                    const i64 rmax0 = V.rmax[2 * (c0 + c2)], rmax1 = V.rmax[2 * (c0 + c2) + 1];
                    for (int k2 = k1 - lane; k2 >= 0; k2 -= G) {
                        const int il = ord2[k2];
                        V.sel[s0 + ch2.seed_beg + il] = 1;
                        adv_count(P.seeds[s0 + ch2.seed_beg + il], rmax0, rmax1, l_query, nL, nR, nB);
                    }
                    cnt += k1 + 1;
                }
                for (int d = G / 2; d >= 1; d >>= 1) { nL += __shfl_xor(nL, d); nR += __shfl_xor(nR, d); nB += __shfl_xor(nB, d); }
                if (lane == 0) { V.act[r] = 1; V.state[r] = make_int4(c, k, cur, 0); V.cntS[r] = (i64)cnt; V.cntL[r] = nL; V.cntR[r] = nR; V.cntB[r] = nB; }
                return;
            }
            if (lane == 0) {                           // this round's seed of the read
                V.sel[s0 + ch.seed_beg + ord[k]] = 1;
                V.act[r] = 1;
                V.state[r] = make_int4(c, k - 1, cur + 1, 0);
                V.cntS[r] = 1;
                i64 nL = 0, nR = 0, nB = 0;
                adv_count(s, V.rmax[2 * (c0 + c)], V.rmax[2 * (c0 + c) + 1], l_query, nL, nR, nB);
                V.cntL[r] = nL; V.cntR[r] = nR; V.cntB[r] = nB;
            }
            return;
        }
        st.y = 0;                                      // (later chains start at their best seed; st.x no longer equals c)
    }
    if (lane == 0) V.state[r] = make_int4(nc, 0, cur, 0);
}

// reads with more than EXT_LIGHT chained seeds: the list the wavefront-per-read launches walk (order is irrelevant: everything is indexed by read)
__global__ void __launch_bounds__(256) k_ext_split(const i64* __restrict__ seed_off, i64 n, i64* __restrict__ list, unsigned long long* __restrict__ cnt) {
    for (i64 r0 = (i64)blockIdx.x * 256; r0 < n; r0 += (i64)gridDim.x * 256) {
        const i64 r = r0 + threadIdx.x;
        const bool heavy = r < n && seed_off[r + 1] - seed_off[r] > EXT_LIGHT;
        const u64 m = __ballot(heavy);
        if (m) {
            const int lane = threadIdx.x & 63;
            unsigned long long base = 0;
            if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(cnt, (unsigned long long)__popcll(m));
            base = __shfl(base, (int)__builtin_ctzll(m));
            if (heavy) list[base + __popcll(m & (((u64)1 << lane) - 1))] = r;
        }
    }
}

unsigned grid_of(i64 items, int per) { i64 b = (items + per - 1) / per; const i64 cap = 256 * 64; return (unsigned)(b < cap ? (b < 1 ? 1 : b) : cap); }

// tuning "ext_live_only": what mem_kernel2_core does first with the stage's records (src/bwamem.cpp:1680-1693: every record with qe <= qb -- the purged
// ones -- is dropped, the others keep their order) done before the records cross to the host: a read's surviving records counted, then packed.
__global__ void __launch_bounds__(256) k_ext_live_count(const i64* __restrict__ seed_off, const meme_alnreg* __restrict__ regs, i64 n, i64* __restrict__ cnt) {
    for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < n; r += (i64)gridDim.x * 256) {
        int m = 0;
        for (i64 i = seed_off[r]; i < seed_off[r + 1]; ++i) m += regs[i].qe > regs[i].qb;
        cnt[r] = m;
    }
}
__global__ void __launch_bounds__(256) k_ext_live_pack(const i64* __restrict__ seed_off, const meme_alnreg* __restrict__ regs, i64 n, const i64* __restrict__ live_off,
                                                       meme_alnreg* __restrict__ out) {
    static_assert(sizeof(meme_alnreg) == 7 * 16, "record copied as seven 16-byte words");
    for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < n; r += (i64)gridDim.x * 256) {
        i64 o = live_off[r];
        if (o == live_off[r + 1]) continue;
        for (i64 i = seed_off[r]; i < seed_off[r + 1]; ++i) {
            if (!(regs[i].qe > regs[i].qb)) continue;
            const uint4* s = (const uint4*)&regs[i];
            uint4* d = (uint4*)&out[o++];
#pragma unroll
            for (int k = 0; k < 7; ++k) d[k] = s[k];
        }
    }
}

// measurement (tuning "ext_census"): jobs whose query equals the first len2 bases of the target (no ambiguous base) -- the jobs a closed form
// could answer without the DP (score = h0 + len2 * a).  SMEM seeds end on a mismatch or at a read end, so few are expected.
__global__ void __launch_bounds__(256) k_ext_census(const meme_seqpair* __restrict__ pairs, i64 n, const uint8_t* __restrict__ seq, int w, unsigned long long* __restrict__ cnt) {
    __shared__ unsigned long long acc[11];
    if (threadIdx.x < 11) acc[threadIdx.x] = 0;
    __syncthreads();
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (i64)gridDim.x * blockDim.x) {
        const meme_seqpair P = pairs[k];
        // [1] cells of the band-limited matrix (rows = target bases, columns = query bases within w of the diagonal)
        unsigned long long cells = 0;
        for (int i = 0; i < P.len1; ++i) {
            const int beg = i > w ? i - w : 0, end = i + w + 1 < P.len2 ? i + w + 1 : P.len2;
            if (end > beg) cells += (unsigned)(end - beg);
        }
        atomicAdd(&acc[1], cells);
        // [2..10] LDS size class of the lane-per-pair kernel (LANE_CLS_Q, meme_bsw.hip)
        const int q = P.len2;
        const int c = q <= 30 ? 0 : q <= 62 ? 1 : q <= 94 ? 2 : q <= 126 ? 3 : q <= 158 ? 4 : q <= 222 ? 5 : q <= 318 ? 6 : q <= 600 ? 7 : 8;
        atomicAdd(&acc[2 + c], 1ull);
        // [0] query == first len2 target bases
        if (P.len2 > P.len1) continue;
        const uint8_t* qq = seq + P.idq;
        const uint8_t* tt = seq + P.idr;
        bool same = true;
        for (int i = 0; same && i < P.len2; ++i) same = qq[i] == tt[i] && qq[i] < 4;
        if (same) atomicAdd(&acc[0], 1ull);
    }
    __syncthreads();
    if (threadIdx.x < 11 && acc[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], acc[threadIdx.x]);
}

}  // namespace

extern "C" int meme_extend_last_batch_host(meme_ctx* ctx, const meme_contig* contigs, int32_t n_contigs, const meme_chain_opt* copt,
                                           const meme_ext_opt* eopt, meme_ext_host_result* out) {
    if (!ctx || !contigs || n_contigs < 1 || !copt || !eopt || !out) { meme_set_error("meme_extend_last_batch_host: null argument"); return MEME_E_ARG; }
    if (copt->max_occ < 1 || copt->l_pac < 1 || eopt->e_del < 1 || eopt->e_ins < 1 || eopt->w < 1) { meme_set_error("meme_extend_last_batch_host: bad options"); return MEME_E_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    memset(out, 0, sizeof(*out));
    const i64 n = ctx->last_seed_reads;
    if (n <= 0 || !ctx->smem_off.p || !ctx->read_off.p || !ctx->reads.p || !ctx->reads_resident) {
        meme_set_error("meme_extend_last_batch_host: no seeded batch on this ctx (a seeding call has to stage the reads; meme_chain_batch_host brings seeds only)");
        return MEME_E_STATE;
    }
    if (copt->l_pac * 2 != ctx->idx.n) { meme_set_error("meme_extend_last_batch_host: l_pac does not match the loaded index"); return MEME_E_ARG; }
    if (ctx->max_batch > 0 && n > ctx->max_batch) { meme_set_error("meme_extend_last_batch_host: %lld reads exceed the ctx's max_batch of %lld", (long long)n, (long long)ctx->max_batch); return MEME_E_CAPACITY; }
    int rc;
    i64 tot[2];
    if ((rc = meme_chain_run(ctx, contigs, n_contigs, copt, tot))) return rc;
    hipEvent_t* ev = ctx->ev_ext;
    for (int i = 0; i < 2; ++i) if (!ev[i]) HIP_TRY(hipEventCreate(&ev[i]));
    HIP_TRY(hipEventRecord(ev[0], ctx->stream));
    DevBuf* B = ctx->chain;
    DevBuf* E = ctx->ext;       // 0 rmax, 1 regs, 2 order, 3 counts + scans, 4 L pairs, 5 R pairs, 6 retry pairs (two halves), 7 sequences, 8 counters,
                                // 9 .. 14 the seed filter's: scores, jobs, counts + new seed offsets, kept seeds, their scores, thresholds; 15, 16 surviving records: counts + scan, packed
    const i64* d_choff = (const i64*)B[5].p;
    const i64* d_sdoff = d_choff + (n + 1);
    const i64 n_chains = tot[0];
    i64 n_seeds = tot[1];
    if ((rc = meme_buf_reserve(ctx, E[8], 256))) return rc;
    // ---- mem_flt_chained_seeds: the read lengths it runs for and their thresholds, evaluated the way the reference's host code does
    // (:579-583: float coefficients, double min_l, libm's log).  Never with reads below ~760 bases unless -W is given.
    const meme_chain_seed* d_seeds = (const meme_chain_seed*)B[7].p;
    const int* d_score = nullptr;
    unsigned long long* d_fltcnt = (unsigned long long*)E[8].p + 2;
    bool flt = false;
    {
        const i64 max_len = ctx->last_seed_max_len;
        std::vector<int> hsp((size_t)max_len + 2, -1);
        for (i64 l = 2; l <= max_len; ++l) {
            const int l_query = (int)l;
            const double min_l = copt->min_chain_weight ? 1.1f * copt->min_chain_weight : 5.5f * log(l_query);     // MEM_HSP_COEF, MEM_MINSC_COEF (:252-253)
            if (min_l > 0.05f * l_query) continue;                                                                  // MEM_SEEDSW_COEF (:254)
            hsp[(size_t)l] = (int)(eopt->a * min_l + .499);
            if (hsp[(size_t)l] < 0) hsp[(size_t)l] = 0;
            flt = true;
        }
        if (flt && n_seeds > 0) {
            if (n_seeds >= 0x7fffffff) { meme_set_error("meme_extend_last_batch_host: %lld chained seeds in one batch", (long long)n_seeds); return MEME_E_CAPACITY; }
            if ((rc = meme_buf_reserve(ctx, E[9], (size_t)(n_seeds + 1) * 4)) || (rc = meme_buf_reserve(ctx, E[10], (size_t)(n_seeds + 1) * sizeof(meme_seedsw_job))) ||
                (rc = meme_buf_reserve(ctx, E[11], (size_t)(n + 1) * 16)) || (rc = meme_buf_reserve(ctx, E[12], (size_t)(n_seeds + 1) * sizeof(meme_chain_seed))) ||
                (rc = meme_buf_reserve(ctx, E[13], (size_t)(n_seeds + 1) * 4)) || (rc = meme_buf_reserve(ctx, E[14], hsp.size() * 4))) return rc;
            HIP_TRY(hipMemcpyAsync(E[14].p, hsp.data(), hsp.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipMemsetAsync(d_fltcnt, 0, 8, ctx->stream));
            FltArgs F;
            memset(&F, 0, sizeof(F));
            F.read_off = (const i64*)ctx->read_off.p; F.n = n; F.chain_off = d_choff; F.chains = (meme_chain*)B[6].p; F.seed_off = d_sdoff; F.seeds = d_seeds;
            F.l_pac = copt->l_pac; F.contig_off = (const i64*)B[4].p; F.contig_len = (const int*)((unsigned char*)B[4].p + (size_t)n_contigs * 8); F.n_contigs = n_contigs;
            F.hsp = (const int*)E[14].p; F.a = eopt->a; F.sc = (int*)E[9].p; F.jobs = (meme_seedsw_job*)E[10].p; F.n_jobs = d_fltcnt;
            F.cnt = (i64*)E[11].p; F.off2 = F.cnt + (n + 1); F.seeds2 = (meme_chain_seed*)E[12].p; F.score2 = (int*)E[13].p;
            hipLaunchKernelGGL(k_flt_pose, dim3((unsigned)n), dim3(64), 0, ctx->stream, F);
            if ((rc = meme_seedsw_launch(ctx, F.jobs, d_fltcnt, n_seeds, F.sc, eopt))) return rc;
            hipLaunchKernelGGL((k_flt_apply<false>), dim3((unsigned)n), dim3(64), 0, ctx->stream, F);
            if ((rc = meme_scan_exclusive(ctx, F.cnt, (i64*)F.off2, n))) return rc;
            hipLaunchKernelGGL((k_flt_apply<true>), dim3((unsigned)n), dim3(64), 0, ctx->stream, F);
            HIP_TRY(hipGetLastError());
            d_sdoff = F.off2; d_seeds = F.seeds2; d_score = F.score2;
        } else flt = false;
    }
    if ((rc = meme_buf_reserve(ctx, E[0], (size_t)(n_chains + 1) * 16)) || (rc = meme_buf_reserve(ctx, E[1], (size_t)(n_seeds + 1) * sizeof(meme_alnreg))) ||
        (rc = meme_buf_reserve(ctx, E[2], (size_t)(n_seeds + 1) * 4)) || (rc = meme_buf_reserve(ctx, E[3], (size_t)(n + 1) * 8 * 8))) return rc;
    ExtArgs A;
    memset(&A, 0, sizeof(A));
    A.reads = (const uint8_t*)ctx->reads.p; A.read_off = (const i64*)ctx->read_off.p; A.g0 = 0; A.ns = n;
    A.chain_off = d_choff; A.chains = (const meme_chain*)B[6].p; A.seed_off = d_sdoff; A.seeds = d_seeds; A.seed_score = d_score; A.frac_rep = (const float*)B[3].p;
    A.pac = ctx->idx.pac; A.l_pac = copt->l_pac;
    A.contig_off = (const i64*)B[4].p; A.contig_len = (const int*)((unsigned char*)B[4].p + (size_t)n_contigs * 8);
    A.o = *eopt;
    A.rmax = (i64*)E[0].p; A.regs = (meme_alnreg*)E[1].p; A.order = (int*)E[2].p;
    i64* d_cnt = (i64*)E[3].p;
    i64* d_off = d_cnt + 3 * (n + 1);
    A.cntL = d_cnt; A.cntR = d_cnt + (n + 1); A.cntB = d_cnt + 2 * (n + 1);
    meme_bsw_opt bl, br;
    memset(&bl, 0, sizeof(bl));
    bl.o_del = eopt->o_del; bl.e_del = eopt->e_del; bl.o_ins = eopt->o_ins; bl.e_ins = eopt->e_ins; bl.zdrop = eopt->zdrop; bl.a = eopt->a; bl.b = eopt->b;
    br = bl;
    bl.end_bonus = eopt->pen_clip5;                   // bswLeft / bswRight, src/bwamem.cpp:2953-2959
    br.end_bonus = eopt->pen_clip3;
    unsigned long long* d_nretry = (unsigned long long*)E[8].p;
    i64* d_cntS = d_cnt + 6 * (n + 1);              // seeds selected per read in a round, and its scan
    i64* d_offS = d_cnt + 7 * (n + 1);
    unsigned long long* d_census = (unsigned long long*)E[8].p + 8;
    if (ctx->ext_census) HIP_TRY(hipMemsetAsync(d_census, 0, 11 * 8, ctx->stream));
    // ---- light and heavy reads (round 6): the list of reads with more than EXT_LIGHT chained seeds; everything else runs eight lanes per read
    i64 n_heavy = 0;
    const i64* d_heavy = nullptr;
    const bool split = ctx->ext_split != 0;
    if (split) {
        if ((rc = meme_buf_reserve(ctx, E[18], (size_t)(n + 1) * 8))) return rc;
        unsigned long long* d_nheavy = (unsigned long long*)E[8].p + 24;
        HIP_TRY(hipMemsetAsync(d_nheavy, 0, 8, ctx->stream));
        hipLaunchKernelGGL(k_ext_split, dim3(grid_of(n, 256)), dim3(256), 0, ctx->stream, d_sdoff, n, (i64*)E[18].p, d_nheavy);
        unsigned long long h = 0;
        HIP_TRY(hipMemcpyAsync(&h, d_nheavy, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        n_heavy = (i64)h; d_heavy = (const i64*)E[18].p;
    }
    // a kernel that walks the reads of [X.g0, X.g0 + X.ns): one launch of wavefronts, or -- split, and the range is the whole batch -- eight lanes per light read + a wavefront per listed read
    auto launch_jobs = [&](ExtArgs X, bool write) {
        const bool two = split && X.g0 == 0 && X.ns == n;
        X.list = nullptr; X.nlist = 0; X.light = 0;
        if (!two) {
            if (write) hipLaunchKernelGGL((k_ext_jobs<true, 64>), dim3((unsigned)((X.ns + 3) / 4)), dim3(256), 0, ctx->stream, X);
            else hipLaunchKernelGGL((k_ext_jobs<false, 64>), dim3((unsigned)((X.ns + 3) / 4)), dim3(256), 0, ctx->stream, X);
            return;
        }
        X.light = 1;
        if (write) hipLaunchKernelGGL((k_ext_jobs<true, 8>), dim3((unsigned)((X.ns + 31) / 32)), dim3(256), 0, ctx->stream, X);
        else hipLaunchKernelGGL((k_ext_jobs<false, 8>), dim3((unsigned)((X.ns + 31) / 32)), dim3(256), 0, ctx->stream, X);
        if (n_heavy > 0) {
            X.light = 0; X.list = d_heavy; X.nlist = n_heavy;
            if (write) hipLaunchKernelGGL((k_ext_jobs<true, 64>), dim3((unsigned)((n_heavy + 3) / 4)), dim3(256), 0, ctx->stream, X);
            else hipLaunchKernelGGL((k_ext_jobs<false, 64>), dim3((unsigned)((n_heavy + 3) / 4)), dim3(256), 0, ctx->stream, X);
        }
    };
    i64 n_pairs = 0, n_retried = 0, n_calls = 0;
    float bsw_ms = 0.f;
    i64 h_flt[2] = {n_seeds, 0};                    // chained seeds after the filter, alignments it ran
    bool flt_read = !flt;
    std::vector<i64> h_off;
    // ---- the jobs of the seeds `A.mode` / `A.sel` name: plan (job and byte counts of every read, their scans), sequences, banded SW left then right
    // with the band doubled once where the reference doubles it, results folded into the records.  *n_sel_out: what k_ext_advance selected for this round.
    auto run_jobs = [&](i64* n_sel_out) -> int {
        int rc;
        if (A.mode != 2) launch_jobs(A, false);      // (in rounds k_ext_advance has counted)
        for (int k = 0; k < 3; ++k) if ((rc = meme_scan_exclusive(ctx, d_cnt + k * (n + 1), d_off + k * (n + 1), n))) return rc;
        i64 tot3[3] = {0, 0, 0};
        for (int k = 0; k < 3; ++k) HIP_TRY(hipMemcpyAsync(&tot3[k], d_off + k * (n + 1) + n, 8, hipMemcpyDeviceToHost, ctx->stream));
        if (n_sel_out) {
            if ((rc = meme_scan_exclusive(ctx, d_cntS, d_offS, n))) return rc;
            HIP_TRY(hipMemcpyAsync(n_sel_out, d_offS + n, 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (!flt_read) {
            HIP_TRY(hipMemcpyAsync(&h_flt[0], d_sdoff + n, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipMemcpyAsync(&h_flt[1], d_fltcnt, 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        flt_read = true;
        // ---- slabs of reads whose jobs' sequences fit 32-bit offsets (SeqPair::idr / idq); the per-read offsets cross to the host only when one slab is not enough
        const i64 SLAB_BYTES = (i64)3 << 29, SLAB_JOBS = 8 << 20;
        const bool one_slab = tot3[2] <= SLAB_BYTES && tot3[0] <= SLAB_JOBS && tot3[1] <= SLAB_JOBS;
        if (tot3[0] + tot3[1] == 0) return MEME_OK;
        if (!one_slab) {
            h_off.resize((size_t)(3 * (n + 1)));
            HIP_TRY(hipMemcpyAsync(h_off.data(), d_off, h_off.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        const i64* oL = one_slab ? nullptr : h_off.data();
        const i64* oR = one_slab ? nullptr : oL + (n + 1);
        const i64* oB = one_slab ? nullptr : oR + (n + 1);
        for (i64 g0 = 0; g0 < n;) {
            i64 g1 = n, nL = tot3[0], nR = tot3[1], nB = tot3[2], j0L = 0, j0R = 0, b0 = 0;
            if (!one_slab) {
                g1 = g0 + 1;
                while (g1 < n && oB[g1 + 1] - oB[g0] <= SLAB_BYTES && oL[g1 + 1] - oL[g0] <= SLAB_JOBS && oR[g1 + 1] - oR[g0] <= SLAB_JOBS) {
                    i64 step = 1;                               // gallop to the slab's end
                    while (g1 + 2 * step < n && oB[g1 + 2 * step + 1] - oB[g0] <= SLAB_BYTES && oL[g1 + 2 * step + 1] - oL[g0] <= SLAB_JOBS &&
                           oR[g1 + 2 * step + 1] - oR[g0] <= SLAB_JOBS) step *= 2;
                    g1 += step;
                }
                if (oB[g1] - oB[g0] >= ((i64)1 << 31)) { meme_set_error("a read's extension jobs need more than 2 GiB of sequence"); return MEME_E_CAPACITY; }
                nL = oL[g1] - oL[g0]; nR = oR[g1] - oR[g0]; nB = oB[g1] - oB[g0];
                j0L = oL[g0]; j0R = oR[g0]; b0 = oB[g0];
            }
            const i64 nmax = nL > nR ? nL : nR;
            if ((rc = meme_buf_reserve(ctx, E[4], (size_t)(nL + 1) * sizeof(meme_seqpair))) || (rc = meme_buf_reserve(ctx, E[5], (size_t)(nR + 1) * sizeof(meme_seqpair))) ||
                (rc = meme_buf_reserve(ctx, E[6], (size_t)(2 * nmax + 2) * sizeof(meme_seqpair))) || (rc = meme_buf_reserve(ctx, E[7], (size_t)nB + 256))) return rc;
            ExtArgs S = A;
            S.g0 = g0; S.ns = g1 - g0;
            // (the scans cover the whole batch: a slab's first read has its own offsets to subtract)
            S.offL = d_off + g0; S.offR = d_off + (n + 1) + g0; S.offB = d_off + 2 * (n + 1) + g0;
            S.job0L = j0L; S.job0R = j0R; S.byte0 = b0;
            S.L = (meme_seqpair*)E[4].p; S.R = (meme_seqpair*)E[5].p; S.seq = (uint8_t*)E[7].p;
            launch_jobs(S, true);
            HIP_TRY(hipGetLastError());
            if (ctx->ext_census) {
                if (nL) hipLaunchKernelGGL(k_ext_census, dim3(grid_of(nL, 256)), dim3(256), 0, ctx->stream, (const meme_seqpair*)S.L, nL, (const uint8_t*)S.seq, eopt->w, d_census);
                if (nR) hipLaunchKernelGGL(k_ext_census, dim3(grid_of(nR, 256)), dim3(256), 0, ctx->stream, (const meme_seqpair*)S.R, nR, (const uint8_t*)S.seq, eopt->w, d_census);
            }
            for (int dir = 0; dir < 2; ++dir) {
                meme_seqpair* P = dir == 0 ? S.L : S.R;
                i64 np = dir == 0 ? nL : nR;
                if (dir == 1 && np > 0) hipLaunchKernelGGL(k_ext_h0, dim3(grid_of(np, 256)), dim3(256), 0, ctx->stream, P, np, (const meme_alnreg*)A.regs);
                for (int attempt = 0; attempt < EXT_BAND_TRIES && np > 0; ++attempt) {
                    const int w = eopt->w << attempt;
                    if ((rc = meme_bsw_launch(ctx, P, S.seq, S.seq, (int)np, w, dir == 0 ? &bl : &br, (int)ctx->last_seed_max_len))) return rc;
                    HIP_TRY(hipMemsetAsync(d_nretry, 0, 8, ctx->stream));
                    FoldArgs F;
                    F.pairs = P; F.n = np; F.regs = A.regs; F.chains = A.chains; F.seed_off = A.seed_off; F.seeds = A.seeds; F.read_off = A.read_off;
                    F.o = *eopt; F.w = w; F.last = attempt + 1 == EXT_BAND_TRIES;
                    F.retry = (meme_seqpair*)E[6].p + (size_t)(attempt & 1) * (size_t)(nmax + 1); F.n_retry = d_nretry;
                    if (dir == 0) hipLaunchKernelGGL((k_ext_fold<true>), dim3(grid_of(np, 256)), dim3(256), 0, ctx->stream, F);
                    else hipLaunchKernelGGL((k_ext_fold<false>), dim3(grid_of(np, 256)), dim3(256), 0, ctx->stream, F);
                    unsigned long long h_retry = 0;
                    HIP_TRY(hipMemcpyAsync(&h_retry, d_nretry, 8, hipMemcpyDeviceToHost, ctx->stream));
                    HIP_TRY(hipStreamSynchronize(ctx->stream));
                    { float ms = 0.f; if (hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]) == hipSuccess) bsw_ms += ms; }
                    n_pairs += np; ++n_calls;
                    if (attempt > 0) n_retried += np;
                    P = F.retry;
                    np = (i64)h_retry;
                }
            }
            g0 = g1;
        }
        return MEME_OK;
    };
    PurgeArgs P;
    P.read_off = A.read_off; P.nreads = n; P.chain_off = A.chain_off; P.chains = A.chains; P.seed_off = A.seed_off; P.seeds = A.seeds;
    P.regs = A.regs; P.order = A.order; P.o = *eopt;
    const i64 rounds = ctx->ext_live_only ? ctx->ext_rounds : 0;
    i64 n_seeds_ext = 0;                                // chained seeds whose extension jobs ran
    if (rounds <= 0) {
        // ---- the reference's batch: every chained seed extended, then the purge
        A.mode = 0;
        if ((rc = run_jobs(nullptr))) return rc;
        hipLaunchKernelGGL(k_ext_purge, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, P);
        HIP_TRY(hipGetLastError());
        n_seeds_ext = h_flt[0];
    } else {
        // ---- in rounds (see k_ext_advance): records + extension order first, then `rounds` rounds of one seed per read, one round with everything still ahead
        if ((rc = meme_buf_reserve(ctx, E[17], (size_t)(n + 1) * 16 + (size_t)(n + 1) + (size_t)n_seeds + 64))) return rc;
        AdvArgs V;
        V.P = P; V.state = (int4*)E[17].p; V.act = (uint8_t*)(V.state + (n + 1)); V.sel = V.act + (n + 1); V.cntS = d_cntS;
        V.rmax = A.rmax; V.cntL = A.cntL; V.cntR = A.cntR; V.cntB = A.cntB;
        A.mode = 1; A.state = V.state;
        launch_jobs(A, true);
        A.sel = V.sel; A.act = V.act;
        auto launch_advance = [&](AdvArgs X) {
            X.list = nullptr; X.nlist = 0; X.light = 0;
            if (!split) { hipLaunchKernelGGL((k_ext_advance<64>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, X); return; }
            X.light = 1;
            hipLaunchKernelGGL((k_ext_advance<8>), dim3((unsigned)((n + 31) / 32)), dim3(256), 0, ctx->stream, X);
            if (n_heavy > 0) { X.light = 0; X.list = d_heavy; X.nlist = n_heavy; hipLaunchKernelGGL((k_ext_advance<64>), dim3((unsigned)((n_heavy + 3) / 4)), dim3(256), 0, ctx->stream, X); }
        };
        for (i64 t = 0; t <= rounds; ++t) {
            HIP_TRY(hipMemsetAsync(V.sel, 0, (size_t)n_seeds, ctx->stream));
            V.mode = t < rounds ? ADV_ONE : ADV_REST;
            launch_advance(V);
            A.mode = 2;
            i64 h_sel = 0;
            if ((rc = run_jobs(&h_sel))) return rc;
            n_seeds_ext += h_sel;
            if (h_sel == 0) break;                      // every read has been walked to its end
        }
        V.mode = ADV_FINISH;
        launch_advance(V);
        HIP_TRY(hipGetLastError());
    }
    const i64 n_flt_dropped = n_seeds - h_flt[0];
    n_seeds = h_flt[0];
    HIP_TRY(hipEventRecord(ev[1], ctx->stream));
    meme_ctx::HostBuf* Hb = ctx->h_ext;
    if ((rc = meme_hostbuf_reserve(ctx, Hb[0], (size_t)(n + 1) * 8))) return rc;
    i64 n_out = n_seeds;                                // records that cross to the host
    if (ctx->ext_live_only) {
        if ((rc = meme_buf_reserve(ctx, E[15], (size_t)(n + 1) * 16))) return rc;
        i64* d_lcnt = (i64*)E[15].p;
        i64* d_loff = d_lcnt + (n + 1);
        hipLaunchKernelGGL(k_ext_live_count, dim3(grid_of(n, 256)), dim3(256), 0, ctx->stream, d_sdoff, (const meme_alnreg*)A.regs, n, d_lcnt);
        if ((rc = meme_scan_exclusive(ctx, d_lcnt, d_loff, n))) return rc;
        HIP_TRY(hipMemcpyAsync(Hb[0].p, d_loff, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        n_out = ((const i64*)Hb[0].p)[n];
        if ((rc = meme_buf_reserve(ctx, E[16], (size_t)(n_out + 1) * sizeof(meme_alnreg))) || (rc = meme_hostbuf_reserve(ctx, Hb[1], (size_t)(n_out + 1) * sizeof(meme_alnreg)))) return rc;
        hipLaunchKernelGGL(k_ext_live_pack, dim3(grid_of(n, 256)), dim3(256), 0, ctx->stream, d_sdoff, (const meme_alnreg*)A.regs, n, (const i64*)d_loff, (meme_alnreg*)E[16].p);
        HIP_TRY(hipGetLastError());
        if (n_out) HIP_TRY(hipMemcpyAsync(Hb[1].p, E[16].p, (size_t)n_out * sizeof(meme_alnreg), hipMemcpyDeviceToHost, ctx->stream));
    } else {
        if ((rc = meme_hostbuf_reserve(ctx, Hb[1], (size_t)(n_seeds + 1) * sizeof(meme_alnreg)))) return rc;
        HIP_TRY(hipMemcpyAsync(Hb[0].p, d_sdoff, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (n_seeds) HIP_TRY(hipMemcpyAsync(Hb[1].p, A.regs, (size_t)n_seeds * sizeof(meme_alnreg), hipMemcpyDeviceToHost, ctx->stream));
    }
    unsigned long long h_census[11] = {0};
    if (ctx->ext_census) HIP_TRY(hipMemcpyAsync(h_census, d_census, 11 * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    out->n_exact_prefix = ctx->ext_census ? (int64_t)h_census[0] : -1;
    out->census_band_cells = ctx->ext_census ? (int64_t)h_census[1] : -1;
    for (int c = 0; c < 9; ++c) out->census_class[c] = ctx->ext_census ? (int64_t)h_census[2 + c] : -1;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
    out->nreads = n; out->reg_off = (const int64_t*)Hb[0].p; out->regs = (const meme_alnreg*)Hb[1].p; out->total_regs = n_out; out->total_seeds = n_seeds; out->n_ext_seeds = n_seeds_ext;
    out->total_chains = n_chains; out->n_pairs = n_pairs; out->n_retried = n_retried; out->n_bsw_calls = n_calls;
    out->n_flt_jobs = h_flt[1]; out->n_flt_dropped = n_flt_dropped;
    out->n_tier2 = ctx->chain_tier2_reads; out->chain_ms = ctx->tm.chain_kernel_ms; out->ext_ms = ms; out->bsw_ms = bsw_ms;
    return MEME_OK;
}
