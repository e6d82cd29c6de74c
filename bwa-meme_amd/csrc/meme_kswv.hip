// Mate-rescue Smith-Waterman of the SAM phase on the device: what mem_sam_pe_batch (reference src/bwamem_pair.cpp:719-818)
// computes with the AVX-512 kswv kernels (src/kswv.cpp:372-712 int8 lanes, :934-1199 int16 lanes) for the jobs mem_matesw_batch_pre
// (src/bwamem_pair.cpp:1060-1223) has posed -- a mate (or its reverse complement) against the window of the reference its partner's
// alignment and the insert-size distribution point at: forward pass (score, end positions, second-best score outside the best hit),
// then, for hits above the threshold, the pass over the reversed prefixes that finds the start.
//
// The reference puts one job in each SIMD lane; so does this kernel, at wavefront width: one job per lane, 64 jobs per wavefront, jobs
// sorted so that a wavefront's jobs have the same padded query length and similar windows.  Per lane the column state {H(i-1,j), F(i,j),
// query code} is one 32-bit LDS word, laid out [column][lane] (conflict-free whatever column a lane is at); E and the diagonal travel
// in registers along the row.  What one SIMD lane of the reference computes does not depend on its neighbours (oracle/meme_oracle.c,
// kswv_pass, spells out why), so every quirk is per-lane arithmetic here too:
//   * int8 jobs (KSW_XBYTE: read length x match score < 250): unsigned bytes biased by `shift`, i.e. H + S capped at 255 - shift; the
//     query padded with a base that scores 0 up to a multiple of 16 columns (8 for int16 jobs), the stripe of bwa's SSE2 ksw_u8 / ksw_i16;
//   * a row's maximum and its first column; the global maximum moves only on a strictly larger one (first row wins ties);
//   * rowMax: the row maximum is kept only for rows that are local peaks of that sequence in the reference's sense (Block I, a two-state
//     mask), reach the KSW_XSUBO threshold and precede the lane's stop (KSW_XSTOP reached; int8: score + shift saturates);
//   * score2 / te2: the largest kept row maximum outside te +- ceil(score / match), first row on ties; int8: 0 means none;
//   * second pass when KSW_XSTART is set and the score reaches the threshold: query[0..qe] and target[0..te] reversed, the target's
//     length unchanged, stop at the first pass's score; tb / qb only if it reaches exactly that score.
// rowMax lives in HBM ([row][lane] per wavefront: one 128-byte line per row), the only global traffic of the row loop besides one
// target base per lane and row.
#include <string.h>
#include <algorithm>
#include <vector>

#include "meme_common.h"

namespace {

constexpr int XBYTE = 0x10000, XSTOP = 0x20000, XSUBO = 0x40000, XSTART = 0x80000;      // src/ksw.h:31-34
constexpr int KSWV_SCORE_LIMIT = 1 << 12;            // H and F fields of the column word
constexpr int N_KSWV_CLS = 6;
constexpr int KSWV_CLS_Q[N_KSWV_CLS] = {64, 128, 160, 256, 384, 528};     // padded query columns per LDS size class: (q + 2) * 256 B per wavefront

struct KswvArgs {
    const meme_kswv_job* jobs;
    const int* order;            // job indices, sorted; this launch: [first, first + count)
    int first, count;
    const uint8_t* ref;
    const uint8_t* qer;
    meme_kswr* out;
    unsigned short* rowmax;      // scratch: per wavefront [row][lane]
    const i64* rm_off;           // first row of each wavefront of this launch in the scratch
    int a, b, o_del, e_del, o_ins, e_ins;
};

typedef __attribute__((address_space(3))) unsigned int* lds_u32;

__device__ __forceinline__ int max3_i32(int a, int b, int c) { int d; asm("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// (f << 12 | h) in bits 8..31 above the low byte of the old word, two instructions
__device__ __forceinline__ unsigned hf_repack(int h, int f, unsigned old) {
    unsigned x, w;
    asm("v_lshl_or_b32 %0, %1, 12, %2" : "=v"(x) : "v"(f), "v"(h));
    asm("v_perm_b32 %0, %1, %2, %3" : "=v"(w) : "v"(x), "v"(old), "s"(0x06050400u));
    return w;
}

struct PassOut { int raw, score, te, qe, score2, te2; };

// one pass of one job per lane.  target(i) = i <= t_rev ? t[t_rev - i] : t[i]; query(j) = q_rev >= 0 ? q[q_rev - j] : q[j].
__device__ __forceinline__ PassOut kswv_pass(lds_u32 W, bool is8, const uint8_t* t, int tlen, int t_rev, const uint8_t* q, int qlen, int q_rev, int xtra, bool want2,
                                             const KswvArgs& A, unsigned short* rm) {
    const int w_match = A.a, w_mismatch = -A.b, w_ambig = -1;
    int mn = w_match < w_mismatch ? w_match : w_mismatch;
    mn = mn < w_ambig ? mn : w_ambig;
    const int shift = is8 ? (256 - (mn & 0xff)) & 0xff : 0;
    int qmax = w_match > w_mismatch ? w_match : w_mismatch;
    qmax = qmax > w_ambig ? qmax : w_ambig;
    const int cap = is8 ? 255 - shift : 0x3fffffff;
    const int none = is8 ? 0 : -1;
    const int quanta = is8 ? (qlen + 15) / 16 * 16 : (qlen + 7) / 8 * 8;
    const int oe_del = A.o_del + A.e_del, oe_ins = A.o_ins + A.e_ins, e_del = A.e_del, e_ins = A.e_ins;
    int v = (xtra & XSUBO) ? (xtra & 0xffff) : 0x10000;
    const bool has_minsc = v <= (is8 ? 255 : 32767);
    const int minsc = v;
    v = (xtra & XSTOP) ? (xtra & 0xffff) : 0x10000;
    const bool has_endsc = v <= (is8 ? 255 : 32767);
    const int endsc = v;
    // column words: word j + 1 = {query code of column j in the low byte (0..3, 4 = N, 5 = padding), H(i-1, j) = 0 in bits 8..19, F = 0 in 20..31}
    for (int j = 0; j < quanta; ++j) {
        unsigned c = 5u;
        if (j < qlen) { c = q[q_rev >= 0 ? q_rev - j : j]; c = c > 4u ? 4u : c; }
        W[(j + 1) * 64] = c;
    }
    int gmax = 0, te = -1, qe = 0, pimax = 0;
    bool mask = false, minsc_ok = false, exited = false;
    int exit_row = -1;
    int tb_next = tlen > 0 ? (int)t[0 <= t_rev ? t_rev : 0] : 4;
    int wave_rows = tlen;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int y = __shfl_xor(wave_rows, d); wave_rows = wave_rows > y ? wave_rows : y; }
    const unsigned mm4 = (unsigned)(w_mismatch & 0xff) * 0x01010101u, am4 = (unsigned)(w_ambig & 0xff) * 0x01010101u;
    const unsigned tab_hi = (unsigned)(w_ambig & 0xff);                        // code 4 (N): ambiguous; code 5 (padding): 0
    for (int i = 0; i < wave_rows; ++i) {
        const bool live = i < tlen && !exited;
        if (!__any(live)) break;
        if (live) {
            const int tb = tb_next;
            if (i + 1 < tlen) tb_next = t[i + 1 <= t_rev ? t_rev - (i + 1) : i + 1];
            // the row's scores as a byte table indexed by the query code (v_perm_b32 picks byte `code` of {tab_hi, tab_lo})
            const unsigned tab_lo = tb > 3 ? am4 : (mm4 & ~(0xffu << (8 * tb))) | ((unsigned)(w_match & 0xff) << (8 * tb));
            int e = 0, diag = 0;
            // Four columns per trip with rotating registers (the next word is requested a cell ahead); the row maximum and its FIRST column as
            // one key (h << 10 | 1023 - column), one accumulator per unrolled position (mk - position, joined below)
#define KSWV_CELL(cur_, nxt_, j_, mk_, jr_)                                                                   \
            {                                                                                                  \
                nxt_ = W[((j_) + 2) * 64];                     /* (one word of slack behind the last column) */  \
                const int hold = (int)((cur_ >> 8) & 0xfffu), f = (int)(cur_ >> 20);                            \
                const int sc = (int)(signed char)(__builtin_amdgcn_perm(tab_hi, tab_lo, cur_) & 0xffu);         \
                int m = diag + sc;                                                                             \
                m = m < cap ? m : cap;                                                                         \
                const int h = max3_i32(m, e, f);               /* (e, f >= 0) */                               \
                const int key = (h << 10) + (jr_);                                                             \
                mk_ = mk_ > key ? mk_ : key;                                                                   \
                e = max3_i32(h - oe_ins, e - e_ins, 0);                                                        \
                const int f2 = max3_i32(h - oe_del, f - e_del, 0);                                             \
                W[((j_) + 1) * 64] = hf_repack(h, f2, cur_);                                                   \
                diag = hold;                                                                                   \
            }
            unsigned wa = W[64], wb = 0u, wc = 0u, wd = 0u;
            int mk0 = -8, mk1 = -8, mk2 = -8, mk3 = -8;
            for (int j = 0; j < quanta; j += 4) {
                const int jr = 1023 - j;
                KSWV_CELL(wa, wb, j, mk0, jr)
                KSWV_CELL(wb, wc, j + 1, mk1, jr)
                KSWV_CELL(wc, wd, j + 2, mk2, jr)
                KSWV_CELL(wd, wa, j + 3, mk3, jr)
            }
#undef KSWV_CELL
            mk1 -= 1; mk2 -= 2; mk3 -= 3;
            const int ka = mk0 > mk1 ? mk0 : mk1, kb = mk2 > mk3 ? mk2 : mk3;
            const int key = ka > kb ? ka : kb;
            const int imax = key < 0 ? 0 : key >> 10;
            const int iqe = 1023 - (key & 1023);
            if (i > 0) {                                                   // Block I (src/kswv.cpp:505-518, :1063-1078)
                const bool msk = imax > pimax || mask;
                rm[(i64)(i - 1) * 64] = (unsigned short)((!msk && minsc_ok) ? pimax : none);
                mask = !msk;
            }
            pimax = imax;
            minsc_ok = has_minsc && imax >= minsc;
            if (imax > gmax) { gmax = imax; te = i; qe = is8 ? (iqe & 0xff) : iqe; }       // Block II
            if ((has_endsc && gmax >= endsc) || (is8 && gmax + shift >= 255)) {
                exited = true;                                             // (this row and all later ones keep nothing)
                exit_row = i;
            }
        }
    }
    if (!exited && tlen > 0) rm[(i64)(tlen - 1) * 64] = (unsigned short)((!mask && minsc_ok) ? pimax : none);
    PassOut R;
    R.raw = gmax;
    R.score = is8 ? (gmax + shift < 255 ? gmax : 255) : gmax;
    R.te = te; R.qe = qe; R.score2 = -1; R.te2 = -1;
    if (want2) {                                                           // (wave-uniform: first passes only)
        const bool on = !(is8 && R.score == 255);
        const int val = (gmax + qmax - 1) / qmax, low = te - val, high = te + val;
        const int last = !on ? -1 : (exited ? exit_row - 1 : tlen - 1);    // rows at and after a stop keep nothing (and were not written)
        int wave_last = last;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const int y = __shfl_xor(wave_last, d); wave_last = wave_last > y ? wave_last : y; }
        int mx = none, t2 = -1;
        __threadfence_block();
        for (int i = 0; i <= wave_last; ++i) {
            const bool in = i <= last && (i < low || i > high);
            if (in) {
                const unsigned short raw = rm[(i64)i * 64];
                const int rv = is8 ? (int)raw : (int)(short)raw;
                if (rv > mx) { mx = rv; t2 = i; }
            }
        }
        if (on) { R.score2 = is8 ? (mx == 0 ? -1 : mx) : mx; R.te2 = t2; }
    }
    return R;
}

__global__ void __launch_bounds__(64) k_kswv(KswvArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned int w_raw[];
    const int lane = threadIdx.x;
    const lds_u32 W = (lds_u32)w_raw + lane;
    const int slot = (int)blockIdx.x * 64 + lane;
    const bool active = slot < A.count;
    const int jid = A.order[A.first + (active ? slot : 0)];
    const meme_kswv_job J = A.jobs[jid];
    const int tlen = active ? J.len1 : 0, qlen = active ? J.len2 : 0, xtra = J.xtra;
    const bool is8 = (xtra & XBYTE) != 0;                                   // sort_classify, src/bwamem.cpp:1798-1825
    const uint8_t* t = A.ref + J.idr;
    const uint8_t* q = A.qer + J.idq;
    unsigned short* rm = A.rowmax + A.rm_off[blockIdx.x] * 64 + lane;
    const PassOut F = kswv_pass(W, is8, t, tlen, -1, q, qlen, -1, xtra, true, A, rm);
    meme_kswr R;
    R.score = F.score; R.te = F.te; R.qe = F.qe; R.score2 = F.score2; R.te2 = F.te2; R.tb = -1; R.qb = -1;
    // second pass (src/bwamem_pair.cpp:768-808): lanes without one run an empty job
    const bool again = active && (xtra & XSTART) != 0 && !((xtra & XSUBO) && R.score < (xtra & 0xffff));
    __builtin_amdgcn_wave_barrier();
    const PassOut B = kswv_pass(W, is8, t, again ? tlen : 0, R.te, q, again ? R.qe + 1 : 0, R.qe, XSTOP | R.score, false, A, rm);
    if (again && R.score == B.raw) { R.tb = R.te - B.te; R.qb = R.qe - B.qe; }
    if (active) A.out[jid] = R;
}

// ---- mem_seed_sw (reference src/bwamem.cpp:494-520): the local alignment score of a chained seed's neighbourhood, at most 199 x 199 bases,
// as ksw_align2 without KSW_XBYTE computes it -- ksw_i16 (src/ksw.cpp:236-320; only .score is used).  One job per lane, the window straight
// from the 2-bit text, the query from the resident read; column word = {query code, H(i-1, j), E(i, j)} as above.
//
// ksw_i16 is Farrar's striped recurrence; its "lazy F" loop raises H where an insertion crosses a lane boundary of the striping but not
// the E the main loop has already stored for the next row (:280-283 vs :289-299), i.e. it loses some insertion -> deletion transitions.
// That is not observable: the same two gaps in the other order (deletion, then insertion: E feeds h, h feeds F, both exact) reach the same
// cell with the same score, so every H equals the plain Gotoh value and the kernel computes that (the oracle restates the striping lane
// by lane and agrees; tests/test_ref_live.py pins it on the compiled reference).  Columns past qlen score 0 in the reference's profile and
// can only repeat values already counted; they are not computed.
//
// The window's bases are the ones the REFERENCE's call reads, which are not the genome's: mem_kernel1_core_Learned is handed worker_t::rc_pac
// as `pac` (src/bwamem.cpp:1770) -- the 2-bit fwd + rc text with the four bases of every BYTE in reverse order, as the learned index's key
// extraction wants them (src/fastmap.cpp:440-457; "BitReverseTable256", src/LearnedIndex_seeding.h:129-137, reverses 2-bit groups) -- and
// bns_get_seq (src/bntseq.cpp:515-539) reads it with the plain _get_pac: within every aligned group of four bases the order is reversed;
// windows on the reverse strand are fetched from the forward half and complemented (:526-531).  SAM output depends on it (under -W), so
// it is reproduced, not corrected.
__device__ __forceinline__ int flt_window_base(const u64* __restrict__ pac, i64 l_pac, i64 p) {
    const i64 k = p < l_pac ? p : (l_pac << 1) - 1 - p;
    const i64 kk = (k & ~3ll) + 3 - (k & 3);
    const int c = kk < l_pac << 1 ? (int)(pac[kk >> 5] >> (62 - 2 * (int)(kk & 31))) & 3 : 0;
    return p < l_pac ? c : 3 - c;
}

__device__ __forceinline__ int seedsw_score(lds_u32 W, const u64* __restrict__ pac, i64 l_pac, i64 rb, int tlen, const uint8_t* __restrict__ q, int qlen, const KswvArgs& A) {
    const int w_match = A.a, w_mismatch = -A.b, w_ambig = -1;
    const int oe_del = A.o_del + A.e_del, oe_ins = A.o_ins + A.e_ins, e_del = A.e_del, e_ins = A.e_ins;
    int wave_q = qlen, wave_rows = tlen;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int y = __shfl_xor(wave_q, d), z = __shfl_xor(wave_rows, d);
        wave_q = wave_q > y ? wave_q : y;
        wave_rows = wave_rows > z ? wave_rows : z;
    }
    for (int j = 0; j <= wave_q; ++j) {                      // (one word of slack behind the last column)
        unsigned c = 5u;
        if (j < qlen) { c = q[j]; c = c > 4u ? 4u : c; }
        W[(j + 1) * 64] = c;
    }
    const unsigned mm4 = (unsigned)(w_mismatch & 0xff) * 0x01010101u;
    const unsigned tab_hi = (unsigned)(w_ambig & 0xff);     // code 4 (N): ambiguous; code 5 (past the query): 0
    int gmax = 0;
    for (int i = 0; i < wave_rows; ++i) {
        if (i < tlen) {
            const int tb = flt_window_base(pac, l_pac, rb + i);
            const unsigned tab_lo = (mm4 & ~(0xffu << (8 * tb))) | ((unsigned)(w_match & 0xff) << (8 * tb));
            int f = 0, diag = 0, rmax = 0;
            unsigned cur = W[64];
            for (int j = 0; j < qlen; ++j) {
                const unsigned nxt = W[(j + 2) * 64];
                const int hold = (int)((cur >> 8) & 0xfffu), e = (int)(cur >> 20);
                const int sc = (int)(signed char)(__builtin_amdgcn_perm(tab_hi, tab_lo, cur) & 0xffu);
                const int h = max3_i32(diag + sc, e, f);        // (:274-277; e, f >= 0: _mm_subs_epu16)
                rmax = rmax > h ? rmax : h;
                const int e2 = max3_i32(h - oe_del, e - e_del, 0);
                f = max3_i32(h - oe_ins, f - e_ins, 0);
                W[(j + 1) * 64] = hf_repack(h, e2, cur);
                diag = hold;
                cur = nxt;
            }
            gmax = gmax > rmax ? gmax : rmax;
        }
    }
    return gmax;
}

__global__ void __launch_bounds__(64) k_seedsw(const meme_seedsw_job* __restrict__ jobs, const unsigned long long* __restrict__ n_jobs, const u64* __restrict__ pac,
                                               i64 l_pac, const uint8_t* __restrict__ reads, int* __restrict__ sc, KswvArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned int w_raw[];
    const int lane = threadIdx.x;
    const lds_u32 W = (lds_u32)w_raw + lane;
    const i64 n = (i64)*n_jobs;
    if ((i64)blockIdx.x * 64 >= n) return;
    const i64 slot = (i64)blockIdx.x * 64 + lane;
    const bool active = slot < n;
    const meme_seedsw_job J = jobs[active ? slot : 0];
    const int score = seedsw_score(W, pac, l_pac, J.rb, active ? (int)J.tlen : 0, reads + J.qoff, active ? (int)J.qlen : 0, A);
    if (active) sc[J.seed] = score;
}

}  // namespace

int meme_seedsw_launch(meme_ctx* ctx, const meme_seedsw_job* d_jobs, const unsigned long long* d_njobs, i64 max_jobs, int* d_sc, const meme_ext_opt* o) {
    if (max_jobs <= 0) return MEME_OK;
    if ((i64)(MEME_SEEDSW_MAX - 1) * o->a >= KSWV_SCORE_LIMIT || o->b > 127 || o->a > 127) {
        meme_set_error("seed filter (mem_flt_chained_seeds): match score %d / mismatch penalty %d beyond the kernel's limits (199 x match < %d)", o->a, o->b, KSWV_SCORE_LIMIT);
        return MEME_E_ARG;
    }
    KswvArgs A;
    memset(&A, 0, sizeof(A));
    A.a = o->a; A.b = o->b; A.o_del = o->o_del; A.e_del = o->e_del; A.o_ins = o->o_ins; A.e_ins = o->e_ins;
    const size_t lds = (size_t)(MEME_SEEDSW_MAX + 2) * 256;
    hipLaunchKernelGGL(k_seedsw, dim3((unsigned)((max_jobs + 63) / 64)), dim3(64), lds, ctx->stream, d_jobs, d_njobs, ctx->idx.pac, ctx->idx.n >> 1, (const uint8_t*)ctx->reads.p, d_sc, A);
    HIP_TRY(hipGetLastError());
    return MEME_OK;
}

extern "C" int meme_kswv_batch_host(meme_ctx* ctx, const meme_kswv_job* jobs, int64_t njobs, const uint8_t* ref, int64_t ref_bytes, const uint8_t* qer,
                                    int64_t qer_bytes, const meme_bsw_opt* opt, meme_kswv_host_result* out) {
    return meme_kswv_run(ctx, jobs, njobs, ref, ref_bytes, qer, qer_bytes, opt, false, out);
}

int meme_kswv_run(meme_ctx* ctx, const meme_kswv_job* jobs, int64_t njobs, const uint8_t* ref, int64_t ref_bytes, const uint8_t* qer, int64_t qer_bytes, const meme_bsw_opt* opt,
                  bool staged, meme_kswv_host_result* out) {
    if (!ctx || !opt || !out || njobs < 0 || (njobs > 0 && (!jobs || (!staged && (!ref || !qer))))) { meme_set_error("meme_kswv_batch_host: null argument"); return MEME_E_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    memset(out, 0, sizeof(*out));
    if (njobs == 0) return MEME_OK;
    if (njobs > 0x7fffffff / 2) { meme_set_error("meme_kswv_batch_host: too many jobs in one call"); return MEME_E_ARG; }
    if (opt->a < 1 || opt->b < 0 || opt->e_del < 0 || opt->e_ins < 0 || opt->o_del < 0 || opt->o_ins < 0) { meme_set_error("meme_kswv_batch_host: bad scoring parameters"); return MEME_E_ARG; }
    const int n = (int)njobs;
    // classes and order: LDS size class, then padded query length, then window length (descending inside a class, so a wavefront's
    // lanes run about the same number of rows)
    std::vector<int> quanta((size_t)n), order((size_t)n);
    for (int k = 0; k < n; ++k) {
        const meme_kswv_job& J = jobs[k];
        const bool is8 = (J.xtra & XBYTE) != 0;
        if (J.len1 < 0 || J.len2 < 0 || J.len1 > 32767 || J.idr < 0 || J.idq < 0 || J.idr + J.len1 > ref_bytes || J.idq + J.len2 > qer_bytes ||
            (int64_t)J.len2 * opt->a >= KSWV_SCORE_LIMIT || J.len2 > KSWV_CLS_Q[N_KSWV_CLS - 1] - 16 || (is8 && J.len2 > 255)) {
            meme_set_error("meme_kswv_batch_host: job %d is malformed or beyond the kernel's limits (window %d at %lld, query %d at %lld, match score %d: "
                           "query x match < %d, query <= %d, window <= 32767)", k, J.len1, (long long)J.idr, J.len2, (long long)J.idq, opt->a, KSWV_SCORE_LIMIT,
                           KSWV_CLS_Q[N_KSWV_CLS - 1] - 16);
            return MEME_E_ARG;
        }
        quanta[(size_t)k] = is8 ? (J.len2 + 15) / 16 * 16 : (J.len2 + 7) / 8 * 8;
        order[(size_t)k] = k;
    }
    auto cls_of = [&](int qn) { int c = 0; while (KSWV_CLS_Q[c] < qn) ++c; return c; };
    std::sort(order.begin(), order.end(), [&](int x, int y) {
        const int cx = cls_of(quanta[(size_t)x]), cy = cls_of(quanta[(size_t)y]);
        if (cx != cy) return cx < cy;
        if (quanta[(size_t)x] != quanta[(size_t)y]) return quanta[(size_t)x] < quanta[(size_t)y];
        if (jobs[x].len1 != jobs[y].len1) return jobs[x].len1 > jobs[y].len1;
        return x < y;
    });
    // launches: one per class; wavefront w of a launch keeps its row maxima at rm_off[w]
    struct Launch { int first, count, qmax; size_t w0; };
    std::vector<Launch> launches;
    std::vector<i64> rm_off;
    i64 rows_total = 0;
    for (int p = 0; p < n;) {
        const int c = cls_of(quanta[(size_t)order[(size_t)p]]);
        int e = p;
        while (e < n && cls_of(quanta[(size_t)order[(size_t)e]]) == c) ++e;
        Launch L;
        L.first = p; L.count = e - p; L.qmax = KSWV_CLS_Q[c]; L.w0 = rm_off.size();
        for (int w = p; w < e; w += 64) {
            int mr = 1;
            for (int k = w; k < e && k < w + 64; ++k) mr = std::max(mr, jobs[order[(size_t)k]].len1);
            rm_off.push_back(rows_total);
            rows_total += mr + 1;
        }
        launches.push_back(L);
        p = e;
    }
    int rc;
    DevBuf* K = ctx->kswv;      // 0 jobs, 1 order, 2 ref bytes, 3 query bytes, 4 results, 5 row maxima, 6 wavefront offsets
    if ((rc = meme_buf_reserve(ctx, K[0], (size_t)n * sizeof(meme_kswv_job))) || (rc = meme_buf_reserve(ctx, K[1], (size_t)n * 4)) ||
        (rc = meme_buf_reserve(ctx, K[2], (size_t)ref_bytes + 64)) || (rc = meme_buf_reserve(ctx, K[3], (size_t)qer_bytes + 64)) ||
        (rc = meme_buf_reserve(ctx, K[4], (size_t)n * sizeof(meme_kswr))) || (rc = meme_buf_reserve(ctx, K[5], (size_t)rows_total * 128 + 256)) ||
        (rc = meme_buf_reserve(ctx, K[6], rm_off.size() * 8 + 8))) return rc;
    hipEvent_t* ev = ctx->ev_kswv;
    for (int i = 0; i < 2; ++i) if (!ev[i]) HIP_TRY(hipEventCreate(&ev[i]));
    if (!staged) HIP_TRY(hipMemcpyAsync(K[0].p, jobs, (size_t)n * sizeof(meme_kswv_job), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(K[1].p, order.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (!staged) {                                     // (staged: the caller's kernels have written the jobs and both sequence buffers where they are read)
        HIP_TRY(hipMemcpyAsync(K[2].p, ref, (size_t)ref_bytes, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(K[3].p, qer, (size_t)qer_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(hipMemcpyAsync(K[6].p, rm_off.data(), rm_off.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipEventRecord(ev[0], ctx->stream));
    for (const Launch& L : launches) {
        KswvArgs A;
        A.jobs = (const meme_kswv_job*)K[0].p; A.order = (const int*)K[1].p; A.first = L.first; A.count = L.count;
        A.ref = (const uint8_t*)K[2].p; A.qer = (const uint8_t*)K[3].p; A.out = (meme_kswr*)K[4].p;
        A.rowmax = (unsigned short*)K[5].p; A.rm_off = (const i64*)K[6].p + L.w0;
        A.a = opt->a; A.b = opt->b; A.o_del = opt->o_del; A.e_del = opt->e_del; A.o_ins = opt->o_ins; A.e_ins = opt->e_ins;
        const size_t lds = (size_t)(L.qmax + 2) * 256;
        if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_kswv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_kswv, dim3((unsigned)((L.count + 63) / 64)), dim3(64), lds, ctx->stream, A);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev[1], ctx->stream));
    meme_ctx::HostBuf& Hb = ctx->h_kswv;
    if ((rc = meme_hostbuf_reserve(ctx, Hb, (size_t)n * sizeof(meme_kswr)))) return rc;
    HIP_TRY(hipMemcpyAsync(Hb.p, K[4].p, (size_t)n * sizeof(meme_kswr), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
    out->njobs = njobs; out->res = (const meme_kswr*)Hb.p; out->kernel_ms = ms;
    return MEME_OK;
}
