// Banded Smith-Waterman seed extension on MI355X.
//
// Replaces BandedPairWiseSW::getScores8 / getScores16 / scalarBandedSWAWrapper (reference
// src/bandedSWA.cpp:242-260, 1970-2261, 2664-2961), whose common semantics are scalarBandedSWA
// (src/bandedSWA.cpp:116-237) == ksw_extend2 (src/ksw.cpp:434-535).
//
// Two kernels, one semantics:
//  * k_bsw_lane -- one pair per lane (64 pairs per wavefront, pairs counting-sorted by query length so the lanes finish
//    together): the inter-task parallelism of the reference's own SIMD code at wavefront width.  Takes every pair whose
//    query fits the LDS classes (<= 600 bases) and whose scores fit 12 bits -- all short-read extensions.
//  * k_bsw<LP>  -- 16 / 32 / 64 lanes per pair (by query length) for the rest (long queries, huge h0) and for small
//    batches, where latency matters: a lone pair takes milliseconds on one lane (see bsw_lane_min_pairs).
//
// Why row-synchronous and not anti-diagonal (k_bsw): the function's observable behaviour is defined row by row
// -- the band [beg,end) of row i+1 is trimmed from the zero runs of row i (:217-221), the z-drop / m==0
// exits (:206-216) and the max/gscore tie rules (:188-189, :202-205) are evaluated per row, and the
// H/E array keeps *stale* cells outside the band that are read again when the band re-grows.  An
// anti-diagonal sweep would have to replay those row-granular decisions anyway.  Instead each row is
// computed across the LP lanes that own a pair: the only in-row dependency, the F (insertion) chain
//   F(i,j+1) = max(0, M(i,j)-oe_ins, F(i,j)-e_ins)
// is a max-plus prefix scan, F(i,j) = max(0, max_{k<j}(M(i,k) + k*e_ins) - oe_ins - (j-1)*e_ins),
// done with sub-wavefront shuffles.  H/E rows and the query live in LDS for the whole pair; the
// reference window streams through registers.  Integer max-plus recurrences: MFMA does not apply.
#include <limits.h>
#include <string.h>
#include <stdlib.h>
#include <chrono>

#include "meme_common.h"

namespace {

constexpr int BSW_BLOCK = 256;            // 4 wavefronts = 4..16 pairs in flight per workgroup
constexpr int NEG = -(1 << 29);

struct BswArgs {
    meme_seqpair* pairs;
    const uint8_t* ref;
    const uint8_t* qer;
    const int* order;        // pair indices of this length class
    int npairs;              // pairs in this class
    int w;
    meme_bsw_opt o;
    int qmax;                // LDS columns per pair (multiple of LP)
    unsigned int* ticket;
    const int* offs;         // != nullptr: the class is order[offs[key_first] .. offs[key_last]) (exclusive scan of the sort keys,
    int key_first, key_last; // read on the device: no host round trip between the sort and the DP kernels)
    unsigned char* gws;      // != nullptr: H/E rows and the staged query live here (queries too long for LDS), per_grp bytes per group
};

// LP lanes cooperate on one pair (64/LP pairs per wavefront): short extensions -- the common case, since most
// reads carry one long SMEM -- would leave most of a 64-lane wavefront idle.
// Cross-lane primitives of the lanes-per-pair kernel as DPP / readlane operations (register file only): the ds_bpermute
// shuffles they replace each cost an LDS round trip on the row's dependent chain (~20 per row).
#define BSW_DPP(old_, src_, ctrl_, rowmask_) __builtin_amdgcn_update_dpp((old_), (src_), (ctrl_), (rowmask_), 0xF, false)

// inclusive max-scan over the LP lanes of a group (identity NEG): row_shr 1/2/4/8, then row_bcast15 / row_bcast31
template <int LP>
__device__ __forceinline__ int grp_incl_max(int v) {
    int y;
    y = BSW_DPP(NEG, v, 0x111, 0xF); v = v > y ? v : y;
    y = BSW_DPP(NEG, v, 0x112, 0xF); v = v > y ? v : y;
    y = BSW_DPP(NEG, v, 0x114, 0xF); v = v > y ? v : y;
    y = BSW_DPP(NEG, v, 0x118, 0xF); v = v > y ? v : y;
    if constexpr (LP >= 32) { y = BSW_DPP(NEG, v, 0x142, 0xA); v = v > y ? v : y; }   // rows 1,3 <- lane 15 of rows 0,2
    if constexpr (LP >= 64) { y = BSW_DPP(NEG, v, 0x143, 0xC); v = v > y ? v : y; }   // rows 2,3 <- lane 31
    return v;
}

// value of the previous lane (lane 0 of the wavefront gets `first`; group heads are overridden by the caller)
__device__ __forceinline__ int lane_shr1(int v, int first) { return BSW_DPP(first, v, 0x138, 0xF); }   // wave_shr:1

// value held by lane `k` of the caller's group, k uniform over the wavefront
template <int LP>
__device__ __forceinline__ int grp_bcast(int v, int k, int lane) {
    k = __builtin_amdgcn_readfirstlane(k);
    if constexpr (LP == 64) return __builtin_amdgcn_readlane(v, k);
    else if constexpr (LP == 32) {
        const int a = __builtin_amdgcn_readlane(v, k), b = __builtin_amdgcn_readlane(v, 32 + k);
        return lane < 32 ? a : b;
    } else {
        const int a = __builtin_amdgcn_readlane(v, k), b = __builtin_amdgcn_readlane(v, 16 + k);
        const int c = __builtin_amdgcn_readlane(v, 32 + k), d = __builtin_amdgcn_readlane(v, 48 + k);
        return lane < 32 ? (lane < 16 ? a : b) : (lane < 48 ? c : d);
    }
}

template <int LP>
__global__ void __launch_bounds__(BSW_BLOCK) k_bsw(BswArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr u64 GMASK = LP == 64 ? ~0ull : ((1ull << LP) - 1ull);
    const int lane = threadIdx.x & 63;
    const int gl = lane & (LP - 1);              // lane within the pair's group
    const int gbase = lane - gl;
    const int gid = threadIdx.x / LP;            // group within the workgroup
    const size_t per_grp = (size_t)(A.qmax + 2) * 8 + (size_t)((A.qmax + 7) & ~7);
    // queries of up to ~4.5 k bases keep their rows in LDS; longer ones (the reference's 16-bit class reaches 32 k) in HBM
    unsigned char* rows = A.gws ? A.gws + per_grp * ((size_t)blockIdx.x * (BSW_BLOCK / LP) + gid) : lds_raw + per_grp * gid;
    int* H = reinterpret_cast<int*>(rows);
    int* E = H + (A.qmax + 2);
    uint8_t* Q = reinterpret_cast<uint8_t*>(E + (A.qmax + 2));
    const int o_del = A.o.o_del, e_del = A.o.e_del, o_ins = A.o.o_ins, e_ins = A.o.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins, zdrop = A.o.zdrop;
    const int sa = A.o.a, sb = -A.o.b;

    if (A.offs) { const int f = A.offs[A.key_first]; A.order += f; A.npairs = A.offs[A.key_last] - f; }
    for (;;) {
        unsigned int tk = 0;
        if (gl == 0) tk = atomicAdd(A.ticket, 1u);
        tk = __shfl(tk, 0, LP);
        if (tk >= (unsigned)A.npairs) break;
        meme_seqpair* P = &A.pairs[A.order ? A.order[tk] : (int)tk];
        const int qlen = P->len2, tlen = P->len1, h0 = P->h0;
        const uint8_t* query = A.qer + P->idq;
        const uint8_t* target = A.ref + P->idr;
        if (qlen > A.qmax || qlen < 0 || tlen < 0) {      // cannot happen: the host sizes qmax per class
            if (gl == 0) { P->score = INT_MIN; P->qle = P->tle = P->gtle = P->gscore = P->max_off = -1; }
            continue;
        }
        // ---- first row (:143-145) and query staging ---------------------------------------------------
        for (int j = gl; j <= qlen + 1; j += LP) {
            int v = 0;
            if (j == 0) v = h0;
            else if (j <= qlen) {
                // eh[1] = max(h0-oe_ins,0); eh[j] = eh[j-1]-e_ins while eh[j-1] > e_ins
                int first = h0 > oe_ins ? h0 - oe_ins : 0;
                int x = first - (j - 1) * e_ins;
                int prev = first - (j - 2) * e_ins;
                v = (j == 1) ? first : (prev > e_ins ? x : 0);
            }
            H[j] = v;
            E[j] = 0;
            if (j < qlen) Q[j] = query[j];
        }
        // band cap (:148-156)
        int w = A.w;
        {
            int mx = sa > 0 ? sa : 0;
            int max_ins = (int)((double)(qlen * mx + A.o.end_bonus - o_ins) / e_ins + 1.);
            if (max_ins < 1) max_ins = 1;
            if (w > max_ins) w = max_ins;
            int max_del = (int)((double)(qlen * mx + A.o.end_bonus - o_del) / e_del + 1.);
            if (max_del < 1) max_del = 1;
            if (w > max_del) w = max_del;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
        int beg = 0, end = qlen;
        int tchunk = 0;                 // lane k of the group holds target[LP*(i/LP) + k]
        for (int i = 0; i < tlen; ++i) {
            if ((i & (LP - 1)) == 0) tchunk = (i + gl < tlen) ? target[i + gl] : 4;
            // the row index is per group (groups of one wavefront may be at different rows after a divergent exit)
            const int tb = __builtin_amdgcn_ds_bpermute((gbase + (i & (LP - 1))) << 2, tchunk);
            if (beg < i - w) beg = i - w;
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            int h1;
            if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
            else h1 = 0;
            int carry_g = NEG;          // running max of M(k) - oe_ins + k*e_ins over finished chunks
            int left_h = h1;            // H(i, j0-1) for the first column of the chunk
            int bh = -1, bj = -1;       // this lane's best cell of the row: score, rightmost column among equal scores
            for (int j0 = beg; j0 < end; j0 += LP) {
                const int j = j0 + gl;
                const bool act = j < end;
                int M = NEG, e = 0;
                if (act) {
                    M = H[j];
                    e = E[j];
                    const int qb = Q[j];
                    const int sc = (tb > 3 || qb > 3) ? -1 : (tb == qb ? sa : sb);
                    M = M ? M + sc : 0;   // :184
                }
                int g = act ? M - oe_ins + j * e_ins : NEG;
                int gi = grp_incl_max<LP>(g);
                int gx = lane_shr1(gi, NEG);
                if (gl == 0) gx = NEG;
                gx = gx > carry_g ? gx : carry_g;
                int f = gx - (j - 1) * e_ins;
                if (f < 0 || gx == NEG) f = 0;
                int h = M > e ? M : e;
                h = h > f ? h : f;
                int hl = lane_shr1(h, 0);
                if (gl == 0) hl = left_h;
                if (act) {
                    H[j] = hl;                             // H(i,j-1) for the next row (:183)
                    int t = M - oe_del;
                    t = t > 0 ? t : 0;
                    e -= e_del;
                    e = e > t ? e : t;
                    E[j] = e;                              // E(i+1,j) (:190-194)
                    if (h >= bh) { bh = h; bj = j; }
                }
                const int nact = end - j0 < LP ? end - j0 : LP;
                left_h = __shfl(h, nact - 1, LP);
                int cg = grp_bcast<LP>(gi, LP - 1, lane);
                carry_g = carry_g > cg ? carry_g : cg;
            }
            // row maximum and its rightmost column: two max-scans, each total read off the group's last lane
            int m = grp_bcast<LP>(grp_incl_max<LP>(bh), LP - 1, lane), mj = -1;
            if (m < 0) m = 0;
            else mj = grp_bcast<LP>(grp_incl_max<LP>(bh == m ? bj : -1), LP - 1, lane);
            h1 = left_h;
            if (gl == 0) { H[end] = h1; E[end] = 0; }      // :201
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if ((beg < end ? end : beg) == qlen) {          // "if (j == qlen)" after the column loop, :202-205
                max_ie = gscore > h1 ? max_ie : i;
                gscore = gscore > h1 ? gscore : h1;
            }
            if (m == 0) break;                             // :206
            if (m > max) {
                max = m; max_i = i; max_j = mj;
                int off = mj - i;
                if (off < 0) off = -off;
                max_off = max_off > off ? max_off : off;
            } else if (zdrop > 0) {
                if (i - max_i > mj - max_j) {
                    if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
                } else {
                    if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
                }
            }
            // band trimming (:217-221): drop leading / trailing columns whose H and E are both zero
            int nbeg = end;
            for (int j0 = beg; j0 < end; j0 += LP) {
                const int j = j0 + gl;
                bool nz = j < end && ((H[j] | E[j]) != 0);
                u64 b = (__ballot(nz) >> gbase) & GMASK;
                if (b) { nbeg = j0 + __ffsll((long long)b) - 1; break; }
            }
            int jl = nbeg - 1;
            for (int hi = end; hi >= nbeg; hi -= LP) {
                const int j = hi - (LP - 1) + gl;
                bool nz = j >= nbeg && j <= hi && ((H[j] | E[j]) != 0);
                u64 b = (__ballot(nz) >> gbase) & GMASK;
                if (b) { jl = hi - (LP - 1) + (63 - __clzll((long long)b)); break; }
            }
            beg = nbeg;
            end = jl + 2 < qlen ? jl + 2 : qlen;
        }
        if (gl == 0) {
            P->score = max;
            P->qle = max_j + 1;
            P->tle = max_i + 1;
            P->gtle = max_ie + 1;
            P->gscore = gscore;
            P->max_off = max_off;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- lane-per-pair kernel ---------------------------------------------------------------------------------------------
// The same inter-task parallelism the reference's SIMD code uses (one pair per SIMD lane, pairs sorted by length so that
// the lanes of a vector finish together: sortPairsLenExt, src/bwamem.cpp:2430-2527), at wavefront width: each of the 64
// lanes runs scalarBandedSWA on its own pair.  No cross-lane traffic at all; per DP cell a lane issues one LDS read,
// one LDS write and ~16 integer ops.  The row state {H(i-1,j-1), E(i,j)} and the query base of column j share one
// 32-bit LDS word (query code in the low byte, then 12 + 12 bits), laid out [column][lane] so every access is bank-conflict free whatever column
// each lane is at.  Pairs whose scores could exceed 12 bits or whose query exceeds LANE_QMAX go to the
// lanes-per-pair kernel above.
constexpr int LANE_QMAX = 600;
constexpr int LANE_SCORE_LIMIT = 1 << 12;
constexpr int N_LANE_CLS = 8;                                        // LDS = (q + 2) * 256 B per wavefront: 8 KB ... 150 KB
constexpr int LANE_CLS_Q[N_LANE_CLS] = {30, 62, 94, 126, 158, 222, 318, LANE_QMAX};
constexpr int TL_BUCKETS = 8;                                        // sub-key: (target length - query length) / 8
constexpr int SORT_KEYS = (LANE_QMAX + 1) * TL_BUCKETS + 1;          // key = query length * 8 + sub-key; last key = the rest
#define QKEY(q_) ((q_) * TL_BUCKETS)

struct LaneArgs {
    meme_seqpair* pairs;
    const uint8_t* ref;
    const uint8_t* qer;
    const int* order;        // pair indices sorted by query length; this launch covers [first, first + count)
    int first, count;
    int w;
    meme_bsw_opt o;
    unsigned int* ticket;
    const int* offs;         // != nullptr: first/count come from the device-side scan, keys [key_first, key_last)
    int key_first, key_last;
};

__device__ __forceinline__ unsigned he_pack(int h, int e, unsigned qbits) { return ((unsigned)h << 8) | ((unsigned)e << 20) | qbits; }
// the same for the inner loop, two instructions: (e << 12 | h), then its three low bytes above the low byte of the old word
__device__ __forceinline__ unsigned he_repack(int h, int e, unsigned old) {
    unsigned x, w;
    asm("v_lshl_or_b32 %0, %1, 12, %2" : "=v"(x) : "v"(e), "v"(h));
    asm("v_perm_b32 %0, %1, %2, %3" : "=v"(w) : "v"(x), "v"(old), "s"(0x06050400u));
    return w;
}
__device__ __forceinline__ int max3_i32(int a, int b, int c) { int d; asm("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

__global__ void __launch_bounds__(64) k_bsw_lane(LaneArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned int he_raw[];
    typedef __attribute__((address_space(3))) unsigned int* lds_u32;
    const int lane = threadIdx.x;
    const lds_u32 he = (lds_u32)he_raw + lane;           // column j of this lane's pair: he[j * 64]
    const int o_del = A.o.o_del, e_del = A.o.e_del, o_ins = A.o.o_ins, e_ins = A.o.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins, zdrop = A.o.zdrop;
    const int sa = A.o.a, sb = -A.o.b;
    constexpr unsigned HE_MASK = 0xffffff00u, QMASK = 0xffu;
    if (A.offs) { A.first = A.offs[A.key_first]; A.count = A.offs[A.key_last] - A.first; }
    for (;;) {
        unsigned int tk = 0;
        if (lane == 0) tk = atomicAdd(A.ticket, 64u);
        tk = __shfl(tk, 0);
        if (tk >= (unsigned)A.count) break;
        const bool active = tk + lane < (unsigned)A.count;
        meme_seqpair* P = &A.pairs[A.order[A.first + (active ? tk + lane : tk)]];
        const int qlen = active ? P->len2 : 0, tlen = active ? P->len1 : 0, h0 = P->h0;
        const uint8_t* query = A.qer + P->idq;
        const uint8_t* target = A.ref + P->idr;
        // ---- first row (:143-145) with the query bases folded in -----------------------------------------
        {
            const int first = h0 > oe_ins ? h0 - oe_ins : 0;
            // query bytes in batches of 8 independent loads (a lane's bytes are consecutive: same cache line)
            for (int j0 = 0; j0 <= qlen; j0 += 8) {
                unsigned qb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) qb[u] = j0 + u < qlen ? (unsigned)query[j0 + u] : 0u;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u;
                    if (j > qlen) break;
                    int v;
                    if (j == 0) v = h0;
                    else if (j == 1) v = first;
                    else { const int prev = first - (j - 2) * e_ins; v = prev > e_ins ? prev - e_ins : 0; }
                    he[j * 64] = he_pack(v, 0, qb[u] > 4u ? 4u : qb[u]);
                }
            }
        }
        // band cap (:148-156)
        int w = A.w;
        {
            const int mx = sa > 0 ? sa : 0;
            int max_ins = (int)((double)(qlen * mx + A.o.end_bonus - o_ins) / e_ins + 1.);
            if (max_ins < 1) max_ins = 1;
            if (w > max_ins) w = max_ins;
            int max_del = (int)((double)(qlen * mx + A.o.end_bonus - o_del) / e_del + 1.);
            if (max_del < 1) max_del = 1;
            if (w > max_del) w = max_del;
        }
        int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
        int beg = 0, end = qlen;
        int tb_next = tlen > 0 ? (int)target[0] : 4;
        for (int i = 0; i < tlen; ++i) {
            const int tb = tb_next;
            if (i + 1 < tlen) tb_next = target[i + 1];        // in flight during the row
            int f = 0, h1, m = 0, mj = -1;
            if (beg < i - w) beg = i - w;
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
            else h1 = 0;
            // the row's substitution scores as a byte table indexed by the query code (v_perm_b32 picks byte q of {hi, lo}): codes 0..3 in
            // tab_lo (the target base's own slot holds the match score), code 4 = N in byte 0 of tab_hi; a row on an N scores -1 throughout
            const unsigned sb4 = (unsigned)(sb & 0xff) * 0x01010101u;
            const unsigned tab_lo = tb > 3 ? 0xffffffffu : (sb4 & ~(0xffu << (8 * tb))) | ((unsigned)(sa & 0xff) << (8 * tb));
            const unsigned tab_hi = 0xffffffffu;
            // Four cells per trip with rotating registers for the column state: the next column's LDS word is
            // requested before the current cell is computed and first touched a whole cell later (a single rotating
            // register made the compiler wait for the prefetch in the middle of the cell).  The row maximum and its rightmost
            // column travel as one key (h << 10 | column), one accumulator per unrolled position (mk_ + its position, joined below).
#define BSW_CELL(cur_, nxt_, j_, mk_, jb_)                                                                    \
            {                                                                                                  \
                nxt_ = he[((j_) + 1) * 64];                    /* (j+1 <= qlen: inside the row) */              \
                int M = (int)((cur_ >> 8) & 0xfffu), e = (int)(cur_ >> 20);                                     \
                /* (selector byte 0 = the query code; the other result bytes are not looked at) */             \
                const int sc = (int)(signed char)(__builtin_amdgcn_perm(tab_hi, tab_lo, cur_) & 0xffu);         \
                M = M ? M + sc : 0;                            /* :184 */                                      \
                const int h = max3_i32(M, e, f);                                                               \
                const int key = (h << 10) + (jb_);                                                             \
                mk_ = mk_ > key ? mk_ : key;                   /* rightmost column among equal maxima */       \
                e = max3_i32(M - oe_del, e - e_del, 0);        /* E(i+1,j) (:190-194) */                       \
                he[(j_) * 64] = he_repack(h1, e, cur_);        /* H(i,j-1) for the next row (:183) */          \
                h1 = h;                                                                                        \
                f = max3_i32(M - oe_ins, f - e_ins, 0);        /* F(i,j+1) (:195-198) */                       \
            }
            unsigned wa = beg < end ? he[beg * 64] : 0u, wb = 0u, wc = 0u, wd = 0u;
            int mk0 = -8, mk1 = -8, mk2 = -8, mk3 = -8;
            int j = beg;
            for (; j + 3 < end; j += 4) {
                BSW_CELL(wa, wb, j, mk0, j)
                BSW_CELL(wb, wc, j + 1, mk1, j)
                BSW_CELL(wc, wd, j + 2, mk2, j)
                BSW_CELL(wd, wa, j + 3, mk3, j)
            }
            for (; j < end; ++j) {
                BSW_CELL(wa, wb, j, mk0, j)
                wa = wb;
            }
#undef BSW_CELL
            {
                mk1 += 1; mk2 += 2; mk3 += 3;
                const int ka = mk0 > mk1 ? mk0 : mk1, kb = mk2 > mk3 ? mk2 : mk3;
                const int key = ka > kb ? ka : kb;
                m = key < 0 ? 0 : key >> 10;
                mj = key < 0 ? -1 : key & 1023;
            }
            he[end * 64] = he_pack(h1, 0, he[end * 64] & QMASK);    // :201
            const unsigned w_front = he[beg * 64];             // (for the band trimming below: requested here, used after the row's bookkeeping)
            if ((beg < end ? end : beg) == qlen) {             // "if (j == qlen)" after the column loop, :202-205
                max_ie = gscore > h1 ? max_ie : i;
                gscore = gscore > h1 ? gscore : h1;
            }
            if (m == 0) break;                                 // :206
            if (m > max) {
                max = m; max_i = i; max_j = mj;
                int off = mj - i;
                if (off < 0) off = -off;
                max_off = max_off > off ? max_off : off;
            } else if (zdrop > 0) {
                if (i - max_i > mj - max_j) {
                    if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
                } else {
                    if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
                }
            }
            // band trimming (:217-221): drop leading / trailing columns whose H and E are both zero
            int jt = beg;
            if (jt < end && (w_front & HE_MASK) == 0u) {
                ++jt;
                while (jt < end && (he[jt * 64] & HE_MASK) == 0u) ++jt;
            }
            beg = jt;
            jt = end;
            if (h1 == 0) {                                     // (column `end` was just written as {h1, 0}: only a zero there asks for the scan)
                --jt;
                while (jt >= beg && (he[jt * 64] & HE_MASK) == 0u) --jt;
            }
            end = jt + 2 < qlen ? jt + 2 : qlen;
        }
        if (active) {
            P->score = max;
            P->qle = max_j + 1;
            P->tle = max_i + 1;
            P->gtle = max_ie + 1;
            P->gscore = gscore;
            P->max_off = max_off;
        }
    }
}

// ---- lane-per-pair kernel with a CIRCULAR band buffer (round 6) -----------------------------------------------------------------
// Row i of the function touches columns [beg, end] with i - w <= beg and end <= i + w + 1 (:151-156, :171-175): at most 2w + 2 columns, and
// beg never decreases.  The linear kernel above keeps a word for every column of the query -- (q + 2) * 256 bytes of LDS per wavefront: 57 KB
// for queries up to 222 bases (two wavefronts per CU), 82 KB up to 318 and 154 KB up to 600 (ONE per CU): the classes that hold 36 % of the
// extension jobs of 250-bp reads.  Here column j lives in slot j mod C of a ring of C >= 2w + 2 columns (C = 208 for w = 100: 53 KB, three
// wavefronts per CU whatever the query length).  What makes that exact:
//   * a slot is reused for column j + C only at a row where i + w + 1 >= j + C, i.e. beg >= i - w > j: column j has left the band for good;
//   * a column the band reaches for the first time must hold the FIRST ROW's value (the "stale cells" the function reads when its band grows,
//     :143-145 with :183-201): columns enter in increasing order, one per row (end <= i + w + 1), so the ring is topped up with
//     {H(-1, j), E = 0, query base j} for j = i + w + 1 at the head of row i -- the query byte requested a row ahead;
//   * columns the band re-reads after shrinking (end = jt + 2 can fall and rise again) are still in the ring: they are above beg.
// The cell, the row bookkeeping, the trimming and the tie rules are the linear kernel's, statement for statement; only addresses differ.  A
// row's columns are walked in runs that do not cross the ring's seam (at most one seam per row: one cell computed on its own).
struct LaneCircArgs { LaneArgs L; int C; };

__global__ void __launch_bounds__(64) k_bsw_lane_circ(LaneCircArgs AC) {
    extern __shared__ __attribute__((aligned(16))) unsigned int he_raw[];
    typedef __attribute__((address_space(3))) unsigned int* lds_u32;
    LaneArgs& A = AC.L;
    const int C = AC.C;
    const int lane = threadIdx.x;
    const lds_u32 he = (lds_u32)he_raw + lane;           // slot s of this lane's pair: he[s * 64]
    const int o_del = A.o.o_del, e_del = A.o.e_del, o_ins = A.o.o_ins, e_ins = A.o.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins, zdrop = A.o.zdrop;
    const int sa = A.o.a, sb = -A.o.b;
    constexpr unsigned HE_MASK = 0xffffff00u, QMASK = 0xffu;
    if (A.offs) { A.first = A.offs[A.key_first]; A.count = A.offs[A.key_last] - A.first; }
    for (;;) {
        unsigned int tk = 0;
        if (lane == 0) tk = atomicAdd(A.ticket, 64u);
        tk = __shfl(tk, 0);
        if (tk >= (unsigned)A.count) break;
        const bool active = tk + lane < (unsigned)A.count;
        meme_seqpair* P = &A.pairs[A.order[A.first + (active ? tk + lane : tk)]];
        const int qlen = active ? P->len2 : 0, tlen = active ? P->len1 : 0, h0 = P->h0;
        const uint8_t* query = A.qer + P->idq;
        const uint8_t* target = A.ref + P->idr;
        const int first = h0 > oe_ins ? h0 - oe_ins : 0;
        // H(-1, j) of the first row (:143-145)
#define BSW_ROW0(j_) ((j_) == 0 ? h0 : ((j_) == 1 ? first : ((first - ((j_) - 2) * e_ins) > e_ins ? (first - ((j_) - 2) * e_ins) - e_ins : 0)))
        int hi = qlen < C - 1 ? qlen : C - 1;                 // highest column the ring holds so far
        for (int j0 = 0; j0 <= hi; j0 += 8) {
            unsigned qb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) qb[u] = j0 + u < qlen ? (unsigned)query[j0 + u] : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                if (j > hi) break;
                he[j * 64] = he_pack(BSW_ROW0(j), 0, qb[u] > 4u ? 4u : qb[u]);
            }
        }
        int w = A.w;
        {
            const int mx = sa > 0 ? sa : 0;
            int max_ins = (int)((double)(qlen * mx + A.o.end_bonus - o_ins) / e_ins + 1.);
            if (max_ins < 1) max_ins = 1;
            if (w > max_ins) w = max_ins;
            int max_del = (int)((double)(qlen * mx + A.o.end_bonus - o_del) / e_del + 1.);
            if (max_del < 1) max_del = 1;
            if (w > max_del) w = max_del;
        }
        int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
        int beg = 0, end = qlen;
        int cb = 0;                                           // multiple of C with cb <= beg < cb + C: column j sits in slot j - cb (- C beyond the seam)
#define BSW_SLOT(j_) (((j_) - cb) >= C ? ((j_) - cb - C) : ((j_) - cb))
        int tb_next = tlen > 0 ? (int)target[0] : 4;
        unsigned q_next = hi + 1 < qlen ? (unsigned)query[hi + 1] : 0u;        // base of the next column to enter the ring
        for (int i = 0; i < tlen; ++i) {
            const int tb = tb_next;
            if (i + 1 < tlen) tb_next = target[i + 1];
            int f = 0, h1, m = 0, mj = -1;
            if (beg < i - w) beg = i - w;
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            while (beg - cb >= C) cb += C;
            // the column the band may reach for the first time in this row enters the ring with the first row's value
            {
                const int need = i + w + 1 < qlen ? i + w + 1 : qlen;
                if (hi < need) {
                    ++hi;
                    he[BSW_SLOT(hi) * 64] = he_pack(BSW_ROW0(hi), 0, q_next > 4u ? 4u : q_next);
                    q_next = hi + 1 < qlen ? (unsigned)query[hi + 1] : 0u;      // in flight during the row
                }
            }
            if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
            else h1 = 0;
            const unsigned sb4 = (unsigned)(sb & 0xff) * 0x01010101u;
            const unsigned tab_lo = tb > 3 ? 0xffffffffu : (sb4 & ~(0xffu << (8 * tb))) | ((unsigned)(sa & 0xff) << (8 * tb));
            const unsigned tab_hi = 0xffffffffu;
            // one cell: `cur_` holds the word of the cell's slot (pc_), the next slot's word (pn_) is requested first
#define BSW_CELLC(cur_, nxt_, pc_, pn_, mk_, jb_)                                                             \
            {                                                                                                  \
                nxt_ = *(pn_);                                                                                 \
                int M = (int)((cur_ >> 8) & 0xfffu), e = (int)(cur_ >> 20);                                     \
                const int sc = (int)(signed char)(__builtin_amdgcn_perm(tab_hi, tab_lo, cur_) & 0xffu);         \
                M = M ? M + sc : 0;                                                                            \
                const int h = max3_i32(M, e, f);                                                               \
                const int key = (h << 10) + (jb_);                                                             \
                mk_ = mk_ > key ? mk_ : key;                                                                   \
                e = max3_i32(M - oe_del, e - e_del, 0);                                                        \
                *(pc_) = he_repack(h1, e, cur_);                                                               \
                h1 = h;                                                                                        \
                f = max3_i32(M - oe_ins, f - e_ins, 0);                                                        \
            }
            int s = BSW_SLOT(beg);
            unsigned wa = beg < end ? he[s * 64] : 0u, wb = 0u, wc = 0u, wd = 0u;
            int mk0 = -8, mk1 = -8, mk2 = -8, mk3 = -8;
            int j = beg;
            while (j < end) {
                // cells whose next slot is s + 1 (the run up to the ring's seam), four per trip
                int run = end - j < C - 1 - s ? end - j : C - 1 - s;
                lds_u32 p = he + s * 64;
                const int j_run = j + run;
                for (; j + 3 < j_run; j += 4, p += 256) {
                    BSW_CELLC(wa, wb, p, p + 64, mk0, j)
                    BSW_CELLC(wb, wc, p + 64, p + 128, mk1, j)
                    BSW_CELLC(wc, wd, p + 128, p + 192, mk2, j)
                    BSW_CELLC(wd, wa, p + 192, p + 256, mk3, j)
                }
                for (; j < j_run; ++j, p += 64) {
                    BSW_CELLC(wa, wb, p, p + 64, mk0, j)
                    wa = wb;
                }
                s += run;
                if (j < end) {                                 // the cell in the ring's last slot: its right neighbour is slot 0
                    BSW_CELLC(wa, wb, he + (C - 1) * 64, he, mk0, j)
                    wa = wb;
                    ++j; s = 0;
                }
            }
#undef BSW_CELLC
            {
                mk1 += 1; mk2 += 2; mk3 += 3;
                const int ka = mk0 > mk1 ? mk0 : mk1, kb = mk2 > mk3 ? mk2 : mk3;
                const int key = ka > kb ? ka : kb;
                m = key < 0 ? 0 : key >> 10;
                mj = key < 0 ? -1 : key & 1023;
            }
            const int s_end = BSW_SLOT(end);
            he[s_end * 64] = he_pack(h1, 0, he[s_end * 64] & QMASK);    // :201
            const unsigned w_front = he[BSW_SLOT(beg) * 64];
            if ((beg < end ? end : beg) == qlen) {
                max_ie = gscore > h1 ? max_ie : i;
                gscore = gscore > h1 ? gscore : h1;
            }
            if (m == 0) break;
            if (m > max) {
                max = m; max_i = i; max_j = mj;
                int off = mj - i;
                if (off < 0) off = -off;
                max_off = max_off > off ? max_off : off;
            } else if (zdrop > 0) {
                if (i - max_i > mj - max_j) {
                    if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
                } else {
                    if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
                }
            }
            int jt = beg;
            if (jt < end && (w_front & HE_MASK) == 0u) {
                ++jt;
                while (jt < end && (he[BSW_SLOT(jt) * 64] & HE_MASK) == 0u) ++jt;
            }
            beg = jt;
            jt = end;
            if (h1 == 0) {
                --jt;
                while (jt >= beg && (he[BSW_SLOT(jt) * 64] & HE_MASK) == 0u) --jt;
            }
            end = jt + 2 < qlen ? jt + 2 : qlen;
        }
#undef BSW_SLOT
#undef BSW_ROW0
        if (active) {
            P->score = max;
            P->qle = max_j + 1;
            P->tle = max_i + 1;
            P->gtle = max_ie + 1;
            P->gscore = gscore;
            P->max_off = max_off;
        }
    }
}

// ---- counting sort of the pairs by query length (the lanes of a wavefront should finish together) -------------------------
// Sort key: query length first (LDS size class, column-loop length), then the surplus of target rows over query columns
// in steps of 8, so that the 64 pairs of a wavefront also run about the same number of rows.
// a >= 0: keys for the lane-per-pair kernel (the last key collects the pairs it cannot take);
// a < 0: the whole batch goes to the lanes-per-pair kernels (queries beyond LANE_QMAX collect in the last key)
__device__ __forceinline__ int lane_key(const meme_seqpair& p, int a) {
    const long long bound = (long long)p.h0 + (long long)p.len2 * (a > 0 ? a : 0) + 1;
    const bool ok = p.len2 >= 0 && p.len2 <= LANE_QMAX && p.len1 >= 0 &&
                    (a < 0 || (p.h0 >= 0 && bound < LANE_SCORE_LIMIT));
    if (!ok) return p.len2 < 0 ? 0 : SORT_KEYS - 1;
    int sub = (p.len1 - p.len2) >> 3;
    sub = sub < 0 ? 0 : (sub > TL_BUCKETS - 1 ? TL_BUCKETS - 1 : sub);
    return QKEY(p.len2) + sub;
}

__global__ void __launch_bounds__(256) k_bsw_hist(const meme_seqpair* __restrict__ pairs, int n, int a, int* __restrict__ hist,
                                                   int* __restrict__ maxq) {
    __shared__ int lh[SORT_KEYS];
    for (int k = threadIdx.x; k < SORT_KEYS; k += 256) lh[k] = 0;
    __syncthreads();
    int mq = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int key = lane_key(pairs[i], a);
        atomicAdd(&lh[key], 1);
        if (key == SORT_KEYS - 1 || a < 0) mq = mq > pairs[i].len2 ? mq : pairs[i].len2;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < SORT_KEYS; k += 256) if (lh[k]) atomicAdd(&hist[k], lh[k]);
    if (mq) atomicMax(maxq, mq);
}

__global__ void __launch_bounds__(1024) k_bsw_scan(const int* __restrict__ hist, int* __restrict__ offs, int* __restrict__ cursor) {
    // exclusive scan of SORT_KEYS counts: every thread owns a run of consecutive keys, the run totals are scanned in LDS
    constexpr int PER = (SORT_KEYS + 1023) / 1024;
    __shared__ int tmp[1024];
    const int t = threadIdx.x, k0 = t * PER;
    int sum = 0;
    for (int k = k0; k < k0 + PER && k < SORT_KEYS; ++k) sum += hist[k];
    tmp[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = t >= d ? tmp[t - d] : 0;
        __syncthreads();
        tmp[t] += v;
        __syncthreads();
    }
    int run = tmp[t] - sum;
    for (int k = k0; k < k0 + PER && k < SORT_KEYS; ++k) { offs[k] = run; cursor[k] = run; run += hist[k]; }
    if (t == 1023) offs[SORT_KEYS] = tmp[t];
}

__global__ void __launch_bounds__(256) k_bsw_scatter(const meme_seqpair* __restrict__ pairs, int n, int a, int* __restrict__ cursor,
                                                      int* __restrict__ order) {
    __shared__ int lh[SORT_KEYS];
    __shared__ int lbase[SORT_KEYS];
    for (int i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {
        for (int k = threadIdx.x; k < SORT_KEYS; k += 256) lh[k] = 0;
        __syncthreads();
        const int i = i0 + threadIdx.x;
        int key = 0, r = 0;
        if (i < n) { key = lane_key(pairs[i], a); r = atomicAdd(&lh[key], 1); }
        __syncthreads();
        for (int k = threadIdx.x; k < SORT_KEYS; k += 256) if (lh[k]) lbase[k] = atomicAdd(&cursor[k], lh[k]);
        __syncthreads();
        if (i < n) order[lbase[key] + r] = i;
        __syncthreads();
    }
}

template <int LP>
int launch_cls(meme_ctx* ctx, BswArgs A, int qmax, i64 dev_cus) {
    constexpr int GROUPS = BSW_BLOCK / LP;
    size_t per_grp = (size_t)(qmax + 2) * 8 + (size_t)((qmax + 7) & ~7);
    size_t lds = per_grp * GROUPS;
    i64 blocks = ctx->bsw_blocks > 0 ? ctx->bsw_blocks : dev_cus * 4;
    i64 want = (A.npairs + GROUPS - 1) / GROUPS;        // (upper bound when the class range is read on the device)
    if (blocks > want) blocks = want;
    if (blocks < 1) blocks = 1;
    A.gws = nullptr;
    if (lds > 160 * 1024) {
        // BandedPairWiseSW's 16-bit class takes sequences below 32768 bases (src/bandedSWA.h:47-86): rows in an HBM workspace
        if (qmax > 32768) { meme_set_error("query of %d bases: beyond the 32768 the reference's banded SW accepts", qmax); return MEME_E_ARG; }
        if (blocks > dev_cus * 2) blocks = dev_cus * 2;
        int rc = meme_buf_reserve(ctx, ctx->bsw_ws, per_grp * GROUPS * (size_t)blocks);
        if (rc) return rc;
        A.gws = (unsigned char*)ctx->bsw_ws.p;
        lds = 0;
    }
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_bsw<LP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    A.qmax = qmax;
    hipLaunchKernelGGL(k_bsw<LP>, dim3((unsigned)blocks), dim3(BSW_BLOCK), lds, ctx->stream, A);
    HIP_TRY(hipGetLastError());
    return MEME_OK;
}

// `host_maxq`: longest query of the batch when the caller knows it (host API), else -1: then the class boundaries and the
// longest query of the pairs the lane kernel cannot take are read back from the device before the DP kernels are sized.
int launch_bsw(meme_ctx* ctx, meme_seqpair* d_pairs, const uint8_t* d_ref, const uint8_t* d_qer, int npairs, int w,
               const meme_bsw_opt* opt, int host_maxq) {
    if (opt->e_ins <= 0 || opt->e_del <= 0) { meme_set_error("gap extension penalties must be positive"); return MEME_E_ARG; }
    int rc;
    // counters (ints): [0..SORT_KEYS) histogram by query length, [SORT_KEYS..2*SORT_KEYS] exclusive offsets (+ total),
    // then the scatter cursors, the longest query of the pairs the lane kernel cannot take, and the kernels' tickets
    const size_t n_ints = 3 * (size_t)SORT_KEYS + 32;
    if ((rc = meme_buf_reserve(ctx, ctx->counters, n_ints * sizeof(int)))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->bsw_order, (size_t)npairs * sizeof(int)))) return rc;
    int* hist = (int*)ctx->counters.p;
    int* offs = hist + SORT_KEYS;
    int* cursor = offs + SORT_KEYS + 1;
    int* maxq = cursor + SORT_KEYS;
    unsigned int* tickets = (unsigned int*)(maxq + 1);
    int* order = (int*)ctx->bsw_order.p;
    const i64 dev_cus = ctx->n_cus;
    HIP_TRY(hipMemsetAsync(hist, 0, n_ints * sizeof(int), ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev[4], ctx->stream));
    // big batches: one pair per lane (throughput).  Small batches (the reference's 512-read call granularity): 16-64
    // lanes per pair, because a lone pair on one lane takes milliseconds.
    // (the lane kernel keeps a row's substitution scores as signed bytes)
    const bool use_lane = (i64)npairs >= ctx->bsw_lane_min_pairs && opt->a >= 0 && opt->a <= 127 && opt->b >= 0 && opt->b <= 128;
    const int key_a = use_lane ? (opt->a > 0 ? opt->a : 0) : -1;
    {
        i64 sblocks = ((i64)npairs + 255) / 256;
        if (sblocks > dev_cus * 8) sblocks = dev_cus * 8;
        hipLaunchKernelGGL(k_bsw_hist, dim3((unsigned)sblocks), dim3(256), 0, ctx->stream, d_pairs, npairs, key_a, hist, maxq);
        hipLaunchKernelGGL(k_bsw_scan, dim3(1), dim3(1024), 0, ctx->stream, hist, offs, cursor);
        hipLaunchKernelGGL(k_bsw_scatter, dim3((unsigned)sblocks), dim3(256), 0, ctx->stream, d_pairs, npairs, key_a, cursor, order);
        HIP_TRY(hipGetLastError());
    }
    int mq = host_maxq;
    if (mq < 0) {
        int h_mq = 0;
        HIP_TRY(hipMemcpyAsync(&h_mq, maxq, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        mq = h_mq;
    }
    if (mq < 1) mq = 1;
    if (use_lane) {
        // ---- lane-per-pair kernel, one launch per LDS size class (pairs are sorted by query length); a class finds its
        // range in the device-side scan and leaves at once when it is empty.  Launches of one stream run one after the
        // other, so a batch that cannot fill the GPU anyway (a combined call of the aligner's worker threads: 20-300 k
        // pairs) goes out as ONE launch with the LDS of the longest query's class: its time is the slowest wavefront's,
        // not the sum over the classes.
        const bool one_launch = host_maxq >= 0 && (i64)npairs < (i64)dev_cus * 64 * 16;
        int qlo = 0;
        for (int c = 0; c < N_LANE_CLS; ++c) {
            const int qhi = LANE_CLS_Q[c];                       // class = query lengths [qlo, qhi]
            LaneArgs L;
            L.pairs = d_pairs; L.ref = d_ref; L.qer = d_qer; L.order = order; L.first = 0; L.count = 0;
            L.w = w; L.o = *opt; L.ticket = tickets + c;
            L.offs = offs; L.key_first = QKEY(qlo); L.key_last = QKEY(qhi + 1);
            qlo = qhi + 1;
            if (host_maxq >= 0 && L.key_first > QKEY(host_maxq)) continue;    // no query of the batch is that long
            if (one_launch) {
                if (qhi < host_maxq && c + 1 < N_LANE_CLS) continue;           // not the top class of this batch yet
                L.key_first = 0;
            }
            i64 want = ((i64)npairs + 63) / 64;
            i64 blocks = ctx->bsw_blocks > 0 ? ctx->bsw_blocks : dev_cus * 16;
            if (blocks > want) blocks = want;
            // classes whose queries are longer than the band is wide keep 2w + 2 columns in a ring instead of one word per query column
            // (round 6: 53 KB per wavefront at w = 100 -- three wavefronts per CU -- where the 222 / 318 / 600-column classes hold two / one / one)
            const int ring = ((2 * (w > 0 ? w : 0) + 4 + 3) / 4) * 4;        // >= 2w + 2 columns (w = 100: 204 columns = 52 224 bytes, three wavefronts in a CU's 160 KB)
            if (ctx->bsw_circ != 0 && ring < qhi + 2) {
                const size_t lds = (size_t)ring * 64 * sizeof(unsigned int);
                if (lds > 64 * 1024)
                    HIP_TRY(hipFuncSetAttribute((const void*)k_bsw_lane_circ, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                LaneCircArgs LC;
                LC.L = L; LC.C = ring;
                hipLaunchKernelGGL(k_bsw_lane_circ, dim3((unsigned)blocks), dim3(64), lds, ctx->stream, LC);
            } else {
                const size_t lds = (size_t)(qhi + 2) * 64 * sizeof(unsigned int);
                if (lds > 64 * 1024)
                    HIP_TRY(hipFuncSetAttribute((const void*)k_bsw_lane, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(k_bsw_lane, dim3((unsigned)blocks), dim3(64), lds, ctx->stream, L);
            }
            HIP_TRY(hipGetLastError());
            if (one_launch) break;
        }
    }
    // ---- lanes-per-pair kernel: the pairs the lane kernel cannot take (long queries, scores beyond 12 bits), or the whole
    // batch when it is small; 16 / 32 / 64 lanes per pair by query length (4 / 2 / 1 pairs per wavefront)
    {
        BswArgs A;
        A.pairs = d_pairs; A.ref = d_ref; A.qer = d_qer; A.w = w; A.o = *opt; A.qmax = 0; A.order = order; A.npairs = npairs;
        A.offs = offs;
        if (!use_lane) {
            A.key_first = 0; A.key_last = QKEY(17); A.ticket = tickets + N_LANE_CLS;
            if ((rc = launch_cls<16>(ctx, A, 16, dev_cus))) return rc;
            if (mq > 16) {
                A.key_first = QKEY(17); A.key_last = QKEY(33); A.ticket = tickets + N_LANE_CLS + 1;
                if ((rc = launch_cls<32>(ctx, A, 32, dev_cus))) return rc;
            }
        }
        if (use_lane || mq > 32) {
            A.key_first = use_lane ? SORT_KEYS - 1 : QKEY(33); A.key_last = SORT_KEYS; A.ticket = tickets + N_LANE_CLS + 2;
            if ((rc = launch_cls<64>(ctx, A, ((mq + 63) / 64) * 64, dev_cus))) return rc;
        }
    }
    HIP_TRY(hipEventRecord(ctx->ev[5], ctx->stream));
    ctx->tm.bsw_launches = 1;
    return MEME_OK;
}

int finish_bsw(meme_ctx* ctx) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]));
    ctx->tm.bsw_kernel_ms = ms;
    return MEME_OK;
}

}  // namespace

int meme_bsw_launch(meme_ctx* ctx, meme_seqpair* d_pairs, const uint8_t* d_ref, const uint8_t* d_qer, int npairs, int w, const meme_bsw_opt* opt,
                    int host_maxq) {
    return launch_bsw(ctx, d_pairs, d_ref, d_qer, npairs, w, opt, host_maxq);
}

extern "C" int meme_bsw_batch_device(meme_ctx* ctx, meme_seqpair* d_pairs, const uint8_t* d_ref, const uint8_t* d_qer,
                                     int32_t npairs, int32_t w, const meme_bsw_opt* opt) {
    if (!ctx || !d_pairs || !d_ref || !d_qer || !opt || npairs < 0) return MEME_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (npairs == 0) return MEME_OK;
    int rc = launch_bsw(ctx, d_pairs, d_ref, d_qer, npairs, w, opt, -1);
    if (rc) return rc;
    return finish_bsw(ctx);
}

extern "C" int meme_bsw_batch(meme_ctx* ctx, meme_seqpair* pairs, const uint8_t* ref_buf, int64_t ref_bytes,
                              const uint8_t* qer_buf, int64_t qer_bytes, int32_t npairs, int32_t w, const meme_bsw_opt* opt) {
    if (!ctx || !pairs || !ref_buf || !qer_buf || !opt || npairs < 0 || ref_bytes < 0 || qer_bytes < 0) return MEME_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (npairs == 0) return MEME_OK;
    static const bool trace = getenv("MEME_BSW_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = trace ? now() : 0;
    int maxq = 0;
    for (int i = 0; i < npairs; ++i) {
        const meme_seqpair& p = pairs[i];
        if (p.len1 < 0 || p.len2 < 0 || p.idr < 0 || p.idq < 0 || (int64_t)p.idr + p.len1 > ref_bytes ||
            (int64_t)p.idq + p.len2 > qer_bytes) {
            meme_set_error("pair %d addresses bytes outside the sequence buffers", i);
            return MEME_E_ARG;
        }
        maxq = p.len2 > maxq ? p.len2 : maxq;
    }
    int rc;
    const double t1 = trace ? now() : 0;
    if ((rc = meme_buf_reserve(ctx, ctx->pairs, (size_t)npairs * sizeof(meme_seqpair)))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->refb, (size_t)ref_bytes + 16))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->qerb, (size_t)qer_bytes + 16))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->pairs.p, pairs, (size_t)npairs * sizeof(meme_seqpair), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->refb.p, ref_buf, (size_t)ref_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->qerb.p, qer_buf, (size_t)qer_bytes, hipMemcpyHostToDevice, ctx->stream));
    // one host synchronisation per call: the class ranges stay on the device (the host knows the longest query)
    const double t2 = trace ? now() : 0;
    rc = launch_bsw(ctx, (meme_seqpair*)ctx->pairs.p, (const uint8_t*)ctx->refb.p, (const uint8_t*)ctx->qerb.p, npairs, w,
                    opt, maxq);
    if (rc) return rc;
    const double t3 = trace ? now() : 0;
    HIP_TRY(hipMemcpyAsync(pairs, ctx->pairs.p, (size_t)npairs * sizeof(meme_seqpair), hipMemcpyDeviceToHost, ctx->stream));
    rc = finish_bsw(ctx);
    if (trace)
        fprintf(stderr, "[meme bsw] %d pairs, %.1f MB in: validate %.2f ms, reserve + H2D enqueue %.2f ms, launches %.2f ms, D2H + wait %.2f ms (GPU %.2f ms)\n",
                npairs, (ref_bytes + qer_bytes + (double)npairs * sizeof(meme_seqpair)) / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3,
                (now() - t3) * 1e3, ctx->tm.bsw_kernel_ms);
    return rc;
}
