// Learned-index seeding kernels (device code), included by meme_seed.hip.
//
// Reference functions restated here (paths in the BWA-MEME tree):
//   Learned_getSMEMsAllPosOneThread        src/LearnedIndex_seeding.cpp:913-972   (rounds 1 and 2)
//     Learned_getSMEMsOnePosOneThread_step1                               :1691-1894
//     Learned_getSMEMsOnePosOneThread                                     :1897-2126
//   Learned_bwtSeedStrategyAllPosOneThread[_mem_tradeoff]                 :974-1466  (round 3)
//   mem_search / right_smem_search [+ _tradeoff]                          :2131-4189
//   learned_index_lookup                                                  :186-210
//   compare_read_and_ref_binary*                                          :226-601
//   read packing of mem_kernel1_core_Learned                              src/bwamem.cpp:1277-1344
//
// Design (MI355X-first, not a translation):
//  * A read is owned by a group of G lanes (G = 4..32, 64/G reads per wavefront).  The pivot state
//    machine is group-uniform scalar state replicated in the group's lanes; only suffix-array probes
//    are lane-parallel.
//  * The pivot logic of rounds 1-3 is flattened into an explicit state machine ("program counter"
//    per read) around ONE search call site, so the 64/G reads of a wavefront execute the search body
//    convergently and their memory requests are issued together -- instead of each read sitting in a
//    different call site of a recursive-descent transcription of the CPU code.
//  * The unit of memory traffic is a *window*: G consecutive 16-byte suffix-array entries, one
//    coalesced load.  Each lane compares its entry's 64-bit key (and, only when all 32 bases agree,
//    2-bit reference words) with the read; a wave ballot yields the partition point, the longest
//    common prefix and -- from the same data -- the SMEM hit interval, so the reference's chain of
//    ~log2(err)+linear dependent single-entry probes collapses to one window in the common case.
//  * The learned model is a hint: when the partition point is outside the first
//    window the group gallops away from the prediction and bisects with group-uniform single-entry
//    probes (broadcast loads), then takes a final window.  Model error bounds are never needed, so
//    any parameter file the reference loads is accepted.
//  * Reads are packed once per batch by k_pack_reads (2 bits/base, both strands, first base in the
//    top bits of each u64, plus N masks) and staged in LDS; a 32-base query word at any offset is a
//    funnel shift of two LDS words, replacing the reference's 8 pre-shifted copies of every read.
#pragma once
#include <limits.h>

#include "meme_common.h"

namespace seedk {

constexpr int BLOCK = 256;
constexpr int MAX_READ_LEN = 500;        // LEARNED_MAX_READ_LEN (reference src/bwamem.cpp:1259)

struct SlotRec {          // search-kernel output, one per SMEM
    int32_t start, end;
    i64 sa_start;
    i64 count;
};

constexpr int N_TIERS = 3;
constexpr int TIER_CAP[N_TIERS] = {64, 2048, 65536};   // global SMEM slots per read; tier 0 is tunable
constexpr int TIER_LCAP[N_TIERS] = {16, 512, 512};     // LDS ring: SMEMs of one first-round pass (<= read length)

struct PackGeom {
    int W;        // u64 words per strand (>= ceil(maxlen/32) + 2)
    int MW;       // u64 N-mask words per strand
    int stride;   // 2*W + 2*MW + 1 (last word = read length)
};

struct SeedArgs {
    DevIndex I;
    const u64* packed;     // [nreads_total * stride]
    const i64* read_off;
    i64 nreads;
    PackGeom geo;
    meme_seed_opt opt;
    SlotRec* slots;        // [nreads * cap] for this tier
    int* slot_cnt;         // [all reads]
    i64* slot_hits;
    i64* slot_loc;         // (tier << 40) | block index inside the tier's slot array
    const i64* pending;    // read ids to re-process in an overflow tier, else nullptr
    i64* ovf_list;
    int cap, lcap, tier;
    unsigned long long* counters;   // [0] ticket, [1] searches, [2] overflowed reads, [3] window loads
};

// ---- read packing -------------------------------------------------------------------------------------
// Layout per read: fw[W] rc[W] nfw[MW] nrc[MW] len.  A workgroup packs PACK_RB consecutive reads: their bytes are
// contiguous in the input, so they are staged in LDS with aligned, coalesced dword loads and the 2-bit words are then
// assembled from LDS bytes (the first version issued 32 scattered byte loads per output word: 11.7 ms per 10 M reads).
__global__ void __launch_bounds__(256) k_pack_reads(const uint8_t* __restrict__ reads, const i64* __restrict__ read_off,
                                                     i64 nreads, i64 total_bytes, PackGeom g, int PACK_RB,
                                                     u64* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pk_raw[];
    uint32_t* stage = reinterpret_cast<uint32_t*>(pk_raw);
    const uint8_t* sb = pk_raw;
    for (i64 r0 = (i64)blockIdx.x * PACK_RB; r0 < nreads; r0 += (i64)gridDim.x * PACK_RB) {
        const int nr = (int)(nreads - r0 < PACK_RB ? nreads - r0 : PACK_RB);
        const i64 b0 = read_off[r0], b1 = read_off[r0 + nr];
        const i64 a0 = b0 & ~3ll;                              // dword-aligned start (reads + a0 is 4-byte aligned)
        const int ndw = (int)((b1 - a0 + 3) >> 2);
        __syncthreads();                                       // previous iteration's readers are done
        const uint32_t* src = reinterpret_cast<const uint32_t*>(reads + a0);
        for (int k = threadIdx.x; k < ndw; k += blockDim.x) {
            uint32_t v;
            if (a0 + 4 * (i64)k + 4 <= total_bytes) v = src[k];
            else {                                              // last dword of the buffer: never read past its end
                v = 0;
                for (int bb = 0; bb < 4; ++bb)
                    if (a0 + 4 * (i64)k + bb < total_bytes) v |= (uint32_t)reads[a0 + 4 * (i64)k + bb] << (8 * bb);
            }
            stage[k] = v;
        }
        __syncthreads();
        const int shift = (int)(b0 - a0);
        for (int wk = threadIdx.x; wk < nr * g.stride; wk += blockDim.x) {
            const int rr = wk / g.stride, k = wk - rr * g.stride;
            const i64 r = r0 + rr;
            const i64 ro = read_off[r];
            int len = (int)(read_off[r + 1] - ro);
            u64 v = 0;
            if (k == g.stride - 1) v = (u64)(unsigned)len;     // length word
            else {
                if (len > MAX_READ_LEN) len = 0;
                const uint8_t* p = sb + shift + (int)(ro - b0);
                if (k < 2 * g.W) {
                    const bool rc = k >= g.W;
                    const int w = rc ? k - g.W : k;
                    for (int j = 0; j < 32; ++j) {
                        const int i = 32 * w + j;
                        u64 c = 0;
                        if (i < len) {
                            const uint8_t bb = rc ? p[len - 1 - i] : p[i];
                            c = bb < 4 ? (rc ? 3 - bb : bb) : 0;   // N packed as A (src/bwamem.cpp:1293-1294)
                        }
                        v = (v << 2) | c;
                    }
                } else {
                    int m = k - 2 * g.W;
                    const bool rc = m >= g.MW;
                    if (rc) m -= g.MW;
                    for (int j = 0; j < 64; ++j) {
                        const int i = 64 * m + j;
                        if (i < len) {
                            const uint8_t bb = rc ? p[len - 1 - i] : p[i];
                            if (bb >= 4) v |= 1ull << j;
                        }
                    }
                }
            }
            out[r * g.stride + k] = v;
        }
    }
}

// ---- per-read state ------------------------------------------------------------------------------------
enum Pc : int {
    PC_ALLPOS_TOP, PC_ZZ_TOP, PC_ZZ_RIGHT, PC_ZZ_END, PC_AFTER_STEP1, PC_R2_LOOP, PC_R2_AFTER, PC_R3_INIT,
    PC_R3_TOP, PC_DONE
};
enum Kind : int { K_S1_RIGHT, K_ZZ_LEFT, K_ZZ_RIGHT, K_OP_MEM, K_OP_SMEM, K_R3 };

struct Req {
    int kind;
    bool rc;          // query strand: reverse complement (left extension) or forward
    int off, vlen;
    int min_intv;
    int mode;         // 0: match length only; 1: + interval with >= min_intv suffixes; 2: third-round levels
    bool exact;       // the interval at the final level is emitted: its edges must be exact
};

struct Res {
    int L;            // match length (mode 2: the advance)
    i64 start, count; // SA interval
    bool emit;        // mode 2: an SMEM is to be emitted
};

// explicit address spaces: LDS (3) for the staged read, global (1) for the index.  Generic pointers would
// compile to FLAT loads whose waits serialise LDS and HBM traffic.
#ifndef CMP_WORDS
#define CMP_WORDS 2          // reference words fetched per round trip once the 32-base key matched
#endif
typedef const __attribute__((address_space(3))) u64* lds_u64;
typedef __attribute__((address_space(3))) int* lds_int;
typedef const __attribute__((address_space(1))) u64* glb_u64;
typedef const __attribute__((address_space(1))) SaEnt* glb_ent;
typedef const __attribute__((address_space(1))) RmiRec* glb_rmi;

__device__ __forceinline__ u64 ext_l(lds_u64 w, int s) {
    int k = s >> 5, sh = (s & 31) * 2;
    u64 a = w[k], b = w[k + 1];
    return sh ? (a << sh) | (b >> (64 - sh)) : a;
}

// cold per-read state kept in LDS (group-uniform redundant stores; every lane reads back what it wrote)
enum StIdx : int { ST_BEFORE, ST_AFTER, ST_R2_K, ST_R2_NEXT, ST_R2_SAVED, ST_ZZ_NEXT, ST_ZZ_SP, ST_ZZ_GUARD, ST_AP_GUARD,
                   ST_SM_BASE, ST_N_SMEMS, ST_HITS_LO, ST_HITS_HI, ST_SEARCHES, ST_FLAGS, ST_LAST_CNT_LO, ST_LAST_CNT_HI,
                   ST_LAST_S_LO, ST_LAST_S_HI, ST_WINDOWS, ST_WORDS };
enum StFlag : int { F_ZZ_CHECK = 1, F_ZZ_RET_ONEPOS = 2, F_REC = 4, F_LDS_OVF = 8 };

template <int G>
struct Grp {
    static constexpr u64 FULL = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
    glb_ent sa;
    glb_u64 pac;
    glb_rmi l2, l1;
    i64 n;
    int shift;
    lds_u64 fw, rc, nfw, nrc;
    lds_int st;
    int t, gbase;

    __device__ __forceinline__ u64 ballot(bool p) const { return (__ballot(p) >> gbase) & FULL; }
    __device__ __forceinline__ int shfl(int v, int src) const { return __shfl(v, gbase + src); }
};

// ---- compare: compare_read_and_ref_binary* (:226-601) ---------------------------------------------------
// L = min(cap, n - pos).  lcp < L: less = ref base < read base.  lcp == L: less = (L < ref_len)
// ("exact": the suffix continues past the query and sorts before it; a suffix that ends first sorts
// after it, as if followed by T-padding).
template <int G>
__device__ __forceinline__ void cmp_entry(const Grp<G>& g, lds_u64 s, int off, int cap, u64 ekey, u64 epos, int& lcp,
                                          bool& less) {
    i64 ref_len = g.n - (i64)epos;
    int L = ref_len < (i64)cap ? (int)ref_len : cap;
    u64 wq = ext_l(s, off);
    u64 x = ekey ^ wq;
    int l;
    bool lt = false;
    if (x) { l = __clzll((long long)x) >> 1; lt = ekey < wq; }
    else {
        l = 32;
        if (l < L) {
            // All 32 key bases agree: the rest comes from the 2-bit text.  Consecutive words are adjacent in
            // memory (same sector), so CMP_WORDS are fetched per round trip instead of one dependent load per word.
            const i64 p0 = (i64)epos + 32;
            glb_u64 pw = g.pac + (p0 >> 5);
            const int sh = (int)(p0 & 31) * 2;
            bool done = false;
            for (int k = 1; !done; k += CMP_WORDS, pw += CMP_WORDS) {
                u64 w[CMP_WORDS + 1];
#pragma unroll
                for (int j = 0; j < CMP_WORDS + 1; ++j) w[j] = pw[j];
#pragma unroll
                for (int j = 0; j < CMP_WORDS; ++j) {
                    if (done) break;
                    u64 wr = sh ? (w[j] << sh) | (w[j + 1] >> (64 - sh)) : w[j];
                    u64 q = ext_l(s, off + 32 * (k + j));
                    u64 y = wr ^ q;
                    if (y) { l += __clzll((long long)y) >> 1; lt = wr < q; done = true; }
                    else { l += 32; if (l >= L) done = true; }
                }
            }
        }
    }
    if (l >= L) { lcp = L; less = (i64)L < ref_len; }
    else { lcp = l; less = lt; }
}

// ---- learned_index_lookup (:186-210): same arithmetic (FP64 FMA + clamp), used as a hint ------------------
template <int G>
__device__ __forceinline__ i64 rmi_lookup(const Grp<G>& g, u64 key) {
    u64 m = g.shift >= 64 ? 0ull : key >> g.shift;
    double icpt = g.l2[m].icpt, slope = g.l2[m].slope;
    u64 err = g.l2[m].err;
    double x = (double)key;
    double f = fma(slope, x, icpt);
    if (err >> 63) {
        u64 ps = (err >> 32) & 0x7fffffffull;
        double pn = (double)(err & 0xffffffffull) - 1.0;
        double c = f < 0.0 ? 0.0 : (f > pn ? pn : f);
        u64 j = ps + (u64)c;
        f = fma(g.l1[j].slope, x, g.l1[j].icpt);
    }
    double top = (double)g.n - 1.0;
    if (f < 0.0) return 0;
    if (f > top) return g.n - 1;
    return (i64)f;
}

template <int G>
__device__ __forceinline__ void scan_window(Grp<G>& g, lds_u64 s, int off, int cap, i64 base, int& lcp, bool& less) {
    u64 k = g.sa[base + g.t].key, p = g.sa[base + g.t].pos;
    cmp_entry(g, s, off, cap, k, p, lcp, less);
    g.st[ST_WINDOWS] = g.st[ST_WINDOWS] + 1;
}

// group-uniform single-slot probe: every lane loads the same entry (one broadcast sector)
template <int G>
__device__ __forceinline__ void probe(Grp<G>& g, lds_u64 s, int off, int cap, i64 slot, int& lcp, bool& less) {
    u64 k = g.sa[slot].key, p = g.sa[slot].pos;
    cmp_entry(g, s, off, cap, k, p, lcp, less);
}

#ifndef EDGE_E
#define EDGE_E 2     // entries per lane when an SMEM interval is followed beyond the first window
#endif

// value of a per-lane array at window slot `idx` (slot j lives in lane j % G, register j / G)
template <int G, int E>
__device__ __forceinline__ int win_at(const Grp<G>& g, const int (&v)[E], int idx) {
    int r = g.shfl(v[0], idx & (G - 1));
#pragma unroll
    for (int e = 1; e < E; ++e) {
        int y = g.shfl(v[e], idx & (G - 1));
        if ((idx / G) == e) r = y;
    }
    return r;
}

template <int G, int E>
__device__ __forceinline__ u64 win_ballot(const Grp<G>& g, const bool (&p)[E]) {
    u64 m = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) m |= g.ballot(p[e]) << (e * G);
    return m;
}

// E*G consecutive slots from `base`: E coalesced loads per lane issued together, then the compares
template <int G, int E>
__device__ __forceinline__ void scan_wide(Grp<G>& g, lds_u64 s, int off, int cap, i64 base, int (&lcp)[E], bool (&less)[E]) {
    u64 ek[E], ep[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { ek[e] = g.sa[base + e * G + g.t].key; ep[e] = g.sa[base + e * G + g.t].pos; }
#pragma unroll
    for (int e = 0; e < E; ++e) cmp_entry(g, s, off, cap, ek[e], ep[e], lcp[e], less[e]);
    g.st[ST_WINDOWS] = g.st[ST_WINDOWS] + 1;
}

// lowest slot s_edge <= cur with [s_edge, cur] all sharing >= L bases with the query (cur does, cur > 0);
// nb = LCP of slot s_edge-1 (0 at the array start).  `need`: the caller only wants to know whether at least
// `need` more slots match (non-emitting searches, third round): stop extending once they do.
template <int G>
__device__ __forceinline__ void edge_down_impl(Grp<G>& g, lds_u64 s, int off, int L, i64 cur, i64 need, i64& s_edge, int& nb) {
    constexpr int EE = (EDGE_E * G > 64) ? 64 / G : EDGE_E;
    constexpr int WE = EE * G;
    const i64 start = cur;
    int iter = 0;
    for (;;) {
        i64 wb = cur - WE;
        if (wb < 0) wb = 0;
        int lcp[EE]; bool less[EE], ge[EE];
        scan_wide<G, EE>(g, s, off, L, wb, lcp, less);
#pragma unroll
        for (int e = 0; e < EE; ++e) ge[e] = lcp[e] >= L;
        const int ncur = (int)(cur - wb);                 // slots [0,ncur) lie below cur
        const u64 z = (~win_ballot<G, EE>(g, ge)) & ((ncur >= 64) ? ~0ull : ((1ull << ncur) - 1ull));
        if (z) {
            const int hz = 63 - __clzll((long long)z);
            s_edge = wb + hz + 1;
            nb = win_at<G, EE>(g, lcp, hz);
            return;
        }
        cur = wb;
        if (cur == 0) { s_edge = 0; nb = 0; return; }
        if (start - cur >= need) { s_edge = cur; nb = L; return; }
        if (++iter >= 2) break;
    }
    // large interval: gallop with single-slot probes, bisect, then one window for the exact edge
    i64 good = cur, bad = -1, step = 4 * WE;
    for (;;) {
        i64 p = good - step;
        if (p < 0) p = 0;
        int lcp; bool less;
        probe(g, s, off, L, p, lcp, less);
        if (lcp >= L) { good = p; if (p == 0) break; if (start - good >= need) { s_edge = good; nb = L; return; } step <<= 1; }
        else { bad = p; break; }
    }
    if (bad < 0) { s_edge = 0; nb = 0; return; }
    while (good - bad > WE) {
        i64 mid = bad + (good - bad) / 2;
        int lcp; bool less;
        probe(g, s, off, L, mid, lcp, less);
        if (lcp >= L) good = mid; else bad = mid;
    }
    {
        i64 wb = good - WE;                               // >= bad - ... : window [wb, good) contains bad
        if (wb < 0) wb = 0;
        int lcp[EE]; bool less[EE], ge[EE];
        scan_wide<G, EE>(g, s, off, L, wb, lcp, less);
#pragma unroll
        for (int e = 0; e < EE; ++e) ge[e] = lcp[e] >= L;
        const int ncur = (int)(good - wb);
        const u64 z = (~win_ballot<G, EE>(g, ge)) & ((ncur >= 64) ? ~0ull : ((1ull << ncur) - 1ull));
        const int hz = 63 - __clzll((long long)z);
        s_edge = wb + hz + 1;
        nb = win_at<G, EE>(g, lcp, hz);
    }
}

// highest slot e_edge >= cur with [cur, e_edge] all matching; nb = LCP of slot e_edge+1 (0 at the end)
template <int G>
__device__ __forceinline__ void edge_up_impl(Grp<G>& g, lds_u64 s, int off, int L, i64 cur, i64 need, i64& e_edge, int& nb) {
    constexpr int EE = (EDGE_E * G > 64) ? 64 / G : EDGE_E;
    constexpr int WE = EE * G;
    constexpr u64 WEFULL = (WE == 64) ? ~0ull : ((1ull << WE) - 1ull);
    const i64 n = g.n;
    const i64 start = cur;
    int iter = 0;
    for (;;) {
        i64 wb = cur + 1;                                 // window [wb, wb+WE) clipped to the array
        if (wb > n - WE) wb = n - WE;
        int lcp[EE]; bool less[EE], ge[EE];
        scan_wide<G, EE>(g, s, off, L, wb, lcp, less);
#pragma unroll
        for (int e = 0; e < EE; ++e) ge[e] = lcp[e] >= L;
        const int first = (int)(cur + 1 - wb);            // slots [first, WE) lie above cur
        const u64 z = (~win_ballot<G, EE>(g, ge)) & WEFULL & ~((1ull << first) - 1ull);
        if (z) {
            const int lz = __ffsll((long long)z) - 1;
            e_edge = wb + lz - 1;
            nb = win_at<G, EE>(g, lcp, lz);
            return;
        }
        cur = wb + WE - 1;
        if (cur == n - 1) { e_edge = n - 1; nb = 0; return; }
        if (cur - start >= need) { e_edge = cur; nb = L; return; }
        if (++iter >= 2) break;
    }
    i64 good = cur, bad = -1, step = 4 * WE;
    for (;;) {
        i64 p = good + step;
        if (p > n - 1) p = n - 1;
        int lcp; bool less;
        probe(g, s, off, L, p, lcp, less);
        if (lcp >= L) { good = p; if (p == n - 1) break; if (good - start >= need) { e_edge = good; nb = L; return; } step <<= 1; }
        else { bad = p; break; }
    }
    if (bad < 0) { e_edge = n - 1; nb = 0; return; }
    while (bad - good > WE) {
        i64 mid = good + (bad - good) / 2;
        int lcp; bool less;
        probe(g, s, off, L, mid, lcp, less);
        if (lcp >= L) good = mid; else bad = mid;
    }
    {
        i64 wb = good + 1;                                // window (good, good+WE] contains bad
        if (wb > n - WE) wb = n - WE;
        int lcp[EE]; bool less[EE], ge[EE];
        scan_wide<G, EE>(g, s, off, L, wb, lcp, less);
#pragma unroll
        for (int e = 0; e < EE; ++e) ge[e] = lcp[e] >= L;
        const int first = (int)(good + 1 - wb);
        const u64 z = (~win_ballot<G, EE>(g, ge)) & WEFULL & ~((1ull << first) - 1ull);
        const int lz = __ffsll((long long)z) - 1;
        e_edge = wb + lz - 1;
        nb = win_at<G, EE>(g, lcp, lz);
    }
}

// partition point outside the first window: gallop away from the prediction, bisect, return the base of a
// window that contains the partition point (or touches the array end it lies beyond)
template <int G>
__device__ __forceinline__ i64 relocate_impl(Grp<G>& g, lds_u64 s, int off, int vlen, i64 base, bool above) {
    const i64 n = g.n;
    // one loop for both directions: `lo` is a slot known to sort before the query (-1: none yet),
    // `hi` a slot known not to (n: none yet)
    i64 lo = above ? base + G - 1 : -1, hi = above ? n : base, step = G;
    for (;;) {
        i64 p = above ? lo + step : hi - step;
        if (p > n - 1) p = n - 1;
        if (p < 0) p = 0;
        int l2; bool ls;
        probe(g, s, off, vlen, p, l2, ls);
        if (ls) lo = p; else hi = p;
        if (above ? (!ls || p == n - 1) : (ls || p == 0)) break;
        step <<= 1;
    }
    if (hi == n) return n - G;            // every suffix sorts before the query
    if (lo < 0) return 0;                 // none does
    while (hi - lo >= G) {
        i64 mid = lo + (hi - lo) / 2;
        int l2; bool ls;
        probe(g, s, off, vlen, mid, l2, ls);
        if (ls) lo = mid; else hi = mid;
    }
    i64 b = hi - G + 1;                   // window [b, hi] contains lo (hi - lo <= G-1)
    if (b < 0) b = 0;
    if (b > n - G) b = n - G;
    return b;
}

// Cold paths (run for a minority of searches) are real function calls with by-value arguments and results: their
// register needs stay out of the hot loop's allocation, which is what decides the kernel's occupancy.
struct EdgeRes {
    i64 edge;
    int nb;
};
#ifndef COLD_ATTR
#define COLD_ATTR __forceinline__   // real calls cost 3x (scratch frames): measured, keep the cold paths inline
#endif
template <int G>
__device__ COLD_ATTR EdgeRes edge_down(Grp<G>& g, lds_u64 s, int off, int L, i64 cur, i64 need) {
    EdgeRes r;
    edge_down_impl(g, s, off, L, cur, need, r.edge, r.nb);
    return r;
}
template <int G>
__device__ COLD_ATTR EdgeRes edge_up(Grp<G>& g, lds_u64 s, int off, int L, i64 cur, i64 need) {
    EdgeRes r;
    edge_up_impl(g, s, off, L, cur, need, r.edge, r.nb);
    return r;
}
template <int G>
__device__ COLD_ATTR i64 relocate(Grp<G>& g, lds_u64 s, int off, int vlen, i64 base, bool above) {
    return relocate_impl(g, s, off, vlen, base, above);
}

// The one search primitive.  Semantics of mem_search / right_smem_search (and the _tradeoff twins):
//   maxLCP = longest prefix of the query (<= vlen bases) occurring in the text;
//   mode 0: L = maxLCP.
//   mode 1: L = largest l <= maxLCP whose SA interval holds >= min_intv suffixes; [start,count) = interval
//           (:2365-2574, :2902-2942).
//   mode 2: third round (:1199-1281): walk the levels maxLCP = L0 > L1 > ... until the interval holds
//           >= min_intv suffixes or the next level is shorter than min_seed_len.
#ifndef WIN_E
#define WIN_E 3      // suffix-array entries per lane in the first window: window = WIN_E * G slots
#endif

template <int G>
__device__ __forceinline__ Res do_search(Grp<G>& g, const Req& q, int msl) {
    constexpr int E = (WIN_E * G > 64) ? 64 / G : WIN_E;   // entries per lane (the window mask is 64 bits)
    constexpr int W = E * G;                               // first-window width in slots
    constexpr u64 WFULL = (W == 64) ? ~0ull : ((1ull << W) - 1ull);
    const i64 n = g.n;
    lds_u64 s = q.rc ? g.rc : g.fw;
    const int off = q.off, vlen = q.vlen;
    u64 key = ext_l(s, off);
    if (vlen < 32) key |= (~0ull) >> (2 * vlen);          // T-pad short queries like Tokenization (:813-817)
    i64 pos = rmi_lookup(g, key);
    i64 base = pos - W / 2;
    if (base < 0) base = 0;
    if (base > n - W) base = n - W;
    int lcp[E];
    bool less[E];
    // first window: WIN_E coalesced loads of G entries each, issued together
    scan_wide<G, E>(g, s, off, vlen, base, lcp, less);
    u64 m = win_ballot(g, less);
    const bool above = (m == WFULL) && base + W < n;
    const bool below = (m == 0) && base > 0;
    if (above || below) {
        i64 b = relocate(g, s, off, vlen, above ? base + W - G : base, above);   // G slots containing the partition point
        base = b - (W - G) / 2;
        if (base < 0) base = 0;
        if (base > n - W) base = n - W;
        scan_wide<G, E>(g, s, off, vlen, base, lcp, less);
        m = win_ballot(g, less);
    }
    // slots [0,P) sort before the query; the longest match is at one of the two boundary neighbours
    const int P = __popcll(m);
    const int la = P > 0 ? win_at(g, lcp, P - 1) : -1;
    const int lb = P < W ? win_at(g, lcp, P < W ? P : W - 1) : -1;
    const int c = (la >= lb) ? P - 1 : P;
    int L = la >= lb ? la : lb;
    Res out;
    out.L = L;
    out.start = base + c;
    out.count = 1;
    out.emit = false;
    if (q.mode == 0) return out;
    if (q.mode == 2 && L < msl) return out;               // :1204-1208
    // interval at level L from the window; extended beyond it only when the run touches a window edge
    i64 s_edge = base + c, e_edge = base + c;
    int nb_lo = 0, nb_hi = 0;
    bool need_lo = true, need_hi = true;
    // third-round bookkeeping (previous level's interval) lives in LDS: rarely touched, 4 registers saved
    i64 cnt, emit_s = s_edge;
    int match_len = L;
    bool have_last = false;
    // searches whose interval is never emitted at the level where the walk stops (left extensions, the third round)
    // only need to know that the interval reached min_intv: following it further is wasted traffic
    const i64 need_more = q.exact ? ((i64)1 << 62) : (i64)q.min_intv;
    for (;;) {
        // Levels usually end inside the first window, whose LCPs (computed against the whole query) are still in
        // registers: resolve the edge from them and touch memory only when the run leaves the window.
        if (need_lo || need_hi) {
            bool ge[E];
#pragma unroll
            for (int e = 0; e < E; ++e) ge[e] = lcp[e] >= L;
            const u64 zm = (~win_ballot(g, ge)) & WFULL;       // slots of the window that do NOT reach level L
            if (need_lo) {
                bool solved = false;
                if (s_edge > base && s_edge <= base + W) {
                    const int ncur = (int)(s_edge - base);      // slots [0,ncur) lie below the current edge
                    const u64 z = zm & ((ncur >= 64) ? ~0ull : ((1ull << ncur) - 1ull));
                    if (z) {
                        const int hz = 63 - __clzll((long long)z);
                        s_edge = base + hz + 1;
                        nb_lo = win_at(g, lcp, hz);
                        solved = true;
                    } else if (base == 0) { s_edge = 0; nb_lo = 0; solved = true; }
                    else s_edge = base;
                } else if (s_edge == 0) { nb_lo = 0; solved = true; }
                if (!solved) { const EdgeRes er = edge_down(g, s, off, L, s_edge, need_more - (e_edge - s_edge + 1)); s_edge = er.edge; nb_lo = er.nb; }
            }
            if (need_hi) {
                bool solved = false;
                if (e_edge >= base - 1 && e_edge < base + W - 1) {
                    const int first = (int)(e_edge + 1 - base);  // slots [first,W) lie above the current edge
                    const u64 z = zm & ~((1ull << first) - 1ull);
                    if (z) {
                        const int lz = __ffsll((long long)z) - 1;
                        e_edge = base + lz - 1;
                        nb_hi = win_at(g, lcp, lz);
                        solved = true;
                    } else if (base + W >= n) { e_edge = n - 1; nb_hi = 0; solved = true; }
                    else e_edge = base + W - 1;
                } else if (e_edge == n - 1) { nb_hi = 0; solved = true; }
                if (!solved) { const EdgeRes er = edge_up(g, s, off, L, e_edge, need_more - (e_edge - s_edge + 1)); e_edge = er.edge; nb_hi = er.nb; }
            }
        }
        cnt = e_edge - s_edge + 1;
        const int nxt = nb_lo > nb_hi ? nb_lo : nb_hi;
        if (q.mode == 1) {                                  // (:2568-2573, :2936-2940)
            if (cnt >= (i64)q.min_intv) { emit_s = s_edge; match_len = L; break; }
        } else {
            if (cnt >= (i64)q.min_intv) {                   // :1243-1251
                if (have_last) {
                    cnt = ((i64)g.st[ST_LAST_CNT_HI] << 32) | (u64)(unsigned)g.st[ST_LAST_CNT_LO];
                    emit_s = ((i64)g.st[ST_LAST_S_HI] << 32) | (u64)(unsigned)g.st[ST_LAST_S_LO];
                } else emit_s = s_edge;
                match_len = L + 1;
                break;
            }
            if (nxt < msl) { match_len = msl; emit_s = s_edge; break; }   // :1252-1258
            have_last = true;
            g.st[ST_LAST_CNT_LO] = (int)(unsigned)(cnt & 0xffffffffll);
            g.st[ST_LAST_CNT_HI] = (int)(cnt >> 32);
            g.st[ST_LAST_S_LO] = (int)(unsigned)(s_edge & 0xffffffffll);
            g.st[ST_LAST_S_HI] = (int)(s_edge >> 32);
        }
        L = nxt;
        need_lo = nb_lo >= L && s_edge > 0;
        need_hi = nb_hi >= L && e_edge < n - 1;
    }
    if (q.mode == 2) {
        out.emit = cnt < (i64)q.min_intv;                  // :1265
        if (match_len < msl) match_len = msl;
    }
    out.L = match_len;
    out.start = emit_s;
    out.count = cnt;
    return out;
}

__device__ __forceinline__ bool is_n(lds_u64 mask, int i) { return (mask[i >> 6] >> (i & 63)) & 1ull; }

// first ambiguous base at/after `from` (Tokenization's *ambiguous_pos, :795-901)
__device__ __forceinline__ int first_n(lds_u64 mask, bool has_n, int from, int l_seq) {
    if (!has_n) return l_seq;
    int w = from >> 6;
    u64 m = mask[w] & (~0ull << (from & 63));
    const int nw = (l_seq + 63) >> 6;
    for (;;) {
        if (m) {
            int p = w * 64 + __ffsll((long long)m) - 1;
            return p < l_seq ? p : l_seq;
        }
        if (++w >= nw) return l_seq;
        m = mask[w];
    }
}

// ---- the search kernel ---------------------------------------------------------------------------------------
template <int G>
#ifndef SEED_MIN_WAVES
#define SEED_MIN_WAVES 5
#endif
__global__ void __launch_bounds__(BLOCK, SEED_MIN_WAVES) k_seed(SeedArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int GROUPS = BLOCK / G;
    const int lane = threadIdx.x & 63;
    const int gib = threadIdx.x / G;
    const int stride = A.geo.stride, W = A.geo.W, MW = A.geo.MW;
    u64* rd = reinterpret_cast<u64*>(smem_raw) + (size_t)gib * stride;
    lds_u64 rdl = (lds_u64)rd;
    // per group after the packed read: the SMEM ring (3 ints per entry) and the cold part of the read's state
    lds_int ring = (lds_int)(reinterpret_cast<int*>(smem_raw + (size_t)GROUPS * stride * 8) + (size_t)gib * (3 * A.lcap + ST_WORDS));
    lds_int sm_start = ring;
    lds_int sm_end = ring + A.lcap;
    lds_int sm_cnt = ring + 2 * A.lcap;
    lds_int st = ring + 3 * A.lcap;
    Grp<G> g;
    g.st = st;
    g.sa = (glb_ent)A.I.sa;
    g.pac = (glb_u64)A.I.pac;
    g.l2 = (glb_rmi)A.I.l2;
    g.l1 = (glb_rmi)A.I.l1;
    g.n = A.I.n;
    g.shift = A.I.shift;
    g.fw = rdl;
    g.rc = rdl + W;
    g.nfw = rdl + 2 * W;
    g.nrc = rdl + 2 * W + MW;
    g.t = threadIdx.x & (G - 1);
    g.gbase = lane & ~(G - 1);
    const int cap = A.cap, lcap = A.lcap;
    const int hits_per_smem = A.opt.hits_per_smem;
    // reads are pulled with one ticket atomic per read (measured 9 % faster than a static round-robin deal: reads differ
    // 3x in cost); the read length travels in the packed record, so no dependent offset loads follow the ticket
    for (;;) {
        unsigned long long ticket = 0;
        if (g.t == 0) ticket = atomicAdd(&A.counters[0], 1ull);
        ticket = __shfl(ticket, g.gbase);
        if (ticket >= (unsigned long long)A.nreads) break;
        const i64 rid = A.pending ? A.pending[ticket] : (i64)ticket;
        const u64* src = A.packed + rid * stride;
        const int l_seq = (int)src[stride - 1];          // k_pack_reads stores the length in the last word
        SlotRec* slots = A.slots + (i64)ticket * cap;
        if (l_seq <= 0 || l_seq > MAX_READ_LEN) {
            // the reference exits on reads longer than LEARNED_MAX_READ_LEN (src/bwamem.cpp:1259-1262);
            // here such a read yields no seeds and is flagged through slot_cnt = -1
            if (g.t == 0) { A.slot_cnt[rid] = l_seq > MAX_READ_LEN ? -1 : 0; A.slot_hits[rid] = 0; A.slot_loc[rid] = 0; }
            continue;
        }
        // ---- stage the packed read in LDS (coalesced 8-byte loads) ------------------------------------
        bool any_n = false;
        for (int k = g.t; k < stride - 1; k += G) {
            u64 v = src[k];
            rd[k] = v;
            if (k >= 2 * W && k < 2 * W + MW) any_n |= (v != 0);
        }
        const bool has_n = g.ballot(any_n) != 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- per-read state (group-uniform) ----------------------------------------------------------------
        int pivot = 0;
        int msl = A.opt.min_seed_len, min_intv = 1;
        int pc = PC_ALLPOS_TOP;
        for (int k = 0; k < ST_WORDS; ++k) st[k] = 0;
#define SET_PIVOT(p_) do { pivot = (p_); } while (0)
#define l_pivot (l_seq - 1 - pivot)
#define FLAG(f_) ((st[ST_FLAGS] & (f_)) != 0)
#define SETFLAG(f_, v_) do { st[ST_FLAGS] = (v_) ? (st[ST_FLAGS] | (f_)) : (st[ST_FLAGS] & ~(f_)); } while (0)
        for (;;) {
            // ---- control: advance to the next search request -------------------------------------------------
            Req q;
            bool have = false;
            while (!have && pc != PC_DONE) {
                switch (pc) {
                case PC_ALLPOS_TOP:   // Learned_getSMEMsAllPosOneThread loop head (:916) + step1 entry (:1691-1723)
                    if (pivot >= l_seq || ++st[ST_AP_GUARD] > 4 * l_seq + 16) { pc = PC_R3_INIT; break; }
                    st[ST_BEFORE] = st[ST_N_SMEMS]; st[ST_SM_BASE] = st[ST_BEFORE]; SETFLAG(F_REC, true);
                    if (is_n(g.nfw, pivot)) {
                        if (l_seq - pivot < msl) SET_PIVOT(l_seq); else SET_PIVOT(pivot + 1);
                        pc = PC_AFTER_STEP1;
                    } else if (pivot != 0 && !is_n(g.nfw, pivot - 1)) {
                        st[ST_ZZ_NEXT] = l_seq; SETFLAG(F_ZZ_CHECK, true); SETFLAG(F_ZZ_RET_ONEPOS, false); st[ST_ZZ_SP] = pivot; st[ST_ZZ_GUARD] = 0;
                        pc = PC_ZZ_TOP;
                    } else {
                        q.kind = K_S1_RIGHT; q.exact = true; q.rc = false; q.off = pivot;
                        q.vlen = first_n(g.nfw, has_n, pivot, l_seq) - pivot; q.min_intv = min_intv; q.mode = 1;
                        have = true;
                    }
                    break;
                case PC_ZZ_TOP:       // zig-zag loop head (:1724-1737, :1969)
                    if (st[ST_ZZ_SP] >= st[ST_ZZ_NEXT] || ++st[ST_ZZ_GUARD] > 4 * l_seq + 16) { pc = PC_ZZ_END; break; }
                    if (FLAG(F_ZZ_CHECK) && is_n(g.nfw, st[ST_ZZ_SP])) {
                        if (l_seq - st[ST_ZZ_SP] < msl) { SET_PIVOT(l_seq); st[ST_ZZ_SP] = l_seq; }
                        else { st[ST_ZZ_SP] += 1; SET_PIVOT(pivot + 1); }
                        break;
                    }
                    q.kind = K_ZZ_LEFT; q.exact = false; q.rc = true; q.off = l_pivot;
                    q.vlen = first_n(g.nrc, has_n, l_pivot, l_seq) - l_pivot; q.min_intv = min_intv;
                    q.mode = min_intv != 1 ? 1 : 0;
                    have = true;
                    break;
                case PC_ZZ_RIGHT:
                    q.kind = K_ZZ_RIGHT; q.exact = true; q.rc = false; q.off = pivot;
                    q.vlen = first_n(g.nfw, has_n, pivot, l_seq) - pivot; q.min_intv = min_intv; q.mode = 1;
                    have = true;
                    break;
                case PC_ZZ_END:       // set_forward_pivot(raux, next_pivot) (:1893, :2125)
                    SET_PIVOT(st[ST_ZZ_NEXT]);
                    pc = FLAG(F_ZZ_RET_ONEPOS) ? PC_R2_AFTER : PC_AFTER_STEP1;
                    break;
                case PC_AFTER_STEP1:  // re-seeding loop entry (:921-923)
                    SETFLAG(F_REC, false);
                    st[ST_AFTER] = st[ST_N_SMEMS];
                    if (A.opt.rounds < 2) { pc = PC_ALLPOS_TOP; break; }
                    if (FLAG(F_LDS_OVF)) { pc = PC_DONE; break; }   // re-run in the next tier (bigger LDS ring)
                    st[ST_R2_K] = st[ST_BEFORE];
                    pc = PC_R2_LOOP;
                    break;
                case PC_R2_LOOP: {    // (:923-947) + OnePos entry (:1917-1930)
                    if (st[ST_R2_K] >= st[ST_AFTER]) { pc = PC_ALLPOS_TOP; break; }
                    const int k = st[ST_R2_K]++ - st[ST_BEFORE];
                    st[ST_R2_NEXT] = pivot; st[ST_R2_SAVED] = min_intv;
                    const int qbeg = sm_start[k], qend = sm_end[k], cnt = sm_cnt[k];
                    if (qend - qbeg < A.opt.split_len || cnt > A.opt.split_width) { SET_PIVOT(st[ST_R2_NEXT]); break; }
                    SET_PIVOT((qbeg + qend) >> 1);
                    min_intv = cnt + 1;
                    if (is_n(g.nfw, pivot)) {
                        if (l_seq - pivot < msl) SET_PIVOT(l_seq); else SET_PIVOT(pivot + 1);
                        pc = PC_R2_AFTER;
                    } else if (pivot != 0 && !is_n(g.nfw, pivot - 1)) {
                        q.kind = K_OP_MEM; q.exact = false; q.rc = false; q.off = pivot;
                        q.vlen = first_n(g.nfw, has_n, pivot, l_seq) - pivot; q.min_intv = min_intv;
                        q.mode = min_intv != 1 ? 1 : 0;
                        have = true;
                    } else {
                        q.kind = K_OP_SMEM; q.exact = true; q.rc = false; q.off = pivot;
                        q.vlen = first_n(g.nfw, has_n, pivot, l_seq) - pivot; q.min_intv = min_intv; q.mode = 1;
                        have = true;
                    }
                    break;
                }
                case PC_R2_AFTER:     // (:945-946)
                    min_intv = st[ST_R2_SAVED];
                    SET_PIVOT(st[ST_R2_NEXT]);
                    pc = PC_R2_LOOP;
                    break;
                case PC_R3_INIT:      // src/bwamem.cpp:1385-1394
                    if (A.opt.rounds >= 3 && A.opt.max_mem_intv > 0 && !FLAG(F_LDS_OVF)) {
                        min_intv = A.opt.max_mem_intv;
                        msl = A.opt.min_seed_len + 1;
                        SET_PIVOT(0);
                        pc = PC_R3_TOP;
                    } else pc = PC_DONE;
                    break;
                case PC_R3_TOP: {     // Learned_bwtSeedStrategyAllPosOneThread loop head (:982-1012)
                    if (!(pivot < l_seq - msl + 1)) { pc = PC_DONE; break; }
                    if (is_n(g.nfw, pivot)) { SET_PIVOT(pivot + 1); break; }
                    const int valid = first_n(g.nfw, has_n, pivot, l_seq) - pivot;
                    if (valid < msl) { SET_PIVOT(pivot + valid); break; }
                    q.kind = K_R3; q.exact = false; q.rc = false; q.off = pivot; q.vlen = valid; q.min_intv = min_intv; q.mode = 2;
                    have = true;
                    break;
                }
                default: pc = PC_DONE; break;
                }
            }
            if (!have) break;
            // ---- the single search call site ----------------------------------------------------------------------
            st[ST_SEARCHES] = st[ST_SEARCHES] + 1;
            const Res r = do_search(g, q, msl);
            // ---- apply -------------------------------------------------------------------------------------------
            bool emit = false;
            int e_start = pivot, e_end = pivot;
            switch (q.kind) {
            case K_S1_RIGHT:          // (:1852-1893)
                emit = r.L >= msl; e_end = pivot + r.L;
                break;
            case K_ZZ_LEFT:           // (:1774-1777)
                SET_PIVOT(pivot - r.L + 1);
                pc = (st[ST_ZZ_NEXT] - pivot < msl) ? PC_ZZ_END : PC_ZZ_RIGHT;
                break;
            case K_ZZ_RIGHT:          // (:1846-1848)
                emit = r.L >= msl; e_end = pivot + r.L;
                break;
            case K_OP_MEM:            // (:1967-1969)
                st[ST_ZZ_NEXT] = pivot + r.L; SETFLAG(F_ZZ_CHECK, false); SETFLAG(F_ZZ_RET_ONEPOS, true); st[ST_ZZ_SP] = pivot; st[ST_ZZ_GUARD] = 0;
                pc = PC_ZZ_TOP;
                break;
            case K_OP_SMEM:           // (:2093-2125)
                emit = r.L >= msl; e_end = pivot + r.L;
                break;
            case K_R3:                // (:1204-1208, :1265-1281)
                if (r.L < msl && !r.emit) { /* too short: advance by min_seed_len */ }
                emit = r.emit; e_end = pivot + r.L;
                break;
            }
            if (emit) {               // kv_push of mem_tl + hits (:2639-2657, :1266-1277)
                if (st[ST_N_SMEMS] < cap && g.t == 0) {
                    SlotRec sr;
                    sr.start = e_start; sr.end = e_end; sr.sa_start = r.start; sr.count = r.count;
                    slots[st[ST_N_SMEMS]] = sr;
                }
                if (FLAG(F_REC)) {
                    const int k = st[ST_N_SMEMS] - st[ST_SM_BASE];
                    if (k < lcap) {
                        // group-uniform redundant LDS stores (every lane writes the same value): no hand-off needed
                        sm_start[k] = e_start;
                        sm_end[k] = e_end;
                        sm_cnt[k] = r.count > (i64)INT_MAX ? INT_MAX : (int)r.count;
                    } else SETFLAG(F_LDS_OVF, true);
                }
                st[ST_N_SMEMS] = st[ST_N_SMEMS] + 1;
                i64 h = r.count;
                if (hits_per_smem > 0 && h > hits_per_smem) h = hits_per_smem;
                h += ((i64)st[ST_HITS_HI] << 32) | (u64)(unsigned)st[ST_HITS_LO];
                st[ST_HITS_LO] = (int)(unsigned)(h & 0xffffffffll);
                st[ST_HITS_HI] = (int)(h >> 32);
            }
            switch (q.kind) {
            case K_S1_RIGHT: SET_PIVOT(pivot + r.L); pc = PC_AFTER_STEP1; break;
            case K_ZZ_RIGHT: st[ST_ZZ_SP] = pivot + r.L; SET_PIVOT(st[ST_ZZ_SP]); pc = PC_ZZ_TOP; break;
            case K_OP_SMEM: SET_PIVOT(pivot + r.L); pc = PC_R2_AFTER; break;
            case K_R3: SET_PIVOT(pivot + (r.L < msl ? msl : r.L)); pc = PC_R3_TOP; break;
            default: break;
            }
        }
#undef SET_PIVOT
#undef l_pivot
        if (g.t == 0) {
            const int n_smems = st[ST_N_SMEMS];
            const unsigned searches = (unsigned)st[ST_SEARCHES];
            const i64 n_hits = ((i64)st[ST_HITS_HI] << 32) | (u64)(unsigned)st[ST_HITS_LO];
            const bool ovf = n_smems > cap || FLAG(F_LDS_OVF);
            A.slot_cnt[rid] = ovf ? 0 : n_smems;
            A.slot_hits[rid] = ovf ? 0 : n_hits;
            A.slot_loc[rid] = ((i64)A.tier << 40) | (i64)ticket;
            if (ovf) A.ovf_list[atomicAdd(&A.counters[2], 1ull)] = rid;
            else { atomicAdd(&A.counters[1], (unsigned long long)searches); atomicAdd(&A.counters[3], (unsigned long long)(unsigned)st[ST_WINDOWS]); }
        }
#undef FLAG
#undef SETFLAG
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace seedk
