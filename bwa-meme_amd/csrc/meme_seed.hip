// Learned-index seeding on MI355X: P-RMI lookup -> suffix-array last-mile search -> SMEM enumeration.
//
// Replaces the per-read loop body of mem_kernel1_core_Learned() (reference src/bwamem.cpp:1249-1394):
//   Learned_getSMEMsAllPosOneThread        src/LearnedIndex_seeding.cpp:913-972   (rounds 1 and 2)
//     Learned_getSMEMsOnePosOneThread_step1                               :1691-1894
//     Learned_getSMEMsOnePosOneThread                                     :1897-2126
//   Learned_bwtSeedStrategyAllPosOneThread[_mem_tradeoff]                 :974-1466  (round 3)
//   mem_search / right_smem_search [+ _tradeoff]                          :2131-4189
//   learned_index_lookup                                                  :186-210
//   compare_read_and_ref_binary*                                          :226-601
//
// Design (MI355X-first, not a translation):
//  * A read is owned by a group of G=16 lanes (4 reads per 64-lane wavefront).  Control flow and the
//    pivot state machine are group-uniform (every lane of the group holds the same scalar state);
//    only the suffix-array probe is lane-parallel.
//  * The unit of memory traffic is a *window*: G consecutive 16-byte suffix-array entries = one
//    coalesced 256-byte read.  Each lane compares its entry's 64-bit key (and, only if all 32 bases
//    agree, 2-bit reference words) with the read; one wave ballot gives the partition point and the
//    longest-common-prefix values needed for the SMEM hit interval, so the reference's chain of
//    ~log2(err)+linear dependent single-entry probes collapses to 1 window load in the common case.
//  * The learned model is used as a hint only: if the partition point is not inside
//    the first window the group gallops away from the prediction and bisects with group-uniform
//    single-entry probes (a broadcast load), then takes one final window.  Model error bounds are
//    not needed, so any parameter file the reference loads is accepted.
//  * The read (both strands, 2 bits/base, first base in the top bits of each u64) and its N-mask live
//    in LDS; arbitrary-offset 32-base query words are funnel-shifted out of two LDS words, which
//    replaces the reference's 8 pre-shifted copies of every read (src/bwamem.cpp:1283-1344).
//  * Output = per-read SMEM slots {start,end,sa_start,count}; a second kernel computes offsets and
//    gathers hit positions (ascending SA order, as mem_chain_Learned expects, src/bwamem.cpp:1154-1160).
#include <limits.h>
#include <string.h>

#include "meme_common.h"

namespace {

constexpr int G = 16;                    // lanes per read
constexpr unsigned GFULL = (1u << G) - 1;
constexpr int MAXW = 17;                 // strand words: 512 bases + 1 so extract32 may touch w[k+1]
constexpr int MASKW = 8;                 // N-mask words (512 bits)
constexpr int MAX_READ_LEN = 500;        // LEARNED_MAX_READ_LEN, reference src/macro.h / bwamem.cpp:1259
constexpr int BLOCK = 256;
constexpr int GROUPS_PER_BLOCK = BLOCK / G;

struct SlotRec {          // search-kernel output, one per SMEM
    int32_t start, end;
    i64 sa_start;
    i64 count;
};

struct SeedArgs {
    DevIndex I;
    const uint8_t* reads;
    const i64* read_off;
    i64 nreads;
    meme_seed_opt opt;
    SlotRec* slots;        // [nreads * cap]
    int* slot_cnt;         // [nreads]  SMEMs the read produced (may exceed cap -> overflow)
    i64* slot_hits;        // [nreads]  hits the read will materialise
    i64* slot_loc;         // [nreads]  (tier << 40) | block index inside that tier's slot array
    const i64* pending;    // list of read ids to re-process in an overflow tier, else nullptr
    i64* ovf_list;         // reads whose SMEMs did not fit this tier (re-run in the next tier)
    int cap;               // SMEM slots per read in this tier
    int lcap;              // LDS ring entries per read (SMEMs of the current first-round pass)
    int tier;
    unsigned long long* counters;   // [0] ticket, [1] searches, [2] overflowed reads
};

constexpr int N_TIERS = 3;
constexpr int TIER_CAP[N_TIERS] = {64, 2048, 65536};   // tier 0 is tunable ("smem_cap")
constexpr int LDS_RING_MAX = 512;                      // >= MAX_READ_LEN: one first-round pass cannot emit more

struct GroupLds {
    u64 fw[MAXW + 1];
    u64 rc[MAXW + 1];
    u64 nfw[MASKW];
    u64 nrc[MASKW];
};

// per-group scalar state (replicated in every lane of the group)
struct RState {
    const DevIndex* I;
    GroupLds* L;
    int* sm_start;   // LDS ring: SMEMs of the current first-round pass (round 2 re-reads them)
    int* sm_end;
    int* sm_cnt;
    SlotRec* slots;  // global slots of this read
    int cap, lcap;
    int sm_base;     // n_smems when the current first-round pass started
    bool rec;        // record emissions in the LDS ring
    bool lds_ovf;
    int l_seq;
    int min_seed_len, min_intv;
    int pivot, l_pivot;
    int n_smems;
    i64 n_hits;
    int hits_per_smem;
    bool has_n;
    unsigned searches;
    int t;           // lane within group
    int gbase;       // first lane of the group within the wave
};

__device__ __forceinline__ unsigned gballot(bool p, int gbase) {
    u64 b = __ballot(p);
    return (unsigned)(b >> gbase) & GFULL;
}

__device__ __forceinline__ int gshfl_i(int v, int src, int gbase) { return __shfl(v, gbase + src); }

// ---- compare: reference compare_read_and_ref_binary* (:226-601) -----------------------------------
// L = min(cap, n - pos).  lcp < L : less = ref base < read base.  lcp == L: less = (L < ref_len)
// (suffix continues: "exact", sorts before; text ends first: sorts after, as if T-padded).
__device__ __forceinline__ void cmp_entry(const DevIndex& I, const u64* s, int off, int cap, SaEnt e, int& lcp,
                                          bool& less) {
    i64 ref_len = I.n - (i64)e.pos;
    int L = ref_len < (i64)cap ? (int)ref_len : cap;
    u64 wr = e.key;
    int l = 0, k = 0;
    bool lt = false;
    for (;;) {
        u64 wq = extract32(s, off + 32 * k);
        u64 x = wr ^ wq;
        if (x) { l = 32 * k + (__clzll((long long)x) >> 1); lt = wr < wq; break; }
        l = 32 * (k + 1);
        if (l >= L) break;
        ++k;
        wr = extract32(I.pac, (i64)e.pos + 32 * k);
    }
    if (l >= L) { lcp = L; less = (i64)L < ref_len; }
    else { lcp = l; less = lt; }
}

// ---- learned_index_lookup (:186-210), same arithmetic (FP64 FMA, clamp), used as a hint ------------
__device__ __forceinline__ i64 rmi_lookup(const DevIndex& I, u64 key) {
    u64 m = I.shift >= 64 ? 0ull : key >> I.shift;
    RmiRec r = I.l2[m];
    double x = (double)key;
    double f = fma(r.slope, x, r.icpt);
    if (r.err >> 63) {
        u64 ps = (r.err >> 32) & 0x7fffffffull;
        double pn = (double)(r.err & 0xffffffffull) - 1.0;
        double c = f < 0.0 ? 0.0 : (f > pn ? pn : f);
        r = I.l1[ps + (u64)c];
        f = fma(r.slope, x, r.icpt);
    }
    double top = (double)I.n - 1.0;
    if (f < 0.0) return 0;
    if (f > top) return I.n - 1;
    return (i64)f;
}

struct SearchOut {
    int L;
    i64 start, count;
};

// window scan: lane t compares slot base+t
__device__ __forceinline__ void scan_window(const RState& R, const u64* s, int off, int cap, i64 base, int& lcp,
                                            bool& less) {
    SaEnt e = R.I->sa[base + R.t];
    cmp_entry(*R.I, s, off, cap, e, lcp, less);
}

// group-uniform single-slot probe (all lanes load the same entry: one broadcast sector)
__device__ __forceinline__ void probe(const RState& R, const u64* s, int off, int cap, i64 slot, int& lcp, bool& less) {
    SaEnt e = R.I->sa[slot];
    cmp_entry(*R.I, s, off, cap, e, lcp, less);
}

// lowest slot s_edge <= cur such that [s_edge, cur] all share >= L bases with the query; nb = LCP of
// slot s_edge-1 (0 at the array start).  cur matches, cur > 0.
__device__ void edge_down(const RState& R, const u64* s, int off, int L, i64 cur, i64& s_edge, int& nb) {
    int iter = 0;
    for (;;) {
        i64 wb = cur - G;
        if (wb < 0) wb = 0;
        int lcp; bool less;
        scan_window(R, s, off, L, wb, lcp, less);
        unsigned mm = gballot(lcp >= L, R.gbase);
        int ncur = (int)(cur - wb);                       // lanes [0,ncur) lie below cur
        unsigned z = (~mm) & ((1u << ncur) - 1u);
        if (z) {
            int hz = 31 - __clz((int)z);
            s_edge = wb + hz + 1;
            nb = gshfl_i(lcp, hz, R.gbase);
            return;
        }
        cur = wb;
        if (cur == 0) { s_edge = 0; nb = 0; return; }
        if (++iter >= 2) break;
    }
    // large interval: gallop with single-slot probes, then bisect, then one window for the exact edge
    i64 good = cur, bad = -1, step = 4 * G;
    for (;;) {
        i64 p = good - step;
        if (p < 0) p = 0;
        int lcp; bool less;
        probe(R, s, off, L, p, lcp, less);
        if (lcp >= L) { good = p; if (p == 0) break; step <<= 1; }
        else { bad = p; break; }
    }
    if (bad < 0) { s_edge = 0; nb = 0; return; }
    while (good - bad > G) {
        i64 mid = bad + (good - bad) / 2;
        int lcp; bool less;
        probe(R, s, off, L, mid, lcp, less);
        if (lcp >= L) good = mid; else bad = mid;
    }
    {
        i64 wb = good - G;                                // >= bad >= 0, window [wb, good) contains bad
        int lcp; bool less;
        scan_window(R, s, off, L, wb, lcp, less);
        unsigned z = (~gballot(lcp >= L, R.gbase)) & GFULL;
        int hz = 31 - __clz((int)z);
        s_edge = wb + hz + 1;
        nb = gshfl_i(lcp, hz, R.gbase);
    }
}

// highest slot e_edge >= cur with [cur, e_edge] all matching; nb = LCP of slot e_edge+1 (0 at the end)
__device__ void edge_up(const RState& R, const u64* s, int off, int L, i64 cur, i64& e_edge, int& nb) {
    const i64 n = R.I->n;
    int iter = 0;
    for (;;) {
        i64 wb = cur + 1;                                 // window [wb, wb+G) clipped to the array
        if (wb > n - G) wb = n - G;
        int lcp; bool less;
        scan_window(R, s, off, L, wb, lcp, less);
        unsigned mm = gballot(lcp >= L, R.gbase);
        int first = (int)(cur + 1 - wb);                  // lanes [first, G) lie above cur
        unsigned z = (~mm) & GFULL & ~((1u << first) - 1u);
        if (z) {
            int lz = __ffs((int)z) - 1;
            e_edge = wb + lz - 1;
            nb = gshfl_i(lcp, lz, R.gbase);
            return;
        }
        cur = wb + G - 1;
        if (cur == n - 1) { e_edge = n - 1; nb = 0; return; }
        if (++iter >= 2) break;
    }
    i64 good = cur, bad = -1, step = 4 * G;
    for (;;) {
        i64 p = good + step;
        if (p > n - 1) p = n - 1;
        int lcp; bool less;
        probe(R, s, off, L, p, lcp, less);
        if (lcp >= L) { good = p; if (p == n - 1) break; step <<= 1; }
        else { bad = p; break; }
    }
    if (bad < 0) { e_edge = n - 1; nb = 0; return; }
    while (bad - good > G) {
        i64 mid = good + (bad - good) / 2;
        int lcp; bool less;
        probe(R, s, off, L, mid, lcp, less);
        if (lcp >= L) good = mid; else bad = mid;
    }
    {
        i64 wb = good + 1;                                // window (good, good+G] contains bad
        if (wb > n - G) wb = n - G;
        int lcp; bool less;
        scan_window(R, s, off, L, wb, lcp, less);
        int first = (int)(good + 1 - wb);
        unsigned z = (~gballot(lcp >= L, R.gbase)) & GFULL & ~((1u << first) - 1u);
        int lz = __ffs((int)z) - 1;
        e_edge = wb + lz - 1;
        nb = gshfl_i(lcp, lz, R.gbase);
    }
}

// The one search primitive.  Semantics of mem_search / right_smem_search (and the _tradeoff twins):
//   maxLCP = longest prefix of the query (<= vlen bases) found anywhere in the text;
//   L      = largest l <= maxLCP whose SA interval holds >= min_intv suffixes;
//   [start, start+count) = that interval.   (count only evaluated when need_count)
// first_only: the third round needs maxLCP and the interval at that first level plus the
// neighbour LCPs to walk levels itself -> handled by seed_strategy through the same pieces.
struct Located {
    int L;         // maxLCP
    i64 s, e;      // interval at level L
    int nb_lo, nb_hi;
};

__device__ Located locate_and_first_level(RState& R, const u64* s, int off, int vlen, bool need_interval) {
    const DevIndex& I = *R.I;
    const i64 n = I.n;
    R.searches++;
    u64 key = extract32(s, off);
    if (vlen < 32) key |= (~0ull) >> (2 * vlen);         // T-pad short queries like Tokenization (:813-817)
    i64 pos = rmi_lookup(I, key);
    i64 base = pos - G / 2;
    if (base < 0) base = 0;
    if (base > n - G) base = n - G;
    int lcp; bool less;
    scan_window(R, s, off, vlen, base, lcp, less);
    unsigned m = gballot(less, R.gbase);
    if (m == GFULL && base + G < n) {
        // every slot of the window sorts before the query: partition point is above
        i64 lo = base + G - 1, hi = -1, step = G;
        for (;;) {
            i64 p = lo + step;
            if (p > n - 1) p = n - 1;
            int l2; bool ls;
            probe(R, s, off, vlen, p, l2, ls);
            if (ls) { lo = p; if (p == n - 1) break; step <<= 1; }
            else { hi = p; break; }
        }
        if (hi < 0) base = n - G;
        else {
            while (hi - lo >= G) {
                i64 mid = lo + (hi - lo) / 2;
                int l2; bool ls;
                probe(R, s, off, vlen, mid, l2, ls);
                if (ls) lo = mid; else hi = mid;
            }
            base = hi - G + 1;
            if (base < 0) base = 0;
        }
        scan_window(R, s, off, vlen, base, lcp, less);
        m = gballot(less, R.gbase);
    } else if (m == 0 && base > 0) {
        i64 hi = base, lo = -1, step = G;
        for (;;) {
            i64 p = hi - step;
            if (p < 0) p = 0;
            int l2; bool ls;
            probe(R, s, off, vlen, p, l2, ls);
            if (!ls) { hi = p; if (p == 0) break; step <<= 1; }
            else { lo = p; break; }
        }
        if (lo < 0) base = 0;
        else {
            while (hi - lo >= G) {
                i64 mid = lo + (hi - lo) / 2;
                int l2; bool ls;
                probe(R, s, off, vlen, mid, l2, ls);
                if (ls) lo = mid; else hi = mid;
            }
            base = lo;
            if (base > n - G) base = n - G;
        }
        scan_window(R, s, off, vlen, base, lcp, less);
        m = gballot(less, R.gbase);
    }
    // partition point inside [base, base+G] : lanes [0,P) sort before the query
    int P = __popc(m);
    int la = P > 0 ? gshfl_i(lcp, P - 1, R.gbase) : -1;
    int lb = P < G ? gshfl_i(lcp, P < G ? P : G - 1, R.gbase) : -1;
    int c = (la >= lb) ? P - 1 : P;
    Located out;
    out.L = la >= lb ? la : lb;
    out.s = out.e = base + c;
    out.nb_lo = out.nb_hi = 0;
    if (!need_interval) return out;
    // interval at level L from the window, extended outside it when the run touches a window edge
    const int L = out.L;
    unsigned mm = gballot(lcp >= L, R.gbase);
    unsigned below = (~mm) & ((1u << c) - 1u);
    unsigned above = (~mm) & GFULL & ~((2u << c) - 1u);
    if (below) {
        int hz = 31 - __clz((int)below);
        out.s = base + hz + 1;
        out.nb_lo = gshfl_i(lcp, hz, R.gbase);
    } else if (base == 0) { out.s = 0; out.nb_lo = 0; }
    else edge_down(R, s, off, L, base, out.s, out.nb_lo);
    if (above) {
        int lz = __ffs((int)above) - 1;
        out.e = base + lz - 1;
        out.nb_hi = gshfl_i(lcp, lz, R.gbase);
    } else if (base + G >= n) { out.e = n - 1; out.nb_hi = 0; }
    else edge_up(R, s, off, L, base + G - 1, out.e, out.nb_hi);
    return out;
}

// lower the level until the interval holds >= min_intv suffixes (:2365-2574, :2902-2942)
__device__ SearchOut search(RState& R, const u64* s, int off, int vlen, int min_intv, bool need_count) {
    bool need_interval = need_count || min_intv != 1;
    Located loc = locate_and_first_level(R, s, off, vlen, need_interval);
    SearchOut o;
    o.L = loc.L;
    o.start = loc.s;
    o.count = loc.e - loc.s + 1;
    if (!need_interval) return o;
    int L = loc.L;
    while (loc.e - loc.s + 1 < (i64)min_intv) {
        L = loc.nb_lo > loc.nb_hi ? loc.nb_lo : loc.nb_hi;
        if (loc.nb_lo >= L && loc.s > 0) edge_down(R, s, off, L, loc.s, loc.s, loc.nb_lo);
        if (loc.nb_hi >= L && loc.e < R.I->n - 1) edge_up(R, s, off, L, loc.e, loc.e, loc.nb_hi);
    }
    o.L = L;
    o.start = loc.s;
    o.count = loc.e - loc.s + 1;
    return o;
}

// ---- read state helpers ---------------------------------------------------------------------------
__device__ __forceinline__ void set_pivot(RState& R, int pivot) {   // set_forward_pivot (:68-71)
    R.pivot = pivot;
    R.l_pivot = R.l_seq - 1 - pivot;
}

__device__ __forceinline__ bool is_n(const u64* mask, int i) { return (mask[i >> 6] >> (i & 63)) & 1ull; }

// first ambiguous base at/after `from` (Tokenization's *ambiguous_pos, :795-901)
__device__ __forceinline__ int first_n(const RState& R, const u64* mask, int from) {
    if (!R.has_n) return R.l_seq;
    int w = from >> 6;
    u64 m = mask[w] & (~0ull << (from & 63));
    const int nw = (R.l_seq + 63) >> 6;
    for (;;) {
        if (m) {
            int p = w * 64 + __ffsll((long long)m) - 1;
            return p < R.l_seq ? p : R.l_seq;
        }
        if (++w >= nw) return R.l_seq;
        m = mask[w];
    }
}

__device__ __forceinline__ void emit(RState& R, int start, int end, i64 sa_start, i64 count) {
    if (R.n_smems < R.cap && R.t == 0) {
        SlotRec r;
        r.start = start; r.end = end; r.sa_start = sa_start; r.count = count;
        R.slots[R.n_smems] = r;
    }
    if (R.rec) {
        int k = R.n_smems - R.sm_base;
        if (k < R.lcap) {
            // group-uniform redundant LDS stores (same value from every lane): no cross-lane hand-off needed
            R.sm_start[k] = start;
            R.sm_end[k] = end;
            R.sm_cnt[k] = count > (i64)INT_MAX ? INT_MAX : (int)count;
        } else R.lds_ovf = true;
    }
    R.n_smems++;
    i64 h = count;
    if (R.hits_per_smem > 0 && h > R.hits_per_smem) h = R.hits_per_smem;
    R.n_hits += h;
}

// right_smem_search (:2131-2664)
__device__ int right_smem(RState& R) {
    int amb = first_n(R, R.L->nfw, R.pivot);
    SearchOut o = search(R, R.L->fw, R.pivot, amb - R.pivot, R.min_intv, true);
    if (o.L >= R.min_seed_len) emit(R, R.pivot, R.pivot + o.L, o.start, o.count);
    return o.L;
}

// mem_search (:2667-3204): right of pivot on the read, or right of l_pivot on the reverse complement
__device__ int mem_only(RState& R, bool right) {
    if (right) {
        int amb = first_n(R, R.L->nfw, R.pivot);
        return search(R, R.L->fw, R.pivot, amb - R.pivot, R.min_intv, false).L;
    }
    int amb = first_n(R, R.L->nrc, R.l_pivot);
    return search(R, R.L->rc, R.l_pivot, amb - R.l_pivot, R.min_intv, false).L;
}

// zig-zag of step1 / OnePos (:1724-1849, :1969-2084)
__device__ void zigzag(RState& R, int next_pivot, bool check_n) {
    int search_pivot = R.pivot;
    int guard = 0;
    while (search_pivot < next_pivot) {
        if (++guard > 4 * R.l_seq + 16) break;
        if (check_n && is_n(R.L->nfw, search_pivot)) {
            if (R.l_seq - search_pivot < R.min_seed_len) { set_pivot(R, R.l_seq); search_pivot = R.l_seq; }
            else { search_pivot += 1; set_pivot(R, R.pivot + 1); }
            continue;
        }
        int ss = mem_only(R, false);
        set_pivot(R, R.pivot - ss + 1);
        if (next_pivot - R.pivot < R.min_seed_len) break;
        ss = right_smem(R);
        search_pivot = R.pivot + ss;
        set_pivot(R, search_pivot);
    }
}

// Learned_getSMEMsOnePosOneThread_step1 (:1691-1894)
__device__ void step1(RState& R) {
    int next_pivot;
    if (is_n(R.L->nfw, R.pivot)) {
        if (R.l_seq - R.pivot < R.min_seed_len) set_pivot(R, R.l_seq);
        else set_pivot(R, R.pivot + 1);
        return;
    }
    if (R.pivot != 0 && !is_n(R.L->nfw, R.pivot - 1)) {
        next_pivot = R.l_seq;
        zigzag(R, next_pivot, true);
    } else {
        next_pivot = R.pivot + right_smem(R);
    }
    set_pivot(R, next_pivot);
}

// Learned_getSMEMsOnePosOneThread (:1897-2126)
__device__ void one_pos(RState& R) {
    int next_pivot;
    if (is_n(R.L->nfw, R.pivot)) {
        if (R.l_seq - R.pivot < R.min_seed_len) set_pivot(R, R.l_seq);
        else set_pivot(R, R.pivot + 1);
        return;
    }
    if (R.pivot != 0 && !is_n(R.L->nfw, R.pivot - 1)) {
        next_pivot = R.pivot + mem_only(R, true);
        zigzag(R, next_pivot, false);
    } else {
        next_pivot = R.pivot + right_smem(R);
    }
    set_pivot(R, next_pivot);
}

// Learned_getSMEMsAllPosOneThread (:913-972)
__device__ void all_pos(RState& R, int split_len, int split_width, bool round2) {
    set_pivot(R, 0);
    int guard = 0;
    while (R.pivot < R.l_seq) {
        if (++guard > 4 * R.l_seq + 16) break;
        int before = R.n_smems;
        R.sm_base = before;
        R.rec = true;
        step1(R);
        R.rec = false;
        int after = R.n_smems;
        if (!round2) continue;
        if (R.lds_ovf) return;                             // re-run in the next tier (bigger LDS ring)
        for (int k = before; k < after; ++k) {
            int next_pivot = R.pivot;
            int saved = R.min_intv;
            int qbeg = R.sm_start[k - before], qend = R.sm_end[k - before], cnt = R.sm_cnt[k - before];
            if (qend - qbeg < split_len || cnt > split_width) { set_pivot(R, next_pivot); continue; }
            set_pivot(R, (qbeg + qend) >> 1);
            R.min_intv = cnt + 1;
            one_pos(R);
            R.min_intv = saved;
            set_pivot(R, next_pivot);
        }
    }
}

// Learned_bwtSeedStrategyAllPosOneThread (:974-1283) / _mem_tradeoff (:1284-1466)
__device__ void seed_strategy(RState& R) {
    const int min_intv = R.min_intv, msl = R.min_seed_len;
    const i64 n = R.I->n;
    set_pivot(R, 0);
    while (R.pivot < R.l_seq - msl + 1) {
        if (is_n(R.L->nfw, R.pivot)) { set_pivot(R, R.pivot + 1); continue; }
        int amb = first_n(R, R.L->nfw, R.pivot);
        int valid = amb - R.pivot;
        if (valid < msl) { set_pivot(R, R.pivot + valid); continue; }
        // maxLCP first; the interval only if the match is long enough (:1204-1208)
        Located loc = locate_and_first_level(R, R.L->fw, R.pivot, valid, false);
        if (loc.L < msl) { set_pivot(R, R.pivot + msl); continue; }
        int L = loc.L;
        // first level interval around the located slot
        i64 s = loc.s, e = loc.e;
        int nb_lo = L, nb_hi = L;                          // "unknown, may extend"
        i64 last_s = s, last_cnt = 0, cnt, emit_s;
        int match_len;
        for (;;) {
            if (nb_lo >= L && s > 0) edge_down(R, R.L->fw, R.pivot, L, s, s, nb_lo);
            else if (s == 0) nb_lo = 0;
            if (nb_hi >= L && e < n - 1) edge_up(R, R.L->fw, R.pivot, L, e, e, nb_hi);
            else if (e == n - 1) nb_hi = 0;
            cnt = e - s + 1;
            if (cnt >= min_intv) {                         // :1243-1251
                cnt = last_cnt ? last_cnt : cnt;
                emit_s = last_s;
                match_len = L + 1;
                break;
            }
            int nxt = nb_lo > nb_hi ? nb_lo : nb_hi;
            if (nxt < msl) { match_len = msl; emit_s = s; break; }   // :1252-1258
            last_cnt = cnt;
            last_s = s;
            L = nxt;
        }
        if (cnt < min_intv) {                              // :1265-1277
            if (match_len < msl) match_len = msl;
            emit(R, R.pivot, R.pivot + match_len, emit_s, cnt);
        }
        set_pivot(R, R.pivot + match_len);
    }
}

// ---- the search kernel -------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) k_seed(SeedArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    GroupLds* lds = reinterpret_cast<GroupLds*>(smem_raw);
    int* sm_lists = reinterpret_cast<int*>(smem_raw + sizeof(GroupLds) * GROUPS_PER_BLOCK);
    const int lane = threadIdx.x & 63;
    const int gib = threadIdx.x / G;
    RState R;
    R.I = &A.I;
    R.L = &lds[gib];
    R.sm_start = sm_lists + (size_t)gib * 3 * A.lcap;
    R.sm_end = R.sm_start + A.lcap;
    R.sm_cnt = R.sm_end + A.lcap;
    R.cap = A.cap;
    R.lcap = A.lcap;
    R.t = threadIdx.x & (G - 1);
    R.gbase = lane & ~(G - 1);
    R.hits_per_smem = A.opt.hits_per_smem;
    for (;;) {
        unsigned long long ticket = 0;
        if (R.t == 0) ticket = atomicAdd(&A.counters[0], 1ull);
        ticket = __shfl(ticket, R.gbase);
        if (ticket >= (unsigned long long)A.nreads) break;
        const i64 rid = A.pending ? A.pending[ticket] : (i64)ticket;
        const i64 ro = A.read_off[rid];
        const int len = (int)(A.read_off[rid + 1] - ro);
        R.slots = A.slots + (i64)ticket * A.cap;
        R.l_seq = len;
        R.rec = false;
        R.lds_ovf = false;
        R.sm_base = 0;
        R.n_smems = 0;
        R.n_hits = 0;
        R.searches = 0;
        if (len <= 0 || len > MAX_READ_LEN) {
            // the reference exits on reads longer than LEARNED_MAX_READ_LEN (src/bwamem.cpp:1259-1262);
            // here the read yields no seeds and is flagged through slot_cnt = -1
            if (R.t == 0) { A.slot_cnt[rid] = len > MAX_READ_LEN ? -1 : 0; A.slot_hits[rid] = 0; A.slot_loc[rid] = 0; }
            continue;
        }
        // ---- stage the read: both strands, 2 bits/base, first base in the top bits; N packed as A
        //      (src/bwamem.cpp:1277-1344) ----------------------------------------------------------
        const int nw = (len + 31) >> 5;
        bool any_n = false;
        for (int k = R.t; k < MAXW + 1; k += G) {
            u64 f = 0, r = 0;
            if (k < nw) {
                for (int j = 0; j < 32; ++j) {
                    int i = 32 * k + j;
                    u64 cf = 0, cr = 0;
                    if (i < len) {
                        uint8_t b = A.reads[ro + i];
                        uint8_t rb = A.reads[ro + len - 1 - i];
                        cf = b < 4 ? b : 0;
                        cr = rb < 4 ? 3 - rb : 0;
                    }
                    f = (f << 2) | cf;
                    r = (r << 2) | cr;
                }
            }
            R.L->fw[k] = f;
            R.L->rc[k] = r;
        }
        for (int k = R.t; k < MASKW; k += G) {
            u64 mf = 0, mr = 0;
            for (int j = 0; j < 64; ++j) {
                int i = 64 * k + j;
                if (i < len) {
                    if (A.reads[ro + i] >= 4) mf |= 1ull << j;
                    if (A.reads[ro + len - 1 - i] >= 4) mr |= 1ull << j;
                }
            }
            R.L->nfw[k] = mf;
            R.L->nrc[k] = mr;
            any_n |= (mf != 0);
        }
        R.has_n = gballot(any_n, R.gbase) != 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- rounds (src/bwamem.cpp:1358-1394) ------------------------------------------------------
        R.min_seed_len = A.opt.min_seed_len;
        R.min_intv = 1;
        all_pos(R, A.opt.split_len, A.opt.split_width, A.opt.rounds >= 2);
        if (A.opt.rounds >= 3 && A.opt.max_mem_intv > 0) {
            R.min_intv = A.opt.max_mem_intv;
            R.min_seed_len = A.opt.min_seed_len + 1;
            seed_strategy(R);
        }
        if (R.t == 0) {
            const bool ovf = R.n_smems > R.cap || R.lds_ovf;
            A.slot_cnt[rid] = ovf ? 0 : R.n_smems;
            A.slot_hits[rid] = ovf ? 0 : R.n_hits;
            A.slot_loc[rid] = ((i64)A.tier << 40) | (i64)ticket;
            if (ovf) A.ovf_list[atomicAdd(&A.counters[2], 1ull)] = rid;
            else atomicAdd(&A.counters[1], (unsigned long long)R.searches);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- offsets: two-level exclusive scan over reads -------------------------------------------------------
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ __forceinline__ i64 block_scan_excl(i64 v, i64* total) {
    __shared__ i64 wsum[SCAN_BLOCK / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    i64 x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        i64 y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    i64 off = 0, tot = 0;
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        if (w < wid) off += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return off + x - v;
}

// pass 1: per-tile sums of (smem count, hit count)
__global__ void __launch_bounds__(SCAN_BLOCK) k_tile_sums(const int* __restrict__ cnt, const i64* __restrict__ hits, i64 n,
                                                          i64* __restrict__ tile_sums) {
    i64 base = (i64)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    i64 a = 0, b = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        i64 i = base + k;
        if (i < n) {
            int c = cnt[i];
            a += c < 0 ? 0 : c;
            b += hits[i];
        }
    }
    i64 ta, tb;
    block_scan_excl(a, &ta);
    block_scan_excl(b, &tb);
    if (threadIdx.x == 0) { tile_sums[2 * blockIdx.x] = ta; tile_sums[2 * blockIdx.x + 1] = tb; }
}

// pass 2: one block scans the tile sums in place (exclusive); totals to out[0..1]
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tiles(i64* __restrict__ tile_sums, i64 ntiles, i64* __restrict__ totals) {
    i64 ca = 0, cb = 0;
    for (i64 base = 0; base < ntiles; base += SCAN_BLOCK) {
        i64 i = base + threadIdx.x;
        i64 a = i < ntiles ? tile_sums[2 * i] : 0, b = i < ntiles ? tile_sums[2 * i + 1] : 0;
        i64 ta, tb;
        i64 ea = block_scan_excl(a, &ta);
        i64 eb = block_scan_excl(b, &tb);
        if (i < ntiles) { tile_sums[2 * i] = ca + ea; tile_sums[2 * i + 1] = cb + eb; }
        ca += ta;
        cb += tb;
    }
    if (threadIdx.x == 0) { totals[0] = ca; totals[1] = cb; }
}

// pass 3: per-read offsets
__global__ void __launch_bounds__(SCAN_BLOCK) k_offsets(const int* __restrict__ cnt, const i64* __restrict__ hits, i64 n,
                                                        const i64* __restrict__ tile_sums, i64* __restrict__ smem_off,
                                                        i64* __restrict__ hit_off) {
    i64 base = (i64)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    i64 va[SCAN_ITEMS], vb[SCAN_ITEMS], a = 0, b = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        i64 i = base + k;
        va[k] = vb[k] = 0;
        if (i < n) {
            int c = cnt[i];
            va[k] = c < 0 ? 0 : c;
            vb[k] = hits[i];
        }
        a += va[k];
        b += vb[k];
    }
    i64 ta, tb;
    i64 ea = block_scan_excl(a, &ta) + tile_sums[2 * blockIdx.x];
    i64 eb = block_scan_excl(b, &tb) + tile_sums[2 * blockIdx.x + 1];
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        i64 i = base + k;
        if (i < n) { smem_off[i] = ea; hit_off[i] = eb; }
        ea += va[k];
        eb += vb[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_BLOCK - 1) { smem_off[n] = ea; hit_off[n] = eb; }
}

// ---- compaction + hit gather: one 16-lane group per read --------------------------------------------------
struct TierTable {
    const SlotRec* base[N_TIERS];
    int cap[N_TIERS];
};

__global__ void __launch_bounds__(BLOCK) k_gather(const SaEnt* __restrict__ sa, TierTable tiers, const i64* __restrict__ slot_loc,
                                                   const int* __restrict__ cnt, i64 n, int hits_per_smem,
                                                   const i64* __restrict__ smem_off, const i64* __restrict__ hit_off,
                                                   meme_mem_tl* __restrict__ smems, u64* __restrict__ hits) {
    const int t = threadIdx.x & (G - 1);
    i64 gid = ((i64)blockIdx.x * BLOCK + threadIdx.x) / G;
    const i64 ngroups = (i64)gridDim.x * BLOCK / G;
    for (i64 r = gid; r < n; r += ngroups) {
        int c = cnt[r];
        if (c <= 0) continue;
        const i64 loc = slot_loc[r];
        const int tier = (int)(loc >> 40);
        const SlotRec* sl = tiers.base[tier] + (loc & ((1ll << 40) - 1)) * tiers.cap[tier];
        meme_mem_tl* out = smems + smem_off[r];
        u64* hout = hits + hit_off[r];
        i64 hb = 0;
        for (int k = 0; k < c; ++k) {
            SlotRec s = sl[k];
            i64 h = s.count;
            if (hits_per_smem > 0 && h > hits_per_smem) h = hits_per_smem;
            if (t == 0) {
                meme_mem_tl m;
                m.start = s.start;
                m.end = s.end;
                m.hitbeg = (int32_t)hb;
                m.hitcount = s.count > (i64)INT_MAX ? INT_MAX : (int32_t)s.count;
                m.cache_refpos = sa[s.sa_start].pos;
                out[k] = m;
            }
            for (i64 i = t; i < h; i += G) hout[hb + i] = sa[s.sa_start + i].pos;
            hb += h;
        }
    }
}

int launch_seed(meme_ctx* ctx, const uint8_t* d_reads, const i64* d_read_off, i64 nreads, const meme_seed_opt* opt,
                meme_seed_result* out) {
    unsigned long long h_counters[4];
    int rc;
    if ((rc = meme_buf_reserve(ctx, ctx->slot_cnt, (size_t)nreads * sizeof(int)))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->slot_hits, (size_t)nreads * sizeof(i64)))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->slot_loc, (size_t)nreads * sizeof(i64)))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->counters, 4 * sizeof(unsigned long long)))) return rc;
    int dev_cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) dev_cus = prop.multiProcessorCount;
    float ms_total = 0.f;
    i64 launches = 0, searches = 0;
    TierTable tiers;
    for (int t = 0; t < N_TIERS; ++t) { tiers.base[t] = nullptr; tiers.cap[t] = 0; }
    i64 n_todo = nreads;
    const i64* pending = nullptr;
    for (int tier = 0;; ++tier) {
        const int cap = tier == 0 ? (int)ctx->smem_cap : TIER_CAP[tier];
        const int lcap = cap < LDS_RING_MAX ? cap : LDS_RING_MAX;
        DevBuf& sb = ctx->slots[tier];
        DevBuf& ob = ctx->ovf[tier & 1];
        if ((rc = meme_buf_reserve(ctx, sb, (size_t)n_todo * cap * sizeof(SlotRec)))) return rc;
        if ((rc = meme_buf_reserve(ctx, ob, (size_t)n_todo * sizeof(i64)))) return rc;
        HIP_TRY(hipMemsetAsync(ctx->counters.p, 0, 4 * sizeof(unsigned long long), ctx->stream));
        SeedArgs A;
        A.I = ctx->idx;
        A.reads = d_reads;
        A.read_off = d_read_off;
        A.nreads = n_todo;
        A.opt = *opt;
        A.slots = (SlotRec*)sb.p;
        A.slot_cnt = (int*)ctx->slot_cnt.p;
        A.slot_hits = (i64*)ctx->slot_hits.p;
        A.slot_loc = (i64*)ctx->slot_loc.p;
        A.pending = pending;
        A.ovf_list = (i64*)ob.p;
        A.cap = cap;
        A.lcap = lcap;
        A.tier = tier;
        A.counters = (unsigned long long*)ctx->counters.p;
        tiers.base[tier] = (const SlotRec*)sb.p;
        tiers.cap[tier] = cap;
        size_t lds = sizeof(GroupLds) * GROUPS_PER_BLOCK + (size_t)GROUPS_PER_BLOCK * 3 * lcap * sizeof(int);
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void*)k_seed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        i64 want = (n_todo + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK;
        i64 blocks = ctx->seed_blocks > 0 ? ctx->seed_blocks : (i64)dev_cus * 4;
        if (blocks > want) blocks = want;
        if (blocks < 1) blocks = 1;
        HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
        hipLaunchKernelGGL(k_seed, dim3((unsigned)blocks), dim3(BLOCK), lds, ctx->stream, A);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
        HIP_TRY(hipMemcpyAsync(h_counters, ctx->counters.p, sizeof(h_counters), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        ms_total += ms;
        ++launches;
        searches += (i64)h_counters[1];
        if (h_counters[2] == 0) break;
        // some reads produced more SMEMs than their slots hold (pathological repeats): re-run only those
        if (tier + 1 >= N_TIERS) {
            meme_set_error("a read produced more than %d SMEMs", TIER_CAP[N_TIERS - 1]);
            return MEME_E_CAPACITY;
        }
        n_todo = (i64)h_counters[2];
        pending = (const i64*)ob.p;
    }
    h_counters[1] = (unsigned long long)searches;
    ctx->tm.seed_kernel_ms = ms_total;
    ctx->tm.seed_launches = launches;
    // offsets
    i64 ntiles = (nreads + SCAN_TILE - 1) / SCAN_TILE;
    if ((rc = meme_buf_reserve(ctx, ctx->scan_tmp, (size_t)(2 * ntiles + 2) * sizeof(i64)))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->smem_off, (size_t)(nreads + 1) * sizeof(i64)))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->hit_off, (size_t)(nreads + 1) * sizeof(i64)))) return rc;
    i64* tiles = (i64*)ctx->scan_tmp.p;
    i64* totals = tiles + 2 * ntiles;
    HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    hipLaunchKernelGGL(k_tile_sums, dim3((unsigned)ntiles), dim3(SCAN_BLOCK), 0, ctx->stream, (const int*)ctx->slot_cnt.p,
                       (const i64*)ctx->slot_hits.p, nreads, tiles);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(SCAN_BLOCK), 0, ctx->stream, tiles, ntiles, totals);
    hipLaunchKernelGGL(k_offsets, dim3((unsigned)ntiles), dim3(SCAN_BLOCK), 0, ctx->stream, (const int*)ctx->slot_cnt.p,
                       (const i64*)ctx->slot_hits.p, nreads, (const i64*)tiles, (i64*)ctx->smem_off.p,
                       (i64*)ctx->hit_off.p);
    HIP_TRY(hipGetLastError());
    i64 h_tot[2];
    HIP_TRY(hipMemcpyAsync(h_tot, totals, sizeof(h_tot), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if ((rc = meme_buf_reserve(ctx, ctx->smems, (size_t)(h_tot[0] + 1) * sizeof(meme_mem_tl)))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->hits, (size_t)(h_tot[1] + 1) * sizeof(u64)))) return rc;
    i64 gblocks = (nreads + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK;
    if (gblocks > (i64)dev_cus * 8) gblocks = (i64)dev_cus * 8;
    if (gblocks < 1) gblocks = 1;
    hipLaunchKernelGGL(k_gather, dim3((unsigned)gblocks), dim3(BLOCK), 0, ctx->stream, ctx->idx.sa, tiers,
                       (const i64*)ctx->slot_loc.p, (const int*)ctx->slot_cnt.p, nreads, opt->hits_per_smem,
                       (const i64*)ctx->smem_off.p, (const i64*)ctx->hit_off.p, (meme_mem_tl*)ctx->smems.p,
                       (u64*)ctx->hits.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float gms = 0.f;
    HIP_TRY(hipEventElapsedTime(&gms, ctx->ev[2], ctx->ev[3]));
    ctx->tm.seed_gather_ms = gms;
    out->d_smems = (const meme_mem_tl*)ctx->smems.p;
    out->d_smem_off = (const i64*)ctx->smem_off.p;
    out->d_hits = (const u64*)ctx->hits.p;
    out->d_hit_off = (const i64*)ctx->hit_off.p;
    out->total_smems = h_tot[0];
    out->total_hits = h_tot[1];
    out->searches = (i64)h_counters[1];
    return MEME_OK;
}

int check_opt(const meme_seed_opt* opt) {
    if (!opt || opt->min_seed_len < 1 || opt->rounds < 1 || opt->rounds > 3 || opt->hits_per_smem < 0) {
        meme_set_error("bad meme_seed_opt");
        return MEME_E_ARG;
    }
    return MEME_OK;
}

}  // namespace

extern "C" int meme_seed_batch_device(meme_ctx* ctx, const uint8_t* d_reads, const int64_t* d_read_off, int64_t nreads,
                                      int64_t total_bases, const meme_seed_opt* opt, meme_seed_result* out) {
    (void)total_bases;
    if (!ctx || !d_reads || !d_read_off || !out || nreads < 0) return MEME_E_ARG;
    int rc = check_opt(opt);
    if (rc) return rc;
    if (!ctx->idx.sa) { meme_set_error("meme_seed_batch: no index loaded"); return MEME_E_STATE; }
    HIP_TRY(hipSetDevice(ctx->device));
    if (nreads == 0) { memset(out, 0, sizeof(*out)); return MEME_OK; }
    return launch_seed(ctx, d_reads, (const i64*)d_read_off, nreads, opt, out);
}

extern "C" int meme_seed_batch(meme_ctx* ctx, const uint8_t* reads, const int64_t* read_off, int64_t nreads,
                               const meme_seed_opt* opt, meme_mem_tl* smems, int64_t smem_capacity, int64_t* smem_off,
                               uint64_t* hits, int64_t hit_capacity, int64_t* hit_off, int64_t* total_smems,
                               int64_t* total_hits) {
    if (!ctx || !reads || !read_off || !smems || !smem_off || !hits || !hit_off || nreads < 0) return MEME_E_ARG;
    int rc = check_opt(opt);
    if (rc) return rc;
    if (!ctx->idx.sa) { meme_set_error("meme_seed_batch: no index loaded"); return MEME_E_STATE; }
    HIP_TRY(hipSetDevice(ctx->device));
    if (nreads == 0) { smem_off[0] = 0; hit_off[0] = 0; if (total_smems) *total_smems = 0; if (total_hits) *total_hits = 0; return MEME_OK; }
    const i64 bases = read_off[nreads] - read_off[0];
    if (read_off[0] != 0) { meme_set_error("read_off[0] must be 0"); return MEME_E_ARG; }
    if ((rc = meme_buf_reserve(ctx, ctx->reads, (size_t)bases + 16))) return rc;
    if ((rc = meme_buf_reserve(ctx, ctx->read_off, (size_t)(nreads + 1) * sizeof(i64)))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->reads.p, reads, (size_t)bases, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->read_off.p, read_off, (size_t)(nreads + 1) * sizeof(i64), hipMemcpyHostToDevice, ctx->stream));
    meme_seed_result res;
    rc = launch_seed(ctx, (const uint8_t*)ctx->reads.p, (const i64*)ctx->read_off.p, nreads, opt, &res);
    if (rc) return rc;
    if (total_smems) *total_smems = res.total_smems;
    if (total_hits) *total_hits = res.total_hits;
    HIP_TRY(hipMemcpyAsync(smem_off, res.d_smem_off, (size_t)(nreads + 1) * sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(hit_off, res.d_hit_off, (size_t)(nreads + 1) * sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    if (res.total_smems > smem_capacity || res.total_hits > hit_capacity) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        meme_set_error("output buffers too small: need %lld SMEMs and %lld hits", (long long)res.total_smems,
                       (long long)res.total_hits);
        return MEME_E_CAPACITY;
    }
    HIP_TRY(hipMemcpyAsync(smems, res.d_smems, (size_t)res.total_smems * sizeof(meme_mem_tl), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(hits, res.d_hits, (size_t)res.total_hits * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MEME_OK;
}
