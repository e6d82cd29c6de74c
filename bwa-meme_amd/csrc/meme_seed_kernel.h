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
//  * A read is owned by a group of G lanes (G = 1..32, default 4: 16 reads per wavefront).  The pivot logic is
//    group-uniform scalar state replicated in the group's lanes; only suffix-array windows are lane-parallel.
//  * ONE flat loop per wavefront: every read is a small machine with a program counter over the pivot logic of
//    rounds 1-3; the loop body is  control -> window -> resolve -> level walk -> apply.  A read that finishes pulls
//    the next one inside the same loop, so reads never wait for each other, and the heavy code (window load +
//    compare + reference-word loop) exists exactly once in the kernel.
//  * The unit of memory traffic is a *window*: E*G consecutive 16-byte suffix-array entries (12 slots = 192 B at G=4),
//    E coalesced loads per lane issued together.  Each lane compares its entries' 64-bit keys (and, only when all 32
//    bases agree, 2-bit reference words) with the read; ballots yield the partition point, the longest common prefix
//    and -- from the same data -- the SMEM hit interval, so the reference's chain of ~log2(err)+linear dependent
//    single-entry probes collapses to one window in the common case.
//  * Partition point, lower and upper interval edge are all "where does a monotone predicate flip?": a window that
//    does not contain the flip moves a bracket [lo, hi] and the next window gallops or bisects -- relocations and
//    edge scans are just more trips through the same loop.
//  * The learned model is a hint: model error bounds are never needed, so any parameter file the
//    reference loads is accepted.
//  * Reads are packed once per batch by k_pack_reads (2 bits/base, both strands, first base in the top bits of each
//    u64, plus N masks) and staged in LDS when pulled; a 32-base query word at any offset is a funnel shift of two
//    LDS words, replacing the reference's 8 pre-shifted copies of every read.
#pragma once
#include <limits.h>

#include "meme_common.h"

namespace seedk {

constexpr int BLOCK = 256;
constexpr int MAX_READ_LEN = 500;        // LEARNED_MAX_READ_LEN (reference src/bwamem.cpp:1259)

struct SlotRec {          // search-kernel output, one per SMEM
    int32_t start, end;
    i64 sa_start;         // first suffix-array slot of the interval, or SLOT_POS | text position of the only hit (+ SLOT_DEFER)
    i64 count;
};
// sa_start flags.  An SMEM with ONE occurrence whose text position the search already holds (it sits in the window that settled the
// search) carries the position itself: the gather kernel need not fetch it again, and the position is what the re-seeding verifier needs.
constexpr i64 SLOT_POS = 1ll << 62;
constexpr i64 SLOT_DEFER = 1ll << 61;     // the SMEM's re-seeding region (round 2) was not searched here: k_reseed walks it on the plcp table
constexpr i64 SLOT_PEND = 1ll << 59;      // (between k_reseed and k_reseed_emit) an SMEM whose suffix-array interval is still to be looked up
constexpr i64 SLOT_VAL = (1ll << 48) - 1;
constexpr int DEFER_MAX_K = 64;           // only the first 64 SMEMs of a read can be deferred (one bit each in RedoRec::mask)


constexpr int N_TIERS = 3;
constexpr int TIER_CAP[N_TIERS] = {64, 2048, 65536};   // global SMEM slots per read; tier 0 is tunable (meme_ctx::smem_cap, default 128)
#ifndef LCAP0
#define LCAP0 16
#endif
constexpr int TIER_LCAP[N_TIERS] = {LCAP0, 512, 512};     // LDS ring: SMEMs of one first-round pass (<= read length)

struct PackGeom {
    int W;        // u64 words per strand (>= ceil(maxlen/32) + 2)
    int MW;       // u64 N-mask words per strand
    int stride;   // 2*W + 2*MW + 1 (last word = read length | has-N flag << 31)
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
    unsigned long long* counters;   // [0] ticket, [1] searches, [2] overflowed reads, [3] window loads, [4..11] SEED_PROF, [12] lane searches of k_reseed
    int defer;                      // 1: re-seeding regions of unique SMEMs are left to k_reseed (tier 0 only)
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
    __shared__ int has_n[64];                                   // per read of the group (PACK_RB <= 64)
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
        const int nw = g.stride - 1;                            // data words per read; the length word follows them
        // does the read contain an ambiguous base?  (dword-wide scan of the staged bytes; almost always "no", and then
        // its N-mask words are zero without looking at the bases again)
        for (int rr = threadIdx.x; rr < nr; rr += blockDim.x) {
            const i64 ro = read_off[r0 + rr];
            const int len = (int)(read_off[r0 + rr + 1] - ro);
            uint32_t acc = 0;
            if (len > 0 && len <= MAX_READ_LEN) {
                const int o = shift + (int)(ro - b0), last = o + len - 1;
                const int d0 = o >> 2, d1 = last >> 2;
                for (int d = d0; d <= d1; ++d) {
                    uint32_t m = 0xFCFCFCFCu;
                    if (d == d0) m &= 0xFFFFFFFFu << (8 * (o & 3));
                    if (d == d1) m &= 0xFFFFFFFFu >> (8 * (3 - (last & 3)));
                    acc |= stage[d] & m;
                }
            }
            has_n[rr] = acc != 0;
        }
        __syncthreads();
        for (int wk = threadIdx.x; wk < nr * nw; wk += blockDim.x) {
            const int rr = wk / nw, k = wk - rr * nw;
            const i64 r = r0 + rr;
            const i64 ro = read_off[r];
            int len = (int)(read_off[r + 1] - ro);
            u64 v = 0;
            if (len > MAX_READ_LEN) len = 0;
            const int po = shift + (int)(ro - b0);              // the read's first byte in the staged area
            const uint8_t* p = sb + po;
            if (k < 2 * g.W) {
                const bool rc = k >= g.W;
                const int w = rc ? k - g.W : k;
                const int nvalid = len - 32 * w;                // bases of this word
                if (nvalid <= 0) v = 0;
                else if (!has_n[rr]) {
                    // Every byte of the read is 0..3: 32 bases = 8 staged dwords (re-aligned with v_alignbyte), four bases
                    // of a dword gathered into one byte by a multiplication, no per-base loop.  Forward: bases 32w..32w+31;
                    // reverse complement: the same for the 32 bases that END at len-1-32w, bytes and dwords in reverse
                    // order, complemented.  Bytes beyond the read (the neighbours' bases) are shifted / masked away.
                    const int s0 = rc ? (nvalid >= 32 ? nvalid - 32 : 0) : 32 * w;
                    const int o = po + s0, dwi = o >> 2;
                    const unsigned sh = (unsigned)(o & 3);
                    uint32_t a[9];
#pragma unroll
                    for (int q = 0; q < 9; ++q) a[q] = stage[dwi + q];
                    if (!rc) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const uint32_t x = __builtin_amdgcn_alignbyte(a[q + 1], a[q], sh) & 0x03030303u;
                            v = (v << 8) | ((x * 0x40100401u) >> 24);          // first base of the dword in the top two bits
                        }
                        if (nvalid < 32) v &= ~0ull << (2 * (32 - nvalid));
                    } else {
#pragma unroll
                        for (int q = 7; q >= 0; --q) {
                            const uint32_t x = (__builtin_amdgcn_alignbyte(a[q + 1], a[q], sh) & 0x03030303u) ^ 0x03030303u;
                            v = (v << 8) | ((x * 0x01041040u) >> 24);          // last base of the dword in the top two bits
                        }
                        if (nvalid < 32) v <<= 2 * (32 - nvalid);
                    }
                } else {
#pragma unroll 16
                    for (int j = 0; j < 32; ++j) {
                        const int i = 32 * w + j;
                        u64 c = 0;
                        if (i < len) {
                            const uint8_t bb = rc ? p[len - 1 - i] : p[i];
                            c = bb < 4 ? (rc ? 3 - bb : bb) : 0;   // N packed as A (src/bwamem.cpp:1293-1294)
                        }
                        v = (v << 2) | c;
                    }
                }
            } else if (has_n[rr]) {
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
            out[r * g.stride + k] = v;
        }
        __syncthreads();
        for (int rr = threadIdx.x; rr < nr; rr += blockDim.x) {  // length word: length | (read has an N) << 31
            const i64 r = r0 + rr;
            const int len = (int)(read_off[r + 1] - read_off[r]);
            out[r * g.stride + nw] = (u64)(unsigned)len | (has_n[rr] ? (1ull << 31) : 0ull);
        }
    }
}

// ---- per-read state ------------------------------------------------------------------------------------
// Every read is a little machine with a "program counter" over the reference's pivot logic (rounds 1-3).  The
// wavefront runs ONE loop whose body is: [control: advance reads that need a new search request] -> [window: load
// E*G suffix-array slots and compare them with the query] -> [resolve: what the window means in the read's current
// phase].  A read that finishes pulls the next one inside the same loop, so the 64/G reads of a wavefront never
// wait for each other and the only heavy code (window load + compare) exists once in the kernel.
enum Pc : int {
    PC_FETCH, PC_ALLPOS_TOP, PC_ZZ_TOP, PC_ZZ_RIGHT, PC_ZZ_END, PC_AFTER_STEP1, PC_R2_LOOP, PC_R2_AFTER, PC_R3_INIT,
    PC_R3_TOP, PC_DONE, PC_EXIT
};
enum Kind : int { K_S1_RIGHT, K_ZZ_LEFT, K_ZZ_RIGHT, K_OP_MEM, K_OP_SMEM, K_R3 };
// what the next window is for: the partition point of the query (first window at the model's prediction, later ones
// gallop/bisect), or the lower / upper end of the run of suffixes sharing >= L bases with it
enum Phase : int { PH_CTRL, PH_PART, PH_EDGE_DN, PH_EDGE_UP };

// explicit address spaces: LDS (3) for the staged read, global (1) for the index.  Generic pointers would
// compile to FLAT loads whose waits serialise LDS and HBM traffic.
#ifndef CMP_WORDS
#define CMP_WORDS 2          // reference words fetched per round trip once the 32-base key matched
#endif
typedef const __attribute__((address_space(3))) u64* lds_u64;
typedef __attribute__((address_space(3))) int* lds_int;
typedef __attribute__((address_space(3))) unsigned short* lds_u16;
typedef const __attribute__((address_space(1))) u64* glb_u64;
typedef const __attribute__((address_space(1))) SaEnt* glb_ent;
typedef const __attribute__((address_space(1))) Rmi32* glb_rmi;

// Where a search reads its query words from.  1 (default): the read's packed record is staged in LDS when the read is
// pulled.  0: straight from the packed-read array in global memory (the 168 B record stays L2-resident for the ~60
// searches of its lifetime); the per-read LDS footprint shrinks to the state words so twice as many reads fit a CU,
// which lifts 2 lanes/read from 81 to 113 M reads/s (512 Mbp probe) but costs 4 lanes/read 5 % (118 -> 112): one more
// dependent L2 round trip per search.  Measured, kept as a build option.
#ifndef SEED_QUERY_LDS
#define SEED_QUERY_LDS 1
#endif
// 1: the model's error bounds place the first window (asymmetric bounds -> asymmetric window); 0: centred on the prediction
#ifndef SEED_USE_ERR
#define SEED_USE_ERR 1
#endif
// 1: first windows start on a 64-byte boundary of the entry array (exactly two 128-byte lines per 12-entry window)
#ifndef SEED_ALIGN_WIN
#define SEED_ALIGN_WIN 0
#endif
#if SEED_QUERY_LDS
typedef lds_u64 q_u64;
#else
typedef const __attribute__((address_space(1))) u64* q_u64;
#endif

__device__ __forceinline__ u64 ext_l(q_u64 w, int s) {
    int k = s >> 5, sh = (s & 31) * 2;
    u64 a = w[k], b = w[k + 1];
    return sh ? (a << sh) | (b >> (64 - sh)) : a;
}

// The re-seeding walks read the plcp table around a region's middle: two windows of PLCP_WIN bytes (the locus, its mirror on the other
// strand), 16-byte aligned so that they can be fetched with aligned 16-byte loads.  The zig-zag moves at most min_seed_len - 1 bases
// per step, so it rarely leaves them (then the byte comes from global memory).
constexpr int PLCP_WIN = 64;
typedef const __attribute__((address_space(3))) uint8_t* lds_u8;
__device__ __forceinline__ i64 plcp_win_fwd(i64 T0, int mid) { const i64 a = T0 + mid - 24; return (a < 0 ? 0 : a) & ~15ll; }
__device__ __forceinline__ i64 plcp_win_rev(i64 T0, int mid, i64 n) { const i64 a = n - 1 - (T0 + mid + 24); return (a < 0 ? 0 : a) & ~15ll; }

__device__ __forceinline__ u64 lowmask(int k) { return k >= 64 ? ~0ull : (k <= 0 ? 0ull : ((1ull << k) - 1ull)); }

// cold per-read state kept in LDS (group-uniform redundant stores; every lane reads back what it wrote)
enum StIdx : int { ST_BEFORE, ST_AFTER, ST_R2_K, ST_R2_NEXT, ST_R2_SAVED, ST_ZZ_NEXT, ST_ZZ_SP, ST_ZZ_GUARD, ST_AP_GUARD,
                   ST_SM_BASE, ST_N_SMEMS, ST_HITS_LO, ST_HITS_HI, ST_SEARCHES, ST_FLAGS, ST_LAST_CNT_LO, ST_LAST_CNT_HI,
                   ST_LAST_S_LO, ST_LAST_S_HI, ST_WINDOWS,
                   // the level walk of the search in flight (do not survive a search)
                   ST_L, ST_SE_LO, ST_SE_HI, ST_EE_LO, ST_EE_HI, ST_NB_LO, ST_NB_HI, ST_LF, ST_CB_LO, ST_CB_HI,
                   ST_TICKET_LO, ST_TICKET_HI, ST_WORDS };
// per-read words are cleared when a read is staged; the group's running totals live in registers
enum StFlag : int { F_ZZ_CHECK = 1, F_ZZ_RET_ONEPOS = 2, F_REC = 4, F_LDS_OVF = 8 };
enum LevelFlag : int { LF_NEED_LO = 1, LF_NEED_HI = 2, LF_HAVE_LAST = 4 };

#ifndef WIN_E
#define WIN_E 3      // suffix-array entries per lane in a window: window = WIN_E * G slots (at most 64)
#endif
#ifndef WIN_SLOTS
#define WIN_SLOTS 12  // ... but at least this many slots
#endif
__host__ __device__ constexpr int win_entries(int G) { return (WIN_E * G > 64) ? 64 / G : (WIN_E * G < WIN_SLOTS ? WIN_SLOTS / G : WIN_E); }

// LDS bytes of one workgroup: per group the packed read, the SMEM ring of one first-round pass (2 ints per entry),
// the cold state words and two windows of 16-bit LCPs (the partition window stays cached while edges are followed)
constexpr int TICKET_CHUNK = 32;     // reads a wavefront draws from the global ticket counter at a time

__host__ __device__ inline size_t seed_lds_group_bytes(int G, int stride, int lcap) {
    const int groups = BLOCK / G;
    const int W = win_entries(G) * G;
    return (((size_t)groups * (SEED_QUERY_LDS ? stride * 8 : 0) + (size_t)groups * (2 * lcap + ST_WORDS + (W + 1) / 2) * sizeof(int)) + 7) & ~(size_t)7;
}
// + per wavefront: the ticket chunk it is handing out (count, base)
inline size_t seed_lds_bytes(int G, const PackGeom& geo, int lcap) {
    return seed_lds_group_bytes(G, geo.stride, lcap) + (size_t)(BLOCK / 64) * 16;
}

// ---- compare: compare_read_and_ref_binary* (:226-601) ---------------------------------------------------
// For E suffix-array entries per lane: L = min(cap, n - pos).  lcp < L: less = ref base < read base.
// lcp == L: less = (L < ref_len) ("exact": the suffix continues past the query and sorts before it; a suffix that
// ends first sorts after it, as if followed by T-padding).
// The 64-bit keys settle most entries; the ones whose 32 key bases all agree continue in the 2-bit text through ONE
// loop instance shared by the lane's E entries (a wavefront usually has one such entry per read).
template <int E>
__device__ __forceinline__ void window_compare(glb_u64 pac, i64 n, q_u64 s, u64 wq, int off, int cap, const u64 (&ek)[E],
                                               const u64 (&ep)[E], int (&lcp)[E], bool (&less)[E]) {
    int l[E], Lc[E];
    bool lt[E];
    unsigned pend = 0;
    const i64 nlimit = n - (i64)cap;                 // suffixes starting beyond it are shorter than cap (rare); may be negative
#pragma unroll
    for (int e = 0; e < E; ++e) {
        Lc[e] = cap;
        if ((i64)ep[e] > nlimit) Lc[e] = (int)(n - (i64)ep[e]);
        const u64 x = ek[e] ^ wq;
        lt[e] = ek[e] < wq;
        l[e] = x ? (__clzll((long long)x) >> 1) : 32;
        if (!x && 32 < Lc[e]) pend |= 1u << e;
    }
    while (pend) {
        const int es = __ffs((int)pend) - 1;
        pend &= pend - 1;
        u64 pos = ep[0];
        int Ls = Lc[0];
#pragma unroll
        for (int e = 1; e < E; ++e)
            if (es == e) { pos = ep[e]; Ls = Lc[e]; }
        // Consecutive text words are adjacent in memory (same sector), so CMP_WORDS are fetched per round trip
        // instead of one dependent load per word.
        const i64 p0 = (i64)pos + 32;
        glb_u64 pw = pac + (p0 >> 5);
        const int sh = (int)(p0 & 31) * 2;
        int ll = 32;
        bool llt = false, done = false;
        for (int k = 1; !done; k += CMP_WORDS, pw += CMP_WORDS) {
            u64 w[CMP_WORDS + 1];
#pragma unroll
            for (int j = 0; j < CMP_WORDS + 1; ++j) w[j] = pw[j];
#pragma unroll
            for (int j = 0; j < CMP_WORDS; ++j) {
                if (done) break;
                const u64 wr = sh ? (w[j] << sh) | (w[j + 1] >> (64 - sh)) : w[j];
                const u64 q = ext_l(s, off + 32 * (k + j));
                const u64 y = wr ^ q;
                if (y) { ll += __clzll((long long)y) >> 1; llt = wr < q; done = true; }
                else { ll += 32; if (ll >= Ls) done = true; }
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (es == e) { l[e] = ll; lt[e] = llt; }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        // Lc == cap < ref_len unless the suffix was clamped above (then Lc == ref_len) or starts exactly at nlimit
        if (l[e] >= Lc[e]) { lcp[e] = Lc[e]; less[e] = (i64)ep[e] < nlimit; }
        else { lcp[e] = l[e]; less[e] = lt[e]; }
    }
}

// ---- learned_index_lookup (:186-210): same arithmetic (FP64 FMA + clamp), used as a hint ------------------
// `err_out`: the record's error word (bits 61..32 = how far the true position can lie below the prediction, bits 30..0
// above), used only to place the first window -- a wrong or loose bound costs a second window, never a wrong answer.
__device__ __forceinline__ i64 rmi_lookup(glb_rmi l2, glb_rmi l1, i64 n_l1, int shift, i64 n, u64 key, u64& err_out) {
    u64 m = shift >= 64 ? 0ull : key >> shift;
    double icpt = l2[m].icpt, slope = l2[m].slope;
    u64 err = l2[m].err;
    double x = (double)key;
    double f = fma(slope, x, icpt);
    if (err >> 63) {
        u64 ps = (err >> 32) & 0x7fffffffull;
        double pn = (double)(err & 0xffffffffull) - 1.0;
        double c = f < 0.0 ? 0.0 : (f > pn ? pn : f);
        u64 j = ps + (u64)c;
        if (j >= (u64)n_l1) j = n_l1 > 0 ? (u64)n_l1 - 1 : 0;   // a malformed table cannot send the load astray
        f = fma(l1[j].slope, x, l1[j].icpt);
        err = l1[j].err;
    }
    err_out = err;
    double top = (double)n - 1.0;
    if (f < 0.0) return 0;
    if (f > top) return n - 1;
    return (i64)f;
}

// OR-reduction over the G lanes of a group without touching LDS: DPP lane permutations inside a 16-lane row
// (quad swaps, then mirrored halves / rows); only groups of 32 need a cross-row exchange.
template <int G>
__device__ __forceinline__ int group_or(int v) {
    if constexpr (G >= 2) v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
    if constexpr (G >= 4) v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v |= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);    // row_half_mirror
    if constexpr (G >= 16) v |= __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
    if constexpr (G >= 32) v |= __shfl_xor(v, 16);
    return v;
}

// values of a per-lane array of window slots (slot j lives in lane j % G, register j / G) at slots ia and ib, packed
// (ia's value << 16) | ib's value; an index outside [0, E*G) yields 0
template <int G, int E>
__device__ __forceinline__ int slot_pair(const int (&v)[E], int t, int ia, int ib) {
    int x = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int j = e * G + t;
        x |= (j == ia) ? (v[e] << 16) : 0;
        x |= (j == ib) ? v[e] : 0;
    }
    return group_or<G>(x);
}

__device__ __forceinline__ bool is_n(q_u64 mask, int i) { return (mask[i >> 6] >> (i & 63)) & 1ull; }

// first ambiguous base at/after `from` (Tokenization's *ambiguous_pos, :795-901)
__device__ __forceinline__ int first_n(q_u64 mask, bool has_n, int from, int l_seq) {
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
// Search semantics (mem_search / right_smem_search and the _tradeoff twins, :2131-4189):
//   maxLCP = longest prefix of the query (<= vlen bases) occurring in the text;
//   mode 0: L = maxLCP.
//   mode 1: L = largest l <= maxLCP whose SA interval holds >= min_intv suffixes; [start,count) = interval
//           (:2365-2574, :2902-2942).
//   mode 2: third round (:1199-1281): walk the levels maxLCP = L0 > L1 > ... until the interval holds
//           >= min_intv suffixes or the next level is shorter than min_seed_len.
// All of it reduces to one question asked of a window of W consecutive slots: where does a predicate that is
// monotone over the suffix array flip from true to false?
//   PH_PART     pred = suffix < query                  flip = partition point; its two neighbours carry maxLCP
//   PH_EDGE_DN  pred = LCP(suffix, query) < L  (below the run)   flip = first slot of the level-L interval
//   PH_EDGE_UP  pred = LCP(suffix, query) >= L (above the run)   flip = one past its last slot
// [lo, hi] brackets the flip (lo: highest slot known true, hi: lowest known false); a window that does not
// contain it moves the bracket and the next window gallops (step doubling) or bisects.
template <int G>
#ifndef SEED_MIN_WAVES
#define SEED_MIN_WAVES 5
#endif
__global__ void __launch_bounds__(BLOCK, (G >= 4 ? SEED_MIN_WAVES : G)) k_seed(SeedArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int GROUPS = BLOCK / G;
    constexpr int E = win_entries(G);                      // entries per lane (the window mask is 64 bits)
    constexpr int W = E * G;                               // window width in slots
    constexpr u64 GFULL = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
    constexpr u64 WFULL = (W == 64) ? ~0ull : ((1ull << W) - 1ull);
    const int lane = threadIdx.x & 63;
    const int gib = threadIdx.x / G;
    const int t = threadIdx.x & (G - 1);
    const int gbase = lane & ~(G - 1);
    const int stride = A.geo.stride, PW = A.geo.W, MW = A.geo.MW;
    const int cap = A.cap, lcap = A.lcap;
#if SEED_QUERY_LDS
    u64* rd = reinterpret_cast<u64*>(smem_raw) + (size_t)gib * stride;
    const q_u64 fw = (lds_u64)rd;
#else
    q_u64 fw = (q_u64)A.packed;                          // the current read's record: fw[PW] rc[PW] nfw[MW] nrc[MW] len
#endif
#define rcs (fw + PW)
#define nfw (fw + 2 * PW)
#define nrc (fw + 2 * PW + MW)
    // per group after the packed reads: SMEM ring (2 ints per entry), cold state, two windows of 16-bit LCPs
    const lds_int ring = (lds_int)(reinterpret_cast<int*>(smem_raw + (SEED_QUERY_LDS ? (size_t)GROUPS * stride * 8 : 0)) +
                                   (size_t)gib * (2 * lcap + ST_WORDS + (W + 1) / 2));
    const lds_int sm_se = ring;               // start | end << 16
    const lds_int sm_cnt = ring + lcap;
    const lds_int st = ring + 2 * lcap;
    const lds_u16 wl = (lds_u16)(st + ST_WORDS);
    // the wavefront's ticket chunk: [0] tickets handed out, [2..3] first ticket
    const lds_int wv = (lds_int)(reinterpret_cast<int*>(smem_raw + seed_lds_group_bytes(G, stride, lcap)) + (threadIdx.x >> 6) * 4);
    const glb_ent sa = (glb_ent)A.I.sa;
    const glb_u64 pac = (glb_u64)A.I.pac;
    const glb_rmi l2 = (glb_rmi)A.I.l2, l1 = (glb_rmi)A.I.l1;
    const i64 n = A.I.n;
    const int hits_per_smem = A.opt.hits_per_smem;
#define GBALLOT(p_) ((__ballot(p_) >> gbase) & GFULL)
#define LD64(lo_) (((i64)st[(lo_) + 1] << 32) | (u64)(unsigned)st[lo_])
#define ST64(lo_, v_) do { const i64 v__ = (v_); st[lo_] = (int)(unsigned)(v__ & 0xffffffffll); st[(lo_) + 1] = (int)(v__ >> 32); } while (0)
#define FLAG(f_) ((st[ST_FLAGS] & (f_)) != 0)
#define SETFLAG(f_, v_) do { st[ST_FLAGS] = (v_) ? (st[ST_FLAGS] | (f_)) : (st[ST_FLAGS] & ~(f_)); } while (0)
#define LDS_HANDOFF() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                           __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

    // ---- per-read registers (group-uniform) ----------------------------------------------------------------------
    int pc = PC_FETCH, phase = PH_CTRL;
    int pivot = 0, l_seq = 0, msl = A.opt.min_seed_len, min_intv = 1;
    bool has_n = false;
    // the request in flight
    int q_kind = 0, q_mode = 0, off = 0, vlen = 0, capc = 0;
    bool q_rc = false, q_exact = false;
    i64 base = 0, lo = -1, hi = n;
    int stepk = 0;
    u64 wq = 0;
    unsigned acc_searches = 0, acc_windows = 0;             // of this group's completed reads
    if (lane == 0) wv[0] = TICKET_CHUNK;                    // empty chunk
    LDS_HANDOFF();

#ifdef SEED_PROF
    // diagnostic build: wall-clock ticks (100 MHz) a wavefront spends in each section of the loop body, summed over
    // groups into counters[4..]; marks sit at points every lane still in the loop passes
    unsigned prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long prof_t = wall_clock64();
#define PROF_MARK(i_) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now__ = wall_clock64(); prof[i_] += (unsigned)(now__ - prof_t); prof_t = now__; } while (0)
#else
#define PROF_MARK(i_) do { } while (0)
#endif
    for (;;) {
        // ================= control: reads without a request in flight produce the next one =========================
        bool newreq = false;
        if (phase == PH_CTRL) {
            // One pass over the states in the order reads usually flow through them, so a read takes several hops per
            // pass (a switch in a loop costs the wavefront one full pass per hop of its slowest read); the outer
            // loop only repeats for the rare backward hops.
            bool have = false;
#define AT(pc_) (!have && pc == (pc_))
            do {
                if (AT(PC_ZZ_TOP)) {       // zig-zag loop head (:1724-1737, :1969)
                    if (st[ST_ZZ_SP] >= st[ST_ZZ_NEXT] || ++st[ST_ZZ_GUARD] > 4 * l_seq + 16) pc = PC_ZZ_END;
                    else if (FLAG(F_ZZ_CHECK) && has_n && is_n(nfw, st[ST_ZZ_SP])) {
                        if (l_seq - st[ST_ZZ_SP] < msl) { pivot = l_seq; st[ST_ZZ_SP] = l_seq; }
                        else { st[ST_ZZ_SP] += 1; pivot = pivot + 1; }
                    } else { q_kind = K_ZZ_LEFT; have = true; }
                }
                if (AT(PC_ZZ_RIGHT)) { q_kind = K_ZZ_RIGHT; have = true; }
                if (AT(PC_ZZ_END)) {       // set_forward_pivot(raux, next_pivot) (:1893, :2125)
                    pivot = st[ST_ZZ_NEXT];
                    pc = FLAG(F_ZZ_RET_ONEPOS) ? PC_R2_AFTER : PC_AFTER_STEP1;
                }
                if (AT(PC_AFTER_STEP1)) {  // re-seeding loop entry (:921-923)
                    SETFLAG(F_REC, false);
                    st[ST_AFTER] = st[ST_N_SMEMS];
                    if (A.opt.rounds < 2) pc = PC_ALLPOS_TOP;
                    else if (FLAG(F_LDS_OVF)) pc = PC_DONE;          // re-run in the next tier (bigger LDS ring)
                    else { st[ST_R2_K] = st[ST_BEFORE]; pc = PC_R2_LOOP; }
                }
                if (AT(PC_R2_AFTER)) {     // (:945-946)
                    min_intv = st[ST_R2_SAVED];
                    pivot = st[ST_R2_NEXT];
                    pc = PC_R2_LOOP;
                }
                if (AT(PC_R2_LOOP)) {      // (:923-947) + OnePos entry (:1917-1930)
                    int k = st[ST_R2_K];
                    const int kb = st[ST_BEFORE], ke = st[ST_AFTER];
                    int qbeg = 0, qend = 0, cnt = 0;
                    bool take = false;
                    while (k < ke) {        // SMEMs that are too short or too frequent are not re-seeded (:929-931)
                        const int se = sm_se[k - kb];
                        cnt = sm_cnt[k - kb];
                        qbeg = se & 0xffff; qend = (int)((unsigned)se >> 16);
                        ++k;
                        if (!(qend - qbeg < A.opt.split_len || cnt > A.opt.split_width)) { take = true; break; }
                    }
                    st[ST_R2_K] = k;
                    if (!take) pc = PC_ALLPOS_TOP;
                    else {
                        st[ST_R2_NEXT] = pivot; st[ST_R2_SAVED] = min_intv;
                        pivot = (qbeg + qend) >> 1;
                        min_intv = cnt + 1;
                        if (has_n && is_n(nfw, pivot)) {
                            pivot = (l_seq - pivot < msl) ? l_seq : pivot + 1;
                            pc = PC_R2_AFTER;                         // backward hop (reads with N only)
                        } else if (pivot != 0 && !(has_n && is_n(nfw, pivot - 1))) { q_kind = K_OP_MEM; have = true; }
                        else { q_kind = K_OP_SMEM; have = true; }
                    }
                }
                if (AT(PC_ALLPOS_TOP)) {   // Learned_getSMEMsAllPosOneThread loop head (:916) + step1 entry (:1691-1723)
                    if (pivot >= l_seq || ++st[ST_AP_GUARD] > 4 * l_seq + 16) pc = PC_R3_INIT;
                    else {
                        st[ST_BEFORE] = st[ST_N_SMEMS]; st[ST_SM_BASE] = st[ST_BEFORE]; SETFLAG(F_REC, true);
                        if (has_n && is_n(nfw, pivot)) {
                            pivot = (l_seq - pivot < msl) ? l_seq : pivot + 1;
                            pc = PC_AFTER_STEP1;                      // backward hop (reads with N only)
                        } else if (pivot != 0 && !(has_n && is_n(nfw, pivot - 1))) {
                            // zig-zag entry: the loop head's checks pass trivially (sp = pivot < next = l_seq, no N here)
                            st[ST_ZZ_NEXT] = l_seq; SETFLAG(F_ZZ_CHECK, true); SETFLAG(F_ZZ_RET_ONEPOS, false); st[ST_ZZ_SP] = pivot; st[ST_ZZ_GUARD] = 1;
                            q_kind = K_ZZ_LEFT; have = true;
                        } else { q_kind = K_S1_RIGHT; have = true; }
                    }
                }
                if (AT(PC_R3_INIT)) {      // src/bwamem.cpp:1385-1394
                    if (A.opt.rounds >= 3 && A.opt.max_mem_intv > 0 && !FLAG(F_LDS_OVF)) {
                        min_intv = A.opt.max_mem_intv;
                        msl = A.opt.min_seed_len + 1;
                        pivot = 0;
                        pc = PC_R3_TOP;
                    } else pc = PC_DONE;
                }
                if (AT(PC_R3_TOP)) {       // Learned_bwtSeedStrategyAllPosOneThread loop head (:982-1012)
                    for (;;) {
                        if (!(pivot < l_seq - msl + 1)) { pc = PC_DONE; break; }
                        if (has_n && is_n(nfw, pivot)) { pivot = pivot + 1; continue; }
                        const int valid = first_n(nfw, has_n, pivot, l_seq) - pivot;
                        if (valid < msl) { pivot = pivot + valid; continue; }
                        q_kind = K_R3; have = true;
                        break;
                    }
                }
                if (AT(PC_DONE)) {         // publish the read's SMEM count / hit count; its slots are already written
                    const unsigned long long ticket = ((unsigned long long)(unsigned)st[ST_TICKET_HI] << 32) | (unsigned)st[ST_TICKET_LO];
                    const bool ovf = st[ST_N_SMEMS] > cap || FLAG(F_LDS_OVF);
                    if (t == 0) {
                        const i64 rid = A.pending ? A.pending[ticket] : (i64)ticket;
                        A.slot_cnt[rid] = ovf ? 0 : st[ST_N_SMEMS];
                        A.slot_hits[rid] = ovf ? 0 : LD64(ST_HITS_LO);
                        A.slot_loc[rid] = ((i64)A.tier << 40) | (i64)ticket;
                        if (ovf) A.ovf_list[atomicAdd(&A.counters[2], 1ull)] = rid;
                    }
                    if (!ovf) { acc_searches += (unsigned)st[ST_SEARCHES]; acc_windows += (unsigned)st[ST_WINDOWS]; }
                    pc = PC_FETCH;
                }
                if (AT(PC_FETCH)) {        // pull the next read (reads differ 3x in cost: dynamic hand-out, no static deal)
                    // One global atomic per TICKET_CHUNK reads: the wavefront keeps a chunk in LDS and its groups draw
                    // from it with an LDS atomic.  (One global atomic per read on a single address caps the whole
                    // kernel at ~30 M reads/s.)  The groups that are here together run in lockstep, so the refill
                    // below cannot race with another draw of the same wavefront.
                    int idx = TICKET_CHUNK;
                    if (t == 0) idx = __hip_atomic_fetch_add(wv, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    const u64 old_base = ((u64)(unsigned)wv[3] << 32) | (unsigned)wv[2];
                    const bool over = t == 0 && idx >= TICKET_CHUNK;
                    const unsigned long long mo = __ballot(over);
                    unsigned long long ticket = old_base + (unsigned)idx;
                    if (mo) {
                        // k groups need tickets from a fresh chunk; a wavefront may hold more groups than a chunk
                        const int first = __ffsll((long long)mo) - 1;
                        const int kreq = __popcll(mo), alloc = kreq > TICKET_CHUNK ? kreq : TICKET_CHUNK;
                        unsigned long long nb = 0;
                        if (lane == first) {
                            nb = atomicAdd(&A.counters[0], (unsigned long long)alloc);
                            const unsigned long long wb = nb + (unsigned)(alloc - TICKET_CHUNK);
                            wv[0] = kreq - (alloc - TICKET_CHUNK); wv[2] = (int)(unsigned)(wb & 0xffffffffull); wv[3] = (int)(wb >> 32);
                        }
                        nb = __shfl(nb, first);
                        if (over) ticket = nb + (unsigned)__popcll(mo & ((1ull << lane) - 1ull));
                    }
                    ticket = __shfl(ticket, gbase);
                    if (ticket >= (unsigned long long)A.nreads) pc = PC_EXIT;
                    else {
                        const i64 rid = A.pending ? A.pending[ticket] : (i64)ticket;
                        const u64* src = A.packed + rid * stride;
#if SEED_QUERY_LDS
                        // stage the packed read in LDS (coalesced 8-byte loads; the length word travels in the same
                        // batch of loads) and clear the cold state
                        const int lw = (stride - 1) % G;               // the lane that loads the length word
                        u64 lenw = 0;
                        for (int k = t; k < stride; k += G) {
                            u64 v = src[k];
                            rd[k] = v;
                            if (k == stride - 1) lenw = v;
                        }
                        lenw = (u64)(unsigned)__shfl((int)lenw, gbase + lw);
#else
                        // the searches read the record in place; only its last word (length, "has an N" flag) is needed now
                        fw = (q_u64)src;
                        const u64 lenw = fw[stride - 1];
#endif
                        l_seq = (int)(lenw & 0x7fffffffull);           // k_pack_reads: length | has-N flag << 31
                        has_n = ((lenw >> 31) & 1ull) != 0;
                        for (int k = t; k < ST_WORDS; k += G) st[k] = 0;
                        LDS_HANDOFF();
                        if (l_seq <= 0 || l_seq > MAX_READ_LEN) {
                            // the reference exits on reads longer than LEARNED_MAX_READ_LEN (src/bwamem.cpp:1259-1262);
                            // here such a read yields no seeds and is flagged through slot_cnt = -1
                            if (t == 0) { A.slot_cnt[rid] = l_seq > MAX_READ_LEN ? -1 : 0; A.slot_hits[rid] = 0; A.slot_loc[rid] = 0; }
                            // stays in PC_FETCH
                        } else {
                            st[ST_TICKET_LO] = (int)(unsigned)(ticket & 0xffffffffull);
                            st[ST_TICKET_HI] = (int)(ticket >> 32);
                            pivot = 0; msl = A.opt.min_seed_len; min_intv = 1;
                            pc = PC_ALLPOS_TOP;
                            if (!(has_n && is_n(nfw, 0))) {                       // first step of PC_ALLPOS_TOP at pivot 0, inlined
                                st[ST_AP_GUARD] = 1; SETFLAG(F_REC, true);
                                q_kind = K_S1_RIGHT; have = true;
                            }
                        }
                    }
                }
            } while (!have && pc != PC_EXIT);
#undef AT
            if (pc == PC_EXIT) {
                if (t == 0) {
                    atomicAdd(&A.counters[1], (unsigned long long)acc_searches); atomicAdd(&A.counters[3], (unsigned long long)acc_windows);
#ifdef SEED_PROF
                    for (int k = 0; k < 8; ++k) atomicAdd(&A.counters[4 + k], (unsigned long long)prof[k]);
#endif
                }
                break;
            }
            newreq = true;
        }
        PROF_MARK(0);
        if (newreq) {
            // ---- the request: query = bases [off, off+vlen) of one strand; first window at the model's prediction
            q_rc = q_kind == K_ZZ_LEFT;
            off = q_rc ? l_seq - 1 - pivot : pivot;
            vlen = first_n(q_rc ? nrc : nfw, has_n, off, l_seq) - off;
            q_exact = q_kind == K_S1_RIGHT || q_kind == K_ZZ_RIGHT || q_kind == K_OP_SMEM;
            q_mode = q_kind == K_R3 ? 2 : ((q_exact || min_intv != 1) ? 1 : 0);
            st[ST_SEARCHES] = st[ST_SEARCHES] + 1;
            wq = ext_l(q_rc ? rcs : fw, off);                 // first 32 bases of the query: every window compares against it
            u64 key = wq;
            if (vlen < 32) key |= (~0ull) >> (2 * vlen);          // T-pad short queries like Tokenization (:813-817)
            u64 err;
            const i64 pos = rmi_lookup(l2, l1, A.I.n_l1, A.I.shift, n, key, err);
#if SEED_USE_ERR
            // the partition point lies in [pos - below, pos + above]; the window has to hold it and its left neighbour.
            // If that span fits, centre it; else split the window in the proportion of the two bounds.
            const i64 below = (i64)((err >> 32) & 0x3fffffffull) + 1, above = (i64)(err & 0x7fffffffull);
            const i64 span = below + above + 1;
            base = span <= W ? pos - below - (W - span) / 2 : pos - (below * W) / span;
#else
            base = pos - W / 2;
#endif
#if SEED_ALIGN_WIN
            // Round 6: the first window starts on a multiple of four entries (64 bytes): 12 entries = 192 bytes from such a start lie in exactly two
            // 128-byte lines (start % 128 is 0 or 64), where an arbitrary start touches three in 3 of 8 cases (2.375 on average); the placement
            // moves by at most two entries.  Any base is a correct base (the window is a hint: a miss costs another window).
            if constexpr (W % 4 == 0) base = (base + 2) & ~3ll;
#endif
            if (base < 0) base = 0;
            if (base > n - W) base = n - W;
            lo = -1; hi = n; stepk = 0; capc = vlen;
            phase = PH_PART;
        }
        PROF_MARK(1);

        // ================= window: E coalesced loads of G entries each, issued together, then the compares ===========
        const q_u64 s = q_rc ? rcs : fw;
        int lcp[E];
        bool less[E];
        u64 ep[E];
        {
            u64 ek[E];
#pragma unroll
            for (int e = 0; e < E; ++e) { ek[e] = sa[base + e * G + t].key; ep[e] = sa[base + e * G + t].pos; }
            st[ST_WINDOWS] = st[ST_WINDOWS] + 1;
            window_compare<E>(pac, n, s, wq, off, capc, ek, ep, lcp, less);
        }
        PROF_MARK(2);
        u64 m = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool p = phase == PH_PART ? less[e] : ((lcp[e] >= capc) == (phase == PH_EDGE_UP));
            m |= GBALLOT(p) << (e * G);
        }
        // slots <= lo are known true, slots >= hi known false (also removes the far side of a clipped edge window)
        if (stepk != 0)                  // not the first window of a partition search: a bracket exists
            m = (m | lowmask((int)((lo - base + 1 > W) ? W : (lo - base + 1 < 0 ? 0 : lo - base + 1)))) &
                lowmask((int)((hi - base > W) ? W : (hi - base < 0 ? 0 : hi - base)));
        const int P = __popcll(m);
        const bool at_lo = base == 0 || (phase == PH_EDGE_UP && base == lo + 1);
        const bool at_hi = base + W == n || (phase == PH_EDGE_DN && base + W == hi);
        const bool found = (P > 0 && P < W) || (P == 0 && at_lo) || (P == W && at_hi);

        // ================= resolve ================================================================================
        bool go_level = false, finished = false;
        int L = 0, nb_lo = 0, nb_hi = 0, lf = 0;
        i64 s_edge = 0, e_edge = 0, cb = base;
        int r_L = 0;
        i64 r_start = 0, r_count = 1, r_T = -1;   // r_T: text position of slot r_start when this window holds it (only used when r_count == 1)
        bool r_emit = false;
        int cl[E];                       // LCPs of the cached partition window (this lane's slots)
        bool cache_in_lds = phase != PH_PART;
        if (found) {
            // LCPs of the two slots around the flip
            const int v2 = slot_pair<G, E>(lcp, t, P - 1, P);
            const int lm = P > 0 ? (v2 >> 16) : -1;
            const int lp = P < W ? (v2 & 0xffff) : -1;
            if (phase == PH_PART) {
                // slots [0,P) sort before the query; the longest match is at one of the two boundary neighbours
                const int c = (lm >= lp) ? P - 1 : P;
                L = lm >= lp ? lm : lp;
                r_L = L; r_start = base + c; r_count = 1;
                if (q_exact || q_kind == K_R3) {              // the kinds that emit: an interval of ONE suffix is this slot, and its position is here
                    u64 tc = 0;
#pragma unroll
                    for (int e = 0; e < E; ++e) tc = (e * G + t == c) ? ep[e] : tc;
                    r_T = (i64)(((u64)(unsigned)group_or<G>((int)(tc >> 32)) << 32) | (u64)(unsigned)group_or<G>((int)(unsigned)tc));
                }
                if (q_mode == 0 || (q_mode == 2 && L < msl)) finished = true;       // (:1204-1208)
                else {
                    s_edge = e_edge = base + c;
                    lf = LF_NEED_LO | LF_NEED_HI;
                    go_level = true;
                }
            } else {
                L = st[ST_L]; nb_lo = st[ST_NB_LO]; nb_hi = st[ST_NB_HI]; lf = st[ST_LF];
                s_edge = LD64(ST_SE_LO); e_edge = LD64(ST_EE_LO); cb = LD64(ST_CB_LO);
                if (phase == PH_EDGE_DN) { s_edge = base + P; nb_lo = P > 0 ? lm : 0; lf &= ~LF_NEED_LO; }
                else { e_edge = base + P - 1; nb_hi = P < W ? lp : 0; lf &= ~LF_NEED_HI; }
                go_level = true;
            }
        } else {
            if (P == W) lo = base + W - 1; else hi = base;
            bool stop = false;
            if (phase != PH_PART && !q_exact) {
                // the interval is not emitted at the level where the walk stops (left extensions, third round): it is
                // enough to know that it reached min_intv suffixes
                s_edge = LD64(ST_SE_LO); e_edge = LD64(ST_EE_LO);
                if (phase == PH_EDGE_DN ? (e_edge - hi + 1 >= (i64)min_intv) : (lo - s_edge + 1 >= (i64)min_intv)) {
                    L = st[ST_L]; nb_lo = st[ST_NB_LO]; nb_hi = st[ST_NB_HI]; lf = st[ST_LF]; cb = LD64(ST_CB_LO);
                    if (phase == PH_EDGE_DN) { s_edge = hi; nb_lo = L; lf &= ~LF_NEED_LO; }
                    else { e_edge = lo; nb_hi = L; lf &= ~LF_NEED_HI; }
                    stop = go_level = true;
                }
            }
            if (!stop) {
                if (lo < 0) { base = hi - ((i64)W << stepk); ++stepk; if (base < 0) base = 0; }                       // gallop down
                else if (hi >= n) { base = lo + 1 + (((i64)W << stepk) - W); ++stepk; if (base > n - W) base = n - W; }  // gallop up
                else if (hi - lo <= W - 1) { base = lo; if (base > n - W) base = n - W; }                            // final window
                else base = lo + (hi - lo) / 2 - W / 2;                                                              // bisect
            }
        }
        PROF_MARK(3);
        if (go_level) {
            // Walk the levels L0 > L1 > ... on the partition window's LCPs, held in registers (fresh from the compare, or
            // re-loaded from LDS after an edge request): per level one ballot mask, two bit scans and one DPP
            // reduction -- no LDS round trip.  An edge that leaves the cached window becomes an edge request.
#pragma unroll
            for (int e = 0; e < E; ++e) cl[e] = cache_in_lds ? (int)wl[e * G + t] : lcp[e];
            for (;;) {
                if (lf & (LF_NEED_LO | LF_NEED_HI)) {
                    u64 ge = 0;
#pragma unroll
                    for (int e = 0; e < E; ++e) ge |= GBALLOT(cl[e] >= L) << (e * G);
                    const u64 nz = ~ge & WFULL;                       // cached slots that do not reach level L
                    const i64 klo = s_edge - cb;                      // cached slots [0,klo) lie below the current edge
                    const i64 khi = e_edge - cb;                      // cached slots (khi,W) lie above it
                    const bool in_lo = (lf & LF_NEED_LO) && klo > 0 && klo <= W;
                    const bool in_hi = (lf & LF_NEED_HI) && khi >= -1 && khi < W - 1;
                    const u64 zlo = in_lo ? (nz & lowmask((int)klo)) : 0ull;
                    const u64 zhi = in_hi ? (nz & ~lowmask((int)khi + 1)) : 0ull;
                    const int hz = zlo ? 63 - __clzll((long long)zlo) : -1;
                    const int lz = zhi ? __ffsll((long long)zhi) - 1 : -1;
                    const int nb2 = slot_pair<G, E>(cl, t, hz, lz);
                    if (lf & LF_NEED_LO) {
                        if (in_lo) {
                            if (zlo) { s_edge = cb + hz + 1; nb_lo = nb2 >> 16; lf &= ~LF_NEED_LO; }
                            else { s_edge = cb; if (cb == 0) { nb_lo = 0; lf &= ~LF_NEED_LO; } }
                        } else if (s_edge == 0) { nb_lo = 0; lf &= ~LF_NEED_LO; }
                    }
                    if (lf & LF_NEED_HI) {
                        if (in_hi) {
                            if (zhi) { e_edge = cb + lz - 1; nb_hi = nb2 & 0xffff; lf &= ~LF_NEED_HI; }
                            else { e_edge = cb + W - 1; if (cb + W >= n) { nb_hi = 0; lf &= ~LF_NEED_HI; } }
                        } else if (e_edge == n - 1) { nb_hi = 0; lf &= ~LF_NEED_HI; }
                    }
                    if (lf & LF_NEED_LO) {                            // the run leaves the cached window: edge request
                        phase = PH_EDGE_DN; lo = -1; hi = s_edge; stepk = 1;
                        base = s_edge - W; if (base < 0) base = 0;
                        break;
                    }
                    if (lf & LF_NEED_HI) {
                        phase = PH_EDGE_UP; lo = e_edge; hi = n; stepk = 1;
                        base = e_edge + 1; if (base > n - W) base = n - W;
                        break;
                    }
                }
                const i64 cnt = e_edge - s_edge + 1;
                const int nxt = nb_lo > nb_hi ? nb_lo : nb_hi;
                if (q_mode == 1) {                                  // (:2568-2573, :2936-2940)
                    if (cnt >= (i64)min_intv) { r_L = L; r_start = s_edge; r_count = cnt; finished = true; break; }
                } else {
                    if (cnt >= (i64)min_intv) {                     // :1243-1251
                        if (lf & LF_HAVE_LAST) { r_count = LD64(ST_LAST_CNT_LO); r_start = LD64(ST_LAST_S_LO); }
                        else { r_count = cnt; r_start = s_edge; }
                        r_L = L + 1;
                        finished = true;
                        break;
                    }
                    if (nxt < msl) { r_L = msl; r_start = s_edge; r_count = cnt; finished = true; break; }   // :1252-1258
                    lf |= LF_HAVE_LAST;
                    ST64(ST_LAST_CNT_LO, cnt);
                    ST64(ST_LAST_S_LO, s_edge);
                }
                L = nxt;
                lf = (lf & LF_HAVE_LAST) | ((nb_lo >= L && s_edge > 0) ? LF_NEED_LO : 0) | ((nb_hi >= L && e_edge < n - 1) ? LF_NEED_HI : 0);
            }
            if (finished) {
                if (q_mode == 2) {
                    r_emit = r_count < (i64)min_intv;              // :1265
                    if (r_L < msl) r_L = msl;
                }
            } else {
                capc = L;
                st[ST_L] = L; st[ST_NB_LO] = nb_lo; st[ST_NB_HI] = nb_hi; st[ST_LF] = lf;
                ST64(ST_SE_LO, s_edge); ST64(ST_EE_LO, e_edge); ST64(ST_CB_LO, cb);
                if (!cache_in_lds) {                              // first edge request of this search: park the cache
#pragma unroll
                    for (int e = 0; e < E; ++e) wl[e * G + t] = (unsigned short)cl[e];
                }
            }
        }
        PROF_MARK(4);
        if (finished) {
            // ---- apply the search result to the read's pivot logic ------------------------------------------------------
            bool emit = false;
            const int e_start = pivot;
            int e_end = pivot;
            switch (q_kind) {
            case K_S1_RIGHT:          // (:1852-1893)
            case K_ZZ_RIGHT:          // (:1846-1848)
            case K_OP_SMEM:           // (:2093-2125)
                emit = r_L >= msl; e_end = pivot + r_L;
                break;
            case K_ZZ_LEFT:           // (:1774-1777)
                pivot = pivot - r_L + 1;
                pc = (st[ST_ZZ_NEXT] - pivot < msl) ? PC_ZZ_END : PC_ZZ_RIGHT;
                break;
            case K_OP_MEM:            // (:1967-1969)
                st[ST_ZZ_NEXT] = pivot + r_L; SETFLAG(F_ZZ_CHECK, false); SETFLAG(F_ZZ_RET_ONEPOS, true); st[ST_ZZ_SP] = pivot; st[ST_ZZ_GUARD] = 0;
                pc = PC_ZZ_TOP;
                break;
            default:                  // K_R3 (:1204-1208, :1265-1281)
                emit = r_emit; e_end = pivot + r_L;
                break;
            }
            if (emit) {               // kv_push of mem_tl + hits (:2639-2657, :1266-1277)
                const int ns = st[ST_N_SMEMS];
                // One occurrence, and its text position is in this window's registers: the record carries the position.  If the SMEM
                // would be re-seeded (:929-931: long enough, at most split_width occurrences) the region -- a zig-zag of ~20 searches
                // around its middle that asks "how long a piece of this locus occurs twice" -- is left to k_reseed, which answers that
                // from the plcp table, and is skipped below by giving the ring entry an occurrence count no split_width admits.
                const bool have_pos = r_count == 1 && r_T >= 0 && !cache_in_lds;
                // (Not where the locus lies in repeated sequence -- its suffix shares min_seed_len bases or more with a neighbour in the suffix
                // array: there the table would send nearly every search of the region back, and the searches are better done here.)
                const bool defer = have_pos && A.defer != 0 && FLAG(F_REC) && ns < DEFER_MAX_K && ns < cap && A.opt.rounds >= 2 &&
                                   e_end - e_start >= A.opt.split_len && A.opt.split_width >= 1 && nb_lo < msl && nb_hi < msl;
                if (ns < cap && t == 0) {
                    const unsigned long long ticket = ((unsigned long long)(unsigned)st[ST_TICKET_HI] << 32) | (unsigned)st[ST_TICKET_LO];
                    SlotRec sr;
                    sr.start = e_start; sr.end = e_end; sr.count = r_count;
                    sr.sa_start = have_pos ? (SLOT_POS | (defer ? SLOT_DEFER : 0ll) | r_T) : r_start;
                    A.slots[(i64)ticket * cap + ns] = sr;
                }
                if (FLAG(F_REC)) {
                    const int k = ns - st[ST_SM_BASE];
                    if (k < lcap) {
                        // group-uniform redundant LDS stores (every lane writes the same value): no hand-off needed
                        sm_se[k] = e_start | (e_end << 16);
                        sm_cnt[k] = (defer || r_count > (i64)INT_MAX) ? INT_MAX : (int)r_count;
                    } else SETFLAG(F_LDS_OVF, true);
                }
                st[ST_N_SMEMS] = ns + 1;
                i64 h = r_count;
                if (hits_per_smem > 0 && h > hits_per_smem) h = hits_per_smem;
                h += LD64(ST_HITS_LO);
                ST64(ST_HITS_LO, h);
            }
            switch (q_kind) {
            case K_S1_RIGHT: pivot = pivot + r_L; pc = PC_AFTER_STEP1; break;
            case K_ZZ_RIGHT: pivot = pivot + r_L; st[ST_ZZ_SP] = pivot; pc = PC_ZZ_TOP; break;
            case K_OP_SMEM: pivot = pivot + r_L; pc = PC_R2_AFTER; break;
            case K_R3: pivot = pivot + (r_L < msl ? msl : r_L); pc = PC_R3_TOP; break;
            default: break;
            }
            phase = PH_CTRL;
        }
        PROF_MARK(5);
    }
#undef PROF_MARK
#undef rcs
#undef nfw
#undef nrc
#undef GBALLOT
#undef LD64
#undef ST64
#undef FLAG
#undef SETFLAG
#undef LDS_HANDOFF
}


// ---- re-seeding of unique SMEMs: k_reseed --------------------------------------------------------------------------------------
// Round 2 (Learned_getSMEMsAllPosOneThread :923-947 -> Learned_getSMEMsOnePosOneThread :1897-2126) re-seeds every SMEM of at least
// split_len bases and at most split_width occurrences from its middle, asking for matches with MORE occurrences than the SMEM has.  For
// an SMEM with one occurrence -- nearly all of them -- every search of that zig-zag lies inside the SMEM, i.e. its query is a piece of
// the text at a known position u, and the answer "the longest prefix of text[u..] that occurs at least twice" is a property of u alone:
// plcp[u] = the longest common prefix of suffix u with its nearer suffix-array neighbour (valid while it stays inside the SMEM:
// plcp[u] < bases left to the SMEM's end; beyond that other loci decide and only a search can tell).  Leftward searches run on the
// reverse complement, i.e. on the mirror position n-1-u.  One LANE owns a read here and walks the zig-zag of each of its regions on
// that table (two 64-byte pieces staged in LDS) where k_seed spends ~20 window searches -- 40 % of all searches of a 150-bp read.
// What the table cannot answer -- a query that leaves the SMEM, a saturated entry, and a match of >= min_seed_len bases with two
// occurrences, which is an SMEM to emit and needs its suffix-array interval -- the lane searches itself (lane_search: the model's
// prediction, gallop + bisection on single entries, then the level walk; ~0.3 such searches per read), and the walk goes on behind it.
// SMEMs found are appended to the read's slots; their order within a read is not observable (the consumer sorts them by (start, end),
// src/bwamem.cpp:1397; equal keys have equal hit lists).
__device__ __forceinline__ u64 ext_g(const u64* __restrict__ w, int s) {
    const int k = s >> 5, sh = (s & 31) * 2;
    const u64 a = w[k], b = w[k + 1];
    return sh ? (a << sh) | (b >> (64 - sh)) : a;
}

struct LaneQuery {
    const u64* s;      // packed words of the strand the query lies on
    int off, vlen;     // query = bases [off, off + vlen) of it
    u64 wq;            // its first 32 bases
};

// window_compare for one suffix-array entry: capped LCP with the query and "sorts before the query"
__device__ __forceinline__ void lane_compare_ent(const DevIndex& I, const LaneQuery& q, int cap, const SaEnt e, int& lcp, bool& less) {
    const i64 nlimit = I.n - (i64)cap;
    int Lc = cap;
    if ((i64)e.pos > nlimit) Lc = (int)(I.n - (i64)e.pos);
    const u64 x = e.key ^ q.wq;
    bool lt = e.key < q.wq;
    int l = x ? (__clzll((long long)x) >> 1) : 32;
    if (!x && 32 < Lc) {
        const i64 p0 = (i64)e.pos + 32;
        for (int k = 1;; ++k) {
            const u64 wr = extract32(I.pac, p0 + 32 * (k - 1));
            const u64 qq = ext_g(q.s, q.off + 32 * k);
            const u64 y = wr ^ qq;
            if (y) { l += __clzll((long long)y) >> 1; lt = wr < qq; break; }
            l += 32;
            if (l >= Lc) break;
        }
    }
    if (l >= Lc) { lcp = Lc; less = (i64)e.pos < nlimit; }
    else { lcp = l; less = lt; }
}
__device__ __forceinline__ void lane_compare(const DevIndex& I, const LaneQuery& q, int cap, i64 slot, int& lcp, bool& less) {
    lane_compare_ent(I, q, cap, I.sa[slot], lcp, less);
}

// mem_search / right_smem_search with an occurrence floor (mode 1 of k_seed): L = the largest l <= maxLCP whose suffix-array interval
// holds >= min_intv suffixes; [start, start + count) = that interval.  One lane, single-entry probes.
// The common case costs two round trips: the model record, then LANE_W consecutive entries around its prediction fetched together; their
// LCPs are parked in the lane's LDS column (lw: 16-bit values, two per dword, dword j of lane L at [j * 256 + L]) and both the partition
// point and the interval usually lie inside.  Anything beyond that window is probed entry by entry (gallop, then bisection).
constexpr int LANE_W = 16;
struct LaneWin {
    uint32_t* lw;          // this lane's column
    i64 base;              // first slot of the cached window, -1: none
    __device__ __forceinline__ bool has(i64 slot) const { return base >= 0 && slot >= base && slot < base + LANE_W; }
    __device__ __forceinline__ int get(i64 slot) const { const int i = (int)(slot - base); return (int)((lw[(i >> 1) * 256] >> (16 * (i & 1))) & 0xffffu); }
};
// LCP of slot with the query, from the cached window when it holds the slot (values there are capped by the query length only,
// which compares the same way against any level L <= that)
__device__ __forceinline__ int lane_lcp(const DevIndex& I, const LaneQuery& q, const LaneWin& C, int cap, i64 slot) {
    if (C.has(slot)) return C.get(slot);
    int lc; bool ls;
    lane_compare(I, q, cap, slot, lc, ls);
    return lc;
}

__device__ __forceinline__ void lane_search(const DevIndex& I, const LaneQuery& q, int min_intv, uint32_t* lw, int& r_L, i64& r_start, i64& r_count) {
    const i64 n = I.n;
    u64 key = q.wq;
    if (q.vlen < 32) key |= (~0ull) >> (2 * q.vlen);
    u64 err;
    const i64 p = rmi_lookup((glb_rmi)I.l2, (glb_rmi)I.l1, I.n_l1, I.shift, n, key, err);
    LaneWin C; C.lw = lw; C.base = -1;
    int L = 0;
    i64 s = 0, e = 0;
    bool located = false;
    int lc; bool ls;
    {
        // first window, placed by the model's error bounds like k_seed's
        const i64 below = (i64)((err >> 32) & 0x3fffffffull) + 1, above = (i64)(err & 0x7fffffffull);
        const i64 span = below + above + 1;
        i64 base = span <= LANE_W ? p - below - (LANE_W - span) / 2 : p - (below * LANE_W) / span;
        if (base < 0) base = 0;
        if (base > n - LANE_W) base = n - LANE_W;
        SaEnt ent[LANE_W];
#pragma unroll
        for (int k = 0; k < LANE_W; ++k) ent[k] = I.sa[base + k];
        unsigned lessm = 0;
        int prev = 0;
#pragma unroll
        for (int k = 0; k < LANE_W; ++k) {
            lane_compare_ent(I, q, q.vlen, ent[k], lc, ls);
            lessm |= (ls ? 1u : 0u) << k;
            if (k & 1) lw[(k >> 1) * 256] = (uint32_t)prev | ((uint32_t)lc << 16); else prev = lc;
        }
        C.base = base;
        const int P = __popc(lessm);
        if (lessm == ((1u << P) - 1u) && ((P > 0 && P < LANE_W) || (P == 0 && base == 0) || (P == LANE_W && base + LANE_W == n))) {
            const int lm = P > 0 ? C.get(base + P - 1) : -1, lp = P < LANE_W ? C.get(base + P) : -1;
            L = lm >= lp ? lm : lp;
            s = e = base + (lm >= lp ? P - 1 : P);
            located = true;
        }
    }
    if (!located) {
        // partition point by single probes: [0, P) sort before the query.  lo = highest slot known to, hi = lowest slot known not to.
        i64 lo = -1, hi = n;
        lane_compare(I, q, q.vlen, p, lc, ls);
        if (ls) {
            lo = p;
            for (i64 st = 1; hi == n && lo < n - 1; st <<= 1) {
                i64 pr = lo + st; if (pr > n - 1) pr = n - 1;
                lane_compare(I, q, q.vlen, pr, lc, ls);
                if (ls) lo = pr; else hi = pr;
            }
        } else {
            hi = p;
            for (i64 st = 1; lo == -1 && hi > 0; st <<= 1) {
                i64 pr = hi - st; if (pr < 0) pr = 0;
                lane_compare(I, q, q.vlen, pr, lc, ls);
                if (ls) lo = pr; else hi = pr;
            }
        }
        while (hi - lo > 1) {
            const i64 mid = lo + (hi - lo) / 2;
            lane_compare(I, q, q.vlen, mid, lc, ls);
            if (ls) lo = mid; else hi = mid;
        }
        const i64 P = hi;
        const int lm = P > 0 ? lane_lcp(I, q, C, q.vlen, P - 1) : -1, lp = P < n ? lane_lcp(I, q, C, q.vlen, P) : -1;
        L = lm >= lp ? lm : lp;
        s = e = lm >= lp ? P - 1 : P;
    }
    for (;;) {
        // the run of slots around [s, e] that share >= L bases with the query: step by step through the cached window, beyond it
        // gallop and bisect (the predicate is true on a run)
        while (s > 0 && C.has(s - 1) && C.get(s - 1) >= L) --s;
        if (s > 0 && !C.has(s - 1)) {
            i64 good = s, bad = -1;
            for (i64 st = 1; good > 0; st <<= 1) {
                i64 pr = good - st; if (pr < 0) pr = 0;
                if (lane_lcp(I, q, C, L, pr) >= L) good = pr; else { bad = pr; break; }
            }
            while (bad >= 0 && good - bad > 1) {
                const i64 mid = bad + (good - bad) / 2;
                if (lane_lcp(I, q, C, L, mid) >= L) good = mid; else bad = mid;
            }
            s = good;
        }
        while (e < n - 1 && C.has(e + 1) && C.get(e + 1) >= L) ++e;
        if (e < n - 1 && !C.has(e + 1)) {
            i64 good = e, bad = n;
            for (i64 st = 1; good < n - 1; st <<= 1) {
                i64 pr = good + st; if (pr > n - 1) pr = n - 1;
                if (lane_lcp(I, q, C, L, pr) >= L) good = pr; else { bad = pr; break; }
            }
            while (bad < n && bad - good > 1) {
                const i64 mid = good + (bad - good) / 2;
                if (lane_lcp(I, q, C, L, mid) >= L) good = mid; else bad = mid;
            }
            e = good;
        }
        if (e - s + 1 >= (i64)min_intv) break;
        const int nlo = s > 0 ? lane_lcp(I, q, C, L, s - 1) : 0, nhi = e < n - 1 ? lane_lcp(I, q, C, L, e + 1) : 0;
        L = nlo > nhi ? nlo : nhi;
    }
    r_L = L; r_start = s; r_count = e - s + 1;
}

// first ambiguous base at/after `from` in a packed N mask (global memory)
__device__ __forceinline__ int first_n_g(const u64* __restrict__ mask, bool has_n, int from, int l_seq) {
    if (!has_n) return l_seq;
    int w = from >> 6;
    u64 m = mask[w] & (~0ull << (from & 63));
    const int nw = (l_seq + 63) >> 6;
    for (;;) {
        if (m) { const int p = w * 64 + __ffsll((long long)m) - 1; return p < l_seq ? p : l_seq; }
        if (++w >= nw) return l_seq;
        m = mask[w];
    }
}

struct PlcpView {           // the plcp table as one lane of k_reseed sees it: two staged windows in LDS, global memory beyond them
    const uint8_t* __restrict__ plcp;
    const uint32_t* win;    // this lane's column of the workgroup's window array: dword j at win[j * 256]
    i64 F0, R0;
    __device__ __forceinline__ int at(i64 pos, i64 w0, int dw0) const {
        const i64 d = pos - w0;
        if (d >= 0 && d < PLCP_WIN) return (int)((win[(dw0 + (int)(d >> 2)) * 256] >> (8 * (int)(d & 3))) & 0xffu);
        return (int)plcp[pos];
    }
    __device__ __forceinline__ int fwd(i64 pos) const { return at(pos, F0, 0); }
    __device__ __forceinline__ int rev(i64 pos) const { return at(pos, R0, PLCP_WIN / 4); }
};

// A region whose walk met something the table cannot answer: where it stands, and (filled in by k_reseed_search) the search's result.
struct BlkRec {
    i64 rid;
    int k;                  // the region's SMEM (slot index in the read)
    int pivot, next, guard;
    int stage;              // 0 right of the middle (-> next), 1 left of the pivot, 2 right of the new pivot; + BLK_MORE: the read's later regions follow
    int r_L;
    i64 r_start, r_count;
};

struct ReseedArgs {
    DevIndex I;
    const u64* packed;      // k_pack_reads output
    PackGeom geo;
    i64 nreads;
    meme_seed_opt opt;
    SlotRec* slots;         // tier 0: block r = read r
    int cap;
    int* slot_cnt;
    i64* slot_hits;
    i64* ovf_list;
    unsigned long long* counters;   // [1] searches, [2] overflowed reads, [12] searches inside the walk kernels, [13] reads with pending SMEMs, [14] blocked regions
    i64* pend_list;         // reads that hold SLOT_PEND records
    BlkRec* blk;            // blocked regions: the list this launch reads (k_reseed: writes), capacity blk_cap, its length in counters[blk_ctr]
    BlkRec* blk_out;        // k_reseed_resume<false>: the regions that block again, length in counters[blk_out_ctr]
    i64 blk_cap;
    int blk_ctr, blk_out_ctr;
};

// One SMEM more for a read.  OWNED: the calling lane is the only one working on the read (plain counter in a register, published by the
// caller); otherwise regions of one read may be in different lanes and the read's counters are bumped atomically.
template <bool OWNED>
struct SmemAppender {
    SlotRec* sl; int cap; int* cnt; i64* hits; i64* ovf_list; unsigned long long* counters; i64 rid; int hps;
    int ns; i64 hits_add;   // OWNED only
    __device__ __forceinline__ void push(int start, int end, i64 sa_start, i64 count) {
        const i64 h = (hps > 0 && count > hps) ? (i64)hps : count;
        if (OWNED) {
            if (ns < cap) { SlotRec sr; sr.start = start; sr.end = end; sr.sa_start = sa_start; sr.count = count; sl[ns] = sr; }
            ++ns; hits_add += h;
        } else {
            const int at = atomicAdd(cnt, 1);
            if (at < cap) { SlotRec sr; sr.start = start; sr.end = end; sr.sa_start = sa_start; sr.count = count; sl[at] = sr; }
            else if (at == cap) ovf_list[atomicAdd(&counters[2], 1ull)] = rid;     // the first one over: the read goes to the next tier (which rewrites its counters)
            atomicAdd(reinterpret_cast<unsigned long long*>(hits), (unsigned long long)h);
        }
    }
};

struct RegionState { int pivot, next, guard, stage; };

// The region (:1959-2084) as far as the table answers: stage 0 = mem_search to the right of the middle (-> next_pivot), then the zig-zag:
// 1 = to the left of the pivot, 2 = to the right of the new pivot (emits an SMEM of >= min_seed_len bases); min_intv = 2 throughout.
// Returns true when the search at (stage, pivot) has to be done for real; stage 3 = the region is finished.
template <bool OWNED>
__device__ __forceinline__ bool region_walk(const PlcpView& V, i64 n, i64 T0, int qbeg, int qend, int l_seq, int msl, RegionState& S, unsigned& hops,
                                            SmemAppender<OWNED>& ap, int& n_pend) {
    for (;;) {
        if (S.stage == 0) {
            const int plc = V.fwd(T0 + S.pivot);
            if (plc == 0 || plc == 255 || plc >= qend - S.pivot) return true;
            ++hops; S.next = S.pivot + plc; S.stage = 1;
        }
        if (S.stage == 1) {
            if (S.pivot >= S.next || ++S.guard > 4 * l_seq + 16) { S.stage = 3; return false; }
            const int plc = (S.pivot >= qbeg && S.pivot < qend) ? V.rev(n - 1 - (T0 + S.pivot)) : 0;
            if (plc == 0 || plc == 255 || plc >= S.pivot - qbeg + 1) return true;
            ++hops; S.pivot = S.pivot - plc + 1;
            if (S.next - S.pivot < msl) { S.stage = 3; return false; }
            S.stage = 2;
        }
        const int plc = (S.pivot >= qbeg && S.pivot < qend) ? V.fwd(T0 + S.pivot) : 0;
        if (plc == 0 || plc == 255 || plc >= qend - S.pivot) return true;
        if (plc >= msl) {
            // an SMEM of plc bases with at least two occurrences (:2639-2657): its length is known, so the walk goes on; the
            // suffix-array interval is looked up later, together with everybody else's (k_reseed_emit).  (The second pass has no
            // such batch behind it: there the search is done on the spot.)
            if (!OWNED) return true;
            ap.push(S.pivot, S.pivot + plc, SLOT_PEND, 0);
            ++n_pend;
        }
        ++hops; S.pivot = S.pivot + plc; S.stage = 1;
    }
}

// the query of the search a blocked region waits for
__device__ __forceinline__ LaneQuery region_query(const u64* __restrict__ rec, const PackGeom& geo, int l_seq, bool has_n, const RegionState& S) {
    LaneQuery q;
    const bool left = S.stage == 1;
    q.s = rec + (left ? geo.W : 0);
    q.off = left ? l_seq - 1 - S.pivot : S.pivot;
    q.vlen = first_n_g(rec + 2 * geo.W + (left ? geo.MW : 0), has_n, q.off, l_seq) - q.off;
    q.wq = ext_g(q.s, q.off);
    return q;
}

// ... and what its result means for the region (:1967-1969, :1774-1777, :1846-1848)
template <bool OWNED>
__device__ __forceinline__ void region_apply(RegionState& S, int msl, int r_L, i64 r_start, i64 r_count, SmemAppender<OWNED>& ap) {
    if (S.stage == 0) { S.next = S.pivot + r_L; S.stage = 1; }
    else if (S.stage == 1) { S.pivot = S.pivot - r_L + 1; S.stage = (S.next - S.pivot < msl) ? 3 : 2; }
    else {
        if (r_L >= msl) ap.push(S.pivot, S.pivot + r_L, r_start, r_count);
        S.pivot = S.pivot + r_L; S.stage = 1;
    }
}

// the two table windows of a region into the lane's LDS column: eight aligned 16-byte loads in flight together
__device__ __forceinline__ void stage_plcp_windows(const uint8_t* __restrict__ plcp, const PlcpView& V, uint32_t* win) {
    uint4 v[2 * (PLCP_WIN / 16)];
#pragma unroll
    for (int q = 0; q < PLCP_WIN / 16; ++q) {
        v[q] = *reinterpret_cast<const uint4*>(plcp + V.F0 + 16 * q);
        v[PLCP_WIN / 16 + q] = *reinterpret_cast<const uint4*>(plcp + V.R0 + 16 * q);
    }
#pragma unroll
    for (int q = 0; q < 2 * (PLCP_WIN / 16); ++q) {
        win[(4 * q + 0) * 256] = v[q].x; win[(4 * q + 1) * 256] = v[q].y; win[(4 * q + 2) * 256] = v[q].z; win[(4 * q + 3) * 256] = v[q].w;
    }   // (all eight loads in flight together: staging in two halves saves 16 VGPRs but adds a round trip per region)
}

constexpr int BLK_PER_READ = 1;      // blocked regions a read may leave behind in the first pass; its last record takes the rest of the read with it
constexpr int BLK_MORE = 16;

// Pass 1: one lane per read, all its regions.
__global__ void __launch_bounds__(256) k_reseed(ReseedArgs A) {
    // per lane two PLCP_WIN-byte windows of the table; dword j of lane L at [j * 256 + L]: whatever bytes the lanes ask for, the
    // bank is the lane's own
    __shared__ uint32_t win[2 * (PLCP_WIN / 4) * 256];
    __shared__ unsigned long long blk_hops, blk_lane;
    __shared__ unsigned blk_pend, blk_pend_base, blk_blk;
    __shared__ unsigned long long blk_blk_base;
    const uint8_t* __restrict__ plcp = A.I.plcp;
    const i64 n = A.I.n;
    const int msl = A.opt.min_seed_len, cap = A.cap;
    for (i64 r0 = (i64)blockIdx.x * blockDim.x; r0 < A.nreads; r0 += (i64)gridDim.x * blockDim.x) {
        if (threadIdx.x == 0) { blk_hops = 0; blk_lane = 0; blk_pend = 0; blk_blk = 0; }
        __syncthreads();
        const i64 r = r0 + threadIdx.x;
        unsigned hops = 0, lsearches = 0;
        int n_pend = 0, n_blk = 0, pend_ns = 0;
        BlkRec held[BLK_PER_READ];
        if (r < A.nreads) {
            const int c0 = A.slot_cnt[r];
            const int c = c0 > DEFER_MAX_K ? DEFER_MAX_K : c0;
            SlotRec* sl = A.slots + r * cap;
            const u64* rec = A.packed + r * A.geo.stride;           // fw[W] rc[W] nfw[MW] nrc[MW] len
            SmemAppender<true> ap;
            ap.sl = sl; ap.cap = cap; ap.hps = A.opt.hits_per_smem; ap.ns = c0; ap.hits_add = 0; ap.rid = r;
            int l_seq = 0;
            // the lanes of a wavefront take their j-th region together (a loop over the slot index would run the region code once per
            // slot position with a few lanes each)
            int k = -1;
            for (;;) {
                i64 f = 0;
                for (++k; k < c; ++k) { f = sl[k].sa_start; if (f & SLOT_DEFER) break; }
                if (k >= c) break;
                if (l_seq == 0) { const u64 lw = rec[A.geo.stride - 1]; l_seq = (int)(lw & 0x7fffffffull); }
                const int qbeg = sl[k].start, qend = sl[k].end;
                const i64 T0 = (f & SLOT_VAL) - qbeg;
                PlcpView V;
                V.plcp = plcp; V.win = win + threadIdx.x;
                RegionState S;
                S.pivot = (qbeg + qend) >> 1; S.next = 0; S.guard = 0; S.stage = 0;
                V.F0 = plcp_win_fwd(T0, S.pivot); V.R0 = plcp_win_rev(T0, S.pivot, n);
                stage_plcp_windows(plcp, V, win + threadIdx.x);
                if (region_walk<true>(V, n, T0, qbeg, qend, l_seq, msl, S, hops, ap, n_pend)) {
                    // blocked: to the batch search (k_reseed_search), resumed by k_reseed_resume.  This kernel does no search itself (it
                    // lives on its occupancy); a read's last record takes the read's remaining regions along.
                    BlkRec b;
                    b.rid = r; b.k = k; b.pivot = S.pivot; b.next = S.next; b.guard = S.guard; b.r_L = 0; b.r_start = 0; b.r_count = 0;
                    b.stage = S.stage | (n_blk == BLK_PER_READ - 1 ? BLK_MORE : 0);
#pragma unroll
                    for (int i = 0; i < BLK_PER_READ; ++i) if (i == n_blk) held[i] = b;
                    if (++n_blk == BLK_PER_READ) break;
                }
            }
            if (ap.ns != c0) {
                if (ap.ns > cap) {            // more SMEMs than the read's slots hold: the whole read goes to the next tier, as in k_seed
                    A.slot_cnt[r] = 0; A.slot_hits[r] = 0;
                    A.ovf_list[atomicAdd(&A.counters[2], 1ull)] = r;
                    n_pend = 0; n_blk = 0;
                } else { A.slot_cnt[r] = ap.ns; A.slot_hits[r] = A.slot_hits[r] + ap.hits_add; pend_ns = ap.ns; }
            }
        }
        // one global atomic per workgroup and list
        unsigned my_pend = 0, my_blk = 0;
        if (n_pend) my_pend = atomicAdd(&blk_pend, 1u);
        if (n_blk) my_blk = atomicAdd(&blk_blk, (unsigned)n_blk);
        if (hops | lsearches) { atomicAdd(&blk_hops, (unsigned long long)(hops + lsearches)); if (lsearches) atomicAdd(&blk_lane, (unsigned long long)lsearches); }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (blk_hops) atomicAdd(&A.counters[1], blk_hops);
            if (blk_lane) atomicAdd(&A.counters[12], blk_lane);
            if (blk_pend) blk_pend_base = (unsigned)atomicAdd(&A.counters[13], (unsigned long long)blk_pend);
            if (blk_blk) blk_blk_base = atomicAdd(&A.counters[A.blk_ctr], (unsigned long long)blk_blk);
        }
        __syncthreads();
        if (n_pend) A.pend_list[blk_pend_base + my_pend] = r | ((i64)pend_ns << 40);     // the read, and how many of its slots this pass filled
#pragma unroll
        for (int i = 0; i < BLK_PER_READ; ++i)
            if (i < n_blk) {
                const unsigned long long at = blk_blk_base + my_blk + i;
                if ((i64)at < A.blk_cap) A.blk[at] = held[i];
            }
        __syncthreads();
    }
}

// The suffix-array intervals of the SMEMs the walk found by their length alone: one lane per read of the pending list, every lane a
// search of its own (the SMEM itself is the query: all of it matches, the interval is what shares all of it) -- a dense batch, the lanes
// of a wavefront in step.
__global__ void __launch_bounds__(256) k_reseed_emit(ReseedArgs A) {
    __shared__ uint32_t lwin[(LANE_W / 2) * 256];
    const i64 n_list = (i64)A.counters[13];
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n_list; i += (i64)gridDim.x * blockDim.x) {
        // (runs beside the blocked regions' passes, which append to the same reads: only the slots the first pass filled are looked
        // at, and the hit count is bumped atomically)
        const i64 ent = A.pend_list[i];
        const i64 r = ent & ((1ll << 40) - 1);
        const int c = (int)(ent >> 40);
        SlotRec* sl = A.slots + r * A.cap;
        const u64* rec = A.packed + r * A.geo.stride;
        i64 hits_add = 0;
        int k = -1;
        for (;;) {
            for (++k; k < c; ++k) if (sl[k].sa_start & SLOT_PEND) break;
            if (k >= c) break;
            LaneQuery q;
            q.s = rec; q.off = sl[k].start; q.vlen = sl[k].end - sl[k].start;
            q.wq = ext_g(q.s, q.off);
            int r_L; i64 r_start, r_count;
            lane_search(A.I, q, 1, lwin + threadIdx.x, r_L, r_start, r_count);
            sl[k].sa_start = r_start; sl[k].count = r_count;
            hits_add += (A.opt.hits_per_smem > 0 && r_count > A.opt.hits_per_smem) ? (i64)A.opt.hits_per_smem : r_count;
        }
        atomicAdd(reinterpret_cast<unsigned long long*>(A.slot_hits + r), (unsigned long long)hits_add);
    }
}

// The searches the blocked regions wait for, as a dense batch: one lane per region.
__global__ void __launch_bounds__(256) k_reseed_search(ReseedArgs A) {
    __shared__ uint32_t lwin[(LANE_W / 2) * 256];
    i64 n_blk = (i64)A.counters[A.blk_ctr];
    if (n_blk > A.blk_cap) n_blk = A.blk_cap;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n_blk; i += (i64)gridDim.x * blockDim.x) {
        BlkRec b = A.blk[i];
        const u64* rec = A.packed + b.rid * A.geo.stride;
        const u64 lw = rec[A.geo.stride - 1];
        RegionState S;
        S.pivot = b.pivot; S.next = b.next; S.guard = b.guard; S.stage = b.stage & ~BLK_MORE;
        const LaneQuery q = region_query(rec, A.geo, (int)(lw & 0x7fffffffull), ((lw >> 31) & 1ull) != 0, S);
        lane_search(A.I, q, 2, lwin + threadIdx.x, b.r_L, b.r_start, b.r_count);
        A.blk[i].r_L = b.r_L; A.blk[i].r_start = b.r_start; A.blk[i].r_count = b.r_count;
    }
}

// Pass 2 (and 3, 4): one lane per blocked region: the search's result, then on with the table.  A region that blocks again goes to the
// next list (and the next batch search) -- except in the LAST pass, which does its searches itself.  Two regions of one read may be in
// different lanes here: SMEMs are appended with atomic counters.
template <bool LAST>
__global__ void __launch_bounds__(256) k_reseed_resume(ReseedArgs A) {
    __shared__ uint32_t win[2 * (PLCP_WIN / 4) * 256];
    __shared__ uint32_t lwin[(LANE_W / 2) * 256];
    __shared__ unsigned long long acc_total, acc_lane, out_base;
    __shared__ unsigned out_n;
    const uint8_t* __restrict__ plcp = A.I.plcp;
    const i64 n = A.I.n;
    const int msl = A.opt.min_seed_len;
    if (threadIdx.x == 0) { acc_total = 0; acc_lane = 0; }
    i64 n_blk = (i64)A.counters[A.blk_ctr];
    if (n_blk > A.blk_cap) n_blk = A.blk_cap;
    unsigned total = 0, lsearches = 0;
    for (i64 i0 = (i64)blockIdx.x * blockDim.x; i0 < n_blk; i0 += (i64)gridDim.x * blockDim.x) {
        if (threadIdx.x == 0) out_n = 0;
        __syncthreads();
        const i64 i = i0 + threadIdx.x;
        bool blocked = false;
        BlkRec nb;
        if (i < n_blk) {
            const BlkRec b = A.blk[i];
            const i64 r = b.rid;
            SlotRec* sl = A.slots + r * A.cap;
            const u64* rec = A.packed + r * A.geo.stride;
            const u64 lw = rec[A.geo.stride - 1];
            const int l_seq = (int)(lw & 0x7fffffffull);
            const bool has_n = ((lw >> 31) & 1ull) != 0;
            SmemAppender<false> ap;
            ap.sl = sl; ap.cap = A.cap; ap.cnt = A.slot_cnt + r; ap.hits = A.slot_hits + r; ap.ovf_list = A.ovf_list; ap.counters = A.counters; ap.rid = r;
            ap.hps = A.opt.hits_per_smem; ap.ns = 0; ap.hits_add = 0;
            unsigned hops = 1;                   // the search k_reseed_search did
            int k = b.k;
            const int c = (b.stage & BLK_MORE) ? (A.slot_cnt[r] > DEFER_MAX_K ? DEFER_MAX_K : A.slot_cnt[r]) : 0;   // (SMEMs appended meanwhile are not deferred ones)
            bool first = true;
            for (;;) {
                const int qbeg = sl[k].start, qend = sl[k].end;
                const i64 T0 = (sl[k].sa_start & SLOT_VAL) - qbeg;
                PlcpView V;
                V.plcp = plcp; V.win = win + threadIdx.x;
                V.F0 = plcp_win_fwd(T0, (qbeg + qend) >> 1); V.R0 = plcp_win_rev(T0, (qbeg + qend) >> 1, n);
                stage_plcp_windows(plcp, V, win + threadIdx.x);
                RegionState S;
                if (first) {
                    S.pivot = b.pivot; S.next = b.next; S.guard = b.guard; S.stage = b.stage & ~BLK_MORE;
                    region_apply<false>(S, msl, b.r_L, b.r_start, b.r_count, ap);
                    first = false;
                } else { S.pivot = (qbeg + qend) >> 1; S.next = 0; S.guard = 0; S.stage = 0; }
                while (S.stage != 3) {
                    int n_pend = 0;
                    if (!region_walk<false>(V, n, T0, qbeg, qend, l_seq, msl, S, hops, ap, n_pend)) break;
                    if (!LAST) {
                        nb.rid = r; nb.k = k; nb.pivot = S.pivot; nb.next = S.next; nb.guard = S.guard; nb.stage = S.stage | (b.stage & BLK_MORE);
                        nb.r_L = 0; nb.r_start = 0; nb.r_count = 0;
                        blocked = true;
                        break;
                    }
                    const LaneQuery q = region_query(rec, A.geo, l_seq, has_n, S);
                    int r_L; i64 r_start, r_count;
                    lane_search(A.I, q, 2, lwin + threadIdx.x, r_L, r_start, r_count);
                    ++lsearches; ++hops;
                    region_apply<false>(S, msl, r_L, r_start, r_count, ap);
                }
                if (blocked) break;
                // the read's later regions, if this record carries them
                for (++k; k < c; ++k) if ((sl[k].sa_start & (SLOT_DEFER | SLOT_POS)) == (SLOT_DEFER | SLOT_POS)) break;
                if (k >= c) break;
            }
            total += hops;
        }
        unsigned my = 0;
        if (blocked) my = atomicAdd(&out_n, 1u);
        __syncthreads();
        if (threadIdx.x == 0 && out_n) out_base = atomicAdd(&A.counters[A.blk_out_ctr], (unsigned long long)out_n);
        __syncthreads();
        if (blocked) A.blk_out[out_base + my] = nb;
        __syncthreads();
    }
    if (total) atomicAdd(&acc_total, (unsigned long long)total);
    if (lsearches) atomicAdd(&acc_lane, (unsigned long long)lsearches);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (acc_total) atomicAdd(&A.counters[1], acc_total);
        if (acc_lane) atomicAdd(&A.counters[12], acc_lane);
    }
}

}  // namespace seedk
