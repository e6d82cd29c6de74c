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
// Design (MI355X-first, round 2):
//  * ONE LANE OWNS ONE READ: its pivot logic (the reference's rounds 1-3), the interpretation of a search window and the
//    interval-level walk all run as ordinary per-lane code, so the divergent control code of a wavefront is shared by 64
//    reads (round 1 shared it among 16 and was bound by instruction issue, 80 % VALU-busy at 11 % of the HBM roofline).
//  * ONE flat loop per wavefront:  control -> request (model lookup) -> window -> resolve -> level walk -> apply.
//    A read that finishes pulls the next one inside the same loop; reads never wait for each other.
//  * The unit of memory traffic is a *window* = one 128-byte line of the key array (16 sorted 8-byte keys, the first
//    32 bases of 16 consecutive suffixes).  A lane fetching a whole line by itself is slow on this hardware (measured:
//    9.6 G lines/s against 48 G lines/s when four lanes share the line, scripts/microbench/line_patterns.hip), so the
//    window phase runs as four sub-passes: in sub-pass k the four lanes of a quad load and compare the line of the
//    quad's k-th owner (request broadcast by DPP quad_perm, no LDS), count the keys below the query with two DPP adds and
//    leave the 16 common-prefix lengths in LDS for the owner.
//  * Keys sorted + predicate monotone => a window is described by two numbers (keys below the query, keys equal to it);
//    everything else -- partition point, interval edges, the level walk -- is binary search over the 16 cached prefix
//    lengths.  Suffixes whose 32 key bases all equal the query's ("ties") are compared further in the 2-bit text, lazily:
//    only the slots a binary search actually probes cost a position fetch (5-byte array, read only here and by the hit
//    gather) and text words.
//  * Partition point, lower and upper interval edge are all "where does a monotone predicate flip?": a window that
//    does not contain the flip moves a bracket [lo, hi] and the next window gallops or bisects.
//  * The learned model is a hint: its error bounds only place the first window, any parameter file the
//    reference loads is accepted.  Records are re-laid out to 32 bytes so a lookup never straddles a line.
//  * Reads are packed once per batch by k_pack_reads (2 bits/base, forward strand, first base in the top bits of each
//    u64, plus N masks); the owner stages its read in LDS, the reverse-complement words are derived on the fly (bit
//    reversal), a 32-base query word at any offset is a funnel shift of two LDS words.
#pragma once
#include <limits.h>

#include "meme_common.h"

namespace seedk {

constexpr int BLOCK = 64;                // one wavefront per workgroup: nothing is shared between wavefronts
constexpr int MAX_READ_LEN = 500;        // LEARNED_MAX_READ_LEN (reference src/bwamem.cpp:1259)
constexpr int WIN = 16;                  // keys per window = one 128-byte line

struct SlotRec {          // search-kernel output, one per SMEM
    int32_t start, end;
    i64 sa_start;
    i64 count;
};

constexpr int N_TIERS = 4;
constexpr int TIER_CAP[N_TIERS] = {64, 512, 8192, 65536};   // SMEM slots per read; tier 0 is tunable

struct PackGeom {
    int W;        // u64 words of the forward strand (ceil(maxlen/32))
    int MW;       // u64 N-mask words
    int stride;   // 1 + W + MW (first word = read length | has-N flag << 31)
};

struct SeedArgs {
    DevIndex I;
    const u64* packed;     // [nreads_total * stride]
    i64 nreads;
    PackGeom geo;
    meme_seed_opt opt;
    SlotRec* slots;        // [nreads * cap] for this tier
    int* slot_cnt;         // [all reads]
    i64* slot_hits;
    i64* slot_loc;         // (tier << 40) | block index inside the tier's slot array
    const i64* pending;    // read ids to re-process in an overflow tier, else nullptr
    i64* ovf_list;
    int cap, tier;
    unsigned long long* counters;   // [0] ticket, [1] searches, [2] overflowed reads, [3] window loads, [4] text compares
};

// ---- read packing -------------------------------------------------------------------------------------
// Layout per read: len fw[W] nfw[MW].  A workgroup packs PACK_RB consecutive reads: their bytes are contiguous in the
// input, so they are staged in LDS with aligned, coalesced dword loads and the 2-bit words are assembled from LDS bytes.
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
            const uint8_t* p = sb + shift + (int)(ro - b0);
            if (k < g.W) {
#pragma unroll 16
                for (int j = 0; j < 32; ++j) {
                    const int i = 32 * k + j;
                    u64 c = 0;
                    if (i < len) {
                        const uint8_t bb = p[i];
                        c = bb < 4 ? bb : 0;                    // N packed as A (src/bwamem.cpp:1293-1294)
                    }
                    v = (v << 2) | c;
                }
            } else if (has_n[rr]) {
                const int m = k - g.W;
                for (int j = 0; j < 64; ++j) {
                    const int i = 64 * m + j;
                    if (i < len && p[i] >= 4) v |= 1ull << j;
                }
            }
            out[r * g.stride + 1 + k] = v;
        }
        __syncthreads();
        for (int rr = threadIdx.x; rr < nr; rr += blockDim.x) {  // length word: length | (read has an N) << 31
            const i64 r = r0 + rr;
            const int len = (int)(read_off[r + 1] - read_off[r]);
            out[r * g.stride] = (u64)(unsigned)len | (has_n[rr] ? (1ull << 31) : 0ull);
        }
    }
}

// ---- per-read state ------------------------------------------------------------------------------------
enum Pc : int {
    PC_FETCH, PC_ALLPOS_TOP, PC_ZZ_TOP, PC_ZZ_RIGHT, PC_ZZ_END, PC_AFTER_STEP1, PC_R2_LOOP, PC_R2_AFTER, PC_R3_INIT,
    PC_R3_TOP, PC_DONE, PC_LOAD, PC_EXIT
};
enum Kind : int { K_S1_RIGHT, K_ZZ_LEFT, K_ZZ_RIGHT, K_OP_MEM, K_OP_SMEM, K_R3 };
// what the next window is for: the partition point of the query (first window at the model's prediction, later ones
// gallop/bisect), or the lower / upper end of the run of suffixes sharing >= L bases with it
enum Phase : int { PH_CTRL, PH_PART, PH_EDGE_DN, PH_EDGE_UP };

#ifndef CMP_WORDS
#define CMP_WORDS 2          // text words fetched per round trip of a tie compare
#endif
// 1: the model's error bounds place the first window; 0: the line that holds the prediction
#ifndef SEED_USE_ERR
#define SEED_USE_ERR 1
#endif
#ifndef SEED_MIN_WAVES
#define SEED_MIN_WAVES 3
#endif
typedef __attribute__((address_space(3))) u64* lds_u64;
typedef __attribute__((address_space(3))) int* lds_int;
typedef __attribute__((address_space(3))) unsigned short* lds_u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u32x4* lds_u32x4;
typedef const __attribute__((address_space(1))) u64* glb_u64;
typedef const __attribute__((address_space(1))) uint8_t* glb_u8;
typedef const __attribute__((address_space(1))) uint32_t* glb_u32;
typedef const __attribute__((address_space(1))) Rmi32* glb_rmi;

__device__ __forceinline__ u64 lowmask(int k) { return k >= 64 ? ~0ull : (k <= 0 ? 0ull : ((1ull << k) - 1ull)); }

// cold per-read state in LDS, [word][lane]
enum StIdx : int { ST_BEFORE, ST_AFTER, ST_R2_K, ST_R2_NEXT, ST_ZZ /* zig-zag start | next pivot << 16 */, ST_GUARDS /* zig-zag | all-pos << 16 */, ST_N_SMEMS,
                   ST_HITS_LO, ST_HITS_HI, ST_TICKET,
                   ST_RING_SE0, ST_RING_C0, ST_RING_SE1, ST_RING_C1,   // the first two SMEMs of the current first-round pass
                   // the level walk of the search in flight (written when an edge request parks it)
                   ST_WALK /* L | lf << 10 | cache array << 16 | cache slots << 17 | cache at text end << 22 */, ST_SE_LO, ST_SE_HI, ST_EE_LO, ST_EE_HI,
                   ST_NB, ST_CB_LO, ST_CB_HI, ST_LAST_CNT_LO,
                   ST_LAST_CNT_HI, ST_LAST_S_LO, ST_LAST_S_HI, ST_WORDS };
enum StFlag : int { F_ZZ_CHECK = 1, F_ZZ_RET_ONEPOS = 2 };
enum LevelFlag : int { LF_NEED_LO = 1, LF_NEED_HI = 2, LF_HAVE_LAST = 4 };

// LDS bytes of one wavefront: the staged reads [W][64] u64, the cold state [ST_WORDS][64] int, two windows of
// 16-bit prefix lengths per read [2][64][16] (the partition window stays cached while edges are followed)
__host__ __device__ inline size_t seed_lds_bytes(int W) {
    return (size_t)W * 64 * 8 + (size_t)ST_WORDS * 64 * 4 + (size_t)2 * 64 * WIN * 2;
}

constexpr int TICKET_CHUNK = 64;     // reads a wavefront draws from the global ticket counter at a time
#ifndef TXT_WORDS
#define TXT_WORDS 3                  // text words compared per round trip of a tie compare (32 key bases + 96 per trip)
#endif
#ifndef LD_WORDS
#define LD_WORDS 10                  // words of a packed read fetched per round trip (a 150-bp record: 6 + 3 + 1)
#endif
// per-search bookkeeping word `tx`
enum TxBits : unsigned { TX_PEND = 1u, TX_FRESH = 2u, TX_SPEC = 4u, TX_W_SHIFT = 3, TX_W_MASK = 1u << 3, TX_S_SHIFT = 4,
                         TX_S_MASK = 15u << 4, TX_T0_SHIFT = 8, TX_T1_SHIFT = 16, TX_T_MASK = (31u << 8) | (31u << 16) };

// quad broadcast of lane k's value (DPP quad_perm [k,k,k,k]): one VALU instruction per dword, no LDS
template <int K>
__device__ __forceinline__ int quad_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, K * 0x55, 0xF, 0xF, true); }
template <int K>
__device__ __forceinline__ u64 quad_bcast64(u64 v) {
    const int lo = quad_bcast<K>((int)(unsigned)(v & 0xffffffffull)), hi = quad_bcast<K>((int)(unsigned)(v >> 32));
    return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ int quad_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    return v;
}

// reverse the order of the 32 two-bit groups of a word and complement them: the reverse-complement of 32 bases
__device__ __forceinline__ u64 revcomp32(u64 w) {
    u64 r = __brevll(w);
    r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
    return ~r;
}

// hash set of window lines that hold a suffix shorter than MAX_READ_LEN + 32 bases (at most ~550 slots of the whole
// array): only there can a prefix length or an order decision depend on the end of the text
__host__ __device__ __forceinline__ uint32_t special_hash(uint32_t line) { return (line * 2654435761u) >> 20; }   // 12 bits

// ---- the search kernel ---------------------------------------------------------------------------------------
// Search semantics (mem_search / right_smem_search and the _tradeoff twins, :2131-4189):
//   maxLCP = longest prefix of the query (<= vlen bases) occurring in the text;
//   mode 0: L = maxLCP.
//   mode 1: L = largest l <= maxLCP whose SA interval holds >= min_intv suffixes; [start,count) = interval
//           (:2365-2574, :2902-2942).
//   mode 2: third round (:1199-1281): walk the levels maxLCP = L0 > L1 > ... until the interval holds
//           >= min_intv suffixes or the next level is shorter than min_seed_len.
// All of it reduces to one question asked of a window of 16 consecutive slots: where does a predicate that is
// monotone over the suffix array flip from true to false?
//   PH_PART     pred = suffix < query                  flip = partition point; its two neighbours carry maxLCP
//   PH_EDGE_DN  pred = LCP(suffix, query) < L  (below the run)   flip = first slot of the level-L interval
//   PH_EDGE_UP  pred = LCP(suffix, query) >= L (above the run)   flip = one past its last slot
// [lo, hi] brackets the flip (lo: highest slot known true, hi: lowest known false; their prefix lengths are kept); a
// window that does not contain it moves the bracket and the next window gallops (step doubling) or bisects.
__global__ void __launch_bounds__(BLOCK, SEED_MIN_WAVES) k_seed(SeedArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x;
    const int t = lane & 3;                                  // position in the quad
    const int PW = A.geo.W, stride = A.geo.stride;
    const int cap = A.cap;
    const lds_u64 Qb = (lds_u64)reinterpret_cast<u64*>(smem_raw);
    const lds_int STb = (lds_int)reinterpret_cast<int*>(smem_raw + (size_t)PW * 64 * 8);
    const lds_u16 LCb = (lds_u16)reinterpret_cast<unsigned short*>(smem_raw + (size_t)PW * 64 * 8 + (size_t)ST_WORDS * 64 * 4);
    const glb_u64 keys = (glb_u64)A.I.keys;
    const glb_u8 pos5 = (glb_u8)A.I.pos5;
    const glb_u64 pac = (glb_u64)A.I.pac;
    const glb_rmi l2 = (glb_rmi)A.I.l2, l1 = (glb_rmi)A.I.l1;
    const glb_u32 special = (glb_u32)A.I.special;
    const i64 n = A.I.n;
    const int hits_per_smem = A.opt.hits_per_smem;
#define Q(k_) Qb[(k_) * 64 + lane]
#define ST(i_) STb[(i_) * 64 + lane]
#define LD64(lo_) (((i64)ST((lo_) + 1) << 32) | (u64)(unsigned)ST(lo_))
#define ST64(lo_, v_) do { const i64 v__ = (v_); ST(lo_) = (int)(unsigned)(v__ & 0xffffffffll); ST((lo_) + 1) = (int)(v__ >> 32); } while (0)
// prefix length of window slot s_ (0..15) of this lane's read in array w_; slot s lives at position 4*(s&3) + (s>>2)
#define LCP_AT(w_, s_) LCb[(((w_) * 64 + lane) << 4) + (((s_) & 3) << 2) + ((s_) >> 2)]

    // ---- per-read registers ------------------------------------------------------------------------------------
    int pc = PC_FETCH, phase = PH_CTRL;
    int pivot = 0, l_seq = 0, msl = A.opt.min_seed_len, min_intv = 1, flags = 0;
    bool has_n = false;
    int nn = 0;                                              // first two N positions of the read (see is_n)
    i64 rid = 0;
    // the request in flight
    int q_kind = 0, q_mode = 0, off = 0, vlen = 0, capc = 0;
    bool q_rc = false, q_exact = false;
    i64 base = 0, lo = -1, hi = n;
    int stepk = 0, lo_lcp = 0, hi_lcp = 0;
    int which = 0;                                           // LCP array the next window of this read is written to
    u64 wq = 0;
    // km: bits 0..15 slots of the current window whose exact prefix length is known, bits 16..31 same for the cached
    // partition window while edges are followed; lessm: order decisions of the current partition window; tx: see TxBits;
    // tk0: 32-base words of the pending text compare already known equal
    unsigned km = 0, lessm = 0, tx = 0;
    int tk0 = 1;
    unsigned acc_searches = 0, acc_windows = 0, acc_deep = 0;   // (reads re-run in an overflow tier are counted in both runs)
    // the wavefront's ticket chunk (wave-uniform values)
    unsigned long long w_next = 0, ld_ticket = 0;
    int w_remain = 0;

    // 32 query bases at offset s of the forward strand (s may reach past the read: zero words follow it in LDS)
    auto ext_fw = [&](int s) -> u64 {
        int k = s >> 5;
        const int sh = (s & 31) * 2;
        if (k > PW - 1) k = PW - 1;                          // (only for offsets beyond the read; the value is not used)
        const u64 a = Q(k), b = k + 1 < PW ? Q(k + 1) : 0ull; // zero bases follow the read
        return sh ? (a << sh) | (b >> (64 - sh)) : a;
    };
    // ... of the strand the current request searches: the reverse complement is derived from the forward words
    auto ext_q = [&](int s) -> u64 {
        if (!q_rc) return ext_fw(s);
        const int s0 = l_seq - s - 32;                       // forward bases [s0, s0 + 32) reversed and complemented
        u64 w;
        if (s0 >= 0) w = ext_fw(s0);
        else if (s0 <= -32) w = ~0ull;
        else w = (ext_fw(0) >> (2 * -s0)) | (~0ull << (64 - 2 * -s0));   // bases before the read: T, complemented to 0
        return revcomp32(w);
    };
    // Ambiguous bases: the positions of a read's first two N live in the register `nn` (n0 | n1 << 10 | count << 20,
    // count 3 = "more than two": only then the N masks of the packed record in global memory are consulted)
    auto nmask = [&](int w) -> u64 { return ((glb_u64)A.packed)[rid * stride + 1 + PW + w]; };
    auto is_n = [&](int i) -> bool {
        const int c = nn >> 20;
        if (c == 3) return (nmask(i >> 6) >> (i & 63)) & 1ull;
        return (c >= 1 && i == (nn & 1023)) || (c == 2 && i == ((nn >> 10) & 1023));
    };
    // first ambiguous base at/after `from` on the forward strand (Tokenization's *ambiguous_pos, :795-901)
    auto first_n_fw = [&](int from) -> int {
        if (!has_n) return l_seq;
        if ((nn >> 20) != 3) {
            const int n0 = nn & 1023, n1 = (nn >> 10) & 1023;
            return n0 >= from ? n0 : (((nn >> 20) == 2 && n1 >= from) ? n1 : l_seq);
        }
        int w = from >> 6;
        u64 m = nmask(w) & (~0ull << (from & 63));
        const int nw = (l_seq + 63) >> 6;
        for (;;) {
            if (m) { const int p = w * 64 + __ffsll((long long)m) - 1; return p < l_seq ? p : l_seq; }
            if (++w >= nw) return l_seq;
            m = nmask(w);
        }
    };
    // ... on the reverse-complement strand: position j there is forward position l_seq-1-j
    auto first_n_rc = [&](int from) -> int {
        if (!has_n) return l_seq;
        const int p = l_seq - 1 - from;                      // last forward position of interest
        if (p < 0) return l_seq;
        if ((nn >> 20) != 3) {
            const int n0 = nn & 1023, n1 = (nn >> 10) & 1023;
            if ((nn >> 20) == 2 && n1 <= p) return l_seq - 1 - n1;
            return n0 <= p ? l_seq - 1 - n0 : l_seq;
        }
        int w = p >> 6;
        u64 m = nmask(w) & lowmask((p & 63) + 1);
        for (;;) {
            if (m) return l_seq - 1 - (w * 64 + 63 - __clzll((long long)m));
            if (--w < 0) return l_seq;
            m = nmask(w);
        }
    };

    // exact prefix length and order of suffix array slot `slot` against the current query, from the 2-bit text
    // (compare_read_and_ref_binary*, :226-601).  L = min(capc, n - pos).  lcp < L: less = text base < read base.
    // lcp == L: less = (L < suffix length): a suffix that continues past the query sorts before it, one that ends first
    // sorts after it (as if followed by T-padding).  `k0` = 32-base words already known equal (1 after a key tie).
    auto text_compare = [&](i64 slot, int k0, int& lcp_out, bool& less_out) {
        // 5-byte records as on disk: u32 LE (pos >> 8) then u8 (pos & 0xff); fetched as the two aligned dwords around them
        const i64 bo = slot * 5;
        const glb_u32 pd = (glb_u32)(pos5 + (bo & ~3ll));
        const u64 two = (u64)pd[0] | ((u64)pd[1] << 32);
        const u64 v5 = (two >> (8 * (int)(bo & 3))) & 0xffffffffffull;
        const u64 pos = ((v5 & 0xffffffffull) << 8) | (v5 >> 32);
        int Ls = capc;
        if ((i64)pos > n - (i64)capc) Ls = (int)(n - (i64)pos);
        int ll = 32 * k0;
        bool llt = false, done = ll >= Ls;
        const i64 p0 = (i64)pos + 32 * k0;
        glb_u64 pw = pac + (p0 >> 5);
        const int sh = (int)(p0 & 31) * 2;
        for (int k = k0; !done; k += CMP_WORDS, pw += CMP_WORDS) {
            u64 w[CMP_WORDS + 1];
#pragma unroll
            for (int j = 0; j < CMP_WORDS + 1; ++j) w[j] = pw[j];
#pragma unroll
            for (int j = 0; j < CMP_WORDS; ++j) {
                if (done) break;
                const u64 wr = sh ? (w[j] << sh) | (w[j + 1] >> (64 - sh)) : w[j];
                const u64 q = ext_q(off + 32 * (k + j));
                const u64 y = wr ^ q;
                if (y) { ll += __clzll((long long)y) >> 1; llt = wr < q; done = true; }
                else { ll += 32; if (ll >= Ls) done = true; }
            }
        }
        ++acc_deep;
        if (ll >= Ls) { lcp_out = Ls; less_out = (i64)pos < n - (i64)capc; }
        else { lcp_out = ll; less_out = llt; }
    };

#ifdef SEED_PROF
    // diagnostic build: shader cycles a wavefront spends in each section of the loop body, summed into counters[5..]
    unsigned long long prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long prof_t = __builtin_amdgcn_s_memtime();
#define PROF_MARK(i_) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now__ = __builtin_amdgcn_s_memtime(); prof[i_] += now__ - prof_t; prof_t = now__; } while (0)
#else
#define PROF_MARK(i_) do { } while (0)
#endif
    for (;;) {
        // ================= control: reads without a request in flight produce the next one =========================
        bool newreq = false;
        PROF_MARK(9);
        if (phase == PH_CTRL && pc != PC_EXIT && pc != PC_FETCH && pc != PC_LOAD) {
            // One pass over the states in the order reads usually flow through them, so a read takes several hops per
            // pass; the inner loop only repeats for the rare backward hops.
            bool have = false;
#define AT(pc_) (!have && pc == (pc_))
            do {
                if (AT(PC_ZZ_TOP)) {       // zig-zag loop head (:1724-1737, :1969)
                    const int zz = ST(ST_ZZ), sp = zz & 0xffff, gw = ST(ST_GUARDS) + 1, guard = gw & 0xffff;
                    ST(ST_GUARDS) = gw;
                    if (sp >= (zz >> 16) || guard > 4 * l_seq + 16) pc = PC_ZZ_END;
                    else if ((flags & F_ZZ_CHECK) && has_n && is_n(sp)) {
                        if (l_seq - sp < msl) { pivot = l_seq; ST(ST_ZZ) = (zz & ~0xffff) | l_seq; }
                        else { ST(ST_ZZ) = zz + 1; pivot = pivot + 1; }
                    } else { q_kind = K_ZZ_LEFT; have = true; }
                }
                if (AT(PC_ZZ_RIGHT)) { q_kind = K_ZZ_RIGHT; have = true; }
                if (AT(PC_ZZ_END)) {       // set_forward_pivot(raux, next_pivot) (:1893, :2125)
                    pivot = ST(ST_ZZ) >> 16;
                    pc = (flags & F_ZZ_RET_ONEPOS) ? PC_R2_AFTER : PC_AFTER_STEP1;
                }
                if (AT(PC_AFTER_STEP1)) {  // re-seeding loop entry (:921-923)
                    const int ns = ST(ST_N_SMEMS);
                    ST(ST_AFTER) = ns;
                    if (A.opt.rounds < 2) pc = PC_ALLPOS_TOP;
                    else if (ns > cap) pc = PC_DONE;                 // re-run in the next tier (more slots per read)
                    else { ST(ST_R2_K) = ST(ST_BEFORE); pc = PC_R2_LOOP; }
                }
                if (AT(PC_R2_AFTER)) {     // (:945-946)
                    min_intv = 1;
                    pivot = ST(ST_R2_NEXT);
                    pc = PC_R2_LOOP;
                }
                if (AT(PC_R2_LOOP)) {      // (:923-947) + OnePos entry (:1917-1930)
                    int k = ST(ST_R2_K);
                    const int ke = ST(ST_AFTER);
                    int qbeg = 0, qend = 0, cnt = 0;
                    bool take = false;
                    // the first two SMEMs of this first-round pass are at hand in LDS; further ones are read back from the
                    // read's own slots (written by this lane)
                    const int kb = ST(ST_BEFORE);
                    const unsigned long long ticket = (unsigned long long)(unsigned)ST(ST_TICKET);
                    const SlotRec* mine = A.slots + (i64)ticket * cap;
                    while (k < ke) {        // SMEMs that are too short or too frequent are not re-seeded (:929-931)
                        const int d = k - kb;
                        if (d < 2) {
                            const int se = ST(d ? ST_RING_SE1 : ST_RING_SE0);
                            cnt = ST(d ? ST_RING_C1 : ST_RING_C0);
                            qbeg = se & 0xffff; qend = (int)((unsigned)se >> 16);
                        } else {
                            const u64 se = __hip_atomic_load((const u64*)&mine[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const i64 c64 = (i64)__hip_atomic_load((const u64*)&mine[k].count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            cnt = c64 > (i64)INT_MAX ? INT_MAX : (int)c64;
                            qbeg = (int)(unsigned)(se & 0xffffffffull); qend = (int)(unsigned)(se >> 32);
                        }
                        ++k;
                        if (!(qend - qbeg < A.opt.split_len || cnt > A.opt.split_width)) { take = true; break; }
                    }
                    ST(ST_R2_K) = k;
                    if (!take) pc = PC_ALLPOS_TOP;
                    else {
                        ST(ST_R2_NEXT) = pivot;
                        pivot = (qbeg + qend) >> 1;
                        min_intv = cnt + 1;
                        if (has_n && is_n(pivot)) {
                            pivot = (l_seq - pivot < msl) ? l_seq : pivot + 1;
                            pc = PC_R2_AFTER;                         // backward hop (reads with N only)
                        } else if (pivot != 0 && !(has_n && is_n(pivot - 1))) { q_kind = K_OP_MEM; have = true; }
                        else { q_kind = K_OP_SMEM; have = true; }
                    }
                }
                if (AT(PC_ALLPOS_TOP)) {   // Learned_getSMEMsAllPosOneThread loop head (:916) + step1 entry (:1691-1723)
                    const int gw = ST(ST_GUARDS) + 0x10000, guard = (int)((unsigned)gw >> 16);
                    ST(ST_GUARDS) = gw;
                    if (pivot >= l_seq || guard > 4 * l_seq + 16) pc = PC_R3_INIT;
                    else {
                        ST(ST_BEFORE) = ST(ST_N_SMEMS);
                        if (has_n && is_n(pivot)) {
                            pivot = (l_seq - pivot < msl) ? l_seq : pivot + 1;
                            pc = PC_AFTER_STEP1;                      // backward hop (reads with N only)
                        } else if (pivot != 0 && !(has_n && is_n(pivot - 1))) {
                            // zig-zag entry: the loop head's checks pass trivially (sp = pivot < next = l_seq, no N here)
                            ST(ST_ZZ) = pivot | (l_seq << 16); flags = F_ZZ_CHECK; ST(ST_GUARDS) = (gw & ~0xffff) | 1;
                            q_kind = K_ZZ_LEFT; have = true;
                        } else { q_kind = K_S1_RIGHT; have = true; }
                    }
                }
                if (AT(PC_R3_INIT)) {      // src/bwamem.cpp:1385-1394
                    if (A.opt.rounds >= 3 && A.opt.max_mem_intv > 0 && ST(ST_N_SMEMS) <= cap) {
                        min_intv = A.opt.max_mem_intv;
                        msl = A.opt.min_seed_len + 1;
                        pivot = 0;
                        pc = PC_R3_TOP;
                    } else pc = PC_DONE;
                }
                if (AT(PC_R3_TOP)) {       // Learned_bwtSeedStrategyAllPosOneThread loop head (:982-1012)
                    for (;;) {
                        if (!(pivot < l_seq - msl + 1)) { pc = PC_DONE; break; }
                        if (has_n && is_n(pivot)) { pivot = pivot + 1; continue; }
                        const int valid = first_n_fw(pivot) - pivot;
                        if (valid < msl) { pivot = pivot + valid; continue; }
                        q_kind = K_R3; have = true;
                        break;
                    }
                }
                if (AT(PC_DONE)) {         // publish the read's SMEM count / hit count; its slots are already written
                    const unsigned long long ticket = (unsigned long long)(unsigned)ST(ST_TICKET);
                    const int ns = ST(ST_N_SMEMS);
                    const bool ovf = ns > cap;
                    A.slot_cnt[rid] = ovf ? 0 : ns;
                    A.slot_hits[rid] = ovf ? 0 : LD64(ST_HITS_LO);
                    A.slot_loc[rid] = ((i64)A.tier << 40) | (i64)ticket;
                    if (ovf) A.ovf_list[atomicAdd(&A.counters[2], 1ull)] = rid;
                    pc = PC_FETCH;
                }
            } while (!have && pc != PC_EXIT && pc != PC_FETCH);
#undef AT
            newreq = have;
        }
        PROF_MARK(0);
        // hand-out of new reads (reads differ 3x in cost: dynamic, no static deal).  Every lane of the wavefront passes
        // here together: the wavefront keeps a chunk of TICKET_CHUNK tickets (w_next, w_remain are wave-uniform) and
        // refills it with ONE global atomic (one atomic per read on a single address caps the kernel at ~30 M reads/s).
        // The read itself is fetched in memory round 1 below, next to the other lanes' model records.
        {
            const bool want = phase == PH_CTRL && pc == PC_FETCH;
            const unsigned long long mw = __ballot(want);
            if (mw) {
                const int k = __popcll(mw);
                const int rank = __popcll(mw & ((1ull << lane) - 1ull));
                unsigned long long ticket = w_next + (unsigned)rank;
                if (k > w_remain) {
                    unsigned long long nb = 0;
                    const int first = __ffsll((long long)mw) - 1;
                    if (lane == first) nb = atomicAdd(&A.counters[0], (unsigned long long)TICKET_CHUNK);
                    nb = ((unsigned long long)(unsigned)__shfl((int)(nb >> 32), first) << 32) | (unsigned)__shfl((int)(nb & 0xffffffffull), first);
                    if (rank >= w_remain) ticket = nb + (unsigned)(rank - w_remain);
                    w_next = nb + (unsigned)(k - w_remain);
                    w_remain = TICKET_CHUNK - (k - w_remain);
                } else { w_next += (unsigned)k; w_remain -= k; }
                if (want) {
                    if (ticket >= (unsigned long long)A.nreads) pc = PC_EXIT;
                    else { ld_ticket = ticket; pc = PC_LOAD; }
                }
            }
        }
        if (!__ballot(pc != PC_EXIT)) break;                 // every read of the batch has been handed out and finished
        PROF_MARK(1);

        // ================= memory round 1: model record (new requests) | position of a tied slot (text compares) =========
        // Every lane does at most TWO dependent memory rounds per iteration -- (model, keys) for a new request, (-, keys)
        // for a relocated or edge window, (position, text words) for a tie compare -- so an iteration costs two round
        // trips whatever the slowest read needs; a read that needs more takes more iterations, the others do not wait.
        const bool live = pc != PC_EXIT;
        const bool do_txt = live && (tx & TX_PEND) != 0;
        u64 m_idx = 0, key = 0;
        double r_icpt = 0.0, r_slope = 0.0;
        u64 r_err = 0;
        if (newreq) {
            // ---- the request: query = bases [off, off+vlen) of one strand; first window at the model's prediction
            q_rc = q_kind == K_ZZ_LEFT;
            off = q_rc ? l_seq - 1 - pivot : pivot;
            vlen = (q_rc ? first_n_rc(off) : first_n_fw(off)) - off;
            q_exact = q_kind == K_S1_RIGHT || q_kind == K_ZZ_RIGHT || q_kind == K_OP_SMEM;
            q_mode = q_kind == K_R3 ? 2 : ((q_exact || min_intv != 1) ? 1 : 0);
            ++acc_searches;
            wq = ext_q(off);                                     // first 32 bases of the query: every window compares against it
            key = wq;
            if (vlen < 32) key |= (~0ull) >> (2 * vlen);          // T-pad short queries like Tokenization (:813-817)
            m_idx = A.I.shift >= 64 ? 0ull : key >> A.I.shift;
            r_icpt = l2[m_idx].icpt; r_slope = l2[m_idx].slope; r_err = l2[m_idx].err;
        }
        // the slot whose text compare is pending: window array tw (base twb), slot ts, tk0 words known equal
        const int tw = (int)((tx >> TX_W_SHIFT) & 1u), ts = (int)((tx >> TX_S_SHIFT) & 15u);
        i64 twb = 0;
        u64 t_two = 0;
        if (do_txt) {
            twb = (tw == which && phase != PH_CTRL) ? base : LD64(ST_CB_LO);
            if (phase == PH_PART) twb = base;
            // 5-byte records as on disk: u32 LE (pos >> 8) then u8 (pos & 0xff); fetched as the two aligned dwords around them
            const i64 bo = (twb + ts) * 5;
            const glb_u32 pd = (glb_u32)(pos5 + (bo & ~3ll));
            t_two = (u64)pd[0] | ((u64)pd[1] << 32);
            t_two = (t_two >> (8 * (int)(bo & 3))) & 0xffffffffffull;
        }
        // a lane that drew a ticket fetches its packed read now (LD_WORDS words per round; a 150-bp record is 10 words) and
        // runs its control pass in the next iteration
        const bool do_load = live && pc == PC_LOAD;
        u64 ld_w[LD_WORDS];
        glb_u64 ld_src = (glb_u64)A.packed;
        if (do_load) {
            rid = A.pending ? A.pending[ld_ticket] : (i64)ld_ticket;
            ld_src = (glb_u64)A.packed + rid * stride;
#pragma unroll
            for (int j = 0; j < LD_WORDS; ++j) ld_w[j] = ld_src[j < stride ? j : stride - 1];
        }
        if (do_load) {
            // stage the forward strand in LDS, note the first two N positions (only reads that have one), clear the cold state
            nn = 0;
            const u64 lenw = ld_w[0];
            const bool any_n = ((lenw >> 31) & 1ull) != 0;
            for (int k0 = 0; k0 < stride; k0 += LD_WORDS) {
                if (k0) {                                        // longer records: further rounds of LD_WORDS words
#pragma unroll
                    for (int j = 0; j < LD_WORDS; ++j) ld_w[j] = ld_src[k0 + j < stride ? k0 + j : stride - 1];
                }
#pragma unroll
                for (int j = 0; j < LD_WORDS; ++j) {
                    const int k2 = k0 + j - 1;                   // word k2 of the record's data (the length word is word -1)
                    if (k2 >= 0 && k2 < PW) Q(k2) = ld_w[j];
                }
                if (any_n) {
#pragma unroll
                    for (int j = 0; j < LD_WORDS; ++j) {
                        const int k2 = k0 + j - 1;
                        if (k2 >= PW && k2 < stride - 1) {
                            u64 mw2 = ld_w[j];
                            for (int it = 0; it < 3 && mw2; ++it) {
                                const int pn = 64 * (k2 - PW) + __ffsll((long long)mw2) - 1;
                                const int c = nn >> 20;
                                if (c == 0) nn = pn | (1 << 20);
                                else if (c == 1) nn = (nn & 1023) | (pn << 10) | (2 << 20);
                                else nn = (nn & 0xfffff) | (3 << 20);
                                mw2 &= mw2 - 1;
                            }
                        }
                    }
                }
            }
            l_seq = (int)(lenw & 0x7fffffffull);               // k_pack_reads: length | has-N flag << 31
            has_n = ((lenw >> 31) & 1ull) != 0;
            for (int k2 = 0; k2 < ST_WORDS; ++k2) ST(k2) = 0;
            if (l_seq <= 0 || l_seq > MAX_READ_LEN) {
                // empty read (longer ones are rejected by the host before the launch): no seeds; draws the next ticket
                A.slot_cnt[rid] = 0; A.slot_hits[rid] = 0; A.slot_loc[rid] = 0;
                pc = PC_FETCH;
            } else {
                ST(ST_TICKET) = (int)(unsigned)ld_ticket;          // (a launch hands out fewer than 2^32 tickets)
                pivot = 0; msl = A.opt.min_seed_len; min_intv = 1; flags = 0;
                pc = PC_ALLPOS_TOP;
            }
        }

        if (newreq) {
            // learned_index_lookup (:186-210): same arithmetic (FP64 FMA + clamp, partial third layer), used as a hint
            const double x = (double)key;
            double f = fma(r_slope, x, r_icpt);
            u64 err = r_err;
            if (err >> 63) {
                const u64 ps = (err >> 32) & 0x7fffffffull;
                const double pn = (double)(err & 0xffffffffull) - 1.0;
                const double c = f < 0.0 ? 0.0 : (f > pn ? pn : f);
                u64 j = ps + (u64)c;
                if (j >= (u64)A.I.n_l1) j = A.I.n_l1 > 0 ? (u64)A.I.n_l1 - 1 : 0;   // a malformed table cannot send the load astray
                f = fma(l1[j].slope, x, l1[j].icpt);
                err = l1[j].err;
            }
            i64 pos = f < 0.0 ? 0 : (f > (double)n - 1.0 ? n - 1 : (i64)f);
#if SEED_USE_ERR
            // the partition point lies in [pos - below, pos + above]: aim at the middle of that span
            const i64 below = (i64)((err >> 32) & 0x3fffffffull), above = (i64)(err & 0x7fffffffull);
            pos += (above - below) / 2;
            if (pos < 0) pos = 0;
            if (pos > n - 1) pos = n - 1;
#endif
            base = pos & ~(i64)(WIN - 1);
            lo = -1; hi = n; stepk = 0; capc = vlen; which = 0;
            phase = PH_PART;
            tx = TX_FRESH;
        }

        PROF_MARK(2);
        // ================= memory round 2: window keys (four sub-passes, the quad works for its k-th owner) | text words ======
        const bool fresh = live && phase != PH_CTRL && (tx & TX_FRESH) != 0;
        // does this window's line hold one of the few suffixes near the end of the text?  (L1-resident 16 KB table)
        uint32_t sp_h = 0, sp_v = 0;
        if (fresh) { sp_h = special_hash((uint32_t)(base >> 4)); sp_v = special[sp_h]; }
        u64 t_pos = 0, t_w[TXT_WORDS + 1];
        int t_Ls = 0, t_sh = 0;
        if (do_txt) {
            t_pos = ((t_two & 0xffffffffull) << 8) | (t_two >> 32);
            t_Ls = capc;
            if ((i64)t_pos > n - (i64)capc) t_Ls = (int)(n - (i64)t_pos);
            const i64 p0 = (i64)t_pos + 32 * (i64)tk0;
            const glb_u64 pw = pac + (p0 >> 5);
            t_sh = (int)(p0 & 31) * 2;
#pragma unroll
            for (int j = 0; j < TXT_WORDS + 1; ++j) t_w[j] = pw[j];
        }
        int cnt_lt = 0, cnt_tie = 0;
        {
            // request word broadcast to the quad: window line (base >> 4) << 2 | LCP array << 1 | active
            const int breq = fresh ? (int)(((unsigned)(base >> 4) << 2) | ((unsigned)which << 1) | 1u) : 0;
            const int cl = capc > 32 ? 33 : capc;                // lanes only need to know whether capc reaches past the key
            u64 kk[4][4];
            int rq[4];
#define SUB_LOAD(K_)                                                                                   \
            rq[K_] = quad_bcast<K_>(breq);                                                                           \
            if (rq[K_] & 1) {                                                                                        \
                const glb_u64 kp = keys + ((i64)((unsigned)rq[K_] >> 2) << 4) + t;                                   \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) kk[K_][e] = kp[4 * e];                              \
            }
            SUB_LOAD(0) SUB_LOAD(1) SUB_LOAD(2) SUB_LOAD(3)
#undef SUB_LOAD
#define SUB_CMP(K_)                                                                                    \
            {                                                                                                        \
                int c = 0;                                                                                           \
                const u64 rw = quad_bcast64<K_>(wq);                                                                 \
                const int rc = quad_bcast<K_>(cl);                                                                   \
                if (rq[K_] & 1) {                                                                                    \
                    unsigned short lv[4];                                                                            \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                               \
                        const u64 xx = kk[K_][e] ^ rw;                                                               \
                        int l = xx ? (__clzll((long long)xx) >> 1) : 32;                                             \
                        const bool full = l >= rc;            /* the whole query matched inside the key */           \
                        const bool tie = !full && xx == 0;   /* 32 bases equal and the query goes on */             \
                        if (full) l = rc;                                                                            \
                        c += tie ? 32 : ((full || kk[K_][e] < rw) ? 1 : 0);                                          \
                        lv[e] = (unsigned short)l;                                                                   \
                    }                                                                                                \
                    const int owner = (lane & ~3) + K_;                                                              \
                    const int wsel = (rq[K_] >> 1) & 1;                                                              \
                    *(__attribute__((address_space(3))) u64*)&LCb[(((wsel * 64) + owner) << 4) + (t << 2)] =         \
                        (u64)lv[0] | ((u64)lv[1] << 16) | ((u64)lv[2] << 32) | ((u64)lv[3] << 48);                   \
                }                                                                                                    \
                c = quad_sum(c);                                                                                     \
                if (t == K_) { cnt_lt = c & 31; cnt_tie = c >> 5; }                                                  \
            }
            // (16 keys per window: the "below" count takes 5 bits of the quad sum, the "equal" count the bits above)
            SUB_CMP(0) SUB_CMP(1) SUB_CMP(2) SUB_CMP(3)
#undef SUB_CMP
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        PROF_MARK(3);
        bool need_eval = false;
        if (do_txt) {
            // ---- finish the text compare (compare_read_and_ref_binary*, :226-601).  L = min(capc, n - pos).
            // lcp < L: less = text base < read base.  lcp == L: less = (L < suffix length): a suffix that continues past the
            // query sorts before it, one that ends first sorts after it (as if followed by T-padding).
            int ll = 32 * tk0;
            bool llt = false, done = ll >= t_Ls;
#pragma unroll
            for (int j = 0; j < TXT_WORDS; ++j) {
                if (!done) {
                    const u64 wr = t_sh ? (t_w[j] << t_sh) | (t_w[j + 1] >> (64 - t_sh)) : t_w[j];
                    const u64 q = ext_q(off + 32 * (tk0 + j));
                    const u64 y = wr ^ q;
                    if (y) { ll += __clzll((long long)y) >> 1; llt = wr < q; done = true; }
                    else { ll += 32; if (ll >= t_Ls) done = true; }
                }
            }
            if (!done) tk0 += TXT_WORDS;                         // a long query: the compare goes on in the next iteration
            else {
                ++acc_deep;
                int lc; bool ls;
                if (ll >= t_Ls) { lc = t_Ls; ls = (i64)t_pos < n - (i64)capc; }
                else { lc = ll; ls = llt; }
                LCP_AT(tw, ts) = (unsigned short)lc;
                const bool cur = phase == PH_PART || tw == which;
                km |= (1u << ts) << (cur ? 0 : 16);
                if (cur && ls) lessm |= 1u << ts;
                tx &= ~(unsigned)TX_PEND;
                need_eval = true;
            }
        }
        PROF_MARK(4);
        const int valid = (n - base) < WIN ? (int)(n - base) : WIN;
        if (fresh) {
            ++acc_windows;
            tx &= ~(unsigned)TX_FRESH;
            // ---- this window in numbers: slots [0, t0) sort below the query by their keys, [t0, t1) tie with it
            // (their order and prefix length need the text), [t1, valid) sort above.  km bits 0..15: slots whose exact
            // prefix length is in the LCP array (an unknown slot is a key tie: >= 32 bases equal).
            int t0 = cnt_lt, t1 = cnt_lt + cnt_tie;
            if (t0 > valid) t0 = valid;                          // (padding keys beyond the array compare as all-T)
            if (t1 > valid) t1 = valid;
            km = (km & 0xffff0000u) | ((unsigned)(lowmask(WIN) & ~(lowmask(t1) & ~lowmask(t0))) & 0xffffu);
            lessm = (unsigned)lowmask(t0);                       // order decisions known so far (bit set = suffix < query)
            bool spec = false;
            for (;;) {
                if (sp_v == (uint32_t)(base >> 4) + 1u) { spec = true; break; }
                if (sp_v == 0u) break;
                sp_h = (sp_h + 1u) & (uint32_t)(SPECIAL_SLOTS - 1);
                sp_v = special[sp_h];
            }
            if (spec) {
                // a suffix near the end of the text lives here (a few windows of the whole array): decide every slot from
                // the text itself, eagerly, and evaluate the window slot by slot like the reference's compare would
                lessm = 0;
                for (int s2 = 0; s2 < valid; ++s2) {
                    int lc; bool ls;
                    text_compare(base + s2, 0, lc, ls);
                    LCP_AT(which, s2) = (unsigned short)lc;
                    if (ls) lessm |= 1u << s2;
                }
                km |= 0xffffu;
                t0 = t1 = 0;
            }
            tx = (tx & ~(unsigned)(TX_T_MASK | TX_SPEC)) | ((unsigned)t0 << TX_T0_SHIFT) | ((unsigned)t1 << TX_T1_SHIFT) | (spec ? TX_SPEC : 0u);
            need_eval = true;
        }

        PROF_MARK(5);
        // ================= evaluate: what does the window (or the cached partition window) say now? ======================
        // Runs to the end and commits, or stops at the first slot whose prefix length is still unknown (a key tie): that slot
        // goes to the text compare of the next iteration and the evaluation simply starts over -- it is a pure function of
        // the cached lengths, which only ever become more complete.
        bool finished = false;
        int r_L = 0;
        i64 r_start = 0, r_count = 1;
        bool r_emit = false;
        if (need_eval) {
            const bool spec = (tx & TX_SPEC) != 0;
            const int wcur = which;
            int need_w = -1, need_s = 0;
            const unsigned known = km & 0xffffu, vmask = (unsigned)lowmask(valid);
            // The 16 prefix lengths of a window live in LDS in position order (position 4t+e = slot 4e+t); one 32-byte read
            // brings them into registers, and every question below is answered with 16-bit masks over the slots.
            unsigned dcur[8];
            {
                const lds_u32x4 lp4 = (lds_u32x4)&LCb[((wcur * 64 + lane) << 4)];
                const u32x4 x0 = lp4[0], x1 = lp4[1];
                dcur[0] = x0.x; dcur[1] = x0.y; dcur[2] = x0.z; dcur[3] = x0.w; dcur[4] = x1.x; dcur[5] = x1.y; dcur[6] = x1.z; dcur[7] = x1.w;
            }
            // slots whose stored prefix length is >= L (an unknown slot stores 32: a key tie shares at least 32 bases)
            auto ge16 = [&](const unsigned (&d)[8], int L) -> unsigned {
                unsigned m = 0;
                const unsigned uL = (unsigned)L;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    constexpr int dummy = 0; (void)dummy;
                    const int p0 = 2 * i, p1 = 2 * i + 1;
                    const int s0 = 4 * (p0 & 3) + (p0 >> 2), s1 = 4 * (p1 & 3) + (p1 >> 2);
                    m |= ((d[i] & 0xffffu) >= uL ? (1u << s0) : 0u) | ((d[i] >> 16) >= uL ? (1u << s1) : 0u);
                }
                return m;
            };
            auto lcp_at = [&](int w, int s) -> int { return (int)LCP_AT(w, s); };
            // the unknown slot of a range to look up first: the one nearest to the middle (the range halves every time)
            auto pick = [&](unsigned um, int a, int b) -> int {
                const int mid = (a + b - 1) >> 1;
                const unsigned up = um >> mid;
                return up ? mid + (__ffs((int)up) - 1) : 31 - __clz((int)(um & (unsigned)lowmask(mid)));
            };

            // ---- number of slots of the window for which the phase's predicate holds (they come first) -----------------
            const int wlo = (lo - base + 1 > valid) ? valid : (lo - base + 1 < 0 ? 0 : (int)(lo - base + 1));   // slots < wlo known true
            const int whi = (hi - base > valid) ? valid : (hi - base < 0 ? 0 : (int)(hi - base));              // slots >= whi known false
            const unsigned brk_lo = (unsigned)lowmask(wlo), brk_hi = (unsigned)lowmask(whi);
            unsigned tru, unk;                                   // slots known true; slots whose answer needs the text
            if (phase == PH_PART) { tru = lessm; unk = ~known & vmask; }
            else {
                const unsigned g = ge16(dcur, capc);
                const bool deep = capc > 32 && !spec;            // an unknown slot only matters for levels beyond the key
                unk = deep ? (~known & vmask) : 0u;
                tru = phase == PH_EDGE_DN ? (~g & vmask & ~unk) : (g & ~unk);
            }
            int P;
            if (spec) {
                // windows at the end of the text: every slot is known; count like the reference's slot-by-slot compare would
                P = wlo + __popc(tru & ~brk_lo & brk_hi);
            } else {
                const unsigned T = (tru | brk_lo) & brk_hi, U = unk & ~brk_lo & brk_hi;
                const int p_lo = __ffs((int)~T) - 1, p_hi = __ffs((int)~(T | U)) - 1;     // (bit 16 and above are zero: <= 16)
                if (p_lo != p_hi) { need_w = wcur; need_s = pick(U, p_lo, p_hi); }
                P = p_lo;
            }

            // is the flip pinned down?  slot f-1 must be known true and slot f known false
            const i64 f = base + P;
            const bool lo_ok = P > 0 || base == 0 || lo == base - 1;
            const bool hi_ok = P < valid || base + valid >= n || hi == base + valid;
            const bool found = lo_ok && hi_ok;
            // exact prefix length of a slot of the current window (noting it as needed when it is still unknown)
            auto cur_lcp = [&](int s) -> int {
                if (!((known >> s) & 1u) && need_w < 0) { need_w = wcur; need_s = s; }
                return lcp_at(wcur, s);
            };

            // ---- resolve (into temporaries: nothing is committed before the evaluation is known to be complete) ----------
            bool go_level = false, c_finished = false, c_park = false;
            int L = 0, nb_lo = 0, nb_hi = 0, lf = 0;
            i64 s_edge = 0, e_edge = 0, cb = base;
            unsigned cknown = known;                             // the cached partition window's bookkeeping
            int cwhich = wcur, cvalid = valid;
            bool cspec = spec, same_cache = true;
            i64 n_lo = lo, n_hi = hi, n_base = base;
            int n_lo_lcp = lo_lcp, n_hi_lcp = hi_lcp, n_stepk = stepk, n_phase = phase;
            if (found) {
                // prefix lengths of the two slots around the flip (in the window, or remembered with the bracket)
                const int lm = P > 0 ? cur_lcp(P - 1) : (f > 0 ? lo_lcp : -1);
                const int lp = P < valid ? cur_lcp(P) : (f < n ? hi_lcp : -1);
                if (phase == PH_PART) {
                    // slots below f sort before the query; the longest match is at one of the two boundary neighbours
                    L = lm >= lp ? lm : lp;
                    const i64 c = lm >= lp ? f - 1 : f;
                    r_L = L; r_start = c; r_count = 1;
                    if (q_mode == 0 || (q_mode == 2 && L < msl)) c_finished = true;       // (:1204-1208)
                    else {
                        s_edge = e_edge = c;
                        lf = LF_NEED_LO | LF_NEED_HI;
                        go_level = true;
                    }
                } else {
                    { const int cc = ST(ST_WALK); L = cc & 1023; lf = (cc >> 10) & 7; cwhich = (cc >> 16) & 1; cvalid = (cc >> 17) & 31; cspec = (cc >> 22) & 1; }
                    { const int nb = ST(ST_NB); nb_lo = nb & 0xffff; nb_hi = (int)((unsigned)nb >> 16); }
                    s_edge = LD64(ST_SE_LO); e_edge = LD64(ST_EE_LO); cb = LD64(ST_CB_LO);
                    cknown = km >> 16; same_cache = false;
                    if (phase == PH_EDGE_DN) { s_edge = f; nb_lo = f > 0 ? lm : 0; lf &= ~LF_NEED_LO; }
                    else { e_edge = f - 1; nb_hi = f < n ? lp : 0; lf &= ~LF_NEED_HI; }
                    go_level = true;
                }
            } else {
                // move the bracket, remembering the prefix length of its new end
                if (P == valid) { n_lo = base + valid - 1; n_lo_lcp = cur_lcp(valid - 1); }
                else { n_hi = base; n_hi_lcp = cur_lcp(0); }
                bool stop = false;
                if (phase != PH_PART && !q_exact) {
                    // the interval is not emitted at the level where the walk stops (left extensions, third round): it is
                    // enough to know that it reached min_intv suffixes
                    s_edge = LD64(ST_SE_LO); e_edge = LD64(ST_EE_LO);
                    if (phase == PH_EDGE_DN ? (e_edge - n_hi + 1 >= (i64)min_intv) : (n_lo - s_edge + 1 >= (i64)min_intv)) {
                        cb = LD64(ST_CB_LO);
                        { const int cc = ST(ST_WALK); L = cc & 1023; lf = (cc >> 10) & 7; cwhich = (cc >> 16) & 1; cvalid = (cc >> 17) & 31; cspec = (cc >> 22) & 1; }
                        { const int nb = ST(ST_NB); nb_lo = nb & 0xffff; nb_hi = (int)((unsigned)nb >> 16); }
                        cknown = km >> 16; same_cache = false;
                        if (phase == PH_EDGE_DN) { s_edge = n_hi; nb_lo = L; lf &= ~LF_NEED_LO; }
                        else { e_edge = n_lo; nb_hi = L; lf &= ~LF_NEED_HI; }
                        stop = go_level = true;
                    }
                }
                if (!stop) {
                    i64 tgt;
                    if (n_lo < 0) { tgt = n_hi - 1 - (((i64)WIN << stepk) - WIN); n_stepk = stepk + 1; if (tgt < 0) tgt = 0; }              // gallop down
                    else if (n_hi >= n) { tgt = n_lo + 1 + (((i64)WIN << stepk) - WIN); n_stepk = stepk + 1; if (tgt > n - 1) tgt = n - 1; }   // gallop up
                    else tgt = n_lo + (n_hi - n_lo) / 2;                                                                       // bisect (hi - lo >= 2 here)
                    n_base = tgt & ~(i64)(WIN - 1);
                }
            }
            if (go_level && need_w < 0) {
                // Walk the levels L0 > L1 > ... on the cached partition window [cb, cb + cvalid): the run of slots sharing
                // >= L bases with the query is contiguous around the partition point, so an edge is the nearest slot on that
                // side that does not reach the level -- one mask, one bit scan.  An edge that leaves the cached window
                // becomes an edge request.
                unsigned dc[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) dc[i] = dcur[i];
                if (!same_cache) {
                    const lds_u32x4 lp4 = (lds_u32x4)&LCb[((cwhich * 64 + lane) << 4)];
                    const u32x4 x0 = lp4[0], x1 = lp4[1];
                    dc[0] = x0.x; dc[1] = x0.y; dc[2] = x0.z; dc[3] = x0.w; dc[4] = x1.x; dc[5] = x1.y; dc[6] = x1.z; dc[7] = x1.w;
                }
                const unsigned cvm = (unsigned)lowmask(cvalid), cunk = ~cknown & cvm;
                for (;;) {
                    if (lf & (LF_NEED_LO | LF_NEED_HI)) {
                        const unsigned gp = ge16(dc, L);                     // pessimistic: an unknown tie does not reach L > 32
                        const unsigned go = L > 32 ? (gp | cunk) : gp;       // optimistic
                        if (lf & LF_NEED_LO) {
                            const i64 klo = s_edge - cb;              // cached slots [0, klo) lie below the current edge
                            if (klo > 0 && klo <= cvalid) {
                                const unsigned below = (unsigned)lowmask((int)klo);
                                const unsigned mp = ~gp & below, mo = ~go & below;
                                const int zp = mp ? 32 - __clz((int)mp) : 0, zo = mo ? 32 - __clz((int)mo) : 0;
                                if (zp != zo) { need_w = cwhich; need_s = pick(cunk & below, zo, zp); break; }
                                if (zp > 0) { s_edge = cb + zp; nb_lo = lcp_at(cwhich, zp - 1); lf &= ~LF_NEED_LO; }
                                else { s_edge = cb; if (cb == 0) { nb_lo = 0; lf &= ~LF_NEED_LO; } }
                            } else if (s_edge == 0) { nb_lo = 0; lf &= ~LF_NEED_LO; }
                        }
                        if (lf & LF_NEED_HI) {
                            const i64 khi = e_edge - cb;              // cached slots (khi, cvalid) lie above it
                            if (khi >= -1 && khi < cvalid - 1) {
                                const unsigned above = cvm & ~(unsigned)lowmask((int)khi + 1);
                                const unsigned mp = ~gp & above, mo = ~go & above;
                                const int zp = mp ? __ffs((int)mp) - 1 : cvalid, zo = mo ? __ffs((int)mo) - 1 : cvalid;
                                if (zp != zo) { need_w = cwhich; need_s = pick(cunk & above, zp, zo); break; }
                                if (zp < cvalid) { e_edge = cb + zp - 1; nb_hi = lcp_at(cwhich, zp); lf &= ~LF_NEED_HI; }
                                else { e_edge = cb + cvalid - 1; if (cb + cvalid >= n) { nb_hi = 0; lf &= ~LF_NEED_HI; } }
                            } else if (e_edge == n - 1) { nb_hi = 0; lf &= ~LF_NEED_HI; }
                        }
                        if (lf & LF_NEED_LO) {                            // the run leaves the cached window: edge request
                            n_phase = PH_EDGE_DN; n_lo = -1; n_hi = s_edge; n_hi_lcp = L; n_stepk = 1;
                            n_base = (s_edge - 1) & ~(i64)(WIN - 1);
                            c_park = true;
                            break;
                        }
                        if (lf & LF_NEED_HI) {
                            n_phase = PH_EDGE_UP; n_lo = e_edge; n_lo_lcp = L; n_hi = n; n_stepk = 1;
                            n_base = (e_edge + 1) & ~(i64)(WIN - 1);
                            c_park = true;
                            break;
                        }
                    }
                    const i64 cnt = e_edge - s_edge + 1;
                    const int nxt = nb_lo > nb_hi ? nb_lo : nb_hi;
                    if (q_mode == 1) {                                  // (:2568-2573, :2936-2940)
                        if (cnt >= (i64)min_intv) { r_L = L; r_start = s_edge; r_count = cnt; c_finished = true; break; }
                    } else {
                        if (cnt >= (i64)min_intv) {                     // :1243-1251
                            if (lf & LF_HAVE_LAST) { r_count = LD64(ST_LAST_CNT_LO); r_start = LD64(ST_LAST_S_LO); }
                            else { r_count = cnt; r_start = s_edge; }
                            r_L = L + 1;
                            c_finished = true;
                            break;
                        }
                        if (nxt < msl) { r_L = msl; r_start = s_edge; r_count = cnt; c_finished = true; break; }   // :1252-1258
                        lf |= LF_HAVE_LAST;
                        ST64(ST_LAST_CNT_LO, cnt);                      // (rewritten identically if the evaluation starts over)
                        ST64(ST_LAST_S_LO, s_edge);
                    }
                    L = nxt;
                    lf = (lf & LF_HAVE_LAST) | ((nb_lo >= L && s_edge > 0) ? LF_NEED_LO : 0) | ((nb_hi >= L && e_edge < n - 1) ? LF_NEED_HI : 0);
                }
            }
            if (need_w >= 0) {
                // a tie has to be looked up in the text first: next iteration (position, then text words), then start over
                tx = (tx & ~(unsigned)(TX_W_MASK | TX_S_MASK)) | TX_PEND | ((unsigned)need_w << TX_W_SHIFT) | ((unsigned)need_s << TX_S_SHIFT);
                tk0 = 1;
            } else if (c_finished) {
                finished = true;
                if (q_mode == 2 && go_level) {                     // (a search that ended in the level walk)
                    r_emit = r_count < (i64)min_intv;              // :1265
                    if (r_L < msl) r_L = msl;
                }
            } else {
                lo = n_lo; hi = n_hi; lo_lcp = n_lo_lcp; hi_lcp = n_hi_lcp; stepk = n_stepk; base = n_base;
                tx |= TX_FRESH;                                   // the next iteration loads the window at `base`
                if (c_park) {
                    // park the walk; the edge windows go to the other LCP array, the cache stays.  The cache's bookkeeping
                    // mask lives in km bits 16..31 while edges are followed.
                    phase = n_phase;
                    capc = L;
                    ST(ST_NB) = (nb_lo & 0xffff) | (nb_hi << 16);
                    ST64(ST_SE_LO, s_edge); ST64(ST_EE_LO, e_edge); ST64(ST_CB_LO, cb);
                    ST(ST_WALK) = L | (lf << 10) | (cwhich << 16) | (cvalid << 17) | ((cspec ? 1 : 0) << 22);
                    km = (km & 0xffffu) | (cknown << 16);
                    which = cwhich ^ 1;
                }
            }
        }
        PROF_MARK(6);
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
                pc = ((ST(ST_ZZ) >> 16) - pivot < msl) ? PC_ZZ_END : PC_ZZ_RIGHT;
                break;
            case K_OP_MEM:            // (:1967-1969)
                ST(ST_ZZ) = pivot | ((pivot + r_L) << 16); flags = F_ZZ_RET_ONEPOS; ST(ST_GUARDS) = ST(ST_GUARDS) & ~0xffff;
                pc = PC_ZZ_TOP;
                break;
            default:                  // K_R3 (:1204-1208, :1265-1281)
                emit = r_emit; e_end = pivot + r_L;
                break;
            }
            if (emit) {               // kv_push of mem_tl + hits (:2639-2657, :1266-1277)
                const int ns = ST(ST_N_SMEMS);
                if (ns < cap) {
                    const unsigned long long ticket = (unsigned long long)(unsigned)ST(ST_TICKET);
                    SlotRec sr;
                    sr.start = e_start; sr.end = e_end; sr.sa_start = r_start; sr.count = r_count;
                    A.slots[(i64)ticket * cap + ns] = sr;
                }
                ST(ST_N_SMEMS) = ns + 1;
                {
                    const int d = ns - ST(ST_BEFORE);
                    const int cc = r_count > (i64)INT_MAX ? INT_MAX : (int)r_count;
                    if (d == 0) { ST(ST_RING_SE0) = e_start | (e_end << 16); ST(ST_RING_C0) = cc; }
                    else if (d == 1) { ST(ST_RING_SE1) = e_start | (e_end << 16); ST(ST_RING_C1) = cc; }
                }
                i64 h = r_count;
                if (hits_per_smem > 0 && h > hits_per_smem) h = hits_per_smem;
                h += LD64(ST_HITS_LO);
                ST64(ST_HITS_LO, h);
            }
            switch (q_kind) {
            case K_S1_RIGHT: pivot = pivot + r_L; pc = PC_AFTER_STEP1; break;
            case K_ZZ_RIGHT: pivot = pivot + r_L; ST(ST_ZZ) = (ST(ST_ZZ) & ~0xffff) | pivot; pc = PC_ZZ_TOP; break;
            case K_OP_SMEM: pivot = pivot + r_L; pc = PC_R2_AFTER; break;
            case K_R3: pivot = pivot + (r_L < msl ? msl : r_L); pc = PC_R3_TOP; break;
            default: break;
            }
            phase = PH_CTRL;
        }
        PROF_MARK(7);
    }
#ifdef SEED_PROF
    if (lane == 0) for (int k = 0; k < 10; ++k) atomicAdd(&A.counters[5 + k], prof[k]);
#endif
#undef PROF_MARK
    for (int d = 32; d >= 1; d >>= 1) {
        acc_searches += (unsigned)__shfl_xor((int)acc_searches, d);
        acc_windows += (unsigned)__shfl_xor((int)acc_windows, d);
        acc_deep += (unsigned)__shfl_xor((int)acc_deep, d);
    }
    if (lane == 0) {
        atomicAdd(&A.counters[1], (unsigned long long)acc_searches);
        atomicAdd(&A.counters[3], (unsigned long long)acc_windows);
        atomicAdd(&A.counters[4], (unsigned long long)acc_deep);
    }
#undef Q
#undef ST
#undef LD64
#undef ST64
#undef LCP_AT
}

}  // namespace seedk
