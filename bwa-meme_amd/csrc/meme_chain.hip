// Chaining on the device: mem_chain_Learned + mem_chain_flt (reference src/bwamem.cpp:1122-1204, 599-717) for every read
// of the batch the ctx has just seeded -- the seeds never leave HBM between the two stages.
//
// The work per read is strictly sequential (every hit is tested against the chain with the closest position at or below it and either
// merged into it or becomes a new chain) and spans four orders of magnitude, so the reads are tiered by their own work; no host fallback:
//   k_chain       one lane per read, position-sorted array of <= 16 chains of <= 8 seeds            97.9 % of the reads
//   k_chain_lds   one wavefront per read, <= 256 / 512 / 1 024 chains in LDS, wave-parallel introsort / filter   repeats (2.1 %)
//   k_chain_wave  one wavefront per read, the reference's B-tree (klib kbtree.h, t = 5) node for node: reads that put two chains on ONE
//                 position (then the tree's shape decides their order and which one a query finds) and reads beyond 1 024 chains
// What the reference sorts with klib's introsort is sorted by the same sequence of comparisons and swaps in every tier, because chains
// of equal weight keep whatever order that algorithm leaves them in and the filter that follows depends on it.
#include <string.h>
#include <algorithm>

#include "meme_common.h"

namespace {

// Tier 1: every read with room for 16 chains of 8 seeds (2.5 KB per read), as long as one lane can walk it quickly.
constexpr int CHAIN_CAP = 16, SEED_CAP = 8;
constexpr int SMEM_CAP1 = 48;      // SMEMs per read in tier 1 (its sorted walk is quadratic in it)
// (hits per read that one lane is asked to walk -- dependent loads, ~1 us each: meme_ctx::chain_lane_hits, default 256)

struct DChain {                    // 32 bytes
    i64 pos;
    int rid, n, w, first;
    short kept, is_alt;
    int row;                       // which seed row of the read's scratch holds its seeds
};
struct DSeed { i64 rbeg; int qbeg, len; };
struct ReadHdr { int tree_size, n_kept, n_seeds, fallback; i64 slot; i64 work; };   // fallback: 0 done, 1 = for tier 2; slot: scratch index, bit 62 = tier 2; work: hits to walk

struct ChainArgs {
    const meme_mem_tl* smems; const i64* smem_off; const u64* hits; const i64* hit_off; const i64* read_off;
    i64 nreads;
    const i64* contig_off; const int* contig_len; const unsigned char* contig_alt; int n_contigs;
    meme_chain_opt o;
    int hit_cap1;                      // hits a lane of tier 1 is asked to walk
    const unsigned char* cls;          // per read: 0 = tier 1 takes it; 1 / 2 = routed to a wavefront tier by its work before tier 1 runs
    DChain* ch; DSeed* sd; ReadHdr* hdr; float* frac_rep;
};

__device__ inline int pos2rid(const ChainArgs& A, i64 pos_f) {            // bns_pos2rid, src/bntseq.cpp:392-406
    if (pos_f >= A.o.l_pac) return -1;
    int left = 0, mid = 0, right = A.n_contigs;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= A.contig_off[mid]) {
            if (mid == A.n_contigs - 1) break;
            if (pos_f < A.contig_off[mid + 1]) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}
__device__ inline i64 depos(const ChainArgs& A, i64 pos) { return pos >= A.o.l_pac ? (A.o.l_pac << 1) - 1 - pos : pos; }
__device__ inline int intv2rid(const ChainArgs& A, i64 rb, i64 re) {      // bns_intv2rid, src/bntseq.cpp:408-416
    if (rb < A.o.l_pac && re > A.o.l_pac) return -2;
    const int rid_b = pos2rid(A, depos(A, rb));
    const int rid_e = rb < re ? pos2rid(A, depos(A, re - 1)) : rid_b;
    return rid_b == rid_e ? rid_b : -1;
}

// The lane-per-read tier keeps a read's chains and seeds in HBM scratch.  The 64 reads of a wavefront interleave their arrays element by
// element (element i of lane L at [i * 64 + L]): when the lanes touch the same element of their own arrays -- which they mostly do, chain
// 0 first -- the wavefront reads consecutive addresses instead of 64 scattered lines.  SPtr<T> is a pointer into such an array.
template <typename T> struct SPtr {
    T* p;
    __device__ T& operator*() const { return *p; }
    __device__ T* operator->() const { return p; }
    __device__ T& operator[](int i) const { return p[(i64)i * 64]; }
    __device__ SPtr operator+(int i) const { return SPtr{p + (i64)i * 64}; }
    __device__ SPtr operator-(int i) const { return SPtr{p - (i64)i * 64}; }
    __device__ int operator-(SPtr o) const { return (int)((p - o.p) >> 6); }
    __device__ SPtr& operator++() { p += 64; return *this; }
    __device__ SPtr& operator--() { p -= 64; return *this; }
    __device__ bool operator<(SPtr o) const { return p < o.p; }
    __device__ bool operator<=(SPtr o) const { return p <= o.p; }
    __device__ bool operator>(SPtr o) const { return p > o.p; }
    __device__ bool operator==(SPtr o) const { return p == o.p; }
    __device__ bool operator!=(SPtr o) const { return p != o.p; }
};
typedef SPtr<DChain> ChP;
typedef SPtr<DSeed> SdP;

__device__ inline int chain_weight(const DChain& c, SdP row) {   // mem_chain_weight, src/bwamem.cpp:522-541
    i64 end = 0;
    int w = 0;
    for (int j = 0; j < c.n; ++j) {
        const DSeed s = row[j];
        if (s.qbeg >= end) w += s.len;
        else if (s.qbeg + s.len > end) w += (int)(s.qbeg + s.len - end);
        end = end > s.qbeg + s.len ? end : s.qbeg + s.len;
    }
    const int tmp = w;
    w = 0; end = 0;
    for (int j = 0; j < c.n; ++j) {
        const DSeed s = row[j];
        if (s.rbeg >= end) w += s.len;
        else if (s.rbeg + s.len > end) w += (int)(s.rbeg + s.len - end);
        end = end > s.rbeg + s.len ? end : s.rbeg + s.len;
    }
    w = w < tmp ? w : tmp;
    return w < 1 << 30 ? w : (1 << 30) - 1;
}

#define FLT_LT(a_, b_) ((a_).w > (b_).w)                                 // flt_lt, src/bwamem.cpp:80
__device__ inline void swap_chain(DChain& a, DChain& b) { const DChain t = a; a = b; b = t; }

// ks_introsort (klib ksort.h), restated: two elements are compared and swapped; otherwise quicksort around the median of first /
// middle / last with an explicit stack, sub-ranges of at most 16 elements are left to the final insertion sort, comb sort takes
// over when the depth budget is spent.  (Up to 16 elements this is ONE partition pass over the whole array + insertion sort.)
__device__ inline void insert_sort(ChP s, ChP t) {
    for (ChP i = s + 1; i < t; ++i)
        for (ChP j = i; j > s && FLT_LT(*j, *(j - 1)); --j) swap_chain(*j, *(j - 1));
}
__device__ void comb_sort(int n, ChP a) {
    const double shrink = 1.2473309501039786540366528676643;
    bool do_swap;
    int gap = n;
    do {
        if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
        do_swap = false;
        for (ChP i = a; i < a + (n - gap); ++i) { ChP j = i + gap; if (FLT_LT(*j, *i)) { swap_chain(*i, *j); do_swap = true; } }
    } while (do_swap || gap > 2);
    if (gap != 1) insert_sort(a, a + n);
}
__device__ void sort_by_weight(ChP a, int n) {
    if (n < 1) return;
    if (n == 2) { if (FLT_LT(a[1], a[0])) swap_chain(a[0], a[1]); return; }
    struct { ChP left, right; int depth; } stack[40], *top = stack;
    int d;
    for (d = 2; (1 << d) < n; ++d) {}
    ChP s = a, t = a + (n - 1);
    d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) { comb_sort((t - s) + 1, s); t = s; continue; }
            ChP i = s, j = t, k = i + (((j - i) >> 1) + 1);
            if (FLT_LT(*k, *i)) { if (FLT_LT(*k, *j)) k = j; }
            else k = FLT_LT(*j, *i) ? i : j;
            const DChain rp = *k;
            if (k != t) swap_chain(*k, *t);
            for (;;) {
                do ++i; while (FLT_LT(*i, rp));
                do --j; while (i <= j && FLT_LT(rp, *j));
                if (j <= i) break;
                swap_chain(*i, *j);
            }
            swap_chain(*i, *t);
            if (i - s > t - i) {
                if (i - s > 16) { top->left = s; top->right = i - 1; top->depth = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { top->left = i + 1; top->right = t; top->depth = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == stack) { insert_sort(a, a + n); return; }
            --top; s = top->left; t = top->right; d = top->depth;
        }
    }
}

template <int CC, int SC>
__global__ void __launch_bounds__(64) k_chain(ChainArgs A) {
    __shared__ i64 lds_contig1[256];                 // bns_intv2rid searches the contig offsets twice per hit: keep them out of the dependent-load chain
    if (A.n_contigs <= 256) {
        for (int i = threadIdx.x; i < A.n_contigs; i += 64) lds_contig1[i] = A.contig_off[i];
        __syncthreads();
        A.contig_off = lds_contig1;
    }
    const i64 tid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= A.nreads) return;
    const i64 r = tid;
    if (A.cls && A.cls[r]) return;                    // a wavefront tier has this read (k_chain_route)
    const meme_chain_opt& o = A.o;
    const meme_mem_tl* sm = A.smems + A.smem_off[r];
    const int ns = (int)(A.smem_off[r + 1] - A.smem_off[r]);
    const u64* ht = A.hits + A.hit_off[r];
    const int len = (int)(A.read_off[r + 1] - A.read_off[r]);
    const ChP ch{A.ch + (i64)blockIdx.x * 64 * CC + threadIdx.x};           // (interleaved over the wavefront's 64 reads: SPtr)
    const SdP sd{A.sd + (i64)blockIdx.x * 64 * (CC * SC) + threadIdx.x};
    ReadHdr H = {0, 0, 0, 0, tid, 0};
    float frac = 0.f;
    int nc = 0;
    if (len >= o.min_seed_len && ns > 0) {                                // (:1138)
        // One lane walks a read's hits one after the other through dependent loads: a read with hundreds of them (repeats) would hold
        // its wavefront -- and with it the kernel -- for milliseconds.  Those go to the wavefront-per-read tier at once.
        i64 work = 0;
        for (int i = 0; i < ns; ++i) work += sm[i].hitcount < o.max_occ ? sm[i].hitcount : o.max_occ;
        H.work = work;
        if (ns > SMEM_CAP1 || work > A.hit_cap1) H.fallback = 1;
        int b = 0, e = 0, l_rep = 0;                                      // frac_rep (:1140-1147)
        // walk the SMEMs in (start, end) order; records with equal (start, end) describe the same substring, hence the same
        // hits, and the second one only meets seeds that are already contained: their relative order cannot matter
        int ps = -1, pe = -1, pi = -1;
        for (int step_i = 0; step_i < ns && !H.fallback; ++step_i) {
            int bi = -1, bs = 0, be = 0;
            for (int i = 0; i < ns; ++i) {
                const int s = sm[i].start, en = sm[i].end;
                const bool after = s > ps || (s == ps && (en > pe || (en == pe && i > pi)));
                if (!after) continue;
                if (bi < 0 || s < bs || (s == bs && (en < be || (en == be && i < bi)))) { bi = i; bs = s; be = en; }
            }
            ps = bs; pe = be; pi = bi;
            const meme_mem_tl p = sm[bi];
            if (p.hitcount > o.max_occ) {
                if (p.start > e) { l_rep += e - b; b = p.start; e = p.end; }
                else e = e > p.end ? e : p.end;
            }
            const int slen = p.end - p.start;
            const int step = p.hitcount > o.max_occ ? p.hitcount / o.max_occ : 1;
            int count = 0;
            for (i64 k = 0; k < p.hitcount && count < o.max_occ; k += step, ++count) {
                DSeed s;
                s.rbeg = (i64)ht[p.hitbeg + k];
                s.qbeg = p.start;
                s.len = slen;
                const int rid = intv2rid(A, s.rbeg, s.rbeg + s.len);
                if (rid < 0) continue;                                    // bridging two sequences or the strands (:1166)
                int lower = -1;                                           // the chain with the largest position <= the seed's
                if (nc <= 8) { for (int i = 0; i < nc; ++i) { if (ch[i].pos <= s.rbeg) lower = i; else break; } }
                else {                                                    // (repeats: up to 128 chains and 500 hits per SMEM)
                    int lo_i = 0, hi_i = nc;                              // first chain with pos > rbeg
                    while (lo_i < hi_i) { const int mid = (lo_i + hi_i) >> 1; if (ch[mid].pos <= s.rbeg) lo_i = mid + 1; else hi_i = mid; }
                    lower = lo_i - 1;
                }
                bool merged = false;
                if (lower >= 0) {                                         // test_and_merge, src/bwamem.cpp:450-492
                    DChain& c = ch[lower];
                    const SdP row = sd + c.row * SC;
                    const DSeed last = row[c.n - 1], first = row[0];
                    const i64 qend = last.qbeg + last.len, rend = last.rbeg + last.len;
                    if (rid == c.rid) {
                        if (s.qbeg >= first.qbeg && s.qbeg + s.len <= qend && s.rbeg >= first.rbeg && s.rbeg + s.len <= rend) merged = true;
                        else if ((last.rbeg < o.l_pac || first.rbeg < o.l_pac) && s.rbeg >= o.l_pac) merged = false;
                        else {
                            const i64 x = s.qbeg - last.qbeg, y = s.rbeg - last.rbeg;
                            if (y >= 0 && x - y <= o.w && y - x <= o.w && x - last.len < o.max_chain_gap && y - last.len < o.max_chain_gap) {
                                if (c.n == SC) { H.fallback = 1; break; }
                                row[c.n++] = s;
                                merged = true;
                            }
                        }
                    }
                }
                if (!merged) {                                            // a new chain (:1172-1191)
                    if (lower >= 0 && ch[lower].pos == s.rbeg) { H.fallback = 1; break; }   // equal B-tree keys: the tier with the B-tree decides
                    if (nc == CC) { H.fallback = 1; break; }
                    for (int i = nc; i > lower + 1; --i) ch[i] = ch[i - 1];
                    DChain c;
                    c.pos = s.rbeg; c.rid = rid; c.n = 1; c.w = 0; c.first = -1; c.kept = 0;
                    c.is_alt = A.contig_alt[rid] ? 1 : 0;
                    c.row = nc;
                    ch[lower + 1] = c;
                    sd[nc * SC] = s;
                    ++nc;
                }
            }
        }
        l_rep += e - b;
        frac = (float)l_rep / len;                                        // (:1199)
    }
    H.tree_size = nc;
    int n = 0;
    if (!H.fallback && nc > 0) {                                          // mem_chain_flt, src/bwamem.cpp:599-717
        for (int i = 0; i < nc; ++i) {
            DChain c = ch[i];
            c.first = -1; c.kept = 0;
            c.w = chain_weight(c, sd + c.row * SC);
            // (all chains below the floor: the reference's compaction copies nothing, its count becomes 0, and the code behind it still
            // makes one range of the array's stale first element (src/bwamem.cpp:604-643) -- the first chain in tree order survives)
            if (c.w >= o.min_chain_weight) ch[n++] = c;
            else if (i == 0) ch[0] = c;
        }
        if (n == 0) n = 1;
        if (n > 0) {
            sort_by_weight(ch, n);
            int kept_idx[CC];
            int nk = 0;
            ch[0].kept = 3;
            kept_idx[nk++] = 0;
            for (int i = 1; i < n; ++i) {
                bool large_ovlp = false;
                int k = 0;
                const SdP ri = sd + ch[i].row * SC;
                const int beg_i = ri[0].qbeg, end_i = ri[ch[i].n - 1].qbeg + ri[ch[i].n - 1].len;
                for (; k < nk; ++k) {
                    const int j = kept_idx[k];
                    const SdP rj = sd + ch[j].row * SC;
                    const int beg_j = rj[0].qbeg, end_j = rj[ch[j].n - 1].qbeg + rj[ch[j].n - 1].len;
                    const int b_max = beg_j > beg_i ? beg_j : beg_i;
                    const int e_min = end_j < end_i ? end_j : end_i;
                    if (e_min > b_max && (!ch[j].is_alt || ch[i].is_alt)) {
                        const int li = end_i - beg_i, lj = end_j - beg_j;
                        const int min_l = li < lj ? li : lj;
                        if ((float)(e_min - b_max) >= (float)min_l * o.mask_level && min_l < o.max_chain_gap) {
                            large_ovlp = true;
                            if (ch[j].first < 0) ch[j].first = i;
                            if ((float)ch[i].w < (float)ch[j].w * o.drop_ratio && ch[j].w - ch[i].w >= o.min_seed_len << 1) break;
                        }
                    }
                }
                if (k == nk) { kept_idx[nk++] = i; ch[i].kept = large_ovlp ? 2 : 3; }
            }
            for (int i = 0; i < nk; ++i) { const int f = ch[kept_idx[i]].first; if (f >= 0) ch[f].kept = 1; }
            int i = 0, k = 0;
            for (; i < n; ++i) {                                          // at most max_chain_extend chains of kind 1 / 2
                if (ch[i].kept == 0 || ch[i].kept == 3) continue;
                if (++k >= o.max_chain_extend) break;
            }
            for (; i < n; ++i) if (ch[i].kept < 3) ch[i].kept = 0;
            k = 0;
            int nseeds = 0;
            for (i = 0; i < n; ++i) if (ch[i].kept != 0) { const DChain c = ch[i]; ch[k++] = c; nseeds += c.n; }
            H.n_kept = k;
            H.n_seeds = nseeds;
        }
    }
    A.hdr[r] = H;
    A.frac_rep[r] = frac;
}

// ---- tier 2: one wavefront per read ---------------------------------------------------------------------------------------------
// klib's B-tree as the reference instantiates it (src/bwamem.cpp:43-44, kb_init(chn, KB_DEFAULT_SIZE + 8), src/kbtree.h:60-77):
// t = ((520 - 4 - 8) / (8 + sizeof(mem_chain_t) = 48) + 1) >> 1 = 5, at most 9 keys per node.  Keys are chain positions; a node
// also carries the chain ids.  The first NODE_LDS nodes of a read live in LDS, the rest in its HBM scratch.
constexpr int T_ORD = 5, T_MAXK = 2 * T_ORD - 1;
struct TNode { int n, internal; i64 key[T_MAXK]; int cid[T_MAXK]; int ptr[T_MAXK + 1]; int pad; };
static_assert(sizeof(TNode) == 160, "TNode layout");
// The kernel is built with 288 nodes (>= 1 150 chains) and 2 048 sort slots in LDS (63 KB per wavefront): a node or sort slot that spills to HBM
// turns every step of the sequential parts from ~30 ns into ~1 us.
constexpr int CONTIG_LDS = 256;    // contig offsets kept in LDS for bns_intv2rid (all of them when the reference has that few)

struct C2 {                        // chain record (64 bytes), indexed by creation order
    i64 pos; int rid, n;
    i64 f_rbeg; int f_qbeg, f_len; // first seed: what test_and_merge reads besides the last one
    i64 l_rbeg; int l_qbeg, l_len; // last seed
    int head, tail;                // the chain's seeds: a list through S2::next
    int w; short is_alt, pad;
};
static_assert(sizeof(C2) == 64, "C2 layout");
struct S2 { i64 rbeg; int qbeg, len; int next, pad; };
struct FRec { int beg, end, w, first; int kept, is_alt, id, pad; };   // the filter's view of a chain (query span, weight, marks)

struct WaveArgs {
    const i64* list; const i64* woff; i64 nlist;     // reads of this tier, exclusive prefix of their work
    const i64* sub; i64 nsub;                        // optional: the entries of `list` this launch takes
    int set;                                         // which scratch set this launch works in (1..4; recorded in ReadHdr.slot for the pack kernel)
    C2* C; S2* S; FRec* F; u64* srt; int* ia; int* ib; TNode* nodes;
};

struct WTree {
    TNode* lds; TNode* glb;
    int root, n_nodes, n_lds;
    __device__ __forceinline__ TNode* nd(int id) const { return id < n_lds ? lds + id : glb + (id - n_lds); }
    __device__ int fresh() { const int id = n_nodes++; TNode* x = nd(id); x->n = 0; x->internal = 0; return id; }
};

// kb_intervalp, lower side (src/kbtree.h:153-176 with __kb_getp_aux :125-140): the chain test_and_merge is tried on
__device__ int tree_lower(const WTree& T, i64 pos, i64* lpos) {
    int x = T.root, lower = -1;
    i64 lp = 0;
    for (;;) {
        const TNode* nd = T.nd(x);
        const int n = nd->n;
        if (n == 0) break;
        int b = 0, e = n;
        while (b < e) { const int m = (b + e) >> 1; if (nd->key[m] < pos) b = m + 1; else e = m; }
        int i;
        bool eq = false;
        if (b == n) i = n - 1;
        else { eq = nd->key[b] == pos; i = eq ? b : b - 1; }
        if (i >= 0 && eq) { *lpos = pos; return nd->cid[i]; }
        if (i >= 0) { lower = nd->cid[i]; lp = nd->key[i]; }
        if (!nd->internal) break;
        x = nd->ptr[i + 1];
    }
    *lpos = lp;
    return lower;
}
// __kb_getp_aux as kb_putp uses it (result only)
__device__ __forceinline__ int tree_slot(const TNode* nd, i64 pos) {
    const int n = nd->n;
    if (n == 0) return -1;
    int b = 0, e = n;
    while (b < e) { const int m = (b + e) >> 1; if (nd->key[m] < pos) b = m + 1; else e = m; }
    if (b == n) return n - 1;
    return nd->key[b] == pos ? b : b - 1;
}
__device__ void tree_split(WTree& T, int xi, int i, int yi) {                // __kb_split (src/kbtree.h:181-197)
    const int zi = T.fresh();
    TNode *x = T.nd(xi), *y = T.nd(yi), *z = T.nd(zi);
    z->internal = y->internal;
    z->n = T_ORD - 1;
    for (int k = 0; k < T_ORD - 1; ++k) { z->key[k] = y->key[T_ORD + k]; z->cid[k] = y->cid[T_ORD + k]; }
    if (y->internal) for (int k = 0; k < T_ORD; ++k) z->ptr[k] = y->ptr[T_ORD + k];
    y->n = T_ORD - 1;
    for (int k = x->n; k > i; --k) x->ptr[k + 1] = x->ptr[k];
    x->ptr[i + 1] = zi;
    for (int k = x->n - 1; k >= i; --k) { x->key[k + 1] = x->key[k]; x->cid[k + 1] = x->cid[k]; }
    x->key[i] = y->key[T_ORD - 1];
    x->cid[i] = y->cid[T_ORD - 1];
    ++x->n;
}
__device__ void tree_put(WTree& T, int id, i64 pos) {                        // kb_putp + __kb_putp_aux (src/kbtree.h:198-233)
    if (T.nd(T.root)->n == T_MAXK) {
        const int r = T.root, s = T.fresh();
        TNode* sn = T.nd(s);
        sn->internal = 1; sn->n = 0; sn->ptr[0] = r;
        T.root = s;
        tree_split(T, s, 0, r);
    }
    int x = T.root;
    for (;;) {
        TNode* nd = T.nd(x);
        int i = tree_slot(nd, pos);
        if (!nd->internal) {
            for (int k = nd->n - 1; k > i; --k) { nd->key[k + 1] = nd->key[k]; nd->cid[k + 1] = nd->cid[k]; }
            nd->key[i + 1] = pos;
            nd->cid[i + 1] = id;
            ++nd->n;
            return;
        }
        ++i;
        if (T.nd(nd->ptr[i])->n == T_MAXK) {
            tree_split(T, x, i, nd->ptr[i]);
            if (pos > nd->key[i]) ++i;
        }
        x = nd->ptr[i];
    }
}

// what a hit does to the chain below it (test_and_merge, src/bwamem.cpp:450-492): 0 nothing (contained), 1 appended, 2 a new chain
__device__ __forceinline__ int hit_outcome(const C2& c, i64 rbeg, int qbeg, int len, int rid, const meme_chain_opt& o) {
    if (rid != c.rid) return 2;
    const i64 qend = c.l_qbeg + c.l_len, rend = c.l_rbeg + c.l_len;
    if (qbeg >= c.f_qbeg && qbeg + len <= qend && rbeg >= c.f_rbeg && rbeg + len <= rend) return 0;
    if ((c.l_rbeg < o.l_pac || c.f_rbeg < o.l_pac) && rbeg >= o.l_pac) return 2;
    const i64 x = qbeg - c.l_qbeg, y = rbeg - c.l_rbeg;
    if (y >= 0 && x - y <= o.w && y - x <= o.w && x - c.l_len < o.max_chain_gap && y - c.l_len < o.max_chain_gap) return 1;
    return 2;
}

// every lane stores through its own view of shared state below ("uniform" code: all 64 lanes execute the same stores with the same
// values), so a later load by any lane is an ordinary read-after-write of that lane; where ONE lane produces what others read, the
// wavefront passes this fence
__device__ __forceinline__ void wave_fence() { __threadfence(); __syncthreads(); }

template <int NODE_LDS, int SRT_LDS>
__global__ void __launch_bounds__(64) k_chain_wave(ChainArgs A, WaveArgs W, i64 t_first) {
    __shared__ TNode lds_nodes[NODE_LDS];
    __shared__ u64 lds_srt[SRT_LDS];
    __shared__ int stk_x[40], stk_i[20];
    __shared__ i64 lds_contig[CONTIG_LDS];
    if (W.sub && (i64)blockIdx.x >= W.nsub) return;
    const i64 t = W.sub ? W.sub[blockIdx.x] : t_first + blockIdx.x;
    if (t >= W.nlist) return;
    const int lane = threadIdx.x;
    if (A.n_contigs <= CONTIG_LDS) {                 // bns_intv2rid searches the contig offsets twice per hit
        for (int i = lane; i < A.n_contigs; i += 64) lds_contig[i] = A.contig_off[i];
        __syncthreads();
        A.contig_off = lds_contig;
    }
    const i64 r = W.list[t];
    const i64 base = W.woff[t];
    const meme_chain_opt& o = A.o;
    const meme_mem_tl* sm = A.smems + A.smem_off[r];
    const int ns = (int)(A.smem_off[r + 1] - A.smem_off[r]);
    const u64* ht = A.hits + A.hit_off[r];
    const int len = (int)(A.read_off[r + 1] - A.read_off[r]);
    C2* C = W.C + base;
    S2* S = W.S + base;
    FRec* F = W.F + base;
    int* ia = W.ia + base;
    int* ib = W.ib + base;
    WTree T;
    T.lds = lds_nodes;
    T.glb = W.nodes + (base / 3 + 4 * t);
    T.n_nodes = 0;
    T.n_lds = NODE_LDS;
    T.root = T.fresh();
    // ---- the SMEMs in (start, end) order (ks_introsort at src/bwamem.cpp:1397; equal keys describe the same substring, hence the
    //      same hits: their order cannot matter): rank of every SMEM by counting
    for (int i = lane; i < ns; i += 64) {
        const int s = sm[i].start, e = sm[i].end;
        int rank = 0;
        for (int j = 0; j < ns; ++j) {
            const int sj = sm[j].start, ej = sm[j].end;
            rank += (sj < s || (sj == s && (ej < e || (ej == e && j < i)))) ? 1 : 0;
        }
        ia[rank] = i;
    }
    wave_fence();
    // ---- mem_chain_Learned (src/bwamem.cpp:1149-1193)
    int nchain = 0, nseed = 0;
    bool has_dups = false;
    int fb = 0, fe = 0, l_rep = 0;
    for (int si = 0; si < ns; ++si) {
        const meme_mem_tl p = sm[ia[si]];
        if (p.hitcount > o.max_occ) {                                     // frac_rep (:1140-1147)
            if (p.start > fe) { l_rep += fe - fb; fb = p.start; fe = p.end; }
            else fe = fe > p.end ? fe : p.end;
        }
        const int slen = p.end - p.start;
        const int step = p.hitcount > o.max_occ ? p.hitcount / o.max_occ : 1;
        int cnt = (p.hitcount + step - 1) / step;
        if (cnt > o.max_occ) cnt = o.max_occ;
        for (int cb = 0; cb < cnt; cb += 64) {
            const int c = cb + lane;
            bool valid = c < cnt;
            const i64 rbeg = valid ? (i64)ht[p.hitbeg + (i64)c * step] : 0;
            const int rid = valid ? intv2rid(A, rbeg, rbeg + slen) : -1;
            valid = rid >= 0;                                             // bridging two sequences or the strands (:1166)
            int low = -1, out = 0;
            i64 lowpos = 0;
            if (valid) {
                low = tree_lower(T, rbeg, &lowpos);
                out = low >= 0 ? hit_outcome(C[low], rbeg, p.start, slen, rid, o) : 2;
            }
            // commit in hit order
            u64 todo = __ballot(valid), stale = 0;
            for (;;) {
                const u64 act = __ballot(valid && out != 0);
                const u64 cand = todo & (act | stale);
                if (!cand) break;
                const int j = __builtin_ctzll(cand);
                todo &= ~(((u64)2 << j) - 1);                             // (j == 63: the shift wraps to 0, the mask is all ones)
                if ((stale >> j) & 1) {                                   // an earlier hit of this batch touched what j (and others) looked at
                    if ((stale >> lane) & 1) {
                        low = tree_lower(T, rbeg, &lowpos);
                        out = low >= 0 ? hit_outcome(C[low], rbeg, p.start, slen, rid, o) : 2;
                    }
                    stale = 0;
                }
                const int out_j = __shfl(out, j);
                if (out_j == 0) continue;
                const int low_j = __shfl(low, j), rid_j = __shfl(rid, j);
                const i64 rbeg_j = __shfl(rbeg, j), lowpos_j = __shfl(lowpos, j);
                const int sid = nseed++;
                S2 sn;
                sn.rbeg = rbeg_j; sn.qbeg = p.start; sn.len = slen; sn.next = -1; sn.pad = 0;
                S[sid] = sn;
                if (out_j == 1) {                                         // grow the chain (:461-489)
                    C2* c = &C[low_j];
                    S[c->tail].next = sid;
                    c->n += 1; c->l_rbeg = rbeg_j; c->l_qbeg = p.start; c->l_len = slen; c->tail = sid;
                    stale |= todo & __ballot(valid && low == low_j);
                } else {                                                  // a new chain (:1172-1191)
                    const int id = nchain++;
                    C2 c;
                    c.pos = rbeg_j; c.rid = rid_j; c.n = 1;
                    c.f_rbeg = c.l_rbeg = rbeg_j; c.f_qbeg = c.l_qbeg = p.start; c.f_len = c.l_len = slen;
                    c.head = c.tail = sid; c.w = 0; c.is_alt = A.contig_alt[rid_j] ? 1 : 0; c.pad = 0;
                    C[id] = c;
                    if (low_j >= 0 && lowpos_j == rbeg_j) has_dups = true;
                    tree_put(T, id, rbeg_j);
                    // whose view is now out of date: hits between the new chain and the chain they found below them; once equal keys
                    // exist, WHICH of them a descent meets first also depends on the tree's shape, so everybody looks again
                    if (has_dups) stale |= todo & __ballot(valid);
                    else stale |= todo & __ballot(valid && rbeg >= rbeg_j && (low < 0 || lowpos <= rbeg_j));
                }
            }
        }
    }
    l_rep += fe - fb;
    // ---- the chains in tree order (__kb_traverse, :1194-1198) into ib[]
    int m = 0;
    {
        int sp = 0;
        stk_x[0] = T.root; stk_i[0] = 0;
        while (sp >= 0) {
            const TNode* nd = T.nd(stk_x[sp]);
            if (!nd->internal) { for (int i = 0; i < nd->n; ++i) ib[m++] = nd->cid[i]; --sp; continue; }
            const int i = stk_i[sp] >> 1, ph = stk_i[sp] & 1;
            if (i > nd->n) { --sp; continue; }
            if (ph == 0) { stk_i[sp] |= 1; ++sp; stk_x[sp] = nd->ptr[i]; stk_i[sp] = 0; continue; }
            if (i < nd->n) ib[m++] = nd->cid[i];
            stk_i[sp] = (i + 1) << 1;
        }
    }
    // ---- mem_chain_flt (src/bwamem.cpp:599-717): weights, one chain per lane
    for (int c = lane; c < nchain; c += 64) {
        const int id = ib[c];
        i64 end = 0;
        int w = 0;
        for (int k = C[id].head; k >= 0; k = S[k].next) {
            const S2 s = S[k];
            if (s.qbeg >= end) w += s.len;
            else if (s.qbeg + s.len > end) w += (int)(s.qbeg + s.len - end);
            end = end > s.qbeg + s.len ? end : s.qbeg + s.len;
        }
        const int tmp = w;
        w = 0; end = 0;
        for (int k = C[id].head; k >= 0; k = S[k].next) {
            const S2 s = S[k];
            if (s.rbeg >= end) w += s.len;
            else if (s.rbeg + s.len > end) w += (int)(s.rbeg + s.len - end);
            end = end > s.rbeg + s.len ? end : s.rbeg + s.len;
        }
        w = w < tmp ? w : tmp;
        C[id].w = w < 1 << 30 ? w : (1 << 30) - 1;
    }
    wave_fence();
    // chains of at least min_chain_weight, in tree order, as (weight, id) pairs
    u64* srt = nchain <= SRT_LDS ? lds_srt : W.srt + base;
    int n = 0;
    for (int cb = 0; cb < nchain; cb += 64) {
        const int c = cb + lane;
        const int id = c < nchain ? ib[c] : 0;
        const int w = c < nchain ? C[id].w : 0;
        const bool keep = c < nchain && w >= o.min_chain_weight;
        const u64 mk = __ballot(keep);
        if (keep) srt[n + __popcll(mk & (((u64)1 << lane) - 1))] = (u64)(unsigned)w << 32 | (unsigned)id;
        n += __popcll(mk);
    }
    if (n == 0 && nchain > 0) {                                 // all below the floor: the reference keeps the first chain in tree order (see k_chain)
        if (lane == 0) { const int id0 = ib[0]; srt[0] = (u64)(unsigned)C[id0].w << 32 | (unsigned)id0; }
        n = 1;
    }
    wave_fence();
    int n_kept = 0, n_seeds = 0;
    if (n > 0) {
        // ks_introsort(mem_flt) by weight, descending: the same comparisons and swaps, so that chains of equal weight end up where klib leaves them
#define W_LT(a_, b_) (((a_) >> 32) > ((b_) >> 32))
#define W_SWAP(a_, b_) do { const u64 t_ = (a_); (a_) = (b_); (b_) = t_; } while (0)
#define W_INSERT(s_, t_) do { for (u64* i_ = (s_) + 1; i_ < (t_); ++i_) for (u64* j_ = i_; j_ > (s_) && W_LT(*j_, *(j_ - 1)); --j_) W_SWAP(*j_, *(j_ - 1)); } while (0)
        if (n == 2) { if (W_LT(srt[1], srt[0])) W_SWAP(srt[0], srt[1]); }
        else if (n > 2) {
            int d;
            for (d = 2; (1 << d) < n; ++d) {}
            d <<= 1;
            u64 *s = srt, *tt = srt + (n - 1);
            // explicit stack of (left, right, depth) in the int stack arrays: ranges as offsets into srt
            int top = 0;
            for (;;) {
                if (s < tt) {
                    if (--d == 0) {                                       // comb sort (ks_combsort)
                        const int cn = (int)(tt - s) + 1;
                        const double shrink = 1.2473309501039786540366528676643;
                        bool do_swap;
                        int gap = cn;
                        do {
                            if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
                            do_swap = false;
                            for (u64* i = s; i < s + cn - gap; ++i) { u64* j = i + gap; if (W_LT(*j, *i)) { W_SWAP(*i, *j); do_swap = true; } }
                        } while (do_swap || gap > 2);
                        if (gap != 1) W_INSERT(s, s + cn);
                        tt = s;
                        continue;
                    }
                    u64 *i = s, *j = tt, *k = i + ((j - i) >> 1) + 1;
                    if (W_LT(*k, *i)) { if (W_LT(*k, *j)) k = j; }
                    else k = W_LT(*j, *i) ? i : j;
                    const u64 rp = *k;
                    if (k != tt) W_SWAP(*k, *tt);
                    for (;;) {
                        do ++i; while (W_LT(*i, rp));
                        do --j; while (i <= j && W_LT(rp, *j));
                        if (j <= i) break;
                        W_SWAP(*i, *j);
                    }
                    W_SWAP(*i, *tt);
                    if (i - s > tt - i) {
                        if (i - s > 16) { stk_x[top] = (int)(s - srt); stk_i[top] = (int)(i - 1 - srt); stk_x[top + 20] = d; ++top; }
                        s = tt - i > 16 ? i + 1 : tt;
                    } else {
                        if (tt - i > 16) { stk_x[top] = (int)(i + 1 - srt); stk_i[top] = (int)(tt - srt); stk_x[top + 20] = d; ++top; }
                        tt = i - s > 16 ? i - 1 : s;
                    }
                } else {
                    if (top == 0) { W_INSERT(srt, srt + n); break; }
                    --top; s = srt + stk_x[top]; tt = srt + stk_i[top]; d = stk_x[top + 20];
                }
            }
        }
#undef W_LT
#undef W_SWAP
#undef W_INSERT
        // the filter's view of the sorted chains
        for (int i = lane; i < n; i += 64) {
            const int id = (int)(unsigned)srt[i];
            const C2 c = C[id];
            FRec f;
            f.beg = c.f_qbeg; f.end = c.l_qbeg + c.l_len; f.w = c.w; f.first = -1; f.kept = 0; f.is_alt = c.is_alt; f.id = id; f.pad = 0;
            F[i] = f;
        }
        wave_fence();
        // kept list in ia[] (the SMEM order is no longer needed); lane L always handles kept entries L, L + 64, ...
        int nk = 1;
        F[0].kept = 3;
        ia[0] = 0;
        for (int i = 1; i < n; ++i) {
            const FRec fi = F[i];
            bool large = false, broke = false;
            for (int kb = 0; kb < nk && !broke; kb += 64) {
                const int k = kb + lane;
                const bool in = k < nk;
                const int j = in ? ia[k] : 0;
                bool lo = false, br = false;
                int first_j = 0;
                if (in) {
                    const FRec fj = F[j];
                    first_j = fj.first;
                    const int b_max = fj.beg > fi.beg ? fj.beg : fi.beg;
                    const int e_min = fj.end < fi.end ? fj.end : fi.end;
                    if (e_min > b_max && (!fj.is_alt || fi.is_alt)) {
                        const int li = fi.end - fi.beg, lj = fj.end - fj.beg;
                        const int min_l = li < lj ? li : lj;
                        if ((float)(e_min - b_max) >= (float)min_l * o.mask_level && min_l < o.max_chain_gap) {
                            lo = true;
                            br = (float)fi.w < (float)fj.w * o.drop_ratio && fj.w - fi.w >= o.min_seed_len << 1;
                        }
                    }
                }
                const u64 lom = __ballot(lo), brm = __ballot(br);
                u64 upto = ~(u64)0;
                if (brm) { const int kx = __builtin_ctzll(brm); upto = ((u64)2 << kx) - 1; broke = true; }
                if (lo && ((upto >> lane) & 1) && first_j < 0) F[j].first = i;
                if (lom & upto) large = true;
            }
            if (!broke) { ia[nk++] = i; F[i].kept = large ? 2 : 3; }
        }
        for (int k = lane; k < nk; k += 64) { const int f = F[ia[k]].first; if (f >= 0) F[f].kept = 1; }
        wave_fence();
        int i = 0, k = 0;
        for (; i < n; ++i) {                                              // at most max_chain_extend chains of kind 1 / 2
            const int kp = F[i].kept;
            if (kp == 0 || kp == 3) continue;
            if (++k >= o.max_chain_extend) break;
        }
        for (; i < n; ++i) if (F[i].kept < 3) F[i].kept = 0;
        for (i = 0; i < n; ++i) {
            const FRec f = F[i];
            if (f.kept == 0) continue;
            F[n_kept++] = f;
            n_seeds += C[f.id].n;
        }
    }
    if (lane == 0) {
        ReadHdr H;
        H.tree_size = nchain; H.n_kept = n_kept; H.n_seeds = n_seeds; H.fallback = 0; H.slot = t | ((i64)W.set << 60); H.work = W.woff[t + 1] - base;
        A.hdr[r] = H;
        A.frac_rep[r] = (float)l_rep / len;
    }
}

// klib's ks_introsort (src/ksort.h) by descending weight on (EW, EID)[0 .. n) in LDS, executed by one wavefront: every Hoare partition
// step from the two lists of scan stops (ballots), the swaps in parallel, the closing insertion sort as a stable rank by counting
// (tests/test_introsort_model.py checks this formulation against the sequential algorithm).  LIDX / RIDX: scratch of n ints each;
// stk: 60 ints.  All lanes call it together.
__device__ void wave_introsort_lds(int* EW, int* EID, int n, int* LIDX, int* RIDX, int* stk, int lane) {
    const u64 below = ((u64)1 << lane) - 1;
#define L_SWAP(i_, j_) do { const int i__ = (i_), j__ = (j_); const int wi_ = EW[i__], di_ = EID[i__], wj_ = EW[j__], dj_ = EID[j__]; \
                            EW[i__] = wj_; EID[i__] = dj_; EW[j__] = wi_; EID[j__] = di_; } while (0)
    if (n == 2) { if (EW[1] > EW[0]) L_SWAP(0, 1); }
    else if (n > 2) {
        int d;
        for (d = 2; (1 << d) < n; ++d) {}
        d <<= 1;
        int s = 0, tt = n - 1, top = 0;
        for (;;) {
            if (s < tt) {
                if (--d == 0) {                               // ks_combsort, as written (rare: the depth budget is 2 log2 n)
                    const int cnn = tt - s + 1;
                    const double shrink = 1.2473309501039786540366528676643;
                    bool do_swap;
                    int gap = cnn;
                    do {
                        if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
                        do_swap = false;
                        for (int i = s; i < s + cnn - gap; ++i) { const int j = i + gap; if (EW[j] > EW[i]) { L_SWAP(i, j); do_swap = true; } }
                    } while (do_swap || gap > 2);
                    if (gap != 1) for (int i = s + 1; i < s + cnn; ++i) for (int j = i; j > s && EW[j] > EW[j - 1]; --j) L_SWAP(j, j - 1);
                    tt = s;
                    continue;
                }
                int k = s + ((tt - s) >> 1) + 1;
                { const int wi = EW[s], wj = EW[tt], wk = EW[k]; if (wk > wi) { if (wk > wj) k = tt; } else k = wj > wi ? s : tt; }
                const int rpw = EW[k];
                if (k != tt) L_SWAP(k, tt);
                // the stops of the up-scan (weight <= pivot; positions s+1 .. tt, ascending) and of the down-scan (weight >= pivot; tt-1 .. s+1, descending)
                int nL = 0, nR = 0;
                for (int cb = s + 1; cb <= tt; cb += 64) {
                    const int p = cb + lane;
                    const bool is = p <= tt && EW[p] <= rpw;
                    const u64 m = __ballot(is);
                    if (is) LIDX[nL + __popcll(m & below)] = p;
                    nL += __popcll(m);
                }
                for (int cb = tt - 1; cb > s; cb -= 64) {
                    const int p = cb - lane;
                    const bool is = p > s && EW[p] >= rpw;
                    const u64 m = __ballot(is);
                    if (is) RIDX[nR + __popcll(m & below)] = p;
                    nR += __popcll(m);
                }
                __syncthreads();
                const int nmin = nL < nR ? nL : nR;
                int mm = 0;
                for (int kb = 0; kb < nmin; kb += 64) {
                    const int kk = kb + lane;
                    const u64 m = __ballot(kk < nmin && LIDX[kk] < RIDX[kk]);
                    mm += __popcll(m);
                    if (m != ~(u64)0) break;                // (the condition holds for a prefix of the pairs)
                }
                for (int kk = lane; kk < mm; kk += 64) L_SWAP(LIDX[kk], RIDX[kk]);
                __syncthreads();
                int i = LIDX[mm];
                if (mm >= 1 && RIDX[mm - 1] < i) i = RIDX[mm - 1];
                L_SWAP(i, tt);
                if (i - s > tt - i) {
                    if (i - s > 16) { stk[top] = s; stk[top + 20] = i - 1; stk[top + 40] = d; ++top; }
                    s = tt - i > 16 ? i + 1 : tt;
                } else {
                    if (tt - i > 16) { stk[top] = i + 1; stk[top + 20] = tt; stk[top + 40] = d; ++top; }
                    tt = i - s > 16 ? i - 1 : s;
                }
            } else {
                if (top == 0) break;
                --top; s = stk[top]; tt = stk[top + 20]; d = stk[top + 40];
            }
        }
        // __ks_insertsort over the whole array: a stable sort by descending weight
        __syncthreads();
        for (int e = lane; e < n; e += 64) {
            const int w = EW[e];
            int rank = 0;
            for (int f = 0; f < n; ++f) { const int wf = EW[f]; rank += (wf > w || (wf == w && f < e)) ? 1 : 0; }
            LIDX[rank] = w; RIDX[rank] = EID[e];
        }
        __syncthreads();
        for (int e = lane; e < n; e += 64) { EW[e] = LIDX[e]; EID[e] = RIDX[e]; }
    }
#undef L_SWAP
}

// ---- wavefront helpers ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int rdl(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ i64 rdl64(i64 v, int l) {
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
    return ((i64)hi << 32) | (i64)(unsigned)lo;
}
__device__ __forceinline__ i64 wave_max64(i64 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const i64 y = __shfl_xor(v, d); v = y > v ? y : v; }
    return v;
}

#ifdef MEME_CHAIN_PROF
__device__ unsigned long long g_prof[32];
#define PROF_T0 unsigned long long prof_t = __builtin_readcyclecounter()
#define PROF(slot_) do { const unsigned long long n__ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_prof[slot_], n__ - prof_t); prof_t = n__; } while (0)
#else
#define PROF_T0 do {} while (0)
#define PROF(slot_) do {} while (0)
#endif
// minimum of an unsigned over the wavefront, through DPP: xor 1, xor 2, half-row mirror, row mirror leave every row of 16 lanes with its
// minimum; the four rows meet on the scalar side.  (Six dependent ds_bpermute round trips per hit was what the lookup used to cost.)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    unsigned y;
    y = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false); v = y < v ? y : v;      // quad_perm [1,0,3,2]
    y = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false); v = y < v ? y : v;      // quad_perm [2,3,0,1]
    y = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false); v = y < v ? y : v;     // row_half_mirror
    y = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false); v = y < v ? y : v;     // row_mirror
    const unsigned a = (unsigned)rdl((int)v, 0), b = (unsigned)rdl((int)v, 16), c = (unsigned)rdl((int)v, 32), d = (unsigned)rdl((int)v, 48);
    const unsigned ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
// distance of a chain position below (or at) rb, saturated: 0xffffffff = no such chain, 0xfffffffe = one that is at least that far away
constexpr unsigned DIST_NONE = 0xffffffffu, DIST_FAR = 0xfffffffeu;
__device__ __forceinline__ unsigned dist_below(i64 pos, i64 rb) {
    const i64 d = rb - pos;
    return (pos < 0 || d < 0) ? DIST_NONE : (d >= (i64)DIST_FAR ? DIST_FAR : (unsigned)d);
}

// ---- tier 2: one wavefront per read, the chains in LDS --------------------------------------------------------------------------------
// Repeat-rich reads make tens to hundreds of chains from hundreds of hits.  As long as all chain positions differ, the B-tree's in-order
// traversal is simply the order of the positions and what the reference asks its tree (the chain with the largest position at or below the
// hit) is a minimum over distances: chain fields as arrays in LDS, the lookup a strided scan of the positions by all lanes plus one
// wave-wide minimum (DPP) of the 32-bit distance; a read that would put two chains on ONE position (then the tree's shape matters) or
// needs more than N chains leaves with fallback = 3 and is done by the B-tree tier.  mem_chain_weight's two coverage sums are kept
// online per chain.  What is sequential per chain in the lane tier is done wave-parallel here, exactly:
//  * klib's introsort (src/ksort.h): each Hoare partition step in O(range / 64) -- the k-th swap of the sequential loop pairs the k-th
//    element from the left that stops the up-scan (weight <= pivot) with the k-th from the right that stops the down-scan (weight >=
//    pivot), as long as the left one lies before the right one; the lists come from ballots, the swaps run in parallel, and the scan
//    position the loop ends on follows from the two lists (checked against the sequential loop on 200 000 random arrays full of ties);
//    the closing insertion sort over the whole array is a stable sort, i.e. a rank by counting;
//  * mem_chain_flt's overlap loop: 64 kept chains per step, the first shadowing one by ballot.
// Built for N = 256 (16 KB of LDS: nine reads per CU), 512 (30 KB: five) and 1 024 (59 KB: two); a read goes to the smallest one whose N
// covers its hits (a read cannot make more chains than it has hits).  The wavefronts raise their issue priority: they run beside the
// lane-per-read tier, which fills every SIMD, and are the long pole.  (Round 3 also had this tier with the chains in registers, slot =
// k * 64 + lane, fields read through v_readlane: same speed as N = 256 here, three hundred lines more; removed.)

constexpr int PAR_MIN_HITS = 3;         // hits of one SMEM batch from which on they may be tested against the chains at once

template <int N>
__global__ void __launch_bounds__(64) k_chain_lds(ChainArgs A, WaveArgs W) {
    static_assert(N <= 2048, "the lookup key keeps the slot in 11 bits");
    __builtin_amdgcn_s_setprio(3);
    __shared__ i64 s_pos[N];
    __shared__ int s_a[12][N];
    __shared__ i64 lds_contig[CONTIG_LDS];
    __shared__ int stk[60];
    int* const RID = s_a[0]; int* const CN = s_a[1]; int* const FQB = s_a[2]; int* const LRB = s_a[3]; int* const LQB = s_a[4]; int* const LLN = s_a[5];
    int* const TAIL = s_a[6]; int* const WQ = s_a[7]; int* const EQ = s_a[8]; int* const WR = s_a[9]; int* const ER = s_a[10];
    // after the walk: per chain ALT = s_a[0], CN, FQB (= query begin), END = s_a[4], CW = s_a[7]; per sorted element:
    int* const EW = s_a[3]; int* const EID = s_a[5]; int* const EBEG = s_a[6]; int* const EEND = s_a[8]; int* const EALT = s_a[9]; int* const EFIRST = s_a[10];
    int* const EKEPT = s_a[11];
    int* const LIDX = reinterpret_cast<int*>(s_pos);            // scratch of the sort / the filter, once the positions are ranked
    int* const RIDX = LIDX + N;
    if (W.sub && (i64)blockIdx.x >= W.nsub) return;
    const i64 t = W.sub ? W.sub[blockIdx.x] : (i64)blockIdx.x;
    if (t >= W.nlist) return;
    const int lane = threadIdx.x;
    const u64 below = ((u64)1 << lane) - 1;
    if (A.n_contigs <= CONTIG_LDS) {
        for (int i = lane; i < A.n_contigs; i += 64) lds_contig[i] = A.contig_off[i];
        __syncthreads();
        A.contig_off = lds_contig;
    }
    PROF_T0;
    const i64 r = W.list[t];
    const i64 base = W.woff[t];
    const meme_chain_opt& o = A.o;
    const meme_mem_tl* sm = A.smems + A.smem_off[r];
    const int ns = (int)(A.smem_off[r + 1] - A.smem_off[r]);
    const u64* ht = A.hits + A.hit_off[r];
    const int len = (int)(A.read_off[r + 1] - A.read_off[r]);
    C2* C = W.C + base;
    S2* S = W.S + base;
    FRec* F = W.F + base;
    int* ia = W.ia + base;
    for (int i = lane; i < ns; i += 64) {                       // SMEMs in (start, end) order (src/bwamem.cpp:1397)
        const int s = sm[i].start, e = sm[i].end;
        int rank = 0;
        for (int j = 0; j < ns; ++j) {
            const int sj = sm[j].start, ej = sm[j].end;
            rank += (sj < s || (sj == s && (ej < e || (ej == e && j < i)))) ? 1 : 0;
        }
        ia[rank] = i;
    }
    wave_fence();
    PROF(8);
    int nchain = 0, nseed = 0, bail = 0;
    int fb = 0, fe = 0, l_rep = 0;
    for (int si = 0; si < ns && !bail; ++si) {
        const meme_mem_tl p = sm[ia[si]];
        if (p.hitcount > o.max_occ) {
            if (p.start > fe) { l_rep += fe - fb; fb = p.start; fe = p.end; }
            else fe = fe > p.end ? fe : p.end;
        }
        const int slen = p.end - p.start, qb = p.start;
        const int step = p.hitcount > o.max_occ ? p.hitcount / o.max_occ : 1;
        int cnt = (p.hitcount + step - 1) / step;
        if (cnt > o.max_occ) cnt = o.max_occ;
        for (int cb = 0; cb < cnt && !bail; cb += 64) {
            const int c = cb + lane;
            const bool have = c < cnt;
            const i64 h_rbeg = have ? (i64)ht[p.hitbeg + (i64)c * step] : 0;
            const int h_rid = have ? intv2rid(A, h_rbeg, h_rbeg + slen) : -1;
            u64 rem = __ballot(h_rid >= 0);                                 // (:1166: seeds bridging two sequences or the strands are dropped)
            // The hits of one SMEM share query span and length; most of them fall far from each other.  With enough of them the 64 hits of the
            // batch are tested against the chains AT ONCE (every lane its own hit, the chains read from LDS) and the longest prefix of them
            // whose outcomes cannot depend on each other is committed in one step; the rest is tested again against the new state.  A hit's
            // outcome depends on an earlier hit of the batch only if that one (a) appended to the chain this one is tested against, or
            // (b) opened a new chain at a position between this hit's lower chain and the hit itself: then that chain becomes the lower one --
            // a single seed of the same span, which this hit can only merge with when it starts within the band w behind it (or at the same
            // position: the B-tree tier's case); farther away the outcome is "new chain" whatever the snapshot said.
            const bool par_mode = __popcll(rem) >= PAR_MIN_HITS && __popcll(rem) * 16 > nchain + 16;   // (one round costs about a hit per 16 chains)
            auto seq_hits = [&](u64 todo) {
            while (todo) {
                const int j = __builtin_ctzll(todo);
                todo &= todo - 1;
                const i64 rb = rdl64(h_rbeg, j);
                const int hr = rdl(h_rid, j);
                // the chain with the largest position <= rb: per lane the closest of its slots (distance in 32 bits), the wave's minimum through
                // DPP; a far-away answer is looked up again as (position << 11 | slot) in 64 bits
                unsigned dbest = DIST_NONE;
                int sbest = 0;
                for (int s = lane; s < nchain; s += 64) { const unsigned d = dist_below(s_pos[s], rb); if (d < dbest) { dbest = d; sbest = s; } }
                const unsigned dm = wave_min_u32(dbest);
                int out = 2, ow = 0;
                i64 mp = -1;
                if (dm < DIST_FAR) { ow = rdl(sbest, __builtin_ctzll(__ballot(dbest == dm))); mp = rb - (i64)dm; }
                else if (dm == DIST_FAR) {
                    i64 best = -1;
                    for (int s = lane; s < nchain; s += 64) { const i64 ps = s_pos[s]; if (ps <= rb) { const i64 key = (ps << 11) | s; best = key > best ? key : best; } }
                    const i64 M = wave_max64(best);
                    ow = (int)(M & 2047);
                    mp = M >> 11;
                }
                if (mp >= 0) {
                    const int c_rid = RID[ow], c_fqb = FQB[ow], c_lrb = LRB[ow], c_lqb = LQB[ow], c_lln = LLN[ow];
                    const i64 l_rbeg = mp + c_lrb;
                    if (hr == c_rid) {                          // test_and_merge (:450-492)
                        const i64 qend = c_lqb + c_lln, rend = l_rbeg + c_lln;
                        if (qb >= c_fqb && qb + slen <= qend && rb >= mp && rb + slen <= rend) out = 0;
                        else if ((l_rbeg < o.l_pac || mp < o.l_pac) && rb >= o.l_pac) out = 2;
                        else {
                            const i64 x = qb - c_lqb, y = rb - l_rbeg;
                            if (y >= 0 && x - y <= o.w && y - x <= o.w && x - c_lln < o.max_chain_gap && y - c_lln < o.max_chain_gap) out = 1;
                        }
                    }
                }
                if (out == 0) continue;
                const int sid = nseed++;
                if (lane == 0) { S2 sn; sn.rbeg = rb; sn.qbeg = qb; sn.len = slen; sn.next = -1; sn.pad = 0; S[sid] = sn; }
                if (out == 1) {                                 // (every lane stores the same values: later reads are the reader's own writes)
                    if (lane == 0) S[TAIL[ow]].next = sid;
                    const int rel = (int)(rb - mp);
                    CN[ow] += 1; LRB[ow] = rel; LQB[ow] = qb; LLN[ow] = slen; TAIL[ow] = sid;
                    const int q_e = EQ[ow], r_e = ER[ow];
                    if (qb >= q_e) WQ[ow] += slen; else if (qb + slen > q_e) WQ[ow] += qb + slen - q_e;
                    EQ[ow] = q_e > qb + slen ? q_e : qb + slen;
                    if (rel >= r_e) WR[ow] += slen; else if (rel + slen > r_e) WR[ow] += rel + slen - r_e;
                    ER[ow] = r_e > rel + slen ? r_e : rel + slen;
                } else {
                    if (mp == rb || nchain == N) { bail = 1; break; }
                    const int id = nchain++;
                    if (lane == 0) C[id].head = sid;
                    s_pos[id] = rb; RID[id] = hr; CN[id] = 1; FQB[id] = qb; LRB[id] = 0; LQB[id] = qb; LLN[id] = slen; TAIL[id] = sid;
                    WQ[id] = slen; EQ[id] = qb + slen; WR[id] = slen; ER[id] = slen;
                }
            }
            };
            if (!par_mode) { seq_hits(rem); continue; }
            __syncthreads();                                    // (seed records stored by lane 0 above are complete before another lane links to them)
            while (rem && !bail) {
                // 1. the lower chain of every remaining hit in the current state
                const i64 rbj = h_rbeg;
                i64 bp = -1;
                int bs = 0;
                for (int s2 = 0; s2 < nchain; ++s2) { const i64 ps = s_pos[s2]; if (ps <= rbj && ps > bp) { bp = ps; bs = s2; } }
                // 2. test_and_merge against it (:450-492)
                int out = 2;
                if (bp >= 0) {
                    const int c_rid = RID[bs], c_fqb = FQB[bs], c_lrb = LRB[bs], c_lqb = LQB[bs], c_lln = LLN[bs];
                    const i64 l_rbeg = bp + c_lrb;
                    if (h_rid == c_rid) {
                        const i64 qend = c_lqb + c_lln, rend = l_rbeg + c_lln;
                        if (qb >= c_fqb && qb + slen <= qend && rbj >= bp && rbj + slen <= rend) out = 0;
                        else if ((l_rbeg < o.l_pac || bp < o.l_pac) && rbj >= o.l_pac) out = 2;
                        else {
                            const i64 x = qb - c_lqb, y = rbj - l_rbeg;
                            if (y >= 0 && x - y <= o.w && y - x <= o.w && x - c_lln < o.max_chain_gap && y - c_lln < o.max_chain_gap) out = 1;
                        }
                    }
                }
                // 3. in hit order: which outcomes stand
                const bool mine = (rem >> lane) & 1;
                i64 mp = bp;
                int conf = 0, newlow = 0;
                u64 com = 0;
                for (u64 scan = rem; scan;) {
                    const int i = __builtin_ctzll(scan);
                    scan &= scan - 1;
                    if (rdl(conf, i)) break;
                    com |= (u64)1 << i;
                    const int oi = rdl(out, i);
                    if (oi == 2) {
                        const i64 rbi = rdl64(h_rbeg, i);
                        if (mine && lane > i && mp < rbi && rbi <= rbj) {
                            if (rbj - rbi <= (i64)(o.w > 0 ? o.w : 0)) conf = 1;
                            else { out = 2; mp = rbi; newlow = 1; }
                        }
                    } else if (oi == 1) {
                        const int ci = rdl(bs, i);
                        if (mine && lane > i && !newlow && bp >= 0 && bs == ci) conf = 1;
                    }
                }
                // 4. commit
                const bool in = (com >> lane) & 1;
                const u64 nzm = __ballot(in && out != 0), newm = __ballot(in && out == 2);
                if (__ballot(in && out == 2 && !newlow && bp == rbj) != 0 || nchain + __popcll(newm) > N) { bail = 1; break; }   // equal positions / too many chains
                const int sid = nseed + __popcll(nzm & below), id = nchain + __popcll(newm & below);
                if (in && out != 0) { S2 sn; sn.rbeg = rbj; sn.qbeg = qb; sn.len = slen; sn.next = -1; sn.pad = 0; S[sid] = sn; }
                if (in && out == 1) {
                    S[TAIL[bs]].next = sid;
                    const int rel = (int)(rbj - bp);
                    CN[bs] += 1; LRB[bs] = rel; LQB[bs] = qb; LLN[bs] = slen; TAIL[bs] = sid;
                    const int q_e = EQ[bs], r_e = ER[bs];
                    if (qb >= q_e) WQ[bs] += slen; else if (qb + slen > q_e) WQ[bs] += qb + slen - q_e;
                    EQ[bs] = q_e > qb + slen ? q_e : qb + slen;
                    if (rel >= r_e) WR[bs] += slen; else if (rel + slen > r_e) WR[bs] += rel + slen - r_e;
                    ER[bs] = r_e > rel + slen ? r_e : rel + slen;
                } else if (in && out == 2) {
                    C[id].head = sid;
                    s_pos[id] = rbj; RID[id] = h_rid; CN[id] = 1; FQB[id] = qb; LRB[id] = 0; LQB[id] = qb; LLN[id] = slen; TAIL[id] = sid;
                    WQ[id] = slen; EQ[id] = qb + slen; WR[id] = slen; ER[id] = slen;
                }
                nseed += __popcll(nzm); nchain += __popcll(newm);
                rem &= ~com;
                __syncthreads();
                // a run of hits that depend on each other (tandem copies): a few of them one by one, then at once again
                if (__popcll(com) <= 2 && rem) {
                    u64 few = 0;
                    for (int k = 0; k < 8 && (rem & ~few); ++k) few |= (rem & ~few) & (~(rem & ~few) + 1);
                    seq_hits(few);
                    rem &= ~few;
                    __syncthreads();
                }
            }
        }
    }
    PROF(9);
    l_rep += fe - fb;
    if (bail) {
        if (lane == 0) { ReadHdr H; H.tree_size = 0; H.n_kept = 0; H.n_seeds = 0; H.fallback = 3; H.slot = t | ((i64)W.set << 60); H.work = W.woff[t + 1] - base; A.hdr[r] = H; }
        return;
    }
    // ---- chain records for the pack kernel; weight, query end and ALT flag per chain
    __syncthreads();
    for (int s = lane; s < nchain; s += 64) {
        int w = WQ[s] < WR[s] ? WQ[s] : WR[s];
        w = w < 1 << 30 ? w : (1 << 30) - 1;
        const int alt = A.contig_alt[RID[s]] ? 1 : 0;
        C2* c = &C[s];
        c->pos = s_pos[s]; c->rid = RID[s]; c->n = CN[s]; c->w = w; c->is_alt = (short)alt; c->tail = TAIL[s];
        WQ[s] = w; RID[s] = alt; LQB[s] = LQB[s] + LLN[s];
    }
    __syncthreads();
    int* const ALT = RID; int* const END = LQB; int* const CW = WQ;
    // tree order = order of the positions: rank by counting, elements (weight, chain) scattered to their rank
    for (int s = lane; s < nchain; s += 64) {
        const i64 ps = s_pos[s];
        int rank = 0;
        for (int u = 0; u < nchain; ++u) rank += s_pos[u] < ps ? 1 : 0;
        EW[rank] = CW[s]; EID[rank] = s;
    }
    __syncthreads();
    PROF(10);
    int n = nchain;
    if (o.min_chain_weight > 0) {                               // (default 0: nothing is dropped)
        n = 0;
        for (int e = 0; e < nchain; ++e) { const int w = EW[e], id = EID[e]; if (w >= o.min_chain_weight) { EW[n] = w; EID[n] = id; ++n; } }
        if (n == 0 && nchain > 0) n = 1;                        // all below the floor: the reference keeps the first chain in tree order (see k_chain)
    }
    int n_kept = 0, n_seeds = 0;
    if (n > 0) {
        PROF(11);
        wave_introsort_lds(EW, EID, n, LIDX, RIDX, stk, lane);
        PROF(12);
        __syncthreads();
        for (int e = lane; e < n; e += 64) { const int id = EID[e]; EBEG[e] = FQB[id]; EEND[e] = END[id]; EALT[e] = ALT[id]; EFIRST[e] = -1; EKEPT[e] = 0; }
        __syncthreads();
        {
            // the filter's view of the 64 heaviest chains lives in the lanes (chain e in lane e): most chains are shadowed by one of those,
            // and then the step costs no LDS round trip; chain i's own values are requested one step ahead
            const int be0 = lane < n ? EBEG[lane] : 0, ee0 = lane < n ? EEND[lane] : 0, we0 = lane < n ? EW[lane] : 0, alt0 = lane < n ? EALT[lane] : 0;
            int kept0 = lane == 0 ? 3 : 0, first0 = -1;
            int nbi = n > 1 ? EBEG[1] : 0, nei = n > 1 ? EEND[1] : 0, nwi = n > 1 ? EW[1] : 0, nai = n > 1 ? EALT[1] : 0;
            for (int i = 1; i < n; ++i) {
                const int bi = nbi, ei = nei, wi = nwi, ai = nai;
                if (i + 1 < n) { nbi = EBEG[i + 1]; nei = EEND[i + 1]; nwi = EW[i + 1]; nai = EALT[i + 1]; }
                bool large = false;
                int brk = -1;
                {
                    bool lo = false, br = false;
                    if (lane < i && kept0 != 0) {
                        const int b_max = be0 > bi ? be0 : bi, e_min = ee0 < ei ? ee0 : ei;
                        if (e_min > b_max && (!alt0 || ai)) {
                            const int li = ei - bi, lj = ee0 - be0;
                            const int min_l = li < lj ? li : lj;
                            if ((float)(e_min - b_max) >= (float)min_l * o.mask_level && min_l < o.max_chain_gap) {
                                lo = true;
                                br = (float)wi < (float)we0 * o.drop_ratio && we0 - wi >= o.min_seed_len << 1;
                            }
                        }
                    }
                    const u64 lom = __ballot(lo), brm = __ballot(br);
                    u64 upto = ~(u64)0;
                    if (brm) { const int kx = __builtin_ctzll(brm); brk = kx; upto = ((u64)2 << kx) - 1; }
                    if (lo && ((upto >> lane) & 1) && first0 < 0) first0 = i;
                    if (lom & upto) large = true;
                }
                for (int cb = 64; cb < i && brk < 0; cb += 64) {
                    const int e = cb + lane;
                    bool lo = false, br = false;
                    if (e < i && EKEPT[e] != 0) {
                        const int be = EBEG[e], ee = EEND[e], we = EW[e];
                        const int b_max = be > bi ? be : bi, e_min = ee < ei ? ee : ei;
                        if (e_min > b_max && (!EALT[e] || ai)) {
                            const int li = ei - bi, lj = ee - be;
                            const int min_l = li < lj ? li : lj;
                            if ((float)(e_min - b_max) >= (float)min_l * o.mask_level && min_l < o.max_chain_gap) {
                                lo = true;
                                br = (float)wi < (float)we * o.drop_ratio && we - wi >= o.min_seed_len << 1;
                            }
                        }
                    }
                    const u64 lom = __ballot(lo), brm = __ballot(br);
                    u64 upto = ~(u64)0;
                    if (brm) { const int kx = __builtin_ctzll(brm); brk = cb + kx; upto = ((u64)2 << kx) - 1; }
                    if (lo && ((upto >> lane) & 1) && EFIRST[e] < 0) EFIRST[e] = i;
                    if (lom & upto) large = true;
                }
                if (brk < 0) {
                    if (i < 64) { if (lane == i) kept0 = large ? 2 : 3; }
                    else EKEPT[i] = large ? 2 : 3;
                }
            }
            if (lane < n) { EFIRST[lane] = first0; EKEPT[lane] = kept0; }
        }
        PROF(13);
        __syncthreads();
        for (int e = lane; e < n; e += 64) LIDX[e] = EKEPT[e];      // the kept list before the marks below
        __syncthreads();
        for (int e = lane; e < n; e += 64) { if (LIDX[e] != 0) { const int f = EFIRST[e]; if (f >= 0) EKEPT[f] = 1; } }
        __syncthreads();
        int n12 = 0;                                                    // chains of kind 1 / 2: at most max_chain_extend of them are kept (default: no limit)
        for (int cb = 0; cb < n; cb += 64) { const int e = cb + lane; n12 += __popcll(__ballot(e < n && (EKEPT[e] == 1 || EKEPT[e] == 2))); }
        if (n12 >= o.max_chain_extend) {
            int i = 0, k = 0;
            for (; i < n; ++i) {
                const int kp = EKEPT[i];
                if (kp == 0 || kp == 3) continue;
                if (++k >= o.max_chain_extend) break;
            }
            for (; i < n; ++i) if (EKEPT[i] < 3) EKEPT[i] = 0;
        }
        int before = 0;
        for (int cb = 0; cb < n; cb += 64) {
            const int e = cb + lane;
            const bool kp = e < n && EKEPT[e] != 0;
            const u64 km = __ballot(kp);
            int nsd = 0;
            if (kp) {
                FRec f;
                f.beg = EBEG[e]; f.end = EEND[e]; f.w = EW[e]; f.first = EFIRST[e]; f.kept = EKEPT[e]; f.is_alt = EALT[e]; f.id = EID[e]; f.pad = 0;
                F[before + __popcll(km & below)] = f;
                nsd = CN[f.id];
            }
            for (int dd = 32; dd >= 1; dd >>= 1) nsd += __shfl_xor(nsd, dd);
            n_seeds += nsd;
            before += __popcll(km);
        }
        n_kept = before;
    }
    PROF(14);
    if (lane == 0) {
        ReadHdr H;
        H.tree_size = nchain; H.n_kept = n_kept; H.n_seeds = n_seeds; H.fallback = 0; H.slot = t | ((i64)W.set << 60); H.work = W.woff[t + 1] - base;
        A.hdr[r] = H;
        A.frac_rep[r] = (float)l_rep / len;
    }
}

// the kept chains and their seeds, densely packed in read order (from the scratch of the tier that finished the read)
struct PackSets { const i64* woff[6]; const C2* C[6]; const S2* S[6]; const FRec* F[6]; };      // [1..5]: the scratch sets of the wavefront tiers

__global__ void __launch_bounds__(256) k_chain_pack(const DChain* __restrict__ ch1, const DSeed* __restrict__ sd1, PackSets P,
                                                     const ReadHdr* __restrict__ hdr, const i64* __restrict__ chain_off,
                                                     const i64* __restrict__ seed_off, i64 nreads, meme_chain* __restrict__ out_ch,
                                                     meme_chain_seed* __restrict__ out_sd) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += (i64)gridDim.x * blockDim.x) {
        const ReadHdr H = hdr[r];
        if (H.n_kept == 0) continue;
        const int set = (int)((H.slot >> 60) & 7);
        const i64 slot = H.slot & (((i64)1 << 60) - 1);
        i64 so = seed_off[r];
        const i64 s0 = so;
        if (set == 0) {
            const SPtr<const DChain> ch{ch1 + (slot >> 6) * 64 * CHAIN_CAP + (slot & 63)};       // (the lane tier's interleaved scratch)
            const SPtr<const DSeed> sd{sd1 + (slot >> 6) * 64 * (i64)CHAIN_CAP * SEED_CAP + (slot & 63)};
            for (int k = 0; k < H.n_kept; ++k) {
                const DChain c = ch[k];
                meme_chain m;
                m.pos = c.pos; m.rid = c.rid; m.n_seeds = c.n; m.w = c.w; m.first = c.first; m.kept = c.kept; m.is_alt = c.is_alt;
                m.seed_beg = (int32_t)(so - s0);
                m.pad = 0;
                out_ch[chain_off[r] + k] = m;
                const SPtr<const DSeed> row = sd + c.row * SEED_CAP;
                for (int j = 0; j < c.n; ++j) { meme_chain_seed s; s.rbeg = row[j].rbeg; s.qbeg = row[j].qbeg; s.len = row[j].len; out_sd[so++] = s; }
            }
        } else {
            const i64 base = P.woff[set][slot];
            const C2* C = P.C[set] + base;
            const S2* S = P.S[set] + base;
            const FRec* F = P.F[set] + base;
            for (int k = 0; k < H.n_kept; ++k) {
                const FRec f = F[k];
                const C2 c = C[f.id];
                meme_chain m;
                m.pos = c.pos; m.rid = c.rid; m.n_seeds = c.n; m.w = f.w; m.first = f.first; m.kept = (int16_t)f.kept; m.is_alt = c.is_alt;
                m.seed_beg = (int32_t)(so - s0);
                m.pad = 0;
                out_ch[chain_off[r] + k] = m;
                for (int j = c.head; j >= 0; j = S[j].next) { meme_chain_seed s; s.rbeg = S[j].rbeg; s.qbeg = S[j].qbeg; s.len = S[j].len; out_sd[so++] = s; }
            }
        }
    }
}

// Before tier 1 runs: the hits every read has to walk (sum over its SMEMs of min(hitcount, max_occ)); reads beyond `light` hits go to the
// LDS tier of 256 chains, beyond `heavy` (256) to the one of 512, beyond `huge` (512) to the one of 1 024, at once and concurrently with
// tier 1 (a read cannot make more chains than it has hits).  One atomic per wavefront and list.
__global__ void __launch_bounds__(256) k_chain_route(const meme_mem_tl* __restrict__ smems, const i64* __restrict__ smem_off, const i64* __restrict__ read_off, i64 nreads,
                                                      int max_occ, int min_seed_len, i64 light, i64 heavy, i64 huge, unsigned char* __restrict__ cls,
                                                      unsigned long long* __restrict__ counts, i64* __restrict__ listA, i64* __restrict__ workA,
                                                      i64* __restrict__ listL, i64* __restrict__ workL, i64* __restrict__ listH, i64* __restrict__ workH) {
    const int lane = threadIdx.x & 63;
    for (i64 r0 = ((i64)blockIdx.x * blockDim.x + threadIdx.x) - lane; r0 < nreads; r0 += (i64)gridDim.x * blockDim.x) {
        const i64 r = r0 + lane;
        i64 work = 0;
        int c = 0;
        if (r < nreads && read_off[r + 1] - read_off[r] >= min_seed_len) {
            for (i64 k = smem_off[r]; k < smem_off[r + 1]; ++k) { const int h = smems[k].hitcount; work += h < max_occ ? h : max_occ; }
            c = work > huge ? 3 : work > heavy ? 2 : (work > light ? 1 : 0);
        }
        if (r < nreads) cls[r] = (unsigned char)c;
        for (int which = 1; which <= 3; ++which) {
            const u64 m = __ballot(c == which);
            if (!m) continue;
            unsigned long long b = 0;
            if (lane == __builtin_ctzll(m)) b = atomicAdd(&counts[which - 1], (unsigned long long)__popcll(m));
            b = __shfl(b, __builtin_ctzll(m));
            if (c == which) {
                const unsigned long long k = b + (unsigned long long)__popcll(m & (((u64)1 << lane) - 1));
                (which == 1 ? listA : which == 2 ? listL : listH)[k] = r;
                (which == 1 ? workA : which == 2 ? workL : workH)[k] = work;
            }
        }
    }
}

// reads the first tier left (fallback == 1): their indices and work, for the second tier
__global__ void __launch_bounds__(256) k_chain_redo(const ReadHdr* __restrict__ hdr, i64 nreads, int want, const unsigned char* __restrict__ skip,
                                                     unsigned long long* __restrict__ count, i64* __restrict__ list, i64* __restrict__ lwork) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += (i64)gridDim.x * blockDim.x)
        if (!(skip && skip[r]) && hdr[r].fallback == want) { const unsigned long long k = atomicAdd(count, 1ull); list[k] = r; lwork[k] = hdr[r].work > 0 ? hdr[r].work : 1; }
}

__global__ void __launch_bounds__(256) k_chain_counts(const ReadHdr* __restrict__ hdr, i64 nreads, i64* __restrict__ nch, i64* __restrict__ nsd,
                                                       int* __restrict__ tree, unsigned char* __restrict__ fb) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += (i64)gridDim.x * blockDim.x) {
        const ReadHdr H = hdr[r];
        nch[r] = H.fallback ? 0 : H.n_kept; nsd[r] = H.fallback ? 0 : H.n_seeds; tree[r] = H.tree_size; fb[r] = H.fallback ? 1 : 0;
    }
}

unsigned blocks_of(i64 items, int per) { i64 b = (items + per - 1) / per; const i64 cap = 256 * 64; return (unsigned)(b < cap ? (b < 1 ? 1 : b) : cap); }

}  // namespace

// Chains of the batch the ctx has just seeded, left in HBM: ctx->chain[5] = {chain_off[n+1], seed_off[n+1], nch[n+1], nsd[n+1], tree[n], fb[n]},
// [6] = packed meme_chain, [7] = packed meme_chain_seed, [3] = frac_rep.  totals[0..1] = chains, seeds.
//
// Order of events: k_chain_route sends reads to the wavefront tiers by the number of hits they have to walk; then the lane-per-read tier
// (everything else, main stream) and the three sizes of the LDS tier (> chain_light_hits, > 256, > 512 hits), each on a side stream of its
// own, run CONCURRENTLY; the reads the lane tier could not hold go through the LDS tier of 256 chains afterwards; whatever is left with
// fallback = 3 (two chains on one position; more than 1 024 chains) goes to the B-tree tier.  Each of the five wavefront launches has a
// scratch set of its own.
namespace {
struct ScratchSet { DevBuf* buf; WaveArgs W; };
int make_set(meme_ctx* ctx, DevBuf& buf, const i64* d_list, const i64* d_woff, i64 nlist, i64 total_work, int set, bool with_tree, WaveArgs* out) {
    const size_t units = (size_t)total_work + 8, nodes = with_tree ? (size_t)total_work / 3 + 4 * (size_t)nlist + 8 : 1;
    const size_t sz[7] = {units * sizeof(C2), units * sizeof(S2), units * sizeof(FRec), with_tree ? units * 8 : 8, units * 4, with_tree ? units * 4 : 8, nodes * sizeof(TNode)};
    size_t need = 0, at[7];
    for (int k = 0; k < 7; ++k) { at[k] = need; need += (sz[k] + 255) / 256 * 256; }
    if (need > buf.cap) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b / 2 + buf.cap) {
            meme_set_error("chaining %lld repeat-rich reads of this batch (%lld hits to walk) needs %.1f GB of scratch, more than half of the free HBM "
                           "(%.1f GB): chain this batch in smaller pieces", (long long)nlist, (long long)total_work, need / 1e9, free_b / 1e9);
            return MEME_E_CAPACITY;
        }
    }
    int rc = meme_buf_reserve(ctx, buf, need);
    if (rc) return rc;
    unsigned char* p = (unsigned char*)buf.p;
    WaveArgs W;
    memset(&W, 0, sizeof(W));
    W.list = d_list; W.woff = d_woff; W.nlist = nlist; W.sub = nullptr; W.nsub = 0; W.set = set;
    W.C = (C2*)(p + at[0]); W.S = (S2*)(p + at[1]); W.F = (FRec*)(p + at[2]); W.srt = (u64*)(p + at[3]); W.ia = (int*)(p + at[4]);
    W.ib = (int*)(p + at[5]); W.nodes = (TNode*)(p + at[6]);
    *out = W;
    return MEME_OK;
}
}  // namespace

int meme_chain_run(meme_ctx* ctx, const meme_contig* contigs, int32_t n_contigs, const meme_chain_opt* opt, i64* totals) {
    const i64 n = ctx->last_seed_reads;
    int rc;
    for (int i = 0; i < n_contigs; ++i)            // bntann1_t: 64-bit offset, 32-bit length; ascending, inside the forward strand
        if (contigs[i].len < 1 || contigs[i].offset < 0 || contigs[i].offset + contigs[i].len > opt->l_pac || (i > 0 && contigs[i].offset < contigs[i - 1].offset + contigs[i - 1].len)) {
            meme_set_error("contig %d (offset %lld, length %d) is not a valid reference sequence of a %lld-base genome", i, (long long)contigs[i].offset, contigs[i].len,
                           (long long)opt->l_pac);
            return MEME_E_ARG;
        }
    // however this function is left (the error returns below included), the kernels on the side streams have finished: a caller that
    // retries with a smaller batch or destroys the ctx must not race them
    struct SideGuard { meme_ctx* c; ~SideGuard() { for (auto st : c->stream_side) if (st) (void)hipStreamSynchronize(st); } } side_guard{ctx};
    DevBuf* B = ctx->chain;     // 0 tier-1 chains, 1 tier-1 seeds, 2 headers, 3 frac, 4 contig table, 5 counts/offsets, 6 packed chains, 7 packed seeds,
                                // 8 lists + work + offsets of the five wavefront launches + read classes, 9 .. 13 their scratch sets
    if ((rc = meme_buf_reserve(ctx, B[0], (size_t)((n + 63) / 64 * 64) * CHAIN_CAP * sizeof(DChain)))) return rc;
    if ((rc = meme_buf_reserve(ctx, B[1], (size_t)((n + 63) / 64 * 64) * CHAIN_CAP * SEED_CAP * sizeof(DSeed)))) return rc;
    if ((rc = meme_buf_reserve(ctx, B[2], (size_t)n * sizeof(ReadHdr)))) return rc;
    if ((rc = meme_buf_reserve(ctx, B[3], (size_t)n * sizeof(float)))) return rc;
    const size_t ctab = (size_t)n_contigs * (8 + 4 + 1) + 64;
    if ((rc = meme_buf_reserve(ctx, B[4], ctab))) return rc;
    const size_t cnt_bytes = ((size_t)(n + 1) * 8 * 4 + (size_t)n * 4 + (size_t)n + 64 + 15) / 16 * 16 + 64;   // (+ five counters at the end)
    if ((rc = meme_buf_reserve(ctx, B[5], cnt_bytes))) return rc;
    if ((rc = meme_buf_reserve(ctx, B[8], (size_t)(n + 1) * 8 * 15 + (size_t)n + 64))) return rc;
    std::vector<unsigned char> tab(ctab, 0);
    i64* t_off = (i64*)tab.data();
    int* t_len = (int*)(tab.data() + (size_t)n_contigs * 8);
    unsigned char* t_alt = tab.data() + (size_t)n_contigs * 12;
    for (int i = 0; i < n_contigs; ++i) { t_off[i] = contigs[i].offset; t_len[i] = contigs[i].len; t_alt[i] = contigs[i].is_alt ? 1 : 0; }
    HIP_TRY(hipMemcpyAsync(B[4].p, tab.data(), ctab, hipMemcpyHostToDevice, ctx->stream));
    unsigned long long* d_cnt4 = (unsigned long long*)((unsigned char*)B[5].p + cnt_bytes - 64);
    HIP_TRY(hipMemsetAsync(d_cnt4, 0, 64, ctx->stream));
    ChainArgs A;
    A.smems = (const meme_mem_tl*)ctx->smems.p; A.smem_off = (const i64*)ctx->smem_off.p;
    A.hits = (const u64*)ctx->hits.p; A.hit_off = (const i64*)ctx->hit_off.p; A.read_off = (const i64*)ctx->read_off.p;
    A.nreads = n;
    A.contig_off = (const i64*)B[4].p; A.contig_len = (const int*)((unsigned char*)B[4].p + (size_t)n_contigs * 8);
    A.contig_alt = (const unsigned char*)B[4].p + (size_t)n_contigs * 12; A.n_contigs = n_contigs;
    A.o = *opt;
    A.hit_cap1 = (int)ctx->chain_lane_hits;
    A.ch = (DChain*)B[0].p; A.sd = (DSeed*)B[1].p; A.hdr = (ReadHdr*)B[2].p; A.frac_rep = (float*)B[3].p;
    // lists: [k] list, work, offsets for k = 0 / 1 / 2 LDS tier of 256 / 512 / 1 024 chains (routed), 3 LDS tier of 256 chains (left by the
    // lane tier), 4 B-tree tier; scratch set k + 1 in B[9 + k]
    i64* L8 = (i64*)B[8].p;
    i64 *d_list[5], *d_work[5], *d_woff[5];
    for (int k = 0; k < 5; ++k) { d_list[k] = L8 + (size_t)(3 * k) * (size_t)(n + 1); d_work[k] = d_list[k] + (n + 1); d_woff[k] = d_work[k] + (n + 1); }
    unsigned char* d_cls = (unsigned char*)(L8 + (size_t)15 * (size_t)(n + 1));
    const bool wave_tiers = ctx->chain_wave_tiers != 0;     // (0: tests drive everything the lane tier leaves through the B-tree tier)
    A.cls = d_cls;
    hipEvent_t* ev = ctx->ev_chain;
    for (int i = 0; i < 5; ++i) if (!ev[i]) HIP_TRY(hipEventCreate(&ev[i]));
    for (int i = 0; i < 3; ++i) {
        { const int src = meme_side_stream(ctx, i); if (src) return src; }     // (tuning "chain_side_priority": the routed tiers' streams above the lane tier's in the hardware queues)
        if (!ctx->ev_side[i]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_side[i], hipEventDisableTiming));
    }
    if (!ctx->ev_aux) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_aux, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev[0], ctx->stream));
    // ---- route by work
    const i64 never = (i64)1 << 60;
    const i64 light = wave_tiers ? ctx->chain_light_hits : never, heavy = wave_tiers ? 256 : never, huge = wave_tiers ? 512 : never;
    hipLaunchKernelGGL(k_chain_route, dim3(blocks_of(n, 256)), dim3(256), 0, ctx->stream, A.smems, A.smem_off, A.read_off, n, opt->max_occ, opt->min_seed_len, light, heavy, huge,
                       d_cls, d_cnt4, d_list[0], d_work[0], d_list[1], d_work[1], d_list[2], d_work[2]);
    unsigned long long h_cnt[5] = {0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt4, 24, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));                           // (also: `tab` is a local)
    i64 nl[5] = {(i64)h_cnt[0], (i64)h_cnt[1], (i64)h_cnt[2], 0, 0}, tw[5] = {0, 0, 0, 0, 0};
    WaveArgs W[5];
    memset(W, 0, sizeof(W));
    bool used[5] = {false, false, false, false, false};
    for (int k = 0; k < 3; ++k) if (nl[k] > 0) { if ((rc = meme_scan_exclusive(ctx, d_work[k], d_woff[k], nl[k]))) return rc; HIP_TRY(hipMemcpyAsync(&tw[k], d_woff[k] + nl[k], 8, hipMemcpyDeviceToHost, ctx->stream)); }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 3; ++k) if (nl[k] > 0) { if ((rc = make_set(ctx, B[9 + k], d_list[k], d_woff[k], nl[k], tw[k], 1 + k, false, &W[k]))) return rc; used[k] = true; }
    HIP_TRY(hipEventRecord(ctx->ev_aux, ctx->stream));
    // ---- four launches at once (the longest-running first)
    for (int k = 2; k >= 0; --k) {
        if (nl[k] <= 0) continue;
        HIP_TRY(hipStreamWaitEvent(ctx->stream_side[k], ctx->ev_aux, 0));
        if (k == 2) hipLaunchKernelGGL((k_chain_lds<1024>), dim3((unsigned)nl[k]), dim3(64), 0, ctx->stream_side[k], A, W[k]);
        else if (k == 1) hipLaunchKernelGGL((k_chain_lds<512>), dim3((unsigned)nl[k]), dim3(64), 0, ctx->stream_side[k], A, W[k]);
        else hipLaunchKernelGGL((k_chain_lds<256>), dim3((unsigned)nl[k]), dim3(64), 0, ctx->stream_side[k], A, W[k]);
        HIP_TRY(hipEventRecord(ctx->ev_side[k], ctx->stream_side[k]));
    }
    hipLaunchKernelGGL((k_chain<CHAIN_CAP, SEED_CAP>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, A);
    HIP_TRY(hipEventRecord(ev[1], ctx->stream));
    // ---- what the lane tier left: LDS tier of 256 chains (or, with that switched off, the B-tree tier)
    hipLaunchKernelGGL(k_chain_redo, dim3(blocks_of(n, 256)), dim3(256), 0, ctx->stream, (const ReadHdr*)B[2].p, n, 1, (const unsigned char*)d_cls, d_cnt4 + 3, d_list[3], d_work[3]);   // (routed reads: their tiers may still be writing)
    HIP_TRY(hipMemcpyAsync(&h_cnt[3], d_cnt4 + 3, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    nl[3] = (i64)h_cnt[3];
    if (nl[3] > 0) {
        if ((rc = meme_scan_exclusive(ctx, d_work[3], d_woff[3], nl[3]))) return rc;
        HIP_TRY(hipMemcpyAsync(&tw[3], d_woff[3] + nl[3], 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if ((rc = make_set(ctx, B[12], d_list[3], d_woff[3], nl[3], tw[3], 4, !wave_tiers, &W[3]))) return rc;
        used[3] = true;
        if (wave_tiers) hipLaunchKernelGGL((k_chain_lds<256>), dim3((unsigned)nl[3]), dim3(64), 0, ctx->stream, A, W[3]);
        else hipLaunchKernelGGL((k_chain_wave<288, 2048>), dim3((unsigned)nl[3]), dim3(64), 0, ctx->stream, A, W[3], (i64)0);
    }
    for (int k = 0; k < 3; ++k) if (nl[k] > 0) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_side[k], 0));
    HIP_TRY(hipEventRecord(ev[4], ctx->stream));
    // ---- the B-tree tier for what is left (fallback == 3)
    hipLaunchKernelGGL(k_chain_redo, dim3(blocks_of(n, 256)), dim3(256), 0, ctx->stream, (const ReadHdr*)B[2].p, n, 3, (const unsigned char*)nullptr, d_cnt4 + 4, d_list[4], d_work[4]);
    HIP_TRY(hipMemcpyAsync(&h_cnt[4], d_cnt4 + 4, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    nl[4] = (i64)h_cnt[4];
    if (nl[4] > 0) {
        if ((rc = meme_scan_exclusive(ctx, d_work[4], d_woff[4], nl[4]))) return rc;
        HIP_TRY(hipMemcpyAsync(&tw[4], d_woff[4] + nl[4], 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if ((rc = make_set(ctx, B[13], d_list[4], d_woff[4], nl[4], tw[4], 5, true, &W[4]))) return rc;
        used[4] = true;
        hipLaunchKernelGGL((k_chain_wave<288, 2048>), dim3((unsigned)nl[4]), dim3(64), 0, ctx->stream, A, W[4], (i64)0);
    }
    HIP_TRY(hipEventRecord(ev[2], ctx->stream));
    i64* d_choff = (i64*)B[5].p;
    i64* d_sdoff = d_choff + (n + 1);
    i64* d_nch = d_sdoff + (n + 1);
    i64* d_nsd = d_nch + (n + 1);
    int* d_tree = (int*)(d_nsd + (n + 1));
    unsigned char* d_fb = (unsigned char*)(d_tree + n);
    hipLaunchKernelGGL(k_chain_counts, dim3(blocks_of(n, 256)), dim3(256), 0, ctx->stream, (const ReadHdr*)B[2].p, n, d_nch, d_nsd, d_tree, d_fb);
    if ((rc = meme_scan_exclusive(ctx, d_nch, d_choff, n))) return rc;
    i64 tot[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&tot[0], d_choff + n, 8, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = meme_scan_exclusive(ctx, d_nsd, d_sdoff, n))) return rc;
    HIP_TRY(hipMemcpyAsync(&tot[1], d_sdoff + n, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if ((rc = meme_buf_reserve(ctx, B[6], (size_t)(tot[0] + 1) * sizeof(meme_chain)))) return rc;
    if ((rc = meme_buf_reserve(ctx, B[7], (size_t)(tot[1] + 1) * sizeof(meme_chain_seed)))) return rc;
    PackSets PS;
    memset(&PS, 0, sizeof(PS));
    for (int k = 0; k < 5; ++k) if (used[k]) { PS.woff[k + 1] = W[k].woff; PS.C[k + 1] = W[k].C; PS.S[k + 1] = W[k].S; PS.F[k + 1] = W[k].F; }
    hipLaunchKernelGGL(k_chain_pack, dim3(blocks_of(n, 256)), dim3(256), 0, ctx->stream, (const DChain*)B[0].p, (const DSeed*)B[1].p, PS,
                       (const ReadHdr*)B[2].p, (const i64*)d_choff, (const i64*)d_sdoff, n, (meme_chain*)B[6].p, (meme_chain_seed*)B[7].p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev[3], ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    {
        float ms = 0.f, ms1 = 0.f, ms3 = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[3]));
        HIP_TRY(hipEventElapsedTime(&ms1, ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&ms3, ev[4], ev[2]));
        ctx->tm.chain_kernel_ms = ms;          // (includes the small host round trips between the tiers)
        ctx->tm.chain_pass2_ms = ms - ms1;     // everything after the lane-per-read tier has finished (the routed tiers run beside it)
        ctx->tm.chain_tier3_ms = ms3;
        ctx->tm.chain_tier2_reads = nl[0] + nl[1] + nl[2] + nl[3];
        ctx->tm.chain_tier3_reads = nl[4];
    }
    ctx->chain_reads = n;
    ctx->chain_tier2_reads = nl[0] + nl[1] + nl[2] + nl[3];
    ctx->chain_tier3_reads = nl[4];
    totals[0] = tot[0]; totals[1] = tot[1];
    return MEME_OK;
}

#ifdef MEME_CHAIN_PROF
extern "C" int meme_debug_chain_prof(unsigned long long* out, int reset) {        // (investigation builds only: cycles per phase, summed over wavefronts)
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

extern "C" int meme_chain_last_batch_host(meme_ctx* ctx, const meme_contig* contigs, int32_t n_contigs, const meme_chain_opt* opt,
                                          meme_chain_host_result* out) {
    if (!ctx || !contigs || n_contigs < 1 || !opt || !out) { meme_set_error("meme_chain_last_batch_host: null argument"); return MEME_E_ARG; }
    if (opt->max_occ < 1 || opt->l_pac < 1) { meme_set_error("meme_chain_last_batch_host: bad options"); return MEME_E_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    memset(out, 0, sizeof(*out));
    const i64 n = ctx->last_seed_reads;
    if (n <= 0 || !ctx->smem_off.p || !ctx->read_off.p) { meme_set_error("meme_chain_last_batch_host: no seeded batch on this ctx"); return MEME_E_STATE; }
    int rc;
    i64 tot[2];
    if ((rc = meme_chain_run(ctx, contigs, n_contigs, opt, tot))) return rc;
    DevBuf* B = ctx->chain;
    const i64* d_choff = (const i64*)B[5].p;
    const i64* d_sdoff = d_choff + (n + 1);
    const int* d_tree = (const int*)(d_sdoff + 3 * (n + 1));
    const unsigned char* d_fb = (const unsigned char*)(d_tree + n);
    meme_ctx::HostBuf* Hb = ctx->h_chain;   // 0 chain_off, 1 chains, 2 seed_off, 3 seeds, 4 tree sizes, 5 frac_rep, 6 fallback flags
    if ((rc = meme_hostbuf_reserve(ctx, Hb[0], (size_t)(n + 1) * 8)) || (rc = meme_hostbuf_reserve(ctx, Hb[1], (size_t)(tot[0] + 1) * sizeof(meme_chain))) ||
        (rc = meme_hostbuf_reserve(ctx, Hb[2], (size_t)(n + 1) * 8)) || (rc = meme_hostbuf_reserve(ctx, Hb[3], (size_t)(tot[1] + 1) * sizeof(meme_chain_seed))) ||
        (rc = meme_hostbuf_reserve(ctx, Hb[4], (size_t)n * 4)) || (rc = meme_hostbuf_reserve(ctx, Hb[5], (size_t)n * 4)) ||
        (rc = meme_hostbuf_reserve(ctx, Hb[6], (size_t)n))) return rc;
    HIP_TRY(hipMemcpyAsync(Hb[0].p, d_choff, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(Hb[2].p, d_sdoff, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (tot[0]) HIP_TRY(hipMemcpyAsync(Hb[1].p, B[6].p, (size_t)tot[0] * sizeof(meme_chain), hipMemcpyDeviceToHost, ctx->stream));
    if (tot[1]) HIP_TRY(hipMemcpyAsync(Hb[3].p, B[7].p, (size_t)tot[1] * sizeof(meme_chain_seed), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(Hb[4].p, d_tree, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(Hb[5].p, B[3].p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(Hb[6].p, d_fb, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    out->nreads = n;
    out->chain_off = (const int64_t*)Hb[0].p; out->chains = (const meme_chain*)Hb[1].p;
    out->seed_off = (const int64_t*)Hb[2].p; out->seeds = (const meme_chain_seed*)Hb[3].p;
    out->tree_size = (const int32_t*)Hb[4].p; out->frac_rep = (const float*)Hb[5].p; out->fallback = (const uint8_t*)Hb[6].p;
    out->total_chains = tot[0]; out->total_seeds = tot[1];
    i64 nfb = 0;
    for (i64 i = 0; i < n; ++i) nfb += out->fallback[i] ? 1 : 0;
    out->n_fallback = nfb;                 // always 0: both tiers run on the device
    out->n_tier2 = ctx->chain_tier2_reads;
    return MEME_OK;
}

// The same for seeds the caller brings (host arrays in the layout meme_seed_batch_host returns them in): any seeder's SMEMs and
// hits can be chained, and the parity tests feed made-up seed sets (chains at equal positions, thousands of chains) through here.
extern "C" int meme_chain_batch_host(meme_ctx* ctx, const meme_mem_tl* smems, const int64_t* smem_off, const uint64_t* hits, const int64_t* hit_off,
                                     const int32_t* read_len, int64_t nreads, const meme_contig* contigs, int32_t n_contigs, const meme_chain_opt* opt,
                                     meme_chain_host_result* out) {
    if (!ctx || !smem_off || !hit_off || !read_len || nreads < 1 || !out) { meme_set_error("meme_chain_batch_host: bad argument"); return MEME_E_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    const i64 ns = smem_off[nreads], nh = hit_off[nreads];
    if (smem_off[0] != 0 || hit_off[0] != 0 || ns < 0 || nh < 0 || (ns > 0 && !smems) || (nh > 0 && !hits)) { meme_set_error("meme_chain_batch_host: bad offsets"); return MEME_E_ARG; }
    int rc;
    if ((rc = meme_buf_reserve(ctx, ctx->smems, (size_t)(ns + 1) * sizeof(meme_mem_tl))) || (rc = meme_buf_reserve(ctx, ctx->hits, (size_t)(nh + 1) * 8)) ||
        (rc = meme_buf_reserve(ctx, ctx->smem_off, (size_t)(nreads + 1) * 8)) || (rc = meme_buf_reserve(ctx, ctx->hit_off, (size_t)(nreads + 1) * 8)) ||
        (rc = meme_buf_reserve(ctx, ctx->read_off, (size_t)(nreads + 1) * 8))) return rc;
    std::vector<i64> roff((size_t)nreads + 1, 0);
    for (i64 i = 0; i < nreads; ++i) roff[(size_t)i + 1] = roff[(size_t)i] + read_len[i];
    if (ns) HIP_TRY(hipMemcpyAsync(ctx->smems.p, smems, (size_t)ns * sizeof(meme_mem_tl), hipMemcpyHostToDevice, ctx->stream));
    if (nh) HIP_TRY(hipMemcpyAsync(ctx->hits.p, hits, (size_t)nh * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->smem_off.p, smem_off, (size_t)(nreads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->hit_off.p, hit_off, (size_t)(nreads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->read_off.p, roff.data(), (size_t)(nreads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->last_seed_reads = nreads;
    ctx->reads_resident = false;       // (the seeds are the caller's; whatever bases an earlier seeding call left here do not belong to them)
    ctx->last_seed_max_len = 0;
    for (i64 i = 0; i < nreads; ++i) ctx->last_seed_max_len = read_len[i] > ctx->last_seed_max_len ? read_len[i] : ctx->last_seed_max_len;
    return meme_chain_last_batch_host(ctx, contigs, n_contigs, opt, out);
}
