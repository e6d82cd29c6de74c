// Chaining on the device: mem_chain_Learned + mem_chain_flt (reference src/bwamem.cpp:1122-1204, 599-717) for every read
// of the batch the ctx has just seeded -- the seeds never leave HBM between the two stages.
//
// One lane per read (the work per read is small and strictly sequential: every hit is tested against the chain with the
// closest position below it and either merged into it or becomes a new chain).  What the reference keeps in a B-tree keyed
// by position is a position-sorted array of chains in a per-read scratch here; what it sorts with klib's introsort is sorted by
// the same sequence of comparisons and swaps, because chains of equal weight keep whatever order that algorithm leaves them in
// and the filter that follows depends on it.
//
// Reads that fit neither scratch size (more than 128 chains, a chain of more than 32 seeds, more than SMEM_CAP SMEMs) and reads
// that would insert two chains at the same position (the B-tree's order of equal keys is an implementation detail) are flagged:
// the caller chains those on the host with the reference's own functions.
#include <hipcub/hipcub.hpp>

#include "meme_common.h"

namespace {

// Two passes with different scratch sizes: every read with room for 16 chains of 8 seeds (2.5 KB per read); the few per cent
// that need more (repeats) again, alone, with room for 128 chains of 32 seeds (68 KB per read).
constexpr int CHAIN_CAP = 16, SEED_CAP = 8;
constexpr int CHAIN_CAP2 = 128, SEED_CAP2 = 32;
constexpr int SMEM_CAP = 256;      // SMEMs per read (the sorted walk is quadratic in it)
constexpr int HIT_CAP = 8192;      // hits per read that one lane is asked to walk

struct DChain {                    // 32 bytes
    i64 pos;
    int rid, n, w, first;
    short kept, is_alt;
    int row;                       // which seed row of the read's scratch holds its seeds
};
struct DSeed { i64 rbeg; int qbeg, len; };
struct ReadHdr { int tree_size, n_kept, n_seeds, fallback; i64 slot; };   // fallback: 0 done, 1 needs the bigger scratch / the host; slot: scratch index, bit 62 = second pass

struct ChainArgs {
    const meme_mem_tl* smems; const i64* smem_off; const u64* hits; const i64* hit_off; const i64* read_off;
    i64 nreads;
    const i64* contig_off; const int* contig_len; const unsigned char* contig_alt; int n_contigs;
    meme_chain_opt o;
    DChain* ch; DSeed* sd; ReadHdr* hdr; float* frac_rep;
    const i64* list; i64 nlist;        // second pass: the reads to redo (nullptr: all reads)
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

__device__ inline int chain_weight(const DChain& c, const DSeed* row) {   // mem_chain_weight, src/bwamem.cpp:522-541
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
__device__ inline void insert_sort(DChain* s, DChain* t) {
    for (DChain* i = s + 1; i < t; ++i)
        for (DChain* j = i; j > s && FLT_LT(*j, *(j - 1)); --j) swap_chain(*j, *(j - 1));
}
__device__ void comb_sort(int n, DChain* a) {
    const double shrink = 1.2473309501039786540366528676643;
    bool do_swap;
    int gap = n;
    do {
        if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
        do_swap = false;
        for (DChain* i = a; i < a + n - gap; ++i) { DChain* j = i + gap; if (FLT_LT(*j, *i)) { swap_chain(*i, *j); do_swap = true; } }
    } while (do_swap || gap > 2);
    if (gap != 1) insert_sort(a, a + n);
}
__device__ void sort_by_weight(DChain* a, int n) {
    if (n < 1) return;
    if (n == 2) { if (FLT_LT(a[1], a[0])) swap_chain(a[0], a[1]); return; }
    struct { DChain *left, *right; int depth; } stack[40], *top = stack;
    int d;
    for (d = 2; (1 << d) < n; ++d) {}
    DChain *s = a, *t = a + (n - 1);
    d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) { comb_sort((int)(t - s) + 1, s); t = s; continue; }
            DChain *i = s, *j = t, *k = i + ((j - i) >> 1) + 1;
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
    const i64 tid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (A.list ? A.nlist : A.nreads)) return;
    const i64 r = A.list ? A.list[tid] : tid;
    const meme_chain_opt& o = A.o;
    const meme_mem_tl* sm = A.smems + A.smem_off[r];
    const int ns = (int)(A.smem_off[r + 1] - A.smem_off[r]);
    const u64* ht = A.hits + A.hit_off[r];
    const int len = (int)(A.read_off[r + 1] - A.read_off[r]);
    DChain* ch = A.ch + tid * CC;
    DSeed* sd = A.sd + tid * (CC * SC);
    ReadHdr H = {0, 0, 0, 0, tid | (A.list ? (i64)1 << 62 : 0)};
    float frac = 0.f;
    int nc = 0;
    if (len >= o.min_seed_len && ns > 0) {                                // (:1138)
        if (ns > SMEM_CAP) H.fallback = 2;                                // the bigger scratch does not help: host
        // One lane walks a read's hits one after the other: a read with tens of thousands of them (hundreds of repeat SMEMs of 500 hits)
        // would hold its wavefront -- and with it the kernel -- for milliseconds, where a host core needs a fraction of one.
        i64 work = 0;
        for (int i = 0; i < ns && i < SMEM_CAP; ++i) work += sm[i].hitcount < o.max_occ ? sm[i].hitcount : o.max_occ;
        if (work > HIT_CAP) H.fallback = 2;
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
                    DSeed* row = sd + c.row * SC;
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
                    if (lower >= 0 && ch[lower].pos == s.rbeg) { H.fallback = 2; break; }   // equal B-tree keys: the host decides
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
            if (c.w >= o.min_chain_weight) ch[n++] = c;
        }
        if (n > 0) {
            sort_by_weight(ch, n);
            int kept_idx[CC];
            int nk = 0;
            ch[0].kept = 3;
            kept_idx[nk++] = 0;
            for (int i = 1; i < n; ++i) {
                bool large_ovlp = false;
                int k = 0;
                const DSeed* ri = sd + ch[i].row * SC;
                const int beg_i = ri[0].qbeg, end_i = ri[ch[i].n - 1].qbeg + ri[ch[i].n - 1].len;
                for (; k < nk; ++k) {
                    const int j = kept_idx[k];
                    const DSeed* rj = sd + ch[j].row * SC;
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

// the kept chains and their seeds, densely packed in read order (the scratch of the pass that finished the read)
__global__ void __launch_bounds__(256) k_chain_pack(const DChain* __restrict__ ch1, const DSeed* __restrict__ sd1, const DChain* __restrict__ ch2,
                                                     const DSeed* __restrict__ sd2, const ReadHdr* __restrict__ hdr,
                                                     const i64* __restrict__ chain_off, const i64* __restrict__ seed_off, i64 nreads,
                                                     meme_chain* __restrict__ out_ch, meme_chain_seed* __restrict__ out_sd) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += (i64)gridDim.x * blockDim.x) {
        const ReadHdr H = hdr[r];
        if (H.fallback || H.n_kept == 0) continue;
        const bool second = (H.slot >> 62) & 1;
        const i64 slot = H.slot & (((i64)1 << 62) - 1);
        const int cc = second ? CHAIN_CAP2 : CHAIN_CAP, sc = second ? SEED_CAP2 : SEED_CAP;
        const DChain* ch = (second ? ch2 : ch1) + slot * cc;
        const DSeed* sd = (second ? sd2 : sd1) + slot * (i64)cc * sc;
        i64 so = seed_off[r];
        const i64 s0 = so;
        for (int k = 0; k < H.n_kept; ++k) {
            const DChain c = ch[k];
            meme_chain m;
            m.pos = c.pos; m.rid = c.rid; m.n_seeds = c.n; m.w = c.w; m.first = c.first; m.kept = c.kept; m.is_alt = c.is_alt;
            m.seed_beg = (int32_t)(so - s0);
            m.pad = 0;
            out_ch[chain_off[r] + k] = m;
            const DSeed* row = sd + c.row * sc;
            for (int j = 0; j < c.n; ++j) { meme_chain_seed s; s.rbeg = row[j].rbeg; s.qbeg = row[j].qbeg; s.len = row[j].len; out_sd[so++] = s; }
        }
    }
}

// reads the first pass could not hold (fallback == 1): their indices, for the second pass
__global__ void __launch_bounds__(256) k_chain_redo(const ReadHdr* __restrict__ hdr, i64 nreads, unsigned long long* __restrict__ count, i64* __restrict__ list) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += (i64)gridDim.x * blockDim.x)
        if (hdr[r].fallback == 1) list[atomicAdd(count, 1ull)] = r;
}

__global__ void __launch_bounds__(256) k_chain_counts(const ReadHdr* __restrict__ hdr, i64 nreads, i64* __restrict__ nch, i64* __restrict__ nsd,
                                                       int* __restrict__ tree, unsigned char* __restrict__ fb) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r <= nreads; r += (i64)gridDim.x * blockDim.x) {
        if (r == nreads) { nch[r] = 0; nsd[r] = 0; continue; }
        const ReadHdr H = hdr[r];
        nch[r] = H.fallback ? 0 : H.n_kept; nsd[r] = H.fallback ? 0 : H.n_seeds; tree[r] = H.tree_size; fb[r] = H.fallback ? 1 : 0;
    }
}

unsigned blocks_of(i64 items, int per) { i64 b = (items + per - 1) / per; const i64 cap = 256 * 64; return (unsigned)(b < cap ? (b < 1 ? 1 : b) : cap); }

}  // namespace

extern "C" int meme_chain_last_batch_host(meme_ctx* ctx, const meme_contig* contigs, int32_t n_contigs, const meme_chain_opt* opt,
                                          meme_chain_host_result* out) {
    if (!ctx || !contigs || n_contigs < 1 || !opt || !out) { meme_set_error("meme_chain_last_batch_host: null argument"); return MEME_E_ARG; }
    if (opt->max_occ < 1 || opt->l_pac < 1) { meme_set_error("meme_chain_last_batch_host: bad options"); return MEME_E_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    memset(out, 0, sizeof(*out));
    const i64 n = ctx->last_seed_reads;
    if (n <= 0 || !ctx->smem_off.p || !ctx->read_off.p) { meme_set_error("meme_chain_last_batch_host: no seeded batch on this ctx"); return MEME_E_STATE; }
    int rc;
    DevBuf* B = ctx->chain;     // 0 chains scratch, 1 seeds scratch, 2 headers, 3 frac, 4 contig table, 5 counts/offsets, 6 packed chains, 7 packed seeds,
                                // 8 redo list, 9 / 10 scratch of the second pass
    if ((rc = meme_buf_reserve(ctx, B[0], (size_t)n * CHAIN_CAP * sizeof(DChain)))) return rc;
    if ((rc = meme_buf_reserve(ctx, B[1], (size_t)n * CHAIN_CAP * SEED_CAP * sizeof(DSeed)))) return rc;
    if ((rc = meme_buf_reserve(ctx, B[2], (size_t)n * sizeof(ReadHdr)))) return rc;
    if ((rc = meme_buf_reserve(ctx, B[3], (size_t)n * sizeof(float)))) return rc;
    const size_t ctab = (size_t)n_contigs * (8 + 4 + 1) + 64;
    if ((rc = meme_buf_reserve(ctx, B[4], ctab))) return rc;
    // counts, their scans, tree sizes, fallback flags
    const size_t cnt_bytes = ((size_t)(n + 1) * 8 * 4 + (size_t)n * 4 + (size_t)n + 64 + 15) / 16 * 16 + 16;   // (+ the redo counter at the end)
    if ((rc = meme_buf_reserve(ctx, B[5], cnt_bytes))) return rc;
    // contig table: offsets | lengths | alt flags
    std::vector<unsigned char> tab(ctab, 0);
    i64* t_off = (i64*)tab.data();
    int* t_len = (int*)(tab.data() + (size_t)n_contigs * 8);
    unsigned char* t_alt = tab.data() + (size_t)n_contigs * 12;
    for (int i = 0; i < n_contigs; ++i) { t_off[i] = contigs[i].offset; t_len[i] = contigs[i].len; t_alt[i] = contigs[i].is_alt ? 1 : 0; }
    HIP_TRY(hipMemcpyAsync(B[4].p, tab.data(), ctab, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));                           // `tab` goes out of scope with the function only, but keep it simple
    ChainArgs A;
    A.smems = (const meme_mem_tl*)ctx->smems.p; A.smem_off = (const i64*)ctx->smem_off.p;
    A.hits = (const u64*)ctx->hits.p; A.hit_off = (const i64*)ctx->hit_off.p; A.read_off = (const i64*)ctx->read_off.p;
    A.nreads = n;
    A.contig_off = (const i64*)B[4].p; A.contig_len = (const int*)((unsigned char*)B[4].p + (size_t)n_contigs * 8);
    A.contig_alt = (const unsigned char*)B[4].p + (size_t)n_contigs * 12; A.n_contigs = n_contigs;
    A.o = *opt;
    A.ch = (DChain*)B[0].p; A.sd = (DSeed*)B[1].p; A.hdr = (ReadHdr*)B[2].p; A.frac_rep = (float*)B[3].p;
    A.list = nullptr; A.nlist = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
    struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int i = 0; i < 4; ++i) if (e[i]) (void)hipEventDestroy(e[i]); } } ev_guard{ev};
    HIP_TRY(hipEventRecord(ev[0], ctx->stream));
    hipLaunchKernelGGL((k_chain<CHAIN_CAP, SEED_CAP>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, A);
    // second pass: the reads that did not fit, with the big scratch
    unsigned long long* d_redo_n = (unsigned long long*)((unsigned char*)B[5].p + cnt_bytes - 16);
    HIP_TRY(hipMemsetAsync(d_redo_n, 0, 8, ctx->stream));
    if ((rc = meme_buf_reserve(ctx, B[8], (size_t)n * 8))) return rc;
    hipLaunchKernelGGL(k_chain_redo, dim3(blocks_of(n, 256)), dim3(256), 0, ctx->stream, (const ReadHdr*)B[2].p, n, d_redo_n, (i64*)B[8].p);
    unsigned long long n_redo = 0;
    HIP_TRY(hipMemcpyAsync(&n_redo, d_redo_n, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (n_redo > 0) {
        size_t free_b = 0, total_b = 0;
        const size_t need = (size_t)n_redo * CHAIN_CAP2 * (sizeof(DChain) + SEED_CAP2 * sizeof(DSeed));
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need < free_b / 2) {          // else: those reads stay flagged for the host
            if ((rc = meme_buf_reserve(ctx, B[9], (size_t)n_redo * CHAIN_CAP2 * sizeof(DChain)))) return rc;
            if ((rc = meme_buf_reserve(ctx, B[10], (size_t)n_redo * CHAIN_CAP2 * SEED_CAP2 * sizeof(DSeed)))) return rc;
            ChainArgs A2 = A;
            A2.ch = (DChain*)B[9].p; A2.sd = (DSeed*)B[10].p; A2.list = (const i64*)B[8].p; A2.nlist = (i64)n_redo;
            HIP_TRY(hipEventRecord(ev[1], ctx->stream));
            // (4 lanes per wavefront: these reads are few and each is a long chain of dependent loads -- more, shorter-lived wavefronts hide
            //  more latency and wait less for their slowest read than fewer full ones: 18.3 -> 12.7 ms per 2 M reads)
            hipLaunchKernelGGL((k_chain<CHAIN_CAP2, SEED_CAP2>), dim3((unsigned)((n_redo + 3) / 4)), dim3(4), 0, ctx->stream, A2);
            HIP_TRY(hipEventRecord(ev[2], ctx->stream));
        }
    }
    i64* d_nch = (i64*)B[5].p;
    i64* d_nsd = d_nch + (n + 1);
    i64* d_choff = d_nsd + (n + 1);
    i64* d_sdoff = d_choff + (n + 1);
    int* d_tree = (int*)(d_sdoff + (n + 1));
    unsigned char* d_fb = (unsigned char*)(d_tree + n);
    hipLaunchKernelGGL(k_chain_counts, dim3(blocks_of(n + 1, 256)), dim3(256), 0, ctx->stream, (const ReadHdr*)B[2].p, n, d_nch, d_nsd, d_tree, d_fb);
    size_t tb = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_nch, d_choff, (i64)(n + 1), ctx->stream));
    if ((rc = meme_buf_reserve(ctx, ctx->scan_tmp, tb + 64))) return rc;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(ctx->scan_tmp.p, tb, d_nch, d_choff, (i64)(n + 1), ctx->stream));
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(ctx->scan_tmp.p, tb, d_nsd, d_sdoff, (i64)(n + 1), ctx->stream));
    i64 tot[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&tot[0], d_choff + n, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&tot[1], d_sdoff + n, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if ((rc = meme_buf_reserve(ctx, B[6], (size_t)(tot[0] + 1) * sizeof(meme_chain)))) return rc;
    if ((rc = meme_buf_reserve(ctx, B[7], (size_t)(tot[1] + 1) * sizeof(meme_chain_seed)))) return rc;
    hipLaunchKernelGGL(k_chain_pack, dim3(blocks_of(n, 256)), dim3(256), 0, ctx->stream, (const DChain*)B[0].p, (const DSeed*)B[1].p,
                       (const DChain*)B[9].p, (const DSeed*)B[10].p, (const ReadHdr*)B[2].p, (const i64*)d_choff, (const i64*)d_sdoff, n,
                       (meme_chain*)B[6].p, (meme_chain_seed*)B[7].p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev[3], ctx->stream));
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
    {
        float ms = 0.f, ms2 = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[3]));
        if (n_redo > 0 && hipEventQuery(ev[2]) == hipSuccess && hipEventElapsedTime(&ms2, ev[1], ev[2]) != hipSuccess) ms2 = 0.f;
        ctx->tm.chain_kernel_ms = ms;          // (includes the two small host round trips between the passes)
        ctx->tm.chain_pass2_ms = ms2;
    }
    out->nreads = n;
    out->chain_off = (const int64_t*)Hb[0].p; out->chains = (const meme_chain*)Hb[1].p;
    out->seed_off = (const int64_t*)Hb[2].p; out->seeds = (const meme_chain_seed*)Hb[3].p;
    out->tree_size = (const int32_t*)Hb[4].p; out->frac_rep = (const float*)Hb[5].p; out->fallback = (const uint8_t*)Hb[6].p;
    out->total_chains = tot[0]; out->total_seeds = tot[1];
    i64 nfb = 0;
    for (i64 i = 0; i < n; ++i) nfb += out->fallback[i] ? 1 : 0;
    out->n_fallback = nfb;
    return MEME_OK;
}
