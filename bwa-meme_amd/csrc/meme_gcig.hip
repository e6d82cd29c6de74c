// Banded global alignment with traceback -- the CIGAR kernel of the SAM phase: ksw_global2 (reference src/ksw.cpp:560-670)
// as bwa_gen_cigar2 (src/bwa.cpp:274-362) calls it from mem_reg2aln (src/bwamem.cpp:2314-2380) for every alignment that is written out.
//
// One wavefront per alignment; the lanes take the query columns of the band (64 at a time), the target rows run in sequence:
//   M(i,j)   = H(i-1,j-1) + S(i,j)                         lane-parallel (H of the row above, shifted by one column, in LDS)
//   E(i+1,j) = max{M(i,j) - gapo_del, E(i,j)} - gape_del    lane-parallel (E stays with its column)
//   F(i,j+1) = max{M(i,j) - gapo_ins, F(i,j)} - gape_ins    along the row: a max-plus prefix scan over the lanes, carried across chunks
//   H(i,j)   = max{M, E, F}; the direction byte of the cell (h from M/E/F, e opened or extended, f opened or extended) goes to the
//   backtrack matrix exactly as the reference stores it, and the backtrack walks it the reference's way -- so that ties (M over E over F)
//   and with them the placement of gaps in the CIGAR come out the same.
// Sequences are not shipped: a job names a read of the batch resident on the ctx, its query span, and a span of the fwd+rc text (2-bit
// pac64 in HBM); alignments on the reverse strand have both sequences reversed, as bwa_gen_cigar2 does to keep gaps left-aligned.
#include <string.h>

#include "meme_common.h"

namespace {

constexpr int MINUS_INF = -0x40000000;       // src/ksw.cpp:545

struct GcigArgs {
    const meme_gjob* jobs; i64 njobs;
    const uint8_t* reads; const i64* read_off; const u64* pac;
    meme_bsw_opt o;                          // o_del, e_del, o_ins, e_ins, a, b (zdrop / end_bonus unused)
    const i64* zoff; uint8_t* z;             // backtrack matrices: n_col * tlen bytes per job
    const i64* coff; uint32_t* cig;          // CIGAR scratch: qlen + tlen + 2 operations per job, filled from the back
    meme_gres* res;                          // score, n_cigar, (cigar_off = first operation inside the job's scratch, for the pack kernel)
    // bwa_gen_cigar2 whole (meme_gen_cigar_batch_host): NM and the MD string of every job; null for the plain ksw_global2 call
    const i64* mdoff; char* md;              // MD scratch: 2 * (qlen + tlen) + 16 bytes per job
    int32_t* nm; int32_t* mdlen;
    int zcap;                                // backtrack matrices of at most this many bytes stay in LDS (0: all in global memory)
    const i64* dp_list;                      // jobs that need the kernel below (the rest were answered by k_gcig_nogap), or null: all
    i64 list_first;                          // k_gcig / k_gcig_grp: this launch's jobs are dp_list[list_first ..) (the list is ordered by class: 16-lane, 32-lane, whole wavefront)
    int grp_qcap, grp_tcap, grp_z;           // k_gcig_grp: bytes of LDS per group for the query, the target and the backtrack matrix
    const u64* packed; int pW, pMW, pstride; // the batch's packed reads (2 bits per base + N masks, k_pack_reads): what k_gcig_nogap compares
};

__device__ __forceinline__ int text_base(const u64* pac, i64 p) { return (int)(pac[p >> 5] >> (62 - 2 * (int)(p & 31))) & 3; }
#define GCIG_DPP(old_, src_, ctrl_, rowmask_) __builtin_amdgcn_update_dpp((old_), (src_), (ctrl_), (rowmask_), 0xF, false)

// TWO (round 6): bands of 65-128 columns as ONE chunk per row, two adjacent columns per lane (the 250-bp / 5 % class: ~117 columns, which the 64-column chunks above ran as
// two passes per row, the second 53 lanes full); H and E in rings of 256 entries as in k_gcig_grp below, the boundary cells by arithmetic; the matrix, the walk back through
// the LDS window, NM and MD are this kernel's own code, unchanged.
template <bool TWO>
__global__ void __launch_bounds__(64) k_gcig_t(GcigArgs A) {
    extern __shared__ int lds[];             // hA[qlen + 2] | hB[qlen + 2] | e[qlen + 2] (TWO: 256 each) | query bytes (as ints, 4 per word)
    if ((i64)blockIdx.x >= A.njobs) return;
    const i64 jb = A.dp_list ? A.dp_list[A.list_first + blockIdx.x] : (i64)blockIdx.x;
    const int lane = threadIdx.x;
    const meme_gjob J = A.jobs[jb];
    const int qlen = J.qlen, tlen = J.tlen, w = J.w;
    constexpr int RM2 = 255;
    int* hA = lds;
    int* hB = hA + (TWO ? 256 : qlen + 2);
    int* eE = hB + (TWO ? 256 : qlen + 2);
    uint8_t* qs = reinterpret_cast<uint8_t*>(eE + (TWO ? 256 : qlen + 2));
    uint8_t* ts = qs + ((qlen + 3) & ~3);    // the target bases in row order: a row must not wait for a load from the text (1-2 us each, 250 in a row)
    uint8_t* zl = ts + ((tlen + 3) & ~3);    // the backtrack matrix, when it fits (A.zcap bytes): the walk back is one dependent load per step
    const uint8_t* rd = A.reads + A.read_off[J.read] + J.qb;
    for (int j = lane; j < qlen; j += 64) qs[j] = J.rev ? rd[qlen - 1 - j] : rd[j];
    if (w >= 0) for (int i = lane; i < tlen; i += 64) ts[i] = (uint8_t)text_base(A.pac, J.rev ? J.rb + tlen - 1 - i : J.rb + i);
    uint32_t* cg = A.cig + A.coff[jb];
    const int cap = qlen + tlen + 2;
    int n = 0;                               // operations so far; cg[cap - 1 - k] = k-th pushed
    int score;
    if (w < 0) {
        // bwa_gen_cigar2's gap-free shortcut (src/bwa.cpp:295-304: equal lengths and w_ == 0): one M operation, the score summed
        __syncthreads();
        int part = 0;
        for (int j = lane; j < qlen; j += 64) {
            const int tb = text_base(A.pac, J.rev ? J.rb + tlen - 1 - j : J.rb + j), qb = qs[j];
            part += qb > 3 ? -1 : (tb == qb ? A.o.a : -A.o.b);
        }
        for (int d = 32; d; d >>= 1) part += __shfl_xor(part, d);
        score = part;
        cg[cap - 1] = (unsigned)qlen << 4;
        n = 1;
    } else {
    const int oe_del = A.o.o_del + A.o.e_del, oe_ins = A.o.o_ins + A.o.e_ins, e_del = A.o.e_del, e_ins = A.o.e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    uint8_t* z = (i64)n_col * tlen <= A.zcap ? zl : A.z + A.zoff[jb];
    int* hp = hA;                            // H(i-1, j-1) at [j]
    int* hn = hB;
    if constexpr (TWO) {
        for (int j = lane + 1; j <= qlen && j <= w + 1; j += 64) hA[j & RM2] = j <= w ? -(A.o.o_ins + e_ins * j) : MINUS_INF;
        __syncthreads();
        for (int i = 0; i < tlen; ++i) {
            const int tb = ts[i];
            const int beg = i > w ? i - w : 0;
            const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
            const int end_prev = i == 0 ? 0 : (i + w < qlen ? i + w : qlen);
            const int j0 = beg + 2 * lane, j1 = j0 + 1;
            const bool in0 = j0 < end, in1 = j1 < end;
            const int qb0 = qs[j0], qb1 = qs[j1];
            const int h0 = hp[j0 & RM2], h1 = hp[j1 & RM2], ep0 = eE[j0 & RM2], ep1 = eE[j1 & RM2];
            const int sc0 = (tb > 3 || qb0 > 3) ? -1 : (tb == qb0 ? A.o.a : -A.o.b);
            const int sc1 = (tb > 3 || qb1 > 3) ? -1 : (tb == qb1 ? A.o.a : -A.o.b);
            const int m0 = (j0 == 0 ? (i == 0 ? 0 : -(A.o.o_del + e_del * i)) : h0) + sc0;
            const int m1 = h1 + sc1;
            const int e0 = j0 >= end_prev ? MINUS_INF : ep0, e1 = j1 >= end_prev ? MINUS_INF : ep1;
            // F along the row (see the one-column form below): G_c = M_c - oe_ins + c * e_ins over the row's columns c = j - beg; F_c = max_{k<c} G_k - (c - 1) * e_ins
            const int g0 = in0 ? m0 - oe_ins + (2 * lane) * e_ins : -2000000000;
            const int g1 = in1 ? m1 - oe_ins + (2 * lane + 1) * e_ins : -2000000000;
            int sg = g0 > g1 ? g0 : g1, y;
            y = GCIG_DPP(-2000000000, sg, 0x111, 0xF); sg = sg > y ? sg : y;
            y = GCIG_DPP(-2000000000, sg, 0x112, 0xF); sg = sg > y ? sg : y;
            y = GCIG_DPP(-2000000000, sg, 0x114, 0xF); sg = sg > y ? sg : y;
            y = GCIG_DPP(-2000000000, sg, 0x118, 0xF); sg = sg > y ? sg : y;
            y = GCIG_DPP(-2000000000, sg, 0x142, 0xA); sg = sg > y ? sg : y;
            y = GCIG_DPP(-2000000000, sg, 0x143, 0xC); sg = sg > y ? sg : y;
            const int ex = GCIG_DPP(-2000000000, sg, 0x138, 0xF);        // max over the lanes below (lane 0: not used)
            const int ex1 = lane == 0 ? g0 : (ex > g0 ? ex : g0);
            int fa = ex - (2 * lane - 1) * e_ins, fb = MINUS_INF - (2 * lane) * e_ins;
            const int f0 = lane == 0 ? MINUS_INF : (fa > fb ? fa : fb);
            fa = ex1 - (2 * lane) * e_ins; fb = MINUS_INF - (2 * lane + 1) * e_ins;
            const int f1 = fa > fb ? fa : fb;
            unsigned d0 = m0 >= e0 ? 0u : 1u, d1 = m1 >= e1 ? 0u : 1u;
            int hh0 = m0 >= e0 ? m0 : e0, hh1 = m1 >= e1 ? m1 : e1;
            d0 = hh0 >= f0 ? d0 : 2u; d1 = hh1 >= f1 ? d1 : 2u;
            hh0 = hh0 >= f0 ? hh0 : f0; hh1 = hh1 >= f1 ? hh1 : f1;
            int t0 = m0 - oe_del, t1 = m1 - oe_del;
            int ea = e0 - e_del, eb = e1 - e_del;
            d0 |= ea > t0 ? 1u << 2 : 0u; d1 |= eb > t1 ? 1u << 2 : 0u;
            ea = ea > t0 ? ea : t0; eb = eb > t1 ? eb : t1;
            t0 = m0 - oe_ins; t1 = m1 - oe_ins;
            d0 |= f0 - e_ins > t0 ? 2u << 4 : 0u; d1 |= f1 - e_ins > t1 ? 2u << 4 : 0u;
            uint8_t* zi = z + (i64)i * n_col;
            if (in0) { eE[j0 & RM2] = ea; hn[(j0 + 1) & RM2] = hh0; zi[2 * lane] = (uint8_t)d0; }
            if (in1) { eE[j1 & RM2] = eb; hn[(j1 + 1) & RM2] = hh1; zi[2 * lane + 1] = (uint8_t)d1; }
            __syncthreads();
            int* tsw = hp; hp = hn; hn = tsw;
        }
        score = hp[qlen & RM2];
    } else {
    // first row (src/ksw.cpp:591-595)
    for (int j = lane; j <= qlen; j += 64) {
        hA[j] = j == 0 ? 0 : (j <= w ? -(A.o.o_ins + e_ins * j) : MINUS_INF);
        eE[j] = MINUS_INF;
    }
    __syncthreads();
    for (int i = 0; i < tlen; ++i) {
        const int tb = ts[i];
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        int f_in = MINUS_INF;                                         // F(i, first column of the chunk)
        int h_in = beg == 0 ? -(A.o.o_del + e_del * (i + 1)) : MINUS_INF;   // H(i, beg - 1): what the reference's h1 starts from
        if (lane == 0 && beg <= qlen) hn[beg] = h_in;                 // (beg > qlen cannot happen for bands the host check admits: w >= |tlen - qlen|)
        uint8_t* zi = z + (i64)i * n_col;
        for (int cb = beg; cb < end; cb += 64) {
            const int j = cb + lane;
            const bool in = j < end;
            int m = MINUS_INF, e = MINUS_INF;
            if (in) {
                const int qb = qs[j];
                const int sc = (tb > 3 || qb > 3) ? -1 : (tb == qb ? A.o.a : -A.o.b);      // bwa_fill_scmat (src/bwa.cpp:262-270)
                m = hp[j] + sc;
                e = eE[j];
            }
            // F along the row: F(i,j) = max( f_in - (j-cb)*e_ins , max_{cb <= k < j} (M(i,k) - oe_ins - (j-1-k)*e_ins) )
            // as an inclusive max-scan of G_k = M(i,k) - oe_ins + k*e_ins over the lanes, shifted by one lane
            const int g = in ? m - oe_ins + (j - cb) * e_ins : -2000000000;
            // (the scan in the register file: row_shr 1/2/4/8, row_bcast15 into rows 1,3, row_bcast31 into rows 2,3, then wave_shr:1 -- as ds_bpermute
            // shuffles the eight steps were eight LDS round trips on the row's dependent chain, 1 000 of its 1 200 cycles)
            int sg = g, y;
            y = GCIG_DPP(-2000000000, sg, 0x111, 0xF); sg = sg > y ? sg : y;
            y = GCIG_DPP(-2000000000, sg, 0x112, 0xF); sg = sg > y ? sg : y;
            y = GCIG_DPP(-2000000000, sg, 0x114, 0xF); sg = sg > y ? sg : y;
            y = GCIG_DPP(-2000000000, sg, 0x118, 0xF); sg = sg > y ? sg : y;
            y = GCIG_DPP(-2000000000, sg, 0x142, 0xA); sg = sg > y ? sg : y;
            y = GCIG_DPP(-2000000000, sg, 0x143, 0xC); sg = sg > y ? sg : y;
            const int ex = GCIG_DPP(-2000000000, sg, 0x138, 0xF);        // max over lanes below (lane 0: not used)
            // f at this column: from the carry (decayed) or from a column of this chunk
            const int from_carry = f_in - lane * e_ins;
            int f = lane == 0 ? f_in : (ex - (lane - 1) * e_ins > from_carry ? ex - (lane - 1) * e_ins : from_carry);
            // the reference's cell, on this lane's values
            unsigned d = m >= e ? 0u : 1u;
            int h = m >= e ? m : e;
            d = h >= f ? d : 2u;
            h = h >= f ? h : f;
            int t = m - oe_del;
            int e2 = e - e_del;
            d |= e2 > t ? 1u << 2 : 0u;
            e2 = e2 > t ? e2 : t;
            t = m - oe_ins;
            int f2 = f - e_ins;
            d |= f2 > t ? 2u << 4 : 0u;
            f2 = f2 > t ? f2 : t;
            if (in) { eE[j] = e2; hn[j + 1] = h; zi[j - beg] = (uint8_t)d; }
            // carry to the next chunk: F(i, cb + 64) = f2 of lane 63
            f_in = __builtin_amdgcn_readlane(f2, 63);
        }
        if (lane == 0) eE[end] = MINUS_INF;                              // eh[end].e (:647); eh[end].h was stored by the row's last column
        __syncthreads();
        int* tsw = hp; hp = hn; hn = tsw;
    }
    score = hp[qlen];
    }
    // backtrack (src/ksw.cpp:650-664), every lane the same walk; the operations are written from the back of the job's scratch
    {
        __threadfence();
        __syncthreads();
        int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1, which = 0;
        const i64 zsize = (i64)n_col * tlen;
        int last_op = -1;
        unsigned cur = 0;
        auto push = [&](int op, int len) {
            if (last_op == op) cur += (unsigned)len << 4;
            else { if (last_op >= 0) { cg[cap - 1 - n] = cur; ++n; } cur = (unsigned)len << 4 | (unsigned)op; last_op = op; }
        };
        // The walk is the same on every lane: the byte it reads goes through readfirstlane so that i, k, `which` and the addresses live in scalar
        // registers.  Measured (profiles/r05_gcig.md): 21-29 % fewer VALU instructions per job and NO change in the kernel's time -- so the kernel is not
        // bound by VALU issue; what it IS bound by is open (four explanations refuted, same file).  Kept because it is exact and costs nothing.
        // A matrix in global memory is walked through a window in LDS: the rows [win_lo, win_hi) the walk is about to cross are fetched by the
        // whole wavefront at once (a walk straight on HBM is 500 dependent loads of a microsecond each -- ten times the DP above it).
        const bool windowed = z != zl && A.zcap >= 2 * n_col + 8;
        const int win_rows = windowed ? (A.zcap - 8) / n_col : 0;
        i64 win_base = 0, win_b0 = zsize;        // zl[x - win_base] = z[x] for the bytes [win_b0, ..) of the window
        while (i >= 0 && k >= 0) {
            i64 zi = (i64)i * n_col + (k - (i > w ? i - w : 0));
            if (zi < 0) zi = 0;
            if (zi >= zsize) zi = zsize - 1;
            if (windowed) {
                if (zi < win_b0) {
                    __syncthreads();
                    const int row = (int)(zi / n_col);      // (the clamps above can leave row i)
                    const int win_lo = row - win_rows + 1 > 0 ? row - win_rows + 1 : 0;
                    const i64 b0 = (i64)win_lo * n_col, b1 = (i64)(row + 1) * n_col;
                    win_b0 = b0;
                    const uint8_t* src = z + b0;
                    const int shift = (int)(reinterpret_cast<uintptr_t>(src) & 3);      // dword loads from the aligned address below
                    const unsigned* src4 = reinterpret_cast<const unsigned*>(src - shift);
                    unsigned* dst4 = reinterpret_cast<unsigned*>(zl);
                    const int nd = (int)((b1 - b0 + shift + 3) >> 2);
                    for (int d = lane; d < nd; d += 64) dst4[d] = src4[d];
                    win_base = b0 - shift;
                    __syncthreads();
                }
                which = __builtin_amdgcn_readfirstlane((int)zl[zi - win_base]) >> (which << 1) & 3;
            } else
            which = __builtin_amdgcn_readfirstlane((int)z[zi]) >> (which << 1) & 3;
            if (which == 0) { push(0, 1); --i; --k; }
            else if (which == 1) { push(2, 1); --i; }
            else { push(1, 1); --k; }
        }
        if (i >= 0) push(2, i + 1);
        if (k >= 0) push(1, k + 1);
        if (last_op >= 0) { cg[cap - 1 - n] = cur; ++n; }
    }
    }
    if (lane == 0) { meme_gres R; R.score = score; R.n_cigar = n; R.cigar_off = cap - n; A.res[jb] = R; }
    if (!A.md) return;
    // NM and MD (src/bwa.cpp:322-355): the operations in CIGAR order over the (possibly reversed) query and target.  Matches are compared 64
    // bases at a time -- one ballot, then only the mismatches are visited --, deleted bases are written by the lanes side by side; the text
    // of the string is identical on every lane, lane 0 stores it.
    __threadfence();
    __syncthreads();
    char* out = A.md + A.mdoff[jb];
    const char* const b2c = J.rev ? "TGCAN" : "ACGTN";
    int len = 0, x = 0, y = 0, u = 0, n_mm = 0, n_gap = 0;
    auto put_num = [&](int v) {              // kputw
        char buf[12];
        int l = 0;
        do { buf[l++] = (char)('0' + v % 10); v /= 10; } while (v);
        if (lane == 0) for (int i = 0; i < l; ++i) out[len + i] = buf[l - 1 - i];
        len += l;
    };
    for (int k = 0; k < n; ++k) {
        const unsigned c = cg[cap - n + k];
        const int op = (int)(c & 0xf), l = (int)(c >> 4);
        if (op == 0) {
            for (int i0 = 0; i0 < l; i0 += 64) {
                const int i = i0 + lane;
                const bool in = i < l;
                int tb = 0, qb = 0;
                if (in) { tb = w >= 0 ? (int)ts[y + i] : text_base(A.pac, J.rev ? J.rb + tlen - 1 - (y + i) : J.rb + y + i); qb = qs[x + i]; }
                unsigned long long mask = __ballot(in && tb != qb);
                int pos0 = 0;
                while (mask) {
                    const int b = __builtin_ctzll(mask);
                    u += b - pos0;
                    put_num(u);
                    const int t = __shfl(tb, b);
                    if (lane == 0) out[len] = b2c[t];
                    ++len; ++n_mm; u = 0; pos0 = b + 1;
                    mask &= mask - 1;
                }
                u += (l - i0 < 64 ? l - i0 : 64) - pos0;
            }
            x += l; y += l;
        } else if (op == 2) {
            if (k > 0 && k < n - 1) {        // (a deletion at either end of the CIGAR is not reported: mem_reg2aln squeezes it out)
                put_num(u);
                if (lane == 0) out[len] = '^';
                ++len;
                for (int i = lane; i < l; i += 64) out[len + i] = b2c[w >= 0 ? (int)ts[y + i] : text_base(A.pac, J.rev ? J.rb + tlen - 1 - (y + i) : J.rb + y + i)];
                len += l; u = 0; n_gap += l;
            }
            y += l;
        } else { x += l; n_gap += l; }
    }
    put_num(u);
    if (lane == 0) { out[len] = 0; A.nm[jb] = n_mm + n_gap; A.mdlen[jb] = len; }
}

// Several jobs per wavefront (round 6).  k_gcig above gives a job all 64 lanes and runs its rows one after the other; the bands bwa_gen_cigar2
// computes for short reads are 5-15 columns wide, so ~9 of the 64 lanes work and the kernel's time follows its instruction count
// (profiles/r05_gcig.md).  Here a job whose band has at most G columns (G = 16, 32 or 64: one, two or four DPP rows; G = 64 is one job per wavefront again, without the chunk loop and with the rings) takes a GROUP of G lanes and a wavefront
// runs 64 / G jobs side by side: the same instruction stream, 4 or 2 jobs per instruction.  A row is ONE chunk (end - beg <= 2w + 1 <= G), so F's
// max-plus scan needs the row_shr steps only (+ row_bcast15 for G = 32) and no carry.  H and E live in rings of 2G entries (a row reads what the row
// before wrote inside its band, never beyond: indices modulo the ring), the backtrack matrix (<= G x tlen bytes) always in LDS; the walk back runs
// per lane, identical within a group.  Same cell, same direction bytes, same walk as above -- so the same CIGAR, NM and MD.
template <int G>
__global__ void __launch_bounds__(64) k_gcig_grp(GcigArgs A) {
    extern __shared__ int lds[];
    constexpr int NG = 64 / G, R = 2 * G, RM = R - 1;
    constexpr unsigned long long GMASK = G == 64 ? ~0ull : G == 32 ? 0xffffffffull : 0xffffull;
    const int lane = threadIdx.x, sub = lane / G, gl = lane % G, gbase = sub * G;
    const i64 slot = (i64)blockIdx.x * NG + sub;
    const bool live = slot < A.njobs;
    const i64 jb = live ? A.dp_list[A.list_first + slot] : 0;
    meme_gjob J;
    if (live) J = A.jobs[jb];
    else { J.rb = 0; J.read = 0; J.qb = 0; J.qlen = 0; J.tlen = 0; J.w = 0; J.rev = 0; }
    const int qlen = J.qlen, tlen = J.tlen, w = J.w;
    const int stride = 3 * R + ((A.grp_qcap + A.grp_tcap + A.grp_z) >> 2);          // ints per group
    int* hA = lds + sub * stride;
    int* hB = hA + R;
    int* eE = hB + R;
    uint8_t* qs = reinterpret_cast<uint8_t*>(eE + R);
    uint8_t* ts = qs + A.grp_qcap;
    uint8_t* zl = ts + A.grp_tcap;
    if (live) {
        const uint8_t* rd = A.reads + A.read_off[J.read] + J.qb;
        for (int j = gl; j < qlen; j += G) qs[j] = J.rev ? rd[qlen - 1 - j] : rd[j];
        for (int i = gl; i < tlen; i += G) ts[i] = (uint8_t)text_base(A.pac, J.rev ? J.rb + tlen - 1 - i : J.rb + i);
    }
    uint32_t* cg = A.cig + (live ? A.coff[jb] : 0);
    const int cap = qlen + tlen + 2;
    int n = 0;
    const int oe_del = A.o.o_del + A.o.e_del, oe_ins = A.o.o_ins + A.o.e_ins, e_del = A.o.e_del, e_ins = A.o.e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    // first row (src/ksw.cpp:591-595): the columns row 0 can read.  The cells left of the band and E above a column the band has just reached are not kept in the
    // rings: H(i-1, -1) is arithmetic (-(o_del + e_del * i), 0 for the first row) and E of a column at or beyond the previous row's end is -inf (:647) -- two
    // single-lane stores per row less, and every load of a row is unconditional (lanes beyond the band compute on whatever the ring holds and store nothing).
    for (int j = gl + 1; j <= qlen && j <= w + 1; j += G) hA[j & RM] = j <= w ? -(A.o.o_ins + e_ins * j) : MINUS_INF;
    int tl_max = tlen;
    for (int d = 32; d >= G; d >>= 1) { const int o = __shfl_xor(tl_max, d); tl_max = tl_max > o ? tl_max : o; }
    __syncthreads();
    int* hp = hA;
    int* hn = hB;
    for (int i = 0; i < tl_max; ++i) {
        const bool row = i < tlen;
        const int tb = ts[i < tlen ? i : 0];
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int end_prev = i == 0 ? 0 : (i + w < qlen ? i + w : qlen);
        const int j = beg + gl;
        const bool in = row && j < end;
        const int qb = qs[j];
        const int hprev = hp[j & RM], eprev = eE[j & RM];
        const int sc = (tb > 3 || qb > 3) ? -1 : (tb == qb ? A.o.a : -A.o.b);
        const int m = (j == 0 ? (i == 0 ? 0 : -(A.o.o_del + e_del * i)) : hprev) + sc;
        const int e = j >= end_prev ? MINUS_INF : eprev;
        const int g = in ? m - oe_ins + gl * e_ins : -2000000000;
        int sg = g, y;
        y = GCIG_DPP(-2000000000, sg, 0x111, 0xF); sg = sg > y ? sg : y;
        y = GCIG_DPP(-2000000000, sg, 0x112, 0xF); sg = sg > y ? sg : y;
        y = GCIG_DPP(-2000000000, sg, 0x114, 0xF); sg = sg > y ? sg : y;
        y = GCIG_DPP(-2000000000, sg, 0x118, 0xF); sg = sg > y ? sg : y;
        if (G >= 32) { y = GCIG_DPP(-2000000000, sg, 0x142, 0xA); sg = sg > y ? sg : y; }
        if (G == 64) { y = GCIG_DPP(-2000000000, sg, 0x143, 0xC); sg = sg > y ? sg : y; }
        const int ex = GCIG_DPP(-2000000000, sg, 0x138, 0xF);            // the lane below (a group's lane 0 does not use it)
        const int f = gl == 0 ? MINUS_INF : (ex - (gl - 1) * e_ins > MINUS_INF - gl * e_ins ? ex - (gl - 1) * e_ins : MINUS_INF - gl * e_ins);
        unsigned d = m >= e ? 0u : 1u;
        int h = m >= e ? m : e;
        d = h >= f ? d : 2u;
        h = h >= f ? h : f;
        int t = m - oe_del;
        int e2 = e - e_del;
        d |= e2 > t ? 1u << 2 : 0u;
        e2 = e2 > t ? e2 : t;
        t = m - oe_ins;
        const int f2 = f - e_ins;
        d |= f2 > t ? 2u << 4 : 0u;
        if (in) { eE[j & RM] = e2; hn[(j + 1) & RM] = h; zl[i * n_col + gl] = (uint8_t)d; }
        __syncthreads();
        if (row) { int* tsw = hp; hp = hn; hn = tsw; }
    }
    const int score = live ? hp[qlen & RM] : 0;
    // backtrack (src/ksw.cpp:650-664): every lane of the group the same walk
    {
        int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1, which = 0;
        const int zsize = n_col * tlen;
        int last_op = -1;
        unsigned cur = 0;
        auto push = [&](int op, int len) {
            if (last_op == op) cur += (unsigned)len << 4;
            else { if (last_op >= 0) { if (gl == 0) cg[cap - 1 - n] = cur; ++n; } cur = (unsigned)len << 4 | (unsigned)op; last_op = op; }
        };
        while (i >= 0 && k >= 0) {
            int zi = i * n_col + (k - (i > w ? i - w : 0));
            if (zi < 0) zi = 0;
            if (zi >= zsize) zi = zsize - 1;
            which = (int)zl[zi] >> (which << 1) & 3;
            if (which == 0) { push(0, 1); --i; --k; }
            else if (which == 1) { push(2, 1); --i; }
            else { push(1, 1); --k; }
        }
        if (i >= 0) push(2, i + 1);
        if (k >= 0) push(1, k + 1);
        if (last_op >= 0) { if (gl == 0) cg[cap - 1 - n] = cur; ++n; }
    }
    if (live && gl == 0) { meme_gres Rr; Rr.score = score; Rr.n_cigar = n; Rr.cigar_off = cap - n; A.res[jb] = Rr; }
    if (!A.md) return;
    // NM and MD (src/bwa.cpp:322-355), as in k_gcig with the group's G lanes
    __threadfence();
    __syncthreads();
    char* out = A.md + (live ? A.mdoff[jb] : 0);
    const char* const b2c = J.rev ? "TGCAN" : "ACGTN";
    int len = 0, x = 0, y = 0, u = 0, n_mm = 0, n_gap = 0;
    auto put_num = [&](int v) {
        char buf[12];
        int l = 0;
        do { buf[l++] = (char)('0' + v % 10); v /= 10; } while (v);
        if (gl == 0) for (int i = 0; i < l; ++i) out[len + i] = buf[l - 1 - i];
        len += l;
    };
    for (int k = 0; k < n; ++k) {
        const unsigned c = cg[cap - n + k];
        const int op = (int)(c & 0xf), l = (int)(c >> 4);
        if (op == 0) {
            for (int i0 = 0; i0 < l; i0 += G) {
                const int i = i0 + gl;
                const bool in = i < l;
                int tb = 0, qb = 0;
                if (in) { tb = (int)ts[y + i]; qb = qs[x + i]; }
                unsigned long long mask = (__ballot(in && tb != qb) >> gbase) & GMASK;
                int pos0 = 0;
                while (mask) {
                    const int b = __builtin_ctzll(mask);
                    u += b - pos0;
                    put_num(u);
                    const int t = __shfl(tb, gbase + b);
                    if (gl == 0) out[len] = b2c[t];
                    ++len; ++n_mm; u = 0; pos0 = b + 1;
                    mask &= mask - 1;
                }
                u += (l - i0 < G ? l - i0 : G) - pos0;
            }
            x += l; y += l;
        } else if (op == 2) {
            if (k > 0 && k < n - 1) {
                put_num(u);
                if (gl == 0) out[len] = '^';
                ++len;
                for (int i = gl; i < l; i += G) out[len + i] = b2c[(int)ts[y + i]];
                len += l; u = 0; n_gap += l;
            }
            y += l;
        } else { x += l; n_gap += l; }
    }
    if (live) {
        put_num(u);
        if (gl == 0) { out[len] = 0; A.nm[jb] = n_mm + n_gap; A.mdlen[jb] = len; }
    }
}

// bwa_gen_cigar2's gap-free shortcut (src/bwa.cpp:295-304) for the jobs of a batch that take it, ONE LANE per job: the query span against
// the text 32 bases per XOR on the packed reads k_pack_reads left on the ctx (2 bits per base, N as A with a mask beside it) and the 2-bit
// text -- score, the one M operation, NM and the MD string (visited mismatch by mismatch, in the reversed order on the reverse strand).
// A batch of 150-bp reads with 1 % errors poses 70 % of its calls this way; a wavefront-wide block each (k_gcig) cost 70 ns per job.
constexpr int NOGAP_MAX_LEN = 500;           // reads beyond LEARNED_MAX_READ_LEN are not packed: k_gcig takes their jobs
__device__ __forceinline__ bool nogap_fast(const meme_gjob& J, const i64* read_off) {
    return J.w < 0 && read_off[J.read + 1] - read_off[J.read] <= NOGAP_MAX_LEN;
}
__global__ void __launch_bounds__(256) k_gcig_nogap(GcigArgs A) {
    for (i64 jb = (i64)blockIdx.x * blockDim.x + threadIdx.x; jb < A.njobs; jb += (i64)gridDim.x * blockDim.x) {
        const meme_gjob J = A.jobs[jb];
        if (!nogap_fast(J, A.read_off)) continue;
        const int qlen = J.qlen;
        const u64* fw = A.packed + (i64)J.read * A.pstride;
        const u64* nmask = fw + 2 * A.pW;
        const bool has_n = (fw[A.pstride - 1] >> 31) & 1ull;
        char* out = A.md ? A.md + A.mdoff[jb] : nullptr;
        const char* const b2c = J.rev ? "TGCA" : "ACGT";
        const int nblk = (qlen + 31) >> 5;
        int len = 0, n_mm = 0, n_n = 0, last = J.rev ? qlen : -1;
        auto put_run = [&](int run) {
            char buf[12];
            int l = 0;
            do { buf[l++] = (char)('0' + run % 10); run /= 10; } while (run);
            for (int i = 0; i < l; ++i) out[len + i] = buf[l - 1 - i];
            len += l;
        };
        for (int kk = 0; kk < nblk; ++kk) {
            const int k = J.rev ? nblk - 1 - kk : kk;
            const int p0 = J.qb + 32 * k, nvalid = qlen - 32 * k < 32 ? qlen - 32 * k : 32;
            const int wi = p0 >> 5, sh = (p0 & 31) * 2;
            const u64 q = sh ? (fw[wi] << sh) | (fw[wi + 1] >> (64 - sh)) : fw[wi];
            const u64 t = extract32(A.pac, J.rb + 32 * k);
            const u64 x = q ^ t;
            u64 y = (x | (x >> 1)) & 0x5555555555555555ull;                     // bit 62 - 2i: base i of the block differs
            if (has_n) {
                const int mw = p0 >> 6, ms = p0 & 63;
                u64 nb = nmask[mw] >> ms;
                if (ms > 32 && mw + 1 < A.pMW) nb |= nmask[mw + 1] << (64 - ms);
                unsigned nbits = (unsigned)(nb & 0xffffffffull);
                if (nvalid < 32) nbits &= (1u << nvalid) - 1u;
                n_n += __popc(nbits);
                while (nbits) { const int i = __ffs((int)nbits) - 1; nbits &= nbits - 1; y |= 1ull << (62 - 2 * i); }
            }
            if (nvalid < 32) y &= ~0ull << (2 * (32 - nvalid));
            n_mm += __popcll(y);
            if (out)
                while (y) {
                    const int i = J.rev ? 31 - (__builtin_ctzll(y) >> 1) : (__clzll((long long)y) >> 1);
                    y &= ~(1ull << (62 - 2 * i));
                    const int j = 32 * k + i;
                    put_run(J.rev ? last - j - 1 : j - last - 1);
                    out[len++] = b2c[(int)(t >> (62 - 2 * i)) & 3];
                    last = j;
                }
        }
        if (out) { put_run(J.rev ? last : qlen - last - 1); out[len] = 0; A.nm[jb] = n_mm; A.mdlen[jb] = len; }
        const int cap = qlen + J.tlen + 2;
        A.cig[A.coff[jb] + cap - 1] = (unsigned)qlen << 4;
        meme_gres R;
        R.score = A.o.a * (qlen - n_mm) - A.o.b * (n_mm - n_n) - n_n;            // bwa_fill_scmat: match a, mismatch -b, an ambiguous base -1
        R.n_cigar = 1; R.cigar_off = cap - 1;
        A.res[jb] = R;
    }
}

// the operations of every job, densely packed in job order (they were pushed back to front: already in CIGAR order)
__global__ void __launch_bounds__(256) k_gcig_pack(const meme_gres* __restrict__ res, const i64* __restrict__ coff, const uint32_t* __restrict__ cig,
                                                    const i64* __restrict__ ooff, i64 njobs, uint32_t* __restrict__ out) {
    for (i64 jb = (i64)blockIdx.x * blockDim.x + threadIdx.x; jb < njobs; jb += (i64)gridDim.x * blockDim.x) {
        const meme_gres R = res[jb];
        const uint32_t* src = cig + coff[jb] + R.cigar_off;
        uint32_t* dst = out + ooff[jb];
        for (int k = 0; k < R.n_cigar; ++k) dst[k] = src[k];
    }
}
// per job: the kernel that takes it -- isdp: k_gcig (a wavefront), is16 / is32: k_gcig_grp<16 / 32> (a group of lanes; z16 / z32 = bytes of LDS a group has for
// the backtrack matrix, 0: the class is not used), none: k_gcig_nogap -- and the sizes of its scratch
__global__ void __launch_bounds__(256) k_gcig_sizes(const meme_gjob* __restrict__ jobs, i64 njobs, const i64* __restrict__ read_off, bool fast, int zcap, int z16, int z32, int z64, bool two, i64* __restrict__ zsz,
                                                     i64* __restrict__ csz, i64* __restrict__ msz, i64* __restrict__ isdp, i64* __restrict__ is16, i64* __restrict__ is32, i64* __restrict__ is64,
                                                     i64* __restrict__ is128) {
    for (i64 jb = (i64)blockIdx.x * blockDim.x + threadIdx.x; jb < njobs; jb += (i64)gridDim.x * blockDim.x) {
        const meme_gjob J = jobs[jb];
        const bool dp = !(fast && nogap_fast(J, read_off));
        const i64 n_col = J.qlen < 2 * J.w + 1 ? J.qlen : 2 * J.w + 1;
        const bool g16 = dp && J.w >= 0 && n_col <= 16 && n_col * J.tlen <= z16;
        const bool g32 = dp && !g16 && J.w >= 0 && n_col <= 32 && n_col * J.tlen <= z32;
        const bool g64 = dp && !g16 && !g32 && J.w >= 0 && n_col <= 64 && n_col * J.tlen <= z64;
        const bool g128 = two && dp && !g16 && !g32 && !g64 && J.w >= 0 && n_col <= 128;      // k_gcig_t<true>: two columns per lane (matrix in LDS or HBM as for k_gcig)
        is16[jb] = g16; is32[jb] = g32; is64[jb] = g64; is128[jb] = g128;
        isdp[jb] = dp && !g16 && !g32 && !g64 && !g128;                      // 1: the job goes to k_gcig
        zsz[jb] = J.w < 0 || g16 || g32 || g64 || n_col * J.tlen <= zcap ? 0 : (n_col * J.tlen + 15) & ~(i64)15;   // (w < 0: the gap-free shortcut, no matrix; small matrices stay in LDS)
        csz[jb] = J.qlen + J.tlen + 2;
        if (msz) msz[jb] = 2 * ((i64)J.qlen + J.tlen) + 16;                // an MD string never has more than two characters per base
    }
}
// the list the three kernels draw from: the 16-lane jobs, then the 32-lane jobs, then the whole-wavefront jobs, each in job order
__global__ void __launch_bounds__(256) k_gcig_dplist(const i64* __restrict__ isdp, const i64* __restrict__ dpoff, const i64* __restrict__ is16, const i64* __restrict__ o16,
                                                      const i64* __restrict__ is32, const i64* __restrict__ o32, const i64* __restrict__ is64, const i64* __restrict__ o64, const i64* __restrict__ is128,
                                                      const i64* __restrict__ o128, i64 njobs, i64* __restrict__ list) {
    const i64 n16 = o16[njobs], n32 = o32[njobs], n64 = o64[njobs], n128 = o128[njobs];
    for (i64 jb = (i64)blockIdx.x * blockDim.x + threadIdx.x; jb < njobs; jb += (i64)gridDim.x * blockDim.x) {
        if (is16[jb]) list[o16[jb]] = jb;
        else if (is32[jb]) list[n16 + o32[jb]] = jb;
        else if (is64[jb]) list[n16 + n32 + o64[jb]] = jb;
        else if (is128[jb]) list[n16 + n32 + n64 + o128[jb]] = jb;
        else if (isdp[jb]) list[n16 + n32 + n64 + n128 + dpoff[jb]] = jb;
    }
}
__global__ void __launch_bounds__(256) k_gcig_ncig(const meme_gres* __restrict__ res, i64 njobs, i64* __restrict__ ncig) {
    for (i64 jb = (i64)blockIdx.x * blockDim.x + threadIdx.x; jb < njobs; jb += (i64)gridDim.x * blockDim.x) ncig[jb] = res[jb].n_cigar;
}
__global__ void __launch_bounds__(256) k_gcig_fix(meme_gres* __restrict__ res, const i64* __restrict__ ooff, i64 njobs) {
    for (i64 jb = (i64)blockIdx.x * blockDim.x + threadIdx.x; jb < njobs; jb += (i64)gridDim.x * blockDim.x) res[jb].cigar_off = ooff[jb];
}

// what the host loop cannot see: a job's query span against the length of the read it names
__global__ void __launch_bounds__(256) k_gcig_check(const meme_gjob* __restrict__ jobs, i64 njobs, const i64* __restrict__ read_off, i64* __restrict__ bad) {
    for (i64 jb = (i64)blockIdx.x * blockDim.x + threadIdx.x; jb < njobs; jb += (i64)gridDim.x * blockDim.x) {
        const meme_gjob J = jobs[jb];
        if ((i64)J.qb + J.qlen > read_off[J.read + 1] - read_off[J.read]) atomicMin((unsigned long long*)bad, (unsigned long long)jb);
    }
}

// bwa_gen_cigar2's own preamble (src/bwa.cpp:288-316) for a batch of its calls: the strand from rb, the gap-free shortcut (equal lengths,
// w_ == 0: band -1 here) or the band it hands to ksw_global2
__global__ void __launch_bounds__(256) k_cjob_prep(const meme_cjob* __restrict__ cj, i64 njobs, i64 l_pac, meme_bsw_opt o, meme_gjob* __restrict__ jobs) {
    for (i64 jb = (i64)blockIdx.x * blockDim.x + threadIdx.x; jb < njobs; jb += (i64)gridDim.x * blockDim.x) {
        const meme_cjob C = cj[jb];
        meme_gjob J;
        J.rb = C.rb; J.read = C.read; J.qb = C.qb; J.qlen = C.qlen; J.tlen = C.tlen; J.rev = C.rb >= l_pac ? 1 : 0;
        if (C.qlen == C.tlen && C.w_ == 0) J.w = -1;
        else {
            const int max_ins = (int)((double)(((C.qlen + 1) >> 1) * o.a - o.o_ins) / o.e_ins + 1.);
            const int max_del = (int)((double)(((C.qlen + 1) >> 1) * o.a - o.o_del) / o.e_del + 1.);
            int max_gap = max_ins > max_del ? max_ins : max_del;
            max_gap = max_gap > 1 ? max_gap : 1;
            const int dl = C.tlen > C.qlen ? C.tlen - C.qlen : C.qlen - C.tlen;
            int w = (max_gap + dl + 1) >> 1;
            w = w < C.w_ ? w : C.w_;
            w = w > dl + 3 ? w : dl + 3;
            J.w = w;
        }
        jobs[jb] = J;
    }
}
// packed MD strings (NUL-terminated, job order) + the per-job result records of meme_gen_cigar_batch_host
__global__ void __launch_bounds__(256) k_md_sizes(const int32_t* __restrict__ mdlen, i64 njobs, i64* __restrict__ sz) {
    for (i64 jb = (i64)blockIdx.x * blockDim.x + threadIdx.x; jb < njobs; jb += (i64)gridDim.x * blockDim.x) sz[jb] = (i64)mdlen[jb] + 1;
}
__global__ void __launch_bounds__(256) k_md_pack(const meme_gres* __restrict__ res, const int32_t* __restrict__ nm, const int32_t* __restrict__ mdlen, const i64* __restrict__ mdoff,
                                                  const char* __restrict__ md, const i64* __restrict__ poff, i64 njobs, char* __restrict__ out, meme_cres* __restrict__ cres) {
    for (i64 jb = (i64)blockIdx.x * blockDim.x + threadIdx.x; jb < njobs; jb += (i64)gridDim.x * blockDim.x) {
        const int l = mdlen[jb];
        const char* src = md + mdoff[jb];
        char* dst = out + poff[jb];
        for (int k = 0; k <= l; ++k) dst[k] = src[k];
        meme_cres R;
        R.score = res[jb].score; R.n_cigar = res[jb].n_cigar; R.nm = nm[jb]; R.md_len = l; R.cigar_off = res[jb].cigar_off; R.md_off = poff[jb];
        cres[jb] = R;
    }
}

unsigned grid_of(i64 items, int per) { i64 b = (items + per - 1) / per; const i64 cap = 256 * 64; return (unsigned)(b < cap ? (b < 1 ? 1 : b) : cap); }

// The batch from the jobs in G[0] (device) to packed results: scratch sizes, the alignment kernel, the packed operations (and MD strings).
// with_md: NM + MD of every job (meme_gen_cigar_batch_host); host_jobs (may be null) only serves the error message of a bad query span.
struct GcigRun { i64 tops = 0, tmd = 0; };
constexpr int Z_LDS_WINDOW = 2048;          // ... and when matrices do not fit anyway: the window the walk back reads them through
constexpr int Z_LDS_CAP = 8192;             // a 250-row matrix of up to 32 band columns: the bands bwa_gen_cigar2 computes for reads with a few small indels
int gcig_run(meme_ctx* ctx, i64 njobs, int qmax, int tmax, const meme_bsw_opt* opt, bool with_md, const char* who, GcigRun* out) {
    int rc;
    const size_t lds_base = (size_t)(3 * (qmax + 2)) * 4 + (size_t)((qmax + 3) & ~3) + (size_t)((tmax + 3) & ~3);
    // The LDS kept per job for its backtrack matrix or window (tuning "gcig_zcap" overrides; 0: none).  Measured on two read classes only
    // (profiles/r05_gcig.md, 400 k calls each): where a typical matrix -- the rows of the longest target x the ~33 band columns bwa_gen_cigar2 computes
    // for a read with a few small indels -- fits, keeping it whole pays (150 bp: 7.1 ms against 9.8 without); where it does not, only the window is used
    // and a small one leaves room for more wavefronts (250 bp: 47.5 ms with 2 KB against 52.9 with 8 KB).
    const int zauto = (size_t)tmax * 33 <= (size_t)Z_LDS_CAP ? Z_LDS_CAP : Z_LDS_WINDOW;
    const int zwant = ctx->gcig_zcap >= 0 ? (int)ctx->gcig_zcap : zauto;
    const int zcap = lds_base + (size_t)zwant <= 32 * 1024 ? zwant : 0;          // (long reads: their rows fill the LDS, the matrix stays in global memory)
    DevBuf* G = ctx->gcig;      // 0 jobs, 1 sizes + offsets (8 x (n+1)), 2 z, 3 cigar scratch, 4 results, 5 packed cigars, 6 MD scratch, 7 nm + mdlen, 8 packed MD, 9 cjobs, 10 cres
    if ((rc = meme_buf_reserve(ctx, G[1], (size_t)(njobs + 1) * 8 * 21 + 64)) || (rc = meme_buf_reserve(ctx, G[4], (size_t)njobs * sizeof(meme_gres)))) return rc;
    i64* d_zsz = (i64*)G[1].p;
    i64* d_csz = d_zsz + (njobs + 1);
    i64* d_zoff = d_csz + (njobs + 1);
    i64* d_coff = d_zoff + (njobs + 1);
    i64* d_ncig = d_coff + (njobs + 1);
    i64* d_ooff = d_ncig + (njobs + 1);
    i64* d_msz = d_ooff + (njobs + 1);
    i64* d_moff = d_msz + (njobs + 1);
    i64* d_psz = d_moff + (njobs + 1);
    i64* d_poff = d_psz + (njobs + 1);
    i64* d_isdp = d_poff + (njobs + 1);
    i64* d_dpoff = d_isdp + (njobs + 1);
    i64* d_dplist = d_dpoff + (njobs + 1);
    i64* d_is16 = d_dplist + (njobs + 1);
    i64* d_o16 = d_is16 + (njobs + 1);
    i64* d_is32 = d_o16 + (njobs + 1);
    i64* d_o32 = d_is32 + (njobs + 1);
    i64* d_is64 = d_o32 + (njobs + 1);
    i64* d_o64 = d_is64 + (njobs + 1);
    i64* d_is128 = d_o64 + (njobs + 1);
    i64* d_o128 = d_is128 + (njobs + 1);
    i64* d_bad = d_o128 + (njobs + 1);
    // Several jobs per wavefront for narrow bands (k_gcig_grp; tuning "gcig_groups" = 0: every job a wavefront): per group rings for H and E, the query, the
    // target and a matrix of 16 (32) columns x the longest target, as long as a wavefront's groups stay within 24 KB (six wavefronts per CU and more)
    const int grp_qcap = (qmax + 3) & ~3, grp_tcap = (tmax + 3) & ~3;
    auto grp_lds = [&](int g, int z) { return (size_t)(64 / g) * ((size_t)3 * 2 * g * 4 + grp_qcap + grp_tcap + z); };
    int z16 = ctx->gcig_groups ? ((16 * tmax + 3) & ~3) : 0, z32 = ctx->gcig_groups ? ((32 * tmax + 3) & ~3) : 0;
    int z64 = ctx->gcig_groups ? ((64 * tmax + 3) & ~3) : 0;
    if (z16 > 4096) z16 = 4096;
    if (z32 > 8192) z32 = 8192;
    if (z64 > 12288) z64 = 12288;
    if (grp_lds(16, z16) > 24 * 1024) z16 = 0;
    if (grp_lds(32, z32) > 24 * 1024) z32 = 0;
    if (grp_lds(64, z64) > 16 * 1024) z64 = 0;
    const size_t lds_two = (size_t)3 * 256 * 4 + (size_t)((qmax + 3) & ~3) + (size_t)((tmax + 3) & ~3) + (size_t)(zcap > 256 ? zcap : 256);
    // (measured, profiles/r06_gcig.md: 23 % fewer instructions per job than two 64-column chunks and 6 % MORE time on the 250-bp class -- the class stays opt-in, gcig_groups = 2)
    const bool two = ctx->gcig_groups >= 2 && lds_two <= 32 * 1024;
    // the gap-free shortcut on the packed reads the seeding call left on the ctx (reads of at most 500 bases)
    const bool fast = ctx->packed.p != nullptr && ctx->last_seed_max_len > 0;
    const int pW = (int)((ctx->last_seed_max_len + 31) / 32) + 2, pMW = (int)((ctx->last_seed_max_len + 63) / 64);      // PackGeom of that batch (meme_seed.hip)
    HIP_TRY(hipMemsetAsync(d_bad, 0xff, 8, ctx->stream));
    hipLaunchKernelGGL(k_gcig_check, dim3(grid_of(njobs, 256)), dim3(256), 0, ctx->stream, (const meme_gjob*)G[0].p, (i64)njobs, (const i64*)ctx->read_off.p, d_bad);
    hipLaunchKernelGGL(k_gcig_sizes, dim3(grid_of(njobs, 256)), dim3(256), 0, ctx->stream, (const meme_gjob*)G[0].p, (i64)njobs, (const i64*)ctx->read_off.p, fast, zcap, z16, z32, z64, two, d_zsz, d_csz,
                       with_md ? d_msz : (i64*)nullptr, d_isdp, d_is16, d_is32, d_is64, d_is128);
    if ((rc = meme_scan_exclusive(ctx, d_zsz, d_zoff, njobs)) || (rc = meme_scan_exclusive(ctx, d_csz, d_coff, njobs)) || (rc = meme_scan_exclusive(ctx, d_isdp, d_dpoff, njobs)) ||
        (rc = meme_scan_exclusive(ctx, d_is16, d_o16, njobs)) || (rc = meme_scan_exclusive(ctx, d_is32, d_o32, njobs)) || (rc = meme_scan_exclusive(ctx, d_is64, d_o64, njobs)) ||
        (rc = meme_scan_exclusive(ctx, d_is128, d_o128, njobs))) return rc;
    hipLaunchKernelGGL(k_gcig_dplist, dim3(grid_of(njobs, 256)), dim3(256), 0, ctx->stream, (const i64*)d_isdp, (const i64*)d_dpoff, (const i64*)d_is16, (const i64*)d_o16,
                       (const i64*)d_is32, (const i64*)d_o32, (const i64*)d_is64, (const i64*)d_o64, (const i64*)d_is128, (const i64*)d_o128, (i64)njobs, d_dplist);
    if (with_md && (rc = meme_scan_exclusive(ctx, d_msz, d_moff, njobs))) return rc;
    i64 tz = 0, tc = 0, tm = 0, ndp = 0, n16 = 0, n32 = 0, n64 = 0, n128 = 0;
    HIP_TRY(hipMemcpyAsync(&n128, d_o128 + njobs, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&n64, d_o64 + njobs, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&ndp, d_dpoff + njobs, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&n16, d_o16 + njobs, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&n32, d_o32 + njobs, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&tz, d_zoff + njobs, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&tc, d_coff + njobs, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (with_md) HIP_TRY(hipMemcpyAsync(&tm, d_moff + njobs, 8, hipMemcpyDeviceToHost, ctx->stream));
    i64 bad = -1;
    HIP_TRY(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (bad >= 0) {
        meme_set_error("%s: job %lld names a query span beyond the end of read it refers to", who, (long long)bad);
        return MEME_E_ARG;
    }
    {
        size_t free_b = 0, total_b = 0;
        const size_t need = (size_t)tz + (size_t)tc * 4 + (size_t)tm;
        const size_t have = G[2].cap + G[3].cap + G[6].cap;
        if (need > have && hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b / 2 + have) {
            meme_set_error("%s: %lld alignments need %.1f GB of backtrack matrices, more than half of the free HBM: submit fewer at a time", who, (long long)njobs, need / 1e9);
            return MEME_E_CAPACITY;
        }
    }
    if ((rc = meme_buf_reserve(ctx, G[2], (size_t)tz + 64)) || (rc = meme_buf_reserve(ctx, G[3], (size_t)(tc + 1) * 4))) return rc;
    if (with_md && ((rc = meme_buf_reserve(ctx, G[6], (size_t)tm + 64)) || (rc = meme_buf_reserve(ctx, G[7], (size_t)njobs * 8 + 64)))) return rc;
    GcigArgs A;
    A.jobs = (const meme_gjob*)G[0].p; A.njobs = njobs; A.reads = (const uint8_t*)ctx->reads.p; A.read_off = (const i64*)ctx->read_off.p; A.pac = ctx->idx.pac;
    A.o = *opt; A.zoff = d_zoff; A.z = (uint8_t*)G[2].p; A.coff = d_coff; A.cig = (uint32_t*)G[3].p; A.res = (meme_gres*)G[4].p;
    A.mdoff = d_moff; A.md = with_md ? (char*)G[6].p : nullptr;
    A.nm = with_md ? (int32_t*)G[7].p : nullptr; A.mdlen = with_md ? (int32_t*)G[7].p + njobs : nullptr;
    A.zcap = zcap; A.dp_list = nullptr; A.packed = (const u64*)ctx->packed.p; A.pW = pW; A.pMW = pMW; A.pstride = 2 * pW + 2 * pMW + 1;
    A.list_first = 0; A.grp_qcap = grp_qcap; A.grp_tcap = grp_tcap; A.grp_z = 0;
    if (ndp + n16 + n32 + n64 + n128 < njobs) hipLaunchKernelGGL(k_gcig_nogap, dim3(grid_of(njobs, 256)), dim3(256), 0, ctx->stream, A);
    if (n16 > 0) {
        GcigArgs D = A;
        D.dp_list = d_dplist; D.njobs = n16; D.list_first = 0; D.grp_z = z16;
        hipLaunchKernelGGL(k_gcig_grp<16>, dim3((unsigned)((n16 + 3) / 4)), dim3(64), grp_lds(16, z16), ctx->stream, D);
    }
    if (n32 > 0) {
        GcigArgs D = A;
        D.dp_list = d_dplist; D.njobs = n32; D.list_first = n16; D.grp_z = z32;
        hipLaunchKernelGGL(k_gcig_grp<32>, dim3((unsigned)((n32 + 1) / 2)), dim3(64), grp_lds(32, z32), ctx->stream, D);
    }
    if (n64 > 0) {
        GcigArgs D = A;
        D.dp_list = d_dplist; D.njobs = n64; D.list_first = n16 + n32; D.grp_z = z64;
        hipLaunchKernelGGL(k_gcig_grp<64>, dim3((unsigned)n64), dim3(64), grp_lds(64, z64), ctx->stream, D);
    }
    if (n128 > 0) {
        GcigArgs D = A;
        D.dp_list = d_dplist; D.njobs = n128; D.list_first = n16 + n32 + n64;
        hipLaunchKernelGGL(k_gcig_t<true>, dim3((unsigned)n128), dim3(64), lds_two, ctx->stream, D);
    }
    ctx->tm.gcig_class_jobs[5] = n128;
    ctx->tm.gcig_class_jobs[0] = n16; ctx->tm.gcig_class_jobs[1] = n32; ctx->tm.gcig_class_jobs[2] = ndp; ctx->tm.gcig_class_jobs[3] = njobs - ndp - n16 - n32 - n64 - n128; ctx->tm.gcig_class_jobs[4] = n64;
    if (ndp > 0) {
        GcigArgs D = A;
        D.dp_list = d_dplist; D.njobs = ndp; D.list_first = n16 + n32 + n64 + n128;
        const size_t lds = lds_base + (size_t)zcap;
        if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_gcig_t<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_gcig_t<false>, dim3((unsigned)ndp), dim3(64), lds, ctx->stream, D);
    }
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_gcig_ncig, dim3(grid_of(njobs, 256)), dim3(256), 0, ctx->stream, (const meme_gres*)G[4].p, (i64)njobs, d_ncig);
    if ((rc = meme_scan_exclusive(ctx, d_ncig, d_ooff, njobs))) return rc;
    if (with_md) {
        hipLaunchKernelGGL(k_md_sizes, dim3(grid_of(njobs, 256)), dim3(256), 0, ctx->stream, (const int32_t*)A.mdlen, (i64)njobs, d_psz);
        if ((rc = meme_scan_exclusive(ctx, d_psz, d_poff, njobs))) return rc;
    }
    i64 tops = 0, tmd = 0;
    HIP_TRY(hipMemcpyAsync(&tops, d_ooff + njobs, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (with_md) HIP_TRY(hipMemcpyAsync(&tmd, d_poff + njobs, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if ((rc = meme_buf_reserve(ctx, G[5], (size_t)(tops + 1) * 4))) return rc;
    hipLaunchKernelGGL(k_gcig_pack, dim3(grid_of(njobs, 256)), dim3(256), 0, ctx->stream, (const meme_gres*)G[4].p, (const i64*)d_coff, (const uint32_t*)G[3].p,
                       (const i64*)d_ooff, (i64)njobs, (uint32_t*)G[5].p);
    hipLaunchKernelGGL(k_gcig_fix, dim3(grid_of(njobs, 256)), dim3(256), 0, ctx->stream, (meme_gres*)G[4].p, (const i64*)d_ooff, (i64)njobs);
    if (with_md) {
        if ((rc = meme_buf_reserve(ctx, G[8], (size_t)tmd + 64)) || (rc = meme_buf_reserve(ctx, G[10], (size_t)njobs * sizeof(meme_cres)))) return rc;
        hipLaunchKernelGGL(k_md_pack, dim3(grid_of(njobs, 256)), dim3(256), 0, ctx->stream, (const meme_gres*)G[4].p, (const int32_t*)A.nm, (const int32_t*)A.mdlen,
                           (const i64*)d_moff, (const char*)G[6].p, (const i64*)d_poff, (i64)njobs, (char*)G[8].p, (meme_cres*)G[10].p);
    }
    HIP_TRY(hipGetLastError());
    out->tops = tops; out->tmd = tmd;
    return MEME_OK;
}

int gcig_preamble(meme_ctx* ctx, const void* jobs, int64_t njobs, const meme_bsw_opt* opt, const void* out, const char* who) {
    if (!ctx || !jobs || njobs < 0 || !opt || !out) { meme_set_error("%s: null argument", who); return MEME_E_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    if (njobs == 0) return MEME_OK;
    const i64 nreads = ctx->last_seed_reads;
    if (nreads <= 0 || !ctx->reads.p || !ctx->read_off.p || !ctx->idx.pac || !ctx->reads_resident) { meme_set_error("%s: no seeded batch on this ctx (the jobs name its reads)", who); return MEME_E_STATE; }
    if (opt->e_del < 1 || opt->e_ins < 1) { meme_set_error("%s: gap extension penalties must be positive", who); return MEME_E_ARG; }
    if (ctx->max_batch > 0 && njobs > ctx->max_batch) { meme_set_error("%s: %lld jobs exceed the ctx's max_batch of %lld", who, (long long)njobs, (long long)ctx->max_batch); return MEME_E_CAPACITY; }
    for (int i = 0; i < 2; ++i) if (!ctx->ev_gcig[i]) HIP_TRY(hipEventCreate(&ctx->ev_gcig[i]));
    return MEME_OK;
}

}  // namespace

extern "C" int meme_global_batch_host(meme_ctx* ctx, const meme_gjob* jobs, int64_t njobs, const meme_bsw_opt* opt, meme_gres_host* out) {
    static const char* const who = "meme_global_batch_host";
    int rc;
    if ((rc = gcig_preamble(ctx, jobs, njobs, opt, out, who))) return rc;
    memset(out, 0, sizeof(*out));
    if (njobs == 0) return MEME_OK;
    const i64 nreads = ctx->last_seed_reads;
    int qmax = 0, tmax = 0;
    for (i64 k = 0; k < njobs; ++k) {
        const meme_gjob& J = jobs[k];
        // (the band must reach the matrix's last cell, w >= |tlen - qlen|, as every band bwa_gen_cigar2 computes does, src/bwa.cpp:306-316:
        // below that ksw_global2 walks rows without cells, which this kernel does not restate)
        if (J.read < 0 || J.read >= nreads || J.qb < 0 || J.qlen < 1 || J.tlen < 1 || J.w < 0 || J.rb < 0 || J.rb + J.tlen > ctx->idx.n || J.qlen > 65535 || J.tlen > 65535 ||
            J.w < (J.tlen > J.qlen ? J.tlen - J.qlen : J.qlen - J.tlen)) {
            meme_set_error("%s: job %lld is malformed (read %d, query %d+%d, target %lld+%d, band %d)", who, (long long)k, J.read, J.qb, J.qlen, (long long)J.rb, J.tlen, J.w);
            return MEME_E_ARG;
        }
        qmax = J.qlen > qmax ? J.qlen : qmax;
        tmax = J.tlen > tmax ? J.tlen : tmax;
    }
    DevBuf* G = ctx->gcig;
    if ((rc = meme_buf_reserve(ctx, G[0], (size_t)njobs * sizeof(meme_gjob)))) return rc;
    HIP_TRY(hipMemcpyAsync(G[0].p, jobs, (size_t)njobs * sizeof(meme_gjob), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev_gcig[0], ctx->stream));
    GcigRun R;
    if ((rc = gcig_run(ctx, njobs, qmax, tmax, opt, false, who, &R))) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_gcig[1], ctx->stream));
    meme_ctx::HostBuf* Hb = ctx->h_gcig;
    if ((rc = meme_hostbuf_reserve(ctx, Hb[0], (size_t)njobs * sizeof(meme_gres))) || (rc = meme_hostbuf_reserve(ctx, Hb[1], (size_t)(R.tops + 1) * 4))) return rc;
    HIP_TRY(hipMemcpyAsync(Hb[0].p, G[4].p, (size_t)njobs * sizeof(meme_gres), hipMemcpyDeviceToHost, ctx->stream));
    if (R.tops) HIP_TRY(hipMemcpyAsync(Hb[1].p, G[5].p, (size_t)R.tops * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_gcig[0], ctx->ev_gcig[1]));
    out->njobs = njobs; out->res = (const meme_gres*)Hb[0].p; out->cigars = (const uint32_t*)Hb[1].p; out->total_ops = R.tops; out->kernel_ms = ms;
    return MEME_OK;
}

extern "C" int meme_gen_cigar_batch_host(meme_ctx* ctx, const meme_cjob* jobs, int64_t njobs, const meme_bsw_opt* opt, meme_cres_host* out) {
    static const char* const who = "meme_gen_cigar_batch_host";
    int rc;
    if ((rc = gcig_preamble(ctx, jobs, njobs, opt, out, who))) return rc;
    memset(out, 0, sizeof(*out));
    if (njobs == 0) return MEME_OK;
    const i64 nreads = ctx->last_seed_reads, l_pac = ctx->idx.n >> 1;
    int qmax = 0, tmax = 0;
    for (i64 k = 0; k < njobs; ++k) {
        const meme_cjob& J = jobs[k];
        // what bwa_gen_cigar2 itself rejects (src/bwa.cpp:285: empty spans, a target bridging the two strands) is not a job
        if (J.read < 0 || J.read >= nreads || J.qb < 0 || J.qlen < 1 || J.tlen < 1 || J.w_ < 0 || J.rb < 0 || J.rb + J.tlen > ctx->idx.n || J.qlen > 65535 || J.tlen > 65535 ||
            (J.rb < l_pac && J.rb + J.tlen > l_pac)) {
            meme_set_error("%s: job %lld is malformed (read %d, query %d+%d, target %lld+%d, w_ %d)", who, (long long)k, J.read, J.qb, J.qlen, (long long)J.rb, J.tlen, J.w_);
            return MEME_E_ARG;
        }
        qmax = J.qlen > qmax ? J.qlen : qmax;
        tmax = J.tlen > tmax ? J.tlen : tmax;
    }
    DevBuf* G = ctx->gcig;
    if ((rc = meme_buf_reserve(ctx, G[0], (size_t)njobs * sizeof(meme_gjob))) || (rc = meme_buf_reserve(ctx, G[9], (size_t)njobs * sizeof(meme_cjob)))) return rc;
    HIP_TRY(hipMemcpyAsync(G[9].p, jobs, (size_t)njobs * sizeof(meme_cjob), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev_gcig[0], ctx->stream));
    hipLaunchKernelGGL(k_cjob_prep, dim3(grid_of(njobs, 256)), dim3(256), 0, ctx->stream, (const meme_cjob*)G[9].p, (i64)njobs, l_pac, *opt, (meme_gjob*)G[0].p);
    GcigRun R;
    if ((rc = gcig_run(ctx, njobs, qmax, tmax, opt, true, who, &R))) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_gcig[1], ctx->stream));
    meme_ctx::HostBuf* Hb = ctx->h_gcig;
    if ((rc = meme_hostbuf_reserve(ctx, Hb[0], (size_t)njobs * sizeof(meme_cres))) || (rc = meme_hostbuf_reserve(ctx, Hb[1], (size_t)(R.tops + 1) * 4)) ||
        (rc = meme_hostbuf_reserve(ctx, Hb[2], (size_t)R.tmd + 64))) return rc;
    HIP_TRY(hipMemcpyAsync(Hb[0].p, G[10].p, (size_t)njobs * sizeof(meme_cres), hipMemcpyDeviceToHost, ctx->stream));
    if (R.tops) HIP_TRY(hipMemcpyAsync(Hb[1].p, G[5].p, (size_t)R.tops * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (R.tmd) HIP_TRY(hipMemcpyAsync(Hb[2].p, G[8].p, (size_t)R.tmd, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_gcig[0], ctx->ev_gcig[1]));
    out->njobs = njobs; out->res = (const meme_cres*)Hb[0].p; out->cigars = (const uint32_t*)Hb[1].p; out->total_ops = R.tops; out->md = (const char*)Hb[2].p; out->md_bytes = R.tmd;
    out->kernel_ms = ms;
    return MEME_OK;
}
