// Mate rescue whole on the device (round 6): the POSING step of worker_sam's paired-end branch -- mem_sam_pe_batch_pre
// (reference src/bwamem_pair.cpp:660-716) with mem_matesw_batch_pre (:1060-1223) -- in front of the Smith-Waterman kernel that was here since
// round 3 (meme_kswv.hip: what mem_sam_pe_batch computes with kswv::getScores8 / getScores16).  Until now the binding ran the reference's own
// posing function for every worker batch on the host, copied the jobs' windows out of the 2-bit reference base by base and shipped ~600 bytes
// of sequence per job; the reads and the text are in HBM already.
//
// What the step does, per read pair and end i: the end's alignment records within pen_unpaired of its best one, at most max_matesw of them
// (:688-696); for each, the four orientations a mate could have (mem_infer_dir :58-65) minus those the insert-size statistics rule out
// (pes[r].failed) and those a record of the mate already explains (:1085-1091); for what is left the window of the reference the statistics
// point at, clamped to the strand and the reference sequence of its midpoint (bns_fetch_seq, src/bntseq.cpp:541-570); a job if the window
// lies on the record's sequence and holds min_seed_len bases (:1131).  The step's outputs are the jobs (in its own order: they are indexed by
// position) and `gar`, four job indices -- relative to the worker batch's first job -- or -1 per (end, record) looked at, which
// mem_sam_pe_batch_post reads to find its results (:1286-1296; -1 = compute it yourself).
//
// Mapping: one LANE per read (its records are few: two or three on average, 50 at most are looked at), two passes with a scan in between
// -- count (quads of gar, jobs, window bases, query bases per read), then write; a wavefront per job then unpacks the window from the 2-bit
// text and copies (or reverse-complements) the mate's bases, straight into the buffers k_kswv reads.  The jobs cross to the host once (32
// bytes each) for the sort into LDS classes the Smith-Waterman launches need; their sequences never do.
#include <string.h>
#include <vector>

#include "meme_common.h"

namespace {

struct MateAux { i64 rb; int read, is_rev; };                  // per job: window start (fwd+rc coordinate), the mate's read, reverse-complemented?

struct MateArgs {
    const meme_mate_reg* regs; const i64* reg_off; i64 n;       // n reads, the reads [first, first + n) of the resident batch (first even): pair p = reads 2p, 2p + 1
    i64 first;
    const i64* read_off;                                        // of the batch resident on the ctx
    meme_pestat pes[4];
    i64 l_pac; const i64* contig_off; const int* contig_len; int n_contigs;
    int a, pen_unpaired, max_matesw, min_seed_len, batch_reads;
    i64 *cntQ, *cntJ, *cntR, *cntY;                             // per read: (end, record) combinations looked at, jobs, window bases, query bases
    const i64 *offQ, *offJ, *offR, *offY;                       // their exclusive scans
    int32_t* gar; meme_kswv_job* jobs; MateAux* aux;
};

__device__ __forceinline__ int infer_dir(i64 l_pac, i64 b1, i64 b2, i64* dist) {          // mem_infer_dir, src/bwamem_pair.cpp:58-65
    const int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
    const i64 p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
    *dist = p2 > b1 ? p2 - b1 : b1 - p2;
    return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

template <bool WRITE>
__global__ void __launch_bounds__(256) k_mate_plan(MateArgs A) {
    for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < A.n; r += (i64)gridDim.x * 256) {
        const i64 m = r ^ 1;                                    // the mate (the pair's other read)
        const meme_mate_reg* a = A.regs + A.reg_off[r];
        const int na = (int)(A.reg_off[r + 1] - A.reg_off[r]);
        const meme_mate_reg* ma = A.regs + A.reg_off[m];
        const int nm = (int)(A.reg_off[m + 1] - A.reg_off[m]);
        const int l_ms = (int)(A.read_off[A.first + m + 1] - A.read_off[A.first + m]);
        const int xtra = 0x40000 | 0x80000 | (l_ms * A.a < 250 ? 0x10000 : 0) | (A.min_seed_len * A.a);      // KSW_XSUBO | KSW_XSTART | KSW_XBYTE | threshold (:1135)
        i64 nq = 0, nj = 0, nr = 0;
        i64 q0 = 0, j0 = 0, r0 = 0, y0 = 0, jb = 0;
        if (WRITE) { q0 = A.offQ[r]; j0 = A.offJ[r]; r0 = A.offR[r]; y0 = A.offY[r]; jb = A.offJ[r / A.batch_reads * A.batch_reads]; }
        const int top = na ? a[0].score : 0;
        for (int j = 0; j < na && nq < A.max_matesw; ++j) {
            const meme_mate_reg rec = a[j];
            if (!(rec.score >= top - A.pen_unpaired)) continue;                          // :690
            int skip = 0;                                                                 // bit o: orientation o is not tried
            for (int o = 0; o < 4; ++o) skip |= (A.pes[o].failed ? 1 : 0) << o;          // :1081-1083
            for (int k = 0; k < nm; ++k) {                                               // :1085-1091
                i64 dist;
                const int o = infer_dir(A.l_pac, rec.rb, ma[k].rb, &dist);
                if (dist >= A.pes[o].low && dist <= A.pes[o].high) skip |= 1 << o;
            }
            int rid = -1;                                                                 // (carried from one orientation to the next, as in the reference)
            for (int o = 0; o < 4; ++o) {
                int idx = -1;
                if (!(skip >> o & 1) && skip != 15) {
                    const int is_rev = (o >> 1) != (o & 1), is_larger = !(o >> 1);       // :1108-1109
                    i64 rb, re;
                    if (!is_rev) {                                                       // :1118-1125
                        rb = is_larger ? rec.rb + A.pes[o].low : rec.rb - A.pes[o].high;
                        re = (is_larger ? rec.rb + A.pes[o].high : rec.rb - A.pes[o].low) + l_ms;
                    } else {
                        rb = (is_larger ? rec.rb + A.pes[o].low : rec.rb - A.pes[o].high) - l_ms;
                        re = is_larger ? rec.rb + A.pes[o].high : rec.rb - A.pes[o].low;
                    }
                    if (rb < 0) rb = 0;
                    if (re > A.l_pac << 1) re = A.l_pac << 1;
                    if (rb < re) {                                                       // bns_fetch_seq: the sequence of the window's midpoint, its strand
                        const i64 mid = (rb + re) >> 1;
                        const bool mrev = mid >= A.l_pac;
                        const i64 fpos = mrev ? (A.l_pac << 1) - 1 - mid : mid;
                        int lo = 0, hi = A.n_contigs - 1;
                        while (lo < hi) { const int c = (lo + hi + 1) >> 1; if (A.contig_off[c] <= fpos) lo = c; else hi = c - 1; }
                        rid = lo;
                        i64 far_beg = A.contig_off[lo], far_end = far_beg + A.contig_len[lo];
                        if (mrev) { const i64 t = far_beg; far_beg = (A.l_pac << 1) - far_end; far_end = (A.l_pac << 1) - t; }
                        rb = rb > far_beg ? rb : far_beg;
                        re = re < far_end ? re : far_end;
                    }
                    if (rec.rid == rid && re - rb >= A.min_seed_len) {                   // :1131
                        if (WRITE) {
                            meme_kswv_job J;
                            J.idr = r0 + nr; J.idq = y0 + nj * l_ms; J.len1 = (int)(re - rb); J.len2 = l_ms; J.xtra = xtra; J.pad = 0;
                            A.jobs[j0 + nj] = J;
                            MateAux X; X.rb = rb; X.read = (int)(A.first + m); X.is_rev = is_rev;
                            A.aux[j0 + nj] = X;
                            idx = (int)(j0 + nj - jb);                                   // relative to the worker batch's first job: what the third step indexes with
                        }
                        ++nj; nr += re - rb;
                    }
                }
                if (WRITE) A.gar[4 * (q0 + nq) + o] = idx;
            }
            ++nq;
        }
        if (!WRITE) { A.cntQ[r] = nq; A.cntJ[r] = nj; A.cntR[r] = nr; A.cntY[r] = nj * l_ms; }
    }
}

// the sequences of the posed jobs, a wavefront per job: the window unpacked from the 2-bit fwd+rc text (what bns_get_seq returns, src/bntseq.cpp:515-539),
// the mate's bases as they are or reversed and complemented (:1111-1116)
__global__ void __launch_bounds__(256) k_mate_seq(const meme_kswv_job* __restrict__ jobs, const MateAux* __restrict__ aux, i64 n, const u64* __restrict__ pac,
                                                  const uint8_t* __restrict__ reads, const i64* __restrict__ read_off, uint8_t* __restrict__ ref, uint8_t* __restrict__ qer) {
    const i64 k = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const int lane = threadIdx.x & 63;
    const meme_kswv_job J = jobs[k];
    const MateAux X = aux[k];
    uint8_t* r = ref + J.idr;
    for (int l = lane; l < J.len1; l += 64) {
        const i64 p = X.rb + l;
        r[l] = (uint8_t)((pac[p >> 5] >> (62 - 2 * (int)(p & 31))) & 3ull);
    }
    const uint8_t* ms = reads + read_off[X.read];
    uint8_t* q = qer + J.idq;
    for (int l = lane; l < J.len2; l += 64) {
        uint8_t c;
        if (X.is_rev) { const uint8_t b = ms[J.len2 - 1 - l]; c = b < 4 ? (uint8_t)(3 - b) : (uint8_t)4; }
        else c = ms[l];
        q[l] = c;
    }
}

// first gar entry and first job of every worker batch (batch b = reads [b * batch_reads, ...)), + the totals behind the last one
__global__ void k_mate_batch_offs(const i64* __restrict__ offQ, const i64* __restrict__ offJ, i64 n, int batch_reads, i64 nb, i64* __restrict__ gar_off, i64* __restrict__ job_off) {
    const i64 b = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    const i64 r = b * batch_reads < n ? b * batch_reads : n;
    gar_off[b] = 4 * offQ[r];
    job_off[b] = offJ[r];
}

unsigned blocks_for(i64 items, int per) { i64 b = (items + per - 1) / per; const i64 cap = 256 * 32; return (unsigned)(b < cap ? (b < 1 ? 1 : b) : cap); }

}  // namespace

extern "C" int meme_matesw_batch_host(meme_ctx* ctx, meme_ctx* reads_of, const meme_mate_reg* regs, const int64_t* reg_off, int64_t first_read, int64_t nreads, const meme_pestat* pes,
                                      const meme_contig* contigs, int32_t n_contigs, int64_t l_pac, const meme_mate_opt* opt, meme_mate_host_result* out) {
    static const char* const who = "meme_matesw_batch_host";
    if (!ctx || !reg_off || !pes || !contigs || n_contigs < 1 || !opt || !out || nreads < 0) { meme_set_error("%s: null argument", who); return MEME_E_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    memset(out, 0, sizeof(*out));
    if (nreads == 0) return MEME_OK;
    // the batch whose reads are named: this ctx's, or -- reads_of -- that of another ctx of the same GPU (a binding that runs this stage beside another one
    // on the ctx that seeded the batch: the bases are only read, device pointers are valid across the ctxs of a device)
    meme_ctx* const rc_ = reads_of ? reads_of : ctx;
    if (rc_->device != ctx->device) { meme_set_error("%s: the ctx that holds the reads is on another device", who); return MEME_E_ARG; }
    if (first_read < 0 || first_read + nreads > rc_->last_seed_reads || !rc_->reads_resident || !rc_->reads.p || !rc_->read_off.p) {
        meme_set_error("%s: reads [%lld, %lld) must lie in the batch resident on the ctx (a seeding call stages it; it holds %lld)", who, (long long)first_read, (long long)(first_read + nreads),
                       (long long)rc_->last_seed_reads);
        return MEME_E_STATE;
    }
    if ((nreads & 1) || (first_read & 1)) { meme_set_error("%s: an odd number of reads (or an odd first read) is not a set of pairs", who); return MEME_E_ARG; }
    if (l_pac * 2 != ctx->idx.n) { meme_set_error("%s: l_pac does not match the loaded index", who); return MEME_E_ARG; }
    if (opt->batch_reads < 2 || (opt->batch_reads & 1) || opt->max_matesw < 0 || opt->a < 1 || opt->min_seed_len < 1) { meme_set_error("%s: bad options", who); return MEME_E_ARG; }
    const i64 nrec = reg_off[nreads];
    if (reg_off[0] != 0 || nrec < 0 || (nrec > 0 && !regs)) { meme_set_error("%s: bad record offsets", who); return MEME_E_ARG; }
    for (int i = 0; i < n_contigs; ++i)
        if (contigs[i].len < 1 || contigs[i].offset < 0 || contigs[i].offset + contigs[i].len > l_pac || (i > 0 && contigs[i].offset < contigs[i - 1].offset + contigs[i - 1].len)) {
            meme_set_error("%s: contig %d is not a valid reference sequence", who, i);
            return MEME_E_ARG;
        }
    const i64 n = nreads, nb = (n + opt->batch_reads - 1) / opt->batch_reads;
    int rc;
    DevBuf* M = ctx->mate;       // 0 records, 1 record offsets, 2 contig table, 3 counts + scans (8 x (n + 1)), 4 gar, 5 aux, 6 batch offsets
    if ((rc = meme_buf_reserve(ctx, M[0], (size_t)(nrec + 1) * sizeof(meme_mate_reg))) || (rc = meme_buf_reserve(ctx, M[1], (size_t)(n + 1) * 8)) ||
        (rc = meme_buf_reserve(ctx, M[2], (size_t)n_contigs * 12 + 64)) || (rc = meme_buf_reserve(ctx, M[3], (size_t)(n + 1) * 8 * 8)) ||
        (rc = meme_buf_reserve(ctx, M[6], (size_t)(nb + 1) * 16))) return rc;
    std::vector<unsigned char> tab((size_t)n_contigs * 12 + 64, 0);
    {
        i64* t_off = (i64*)tab.data();
        int* t_len = (int*)(tab.data() + (size_t)n_contigs * 8);
        for (int i = 0; i < n_contigs; ++i) { t_off[i] = contigs[i].offset; t_len[i] = contigs[i].len; }
    }
    hipEvent_t* ev = ctx->ev_kswv;
    for (int i = 0; i < 2; ++i) if (!ev[i]) HIP_TRY(hipEventCreate(&ev[i]));
    if (nrec) HIP_TRY(hipMemcpyAsync(M[0].p, regs, (size_t)nrec * sizeof(meme_mate_reg), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(M[1].p, reg_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(M[2].p, tab.data(), tab.size(), hipMemcpyHostToDevice, ctx->stream));
    MateArgs A;
    memset(&A, 0, sizeof(A));
    A.regs = (const meme_mate_reg*)M[0].p; A.reg_off = (const i64*)M[1].p; A.n = n; A.first = first_read; A.read_off = (const i64*)rc_->read_off.p;
    for (int o = 0; o < 4; ++o) A.pes[o] = pes[o];
    A.l_pac = l_pac; A.contig_off = (const i64*)M[2].p; A.contig_len = (const int*)((unsigned char*)M[2].p + (size_t)n_contigs * 8); A.n_contigs = n_contigs;
    A.a = opt->a; A.pen_unpaired = opt->pen_unpaired; A.max_matesw = opt->max_matesw; A.min_seed_len = opt->min_seed_len; A.batch_reads = opt->batch_reads;
    i64* C = (i64*)M[3].p;
    A.cntQ = C; A.cntJ = C + (n + 1); A.cntR = C + 2 * (n + 1); A.cntY = C + 3 * (n + 1);
    i64* O = C + 4 * (n + 1);
    A.offQ = O; A.offJ = O + (n + 1); A.offR = O + 2 * (n + 1); A.offY = O + 3 * (n + 1);
    HIP_TRY(hipEventRecord(ev[0], ctx->stream));
    hipLaunchKernelGGL((k_mate_plan<false>), dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, A);
    for (int k = 0; k < 4; ++k) if ((rc = meme_scan_exclusive(ctx, C + k * (n + 1), O + k * (n + 1), n))) return rc;
    i64 tot[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) HIP_TRY(hipMemcpyAsync(&tot[k], O + k * (n + 1) + n, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));                                       // (also: `tab` is a local)
    const i64 nq = tot[0], nj = tot[1], ref_bytes = tot[2], qer_bytes = tot[3];
    if (nj > 0x7fffffff / 2 || 4 * nq > 0x7fffffff) { meme_set_error("%s: too many jobs in one call", who); return MEME_E_CAPACITY; }
    DevBuf* K = ctx->kswv;        // (k_kswv's buffers: 0 jobs, 2 window bases, 3 query bases -- written here, read there)
    meme_ctx::HostBuf* H = ctx->h_mate;      // 0 gar, 1 batch offsets (gar, jobs), 2 jobs
    if ((rc = meme_buf_reserve(ctx, M[4], (size_t)(4 * nq + 4) * 4)) || (rc = meme_buf_reserve(ctx, M[5], (size_t)(nj + 1) * sizeof(MateAux))) ||
        (rc = meme_buf_reserve(ctx, K[0], (size_t)(nj + 1) * sizeof(meme_kswv_job))) || (rc = meme_buf_reserve(ctx, K[2], (size_t)ref_bytes + 64)) ||
        (rc = meme_buf_reserve(ctx, K[3], (size_t)qer_bytes + 64)) || (rc = meme_hostbuf_reserve(ctx, H[0], (size_t)(4 * nq + 4) * 4)) ||
        (rc = meme_hostbuf_reserve(ctx, H[1], (size_t)(nb + 1) * 16)) || (rc = meme_hostbuf_reserve(ctx, H[2], (size_t)(nj + 1) * sizeof(meme_kswv_job)))) return rc;
    A.gar = (int32_t*)M[4].p; A.jobs = (meme_kswv_job*)K[0].p; A.aux = (MateAux*)M[5].p;
    hipLaunchKernelGGL((k_mate_plan<true>), dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, A);
    if (nj) hipLaunchKernelGGL(k_mate_seq, dim3((unsigned)((nj + 3) / 4)), dim3(256), 0, ctx->stream, (const meme_kswv_job*)K[0].p, (const MateAux*)M[5].p, nj, ctx->idx.pac,
                               (const uint8_t*)rc_->reads.p, (const i64*)rc_->read_off.p, (uint8_t*)K[2].p, (uint8_t*)K[3].p);
    i64* d_goff = (i64*)M[6].p;
    i64* d_joff = d_goff + (nb + 1);
    hipLaunchKernelGGL(k_mate_batch_offs, dim3((unsigned)((nb + 256) / 256)), dim3(256), 0, ctx->stream, A.offQ, A.offJ, n, opt->batch_reads, nb, d_goff, d_joff);
    HIP_TRY(hipGetLastError());
    if (nq) HIP_TRY(hipMemcpyAsync(H[0].p, M[4].p, (size_t)(4 * nq) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(H[1].p, M[6].p, (size_t)(nb + 1) * 16, hipMemcpyDeviceToHost, ctx->stream));
    if (nj) HIP_TRY(hipMemcpyAsync(H[2].p, K[0].p, (size_t)nj * sizeof(meme_kswv_job), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipEventRecord(ev[1], ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float pose_ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&pose_ms, ev[0], ev[1]));
    meme_kswv_host_result R;
    memset(&R, 0, sizeof(R));
    if (nj) {
        meme_bsw_opt bo;
        memset(&bo, 0, sizeof(bo));
        bo.a = opt->a; bo.b = opt->b; bo.o_del = opt->o_del; bo.e_del = opt->e_del; bo.o_ins = opt->o_ins; bo.e_ins = opt->e_ins;
        if ((rc = meme_kswv_run(ctx, (const meme_kswv_job*)H[2].p, nj, nullptr, ref_bytes, nullptr, qer_bytes, &bo, true, &R))) return rc;
    }
    out->nreads = n; out->nbatches = nb; out->njobs = nj; out->n_gar = 4 * nq;
    out->gar = (const int32_t*)H[0].p; out->gar_off = (const int64_t*)H[1].p; out->job_off = (const int64_t*)H[1].p + (nb + 1);
    out->jobs = (const meme_kswv_job*)H[2].p; out->res = R.res; out->pose_ms = pose_ms; out->kernel_ms = R.kernel_ms;
    return MEME_OK;
}
