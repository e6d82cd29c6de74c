// SAM text on the device -- mem_aln2sam (reference src/bwamem.cpp:2174-2312, add_cigar :2161-2172, get_rlen :2402-2410) for the plain records of a
// chunk: the only record of its read, no comment, no pa / SA / XR tag.  The host's per-alignment text work -- the last stage of the
// SAM phase that was still the reference's -- becomes one descriptor per record; names, qualities and bases are in HBM already.
//
// One wavefront per record.  The header fields (name, flag, contig, position, mapping quality, CIGAR, mate fields) and the tags are a few dozen
// bytes written by lane 0; SEQ and QUAL -- two thirds of a record -- are written by the 64 lanes side by side, reversed and complemented on
// the reverse strand.  Every record is written into a scratch slot of its own (an upper bound of its length is known from its parts); a
// second kernel packs the texts back to back in record order.
#include <string.h>

#include "meme_common.h"

namespace {

struct SamArgs {
    const meme_sam_rec* recs; i64 nrecs;
    const uint8_t* blob;
    const uint8_t* reads; const i64* read_off;          // bases (codes) of the batch resident on the ctx
    const char* names; const i64* name_off;             // staged by meme_sam_stage_text
    const char* quals;                                  // same offsets as the bases; null: no qualities
    const char* contig_names; const int32_t* contig_name_off;
    int softclip;
    const char* rg; int rg_len;
    const i64* soff; char* scratch;                      // per record: its scratch slot
    i64* len;                                           // per record: bytes written
    i64* over;                                          // smallest record index whose text did not fit its scratch slot (a formatter / bound mismatch: fails the call)
};

__device__ __forceinline__ int put_num(char* o, long long v) {        // kputw / kputl
    char buf[24];
    int l = 0, n = 0;
    unsigned long long x = v < 0 ? (unsigned long long)(-v) : (unsigned long long)v;
    do { buf[l++] = (char)('0' + (int)(x % 10)); x /= 10; } while (x);
    if (v < 0) o[n++] = '-';
    while (l) o[n++] = buf[--l];
    return n;
}
__device__ __forceinline__ int put_str(char* o, const char* s, int l) { for (int i = 0; i < l; ++i) o[i] = s[i]; return l; }
__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }   // (blob offsets need not be aligned)
__device__ int put_cigar(char* o, const uint8_t* cg, int n_cigar, int softclip, int is_alt, int which) {      // add_cigar
    int n = 0;
    if (!n_cigar) { o[n++] = '*'; return n; }
    for (int i = 0; i < n_cigar; ++i) {
        const uint32_t c0 = ld_u32(cg + 4 * i);
        int c = (int)(c0 & 0xf);
        if (!softclip && !is_alt && (c == 3 || c == 4)) c = which ? 4 : 3;
        n += put_num(o + n, (long long)(c0 >> 4));
        o[n++] = "MIDSH"[c];
    }
    return n;
}
__device__ long long ref_len(int n_cigar, const uint8_t* cg) {      // get_rlen
    long long l = 0;
    for (int k = 0; k < n_cigar; ++k) { const uint32_t c = ld_u32(cg + 4 * k); const int op = (int)(c & 0xf); if (op == 0 || op == 2) l += c >> 4; }
    return l;
}

__global__ void __launch_bounds__(256) k_sam_format(SamArgs A) {
    const i64 k = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);         // a wavefront per record, four to a workgroup (one-wave workgroups are dispatch-bound)
    if (k >= A.nrecs) return;
    const int lane = threadIdx.x & 63;
    const meme_sam_rec R = A.recs[k];
    if (R.read < 0) { if (lane == 0) A.len[k] = 0; return; }        // an empty slot (the caller submits one per read, in read order)
    char* o = A.scratch + A.soff[k];
    const i64 rb0 = A.read_off[R.read];
    const int l_seq = (int)(A.read_off[R.read + 1] - rb0);
    const uint8_t* seq = A.reads + rb0;
    const char* qual = A.quals ? A.quals + rb0 : nullptr;
    const i64 n0 = A.name_off[R.read];
    const int l_name = (int)(A.name_off[R.read + 1] - n0);
    const uint8_t* cg = A.blob + (R.n_cigar > 0 ? R.cigar_off : 0);
    const uint8_t* mcg = A.blob + (R.m_n_cigar > 0 ? R.m_cigar_off : 0);
    // the record and its mate as mem_aln2sam sees them after its first lines (:2181-2192): every lane the same values
    int flag = R.flag, rid = R.rid, is_rev = R.is_rev, n_cigar = R.n_cigar;
    i64 pos = R.pos;
    const int has_m = R.has_mate;
    int m_rid = R.m_rid, m_is_rev = R.m_is_rev, m_n_cigar = R.m_n_cigar;
    i64 m_pos = R.m_pos;
    flag |= has_m ? 0x1 : 0;
    flag |= rid < 0 ? 0x4 : 0;
    flag |= has_m && m_rid < 0 ? 0x8 : 0;
    if (rid < 0 && has_m && m_rid >= 0) { rid = m_rid; pos = m_pos; is_rev = m_is_rev; n_cigar = 0; }
    if (has_m && m_rid < 0 && rid >= 0) { m_rid = rid; m_pos = pos; m_is_rev = is_rev; m_n_cigar = 0; }
    flag |= is_rev ? 0x10 : 0;
    flag |= has_m && m_is_rev ? 0x20 : 0;
    // ---- QNAME .. TLEN: lane 0 (the name by all lanes)
    for (int i = lane; i < l_name; i += 64) o[i] = A.names[n0 + i];
    int n = l_name;
    int hdr = 0;                              // bytes lane 0 writes behind the name
    if (lane == 0) {
        char* h = o + n;
        int m = 0;
        h[m++] = '\t';
        m += put_num(h + m, (flag & 0xffff) | (flag & 0x10000 ? 0x100 : 0)); h[m++] = '\t';
        if (rid >= 0) {
            m += put_str(h + m, A.contig_names + A.contig_name_off[rid], A.contig_name_off[rid + 1] - A.contig_name_off[rid]); h[m++] = '\t';
            m += put_num(h + m, pos + 1); h[m++] = '\t';
            m += put_num(h + m, R.mapq); h[m++] = '\t';
            m += put_cigar(h + m, cg, n_cigar, A.softclip, R.is_alt, R.which);
        } else m += put_str(h + m, "*\t0\t0\t*", 7);
        h[m++] = '\t';
        if (has_m && m_rid >= 0) {
            if (rid == m_rid) h[m++] = '=';
            else m += put_str(h + m, A.contig_names + A.contig_name_off[m_rid], A.contig_name_off[m_rid + 1] - A.contig_name_off[m_rid]);
            h[m++] = '\t';
            m += put_num(h + m, m_pos + 1); h[m++] = '\t';
            if (rid == m_rid) {
                const long long p0 = pos + (is_rev ? ref_len(n_cigar, cg) - 1 : 0);
                const long long p1 = m_pos + (m_is_rev ? ref_len(m_n_cigar, mcg) - 1 : 0);
                if (m_n_cigar == 0 || n_cigar == 0) h[m++] = '0';
                else m += put_num(h + m, -(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
            } else h[m++] = '0';
        } else m += put_str(h + m, "*\t0\t0", 5);
        h[m++] = '\t';
        hdr = m;
    }
    n += __shfl(hdr, 0);
    // ---- SEQ and QUAL: all lanes
    if (flag & 0x100) { if (lane == 0) { o[n] = '*'; o[n + 1] = '\t'; o[n + 2] = '*'; } n += 3; }
    else {
        int qb = 0, qe = l_seq;
        if (n_cigar && R.which && !A.softclip && !R.is_alt) {
            const uint32_t cf = ld_u32(cg), cl = ld_u32(cg + 4 * (n_cigar - 1));
            const int c0 = (int)(cf & 0xf), c1 = (int)(cl & 0xf);
            if (!is_rev) { if (c0 == 4 || c0 == 3) qb += (int)(cf >> 4); if (c1 == 4 || c1 == 3) qe -= (int)(cl >> 4); }
            else { if (c0 == 4 || c0 == 3) qe -= (int)(cf >> 4); if (c1 == 4 || c1 == 3) qb += (int)(cl >> 4); }
        }
        const int m = qe > qb ? qe - qb : 0;
        for (int i = lane; i < m; i += 64) {
            const int c = seq[is_rev ? qe - 1 - i : qb + i];
            o[n + i] = (is_rev ? "TGCAN" : "ACGTN")[c > 4 ? 4 : c];
        }
        n += m;
        if (lane == 0) o[n] = '\t';
        ++n;
        if (qual) {
            for (int i = lane; i < m; i += 64) o[n + i] = qual[is_rev ? qe - 1 - i : qb + i];
            n += m;
        } else { if (lane == 0) o[n] = '*'; ++n; }
    }
    // ---- tags: lane 0 (MD and XA strings by all lanes)
    if (n_cigar) {
        int m = 0;
        if (lane == 0) { m += put_str(o + n, "\tNM:i:", 6); m += put_num(o + n + m, R.NM); m += put_str(o + n + m, "\tMD:Z:", 6); }
        n += __shfl(m, 0);
        const uint8_t* md = cg + 4 * (i64)R.n_cigar;
        int l = 0;
        for (;;) {                            // strlen, 64 bytes at a time
            const bool z = md[l + lane] == 0;
            const unsigned long long zm = __ballot(z);
            if (zm) { l += __ffsll((long long)zm) - 1; break; }
            l += 64;
        }
        for (int i = lane; i < l; i += 64) o[n + i] = (char)md[i];
        n += l;
    }
    {
        int m = 0;
        if (lane == 0) {
            if (has_m && m_n_cigar) { m += put_str(o + n + m, "\tMC:Z:", 6); m += put_cigar(o + n + m, mcg, m_n_cigar, A.softclip, R.m_is_alt, R.which); }
            if (R.score >= 0) { m += put_str(o + n + m, "\tAS:i:", 6); m += put_num(o + n + m, R.score); }
            if (R.sub >= 0) { m += put_str(o + n + m, "\tXS:i:", 6); m += put_num(o + n + m, R.sub); }
            if (A.rg_len) { m += put_str(o + n + m, "\tRG:Z:", 6); m += put_str(o + n + m, A.rg, A.rg_len); }
            if (R.xa_off >= 0) m += put_str(o + n + m, "\tXA:Z:", 6);
        }
        n += __shfl(m, 0);
    }
    if (R.xa_off >= 0) {
        const uint8_t* xa = A.blob + R.xa_off;
        int l = 0;
        for (;;) {
            const bool z = xa[l + lane] == 0;
            const unsigned long long zm = __ballot(z);
            if (zm) { l += __ffsll((long long)zm) - 1; break; }
            l += 64;
        }
        for (int i = lane; i < l; i += 64) o[n + i] = (char)xa[i];
        n += l;
    }
    if (lane == 0) {
        o[n] = '\n'; A.len[k] = n + 1;
        // k_sam_bounds sized the slot from the record's parts + a fixed allowance for the numeric fields and tags: a text beyond it has
        // written into the next record's slot -- never silently (advisor, round 5)
        if ((i64)n + 1 > A.soff[k + 1] - A.soff[k]) atomicMin((unsigned long long*)A.over, (unsigned long long)k);
    }
}

// upper bound of a record's text: name + SEQ + QUAL + 12 bytes per CIGAR operation (its own, the mate's in MC) + the strings + the fixed parts
__global__ void __launch_bounds__(256) k_sam_bounds(const meme_sam_rec* __restrict__ recs, i64 n, const i64* __restrict__ read_off, const i64* __restrict__ name_off,
                                                    const uint8_t* __restrict__ blob, i64 blob_bytes, int max_contig, int rg_len, i64* __restrict__ bound, i64* __restrict__ bad) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (i64)gridDim.x * blockDim.x) {
        const meme_sam_rec R = recs[k];
        if (R.read < 0) { bound[k] = 0; continue; }
        const i64 l_seq = read_off[R.read + 1] - read_off[R.read], l_name = name_off[R.read + 1] - name_off[R.read];
        i64 b = l_name + 2 * l_seq + 12 * ((i64)(R.n_cigar > 0 ? R.n_cigar : 0) + (R.m_n_cigar > 0 ? R.m_n_cigar : 0)) + 2 * max_contig + rg_len + 192;
        bool ok = true;
        if (R.n_cigar > 0) {                  // the MD string behind the operations must end inside the blob
            i64 p = R.cigar_off + 4 * (i64)R.n_cigar;
            while (p < blob_bytes && blob[p]) ++p;
            ok = ok && p < blob_bytes;
            b += p - (R.cigar_off + 4 * (i64)R.n_cigar);
        }
        if (R.xa_off >= 0) {
            i64 p = R.xa_off;
            while (p < blob_bytes && blob[p]) ++p;
            ok = ok && p < blob_bytes;
            b += p - R.xa_off;
        }
        if (!ok) atomicMin((unsigned long long*)bad, (unsigned long long)k);
        bound[k] = (b + 15) & ~(i64)15;
    }
}
__global__ void __launch_bounds__(256) k_sam_pack(const i64* __restrict__ soff, const char* __restrict__ scratch, const i64* __restrict__ toff, i64 n, char* __restrict__ out) {
    const i64 k = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const i64 l = toff[k + 1] - toff[k];
    const char* src = scratch + soff[k];
    char* dst = out + toff[k];
    for (i64 i = threadIdx.x & 63; i < l; i += 64) dst[i] = src[i];
}

unsigned grid_of(i64 items, int per) { i64 b = (items + per - 1) / per; const i64 cap = 256 * 64; return (unsigned)(b < cap ? (b < 1 ? 1 : b) : cap); }

}  // namespace

extern "C" int meme_sam_stage_text(meme_ctx* ctx, const char* names, const int64_t* name_off, const char* quals) {
    if (!ctx || !names || !name_off) { meme_set_error("meme_sam_stage_text: null argument"); return MEME_E_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    const i64 n = ctx->last_seed_reads;
    if (n <= 0 || !ctx->reads_resident || !ctx->read_off.p) { meme_set_error("meme_sam_stage_text: no seeded batch on this ctx"); return MEME_E_STATE; }
    if (name_off[0] != 0) { meme_set_error("meme_sam_stage_text: name_off[0] must be 0"); return MEME_E_ARG; }
    for (i64 r = 0; r < n; ++r) if (name_off[r + 1] < name_off[r]) { meme_set_error("meme_sam_stage_text: name offsets must not decrease (read %lld)", (long long)r); return MEME_E_ARG; }
    int rc;
    DevBuf* S = ctx->sam;        // 0 names, 1 name offsets, 2 qualities, 3 records, 4 blob, 5 bounds + offsets + lengths, 6 scratch, 7 text, 8 contig names + offsets + rg
    const i64 nb = name_off[n];
    if ((rc = meme_buf_reserve(ctx, S[0], (size_t)nb + 64)) || (rc = meme_buf_reserve(ctx, S[1], (size_t)(n + 1) * 8))) return rc;
    HIP_TRY(hipMemcpyAsync(S[0].p, names, (size_t)nb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(S[1].p, name_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    ctx->sam_has_quals = quals != nullptr;
    if (quals) {
        i64 total = 0;
        HIP_TRY(hipMemcpyAsync(&total, (const i64*)ctx->read_off.p + n, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if ((rc = meme_buf_reserve(ctx, S[2], (size_t)total + 64))) return rc;
        HIP_TRY(hipMemcpyAsync(S[2].p, quals, (size_t)total, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->sam_text_reads = n;
    return MEME_OK;
}

extern "C" int meme_sam_format_batch_host(meme_ctx* ctx, const meme_sam_rec* recs, int64_t nrecs, const uint8_t* blob, int64_t blob_bytes, const char* contig_names,
                                          const int32_t* contig_name_off, int32_t n_contigs, int32_t softclip, const char* rg_id, meme_sam_host_result* out) {
    static const char* const who = "meme_sam_format_batch_host";
    if (!ctx || !out || nrecs < 0 || blob_bytes < 0 || (nrecs > 0 && (!recs || !contig_names || !contig_name_off)) || n_contigs < 0 || (blob_bytes > 0 && !blob)) {
        meme_set_error("%s: null argument", who);
        return MEME_E_ARG;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    memset(out, 0, sizeof(*out));
    if (nrecs == 0) return MEME_OK;
    if (ctx->sam_max_batch > 0 && nrecs > ctx->sam_max_batch) { meme_set_error("%s: %lld record slots exceed the ctx's sam_max_batch of %lld", who, (long long)nrecs, (long long)ctx->sam_max_batch); return MEME_E_CAPACITY; }
    const i64 nreads = ctx->last_seed_reads;
    if (nreads <= 0 || !ctx->reads_resident || !ctx->reads.p || ctx->sam_text_reads != nreads) {
        meme_set_error("%s: the batch on this ctx has no names / qualities staged (meme_sam_stage_text after the seeding call)", who);
        return MEME_E_STATE;
    }
    int max_contig = 0;
    for (int i = 0; i < n_contigs; ++i) {
        if (contig_name_off[i + 1] < contig_name_off[i]) { meme_set_error("%s: contig name offsets must not decrease", who); return MEME_E_ARG; }
        max_contig = contig_name_off[i + 1] - contig_name_off[i] > max_contig ? contig_name_off[i + 1] - contig_name_off[i] : max_contig;
    }
    for (i64 k = 0; k < nrecs; ++k) {
        const meme_sam_rec& R = recs[k];
        if (R.read < 0) continue;                                    // an empty slot: no text
        if (R.read >= nreads || R.rid >= n_contigs || R.m_rid >= n_contigs || R.n_cigar < 0 || R.m_n_cigar < 0 || R.which < 0 ||
            (R.n_cigar > 0 && (R.cigar_off < 0 || R.cigar_off + 4 * (i64)R.n_cigar >= blob_bytes)) ||
            (R.m_n_cigar > 0 && (R.m_cigar_off < 0 || R.m_cigar_off + 4 * (i64)R.m_n_cigar > blob_bytes)) || (R.xa_off >= blob_bytes)) {
            meme_set_error("%s: record %lld is malformed (read %d, contigs %d / %d, %d + %d operations)", who, (long long)k, R.read, R.rid, R.m_rid, R.n_cigar, R.m_n_cigar);
            return MEME_E_ARG;
        }
    }
    const int rg_len = rg_id ? (int)strlen(rg_id) : 0;
    int rc;
    DevBuf* S = ctx->sam;
    const size_t ctab = (size_t)contig_name_off[n_contigs] + (size_t)(n_contigs + 1) * 4 + (size_t)rg_len + 64;
    if ((rc = meme_buf_reserve(ctx, S[3], (size_t)nrecs * sizeof(meme_sam_rec))) || (rc = meme_buf_reserve(ctx, S[4], (size_t)blob_bytes + 128)) ||
        (rc = meme_buf_reserve(ctx, S[5], (size_t)(nrecs + 1) * 8 * 4 + 64)) || (rc = meme_buf_reserve(ctx, S[8], ctab))) return rc;
    for (int i = 0; i < 2; ++i) if (!ctx->ev_sam[i]) HIP_TRY(hipEventCreate(&ctx->ev_sam[i]));
    HIP_TRY(hipMemcpyAsync(S[3].p, recs, (size_t)nrecs * sizeof(meme_sam_rec), hipMemcpyHostToDevice, ctx->stream));
    if (blob_bytes) HIP_TRY(hipMemcpyAsync(S[4].p, blob, (size_t)blob_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync((char*)S[4].p + blob_bytes, 0, 128, ctx->stream));            // (the strlen loops read 64 bytes at a time)
    // contig table: offsets (4-byte aligned first), names, read group
    char* d_tab = (char*)S[8].p;
    const size_t off_bytes = (size_t)(n_contigs + 1) * 4;
    HIP_TRY(hipMemcpyAsync(d_tab, contig_name_off, off_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (contig_name_off[n_contigs]) HIP_TRY(hipMemcpyAsync(d_tab + off_bytes, contig_names, (size_t)contig_name_off[n_contigs], hipMemcpyHostToDevice, ctx->stream));
    if (rg_len) HIP_TRY(hipMemcpyAsync(d_tab + off_bytes + contig_name_off[n_contigs], rg_id, (size_t)rg_len, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev_sam[0], ctx->stream));
    i64* d_bound = (i64*)S[5].p;
    i64* d_soff = d_bound + (nrecs + 1);
    i64* d_len = d_soff + (nrecs + 1);
    i64* d_toff = d_len + (nrecs + 1);
    i64* d_bad = d_toff + (nrecs + 1);
    HIP_TRY(hipMemsetAsync(d_bad, 0xff, 8, ctx->stream));
    hipLaunchKernelGGL(k_sam_bounds, dim3(grid_of(nrecs, 256)), dim3(256), 0, ctx->stream, (const meme_sam_rec*)S[3].p, (i64)nrecs, (const i64*)ctx->read_off.p, (const i64*)S[1].p,
                       (const uint8_t*)S[4].p, (i64)blob_bytes, max_contig, rg_len, d_bound, d_bad);
    if ((rc = meme_scan_exclusive(ctx, d_bound, d_soff, nrecs))) return rc;
    i64 total_scratch = 0, bad = -1;
    HIP_TRY(hipMemcpyAsync(&total_scratch, d_soff + nrecs, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (bad >= 0) { meme_set_error("%s: record %lld names a string that does not end inside the blob", who, (long long)bad); return MEME_E_ARG; }
    if ((rc = meme_buf_reserve(ctx, S[6], (size_t)total_scratch + 64))) return rc;
    SamArgs A;
    A.recs = (const meme_sam_rec*)S[3].p; A.nrecs = nrecs; A.blob = (const uint8_t*)S[4].p; A.reads = (const uint8_t*)ctx->reads.p; A.read_off = (const i64*)ctx->read_off.p;
    A.names = (const char*)S[0].p; A.name_off = (const i64*)S[1].p; A.quals = ctx->sam_has_quals ? (const char*)S[2].p : nullptr;
    A.contig_name_off = (const int32_t*)d_tab; A.contig_names = d_tab + off_bytes; A.softclip = softclip ? 1 : 0;
    A.rg = d_tab + off_bytes + contig_name_off[n_contigs]; A.rg_len = rg_len;
    A.soff = d_soff; A.scratch = (char*)S[6].p; A.len = d_len; A.over = d_bad;            // (d_bad is all-ones again: the bounds kernel found nothing)
    hipLaunchKernelGGL(k_sam_format, dim3((unsigned)((nrecs + 3) / 4)), dim3(256), 0, ctx->stream, A);
    if ((rc = meme_scan_exclusive(ctx, d_len, d_toff, nrecs))) return rc;
    i64 total = 0, over = -1;
    HIP_TRY(hipMemcpyAsync(&total, d_toff + nrecs, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&over, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (over >= 0) { meme_set_error("%s: the text of record %lld is longer than the slot k_sam_bounds gave it (formatter and bound disagree)", who, (long long)over); return MEME_E_STATE; }
    if ((rc = meme_buf_reserve(ctx, S[7], (size_t)total + 64))) return rc;
    hipLaunchKernelGGL(k_sam_pack, dim3((unsigned)((nrecs + 3) / 4)), dim3(256), 0, ctx->stream, (const i64*)d_soff, (const char*)S[6].p, (const i64*)d_toff, (i64)nrecs, (char*)S[7].p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_sam[1], ctx->stream));
    meme_ctx::HostBuf* Hb = ctx->h_sam;
    if ((rc = meme_hostbuf_reserve(ctx, Hb[0], (size_t)(nrecs + 1) * 8)) || (rc = meme_hostbuf_reserve(ctx, Hb[1], (size_t)total + 64))) return rc;
    HIP_TRY(hipMemcpyAsync(Hb[0].p, d_toff, (size_t)(nrecs + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (total) HIP_TRY(hipMemcpyAsync(Hb[1].p, S[7].p, (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_sam[0], ctx->ev_sam[1]));
    out->nrecs = nrecs; out->text_off = (const int64_t*)Hb[0].p; out->text = (const char*)Hb[1].p; out->text_bytes = total; out->kernel_ms = ms;
    return MEME_OK;
}
