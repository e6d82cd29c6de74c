/* TEST / MEASUREMENT INFRASTRUCTURE ONLY.
 *
 * Instrumented CPU restatement of the reference's *probe sequence* (MODE 2: P-RMI + keys, no ISA):
 * where meme_oracle.c restates what a search returns, this file restates how the reference gets there
 * -- learned_index_lookup, the error-bounded binary search, the +-1 linear fix-up and the neighbour
 * counting loops -- so that the work counters of SURVEY.md section 8(d) (the reference's own
 * `Count_mem_ref` instrumentation, src/LearnedIndex_seeding.h:96) can be produced deterministically
 * for any data set:   lookups, partial lookups, SA-entry compares, extra 8-byte reference words,
 * hits, SMEMs.  bench.py turns them into "algorithmic bytes per read".  It is also usable as a
 * "port" CPU baseline.  Results (SMEMs) are asserted equal to the semantic oracle in tests.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "meme_oracle.h"

typedef struct { double icpt, slope; uint64_t err; } rp_rec;

typedef struct {
    const orc_index* idx;
    const rp_rec* l2;
    const rp_rec* l1;
    int bits;
} rp_model;

typedef struct {
    int64_t lookups, partial_lookups, compares, extra_words, hits, smems, searches;
} rp_counters;

typedef struct {
    const rp_model* M;
    const uint8_t* fw;
    uint8_t* rc;
    int l_seq, min_seed_len, min_intv, pivot, l_pivot;
    orc_mem_tl* smems;
    int smem_cap, n_smems;
    int64_t n_hits;
    int overflow;
    rp_counters* c;
} rp_state;

static size_t fclamp(double v, double bound) {
    if (v < 0.0) return 0;
    return v > bound ? (size_t)bound : (size_t)v;
}

/* learned_index_lookup (src/LearnedIndex_seeding.cpp:186-210) */
static uint64_t rp_lookup(const rp_model* M, uint64_t key, uint64_t* err, rp_counters* c) {
    size_t m = M->bits ? (size_t)(key >> (64 - M->bits)) : 0;
    double f = fma(M->l2[m].slope, (double)key, M->l2[m].icpt);
    *err = M->l2[m].err;
    c->lookups++;
    if (*err >> 63) {
        size_t ps = (size_t)((*err >> 32) & 0x7fffffff);
        double pn = (double)(*err & 0xffffffffu);
        m = ps + fclamp(f, pn - 1);
        f = fma(M->l1[m].slope, (double)key, M->l1[m].icpt);
        *err = M->l1[m].err;
        c->partial_lookups++;
    }
    return (uint64_t)fclamp(f, (double)M->idx->n - 1.0);
}

/* Tokenization, scalar variant (:795-901): 32 bases from `from`, T-padded */
static uint64_t rp_key(const uint8_t* buf, int from, int l_seq) {
    uint64_t key = 0;
    int r, len = l_seq - from;
    for (r = 0; r < len && r < 32; ++r) {
        if (buf[from + r] >= 4) break;
        key = (key << 2) | buf[from + r];
    }
    for (; r < 32; ++r) key = (key << 2) | 3;
    return key;
}

static int first_n(const uint8_t* buf, int from, int l_seq) {
    for (int i = from; i < l_seq; ++i)
        if (buf[i] >= 4) return i;
    return l_seq;
}

/* compare_read_and_ref_binary_LOADSUFFIX (:226-329) with the traffic accounting of SURVEY 8(d) */
static int rp_cmp(rp_state* s, const uint8_t* q, uint64_t slot, uint64_t valid_len, uint32_t* match_len, int* exact) {
    int r = orc_compare(s->M->idx, slot, q, (int64_t)valid_len, match_len, exact);
    uint64_t pos = s->M->idx->sa[slot];
    s->c->compares++;
    int64_t w = ((int64_t)*match_len + (int64_t)(pos & 3) + 31) / 32 - 1;
    if (w > 0) s->c->extra_words += w;
    return r;
}

/* window + lower-bound loop + linear fix-up (:2262-2357 == :2802-2894) */
static uint64_t rp_locate(rp_state* s, const uint8_t* q, uint64_t valid, uint64_t est, uint64_t enc_err,
                          uint32_t* match_len_out) {
    const uint64_t n = (uint64_t)s->M->idx->n;
    uint64_t curr_err = (enc_err >> 32) & 0x3fffffff, err = enc_err & 0x7fffffff;
    uint64_t iter = est;
    uint64_t upper_b = iter + err >= n - 1 ? n - 2 : iter + err;
    uint64_t lower_b = iter > curr_err ? iter - curr_err : 1;
    lower_b = lower_b >= upper_b ? upper_b - 1 : lower_b;
    uint64_t cnt = upper_b - lower_b + 1, middle = iter, half;
    uint32_t match_len = 0, last;
    int exact = 0;
    while ((half = cnt >> 1)) {
        middle = lower_b + half;
        lower_b = rp_cmp(s, q, middle, valid, &match_len, &exact) ? middle : lower_b;
        if (exact) break;
        cnt -= half;
    }
    if (exact) iter = lower_b;
    else if (middle != lower_b) {
        last = match_len;
        iter = lower_b;
        while (!rp_cmp(s, q, iter, valid, &match_len, &exact) && !exact) {
            if (iter == 0) break;
            iter--;
            last = match_len;
        }
        if (last > match_len) { iter++; match_len = last; }
    } else {
        last = match_len;
        iter = lower_b + 1;
        if (iter >= n) iter = n - 1;
        while (rp_cmp(s, q, iter, valid, &match_len, &exact) && !exact) {
            if (iter == n - 1) break;
            iter++;
            last = match_len;
        }
        if (last > match_len) { iter--; match_len = last; }
    }
    *match_len_out = match_len;
    return iter;
}

/* mem_search's counting loop (:2902-2942): probes alternate below / above */
static uint32_t rp_count_simple(rp_state* s, const uint8_t* q, uint64_t start, uint32_t match_len, int min_intv) {
    const uint64_t n = (uint64_t)s->M->idx->n;
    uint64_t up = 1, low = 1;
    uint32_t up_match = match_len, low_match = match_len;
    int ex;
    for (;;) {
        for (;;) {
            int lf = low_match >= match_len && low <= start;
            int uf = up_match >= match_len && n - 1 >= up + start;
            if (lf) { rp_cmp(s, q, start - low, match_len, &low_match, &ex); low++; }
            if (uf) { rp_cmp(s, q, start + up, match_len, &up_match, &ex); up++; }
            if (up + low - 3 >= (uint64_t)min_intv || (!lf && !uf)) break;
        }
        if (low_match == match_len && low > start) { low += 1; low_match = 0; }
        if (up_match == match_len && up + start > n - 1) { up += 1; up_match = 0; }
        if (up + low - 3 >= (uint64_t)min_intv) break;
        match_len = up_match > low_match ? up_match : low_match;
    }
    return match_len;
}

/* right_smem_search's counting (:2365-2574): <=15 linear probes, then exponential + binary */
static uint32_t rp_count_smem(rp_state* s, const uint8_t* q, uint64_t start, uint32_t match_len, int min_intv,
                              uint64_t* iter_out, uint64_t* num_out, int* early) {
    const uint64_t n = (uint64_t)s->M->idx->n;
    uint64_t up = 0, low = 0, upper_b, lower_b, cnt, half, middle, match_num, iter = start;
    uint32_t up_match = match_len, low_match = match_len;
    const int msl = s->min_seed_len;
    int ex;
    *early = 0;
    for (;;) {
        while (up + low < 15) {
            if ((int)match_len < msl && (up + low + (up == 0) + (low == 0) - 1) >= (uint64_t)min_intv) goto early_out;
            if (low_match >= match_len && low + 1 <= start) { low++; rp_cmp(s, q, start - low, match_len, &low_match, &ex); }
            else if (up_match >= match_len && n - 2 >= up + start) { up++; rp_cmp(s, q, start + up, match_len, &up_match, &ex); }
            else break;
            if ((int)match_len < msl && (up + low + (up == 0) - 1) >= (uint64_t)min_intv) goto early_out;
        }
        uint64_t mv = 5;
        if (low_match >= match_len && low < start) {
            upper_b = start - low;
            mv = upper_b < mv ? upper_b : mv;
            lower_b = upper_b - mv;
            cnt = upper_b - lower_b + 1;
            while (low_match >= match_len) {
                rp_cmp(s, q, lower_b, match_len, &low_match, &ex);
                if (low_match < match_len) break;
                if (lower_b == 0) break;
                mv <<= 1;
                mv = mv < lower_b ? mv : lower_b;
                lower_b -= mv;
                upper_b = lower_b + mv;
                cnt = upper_b - lower_b + 1;
            }
            while ((half = cnt >> 1)) {
                if ((int)match_len < msl && up + start - upper_b >= (uint64_t)min_intv) goto early_out;
                middle = upper_b - half;
                rp_cmp(s, q, middle, match_len, &low_match, &ex);
                if (low_match >= match_len) {
                    upper_b = middle;
                    if (upper_b != 0 && cnt < 3) rp_cmp(s, q, upper_b - 1, match_len, &low_match, &ex);
                }
                cnt -= half;
            }
            low = start + 1 - upper_b;
            if ((int)match_len < msl && up + low - 1 >= (uint64_t)min_intv) goto early_out;
        }
        if (up_match >= match_len && n - 1 > up + start) {
            mv = 5;
            lower_b = start + up;
            mv = n - 1 - lower_b < mv ? n - 1 - lower_b : mv;
            upper_b = lower_b + mv;
            cnt = upper_b - lower_b + 1;
            while (up_match >= match_len) {
                rp_cmp(s, q, upper_b, match_len, &up_match, &ex);
                if (up_match < match_len) break;
                if (upper_b == n - 1) break;
                mv <<= 1;
                mv = mv < n - 1 - upper_b ? mv : n - 1 - upper_b;
                upper_b += mv;
                lower_b = upper_b - mv;
                cnt = upper_b - lower_b + 1;
            }
            while ((half = cnt >> 1)) {
                if ((int)match_len < msl && lower_b - start + low >= (uint64_t)min_intv) goto early_out;
                middle = lower_b + half;
                rp_cmp(s, q, middle, match_len, &up_match, &ex);
                if (up_match >= match_len) {
                    lower_b = middle;
                    if (lower_b != n - 1 && cnt < 3) rp_cmp(s, q, lower_b + 1, match_len, &up_match, &ex);
                }
                cnt -= half;
            }
            up = lower_b + 1 - start;
            if ((int)match_len < msl && up + low - 1 >= (uint64_t)min_intv) goto early_out;
        }
        if (low_match >= match_len && low >= start) { low = start + 1; low_match = 0; }
        if (up_match >= match_len && up + start >= n - 1) { up = n - start; up_match = 0; }
        match_num = up + low - 1;
        iter = start - low + 1;
        if (match_num >= (uint64_t)min_intv) break;
        match_len = up_match > low_match ? up_match : low_match;
    }
    *iter_out = iter;
    *num_out = match_num;
    return match_len;
early_out:
    *early = 1;
    *iter_out = iter;
    *num_out = 0;
    return match_len;
}

static void rp_set_pivot(rp_state* s, int p) { s->pivot = p; s->l_pivot = s->l_seq - 1 - p; }

static void rp_emit(rp_state* s, int start, int end, uint64_t iter, uint64_t num) {
    if (s->n_smems >= s->smem_cap) { s->overflow = 1; return; }
    orc_mem_tl* m = &s->smems[s->n_smems++];
    m->start = start; m->end = end; m->hitbeg = (int32_t)s->n_hits; m->hitcount = (int32_t)num;
    m->cache_refpos = s->M->idx->sa[iter];
    s->n_hits += (int64_t)num;
    s->c->smems++;
    s->c->hits += (int64_t)num;
}

static uint32_t rp_right_smem(rp_state* s) {
    int amb = first_n(s->fw, s->pivot, s->l_seq);
    const uint8_t* q = s->fw + s->pivot;
    uint64_t err, iter, num;
    uint32_t ml;
    int early;
    s->c->searches++;
    uint64_t est = rp_lookup(s->M, rp_key(s->fw, s->pivot, s->l_seq), &err, s->c);
    uint64_t start = rp_locate(s, q, (uint64_t)(amb - s->pivot), est, err, &ml);
    ml = rp_count_smem(s, q, start, ml, s->min_intv, &iter, &num, &early);
    if (!early && (int)ml >= s->min_seed_len) rp_emit(s, s->pivot, s->pivot + (int)ml, iter, num);
    return ml;
}

static uint32_t rp_mem_only(rp_state* s, int right) {
    const uint8_t* buf = right ? s->fw : s->rc;
    int from = right ? s->pivot : s->l_pivot;
    int amb = first_n(buf, from, s->l_seq);
    uint64_t err;
    uint32_t ml;
    s->c->searches++;
    uint64_t est = rp_lookup(s->M, rp_key(buf, from, s->l_seq), &err, s->c);
    uint64_t start = rp_locate(s, buf + from, (uint64_t)(amb - from), est, err, &ml);
    if (s->min_intv != 1) ml = rp_count_simple(s, buf + from, start, ml, s->min_intv);
    return ml;
}

static void rp_zigzag(rp_state* s, int next_pivot, int check_n) {
    int sp = s->pivot, guard = 0;
    while (sp < next_pivot) {
        if (++guard > 4 * s->l_seq + 16) break;
        if (check_n && s->fw[sp] >= 4) {
            if (s->l_seq - sp < s->min_seed_len) { rp_set_pivot(s, s->l_seq); sp = s->l_seq; }
            else { sp += 1; rp_set_pivot(s, s->pivot + 1); }
            continue;
        }
        uint32_t ss = rp_mem_only(s, 0);
        rp_set_pivot(s, s->pivot - (int)ss + 1);
        if (next_pivot - s->pivot < s->min_seed_len) break;
        ss = rp_right_smem(s);
        sp = s->pivot + (int)ss;
        rp_set_pivot(s, sp);
    }
}

static void rp_step(rp_state* s, int one_pos) {
    int next;
    if (s->fw[s->pivot] >= 4) {
        if (s->l_seq - s->pivot < s->min_seed_len) rp_set_pivot(s, s->l_seq);
        else rp_set_pivot(s, s->pivot + 1);
        return;
    }
    if (s->pivot != 0 && s->fw[s->pivot - 1] < 4) {
        if (one_pos) { next = s->pivot + (int)rp_mem_only(s, 1); rp_zigzag(s, next, 0); }
        else { next = s->l_seq; rp_zigzag(s, next, 1); }
    } else next = s->pivot + (int)rp_right_smem(s);
    rp_set_pivot(s, next);
}

/* Learned_bwtSeedStrategyAllPosOneThread (:974-1283) */
static void rp_round3(rp_state* s) {
    const uint64_t n = (uint64_t)s->M->idx->n;
    const int min_intv = s->min_intv, msl = s->min_seed_len;
    rp_set_pivot(s, 0);
    while (s->pivot < s->l_seq - msl + 1) {
        if (s->fw[s->pivot] >= 4) { rp_set_pivot(s, s->pivot + 1); continue; }
        int amb = first_n(s->fw, s->pivot, s->l_seq), valid = amb - s->pivot;
        if (valid < msl) { rp_set_pivot(s, s->pivot + valid); continue; }
        const uint8_t* q = s->fw + s->pivot;
        uint64_t err;
        uint32_t match_len;
        int ex;
        s->c->searches++;
        uint64_t est = rp_lookup(s->M, rp_key(s->fw, s->pivot, s->l_seq), &err, s->c);
        uint64_t start = rp_locate(s, q, (uint64_t)valid, est, err, &match_len), iter = start;
        if ((int)match_len < msl) { rp_set_pivot(s, s->pivot + msl); continue; }
        uint64_t up = 1, low = 1, match_num = 1, last_num = 0, last_iter = iter;
        uint32_t up_match = match_len, low_match = match_len;
        for (;;) {
            for (;;) {
                int lf = low_match >= match_len && low <= start;
                int uf = up_match >= match_len && n - 1 >= up + start;
                if (lf) { rp_cmp(s, q, start - low, match_len, &low_match, &ex); low++; }
                if (uf) { rp_cmp(s, q, start + up, match_len, &up_match, &ex); up++; }
                if (up + low - 3 >= (uint64_t)min_intv || (!lf && !uf)) break;
            }
            if (low_match == match_len && low > start) { low += 1; low_match = 0; }
            if (up_match == match_len && up + start > n - 1) { up += 1; up_match = 0; }
            match_num = up + low - 3;
            if (match_num >= (uint64_t)min_intv) {
                match_num = last_num ? last_num : match_num;
                iter = last_iter;
                match_len = match_len + 1;
                break;
            }
            if ((int)(up_match > low_match ? up_match : low_match) < msl) {
                match_len = (uint32_t)msl;
                iter = start - low + 2;
                break;
            }
            last_num = match_num;
            match_len = up_match > low_match ? up_match : low_match;
            iter = start - low + 2;
            last_iter = iter;
        }
        if (match_num < (uint64_t)min_intv) {
            if ((int)match_len < msl) match_len = (uint32_t)msl;
            rp_emit(s, s->pivot, s->pivot + (int)match_len, iter, match_num);
        }
        rp_set_pivot(s, s->pivot + (int)match_len);
    }
}

int rp_seed_batch(const orc_index* idx, const void* l1, const void* l2, int64_t l2_records, const uint8_t* reads,
                  const int64_t* read_off, int64_t nreads, const orc_seed_params* p, orc_mem_tl* smems,
                  int32_t smem_cap, int32_t* n_smems, int64_t* counters7, int threads) {
    rp_model M;
    M.idx = idx; M.l1 = (const rp_rec*)l1; M.l2 = (const rp_rec*)l2;
    M.bits = 0;
    while (((int64_t)1 << M.bits) < l2_records) M.bits++;
    rp_counters tot;
    memset(&tot, 0, sizeof(tot));
    int bad = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
    {
        rp_counters c;
        memset(&c, 0, sizeof(c));
        orc_mem_tl* local = smems ? NULL : (orc_mem_tl*)malloc(sizeof(orc_mem_tl) * (size_t)smem_cap);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
        for (int64_t i = 0; i < nreads; ++i) {
            rp_state s;
            memset(&s, 0, sizeof(s));
            int len = (int)(read_off[i + 1] - read_off[i]);
            const uint8_t* rd = reads + read_off[i];
            uint8_t* rc = (uint8_t*)malloc((size_t)len + 1);
            for (int k = 0; k < len; ++k) rc[len - 1 - k] = rd[k] < 4 ? 3 - rd[k] : 4;
            s.M = &M; s.fw = rd; s.rc = rc; s.l_seq = len; s.c = &c;
            s.smems = smems ? smems + i * smem_cap : local;
            s.smem_cap = smem_cap;
            s.min_seed_len = p->min_seed_len;
            s.min_intv = 1;
            if (len > 0) {
                rp_set_pivot(&s, 0);
                int guard = 0;
                while (s.pivot < s.l_seq && ++guard < 4 * len + 16) {
                    int before = s.n_smems;
                    rp_step(&s, 0);
                    int after = s.n_smems;
                    if (p->steps < 2) continue;
                    for (int k = before; k < after; ++k) {
                        int next = s.pivot, saved = s.min_intv;
                        int qb = s.smems[k].start, qe = s.smems[k].end;
                        if (qe - qb < p->split_len || s.smems[k].hitcount > p->split_width) continue;
                        rp_set_pivot(&s, (qb + qe) >> 1);
                        s.min_intv = s.smems[k].hitcount + 1;
                        rp_step(&s, 1);
                        s.min_intv = saved;
                        rp_set_pivot(&s, next);
                    }
                }
                if (p->steps >= 3 && p->max_mem_intv > 0) {
                    s.min_intv = p->max_mem_intv;
                    s.min_seed_len = p->min_seed_len + 1;
                    rp_round3(&s);
                }
            }
            free(rc);
            if (n_smems) n_smems[i] = s.n_smems;
            if (s.overflow) bad = 1;
        }
        free(local);
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            tot.lookups += c.lookups; tot.partial_lookups += c.partial_lookups; tot.compares += c.compares;
            tot.extra_words += c.extra_words; tot.hits += c.hits; tot.smems += c.smems; tot.searches += c.searches;
        }
    }
    if (counters7) {
        counters7[0] = tot.lookups; counters7[1] = tot.partial_lookups; counters7[2] = tot.compares;
        counters7[3] = tot.extra_words; counters7[4] = tot.hits; counters7[5] = tot.smems; counters7[6] = tot.searches;
    }
    return bad ? -1 : 0;
}

/* rdtsc frequency, to turn the reference harness' "Consumed: N cycles" into seconds */
#if defined(__x86_64__)
#include <time.h>
#include <x86intrin.h>
double orc_tsc_hz(void) {
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    unsigned long long t0 = __rdtsc();
    do { clock_gettime(CLOCK_MONOTONIC, &b); } while ((b.tv_sec - a.tv_sec) * 1e9 + (b.tv_nsec - a.tv_nsec) < 2e8);
    unsigned long long t1 = __rdtsc();
    return (double)(t1 - t0) / (((b.tv_sec - a.tv_sec) * 1e9 + (b.tv_nsec - a.tv_nsec)) * 1e-9);
}
#else
double orc_tsc_hz(void) { return 0.0; }
#endif
