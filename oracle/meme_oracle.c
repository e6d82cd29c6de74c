/* TEST INFRASTRUCTURE ONLY -- see meme_oracle.h.  Plain C restatement of
 *   (1) BWA-MEME learned-index seeding  (reference src/LearnedIndex_seeding.cpp, src/bwamem.cpp:1230-1413)
 *   (2) banded Smith-Waterman extension (reference src/bandedSWA.cpp:116-260 == src/ksw.cpp:434-535)
 *
 * Seeding is restated at the level of its *semantic contract* (SURVEY.md Appendix B): every search of
 * the reference ends in the triple (match_len, range_start, count) that depends only on the suffix
 * array order and the text, never on the learned model or on the probe sequence.  The oracle therefore
 * locates by a plain binary search over the whole suffix array on the 1-byte/base text and derives the
 * triple from longest-common-prefix values; the per-read pivot state machines (rounds 1-3) are restated
 * statement by statement because their order of searches *is* output-visible.
 */
#include "meme_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * compare: reference compare_read_and_ref_binary* (src/LearnedIndex_seeding.cpp:226-601).
 *   ref_len = n - SA[slot];  L = min(valid_len, ref_len);  lcp = common prefix of q[0,L) and text.
 *   lcp <  L              -> match_len = lcp,     result = text[SA+lcp] < q[lcp]
 *   lcp == L <  ref_len   -> match_len = L, exact, result = true   (suffix continues past the query)
 *   lcp == L == ref_len   -> match_len = ref_len, result = false   (text ends: sorts as T-padded)
 * ------------------------------------------------------------------------------------------------ */
int orc_compare(const orc_index* idx, uint64_t slot, const uint8_t* q, int64_t valid_len,
                uint32_t* match_len, int* exact) {
    uint64_t pos = idx->sa[slot];
    int64_t ref_len = idx->n - (int64_t)pos;
    int64_t L = valid_len < ref_len ? valid_len : ref_len;
    const uint8_t* t = idx->text + pos;
    int64_t l = 0;
    while (l < L && t[l] == q[l]) ++l;
    *exact = 0;
    if (l < L) {
        *match_len = (uint32_t)l;
        return t[l] < q[l];
    }
    if (L < ref_len) {
        *match_len = (uint32_t)L;
        *exact = 1;
        return 1;
    }
    *match_len = (uint32_t)ref_len;
    return 0;
}

/* capped LCP of q with the suffix in SA slot j */
static uint32_t lcp_at(const orc_index* idx, int64_t j, const uint8_t* q, int64_t valid_len) {
    uint32_t m;
    int ex;
    orc_compare(idx, (uint64_t)j, q, valid_len, &m, &ex);
    return m;
}

static int ge_at(const orc_index* idx, int64_t j, const uint8_t* q, uint32_t L) {
    return lcp_at(idx, j, q, (int64_t)L) >= L;
}

/* lowest slot lo' <= lo such that every slot in [lo', lo] shares >= L bases with q (lo does) */
static int64_t extend_down(const orc_index* idx, const uint8_t* q, uint32_t L, int64_t lo) {
    int64_t step = 1, good = lo, bad = -1;
    while (good > 0) {
        int64_t probe = good - step;
        if (probe < 0) probe = 0;
        if (ge_at(idx, probe, q, L)) { good = probe; step <<= 1; }
        else { bad = probe; break; }
    }
    /* invariant: good matches, bad (if >=0) does not, bad < good */
    while (good - bad > 1 && bad >= 0) {
        int64_t mid = bad + (good - bad) / 2;
        if (ge_at(idx, mid, q, L)) good = mid; else bad = mid;
    }
    return good;
}

static int64_t extend_up(const orc_index* idx, const uint8_t* q, uint32_t L, int64_t hi) {
    int64_t step = 1, good = hi, bad = idx->n;
    while (good < idx->n - 1) {
        int64_t probe = good + step;
        if (probe > idx->n - 1) probe = idx->n - 1;
        if (ge_at(idx, probe, q, L)) { good = probe; step <<= 1; }
        else { bad = probe; break; }
    }
    while (bad - good > 1 && bad < idx->n) {
        int64_t mid = good + (bad - good) / 2;
        if (ge_at(idx, mid, q, L)) good = mid; else bad = mid;
    }
    return good;
}

/* locate (mem_search / right_smem_search up to the "iter_pos points to best exact matching position"
 * comment, src/LearnedIndex_seeding.cpp:2262-2358, 2802-2894): boundary between compare()==true and
 * ==false in SA order; best = larger LCP of the two boundary neighbours. */
static uint32_t locate(const orc_index* idx, const uint8_t* q, int64_t valid_len, int64_t* best) {
    int64_t lo = 0, hi = idx->n;
    uint32_t m;
    int ex;
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        if (orc_compare(idx, (uint64_t)mid, q, valid_len, &m, &ex)) lo = mid + 1; else hi = mid;
    }
    uint32_t m_lo = 0, m_hi = 0;
    int have_lo = lo > 0, have_hi = lo < idx->n;
    if (have_lo) m_lo = lcp_at(idx, lo - 1, q, valid_len);
    if (have_hi) m_hi = lcp_at(idx, lo, q, valid_len);
    if (have_lo && (!have_hi || m_lo >= m_hi)) { *best = lo - 1; return m_lo; }
    *best = lo;
    return m_hi;
}

/* Level search shared by right_smem_search (src/LearnedIndex_seeding.cpp:2359-2574), mem_search
 * (:2898-2943, 3155-3196) and their ISA-seeded twins (:3417-3461 ...):
 *   L = maxLCP; loop { [s,e] = maximal SA run around best with LCP >= L; if (e-s+1 >= min_intv) stop;
 *                      L = max(LCP(s-1), LCP(e+1)) (0 beyond the array ends) }                      */
uint32_t orc_search(const orc_index* idx, const uint8_t* q, int64_t valid_len, int32_t min_intv,
                    int64_t* start, int64_t* count, orc_counters* ctr) {
    int64_t best;
    uint32_t L = locate(idx, q, valid_len, &best);
    int64_t s = best, e = best;
    if (ctr) ctr->searches++;
    for (;;) {
        s = extend_down(idx, q, L, s);
        e = extend_up(idx, q, L, e);
        if (ctr) ctr->level_steps++;
        if (e - s + 1 >= (int64_t)min_intv) break;
        uint32_t lm = s > 0 ? lcp_at(idx, s - 1, q, (int64_t)L) : 0;
        uint32_t um = e < idx->n - 1 ? lcp_at(idx, e + 1, q, (int64_t)L) : 0;
        L = um > lm ? um : lm;
    }
    *start = s;
    *count = e - s + 1;
    return L;
}

/* ---------------------------------------------------------------------------------------------- */
typedef struct {
    const orc_index* idx;
    const uint8_t* fw;       /* read, codes 0..4           (unpacked_queue_buf)    */
    uint8_t* rc;             /* reverse complement, N -> 4 (unpacked_rc_queue_buf) */
    int l_seq;
    int min_seed_len, min_intv_limit;
    int pivot, l_pivot;
    orc_mem_tl* smems;
    int smem_cap, n_smems;
    uint64_t* hits;
    int64_t hit_cap, n_hits;
    int overflow;
    orc_counters* ctr;
    /* measurement only (orc_diag_census_*): the diagonals (text position of read base 0) of the unique loci this read has met so far */
    int n_diag_smem, n_diag_any, round_tag;
    int64_t diag_smem[8], diag_any[8];
} rstate;

/* ---- measurement: how many searches a "diagonal + plcp" shortcut could answer (profiles/r05_diag_census.md) -------------------------------
 * A read that has met a locus with ONE occurrence knows where on the text each of its bases should lie (the diagonal d: read base p at text
 * position d + p; a search to the left of p is a search to the right of the mirror position n - 1 - (d + p)).  Let L_d be the number of bases
 * the query shares with the text at that candidate u.  Every other suffix v shares lcp(u, v) <= plcp[u] bases with suffix u, so if
 * plcp[u] < L_d no other suffix reaches L_d bases of the query: the search's answer is (L_d, one occurrence, u) -- no model look-up, no window.
 * The census counts, per round, the searches for which that holds with a diagonal from (a) an earlier emitted SMEM of one occurrence, (b) any
 * earlier search of the read that ended on one occurrence, and checks the claimed answer against the search's real one. */
static const uint8_t* g_census_plcp = NULL;
static long long g_census[4][6];     /* [round 1, re-seeding, third round][searches, eligible (a), eligible (b), wrong claims, L_d < min_seed_len among eligible (b), spare] */
void orc_diag_census_enable(const uint8_t* plcp) { g_census_plcp = plcp; memset(g_census, 0, sizeof(g_census)); }
void orc_diag_census_get(long long* out) { memcpy(out, g_census, sizeof(g_census)); }
/* plcp[u] = min(255, LCP of the suffix at text position u with the nearer of its two suffix-array neighbours) -- the table k_build_plcp makes on the device */
void orc_build_plcp(const uint8_t* text, const uint64_t* sa, int64_t n, uint8_t* plcp) {
    uint8_t* adj = (uint8_t*)malloc((size_t)n + 1);          /* adj[i] = min(255, lcp(sa[i-1], sa[i])) */
    adj[0] = 0; adj[n] = 0;
#pragma omp parallel for schedule(static)
    for (int64_t i = 1; i < n; ++i) {
        const int64_t a = (int64_t)sa[i - 1], b = (int64_t)sa[i];
        int l = 0;
        while (l < 255 && a + l < n && b + l < n && text[a + l] == text[b + l]) ++l;
        adj[i] = (uint8_t)l;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) plcp[sa[i]] = adj[i] > adj[i + 1] ? adj[i] : adj[i + 1];
    free(adj);
}
static void census_search(rstate* r, int right, const uint8_t* q, int64_t valid, uint32_t L, int64_t s, int64_t c) {
    if (!g_census_plcp) return;
    const orc_index* idx = r->idx;
    const int64_t n = idx->n;
    const int p = right ? r->pivot : r->pivot;           /* read position the query starts at (right) / ends at (left) */
    int elig[2] = {0, 0}, wrong = 0, shortd = 0;
    for (int kind = 0; kind < 2; ++kind) {
        const int nd = kind ? r->n_diag_any : r->n_diag_smem;
        const int64_t* dg = kind ? r->diag_any : r->diag_smem;
        for (int k = 0; k < nd && !elig[kind]; ++k) {
            const int64_t u = right ? dg[k] + p : n - 1 - (dg[k] + p);
            if (u < 0 || u >= n) continue;
            int64_t Ld = 0;
            while (Ld < valid && u + Ld < n && idx->text[u + Ld] == q[Ld]) ++Ld;
            const int pl = g_census_plcp[u];
            if (pl != 255 && pl < Ld) {
                elig[kind] = 1;
                if ((uint32_t)Ld != L || c != 1 || (int64_t)idx->sa[s] != u) wrong = 1;
                if (kind == 1 && Ld < r->min_seed_len) shortd = 1;
            }
        }
    }
    long long* g = g_census[r->round_tag];
#pragma omp atomic
    g[0] += 1;
#pragma omp atomic
    g[1] += elig[0];
#pragma omp atomic
    g[2] += elig[1];
#pragma omp atomic
    g[3] += wrong;
#pragma omp atomic
    g[4] += shortd;
    if (c == 1 && r->n_diag_any < 8) {                   /* what this search teaches the read */
        const int64_t v = (int64_t)idx->sa[s];
        const int64_t d = right ? v - p : n - 1 - v - p;
        int seen = 0;
        for (int k = 0; k < r->n_diag_any; ++k) seen |= r->diag_any[k] == d;
        if (!seen) r->diag_any[r->n_diag_any++] = d;
    }
}
static void census_smem(rstate* r, int start, int64_t pos, int64_t c) {
    if (!g_census_plcp || c != 1 || r->n_diag_smem >= 8) return;
    const int64_t d = pos - start;
    for (int k = 0; k < r->n_diag_smem; ++k) if (r->diag_smem[k] == d) return;
    r->diag_smem[r->n_diag_smem++] = d;
}

static void set_pivot(rstate* r, int pivot) {   /* set_forward_pivot, :68-71 */
    r->pivot = pivot;
    r->l_pivot = r->l_seq - 1 - pivot;
}

/* first ambiguous base at/after `from` (Tokenization's *ambiguous_pos, :795-901) */
static int first_n(const uint8_t* buf, int from, int l_seq) {
    for (int i = from; i < l_seq; ++i)
        if (buf[i] >= 4) return i;
    return l_seq;
}

/* right_smem_search (:2131-2664): search to the right of pivot, emit an SMEM if long enough */
static uint32_t right_smem(rstate* r) {
    int amb = first_n(r->fw, r->pivot, r->l_seq);
    int64_t valid = amb - r->pivot, s, c;
    uint32_t L = orc_search(r->idx, r->fw + r->pivot, valid, r->min_intv_limit, &s, &c, r->ctr);
    census_search(r, 1, r->fw + r->pivot, valid, L, s, c);
    if ((int)L >= r->min_seed_len) {
        if (r->n_smems >= r->smem_cap || r->n_hits + c > r->hit_cap) { r->overflow = 1; return L; }
        census_smem(r, r->pivot, (int64_t)r->idx->sa[s], c);
        orc_mem_tl* m = &r->smems[r->n_smems++];
        m->start = r->pivot;
        m->end = r->pivot + (int)L;
        m->hitbeg = (int32_t)r->n_hits;
        m->hitcount = (int32_t)c;
        m->cache_refpos = r->idx->sa[s];
        for (int64_t i = 0; i < c; ++i) r->hits[r->n_hits++] = r->idx->sa[s + i];
        if (r->ctr) { r->ctr->smems++; r->ctr->hits += c; }
    }
    return L;
}

/* mem_search (:2667-3204): direction 1 = right of pivot on the read, 0 = left of pivot, i.e. to the
 * right of l_pivot on the reverse-complemented read */
static uint32_t mem_only(rstate* r, int right) {
    int64_t s, c;
    if (right) {
        int amb = first_n(r->fw, r->pivot, r->l_seq);
        uint32_t L = orc_search(r->idx, r->fw + r->pivot, amb - r->pivot, r->min_intv_limit, &s, &c, r->ctr);
        census_search(r, 1, r->fw + r->pivot, amb - r->pivot, L, s, c);
        return L;
    }
    int amb = first_n(r->rc, r->l_pivot, r->l_seq);
    uint32_t L = orc_search(r->idx, r->rc + r->l_pivot, amb - r->l_pivot, r->min_intv_limit, &s, &c, r->ctr);
    census_search(r, 0, r->rc + r->l_pivot, amb - r->l_pivot, L, s, c);
    return L;
}

/* the zig-zag shared by step1 and OnePos: from search_pivot extend left to the MEM start, then right
 * to the SMEM end; repeat until next_pivot is reached (:1724-1849, 1969-2084).  check_n: step1 skips
 * ambiguous bases inside the loop (:1725-1736), OnePos cannot meet one. */
static void zigzag(rstate* r, int next_pivot, int check_n) {
    int search_pivot = r->pivot;
    int guard = 0;
    while (search_pivot < next_pivot) {
        if (++guard > 4 * r->l_seq + 16) break;   /* the reference would spin; never seen */
        if (check_n && r->fw[search_pivot] >= 4) {
            if (r->l_seq - search_pivot < r->min_seed_len) {
                set_pivot(r, r->l_seq);
                search_pivot = r->l_seq;
            } else {
                search_pivot += 1;
                set_pivot(r, r->pivot + 1);
            }
            continue;
        }
        uint32_t ss = mem_only(r, 0);
        set_pivot(r, r->pivot - (int)ss + 1);
        if (next_pivot - r->pivot < r->min_seed_len) break;
        ss = right_smem(r);
        search_pivot = r->pivot + (int)ss;
        set_pivot(r, search_pivot);
    }
}

/* Learned_getSMEMsOnePosOneThread_step1 (:1691-1894) */
static void step1(rstate* r) {
    int next_pivot;
    if (r->fw[r->pivot] >= 4) {
        if (r->l_seq - r->pivot < r->min_seed_len) set_pivot(r, r->l_seq);
        else set_pivot(r, r->pivot + 1);
        return;
    }
    if (r->pivot != 0 && r->fw[r->pivot - 1] < 4) {
        next_pivot = r->l_seq;
        zigzag(r, next_pivot, 1);
    } else {
        uint32_t L = right_smem(r);
        next_pivot = r->pivot + (int)L;
    }
    set_pivot(r, next_pivot);
}

/* Learned_getSMEMsOnePosOneThread (:1897-2126) */
static void one_pos(rstate* r) {
    int next_pivot;
    if (r->fw[r->pivot] >= 4) {
        if (r->l_seq - r->pivot < r->min_seed_len) set_pivot(r, r->l_seq);
        else set_pivot(r, r->pivot + 1);
        return;
    }
    if (r->pivot != 0 && r->fw[r->pivot - 1] < 4) {
        uint32_t L = mem_only(r, 1);
        next_pivot = r->pivot + (int)L;
        zigzag(r, next_pivot, 0);
    } else {
        uint32_t L = right_smem(r);
        next_pivot = r->pivot + (int)L;
    }
    set_pivot(r, next_pivot);
}

/* Learned_getSMEMsAllPosOneThread (:913-972) / _step1only (:904-911) */
static void all_pos(rstate* r, int split_len, int split_width, int with_round2) {
    set_pivot(r, 0);
    int guard = 0;
    while (r->pivot < r->l_seq) {
        if (++guard > 4 * r->l_seq + 16) break;
        int before = r->n_smems;
        step1(r);
        int after = r->n_smems;
        if (!with_round2) continue;
        for (int k = before; k < after; ++k) {
            int next_pivot = r->pivot;
            int saved = r->min_intv_limit;
            int qbeg = r->smems[k].start, qend = r->smems[k].end;
            if (qend - qbeg < split_len || r->smems[k].hitcount > split_width) {
                set_pivot(r, next_pivot);
                continue;
            }
            set_pivot(r, (qbeg + qend) >> 1);
            r->min_intv_limit = r->smems[k].hitcount + 1;
            r->round_tag = 1;
            one_pos(r);
            r->round_tag = 0;
            r->min_intv_limit = saved;
            set_pivot(r, next_pivot);
        }
    }
}

/* Learned_bwtSeedStrategyAllPosOneThread (:974-1283) and its ISA twin (:1284-1466): from every pivot
 * the shortest match of >= min_seed_len bases that occurs fewer than min_intv times. */
static void seed_strategy(rstate* r) {
    const orc_index* idx = r->idx;
    const int min_intv = r->min_intv_limit, msl = r->min_seed_len;
    set_pivot(r, 0);
    while (r->pivot < r->l_seq - msl + 1) {
        if (r->fw[r->pivot] >= 4) { set_pivot(r, r->pivot + 1); continue; }
        int amb = first_n(r->fw, r->pivot, r->l_seq);
        int valid = amb - r->pivot;
        if (valid < msl) { set_pivot(r, r->pivot + valid); continue; }
        const uint8_t* q = r->fw + r->pivot;
        int64_t best;
        uint32_t L = locate(idx, q, valid, &best);
        if (r->ctr) r->ctr->searches++;
        if (g_census_plcp) {                 /* third round: the longest match and whether it is alone -- what a diagonal with plcp[u] < L_d also tells */
            int64_t s1 = extend_down(idx, q, L, best), e1 = extend_up(idx, q, L, best);
            r->round_tag = 2;
            census_search(r, 1, q, valid, L, s1, e1 - s1 + 1);
            r->round_tag = 0;
        }
        if ((int)L < msl) { set_pivot(r, r->pivot + msl); continue; }
        int64_t s = best, e = best, last_s = best, last_cnt = 0, cnt, emit_s;
        uint32_t match_len;
        for (;;) {
            s = extend_down(idx, q, L, s);
            e = extend_up(idx, q, L, e);
            if (r->ctr) r->ctr->level_steps++;
            cnt = e - s + 1;
            if (cnt >= min_intv) {            /* :1243-1251 */
                cnt = last_cnt ? last_cnt : cnt;
                emit_s = last_s;
                match_len = L + 1;
                break;
            }
            uint32_t lm = s > 0 ? lcp_at(idx, s - 1, q, (int64_t)L) : 0;
            uint32_t um = e < idx->n - 1 ? lcp_at(idx, e + 1, q, (int64_t)L) : 0;
            uint32_t nxt = um > lm ? um : lm;
            if ((int)nxt < msl) {             /* :1252-1258 */
                match_len = (uint32_t)msl;
                emit_s = s;
                break;
            }
            last_cnt = cnt;                   /* :1259-1262 */
            last_s = s;
            L = nxt;
        }
        if (cnt < min_intv) {                 /* :1265-1277 */
            if ((int)match_len < msl) match_len = (uint32_t)msl;
            if (r->n_smems >= r->smem_cap || r->n_hits + cnt > r->hit_cap) { r->overflow = 1; return; }
            orc_mem_tl* m = &r->smems[r->n_smems++];
            m->start = r->pivot;
            m->end = r->pivot + (int)match_len;
            m->hitbeg = (int32_t)r->n_hits;
            m->hitcount = (int32_t)cnt;
            m->cache_refpos = idx->sa[emit_s];
            for (int64_t i = 0; i < cnt; ++i) r->hits[r->n_hits++] = idx->sa[emit_s + i];
            if (r->ctr) { r->ctr->smems++; r->ctr->hits += cnt; }
        }
        set_pivot(r, r->pivot + (int)match_len);
    }
}

/* per-read driver: the seeding part of mem_kernel1_core_Learned (src/bwamem.cpp:1249-1394) */
int orc_seed_read(const orc_index* idx, const uint8_t* read, int32_t len, const orc_seed_params* p,
                  orc_mem_tl* smems, int32_t smem_cap, int32_t* n_smems,
                  uint64_t* hits, int64_t hit_cap, int64_t* n_hits, orc_counters* ctr) {
    rstate r;
    memset(&r, 0, sizeof(r));
    uint8_t* rc = (uint8_t*)malloc((size_t)len + 1);
    for (int i = 0; i < len; ++i) rc[len - 1 - i] = read[i] < 4 ? 3 - read[i] : 4;  /* :1277-1280 */
    r.idx = idx; r.fw = read; r.rc = rc; r.l_seq = len;
    r.min_seed_len = p->min_seed_len;
    r.min_intv_limit = 1;
    r.smems = smems; r.smem_cap = smem_cap; r.hits = hits; r.hit_cap = hit_cap; r.ctr = ctr;
    if (len > 0) {
        all_pos(&r, p->split_len, p->split_width, p->steps >= 2);
        if (p->steps >= 3 && p->max_mem_intv > 0 && !r.overflow) {   /* :1385-1394 */
            r.min_intv_limit = p->max_mem_intv;
            r.min_seed_len = p->min_seed_len + 1;
            seed_strategy(&r);
        }
    }
    free(rc);
    *n_smems = r.n_smems;
    *n_hits = r.n_hits;
    return r.overflow ? -1 : 0;
}

int orc_seed_batch(const orc_index* idx, const uint8_t* reads, const int64_t* read_off, int64_t nreads,
                   const orc_seed_params* p, orc_mem_tl* smems, int32_t smem_cap, int32_t* n_smems,
                   uint64_t* hits, int64_t hit_cap_per_read, int64_t* n_hits, orc_counters* ctr,
                   int threads) {
    int rc_all = 0;
    orc_counters total;
    memset(&total, 0, sizeof(total));
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
    {
        orc_counters mine;
        memset(&mine, 0, sizeof(mine));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
        for (int64_t i = 0; i < nreads; ++i) {
            int rc = orc_seed_read(idx, reads + read_off[i], (int32_t)(read_off[i + 1] - read_off[i]), p,
                                   smems + i * smem_cap, smem_cap, &n_smems[i],
                                   hits + i * hit_cap_per_read, hit_cap_per_read, &n_hits[i], &mine);
            if (rc) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
                rc_all = -1;
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            total.searches += mine.searches; total.level_steps += mine.level_steps;
            total.smems += mine.smems; total.hits += mine.hits;
        }
    }
    if (ctr) *ctr = total;
    return rc_all;
}

/* ================================================================================================
 * Banded Smith-Waterman extension: scalarBandedSWA (reference src/bandedSWA.cpp:116-237), which is
 * ksw_extend2 (src/ksw.cpp:434-535).  Rows = target (reference) bases, columns = query bases.
 * eh[j] keeps { H(i-1,j-1), E(i,j) } exactly like the reference, *including* the stale entries it
 * leaves outside the current band -- they are read again when the band re-grows, so they are part of
 * the function's observable behaviour.
 * ================================================================================================ */
int orc_bsw_extend(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int w, int h0,
                   const orc_bsw_params* p, int* _qle, int* _tle, int* _gtle, int* _gscore,
                   int* _max_off, int64_t* cells) {
    const int o_del = p->o_del, e_del = p->e_del, o_ins = p->o_ins, e_ins = p->e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int32_t* H = (int32_t*)calloc((size_t)qlen + 1, sizeof(int32_t));
    int32_t* E = (int32_t*)calloc((size_t)qlen + 1, sizeof(int32_t));
    /* scoring matrix of bwa_fill_scmat (src/bwa.cpp:262-270): a on the diagonal, -b off it, -1 with N */
    int i, j, beg, end, max, max_i, max_j, max_ie, gscore, max_off;
    int64_t ncell = 0;
    /* first row (:143-145) */
    H[0] = h0;
    H[1] = h0 > oe_ins ? h0 - oe_ins : 0;
    for (j = 2; j <= qlen && H[j - 1] > e_ins; ++j) H[j] = H[j - 1] - e_ins;
    /* band cap from the best possible score (:148-156); max of the matrix is a (a>0) */
    {
        int mx = p->a > 0 ? p->a : 0;
        int max_ins = (int)((double)(qlen * mx + p->end_bonus - o_ins) / e_ins + 1.);
        if (max_ins < 1) max_ins = 1;
        if (w > max_ins) w = max_ins;
        int max_del = (int)((double)(qlen * mx + p->end_bonus - o_del) / e_del + 1.);
        if (max_del < 1) max_del = 1;
        if (w > max_del) w = max_del;
    }
    max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
    beg = 0; end = qlen;
    for (i = 0; i < tlen; ++i) {
        int f = 0, h1, m = 0, mj = -1;
        const int tb = target[i];
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) {
            h1 = h0 - (o_del + e_del * (i + 1));
            if (h1 < 0) h1 = 0;
        } else h1 = 0;
        for (j = beg; j < end; ++j) {
            int M = H[j], e = E[j], h, t;
            int qb = query[j];
            int sc = (tb > 3 || qb > 3) ? -1 : (tb == qb ? p->a : -p->b);
            H[j] = h1;
            M = M ? M + sc : 0;
            h = M > e ? M : e;
            h = h > f ? h : f;
            h1 = h;
            mj = m > h ? mj : j;
            m = m > h ? m : h;
            t = M - oe_del; t = t > 0 ? t : 0;
            e -= e_del; e = e > t ? e : t;
            E[j] = e;
            t = M - oe_ins; t = t > 0 ? t : 0;
            f -= e_ins; f = f > t ? f : t;
            ++ncell;
        }
        H[end] = h1; E[end] = 0;
        if (j == qlen) {
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (m == 0) break;
        if (m > max) {
            max = m; max_i = i; max_j = mj;
            int off = mj - i; if (off < 0) off = -off;
            max_off = max_off > off ? max_off : off;
        } else if (p->zdrop > 0) {
            if (i - max_i > mj - max_j) {
                if (max - m - ((i - max_i) - (mj - max_j)) * e_del > p->zdrop) break;
            } else {
                if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > p->zdrop) break;
            }
        }
        for (j = beg; j < end && H[j] == 0 && E[j] == 0; ++j) {}
        beg = j;
        for (j = end; j >= beg && H[j] == 0 && E[j] == 0; --j) {}
        end = j + 2 < qlen ? j + 2 : qlen;
    }
    free(H); free(E);
    if (_qle) *_qle = max_j + 1;
    if (_tle) *_tle = max_i + 1;
    if (_gtle) *_gtle = max_ie + 1;
    if (_gscore) *_gscore = gscore;
    if (_max_off) *_max_off = max_off;
    if (cells) *cells += ncell;
    return max;
}

/* scalarBandedSWAWrapper (src/bandedSWA.cpp:242-260) -- also what getScores8/16 compute */
void orc_bsw_batch(orc_seqpair* pairs, const uint8_t* ref, const uint8_t* qer, int32_t n, int32_t w,
                   const orc_bsw_params* p, int threads, int64_t* cells) {
    int64_t total = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : total)
#endif
    for (int32_t i = 0; i < n; ++i) {
        orc_seqpair* s = &pairs[i];
        int64_t c = 0;
        s->score = orc_bsw_extend(s->len2, qer + s->idq, s->len1, ref + s->idr, w, s->h0, p, &s->qle,
                                  &s->tle, &s->gtle, &s->gscore, &s->max_off, &c);
        total += c;
    }
    if (cells) *cells = total;
}

/* ------------------------------------------------------------------------------------------------
 * (3) chaining: mem_chain_Learned (reference src/bwamem.cpp:1122-1204) + test_and_merge (:450-492)
 *     + mem_chain_weight (:522-541) + mem_chain_flt (:599-717), one read at a time.
 *
 * The reference keeps the chains of a read in a B-tree keyed by position (klib kbtree) and sorts them by
 * weight with klib's introsort (ksort.h) -- both third-party code vendored in the reference tree.  What is
 * output-visible of them is restated here: the chain with the largest position <= the seed's is the merge
 * candidate, traversal is by ascending position, and the sort performs klib's sequence of comparisons and
 * swaps (median-of-three quicksort with an explicit stack, partitions of <= 16 elements left to a final
 * insertion sort, comb sort when the depth budget runs out) because chains of EQUAL weight stay in the order
 * that algorithm leaves them in and the overlap filter depends on it.  A read that inserts a second chain at
 * a position that already has one makes the reference depend on the B-tree's order of equal keys: the oracle
 * reports such reads as undefined (-1) instead of guessing.
 * ------------------------------------------------------------------------------------------------ */
typedef struct { int64_t pos; int32_t rid, n, m, w, first, kept, is_alt; orc_cseed* seeds; } ochain;

static int o_pos2rid(const int64_t* off, int n_contigs, int64_t l_pac, int64_t pos_f) {   /* bns_pos2rid, src/bntseq.cpp:392-406 */
    int left = 0, mid = 0, right = n_contigs;
    if (pos_f >= l_pac) return -1;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= off[mid]) {
            if (mid == n_contigs - 1) break;
            if (pos_f < off[mid + 1]) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}
static int o_intv2rid(const int64_t* off, int n_contigs, int64_t l_pac, int64_t rb, int64_t re) {   /* bns_intv2rid, :408-416 */
    int rid_b, rid_e;
    if (rb < l_pac && re > l_pac) return -2;
    rid_b = o_pos2rid(off, n_contigs, l_pac, rb >= l_pac ? (l_pac << 1) - 1 - rb : rb);
    rid_e = rb < re ? o_pos2rid(off, n_contigs, l_pac, re - 1 >= l_pac ? (l_pac << 1) - 1 - (re - 1) : re - 1) : rid_b;
    return rid_b == rid_e ? rid_b : -1;
}
static int o_weight(const ochain* c) {
    int64_t end = 0;
    int j, w = 0, tmp;
    for (j = 0; j < c->n; ++j) {
        const orc_cseed* s = &c->seeds[j];
        if (s->qbeg >= end) w += s->len;
        else if (s->qbeg + s->len > end) w += (int)(s->qbeg + s->len - end);
        end = end > s->qbeg + s->len ? end : s->qbeg + s->len;
    }
    tmp = w; w = 0;
    for (j = 0, end = 0; j < c->n; ++j) {
        const orc_cseed* s = &c->seeds[j];
        if (s->rbeg >= end) w += s->len;
        else if (s->rbeg + s->len > end) w += (int)(s->rbeg + s->len - end);
        end = end > s->rbeg + s->len ? end : s->rbeg + s->len;
    }
    w = w < tmp ? w : tmp;
    return w < 1 << 30 ? w : (1 << 30) - 1;
}
#define O_LT(a, b) ((a).w > (b).w)            /* flt_lt, src/bwamem.cpp:80 */
#define O_SWAP(a, b) do { ochain t_ = (a); (a) = (b); (b) = t_; } while (0)
static void o_insertsort(ochain* s, ochain* t) {
    ochain *i, *j;
    for (i = s + 1; i < t; ++i) for (j = i; j > s && O_LT(*j, *(j - 1)); --j) O_SWAP(*j, *(j - 1));
}
static void o_combsort(size_t n, ochain* a) {
    const double shrink = 1.2473309501039786540366528676643;
    int do_swap;
    size_t gap = n;
    ochain *i, *j;
    do {
        if (gap > 2) { gap = (size_t)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
        do_swap = 0;
        for (i = a; i < a + n - gap; ++i) { j = i + gap; if (O_LT(*j, *i)) { O_SWAP(*i, *j); do_swap = 1; } }
    } while (do_swap || gap > 2);
    if (gap != 1) o_insertsort(a, a + n);
}
static void o_introsort(size_t n, ochain* a) {
    struct { ochain *left, *right; int depth; } stack[160], *top = stack;
    ochain rp, *s, *t, *i, *j, *k;
    int d;
    if (n < 1) return;
    if (n == 2) { if (O_LT(a[1], a[0])) O_SWAP(a[0], a[1]); return; }
    for (d = 2; 1ul << d < n; ++d);
    s = a; t = a + (n - 1); d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) { o_combsort((size_t)(t - s) + 1, s); t = s; continue; }
            i = s; j = t; k = i + ((j - i) >> 1) + 1;
            if (O_LT(*k, *i)) { if (O_LT(*k, *j)) k = j; }
            else k = O_LT(*j, *i) ? i : j;
            rp = *k;
            if (k != t) O_SWAP(*k, *t);
            for (;;) {
                do ++i; while (O_LT(*i, rp));
                do --j; while (i <= j && O_LT(rp, *j));
                if (j <= i) break;
                O_SWAP(*i, *j);
            }
            O_SWAP(*i, *t);
            if (i - s > t - i) {
                if (i - s > 16) { top->left = s; top->right = i - 1; top->depth = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { top->left = i + 1; top->right = t; top->depth = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == stack) { o_insertsort(a, a + n); return; }
            --top; s = top->left; t = top->right; d = top->depth;
        }
    }
}
static int smem_cmp(const void* a_, const void* b_) {      /* (start, end) ascending; ties cannot matter (same substring, same hits) */
    const orc_mem_tl* a = (const orc_mem_tl*)a_; const orc_mem_tl* b = (const orc_mem_tl*)b_;
    if (a->start != b->start) return a->start < b->start ? -1 : 1;
    if (a->end != b->end) return a->end < b->end ? -1 : 1;
    return a->hitbeg < b->hitbeg ? -1 : (a->hitbeg > b->hitbeg);
}

/* ---- the reference's position-keyed B-tree (klib kbtree.h, instantiated src/bwamem.cpp:43-44 with key = mem_chain_t, compare = position only;
 * kb_init(chn, KB_DEFAULT_SIZE + 8): t = ((520 - 4 - 8) / (8 + sizeof(mem_chain_t) = 48) + 1) >> 1 = 5, at most 9 keys per node).
 * Restated on chain ids so that chains with EQUAL positions come out of the traversal -- and are found by the interval query --
 * exactly where the reference's tree puts them (equal keys go after the first equal key of the leaf the descent reaches;
 * splits move the median up).  Nodes live in one growing array; child / key slots hold indices. */
#define OBT_T 5
#define OBT_MAXK (2 * OBT_T - 1)
typedef struct { int n, internal; int key[OBT_MAXK]; int ptr[OBT_MAXK + 1]; } obt_node;
typedef struct { obt_node* nd; int n_nodes, cap, root, n_keys; const ochain* ch; } obt_tree;
static int obt_new(obt_tree* T) {
    if (T->n_nodes == T->cap) { T->cap = T->cap ? T->cap * 2 : 16; T->nd = (obt_node*)realloc(T->nd, sizeof(obt_node) * (size_t)T->cap); }
    memset(&T->nd[T->n_nodes], 0, sizeof(obt_node));
    return T->n_nodes++;
}
/* __kb_getp_aux: first key >= pos; if that key is greater than pos, the one before it; *r = sign of (pos - key found) as the macro leaves it */
static int obt_getp_aux(const obt_tree* T, const obt_node* x, int64_t pos, int* r) {
    int begin = 0, end = x->n, tr;
    if (!r) r = &tr;
    if (x->n == 0) return -1;
    while (begin < end) {
        const int mid = (begin + end) >> 1;
        if (T->ch[x->key[mid]].pos < pos) begin = mid + 1; else end = mid;
    }
    if (begin == x->n) { *r = 1; return x->n - 1; }
    *r = (T->ch[x->key[begin]].pos < pos) - (pos < T->ch[x->key[begin]].pos);
    if (*r < 0) --begin;
    return begin;
}
/* kb_intervalp, lower bound only: the chain id test_and_merge is tried on, or -1 */
static int obt_lower(const obt_tree* T, int64_t pos) {
    int x = T->root, lower = -1;
    while (x >= 0) {
        const obt_node* nd = &T->nd[x];
        int r = 0;
        const int i = obt_getp_aux(T, nd, pos, &r);
        if (i >= 0 && r == 0) return nd->key[i];
        if (i >= 0) lower = nd->key[i];
        if (!nd->internal) return lower;
        x = nd->ptr[i + 1];
    }
    return lower;
}
static void obt_split(obt_tree* T, int xi, int i, int yi) {                  /* __kb_split: child yi of xi (at slot i) is full */
    const int zi = obt_new(T);
    obt_node *x = &T->nd[xi], *y = &T->nd[yi], *z = &T->nd[zi];
    z->internal = y->internal;
    z->n = OBT_T - 1;
    memcpy(z->key, y->key + OBT_T, sizeof(int) * (OBT_T - 1));
    if (y->internal) memcpy(z->ptr, y->ptr + OBT_T, sizeof(int) * OBT_T);
    y->n = OBT_T - 1;
    memmove(x->ptr + i + 2, x->ptr + i + 1, sizeof(int) * (size_t)(x->n - i));
    x->ptr[i + 1] = zi;
    memmove(x->key + i + 1, x->key + i, sizeof(int) * (size_t)(x->n - i));
    x->key[i] = y->key[OBT_T - 1];
    ++x->n;
}
static void obt_put(obt_tree* T, int id) {                                    /* kb_putp */
    const int64_t pos = T->ch[id].pos;
    int x;
    ++T->n_keys;
    if (T->nd[T->root].n == OBT_MAXK) {
        const int s = obt_new(T), r = T->root;
        T->nd[s].internal = 1; T->nd[s].n = 0; T->nd[s].ptr[0] = r;
        T->root = s;
        obt_split(T, s, 0, r);
    }
    x = T->root;
    for (;;) {                                                                /* __kb_putp_aux, iteratively */
        obt_node* nd = &T->nd[x];
        int i = obt_getp_aux(T, nd, pos, NULL);
        if (!nd->internal) {
            if (i != nd->n - 1) memmove(nd->key + i + 2, nd->key + i + 1, sizeof(int) * (size_t)(nd->n - i - 1));
            nd->key[i + 1] = id;
            ++nd->n;
            return;
        }
        ++i;
        if (T->nd[nd->ptr[i]].n == OBT_MAXK) {
            obt_split(T, x, i, nd->ptr[i]);
            nd = &T->nd[x];                                                   /* (the node array may have moved) */
            if (pos > T->ch[nd->key[i]].pos) ++i;
        }
        x = nd->ptr[i];
    }
}
static void obt_traverse(const obt_tree* T, int x, int* out, int* n) {       /* __kb_traverse: in order */
    const obt_node* nd = &T->nd[x];
    int i;
    for (i = 0; i <= nd->n; ++i) {
        if (nd->internal) obt_traverse(T, nd->ptr[i], out, n);
        if (i < nd->n) out[(*n)++] = nd->key[i];
    }
}

int orc_chain_read(const orc_mem_tl* smems_in, int n_smems, const uint64_t* hits, int len, const int64_t* contig_off,
                   const uint8_t* contig_alt, int n_contigs, const orc_chain_opt* o, orc_chain* out, int chain_cap,
                   orc_cseed* seeds_out, int seed_cap, int* tree_size, float* frac_rep) {
    ochain* ch = NULL;
    orc_mem_tl* sm = NULL;
    obt_tree T;
    int nc = 0, cap = 0, i, k, n, rc = 0, b = 0, e = 0, l_rep = 0;
    *tree_size = 0; *frac_rep = 0.f;
    if (len < o->min_seed_len) return 0;
    sm = (orc_mem_tl*)malloc(sizeof(orc_mem_tl) * (size_t)(n_smems > 0 ? n_smems : 1));
    memcpy(sm, smems_in, sizeof(orc_mem_tl) * (size_t)n_smems);
    qsort(sm, (size_t)n_smems, sizeof(orc_mem_tl), smem_cmp);
    for (i = 0; i < n_smems; ++i) {                           /* frac_rep, :1140-1147 */
        if (sm[i].hitcount <= o->max_occ) continue;
        if (sm[i].start > e) { l_rep += e - b; b = sm[i].start; e = sm[i].end; }
        else e = e > sm[i].end ? e : sm[i].end;
    }
    l_rep += e - b;
    memset(&T, 0, sizeof(T));
    T.root = obt_new(&T);                                     /* kb_init: an empty leaf */
    for (i = 0; i < n_smems && rc == 0; ++i) {                /* :1149-1193 */
        const orc_mem_tl* p = &sm[i];
        const int step = p->hitcount > o->max_occ ? p->hitcount / o->max_occ : 1;
        int64_t kk; int count;
        for (kk = 0, count = 0; kk < p->hitcount && count < o->max_occ; kk += step, ++count) {
            orc_cseed s; int rid, lower = -1, merged = 0;
            s.rbeg = (int64_t)hits[p->hitbeg + kk]; s.qbeg = p->start; s.len = p->end - p->start;
            rid = o_intv2rid(contig_off, n_contigs, o->l_pac, s.rbeg, s.rbeg + s.len);
            if (rid < 0) continue;
            T.ch = ch;
            if (T.n_keys) lower = obt_lower(&T, s.rbeg);      /* kb_intervalp (:1167-1169) */
            if (lower >= 0) {                                 /* test_and_merge */
                ochain* c = &ch[lower];
                const orc_cseed* last = &c->seeds[c->n - 1];
                const int64_t qend = last->qbeg + last->len, rend = last->rbeg + last->len;
                if (rid == c->rid) {
                    if (s.qbeg >= c->seeds[0].qbeg && s.qbeg + s.len <= qend && s.rbeg >= c->seeds[0].rbeg && s.rbeg + s.len <= rend) merged = 1;
                    else if ((last->rbeg < o->l_pac || c->seeds[0].rbeg < o->l_pac) && s.rbeg >= o->l_pac) merged = 0;
                    else {
                        const int64_t x = s.qbeg - last->qbeg, y = s.rbeg - last->rbeg;
                        if (y >= 0 && x - y <= o->w && y - x <= o->w && x - last->len < o->max_chain_gap && y - last->len < o->max_chain_gap) {
                            if (c->n == c->m) { c->m <<= 1; c->seeds = (orc_cseed*)realloc(c->seeds, sizeof(orc_cseed) * (size_t)c->m); }
                            c->seeds[c->n++] = s;
                            merged = 1;
                        }
                    }
                }
            }
            if (!merged) {
                ochain c;
                if (nc == cap) { cap = cap ? cap * 2 : 8; ch = (ochain*)realloc(ch, sizeof(ochain) * (size_t)cap); }
                c.pos = s.rbeg; c.rid = rid; c.n = 1; c.m = 4; c.w = 0; c.first = -1; c.kept = 0; c.is_alt = contig_alt[rid] ? 1 : 0;
                c.seeds = (orc_cseed*)malloc(sizeof(orc_cseed) * 4);
                c.seeds[0] = s;
                ch[nc] = c;                                   /* chains are stored in creation order; the tree orders their ids */
                T.ch = ch;
                obt_put(&T, nc);                              /* kb_putp (:1190) */
                ++nc;
            }
        }
    }
    *tree_size = nc;
    *frac_rep = (float)l_rep / len;
    if (nc > 0) {                                             /* __kb_traverse into chain->a (:1194-1198) */
        int* ord = (int*)malloc(sizeof(int) * (size_t)nc);
        ochain* sorted = (ochain*)malloc(sizeof(ochain) * (size_t)nc);
        int m = 0;
        T.ch = ch;
        obt_traverse(&T, T.root, ord, &m);
        for (i = 0; i < nc; ++i) sorted[i] = ch[ord[i]];
        free(ch); free(ord);
        ch = sorted;
    }
    free(T.nd);
    n = 0;
    if (rc == 0 && nc > 0) {                                  /* mem_chain_flt */
        int* kept_idx = (int*)malloc(sizeof(int) * (size_t)nc);
        int nk = 0, ns_out = 0;
        /* (:604-619: chains below min_chain_weight are dropped by compaction.  When ALL of them are below it, nothing is copied, the count
         * becomes 0 -- and the code that follows (:620-643) still builds one range [0, 1) from the array's stale first element: the
         * reference returns the FIRST chain in tree order, kept = 3.  Restated as it behaves.) */
        int below0 = 0;                                       /* chain 0 is below the floor and still sits in slot 0 */
        for (i = 0; i < nc; ++i) {
            ch[i].first = -1; ch[i].kept = 0; ch[i].w = o_weight(&ch[i]);
            if (ch[i].w < o->min_chain_weight) { if (i == 0) below0 = 1; else free(ch[i].seeds); }
            else { if (below0) { free(ch[0].seeds); below0 = 0; } ch[n++] = ch[i]; }
        }
        if (n == 0) n = 1;                                    /* slot 0 untouched: the stale first element */
        nc = n;                                               /* (dropped chains are gone) */
        if (n > 0) {
            o_introsort((size_t)n, ch);
            ch[0].kept = 3; kept_idx[nk++] = 0;
            for (i = 1; i < n; ++i) {
                int large_ovlp = 0;
                const int beg_i = ch[i].seeds[0].qbeg, end_i = ch[i].seeds[ch[i].n - 1].qbeg + ch[i].seeds[ch[i].n - 1].len;
                for (k = 0; k < nk; ++k) {
                    const int j = kept_idx[k];
                    const int beg_j = ch[j].seeds[0].qbeg, end_j = ch[j].seeds[ch[j].n - 1].qbeg + ch[j].seeds[ch[j].n - 1].len;
                    const int b_max = beg_j > beg_i ? beg_j : beg_i, e_min = end_j < end_i ? end_j : end_i;
                    if (e_min > b_max && (!ch[j].is_alt || ch[i].is_alt)) {
                        const int li = end_i - beg_i, lj = end_j - beg_j, min_l = li < lj ? li : lj;
                        if (e_min - b_max >= min_l * o->mask_level && min_l < o->max_chain_gap) {
                            large_ovlp = 1;
                            if (ch[j].first < 0) ch[j].first = i;
                            if (ch[i].w < ch[j].w * o->drop_ratio && ch[j].w - ch[i].w >= o->min_seed_len << 1) break;
                        }
                    }
                }
                if (k == nk) { kept_idx[nk++] = i; ch[i].kept = large_ovlp ? 2 : 3; }
            }
            for (i = 0; i < nk; ++i) if (ch[kept_idx[i]].first >= 0) ch[ch[kept_idx[i]].first].kept = 1;
            for (i = k = 0; i < n; ++i) {
                if (ch[i].kept == 0 || ch[i].kept == 3) continue;
                if (++k >= o->max_chain_extend) break;
            }
            for (; i < n; ++i) if (ch[i].kept < 3) ch[i].kept = 0;
            for (i = k = 0; i < n; ++i) {
                int j;
                if (ch[i].kept == 0) continue;
                if (k >= chain_cap || ns_out + ch[i].n > seed_cap) { rc = -2; break; }
                out[k].pos = ch[i].pos; out[k].rid = ch[i].rid; out[k].n_seeds = ch[i].n; out[k].w = ch[i].w; out[k].first = ch[i].first;
                out[k].kept = ch[i].kept; out[k].is_alt = ch[i].is_alt; out[k].seed_beg = ns_out;
                for (j = 0; j < ch[i].n; ++j) seeds_out[ns_out++] = ch[i].seeds[j];
                ++k;
            }
            n = k;
        }
        free(kept_idx);
    }
    for (i = 0; i < nc; ++i) free(ch[i].seeds);
    free(ch);
    free(sm);
    return rc ? rc : n;
}

/* Batched checker: the chains of every read of a batch (flat arrays as meme_chain_last_batch_host returns them; dev_* = what the
 * device computed) against orc_chain_read, on `threads` OpenMP threads.  Returns the number of reads that differ in any field of any
 * chain or seed, in tree size or in frac_rep; *first_bad = the first such read (-1 if none). */
typedef struct { int64_t pos; int32_t rid, n_seeds, w, first; int16_t kept, is_alt; int32_t seed_beg, pad; } orc_dev_chain;   /* = meme_chain */
int64_t orc_chain_compare_batch(const orc_mem_tl* smems, const int64_t* smem_off, const uint64_t* hits, const int64_t* hit_off, const int32_t* read_len,
                                int64_t nreads, const int64_t* contig_off, const uint8_t* contig_alt, int n_contigs, const orc_chain_opt* o,
                                const int64_t* dev_chain_off, const orc_dev_chain* dev_chains, const int64_t* dev_seed_off, const orc_cseed* dev_seeds,
                                const int32_t* dev_tree, const float* dev_frac, int threads, int64_t* first_bad) {
    int64_t n_bad = 0, first = -1, r;
    if (threads < 1) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads) reduction(+ : n_bad)
    for (r = 0; r < nreads; ++r) {
        const int ns = (int)(smem_off[r + 1] - smem_off[r]);
        int64_t work = 0;
        int i, cap, rc, tree = 0, bad = 0;
        float frac = 0.f;
        orc_chain* ch; orc_cseed* sd;
        for (i = 0; i < ns; ++i) work += smems[smem_off[r] + i].hitcount < o->max_occ ? smems[smem_off[r] + i].hitcount : o->max_occ;
        cap = (int)(work + 8);
        ch = (orc_chain*)malloc(sizeof(orc_chain) * (size_t)cap);
        sd = (orc_cseed*)malloc(sizeof(orc_cseed) * (size_t)cap);
        rc = orc_chain_read(smems + smem_off[r], ns, hits + hit_off[r], read_len[r], contig_off, contig_alt, n_contigs, o, ch, cap, sd, cap, &tree, &frac);
        if (rc < 0 || rc != dev_chain_off[r + 1] - dev_chain_off[r] || tree != dev_tree[r]) bad = 1;
        else {
            const orc_dev_chain* d = dev_chains + dev_chain_off[r];
            const orc_cseed* ds = dev_seeds + dev_seed_off[r];
            int k, j, nsd = 0;
            if (rc > 0 && memcmp(&frac, &dev_frac[r], 4) != 0) bad = 1;
            for (k = 0; k < rc && !bad; ++k) {
                if (d[k].pos != ch[k].pos || d[k].rid != ch[k].rid || d[k].n_seeds != ch[k].n_seeds || d[k].w != ch[k].w || d[k].first != ch[k].first ||
                    d[k].kept != ch[k].kept || d[k].is_alt != ch[k].is_alt || d[k].seed_beg != ch[k].seed_beg) bad = 1;
                for (j = 0; j < ch[k].n_seeds && !bad; ++j, ++nsd)
                    if (ds[nsd].rbeg != sd[nsd].rbeg || ds[nsd].qbeg != sd[nsd].qbeg || ds[nsd].len != sd[nsd].len) bad = 1;
            }
            if (!bad && nsd != dev_seed_off[r + 1] - dev_seed_off[r]) bad = 1;
        }
        free(ch); free(sd);
        if (bad) {
            n_bad += 1;
#pragma omp critical
            { if (first < 0 || r < first) first = r; }
        }
    }
    *first_bad = first;
    return n_bad;
}

/* ---- seed extension: mem_chain2aln_across_reads_V2 (reference src/bwamem.cpp:2573-3497), one read at a time ----------------------
 * The reference works through a 512-read batch stage by stage (all left extensions of the batch in its 8-bit / 16-bit / scalar
 * SIMD classes, then the band retries, then the right extensions ...); per read that is: one record per chained seed in extension
 * order, left extension (band w, once more with 2w when the result touches the band edge and changed the score), right extension
 * likewise from the left result's score, then the purge of records an earlier record of the read explains.  The classes compute the
 * same function (pinned: tests/test_oracle_golden.py), so their batching does not show in the result. */
#define O_H0 (-99)                                            /* H0_, src/macro.h:44 */
static int o_max_gap(const orc_ext_opt* o, int qlen) {        /* cal_max_gap, :85-95 */
    const int l_del = (int)((double)(qlen * o->a - o->o_del) / o->e_del + 1.);
    const int l_ins = (int)((double)(qlen * o->a - o->o_ins) / o->e_ins + 1.);
    int l = l_del > l_ins ? l_del : l_ins;
    l = l > 1 ? l : 1;
    return l < o->w << 1 ? l : o->w << 1;
}
static void o_seedcov(orc_alnreg* a, const orc_cseed* sd, int n) {   /* :2907-2917 */
    int i, cov = 0;
    if (a->rb == O_H0 || a->qb == O_H0 || a->qe == O_H0 || a->re == O_H0) return;
    for (i = 0; i < n; ++i)
        if (sd[i].qbeg >= a->qb && sd[i].qbeg + sd[i].len <= a->qe && sd[i].rbeg >= a->rb && sd[i].rbeg + sd[i].len <= a->re) cov += sd[i].len;
    a->seedcov = cov;
}
static int o_u64_cmp(const void* a, const void* b) { const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }

int orc_extend_read(const uint8_t* read, int l_query, const orc_chain* chains, int n_chains, const orc_cseed* seeds, float frac_rep,
                    const uint8_t* text, int64_t l_pac, const int64_t* contig_off, const int32_t* contig_len, const orc_ext_opt* o,
                    orc_alnreg* out, int64_t* n_jobs, int64_t* n_retried) {
    return orc_extend_read_scored(read, l_query, chains, n_chains, seeds, NULL, frac_rep, text, l_pac, contig_off, contig_len, o, out, n_jobs, n_retried);
}

int orc_extend_read_scored(const uint8_t* read, int l_query, const orc_chain* chains, int n_chains, const orc_cseed* seeds, const int32_t* seed_score,
                           float frac_rep, const uint8_t* text, int64_t l_pac, const int64_t* contig_off, const int32_t* contig_len, const orc_ext_opt* o,
                           orc_alnreg* out, int64_t* n_jobs, int64_t* n_retried) {
    int c, i, k, n_reg = 0, total = 0, max_n = 1, kept = 0, cur = 0;
    orc_bsw_params bl, br;
    uint64_t* srt;
    int* ord;                                                 /* per chain: seed indices in ascending (score, index) order; -1 = purged */
    uint8_t *qs, *rs;
    for (c = 0; c < n_chains; ++c) { total += chains[c].n_seeds; if (chains[c].n_seeds > max_n) max_n = chains[c].n_seeds; }
    memset(&bl, 0, sizeof(bl));
    bl.o_del = o->o_del; bl.e_del = o->e_del; bl.o_ins = o->o_ins; bl.e_ins = o->e_ins; bl.zdrop = o->zdrop; bl.a = o->a; bl.b = o->b;
    br = bl;
    bl.end_bonus = o->pen_clip5; br.end_bonus = o->pen_clip3;  /* bswLeft / bswRight, :2953-2959 */
    srt = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)max_n);
    ord = (int*)malloc(sizeof(int) * (size_t)(total + 1));
    qs = (uint8_t*)malloc((size_t)l_query + 8);
    rs = NULL;
    for (c = 0; c < n_chains; ++c) {
        const orc_chain* ch = &chains[c];
        const orc_cseed* sd = seeds + ch->seed_beg;
        int64_t rmax0 = l_pac << 1, rmax1 = 0, far_beg, far_end;
        int* co = ord + ch->seed_beg;
        if (ch->n_seeds == 0) continue;
        for (i = 0; i < ch->n_seeds; ++i) {                   /* the widest span any seed may reach (:2648-2666) */
            const int64_t b = sd[i].rbeg - (sd[i].qbeg + o_max_gap(o, sd[i].qbeg));
            const int tail = l_query - sd[i].qbeg - sd[i].len;
            const int64_t e = sd[i].rbeg + sd[i].len + (tail + o_max_gap(o, tail));
            if (b < rmax0) rmax0 = b;
            if (e > rmax1) rmax1 = e;
        }
        if (rmax0 < 0) rmax0 = 0;
        if (rmax1 > l_pac << 1) rmax1 = l_pac << 1;
        if (rmax0 < l_pac && l_pac < rmax1) { if (sd[0].rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac; }
        far_beg = contig_off[ch->rid]; far_end = far_beg + contig_len[ch->rid];      /* bns_fetch_seq_v2, src/bntseq.cpp:492-501 */
        if (sd[0].rbeg >= l_pac) { const int64_t t = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t; }
        if (rmax0 < far_beg) rmax0 = far_beg;
        if (rmax1 > far_end) rmax1 = far_end;
        rs = (uint8_t*)realloc(rs, (size_t)(rmax1 - rmax0) + 8);
        for (i = 0; i < ch->n_seeds; ++i)                     /* score = length unless mem_flt_chained_seeds has set it; keys unique */
            srt[i] = (uint64_t)(seed_score ? seed_score[ch->seed_beg + i] : sd[i].len) << 32 | (uint32_t)i;
        qsort(srt, (size_t)ch->n_seeds, 8, o_u64_cmp);
        for (i = 0; i < ch->n_seeds; ++i) co[i] = (int)(uint32_t)srt[i];
        for (k = ch->n_seeds - 1; k >= 0; --k) {              /* best seed first (:2702-2850) */
            const orc_cseed* s = &sd[co[k]];
            orc_alnreg* a = &out[n_reg++];
            int side;
            memset(a, 0, sizeof(*a));
            a->w = o->w; a->score = a->truesc = -1; a->rid = ch->rid; a->frac_rep = frac_rep; a->seedlen0 = s->len;
            a->rb = a->re = O_H0; a->qb = a->qe = O_H0;
            if (s->qbeg) { a->qb = s->qbeg; a->rb = s->rbeg; }
            else { a->score = a->truesc = s->len * o->a; a->qb = 0; a->rb = s->rbeg; }
            if (s->qbeg + s->len != l_query) { a->qe = s->qbeg + s->len; a->re = s->rbeg + s->len; }
            else { a->qe = l_query; a->re = s->rbeg + s->len; o_seedcov(a, sd, ch->n_seeds); }
            for (side = 0; side < 2; ++side) {                /* left, then right from the left result's score (:3371-3376) */
                int len1, len2, h0, attempt, j;
                if (side == 0 && !s->qbeg) continue;
                if (side == 1 && s->qbeg + s->len == l_query) continue;
                if (side == 0) {                              /* both sequences run away from the seed */
                    len2 = s->qbeg; len1 = (int)(s->rbeg - rmax0); h0 = s->len * o->a;
                    for (j = 0; j < len2; ++j) qs[j] = read[s->qbeg - 1 - j];
                    for (j = 0; j < len1; ++j) rs[j] = text[s->rbeg - 1 - j];
                } else {
                    const int qe = s->qbeg + s->len;
                    len2 = l_query - qe; len1 = (int)(rmax1 - (s->rbeg + s->len)); h0 = a->score;
                    memcpy(qs, read + qe, (size_t)len2);
                    memcpy(rs, text + s->rbeg + s->len, (size_t)len1);
                }
                for (attempt = 0; attempt < 2; ++attempt) {   /* MAX_BAND_TRY (:62); fold: :2985-3018 and siblings */
                    const int w = o->w << attempt, prev = a->score;
                    int qle, tle, gtle, gscore, max_off;
                    if (n_jobs) ++*n_jobs;
                    if (attempt && n_retried) ++*n_retried;
                    a->score = orc_bsw_extend(len2, qs, len1, rs, w, h0, side ? &br : &bl, &qle, &tle, &gtle, &gscore, &max_off, NULL);
                    if (a->score == prev || max_off < (w >> 1) + (w >> 2) || attempt == 1) {
                        if (side == 0) {
                            if (gscore <= 0 || gscore <= a->score - o->pen_clip5) { a->qb -= qle; a->rb -= tle; a->truesc = a->score; }
                            else { a->qb = 0; a->rb -= gtle; a->truesc = gscore; }
                        } else {
                            if (gscore <= 0 || gscore <= a->score - o->pen_clip3) { a->qe += qle; a->re += tle; a->truesc += a->score - h0; }
                            else { a->qe = l_query; a->re += gtle; a->truesc += gscore - h0; }
                        }
                        a->w = a->w > w ? a->w : w;
                        o_seedcov(a, sd, ch->n_seeds);
                        break;
                    }
                }
            }
        }
    }
    /* purge (:3389-3485), in the same order */
    for (c = 0; c < n_chains; ++c) {
        const orc_chain* ch = &chains[c];
        const orc_cseed* sd = seeds + ch->seed_beg;
        int* co = ord + ch->seed_beg;
        for (k = ch->n_seeds - 1; k >= 0; --k, ++cur) {
            const orc_cseed* s = &sd[co[k]];
            int v = 0;
            for (i = 0; i < n_reg && v < kept; ++i) {
                const orc_alnreg* p = &out[i];
                int qd, max_gap, band;
                int64_t rd;
                if (p->qb == -1 && p->qe == -1) continue;
                if (s->rbeg < p->rb || s->rbeg + s->len > p->re || s->qbeg < p->qb || s->qbeg + s->len > p->qe) { ++v; continue; }
                if (s->len - p->seedlen0 > .1 * l_query) { ++v; continue; }
                qd = s->qbeg - p->qb; rd = s->rbeg - p->rb;
                max_gap = o_max_gap(o, qd < rd ? qd : (int)rd);
                band = max_gap < p->w ? max_gap : p->w;
                if (qd - rd < band && rd - qd < band) break;
                qd = p->qe - (s->qbeg + s->len); rd = p->re - (s->rbeg + s->len);
                max_gap = o_max_gap(o, qd < rd ? qd : (int)rd);
                band = max_gap < p->w ? max_gap : p->w;
                if (qd - rd < band && rd - qd < band) break;
                ++v;
            }
            if (v < kept) {
                int u;
                for (u = k + 1; u < ch->n_seeds; ++u) {
                    const orc_cseed* t;
                    if (co[u] < 0) continue;
                    t = &sd[co[u]];
                    if (t->len < s->len * .95) continue;
                    if (s->qbeg <= t->qbeg && s->qbeg + s->len - t->qbeg >= s->len >> 2 && t->qbeg - s->qbeg != t->rbeg - s->rbeg) break;
                    if (t->qbeg <= s->qbeg && t->qbeg + t->len - s->qbeg >= s->len >> 2 && s->qbeg - t->qbeg != s->rbeg - t->rbeg) break;
                }
                if (u == ch->n_seeds) { out[cur].qb = out[cur].qe = -1; co[k] = -1; continue; }
            }
            ++kept;
        }
    }
    free(srt); free(ord); free(qs); free(rs);
    return n_reg;
}

/* batch form: reads flat, chains / seeds flat with per-read offsets (seed_beg relative to the read's first seed); out[seed_off[r] + i] */
int orc_extend_batch(const uint8_t* reads, const int64_t* read_off, int64_t nreads, const int64_t* chain_off, const orc_chain* chains,
                     const int64_t* seed_off, const orc_cseed* seeds, const float* frac_rep, const uint8_t* text, int64_t l_pac,
                     const int64_t* contig_off, const int32_t* contig_len, const orc_ext_opt* o, orc_alnreg* out, int threads, int64_t* stats) {
    return orc_extend_batch_scored(reads, read_off, nreads, chain_off, chains, seed_off, seeds, NULL, frac_rep, text, l_pac, contig_off, contig_len, o, out, threads, stats);
}

int orc_extend_batch_scored(const uint8_t* reads, const int64_t* read_off, int64_t nreads, const int64_t* chain_off, const orc_chain* chains,
                            const int64_t* seed_off, const orc_cseed* seeds, const int32_t* seed_score, const float* frac_rep, const uint8_t* text, int64_t l_pac,
                            const int64_t* contig_off, const int32_t* contig_len, const orc_ext_opt* o, orc_alnreg* out, int threads, int64_t* stats) {
    int64_t r, jobs = 0, retried = 0;
    int bad = 0;
    if (threads < 1) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads) reduction(+ : jobs, retried) reduction(| : bad)
    for (r = 0; r < nreads; ++r) {
        int64_t j = 0, t = 0;
        const int n = orc_extend_read_scored(reads + read_off[r], (int)(read_off[r + 1] - read_off[r]), chains + chain_off[r], (int)(chain_off[r + 1] - chain_off[r]),
                                      seeds + seed_off[r], seed_score ? seed_score + seed_off[r] : NULL, frac_rep[r], text, l_pac, contig_off, contig_len, o, out + seed_off[r], &j, &t);
        if (n != seed_off[r + 1] - seed_off[r]) bad = 1;
        jobs += j; retried += t;
    }
    if (stats) { stats[0] = jobs; stats[1] = retried; }
    return bad ? -1 : 0;
}

/* ---- mem_flt_chained_seeds (reference src/bwamem.cpp:565-598) with mem_seed_sw (:494-520) ----------------------------------------------
 * The alignment is ksw_align2 without KSW_XBYTE = ksw_i16 (src/ksw.cpp:236-320), of which only the score is used.  Restated with the
 * eight 16-bit lanes as arrays, lane by lane and statement by statement (the lazy-F loop raises H but not the E the main loop has already
 * stored, :280-283 vs :289-299 -- which never shows in a score: the two gaps in the other order reach the same cell). */
static int o_ksw_i16_score(int qlen, const uint8_t* query, int tlen, const uint8_t* target, const orc_ext_opt* o) {
    const int slen = (qlen + 7) / 8;                         /* ksw_qinit, :68-69 */
    const int oe_del = o->o_del + o->e_del, oe_ins = o->o_ins + o->e_ins;
    int i, j, k, l, gmax = 0;
    int *H0 = (int*)calloc((size_t)slen * 8 * 3, sizeof(int)), *H1 = H0 + slen * 8, *E = H1 + slen * 8;    /* [segment][lane] */
#define O_SUBS(x, y) ((x) > (y) ? (x) - (y) : 0)            /* _mm_subs_epu16 */
    for (i = 0; i < tlen; ++i) {
        int f[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mx = 0, h[8], live;
        int* t;
        for (l = 7; l > 0; --l) h[l] = H0[(slen - 1) * 8 + l - 1];        /* _mm_slli_si128(H0[slen - 1], 2), :271-272 */
        h[0] = 0;
        for (j = 0; j < slen; ++j) {
            for (l = 0; l < 8; ++l) {
                const int col = j + l * slen;                                  /* the profile's segmentation, :87-88, :104-107 */
                int sc = 0, e = E[j * 8 + l], hh;
                if (col < qlen) { const int qc = query[col], tc = target[i]; sc = (qc > 3 || tc > 3) ? -1 : (qc == tc ? o->a : -o->b); }
                hh = h[l] + sc;
                if (hh < e) hh = e;
                if (hh < f[l]) hh = f[l];
                if (hh > mx) mx = hh;
                H1[j * 8 + l] = hh;
                e = O_SUBS(e, o->e_del);
                { const int t2 = O_SUBS(hh, oe_del); if (e < t2) e = t2; }
                E[j * 8 + l] = e;
                f[l] = O_SUBS(f[l], o->e_ins);
                { const int t2 = O_SUBS(hh, oe_ins); if (f[l] < t2) f[l] = t2; }
                h[l] = H0[j * 8 + l];
            }
        }
        for (k = 0, live = 1; k < 16 && live; ++k) {                          /* lazy F, :289-299 */
            for (l = 7; l > 0; --l) f[l] = f[l - 1];
            f[0] = 0;
            for (j = 0; j < slen && live; ++j) {
                live = 0;
                for (l = 0; l < 8; ++l) {
                    int hh = H1[j * 8 + l];
                    if (hh < f[l]) hh = f[l];
                    H1[j * 8 + l] = hh;
                    hh = O_SUBS(hh, oe_ins);
                    f[l] = O_SUBS(f[l], o->e_ins);
                    if (f[l] > hh) live = 1;
                }
            }
        }
        if (mx > gmax) gmax = mx;
        t = H0; H0 = H1; H1 = t;
    }
#undef O_SUBS
    free(H0 < H1 ? H0 : H1);
    return gmax;
}

/* The bases mem_seed_sw's bns_fetch_seq returns in the aligner: mem_kernel1_core_Learned is handed worker_t::rc_pac as `pac`
 * (src/bwamem.cpp:1770), the 2-bit fwd + rc text with the four bases of every BYTE in reverse order for the learned index
 * (src/fastmap.cpp:440-457; the table of src/LearnedIndex_seeding.h:129-137 reverses 2-bit groups), and bns_get_seq
 * (src/bntseq.cpp:515-539) reads it with the plain _get_pac -- within every aligned four bases the order is reversed; reverse-strand
 * windows come from the forward half, complemented.  text = the true fwd + rc codes. */
static int o_flt_window_base(const uint8_t* text, int64_t l_pac, int64_t p) {
    const int64_t k = p < l_pac ? p : (l_pac << 1) - 1 - p;
    const int64_t kk = (k & ~(int64_t)3) + 3 - (k & 3);
    const int c = kk < l_pac << 1 ? text[kk] & 3 : 0;
    return p < l_pac ? c : 3 - c;
}

int orc_seed_sw(const uint8_t* read, int l_query, const orc_cseed* s, const uint8_t* text, int64_t l_pac, const int64_t* contig_off,
                const int32_t* contig_len, int n_contigs, const orc_ext_opt* o) {
    int qb, qe, lo, hi, i, sc;
    uint8_t win[200];
    int64_t rb, re, mid, fpos, far_beg, far_end;
    if (s->len >= 200) return -1;                            /* MEM_SHORT_LEN, :250, :502 */
    qb = s->qbeg; qe = s->qbeg + s->len;
    rb = s->rbeg; re = s->rbeg + s->len;
    mid = (rb + re) >> 1;
    qb -= 50; qb = qb > 0 ? qb : 0;                          /* MEM_SHORT_EXT, :249 */
    qe += 50; qe = qe < l_query ? qe : l_query;
    rb -= 50; rb = rb > 0 ? rb : 0;
    re += 50; re = re < l_pac << 1 ? re : l_pac << 1;
    if (rb < l_pac && l_pac < re) { if (mid < l_pac) re = l_pac; else rb = l_pac; }
    if (qe - qb >= 200 || re - rb >= 200) return -1;
    /* bns_fetch_seq (src/bntseq.cpp:541-570): inside the reference sequence of the midpoint, on its strand */
    fpos = mid >= l_pac ? (l_pac << 1) - 1 - mid : mid;
    lo = 0; hi = n_contigs - 1;
    while (lo < hi) { const int m = (lo + hi + 1) >> 1; if (contig_off[m] <= fpos) lo = m; else hi = m - 1; }
    far_beg = contig_off[lo]; far_end = far_beg + contig_len[lo];
    if (mid >= l_pac) { const int64_t t = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t; }
    if (rb < far_beg) rb = far_beg;
    if (re > far_end) re = far_end;
    for (i = 0; i < (int)(re - rb); ++i) win[i] = (uint8_t)o_flt_window_base(text, l_pac, rb + i);
    sc = o_ksw_i16_score(qe - qb, read + qb, (int)(re - rb), win, o);
    return sc;
}

/* One read: seeds that fail the test leave their chains (the read's seeds are packed to the front, chain after chain; seed_beg / n_seeds
 * of its chains follow); score[] receives what the reference leaves in mem_seed_t::score.  Returns the number of seeds that stay;
 * *n_sw counts the alignments run. */
int orc_flt_chained_seeds(const uint8_t* read, int l_query, orc_chain* chains, int n_chains, orc_cseed* seeds, int32_t* score, const uint8_t* text,
                          int64_t l_pac, const int64_t* contig_off, const int32_t* contig_len, int n_contigs, const orc_ext_opt* o, int min_chain_weight,
                          int64_t* n_sw) {
    const double min_l = min_chain_weight ? 1.1f * min_chain_weight : 5.5f * log(l_query);      /* MEM_HSP_COEF, MEM_MINSC_COEF, :252-253, :579-580 */
    const int min_hsp = (int)(o->a * min_l + .499);
    const int off = min_l > 0.05f * l_query;                                                     /* MEM_SEEDSW_COEF, :254, :583 */
    int c, j, kept = 0;
    for (c = 0; c < n_chains; ++c) {
        orc_chain* ch = &chains[c];
        const int beg = ch->seed_beg;
        int k = 0;
        for (j = 0; j < ch->n_seeds; ++j) {
            const orc_cseed s = seeds[beg + j];
            int sc = s.len;
            if (!off) {
                sc = orc_seed_sw(read, l_query, &s, text, l_pac, contig_off, contig_len, n_contigs, o);
                if (n_sw && !(s.len >= 200) && sc >= 0) ++*n_sw;
                if (!(sc < 0 || sc >= min_hsp)) continue;
                sc = sc < 0 ? s.len * o->a : sc;
            }
            seeds[kept + k] = s;
            score[kept + k] = sc;
            ++k;
        }
        ch->seed_beg = kept;
        ch->n_seeds = k;
        kept += k;
    }
    return kept;
}

int orc_flt_batch(const uint8_t* reads, const int64_t* read_off, int64_t nreads, const int64_t* chain_off, orc_chain* chains, const int64_t* seed_off,
                  orc_cseed* seeds, int32_t* score, const uint8_t* text, int64_t l_pac, const int64_t* contig_off, const int32_t* contig_len, int n_contigs,
                  const orc_ext_opt* o, int min_chain_weight, int64_t* kept, int threads, int64_t* n_sw_) {
    int64_t r, n_sw = 0;
    if (threads < 1) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads) reduction(+ : n_sw)
    for (r = 0; r < nreads; ++r) {
        int64_t k = 0;
        kept[r] = orc_flt_chained_seeds(reads + read_off[r], (int)(read_off[r + 1] - read_off[r]), chains + chain_off[r], (int)(chain_off[r + 1] - chain_off[r]),
                                        seeds + seed_off[r], score + seed_off[r], text, l_pac, contig_off, contig_len, n_contigs, o, min_chain_weight, &k);
        n_sw += k;
    }
    if (n_sw_) *n_sw_ = n_sw;
    return 0;
}

/* ---- banded global alignment with traceback: ksw_global2 (reference src/ksw.cpp:560-670) --------------------------------------------
 * Restated cell by cell: M separated from H, ties M >= E >= F, direction byte f<<4 | e<<2 | h per cell, backtrack from the last cell of
 * the last row's band.  mat = bwa_fill_scmat(a, b): match a, mismatch -b, anything with an ambiguous base -1.  cigar[] (capacity
 * qlen + tlen + 2) receives the operations in CIGAR order, BAM encoding.  Returns the score. */
#define O_MINUS_INF (-0x40000000)
int orc_ksw_global2(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int a, int b, int o_del, int e_del, int o_ins, int e_ins, int w,
                    int* n_cigar_, uint32_t* cigar) {
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    int i, j, score, n = 0;
    int* eh_h = (int*)calloc((size_t)qlen + 2, sizeof(int));
    int* eh_e = (int*)calloc((size_t)qlen + 2, sizeof(int));
    uint8_t* z = (uint8_t*)malloc((size_t)n_col * (size_t)tlen + 1);
    eh_h[0] = 0; eh_e[0] = O_MINUS_INF;
    for (j = 1; j <= qlen && j <= w; ++j) { eh_h[j] = -(o_ins + e_ins * j); eh_e[j] = O_MINUS_INF; }
    for (; j <= qlen; ++j) eh_h[j] = eh_e[j] = O_MINUS_INF;
    for (i = 0; i < tlen; ++i) {
        int f = O_MINUS_INF, h1, beg, end, t;
        uint8_t* zi = &z[(size_t)i * n_col];
        beg = i > w ? i - w : 0;
        end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : O_MINUS_INF;
        for (j = beg; j < end; ++j) {
            int h, m = eh_h[j], e = eh_e[j];
            uint8_t d;
            eh_h[j] = h1;
            m += (target[i] > 3 || query[j] > 3) ? -1 : (target[i] == query[j] ? a : -b);
            d = m >= e ? 0 : 1;
            h = m >= e ? m : e;
            d = h >= f ? d : 2;
            h = h >= f ? h : f;
            h1 = h;
            t = m - oe_del;
            e -= e_del;
            d |= e > t ? 1 << 2 : 0;
            e = e > t ? e : t;
            eh_e[j] = e;
            t = m - oe_ins;
            f -= e_ins;
            d |= f > t ? 2 << 4 : 0;
            f = f > t ? f : t;
            zi[j - beg] = d;
        }
        eh_h[end] = h1; eh_e[end] = O_MINUS_INF;
    }
    score = eh_h[qlen];
    if (n_cigar_ && cigar) {
        int k, which = 0, last = -1;
        uint32_t tmp;
        i = tlen - 1; k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
#define O_PUSH(op_, len_) do { if (last == (op_)) cigar[n - 1] += (uint32_t)(len_) << 4; else { cigar[n++] = (uint32_t)(len_) << 4 | (uint32_t)(op_); last = (op_); } } while (0)
        while (i >= 0 && k >= 0) {
            which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
            if (which == 0) { O_PUSH(0, 1); --i; --k; }
            else if (which == 1) { O_PUSH(2, 1); --i; }
            else { O_PUSH(1, 1); --k; }
        }
        if (i >= 0) O_PUSH(2, i + 1);
        if (k >= 0) O_PUSH(1, k + 1);
#undef O_PUSH
        for (i = 0; i < n >> 1; ++i) { tmp = cigar[i]; cigar[i] = cigar[n - 1 - i]; cigar[n - 1 - i] = tmp; }
        *n_cigar_ = n;
    }
    free(eh_h); free(eh_e); free(z);
    return score;
}

/* ---- bwa_gen_cigar2 whole (reference src/bwa.cpp:274-362): CIGAR + NM + MD of one call ---------------------------------------------------
 * text = fwd+rc codes (what bns_get_seq unpacks for [rb, re)), query = l_query codes 0..4 (the `&query[qb]` of mem_reg2aln).  Both sequences
 * reversed when rb >= l_pac (:288-293); equal lengths with w_ == 0 take the gap-free shortcut (:295-304), everything else the band of
 * :306-316 and ksw_global2; NM / MD as :322-355 (a deletion at either end of the CIGAR is not reported; "TGCAN" on the reverse strand).
 * cigar: capacity l_query + (re - rb) + 2; md: capacity 2 * (l_query + re - rb) + 16.  Returns 0, -1 for a call the function rejects. */
int orc_gen_cigar2(const uint8_t* text, int64_t l_pac, int a, int b, int o_del, int e_del, int o_ins, int e_ins, int w_, int l_query, const uint8_t* query,
                   int64_t rb, int64_t re, int* score, int* n_cigar, uint32_t* cigar, int* NM, char* md) {
    *n_cigar = 0; *NM = -1; *score = 0;
    if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac) || rb < 0 || re > 2 * l_pac) return -1;
    const int rlen = (int)(re - rb), rev = rb >= l_pac;
    uint8_t* q = (uint8_t*)malloc((size_t)l_query);
    uint8_t* t = (uint8_t*)malloc((size_t)rlen);
    int i, k;
    for (i = 0; i < l_query; ++i) q[i] = rev ? query[l_query - 1 - i] : query[i];
    for (i = 0; i < rlen; ++i) t[i] = rev ? text[re - 1 - i] : text[rb + i];
    if (l_query == rlen && w_ == 0) {
        cigar[0] = (uint32_t)l_query << 4;
        *n_cigar = 1;
        for (i = 0; i < l_query; ++i) *score += (t[i] > 3 || q[i] > 3) ? -1 : (t[i] == q[i] ? a : -b);
    } else {
        int max_ins = (int)((double)(((l_query + 1) >> 1) * a - o_ins) / e_ins + 1.);
        int max_del = (int)((double)(((l_query + 1) >> 1) * a - o_del) / e_del + 1.);
        int max_gap = max_ins > max_del ? max_ins : max_del;
        max_gap = max_gap > 1 ? max_gap : 1;
        int w = (max_gap + abs(rlen - l_query) + 1) >> 1;
        w = w < w_ ? w : w_;
        int min_w = abs(rlen - l_query) + 3;
        w = w > min_w ? w : min_w;
        *score = orc_ksw_global2(l_query, q, rlen, t, a, b, o_del, e_del, o_ins, e_ins, w, n_cigar, cigar);
    }
    {
        const char* int2base = rev ? "TGCAN" : "ACGTN";
        int x = 0, y = 0, u = 0, n_mm = 0, n_gap = 0, l = 0;
        for (k = 0; k < *n_cigar; ++k) {
            const int op = (int)(cigar[k] & 0xf), len = (int)(cigar[k] >> 4);
            if (op == 0) {
                for (i = 0; i < len; ++i) {
                    if (q[x + i] != t[y + i]) { l += sprintf(md + l, "%d%c", u, int2base[t[y + i]]); ++n_mm; u = 0; }
                    else ++u;
                }
                x += len; y += len;
            } else if (op == 2) {
                if (k > 0 && k < *n_cigar - 1) {
                    l += sprintf(md + l, "%d^", u);
                    for (i = 0; i < len; ++i) md[l++] = int2base[t[y + i]];
                    u = 0; n_gap += len;
                }
                y += len;
            } else if (op == 1) { x += len; n_gap += len; }
        }
        l += sprintf(md + l, "%d", u);
        md[l] = 0;
        *NM = n_mm + n_gap;
    }
    free(q); free(t);
    return 0;
}

/* ---- SAM text of one record: mem_aln2sam (reference src/bwamem.cpp:2174-2312) for the only record of a read (no SA / pa / XR tag, no comment) ------
 * rec = the fields of the record and of its mate (meme_sam_rec of include/meme_hip.h, restated here as orc_sam_rec), cigars / MD / XA in `blob`;
 * name, seq (codes), qual (or NULL) of the read; contig names.  out: capacity as the caller likes (a record is < l_name + 2 l_seq + 16 n_cigar + |MD| +
 * |XA| + 256 bytes).  Returns the text length (no terminator written). */
static int o_putw(char* o, long long v) {            /* kputw / kputl */
    char buf[24];
    int l = 0, n = 0;
    unsigned long long x = v < 0 ? (unsigned long long)(-v) : (unsigned long long)v;
    do { buf[l++] = (char)('0' + x % 10); x /= 10; } while (x);
    if (v < 0) o[n++] = '-';
    while (l) o[n++] = buf[--l];
    return n;
}
static int o_add_cigar(char* o, const uint32_t* cg, int n_cigar, int softclip, int is_alt, int which) {     /* add_cigar, :2161-2172 */
    int n = 0;
    if (!n_cigar) { o[n++] = '*'; return n; }
    for (int i = 0; i < n_cigar; ++i) {
        int c = (int)(cg[i] & 0xf);
        if (!softclip && !is_alt && (c == 3 || c == 4)) c = which ? 4 : 3;
        n += o_putw(o + n, cg[i] >> 4);
        o[n++] = "MIDSH"[c];
    }
    return n;
}
static long long o_rlen(int n_cigar, const uint32_t* cg) {          /* get_rlen, :2402-2410 */
    long long l = 0;
    for (int k = 0; k < n_cigar; ++k) { const int op = (int)(cg[k] & 0xf); if (op == 0 || op == 2) l += cg[k] >> 4; }
    return l;
}
int64_t orc_aln2sam(const orc_sam_rec* r, const uint8_t* blob, const char* name, int l_name, const uint8_t* seq, int l_seq, const char* qual,
                    const char* contig_names, const int32_t* contig_name_off, int softclip, const char* rg_id, char* o) {
    int64_t n = 0;
    int flag = r->flag, rid = r->rid, is_rev = r->is_rev, n_cigar = r->n_cigar;
    int64_t pos = r->pos;
    const int has_m = r->has_mate;
    int m_rid = r->m_rid, m_is_rev = r->m_is_rev, m_n_cigar = r->m_n_cigar;
    int64_t m_pos = r->m_pos;
    uint32_t cgb[1], mcb[1];
    const uint32_t* cg = r->n_cigar > 0 ? (const uint32_t*)(blob + r->cigar_off) : cgb;       /* (blob offsets are 4-byte aligned by the caller) */
    const uint32_t* mcg = r->m_n_cigar > 0 ? (const uint32_t*)(blob + r->m_cigar_off) : mcb;
    const char* md = r->n_cigar > 0 ? (const char*)(blob + r->cigar_off + 4 * (int64_t)r->n_cigar) : "";
    flag |= has_m ? 0x1 : 0;
    flag |= rid < 0 ? 0x4 : 0;
    flag |= has_m && m_rid < 0 ? 0x8 : 0;
    if (rid < 0 && has_m && m_rid >= 0) { rid = m_rid; pos = m_pos; is_rev = m_is_rev; n_cigar = 0; }
    if (has_m && m_rid < 0 && rid >= 0) { m_rid = rid; m_pos = pos; m_is_rev = is_rev; m_n_cigar = 0; }
    flag |= is_rev ? 0x10 : 0;
    flag |= has_m && m_is_rev ? 0x20 : 0;
    memcpy(o + n, name, (size_t)l_name); n += l_name; o[n++] = '\t';
    n += o_putw(o + n, (flag & 0xffff) | (flag & 0x10000 ? 0x100 : 0)); o[n++] = '\t';
    if (rid >= 0) {
        const int ln = contig_name_off[rid + 1] - contig_name_off[rid];
        memcpy(o + n, contig_names + contig_name_off[rid], (size_t)ln); n += ln; o[n++] = '\t';
        n += o_putw(o + n, pos + 1); o[n++] = '\t';
        n += o_putw(o + n, r->mapq); o[n++] = '\t';
        n += o_add_cigar(o + n, cg, n_cigar, softclip, r->is_alt, r->which);
    } else { memcpy(o + n, "*\t0\t0\t*", 7); n += 7; }
    o[n++] = '\t';
    if (has_m && m_rid >= 0) {
        if (rid == m_rid) o[n++] = '=';
        else { const int ln = contig_name_off[m_rid + 1] - contig_name_off[m_rid]; memcpy(o + n, contig_names + contig_name_off[m_rid], (size_t)ln); n += ln; }
        o[n++] = '\t';
        n += o_putw(o + n, m_pos + 1); o[n++] = '\t';
        if (rid == m_rid) {
            const long long p0 = pos + (is_rev ? o_rlen(n_cigar, cg) - 1 : 0);
            const long long p1 = m_pos + (m_is_rev ? o_rlen(m_n_cigar, mcg) - 1 : 0);
            if (m_n_cigar == 0 || n_cigar == 0) o[n++] = '0';
            else n += o_putw(o + n, -(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
        } else o[n++] = '0';
    } else { memcpy(o + n, "*\t0\t0", 5); n += 5; }
    o[n++] = '\t';
    if (flag & 0x100) { memcpy(o + n, "*\t*", 3); n += 3; }
    else {
        int qb = 0, qe = l_seq;
        if (n_cigar && r->which && !softclip && !r->is_alt) {
            const int c0 = (int)(cg[0] & 0xf), c1 = (int)(cg[n_cigar - 1] & 0xf);
            if (!is_rev) { if (c0 == 4 || c0 == 3) qb += (int)(cg[0] >> 4); if (c1 == 4 || c1 == 3) qe -= (int)(cg[n_cigar - 1] >> 4); }
            else { if (c0 == 4 || c0 == 3) qe -= (int)(cg[0] >> 4); if (c1 == 4 || c1 == 3) qb += (int)(cg[n_cigar - 1] >> 4); }
        }
        if (!is_rev) for (int i = qb; i < qe; ++i) o[n++] = "ACGTN"[seq[i] > 4 ? 4 : seq[i]];
        else for (int i = qe - 1; i >= qb; --i) o[n++] = "TGCAN"[seq[i] > 4 ? 4 : seq[i]];
        o[n++] = '\t';
        if (qual) {
            if (!is_rev) for (int i = qb; i < qe; ++i) o[n++] = qual[i];
            else for (int i = qe - 1; i >= qb; --i) o[n++] = qual[i];
        } else o[n++] = '*';
    }
    if (n_cigar) {
        memcpy(o + n, "\tNM:i:", 6); n += 6; n += o_putw(o + n, r->NM);
        memcpy(o + n, "\tMD:Z:", 6); n += 6; { const size_t l = strlen(md); memcpy(o + n, md, l); n += (int64_t)l; }
    }
    if (has_m && m_n_cigar) { memcpy(o + n, "\tMC:Z:", 6); n += 6; n += o_add_cigar(o + n, mcg, m_n_cigar, softclip, r->m_is_alt, r->which); }
    if (r->score >= 0) { memcpy(o + n, "\tAS:i:", 6); n += 6; n += o_putw(o + n, r->score); }
    if (r->sub >= 0) { memcpy(o + n, "\tXS:i:", 6); n += 6; n += o_putw(o + n, r->sub); }
    if (rg_id && rg_id[0]) { memcpy(o + n, "\tRG:Z:", 6); n += 6; { const size_t l = strlen(rg_id); memcpy(o + n, rg_id, l); n += (int64_t)l; } }
    if (r->xa_off >= 0) { const char* xa = (const char*)(blob + r->xa_off); const size_t l = strlen(xa); memcpy(o + n, "\tXA:Z:", 6); n += 6; memcpy(o + n, xa, l); n += (int64_t)l; }
    o[n++] = '\n';
    return n;
}

/* ---- mate-rescue Smith-Waterman of the SAM phase (kswv, reference src/kswv.cpp; driver mem_sam_pe_batch, src/bwamem_pair.cpp:719-818) --------
 * The reference runs 64 (int8) or 32 (int16) pairs per AVX-512 vector, one pair per SIMD lane; what one lane computes does not depend on
 * its neighbours (padding rows / columns never hold a row maximum; the cross-lane loop bounds maxl / minh / limit only widen ranges that
 * per-lane masks cut back), so this is the per-lane arithmetic, literally:
 *  - local alignment, E along the query and F along the target, both opened from H (MAIN_SAM_CODE8_OPT / 16_OPT, :60-107); int8 lanes
 *    are unsigned bytes biased by `shift` with saturating adds / subs, int16 lanes plain signed arithmetic;
 *  - the query is padded to the SSE2 stripe of bwa's ksw_u8 / ksw_i16 (16 / 8 columns) with a base that scores 0 against everything
 *    (:296-310, :872-887) -- those columns can hold a row's maximum (never the global one) and so shape score2;
 *  - per row i the maximum and its first column; rowMax[i] keeps it only for rows that are local peaks of that sequence, reach the
 *    KSW_XSUBO threshold and precede the lane's stop (Block I, :505-518 / :1063-1078);
 *  - the global maximum moves only on a strictly larger row maximum (Block II); KSW_XSTOP and, for int8, saturation stop the lane;
 *  - score2 / te2: the largest kept row maximum (first row on ties) outside te +- ceil(score / match) (:594-646 / :1141-1183);
 *  - second pass for KSW_XSTART (bwamem_pair.cpp:768-808): the prefixes query[0..qe], target[0..te] reversed in place, target length
 *    unchanged, KSW_XSTOP | score; tb / qb are set when that pass reaches the same score. */
#define KSW_XBYTE_ 0x10000
#define KSW_XSTOP_ 0x20000
#define KSW_XSUBO_ 0x40000
#define KSW_XSTART_ 0x80000

typedef struct { int score, te, qe, score2, te2; } kswv_fwd;

static void kswv_pass(int is8, const uint8_t* t, int tlen, const uint8_t* q, int qlen, int xtra, int want2, int a, int b, int o_del, int e_del, int o_ins,
                      int e_ins, kswv_fwd* out, int* raw_score, int64_t* cells) {
    const int w_match = a, w_mismatch = -b, w_ambig = -1;
    int mn = w_match < w_mismatch ? w_match : w_mismatch;
    mn = mn < w_ambig ? mn : w_ambig;
    const int shift = is8 ? (256 - (mn & 0xff)) & 0xff : 0;                 /* :398-405 */
    int qmax = w_match > w_mismatch ? w_match : w_mismatch;
    qmax = qmax > w_ambig ? qmax : w_ambig;                                 /* g_qmax, :131-132 */
    const int none = is8 ? 0 : -1;
    const int quanta = is8 ? (qlen + 15) / 16 * 16 : (qlen + 7) / 8 * 8;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int v = (xtra & KSW_XSUBO_) ? (xtra & 0xffff) : 0x10000;
    const int has_minsc = v <= (is8 ? 255 : 32767), minsc = v;
    v = (xtra & KSW_XSTOP_) ? (xtra & 0xffff) : 0x10000;
    const int has_endsc = v <= (is8 ? 255 : 32767), endsc = v;
    int* H0 = (int*)calloc((size_t)quanta + 2, sizeof(int));
    int* H1 = (int*)calloc((size_t)quanta + 2, sizeof(int));
    int* F = (int*)calloc((size_t)quanta + 2, sizeof(int));
    int* rowMax = (int*)malloc(((size_t)tlen + 1) * sizeof(int));
    int gmax = 0, te = -1, qe = 0, pimax = 0, mask = 0, minsc_ok = 0, exited = 0, rows = 0;
    for (int i = 0; i < tlen; ++i) {
        int e = 0, imax = 0, iqe = -1;
        const int tb = t[i];
        for (int j = 0; j < quanta; ++j) {
            int sc, m, h;
            if (j >= qlen) sc = 0;
            else if (tb > 3 || q[j] > 3) sc = w_ambig;
            else sc = tb == q[j] ? w_match : w_mismatch;
            if (is8) {
                m = H0[j] + sc + shift;
                m = m > 255 ? 255 : m;
                m -= shift;
                m = m < 0 ? 0 : m;
                h = m > e ? m : e;
                h = h > F[j + 1] ? h : F[j + 1];
            } else {
                m = H0[j] + sc;
                h = m > e ? m : e;
                h = h > F[j + 1] ? h : F[j + 1];
                h = h > 0 ? h : 0;
            }
            if (h > imax) { imax = h; iqe = j; }
            H1[j + 1] = h;
            int g = h - oe_ins, e2 = e - e_ins, d = h - oe_del, f2 = F[j + 1] - e_del;
            if (is8) { g = g < 0 ? 0 : g; e2 = e2 < 0 ? 0 : e2; d = d < 0 ? 0 : d; f2 = f2 < 0 ? 0 : f2; }
            e = g > e2 ? g : e2;
            F[j + 1] = d > f2 ? d : f2;
        }
        ++rows;
        if (i > 0) {                                                        /* Block I */
            const int msk = (imax > pimax) || mask;
            rowMax[i - 1] = (!msk && minsc_ok && !exited) ? pimax : none;
            mask = !msk;
        }
        pimax = imax;
        minsc_ok = has_minsc && imax >= minsc;
        if (imax > gmax && !exited) { gmax = imax; te = i; qe = is8 ? (iqe & 0xff) : iqe; }   /* Block II */
        if ((has_endsc && gmax >= endsc) || (is8 && gmax + shift >= 255)) exited = 1;
        { int* s = H1; H1 = H0; H0 = s; }
        H0[0] = 0;
        if (exited) {                                                       /* the lane is frozen: later rows keep nothing */
            for (int r = i; r < tlen; ++r) rowMax[r] = none;
            break;
        }
    }
    if (!exited && tlen > 0) rowMax[tlen - 1] = (!mask && minsc_ok) ? pimax : none;
    *raw_score = gmax;
    out->score = is8 ? (gmax + shift < 255 ? gmax : 255) : gmax;
    out->te = te; out->qe = qe; out->score2 = -1; out->te2 = -1;
    if (want2 && !(is8 && out->score == 255)) {
        const int val = (gmax + qmax - 1) / qmax, low = te - val, high = te + val;
        int mx = none, t2 = -1;
        for (int i = 0; i < low; ++i) if (rowMax[i] > mx) { mx = rowMax[i]; t2 = i; }
        for (int i = high + 1; i < tlen; ++i) if (i > high && rowMax[i] > mx) { mx = rowMax[i]; t2 = i; }
        out->score2 = is8 ? (mx == 0 ? -1 : mx) : mx;
        out->te2 = t2;
    }
    if (cells) *cells += (int64_t)rows * quanta;
    free(H0); free(H1); free(F); free(rowMax);
}

void orc_kswv_pair(const uint8_t* target, int tlen, const uint8_t* query, int qlen, int xtra, int a, int b, int o_del, int e_del, int o_ins, int e_ins,
                   orc_kswr* r, int64_t* cells) {
    const int is8 = (xtra & KSW_XBYTE_) != 0;                               /* sort_classify, src/bwamem.cpp:1798-1825 */
    kswv_fwd f;
    int raw = 0;
    kswv_pass(is8, target, tlen, query, qlen, xtra, 1, a, b, o_del, e_del, o_ins, e_ins, &f, &raw, cells);
    r->score = f.score; r->te = f.te; r->qe = f.qe; r->score2 = f.score2; r->te2 = f.te2; r->tb = -1; r->qb = -1;
    if ((xtra & KSW_XSTART_) == 0 || ((xtra & KSW_XSUBO_) && r->score < (xtra & 0xffff))) return;   /* bwamem_pair.cpp:775, :793 */
    const int ql2 = r->qe + 1, tl_rev = r->te + 1;
    uint8_t* q2 = (uint8_t*)malloc((size_t)(ql2 > 0 ? ql2 : 1));
    uint8_t* t2 = (uint8_t*)malloc((size_t)(tlen > 0 ? tlen : 1));
    for (int k = 0; k < ql2; ++k) q2[k] = query[ql2 - 1 - k];
    memcpy(t2, target, (size_t)tlen);
    for (int k = 0; k < tl_rev; ++k) t2[k] = target[tl_rev - 1 - k];
    kswv_fwd g;
    kswv_pass(is8, t2, tlen, q2, ql2, KSW_XSTOP_ | r->score, 0, a, b, o_del, e_del, o_ins, e_ins, &g, &raw, cells);
    if (r->score == raw) { r->tb = r->te - g.te; r->qb = r->qe - g.qe; }
    free(q2); free(t2);
}

int orc_kswv_batch(const orc_kswv_job* jobs, int64_t n, const uint8_t* ref, const uint8_t* qer, int a, int b, int o_del, int e_del, int o_ins, int e_ins,
                   orc_kswr* out, int threads, int64_t* cells) {
    int64_t total = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads) reduction(+ : total)
    for (int64_t i = 0; i < n; ++i) {
        int64_t c = 0;
        orc_kswv_pair(ref + jobs[i].idr, jobs[i].len1, qer + jobs[i].idq, jobs[i].len2, jobs[i].xtra, a, b, o_del, e_del, o_ins, e_ins, &out[i], &c);
        total += c;
    }
    if (cells) *cells = total;
    return 0;
}


/* ------------------------------------------------------------------------------------------------
 * Mate rescue, the posing step (see meme_oracle.h): mem_sam_pe_batch_pre (src/bwamem_pair.cpp:660-716) + mem_matesw_batch_pre (:1060-1223).
 * ------------------------------------------------------------------------------------------------ */
static int o_infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t* dist) {      /* mem_infer_dir, :58-65 */
    const int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
    const int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
    *dist = p2 > b1 ? p2 - b1 : b1 - p2;
    return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}
int64_t orc_matesw_pose(const orc_mate_reg* regs, const int64_t* reg_off, int64_t first, int64_t count, const int32_t* read_len, const orc_pestat* pes,
                        int64_t l_pac, const int64_t* contig_off, const int32_t* contig_len, int n_contigs, const orc_mate_opt* opt,
                        int32_t* gar, int64_t gar_cap, int64_t* n_gar, orc_mate_job* jobs, int64_t job_cap) {
    int64_t pcnt = 0, gcnt = 0;
    for (int64_t p = first; p + 1 < first + count; p += 2) {                /* worker_sam's loop over the batch's pairs (src/bwamem.cpp:1855-1866) */
        for (int i = 0; i < 2; ++i) {                                        /* :694-706 */
            const int64_t r = p + i, m = p + (1 - i);
            const orc_mate_reg* a = regs + reg_off[r];
            const int64_t na = reg_off[r + 1] - reg_off[r];
            const orc_mate_reg* ma = regs + reg_off[m];
            const int64_t nm = reg_off[m + 1] - reg_off[m];
            const int l_ms = read_len[m];
            int taken = 0;
            for (int64_t j = 0; j < na && taken < opt->max_matesw; ++j) {     /* b[i] = the records within pen_unpaired of the best (:688-692), the first max_matesw of them (:696) */
                if (!(a[j].score >= a[0].score - opt->pen_unpaired)) continue;
                ++taken;
                if (gcnt + 4 > gar_cap) return -1;
                int skip[4];
                for (int o = 0; o < 4; ++o) skip[o] = pes[o].failed ? 1 : 0;            /* :1081-1083 */
                for (int64_t k = 0; k < nm; ++k) {                                      /* :1085-1091 */
                    int64_t dist;
                    const int o = o_infer_dir(l_pac, a[j].rb, ma[k].rb, &dist);
                    if (dist >= pes[o].low && dist <= pes[o].high) skip[o] = 1;
                }
                if (skip[0] + skip[1] + skip[2] + skip[3] == 4) { for (int o = 0; o < 4; ++o) gar[gcnt + o] = -1; gcnt += 4; continue; }   /* :1094-1098 */
                int rid = -1;                                                           /* (kept across the orientations, as in the reference: :1080) */
                for (int o = 0; o < 4; ++o) {
                    if (skip[o]) { gar[gcnt + o] = -1; continue; }
                    const int is_rev = (o >> 1) != (o & 1), is_larger = !(o >> 1);      /* :1108-1109 */
                    int64_t rb, re;
                    if (!is_rev) {                                                      /* :1118-1125 */
                        rb = is_larger ? a[j].rb + pes[o].low : a[j].rb - pes[o].high;
                        re = (is_larger ? a[j].rb + pes[o].high : a[j].rb - pes[o].low) + l_ms;
                    } else {
                        rb = (is_larger ? a[j].rb + pes[o].low : a[j].rb - pes[o].high) - l_ms;
                        re = is_larger ? a[j].rb + pes[o].high : a[j].rb - pes[o].low;
                    }
                    if (rb < 0) rb = 0;
                    if (re > l_pac << 1) re = l_pac << 1;
                    if (rb < re) {                                                      /* bns_fetch_seq (:1129; src/bntseq.cpp:541-570) */
                        const int64_t mid = (rb + re) >> 1;
                        const int mrev = mid >= l_pac;
                        rid = o_pos2rid(contig_off, n_contigs, l_pac, mrev ? (l_pac << 1) - 1 - mid : mid);
                        int64_t far_beg = contig_off[rid], far_end = far_beg + contig_len[rid];
                        if (mrev) { const int64_t t = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t; }
                        rb = rb > far_beg ? rb : far_beg;
                        re = re < far_end ? re : far_end;
                    }
                    if (a[j].rid == rid && re - rb >= opt->min_seed_len) {              /* :1131-1218 */
                        if (pcnt >= job_cap) return -1;
                        orc_mate_job J;
                        J.rb = rb; J.read = (int32_t)m; J.len1 = (int32_t)(re - rb); J.len2 = l_ms; J.is_rev = is_rev; J.pad = 0;
                        J.xtra = 0x40000 | 0x80000 | (l_ms * opt->a < 250 ? 0x10000 : 0) | (opt->min_seed_len * opt->a);     /* KSW_XSUBO | KSW_XSTART | KSW_XBYTE | threshold, :1135 */
                        gar[gcnt + o] = (int32_t)pcnt;
                        jobs[pcnt++] = J;
                    } else gar[gcnt + o] = -1;     /* (the reference leaves the entry as it was -- mem_sam_pe_batch_post only looks at entries of orientations it knows were posed; -1 here so that the arrays compare) */
                }
                gcnt += 4;
            }
        }
    }
    *n_gar = gcnt;
    return pcnt;
}

void orc_matesw_job_seqs(const orc_mate_job* j, const uint8_t* text, const uint8_t* reads, const int64_t* read_off, uint8_t* ref, uint8_t* qer) {
    for (int l = 0; l < j->len1; ++l) ref[l] = text[j->rb + l];                       /* bns_get_seq: bases [rb, re) of the fwd+rc text */
    const uint8_t* ms = reads + read_off[j->read];
    for (int l = 0; l < j->len2; ++l) qer[l] = j->is_rev ? (ms[j->len2 - 1 - l] < 4 ? 3 - ms[j->len2 - 1 - l] : 4) : ms[l];      /* :1111-1116 */
}
