// P-RMI ("partial 3-layer recursive model index") trainer and evaluator, C++17 + OpenMP.
//
// The reference trains this model with an offline Rust tool (RMI/rmi_lib/src/train/two_layer.rs:
// 1406-1995) that cannot be built here (no Rust toolchain); the aligner only ever *reads* the two
// parameter files.  This trainer writes files with exactly the layout the aligner loads
// (reference src/LearnedIndex_seeding.cpp:74-122, 186-210; RMI/rmi_lib/src/codegen.rs:1135-1158):
//
//   <prefix>.suffixarray_uint64_L2_PARAMETERS : 2^bits records {f64 intercept, f64 slope, u64 err},
//        leaf m serves keys with (key >> (64-bits)) == m
//   <prefix>.suffixarray_uint64_L1_PARAMETERS : the concatenated partial third-layer records
//   <prefix>.suffixarray_uint64_L0_PARAMETERS : one u64 (bits); never read by the aligner
//
//   err of a plain leaf      : bits 61..32 = error below the prediction, bits 30..0 = error above
//   err of a leaf with partials (bucket > threshold keys): bit 63 | first_partial<<32 | n_partial,
//        and the leaf's line predicts the *local partial index* (two_layer.rs:392-402, 1536).
//
// The model is a search hint: seeding results do not depend on its values, only
// the number of probes does.  Parameter-level parity with the Rust trainer is therefore neither
// possible nor needed; what matters is that the reference binary loads these files (it does) and
// that the errors bound the true lower-bound position of any query key that reaches the record.
#include "meme_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <omp.h>

namespace meme {

namespace {

inline double linear(double alpha, double beta, double inp) { return std::fma(beta, inp, alpha); }
inline size_t fclamp(double inp, double bound) {
    if (inp < 0.0) return 0;
    return inp > bound ? (size_t)bound : (size_t)inp;
}

// Fit one record on keys[s,e) (non-empty).  The line is anchored on (lo_key -> s) and
// (hi_key -> e): lo_key/hi_key are the nearest keys just outside the bucket (or the bucket's own
// end keys when the span would make the slope numerically unsafe), so that every query routed
// here is predicted inside [s, e] up to rounding.  Errors are then measured, including the
// lower-bound positions of absent keys between neighbours.
RmiRecord fit_segment(const uint64_t* keys, int64_t n, int64_t s, int64_t e, uint64_t dom_lo,
                      uint64_t dom_hi) {
    RmiRecord r;
    uint64_t k0 = keys[s], k1 = keys[e - 1];
    // candidate anchors: domain bounds clipped by neighbouring keys
    uint64_t a_lo = dom_lo, a_hi = dom_hi;
    if (s > 0 && keys[s - 1] >= dom_lo) a_lo = keys[s - 1];
    if (e < n && keys[e] <= dom_hi) a_hi = keys[e];
    (void)k0; (void)k1;
    double x0 = (double)a_lo, x1 = (double)a_hi;
    double y0 = (double)s, y1 = (double)e;
    double slope = 0.0, icpt = y0;
    if (x1 > x0) {
        slope = (y1 - y0) / (x1 - x0);
        // cancellation guard: |slope * key| * 2^-53 must stay well below one position
        if (slope * 1.8446744073709552e19 * 1.1102230246251565e-16 > 0.25) {
            slope = 0.0;
            icpt = 0.5 * (y0 + y1);
        } else {
            icpt = y0 - slope * x0;
        }
    } else {
        icpt = 0.5 * (y0 + y1);
    }
    r.intercept = icpt;
    r.slope = slope;
    // measure errors.  For a query q with keys[i-1] < q <= keys[i] the lower-bound position is i
    // (first slot whose key >= q); with equal keys the reads' longer suffixes can land anywhere in
    // the run, so cover [first, last+1] of every run; predictions are monotone in q.
    double lo_err = 1.0, hi_err = 1.0;
    int64_t i = s;
    double pred_prev = linear(icpt, slope, (double)a_lo);
    while (i < e) {
        int64_t j = i;
        while (j + 1 < e && keys[j + 1] == keys[i]) ++j;
        double p = linear(icpt, slope, (double)keys[i]);
        // queries in (prev key, keys[i]] : truth in [i, j+1], prediction in [pred_prev, p]
        lo_err = std::max(lo_err, p - (double)i);
        hi_err = std::max(hi_err, (double)(j + 1) - pred_prev);
        pred_prev = p;
        i = j + 1;
    }
    double p_hi = linear(icpt, slope, (double)a_hi);
    lo_err = std::max(lo_err, p_hi - (double)e);
    hi_err = std::max(hi_err, (double)e - pred_prev);
    uint64_t lo = (uint64_t)std::ceil(lo_err) + 2, hi = (uint64_t)std::ceil(hi_err) + 2;
    lo = std::min<uint64_t>(lo, 0x3fffffffu);
    hi = std::min<uint64_t>(hi, 0x7fffffffu);
    r.err = (lo << 32) | hi;
    return r;
}

RmiRecord constant_record(int64_t idx) {
    RmiRecord r;
    r.intercept = (double)idx;
    r.slope = 0.0;
    r.err = ((uint64_t)2 << 32) | 2u;
    return r;
}

}  // namespace

int default_rmi_bits(int64_t sa_num) {
    // build_rmis_dna.sh:68-77 keys the choice on the size of .suffixarray_uint64 (8 B per entry)
    double bytes = 8.0 * (double)sa_num + 8.0;
    if (bytes > 8.0e9) return 28;
    if (bytes > 1.0e9) return 26;
    return 24;
}

void train_prmi(const uint64_t* keys, int64_t n, int bits, int partial_threshold, Prmi& out,
                int threads) {
    if (threads > 0) omp_set_num_threads(threads);
    const int64_t nleaf = (int64_t)1 << bits;
    const int shift = 64 - bits;
    out.bits = bits;
    out.l2.assign((size_t)nleaf, RmiRecord{0, 0, 0});
    // leaf boundaries: start[m] = first index with key>>shift >= m
    std::vector<int64_t> start((size_t)nleaf + 1);
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m <= nleaf; ++m) {
        if (m == nleaf) { start[(size_t)m] = n; continue; }
        uint64_t lo_key = shift == 64 ? 0 : ((uint64_t)m << shift);
        start[(size_t)m] = std::lower_bound(keys, keys + n, lo_key) - keys;
    }
    // first pass: count partial models needed per leaf
    std::vector<int64_t> np((size_t)nleaf, 0), pstart((size_t)nleaf + 1, 0);
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < nleaf; ++m) {
        int64_t c = start[(size_t)m + 1] - start[(size_t)m];
        if (c > partial_threshold) np[(size_t)m] = (int64_t)std::llround((double)c / 20.0);
    }
    for (int64_t m = 0; m < nleaf; ++m) pstart[(size_t)m + 1] = pstart[(size_t)m] + np[(size_t)m];
    out.l1.assign((size_t)std::max<int64_t>(pstart[(size_t)nleaf], 1), RmiRecord{0, 0, 0});

#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t m = 0; m < nleaf; ++m) {
        int64_t s = start[(size_t)m], e = start[(size_t)m + 1];
        uint64_t dom_lo = (uint64_t)m << shift;
        uint64_t dom_hi = dom_lo + (((uint64_t)1 << shift) - 1);
        if (e == s) { out.l2[(size_t)m] = constant_record(s); continue; }
        if (np[(size_t)m] == 0) { out.l2[(size_t)m] = fit_segment(keys, n, s, e, dom_lo, dom_hi); continue; }
        // leaf with a partial third layer: the leaf line maps key -> local partial index
        const int64_t P = np[(size_t)m];
        RmiRecord leaf;
        double x0 = (double)keys[s], x1 = (double)keys[e - 1];
        if (x1 > x0) {
            leaf.slope = (double)P / (x1 - x0) * (1.0 - 1e-9);
            if (leaf.slope * 1.8446744073709552e19 * 1.1102230246251565e-16 > 0.25) {
                leaf.slope = (double)P / ((double)dom_hi - (double)dom_lo);
                leaf.intercept = -leaf.slope * (double)dom_lo;
            } else {
                leaf.intercept = -leaf.slope * x0;
            }
        } else {  // every key identical: a single partial would do; keep P with constant routing
            leaf.slope = 0.0;
            leaf.intercept = 0.0;
        }
        leaf.err = ((uint64_t)1 << 63) | ((uint64_t)pstart[(size_t)m] << 32) | (uint64_t)P;
        out.l2[(size_t)m] = leaf;
        // route keys with the very same arithmetic the aligner uses
        auto route = [&](uint64_t k) -> int64_t {
            return (int64_t)fclamp(linear(leaf.intercept, leaf.slope, (double)k), (double)P - 1.0);
        };
        int64_t i = s;
        for (int64_t j = 0; j < P; ++j) {
            int64_t ps = i;
            while (i < e && route(keys[i]) <= j) ++i;
            // (route is monotone non-decreasing in the key, keys are sorted)
            RmiRecord rec;
            if (i == ps) rec = constant_record(ps);
            else rec = fit_segment(keys, n, ps, i, dom_lo, dom_hi);
            out.l1[(size_t)(pstart[(size_t)m] + j)] = rec;
        }
    }
}

bool write_prmi(const std::string& prefix, const Prmi& m) {
    auto wr = [&](const std::string& path, const void* p, size_t bytes) {
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) return false;
        size_t w = fwrite(p, 1, bytes, f);
        fclose(f);
        return w == bytes;
    };
    uint64_t b = (uint64_t)m.bits;
    return wr(prefix + ".suffixarray_uint64_L0_PARAMETERS", &b, 8) &&
           wr(prefix + ".suffixarray_uint64_L1_PARAMETERS", m.l1.data(), m.l1.size() * 24) &&
           wr(prefix + ".suffixarray_uint64_L2_PARAMETERS", m.l2.data(), m.l2.size() * 24);
}

bool read_prmi(const std::string& prefix, Prmi& m) {
    std::vector<uint8_t> a, b;
    if (!read_file(prefix + ".suffixarray_uint64_L1_PARAMETERS", a)) return false;
    if (!read_file(prefix + ".suffixarray_uint64_L2_PARAMETERS", b)) return false;
    m.l1.resize(a.size() / 24);
    m.l2.resize(b.size() / 24);
    if (!a.empty()) memcpy(m.l1.data(), a.data(), m.l1.size() * 24);
    memcpy(m.l2.data(), b.data(), m.l2.size() * 24);
    size_t nm = m.l2.size();
    if (nm == 0 || (nm & (nm - 1)) != 0) {
        fprintf(stderr, "[meme] L2 parameter file must hold a power-of-two number of records\n");
        return false;
    }
    m.bits = __builtin_ctzll(nm);
    return true;
}

uint64_t prmi_lookup(const Prmi& m, double sa_num, uint64_t key, uint64_t* err) {
    size_t idx = m.bits == 0 ? 0 : (size_t)(key >> (64 - m.bits));
    const RmiRecord* r = &m.l2[idx];
    double fpred = linear(r->intercept, r->slope, (double)key);
    *err = r->err;
    if (*err >> 63) {
        size_t pstart = (size_t)((*err >> 32) & 0x7fffffff);
        double pnum = (double)(*err & 0xffffffffu);
        r = &m.l1[pstart + fclamp(fpred, pnum - 1)];
        fpred = linear(r->intercept, r->slope, (double)key);
        *err = r->err;
    }
    return (uint64_t)fclamp(fpred, sa_num - 1.0);
}

}  // namespace meme
