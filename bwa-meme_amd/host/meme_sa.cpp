// Suffix-array construction for the BWA-MEME learned index (host side, C++17 + OpenMP).
//
// What has to be reproduced (reference src/Learnedindex.cpp:157-229, 242, 456-548):
//   text  = fwd || revcomp(fwd) || T^k,  k = max(longest A-run, longest T-run) + 1
//   SA    = suffix array of that padded text with the usual "shorter string first" rule
//           (the reference calls saisxx, src/Learnedindex.cpp:242)
//   entries that point into the T padding are dropped  -> 2*l_pac entries survive.
// Any correct suffix sorter yields the same array; this one is a parallel prefix-doubling
// sorter (Larsson-Sadakane style refinement of unsorted groups only), chosen because it is
// insensitive to long repeats and has the same shape as a GPU radix-sort implementation.
#include "meme_host.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>
#include <parallel/algorithm>
#include <omp.h>

namespace meme {

namespace {

// base-5 packing of the first H0 symbols ($=0 < A=1 < C=2 < G=3 < T=4); 5^24 < 2^63
constexpr int H0 = 24;

struct KeyPos {
    uint64_t key;
    uint64_t pos;
};

}  // namespace

int64_t padding_len(const uint8_t* text, int64_t n) {
    int64_t max_a = 0, max_t = 0, ca = 0, ct = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint8_t c = text[i];
        if (c == 0) { ++ca; ct = 0; }
        else if (c == 3) { ++ct; ca = 0; }
        else { ca = ct = 0; }
        max_a = std::max(max_a, ca);
        max_t = std::max(max_t, ct);
    }
    return std::max(max_a, max_t) + 1;
}

// text: codes 0..3, length n (= 2*l_pac, fwd + revcomp).  sa_out: n entries.
void build_suffix_array(const uint8_t* text, int64_t n, uint64_t* sa_out, int threads) {
    if (threads > 0) omp_set_num_threads(threads);
    const int64_t k = padding_len(text, n);
    const int64_t N = n + k;  // padded length
    auto sym = [&](int64_t i) -> uint64_t {  // 0 = end sentinel
        return i < n ? (uint64_t)text[i] + 1 : (i < N ? 4u : 0u);
    };

    // ---- round 0: sort by the first H0 symbols -------------------------------------------
    std::vector<KeyPos> kp((size_t)N);
#pragma omp parallel
    {
        // rolling computation per contiguous chunk
        int nt = omp_get_num_threads(), t = omp_get_thread_num();
        int64_t lo = N * t / nt, hi = N * (t + 1) / nt;
        if (lo < hi) {
            uint64_t key = 0;
            for (int j = 0; j < H0; ++j) key = key * 5 + sym(lo + j);
            uint64_t top = 1;
            for (int j = 1; j < H0; ++j) top *= 5;
            for (int64_t i = lo; i < hi; ++i) {
                kp[(size_t)i] = {key, (uint64_t)i};
                key = (key - sym(i) * top) * 5 + sym(i + H0);
            }
        }
    }
    __gnu_parallel::sort(kp.begin(), kp.end(), [](const KeyPos& a, const KeyPos& b) {
        return a.key < b.key || (a.key == b.key && a.pos < b.pos);
    });

    std::vector<uint64_t> sa((size_t)N);
    std::vector<int64_t> rank((size_t)N);  // rank[pos] = index of the first member of pos's group
    // group boundaries
    std::vector<std::pair<int64_t, int64_t>> groups;  // unsorted groups [s,e)
    {
        std::vector<uint8_t> head((size_t)N + 1);
        head[(size_t)N] = 1;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < N; ++i) {
            sa[(size_t)i] = kp[(size_t)i].pos;
            head[(size_t)i] = (i == 0) || kp[(size_t)i].key != kp[(size_t)i - 1].key;
        }
        // ranks and the list of groups with more than one member, chunk-parallel: a thread owns the groups
        // whose head lies in its chunk and follows the last one past the chunk end
        const int nt = omp_get_max_threads();
        std::vector<std::vector<std::pair<int64_t, int64_t>>> tl((size_t)nt);
#pragma omp parallel num_threads(nt)
        {
            const int t = omp_get_thread_num();
            int64_t lo = N * t / nt, hi = N * (t + 1) / nt;
            while (lo < hi && !head[(size_t)lo]) ++lo;     // first head in the chunk
            int64_t i = lo;
            while (i < hi) {                                // i is a head
                int64_t e = i + 1;
                while (!head[(size_t)e]) ++e;               // head[N] = 1 terminates
                for (int64_t k = i; k < e; ++k) rank[(size_t)sa[(size_t)k]] = i;
                if (e - i > 1) tl[(size_t)t].emplace_back(i, e);
                i = e;
            }
        }
        for (auto& v : tl) groups.insert(groups.end(), v.begin(), v.end());
    }
    std::vector<KeyPos>().swap(kp);

    // ---- refinement rounds ----------------------------------------------------------------
    int64_t h = H0;
    std::vector<std::pair<int64_t, int64_t>> next_groups;
    struct Upd { uint64_t pos; int64_t rank; };
    while (!groups.empty()) {
        std::vector<std::vector<std::pair<int64_t, int64_t>>> tl_groups(omp_get_max_threads());
        std::vector<std::vector<Upd>> tl_upd(omp_get_max_threads());
#pragma omp parallel
        {
            auto& myg = tl_groups[omp_get_thread_num()];
            auto& myu = tl_upd[omp_get_thread_num()];
            std::vector<std::pair<int64_t, uint64_t>> tmp;
#pragma omp for schedule(dynamic, 64)
            for (int64_t gi = 0; gi < (int64_t)groups.size(); ++gi) {
                int64_t s = groups[(size_t)gi].first, e = groups[(size_t)gi].second;
                tmp.clear();
                for (int64_t i = s; i < e; ++i) {
                    uint64_t p = sa[(size_t)i];
                    int64_t r2 = (int64_t)(p + h) < N ? rank[(size_t)(p + h)] : -1;
                    tmp.emplace_back(r2, p);
                }
                std::sort(tmp.begin(), tmp.end());
                int64_t gs = s;
                for (int64_t i = s; i < e; ++i) {
                    sa[(size_t)i] = tmp[(size_t)(i - s)].second;
                    if (i > s && tmp[(size_t)(i - s)].first != tmp[(size_t)(i - s - 1)].first) {
                        if (i - gs > 1) myg.emplace_back(gs, i);
                        gs = i;
                    }
                    myu.push_back({tmp[(size_t)(i - s)].second, gs});
                }
                if (e - gs > 1) myg.emplace_back(gs, e);
            }
        }
        // apply rank updates after the whole round (all reads above saw the previous round's ranks)
#pragma omp parallel for schedule(dynamic, 1)
        for (int t = 0; t < (int)tl_upd.size(); ++t)
            for (const Upd& u : tl_upd[(size_t)t]) rank[(size_t)u.pos] = u.rank;
        next_groups.clear();
        for (auto& v : tl_groups) next_groups.insert(next_groups.end(), v.begin(), v.end());
        groups.swap(next_groups);
        h *= 2;
    }

    // ---- drop the padding suffixes (stable, chunk-parallel compaction) -----------------------------
    {
        const int nt = omp_get_max_threads();
        std::vector<int64_t> cnt((size_t)nt + 1, 0);
#pragma omp parallel num_threads(nt)
        {
            const int t = omp_get_thread_num();
            int64_t lo = N * t / nt, hi = N * (t + 1) / nt, c = 0;
            for (int64_t i = lo; i < hi; ++i) c += (int64_t)sa[(size_t)i] < n;
            cnt[(size_t)t + 1] = c;
#pragma omp barrier
#pragma omp single
            for (int k = 0; k < nt; ++k) cnt[(size_t)k + 1] += cnt[(size_t)k];
            int64_t w = cnt[(size_t)t];
            for (int64_t i = lo; i < hi; ++i)
                if ((int64_t)sa[(size_t)i] < n) sa_out[w++] = sa[(size_t)i];
        }
    }
}

}  // namespace meme
