// C API of the host library (libmeme_host.so) for Python/ctypes callers: in-memory index construction
// (suffix array + P-RMI) and writers for the reference's on-disk formats.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <omp.h>

#include "meme_host.h"

using namespace meme;

extern "C" {

int meme_host_default_bits(int64_t sa_num) { return default_rmi_bits(sa_num); }

// fwd: l_pac codes 0..3.  text_out: 2*l_pac bytes (fwd + revcomp).  sa_out: 2*l_pac entries.
int meme_host_build_sa(const uint8_t* fwd, int64_t l_pac, uint8_t* text_out, uint64_t* sa_out, int threads) {
    if (!fwd || !text_out || !sa_out || l_pac <= 0) return -2;
    if (threads > 0) omp_set_num_threads(threads);
    std::vector<uint8_t> t = make_fwd_rc(fwd, l_pac);
    memcpy(text_out, t.data(), t.size());
    build_suffix_array(text_out, 2 * l_pac, sa_out, threads);
    return 0;
}

int meme_host_train_prmi(const uint8_t* text, int64_t n, const uint64_t* sa, int bits, int partial_threshold,
                         int threads, void** l2, int64_t* l2_records, void** l1, int64_t* l1_records) {
    if (!text || !sa || !l2 || !l1 || n <= 0) return -2;
    if (threads > 0) omp_set_num_threads(threads);
    if (bits <= 0) bits = default_rmi_bits(n);
    std::vector<uint64_t> keys((size_t)n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) keys[(size_t)i] = train_key(text, n, sa[i]);
    Prmi m;
    train_prmi(keys.data(), n, bits, partial_threshold > 0 ? partial_threshold : 1000, m, threads);
    *l2_records = (int64_t)m.l2.size();
    *l1_records = (int64_t)m.l1.size();
    *l2 = malloc(m.l2.size() * 24);
    *l1 = malloc(m.l1.size() * 24 + 24);
    if (!*l2 || !*l1) return -1;
    memcpy(*l2, m.l2.data(), m.l2.size() * 24);
    memcpy(*l1, m.l1.data(), m.l1.size() * 24);
    return 0;
}

void meme_host_free(void* p) { free(p); }

// every file the reference aligner / seeding harness opens for -7; the training-key file
// (.suffixarray_uint64) only when with_keys != 0
int meme_host_write_index(const char* prefix, const uint8_t* fwd, int64_t l_pac, const uint8_t* text,
                          const uint64_t* sa, const void* l1, int64_t l1_records, const void* l2,
                          int64_t l2_records, int n_contigs, int with_keys) {
    if (!prefix || !fwd || !text || !sa || !l2) return -2;
    std::string p(prefix);
    std::vector<Contig> contigs;
    if (n_contigs < 1) n_contigs = 1;
    for (int c = 0; c < n_contigs; ++c) {
        int64_t a = l_pac * c / n_contigs, b = l_pac * (c + 1) / n_contigs;
        contigs.push_back({"chrS" + std::to_string(c + 1), a, (int32_t)(b - a)});
    }
    if (!write_pac_ann_amb(p, fwd, l_pac, contigs)) return -3;
    // write_pac_ann_amb leaves no annotation column; the reference parser accepts both forms
    if (!write_0123(p, text, 2 * l_pac)) return -3;
    if (!write_pos_packed(p, sa, 2 * l_pac)) return -3;
    if (with_keys && !write_suffixarray_uint64(p, text, 2 * l_pac, sa)) return -3;
    Prmi m;
    m.l2.resize((size_t)l2_records);
    memcpy(m.l2.data(), l2, (size_t)l2_records * 24);
    m.l1.resize((size_t)(l1_records > 0 ? l1_records : 1));
    if (l1_records > 0) memcpy(m.l1.data(), l1, (size_t)l1_records * 24);
    m.bits = __builtin_ctzll((unsigned long long)l2_records);
    if (!write_prmi(p, m)) return -3;
    return 0;
}

}  // extern "C"
