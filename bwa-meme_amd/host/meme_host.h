// Host-side (CPU, C++17) pieces of the MI355X BWA-MEME seeding/extension path:
// index construction in the reference's on-disk formats, the P-RMI trainer, file I/O.
// None of this is on the per-read hot path; it produces / loads what the HIP backend stages in HBM.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace meme {

// ---- suffix array (meme_sa.cpp) ------------------------------------------------------------
int64_t padding_len(const uint8_t* text, int64_t n);
void build_suffix_array(const uint8_t* text, int64_t n, uint64_t* sa_out, int threads);

// ---- reference text helpers (meme_formats.cpp) ----------------------------------------------
// fwd codes 0..3 -> fwd || revcomp  (the ".0123" image, reference src/Learnedindex.cpp:72-131)
std::vector<uint8_t> make_fwd_rc(const uint8_t* fwd, int64_t l_pac);
// 32-base key of the suffix at pos, first base in the top bits, reading the *padded* text
// (T beyond 2*l_pac) -- the ".suffixarray_uint64" training key (src/Learnedindex.cpp:481-500)
uint64_t train_key(const uint8_t* text, int64_t n, uint64_t pos);

struct Contig {
    std::string name;
    int64_t offset;
    int32_t len;
};

// writers for every file `bwa-meme index -a meme` leaves behind
bool write_pac_ann_amb(const std::string& prefix, const uint8_t* fwd, int64_t l_pac,
                       const std::vector<Contig>& contigs);
bool write_0123(const std::string& prefix, const uint8_t* text, int64_t n);
bool write_pos_packed(const std::string& prefix, const uint64_t* sa, int64_t n);
bool write_suffixarray_uint64(const std::string& prefix, const uint8_t* text, int64_t n,
                              const uint64_t* sa);
bool read_pos_packed(const std::string& prefix, std::vector<uint64_t>& sa);
bool read_0123(const std::string& prefix, std::vector<uint8_t>& text);
bool read_file(const std::string& path, std::vector<uint8_t>& bytes);

// ---- P-RMI trainer (meme_prmi.cpp) -----------------------------------------------------------
// 24-byte record of the reference's parameter files (reference src/LearnedIndex_seeding.cpp:197-206)
struct RmiRecord {
    double intercept;
    double slope;
    uint64_t err;
};
struct Prmi {
    int bits = 0;                 // 2^bits leaf models, addressed by key >> (64-bits)
    std::vector<RmiRecord> l2;    // leaves           -> "_L2_PARAMETERS"
    std::vector<RmiRecord> l1;    // partial 3rd layer -> "_L1_PARAMETERS"
};
int default_rmi_bits(int64_t sa_num);  // build_rmis_dna.sh:68-77
// keys[i] = train_key of SA slot i (sorted ascending)
void train_prmi(const uint64_t* keys, int64_t n, int bits, int partial_threshold, Prmi& out,
                int threads);
bool write_prmi(const std::string& prefix, const Prmi& m);
bool read_prmi(const std::string& prefix, Prmi& m);
// the reference's learned_index_lookup(), bit for bit (src/LearnedIndex_seeding.cpp:186-210)
uint64_t prmi_lookup(const Prmi& m, double sa_num, uint64_t key, uint64_t* err);

}  // namespace meme
