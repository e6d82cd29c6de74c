// On-disk formats of the BWA-MEME learned index, written and read by our
// own code.  Each writer cites the reference producer it must stay byte-compatible with.
#include "meme_host.h"

#include <cstdio>
#include <cstring>
#include <omp.h>

namespace meme {

std::vector<uint8_t> make_fwd_rc(const uint8_t* fwd, int64_t l_pac) {
    std::vector<uint8_t> t((size_t)(2 * l_pac));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < l_pac; ++i) {
        t[(size_t)i] = fwd[i];
        t[(size_t)(2 * l_pac - 1 - i)] = 3 - fwd[i];
    }
    return t;
}

uint64_t train_key(const uint8_t* text, int64_t n, uint64_t pos) {
    // T-filled past the text end: monotone in SA order, which is what the trainer needs.  (The
    // reference's file writer wraps modulo the padded length instead, see train_key_wrap below.)
    uint64_t key = 0;
    for (int r = 0; r < 32; ++r) {
        uint64_t p = pos + (uint64_t)r;
        uint8_t c = p < (uint64_t)n ? text[p] : 3;
        key = (key << 2) | c;
    }
    return key;
}

// exact variant honouring the modulo wrap of the reference writer
static uint64_t train_key_wrap(const uint8_t* text, int64_t n, int64_t k, uint64_t pos) {
    uint64_t key = 0;
    const uint64_t N = (uint64_t)(n + k);
    for (int r = 0; r < 32; ++r) {
        uint64_t p = (pos + (uint64_t)r) % N;
        uint8_t c = p < (uint64_t)n ? text[p] : 3;
        key = (key << 2) | c;
    }
    return key;
}

static bool write_all(const std::string& path, const void* p, size_t bytes) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "[meme] cannot write %s\n", path.c_str()); return false; }
    size_t w = bytes ? fwrite(p, 1, bytes, f) : 0;
    fclose(f);
    return w == bytes;
}

bool read_file(const std::string& path, std::vector<uint8_t>& bytes) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "[meme] cannot open %s\n", path.c_str()); return false; }
    fseek(f, 0, SEEK_END);
    long long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    bytes.resize((size_t)sz);
    size_t r = sz ? fread(bytes.data(), 1, (size_t)sz, f) : 0;
    fclose(f);
    return r == (size_t)sz;
}

// .pac / .ann / .amb  -- bns_fasta2bntseq(fp, prefix, for_only=1) + bns_dump
// (reference src/bntseq.cpp:82-113, 313-371).  Input has no ambiguous bases left (the caller
// replaced them the way add1() does), so n_holes = 0.
bool write_pac_ann_amb(const std::string& prefix, const uint8_t* fwd, int64_t l_pac,
                       const std::vector<Contig>& contigs) {
    std::vector<uint8_t> pac((size_t)(l_pac / 4 + 2), 0);
    for (int64_t l = 0; l < l_pac; ++l) pac[(size_t)(l >> 2)] |= fwd[l] << ((~l & 3) << 1);
    size_t bytes = (size_t)((l_pac >> 2) + ((l_pac & 3) == 0 ? 0 : 1));
    if ((l_pac & 3) == 0) pac[bytes++] = 0;
    pac[bytes++] = (uint8_t)(l_pac & 3);
    if (!write_all(prefix + ".pac", pac.data(), bytes)) return false;
    FILE* f = fopen((prefix + ".ann").c_str(), "w");
    if (!f) return false;
    fprintf(f, "%lld %d %u\n", (long long)l_pac, (int)contigs.size(), 11u);
    for (const Contig& c : contigs) {
        fprintf(f, "%d %s\n", 0, c.name.c_str());
        fprintf(f, "%lld %d %d\n", (long long)c.offset, c.len, 0);
    }
    fclose(f);
    f = fopen((prefix + ".amb").c_str(), "w");
    if (!f) return false;
    fprintf(f, "%lld %d %u\n", (long long)l_pac, (int)contigs.size(), 0u);
    fclose(f);
    return true;
}

// .0123 -- 1 byte/base fwd+rc (reference src/Learnedindex.cpp:199-223)
bool write_0123(const std::string& prefix, const uint8_t* text, int64_t n) {
    return write_all(prefix + ".0123", text, (size_t)n);
}

// .pos_packed -- u32 LE (pos>>8) then u8 (pos&0xff), SA order (src/Learnedindex.cpp:475-478)
bool write_pos_packed(const std::string& prefix, const uint64_t* sa, int64_t n) {
    std::vector<uint8_t> buf((size_t)n * 5);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        uint32_t hi = (uint32_t)(sa[i] >> 8);
        memcpy(&buf[(size_t)i * 5], &hi, 4);
        buf[(size_t)i * 5 + 4] = (uint8_t)(sa[i] & 0xff);
    }
    return write_all(prefix + ".pos_packed", buf.data(), buf.size());
}

// .suffixarray_uint64 -- u64 count then count keys (src/Learnedindex.cpp:239-257, 481-500)
bool write_suffixarray_uint64(const std::string& prefix, const uint8_t* text, int64_t n,
                              const uint64_t* sa) {
    std::vector<uint64_t> buf((size_t)n + 1);
    buf[0] = (uint64_t)n;
    const int64_t k = padding_len(text, n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) buf[(size_t)i + 1] = train_key_wrap(text, n, k, sa[i]);
    return write_all(prefix + ".suffixarray_uint64", buf.data(), buf.size() * 8);
}

bool read_pos_packed(const std::string& prefix, std::vector<uint64_t>& sa) {
    std::vector<uint8_t> raw;
    if (!read_file(prefix + ".pos_packed", raw)) return false;
    size_t n = raw.size() / 5;
    sa.resize(n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        uint32_t hi;
        memcpy(&hi, &raw[(size_t)i * 5], 4);
        sa[(size_t)i] = ((uint64_t)hi << 8) | raw[(size_t)i * 5 + 4];
    }
    return true;
}

bool read_0123(const std::string& prefix, std::vector<uint8_t>& text) {
    return read_file(prefix + ".0123", text);
}

}  // namespace meme
