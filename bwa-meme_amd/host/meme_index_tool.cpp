// meme-index: builds the BWA-MEME learned index in the reference's on-disk formats.
//
//   meme-index build <ref.fa> [-p prefix] [-t threads] [-b rmi_bits] [-T partial_threshold]
//       FASTA -> .pac .ann .amb .0123 .pos_packed .suffixarray_uint64 + _L{0,1,2}_PARAMETERS
//       (what `bwa-meme index -a meme` + build_rmis_dna.sh leave behind; reference
//        src/bwtindex.cpp:344-376, src/Learnedindex.cpp:134-555, build_rmis_dna.sh:68-128)
//   meme-index train <prefix> [-t threads] [-b rmi_bits] [-T partial_threshold]
//       only the P-RMI, from an existing .pos_packed + .0123 (e.g. written by the reference's indexer)
#include "meme_host.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <omp.h>

using namespace meme;

namespace {

struct Hole { int64_t offset; int32_t len; char amb; };

double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// FASTA -> forward codes; ambiguous bases are replaced by lrand48()&3 after srand48(11), the
// same stream bns_fasta2bntseq consumes (reference src/bntseq.cpp:264-311, 331-332).
bool load_fasta(const char* path, std::vector<uint8_t>& fwd, std::vector<Contig>& contigs,
                std::vector<std::string>& annos, std::vector<Hole>& holes,
                std::vector<int>& n_ambs) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "[meme-index] cannot open %s\n", path); return false; }
    static uint8_t tab[256];
    memset(tab, 4, sizeof(tab));
    tab['A'] = tab['a'] = 0; tab['C'] = tab['c'] = 1; tab['G'] = tab['g'] = 2; tab['T'] = tab['t'] = 3;
    srand48(11);
    std::vector<char> line(1 << 20);
    int lasts = 0;
    while (fgets(line.data(), (int)line.size(), f)) {
        size_t L = strlen(line.data());
        while (L && (line[L - 1] == '\n' || line[L - 1] == '\r')) line[--L] = 0;
        if (line[0] == '>') {
            std::string hdr(line.data() + 1);
            size_t sp = hdr.find_first_of(" \t");
            Contig c;
            c.name = hdr.substr(0, sp);
            c.offset = (int64_t)fwd.size();
            c.len = 0;
            contigs.push_back(c);
            annos.push_back(sp == std::string::npos ? "(null)" : hdr.substr(sp + 1));
            n_ambs.push_back(0);
            lasts = 0;
            continue;
        }
        if (contigs.empty()) continue;
        for (size_t i = 0; i < L; ++i) {
            int ch = (unsigned char)line[i];
            if (ch == ' ' || ch == '\t') continue;
            int c = tab[ch];
            if (c >= 4) {
                if (lasts == ch) ++holes.back().len;
                else {
                    holes.push_back({(int64_t)fwd.size(), 1, (char)ch});
                    ++n_ambs.back();
                }
                c = (int)(lrand48() & 3);
            }
            lasts = ch;
            fwd.push_back((uint8_t)c);
            ++contigs.back().len;
        }
    }
    fclose(f);
    return !fwd.empty();
}

bool write_ann_amb(const std::string& prefix, int64_t l_pac, const std::vector<Contig>& contigs,
                   const std::vector<std::string>& annos, const std::vector<Hole>& holes,
                   const std::vector<int>& n_ambs) {
    FILE* f = fopen((prefix + ".ann").c_str(), "w");
    if (!f) return false;
    fprintf(f, "%lld %d %u\n", (long long)l_pac, (int)contigs.size(), 11u);
    for (size_t i = 0; i < contigs.size(); ++i) {
        fprintf(f, "%d %s", 0, contigs[i].name.c_str());
        if (!annos[i].empty()) fprintf(f, " %s\n", annos[i].c_str());
        else fprintf(f, "\n");
        fprintf(f, "%lld %d %d\n", (long long)contigs[i].offset, contigs[i].len, n_ambs[i]);
    }
    fclose(f);
    f = fopen((prefix + ".amb").c_str(), "w");
    if (!f) return false;
    fprintf(f, "%lld %d %u\n", (long long)l_pac, (int)contigs.size(), (unsigned)holes.size());
    for (const Hole& h : holes) fprintf(f, "%lld %d %c\n", (long long)h.offset, h.len, h.amb);
    fclose(f);
    return true;
}

int train_from(const std::string& prefix, const std::vector<uint8_t>& text,
               const std::vector<uint64_t>& sa, int bits, int pthr, int threads) {
    const int64_t n = (int64_t)sa.size();
    if (bits <= 0) bits = default_rmi_bits(n);
    double t0 = now();
    std::vector<uint64_t> keys((size_t)n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) keys[(size_t)i] = train_key(text.data(), n, sa[(size_t)i]);
    Prmi m;
    train_prmi(keys.data(), n, bits, pthr, m, threads);
    if (!write_prmi(prefix, m)) return 1;
    fprintf(stderr, "[meme-index] P-RMI: 2^%d leaves, %zu partial models, %.2f s\n", bits, m.l1.size(),
            now() - t0);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: meme-index build <ref.fa> [-p prefix] [-t threads] [-b bits] [-T thr]\n"
                        "       meme-index train <prefix> [-t threads] [-b bits] [-T thr]\n");
        return 2;
    }
    std::string cmd = argv[1], in = argv[2], prefix = argv[2];
    int threads = omp_get_max_threads(), bits = 0, pthr = 1000;
    for (int i = 3; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "-p")) prefix = argv[i + 1];
        else if (!strcmp(argv[i], "-t")) threads = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-b")) bits = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-T")) pthr = atoi(argv[i + 1]);
    }
    omp_set_num_threads(threads);
    if (cmd == "train") {
        std::vector<uint8_t> text;
        std::vector<uint64_t> sa;
        if (!read_0123(prefix, text) || !read_pos_packed(prefix, sa)) return 1;
        if (text.size() != sa.size()) { fprintf(stderr, "[meme-index] .0123/.pos_packed size mismatch\n"); return 1; }
        return train_from(prefix, text, sa, bits, pthr, threads);
    }
    if (cmd != "build") { fprintf(stderr, "unknown command %s\n", cmd.c_str()); return 2; }

    std::vector<uint8_t> fwd;
    std::vector<Contig> contigs;
    std::vector<std::string> annos;
    std::vector<Hole> holes;
    std::vector<int> n_ambs;
    double t0 = now();
    if (!load_fasta(in.c_str(), fwd, contigs, annos, holes, n_ambs)) return 1;
    const int64_t l_pac = (int64_t)fwd.size();
    fprintf(stderr, "[meme-index] %lld bases in %zu contigs (%.2f s)\n", (long long)l_pac, contigs.size(), now() - t0);
    if (!write_pac_ann_amb(prefix, fwd.data(), l_pac, contigs)) return 1;
    if (!write_ann_amb(prefix, l_pac, contigs, annos, holes, n_ambs)) return 1;
    std::vector<uint8_t> text = make_fwd_rc(fwd.data(), l_pac);
    const int64_t n = 2 * l_pac;
    if (!write_0123(prefix, text.data(), n)) return 1;
    t0 = now();
    std::vector<uint64_t> sa((size_t)n);
    build_suffix_array(text.data(), n, sa.data(), threads);
    fprintf(stderr, "[meme-index] suffix array of %lld suffixes: %.2f s\n", (long long)n, now() - t0);
    if (!write_pos_packed(prefix, sa.data(), n)) return 1;
    if (!write_suffixarray_uint64(prefix, text.data(), n, sa.data())) return 1;
    return train_from(prefix, text, sa, bits, pthr, threads);
}
