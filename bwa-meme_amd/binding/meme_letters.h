// Letters of a read to base codes in place (the binding's own loop, round 6) -- a header of its own so that a host-side test can compile it
// (tests/test_letters_host.py: every byte value, every length and offset against the reference's table).
#pragma once
#include <immintrin.h>

extern unsigned char nst_nt4_table[256];      // reference src/bntseq.cpp:63-80 (libbwa_pic.so)

// Letters to base codes in place, as mem_kernel1_core_Learned leaves a read for the later stages (src/bwamem.cpp:1277-1279: c < 4 ? c : nst_nt4_table[c]).  The byte loop
// was 1.3-1.7 of the binding's 7.6 CPU-seconds per 8 M reads (profiles/r06_host_cpu.md): 64 bytes at a time where the build has AVX-512BW (the reference's
// own build flag); a block that holds anything but A C G T N (either case) or codes below 4 goes through the table, so every byte gets the table's value.
static inline void letters_to_codes(char* p, int n) {
    int k = 0;
#if defined(__AVX512BW__)
    const __m512i c_df = _mm512_set1_epi8((char)0xDF), cA = _mm512_set1_epi8('A'), cC = _mm512_set1_epi8('C'), cG = _mm512_set1_epi8('G'), cT = _mm512_set1_epi8('T'),
                  cN = _mm512_set1_epi8('N'), four = _mm512_set1_epi8(4);
    for (; k + 64 <= n; k += 64) {
        const __m512i v = _mm512_loadu_si512((const void*)(p + k));
        const __m512i u = _mm512_and_si512(v, c_df);
        const __mmask64 mA = _mm512_cmpeq_epi8_mask(u, cA), mC = _mm512_cmpeq_epi8_mask(u, cC), mG = _mm512_cmpeq_epi8_mask(u, cG), mT = _mm512_cmpeq_epi8_mask(u, cT),
                        mN = _mm512_cmpeq_epi8_mask(u, cN), mS = _mm512_cmplt_epu8_mask(v, four);
        if ((mA | mC | mG | mT | mN | mS) != ~(__mmask64)0) {                  // some other byte: the table, byte by byte
            for (int j = k; j < k + 64; ++j) { const char c = p[j]; p[j] = c < 4 ? c : (char)nst_nt4_table[(int)c]; }
            continue;
        }
        __m512i r = four;                                                       // N
        r = _mm512_mask_mov_epi8(r, mT, _mm512_set1_epi8(3));
        r = _mm512_mask_mov_epi8(r, mG, _mm512_set1_epi8(2));
        r = _mm512_mask_mov_epi8(r, mC, _mm512_set1_epi8(1));
        r = _mm512_mask_mov_epi8(r, mA, _mm512_setzero_si512());
        r = _mm512_mask_mov_epi8(r, mS, v);                                     // codes stay what they are
        _mm512_storeu_si512((void*)(p + k), r);
    }
#endif
    for (; k < n; ++k) { const char c = p[k]; p[k] = c < 4 ? c : (char)nst_nt4_table[(int)c]; }
}
