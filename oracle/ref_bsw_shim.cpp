// TEST INFRASTRUCTURE ONLY (oracle/): a C shim of OURS over the *reference's* C++ class
// BandedPairWiseSW (reference src/bandedSWA.h:118-135, 257-297) so the test-suite can call the
// reference's scalar and SIMD banded-SW kernels through ctypes.  It is compiled together with the
// reference's own src/bandedSWA.cpp and src/ksw.cpp, where they lie, by oracle/Makefile.ref into
// oracle/_ref/libbsw_ref.so.  Contains no reference code.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "bandedSWA.h"   // reference header, found via -I$(REF)/src at build time
#include "ksw.h"

// globals the reference objects expect from the rest of the binary
uint64_t prof[10][112];
uint64_t tprof[LIM_R][LIM_C];
uint64_t proc_freq = 1, tprof_unused;

extern "C" {

struct bsw_params {
    int32_t o_del, e_del, o_ins, e_ins, zdrop, end_bonus, a, b;
};

static void fill_mat(int8_t mat[25], int a, int b) {
    // bwa_fill_scmat semantics (reference src/bwa.cpp:262-270): match a, mismatch -b, ambiguous -1
    int k = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) mat[k++] = i == j ? a : -b;
        mat[k++] = -1;
    }
    for (int j = 0; j < 5; ++j) mat[k++] = -1;
}

// kind: 0 = scalarBandedSWAWrapper, 16 = getScores16, 8 = getScores8
// The pair array must have room for +64 entries of padding (the SIMD wrappers write past numPairs).
int ref_bsw_run(int kind, SeqPair* pairs, uint8_t* ref, uint8_t* qer, int32_t n, int32_t w,
                const bsw_params* p) {
    int8_t mat[25];
    fill_mat(mat, p->a, p->b);
    BandedPairWiseSW bsw(p->o_del, p->e_del, p->o_ins, p->e_ins, p->zdrop, p->end_bonus, mat,
                         (int8_t)p->a, (int8_t)p->b, 1);
    if (kind == 0) bsw.scalarBandedSWAWrapper(pairs, ref, qer, n, 1, w);
    else if (kind == 16) bsw.getScores16(pairs, ref, qer, n, 1, w);
    else if (kind == 8) bsw.getScores8(pairs, ref, qer, n, 1, w);
    else return -1;
    return 0;
}

// ksw_extend2 (reference src/ksw.cpp:434-535), the scalar twin of scalarBandedSWA
int ref_ksw_extend2(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int w, int h0,
                    const bsw_params* p, int32_t out[6]) {
    int8_t mat[25];
    fill_mat(mat, p->a, p->b);
    int qle, tle, gtle, gscore, max_off;
    int sc = ksw_extend2(qlen, query, tlen, target, 5, mat, p->o_del, p->e_del, p->o_ins, p->e_ins, w,
                         p->end_bonus, p->zdrop, h0, &qle, &tle, &gtle, &gscore, &max_off);
    out[0] = sc; out[1] = tle; out[2] = gtle; out[3] = qle; out[4] = gscore; out[5] = max_off;
    return 0;
}

int ref_sizeof_seqpair(void) { return (int)sizeof(SeqPair); }

}  // extern "C"
