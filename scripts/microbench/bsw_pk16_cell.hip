// Microbenchmark for the question the round-3 review left open: would a packed-16-bit cell (two pairs per lane, v_pk_* arithmetic) beat
// k_bsw_lane's 32-bit cell?  Both kernels below run the SAME simplified extension DP -- ksw_extend2's recurrences (H from M/E/F with the
// "M ? M + s : 0" rule, E and F with open+extend, row maximum with its rightmost column) over the FULL rectangle: no band, no band
// trimming, no z-drop, no early exit, every pair of a wavefront the same shape -- i.e. the best case for the packed variant, which in
// the real function would also have to keep two pairs' diverging bands apart.  Scores: match a, mismatch -b (no N).
//   k_cell32 : k_bsw_lane's cell as it is (one pair per lane, one 32-bit LDS word per column: q | H << 8 | E << 20)
//   k_cell16 : two pairs per lane, one 64-bit LDS word per column: {q << 12 | H} x 2 in the low dword, E x 2 in the high one
// Build + run (MI355X):  hipcc --offload-arch=gfx950 -O3 -o bsw_pk16_cell bsw_pk16_cell.hip && ./bsw_pk16_cell
// Output: GCUPS of both, and both are checked against a scalar CPU version of the same recurrences.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int QL = 128, TL = 128;          // every pair: 128 x 128 cells
constexpr int A_ = 1, B_ = 4, O_DEL = 6, E_DEL = 1, O_INS = 6, E_INS = 1;

typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short u2 __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
union W32 { unsigned u; s2 s; u2 us; };

__device__ __forceinline__ int max3_i32(int a, int b, int c) { int d; asm("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ unsigned he_repack(int h, int e, unsigned old) {
    unsigned x, w;
    asm("v_lshl_or_b32 %0, %1, 12, %2" : "=v"(x) : "v"(e), "v"(h));
    asm("v_perm_b32 %0, %1, %2, %3" : "=v"(w) : "v"(x), "v"(old), "s"(0x06050400u));
    return w;
}

// packed 16-bit instructions by name: left to itself the compiler turns min(x, 1) and the multiply by a 0/1 mask into per-half
// compares and selects (12 v_cndmask + 17 v_cmp per four columns in the first build of this file)
#define PK2(name_, insn_) __device__ __forceinline__ unsigned name_(unsigned a, unsigned b) { unsigned d; asm(insn_ " %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
PK2(pk_min_u16, "v_pk_min_u16")
PK2(pk_max_i16, "v_pk_max_i16")
PK2(pk_add_u16, "v_pk_add_u16")
PK2(pk_sub_i16, "v_pk_sub_i16")
PK2(pk_mul_u16, "v_pk_mul_lo_u16")
// (an inline constant feeds only the low half of a packed operand: the shift count comes from a register holding 15 in both halves)
__device__ __forceinline__ unsigned pk_ashr15(unsigned a) { unsigned d; asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(d) : "v"(0x000f000fu), "v"(a)); return d; }
__device__ __forceinline__ unsigned pk_mad_u16(unsigned a, unsigned b, unsigned c) { unsigned d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

// one pair per lane; out[p] = (max score << 16) | (row of the maximum << 8 ... ) simplified: max score and its rightmost column summed over rows
__global__ void __launch_bounds__(64) k_cell32(const uint8_t* __restrict__ q, const uint8_t* __restrict__ t, int npairs, int* __restrict__ out) {
    extern __shared__ unsigned he_raw[];
    typedef __attribute__((address_space(3))) unsigned* lds_u32;
    const int lane = threadIdx.x;
    const lds_u32 he = (lds_u32)he_raw + lane;
    const int oe_del = O_DEL + E_DEL, oe_ins = O_INS + E_INS;
    for (int p0 = blockIdx.x * 64; p0 < npairs; p0 += gridDim.x * 64) {
        const int p = p0 + lane;
        const uint8_t* qs = q + (size_t)p * QL;
        const uint8_t* ts = t + (size_t)p * TL;
        const int h0 = 40;
        for (int j = 0; j <= QL; ++j) {
            int v = j == 0 ? h0 : (h0 - (O_INS + E_INS * j) > 0 ? h0 - (O_INS + E_INS * j) : 0);
            he[j * 64] = ((unsigned)v << 8) | (j < QL ? qs[j] : 0u);
        }
        int acc = 0;
        for (int i = 0; i < TL; ++i) {
            const int tb = ts[i];
            int f = 0, h1 = h0 - (O_DEL + E_DEL * (i + 1));
            if (h1 < 0) h1 = 0;
            const unsigned sb4 = (unsigned)((-B_) & 0xff) * 0x01010101u;
            const unsigned tab_lo = (sb4 & ~(0xffu << (8 * tb))) | ((unsigned)(A_ & 0xff) << (8 * tb));
            const unsigned tab_hi = 0xffffffffu;
            unsigned wa = he[0], wb = 0, wc = 0, wd = 0;
            int mk0 = -8, mk1 = -8, mk2 = -8, mk3 = -8;
#define CELL(cur_, nxt_, j_, mk_, jb_)                                                            \
            {                                                                                      \
                nxt_ = he[((j_) + 1) * 64];                                                        \
                int M = (int)((cur_ >> 8) & 0xfffu), e = (int)(cur_ >> 20);                         \
                const int sc = (int)(signed char)(__builtin_amdgcn_perm(tab_hi, tab_lo, cur_) & 0xffu); \
                M = M ? M + sc : 0;                                                                \
                const int h = max3_i32(M, e, f);                                                   \
                const int key = (h << 10) + (jb_);                                                 \
                mk_ = mk_ > key ? mk_ : key;                                                       \
                e = max3_i32(M - oe_del, e - E_DEL, 0);                                            \
                he[(j_) * 64] = he_repack(h1, e, cur_);                                            \
                h1 = h;                                                                            \
                f = max3_i32(M - oe_ins, f - E_INS, 0);                                            \
            }
            for (int j = 0; j < QL; j += 4) {
                CELL(wa, wb, j, mk0, j)
                CELL(wb, wc, j + 1, mk1, j)
                CELL(wc, wd, j + 2, mk2, j)
                CELL(wd, wa, j + 3, mk3, j)
            }
#undef CELL
            mk1 += 1; mk2 += 2; mk3 += 3;
            const int ka = mk0 > mk1 ? mk0 : mk1, kb = mk2 > mk3 ? mk2 : mk3;
            const int key = ka > kb ? ka : kb;
            he[QL * 64] = ((unsigned)h1 << 8);
            acc += key;                       // (row maximum << 10 | rightmost column), summed over the rows
        }
        if (p < npairs) out[p] = acc;
    }
}

// two pairs per lane: pair 2*l in the low halves, pair 2*l+1 in the high halves
__global__ void __launch_bounds__(64) k_cell16(const uint8_t* __restrict__ q, const uint8_t* __restrict__ t, int npairs, int* __restrict__ out) {
    extern __shared__ unsigned he_raw[];
    typedef __attribute__((address_space(3))) v2u* lds_u64;
    const int lane = threadIdx.x;
    const lds_u64 he = (lds_u64)he_raw + lane;            // column j: he[j * 64] = {H halves with q in their top 4 bits, E halves}
    const W32 oe_del = {(unsigned)(O_DEL + E_DEL) * 0x00010001u}, oe_ins = {(unsigned)(O_INS + E_INS) * 0x00010001u};
    const W32 e_del = {(unsigned)E_DEL * 0x00010001u}, e_ins = {(unsigned)E_INS * 0x00010001u};
    const W32 a_pk = {(unsigned)A_ * 0x00010001u}, one = {0x00010001u}, zero = {0u};
    const unsigned mab_pk = (unsigned)((-(A_ + B_)) & 0xffff) * 0x00010001u;
    for (int p0 = blockIdx.x * 128; p0 < npairs; p0 += gridDim.x * 128) {
        const int pa = p0 + 2 * lane, pb = pa + 1;
        const uint8_t *qa = q + (size_t)pa * QL, *qb = q + (size_t)pb * QL, *ta = t + (size_t)pa * TL, *tb_ = t + (size_t)pb * TL;
        const int h0 = 40;
        for (int j = 0; j <= QL; ++j) {
            unsigned v = j == 0 ? h0 : (h0 - (O_INS + E_INS * j) > 0 ? h0 - (O_INS + E_INS * j) : 0);
            const unsigned qq = j < QL ? ((unsigned)qa[j] << 12) | ((unsigned)qb[j] << 28) : 0u;
            he[j * 64] = v2u{v * 0x00010001u | qq, 0u};
        }
        int acc_a = 0, acc_b = 0;
        for (int i = 0; i < TL; ++i) {
            W32 tbpk; tbpk.u = (unsigned)ta[i] | ((unsigned)tb_[i] << 16);
            int h1i = h0 - (O_DEL + E_DEL * (i + 1));
            if (h1i < 0) h1i = 0;
            W32 h1; h1.u = (unsigned)h1i * 0x00010001u;
            W32 f = zero, m, mj, jpk = zero;
            m.u = 0xffffffffu; mj.u = 0;             // (-1: below every H, and h - m cannot overflow 16 bits)
            v2u cur = he[0];
#pragma unroll 4
            for (int j = 0; j < QL; ++j) {
                const v2u nxt = he[(j + 1) * 64];
                const unsigned M = cur.x & 0x0fff0fffu;
                const unsigned qv = (cur.x >> 12) & 0x000f000fu;
                unsigned e = cur.y;
                const unsigned nz = pk_min_u16(qv ^ tbpk.u, one.u);
                const unsigned sc = pk_mad_u16(nz, mab_pk, a_pk.u);                  // a - (a + b) * (q != t)
                const unsigned M2 = pk_mul_u16(pk_add_u16(M, sc), pk_min_u16(M, one.u));   // M ? M + s : 0
                const unsigned h = pk_max_i16(pk_max_i16(M2, e), f.u);
                const unsigned ge = ~pk_ashr15(pk_sub_i16(h, m.u));                  // all ones where h >= m: the rightmost column among equal maxima
                mj.u = (mj.u & ~ge) | (jpk.u & ge);
                m.u = pk_max_i16(m.u, h);
                e = pk_max_i16(pk_max_i16(pk_sub_i16(M2, oe_del.u), pk_sub_i16(e, e_del.u)), 0u);
                he[j * 64] = v2u{h1.u | (cur.x & 0xf000f000u), e};
                h1.u = h;
                f.u = pk_max_i16(pk_max_i16(pk_sub_i16(M2, oe_ins.u), pk_sub_i16(f.u, e_ins.u)), 0u);
                jpk.u += 0x00010001u;
                cur = nxt;
            }
            he[QL * 64] = v2u{h1.u, 0u};
            acc_a += ((int)m.s.x << 10) + (int)mj.us.x;
            acc_b += ((int)m.s.y << 10) + (int)mj.us.y;
        }
        if (pa < npairs) out[pa] = acc_a;
        if (pb < npairs) out[pb] = acc_b;
    }
}

static int cpu_pair(const uint8_t* q, const uint8_t* t) {
    int H[QL + 1], E[QL + 1];
    const int h0 = 40, oe_del = O_DEL + E_DEL, oe_ins = O_INS + E_INS;
    for (int j = 0; j <= QL; ++j) { H[j] = j == 0 ? h0 : (h0 - (O_INS + E_INS * j) > 0 ? h0 - (O_INS + E_INS * j) : 0); E[j] = 0; }
    int acc = 0;
    for (int i = 0; i < TL; ++i) {
        int f = 0, h1 = h0 - (O_DEL + E_DEL * (i + 1)), m = -1, mj = 0;
        if (h1 < 0) h1 = 0;
        for (int j = 0; j < QL; ++j) {
            int M = H[j], e = E[j];
            H[j] = h1;
            M = M ? M + (q[j] == t[i] ? A_ : -B_) : 0;
            int h = M > e ? M : e; h = h > f ? h : f;
            h1 = h;
            if (h >= m) { m = h; mj = j; }
            int tt = M - oe_del; e -= E_DEL; e = e > tt ? e : tt; E[j] = e > 0 ? e : 0;
            tt = M - oe_ins; f -= E_INS; f = f > tt ? f : tt; f = f > 0 ? f : 0;
        }
        H[QL] = h1; E[QL] = 0;
        acc += (m << 10) + mj;
    }
    return acc;
}

int main() {
    const int npairs = 1 << 20;
    std::vector<uint8_t> q((size_t)npairs * QL), t((size_t)npairs * TL);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int p = 0; p < npairs; ++p) {
        for (int j = 0; j < QL; ++j) q[(size_t)p * QL + j] = (uint8_t)(rnd() & 3);
        for (int j = 0; j < TL; ++j) t[(size_t)p * TL + j] = (rnd() % 100 < 96 && j < QL) ? q[(size_t)p * QL + j] : (uint8_t)(rnd() & 3);   // ~3 % mismatches
    }
    uint8_t *dq, *dt; int *d32, *d16;
    CK(hipMalloc(&dq, q.size())); CK(hipMalloc(&dt, t.size())); CK(hipMalloc(&d32, npairs * 4)); CK(hipMalloc(&d16, npairs * 4));
    CK(hipMemcpy(dq, q.data(), q.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dt, t.data(), t.size(), hipMemcpyHostToDevice));
    const size_t lds32 = (size_t)(QL + 2) * 64 * 4, lds16 = (size_t)(QL + 2) * 64 * 8;
    CK(hipFuncSetAttribute((const void*)k_cell16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double cells = (double)npairs * QL * TL;
    for (int which = 0; which < 2; ++which) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            if (which == 0) hipLaunchKernelGGL(k_cell32, dim3(npairs / 64), dim3(64), lds32, 0, dq, dt, npairs, d32);
            else hipLaunchKernelGGL(k_cell16, dim3(npairs / 128), dim3(64), lds16, 0, dq, dt, npairs, d16);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("%s: %.2f ms for %d pairs of %d x %d cells = %.0f GCUPS\n", which ? "k_cell16 (two pairs per lane, packed 16-bit)" : "k_cell32 (k_bsw_lane's cell)            ", best, npairs,
               QL, TL, cells / best / 1e6);
    }
    std::vector<int> o32(npairs), o16(npairs);
    CK(hipMemcpy(o32.data(), d32, npairs * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o16.data(), d16, npairs * 4, hipMemcpyDeviceToHost));
    int bad32 = 0, bad16 = 0;
    for (int p = 0; p < 20000; ++p) { const int w = cpu_pair(&q[(size_t)p * QL], &t[(size_t)p * TL]); bad32 += o32[p] != w; bad16 += o16[p] != w; }
    for (int p = 0; p < npairs; ++p) bad16 += o16[p] != o32[p];
    printf("mismatches against the scalar CPU version (20 000 pairs): 32-bit %d, packed %d (and packed vs 32-bit on all pairs)\n", bad32, bad16);
    return bad32 || bad16;
}
