// Shared declarations of the HIP backend (gfx950 / MI355X only -- no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <utility>
#include <vector>

#include "meme_hip.h"

typedef uint64_t u64;
typedef int64_t i64;

// ---- HBM layout of the index ---------------------------------------------------------------------
//   keys[n16 + 16]  u64: the first 32 bases of every suffix, in suffix-array order, as an integer whose unsigned order
//                   is the lexicographic order (first base in bits 63..62, T-filled past the text end).  128-byte
//                   aligned: one search window = 16 keys = one line.  Slots >= n hold ~0.
//   pos5[5n + 16]   the reference's .pos_packed image as on disk (u32 LE pos >> 8, then u8 pos & 0xff): text position of
//                   every slot, read only for key ties and by the hit gather.
//   pac[n/32 + 8]   2-bit fwd+rc text, base i in bits (62 - 2*(i&31)) of word i>>5, T past the end.
//   l2[], l1[]      P-RMI records (reference src/LearnedIndex_seeding.cpp:197-206) padded to 32 bytes.
//   special[4096]   open-addressing hash set of the window lines (slot >> 4, stored +1) that hold a suffix shorter than
//                   SPECIAL_SPAN bases -- the only places where the end of the text can influence a compare.
struct Rmi32 {
    double icpt;
    double slope;
    u64 err;
    u64 pad;
};

struct RmiRec {   // on-disk P-RMI record
    double icpt;
    double slope;
    u64 err;
};

constexpr int SPECIAL_SPAN = 544;          // >= LEARNED_MAX_READ_LEN + 32 + slack
constexpr int SPECIAL_SLOTS = 4096;

struct DevIndex {
    i64 n = 0;                 // sa_num = 2 * l_pac
    const u64* keys = nullptr;
    const uint8_t* pos5 = nullptr;
    const u64* pac = nullptr;
    const Rmi32* l2 = nullptr;
    const Rmi32* l1 = nullptr;
    i64 n_l2 = 0, n_l1 = 0;
    int shift = 64;            // key >> shift = leaf index
    const uint32_t* special = nullptr;
};

struct DevBuf {   // growable device workspace
    void* p = nullptr;
    size_t cap = 0;
};

struct meme_ctx {
    int device = 0;
    int n_cus = 256;                   // compute units of the device (cached: hipGetDeviceProperties is slow)
    hipStream_t stream = nullptr;
    DevIndex idx;
    bool owns_index = false;
    std::vector<std::pair<void*, size_t>> owned;   // device allocations of the index (pointer, bytes)
    // workspaces
    DevBuf reads, read_off, slots[4], ovf[2], slot_cnt, slot_hits, slot_loc, smem_off, hit_off, smems, hits,
           scan_tmp, counters, pairs, refb, qerb, packed, bsw_order, bsw_ws;
    // pinned host staging owned by the ctx (results of meme_seed_batch_host, inputs of meme_bsw_batch)
    struct HostBuf { void* p = nullptr; size_t cap = 0; } h_smems, h_hits, h_smem_off, h_hit_off, h_misc;
    // tuning
    i64 seed_blocks = 0;               // 0 = auto
    i64 smem_cap = 64;                 // per-read SMEM slots in the search kernel's scratch (tier 0)
    i64 seed_waves_per_cu = 0;         // resident wavefronts per CU of the search kernel (0 = what LDS and registers allow)
    i64 bsw_blocks = 0;
    i64 bsw_lane_min_pairs = 32768;   // batches at least this big use the lane-per-pair kernel (throughput); smaller ones the
                                       // lanes-per-pair kernel (latency: a lone pair takes ~6 ms on one lane, ~0.3 ms on 64)
    // timings
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    meme_timings tm = {0, 0, 0, 0, 0, 0, 0, 0};
};

void meme_set_error(const char* fmt, ...);
int meme_buf_reserve(meme_ctx* ctx, DevBuf& b, size_t bytes);
int meme_hostbuf_reserve(meme_ctx* ctx, meme_ctx::HostBuf& b, size_t bytes);

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            meme_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MEME_E_HIP;                                                                 \
        }                                                                                      \
    } while (0)

// ---- device helpers --------------------------------------------------------------------------------
// text position of suffix-array slot i from the 5-byte image: the two aligned dwords around the record
__device__ __forceinline__ u64 load_pos5(const uint8_t* pos5, i64 i) {
    const i64 bo = i * 5;
    const uint32_t* pd = reinterpret_cast<const uint32_t*>(pos5 + (bo & ~3ll));
    const u64 two = (u64)pd[0] | ((u64)pd[1] << 32);
    const u64 v5 = (two >> (8 * (int)(bo & 3))) & 0xffffffffffull;
    return ((v5 & 0xffffffffull) << 8) | (v5 >> 32);
}
// 32 bases starting at base offset s from a big-endian-within-word 2-bit array
__device__ __forceinline__ u64 extract32(const u64* w, i64 s) {
    i64 k = s >> 5;
    int sh = (int)(s & 31) * 2;
    u64 a = w[k], b = w[k + 1];
    return sh ? (a << sh) | (b >> (64 - sh)) : a;
}
