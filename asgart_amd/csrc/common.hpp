// common.hpp -- shared host/device definitions of libasgart_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/asgart_hip.h"

namespace asgart {

void set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            ::asgart::set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                                __LINE__);                                                  \
            return e_ == hipErrorOutOfMemory ? ASGART_E_OOM : ASGART_E_HIP;                 \
        }                                                                                   \
    } while (0)

#define RC_TRY(expr)            \
    do {                        \
        int32_t rc_ = (expr);   \
        if (rc_ != 0) return rc_; \
    } while (0)

// Grow-only device buffer (workspace reuse across calls: no hipMalloc in the
// steady state).
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int32_t reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) {
            (void)hipFree(p);
            p = nullptr;
            cap = 0;
        }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipMalloc(&p, bytes + 256);
            want = bytes + 256;
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
            return ASGART_E_OOM;
        }
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T *as() const {
        return reinterpret_cast<T *>(p);
    }
};

// ---- base codes ----------------------------------------------------------
// 3-bit codes that preserve the byte order of the reference's alphabet
// ('$' 0x24 < 'A' < 'C' < 'G' < 'N' < 'T'), so that integer comparison of
// packed k-mers == bytewise comparison of the k-mers (reference
// src/searcher.rs:150 `a.cmp(b)`).
//   '$' (or past the end of the text) = 0, A=1, C=2, G=3, N=4, T=5
__host__ __device__ inline uint32_t base_code(uint8_t c) {
    // h = (c>>1)&7 : A->0 C->1 T->2 G->3 N->7
    const uint32_t lut = (1u << 0) | (2u << 4) | (5u << 8) | (3u << 12) | (4u << 28);
    return c < 0x40 ? 0u : ((lut >> (4 * ((c >> 1) & 7))) & 7u);
}
// complement on codes (reference src/utils.rs:1-17): A<->T, C<->G, N->N
__host__ __device__ inline uint32_t comp_code(uint32_t code) {
    const uint32_t lut = (0u) | (5u << 4) | (3u << 8) | (2u << 12) | (4u << 16) | (1u << 20);
    return (lut >> (4 * code)) & 7u;
}
// code -> 2-bit digit for the ACGT-only prefix table; 4 = not representable
__host__ __device__ inline uint32_t acgt_digit(uint32_t code) {
    const uint32_t lut = (4u) | (0u << 4) | (1u << 8) | (2u << 12) | (4u << 16) | (3u << 20);
    return (lut >> (4 * code)) & 7u;
}

constexpr int kMaxKey = 21;        // bases in one key word: 21 * 3 bits = 63
constexpr int kMaxK = 2 * kMaxKey; // probe sizes 22..42: the first 21 bases are the key word, the rest is
                                   // compared through the text (search_dev.hpp: tail_key)
constexpr int kCacheLen = 8;       // reference src/searcher.rs:15
constexpr int kCacheEntries = 390625;  // 5^8
constexpr uint32_t kSkipN = 0xFFFFFFFFu;     // probe skipped: first base 'N'
constexpr uint32_t kSkipCard = 0xFFFFFFFEu;  // probe skipped: > max_cardinality
constexpr uint32_t kPending = 0xFFFFFFFDu;   // large interval, counted by the wave kernel

inline bool valid_text_byte(uint8_t c) {
    return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N' || c == '$';
}

}  // namespace asgart
