// Shared layout helpers of the decode kernels (decode.hip, decode_fused.hip); see decode.hip's header comment.
#pragma once
#include "common.h"

constexpr int PAGE = 64;
constexpr int PAGE_ELEMS = PAGE * 128;     // per (kv head, K|V)

// Fragment-order operand layouts of the skinny (M <= 16) GEMMs: one 1 KiB chunk per MFMA operand,
//   weights  Wd[(n_tile*(K/32) + kstep)*64 + lane][8],  lane = g*16 + i  <->  W[16*n_tile + i][32*kstep + 8g .. +7]
//   inputs   Xf[kstep*64 + lane][8],                     lane = g*16 + m  <->  X[m][32*kstep + 8g .. +7]
// so every wave-level load is one contiguous, fully used 1 KiB global_load_dwordx4 (8 cache lines per
// instruction instead of 64 quarter-used sectors with row-major operands: lm_head 2.1 -> 6.0 TB/s in round 1).
DEVI size_t frag_off(int m, int k) { return ((size_t)((k >> 5) * 64 + ((k >> 3) & 3) * 16 + m)) * 8 + (k & 7); }

// 4 features (m, k..k+3) -> fragment-order input buffer (8-byte store)
DEVI void store_frag4(bf16_t* __restrict__ xf, int m, int k, float a, float b, float c, float d) {
    u32x2 pk = {pack_bf2(a, b), pack_bf2(c, d)};
    *reinterpret_cast<u32x2*>(xf + frag_off(m, k)) = pk;
}

// KV page element offsets (layout in decode.hip's header)
DEVI int k_chunk(int key, int d) { return ((key >> 4) * 4 + (d >> 5)) * 64 + ((d >> 3) & 3) * 16 + (key & 15); }
DEVI int v_off(int key, int d) {
    const int kk = key & 31;
    return (((key >> 5) * 8 + (d >> 4)) * 64 + ((kk >> 2) & 3) * 16 + (d & 15)) * 8 + 4 * (kk >> 4) + (kk & 3);
}
