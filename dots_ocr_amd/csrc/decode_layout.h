// Shared layout helpers of the decode kernels (decode.hip, decode_fused.hip); see decode.hip's header comment.
#pragma once
#include "common.h"

constexpr int PAGE = 64;
constexpr int PAGE_ELEMS = PAGE * 128;     // per (kv head, K|V)

// Operand layouts of the skinny (B <= 16 rows) GEMMs of the decode step, both in MFMA fragment order
// (v_mfma_f32_16x16x32_bf16: lane = g*16 + i holds 8 consecutive k of row i, k-step = 32 k):
//   weights  Wd[(n_tile*(K/32) + kstep)*64 + lane][8],  lane = g*16 + i  <->  W[16*n_tile + i][32*kstep + 8g .. +7]
//            one contiguous 1 KiB chunk per (tile, k-step); the two 8-row halves of a chunk are four whole 128-B lines each
//   inputs   X[(k/8)*XR + m][8]  <->  X[m][8*(k/8) .. +7],   XR = 8 (B <= 8) or 16 rows
//            lane (g, m) of k-step ks reads slot (4*ks + g)*XR + (m & (XR-1)): with XR = 8 lanes m and m+8 read the same 16 B,
//            so an 8-row batch costs half the L2 traffic and half the LDS of the 16-row image.
// fp8 weights (quant.hip): the chunk of a (tile, k-step) is 512 B, lane (g, i)'s 8 bytes sit at 8-byte slot [i >> 3][g][i & 7]: each 8-row half
// of the tile is 256 contiguous bytes (two whole lines).
DEVI int fp8_lane_slot(int g, int i) { return ((i >> 3) * 4 + g) * 8 + (i & 7); }
DEVI int xr_of(int B) { return B <= 8 ? 8 : 16; }
//            Batches above 16 rows are split into 16-ROW TILES (one MFMA column tile each): tile t = rows 16t .. 16t+15 is a complete
//            XR = 16 image of its own at element offset t * 16 * K, and every dense decode kernel runs one workgroup set per tile
//            (blockIdx.y), i.e. sees a batch of <= 16 rows.
DEVI size_t xfrag_off(int m, int k, int XR) { return ((size_t)(k >> 3) * XR + m) * 8 + (k & 7); }       // inside one tile: m < 16
DEVI size_t ximage_off(int m, int k, int XR, int K) { return (size_t)(m >> 4) * 16 * K + xfrag_off(m & 15, k, XR); }   // any row of the batch
constexpr int MAX_DECODE_ROWS = 64;        // sequences per decode step (4 tiles)

// 4 features (m, k..k+3), k % 4 == 0 -> X image (8-byte store)
DEVI void store_frag4(bf16_t* __restrict__ xf, int m, int k, int XR, float a, float b, float c, float d) {
    u32x2 pk = {pack_bf2(a, b), pack_bf2(c, d)};
    *reinterpret_cast<u32x2*>(xf + xfrag_off(m, k, XR)) = pk;
}

// KV page element offsets (layout in decode.hip's header)
DEVI int k_chunk(int key, int d) { return ((key >> 4) * 4 + (d >> 5)) * 64 + ((d >> 3) & 3) * 16 + (key & 15); }
DEVI int v_off(int key, int d) {
    const int kk = key & 31;
    return (((key >> 5) * 8 + (d >> 4)) * 64 + ((kk >> 2) & 3) * 16 + (d & 15)) * 8 + 4 * (kk >> 4) + (kk & 3);
}
