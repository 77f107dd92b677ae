// skyjo_layout.h - per-game record layout shared by host and device code.
//
// One game = one packed record of `state_bytes` (multiple of 16).  64 games form a TILE owned by
// one wavefront.  In HBM a tile is stored chunk-major: 16-byte chunk c of lane l sits at
//     tile_base + (c * 64 + l) * 16
// so that one global_load_dwordx4 per lane moves 1 KiB fully coalesced.  In LDS the same tile is
// dword-interleaved: byte b of lane l sits at
//     (b >> 2) * 256 + l * 4 + (b & 3)
// so lane l only ever touches LDS bank l % 32 - every byte access, however data-dependent its
// index, is bank-conflict free.
#pragma once
#include <stdint.h>

#define SK_TILE 64
#define SK_NCARDS 150

// ---- fixed header (bytes) ----
// words 0..2 are the hot fields: the step kernel keeps them in three registers for a whole launch
#define H_PHASE 0     // 0 draw, 1 place              (expected_action[1], skyjo.py:144)
#define H_PLAYER 1    // expected player              (expected_action[0])
#define H_FLAGS 2     // F_* bits
#define H_STATUS 3    // SKYJO_ST_* of the last step
#define H_NDRAW 4     // len(drawpile)
#define H_NDISC 5     // len(discard_pile)
#define H_ROLE 6      // which end of the pile buffer holds the draw pile (flips at a reshuffle)
#define H_TOP 7       // discard top or -3            (skyjo.py:254)  -> obs[17]
#define H_EPLEN 8     // u16 steps in this episode
#define H_RESH 10     // reshuffles in this episode (saturating)
#define H_HAND 11     // hand card or 15              (skyjo.py:61)   -> obs[18]
#define H_EPISODE 12  // u32 deal index of the live episode
// bytes 16..32 are laid out so that obs[k] == state[16 + k] for k < 17 (skyjo.py:180-184)
#define H_MINSUM 16   // min(min_p revealed_sum_p, 127)
#define H_MINHID 17   // min_p hidden_count_p
#define H_HIST 18     // u8[15] bincount of values -2..12 (updated with dword LDS atomics)
#define H_FINISHER 33
#define H_BANK 34      // which of the game's SK_BANK pre-dealt episodes is taken next (mirrors SkParams.bank_head)
#define H_END 36

#define F_TERMINATED 1  // is_terminated (skyjo.py:54)
#define F_DONE 2        // env-level done (natural end or illegal action)
#define F_VALID 4       // game slot in use (padding lanes of the last tile are not)

struct SkLayout {
  int32_t N, indirect, D, Dp, rec_bytes;
  int32_t off_sums, off_placed, off_hidden, off_refunded, off_cards, off_vis, off_pile;
  int32_t state_bytes, chunks;  // chunks = state_bytes / 16
};

#ifdef __HIPCC__
#define SK_HD __host__ __device__
#else
#define SK_HD
#endif
SK_HD static inline SkLayout sk_make_layout(int N, int indirect) {
  SkLayout L;
  L.N = N;
  L.indirect = indirect ? 1 : 0;
  L.D = indirect ? 31 : 19 + 12 * N;  // skyjo.py:43-45
  L.Dp = (L.D + 3) & ~3;
  L.rec_bytes = (L.Dp + 32 + 15) & ~15;
  L.off_sums = H_END;                 // i16[N] revealed sums
  L.off_placed = L.off_sums + 2 * N;  // u16[N] num_placed
  L.off_hidden = L.off_placed + 2 * N;   // u8[N] hidden counts
  L.off_refunded = L.off_hidden + N;     // u8[N] num_refunded
  L.off_cards = (L.off_refunded + N + 3) & ~3;  // i8[N][12] true cards (-14 once refunded)
  L.off_vis = L.off_cards + 12 * N;   // i8[N][12] what an observer sees: card / 15 hidden / -14 refunded
  L.off_pile = L.off_vis + 12 * N;    // i8[150] draw pile from one end, discard pile from the other
  L.state_bytes = (L.off_pile + SK_NCARDS + 15) & ~15;
  L.chunks = L.state_bytes / 16;
  return L;
}
