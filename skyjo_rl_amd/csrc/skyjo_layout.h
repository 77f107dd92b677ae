// skyjo_layout.h - per-game record layout shared by host and device code.
//
// One game = one packed record of `state_bytes` (multiple of 16).  64 games form a TILE owned by
// one wavefront.  In HBM and in LDS a tile is stored chunk-major: 16-byte chunk c of lane l sits at
//     tile_base + (c * 64 + l) * 16
// so that one 16-byte-per-lane access of a wavefront (global_load_dwordx4, ds_read_b128, or an
// LDS-DMA global_load_lds_dwordx4, which writes  M0 + offset + lane * 16) moves 1 KiB contiguously.
// The record is laid out so that everything a turn reads together is ONE chunk: the observation
// statistics (chunk 1), a player's card row with his sum / hidden / refunded counters (chunk
// 12 + 2p at three players), the row an observer sees with the placed counter (chunk 13 + 2p).
// Data-dependent byte accesses (pile top, card slot, histogram bin) of different lanes fall into
// lane-private 16-byte columns: at worst a few lanes share a bank.
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
// bytes 16..32 are laid out so that obs[k] == state[16 + k] for k < 17 (skyjo.py:180-184): chunk 1 is obs[0..15]
#define H_MINSUM 16   // min(min_p revealed_sum_p, 127)
#define H_MINHID 17   // min_p hidden_count_p
#define H_HIST 18     // u8[15] bincount of values -2..12 (updated with dword LDS atomics)
#define H_FINISHER 33
#define H_BANK 34      // which of the game's SK_BANK pre-dealt episodes is taken next (mirrors SkParams.bank_head)
#define H_END 36
// the pile buffer follows the header: i8[150], draw pile from one end, discard pile from the other
#define H_PILE H_END
// per-player block: two 16-byte chunks, 16-byte aligned
#define PB_CARDS 0     // i8[12] true cards (-14 once refunded)
#define PB_SUM 12      // i16 sum of the open cards
#define PB_HIDDEN 14   // u8 hidden cards
#define PB_REFUNDED 15 // u8 num_refunded (skyjo.py:57)
#define PB_VIS 16      // i8[12] what an observer sees: card / 15 hidden / -14 refunded
#define PB_PLACED 28   // u16 num_placed (skyjo.py:58)
#define PB_BYTES 32

#define F_TERMINATED 1  // is_terminated (skyjo.py:54)
#define F_DONE 2        // env-level done (natural end or illegal action)
#define F_VALID 4       // game slot in use (padding lanes of the last tile are not)

struct SkLayout {
  int32_t N, indirect, D, Dp, rec_bytes;
  int32_t off_pile, off_players;  // byte offsets: pile buffer, block of player 0 (PB_BYTES each)
  int32_t state_bytes, chunks;    // chunks = state_bytes / 16
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
  L.Dp = (L.D + 3) & ~3;              // D is odd: byte D of a record is free and carries the action
  L.rec_bytes = (L.Dp + 32 + 15) & ~15;
  L.off_pile = H_PILE;
  L.off_players = (H_PILE + SK_NCARDS + 15) & ~15;  // 192
  L.state_bytes = L.off_players + PB_BYTES * N;     // 256 / 288 / 320 for 2 / 3 / 4 players
  L.chunks = L.state_bytes / 16;
  return L;
}
SK_HD static inline int sk_pb(const SkLayout &L, int p) { return L.off_players + PB_BYTES * p; }
