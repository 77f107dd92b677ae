// skyjo_record.h - part of skyjo_device.h (included from there, in its place: the parts build on each other in that order).
// The output record (collect_observation + action mask) and the bank of pre-dealt episodes.
#pragma once
#ifndef SKYJO_DEVICE_PARTS
#error "include skyjo_device.h"
#endif

// ------------------------------------------------------------------------------------------
// collect_observation (skyjo.py:148-199) + action mask (skyjo.py:201-224) -> output record.
// obs[0..16] are a straight copy of state bytes 16..32, obs[17] / obs[18] come from the header
// registers; the card part is the observer's `vis` row (indirect) or all rows in absolute seat
// order (direct, skyjo.py:279-302).
// ------------------------------------------------------------------------------------------
template <bool INDIRECT>
__device__ __forceinline__ void emit_record(const SkParams &P, uint8_t *lp, const HdrRegs &h, const ObsRegs &ob, int action,
                                            uint8_t *out, uint4 *held = nullptr, const uint4 *pre_a = nullptr, uint32_t pre_b = 0) {
  const int phase = h.w0 & 0xff;
  const uint32_t q0 = ob.q0, q1 = ob.q1, q2 = ob.q2;
  const uint32_t act24 = ((uint32_t)action & 0xffu) << 24;  // byte D of the record: the action this step applied (-1: none)
  uint32_t m[8];
  {
    // (computed in both phases and masked: a branch on the phase measured 2 us slower per launch)
    const uint32_t pm = phase ? 0xffffffffu : 0u;
    m[0] = ob.nz0 & pm, m[1] = ob.nz1 & pm, m[2] = ob.nz2 & pm, m[3] = ob.hd0 & pm, m[4] = ob.hd1 & pm, m[5] = ob.hd2 & pm;
    m[6] = (phase ? 0u : 0x0101u) | (((h.w0 >> 8) & 0xffu) << 16) | ((uint32_t)phase << 24);
    m[7] = (((h.w0 >> 16) & F_DONE) ? 1u : 0u) | ((h.w0 >> 24) << 8) | ((h.w2 & 0xffffu) << 16);
  }
  // obs[0..15] are chunk 1 of the record; obs[16] = hist[14], obs[17] = discard top, obs[18] = hand card
  const uint4 a = pre_a ? *pre_a : LQ(1);  // (pre_*: the caller has requested them together with the row)
  const uint32_t s8 = (pre_a ? pre_b : (uint32_t)LB(32)) | ((h.w1 >> 24) << 8) | ((h.w2 >> 24) << 16);
  if (INDIRECT) {
    uint4 *o = (uint4 *)out;
    uint4 b;
    b.x = s8 | (q0 << 24), b.y = (q0 >> 8) | (q1 << 24), b.z = (q1 >> 8) | (q2 << 24), b.w = (q2 >> 8) | act24;
    if (held) {  // the caller stores the record itself
      held[0] = a, held[1] = b, held[2] = make_uint4(m[0], m[1], m[2], m[3]), held[3] = make_uint4(m[4], m[5], m[6], m[7]);
    } else {
      o[0] = a, o[1] = b;
      o[2] = make_uint4(m[0], m[1], m[2], m[3]);
      o[3] = make_uint4(m[4], m[5], m[6], m[7]);
    }
  } else {
    // direct observation (skyjo.py:279-302): every player's visible row in absolute seat order, 12 bytes each, packed
    // behind obs[18]; a row is one chunk read
    uint32_t *o = (uint32_t *)out;
    const int N = P.L.N;
    o[0] = a.x, o[1] = a.y, o[2] = a.z, o[3] = a.w;
    uint32_t carry = s8;  // three bytes waiting for the next word's top byte
    for (int p = 0; p < N; p++) {
      const uint4 r = LQ((sk_pb(P.L, p) + PB_VIS) >> 4);
      o[4 + 3 * p] = carry | (r.x << 24);
      o[5 + 3 * p] = (r.x >> 8) | (r.y << 24);
      o[6 + 3 * p] = (r.y >> 8) | (r.z << 24);
      carry = r.z >> 8;
    }
    o[4 + 3 * N] = carry | act24;
    uint32_t *om = o + (P.L.Dp >> 2);
#pragma unroll
    for (int w = 0; w < 8; w++) om[w] = m[w];
  }
}

// ------------------------------------------------------------------------------------------
// Take the pre-dealt next episode (SkyjoGame.reset, skyjo.py:52-74; the dealing itself is k_deal).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t bank_rec16(const SkParams &P, int slot, int g) {  // first 16-byte piece of the bank record (slot, game)
  return ((size_t)slot * P.tiles * SK_TILE + (size_t)g) * P.L.chunks;
}
__device__ __forceinline__ void load_spare(const SkParams &P, uint8_t *lp, int slot, int tile, int lane) {
  const uint4 *s = P.spare + bank_rec16(P, slot, tile * SK_TILE + lane);
  const int n = P.L.chunks;
  for (int c = 0; c < n; c += 6) {  // (groups of six as in tile_load: one memory round trip per group)
    uint4 v[6];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (c + k < n) v[k] = s[c + k];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (c + k < n) LQ(c + k) = v[k];
  }
}
__device__ __forceinline__ void store_spare(const SkParams &P, uint8_t *lp, int slot, int g) {  // the lane's record in LDS -> its bank slot
  uint4 *d = P.spare + bank_rec16(P, slot, g);
  const int n = P.L.chunks;
  for (int c = 0; c < n; c += 6) {
    uint4 v[6];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (c + k < n) v[k] = LQ(c + k);
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (c + k < n) d[c + k] = v[k];
  }
}

__device__ __forceinline__ void bank_advance(const SkParams &P, uint8_t *lp, int g, int head, uint32_t dc) {
  const uint8_t nh = (uint8_t)((head + 1) % SK_BANK);
  P.bank_head[g] = nh;
  LB(H_BANK) = nh;
  P.deals_consumed[g] = dc + 1;
  P.done[g] = 0;
}

// k_reset / generic form: plain loads, the record passes through registers.  Returns false (slot untouched) when the
// bank is empty.
__device__ __forceinline__ bool consume_spare(const SkParams &P, uint8_t *lp, int tile, int lane, int g, int head) {
  const size_t G = (size_t)P.tiles * SK_TILE;
  const uint8_t ready = P.spare_ready[(size_t)head * G + g];
  const uint32_t dc = P.deals_consumed[g];
  if (!ready) return false;
  load_spare(P, lp, head, tile, lane);
  P.spare_ready[(size_t)head * G + g] = 0;  // k_scan finds the banks that are not full
  bank_advance(P, lp, g, head, dc);
  return true;
}

// The step kernel's form, in two halves.  `spare_issue` asks for the whole record by LDS-DMA straight into the lane's
// own (dead: its game is over) slot of the tile - no registers, no LDS writes - and for the two words of bookkeeping;
// `spare_commit` waits for everything this wavefront has in flight and finishes the hand-over.  Between the two the
// wavefront steps its live games, which hides the memory round trip of the few lanes that are resetting (about every
// second iteration has one).  If the bank turns out to be empty the slot holds a stale record: the caller deals in
// place, which rewrites every word of it.  (Measured and dropped in round 3, EXPERIMENTS.md: the bookkeeping words kept in
// registers for the whole launch - same time; the record requested a whole iteration earlier, when the game ends - slower.)
struct SpareRegs {
  uint32_t dc;
  int head;
  uint8_t ready;
};
__device__ __forceinline__ void spare_issue(const SkParams &P, uint8_t *lp, uint32_t lds_tile, int tile, int lane, int g, SpareRegs &r) {
  const size_t G = (size_t)P.tiles * SK_TILE;
  r.head = LB(H_BANK) % SK_BANK;  // (read before the record is overwritten)
  r.ready = P.spare_ready[(size_t)r.head * G + g];
  r.dc = P.deals_consumed[g];
  const uint32_t voff = (uint32_t)(bank_rec16(P, r.head, g) * 16);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every LDS read of the old record has returned
  dma_bank_record<false>((const uint8_t *)P.spare, voff, lds_tile, P.L.chunks);
}
__device__ __forceinline__ bool spare_commit(const SkParams &P, uint8_t *lp, int g, const SpareRegs &r) {
  sk_vm_drain();
  if (!r.ready) return false;
  P.spare_ready[(size_t)r.head * (size_t)P.tiles * SK_TILE + g] = 0;  // the dealing run finds the banks that are not full
  bank_advance(P, lp, g, r.head, r.dc);
  return true;
}

// LDS stride of one staged record: the record's own size, plus 16 bytes when that is a multiple of 32 dwords / 4 - the
// lane-per-record dword writes then spread over 8 banks groups instead of 4.
__device__ __forceinline__ constexpr int sk_stage_stride(int rec_bytes) { return ((rec_bytes >> 2) & 7) == 0 ? rec_bytes + 16 : rec_bytes; }

// Fallback when the pre-dealt episode is not available inside a launch (a mid-game reshuffle just
// invalidated it, or the game already took one in this launch): deal right here, on this lane, from
// the game's current stream position.  Rare and slow (one lane active), never changes results.
// When the game's stream cannot be had (wait_deal_done timed out: sticky device error) stream and bank stay untouched and
// the slot is left as a finished game: it asks again in the next iteration.
// Returns false in that case: the caller must not present the slot as a freshly re-dealt game (status stays ERROR, no reset counted).
__device__ __forceinline__ bool deal_inline(const SkParams &P, uint8_t *lp, uint8_t *fp, int g, int tile, int lane, int head);
