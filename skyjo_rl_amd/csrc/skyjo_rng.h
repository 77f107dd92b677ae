// skyjo_rng.h - part of skyjo_device.h (included from there, in its place: the parts build on each other in that order).
// RNG streams: numpy's legacy MT19937 restated (lazily regenerated, invertible), Philox sessions, rk_interval.
#pragma once
#ifndef SKYJO_DEVICE_PARTS
#error "include skyjo_device.h"
#endif

// ------------------------------------------------------------------------------------------
// RNG.  MT mode restates numpy's legacy RandomState (requirements.txt:3 pins numpy==1.21.5; call
// sites skyjo.py:81,94,101,135): init_genrand, tempering, rk_interval, Fisher-Yates.  The state
// lives in HBM (2496 B per game, touched only when dealing / reshuffling) and is regenerated
// lazily and in place, in stream order, which yields the same stream as the classic 624-word
// block twist (element i only needs old[i], old[i+1] and element i+397 mod 624).
// ------------------------------------------------------------------------------------------
// Regeneration is done 16 elements at a time: all loads of a chunk are issued together, so HBM/L2
// latency is paid once per 16 draws; the tempered outputs wait in a per-lane LDS ring of DEPTH
// words (fp[k << 8]).  The persistent state word of a stream is  idx | ahead << 16 : `idx` in
// [0, 624) is the next position to consume, the `ahead` positions from idx on have already been
// regenerated in memory (their outputs are re-read from there when the next session opens).
//   DEPTH 16: the simple form used by the rare in-kernel paths (each lane refills on its own).
//   DEPTH 64: the dealing kernel; refills happen for the whole wavefront at once (service()), so
//             the ~450-instruction chunk regeneration never runs for a single lane at a time.
#define MT_FIFO(k) (*(uint32_t *)(fp + ((k) << 8)))
__device__ __forceinline__ uint32_t mt_temper(uint32_t v) {
  v ^= v >> 11;
  v ^= (v << 7) & 0x9d2c5680u;
  v ^= (v << 15) & 0xefc60000u;
  v ^= v >> 18;
  return v;
}
// The same with gfx950's three-input bit operation (truth table 0x78: a ^ (b & c), 0x96: a ^ b ^ c, 0xd8: c ? b : a)
__device__ __forceinline__ uint32_t mt_temper3(uint32_t v) {
  v ^= v >> 11;
  v = __builtin_amdgcn_bitop3_b32(v, v << 7, 0x9d2c5680u, 0x78);
  v = __builtin_amdgcn_bitop3_b32(v, v << 15, 0xefc60000u, 0x78);
  v ^= v >> 18;
  return v;
}
__device__ __forceinline__ uint32_t mt_twist3(uint32_t o0, uint32_t o1, uint32_t x) {  // new element from old[i], old[i+1], [i+397]
  const uint32_t y = __builtin_amdgcn_bitop3_b32(o0, o1, 0x7fffffffu, 0xd8);
  const uint32_t mag = (uint32_t)((int32_t)(o1 << 31) >> 31) & 0x9908b0dfu;
  return __builtin_amdgcn_bitop3_b32(x, y >> 1, mag, 0x96);
}
template <int DEPTH>
struct MtStream {
  uint32_t *mt;
  uint8_t *fp;
  Stamps *stp = nullptr;  // diagnostics only
  int idx, gen, rp, wp, pend, used;
  __device__ __forceinline__ static int wrap(int x) { return x >= 624 ? x - 624 : x; }
  __device__ __forceinline__ void open(uint32_t *mt_, int packed, uint8_t *fp_) {
    mt = mt_, fp = fp_;
    idx = packed & 0xffff;
    idx = idx >= 624 ? 0 : idx;
    const int ahead = packed >> 16;
    gen = wrap(idx + ahead), rp = 0, wp = 0, used = 0, pend = 0;
    if (DEPTH >= 64) {
      // the stream position idx + ahead is a multiple of 16: start the ring so that wp stays one too
      rp = (16 - (ahead & 15)) & 15;
      for (int j0 = 0; j0 < ahead; j0 += 16) {  // 16 independent loads per round trip
        uint32_t t[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = j0 + k < ahead ? mt[wrap(idx + j0 + k)] : 0u;
#pragma unroll
        for (int k = 0; k < 16; k++)
          if (j0 + k < ahead) MT_FIFO(rp + j0 + k) = mt_temper(t[k]);
      }
      wp = rp + ahead;
    } else {
      pend = ahead;
    }
  }
  __device__ __forceinline__ int close() const { return wrap(idx + used) | ((pend + wp - rp) << 16); }
  // Regenerating elements gen .. gen+15 in place is split in two: refill_issue() starts the 33 loads,
  // refill_finish() twists, tempers and stores.  The dealing kernel calls them a few shuffle iterations
  // apart (service()), so the memory latency of a chunk hides behind the LDS work of the shuffle.
  uint32_t o[17], x[16];
  bool issued = false;
  __device__ __forceinline__ void refill_issue() {
    const int c = gen;  // multiple of 16, so &mt[c] is 64-byte aligned
    const uint4 *po = (const uint4 *)(mt + c);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint4 q = po[k];
      o[4 * k] = q.x, o[4 * k + 1] = q.y, o[4 * k + 2] = q.z, o[4 * k + 3] = q.w;
    }
    o[16] = mt[c + 16 == 624 ? 0 : c + 16];
    if (c != 224) {  // elements i + 397 (mod 624) are contiguous for the whole chunk: four 16-byte loads
      typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // only dword-aligned
      const u32x4_a4 *px = (const u32x4_a4 *)(mt + (c < 224 ? c + 397 : c - 227));
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32x4_a4 q = px[k];
        x[4 * k] = q.x, x[4 * k + 1] = q.y, x[4 * k + 2] = q.z, x[4 * k + 3] = q.w;
      }
    } else {  // the one chunk that straddles the wrap (i = 224..226 -> 621..623, i = 227.. -> 0..)
#pragma unroll
      for (int k = 0; k < 16; k++) x[k] = mt[k < 3 ? 621 + k : k - 3];
    }
    issued = true;
  }
  __device__ __forceinline__ void refill_finish() {
    const int c = gen;
    uint32_t v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      uint32_t y = (o[k] & 0x80000000u) | (o[k + 1] & 0x7fffffffu);
      v[k] = x[k] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    uint4 *pm = (uint4 *)(mt + c);
#pragma unroll
    for (int k = 0; k < 4; k++) pm[k] = make_uint4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    const int w0 = wp & (DEPTH - 1);  // wp is a multiple of 16 whenever a chunk is appended
#pragma unroll
    for (int k = 0; k < 16; k++) MT_FIFO(w0 + k) = mt_temper(v[k]);
    wp += 16;
    gen = c + 16 == 624 ? 0 : c + 16;
    issued = false;
  }
  __device__ __forceinline__ void refill() {
    if (!issued) refill_issue();
    refill_finish();
  }
  // Called once per 16 draws of the dealing kernel's shuffle loop, where the active lanes are converged and
  // have each consumed the same number of draws.  Steady state: every lane has room for a chunk, so the chunk
  // whose loads were started one call earlier is finished (twist, temper, store) and the loads of the next one
  // are started - their latency hides behind the next 16 draws.  A lane left with fewer than `need` draws
  // (it joined with an emptier ring) is served on its own.  One code copy of issue / finish serves both cases.
  __device__ __forceinline__ void service(int need = 0) {
    if (DEPTH < 64) return;
#ifdef SK_STAMPS
    Stamps &st = *stp;
    STAMP(3);
#endif
#pragma unroll 1
    for (int pass = 0; pass < 4; pass++) {
      const bool room = DEPTH - (wp - rp) >= 16;
      const bool dry = wp - rp < need;
      const bool all_room = pass == 0 && __all(room);
      if (!all_room && !__any(dry)) break;
      if (issued && (dry || all_room)) refill_finish();
      if (!issued && ((all_room && DEPTH - (wp - rp) >= 16) || wp - rp < need)) refill_issue();
    }
#ifdef SK_STAMPS
    STAMP(5);
#endif
  }
  __device__ __forceinline__ void unget() {  // give back the draw returned by the last next()
    used--;
    if (DEPTH < 64 && rp == 0 && wp == 0) pend++;  // still in the read-from-memory phase of open()
    else rp--;
  }
  __device__ __forceinline__ uint32_t next() {
    uint32_t v;
    if (DEPTH < 64 && pend > 0) {
      v = mt_temper(mt[wrap(idx + used)]);
      pend--;
    } else {
      if (rp == wp) refill();
      v = MT_FIFO(rp & (DEPTH - 1));
      rp++;
    }
    used++;
    return v;
  }
};

// Philox "session": ctr = (block, episode, reshuffle index, domain), key = seed + 1.
struct PhiloxStream {
  uint32_t k0, k1, blk, c1, c2, c3, b0, b1, b2, b3;
  int pos;
  __device__ __forceinline__ void open(uint64_t key, uint32_t episode, uint32_t resh, uint32_t domain) {
    k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32), blk = 0, c1 = episode, c2 = resh, c3 = domain, pos = 4;
  }
  __device__ __forceinline__ void service() {}
  __device__ __forceinline__ void unget() { pos--; }  // pos >= 1 after any next()
  __device__ __forceinline__ uint32_t next() {
    if (pos >= 4) {
      philox4x32_10(blk, c1, c2, c3, k0, k1, b0, b1, b2, b3);
      blk++, pos = 0;
    }
    uint32_t v = pos == 0 ? b0 : pos == 1 ? b1 : pos == 2 ? b2 : b3;
    pos++;
    return v;
  }
};

// legacy rk_interval (32-bit path): smallest all-ones mask >= max, rejection sampling
template <class Rng>
__device__ __forceinline__ uint32_t rng_interval(Rng &r, uint32_t max) {
  uint32_t mask = 0xffffffffu >> __clz((int)(max | 1u));
  uint32_t v;
  do v = r.next() & mask;
  while (v > max);
  return v;
}

// pile addressing: region A grows up from byte 0, region B grows down from byte 149.
// role 0: draw pile in A, discard pile in B; a mid-game reshuffle flips the role.
__device__ __forceinline__ int pile_addr(int region_b, int k) { return region_b ? (SK_NCARDS - 1 - k) : k; }
