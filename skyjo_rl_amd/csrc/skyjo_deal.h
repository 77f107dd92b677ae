// skyjo_deal.h - part of skyjo_device.h (included from there, in its place: the parts build on each other in that order).
// K_seed and the dealing run: deck walk, compact deal, in-place deal, k_scan / k_publish, deal_body, k_deal.
#pragma once
#ifndef SKYJO_DEVICE_PARTS
#error "include skyjo_device.h"
#endif

// ------------------------------------------------------------------------------------------
// k_seed: np.random.seed(value + 1) per game (skyjo.py:84-94; legacy init_genrand)
// ------------------------------------------------------------------------------------------
__global__ void k_seed(SkParams P, const uint64_t *seeds, uint64_t base, int first, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int g = first + i;
  const uint64_t value = seeds ? seeds[i] : base + P.game_id0 + (uint64_t)g;
  const size_t G = (size_t)P.tiles * SK_TILE;
  P.seeds[g] = value;
  P.deals_consumed[g] = 0;
  for (int k = 0; k < SK_BANK; k++) P.spare_ready[(size_t)k * G + g] = 0;
  P.bank_head[g] = 0, P.busy[g] = 0, P.cancel[g] = 0, P.done_flag[g] = 0;
  if (P.rng_mode == SKYJO_RNG_MT19937) {
    uint32_t *mt = P.mt + (size_t)g * 624;
    uint32_t x = (uint32_t)(value + 1);
    mt[0] = x;
    for (int k = 1; k < 624; k++) {
      x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)k;
      mt[k] = x;
    }
    P.mt_idx[g] = 0;
  }
}

// np.random.seed(value) on the CURRENT stream of one game, no +1, no deal (fixture injection)
__global__ void k_seed_raw(SkParams P, int g, uint32_t value) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t *mt = P.mt + (size_t)g * 624;
  uint32_t x = value;
  mt[0] = x;
  for (int k = 1; k < 624; k++) {
    x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)k;
    mt[k] = x;
  }
  P.mt_idx[g] = 0;
  const size_t G = (size_t)P.tiles * SK_TILE;
  for (int k = 0; k < SK_BANK; k++) P.spare_ready[(size_t)k * G + g] = 0;  // pre-dealt from the old stream
}

// ------------------------------------------------------------------------------------------
// k_deal: SkyjoGame.reset's dealing (skyjo.py:52-74 with :76-82, :96-103, :105-125, :127-138) for
// the games whose spare record is empty, one lane per game, written to the game's SPARE record.
// RNG order per deal: shuffle(150) -> shuffle(150-12N) -> N x permutation(12)[:2].
// ------------------------------------------------------------------------------------------
// Legacy RandomState.shuffle: for i = n-1 .. 1: j = rk_interval(i); swap(a[i], a[j]).  Every lane walks
// its own i, so a loop iteration is one draw for every lane (a rejected draw just does not advance i):
// the trip count is the largest per-lane draw total, not the sum over i of the unluckiest lane's
// rejections.  The draw for the next iteration is fetched before the swap to overlap the LDS round trips.
template <class Rng>
__device__ __forceinline__ void shuffle_lds(uint8_t *lp, int base, int n, Rng &r) {
  int i = n - 1;
  if (i < 1) return;
  uint32_t mask = 0xffffffffu >> __clz(i);
  r.service();
  uint32_t pre = r.next();
  while (i >= 1) {
    const uint32_t v = pre & mask;
    r.service();
    pre = r.next();
    if (v <= (uint32_t)i) {
      const uint8_t t = LB(base + i);
      LB(base + i) = LB(base + (int)v), LB(base + (int)v) = t;
      i--;
      mask = 0xffffffffu >> __clz(i | 1);
    }
  }
  r.unget();  // the prefetched draw belongs to whoever consumes the stream next
}

// Dealing-kernel form: inside the loop every lane consumes exactly one draw per step, so the ring is serviced
// for the whole wavefront once per 16 draws and the lanes fetch their next four draws together (one LDS round
// trip per four draws, no selects: the four steps are unrolled).
#define SK_SHUFFLE_STEP(q)                                                                        \
  if (i >= 1) {                                                                                   \
    const uint32_t v = (q) & mask;                                                                \
    r.rp++, r.used++;                                                                             \
    if (v <= (uint32_t)i) {                                                                       \
      const int bv = base + (int)v, bi = base + i;                                                \
      const uint8_t av = LB(bv);                                                                  \
      LB(bi) = av, LB(bv) = (uint8_t)ai;                                                          \
      i--;                                                                                        \
      mask = 0xffffffffu >> __clz(i | 1);                                                         \
      ai = LB(base + i);                                                                          \
    }                                                                                             \
  }
__device__ __forceinline__ void shuffle_lds(uint8_t *lp, int base, int n, MtStream<64> &r) {
  int i = n - 1;
  if (i < 1) return;
  uint8_t *fp = r.fp;
  uint32_t mask = 0xffffffffu >> __clz(i);
  int ai = LB(base + i);  // a[i] is read one step ahead (after a swap it is read behind the two writes)
  while (__any(i >= 1)) {
    r.service(16);
#pragma unroll 1
    for (int grp = 0; grp < 4; grp++) {
      const uint32_t q0 = MT_FIFO(r.rp & 63), q1 = MT_FIFO((r.rp + 1) & 63), q2 = MT_FIFO((r.rp + 2) & 63),
                     q3 = MT_FIFO((r.rp + 3) & 63);
      SK_SHUFFLE_STEP(q0)
      SK_SHUFFLE_STEP(q1)
      SK_SHUFFLE_STEP(q2)
      SK_SHUFFLE_STEP(q3)
    }
  }
}

// ------------------------------------------------------------------------------------------
// Dealing kernel, fixed player count.  Everything that decides WHICH cards are swapped (the draw, its mask,
// the rejection test, the shuffle index) is register arithmetic on the RNG outputs alone; the deck only ever
// receives the swaps.  So the loop takes its draws four at a time: the eight deck words the four steps touch
// are requested together, the few ways two of those steps can meet on the same position are resolved with
// selects in registers, and the eight results are written back fire-and-forget (LDS executes a wavefront's
// accesses in order, the next batch's reads see them).  One LDS round trip per four draws, no branch inside.
//
// RNG side: a "chunk stream" hands out 16 outputs at a time in registers R[0..15].
//   MtChunkStream      numpy-legacy MT19937, regenerated in place 16 elements at a time; the 33 loads of the next
//                      chunk are in flight during the 16 draws of the current one.  A lane may enter in the middle
//                      of a chunk (leftover outputs of its previous session) - the prologue of each loop skips
//                      the outputs before its position - and from then on every lane is chunk-aligned.
//   PhiloxChunkStream  the counter-based session of PhiloxStream, four blocks per refill.
// ------------------------------------------------------------------------------------------
#define SK_STG_STRIDE 80  // 64 bytes of chunk + 4 of address + pad: an odd number of 16-byte units, rows spread over the banks
struct MtChunkStream {
  uint32_t *mt;
  int base, pos, gen, chunks_made;  // outputs R[pos..15] of chunk `base` are unconsumed; next chunk starts at gen
  uint32_t R[16], o[17], x[16], xw;
  static constexpr bool kLockstep = true;  // the refill loops are run by the whole wavefront (see refill)
  uint32_t *mt0;     // all states; my_off = word offset of this lane's state in it
  uint32_t my_off;
  uint8_t *stg;      // the wavefront's staging rows in LDS (SK_STG_STRIDE bytes per lane) for the cooperative store
  int lane;
  __device__ __forceinline__ static int wrap(int v) { return v >= 624 ? v - 624 : v; }
  __device__ __forceinline__ void open(uint32_t *mt0_, uint32_t my_off_, int packed, uint8_t *stg_, int lane_) {
    mt0 = mt0_, my_off = my_off_, mt = mt0_ + my_off_, stg = stg_, lane = lane_, chunks_made = 0;
    int idx = packed & 0xffff;
    idx = idx >= 624 ? 0 : idx;
    const int ahead = packed >> 16;  // <= 16: outputs idx .. idx+ahead-1 are already regenerated in memory
    gen = wrap(idx + ahead);  // always a multiple of 16
    base = gen == 0 ? 608 : gen - 16;
    pos = 16 - ahead;  // 16: nothing pending, the first block starts with a refill
    if (ahead > 0) {
#pragma unroll
      for (int k = 0; k < 16; k++) R[k] = mt_temper(mt[base + k]);
    }
  }
  __device__ __forceinline__ int close() const { return wrap(base + pos) | (((16 - pos) & 31) << 16); }
  __device__ __forceinline__ void issue() {
    const int c = gen;
    const uint4 *po = (const uint4 *)(mt + c);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint4 q = po[k];
      o[4 * k] = q.x, o[4 * k + 1] = q.y, o[4 * k + 2] = q.z, o[4 * k + 3] = q.w;
    }
    o[16] = mt[c + 16 == 624 ? 0 : c + 16];
    // The partners i + 397 (mod 624), without a branch (with the two cases in two exec-masked blocks the compiler merges their
    // registers behind each block and waits for the loads right there, a few instructions after they were issued).  Chunk 224 is the one whose partners wrap (621, 622, 623,
    // 0 .. 12): its pieces 1 .. 3 are words 1 .. 12 = (224 - 227) + 4 k like every chunk above it, piece 0 is read at 621
    // (three partners and one word beyond the state: the allocation has the slack) and word 0 comes with a one-word
    // load that every other chunk points at a line it is requesting anyway.
    typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    const uint32_t *pb = mt + (c < 224 ? c + 397 : c - 227);
    const u32x4_a4 *p0 = (const u32x4_a4 *)(c == 224 ? mt + 621 : pb);
    {
      const u32x4_a4 q = *p0;
      x[0] = q.x, x[1] = q.y, x[2] = q.z, x[3] = q.w;
    }
#pragma unroll
    for (int k = 1; k < 4; k++) {
      const u32x4_a4 q = ((const u32x4_a4 *)pb)[k];
      x[4 * k] = q.x, x[4 * k + 1] = q.y, x[4 * k + 2] = q.z, x[4 * k + 3] = q.w;
    }
    xw = mt[c == 224 ? 0 : c];
  }
  __device__ __forceinline__ void pre_loop() { issue(); }  // the first chunk's loads (once per deal, every lane)
  // Regenerate chunk `gen` in place (its loads were started a chunk earlier), write it back and start the loads of
  // the chunk after it.
  //
  // The dealing kernel is bound by the memory system (5.3 TB/s of 128-byte line reads and 64-byte write-backs at the
  // fabric, EXPERIMENTS.md), and most sensitive to how the state is WRITTEN: stored by its owner, a chunk is four
  // 16-byte pieces in four instructions, each of which scatters 64 pieces over 64 lines.  So the wavefront writes
  // TOGETHER: every lane puts its chunk and its address into its staging row in LDS, and store instruction k is lane i
  // writing piece i & 3 of the lane 16 k + (i >> 2) - four neighbouring lanes one whole 64-byte line, a quarter of the
  // write requests.  That takes all 64 lanes: the refill loops are run by the whole wavefront until its last lane is
  // through, and a lane that is (`live` false) keeps its stream where it is and stores nothing.  The stores go out
  // BEFORE the next chunk's loads (the other order: 81 instead of 76 us per run).
  __device__ __forceinline__ void refill(const bool live) {
    const int c = gen;
    uint32_t v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = mt_twist3(o[k], o[k + 1], k == 3 && c == 224 ? xw : x[k]);
    base = live ? c : base, pos = live ? 0 : pos, chunks_made += live ? 1 : 0;
    gen = live ? (c + 16 == 624 ? 0 : c + 16) : c;
    const uint32_t lm = live ? 0xffffffffu : 0u;  // (a select the compiler cannot turn into a branch around the tempering)
#pragma unroll
    for (int k = 0; k < 16; k++) R[k] = __builtin_amdgcn_bitop3_b32(R[k], mt_temper3(v[k]), lm, 0xd8);
    uint8_t *row = stg + lane * SK_STG_STRIDE;
#pragma unroll
    for (int k = 0; k < 4; k++) ((uint4 *)row)[k] = make_uint4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    *(uint32_t *)(row + 64) = live ? my_off + (uint32_t)c : 0xffffffffu;
    __builtin_amdgcn_wave_barrier();  // (LDS runs a wavefront's accesses in order: no wait, only no reordering by the compiler)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint8_t *orow = stg + (16 * k + (lane >> 2)) * SK_STG_STRIDE;
      const uint4 q = *(const uint4 *)(orow + (lane & 3) * 16);
      const uint32_t off = *(const uint32_t *)(orow + 64);
      if (off != 0xffffffffu) *(uint4 *)(mt0 + off + 4 * (lane & 3)) = q;
    }
    __builtin_amdgcn_wave_barrier();
    issue();  // (a lane that is through asks for the same chunk again)
  }
};

struct PhiloxChunkStream {  // same output sequence as PhiloxStream (block b -> words 4b .. 4b+3)
  uint32_t R[16], k0, k1, blk, c1, c2, c3;
  int pos;
  __device__ __forceinline__ void open(uint64_t key, uint32_t episode, uint32_t resh, uint32_t domain) {
    k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32), blk = 0, c1 = episode, c2 = resh, c3 = domain, pos = 16;
  }
  static constexpr bool kLockstep = false;
  __device__ __forceinline__ void pre_loop() {}
  __device__ __forceinline__ void refill(bool) {
#pragma unroll
    for (int b = 0; b < 4; b++) philox4x32_10(blk + b, c1, c2, c3, k0, k1, R[4 * b], R[4 * b + 1], R[4 * b + 2], R[4 * b + 3]);
    blk += 4, pos = 0;
  }
};

// The lane's deck: card k is the BYTE at LDS address  dk + k,  dk = lane * SK_DECK_STRIDE  (a position is still an
// address: one add).  A stride of 39 dwords spreads the lanes' equal positions over all banks; data-dependent
// positions of different lanes collide two- or three-way now and then, which the LDS unit absorbs - what the 9.75 KB
// per wavefront buy (37.5 KB with one card per dword) is room for the dealing wavefronts BESIDE the step kernel's
// four per CU, so that a dealing run can hide behind the step launches that follow it.
#define SK_DECK_STRIDE 156
#define DK_AT(addr) (*((uint8_t *)lds_raw_base + (addr)))
#define DK_AT32(addr) (*(uint32_t *)((uint8_t *)lds_raw_base + (addr)))

// Legacy RandomState.shuffle (for i = n-1 .. 1: j = rk_interval(i); swap(a[i], a[j])) of the whole deck and then
// of the rest behind the 12 NP dealt cards (skyjo.py:76-82 and :68-70,:127-138), as ONE lane-private walk: a lane
// that accepts the last draw of the first shuffle starts the second with its very next draw, so lanes only
// re-converge once, at the end.  A draw that is not used for a swap (rejected, lane not there yet, lane finished)
// swaps the current position with itself.
struct DeckWalk {
  uint32_t pcur, pb;  // LDS addresses of a[i] and a[0] of the current shuffle
  uint32_t n;         // i + 1: the draw v is accepted iff v < n; 0 when the lane has finished both shuffles
  uint32_t nxt_n;     // what n becomes when the current shuffle completes (R, then 0)
  uint32_t mask;      // rk_interval's mask for max = n - 1
};
#ifndef SK_DECK_BS
#define SK_DECK_BS 2  // draws per batch (2: 96.6 us, 4: 99.9 us, 8: 102.2 us per dealing run at the headline size)
#endif
template <bool PRO, int NP, class Rng>
__device__ __forceinline__ void deck_batch(Rng &r, const int s, uint32_t *lds_raw_base, const uint32_t dk, DeckWalk &w) {
  constexpr uint32_t R = SK_NCARDS - 12 * NP;
  constexpr int BS = SK_DECK_BS;
  uint32_t pI[BS], pJ[BS];
#ifdef SK_EXP_DECK_FAST
  // round-6 experiment: while no lane of the wavefront can complete a shuffle inside this batch - every n is
  // either 0 (done) or beyond BS + 1 - the five instructions per draw that handle a completion are left out
  if (__all(w.n == 0u || w.n > (uint32_t)(BS + 1))) {
#pragma unroll
    for (int k = 0; k < BS; k++) {
      const uint32_t v = r.R[s + k] & w.mask;
      const bool acc = PRO ? (v < w.n && s + k >= r.pos) : (v < w.n);
      pI[k] = w.pcur;
      pJ[k] = acc ? w.pb + v : w.pcur;
      const uint32_t d = acc ? 0xffffffffu : 0u;
      w.n += d;
      w.pcur += d;
      w.mask = 0xffffffffu >> __builtin_clz(w.n - 1u);
    }
  } else
#endif
  {
#pragma unroll
  for (int k = 0; k < BS; k++) {
    const uint32_t v = r.R[s + k] & w.mask;
    const bool acc = PRO ? (v < w.n && s + k >= r.pos) : (v < w.n);
    pI[k] = w.pcur;
    pJ[k] = acc ? w.pb + v : w.pcur;
    const uint32_t d = acc ? 0xffffffffu : 0u;
    const uint32_t n2 = w.n + d;
    const bool t = n2 == 1u;  // this shuffle is complete (i reached 0): on to the rest, or done
    w.n = t ? w.nxt_n : n2;
    w.pcur = t ? dk + (SK_NCARDS - 1) : w.pcur + d;
    w.pb = t ? dk + 12 * NP : w.pb;
    r.pos = t ? s + k + 1 : r.pos;  // (a lane that is still shuffling after the block gets pos = 16 from the caller)
    w.mask = 0xffffffffu >> __builtin_clz(w.n - 1u);  // (n - 1 is never 0)
  }
  w.nxt_n = w.pb == dk ? R : 0u;  // (a batch never holds two completions: the rest takes > 100 draws)
  }
  uint32_t cI[BS], cJ[BS];
#pragma unroll
  for (int k = 0; k < BS; k++) cI[k] = DK_AT(pI[k]), cJ[k] = DK_AT(pJ[k]);
  // What step k finds at its two positions is what the batch's earlier steps left there.  Only an earlier step's
  // j-position can be met again: its i-position lies above everything that follows (or, for an unused draw, is
  // its j-position).  The latest writer wins, hence ascending m.
#pragma unroll
  for (int k = 1; k < BS; k++)
#pragma unroll
    for (int m = 0; m < k; m++) {
      cI[k] = pJ[m] == pI[k] ? cI[m] : cI[k];
      cJ[k] = pJ[m] == pJ[k] ? cI[m] : cJ[k];
    }
#pragma unroll
  for (int k = 0; k < BS; k++) DK_AT(pI[k]) = (uint8_t)cJ[k], DK_AT(pJ[k]) = (uint8_t)cI[k];
}

// _reset_card_mask (skyjo.py:96-103): choice(12, 2, replace=False) == permutation(12)[:2] per player, i.e. a full
// 11-step shuffle of arange(12) each.  The permutation is twelve nibbles of one 64-bit register, no memory.
struct PermWalk {
  uint64_t pm;
  uint32_t n, mask;   // as in DeckWalk; n == 0: all players done
  uint32_t sh, open;  // open: byte p = slot0 | slot1 << 4 of player p; sh = 8 p
};
template <bool PRO, int NP, class Rng>
__device__ __forceinline__ void perm_batch(Rng &r, const int s, PermWalk &w) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t v = r.R[s + k] & w.mask;
    const bool acc = PRO ? (v < w.n && s + k >= r.pos) : (v < w.n);
    const uint32_t si = 4u * (w.n - 1u) & 63u, sv = 4u * v & 63u;
    uint64_t x = ((w.pm >> si) ^ (w.pm >> sv)) & 0xfull;
    x = acc ? x : 0ull;
    w.pm ^= (x << si) ^ (x << sv);
    const uint32_t n2 = w.n + (acc ? 0xffffffffu : 0u);
    const bool t = n2 == 1u;
    w.open |= t ? ((uint32_t)w.pm & 0xffu) << w.sh : 0u;
    w.sh += t ? 8u : 0u;
    w.pm = t ? 0xBA9876543210ull : w.pm;
    w.n = t ? (w.sh == 8u * NP ? 0u : 12u) : n2;
    r.pos = t ? s + k + 1 : r.pos;
    w.mask = 0xffffffffu >> __builtin_clz(w.n - 1u);  // (n - 1 is never 0)
  }
}

#define SK_FOUR_BATCHES(CALL) CALL(0) CALL(4) CALL(8) CALL(12)
#if SK_DECK_BS == 4
#define SK_DECK_BATCHES(CALL) CALL(0) CALL(4) CALL(8) CALL(12)
#elif SK_DECK_BS == 2
#define SK_DECK_BATCHES(CALL) CALL(0) CALL(2) CALL(4) CALL(6) CALL(8) CALL(10) CALL(12) CALL(14)
#elif SK_DECK_BS == 8
#define SK_DECK_BATCHES(CALL) CALL(0) CALL(8)
#endif

// Compact deal (fixed player count NP): only the deck lives in LDS; the game record is assembled in registers.
// RNG order: shuffle(150) -> shuffle(rest) -> NP x permutation(12)[:2].
template <int NP, class Rng>
__device__ __forceinline__ void deal_compact(const SkParams &P, uint32_t *lds_raw_base, const int lane, Rng &r, uint32_t episode,
                                             uint4 *dst, const bool act) {
  constexpr int R = SK_NCARDS - 12 * NP;
  const SkLayout L = sk_make_layout(NP, P.L.indirect);
  const uint32_t dk = (uint32_t)lane * SK_DECK_STRIDE;  // LDS address of the lane's card 0
#define DKW(k) ((uint32_t)DK_AT(dk + (k)))
#pragma unroll
  for (int d = 0; d < (SK_NCARDS + 3) / 4; d++) {  // _new_drawpile (skyjo.py:76-82), four cards per write
    uint32_t w4 = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (4 * d + j < SK_NCARDS) w4 |= (uint32_t)((-2 + (4 * d + j) / 10) & 0xff) << (8 * j);
    DK_AT32(dk + 4 * d) = w4;
  }
  // A lockstep stream (MtChunkStream) has every lane of the wavefront in the refill loops, also the lanes without a deal
  // (act false: they walk nothing, n = 0 from the start) and the lanes that are through: see MtChunkStream::refill.
#define SK_WALKING(n) (Rng::kLockstep ? __any((n) != 0u) : (n) != 0u)
  {
    DeckWalk w;
    w.pb = dk, w.pcur = dk + (SK_NCARDS - 1), w.n = act ? SK_NCARDS : 0u, w.nxt_n = R, w.mask = 0xffu;
    if (r.pos < 16) {
#define SK_CALL(s) deck_batch<true, NP>(r, s, lds_raw_base, dk, w);
      SK_DECK_BATCHES(SK_CALL)
#undef SK_CALL
      r.pos = w.n ? 16 : r.pos;
    }
    if (SK_WALKING(w.n)) r.pre_loop();
    while (SK_WALKING(w.n)) {
      r.refill(w.n != 0u);
#define SK_CALL(s) deck_batch<false, NP>(r, s, lds_raw_base, dk, w);
      SK_DECK_BATCHES(SK_CALL)
#undef SK_CALL
      r.pos = w.n ? 16 : r.pos;
    }
  }
  PermWalk pw;
  pw.pm = 0xBA9876543210ull, pw.n = act ? 12u : 0u, pw.mask = 0xfu, pw.sh = 0u, pw.open = 0u;
  {
    if (r.pos < 16) {
#define SK_CALL(s) perm_batch<true, NP>(r, s, pw);
      SK_FOUR_BATCHES(SK_CALL)
#undef SK_CALL
      r.pos = pw.n ? 16 : r.pos;
    }
    while (SK_WALKING(pw.n)) {  // (the next chunk's loads are in flight since the deck's last refill)
      r.refill(pw.n != 0u);
#define SK_CALL(s) perm_batch<false, NP>(r, s, pw);
      SK_FOUR_BATCHES(SK_CALL)
#undef SK_CALL
      r.pos = pw.n ? 16 : r.pos;
    }
  }
#undef SK_WALKING
  if (!act) return;
  // ---- assemble the record (skyjo_layout.h) in registers ----
  uint32_t rec[20 * 4];
  const int nwords = L.chunks * 4;
#pragma unroll
  for (int w = 0; w < 20 * 4; w++) rec[w] = 0;
  auto setb = [&](int off, uint32_t val) { rec[off >> 2] |= (val & 0xffu) << ((off & 3) * 8); };
  auto pack4 = [&](int k) {  // deck cards k .. k+3 as four bytes: one aligned word of the lane's deck (cards beyond the deck read as 0)
    const uint32_t w = DK_AT32(dk + k);
    return k + 4 <= SK_NCARDS ? w : (w & (0xffffffffu >> (8 * (k + 4 - SK_NCARDS))));
  };
  const int last = (int)(int8_t)DKW(SK_NCARDS - 1);
  int best = 0, bs = -1000, ms = 1000;
#pragma unroll
  for (int p = 0; p < NP; p++) {
    const int s0 = (int)((pw.open >> (8 * p)) & 0xfu), s1 = (int)((pw.open >> (8 * p + 4)) & 0xfu);
    const int c0 = (int)(int8_t)DKW(12 * p + s0), c1 = (int)(int8_t)DKW(12 * p + s1), sum = c0 + c1;
    if (sum > bs) bs = sum, best = p;  // first argmax of revealed sums (skyjo.py:105-125)
    ms = sum < ms ? sum : ms;
    const int blk = sk_pb(L, p) >> 2;  // word index of the player's block: cards[3], counters, vis[3], placed
    rec[blk + 3] = ((uint32_t)sum & 0xffffu) | (10u << 16);  // sum, hidden = 10, refunded = 0
#pragma unroll
    for (int j = 0; j < 3; j++) {  // vis row: 15 everywhere but the two open slots; cards row-major (skyjo.py:63-65)
      uint32_t w = 0x0f0f0f0fu;
      if ((s0 >> 2) == j) w = (w & ~(0xffu << ((s0 & 3) * 8))) | (((uint32_t)c0 & 0xffu) << ((s0 & 3) * 8));
      if ((s1 >> 2) == j) w = (w & ~(0xffu << ((s1 & 3) * 8))) | (((uint32_t)c1 & 0xffu) << ((s1 & 3) * 8));
      rec[blk + 4 + j] = w;
      rec[blk + j] = pack4(12 * p + 4 * j);
    }
    if (!L.indirect) {  // direct observation: open cards are counted too (skyjo.py:160,236-248)
#pragma unroll
      for (int w = 4; w <= 8; w++) {
        const int b0 = H_HIST + 2 + c0, b1 = H_HIST + 2 + c1;
        rec[w] += ((b0 >> 2) == w ? 1u << ((b0 & 3) * 8) : 0u) + ((b1 >> 2) == w ? 1u << ((b1 & 3) * 8) : 0u);
      }
    }
  }
  {
    const int bl = H_HIST + 2 + last;
#pragma unroll
    for (int w = 4; w <= 8; w++) rec[w] += (bl >> 2) == w ? 1u << ((bl & 3) * 8) : 0u;
  }
  static_assert((H_PILE & 3) == 0, "the pile buffer starts on a word");
#pragma unroll
  for (int d = 0; d < (SK_NCARDS + 3) / 4; d++) {  // draw pile = rest[0 .. R-2], discard pile = [rest[R-1]] at the far end
    uint32_t m = 0;
    for (int j = 0; j < 4; j++)
      if (4 * d + j < R - 1) m |= 0xffu << (8 * j);
    uint32_t w = m ? (pack4(12 * NP + 4 * d) & m) : 0u;
    if (d == (SK_NCARDS - 1) / 4) w |= ((uint32_t)last & 0xffu) << (((SK_NCARDS - 1) & 3) * 8);
    rec[(H_PILE >> 2) + d] |= w;
  }
  rec[0] = (uint32_t)best << 8 | (uint32_t)F_VALID << 16 | (uint32_t)SKYJO_ST_RESET << 24;
  rec[1] = (uint32_t)(R - 1) | 1u << 8 | ((uint32_t)last & 0xffu) << 24;
  rec[2] = (uint32_t)SKYJO_HAND_NONE << 24;
  rec[3] = episode;
  setb(H_MINSUM, (uint32_t)(ms < 127 ? ms : 127)), setb(H_MINHID, 10);
#pragma unroll
  for (int c = 0; c < 20; c++)
    if (4 * c < nwords) dst[c] = make_uint4(rec[4 * c], rec[4 * c + 1], rec[4 * c + 2], rec[4 * c + 3]);
}
#undef DKW

template <class Rng>
__device__ __forceinline__ void deal_into_lds(const SkParams &P, uint8_t *lp, Rng &r, uint32_t episode) {
  const int N = P.L.N, pb = P.L.off_pile, R = SK_NCARDS - 12 * N;
  const int pw = pb >> 2;   // word index of the pile buffer (4-byte aligned)
  const int tmp = pb + R;   // 12 free bytes behind the rest (R + 12 <= 150)
  for (int c = 0; c < P.L.chunks; c++) LQ(c) = make_uint4(0u, 0u, 0u, 0u);
  // _new_drawpile: repeat(arange(-2, 13), 10) then shuffle (skyjo.py:76-82); written four cards per word
  for (int d = 0; d < (SK_NCARDS + 3) / 4; d++) {
    uint32_t w = 0;
    for (int j = 0; j < 4; j++) {
      const int i = 4 * d + j;
      w |= (i < SK_NCARDS ? (uint32_t)((-2 + i / 10) & 0xff) : 0u) << (8 * j);
    }
    LW(pw + d) = w;
  }
  // The N + 2 shuffles of a deal, in numpy's order, share ONE inlined copy of the
  // shuffle loop (and of the MT19937 refill code in it): segment 0 = the deck, 1 = the rest,
  // 2 + p = permutation(12) of player p.
#pragma unroll 1
  for (int seg = 0; seg < N + 2; seg++) {
    int base = pb, n = SK_NCARDS;
    if (seg == 1) {
      // first 12N cards row-major to players 0..N-1 (skyjo.py:63-65)
      for (int p = 0; p < N; p++)
        for (int j = 0; j < 3; j++) LW((sk_pb(P.L, p) >> 2) + j) = LW(pw + 3 * p + j);
      // the rest is shuffled again; all but its last card form the draw pile (skyjo.py:68-70,127-138).
      // Word-wise move down by 3N words, 8 words at a time (reads of a batch precede its writes).
      for (int d = 0; d < (R + 3) / 4; d += 8) {
        uint32_t t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = LW(pw + 3 * N + d + k);  // may run a few words past the pile: padding
#pragma unroll
        for (int k = 0; k < 8; k++)
          if (d + k < (R + 3) / 4) LW(pw + d + k) = t[k];
      }
      n = R;
    } else if (seg >= 2) {
      // _reset_card_mask: two open cards per player = permutation(12)[:2] (skyjo.py:96-103)
      const int p = seg - 2;
      for (int k = 0; k < 12; k++) LB(tmp + k) = (uint8_t)k, LB(sk_pb(P.L, p) + PB_VIS + k) = SKYJO_HAND_NONE;
      base = tmp, n = 12;
    }
    shuffle_lds(lp, base, n, r);
    if (seg >= 2) {
      const int p = seg - 2;
      int s0 = LB(tmp), s1 = LB(tmp + 1);
      const int blk = sk_pb(P.L, p);
      int c0 = LI(blk + PB_CARDS + s0), c1 = LI(blk + PB_CARDS + s1);
      LB(blk + PB_VIS + s0) = (uint8_t)c0, LB(blk + PB_VIS + s1) = (uint8_t)c1;
      LSH(blk + PB_SUM) = (int16_t)(c0 + c1);
      LB(blk + PB_HIDDEN) = 10;
      if (!P.L.indirect) LB(H_HIST + 2 + c0)++, LB(H_HIST + 2 + c1)++;
    }
  }
  for (int k = R; k < SK_NCARDS; k++) LB(pb + k) = 0;
  const int last = LI(pb + R - 1);
  LB(pb + R - 1) = 0;
  LB(pb + SK_NCARDS - 1) = (uint8_t)last;  // discard pile = [last], stored from the far end
  LB(H_HIST + 2 + last)++;
  // _reset_start_player: first argmax of revealed sums draws first (skyjo.py:105-125)
  int best = 0, bs = LSH(sk_pb(P.L, 0) + PB_SUM);
  for (int p = 1; p < N; p++) {
    int s = LSH(sk_pb(P.L, p) + PB_SUM);
    if (s > bs) bs = s, best = p;
  }
  LB(H_PHASE) = 0, LB(H_PLAYER) = (uint8_t)best, LB(H_FLAGS) = F_VALID, LB(H_STATUS) = SKYJO_ST_RESET;
  LB(H_NDRAW) = (uint8_t)(R - 1), LB(H_NDISC) = 1, LB(H_ROLE) = 0;
  LB(H_TOP) = (uint8_t)last, LB(H_HAND) = SKYJO_HAND_NONE;
  *(uint32_t *)(lp + LIDX(H_EPISODE)) = episode;
  refresh_minima(P, lp);
}

__device__ __forceinline__ bool deal_inline(const SkParams &P, uint8_t *lp, uint8_t *fp, int g, int tile, int lane, int head) {
  const uint32_t ep = P.deals_consumed[g];
  const int busy = P.busy[g];
  if (busy) {
    // The bank is empty, but the dealing launch that overlaps this kernel is dealing exactly the episode needed
    // (slot `head`, the next in stream order) - unless a reshuffle already rolled that deal back.
    const bool cancelled = P.cancel[g] != 0;
    if (P.rng_mode == SKYJO_RNG_MT19937 && !cancelled) {  // the stream is shared: wait for that deal and take it
      const int w = wait_deal_done(P, g);
      if (w == SK_WAIT_TIMEOUT) {
        LB(H_FLAGS) = F_VALID | F_DONE, LB(H_STATUS) = SKYJO_ST_ERROR;
        for (int q = 0; q < P.L.N; q++) P.rewards[(size_t)g * P.L.N + q] = 0.0;  // (nothing stale for an episode-end column to pass on)
        return false;
      }
      P.cancel[g] = 1;  // taken here: the publishing kernel must not mark the slot ready
      if (w == SK_WAIT_OK) {
        load_spare(P, lp, busy - 1, tile, lane);
        bank_advance(P, lp, g, head, ep);
        return true;
      }
    }  // Philox deals do not depend on a stream position (and a cancelled / overrun MT deal has finished): deal here
    P.cancel[g] = 1;  // superseded: the publishing kernel must not mark the slot ready
  }
  if (P.rng_mode == SKYJO_RNG_MT19937) {
    MtStream<16> r;
    r.open(P.mt + (size_t)g * 624, P.mt_idx[g], fp);
    deal_into_lds(P, lp, r, ep);
    P.mt_idx[g] = r.close();
  } else {
    PhiloxStream r;
    r.open(P.seeds[g] + 1, ep, 0u, 0u);
    deal_into_lds(P, lp, r, ep);
  }
  LB(H_BANK) = (uint8_t)head;  // the bank is empty; its head pointer survives the new record
  P.deals_consumed[g] = ep + 1;
  P.done[g] = 0;
  return true;
}

// ------------------------------------------------------------------------------------------
// Dealing pipeline, once per dealing interval (80 lockstep iterations by default for three and more players):
//   k_scan    (caller's stream)  finds the banks that are not full with a wavefront ballot + prefix popcount,
//                                appends (game, episode) to the work list and marks the games busy;
//   k_deal    (own stream, may overlap the following k_step launches) deals one episode per listed game,
//                                one lane per game on densely filled wavefronts;
//   k_publish (caller's stream, after k_deal has finished) marks the new slots ready and clears busy.
// All bank bookkeeping (head, ready flags, busy, cancel) is only ever written on the caller's stream.
// ------------------------------------------------------------------------------------------
#define SK_SCAN_BLOCK 1024
__global__ __launch_bounds__(SK_SCAN_BLOCK) void k_scan(SkParams P, int list_sel) {
  // One atomic per 1024 games reserves the block's stretch of the work list (same-address atomics serialise at
  // ~12 ns each: one per wavefront made this kernel 13 us long, two thirds of it queueing on deal_count).
  __shared__ uint32_t wave_need[SK_SCAN_BLOCK / 64], block_first;
  const size_t G = (size_t)P.tiles * SK_TILE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int32_t *list = P.deal_list + (size_t)list_sel * G;
  uint32_t *eps = P.deal_ep + (size_t)list_sel * G;
  for (int base = blockIdx.x * SK_SCAN_BLOCK; base < P.B; base += gridDim.x * SK_SCAN_BLOCK) {
    const int g = base + (int)threadIdx.x;
    bool need = false;
    int slot = 0, r = 0;
    uint32_t consumed = 0;
    if (g < P.B) {
      // every flag is requested before the first is looked at: one memory round trip per game, not SK_BANK + 2
      const uint8_t busy = P.busy[g];
      const int head = P.bank_head[g] % SK_BANK;
      consumed = P.deals_consumed[g];
      uint8_t ready[SK_BANK];
#pragma unroll
      for (int k = 0; k < SK_BANK; k++) ready[k] = P.spare_ready[(size_t)k * G + g];
      bool open = true;
#pragma unroll
      for (int k = 0; k < SK_BANK; k++) {  // r = number of ready slots in stream order from `head`
        uint8_t f = 0;
#pragma unroll
        for (int j = 0; j < SK_BANK; j++) f = (head + k) % SK_BANK == j ? ready[j] : f;
        open = open && f != 0;
        r += open ? 1 : 0;
      }
      need = !busy && r < SK_BANK;
      slot = (head + r) % SK_BANK;  // slots fill in stream order
    }
    const unsigned long long b = __ballot(need);
    const unsigned long long be = __ballot(need && r == 0);  // nothing in the bank: one more game end before the next run deals in place
    if (lane == 0) {
      wave_need[wave] = (uint32_t)__popcll(b);
      if (be) atomicAdd(P.bank_empty, (uint32_t)__popcll(be));  // (rare)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t total = 0;
      for (int w = 0; w < SK_SCAN_BLOCK / 64; w++) {
        const uint32_t n = wave_need[w];
        wave_need[w] = total;  // -> offset of the wavefront inside the block's stretch
        total += n;
      }
      block_first = total ? atomicAdd(&P.deal_count[list_sel], total) : 0u;
    }
    __syncthreads();
    if (need) {
      const uint32_t pos = block_first + wave_need[wave] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
      list[pos] = g;
      eps[pos] = consumed + (uint32_t)r;
      P.busy[g] = (uint8_t)(1 + slot);
      P.cancel[g] = 0;
      P.plan_tag[g] = P.deal_tag;
    }
    __syncthreads();  // (the shared words are reused by the next stretch)
  }
}

// Hand a dealing run's episodes to the step kernel: on the caller's stream after k_deal when that ran on a stream
// of its own; k_deal does the same per lane itself when it runs in line (publish_inline).  Also clears the other
// work list's counter for the next run's k_scan.
__global__ __launch_bounds__(256) void k_publish(SkParams P, int list_sel) {
  const size_t G = (size_t)P.tiles * SK_TILE;
  const int32_t *list = P.deal_list + (size_t)list_sel * G;
  const int count = (int)P.deal_count[list_sel];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    P.deal_count[list_sel ^ 1] = 0;
    P.health_host[0] = *P.bank_empty, P.health_host[1] = P.deal_tag;
    *P.bank_empty = 0;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const int g = list[i];
    const int slot = P.busy[g] - 1;
    if (slot >= 0 && !P.cancel[g] && P.done_flag[g] == P.deal_tag) P.spare_ready[(size_t)slot * G + g] = 1;
    P.busy[g] = 0, P.cancel[g] = 0;
  }
}

// The pipelined form's hand-over as a kernel of its own, for the host's synchronisation points (get_state, snapshot,
// seed ...): every dealing launch has finished (the caller's stream waited for the dealing stream), lane = game.
__global__ __launch_bounds__(256) void k_publish_all(SkParams P) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < P.tiles * SK_TILE) sk_publish_deals(P, g);
}

// One dealing wavefront: games block * 64 .. (lane = game in the forms without a work list), its own LDS region `lds_raw`.
template <int NP>
__device__ __forceinline__ void deal_body(const SkParams &P, int list_sel, int publish_inline, const int block, const int lane, uint32_t *lds_raw,
                                          const bool report_health = true) {
  TRACE_DECL;
  uint8_t *lp = (uint8_t *)lds_raw + lane * 16;
  const size_t G = (size_t)P.tiles * SK_TILE;
  const int count = (int)P.deal_count[list_sel];
  const int i = block * SK_TILE + lane;
  // publish_inline == 2: in line AND its own scan - lane = game, every lane looks at its game's bank itself (what k_scan
  // does, minus the work list: no launch in front of this one; the lanes whose bank is full idle through the refill loops)
  // publish_inline == 3: beside the step kernel, lane = game as well - the step kernel planned this run on its way out
  // (sk_plan_deals) and publishes it on its way into a later launch; this kernel deals and signals, as with a work list
  const bool fused = publish_inline == 2, piped = publish_inline == 3;
  if (piped && report_health && block == 0 && lane == 0) {
    uint32_t *be = P.bank_empty + (P.be_read & 1u);  // the launch(es) that counted into this word have finished: the count is complete
    P.health_host[0] = *be, P.health_host[1] = P.deal_tag;
    *be = 0;
  }
  if (!piped && publish_inline && block == 0 && lane == 0) {
    P.deal_count[list_sel ^ 1] = 0;  // for the next run's k_scan
    // (host-mapped memory: the host adapts the dealing interval.)  Fused, this run's count of empty banks is still being
    // added up by the other wavefronts: the previous run's goes out, each run counts into the word of its list_sel.
    uint32_t *be = P.bank_empty + (fused ? (list_sel ^ 1) & 1 : 0);
    P.health_host[0] = *be, P.health_host[1] = P.deal_tag;
    *be = 0;
  }
  if (!fused && !piped && block * SK_TILE >= count) return;
  for (uint32_t k = 0; k < P.debug_deal_delay; k++) __builtin_amdgcn_s_sleep(127);  // (fault injection only: 0 in production)
  const int tile = block;  // stamp slot
  (void)tile;
  STAMP_DECL;
  STAMP(0);
  int g, slot;
  uint32_t ep;
  bool act;
  if (fused) {
    const bool listed = i < P.B;
    g = listed ? i : 0;
    const uint8_t busy = P.busy[g];
    const int head = P.bank_head[g] % SK_BANK;
    const uint32_t consumed = P.deals_consumed[g];
    uint8_t ready[SK_BANK];
#pragma unroll
    for (int k = 0; k < SK_BANK; k++) ready[k] = P.spare_ready[(size_t)k * G + g];
    bool open = true;
    int r = 0;
#pragma unroll
    for (int k = 0; k < SK_BANK; k++) {  // r = number of ready slots in stream order from `head` (as in k_scan)
      uint8_t f = 0;
#pragma unroll
      for (int j = 0; j < SK_BANK; j++) f = (head + k) % SK_BANK == j ? ready[j] : f;
      open = open && f != 0;
      r += open ? 1 : 0;
    }
    act = listed && !busy && r < SK_BANK;
    slot = (head + r) % SK_BANK;
    ep = consumed + (uint32_t)r;
    const unsigned long long be = __ballot(act && r == 0);
    if (be && lane == 0) atomicAdd(P.bank_empty + (list_sel & 1), (uint32_t)__popcll(be));  // (rare)
  } else if (piped) {
    const bool listed = i < P.B;
    g = listed ? i : 0;
    const int owner = listed ? P.busy[g] : 0;
    act = owner > 0 && P.plan_tag[g] == P.deal_tag;  // (busy with an older id: a deal of an earlier run that is not published yet)
    slot = act ? owner - 1 : 0;
    ep = P.plan_ep[g];
  } else {
    const bool listed = i < count;
    g = listed ? P.deal_list[(size_t)list_sel * G + i] : 0;
    ep = listed ? P.deal_ep[(size_t)list_sel * G + i] : 0u;
    const int owner = listed ? P.busy[g] : 0;
    act = owner > 0;  // (an entry whose game is not marked busy would be a stale list: never dealt)
    slot = act ? owner - 1 : 0;
  }
  uint4 *dst = P.spare + bank_rec16(P, slot, g);  // (game-major: the record's pieces are consecutive)
  bool mt_overrun = false;
  if (P.rng_mode == SKYJO_RNG_MT19937) {
    // The stream advances in place; the position it had before this deal is kept with the slot so that a mid-game
    // reshuffle of the live episode (which numpy would have drawn BEFORE this deal) can step the stream back
    // (mt_untwist) and have the deal redone (reshuffle_dispatch).
    // (the lanes without a deal go through the compact deal too, walking nothing: MtChunkStream::refill needs the whole wavefront)
    const int packed = P.mt_idx[g];
    if (act) P.mt_idx[(size_t)(1 + slot) * G + g] = packed;
    int generated = 0;
    if (NP > 0) {
      MtChunkStream r;
      // (a lane without a deal loads along - from the very first state, whose lines every such lane of the chip asks for)
      r.open(P.mt, act ? (uint32_t)g * 624u : 0u, packed, (uint8_t *)lds_raw + SK_TILE * SK_DECK_STRIDE, lane);
      STAMP(2);
      deal_compact<NP>(P, lds_raw, lane, r, ep, dst, act);
      if (act) P.mt_idx[g] = r.close();
      generated = r.chunks_made * 16;
    } else if (act) {
      uint8_t *fp = (uint8_t *)lds_raw + P.L.chunks * 1024 + lane * 4;
      MtStream<64> r;
      r.open(P.mt + (size_t)g * 624, packed, fp);
      r.stp = &st;
      STAMP(2);
      deal_into_lds(P, lp, r, ep);
      P.mt_idx[g] = r.close();
      generated = r.wp - (((16 - ((packed >> 16) & 15)) & 15) + (packed >> 16));
    }
    if (act) {
      uint32_t *mt = P.mt + (size_t)g * 624;
      mt_overrun = generated > 624 - 64;  // close to a full turn of the state: positions alone could no longer tell
      if (mt_overrun) {                   // how far a rollback has to go, so give this speculation up right here
        int k0 = (packed & 0xffff) + (packed >> 16);
        k0 = k0 >= 624 ? k0 - 624 : k0;
        int k1 = k0 + generated;
        k1 = k1 >= 624 ? k1 - 624 : k1;
        mt_untwist(mt, k0, k1);
        P.mt_idx[g] = packed;
      }
      STAMP(3);
    }
  } else if (act) {
    if (NP > 0) {
      PhiloxChunkStream r;
      r.open(P.seeds[g] + 1, ep, 0u, 0u);
      deal_compact<NP>(P, lds_raw, lane, r, ep, dst, true);
    } else {
      PhiloxStream r;
      r.open(P.seeds[g] + 1, ep, 0u, 0u);
      deal_into_lds(P, lp, r, ep);
    }
  }
  if (act) {
    if (NP == 0) store_spare(P, lp, slot, g);
    STAMP(4);
  }
  if (publish_inline == 1 || publish_inline == 2) {
    // in line on the caller's stream: no step kernel runs beside this one, so nothing can have cancelled the deal
    // and the next kernel on the stream sees every store - mark the slot ready right here (what k_publish does)
    if (act) {
      if (!mt_overrun) P.spare_ready[(size_t)slot * G + g] = 1;
      P.busy[g] = 0;
    }
  } else {
    // hand the finished deals over: every store above must be visible device-wide before the flag is
    SK_FENCE_RELEASE(P);
    if (act)
      __hip_atomic_store(&P.done_flag[g], P.deal_tag | (mt_overrun ? 0x80000000u : 0u), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
#ifdef SK_STAMPS
  if (lane == 0 && tile < P.tiles)
    for (int k = 0; k < 8; k++) P.stamps[(size_t)(P.tiles + tile) * 8 + k] += st.acc[k];
#endif
  TRACE_STORE(1, P.deal_tag, lane, block);
}

template <int NP>
__global__ __launch_bounds__(SK_TILE) void k_deal(SkParams P, int list_sel, int publish_inline) {
  extern __shared__ uint32_t lds_raw[];
  deal_body<NP>(P, list_sel, publish_inline, (int)blockIdx.x, (int)threadIdx.x, lds_raw);
}
