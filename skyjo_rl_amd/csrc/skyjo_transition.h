// skyjo_transition.h - part of skyjo_device.h (included from there, in its place: the parts build on each other in that order).
// The state transition: minima, mid-game reshuffle (with the stream's roll-back), scoring and final rewards, the hot path's header / observation registers, the on-device policy's pick, SkyjoGame.act.
#pragma once
#ifndef SKYJO_DEVICE_PARTS
#error "include skyjo_device.h"
#endif

// ------------------------------------------------------------------------------------------
// min over players of revealed sums / hidden counts -> obs[0], obs[1] (skyjo.py:182-183)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void refresh_minima(const SkParams &P, uint8_t *lp) {
  int ms = LSH(sk_pb(P.L, 0) + PB_SUM), mh = LB(sk_pb(P.L, 0) + PB_HIDDEN);
  for (int q = 1; q < P.L.N; q++) {
    int s = LSH(sk_pb(P.L, q) + PB_SUM), h = LB(sk_pb(P.L, q) + PB_HIDDEN);
    ms = s < ms ? s : ms, mh = h < mh ? h : mh;
  }
  LB(H_MINSUM) = (uint8_t)(int8_t)(ms < 127 ? ms : 127);
  LB(H_MINHID) = (uint8_t)mh;
}

// ------------------------------------------------------------------------------------------
// _reshuffle_discard_pile mid-game (skyjo.py:127-138, 361-365): the WHOLE discard pile incl. its
// top is shuffled in place, becomes the draw pile, and its last card opens the new discard pile.
// ------------------------------------------------------------------------------------------
template <class Rng>
__device__ __forceinline__ void reshuffle_discard(const SkParams &P, uint8_t *lp, Rng &r) {
  const int n = LB(H_NDISC), role = LB(H_ROLE), pb = P.L.off_pile;
  const int reg = role ? 0 : 1;  // region holding the discard pile
  for (int i = n - 1; i >= 1; i--) {
    int j = (int)rng_interval(r, (uint32_t)i);
    int ai = pb + pile_addr(reg, i), aj = pb + pile_addr(reg, j);
    uint8_t t = LB(ai);
    LB(ai) = LB(aj), LB(aj) = t;
  }
  // cards that leave the discard pile leave the histogram (skyjo.py:236-248 counts the pile)
  for (int k = 0; k < n - 1; k++) LB(H_HIST + 2 + LI(pb + pile_addr(reg, k)))--;
  int last = LI(pb + pile_addr(reg, n - 1));
  LB(pb + pile_addr(reg ^ 1, 0)) = (uint8_t)last;
  LB(H_NDRAW) = (uint8_t)(n - 1), LB(H_NDISC) = 1, LB(H_ROLE) = (uint8_t)(role ^ 1);
  LB(H_TOP) = (uint8_t)last;
  int rs = LB(H_RESH);
  LB(H_RESH) = (uint8_t)(rs < 255 ? rs + 1 : 255);
}

// While a dealing launch overlaps this kernel, the games it deals for are marked busy: it owns their RNG stream
// and one bank slot.  The rare paths that need the stream wait for that one deal to finish (the dealing launch
// never waits for anybody, so this cannot deadlock; the spin is bounded all the same).
// Returns SK_WAIT_OK when that deal is finished, SK_WAIT_GAVE_UP when it gave itself up (close to a full turn of the generator
// state, see k_deal: it then left no record and no trace in the stream), SK_WAIT_TIMEOUT when the dealing launch never showed
// up (it is not resident beside this kernel and this kernel cannot end before it starts).  After a timeout the dealing
// kernel may still be writing the game's stream: the caller must leave the stream and the bank slot alone.  It marks the
// game done on the spot (this episode is cut off - results are void from here on) and the sticky error word makes every
// later synchronising host call on the handle fail until it is re-seeded.  (Kept as small as this on purpose: the rare
// paths are inlined into the step kernel, and what they contain moves the register allocation of its hot loop - an early
// exit on the sticky word and a host-mapped store in here cost the fused rollout 2 % of its time.)
#define SK_WAIT_OK 0
#define SK_WAIT_GAVE_UP 1
#define SK_WAIT_TIMEOUT 2
__device__ __forceinline__ int wait_deal_done(const SkParams &P, int g) {
  uint32_t f = 0;
  const uint32_t tag = P.plan_tag[g];  // the run that owns the game's busy slot (written on this stream, before this kernel or by this lane)
  for (int spin = 0; spin < (1 << P.spin_log2); spin++) {
    f = __hip_atomic_load(&P.done_flag[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((f & 0x7fffffffu) == tag) break;
    __builtin_amdgcn_s_sleep(32);
  }
  if ((f & 0x7fffffffu) != tag) {
    atomicOr(P.dev_error, SK_ERR_DEAL_TIMEOUT);  // (the host-style kernels hand the word to the host: sk_error_to_host)
    return SK_WAIT_TIMEOUT;
  }
  SK_FENCE_ACQUIRE(P);
  return (f >> 31) != 0 ? SK_WAIT_GAVE_UP : SK_WAIT_OK;
}

// MT19937's in-place regeneration is invertible, so a deal that has to be taken back needs no log of the values it
// overwrote.  Element i was made as  new[i] = S[i+397] ^ twist((S[i] & 0x80000000) | (S[i+1] & 0x7fffffff)) ; walking
// backwards from the newest element, S[i+397] and S[i+1] are exactly what they were when i was made, so
// twist(y) = new[i] ^ S[i+397]; bit 31 of twist(y) tells whether y was odd (the magic constant has it set, y >> 1 has
// not), which gives y back: its top bit is old S[i]'s, its low 31 bits are old S[i+1]'s.  The low 31 bits of the
// OLDEST element undone stay unknown (zero) - and are never looked at again: the only thing they ever feed is the
// element before it, which was made earlier and is still in place (tests/test_untwist_identity.py shows the stream
// continuing identically; undoing the deal before this one restores them first thing).
__device__ __forceinline__ uint32_t mt_untwist_y(const uint32_t *mt, int i) {
  uint32_t t = mt[i] ^ mt[i + 397 >= 624 ? i + 397 - 624 : i + 397];
  const uint32_t odd = t >> 31;
  t ^= odd ? 0x9908b0dfu : 0u;
  return (t << 1) | odd;
}
__device__ __forceinline__ void mt_untwist(uint32_t *mt, int from, int to) {  // undo elements [from, to) in stream order
  for (int i = to; i != from;) {
    const int nx = i == 624 ? 0 : i;  // (to may be given as 624)
    i = nx == 0 ? 623 : nx - 1;
    const uint32_t y = mt_untwist_y(mt, i);
    const int ip1 = i == 623 ? 0 : i + 1;
    mt[ip1] = (mt[ip1] & 0x80000000u) | (y & 0x7fffffffu);
    mt[i] = y & 0x80000000u;
  }
}

// Take back the deal that filled bank slot `slot`: the stream returns to where it stood before that deal.
__device__ __forceinline__ int mt_rollback(const SkParams &P, uint32_t *mt, int g, int slot, int packed) {
  const size_t G = (size_t)P.tiles * SK_TILE;
  const int snap = P.mt_idx[(size_t)(1 + slot) * G + g];
  int from = (snap & 0xffff) + (snap >> 16), to = (packed & 0xffff) + (packed >> 16);
  from = from >= 624 ? from - 624 : from, to = to >= 624 ? to - 624 : to;
  mt_untwist(mt, from, to);
  return snap;
}

// When the game's stream cannot be had (wait_deal_done timed out: sticky device error) nothing of it is touched: the game
// is marked done in its LDS header - status SKYJO_ST_ERROR, one stale card left on the draw pile for the draw that called -
// and the step kernel carries on without a branch of its own for this.
__device__ __forceinline__ void reshuffle_dispatch(const SkParams &P, uint8_t *lp, uint8_t *fp, int g) {
  if (P.rng_mode == SKYJO_RNG_MT19937) {
    const size_t G = (size_t)P.tiles * SK_TILE;
    uint32_t *mt = P.mt + (size_t)g * 624;
    // The pre-dealt episodes consumed the stream beyond this point (numpy draws the reshuffle first): roll the
    // state back over them (mt_untwist), newest deal first; the dealing kernel deals them again afterwards.
    const int head = LB(H_BANK) % SK_BANK;
    const int busy = P.busy[g];
    const bool inflight = busy && !P.cancel[g];  // (already cancelled = already finished and undone)
    bool undo_inflight = false;
    if (inflight) {  // a deal is in flight for this game: let it finish, then undo it as well
      const int w = wait_deal_done(P, g);
      if (w == SK_WAIT_TIMEOUT) {
        LB(H_FLAGS) |= F_DONE, LB(H_STATUS) = SKYJO_ST_ERROR, LB(H_NDRAW) = 1;
        P.done[g] = 1;
        for (int q = 0; q < P.L.N; q++) P.rewards[(size_t)g * P.L.N + q] = 0.0;  // (an episode-end column must not pass stale values on)
        return;
      }
      undo_inflight = w == SK_WAIT_OK;
      P.cancel[g] = 1;
    }
    int packed = P.mt_idx[g];
    if (undo_inflight) packed = mt_rollback(P, mt, g, busy - 1, packed);
    for (int k = SK_BANK - 1; k >= 0; k--) {
      const int slot = (head + k) % SK_BANK;
      if (P.spare_ready[(size_t)slot * G + g]) {
        packed = mt_rollback(P, mt, g, slot, packed);
        P.spare_ready[(size_t)slot * G + g] = 0;
      }
    }
    MtStream<16> r;
    r.open(mt, packed, fp);
    reshuffle_discard(P, lp, r);
    P.mt_idx[g] = r.close();
  } else {
    PhiloxStream r;
    r.open(P.seeds[g] + 1, *(uint32_t *)(lp + LIDX(H_EPISODE)), LB(H_RESH), 1u);
    reshuffle_discard(P, lp, r);
  }
}

// ------------------------------------------------------------------------------------------
// _evaluate_game + _calc_final_rewards (skyjo.py:477-498, skyjo_env.py:293-312), float64, no FMA
// contraction (compiled with -ffp-contract=off), numpy's pairwise summation order for the mean.
// ------------------------------------------------------------------------------------------
// Per-LANE float64 statistics (SK_ACC_KINDS x N doubles per lane behind the record staging area): element k of lane l at
// ap + k * 512 (ap = base + l * 8).  Only lanes whose game has just ended add to them, as fire-and-forget LDS atomics on
// lane-private addresses.  (One shared set per wavefront would save 6 KB of LDS, but for atomics on a wavefront-uniform
// address the compiler emits a scalar loop over the active lanes per atomic - twelve loops per game end: k_step +10 %.)
#define ACC(k) (*(double *)(ap + ((k) << 9)))
__device__ __forceinline__ void acc_add(uint8_t *ap, int k, double v) {
  __hip_atomic_fetch_add(&ACC(k), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void acc_episode(uint8_t *ap, int N, int p, double score, double reward, int refunded) {
  acc_add(ap, p, score), acc_add(ap, N + p, reward), acc_add(ap, 2 * N + p, reward * reward);
  if (refunded) acc_add(ap, 3 * N + p, (double)refunded);
}

__device__ __forceinline__ void finish_game(const SkParams &P, uint8_t *lp, uint8_t *fp, uint8_t *ap, int g,
                                            int finisher) {
  const int N = P.L.N;
  double *sc = P.scores + (size_t)g * N, *rw = P.rewards + (size_t)g * N;
  // raw integer scores wait in the lane's (idle) RNG FIFO words: MT_FIFO(p), p < 12
  int mn = 0, fs = 0;
  for (int p = 0; p < N; p++) {
    int s = 0;
    for (int c = 0; c < 4; c++) {
      int b = sk_pb(P.L, p) + PB_CARDS + 3 * c;
      int t0 = LI(b), t1 = LI(b + 1), t2 = LI(b + 2);
      if (!(t0 == t1 && t1 == t2)) s += t0 + t1 + t2;  // skyjo.py:488-493, hidden cards included
    }
    MT_FIFO(p) = (uint32_t)s;
    mn = (p == 0 || s < mn) ? s : mn;
    fs = p == finisher ? s : fs;
  }
  const bool penal = mn != fs;  // skyjo.py:496-497 (integer compare == the float compare of equal-typed sums)
#define SCORE(p) ((penal && (p) == finisher) ? (double)(int)MT_FIFO(p) * P.score_penalty : (double)(int)MT_FIFO(p))
  double sum;
  if (N < 8) {
    sum = 0.0;
    for (int p = 0; p < N; p++) sum += SCORE(p);
  } else {
    sum = ((SCORE(0) + SCORE(1)) + (SCORE(2) + SCORE(3))) + ((SCORE(4) + SCORE(5)) + (SCORE(6) + SCORE(7)));
    for (int p = 8; p < N; p++) sum += SCORE(p);
  }
  const double mean = sum / (double)N;
  for (int p = 0; p < N; p++) {
    const double d = SCORE(p);
    double r = (-d + mean) + P.mean_reward;
    const int rf = LB(sk_pb(P.L, p) + PB_REFUNDED);
    if (P.reward_refunded != 0.0) r += (double)rf * P.reward_refunded;
    sc[p] = d, rw[p] = r;
    acc_episode(ap, N, p, d, r, rf);
  }
#undef SCORE
  P.done[g] = 1;
}

// The same for a compile-time player count: card rows come in as dwords, scores stay in registers.  `rows`: this lane's
// card chunk of player 0, player p's `stride` bytes further - the live tile, or the copy a deferred scoring works on.
// `racc` != nullptr: the per-seat statistics are kept in the lane's REGISTERS (SK_ACC_KINDS x NP doubles, constant indices
// after unrolling) instead of the LDS slots behind `ap`: no LDS atomics, and 6 KB less LDS per wavefront at three players
// (the step kernels with a compile-time player count; EXPERIMENTS.md round 3).
template <int NP>
__device__ __forceinline__ void finish_game_fixed(const SkParams &P, const uint8_t *rows, int stride, uint8_t *ap, int g, int finisher,
                                                  double *racc = nullptr) {
  double *sc = P.scores + (size_t)g * NP, *rw = P.rewards + (size_t)g * NP;
  int s[NP], refunded[NP];
  int mn = 0, fs = 0;
#pragma unroll
  for (int p = 0; p < NP; p++) {
    const uint4 row = *(const uint4 *)(rows + p * stride);  // cards + the player's counters in one read
    const uint32_t c0 = row.x, c1 = row.y, c2 = row.z;
    refunded[p] = (int)(row.w >> 24);
    const uint32_t tri[4] = {c0 & 0xffffffu, (c0 >> 24) | ((c1 & 0xffffu) << 8), (c1 >> 16) | ((c2 & 0xffu) << 16), c2 >> 8};
    int t = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int t0 = (int)(int8_t)tri[c], t1 = (int)(int8_t)(tri[c] >> 8), t2 = (int)(int8_t)(tri[c] >> 16);
      t += (t0 == t1 && t1 == t2) ? 0 : t0 + t1 + t2;  // skyjo.py:488-493, hidden cards included
    }
    s[p] = t;
    mn = (p == 0 || t < mn) ? t : mn;
    fs = p == finisher ? t : fs;
  }
  const bool penal = mn != fs;  // skyjo.py:496-497
  double d[NP], sum = 0.0;
#pragma unroll
  for (int p = 0; p < NP; p++) {
    d[p] = (penal && p == finisher) ? (double)s[p] * P.score_penalty : (double)s[p];
    sum += d[p];  // NP < 8: numpy's pairwise sum is the plain left-to-right sum
  }
  const double mean = sum / (double)NP;
#pragma unroll
  for (int p = 0; p < NP; p++) {
    double r = (-d[p] + mean) + P.mean_reward;
    if (P.reward_refunded != 0.0) r += (double)refunded[p] * P.reward_refunded;
    sc[p] = d[p], rw[p] = r;
    if (racc) {
      racc[p] += d[p], racc[NP + p] += r, racc[2 * NP + p] += r * r, racc[3 * NP + p] += (double)refunded[p];
    } else {
      acc_episode(ap, NP, p, d[p], r, refunded[p]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Hot path.  The three header words live in registers (HdrRegs) for a whole launch; one turn costs
// three dependent LDS round trips: (A) the acting player's card / vis rows, (B) the pile byte that
// is drawn, (C) the histogram words and the next player's vis row for the output record.
// Histogram bins are bumped with fire-and-forget dword LDS atomics (bin k is byte k & 3 of its
// word; counts stay far below 256 so no carry crosses a byte).
// ------------------------------------------------------------------------------------------
struct HdrRegs {
  uint32_t w0, w1, w2;  // bytes 0..3, 4..7, 8..11 of the record (skyjo_layout.h)
};
#define HDR_LOAD(h) ((h).w0 = LW(0), (h).w1 = LW(1), (h).w2 = LW(2))
#define HDR_FLUSH(h) (LW(0) = (h).w0, LW(1) = (h).w1, LW(2) = (h).w2)

__device__ __forceinline__ int byte3(uint32_t a, uint32_t b, uint32_t c, int k) {  // signed byte k of a 12-byte row
  const uint32_t w = k < 4 ? a : (k < 8 ? b : c);
  return (int)(int8_t)(w >> ((k & 3) * 8));
}
__device__ __forceinline__ void put3(uint32_t &a, uint32_t &b, uint32_t &c, int k, int val) {
  const uint32_t sh = (uint32_t)(k & 3) * 8u, m = ~(0xffu << sh), v = ((uint32_t)val & 0xffu) << sh;
  if (k < 4) a = (a & m) | v;
  else if (k < 8) b = (b & m) | v;
  else c = (c & m) | v;
}
__device__ __forceinline__ void hist_add(uint8_t *lp, int value, int delta) {  // bins live at bytes 18..32
  const int b = H_HIST + 2 + value;
  uint32_t *w = (uint32_t *)(lp + LIDX(b & ~3));
  const uint32_t d = (uint32_t)delta << ((b & 3) * 8);
  __hip_atomic_fetch_add(w, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ uint32_t swar_nonzero01(uint32_t x) {
  return ((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u) >> 7;
}
__device__ __forceinline__ uint32_t pack12(uint32_t a, uint32_t b, uint32_t c) {  // 0/1 bytes -> 12 bits
  return ((a * 0x00204081u >> 21) & 0xfu) | (((b * 0x00204081u >> 21) & 0xfu) << 4) |
         (((c * 0x00204081u >> 21) & 0xfu) << 8);
}

// What an observer of the table needs from the expected player's `vis` row, kept in registers from the record of
// one iteration to the policy pick and the legality test of the next (same player, same row): the row itself and,
// per slot, "not refunded" (players_masked != 0) and "hidden" (players_masked == 2) as 0/1 bytes
// (skyjo.py:201-224).
struct ObsRegs {
  uint32_t q0, q1, q2, nz0, nz1, nz2, hd0, hd1, hd2;
};
__device__ __forceinline__ void obs_from_row(const uint4 &row, ObsRegs &o) {  // (the fourth word is the placed counter)
  o.q0 = row.x, o.q1 = row.y, o.q2 = row.z;
  o.nz0 = swar_nonzero01(o.q0 ^ 0xf2f2f2f2u);  // vis != -14  <=> players_masked != 0
  o.nz1 = swar_nonzero01(o.q1 ^ 0xf2f2f2f2u);
  o.nz2 = swar_nonzero01(o.q2 ^ 0xf2f2f2f2u);
  o.hd0 = swar_nonzero01(o.q0 ^ 0x0f0f0f0fu) ^ 0x01010101u;  // vis == 15 <=> players_masked == 2
  o.hd1 = swar_nonzero01(o.q1 ^ 0x0f0f0f0fu) ^ 0x01010101u;
  o.hd2 = swar_nonzero01(o.q2 ^ 0x0f0f0f0fu) ^ 0x01010101u;
}
__device__ __forceinline__ void obs_load(const SkParams &P, uint8_t *lp, int q, ObsRegs &o) {
  const uint4 row = LQ((sk_pb(P.L, q) + PB_VIS) >> 4);
  obs_from_row(row, o);
}

// uniform choice over the legal actions == policy_ra's p = mask / sum(mask)
// (rlskyjo/models/random_admissible_policy.py:26-28); word = Philox4x32-10 output for this
// (game, iteration), k = mulhi(word, n_legal), action = k-th legal action in ascending order.
__device__ __forceinline__ int policy_pick(int phase, const ObsRegs &o, uint32_t word) {
  if (phase == 0) return 24 + (int)__umulhi(word, 2u);
  uint32_t legal = pack12(o.nz0, o.nz1, o.nz2) | (pack12(o.hd0, o.hd1, o.hd2) << 12);
  const int n = __popc(legal);
  if (n == 0) return 24;
  int k = (int)__umulhi(word, (uint32_t)n), pos = 0;
  // position of the k-th set bit of a 24-bit mask: halving search on popcounts, no loop
  int c = __popc(legal & 0xfffu);
  if (k >= c) k -= c, pos = 12, legal >>= 12;
  c = __popc(legal & 0x3fu);
  if (k >= c) k -= c, pos += 6, legal >>= 6;
  c = __popc(legal & 0x7u);
  if (k >= c) k -= c, pos += 3, legal >>= 3;
  c = (int)(legal & 1u);
  if (k >= c) {
    k -= c, pos += 1;
    c = (int)((legal >> 1) & 1u);
    if (k >= c) pos += 1;
  }
  return pos;
}

// ------------------------------------------------------------------------------------------
// SkyjoGame.act (skyjo.py:308-335) for the expected player, preceded by the legality test of
// TerminateIllegalWrapper (skyjo_env.py:23) on the action mask of skyjo.py:201-224.
// v0..v2: the acting player's vis row (already loaded by the caller for the policy).
// Caller guarantees the game is valid and not done.
// ------------------------------------------------------------------------------------------
// TRUSTED: the action comes from policy_pick, which only ever returns legal actions - no legality test.
// `pendp` != nullptr: the scoring of a finished game is DEFERRED - its card chunks are copied to pendp (chunk of player p
// at pendp + p * 1024) and the finisher is left in pend_fin; the caller scores all such games of the wavefront together
// every few iterations (the float64 arithmetic of one or two lanes is a section the other 62 wait for).
template <bool INDIRECT, int NP, bool TRUSTED>
__device__ __forceinline__ void apply_action(const SkParams &P, uint8_t *lp, uint8_t *fp, uint8_t *ap, HdrRegs &h, uint32_t v0,
                                             uint32_t v1, uint32_t v2, int a, int g, LaneCounters &cnt, Stamps &st,
                                             uint8_t *pendp, int &pend_fin, const uint4 &row_pre, double *racc = nullptr) {
  const int N = P.L.N;
  const int phase = h.w0 & 0xff, p = (h.w0 >> 8) & 0xff;
  const int blk = sk_pb(P.L, p), cardb = blk + PB_CARDS, visb = blk + PB_VIS, pb = P.L.off_pile;
  const unsigned ua = (unsigned)a;
  int slot = 0, sv = 0;
  bool legal;
  if (ua < 24u) {
    slot = a < 12 ? a : a - 12;
    sv = byte3(v0, v1, v2, slot);
    legal = phase == 1 && (TRUSTED || (a < 12 ? sv != SKYJO_REFUNDED : sv == SKYJO_HAND_NONE));
  } else {
    legal = (TRUSTED || ua <= 25u) && phase == 0;  // (a trusted pick is still refused in the wrong phase: a place turn with no slot left)
  }
  if (!legal) {  // offender gets illegal_reward, everybody else 0, all done
    double *rw = P.rewards + (size_t)g * N;
    for (int q = 0; q < N; q++) rw[q] = q == p ? P.illegal_reward : 0.0;
    if (NP > 0 && racc) {
#pragma unroll
      for (int q = 0; q < (NP > 0 ? NP : 1); q++)
        racc[NP + q] += q == p ? P.illegal_reward : 0.0, racc[2 * NP + q] += q == p ? P.illegal_reward * P.illegal_reward : 0.0;
    } else {
      acc_add(ap, N + p, P.illegal_reward), acc_add(ap, 2 * N + p, P.illegal_reward * P.illegal_reward);
    }
    h.w0 = (h.w0 & 0x0000ffffu) | ((((h.w0 >> 16) & 0xffu) | F_DONE) << 16) | ((uint32_t)SKYJO_ST_ILLEGAL << 24);
    P.done[g] = 1;
    cnt.illegal++;
    return;
  }
  const int eplen = (int)(h.w2 & 0xffffu) + 1;
  h.w2 = (h.w2 & 0xffff0000u) | (uint32_t)eplen;
  h.w0 &= 0x00ffffffu;  // status OK
  cnt.steps++;
  if (phase == 0) {
    // _action_draw_card (skyjo.py:337-374): goal check first, on the drawing player.  The bytes either kind of
    // draw could need are requested together with the goal test's hidden count: one LDS round trip.
    const int role = (h.w1 >> 16) & 1;
    int nd = h.w1 & 0xff;
    const int ns = (h.w1 >> 8) & 0xff;
    const int hidden_p = (int)((row_pre.w >> 16) & 0xffu);
    int pile_top = LI(pb + pile_addr(role, nd > 0 ? nd - 1 : 0));
    const int disc_top = LI(pb + pile_addr(role ^ 1, ns > 0 ? ns - 1 : 0));
    const int disc_below = LI(pb + pile_addr(role ^ 1, ns > 1 ? ns - 2 : 0));

    if (hidden_p == 0) {
      h.w0 |= (uint32_t)(F_TERMINATED | F_DONE) << 16;
      LB(H_FINISHER) = (uint8_t)p;
      if (NP > 0 && NP < 8) {
        constexpr int NQ = (NP > 0 && NP < 8) ? NP : 1;
        P.done[g] = 1;
        if (pendp) {
#pragma unroll
          for (int q = 0; q < NQ; q++) *(uint4 *)(pendp + q * 1024) = LQ(sk_pb(P.L, q) >> 4);
          pend_fin = p;
        } else {
          finish_game_fixed<NQ>(P, lp + (P.L.off_players >> 4) * 1024, 2048, ap, g, p, racc);
        }
      } else {
        finish_game(P, lp, fp, ap, g, p);
      }
      cnt.episodes++;
      cnt.sum_len += eplen;
#ifdef SK_STAMPS_FINE
      STAMP(3);
#endif
      return;  // nothing drawn, turn not advanced (skyjo.py:350-356)
    }
    const bool from_pile = a == 24;
    if (from_pile && nd == 0) {  // rare: works on the LDS copy of the header
      HDR_FLUSH(h);
      reshuffle_dispatch(P, lp, fp, g);
      HDR_LOAD(h);
      cnt.reshuffles++;
      nd = h.w1 & 0xff;
      pile_top = LI(pb + pile_addr((h.w1 >> 16) & 1, nd - 1));
    }
    int hand;
    if (from_pile) {
      hand = pile_top;
      h.w1 = (h.w1 & 0xffffff00u) | (uint32_t)(nd - 1);
    } else {
      hand = disc_top;
      hist_add(lp, hand, -1);
      const int top = ns > 1 ? disc_below : -3;  // skyjo.py:254
      h.w1 = (h.w1 & 0x00ff00ffu) | ((uint32_t)(ns - 1) << 8) | (((uint32_t)top & 0xffu) << 24);
    }
    h.w2 = (h.w2 & 0x00ffffffu) | (((uint32_t)hand & 0xffu) << 24);
    h.w0 = (h.w0 & 0xffffff00u) | 1u;  // phase = place
#ifdef SK_STAMPS_FINE
    STAMP(3);
#endif
    return;
  }
  STAMP_N(4);
  // _action_place (skyjo.py:376-427)
  const int hand = (int)(int8_t)(h.w2 >> 24);
  const int reg = ((h.w1 >> 16) & 1) ^ 1;
  int ns = (h.w1 >> 8) & 0xff;
  const uint4 row = row_pre;  // the acting player's cards and his counters: requested before the policy picked
  const uint32_t c0 = row.x, c1 = row.y, c2 = row.z;
  int sum = (int)(int16_t)(row.w & 0xffffu), hid = (int)((row.w >> 16) & 0xffu), refunded = (int)(row.w >> 24);
  // minima over the OTHER players do not change in this turn (skyjo.py:182-183)
  int oms = 1 << 20, omh = 1 << 20;
  for (int q = 0; q < N; q++) {
    const uint32_t cq = LW((sk_pb(P.L, q) + PB_SUM) >> 2);
    const int s = (int)(int16_t)(cq & 0xffffu), hq = (int)((cq >> 16) & 0xffu);
    oms = (q != p && s < oms) ? s : oms, omh = (q != p && hq < omh) ? hq : omh;
  }
#ifdef SK_STAMPS_FINE
  STAMP(5);
#endif
  // one straight-line update for both kinds of place action:
  //   a < 12 : the hand card takes slot a, the card that lay there (open or hidden) goes to the discard pile
  //   a >= 12: the hand card goes to the discard pile, slot a - 12 is revealed (its card value stays)
  const bool swap = a < 12, was_hidden = sv == SKYJO_HAND_NONE;
  const int under = byte3(c0, c1, c2, slot);  // the true card in the slot
  const int shown = swap ? hand : under, gone = swap ? under : hand;
  LB(pb + pile_addr(reg, ns)) = (uint8_t)gone;
  ns++;
  hist_add(lp, gone, 1);
  LB(cardb + slot) = (uint8_t)shown;
  LB(visb + slot) = (uint8_t)shown;
  put3(v0, v1, v2, slot, shown);
  sum += shown - (was_hidden ? 0 : under);
  hid -= was_hidden ? 1 : 0;
  if (!INDIRECT) {
    if (!was_hidden) hist_add(lp, under, -1);  // an open card leaves the table (skyjo.py:240-244)
    hist_add(lp, shown, 1);
  }
  int top = gone;
  // _remask_refunded_player_cards_jit (skyjo.py:431-469): all 4 columns of the acting player, every place action.
  // A column is complete when its three visible bytes are equal and neither hidden nor refunded; the test is
  // branch-free and the (rare) collapse itself sits behind one branch.
  {
    const uint32_t tri[4] = {v0 & 0xffffffu, (v0 >> 24) | ((v1 & 0xffffu) << 8), (v1 >> 16) | ((v2 & 0xffu) << 16), v2 >> 8};
    bool full[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint32_t b0 = tri[c] & 0xffu;
      full[c] = ((tri[c] ^ (tri[c] >> 8)) & 0xffffu) == 0 && b0 != (uint32_t)SKYJO_HAND_NONE && b0 != ((uint32_t)SKYJO_REFUNDED & 0xffu);
    }
    if (full[0] | full[1] | full[2] | full[3]) {
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (full[c]) {
          const int t0 = (int)(int8_t)(tri[c] & 0xff);
          for (int k = 0; k < 3; k++) {
            LB(cardb + 3 * c + k) = (uint8_t)(int8_t)SKYJO_REFUNDED;
            LB(visb + 3 * c + k) = (uint8_t)(int8_t)SKYJO_REFUNDED;
            // skyjo.py:454-458: the slice appended to the discard pile is the zeroed MASK -> three 0s
            LB(pb + pile_addr(reg, ns)) = 0;
            ns++;
          }
          hist_add(lp, 0, 3);
          if (!INDIRECT) hist_add(lp, t0, -3);
          sum -= 3 * t0;
          top = 0;
        }
      refunded++;  // +1 per action, not per column (skyjo.py:418-419)
    }
  }
  // sum / hidden / refunded of the acting player go back as one word
  LW((blk + PB_SUM) >> 2) = ((uint32_t)sum & 0xffffu) | ((uint32_t)hid << 16) | ((uint32_t)refunded << 24);
  // num_placed[p]++ (skyjo.py:424) as a fire-and-forget add on the dword that holds the u16: no read, no wait
  __hip_atomic_fetch_add((uint32_t *)(lp + LIDX(blk + PB_PLACED)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  const int ms = sum < oms ? sum : oms, mh = hid < omh ? hid : omh;
  LB(H_MINSUM) = (uint8_t)(int8_t)(ms < 127 ? ms : 127);
  LB(H_MINHID) = (uint8_t)mh;
  h.w1 = (h.w1 & 0x00ff00ffu) | ((uint32_t)ns << 8) | (((uint32_t)top & 0xffu) << 24);
  h.w2 = (h.w2 & 0x00ffffffu) | ((uint32_t)SKYJO_HAND_NONE << 24);
  const int np = p + 1 == N ? 0 : p + 1;  // skyjo.py:114-120,142-144
  h.w0 = (h.w0 & 0xffff0000u) | ((uint32_t)np << 8);  // phase = draw
}
