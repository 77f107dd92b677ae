// skyjo_step.h - part of skyjo_device.h (included from there, in its place: the parts build on each other in that order).
// Step_body (the fused lockstep loop), the pipelined dealing protocol it speaks, k_step / k_observe / k_reset.
#pragma once
#ifndef SKYJO_DEVICE_PARTS
#error "include skyjo_device.h"
#endif

// ------------------------------------------------------------------------------------------
// k_step: `iters` lockstep iterations over all tiles.  POLICY=false: one iteration with the
// caller's actions (SimpleSkyjoEnv.step); POLICY=true: on-device random admissible policy.
// ------------------------------------------------------------------------------------------
// NP > 0 fixes the player count at compile time (2, 3 and 4 are instantiated): every record offset becomes
// an immediate and the per-player loops unroll; NP == 0 is the generic kernel for any 1..12 players.
// SKYJO_ACTION_SKIP as a caller action leaves the game exactly as it is (no step, no reset; its record is still
// written): that is how the single-game views step ONE game of a shared engine.
// ------------------------------------------------------------------------------------------
// Pipelined dealing beside the step wavefronts - between two kernels on two streams, or between the wavefronts of one k_cycle
// workgroup, the protocol is the same: the step kernel does the bank bookkeeping
// of its own games itself - lane = game - so that a dealing cycle is ONE launch on the caller's stream and nothing on that
// stream ever waits for the dealing stream:
//   on the way out of the launch after which a run is due   sk_plan_deals    what k_scan does, minus the work list: the
//        slot to fill, its episode index and the run's id go into per-game words, the game is marked busy;
//   [dealing stream, behind an event for that launch]        k_deal, mode 3   lane = game again: deals the planned slot,
//        releases its stores and sets done_flag = the run's id;
//   on the way into every later launch                        sk_publish_deals a busy game whose done_flag carries its
//        plan's id is acquired and its slot marked ready (what k_publish does).  A deal that is not finished yet
//        stays busy and is looked at again by the next launch; the rare paths that need a busy game's stream wait
//        for exactly that deal as before (wait_deal_done).
// All bank bookkeeping is still written by the caller's stream only, and only by the lane that owns the game.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void sk_publish_deals(const SkParams &P, int g) {
  const size_t G = (size_t)P.tiles * SK_TILE;
  const int b = P.busy[g];
  if (b) {
    const uint32_t tag = P.plan_tag[g];
    const uint32_t f = __hip_atomic_load(&P.done_flag[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((f & 0x7fffffffu) == tag) {
      SK_FENCE_ACQUIRE(P);  // the record the dealing lane released is what a later reset of this game reads
      if (!P.cancel[g] && f == tag) P.spare_ready[(size_t)(b - 1) * G + g] = 1;  // (bit 31: the deal gave itself up)
      P.busy[g] = 0, P.cancel[g] = 0;
    }
  }
}
__device__ __forceinline__ void sk_plan_deals(const SkParams &P, int g, int lane) {
  const size_t G = (size_t)P.tiles * SK_TILE;
  bool need = false, empty = false;
  if (g < P.B) {
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
    need = !busy && r < SK_BANK;
    empty = need && r == 0;
    if (need) {
      P.busy[g] = (uint8_t)(1 + (head + r) % SK_BANK);
      P.cancel[g] = 0;
      P.plan_ep[g] = consumed + (uint32_t)r;
      P.plan_tag[g] = P.plan_new_tag;
    }
  }
  // Which word: P.be_add.  A k_cycle launch counts ALL its plans - the cycle ends inside it and the one on its way out - into the word
  // of its launch parity, and its first dealing wavefront hands the OTHER word, the previous launch's complete sum, to the host
  // (deal_body): no reader and writer ever share a word, and adapt_interval sees every cycle's empty banks, one launch late
  // (until round 6 only the plan on the way out counted, into the word of its run's parity).
  const unsigned long long be = __ballot(empty);
  if (be && lane == 0) atomicAdd(P.bank_empty + (P.be_add & 1u), (uint32_t)__popcll(be));  // (rare)
}

// Small batches, host-style calls (single-game views): the lane hands its whole game to the host with the records - the
// packed record as it lies in LDS (chunk c at byte 16 c), then rewards[N], scores[N] (float64), the stream position word
// and the done byte - into host-mapped memory, so that skyjo_vec_get_state / get_rewards_host after a *_host call cost
// no device traffic at all (skyjo_capi.hip: raw_valid).
// The sticky device error (SK_ERR_*) goes to the host-mapped word that every synchronising host call looks at - written
// by the kernels behind those calls (k_step with caller actions, k_reset, k_observe), not by the fused rollout kernel.
// EVERY wavefront looks at the device's word on its way out (an atomic load: other CUs set it with atomicOr) and passes a
// raised error on: the wavefront that raises one reaches its own end after the atomicOr, however long it spun - so the very
// host call whose kernel timed out sees the error.  Kernels never clear the host's word (skyjo_vec_seed and
// skyjo_vec_snapshot_restore do).
__device__ __forceinline__ void sk_error_to_host(const SkParams &P, int lane) {
  if (lane == 0) {
    const uint32_t e = __hip_atomic_load(P.dev_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e) P.health_host[2] = e;
  }
}
__device__ __forceinline__ void sk_export_raw(const SkParams &P, uint8_t *lp, int g, uint8_t *o) {
  for (int c = 0; c < P.L.chunks; c++) ((uint4 *)o)[c] = LQ(c);
  double *d = (double *)(o + P.L.state_bytes);
  for (int p = 0; p < P.L.N; p++) d[p] = P.rewards[(size_t)g * P.L.N + p], d[P.L.N + p] = P.scores[(size_t)g * P.L.N + p];
  uint32_t *m = (uint32_t *)(d + 2 * P.L.N);
  m[0] = P.rng_mode == SKYJO_RNG_MT19937 ? (uint32_t)P.mt_idx[g] : 0u;
  m[1] = P.done[g];
}

// The body of the step kernel for ONE wavefront: tile `tile`, its lanes 0..63, its own LDS region `lds_raw` (k_step: the
// workgroup IS that wavefront; k_cycle: four such wavefronts share a workgroup with four dealing wavefronts).
// `cycle_len` != 0 (k_cycle only): the launch spans several dealing cycles of that many iterations - at every cycle end inside the
// launch the wavefront does what the way out of a launch does (publish the run beside it, plan the next) and meets the dealing
// wavefronts of its workgroup at a barrier, after which they deal the run just planned; the tile never leaves LDS.
__device__ __forceinline__ uint32_t sk_next_tag(uint32_t t) {
  t = (t + 1u) & 0x7fffffffu;
  return t ? t : 1u;
}
// PLANAR (the fused rollout of k_cycle): the records leave in the tile-planar layout (include/skyjo_vec.h,
// SKYJO_OPT_RECORD_LAYOUT) - piece p (16 bytes) of lane l's record at  block + p * 1024 + l * 16  of the tile's 4 KiB block - so every
// store instruction writes 1 KiB contiguously STRAIGHT FROM THE REGISTERS the record was assembled in: no LDS staging, no
// read-back, no wait between assembling a record and the next iteration.
template <bool INDIRECT, bool POLICY, int NP, bool PLANAR = false>
__device__ __forceinline__ void step_body(const SkParams &Pin, const int tile, const int lane, uint32_t *lds_raw, const int32_t *actions,
                                          uint8_t *rec_out, int32_t *act_out, int iters, uint64_t policy_seed, uint64_t iter0,
                                          double *end_rew_out, uint8_t *end_out, uint8_t *raw_out, int raw_stride, const int cycle_len = 0,
                                          const bool defer_ok = true) {
  SkParams P = Pin;
  TRACE_DECL;
  if (NP > 0) P.L = sk_make_layout(NP, INDIRECT ? 1 : 0);  // same values as the host computed, now constants
  const int g = tile * SK_TILE + lane;
  const uint32_t lds_tile = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_raw;
  uint8_t *lp = (uint8_t *)lds_raw + lane * 16;
  // LDS map: the tile | one iteration's records (4 KiB for the 64-byte records) | the wavefront's statistics.  The 16-word
  // per-lane scratch of the rare paths (RNG FIFO of a mid-game reshuffle / an in-place deal) ALIASES the record staging
  // area: those paths run inside the step, before this iteration's records are staged and after the previous
  // iteration's were read back (LDS executes a wavefront's accesses in order).
  uint8_t *stg = (uint8_t *)lds_raw + P.L.chunks * 1024;
  uint8_t *fp = stg + lane * 4;
  // (tile-planar records leave from registers: the staging area then only is the rare paths' 4 KiB of scratch, also for the wide
  // records of the direct observation)
  const int stg_bytes = (INDIRECT || PLANAR) ? 4096 : SK_TILE * (P.L.rec_bytes + 16);
  uint8_t *ap = stg + stg_bytes + lane * 8;
  // deferred scoring (fixed player counts under the on-device policy): one card chunk per player and lane.  The kernels with
  // a compile-time player count keep the per-seat statistics in registers (REGACC: fewer LDS atomics, -2 % for the fused
  // rollout, -8 % for a step with caller actions) - their LDS is tile | staging (| card chunks of the deferred scoring),
  // nothing else: 25 KB per wavefront for the fused rollout at three players.
  constexpr bool DEFER = POLICY && NP > 0 && NP < 8;
  constexpr bool REGACC = NP > 0 && NP < 8;  // (every kernel with a compile-time player count)
  constexpr int NACC = REGACC ? SK_ACC_KINDS * NP : 1;
  double racc_store[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) racc_store[k] = 0.0;
  double *racc = REGACC ? racc_store : nullptr;
  // (defer_ok false - k_cycle where four step regions WITH the card chunks of the deferred scoring would overflow a CU's LDS, four
  // players / the direct observation on a full chip: games are scored in the iteration they end, the region is tile + staging)
  uint8_t *pendp = DEFER && defer_ok ? stg + stg_bytes + (REGACC ? 0 : SK_ACC_KINDS * P.L.N * 512) + lane * 16 : nullptr;
  int pend_fin = -1;
  if (!REGACC)
    for (int k = 0; k < SK_ACC_KINDS * P.L.N; k++) ACC(k) = 0.0;
  STAMP_DECL;
  // the tile comes in by LDS-DMA as well (non-temporal: it is read once per launch); a step with caller actions - ONE iteration per
  // launch, the tile's round trip IS the kernel - takes it through registers instead (EXPERIMENTS round 6: an LDS-DMA piece costs the
  // issuing wavefront 100 - 180 cycles and lands slowly; config 5's step kernel 11.7 -> 11.1 us)
  if (!POLICY) tile_load(P, P.state, tile, lane, lp);
  else dma_record<true>((const uint8_t *)P.state, (uint32_t)((((size_t)tile * P.L.chunks) * SK_TILE + lane) * 16), lds_tile, P.L.chunks);
  if (P.ov_flags & 1u) sk_publish_deals(P, g);  // (while the tile is on its way)
  sk_vm_drain();
  HdrRegs h;
  HDR_LOAD(h);
  STAMP(0);
  const bool valid = ((h.w0 >> 16) & F_VALID) != 0;
  LaneCounters cnt;
  uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  const uint64_t gid = P.game_id0 + (uint64_t)g;
  ObsRegs ob;
  if (valid) obs_load(P, lp, (h.w0 >> 8) & 0xff, ob);
  for (int it = 0; it < iters; it++) {
    const uint64_t iter = iter0 + (uint64_t)it;
    if (POLICY && (it == 0 || (iter & 3) == 0)) {
      philox4x32_10((uint32_t)(iter >> 2), (uint32_t)gid, (uint32_t)(gid >> 32), 0x504F4C00u, (uint32_t)policy_seed,
                    (uint32_t)(policy_seed >> 32), r0, r1, r2, r3);
      for (int k = (int)(iter & 3); k > 0; k--) {  // (a launch may start inside a block of four: r0 is always the word of this iteration)
        const uint32_t t = r0;
        r0 = r1, r1 = r2, r2 = r3, r3 = t;
      }
    }
    STAMP_T(3);
    const uint32_t word = r0;
    if (POLICY) {  // next iteration's word moves up (a select on the iteration number compiles to three scalar branches)
      r0 = r1, r1 = r2, r2 = r3, r3 = word;
    }
    int a = -1;
    if (valid) {
      const bool over = ((h.w0 >> 16) & F_DONE) != 0;
      if (!POLICY) a = actions[g];
      const bool skip = !POLICY && a == SKYJO_ACTION_SKIP;
      const bool acted = !over && !skip;  // this iteration applies (or refuses) an action of this game
      SpareRegs sp;
      const bool resetting = over && P.auto_reset && !skip;
      if (resetting) spare_issue(P, lp, lds_tile, tile, lane, g, sp);  // lands while the live games step
      STAMP_T(4);
      if (!over && !skip) {
        const uint32_t v0 = ob.q0, v1 = ob.q1, v2 = ob.q2;  // the acting player's row, read for the previous record
        STAMP_N(2);
        // the acting player's card chunk (cards, sum, hidden, refunded) is on its way while the policy picks
        const uint4 row_pre = LQ(sk_pb(P.L, (h.w0 >> 8) & 0xff) >> 4);
        asm volatile("" ::: "memory");  // (the request stays up here: the compiler would sink it to its first use)
        if (POLICY) a = policy_pick(h.w0 & 0xff, ob, word);
#ifndef SK_STAMPS_FINE
        STAMP_N(3);
#endif
        apply_action<INDIRECT, NP, POLICY>(P, lp, fp, ap, h, v0, v1, v2, a, g, cnt, st, pendp, pend_fin, row_pre, racc);
#ifdef SK_STAMPS_FINE
        STAMP(6);
#else
        STAMP(5);
#endif
      } else if (!skip) {
        a = -1;
        if (P.auto_reset) {
          bool dealt = true;
          if (!spare_commit(P, lp, g, sp)) {
            dealt = deal_inline(P, lp, fp, g, tile, lane, sp.head);
            cnt.waits++;  // counts the slow-path deals
          }
          HDR_LOAD(h);
          if (SK_OFTEN(dealt)) {
            h.w0 = (h.w0 & 0x00ffffffu) | ((uint32_t)SKYJO_ST_RESET << 24);
            cnt.resets++;
          }  // (else: the dealing launch never came - the record shows done / SKYJO_ST_ERROR, include/skyjo_vec.h)
        } else {
          h.w0 = (h.w0 & 0x00ffffffu) | ((uint32_t)SKYJO_ST_NOOP_DONE << 24);
        }
        STAMP(1);
      } else {
        a = -1;
      }
      // byte D of the record: a caller's action outside 0 .. 25 (refused: status ILLEGAL) is written as -2, so that it can
      // neither read as "none" (-1) nor alias a legal action
      if (!POLICY && acted && (unsigned)a >= (unsigned)SKYJO_NUM_ACTIONS) a = -2;
      if (!POLICY && end_out) {
        // rollout collection: the lane that ends an episode - by its natural end or by an illegal move - says so
        // and hands out the final rewards of skyjo_env.py:293-312 it has just computed; zeros everywhere else.  A game that is
        // only re-dealt, left alone (SKYJO_ACTION_SKIP) or already over does not end anything.
        const bool end = acted && ((h.w0 >> 16) & F_DONE) != 0;
        end_out[g] = end ? 1 : 0;
        for (int q = 0; q < P.L.N; q++) end_rew_out[(size_t)g * P.L.N + q] = end ? P.rewards[(size_t)g * P.L.N + q] : 0.0;
      }
      // One read of the expected player's row serves this record and the next iteration's turn.  A draw leaves both the
      // player and his row as they were (the phase is 1 after an applied draw, 0 after a place, a reset or the final draw).
      // The expected player's row serves this record and the next iteration's turn; the record's own two reads (the
      // histogram chunk and bin 14) go out with it: ONE LDS round trip for the whole record.  (The row is re-read even after
      // a draw, which leaves it as it was: skipped in a branch, its wait sits inside the branch and the record's reads behind
      // it - k_step 121 -> 117 us.)
      const uint4 rec_a = LQ(1);
      const uint32_t rec_b = LB(32);
      const uint4 rec_row = LQ((sk_pb(P.L, (h.w0 >> 8) & 0xff) + PB_VIS) >> 4);
      asm volatile("" ::: "memory");  // (all three requests stay up here: the compiler would sink the record's two into the record's branch)
      obs_from_row(rec_row, ob);
      if (rec_out) {
        if (INDIRECT) {  // staged in LDS, written by the whole wavefront below
          uint4 rr[4];
          emit_record<INDIRECT>(P, lp, h, ob, a, nullptr, rr, &rec_a, rec_b);
          if (PLANAR) {  // (a partial last tile: the lanes without a game are switched off here, their slots stay as they were)
            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
            uint8_t *blk = rec_out + ((size_t)it * P.tiles + (size_t)tile) * (SK_TILE * 64) + lane * 16;
#pragma unroll
            for (int p = 0; p < 4; p++) __builtin_nontemporal_store((u32x4_t){rr[p].x, rr[p].y, rr[p].z, rr[p].w}, (u32x4_t *)(blk + p * 1024));
          } else {
#pragma unroll
            for (int p = 0; p < 4; p++) *(uint4 *)(stg + lane * 64 + ((p + (lane >> 1)) & 3) * 16) = rr[p];
          }
        } else if (PLANAR) {  // direct observation, tile-planar: rec_bytes / 16 pieces (5 / 6 / 7), assembled in registers
          constexpr int PIECES = NP > 0 ? (((19 + 12 * NP + 3) & ~3) + 32 + 15) / 16 : 1;
          uint32_t wbuf[4 * PIECES];
#pragma unroll
          for (int k = 0; k < 4 * PIECES; k++) wbuf[k] = 0u;
          emit_record<INDIRECT>(P, lp, h, ob, a, (uint8_t *)wbuf, nullptr, &rec_a, rec_b);
          typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
          uint8_t *blk = rec_out + ((size_t)it * P.tiles + (size_t)tile) * (SK_TILE * 16 * PIECES) + lane * 16;
#pragma unroll
          for (int p = 0; p < PIECES; p++)
            __builtin_nontemporal_store((u32x4_t){wbuf[4 * p], wbuf[4 * p + 1], wbuf[4 * p + 2], wbuf[4 * p + 3]}, (u32x4_t *)(blk + p * 1024));
        } else {  // direct observation: rec_bytes = 80 / 96 / 112 ...; staged record-major with a stride that spreads the banks
          emit_record<INDIRECT>(P, lp, h, ob, a, stg + lane * sk_stage_stride(P.L.rec_bytes), nullptr, &rec_a, rec_b);
        }
      }
      if (act_out) __builtin_nontemporal_store(a, &act_out[(size_t)it * P.B + g]);
#ifdef SK_STAMPS_FINE
      STAMP(7);
#else
      STAMP(6);
#endif
    }
    if (DEFER && ((it & (SK_SCORE_EVERY - 1)) == SK_SCORE_EVERY - 1 || it == iters - 1)) {
      // Scores, rewards and statistics of the games that ended since the last service point (skyjo.py:477-498,
      // skyjo_env.py:293-312), all lanes of the wavefront in one section.  A game cannot end twice in between (an episode
      // is far longer than SK_SCORE_EVERY iterations and the on-device policy makes no illegal move), and the launch does
      // not end before its last service point - the host never sees an unscored finished game.
      if (SK_RARE(__any(pend_fin >= 0))) {
        if (pend_fin >= 0) {
          finish_game_fixed<(NP > 0 && NP < 8) ? NP : 1>(P, pendp, 1024, ap, g, pend_fin, racc);
          pend_fin = -1;
        }
      }
    }
    if (!INDIRECT && !PLANAR && rec_out) {  // same idea for the wider records of the direct observation: rec_bytes / 16 pieces each
      typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
      const int pieces = P.L.rec_bytes >> 4, stride = sk_stage_stride(P.L.rec_bytes);
      uint8_t *blk = rec_out + ((size_t)it * P.B + (size_t)tile * SK_TILE) * (size_t)P.L.rec_bytes;
      const int live = P.B - tile * SK_TILE;
      // piece q of the tile's block: record q / pieces, piece q % pieces.  Up to four pieces are read back together and
      // leave behind a wavefront-uniform branch (see the 64-byte records below: one LDS round trip, not one per piece)
      for (int q0 = lane; q0 < pieces * SK_TILE; q0 += 4 * SK_TILE) {
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int q = q0 + j * SK_TILE, r = q / pieces, p = q - r * pieces;
          if (q < pieces * SK_TILE) v[j] = *(const uint4 *)(stg + r * stride + p * 16);
        }
        if (SK_OFTEN(live >= SK_TILE)) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int q = q0 + j * SK_TILE;
            if (q0 - lane + j * SK_TILE < pieces * SK_TILE)  // (uniform: whole rows of 64 pieces)
              __builtin_nontemporal_store((u32x4_t){v[j].x, v[j].y, v[j].z, v[j].w}, (u32x4_t *)(blk + (size_t)q * 16));
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int q = q0 + j * SK_TILE, r = q / pieces;
            if (q < pieces * SK_TILE && r < live)
              __builtin_nontemporal_store((u32x4_t){v[j].x, v[j].y, v[j].z, v[j].w}, (u32x4_t *)(blk + (size_t)q * 16));
          }
        }
      }
    }
    if (INDIRECT && !PLANAR && rec_out) {
      // The 64 records of the tile are one contiguous 4 KiB block of the output.  They pass through LDS so that each
      // store instruction writes 1 KiB of it contiguously - full lines, one request per 64 bytes, instead of 64 pieces
      // of 16 bytes at a 64-byte stride - and they are written non-temporally: the records are a stream nobody on
      // this chip reads back, and kept out of the memory-side cache they leave the dealing kernel's generator state
      // in it (k_step -7 %, k_deal -12 % together).  Staging slot of (record r, piece p): r * 64 + ((p + (r >> 1)) & 3)
      // * 16 - both the lane-per-record writes above and the lane-per-16-bytes reads here are bank-conflict free.
      typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
      uint8_t *blk = rec_out + ((size_t)it * P.B + (size_t)tile * SK_TILE) * 64;
      const int live = P.B - tile * SK_TILE;  // records of this tile that exist (the last tile may be partial)
      // all four pieces are requested before the first is used: ONE LDS round trip (guarded one by one, each read sat
      // behind its own wait inside its own exec-masked block: four round trips and eight branches per iteration)
      uint4 v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int r = 16 * j + (lane >> 2), p = lane & 3;
        v[j] = *(const uint4 *)(stg + r * 64 + ((p + (r >> 1)) & 3) * 16);
      }
      if (SK_OFTEN(live >= SK_TILE)) {  // (wavefront-uniform: a scalar branch)
#pragma unroll
        for (int j = 0; j < 4; j++)
          __builtin_nontemporal_store((u32x4_t){v[j].x, v[j].y, v[j].z, v[j].w}, (u32x4_t *)(blk + j * 1024 + lane * 16));
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (16 * j + (lane >> 2) < live)
            __builtin_nontemporal_store((u32x4_t){v[j].x, v[j].y, v[j].z, v[j].w}, (u32x4_t *)(blk + j * 1024 + lane * 16));
      }
    }
    STAMP_T(2);
    if (POLICY && cycle_len && it + 1 < iters && (it + 1) % cycle_len == 0) {  // (wavefront-uniform) a dealing cycle ends inside the launch
      if (P.busy[g]) (void)wait_deal_done(P, g);
      sk_publish_deals(P, g);
      sk_plan_deals(P, g, lane);
      P.plan_new_tag = sk_next_tag(P.plan_new_tag);
      TRACE_WAIT_BEGIN;
      __syncthreads();  // k_cycle: the dealing wavefronts of this workgroup take the run just planned from here
      TRACE_WAIT_END;
    }
  }
  HDR_FLUSH(h);
  tile_store_nt(P, P.state, tile, lane, lp);
  if (!POLICY && raw_out && valid) sk_export_raw(P, lp, g, raw_out + (size_t)g * raw_stride);
  if (P.ov_flags & 2u) {
    // (first what the run beside THIS launch has dealt: a game whose busy mark outlived the launch in which it is dealt would
    // get a new episode only every second run.  The dealing kernel was started before this launch and is as good as through:
    // a lane whose deal is still under way waits for it - the one place where this stream waits for the other, and only
    // for as long as the dealing kernel really needs beyond this launch.)
    if (P.ov_flags & 1u) {
      if (P.busy[g]) (void)wait_deal_done(P, g);  // (after a timeout the deal stays busy: sk_publish_deals looks at its flag again)
      sk_publish_deals(P, g);
    }
    sk_plan_deals(P, g, lane);
  }
  // (after the way out: its wait for the run beside this launch can be what raises the error - ADVICE r3's test found a host call
  // that came back clean with the word already set)
  if (!POLICY) {
    sk_error_to_host(P, lane);
    if (P.host_seq) {
      // ONE tile = this wavefront is the whole launch: everything the host reads back (records, exported games, error word)
      // has been stored by it - make that visible system-wide, then tell the host, which is spinning on the word instead of
      // paying for a stream synchronisation (skyjo_vec_step_host)
      __threadfence_system();
      if (lane == 0) __hip_atomic_store(&P.health_host[3], P.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // per-wavefront event counts go to the tile's own slot: thousands of same-address atomics at the
  // end of a launch would serialise at ~12 ns each (MI355X_MICROARCH.md, "fanin")
  uint32_t v[7] = {cnt.steps, cnt.episodes, cnt.illegal, cnt.resets, cnt.sum_len, cnt.reshuffles, cnt.waits};
#pragma unroll
  for (int k = 0; k < 7; k++)
    for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
  if (lane == 0) {
    unsigned long long *c = P.tile_counters + (size_t)tile * 8;
#pragma unroll
    for (int k = 0; k < 7; k++) c[k] += v[k];
  }
  // per-seat statistics of this launch: one slot per tile
  if (__any(cnt.episodes | cnt.illegal)) {
    double mine = 0.0;
    if (REGACC) {
#pragma unroll
      for (int k = 0; k < NACC; k++) {
        double x = racc_store[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        x = __shfl(x, 0, 64);
        mine = lane == k ? x : mine;
      }
    } else {
      for (int k = 0; k < SK_ACC_KINDS * P.L.N; k++) {
        double x = ACC(k);
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        x = __shfl(x, 0, 64);
        mine = lane == k ? x : mine;
      }
    }
    if (lane < SK_ACC_KINDS * P.L.N)  // lane = kind * N + seat -> slot kind * 12 + seat of the tile
      P.acc_tile[(size_t)tile * SK_ACC_KINDS * SKYJO_MAX_PLAYERS + (lane / P.L.N) * SKYJO_MAX_PLAYERS + lane % P.L.N] += mine;
  }
  STAMP(7);
  STAMP_STORE;
  TRACE_STORE(0, (uint32_t)(iter0 / (uint64_t)(iters > 0 ? iters : 1)), lane, tile);
}

template <bool INDIRECT, bool POLICY, int NP, bool PLANAR = false>
__global__ __launch_bounds__(SK_TILE) void k_step(SkParams Pin, const int32_t *actions, uint8_t *rec_out,
                                                  int32_t *act_out, int iters, uint64_t policy_seed, uint64_t iter0,
                                                  double *end_rew_out, uint8_t *end_out, uint8_t *raw_out, int raw_stride) {
  extern __shared__ uint32_t lds_raw[];
  step_body<INDIRECT, POLICY, NP, PLANAR>(Pin, (int)blockIdx.x, (int)threadIdx.x, lds_raw, actions, rec_out, act_out, iters, policy_seed, iter0,
                                          end_rew_out, end_out, raw_out, raw_stride);
}

// One record of the indirect observation into rec_out in either layout (k_observe, k_reset: `planar` - SKYJO_REC_TILE_PLANAR_ALL -
// puts piece p of lane l's record at tile block + p * 1024 + l * 16, like the fused rollout's).
template <bool INDIRECT>
__device__ __forceinline__ void sk_emit_to(const SkParams &P, uint8_t *lp, const HdrRegs &h, const ObsRegs &ob, uint8_t *rec_out, int tile, int lane,
                                           int planar) {
  const int g = tile * SK_TILE + lane;
  if (INDIRECT && planar) {
    uint4 rr[4];
    emit_record<INDIRECT>(P, lp, h, ob, -1, nullptr, rr);
    uint4 *o = (uint4 *)(rec_out + (size_t)tile * SK_TILE * P.L.rec_bytes) + lane;
#pragma unroll
    for (int k = 0; k < 4; k++) o[k * SK_TILE] = rr[k];
  } else {
    emit_record<INDIRECT>(P, lp, h, ob, -1, rec_out + (size_t)g * P.L.rec_bytes);
  }
}

// SimpleSkyjoEnv.observe(agent) (skyjo_env.py:199-214) for arbitrary players; state untouched.
template <bool INDIRECT>
__global__ __launch_bounds__(SK_TILE) void k_observe(SkParams P, const int32_t *players, uint8_t *rec_out, int planar) {
  extern __shared__ uint32_t lds_raw[];
  const int tile = blockIdx.x, lane = threadIdx.x, g = tile * SK_TILE + lane;
  uint8_t *lp = (uint8_t *)lds_raw + lane * 16;
  sk_error_to_host(P, lane);
  tile_load(P, P.state, tile, lane, lp);
  if (!(LB(H_FLAGS) & F_VALID)) return;
  HdrRegs h;
  HDR_LOAD(h);
  int q = players ? players[g] : LB(H_PLAYER);
  q = q < 0 ? 0 : (q >= P.L.N ? P.L.N - 1 : q);
  ObsRegs ob;
  obs_load(P, lp, q, ob);
  sk_emit_to<INDIRECT>(P, lp, h, ob, rec_out, tile, lane, planar);
}

// SkyjoGame.reset for the masked games: take the pre-dealt episode.
template <bool INDIRECT>
__global__ __launch_bounds__(SK_TILE) void k_reset(SkParams P, const uint8_t *mask, uint8_t *rec_out, uint8_t *raw_out, int raw_stride, int planar) {
  extern __shared__ uint32_t lds_raw[];
  const int tile = blockIdx.x, lane = threadIdx.x, g = tile * SK_TILE + lane;
  uint8_t *lp = (uint8_t *)lds_raw + lane * 16;
  uint8_t *fp = (uint8_t *)lds_raw + P.L.chunks * 1024 + lane * 4;
  if (g >= P.B) return;
  const bool want = !mask || mask[g];
  bool want_counted = false;  // (a reset whose deal timed out is not one)
  tile_load(P, P.state, tile, lane, lp);
  if (want) {
    const int head = P.bank_head[g] % SK_BANK;
    bool dealt = true;
    if (!consume_spare(P, lp, tile, lane, g, head)) dealt = deal_inline(P, lp, fp, g, tile, lane, head);
    if (dealt) LB(H_STATUS) = SKYJO_ST_RESET;
    want_counted = dealt;
  }
  HdrRegs h;
  HDR_LOAD(h);
  if (rec_out) {
    ObsRegs ob;
    obs_load(P, lp, LB(H_PLAYER), ob);
    sk_emit_to<INDIRECT>(P, lp, h, ob, rec_out, tile, lane, planar);
  }
  if (want) tile_store(P, P.state, tile, lane, lp);
  if (raw_out && (LB(H_FLAGS) & F_VALID)) sk_export_raw(P, lp, g, raw_out + (size_t)g * raw_stride);
  sk_error_to_host(P, lane);
  const unsigned long long wb = __ballot(want_counted);
  if (want_counted && lane == __ffsll((long long)wb) - 1) P.tile_counters[(size_t)tile * 8 + 3] += __popcll(wb);
}
