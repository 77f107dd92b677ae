// skyjo_device.h - gfx950 device code of the vectorised SkyJo environment.
//
// Execution model: one wavefront (64 lanes) owns one TILE of 64 games; lane l owns game l of
// the tile for the whole launch.  The tile's packed state comes HBM -> LDS by LDS-DMA (memory and LDS
// share one chunk-major layout, skyjo_layout.h), every data-dependent byte access of the transition
// (card slots, pile tops, histogram bins) then stays inside the lane's private 16-byte columns, and
// the tile is streamed back once per launch.  No MFMA: the path is integer / control work bounded by
// HBM traffic (DESIGN.md has the byte counts).  What only one or two lanes of a wavefront ever do at a
// time - scoring a finished game, re-dealing it - is where the other 62 wait: the scoring is deferred
// and batched (SK_SCORE_EVERY), the re-deal overlaps its memory round trip with the live lanes' step.
//
// Kernels: k_step (the body below, one wavefront per workgroup: caller-action steps, and the fused rollout of the engines k_cycle
// does not cover), k_deal (the dealing run on its own), and k_cycle - the default for the fused rollout: ONE launch per dealing
// cycle whose workgroups hold step AND dealing wavefronts of the same games, so that a dealt episode changes hands inside a CU
// (workgroup-scope release / acquire instead of an L2 write-back / invalidation per wavefront: see k_cycle).
//
// Diagnostic builds (tools/dev/): -DSK_STAMPS / -DSK_STAMPS_FINE (section cycle counters), -DSK_TRACE (placement, time span and
// clock of every wavefront).  Neither changes a result.  The round-4 timing switches that left pieces out on purpose (wrong
// results) are no longer part of this file: tools/dev/sk_exp_switches.patch puts them back for whoever wants to repeat the
// experiments of EXPERIMENTS.md round 4.
//
// Semantics follow rlskyjo/game/skyjo.py and rlskyjo/environment/skyjo_env.py; each function
// cites the lines it restates.  Nothing here shares code with oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/skyjo_vec.h"
#include "skyjo_layout.h"
#include "skyjo_draw.h"

struct SkCounters {
  unsigned long long steps, episodes, illegal, resets, sum_len, reshuffles, iters, waits;
  double sum_score[SKYJO_MAX_PLAYERS];
  double sum_reward[SKYJO_MAX_PLAYERS];
  double sum_reward_sq[SKYJO_MAX_PLAYERS];
  double sum_refunded[SKYJO_MAX_PLAYERS];
};
#define SK_ERR_DEAL_TIMEOUT 1u  // bits of SkParams.dev_error
// Deferred scoring: lockstep iterations between two service points (a power of two).  It must stay below the shortest
// possible episode - 20 N + 1 steps (ten place turns of the finisher, everybody else in between, the final draw): 41 for
// two players - so that no game can end twice between two service points.  4 / 8 / 16 / 32: k_step 134.7 / 128.3 / 125.2 /
// 123.4 us per 88 iterations (139.8 with every game scored in the iteration it ends).
#ifndef SK_SCORE_EVERY
#define SK_SCORE_EVERY 32
#endif
#define SK_ACC_KINDS 4  // per-seat float64 statistics kept per tile: score, reward, reward^2, refunded

// Pre-dealt episodes per game.  Deeper banks ride out longer gaps between dealing runs, but their records share the
// 256 MB memory-side cache with the 164 MB of generator state the dealing kernel works on: at 65 536 three-player
// games a bank of 4 / 3 / 2 gives 20.9 / 23.6 / 23.9 x 10^9 steps/s (k_deal 105 / 80 / 79 us), none of them ever
// running dry at the default dealing interval.
#ifndef SK_BANK
#define SK_BANK 3
#endif

struct SkParams {
  SkLayout L;
  int32_t B, tiles, rng_mode, auto_reset;
  uint32_t deal_tag;        // id of the dealing launch that may still be running while this kernel runs
  uint32_t spin_log2;       // wait_deal_done gives up after 2^spin_log2 polls (22; lowered by the fault-injection test)
  uint32_t debug_deal_delay;  // fault injection: every dealing wavefront sleeps this many times 127 x 64 cycles before it starts
  double score_penalty, mean_reward, reward_refunded, illegal_reward;
  uint64_t game_id0;
  uint4 *state;             // [tiles][chunks][64] live games
  uint4 *spare;             // [SK_BANK][tiles * 64][chunks] the game's bank of pre-dealt next episodes, GAME-major (bank_rec16)
  uint8_t *spare_ready;     // [SK_BANK][tiles*64]; the ready slots of a game are head, head+1, ... (mod SK_BANK)
  uint8_t *bank_head;       // [tiles*64] slot that is taken next (mirrored in the record header, H_BANK)
  uint8_t *busy;            // [tiles*64] 0, or 1 + slot while the dealing kernel owns the game's stream and that slot
  uint8_t *cancel;          // [tiles*64] the in-flight deal was overtaken (rolled back or taken early): do not publish
  uint32_t *done_flag;      // [tiles*64] dealing-kernel launch id that last finished a deal for the game
  uint32_t *plan_tag;       // [tiles*64] id of the dealing launch that deals (dealt) the game's busy slot
  uint32_t *plan_ep;        // [tiles*64] episode index of that deal (pipelined dealing: see sk_plan_deals)
  uint32_t ov_flags;        // k_step, dealing beside it: 1 = publish finished deals on the way in, 2 = plan the next run on the way out
  uint32_t plan_new_tag;    // the id the planned run will have
  uint32_t wg_local;        // 1 inside k_cycle: the dealing wavefront of a game and its step wavefront share a workgroup (SK_FENCE_*)
  uint32_t host_seq;        // != 0 (single-tile engines, host-style step): the wavefront's last store is this number into health_host[3]
  int32_t *deal_list;       // [2][tiles*64] games of the current / previous dealing launch (k_scan)
  uint32_t *deal_ep;        // [2][tiles*64] episode index of each listed deal
  uint32_t *deal_count;     // [2]
  uint32_t *bank_empty;     // [2] games whose bank held no episode when the last scan looked (early warning of a drain); [1]: see k_deal
  volatile uint32_t *health_host;  // [4] host-mapped: {that count, dealing-run tag} - written once per run, read by the host; [2] = SK_ERR_* (sticky)
  uint32_t *mt;             // [tiles*64][624] numpy-legacy MT19937 state, advanced in place (mt_untwist steps it back)
  int32_t *mt_idx;          // [1+SK_BANK][tiles*64]: [0] stream position (idx | ahead << 16), [1 + slot] position before its deal
  uint64_t *seeds;          // [tiles*64] value given to set_seed
  uint32_t *deals_consumed; // [tiles*64]
  double *rewards;          // [tiles*64][N]
  double *scores;           // [tiles*64][N]
  uint8_t *done;            // [tiles*64]
  double *acc_tile;         // [tiles][SK_ACC_KINDS][12] per-wavefront sums per seat: final score, reward, reward^2, num_refunded
  uint32_t *dev_error;      // [1] sticky: set by a kernel that had to give up (SK_ERR_*), reported by skyjo_vec_get_counters
  SkCounters *counters;
  unsigned long long *tile_counters;  // [tiles][8] per-wavefront event counts (no same-address atomics)
  unsigned long long *stamps;         // [tiles][8] section cycle sums, written only by -DSK_STAMPS diagnostic builds
};

struct LaneCounters {
  uint32_t steps = 0, episodes = 0, illegal = 0, resets = 0, sum_len = 0, reshuffles = 0, waits = 0;
};

// ------------------------------------------------------------------------------------------
// LDS addressing: lp = tile base + lane * 16 ; byte b of this lane's record lives at lp[LIDX(b)]
// (chunk-major, skyjo_layout.h); LQ(c) is the lane's whole 16-byte chunk c (one ds_read_b128 / ds_write_b128).
// ------------------------------------------------------------------------------------------
// Hand-over of a dealt episode from the dealing wavefront (release) to the step wavefront (acquire).  Between two KERNELS
// (the two-stream form) the two may sit on different XCDs, whose L2s are not coherent: agent scope, i.e. buffer_wbl2 /
// buffer_inv sc1 - a write-back / an invalidation of a whole L2 per wavefront, which is what made that form a net loss on a full
// chip (EXPERIMENTS.md round 4: 25.7 -> 34.0 x 10^9 steps/s without them).  Inside k_cycle both are wavefronts of ONE workgroup,
// i.e. of one CU - they share its vector L1 (write-through) and its XCD's L2: workgroup scope is all the memory model asks for,
// and on gfx950 that is a wait for the wavefront's own stores and nothing else (P.wg_local).
#define SK_FENCE_ACQUIRE(P)                                              \
  do {                                                                   \
    if ((P).wg_local) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");              \
  } while (0)
#define SK_FENCE_RELEASE(P)                                              \
  do {                                                                   \
    if ((P).wg_local) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");              \
  } while (0)
#define SK_RARE(x) __builtin_expect(!!(x), 0)
#define SK_OFTEN(x) __builtin_expect(!!(x), 1)
#define LIDX(b) ((((b) >> 4) << 10) | ((b) & 15))
#define LB(b) (lp[LIDX(b)])
#define LI(b) ((int)(int8_t)lp[LIDX(b)])
#define LW(w) (*(uint32_t *)(lp + LIDX(4 * (w))))
#define LH(b) (*(uint16_t *)(lp + LIDX(b)))
#define LSH(b) (*(int16_t *)(lp + LIDX(b)))
#define LQ(c) (*(uint4 *)(lp + ((c) << 10)))

// Groups of 6 chunks: all global loads of a group are issued before the first LDS write, so a tile
// costs ceil(chunks / 6) memory round trips instead of one per chunk (the trip count is a run-time
// value, the compiler does not pipeline this loop by itself).
__device__ __forceinline__ void tile_load(const SkParams &P, const uint4 *src, int tile, int lane, uint8_t *lp) {
  const uint4 *s = src + (size_t)tile * P.L.chunks * SK_TILE + lane;
  const int n = P.L.chunks;
  for (int c = 0; c < n; c += 6) {
    uint4 v[6];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (c + k < n) v[k] = s[(size_t)(c + k) * SK_TILE];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (c + k < n) LQ(c + k) = v[k];
  }
}

__device__ __forceinline__ void tile_store(const SkParams &P, uint4 *dst, int tile, int lane, uint8_t *lp) {
  uint4 *d = dst + (size_t)tile * P.L.chunks * SK_TILE + lane;
  const int n = P.L.chunks;
  for (int c = 0; c < n; c += 6) {
    uint4 v[6];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (c + k < n) v[k] = LQ(c + k);
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (c + k < n) d[(size_t)(c + k) * SK_TILE] = v[k];
  }
}

// k_step's own tile I/O: non-temporal.  The live tiles are read at the start and written at the end of a launch of
// ~140 us; kept out of the memory-side cache they leave it to the generator state and the bank - with them inside,
// about every third process ran its dealing kernel at 95 instead of 80 us (where the driver had put the pages), with
// them outside none of 8 did, at +2 % for k_step.
typedef uint32_t sk_u32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void tile_store_nt(const SkParams &P, uint4 *dst, int tile, int lane, uint8_t *lp) {
  sk_u32x4_nt *d = (sk_u32x4_nt *)(dst + (size_t)tile * P.L.chunks * SK_TILE + lane);
  const int n = P.L.chunks;
  for (int c = 0; c < n; c += 6) {
    uint4 v[6];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (c + k < n) v[k] = LQ(c + k);
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (c + k < n) __builtin_nontemporal_store((sk_u32x4_nt){v[k].x, v[k].y, v[k].z, v[k].w}, d + (size_t)(c + k) * SK_TILE);
  }
}

// ------------------------------------------------------------------------------------------
// LDS-DMA: a record in tile layout (chunk c of this lane at  base + voff + c * 1024  bytes in memory) is requested
// straight into the lane's LDS slot (chunk c at  lds_tile + c * 1024 + lane * 16): global_load_lds_dwordx4 writes
// M0 + offset + lane * 16 and applies its immediate offset to both addresses, so memory layout == LDS layout and a
// record costs one instruction per chunk, no registers and no LDS-write instructions.  Lanes that are switched off
// (EXEC) neither load nor write: a wavefront resets only the lanes whose game has ended, while the others keep playing
// in their own columns of the tile.  The compiler does not count these loads (inline asm, no destination register):
// whoever reads the slot calls sk_vm_drain() first.  M0 is saved and restored around each statement.
// ------------------------------------------------------------------------------------------
#define SK_DMA4(NT)                                                                                                     \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"                                                    \
               "global_load_lds_dwordx4 %1, %2" NT "\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" NT "\n\t"           \
               "global_load_lds_dwordx4 %1, %2 offset:2048" NT "\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072" NT "\n\t" \
               "s_mov_b32 m0, %0"                                                                                       \
               : "=&s"(keep)                                                                                            \
               : "v"(voff), "s"(base), "s"(lds)                                                                         \
               : "memory")
#define SK_DMA1(NT)                                                                                              \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" NT "\n\ts_mov_b32 m0, %0" \
               : "=&s"(keep)                                                                                     \
               : "v"(voff), "s"(base), "s"(lds)                                                                  \
               : "memory")
template <bool NT>
__device__ __forceinline__ void dma_record(const uint8_t *base, uint32_t voff, uint32_t lds, int chunks) {
  uint32_t keep;
  int c = 0;
  for (; c + 4 <= chunks; c += 4) {
    if (NT) SK_DMA4(" nt");
    else SK_DMA4("");
    base += 4096, lds += 4096;
  }
  for (; c < chunks; c++) {
    if (NT) SK_DMA1(" nt");
    else SK_DMA1("");
    base += 1024, lds += 1024;
  }
}
// The BANK of pre-dealt episodes is game-major (round 5): the record of (slot, game) is `chunks` consecutive 16-byte pieces, so that
// the ONE lane that takes it reads three 128-byte lines, every byte of them used - in the tile layout its 18 pieces lay 1 KiB apart,
// 18 lines fetched for 288 bytes: 1.0 GB of the 10.1 GB a launch of 1 024 iterations moved (profiles/r5_hbm_traffic_attribution.json).
// The LDS side is still the lane's column (chunk c at lds_tile + c * 1024 + lane * 16) and the LDS-DMA adds its immediate offset to
// BOTH addresses: chunk c goes out with offset 16 c and M0 = row c's base - 16 c, i.e. M0 moves on by 1024 - 16 per chunk.
#define SK_DMAB4(NT)                                                                                                        \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" NT "\n\t"                \
               "s_add_u32 m0, m0, 0x3f0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:16" NT "\n\t"                   \
               "s_add_u32 m0, m0, 0x3f0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:32" NT "\n\t"                   \
               "s_add_u32 m0, m0, 0x3f0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:48" NT "\n\t"                   \
               "s_mov_b32 m0, %0"                                                                                           \
               : "=&s"(keep)                                                                                                \
               : "v"(voff), "s"(base), "s"(lds)                                                                             \
               : "memory", "scc")
template <bool NT>
__device__ __forceinline__ void dma_bank_record(const uint8_t *base, uint32_t voff, uint32_t lds, int chunks) {
  uint32_t keep;
  int c = 0;
  for (; c + 4 <= chunks; c += 4) {
    if (NT) SK_DMAB4(" nt");
    else SK_DMAB4("");
    base += 64, lds += 4096;
  }
  for (; c < chunks; c++) {
    if (NT) SK_DMA1(" nt");
    else SK_DMA1("");
    base += 16, lds += 1024;
  }
}
// every vector-memory operation this wavefront has issued so far is complete (the LDS-DMA data is in LDS)
__device__ __forceinline__ void sk_vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Diagnostic builds (-DSK_STAMPS) sum s_memtime deltas per section into P.stamps; the shipped build has none.
struct Stamps {
  unsigned long long t, acc[8];
};
#ifdef SK_STAMPS
#define STAMP_DECL Stamps st; st.t = __builtin_amdgcn_s_memtime(); for (int k_ = 0; k_ < 8; k_++) st.acc[k_] = 0
#define STAMP(i)                                                 \
  do {                                                           \
    __builtin_amdgcn_sched_barrier(0);                           \
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0) only: stores stay in flight as in the shipped build */ \
    unsigned long long st_n = __builtin_amdgcn_s_memtime();      \
    __builtin_amdgcn_s_waitcnt(0xc07f);                          \
    st.acc[i] += st_n - st.t;                                    \
    st.t = st_n;                                                 \
    __builtin_amdgcn_sched_barrier(0);                           \
  } while (0)
#define STAMP_STORE                                                        \
  do {                                                                     \
    if (lane == 0)                                                         \
      for (int k = 0; k < 8; k++) P.stamps[(size_t)tile * 8 + k] += st.acc[k]; \
  } while (0)
#else
#define STAMP_DECL Stamps st
#define STAMP(i)
#define STAMP_STORE
#endif
// -DSK_STAMPS_TOP (with -DSK_STAMPS): the step loop's sections cut differently - 2 = read-back + stores + service points at the end
// of an iteration, 3 = the policy's Philox block, 4 = the reset's request (spare_issue), 5 = card row + pick + transition, 6 = record
#ifdef SK_STAMPS_TOP
#define STAMP_N(n)
#define STAMP_T(t) STAMP(t)
#else
#define STAMP_N(n) STAMP(n)
#define STAMP_T(t)
#endif

// Diagnostic builds (-DSK_TRACE): where and when every wavefront ran - {HW_ID, XCC_ID, start, end (100 MHz real-time clock),
// tag} per wavefront in P.stamps, two launches deep (slot = tag & 1): k_step rows [slot][tile], k_deal rows [2 + slot][tile]
// (tools/dev/placement.py reads them through skyjo_vec_debug_trace).  The shipped build has none of it.
#ifdef SK_TRACE
#define TRACE_DECL const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime(), trace_c0 = __builtin_amdgcn_s_memtime(); unsigned long long trace_wait = 0, trace_w0 = 0
#define TRACE_WAIT_BEGIN trace_w0 = __builtin_amdgcn_s_memtime()
#define TRACE_WAIT_END trace_wait += __builtin_amdgcn_s_memtime() - trace_w0
#define TRACE_STORE(kind, tag, lane_, block_)                                                                    \
  do {                                                                                                           \
    if ((lane_) == 0) {                                                                                          \
      unsigned long long *tr = P.stamps + (((size_t)(2 * (kind) + ((tag) & 1u)) * P.tiles + (block_)) * 8);     \
      tr[0] = __builtin_amdgcn_s_getreg(63492); /* HW_REG_HW_ID, 32 bits */                                      \
      tr[1] = __builtin_amdgcn_s_getreg(63508); /* HW_REG_XCC_ID */                                              \
      tr[2] = trace_t0, tr[3] = __builtin_amdgcn_s_memrealtime(), tr[4] = (tag);                                 \
      tr[5] = __builtin_amdgcn_s_memtime() - trace_c0; /* shader cycles: / (end - start) x 100 MHz = the clock */ \
      tr[6] = trace_wait; /* cycles spent at the cycle-end barriers of a k_cycle launch */                       \
    }                                                                                                            \
  } while (0)
#else
#define TRACE_DECL
#define TRACE_WAIT_BEGIN
#define TRACE_WAIT_END
#define TRACE_STORE(kind, tag, lane_, block_)
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
__device__ __forceinline__ void sk_plan_deals(const SkParams &P, int g, int lane, const bool report = true) {
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
  // `report` false - the cycle ends INSIDE a k_cycle launch: the dealing wavefront that hands the count to the host (deal_body,
  // piped form) reads and clears the word of its run's parity while other workgroups of the same launch may be planning that
  // very run or one two cycles on - counts would be lost or land on the wrong run.  So only the plan on the way OUT of a launch
  // counts (complete when the next launch's dealing wavefronts look at it): adapt_interval sees one run per launch.
  const unsigned long long be = __ballot(empty && report);
  if (be && lane == 0) atomicAdd(P.bank_empty + (P.plan_new_tag & 1u), (uint32_t)__popcll(be));  // (rare)
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
      sk_plan_deals(P, g, lane, false);
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

template <bool INDIRECT, bool POLICY, int NP>
__global__ __launch_bounds__(SK_TILE) void k_step(SkParams Pin, const int32_t *actions, uint8_t *rec_out,
                                                  int32_t *act_out, int iters, uint64_t policy_seed, uint64_t iter0,
                                                  double *end_rew_out, uint8_t *end_out, uint8_t *raw_out, int raw_stride) {
  extern __shared__ uint32_t lds_raw[];
  step_body<INDIRECT, POLICY, NP>(Pin, (int)blockIdx.x, (int)threadIdx.x, lds_raw, actions, rec_out, act_out, iters, policy_seed, iter0,
                                  end_rew_out, end_out, raw_out, raw_stride);
}

// SimpleSkyjoEnv.observe(agent) (skyjo_env.py:199-214) for arbitrary players; state untouched.
template <bool INDIRECT>
__global__ __launch_bounds__(SK_TILE) void k_observe(SkParams P, const int32_t *players, uint8_t *rec_out) {
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
  emit_record<INDIRECT>(P, lp, h, ob, -1, rec_out + (size_t)g * P.L.rec_bytes);
}

// SkyjoGame.reset for the masked games: take the pre-dealt episode.
template <bool INDIRECT>
__global__ __launch_bounds__(SK_TILE) void k_reset(SkParams P, const uint8_t *mask, uint8_t *rec_out, uint8_t *raw_out, int raw_stride) {
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
    emit_record<INDIRECT>(P, lp, h, ob, -1, rec_out + (size_t)g * P.L.rec_bytes);
  }
  if (want) tile_store(P, P.state, tile, lane, lp);
  if (raw_out && (LB(H_FLAGS) & F_VALID)) sk_export_raw(P, lp, g, raw_out + (size_t)g * raw_stride);
  sk_error_to_host(P, lane);
  const unsigned long long wb = __ballot(want_counted);
  if (want_counted && lane == __ffsll((long long)wb) - 1) P.tile_counters[(size_t)tile * 8 + 3] += __popcll(wb);
}

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
    uint32_t *be = P.bank_empty + (P.deal_tag & 1u);  // the launch that planned this run has finished: its count is complete
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

// ------------------------------------------------------------------------------------------
// k_cycle: one kernel for one to sixteen dealing cycles - stepping and dealing side by side inside every workgroup (fused rollout,
// two to four players, either observation; PLANAR: the indirect observation's records leave tile-planar, SKYJO_OPT_RECORD_LAYOUT).
//
// A workgroup is S step wavefronts + S dealing wavefronts on one CU (S = 4 on a full chip: eight wavefronts, two per SIMD;
// S = 1 .. 3 for batches of up to 256 .. 768 tiles, every wavefront on a SIMD of its own).  The dealing wavefronts deal for the
// games of THEIR OWN workgroup's tiles, the run that the previous launch planned on its way out (the pipelined protocol of
// the two-stream form, unchanged: sk_plan_deals / sk_publish_deals / wait_deal_done; `deal_tag_run` is that run's id, 0 =
// nothing to deal in this launch).  So the hand-over of a dealt episode never leaves the CU: the two wavefronts share its
// vector L1 and its XCD's L2, and workgroup-scope release / acquire - a wait for the wavefront's own stores - is all it
// takes (P.wg_local).  The two-stream form has to use agent scope (the kernels' wavefronts may sit on different XCDs), i.e. a
// write-back / an invalidation of a whole L2 per wavefront: THAT, not the sharing of SIMDs, is what made dealing beside the
// step kernel a net loss on a full chip for three rounds (EXPERIMENTS.md round 4: two-stream form 25.7, the same without its
// cache maintenance 34.0, this kernel 36 - 37 x 10^9 env-steps/s against 30.5 in line).
//
// Roles on a full chip: the first wavefront to arrive on a SIMD steps, the second deals - one of each per SIMD, where they
// hide each other's latencies (two dealing wavefronts on one SIMD saturate its vector ALU: 32.5 instead of 35.8 x 10^9).  Tiles
// are claimed through LDS counters; nothing depends on how the hardware spreads the wavefronts.
// ------------------------------------------------------------------------------------------
#define SK_CYCLE_MAX_S 4
template <bool INDIRECT, int NP, bool PLANAR>
__global__ __launch_bounds__(2 * SK_CYCLE_MAX_S *SK_TILE) void k_cycle(SkParams Pin, uint8_t *rec_out, int32_t *act_out, int iters, uint64_t policy_seed,
                                                                        uint64_t iter0, uint32_t deal_tag_run, uint32_t lds_step_bytes,
                                                                        uint32_t lds_deal_bytes, int cycle_len) {
  extern __shared__ uint32_t lds_raw[];
  const uint32_t S = blockDim.x >> 7;  // step (= dealing) wavefronts per workgroup
  const uint32_t split = lds_deal_bytes >> 30;  // (diagnostic role splits, see below)
  const bool defer_ok = ((lds_deal_bytes >> 29) & 1u) == 0;  // (bit 29: the step regions have no room for deferred scoring)
  lds_deal_bytes &= 0x1fffffffu;
  uint32_t *claim = lds_raw + (((size_t)S * (lds_step_bytes + lds_deal_bytes)) >> 2);  // six words behind the 2 S regions
  const int lane = (int)(threadIdx.x & 63u);
  if (threadIdx.x < 6) claim[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t simd = (__builtin_amdgcn_s_getreg(63492) >> 4) & 3u;  // HW_REG_HW_ID[5:4]
  uint32_t role = 0, slot = 0;
  if (split == 0) {
    if (lane == 0) role = atomicAdd(&claim[2 + simd], 1u) & 1u;
    role = (uint32_t)__builtin_amdgcn_readfirstlane((int)role);
  } else {  // diagnostic: two of a kind per SIMD - by SIMD parity (1) or by SIMD pair (2)
    role = split == 1 ? simd & 1u : (simd >> 1) & 1u;
  }
  if (lane == 0) slot = atomicAdd(&claim[role], 1u);
  slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
  if (slot >= S) {  // the preferred role is taken S times over (always the case for S < 4, where every wavefront asks to step first)
    role ^= 1u;
    if (lane == 0) slot = atomicAdd(&claim[role], 1u);
    slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
  }
  const int unit = (int)(blockIdx.x * S + slot);
  if (unit >= Pin.tiles) {
    // a surplus wavefront of the last workgroup (tiles % S != 0): it has no tile, but it is a member of the workgroup - it meets the
    // others at every cycle-end barrier of a launch of several cycles (step and dealing wavefronts both pass (iters - 1) / cycle_len
    // of them), so that no barrier is ever executed by a part of the workgroup only
    for (int c = cycle_len > 0 ? (iters - 1) / cycle_len : 0; c > 0; c--) __syncthreads();
    return;
  }
  Pin.wg_local = 1u;  // the games of tile `unit` are dealt by dealing slot `slot` of THIS workgroup: hand-overs stay inside the CU
  if (role == 0) {
    step_body<INDIRECT, true, NP, PLANAR>(Pin, unit, lane, lds_raw + (size_t)slot * (lds_step_bytes >> 2), nullptr, rec_out, act_out, iters, policy_seed, iter0,
                                  nullptr, nullptr, nullptr, 0, cycle_len, NP >= 4 ? true : defer_ok);  // (four players: never without - the host does not ask for it)
  } else {
    // the run the previous launch planned (deal_tag_run, 0 = none), then - a launch of several dealing cycles - the runs its step
    // wavefronts plan at the cycle ends inside it: ids plan_new_tag, + 1, ... (the last one is left to the next launch)
    SkParams P = Pin;
    const int cycles = cycle_len > 0 ? (iters + cycle_len - 1) / cycle_len : 1;
    uint32_t tag = deal_tag_run, planned = Pin.plan_new_tag;
    for (int c = 0; c < cycles; c++) {
      if (tag) {
        P.deal_tag = tag;
        deal_body<NP>(P, 0, 3, unit, lane, lds_raw + (((size_t)S * lds_step_bytes + (size_t)slot * lds_deal_bytes) >> 2), c == 0);
      }
      if (c + 1 < cycles) {
#ifdef SK_TRACE
        const unsigned long long dw0 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();  // (the step wavefronts arrive when they have planned the next run)
#ifdef SK_TRACE
        if (lane == 0) P.stamps[((size_t)2 * P.tiles + unit) * 8 + 7] += __builtin_amdgcn_s_memtime() - dw0, P.stamps[((size_t)3 * P.tiles + unit) * 8 + 7] = 0;
#endif
        tag = planned, planned = sk_next_tag(planned);
      }
    }
  }
}

// per-seat sums over all games for skyjo_vec_get_counters (the hot path keeps per-game sums, no atomics)
__global__ void k_reduce_stats(SkParams P) {
  for (int kind = 0; kind < SK_ACC_KINDS; kind++)
    for (int p = 0; p < P.L.N; p++) {
      double a = 0.0;
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.tiles; i += gridDim.x * blockDim.x)
        a += P.acc_tile[((size_t)i * SK_ACC_KINDS + kind) * SKYJO_MAX_PLAYERS + p];
      for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
      if ((threadIdx.x & 63) == 0 && a != 0.0) atomicAdd(&P.counters->sum_score[kind * SKYJO_MAX_PLAYERS + p], a);  // the four arrays are contiguous
    }
  // order of SkCounters' leading fields: steps, episodes, illegal, resets, sum_len, reshuffles, iters, waits
  const int dst[7] = {0, 1, 2, 3, 4, 5, 7};
  for (int k = 0; k < 7; k++) {
    unsigned long long t = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.tiles; i += gridDim.x * blockDim.x)
      t += P.tile_counters[(size_t)i * 8 + k];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if ((threadIdx.x & 63) == 0 && t) atomicAdd(&(&P.counters->steps)[dst[k]], t);
  }
}

// ------------------------------------------------------------------------------------------
// Config 5 caller piece: TorchActionMaskModel.forward's masking (rlskyjo/models/action_mask_model.py:58-74) and the
// categorical draw, one lane per game.  256 games per block: their 256 x 26 logits are one contiguous 26.6 KB
// stretch that is copied to LDS with coalesced 16-byte loads; a lane then walks its own row (stride 26 words:
// two lanes per bank).  The mask bytes come straight out of the engine's records.
// ------------------------------------------------------------------------------------------
// Byte k of record r in either layout.  `planar`: the records lie tile-planar
// (SKYJO_REC_TILE_PLANAR: byte k of record r at  (r / 64) * 64 * rec_bytes + (k / 16) * 1024 + (r % 64) * 16 + k % 16).
__device__ __forceinline__ const uint8_t *sk_rec_byte(const uint8_t *rec, long long r, int k, int rec_bytes, int planar) {
  return planar ? rec + (r >> 6) * (64LL * rec_bytes) + (long long)(k >> 4) * 1024 + (r & 63) * 16 + (k & 15) : rec + r * rec_bytes + k;
}
#define SK_SAMPLE_BLOCK 256
__global__ __launch_bounds__(SK_SAMPLE_BLOCK) void k_sample(SkLayout L, const uint8_t *rec, const float *logits, long long n,
                                                            uint64_t seed, uint64_t ticket, uint64_t game_id0, int no_masking,
                                                            int32_t *actions, float *logp, float *uniform, int planar) {
  __shared__ float rows[SK_SAMPLE_BLOCK * SKYJO_NUM_ACTIONS];
  const long long g0 = (long long)blockIdx.x * SK_SAMPLE_BLOCK;
  const int nb = (int)(n - g0 < SK_SAMPLE_BLOCK ? n - g0 : SK_SAMPLE_BLOCK);
  const float *src = logits + g0 * SKYJO_NUM_ACTIONS;  // (256 * 26 * 4 bytes per block: 16-byte aligned)
  const int words = nb * SKYJO_NUM_ACTIONS;
  for (int w = threadIdx.x * 4; w < words; w += SK_SAMPLE_BLOCK * 4) {
    if (w + 4 <= words) {
      const float4 v = *(const float4 *)(src + w);
      rows[w] = v.x, rows[w + 1] = v.y, rows[w + 2] = v.z, rows[w + 3] = v.w;
    } else {
      for (int k = w; k < words; k++) rows[k] = src[k];
    }
  }
  __syncthreads();
  if ((int)threadIdx.x >= nb) return;
  const long long g = g0 + threadIdx.x;
  uint32_t mw[7];  // 26 mask bytes from offset Dp (4-byte aligned: a word never straddles two 16-byte pieces)
#pragma unroll
  for (int k = 0; k < 7; k++) mw[k] = *(const uint32_t *)sk_rec_byte(rec, g, L.Dp + 4 * k, L.rec_bytes, planar);
  const float *row = rows + threadIdx.x * SKYJO_NUM_ACTIONS;
  float lp_ = 0.f, u_ = 0.f;
  actions[g] = sk_draw_action(row, mw, no_masking, seed, ticket, game_id0 + (uint64_t)g, logp ? &lp_ : nullptr, &u_);
  if (logp) logp[g] = lp_;
  if (uniform) uniform[g] = u_;
}

// Rollout collection: from the records a step has just written, mark the games whose episode ended in that
// step and copy their final rewards (skyjo_env.py:293-312) - zeros elsewhere.  (skyjo_vec_step_collect has the step kernel
// do the same on its way: no extra launch.)
__global__ void k_episode_ends(SkParams P, const uint8_t *rec, double *rew_out, uint8_t *end_out, int planar) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P.B) return;
  const uint8_t *meta = sk_rec_byte(rec, g, P.L.Dp + 26, P.L.rec_bytes, planar);  // agent, phase, done, status (one 4-byte word)
  // done, and the step that wrote the record applied (or refused) an action: byte D is -1 for a game that was re-dealt,
  // already over or left alone (SKYJO_ACTION_SKIP) - none of those ends an episode (again)
  const bool end = meta[2] != 0 && (int8_t)*sk_rec_byte(rec, g, P.L.D, P.L.rec_bytes, planar) != -1;
  end_out[g] = end ? 1 : 0;
  for (int p = 0; p < P.L.N; p++) rew_out[(size_t)g * P.L.N + p] = end ? P.rewards[(size_t)g * P.L.N + p] : 0.0;
}

// records -> the reference's dense arrays (obs int8[n][D], mask int8[n][26], ...) from either layout
__global__ void k_unpack(SkLayout L, const uint8_t *rec, long long n, int8_t *obs, int8_t *mask, uint8_t *agent,
                         uint8_t *phase, uint8_t *done, uint8_t *status, int planar) {
  const long long total = n * (long long)(L.D + 26);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / (L.D + 26);
    const int k = (int)(i % (L.D + 26));
#define SRC(b) (*sk_rec_byte(rec, r, (b), L.rec_bytes, planar))
    if (k < L.D) {
      if (obs) obs[r * L.D + k] = (int8_t)SRC(k);
    } else {
      if (mask) mask[r * 26 + (k - L.D)] = (int8_t)SRC(L.Dp + (k - L.D));
    }
    if (k == 0) {
      if (agent) agent[r] = SRC(L.Dp + 26);
      if (phase) phase[r] = SRC(L.Dp + 27);
      if (done) done[r] = SRC(L.Dp + 28);
      if (status) status[r] = SRC(L.Dp + 29);
    }
#undef SRC
  }
}

// ------------------------------------------------------------------------------------------
// The reference's two scoring helpers for CALLER-SUPPLIED hands (its notebook calls them directly), one lane per hand set:
//   k_evaluate_game   SkyjoGame._evaluate_game(players_cards, player_won_id, score_penalty)   skyjo.py:477-498
//   k_final_rewards   SimpleSkyjoEnv._calc_final_rewards(final_score, num_refunded)           skyjo_env.py:293-312
// float64 with numpy's operation order (np.mean = pairwise sum: left to right below eight addends, eight partial sums
// from eight on), no contraction (-ffp-contract=off) - the arithmetic of finish_game, outside a game.
// ------------------------------------------------------------------------------------------
__global__ void k_evaluate_game(int n, int N, const int8_t *cards, const int32_t *won, double penalty, double *scores) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int8_t *c = cards + (size_t)i * N * 12;
  double sc[SKYJO_MAX_PLAYERS], mn = 0.0;
  for (int p = 0; p < N; p++) {
    int s = 0;
    for (int k = 0; k < 4; k++) {
      const int t0 = c[12 * p + 3 * k], t1 = c[12 * p + 3 * k + 1], t2 = c[12 * p + 3 * k + 2];
      if (!(t0 == t1 && t1 == t2)) s += t0 + t1 + t2;  // skyjo.py:488-493: min != max of the stack of three
    }
    sc[p] = (double)s;
    mn = (p == 0 || sc[p] < mn) ? sc[p] : mn;
  }
  const int w = won[i];
  for (int p = 0; p < N; p++) scores[(size_t)i * N + p] = (p == w && mn != sc[p]) ? sc[p] * penalty : sc[p];  // skyjo.py:496-497
}

__global__ void k_final_rewards(int n, int N, const double *score, const int32_t *refunded, double mean_reward, double reward_refunded,
                                double *rew) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double *a = score + (size_t)i * N;
  double sum;
  if (N < 8) {
    sum = 0.0;
    for (int p = 0; p < N; p++) sum += a[p];
  } else {
    sum = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    for (int p = 8; p < N; p++) sum += a[p];
  }
  const double mean = sum / (double)N;
  for (int p = 0; p < N; p++) {
    double r = (-a[p] + mean) + mean_reward;
    if (reward_refunded != 0.0) r += (double)refunded[(size_t)i * N + p] * reward_refunded;  // skyjo_env.py:309-310
    rew[(size_t)i * N + p] = r;
  }
}
