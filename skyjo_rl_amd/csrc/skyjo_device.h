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
  uint32_t be_add, be_read; // bank_empty[be_add]: where this launch's plans count the banks they find empty; [be_read]: the word its dealing run hands to the host
  uint32_t wg_local;        // 1 inside k_cycle: the dealing wavefront of a game and its step wavefront share a workgroup (SK_FENCE_*)
  uint32_t host_seq;        // != 0 (single-tile engines, host-style step): the wavefront's last store is this number into health_host[3]
  int32_t *deal_list;       // [2][tiles*64] games of the current / previous dealing launch (k_scan)
  uint32_t *deal_ep;        // [2][tiles*64] episode index of each listed deal
  uint32_t *deal_count;     // [2]
  uint32_t *bank_empty;     // [2] games whose bank held no episode when a scan / plan looked (early warning of a drain); which word: be_add / be_read, k_deal
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

// The rest of the device code, in dependency order (round 6: one header per concern instead of 2 700 lines in this one)
#define SKYJO_DEVICE_PARTS 1
#include "skyjo_rng.h"
#include "skyjo_transition.h"
#include "skyjo_record.h"
#include "skyjo_step.h"
#include "skyjo_deal.h"
#include "skyjo_cycle.h"
#include "skyjo_callers.h"
