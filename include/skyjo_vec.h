/*
 * skyjo_vec.h - C ABI of the MI355X-native vectorised SkyJo environment (libskyjo_vec.so).
 *
 * This is the drop-in boundary for ONE hot path of michaelfeil/skyjo_rl: the SkyJo state transition
 * plus observation / action-mask build.  The reference has no FFI: the path sits behind two Python
 * classes, so every entry point below cites the Python interface it replaces
 * (paths relative to the reference checkout):
 *
 *   SkyjoGame.__init__ / reset / set_seed      rlskyjo/game/skyjo.py:20-49, 52-74, 84-94
 *   SkyjoGame.collect_observation              rlskyjo/game/skyjo.py:148-199 (+ helpers :201-302)
 *   SkyjoGame.act                              rlskyjo/game/skyjo.py:308-335 (+ :337-498)
 *   SkyjoGame.get_game_metrics/expected_action rlskyjo/game/skyjo.py:500-504
 *   SimpleSkyjoEnv.step / reset / seed         rlskyjo/environment/skyjo_env.py:216-252, 254-267, 280-290
 *   SimpleSkyjoEnv._calc_final_rewards         rlskyjo/environment/skyjo_env.py:293-312
 *   TerminateIllegalWrapper(illegal_reward=-1) rlskyjo/environment/skyjo_env.py:23
 *
 * All games of a handle share one configuration and live on ONE GPU in an LDS-friendly tiled
 * structure-of-arrays layout.  Unless a parameter is named *_host, data pointers are
 * DEVICE pointers (e.g. torch tensor .data_ptr()); buffers are caller-owned, the handle owns its
 * state.  `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls on one
 * handle must be serialised by the caller; every call is asynchronous on `stream` unless it has a
 * host output.  Return value: 0 = ok, negative = SKYJO_E_*; skyjo_vec_last_error() describes the
 * last failure on the calling thread.  Game-level illegal actions are data (status byte), never
 * errors.  There is NO CPU fallback: without a gfx950 device skyjo_vec_create fails.
 */
#ifndef SKYJO_VEC_H
#define SKYJO_VEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKYJO_ABI_VERSION 4
#define SKYJO_MAX_PLAYERS 12 /* skyjo.py:24-26 */
#define SKYJO_NUM_ACTIONS 26 /* skyjo.py:46 */
#define SKYJO_NUM_CARDS 150  /* skyjo.py:80 */
#define SKYJO_HAND_NONE 15   /* skyjo.py:33 */
#define SKYJO_REFUNDED (-14) /* skyjo.py:34 */

/* error codes */
#define SKYJO_OK 0
#define SKYJO_E_INVALID (-1) /* bad argument / handle */
#define SKYJO_E_DEVICE (-2)  /* HIP runtime error (text in last_error) */
#define SKYJO_E_NOGPU (-3)   /* no usable gfx950 device */
#define SKYJO_E_STATE (-4)   /* call not valid in the current state (e.g. step before seed) */

/* per-game status byte of the last step */
#define SKYJO_ST_OK 0        /* action applied */
#define SKYJO_ST_ILLEGAL 1   /* action masked out or out of range: TerminateIllegalWrapper semantics */
#define SKYJO_ST_NOOP_DONE 2 /* game already over and auto_reset off (skyjo.py:316-321) */
#define SKYJO_ST_RESET 3     /* game was over: a new episode was dealt, the action was ignored */
#define SKYJO_ST_ERROR 4     /* the engine raised a device error (skyjo_vec_check_error): this episode was cut off / could not be dealt */

/* skyjo_vec_step: a game whose action is SKYJO_ACTION_SKIP is left exactly as it is (no step, no reset; its record is
 * still written) - the single-game views use it to step ONE game of a shared engine (SkyjoGame(engine=, index=)). */
#define SKYJO_ACTION_SKIP (-1000)

/* RNG modes */
#define SKYJO_RNG_MT19937 0 /* numpy legacy stream, bit-identical deals to the reference */
#define SKYJO_RNG_PHILOX 1  /* counter-based Philox4x32-10 sessions, same shuffle algorithm */

typedef struct skyjo_vec skyjo_vec; /* opaque */

typedef struct skyjo_vec_config {
  int32_t abi_version;      /* SKYJO_ABI_VERSION */
  int32_t num_envs;         /* parallel games on this device */
  int32_t num_players;      /* 1..12                       (skyjo.py:21) */
  int32_t observe_indirect; /* observe_other_player_indirect (skyjo.py:42-45) */
  double score_penalty;     /* skyjo.py:21,496-497 */
  double mean_reward;       /* skyjo_env.py:43,308 */
  double reward_refunded;   /* skyjo_env.py:44,310-311 */
  double illegal_reward;    /* skyjo_env.py:23 (-1) */
  int32_t device_id;        /* HIP device ordinal */
  int32_t rng_mode;         /* SKYJO_RNG_* */
  int32_t auto_reset;       /* 1: a finished game is re-dealt by the next step */
  int32_t reserved;
  uint64_t game_id0;        /* global id of local game 0 (multi-GPU shards; seeds and policy keys use it) */
} skyjo_vec_config;

/* Output record: one per game per step, `record_bytes` long (64 for the indirect observation):
 *   [0, D)            int8 observations            (skyjo.py:180-190)
 *   [D]               int8 the action the step that wrote this record applied to the game (-1: none - the game was
 *                     reset / already over / skipped, or the record comes from reset / observe; -2: the caller's action
 *                     was outside 0 .. 25 and refused as illegal); D is odd, so the byte is the padding between the
 *                     observation and the mask
 *   [Dp, Dp+26)       int8 action_mask, Dp = (D+3)&~3 (skyjo.py:201-224)
 *   [Dp+26]           expected player (agent id)   (skyjo.py:503)
 *   [Dp+27]           phase 0 = draw, 1 = place
 *   [Dp+28]           done                         (skyjo_env.py:247)
 *   [Dp+29]           status SKYJO_ST_*
 *   [Dp+30, Dp+32)    uint16 steps in this episode */
typedef struct skyjo_vec_info {
  int32_t num_envs, num_players, obs_dim, record_bytes;
  int32_t mask_offset, meta_offset, state_bytes, tile_games;
} skyjo_vec_info;

typedef struct skyjo_vec_counters {
  uint64_t steps;    /* applied actions incl. the consumed terminal draw */
  uint64_t episodes; /* games that reached their natural end */
  uint64_t illegal;  /* games ended by an illegal action */
  uint64_t resets;   /* deals consumed */
  uint64_t sum_len;  /* sum of episode lengths of finished episodes */
  uint64_t reshuffles;
  uint64_t iters;    /* lockstep iterations executed */
  uint64_t waits;    /* deals made on the in-kernel slow path (pre-dealt episode not available) */
  double sum_score[SKYJO_MAX_PLAYERS];     /* per seat, finished episodes (skyjo.py:59) */
  double sum_reward[SKYJO_MAX_PLAYERS];    /* per seat, finished + illegal episodes (skyjo_env.py:293-312) */
  double sum_reward_sq[SKYJO_MAX_PLAYERS]; /* per seat, squares of the same rewards */
  double sum_refunded[SKYJO_MAX_PLAYERS];  /* per seat, num_refunded of finished episodes (skyjo.py:57) */
} skyjo_vec_counters;

/* Canonical per-game state (debug, fixtures, snapshot/restore).  Field names follow skyjo.py. */
typedef struct skyjo_game_state {
  int8_t players_cards[SKYJO_MAX_PLAYERS][12];  /* skyjo.py:63 */
  int8_t players_masked[SKYJO_MAX_PLAYERS][12]; /* skyjo.py:72, 0 refunded / 1 open / 2 hidden */
  int8_t drawpile[SKYJO_NUM_CARDS];             /* pop from the end, skyjo.py:366 */
  int8_t discard_pile[SKYJO_NUM_CARDS];         /* push/pop at the end, skyjo.py:370,393 */
  int16_t n_draw, n_disc;
  int8_t hand_card;        /* 15 = none */
  uint8_t expected_player; /* skyjo.py:144 */
  uint8_t expected_phase;  /* 0 draw / 1 place */
  uint8_t is_terminated;   /* skyjo.py:54 */
  uint8_t done;            /* env-level: terminated or ended by illegal action */
  uint8_t status;
  uint16_t episode_steps;
  uint32_t episode;        /* deals since seeding, 0 = the deal made by seed() */
  uint32_t reshuffles;
  int32_t num_refunded[SKYJO_MAX_PLAYERS]; /* skyjo.py:57 */
  int32_t num_placed[SKYJO_MAX_PLAYERS];   /* skyjo.py:58 */
  double final_score[SKYJO_MAX_PLAYERS];   /* skyjo.py:59, valid when is_terminated */
  double rewards[SKYJO_MAX_PLAYERS];       /* skyjo_env.py:244, valid when done */
} skyjo_game_state;

const char *skyjo_vec_last_error(void);
int skyjo_vec_abi_version(void);

/* SkyjoGame.__init__ + SimpleSkyjoEnv.__init__ (skyjo.py:20-49, skyjo_env.py:38-153) for num_envs games.
 * Unlike the reference no cards are dealt until skyjo_vec_seed. */
int skyjo_vec_create(const skyjo_vec_config *cfg, skyjo_vec **out);
int skyjo_vec_destroy(skyjo_vec *h);
int skyjo_vec_get_info(const skyjo_vec *h, skyjo_vec_info *out);

/* SkyjoGame.set_seed (skyjo.py:84-88) per game: game i is seeded with seeds_host[i], or with
 * base_seed + game_id0 + i when seeds_host is NULL; the legacy stream is seeded with value+1 and
 * the first deal is made immediately, exactly like set_seed.  Must precede every other call. */
int skyjo_vec_seed(skyjo_vec *h, const uint64_t *seeds_host, uint64_t base_seed, void *stream);
/* SkyjoGame.set_seed (skyjo.py:84-88) for ONE game of the batch: its stream is re-seeded with value + 1, its bank of
 * pre-dealt episodes is dealt again and the first deal becomes the live game.  The other games are untouched. */
int skyjo_vec_seed_one(skyjo_vec *h, int32_t game, uint64_t value, void *stream);

/* SkyjoGame.reset / SimpleSkyjoEnv.reset (skyjo.py:52-74, skyjo_env.py:254-267) for the games whose
 * mask byte is non-zero (all when mask is NULL).  records_out (may be NULL): [num_envs][record_bytes]. */
int skyjo_vec_reset(skyjo_vec *h, const uint8_t *mask, void *records_out, void *stream);

/* SimpleSkyjoEnv.step (skyjo_env.py:216-252) = SkyjoGame.act (skyjo.py:308-335) for the expected player
 * of every game, then the observation of the next expected player.  actions: int32[num_envs]. */
int skyjo_vec_step(skyjo_vec *h, const int32_t *actions, void *records_out, void *stream);

/* `iters` lockstep iterations with the uniform random admissible policy (rlskyjo/models/random_admissible_policy.py:6-28)
 * evaluated on device.  The call is cut into launches at the dealing cadence (SKYJO_OPT_DEAL_INTERVAL); in the one-kernel
 * form (SKYJO_OPT_OVERLAP 3) up to 16 whole dealing cycles share ONE launch when the call asks for that many iterations at
 * once - the games' tiles then stay in LDS across the cycle ends (65 536 x 3: 37 -> 44 x 10^9 env-steps/s from 64 to 512+
 * iterations per call).  Results do not depend on how a caller slices its iterations into calls.
 * records_out: NULL, or [iters][num_envs][record_bytes] (record AFTER each iteration);
 * actions_out: NULL, or int32[iters][num_envs] (-1 where no action was applied) - the same value is byte D of every
 * record, so a caller that keeps the records does not need this array. */
int skyjo_vec_rollout(skyjo_vec *h, int32_t iters, uint64_t policy_seed, void *records_out, int32_t *actions_out,
                      void *stream);

/* SimpleSkyjoEnv.observe (skyjo_env.py:199-214) = collect_observation(player) (skyjo.py:148-199) without
 * changing state.  players: int32[num_envs] or NULL for the expected player of each game. */
int skyjo_vec_observe(skyjo_vec *h, const int32_t *players, void *records_out, void *stream);

/* Split records into the reference's dense arrays: obs int8[n][D], mask int8[n][26]; any output may be NULL. */
int skyjo_vec_unpack(skyjo_vec *h, const void *records, int64_t n_records, int8_t *obs, int8_t *mask,
                     uint8_t *agent, uint8_t *phase, uint8_t *done, uint8_t *status, void *stream);
/* The same for records in the tile-planar layout (SKYJO_OPT_RECORD_LAYOUT = SKYJO_REC_TILE_PLANAR): `n_tiles` blocks of 64 records
 * (64 * record_bytes bytes each) in, dense rows of 64 n_tiles records out - [iters * tiles] blocks of a rollout give rows ordered
 * (iteration, tile, lane), i.e. (iteration, game) with the games of a partial last tile padded to 64. */
int skyjo_vec_unpack_tiles(skyjo_vec *h, const void *records, int64_t n_tiles, int8_t *obs, int8_t *mask,
                           uint8_t *agent, uint8_t *phase, uint8_t *done, uint8_t *status, void *stream);

/* rewards double[num_envs][num_players] (skyjo_env.py:293-312; valid where done), device pointers owned by h */
const double *skyjo_vec_rewards_ptr(const skyjo_vec *h);
const double *skyjo_vec_scores_ptr(const skyjo_vec *h);
const uint8_t *skyjo_vec_done_ptr(const skyjo_vec *h);

/* Sticky device error of the handle: 0, or SKYJO_E_DEVICE with the reason in skyjo_vec_last_error() - today the one case
 * is a step kernel that gave up waiting for the dealing kernel that should run beside it (SKYJO_OPT_OVERLAP); the games
 * concerned show SKYJO_ST_ERROR once (the stream the dealing kernel may still hold is left alone), results since then
 * are void, skyjo_vec_seed clears it (and skyjo_vec_snapshot_restore: a snapshot is only ever taken of a clean run).  A game
 * whose in-place deal or reset timed out shows done / SKYJO_ST_ERROR (not RESET; no reset is counted), its rewards are zero.
 * Synchronises `stream`.
 * Every synchronising call below (and the *_host conveniences, snapshot_create) makes the same check by itself. */
int skyjo_vec_check_error(skyjo_vec *h, void *stream);

/* synchronising host-side accessors */
int skyjo_vec_get_counters(skyjo_vec *h, skyjo_vec_counters *out_host, void *stream);
int skyjo_vec_reset_counters(skyjo_vec *h, void *stream);
int skyjo_vec_get_state(skyjo_vec *h, int32_t game, skyjo_game_state *out_host, void *stream);
int skyjo_vec_set_state(skyjo_vec *h, int32_t game, const skyjo_game_state *in_host, void *stream);
/* np.random.seed(value) on one game's legacy stream without dealing (fixture injection) */
int skyjo_vec_seed_raw(skyjo_vec *h, int32_t game, uint32_t value, void *stream);

/* The reference draws its deals and mid-game reshuffles from numpy's process-global legacy stream (skyjo.py:81,94,101,135),
 * the same one policy_ra(obs, mask) without a generator draws from (random_admissible_policy.py:22-23).  For a handle in
 * MT19937 mode with SKYJO_OPT_NO_BANK these two calls move that stream in and out of one game in numpy's own terms -
 * np.random.get_state(): key uint32[624], pos 0..624 - so a single-game view can hand the caller's stream to the device
 * before a deal or a reshuffle and hand it back afterwards (skyjo_rl_amd/game.py: SkyjoGame(global_rng=True)).
 * get_state returns the block completed the way numpy keeps it (the engine regenerates the state lazily).  Both synchronise.
 * (This is the reference's behaviour with numba's JIT DISABLED - the mode its seeded test pins.  Jitted, the reference's
 * np.random calls use numba's private generator and never touch the caller's global stream; a view in this mode re-seeds
 * and advances the caller's stream instead: skyjo_rl_amd/game.py, module docstring.) */
int skyjo_vec_rng_set_state(skyjo_vec *h, int32_t game, const uint32_t *key_host, int32_t pos, void *stream);
int skyjo_vec_rng_get_state(skyjo_vec *h, int32_t game, uint32_t *key_out_host, int32_t *pos_out_host, void *stream);

/* Kernel timing with HIP events on the launch stream: while enabled, every kernel of the path is launched with a
 * (start, stop) event pair that receives the kernel's own begin and end timestamps.  Returns and clears what was
 * collected since the last call - sum of milliseconds and launch count per kernel: [0] k_step, [1] k_scan, [2] k_deal,
 * [3] k_publish, [4] the policy / value net launches of skyjo_vec_model_rollout - then switches collection on (1) or off (0).
 * Synchronises the device. */
#define SKYJO_PROF_KERNELS 5
int skyjo_vec_profile(skyjo_vec *h, int enable, double ms_out[SKYJO_PROF_KERNELS], int64_t launches_out[SKYJO_PROF_KERNELS]);

/* Snapshot / restore of the WHOLE engine (the reference has no such feature): every live game, every bank
 * of pre-dealt episodes, every RNG stream with its position, statistics and the policy counter.  A snapshot lives in
 * device memory owned by the library; restoring it into the handle it was taken from makes every later call produce
 * exactly what it produced after the snapshot was taken.  Both calls drain the dealing pipeline and synchronise. */
typedef struct skyjo_vec_snapshot skyjo_vec_snapshot; /* opaque */
int skyjo_vec_snapshot_create(skyjo_vec *h, skyjo_vec_snapshot **out, void *stream);
int skyjo_vec_snapshot_restore(skyjo_vec *h, const skyjo_vec_snapshot *snap, void *stream);
int skyjo_vec_snapshot_bytes(const skyjo_vec_snapshot *snap, size_t *bytes_out);
int skyjo_vec_snapshot_destroy(skyjo_vec_snapshot *snap);

/* Diagnostic builds only (-DSK_STAMPS): per-section shader-cycle sums, 8 for the step kernel followed by 8 for
 * the dealing kernel, summed over wavefronts, cleared on read.  The shipped build returns zeros. */
int skyjo_vec_debug_stamps(skyjo_vec *h, uint64_t *out16_host);
/* Diagnostic builds (-DSK_TRACE) only: placement and time span of every wavefront of the last two step / dealing launches,
 * uint64 [4][tiles][8] (tools/dev/placement.py).  Zeros in the shipped build. */
int skyjo_vec_debug_trace(skyjo_vec *h, uint64_t *out_host);

/* Tunables.  SKYJO_OPT_DEAL_INTERVAL: lockstep iterations (steps or rollout iterations) between two runs of the
 * dealing kernel (1..1024).  Unless it is set, the engine starts at 88 (three
 * and more players) or 64 - 80 or 56 with the dealing kernel beside the step kernel - and adapts: every dealing run reports how many banks it found empty, any empty bank shortens
 * the interval, a long calm stretch lengthens it again - so it settles below the episode length of the policy in use.  Every game owns a bank of three
 * pre-dealt episodes and a dealing run adds at most one per game; a finished game whose bank is empty deals in
 * place (slow path, same result, counted in skyjo_vec_counters.waits). */
#define SKYJO_OPT_DEAL_INTERVAL 1
/* SKYJO_OPT_OVERLAP - where the dealing runs (results do not depend on it):
 *   0  in line on the caller's stream, after the step launch that makes it due;
 *   2  the two-stream form: the dealing kernel on a stream of its own beside the step kernels that follow (its episodes are
 *      published one dealing cycle later; the step kernel plans and publishes the runs itself, nothing on the caller's
 *      stream waits for the dealing stream).  Default, for batches that leave SIMDs idle (at most 768 tiles of 64 games), of
 *      the engines form 3 does not cover;
 *   3  the one-kernel form (k_cycle): every workgroup is one CU's worth of wavefronts, one step and one dealing wavefront per
 *      SIMD, the dealing wavefronts work for the games of their own workgroup - the hand-over never leaves the CU, so it needs
 *      no L2 write-back / invalidation.  Two to four players, either observation; S = 1 .. 4 step and as many dealing wavefronts
 *      per workgroup, whichever spreads the batch over the CUs and fits their LDS regions into 160 KB.  A launch may span up to
 *      16 dealing cycles.  The default wherever it exists, except where the LDS forces S below the batch's share of tiles per CU
 *      (a full chip of four-player games: form 0 there); the fused rollout only - other calls deal
 *      as in form 0;
 *   1  "beside the step kernel" in whichever of the two forms the engine prefers.
 * skyjo_vec_get_option returns 0, 2 or 3. */
#define SKYJO_OPT_OVERLAP 2
/* Fault injection for the tests (never needed in production): SKYJO_OPT_DEBUG_SPIN_LOG2 - a step kernel that has to wait
 * for an overlapped dealing kernel gives up after 2^value polls (default 22) and raises the sticky device error that
 * skyjo_vec_get_counters reports; SKYJO_OPT_DEBUG_DEAL_DELAY - every dealing wavefront sleeps value x 8128 cycles first. */
#define SKYJO_OPT_DEBUG_SPIN_LOG2 3
#define SKYJO_OPT_DEBUG_DEAL_DELAY 4
/* SKYJO_OPT_NO_BANK (set before skyjo_vec_seed): 1 = no episodes are dealt ahead - every reset deals in place from the
 * stream's current position, exactly when the reference would (skyjo.py:52-74).  Slow for batches; it is what lets ONE
 * game share the process-global numpy stream with its caller (skyjo_vec_rng_set_state / _get_state below). */
#define SKYJO_OPT_NO_BANK 5
/* SKYJO_OPT_RECORD_LAYOUT - how skyjo_vec_rollout lays out records_out (every other call writes row-major records unless ..._ALL below is chosen):
 *   SKYJO_REC_ROW_MAJOR    [iters][num_envs][record_bytes] - the default;
 *   SKYJO_REC_TILE_PLANAR  [iters][tiles][P][64][16], tiles = ceil(num_envs / 64), P = record_bytes / 16 (4 for the indirect
 *                          observation; 5 / 6 / 7 for the direct one with 2 / 3 / 4 players): the record of game 64 t + l is cut into
 *                          16-byte pieces, piece p at byte ((it * tiles + t) * P + p) * 1024 + l * 16.  A tile's records of one
 *                          iteration are still one contiguous block (P KiB), but every store instruction of the step wavefront now
 *                          writes 1 KiB of it straight from the registers the record was assembled in - no staging through LDS.
 *                          The buffer holds tiles * 64 records per iteration (the slots of a partial last tile beyond num_envs are
 *                          not written).  The one-kernel form (SKYJO_OPT_OVERLAP 3) only; skyjo_vec_unpack_tiles turns such blocks
 *                          into the reference's dense arrays. */
#define SKYJO_OPT_RECORD_LAYOUT 6
#define SKYJO_REC_ROW_MAJOR 0
#define SKYJO_REC_TILE_PLANAR 1
/* SKYJO_REC_TILE_PLANAR_ALL (round 6; indirect observation, two to four players): as SKYJO_REC_TILE_PLANAR, and every other device-style call
 * that writes records - skyjo_vec_reset, _observe, _step, _step_collect, _model_rollout (its records buffer is then
 * [T + 1][tiles][P][64][16]) - writes them tile-planar too, so that a rollout with the policy net never holds a row-major record
 * (the net, the masked draw and the episode-end columns read the blocks in place: skyjo_vec_*_layout).  The *_host calls stay row-major. */
#define SKYJO_REC_TILE_PLANAR_ALL 2
/* The older forms of a dealing run, kept for the engines the one-kernel form does not cover and selectable so that the parity tests
 * can hold them against the oracle (results never depend on them):
 *   SKYJO_OPT_INLINE_WORK_LIST 1: an in-line run uses the k_scan + work-list form (default 0: the dealing kernel scans the banks itself);
 *   SKYJO_OPT_UNPIPELINED      1: a run beside the step kernel uses the k_scan / k_publish form, whose caller's stream waits for every
 *                                 run (default 0: the step kernel plans and publishes the runs itself).
 * SKYJO_OPT_CYCLE_S (before skyjo_vec_seed): step - and dealing - wavefronts per workgroup of the one-kernel form, 1 .. 4; 0 = the
 * batch's share of tiles per compute unit (the default).  SKYJO_OPT_MAX_CYCLES_PER_LAUNCH: dealing cycles ONE launch of
 * skyjo_vec_rollout may span, 1 .. 16 (default 16).  The shipped library reads no environment variable; the measurement switches of
 * tools/dev/README.md exist in -DSK_DIAG builds only. */
#define SKYJO_OPT_INLINE_WORK_LIST 7
#define SKYJO_OPT_UNPIPELINED 8
#define SKYJO_OPT_CYCLE_S 9
#define SKYJO_OPT_MAX_CYCLES_PER_LAUNCH 10
int skyjo_vec_set_option(skyjo_vec *h, int option, int64_t value);
int skyjo_vec_get_option(const skyjo_vec *h, int option, int64_t *value_out);

/* Caller piece of config 5: the masking and sampling step of the action-mask policy model,
 * rlskyjo/models/action_mask_model.py:58-74 -  masked = logits + clamp(log(action_mask), min=FLOAT_MIN)  - followed
 * by RLlib's categorical draw from softmax(masked), fused into one pass over the records the engine has just
 * written (the mask bytes are read in place, offset mask_offset of each record).
 *   logits     float32 [n][26]  the policy net's outputs (device)
 *   records    [n][record_bytes] as written by step / reset / observe / rollout (device)
 *   seed, ticket   the uniform of game i is word 0 of Philox4x32-10(counter = (ticket, game_id0 + i, 0x53414D50),
 *              key = seed): the same (seed, ticket) reproduces the same draws; advance ticket once per step
 *   no_masking != 0: the mask is ignored (action_mask_model.py:53-56,66-67)
 *   actions_out int32 [n]; logp_out float32 [n] (log-probability of the drawn action) or NULL;
 *   uniform_out float32 [n] (the uniforms in [0, 1), for tests) or NULL.
 * Arithmetic is float32: probabilities agree with torch.softmax to 1e-6 (tests/test_gpu_sampler.py).  The order of the sums is part
 * of the definition (csrc/skyjo_draw.h: exponentials in blocks of four actions, block totals left to right, the CDF of a block
 * starting from the total before it; the action is the smallest k of non-zero probability whose CDF value exceeds uniform * sum), so that the draw made in
 * the net's own launch (skyjo_vec_mlp_act_value: two lanes per game) and this one (one lane per game) give the same bits. */
int skyjo_vec_sample_actions(skyjo_vec *h, const void *records, const float *logits, int64_t n, uint64_t seed,
                             uint64_t ticket, int32_t no_masking, int32_t *actions_out, float *logp_out,
                             float *uniform_out, void *stream);

/* The policy net of config 5 on the matrix cores: RLlib's default fully connected net as the reference's
 * TorchActionMaskModel builds it (rlskyjo/models/action_mask_model.py:41-52: obs -> 256 tanh -> 256 tanh -> outputs),
 * evaluated for n records in one kernel that reads the int8 observation out of each record (indirect observation,
 * obs_dim <= 31).  Weights are given once in torch.nn.Linear layout (row-major [out][in], float32, HOST pointers) and
 * kept on the device as bf16 MFMA fragments (one per weight, or a high and a low one: see SKYJO_MLP_*); accumulation is
 * float32.  out: float32 [n][out_dim] (device), out_dim <= 32 (26 logits for the policy branch, 1 for the value branch).
 * tests/test_gpu_policy_net.py states the tolerance of either precision against the float32 torch module. */
typedef struct skyjo_vec_mlp skyjo_vec_mlp; /* opaque */
/* precision of a packed net.  SKYJO_MLP_FP32 is the drop-in for the reference's float32 TorchFC
 * (action_mask_model.py:43-49): every operand is the sum of two bf16 values and every product three MFMAs into a float32
 * accumulator, logits / values within 1e-4 of the float32 module (tests/test_gpu_policy_net.py).  SKYJO_MLP_BF16 keeps
 * weights and activations as single bf16 values: a third of the matrix work, logits within 8e-2. */
#define SKYJO_MLP_BF16 0
#define SKYJO_MLP_FP32 1
int skyjo_vec_mlp_create(int32_t device_id, int32_t obs_dim, int32_t out_dim, int32_t precision, const float *w1, const float *b1,
                         const float *w2, const float *b2, const float *w3, const float *b3, skyjo_vec_mlp **out);
int skyjo_vec_mlp_destroy(skyjo_vec_mlp *m);
int skyjo_vec_mlp_forward(const skyjo_vec_mlp *m, const void *records, int32_t record_bytes, int64_t n, float *out,
                          void *stream);
/* Policy branch (out_dim == 26) and the draw of skyjo_vec_sample_actions in ONE launch: the logits never leave the
 * registers (logits_out, float32 [n][26], may be NULL).  Same (seed, ticket) -> the same actions as
 * skyjo_vec_mlp_forward followed by skyjo_vec_sample_actions, bit for bit. */
int skyjo_vec_mlp_act(skyjo_vec *h, const skyjo_vec_mlp *m, const void *records, int64_t n, uint64_t seed, uint64_t ticket,
                      int32_t no_masking, int32_t *actions_out, float *logp_out, float *logits_out, void *stream);

/* Policy AND value branch of the action-mask model (two nets of skyjo_vec_mlp_create over the same records) in ONE
 * launch: the draw of skyjo_vec_mlp_act for the policy net, values_out float32 [n][value out_dim] for the other. */
int skyjo_vec_mlp_act_value(skyjo_vec *h, const skyjo_vec_mlp *policy, const skyjo_vec_mlp *value, const void *records, int64_t n,
                            uint64_t seed, uint64_t ticket, int32_t no_masking, int32_t *actions_out, float *logp_out,
                            float *logits_out, float *values_out, void *stream);
/* Rollout collection: for the records a step has just written (device, [num_envs][record_bytes]) mark the games whose
 * episode ended in that step - episode_end_out uint8 [num_envs] - and copy their final rewards (skyjo_env.py:293-312)
 * into final_rewards_out double [num_envs][num_players], zeros elsewhere.  One small kernel, no host traffic. */
int skyjo_vec_episode_ends(skyjo_vec *h, const void *records, double *final_rewards_out, uint8_t *episode_end_out, void *stream);

/* The same consumers for records in an explicit layout (SKYJO_REC_ROW_MAJOR, or SKYJO_REC_TILE_PLANAR: what skyjo_vec_rollout writes with
 * SKYJO_OPT_RECORD_LAYOUT = tile-planar - piece p of game 64 t + l at block t * 64 * record_bytes + p * 1024 + l * 16 - is read in place,
 * no skyjo_vec_unpack_tiles pass; n counts games, the last tile's block is complete in memory whatever n).  Same results, bit for
 * bit, as the row-major calls on the same records (tests/test_gpu_record_layout.py).  value / values_out of ..._act_value_layout
 * may be NULL (then it is skyjo_vec_mlp_act). */
int skyjo_vec_mlp_forward_layout(const skyjo_vec_mlp *m, const void *records, int32_t record_bytes, int32_t layout, int64_t n, float *out,
                                 void *stream);
int skyjo_vec_mlp_act_value_layout(skyjo_vec *h, const skyjo_vec_mlp *policy, const skyjo_vec_mlp *value, const void *records, int32_t layout,
                                   int64_t n, uint64_t seed, uint64_t ticket, int32_t no_masking, int32_t *actions_out, float *logp_out,
                                   float *logits_out, float *values_out, void *stream);
int skyjo_vec_sample_actions_layout(skyjo_vec *h, const void *records, int32_t layout, const float *logits, int64_t n, uint64_t seed,
                                    uint64_t ticket, int32_t no_masking, int32_t *actions_out, float *logp_out,
                                    float *uniform_out, void *stream);
int skyjo_vec_episode_ends_layout(skyjo_vec *h, const void *records, int32_t layout, double *final_rewards_out, uint8_t *episode_end_out,
                                  void *stream);

/* skyjo_vec_step that also does what skyjo_vec_episode_ends does, inside the step kernel (no extra launch): the lane that
 * ends an episode writes episode_end_out[game] = 1 and the game's final rewards, every other game 0 / zeros. */
int skyjo_vec_step_collect(skyjo_vec *h, const int32_t *actions, void *records_out, double *final_rewards_out,
                           uint8_t *episode_end_out, void *stream);

/* Config 5's collection loop in ONE call (what rlskyjo/models/train_model_simple_rllib.py:22-59 has RLlib's
 * rollout workers do, per step: forward of the action-mask model, masked draw, env step, bookkeeping of episode ends): T
 * lockstep iterations of  [policy (+ value) net with the draw in its epilogue] -> [step kernel with the episode-end
 * columns] - two launches per iteration, nothing on the host in between.  All buffers are device memory of the caller:
 *   records        [T+1][num_envs][record_bytes]  IN: records[0] = what the first actor sees (skyjo_vec_observe / the last
 *                                                  records of the previous call); OUT: records[t+1] = after step t
 *   actions        int32 [T][num_envs]            the drawn actions
 *   logp           float32 [T][num_envs] or NULL  their log-probabilities
 *   values         float32 [T+1][num_envs][value out_dim]  (required iff `value` is given; [T] = bootstrap value)
 *   final_rewards  double [T][num_envs][num_players] and episode_end uint8 [T][num_envs]: both or neither
 * The draw of iteration t uses (seed, first_ticket + t) as in skyjo_vec_mlp_act: the same call sequence made one launch at a
 * time from the host gives the same bits. */
typedef struct skyjo_vec_rollout_buffers {
  void *records;
  int32_t *actions;
  float *logp;
  float *values;
  double *final_rewards;
  uint8_t *episode_end;
} skyjo_vec_rollout_buffers;
int skyjo_vec_model_rollout(skyjo_vec *h, const skyjo_vec_mlp *policy, const skyjo_vec_mlp *value, int32_t T, uint64_t seed,
                            uint64_t first_ticket, int32_t no_masking, const skyjo_vec_rollout_buffers *buffers, void *stream);

/* host-pointer conveniences for small batches (single-game AEC view): synchronous.  Up to 4096 games they go through
 * host-mapped memory (one launch + one synchronisation per call, no copies), and step_host / reset_host bring every game's
 * state and rewards back with the records: skyjo_vec_get_state and skyjo_vec_get_rewards_host right after them cost no
 * device traffic. */
int skyjo_vec_step_host(skyjo_vec *h, const int32_t *actions_host, void *records_out_host);
int skyjo_vec_observe_host(skyjo_vec *h, const int32_t *players_host, void *records_out_host);
int skyjo_vec_reset_host(skyjo_vec *h, const uint8_t *mask_host, void *records_out_host);
int skyjo_vec_get_rewards_host(skyjo_vec *h, double *rewards_out_host, double *scores_out_host,
                               uint8_t *done_out_host);

/* The reference's two scoring helpers for CALLER-SUPPLIED hands (its notebook calls them directly; inside a game the step
 * kernel computes the same).  Host pointers, computed on device `device_id`, synchronous; float64 in numpy's operation order.
 *   skyjo_vec_evaluate_game       SkyjoGame._evaluate_game(players_cards, player_won_id, score_penalty)  (skyjo.py:477-498):
 *                             players_cards int8[n][num_players][12], player_won_id int32[n] -> scores double[n][num_players]
 *   skyjo_vec_calc_final_rewards  SimpleSkyjoEnv._calc_final_rewards(final_score, num_refunded)          (skyjo_env.py:293-312):
 *                             final_score double[n][num_players], num_refunded int32[n][num_players] -> rewards double[n][num_players] */
int skyjo_vec_evaluate_game(int32_t device_id, int32_t n, int32_t num_players, const int8_t *players_cards_host,
                        const int32_t *player_won_id_host, double score_penalty, double *scores_out_host);
int skyjo_vec_calc_final_rewards(int32_t device_id, int32_t n, int32_t num_players, const double *final_score_host,
                             const int32_t *num_refunded_host, double mean_reward, double reward_refunded,
                             double *rewards_out_host);

/* plain device-memory helpers so that a caller without torch can own buffers */
int skyjo_dev_malloc(int device_id, size_t bytes, void **out);
int skyjo_dev_free(void *p);
int skyjo_dev_copy(void *dst, const void *src, size_t bytes, int kind /*1 h2d, 2 d2h, 3 d2d*/, void *stream);
int skyjo_dev_sync(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SKYJO_VEC_H */
