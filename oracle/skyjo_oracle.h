/*
 * skyjo_oracle.h - CPU restatement of the rlskyjo hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle for the MI355X build: a plain scalar C restatement of
 *   rlskyjo/game/skyjo.py:52-138   (reset / deal / legacy-RNG shuffles)
 *   rlskyjo/game/skyjo.py:148-302  (collect_observation, action mask)
 *   rlskyjo/game/skyjo.py:308-498  (act, draw, place, column collapse, goal check, scoring)
 *   rlskyjo/environment/skyjo_env.py:216-252,293-312 (final rewards, done bookkeeping)
 * plus numpy's legacy RandomState stream (numpy==1.21.5 pinned by requirements.txt:3; MT19937
 * init_genrand / random_interval / Fisher-Yates) which the reference consumes through
 * np.random.seed/shuffle/choice (skyjo.py:81,94,101,135).
 *
 * Pinning: every function here is checked against the .npz fixtures under tests/golden/, which were produced by
 * running the real reference in the build container (oracle/gen_golden.py).  Nothing under
 * skyjo_rl_amd/ may include, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do.
 */
#ifndef SKYJO_ORACLE_H
#define SKYJO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKO_MAXP 12
#define SKO_NCARDS 150
#define SKO_HAND_NONE 15   /* fill_masked_unk_value, skyjo.py:33 */
#define SKO_REFUNDED (-14) /* fill_masked_refunded_value, skyjo.py:34 */

/* sko_act return codes (>= 0: the reference's bool game_over; < 0: the AssertionError it raises) */
#define SKO_ERR_PLAYER (-1)   /* skyjo.py:310 */
#define SKO_ERR_RANGE (-2)    /* skyjo.py:314 */
#define SKO_ERR_HAS_HAND (-3) /* skyjo.py:324 */
#define SKO_ERR_NO_HAND (-4)  /* skyjo.py:331 */
#define SKO_ERR_REVEALED (-5) /* skyjo.py:399 */

/* vector-step status byte (shared vocabulary with include/skyjo_vec.h) */
#define SKO_ST_OK 0
#define SKO_ST_ILLEGAL 1
#define SKO_ST_NOOP_DONE 2
#define SKO_ST_RESET 3
#define SKO_ACTION_SKIP (-1000) /* sko_vec_step: leave this game as it is */

#define SKO_RNG_MT19937 0 /* numpy legacy stream: bit-identical to the reference */
#define SKO_RNG_PHILOX 1  /* counter-based Philox4x32-10 sessions (build's own definition) */

typedef struct {
  int mode;
  uint32_t mt[624];
  int idx;
  uint64_t px_key;  /* per-game seed */
  uint32_t px_ctr[4];
  uint32_t px_buf[4];
  int px_pos;
} sko_rng;

typedef struct {
  /* config */
  int num_players;
  double score_penalty;
  int indirect;
  /* state (names follow skyjo.py) */
  int8_t players_cards[SKO_MAXP][12];
  int8_t players_masked[SKO_MAXP][12];
  int8_t drawpile[SKO_NCARDS + 8];
  int n_draw;
  int8_t discard_pile[SKO_NCARDS + 8];
  int n_disc;
  int hand_card;
  int exp_player, exp_phase; /* expected_action; phase 0 = draw, 1 = place */
  int is_terminated;
  int num_refunded[SKO_MAXP];
  int num_placed[SKO_MAXP];
  double final_score[SKO_MAXP];
  /* bookkeeping of the vector layer */
  uint32_t episode;    /* deals since seeding (0 = the deal done by set_seed) */
  uint32_t reshuffles; /* mid-game reshuffles in this episode */
  uint64_t reshuffles_total;
  sko_rng rng;
} sko_game;

size_t sko_sizeof(void);
void sko_init(sko_game *g, int num_players, double score_penalty, int indirect, int rng_mode);
int sko_obs_dim(const sko_game *g);

/* RNG */
void sko_rng_seed_legacy(sko_rng *r, uint32_t seed); /* np.random.seed(seed) */
uint32_t sko_rng_next(sko_rng *r);
uint32_t sko_rng_interval(sko_rng *r, uint32_t max);
void sko_shuffle_i8(sko_rng *r, int8_t *a, int n);
void sko_shuffle_i32(sko_rng *r, int32_t *a, int n);
void sko_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* game core */
void sko_set_seed(sko_game *g, uint64_t value); /* skyjo.py:84-88: seed(value+1) then reset() */
void sko_reset(sko_game *g);                    /* skyjo.py:52-74 */
void sko_observe(const sko_game *g, int player, int8_t *obs, int8_t *mask); /* skyjo.py:148-199 */
int sko_act(sko_game *g, int player, int action);                           /* skyjo.py:308-335 */
void sko_evaluate_game(const int8_t cards[][12], int num_players, int finisher, double penalty,
                       double *score); /* skyjo.py:477-498 */
void sko_final_rewards(const sko_game *g, double mean_reward, double reward_refunded,
                       double *out); /* skyjo_env.py:293-312 */
/* fixture injection (the reference has no such API; tests assign attributes directly) */
void sko_set_state(sko_game *g, const int8_t *cards, const int8_t *masked, const int8_t *draw, int n_draw,
                   const int8_t *disc, int n_disc, int hand, int player, int phase);

/* ---- vector layer: the semantics the HIP engine implements, as a plain loop over games ---- */
typedef struct {
  int num_envs, num_players, indirect, rng_mode, auto_reset;
  double score_penalty, mean_reward, reward_refunded, illegal_reward;
  uint64_t game_id0; /* global id of game 0 (multi-GPU shards) */
  sko_game *games;
  uint8_t *done;   /* [B] */
  uint8_t *status; /* [B] */
  double *rewards; /* [B][N], valid while done */
  /* counters */
  uint64_t steps, episodes, illegal, resets;
  uint64_t sum_len;
  uint64_t iter; /* lockstep iterations executed by sko_vec_rollout */
  uint32_t *ep_len;
  double *acc_score;    /* [B][N] sums over the game's finished episodes: final score per seat ... */
  double *acc_refunded; /* ... and num_refunded per seat */
} sko_vec;

sko_vec *sko_vec_create(int num_envs, int num_players, double score_penalty, int indirect, double mean_reward,
                        double reward_refunded, int rng_mode, int auto_reset, uint64_t game_id0);
void sko_vec_destroy(sko_vec *v);
void sko_vec_seed(sko_vec *v, const uint64_t *seeds, uint64_t base); /* per game set_seed(seed) */
void sko_vec_seed_one(sko_vec *v, int i, uint64_t value);
void sko_vec_reset(sko_vec *v, const uint8_t *mask);
void sko_vec_step(sko_vec *v, const int32_t *actions, int threads);
void sko_vec_observe(const sko_vec *v, const int32_t *players, int8_t *obs, int8_t *mask, uint8_t *agent,
                     uint8_t *phase);
/* K lockstep iterations with the on-device uniform-random admissible policy restated */
void sko_vec_rollout(sko_vec *v, int iters, uint64_t policy_seed, int32_t *actions_out, int threads);
void sko_vec_rollout_rec(sko_vec *v, int iters, uint64_t policy_seed, int32_t *actions_out, int8_t *obs_out, int8_t *mask_out,
                         uint8_t *meta_out, uint16_t *eplen_out, int threads); /* + every iteration's record */
int sko_policy_action(uint64_t policy_seed, uint64_t game_id, uint64_t iter, const int8_t *mask);

#ifdef __cplusplus
}
#endif
#endif
