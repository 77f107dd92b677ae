/*
 * skyjo_oracle.c - CPU restatement of the rlskyjo hot path.  TEST INFRASTRUCTURE ONLY
 * (see skyjo_oracle.h for scope, citations and the pinning statement).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp -shared)
 */
#include "skyjo_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ===================================================================================
 * numpy legacy RandomState (numpy/random/mtrand + legacy-distributions)
 * =================================================================================== */

/* np.random.seed(int) -> _legacy_seeding -> mt19937_seed == init_genrand */
void sko_rng_seed_legacy(sko_rng *r, uint32_t seed) {
  r->mode = SKO_RNG_MT19937;
  r->mt[0] = seed;
  for (int i = 1; i < 624; i++) r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
  r->idx = 624;
}

static void mt_twist(sko_rng *r) {
  uint32_t *mt = r->mt;
  for (int i = 0; i < 624; i++) {
    uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
    mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  r->idx = 0;
}

void sko_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int round = 0; round < 10; round++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0, c1 = n1, c2 = n2, c3 = n3;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
  out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

/* Philox "session": ctr = (block, episode, reshuffle index, domain); outputs consumed in order. */
static void px_session(sko_rng *r, uint32_t episode, uint32_t resh, uint32_t domain) {
  r->px_ctr[0] = 0, r->px_ctr[1] = episode, r->px_ctr[2] = resh, r->px_ctr[3] = domain;
  r->px_pos = 4;
}

uint32_t sko_rng_next(sko_rng *r) {
  if (r->mode == SKO_RNG_MT19937) {
    if (r->idx >= 624) mt_twist(r);
    uint32_t y = r->mt[r->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  if (r->px_pos >= 4) {
    uint32_t key[2] = {(uint32_t)r->px_key, (uint32_t)(r->px_key >> 32)};
    sko_philox4x32_10(r->px_ctr, key, r->px_buf);
    r->px_ctr[0]++;
    r->px_pos = 0;
  }
  return r->px_buf[r->px_pos++];
}

/* legacy rk_interval: smallest all-ones mask >= max, rejection (32-bit path) */
uint32_t sko_rng_interval(sko_rng *r, uint32_t max) {
  if (max == 0) return 0;
  uint32_t mask = max;
  mask |= mask >> 1, mask |= mask >> 2, mask |= mask >> 4, mask |= mask >> 8, mask |= mask >> 16;
  uint32_t v;
  do v = sko_rng_next(r) & mask;
  while (v > max);
  return v;
}

/* legacy RandomState.shuffle on a 1-d array: for i = n-1 .. 1: j = interval(i); swap */
void sko_shuffle_i8(sko_rng *r, int8_t *a, int n) {
  for (int i = n - 1; i >= 1; i--) {
    int j = (int)sko_rng_interval(r, (uint32_t)i);
    int8_t t = a[i];
    a[i] = a[j], a[j] = t;
  }
}
void sko_shuffle_i32(sko_rng *r, int32_t *a, int n) {
  for (int i = n - 1; i >= 1; i--) {
    int j = (int)sko_rng_interval(r, (uint32_t)i);
    int32_t t = a[i];
    a[i] = a[j], a[j] = t;
  }
}

/* ===================================================================================
 * game core (rlskyjo/game/skyjo.py)
 * =================================================================================== */
size_t sko_sizeof(void) { return sizeof(sko_game); }

/* skyjo.py:20-49 without the trailing reset() (callers seed first) */
void sko_init(sko_game *g, int num_players, double score_penalty, int indirect, int rng_mode) {
  memset(g, 0, sizeof(*g));
  g->num_players = num_players;
  g->score_penalty = score_penalty;
  g->indirect = indirect ? 1 : 0;
  g->hand_card = SKO_HAND_NONE;
  g->rng.mode = rng_mode;
  g->rng.idx = 624;
  g->rng.px_pos = 4;
}

/* skyjo.py:43-45 */
int sko_obs_dim(const sko_game *g) { return g->indirect ? 31 : 19 + 12 * g->num_players; }

/* skyjo.py:226-257 pieces */
static void revealed_sums(const sko_game *g, int *sums, int *hidden) {
  for (int p = 0; p < g->num_players; p++) {
    int s = 0, h = 0;
    for (int k = 0; k < 12; k++) {
      if (g->players_masked[p][k] == 1) s += g->players_cards[p][k];
      if (g->players_masked[p][k] == 2) h++;
    }
    sums[p] = s, hidden[p] = h;
  }
}

/* skyjo.py:127-138: shuffle in place, drawpile = all but last, discard = [last] */
static void reshuffle_discard_pile(sko_game *g, int8_t *old_pile, int n) {
  sko_shuffle_i8(&g->rng, old_pile, n);
  memcpy(g->drawpile, old_pile, (size_t)(n - 1));
  g->n_draw = n - 1;
  g->discard_pile[0] = old_pile[n - 1];
  g->n_disc = 1;
}

/* skyjo.py:52-74 */
void sko_reset(sko_game *g) {
  const int N = g->num_players;
  if (g->rng.mode == SKO_RNG_PHILOX) px_session(&g->rng, g->episode, 0, 0);
  g->is_terminated = 0;
  memset(g->num_refunded, 0, sizeof(g->num_refunded));
  memset(g->num_placed, 0, sizeof(g->num_placed));
  memset(g->final_score, 0, sizeof(g->final_score));
  g->hand_card = SKO_HAND_NONE;
  /* skyjo.py:76-82: repeat(arange(-2,13),10) then shuffle */
  int8_t deck[SKO_NCARDS];
  for (int i = 0; i < SKO_NCARDS; i++) deck[i] = (int8_t)(-2 + i / 10);
  sko_shuffle_i8(&g->rng, deck, SKO_NCARDS);
  /* skyjo.py:63-65: first 12N cards row-major */
  for (int p = 0; p < N; p++) memcpy(g->players_cards[p], deck + 12 * p, 12);
  /* skyjo.py:68-70 */
  reshuffle_discard_pile(g, deck + 12 * N, SKO_NCARDS - 12 * N);
  /* skyjo.py:96-103: choice(12, 2, replace=False) == permutation(12)[:2] */
  for (int p = 0; p < N; p++) {
    int32_t perm[12];
    for (int k = 0; k < 12; k++) perm[k] = k, g->players_masked[p][k] = 2;
    sko_shuffle_i32(&g->rng, perm, 12);
    g->players_masked[p][perm[0]] = 1;
    g->players_masked[p][perm[1]] = 1;
  }
  /* skyjo.py:105-125: first argmax of revealed sums draws first */
  int sums[SKO_MAXP], hidden[SKO_MAXP], best = 0;
  revealed_sums(g, sums, hidden);
  for (int p = 1; p < N; p++)
    if (sums[p] > sums[best]) best = p;
  g->exp_player = best, g->exp_phase = 0;
  g->reshuffles = 0;
  g->episode++;
}

/* skyjo.py:84-88: np.random.seed(value + 1) then reset() */
void sko_set_seed(sko_game *g, uint64_t value) {
  if (g->rng.mode == SKO_RNG_MT19937) {
    sko_rng_seed_legacy(&g->rng, (uint32_t)(value + 1));
  } else {
    g->rng.px_key = value + 1;
  }
  g->episode = 0;
  g->reshuffles_total = 0;
  sko_reset(g);
}

/* skyjo.py:142-144 with the cycle of :114-120 */
static void next_action(sko_game *g) {
  if (g->exp_phase == 0) {
    g->exp_phase = 1;
  } else {
    g->exp_phase = 0;
    g->exp_player = (g->exp_player + 1) % g->num_players;
  }
}

/* skyjo.py:148-199 */
void sko_observe(const sko_game *g, int player, int8_t *obs, int8_t *mask) {
  const int N = g->num_players;
  int counts[15] = {0}, sums[SKO_MAXP], hidden[SKO_MAXP];
  /* skyjo.py:236-248: bincount over the discard pile (+ open cards when count_players_cards,
   * which is `not observe_other_player_indirect`, skyjo.py:160) */
  for (int i = 0; i < g->n_disc; i++) counts[g->discard_pile[i] + 2]++;
  if (!g->indirect)
    for (int p = 0; p < N; p++)
      for (int k = 0; k < 12; k++)
        if (g->players_masked[p][k] == 1) counts[g->players_cards[p][k] + 2]++;
  revealed_sums(g, sums, hidden);
  int min_sum = sums[0], min_hidden = hidden[0];
  for (int p = 1; p < N; p++) {
    if (sums[p] < min_sum) min_sum = sums[p];
    if (hidden[p] < min_hidden) min_hidden = hidden[p];
  }
  int top = g->n_disc ? g->discard_pile[g->n_disc - 1] : -3; /* skyjo.py:254 */
  int o = 0;
  obs[o++] = (int8_t)(min_sum < 127 ? min_sum : 127); /* skyjo.py:182 */
  obs[o++] = (int8_t)min_hidden;
  for (int i = 0; i < 15; i++) obs[o++] = (int8_t)counts[i];
  obs[o++] = (int8_t)top;
  obs[o++] = (int8_t)g->hand_card;
  if (g->indirect) { /* skyjo.py:259-277 */
    for (int k = 0; k < 12; k++)
      obs[o++] = g->players_masked[player][k] != 2 ? g->players_cards[player][k] : (int8_t)SKO_HAND_NONE;
  } else { /* skyjo.py:279-302: absolute seat order */
    for (int p = 0; p < N; p++)
      for (int k = 0; k < 12; k++)
        obs[o++] = g->players_masked[p][k] != 2 ? g->players_cards[p][k] : (int8_t)SKO_HAND_NONE;
  }
  /* skyjo.py:201-224: queried player's row, global phase */
  if (g->exp_phase == 1) {
    for (int k = 0; k < 12; k++) mask[k] = g->players_masked[player][k] != 0;
    for (int k = 0; k < 12; k++) mask[12 + k] = g->players_masked[player][k] == 2;
    mask[24] = mask[25] = 0;
  } else {
    memset(mask, 0, 24);
    mask[24] = mask[25] = 1;
  }
}

/* skyjo.py:477-498 */
void sko_evaluate_game(const int8_t cards[][12], int num_players, int finisher, double penalty, double *score) {
  for (int p = 0; p < num_players; p++) {
    double s = 0.0;
    for (int c = 0; c < 4; c++) {
      const int8_t *t = &cards[p][3 * c];
      int mn = t[0], mx = t[0];
      for (int k = 1; k < 3; k++) {
        if (t[k] < mn) mn = t[k];
        if (t[k] > mx) mx = t[k];
      }
      if (mn != mx) s += (double)(t[0] + t[1] + t[2]);
    }
    score[p] = s;
  }
  double mn = score[0];
  for (int p = 1; p < num_players; p++)
    if (score[p] < mn) mn = score[p];
  if (mn != score[finisher]) score[finisher] *= penalty;
}

/* skyjo.py:337-374 */
static int action_draw_card(sko_game *g, int player, int draw_from) {
  int done = 1; /* skyjo.py:471-475 */
  for (int k = 0; k < 12; k++)
    if (g->players_masked[player][k] == 2) done = 0;
  if (done) {
    g->is_terminated = 1;
    sko_evaluate_game((const int8_t(*)[12])g->players_cards, g->num_players, player, g->score_penalty,
                      g->final_score);
    return 1;
  }
  if (draw_from == 24) {
    if (g->n_draw == 0) { /* skyjo.py:361-365: whole discard pile incl. its top */
      if (g->rng.mode == SKO_RNG_PHILOX) px_session(&g->rng, g->episode - 1, g->reshuffles, 1);
      int8_t tmp[SKO_NCARDS + 8];
      int n = g->n_disc;
      memcpy(tmp, g->discard_pile, (size_t)n);
      reshuffle_discard_pile(g, tmp, n);
      g->reshuffles++;
      g->reshuffles_total++;
    }
    g->hand_card = g->drawpile[--g->n_draw];
  } else {
    g->hand_card = g->discard_pile[--g->n_disc];
  }
  next_action(g);
  return 0;
}

/* skyjo.py:376-427 with :431-469 */
static int action_place(sko_game *g, int player, int a) {
  int8_t *cards = g->players_cards[player], *masked = g->players_masked[player];
  if (a < 12) {
    g->discard_pile[g->n_disc++] = cards[a];
    masked[a] = 1;
    cards[a] = (int8_t)g->hand_card;
  } else {
    int pos = a - 12;
    if (masked[pos] != 2) return SKO_ERR_REVEALED;
    g->discard_pile[g->n_disc++] = (int8_t)g->hand_card;
    masked[pos] = 1;
  }
  int updated = 0;
  for (int c = 0; c < 4; c++) {
    int8_t *t = &cards[3 * c], *m = &masked[3 * c];
    if (t[0] == t[1] && t[1] == t[2] && m[0] == 1 && m[1] == 1 && m[2] == 1) {
      m[0] = m[1] = m[2] = 0;
      /* skyjo.py:454-458: the slice appended to the discard pile is the ZEROED mask */
      g->discard_pile[g->n_disc++] = 0, g->discard_pile[g->n_disc++] = 0, g->discard_pile[g->n_disc++] = 0;
      t[0] = t[1] = t[2] = SKO_REFUNDED;
      updated = 1;
    }
  }
  if (updated) g->num_refunded[player]++;
  g->num_placed[player]++;
  g->hand_card = SKO_HAND_NONE;
  next_action(g);
  return 0;
}

/* skyjo.py:308-335 */
int sko_act(sko_game *g, int player, int action) {
  if (g->exp_player != player) return SKO_ERR_PLAYER;
  if (action < 0 || action > 25) return SKO_ERR_RANGE;
  if (g->is_terminated) return 1;
  if (action >= 24) {
    if (g->hand_card != SKO_HAND_NONE) return SKO_ERR_HAS_HAND;
    return action_draw_card(g, player, action);
  }
  if (g->hand_card == SKO_HAND_NONE) return SKO_ERR_NO_HAND;
  return action_place(g, player, action);
}

/* numpy pairwise summation as used by np.mean on a contiguous float64 vector (n <= 128) */
static double np_sum_f64(const double *a, int n) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; i++) res += a[i];
    return res;
  }
  double r[8];
  for (int j = 0; j < 8; j++) r[j] = a[j];
  int i;
  for (i = 8; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; j++) r[j] += a[i + j];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; i++) res += a[i];
  return res;
}

/* skyjo_env.py:293-312 */
void sko_final_rewards(const sko_game *g, double mean_reward, double reward_refunded, double *out) {
  const int N = g->num_players;
  double mean = np_sum_f64(g->final_score, N) / (double)N;
  for (int p = 0; p < N; p++) {
    double r = (-g->final_score[p] + mean) + mean_reward;
    if (reward_refunded != 0.0) r += (double)g->num_refunded[p] * reward_refunded;
    out[p] = r;
  }
}

void sko_set_state(sko_game *g, const int8_t *cards, const int8_t *masked, const int8_t *draw, int n_draw,
                   const int8_t *disc, int n_disc, int hand, int player, int phase) {
  for (int p = 0; p < g->num_players; p++) {
    memcpy(g->players_cards[p], cards + 12 * p, 12);
    memcpy(g->players_masked[p], masked + 12 * p, 12);
  }
  memcpy(g->drawpile, draw, (size_t)n_draw);
  g->n_draw = n_draw;
  memcpy(g->discard_pile, disc, (size_t)n_disc);
  g->n_disc = n_disc;
  g->hand_card = hand;
  g->exp_player = player, g->exp_phase = phase;
  g->is_terminated = 0;
  memset(g->num_refunded, 0, sizeof(g->num_refunded));
  memset(g->num_placed, 0, sizeof(g->num_placed));
}

/* ===================================================================================
 * vector layer
 * =================================================================================== */
sko_vec *sko_vec_create(int num_envs, int num_players, double score_penalty, int indirect, double mean_reward,
                        double reward_refunded, int rng_mode, int auto_reset, uint64_t game_id0) {
  if (num_envs <= 0 || num_players <= 0 || num_players > SKO_MAXP) return NULL;
  sko_vec *v = (sko_vec *)calloc(1, sizeof(sko_vec));
  v->num_envs = num_envs, v->num_players = num_players, v->indirect = indirect ? 1 : 0;
  v->rng_mode = rng_mode, v->auto_reset = auto_reset;
  v->score_penalty = score_penalty, v->mean_reward = mean_reward, v->reward_refunded = reward_refunded;
  v->illegal_reward = -1.0; /* skyjo_env.py:23 */
  v->game_id0 = game_id0;
  v->games = (sko_game *)calloc((size_t)num_envs, sizeof(sko_game));
  v->done = (uint8_t *)calloc((size_t)num_envs, 1);
  v->status = (uint8_t *)calloc((size_t)num_envs, 1);
  v->rewards = (double *)calloc((size_t)num_envs * (size_t)num_players, sizeof(double));
  v->ep_len = (uint32_t *)calloc((size_t)num_envs, sizeof(uint32_t));
  v->acc_score = (double *)calloc((size_t)num_envs * (size_t)num_players, sizeof(double));
  v->acc_refunded = (double *)calloc((size_t)num_envs * (size_t)num_players, sizeof(double));
  for (int i = 0; i < num_envs; i++) sko_init(&v->games[i], num_players, score_penalty, indirect, rng_mode);
  return v;
}

void sko_vec_destroy(sko_vec *v) {
  if (!v) return;
  free(v->games), free(v->done), free(v->status), free(v->rewards), free(v->ep_len), free(v->acc_score), free(v->acc_refunded), free(v);
}

static void vec_new_episode(sko_vec *v, int i) {
  v->done[i] = 0;
  v->ep_len[i] = 0;
  memset(&v->rewards[(size_t)i * v->num_players], 0, sizeof(double) * (size_t)v->num_players);
}

/* game i is seeded like SkyjoGame.set_seed(seeds[i]) (default seeds[i] = base + global game id) */
void sko_vec_seed(sko_vec *v, const uint64_t *seeds, uint64_t base) {
  for (int i = 0; i < v->num_envs; i++) {
    sko_set_seed(&v->games[i], seeds ? seeds[i] : base + v->game_id0 + (uint64_t)i);
    vec_new_episode(v, i);
    v->status[i] = SKO_ST_RESET;
  }
}

/* set_seed(value) for one game of the batch (skyjo.py:84-88); the others are untouched */
void sko_vec_seed_one(sko_vec *v, int i, uint64_t value) {
  sko_set_seed(&v->games[i], value);
  vec_new_episode(v, i);
  v->status[i] = SKO_ST_RESET;
  v->resets++; /* one deal consumed, like a reset (the whole-batch seed starts the counters afresh instead) */
}

void sko_vec_reset(sko_vec *v, const uint8_t *mask) {
  for (int i = 0; i < v->num_envs; i++)
    if (!mask || mask[i]) {
      sko_reset(&v->games[i]);
      vec_new_episode(v, i);
      v->status[i] = SKO_ST_RESET;
      v->resets++;
    }
}

static void vec_step_one(sko_vec *v, int i, int action, uint64_t *steps, uint64_t *episodes, uint64_t *illegal,
                         uint64_t *resets, uint64_t *sum_len) {
  sko_game *g = &v->games[i];
  const int N = v->num_players;
  double *rew = &v->rewards[(size_t)i * N];
  if (action == SKO_ACTION_SKIP) return; /* the game is left exactly as it is (include/skyjo_vec.h SKYJO_ACTION_SKIP) */
  if (v->done[i]) {
    if (v->auto_reset) {
      sko_reset(g);
      vec_new_episode(v, i);
      v->status[i] = SKO_ST_RESET;
      (*resets)++;
    } else {
      v->status[i] = SKO_ST_NOOP_DONE;
    }
    return;
  }
  int8_t obs[19 + 12 * SKO_MAXP], mask[26];
  int cur = g->exp_player;
  sko_observe(g, cur, obs, mask);
  if (action < 0 || action > 25 || !mask[action]) {
    /* TerminateIllegalWrapper(illegal_reward=-1), skyjo_env.py:23: offender -1, others 0, all done */
    for (int p = 0; p < N; p++) rew[p] = p == cur ? v->illegal_reward : 0.0;
    v->done[i] = 1;
    v->status[i] = SKO_ST_ILLEGAL;
    (*illegal)++;
    return;
  }
  int over = sko_act(g, cur, action);
  v->status[i] = SKO_ST_OK;
  v->ep_len[i]++;
  (*steps)++;
  if (over > 0) { /* skyjo_env.py:242-247 */
    sko_final_rewards(g, v->mean_reward, v->reward_refunded, rew);
    for (int p = 0; p < N; p++) /* per-seat statistics of the finished episodes (the engine's sum_score / sum_refunded) */
      v->acc_score[(size_t)i * N + p] += g->final_score[p], v->acc_refunded[(size_t)i * N + p] += (double)g->num_refunded[p];
    v->done[i] = 1;
    (*episodes)++;
    (*sum_len) += v->ep_len[i];
  }
}

void sko_vec_step(sko_vec *v, const int32_t *actions, int threads) {
  uint64_t steps = 0, episodes = 0, illegal = 0, resets = 0, sum_len = 0;
  (void)threads;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) reduction(+ : steps, episodes, illegal, resets, sum_len) schedule(static)
  for (int i = 0; i < v->num_envs; i++) vec_step_one(v, i, actions[i], &steps, &episodes, &illegal, &resets, &sum_len);
  v->steps += steps, v->episodes += episodes, v->illegal += illegal, v->resets += resets, v->sum_len += sum_len;
}

void sko_vec_observe(const sko_vec *v, const int32_t *players, int8_t *obs, int8_t *mask, uint8_t *agent,
                     uint8_t *phase) {
  const int D = v->indirect ? 31 : 19 + 12 * v->num_players;
  for (int i = 0; i < v->num_envs; i++) {
    const sko_game *g = &v->games[i];
    int p = players ? players[i] : g->exp_player;
    sko_observe(g, p, obs + (size_t)i * D, mask + (size_t)i * 26);
    if (agent) agent[i] = (uint8_t)g->exp_player;
    if (phase) phase[i] = (uint8_t)g->exp_phase;
  }
}

/* uniform choice over legal actions == policy_ra's p = mask / sum(mask)
 * (random_admissible_policy.py:26-28), drawn from Philox4x32-10 keyed by the policy seed:
 * ctr = (iter >> 2, game id lo, game id hi, 'POL\0'), word iter & 3, k = mulhi(word, n_legal). */
int sko_policy_action(uint64_t policy_seed, uint64_t game_id, uint64_t iter, const int8_t *mask) {
  uint32_t ctr[4] = {(uint32_t)(iter >> 2), (uint32_t)game_id, (uint32_t)(game_id >> 32), 0x504F4C00u};
  uint32_t key[2] = {(uint32_t)policy_seed, (uint32_t)(policy_seed >> 32)}, out[4];
  sko_philox4x32_10(ctr, key, out);
  int n = 0;
  for (int a = 0; a < 26; a++) n += mask[a] != 0;
  if (n == 0) return 24;
  int k = (int)(((uint64_t)out[iter & 3] * (uint64_t)n) >> 32);
  for (int a = 0; a < 26; a++)
    if (mask[a] && k-- == 0) return a;
  return 24;
}

/* The same, and what the engine's record of every iteration holds (include/skyjo_vec.h: observation and action mask of the
 * player expected next, skyjo.py:148-224; agent, phase, done, status; steps applied in the episode), observed AFTER the
 * iteration's step: obs_out [iters][B][D], mask_out [iters][B][26], meta_out [iters][B][4], eplen_out [iters][B]. */
void sko_vec_rollout_rec(sko_vec *v, int iters, uint64_t policy_seed, int32_t *actions_out, int8_t *obs_out, int8_t *mask_out,
                         uint8_t *meta_out, uint16_t *eplen_out, int threads) {
  uint64_t steps = 0, episodes = 0, illegal = 0, resets = 0, sum_len = 0;
  const uint64_t iter0 = v->iter;
  const int D = v->indirect ? 31 : 19 + 12 * v->num_players;
  const size_t B = (size_t)v->num_envs;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) reduction(+ : steps, episodes, illegal, resets, sum_len) schedule(static)
  for (int i = 0; i < v->num_envs; i++) {
    for (int t = 0; t < iters; t++) {
      int a = -1;
      if (!v->done[i]) {
        int8_t obs[19 + 12 * SKO_MAXP], mask[26];
        sko_observe(&v->games[i], v->games[i].exp_player, obs, mask);
        a = sko_policy_action(policy_seed, v->game_id0 + (uint64_t)i, iter0 + (uint64_t)t, mask);
      }
      vec_step_one(v, i, a, &steps, &episodes, &illegal, &resets, &sum_len);
      if (actions_out) actions_out[(size_t)t * B + i] = a;
      if (obs_out) {
        const sko_game *g = &v->games[i];
        const size_t at = (size_t)t * B + (size_t)i;
        sko_observe(g, g->exp_player, obs_out + at * (size_t)D, mask_out + at * 26);
        meta_out[at * 4 + 0] = (uint8_t)g->exp_player, meta_out[at * 4 + 1] = (uint8_t)g->exp_phase;
        meta_out[at * 4 + 2] = v->done[i], meta_out[at * 4 + 3] = v->status[i];
        eplen_out[at] = (uint16_t)v->ep_len[i];
      }
    }
  }
  v->iter += (uint64_t)iters;
  v->steps += steps, v->episodes += episodes, v->illegal += illegal, v->resets += resets, v->sum_len += sum_len;
}

void sko_vec_rollout(sko_vec *v, int iters, uint64_t policy_seed, int32_t *actions_out, int threads) {
  sko_vec_rollout_rec(v, iters, policy_seed, actions_out, NULL, NULL, NULL, NULL, threads);
}
