// skyjo_callers.h - part of skyjo_device.h (included from there, in its place: the parts build on each other in that order).
// Caller-side kernels: masked draw on given logits, episode-end columns, unpack, the two scoring helpers.
#pragma once
#ifndef SKYJO_DEVICE_PARTS
#error "include skyjo_device.h"
#endif

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
