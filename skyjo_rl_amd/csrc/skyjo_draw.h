// skyjo_draw.h - the two device functions that the environment's translation unit (skyjo_device.h: on-device policy, k_sample)
// and the policy net's (skyjo_policy.hip: the masked draw in the net's epilogue) share: Philox4x32-10 and the masked
// categorical draw of TorchActionMaskModel.forward (rlskyjo/models/action_mask_model.py:58-74).  Both translation units
// compile this text, so that "net + draw in one launch" and "net, then k_sample" give the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/skyjo_vec.h"

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0, c1 = l1, c2 = n2, c3 = l0;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
  o0 = c0, o1 = c1, o2 = c2, o3 = c3;
}


// One game's draw: masked = logits + clamp(log(mask), FLOAT_MIN) (action_mask_model.py:70-71), softmax, inverse CDF of
// the Philox uniform of (seed, ticket, game).  Two forms of the SAME arithmetic: sk_draw_action (one lane holds the 26 logits:
// k_sample) and sk_draw_action_pair (the two lanes that hold a game's logits after the net's last MFMA: skyjo_policy.hip).  The
// order of the float32 sums is part of the definition, so that "net + draw in one launch" and "net, then k_sample" give the same
// bits: the exponentials are summed in BLOCKS of four actions (block j = actions 4j .. 4j + 3, left to right; block 6 = actions
// 24, 25), the block totals left to right (P[j + 1] = P[j] + B[j], sum = P[7]), and the CDF inside block j starts from P[j].  The
// action is the smallest k WITH A NON-ZERO PROBABILITY whose CDF value exceeds u * sum.  (A block's CDF starts from a total that was
// rounded on another path than the running sum of the block before: it can lie one ulp above it, and without the condition a masked
// action at the head of a block was drawn once in ~ 10^7 draws - tests/test_gpu_sampler.py draws 2 x 10^8.)
#define SK_DRAW_FLOAT_MIN (-3.4028234663852886e38f)  // torch.finfo(float32).min == ray's FLOAT_MIN
__device__ __forceinline__ float sk_draw_uniform(uint64_t seed, uint64_t ticket, uint64_t gid) {
  uint32_t u0, u1, u2, u3;
  philox4x32_10((uint32_t)ticket, (uint32_t)(ticket >> 32), (uint32_t)gid, 0x53414D50u ^ (uint32_t)(gid >> 32), (uint32_t)seed,
                (uint32_t)(seed >> 32), u0, u1, u2, u3);
  return (float)(u0 >> 8) * (1.0f / 16777216.0f);  // 24 bits -> [0, 1)
}

__device__ __forceinline__ int sk_draw_action(const float *row, const uint32_t *mw, int no_masking, uint64_t seed, uint64_t ticket,
                                              uint64_t gid, float *logp_out, float *uniform_out) {
  float m[SKYJO_NUM_ACTIONS], mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < SKYJO_NUM_ACTIONS; k++) {
    const bool on = no_masking || ((mw[k >> 2] >> ((k & 3) * 8)) & 0xffu) != 0;
    m[k] = on ? row[k] : row[k] + SK_DRAW_FLOAT_MIN;  // log(1) = 0, clamp(log(0)) = FLOAT_MIN
    mx = fmaxf(mx, m[k]);
  }
  constexpr int NB = (SKYJO_NUM_ACTIONS + 3) / 4;
  float e[SKYJO_NUM_ACTIONS], P[NB + 1];
  P[0] = 0.f;
#pragma unroll
  for (int j = 0; j < NB; j++) {
    float b = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (4 * j + i < SKYJO_NUM_ACTIONS) {
        e[4 * j + i] = __expf(m[4 * j + i] - mx);
        b = i ? b + e[4 * j + i] : e[4 * j + i];
      }
    P[j + 1] = P[j] + b;
  }
  const float sum = P[NB];
  const float u = sk_draw_uniform(seed, ticket, gid);
  const float target = u * sum;
  int a = -1, last_on = 0;
#pragma unroll
  for (int j = 0; j < NB; j++) {
    float acc = P[j];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int k = 4 * j + i;
      if (k < SKYJO_NUM_ACTIONS) {
        acc += e[k];
        last_on = e[k] > 0.f ? k : last_on;
        a = (a < 0 && e[k] > 0.f && acc > target) ? k : a;
      }
    }
  }
  a = a < 0 ? last_on : a;  // (rounding at the very top of the distribution)
  if (logp_out) {
    float ma = m[0];
#pragma unroll
    for (int k = 1; k < SKYJO_NUM_ACTIONS; k++) ma = a == k ? m[k] : ma;
    *logp_out = (ma - mx) - __logf(sum);
  }
  if (uniform_out) *uniform_out = u;
  return a;
}

// The pair form.  After the net's last MFMA (32 x 32 accumulator tile, transposed) lane l and lane l ^ 32 hold one game's outputs:
// half hh = lane >> 5 has v[r] = output (r & 3) + 8 (r >> 2) + 4 hh, that is blocks 2q + hh (q = r >> 2) of the definition above -
// hh = 0: blocks 0, 2, 4, 6 (14 actions), hh = 1: blocks 1, 3, 5 (12).  mw4[q]: the mask word of block 2q + hh.  Every lane of the
// wavefront calls this (the halves exchange eight values); both lanes of a pair return the action, *logp_out likewise.
__device__ __forceinline__ int sk_draw_action_pair(const float (&v)[16], const uint32_t (&mw4)[4], const int hh, int no_masking, uint64_t seed,
                                                   uint64_t ticket, uint64_t gid, float *logp_out) {
  static_assert(SKYJO_NUM_ACTIONS == 26, "the pair form is laid out for 26 actions: seven blocks, the last one of two");
  float m[16], mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int q = r >> 2, i = r & 3;
    if (q == 3 && i >= 2) continue;  // (outputs 26 .. 31 are padding)
    const bool on = no_masking || ((mw4[q] >> (i * 8)) & 0xffu) != 0;
    m[r] = on ? v[r] : v[r] + SK_DRAW_FLOAT_MIN;
    if (q == 3) m[r] = hh ? -INFINITY : m[r];  // (hh = 1 has no block 7)
    mx = fmaxf(mx, m[r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float e[16], B[4], OB[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
#pragma unroll
    for (int i = 0; i < (q == 3 ? 2 : 4); i++) {
      e[4 * q + i] = __expf(m[4 * q + i] - mx);
      B[q] = i ? B[q] + e[4 * q + i] : e[4 * q + i];
    }
    OB[q] = __shfl_xor(B[q], 32, 64);
  }
  float P[8];
  P[0] = 0.f;
#pragma unroll
  for (int j = 0; j < 7; j++) P[j + 1] = P[j] + (((j & 1) == hh) ? B[j >> 1] : OB[j >> 1]);
  const float sum = P[7];
  const float target = sk_draw_uniform(seed, ticket, gid) * sum;
  int a = 99, last_on = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    float acc = hh ? P[(2 * q + 1) & 7] : P[2 * q];
#pragma unroll
    for (int i = 0; i < (q == 3 ? 2 : 4); i++) {
      const int k = i + 8 * q + 4 * hh;
      const bool mine = q < 3 || hh == 0;
      acc += e[4 * q + i];
      last_on = (mine && e[4 * q + i] > 0.f) ? k : last_on;
      a = (a == 99 && mine && e[4 * q + i] > 0.f && acc > target) ? k : a;
    }
  }
  a = min(a, __shfl_xor(a, 32, 64));  // the smallest k of either half (a half's own candidates are in increasing k)
  last_on = max(last_on, __shfl_xor(last_on, 32, 64));
  a = a == 99 ? last_on : a;
  if (logp_out) {
    const int rel = a - 4 * hh;
    float ma = -INFINITY;
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int i = 0; i < (q == 3 ? 2 : 4); i++) ma = rel == i + 8 * q ? m[4 * q + i] : ma;
    ma = fmaxf(ma, __shfl_xor(ma, 32, 64));
    *logp_out = (ma - mx) - __logf(sum);
  }
  return a;
}
