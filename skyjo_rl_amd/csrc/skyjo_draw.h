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
// the Philox uniform of (seed, ticket, game).  Shared by k_sample and by the policy net's epilogue (skyjo_policy.hip).
__device__ __forceinline__ int sk_draw_action(const float *row, const uint32_t *mw, int no_masking, uint64_t seed, uint64_t ticket,
                                              uint64_t gid, float *logp_out, float *uniform_out) {
  const float FLOAT_MIN = -3.4028234663852886e38f;  // torch.finfo(float32).min == ray's FLOAT_MIN
  float m[SKYJO_NUM_ACTIONS], mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < SKYJO_NUM_ACTIONS; k++) {
    const bool on = no_masking || ((mw[k >> 2] >> ((k & 3) * 8)) & 0xffu) != 0;
    m[k] = on ? row[k] : row[k] + FLOAT_MIN;  // log(1) = 0, clamp(log(0)) = FLOAT_MIN
    mx = fmaxf(mx, m[k]);
  }
  float e[SKYJO_NUM_ACTIONS], sum = 0.f;
#pragma unroll
  for (int k = 0; k < SKYJO_NUM_ACTIONS; k++) e[k] = __expf(m[k] - mx), sum += e[k];
  uint32_t u0, u1, u2, u3;
  philox4x32_10((uint32_t)ticket, (uint32_t)(ticket >> 32), (uint32_t)gid, 0x53414D50u ^ (uint32_t)(gid >> 32), (uint32_t)seed,
                (uint32_t)(seed >> 32), u0, u1, u2, u3);
  const float u = (float)(u0 >> 8) * (1.0f / 16777216.0f);  // 24 bits -> [0, 1)
  const float target = u * sum;
  float acc = 0.f;
  int a = -1, last_on = 0;
#pragma unroll
  for (int k = 0; k < SKYJO_NUM_ACTIONS; k++) {
    acc += e[k];
    last_on = e[k] > 0.f ? k : last_on;
    a = (a < 0 && acc > target) ? k : a;
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

