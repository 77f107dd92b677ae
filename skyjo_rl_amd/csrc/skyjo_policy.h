// skyjo_policy.h - what the C ABI (skyjo_capi.hip) and the policy net's translation unit (skyjo_policy.hip) share: the
// packed-net descriptor, the draw descriptor and the one host entry point that launches the net kernels.  The kernels
// live in a translation unit of their own so that each side gets the instruction scheduler that suits it (the
// environment kernels gain 1 - 2 % under `-amdgpu-sched-strategy=max-ilp`, the net kernels lose 18 %: EXPERIMENTS r5 #11).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SKP_HIDDEN 256
#define SKP_IN 32    // 31 observation features + a constant 1 that carries the first layer's bias
#define SKP_OUT 32   // up to 32 outputs (26 logits, or 1 value)
#define SKP_WG 8     // wavefronts per workgroup: 8 x 32 games share one copy of the 256 x 256 layer in LDS
#define SKP_GAMES_PER_WG (32 * SKP_WG)
// 2 / ln 2: tanh(x) = 1 - 2 / (2^(x * 2 / ln 2) + 1); the hidden layers' biases are stored pre-multiplied (see SkMlpDev)
#define SKP_SCALE 2.8853900817779268f

// Optional epilogue of the policy branch: the masked categorical draw of k_sample on the logits that have just been
// computed, without their round trip through memory.
struct SkMlpDraw {
  int enable, mask_offset, no_masking;
  uint64_t seed, ticket, game_id0;
  int32_t *actions;
  float *logp;
};

struct SkMlpDev {
  const uint4 *w1;   // [8 m-tiles][2 k-steps][64 lanes] fragments, natural k order (k = feature; k = 31 carries the bias)
  const uint4 *w2;   // [8][16][64] fragments, accumulator k order
  const uint4 *w3;   // [1][16][64]
  const float *b2;   // [256] bias of layer 2, times SKP_SCALE (bf16: + W 1 of the r-fold); the kernels read a lane's eight rows once per
                     // launch and feed them to the matrix pipe as a 17th k-step (bf16 pairs against ones: skyjo_policy.hip)
  const float *b3;   // [1][64 lanes][16 regs] bias of layer 3 in accumulator layout (read the same way: row (r & 3) + 8 (r >> 2) + 4 h)
  int out_dim;
  // float32-grade mode (SKYJO_MLP_FP32): every weight is the sum of two bf16 values, w = hi + lo; w1 / w2 / w3 above hold
  // the high halves, these the low halves in the same fragment layout
  int split;
  const uint4 *w1l, *w2l, *w3l;
};

// Records as the engine writes them: row-major (record g at g * rec_bytes) or tile-planar (SKYJO_REC_TILE_PLANAR: piece p of
// game 64 t + l at  t * 64 * rec_bytes + p * 1024 + l * 16).
struct SkMlpRecords {
  const uint8_t *base;
  int rec_bytes, obs_dim, planar;
  long long n;
};

// One launch of the policy net (nets == 2: policy and value branch over the same records, grid.y = 2) in the nets'
// precision.  e0 / e1: events recorded around the kernel (hipExtLaunchKernelGGL), or null.  Returns a hipError_t.
int sk_launch_mlp(const SkMlpDev &a, const SkMlpDev &b, int nets, const SkMlpRecords &r, float *out_a, const SkMlpDraw &draw,
                  float *out_b, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
