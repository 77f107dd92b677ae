// skyjo_policy.h - config 5 caller: the fully connected net of the action-mask policy model
// (rlskyjo/models/action_mask_model.py:41-52 builds RLlib's TorchFC: obs -> 256 tanh -> 256 tanh -> outputs) as ONE
// gfx950 kernel on the matrix cores, reading the observation bytes straight out of the engine's records.
//
// Orientation: everything is computed transposed, H_next^T = W^T * H^T, with the 32 games of a wavefront on the lanes
// (MFMA column index) and the hidden units on the accumulator registers (row index).  A 32x32 accumulator tile of
// v_mfma_f32_32x32x16_bf16 can then be fed to the next layer as the B operand without any lane movement or LDS: its
// registers 8s .. 8s+7, packed to bf16, ARE the fragment of k-step s - in a permuted k order (element j of lane half h
// is row 16s + 8(j>>2) + 4h + (j&3) of the tile), which the weights follow: they are packed on the host, once, into
// exactly the per-lane fragments the kernel loads (one 16-byte load per lane and MFMA).
// Lane maps (gfx950): A[row l&31][k = 8(l>>5)+j], B[k = 8(l>>5)+j][col l&31], C[row (r&3)+8(r>>2)+4(l>>5)][col l&31].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SKP_HIDDEN 256
#define SKP_IN 32    // 31 observation features + a constant 1 that carries the first layer's bias
#define SKP_OUT 32   // up to 32 outputs (26 logits, or 1 value)
#ifndef SKP_UG
#define SKP_UG 2     // output tiles of layer 2 that are accumulated side by side (2 / 4 / 8: 32.7 / 37.2 / 41.6 us per
                     // 65 536 records: beyond 2 the kernel needs more than 256 VGPRs and loses its second wavefront per SIMD)
#endif

typedef __bf16 skp_bf16x8 __attribute__((ext_vector_type(8)));
typedef float skp_f32x16 __attribute__((ext_vector_type(16)));

// Optional epilogue of the policy branch: the masked categorical draw of k_sample on the logits that have just been
// computed, without their round trip through memory.
struct SkMlpDraw {
  int enable, mask_offset, no_masking;
  uint64_t seed, ticket, game_id0;
  int32_t *actions;
  float *logp;
};

struct SkMlpDev {
  const uint4 *w1;   // [8 m-tiles][2 k-steps][64 lanes] fragments, natural k order (k = feature)
  const uint4 *w2;   // [8][16][64] fragments, accumulator k order
  const uint4 *w3;   // [1][16][64]
  const float *b2;   // [8][64 lanes][16 regs] bias of layer 2 in accumulator layout
  const float *b3;   // [1][64][16]
  int out_dim;
  // float32-grade mode (SKYJO_MLP_FP32): every weight is the sum of two bf16 values, w = hi + lo; w1 / w2 / w3 above hold
  // the high halves, these the low halves in the same fragment layout (k_mlp_forward_split)
  int split;
  const uint4 *w1l, *w2l, *w3l;
};

// tanh(x) = 1 - 2 / (e^(2x) + 1) on two values at a time: the multiply, the add and the final multiply-add are packed
// fp32 instructions (v_pk_*_f32, two values per issue slot), the exponential and the reciprocal are the hardware's
// approximations (v_exp_f32, v_rcp_f32: 1 ulp - the result is rounded to bf16 anyway).  An IEEE division here
// (v_div_scale x 2, v_rcp, four v_fma, v_div_fmas, v_div_fixup per value) made the activation 13 instructions per
// value and the whole kernel a third slower.  e^(2x) = inf for large x -> 1, 0 for very negative x -> -1.
typedef float skp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ skp_f32x2 skp_tanh2(skp_f32x2 x) {
  const skp_f32x2 y = x * 2.8853900817779268f;  // 2 / ln 2
  skp_f32x2 t;
  t.x = __builtin_amdgcn_exp2f(y.x), t.y = __builtin_amdgcn_exp2f(y.y);
  t = t + 1.0f;
  skp_f32x2 r;
  r.x = __builtin_amdgcn_rcpf(t.x), r.y = __builtin_amdgcn_rcpf(t.y);
  return __builtin_elementwise_fma(r, (skp_f32x2){-2.0f, -2.0f}, (skp_f32x2){1.0f, 1.0f});
}
__device__ __forceinline__ skp_bf16x8 skp_pack8(const skp_f32x16 &a, int s, bool act) {
  skp_bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    skp_f32x2 v = {a[8 * s + j], a[8 * s + j + 1]};
    if (act) v = skp_tanh2(v);
    r[j] = (__bf16)v.x, r[j + 1] = (__bf16)v.y;
  }
  return r;
}
__device__ __forceinline__ skp_bf16x8 skp_frag(const uint4 *p) {
  const uint4 q = *p;
  skp_bf16x8 r;
  __builtin_memcpy(&r, &q, 16);
  return r;
}

// What follows the last layer, for the 32 games of a wavefront: the outputs go to memory (out: float32 [n][out_dim], may be
// null) and - policy branch - the masked categorical draw is made on the logits in registers (sk_draw_action).
__device__ __forceinline__ void skp_finish(const skp_f32x16 &acc, const int lane, const long long g, const long long n, const int out_dim,
                                           float *out, const SkMlpDraw &draw, const uint8_t *rec, const int rec_bytes) {
  const int h = lane >> 5;
  if (out && g < n) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < out_dim) out[g * out_dim + row] = acc[r];
    }
  }
  if (draw.enable) {
    // a game's 32 outputs sit in two lanes (this one and lane ^ 32: rows 4h .. 4h+3 of every block of 8): swap halves
    float full[32];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float other = __shfl_xor(acc[r], 32, 64);
      const int blk8 = r >> 2, i4 = r & 3;
      full[8 * blk8 + i4] = h ? other : acc[r];      // rows 0..3 of the block belong to the h = 0 lane
      full[8 * blk8 + 4 + i4] = h ? acc[r] : other;  // rows 4..7 to the h = 1 lane
    }
    if (h == 0 && g < n) {
      const uint32_t *mp = (const uint32_t *)(rec + g * rec_bytes + draw.mask_offset);
      uint32_t mw[7];
#pragma unroll
      for (int k = 0; k < 7; k++) mw[k] = mp[k];
      float lp = 0.f;
      draw.actions[g] = sk_draw_action(full, mw, draw.no_masking, draw.seed, draw.ticket, draw.game_id0 + (uint64_t)g,
                                       draw.logp ? &lp : nullptr, nullptr);
      if (draw.logp) draw.logp[g] = lp;
    }
  }
}

// One wavefront = SKP_GT column tiles of 32 games: every weight fragment that is loaded feeds SKP_GT independent MFMAs
// (half the weight traffic per game at 2, and two accumulators in flight instead of one dependent chain).
// rec_bytes / obs_dim as in the engine's records (indirect observation: 31 int8 features).
#ifndef SKP_GT
#define SKP_GT 1  // (2, at one wavefront per SIMD: 31.3 vs 32.9 us per iteration of config 5 before the weights moved to LDS)
#endif
#ifndef SKP_WAVES
#define SKP_WAVES 2
#endif
// A launch may carry TWO nets over the same records (grid.y = 2): workgroups with blockIdx.y == 1 evaluate `net_b` into
// `out_b` (no draw) - the policy and the value branch of the action-mask model in one launch (skyjo_vec_mlp_act_value).
// A workgroup is SKP_WG wavefronts (eight: two per SIMD, one workgroup per CU) that share ONE copy of the 256 x 256 layer's
// fragments in LDS (128 KB): read from L2 by every wavefront they were 170 KB per 32 games, 350 MB per 65 536-game launch.
#ifndef SKP_WG
#define SKP_WG 8
#endif
__global__ __launch_bounds__(64 * SKP_WG) __attribute__((amdgpu_waves_per_eu(SKP_WAVES, SKP_WAVES))) void k_mlp_forward(SkMlpDev net_a, const uint8_t *rec, int rec_bytes, int obs_dim, long long n,
                                                     float *out_a, SkMlpDraw draw_a, SkMlpDev net_b, float *out_b) {
  __shared__ uint4 w2s[8 * 16 * 64];
  const bool second = blockIdx.y == 1;
  const SkMlpDev net = second ? net_b : net_a;
  float *out = second ? out_b : out_a;
  SkMlpDraw draw = draw_a;
  draw.enable = second ? 0 : draw_a.enable;
  for (int i = threadIdx.x; i < 8 * 16 * 64; i += 64 * SKP_WG) w2s[i] = net.w2[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
  const long long tile0 = ((long long)blockIdx.x * SKP_WG + (threadIdx.x >> 6)) * SKP_GT;
  long long g[SKP_GT];
  skp_bf16x8 x[SKP_GT][2];
#pragma unroll
  for (int c = 0; c < SKP_GT; c++) {
    g[c] = (tile0 + c) * 32 + col;
    // ---- input fragments: features 16s + 8h .. 16s + 8h + 7 of this lane's game, int8 -> bf16 (exact) ----
    uint32_t ob[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (g[c] < n) {
      const uint4 *r = (const uint4 *)(rec + g[c] * rec_bytes);
      const uint4 a = r[0], b = r[1];
      ob[0] = a.x, ob[1] = a.y, ob[2] = a.z, ob[3] = a.w, ob[4] = b.x, ob[5] = b.y, ob[6] = b.z, ob[7] = b.w;
    }
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int k0 = 16 * s + j, k1 = 16 * s + 8 + j;  // the feature index 16s + 8h + j for h = 0 / h = 1
        const float v0 = k0 < obs_dim ? (float)(int8_t)(ob[k0 >> 2] >> ((k0 & 3) * 8)) : (k0 == SKP_IN - 1 ? 1.0f : 0.0f);
        const float v1 = k1 < obs_dim ? (float)(int8_t)(ob[k1 >> 2] >> ((k1 & 3) * 8)) : (k1 == SKP_IN - 1 ? 1.0f : 0.0f);
        x[c][s][j] = (__bf16)(h ? v1 : v0);
      }
  }
  // ---- layer 1: 31 (+1) -> 256, tanh; the result tiles become the 16 k-step fragments of layer 2 ----
  skp_bf16x8 h1[SKP_GT][16];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    skp_f32x16 acc[SKP_GT];
#pragma unroll
    for (int c = 0; c < SKP_GT; c++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const skp_bf16x8 w = skp_frag(net.w1 + (u * 2 + s) * 64 + lane);
#pragma unroll
      for (int c = 0; c < SKP_GT; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x[c][s], acc[c], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < SKP_GT; c++) h1[c][2 * u] = skp_pack8(acc[c], 0, true), h1[c][2 * u + 1] = skp_pack8(acc[c], 1, true);
  }
  // ---- layer 2: 256 -> 256, tanh.  SKP_UG output tiles at a time: their MFMAs of a k-step are independent, so the
  // matrix pipe is not waiting for the previous result of the same accumulator ----
  skp_bf16x8 h2[SKP_GT][16];
#pragma unroll
  for (int ug = 0; ug < 8 / SKP_UG; ug++) {
    skp_f32x16 acc[SKP_UG][SKP_GT];
#pragma unroll
    for (int q4 = 0; q4 < SKP_UG; q4++) {
      const float4 *bp = (const float4 *)(net.b2 + ((size_t)(SKP_UG * ug + q4) * 64 + lane) * 16);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float4 b = bp[q];
#pragma unroll
        for (int c = 0; c < SKP_GT; c++)
          acc[q4][c][4 * q] = b.x, acc[q4][c][4 * q + 1] = b.y, acc[q4][c][4 * q + 2] = b.z, acc[q4][c][4 * q + 3] = b.w;
      }
    }
#pragma unroll
    for (int ks = 0; ks < 16; ks++) {
      skp_bf16x8 w[SKP_UG];
#pragma unroll
      for (int q4 = 0; q4 < SKP_UG; q4++) w[q4] = skp_frag(w2s + ((SKP_UG * ug + q4) * 16 + ks) * 64 + lane);
#pragma unroll
      for (int q4 = 0; q4 < SKP_UG; q4++)
#pragma unroll
        for (int c = 0; c < SKP_GT; c++) acc[q4][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[q4], h1[c][ks], acc[q4][c], 0, 0, 0);
    }
#pragma unroll
    for (int q4 = 0; q4 < SKP_UG; q4++)
#pragma unroll
      for (int c = 0; c < SKP_GT; c++)
        h2[c][2 * (SKP_UG * ug + q4)] = skp_pack8(acc[q4][c], 0, true), h2[c][2 * (SKP_UG * ug + q4) + 1] = skp_pack8(acc[q4][c], 1, true);
  }
  // ---- layer 3: 256 -> outputs (no activation) ----
  skp_f32x16 acc[SKP_GT];
  {
    const float4 *bp = (const float4 *)(net.b3 + (size_t)lane * 16);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 b = bp[q];
#pragma unroll
      for (int c = 0; c < SKP_GT; c++) acc[c][4 * q] = b.x, acc[c][4 * q + 1] = b.y, acc[c][4 * q + 2] = b.z, acc[c][4 * q + 3] = b.w;
    }
  }
#pragma unroll
  for (int ks = 0; ks < 16; ks++) {
    const skp_bf16x8 w = skp_frag(net.w3 + ks * 64 + lane);
#pragma unroll
    for (int c = 0; c < SKP_GT; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, h2[c][ks], acc[c], 0, 0, 0);
  }
#pragma unroll
  for (int c = 0; c < SKP_GT; c++) skp_finish(acc[c], lane, g[c], n, net.out_dim, out, draw, rec, rec_bytes);
}

// ------------------------------------------------------------------------------------------------------------------
// The float32-grade form (SKYJO_MLP_FP32).  The reference evaluates RLlib's TorchFC in float32
// (rlskyjo/models/action_mask_model.py:43-49); bf16 operands alone leave the logits 8e-2 away from it.  Here every operand
// of every product is the sum of two bf16 values - w = w_hi + w_lo, h = h_hi + h_lo, 16 significant bits each - and a
// product is three MFMAs into the same float32 accumulator, w_hi h_lo + w_lo h_hi + w_hi h_hi (w_lo h_lo, 2^-16 of the
// product, is left out): the logits and values agree with the float32 module to 1e-4 (tests/test_gpu_policy_net.py) at
// three times the matrix work of the bf16 form.  The observations are int8 and exact in one bf16, so layer 1 takes two.
//   * The activations stay in the accumulator layout as before; h1 (hi and lo: 128 registers) is held for all of layer
//     2, while h2 is never held: as soon as an output tile of layer 2 is through the tanh, its two k-steps of layer 3
//     are accumulated (layer 3 rides inside layer 2's loop), so the kernel keeps to 256 registers and two wavefronts
//     per SIMD.
//   * The 256 x 256 layer is 256 KB now (hi + lo) against 160 KB of LDS: the workgroup stages it in two halves of four
//     output tiles (hi + lo: 128 KB), with a barrier either side of the reload.
// tanh: the same v_exp_f32 / v_rcp_f32 form as above (1 ulp each: 2e-7 absolute on a value in [-1, 1]).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void skp_split8(const skp_f32x16 &a, int s, bool act, skp_bf16x8 &hi, skp_bf16x8 &lo) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    skp_f32x2 v = {a[8 * s + j], a[8 * s + j + 1]};
    if (act) v = skp_tanh2(v);
    const __bf16 h0 = (__bf16)v.x, h1 = (__bf16)v.y;
    hi[j] = h0, hi[j + 1] = h1;
    lo[j] = (__bf16)(v.x - (float)h0), lo[j + 1] = (__bf16)(v.y - (float)h1);
  }
}
#define SKP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
__global__ __launch_bounds__(64 * SKP_WG) __attribute__((amdgpu_waves_per_eu(SKP_WAVES, SKP_WAVES))) void k_mlp_forward_split(
    SkMlpDev net_a, const uint8_t *rec, int rec_bytes, int obs_dim, long long n, float *out_a, SkMlpDraw draw_a, SkMlpDev net_b, float *out_b) {
  __shared__ uint4 w2s[8 * 16 * 64];  // [4 tiles of this half][hi, lo][16 k-steps][64 lanes]
  const bool second = blockIdx.y == 1;
  const SkMlpDev net = second ? net_b : net_a;
  float *out = second ? out_b : out_a;
  SkMlpDraw draw = draw_a;
  draw.enable = second ? 0 : draw_a.enable;
  const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
  const long long g = ((long long)blockIdx.x * SKP_WG + (threadIdx.x >> 6)) * 32 + col;
  // ---- input fragments (exact in bf16) ----
  skp_bf16x8 x[2];
  {
    uint32_t ob[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (g < n) {
      const uint4 *r = (const uint4 *)(rec + g * rec_bytes);
      const uint4 a = r[0], b = r[1];
      ob[0] = a.x, ob[1] = a.y, ob[2] = a.z, ob[3] = a.w, ob[4] = b.x, ob[5] = b.y, ob[6] = b.z, ob[7] = b.w;
    }
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int k0 = 16 * s + j, k1 = 16 * s + 8 + j;
        const float v0 = k0 < obs_dim ? (float)(int8_t)(ob[k0 >> 2] >> ((k0 & 3) * 8)) : (k0 == SKP_IN - 1 ? 1.0f : 0.0f);
        const float v1 = k1 < obs_dim ? (float)(int8_t)(ob[k1 >> 2] >> ((k1 & 3) * 8)) : (k1 == SKP_IN - 1 ? 1.0f : 0.0f);
        x[s][j] = (__bf16)(h ? v1 : v0);
      }
  }
  // ---- layer 1: (hi + lo) weights x exact inputs ----
  skp_bf16x8 h1h[16], h1l[16];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    skp_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      acc = SKP_MFMA(skp_frag(net.w1l + (u * 2 + s) * 64 + lane), x[s], acc);
      acc = SKP_MFMA(skp_frag(net.w1 + (u * 2 + s) * 64 + lane), x[s], acc);
    }
    skp_split8(acc, 0, true, h1h[2 * u], h1l[2 * u]);
    skp_split8(acc, 1, true, h1h[2 * u + 1], h1l[2 * u + 1]);
  }
  // ---- layers 2 and 3 ----
  skp_f32x16 acc3;
  {
    const float4 *bp = (const float4 *)(net.b3 + (size_t)lane * 16);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 b = bp[q];
      acc3[4 * q] = b.x, acc3[4 * q + 1] = b.y, acc3[4 * q + 2] = b.z, acc3[4 * q + 3] = b.w;
    }
  }
#pragma unroll 1
  for (int half = 0; half < 2; half++) {
    __syncthreads();  // (everybody is through with the previous half)
    for (int i = threadIdx.x; i < 4 * 16 * 64; i += 64 * SKP_WG) {
      const int t = i >> 10, r = i & 1023;  // tile of this half, (k-step, lane)
      w2s[(2 * t) * 1024 + r] = net.w2[(size_t)(4 * half + t) * 1024 + r];
      w2s[(2 * t + 1) * 1024 + r] = net.w2l[(size_t)(4 * half + t) * 1024 + r];
    }
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < 4; t++) {
      const int u = 4 * half + t;
      skp_f32x16 acc;
      {
        const float4 *bp = (const float4 *)(net.b2 + ((size_t)u * 64 + lane) * 16);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float4 b = bp[q];
          acc[4 * q] = b.x, acc[4 * q + 1] = b.y, acc[4 * q + 2] = b.z, acc[4 * q + 3] = b.w;
        }
      }
      const uint4 *wh = w2s + (2 * t) * 1024 + lane, *wl = wh + 1024;
#pragma unroll
      for (int ks = 0; ks < 16; ks++) {
        const skp_bf16x8 a_hi = skp_frag(wh + ks * 64), a_lo = skp_frag(wl + ks * 64);
        acc = SKP_MFMA(a_hi, h1l[ks], acc);
        acc = SKP_MFMA(a_lo, h1h[ks], acc);
        acc = SKP_MFMA(a_hi, h1h[ks], acc);
      }
      // this tile's 32 hidden units are k-steps 2u and 2u + 1 of layer 3
#pragma unroll
      for (int s = 0; s < 2; s++) {
        skp_bf16x8 h2h, h2l;
        skp_split8(acc, s, true, h2h, h2l);
        const skp_bf16x8 a_hi = skp_frag(net.w3 + (2 * u + s) * 64 + lane), a_lo = skp_frag(net.w3l + (2 * u + s) * 64 + lane);
        acc3 = SKP_MFMA(a_hi, h2l, acc3);
        acc3 = SKP_MFMA(a_lo, h2h, acc3);
        acc3 = SKP_MFMA(a_hi, h2h, acc3);
      }
    }
  }
  skp_finish(acc3, lane, g, n, net.out_dim, out, draw, rec, rec_bytes);
}
