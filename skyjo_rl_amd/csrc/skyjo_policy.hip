// skyjo_policy.hip - config 5 caller: the fully connected net of the action-mask policy model
// (rlskyjo/models/action_mask_model.py:41-52 builds RLlib's TorchFC: obs -> 256 tanh -> 256 tanh -> outputs) as gfx950
// kernels on the matrix cores, reading the observation bytes straight out of the engine's records (row-major or
// tile-planar).  A translation unit of its own (skyjo_policy.h says why).
//
// Orientation: everything is computed transposed, H_next^T = W^T * H^T, with the 32 games of a wavefront on the lanes
// (MFMA column index) and the hidden units on the accumulator registers (row index).  A 32x32 accumulator tile of
// v_mfma_f32_32x32x16_bf16 can then be fed to the next layer as the B operand without any lane movement or LDS: its
// registers 8s .. 8s+7, packed to bf16, ARE the fragment of k-step s - in a permuted k order (element j of lane half h
// is row 16s + 8(j>>2) + 4h + (j&3) of the tile), which the weights follow: they are packed on the host, once, into
// exactly the per-lane fragments the kernel loads (one 16-byte load per lane and MFMA).
// Lane maps (gfx950): A[row l&31][k = 8(l>>5)+j], B[k = 8(l>>5)+j][col l&31], C[row (r&3)+8(r>>2)+4(l>>5)][col l&31].
//
// Round 6: what the counters said about the round-2 kernels (profiles/r6_cfg5_pmc_round2_kernels.json: the matrix pipe busy 26 % / 43 % of
// the kernel, the vector ALU 50 % / 38 %, both at once 15 % / 7 %) is that a wavefront did its MFMAs and its activations in turn, and
// that the workgroup's barriers kept the two wavefronts of a SIMD in the same phase - the two pipes took turns.  Here every
// wavefront runs a software pipeline of its own: the 16 (48) MFMAs of output tile u + 1 are issued between the activations of
// tile u (one activation per MFMA gap: v_exp, v_add, v_rcp (+ v_fma, float32-grade) + half a v_cvt_pk = 22.5 / 26.5 issue cycles against the
// MFMA's 32 - the factor 2 / ln 2 is in the packed weights, the bias is the accumulator's initial value, bf16: 1 - 2 r is in the next layer's), layer 3 rides behind (two MFMAs per
// tile) and - float32-grade - layer 2's first chain runs inside layer 1's activations.  The weights of the 256 x 256 layer travel
// into LDS THROUGH REGISTERS (LDS-DMA needs none, but a piece costs the issuing wavefront 100 - 180 cycles, lands slowly, and every
// other vector-memory operation of the wavefront queues behind it: EXPERIMENTS.md round 6); a workgroup keeps them for `passes`
// batches of 256 games, so that a 65 536-game launch is ONE round of 256 workgroups.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skyjo_draw.h"
#include "skyjo_policy.h"

typedef __bf16 skp_bf16x8 __attribute__((ext_vector_type(8)));
typedef float skp_f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t skp_u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) float *skp_lds_f32;  // (an LDS pointer that stays one: ds_read, not flat_load)
#define SKP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define SKP_LDS32(p) ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(p))  // the LDS offset of a __shared__ object

__device__ __forceinline__ skp_bf16x8 skp_frag(const uint4 *p) {
  const uint4 q = *p;
  skp_bf16x8 r;
  __builtin_memcpy(&r, &q, 16);
  return r;
}
__device__ __forceinline__ skp_f32x16 skp_zero() {
  skp_f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; r++) z[r] = 0.f;
  return z;
}

// tanh(x) = 1 - 2 / (2^(x * 2 / ln 2) + 1): the exponential and the reciprocal are the hardware's approximations (v_exp_f32,
// v_rcp_f32: 1 ulp - 2e-7 absolute on a value in [-1, 1]); an IEEE division made the activation 13 instructions per value.
// The factor 2 / ln 2 is in the packed weights and biases of the two hidden layers (skyjo_vec_mlp_create), so an accumulator is the
// exponent as it stands.  2^y = inf for large x -> 1, 0 -> -1.  Plain (unpacked) float32 instructions on purpose: beside MFMAs a
// v_pk_*_f32 costs more than the two instructions it replaces (MI355X_MICROARCH.md, per-instruction constants).

// the 16-byte piece p of game g's record
__device__ __forceinline__ const uint4 *skp_piece(const SkMlpRecords &R, long long g, int p) {
  const uint8_t *b = R.planar ? R.base + (g >> 6) * (64LL * R.rec_bytes) + (long long)p * 1024 + (g & 63) * 16
                              : R.base + g * R.rec_bytes + (long long)p * 16;
  return (const uint4 *)b;
}

// the first 32 bytes of game g's record (zeros beyond the last game)
__device__ __forceinline__ void skp_record_load(const SkMlpRecords &R, long long g, uint32_t (&ob)[8]) {
  uint4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
  if (g < R.n) a = *skp_piece(R, g, 0), b = *skp_piece(R, g, 1);
  ob[0] = a.x, ob[1] = a.y, ob[2] = a.z, ob[3] = a.w, ob[4] = b.x, ob[5] = b.y, ob[6] = b.z, ob[7] = b.w;
}
// input fragments: features 16s + 8h .. 16s + 8h + 7 of this lane's game, int8 -> bf16 (exact); feature 31 is the constant 1
// (obs_dim <= 31: byte 31 of a record is never an observation).  Bytes from obs_dim on are cleared word-wise with masks that
// depend on obs_dim alone (eight scalar registers; a comparison per feature kept 32 scalar conditions alive through the kernel).
__device__ __forceinline__ void skp_inputs(const SkMlpRecords &R, const uint32_t (&obr)[8], int h, skp_bf16x8 x[2]) {
  uint32_t ob[8];
#pragma unroll
  for (int w = 0; w < 8; w++) {
    const int left = R.obs_dim - 4 * w;  // observation bytes in this word
    const uint32_t m = left >= 4 ? 0xffffffffu : left <= 0 ? 0u : (1u << (8 * left)) - 1u;
    ob[w] = obr[w] & m;
  }
  ob[7] = (ob[7] & 0x00ffffffu) | 0x01000000u;
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k0 = 16 * s + j, k1 = 16 * s + 8 + j;  // the feature index 16s + 8h + j for h = 0 / h = 1
      const float v0 = (float)(int8_t)(ob[k0 >> 2] >> ((k0 & 3) * 8)), v1 = (float)(int8_t)(ob[k1 >> 2] >> ((k1 & 3) * 8));
      x[s][j] = (__bf16)(h ? v1 : v0);
    }
}

// What follows the last layer, for the 32 games of a wavefront: the outputs go to memory (out: float32 [n][out_dim], may be
// null) and - policy branch - the masked categorical draw is made on the logits in registers (sk_draw_action).
__device__ __forceinline__ void skp_finish(const skp_f32x16 &acc, const int lane, const long long g, const SkMlpRecords &R, const int out_dim,
                                           float *out, const SkMlpDraw &draw) {
  const int h = lane >> 5;
  if (out && g < R.n) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < out_dim) out[g * out_dim + row] = acc[r];
    }
  }
  if (draw.enable) {
    // a game's 32 outputs sit in two lanes (this one and lane ^ 32: rows 4h .. 4h + 3 of every block of 8) - each half works on
    // the blocks of four actions it holds, eight values cross (sk_draw_action_pair: the same arithmetic, in the same order, as
    // k_sample's one-lane form)
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = acc[r];
    uint32_t mw4[4] = {0u, 0u, 0u, 0u};
    if (g < R.n) {
      // (this half's word offsets are the same in every batch: left alone the compiler computes them - as 64-bit pairs - ahead of
      // the batch loop and spills them, and a kernel with a private segment costs ~ 1 us more per launch.  hq is opaque.)
      int hq = h;
      asm volatile("" : "+v"(hq));
      const uint8_t *rec = (const uint8_t *)skp_piece(R, g, 0);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int off = draw.mask_offset + 4 * min(2 * q + hq, 6);  // (the mask starts on a 4-byte boundary: a word never straddles two pieces)
        mw4[q] = *(const uint32_t *)(rec + (R.planar ? (off >> 4) * 1024 + (off & 15) : off));
      }
    }
    float lp = 0.f;
    const int a = sk_draw_action_pair(v, mw4, h, draw.no_masking, draw.seed, draw.ticket, draw.game_id0 + (uint64_t)g, draw.logp ? &lp : nullptr);
    if (h == 0 && g < R.n) {
      draw.actions[g] = a;
      if (draw.logp) draw.logp[g] = lp;
    }
  }
}

// Diagnostic build (-DSKP_STAMPS, tools/dev/policy_stamps.py): s_memtime at the phase boundaries of every wavefront, into a buffer
// of its own that nothing else reads.  The shipped build has none.
#ifdef SKP_STAMPS
#define SKP_NSTAMP 32
__device__ unsigned long long skp_stamp_buf[4096 * SKP_NSTAMP];
#define SKP_STAMP(k) if (gridDim.y == 2) skp_stamp_buf[(size_t)skp_wave_id * SKP_NSTAMP + (k)] = __builtin_amdgcn_s_memtime()  // (two-net launches only)
#define SKP_RSTAMP(k) if (gridDim.y == 2) skp_stamp_buf[(size_t)skp_wave_id * SKP_NSTAMP + (k)] = __builtin_amdgcn_s_memrealtime()
#define SKP_STAMP_DECL const int skp_wave_id = (blockIdx.y * gridDim.x + blockIdx.x) * SKP_WG + (threadIdx.x >> 6)
extern "C" int skyjo_debug_policy_stamps(unsigned long long *out, int n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(skp_stamp_buf), (size_t)n_words * 8);
}
#else
#define SKP_STAMP(k)
#define SKP_RSTAMP(k)
#define SKP_STAMP_DECL
#endif

// Everything a launch needs, as ONE kernel argument: net[1] / out[1] belong to the second branch (blockIdx.y == 1: no draw).
struct SkMlpArgs {
  SkMlpDev net[2];
  float *out[2];
  SkMlpRecords R;
  SkMlpDraw draw;
  int passes;
};


// Two activated values -> one register of a bf16 fragment (v_cvt_pk_bf16_f32)
typedef __bf16 skp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float skp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t skp_pk(float a, float b) {
  const skp_bf16x2 p = __builtin_convertvector((skp_f32x2){a, b}, skp_bf16x2);
  uint32_t r;
  __builtin_memcpy(&r, &p, 4);
  return r;
}
__device__ __forceinline__ skp_bf16x8 skp_frag4(const uint32_t (&w)[4]) {
  skp_bf16x8 r;
  __builtin_memcpy(&r, w, 16);
  return r;
}

// The activation of one accumulator tile as a three-stage software pipeline over "gaps" (one gap = the issue slots between two
// MFMAs): gap i starts value i (multiply-add with scale and bias, v_exp_f32), continues value i - 1 (+ 1, v_rcp_f32) and finishes
// value i - 2 (1 - 2 r; every second gap the conversion of a pair) - three independent short chains per gap instead of one
// dependent chain of six.  A gap is closed by a scheduling barrier: the compiler may order the handful of instructions inside
// it, nothing moves across (left alone it issues a tile's MFMAs in one burst and the activations after them).
// Pure arithmetic floats freely between the barriers when the compiler linearises the unrolled body (it ended up behind all of
// them): every value that crosses a gap boundary is therefore "pinned" - an empty asm statement that redefines it - at the start
// of the gap that consumes it and at the end of the gap that produced it.  The pins and the barriers are ordered among
// themselves, so an operation sits between its input's pin and its output's.  No instruction is emitted for a pin.
#define SKP_PIN(v) asm volatile("" : "+v"(v))
struct SkpAct {
  float e[16];
  uint32_t w[8];
  float hprev;
};
// bf16 mode: what is packed is r = 1 / (2^y + 1), not tanh = 1 - 2 r - the next layer's packed weights are - 2 W and its bias
// b + W 1 (skyjo_vec_mlp_create), so the activation is v_exp, v_add, v_rcp and half a v_cvt_pk: 22.5 issue cycles per value
// (same-box A/B: net 30.6 -> 28.9 us).  r is what gets rounded to bf16 then: + 20 % on this mode's error against the float32 module,
// which the weights' rounding dominates - 0.0020 mean against the tolerance of 0.01 (tests/test_gpu_policy_net.py).  The
// float32-grade mode keeps 1 - 2 r: there the fold saves 1 % (both pipes bound it) and costs half as much again in error.
__device__ __forceinline__ void skp_act_gap(SkpAct &a, const int i, const skp_f32x16 &acc) {
  const bool sa = i < 16, sb = i >= 1 && i <= 16, pair = sb && ((i - 1) & 1);
  float ein = 0.f, eo = 0.f, ro = 0.f;
  uint32_t w = 0;
  if (sb) {
    ein = a.e[i - 1];
    SKP_PIN(ein);
  }
  if (sa) eo = __builtin_amdgcn_exp2f(acc[i]);  // (the accumulator IS (2 / ln 2) (w x + b): scaled weights, bias as its initial value)
  if (sb) {
    ro = __builtin_amdgcn_rcpf(ein + 1.0f);
    if (pair) w = skp_pk(a.hprev, ro);
  }
  if (sa) {
    SKP_PIN(eo);
    a.e[i] = eo;
  }
  if (pair) {
    SKP_PIN(w);
    a.w[(i - 1) >> 1] = w;
  } else if (sb) {
    SKP_PIN(ro);
    a.hprev = ro;
  }
}
#define SKP_GAP_END __builtin_amdgcn_sched_barrier(0)

// ------------------------------------------------------------------------------------------------------------------
// bf16 mode (SKYJO_MLP_BF16): single bf16 weights and inter-layer activations, float32 accumulation.
// LDS: the 256 x 256 layer (128 KB), layer 3 (16 KB), layer 1 (16 KB) - all 160 KB; the biases come from memory.
// ------------------------------------------------------------------------------------------------------------------
// Stage U of layers 2 / 3: the 16 MFMAs of tile U + 1 (gaps 0 .. 15), layer 3's two k-steps of tile U - 1 (gaps 16, 17) and the bias
// of tile U + 2 (gap 18) between the activations of tile U.  a0 / a1: the weight fragments of the first two MFMAs (read two gaps
// ahead, like all the others).
// The biases are MFMAs too: a 17th k-step whose weight fragment holds (hi, lo) - the float32 bias as two bf16 - in k = 0, 1 against a
// fragment of ones gives hi + lo in every game's column, exact to 2^-17.  That is ONE register per tile and lane, held for the whole
// launch (bq), no LDS (the three layers fill all 160 KB) and no loads in here: a load from memory inside this pipeline costs the
// wavefront ~ 100 cycles, four per stage made every stage a third longer (EXPERIMENTS round 6, 24).
__device__ __forceinline__ skp_bf16x8 skp_bias_frag(const uint32_t w) {
  const uint32_t q[4] = {w, 0u, 0u, 0u};
  return skp_frag4(q);
}
template <int U>
__device__ __forceinline__ void skp_stage_bf16(const uint4 *w2s, const uint4 *w3s, const uint32_t (&bq)[9], const skp_bf16x8 &ones, const int lane,
                                               const skp_bf16x8 (&h1)[16], skp_f32x16 &cur, skp_f32x16 &acc3, skp_bf16x8 &f0, skp_bf16x8 &f1,
                                               skp_bf16x8 &a0, skp_bf16x8 &a1, skp_f32x16 &bias_next) {
  skp_f32x16 nxt = bias_next;  // (scaled) bias of tile U + 1: the chain's initial accumulator
  skp_bf16x8 a[20];
  a[0] = a0, a[1] = a1;
  SkpAct act;
  skp_f32x16 bn;
#pragma unroll
  for (int i = 0; i < 19; i++) {
    // LDS reads, two gaps ahead of their MFMA: the rest of chain U + 1, layer 3's fragments, the head of chain U + 2
    if (i + 2 < 16) {
      if (U < 7) a[i + 2] = skp_frag(w2s + ((U + 1) * 16 + i + 2) * 64 + lane);
    } else if (i + 2 < 18) {
      if (U > 0) a[i + 2] = skp_frag(w3s + (2 * (U - 1) + (i + 2 - 16)) * 64 + lane);
    } else if (i + 2 < 20 && U < 6) {
      a[i + 2] = skp_frag(w2s + ((U + 2) * 16 + (i - 16)) * 64 + lane);
    }
    if (i < 16) {
      if (U < 7) nxt = SKP_MFMA(a[i], h1[i], nxt);
    } else if (i < 18) {
      if (U > 0) acc3 = SKP_MFMA(a[i], i == 16 ? f0 : f1, acc3);
    } else if (U < 6) {
      bn = SKP_MFMA(skp_bias_frag(bq[U + 2]), ones, skp_zero());
    }
    skp_act_gap(act, i, cur);
    SKP_GAP_END;
  }
  f0 = skp_frag4((const uint32_t(&)[4])act.w[0]), f1 = skp_frag4((const uint32_t(&)[4])act.w[4]);
  cur = nxt;
  if (U < 6) a0 = a[18], a1 = a[19], bias_next = bn;
}

// One batch of 256 games (32 per wavefront).  FIRST: the workgroup's first batch, which also brings the weights of layers 2 and 3
// into LDS; `more`: another batch follows (its record is requested behind layer 1).
template <bool FIRST>
__device__ __forceinline__ void skp_batch_bf16(const SkMlpDev &net, const SkMlpRecords &R, float *const out, const SkMlpDraw &draw, uint4 *w2s,
                                               uint4 *w3s, uint4 *w1s, const long long batch, const bool more, const int lane, const int wave,
                                               const int col, const int h, uint32_t (&ob)[8], uint32_t (&bq)[9]) {
  SKP_STAMP_DECL;
  const long long g = (batch * SKP_WG + wave) * 32 + col;
  // ---- How a launch starts (stamped, EXPERIMENTS round 6, 24): the compute unit has ONE vector memory pipe, a 16-byte-per-lane load
  // keeps it busy for 16 cycles, and a wavefront's loads queue in it behind everybody else's.  Eight wavefronts that each requested
  // the record (2 loads), layer 1's sixteen fragments, layer 3's bias (4) and their 18 KiB of layers 2 / 3 at once waited 6 200
  // cycles for layer 1's first operand.  So: all three layers' weights live in LDS (128 + 16 + 16 KB: all of it), a wavefront
  // requests its record, its two KiB of layer 1 and nine bias values (4-byte loads), writes layer 1 to LDS - ONE workgroup barrier,
  // 2 500 cycles after the wavefront's first instruction: kernel arguments, then one round trip to memory - and starts; its 18 KiB
  // of layers 2 / 3 are requested three pieces per tile of layer 1 and written to LDS two tiles later (through registers: LDS-DMA
  // needs none, but a piece costs the issuing wavefront 100 - 180 cycles and the 144 KB took ~ 20 000 cycles to land; the
  // registers are free here, layer 1's results have not been produced yet).  The biases never touch LDS: skp_stage_bf16. ----
  skp_u32x4 stg[18];  // (a native vector type: HIP's uint4 is a struct of unions, and an array of them stays in scratch)
  float bf[9];
  if (FIRST) {
    const skp_u32x4 s0 = *(const skp_u32x4 *)(net.w1 + (2 * wave) * 64 + lane), s1 = *(const skp_u32x4 *)(net.w1 + (2 * wave + 1) * 64 + lane);
    // this lane's biases: output row 32 u + col of layer 2 (net.b2: float32 [256], scaled), row col of layer 3 (net.b3 is laid out
    // per lane and accumulator register: row (r & 3) + 8 (r >> 2) + 4 h' at [32 h'][r])
#pragma unroll
    for (int u = 0; u < 8; u++) bf[u] = net.b2[32 * u + col];
    bf[8] = net.b3[(32 * ((col >> 2) & 1)) * 16 + (col & 3) + 4 * (col >> 3)];
    *(skp_u32x4 *)(w1s + (2 * wave) * 64 + lane) = s0, *(skp_u32x4 *)(w1s + (2 * wave + 1) * 64 + lane) = s1;
    SKP_STAMP(2);
    __syncthreads();  // layer 1 is in LDS
    SKP_STAMP(16);
  }
  skp_bf16x8 x[2], w1f[16];
#pragma unroll
  for (int k = 0; k < 16; k++) w1f[k] = skp_frag(w1s + k * 64 + lane);
  skp_f32x16 acc3, cur, bias_next;
  skp_inputs(R, ob, h, x);
  SKP_STAMP(1 + 14 * (FIRST ? 0 : 1));
  // ---- layer 1: 31 (+1) -> 256, tanh (the bias rides in the product: feature 31 is 1); the result tiles become the 16
  // k-step fragments of layer 2.  Tile t + 1's two MFMAs sit in the first gaps of tile t's activations ----
  skp_bf16x8 h1[16];
  {
    skp_f32x16 acc = SKP_MFMA(w1f[0], x[0], skp_zero());
    acc = SKP_MFMA(w1f[1], x[1], acc);
    SKP_GAP_END;
#pragma unroll
    for (int t = 0; t < 8; t++) {
      SkpAct act;
      skp_f32x16 nxt;
#pragma unroll
      for (int i = 0; i < 18; i++) {
        if (t < 7 && i == 0) nxt = SKP_MFMA(w1f[2 * t + 2], x[0], skp_zero());
        if (t < 7 && i == 1) nxt = SKP_MFMA(w1f[2 * t + 3], x[1], nxt);
        // the wavefront's 18 KiB of layers 2 / 3: piece j requested in tile j / 3, written to LDS two tiles on.  (All eighteen requested
        // up front - eight wavefronts at once, right behind their barrier - kept every wavefront 2 500 cycles in the queue of the pipe.)
        if (FIRST && t < 6 && (i == 2 || i == 7 || i == 12)) {
          const int j = 3 * t + (i == 7) + 2 * (i == 12);
          stg[j] = *(const skp_u32x4 *)(j < 16 ? net.w2 + (j * 8 + wave) * 64 + lane : net.w3 + ((j - 16) * 8 + wave) * 64 + lane);
        }
        if (FIRST && t >= 2 && (i == 4 || i == 9 || i == 14)) {
          const int j = 3 * (t - 2) + (i == 9) + 2 * (i == 14);
          if (j < 16) *(skp_u32x4 *)(w2s + (j * 8 + wave) * 64 + lane) = stg[j];
          else *(skp_u32x4 *)(w3s + ((j - 16) * 8 + wave) * 64 + lane) = stg[j];
        }
        skp_act_gap(act, i, acc);
        SKP_GAP_END;
      }
      h1[2 * t] = skp_frag4((const uint32_t(&)[4])act.w[0]), h1[2 * t + 1] = skp_frag4((const uint32_t(&)[4])act.w[4]);
      acc = nxt;
    }
  }
  SKP_STAMP(3 + 14 * (FIRST ? 0 : 1));
  if (FIRST) {  // (before the next record is requested: the waits count in order, and behind a branch the compiler waits for everything)
#pragma unroll
    for (int u = 0; u < 9; u++) {
      const uint32_t hi = skp_pk(bf[u], 0.f);
      bq[u] = h ? 0u : skp_pk(bf[u], bf[u] - __uint_as_float(hi << 16));
    }
  }
  if (FIRST) __syncthreads();  // everybody's share of the weights is in LDS
  // the next batch's record, requested now: it arrives while layers 2 and 3 run
  if (more) skp_record_load(R, ((batch + 1) * SKP_WG + wave) * 32 + col, ob);
  SKP_STAMP(4 + 14 * (FIRST ? 0 : 1));
  // ---- layers 2 and 3 ----
  const uint32_t oq[4] = {h ? 0u : 0x3f803f80u, 0u, 0u, 0u};  // ones in k = 0, 1
  const skp_bf16x8 ones = skp_frag4(oq);
  cur = SKP_MFMA(skp_bias_frag(bq[0]), ones, skp_zero());
  acc3 = SKP_MFMA(skp_bias_frag(bq[8]), ones, skp_zero());
  bias_next = SKP_MFMA(skp_bias_frag(bq[1]), ones, skp_zero());
#pragma unroll
  for (int ks = 0; ks < 16; ks++) cur = SKP_MFMA(skp_frag(w2s + ks * 64 + lane), h1[ks], cur);
  skp_bf16x8 f0, f1, a0 = skp_frag(w2s + 16 * 64 + lane), a1 = skp_frag(w2s + 17 * 64 + lane);
  SKP_GAP_END;
  SKP_STAMP(5 + 14 * (FIRST ? 0 : 1));
  skp_stage_bf16<0>(w2s, w3s, bq, ones, lane, h1, cur, acc3, f0, f1, a0, a1, bias_next);
  SKP_STAMP(6 + 14 * (FIRST ? 0 : 1));
  skp_stage_bf16<1>(w2s, w3s, bq, ones, lane, h1, cur, acc3, f0, f1, a0, a1, bias_next);
  SKP_STAMP(7 + 14 * (FIRST ? 0 : 1));
  skp_stage_bf16<2>(w2s, w3s, bq, ones, lane, h1, cur, acc3, f0, f1, a0, a1, bias_next);
  SKP_STAMP(8 + 14 * (FIRST ? 0 : 1));
  skp_stage_bf16<3>(w2s, w3s, bq, ones, lane, h1, cur, acc3, f0, f1, a0, a1, bias_next);
  SKP_STAMP(9 + 14 * (FIRST ? 0 : 1));
  skp_stage_bf16<4>(w2s, w3s, bq, ones, lane, h1, cur, acc3, f0, f1, a0, a1, bias_next);
  SKP_STAMP(10 + 14 * (FIRST ? 0 : 1));
  skp_stage_bf16<5>(w2s, w3s, bq, ones, lane, h1, cur, acc3, f0, f1, a0, a1, bias_next);
  SKP_STAMP(11 + 14 * (FIRST ? 0 : 1));
  skp_stage_bf16<6>(w2s, w3s, bq, ones, lane, h1, cur, acc3, f0, f1, a0, a1, bias_next);
  SKP_STAMP(12 + 14 * (FIRST ? 0 : 1));
  skp_stage_bf16<7>(w2s, w3s, bq, ones, lane, h1, cur, acc3, f0, f1, a0, a1, bias_next);
  acc3 = SKP_MFMA(skp_frag(w3s + 14 * 64 + lane), f0, acc3);
  acc3 = SKP_MFMA(skp_frag(w3s + 15 * 64 + lane), f1, acc3);
  SKP_STAMP(13 + 14 * (FIRST ? 0 : 1));
  skp_finish(acc3, lane, g, R, net.out_dim, out, draw);
  SKP_STAMP(14 + 14 * (FIRST ? 0 : 1));
}

__global__ __launch_bounds__(64 * SKP_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_net_bf16(const SkMlpArgs A) {
  __shared__ uint4 w2s[8 * 16 * 64];  // the 256 x 256 layer (128 KB)
  __shared__ uint4 w3s[16 * 64];      // layer 3 (16 KB)
  __shared__ uint4 w1s[16 * 64];      // layer 1 (16 KB): 160 KB, all of the compute unit's LDS
  // (the branch's descriptor is read from the kernel-argument segment by index: one set of scalar registers, not two and a select)
  const SkMlpDev &net = A.net[blockIdx.y];
  float *const out = A.out[blockIdx.y];
  const SkMlpRecords &R = A.R;
  const int passes = A.passes;
  SkMlpDraw draw = A.draw;
  draw.enable = blockIdx.y ? 0 : A.draw.enable;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), col = lane & 31, h = lane >> 5;
  SKP_STAMP_DECL;
  SKP_STAMP(0);
  SKP_RSTAMP(29);
  uint32_t ob[8];
  skp_record_load(R, ((long long)blockIdx.x * passes * SKP_WG + wave) * 32 + col, ob);
  uint32_t bq[9];
  skp_batch_bf16<true>(net, R, out, draw, w2s, w3s, w1s, (long long)blockIdx.x * passes, passes > 1, lane, wave, col, h, ob, bq);
#pragma unroll 1
  for (int pass = 1; pass < passes; pass++) {
    const long long batch = (long long)blockIdx.x * passes + pass;
    if (batch * SKP_GAMES_PER_WG >= R.n) break;  // (uniform: a workgroup's later batches may lie beyond the last game)
    // (everything that depends on the lane alone is the same in every batch: the compiler would compute it all ahead of the loop
    // and, 256 registers being in use, spill it - a private segment costs the launch ~ 1 us.  The lane number is opaque in here.)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    skp_batch_bf16<false>(net, R, out, draw, w2s, w3s, w1s, batch, pass + 1 < passes, lane_o, wave, lane_o & 31, lane_o >> 5, ob, bq);
  }
  SKP_RSTAMP(30);
}

// ------------------------------------------------------------------------------------------------------------------
// The float32-grade form (SKYJO_MLP_FP32).  The reference evaluates RLlib's TorchFC in float32
// (rlskyjo/models/action_mask_model.py:43-49); bf16 operands alone leave the logits 8e-2 away from it.  Here every operand
// of every product is the sum of two bf16 values - w = w_hi + w_lo, h = h_hi + h_lo, 16 significant bits each - and a
// product is three MFMAs into the same float32 accumulator, w_hi h_lo + w_lo h_hi + w_hi h_hi (w_lo h_lo, 2^-16 of the
// product, is left out): the logits and values agree with the float32 module to 1e-4 (tests/test_gpu_policy_net.py) at
// three times the matrix work of the bf16 form.  The observations are int8 and exact in one bf16, so layer 1 takes two.
//   * 48 (+ 1: the bias, skp_bias_split) MFMAs per output tile of layer 2 and 6 of layer 3 against 16 activations: the matrix pipe
//     bounds this form.  A stage is 55 gaps; value v's activation is spread over gaps 3v + 1 .. 3v + 3, a pair's split into hi and
//     lo over the three after.
//   * The 256 x 256 layer is 256 KB (hi + lo) against 160 KB of LDS: it passes through a ring of three 32 KB output tiles,
//     tile U + 3 travelling - through registers, one KiB per wavefront at a time - into the slot tile U has left while stage U
//     runs; ONE workgroup barrier per stage (everybody's share of the next tile is written, everybody is through with the slot
//     that is refilled next).
//   * Layers 1 and 3 stay in LDS for the whole launch (32 KB each, hi + lo): with the ring that is all 160 KB.  Layer 1's tiles'
//     activations overlap nothing (EXPERIMENTS round 6, 10 and 18).
// ------------------------------------------------------------------------------------------------------------------
struct SkpActS {
  float e[16], r[16], hv[16];
  uint32_t hi[8], lo[8];
  float da[8], db[8];
};
// The operations of gap g when value v's activation starts in gap G v + OFF (phases in consecutive gaps: v_exp, + 1 + v_rcp,
// 1 - 2 r) and the split of pair p = (2p, 2p + 1) into hi and lo follows in the three gaps after its second value.  (The bias is in
// the accumulator already: layer 1's rides in the product, layer 2's is a 17th k-step - skp_bias_split.)
template <int G, int OFF>
__device__ __forceinline__ void skp_act_split_gap(SkpActS &a, const int g, const skp_f32x16 &acc) {
  // (no loops over v or p here: g is a constant once the caller's gap loop is unrolled, and so is everything derived from it)
  const int ta = g - OFF, tb = ta - 1, tc = ta - 2;
  if (ta >= 0 && ta % G == 0 && ta / G < 16) {
    const int v = ta / G;
    float eo = __builtin_amdgcn_exp2f(acc[v]);
    SKP_PIN(eo);
    a.e[v] = eo;
  }
  if (tb >= 0 && tb % G == 0 && tb / G < 16) {
    const int v = tb / G;
    float ein = a.e[v];
    SKP_PIN(ein);
    float ro = __builtin_amdgcn_rcpf(ein + 1.0f);
    SKP_PIN(ro);
    a.r[v] = ro;
  }
  if (tc >= 0 && tc % G == 0 && tc / G < 16) {
    const int v = tc / G;
    float rin = a.r[v];
    SKP_PIN(rin);
    float hv = __builtin_fmaf(rin, -2.0f, 1.0f);
    SKP_PIN(hv);
    a.hv[v] = hv;
  }
  const int tp = g - OFF - 3 - G, tq = tp - 1, tr = tp - 2;  // pair p's three gaps start at G (2p + 1) + OFF + 3
  if (tp >= 0 && tp % (2 * G) == 0 && tp / (2 * G) < 8) {
    const int p = tp / (2 * G);
    float x0 = a.hv[2 * p], x1 = a.hv[2 * p + 1];
    SKP_PIN(x0);
    SKP_PIN(x1);
    uint32_t w = skp_pk(x0, x1);
    SKP_PIN(w);
    a.hv[2 * p] = x0, a.hv[2 * p + 1] = x1, a.hi[p] = w;
  }
  if (tq >= 0 && tq % (2 * G) == 0 && tq / (2 * G) < 8) {
    const int p = tq / (2 * G);
    uint32_t w = a.hi[p];
    float x0 = a.hv[2 * p], x1 = a.hv[2 * p + 1];
    SKP_PIN(w);
    SKP_PIN(x0);
    SKP_PIN(x1);
    float d0 = x0 - __uint_as_float(w << 16), d1 = x1 - __uint_as_float(w & 0xffff0000u);
    SKP_PIN(d0);
    SKP_PIN(d1);
    a.hi[p] = w, a.da[p] = d0, a.db[p] = d1;
  }
  if (tr >= 0 && tr % (2 * G) == 0 && tr / (2 * G) < 8) {
    const int p = tr / (2 * G);
    float d0 = a.da[p], d1 = a.db[p];
    SKP_PIN(d0);
    SKP_PIN(d1);
    uint32_t w = skp_pk(d0, d1);
    SKP_PIN(w);
    a.lo[p] = w;
  }
}
#define SKP_SPLIT_GAPS 55  // a stage of layers 2 / 3: G = 3, OFF = 1; the last gap holds the bias MFMA alone
#define SKP_L1_GAPS 21     // a tile of layer 1: G = 1, OFF = 0

// One output tile of the 256 x 256 layer in LDS: [hi, lo][16 k-steps][64 lanes] fragments = 32 KB; a ring of three of them.
// A wavefront's share of a tile on its way from memory: pieces 4 w .. 4 w + 3 of its 32 one-KiB pieces, through registers
// (LDS-DMA needs none, but a piece costs the issuing wavefront 100 - 180 cycles and arrives slowly - stamped in round 6 -, and
// every other vector-memory operation of the wavefront queues behind it).
#define SKP_TILE_U4 2048
struct SkpStage {
  skp_u32x4 r[2];
};
__device__ __forceinline__ void skp_stage_load(const SkMlpDev &net, const int tile, const int wave, const int lane, const int half, SkpStage &st) {
  const uint4 *src = ((wave >> 2) ? net.w2l : net.w2) + (size_t)(tile & 7) * 1024 + (wave & 3) * 256 + half * 128 + lane;
  st.r[0] = *(const skp_u32x4 *)src, st.r[1] = *(const skp_u32x4 *)(src + 64);
}
__device__ __forceinline__ void skp_stage_store(uint4 *slot_lane, const int wave, const int half, const int k, const SkpStage &st) {
  *(skp_u32x4 *)(slot_lane + wave * 256 + half * 128 + k * 64) = st.r[k];
}

// Stage U (55 gaps): the 48 MFMAs of tile U + 1 in gaps 0 .. 47 (k-step g / 3), the activations of tile U beside them (value v in
// gaps 3 v + 1 .., the last pair's low halves in gap 51), layer 3's six MFMAs of the same tile U in gaps 48 .. 53 - its fragments
// never outlive the stage -, tile U + 1's bias in gap 54; on the side, tile U + 3 (the next batch's tiles 0, 1 from U = 5 on) travels into the ring slot that
// tile U has just left, one KiB at a time.  The weight fragments (hi, lo) of a k-step are read three gaps ahead of its first
// MFMA (k-step 0's by the stage before: ah / al), layer 3's (from LDS) in gaps 45 and 48.
// This kernel has no registers to spare: the eight bias pairs of a row (one per output tile of layer 2) are spread over the TWO lanes
// that share the row - lane l < 32 keeps tiles 0 .. 3 (k = 0, 1 of the extra k-step), lane l + 32 tiles 4 .. 7 (k = 8, 9) - four
// registers; bq[4]: layer 3's pair (lower half).  The fragment of ones has its ones where the tile's pair is.
__device__ __forceinline__ skp_f32x16 skp_bias_split(const uint32_t (&bq)[5], const int u, const int h, const skp_f32x16 &acc) {
  // (built where they are used: the value is opaque, or the compiler keeps all nine fragment pairs - 72 registers - from before the batch
  // loop on and spills them)
  uint32_t w = bq[u == 8 ? 4 : (u & 3)];
  asm volatile("" : "+v"(w));
  const bool mine = u == 8 ? h == 0 : h == (u >> 2);
  const uint32_t a[4] = {mine ? w : 0u, 0u, 0u, 0u}, b[4] = {mine ? 0x3f803f80u : 0u, 0u, 0u, 0u};
  return SKP_MFMA(skp_frag4(a), skp_frag4(b), acc);
}
template <int U>
__device__ __forceinline__ void skp_stage_split(const SkMlpDev &net, uint4 *const (&rot)[3], const uint4 *w3s, const int lane,
                                                const int wave, const int h, const skp_bf16x8 (&h1h)[16], const skp_bf16x8 (&h1l)[16],
                                                skp_f32x16 &cur, skp_f32x16 &acc3, skp_bf16x8 &ah, skp_bf16x8 &al, const uint32_t (&bq)[5]) {
  const uint4 *wt = rot[(U + 1) % 3];  // tile U + 1 (this lane's fragment of k-step 0, hi)
  const uint4 *wn = rot[(U + 2) % 3];  // tile U + 2 (its k-step 0 is read on the way out)
  uint4 *const wr = rot[U % 3];        // tile U's slot: free since the barrier before this stage
  skp_f32x16 nxt = skp_zero();
  SkpActS act;
  skp_u32x4 st;
  skp_bf16x8 nh, nl, w3h, w3l;
#pragma unroll
  for (int g = 0; g < SKP_SPLIT_GAPS; g++) {
    const int ks = g / 3, part = g % 3;
    if (U < 7 && g < 48 && part == 0 && ks < 15) nh = skp_frag(wt + (ks + 1) * 64), nl = skp_frag(wt + 1024 + (ks + 1) * 64);
    // layer 3's fragments: k-step 2 U into (w3h, w3l), k-step 2 U + 1 into (nh, nl) - free since k-step 15's pair moved on in gap 44
    if (g == 45) w3h = skp_frag(w3s + (2 * U) * 64 + lane), w3l = skp_frag(w3s + 1024 + (2 * U) * 64 + lane);
    if (g == 48) nh = skp_frag(w3s + (2 * U + 1) * 64 + lane), nl = skp_frag(w3s + 1024 + (2 * U + 1) * 64 + lane);
    if (U < 7 && g >= 6 && g < 54 && (g - 6) % 12 == 0) {  // piece j of this wavefront's four of tile U + 3: requested in gap 6 + 12 j ...
      const int j = (g - 6) / 12;
      st = *(const skp_u32x4 *)(((wave >> 2) ? net.w2l : net.w2) + (size_t)((U + 3) & 7) * 1024 + (wave & 3) * 256 + j * 64 + lane);
    }
    if (U < 7 && g >= 16 && (g - 16) % 12 == 0) *(skp_u32x4 *)(wr + wave * 256 + ((g - 16) / 12) * 64) = st;  // ... written ten gaps on
    if (g < 48) {
      if (U < 7) {
        nxt = SKP_MFMA(part == 1 ? al : ah, part == 0 ? h1l[ks] : h1h[ks], nxt);
        if (part == 2 && ks < 15) ah = nh, al = nl;
      }
    } else if (g == 54) {  // tile U + 1's bias: the float32 value as a bf16 pair in a 17th k-step, against ones (skp_stage_bf16, skp_bias_split)
      if (U < 7) nxt = skp_bias_split(bq, U + 1, h, nxt);
    } else {
      // layer 3, k-steps 2 U (gaps 48 .. 50) and 2 U + 1 (gaps 51 .. 53: its low halves come last, they are ready in gap 51)
      const int s2 = (g - 48) / 3, k = (g - 48) % 3;
      const skp_bf16x8 fhi = skp_frag4((const uint32_t(&)[4])act.hi[4 * s2]), flo = skp_frag4((const uint32_t(&)[4])act.lo[4 * s2]);
      acc3 = SKP_MFMA(s2 ? (k == 1 ? nl : nh) : (k == 1 ? w3l : w3h), k == 2 ? flo : fhi, acc3);
    }
    if (U < 6 && g == 51) w3h = skp_frag(wn), w3l = skp_frag(wn + 1024);  // (tile U + 2's first pair, into the registers layer 3's first k-step has left)
    skp_act_split_gap<3, 1>(act, g, cur);
    SKP_GAP_END;
  }
  cur = nxt;
  if (U < 6) ah = w3h, al = w3l;
}

__global__ __launch_bounds__(64 * SKP_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_net_split(const SkMlpArgs A) {
  __shared__ uint4 ring[3 * SKP_TILE_U4];  // three output tiles of the 256 x 256 layer (96 KB)
  __shared__ uint4 w3s[2 * 16 * 64];       // layer 3, [hi, lo][16 k-steps][64 lanes] (32 KB)
  __shared__ uint4 w1s[2 * 16 * 64];       // layer 1, [hi, lo][8 tiles][2 k-steps][64 lanes] (32 KB): 160 KB, all of the CU's LDS
  // (the branch's descriptor is read from the kernel-argument segment by index: one set of scalar registers, not two and a select)
  const SkMlpDev &net = A.net[blockIdx.y];
  float *const out = A.out[blockIdx.y];
  const SkMlpRecords &R = A.R;
  const int passes = A.passes;
  SkMlpDraw draw = A.draw;
  draw.enable = blockIdx.y ? 0 : A.draw.enable;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), col = lane & 31, h = lane >> 5;
  SKP_STAMP_DECL;
  SKP_STAMP(0);
  SKP_RSTAMP(29);
  uint32_t ob[8];
  skp_record_load(R, ((long long)blockIdx.x * passes * SKP_WG + wave) * 32 + col, ob);
  float bf[5];
  {
    // what stays in LDS for the whole launch (layers 1 and 3) and the first two tiles of the ring; the biases of layers 2 / 3 stay in
    // registers (bq) and enter as MFMAs - no LDS left for them, and no load inside the pipeline (EXPERIMENTS round 6, 24)
    skp_u32x4 t1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) t1[j] = *(const skp_u32x4 *)(((wave >> 2) ? net.w1l : net.w1) + (wave & 3) * 256 + j * 64 + lane);
#pragma unroll
    for (int j = 0; j < 4; j++) bf[j] = net.b2[32 * (j + 4 * h) + col];
    bf[4] = net.b3[(32 * ((col >> 2) & 1)) * 16 + (col & 3) + 4 * (col >> 3)];
#pragma unroll
    for (int j = 0; j < 4; j++) *(skp_u32x4 *)(w1s + wave * 256 + j * 64 + lane) = t1[j];
    skp_u32x4 t3[4];
#pragma unroll
    for (int j = 0; j < 4; j++) t3[j] = *(const skp_u32x4 *)(((wave >> 2) ? net.w3l : net.w3) + (wave & 3) * 256 + j * 64 + lane);
#pragma unroll
    for (int j = 0; j < 4; j++) *(skp_u32x4 *)(w3s + wave * 256 + j * 64 + lane) = t3[j];
#pragma unroll
    for (int tile = 0; tile < 2; tile++) {
      SkpStage a, b;
      skp_stage_load(net, tile, wave, lane, 0, a);
      skp_stage_load(net, tile, wave, lane, 1, b);
#pragma unroll
      for (int k = 0; k < 2; k++) skp_stage_store(ring + tile * SKP_TILE_U4 + lane, wave, 0, k, a), skp_stage_store(ring + tile * SKP_TILE_U4 + lane, wave, 1, k, b);
    }
  }
  __syncthreads();
  // this lane's bias values as bf16 pairs (hi, lo: exact to 2^-17): skp_bias_split
  uint32_t bq[5];
#pragma unroll
  for (int j = 0; j < 5; j++) bq[j] = skp_pk(bf[j], bf[j] - __uint_as_float(skp_pk(bf[j], 0.f) << 16));
  bq[4] = h ? 0u : bq[4];
#pragma unroll 1
  for (int pass = 0; pass < passes; pass++) {
    const long long batch = (long long)blockIdx.x * passes + pass;
    if (batch * SKP_GAMES_PER_WG >= R.n && pass > 0) break;
    const bool more = pass + 1 < passes && (batch + 1) * SKP_GAMES_PER_WG < R.n;  // this workgroup has another batch
    const long long g = (batch * SKP_WG + wave) * 32 + col;
    // this batch's tile u lives in ring slot (8 pass + u) % 3: rot[u % 3] is this lane's fragment of its k-step 0 (hi)
    uint4 *rot[3];
#pragma unroll
    for (int k = 0; k < 3; k++) rot[k] = ring + ((8 * pass + k) % 3) * SKP_TILE_U4 + lane;
    // ---- layer 1: (hi + lo) weights x exact inputs, the four fragments of tile t + 2 requested while tile t is activated;
    // layer 2's first chain (output tile 0) follows it k-step by k-step: tile t - 1's two
    // k-steps between the activations of tile t.  Tile 2 travels into the ring on the side ----
    skp_bf16x8 x[2], w1f[4], w1n[4];  // [k-step][lo, hi] of the tile whose MFMAs come next / the one after
#pragma unroll
    for (int k = 0; k < 4; k++) w1f[k] = skp_frag(w1s + ((k & 1) ? 0 : 1024) + (k >> 1) * 64 + lane);
#pragma unroll
    for (int k = 0; k < 4; k++) w1n[k] = skp_frag(w1s + ((k & 1) ? 0 : 1024) + (2 + (k >> 1)) * 64 + lane);
    skp_inputs(R, ob, h, x);
    SKP_STAMP(1 + 14 * (pass > 0));
    skp_bf16x8 h1h[16], h1l[16];
    skp_f32x16 cur = skp_zero();
    {
      skp_f32x16 acc = skp_zero();
#pragma unroll
      for (int k = 0; k < 4; k++) acc = SKP_MFMA(w1f[k], x[k >> 1], acc);
      SKP_GAP_END;
      SkpStage st;
#pragma unroll
      for (int t = 0; t < 8; t++) {
        SkpActS act;
        skp_f32x16 nxt = skp_zero();
        skp_bf16x8 c0h, c0l, c1h, c1l;
        const int ks0 = 2 * (t - 1), ks1 = ks0 + 1;
#pragma unroll
        for (int gp = 0; gp < SKP_L1_GAPS; gp++) {
          if (t < 7 && gp < 4) nxt = SKP_MFMA(w1n[gp], x[gp >> 1], nxt);
          if (t < 6 && gp >= 4 && gp < 8)
            w1n[gp - 4] = skp_frag(w1s + (((gp - 4) & 1) ? 0 : 1024) + ((t + 2) * 2 + ((gp - 4) >> 1)) * 64 + lane);
          if (t >= 1) {
            if (gp == 2) c0h = skp_frag(rot[0] + ks0 * 64), c0l = skp_frag(rot[0] + 1024 + ks0 * 64);
            if (gp == 8) c1h = skp_frag(rot[0] + ks1 * 64), c1l = skp_frag(rot[0] + 1024 + ks1 * 64);
            if (gp >= 6 && gp < 9) cur = SKP_MFMA(gp == 7 ? c0l : c0h, gp == 6 ? h1l[ks0] : h1h[ks0], cur);
            if (gp >= 12 && gp < 15) cur = SKP_MFMA(gp == 13 ? c1l : c1h, gp == 12 ? h1l[ks1] : h1h[ks1], cur);
          }
          // tile 2's two halves: requested in tiles 0 / 4 of layer 1, written in tiles 3 / 7
          if ((t == 0 || t == 4) && gp == 10) skp_stage_load(net, 2, wave, lane, t == 4, st);
          if ((t == 3 || t == 7) && (gp == 16 || gp == 17)) skp_stage_store(rot[2], wave, t == 7, gp - 16, st);
          skp_act_split_gap<1, 0>(act, gp, acc);
          SKP_GAP_END;
        }
        h1h[2 * t] = skp_frag4((const uint32_t(&)[4])act.hi[0]), h1h[2 * t + 1] = skp_frag4((const uint32_t(&)[4])act.hi[4]);
        h1l[2 * t] = skp_frag4((const uint32_t(&)[4])act.lo[0]), h1l[2 * t + 1] = skp_frag4((const uint32_t(&)[4])act.lo[4]);
        acc = nxt;
      }
#pragma unroll
      for (int ks = 14; ks < 16; ks++) {
        const skp_bf16x8 ch = skp_frag(rot[0] + ks * 64), cl = skp_frag(rot[0] + 1024 + ks * 64);
        cur = SKP_MFMA(ch, h1l[ks], cur);
        cur = SKP_MFMA(cl, h1h[ks], cur);
        cur = SKP_MFMA(ch, h1h[ks], cur);
      }
      cur = skp_bias_split(bq, 0, h, cur);
    }
    SKP_STAMP(3 + 14 * (pass > 0));
    // ---- layers 2 and 3.  One barrier per stage: everybody's share of the tile that is read two stages on has been written, and
    // everybody is through with the slot that the next stage refills ----
    skp_f32x16 acc3 = skp_bias_split(bq, 8, h, skp_zero());
    SKP_GAP_END;
    __syncthreads();
    skp_bf16x8 ah = skp_frag(rot[1]), al = skp_frag(rot[1] + 1024);
    SKP_STAMP(5 + 14 * (pass > 0));
    skp_stage_split<0>(net, rot, w3s, lane, wave, h, h1h, h1l, cur, acc3, ah, al, bq);
    __syncthreads();
    SKP_STAMP(6 + 14 * (pass > 0));
    skp_stage_split<1>(net, rot, w3s, lane, wave, h, h1h, h1l, cur, acc3, ah, al, bq);
    __syncthreads();
    SKP_STAMP(7 + 14 * (pass > 0));
    skp_stage_split<2>(net, rot, w3s, lane, wave, h, h1h, h1l, cur, acc3, ah, al, bq);
    __syncthreads();
    SKP_STAMP(8 + 14 * (pass > 0));
    skp_stage_split<3>(net, rot, w3s, lane, wave, h, h1h, h1l, cur, acc3, ah, al, bq);
    __syncthreads();
    SKP_STAMP(9 + 14 * (pass > 0));
    skp_stage_split<4>(net, rot, w3s, lane, wave, h, h1h, h1l, cur, acc3, ah, al, bq);
    __syncthreads();
    SKP_STAMP(10 + 14 * (pass > 0));
    skp_stage_split<5>(net, rot, w3s, lane, wave, h, h1h, h1l, cur, acc3, ah, al, bq);
    __syncthreads();
    SKP_STAMP(11 + 14 * (pass > 0));
    skp_stage_split<6>(net, rot, w3s, lane, wave, h, h1h, h1l, cur, acc3, ah, al, bq);
    __syncthreads();
    if (more) skp_record_load(R, ((batch + 1) * SKP_WG + wave) * 32 + col, ob);
    SKP_STAMP(12 + 14 * (pass > 0));
    skp_stage_split<7>(net, rot, w3s, lane, wave, h, h1h, h1l, cur, acc3, ah, al, bq);
    SKP_STAMP(13 + 14 * (pass > 0));
    skp_finish(acc3, lane, g, R, net.out_dim, out, draw);
    SKP_STAMP(14 + 14 * (pass > 0));
  }
  SKP_RSTAMP(30);
}

int sk_launch_mlp(const SkMlpDev &a, const SkMlpDev &b, int nets, const SkMlpRecords &r, float *out_a, const SkMlpDraw &draw,
                  float *out_b, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  const long long batches = (r.n + SKP_GAMES_PER_WG - 1) / SKP_GAMES_PER_WG;
  // one round of workgroups where the batch allows it: a workgroup owns a CU (its LDS), 256 CUs, `nets` workgroups per batch
  int passes = (int)((batches * nets + 255) / 256);
  if (passes < 1) passes = 1;
  if (passes > 8) passes = 8;
  const dim3 grid((unsigned)((batches + passes - 1) / passes), (unsigned)nets), block(64 * SKP_WG);
  SkMlpArgs A;
  A.net[0] = a, A.net[1] = b, A.out[0] = out_a, A.out[1] = out_b, A.R = r, A.draw = draw, A.passes = passes;
  if (a.split)
    hipExtLaunchKernelGGL(k_net_split, grid, block, 0, s, e0, e1, 0, A);
  else
    hipExtLaunchKernelGGL(k_net_bf16, grid, block, 0, s, e0, e1, 0, A);
  return (int)hipGetLastError();
}
