// skyjo_cycle.h - part of skyjo_device.h (included from there, in its place: the parts build on each other in that order).
// K_cycle (stepping and dealing side by side in every workgroup) and the statistics reduction.
#pragma once
#ifndef SKYJO_DEVICE_PARTS
#error "include skyjo_device.h"
#endif

// ------------------------------------------------------------------------------------------
// k_cycle: one kernel for one to sixteen dealing cycles - stepping and dealing side by side inside every workgroup (fused rollout,
// two to four players, either observation; PLANAR: the indirect observation's records leave tile-planar, SKYJO_OPT_RECORD_LAYOUT).
//
// A workgroup is S step wavefronts + S dealing wavefronts on one CU (S = 4 on a full chip: eight wavefronts, two per SIMD;
// S = 1 .. 3 for batches of up to 256 .. 768 tiles, every wavefront on a SIMD of its own).  The dealing wavefronts deal for the
// games of THEIR OWN workgroup's tiles, the run that the previous launch planned on its way out (the pipelined protocol of
// the two-stream form, unchanged: sk_plan_deals / sk_publish_deals / wait_deal_done; `deal_tag_run` is that run's id, 0 =
// nothing to deal in this launch).  So the hand-over of a dealt episode never leaves the CU: the two wavefronts share its
// vector L1 and its XCD's L2, and workgroup-scope release / acquire - a wait for the wavefront's own stores - is all it
// takes (P.wg_local).  The two-stream form has to use agent scope (the kernels' wavefronts may sit on different XCDs), i.e. a
// write-back / an invalidation of a whole L2 per wavefront: THAT, not the sharing of SIMDs, is what made dealing beside the
// step kernel a net loss on a full chip for three rounds (EXPERIMENTS.md round 4: two-stream form 25.7, the same without its
// cache maintenance 34.0, this kernel 36 - 37 x 10^9 env-steps/s against 30.5 in line).
//
// Roles on a full chip: the first wavefront to arrive on a SIMD steps, the second deals - one of each per SIMD, where they
// hide each other's latencies (two dealing wavefronts on one SIMD saturate its vector ALU: 32.5 instead of 35.8 x 10^9).  Tiles
// are claimed through LDS counters; nothing depends on how the hardware spreads the wavefronts.
// ------------------------------------------------------------------------------------------
#define SK_CYCLE_MAX_S 4
template <bool INDIRECT, int NP, bool PLANAR>
__global__ __launch_bounds__(2 * SK_CYCLE_MAX_S *SK_TILE) void k_cycle(SkParams Pin, uint8_t *rec_out, int32_t *act_out, int iters, uint64_t policy_seed,
                                                                        uint64_t iter0, uint32_t deal_tag_run, uint32_t lds_step_bytes,
                                                                        uint32_t lds_deal_bytes, int cycle_len) {
  extern __shared__ uint32_t lds_raw[];
  const uint32_t S = blockDim.x >> 7;  // step (= dealing) wavefronts per workgroup
  const uint32_t split = lds_deal_bytes >> 30;  // (diagnostic role splits, see below)
  const bool defer_ok = ((lds_deal_bytes >> 29) & 1u) == 0;  // (bit 29: the step regions have no room for deferred scoring)
  lds_deal_bytes &= 0x1fffffffu;
  uint32_t *claim = lds_raw + (((size_t)S * (lds_step_bytes + lds_deal_bytes)) >> 2);  // six words behind the 2 S regions
  const int lane = (int)(threadIdx.x & 63u);
  if (threadIdx.x < 6) claim[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t simd = (__builtin_amdgcn_s_getreg(63492) >> 4) & 3u;  // HW_REG_HW_ID[5:4]
  uint32_t role = 0, slot = 0;
  if (split == 0) {
    if (lane == 0) role = atomicAdd(&claim[2 + simd], 1u) & 1u;
    role = (uint32_t)__builtin_amdgcn_readfirstlane((int)role);
  } else {  // diagnostic: two of a kind per SIMD - by SIMD parity (1) or by SIMD pair (2)
    role = split == 1 ? simd & 1u : (simd >> 1) & 1u;
  }
  if (lane == 0) slot = atomicAdd(&claim[role], 1u);
  slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
  if (slot >= S) {  // the preferred role is taken S times over (always the case for S < 4, where every wavefront asks to step first)
    role ^= 1u;
    if (lane == 0) slot = atomicAdd(&claim[role], 1u);
    slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
  }
  const int unit = (int)(blockIdx.x * S + slot);
  if (unit >= Pin.tiles) {
    // a surplus wavefront of the last workgroup (tiles % S != 0): it has no tile, but it is a member of the workgroup - it meets the
    // others at every cycle-end barrier of a launch of several cycles (step and dealing wavefronts both pass (iters - 1) / cycle_len
    // of them), so that no barrier is ever executed by a part of the workgroup only
    for (int c = cycle_len > 0 ? (iters - 1) / cycle_len : 0; c > 0; c--) __syncthreads();
    return;
  }
  Pin.wg_local = 1u;  // the games of tile `unit` are dealt by dealing slot `slot` of THIS workgroup: hand-overs stay inside the CU
  if (role == 0) {
    step_body<INDIRECT, true, NP, PLANAR>(Pin, unit, lane, lds_raw + (size_t)slot * (lds_step_bytes >> 2), nullptr, rec_out, act_out, iters, policy_seed, iter0,
                                  nullptr, nullptr, nullptr, 0, cycle_len, NP >= 4 ? true : defer_ok);  // (four players: never without - the host does not ask for it)
  } else {
    // the run the previous launch planned (deal_tag_run, 0 = none), then - a launch of several dealing cycles - the runs its step
    // wavefronts plan at the cycle ends inside it: ids plan_new_tag, + 1, ... (the last one is left to the next launch)
    SkParams P = Pin;
    const int cycles = cycle_len > 0 ? (iters + cycle_len - 1) / cycle_len : 1;
    uint32_t tag = deal_tag_run, planned = Pin.plan_new_tag;
    for (int c = 0; c < cycles; c++) {
      if (tag) {
        P.deal_tag = tag;
        deal_body<NP>(P, 0, 3, unit, lane, lds_raw + (((size_t)S * lds_step_bytes + (size_t)slot * lds_deal_bytes) >> 2), c == 0);
      }
      if (c + 1 < cycles) {
#ifdef SK_TRACE
        const unsigned long long dw0 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();  // (the step wavefronts arrive when they have planned the next run)
#ifdef SK_TRACE
        if (lane == 0) P.stamps[((size_t)2 * P.tiles + unit) * 8 + 7] += __builtin_amdgcn_s_memtime() - dw0, P.stamps[((size_t)3 * P.tiles + unit) * 8 + 7] = 0;
#endif
        tag = planned, planned = sk_next_tag(planned);
      }
    }
  }
}

// per-seat sums over all games for skyjo_vec_get_counters (the hot path keeps per-game sums, no atomics)
__global__ void k_reduce_stats(SkParams P) {
  for (int kind = 0; kind < SK_ACC_KINDS; kind++)
    for (int p = 0; p < P.L.N; p++) {
      double a = 0.0;
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.tiles; i += gridDim.x * blockDim.x)
        a += P.acc_tile[((size_t)i * SK_ACC_KINDS + kind) * SKYJO_MAX_PLAYERS + p];
      for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
      if ((threadIdx.x & 63) == 0 && a != 0.0) atomicAdd(&P.counters->sum_score[kind * SKYJO_MAX_PLAYERS + p], a);  // the four arrays are contiguous
    }
  // order of SkCounters' leading fields: steps, episodes, illegal, resets, sum_len, reshuffles, iters, waits
  const int dst[7] = {0, 1, 2, 3, 4, 5, 7};
  for (int k = 0; k < 7; k++) {
    unsigned long long t = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.tiles; i += gridDim.x * blockDim.x)
      t += P.tile_counters[(size_t)i * 8 + k];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if ((threadIdx.x & 63) == 0 && t) atomicAdd(&(&P.counters->steps)[dst[k]], t);
  }
}
