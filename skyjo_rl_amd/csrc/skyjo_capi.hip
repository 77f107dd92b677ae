// skyjo_capi.hip - host side of libskyjo_vec.so: the extern "C" boundary declared in
// include/skyjo_vec.h, handle / memory management and kernel launches.  gfx950 only, no CPU path.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "skyjo_device.h"
#include "skyjo_policy.h"

// Measurement switches (environment variables: tools/dev/README.md) exist in -DSK_DIAG builds only; the shipped library reads no
// environment variable at all - what a caller or a test may choose is an option of skyjo_vec_set_option.
#ifdef SK_DIAG
static const char *sk_diag_env(const char *name) { return getenv(name); }
#else
static inline const char *sk_diag_env(const char *) { return nullptr; }
#endif

namespace {

thread_local std::string g_err;
std::atomic<uint64_t> g_generation{0};

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail(SKYJO_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));             \
  } while (0)

// Every entry point that takes a handle runs on the handle's device, whatever the calling thread's current device is
// (a second thread starts on device 0; torch.cuda.set_device may have switched it), and leaves the caller's current
// device as it found it.
struct DevGuard {
  int prev = -1;
  bool switched = false;
  explicit DevGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DevGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};
#define GUARD(h) DevGuard guard_((h)->cfg.device_id)

constexpr int kMaxRolloutChunk = 128;  // lockstep iterations per k_step launch (the tile stays in LDS for a whole launch)
constexpr int kMaxCyclesPerLaunch = 16;  // k_cycle: whole dealing cycles per launch (each of deal_every_iters iterations)
// Default number of lockstep iterations between two dealing runs.  A run adds one episode to every bank that is not
// full, so the interval has to stay below the mean episode length of the policy in use (random admissible policy:
// 76 / 105 / 134 steps for 2 / 3 / 4 players) or the banks of SK_BANK episodes drain and finished games deal in
// place (deal_inline: slow, same result).  Measured at 65 536 three-player games (round 2 kernels): 80 -> 23.8, 88 -> 24.5,
// 92 -> 24.9, 96 -> 25.0 x 10^9 steps/s; over 2 x 10^8 episodes 88 and 92 never emptied a bank, 96 did 35 times in
// 1.2 x 10^8 episodes: 88 it is (0.84 of the mean episode length, like 64 for two players).
// With the dealing kernel on its own stream a run's episodes arrive one interval later, so the interval is a step
// shorter there (the runs are hidden behind the step kernel anyway): 64 / 48 with the k_scan + k_publish form, whose
// caller's stream waits for every run; 80 / 56 in the pipelined form (the default), where a step launch should outlast
// the dealing kernel beside it (32 768 x 3 players: 64 -> 17.5, 72 -> 19.0, 80 -> 19.4, 88 -> 19.9 x 10^9 steps/s with one
// bank run dry; 4 096 x 2 players: 48 -> 2.3, 56 -> 2.7, 64 -> 2.9 x 10^9 with twelve).
// The one-kernel form (k_cycle, merged_s = its S): on a full chip (S = 4, a step and a dealing wavefront on every SIMD) a launch
// should last about as long as its dealing wavefronts - 65 536 x 3 players: 48 -> 32.7, 56 -> 36.0, 60 -> 37.1, 64 -> 37.3, 68 ->
// 36.8, 72 -> 36.5, 80 -> 35.7 x 10^9 steps/s; two players: 40 -> 28.3, 48 -> 30.3, 56 -> 31.3 (64 empties banks).  With S < 4 every
// wavefront has a SIMD to itself, the dealing is hidden anyway and longer launches win (32 768 x 3: 56 -> 22.4, 72 -> 23.2, 80 -> 23.4;
// 32 768 x 4: 64 -> 22.6, 80 -> 23.1, 96 -> 23.7, 112 -> 24.0; 4 096 x 2: 40 -> 2.44, 48 -> 2.88, 56 -> 2.98).
constexpr int deal_interval_default(int num_players, bool overlap, bool piped, int merged_s = 0) {
  if (merged_s >= 4) return num_players >= 3 ? 64 : 56;
  if (merged_s > 0) return num_players >= 4 ? 104 : (num_players == 3 ? 80 : 56);
  return overlap ? (piped ? (num_players >= 3 ? 80 : 56) : (num_players >= 3 ? 64 : 48)) : (num_players >= 3 ? 88 : 64);
}

}  // namespace

struct skyjo_vec_mlp {
  SkMlpDev net{};
  void *blob = nullptr;
  int device_id = 0, obs_dim = 0;
};

struct skyjo_vec {
  uint64_t generation = 0;  // unique per created handle in this process (snapshots remember it)
  skyjo_vec_config cfg{};
  SkParams P{};
  size_t G = 0;           // tiles * 64
  size_t lds_bytes = 0, lds_tile = 0;
  size_t lds_rollout = 0, lds_step = 0;  // the step kernels of 2 / 3 / 4 players keep their statistics in registers: smaller footprints (fused rollout / caller actions)
  bool seeded = false;
  int pending_iters = 0;  // lockstep iterations since the dealing kernel last ran
  int deal_every_iters = 64;  // set from deal_interval_default() in skyjo_vec_create
  // The interval adapts itself unless it was set explicitly: every dealing run reports how many banks it found empty
  // (host-mapped word, no synchronisation); any empty bank shortens the interval, a long calm stretch lengthens it
  // back towards the default.  Results never depend on the cadence (tests/test_gpu_parity.py), only the speed does.
  bool auto_interval = true;
  int interval_default = 64, calm_runs = 0;
  uint32_t health_seen = 0;
  uint32_t *health_host = nullptr;
  // dealing pipeline: k_scan / k_publish on the caller's stream, k_deal on deal_stream when overlap is on
  bool overlap = true;
  bool fused_scan = true;  // in line: k_deal looks at the banks itself (SKYJO_FUSED_SCAN=0: diagnostic, the k_scan + work list form)
  hipStream_t deal_stream = nullptr;
  hipEvent_t ev_scan = nullptr, ev_dealt = nullptr;
  bool deal_inflight = false;
  bool piped = true;           // beside the step kernel: the step kernel plans / publishes the runs itself (SKYJO_PIPELINED=0: k_scan + k_publish)
  // Full-chip batches of two / three players (indirect observation): ONE kernel per dealing cycle, k_cycle - eight wavefronts per
  // CU, the four on SIMDs 0 / 2 step, the four on SIMDs 1 / 3 deal the run the previous launch planned (the pipelined protocol,
  // no second stream).  cycle_deal_tag != 0: a planned run waits for the next launch to carry it.
  bool merged = false, merged_capable = false, prefer_merged = false;
  uint32_t cycle_deal_tag = 0;
  size_t lds_cycle = 0;
  int cycle_s = SK_CYCLE_MAX_S;  // step (= dealing) wavefronts per workgroup of k_cycle
  size_t lds_cycle_step = 0;     // one step wavefront's LDS region in k_cycle (lds_rollout, or lds_step when the deferred scoring has no room)
  bool cycle_no_defer = false;
  bool inflight_piped = false; // the run(s) in flight were planned that way
  int list_sel = 0;
  uint32_t deal_tag = 0;  // k_step launches between two k_deal launches inside skyjo_vec_rollout
  uint64_t iter = 0;        // rollout iterations (the policy's Philox counter)
  uint64_t iters_total = 0; // lockstep iterations of any kind since the counters were reset
  // Small batches (single-game views): the *_host conveniences go through host-mapped memory - the kernel reads the actions
  // from it and writes the records AND every game's packed state, rewards and scores to it (sk_export_raw), so a call is one
  // launch + one stream synchronisation, and skyjo_vec_get_state / get_rewards_host afterwards are served from `hm_raw`
  // (raw_valid) without touching the device.
  bool fast_host = false, raw_valid = false;
  bool host_spin = true;    // single-tile engines: spin on the kernel's sign-off word (SKYJO_NO_SPIN=1 at create time: synchronise the stream)
  bool raw_export = false;  // the kernel also writes every game's packed state to host-mapped memory (few tiles only, see skyjo_vec_create)
  uint8_t *hm_block = nullptr;  // one hipHostMalloc: actions | mask | records | raw
  int32_t *hm_actions = nullptr;
  uint8_t *hm_mask = nullptr, *hm_records = nullptr, *hm_raw = nullptr;
  int32_t *hm_actions_d = nullptr;
  uint8_t *hm_mask_d = nullptr, *hm_records_d = nullptr, *hm_raw_d = nullptr;
  int raw_stride = 0;
  uint32_t host_seq = 0;  // sequence number of the last host-style step of a single-tile engine (see skyjo_vec_step_host)
  uint32_t cycle_seq = 0;   // k_cycle launches so far: launch L counts empty banks into bank_empty[L & 1] and reports the other word
  int ncu = 256;            // compute units of the device (k_cycle's workgroup sizing)
  int max_cycles = 0;       // SKYJO_OPT_MAX_CYCLES_PER_LAUNCH (0: the ABI's maximum, kMaxCyclesPerLaunch)
  bool rec_planar_all = false;  // ... = SKYJO_REC_TILE_PLANAR_ALL: reset / observe / step / step_collect / model_rollout write their records tile-planar as well
  bool rec_planar = false;  // SKYJO_OPT_RECORD_LAYOUT: skyjo_vec_rollout writes its records tile-planar (one-kernel form, either observation)
  bool no_bank = false;  // SKYJO_OPT_NO_BANK: no pre-dealt episodes, every deal is made in place from the stream's position
  // lazily allocated scratch for the *_host conveniences
  int32_t *d_actions = nullptr;
  uint8_t *d_records = nullptr;
  uint8_t *d_mask = nullptr;
  std::vector<std::pair<void *, size_t>> allocs;  // every device array of the handle with its size (snapshots copy them all)
  // The arrays made at create time are carved out of ONE allocation, the large ones at 2 MB boundaries: how the generator
  // states, the banks and the live tiles lie relative to each other in the memory channels is then the same in every
  // process.  (With one hipMalloc per array it was not: about one process in twenty - one in three when processes of
  // different shapes alternate - ran the dealing kernel at 90 instead of 70 us for its whole life.)
  std::vector<void *> owned;  // what hipFree gets
  uint8_t *arena = nullptr;
  size_t arena_off = 0;
  int arena_mode = 0;  // 1: adding up, 2: carving
  // optional per-launch timing with HIP events on the launch stream (bench.py roofline leg)
  bool profile = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[SKYJO_PROF_KERNELS];  // k_step, k_scan, k_deal, k_publish, k_net_*
};

struct skyjo_vec_snapshot {
  const skyjo_vec *owner = nullptr;
  uint64_t owner_generation = 0;     // the handle's id (an address can be reused by a later handle)
  std::vector<size_t> array_bytes;   // size of every array in the blob, in the order of the handle's table
  int device_id = 0;
  void *blob = nullptr;
  size_t bytes = 0;
  // host side of the engine's state
  int pending_iters = 0, deal_every_iters = 0, calm_runs = 0, list_sel = 0;
  bool auto_interval = true;
  uint32_t health_seen = 0, deal_tag = 0, health[2] = {0, 0};
  uint64_t iter = 0, iters_total = 0;
};

namespace {

// The sticky device error (skyjo_device.h: SK_ERR_*) as the host-style kernels (step with caller actions, reset, observe)
// left it in the host-mapped word: valid after any synchronisation with the stream they ran on.  The fused rollout kernel does
// not write that word: dev_error_fetch() reads the device's own copy first (get_counters, check_error, snapshot_create).
int dev_error_check(const skyjo_vec *h) {
  if (h->health_host && (h->health_host[2] & SK_ERR_DEAL_TIMEOUT))
    return fail(SKYJO_E_DEVICE, "a step kernel gave up waiting for the dealing kernel that should run beside it (SKYJO_OPT_OVERLAP): "
                                "results since then are void; switch the option off or re-seed");
  return SKYJO_OK;
}

int dev_error_fetch(skyjo_vec *h, hipStream_t s) {
  uint32_t err = 0;
  HIPCHK(hipMemcpyAsync(&err, h->P.dev_error, sizeof(err), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  h->health_host[2] |= err;
  return dev_error_check(h);
}

template <class T>
int dalloc(skyjo_vec *h, T **out, size_t count, bool zero = true) {
  const size_t bytes = count * sizeof(T);
  void *p = nullptr;
  if (h->arena_mode) {
    const size_t al = bytes >= ((size_t)1 << 20) ? ((size_t)2 << 20) : 256;
    const size_t off = (h->arena_off + al - 1) & ~(al - 1);
    h->arena_off = off + bytes;
    if (h->arena_mode == 1) {
      *out = nullptr;
      return SKYJO_OK;
    }
    p = h->arena + off;
  } else {
    HIPCHK(hipMalloc(&p, bytes));
    h->owned.push_back(p);
  }
  if (zero) HIPCHK(hipMemset(p, 0, bytes));
  h->allocs.emplace_back(p, bytes);
  *out = (T *)p;
  return SKYJO_OK;
}

// Profiling: the kernel's own begin / end timestamps (the dispatch packet's completion signal, the same
// source rocprofv3's kernel trace reads) are attached to a pair of events by hipExtLaunchKernelGGL.
int prof_events(skyjo_vec *h, int kernel, hipEvent_t *a, hipEvent_t *b) {
  *a = *b = nullptr;
  if (!h->profile) return SKYJO_OK;
  HIPCHK(hipEventCreate(a));
  HIPCHK(hipEventCreate(b));
  h->ev[kernel].emplace_back(*a, *b);
  return SKYJO_OK;
}

// Make the episodes of the dealing launch that may still be running available (k_publish on the caller's stream).
static int launch_deal_kernel(skyjo_vec *h, hipStream_t ds, int mode, int be_read = -1);
// k_cycle form: a run that was planned but whose carrying launch never came (the caller went on with another kind of call) is
// dealt here, by the dealing kernel alone on the caller's stream.
int flush_cycle_deal(skyjo_vec *h, hipStream_t s) {
  if (!h->cycle_deal_tag) return SKYJO_OK;
  const uint32_t keep = h->P.deal_tag;
  h->P.deal_tag = h->cycle_deal_tag;
  h->cycle_deal_tag = 0;
  const int rc = launch_deal_kernel(h, s, 3, (int)((h->cycle_seq - 1u) & 1u));  // (the word the k_cycle launch that planned it counted into)
  h->P.deal_tag = keep;
  return rc;
}

int publish_deals(skyjo_vec *h, hipStream_t s) {
  int rcf = flush_cycle_deal(h, s);
  if (rcf) return rcf;
  if (!h->deal_inflight) return SKYJO_OK;
  if (h->overlap && !h->merged) HIPCHK(hipStreamWaitEvent(s, h->ev_dealt, 0));
  hipEvent_t e0, e1;
  int rc = prof_events(h, 3, &e0, &e1);
  if (rc) return rc;
  if (h->inflight_piped) {
    const int G = h->P.tiles * SK_TILE;
    hipExtLaunchKernelGGL(k_publish_all, dim3((G + 255) / 256), dim3(256), 0, s, e0, e1, 0, h->P);
  } else {
    hipExtLaunchKernelGGL(k_publish, dim3(64), dim3(256), 0, s, e0, e1, 0, h->P, h->list_sel);
  }
  HIPCHK(hipGetLastError());
  h->deal_inflight = false, h->inflight_piped = false;
  return SKYJO_OK;
}

// mode: 0 = work list, beside the step kernel (k_publish follows); 1 = work list, in line; 2 = in line, lane = game, own
// scan; 3 = beside the step kernel, lane = game, planned by the step kernel (sk_plan_deals)
int launch_deal_kernel(skyjo_vec *h, hipStream_t ds, int mode, int be_read) {
  hipEvent_t e0, e1;
  int rc;
  h->P.be_read = be_read >= 0 ? (uint32_t)be_read : (h->P.deal_tag & 1u);  // (two-stream form: the run's own parity, as the step kernel that planned it counted)
  if ((rc = prof_events(h, 2, &e0, &e1))) return rc;
  // fixed player counts deal from a byte deck per lane; the generic kernel needs the tile + ring
  const uint32_t lds_compact = SK_TILE * (SK_DECK_STRIDE + SK_STG_STRIDE),  // decks + MtChunkStream's staging rows
                  lds_generic = (uint32_t)(h->lds_tile + 16384);
  switch (h->P.L.N) {
    case 2: hipExtLaunchKernelGGL(k_deal<2>, dim3(h->P.tiles), dim3(SK_TILE), lds_compact, ds, e0, e1, 0, h->P, h->list_sel, mode); break;
    case 3: hipExtLaunchKernelGGL(k_deal<3>, dim3(h->P.tiles), dim3(SK_TILE), lds_compact, ds, e0, e1, 0, h->P, h->list_sel, mode); break;
    case 4: hipExtLaunchKernelGGL(k_deal<4>, dim3(h->P.tiles), dim3(SK_TILE), lds_compact, ds, e0, e1, 0, h->P, h->list_sel, mode); break;
    default: hipExtLaunchKernelGGL(k_deal<0>, dim3(h->P.tiles), dim3(SK_TILE), lds_generic, ds, e0, e1, 0, h->P, h->list_sel, mode); break;
  }
  HIPCHK(hipGetLastError());
  return SKYJO_OK;
}

static void adapt_interval(skyjo_vec *h) {
  if (h->auto_interval) {
    const volatile uint32_t *hh = h->health_host;
    const uint32_t tag = hh[1], empty = hh[0];
    if (tag != h->health_seen) {  // the report of a run that has finished since the last look (one or two runs old)
      h->health_seen = tag;
      // An empty bank at scan time is a warning (one more game end before the next run deals in place), not yet a
      // stall: a handful of them is tolerated, one game in a thousand is not.
      const uint32_t many = (uint32_t)(h->P.B / 1024 > 1 ? h->P.B / 1024 : 1);
      if (empty >= many) {
        const int cut = h->deal_every_iters / 8 > 4 ? h->deal_every_iters / 8 : 4;
        h->deal_every_iters = h->deal_every_iters - cut > 8 ? h->deal_every_iters - cut : 8;
        h->calm_runs = 0;
      } else if (empty) {
        h->calm_runs = 0;
      } else if (++h->calm_runs >= 8 && h->deal_every_iters < h->interval_default) {
        h->deal_every_iters += 2;
        h->calm_runs = 0;
      }
    }
  }
}
static void next_deal_tag(skyjo_vec *h, bool flip_list) {
  if (flip_list) h->list_sel ^= 1;  // (the work lists alternate: k_publish of one run clears the other's counter)
  h->deal_tag = (h->deal_tag + 1) & 0x7fffffffu;
  if (h->deal_tag == 0) h->deal_tag = 1;
  h->P.deal_tag = h->deal_tag;
}

// One dealing cycle: publish the previous one, list the banks that are not full, deal one episode for each.
// With overlap on, k_deal runs on its own stream beside the k_step launches that follow; its episodes are
// published at the start of the next cycle (one dealing interval later), long before a bank of SK_BANK runs dry.
int start_deals(skyjo_vec *h, hipStream_t s) {
  int rc;
  if (h->no_bank) {  // nothing is ever dealt ahead: a reset deals in place (deal_inline)
    h->pending_iters = 0;
    return SKYJO_OK;
  }
  if ((rc = publish_deals(h, s))) return rc;
  adapt_interval(h);
  next_deal_tag(h, true);
  // (the list's counter was cleared by the previous run's publish step)
  hipEvent_t e0, e1;
  // In line, the dealing kernel looks at the banks itself (lane = game): no k_scan launch, no work list.  Beside the step
  // kernel the list is made here, on the caller's stream, where it is ordered with the step launches around it.
  const bool fused = !h->overlap && h->fused_scan;
  if (!fused) {
    if ((rc = prof_events(h, 1, &e0, &e1))) return rc;
    hipExtLaunchKernelGGL(k_scan, dim3((h->P.B + SK_SCAN_BLOCK - 1) / SK_SCAN_BLOCK), dim3(SK_SCAN_BLOCK), 0, s, e0, e1, 0, h->P,
                          h->list_sel);
    HIPCHK(hipGetLastError());
  }
  hipStream_t ds = s;
  if (h->overlap) {
    HIPCHK(hipEventRecord(h->ev_scan, s));
    HIPCHK(hipStreamWaitEvent(h->deal_stream, h->ev_scan, 0));
    ds = h->deal_stream;
  }
  const int inl = h->overlap ? 0 : (fused ? 2 : 1);  // in line: k_deal publishes its own episodes, no k_publish launch
  if ((rc = launch_deal_kernel(h, ds, inl))) return rc;
  if (h->overlap) HIPCHK(hipEventRecord(h->ev_dealt, ds));
  h->deal_inflight = h->overlap;
  h->pending_iters = 0;
  return SKYJO_OK;
}

// The pipelined form of a dealing cycle beside the step kernel (skyjo_device.h, sk_plan_deals): plan_cycle() before the
// step launch after which the run is due - that launch plans the run on its way out -, start_deals_piped() after it:
// the dealing kernel goes to its own stream behind an event for that launch.  Nothing on the caller's stream waits.
static bool piped_mode(const skyjo_vec *h) { return ((h->overlap && h->piped) || h->merged) && !h->no_bank; }
static void plan_cycle(skyjo_vec *h) {
  next_deal_tag(h, false);
  h->P.plan_new_tag = h->deal_tag;
  h->P.be_add = h->deal_tag & 1u;  // (a k_cycle launch overrides this with its launch parity: launch_step)
  h->P.ov_flags |= 2u;
}
int start_deals_piped(skyjo_vec *h, hipStream_t s) {
  int rc;
  if (h->merged) {  // the next k_cycle launch carries the run (its wavefronts on SIMDs 1 / 3); nothing is launched here
    if ((rc = flush_cycle_deal(h, s))) return rc;  // (an older run that no launch has carried yet: deal it now)
    h->cycle_deal_tag = h->deal_tag;
    h->deal_inflight = true, h->inflight_piped = true;
    h->pending_iters = 0;
    adapt_interval(h);
    return SKYJO_OK;
  }
  HIPCHK(hipEventRecord(h->ev_scan, s));
  HIPCHK(hipStreamWaitEvent(h->deal_stream, h->ev_scan, 0));
  if ((rc = launch_deal_kernel(h, h->deal_stream, 3))) return rc;
  HIPCHK(hipEventRecord(h->ev_dealt, h->deal_stream));
  h->deal_inflight = true, h->inflight_piped = true;
  h->pending_iters = 0;
  adapt_interval(h);  // (for the cycles to come)
  return SKYJO_OK;
}

int launch_step(skyjo_vec *h, hipStream_t s, bool policy, const int32_t *actions, uint8_t *rec, int32_t *act_out,
                int iters, uint64_t policy_seed, double *end_rew = nullptr, uint8_t *end_flag = nullptr, uint8_t *raw_out = nullptr,
                int cycle_len = 0, bool planar_out = false) {
  h->raw_valid = false;  // (the host's copy of the games is stale from here on; step_host sets it again)
  dim3 grid(h->P.tiles), block(SK_TILE);
  const bool ind = h->P.L.indirect != 0;
  int rc;
  hipEvent_t e0, e1;
  if ((rc = prof_events(h, 0, &e0, &e1))) return rc;
  if (h->deal_inflight && h->inflight_piped) h->P.ov_flags |= 1u;  // publish what has been dealt since (sk_publish_deals)
  if (h->merged && policy && !end_rew && !raw_out) {
    // one kernel for the whole dealing cycle: S step + S dealing wavefronts per workgroup (= per CU)
    uint32_t lds_deal = SK_TILE * (SK_DECK_STRIDE + SK_STG_STRIDE), tag = h->cycle_deal_tag;
    h->cycle_deal_tag = 0;
    size_t lds_step_region = h->lds_cycle_step, lds_total = h->lds_cycle;
    bool no_defer = h->cycle_no_defer;
    if (h->rec_planar && !ind) {
      // the direct observation's wide records leave from registers in this layout: the staging area shrinks to the rare paths' 4 KiB of
      // scratch, and where the card chunks of the deferred scoring did not fit beside it (three players on a full chip) they do now
      const size_t base = h->lds_tile + 4096, with_defer = base + (size_t)h->P.L.N * 1024;
      no_defer = (size_t)h->cycle_s * (with_defer + lds_deal) + 32 > 160 * 1024;
      lds_step_region = no_defer ? base : with_defer;
      lds_total = (size_t)h->cycle_s * (lds_step_region + lds_deal) + 32;
    }
    if (no_defer) lds_deal |= 1u << 29;
    if (const char *e = sk_diag_env("SKYJO_CYCLE_SPLIT")) lds_deal |= (uint32_t)(atoi(e) & 3) << 30;  // diagnostic: 1 = roles by SIMD parity, 2 = by SIMD pair
    h->P.be_add = h->cycle_seq & 1u, h->P.be_read = (h->cycle_seq & 1u) ^ 1u;
    h->cycle_seq++;
    const int S = h->cycle_s;
    dim3 cgrid((h->P.tiles + S - 1) / S), cblock(2 * S * SK_TILE);
#define LAUNCHC(I, NP, PL)                                                                                                                 \
  hipExtLaunchKernelGGL((k_cycle<I, NP, PL>), cgrid, cblock, (uint32_t)lds_total, s, e0, e1, 0, h->P, rec, act_out, iters, policy_seed, \
                        h->iter, tag, (uint32_t)lds_step_region, lds_deal, cycle_len)
    switch (h->P.L.N * 2 + (ind ? 1 : 0) + (h->rec_planar ? 100 : 0)) {
      case 5: LAUNCHC(true, 2, false); break;
      case 7: LAUNCHC(true, 3, false); break;
      case 9: LAUNCHC(true, 4, false); break;
      case 105: LAUNCHC(true, 2, true); break;
      case 107: LAUNCHC(true, 3, true); break;
      case 109: LAUNCHC(true, 4, true); break;
      case 104: LAUNCHC(false, 2, true); break;
      case 106: LAUNCHC(false, 3, true); break;
      case 108: LAUNCHC(false, 4, true); break;
      case 4: LAUNCHC(false, 2, false); break;
      case 6: LAUNCHC(false, 3, false); break;
      default: LAUNCHC(false, 4, false); break;
    }
#undef LAUNCHC
    HIPCHK(hipGetLastError());
    h->P.ov_flags = 0;
    h->iters_total += (uint64_t)iters;
    h->iter += (uint64_t)iters;
    h->pending_iters += iters;
    return SKYJO_OK;
  }
  if ((rc = flush_cycle_deal(h, s))) return rc;  // (another kind of step: a planned run does not wait for a k_cycle launch)
#define LAUNCH3(I, Pol, NP)                                                                                       \
  hipExtLaunchKernelGGL((k_step<I, Pol, NP>), grid, block, (uint32_t)(Pol ? h->lds_rollout : h->lds_step), s, e0, e1, 0, h->P, actions,   \
                        rec, act_out, iters, policy_seed, h->iter, end_rew, end_flag, raw_out, h->raw_stride)
#define LAUNCH(I, Pol)                                \
  switch (h->P.L.N) {                                 \
    case 2: LAUNCH3(I, Pol, 2); break;                \
    case 3: LAUNCH3(I, Pol, 3); break;                \
    case 4: LAUNCH3(I, Pol, 4); break;                \
    default: LAUNCH3(I, Pol, 0); break;               \
  }
  if (ind && !policy && planar_out) {  // (SKYJO_REC_TILE_PLANAR_ALL: two to four players, indirect observation - checked when the option is set)
#define LAUNCHP(NP) hipExtLaunchKernelGGL((k_step<true, false, NP, true>), grid, block, (uint32_t)h->lds_step, s, e0, e1, 0, h->P, actions, rec, act_out, iters, \
                                          policy_seed, h->iter, end_rew, end_flag, raw_out, h->raw_stride)
    switch (h->P.L.N) {
      case 2: LAUNCHP(2); break;
      case 3: LAUNCHP(3); break;
      default: LAUNCHP(4); break;
    }
#undef LAUNCHP
  } else if (ind && policy) LAUNCH(true, true)
  else if (ind) LAUNCH(true, false)
  else if (policy) LAUNCH(false, true)
  else LAUNCH(false, false)
#undef LAUNCH3
#undef LAUNCH
  HIPCHK(hipGetLastError());
  h->P.ov_flags = 0;
  h->iters_total += (uint64_t)iters;
  if (policy) h->iter += (uint64_t)iters;  // the policy's Philox counter counts rollout iterations only
  h->pending_iters += iters;
  return SKYJO_OK;
}

int fetch_record(skyjo_vec *h, const uint4 *base, int game, std::vector<uint8_t> &raw, hipStream_t s) {
  const SkLayout &L = h->P.L;
  raw.assign((size_t)L.state_bytes, 0);
  const uint8_t *src = (const uint8_t *)(base + ((size_t)(game / SK_TILE) * L.chunks) * SK_TILE + game % SK_TILE);
  HIPCHK(hipMemcpy2DAsync(raw.data(), 16, src, SK_TILE * 16, 16, L.chunks, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return SKYJO_OK;
}

// One launch of the policy net (nets == 2: policy and value branch over the same records, grid.y = 2) in the net's precision
// (the kernels live in skyjo_policy.hip).  `planar`: the records lie tile-planar (SKYJO_REC_TILE_PLANAR).
// `prof`: the engine whose kernel timing (skyjo_vec_profile, slot 4) collects this launch, or null.
int launch_mlp(const skyjo_vec_mlp *ma, const skyjo_vec_mlp *mb, int nets, const uint8_t *rec, int rec_bytes, int obs_dim, int64_t n, float *out_a,
               const SkMlpDraw &draw, float *out_b, hipStream_t s, skyjo_vec *prof = nullptr, int planar = 0) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof) {
    int rc = prof_events(prof, 4, &e0, &e1);
    if (rc) return rc;
  }
  SkMlpRecords r;
  r.base = rec, r.rec_bytes = rec_bytes, r.obs_dim = obs_dim, r.planar = planar, r.n = (long long)n;
  HIPCHK((hipError_t)sk_launch_mlp(ma->net, mb->net, nets, r, out_a, draw, out_b, s, e0, e1));
  return SKYJO_OK;
}

// (Re)sizes the one-kernel form for this engine: called by skyjo_vec_create and by SKYJO_OPT_CYCLE_S.
static void configure_cycle(skyjo_vec *h, int s_override) {
  // k_cycle: S step + S dealing wavefronts per workgroup, S = what spreads the batch over the 256 CUs (1 .. 4); their LDS
    // regions and the claim words must fit one CU's 160 KB
    const int ncu = h->ncu;
    const SkParams &P = h->P;
    const bool fixed_n = P.L.N >= 2 && P.L.N <= 4;
    int S = (P.tiles + ncu - 1) / ncu;
    S = S < 1 ? 1 : (S > SK_CYCLE_MAX_S ? SK_CYCLE_MAX_S : S);
    if (s_override >= 1 && s_override <= SK_CYCLE_MAX_S) S = s_override;  // (SKYJO_OPT_CYCLE_S: fewer / more wavefronts per workgroup than the batch's share)
    const int natural_s = S;
    const size_t deal_region = (size_t)SK_TILE * (SK_DECK_STRIDE + SK_STG_STRIDE);
    size_t per_s = h->lds_rollout + deal_region;
    h->lds_cycle_step = h->lds_rollout, h->cycle_no_defer = false;
    if ((size_t)S * per_s + 32 > 160 * 1024 && (size_t)S * (h->lds_step + deal_region) + 32 <= 160 * 1024 && P.L.N <= 3 &&
        !sk_diag_env("SKYJO_CYCLE_DEFER_ONLY")) {
      // the direct observation of three players on a full chip: the CU's share of tiles fits without the card chunks of the deferred
      // scoring (the games are then scored in the iteration they end: a slower step, but one round of workgroups - 31.0 against 23.5
      // x 10^9 in line; four players, where scoring on the spot costs more: 22.5 against 30.1 - they stay in line)
      per_s = h->lds_step + deal_region;
      h->lds_cycle_step = h->lds_step, h->cycle_no_defer = true;
    }
    while (S > 1 && (size_t)S * per_s + 32 > 160 * 1024) S--;  // (fewer wavefronts per workgroup, more workgroups)
    const size_t need = (size_t)S * per_s + 32;
    const bool fits = fixed_n && need <= 160 * 1024 && !sk_diag_env("SKYJO_LDS_PAD");
    h->lds_cycle = need, h->cycle_s = S;
    h->merged_capable = fits;
    if (fits) {
      const void *fn = P.L.indirect ? (P.L.N == 2 ? (const void *)k_cycle<true, 2, false> : P.L.N == 3 ? (const void *)k_cycle<true, 3, false> : (const void *)k_cycle<true, 4, false>)
                                    : (P.L.N == 2 ? (const void *)k_cycle<false, 2, false> : P.L.N == 3 ? (const void *)k_cycle<false, 3, false> : (const void *)k_cycle<false, 4, false>);
      const void *fnp = P.L.indirect ? (P.L.N == 2 ? (const void *)k_cycle<true, 2, true> : P.L.N == 3 ? (const void *)k_cycle<true, 3, true> : (const void *)k_cycle<true, 4, true>)
                                     : (P.L.N == 2 ? (const void *)k_cycle<false, 2, true> : P.L.N == 3 ? (const void *)k_cycle<false, 3, true> : (const void *)k_cycle<false, 4, true>);
      // (the attribute belongs to the function, not to the handle: always the whole CU, so that engines of different batch sizes -
      // different S - can live side by side in one process)
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
          (fnp && hipFuncSetAttribute(fnp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)) {
        (void)hipGetLastError();
        h->merged_capable = false;
      }
    }
    // Default: the one-kernel form wherever it fits (measured at 4 096 .. 65 536 games and two to four players: + 9 .. 32 % over the
    // two-stream form, + 18 .. 57 % over dealing in line, EXPERIMENTS.md round 4); SKYJO_MERGED=0 falls back to the older forms.
    // (Batches beyond four tiles per CU run in two or more rounds of workgroups - a workgroup's LDS fills its CU: 98 304 x 3 29.1
    // against 23.8 in line, 131 072 x 3 34.0 against 21.0; counter-based deals 37.5 / 49.4 against 39.6 / 32.5: the one-kernel
    // form still wins everywhere but at one and a half rounds of counter-based deals.  Not so where the LDS - not the batch - makes
    // the workgroups smaller than a CU's share of the tiles, i.e. a full chip of four-player games or of the direct observation: S = 3,
    // 342 workgroups in one and a third rounds, 25.5 against 30.0 and 17.5 against 23.6 in line.)
    h->prefer_merged = S == natural_s;
    if (const char *e = sk_diag_env("SKYJO_MERGED")) h->prefer_merged = atoi(e) != 0;
    h->merged = h->merged_capable && h->prefer_merged && !sk_diag_env("SKYJO_OVERLAP");
    if (h->merged) h->overlap = false;  // (no second stream in this form)
}

}  // namespace

extern "C" {

const char *skyjo_vec_last_error(void) { return g_err.c_str(); }
int skyjo_vec_abi_version(void) { return SKYJO_ABI_VERSION; }

int skyjo_vec_create(const skyjo_vec_config *cfg, skyjo_vec **out) {
  if (!cfg || !out) return fail(SKYJO_E_INVALID, "null argument");
  *out = nullptr;
  if (cfg->abi_version != SKYJO_ABI_VERSION) return fail(SKYJO_E_INVALID, "abi_version mismatch");
  if (cfg->num_envs <= 0) return fail(SKYJO_E_INVALID, "num_envs must be > 0");
  if (cfg->num_players <= 0 || cfg->num_players > SKYJO_MAX_PLAYERS)  // skyjo.py:24-26
    return fail(SKYJO_E_INVALID, "Skyjo can be played from 1 up to 8 (recommended) / 12 (theoretical) players");
  if (cfg->rng_mode != SKYJO_RNG_MT19937 && cfg->rng_mode != SKYJO_RNG_PHILOX)
    return fail(SKYJO_E_INVALID, "unknown rng_mode");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(SKYJO_E_NOGPU, "no HIP device visible: this library has no CPU fallback");
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(SKYJO_E_INVALID, "device_id out of range");
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, cfg->device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !sk_diag_env("SKYJO_ALLOW_ANY_GPU"))
    return fail(SKYJO_E_NOGPU, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950");
  DevGuard guard_(cfg->device_id);  // (the caller's current device is restored on return)

  skyjo_vec *h = new (std::nothrow) skyjo_vec();
  if (!h) return fail(SKYJO_E_INVALID, "out of host memory");
  h->cfg = *cfg;
  h->generation = ++g_generation;
  SkParams &P = h->P;
  P.L = sk_make_layout(cfg->num_players, cfg->observe_indirect);
  P.B = cfg->num_envs;
  P.tiles = (cfg->num_envs + SK_TILE - 1) / SK_TILE;
  P.rng_mode = cfg->rng_mode;
  P.auto_reset = cfg->auto_reset ? 1 : 0;
  P.score_penalty = cfg->score_penalty, P.mean_reward = cfg->mean_reward;
  P.reward_refunded = cfg->reward_refunded, P.illegal_reward = cfg->illegal_reward;
  P.game_id0 = cfg->game_id0;
  P.spin_log2 = 22, P.debug_deal_delay = 0;
  h->G = (size_t)P.tiles * SK_TILE;
  h->lds_tile = (size_t)P.L.chunks * 1024;
  // tile + one iteration's records (64 B each, or rec_bytes + 16 for the direct observation; the rare paths' RNG scratch
  // aliases this area) + per-lane per-seat float64 statistics
  h->lds_bytes = h->lds_tile + (size_t)SK_TILE * (P.L.indirect ? 64 : P.L.rec_bytes + 16) + (size_t)SK_ACC_KINDS * cfg->num_players * 512 +
                 (cfg->num_players < 8 ? (size_t)cfg->num_players * 1024 : 0);  // + the card chunks of games waiting to be scored
  const bool fixed_n = cfg->num_players >= 2 && cfg->num_players <= 4;
  h->lds_step = fixed_n ? h->lds_tile + (size_t)SK_TILE * (P.L.indirect ? 64 : P.L.rec_bytes + 16) : h->lds_bytes;
  h->lds_rollout = fixed_n ? h->lds_step + (size_t)cfg->num_players * 1024 : h->lds_bytes;
  if (const char *e = sk_diag_env("SKYJO_LDS_PAD"))  // diagnostic: caps the wavefronts per CU
    h->lds_bytes += (size_t)atoi(e), h->lds_rollout += (size_t)atoi(e), h->lds_step += (size_t)atoi(e);
  const size_t rec16 = (size_t)P.tiles * P.L.chunks * SK_TILE;
  if ((uint64_t)SK_BANK * rec16 * 16 >= (1ull << 32)) {  // (LDS-DMA addresses the bank with 32-bit offsets)
    delete h;
    return fail(SKYJO_E_INVALID, "num_envs too large for one handle");
  }
  int rc = SKYJO_OK;
  const size_t N = (size_t)cfg->num_players;
  auto carve = [&]() -> int {
    int rc = SKYJO_OK;
    // (the largest array first: 164 MB at the headline size, walked in 64-byte pieces by the dealing kernel - it gets the
    // most contiguous backing a freshly started process can have)
    if (cfg->rng_mode == SKYJO_RNG_MT19937 && (rc = dalloc(h, &P.mt, h->G * 624 + 16, false)  /* + 16: MtChunkStream::issue reads one word beyond a state */)) {
      return rc;
    }
    if ((rc = dalloc(h, &P.state, rec16)) || (rc = dalloc(h, &P.spare, SK_BANK * rec16)) ||
        (rc = dalloc(h, &P.spare_ready, SK_BANK * h->G)) || (rc = dalloc(h, &P.bank_head, h->G)) ||
        (rc = dalloc(h, &P.busy, h->G)) || (rc = dalloc(h, &P.cancel, h->G)) || (rc = dalloc(h, &P.done_flag, h->G)) || (rc = dalloc(h, &P.plan_tag, h->G)) || (rc = dalloc(h, &P.plan_ep, h->G)) ||
        (rc = dalloc(h, &P.deal_list, 2 * h->G)) || (rc = dalloc(h, &P.deal_ep, 2 * h->G)) ||
        (rc = dalloc(h, &P.deal_count, 2)) || (rc = dalloc(h, &P.bank_empty, 2)) ||
        (rc = dalloc(h, &P.mt_idx, (1 + SK_BANK) * h->G)) || (rc = dalloc(h, &P.seeds, h->G)) ||
        (rc = dalloc(h, &P.deals_consumed, h->G)) || (rc = dalloc(h, &P.rewards, h->G * N)) ||
        (rc = dalloc(h, &P.scores, h->G * N)) || (rc = dalloc(h, &P.done, h->G)) ||
        (rc = dalloc(h, &P.acc_tile, (size_t)P.tiles * SK_ACC_KINDS * SKYJO_MAX_PLAYERS)) || (rc = dalloc(h, &P.dev_error, 1)) ||
        (rc = dalloc(h, &P.counters, 1)) || (rc = dalloc(h, &P.tile_counters, (size_t)P.tiles * 8)) || (rc = dalloc(h, &P.stamps, (size_t)P.tiles * 32))) {
      return rc;
    }
    return SKYJO_OK;
  };
  h->arena_mode = 1, h->arena_off = 0;
  (void)carve();
  {
    const size_t two_mb = (size_t)2 << 20, total = h->arena_off + two_mb;
    void *raw = nullptr;
    if (hipMalloc(&raw, total) != hipSuccess) {
      skyjo_vec_destroy(h);
      return fail(SKYJO_E_DEVICE, "hipMalloc failed for the engine's arrays");
    }
    h->owned.push_back(raw);
    h->arena = (uint8_t *)(((uintptr_t)raw + two_mb - 1) & ~(uintptr_t)(two_mb - 1));
  }
  h->arena_mode = 2, h->arena_off = 0;
  rc = carve();
  h->arena_mode = 0;
  if (rc) {
    skyjo_vec_destroy(h);
    return rc;
  }
  int prio_least = 0, prio_greatest = 0;  // the dealing kernel fills what the step kernel leaves idle: lowest priority
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  if (hipStreamCreateWithPriority(&h->deal_stream, hipStreamNonBlocking, prio_least) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_scan, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_dealt, hipEventDisableTiming) != hipSuccess) {
    skyjo_vec_destroy(h);
    return fail(SKYJO_E_DEVICE, "cannot create the dealing stream / events");
  }
  {
    void *dp = nullptr;
    if (hipHostMalloc((void **)&h->health_host, 4 * sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(&dp, h->health_host, 0) != hipSuccess) {
      skyjo_vec_destroy(h);
      return fail(SKYJO_E_DEVICE, "cannot map the bank-health word");
    }
    h->health_host[0] = h->health_host[1] = h->health_host[2] = h->health_host[3] = 0;
    P.health_host = (volatile uint32_t *)dp;
  }
  h->raw_stride = (int)((P.L.state_bytes + 16 * cfg->num_players + 8 + 15) & ~15);
  if (h->G <= 4096) {  // small batches: the *_host calls go through host-mapped memory (see the handle's fields)
    const size_t a = (h->G * 4 + 255) & ~(size_t)255, m = (h->G + 255) & ~(size_t)255, r = (h->G * (size_t)P.L.rec_bytes + 255) & ~(size_t)255,
                 w = h->G * (size_t)h->raw_stride;
    void *dp = nullptr;
    if (hipHostMalloc((void **)&h->hm_block, a + m + r + w, hipHostMallocMapped) == hipSuccess &&
        hipHostGetDevicePointer(&dp, h->hm_block, 0) == hipSuccess) {
      memset(h->hm_block, 0, a + m + r + w);
      uint8_t *d = (uint8_t *)dp;
      h->hm_actions = (int32_t *)h->hm_block, h->hm_mask = h->hm_block + a, h->hm_records = h->hm_block + a + m, h->hm_raw = h->hm_block + a + m + r;
      h->hm_actions_d = (int32_t *)d, h->hm_mask_d = d + a, h->hm_records_d = d + a + m, h->hm_raw_d = d + a + m + r;
      h->fast_host = !sk_diag_env("SKYJO_NO_FAST_HOST");
      h->host_spin = !sk_diag_env("SKYJO_NO_SPIN");
      // Whole games come back with the records only for a few tiles (the single-game views: get_state / rewards after a step
      // without device traffic).  At raw_stride ~ 400 B per game the export is per-lane 16-byte stores at that stride over
      // PCIe - for a mid-size batch whose caller may never ask for a state that is dead weight.
      h->raw_export = h->fast_host && P.tiles <= 4 && !sk_diag_env("SKYJO_NO_RAW_EXPORT");
    }
  }
  // The dealing kernel runs beside the step kernel (own stream) when the batch leaves SIMDs free: up to 768 tiles of
  // the 1024 one-wavefront-per-SIMD slots (three-player games, beside / in line, x 10^9 steps/s: 32 768: 17.1 / 15.5,
  // 49 152: 23.1 / 22.6, 57 344: 25.2 / 26.1, 65 536: 20.3 / 29.0 - on a full chip the two kernels compete for the same
  // vector ALUs), in line above that.  SKYJO_OPT_OVERLAP / SKYJO_OVERLAP override.
  h->overlap = P.tiles <= 768;
  if (const char *e = sk_diag_env("SKYJO_OVERLAP")) h->overlap = atoi(e) != 0;
  h->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  configure_cycle(h, 0);
  if (const char *e = sk_diag_env("SKYJO_FUSED_SCAN")) h->fused_scan = atoi(e) != 0;
  if (const char *e = sk_diag_env("SKYJO_PIPELINED")) h->piped = atoi(e) != 0;
  h->deal_every_iters = h->interval_default = deal_interval_default(cfg->num_players, h->overlap, h->piped, h->merged ? h->cycle_s : 0);
  if (const char *e = sk_diag_env("SKYJO_DEAL_INTERVAL")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 1024) h->deal_every_iters = v, h->auto_interval = false;
  }
  *out = h;
  return SKYJO_OK;
}

int skyjo_vec_destroy(skyjo_vec *h) {
  if (!h) return SKYJO_OK;
  GUARD(h);
  (void)hipDeviceSynchronize();
  if (h->deal_stream) (void)hipStreamDestroy(h->deal_stream);
  if (h->ev_scan) (void)hipEventDestroy(h->ev_scan);
  if (h->ev_dealt) (void)hipEventDestroy(h->ev_dealt);
  for (void *p : h->owned) (void)hipFree(p);
  for (auto &v : h->ev)
    for (auto &e : v) (void)hipEventDestroy(e.first), (void)hipEventDestroy(e.second);
  if (h->health_host) (void)hipHostFree(h->health_host);
  if (h->hm_block) (void)hipHostFree(h->hm_block);
  delete h;
  return SKYJO_OK;
}

int skyjo_vec_get_info(const skyjo_vec *h, skyjo_vec_info *out) {
  if (!h || !out) return fail(SKYJO_E_INVALID, "null argument");
  const SkLayout &L = h->P.L;
  out->num_envs = h->P.B, out->num_players = L.N, out->obs_dim = L.D, out->record_bytes = L.rec_bytes;
  out->mask_offset = L.Dp, out->meta_offset = L.Dp + 26, out->state_bytes = L.state_bytes, out->tile_games = SK_TILE;
  return SKYJO_OK;
}

int skyjo_vec_seed(skyjo_vec *h, const uint64_t *seeds_host, uint64_t base_seed, void *stream) {
  if (!h) return fail(SKYJO_E_INVALID, "null handle");
  hipStream_t s = (hipStream_t)stream;
  GUARD(h);
  h->raw_valid = false;
  uint64_t *d_seeds = nullptr;
  if (seeds_host) {
    HIPCHK(hipMalloc((void **)&d_seeds, sizeof(uint64_t) * (size_t)h->P.B));
    HIPCHK(hipMemcpyAsync(d_seeds, seeds_host, sizeof(uint64_t) * (size_t)h->P.B, hipMemcpyHostToDevice, s));
  }
  HIPCHK(hipDeviceSynchronize());  // (nothing of the old life of the handle is still running)
  HIPCHK(hipMemsetAsync(h->P.done, 0, h->G, s));
  HIPCHK(hipMemsetAsync(h->P.dev_error, 0, sizeof(uint32_t), s));  // a new seeding starts from a clean slate
  h->health_host[2] = 0;
  int rc0;
  if ((rc0 = publish_deals(h, s))) return rc0;  // drain the dealing pipeline of the previous seeding, if any
  hipLaunchKernelGGL(k_seed, dim3((h->P.B + 255) / 256), dim3(256), 0, s, h->P, (const uint64_t *)d_seeds, base_seed, 0,
                     h->P.B);
  HIPCHK(hipGetLastError());
  int rc;
  // set_seed deals immediately (skyjo.py:88): deal #0 becomes the live game, the following ones fill the bank
  h->deal_inflight = false, h->inflight_piped = false;
  if ((rc = start_deals(h, s)) || (rc = publish_deals(h, s))) return rc;
  h->seeded = true;
  if ((rc = skyjo_vec_reset(h, nullptr, nullptr, stream))) return rc;
  for (int k = 1; k < SK_BANK; k++)
    if ((rc = start_deals(h, s)) || (rc = publish_deals(h, s))) return rc;
  if (d_seeds) {
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(d_seeds));
  }
  h->iter = 0;
  return skyjo_vec_reset_counters(h, stream);
}

static int ensure_scratch(skyjo_vec *h);

int skyjo_vec_seed_one(skyjo_vec *h, int32_t game, uint64_t value, void *stream) {
  if (!h || game < 0 || game >= h->P.B) return fail(SKYJO_E_INVALID, "bad argument");
  if (!h->seeded) return fail(SKYJO_E_STATE, "skyjo_vec_seed must be called first");
  GUARD(h);
  h->raw_valid = false;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if ((rc = ensure_scratch(h)) || (rc = publish_deals(h, s))) return rc;  // nobody else may be using the stream that is re-seeded
  uint64_t *d_seed = (uint64_t *)h->d_actions;                               // (8 bytes of the host-style scratch)
  HIPCHK(hipMemcpyAsync(d_seed, &value, sizeof(value), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));  // (`value` lives on this stack frame)
  hipLaunchKernelGGL(k_seed, dim3(1), dim3(64), 0, s, h->P, (const uint64_t *)d_seed, (uint64_t)0, (int)game, 1);
  HIPCHK(hipGetLastError());
  if ((rc = start_deals(h, s)) || (rc = publish_deals(h, s))) return rc;  // deal #0 of the new stream ...
  HIPCHK(hipMemsetAsync(h->d_mask, 0, h->G, s));
  const uint8_t one = 1;
  HIPCHK(hipMemcpyAsync(h->d_mask + game, &one, 1, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  if ((rc = skyjo_vec_reset(h, h->d_mask, nullptr, stream))) return rc;    // ... becomes the live game
  for (int k = 1; k < SK_BANK; k++)
    if ((rc = start_deals(h, s)) || (rc = publish_deals(h, s))) return rc;
  return SKYJO_OK;
}

int skyjo_vec_snapshot_create(skyjo_vec *h, skyjo_vec_snapshot **out, void *stream) {
  if (!h || !out) return fail(SKYJO_E_INVALID, "null argument");
  if (!h->seeded) return fail(SKYJO_E_STATE, "skyjo_vec_seed must be called first");
  GUARD(h);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if ((rc = publish_deals(h, s))) return rc;  // no deal in flight: the arrays below are the whole truth
  HIPCHK(hipDeviceSynchronize());
  if ((rc = dev_error_fetch(h, s))) return rc;   // (a voided run is not worth keeping)
  skyjo_vec_snapshot *sn = new (std::nothrow) skyjo_vec_snapshot();
  if (!sn) return fail(SKYJO_E_INVALID, "out of host memory");
  sn->owner = h, sn->owner_generation = h->generation, sn->device_id = h->cfg.device_id;
  for (auto &a : h->allocs) sn->bytes += (a.second + 255) & ~(size_t)255, sn->array_bytes.push_back(a.second);
  if (hipMalloc(&sn->blob, sn->bytes) != hipSuccess) {
    delete sn;
    return fail(SKYJO_E_DEVICE, "hipMalloc failed for the snapshot");
  }
  size_t off = 0;
  for (auto &a : h->allocs) {
    hipError_t e = hipMemcpyAsync((uint8_t *)sn->blob + off, a.first, a.second, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) {
      (void)hipFree(sn->blob);
      delete sn;
      return fail(SKYJO_E_DEVICE, std::string("snapshot copy: ") + hipGetErrorString(e));
    }
    off += (a.second + 255) & ~(size_t)255;
  }
  if (hipError_t e = hipStreamSynchronize(s); e != hipSuccess) {
    (void)hipFree(sn->blob);
    delete sn;
    return fail(SKYJO_E_DEVICE, std::string("snapshot copy: ") + hipGetErrorString(e));
  }
  sn->pending_iters = h->pending_iters, sn->deal_every_iters = h->deal_every_iters, sn->calm_runs = h->calm_runs;
  sn->list_sel = h->list_sel, sn->auto_interval = h->auto_interval, sn->health_seen = h->health_seen;
  sn->deal_tag = h->deal_tag, sn->iter = h->iter, sn->iters_total = h->iters_total;
  sn->health[0] = h->health_host[0], sn->health[1] = h->health_host[1];
  *out = sn;
  return SKYJO_OK;
}

int skyjo_vec_snapshot_restore(skyjo_vec *h, const skyjo_vec_snapshot *sn, void *stream) {
  if (!h || !sn) return fail(SKYJO_E_INVALID, "null argument");
  if (sn->owner != h || sn->owner_generation != h->generation)  // (a destroyed handle's address may have been reused)
    return fail(SKYJO_E_INVALID, "the snapshot was taken from another handle");
  // the blob is the handle's arrays in table order: the same count (or fewer: the host-style scratch may have been
  // allocated since) of the same sizes
  if (sn->array_bytes.size() > h->allocs.size()) return fail(SKYJO_E_INVALID, "the snapshot does not fit the handle's arrays");
  for (size_t k = 0; k < sn->array_bytes.size(); k++)
    if (sn->array_bytes[k] != h->allocs[k].second) return fail(SKYJO_E_INVALID, "the snapshot does not fit the handle's arrays");
  GUARD(h);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if ((rc = publish_deals(h, s))) return rc;
  HIPCHK(hipDeviceSynchronize());
  size_t off = 0;
  for (size_t k = 0; k < sn->array_bytes.size(); k++) {  // (arrays allocated after the snapshot are not part of it)
    const auto &a = h->allocs[k];
    HIPCHK(hipMemcpyAsync(a.first, (const uint8_t *)sn->blob + off, a.second, hipMemcpyDeviceToDevice, s));
    off += (a.second + 255) & ~(size_t)255;
  }
  HIPCHK(hipStreamSynchronize(s));
  h->pending_iters = sn->pending_iters, h->deal_every_iters = sn->deal_every_iters, h->calm_runs = sn->calm_runs;
  h->list_sel = sn->list_sel, h->auto_interval = sn->auto_interval, h->health_seen = sn->health_seen;
  h->deal_tag = sn->deal_tag, h->P.deal_tag = sn->deal_tag, h->iter = sn->iter, h->iters_total = sn->iters_total;
  h->health_host[0] = sn->health[0], h->health_host[1] = sn->health[1];
  h->health_host[2] = 0;  // the restored device word is clean (snapshot_create refuses a voided run): so is the host's copy
  h->deal_inflight = false, h->inflight_piped = false;
  h->raw_valid = false;
  return SKYJO_OK;
}

int skyjo_vec_snapshot_bytes(const skyjo_vec_snapshot *sn, size_t *bytes_out) {
  if (!sn || !bytes_out) return fail(SKYJO_E_INVALID, "null argument");
  *bytes_out = sn->bytes;
  return SKYJO_OK;
}

int skyjo_vec_snapshot_destroy(skyjo_vec_snapshot *sn) {
  if (!sn) return SKYJO_OK;
  DevGuard guard_(sn->device_id);
  (void)hipFree(sn->blob);
  delete sn;
  return SKYJO_OK;
}

static int reset_impl(skyjo_vec *h, const uint8_t *mask, void *records_out, hipStream_t s, uint8_t *raw_out, int planar = 0) {
  int rc;
  h->raw_valid = false;
  if ((rc = publish_deals(h, s))) return rc;  // make every dealt episode available
  dim3 grid(h->P.tiles), block(SK_TILE);
  if (h->P.L.indirect)
    hipLaunchKernelGGL((k_reset<true>), grid, block, h->lds_bytes, s, h->P, mask, (uint8_t *)records_out, raw_out, h->raw_stride, planar);
  else
    hipLaunchKernelGGL((k_reset<false>), grid, block, h->lds_bytes, s, h->P, mask, (uint8_t *)records_out, raw_out, h->raw_stride, 0);
  HIPCHK(hipGetLastError());
  if ((rc = start_deals(h, s))) return rc;  // refill what was taken (one episode per game and cycle)
  return publish_deals(h, s);
}

int skyjo_vec_reset(skyjo_vec *h, const uint8_t *mask, void *records_out, void *stream) {
  if (!h) return fail(SKYJO_E_INVALID, "null handle");
  GUARD(h);
  if (!h->seeded) return fail(SKYJO_E_STATE, "skyjo_vec_seed must be called first");
  return reset_impl(h, mask, records_out, (hipStream_t)stream, nullptr, h->rec_planar_all ? 1 : 0);
}

static int step_once(skyjo_vec *h, const int32_t *actions, void *records_out, double *end_rew, uint8_t *end_flag, hipStream_t s,
                     uint8_t *raw_out = nullptr, bool planar_out = false) {
  const bool due = h->pending_iters + 1 >= h->deal_every_iters, piped = piped_mode(h);
  if (due && piped) plan_cycle(h);
  int rc = launch_step(h, s, false, actions, (uint8_t *)records_out, nullptr, 1, 0, end_rew, end_flag, raw_out, 0, planar_out && records_out);
  if (rc) return rc;
  if (due) return piped ? start_deals_piped(h, s) : start_deals(h, s);
  return SKYJO_OK;
}

int skyjo_vec_step(skyjo_vec *h, const int32_t *actions, void *records_out, void *stream) {
  if (!h || !actions) return fail(SKYJO_E_INVALID, "null argument");
  GUARD(h);
  if (!h->seeded) return fail(SKYJO_E_STATE, "skyjo_vec_seed must be called first");
  return step_once(h, actions, records_out, nullptr, nullptr, (hipStream_t)stream, nullptr, h->rec_planar_all);
}

int skyjo_vec_step_collect(skyjo_vec *h, const int32_t *actions, void *records_out, double *final_rewards_out,
                           uint8_t *episode_end_out, void *stream) {
  if (!h || !actions || !final_rewards_out || !episode_end_out) return fail(SKYJO_E_INVALID, "null argument");
  GUARD(h);
  if (!h->seeded) return fail(SKYJO_E_STATE, "skyjo_vec_seed must be called first");
  return step_once(h, actions, records_out, final_rewards_out, episode_end_out, (hipStream_t)stream, nullptr, h->rec_planar_all);
}

// ---- config 5's collection loop (include/skyjo_vec.h: skyjo_vec_model_rollout) ----
static int model_check(skyjo_vec *h, const skyjo_vec_mlp *policy, const skyjo_vec_mlp *value, int32_t T, const skyjo_vec_rollout_buffers *b) {
  if (!h || !policy || !b || T < 0 || !b->records || !b->actions) return fail(SKYJO_E_INVALID, "skyjo_vec_model_rollout: bad argument");
  if (value && !b->values) return fail(SKYJO_E_INVALID, "skyjo_vec_model_rollout: a value net needs a values buffer");
  if ((b->final_rewards == nullptr) != (b->episode_end == nullptr))
    return fail(SKYJO_E_INVALID, "skyjo_vec_model_rollout: final_rewards and episode_end go together");
  if (!h->seeded) return fail(SKYJO_E_STATE, "skyjo_vec_seed must be called first");
  if (policy->net.out_dim != SKYJO_NUM_ACTIONS) return fail(SKYJO_E_INVALID, "the policy net needs 26 outputs");
  if (policy->device_id != h->cfg.device_id || policy->obs_dim != h->P.L.D || !h->P.L.indirect)
    return fail(SKYJO_E_INVALID, "the policy net must live on the engine's device and take the engine's (indirect) observation");
  if (value && (policy->obs_dim != value->obs_dim || policy->device_id != value->device_id || policy->net.split != value->net.split))
    return fail(SKYJO_E_INVALID, "policy and value net must share the observation size, the precision and the device");
  return SKYJO_OK;
}
// iteration t: [policy (+ value) net with the draw in its epilogue] -> [step kernel with the episode-end columns]
static int model_iter(skyjo_vec *h, const skyjo_vec_mlp *policy, const skyjo_vec_mlp *value, int t, uint64_t seed, uint64_t first_ticket,
                      int32_t no_masking, const skyjo_vec_rollout_buffers *b, hipStream_t s) {
  const size_t B = (size_t)h->P.B, rb = (size_t)h->P.L.rec_bytes, N = (size_t)h->P.L.N, vd = value ? (size_t)value->net.out_dim : 0;
  const bool planar = h->rec_planar_all;        // (SKYJO_REC_TILE_PLANAR_ALL: an iteration's records are tiles * 64 slots, piece-planar)
  const size_t per_it = (planar ? h->G : B) * rb;
  uint8_t *rec = (uint8_t *)b->records;
  SkMlpDraw d{};
  d.enable = 1, d.mask_offset = h->P.L.Dp, d.no_masking = no_masking, d.seed = seed, d.ticket = first_ticket + (uint64_t)t;
  d.game_id0 = h->P.game_id0, d.actions = b->actions + (size_t)t * B, d.logp = b->logp ? b->logp + (size_t)t * B : nullptr;
  int rc = launch_mlp(policy, value ? value : policy, value ? 2 : 1, rec + (size_t)t * per_it, (int)rb, policy->obs_dim, (int64_t)B, nullptr, d,
                      value ? b->values + (size_t)t * B * vd : nullptr, s, h, planar ? 1 : 0);
  if (rc) return rc;
  return step_once(h, b->actions + (size_t)t * B, rec + (size_t)(t + 1) * per_it, b->final_rewards ? b->final_rewards + (size_t)t * B * N : nullptr,
                   b->episode_end ? b->episode_end + (size_t)t * B : nullptr, s, nullptr, planar);
}
// the bootstrap value of the records a rollout ends on
static int model_tail(skyjo_vec *h, const skyjo_vec_mlp *value, int T, const skyjo_vec_rollout_buffers *b, hipStream_t s) {
  if (!value) return SKYJO_OK;
  const size_t B = (size_t)h->P.B, rb = (size_t)h->P.L.rec_bytes, vd = (size_t)value->net.out_dim;
  SkMlpDraw nodraw{};
  return launch_mlp(value, value, 1, (const uint8_t *)b->records + (size_t)T * (h->rec_planar_all ? h->G : B) * rb, (int)rb, value->obs_dim, (int64_t)B,
                    b->values + (size_t)T * B * vd, nodraw, nullptr, s, nullptr, h->rec_planar_all ? 1 : 0);
}

int skyjo_vec_model_rollout(skyjo_vec *h, const skyjo_vec_mlp *policy, const skyjo_vec_mlp *value, int32_t T, uint64_t seed,
                            uint64_t first_ticket, int32_t no_masking, const skyjo_vec_rollout_buffers *b, void *stream) {
  int rc = model_check(h, policy, value, T, b);
  if (rc) return rc;
  GUARD(h);
  for (int t = 0; t < T; t++)
    if ((rc = model_iter(h, policy, value, t, seed, first_ticket, no_masking, b, (hipStream_t)stream))) return rc;
  return model_tail(h, value, T, b, (hipStream_t)stream);
}

int skyjo_vec_rollout(skyjo_vec *h, int32_t iters, uint64_t policy_seed, void *records_out, int32_t *actions_out,
                      void *stream) {
  if (!h || iters < 0) return fail(SKYJO_E_INVALID, "bad argument");
  GUARD(h);
  if (!h->seeded) return fail(SKYJO_E_STATE, "skyjo_vec_seed must be called first");
  hipStream_t s = (hipStream_t)stream;
  uint8_t *rec = (uint8_t *)records_out;
  const bool planar = h->rec_planar;
  if (planar && rec && !h->merged)
    return fail(SKYJO_E_STATE, "SKYJO_OPT_RECORD_LAYOUT = tile-planar needs the one-kernel dealing form (SKYJO_OPT_OVERLAP 3)");
  for (int done = 0; done < iters;) {
    int n = iters - done < kMaxRolloutChunk ? iters - done : kMaxRolloutChunk;
    // a launch ends where the next dealing run is due, so the cadence does not depend on how the caller slices its calls
    const int due = h->deal_every_iters - h->pending_iters;
    if (n > due) n = due > 0 ? due : 1;
    const bool run_due = h->pending_iters + n >= h->deal_every_iters, piped = piped_mode(h);
    // The one-kernel form keeps its tiles in LDS over several dealing cycles when the caller asks for that many iterations at once:
    // up to kMaxCyclesPerLaunch whole cycles in ONE launch (the cycle ends inside it are handled by the kernel: step_body, k_cycle)
    int cycles = 1;
    // (only where the kernel's cycle ends have a run to hand over: an engine that never deals ahead - SKYJO_OPT_NO_BANK - plans nothing,
    // and a cycle end inside its launch would publish and plan with a stale id: ADVICE r4)
    if (h->merged && piped && run_due && h->pending_iters == 0) {
      const int cap = h->max_cycles > 0 ? h->max_cycles : kMaxCyclesPerLaunch;
      cycles = (iters - done) / h->deal_every_iters;
      cycles = cycles < 1 ? 1 : (cycles > cap ? cap : cycles);
      n = cycles * h->deal_every_iters;
    }
    if (run_due && piped) {
      plan_cycle(h);
      const uint32_t first = h->deal_tag;  // the id of the first run planned by this launch; the kernel counts on from it
      for (int c = 1; c < cycles; c++) next_deal_tag(h, false);
      h->P.plan_new_tag = first;
    }
    int rc = launch_step(h, s, true, nullptr, rec, actions_out, n, policy_seed, nullptr, nullptr, nullptr, cycles > 1 ? h->deal_every_iters : 0);
    if (rc) return rc;
    done += n;
    if (run_due && (rc = piped ? start_deals_piped(h, s) : start_deals(h, s))) return rc;
    if (rec) rec += (size_t)n * (planar ? h->G : (size_t)h->P.B) * h->P.L.rec_bytes;
    if (actions_out) actions_out += (size_t)n * h->P.B;
  }
  return SKYJO_OK;
}

static int observe_impl(skyjo_vec *h, const int32_t *players, void *records_out, void *stream, int planar) {
  if (!h || !records_out) return fail(SKYJO_E_INVALID, "null argument");
  GUARD(h);
  if (!h->seeded) return fail(SKYJO_E_STATE, "skyjo_vec_seed must be called first");
  dim3 grid(h->P.tiles), block(SK_TILE);
  hipStream_t s = (hipStream_t)stream;
  if (h->P.L.indirect)
    hipLaunchKernelGGL((k_observe<true>), grid, block, h->lds_bytes, s, h->P, players, (uint8_t *)records_out, planar);
  else
    hipLaunchKernelGGL((k_observe<false>), grid, block, h->lds_bytes, s, h->P, players, (uint8_t *)records_out, 0);
  HIPCHK(hipGetLastError());
  return SKYJO_OK;
}
int skyjo_vec_observe(skyjo_vec *h, const int32_t *players, void *records_out, void *stream) {
  return observe_impl(h, players, records_out, stream, h && h->rec_planar_all ? 1 : 0);
}

static int unpack_impl(skyjo_vec *h, const void *records, int64_t n, int8_t *obs, int8_t *mask, uint8_t *agent, uint8_t *phase, uint8_t *done,
                       uint8_t *status, void *stream, int planar) {
  if (!h || !records || n < 0) return fail(SKYJO_E_INVALID, "bad argument");
  GUARD(h);
  if (n == 0) return SKYJO_OK;
  long long total = n * (long long)(h->P.L.D + 26);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_unpack, dim3(blocks), dim3(256), 0, (hipStream_t)stream, h->P.L, (const uint8_t *)records,
                     (long long)n, obs, mask, agent, phase, done, status, planar);
  HIPCHK(hipGetLastError());
  return SKYJO_OK;
}

int skyjo_vec_unpack(skyjo_vec *h, const void *records, int64_t n, int8_t *obs, int8_t *mask, uint8_t *agent,
                     uint8_t *phase, uint8_t *done, uint8_t *status, void *stream) {
  return unpack_impl(h, records, n, obs, mask, agent, phase, done, status, stream, 0);
}

int skyjo_vec_unpack_tiles(skyjo_vec *h, const void *records, int64_t n_tiles, int8_t *obs, int8_t *mask, uint8_t *agent,
                           uint8_t *phase, uint8_t *done, uint8_t *status, void *stream) {
  return unpack_impl(h, records, n_tiles * SK_TILE, obs, mask, agent, phase, done, status, stream, 1);
}

const double *skyjo_vec_rewards_ptr(const skyjo_vec *h) { return h ? h->P.rewards : nullptr; }
const double *skyjo_vec_scores_ptr(const skyjo_vec *h) { return h ? h->P.scores : nullptr; }
const uint8_t *skyjo_vec_done_ptr(const skyjo_vec *h) { return h ? h->P.done : nullptr; }

int skyjo_vec_sample_actions_layout(skyjo_vec *h, const void *records, int32_t layout, const float *logits, int64_t n, uint64_t seed,
                                    uint64_t ticket, int32_t no_masking, int32_t *actions_out, float *logp_out,
                                    float *uniform_out, void *stream) {
  if (!h || !records || !logits || !actions_out || n < 0) return fail(SKYJO_E_INVALID, "null argument");
  if (layout != SKYJO_REC_ROW_MAJOR && layout != SKYJO_REC_TILE_PLANAR) return fail(SKYJO_E_INVALID, "layout must be SKYJO_REC_ROW_MAJOR or SKYJO_REC_TILE_PLANAR");
  GUARD(h);
  if (n == 0) return SKYJO_OK;
  const int64_t blocks = (n + SK_SAMPLE_BLOCK - 1) / SK_SAMPLE_BLOCK;
  hipLaunchKernelGGL(k_sample, dim3((unsigned)blocks), dim3(SK_SAMPLE_BLOCK), 0, (hipStream_t)stream, h->P.L,
                     (const uint8_t *)records, logits, (long long)n, seed, ticket, h->P.game_id0, (int)no_masking,
                     actions_out, logp_out, uniform_out, (int)(layout == SKYJO_REC_TILE_PLANAR));
  HIPCHK(hipGetLastError());
  return SKYJO_OK;
}
int skyjo_vec_sample_actions(skyjo_vec *h, const void *records, const float *logits, int64_t n, uint64_t seed,
                             uint64_t ticket, int32_t no_masking, int32_t *actions_out, float *logp_out,
                             float *uniform_out, void *stream) {
  return skyjo_vec_sample_actions_layout(h, records, SKYJO_REC_ROW_MAJOR, logits, n, seed, ticket, no_masking, actions_out, logp_out, uniform_out,
                                         stream);
}

int skyjo_vec_mlp_create(int32_t device_id, int32_t obs_dim, int32_t out_dim, int32_t precision, const float *w1, const float *b1,
                         const float *w2, const float *b2, const float *w3, const float *b3, skyjo_vec_mlp **out) {
  if (!out || !w1 || !b1 || !w2 || !b2 || !w3 || !b3) return fail(SKYJO_E_INVALID, "null argument");
  if (obs_dim < 1 || obs_dim > SKP_IN - 1 || out_dim < 1 || out_dim > SKP_OUT)
    return fail(SKYJO_E_INVALID, "skyjo_vec_mlp: obs_dim must be 1..31 and out_dim 1..32");
  if (precision != SKYJO_MLP_BF16 && precision != SKYJO_MLP_FP32) return fail(SKYJO_E_INVALID, "skyjo_vec_mlp: unknown precision");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return fail(SKYJO_E_INVALID, "device_id out of range");
  DevGuard guard_(device_id);
  auto bf16 = [](float f) {  // round to nearest even
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  };
  auto bf16_to_float = [](uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
  };
  auto bf16_lo = [&](float f) {  // what the high half leaves over, again rounded to bf16: f = hi + lo to 16 significant bits
    const uint32_t hi = (uint32_t)bf16(f) << 16;
    float fh;
    memcpy(&fh, &hi, 4);
    return bf16(f - fh);
  };
  const bool split = precision == SKYJO_MLP_FP32;
  // The two hidden layers are stored times 2 / ln 2 (weights AND biases, before the rounding to bf16 / the split into two bf16): their
  // accumulators are then the exponent of tanh(x) = 1 - 2 / (2^(x 2 / ln 2) + 1) as they stand - no multiply per activation (skyjo_policy.hip)
  auto acc_k = [](int ks, int hh, int j) { return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * hh + (j & 3); };
  const int H = SKP_HIDDEN;
  const size_t e1 = (size_t)8 * 2 * 64 * 8, e2 = (size_t)8 * 16 * 64 * 8, e3 = (size_t)16 * 64 * 8;
  std::vector<uint16_t> f1(e1), f2(e2), f3(e3), g1(split ? e1 : 0), g2(split ? e2 : 0), g3(split ? e3 : 0);
  std::vector<float> c2((size_t)H), c3((size_t)64 * 16);
  for (int u = 0; u < 8; u++)
    for (int l = 0; l < 64; l++) {
      const int m = 32 * u + (l & 31), hh = l >> 5;
      for (int s = 0; s < 2; s++)
        for (int j = 0; j < 8; j++) {
          const int k = 16 * s + 8 * hh + j;  // natural order: the kernel builds this operand from the record itself
          const float v = SKP_SCALE * (k < obs_dim ? w1[(size_t)m * obs_dim + k] : (k == SKP_IN - 1 ? b1[m] : 0.f));
          const size_t at = (((size_t)u * 2 + s) * 64 + l) * 8 + j;
          f1[at] = bf16(v);
          if (split) g1[at] = bf16_lo(v);
        }
      for (int ks = 0; ks < 16; ks++)
        for (int j = 0; j < 8; j++) {
          const float v = SKP_SCALE * w2[(size_t)m * H + acc_k(ks, hh, j)];
          const size_t at = (((size_t)u * 16 + ks) * 64 + l) * 8 + j;
          // bf16 mode: the layer takes r = (1 - tanh) / 2 of the layer before: - 2 W as weights (exact: a power of two), W 1 joins the bias
          f2[at] = split ? bf16(v) : bf16(-2.0f * v);
          if (split) g2[at] = bf16_lo(v);
        }
    }
  for (int l = 0; l < 64; l++) {
    const int m = l & 31, hh = l >> 5;
    for (int ks = 0; ks < 16; ks++)
      for (int j = 0; j < 8; j++) {
        const float v = m < out_dim ? w3[(size_t)m * H + acc_k(ks, hh, j)] : 0.f;
        const size_t at = ((size_t)ks * 64 + l) * 8 + j;
        f3[at] = split ? bf16(v) : bf16(-2.0f * v);
        if (split) g3[at] = bf16_lo(v);
      }
    for (int r = 0; r < 16; r++) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
      double b = row < out_dim ? (double)b3[row] : 0.0;
      if (!split && row < out_dim)  // (+ W 1 with the weights as they are stored: bf16-rounded)
        for (int k = 0; k < H; k++) b += (double)bf16_to_float(bf16(w3[(size_t)row * H + k]));
      c3[(size_t)l * 16 + r] = (float)b;
    }
  }
  for (int u = 0; u < H; u++) {  // (bf16: the accumulator's initial value; float32-grade: the addend in front of the exponential - skyjo_policy.hip)
    double b = (double)(SKP_SCALE * b2[u]);
    if (!split)
      for (int k = 0; k < H; k++) b += (double)bf16_to_float(bf16(SKP_SCALE * w2[(size_t)u * H + k]));
    c2[u] = (float)b;
  }
  skyjo_vec_mlp *m = new skyjo_vec_mlp();
  m->device_id = device_id, m->obs_dim = obs_dim;
  struct Piece { const void *src; size_t bytes; };
  const Piece pieces[8] = {{f1.data(), e1 * 2}, {f2.data(), e2 * 2}, {f3.data(), e3 * 2}, {c2.data(), c2.size() * 4}, {c3.data(), c3.size() * 4},
                           {g1.data(), g1.size() * 2}, {g2.data(), g2.size() * 2}, {g3.data(), g3.size() * 2}};
  size_t total = 0, offs[8];
  for (int k = 0; k < 8; k++) offs[k] = total, total += (pieces[k].bytes + 255) & ~(size_t)255;
  if (hipMalloc(&m->blob, total) != hipSuccess) {
    delete m;
    return fail(SKYJO_E_DEVICE, "hipMalloc failed for the packed weights");
  }
  uint8_t *p = (uint8_t *)m->blob;
  hipError_t e = hipSuccess;
  for (int k = 0; k < 8 && e == hipSuccess; k++)
    if (pieces[k].bytes) e = hipMemcpy(p + offs[k], pieces[k].src, pieces[k].bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(m->blob);
    delete m;
    return fail(SKYJO_E_DEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e));
  }
  m->net.w1 = (const uint4 *)(p + offs[0]), m->net.w2 = (const uint4 *)(p + offs[1]), m->net.w3 = (const uint4 *)(p + offs[2]);
  m->net.b2 = (const float *)(p + offs[3]), m->net.b3 = (const float *)(p + offs[4]);
  m->net.split = split ? 1 : 0;
  m->net.w1l = split ? (const uint4 *)(p + offs[5]) : nullptr, m->net.w2l = split ? (const uint4 *)(p + offs[6]) : nullptr;
  m->net.w3l = split ? (const uint4 *)(p + offs[7]) : nullptr;
  m->net.out_dim = out_dim;
  *out = m;
  return SKYJO_OK;
}

int skyjo_vec_mlp_destroy(skyjo_vec_mlp *m) {
  if (!m) return SKYJO_OK;
  DevGuard guard_(m->device_id);
  (void)hipFree(m->blob);
  delete m;
  return SKYJO_OK;
}

int skyjo_vec_mlp_forward_layout(const skyjo_vec_mlp *m, const void *records, int32_t record_bytes, int32_t layout, int64_t n, float *out,
                                 void *stream) {
  if (!m || !records || !out || n < 0 || record_bytes < 32 || (record_bytes & 15))
    return fail(SKYJO_E_INVALID, "skyjo_vec_mlp_forward: bad argument");
  if (layout != SKYJO_REC_ROW_MAJOR && layout != SKYJO_REC_TILE_PLANAR) return fail(SKYJO_E_INVALID, "layout must be SKYJO_REC_ROW_MAJOR or SKYJO_REC_TILE_PLANAR");
  if (n == 0) return SKYJO_OK;
  DevGuard guard_(m->device_id);
  SkMlpDraw nodraw{};
  return launch_mlp(m, m, 1, (const uint8_t *)records, (int)record_bytes, m->obs_dim, n, out, nodraw, nullptr, (hipStream_t)stream, nullptr,
                    (int)(layout == SKYJO_REC_TILE_PLANAR));
}
int skyjo_vec_mlp_forward(const skyjo_vec_mlp *m, const void *records, int32_t record_bytes, int64_t n, float *out,
                          void *stream) {
  return skyjo_vec_mlp_forward_layout(m, records, record_bytes, SKYJO_REC_ROW_MAJOR, n, out, stream);
}

int skyjo_vec_mlp_act_value_layout(skyjo_vec *h, const skyjo_vec_mlp *policy, const skyjo_vec_mlp *value, const void *records, int32_t layout,
                                   int64_t n, uint64_t seed, uint64_t ticket, int32_t no_masking, int32_t *actions_out, float *logp_out,
                                   float *logits_out, float *values_out, void *stream) {
  if (!h || !policy || !records || !actions_out || n < 0 || (value && !values_out))
    return fail(SKYJO_E_INVALID, "skyjo_vec_mlp_act_value: bad argument");
  if (layout != SKYJO_REC_ROW_MAJOR && layout != SKYJO_REC_TILE_PLANAR) return fail(SKYJO_E_INVALID, "layout must be SKYJO_REC_ROW_MAJOR or SKYJO_REC_TILE_PLANAR");
  GUARD(h);
  if (policy->net.out_dim != SKYJO_NUM_ACTIONS) return fail(SKYJO_E_INVALID, "the policy net needs 26 outputs");
  if (value && (policy->obs_dim != value->obs_dim || policy->device_id != value->device_id || policy->device_id != h->cfg.device_id ||
                policy->net.split != value->net.split))
    return fail(SKYJO_E_INVALID, "policy and value net must share the observation size, the precision and the engine's device");
  if (n == 0) return SKYJO_OK;
  SkMlpDraw d{};
  d.enable = 1, d.mask_offset = h->P.L.Dp, d.no_masking = no_masking, d.seed = seed, d.ticket = ticket;
  d.game_id0 = h->P.game_id0, d.actions = actions_out, d.logp = logp_out;
  return launch_mlp(policy, value ? value : policy, value ? 2 : 1, (const uint8_t *)records, (int)h->P.L.rec_bytes, policy->obs_dim, n, logits_out,
                    d, value ? values_out : nullptr, (hipStream_t)stream, nullptr, (int)(layout == SKYJO_REC_TILE_PLANAR));
}
int skyjo_vec_mlp_act(skyjo_vec *h, const skyjo_vec_mlp *m, const void *records, int64_t n, uint64_t seed, uint64_t ticket,
                      int32_t no_masking, int32_t *actions_out, float *logp_out, float *logits_out, void *stream) {
  if (!m) return fail(SKYJO_E_INVALID, "skyjo_vec_mlp_act: bad argument");
  return skyjo_vec_mlp_act_value_layout(h, m, nullptr, records, SKYJO_REC_ROW_MAJOR, n, seed, ticket, no_masking, actions_out, logp_out, logits_out,
                                        nullptr, stream);
}
int skyjo_vec_mlp_act_value(skyjo_vec *h, const skyjo_vec_mlp *policy, const skyjo_vec_mlp *value, const void *records, int64_t n,
                            uint64_t seed, uint64_t ticket, int32_t no_masking, int32_t *actions_out, float *logp_out,
                            float *logits_out, float *values_out, void *stream) {
  if (!value || !values_out) return fail(SKYJO_E_INVALID, "skyjo_vec_mlp_act_value: bad argument");
  return skyjo_vec_mlp_act_value_layout(h, policy, value, records, SKYJO_REC_ROW_MAJOR, n, seed, ticket, no_masking, actions_out, logp_out,
                                        logits_out, values_out, stream);
}

int skyjo_vec_episode_ends_layout(skyjo_vec *h, const void *records, int32_t layout, double *final_rewards_out, uint8_t *episode_end_out,
                                  void *stream) {
  if (!h || !records || !final_rewards_out || !episode_end_out) return fail(SKYJO_E_INVALID, "null argument");
  if (layout != SKYJO_REC_ROW_MAJOR && layout != SKYJO_REC_TILE_PLANAR) return fail(SKYJO_E_INVALID, "layout must be SKYJO_REC_ROW_MAJOR or SKYJO_REC_TILE_PLANAR");
  GUARD(h);
  hipLaunchKernelGGL(k_episode_ends, dim3((h->P.B + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->P, (const uint8_t *)records,
                     final_rewards_out, episode_end_out, (int)(layout == SKYJO_REC_TILE_PLANAR));
  HIPCHK(hipGetLastError());
  return SKYJO_OK;
}
int skyjo_vec_episode_ends(skyjo_vec *h, const void *records, double *final_rewards_out, uint8_t *episode_end_out, void *stream) {
  return skyjo_vec_episode_ends_layout(h, records, SKYJO_REC_ROW_MAJOR, final_rewards_out, episode_end_out, stream);
}

int skyjo_vec_get_counters(skyjo_vec *h, skyjo_vec_counters *out, void *stream) {
  if (!h || !out) return fail(SKYJO_E_INVALID, "null argument");
  GUARD(h);
  static_assert(sizeof(SkCounters) == sizeof(skyjo_vec_counters), "counter structs must match");
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(h->P.counters, 0, sizeof(SkCounters), s));
  hipLaunchKernelGGL(k_reduce_stats, dim3(64), dim3(256), 0, s, h->P);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, h->P.counters, sizeof(SkCounters), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  out->iters = h->iters_total;
  return dev_error_fetch(h, s);
}

int skyjo_vec_check_error(skyjo_vec *h, void *stream) {
  if (!h) return fail(SKYJO_E_INVALID, "null handle");
  GUARD(h);
  return dev_error_fetch(h, (hipStream_t)stream);
}

int skyjo_vec_reset_counters(skyjo_vec *h, void *stream) {
  if (!h) return fail(SKYJO_E_INVALID, "null handle");
  GUARD(h);
  const size_t n = (size_t)h->P.tiles * SK_ACC_KINDS * SKYJO_MAX_PLAYERS * sizeof(double);
  h->iters_total = 0;
  HIPCHK(hipMemsetAsync(h->P.counters, 0, sizeof(SkCounters), (hipStream_t)stream));
  HIPCHK(hipMemsetAsync(h->P.tile_counters, 0, (size_t)h->P.tiles * 8 * sizeof(unsigned long long), (hipStream_t)stream));
  HIPCHK(hipMemsetAsync(h->P.acc_tile, 0, n, (hipStream_t)stream));
  return SKYJO_OK;
}

// raw packed record (skyjo_layout.h) -> the canonical form of include/skyjo_vec.h (rewards / final_score are filled by the caller)
static void decode_state(const SkLayout &L, const uint8_t *r, skyjo_game_state *o) {
  memset(o, 0, sizeof(*o));
  for (int p = 0; p < L.N; p++)
    for (int k = 0; k < 12; k++) {
      int8_t c = (int8_t)r[sk_pb(L, p) + PB_CARDS + k], v = (int8_t)r[sk_pb(L, p) + PB_VIS + k];
      o->players_cards[p][k] = c;
      o->players_masked[p][k] = v == SKYJO_HAND_NONE ? 2 : (v == SKYJO_REFUNDED ? 0 : 1);
    }
  const int role = r[H_ROLE];
  o->n_draw = r[H_NDRAW], o->n_disc = r[H_NDISC];
  for (int k = 0; k < o->n_draw; k++) o->drawpile[k] = (int8_t)r[L.off_pile + (role ? SK_NCARDS - 1 - k : k)];
  for (int k = 0; k < o->n_disc; k++) o->discard_pile[k] = (int8_t)r[L.off_pile + (role ? k : SK_NCARDS - 1 - k)];
  o->hand_card = (int8_t)r[H_HAND];
  o->expected_player = r[H_PLAYER], o->expected_phase = r[H_PHASE];
  o->is_terminated = (r[H_FLAGS] & F_TERMINATED) ? 1 : 0;
  o->done = (r[H_FLAGS] & F_DONE) ? 1 : 0;
  o->status = r[H_STATUS];
  memcpy(&o->episode_steps, &r[H_EPLEN], 2);
  memcpy(&o->episode, &r[H_EPISODE], 4);
  o->reshuffles = r[H_RESH];
  for (int p = 0; p < L.N; p++) {
    o->num_refunded[p] = r[sk_pb(L, p) + PB_REFUNDED];
    uint16_t pl;
    memcpy(&pl, &r[sk_pb(L, p) + PB_PLACED], 2);
    o->num_placed[p] = pl;
  }
}

int skyjo_vec_get_state(skyjo_vec *h, int32_t game, skyjo_game_state *o, void *stream) {
  if (!h || !o || game < 0 || game >= h->P.B) return fail(SKYJO_E_INVALID, "bad argument");
  GUARD(h);
  hipStream_t s = (hipStream_t)stream;
  const SkLayout &L = h->P.L;
  if (h->raw_valid) {  // the last *_host call brought every game back with its records: no device traffic
    const uint8_t *raw = h->hm_raw + (size_t)game * h->raw_stride;
    decode_state(L, raw, o);
    const double *d = (const double *)(raw + L.state_bytes);
    if (o->done) {
      memcpy(o->rewards, d, sizeof(double) * L.N);
      if (o->is_terminated) memcpy(o->final_score, d + L.N, sizeof(double) * L.N);
    }
    return dev_error_check(h);
  }
  std::vector<uint8_t> r;
  int rc = fetch_record(h, h->P.state, game, r, s);
  if (rc) return rc;
  decode_state(L, r.data(), o);
  if (o->done) {
    HIPCHK(hipMemcpyAsync(o->rewards, h->P.rewards + (size_t)game * L.N, sizeof(double) * L.N, hipMemcpyDeviceToHost, s));
    if (o->is_terminated)
      HIPCHK(hipMemcpyAsync(o->final_score, h->P.scores + (size_t)game * L.N, sizeof(double) * L.N,
                            hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  return dev_error_check(h);
}

int skyjo_vec_set_state(skyjo_vec *h, int32_t game, const skyjo_game_state *in, void *stream) {
  if (!h || !in || game < 0 || game >= h->P.B) return fail(SKYJO_E_INVALID, "bad argument");
  GUARD(h);
  if (!h->seeded) return fail(SKYJO_E_STATE, "skyjo_vec_seed must be called first");
  h->raw_valid = false;
  const SkLayout &L = h->P.L;
  if (in->n_draw < 0 || in->n_disc < 0 || in->n_draw + in->n_disc > SK_NCARDS)
    return fail(SKYJO_E_INVALID, "pile sizes out of range");
  if (in->expected_player >= L.N || in->expected_phase > 1) return fail(SKYJO_E_INVALID, "bad expected action");
  hipStream_t s = (hipStream_t)stream;
  std::vector<uint8_t> r((size_t)L.state_bytes, 0);
  int ms = 1 << 30, mh = 1 << 30;
  for (int p = 0; p < L.N; p++) {
    int sum = 0, hid = 0;
    for (int k = 0; k < 12; k++) {
      int8_t c = in->players_cards[p][k];
      int m = in->players_masked[p][k];
      if (c < -2 && !(m == 0 && c == SKYJO_REFUNDED)) return fail(SKYJO_E_INVALID, "card value out of range");
      if (c > 12) return fail(SKYJO_E_INVALID, "card value out of range");
      int8_t v = m == 2 ? (int8_t)SKYJO_HAND_NONE : (m == 0 ? (int8_t)SKYJO_REFUNDED : c);
      r[sk_pb(L, p) + PB_CARDS + k] = (uint8_t)(m == 0 ? (int8_t)SKYJO_REFUNDED : c);
      r[sk_pb(L, p) + PB_VIS + k] = (uint8_t)v;
      if (m == 1) {
        sum += c;
        if (!L.indirect) r[H_HIST + 2 + c]++;
      }
      if (m == 2) hid++;
    }
    int16_t s16 = (int16_t)sum;
    memcpy(&r[sk_pb(L, p) + PB_SUM], &s16, 2);
    r[sk_pb(L, p) + PB_HIDDEN] = (uint8_t)hid;
    r[sk_pb(L, p) + PB_REFUNDED] = (uint8_t)in->num_refunded[p];
    uint16_t pl = (uint16_t)in->num_placed[p];
    memcpy(&r[sk_pb(L, p) + PB_PLACED], &pl, 2);
    ms = sum < ms ? sum : ms, mh = hid < mh ? hid : mh;
  }
  for (int k = 0; k < in->n_draw; k++) r[L.off_pile + k] = (uint8_t)in->drawpile[k];
  for (int k = 0; k < in->n_disc; k++) {
    int8_t c = in->discard_pile[k];
    if (c < -2 || c > 12) return fail(SKYJO_E_INVALID, "discard value out of range");
    r[L.off_pile + SK_NCARDS - 1 - k] = (uint8_t)c;
    r[H_HIST + 2 + c]++;
  }
  r[H_PHASE] = in->expected_phase, r[H_PLAYER] = in->expected_player;
  r[H_FLAGS] = F_VALID | (in->is_terminated ? (F_TERMINATED | F_DONE) : 0) | (in->done ? F_DONE : 0);
  r[H_STATUS] = in->status;
  r[H_NDRAW] = (uint8_t)in->n_draw, r[H_NDISC] = (uint8_t)in->n_disc, r[H_ROLE] = 0;
  memcpy(&r[H_EPLEN], &in->episode_steps, 2);
  r[H_RESH] = (uint8_t)(in->reshuffles > 255 ? 255 : in->reshuffles);
  memcpy(&r[H_EPISODE], &in->episode, 4);
  r[H_MINSUM] = (uint8_t)(int8_t)(ms < 127 ? ms : 127);
  r[H_MINHID] = (uint8_t)mh;
  r[H_TOP] = in->n_disc ? (uint8_t)in->discard_pile[in->n_disc - 1] : (uint8_t)(int8_t)-3;
  r[H_HAND] = (uint8_t)in->hand_card;
  HIPCHK(hipMemcpyAsync(&r[H_BANK], h->P.bank_head + game, 1, hipMemcpyDeviceToHost, s));  // keep the bank pointer
  HIPCHK(hipStreamSynchronize(s));
  uint8_t *dst = (uint8_t *)(h->P.state + ((size_t)(game / SK_TILE) * L.chunks) * SK_TILE + game % SK_TILE);
  HIPCHK(hipMemcpy2DAsync(dst, SK_TILE * 16, r.data(), 16, 16, L.chunks, hipMemcpyHostToDevice, s));
  uint8_t dn = (r[H_FLAGS] & F_DONE) ? 1 : 0;
  HIPCHK(hipMemcpyAsync(h->P.done + game, &dn, 1, hipMemcpyHostToDevice, s));
  if (dn) {
    HIPCHK(hipMemcpyAsync(h->P.rewards + (size_t)game * L.N, in->rewards, sizeof(double) * L.N, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->P.scores + (size_t)game * L.N, in->final_score, sizeof(double) * L.N,
                          hipMemcpyHostToDevice, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  return SKYJO_OK;
}

int skyjo_vec_seed_raw(skyjo_vec *h, int32_t game, uint32_t value, void *stream) {
  if (!h || game < 0 || game >= h->P.B) return fail(SKYJO_E_INVALID, "bad argument");
  GUARD(h);
  if (!h->seeded) return fail(SKYJO_E_STATE, "skyjo_vec_seed must be called first");
  if (h->P.rng_mode != SKYJO_RNG_MT19937) return fail(SKYJO_E_STATE, "seed_raw needs the MT19937 mode");
  h->raw_valid = false;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if ((rc = publish_deals(h, s))) return rc;  // nobody else may be using the stream that is re-seeded
  hipLaunchKernelGGL(k_seed_raw, dim3(1), dim3(64), 0, s, h->P, game, value);
  HIPCHK(hipGetLastError());
  if ((rc = start_deals(h, s))) return rc;
  return publish_deals(h, s);
}

int skyjo_vec_rng_set_state(skyjo_vec *h, int32_t game, const uint32_t *key_host, int32_t pos, void *stream) {
  if (!h || !key_host || game < 0 || game >= h->P.B || pos < 0 || pos > 624) return fail(SKYJO_E_INVALID, "bad argument");
  GUARD(h);
  if (!h->seeded) return fail(SKYJO_E_STATE, "skyjo_vec_seed must be called first");
  if (h->P.rng_mode != SKYJO_RNG_MT19937 || !h->no_bank)
    return fail(SKYJO_E_STATE, "skyjo_vec_rng_set_state needs the MT19937 mode and SKYJO_OPT_NO_BANK (episodes dealt ahead would belong to the old stream)");
  hipStream_t s = (hipStream_t)stream;
  // numpy's (key, pos): the words from pos on are regenerated and unconsumed, everything is of one generation - in the
  // engine's terms position pos with 624 - pos words regenerated ahead (MtStream: idx | ahead << 16); pos == 624 is a fresh block
  const int32_t packed = pos >= 624 ? 0 : (pos | ((624 - pos) << 16));
  HIPCHK(hipMemcpyAsync(h->P.mt + (size_t)game * 624, key_host, 624 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(h->P.mt_idx + game, &packed, sizeof(packed), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));  // (`packed` lives on this stack frame, `key_host` belongs to the caller)
  return SKYJO_OK;
}

int skyjo_vec_rng_get_state(skyjo_vec *h, int32_t game, uint32_t *key_out_host, int32_t *pos_out_host, void *stream) {
  if (!h || !key_out_host || !pos_out_host || game < 0 || game >= h->P.B) return fail(SKYJO_E_INVALID, "bad argument");
  GUARD(h);
  if (!h->seeded) return fail(SKYJO_E_STATE, "skyjo_vec_seed must be called first");
  if (h->P.rng_mode != SKYJO_RNG_MT19937 || !h->no_bank)
    return fail(SKYJO_E_STATE, "skyjo_vec_rng_get_state needs the MT19937 mode and SKYJO_OPT_NO_BANK");
  hipStream_t s = (hipStream_t)stream;
  int32_t packed = 0;
  HIPCHK(hipMemcpyAsync(key_out_host, h->P.mt + (size_t)game * 624, 624 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(&packed, h->P.mt_idx + game, sizeof(packed), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  int idx = packed & 0xffff, ahead = packed >> 16;
  if (idx >= 624) idx = 0;
  if (idx + ahead > 624) return fail(SKYJO_E_STATE, "the stream's regenerated window wraps: not a numpy state (cannot happen without a bank)");
  if (idx == 0 && ahead == 0) {  // everything consumed, nothing of the next block made yet: numpy's pos == 624
    *pos_out_host = 624;
    return SKYJO_OK;
  }
  // The engine regenerates lazily, 16 words at a time; numpy makes the whole block at once.  Finish the block: words
  // idx + ahead .. 623 from the old word, its successor and the (already new, for i >= 227) word 397 further on.
  uint32_t *mt = key_out_host;
  for (int i = idx + ahead; i < 624; i++) {
    const uint32_t y = (mt[i] & 0x80000000u) | (mt[i + 1 == 624 ? 0 : i + 1] & 0x7fffffffu);
    mt[i] = mt[i + 397 >= 624 ? i + 397 - 624 : i + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  *pos_out_host = idx;
  return SKYJO_OK;
}

int skyjo_vec_profile(skyjo_vec *h, int enable, double ms_out[SKYJO_PROF_KERNELS], int64_t launches_out[SKYJO_PROF_KERNELS]) {
  if (!h) return fail(SKYJO_E_INVALID, "null handle");
  GUARD(h);
  HIPCHK(hipDeviceSynchronize());
  for (int k = 0; k < SKYJO_PROF_KERNELS; k++) {
    double tot = 0;
    int64_t cnt = 0;
    for (auto &p : h->ev[k]) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) tot += ms, cnt++;
      (void)hipEventDestroy(p.first);
      (void)hipEventDestroy(p.second);
    }
    h->ev[k].clear();
    if (ms_out) ms_out[k] = tot;
    if (launches_out) launches_out[k] = cnt;
  }
  h->profile = enable != 0;
  return SKYJO_OK;
}

int skyjo_vec_debug_stamps(skyjo_vec *h, uint64_t *out16_host) {
  if (!h || !out16_host) return fail(SKYJO_E_INVALID, "null argument");
  GUARD(h);
  const size_t half = (size_t)h->P.tiles * 8;
  std::vector<unsigned long long> t(2 * half);
  HIPCHK(hipMemcpy(t.data(), h->P.stamps, t.size() * 8, hipMemcpyDeviceToHost));
  for (int k = 0; k < 16; k++) out16_host[k] = 0;
  for (size_t i = 0; i < 2 * half; i++) out16_host[(i & 7) + (i >= half ? 8 : 0)] += t[i];  // [0,8) step, [8,16) deal
  HIPCHK(hipMemset(h->P.stamps, 0, t.size() * 8));
  return SKYJO_OK;
}

int skyjo_vec_debug_trace(skyjo_vec *h, uint64_t *out_host) {  // [4][tiles][8]: see TRACE_STORE (skyjo_device.h); zeros unless built with -DSK_TRACE
  if (!h || !out_host) return fail(SKYJO_E_INVALID, "null argument");
  GUARD(h);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out_host, h->P.stamps, (size_t)h->P.tiles * 32 * 8, hipMemcpyDeviceToHost));
  return SKYJO_OK;
}

int skyjo_vec_get_option(const skyjo_vec *h, int option, int64_t *value_out) {
  if (!h || !value_out) return fail(SKYJO_E_INVALID, "null argument");
  switch (option) {
    case SKYJO_OPT_DEAL_INTERVAL: *value_out = h->deal_every_iters; return SKYJO_OK;
    case SKYJO_OPT_OVERLAP: *value_out = h->merged ? 3 : h->overlap ? 2 : 0; return SKYJO_OK;
    case SKYJO_OPT_NO_BANK: *value_out = h->no_bank ? 1 : 0; return SKYJO_OK;
    case SKYJO_OPT_RECORD_LAYOUT: *value_out = h->rec_planar_all ? SKYJO_REC_TILE_PLANAR_ALL : h->rec_planar ? SKYJO_REC_TILE_PLANAR : SKYJO_REC_ROW_MAJOR; return SKYJO_OK;
    case SKYJO_OPT_INLINE_WORK_LIST: *value_out = h->fused_scan ? 0 : 1; return SKYJO_OK;
    case SKYJO_OPT_UNPIPELINED: *value_out = h->piped ? 0 : 1; return SKYJO_OK;
    case SKYJO_OPT_CYCLE_S: *value_out = h->cycle_s; return SKYJO_OK;
    case SKYJO_OPT_MAX_CYCLES_PER_LAUNCH: *value_out = h->max_cycles > 0 ? h->max_cycles : kMaxCyclesPerLaunch; return SKYJO_OK;
    default: return fail(SKYJO_E_INVALID, "unknown option");
  }
}

int skyjo_vec_set_option(skyjo_vec *h, int option, int64_t value) {
  if (!h) return fail(SKYJO_E_INVALID, "null handle");
  GUARD(h);
  switch (option) {
    case SKYJO_OPT_DEAL_INTERVAL:
      if (value < 1 || value > 1024) return fail(SKYJO_E_INVALID, "deal interval must be in 1..1024");
      h->deal_every_iters = (int)value, h->auto_interval = false;
      return SKYJO_OK;
    case SKYJO_OPT_OVERLAP: {
      int rc = publish_deals(h, nullptr);  // drain the pipeline before changing its shape
      if (rc) return rc;
      HIPCHK(hipDeviceSynchronize());
      // 0: in line; 1: beside the step kernel, in the form this engine prefers; 2: the two-stream form; 3: the one-kernel form (k_cycle)
      if (value < 0 || value > 3) return fail(SKYJO_E_INVALID, "SKYJO_OPT_OVERLAP takes 0 .. 3");
      if (value == 3 && !h->merged_capable)
        return fail(SKYJO_E_INVALID, "the one-kernel form needs two to four players");
      h->merged = value == 3 || (value == 1 && h->merged_capable && h->prefer_merged);
      h->overlap = value != 0 && !h->merged;
      h->interval_default = deal_interval_default(h->P.L.N, h->overlap, h->piped, h->merged ? h->cycle_s : 0);
      if (h->auto_interval) h->deal_every_iters = h->interval_default;
      return SKYJO_OK;
    }
    case SKYJO_OPT_NO_BANK:
      if (h->seeded) return fail(SKYJO_E_STATE, "SKYJO_OPT_NO_BANK must be set before skyjo_vec_seed");
      h->no_bank = value != 0;
      return SKYJO_OK;
    case SKYJO_OPT_RECORD_LAYOUT:
      if (value != SKYJO_REC_ROW_MAJOR && value != SKYJO_REC_TILE_PLANAR && value != SKYJO_REC_TILE_PLANAR_ALL)
        return fail(SKYJO_E_INVALID, "SKYJO_OPT_RECORD_LAYOUT takes SKYJO_REC_ROW_MAJOR, SKYJO_REC_TILE_PLANAR or SKYJO_REC_TILE_PLANAR_ALL");
      if (value != SKYJO_REC_ROW_MAJOR && !h->merged_capable)
        return fail(SKYJO_E_INVALID, "the tile-planar record layout exists for the one-kernel form of the fused rollout (two to four players)");
      if (value == SKYJO_REC_TILE_PLANAR_ALL && !h->P.L.indirect)
        return fail(SKYJO_E_INVALID, "SKYJO_REC_TILE_PLANAR_ALL: reset / observe / step write tile-planar records of the indirect observation only");
      h->rec_planar = value != SKYJO_REC_ROW_MAJOR;
      h->rec_planar_all = value == SKYJO_REC_TILE_PLANAR_ALL;
      return SKYJO_OK;
    case SKYJO_OPT_INLINE_WORK_LIST:
      h->fused_scan = value == 0;
      return SKYJO_OK;
    case SKYJO_OPT_UNPIPELINED: {
      int rc = publish_deals(h, nullptr);  // drain the pipeline before changing its shape
      if (rc) return rc;
      HIPCHK(hipDeviceSynchronize());
      h->piped = value == 0;
      h->interval_default = deal_interval_default(h->P.L.N, h->overlap, h->piped, h->merged ? h->cycle_s : 0);
      if (h->auto_interval) h->deal_every_iters = h->interval_default;
      return SKYJO_OK;
    }
    case SKYJO_OPT_CYCLE_S: {
      if (h->seeded) return fail(SKYJO_E_STATE, "SKYJO_OPT_CYCLE_S must be set before skyjo_vec_seed");
      if (value < 0 || value > SK_CYCLE_MAX_S) return fail(SKYJO_E_INVALID, "SKYJO_OPT_CYCLE_S takes 0 (the batch's share) .. 4");
      const bool was = h->merged;
      configure_cycle(h, (int)value);
      if (was && !h->merged_capable) return fail(SKYJO_E_INVALID, "that many step regions do not fit a compute unit's LDS");
      h->interval_default = deal_interval_default(h->P.L.N, h->overlap, h->piped, h->merged ? h->cycle_s : 0);
      if (h->auto_interval) h->deal_every_iters = h->interval_default;
      return SKYJO_OK;
    }
    case SKYJO_OPT_MAX_CYCLES_PER_LAUNCH:
      if (value < 1 || value > kMaxCyclesPerLaunch) return fail(SKYJO_E_INVALID, "SKYJO_OPT_MAX_CYCLES_PER_LAUNCH takes 1 .. 16");
      h->max_cycles = (int)value;
      return SKYJO_OK;
    case SKYJO_OPT_DEBUG_SPIN_LOG2:
      if (value < 1 || value > 30) return fail(SKYJO_E_INVALID, "spin limit must be 2^1 .. 2^30");
      h->P.spin_log2 = (uint32_t)value;
      return SKYJO_OK;
    case SKYJO_OPT_DEBUG_DEAL_DELAY:
      if (value < 0 || value > (1 << 24)) return fail(SKYJO_E_INVALID, "deal delay out of range");
      h->P.debug_deal_delay = (uint32_t)value;
      return SKYJO_OK;
    default:
      return fail(SKYJO_E_INVALID, "unknown option");
  }
}

// ---- host-pointer conveniences ------------------------------------------------------------
static int ensure_scratch(skyjo_vec *h) {
  int rc;
  if (!h->d_actions && (rc = dalloc(h, &h->d_actions, h->G))) return rc;
  if (!h->d_records && (rc = dalloc(h, &h->d_records, h->G * (size_t)h->P.L.rec_bytes))) return rc;
  if (!h->d_mask && (rc = dalloc(h, &h->d_mask, h->G))) return rc;
  return SKYJO_OK;
}

int skyjo_vec_step_host(skyjo_vec *h, const int32_t *actions_host, void *records_out_host) {
  if (!h || !actions_host) return fail(SKYJO_E_INVALID, "null argument");
  GUARD(h);
  if (!h->seeded) return fail(SKYJO_E_STATE, "skyjo_vec_seed must be called first");
  if (h->fast_host) {  // actions read from / records and whole games written to host-mapped memory: one launch, one synchronisation
    memcpy(h->hm_actions, actions_host, sizeof(int32_t) * (size_t)h->P.B);
    // A single tile is a single wavefront: it signs off with a sequence number in host-mapped memory as its very last
    // store, and the host spins on that word - a stream synchronisation costs ~10 us more than the kernel takes.  (The
    // stream stays ordered: whatever is launched next runs behind this kernel as usual.)
    const bool spin = h->P.tiles == 1 && h->host_spin;
    if (spin) {
      h->host_seq = h->host_seq + 1 ? h->host_seq + 1 : 1;
      h->P.host_seq = h->host_seq;
    }
    int rc = step_once(h, h->hm_actions_d, h->hm_records_d, nullptr, nullptr, nullptr, h->raw_export ? h->hm_raw_d : nullptr);
    h->P.host_seq = 0;
    if (rc) return rc;
    bool arrived = false;
    if (spin) {
      const volatile uint32_t *flag = h->health_host + 3;
      for (long k = 0; k < 4000000 && !(arrived = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == h->host_seq); k++) __builtin_ia32_pause();
    }
    if (!arrived) HIPCHK(hipStreamSynchronize(nullptr));  // (also the fallback when the word does not come: ~0.1 s of spinning)
    h->raw_valid = h->raw_export;
    if (records_out_host) memcpy(records_out_host, h->hm_records, (size_t)h->P.B * h->P.L.rec_bytes);
    return dev_error_check(h);
  }
  int rc = ensure_scratch(h);
  if (rc) return rc;
  HIPCHK(hipMemcpy(h->d_actions, actions_host, sizeof(int32_t) * (size_t)h->P.B, hipMemcpyHostToDevice));
  if ((rc = skyjo_vec_step(h, h->d_actions, h->d_records, nullptr))) return rc;
  if (records_out_host)
    HIPCHK(hipMemcpy(records_out_host, h->d_records, (size_t)h->P.B * h->P.L.rec_bytes, hipMemcpyDeviceToHost));
  else
    HIPCHK(hipStreamSynchronize(nullptr));
  return dev_error_check(h);
}

int skyjo_vec_observe_host(skyjo_vec *h, const int32_t *players_host, void *records_out_host) {
  if (!h || !records_out_host) return fail(SKYJO_E_INVALID, "null argument");
  GUARD(h);
  if (h->fast_host) {  // (the games do not change: a valid host copy of them stays valid)
    if (players_host) memcpy(h->hm_actions, players_host, sizeof(int32_t) * (size_t)h->P.B);
    int rc = observe_impl(h, players_host ? h->hm_actions_d : nullptr, h->hm_records_d, nullptr, 0);  // (host-style records are row-major whatever the option)
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    memcpy(records_out_host, h->hm_records, (size_t)h->P.B * h->P.L.rec_bytes);
    return dev_error_check(h);
  }
  int rc = ensure_scratch(h);
  if (rc) return rc;
  if (players_host)
    HIPCHK(hipMemcpy(h->d_actions, players_host, sizeof(int32_t) * (size_t)h->P.B, hipMemcpyHostToDevice));
  if ((rc = observe_impl(h, players_host ? h->d_actions : nullptr, h->d_records, nullptr, 0))) return rc;
  HIPCHK(hipMemcpy(records_out_host, h->d_records, (size_t)h->P.B * h->P.L.rec_bytes, hipMemcpyDeviceToHost));
  return dev_error_check(h);
}

int skyjo_vec_reset_host(skyjo_vec *h, const uint8_t *mask_host, void *records_out_host) {
  if (!h) return fail(SKYJO_E_INVALID, "null handle");
  GUARD(h);
  if (!h->seeded) return fail(SKYJO_E_STATE, "skyjo_vec_seed must be called first");
  if (h->fast_host) {
    if (mask_host) memcpy(h->hm_mask, mask_host, (size_t)h->P.B);
    int rc = reset_impl(h, mask_host ? h->hm_mask_d : nullptr, h->hm_records_d, nullptr, h->raw_export ? h->hm_raw_d : nullptr);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    h->raw_valid = h->raw_export;
    if (records_out_host) memcpy(records_out_host, h->hm_records, (size_t)h->P.B * h->P.L.rec_bytes);
    return dev_error_check(h);
  }
  int rc = ensure_scratch(h);
  if (rc) return rc;
  if (mask_host) HIPCHK(hipMemcpy(h->d_mask, mask_host, (size_t)h->P.B, hipMemcpyHostToDevice));
  if ((rc = skyjo_vec_reset(h, mask_host ? h->d_mask : nullptr, h->d_records, nullptr))) return rc;
  if (records_out_host)
    HIPCHK(hipMemcpy(records_out_host, h->d_records, (size_t)h->P.B * h->P.L.rec_bytes, hipMemcpyDeviceToHost));
  else
    HIPCHK(hipStreamSynchronize(nullptr));
  return dev_error_check(h);
}

int skyjo_vec_get_rewards_host(skyjo_vec *h, double *rewards_out, double *scores_out, uint8_t *done_out) {
  if (!h) return fail(SKYJO_E_INVALID, "null handle");
  GUARD(h);
  const size_t n = (size_t)h->P.B * h->P.L.N;
  if (h->raw_valid) {  // served from what the last *_host call brought back
    const size_t N = (size_t)h->P.L.N;
    for (int g = 0; g < h->P.B; g++) {
      const uint8_t *raw = h->hm_raw + (size_t)g * h->raw_stride + h->P.L.state_bytes;
      if (rewards_out) memcpy(rewards_out + g * N, raw, 8 * N);
      if (scores_out) memcpy(scores_out + g * N, raw + 8 * N, 8 * N);
      if (done_out) done_out[g] = raw[16 * N + 4];
    }
    return dev_error_check(h);
  }
  if (rewards_out) HIPCHK(hipMemcpy(rewards_out, h->P.rewards, n * sizeof(double), hipMemcpyDeviceToHost));
  if (scores_out) HIPCHK(hipMemcpy(scores_out, h->P.scores, n * sizeof(double), hipMemcpyDeviceToHost));
  if (done_out) HIPCHK(hipMemcpy(done_out, h->P.done, (size_t)h->P.B, hipMemcpyDeviceToHost));
  return dev_error_check(h);
}

int skyjo_dev_malloc(int device_id, size_t bytes, void **out) {
  if (!out) return fail(SKYJO_E_INVALID, "null argument");
  DevGuard guard_(device_id);
  HIPCHK(hipMalloc(out, bytes ? bytes : 1));
  return SKYJO_OK;
}
int skyjo_dev_free(void *p) {
  if (p) HIPCHK(hipFree(p));
  return SKYJO_OK;
}
// ---- the reference's scoring helpers for caller-supplied hands (host pointers; computed on the device) ----
namespace {
struct DevBuf {  // a scratch allocation that frees itself
  void *p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};
int score_device(int32_t device_id, int *ndev_out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SKYJO_E_NOGPU, "no HIP device visible: this library has no CPU fallback");
  if (device_id < 0 || device_id >= ndev) return fail(SKYJO_E_INVALID, "device_id out of range");
  *ndev_out = ndev;
  return SKYJO_OK;
}
}  // namespace

int skyjo_vec_evaluate_game(int32_t device_id, int32_t n, int32_t num_players, const int8_t *players_cards_host, const int32_t *player_won_id_host,
                        double score_penalty, double *scores_out_host) {
  if (!players_cards_host || !player_won_id_host || !scores_out_host || n < 0) return fail(SKYJO_E_INVALID, "skyjo_vec_evaluate_game: bad argument");
  if (num_players <= 0 || num_players > SKYJO_MAX_PLAYERS) return fail(SKYJO_E_INVALID, "skyjo_vec_evaluate_game: num_players must be 1..12");
  for (int32_t i = 0; i < n; i++)
    if (player_won_id_host[i] < 0 || player_won_id_host[i] >= num_players) return fail(SKYJO_E_INVALID, "skyjo_vec_evaluate_game: player_won_id out of range");
  int ndev, rc;
  if ((rc = score_device(device_id, &ndev))) return rc;
  if (n == 0) return SKYJO_OK;
  DevGuard guard_(device_id);
  const size_t cb = (size_t)n * num_players * 12, wb = (size_t)n * 4, sb = (size_t)n * num_players * 8;
  const size_t o_w = (cb + 255) & ~(size_t)255, o_s = o_w + ((wb + 255) & ~(size_t)255);
  DevBuf d;
  HIPCHK(hipMalloc(&d.p, o_s + sb));
  uint8_t *b = (uint8_t *)d.p;
  HIPCHK(hipMemcpy(b, players_cards_host, cb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b + o_w, player_won_id_host, wb, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_evaluate_game, dim3((n + 63) / 64), dim3(64), 0, nullptr, (int)n, (int)num_players, (const int8_t *)b, (const int32_t *)(b + o_w),
                     score_penalty, (double *)(b + o_s));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(scores_out_host, b + o_s, sb, hipMemcpyDeviceToHost));
  return SKYJO_OK;
}

int skyjo_vec_calc_final_rewards(int32_t device_id, int32_t n, int32_t num_players, const double *final_score_host, const int32_t *num_refunded_host,
                             double mean_reward, double reward_refunded, double *rewards_out_host) {
  if (!final_score_host || !num_refunded_host || !rewards_out_host || n < 0) return fail(SKYJO_E_INVALID, "skyjo_vec_calc_final_rewards: bad argument");
  if (num_players <= 0 || num_players > SKYJO_MAX_PLAYERS) return fail(SKYJO_E_INVALID, "skyjo_vec_calc_final_rewards: num_players must be 1..12");
  int ndev, rc;
  if ((rc = score_device(device_id, &ndev))) return rc;
  if (n == 0) return SKYJO_OK;
  DevGuard guard_(device_id);
  const size_t sb = (size_t)n * num_players * 8, fb = (size_t)n * num_players * 4;
  const size_t o_f = (sb + 255) & ~(size_t)255, o_r = o_f + ((fb + 255) & ~(size_t)255);
  DevBuf d;
  HIPCHK(hipMalloc(&d.p, o_r + sb));
  uint8_t *b = (uint8_t *)d.p;
  HIPCHK(hipMemcpy(b, final_score_host, sb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b + o_f, num_refunded_host, fb, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_final_rewards, dim3((n + 63) / 64), dim3(64), 0, nullptr, (int)n, (int)num_players, (const double *)b, (const int32_t *)(b + o_f),
                     mean_reward, reward_refunded, (double *)(b + o_r));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(rewards_out_host, b + o_r, sb, hipMemcpyDeviceToHost));
  return SKYJO_OK;
}

int skyjo_dev_copy(void *dst, const void *src, size_t bytes, int kind, void *stream) {
  hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  HIPCHK(hipMemcpyAsync(dst, src, bytes, k, (hipStream_t)stream));
  if (kind == 2) HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return SKYJO_OK;
}
int skyjo_dev_sync(void *stream) {
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return SKYJO_OK;
}

}  // extern "C"
