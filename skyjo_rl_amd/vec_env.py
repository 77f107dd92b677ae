"""SkyjoVecEnv - batched SkyJo games on one MI355X through the C ABI of libskyjo_vec.so.

The class is a thin owner of a native handle.  Two call styles:

* device style (torch tensors on the GPU, zero host traffic): ``step(actions)``, ``rollout(k)``,
  ``observe()``, ``reset()`` return the raw output records as a ``torch.uint8`` tensor
  ``[..., record_bytes]``; ``split(records)`` gives zero-copy strided views named like the
  reference's observation dict (``observations`` int8[..., D], ``action_mask`` int8[..., 26],
  rlskyjo/environment/skyjo_env.py:199-214) plus ``agent``, ``phase``, ``done``, ``status``.
* host style (numpy, synchronous; used by the single-game AEC / SkyjoGame views):
  ``step_host``, ``observe_host``, ``reset_host``, ``rewards_host``.

There is no CPU implementation behind this class: construction fails without the HIP library
and a gfx950 device.
"""
import collections
import ctypes as C

import numpy as np

from . import _lib

Obs = collections.namedtuple("Obs", "observations action_mask agent phase done status episode_steps action",
                             defaults=(None,))


def split_records_np(rec, obs_dim):
    """Strided numpy views into host records [..., record_bytes] (layout: include/skyjo_vec.h)."""
    rec = np.asarray(rec)
    dp = (obs_dim + 3) & ~3
    i8 = rec.view(np.int8)
    steps = rec[..., dp + 30].astype(np.uint16) | (rec[..., dp + 31].astype(np.uint16) << 8)
    return Obs(i8[..., :obs_dim], i8[..., dp:dp + 26], rec[..., dp + 26], rec[..., dp + 27], rec[..., dp + 28],
               rec[..., dp + 29], steps, i8[..., obs_dim])


class _Snapshot:
    def __init__(self, L, p):
        self._L, self._p = L, p

    @property
    def nbytes(self):
        n = C.c_size_t()
        _lib.check(self._L.skyjo_vec_snapshot_bytes(self._p, C.byref(n)))
        return int(n.value)

    def close(self):
        if self._p is not None and self._p.value:
            self._L.skyjo_vec_snapshot_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SkyjoVecEnv:
    ACTION_SKIP = _lib.ACTION_SKIP

    def __init__(self, num_envs, num_players=3, score_penalty=2.0, observe_other_player_indirect=True,
                 mean_reward=1.0, reward_refunded=0.001, device=0, rng_mode=_lib.RNG_MT19937, auto_reset=True,
                 game_id0=0, illegal_reward=-1.0, no_bank=False):
        # same precondition and message as rlskyjo/game/skyjo.py:24-26
        assert 0 < num_players <= 12, "Skyjo can be played from 1 up to 8 (recommended) / 12 (theoretical) players"
        self._L = _lib.load()
        self._h = C.c_void_p()
        cfg = _lib.Config(_lib.ABI_VERSION, int(num_envs), int(num_players), int(bool(observe_other_player_indirect)),
                          float(score_penalty), float(mean_reward), float(reward_refunded), float(illegal_reward),
                          int(device), int(rng_mode), int(bool(auto_reset)), 0, int(game_id0))
        _lib.check(self._L.skyjo_vec_create(C.byref(cfg), C.byref(self._h)))
        info = _lib.Info()
        _lib.check(self._L.skyjo_vec_get_info(self._h, C.byref(info)))
        self.num_envs, self.num_players = info.num_envs, info.num_players
        self.obs_dim, self.record_bytes = info.obs_dim, info.record_bytes
        self.mask_offset, self.meta_offset, self.state_bytes = info.mask_offset, info.meta_offset, info.state_bytes
        self.device_index = int(device)
        self.rng_mode, self.auto_reset = int(rng_mode), bool(auto_reset)
        self.game_id0 = int(game_id0)
        self.obs_shape = (self.obs_dim,)       # skyjo.py:43-45
        self.action_mask_shape = (26,)         # skyjo.py:46
        self.no_bank = bool(no_bank)
        if no_bank:  # SKYJO_OPT_NO_BANK: deals are made in place, when the reference makes them (global-RNG single-game views)
            _lib.check(self._L.skyjo_vec_set_option(self._h, 5, 1))

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.skyjo_vec_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ torch plumbing
    @staticmethod
    def _torch():
        import torch
        return torch

    def _stream(self):
        torch = self._torch()
        return C.c_void_p(torch.cuda.current_stream(self.device_index).cuda_stream)

    def _dev(self):
        return self._torch().device("cuda", self.device_index)

    def new_records(self, *lead):
        return self._torch().empty((*lead, self.num_envs, self.record_bytes), dtype=self._torch().uint8,
                                   device=self._dev())

    # ------------------------------------------------------------------ tile-planar records of the fused rollout
    @property
    def tiles(self):
        return (self.num_envs + 63) // 64

    def set_record_layout(self, layout):
        """'row-major' (default), 'tile-planar' or 'tile-planar-all' (include/skyjo_vec.h: SKYJO_OPT_RECORD_LAYOUT): how ``rollout`` - and
        with '-all' ``reset`` / ``observe`` / ``step`` / the model rollout as well - lays out its records.  Tile-planar records come as
        ``new_planar_records(iters)`` = uint8 [iters, tiles, record_bytes / 16, 64, 16] (one iteration: ``new_planar_records()``);
        ``FusedNet`` / ``sample_actions`` / ``episode_ends`` read them in place (``planar=True``)."""
        _lib.check(self._L.skyjo_vec_set_option(self._h, _lib.OPT_RECORD_LAYOUT,
                                                {"row-major": _lib.REC_ROW_MAJOR, "tile-planar": _lib.REC_TILE_PLANAR,
                                                 "tile-planar-all": _lib.REC_TILE_PLANAR_ALL}[layout]))

    @property
    def record_layout(self):
        """The layout ``rollout`` writes, as the engine has it (SKYJO_OPT_RECORD_LAYOUT)."""
        v = C.c_int64()
        _lib.check(self._L.skyjo_vec_get_option(self._h, _lib.OPT_RECORD_LAYOUT, C.byref(v)))
        return {_lib.REC_ROW_MAJOR: "row-major", _lib.REC_TILE_PLANAR: "tile-planar", _lib.REC_TILE_PLANAR_ALL: "tile-planar-all"}[int(v.value)]

    def _new_step_records(self):
        """One iteration's records in the layout ``reset`` / ``observe`` / ``step`` write."""
        return self.new_planar_records() if self.record_layout == "tile-planar-all" else self.new_records()

    def new_planar_records(self, *iters):
        return self._torch().empty((*iters, self.tiles, self.record_bytes // 16, 64, 16), dtype=self._torch().uint8, device=self._dev())

    def rows_from_planar(self, records):
        """Row-major copy [iters, num_envs, record_bytes] of tile-planar records (``split`` / ``unpack`` take it from there)."""
        if records.dim() == 4:  # one iteration
            return records.permute(0, 2, 1, 3).reshape(records.shape[0] * 64, self.record_bytes)[:self.num_envs]
        it, t = records.shape[0], records.shape[1]
        return records.permute(0, 1, 3, 2, 4).reshape(it, t * 64, self.record_bytes)[:, :self.num_envs]

    def unpack_tiles(self, records):
        """Dense obs int8[n, D], mask int8[n, 26] straight from tile-planar blocks (n = 64 x number of blocks)."""
        torch = self._torch()
        nt = records.numel() // (64 * self.record_bytes)
        obs = torch.empty((nt * 64, self.obs_dim), dtype=torch.int8, device=self._dev())
        mask = torch.empty((nt * 64, 26), dtype=torch.int8, device=self._dev())
        _lib.check(self._L.skyjo_vec_unpack_tiles(self._h, C.c_void_p(records.data_ptr()), nt, C.c_void_p(obs.data_ptr()),
                                                  C.c_void_p(mask.data_ptr()), None, None, None, None, self._stream()))
        return obs, mask

    def split(self, records):
        """Zero-copy views of a records tensor: the reference's {"observations","action_mask"} + meta."""
        torch = self._torch()
        dp = self.mask_offset
        i8 = records.view(torch.int8)
        steps = records[..., dp + 30].to(torch.int32) | (records[..., dp + 31].to(torch.int32) << 8)
        return Obs(i8[..., :self.obs_dim], i8[..., dp:dp + 26], records[..., dp + 26], records[..., dp + 27],
                   records[..., dp + 28], records[..., dp + 29], steps, i8[..., self.obs_dim])

    # ------------------------------------------------------------------ device style API
    def seed(self, seeds=None, base_seed=0):
        """SkyjoGame.set_seed per game (skyjo.py:84-88): game i <- seeds[i] (default base_seed + game id)."""
        ptr = None
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
            assert seeds.shape == (self.num_envs,)
            ptr = seeds.ctypes.data_as(C.c_void_p)
        _lib.check(self._L.skyjo_vec_seed(self._h, ptr, int(base_seed), None))
        _lib.check(self._L.skyjo_dev_sync(None))

    def seed_one(self, game, value):
        """SkyjoGame.set_seed(value) (skyjo.py:84-88) for ONE game of the batch; the others are untouched."""
        _lib.check(self._L.skyjo_vec_seed_one(self._h, int(game), int(value), None))
        _lib.check(self._L.skyjo_dev_sync(None))

    def reset(self, mask=None, out=None):
        torch = self._torch()
        out = self._new_step_records() if out is None else out
        mp = None
        if mask is not None:
            mask = mask.to(device=self._dev(), dtype=torch.uint8).contiguous()
            mp = C.c_void_p(mask.data_ptr())
        _lib.check(self._L.skyjo_vec_reset(self._h, mp, C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def step(self, actions, out=None):
        """SimpleSkyjoEnv.step for every game (skyjo_env.py:216-252); actions: int32 cuda tensor [num_envs]."""
        torch = self._torch()
        assert actions.is_cuda and actions.dtype == torch.int32 and actions.is_contiguous()
        assert actions.numel() == self.num_envs
        out = self._new_step_records() if out is None else out
        _lib.check(self._L.skyjo_vec_step(self._h, C.c_void_p(actions.data_ptr()), C.c_void_p(out.data_ptr()),
                                          self._stream()))
        return out

    def rollout(self, iters, policy_seed=0, records=None, actions=None):
        """`iters` lockstep iterations with the on-device random admissible policy.

        records: None or uint8 cuda tensor [iters, num_envs, record_bytes]; actions: None or int32 [iters, num_envs].
        """
        rp = C.c_void_p(records.data_ptr()) if records is not None else None
        ap = C.c_void_p(actions.data_ptr()) if actions is not None else None
        if records is not None:
            planar = self.record_layout != "row-major"
            per_it = self.tiles * 64 if planar else self.num_envs
            assert records.is_contiguous() and records.numel() == iters * per_it * self.record_bytes
        if actions is not None:
            assert actions.is_contiguous() and actions.numel() == iters * self.num_envs
        _lib.check(self._L.skyjo_vec_rollout(self._h, int(iters), int(policy_seed), rp, ap, self._stream()))

    def observe(self, players=None, out=None):
        out = self._new_step_records() if out is None else out
        pp = C.c_void_p(players.data_ptr()) if players is not None else None
        _lib.check(self._L.skyjo_vec_observe(self._h, pp, C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def unpack(self, records):
        """Dense copies shaped like the reference's arrays: obs int8[n, D], mask int8[n, 26]."""
        torch = self._torch()
        n = records.numel() // self.record_bytes
        obs = torch.empty((n, self.obs_dim), dtype=torch.int8, device=self._dev())
        mask = torch.empty((n, 26), dtype=torch.int8, device=self._dev())
        _lib.check(self._L.skyjo_vec_unpack(self._h, C.c_void_p(records.data_ptr()), n, C.c_void_p(obs.data_ptr()),
                                            C.c_void_p(mask.data_ptr()), None, None, None, None, self._stream()))
        return obs, mask

    def sample_actions(self, logits, records, seed=0, ticket=0, no_masking=False, actions=None, logp=None, uniform=None, planar=False):
        """Masked categorical draw of config 5 (rlskyjo/models/action_mask_model.py:58-74 + the sampling RLlib does on
        the masked logits) fused on the GPU: ``logits`` float32 [n, 26]; the action mask is read in place from
        ``records`` (``planar``: one iteration's tile-planar block [tiles, P, 64, 16], n = num_envs).  Returns int32 actions;
        ``logp`` / ``uniform`` (float32 [n]) are filled when given."""
        torch = self._torch()
        n = self.num_envs if planar else records.numel() // self.record_bytes
        assert not planar or records.numel() == self.tiles * 64 * self.record_bytes
        assert logits.dtype == torch.float32 and logits.is_contiguous() and logits.numel() == n * 26
        assert records.is_contiguous()
        if actions is None:
            actions = torch.empty((n,), dtype=torch.int32, device=self._dev())
        vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(self._L.skyjo_vec_sample_actions_layout(self._h, vp(records), _lib.REC_TILE_PLANAR if planar else _lib.REC_ROW_MAJOR,
                                                           vp(logits), n, int(seed), int(ticket), 1 if no_masking else 0, vp(actions),
                                                           vp(logp), vp(uniform), self._stream()))
        return actions

    def episode_ends(self, records, final_rewards=None, episode_end=None, planar=False):
        """For the records a step has just written: ``episode_end`` uint8 [num_envs] (1 where that step ended the episode) and
        ``final_rewards`` float64 [num_envs, num_players] (skyjo_env.py:293-312; zeros elsewhere) - skyjo_vec_episode_ends."""
        torch = self._torch()
        if final_rewards is None:
            final_rewards = torch.empty((self.num_envs, self.num_players), dtype=torch.float64, device=self._dev())
        if episode_end is None:
            episode_end = torch.empty((self.num_envs,), dtype=torch.uint8, device=self._dev())
        assert records.is_contiguous() and records.numel() == (self.tiles * 64 if planar else self.num_envs) * self.record_bytes
        vp = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(self._L.skyjo_vec_episode_ends_layout(self._h, vp(records), _lib.REC_TILE_PLANAR if planar else _lib.REC_ROW_MAJOR,
                                                         vp(final_rewards), vp(episode_end), self._stream()))
        return final_rewards, episode_end

    def rewards_tensor(self):
        torch = self._torch()
        out = torch.empty((self.num_envs, self.num_players), dtype=torch.float64, device=self._dev())
        src = self._L.skyjo_vec_rewards_ptr(self._h)
        _lib.check(self._L.skyjo_dev_copy(C.c_void_p(out.data_ptr()), C.c_void_p(src), out.numel() * 8, 3,
                                          self._stream()))
        return out

    def sync(self):
        _lib.check(self._L.skyjo_dev_sync(None))

    def check_error(self):
        """Synchronise and raise if the engine carries a sticky device error (include/skyjo_vec.h: skyjo_vec_check_error)."""
        _lib.check(self._L.skyjo_vec_check_error(self._h, self._stream()))

    # ------------------------------------------------------------------ host style API
    def _host_records(self):
        return np.zeros((self.num_envs, self.record_bytes), dtype=np.uint8)

    def step_host(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.shape == (self.num_envs,)
        rec = self._host_records()
        _lib.check(self._L.skyjo_vec_step_host(self._h, a.ctypes.data_as(C.c_void_p), rec.ctypes.data_as(C.c_void_p)))
        return split_records_np(rec, self.obs_dim)

    def step_one(self, game, action):
        """Host-style step of ONE game (every other game gets ACTION_SKIP) with nothing allocated per call: the single-game
        views' fast path.  Returns that game's record as a uint8 row (a view that the next ``step_one`` overwrites)."""
        one = getattr(self, "_one", None)
        if one is None:
            acts = np.full(self.num_envs, self.ACTION_SKIP, dtype=np.int32)
            rec = np.zeros((self.num_envs, self.record_bytes), dtype=np.uint8)
            one = self._one = (acts, rec, acts.ctypes.data_as(C.c_void_p), rec.ctypes.data_as(C.c_void_p))
        acts, rec, pa, pr = one
        acts[game] = action
        rc = self._L.skyjo_vec_step_host(self._h, pa, pr)
        acts[game] = self.ACTION_SKIP
        _lib.check(rc)
        return rec[game]

    def observe_host(self, players=None):
        rec = self._host_records()
        pp = None
        if players is not None:
            players = np.ascontiguousarray(players, dtype=np.int32)
            pp = players.ctypes.data_as(C.c_void_p)
        _lib.check(self._L.skyjo_vec_observe_host(self._h, pp, rec.ctypes.data_as(C.c_void_p)))
        return split_records_np(rec, self.obs_dim)

    def reset_host(self, mask=None):
        rec = self._host_records()
        mp = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            mp = mask.ctypes.data_as(C.c_void_p)
        _lib.check(self._L.skyjo_vec_reset_host(self._h, mp, rec.ctypes.data_as(C.c_void_p)))
        return split_records_np(rec, self.obs_dim)

    def rewards_host(self):
        rew = np.zeros((self.num_envs, self.num_players), dtype=np.float64)
        sc = np.zeros((self.num_envs, self.num_players), dtype=np.float64)
        done = np.zeros(self.num_envs, dtype=np.uint8)
        _lib.check(self._L.skyjo_vec_get_rewards_host(self._h, rew.ctypes.data_as(C.c_void_p),
                                                      sc.ctypes.data_as(C.c_void_p), done.ctypes.data_as(C.c_void_p)))
        return rew, sc, done

    def rollout_host(self, iters, policy_seed=0):
        """rollout without recording (counters only), synchronous."""
        _lib.check(self._L.skyjo_vec_rollout(self._h, int(iters), int(policy_seed), None, None, None))
        _lib.check(self._L.skyjo_dev_sync(None))

    # ------------------------------------------------------------------ counters / state
    def counters(self):
        c = _lib.Counters()
        _lib.check(self._L.skyjo_vec_get_counters(self._h, C.byref(c), None))
        N = self.num_players
        return dict(steps=c.steps, episodes=c.episodes, illegal=c.illegal, resets=c.resets, sum_len=c.sum_len,
                    reshuffles=c.reshuffles, iters=c.iters, waits=c.waits,
                    sum_score=np.array(c.sum_score[:N]), sum_reward=np.array(c.sum_reward[:N]),
                    sum_reward_sq=np.array(c.sum_reward_sq[:N]), sum_refunded=np.array(c.sum_refunded[:N]))

    def profile(self, enable):
        """Collect-and-clear HIP-event timings of the kernels launched since the last call (bench roofline leg)."""
        K = len(_lib.PROF_KERNELS)
        ms, n = (C.c_double * K)(), (C.c_int64 * K)()
        _lib.check(self._L.skyjo_vec_profile(self._h, int(enable), ms, n))
        out = {}
        for k, name in enumerate(_lib.PROF_KERNELS):
            out[name + "_ms"], out[name + "_launches"] = ms[k], n[k]
        out.update(step_ms=ms[0], step_launches=n[0], deal_ms=ms[2], deal_launches=n[2])
        return out

    # ------------------------------------------------------------------ snapshot / restore
    def snapshot(self):
        """Device-resident copy of the whole engine (games, banks, RNG streams and positions, statistics, policy
        counter).  ``restore(snap)`` makes every later call repeat what followed the snapshot."""
        sp = C.c_void_p()
        _lib.check(self._L.skyjo_vec_snapshot_create(self._h, C.byref(sp), None))
        return _Snapshot(self._L, sp)

    def restore(self, snap):
        _lib.check(self._L.skyjo_vec_snapshot_restore(self._h, snap._p, None))

    def set_deal_interval(self, n):
        _lib.check(self._L.skyjo_vec_set_option(self._h, 1, int(n)))

    def deal_interval(self):
        """Lockstep iterations between two dealing runs right now (it adapts itself unless set explicitly)."""
        v = C.c_int64()
        _lib.check(self._L.skyjo_vec_get_option(self._h, 1, C.byref(v)))
        return int(v.value)

    def overlap(self):
        """True when the dealing kernel runs on its own stream beside the step kernels (default up to 768 tiles = 49 152 games)."""
        v = C.c_int64()
        _lib.check(self._L.skyjo_vec_get_option(self._h, 2, C.byref(v)))
        return bool(v.value)

    def set_overlap(self, on):
        """Dealing beside the step kernel (results do not depend on it).  False / 0: in line; True / 1: beside, in the form the
        engine prefers; 2: the two-stream form (k_deal on a stream of its own); 3: the one-kernel form (k_cycle: step and dealing
        wavefronts share every CU; two to four players)."""
        _lib.check(self._L.skyjo_vec_set_option(self._h, 2, int(on)))

    def set_option(self, option, value):
        """skyjo_vec_set_option with one of ``_lib.OPT_*`` (include/skyjo_vec.h)."""
        _lib.check(self._L.skyjo_vec_set_option(self._h, int(option), int(value)))

    def dealing_form(self):
        """'in line', 'two streams' or 'one kernel' (include/skyjo_vec.h: SKYJO_OPT_OVERLAP)."""
        v = C.c_int64()
        _lib.check(self._L.skyjo_vec_get_option(self._h, 2, C.byref(v)))
        return {0: "in line", 2: "two streams", 3: "one kernel"}[int(v.value)]


    def set_debug_option(self, option, value):
        """Fault injection (include/skyjo_vec.h SKYJO_OPT_DEBUG_*): 3 = spin limit (log2), 4 = dealing delay."""
        _lib.check(self._L.skyjo_vec_set_option(self._h, int(option), int(value)))

    def reset_counters(self):
        _lib.check(self._L.skyjo_vec_reset_counters(self._h, None))

    def get_state(self, game):
        """Canonical form of one game (include/skyjo_vec.h: skyjo_game_state).  Right after a host-style step / reset of a
        small batch this costs no device traffic (the call brought every game back with its records)."""
        buf = np.zeros(1, dtype=_lib.GAME_STATE_DTYPE)
        _lib.check(self._L.skyjo_vec_get_state(self._h, int(game), buf.ctypes.data_as(C.POINTER(_lib.GameState)), None))
        s = buf[0]
        N, nd, ns = self.num_players, int(s["n_draw"]), int(s["n_disc"])
        return dict(
            cards=s["players_cards"][:N].copy(), masked=s["players_masked"][:N].copy(),
            draw=s["drawpile"][:nd].copy(), n_draw=nd, disc=s["discard_pile"][:ns].copy(), n_disc=ns,
            hand=int(s["hand_card"]), player=int(s["expected_player"]), phase=int(s["expected_phase"]),
            is_terminated=bool(s["is_terminated"]), done=bool(s["done"]), status=int(s["status"]),
            episode_steps=int(s["episode_steps"]), episode=int(s["episode"]), reshuffles=int(s["reshuffles"]),
            num_refunded=s["num_refunded"][:N].copy(), num_placed=s["num_placed"][:N].copy(),
            final_score=s["final_score"][:N].copy(), rewards=s["rewards"][:N].copy())

    # ------------------------------------------------------------------ the caller's numpy stream (no_bank engines)
    def rng_set(self, game, key, pos):
        """Hand numpy's legacy stream - ``np.random.get_state()``: key uint32[624], pos - to one game (MT19937 mode, no_bank)."""
        key = np.ascontiguousarray(key, dtype=np.uint32)
        assert key.shape == (624,)
        _lib.check(self._L.skyjo_vec_rng_set_state(self._h, int(game), key.ctypes.data_as(C.c_void_p), int(pos), None))

    def rng_get(self, game):
        """The game's stream in numpy's terms: (key uint32[624], pos) for ``np.random.set_state``."""
        key = np.zeros(624, dtype=np.uint32)
        pos = C.c_int32()
        _lib.check(self._L.skyjo_vec_rng_get_state(self._h, int(game), key.ctypes.data_as(C.c_void_p), C.byref(pos), None))
        return key, int(pos.value)

    def set_state(self, game, cards, masked, draw, disc, hand=15, player=0, phase=0, num_refunded=None,
                  num_placed=None, episode=0):
        """Fixture injection / restore: overwrite one live game (the RNG streams are left untouched)."""
        s = _lib.GameState()
        N = self.num_players
        cards = np.asarray(cards, dtype=np.int8).reshape(N, 12)
        masked = np.asarray(masked, dtype=np.int8).reshape(N, 12)
        for p in range(N):
            for k in range(12):
                s.players_cards[p][k] = int(cards[p, k])
                s.players_masked[p][k] = int(masked[p, k])
        draw, disc = np.asarray(draw, dtype=np.int8).ravel(), np.asarray(disc, dtype=np.int8).ravel()
        for k, c in enumerate(draw):
            s.drawpile[k] = int(c)
        for k, c in enumerate(disc):
            s.discard_pile[k] = int(c)
        s.n_draw, s.n_disc = len(draw), len(disc)
        s.hand_card, s.expected_player, s.expected_phase = int(hand), int(player), int(phase)
        s.episode = int(episode)
        for p in range(N):
            s.num_refunded[p] = int(num_refunded[p]) if num_refunded is not None else 0
            s.num_placed[p] = int(num_placed[p]) if num_placed is not None else 0
        _lib.check(self._L.skyjo_vec_set_state(self._h, int(game), C.byref(s), None))

    def seed_raw(self, game, value):
        """np.random.seed(value) on one game's legacy stream, no deal (fixtures)."""
        _lib.check(self._L.skyjo_vec_seed_raw(self._h, int(game), int(value) & 0xFFFFFFFF, None))
        _lib.check(self._L.skyjo_dev_sync(None))
