"""The tile-planar record layout of the fused rollout (include/skyjo_vec.h: SKYJO_OPT_RECORD_LAYOUT): the same
records, cut into 16-byte pieces so that the step wavefront stores them straight from its registers.  Two engines with the same
seeds, one per layout: every byte of every record equal (bit-equality of the unpacked records), through the torch view and through
skyjo_vec_unpack_tiles; the row-major engine itself is pinned to the oracle by tests/test_gpu_full_batch.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    # name                     B      N  cycles per launch, launches, indirect observation
    ("headline_eight_cycles", 65536, 3, 8, 2, True),
    ("cfg2_eight_cycles", 4096, 2, 8, 3, True),
    ("cfg4_shard_S2", 32768, 3, 4, 2, True),
    ("N4_S3", 49152, 4, 2, 2, True),
    ("partial_tile_partial_wg", 40010, 2, 4, 2, True),
    ("single_cycle_calls", 16400, 3, 1, 5, True),
    # the direct observation's wider records (80 / 96 / 112 bytes: 5 / 6 / 7 pieces); at 65 536 x 3 the planar engine has LDS for the
    # deferred scoring again, the row-major one scores on the spot - same records
    ("direct_obs_headline_size", 65536, 3, 8, 2, False),
    ("direct_obs_N2_partial", 40010, 2, 4, 2, False),
    ("direct_obs_N4_S2", 32768, 4, 4, 2, False),
]


@pytest.mark.parametrize("name,B,N,cycles,launches,indirect", CASES, ids=[c[0] for c in CASES])
def test_tile_planar_records_equal_row_major_records(name, B, N, cycles, launches, indirect):
    import torch
    from skyjo_rl_amd import SkyjoVecEnv

    cfg = dict(num_players=N, auto_reset=True, observe_other_player_indirect=indirect)
    a, b = SkyjoVecEnv(B, **cfg), SkyjoVecEnv(B, **cfg)
    assert a.dealing_form() == "one kernel"
    b.set_record_layout("tile-planar")
    for e in (a, b):
        e.set_deal_interval(e.deal_interval())
        e.seed(None, 11)
    K = cycles * a.deal_interval()
    ra, rb = a.new_records(K), b.new_planar_records(K)
    rb.fill_(0xEE)
    for r in range(launches):
        a.rollout(K, policy_seed=3, records=ra)
        b.rollout(K, policy_seed=3, records=rb)
        rows = b.rows_from_planar(rb)
        assert rows.shape == ra.shape
        used = a.mask_offset + 32  # (the direct observation's records end in a few bytes of padding that no layout defines)
        assert torch.equal(rows[..., :used], ra[..., :used]), f"{name}: launch {r}: {(rows[..., :used] != ra[..., :used]).sum().item()} bytes differ"
        # the dense arrays straight from the planar blocks == the row-major engine's
        obs_t, mask_t = b.unpack_tiles(rb)
        obs_r, mask_r = a.unpack(ra)
        G = b.tiles * 64
        assert torch.equal(obs_t.view(K, G, -1)[:, :B], obs_r.view(K, B, -1))
        assert torch.equal(mask_t.view(K, G, 26)[:, :B], mask_r.view(K, B, 26))
    if B % 64:  # the slots of the partial last tile beyond num_envs are never written
        pad = rb.permute(0, 1, 3, 2, 4).reshape(K, b.tiles * 64, b.record_bytes)[:, B:]
        assert bool((pad == 0xEE).all())
    ca, cb = a.counters(), b.counters()
    for k in ("steps", "episodes", "resets", "sum_len", "waits"):
        assert ca[k] == cb[k], (k, ca[k], cb[k])
    assert ca["episodes"] > 0
    a.close(), b.close()


def test_the_layout_option_is_refused_where_the_kernel_does_not_exist():
    from skyjo_rl_amd import SkyjoNativeError, SkyjoVecEnv

    e = SkyjoVecEnv(256, num_players=5)
    with pytest.raises(SkyjoNativeError):
        e.set_record_layout("tile-planar")  # generic player count: no one-kernel form
    e.close()
    e = SkyjoVecEnv(256, num_players=3)
    e.set_record_layout("tile-planar")
    e.set_overlap(0)  # dealing in line: the step kernel of that form writes row-major records only
    e.seed(None, 0)
    with pytest.raises(SkyjoNativeError):
        e.rollout(8, records=e.new_planar_records(8))
    e.set_record_layout("row-major")
    e.rollout(8, records=e.new_records(8))
    e.close()


@pytest.mark.parametrize("B,N,precision", [(32768, 4, "bf16"), (40010, 3, "fp32"), (4096, 2, "fp32")])
def test_the_consumers_read_tile_planar_records_in_place(B, N, precision):
    """the policy / value net (its observation bytes and, for the draw in its epilogue, the mask words), the masked
    draw on given logits and the episode-end columns take the tile-planar records of a fused rollout as they lie
    (skyjo_vec_*_layout, SKYJO_REC_TILE_PLANAR) - the same bits as from the row-major copy of the same records, no unpack pass."""
    import torch
    from skyjo_rl_amd import SkyjoVecEnv
    from skyjo_rl_amd.action_mask_model import ActionMaskModel, FusedNet

    torch.manual_seed(5)
    env = SkyjoVecEnv(B, num_players=N, auto_reset=True)
    assert env.dealing_form() == "one kernel"
    env.set_record_layout("tile-planar")
    assert env.record_layout == "tile-planar"
    env.seed(None, 23)
    K = env.deal_interval()
    rp = env.new_planar_records(K)
    rp.zero_()
    model = ActionMaskModel(obs_dim=env.obs_dim).cuda()
    pol, val = FusedNet(model.policy, precision=precision), FusedNet(model.value, precision=precision)
    ends = 0
    for launch in range(3):
        env.rollout(K, policy_seed=9, records=rp)
        rows = env.rows_from_planar(rp).contiguous()
        for t in (0, K // 2, K - 1):
            out = {}
            for planar, rec in ((False, rows[t]), (True, rp[t])):
                a = torch.empty(B, dtype=torch.int32, device="cuda")
                lp, lg, v = torch.empty(B, device="cuda"), torch.empty((B, 26), device="cuda"), torch.empty((B, 1), device="cuda")
                pol.act(env, rec, seed=4, ticket=7 * launch + t, actions=a, logp=lp, logits=lg, value_net=val, values=v, planar=planar)
                a1 = pol.act(env, rec, seed=4, ticket=7 * launch + t, planar=planar)  # the policy branch alone
                u = torch.empty(B, device="cuda")
                a2 = env.sample_actions(lg, rec, seed=4, ticket=7 * launch + t, uniform=u, planar=planar)
                fr, ee = env.episode_ends(rec, planar=planar)
                fwd = val(rec, planar=planar)
                out[planar] = (a, lp, lg, v, a1, a2, u, fr, ee, fwd[:B])
            for x, y in zip(out[False], out[True]):
                assert torch.equal(x, y)
            a, _, _, v, a1, a2, _, _, ee, fwd = out[True]
            assert torch.equal(a, a1) and torch.equal(a, a2) and torch.equal(v, fwd)
            ends += int(ee.sum())
    assert ends > 0  # (some of the sampled iterations ended episodes: the columns are not trivially equal)
    pol.close(), val.close(), env.close()


@pytest.mark.parametrize("B,N,precision", [(4106, 3, "bf16"), (8192, 4, "fp32")])
def test_a_model_rollout_that_never_holds_a_row_major_record(B, N, precision):
    """SKYJO_REC_TILE_PLANAR_ALL: reset / observe / step_collect / skyjo_vec_model_rollout write tile-planar records too (the step kernel
    from the registers the record was assembled in), the nets read them in place: every column of the rollout buffer - actions,
    log-probabilities, values, episode ends, final rewards - and every record equal the row-major engine's, bit for bit; so does the
    same loop made one launch at a time."""
    import torch
    from skyjo_rl_amd import SkyjoVecEnv
    from skyjo_rl_amd.action_mask_model import ActionMaskModel, FusedNet
    from skyjo_rl_amd.rollout import RolloutBuffer, collect, collect_stepwise

    torch.manual_seed(9)
    T = 48
    model = ActionMaskModel(obs_dim=31).cuda()
    pol, val = FusedNet(model.policy, precision=precision), FusedNet(model.value, precision=precision)
    bufs = []
    for layout, stepwise in (("row-major", False), ("tile-planar-all", False), ("tile-planar-all", True)):
        env = SkyjoVecEnv(B, num_players=N, auto_reset=True)
        env.set_record_layout(layout)
        assert env.record_layout == layout
        env.seed(None, 77)
        first = env.reset()
        assert tuple(first.shape) == ((env.tiles, env.record_bytes // 16, 64, 16) if layout != "row-major" else (B, env.record_bytes))
        buf = RolloutBuffer(env, T)
        for r in range(3):
            (collect_stepwise if stepwise else collect)(env, pol, val, buf, seed=3, first_ticket=r * T, first_records=first if r == 0 else buf.records[T].clone())
        rows = env.rows_from_planar(buf.records) if buf.planar else buf.records
        bufs.append((rows.clone(), buf.actions.clone(), buf.logp.clone(), buf.values.clone(), buf.episode_end.clone(), buf.final_rewards.clone(),
                     env.observe() if not buf.planar else env.rows_from_planar(env.observe()), env.counters()))
        env.close()
    ref = bufs[0]
    assert int(ref[4].sum()) > 0 and ref[7]["episodes"] > 0
    for other in bufs[1:]:
        for k in range(7):
            assert torch.equal(ref[k], other[k]), k
        for k in ("steps", "episodes", "resets", "sum_len", "illegal"):
            assert ref[7][k] == other[7][k]
    pol.close(), val.close()
