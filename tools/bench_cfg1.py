"""BASELINE.json configs[0] as a drop-in cost figure: ONE three-player game driven from Python through the reference's own
caller loops - `simple_episode` (rlskyjo/environment/vanilla_env_example.py:6-41: env(**DEFAULT_CONFIG), agent_iter / last /
step) and `sample_run` (rlskyjo/game/sample_game.py:5-28: SkyjoGame core loop) with policy_ra on a seeded Generator.
Reference (BASELINE.md section 2, 8-core Xeon 2.1 GHz build container): 6.9 k env-steps/s through the env, 8.3 k through the
core loop.  Every step here is a kernel launch over one tile with one live lane plus the host round trip - the number says
what the drop-in costs per call, not what the GPU can do.   python tools/bench_cfg1.py [episodes] [--global-rng]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from skyjo_rl_amd import aec_env
from skyjo_rl_amd.game import SkyjoGame
from skyjo_rl_amd.policy import policy_ra

EPISODES = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
GLOBAL = "--global-rng" in sys.argv
kw = dict(global_rng=True) if GLOBAL else {}
out = {"episodes": EPISODES, "global_rng": GLOBAL}

e = aec_env.env(**aec_env.DEFAULT_CONFIG, **kw)
e.seed(42)
rng = None if GLOBAL else np.random.default_rng(42)
steps, t_render = 0, 0.0
e.reset()
for agent in e.agent_iter(max_iter=900):  # warm-up episode
    obs, reward, done, info = e.last()
    e.step(None if done else policy_ra(obs["observations"], obs["action_mask"], rng=rng))
t0 = time.perf_counter()
for ep in range(EPISODES):
    e.reset()
    for agent in e.agent_iter(max_iter=900):
        obs, reward, done, info = e.last()
        if done:
            e.step(None)
        else:
            e.step(policy_ra(obs["observations"], obs["action_mask"], rng=rng))
            steps += 1
dt = time.perf_counter() - t0
out["env_steps_per_s"] = steps / dt
out["env_us_per_step"] = 1e6 * dt / steps
t0 = time.perf_counter()
for _ in range(50):
    e.render()
out["render_us"] = 1e6 * (time.perf_counter() - t0) / 50

g = SkyjoGame(num_players=3, score_penalty=2.0, observe_other_player_indirect=True, **kw)
g.set_seed(42)
steps = 0
t0 = time.perf_counter()
for ep in range(EPISODES):
    g.reset()
    while not g.is_terminated:
        pid, _ = g.expected_action
        obs, mask = g.collect_observation(pid)
        g.act(pid, policy_ra(obs, mask, rng=rng))
        steps += 1
dt = time.perf_counter() - t0
out["core_steps_per_s"] = steps / dt
out["core_us_per_step"] = 1e6 * dt / steps
t0 = time.perf_counter()
for _ in range(200):
    g._state = None
    _ = g.players_cards
out["state_fetch_us"] = 1e6 * (time.perf_counter() - t0) / 200
# the floor of one host-style call: a step in which the game is left alone (ACTION_SKIP) - launch + stream synchronisation, no Python around it
eng = g._engine
if hasattr(eng, "step_one"):
    t0 = time.perf_counter()
    for _ in range(3000):
        eng.step_one(0, eng.ACTION_SKIP)
    out["native_call_floor_us"] = 1e6 * (time.perf_counter() - t0) / 3000
out["reference_constants"] = {"env_steps_per_s": 6.9e3, "core_steps_per_s": 8.3e3, "source": "BASELINE.md section 2"}
print(json.dumps(out))
