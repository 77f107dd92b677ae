"""Config 5 of BASELINE.json: 65 536 parallel 4-player games stepped by the action-mask policy model
(rlskyjo/models/action_mask_model.py:13-77 restated without Ray, RLlib's default 256-256 tanh net, random weights)
end-to-end on one GPU.  Per lockstep iteration: records -> zero-copy views -> policy net (torch / hipBLASLt GEMMs)
-> masking + categorical draw -> skyjo_vec_step.  Two forms of the draw: plain torch (softmax + multinomial) and
the fused HIP pass (skyjo_vec_sample_actions); "mfma" runs the policy net as the hand-written MFMA kernel with the draw in
its epilogue (skyjo_vec_mlp_act) instead of torch, "mfma_value" the policy AND the value branch in that one launch
(skyjo_vec_mlp_act_value: what a PPO rollout needs per step).   python tools/bench_cfg5.py [B] [iters] [only-this-form]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skyjo_rl_amd import SkyjoVecEnv
from skyjo_rl_amd.action_mask_model import ActionMaskModel, FusedNet, sample_actions, sample_actions_fused

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 600
torch.manual_seed(0)
out = {"config": "65536 x 4 players, action-mask model in the loop" if B == 65536 else f"{B} x 4 players", "iters": ITERS}
for dtype in (torch.float32, torch.bfloat16):
    for form in ("torch", "fused", "mfma", "mfma_value"):
        if form.startswith("mfma") and dtype != torch.bfloat16:
            continue
        if len(sys.argv) > 3 and not form.startswith(sys.argv[3]):
            continue
        env = SkyjoVecEnv(B, num_players=4)
        env.seed(None, 3)
        model = ActionMaskModel(obs_dim=env.obs_dim).cuda()
        rec = env.reset()
        gen = torch.Generator(device="cuda").manual_seed(1)
        pol = FusedNet(model.policy) if form.startswith("mfma") else None
        val = FusedNet(model.value) if form == "mfma_value" else None
        act_buf = torch.empty((B,), dtype=torch.int32, device="cuda")
        val_buf = torch.empty((B, 1), dtype=torch.float32, device="cuda")

        def one(t, rec):
            if form == "mfma":  # policy net on the matrix cores with the draw in its epilogue (skyjo_vec_mlp_act) + env step: 2 launches
                return env.step(pol.act(env, rec, seed=9, ticket=t, actions=act_buf), out=rec)
            if form == "mfma_value":  # the same launch also evaluates the value branch
                return env.step(pol.act(env, rec, seed=9, ticket=t, actions=act_buf, value_net=val, values=val_buf), out=rec)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
                if form == "torch":
                    v = env.split(rec)
                    a = sample_actions(model, {"observations": v.observations, "action_mask": v.action_mask}, gen)
                else:
                    v = env.split(rec)
                    logits = model.policy(v.observations.to(torch.float32)).float().contiguous()
                    a = env.sample_actions(logits, rec, seed=9, ticket=t)
            return env.step(a, out=rec)

        for t in range(50):
            rec = one(t, rec)
        torch.cuda.synchronize()
        c0 = env.counters()
        t0 = time.perf_counter()
        for t in range(ITERS):
            rec = one(50 + t, rec)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c1 = env.counters()
        key = f"{form}_{'bf16' if dtype == torch.bfloat16 else 'fp32'}"
        out[key] = {"env_steps_per_s": (c1["steps"] - c0["steps"]) / dt, "us_per_iteration": 1e6 * dt / ITERS,
                    "illegal": c1["illegal"]}
        env.close()
print(json.dumps(out))
