"""Action-mask policy model: the second caller of the env hot path (config 5 of BASELINE.json).

Restates ``TorchActionMaskModel`` (rlskyjo/models/action_mask_model.py:13-77) without Ray: RLlib's
default fully connected net (two tanh layers of 256 units, separate value branch) on
``obs["observations"]``, and ``logits + clamp(log(action_mask), min=FLOAT_MIN)`` as in :58-74.  It
consumes the zero-copy views of the engine's record tensor directly on the GPU (``SkyjoVecEnv.split``),
so a PPO-style rollout never leaves the device.  ``ActionMaskModel`` / ``sample_actions`` are the plain-torch
statement of the model (float32) and the test reference.  Behind the C ABI the same model runs as hand-written
gfx950 kernels: ``FusedNet`` packs one branch for the matrix cores (``skyjo_vec_mlp_*``, csrc/skyjo_policy.hip: float32-grade
by default - bf16-pair operands, float32 accumulation, within 1e-4 of the float32 module - or plain bf16 operands as the
fast mode; tolerances in tests/test_gpu_policy_net.py), ``FusedNet.act`` adds the masking + categorical draw in the net's epilogue, and with
``value_net=`` the value branch rides in the same launch (``skyjo_vec_mlp_act_value``).
"""
import torch
from torch import nn

FLOAT_MIN = torch.finfo(torch.float32).min  # ray.rllib.utils.torch_utils.FLOAT_MIN


class ActionMaskModel(nn.Module):
    def __init__(self, obs_dim=31, num_outputs=26, hiddens=(256, 256), no_masking=False):
        super().__init__()
        self.no_masking = no_masking  # action_mask_model.py:53-56
        layers, d = [], obs_dim
        for h in hiddens:
            layers += [nn.Linear(d, h), nn.Tanh()]
            d = h
        self.policy = nn.Sequential(*layers, nn.Linear(d, num_outputs))
        vlayers, d = [], obs_dim
        for h in hiddens:
            vlayers += [nn.Linear(d, h), nn.Tanh()]
            d = h
        self.value = nn.Sequential(*vlayers, nn.Linear(d, 1))
        self._last_obs = None

    def forward(self, obs):
        """obs: {"observations": int8/float [B, D], "action_mask": int8/float [B, 26]} -> masked logits [B, 26]."""
        x = obs["observations"].to(torch.float32)
        self._last_obs = x
        logits = self.policy(x)
        if self.no_masking:
            return logits
        inf_mask = torch.clamp(torch.log(obs["action_mask"].to(torch.float32)), min=FLOAT_MIN)
        return logits + inf_mask

    def value_function(self):
        return self.value(self._last_obs).squeeze(-1)


@torch.no_grad()
def sample_actions(model, obs, generator=None):
    """Categorical sample from the masked logits -> int32 actions for SkyjoVecEnv.step."""
    probs = torch.softmax(model(obs), dim=-1)
    return torch.multinomial(probs, 1, generator=generator).squeeze(-1).to(torch.int32)


@torch.no_grad()
def sample_actions_fused(model, env, records, seed=0, ticket=0, logp=None):
    """The same draw with the masking, softmax and sampling fused into one HIP pass over the engine's records
    (``SkyjoVecEnv.sample_actions``): only the policy net's three GEMMs run in torch."""
    v = env.split(records)
    x = v.observations.to(torch.float32)
    model._last_obs = x
    logits = model.policy(x)
    return env.sample_actions(logits, records, seed=seed, ticket=ticket, no_masking=model.no_masking, logp=logp)


class FusedNet:
    """One branch of the model (``model.policy`` or ``model.value``: Linear-Tanh-Linear-Tanh-Linear with 256 hidden
    units) packed for the MI355X matrix cores (``skyjo_vec_mlp_*``, csrc/skyjo_policy.hip): weights as bf16 MFMA
    fragments (pairs of them in the float32-grade mode), float32 accumulation, observations read straight from the
    engine's records."""

    def __init__(self, seq, device=0, precision="fp32"):
        """``precision``: "fp32" (default; the reference's TorchFC is float32 - every operand a bf16 pair, outputs within 1e-4
        of ``seq`` itself) or "bf16" (single bf16 operands: a third of the matrix work, outputs within 8e-2)."""
        import ctypes as C

        import numpy as np

        from . import _lib
        lins = [m for m in seq if isinstance(m, nn.Linear)]
        assert len(lins) == 3 and lins[0].out_features == 256 and lins[1].in_features == 256 and lins[1].out_features == 256
        self.obs_dim, self.out_dim = lins[0].in_features, lins[2].out_features
        arrs = []
        for lin in lins:
            arrs += [np.ascontiguousarray(lin.weight.detach().float().cpu().numpy()),
                     np.ascontiguousarray(lin.bias.detach().float().cpu().numpy())]
        self._L = _lib.load()
        h = C.c_void_p()
        self.precision = precision
        _lib.check(self._L.skyjo_vec_mlp_create(int(device), self.obs_dim, self.out_dim,
                                                {"bf16": _lib.MLP_BF16, "fp32": _lib.MLP_FP32}[precision],
                                                *[a.ctypes.data_as(C.c_void_p) for a in arrs], C.byref(h)))
        self._h, self._C, self._check = h, C, _lib.check

    def __call__(self, records, out=None, planar=False):
        """Outputs float32 [n, out_dim] for the records' games.  ``planar``: ``records`` is tile-planar ([..., tiles, P, 64, 16], what
        ``SkyjoVecEnv.rollout`` writes with ``set_record_layout("tile-planar")``), read in place; n = 64 x the number of blocks
        (the rows beyond ``num_envs`` of a partial last tile are computed from whatever the block holds)."""
        rb = int(records.shape[-3] * 16) if planar else int(records.shape[-1])
        n = records.numel() // rb
        if out is not None:  # (planar blocks are whole tiles: the caller's output says how many games there are)
            n = min(n, out.numel() // self.out_dim)
        if out is None:
            out = torch.empty((n, self.out_dim), dtype=torch.float32, device=records.device)
        C = self._C
        from . import _lib
        self._check(self._L.skyjo_vec_mlp_forward_layout(self._h, C.c_void_p(records.data_ptr()), rb,
                                                         _lib.REC_TILE_PLANAR if planar else _lib.REC_ROW_MAJOR, n,
                                                         C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out

    def act(self, env, records, seed=0, ticket=0, no_masking=False, actions=None, logp=None, logits=None, value_net=None,
            values=None, planar=False):
        """Policy branch + masked categorical draw in one launch (``skyjo_vec_mlp_act``): int32 actions for ``env.step``.
        With ``value_net`` (the ``FusedNet`` of the value branch) and ``values`` (float32 [n, 1]) the value estimates of
        the same records are computed by the same launch (``skyjo_vec_mlp_act_value``).  ``planar``: ``records`` is ONE iteration's
        tile-planar block of ``env`` ([tiles, P, 64, 16]), read in place; n = env.num_envs."""
        from . import _lib
        n = env.num_envs if planar else records.numel() // records.shape[-1]
        if actions is None:
            actions = torch.empty((n,), dtype=torch.int32, device=records.device)
        C = self._C
        vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if value_net is not None:
            assert values is not None and values.numel() == n * value_net.out_dim and values.dtype == torch.float32
        self._check(self._L.skyjo_vec_mlp_act_value_layout(env._h, self._h, value_net._h if value_net is not None else None, vp(records),
                                                           _lib.REC_TILE_PLANAR if planar else _lib.REC_ROW_MAJOR, n, int(seed), int(ticket),
                                                           1 if no_masking else 0, vp(actions), vp(logp), vp(logits),
                                                           vp(values) if value_net is not None else None, stream))
        return actions

    def close(self):
        if self._h:
            self._L.skyjo_vec_mlp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
