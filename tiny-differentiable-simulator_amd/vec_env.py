"""Python mirror of the reference's vectorised environments
(``pytinydiffsim.VectorizedAntEnv`` / ``VectorizedLaikagoEnv``: python/pytinydiffsim_includes.h:58-220,
bound in python/pytinydiffsim.inl:1152-1191; driven by python/examples/vec_ant.py).

Same surface — ``reset()``, ``step(actions)`` returning obs / rewards / dones /
visual_world_transforms, ``action_dim()`` / ``obs_dim()`` — but everything stays on the GPU:
state, PD control, physics, reward / termination and auto-reset run inside the HIP step kernel
(``tds_hip_step_obs``), and the outputs are torch tensors (zero-copy views of device memory)
instead of nested Python lists of floats.
"""
from __future__ import annotations

from . import model as _model
from .hip_backend import HipSim


class VectorizedEnvOutput:
    """obs [N, obs_dim], rewards [N], dones [N] (1.0 / 0.0 like the reference's float list),
    visual_world_transforms [N, output_dim] (the step's full y record: q | qd | 7 per visual | up)."""

    __slots__ = ("obs", "rewards", "dones", "visual_world_transforms")

    def __init__(self, obs, rewards, dones, visual_world_transforms):
        self.obs = obs
        self.rewards = rewards
        self.dones = dones
        self.visual_world_transforms = visual_world_transforms

    def dlpack(self) -> dict:
        """The four outputs as DLPack capsules (zero-copy hand-over to any DLPack consumer:
        ``torch.from_dlpack``, ``cupy.from_dlpack``, ``jax.dlpack.from_dlpack`` ...)."""
        from torch.utils.dlpack import to_dlpack

        return {k: to_dlpack(getattr(self, k)) for k in self.__slots__}

    def as_lists(self) -> dict:
        """The reference binding's own return types — nested Python lists of floats
        (python/pytinydiffsim_includes.h:51-56: obs / visual_world_transforms list[list[float]], rewards / dones
        list[float]).  Copies to the host: for drop-in scripts, not for throughput."""
        return {k: getattr(self, k).float().cpu().tolist() for k in self.__slots__}


class VectorizedEnv:
    def __init__(self, model_name: str, num_envs: int, auto_reset_when_done: bool = True, device: int = 0,
                 seed: int = 0, dtype: str = "f64", kp=None, kd=None, max_force=None):
        import torch

        m = _model.load_model(model_name)
        if m.step_mode != _model.TDS_STEP_LOCOMOTION:
            raise ValueError("VectorizedEnv mirrors the locomotion environments (Ant, Laikago)")
        defaults = {"ant": (15.0, 0.3, 3.0), "laikago": (100.0, 2.0, 50.0)}
        base = "ant" if model_name.startswith("ant") else "laikago"
        dkp, dkd, dmf = defaults[base]  # ant_environment2.h:66-68, laikago_environment2.h:43-45
        self.kp, self.kd, self.max_force = (kp if kp is not None else dkp, kd if kd is not None else dkd,
                                            max_force if max_force is not None else dmf)
        self.sim = HipSim(m, num_envs, device=device, dtype=dtype)
        self.model = self.sim.model
        self.num_envs = num_envs
        self.auto_reset = bool(auto_reset_when_done)
        self.sim.set_auto_reset(self.auto_reset, seed)
        self._obs_rec = torch.zeros((num_envs, self.sim.obs_dim + 2), dtype=self.sim.torch_dtype,
                                    device=f"cuda:{device}")
        # kp, kd, max_force live in the last three slots of every x record
        # (prepare_sim_state_with_action_and_variables, locomotion_contact_simulation.h:138-148)
        self.sim.x[:, -3] = self.kp
        self.sim.x[:, -2] = self.kd
        self.sim.x[:, -1] = self.max_force

    def action_dim(self) -> int:
        return self.model.action_dim

    def obs_dim(self) -> int:
        return self.sim.obs_dim

    def set_state(self, state):
        """Overwrite [q | qd] of every environment ([N, obs_dim]; not part of the reference binding, whose only way
        to a state is reset(): used to start from given states, e.g. in the parity tests)."""
        import torch

        st = torch.as_tensor(state, dtype=self.sim.torch_dtype, device=self._obs_rec.device)
        self.sim.x[:, : self.sim.obs_dim] = st

    def reset(self):
        """All environments: reset distribution + settle steps on device; returns obs [N, obs_dim]."""
        self.sim.reset(None, self._obs_rec)
        return self._obs_rec[:, : self.sim.obs_dim]

    def step(self, actions) -> VectorizedEnvOutput:
        """actions: [N, action_dim] device tensor (or anything torch.as_tensor accepts)."""
        import torch

        a = torch.as_tensor(actions, dtype=self.sim.torch_dtype, device=self._obs_rec.device).contiguous()
        self.sim.step(a, 1, self._obs_rec)
        od = self.sim.obs_dim
        return VectorizedEnvOutput(self._obs_rec[:, :od], self._obs_rec[:, od], self._obs_rec[:, od + 1], self.sim.y)

    def step_many(self, action_blocks, n_steps: int, first_block: int = 0) -> VectorizedEnvOutput:
        """``n_steps`` calls of step() in one host call (tds_hip_step_many): step k takes
        ``action_blocks[(first_block + k) % len(action_blocks)]`` ([B, N, action_dim] device tensor); with
        auto_reset_when_done every step resets the environments it ends with done.  Returns the last step's output."""
        import torch

        a = torch.as_tensor(action_blocks, dtype=self.sim.torch_dtype, device=self._obs_rec.device).contiguous()
        self.sim.step_many(a, int(n_steps), self._obs_rec, first_block=first_block)
        od = self.sim.obs_dim
        return VectorizedEnvOutput(self._obs_rec[:, :od], self._obs_rec[:, od], self._obs_rec[:, od + 1], self.sim.y)



def VectorizedAntEnv(num_envs: int, auto_reset_when_done: bool = True, **kw) -> VectorizedEnv:
    return VectorizedEnv("ant", num_envs, auto_reset_when_done, **kw)


def VectorizedLaikagoEnv(num_envs: int, auto_reset_when_done: bool = True, **kw) -> VectorizedEnv:
    return VectorizedEnv("laikago", num_envs, auto_reset_when_done, **kw)
