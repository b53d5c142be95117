"""GymEnvModel -- same constructor, parameter names and state_dict layout as the reference
(networks/neural_network.py:8-56), so checkpoints written by either side load in the other.

The module is only the PARAMETER CONTAINER.  Arithmetic never runs in torch: a call evaluates the
policy with the gfx950 kernel behind ses_policy_forward (batch of one), the population rollout uses the
fused kernel on the flat float32[P] form (`flat()` / `load_flat()`).
"""
import numpy as np
import torch
from torch import nn

from .abstracts import BaseNetwork

HIDDEN = 32


class GymEnvModel(BaseNetwork):
    def __init__(self, num_state=8, num_action=4, discrete_action=True, gru=True):
        super().__init__()
        self.num_state = num_state
        self.num_action = num_action
        self.discrete_action = discrete_action
        self.use_gru = gru
        # registration order == flat layout: fc1.{weight,bias} [gru.*_l0] fc2.{weight,bias}
        self.fc1 = nn.Linear(num_state, HIDDEN)
        if gru:
            self.gru = nn.GRU(HIDDEN, HIDDEN)
            self.h = torch.zeros([1, 1, HIDDEN], dtype=torch.float)
        self.fc2 = nn.Linear(HIDDEN, num_action)
        for p in self.parameters():
            p.requires_grad_(False)
        self._dev = None          # lazily created HipES handle for single-observation calls

    # -- reference API ------------------------------------------------------------------------
    def zero_init(self):
        for p in self.parameters():
            p.data = torch.zeros(p.shape)

    def reset(self):
        if self.use_gru:
            self.h = torch.zeros([1, 1, HIDDEN], dtype=torch.float)

    def get_param_list(self):
        # numpy VIEWS of the live tensors, like the reference: `param += noise` edits the module in place
        return [p.data.numpy() for p in self.parameters()]

    def apply_param(self, param_lst: list):
        for p, new in zip(self.parameters(), param_lst):
            p.data = torch.tensor(new).float()

    def forward(self, x):
        """x: ndarray (1, num_state) -> 0-d int64 ndarray (discrete) or float32[num_action] (continuous)."""
        from ses import HipES
        if self._dev is None:
            self._dev = HipES(None, self.num_state, self.num_action, self.discrete_action, self.use_gru)
        dev = self._dev
        obs = torch.from_numpy(np.asarray(x, dtype=np.float32).reshape(1, self.num_state)).to(dev.device)
        theta = torch.from_numpy(self.flat()[None, :]).to(dev.device)
        hidden = self.h.reshape(1, HIDDEN).to(dev.device).contiguous() if self.use_gru else None
        action, _logits, act = dev.policy_forward(theta, obs, hidden)
        if self.use_gru:
            self.h = hidden.cpu().reshape(1, 1, HIDDEN)
        if self.discrete_action:
            return np.array(int(action.item()), dtype=np.int64)
        return act[0].cpu().numpy()

    # -- flat form used by the device path ----------------------------------------------------
    def param_count(self):
        return sum(p.numel() for p in self.parameters())

    def flat(self):
        return np.concatenate([p.data.numpy().reshape(-1) for p in self.parameters()]).astype(np.float32)

    def load_flat(self, vec):
        vec = np.asarray(vec, dtype=np.float32).reshape(-1)
        assert vec.size == self.param_count(), (vec.size, self.param_count())
        off = 0
        for p in self.parameters():
            n = p.numel()
            p.data = torch.from_numpy(vec[off:off + n].copy()).reshape(p.shape)
            off += n
        return self
