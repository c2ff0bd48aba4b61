"""Where goal_xseq / init_useq come from.  In the reference they are produced by a pretrained
behaviour-cloning sequence model (reference expert/expert_model.py:60-91, expert/nn.py) whose
parameters are not shipped; that model is the row before the hot path and is not
rebuilt here.  The policy only needs the protocol below."""

import numpy as np


class ExpertProtocol:
    def init(self, *args):
        return None

    def get_goal_states_init_actions(self, history_X, expert_params):
        """history_X (B, hist+1, n) -> goal (B, T+1, n) with goal[:, 0] = history_X[:, -1],
        init_U (B, T, m)   (reference policy/base.py:41-61)."""
        raise NotImplementedError


class HoldExpert(ExpertProtocol):
    """Goal = hold the current state, zero initial controls."""

    def __init__(self, horizon, u_size):
        self.T, self.m = int(horizon), int(u_size)

    def get_goal_states_init_actions(self, history_X, expert_params):
        x = np.asarray(history_X, np.float32)[:, -1]
        goal = np.repeat(x[:, None, :], self.T + 1, axis=1)
        return goal, np.zeros((x.shape[0], self.T, self.m), np.float32)


class TableExpert(ExpertProtocol):
    """Goal / initial controls looked up from arrays (synthetic workloads and tests): row i of the
    tables belongs to the sample whose history is history_X[i]."""

    def __init__(self, goal, init_U):
        self.goal = np.asarray(goal, np.float32)
        self.init_U = np.asarray(init_U, np.float32)
        self._cursor = None

    def select(self, idx):
        self._cursor = np.asarray(idx)
        return self

    def get_goal_states_init_actions(self, history_X, expert_params):
        idx = self._cursor if self._cursor is not None else np.arange(len(history_X))
        self._cursor = None
        return self.goal[idx], self.init_U[idx]
