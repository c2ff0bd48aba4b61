"""The two buffers around the environment loop (reference data_buffers.py:8-77, same names):
`Buffer` keeps the recent normalised history the policy is conditioned on, `ReplayBuffer` the
horizon-long (states, actions, next states) windows the dynamics model is refitted on.

Windows are cut with one strided view per trajectory instead of a Python loop per window, and the
replay buffer keeps whole window blocks (one per added trajectory) rather than single windows."""

import collections

import numpy as np


def sliding_windows(traj, length, count, start=0):
    """`count` windows traj[start+i : start+i+length], i = 0..count-1, as one array (a copy)."""
    traj = np.asarray(traj)
    if count <= 0:
        return np.empty((0, length) + traj.shape[1:], traj.dtype)
    view = np.lib.stride_tricks.sliding_window_view(traj, length, axis=0)   # [L-length+1, ..., length]
    view = np.moveaxis(view, -1, 1)                                          # [., length, ...]
    return np.array(view[start:start + count])          # always a writable copy


class Buffer:
    """The last `maxlen` + 1 normalised states and `maxlen` actions (reference :8-30)."""

    def __init__(self, maxlen, normalizer):
        self.normalizer = normalizer
        self.x_queue = collections.deque(maxlen=maxlen + 1)
        self.u_queue = collections.deque(maxlen=maxlen)

    def append_state(self, x, *args):
        self.x_queue.append(self.normalizer.normalize_state(x))

    def append_action(self, u, *args):
        self.u_queue.append(self.normalizer.normalize_action(u))

    def get_state_data(self):
        return np.array(self.x_queue)

    def get_action_data(self):
        return np.array(self.u_queue)

    def clear(self):
        for queue in (self.x_queue, self.u_queue):
            queue.clear()


class ReplayBuffer:
    """First-in-first-out store of at most `q_maxlen` windows (reference :33-77)."""

    _FIELDS = ("state", "action", "next_state")

    def __init__(self, horizon, q_maxlen, normalizer):
        self.horizon = horizon
        self.q_maxlen = q_maxlen
        self.normalizer = normalizer
        self.clear()

    def clear(self):
        self._blocks = {name: [] for name in self._FIELDS}     # arrays of windows, oldest first
        self._count = 0

    def __len__(self):
        return self._count

    def from_traj_to_seq(self, state_traj, action_traj):
        """(states, actions, next states) windows of one trajectory; a trajectory no longer than the
        horizon yields three empty arrays of shape (0,), like the reference's np.array([])."""
        state_traj, action_traj = np.asarray(state_traj), np.asarray(action_traj)
        count = len(state_traj) - self.horizon
        if count <= 0:
            return np.array([]), np.array([]), np.array([])
        H = self.horizon
        return (sliding_windows(state_traj, H, count), sliding_windows(action_traj, H, count),
                sliding_windows(state_traj, H, count, start=1))

    def add(self, state_traj, action_traj):
        windows = self.from_traj_to_seq(self.normalizer.normalize_state(state_traj),
                                        self.normalizer.normalize_action(action_traj))
        if len(windows[0]) == 0:
            return
        for name, block in zip(self._FIELDS, windows):
            self._blocks[name].append(block)
        self._count += len(windows[0])
        self._trim()

    def _trim(self):
        """Drop the oldest windows beyond q_maxlen (whole blocks first, then the head of a block)."""
        excess = self._count - self.q_maxlen
        while excess > 0:
            head = len(self._blocks["state"][0])
            if head <= excess:
                for name in self._FIELDS:
                    self._blocks[name].pop(0)
                excess -= head
                self._count -= head
            else:
                for name in self._FIELDS:
                    self._blocks[name][0] = self._blocks[name][0][excess:]
                self._count -= excess
                excess = 0

    def get_dataset(self):
        if self._count == 0:
            return np.array([]), np.array([]), np.array([])
        return tuple(np.concatenate(self._blocks[name], axis=0) for name in self._FIELDS)

    # the reference exposes its three deques; keep read access to the same names
    state_queue = property(lambda self: list(self.get_dataset()[0]))
    action_queue = property(lambda self: list(self.get_dataset()[1]))
    next_state_queue = property(lambda self: list(self.get_dataset()[2]))
