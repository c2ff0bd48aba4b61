"""History buffer for acting in an environment and the replay buffer of (x, u, x') windows
(reference data_buffers.py:8-77, same names).  Windows are cut with one strided view per trajectory
instead of a Python loop per element."""

import collections

import numpy as np


def sliding_windows(traj, length, count, start=0):
    """`count` windows traj[start+i : start+i+length], i = 0..count-1, as one array (a copy)."""
    traj = np.asarray(traj)
    if count <= 0:
        return np.empty((0, length) + traj.shape[1:], traj.dtype)
    view = np.lib.stride_tricks.sliding_window_view(traj, length, axis=0)   # [L-length+1, ..., length]
    view = np.moveaxis(view, -1, 1)                                          # [., length, ...]
    return np.ascontiguousarray(view[start:start + count])


class Buffer:
    """Last `maxlen`+1 normalised states and `maxlen` actions (reference data_buffers.py:8-30)."""

    def __init__(self, maxlen, normalizer):
        self.x_queue = collections.deque(maxlen=maxlen + 1)
        self.u_queue = collections.deque(maxlen=maxlen)
        self.normalizer = normalizer

    def append_state(self, x, *args):
        self.x_queue.append(self.normalizer.normalize_state(x))

    def append_action(self, u, *args):
        self.u_queue.append(self.normalizer.normalize_action(u))

    def get_state_data(self):
        return np.array(self.x_queue)

    def get_action_data(self):
        return np.array(self.u_queue)

    def clear(self):
        self.x_queue.clear()
        self.u_queue.clear()


class ReplayBuffer:
    """FIFO of horizon-long (states, actions, next states) windows (reference :33-77)."""

    def __init__(self, horizon, q_maxlen, normalizer):
        self.horizon = horizon
        self.state_queue = collections.deque(maxlen=q_maxlen)
        self.action_queue = collections.deque(maxlen=q_maxlen)
        self.next_state_queue = collections.deque(maxlen=q_maxlen)
        self.normalizer = normalizer

    def clear(self):
        self.state_queue.clear()
        self.action_queue.clear()
        self.next_state_queue.clear()

    def from_traj_to_seq(self, state_traj, action_traj):
        H = self.horizon
        count = len(state_traj) - H
        state_traj, action_traj = np.asarray(state_traj), np.asarray(action_traj)
        if count <= 0:       # np.array([]) in the reference: keep its (0,) shape
            return np.array([]), np.array([]), np.array([])
        return (sliding_windows(state_traj, H, count), sliding_windows(action_traj, H, count),
                sliding_windows(state_traj, H, count, start=1))

    def add(self, state_traj, action_traj):
        state_traj = self.normalizer.normalize_state(state_traj)
        action_traj = self.normalizer.normalize_action(action_traj)
        xs, us, ys = self.from_traj_to_seq(state_traj, action_traj)
        self.state_queue.extend(xs)
        self.action_queue.extend(us)
        self.next_state_queue.extend(ys)

    def get_dataset(self):
        return (np.array(self.state_queue), np.array(self.action_queue),
                np.array(self.next_state_queue))
