"""Host-side mirror of mjpc::spline::TimeSpline (mjpc/spline/spline.{h,cc}).

The planners keep `policy` / `previous_policy` as host splines so that
`ActionFromPolicy` never waits on the GPU (threading
contract); only candidate evaluation runs on the device."""
from __future__ import annotations

import bisect

import numpy as np

ZERO, LINEAR, CUBIC = 0, 1, 2  # SplineInterpolation, spline.h:29-33


class TimeSpline:
    def __init__(self, dim=0, interpolation=ZERO, initial_capacity=1):
        self._dim = int(dim)
        self._interp = interpolation
        self._times: list[float] = []
        self._values: list[np.ndarray] = []  # time-ordered node values (views are handed out)

    # ---- container
    def size(self):
        return len(self._times)

    def dim(self):
        return self._dim

    def reserve(self, num_nodes):  # capacity is an implementation detail of the ring buffer
        return None

    def set_interpolation(self, interpolation):
        self._interp = interpolation

    def interpolation(self):
        return self._interp

    def node_at(self, index):
        return self._times[index], self._values[index]

    def times(self):
        return np.array(self._times, dtype=np.float64)

    def values(self):
        if not self._values:
            return np.zeros((0, self._dim))
        return np.stack(self._values)

    def clear(self):
        self._times, self._values = [], []

    def copy(self):
        s = TimeSpline(self._dim, self._interp)
        s._times = list(self._times)
        s._values = [v.copy() for v in self._values]
        return s

    # ---- spline.cc:213-249
    def add_node(self, time, values=None):
        time = float(time)
        if self._times and not (time > self._times[-1] or time < self._times[0]):
            raise ValueError("Adding nodes to the middle of the spline isn't supported.")
        v = np.zeros(self._dim) if values is None else np.array(values, dtype=np.float64).reshape(-1)
        if v.size != self._dim:
            raise ValueError(f"expected {self._dim} values, got {v.size}")
        if not self._times or time > self._times[-1]:
            self._times.append(time)
            self._values.append(v)
        else:
            self._times.insert(0, time)
            self._values.insert(0, v)
        return v

    # ---- spline.cc:269-287
    def _slope(self, node, k):
        t, v = self._times, self._values
        if node == 0:
            return (v[1][k] - v[0][k]) / (t[1] - t[0])
        if node == len(t) - 1:
            return (v[node][k] - v[node - 1][k]) / (t[node] - t[node - 1])
        return (0.5 * (v[node + 1][k] - v[node][k]) / (t[node + 1] - t[node])
                + 0.5 * (v[node][k] - v[node - 1][k]) / (t[node] - t[node - 1]))

    # ---- spline.cc:103-156
    def sample(self, time, out=None):
        out = np.zeros(self._dim) if out is None else out
        if not self._times:
            out[:] = 0.0
            return out
        up = bisect.bisect_right(self._times, time)  # std::upper_bound
        if up == len(self._times):
            out[:] = self._values[up - 1]
            return out
        if up == 0:
            out[:] = self._values[0]
            return out
        lo = up - 1
        tl, tu = self._times[lo], self._times[up]
        t = (time - tl) / (tu - tl)
        vl, vu = self._values[lo], self._values[up]
        if self._interp == ZERO:
            out[:] = vl
        elif self._interp == LINEAR:
            out[:] = vl * (1 - t) + vu * t
        elif self._interp == CUBIC:
            c0 = 2.0 * t * t * t - 3.0 * t * t + 1.0
            c1 = (t * t * t - 2.0 * t * t + t) * (tu - tl)
            c2 = -2.0 * t * t * t + 3 * t * t
            c3 = (t * t * t - t * t) * (tu - tl)
            for i in range(self._dim):
                out[i] = c0 * vl[i] + c1 * self._slope(lo, i) + c2 * vu[i] + c3 * self._slope(up, i)
        else:
            raise ValueError(f"Unknown interpolation: {self._interp}")
        return out

    # ---- spline.cc:164-187
    def discard_before(self, time):
        last = bisect.bisect_right(self._times, time)
        if last == 0:
            return 0
        keep = 1 if self._interp == CUBIC else 0
        last -= 1
        while last != 0 and keep:
            last -= 1
            keep -= 1
        del self._times[:last]
        del self._values[:last]
        return last

    # ---- spline.cc:189-197
    def shift_time(self, start_time):
        if not self._times:
            return
        shift = start_time - self._times[0]
        self._times = [t + shift for t in self._times]
