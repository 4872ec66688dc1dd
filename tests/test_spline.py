"""Known-answer tests of TimeSpline, ported from the reference's own suite
mjpc/test/spline/spline_test.cc (line numbers cited per test). Each test runs against
BOTH restatements: the CPU oracle (oracle/spline.c) and the product's host-side mirror
(mujoco_mpc_amd/spline.py) -- this pins the oracle to the reference's golden values."""
import numpy as np
import pytest

from mujoco_mpc_amd import spline as hostspline
from oracle import pyoracle

ZERO, LINEAR, CUBIC = 0, 1, 2


class OracleAdapter:
    def __init__(self, dim, interp=ZERO):
        self.s = pyoracle.Spline(dim, interp)

    def __getattr__(self, k):
        return getattr(self.s, k)

    def node_values_at(self, i):
        return self.s.node_values()[i]

    def set_node(self, i, values):
        self.s.set_node_values(i, values)


class HostAdapter:
    def __init__(self, dim, interp=ZERO):
        self.s = hostspline.TimeSpline(dim, interp)

    def __getattr__(self, k):
        return getattr(self.s, k)

    def set_node(self, i, values):
        self.s.node_at(i)[1][:] = values

    def copy(self):
        h = HostAdapter(self.s.dim())
        h.s = self.s.copy()
        return h


@pytest.fixture(params=["oracle", "host"])
def make(request):
    return OracleAdapter if request.param == "oracle" else HostAdapter


ALL = [ZERO, LINEAR, CUBIC]


def test_empty(make):  # spline_test.cc:41-50
    s = make(10)
    assert s.size() == 0
    assert np.all(s.sample(2.0) == 0.0)


@pytest.mark.parametrize("interp", ALL)
def test_one_node(make, interp):  # :52-64
    s = make(2, interp)
    s.add_node(1.0, [1.0, 2.0])
    assert s.size() == 1
    for t in (0.0, 2.0, 4.0):
        assert list(s.sample(t)) == [1.0, 2.0]


@pytest.mark.parametrize("interp", ALL)
def test_two_nodes(make, interp):  # :66-83
    s = make(2, interp)
    s.add_node(1.0, [1.0, 2.0])
    i = s.add_node(2.0)
    s.set_node(1, [3.0, 4.0])
    assert s.size() == 2
    assert list(s.sample(0)) == [1.0, 2.0]
    assert list(s.sample(1)) == [1.0, 2.0]
    assert list(s.sample(2)) == [3.0, 4.0]
    assert list(s.sample(3)) == [3.0, 4.0]


def test_add_node_before_start(make):  # :85-100
    s = make(2)
    s.add_node(2.0, [2.0, 3.0])
    s.add_node(1.0, [1.0, 2.0])
    s.add_node(3.0, [3.0, 4.0])
    s.add_node(0.0, [0.0, 1.0])
    for t in range(4):
        assert list(s.sample(t)) == [float(t), float(t + 1)]


def test_add_node_in_the_middle_is_rejected(make):  # spline.cc:217-219 CHECK
    s = make(1)
    s.add_node(0.0, [0.0])
    s.add_node(2.0, [2.0])
    with pytest.raises(ValueError):
        s.add_node(1.0, [1.0])


def test_add_node_resets_to_zero(make):  # :102-118
    s = make(1)
    for i in range(6):
        s.add_node(2.0 * i, [1.0])
    s.clear()
    s.add_node(1.0)
    assert list(s.sample(0)) == [0.0]


def test_zero_order(make):  # :120-127
    s = make(2, ZERO)
    s.add_node(1.0, [1.0, 2.0])
    s.add_node(2.0, [3.0, 4.0])
    assert list(s.sample(1.5)) == [1.0, 2.0]


def test_linear(make):  # :129-136
    s = make(2, LINEAR)
    s.add_node(1.0, [1.0, 2.0])
    s.add_node(2.0, [3.0, 4.0])
    assert list(s.sample(1.5)) == [2.0, 3.0]


def test_cubic(make):  # :138-163
    s = make(2, CUBIC)
    s.add_node(1.0, [1.0, 2.0])
    s.add_node(2.0, [3.0, 4.0])
    assert list(s.sample(1.5)) == [2.0, 3.0]
    s.clear()
    for t, v in ((0.0, [1.0, 2.0]), (1.0, [1.0, 2.0]), (2.0, [3.0, 4.0]), (3.0, [3.0, 4.0])):
        s.add_node(t, v)
    assert list(s.sample(1.5)) == [2.0, 3.0]
    s = make(1, CUBIC)
    s.add_node(-1.0, [1.0])
    s.add_node(0.0, [0.0])
    s.add_node(1.0, [1.0])
    for x in np.arange(0.0, 1.0001, 0.125):
        assert s.sample(x)[0] == -x ** 3 + 2 * x ** 2  # exact, as in the reference (ElementsAre)


def test_shift_time(make):  # :165-185
    s = make(2, LINEAR)
    for k in range(1, 5):
        s.add_node(float(k), [float(k), float(k + 1)])
    assert list(s.sample(1.0)) == [1.0, 2.0]
    assert list(s.sample(1.5)) == [1.5, 2.5]
    s.shift_time(1.5)
    assert s.size() == 4
    assert list(s.sample(1.5)) == [1.0, 2.0]
    assert list(s.sample(2.0)) == [1.5, 2.5]


@pytest.mark.parametrize("interp", ALL)
def test_discard_before(make, interp):  # :187-231
    s = make(2, interp)
    for k in range(1, 5):
        s.add_node(float(k), [float(k), float(k + 1)])
    assert s.discard_before(0.9) == 0
    assert s.size() == 4
    assert list(s.sample(0.0)) == [1.0, 2.0]
    n = s.discard_before(3.0)
    if interp == CUBIC:
        assert n == 1 and s.size() == 3 and list(s.sample(1.0)) == [2.0, 3.0]
    else:
        assert n == 2 and s.size() == 2 and list(s.sample(1.0)) == [3.0, 4.0]
    assert s.discard_before(3.9) == 0
    if interp == CUBIC:
        assert s.size() == 3 and list(s.sample(1.0)) == [2.0, 3.0]
    else:
        assert s.size() == 2 and list(s.sample(1.0)) == [3.0, 4.0]


def test_discard_before_ring_loop(make):  # :233-257
    s = make(1)
    for k in range(1, 5):
        s.add_node(float(k), [float(k)])
    assert s.discard_before(3) == 2
    s.add_node(5.0, [5.0])
    s.add_node(6.0, [6.0])
    assert s.discard_before(6.0) == 3
    assert s.size() == 1
    assert s.sample(1.0)[0] == 6.0


def test_reserve_after_add(make):  # :259-274
    s = make(2, LINEAR)
    s.add_node(1.0, [1.0, 2.0])
    s.add_node(2.0, [2.0, 3.0])
    s.add_node(3.0, [4.0, 5.0])
    assert s.size() == 3
    assert list(s.sample(2.5)) == [3.0, 4.0]


def test_copy_is_deep(make):  # :313-366 (CopyConstructor / CopyAssignment)
    s = make(2, LINEAR)
    s.add_node(1.0, [1.0, 2.0])
    s.add_node(2.0, [2.0, 3.0])
    s.discard_before(2.0)
    s.add_node(3.0, [3.0, 4.0])
    s.add_node(4.0, [4.0, 5.0])
    s.add_node(5.0, [5.0, 6.0])
    s2 = s.copy()
    assert s2.size() == 4
    for i in range(s.size()):
        s.set_node(i, [3.0, 4.0])
    s.clear()
    assert list(s2.sample(1.5)) == [2.0, 3.0]
    assert list(s2.sample(2.5)) == [2.5, 3.5]


def test_clear(make):  # :368-383
    s = make(2)
    s.add_node(1.0, [1.0, 2.0])
    assert list(s.sample(0)) == [1.0, 2.0]
    s.clear()
    assert s.size() == 0
    assert list(s.sample(0)) == [0.0, 0.0]
    s.add_node(1.0, [3.0, 4.0])
    assert list(s.sample(1)) == [3.0, 4.0]


def test_dim0(make):  # :385-398
    s = make(0, ZERO)
    s.add_node(1.0, [])
    s.add_node(2.0, [])
    assert s.size() == 2
    assert s.discard_before(2.0) == 1
    assert s.size() == 1
    assert s.sample(1).size == 0


def test_oracle_equals_host_on_random_splines():
    rng = np.random.default_rng(3)
    for interp in ALL:
        for P in (1, 2, 3, 7):
            o, h = pyoracle.Spline(3, interp), hostspline.TimeSpline(3, interp)
            t = np.cumsum(rng.uniform(0.05, 0.4, P))
            for k in range(P):
                v = rng.normal(size=3)
                o.add_node(t[k], v)
                h.add_node(t[k], v)
            for x in np.linspace(t[0] - 0.2, t[-1] + 0.2, 57):
                assert np.array_equal(o.sample(x), h.sample(x))
