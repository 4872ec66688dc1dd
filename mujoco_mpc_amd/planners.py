"""Host-side planners over the C ABI: drop-in mirrors of the reference's
`mjpc::SamplingPlanner` (mjpc/planners/sampling/planner.{h,cc}) and
`mjpc::SamplingPolicy` (sampling/policy.{h,cc}).

Only `Rollouts(...)` (planner.cc:355-393) and the sort (planner.cc:184-188) are
replaced by device work; nominal resampling, policy copy and `ActionFromPolicy`
stay host-side with the reference's semantics. Differences, both forced by
SURVEY.md findings: the 128-candidate cap of kMaxTrajectory is lifted (F5) and
noise comes from a seedable counter-based generator (F4).
"""
from __future__ import annotations

import threading
import time as _time

import numpy as np

from . import capi
from .spline import CUBIC, LINEAR, ZERO, TimeSpline
from .task import Task

K_MAX_TRAJECTORY_HORIZON = 512  # mjpc/trajectory.h:27


def sync_task(ctx, task):
    """The per-plan frozen ResidualFn copy (Agent::PlanIteration, agent.cc:319): weights, norm / residual parameters,
    risk and the task-specific residual state reach the device before the plan's rollouts."""
    if hasattr(ctx, "set_task_params"):
        ctx.set_task_params(task.weight, task.norm_parameter, task.parameters, task.risk)
    if hasattr(ctx, "set_residual_state") and (task.residual_int or task.residual_real):
        ctx.set_residual_state(task.residual_int, task.residual_real)


def clamp(x, bounds):
    """Clamp, mjpc/utilities.cc:112-116 (mju_clip = max(lo, min(hi, x)))."""
    b = np.asarray(bounds, dtype=np.float64).reshape(-1, 2)
    np.maximum(b[:, 0], np.minimum(b[:, 1], x), out=x)
    return x


class State:
    """mjpc::State (mjpc/states/state.{h,cc}): the snapshot handed to Planner::SetState."""

    def __init__(self, model):
        self._lock = threading.RLock()
        self.state = np.zeros(model.nq + model.nv + model.na)
        self.mocap = np.zeros(7 * model.nmocap)
        self.userdata = np.zeros(model.nuserdata)
        self.time = 0.0
        # mocap bodies start at their model pose
        for b in range(model.nbody):
            i = int(model.body_mocapid[b])
            if i >= 0:
                self.mocap[7 * i:7 * i + 3] = model.body_pos[b]
                self.mocap[7 * i + 3:7 * i + 7] = model.body_quat[b]

    def set(self, qpos, qvel, act=(), mocap_pos=None, mocap_quat=None, userdata=None, time=0.0):
        with self._lock:
            self.state[:] = np.concatenate([np.asarray(qpos, float), np.asarray(qvel, float), np.asarray(act, float)])
            if mocap_pos is not None:
                mp, mq = np.asarray(mocap_pos, float).reshape(-1, 3), np.asarray(mocap_quat, float).reshape(-1, 4)
                for i in range(mp.shape[0]):
                    self.mocap[7 * i:7 * i + 3], self.mocap[7 * i + 3:7 * i + 7] = mp[i], mq[i]
            if userdata is not None:
                self.userdata[:] = userdata
            self.time = float(time)

    def copy_to(self):
        with self._lock:
            return self.state.copy(), self.mocap.copy(), self.userdata.copy(), self.time


class SamplingPolicy:
    """mjpc::SamplingPolicy (sampling/policy.{h,cc})."""

    def __init__(self):
        self.model = None
        self.plan = TimeSpline(0)
        self.num_spline_points = 0

    def allocate(self, model, task, horizon):
        self.model = model
        self.num_spline_points = int(model.get_number("sampling_spline_points", K_MAX_TRAJECTORY_HORIZON))
        self.plan = TimeSpline(model.nu)
        self.plan.reserve(self.num_spline_points)

    def reset(self, horizon, initial_repeated_action=None):
        self.plan.clear()
        if initial_repeated_action is not None:
            self.plan.add_node(0, initial_repeated_action)

    def action(self, action, state, time):
        self.plan.sample(time, action)
        return clamp(action, self.model.actuator_ctrlrange)

    def copy_from(self, policy, horizon=None):
        self.model = policy.model
        self.plan = policy.plan.copy()
        self.num_spline_points = policy.num_spline_points


class GpuSamplingPlanner:
    """mjpc::SamplingPlanner with the candidate fan-out on the GPU.

    `group`: optional rank group (see distributed.py) sharding the candidates over
    GPUs; every rank keeps an identical host policy."""

    def __init__(self, device=0, precision=64, seed=0, group=None, backend_factory=None):
        self.device, self.precision, self.seed = device, precision, seed
        self.group = group
        self._backend_factory = backend_factory
        self.model = None
        self.task = None
        self.ctx = None
        self.mtx_ = threading.RLock()
        self.noise_exploration = [0.0, 0.0]
        self.iteration = 0
        self.winner = 0
        self.improvement = 0.0
        self.noise_compute_time = 0.0
        self.rollouts_compute_time = 0.0
        self.policy_update_compute_time = 0.0
        self.trajectory_order = []
        self._scores = []

    # ---- Planner::Initialize, planner.cc:41-77
    def initialize(self, model, task: Task):
        self.model, self.task = model, task
        self.noise_exploration[0] = model.get_number("sampling_exploration", 0.1)
        se = model.numeric.get("sampling_exploration")
        self.noise_exploration[1] = float(se[1]) if se is not None and len(se) > 1 else 0.0
        self.num_trajectory_ = int(model.get_number("sampling_trajectories", 10))
        self.interpolation_ = int(model.get_number("sampling_representation", CUBIC))
        self.sliding_plan_ = int(model.get_number("sampling_sliding_plan", 0))
        self.winner = 0

    # ---- Planner::Allocate, planner.cc:80-107
    def allocate(self):
        m = self.model
        self.state = np.zeros(m.nq + m.nv + m.na)
        self.mocap = np.zeros(7 * m.nmocap)
        self.userdata = np.zeros(m.nuserdata)
        self.time = 0.0
        self.policy, self.previous_policy = SamplingPolicy(), SamplingPolicy()
        self.policy.allocate(m, self.task, K_MAX_TRAJECTORY_HORIZON)
        self.previous_policy.allocate(m, self.task, K_MAX_TRAJECTORY_HORIZON)
        self.winner_policy = SamplingPolicy()  # candidate_policy[winner]
        self.winner_policy.allocate(m, self.task, K_MAX_TRAJECTORY_HORIZON)
        self.plan_scratch = TimeSpline(m.nu)
        self._best = None
        if self._backend_factory is not None:
            self.ctx = self._backend_factory(self.task)
        else:
            self.ctx = capi.Context(self.task.packed_model(), self.task.packed(), self.device, self.precision)

    # ---- Planner::Reset, planner.cc:110-147
    def reset(self, horizon, initial_repeated_action=None):
        self.state[:] = 0
        self.mocap[:] = 0
        self.userdata[:] = 0
        self.time = 0.0
        with self.mtx_:
            self.policy.reset(horizon, initial_repeated_action)
            self.previous_policy.reset(horizon, initial_repeated_action)
        self.winner_policy.reset(horizon, initial_repeated_action)
        self.plan_scratch.clear()
        self.improvement = 0.0
        self.winner = 0
        self._best = None

    # ---- Planner::SetState, planner.cc:150-153
    def set_state(self, state: State):
        self.state, self.mocap, self.userdata, self.time = state.copy_to()

    # ---- UpdateNominalPolicy, planner.cc:240-323
    def update_nominal_policy(self, horizon):
        num_spline_points = self.winner_policy.num_spline_points
        nominal_time = self.time
        time_horizon = (horizon - 1) * self._timestep()
        if self.sliding_plan_:
            extra = {ZERO: 1, LINEAR: 2, CUBIC: 4}[self.interpolation_]
            if num_spline_points > extra:
                time_shift = max(time_horizon / (num_spline_points - extra), 1.0e-5)
            else:
                time_shift = time_horizon
            with self.mtx_:
                plan = self.policy.plan
                if plan.size() and plan.node_at(0)[0] > nominal_time:
                    plan.shift_time(nominal_time)
                    self.previous_policy.plan.shift_time(nominal_time)
                plan.discard_before(nominal_time)
                if plan.size() == 0:
                    plan.add_node(nominal_time)
                while plan.size() < num_spline_points:
                    t_last, v_last = plan.node_at(plan.size() - 1)
                    plan.add_node(t_last + time_shift, v_last.copy())
        else:
            if self.interpolation_ == ZERO:
                time_shift = max(time_horizon / num_spline_points, 1.0e-5)
            else:
                time_shift = max(time_horizon / (num_spline_points - 1), 1.0e-5)
            self.plan_scratch.clear()
            self.plan_scratch.set_interpolation(self.interpolation_)
            for _ in range(num_spline_points):
                node = self.plan_scratch.add_node(nominal_time)
                self.winner_policy.action(node, None, nominal_time)
                nominal_time += time_shift
            with self.mtx_:
                self.policy.plan = self.plan_scratch.copy()

    def _timestep(self):
        return self.model.get_number("agent_timestep", self.model.timestep)

    # ---- Rollouts, planner.cc:355-393: the device fan-out
    def rollouts(self, num_trajectory, horizon):
        plan = self.policy.plan
        rank, world = (self.group.rank, self.group.world) if self.group else (0, 1)
        # candidates [rank*n, (rank+1)*n) of the global batch; candidate 0 is the un-noised nominal
        if world > num_trajectory:
            raise ValueError("more ranks than candidates: every rank needs at least one rollout")
        q, r = divmod(num_trajectory, world)   # contiguous ranges, the first (N % world) ranks take one more
        n_local = q + (1 if rank < r else 0)
        offset = rank * q + min(rank, r)
        if rank == world - 1:
            n_local = num_trajectory - offset
        ns = capi.make_noise_spec(seed=self.seed, iteration=self.iteration, mode=capi.NOISE_SAMPLING,
                                  candidate_offset=offset, nominal_candidate=0,
                                  std0=self.noise_exploration[0], std1=self.noise_exploration[1])
        sync_task(self.ctx, self.task)
        self.ctx.set_state(self.state, self.time, self.mocap, self.userdata)
        self.ctx.rollout_noise(n_local, horizon, plan.interpolation(), plan.times(), plan.values(), ns)
        self._offset, self._n_local = offset, n_local

    # ---- OptimizePolicyCandidates, planner.cc:155-194
    def optimize_policy_candidates(self, ncandidates, horizon, pool=None):
        self.update_nominal_policy(horizon)
        num_trajectory = self.num_trajectory_
        ncandidates = min(ncandidates, num_trajectory)
        t0 = _time.perf_counter()
        self.policy.plan.set_interpolation(self.interpolation_)
        self.rollouts(num_trajectory, horizon)
        k = min(ncandidates, self._n_local)
        idx, ret = self.ctx.topk(k)  # device selection replaces std::partial_sort
        idx = idx + self._offset
        if self.group and self.group.world > 1:
            idx, ret = self.group.merge_topk(idx, ret, ncandidates)
        self.trajectory_order = [int(i) for i in idx]
        self._scores = [float(r) for r in ret]
        self.rollouts_compute_time = (_time.perf_counter() - t0) * 1e6
        self.iteration += 1
        return len(self.trajectory_order)

    # ---- OptimizePolicy, planner.cc:197-212
    def optimize_policy(self, horizon, pool=None):
        """OptimizePolicyCandidates(1) + CopyCandidateToPolicy(0), with the device selection, the
        winner's spline and the nominal's return fetched in ONE launch + ONE sync (mjpcx_best)."""
        self.update_nominal_policy(horizon)
        t0 = _time.perf_counter()
        self.policy.plan.set_interpolation(self.interpolation_)
        self.rollouts(self.num_trajectory_, horizon)
        ref = 0 if self._offset == 0 else -1                 # trajectory[0] lives on rank 0
        idx, best_ret, nominal_ret, values = self.ctx.best(ref)
        idx += self._offset
        if self.group and self.group.world > 1:
            idx, best_ret, nominal_ret, values = self.group.exchange_best(idx, best_ret, nominal_ret, values)
        self.trajectory_order = [int(idx)]
        self._scores = [float(best_ret)]
        self.rollouts_compute_time = (_time.perf_counter() - t0) * 1e6
        self.iteration += 1
        t0 = _time.perf_counter()
        self._set_winner(int(idx), values)
        self.improvement = max(nominal_ret - best_ret, 0.0)
        self.policy_update_compute_time = (_time.perf_counter() - t0) * 1e6

    def _set_winner(self, global_idx, values):
        """CopyCandidateToPolicy, planner.cc:534-543, given candidate_policy[winner]'s spline values."""
        self.winner = global_idx
        times = self.policy.plan.times()
        plan = TimeSpline(self.model.nu, self.policy.plan.interpolation())
        for t, v in zip(times, np.asarray(values).reshape(len(times), -1)):
            plan.add_node(t, v)
        self.winner_policy.model = self.model
        self.winner_policy.plan = plan
        self.winner_policy.num_spline_points = self.policy.num_spline_points
        self._best = None                                        # BestTrajectory() is fetched lazily
        with self.mtx_:
            self.previous_policy.copy_from(self.policy)
            self.policy.copy_from(self.winner_policy)

    # ---- NominalTrajectory, planner.cc:215-227
    def nominal_trajectory(self, horizon, pool=None):
        plan = self.winner_policy.plan if self.winner_policy.plan.size() else self.policy.plan
        sync_task(self.ctx, self.task)
        self.ctx.set_state(self.state, self.time, self.mocap, self.userdata)
        if plan.size() == 0:
            times, values = np.array([self.time]), np.zeros((1, 1, self.model.nu))
        else:
            times, values = plan.times(), plan.values()[None]
        self.ctx.rollout_splines(horizon, plan.interpolation(), times, values)
        self._best = self.ctx.fetch_trajectory(0)
        return self._best

    # ---- ActionFromPolicy, planner.cc:230-238: never touches the GPU
    def action_from_policy(self, action, state, time, use_previous=False):
        with self.mtx_:
            return (self.previous_policy if use_previous else self.policy).action(action, state, time)

    # ---- BestTrajectory, planner.cc:396-398 (the reference returns &trajectory[winner]; here the
    # winner's buffers stay on the device until someone asks for them; with several ranks only the
    # owner of the winner has them)
    def best_trajectory(self):
        if self._best is None and self.ctx is not None and self.ctx.N > 0:
            local = self.winner - self._offset
            if 0 <= local < self._n_local:
                self._best = self.ctx.fetch_trajectory(local)
        return self._best

    def num_parameters(self):
        return self.policy.num_spline_points * self.model.nu

    # ---- RankedPlanner quartet, planner.cc:520-543
    def candidate_score(self, candidate):
        return self._scores[candidate]

    def action_from_candidate_policy(self, action, candidate, state, time):
        p = SamplingPolicy()
        p.copy_from(self.policy)
        self._load_candidate_plan(p, self.trajectory_order[candidate])
        return p.action(action, state, time)

    def _load_candidate_plan(self, policy, global_idx):
        """candidate_policy[i].plan: fetched from the rank that rolled it out."""
        local = global_idx - self._offset
        values = None
        owner = 0
        if 0 <= local < self._n_local:
            values = self.ctx.fetch_spline(local)
        if self.group and self.group.world > 1:
            owner = self.group.owner_of(global_idx, self.num_trajectory_)
            values = self.group.broadcast_array(values, (self.ctx.P, self.model.nu), src=owner)
        times = self.policy.plan.times()
        policy.plan = TimeSpline(self.model.nu, self.policy.plan.interpolation())
        for t, v in zip(times, values):
            policy.plan.add_node(t, v)

    def copy_candidate_to_policy(self, candidate):
        p = SamplingPolicy()
        self._load_candidate_plan(p, self.trajectory_order[candidate])
        self._set_winner(self.trajectory_order[candidate], p.plan.values())


class GpuCrossEntropyPlanner:
    """mjpc::CrossEntropyPlanner (mjpc/planners/cross_entropy/planner.{h,cc}) with the candidate
    fan-out, the sort and the elite statistics on the GPU. Differences forced by SURVEY F4/F5 as for
    the sampling planner; the extra nominal rollout (planner.cc:435) rides along as one more candidate."""

    def __init__(self, device=0, precision=64, seed=0, group=None, backend_factory=None):
        self.device, self.precision, self.seed = device, precision, seed
        self.group = group
        self._backend_factory = backend_factory
        self.mtx_ = threading.RLock()
        self.iteration = 0
        self.improvement = 0.0
        self.noise_compute_time = self.rollouts_compute_time = self.policy_update_compute_time = 0.0
        self.trajectory_order = []
        self.interpolation_ = ZERO  # member default kZeroSpline, planner.h:141-142 (CE never reads sampling_representation)

    # ---- Initialize, planner.cc:41-74
    def initialize(self, model, task: Task):
        self.model, self.task = model, task
        self.std_initial_ = model.get_number("sampling_exploration", 0.1)
        self.std_min_ = model.get_number("std_min", 0.01)
        self.explore_fraction_ = model.get_number("explore_fraction", 0.0)
        self.num_trajectory_ = int(model.get_number("sampling_trajectories", 10))
        self.n_elite_ = int(model.get_number("n_elite", max(self.num_trajectory_ // 10, 2)))

    # ---- Allocate, planner.cc:77-119
    def allocate(self):
        m = self.model
        self.state = np.zeros(m.nq + m.nv + m.na)
        self.mocap = np.zeros(7 * m.nmocap)
        self.userdata = np.zeros(m.nuserdata)
        self.time = 0.0
        self.policy, self.resampled_policy, self.previous_policy = SamplingPolicy(), SamplingPolicy(), SamplingPolicy()
        for p in (self.policy, self.resampled_policy, self.previous_policy):
            p.allocate(m, self.task, K_MAX_TRAJECTORY_HORIZON)
        self.variance = np.zeros(m.nu * K_MAX_TRAJECTORY_HORIZON)
        self.times_scratch = np.zeros(K_MAX_TRAJECTORY_HORIZON)
        self.parameters_scratch = np.zeros(m.nu * K_MAX_TRAJECTORY_HORIZON)
        self._nominal = None
        if self._backend_factory is not None:
            self.ctx = self._backend_factory(self.task)
        else:
            self.ctx = capi.Context(self.task.packed_model(), self.task.packed(), self.device, self.precision)

    # ---- Reset, planner.cc:122-160
    def reset(self, horizon, initial_repeated_action=None):
        self.state[:] = 0
        self.mocap[:] = 0
        self.userdata[:] = 0
        self.time = 0.0
        for p in (self.policy, self.resampled_policy, self.previous_policy):
            p.reset(horizon, initial_repeated_action)
        self.variance[:] = self.std_initial_ ** 2
        self.improvement = 0.0
        self._nominal = None

    def set_state(self, state: State):
        self.state, self.mocap, self.userdata, self.time = state.copy_to()

    def _timestep(self):
        return self.model.get_number("agent_timestep", self.model.timestep)

    # ---- ResamplePolicy, planner.cc:322-348
    def resample_policy(self, horizon):
        P = self.resampled_policy.num_spline_points
        nu = self.model.nu
        nominal_time = self.time
        time_shift = max((horizon - 1) * self._timestep() / (P - 1), 1.0e-5)
        for t in range(P):
            self.times_scratch[t] = nominal_time
            self.resampled_policy.action(self.parameters_scratch[t * nu:(t + 1) * nu], None, nominal_time)
            nominal_time += time_shift
        interp = self.policy.plan.interpolation()
        self.resampled_policy.plan.clear()
        for t in range(P):
            self.resampled_policy.plan.add_node(self.times_scratch[t], self.parameters_scratch[t * nu:(t + 1) * nu])
        self.resampled_policy.plan.set_interpolation(interp)

    # ---- OptimizePolicy, planner.cc:168-291
    def optimize_policy(self, horizon, pool=None):
        self.resampled_policy.plan.set_interpolation(self.interpolation_)
        num_trajectory = self.num_trajectory_
        self.n_elite_ = min(self.n_elite_, num_trajectory)
        n_elite = self.n_elite_
        with self.mtx_:
            self.resampled_policy.copy_from(self.policy, self.policy.num_spline_points)
        self.resample_policy(horizon)
        t0 = _time.perf_counter()
        P, nu = self.resampled_policy.num_spline_points, self.model.nu
        np_ = P * nu
        # ---- Rollouts, planner.cc:388-443: N noised candidates + the nominal as global candidate N
        rank, world = (self.group.rank, self.group.world) if self.group else (0, 1)
        if world > num_trajectory:
            raise ValueError("more ranks than candidates: every rank needs at least one rollout")
        q, r = divmod(num_trajectory, world)   # contiguous ranges, the first (N % world) ranks take one more
        n_local = q + (1 if rank < r else 0)
        offset = rank * q + min(rank, r)
        if rank == world - 1:
            n_local = num_trajectory - offset + 1          # the last rank also rolls out the nominal
        explore_count = int(np.sum(np.arange(num_trajectory) < num_trajectory * self.explore_fraction_))
        ns = capi.make_noise_spec(seed=self.seed, iteration=self.iteration, mode=capi.NOISE_CROSS_ENTROPY,
                                  candidate_offset=offset, nominal_candidate=num_trajectory, explore_count=explore_count,
                                  std0=self.std_initial_, std1=self.std_min_, param_variance=self.variance[:np_])
        plan = self.resampled_policy.plan
        sync_task(self.ctx, self.task)
        self.ctx.set_state(self.state, self.time, self.mocap, self.userdata)
        self.ctx.rollout_noise(n_local, horizon, plan.interpolation(), plan.times(), plan.values(), ns)
        self._offset, self._n_local = offset, n_local
        # ---- full sort in the reference (planner.cc:206-211); only the elites matter downstream
        k = min(n_elite + 1, n_local)
        idx, ret = self.ctx.topk(k)
        idx = idx.astype(np.int64) + offset
        if world > 1:
            idx, ret = self.group.merge_topk(idx, ret, n_elite + 1)
        keep = idx != num_trajectory                      # the nominal rollout is not a candidate
        idx, ret = idx[keep][:n_elite], ret[keep][:n_elite]
        self.trajectory_order = [int(i) for i in idx]
        self.rollouts_compute_time = (_time.perf_counter() - t0) * 1e6
        # ---- elite mean / variance, planner.cc:216-270
        t0 = _time.perf_counter()
        mine = np.array([i - offset for i in self.trajectory_order if offset <= i < offset + n_local and i != num_trajectory],
                        dtype=np.int32)
        s, sret = self.ctx.elite_moments(mine)
        if world > 1:
            tot = self.group.sum_array(np.concatenate([s.reshape(-1), [sret]]))
            s, sret = tot[:-1].reshape(P, nu), tot[-1]
        mean = s / n_elite
        avg_return = sret / n_elite
        sq, _ = self.ctx.elite_moments(mine, mean)
        if world > 1:
            sq = self.group.sum_array(sq.reshape(-1)).reshape(P, nu)
        self.variance[:] = 0.0
        with np.errstate(divide="ignore", invalid="ignore"):
            self.variance[:np_] = (sq / (n_elite - 1)).reshape(-1)   # n_elite == 1 -> inf/nan, as the reference
        self.parameters_scratch[:np_] = mean.reshape(-1)
        with self.mtx_:
            self.previous_policy.copy_from(self.policy)
            self.policy.plan.clear()
            self.policy.plan.set_interpolation(self.interpolation_)
            for t in range(P):
                self.policy.plan.add_node(self.times_scratch[t], mean[t])
        self.improvement = max(avg_return - float(ret[0]), 0.0)
        self._nominal = None
        self.iteration += 1
        self.policy_update_compute_time = (_time.perf_counter() - t0) * 1e6

    # ---- NominalTrajectory, planner.cc:294-308
    def nominal_trajectory(self, horizon, pool=None):
        plan = self.resampled_policy.plan
        sync_task(self.ctx, self.task)
        self.ctx.set_state(self.state, self.time, self.mocap, self.userdata)
        self.ctx.rollout_splines(horizon, plan.interpolation(), plan.times(), plan.values()[None])
        self._nominal = self.ctx.fetch_trajectory(0)
        self._n_local = 0
        return self._nominal

    def action_from_policy(self, action, state, time, use_previous=False):
        with self.mtx_:
            return (self.previous_policy if use_previous else self.policy).action(action, state, time)

    # ---- BestTrajectory, planner.cc:446-448: the NOMINAL trajectory
    def best_trajectory(self):
        if self._nominal is None and self.ctx is not None and getattr(self, "_n_local", 0) > 0:
            local = self.num_trajectory_ - self._offset
            if 0 <= local < self._n_local:
                self._nominal = self.ctx.fetch_trajectory(local)
        return self._nominal

    def num_parameters(self):
        return self.policy.num_spline_points * self.model.nu


# ====================================================================================== iLQG
def log_scale(max_value, min_value, steps):
    """LogScale, mjpc/utilities.cc:819-826 (ascending from min_value to max_value)."""
    step = (np.log(max_value) - np.log(min_value)) / max(steps - 1, 1)
    return np.exp(np.log(min_value) + np.arange(steps) * step)


def find_interval(xs, value, length):
    """FindInterval, mjpc/utilities.h:124-144."""
    up = int(np.searchsorted(np.asarray(xs[:length]), value, side="right"))
    lo = up - 1
    if lo < 0:
        return 0, 0
    if lo > length - 1:
        return length - 1, length - 1
    return lo, min(up, length - 1)


def _quat_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def state_diff(model, s1, s2, h=1.0):
    """StateDiff, mjpc/utilities.cc:543-553: (s2 - s1) / h in the tangent space (mj_differentiatePos + velocities)."""
    nq, nv = model.nq, model.nv
    if nq == nv:
        return (np.asarray(s2, float) - np.asarray(s1, float)) / h
    a = model.arrays
    dx = np.zeros(2 * nv)
    for j in range(model.njnt):
        qa, da, t = int(a["jnt_qposadr"][j]), int(a["jnt_dofadr"][j]), int(a["jnt_type"][j])
        if t == 0:  # free
            dx[da:da + 3] = (s2[qa:qa + 3] - s1[qa:qa + 3]) / h
            qa, da = qa + 3, da + 3
        if t in (0, 1):  # free / ball: mju_subQuat
            qb, qa_ = np.asarray(s1[qa:qa + 4], float), np.asarray(s2[qa:qa + 4], float)
            qd = _quat_mul(np.array([qb[0], -qb[1], -qb[2], -qb[3]]), qa_)
            ax = qd[1:].copy()
            sn = np.linalg.norm(ax)
            if sn > 1e-15:
                ax /= sn
            speed = 2 * np.arctan2(sn, qd[0])
            if speed > np.pi:
                speed -= 2 * np.pi
            dx[da:da + 3] = ax * speed / h
        else:
            dx[da] = (s2[qa] - s1[qa]) / h
    dx[nv:] = (np.asarray(s2[nq:], float) - np.asarray(s1[nq:], float)) / h
    return dx


def normalize_state_quaternions(model, x):
    """mj_normalizeQuat on an interpolated state (ilqg/policy.cc:118-125)."""
    a = model.arrays
    for j in range(model.njnt):
        t = int(a["jnt_type"][j])
        if t in (0, 1):
            qa = int(a["jnt_qposadr"][j]) + (3 if t == 0 else 0)
            n = np.linalg.norm(x[qa:qa + 4])
            x[qa:qa + 4] = [1, 0, 0, 0] if n < 1e-15 else x[qa:qa + 4] / n
    return x


class ILQGSettings:
    """iLQGSettings, mjpc/planners/ilqg/settings.h."""
    min_linesearch_step = 1.0e-3
    fd_tolerance = 1.0e-6
    fd_mode = 0
    min_regularization = 1.0e-6
    max_regularization = 1.0e6
    regularization_type = 0
    max_regularization_iterations = 5
    action_limits = 1
    nominal_feedback_scaling = 1
    verbose = 0


class ILQGPolicy:
    """iLQGPolicy, mjpc/planners/ilqg/policy.{h,cc}: a nominal trajectory + time-varying linear feedback."""

    def __init__(self, model, task, horizon_cap=K_MAX_TRAJECTORY_HORIZON):
        self.model = model
        nu, ds, ndx = model.nu, model.nq + model.nv + model.na, 2 * model.nv + model.na
        self.trajectory = capi.Trajectory(ds, nu, task.num_residual, task.num_trace, horizon_cap)
        self.feedback_gain = np.zeros((horizon_cap, nu, ndx))
        self.action_improvement = np.zeros((horizon_cap, nu))
        self.feedback_scaling = 1.0
        self.representation = int(model.get_number("ilqg_representation", 1))

    def reset(self, horizon, initial_repeated_action=None):
        tr = self.trajectory
        for a in (tr.states, tr.times, tr.residual, tr.costs, tr.trace):
            a[...] = 0
        tr.actions[...] = 0 if initial_repeated_action is None else np.asarray(initial_repeated_action)
        tr.total_return, tr.failure = 0.0, False
        self.feedback_gain[:horizon] = 0
        self.action_improvement[:horizon] = 0
        self.feedback_scaling = 1.0

    def copy_from(self, other, horizon):
        import copy
        self.trajectory = copy.deepcopy(other.trajectory)
        self.feedback_gain[:horizon] = other.feedback_gain[:horizon]
        self.action_improvement[:horizon] = other.action_improvement[:horizon]

    @staticmethod
    def _slope(g, xs, ys, length):
        """FiniteDifferenceSlope (utilities.cc:362-395) at grid point g"""
        if g == length - 1:
            return (ys[g] - ys[g - 1]) / (xs[g] - xs[g - 1]) if length > 2 else np.zeros_like(ys[g])
        if g == 0:
            return (ys[1] - ys[0]) / (xs[1] - xs[0])
        return 0.5 * (ys[g + 1] - ys[g]) / (xs[g + 1] - xs[g]) + 0.5 * (ys[g] - ys[g - 1]) / (xs[g] - xs[g - 1])

    @classmethod
    def _interp(cls, x, xs, ys, length, zero, representation=1):
        b0, b1 = find_interval(xs, x, length)
        if zero or b0 == b1:
            return ys[b0].copy()
        span = xs[b1] - xs[b0]
        t = (x - xs[b0]) / span
        if representation != 2:
            return ys[b0] * (1.0 - t) + ys[b1] * t
        c0, c1 = 2.0 * t ** 3 - 3.0 * t * t + 1.0, (t ** 3 - 2.0 * t * t + t) * span
        c2, c3 = -2.0 * t ** 3 + 3 * t * t, (t ** 3 - t * t) * span
        return c0 * ys[b0] + c1 * cls._slope(b0, xs, ys, length) + c2 * ys[b1] + c3 * cls._slope(b1, xs, ys, length)

    def action(self, action, state, time):
        """policy.cc:82-161 (zero-order / linear / cubic representations)."""
        tr, H = self.trajectory, self.trajectory.horizon
        b0, b1 = find_interval(tr.times, time, H)
        zero = b0 == b1 or self.representation == 0
        rep = self.representation
        action[:] = self._interp(time, tr.times, tr.actions, H - 1, zero, rep)
        if state is not None:
            xi = self._interp(time, tr.times, tr.states, H, zero, rep)
            if self.model.nq != self.model.nv:
                xi = normalize_state_quaternions(self.model, xi)
            K = self._interp(time, tr.times, self.feedback_gain, H - 1, zero, rep)
            action += self.feedback_scaling * (K @ state_diff(self.model, xi, np.asarray(state, float)))
        return clamp(action, self.model.actuator_ctrlrange)


class GpuILQGPlanner:
    """mjpc::iLQGPlanner (mjpc/planners/ilqg/planner.{h,cc}) with every data-parallel piece on the GPU:
    feedback / line-search rollouts (mjpcx_rollout_feedback), finite-difference model derivatives
    (mjpcx_transition_fd), cost derivatives (mjpcx_cost_derivatives) and the Riccati sweep on the matrix
    cores (mjpcx_backward_pass). Host side: the regularisation schedule, BestRollout, policy bookkeeping."""

    def __init__(self, device=0, precision=64, backend_factory=None):
        self.device, self.precision = device, precision
        self._backend_factory = backend_factory
        self.settings = ILQGSettings()
        self.mtx_ = threading.RLock()

    def initialize(self, model, task: Task):
        self.model, self.task = model, task
        self.dim_state = model.nq + model.nv + model.na
        self.dim_state_derivative = 2 * model.nv + model.na
        self.dim_action = model.nu
        self.num_rollouts_gui_ = int(model.get_number("ilqg_num_rollouts", 10))
        self.settings.regularization_type = int(model.get_number("ilqg_regularization_type", self.settings.regularization_type))
        self.num_trajectory_ = self.num_rollouts_gui_

    def allocate(self):
        m = self.model
        self.state = np.zeros(self.dim_state)
        self.mocap = np.zeros(7 * m.nmocap)
        self.userdata = np.zeros(m.nuserdata)
        self.time = 0.0
        self.policy = ILQGPolicy(m, self.task)
        self.previous_policy = ILQGPolicy(m, self.task)
        self.candidate0 = ILQGPolicy(m, self.task)          # candidate_policy[0]
        # gradient-based planners plan on the differentiable model copy unless agent_differentiable says otherwise
        self.differentiable_ = bool(int(m.get_number("agent_differentiable", 1)))
        self.ctx = (self._backend_factory(self.task) if self._backend_factory   # test backends: see tests/oracle_backend.py
                    else capi.Context(self.task.packed_model(differentiable=self.differentiable_), self.task.packed(), self.device,
                                      self.precision))

    def reset(self, horizon, initial_repeated_action=None):
        self.state[:] = 0; self.mocap[:] = 0; self.userdata[:] = 0
        self.time = 0.0
        for p in (self.policy, self.previous_policy, self.candidate0):
            p.reset(horizon, initial_repeated_action)
        # iLQGBackwardPass::Reset, backward_pass.cc:50-62
        self.regularization, self.regularization_rate, self.regularization_factor = 1.0, 1.0, 2.0
        self.dV = np.zeros(2)
        self.action_step = self.feedback_scaling = self.improvement = self.expected = self.surprise = 0.0
        self.derivative_skip_ = int(self.model.get_number("derivative_skip", 0))
        self.winner = 0
        self.timers = {}

    def set_state(self, state: State):
        self.state, self.mocap, self.userdata, self.time = state.copy_to()

    # ---- backward_pass.cc:327-356
    def scale_regularization(self, factor, reg_min, reg_max):
        if factor > 1:
            self.regularization_rate = max(self.regularization_rate * factor, factor)
        else:
            self.regularization_rate = min(self.regularization_rate * factor, factor)
        self.regularization = min(max(self.regularization * self.regularization_rate, reg_min), reg_max)

    def update_regularization(self, reg_min, reg_max, z, s):
        bad = lambda v: not np.isfinite(v) or abs(v) > 1e10
        f = self.regularization_factor
        if bad(z) or bad(s):
            self.scale_regularization(f * f, reg_min, reg_max)
        elif z > 0.5 or s > 0.3:
            self.scale_regularization(1.0 / f, reg_min, reg_max)
        elif z < 0.1 or s < 0.06:
            self.scale_regularization(f, reg_min, reg_max)

    def _linesearch_steps(self):
        n = self.num_trajectory_
        steps = np.zeros(n)
        steps[:n - 1] = log_scale(1.0, self.settings.min_linesearch_step, n - 1)
        steps[n - 1] = 0.0
        return steps

    @staticmethod
    def best_rollout(returns, failure):
        """iLQGPlanner::BestRollout, planner.cc:727-740 (scan from the last index, strict <)."""
        best, best_return = -1, 0.0
        for j in range(len(returns) - 1, -1, -1):
            if failure[j]:
                continue
            if best == -1 or returns[j] < best_return:
                best, best_return = j, returns[j]
        return best

    # ---- OptimizePolicy, planner.cc:156-164
    def optimize_policy(self, horizon, pool=None):
        self.num_trajectory_ = self.num_rollouts_gui_       # the reference clamps to kMaxTrajectory = 128 (lifted)
        self.nominal_trajectory(horizon)
        self.iteration(horizon)

    # ---- NominalTrajectory, planner.cc:167-223 + FeedbackRollouts :695-724
    def nominal_trajectory(self, horizon, pool=None):
        if self.num_trajectory_ == 0:
            return
        t0 = _time.perf_counter()
        self.policy.trajectory.horizon = horizon
        steps = self._linesearch_steps()
        tr = self.policy.trajectory
        sync_task(self.ctx, self.task)
        self.ctx.set_state(self.state, self.time, self.mocap, self.userdata)
        self.ctx.rollout_feedback(horizon, 1, self.policy.representation, self.settings.nominal_feedback_scaling,
                                  tr.times[:horizon], tr.states[:horizon], tr.actions[:horizon],
                                  self.policy.feedback_gain[:horizon], self.policy.action_improvement[:horizon], steps)
        ret, fail = self.ctx.returns()
        best = self.best_rollout(ret, fail)
        if best == -1:
            import copy
            self.candidate0.trajectory = copy.deepcopy(self.policy.trajectory)
            self.feedback_scaling = 0.0
        else:
            self._take_trajectory(self.candidate0, best, horizon)
            self.feedback_scaling = steps[best]
        self.candidate0.feedback_gain[:horizon] = self.policy.feedback_gain[:horizon]
        self.candidate0.action_improvement[:horizon] = self.policy.action_improvement[:horizon]
        self.candidate0.representation = self.policy.representation
        self.timers["nominal"] = (_time.perf_counter() - t0) * 1e6

    def _take_trajectory(self, policy, index, horizon):
        """candidate_policy[0].trajectory = trajectory[index] (buffers keep their allocated capacity)."""
        got = self.ctx.fetch_trajectory(index)
        tr = policy.trajectory
        for name in ("states", "actions", "times", "residual", "costs", "trace"):
            getattr(tr, name)[:horizon] = getattr(got, name)
        tr.horizon, tr.total_return, tr.failure = horizon, got.total_return, got.failure

    # ---- ModelDerivatives::Compute incl. skip + interpolation, model_derivatives.cc:45-165
    def _model_derivatives(self, tr, T):
        s = self.derivative_skip_ + 1
        evaluate = [0] + list(range(s, T - s, s)) + [T - 2, T - 1]
        evaluate = sorted(set(e for e in evaluate if 0 <= e < T))
        A, B, C, D = self.ctx.transition_fd(tr.times[evaluate], tr.states[evaluate], tr.actions[evaluate],
                                            self.settings.fd_tolerance, int(self.settings.fd_mode))
        if len(evaluate) == T:
            return A, B, C, D
        full = [np.zeros((T,) + x.shape[1:]) for x in (A, B, C, D)]
        ev = np.array(evaluate)
        for t in range(T):
            k = int(np.searchsorted(ev, t, side="right")) - 1
            e0 = k
            e1 = min(k + 1, len(ev) - 1)
            tt = 0.0 if (ev[e0] == t or e0 == e1) else (t - ev[e0]) / (ev[e1] - ev[e0])
            for f, x in zip(full, (A, B, C, D)):
                f[t] = x[e0] * (1.0 - tt) + x[e1] * tt
        return full

    # ---- Iteration, planner.cc:377-627
    def iteration(self, horizon, pool=None):
        st = self.settings
        c0 = self.candidate0
        tr = c0.trajectory
        T, n, m = horizon, self.dim_state_derivative, self.dim_action
        previous_return = tr.total_return
        steps = self._linesearch_steps()
        t0 = _time.perf_counter()
        A, B, C, D = self._model_derivatives(tr, T)
        # the last step has no transition (model_derivatives.cc:88-92 computes only C there)
        A[T - 1] = 0; B[T - 1] = 0; D[T - 1] = 0
        self.timers["model_derivative"] = (_time.perf_counter() - t0) * 1e6
        t0 = _time.perf_counter()
        cx, cu, cxx, cxu, cuu = self.ctx.cost_derivatives(tr.residual[:T], C, D)
        self.timers["cost_derivative"] = (_time.perf_counter() - t0) * 1e6
        # ---- backward pass with regularisation retries, planner.cc:429-520
        t0 = _time.perf_counter()
        ok, reg_iter = False, 0
        limits = np.asarray(self.model.actuator_ctrlrange, float).reshape(-1, 2)
        while reg_iter < st.max_regularization_iterations and not ok:
            out = self.ctx.backward_pass(self.regularization, st.regularization_type, st.action_limits, A, B, cx, cu, cxx,
                                         cxu, cuu, tr.actions[:T], limits)
            ok = out["ok"]
            if not ok and self.regularization <= st.max_regularization:
                self.scale_regularization(self.regularization_factor, st.min_regularization, st.max_regularization)
                reg_iter += 1
            elif not ok:
                break
        self.timers["backward_pass"] = (_time.perf_counter() - t0) * 1e6
        if not ok:
            return
        self.dV = out["dV"]
        c0.feedback_gain[:T] = out["K"]
        c0.action_improvement[:T] = out["du"]
        # ---- ActionRollouts, planner.cc:630-692: line search over the improvement step
        t0 = _time.perf_counter()
        sync_task(self.ctx, self.task)
        self.ctx.set_state(self.state, self.time, self.mocap, self.userdata)
        self.ctx.rollout_feedback(T, 0, 0, 1, tr.times[:T], tr.states[:T], tr.actions[:T], c0.feedback_gain[:T],
                                  c0.action_improvement[:T], steps)
        ret, fail = self.ctx.returns()
        best = self.best_rollout(ret, fail)
        if best == -1:
            return
        self.winner = best
        # candidate_policy[winner]: the nominal trajectory with actions += step * improvement (NOT re-rolled)
        import copy
        winner_policy = ILQGPolicy.__new__(ILQGPolicy)
        winner_policy.__dict__ = dict(c0.__dict__)
        winner_policy.trajectory = copy.deepcopy(c0.trajectory)
        winner_policy.trajectory.actions[:T] = tr.actions[:T] + steps[best] * c0.action_improvement[:T]
        self._take_trajectory(c0, best, T)
        if best == 0:
            winner_policy.trajectory = copy.deepcopy(c0.trajectory)
        self.action_step = steps[best]
        self.expected = -1.0 * self.action_step * (self.dV[0] + self.action_step * self.dV[1]) + 1.0e-16
        self.improvement = previous_return - float(ret[best])
        self.surprise = min(max(0.0, self.improvement / self.expected), 2.0)
        self.update_regularization(st.min_regularization, st.max_regularization, self.surprise, self.action_step)
        self.timers["rollouts"] = (_time.perf_counter() - t0) * 1e6
        with self.mtx_:
            self.previous_policy.copy_from(self.policy, T)
            self.previous_policy.feedback_scaling = self.policy.feedback_scaling
            self.policy.copy_from(winner_policy, T)
            self.policy.feedback_scaling = 1.0

    def action_from_policy(self, action, state, time, use_previous=False):
        with self.mtx_:
            return (self.previous_policy if use_previous else self.policy).action(action, state, time)

    def best_trajectory(self):
        with self.mtx_:
            return self.policy.trajectory

    def num_parameters(self):
        return self.dim_action * K_MAX_TRAJECTORY_HORIZON
