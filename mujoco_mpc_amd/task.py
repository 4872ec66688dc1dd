"""Host-side mirror of mjpc::Task (mjpc/task.{h,cc}): parses the cost
specification from a compiled model exactly as `Task::Reset` does
(task.cc:147-248) and maps each reference task to its device residual id."""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

from . import mjcf
from .cstructs import PackedModel, PackedTask

MODELS_DIR = os.path.join(os.path.dirname(__file__), "models")

# mjpc::NormType (mjpc/norm.h:24-35)
NORM_NULL, NORM_QUADRATIC, NORM_L22, NORM_L2, NORM_COSH = -1, 0, 1, 2, 3
NORM_POWER_LOSS, NORM_SMOOTH_ABS, NORM_SMOOTH_ABS2, NORM_RECTIFY = 5, 6, 7, 8
K_MAX_COST_TERMS = 128  # task.h:31
K_MAX_TRACES = 99       # "Number of traces should be less than 100"

RESIDUAL_PARTICLE, RESIDUAL_PARTICLE_COPY, RESIDUAL_CARTPOLE, RESIDUAL_QUADRUPED_FLAT, RESIDUAL_HUMANOID_TRACK = 1, 2, 3, 4, 5


def norm_parameter_dimension(norm_type: int) -> int:
    """NormParameterDimension, mjpc/norm.cc:25-47."""
    return {NORM_L22: 2, NORM_L2: 1, NORM_COSH: 1, NORM_POWER_LOSS: 1, NORM_SMOOTH_ABS: 1,
            NORM_SMOOTH_ABS2: 2, NORM_RECTIFY: 1}.get(int(norm_type), 0)


class TaskError(ValueError):
    """The reference calls mju_error (abort) for these; we raise."""


@dataclass
class Task:
    """mjpc::Task public data members after Reset() (task.h:136-150)."""
    name: str
    residual_id: int
    model: mjcf.FlatModel
    mode: int = 0
    risk: float = 0.0
    num_residual: int = 0
    num_term: int = 0
    num_trace: int = 0
    dim_norm_residual: list = field(default_factory=list)
    num_norm_parameter: list = field(default_factory=list)
    norm: list = field(default_factory=list)
    weight: list = field(default_factory=list)
    weight_names: list = field(default_factory=list)
    norm_parameter: list = field(default_factory=list)
    parameters: list = field(default_factory=list)
    trace_site: list = field(default_factory=list)
    residual_int: list = field(default_factory=list)    # task-specific frozen ResidualFn state (mjpcx_task::residual_int)
    residual_real: list = field(default_factory=list)

    def parameter_index(self, name: str) -> int:
        """ParameterIndex, mjpc/utilities.cc:207-223: index among the "residual_" numerics."""
        keys = [k for k in self.model.numeric if k.startswith("residual_")]
        return keys.index("residual_" + name)

    def cost_term_by_name(self, name: str) -> int:
        return self.weight_names.index(name)

    def reset(self):
        """Task::Reset, mjpc/task.cc:147-248."""
        m = self.model
        self.mode = 0
        self.risk = m.get_number("task_risk", 0.0)
        sensors = m.sensors
        if not sensors or sensors[0]["type"] != "user":
            raise TaskError("Cost construction from XML: User sensors specifying residuals must be "
                            "specified first and sequentially")
        num_term = len(sensors)
        for i in range(1, len(sensors)):
            if sensors[i]["type"] != "user":
                num_term = i
                break
        if num_term > K_MAX_COST_TERMS:
            raise TaskError("Number of cost terms exceeds maximum.")
        self.num_term = num_term
        traces = [s for s in sensors if s["name"].startswith("trace")]
        if len(traces) > K_MAX_TRACES:
            raise TaskError("Number of traces should be less than 100")
        self.num_trace = len(traces)
        # GetTraces resolves sensor "trace%i" at run time (utilities.cc:268-286);
        # resolve to site ids once, at bake time.
        self.trace_site = []
        for i in range(self.num_trace):
            s = next((s for s in sensors if s["name"] == f"trace{i}"), None)
            if s is None or s["type"] != "framepos" or s["objtype"] not in ("site", "body"):
                raise TaskError(f"trace{i}: only framepos site / body sensors are supported")
            # site id, or -1 - body id for a body frame (mjpcx_task::trace_site)
            self.trace_site.append(m.name2id("site", s["objname"]) if s["objtype"] == "site" else -1 - m.name2id("body", s["objname"]))
        self.num_residual = 0
        self.dim_norm_residual, self.num_norm_parameter, self.norm = [], [], []
        self.weight, self.weight_names, self.norm_parameter = [], [], []
        for i in range(num_term):
            s = sensors[i]
            user = list(s["user"]) + [0.0] * (m.nuser_sensor - len(s["user"]))
            self.num_residual += s["dim"]
            npar = norm_parameter_dimension(int(user[0]))
            if 4 + npar > m.nuser_sensor:
                raise TaskError(f"Cost construction from XML: Missing parameter value. sensor ID = {i} ({s['name']})")
            for j in range(npar):
                if user[4 + j] <= 0.0:
                    raise TaskError(f"Cost construction from XML: Missing parameter value. sensor ID = {i} ({s['name']})")
            if int(user[0]) == NORM_NULL and s["dim"] != 1:
                raise TaskError(f"Cost construction from XML: Missing parameter value. sensor ID = {i} ({s['name']})")
            self.dim_norm_residual.append(s["dim"])
            self.norm.append(int(user[0]))
            self.weight.append(user[1])
            self.weight_names.append(s["name"])
            self.num_norm_parameter.append(npar)
            self.norm_parameter += user[4:4 + npar]
        # SetFeatureParameters, task.cc:38-64 ("residual_select_*" default selection = first value)
        self.parameters = [float(v[0]) for k, v in m.numeric.items() if k.startswith("residual_")]
        return self

    # ---- host-side cost (BaseResidualFn::CostTerms/CostValue are evaluated on
    # the device in the hot path; this is only the struct handed to the C ABI)
    def spec(self) -> dict:
        return dict(residual_id=self.residual_id, num_residual=self.num_residual, num_term=self.num_term,
                    num_trace=self.num_trace, num_parameter=len(self.parameters),
                    dim_norm_residual=self.dim_norm_residual, norm=self.norm,
                    num_norm_parameter=self.num_norm_parameter, weight=self.weight,
                    norm_parameter=self.norm_parameter, parameters=self.parameters,
                    trace_site=self.trace_site, risk=float(self.risk),
                    num_residual_int=len(self.residual_int), num_residual_real=len(self.residual_real),
                    residual_int=self.residual_int, residual_real=self.residual_real)

    def packed(self) -> PackedTask:
        return PackedTask(self.spec())

    def packed_model(self, planning=True, differentiable=False) -> PackedModel:
        """The agent plans on its own model copy with opt.timestep=agent_timestep and
        opt.integrator=agent_integrator (mjpc/agent.cc:97-107, 288-291); gradient-based planners plan on a
        "differentiable" copy (MakeDifferentiable, agent.cc:156-164, 295-311)."""
        m = self.model
        if not planning:
            return PackedModel(m)
        ts = m.get_number("agent_timestep", m.timestep)
        integ = int(m.get_number("agent_integrator", m.integrator))
        return PackedModel(m, timestep=ts, integrator=integ, differentiable=differentiable)

    def planning_steps(self) -> int:
        """steps_ = clamp(horizon/timestep + 1, 1, 512), mjpc/agent.cc:288-293."""
        m = self.model
        ts = m.get_number("agent_timestep", m.timestep)
        horizon = m.get_number("agent_horizon", 0.5)
        return int(max(min(horizon / ts + 1, 512), 1))


class QuadrupedFlat(Task):
    """mjpc::QuadrupedFlat (mjpc/tasks/quadruped/quadruped.{h,cc}): ResetLocked (:520-607) resolves the ids and the flip
    kinematics; `transition` is TransitionLocked (:229-391) for the state it can maintain without sensor feedback
    (phase clock, manual gait switch, mode); the frozen ResidualFn copy handed to the device is residual_int/real in
    the layout documented in csrc/residuals.h."""
    MODE_QUADRUPED, MODE_BIPED, MODE_WALK, MODE_SCRAMBLE, MODE_FLIP = range(5)
    GAIT_PARAM = [(1, 1, 0, 0, 1, 1), (0.75, 1, 0.03, 0, 1, 1), (0.45, 2, 0.03, 0.2, 1, 1), (0.4, 4, 0.05, 0.03, 0.5, 0.2),
                  (0.3, 3.5, 0.10, 0.03, 0.2, 0.1)]   # kGaitParam, quadruped.h:99-108
    K_MAX_HEIGHT, K_LEAP_HEIGHT, K_CROUCH_HEIGHT, K_HEIGHT_QUADRUPED = 0.8, 0.5, 0.15, 0.25

    def reset(self):
        super().reset()
        m = self.model
        self.ids = dict(
            gait=self.parameter_index("select_Gait"), gait_switch=self.parameter_index("select_Gait switch"),
            flip_dir=self.parameter_index("select_Flip dir"), biped_type=self.parameter_index("select_Biped type"),
            cadence=self.parameter_index("Cadence"), amplitude=self.parameter_index("Amplitude"),
            duty=self.parameter_index("Duty ratio"), arm_posture=self.parameter_index("Arm posture"),
            heading=self.parameter_index("Heading"),
            balance=self.cost_term_by_name("Balance"), upright=self.cost_term_by_name("Upright"),
            height=self.cost_term_by_name("Height"),
            torso=m.name2id("body", "trunk"), head=m.name2id("site", "head"),
            goal_mocap=int(m.arrays["body_mocapid"][m.name2id("body", "goal")]),
            feet=[m.name2id("geom", n) for n in ("FL", "HL", "FR", "HR")],
            key_home=m.names["key"].index("home"), key_crouch=m.names["key"].index("crouch"))
        # task state managed by Transition (quadruped.h:186-214)
        self.current_mode, self.last_transition_time = self.MODE_QUADRUPED, -1.0
        self.mode_start_time, self.position, self.heading_vec = 0.0, [0.0, 0.0, 0.0], [0.0, 0.0]
        self.speed, self.angvel, self.ground, self.orientation = 0.0, 0.0, 0.0, [0.0, 0.0, 0.0, 0.0]
        self.current_gait, self.phase_start, self.phase_start_time, self.phase_velocity = 0, 0.0, 0.0, 0.0
        # derived kinematic quantities of the flip (quadruped.cc:566-606)
        g = float(np.linalg.norm(m.gravity))
        jump_vel = np.sqrt(2 * g * (self.K_MAX_HEIGHT - self.K_LEAP_HEIGHT))
        flight_time = 2 * jump_vel / g
        jump_acc = jump_vel * jump_vel / (2 * (self.K_LEAP_HEIGHT - self.K_CROUCH_HEIGHT))
        crouch_time = np.sqrt(2 * (self.K_HEIGHT_QUADRUPED - self.K_CROUCH_HEIGHT) / jump_acc)
        leap_time = jump_vel / jump_acc
        jump_time = crouch_time + leap_time
        crouch_vel = -jump_acc * crouch_time
        land_time = 2 * (self.K_LEAP_HEIGHT - self.K_HEIGHT_QUADRUPED) / jump_vel
        land_acc = jump_vel / land_time
        flight_rot_vel = 1.25 * np.pi / flight_time
        jump_rot_vel = np.pi / leap_time - flight_rot_vel
        jump_rot_acc = (flight_rot_vel - jump_rot_vel) / leap_time
        land_rot_acc = 2 * (flight_rot_vel * land_time - np.pi / 4) / (land_time * land_time)
        self.flip = [g, jump_vel, flight_time, jump_acc, crouch_time, leap_time, jump_time, crouch_vel, land_time, land_acc,
                     flight_rot_vel, jump_rot_vel, jump_rot_acc, land_rot_acc]
        self._freeze()
        return self

    def get_phase(self, time):
        return self.phase_start + (time - self.phase_start_time) * self.phase_velocity

    def transition(self, time, mode=None):
        """TransitionLocked without the sensor-driven parts (automatic gait switching, Walk goal motion, Flip)."""
        mode = self.mode if mode is None else mode
        if time < self.last_transition_time or self.last_transition_time == -1:
            if mode not in (self.MODE_QUADRUPED, self.MODE_BIPED):
                mode = self.MODE_QUADRUPED
            self.last_transition_time = self.phase_start_time = self.phase_start = time
        if mode != self.current_mode and self.current_mode != self.MODE_QUADRUPED and mode in (self.MODE_WALK, self.MODE_FLIP):
            mode = self.MODE_QUADRUPED
        phase_velocity = 2 * np.pi * self.parameters[self.ids["cadence"]]
        if phase_velocity != self.phase_velocity:
            self.phase_start = self.get_phase(time)
            self.phase_start_time = time
            self.phase_velocity = phase_velocity
        if mode == self.MODE_BIPED:
            self.parameters[self.ids["gait"]] = 2.0
        gait_selection = int(self.parameters[self.ids["gait"]])
        if gait_selection != self.current_gait:
            self.current_gait = gait_selection
            gp = self.GAIT_PARAM[2 if self.current_mode == self.MODE_BIPED else gait_selection]
            self.parameters[self.ids["duty"]], self.parameters[self.ids["cadence"]], self.parameters[self.ids["amplitude"]] = gp[:3]
            self.weight[self.ids["balance"]], self.weight[self.ids["upright"]], self.weight[self.ids["height"]] = gp[3:]
        self.mode = self.current_mode = mode
        self.last_transition_time = time
        self._freeze()

    def _freeze(self):
        i = self.ids
        self.residual_int = [self.current_mode, i["torso"], i["head"], i["goal_mocap"], *i["feet"], int(self.current_gait),
                             int(self.parameters[i["flip_dir"]]), int(self.parameters[i["biped_type"]]), i["amplitude"],
                             i["duty"], i["arm_posture"], i["heading"], i["key_home"], i["key_crouch"]]
        self.residual_real = [self.mode_start_time, *self.position, *self.heading_vec, self.speed, self.angvel, self.ground,
                              *self.orientation, self.phase_start, self.phase_start_time, self.phase_velocity, *self.flip]


class HumanoidTrack(Task):
    """mjpc::humanoid::Tracking (mjpc/tasks/humanoid/tracking/tracking.{h,cc}): the residual follows CMU mocap keyframes
    (`model.key_mpos`, 30 fps, linear interpolation); `transition` is TransitionLocked (:219-264). Frozen ResidualFn
    state for the device: residual_int = [first key, last key, 16 tracking-site ids, 16 mocap ids], residual_real =
    [reference_time] (csrc/wave_residual.h)."""
    K_FPS = 30.0
    MOTION_LENGTHS = [121, 154, 115, 78, 145, 188, 260, 279, 39, 510]   # kMotionLengths, tracking.cc:45-56
    BODY_NAMES = ["pelvis", "head", "ltoe", "rtoe", "lheel", "rheel", "lknee", "rknee", "lhand", "rhand", "lelbow",
                  "relbow", "lshoulder", "rshoulder", "lhip", "rhip"]           # tracking.cc:71-75

    def reset(self):
        super().reset()
        m = self.model
        if m.scalars["nkey"] == 0:   # the keyframe table travels as a compact npz next to the task XML
            kf = np.load(os.path.join(os.path.dirname(m.source), "tracking_keyframes.npz"))
            qpos = np.where(np.isnan(kf["qpos"]), np.asarray(m.arrays["qpos0"], float)[None, :], kf["qpos"])
            mjcf.attach_keyframes(m, [str(n) for n in kf["names"]], qpos, kf["qvel"], kf["mpos"])
        self.site_ids = [m.name2id("site", f"tracking[{n}]") for n in self.BODY_NAMES]
        self.mocap_ids = [int(m.arrays["body_mocapid"][m.name2id("body", f"mocap[{n}]")]) for n in self.BODY_NAMES]
        self.current_mode, self.reference_time = 0, 0.0
        self._freeze()
        return self

    def motion_start(self, mode):
        return int(sum(self.MOTION_LENGTHS[:mode]))

    @staticmethod
    def interpolation_values(index, max_index):
        """ComputeInterpolationValues, tracking.cc:29-38."""
        c = min(max(index, 0.0), float(max_index))
        i0 = int(np.floor(c))
        i1 = min(i0 + 1, max_index)
        w1 = c - i0
        return i0, i1, 1.0 - w1, w1

    def transition(self, time, mode=None):
        """TransitionLocked: returns the simulation-state edits the reference applies to mjData: always `mocap_pos`
        [16, 3]; on a motion switch or at time 0 also `qpos` / `qvel` of the motion's first keyframe."""
        mode = self.mode if mode is None else mode
        m = self.model
        start, length = self.motion_start(mode), self.MOTION_LENGTHS[mode]
        edits = {}
        if self.current_mode != mode or time == 0.0:
            self.current_mode = mode
            self.reference_time = time
            edits["qpos"] = np.array(m.arrays["key_qpos"][start], float)
            edits["qvel"] = np.array(m.arrays["key_qvel"][start], float)
        self.mode = mode
        i0, i1, w0, w1 = self.interpolation_values((time - self.reference_time) * self.K_FPS + start, start + length - 1)
        mpos = np.asarray(m.arrays["key_mpos"], float)
        edits["mocap_pos"] = (mpos[i0] * w0 + mpos[i1] * w1).reshape(-1, 3)
        self._freeze()
        return edits

    def _freeze(self):
        start = self.motion_start(self.current_mode)
        self.residual_int = [start, start + self.MOTION_LENGTHS[self.current_mode] - 1, *self.site_ids, *self.mocap_ids]
        self.residual_real = [self.reference_time]


_REGISTRY = {
    # name -> (xml path under models/, residual id)
    "Cartpole": ("cartpole/task.xml", RESIDUAL_CARTPOLE),      # mjpc/tasks/cartpole/cartpole.cc
    "Particle": ("particle/task.xml", RESIDUAL_PARTICLE),      # mjpc/test/testdata/particle_residual.h
    "ParticleCopy": ("particle/task.xml", RESIDUAL_PARTICLE_COPY),  # mjpc/test/agent/rollout_test.cc:28-58
    "QuadrupedFlat": ("quadruped/task_flat.xml", RESIDUAL_QUADRUPED_FLAT),  # mjpc/tasks/quadruped/quadruped.cc
    "HumanoidTrack": ("humanoid/tracking/task.xml", RESIDUAL_HUMANOID_TRACK),  # mjpc/tasks/humanoid/tracking/tracking.cc
}
_CLASSES = {"QuadrupedFlat": QuadrupedFlat, "HumanoidTrack": HumanoidTrack}


def task_names():
    return list(_REGISTRY)


def load_task(name: str) -> Task:
    path, rid = _REGISTRY[name]
    model = mjcf.load_xml(os.path.join(MODELS_DIR, path))
    return _CLASSES.get(name, Task)(name=name, residual_id=rid, model=model).reset()
