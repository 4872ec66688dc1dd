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

RESIDUAL_PARTICLE, RESIDUAL_PARTICLE_COPY, RESIDUAL_CARTPOLE = 1, 2, 3


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
            if s is None or s["type"] != "framepos" or s["objtype"] != "site":
                raise TaskError(f"trace{i}: only framepos site sensors are supported")
            self.trace_site.append(m.name2id("site", s["objname"]))
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
                    trace_site=self.trace_site, risk=float(self.risk))

    def packed(self) -> PackedTask:
        return PackedTask(self.spec())

    def packed_model(self, planning=True) -> PackedModel:
        """The agent plans on its own model copy with opt.timestep=agent_timestep and
        opt.integrator=agent_integrator (mjpc/agent.cc:97-107, 288-291)."""
        m = self.model
        if not planning:
            return PackedModel(m)
        ts = m.get_number("agent_timestep", m.timestep)
        integ = int(m.get_number("agent_integrator", m.integrator))
        return PackedModel(m, timestep=ts, integrator=integ)

    def planning_steps(self) -> int:
        """steps_ = clamp(horizon/timestep + 1, 1, 512), mjpc/agent.cc:288-293."""
        m = self.model
        ts = m.get_number("agent_timestep", m.timestep)
        horizon = m.get_number("agent_horizon", 0.5)
        return int(max(min(horizon / ts + 1, 512), 1))


_REGISTRY = {
    # name -> (xml path under models/, residual id)
    "Cartpole": ("cartpole/task.xml", RESIDUAL_CARTPOLE),      # mjpc/tasks/cartpole/cartpole.cc
    "Particle": ("particle/task.xml", RESIDUAL_PARTICLE),      # mjpc/test/testdata/particle_residual.h
    "ParticleCopy": ("particle/task.xml", RESIDUAL_PARTICLE_COPY),  # mjpc/test/agent/rollout_test.cc:28-58
}


def task_names():
    return list(_REGISTRY)


def load_task(name: str) -> Task:
    path, rid = _REGISTRY[name]
    model = mjcf.load_xml(os.path.join(MODELS_DIR, path))
    return Task(name=name, residual_id=rid, model=model).reset()
