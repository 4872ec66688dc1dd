"""-m gpu: the agent.proto service end to end over a real localhost channel (mjpc/grpc/agent_service.cc semantics): a client
that only knows the wire schema drives Init / SetState / PlannerStep / GetAction / Step / GetBestTrajectory and the by-name
setters on the GPU planners. Mirrors what python/mujoco_mpc/agent_test.py exercises through mujoco_mpc.Agent."""
import math

import grpc
import numpy as np
import pytest

from mujoco_mpc_amd import grpc_service as gs

pytestmark = pytest.mark.gpu
M = gs.message


@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    from mujoco_mpc_amd.build import build_host
    build_host()
    d = str(tmp_path_factory.mktemp("models"))
    gs.write_task_blobs(d)
    srv, port, servicer = gs.serve(d, 0, num_candidates=256)
    channel = grpc.insecure_channel(f"127.0.0.1:{port}")
    yield gs.AgentStub(channel)
    channel.close()
    srv.stop(0)
    servicer.close()


def rpc_error(call, *a):
    with pytest.raises(grpc.RpcError) as e:
        call(*a)
    return e.value.code(), e.value.details()


def test_cartpole_session(stub):
    stub.Init(M("InitRequest")(task_id="Cartpole"))
    assert list(stub.GetAllModes(M("GetAllModesRequest")()).mode_names) == ["default_mode"]
    assert stub.GetMode(M("GetModeRequest")()).mode == "default_mode"
    st = stub.GetState(M("GetStateRequest")()).state
    assert len(st.qpos) == 2 and len(st.qvel) == 2 and st.time == 0.0 and len(st.mocap_pos) == 0
    # SetState: size check (grpc_agent_util.cc:95-115), then the state is what GetState returns
    code, details = rpc_error(stub.SetState, M("SetStateRequest")(state=M("State")(qpos=[0.0, 1.0, 2.0])))
    assert code == grpc.StatusCode.INVALID_ARGUMENT and "expected qpos size 2, got 3" in details
    stub.SetState(M("SetStateRequest")(state=M("State")(time=0.0, qpos=[0.3, 1.0], qvel=[0.5, -0.2])))
    st = stub.GetState(M("GetStateRequest")()).state
    assert list(st.qpos) == [0.3, 1.0] and list(st.qvel) == [0.5, -0.2]
    # residuals and cost terms at that state (cartpole.cc:35-57: cos(angle) - 1, position - goal, angular velocity, control)
    res = stub.GetResiduals(M("GetResidualsRequest")()).values
    assert sorted(res) == ["Centered", "Control", "Velocity", "Vertical"]
    assert res["Vertical"].values[0] == pytest.approx(math.cos(1.0) - 1, abs=1e-12)
    assert res["Centered"].values[0] == pytest.approx(0.3, abs=1e-12) and res["Velocity"].values[0] == pytest.approx(-0.2, abs=1e-12)
    vw = stub.GetCostValuesAndWeights(M("GetCostValuesAndWeightsRequest")()).values_weights
    w0 = vw["Velocity"].weight
    stub.SetCostWeights(M("SetCostWeightsRequest")(cost_weights={"Velocity": 0.25}))
    assert stub.GetCostValuesAndWeights(M("GetCostValuesAndWeightsRequest")()).values_weights["Velocity"].weight == 0.25
    code, details = rpc_error(stub.SetCostWeights, M("SetCostWeightsRequest")(cost_weights={"Nope": 1.0}))
    assert code == grpc.StatusCode.INVALID_ARGUMENT and details.startswith("Weight 'Nope' not found in task. Available names are:\n  Vertical")
    stub.SetCostWeights(M("SetCostWeightsRequest")(reset_to_defaults=True))
    assert stub.GetCostValuesAndWeights(M("GetCostValuesAndWeightsRequest")()).values_weights["Velocity"].weight == w0
    # task parameters by name
    params = stub.GetTaskParameters(M("GetTaskParametersRequest")()).parameters
    assert list(params) == ["Goal"] and params["Goal"].WhichOneof("value") == "numeric" and params["Goal"].numeric == 0.0
    stub.SetTaskParameters(M("SetTaskParametersRequest")(parameters={"Goal": M("TaskParameterValue")(numeric=0.5)}))
    assert stub.GetTaskParameters(M("GetTaskParametersRequest")()).parameters["Goal"].numeric == 0.5
    assert stub.GetResiduals(M("GetResidualsRequest")()).values["Centered"].values[0] == pytest.approx(0.3 - 0.5, abs=1e-12)
    code, details = rpc_error(stub.SetTaskParameters, M("SetTaskParametersRequest")(parameters={"Nope": M("TaskParameterValue")(numeric=1)}))
    assert code == grpc.StatusCode.INVALID_ARGUMENT and "Parameter Nope not found in task." in details and "Goal" in details
    stub.SetTaskParameters(M("SetTaskParametersRequest")(parameters={"Goal": M("TaskParameterValue")(numeric=0.0)}))
    # plan, read the plan, act
    for _ in range(3):
        stub.PlannerStep(M("PlannerStepRequest")())
    tr = stub.GetBestTrajectory(M("GetBestTrajectoryRequest")())
    assert tr.steps == 101 and len(tr.states) == 101 * 4 and len(tr.actions) == 100 and len(tr.times) == 101
    assert np.allclose(np.diff(tr.times), 0.01) and list(tr.states[:4]) == [0.3, 1.0, 0.5, -0.2]
    action = stub.GetAction(M("GetActionRequest")()).action
    assert len(action) == 1 and action[0] == pytest.approx(tr.actions[0], abs=1e-6) and abs(action[0]) <= 1.0
    nominal = stub.GetAction(M("GetActionRequest")(time=0.0, averaging_duration=0.05, nominal_action=True)).action
    assert abs(nominal[0]) <= 1.0
    averaged = stub.GetAction(M("GetActionRequest")(averaging_duration=0.01)).action  # rolls the physics out under the policy
    assert abs(averaged[0]) <= 1.0
    # closed loop: Step advances the service's physics by the model's own timestep under the planner's policy
    t0 = stub.GetState(M("GetStateRequest")()).state.time
    for i in range(20):
        if i % 5 == 0:
            stub.PlannerStep(M("PlannerStepRequest")())
        stub.Step(M("StepRequest")(use_previous_policy=(i == 3)))
    st = stub.GetState(M("GetStateRequest")()).state
    assert st.time == pytest.approx(t0 + 20 * 0.001, abs=1e-12) and list(st.qpos) != [0.3, 1.0]
    stub.Reset(M("ResetRequest")())
    st = stub.GetState(M("GetStateRequest")()).state
    assert st.time == 0.0 and list(st.qvel) == [0.0, 0.0]


def test_quadruped_session(stub):
    stub.Init(M("InitRequest")(task_id="Quadruped Flat"))  # the reference's task name (quadruped.cc:31)
    assert list(stub.GetAllModes(M("GetAllModesRequest")()).mode_names) == ["Quadruped", "Biped", "Walk", "Scramble", "Flip"]
    assert stub.GetMode(M("GetModeRequest")()).mode == "Quadruped"
    code, details = rpc_error(stub.SetMode, M("SetModeRequest")(mode="Moonwalk"))
    assert code == grpc.StatusCode.INVALID_ARGUMENT and details.startswith("Mode 'Moonwalk' not found in task. Available names are:\n  Quadruped")
    params = stub.GetTaskParameters(M("GetTaskParametersRequest")()).parameters
    assert params["Gait"].selection == "Stand" and params["Gait switch"].selection == "Automatic" and params["Cadence"].numeric == 2
    assert params["Flip dir"].selection == "Back Flip" and len(params) == 11
    st = stub.GetState(M("GetStateRequest")()).state
    assert len(st.qpos) == 19 and len(st.qvel) == 18 and len(st.mocap_pos) == 6 and len(st.mocap_quat) == 8
    assert st.qpos[2] == pytest.approx(0.26, abs=0.05)  # the home keyframe
    # selections by option string; the next Transition (SetState) applies the gait's parameters (quadruped.cc:299-317)
    stub.SetTaskParameters(M("SetTaskParametersRequest")(parameters={"Gait switch": M("TaskParameterValue")(selection="Manual"),
                                                                       "Gait": M("TaskParameterValue")(selection="Trot")}))
    stub.SetState(M("SetStateRequest")(state=M("State")(time=0.0)))
    params = stub.GetTaskParameters(M("GetTaskParametersRequest")()).parameters
    assert params["Gait"].selection == "Trot" and params["Duty ratio"].numeric == 0.45 and params["Amplitude"].numeric == 0.03
    vw = stub.GetCostValuesAndWeights(M("GetCostValuesAndWeightsRequest")()).values_weights
    assert vw["Balance"].weight == 0.2 and len(vw) == 9
    # SetAnything: mocap pose by body name, mode; errors as grpc_agent_util.cc:439-480
    req = M("SetAnythingRequest")(mode="Walk")
    req.mocap["goal"].pos.extend([1.0, 0.5, 0.26])
    stub.SetAnything(req)
    assert stub.GetMode(M("GetModeRequest")()).mode == "Walk"
    assert list(stub.GetState(M("GetStateRequest")()).state.mocap_pos[:3]) == [1.0, 0.5, 0.26]
    bad = M("SetAnythingRequest")()
    bad.mocap["nope"].pos.extend([0, 0, 0])
    assert rpc_error(stub.SetAnything, bad) == (grpc.StatusCode.INVALID_ARGUMENT, "Body 'nope' not found.")
    bad = M("SetAnythingRequest")()
    bad.mocap["trunk"].pos.extend([0, 0, 0])
    assert rpc_error(stub.SetAnything, bad) == (grpc.StatusCode.INVALID_ARGUMENT, "Body 'trunk' is not a mocap body.")
    stub.SetMode(M("SetModeRequest")(mode="Quadruped"))
    # plan + act on the contact model (the task XML asks for planner 2 = iLQG)
    stub.PlannerStep(M("PlannerStepRequest")())
    tr = stub.GetBestTrajectory(M("GetBestTrajectoryRequest")())
    assert tr.steps == 36 and len(tr.states) == 36 * 37 and len(tr.actions) == 35 * 12
    action = np.array(stub.GetAction(M("GetActionRequest")()).action)
    assert action.shape == (12,) and np.all(np.abs(action) <= 1.0)
    z0 = stub.GetState(M("GetStateRequest")()).state
    for _ in range(10):
        stub.Step(M("StepRequest")())
    z1 = stub.GetState(M("GetStateRequest")()).state
    assert z1.time > z0.time and np.all(np.isfinite(z1.qpos)) and abs(z1.qpos[2] - z0.qpos[2]) < 0.1


def test_init_with_a_model_override(stub):
    """InitRequest.model.xml (grpc_agent_util.cc:535-560, Agent::OverrideModel): the client sends a self-contained MJCF -- here
    the Cartpole task with its includes expanded and a shorter horizon -- and the agent plans on THAT model"""
    import os
    import xml.etree.ElementTree as ET
    from mujoco_mpc_amd import mjcf
    from mujoco_mpc_amd.task import MODELS_DIR
    path = os.path.join(MODELS_DIR, "cartpole", "task.xml")
    root = ET.parse(path).getroot()
    mjcf._expand_includes(root, os.path.dirname(path))
    for n in root.iter("numeric"):
        if n.get("name") == "agent_horizon":
            n.set("data", "0.5")
    stub.Init(M("InitRequest")(task_id="Cartpole", model=M("MjModel")(xml=ET.tostring(root, encoding="unicode"))))
    stub.PlannerStep(M("PlannerStepRequest")())
    tr = stub.GetBestTrajectory(M("GetBestTrajectoryRequest")())
    assert tr.steps == 51 and len(tr.actions) == 50  # 0.5 s / 0.01 s + 1 (agent.cc:288-293), not the task file's 1.0 s
    stub.Init(M("InitRequest")(task_id="Cartpole"))   # a later Init without a model starts afresh on the registered one
    stub.PlannerStep(M("PlannerStepRequest")())
    assert stub.GetBestTrajectory(M("GetBestTrajectoryRequest")()).steps == 101
