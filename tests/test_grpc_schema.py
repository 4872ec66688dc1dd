"""The gRPC front end's wire schema (mujoco_mpc_amd/grpc_service.py) against the reference's agent.proto, through the golden
table tools/dump_agent_proto_fields.py extracted from it (tests/golden/agent_proto_fields.json), plus known-answer encodings
and the status codes of the handlers that need no GPU (agent_service.cc: FAILED_PRECONDITION before Init, INVALID_ARGUMENT for
an unknown task)."""
import json
import os
import struct

import grpc
import pytest
from google.protobuf import descriptor_pb2

from mujoco_mpc_amd import grpc_service as gs

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "agent_proto_fields.json")))
T = descriptor_pb2.FieldDescriptorProto
SCALAR = {"double": T.TYPE_DOUBLE, "float": T.TYPE_FLOAT, "bool": T.TYPE_BOOL, "string": T.TYPE_STRING, "bytes": T.TYPE_BYTES,
          "int32": T.TYPE_INT32}


def test_schema_equals_the_reference_proto():
    fd = gs.file_descriptor_proto()
    assert fd.package == GOLDEN["package"] == "agent" and fd.syntax == "proto3"
    ours = {m.name: m for m in fd.message_type}
    assert sorted(ours) == sorted(GOLDEN["messages"])
    for name, fields in GOLDEN["messages"].items():
        msg = ours[name]
        assert [f.number for f in msg.field] == [f["number"] for f in fields], name
        for f, g in zip(msg.field, fields):
            where = f"{name}.{g['name']}"
            assert f.name == g["name"], where
            if g["type"].startswith("map<"):
                k, v = g["type"][4:-1].split(",")
                entry = next(n for n in msg.nested_type if f.type_name.endswith("." + n.name))
                assert entry.options.map_entry and f.label == T.LABEL_REPEATED and f.type == T.TYPE_MESSAGE, where
                assert entry.field[0].type == SCALAR[k] and entry.field[0].number == 1, where
                if v in SCALAR:
                    assert entry.field[1].type == SCALAR[v], where
                else:
                    assert entry.field[1].type == T.TYPE_MESSAGE and entry.field[1].type_name == ".agent." + v, where
                continue
            if g["type"] in SCALAR:
                assert f.type == SCALAR[g["type"]], where
            else:
                assert f.type == T.TYPE_MESSAGE and f.type_name == ".agent." + g["type"], where
            assert (f.label == T.LABEL_REPEATED) == (g["label"] == "repeated"), where
            assert f.proto3_optional == (g["label"] == "optional"), where
            assert f.options.packed == g["packed"], where
            if g["oneof"]:
                assert msg.oneof_decl[f.oneof_index].name == g["oneof"], where
    svc = fd.service[0]
    assert svc.name == GOLDEN["service"] == "Agent"
    assert [[m.name, m.input_type, m.output_type] for m in svc.method] == [[n, ".agent." + i, ".agent." + o] for n, i, o in GOLDEN["methods"]]


def test_known_answer_encodings():
    st = gs.message("State")(time=1.5, qpos=[1.0, 2.0])
    assert st.SerializeToString() == b"\x09" + struct.pack("<d", 1.5) + b"\x12\x10" + struct.pack("<dd", 1.0, 2.0)
    assert not gs.message("State")().HasField("time") and st.HasField("time")  # explicit presence, as `optional double time`
    assert gs.message("TaskParameterValue")(selection="Trot").SerializeToString() == b"\x12\x04Trot"
    assert gs.message("GetActionResponse")(action=[0.5]).SerializeToString() == b"\x0a\x04" + struct.pack("<f", 0.5)
    req = gs.message("SetCostWeightsRequest")(reset_to_defaults=True)
    req.cost_weights["Velocity"] = 2.0
    assert req.SerializeToString() == b"\x08\x01\x12\x13\x0a\x08Velocity\x11" + struct.pack("<d", 2.0)
    back = gs.message("SetCostWeightsRequest").FromString(req.SerializeToString())
    assert dict(back.cost_weights) == {"Velocity": 2.0}


@pytest.fixture(scope="module")
def server(tmp_path_factory):
    from mujoco_mpc_amd.build import build_host
    build_host()
    d = str(tmp_path_factory.mktemp("models"))
    gs.write_task_blobs(d)
    srv, port, servicer = gs.serve(d, 0)
    channel = grpc.insecure_channel(f"127.0.0.1:{port}")
    yield gs.AgentStub(channel)
    channel.close()
    srv.stop(0)
    servicer.close()


def test_handlers_before_init_and_unknown_task(server):
    for rpc, req in (("GetState", "GetStateRequest"), ("PlannerStep", "PlannerStepRequest"), ("GetAllModes", "GetAllModesRequest"),
                     ("GetBestTrajectory", "GetBestTrajectoryRequest"), ("Reset", "ResetRequest")):
        with pytest.raises(grpc.RpcError) as e:
            getattr(server, rpc)(gs.message(req)())
        assert e.value.code() == grpc.StatusCode.FAILED_PRECONDITION and e.value.details() == "Init not called."
    with pytest.raises(grpc.RpcError) as e:
        server.Init(gs.message("InitRequest")(task_id="No Such Task"))
    assert e.value.code() == grpc.StatusCode.INVALID_ARGUMENT and e.value.details() == "Invalid task_id: 'No Such Task'"
    with pytest.raises(grpc.RpcError) as e:
        server.Init(gs.message("InitRequest")(task_id="Cartpole", model=gs.message("MjModel")(mjb=b"\x00")))
    assert e.value.code() == grpc.StatusCode.INVALID_ARGUMENT


def test_init_without_a_gpu_fails_loudly(server):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_gpu_grpc.py")
    with pytest.raises(grpc.RpcError) as e:
        server.Init(gs.message("InitRequest")(task_id="Cartpole"))
    assert e.value.code() == grpc.StatusCode.INTERNAL and "HIP" in e.value.details()


def test_init_with_an_unloadable_model_xml(server):
    """InitRequest.model.xml goes through this package's MJCF compiler; a document it cannot compile is reported as the
    reference reports a model that fails to load (agent_service.cc:99-104)"""
    with pytest.raises(grpc.RpcError) as e:
        server.Init(gs.message("InitRequest")(task_id="Cartpole", model=gs.message("MjModel")(xml="<mujoco><include file='nope.xml'/></mujoco>")))
    assert e.value.code() == grpc.StatusCode.INTERNAL and e.value.details().startswith("Failed to load model:")
