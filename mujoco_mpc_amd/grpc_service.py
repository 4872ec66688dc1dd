"""gRPC `agent.Agent` service (mjpc/grpc/agent.proto, agent_service.cc, agent_server.cc) in front of the GPU planners.

The wire schema is restated here as protobuf descriptors (package `agent`, service `Agent`; the same message names, field
names, numbers, types, labels, oneofs and map fields as agent.proto:19-226), so a client generated from the reference's proto
-- mujoco_mpc.Agent in python/mujoco_mpc/agent.py, or any other language's stub -- talks to this server unchanged. Each handler
is one call into the transport-free C++ service core (host/mjpc/grpc/agent_service.{h,cc} through host/agent_c_api.cc), which
mirrors agent_service.cc / grpc_agent_util.cc on top of mjpc::Agent. This image has grpcio for Python and no C++ gRPC, hence
the split. No CPU fallback: without libmjpcx.so / a GPU, Init fails with the library's error.
"""
import ctypes as C
import os
import tempfile
from concurrent import futures

import grpc
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_T = descriptor_pb2.FieldDescriptorProto
_OPT, _REP = _T.LABEL_OPTIONAL, _T.LABEL_REPEATED
_STR, _DBL = _T.TYPE_STRING, _T.TYPE_DOUBLE

# message -> [(field name, number, type, label, extra)]. extra: None | "packed" | "optional" (proto3 explicit presence) |
# ("msg", message) | ("optmsg", message) | ("map", key type, value type or message) | ("oneof", oneof name)
SCHEMA = {
    "MjModel": [("mjb", 1, _T.TYPE_BYTES, _OPT, "optional"), ("xml", 2, _STR, _OPT, "optional")],
    "InitRequest": [("task_id", 1, _STR, _OPT, "optional"), ("model", 2, _T.TYPE_MESSAGE, _OPT, ("optmsg", "MjModel")),
                    ("real_time_speed", 3, _T.TYPE_FLOAT, _OPT, "optional")],
    "InitResponse": [],
    "State": [("time", 1, _DBL, _OPT, "optional"), ("qpos", 2, _DBL, _REP, "packed"), ("qvel", 3, _DBL, _REP, "packed"),
              ("act", 4, _DBL, _REP, "packed"), ("mocap_pos", 5, _DBL, _REP, "packed"), ("mocap_quat", 6, _DBL, _REP, "packed"),
              ("userdata", 7, _DBL, _REP, "packed")],
    "GetStateRequest": [],
    "GetStateResponse": [("state", 1, _T.TYPE_MESSAGE, _OPT, ("msg", "State"))],
    "SetStateRequest": [("state", 1, _T.TYPE_MESSAGE, _OPT, ("msg", "State"))],
    "SetStateResponse": [],
    "GetActionRequest": [("time", 1, _T.TYPE_FLOAT, _OPT, "optional"), ("averaging_duration", 2, _T.TYPE_FLOAT, _OPT, "optional"),
                         ("nominal_action", 3, _T.TYPE_BOOL, _OPT, "optional")],
    "GetActionResponse": [("action", 1, _T.TYPE_FLOAT, _REP, "packed")],
    "GetResidualsRequest": [],
    "Residual": [("values", 1, _DBL, _REP, None)],
    "GetResidualsResponse": [("values", 1, None, _REP, ("map", _STR, "Residual"))],
    "GetCostValuesAndWeightsRequest": [],
    "ValueAndWeight": [("value", 1, _DBL, _OPT, None), ("weight", 2, _DBL, _OPT, None)],
    "GetCostValuesAndWeightsResponse": [("values_weights", 1, None, _REP, ("map", _STR, "ValueAndWeight"))],
    "PlannerStepRequest": [], "PlannerStepResponse": [],
    "StepRequest": [("use_previous_policy", 1, _T.TYPE_BOOL, _OPT, None)], "StepResponse": [],
    "ResetRequest": [], "ResetResponse": [],
    "TaskParameterValue": [("numeric", 1, _DBL, _OPT, ("oneof", "value")), ("selection", 2, _STR, _OPT, ("oneof", "value"))],
    "SetTaskParametersRequest": [("parameters", 1, None, _REP, ("map", _STR, "TaskParameterValue"))],
    "SetTaskParametersResponse": [],
    "GetTaskParametersRequest": [],
    "GetTaskParametersResponse": [("parameters", 1, None, _REP, ("map", _STR, "TaskParameterValue"))],
    "SetCostWeightsRequest": [("reset_to_defaults", 1, _T.TYPE_BOOL, _OPT, None), ("cost_weights", 2, None, _REP, ("map", _STR, _DBL))],
    "SetCostWeightsResponse": [],
    "GetModeRequest": [], "GetModeResponse": [("mode", 1, _STR, _OPT, None)],
    "SetModeRequest": [("mode", 1, _STR, _OPT, None)], "SetModeResponse": [],
    "GetAllModesRequest": [], "GetAllModesResponse": [("mode_names", 1, _STR, _REP, None)],
    "GetBestTrajectoryRequest": [],
    "GetBestTrajectoryResponse": [("states", 1, _DBL, _REP, "packed"), ("actions", 2, _DBL, _REP, "packed"),
                                  ("times", 3, _DBL, _REP, "packed"), ("steps", 4, _T.TYPE_INT32, _OPT, None)],
    "Pose": [("pos", 1, _DBL, _REP, "packed"), ("quat", 2, _DBL, _REP, "packed")],
    "SetAnythingRequest": [("state", 1, _T.TYPE_MESSAGE, _OPT, ("msg", "State")),
                           ("parameters", 2, None, _REP, ("map", _STR, "TaskParameterValue")),
                           ("cost_weights", 3, None, _REP, ("map", _STR, _DBL)), ("mode", 4, _STR, _OPT, None),
                           ("mocap", 5, None, _REP, ("map", _STR, "Pose"))],
    "SetAnythingResponse": [],
}
# agent.proto:19-59
METHODS = ["Init", "GetState", "SetState", "GetAction", "PlannerStep", "Step", "Reset", "SetTaskParameters", "GetTaskParameters",
           "SetCostWeights", "GetResiduals", "GetCostValuesAndWeights", "SetMode", "GetMode", "GetAllModes", "GetBestTrajectory",
           "SetAnything"]


def _camel(name):
    return "".join(p.capitalize() for p in name.split("_"))


def file_descriptor_proto():
    fd = descriptor_pb2.FileDescriptorProto(name="mjpc/grpc/agent.proto", package="agent", syntax="proto3")
    for mname, fields in SCHEMA.items():
        msg = fd.message_type.add(name=mname)
        explicit = []
        for fname, number, ftype, label, extra in fields:
            f = msg.field.add(name=fname, number=number, label=label)
            kind = extra[0] if isinstance(extra, tuple) else extra
            if kind == "map":
                entry = msg.nested_type.add(name=_camel(fname) + "Entry")
                entry.options.map_entry = True
                entry.field.add(name="key", number=1, label=_OPT, type=extra[1])
                v = entry.field.add(name="value", number=2, label=_OPT)
                if isinstance(extra[2], str):
                    v.type, v.type_name = _T.TYPE_MESSAGE, ".agent." + extra[2]
                else:
                    v.type = extra[2]
                f.type, f.type_name = _T.TYPE_MESSAGE, f".agent.{mname}.{entry.name}"
                continue
            f.type = ftype
            if kind in ("msg", "optmsg"):
                f.type_name = ".agent." + extra[1]
            if kind == "oneof":
                names = [o.name for o in msg.oneof_decl]
                if extra[1] not in names:
                    msg.oneof_decl.add(name=extra[1])
                    names.append(extra[1])
                f.oneof_index = names.index(extra[1])
            if kind == "packed":
                f.options.packed = True
            if kind in ("optional", "optmsg"):
                explicit.append(f)
        for f in explicit:  # proto3 `optional`: a synthetic one-field oneof, declared after the real oneofs
            msg.oneof_decl.add(name="_" + f.name)
            f.oneof_index = len(msg.oneof_decl) - 1
            f.proto3_optional = True
    svc = fd.service.add(name="Agent")
    for m in METHODS:
        svc.method.add(name=m, input_type=f".agent.{m}Request", output_type=f".agent.{m}Response")
    return fd


_pool = descriptor_pool.DescriptorPool()
_pool.AddSerializedFile(file_descriptor_proto().SerializeToString())
_classes = {}


def message(name):
    """the protobuf class of agent.<name>"""
    if name not in _classes:
        _classes[name] = message_factory.GetMessageClass(_pool.FindMessageTypeByName("agent." + name))
    return _classes[name]


# ----------------------------------------------------------------------------------------------------------------- C side
_HOST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "build", "libmjpc_host.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        from . import capi
        capi.lib()  # libmjpcx.so first, then the host layer that links it
        if not os.path.exists(_HOST):
            raise RuntimeError(f"{_HOST} is missing: run __graft_entry__.build()")
        L = C.CDLL(_HOST, mode=C.RTLD_GLOBAL)
        L.mjpc_agent_service_create.restype = C.c_void_p
        L.mjpc_agent_service_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.mjpc_agent_service_destroy.argtypes = [C.c_void_p]
        L.mjpc_agent_service_error.restype = C.c_char_p
        L.mjpc_agent_service_error.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def write_task_blobs(directory):
    """compiles every registered task model into <directory>/<TaskName>.mjpx (what AgentService::Init loads)"""
    from . import mjcf
    from .task import load_task, task_names
    for name in task_names():
        mjcf.save_blob(load_task(name).model, os.path.join(directory, f"{name}.mjpx"))
    return directory


_CODES = {3: grpc.StatusCode.INVALID_ARGUMENT, 9: grpc.StatusCode.FAILED_PRECONDITION, 13: grpc.StatusCode.INTERNAL}
_SIZE_NAMES = ("nq", "nv", "na", "nmocap", "nuserdata", "nu", "num_term", "num_residual", "nparam", "steps")
_STATE_FIELDS = ("qpos", "qvel", "act", "mocap_pos", "mocap_quat", "userdata")


def _arr(values):
    return (C.c_double * max(len(values), 1))(*values), len(values)


class AgentServicer:
    """One handler per RPC of agent.proto; `model_dir` holds the compiled task models (<TaskName>.mjpx, write_task_blobs)."""

    def __init__(self, model_dir, device=0, precision=64, num_candidates=0):
        self.L = lib()
        self.model_dir, self.device, self.precision, self.num_candidates = model_dir, device, precision, num_candidates
        self.h = None
        self._tmp = None
        self._create(model_dir)

    def _create(self, model_dir):
        self.close()
        self.h = C.c_void_p(self.L.mjpc_agent_service_create(model_dir.encode(), self.device, self.precision, self.num_candidates))
        if not self.h:
            raise RuntimeError("mjpc_agent_service_create failed")

    def close(self):
        if self.h:
            self.L.mjpc_agent_service_destroy(self.h)
            self.h = None

    # -- helpers
    def _check(self, rc, context):
        if rc != 0:
            context.abort(_CODES.get(rc, grpc.StatusCode.UNKNOWN), self.L.mjpc_agent_service_error(self.h).decode())

    def _sizes(self, context):
        s = (C.c_int * 10)()
        self._check(self.L.mjpc_agent_service_sizes(self.h, s), context)
        return dict(zip(_SIZE_NAMES, s))

    @staticmethod
    def _state_args(st):
        args = [int(st.HasField("time")), C.c_double(st.time)]
        for f in _STATE_FIELDS:
            args += list(_arr(list(getattr(st, f))))
        return args

    # -- RPCs
    def Init(self, request, context):
        model_dir = self.model_dir
        if request.HasField("model"):
            # InitRequest.model (grpc_agent_util.cc:535-560): an XML is compiled by this package's model compiler; a binary
            # .mjb is MuJoCo's own serialisation, which this build cannot read
            if request.model.HasField("mjb"):
                context.abort(grpc.StatusCode.INVALID_ARGUMENT, "InitRequest.model.mjb is not supported: send the MJCF in model.xml")
            if request.model.HasField("xml"):
                from . import mjcf
                from .task import task_names
                # the task id names the files written below: only a registered task name is accepted (never a path --
                # "../x" or an absolute path would let a client choose where the server writes), and <include> is resolved
                # inside the temporary directory only
                name = request.task_id.replace(" ", "")
                if name not in {t.replace(" ", "") for t in task_names()} or os.path.basename(name) != name:
                    context.abort(grpc.StatusCode.INVALID_ARGUMENT, f"Invalid task_id: '{request.task_id}'")
                self._tmp = tempfile.TemporaryDirectory(prefix="mjpc_grpc_")
                xml_path = os.path.join(self._tmp.name, name + ".xml")
                with open(xml_path, "w") as f:
                    f.write(request.model.xml)
                try:
                    mjcf.save_blob(mjcf.load_xml(xml_path, include_root=self._tmp.name), os.path.join(self._tmp.name, name + ".mjpx"))
                except Exception as e:  # noqa: BLE001 -- reported to the client as the reference reports a load error
                    context.abort(grpc.StatusCode.INTERNAL, f"Failed to load model: {e}")
                model_dir = self._tmp.name
        self._create(model_dir)  # the reference allows one Init per service instance; here a later Init starts afresh
        self._check(self.L.mjpc_agent_service_init(self.h, request.task_id.encode()), context)
        return message("InitResponse")()

    def GetState(self, request, context):
        z = self._sizes(context)
        counts = (z["nq"], z["nv"], z["na"], 3 * z["nmocap"], 4 * z["nmocap"], z["nuserdata"])
        t = C.c_double()
        bufs = [(C.c_double * max(n, 1))() for n in counts]
        self._check(self.L.mjpc_agent_service_get_state(self.h, C.byref(t), *bufs), context)
        resp = message("GetStateResponse")()
        resp.state.time = t.value
        for f, b, n in zip(_STATE_FIELDS, bufs, counts):
            getattr(resp.state, f).extend(list(b)[:n])
        return resp

    def SetState(self, request, context):
        self._check(self.L.mjpc_agent_service_set_state(self.h, *self._state_args(request.state)), context)
        return message("SetStateResponse")()

    def GetAction(self, request, context):
        z = self._sizes(context)
        a = (C.c_double * max(z["nu"], 1))()
        self._check(self.L.mjpc_agent_service_get_action(self.h, int(request.HasField("time")), C.c_double(request.time),
                                                         C.c_double(request.averaging_duration), int(request.nominal_action), a), context)
        resp = message("GetActionResponse")()
        resp.action.extend(list(a)[:z["nu"]])
        return resp

    def PlannerStep(self, request, context):
        self._check(self.L.mjpc_agent_service_planner_step(self.h), context)
        return message("PlannerStepResponse")()

    def Step(self, request, context):
        self._check(self.L.mjpc_agent_service_step(self.h, int(request.use_previous_policy)), context)
        return message("StepResponse")()

    def Reset(self, request, context):
        self._check(self.L.mjpc_agent_service_reset(self.h), context)
        return message("ResetResponse")()

    def SetTaskParameters(self, request, context):
        self._sizes(context)
        for name, value in request.parameters.items():
            which = value.WhichOneof("value")
            if which is None:
                context.abort(grpc.StatusCode.INVALID_ARGUMENT, f"Missing value for parameter {name}")
            self._check(self.L.mjpc_agent_service_set_task_parameter(self.h, name.encode(), int(which == "selection"),
                                                                     C.c_double(value.numeric), value.selection.encode()), context)
        return message("SetTaskParametersResponse")()

    def GetTaskParameters(self, request, context):
        n = C.c_int()
        self._check(self.L.mjpc_agent_service_get_task_parameters(self.h, C.byref(n)), context)
        resp = message("GetTaskParametersResponse")()
        name, sel = C.create_string_buffer(256), C.create_string_buffer(256)
        is_sel, num = C.c_int(), C.c_double()
        for i in range(n.value):
            self._check(self.L.mjpc_agent_service_task_parameter_at(self.h, i, name, 256, C.byref(is_sel), C.byref(num), sel, 256), context)
            if is_sel.value:
                resp.parameters[name.value.decode()].selection = sel.value.decode()
            else:
                resp.parameters[name.value.decode()].numeric = num.value
        return resp

    def SetCostWeights(self, request, context):
        self._sizes(context)
        if request.reset_to_defaults:
            self._check(self.L.mjpc_agent_service_reset_cost_weights(self.h), context)
        for name, w in request.cost_weights.items():
            self._check(self.L.mjpc_agent_service_set_cost_weight(self.h, name.encode(), C.c_double(w)), context)
        return message("SetCostWeightsResponse")()

    def _cost_terms(self, context):
        z = self._sizes(context)
        n = C.c_int()
        self._check(self.L.mjpc_agent_service_get_cost_terms(self.h, C.byref(n)), context)
        out = []
        name = C.create_string_buffer(256)
        value, weight, dim = C.c_double(), C.c_double(), C.c_int()
        res = (C.c_double * max(z["num_residual"], 1))()
        for i in range(n.value):
            self._check(self.L.mjpc_agent_service_cost_term_at(self.h, i, name, 256, C.byref(value), C.byref(weight), res,
                                                               z["num_residual"], C.byref(dim)), context)
            out.append((name.value.decode(), value.value, weight.value, list(res)[:dim.value]))
        return out

    def GetResiduals(self, request, context):
        resp = message("GetResidualsResponse")()
        for name, _v, _w, residual in self._cost_terms(context):
            resp.values[name].values.extend(residual)
        return resp

    def GetCostValuesAndWeights(self, request, context):
        resp = message("GetCostValuesAndWeightsResponse")()
        for name, v, w, _r in self._cost_terms(context):
            resp.values_weights[name].value = v
            resp.values_weights[name].weight = w
        return resp

    def SetMode(self, request, context):
        self._check(self.L.mjpc_agent_service_set_mode(self.h, request.mode.encode()), context)
        return message("SetModeResponse")()

    def GetMode(self, request, context):
        buf = C.create_string_buffer(256)
        self._check(self.L.mjpc_agent_service_get_mode(self.h, buf, 256), context)
        return message("GetModeResponse")(mode=buf.value.decode())

    def GetAllModes(self, request, context):
        buf = C.create_string_buffer(4096)
        self._check(self.L.mjpc_agent_service_get_all_modes(self.h, buf, 4096), context)
        resp = message("GetAllModesResponse")()
        resp.mode_names.extend([m for m in buf.value.decode().split("|") if m])
        return resp

    def GetBestTrajectory(self, request, context):
        z = self._sizes(context)
        T, ds, nu = z["steps"], z["nq"] + z["nv"] + z["na"], z["nu"]
        na = max((T - 1) * nu, 1)
        s, a, t = (C.c_double * (T * ds))(), (C.c_double * na)(), (C.c_double * T)()
        steps = C.c_int()
        self._check(self.L.mjpc_agent_service_best_trajectory(self.h, s, T * ds, a, na, t, T, C.byref(steps)), context)
        resp = message("GetBestTrajectoryResponse")(steps=steps.value)
        resp.states.extend(list(s)[:steps.value * ds])
        resp.actions.extend(list(a)[:(steps.value - 1) * nu])
        resp.times.extend(list(t)[:steps.value])
        return resp

    def SetAnything(self, request, context):
        # grpc_agent_util.cc:497-520: state (no Transition), cost weights, mode, mocap poses -- `parameters` is not read there either
        self._sizes(context)
        if request.HasField("state"):
            self._check(self.L.mjpc_agent_service_set_anything_state(self.h, *self._state_args(request.state)), context)
        for name, w in request.cost_weights.items():
            self._check(self.L.mjpc_agent_service_set_cost_weight(self.h, name.encode(), C.c_double(w)), context)
        if request.mode:
            self._check(self.L.mjpc_agent_service_set_mode(self.h, request.mode.encode()), context)
        for name, pose in request.mocap.items():
            p, n_p = _arr(list(pose.pos))
            q, n_q = _arr(list(pose.quat))
            self._check(self.L.mjpc_agent_service_set_mocap(self.h, name.encode(), p, n_p, q, n_q), context)
        return message("SetAnythingResponse")()


def generic_handler(servicer):
    handlers = {m: grpc.unary_unary_rpc_method_handler(getattr(servicer, m), request_deserializer=message(m + "Request").FromString,
                                                       response_serializer=lambda r: r.SerializeToString()) for m in METHODS}
    return grpc.method_handlers_generic_handler("agent.Agent", handlers)


def serve(model_dir, port=0, device=0, precision=64, num_candidates=0, host="127.0.0.1"):
    """starts the server (one worker thread: the reference's service instance is single-threaded); returns
    (server, bound port, servicer). mjpc/grpc/agent_server.cc."""
    servicer = AgentServicer(model_dir, device, precision, num_candidates)
    server = grpc.server(futures.ThreadPoolExecutor(max_workers=1))
    server.add_generic_rpc_handlers((generic_handler(servicer),))
    bound = server.add_insecure_port(f"{host}:{port}")
    server.start()
    return server, bound, servicer


class AgentStub:
    """what agent_pb2_grpc.AgentStub is for agent.proto: one unary-unary callable per RPC (used by the tests as the client)"""

    def __init__(self, channel):
        for m in METHODS:
            setattr(self, m, channel.unary_unary(f"/agent.Agent/{m}", request_serializer=lambda r: r.SerializeToString(),
                                                 response_deserializer=message(m + "Response").FromString))


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="MJPC agent gRPC server on the GPU planners (mjpc/grpc/agent_server.cc)")
    ap.add_argument("--mjpc_port", type=int, default=10000)
    ap.add_argument("--mjpc_workers", type=int, default=-1, help="accepted for interface parity; rollouts run on the GPU")
    ap.add_argument("--model_dir", default=None, help="directory of compiled task models (default: compiled at start-up)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--precision", type=int, default=64)
    a = ap.parse_args(argv)
    model_dir = a.model_dir or write_task_blobs(tempfile.mkdtemp(prefix="mjpc_models_"))
    server, port, _ = serve(model_dir, a.mjpc_port, a.device, a.precision)
    print(f"Server listening on 127.0.0.1:{port}", flush=True)
    server.wait_for_termination()


if __name__ == "__main__":
    main()
