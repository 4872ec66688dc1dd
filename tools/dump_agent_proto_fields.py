#!/usr/bin/env python3
"""Extracts the wire schema of the reference's mjpc/grpc/agent.proto -- (message, field, number, type, label / oneof / map)
and the service's method list -- into tests/golden/agent_proto_fields.json, which tests/test_grpc_schema.py compares with the
descriptors mujoco_mpc_amd/grpc_service.py builds. Run where /root/reference exists:
    python tools/dump_agent_proto_fields.py /root/reference/mjpc/grpc/agent.proto tests/golden/agent_proto_fields.json"""
import json
import re
import sys

src = re.sub(r"//[^\n]*", "", open(sys.argv[1]).read())
out = {"package": re.search(r"package\s+(\w+);", src).group(1), "messages": {}, "methods": []}
svc = re.search(r"service\s+(\w+)\s*\{(.*?)\n\}", src, re.S)
out["service"] = svc.group(1)
for m in re.finditer(r"rpc\s+(\w+)\s*\(\s*(\w+)\s*\)\s*returns\s*\(\s*(\w+)\s*\)", svc.group(2)):
    out["methods"].append([m.group(1), m.group(2), m.group(3)])
for m in re.finditer(r"message\s+(\w+)\s*\{([^{}]*(?:\{[^{}]*\}[^{}]*)*)\}", src):
    name, body = m.group(1), m.group(2)
    fields = []
    oneof_spans = [(o.start(), o.end(), o.group(1)) for o in re.finditer(r"oneof\s+(\w+)\s*\{.*?\}", body, re.S)]
    for f in re.finditer(r"(optional\s+|repeated\s+)?(map<\s*\w+\s*,\s*\w+\s*>|\w+)\s+(\w+)\s*=\s*(\d+)\s*(\[[^\]]*\])?\s*;", body):
        label, ftype, fname, number, opts = f.groups()
        if ftype in ("oneof", "message", "rpc"):
            continue
        oneof = next((o[2] for o in oneof_spans if o[0] <= f.start() < o[1]), None)
        fields.append({"name": fname, "number": int(number), "type": re.sub(r"\s+", "", ftype), "label": (label or "").strip(),
                       "packed": bool(opts and "packed" in opts and "true" in opts), "oneof": oneof})
    out["messages"][name] = fields
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print(len(out["messages"]), "messages,", len(out["methods"]), "methods")
