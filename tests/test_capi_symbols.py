"""CPU-side checks of the drop-in boundary: libmjpcx.so builds (hipcc cross-compiles gfx950 without a
GPU), loads, and exports every symbol include/mjpcx.h declares; struct layouts agree with the header."""
import ctypes as C
import os
import re

import numpy as np

import pytest

from mujoco_mpc_amd import capi, cstructs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "mjpcx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mjpcx_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = capi.lib()
    declared = header_functions()
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mjpcx.h but not exported"
    assert sorted(capi.EXPORTS) == declared


def test_error_strings():
    lib = capi.lib()
    assert lib.mjpcx_error_string(0) == b"ok"
    for code in (-1, -2, -3, -4, -5):
        assert lib.mjpcx_error_string(code) not in (b"ok", b"unknown error")


def test_struct_layout_matches_header():
    """Field order / count of the ctypes mirrors vs the C structs (parsed from the header)."""
    src = open(os.path.join(ROOT, "include", "mjpcx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)

    def fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), src, re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r"^(const\s+)?(u?int\d+_t|double)\s*\*?\s*", "", decl)
            for n in decl.split(","):
                names.append(re.sub(r"\[.*\]", "", n.strip().lstrip("*").strip()))
        return names

    assert fields("mjpcx_model") == [f[0] for f in cstructs.MjpcxModel._fields_]
    assert fields("mjpcx_task") == [f[0] for f in cstructs.MjpcxTask._fields_]
    assert fields("mjpcx_traj_view") == [f[0] for f in cstructs.MjpcxTrajView._fields_]
    assert fields("mjpcx_noise_spec") == [f[0] for f in cstructs.MjpcxNoiseSpec._fields_]


def test_create_rejects_bad_arguments(cartpole):
    """Argument validation happens before any device call, so it is testable without a GPU."""
    lib = capi.lib()
    pm, pt = cartpole.packed_model(), cartpole.packed()
    h = C.c_void_p()
    assert lib.mjpcx_create(None, pt.ptr, 0, 64, C.byref(h)) == -1
    assert lib.mjpcx_create(pm.ptr, pt.ptr, 0, 16, C.byref(h)) == -1
    assert b"precision" in lib.mjpcx_create_error()


def test_unsupported_models_fail_loudly(cartpole):
    import copy
    lib = capi.lib()
    t = copy.copy(cartpole)
    pm = t.packed_model()
    pm.struct.integrator = 2  # mjINT_IMPLICIT has no device kernel (Euler and RK4 do)
    h = C.c_void_p()
    assert lib.mjpcx_create(pm.ptr, t.packed().ptr, 0, 64, C.byref(h)) == -2
    assert b"Euler" in lib.mjpcx_create_error()
    # an unknown topology names the missing instantiation
    spec = t.spec(); spec["residual_id"] = 99
    from mujoco_mpc_amd.cstructs import PackedTask
    assert lib.mjpcx_create(t.packed_model().ptr, PackedTask(spec).ptr, 0, 64, C.byref(h)) == -2
    assert b"no rollout kernel is instantiated" in lib.mjpcx_create_error()


def test_no_cpu_fallback_in_product():
    """The product package must not import or link the oracle."""
    pkg = os.path.join(ROOT, "mujoco_mpc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".cc")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "liboracle" not in txt and "oracle.h" not in txt, f


@pytest.mark.gpu
def test_moving_geom_pairs_the_kernels_cannot_collide_are_reported():
    """Between two moving bodies the kernels (and the oracle) collide sphere | capsule pairs and (sphere | capsule) x (box | cylinder) pairs.
    Two solids have no narrow phase: such a pair is dropped only if it is proven apart over the joint ranges (csrc/pair_cull.h), otherwise
    it is left out and mjpcx_create says so instead of staying silent. The A1's 210 pairs with a box or a cylinder (the trunk's and the
    hips' against the legs') are all collided or proven apart: no warning, and a strict caller gets its context. Two free boxes can
    touch: reported, and refused under MJPCX_STRICT_PAIRS."""
    from mujoco_mpc_amd import mjcf
    from mujoco_mpc_amd.task import Task, load_task
    quad = load_task("QuadrupedFlat")
    os.environ["MJPCX_STRICT_PAIRS"] = "1"
    try:
        ctx = capi.Context(quad.packed_model(), quad.packed(), 0, 64)
        assert ctx.create_warning == ""
        ctx.close()
        fm = mjcf.load_xml(os.path.join(ROOT, "tests", "models", "two_boxes.xml"))
        task = Task(name="scene", residual_id=0, model=fm).reset()
        with pytest.raises(capi.MjpcxError) as err:
            capi.Context(task.packed_model(), task.packed(), 0, 64)
        assert "NOT collided" in str(err.value)
    finally:
        os.environ.pop("MJPCX_STRICT_PAIRS", None)
    ctx = capi.Context(task.packed_model(), task.packed(), 0, 64)
    assert "NOT collided" in ctx.create_warning and "no narrow phase" in ctx.create_warning
    ctx.close()
    # two solids that already touch at qpos0 (one box resting on the other): every rollout would fail at its first step (warning bit 128), so
    # the model is refused outright with one clear message, strict or not
    import tempfile
    xml = open(os.path.join(ROOT, "tests", "models", "two_boxes.xml")).read().replace('pos="0.5 0 0.05"', 'pos="0.05 0 0.15"')
    with tempfile.NamedTemporaryFile("w", suffix=".xml", delete=False) as f:
        f.write(xml)
    try:
        stacked = Task(name="scene", residual_id=0, model=mjcf.load_xml(f.name)).reset()
    finally:
        os.unlink(f.name)
    with pytest.raises(capi.MjpcxError) as err:
        capi.Context(stacked.packed_model(), stacked.packed(), 0, 64)
    assert "within their contact margin at qpos0" in str(err.value)
    hum = load_task("HumanoidTrack")
    ctx = capi.Context(hum.packed_model(), hum.packed(), 0, 64)
    assert ctx.create_warning == ""   # nothing left uncollided, and the model has a registered kernel configuration (tree_registry.h)
    assert "rollout_limb_kernel" in ctx.kernel_name   # (the limb kernel, with rollout_tree_kernel<Humanoid> behind it for what it hands on)
    ctx.close()


def test_the_a1s_solid_pairs_are_collided_or_proven_apart():
    """csrc/pair_cull.h on the north-star model (CPU build, tests/solidpairs): of the 474 geom pairs MuJoCo's filters leave between moving
    bodies, 264 are sphere | capsule pairs, 201 pair a sphere | capsule with a box or a cylinder (146 proven apart with every joint 0.2 rad past
    its range, 10 -- a calf or foot against the leg's own hip -- only with the knee held to 0.1 rad past its fold limit, 45 -- calves and feet
    against other legs' hips -- not at all: those can touch) and 9 are two cylinders, all proven apart."""
    import ctypes as C
    from collections import Counter
    from mujoco_mpc_amd.task import load_task
    from tests import solidpairs
    quad = load_task("QuadrupedFlat")
    pm = quad.packed_model()
    out = (C.c_int * (6 * 1024))()
    n = solidpairs.lib().sp_moving_pairs(C.cast(pm.ptr, C.c_void_p), out, 1024)
    rows = [tuple(out[6 * i:6 * i + 6]) for i in range(n)]
    kinds = Counter((r[2], r[3], r[4] >= 0) for r in rows)
    assert n == 474 and kinds == {(0, 0, False): 264, (1, 1, False): 146, (1, 1, True): 10, (1, 0, False): 45, (2, 1, False): 9}
    a = quad.model.arrays
    for g1, g2, kind, apart, tj, ts in rows:
        if tj >= 0:   # the tight proofs: same leg, the knee's lower (fold) side
            b1, b2 = int(a["geom_bodyid"][g1]), int(a["geom_bodyid"][g2])
            assert (b1 - 4) // 3 == (b2 - 4) // 3 and ts == 0 and (tj - 1) % 3 == 2
        if kind == 1 and not apart:
            b1, b2 = int(a["geom_bodyid"][g1]), int(a["geom_bodyid"][g2])
            assert (b1 - 4) // 3 != (b2 - 4) // 3 and (b1 - 4) % 3 == 2 and (b2 - 4) % 3 == 0   # a calf body's geom, another leg's hip


def test_registered_tree_configs_match_the_shipped_models():
    """csrc/tree_registry.h fixes the dimensions of the LDS-staged kernel at build time; they must be the shipped A1 model's, or
    the north-star workload silently runs the generic kernel (the smoke test asserts the kernel name on the GPU; this one needs none)"""
    import re
    from mujoco_mpc_amd.task import load_task
    src = open(os.path.join(ROOT, "mujoco_mpc_amd", "csrc", "tree_registry.h")).read()
    body = src[src.index("struct TreeCfgA1"):]
    cfg = {k: int(v) for k, v in re.findall(r"\b(N[A-Z]+) = (\d+)", body[:body.index("};")])}
    t = load_task("QuadrupedFlat")
    m, a = t.model, t.model.arrays
    st = t.packed().struct
    dofs_of_body = np.zeros(m.nbody, int)
    for b in range(1, m.nbody):
        dofs_of_body[b] = dofs_of_body[a["body_parentid"][b]] + a["body_dofnum"][b]
    coll = [g for g in range(m.ngeom) if a["geom_contype"][g] or a["geom_conaffinity"][g]]
    expect = dict(NQ=m.nq, NV=m.nv, NU=m.nu, NB=m.nbody, NJ=m.njnt, NS=m.nsite, NG=m.ngeom, NKEY=len(m.keyframes), NMOCAP=m.nmocap,
                  NSG=sum(dofs_of_body[a["geom_bodyid"][g]] == 0 for g in coll), NDG=sum(dofs_of_body[a["geom_bodyid"][g]] > 0 for g in coll),
                  NRAY=sum(a["geom_group"][g] == 0 and a["geom_type"][g] in (0, 2, 6) for g in range(m.ngeom)),
                  NR=st.num_residual, NTERM=st.num_term, NTRACE=st.num_trace)
    expect.update(NBM=m.nbody, NT=0, NMAX=18)
    assert cfg == {k: int(v) for k, v in expect.items()}, (cfg, expect)
    # the Humanoid of configs[3]: NB / NS are the live prefix (the mocap marker bodies and their sites trail the model), NKEY = 0 (the
    # keyframes are not staged)
    body = src[src.index("struct TreeCfgHumanoid"):]
    cfg = {k: int(v) for k, v in re.findall(r"\b(N[A-Z]+) = (\d+)", body[:body.index("};")])}
    t = load_task("HumanoidTrack")
    m, a = t.model, t.model.arrays
    st = t.packed().struct
    dofs_of_body = np.zeros(m.nbody, int)
    for b in range(1, m.nbody):
        dofs_of_body[b] = dofs_of_body[a["body_parentid"][b]] + a["body_dofnum"][b]
    live = 1 + max(b for b in range(m.nbody) if dofs_of_body[b] > 0)
    coll = [g for g in range(m.ngeom) if a["geom_contype"][g] or a["geom_conaffinity"][g]]
    assert cfg["NQ"] == m.nq and cfg["NV"] == m.nv and cfg["NU"] == m.nu and cfg["NJ"] == m.njnt and cfg["NG"] == m.ngeom and cfg["NMOCAP"] == m.nmocap
    assert cfg["NBM"] == m.nbody and live <= cfg["NB"] <= m.nbody and cfg["NS"] <= m.nsite and cfg["NKEY"] == 0 and cfg["NMAX"] >= m.nv
    assert cfg["NSG"] == sum(dofs_of_body[a["geom_bodyid"][g]] == 0 for g in coll) and cfg["NDG"] == sum(dofs_of_body[a["geom_bodyid"][g]] > 0 for g in coll)
    assert (cfg["NR"], cfg["NTERM"], cfg["NTRACE"]) == (st.num_residual, st.num_term, st.num_trace) and cfg["NT"] == int(np.sum(a["tendon_limited"] != 0))
