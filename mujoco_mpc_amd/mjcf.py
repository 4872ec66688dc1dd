"""MJCF-subset compiler: task XML -> flat model (the `mjpcx_model` arrays).

The reference loads its task files with MuJoCo's `mj_loadXML`
(mjpc/testspeed.cc:54-68); MuJoCo is a third-party dependency that is not in
the reference tree nor in this image, so this module compiles
the subset of MJCF that the hot-path models use into the flat arrays consumed
by the C ABI (include/mjpcx.h). It follows MuJoCo's documented compilation
rules (defaults classes, childclass, degree angles, fromto geoms, inertia from
geoms, qpos0/ref, `mj_setConst` for dof_invweight0 / meaninertia).

Supported: include, compiler(angle, eulerseq, autolimits, inertiafromgeom),
option(+flag), default classes, body/inertial/joint/freejoint/geom/site,
motor/general/position actuators on joints, custom numeric/text, sensors
(user + named frame sensors as metadata), keyframes. Anything touching
contacts, tendons or equality constraints is parsed only as far as inertia
needs it.
"""
from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
_JNT_TYPES = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}

DSBL = {  # mjtDisableBit
    "constraint": 1 << 0, "equality": 1 << 1, "frictionloss": 1 << 2, "limit": 1 << 3,
    "contact": 1 << 4, "passive": 1 << 5, "gravity": 1 << 6, "clampctrl": 1 << 7,
    "warmstart": 1 << 8, "filterparent": 1 << 9, "actuation": 1 << 10, "refsafe": 1 << 11,
    "sensor": 1 << 12, "midphase": 1 << 13, "eulerdamp": 1 << 14,
}

DEFAULT_SOLREF = (0.02, 1.0)
DEFAULT_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)


def _floats(s, n=None):
    v = [float(x) for x in s.split()]
    if n is not None and len(v) != n:
        raise ValueError(f"expected {n} numbers, got {s!r}")
    return v


# ----------------------------------------------------------------------------- math
def quat_mul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def mat_to_quat(R):
    # robust rotation matrix -> unit quaternion (w,x,y,z)
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    q = np.array(q)
    return q / np.linalg.norm(q)


def axis_angle_quat(axis, angle):
    axis = np.asarray(axis, float)
    n = np.linalg.norm(axis)
    if n < 1e-15:
        return np.array([1.0, 0, 0, 0])
    s = math.sin(angle / 2)
    return np.concatenate([[math.cos(angle / 2)], axis / n * s])


def z_to_quat(vec):
    """quaternion rotating the z axis onto `vec` (mjuu_z2quat)."""
    v = np.asarray(vec, float)
    v = v / np.linalg.norm(v)
    z = np.array([0.0, 0, 1])
    ax = np.cross(z, v)
    s = np.linalg.norm(ax)
    if s < 1e-10:
        return np.array([1.0, 0, 0, 0]) if v[2] > 0 else np.array([0.0, 1, 0, 0])
    ang = math.atan2(s, v[2])
    return axis_angle_quat(ax / s, ang)


# ----------------------------------------------------------------------------- model
@dataclass
class FlatModel:
    """Compiled model: numpy arrays named as in include/mjpcx.h + metadata."""
    arrays: dict
    scalars: dict
    names: dict          # kind -> list of names (body, joint, site, actuator, sensor)
    numeric: dict        # custom numeric name -> np.ndarray
    text: dict
    sensors: list        # dicts: name,type,dim,user,objtype,objname
    keyframes: dict      # name -> dict(qpos,qvel,ctrl,mpos,mquat)
    nuser_sensor: int = 0
    source: str = ""

    def __getattr__(self, k):
        if k in ("arrays", "scalars"):
            raise AttributeError(k)
        if k in self.scalars:
            return self.scalars[k]
        if k in self.arrays:
            return self.arrays[k]
        raise AttributeError(k)

    def name2id(self, kind, name):
        try:
            return self.names[kind].index(name)
        except ValueError:
            return -1

    def get_number(self, name, default=None):
        """GetNumberOrDefault, mjpc/utilities.h:53-68."""
        if name in self.numeric:
            return float(self.numeric[name][0])
        return default

    def key_qpos(self, name):
        return self.keyframes[name]["qpos"]

    def dim_state(self):
        return self.nq + self.nv + self.na


class _Defaults:
    """MJCF default classes: tag -> attribute dict, with inheritance."""

    def __init__(self):
        self.classes = {"main": {}}
        self.parent = {"main": None}

    def add(self, elem, parent_class):
        name = "main" if parent_class is None else elem.get("class")
        cls = self.classes.setdefault(name, {})
        if name != "main":
            self.parent[name] = parent_class
        for child in elem:
            if child.tag == "default":
                self.add(child, name)
            else:
                cls.setdefault(child.tag, {}).update(child.attrib)

    def resolve(self, tag, class_name):
        chain = []
        c = class_name or "main"
        while c is not None:
            chain.append(c)
            c = self.parent.get(c)
        out = {}
        for c in reversed(chain):
            out.update(self.classes.get(c, {}).get(tag, {}))
        return out


def _expand_includes(elem, base_dir, include_root=None):
    """include_root: when given, every included file must resolve inside that directory (models sent by a client)"""
    i = 0
    children = list(elem)
    for child in children:
        if child.tag == "include":
            path = os.path.join(base_dir, child.get("file"))
            if include_root is not None:
                real, root = os.path.realpath(path), os.path.realpath(include_root)
                if os.path.commonpath([real, root]) != root:
                    raise ValueError(f"<include file='{child.get('file')}'> resolves outside the model directory")
            sub = ET.parse(path).getroot()
            _expand_includes(sub, os.path.dirname(path), include_root)
            idx = list(elem).index(child)
            elem.remove(child)
            for k, sc in enumerate(list(sub)):
                elem.insert(idx + k, sc)
        else:
            _expand_includes(child, base_dir, include_root)
        i += 1


class _Compiler:
    def __init__(self, root, source):
        self.root = root
        self.source = source
        self.defaults = _Defaults()
        comp = {}
        for c in root.findall("compiler"):
            comp.update(c.attrib)
        self.degree = comp.get("angle", "degree") == "degree"
        self.eulerseq = comp.get("eulerseq", "xyz")
        self.autolimits = comp.get("autolimits", "true") == "true"
        self.inertiafromgeom = comp.get("inertiafromgeom", "auto")
        self.settotalmass = float(comp.get("settotalmass", -1))
        for d in root.findall("default"):
            self.defaults.add(d, None)
        self.bodies, self.joints, self.sites, self.geoms = [], [], [], []

    # ---- orientation of a frame-like element
    def _orient(self, a):
        if "quat" in a:
            q = np.array(_floats(a["quat"], 4))
            return q / np.linalg.norm(q)
        if "axisangle" in a:
            v = _floats(a["axisangle"], 4)
            ang = math.radians(v[3]) if self.degree else v[3]
            return axis_angle_quat(v[:3], ang)
        if "euler" in a:
            e = _floats(a["euler"], 3)
            if self.degree:
                e = [math.radians(x) for x in e]
            q = np.array([1.0, 0, 0, 0])
            for ch, ang in zip(self.eulerseq, e):
                ax = {"x": [1, 0, 0], "y": [0, 1, 0], "z": [0, 0, 1]}[ch.lower()]
                r = axis_angle_quat(ax, ang)
                q = quat_mul(q, r) if ch.islower() else quat_mul(r, q)
            return q
        if "xyaxes" in a:
            v = _floats(a["xyaxes"], 6)
            x = np.array(v[:3]); x /= np.linalg.norm(x)
            y = np.array(v[3:]); y -= x * np.dot(x, y); y /= np.linalg.norm(y)
            z = np.cross(x, y)
            return mat_to_quat(np.stack([x, y, z], axis=1))
        if "zaxis" in a:
            return z_to_quat(_floats(a["zaxis"], 3))
        return np.array([1.0, 0, 0, 0])

    # ---- geoms: frame, size, mass, inertia
    def _geom(self, elem, childclass):
        a = self.defaults.resolve("geom", elem.get("class", childclass))
        a.update(elem.attrib)
        gtype = a.get("type", "sphere")
        size = _floats(a.get("size", "0 0 0"))
        size = size + [0.0] * (3 - len(size))
        pos = np.array(_floats(a.get("pos", "0 0 0"), 3))
        quat = self._orient(a)
        if "fromto" in a:
            ft = np.array(_floats(a["fromto"], 6))
            p0, p1 = ft[:3], ft[3:]
            pos = 0.5 * (p0 + p1)
            quat = z_to_quat(p1 - p0)
            half = 0.5 * np.linalg.norm(p1 - p0)
            if gtype in ("capsule", "cylinder"):
                size = [size[0], half, 0.0]
            elif gtype in ("box", "ellipsoid"):
                size = [size[0], size[0], half]
        g = dict(name=a.get("name", ""), type=gtype, size=np.array(size), pos=pos, quat=quat,
                 density=float(a.get("density", 1000.0)), mass=float(a["mass"]) if "mass" in a else None,
                 attrib=a)
        fr = _floats(a.get("friction", "1 0.005 0.0001"))
        fr = fr + [0.005, 0.0001][len(fr) - 1:] if len(fr) < 3 else fr
        simp = list(DEFAULT_SOLIMP)
        if "solimp" in a:
            v = _floats(a["solimp"]); simp[:len(v)] = v
        g.update(contype=int(a.get("contype", 1)), conaffinity=int(a.get("conaffinity", 1)), condim=int(a.get("condim", 3)),
                 priority=int(a.get("priority", 0)), group=int(a.get("group", 0)), friction=np.array(fr[:3]),
                 solref=np.array(_floats(a.get("solref", "0.02 1"), 2)), solimp=np.array(simp),
                 margin=float(a.get("margin", 0)), gap=float(a.get("gap", 0)), solmix=float(a.get("solmix", 1)))
        vol, inertia_unit = _geom_volume_inertia(gtype, g["size"])
        if g["mass"] is None:
            g["mass"] = g["density"] * vol
        g["inertia"] = inertia_unit(g["mass"]) if vol > 0 else np.zeros(3)
        if vol <= 0:
            g["mass"] = 0.0
        return g

    def _body(self, elem, parent_id, childclass):
        bid = len(self.bodies)
        a = elem.attrib
        childclass = a.get("childclass", childclass)
        body = dict(name=a.get("name", ""), parent=parent_id,
                    pos=np.array(_floats(a.get("pos", "0 0 0"), 3)), quat=self._orient(a),
                    mocap=a.get("mocap", "false") == "true", joints=[], inertial=None, geoms=[])
        self.bodies.append(body)
        for child in elem:
            if child.tag == "inertial":
                ia = child.attrib
                inert = dict(pos=np.array(_floats(ia.get("pos", "0 0 0"), 3)), quat=self._orient(ia),
                             mass=float(ia["mass"]))
                if "fullinertia" in ia:
                    f = _floats(ia["fullinertia"], 6)
                    I = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
                    w, V = np.linalg.eigh(I)
                    order = np.argsort(-w)
                    w, V = w[order], V[:, order]
                    if np.linalg.det(V) < 0:
                        V[:, 2] = -V[:, 2]
                    inert["inertia"] = w
                    inert["quat"] = quat_mul(inert["quat"], mat_to_quat(V))
                else:
                    inert["inertia"] = np.array(_floats(ia["diaginertia"], 3))
                body["inertial"] = inert
            elif child.tag in ("joint", "freejoint"):
                ja = {} if child.tag == "freejoint" else self.defaults.resolve("joint", child.get("class", childclass))
                ja.update(child.attrib)
                if child.tag == "freejoint":
                    ja["type"] = "free"
                self.joints.append(dict(body=bid, attrib=ja))
                body["joints"].append(len(self.joints) - 1)
            elif child.tag == "geom":
                g = self._geom(child, childclass)
                if g["type"] != "mesh":  # mesh assets are not available; such geoms are visual-only in the models used
                    body["geoms"].append(g)
            elif child.tag == "site":
                sa = self.defaults.resolve("site", child.get("class", childclass))
                sa.update(child.attrib)
                self.sites.append(dict(name=sa.get("name", ""), body=bid,
                                       pos=np.array(_floats(sa.get("pos", "0 0 0"), 3)), quat=self._orient(sa)))
        for child in elem:
            if child.tag == "body":
                self._body(child, bid, childclass)
        return bid

    def compile(self) -> FlatModel:
        root = self.root
        # ---- options
        opt = {}
        flags = {}
        for o in root.findall("option"):
            opt.update(o.attrib)
            for f in o.findall("flag"):
                flags.update(f.attrib)
        disable = 0
        for k, v in flags.items():
            if k in DSBL and v == "disable":
                disable |= DSBL[k]
        integrator = {"Euler": 0, "RK4": 1, "implicit": 2, "implicitfast": 3}[opt.get("integrator", "Euler")]
        scal = dict(timestep=float(opt.get("timestep", 0.002)),
                    gravity=np.array(_floats(opt.get("gravity", "0 0 -9.81"), 3)),
                    integrator=integrator, disableflags=disable,
                    solver_iterations=int(opt.get("iterations", 100)),
                    solver_tolerance=float(opt.get("tolerance", 1e-8)),
                    cone={"pyramidal": 0, "elliptic": 1}[opt.get("cone", "pyramidal")],
                    impratio=float(opt.get("impratio", 1.0)))

        # ---- body tree (world = body 0)
        world = dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), mocap=False,
                     joints=[], inertial=None, geoms=[])
        self.bodies.append(world)
        for wb in root.findall("worldbody"):
            for child in wb:
                if child.tag == "body":
                    self._body(child, 0, None)
                elif child.tag == "geom":
                    world["geoms"].append(self._geom(child, None))
                elif child.tag == "site":
                    sa = self.defaults.resolve("site", child.get("class"))
                    sa.update(child.attrib)
                    self.sites.append(dict(name=sa.get("name", ""), body=0,
                                           pos=np.array(_floats(sa.get("pos", "0 0 0"), 3)), quat=self._orient(sa)))
        nb = len(self.bodies)

        # ---- inertial properties
        body_mass = np.zeros(nb); body_inertia = np.zeros((nb, 3))
        body_ipos = np.zeros((nb, 3)); body_iquat = np.tile([1.0, 0, 0, 0], (nb, 1))
        for i, b in enumerate(self.bodies):
            if i == 0:
                continue
            use_geoms = self.inertiafromgeom == "true" or (self.inertiafromgeom == "auto" and b["inertial"] is None)
            if not use_geoms and b["inertial"] is not None:
                body_mass[i] = b["inertial"]["mass"]
                body_inertia[i] = b["inertial"]["inertia"]
                body_ipos[i] = b["inertial"]["pos"]
                body_iquat[i] = b["inertial"]["quat"]
                continue
            gs = [g for g in b["geoms"] if g["mass"] > 0]
            if not gs:
                continue
            if len(gs) == 1:
                g = gs[0]
                body_mass[i], body_inertia[i], body_ipos[i], body_iquat[i] = g["mass"], g["inertia"], g["pos"], g["quat"]
                continue
            mass = sum(g["mass"] for g in gs)
            com = sum(g["mass"] * g["pos"] for g in gs) / mass
            I = np.zeros((3, 3))
            for g in gs:
                R = quat_to_mat(g["quat"])
                d = g["pos"] - com
                I += R @ np.diag(g["inertia"]) @ R.T + g["mass"] * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
            w, V = np.linalg.eigh(I)
            order = np.argsort(-w)
            w, V = w[order], V[:, order]
            if np.linalg.det(V) < 0:
                V[:, 2] = -V[:, 2]
            body_mass[i], body_inertia[i], body_ipos[i], body_iquat[i] = mass, w, com, mat_to_quat(V)

        # ---- joints / dofs / qpos
        nj = len(self.joints)
        jnt_type = np.zeros(nj, np.int32); jnt_qposadr = np.zeros(nj, np.int32); jnt_dofadr = np.zeros(nj, np.int32)
        jnt_bodyid = np.zeros(nj, np.int32); jnt_limited = np.zeros(nj, np.int32)
        jnt_pos = np.zeros((nj, 3)); jnt_axis = np.zeros((nj, 3)); jnt_stiffness = np.zeros(nj)
        jnt_range = np.zeros((nj, 2)); jnt_margin = np.zeros(nj)
        jnt_solref = np.tile(DEFAULT_SOLREF, (nj, 1)).astype(float)
        jnt_solimp = np.tile(DEFAULT_SOLIMP, (nj, 1)).astype(float)
        qpos0, qpos_spring = [], []
        dof_bodyid, dof_jntid, dof_parentid, dof_armature, dof_damping, dof_frictionloss = [], [], [], [], [], []
        dof_solref, dof_solimp = [], []
        body_jntnum = np.zeros(nb, np.int32); body_jntadr = -np.ones(nb, np.int32)
        body_dofnum = np.zeros(nb, np.int32); body_dofadr = -np.ones(nb, np.int32)
        last_dof_of_body = -np.ones(nb, np.int64)
        for j, jd in enumerate(self.joints):
            a = jd["attrib"]; b = jd["body"]
            t = _JNT_TYPES[a.get("type", "hinge")]
            jnt_type[j] = t; jnt_bodyid[j] = b
            jnt_qposadr[j] = len(qpos0); jnt_dofadr[j] = len(dof_bodyid)
            if body_jntadr[b] < 0:
                body_jntadr[b] = j; body_dofadr[b] = len(dof_bodyid)
            body_jntnum[b] += 1
            jnt_pos[j] = _floats(a.get("pos", "0 0 0"), 3)
            ax = np.array(_floats(a.get("axis", "0 0 1"), 3))
            jnt_axis[j] = ax / np.linalg.norm(ax) if t in (JNT_SLIDE, JNT_HINGE) else [0, 0, 1]
            jnt_stiffness[j] = float(a.get("stiffness", 0))
            jnt_margin[j] = float(a.get("margin", 0))
            ang = (t == JNT_HINGE or t == JNT_BALL) and self.degree
            if "range" in a:
                r = _floats(a["range"], 2)
                jnt_range[j] = [math.radians(x) for x in r] if ang else r
            lim = a.get("limited", "auto")
            jnt_limited[j] = 1 if lim == "true" else (0 if lim == "false" else int(self.autolimits and "range" in a))
            if "solreflimit" in a:
                jnt_solref[j] = _floats(a["solreflimit"], 2)
            if "solimplimit" in a:
                v = _floats(a["solimplimit"])
                jnt_solimp[j, :len(v)] = v
            ref = float(a.get("ref", 0)); sref = float(a.get("springref", 0))
            if ang:
                ref, sref = math.radians(ref), math.radians(sref)
            ndof = {JNT_FREE: 6, JNT_BALL: 3, JNT_SLIDE: 1, JNT_HINGE: 1}[t]
            if t == JNT_FREE:
                qpos0 += list(self.bodies[b]["pos"]) + list(self.bodies[b]["quat"])
                qpos_spring += list(self.bodies[b]["pos"]) + list(self.bodies[b]["quat"])
            elif t == JNT_BALL:
                qpos0 += [1, 0, 0, 0]; qpos_spring += [1, 0, 0, 0]
            else:
                qpos0.append(ref); qpos_spring.append(sref)
            # parent dof: previous dof in the same body, else the last dof up the tree
            for k in range(ndof):
                d = len(dof_bodyid)
                if last_dof_of_body[b] >= 0:
                    parent = int(last_dof_of_body[b])
                else:
                    parent = -1
                    p = self.bodies[b]["parent"]
                    while p > 0:
                        if last_dof_of_body[p] >= 0:
                            parent = int(last_dof_of_body[p]); break
                        p = self.bodies[p]["parent"]
                dof_bodyid.append(b); dof_jntid.append(j); dof_parentid.append(parent)
                dof_armature.append(float(a.get("armature", 0)))
                dof_damping.append(float(a.get("damping", 0)))
                dof_frictionloss.append(float(a.get("frictionloss", 0)))
                dof_solref.append(_floats(a.get("solreffriction", "0.02 1"), 2))
                fimp = list(DEFAULT_SOLIMP)
                if "solimpfriction" in a:
                    v = _floats(a["solimpfriction"]); fimp[:len(v)] = v
                dof_solimp.append(fimp)
                last_dof_of_body[b] = d
                body_dofnum[b] += 1
        nq, nv = len(qpos0), len(dof_bodyid)

        body_parentid = np.array([b["parent"] for b in self.bodies], np.int32)
        body_rootid = np.zeros(nb, np.int32)
        for i in range(1, nb):
            body_rootid[i] = i if body_parentid[i] == 0 else body_rootid[body_parentid[i]]
        body_mocapid = -np.ones(nb, np.int32)
        nmocap = 0
        for i, b in enumerate(self.bodies):
            if b["mocap"]:
                body_mocapid[i] = nmocap; nmocap += 1

        # ---- actuators
        acts = []
        for sec in root.findall("actuator"):
            for e in sec:
                a = self.defaults.resolve(e.tag, e.get("class"))
                if e.tag != "general":
                    a = {**self.defaults.resolve("general", e.get("class")), **a}
                a.update(e.attrib)
                acts.append((e.tag, a))
        nu = len(acts)
        joint_names = [jd["attrib"].get("name", "") for jd in self.joints]
        A = dict(actuator_trnid=np.zeros(nu, np.int32), actuator_gaintype=np.zeros(nu, np.int32),
                 actuator_biastype=np.zeros(nu, np.int32), actuator_ctrllimited=np.zeros(nu, np.int32),
                 actuator_forcelimited=np.zeros(nu, np.int32), actuator_gear=np.ones(nu),
                 actuator_gainprm=np.zeros((nu, 3)), actuator_biasprm=np.zeros((nu, 3)),
                 actuator_ctrlrange=np.zeros((nu, 2)), actuator_forcerange=np.zeros((nu, 2)))
        act_names = []
        for i, (tag, a) in enumerate(acts):
            act_names.append(a.get("name", ""))
            if "joint" not in a:
                raise NotImplementedError("only joint transmissions are supported")
            A["actuator_trnid"][i] = joint_names.index(a["joint"])
            A["actuator_gear"][i] = _floats(a.get("gear", "1"))[0]
            if tag == "motor":
                A["actuator_gainprm"][i, 0] = 1.0
            elif tag == "position":
                kp = float(a.get("kp", 1)); kv = float(a.get("kv", 0))
                A["actuator_gainprm"][i, 0] = kp
                A["actuator_biastype"][i] = 1
                A["actuator_biasprm"][i] = [0, -kp, -kv]
            elif tag == "general":
                g = _floats(a.get("gainprm", "1"))
                A["actuator_gainprm"][i, :min(3, len(g))] = g[:3]
                bt = a.get("biastype", "none")
                A["actuator_biastype"][i] = {"none": 0, "affine": 1}[bt]
                bp = _floats(a.get("biasprm", "0"))
                A["actuator_biasprm"][i, :min(3, len(bp))] = bp[:3]
            else:
                raise NotImplementedError(f"actuator <{tag}>")
            for key, lim in (("ctrl", "ctrllimited"), ("force", "forcelimited")):
                if key + "range" in a:
                    A[f"actuator_{key}range"][i] = _floats(a[key + "range"], 2)
                v = a.get(lim, "auto")
                A[f"actuator_{key}limited"][i] = 1 if v == "true" else (0 if v == "false" else int(self.autolimits and key + "range" in a))

        _GEOM_TYPE = {"plane": 0, "hfield": 1, "sphere": 2, "capsule": 3, "ellipsoid": 4, "cylinder": 5, "box": 6, "mesh": 7}
        geoms = [(bi, g) for bi, b in enumerate(self.bodies) for g in b["geoms"]]
        ng = len(geoms)
        G = dict(
            geom_type=np.array([_GEOM_TYPE[g["type"]] for _, g in geoms], np.int32),
            geom_bodyid=np.array([bi for bi, _ in geoms], np.int32),
            geom_contype=np.array([g["contype"] for _, g in geoms], np.int32),
            geom_conaffinity=np.array([g["conaffinity"] for _, g in geoms], np.int32),
            geom_condim=np.array([g["condim"] for _, g in geoms], np.int32),
            geom_priority=np.array([g["priority"] for _, g in geoms], np.int32),
            geom_group=np.array([g["group"] for _, g in geoms], np.int32),
            geom_size=np.array([g["size"] for _, g in geoms], float).reshape(ng, 3),
            geom_pos=np.array([g["pos"] for _, g in geoms], float).reshape(ng, 3),
            geom_quat=np.array([g["quat"] for _, g in geoms], float).reshape(ng, 4),
            geom_friction=np.array([g["friction"] for _, g in geoms], float).reshape(ng, 3),
            geom_solref=np.array([g["solref"] for _, g in geoms], float).reshape(ng, 2),
            geom_solimp=np.array([g["solimp"] for _, g in geoms], float).reshape(ng, 5),
            geom_margin=np.array([g["margin"] for _, g in geoms], float),
            geom_gap=np.array([g["gap"] for _, g in geoms], float),
            geom_solmix=np.array([g["solmix"] for _, g in geoms], float))
        self.geom_names = [g["name"] for _, g in geoms]
        arrays = dict(
            **G, dof_solref=np.array(dof_solref, float).reshape(-1, 2), dof_solimp=np.array(dof_solimp, float).reshape(-1, 5),
            body_invweight0=np.zeros((nb, 2)), body_subtreemass=np.zeros(nb),
            body_parentid=body_parentid, body_rootid=body_rootid, body_jntnum=body_jntnum, body_jntadr=body_jntadr,
            body_dofnum=body_dofnum, body_dofadr=body_dofadr, body_mocapid=body_mocapid,
            body_pos=np.array([b["pos"] for b in self.bodies]), body_quat=np.array([b["quat"] for b in self.bodies]),
            body_ipos=body_ipos, body_iquat=body_iquat, body_mass=body_mass, body_inertia=body_inertia,
            jnt_type=jnt_type, jnt_qposadr=jnt_qposadr, jnt_dofadr=jnt_dofadr, jnt_bodyid=jnt_bodyid,
            jnt_limited=jnt_limited, jnt_pos=jnt_pos, jnt_axis=jnt_axis, jnt_stiffness=jnt_stiffness,
            jnt_range=jnt_range, jnt_margin=jnt_margin, jnt_solref=jnt_solref, jnt_solimp=jnt_solimp,
            dof_bodyid=np.array(dof_bodyid, np.int32), dof_jntid=np.array(dof_jntid, np.int32),
            dof_parentid=np.array(dof_parentid, np.int32), dof_armature=np.array(dof_armature, float),
            dof_damping=np.array(dof_damping, float), dof_frictionloss=np.array(dof_frictionloss, float),
            dof_invweight0=np.zeros(nv), qpos0=np.array(qpos0, float), qpos_spring=np.array(qpos_spring, float),
            site_bodyid=np.array([s["body"] for s in self.sites], np.int32),
            site_pos=np.array([s["pos"] for s in self.sites], float).reshape(-1, 3),
            site_quat=np.array([s["quat"] for s in self.sites], float).reshape(-1, 4), **A)
        scal.update(nq=nq, nv=nv, nu=nu, na=0, nbody=nb, njnt=nj, nsite=len(self.sites), nmocap=nmocap, nuserdata=0, ngeom=ng)

        # ---- fixed tendons (joint wraps only), contact excludes, weld ids
        tend = []
        for sec in root.findall("tendon"):
            for e in sec:
                if e.tag != "fixed":
                    raise NotImplementedError(f"tendon <{e.tag}> (only fixed tendons are supported)")
                ta = self.defaults.resolve("tendon", e.get("class"))
                ta.update(e.attrib)
                wraps = [(joint_names.index(w.get("joint")), float(w.get("coef"))) for w in e.findall("joint")]
                tend.append((ta, wraps))
        nt = len(tend)
        T = dict(tendon_adr=np.zeros(nt, np.int32), tendon_num=np.zeros(nt, np.int32), tendon_limited=np.zeros(nt, np.int32),
                 wrap_objid=[], wrap_prm=[], tendon_range=np.zeros((nt, 2)), tendon_margin=np.zeros(nt),
                 tendon_solref_lim=np.tile(DEFAULT_SOLREF, (nt, 1)).astype(float).reshape(nt, 2),
                 tendon_solimp_lim=np.tile(DEFAULT_SOLIMP, (nt, 1)).astype(float).reshape(nt, 5),
                 tendon_invweight0=np.zeros(nt))
        tendon_names = []
        for i, (ta, wraps) in enumerate(tend):
            tendon_names.append(ta.get("name", ""))
            T["tendon_adr"][i] = len(T["wrap_objid"]); T["tendon_num"][i] = len(wraps)
            for jid, coef in wraps:
                T["wrap_objid"].append(jid); T["wrap_prm"].append(coef)
            if "range" in ta:
                T["tendon_range"][i] = _floats(ta["range"], 2)
            lim = ta.get("limited", "auto")
            T["tendon_limited"][i] = 1 if lim == "true" else (0 if lim == "false" else int(self.autolimits and "range" in ta))
            T["tendon_margin"][i] = float(ta.get("margin", 0))
            if "solreflimit" in ta:
                T["tendon_solref_lim"][i] = _floats(ta["solreflimit"], 2)
            if "solimplimit" in ta:
                v = _floats(ta["solimplimit"]); T["tendon_solimp_lim"][i, :len(v)] = v
        T["wrap_objid"] = np.array(T["wrap_objid"], np.int32); T["wrap_prm"] = np.array(T["wrap_prm"], float)
        body_names = [b["name"] for b in self.bodies]
        excl = []
        for sec in root.findall("contact"):
            for e in sec:
                if e.tag == "exclude":
                    b1, b2 = body_names.index(e.get("body1")), body_names.index(e.get("body2"))
                    excl.append((min(b1, b2) << 16) + max(b1, b2))
                elif e.tag == "pair":
                    raise NotImplementedError("explicit contact pairs")
        weld = np.zeros(nb, np.int32)
        for i in range(1, nb):
            weld[i] = i if body_dofnum[i] > 0 else weld[body_parentid[i]]
        arrays.update(T, exclude_signature=np.array(excl, np.int32), body_weldid=weld)
        scal.update(ntendon=nt, nwrap=len(T["wrap_objid"]), nexclude=len(excl))

        # ---- custom, sensors, keyframes
        numeric, text = {}, {}
        for sec in root.findall("custom"):
            for e in sec:
                if e.tag == "numeric":
                    numeric[e.get("name")] = np.array(_floats(e.get("data", "0")))
                elif e.tag == "text":
                    text[e.get("name")] = e.get("data", "")
        sensors = []
        nuser_sensor = 0
        for sec in root.findall("sensor"):
            for e in sec:
                dim = {"user": int(e.get("dim", 0)), "framepos": 3, "framelinvel": 3, "frameangvel": 3,
                       "framequat": 4, "jointpos": 1, "jointvel": 1, "subtreecom": 3, "subtreelinvel": 3,
                       "subtreeangmom": 3, "touch": 1, "actuatorfrc": 1, "framexaxis": 3, "frameyaxis": 3,
                       "framezaxis": 3, "framelinacc": 3, "frameangacc": 3, "accelerometer": 3,
                       "velocimeter": 3, "gyro": 3, "force": 3, "torque": 3}.get(e.tag, 1)
                user = _floats(e.get("user", "")) if e.get("user") else []
                nuser_sensor = max(nuser_sensor, len(user))
                sensors.append(dict(name=e.get("name", ""), type=e.tag, dim=dim, user=user,
                                    objtype=e.get("objtype", ""), objname=e.get("objname", e.get("joint", e.get("site", e.get("body", ""))))))
        keyframes = {}
        for sec in root.findall("keyframe"):
            for e in sec:
                k = dict(qpos=np.array(qpos0, float), qvel=np.zeros(nv), ctrl=np.zeros(nu))
                for f in ("qpos", "qvel", "ctrl", "mpos", "mquat"):
                    if e.get(f) is not None:
                        k[f] = np.array(_floats(e.get(f)))
                keyframes[e.get("name", f"key{len(keyframes)}")] = k

        scal["nkey"] = len(keyframes)
        arrays["key_qpos"] = (np.array([k["qpos"] for k in keyframes.values()], float).reshape(len(keyframes), nq)
                              if keyframes else np.zeros((0, nq)))
        arrays["key_qvel"] = (np.array([k["qvel"] for k in keyframes.values()], float).reshape(len(keyframes), nv)
                              if keyframes else np.zeros((0, nv)))
        # a key without mpos keeps the mocap bodies where the model puts them (mj_resetDataKeyframe after mj_resetData)
        mpos0 = np.zeros(3 * nmocap)
        for i in range(nb):
            if body_mocapid[i] >= 0:
                mpos0[3 * body_mocapid[i]:3 * body_mocapid[i] + 3] = self.bodies[i]["pos"]
        arrays["key_mpos"] = (np.array([k.get("mpos", mpos0) for k in keyframes.values()], float).reshape(len(keyframes), 3 * nmocap)
                              if keyframes and nmocap else np.zeros((0, 3 * nmocap)))
        arrays["key_ctrl"] = (np.array([k["ctrl"] for k in keyframes.values()], float).reshape(len(keyframes), nu)
                              if keyframes else np.zeros((0, nu)))
        names = dict(body=[b["name"] for b in self.bodies], joint=joint_names, geom=self.geom_names,
                     key=list(keyframes.keys()), tendon=tendon_names,
                     site=[s["name"] for s in self.sites], actuator=act_names,
                     sensor=[s["name"] for s in sensors])
        fm = FlatModel(arrays=arrays, scalars=scal, names=names, numeric=numeric, text=text,
                       sensors=sensors, keyframes=keyframes, nuser_sensor=nuser_sensor, source=self.source)
        _set_const(fm)
        return fm


def _geom_volume_inertia(gtype, size):
    """(volume, mass -> principal inertia in the geom frame) per MuJoCo's geom formulas."""
    if gtype == "sphere":
        r = size[0]
        return 4 / 3 * math.pi * r ** 3, lambda m: np.full(3, 0.4 * m * r * r)
    if gtype == "capsule":
        r, h = size[0], 2 * size[1]
        vol = math.pi * r * r * h + 4 / 3 * math.pi * r ** 3

        def inert(m):
            ms = m * (4 * r) / (4 * r + 3 * h)   # two hemispheres
            mc = m - ms                           # cylinder
            ixx = mc * (3 * r * r + h * h) / 12 + 0.4 * ms * r * r + ms * h * (3 * r + 2 * h) / 8
            izz = mc * r * r / 2 + 0.4 * ms * r * r
            return np.array([ixx, ixx, izz])
        return vol, inert
    if gtype == "cylinder":
        r, h = size[0], 2 * size[1]
        return math.pi * r * r * h, lambda m: np.array([m * (3 * r * r + h * h) / 12] * 2 + [m * r * r / 2])
    if gtype == "ellipsoid":
        a, b, c = size
        return 4 / 3 * math.pi * a * b * c, lambda m: m / 5 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    if gtype == "box":
        a, b, c = size
        return 8 * a * b * c, lambda m: m / 3 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    return 0.0, lambda m: np.zeros(3)


# ----------------------------------------------------------------------------- mj_setConst
def forward_kinematics(fm: FlatModel, qpos):
    """numpy forward kinematics + composite inertia at `qpos` (compile-time use)."""
    a = fm.arrays
    nb = fm.nbody
    xpos = np.zeros((nb, 3)); xquat = np.tile([1.0, 0, 0, 0], (nb, 1)); xmat = np.tile(np.eye(3), (nb, 1, 1))
    xanchor = np.zeros((fm.njnt, 3)); xaxis = np.zeros((fm.njnt, 3))
    for i in range(1, nb):
        p = a["body_parentid"][i]
        ja, jn = a["body_jntadr"][i], a["body_jntnum"][i]
        if jn == 1 and a["jnt_type"][ja] == JNT_FREE:
            qa = a["jnt_qposadr"][ja]
            pos = np.array(qpos[qa:qa + 3]); quat = np.array(qpos[qa + 3:qa + 7]); quat /= np.linalg.norm(quat)
            xanchor[ja] = pos; xaxis[ja] = [0, 0, 1]
        else:
            pos = xpos[p] + xmat[p] @ a["body_pos"][i]
            quat = quat_mul(xquat[p], a["body_quat"][i])
            for j in range(ja, ja + jn):
                qa = a["jnt_qposadr"][j]
                R = quat_to_mat(quat)
                xanchor[j] = R @ a["jnt_pos"][j] + pos
                xaxis[j] = R @ a["jnt_axis"][j]
                t = a["jnt_type"][j]
                if t == JNT_SLIDE:
                    pos = pos + xaxis[j] * (qpos[qa] - a["qpos0"][qa])
                elif t in (JNT_HINGE, JNT_BALL):
                    ql = (np.array(qpos[qa:qa + 4]) if t == JNT_BALL
                          else axis_angle_quat(a["jnt_axis"][j], qpos[qa] - a["qpos0"][qa]))
                    quat = quat_mul(quat, ql)
                    pos = xanchor[j] - quat_to_mat(quat) @ a["jnt_pos"][j]
        quat = quat / np.linalg.norm(quat)
        xpos[i], xquat[i], xmat[i] = pos, quat, quat_to_mat(quat)
    xipos = xpos + np.einsum("bij,bj->bi", xmat, a["body_ipos"])
    ximat = np.array([quat_to_mat(quat_mul(xquat[i], a["body_iquat"][i])) for i in range(nb)])
    return dict(xpos=xpos, xquat=xquat, xmat=xmat, xipos=xipos, ximat=ximat, xanchor=xanchor, xaxis=xaxis)


def mass_matrix(fm: FlatModel, qpos, jacobians=None):
    """Dense joint-space inertia via body Jacobians (independent of the CRB recursion). If `jacobians` is a dict it
    receives body id -> (Jp at the body's centre of mass, Jr)."""
    a = fm.arrays
    k = forward_kinematics(fm, qpos)
    nv, nb = fm.nv, fm.nbody
    M = np.diag(a["dof_armature"].astype(float)) if nv else np.zeros((0, 0))
    # per-dof world-frame motion axes: (angular axis, point on axis or None for translation)
    dof_ang = np.zeros((nv, 3)); dof_lin = np.zeros((nv, 3)); dof_pt = np.zeros((nv, 3))
    for j in range(fm.njnt):
        t = a["jnt_type"][j]; d = a["jnt_dofadr"][j]; b = a["jnt_bodyid"][j]
        if t == JNT_FREE:
            for c in range(3):
                dof_lin[d + c] = np.eye(3)[c]
                dof_ang[d + 3 + c] = k["xmat"][b][:, c]; dof_pt[d + 3 + c] = k["xanchor"][j]
        elif t == JNT_BALL:
            for c in range(3):
                dof_ang[d + c] = k["xmat"][b][:, c]; dof_pt[d + c] = k["xanchor"][j]
        elif t == JNT_SLIDE:
            dof_lin[d] = k["xaxis"][j]
        else:
            dof_ang[d] = k["xaxis"][j]; dof_pt[d] = k["xanchor"][j]
    for i in range(1, nb):
        # ancestors' dofs move body i
        chain = []
        b = i
        while b > 0:
            da, dn = a["body_dofadr"][b], a["body_dofnum"][b]
            chain += list(range(da, da + dn)) if dn else []
            b = a["body_parentid"][b]
        if not chain:
            continue
        Jp = np.zeros((3, nv)); Jr = np.zeros((3, nv))
        for d in chain:
            Jr[:, d] = dof_ang[d]
            Jp[:, d] = dof_lin[d] + np.cross(dof_ang[d], k["xipos"][i] - dof_pt[d])
        Iw = k["ximat"][i] @ np.diag(a["body_inertia"][i]) @ k["ximat"][i].T
        M += a["body_mass"][i] * Jp.T @ Jp + Jr.T @ Iw @ Jr
        if jacobians is not None:
            jacobians[i] = (Jp, Jr)
    return M


def _set_const(fm: FlatModel):
    """dof_invweight0 and stat.meaninertia at qpos0 (MuJoCo mj_setConst / set0)."""
    nv = fm.nv
    if nv == 0:
        fm.scalars["meaninertia"] = 1.0
        return
    jac = {}
    M = mass_matrix(fm, fm.arrays["qpos0"], jac)
    Minv = np.linalg.inv(M)
    inv = np.diag(Minv).copy()
    a = fm.arrays
    # body_invweight0: mean diagonal of J M^-1 J' for the translational (at the body COM) and rotational Jacobians
    biw = np.zeros((fm.nbody, 2))
    for i, (Jp, Jr) in jac.items():
        biw[i, 0] = np.trace(Jp @ Minv @ Jp.T) / 3.0
        biw[i, 1] = np.trace(Jr @ Minv @ Jr.T) / 3.0
    a["body_invweight0"] = biw
    sub = np.array(a["body_mass"], float).copy()
    for i in range(fm.nbody - 1, 0, -1):
        sub[a["body_parentid"][i]] += sub[i]
    a["body_subtreemass"] = sub
    for j in range(fm.njnt):  # free/ball: average over each 3-dof block
        t, d = a["jnt_type"][j], a["jnt_dofadr"][j]
        if t == JNT_FREE:
            inv[d:d + 3] = inv[d:d + 3].mean(); inv[d + 3:d + 6] = inv[d + 3:d + 6].mean()
        elif t == JNT_BALL:
            inv[d:d + 3] = inv[d:d + 3].mean()
    a["dof_invweight0"] = inv
    fm.scalars["meaninertia"] = float(np.mean(np.diag(M)))
    # tendon_invweight0 = J M^-1 J' with the (constant) Jacobian of a fixed tendon
    for t in range(fm.scalars.get("ntendon", 0)):
        J = np.zeros(nv)
        for w in range(a["tendon_adr"][t], a["tendon_adr"][t] + a["tendon_num"][t]):
            J[a["jnt_dofadr"][a["wrap_objid"][w]]] = a["wrap_prm"][w]
        a["tendon_invweight0"][t] = float(J @ Minv @ J)


def attach_keyframes(fm: FlatModel, names, qpos, qvel, mpos):
    """Keyframes that do not come from the XML (large tables converted to npz): same arrays a <keyframe> section fills."""
    n = len(names)
    fm.scalars["nkey"] = n
    fm.arrays["key_qpos"] = np.asarray(qpos, float).reshape(n, fm.nq)
    fm.arrays["key_qvel"] = np.asarray(qvel, float).reshape(n, fm.nv)
    fm.arrays["key_mpos"] = np.asarray(mpos, float).reshape(n, 3 * fm.nmocap)
    fm.arrays["key_ctrl"] = np.zeros((n, fm.nu))
    fm.names["key"] = list(names)
    fm.keyframes.clear()
    for i, k in enumerate(names):
        fm.keyframes[k] = dict(qpos=fm.arrays["key_qpos"][i], qvel=fm.arrays["key_qvel"][i], ctrl=np.zeros(fm.nu),
                               mpos=fm.arrays["key_mpos"][i])


def load_xml(path: str, include_root: str = None) -> FlatModel:
    path = os.path.abspath(path)
    root = ET.parse(path).getroot()
    _expand_includes(root, os.path.dirname(path), include_root)
    return _Compiler(root, path).compile()


# ----------------------------------------------------------------------------- binary blob for the C++ host
_SENSOR_TYPE = {"user": 100, "framepos": 25}   # mjtSensor values the host code inspects; others -> 0
_OBJ_TYPE = {"body": 1, "site": 6}             # mjtObj


def save_blob(fm: FlatModel, path: str):
    """Serialise a compiled model for mujoco_mpc_amd/host/model_io.cc (the C++ stand-in for mj_loadXML).
    Arrays use MuJoCo's mjModel strides (gear x6, gainprm/biasprm x10, trnid x2, limited flags as bytes)."""
    import struct
    a, sc = fm.arrays, fm.scalars
    nu, nsens = sc["nu"], len(fm.sensors)
    entries = []

    def put_i(name, v):
        entries.append((name, 0, np.ascontiguousarray(v, dtype=np.int32).reshape(-1)))

    def put_r(name, v):
        entries.append((name, 1, np.ascontiguousarray(v, dtype=np.float64).reshape(-1)))

    def put_b(name, v):
        entries.append((name, 2, np.ascontiguousarray(v, dtype=np.uint8).reshape(-1)))

    names = bytearray()

    def name_table(lst):
        adr = []
        for n in lst:
            adr.append(len(names))
            names.extend(n.encode() + b"\0")
        return adr

    keys = list(fm.keyframes.items())
    put_i("sizes", [sc["nq"], sc["nv"], nu, sc["na"], sc["nbody"], sc["njnt"], sc["nsite"], sc["nmocap"],
                    sc["nuserdata"], nsens, fm.nuser_sensor, len(fm.numeric), len(fm.text), int(sc["nkey"])])
    put_r("opt", [sc["timestep"], *sc["gravity"], sc["solver_tolerance"], sc["meaninertia"]])
    put_i("opt_int", [sc["integrator"], sc["solver_iterations"], sc["disableflags"], sc.get("cone", 0), sc.get("ngeom", 0)])
    put_r("opt_impratio", [sc.get("impratio", 1.0)])
    for k in ("geom_type", "geom_bodyid", "geom_contype", "geom_conaffinity", "geom_condim", "geom_priority", "geom_group"):
        put_i(k, a[k])
    for k in ("geom_size", "geom_pos", "geom_quat", "geom_friction", "geom_solref", "geom_solimp", "geom_margin", "geom_gap",
              "geom_solmix", "body_invweight0", "body_subtreemass", "dof_solref", "dof_solimp"):
        put_r(k, a[k])
    for k in ("body_parentid", "body_rootid", "body_jntnum", "body_jntadr", "body_dofnum", "body_dofadr", "body_mocapid",
              "jnt_type", "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "dof_bodyid", "dof_jntid", "dof_parentid",
              "site_bodyid", "actuator_gaintype", "actuator_biastype"):
        put_i(k, a[k])
    for k in ("body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia", "jnt_pos", "jnt_axis",
              "jnt_stiffness", "jnt_range", "jnt_margin", "jnt_solref", "jnt_solimp", "dof_armature", "dof_damping",
              "dof_frictionloss", "dof_invweight0", "qpos0", "qpos_spring", "site_pos", "site_quat",
              "actuator_ctrlrange", "actuator_forcerange"):
        put_r(k, a[k])
    put_b("jnt_limited", a["jnt_limited"])
    put_b("actuator_ctrllimited", a["actuator_ctrllimited"])
    put_b("actuator_forcelimited", a["actuator_forcelimited"])
    trnid = np.full((nu, 2), -1, np.int32); trnid[:, 0] = a["actuator_trnid"]
    gear = np.zeros((nu, 6)); gear[:, 0] = a["actuator_gear"]
    gain = np.zeros((nu, 10)); gain[:, :3] = np.asarray(a["actuator_gainprm"]).reshape(nu, 3)
    bias = np.zeros((nu, 10)); bias[:, :3] = np.asarray(a["actuator_biasprm"]).reshape(nu, 3)
    put_i("actuator_trnid", trnid); put_r("actuator_gear", gear); put_r("actuator_gainprm", gain); put_r("actuator_biasprm", bias)
    # sensors
    stype, sobjtype, sobjid, sdim, sadr, suser = [], [], [], [], [], np.zeros((nsens, max(fm.nuser_sensor, 1)))
    adr = 0
    for i, s in enumerate(fm.sensors):
        stype.append(_SENSOR_TYPE.get(s["type"], 0))
        ot = _OBJ_TYPE.get(s["objtype"], 0)
        sobjtype.append(ot)
        sobjid.append(fm.name2id(s["objtype"], s["objname"]) if ot else -1)
        sdim.append(s["dim"]); sadr.append(adr); adr += s["dim"]
        suser[i, :len(s["user"])] = s["user"]
    put_i("sensor_type", stype); put_i("sensor_objtype", sobjtype); put_i("sensor_objid", sobjid)
    put_i("sensor_dim", sdim); put_i("sensor_adr", sadr)
    put_r("sensor_user", suser[:, :fm.nuser_sensor] if fm.nuser_sensor else np.zeros(0))
    # custom numerics
    nadr, nsize, ndata = [], [], []
    for k, v in fm.numeric.items():
        nadr.append(len(ndata)); nsize.append(len(v)); ndata += list(v)
    put_i("numeric_adr", nadr); put_i("numeric_size", nsize); put_r("numeric_data", ndata)
    put_r("key_qpos", a["key_qpos"]); put_r("key_qvel", a["key_qvel"]); put_r("key_mpos", a["key_mpos"])
    put_r("key_ctrl", a.get("key_ctrl", np.zeros((a["key_qpos"].shape[0] if hasattr(a["key_qpos"], "shape") else 0, sc["nu"]))))
    put_i("tendon_sizes", [sc.get("ntendon", 0), sc.get("nwrap", 0), sc.get("nexclude", 0)])
    for k in ("tendon_adr", "tendon_num", "tendon_limited", "wrap_objid", "exclude_signature", "body_weldid"):
        put_i(k, a[k])
    for k in ("wrap_prm", "tendon_range", "tendon_margin", "tendon_solref_lim", "tendon_solimp_lim", "tendon_invweight0"):
        put_r(k, a[k])
    put_i("name_bodyadr", name_table(fm.names["body"])); put_i("name_jntadr", name_table(fm.names["joint"]))
    put_i("name_siteadr", name_table(fm.names["site"])); put_i("name_sensoradr", name_table(fm.names["sensor"]))
    put_i("name_numericadr", name_table(list(fm.numeric.keys()))); put_i("name_keyadr", name_table([k[0] for k in keys]))
    # custom text (mjModel.text_adr / text_size / text_data: zero-terminated strings; task_transition, residual_list_*)
    tadr, tsize, tdata = [], [], bytearray()
    for v in fm.text.values():
        b = v.encode() + b"\0"
        tadr.append(len(tdata)); tsize.append(len(b)); tdata.extend(b)
    put_i("text_adr", tadr); put_i("text_size", tsize); put_b("text_data", np.frombuffer(bytes(tdata), dtype=np.uint8))
    put_i("name_textadr", name_table(list(fm.text.keys())))
    put_i("name_geomadr", name_table(fm.names.get("geom", [])))
    put_b("names", np.frombuffer(bytes(names) or b"\0", dtype=np.uint8))
    with open(path, "wb") as f:
        f.write(b"MJPXBLOB1\n")
        f.write(struct.pack("<I", len(entries)))
        for name, kind, arr in entries:
            nb = name.encode()
            f.write(struct.pack("<I", len(nb))); f.write(nb)
            f.write(struct.pack("<BQ", kind, arr.size)); f.write(arr.tobytes())
    return path
