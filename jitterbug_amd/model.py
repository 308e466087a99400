"""Model compiler: spec (``model_spec.SPEC``) -> flat parameter table.

Restates the parts of MuJoCo's MJCF compilation the Jitterbug model relies on
(reference: jitterbug_dmc/jitterbug.xml; the compiler itself is third-party and
not under /root/reference, so the rules below are restated from MuJoCo's
published documentation - see DESIGN.md "oracle pinning"):

 * geom mass = density x volume of the solid primitive; inertia of the solid
   primitive about its centre (box, cylinder, ellipsoid, sphere);
 * ``fromto`` cylinders: centre = midpoint, half length = |to-from|/2, geom z
   axis along (from - to);
 * body mass / COM / inertia = sum over its geoms (parallel-axis theorem);
 * ``body_invweight0`` = trace(J M^-1 J^T)/3 of the body-COM translational and
   rotational Jacobians at qpos0.

Output layout: ``include/jitterbug_model.h``.  All positions are relative to
the root body origin at qpos0, in root (== world at qpos0) axes.
"""
import math

import numpy as np

from . import model_spec

# ---- layout constants: keep in step with include/jitterbug_model.h ----------
NBODY, NHINGE, NGEOM, NLEG, NQ, NV = 10, 9, 22, 4, 16, 15
P_TIMESTEP, P_GRAVITY, P_SOLREF, P_SOLIMP, P_FRICTION = 0, 1, 4, 6, 11
P_GEAR, P_GAIN, P_BIASPRM, P_CTRLRANGE, P_ROOTPOS0, P_TARGETZ, P_IMPRATIO = 12, 13, 14, 17, 19, 22, 23
OPT_SIZE = 24
P_BODY, BODY_STRIDE = OPT_SIZE, 12
B_MASS, B_COM, B_INERTIA, B_INVW_TRAN, B_INVW_ROT = 0, 1, 4, 10, 11
P_HINGE, HINGE_STRIDE = P_BODY + NBODY * BODY_STRIDE, 8
H_ANCHOR, H_AXIS, H_STIFFNESS, H_DAMPING = 0, 3, 6, 7
P_GEOM, GEOM_STRIDE = P_HINGE + NHINGE * HINGE_STRIDE, 18
G_TYPE, G_BODY, G_CENTER, G_ROT, G_SIZE = 0, 1, 2, 5, 14
NPARAM = P_GEOM + NGEOM * GEOM_STRIDE
GEOM_SPHERE, GEOM_CYLINDER, GEOM_BOX, GEOM_ELLIPSOID = 0, 1, 2, 3
GEOM_TYPE_ID = dict(sphere=GEOM_SPHERE, cylinder=GEOM_CYLINDER, box=GEOM_BOX, ellipsoid=GEOM_ELLIPSOID)

BODY_PARENT = (-1, 0, 1, 0, 3, 0, 5, 0, 7, 0)
BODY_NAMES = ("jitterbug", "leg2upper", "leg2lower", "leg3upper", "leg3lower",
              "leg1upper", "leg1lower", "leg4upper", "leg4lower", "mass")

TASKS = ("move_from_origin", "face_direction", "move_in_direction", "move_to_position", "move_to_pose")
OBS_DIM = dict(move_from_origin=15, face_direction=16, move_in_direction=19, move_to_position=18, move_to_pose=19)


def _quat_z2vec(vec):
    """Rotation matrix taking the z axis onto ``vec`` by the minimal rotation
    (MuJoCo's rule for ``fromto`` geoms)."""
    v = np.asarray(vec, dtype=np.float64)
    n = np.linalg.norm(v)
    if n < 1e-15:
        return np.eye(3)
    v = v / n
    z = np.array([0.0, 0.0, 1.0])
    axis = np.cross(z, v)
    s = np.linalg.norm(axis)
    if s < 1e-15:
        if v[2] < 0:                      # opposite: half turn about x
            return np.diag([1.0, -1.0, -1.0])
        return np.eye(3)
    axis = axis / s
    ang = math.atan2(s, float(v @ z))
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)


def _geom_compile(g, default_density):
    """-> dict(type, center(world), R, size3, mass, inertia(world axes, about centre))."""
    t = g["type"]
    rho = float(g.get("density", default_density))
    size = [float(s) for s in g["size"]]
    R = np.eye(3)
    if "fromto" in g:
        ft = np.asarray(g["fromto"], dtype=np.float64)
        p0, p1 = ft[:3], ft[3:]
        center = 0.5 * (p0 + p1)
        half = 0.5 * np.linalg.norm(p1 - p0)
        R = _quat_z2vec(p0 - p1)
        size = [size[0], half, 0.0]
    else:
        center = np.asarray(g["pos"], dtype=np.float64)
    if t == "sphere":
        r = size[0]
        vol = 4.0 / 3.0 * math.pi * r ** 3
        m = rho * vol
        Il = np.diag([0.4 * m * r * r] * 3)
        size3 = [r, 0.0, 0.0]
    elif t == "cylinder":
        r, h = size[0], size[1]
        vol = math.pi * r * r * 2 * h
        m = rho * vol
        ixx = m * (3 * r * r + 4 * h * h) / 12.0     # = m (r^2/4 + (2h)^2/12)
        Il = np.diag([ixx, ixx, 0.5 * m * r * r])
        size3 = [r, h, 0.0]
    elif t == "box":
        a, b, c = size
        vol = 8 * a * b * c
        m = rho * vol
        Il = np.diag([m / 3 * (b * b + c * c), m / 3 * (a * a + c * c), m / 3 * (a * a + b * b)])
        size3 = [a, b, c]
    elif t == "ellipsoid":
        a, b, c = size
        vol = 4.0 / 3.0 * math.pi * a * b * c
        m = rho * vol
        Il = np.diag([m / 5 * (b * b + c * c), m / 5 * (a * a + c * c), m / 5 * (a * a + b * b)])
        size3 = [a, b, c]
    else:
        raise ValueError("unsupported geom type %r" % t)
    return dict(type=t, center=center, R=R, size=size3, mass=m, inertia=R @ Il @ R.T, name=g.get("name"))


def _body_compile(geoms):
    m = sum(g["mass"] for g in geoms)
    com = sum(g["mass"] * g["center"] for g in geoms) / m
    I = np.zeros((3, 3))
    for g in geoms:
        d = g["center"] - com
        I += g["inertia"] + g["mass"] * ((d @ d) * np.eye(3) - np.outer(d, d))
    return m, com, I


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def mass_matrix_qpos0(params):
    """Joint-space inertia M (15x15, MuJoCo dof order) at qpos0 by the Jacobian
    sum  M = sum_b m Jv^T Jv + Jw^T I Jw  - the plain-numpy twin of the oracle's
    routine, used for body_invweight0 and as an independent check in the tests.
    Returns (M, Jv[NBODY,3,NV], Jw[NBODY,3,NV])."""
    p = np.asarray(params, dtype=np.float64)
    M = np.zeros((NV, NV))
    Jvs = np.zeros((NBODY, 3, NV))
    Jws = np.zeros((NBODY, 3, NV))
    for b in range(NBODY):
        o = P_BODY + b * BODY_STRIDE
        m = p[o + B_MASS]
        c = p[o + B_COM:o + B_COM + 3]
        ii = p[o + B_INERTIA:o + B_INERTIA + 6]
        I = np.array([[ii[0], ii[3], ii[4]], [ii[3], ii[1], ii[5]], [ii[4], ii[5], ii[2]]])
        Jv = np.zeros((3, NV))
        Jw = np.zeros((3, NV))
        Jv[:, 0:3] = np.eye(3)
        for j in range(3):
            e = np.eye(3)[j]
            Jw[:, 3 + j] = e
            Jv[:, 3 + j] = np.cross(e, c)          # root origin is the pivot (0,0,0)
        a = b
        while a > 0:
            h = a - 1
            ho = P_HINGE + h * HINGE_STRIDE
            anchor = p[ho + H_ANCHOR:ho + H_ANCHOR + 3]
            axis = p[ho + H_AXIS:ho + H_AXIS + 3]
            Jw[:, 6 + h] = axis
            Jv[:, 6 + h] = np.cross(axis, c - anchor)
            a = BODY_PARENT[a]
        M += m * Jv.T @ Jv + Jw.T @ I @ Jw
        Jvs[b], Jws[b] = Jv, Jw
    return M, Jvs, Jws


def compile_spec(spec=None):
    """Compile a model spec to the flat parameter table (float64[NPARAM])."""
    spec = spec if spec is not None else model_spec.SPEC
    rho0 = model_spec.DEFAULT_DENSITY if "default_density" not in spec else float(spec["default_density"])
    p = np.zeros(NPARAM, dtype=np.float64)
    p[P_TIMESTEP] = spec["timestep"]
    p[P_GRAVITY:P_GRAVITY + 3] = spec["gravity"]
    p[P_SOLREF:P_SOLREF + 2] = spec["solref"]
    p[P_SOLIMP:P_SOLIMP + 5] = spec["solimp"]
    p[P_FRICTION] = spec["friction"]
    p[P_IMPRATIO] = spec.get("impratio", 1.0)
    act = spec["actuator"]
    p[P_GEAR] = act["gear"]
    p[P_GAIN] = act["gainprm"][0]
    p[P_BIASPRM:P_BIASPRM + 3] = act["biasprm"]
    p[P_CTRLRANGE:P_CTRLRANGE + 2] = act["ctrlrange"]
    root_pos = np.asarray(spec["root"]["pos"], dtype=np.float64)
    p[P_ROOTPOS0:P_ROOTPOS0 + 3] = root_pos
    p[P_TARGETZ] = spec["target"]["pos"][2]

    body_geoms = [[_geom_compile(g, rho0) for g in spec["root"]["geoms"]]]
    hinges = []
    for leg in spec["legs"]:
        for part in ("upper", "lower"):
            body_geoms.append([_geom_compile(g, rho0) for g in leg[part]["geoms"]])
            hinges.append(leg[part]["joint"])
    body_geoms.append([_geom_compile(g, rho0) for g in spec["mass"]["geoms"]])
    hinges.append(spec["mass"]["joint"])
    assert len(body_geoms) == NBODY and len(hinges) == NHINGE

    gi = 0
    for b, geoms in enumerate(body_geoms):
        m, com, I = _body_compile(geoms)
        o = P_BODY + b * BODY_STRIDE
        p[o + B_MASS] = m
        p[o + B_COM:o + B_COM + 3] = com - root_pos
        p[o + B_INERTIA:o + B_INERTIA + 6] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]
        for g in geoms:
            go = P_GEOM + gi * GEOM_STRIDE
            p[go + G_TYPE] = GEOM_TYPE_ID[g["type"]]
            p[go + G_BODY] = b
            p[go + G_CENTER:go + G_CENTER + 3] = g["center"] - root_pos
            p[go + G_ROT:go + G_ROT + 9] = g["R"].reshape(9)
            p[go + G_SIZE:go + G_SIZE + 3] = g["size"]
            gi += 1
    assert gi == NGEOM
    for h, j in enumerate(hinges):
        ho = P_HINGE + h * HINGE_STRIDE
        p[ho + H_ANCHOR:ho + H_ANCHOR + 3] = np.asarray(j["pos"], dtype=np.float64) - root_pos
        p[ho + H_AXIS:ho + H_AXIS + 3] = _unit(j["axis"])
        p[ho + H_STIFFNESS] = j.get("stiffness", 0.0)
        p[ho + H_DAMPING] = j.get("damping", 0.0)

    # body_invweight0 at qpos0
    M, Jvs, Jws = mass_matrix_qpos0(p)
    Minv = np.linalg.inv(M)
    for b in range(NBODY):
        o = P_BODY + b * BODY_STRIDE
        p[o + B_INVW_TRAN] = np.trace(Jvs[b] @ Minv @ Jvs[b].T) / 3.0
        p[o + B_INVW_ROT] = np.trace(Jws[b] @ Minv @ Jws[b].T) / 3.0
    return p


_DEFAULT = None


def default_params():
    """Compiled table of the nominal model (cached; returns a copy)."""
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = compile_spec(model_spec.SPEC)
    return _DEFAULT.copy()


def qpos0(params=None):
    p = default_params() if params is None else params
    q = np.zeros(NQ)
    q[0:3] = p[P_ROOTPOS0:P_ROOTPOS0 + 3]
    q[3] = 1.0
    return q


def body_mass(params, b):
    return float(params[P_BODY + b * BODY_STRIDE + B_MASS])


def geom_masses(spec=None):
    """name/index -> geom mass, for the known-answer tests."""
    spec = spec if spec is not None else model_spec.SPEC
    out = []
    groups = [spec["root"]["geoms"]]
    for leg in spec["legs"]:
        groups += [leg["upper"]["geoms"], leg["lower"]["geoms"]]
    groups.append(spec["mass"]["geoms"])
    for geoms in groups:
        for g in geoms:
            out.append(_geom_compile(g, model_spec.DEFAULT_DENSITY)["mass"])
    return np.array(out)
