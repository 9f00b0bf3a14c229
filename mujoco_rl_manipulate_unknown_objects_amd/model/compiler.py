"""Offline model compiler: MJCF subset + binary STL -> GRPM model blob.

Replaces, for this one model family, what ``mujoco.Physics.from_xml_path``
does at ``simulation/environment/robot_env.py:26`` of the reference: parse
``xmls/<object>_env.xml`` (options ``:3``, defaults ``:11-23``, gripper tree
``:58-93``, object ``:96-99``, actuators ``:103-111``), load the binary STL
meshes, build convex hulls, derive mesh mass / centre of mass / inertia and
emit every constant the step kernels need.

It is run once, in the build container where ``/root/reference`` is mounted;
the resulting blobs under ``assets/`` are data, committed, and are the only
thing the GPU box ever sees (``/root/reference`` does not exist there).

MuJoCo itself is not installed anywhere in this environment, so each rule
below marked [3P-recall] restates MuJoCo's published compile behaviour from
memory and is *unpinned* (see DESIGN.md "parity unpinned").

Fixed topology produced (indices used by every other component):
  bodies : 0 world, 1 ee, 2 base, 3 left knuckle, 4 left finger,
           5 right knuckle, 6 right finger, 7 object
  dofs   : 0-2 ee slides x,y,z; 3 roll; 4 yaw; 5 left knuckle; 6 right
           knuckle; 7-9 object linear (world); 10-12 object angular (local)
  geoms  : 0 floor plane, 1 base, 2 lk, 3 lf, 4 rk, 5 rf, 6 object
"""
import argparse
import math
import os
import struct
import xml.etree.ElementTree as ET

import numpy as np
from scipy.spatial import ConvexHull

from .blob import write_blob

BODY_NAMES = ["world", "ee", "robotiq_85_base_link", "left_inner_knuckle",
              "left_inner_finger", "right_inner_knuckle", "right_inner_finger",
              "object"]
NB, NV, NQ, NU, NG = 8, 13, 14, 7, 7


# --------------------------------------------------------------------------
# small math helpers (wxyz quaternions, as MJCF)
# --------------------------------------------------------------------------
def quat_mul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1*w2 - x1*x2 - y1*y2 - z1*z2,
                     w1*x2 + x1*w2 + y1*z2 - z1*y2,
                     w1*y2 - x1*z2 + y1*w2 + z1*x2,
                     w1*z2 + x1*y2 - y1*x2 + z1*w2])


def quat_to_mat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1-2*(y*y+z*z), 2*(x*y-w*z), 2*(x*z+w*y)],
                     [2*(x*y+w*z), 1-2*(x*x+z*z), 2*(y*z-w*x)],
                     [2*(x*z-w*y), 2*(y*z+w*x), 1-2*(x*x+y*y)]])


def mat_to_quat(R):
    # robust (Shepperd)
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = np.array([0.25*s, (R[2, 1]-R[1, 2])/s, (R[0, 2]-R[2, 0])/s, (R[1, 0]-R[0, 1])/s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i+1) % 3, (i+2) % 3
        s = math.sqrt(R[i, i]-R[j, j]-R[k, k]+1.0) * 2
        q = np.zeros(4)
        q[0] = (R[k, j]-R[j, k])/s
        q[1+i] = 0.25*s
        q[1+j] = (R[j, i]+R[i, j])/s
        q[1+k] = (R[k, i]+R[i, k])/s
    if q[0] < 0:
        q = -q
    return q / np.linalg.norm(q)


def axis_quat(axis, ang):
    axis = np.asarray(axis, float)
    return np.r_[math.cos(ang/2), math.sin(ang/2) * axis / np.linalg.norm(axis)]


def euler_xyz_quat(e):
    # MJCF default eulerseq "xyz": intrinsic x, then y, then z.
    q = axis_quat([1, 0, 0], e[0])
    q = quat_mul(q, axis_quat([0, 1, 0], e[1]))
    return quat_mul(q, axis_quat([0, 0, 1], e[2]))


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


# --------------------------------------------------------------------------
# meshes
# --------------------------------------------------------------------------
def load_stl(path):
    """Binary STL -> (ntri,3,3) float64 (vertices are stored as float32)."""
    b = open(path, "rb").read()
    n = struct.unpack("<I", b[80:84])[0]
    rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
    a = np.frombuffer(b[84:84 + 50*n], dtype=rec)
    return a["v"].astype(np.float64)


def acorn_standin_triangles():
    """Labelled stand-in for the missing ``xmls/meshes/acorn.stl``
    (``.MISSING_LARGE_BLOBS:30``).

    A convex acorn-like body of revolution resting cap-down: flat cap disc,
    cap dome, nut tapering to a blunt tip. About 0.54 m tall and 0.5 m across,
    i.e. the same scale as the three objects that are present. Centred in xy
    at (0.025,-0.05) so that the geom offset (-0.025,0.05,0) of
    ``acorn_env.xml:98`` puts it over the object body origin. NOT the real
    acorn: numbers measured on it are never parity-comparable with a MuJoCo
    run of the real ``acorn_env.xml``.
    """
    rings = [  # (z, radius)
        (0.000, 0.170), (0.025, 0.235), (0.075, 0.262), (0.130, 0.250),
        (0.170, 0.215), (0.250, 0.200), (0.340, 0.165), (0.420, 0.115),
        (0.490, 0.060), (0.535, 0.018),
    ]
    naz = 10
    pts = []
    for k, (z, r) in enumerate(rings):
        off = 0.5 * (k & 1)
        for i in range(naz):
            a = 2 * math.pi * (i + off) / naz
            pts.append([0.025 + r * math.cos(a), -0.05 + r * math.sin(a), z])
    pts.append([0.025, -0.05, 0.545])
    pts = np.round(np.array(pts, dtype=np.float32).astype(np.float64), 7)
    h = ConvexHull(pts)
    tris = pts[h.simplices]
    # orient outward
    c = pts.mean(0)
    for t in tris:
        n = np.cross(t[1]-t[0], t[2]-t[0])
        if np.dot(n, t[0]-c) < 0:
            t[[1, 2]] = t[[2, 1]]
    return tris


def mesh_inertia_legacy(tris):
    """Mass properties of a triangle soup at unit density.

    [3P-recall] MuJoCo 2.2 ("legacy" mesh inertia): reference point = the
    area-weighted mean of the triangle centroids; every triangle spans a
    tetrahedron with that point; volumes enter by absolute value (exact for a
    convex, outward- or inward-wound mesh; over-counts a non-convex one).
    Returns volume, com[3], inertia tensor about the com (3x3).
    """
    a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
    area = 0.5 * np.linalg.norm(np.cross(b-a, c-a), axis=1)
    cen0 = ((a+b+c)/3 * area[:, None]).sum(0) / area.sum()
    A, B, C = a-cen0, b-cen0, c-cen0
    vol = np.abs(np.einsum("ij,ij->i", A, np.cross(B, C))) / 6.0
    V = vol.sum()
    com = (vol[:, None] * (A+B+C) / 4).sum(0) / V + cen0
    # second moment of each tet (vertices 0,p1,p2,p3 relative to com)
    P0 = (cen0 - com)[None, :].repeat(len(tris), 0)
    P = [P0, a-com, b-com, c-com]
    S = sum(P)
    cov = np.zeros((3, 3))
    for p in P:
        cov += np.einsum("i,ij,ik->jk", vol, p, p)
    cov += np.einsum("i,ij,ik->jk", vol, S, S)
    cov /= 20.0
    inertia = np.trace(cov) * np.eye(3) - cov
    return V, com, inertia


def principal_axes(I):
    """Symmetric inertia -> (diag sorted descending as MuJoCo does, quat)."""
    w, v = np.linalg.eigh(I)
    order = np.argsort(-w)
    w, v = w[order], v[:, order]
    if np.linalg.det(v) < 0:
        v[:, 2] = -v[:, 2]
    return w, mat_to_quat(v)


def _convex_loop(pts2):
    """indices of the boundary points of a planar point set's convex hull, counter-clockwise (monotone chain). Points ON an edge stay in the loop: they are
    corners of the neighbouring faces, and the rasteriser's edge tests are bit-consistent only between faces that share the SAME edge segments."""
    order = np.lexsort((pts2[:, 1], pts2[:, 0]))
    scale = max(1e-30, np.abs(pts2 - pts2.mean(0)).max()) ** 2

    def half(seq):
        out = []
        for i in seq:
            while len(out) >= 2:
                a, b = pts2[out[-2]], pts2[out[-1]]
                if (b[0] - a[0]) * (pts2[i][1] - a[1]) - (b[1] - a[1]) * (pts2[i][0] - a[0]) < -1e-9 * scale:
                    out.pop()
                else:
                    break
            out.append(int(i))
        return out
    lo, up = half(order), half(order[::-1])
    return lo[:-1] + up[:-1]


def build_hull(points, with_loops=False):
    """Convex hull of a point cloud -> verts, CSR adjacency, face planes (with_loops: and every face's corner loop, counter-clockwise seen from outside)."""
    uniq = np.unique(points, axis=0)
    h = ConvexHull(uniq)
    vid = np.sort(h.vertices)
    remap = -np.ones(len(uniq), dtype=int)
    remap[vid] = np.arange(len(vid))
    verts = uniq[vid]
    nbr = [set() for _ in vid]
    for s in h.simplices:
        s = remap[s]
        for i in range(3):
            nbr[s[i]].add(int(s[(i+1) % 3]))
            nbr[s[i]].add(int(s[(i+2) % 3]))
    nadr = np.zeros(len(vid)+1, dtype=np.int32)
    flat = []
    for i, s in enumerate(nbr):
        flat += sorted(s)
        nadr[i+1] = len(flat)
    # merged (unique) face planes n.x <= d, outward normals
    eq = h.equations
    key = np.round(eq / np.maximum(1e-12, np.linalg.norm(eq[:, :3], axis=1))[:, None], 7)
    _, idx, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    first = np.sort(idx)
    eq = eq[first]
    planes = np.c_[eq[:, :3], -eq[:, 3]]
    if not with_loops:
        return verts, nadr, np.array(flat, dtype=np.int32), planes
    # the corners of every listed face: the vertices of the simplices that lie in its plane, as a convex loop in the plane. "Lie in" means the rounded key of the plane list or
    # the next rounding bucket (2e-7 on every component of the unit equation), not only "same rounded key": where rounding split one flat face into two listed planes, BOTH get the whole
    # face's loop (a part of a convex polygon's triangulation need not be convex) -- a ray then finds the face twice, at the same depth.
    unit = h.equations / np.maximum(1e-12, np.linalg.norm(h.equations[:, :3], axis=1))[:, None]
    loops = []
    for j, f in enumerate(first):
        same = np.abs(unit - unit[f]).max(1) < 2e-7
        on = np.unique(remap[h.simplices[same].ravel()])
        n = unit[f, :3]
        a = np.cross(n, [1.0, 0, 0]) if abs(n[0]) < 0.9 else np.cross(n, [0, 1.0, 0])
        a /= np.linalg.norm(a); b = np.cross(n, a)                       # a x b = n: counter-clockwise in (a, b) = seen from outside
        loop = on[_convex_loop(np.c_[verts[on] @ a, verts[on] @ b])]
        assert len(loop) >= 3
        loops.append(loop.astype(np.int32))
    return verts, nadr, np.array(flat, dtype=np.int32), planes, loops


LUT_RES = 8


def lut_cell_dirs(res=LUT_RES):
    """Centre directions of a 6 x res x res cube map; cell id = face * res^2 + iu * res + iv.
    face = 2 * major_axis + (0 if the major component is positive else 1); (u, v) are the two other
    components divided by |major|, in axis order, mapped from [-1, 1] to [0, res)."""
    dirs = np.zeros((6 * res * res, 3))
    for face in range(6):
        ax, neg = face // 2, face % 2
        o = [a for a in range(3) if a != ax]
        for iu in range(res):
            for iv in range(res):
                d = np.zeros(3)
                d[ax] = -1.0 if neg else 1.0
                d[o[0]] = (iu + 0.5) / res * 2 - 1
                d[o[1]] = (iv + 0.5) / res * 2 - 1
                dirs[face * res * res + iu * res + iv] = d
    return dirs


def support_lut(verts):
    """Start vertex for hill-climbing support queries: exhaustive argmax at every cube-map cell centre."""
    return np.argmax(lut_cell_dirs() @ verts.T, axis=1).astype(np.int32)


# --------------------------------------------------------------------------
# MJCF subset
# --------------------------------------------------------------------------
def _floats(s):
    return np.array([float(x) for x in s.split()])


class _Defaults:
    def __init__(self, root):
        self.cls = {}
        d = root.find("default")
        self._walk(d, "main", {})

    def _walk(self, node, name, inherited):
        cur = {k: dict(v) for k, v in inherited.items()}
        for child in node:
            if child.tag == "default":
                continue
            cur.setdefault(child.tag, {}).update(child.attrib)
        self.cls[name] = cur
        for child in node.findall("default"):
            self._walk(child, child.attrib["class"], cur)

    def get(self, tag, elem):
        base = dict(self.cls[elem.attrib.get("class", "main")].get(tag, {}))
        base.update(elem.attrib)
        return base


def parse_mjcf(xml_path):
    root = ET.parse(xml_path).getroot()
    comp = root.find("compiler").attrib
    assert comp.get("angle") == "radian" and comp.get("inertiafromgeom") == "true"
    opt = root.find("option").attrib
    dfl = _Defaults(root)
    meshes = {}
    for m in root.find("asset").findall("mesh"):
        fn = m.attrib["file"]
        meshes[m.attrib.get("name", os.path.splitext(fn)[0])] = fn
    mats = {m.attrib["name"]: m.attrib for m in root.find("asset").findall("material")}
    texs = {t.attrib.get("name", t.attrib["type"]): t.attrib for t in root.find("asset").findall("texture")}

    bodies = {}

    def walk(elem, parent):
        name = elem.attrib.get("name")
        if name in BODY_NAMES:
            b = dict(parent=parent,
                     pos=_floats(elem.attrib.get("pos", "0 0 0")),
                     quat=_floats(elem.attrib["quat"]) if "quat" in elem.attrib else np.array([1., 0, 0, 0]),
                     joints=[dfl.get("joint", j) for j in elem.findall("joint")],
                     free=elem.find("freejoint") is not None,
                     geoms=[dfl.get("geom", g) for g in elem.findall("geom")],
                     inertial=elem.find("inertial").attrib if elem.find("inertial") is not None else None,
                     cameras=[c.attrib for c in elem.findall("camera")])
            bodies[name] = b
            parent = name
        for c in elem.findall("body"):
            walk(c, parent)

    wb = root.find("worldbody")
    for c in wb.findall("body"):
        walk(c, "world")
    # the two static cameras of RobotEnv.render (robot xml :41-47): unnamed world bodies holding a `targetbodycom` camera on the object
    static_cams = [dict(pos=_floats(c.attrib.get("pos", "0 0 0")), name=cam.attrib["name"], fovy=float(cam.attrib["fovy"]))
                   for c in wb.findall("body") for cam in c.findall("camera") if cam.attrib.get("mode") == "targetbodycom"]
    floor = [dfl.get("geom", g) for g in wb.findall("geom")]
    lights = [dfl.get("light", l) for l in wb.findall("light")]        # with the <default><light .../> class (diffuse "1 1 1" here)
    acts = [dfl.get("motor", a) for a in root.find("actuator").findall("motor")]
    znear = float(root.find("visual").find("map").attrib["znear"])
    hl = root.find("visual").find("headlight")
    headlight = dict(hl.attrib) if hl is not None else {}
    return dict(headlight=headlight, opt=opt, bodies=bodies, floor=floor, acts=acts, meshes=meshes,
                mats=mats, texs=texs, lights=lights, znear=znear, static_cams=static_cams,
                name=root.attrib.get("model", ""))


# --------------------------------------------------------------------------
# rigid-body quantities at a configuration (numpy; used for invweight0 and as
# an independent check of the C oracle in tests)
# --------------------------------------------------------------------------
def body_jacobians(mdl, qpos):
    """World poses and 6x13 COM Jacobians (rows 0-2 linear, 3-5 angular).

    Returns xpos[8,3], xmat[8,3,3], xipos[8,3], J[8,6,13].
    """
    xpos = np.zeros((NB, 3)); xmat = np.zeros((NB, 3, 3)); xmat[0] = np.eye(3)
    axes = np.zeros((NV, 3)); anchors = np.zeros((NV, 3))
    # ee: slides then roll then yaw (MJCF joint order, robot xml :61-65)
    p = mdl["body_pos"][1] + qpos[0:3]
    R0 = quat_to_mat(mdl["body_quat"][1])
    Rr = R0 @ quat_to_mat(axis_quat([1, 0, 0], qpos[3]))
    Ree = Rr @ quat_to_mat(axis_quat([0, 0, 1], qpos[4]))
    xpos[1], xmat[1] = p, Ree
    axes[0], axes[1], axes[2] = R0[:, 0], R0[:, 1], R0[:, 2]
    axes[3] = R0[:, 0]; axes[4] = Rr[:, 2]
    anchors[3] = anchors[4] = p
    # base (fixed)
    xpos[2] = p + Ree @ mdl["body_pos"][2]
    xmat[2] = Ree @ quat_to_mat(mdl["body_quat"][2])
    for kb, fb, dof in ((3, 4, 5), (5, 6, 6)):
        pk = xpos[2] + xmat[2] @ mdl["body_pos"][kb]
        Rk0 = xmat[2] @ quat_to_mat(mdl["body_quat"][kb])
        axes[dof] = Rk0[:, 1]; anchors[dof] = pk
        Rk = Rk0 @ quat_to_mat(axis_quat([0, 1, 0], qpos[dof]))
        xpos[kb], xmat[kb] = pk, Rk
        xpos[fb] = pk + Rk @ mdl["body_pos"][fb]
        xmat[fb] = Rk @ quat_to_mat(mdl["body_quat"][fb])
    xpos[7] = qpos[7:10]
    xmat[7] = quat_to_mat(qpos[10:14])
    xipos = np.array([xpos[b] + xmat[b] @ mdl["body_ipos"][b] for b in range(NB)])
    J = np.zeros((NB, 6, NV))
    chain = {1: [0, 1, 2, 3, 4], 2: [0, 1, 2, 3, 4], 3: [0, 1, 2, 3, 4, 5],
             4: [0, 1, 2, 3, 4, 5], 5: [0, 1, 2, 3, 4, 6], 6: [0, 1, 2, 3, 4, 6]}
    for b, dofs in chain.items():
        for d in dofs:
            if d < 3:
                J[b, 0:3, d] = axes[d]
            else:
                J[b, 3:6, d] = axes[d]
                J[b, 0:3, d] = np.cross(axes[d], xipos[b] - anchors[d])
    Ro = xmat[7]
    J[7, 0:3, 7:10] = np.eye(3)
    for i in range(3):
        J[7, 3:6, 10+i] = Ro[:, i]
        J[7, 0:3, 10+i] = np.cross(Ro[:, i], xipos[7] - xpos[7])
    return xpos, xmat, xipos, J


def mass_matrix(mdl, qpos):
    xpos, xmat, xipos, J = body_jacobians(mdl, qpos)
    M = np.diag(mdl["dof_armature"].astype(float))
    for b in range(1, NB):
        Ri = xmat[b] @ quat_to_mat(mdl["body_iquat"][b])
        Iw = Ri @ np.diag(mdl["body_inertia"][b]) @ Ri.T
        M += mdl["body_mass"][b] * J[b, 0:3].T @ J[b, 0:3] + J[b, 3:6].T @ Iw @ J[b, 3:6]
    return M, J


# --------------------------------------------------------------------------
# compile
# --------------------------------------------------------------------------
def compile_model(xml_path, mesh_dir, standin_acorn=False):
    x = parse_mjcf(xml_path)
    o = x["opt"]
    assert o.get("cone") == "elliptic"
    mdl = {}
    gravity = _floats(o["gravity"])
    mdl["opt"] = np.array([float(o["timestep"]), gravity[2], float(o["impratio"]),
                           float(o["iterations"]), float(o["tolerance"])])
    B = x["bodies"]
    body_pos = np.zeros((NB, 3)); body_quat = np.tile([1., 0, 0, 0], (NB, 1))
    body_mass = np.zeros(NB); body_ipos = np.zeros((NB, 3))
    body_iquat = np.tile([1., 0, 0, 0], (NB, 1)); body_inertia = np.zeros((NB, 3))
    parent = np.zeros(NB, dtype=np.int32)
    for i, n in enumerate(BODY_NAMES[1:], start=1):
        body_pos[i] = B[n]["pos"]; body_quat[i] = B[n]["quat"] / np.linalg.norm(B[n]["quat"])
        parent[i] = BODY_NAMES.index(B[n]["parent"])
    # --- joints (ee: 5, knuckles: 1 each) -> dof params
    arm = np.zeros(NV); damp = np.zeros(NV); rng = np.zeros((NU, 2))
    ee_j = B["ee"]["joints"]
    assert [j["name"] for j in ee_j] == ["gripper_x", "gripper_y", "gripper_z", "gripper_roll", "gripper_yaw"]
    assert [j.get("type", "hinge") for j in ee_j] == ["slide"]*3 + ["hinge"]*2
    jl = ee_j + B["left_inner_knuckle"]["joints"] + B["right_inner_knuckle"]["joints"]
    jnames = [j["name"] for j in jl]
    for d, j in enumerate(jl):
        arm[d] = float(j.get("armature", 0)); damp[d] = float(j.get("damping", 0))
        assert j.get("limited") == "true"
        rng[d] = _floats(j["range"])
        assert np.allclose(_floats(j.get("pos", "0 0 0")), 0)
    assert np.allclose(_floats(ee_j[3]["axis"]), [1, 0, 0]) and np.allclose(_floats(ee_j[4]["axis"]), [0, 0, 1])
    assert np.allclose(_floats(jl[5]["axis"]), [0, 1, 0]) and np.allclose(_floats(jl[6]["axis"]), [0, 1, 0])
    assert B["object"]["free"]  # <freejoint/> takes no defaults: armature = damping = 0 [3P-recall]
    mdl["dof_armature"], mdl["dof_damping"], mdl["jnt_range"] = arm, damp, rng
    # --- actuators
    gear = np.zeros(NU); crange = np.zeros((NU, 2))
    for a in x["acts"]:
        d = jnames.index(a["joint"])
        gear[d] = float(a["gear"].split()[0]); crange[d] = _floats(a["ctrlrange"])
        assert a.get("ctrllimited") == "true"
    mdl["gear"], mdl["ctrlrange"] = gear, crange
    # --- geoms + meshes
    geom_body = np.zeros(NG, dtype=np.int32)
    geom_fric = np.zeros((NG, 3)); geom_center = np.zeros((NG, 3)); geom_rbound = np.zeros(NG)
    geom_rgba = np.ones((NG, 4)); geom_condim = np.zeros(NG, dtype=np.int32)
    geom_material = np.tile(np.array([0.5, 0.5, 0.0]), (NG, 1))        # specular, shininess, emission
    fl = x["floor"][0]
    assert fl["type"] == "plane"
    geom_fric[0] = _floats(fl.get("friction", "1 0.005 0.0001")); geom_condim[0] = int(fl["condim"])
    geom_material[0] = material_of(x, fl)
    gm = fl  # margin / solref / solimp are identical for every geom (one default class)
    mdl["geom_margin"] = np.array([float(gm["margin"])])
    mdl["geom_solref"] = _floats(gm["solref"]); mdl["geom_solimp"] = _floats(gm["solimp"])
    mdl["lim_solref"] = np.array([0.02, 1.0])               # MuJoCo joint default [3P-recall]
    mdl["lim_solimp"] = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
    hv, hvadr, hvnum = [], [0], []
    hn_adr, hn = [np.zeros(1, dtype=np.int32)], []
    hp, hpadr, hpnum = [], [0], []
    hlut = []
    hloops, hladr = [], [0]          # every face's corner loop (local vertex ids, counter-clockwise seen from outside), CSR over all hull planes
    standin = 0
    for gi, bn in enumerate(BODY_NAMES[2:], start=1):
        bi = BODY_NAMES.index(bn)
        g = B[bn]["geoms"][0]
        assert g["type"] == "mesh" and g["margin"] == gm["margin"] and g["solref"] == gm["solref"]
        fn = x["meshes"][g["mesh"]]
        path = os.path.join(mesh_dir, fn)
        if os.path.exists(path):
            tris = load_stl(path)
        elif standin_acorn and fn == "acorn.stl":
            tris = acorn_standin_triangles(); standin = 1
        else:
            raise FileNotFoundError(path)
        gpos = _floats(g.get("pos", "0 0 0"))
        assert "quat" not in g and "euler" not in g
        tris = tris + gpos                      # bake the geom offset: vertices in BODY frame
        V, com, Ic = mesh_inertia_legacy(tris)
        if "mass" in g:
            mass = float(g["mass"])
        else:
            mass = float(g.get("density", 1000.0)) * V
        Ic = Ic * (mass / V)
        diag, iq = principal_axes(Ic)
        body_mass[bi], body_ipos[bi], body_iquat[bi], body_inertia[bi] = mass, com, iq, diag
        verts, nadr, nbr, planes, loops = build_hull(tris.reshape(-1, 3), with_loops=True)
        geom_body[gi] = bi
        geom_fric[gi] = _floats(g.get("friction", "1 0.005 0.0001")); geom_condim[gi] = int(g["condim"])
        geom_center[gi] = com                     # MuJoCo re-centres a mesh geom at its COM [3P-recall]
        geom_rbound[gi] = np.linalg.norm(tris.reshape(-1, 3) - com, axis=1).max()
        if "rgba" in g:
            geom_rgba[gi] = _floats(g["rgba"])
        elif "material" in g:
            geom_rgba[gi] = _floats(x["mats"][g["material"]]["rgba"])
        geom_material[gi] = material_of(x, g)
        hv.append(verts); hvnum.append(len(verts)); hvadr.append(hvadr[-1] + len(verts))
        hn_adr.append(nadr[1:] + sum(len(a) for a in hn))
        hn.append(nbr)
        hp.append(planes); hpnum.append(len(planes)); hpadr.append(hpadr[-1] + len(planes))
        hlut.append(support_lut(verts))
        seen = {}
        for lp in loops:                          # listed planes that are one geometric face (rounding split it) share a loop: the first keeps it, the others get none
            k = tuple(sorted(int(v) for v in lp))
            if k in seen:
                lp = lp[:0]
            seen[k] = True
            hloops.append(lp); hladr.append(hladr[-1] + len(lp))
    # ee keeps its explicit <inertial> (no geoms on it)
    ine = B["ee"]["inertial"]
    body_mass[1] = float(ine["mass"]); body_ipos[1] = _floats(ine["pos"]); body_inertia[1] = _floats(ine["diaginertia"])
    mdl.update(body_parent=parent, body_pos=body_pos, body_quat=body_quat, body_mass=body_mass,
               body_ipos=body_ipos, body_iquat=body_iquat, body_inertia=body_inertia)
    mdl.update(geom_body=geom_body, geom_friction=geom_fric, geom_center=geom_center,
               geom_rbound=geom_rbound, geom_rgba=geom_rgba, geom_condim=geom_condim)
    # hull tables are indexed by geom id - 1 (the floor has none)
    mdl["hull_vadr"] = np.array(hvadr[:-1], dtype=np.int32); mdl["hull_vnum"] = np.array(hvnum, dtype=np.int32)
    mdl["hull_verts"] = np.vstack(hv)
    mdl["hull_nadr"] = np.concatenate(hn_adr).astype(np.int32)   # CSR over all hull vertices, local ids
    mdl["hull_nbr"] = np.concatenate(hn).astype(np.int32)
    mdl["hull_padr"] = np.array(hpadr[:-1], dtype=np.int32); mdl["hull_pnum"] = np.array(hpnum, dtype=np.int32)
    mdl["hull_planes"] = np.vstack(hp)
    mdl["hull_lut"] = np.concatenate(hlut).astype(np.int32)          # [6 hulls][6 * LUT_RES^2]
    # face polygons for the observation kernel's rasteriser (round 4): plane j's corners are hull_loops[hull_ladr[j] : hull_ladr[j + 1]] (vertex ids local
    # to the plane's hull; an empty range = a duplicate of an earlier plane of the same face)
    mdl["hull_ladr"] = np.array(hladr, dtype=np.int32); mdl["hull_loops"] = np.concatenate(hloops).astype(np.int32)
    # --- qpos0
    qpos0 = np.zeros(NQ); qpos0[7:10] = body_pos[7]; qpos0[10:14] = body_quat[7]
    mdl["qpos0"] = qpos0
    # --- collision pair list after MuJoCo's filters [3P-recall]: same body, parent-child
    # (both with geoms) and world-vs-static are excluded; contype/conaffinity are all 1.
    pairs = []
    for a in range(1, NG):
        for b in range(a+1, NG):
            ba, bb = geom_body[a], geom_body[b]
            if parent[ba] == bb or parent[bb] == ba:
                continue
            pairs.append((a, b))
    mdl["hull_pairs"] = np.array(pairs, dtype=np.int32)
    # --- invweight0 / meaninertia at qpos0 [3P-recall: engine_setconst]
    M, J = mass_matrix(mdl, qpos0)
    Minv = np.linalg.inv(M)
    biw = np.zeros((NB, 2))
    for b in range(1, NB):
        A = J[b] @ Minv @ J[b].T
        biw[b] = [np.trace(A[:3, :3]) / 3, np.trace(A[3:, 3:]) / 3]
    mdl["body_invweight0"] = biw
    mdl["dof_invweight0"] = np.diag(Minv).copy()
    mdl["meaninertia"] = np.array([np.mean(np.diag(M))])
    # --- camera on ee (robot xml :60), lights, colours
    cam = [c for c in B["ee"]["cameras"] if c["name"] == "gripper_camera"][0]
    mdl["cam_pos"] = _floats(cam["pos"]); mdl["cam_quat"] = euler_xyz_quat(_floats(cam["euler"]))
    mdl["cam_fovy"] = np.array([float(cam["fovy"])])
    sc = {c["name"]: c for c in x["static_cams"]}                                 # camera ids 0 and 1 of RobotEnv.render (robot_env.py:302-340)
    mdl["static_cam_pos"] = np.array([sc["workbench_camera"]["pos"], sc["upper_camera"]["pos"]])
    mdl["static_cam_fovy"] = np.array([sc["workbench_camera"]["fovy"], sc["upper_camera"]["fovy"]])
    xpos, xmat, _, _ = body_jacobians(mdl, qpos0)
    cen = np.array([xpos[geom_body[g]] + xmat[geom_body[g]] @ geom_center[g] for g in range(1, NG)])
    lo = (cen - geom_rbound[1:, None]).min(0); hi = (cen + geom_rbound[1:, None]).max(0)
    extent = 0.5 * np.linalg.norm(hi - lo)
    mdl["visual"] = np.array([extent, x["znear"] * extent, 50.0 * extent])  # extent, znear, zfar [3P-recall]
    grid = x["texs"]["grid"]
    mdl["floor_rgb"] = np.r_[_floats(grid["rgb1"]), _floats(grid["rgb2"])]
    sky = x["texs"]["skybox"]
    mdl["sky_rgb"] = np.r_[_floats(sky["rgb1"]), _floats(sky["rgb2"])]
    ld = np.array([_floats(l["dir"]) for l in x["lights"]]); lp = np.array([_floats(l["pos"]) for l in x["lights"]])
    mdl["light_dir"], mdl["light_pos"] = ld, lp
    mdl["light_directional"] = np.array([1 if l["directional"] == "true" else 0 for l in x["lights"]], dtype=np.int32)
    # materials and light colours of the fixed-function lighting MuJoCo's renderer applies (robot xml :29-34, :50-51; n4). Per geom: specular,
    # shininess, emission of its <material> (MuJoCo's defaults 0.5 / 0.5 / 0 where the geom names none; reflectance is 0 for every material a geom
    # of these scenes uses). Per light: the mean of its diffuse / specular / ambient colour (the scenes' lights are white), spot cutoff in degrees,
    # spot exponent -- MuJoCo's defaults 0.7 / 0.3 / 0 / 45 / 10 under the xml's <default><light diffuse="1 1 1"/>. The headlight: MuJoCo's
    # defaults ambient 0.1, diffuse 0.4, specular 0.5 unless <visual><headlight/> says otherwise. [3P-recall]
    mdl["geom_material"] = geom_material
    col = lambda l, k, d: float(np.mean(_floats(l[k]))) if k in l else d
    mdl["light_params"] = np.array([[col(l, "diffuse", 0.7), col(l, "specular", 0.3), col(l, "ambient", 0.0), float(l.get("cutoff", 45.0)), float(l.get("exponent", 10.0))]
                                    for l in x["lights"]])
    h = x["headlight"]
    mdl["headlight"] = np.array([col(h, "ambient", 0.1), col(h, "diffuse", 0.4), col(h, "specular", 0.5)]) * (0.0 if h.get("active", "1") == "0" else 1.0)
    mdl["flags"] = np.array([standin], dtype=np.int32)
    return mdl


def material_of(x, g):
    """(specular, shininess, emission) of a geom: its <material>'s attributes, MuJoCo's material defaults 0.5 / 0.5 / 0 otherwise [3P-recall]."""
    mt = x["mats"].get(g.get("material", ""), {})
    return np.array([float(mt.get("specular", 0.5)), float(mt.get("shininess", 0.5)), float(mt.get("emission", 0.0))])


OBJECTS = ["acorn", "sand_ball", "sugar_cube", "bread_crumb"]


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "assets"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    for obj in OBJECTS:
        mdl = compile_model(os.path.join(a.reference, "xmls", f"{obj}_env.xml"),
                            os.path.join(a.reference, "xmls", "meshes"), standin_acorn=True)
        path = os.path.join(a.out, f"{obj}_env.grpm")
        write_blob(path, mdl)
        print(f"{obj}: hull verts {mdl['hull_vnum'].tolist()} planes {mdl['hull_pnum'].tolist()} "
              f"gripper mass {mdl['body_mass'][1:7].sum():.4f} obj com {mdl['body_ipos'][7].round(4)} "
              f"inertia {mdl['body_inertia'][7].round(5)} standin={int(mdl['flags'][0])} -> {path}")


if __name__ == "__main__":
    main()
