"""ORACLE — test infrastructure only.  ctypes binding of oracle/liboracle.so (the
CPU restatement) and oracle/_ref/libref.so (built from the reference's own
sources).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; nothing under spcbpt-optix7_amd/ does."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libref.so")


def _pkg():
    import sys
    root = os.path.dirname(_HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    import __graft_entry__ as g
    return g.load_package()


class EyeVertex(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("normal", C.c_float * 3), ("flux", C.c_float * 3), ("color", C.c_float * 3),
                ("last_position", C.c_float * 3), ("rmis3", C.c_float * 3), ("pdf", C.c_float), ("single_pdf", C.c_float),
                ("last_normal_projection", C.c_float), ("material_id", C.c_int32), ("subspace_id", C.c_int32),
                ("depth", C.c_int32), ("last_zone_id", C.c_int32)]


EYE_VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("flux", "<f4", 3), ("color", "<f4", 3),
                             ("last_position", "<f4", 3), ("rmis3", "<f4", 3), ("pdf", "<f4"), ("single_pdf", "<f4"),
                             ("last_normal_projection", "<f4"), ("material_id", "<i4"), ("subspace_id", "<i4"),
                             ("depth", "<i4"), ("last_zone_id", "<i4")])
assert EYE_VERTEX_DTYPE.itemsize == C.sizeof(EyeVertex)

EYE_STEP_IN_DTYPE = np.dtype([("last", EYE_VERTEX_DTYPE), ("next_flux", "<f4", 3), ("next_single_pdf", "<f4"), ("seed", "<u4"),
                              ("dir", "<f4", 3), ("flags", "<u4"), ("pad", "<u4", 2)])
EYE_STEP_OUT_DTYPE = np.dtype([("kind", "<u4"), ("mid", EYE_VERTEX_DTYPE), ("dir", "<f4", 3), ("next_flux", "<f4", 3),
                               ("next_single_pdf", "<f4"), ("seed", "<u4"), ("done", "<u4"), ("emit", "<f4", 3), ("t_hit", "<f4"), ("pad", "<u4")])
assert EYE_STEP_IN_DTYPE.itemsize == 36 * 4 and EYE_STEP_OUT_DTYPE.itemsize == 40 * 4   # include/spcbpt.h: SPCBPT_UNIT_EYE_STEP

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} not built: make -C oracle")
        _lib = C.CDLL(LIB)
        _lib.orc_create.restype = C.c_void_p
        _lib.orc_rnd.restype = C.c_float
        _lib.orc_tea4.restype = C.c_uint32
    return _lib


def ref_lib():
    if not os.path.exists(REF_LIB):
        return None
    r = C.CDLL(REF_LIB)
    r.ref_rnd.restype = C.c_float
    r.ref_tea4.restype = C.c_uint32
    r.ref_lcg.restype = C.c_uint32
    return r


REF_LAYOUT_LIB = os.path.join(_HERE, "_ref", "libref_layout.so")


def ref_layout():
    """sizeof / offsetof / default values of the reference's own Light, MaterialData, LightParameter, MaterialParameter
    (oracle/ref_layout.cpp over cuda/Light.h, cuda/MaterialData.h, OptiXPathTracer/{light,material}_parameters.h), or None
    when oracle/_ref has not been built."""
    if not os.path.exists(REF_LAYOUT_LIB):
        return None
    import json
    l = C.CDLL(REF_LAYOUT_LIB)
    l.ref_layout_json.restype = C.c_char_p
    return json.loads(l.ref_layout_json().decode())


REF_VIEWER_LIB = os.path.join(_HERE, "_ref", "libref_viewer.so")


def ref_viewer_replay(eye, lookat, up, fov_y, aspect, events):
    """The reference's own sutil::Trackball + sutil::Camera driven by an event list (oracle/ref_viewer.cpp); None when
    oracle/_ref has not been built (GPU box without the prebuilt files)."""
    if not os.path.exists(REF_VIEWER_LIB):
        return None
    r = C.CDLL(REF_VIEWER_LIB)
    ev = np.ascontiguousarray(events, np.float64)
    out = np.zeros((ev.shape[0], 18), np.float32)
    f = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(C.c_void_p)
    r.ref_viewer_replay(f(eye), f(lookat), f(up), C.c_float(fov_y), C.c_float(aspect), C.c_void_p(ev.ctypes.data), ev.shape[0],
                        C.c_void_p(out.ctypes.data))
    return out


class Oracle:
    """CPU restatement driven like the product Renderer."""

    def __init__(self, scene, nthreads: int = 0):
        self.pkg = _pkg()
        self.l = lib()
        sd, self._keep = scene.desc()
        self.h = C.c_void_p(self.l.orc_create(C.byref(sd)))
        self.width = self.height = 0
        self.nthreads = nthreads or (os.cpu_count() or 1)
        self.lt = None

    def close(self):
        if self.h:
            self.l.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_camera(self, eye, U, V, W):
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (eye, U, V, W)]
        self.l.orc_set_camera(self.h, *[C.c_void_p(x.ctypes.data) for x in a])

    def set_camera_lookat(self, eye, lookat, up, fov, aspect):
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (eye, lookat, up)]
        U, V, W = (np.zeros(3, np.float32) for _ in range(3))
        self.l.orc_camera_frame(*[C.c_void_p(x.ctypes.data) for x in a], C.c_float(fov), C.c_float(aspect),
                                C.c_void_p(U.ctypes.data), C.c_void_p(V.ctypes.data), C.c_void_p(W.ctypes.data))
        self.set_camera(a[0], U, V, W)
        return U, V, W

    def resize(self, w, h):
        self.l.orc_resize(self.h, w, h)
        self.width, self.height = w, h

    def set_subspace(self, eye_tree=None, light_tree=None, q=None, cmf_gamma=None):
        p = self.pkg
        if eye_tree is None:
            self.l.orc_set_subspace(self.h, None, 0, None, 0, None, None)
            return
        et = np.ascontiguousarray(eye_tree, dtype=p.TREE_NODE_DTYPE)
        lt = np.ascontiguousarray(light_tree, dtype=p.TREE_NODE_DTYPE)
        q = np.ascontiguousarray(q, dtype=np.float32)
        g = np.ascontiguousarray(cmf_gamma, dtype=np.float32)
        self._sub_keep = (et, lt, q, g)
        self.l.orc_set_subspace(self.h, C.c_void_p(et.ctypes.data), et.shape[0], C.c_void_p(lt.ctypes.data), lt.shape[0],
                                C.c_void_p(q.ctypes.data), C.c_void_p(g.ctypes.data))

    def set_light_trace(self, num_core, core_padding, m_per_core, decorrelate=None):
        if decorrelate is None:
            decorrelate = m_per_core == 1   # same default as the product binding
        self.l.orc_set_light_trace(self.h, num_core, core_padding, m_per_core)
        self.l.orc_set_light_decorrelate(self.h, int(decorrelate))
        self.lt = (num_core, core_padding, m_per_core)

    def set_cmf_double(self, on):
        """Test knob: double-precision CMF accumulation (product behaviour) instead of the reference's float prefix sums."""
        self.l.orc_set_cmf_double(self.h, int(on))

    def set_skip_null_connections(self, on):
        """Test knob: like the product, skip work whose contribution is exactly zero -- shadow rays of connections whose BSDF
        factor is zero and the rest of an eye path after a sampled direction with zero BSDF value (d11).  The
        image is the same either way; only the event counts change."""
        self.l.orc_set_skip_null_connections(self.h, int(on))

    def set_environment(self, rgba, center, radius):
        """The environment map as one more light (env_params_setup + LightSource_shift): rgba = (h, w, 4) float32 as the .hdr stores it."""
        a = np.ascontiguousarray(rgba, dtype=np.float32)
        c3 = np.ascontiguousarray(center, dtype=np.float32)
        rc = self.l.orc_set_environment(self.h, C.c_void_p(a.ctypes.data), a.shape[1], a.shape[0], C.c_void_p(c3.ctypes.data), C.c_float(radius))
        assert rc == 0, rc

    def set_pt_env_nee_fixed(self, on):
        """Test knob: "pt" aims the shadow ray of its sky sample along the sampled direction (upstream aims it at P + d + 2r(1,1,1))."""
        self.l.orc_set_pt_env_nee_fixed(self.h, int(on))

    def set_env_miss_strategy(self, on):
        """Test knob: eye sub-paths that leave the scene see the sky, weighted by rmis::light_hit_env (uncalled upstream): completes the estimator."""
        self.l.orc_set_env_miss_strategy(self.h, int(on))

    def env_partition(self, depth, n, frame=1, vertices=False):
        """Test utility (orc_debug_env_partition): RMIS weights of every strategy of n camera paths that leave the scene after `depth`
        surface vertices: (weights [n, 6]: sum, miss, k = 0 .. 3; first-principles weights [n, 5]: miss, k = 0 .. 3)."""
        out = np.zeros((n, 7), np.float32); truth = np.zeros((n, 5), np.float32)
        ev = np.zeros((n, 4), EYE_VERTEX_DTYPE); lv = np.zeros((n, 4), self.pkg.LIGHT_VERTEX_DTYPE)
        self.l.orc_debug_env_partition.restype = C.c_int
        m = self.l.orc_debug_env_partition(self.h, int(depth), int(n), C.c_uint(frame), out.ctypes.data_as(C.c_void_p), truth.ctypes.data_as(C.c_void_p),
                                          ev.ctypes.data_as(C.c_void_p) if vertices else None, lv.ctypes.data_as(C.c_void_p) if vertices else None)
        if m < 0: raise RuntimeError("env_partition: needs an environment map and 1 <= depth <= 4")
        if vertices: return out[:m, :6], truth[:m], ev[:m], lv[:m]     # ev / lv [path, k]: the (eye vertex, light vertex) pair of strategy k
        return out[:m, :6], truth[:m]

    def quad_partition(self, depth, n, frame=1, vertices=False):
        """Test utility (orc_debug_quad_partition): the same for camera paths that END ON A QUAD EMITTER after `depth` surface vertices:
        (weights [n, 6]: sum, emitter hit, k = 0 .. 3; first-principles weights [n, 5])."""
        out = np.zeros((n, 7), np.float32); truth = np.zeros((n, 5), np.float32)
        ev = np.zeros((n, 4), EYE_VERTEX_DTYPE); lv = np.zeros((n, 4), self.pkg.LIGHT_VERTEX_DTYPE)
        self.l.orc_debug_quad_partition.restype = C.c_int
        m = self.l.orc_debug_quad_partition(self.h, int(depth), int(n), C.c_uint(frame), out.ctypes.data_as(C.c_void_p), truth.ctypes.data_as(C.c_void_p),
                                          ev.ctypes.data_as(C.c_void_p) if vertices else None, lv.ctypes.data_as(C.c_void_p) if vertices else None)
        if m < 0: raise RuntimeError("quad_partition: 1 <= depth <= 4")
        if vertices: return out[:m, :6], truth[:m], ev[:m], lv[:m]     # ev / lv [path, k]: the (eye vertex, light vertex) pair of strategy k
        return out[:m, :6], truth[:m]

    def set_count_as_executed(self, on):
        """Test knob, counters only: charge the classification and first-stage-sampling events the product's timed kernels execute
        (labels cached per vertex, guide table + windows per resampling stage, one Gamma / Q read) instead of the reference's."""
        self.l.orc_set_count_as_executed(self.h, int(on))

    def enable_counters(self, on):
        self.l.orc_enable_counters(self.h, int(on))

    def launch(self, name, frame, rows=None, nthreads=None):
        r0, r1, rs = rows if rows is not None else (0, self.height, 1)
        rc = self.l.orc_launch(self.h, name.encode(), C.c_uint(frame), r0, r1, rs, nthreads or self.nthreads)
        if rc != 0:
            raise RuntimeError(f"orc_launch({name}) -> {rc}")

    def build_sampler(self):
        self.l.orc_build_sampler(self.h)

    def render_frame(self, alg, subframe, launch_frame=None, rows=None):
        if alg in ("SPCBPT_eye", "SPCBPT_no_rmis"):
            self.launch("light trace", subframe + 1 if launch_frame is None else launch_frame)
            self.build_sampler()
        self.launch(alg, subframe, rows)

    def read_accum(self):
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self.l.orc_read_accum(self.h, C.c_void_p(out.ctypes.data))
        return out

    def read_frame(self):
        out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self.l.orc_read_frame(self.h, C.c_void_p(out.ctypes.data))
        return out

    def clear_accum(self):
        self.l.orc_clear_accum(self.h)

    def lvc_read(self):
        n = C.c_int()
        self.l.orc_lvc_read(self.h, None, 0, C.byref(n))
        out = np.zeros(n.value, dtype=self.pkg.LIGHT_VERTEX_DTYPE)
        self.l.orc_lvc_read(self.h, C.c_void_p(out.ctypes.data), n.value, C.byref(n))
        return out

    def lvc_import(self, verts):
        v = np.ascontiguousarray(verts, dtype=self.pkg.LIGHT_VERTEX_DTYPE)
        self.l.orc_lvc_import(self.h, C.c_void_p(v.ctypes.data), v.shape[0])

    def sampler_read(self, capacity=1 << 22):
        p = self.pkg
        sub = np.zeros(p.NUM_SUBSPACE, dtype=p.SUBSPACE_DTYPE)
        cmfs = np.zeros(capacity, dtype=np.float32)
        jump = np.zeros(capacity, dtype=np.int32)
        vc, pc = C.c_int(), C.c_int()
        rc = self.l.orc_sampler_read(self.h, C.c_void_p(sub.ctypes.data), C.c_void_p(cmfs.ctypes.data),
                                     C.c_void_p(jump.ctypes.data), capacity, C.byref(vc), C.byref(pc))
        if rc != 0:
            raise RuntimeError(f"orc_sampler_read -> {rc}")
        return sub, cmfs[:vc.value].copy(), jump[:vc.value].copy(), vc.value, pc.value

    def counters(self):
        from_api = self.pkg.api.Counters()
        self.l.orc_get_counters(self.h, C.byref(from_api))
        return from_api.as_dict()

    def reset_counters(self):
        self.l.orc_reset_counters(self.h)

    def trace_closest(self, rays):
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = r.shape[0]
        t = np.zeros(n, np.float32)
        tri = np.zeros(n, np.int32)
        uv = np.zeros((n, 2), np.float32)
        self.l.orc_trace_closest(self.h, C.c_void_p(r.ctypes.data), n, C.c_void_p(t.ctypes.data), C.c_void_p(tri.ctypes.data),
                                 C.c_void_p(uv.ctypes.data), self.nthreads)
        return t, tri, uv

    def trace_any(self, rays):
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = r.shape[0]
        vis = np.zeros(n, np.int32)
        self.l.orc_trace_any(self.h, C.c_void_p(r.ctypes.data), n, C.c_void_p(vis.ctypes.data), self.nthreads)
        return vis

    def set_uniform_lvc(self, on):
        """"plain BDPT": uniformSample (cuProg.h:283-289) instead of the two-stage subspace sampler"""
        self.l.orc_set_uniform_lvc(self.h, int(bool(on)))

    def eye_step(self, records):
        """One step of the eye walk per record (EYE_STEP_IN_DTYPE -> EYE_STEP_OUT_DTYPE), see orc_eye_step."""
        a = np.ascontiguousarray(records, dtype=EYE_STEP_IN_DTYPE)
        out = np.zeros(a.shape[0], EYE_STEP_OUT_DTYPE)
        self.l.orc_eye_step(self.h, C.c_void_p(a.ctypes.data), a.shape[0], C.c_void_p(out.ctypes.data))
        return out

    def connect(self, eye_vertices, light_vertices):
        a = np.ascontiguousarray(eye_vertices, dtype=EYE_VERTEX_DTYPE)
        b = np.ascontiguousarray(light_vertices, dtype=self.pkg.LIGHT_VERTEX_DTYPE)
        n = a.shape[0]
        rgb = np.zeros((n, 3), np.float32)
        w = np.zeros(n, np.float32)
        self.l.orc_connect(self.h, C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), n, C.c_void_p(rgb.ctypes.data),
                           C.c_void_p(w.ctypes.data))
        return rgb, w


def material_struct(d):
    p = _pkg()
    m = p.api.Material()
    m.base_color[:] = [float(x) for x in d.get("color", (1, 1, 1))]
    m.metallic = d.get("metallic", 0.0); m.roughness = d.get("roughness", 0.5); m.specular = d.get("specular", 0.5)
    m.specular_tint = d.get("specular_tint", 0.0); m.subsurface = d.get("subsurface", 0.0); m.sheen = d.get("sheen", 0.0)
    m.sheen_tint = d.get("sheen_tint", 0.5); m.clearcoat = d.get("clearcoat", 0.0); m.clearcoat_gloss = d.get("clearcoat_gloss", 1.0)
    m.albedo_tex = 0
    m.brdf = int(d.get("brdf", 0))
    return m


def bsdf_eval_pdf(mat: dict, nvl: np.ndarray):
    m = material_struct(mat)
    a = np.ascontiguousarray(nvl, dtype=np.float32).reshape(-1, 9)
    f = np.zeros((a.shape[0], 3), np.float32)
    pdf = np.zeros(a.shape[0], np.float32)
    lib().orc_bsdf_eval_pdf(C.byref(m), C.c_void_p(a.ctypes.data), a.shape[0], C.c_void_p(f.ctypes.data), C.c_void_p(pdf.ctypes.data))
    return f, pdf


def bsdf_sample(mat: dict, nv: np.ndarray, seeds: np.ndarray):
    m = material_struct(mat)
    a = np.ascontiguousarray(nv, dtype=np.float32).reshape(-1, 6)
    s = np.ascontiguousarray(seeds, dtype=np.uint32).copy()
    out = np.zeros((a.shape[0], 3), np.float32)
    lib().orc_bsdf_sample(C.byref(m), C.c_void_p(a.ctypes.data), C.c_void_p(s.ctypes.data), a.shape[0], C.c_void_p(out.ctypes.data))
    return out, s


def binary_sample(cmf: np.ndarray, seed: int):
    c = np.ascontiguousarray(cmf, dtype=np.float32)
    s = C.c_uint32(seed)
    pmf = C.c_float()
    idx = lib().orc_binary_sample(C.c_void_p(c.ctypes.data), c.shape[0], C.byref(s), C.byref(pmf))
    return idx, pmf.value, s.value


def tree_index(tree: np.ndarray, pnd: np.ndarray):
    p = _pkg()
    t = np.ascontiguousarray(tree, dtype=p.TREE_NODE_DTYPE)
    a = np.ascontiguousarray(pnd, dtype=np.float32).reshape(-1, 9)
    out = np.zeros(a.shape[0], np.int32)
    lib().orc_tree_index(C.c_void_p(t.ctypes.data), t.shape[0], C.c_void_p(a.ctypes.data), a.shape[0], C.c_void_p(out.ctypes.data))
    return out


# ---- preprocessing (oracle/preprocess_ref.h) --------------------------------------------------------------------
def _pre_methods():
    def pretrace(self, iteration, num_core=10000, padding=10, nthreads=None):
        return self.l.orc_pretrace(self.h, iteration, num_core, padding, nthreads or self.nthreads)

    def train_records(self):
        p = self.pkg
        a, b = C.c_int(), C.c_int()
        self.l.orc_train_records_count(self.h, C.byref(a), C.byref(b))
        paths = np.zeros(a.value, dtype=p.PRETRACE_PATH_DTYPE)
        nodes = np.zeros(b.value, dtype=p.PRETRACE_NODE_DTYPE)
        self.l.orc_train_records_read(self.h, C.c_void_p(paths.ctypes.data), C.c_void_p(nodes.ctypes.data))
        return paths, nodes

    def train_records_import(self, paths, nodes):
        p = self.pkg
        pa = np.ascontiguousarray(paths, dtype=p.PRETRACE_PATH_DTYPE)
        no = np.ascontiguousarray(nodes, dtype=p.PRETRACE_NODE_DTYPE)
        self.l.orc_train_records_import(self.h, C.c_void_p(pa.ctypes.data), pa.shape[0], C.c_void_p(no.ctypes.data), no.shape[0])

    def train_records_clear(self):
        self.l.orc_train_records_clear(self.h)

    def preprocess_stage(self, stage, arg=0, nthreads=None):
        rc = self.l.orc_preprocess_stage(self.h, stage, arg, nthreads or self.nthreads)
        if rc != 0:
            raise RuntimeError(f"orc_preprocess_stage({stage}) -> {rc}")

    def get_gamma(self):
        n = self.pkg.NUM_SUBSPACE
        g = np.zeros((n, n), np.float32)
        self.l.orc_get_gamma(self.h, C.c_void_p(g.ctypes.data))
        return g

    def get_q(self):
        q = np.zeros(self.pkg.NUM_SUBSPACE, np.float32)
        self.l.orc_get_q(self.h, C.c_void_p(q.ctypes.data))
        return q

    def get_cmf_gamma(self):
        n = self.pkg.NUM_SUBSPACE
        g = np.zeros((n, n), np.float32)
        self.l.orc_get_cmf_gamma(self.h, C.c_void_p(g.ctypes.data))
        return g

    def get_tree(self, light: bool):
        n = C.c_int()
        self.l.orc_get_tree(self.h, int(light), None, 0, C.byref(n))
        t = np.zeros(n.value, dtype=self.pkg.TREE_NODE_DTYPE)
        self.l.orc_get_tree(self.h, int(light), C.c_void_p(t.ctypes.data), n.value, C.byref(n))
        return t

    def preprocess(self, target_paths, target_q_paths, train=True, num_core=10000, batch=20000):
        self.train_records_clear()
        it = 0
        while self.train_records_count() < target_paths:
            it += 1
            self.pretrace(it, num_core)
        n_train = target_paths // batch * batch if (train and target_paths >= batch) else target_paths
        self.preprocess_stage(1)
        self.preprocess_stage(2, target_q_paths)
        self.preprocess_stage(3, n_train)
        if train:
            self.preprocess_stage(4, min(batch, n_train))
        self.preprocess_stage(5)

    def train_records_count(self):
        a, b = C.c_int(), C.c_int()
        self.l.orc_train_records_count(self.h, C.byref(a), C.byref(b))
        return a.value

    for f in (pretrace, train_records, train_records_import, train_records_clear, preprocess_stage, get_gamma, get_q, get_cmf_gamma,
              get_tree, preprocess, train_records_count):
        setattr(Oracle, f.__name__, f)


_pre_methods()


def build_tree(samples: np.ndarray, subspace_size: int, label_bias: int = 0):
    """samples: (n, 10) float32 rows of position(3), dir(3), normal(3), weight."""
    p = _pkg()
    s = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1, 10)
    cap = max(16, 9 * 40 * s.shape[0])
    out = np.zeros(cap, dtype=p.TREE_NODE_DTYPE)
    n = C.c_int()
    rc = lib().orc_build_tree(C.c_void_p(s.ctypes.data), s.shape[0], subspace_size, label_bias, C.c_void_p(out.ctypes.data), cap, C.byref(n))
    if rc != 0:
        raise RuntimeError(f"orc_build_tree -> {rc}")
    return out[:n.value].copy()
