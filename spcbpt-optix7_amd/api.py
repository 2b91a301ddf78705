"""ctypes binding of include/spcbpt.h — the C ABI of the MI355X SPCBPT hot path.

The host-side mirror of the reference's driver (optixPathTracer.cpp:491-635):
``Renderer.launch("light trace" | "SPCBPT_eye" | "pt" | "pretrace")`` keeps the
four names `sutil::Scene::switchRaygen` dispatches on (sutil/Scene.cpp:1642-1789).
Nothing here falls back to a CPU path: if libspcbpt_hip.so is missing or no HIP
device is present the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

NUM_SUBSPACE = 1000
NUM_SUBSPACE_LIGHTSOURCE = 200
CONNECTION_N = 3

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libspcbpt_hip.so")
# the opt-in approximate-arithmetic build of the same sources (csrc/Makefile: libspcbpt_hip_fast.so; spcbpt_build_arithmetic() == "approx").
# A process picks it with SPCBPT_LIB=<this path> before the first load_library(); never the default.
FAST_LIB_PATH = os.path.join(_HERE, "csrc", "libspcbpt_hip_fast.so")


class Material(C.Structure):
    _fields_ = [("base_color", C.c_float * 3), ("metallic", C.c_float), ("roughness", C.c_float),
                ("specular", C.c_float), ("specular_tint", C.c_float), ("subsurface", C.c_float),
                ("sheen", C.c_float), ("sheen_tint", C.c_float), ("clearcoat", C.c_float),
                ("clearcoat_gloss", C.c_float), ("albedo_tex", C.c_int32), ("brdf", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class QuadLight(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3),
                ("emission", C.c_float * 3), ("div_level", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("texcoords", C.c_void_p), ("n_vertices", C.c_int32),
                ("indices", C.c_void_p), ("tri_material", C.c_void_p), ("n_triangles", C.c_int32),
                ("materials", C.POINTER(Material)), ("n_materials", C.c_int32),
                ("textures", C.POINTER(Texture)), ("n_textures", C.c_int32),
                ("lights", C.POINTER(QuadLight)), ("n_lights", C.c_int32)]


class TreeNode(C.Structure):
    _fields_ = [("mid", C.c_float * 3), ("child", C.c_int32 * 8), ("label", C.c_int32),
                ("type", C.c_int32), ("leaf", C.c_int32)]


class LightTraceParams(C.Structure):
    _fields_ = [("num_core", C.c_int32), ("core_padding", C.c_int32), ("m_per_core", C.c_int32),
                ("core_begin", C.c_int32), ("core_count", C.c_int32), ("decorrelate_bsdf_stream", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "closest_rays", "shadow_rays", "node_visits", "tri_tests", "surface_vertices", "textured_hits",
        "tree_nodes", "cmf_probes", "connections", "gamma_q_reads", "lvc_stores", "pixel_samples",
        "eye_paths", "light_paths")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


# numpy views of the POD records
LIGHT_VERTEX_DTYPE = np.dtype([
    ("position", "<f4", 3), ("pdf", "<f4"), ("normal", "<f4", 3), ("single_pdf", "<f4"),
    ("flux", "<f4", 3), ("rmis_pointer", "<f4"), ("color", "<f4", 3), ("last_lum", "<f4"),
    ("last_position", "<f4", 3), ("last_normal_projection", "<f4"),
    ("material_id", "<i2"), ("subspace_id", "<i2"), ("depth", "<i2"), ("last_zone_id", "<i2"),
    ("path_id", "<u4"), ("pad", "<u4")])
assert LIGHT_VERTEX_DTYPE.itemsize == 96
SUBSPACE_DTYPE = np.dtype([("jump_bias", "<i4"), ("id", "<i4"), ("size", "<i4"), ("sum_pmf", "<f4"), ("q", "<f4")])
TREE_NODE_DTYPE = np.dtype([("mid", "<f4", 3), ("child", "<i4", 8), ("label", "<i4"), ("type", "<i4"), ("leaf", "<i4")])
assert TREE_NODE_DTYPE.itemsize == C.sizeof(TreeNode)

PRETRACE_PATH_DTYPE = np.dtype([("contri", "<f4", 3), ("sample_pdf", "<f4"), ("fix_pdf", "<f4"), ("begin_ind", "<i4"),
                                ("end_ind", "<i4"), ("choice_id", "<i4"), ("pixel_id", "<i4", 2), ("valid", "<i4"), ("pad", "<i4")])
PRETRACE_NODE_DTYPE = np.dtype([("a_position", "<f4", 3), ("b_position", "<f4", 3), ("a_dir", "<f4", 3), ("b_dir", "<f4", 3),
                                ("a_normal", "<f4", 3), ("b_normal", "<f4", 3), ("peak_pdf", "<f4"), ("path_id", "<i4"),
                                ("label_a", "<i4"), ("label_b", "<i4"), ("valid", "<i4"), ("light_source", "<i4")])
assert PRETRACE_PATH_DTYPE.itemsize == 48 and PRETRACE_NODE_DTYPE.itemsize == 96

# Algorithmic byte constants of SURVEY.md 8(d) (fixed by the survey, not tuned).
BYTES = dict(node=64, tri=48, hit=72, mat=144, tex=16, tree=56, cmf=4, sub=20, jump=4, lvc=120, gq=4, lvcw=121, fb=36)


# The same events at the record sizes THIS build fetches (csrc/layout.h): 64-B quantised 4-wide node, 48-B triangle test (+ the
# fourth quad for the hit), 64-B material, 16-B classifier node, 16-B subspace record, 96-B light vertex (32 B of it for the
# shadow ray alone), 16-B radiance store + the film merge (16 R + 16 W + 4 W).  Reported next to the contract number, which
# credits the kernel with bytes it no longer needs.
BYTES_ACTUAL = dict(node=64, tri=48, hit=64, mat=64, tex=16, tree=16, cmf=4, sub=16, jump=4, lvc=96, gq=4, lvcw=96, fb=52)


def source_hash() -> str:
    """csrc/source_hash.py: the hash the Makefile embeds in the library (spcbpt_build_source_hash)."""
    import runpy
    return runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "source_hash.py"))["source_hash"]()


def kernel_hash() -> str:
    """csrc/source_hash.py: hash of the eye megakernel's device sources (what profiles/traffic_latest.json is valid for)."""
    import runpy
    return runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "source_hash.py"))["kernel_hash"]()


def algorithmic_bytes(c: dict, table: Optional[dict] = None) -> int:
    """bytes = sum_e count_e * B_e  (SURVEY.md 8(d))."""
    b = table or BYTES
    return (c["node_visits"] * b["node"] + c["tri_tests"] * b["tri"] + c["surface_vertices"] * (b["hit"] + b["mat"])
            + c["textured_hits"] * b["tex"] + c["tree_nodes"] * b["tree"] + c["cmf_probes"] * b["cmf"]
            + c["connections"] * (b["sub"] + b["jump"] + b["lvc"] + 2 * b["mat"]) + c["gamma_q_reads"] * b["gq"]
            + c["lvc_stores"] * b["lvcw"] + c["pixel_samples"] * b["fb"])


@dataclass
class Scene:
    """Flat triangle soup + materials + quad lights, as the C ABI takes it."""
    vertices: np.ndarray            # (nv, 3) f32
    indices: np.ndarray             # (nt, 3) u32
    tri_material: np.ndarray        # (nt,) i32
    materials: List[dict]
    lights: List[dict]
    texcoords: Optional[np.ndarray] = None   # (nv, 2) f32
    textures: List[np.ndarray] = field(default_factory=list)  # (h, w, 4) u8
    camera: dict = field(default_factory=dict)  # eye, lookat, up, fov
    name: str = "scene"
    environment: Optional[dict] = None   # rgba (h, w, 4) f32 as a .hdr stores it (row 0 = top), center (3,), radius: Renderer.set_environment(**scene.environment)

    def desc(self):
        """Returns (SceneDesc, keepalive)."""
        v = np.ascontiguousarray(self.vertices, dtype=np.float32)
        i = np.ascontiguousarray(self.indices, dtype=np.uint32)
        m = np.ascontiguousarray(self.tri_material, dtype=np.int32)
        assert v.ndim == 2 and v.shape[1] == 3 and i.ndim == 2 and i.shape[1] == 3 and m.shape[0] == i.shape[0]
        assert int(i.max(initial=0)) < v.shape[0] and int(m.max(initial=0)) < len(self.materials)
        t = None if self.texcoords is None else np.ascontiguousarray(self.texcoords, dtype=np.float32)
        mats = (Material * max(1, len(self.materials)))()
        for k, d in enumerate(self.materials):
            mm = mats[k]
            mm.base_color[:] = [float(x) for x in d.get("color", (1, 1, 1))]
            mm.metallic = d.get("metallic", 0.0)
            mm.roughness = d.get("roughness", 0.5)
            # q17: the .scene hand-off keeps MaterialData() defaults for the other Disney parameters
            mm.specular = d.get("specular", 0.5)
            mm.specular_tint = d.get("specular_tint", 0.0)
            mm.subsurface = d.get("subsurface", 0.0)
            mm.sheen = d.get("sheen", 0.0)
            mm.sheen_tint = d.get("sheen_tint", 0.5)
            mm.clearcoat = d.get("clearcoat", 0.0)
            mm.clearcoat_gloss = d.get("clearcoat_gloss", 1.0)
            mm.albedo_tex = d.get("albedo_tex", 0)
            mm.brdf = int(d.get("brdf", 0))   # Pbr::brdf (MaterialData.h:99): `brdf <int>` of a .scene material block
            assert 0 <= mm.albedo_tex <= len(self.textures)
        texs = (Texture * max(1, len(self.textures)))()
        tex_keep = []
        for k, im in enumerate(self.textures):
            im = np.ascontiguousarray(im, dtype=np.uint8)
            assert im.ndim == 3 and im.shape[2] == 4
            tex_keep.append(im)
            texs[k].rgba = im.ctypes.data
            texs[k].height, texs[k].width = im.shape[0], im.shape[1]
        ls = (QuadLight * max(1, len(self.lights)))()
        for k, d in enumerate(self.lights):
            ls[k].position[:] = [float(x) for x in d["position"]]
            ls[k].u[:] = [float(x) for x in d["u"]]
            ls[k].v[:] = [float(x) for x in d["v"]]
            ls[k].emission[:] = [float(x) for x in d["emission"]]
            ls[k].div_level = int(d.get("div_level", 1))
        assert sum(int(d.get("div_level", 1)) ** 2 for d in self.lights) <= NUM_SUBSPACE_LIGHTSOURCE
        sd = SceneDesc()
        sd.vertices = v.ctypes.data
        sd.texcoords = None if t is None else t.ctypes.data
        sd.n_vertices = v.shape[0]
        sd.indices = i.ctypes.data
        sd.tri_material = m.ctypes.data
        sd.n_triangles = i.shape[0]
        sd.materials = mats
        sd.n_materials = len(self.materials)
        sd.textures = texs
        sd.n_textures = len(self.textures)
        sd.lights = ls
        sd.n_lights = len(self.lights)
        return sd, (v, i, m, t, mats, texs, tex_keep, ls)


def load_scene_file(scene_path: str, data_root: str):
    """Parses a reference-format `.scene` file with the library's C++ loader (spcbpt_scene_file_load) and copies the
    result into a Scene.  Returns (scene, warnings)."""
    lib = load_library()
    h = C.c_void_p()
    rc = lib.spcbpt_scene_file_load(scene_path.encode(), data_root.encode(), C.byref(h))
    if rc != 0:
        raise SpcbptError(f"spcbpt_scene_file_load({scene_path}) failed ({rc})")
    return _scene_from_handle(lib, h, os.path.basename(scene_path))


def load_gltf(path: str, lights=None):
    """Reads a glTF 2.0 file (.gltf / .glb) with the library's C++ reader (spcbpt_gltf_load).  `lights` (list of quad-light
    dicts) replaces whatever the file's `extras.spcbpt_quad_lights` holds.  Returns (scene, warnings)."""
    lib = load_library()
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = lib.spcbpt_gltf_load(path.encode(), C.byref(h), err, 512)
    if rc != 0:
        raise SpcbptError(f"spcbpt_gltf_load({path}) failed ({rc}): {err.value.decode()}")
    scene, warn = _scene_from_handle(lib, h, os.path.basename(path))
    if lights is not None:
        scene.lights = list(lights)
    return scene, warn


def _scene_from_handle(lib, h, name):
    try:
        d = SceneDesc()
        lib.spcbpt_scene_file_desc(h, C.byref(d))
        nv, nt = d.n_vertices, d.n_triangles
        as_np = lambda ptr, ct, n: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).copy() if n else np.zeros(0, ct)
        V = as_np(d.vertices, C.c_float, 3 * nv).reshape(-1, 3)
        UV = as_np(d.texcoords, C.c_float, 2 * nv).reshape(-1, 2)
        I = as_np(d.indices, C.c_uint32, 3 * nt).reshape(-1, 3)
        M = as_np(d.tri_material, C.c_int32, nt)
        mats = []
        for k in range(d.n_materials):
            m = d.materials[k]
            mats.append(dict(color=tuple(m.base_color), metallic=m.metallic, roughness=m.roughness, albedo_tex=m.albedo_tex, brdf=m.brdf))
        lights = []
        for k in range(d.n_lights):
            l = d.lights[k]
            lights.append(dict(position=tuple(l.position), u=tuple(l.u), v=tuple(l.v), emission=tuple(l.emission), div_level=l.div_level))
        texs = []
        for k in range(d.n_textures):
            t = d.textures[k]
            texs.append(np.ctypeslib.as_array(C.cast(t.rgba, C.POINTER(C.c_uint8)), shape=(t.height, t.width, 4)).copy())
        eye, look, up = (np.zeros(3, np.float32) for _ in range(3))
        fov, w, hh = C.c_float(), C.c_int(), C.c_int()
        lib.spcbpt_scene_file_camera(h, _fp(eye), _fp(look), _fp(up), C.byref(fov), C.byref(w), C.byref(hh))
        cam = dict(eye=tuple(eye), lookat=tuple(look), up=tuple(up), fov=fov.value, width=w.value, height=hh.value)
        env = None
        ep, ew, eh, er = C.c_void_p(), C.c_int(), C.c_int(), C.c_float()
        ec = np.zeros(3, np.float32)
        if hasattr(lib, "spcbpt_scene_file_environment"):    # (absent only from an older SPCBPT_LIB build)
            lib.spcbpt_scene_file_environment(h, C.byref(ep), C.byref(ew), C.byref(eh), _fp(ec), C.byref(er))
        if ew.value > 0:
            px = np.ctypeslib.as_array(C.cast(ep, C.POINTER(C.c_float)), shape=(eh.value, ew.value, 4)).copy()
            env = dict(rgba=px, center=ec.copy(), radius=float(er.value))
        warn = lib.spcbpt_scene_file_warnings(h).decode()
        return Scene(vertices=V, indices=I, tri_material=M, materials=mats, lights=lights, texcoords=UV, textures=texs,
                     camera=cam, name=name, environment=env), warn
    finally:
        lib.spcbpt_scene_file_free(h)


def camera_frame(eye, lookat, up, fov_y_deg, aspect):
    """sutil::Camera::UVWFrame (sutil/Camera.cpp:34-45) in float32."""
    f = np.float32
    eye, lookat, up = (np.asarray(a, dtype=f) for a in (eye, lookat, up))
    W = lookat - eye
    wlen = np.sqrt(np.dot(W, W), dtype=f)
    U = np.cross(W, up).astype(f)
    U = U * (f(1.0) / np.sqrt(np.dot(U, U), dtype=f))
    V = np.cross(U, W).astype(f)
    V = V * (f(1.0) / np.sqrt(np.dot(V, V), dtype=f))
    vlen = wlen * f(np.tan(f(0.5) * f(fov_y_deg) * f(np.pi) / f(180.0)))
    V = V * vlen
    U = U * (vlen * f(aspect))
    return U.astype(f), V.astype(f), W.astype(f)


def single_leaf_tree(label=0):
    t = np.zeros(1, dtype=TREE_NODE_DTYPE)
    t[0]["leaf"] = 1
    t[0]["label"] = label
    return t


class SpcbptError(RuntimeError):
    pass


_lib = None


class ViewerState(C.Structure):  # spcbpt_viewer_state
    _fields_ = [("eye", C.c_float * 3), ("lookat", C.c_float * 3), ("up", C.c_float * 3),
                ("U", C.c_float * 3), ("V", C.c_float * 3), ("W", C.c_float * 3),
                ("fov_y", C.c_float), ("aspect", C.c_float), ("width", C.c_int32), ("height", C.c_int32),
                ("subframe_index", C.c_uint32), ("alg_id", C.c_int32), ("should_close", C.c_int32),
                ("one_frame_render_only", C.c_int32), ("camera_changed", C.c_int32), ("render_fps", C.c_float)]


KEY = {"ESCAPE": 256, "SPACE": 32, "C": 67, "G": 71, "P": 80, "W": 87}   # GLFW key codes the reference's keyCallback tests
BUTTON = {"left": 0, "right": 1, "middle": 2}


class Viewer:
    """Mirror of the spcbpt_viewer_* entry points (row f3): the reference application's event handling and render loop
    without a window.  `renderer` may be None (state machine only, no GPU)."""

    def __init__(self, renderer, eye, lookat, up, fov_y, width, height):
        self.lib = load_library()
        self.renderer = renderer
        self.h = C.c_void_p()
        rc = self.lib.spcbpt_viewer_create(renderer.h if renderer is not None else None, _fp(np.asarray(eye, np.float32)),
                                           _fp(np.asarray(lookat, np.float32)), _fp(np.asarray(up, np.float32)),
                                           C.c_float(fov_y), width, height, C.byref(self.h))
        if rc:
            raise SpcbptError(f"viewer_create failed ({rc})")

    def close(self):
        if self.h:
            self.lib.spcbpt_viewer_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc:
            msg = self.lib.spcbpt_last_error(self.renderer.h) if self.renderer is not None else None
            raise SpcbptError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def mouse_button(self, button, action, x, y):
        self._chk(self.lib.spcbpt_viewer_mouse_button(self.h, BUTTON.get(button, button), int(action), float(x), float(y)), "mouse_button")

    def cursor_pos(self, x, y):
        self._chk(self.lib.spcbpt_viewer_cursor_pos(self.h, float(x), float(y)), "cursor_pos")

    def scroll(self, yscroll):
        self._chk(self.lib.spcbpt_viewer_scroll(self.h, 0.0, float(yscroll)), "scroll")

    def window_size(self, w, h):
        self._chk(self.lib.spcbpt_viewer_window_size(self.h, int(w), int(h)), "window_size")

    def iconify(self, on):
        self._chk(self.lib.spcbpt_viewer_iconify(self.h, int(on)), "iconify")

    def key(self, key, action=1):
        self._chk(self.lib.spcbpt_viewer_key(self.h, KEY.get(key, key), int(action)), "key")

    def set_fps(self, fps):
        self._chk(self.lib.spcbpt_viewer_set_fps(self.h, C.c_float(fps)), "set_fps")

    def set_light_ahead(self, on):
        self._chk(self.lib.spcbpt_viewer_set_light_ahead(self.h, int(on)), "viewer_set_light_ahead")

    def set_pipeline(self, mode: int):
        """0 the reference's order, 1 next light pass beside the eye kernel, 2 (default) next frame traced while this one is shown."""
        self._chk(self.lib.spcbpt_viewer_set_pipeline(self.h, int(mode)), "viewer_set_pipeline")

    def frame(self):
        self._chk(self.lib.spcbpt_viewer_frame(self.h), "viewer_frame")

    def state(self):
        s = ViewerState()
        self._chk(self.lib.spcbpt_viewer_get_state(self.h, C.byref(s)), "get_state")
        d = {k: (np.array(getattr(s, k), dtype=np.float32) if k in ("eye", "lookat", "up", "U", "V", "W") else getattr(s, k))
             for k, _ in ViewerState._fields_}
        d["alg"] = self.lib.spcbpt_viewer_alg_name(s.alg_id).decode()
        return d

    def replay(self, events):
        """events as in oracle/ref_viewer.cpp: rows (type, a, b, c) with 0 press(button, x, y), 1 release(button),
        2 cursor(x, y), 3 scroll(yscroll), 4 key W at render_fps a.  Returns the camera (eye, lookat, up, U, V, W) after each."""
        out = np.zeros((len(events), 18), np.float32)
        for i, (t, a, b, c) in enumerate(np.asarray(events, np.float64)):
            t = int(t)
            if t == 0: self.mouse_button(int(a), 1, b, c)
            elif t == 1: self.mouse_button(int(a), 0, b, c)
            elif t == 2: self.cursor_pos(b, c)
            elif t == 3: self.scroll(a)
            elif t == 4:
                self.set_fps(a)
                self.key("W", 1)
            s = self.state()
            out[i] = np.concatenate([s["eye"], s["lookat"], s["up"], s["U"], s["V"], s["W"]])
        return out


def hdr_load(path: str):
    """(h, w, 4) float32 raster of a Radiance .hdr file, decoded as the reference's HDRLoader (scene_shift.cpp:334-500); row 0 = top."""
    lib = load_library()
    w, h = C.c_int(), C.c_int()
    rc = lib.spcbpt_hdr_load(os.fsencode(path), C.byref(w), C.byref(h), None, 0)
    if rc:
        raise SpcbptError(f"hdr_load({path}) failed ({rc})")
    px = np.zeros((h.value, w.value, 4), np.float32)
    rc = lib.spcbpt_hdr_load(os.fsencode(path), C.byref(w), C.byref(h), px.ctypes.data, px.size)
    if rc:
        raise SpcbptError(f"hdr_load({path}) failed ({rc})")
    return px


def image_load(path: str):
    """RGBA8 image (h, w, 4) of a JPEG / PNG / binary PPM file, decoded as the reference's stbi_load(..., STBI_rgb_alpha)."""
    lib = load_library()
    w, h = C.c_int(), C.c_int()
    rc = lib.spcbpt_image_load(os.fsencode(path), C.byref(w), C.byref(h), None, 0)
    if rc:
        raise SpcbptError(f"image_load({path}) failed ({rc})")
    px = np.zeros((h.value, w.value, 4), np.uint8)
    rc = lib.spcbpt_image_load(os.fsencode(path), C.byref(w), C.byref(h), px.ctypes.data, px.nbytes)
    if rc:
        raise SpcbptError(f"image_load({path}) failed ({rc})")
    return px


def checkpoint_write(directory: str, eye_tree, light_tree, q, gamma):
    """Context-free writer of the reference's checkpoint files (spcbpt_checkpoint_write)."""
    lib = load_library()
    et = np.ascontiguousarray(eye_tree, dtype=TREE_NODE_DTYPE)
    lt = np.ascontiguousarray(light_tree, dtype=TREE_NODE_DTYPE)
    q = np.ascontiguousarray(q, dtype=np.float32)
    g = np.ascontiguousarray(gamma, dtype=np.float32)
    assert q.size == NUM_SUBSPACE and g.size == NUM_SUBSPACE * NUM_SUBSPACE
    rc = lib.spcbpt_checkpoint_write(os.fsencode(directory), et.ctypes.data, et.shape[0], lt.ctypes.data, lt.shape[0], q.ctypes.data, g.ctypes.data)
    if rc:
        raise SpcbptError(f"checkpoint_write failed ({rc})")


def checkpoint_read(directory: str, current_gamma=None, cap=1 << 20):
    """Context-free reader (tree_load + load_Q_file + load_Gamma_file): returns (eye_tree, light_tree, Q, Gamma).  With
    `current_gamma` the emitter-subspace columns keep its values, as load_Gamma_file does."""
    lib = load_library()
    et = np.zeros(cap, dtype=TREE_NODE_DTYPE)
    lt = np.zeros(cap, dtype=TREE_NODE_DTYPE)
    q = np.zeros(NUM_SUBSPACE, dtype=np.float32)
    g = np.zeros(NUM_SUBSPACE * NUM_SUBSPACE, dtype=np.float32) if current_gamma is None else \
        np.ascontiguousarray(current_gamma, dtype=np.float32).reshape(-1).copy()
    ne, nl = C.c_int(), C.c_int()
    rc = lib.spcbpt_checkpoint_read(os.fsencode(directory), et.ctypes.data, C.byref(ne), cap, lt.ctypes.data, C.byref(nl), cap,
                                    q.ctypes.data, g.ctypes.data, 0 if current_gamma is None else 1)
    if rc:
        raise SpcbptError(f"checkpoint_read failed ({rc})")
    return et[:ne.value].copy(), lt[:nl.value].copy(), q, g.reshape(NUM_SUBSPACE, NUM_SUBSPACE)


def gamma_to_cmf(gamma):
    lib = load_library()
    g = np.ascontiguousarray(gamma, dtype=np.float32).reshape(-1)
    out = np.zeros_like(g)
    lib.spcbpt_gamma_to_cmf(g.ctypes.data, out.ctypes.data)
    return out.reshape(NUM_SUBSPACE, NUM_SUBSPACE)


def load_library(path: str = LIB_PATH):
    """Loads libspcbpt_hip.so.  Raises if it has not been built — there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if os.environ.get("SPCBPT_LIB"):   # developer A/B runs: another build of the same ABI (no source-hash check)
        path = os.environ["SPCBPT_LIB"]
    if not os.path.exists(path):
        raise SpcbptError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's).
    # Importing torch first makes this library bind to the runtime torch uses, so torch.distributed (RCCL) buffers
    # and the context's buffers live in the same HIP context.  Without torch the system runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)   # libspcbpt_mgpu.so resolves the C ABI against this handle
    # a library built from other sources than the tree holds must not be tested or benchmarked silently
    try:
        lib.spcbpt_build_source_hash.restype = C.c_char_p
        built = lib.spcbpt_build_source_hash().decode()
    except AttributeError:
        built = "none"
    if path in (LIB_PATH, FAST_LIB_PATH) and built != source_hash() and not os.environ.get("SPCBPT_ALLOW_STALE_LIB"):
        raise SpcbptError(f"{path} was built from other sources (library {built}, tree {source_hash()}): run `make -C spcbpt-optix7_amd/csrc`")
    vp, i32, u32, f32p = C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_float)
    sig = {
        "spcbpt_create": [C.POINTER(SceneDesc), i32, C.POINTER(vp)],
        "spcbpt_destroy": [vp],
        "spcbpt_set_camera": [vp, f32p, f32p, f32p, f32p],
        "spcbpt_set_camera_lookat": [vp, f32p, f32p, f32p, C.c_float, C.c_float],
        "spcbpt_resize": [vp, i32, i32],
        "spcbpt_set_subspace": [vp, vp, i32, vp, i32, vp, vp],
        "spcbpt_set_light_trace": [vp, C.POINTER(LightTraceParams)],
        "spcbpt_launch": [vp, C.c_char_p, u32, i32, i32, i32],
        "spcbpt_build_sampler": [vp],
        "spcbpt_build_sampler_batch": [vp, C.c_int],
        "spcbpt_launch_eye_batch": [vp, i32, C.POINTER(u32), i32, i32, i32],
        "spcbpt_launch_light_batch": [vp, u32, i32],
        "spcbpt_lvc_export": [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32)],
        "spcbpt_lvc_import": [vp, vp, i32, i32],
        "spcbpt_set_environment": [vp, vp, i32, i32, vp, C.c_float],
        "spcbpt_get_environment": [vp, C.POINTER(i32), C.POINTER(i32), f32p, C.POINTER(C.c_float), C.POINTER(i32)],
        "spcbpt_hdr_load": [C.c_char_p, C.POINTER(i32), C.POINTER(i32), vp, C.c_size_t],
        "spcbpt_lvc_set_capacity": [vp, i32],
        "spcbpt_lvc_get_capacity": [vp, C.POINTER(i32), C.POINTER(i32)],
        "spcbpt_lvc_read": [vp, vp, i32, C.POINTER(i32)],
        "spcbpt_sampler_read": [vp, vp, vp, vp, i32, C.POINTER(i32), C.POINTER(i32)],
        "spcbpt_read_accum": [vp, vp],
        "spcbpt_read_frame": [vp, vp],
        "spcbpt_accum_device_ptr": [vp, C.POINTER(vp)],
        "spcbpt_clear_accum": [vp],
        "spcbpt_get_counters": [vp, C.POINTER(Counters)],
        "spcbpt_reset_counters": [vp],
        "spcbpt_debug_phase_clocks": [vp, C.POINTER(C.c_uint64)],
        "spcbpt_set_connection_sampler": [vp, i32],
        "spcbpt_debug_unit": [vp, i32, vp, i32, vp, i32, i32, vp, i32],
        "spcbpt_debug_trace_bench": [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp],
        "spcbpt_debug_spill_arm": [vp],
        "spcbpt_debug_spill_count": [vp, C.POINTER(C.c_uint64), C.POINTER(i32)],
        "spcbpt_enable_counters": [vp, i32],
        "spcbpt_stream": [vp, C.POINTER(vp)],
        "spcbpt_sync": [vp], "spcbpt_sync_film": [vp], "spcbpt_merge_deferred": [vp, i32], "spcbpt_launch_deferred": [vp, C.c_char_p, u32, i32, i32, i32],
        "spcbpt_sync_light": [vp],
        "spcbpt_set_light_ahead": [vp, i32],
        "spcbpt_get_pipeline_state": [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)],
        "spcbpt_reuse_sampler": [vp],
        "spcbpt_read_film": [vp, vp, vp],
        "spcbpt_debug_batch_scratch": [vp, C.POINTER(C.c_int64), C.POINTER(i32), C.POINTER(i32)],
        "spcbpt_debug_read_sampling_tables": [vp, vp, i32, vp, vp],
        "spcbpt_lvc_import_wait": [vp],
        "spcbpt_kernel_time": [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(i32)],
        "spcbpt_reset_kernel_time": [vp],
        "spcbpt_enable_kernel_timing": [vp, i32],
        "spcbpt_trace_closest": [vp, vp, i32, vp, vp, vp],
        "spcbpt_trace_any": [vp, vp, i32, vp],
        "spcbpt_preprocess": [vp, i32, i32, i32],
        "spcbpt_set_pretrace": [vp, i32, i32],
        "spcbpt_train_records_count": [vp, C.POINTER(i32), C.POINTER(i32)],
        "spcbpt_train_records_read": [vp, vp, i32, vp, i32],
        "spcbpt_train_records_import": [vp, vp, i32, vp, i32],
        "spcbpt_train_records_clear": [vp],
        "spcbpt_preprocess_stage": [vp, i32, i32],
        "spcbpt_get_gamma": [vp, vp],
        "spcbpt_checkpoint_write": [C.c_char_p, vp, i32, vp, i32, vp, vp],
        "spcbpt_checkpoint_read": [C.c_char_p, vp, C.POINTER(i32), i32, vp, C.POINTER(i32), i32, vp, vp, i32],
        "spcbpt_gamma_to_cmf": [vp, vp],
        "spcbpt_viewer_create": [vp, f32p, f32p, f32p, C.c_float, i32, i32, C.POINTER(vp)],
        "spcbpt_viewer_mouse_button": [vp, i32, i32, C.c_double, C.c_double],
        "spcbpt_viewer_cursor_pos": [vp, C.c_double, C.c_double],
        "spcbpt_viewer_scroll": [vp, C.c_double, C.c_double],
        "spcbpt_viewer_window_size": [vp, i32, i32],
        "spcbpt_viewer_iconify": [vp, i32],
        "spcbpt_viewer_key": [vp, i32, i32],
        "spcbpt_viewer_set_fps": [vp, C.c_float],
        "spcbpt_viewer_set_light_ahead": [vp, i32], "spcbpt_viewer_set_pipeline": [vp, i32],
        "spcbpt_viewer_frame": [vp],
        "spcbpt_viewer_get_state": [vp, C.POINTER(ViewerState)],
        "spcbpt_image_load": [C.c_char_p, C.POINTER(i32), C.POINTER(i32), vp, C.c_size_t],
        "spcbpt_checkpoint_save": [vp, C.c_char_p],
        "spcbpt_checkpoint_load": [vp, C.c_char_p],
        "spcbpt_gltf_load": [C.c_char_p, C.POINTER(vp), C.c_char_p, i32],
        "spcbpt_scene_file_load": [C.c_char_p, C.c_char_p, C.POINTER(vp)],
        "spcbpt_scene_file_desc": [vp, C.POINTER(SceneDesc)],
        "spcbpt_scene_file_camera": [vp, f32p, f32p, f32p, f32p, C.POINTER(i32), C.POINTER(i32)],
        "spcbpt_scene_file_environment": [vp, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), f32p, C.POINTER(C.c_float)],
        "spcbpt_scene_file_free": [vp],
        "spcbpt_get_subspace": [vp, vp, C.POINTER(i32), i32, vp, C.POINTER(i32), i32, vp, vp],
        "spcbpt_scene_info": [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)],
    }
    for name, args in sig.items():
        if os.environ.get("SPCBPT_LIB") and not hasattr(lib, name):
            continue                   # an older build of the ABI in a developer A/B run
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.spcbpt_viewer_destroy.argtypes = [vp]
    lib.spcbpt_viewer_destroy.restype = None
    lib.spcbpt_viewer_alg_name.argtypes = [i32]
    lib.spcbpt_viewer_alg_name.restype = C.c_char_p
    lib.spcbpt_last_error.argtypes = [vp]
    lib.spcbpt_last_error.restype = C.c_char_p
    lib.spcbpt_build_arithmetic.restype = C.c_char_p
    lib.spcbpt_scene_file_warnings.argtypes = [vp]
    lib.spcbpt_scene_file_warnings.restype = C.c_char_p
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "spcbpt_create", "spcbpt_destroy", "spcbpt_last_error", "spcbpt_set_camera", "spcbpt_set_camera_lookat",
    "spcbpt_resize", "spcbpt_set_subspace", "spcbpt_set_light_trace", "spcbpt_launch", "spcbpt_launch_eye_batch", "spcbpt_launch_light_batch", "spcbpt_build_sampler", "spcbpt_build_sampler_batch",
    "spcbpt_lvc_export", "spcbpt_lvc_import", "spcbpt_lvc_set_capacity", "spcbpt_lvc_get_capacity", "spcbpt_set_environment", "spcbpt_get_environment", "spcbpt_hdr_load", "spcbpt_lvc_read", "spcbpt_sampler_read", "spcbpt_read_accum",
    "spcbpt_read_frame", "spcbpt_accum_device_ptr", "spcbpt_clear_accum", "spcbpt_get_counters",
    "spcbpt_reset_counters", "spcbpt_debug_phase_clocks", "spcbpt_debug_spill_arm", "spcbpt_debug_spill_count", "spcbpt_set_connection_sampler", "spcbpt_debug_unit", "spcbpt_debug_trace_bench",
    "spcbpt_build_source_hash", "spcbpt_build_arithmetic", "spcbpt_abi_struct_sizes", "spcbpt_lvc_export_on", "spcbpt_lvc_import_gathered", "spcbpt_lvc_export_batch_on", "spcbpt_lvc_import_gathered_batch", "spcbpt_film_pack_bands", "spcbpt_film_unpack_bands", "spcbpt_image_size", "spcbpt_get_light_trace", "spcbpt_enable_counters", "spcbpt_stream", "spcbpt_sync", "spcbpt_sync_light", "spcbpt_launch_deferred", "spcbpt_merge_deferred", "spcbpt_sync_film", "spcbpt_set_light_ahead", "spcbpt_get_pipeline_state", "spcbpt_reuse_sampler", "spcbpt_read_film", "spcbpt_debug_batch_scratch", "spcbpt_debug_read_sampling_tables", "spcbpt_lvc_import_wait", "spcbpt_kernel_time",
    "spcbpt_reset_kernel_time", "spcbpt_enable_kernel_timing", "spcbpt_trace_closest", "spcbpt_trace_any",
    "spcbpt_preprocess", "spcbpt_get_subspace", "spcbpt_scene_info", "spcbpt_set_pretrace", "spcbpt_train_records_count",
    "spcbpt_train_records_read", "spcbpt_train_records_import", "spcbpt_train_records_clear", "spcbpt_preprocess_stage",
    "spcbpt_get_gamma", "spcbpt_gltf_load", "spcbpt_scene_file_load", "spcbpt_scene_file_desc", "spcbpt_scene_file_camera",
    "spcbpt_scene_file_warnings", "spcbpt_scene_file_environment", "spcbpt_scene_file_free",
    "spcbpt_viewer_create", "spcbpt_viewer_destroy", "spcbpt_viewer_mouse_button", "spcbpt_viewer_cursor_pos", "spcbpt_viewer_scroll",
    "spcbpt_viewer_window_size", "spcbpt_viewer_iconify", "spcbpt_viewer_key", "spcbpt_viewer_set_fps", "spcbpt_viewer_set_light_ahead", "spcbpt_viewer_set_pipeline", "spcbpt_viewer_frame",
    "spcbpt_viewer_get_state", "spcbpt_viewer_alg_name",
    "spcbpt_image_load", "spcbpt_checkpoint_write", "spcbpt_checkpoint_read", "spcbpt_gamma_to_cmf", "spcbpt_checkpoint_save", "spcbpt_checkpoint_load",
]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Renderer:
    """One context per GPU (the `sutil::Scene` + `MyParams` pair of the reference driver)."""

    def __init__(self, scene: Scene, device: int = 0):
        self.lib = load_library()
        self.scene = scene
        sd, keep = scene.desc()
        h = C.c_void_p()
        rc = self.lib.spcbpt_create(C.byref(sd), device, C.byref(h))
        if rc != 0:
            msg = self.lib.spcbpt_last_error(None)
            raise SpcbptError(f"spcbpt_create failed ({rc}): {msg.decode() if msg else ''}")
        self.h = h
        self.width = self.height = 0
        self.lt = None

    def _chk(self, rc, what):
        if rc != 0:
            msg = self.lib.spcbpt_last_error(self.h)
            raise SpcbptError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.spcbpt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state --------------------------------------------------------------
    def set_camera(self, eye, U, V, W):
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (eye, U, V, W)]
        self._chk(self.lib.spcbpt_set_camera(self.h, *[_fp(x) for x in a]), "set_camera")

    def set_camera_lookat(self, eye, lookat, up, fov, aspect):
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (eye, lookat, up)]
        self._chk(self.lib.spcbpt_set_camera_lookat(self.h, *[_fp(x) for x in a], fov, aspect), "set_camera_lookat")

    def resize(self, w, h):
        self._chk(self.lib.spcbpt_resize(self.h, w, h), "resize")
        self.width, self.height = w, h

    def set_subspace(self, eye_tree=None, light_tree=None, q=None, cmf_gamma=None):
        if eye_tree is None and light_tree is None and q is None and cmf_gamma is None:
            self._chk(self.lib.spcbpt_set_subspace(self.h, None, 0, None, 0, None, None), "set_subspace")
            return
        et = np.ascontiguousarray(eye_tree, dtype=TREE_NODE_DTYPE)
        lt = np.ascontiguousarray(light_tree, dtype=TREE_NODE_DTYPE)
        q = np.ascontiguousarray(q, dtype=np.float32)
        g = np.ascontiguousarray(cmf_gamma, dtype=np.float32)
        assert q.size == NUM_SUBSPACE and g.size == NUM_SUBSPACE * NUM_SUBSPACE
        self._chk(self.lib.spcbpt_set_subspace(self.h, et.ctypes.data, et.shape[0], lt.ctypes.data, lt.shape[0],
                                               q.ctypes.data, g.ctypes.data), "set_subspace")

    def get_subspace(self, cap=1 << 20):
        et = np.zeros(cap, dtype=TREE_NODE_DTYPE)
        lt = np.zeros(cap, dtype=TREE_NODE_DTYPE)
        q = np.zeros(NUM_SUBSPACE, dtype=np.float32)
        g = np.zeros(NUM_SUBSPACE * NUM_SUBSPACE, dtype=np.float32)
        ne, nl = C.c_int(), C.c_int()
        self._chk(self.lib.spcbpt_get_subspace(self.h, et.ctypes.data, C.byref(ne), cap, lt.ctypes.data, C.byref(nl), cap,
                                               q.ctypes.data, g.ctypes.data), "get_subspace")
        return et[:ne.value].copy(), lt[:nl.value].copy(), q, g.reshape(NUM_SUBSPACE, NUM_SUBSPACE)

    def set_light_trace(self, num_core, core_padding, m_per_core, core_begin=0, core_count=0, decorrelate=None):
        if decorrelate is None:
            decorrelate = m_per_core == 1   # one path per lane: see spcbpt_light_trace_params in include/spcbpt.h
        p = LightTraceParams(num_core, core_padding, m_per_core, core_begin, core_count, int(decorrelate))
        self._chk(self.lib.spcbpt_set_light_trace(self.h, C.byref(p)), "set_light_trace")
        self.lt = p

    # -- launches -----------------------------------------------------------
    def launch(self, name: str, frame: int, rows=None):
        r0, r1, rs = rows if rows is not None else (0, self.height, 1)
        self._chk(self.lib.spcbpt_launch(self.h, name.encode(), frame, r0, r1, rs), f"launch({name})")

    def launch_light_batch(self, first_frame, n):
        """The light passes of launch frames first_frame .. first_frame + n - 1 as one persistent launch (spcbpt_launch_light_batch)."""
        self._chk(self.lib.spcbpt_launch_light_batch(self.h, first_frame, n), "launch_light_batch")

    def launch_eye_batch(self, subframes, rows=None):
        """One persistent eye kernel over the samplers of the last len(subframes) build_sampler calls (spcbpt_launch_eye_batch)."""
        sf = (C.c_uint32 * len(subframes))(*[int(v) for v in subframes])
        r0, r1, rs = rows if rows is not None else (0, self.height, 1)
        self._chk(self.lib.spcbpt_launch_eye_batch(self.h, len(subframes), sf, r0, r1, rs), "launch_eye_batch")

    def build_sampler(self):
        self._chk(self.lib.spcbpt_build_sampler(self.h), "build_sampler")

    def build_sampler_batch(self, n: int):
        """n build_sampler calls (the n oldest queued light passes) as one set of four launches (spcbpt_build_sampler_batch)."""
        self._chk(self.lib.spcbpt_build_sampler_batch(self.h, int(n)), "build_sampler_batch")

    def sync(self):
        self._chk(self.lib.spcbpt_sync(self.h), "sync")

    def launch_deferred(self, alg: str, subframe: int, rows=None):
        r0, r1, rs = rows if rows is not None else (0, self.height, 1)
        self._chk(self.lib.spcbpt_launch_deferred(self.h, alg.encode(), int(subframe), r0, r1, rs), "launch_deferred")

    def merge_deferred(self, keep: bool):
        self._chk(self.lib.spcbpt_merge_deferred(self.h, 1 if keep else 0), "merge_deferred")

    def sync_film(self):
        self._chk(self.lib.spcbpt_sync_film(self.h), "sync_film")

    def preprocess(self, target_paths=2_000_000, target_q_paths=2_000_000, train=True):
        self._chk(self.lib.spcbpt_preprocess(self.h, target_paths, target_q_paths, int(train)), "preprocess")

    def set_pretrace(self, num_core=10000, padding=10):
        self._chk(self.lib.spcbpt_set_pretrace(self.h, num_core, padding), "set_pretrace")

    def preprocess_stage(self, stage: int, arg: int = 0):
        self._chk(self.lib.spcbpt_preprocess_stage(self.h, stage, arg), f"preprocess_stage({stage})")

    def train_records(self):
        a, b = C.c_int(), C.c_int()
        self._chk(self.lib.spcbpt_train_records_count(self.h, C.byref(a), C.byref(b)), "train_records_count")
        paths = np.zeros(a.value, dtype=PRETRACE_PATH_DTYPE)
        nodes = np.zeros(b.value, dtype=PRETRACE_NODE_DTYPE)
        if a.value:
            self._chk(self.lib.spcbpt_train_records_read(self.h, paths.ctypes.data, a.value, nodes.ctypes.data, max(b.value, 1)),
                      "train_records_read")
        return paths, nodes

    def train_records_import(self, paths, nodes):
        pa = np.ascontiguousarray(paths, dtype=PRETRACE_PATH_DTYPE)
        no = np.ascontiguousarray(nodes, dtype=PRETRACE_NODE_DTYPE)
        self._chk(self.lib.spcbpt_train_records_import(self.h, pa.ctypes.data, pa.shape[0], no.ctypes.data, no.shape[0]),
                  "train_records_import")

    def train_records_clear(self):
        self._chk(self.lib.spcbpt_train_records_clear(self.h), "train_records_clear")

    def checkpoint_save(self, directory: str):
        """tree_eye.txt, tree_light.txt, Q.txt, E.txt in the reference's formats (needs a preprocessing run for Gamma)."""
        self._chk(self.lib.spcbpt_checkpoint_save(self.h, os.fsencode(directory)), "checkpoint_save")

    def checkpoint_load(self, directory: str):
        self._chk(self.lib.spcbpt_checkpoint_load(self.h, os.fsencode(directory)), "checkpoint_load")

    def get_gamma(self):
        g = np.zeros((NUM_SUBSPACE, NUM_SUBSPACE), dtype=np.float32)
        self._chk(self.lib.spcbpt_get_gamma(self.h, g.ctypes.data), "get_gamma")
        return g

    # -- readback -----------------------------------------------------------
    def image_size(self):
        """(width, height) as the library holds them (the viewer resizes the context itself)."""
        w, h = C.c_int32(), C.c_int32()
        self._chk(self.lib.spcbpt_image_size(self.h, C.byref(w), C.byref(h)), "image_size")
        self.width, self.height = int(w.value), int(h.value)
        return self.width, self.height

    def read_accum(self):
        self.image_size()
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self._chk(self.lib.spcbpt_read_accum(self.h, out.ctypes.data), "read_accum")
        return out

    def read_frame(self):
        self.image_size()
        out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self._chk(self.lib.spcbpt_read_frame(self.h, out.ctypes.data), "read_frame")
        return out

    def clear_accum(self):
        self._chk(self.lib.spcbpt_clear_accum(self.h), "clear_accum")

    def read_film(self, accum=True, frame=True):
        """(accum, frame) as of the last queued film merge, without waiting for launches queued behind it (spcbpt_read_film)."""
        self.image_size()
        a = np.zeros((self.height, self.width, 4), dtype=np.float32) if accum else None
        f = np.zeros((self.height, self.width, 4), dtype=np.uint8) if frame else None
        self._chk(self.lib.spcbpt_read_film(self.h, a.ctypes.data if accum else None, f.ctypes.data if frame else None), "read_film")
        return a, f

    def pipeline_state(self):
        v = [C.c_int32() for _ in range(4)]
        self._chk(self.lib.spcbpt_get_pipeline_state(self.h, *[C.byref(x) for x in v]), "get_pipeline_state")
        return dict(zip(("light_ahead", "pending_passes", "sampler_intact", "deferred"), (int(x.value) for x in v)))

    def reuse_sampler(self):
        self._chk(self.lib.spcbpt_reuse_sampler(self.h), "reuse_sampler")

    def batch_scratch(self):
        b, f, k = C.c_int64(), C.c_int32(), C.c_int32()
        self._chk(self.lib.spcbpt_debug_batch_scratch(self.h, C.byref(b), C.byref(f), C.byref(k)), "debug_batch_scratch")
        return {"bytes": int(b.value), "frames": int(f.value), "fallbacks": int(k.value)}

    def sampling_tables(self, vertex_count):
        """(guide2, guide1, gamma_q): the tables the eye kernel samples through (spcbpt_debug_read_sampling_tables)."""
        g2 = np.zeros(max(vertex_count, 1), dtype=np.uint32)
        g1 = np.zeros((NUM_SUBSPACE, 1024), dtype=np.uint16)
        gq = np.zeros((NUM_SUBSPACE, NUM_SUBSPACE), dtype=np.float32)
        self._chk(self.lib.spcbpt_debug_read_sampling_tables(self.h, g2.ctypes.data if vertex_count > 0 else None, len(g2), g1.ctypes.data, gq.ctypes.data),
                  "debug_read_sampling_tables")
        return g2[:vertex_count], g1, gq

    def lvc_read(self, capacity=None):
        if capacity is None:
            capacity = self.lt.num_core * self.lt.core_padding if self.lt else 1 << 22
        out = np.zeros(capacity, dtype=LIGHT_VERTEX_DTYPE)
        n = C.c_int()
        self._chk(self.lib.spcbpt_lvc_read(self.h, out.ctypes.data, capacity, C.byref(n)), "lvc_read")
        return out[:n.value].copy()

    def lvc_import(self, verts: np.ndarray):
        v = np.ascontiguousarray(verts, dtype=LIGHT_VERTEX_DTYPE)
        self._chk(self.lib.spcbpt_lvc_import(self.h, v.ctypes.data, v.shape[0], 0), "lvc_import")

    def lvc_import_device(self, d_ptr: int, count: int):
        self._chk(self.lib.spcbpt_lvc_import(self.h, C.c_void_p(d_ptr), count, 1), "lvc_import")

    def set_environment(self, rgba, center=None, radius=0.0):
        """The environment map as one more light (spcbpt_set_environment): rgba = (h, w, 4) float32 as the .hdr stores it (row 0 = top)."""
        a = np.ascontiguousarray(rgba, dtype=np.float32)
        assert a.ndim == 3 and a.shape[2] == 4
        c3 = None if center is None else np.ascontiguousarray(center, dtype=np.float32)
        self._chk(self.lib.spcbpt_set_environment(self.h, a.ctypes.data, a.shape[1], a.shape[0], None if c3 is None else c3.ctypes.data, float(radius)), "set_environment")

    def environment(self):
        w, h, n = C.c_int(), C.c_int(), C.c_int()
        c3 = np.zeros(3, np.float32)
        r = C.c_float()
        self._chk(self.lib.spcbpt_get_environment(self.h, C.byref(w), C.byref(h), _fp(c3), C.byref(r), C.byref(n)), "get_environment")
        return dict(width=w.value, height=h.value, center=c3, radius=r.value, n_lights=n.value)

    def lvc_set_capacity(self, vertices: int):
        """Vertices per buffer set, fixed by hand (0 = sized from a probe pass: spcbpt_lvc_set_capacity)."""
        self._chk(self.lib.spcbpt_lvc_set_capacity(self.h, int(vertices)), "lvc_set_capacity")

    def lvc_capacity(self):
        """(vertices per buffer set, number of sets) as allocated now."""
        v, n = C.c_int(), C.c_int()
        self._chk(self.lib.spcbpt_lvc_get_capacity(self.h, C.byref(v), C.byref(n)), "lvc_get_capacity")
        return v.value, n.value

    def lvc_export(self):
        dv, dc, cap = C.c_void_p(), C.c_void_p(), C.c_int()
        self._chk(self.lib.spcbpt_lvc_export(self.h, C.byref(dv), C.byref(dc), C.byref(cap)), "lvc_export")
        return dv.value, dc.value, cap.value

    def sampler_read(self, capacity=None):
        if capacity is None:
            capacity = self.lt.num_core * self.lt.core_padding if self.lt else 1 << 22
        sub = np.zeros(NUM_SUBSPACE, dtype=SUBSPACE_DTYPE)
        cmfs = np.zeros(capacity, dtype=np.float32)
        jump = np.zeros(capacity, dtype=np.int32)
        vc, pc = C.c_int(), C.c_int()
        self._chk(self.lib.spcbpt_sampler_read(self.h, sub.ctypes.data, cmfs.ctypes.data, jump.ctypes.data, capacity,
                                               C.byref(vc), C.byref(pc)), "sampler_read")
        return sub, cmfs[:vc.value].copy(), jump[:vc.value].copy(), vc.value, pc.value

    def accum_device_ptr(self):
        p = C.c_void_p()
        self._chk(self.lib.spcbpt_accum_device_ptr(self.h, C.byref(p)), "accum_device_ptr")
        return p.value

    def set_light_ahead(self, on: bool):
        """Light passes may be launched one frame ahead of their exchange / sampler build (see spcbpt_set_light_ahead)."""
        self._chk(self.lib.spcbpt_set_light_ahead(self.h, int(on)), "set_light_ahead")

    def lvc_import_wait(self):
        """Before overwriting one of two alternating device staging buffers of lvc_import_device (spcbpt_lvc_import_wait)."""
        self._chk(self.lib.spcbpt_lvc_import_wait(self.h), "lvc_import_wait")

    def sync_light(self):
        self._chk(self.lib.spcbpt_sync_light(self.h), "sync_light")

    def stream(self):
        p = C.c_void_p()
        self._chk(self.lib.spcbpt_stream(self.h, C.byref(p)), "stream")
        return p.value or 0

    # -- instrumentation ----------------------------------------------------
    def counters(self):
        c = Counters()
        self._chk(self.lib.spcbpt_get_counters(self.h, C.byref(c)), "get_counters")
        return c.as_dict()

    def phase_clocks(self):
        out = (C.c_uint64 * 19)()
        self._chk(self.lib.spcbpt_debug_phase_clocks(self.h, out), "debug_phase_clocks")
        return dict(zip(("regen", "closest", "shade", "shadow_pool", "connect", "node_slots", "node_lanes", "tri_slots", "tri_lanes", "sample_lane_clocks",
                         "wave_start_min", "wave_end_max", "wave_end_sum", "waves", "tail_slots", "tail_closest_lanes", "tail_shadow_lanes", "job_slots", "job_lanes"),
                        [int(v) for v in out]))

    def set_connection_sampler(self, mode: int):
        """0 = the subspace sampler (sampleFirstStage + sampleSecondStage), 1 = uniformSample ("plain BDPT", cuProg.h:283-289)"""
        self._chk(self.lib.spcbpt_set_connection_sampler(self.h, int(mode)), "set_connection_sampler")

    def unit(self, op: int, records: np.ndarray, out_words: int, aux: Optional[np.ndarray] = None) -> np.ndarray:
        """spcbpt_debug_unit: `records` is an (n, in_words) array of 32-bit words (any 4-byte dtype / structured dtype of that
        size); returns an (n, out_words) uint32 array."""
        a = np.ascontiguousarray(records)
        n = a.shape[0]
        in_words = a.dtype.itemsize // 4 if a.ndim == 1 else a.shape[1] * a.dtype.itemsize // 4
        out = np.zeros((n, out_words), np.uint32)
        ax = None if aux is None else np.ascontiguousarray(aux, dtype=np.float32)
        self._chk(self.lib.spcbpt_debug_unit(self.h, int(op), C.c_void_p(a.ctypes.data), in_words, C.c_void_p(out.ctypes.data), out_words, n,
                                             None if ax is None else C.c_void_p(ax.ctypes.data), 0 if ax is None else int(ax.size)), "debug_unit")
        return out

    def spill_arm(self):
        self._chk(self.lib.spcbpt_debug_spill_arm(self.h), "debug_spill_arm")

    def spill_count(self):
        """(words of the HBM traversal-stack spill areas written since spill_arm(), spill entries per thread)"""
        n, e = C.c_uint64(), C.c_int32()
        self._chk(self.lib.spcbpt_debug_spill_count(self.h, C.byref(n), C.byref(e)), "debug_spill_count")
        return int(n.value), int(e.value)

    def reset_counters(self):
        self._chk(self.lib.spcbpt_reset_counters(self.h), "reset_counters")

    def enable_counters(self, on: bool):
        self._chk(self.lib.spcbpt_enable_counters(self.h, int(on)), "enable_counters")

    def enable_kernel_timing(self, on: bool):
        self._chk(self.lib.spcbpt_enable_kernel_timing(self.h, int(on)), "enable_kernel_timing")

    def reset_kernel_time(self):
        self._chk(self.lib.spcbpt_reset_kernel_time(self.h), "reset_kernel_time")

    def kernel_time(self, name: str):
        ms, n = C.c_double(), C.c_int()
        self._chk(self.lib.spcbpt_kernel_time(self.h, name.encode(), C.byref(ms), C.byref(n)), "kernel_time")
        return ms.value, n.value

    def scene_info(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.lib.spcbpt_scene_info(self.h, C.byref(a), C.byref(b), C.byref(c)), "scene_info")
        return dict(n_triangles=a.value, n_bvh_nodes=b.value, bvh_depth=c.value)

    def trace_closest(self, rays: np.ndarray):
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = r.shape[0]
        t = np.zeros(n, dtype=np.float32)
        tri = np.zeros(n, dtype=np.int32)
        uv = np.zeros((n, 2), dtype=np.float32)
        self._chk(self.lib.spcbpt_trace_closest(self.h, r.ctypes.data, n, t.ctypes.data, tri.ctypes.data, uv.ctypes.data),
                  "trace_closest")
        return t, tri, uv

    def trace_any(self, rays: np.ndarray):
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = r.shape[0]
        vis = np.zeros(n, dtype=np.int32)
        self._chk(self.lib.spcbpt_trace_any(self.h, r.ctypes.data, n, vis.ctypes.data), "trace_any")
        return vis

    def trace_bench(self, rays: np.ndarray, mode: int, any_hit: bool, repeat: int = 5, stats: bool = True):
        """spcbpt_debug_trace_bench: the same rays through the lane-per-ray (mode 0) or quad-per-ray (mode 1) pool-fed kernels.
        Returns (outputs, avg_ms, stats dict or None); outputs = visibility, or (t, tri, uv)."""
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = r.shape[0]
        ms = C.c_double(0.0)
        st = (C.c_uint64 * 5)()
        if any_hit:
            vis = np.zeros(n, dtype=np.int32)
            self._chk(self.lib.spcbpt_debug_trace_bench(self.h, r.ctypes.data, n, int(mode), 1, int(repeat), None, None, None, vis.ctypes.data,
                                                        C.byref(ms), st if stats else None), "debug_trace_bench")
            out = vis
        else:
            t = np.zeros(n, dtype=np.float32); tri = np.zeros(n, dtype=np.int32); uv = np.zeros((n, 2), dtype=np.float32)
            self._chk(self.lib.spcbpt_debug_trace_bench(self.h, r.ctypes.data, n, int(mode), 0, int(repeat), t.ctypes.data, tri.ctypes.data, uv.ctypes.data, None,
                                                        C.byref(ms), st if stats else None), "debug_trace_bench")
            out = (t, tri, uv)
        sd = dict(zip(("node_visits", "leaf_visits", "tri_tests", "lane_slots", "lanes_busy"), [int(x) for x in st])) if stats else None
        return out, float(ms.value), sd

    # -- the reference's per-frame sequence (optixPathTracer.cpp:791-822) ------
    def render_frame(self, alg: str, subframe: int, launch_frame: Optional[int] = None, rows=None):
        if alg in ("SPCBPT_eye", "SPCBPT_no_rmis"):
            self.launch("light trace", subframe + 1 if launch_frame is None else launch_frame)
            self.build_sampler()
        self.launch(alg, subframe, rows)
