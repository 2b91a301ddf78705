"""Generates tests/golden/*.npz.  Run in the build container (needs /root/reference for the
reference-sourced vectors): `python tests/golden/make_golden.py`.

ref_*.npz  : outputs of the reference's OWN sources (cuda/random.h, cuda/helpers.h, sutil/Camera.cpp) compiled by
             `make -C oracle ref` — real reference outputs.
survey_kat.npz : known answers the survey recorded from the reference's device code.
oracle_*.npz : regression vectors of the CPU restatement itself (NOT reference outputs; they pin the oracle against
             accidental change and let the GPU box check the HIP path without rebuilding anything).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
from oracle import binding as ob  # noqa: E402
from tests.parity_util import minimal_tuple  # noqa: E402


def ref_vectors():
    r = ob.ref_lib()
    assert r is not None, "build oracle/_ref first: make -C oracle ref"
    rng = np.random.default_rng(123)
    a = rng.integers(0, 2**32, size=256, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 2**32, size=256, dtype=np.uint64).astype(np.uint32)
    a[:8] = [0, 1, 7, 2**32 - 1, 1920 * 1080, 12345, 1, 2]
    b[:8] = [0, 0, 3, 2**32 - 1, 1, 64, 2, 1]
    tea = np.array([r.ref_tea4(C.c_uint(int(x)), C.c_uint(int(y))) for x, y in zip(a, b)], dtype=np.uint32)
    seq = np.zeros((256, 8), dtype=np.float32)
    seeds_after = np.zeros(256, dtype=np.uint32)
    for i, s0 in enumerate(tea):
        s = C.c_uint(int(s0))
        for k in range(8):
            seq[i, k] = r.ref_rnd(C.byref(s))
        seeds_after[i] = s.value
    np.savez(os.path.join(HERE, "ref_random.npz"), a=a, b=b, tea4=tea, rnd=seq, seed_after=seeds_after)
    rgb = np.concatenate([rng.uniform(0, 1.2, size=(500, 3)), rng.uniform(0, 0.01, size=(100, 3)),
                          np.array([[0, 0, 0], [1, 1, 1], [0.0031308, 0.5, 2.0], [0.2, 0.5, 0.001]])]).astype(np.float32)
    srgb = np.zeros_like(rgb)
    q = np.zeros((rgb.shape[0], 4), dtype=np.uint8)
    r.ref_toSRGB(C.c_void_p(rgb.ctypes.data), rgb.shape[0], C.c_void_p(srgb.ctypes.data))
    r.ref_make_color(C.c_void_p(rgb.ctypes.data), rgb.shape[0], C.c_void_p(q.ctypes.data))
    np.savez(os.path.join(HERE, "ref_color.npz"), rgb=rgb, srgb=srgb, rgba8=q)
    cams = []
    for k in range(32):
        eye = rng.uniform(-5, 5, 3).astype(np.float32)
        look = rng.uniform(-5, 5, 3).astype(np.float32)
        up = np.array([0, 1, 0], np.float32) if k % 2 == 0 else rng.normal(size=3).astype(np.float32)
        fov = np.float32(rng.uniform(20, 90))
        asp = np.float32(rng.uniform(0.5, 2.5))
        U, V, W = (np.zeros(3, np.float32) for _ in range(3))
        r.ref_uvw(C.c_void_p(eye.ctypes.data), C.c_void_p(look.ctypes.data), C.c_void_p(up.ctypes.data), C.c_float(fov),
                  C.c_float(asp), C.c_void_p(U.ctypes.data), C.c_void_p(V.ctypes.data), C.c_void_p(W.ctypes.data))
        cams.append(np.concatenate([eye, look, up, [fov, asp], U, V, W]))
    np.savez(os.path.join(HERE, "ref_camera.npz"), cams=np.array(cams, dtype=np.float32))



OBJ_CASES = {
    # authored here (data, not reference text): the OBJ features the scene hand-off can meet
    "tri_quad_fan": "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 1.5 0\nf 1 2 3\nf 1 2 3 4\nf 1 2 3 4 5\n",
    "uv_shared_and_split": "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvt 0.5 0.5\n"
                           "f 1/1 2/2 3/3\nf 1/1 3/3 4/4\nf 1/5 2/2 4/4\n",
    "negative_and_normals": "v 0 0 0\nv 2 0 0\nv 2 2 0\nvn 0 0 1\nvt 0.25 0.75\nf -3//1 -2//1 -1//1\nv 0 2 0\nf -4/1/1 -2/1/1 -1/1/1\n",
    "groups_and_objects": "o first\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\ng second\nv 0 0 1\nv 1 0 1\nv 0 1 1\nf 4 5 6\no third\nf 1 5 3\n",
    "mixed_uv_then_none": "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0.1 0.2\nvt 0.3 0.4\nvt 0.5 0.6\nf 1/1 2/2 3/3\nf 2 4 3\n",
    "comments_blank_tabs": "# a comment\n\nv\t0 0 0\nv 1   0 0\n  v 0 1 0\n#f 1 2 3\nf 1 2 3\n\n",
    "number_formats": "v 0.1 1e-3 123456.789\nv -0.000123 3.14159265358979 .5\nv 5. -7 +2.5E+2\nv 1 2\nv 1e 2e+ -x\nv 0.30000001 7.0E-08 16777217\nvt 0.333333333 0.6666667\nvt 1\nvt 2 3\nf 1/1 2/2 3/3\nf 4/1 1/2 2/3\nf 5/1 6/2 1/3\n",
    "zero_index_and_short_faces": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 2 3\nf 1 2\nf 1 2 3\n",
    "usemtl_splits_shapes": "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nusemtl a\nf 1 2 3\nusemtl b\nf 2 4 3\n",
}


def ref_loaders():
    """Row f1 pins: the reference's vendored tinyobj (old API) and stb_image, driven as scene_shift.cpp drives them."""
    import tempfile
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref.so"))
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for name, text in OBJ_CASES.items():
            p = os.path.join(d, name + ".obj")
            open(p, "w").write(text)
            nv, ni = C.c_int(), C.c_int()
            sv, si = (C.c_int * 16)(), (C.c_int * 16)()
            ns = lib.ref_obj_load(p.encode(), C.byref(nv), C.byref(ni), sv, si, 16, None, None, None)
            pos = np.zeros((nv.value, 3), np.float32); uv = np.zeros((nv.value, 2), np.float32); idx = np.zeros(ni.value, np.uint32)
            lib.ref_obj_load(p.encode(), C.byref(nv), C.byref(ni), sv, si, 16, pos.ctypes.data_as(C.c_void_p),
                             uv.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p))
            out[name + "__text"] = np.frombuffer(text.encode(), dtype=np.uint8)
            out[name + "__pos"] = pos; out[name + "__uv"] = uv; out[name + "__idx"] = idx
            out[name + "__shape_v"] = np.array(sv[:ns], np.int32); out[name + "__shape_i"] = np.array(si[:ns], np.int32)
        rng = np.random.default_rng(11)
        for k, (w, h) in enumerate([(1, 1), (5, 3), (16, 16)]):
            rgb = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
            p = os.path.join(d, f"t{k}.ppm")
            open(p, "wb").write(b"P6\n# made by make_golden\n%d %d\n255\n" % (w, h) + rgb.tobytes())
            ww, hh = C.c_int(), C.c_int()
            px = np.zeros((h, w, 4), np.uint8)
            rc = lib.ref_image_load(p.encode(), C.byref(ww), C.byref(hh), px.ctypes.data_as(C.c_void_p), px.nbytes)
            assert rc == 0 and (ww.value, hh.value) == (w, h)
            out[f"ppm{k}__file"] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
            out[f"ppm{k}__rgba"] = px
    np.savez_compressed(os.path.join(HERE, "ref_loaders.npz"), **out)


def gltf_cases():
    """Authored glTF files (data, not reference text): name -> {filename: bytes}, main file first."""
    import base64, json, struct
    rng = np.random.default_rng(5)
    cases = {}

    def quad_mesh(n=4):
        p = rng.normal(size=(n, 3)).astype(np.float32)
        t = rng.random(size=(n, 2)).astype(np.float32)
        return p, t

    def doc_base():
        return {"asset": {"version": "2.0"}}

    # 1. nested TRS + matrix child, u16 indices
    p, t = quad_mesh(5)
    idx = np.array([0, 1, 2, 2, 3, 4, 0, 2, 4], np.uint16)
    blob = p.tobytes() + t.tobytes() + idx.tobytes() + b"\0\0"
    d = doc_base()
    d.update(buffers=[{"uri": "a.bin", "byteLength": len(blob)}],
             bufferViews=[{"buffer": 0, "byteOffset": 0, "byteLength": 60}, {"buffer": 0, "byteOffset": 60, "byteLength": 40},
                          {"buffer": 0, "byteOffset": 100, "byteLength": 18}],
             accessors=[{"bufferView": 0, "componentType": 5126, "count": 5, "type": "VEC3"},
                        {"bufferView": 1, "componentType": 5126, "count": 5, "type": "VEC2"},
                        {"bufferView": 2, "componentType": 5123, "count": 9, "type": "SCALAR"}],
             materials=[{"pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.4, 0.2, 1.0], "metallicFactor": 0.25, "roughnessFactor": 0.6}}],
             meshes=[{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1}, "indices": 2, "material": 0}]}],
             nodes=[{"translation": [1.5, -2.25, 0.125], "rotation": [0.18257419, 0.36514837, 0.54772256, 0.73029674], "scale": [2.0, 0.5, 1.25], "children": [1]},
                    {"mesh": 0, "matrix": [0.0, 1.0, 0.0, 0.0, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 3.0, 4.0, 5.0, 1.0]}],
             scenes=[{"nodes": [0]}], scene=0)
    cases["trs_nested"] = {"m.gltf": json.dumps(d).encode(), "a.bin": blob}

    # 2. interleaved view with byteStride + accessor byteOffset, u32 indices, node with BOTH matrix and TRS
    p, t = quad_mesh(6)
    inter = np.zeros((6, 5), np.float32); inter[:, :3] = p; inter[:, 3:] = t
    idx = np.array([5, 4, 3, 2, 1, 0], np.uint32)
    blob = b"\x11" * 8 + inter.tobytes() + idx.tobytes()
    d = doc_base()
    d.update(buffers=[{"uri": "b.bin", "byteLength": len(blob)}],
             bufferViews=[{"buffer": 0, "byteOffset": 8, "byteLength": 120, "byteStride": 20}, {"buffer": 0, "byteOffset": 128, "byteLength": 24}],
             accessors=[{"bufferView": 0, "byteOffset": 0, "componentType": 5126, "count": 6, "type": "VEC3"},
                        {"bufferView": 0, "byteOffset": 12, "componentType": 5126, "count": 6, "type": "VEC2"},
                        {"bufferView": 1, "componentType": 5125, "count": 6, "type": "SCALAR"}],
             meshes=[{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1}, "indices": 2}]}],
             nodes=[{"mesh": 0, "matrix": [2.0, 0, 0, 0, 0, 2.0, 0, 0, 0, 0, 2.0, 0, 0.5, 0.25, -1.0, 1.0], "translation": [0.1, 0.2, 0.3],
                     "rotation": [0.0, 0.70710678, 0.0, 0.70710678], "scale": [1.0, 3.0, 1.0]}])
    cases["interleaved_matrix_and_trs"] = {"m.gltf": json.dumps(d).encode(), "b.bin": blob}

    # 3. two primitives / materials (one without UVs, one material without factors), a LINES primitive that is skipped,
    #    the same mesh instanced twice, a node outside scenes[0], base64 buffer
    p1, t1 = quad_mesh(3); p2, _ = quad_mesh(4)
    i1 = np.array([0, 1, 2], np.uint16); i2 = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    blob = p1.tobytes() + t1.tobytes() + p2.tobytes() + i1.tobytes() + b"\0\0" + i2.tobytes()
    off = [0, 36, 60, 108, 116]
    d = doc_base()
    d.update(buffers=[{"uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode(), "byteLength": len(blob)}],
             bufferViews=[{"buffer": 0, "byteOffset": off[0], "byteLength": 36}, {"buffer": 0, "byteOffset": off[1], "byteLength": 24},
                          {"buffer": 0, "byteOffset": off[2], "byteLength": 48}, {"buffer": 0, "byteOffset": off[3], "byteLength": 6},
                          {"buffer": 0, "byteOffset": off[4], "byteLength": 12}],
             accessors=[{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}, {"bufferView": 1, "componentType": 5126, "count": 3, "type": "VEC2"},
                        {"bufferView": 2, "componentType": 5126, "count": 4, "type": "VEC3"}, {"bufferView": 3, "componentType": 5123, "count": 3, "type": "SCALAR"},
                        {"bufferView": 4, "componentType": 5123, "count": 6, "type": "SCALAR"}],
             materials=[{"pbrMetallicRoughness": {"baseColorFactor": [0.1, 0.2, 0.3, 1.0]}}, {"name": "plain"},
                        {"pbrMetallicRoughness": {"metallicFactor": 0.0, "roughnessFactor": 0.05}}],
             meshes=[{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1}, "indices": 3, "material": 2},
                                     {"attributes": {"POSITION": 2}, "indices": 4, "material": 0},
                                     {"attributes": {"POSITION": 2}, "indices": 4, "material": 1, "mode": 1}]}],
             nodes=[{"mesh": 0}, {"mesh": 0, "translation": [10.0, 0.0, 0.0]}, {"mesh": 0, "scale": [-1.0, 1.0, 1.0], "rotation": [0.5, 0.5, 0.5, 0.5]}],
             scenes=[{"nodes": [0, 1]}], scene=0)
    cases["multi_prim_instances_base64"] = {"m.gltf": json.dumps(d).encode()}

    # 4. camera under a transformed parent + mesh; as .glb
    p, t = quad_mesh(3)
    idx = np.array([0, 1, 2], np.uint32)
    blob = p.tobytes() + t.tobytes() + idx.tobytes()
    d = doc_base()
    d.update(buffers=[{"byteLength": len(blob)}],
             bufferViews=[{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 24}, {"buffer": 0, "byteOffset": 60, "byteLength": 12}],
             accessors=[{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}, {"bufferView": 1, "componentType": 5126, "count": 3, "type": "VEC2"},
                        {"bufferView": 2, "componentType": 5125, "count": 3, "type": "SCALAR"}],
             cameras=[{"type": "orthographic", "orthographic": {"xmag": 1, "ymag": 1, "zfar": 10, "znear": 0.1}},
                      {"type": "perspective", "perspective": {"yfov": 0.6108652381980153, "aspectRatio": 1.7777, "znear": 0.01}}],
             meshes=[{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1}, "indices": 2}]}],
             nodes=[{"children": [1, 2, 3], "translation": [0.0, 1.0, 0.0], "rotation": [0.0, 0.38268343, 0.0, 0.92387953]},
                    {"camera": 0}, {"camera": 1, "translation": [1.0, 2.0, 3.0], "rotation": [0.25881905, 0.0, 0.0, 0.96592583]}, {"mesh": 0}])
    js = json.dumps(d, separators=(",", ":")).encode(); js += b" " * (-len(js) % 4)
    glb = struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(blob)) + struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(blob), 0x004E4942) + blob
    cases["camera_glb"] = {"m.glb": glb}
    return cases


def ref_gltf():
    """Row f2 pins: the reference's vendored tinygltf + sutil Matrix4x4/Quaternion walking the model as sutil::loadScene does."""
    import tempfile
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_gltf.so"))
    out = {}
    for name, files in gltf_cases().items():
        with tempfile.TemporaryDirectory() as d:
            for fn, data in files.items():
                open(os.path.join(d, fn), "wb").write(data)
            main = os.path.join(d, next(iter(files)))
            nv, nt, nm = C.c_int(), C.c_int(), C.c_int()
            rc = lib.ref_gltf_load(main.encode(), C.byref(nv), C.byref(nt), C.byref(nm), None, None, None, None, None, None)
            assert rc == 0, name
            pos = np.zeros((nv.value, 3), np.float32); uv = np.zeros((nv.value, 2), np.float32)
            idx = np.zeros((nt.value, 3), np.uint32); tm = np.zeros(nt.value, np.int32)
            mats = np.zeros((max(nm.value, 1), 6), np.float32); cam = np.zeros(9, np.float32)
            vp = lambda a: a.ctypes.data_as(C.c_void_p)
            lib.ref_gltf_load(main.encode(), C.byref(nv), C.byref(nt), C.byref(nm), vp(pos), vp(uv), vp(idx), vp(tm), vp(mats), vp(cam))
            for fn, data in files.items():
                out[f"{name}__file__{fn}"] = np.frombuffer(data, dtype=np.uint8)
            out[name + "__pos"] = pos; out[name + "__uv"] = uv; out[name + "__idx"] = idx; out[name + "__tri_mat"] = tm
            out[name + "__materials"] = mats[: nm.value]; out[name + "__camera"] = cam
    np.savez_compressed(os.path.join(HERE, "ref_gltf.npz"), **out)


def survey_kat():
    np.savez(os.path.join(HERE, "survey_kat.npz"),
             tea4_7_3=np.uint32(2175312897),
             rnd3=np.array([0.440146208, 0.79995507, 0.646324039], np.float32),
             bsdf_mat=np.array([0.8, 0.5, 0.3, 0.0, 0.5], np.float32),  # base rgb, metallic, roughness
             bsdf_N=np.array([0, 0, 1], np.float32), bsdf_V=np.array([0.3, 0.2, 0.9], np.float32),
             bsdf_seed_args=np.array([1, 2], np.uint32),
             bsdf_L=np.array([0.9040936, -0.304971, 0.2993451], np.float32),
             bsdf_f=np.array([0.2700801, 0.1705985, 0.1042775], np.float32),
             bsdf_pdf=np.float32(0.06876558))


def oracle_regression():
    scene = pkg.scenes.cornell_box()
    W = H = 32
    o = ob.Oracle(scene, nthreads=1)
    cam = scene.camera
    o.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
    o.resize(W, H)
    o.set_light_trace(500, 64, 1)
    for f in range(4):
        o.launch("pt", f)
    pt = o.read_accum().copy()
    tup = minimal_tuple(o, 2)
    o.set_subspace(*tup)
    o.clear_accum()
    for f in range(4):
        o.render_frame("SPCBPT_eye", f)
    sp = o.read_accum().copy()
    o.launch("light trace", 1)
    lvc = o.lvc_read()
    o.build_sampler()
    sub, cmfs, jump, vc, pc = o.sampler_read()
    np.savez_compressed(os.path.join(HERE, "oracle_cornell32.npz"), pt=pt, spcbpt=sp, lvc=lvc, sub=sub, cmfs=cmfs, jump=jump,
                        vc=vc, pc=pc, q=tup[2], gamma_row=tup[3][0])


def ref_viewer_vectors():
    """ref_trackball.npz: camera after every event of the scripts in tests/viewer_scripts.py, from the reference's own
    sutil/Trackball.cpp + sutil/Camera.cpp (oracle/ref_viewer.cpp -> oracle/_ref/libref_viewer.so)."""
    from tests.viewer_scripts import scripts
    data = {}
    for k, s in enumerate(scripts()):
        cam = ob.ref_viewer_replay(s["eye"], s["lookat"], s["up"], float(s["fov"]), float(s["aspect"]), s["events"])
        assert cam is not None, "build oracle/_ref first: make -C oracle ref"
        data[f"cam{k}"] = cam
        data[f"events{k}"] = s["events"]
    np.savez_compressed(os.path.join(HERE, "ref_trackball.npz"), **data)
    print("ref_trackball.npz:", {k: v.shape for k, v in data.items()})


def ref_layout_vectors():
    """ref_layout.json: sizeof / offsetof / constructor defaults of the reference's own POD headers that compile without the
    OptiX SDK (oracle/ref_layout.cpp -> oracle/_ref/libref_layout.so): pins SURVEY.md a21 (Light 80 B, Pbr 144 B) and q17."""
    import json
    d = ob.ref_layout()
    assert d is not None, "build oracle/_ref first: make -C oracle ref"
    with open(os.path.join(HERE, "ref_layout.json"), "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)
    print("ref_layout.json:", len(d), "entries")


if __name__ == "__main__":
    if "--viewer" in sys.argv:      # only the row-f3 vectors
        ref_viewer_vectors()
    elif "--layout" in sys.argv:    # only the a21 / q17 layout pins
        ref_layout_vectors()
    else:
        ref_layout_vectors()
        ref_vectors()
        ref_loaders()
        ref_gltf()
        ref_viewer_vectors()
        survey_kat()
        oracle_regression()
    print("golden vectors written to", HERE)
