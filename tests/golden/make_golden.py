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


if __name__ == "__main__":
    ref_vectors()
    survey_kat()
    oracle_regression()
    print("golden vectors written to", HERE)
