"""Renders three frames of three scenes (SPCBPT with a tuple trained on the spot, then pt) with whatever library SPCBPT_LIB names
(default: the one in the tree) and writes the accumulation buffers to gpurun_out/film_<tag>.npz -- for bit-for-bit comparisons ACROSS
builds (tools/film_cmp.py): a change of the traversal schedule must not change a film.  `films(pkg)` is also what
tests/test_gpu_film_golden.py hashes against tests/golden/film_hashes.json (--hashes prints that file's content).
usage: [SPCBPT_LIB=.ab/libA.so] python tools/film_dump.py <tag> [--hashes]"""
import hashlib, json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def films(pkg):
    out = {}
    for name, scene, W, H, lt in (("cornell", pkg.scenes.cornell_box(), 256, 256, (4000, 64, 1)),
                                  ("bedroom", pkg.scenes.bedroom(target_tris=60000, tex_size=64), 320, 180, (8000, 64, 1)),
                                  ("needles", pkg.scenes.needle_room(20000), 160, 120, (4000, 64, 1))):
        r = pkg.Renderer(scene, 0)
        cam = scene.camera
        r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
        r.resize(W, H)
        r.set_light_trace(*lt)
        r.set_pretrace(20000, 10)
        r.preprocess(target_paths=100000, target_q_paths=100000, train=True)
        for f in range(3):
            r.render_frame("SPCBPT_eye", f)
        r.sync()
        out[name] = r.read_accum().copy()
        for f in range(3):
            r.render_frame("pt", f)
        r.sync()
        out[name + "_pt"] = r.read_accum().copy()
    return out


def hashes(out):
    return {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in sorted(out.items())}


if __name__ == "__main__":
    import __graft_entry__ as g
    out = films(g.load_package())
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez(f"gpurun_out/film_{sys.argv[1]}.npz", **out)
    print("wrote", sorted(out))
    if "--hashes" in sys.argv:
        print(json.dumps(hashes(out), indent=1))
