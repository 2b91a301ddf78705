"""A/B of the two traversal schedules of csrc/quad_trace.hip on ray sets of the bench scene:
  primary    camera rays, 1920 x 1080, in the render kernels' 8x8-tile order (coherent)
  bounce     cosine-hemisphere rays from the primary hit points (what a path's second segment looks like: incoherent)
  shadow     hit point -> a random OTHER hit point of the set, tmax = distance - eps (visibilityTest-like: about half occluded)
Prints per set and mode: ms per launch, Mrays/s, node / leaf visits and triangle tests per ray, lane utilisation; checks that
both modes return the same hits.  Usage (GPU box): python tools/quad_trace_report.py [--tris 1000000] [--out profiles/r04_quad_trace.json]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def ray_sets(pkg, scene, r, width, height):
    cam = scene.camera
    U, V, W = pkg.camera_frame(np.array(cam["eye"], np.float32), np.array(cam["lookat"], np.float32), np.array(cam["up"], np.float32),
                               np.float32(cam["fov"]), np.float32(width / height))
    ys, xs = np.mgrid[0:height, 0:width]
    order = np.lexsort((xs.ravel() % 8, ys.ravel() % 8, xs.ravel() // 8, ys.ravel() // 8))   # 8x8 pixel tiles, tile after tile
    px = (xs.ravel()[order] + 0.5) / width * 2 - 1
    py = (ys.ravel()[order] + 0.5) / height * 2 - 1
    d = px[:, None] * U[None] + py[:, None] * V[None] + W[None]
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    n = len(d)
    eye = np.array(cam["eye"], np.float32)
    col = lambda v, k: np.full((k, 1), v, np.float32)
    rays = np.concatenate([np.tile(eye, (n, 1)), col(1e-3, n), d, col(1e16, n)], 1).astype(np.float32)
    sets = {"primary": (rays, False)}
    (t, tri, uv), _, _ = r.trace_bench(rays, 0, False, repeat=1, stats=False)
    sel = (tri >= 0) & (tri < len(scene.indices))           # scene triangles (the quad lights' follow them)
    P = (eye[None] + t[:, None] * d)[sel]
    T = scene.vertices[scene.indices[tri[sel]]]
    N = np.cross(T[:, 1] - T[:, 0], T[:, 2] - T[:, 0])
    N /= np.linalg.norm(N, axis=1, keepdims=True) + 1e-30
    N = np.where((N * d[sel]).sum(1, keepdims=True) > 0, -N, N).astype(np.float32)
    rng = np.random.default_rng(7)
    m = len(P)
    u1, u2 = rng.random(m), rng.random(m)
    rr, phi = np.sqrt(u1), 2 * np.pi * u2
    lx, ly, lz = rr * np.cos(phi), rr * np.sin(phi), np.sqrt(np.maximum(0, 1 - u1))
    helper = np.where(np.abs(N[:, :1]) > 0.9, np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))
    Tn = np.cross(helper, N); Tn /= np.linalg.norm(Tn, axis=1, keepdims=True)
    Bn = np.cross(N, Tn)
    bd = lx[:, None] * Tn + ly[:, None] * Bn + lz[:, None] * N
    bd = (bd / np.linalg.norm(bd, axis=1, keepdims=True)).astype(np.float32)
    sets["bounce"] = (np.concatenate([P, col(1e-3, m), bd, col(1e16, m)], 1).astype(np.float32), False)
    other = P[rng.permutation(m)]
    sv = other - P
    dist = np.linalg.norm(sv, axis=1)
    keep = dist > 1e-2
    sd = (sv[keep] / dist[keep, None]).astype(np.float32)
    k = int(keep.sum())
    sets["shadow"] = (np.concatenate([P[keep], col(1e-3, k), sd, (dist[keep, None] - 1e-3).astype(np.float32)], 1).astype(np.float32), True)
    return sets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--repeat", type=int, default=5)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    pkg = g.load_package()
    scene = pkg.scenes.bedroom(target_tris=a.tris)
    r = pkg.Renderer(scene, 0)
    report = {"scene": dict(r.scene_info(), name="bedroom"), "sets": {}}
    for name, (rs, any_hit) in ray_sets(pkg, scene, r, a.width, a.height).items():
        res, outs, nr = {}, [], len(rs)
        for mode in (0, 1, 2, 3, 4):
            out, ms, st = r.trace_bench(rs, mode, any_hit, repeat=a.repeat)
            outs.append(out)
            res[("lane", "quad", "quad_x2", "quad_x4", "quad_lean")[mode]] = dict(
                ms=round(ms, 4), mrays_per_s=round(nr / ms / 1e3, 1), node_visits_per_ray=round(st["node_visits"] / nr, 2),
                leaf_visits_per_ray=round(st["leaf_visits"] / nr, 2), tri_tests_per_ray=round(st["tri_tests"] / nr, 2),
                lane_utilisation=round(st["lanes_busy"] / max(st["lane_slots"], 1), 4))
        if any_hit:
            res["agreement"] = dict(same_visibility=[float((outs[0] == o).mean()) for o in outs[1:]], visible_share=float(outs[0].mean()))
        else:
            res["agreement"] = dict(same_triangle=[float((outs[0][1] == o[1]).mean()) for o in outs[1:]],
                                    max_dt=[float(np.abs(outs[0][0] - o[0])[outs[0][1] == o[1]].max()) for o in outs[1:]],
                                    hit_share=float((outs[0][1] >= 0).mean()))
        res["rays"] = nr
        res["speedup_over_lane"] = {k: round(res["lane"]["ms"] / res[k]["ms"], 3) for k in ("quad", "quad_x2", "quad_x4", "quad_lean")}
        report["sets"][name] = res
        print(name, json.dumps(res))
    if a.out:
        json.dump(report, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
