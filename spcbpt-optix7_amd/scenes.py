"""Synthetic scenes for the BASELINE.json configs (the reference ships only the
incomplete `house` scene; SURVEY.md 8(d) C1-C5 name build-authored inputs).

All scenes are closed or near-closed rooms with the quad emitters flush against
a wall/ceiling so the single-sided-emitter quirk does not bias the
PT-vs-SPCBPT comparison.  Geometry is deterministic (seeded numpy RNG).
"""
from __future__ import annotations

import numpy as np

from .api import Scene


class _Builder:
    def __init__(self):
        self.v, self.uv, self.i, self.m = [], [], [], []
        self.nv = 0

    def add(self, verts, uvs, tris, mat):
        verts = np.asarray(verts, dtype=np.float32).reshape(-1, 3)
        uvs = np.asarray(uvs, dtype=np.float32).reshape(-1, 2)
        tris = np.asarray(tris, dtype=np.uint32).reshape(-1, 3)
        self.v.append(verts)
        self.uv.append(uvs)
        self.i.append(tris + np.uint32(self.nv))
        self.m.append(np.full(tris.shape[0], mat, dtype=np.int32))
        self.nv += verts.shape[0]

    def grid(self, p0, du, dv, nu, nv, mat, height=None, uv_scale=1.0):
        """(nu x nv) quad grid spanning p0 + s*du + t*dv; optional height field displaced along normal."""
        p0, du, dv = (np.asarray(a, dtype=np.float64) for a in (p0, du, dv))
        s = np.linspace(0.0, 1.0, nu + 1)
        t = np.linspace(0.0, 1.0, nv + 1)
        S, T = np.meshgrid(s, t, indexing="ij")
        P = p0[None, None, :] + S[..., None] * du[None, None, :] + T[..., None] * dv[None, None, :]
        if height is not None:
            n = np.cross(du, dv)
            n = n / np.linalg.norm(n)
            P = P + height(S, T)[..., None] * n[None, None, :]
        idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
        a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
        tris = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)], 0)
        uv = np.stack([S * uv_scale, T * uv_scale], -1)
        self.add(P.reshape(-1, 3), uv.reshape(-1, 2), tris, mat)

    def box(self, lo, hi, mat, sub=1, rot_y=0.0, skip_bottom=False):
        lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
        c = 0.5 * (lo + hi)
        e = hi - lo
        cs, sn = np.cos(rot_y), np.sin(rot_y)
        R = np.array([[cs, 0, sn], [0, 1, 0], [-sn, 0, cs]])
        faces = [
            ((lo[0], lo[1], hi[2]), (e[0], 0, 0), (0, e[1], 0)),    # +z
            ((hi[0], lo[1], lo[2]), (-e[0], 0, 0), (0, e[1], 0)),   # -z
            ((hi[0], lo[1], hi[2]), (0, 0, -e[2]), (0, e[1], 0)),   # +x
            ((lo[0], lo[1], lo[2]), (0, 0, e[2]), (0, e[1], 0)),    # -x
            ((lo[0], hi[1], hi[2]), (e[0], 0, 0), (0, 0, -e[2])),   # +y
        ]
        if not skip_bottom:
            faces.append(((lo[0], lo[1], lo[2]), (e[0], 0, 0), (0, 0, e[2])))  # -y
        for p0, du, dv in faces:
            p0 = R @ (np.asarray(p0) - c) + c
            self.grid(p0, R @ np.asarray(du, dtype=np.float64), R @ np.asarray(dv, dtype=np.float64), sub, sub, mat)

    def sphere(self, center, radius, nu, nv, mat, squash=(1, 1, 1)):
        u = np.linspace(0, 2 * np.pi, nu + 1)
        v = np.linspace(0, np.pi, nv + 1)
        U, V = np.meshgrid(u, v, indexing="ij")
        P = np.stack([np.cos(U) * np.sin(V) * squash[0], np.cos(V) * squash[1], np.sin(U) * np.sin(V) * squash[2]], -1)
        P = np.asarray(center)[None, None, :] + radius * P
        idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
        a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
        tris = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)], 0)
        # drop degenerate pole triangles
        Pf = P.reshape(-1, 3)
        e1 = Pf[tris[:, 1]] - Pf[tris[:, 0]]
        e2 = Pf[tris[:, 2]] - Pf[tris[:, 0]]
        keep = np.linalg.norm(np.cross(e1, e2), axis=1) > 1e-12 * radius * radius
        uv = np.stack([U / (2 * np.pi), V / np.pi], -1)
        self.add(Pf, uv.reshape(-1, 2), tris[keep], mat)

    def finish(self, materials, lights, textures=(), camera=None, name="scene"):
        return Scene(vertices=np.concatenate(self.v), indices=np.concatenate(self.i),
                     tri_material=np.concatenate(self.m), materials=list(materials), lights=list(lights),
                     texcoords=np.concatenate(self.uv), textures=list(textures), camera=camera or {}, name=name)


def cornell_box(div_level: int = 10) -> Scene:
    """C1/C2: Cornell box, 5 walls + 2 boxes (34 triangles), diffuse `roughness .5 metallic 0`,
    one Quad ceiling light with divLevel 10, fov 35."""
    b = _Builder()
    W, R, G = 0, 1, 2
    b.grid((-1, 0, 1), (2, 0, 0), (0, 0, -2), 1, 1, W)      # floor (normal +y)
    b.grid((-1, 2, -1), (2, 0, 0), (0, 0, 2), 1, 1, W)      # ceiling (normal -y)
    b.grid((-1, 0, -1), (2, 0, 0), (0, 2, 0), 1, 1, W)      # back wall
    b.grid((-1, 0, 1), (0, 0, -2), (0, 2, 0), 1, 1, R)      # left wall (red)
    b.grid((1, 0, -1), (0, 0, 2), (0, 2, 0), 1, 1, G)       # right wall (green)
    b.box((0.1, 0.0, -0.1), (0.7, 0.6, 0.5), W, rot_y=-0.3, skip_bottom=True)   # short box
    b.box((-0.7, 0.0, -0.7), (-0.1, 1.2, -0.1), W, rot_y=0.3, skip_bottom=True)  # tall box
    mats = [dict(color=(0.73, 0.73, 0.73), roughness=0.5, metallic=0.0),
            dict(color=(0.65, 0.05, 0.05), roughness=0.5, metallic=0.0),
            dict(color=(0.12, 0.45, 0.15), roughness=0.5, metallic=0.0)]
    # normal = normalize(cross(u, v)) = -y : emits downward; 2 mm below the ceiling
    lights = [dict(position=(-0.25, 1.998, -0.25), u=(0.5, 0, 0), v=(0, 0, 0.5), emission=(17, 12, 4), div_level=div_level)]
    cam = dict(eye=(0.0, 1.0, 5.4), lookat=(0.0, 1.0, 0.0), up=(0, 1, 0), fov=35.0)
    return b.finish(mats, lights, camera=cam, name="cornell")


def _texture(kind: int, size: int, rng: np.random.Generator) -> np.ndarray:
    y, x = np.mgrid[0:size, 0:size].astype(np.float32) / size
    if kind == 0:    # checker
        c = ((np.floor(x * 16) + np.floor(y * 16)) % 2)
        rgb = np.stack([0.25 + 0.6 * c, 0.22 + 0.55 * c, 0.2 + 0.5 * c], -1)
    elif kind == 1:  # wood-ish rings
        r = np.sqrt((x - 0.3) ** 2 + (y - 0.6) ** 2)
        c = 0.5 + 0.5 * np.sin(r * 120 + 3 * np.sin(x * 20))
        rgb = np.stack([0.45 + 0.3 * c, 0.28 + 0.2 * c, 0.12 + 0.1 * c], -1)
    elif kind == 2:  # fabric stripes
        c = 0.5 + 0.5 * np.sin(x * 200) * np.sin(y * 200)
        rgb = np.stack([0.2 + 0.2 * c, 0.3 + 0.3 * c, 0.6 + 0.3 * c], -1)
    else:            # value noise
        n = rng.random((size // 8, size // 8, 3)).astype(np.float32)
        rgb = np.kron(n, np.ones((8, 8, 1), dtype=np.float32)) * 0.6 + 0.3
    rgba = np.concatenate([np.clip(rgb, 0, 1), np.ones((size, size, 1), np.float32)], -1)
    return (rgba * 255.0 + 0.5).astype(np.uint8)


def bedroom(target_tris: int = 1_000_000, seed: int = 0x5EED, tex_size: int = 1024) -> Scene:
    """C3/C4: procedural 'bedroom-class' room: tessellated shell, bed with a height-field blanket,
    furniture boxes and spheres, 12 Disney materials (roughness .05-.8, metallic 0-1), 2 quad lights
    partly occluded, four RGBA8 textures.  target_tris scales the tessellation."""
    rng = np.random.default_rng(seed)
    s = max(0.02, (target_tris / 1_000_000.0)) ** 0.5  # linear tessellation scale
    n = lambda k: max(1, int(round(k * s)))
    b = _Builder()
    mats = [
        dict(color=(0.8, 0.8, 0.8), roughness=0.6, metallic=0.0, albedo_tex=1),   # 0 floor (checker)
        dict(color=(0.75, 0.72, 0.68), roughness=0.8, metallic=0.0),              # 1 walls
        dict(color=(0.8, 0.8, 0.8), roughness=0.7, metallic=0.0),                 # 2 ceiling
        dict(color=(0.6, 0.4, 0.2), roughness=0.4, metallic=0.0, albedo_tex=2),   # 3 wood
        dict(color=(0.3, 0.4, 0.8), roughness=0.8, metallic=0.0, albedo_tex=3),   # 4 fabric
        dict(color=(0.9, 0.9, 0.9), roughness=0.05, metallic=1.0),                # 5 mirror-ish metal
        dict(color=(0.95, 0.64, 0.54), roughness=0.2, metallic=1.0),              # 6 copper
        dict(color=(0.8, 0.1, 0.1), roughness=0.3, metallic=0.0),                 # 7 red plastic
        dict(color=(0.1, 0.6, 0.2), roughness=0.5, metallic=0.0),                 # 8 green
        dict(color=(0.9, 0.85, 0.3), roughness=0.15, metallic=0.5),               # 9 gold-ish
        dict(color=(0.5, 0.5, 0.5), roughness=0.6, metallic=0.0, albedo_tex=4),   # 10 noise
        dict(color=(0.2, 0.2, 0.25), roughness=0.1, metallic=0.0),                # 11 dark glossy
    ]
    X, Y, Z = 6.0, 3.0, 5.0  # room [0,X]x[0,Y]x[0,Z]
    bump = lambda a, f: (lambda S, T: a * np.sin(S * f) * np.cos(T * f * 0.7))
    b.grid((0, 0, Z), (X, 0, 0), (0, 0, -Z), n(280), n(240), 0, height=bump(0.002, 90.0), uv_scale=4.0)   # floor
    b.grid((0, Y, 0), (X, 0, 0), (0, 0, Z), n(120), n(100), 2)                                           # ceiling
    b.grid((0, 0, 0), (X, 0, 0), (0, Y, 0), n(140), n(80), 1, height=bump(0.003, 40.0))                  # back  z=0
    b.grid((X, 0, Z), (-X, 0, 0), (0, Y, 0), n(140), n(80), 1)                                           # front z=Z
    b.grid((0, 0, Z), (0, 0, -Z), (0, Y, 0), n(120), n(80), 1)                                           # left  x=0
    b.grid((X, 0, 0), (0, 0, Z), (0, Y, 0), n(120), n(80), 1)                                            # right x=X
    # bed: frame + blanket height field
    b.box((0.4, 0.0, 0.3), (2.6, 0.45, 3.4), 3, sub=n(24))
    b.grid((0.4, 0.47, 3.4), (2.2, 0, 0), (0, 0, -3.1), n(260), n(300), 4,
           height=lambda S, T: 0.04 * np.sin(S * 23) * np.sin(T * 17) + 0.03 * np.sin(S * 61 + T * 47), uv_scale=3.0)
    b.box((0.4, 0.45, 0.3), (2.6, 1.2, 0.45), 3, sub=n(16))   # headboard
    # wardrobe, desk, shelves
    b.box((4.6, 0.0, 0.1), (5.9, 2.4, 0.9), 3, sub=n(28))
    b.box((3.2, 0.7, 3.9), (5.6, 0.76, 4.8), 3, sub=n(28))
    for lx in (3.3, 5.4):
        for lz in (4.0, 4.7):
            b.box((lx, 0.0, lz), (lx + 0.08, 0.7, lz + 0.08), 11, sub=n(6))
    # scattered objects
    n_sph = 22
    for k in range(n_sph):
        r = float(rng.uniform(0.08, 0.28))
        c = (float(rng.uniform(2.9, 5.6)), r + (0.76 if k % 3 == 0 else 0.0), float(rng.uniform(1.2, 4.7) if k % 3 else rng.uniform(4.0, 4.7)))
        if k % 3 == 0:
            c = (float(rng.uniform(3.4, 5.4)), 0.76 + r, float(rng.uniform(4.05, 4.65)))
        b.sphere(c, r, n(150), n(75), [5, 6, 7, 8, 9, 11][k % 6], squash=(1, float(rng.uniform(0.6, 1.0)), 1))
    for k in range(14):
        w = rng.uniform(0.15, 0.4, size=3)
        p = np.array([rng.uniform(2.9, 5.5), 0.0, rng.uniform(1.0, 3.6)])
        b.box(p, p + w, [7, 8, 9, 10, 3][k % 5], sub=n(20), rot_y=float(rng.uniform(0, 1.5)))
    # occluders in front of the lights (partly occluded emitters)
    b.box((2.2, 2.55, 1.8), (3.8, 2.6, 2.0), 11, sub=n(10))
    b.box((0.3, 1.3, 4.0), (0.35, 2.3, 4.6), 10, sub=n(10))
    lights = [
        # ceiling panel, emits downward (cross(u,v) = -y), 3 mm below the ceiling
        dict(position=(2.5, Y - 0.003, 2.0), u=(1.0, 0, 0), v=(0, 0, 1.0), emission=(22, 20, 17), div_level=10),
        # window-like panel on the left wall x=0, emits +x : u=(0,0,1), v=(0,1,0) -> cross = (-1,0,0)?  use u=(0,1,0), v=(0,0,1) -> (+1,0,0)
        dict(position=(0.003, 1.2, 3.2), u=(0, 1.2, 0), v=(0, 0, 1.4), emission=(9, 11, 14), div_level=10),
    ]
    texs = [_texture(k, tex_size, rng) for k in range(4)]
    cam = dict(eye=(5.6, 1.7, 4.6), lookat=(1.8, 0.8, 1.4), up=(0, 1, 0), fov=55.0)
    return b.finish(mats, lights, textures=texs, camera=cam, name="bedroom")


def hallway(target_tris: int = 200_000, seed: int = 7) -> Scene:
    """C5: hallway/door SDS scene — the light sits in an adjacent room behind a door ajar and reaches the
    camera's room only through the gap and off a glossy floor."""
    rng = np.random.default_rng(seed)
    s = max(0.02, target_tris / 200_000.0) ** 0.5
    n = lambda k: max(1, int(round(k * s)))
    b = _Builder()
    mats = [
        dict(color=(0.6, 0.6, 0.62), roughness=0.08, metallic=0.0),   # 0 glossy floor
        dict(color=(0.7, 0.7, 0.7), roughness=0.8, metallic=0.0),     # 1 walls
        dict(color=(0.5, 0.3, 0.15), roughness=0.5, metallic=0.0),    # 2 door
        dict(color=(0.9, 0.9, 0.9), roughness=0.1, metallic=1.0),     # 3 metal
    ]
    X, Y, Z = 8.0, 3.0, 3.0   # corridor x in [0,8]; lit room x in [8,11]
    b.grid((0, 0, Z), (11, 0, 0), (0, 0, -Z), n(220), n(60), 0)
    b.grid((0, Y, 0), (11, 0, 0), (0, 0, Z), n(110), n(30), 1)
    b.grid((0, 0, 0), (11, 0, 0), (0, Y, 0), n(110), n(30), 1)
    b.grid((11, 0, Z), (-11, 0, 0), (0, Y, 0), n(110), n(30), 1)
    b.grid((0, 0, Z), (0, 0, -Z), (0, Y, 0), n(30), n(30), 1)
    b.grid((11, 0, 0), (0, 0, Z), (0, Y, 0), n(30), n(30), 1)
    # partition wall at x=8 with a door opening z in [1.0, 2.0], y < 2.2
    b.box((7.95, 0, 0), (8.05, Y, 1.0), 1, sub=n(12))
    b.box((7.95, 0, 2.0), (8.05, Y, Z), 1, sub=n(12))
    b.box((7.95, 2.2, 1.0), (8.05, Y, 2.0), 1, sub=n(8))
    # door ajar: hinged at z=1.0, rotated ~20 degrees into the lit room
    b.box((8.05, 0.0, 0.98), (8.1, 2.2, 1.96), 2, sub=n(16), rot_y=0.35)
    for k in range(6):
        b.sphere((float(rng.uniform(1, 7)), 0.25, float(rng.uniform(0.5, 2.5))), 0.25, n(60), n(30), 3)
    lights = [dict(position=(9.0, Y - 0.003, 1.0), u=(1.0, 0, 0), v=(0, 0, 1.0), emission=(60, 55, 45), div_level=10)]
    cam = dict(eye=(0.6, 1.5, 1.5), lookat=(8.0, 1.0, 1.5), up=(0, 1, 0), fov=50.0)
    return b.finish(mats, lights, camera=cam, name="hallway")


def simple_room(n: int = 4) -> Scene:
    """Tiny closed 7-quad room used by unit tests: floor, glossy wall, blocker, quad light."""
    b = _Builder()
    b.grid((-1, 0, 1), (2, 0, 0), (0, 0, -2), n, n, 0)
    b.grid((-1, 2, -1), (2, 0, 0), (0, 0, 2), n, n, 0)
    b.grid((-1, 0, -1), (2, 0, 0), (0, 2, 0), n, n, 1)
    b.grid((1, 0, 1), (-2, 0, 0), (0, 2, 0), n, n, 0)
    b.grid((-1, 0, 1), (0, 0, -2), (0, 2, 0), n, n, 2)
    b.grid((1, 0, -1), (0, 0, 2), (0, 2, 0), n, n, 0)
    b.box((-0.3, 0.0, -0.3), (0.3, 0.7, 0.3), 2, sub=max(1, n // 2), rot_y=0.4)
    mats = [dict(color=(0.7, 0.7, 0.7), roughness=0.5, metallic=0.0),
            dict(color=(0.8, 0.5, 0.3), roughness=0.15, metallic=0.6),
            dict(color=(0.2, 0.3, 0.8), roughness=0.4, metallic=0.0)]
    lights = [dict(position=(-0.4, 1.997, -0.4), u=(0.8, 0, 0), v=(0, 0, 0.8), emission=(10, 10, 10), div_level=4)]
    cam = dict(eye=(0.0, 1.0, 0.95), lookat=(0.0, 0.8, -1.0), up=(0, 1, 0), fov=60.0)
    return b.finish(mats, lights, camera=cam, name="simple_room")


def sky_texture(width: int = 64, height: int = 32, sun=(0.62, 0.30), sun_lum: float = 60.0) -> np.ndarray:
    """A small procedural environment map as a .hdr raster ((h, w, 4) float32, row 0 = top): blue-to-white gradient plus a sun blob
    at raster position (u, v) = sun.  (The reference samples the raster as read but looks colours up in the row-flipped texture,
    so the sun is IMPORTANCE-sampled upside down: the scene keeps it away from the horizon to make that visible in the tests.)"""
    y, x = np.mgrid[0:height, 0:width].astype(np.float32)
    u, v = (x + 0.5) / width, (y + 0.5) / height
    sky = np.stack([0.25 + 0.35 * v, 0.35 + 0.35 * v, 0.7 + 0.2 * v], -1)
    d2 = ((u - sun[0]) * 2.0) ** 2 + (v - sun[1]) ** 2
    blob = sun_lum * np.exp(-d2 / (2 * 0.03 ** 2))
    rgb = sky + blob[..., None] * np.array([1.0, 0.9, 0.7], np.float32)
    return np.concatenate([rgb, np.zeros((height, width, 1))], -1).astype(np.float32)


def courtyard(n: int = 6, with_quad_light: bool = True) -> Scene:
    """An open-top yard for the environment-map rows (f4): floor, three walls (one glossy), a box that throws a sky shadow, NO
    ceiling; a dim quad light on a wall (the reference needs at least one QUAD light for its training pass) and a sky with a sun."""
    b = _Builder()
    b.grid((-1.5, 0, 1.5), (3, 0, 0), (0, 0, -3), n, n, 0)          # floor (normal +y)
    b.grid((-1.5, 0, -1.5), (3, 0, 0), (0, 1.6, 0), n, n, 1)        # back wall, glossy
    b.grid((-1.5, 0, 1.5), (0, 0, -3), (0, 1.6, 0), n, n, 2)        # left wall
    b.grid((1.5, 0, -1.5), (0, 0, 3), (0, 1.6, 0), n, n, 0)         # right wall
    b.box((-0.45, 0.0, -0.35), (0.35, 0.8, 0.45), 2, sub=max(1, n // 2), rot_y=0.5)
    mats = [dict(color=(0.7, 0.68, 0.62), roughness=0.6, metallic=0.0),
            dict(color=(0.8, 0.6, 0.4), roughness=0.2, metallic=0.5),
            dict(color=(0.25, 0.35, 0.75), roughness=0.45, metallic=0.0)]
    lights = [dict(position=(1.497, 0.9, -0.3), u=(0, 0, 0.6), v=(0, 0.4, 0), emission=(2.0, 1.6, 1.2), div_level=3)] if with_quad_light else []
    cam = dict(eye=(0.0, 1.9, 3.6), lookat=(0.0, 0.5, 0.0), up=(0, 1, 0), fov=45.0)
    sc = b.finish(mats, lights, camera=cam, name="courtyard")
    lo, hi = sc.vertices.min(0), sc.vertices.max(0)
    sc.environment = dict(rgba=sky_texture(), center=(0.5 * (lo + hi)).astype(np.float32), radius=float(np.linalg.norm(hi - lo)))
    return sc


def write_hdr(path: str, rgba: np.ndarray, rle: bool = True, exposure: float = None) -> None:
    """Writes an (h, w, >= 3) float raster as a Radiance RGBE .hdr (new-style RLE scanlines when `rle` and 8 <= w < 32768, else flat):
    the file format HDRLoader reads (scene_shift.cpp:334-500).  A test / authoring helper; mantissas are floor(value / 2^(e - 8))."""
    a = np.asarray(rgba, np.float64)[..., :3]
    h, w = a.shape[:2]
    m = a.max(-1)
    e = np.zeros_like(m, dtype=np.int64)
    nz = m > 1e-38
    e[nz] = np.floor(np.log2(m[nz])).astype(np.int64) + 1          # m < 2^e
    scale = np.where(nz, np.ldexp(1.0, (8 - e).astype(np.int64)), 0.0)
    mant = np.clip(np.floor(a * scale[..., None]), 0, 255).astype(np.uint8)
    rgbe = np.concatenate([mant, np.where(nz, e + 128, 0).astype(np.uint8)[..., None]], -1)
    rgbe[~nz] = 0
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\n# written by spcbpt-optix7_amd/scenes.py\nFORMAT=32-bit_rle_rgbe\n")
        if exposure is not None:
            f.write(b"EXPOSURE=%g\n" % exposure)
        f.write(b"\n-Y %d +X %d\n" % (h, w))
        for y in range(h):
            row = rgbe[y]
            if not rle or w < 8 or w > 0x7fff:
                f.write(row.tobytes()); continue
            f.write(bytes([2, 2, (w >> 8) & 0xff, w & 0xff]))
            for ch in range(4):
                col = row[:, ch]
                x = 0
                while x < w:
                    run = 1
                    while x + run < w and run < 127 and col[x + run] == col[x]:
                        run += 1
                    if run >= 4:
                        f.write(bytes([128 + run, int(col[x])])); x += run
                    else:   # literal span up to the next run of >= 4 (or 128 bytes)
                        start = x
                        while x < w and x - start < 128:
                            r2 = 1
                            while x + r2 < w and r2 < 4 and col[x + r2] == col[x]:
                                r2 += 1
                            if r2 >= 4:
                                break
                            x += 1
                        f.write(bytes([x - start]) + col[start:x].tobytes())


def needle_room(n_needles: int = 20000, seed: int = 11) -> Scene:
    """Test scene for deep traversal stacks: the Cornell room with a cloud of long sliver triangles that span the whole room.
    Every sliver's bounding box covers a large part of the scene, so sibling boxes overlap at every level of the BVH, a ray
    enters nearly all of them, and the per-lane traversal stack (up to three pushes per 4-wide node visit) runs far past the
    16 entries it holds in LDS -- the HBM spill path of TravStack, which compact furniture scenes hardly ever reach."""
    rng = np.random.default_rng(seed)
    b = _Builder()
    W, R, G = 0, 1, 2
    b.grid((-1, 0, 1), (2, 0, 0), (0, 0, -2), 1, 1, W)
    b.grid((-1, 2, -1), (2, 0, 0), (0, 0, 2), 1, 1, W)
    b.grid((-1, 0, -1), (2, 0, 0), (0, 2, 0), 1, 1, W)
    b.grid((-1, 0, 1), (0, 0, -2), (0, 2, 0), 1, 1, R)
    b.grid((1, 0, -1), (0, 0, 2), (0, 2, 0), 1, 1, G)
    b.grid((1, 0, 1), (-2, 0, 0), (0, 2, 0), 1, 1, W)       # front wall: closed room (the camera sits inside)
    lo, hi = np.array([-0.95, 0.05, -0.95]), np.array([0.95, 1.9, 0.95])
    p0 = rng.uniform(lo, hi, size=(n_needles, 3))
    p1 = rng.uniform(lo, hi, size=(n_needles, 3))
    d = p1 - p0
    side = np.cross(d, rng.normal(size=(n_needles, 3)))
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    p2 = p1 + side * rng.uniform(0.002, 0.006, size=(n_needles, 1))
    verts = np.stack([p0, p1, p2], 1).reshape(-1, 3)
    tris = np.arange(3 * n_needles, dtype=np.uint32).reshape(-1, 3)
    b.add(verts, np.zeros((3 * n_needles, 2), np.float32), tris, 3)
    mats = [dict(color=(0.73, 0.73, 0.73), roughness=0.5, metallic=0.0),
            dict(color=(0.65, 0.05, 0.05), roughness=0.5, metallic=0.0),
            dict(color=(0.12, 0.45, 0.15), roughness=0.5, metallic=0.0),
            dict(color=(0.8, 0.7, 0.3), roughness=0.3, metallic=0.5)]
    lights = [dict(position=(-0.25, 1.998, -0.25), u=(0.5, 0, 0), v=(0, 0, 0.5), emission=(17, 12, 4), div_level=4)]
    cam = dict(eye=(0.0, 1.0, 0.9), lookat=(0.0, 1.0, 0.0), up=(0, 1, 0), fov=70.0)
    return b.finish(mats, lights, camera=cam, name="needle_room")


def write_scene(scene: Scene, data_root: str, rel_dir: str) -> str:
    """Writes `scene` in the reference's `.scene` + OBJ format (sceneLoader.cpp grammar) under
    data_root/rel_dir: one OBJ per material (the k-th mesh block uses the k-th material), binary PPM textures,
    Quad lights with absolute corner points v1/v2.  Returns the path of the .scene file."""
    import os
    out_dir = os.path.join(data_root, rel_dir)
    os.makedirs(out_dir, exist_ok=True)
    lines = ["# written by spcbpt-optix7_amd/scenes.py in the .scene syntax of ssufujia/SPCBPT-OptiX7", ""]
    for k, m in enumerate(scene.materials):
        lines += [f"material mat{k}", "{", "   color %.9g %.9g %.9g" % tuple(m.get("color", (1, 1, 1))),
                  "   roughness %.9g" % m.get("roughness", 0.5), "   metallic %.9g" % m.get("metallic", 0.0), "   specular 0.5"]
        if m.get("brdf", 0):
            lines.append("   brdf %d" % int(m["brdf"]))   # sceneLoader.cpp:107 (the shipped house scene's `Glass`)
        if m.get("albedo_tex", 0) > 0:
            t = m["albedo_tex"] - 1
            name = f"{rel_dir}/tex{t}.ppm"
            img = scene.textures[t]
            with open(os.path.join(data_root, name), "wb") as f:
                f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
                f.write(np.ascontiguousarray(img[..., :3]).tobytes())
            lines.append(f"   albedoTex {name}")
        lines += ["}", ""]
    c = scene.camera
    lines += ["cameraSetting", "{", "    eye %.9g %.9g %.9g" % tuple(c["eye"]), "    lookat %.9g %.9g %.9g" % tuple(c["lookat"]),
              "    up %.9g %.9g %.9g" % tuple(c.get("up", (0, 1, 0))), "    fov %.9g" % c.get("fov", 35.0)]
    if getattr(scene, "environment", None):
        write_hdr(os.path.join(data_root, f"{rel_dir}/sky.hdr"), scene.environment["rgba"])
        lines.append(f"    env_file {rel_dir}/sky.hdr")
    lines += ["}", ""]
    uv = scene.texcoords if scene.texcoords is not None else np.zeros((scene.vertices.shape[0], 2), np.float32)
    for k in range(len(scene.materials)):
        tris = scene.indices[scene.tri_material == k]
        used, inv = np.unique(tris.reshape(-1), return_inverse=True)
        name = f"{rel_dir}/mesh{k}.obj"
        with open(os.path.join(data_root, name), "w") as f:
            for v in scene.vertices[used]:
                f.write("v %.9g %.9g %.9g\n" % tuple(v))
            for t in uv[used]:
                f.write("vt %.9g %.9g\n" % tuple(t))
            for a, b, cc in inv.reshape(-1, 3) + 1:
                f.write(f"f {a}/{a} {b}/{b} {cc}/{cc}\n")
        lines += ["mesh", "{", f"    file {name}", f"    material mat{k}", "}", ""]
    for l in scene.lights:
        p, u, v = (np.asarray(l[x], np.float64) for x in ("position", "u", "v"))
        lines += ["light", "{", "    position %.9g %.9g %.9g" % tuple(p), "    v1 %.9g %.9g %.9g" % tuple(p + u),
                  "    v2 %.9g %.9g %.9g" % tuple(p + v), "    emission %.9g %.9g %.9g" % tuple(l["emission"]), "    type Quad",
                  "    divLevel %d" % l.get("div_level", 1), "}", ""]
    path = os.path.join(out_dir, scene.name + ".scene")
    with open(path, "w") as f:
        f.write("\n".join(lines))
    return path


def write_gltf(scene: Scene, out_dir: str, name: str = "scene", binary: bool = False) -> str:
    """Writes `scene` as glTF 2.0 (`name`.gltf + `name`.bin + binary-PPM textures, or one `name`.glb when `binary`): one mesh
    with one TRIANGLES primitive per material (float POSITION / TEXCOORD_0, u32 indices), pbrMetallicRoughness factors, a
    perspective camera node, and this build's extensions `extras.spcbpt_quad_lights` (root) / `extras.spcbpt_lookat` (camera
    node) that csrc/gltf_file.cpp reads back.  Returns the path of the .gltf / .glb file."""
    import json
    import os
    import struct
    os.makedirs(out_dir, exist_ok=True)
    V = np.ascontiguousarray(scene.vertices, dtype=np.float32)
    UV = np.zeros((V.shape[0], 2), np.float32) if scene.texcoords is None else np.ascontiguousarray(scene.texcoords, dtype=np.float32)
    I = np.ascontiguousarray(scene.indices, dtype=np.uint32)
    M = np.asarray(scene.tri_material)
    blob = bytearray()
    views, accessors, prims = [], [], []

    def add(data: bytes, target=None):
        while len(blob) % 4: blob.append(0)
        v = {"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)}
        if target: v["target"] = target
        blob.extend(data)
        views.append(v)
        return len(views) - 1

    for k in range(len(scene.materials)):
        tri = I[M == k]
        if tri.shape[0] == 0: continue
        used, inv = np.unique(tri.reshape(-1), return_inverse=True)
        p, t, idx = V[used], UV[used], inv.astype(np.uint32)
        vp = add(p.tobytes(), 34962); vt = add(t.tobytes(), 34962); vi = add(idx.tobytes(), 34963)
        accessors += [
            {"bufferView": vp, "componentType": 5126, "count": int(p.shape[0]), "type": "VEC3",
             "min": [float(x) for x in p.min(0)], "max": [float(x) for x in p.max(0)]},
            {"bufferView": vt, "componentType": 5126, "count": int(t.shape[0]), "type": "VEC2"},
            {"bufferView": vi, "componentType": 5125, "count": int(idx.shape[0]), "type": "SCALAR"}]
        a0 = len(accessors) - 3
        prims.append({"attributes": {"POSITION": a0, "TEXCOORD_0": a0 + 1}, "indices": a0 + 2, "material": k, "mode": 4})
    mats, textures, images = [], [], []
    for k, m in enumerate(scene.materials):
        pbr = {"baseColorFactor": [float(x) for x in m.get("color", (1, 1, 1))] + [1.0],
               "metallicFactor": float(m.get("metallic", 0.0)), "roughnessFactor": float(m.get("roughness", 0.5))}
        if m.get("albedo_tex", 0) > 0:
            t = m["albedo_tex"] - 1
            img = np.ascontiguousarray(scene.textures[t], dtype=np.uint8)
            fn = f"{name}_tex{t}.ppm"
            with open(os.path.join(out_dir, fn), "wb") as f:
                f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]) + img[..., :3].tobytes())
            images.append({"uri": fn, "mimeType": "image/x-portable-pixmap"})
            textures.append({"source": len(images) - 1})
            pbr["baseColorTexture"] = {"index": len(textures) - 1}
        mats.append({"name": f"mat{k}", "pbrMetallicRoughness": pbr, "doubleSided": True})
    cam = scene.camera or dict(eye=(0, 0, 5), lookat=(0, 0, 0), up=(0, 1, 0), fov=35.0)
    # the camera node only carries position and up (what the reference's loader reads): a translation, no rotation
    nodes = [{"mesh": 0, "name": "geometry"},
             {"camera": 0, "name": "camera", "translation": [float(x) for x in cam["eye"]],
              "extras": {"spcbpt_lookat": [float(x) for x in cam["lookat"]]}}]
    doc = {"asset": {"version": "2.0", "generator": "spcbpt-optix7_amd/scenes.py"},
           "scene": 0, "scenes": [{"nodes": [0, 1]}], "nodes": nodes,
           "meshes": [{"name": scene.name, "primitives": prims}],
           "cameras": [{"type": "perspective", "perspective": {"yfov": float(np.deg2rad(cam.get("fov", 35.0))), "znear": 0.01,
                                                                 "aspectRatio": 16.0 / 9.0}}],
           "materials": mats, "accessors": accessors, "bufferViews": views,
           "extras": {"spcbpt_quad_lights": [dict(position=[float(x) for x in l["position"]], u=[float(x) for x in l["u"]],
                                                  v=[float(x) for x in l["v"]], emission=[float(x) for x in l["emission"]],
                                                  divLevel=int(l.get("div_level", 1))) for l in scene.lights]}}
    if textures: doc["textures"] = textures; doc["images"] = images
    if binary:
        doc["buffers"] = [{"byteLength": len(blob)}]
        js = json.dumps(doc, separators=(",", ":")).encode()
        js += b" " * (-len(js) % 4)
        while len(blob) % 4: blob.append(0)
        path = os.path.join(out_dir, name + ".glb")
        with open(path, "wb") as f:
            f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(blob)))
            f.write(struct.pack("<II", len(js), 0x4E4F534A)); f.write(js)
            f.write(struct.pack("<II", len(blob), 0x004E4942)); f.write(bytes(blob))
        return path
    doc["buffers"] = [{"uri": name + ".bin", "byteLength": len(blob)}]
    with open(os.path.join(out_dir, name + ".bin"), "wb") as f:
        f.write(bytes(blob))
    path = os.path.join(out_dir, name + ".gltf")
    with open(path, "w") as f:
        json.dump(doc, f)
    return path
