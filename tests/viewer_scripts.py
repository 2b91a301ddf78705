"""Event scripts shared by tests/golden/make_golden.py (reference side) and tests/test_viewer.py (product side)."""
import numpy as np


def scripts():
    rng = np.random.default_rng(77)
    out = []
    cams = [((0.0, 1.0, 3.5), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 35.0, 1920 / 1000),
            ((2.5, 1.7, -4.0), (0.3, 0.9, 0.2), (0.0, 1.0, 0.0), 45.0, 1.0),
            ((-3.0, 0.4, 0.5), (1.0, 2.0, -1.0), (0.1, 0.9, 0.2), 60.0, 16 / 9),
            ((0.0, 6.0, 0.01), (0.0, 0.0, 0.0), (0.0, 0.0, -1.0), 30.0, 4 / 3)]
    for k, (eye, lookat, up, fov, aspect) in enumerate(cams):
        ev = []
        x, y = 500.0, 400.0
        for seg in range(6):
            button = int(rng.integers(0, 3))           # left orbit, right turn, middle: ignored by the callbacks
            ev.append((0, button, x, y))
            for _ in range(int(rng.integers(3, 12))):
                x += float(rng.integers(-60, 61)) + (0.5 if seg % 2 else 0.0)   # fractional cursor positions truncate
                y += float(rng.integers(-45, 46))
                ev.append((2, 0, x, y))
            ev.append((1, button, x, y))
            ev.append((2, 0, x + 13, y - 7))           # a move with no button down changes nothing
            for _ in range(int(rng.integers(0, 4))):
                ev.append((3, float(rng.choice([-1, 1, 2])), 0, 0))
            if seg % 2 == 0:
                for _ in range(int(rng.integers(1, 5))):
                    ev.append((4, float(rng.choice([60.0, 23.7, 144.0, 8.25])), 0, 0))
        # latitude clamp at +-89 degrees and longitude wrap through 360
        ev.append((0, 0, 100.0, 100.0))
        for i in range(1, 30):
            ev.append((2, 0, 100.0 + 97 * i, 100.0 + 31 * i))
        for i in range(1, 30):
            ev.append((2, 0, 100.0 + 97 * 29 - 55 * i, 100.0 + 31 * 29 - 77 * i))
        ev.append((1, 0, 0, 0))
        out.append(dict(eye=np.array(eye, np.float32), lookat=np.array(lookat, np.float32), up=np.array(up, np.float32),
                        fov=np.float32(fov), aspect=np.float32(aspect), events=np.array(ev, np.float64)))
    return out
