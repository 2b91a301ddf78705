"""Compares two dumps of tools/film_dump.py bit for bit.  usage: python tools/film_cmp.py gpurun_out/film_A.npz gpurun_out/film_B.npz"""
import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = 0
for k in a.files:
    same = np.array_equal(a[k], b[k])
    print(k, "identical" if same else f"DIFFERENT in {int((a[k] != b[k]).any(-1).sum())} pixels, max |d| {float(np.abs(a[k] - b[k]).max()):.3g}")
    bad += not same
sys.exit(1 if bad else 0)
