"""World-size-N gloo worker for tests/test_distributed_cpu.py: exercises the multi-GPU decomposition of
spcbpt-optix7_amd/dist.py (core sharding, LVC all-gather, band interleave, framebuffer sum) with the oracle standing in
for the GPU kernels.  Exits non-zero on any mismatch with the single-process result."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch.distributed as dist

import __graft_entry__ as g
from oracle import binding as ob
from tests.parity_util import minimal_tuple


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg = g.load_package()
    scene = pkg.scenes.cornell_box()
    W, H, M = 40, 56, 1200   # 7 bands of 8 rows: uneven split over 2 ranks
    o = ob.Oracle(scene, nthreads=2)
    cam = scene.camera
    o.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
    o.resize(W, H)
    o.set_light_trace(M, 52, 1)
    tup = minimal_tuple(o, 1)
    o.set_subspace(*tup)
    # start-up exchange: rank 0's tuple on every rank, bit for bit (the other ranks offer nothing)
    got = pkg.dist.broadcast_subspace(tup if rank == 0 else None, 0)
    assert all(a.tobytes() == np.ascontiguousarray(b).tobytes() for a, b in zip(got, tup)), "broadcast tuple differs"
    assert got[0].dtype == pkg.api.TREE_NODE_DTYPE and got[3].shape == (1000, 1000)
    # single-process reference of the frame
    o.launch("light trace", 3)
    full_lvc = o.lvc_read()
    o.build_sampler()
    full_tables = o.sampler_read()
    o.launch("SPCBPT_eye", 2)
    full_img = o.read_accum().copy()
    # --- the decomposition
    begin, count = pkg.dist.core_range(M, rank, world)
    shard = full_lvc[(full_lvc["path_id"] >= begin) & (full_lvc["path_id"] < begin + count)]   # what this rank's light pass yields
    gathered = pkg.dist.allgather_lvc_host(shard)
    assert gathered.tobytes() == full_lvc.tobytes(), "all-gathered LVC differs from the single-process LVC"
    o.lvc_import(gathered)
    o.build_sampler()
    t = o.sampler_read()
    assert t[3:] == full_tables[3:] and np.array_equal(t[1], full_tables[1]) and np.array_equal(t[0]["size"], full_tables[0]["size"])
    o.clear_accum()
    o.launch("SPCBPT_eye", 2, rows=pkg.dist.band_rows(H, rank, world))
    mine = o.read_accum()
    rows = pkg.dist.rows_of_rank(H, rank, world)
    other = [y for y in range(H) if y not in rows]
    assert (mine[other] == 0).all() and (mine[rows, :, 3] == 1).all()
    total = pkg.dist.allreduce_image_host(mine.copy())
    # the imported LVC is compact (slot i = vertex i) while the single-process one is padded: identical samples, so the
    # images must agree exactly
    assert np.array_equal(total, full_img), float(np.abs(total - full_img).max())
    # every core belongs to exactly one rank
    owned = np.zeros(M, np.int32)
    owned[begin:begin + count] += 1
    import torch
    tt = torch.from_numpy(owned)
    dist.all_reduce(tt)
    assert (tt.numpy() == 1).all()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("DIST_OK")


if __name__ == "__main__":
    main()
