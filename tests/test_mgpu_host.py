"""The C++ N-GPU host (libspcbpt_mgpu.so, include/spcbpt_mgpu.h).

CPU: the library loads and exports every symbol its header declares.
GPU: (a) `world` ranks sharing the one GPU of the test box (local transport: device copies stand in for RCCL, everything else --
fixed-capacity shards, device-side counts, gathered compaction, sampler build over an upper bound, band gather -- is the code the
RCCL ranks run): the gathered film must equal, bit for bit, the frames ONE context renders the plain way; (b) the RCCL transport
itself at world size 1 (ncclAllGather / ncclBroadcast / ncclAllReduce really run); (c) a shard that does not fit the agreed
capacity is reported, not truncated.  The 8-GPU run of BASELINE config 4 is the driver's (bench.py --gpus 8 uses this host)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 96, 88          # 11 bands: not a multiple of the rank count


def test_every_declared_symbol_is_exported(hip_lib, pkg):
    src = open(os.path.join(ROOT, "include", "spcbpt_mgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(spcbpt_comm_[a-z_0-9]+)\s*\(", src)))
    lib = pkg.dist.load_mgpu()
    assert sorted(pkg.dist.MGPU_SYMBOLS) == names and len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert lib.spcbpt_comm_exchange_lvc(None) == -1 and lib.spcbpt_comm_gather_film(None, None) == -1   # null communicator: an error, not a crash


def _make(pkg, scene, batch, lt=(3000, 64, 1)):
    """(a scene with an environment map gets it installed: the sky is then light n_lights - 1 of every context)"""
    if batch > 1:
        os.environ["SPCBPT_EYE_BATCH"] = str(batch)
    try:
        r = pkg.Renderer(scene, 0)
    finally:
        os.environ.pop("SPCBPT_EYE_BATCH", None)
    cam = scene.camera
    r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
    r.resize(W, H)
    if getattr(scene, "environment", None):
        e = scene.environment
        r.set_environment(e["rgba"], e["center"], e["radius"])
    r.set_light_trace(*lt)
    return r


def _single(pkg, scene, nf, lt=(3000, 64, 1)):
    r = _make(pkg, scene, 1, lt)
    r.set_subspace()
    for f in range(nf):
        r.launch("light trace", f + 1); r.build_sampler(); r.launch("SPCBPT_eye", f)
    r.sync()
    return r, r.read_accum().copy()


@pytest.mark.gpu
@pytest.mark.parametrize("world,batch,lt,lbatch,xbatch,scene_name", [(2, 2, (3000, 64, 1), False, False, "cornell"), (3, 1, (3000, 64, 1), False, False, "cornell"), (2, 2, (40, 200, 60), False, False, "cornell"),
                                                          (3, 3, (3000, 64, 1), True, False, "cornell"),   # lbatch: a batch's light passes as one launch too
                                                          (3, 3, (3000, 64, 1), True, True, "cornell"),    # xbatch: ... and their shards as ONE exchange
                                                          (3, 2, (1000, 64, 1), True, True, "cornell"),    # 1000 cores on 3 ranks: 333 / 333 / 334 (uneven scratch sizes)
                                                          (7, 3, (1000, 64, 1), True, True, "cornell"),
                                                          (3, 3, (3000, 64, 1), True, True, "courtyard"),  # environment map: direction flags of the light vertices travel through the exchange
                                                          (2, 1, (3000, 64, 1), False, False, "courtyard")])
def test_local_ranks_reproduce_the_single_gpu_film(gpu, pkg, world, batch, lt, lbatch, xbatch, scene_name):
    scene = pkg.scenes.cornell_box() if scene_name == "cornell" else pkg.scenes.courtyard()
    NF = 6
    single, want = _single(pkg, scene, NF, lt)
    ranks = []
    for k in range(world):
        r = _make(pkg, scene, batch, lt)
        b, c = pkg.dist.core_range(lt[0], k, world)
        r.set_light_trace(*lt, core_begin=b, core_count=c)
        ranks.append(r)
    ranks[0].set_subspace(*single.get_subspace())             # "rank 0 trained"
    comms = pkg.dist.Comm.local(ranks)
    for c in reversed(comms):                                 # non-root ranks may ask before the root has published
        c.broadcast_subspace(0)
    for c in comms:
        c.calibrate(passes=2, slack=1.5)
    cap = comms[0].shard_capacity
    assert all(c.shard_capacity == cap for c in comms) and cap < lt[0] * lt[1]
    assert all(c.info() == (k, world, "local") for k, c in enumerate(comms))
    for r in ranks:
        r.set_light_ahead(True)
        if lbatch: r.launch_light_batch(1, batch)             # the passes running ahead: one batch
        else: r.launch("light trace", 1)
    queued = []
    for f in range(NF):
        for r in ranks:
            if not lbatch: r.launch("light trace", f + 2)
            elif f % batch == 0: r.launch_light_batch(f + 1 + batch, batch)
        if xbatch:
            if f % batch == 0:
                for c in comms:
                    c.exchange_lvc_batch(batch)               # the batch's oldest pending passes: one gather, one compaction kernel
        else:
            for c in comms:
                c.exchange_lvc()                              # no host wait; completes when the last rank has posted
        for r in ranks:
            r.build_sampler()                                 # item count stays on the device
        queued.append(f)
        if len(queued) == batch:
            for k, r in enumerate(ranks):
                if batch == 1: r.launch("SPCBPT_eye", queued[0], pkg.dist.band_rows(H, k, world))
                else: r.launch_eye_batch(queued, pkg.dist.band_rows(H, k, world))
            queued = []
    for r in ranks:
        r.sync()
    own = [r.read_accum().copy() for r in ranks]
    for k in range(world):                                    # before the gather a rank holds its own bands only
        others = [y for y in range(H) if (y // 8) % world != k]
        assert (own[k][others] == 0).all()
    for c in comms:
        c.gather_film()
    for r in ranks:
        assert np.array_equal(r.read_accum(), want)
    for c in comms:                                           # a second read-out gathers the same image (no double counting)
        c.gather_film()
    assert np.array_equal(ranks[-1].read_accum(), want)
    # the gathered sampler is the single-GPU sampler
    sub, cmfs, jump, vc, pc = ranks[0].sampler_read()
    single.launch("light trace", NF); single.build_sampler()  # the frame the last exchange carried: launch frame NF
    s2 = single.sampler_read()
    assert (vc, pc) == (s2[3], s2[4]) and np.array_equal(jump, s2[2]) and np.array_equal(cmfs, s2[1])
    if scene_name == "courtyard":                             # the scene really has sky vertices (spcbpt.h: pad bits 31 / 30)
        lv = single.lvc_read()                                # (the films above can only agree if the flags survived the exchange: a sky vertex connects differently)
        assert ((lv["pad"] & 0x80000000) != 0).sum() > 500 and ((lv["pad"] & 0x40000000) != 0).sum() > 100
    for c in comms:
        c.close()


@pytest.mark.gpu
def test_rccl_transport_at_world_size_one(gpu, pkg):
    scene = pkg.scenes.cornell_box()
    NF = 4
    single, want = _single(pkg, scene, NF)
    r = _make(pkg, scene, 2)
    c = pkg.dist.Comm(r, 0, 1, pkg.dist.unique_id())          # ncclCommInitRank
    r.set_subspace(*single.get_subspace())
    c.broadcast_subspace(0)
    c.calibrate(passes=1)
    assert c.info() == (0, 1, "rccl")                         # ncclCommUserRank / ncclCommCount
    r.set_light_ahead(True)
    r.launch("light trace", 1)
    q = []
    for f in range(NF):
        r.launch("light trace", f + 2)
        c.exchange_lvc()                                      # ncclAllGather of counts and shard on the communicator's stream
        r.build_sampler()
        q.append(f)
        if len(q) == 2:
            r.launch_eye_batch(q); q = []
    c.barrier()
    c.gather_film()
    assert np.array_equal(r.read_accum(), want)
    assert c.max_double(3.5) == 3.5
    c.close()


@pytest.mark.gpu
def test_rccl_batched_exchange_at_world_size_one(gpu, pkg):
    """spcbpt_comm_exchange_lvc_batch over the RCCL transport: pack -> ncclAllGather of 2 n counts and n x capacity vertices (one
    group) -> one compaction kernel with grid.y = frame."""
    scene = pkg.scenes.cornell_box()
    NF, B = 6, 3
    single, want = _single(pkg, scene, NF)
    r = _make(pkg, scene, B)
    c = pkg.dist.Comm(r, 0, 1, pkg.dist.unique_id())
    r.set_subspace(*single.get_subspace())
    c.broadcast_subspace(0)
    c.calibrate(passes=1)
    r.set_light_ahead(True)
    r.launch_light_batch(1, B)
    q = []
    for f in range(NF):
        if f % B == 0:
            r.launch_light_batch(f + 1 + B, B)
            c.exchange_lvc_batch(B)
        r.build_sampler()
        q.append(f)
        if len(q) == B:
            r.launch_eye_batch(q); q = []
    c.barrier()
    c.gather_film()
    assert np.array_equal(r.read_accum(), want)
    c.exchange_lvc_batch(B)                                   # the batch still ahead
    for _ in range(B):
        r.build_sampler()
    with pytest.raises(pkg.SpcbptError, match="pending"):     # nothing left to exchange: an error, not a stale gather
        c.exchange_lvc_batch(B)
    c.close()


@pytest.mark.gpu
def test_batched_builds_after_a_batched_exchange(gpu, pkg):
    """spcbpt_build_sampler_batch on gathered imports (the sets' totals live on the device: the builds run over the agreed upper
    bound): the loop of bench.py -- light batch, ONE exchange per batch, ONE set of build launches per batch, one eye launch -- over the
    RCCL transport at world size 1, against the single-GPU film."""
    scene = pkg.scenes.cornell_box()
    NF, B = 6, 3
    single, want = _single(pkg, scene, NF)
    r = _make(pkg, scene, B)
    c = pkg.dist.Comm(r, 0, 1, pkg.dist.unique_id())
    r.set_subspace(*single.get_subspace())
    c.broadcast_subspace(0)
    c.calibrate(passes=1)
    r.set_light_ahead(True)
    r.launch_light_batch(1, B)
    for f0 in range(0, NF, B):
        r.launch_light_batch(f0 + 1 + B, B)
        c.exchange_lvc_batch(B)
        r.build_sampler_batch(B)
        r.launch_eye_batch(list(range(f0, f0 + B)))
    c.barrier()
    c.gather_film()
    assert np.array_equal(r.read_accum(), want)
    c.close()


@pytest.mark.gpu
def test_an_uncalibrated_shard_capacity_beyond_the_cache_is_staged_not_refused(gpu, pkg):
    """The caches are sized per rank from each rank's own probe pass (spcbpt_lvc_set_capacity) while the default shard capacity is
    the largest padded scratch of any rank: without spcbpt_comm_calibrate a shard of shard_cap slots is more than a cache holds.
    That used to be refused -- per rank, BEFORE the collective, so one rank could refuse and leave the others in the all-gather.
    Now such a cache is staged through the communicator's send buffer: every rank always posts, nothing is read out of bounds,
    and the gathered cache is the single-GPU cache."""
    scene = pkg.scenes.cornell_box()
    ranks = []
    for k in range(2):
        r = _make(pkg, scene, 1, (3000, 400, 1))              # 400 padded slots per core: scratch 600 000 per rank
        b, c = pkg.dist.core_range(3000, k, 2)
        r.set_light_trace(3000, 400, 1, core_begin=b, core_count=c)
        ranks.append(r)
    single = _make(pkg, scene, 1, (3000, 400, 1))
    single.set_subspace()
    for r in ranks:
        r.set_subspace(*single.get_subspace())
    comms = pkg.dist.Comm.local(ranks)
    for r in ranks:
        r.launch("light trace", 1)
    v, _ = ranks[0].lvc_capacity()
    assert 0 < v < 600000 and comms[0].shard_capacity == 600000
    for c in comms:
        c.exchange_lvc()
    single.launch("light trace", 1); single.build_sampler()
    want = single.sampler_read()
    for r in ranks:
        r.build_sampler()
        sub, cmfs, jump, vc, pc = r.sampler_read()
        assert (vc, pc) == (want[3], want[4]) and np.array_equal(jump, want[2]) and np.array_equal(cmfs, want[1])
    for c in comms:
        c.close()


@pytest.mark.gpu
def test_a_shard_that_does_not_fit_is_reported(gpu, pkg):
    scene = pkg.scenes.cornell_box()
    ranks = []
    for k in range(2):
        r = _make(pkg, scene, 1)
        b, c = pkg.dist.core_range(3000, k, 2)
        r.set_light_trace(3000, 64, 1, core_begin=b, core_count=c)
        ranks.append(r)
    ranks[0].set_subspace(); ranks[1].set_subspace(*ranks[0].get_subspace())
    comms = pkg.dist.Comm.local(ranks)
    for c in comms:
        c.set_shard_capacity(1024)                            # a rank's shard holds ~4 500 vertices
    for r in ranks:
        r.launch("light trace", 1)
    for c in comms:
        c.exchange_lvc()
    for r in ranks:
        r.build_sampler()
    with pytest.raises(pkg.SpcbptError, match="shard"):
        ranks[0].sync()
    for c in comms:
        c.close()
