"""Host-loop orderings of the frame pipeline (no reference counterpart: the reference syncs the device after every launch).
The light pass of frame f + 1 may be launched before frame f's sampler is built (spcbpt_set_light_ahead); export / import /
sync_light / build_sampler then address the OLDEST queued pass.  Whatever the order, a frame must be rendered from the light
pass it names: images are compared bit for bit with the plain order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
W = H = 96
FRAMES = 5


def _renderer(pkg, streams_env=None):
    scene = pkg.scenes.cornell_box()
    cam = scene.camera
    r = pkg.Renderer(scene, 0)
    r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
    r.resize(W, H)
    r.set_light_trace(3000, 64, 1)
    r.set_subspace()
    return r


def _plain(pkg):
    r = _renderer(pkg)
    lvcs = []
    for f in range(FRAMES):
        r.launch("light trace", f + 1)
        lvcs.append(r.lvc_read())
        r.build_sampler()
        r.launch("SPCBPT_eye", f)
    r.sync()
    return r.read_accum().copy(), lvcs


def test_light_pass_one_frame_ahead_renders_the_same_frames(gpu, pkg):
    want, _ = _plain(pkg)
    r = _renderer(pkg)
    r.set_light_ahead(True)
    r.launch("light trace", 1)
    for f in range(FRAMES):
        r.launch("light trace", f + 2)      # consumed by the next iteration (the last one is never built)
        r.build_sampler()                   # the OLDEST queued pass: launch frame f + 1
        r.launch("SPCBPT_eye", f)
    r.sync()
    assert np.array_equal(r.read_accum(), want)
    # back to the default: the build takes the latest pass again
    r.set_light_ahead(False)
    r.clear_accum()
    for f in range(FRAMES):
        r.launch("light trace", f + 1); r.build_sampler(); r.launch("SPCBPT_eye", f)
    r.sync()
    assert np.array_equal(r.read_accum(), want)


def test_ahead_with_export_and_device_import(gpu, pkg):
    """The sharded job's sequence on one GPU: export the oldest pass, stage it in another device buffer, import it back
    without a host wait (two staging buffers alternate), build, render -- while the next light pass is already queued."""
    import torch
    want, lvcs = _plain(pkg)
    r = _renderer(pkg)
    r.set_light_ahead(True)
    dev = torch.device("cuda", 0)
    stage = [None, None]
    r.launch("light trace", 1)
    for f in range(FRAMES):
        r.launch("light trace", f + 2)
        dv, dc, cap = r.lvc_export()
        r.sync_light()                                                      # that pass only
        n = int(pkg.dist.device_view(dc, 8, dev).view(torch.int32)[0].item())
        assert n == len(lvcs[f])
        shard = pkg.dist.device_view(dv, n * pkg.dist.VERTEX_BYTES, dev)
        r.lvc_import_wait()                                                 # the import that read stage[f & 1] has copied
        stage[f & 1] = shard.clone()
        torch.cuda.current_stream(dev).synchronize()
        r.lvc_import_device(stage[f & 1].data_ptr(), n)
        r.build_sampler()
        r.launch("SPCBPT_eye", f)
    r.sync()
    assert np.array_equal(r.read_accum(), want)


def test_default_order_still_builds_the_latest_pass(gpu, pkg):
    """Without light-ahead, light passes that were never built (the Q passes of the preprocessing, a discarded pass) do not
    queue up: the next build takes the latest one."""
    want, _ = _plain(pkg)
    r = _renderer(pkg)
    for k in range(7):
        r.launch("light trace", 100 + k)    # never built
    for f in range(FRAMES):
        r.launch("light trace", f + 1); r.build_sampler(); r.launch("SPCBPT_eye", f)
    r.sync()
    assert np.array_equal(r.read_accum(), want)


def test_batched_eye_launch_equals_frame_by_frame(gpu, pkg):
    """spcbpt_launch_eye_batch: several frames' tiles in one persistent kernel's queue.  Every pixel-sample is computed as in a
    launch of its own frame, and the merges run in frame order: the film must match bit for bit, with full images, with
    interleaved bands (a rank's share) and with a batch that is smaller than the one the context was sized for."""
    import os
    want, _ = _plain(pkg)                       # FRAMES = 5 frames, launch frames 1..5
    os.environ["SPCBPT_EYE_BATCH"] = "4"
    try:
        r = _renderer(pkg)
    finally:
        del os.environ["SPCBPT_EYE_BATCH"]
    done = []
    for f in range(FRAMES):
        r.launch("light trace", f + 1)
        r.build_sampler()
        done.append(f)
        if len(done) == 3 or f == FRAMES - 1:   # batches of 3 and 2
            r.launch_eye_batch(done)
            done = []
    r.sync()
    assert np.array_equal(r.read_accum(), want)
    # bands of a rank (every 3rd band from band 1) against the same bands rendered frame by frame
    rows = (8, H, 3)
    a = _renderer(pkg)
    for f in range(4):
        a.launch("light trace", f + 1); a.build_sampler(); a.launch("SPCBPT_eye", f, rows)
    a.sync()
    r.clear_accum()
    for f in range(4):
        r.launch("light trace", f + 1); r.build_sampler()
    r.launch_eye_batch([0, 1, 2, 3], rows)
    r.sync()
    assert np.array_equal(r.read_accum(), a.read_accum())
    # a context that was not sized for batches still renders them (its sets are guarded by events, not by the sizing)
    a.clear_accum()
    for f in range(2):
        a.launch("light trace", f + 1); a.build_sampler()
    a.launch_eye_batch([0, 1], rows)
    b = _renderer(pkg)
    for f in range(2):
        b.launch("light trace", f + 1); b.build_sampler(); b.launch("SPCBPT_eye", f, rows)
    a.sync(); b.sync()
    assert np.array_equal(a.read_accum(), b.read_accum())
    # errors: more frames than intact samplers, more than 8, a batch before any build
    c = _renderer(pkg)
    with pytest.raises(pkg.SpcbptError):
        c.launch_eye_batch([0])
    c.launch("light trace", 1); c.build_sampler()
    with pytest.raises(pkg.SpcbptError):
        c.launch_eye_batch([0, 1])
    with pytest.raises(pkg.SpcbptError):
        c.launch_eye_batch(list(range(9)))
    c.launch_eye_batch([0])
    c.sync()


def test_two_ranks_on_one_gpu_with_the_sharded_host_loop(gpu, pkg):
    """The sharded job's host loop as bench.py runs it for world > 1 -- light passes launched ahead, export of the oldest pending
    shard, gather, import without host waits, sampler build, several frames per eye launch on interleaved bands -- with two
    contexts on one GPU standing in for two ranks (the all-gather is a device-side concatenation in rank order).  The sum of the
    two films must equal the film of one context rendering whole frames the plain way, bit for bit."""
    import os
    import torch
    dev = torch.device("cuda", 0)
    scene = pkg.scenes.cornell_box()
    cam = scene.camera
    M, NF, WORLD, BATCH = 3000, 6, 2, 2
    VB = pkg.dist.VERTEX_BYTES

    def make(batch):
        if batch > 1:
            os.environ["SPCBPT_EYE_BATCH"] = str(batch)
        try:
            r = pkg.Renderer(scene, 0)
        finally:
            os.environ.pop("SPCBPT_EYE_BATCH", None)
        r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
        r.resize(W, H)
        r.set_light_trace(M, 64, 1)
        r.set_subspace()
        return r

    single = make(1)
    tup = single.get_subspace()
    for f in range(NF):
        single.launch("light trace", f + 1); single.build_sampler(); single.launch("SPCBPT_eye", f)
    single.sync()
    want = single.read_accum().copy()

    ranks = []
    for k in range(WORLD):
        r = make(BATCH)
        r.set_subspace(*tup)                                   # the broadcast tuple
        b, c = pkg.dist.core_range(M, k, WORLD)
        r.set_light_trace(M, 64, 1, core_begin=b, core_count=c)
        r.set_light_ahead(True)
        r.launch("light trace", 1)                             # the pass running ahead
        ranks.append(r)
    stage = [[None, None] for _ in range(WORLD)]
    queued = []
    for f in range(NF):
        shards = []
        for r in ranks:
            r.launch("light trace", f + 2)
            dv, dc, cap = r.lvc_export()
            r.sync_light()
            n = int(pkg.dist.device_view(dc, 8, dev).view(torch.int32)[0].item())
            shards.append(pkg.dist.device_view(dv, n * VB, dev))
        gathered = torch.cat(shards)                           # rank order = global (path, depth) order
        total = gathered.numel() // VB
        torch.cuda.current_stream(dev).synchronize()
        for k, r in enumerate(ranks):
            r.lvc_import_wait()
            stage[k][f & 1] = gathered.clone()
            torch.cuda.current_stream(dev).synchronize()
            r.lvc_import_device(stage[k][f & 1].data_ptr(), total)
            r.build_sampler()
        queued.append(f)
        if len(queued) == BATCH:
            for k, r in enumerate(ranks):
                r.launch_eye_batch(queued, pkg.dist.band_rows(H, k, WORLD))
            queued = []
    for r in ranks:
        r.sync()
    films = [r.read_accum() for r in ranks]
    rows0 = pkg.dist.rows_of_rank(H, 0, WORLD)
    rows1 = pkg.dist.rows_of_rank(H, 1, WORLD)
    assert (films[0][rows1] == 0).all() and (films[1][rows0] == 0).all()      # a rank writes its own bands only
    assert np.array_equal(films[0] + films[1], want)


def test_batched_eye_launch_where_waves_cross_frame_boundaries(gpu, pkg):
    """At 96 x 96 a frame is 144 tiles and a wave hardly ever holds paths of two frames at once; at 448 x 448 with four frames in
    the queue (12 544 tiles for ~3 900 resident waves) most waves regenerate across a frame boundary while a lane still owes the
    connections of a vertex of the previous frame (the parked pixel): those connections must read the sampler tables of THEIR
    frame.  (A first version published the lane's new frame id with the old vertex; only a run at bench scale showed it.)"""
    import os
    scene = pkg.scenes.cornell_box()
    cam = scene.camera
    S, NF = 448, 4

    def make(batch):
        if batch > 1:
            os.environ["SPCBPT_EYE_BATCH"] = str(batch)
        try:
            r = pkg.Renderer(scene, 0)
        finally:
            os.environ.pop("SPCBPT_EYE_BATCH", None)
        r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], 1.0)
        r.resize(S, S)
        r.set_light_trace(20000, 64, 1)
        r.set_subspace()
        return r

    a = make(1)
    for f in range(NF):
        a.launch("light trace", f + 1); a.build_sampler(); a.launch("SPCBPT_eye", f)
    a.sync()
    want = a.read_accum().copy()
    b = make(NF)
    for rep in range(2):                      # twice: the result must not depend on how the waves happened to be scheduled
        b.clear_accum()
        for f in range(NF):
            b.launch("light trace", f + 1); b.build_sampler()
        b.launch_eye_batch(list(range(NF)))
        b.sync()
        got = b.read_accum()
        assert np.array_equal(got, want), (rep, int((np.abs(got - want).max(axis=2) > 0).sum()))


def test_cache_sets_are_sized_from_a_probe_pass_and_an_overflow_is_reported(gpu, pkg):
    """The compact light-vertex caches exist once per frame in flight, so they are sized from the cache a pass really produces
    (probe pass at the first light pass: 2 x its vertex count), not from num_core x core_padding; a pass that outgrows a
    hand-set capacity is cut off and reported at the next sync (SPCBPT_ERR_CAPACITY), never written past the end."""
    want, lvcs = _plain(pkg)
    r = _renderer(pkg)
    r.launch("light trace", 1)
    n = len(r.lvc_read())
    v, sets = r.lvc_capacity()
    assert n == len(lvcs[0]) and n < v <= max(2 * n, n + 65536) + 4096 and v <= 3000 * 64 and sets >= 3
    # by hand: too small -> reported, images invalid; large enough -> the plain frames again
    scene = pkg.scenes.cornell_box()
    cam = scene.camera
    r2 = pkg.Renderer(scene, 0)
    r2.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
    r2.resize(W, H)
    r2.set_light_trace(3000, 64, 1)
    r2.lvc_set_capacity(n // 2)                          # before anything is allocated (the sets never shrink)
    r2.set_subspace(*r.get_subspace())
    r2.launch("light trace", 1); r2.build_sampler(); r2.launch("SPCBPT_eye", 0)
    with pytest.raises(pkg.SpcbptError, match="cache overflow"):
        r2.sync()
    assert len(r2.lvc_read()) == n // 2                  # cut off at the capacity
    r2.lvc_set_capacity(n + n // 4)
    r2.clear_accum()
    for f in range(FRAMES):
        r2.launch("light trace", f + 1); r2.build_sampler(); r2.launch("SPCBPT_eye", f)
    r2.sync()
    assert np.array_equal(r2.read_accum(), want)
    assert r2.lvc_capacity()[0] == n + n // 4
