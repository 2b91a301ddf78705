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
