"""Row f3: the windowless viewer (csrc/viewer.cpp, spcbpt_viewer_*).  The camera / trackball arithmetic is pinned BIT-EXACTLY
against the reference's own sutil/Trackball.cpp + sutil/Camera.cpp (tests/golden/ref_trackball.npz from oracle/_ref, and the
library itself when it is present); the application glue (callbacks, updateState, render loop) is checked as a state
machine on the CPU and end to end on the GPU."""
import os

import numpy as np
import pytest

from tests.viewer_scripts import scripts

G = os.path.join(os.path.dirname(__file__), "golden")


def _viewer(pkg, s, renderer=None, w=1920, h=1000):
    return pkg.api.Viewer(renderer, s["eye"], s["lookat"], s["up"], float(s["fov"]), w, h)


def test_trackball_and_camera_match_reference_vectors_bit_exactly(hip_lib, pkg):
    d = np.load(os.path.join(G, "ref_trackball.npz"))
    for k, s in enumerate(scripts()):
        assert np.array_equal(d[f"events{k}"], s["events"])
        # U scales with the window aspect: give the viewer a window of exactly the script's aspect
        w, h = {0: (1920, 1000), 1: (800, 800), 2: (1920, 1080), 3: (1024, 768)}[k]
        assert np.float32(w) / np.float32(h) == s["aspect"]
        got = _viewer(pkg, s, w=w, h=h).replay(s["events"])
        want = d[f"cam{k}"]
        bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))[0]
        assert bad.size == 0, (k, bad[:5], got[bad[:1]], want[bad[:1]])


def test_against_the_reference_library_when_present(hip_lib, pkg, ob):
    s = scripts()[1]
    ref = ob.ref_viewer_replay(s["eye"], s["lookat"], s["up"], float(s["fov"]), float(s["aspect"]), s["events"])
    if ref is None:
        pytest.skip("oracle/_ref not built here")
    got = _viewer(pkg, s, w=800, h=800).replay(s["events"])
    assert got.tobytes() == ref.tobytes()


def test_state_machine_without_a_context(hip_lib, pkg):
    s = scripts()[0]
    v = _viewer(pkg, s)
    st = v.state()
    assert st["alg"] == "SPCBPT_eye" and st["subframe_index"] == 0 and st["camera_changed"] == 1 and st["render_fps"] == 60.0
    for k in range(3):
        v.frame()
    assert v.state()["subframe_index"] == 3 and v.state()["camera_changed"] == 0
    # a drag restarts the accumulation at the next frame, and only once
    v.mouse_button("left", 1, 10, 10); v.cursor_pos(30, 18); v.mouse_button("left", 0, 30, 18)
    assert v.state()["camera_changed"] == 1 and v.state()["subframe_index"] == 3
    v.frame()
    assert v.state()["subframe_index"] == 1
    v.frame()
    assert v.state()["subframe_index"] == 2
    # a cursor move without a button, or with the middle button, is not a camera change
    before = v.state()
    v.cursor_pos(300, 300); v.mouse_button("middle", 1, 300, 300); v.cursor_pos(350, 320); v.mouse_button("middle", 0, 350, 320)
    after = v.state()
    assert after["camera_changed"] == 0 and np.array_equal(before["eye"], after["eye"])
    # SPACE cycles pt <-> SPCBPT_eye and restarts; P restarts on every frame until pressed again
    v.key("SPACE"); assert v.state()["alg"] == "pt"
    v.frame(); assert v.state()["subframe_index"] == 1
    v.key("SPACE"); assert v.state()["alg"] == "SPCBPT_eye"
    v.frame(); v.frame()
    v.key("P"); v.frame(); v.frame()
    assert v.state()["one_frame_render_only"] == 1 and v.state()["subframe_index"] == 1
    v.key("P"); v.frame()
    assert v.state()["subframe_index"] == 2
    # key release / repeat only matter for W (outside the PRESS test in the reference)
    v.key("SPACE", 0); assert v.state()["alg"] == "SPCBPT_eye"
    e0 = v.state()["eye"].copy()
    v.set_fps(50.0); v.key("W", 2); v.key("W", 0)
    st = v.state()
    d = st["lookat"] - st["eye"]
    assert np.allclose(st["eye"] - e0, 2 * 0.5 / 50.0 * d / np.linalg.norm(d), atol=1e-6) and st["camera_changed"] == 1
    # resize: clamped to >= 1, ignored while minimised, changes the aspect of U
    v.window_size(800, 0); assert (v.state()["width"], v.state()["height"]) == (800, 1)
    v.iconify(1); v.window_size(640, 480); assert v.state()["width"] == 800
    v.iconify(0); v.window_size(640, 480)
    st = v.state()
    assert (st["width"], st["height"]) == (640, 480) and abs(np.linalg.norm(st["U"]) / np.linalg.norm(st["V"]) - 640 / 480) < 1e-6
    # wheel: zoom in divides the eye-lookat distance by 1.1
    d0 = np.linalg.norm(st["eye"] - st["lookat"])
    v.scroll(1)
    st = v.state()
    assert abs(np.linalg.norm(st["eye"] - st["lookat"]) * 1.1 - d0) < 1e-5
    v.key("ESCAPE"); assert v.state()["should_close"] == 1
    assert hip_lib.spcbpt_viewer_frame(None) == -1 and hip_lib.spcbpt_viewer_key(None, 32, 1) == -1


@pytest.mark.gpu
def test_viewer_loop_renders_what_the_plain_loop_renders(gpu, pkg):
    """The viewer's frames are the render loop of optixPathTracer.cpp:791-822: the image after a drag + N frames equals N
    subframes rendered by hand with the camera the drag produced; switching the algorithm restarts the accumulation."""
    scene = pkg.scenes.cornell_box()
    cam = scene.camera
    W = H = 96

    def make():
        r = pkg.Renderer(scene, 0)
        r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], 1.0)
        r.resize(W, H)
        r.set_light_trace(3000, 64, 1)
        r.set_subspace()
        return r

    a = make()
    v = pkg.api.Viewer(a, cam["eye"], cam["lookat"], cam["up"], cam["fov"], W, H)
    v.frame(); v.frame()
    v.mouse_button("left", 1, 40, 40); v.cursor_pos(65, 52); v.mouse_button("left", 0, 65, 52)
    v.scroll(1)
    for _ in range(3):
        v.frame()
    st = v.state()
    assert st["subframe_index"] == 3 and st["alg"] == "SPCBPT_eye"
    img_v = a.read_accum().copy()
    b = make()
    b.set_camera(st["eye"], st["U"], st["V"], st["W"])
    for f in range(3):
        b.render_frame("SPCBPT_eye", f, launch_frame=3 + f)   # the viewer's light pass counter kept running: frames 3, 4, 5
    b.sync()
    assert np.array_equal(img_v, b.read_accum())
    v.key("SPACE"); v.frame()
    b.clear_accum(); b.render_frame("pt", 0); b.sync()
    assert v.state()["alg"] == "pt" and np.array_equal(a.read_accum(), b.read_accum())


@pytest.mark.gpu
def test_viewer_pipeline_modes_render_the_same_frames(gpu, pkg):
    """spcbpt_viewer_set_pipeline: 0 = the reference's order, 1 = the next frame's light pass beside this frame's eye kernel, 2 (the
    default) = the next frame traced speculatively while this one is shown (deferred film merge; dropped when the camera, the size,
    the algorithm or the subframe counter changed under it).  Same launch frames, same caches, the same images bit for bit after
    EVERY call -- steady views, a camera drag, a scroll, the W key, an algorithm switch and back, a resize, the P key (every frame
    restarts) and out of it again."""
    scene = pkg.scenes.cornell_box()
    cam = scene.camera
    W = H = 96
    runs, stats = [], []
    for mode in (0, 1, 2):
        r = pkg.Renderer(scene, 0)
        r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], 1.0)
        r.resize(W, H)
        r.set_light_trace(3000, 64, 1)
        r.set_subspace()
        v = pkg.api.Viewer(r, cam["eye"], cam["lookat"], cam["up"], cam["fov"], W, H)
        v.set_fps(60.0)                                     # the W key's step divides by the frame rate: fixed, not measured
        if mode != 2:
            v.set_pipeline(mode)                            # 2 is what a new viewer does
        shots = []

        def show(n=1):
            for _ in range(n):
                v.frame()
                shots.append((v.state()["subframe_index"], v.state()["alg"], r.read_accum().copy(), r.read_frame().copy()))

        show(4)
        v.mouse_button("left", 1, 40, 40); v.cursor_pos(60, 50); v.mouse_button("left", 0, 60, 50)
        show(3)
        v.scroll(1); show(2)
        v.key("W", 1); show(2)
        v.key("SPACE"); show(3)                             # "pt": no light pass; one (and a built sampler) may be held from before
        assert v.state()["alg"] == "pt"
        v.key("SPACE"); show(3)                             # back: the held sampler is the next light pass in sequence
        v.window_size(80, 64); show(3)
        v.key("P"); show(3)                                 # one_frame_render_only: subframe 0 every frame, nothing to speculate on
        v.key("P"); show(3)
        runs.append(shots)
    for k, (a, b, c) in enumerate(zip(*runs)):
        assert a[0] == b[0] == c[0] and a[1] == b[1] == c[1], k
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[2], c[2]), k          # linear accumulation buffer
        assert np.array_equal(a[3], b[3]) and np.array_equal(a[3], c[3]), k          # tone-mapped frame


@pytest.mark.gpu
def test_deferred_launch_is_the_plain_launch_once_merged(gpu, pkg):
    """spcbpt_launch_deferred + spcbpt_merge_deferred(1) = spcbpt_launch; (0) leaves accum and frame untouched; a second render launch
    while one is outstanding is refused; spcbpt_sync_film returns with the shown frame complete."""
    scene = pkg.scenes.cornell_box()
    cam = scene.camera

    def make():
        r = pkg.Renderer(scene, 0)
        r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], 1.0)
        r.resize(64, 64)
        r.set_light_trace(3000, 64, 1)
        r.set_subspace()
        return r

    a, b = make(), make()
    for f in range(3):
        a.render_frame("SPCBPT_eye", f)
        b.launch("light trace", f + 1); b.build_sampler()
        b.launch_deferred("SPCBPT_eye", f)
        with pytest.raises(pkg.SpcbptError, match="deferred"):
            b.launch("pt", 0)
        b.merge_deferred(True)
        b.sync_film()
    a.sync()
    assert np.array_equal(a.read_accum(), b.read_accum())
    before = b.read_accum().copy()
    b.launch_deferred("pt", 3)
    b.merge_deferred(False)
    b.sync()
    assert np.array_equal(before, b.read_accum())
    with pytest.raises(pkg.SpcbptError, match="no deferred"):
        b.merge_deferred(True)


def _small_renderer(pkg, size=64):
    scene = pkg.scenes.cornell_box()
    cam = scene.camera
    r = pkg.Renderer(scene, 0)
    r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], 1.0)
    r.resize(size, size)
    r.set_light_trace(3000, 64, 1)
    r.set_subspace()
    return r, cam


@pytest.mark.gpu
def test_destroying_the_viewer_hands_the_context_back_as_it_was_found(gpu, pkg):
    """After spcbpt_viewer_frame the context holds a frame traced ahead (deferred), a light pass launched ahead and light-ahead
    mode.  spcbpt_viewer_destroy drops all of it and restores the mode it found: the plain loop (light pass, build, eye launch)
    on the same context then renders what it renders on a fresh one -- no SPCBPT_ERR_STATE, no sampler of a queued pass."""
    a, cam = _small_renderer(pkg)
    b, _ = _small_renderer(pkg)
    assert a.pipeline_state()["light_ahead"] == 0
    v = pkg.api.Viewer(a, cam["eye"], cam["lookat"], cam["up"], cam["fov"], 64, 64)
    assert a.pipeline_state()["light_ahead"] == 1
    for _ in range(3):
        v.frame()
    st = a.pipeline_state()
    assert st["deferred"] == 1 and st["pending_passes"] == 1 and st["sampler_intact"] == 1
    with pytest.raises(pkg.SpcbptError, match="deferred"):
        a.launch("SPCBPT_eye", 0)
    v.close()
    st = a.pipeline_state()
    assert st == {"light_ahead": 0, "pending_passes": 0, "sampler_intact": st["sampler_intact"], "deferred": 0}
    for r in (a, b):
        r.clear_accum()
        for f in range(3):
            r.render_frame("SPCBPT_eye", f, launch_frame=100 + f)
        r.sync()
    assert np.array_equal(a.read_accum(), b.read_accum())
    # a viewer created on a context that already runs passes ahead leaves that mode on
    a.set_light_ahead(True)
    v = pkg.api.Viewer(a, cam["eye"], cam["lookat"], cam["up"], cam["fov"], 64, 64)
    v.frame(); v.close()
    assert a.pipeline_state()["light_ahead"] == 1


@pytest.mark.gpu
def test_context_destroyed_before_its_viewer(gpu, pkg):
    """The host tears down in the "wrong" order (Renderer.close(), then Viewer.close() -- or the garbage collector does): the
    library forgets the context in its live viewers at spcbpt_destroy, the viewer goes on as the state machine of a null context
    (events and frames still advance its state, nothing is launched) and its destruction touches no freed context."""
    r, cam = _small_renderer(pkg)
    v = pkg.api.Viewer(r, cam["eye"], cam["lookat"], cam["up"], cam["fov"], 64, 64)
    for _ in range(3):
        v.frame()
    assert r.pipeline_state()["deferred"] == 1          # a frame traced ahead and a light pass are queued on the context
    r.close()                                           # waits for what is queued, then frees the context
    before = v.state()["subframe_index"]
    v.mouse_button("left", 1, 10, 10); v.cursor_pos(20, 14); v.mouse_button("left", 0, 20, 14)
    v.frame()                                           # state machine only: camera changed -> subframe restarts at 0, then counts
    assert v.state()["subframe_index"] == 1 and before == 3
    v.close()
    # and the usual order still works on a fresh context
    r2, cam = _small_renderer(pkg)
    v2 = pkg.api.Viewer(r2, cam["eye"], cam["lookat"], cam["up"], cam["fov"], 64, 64)
    v2.frame(); v2.close(); r2.close()


@pytest.mark.gpu
def test_moving_camera_does_not_speculate_and_read_film_shows_the_frame(gpu, pkg):
    """Mode 2 speculates from a steady view only: during a drag (every call sees a camera change) no frame is queued to be dropped
    -- the context holds no deferred frame after such a call -- and the first steady calls trace ahead again.  The frames are the
    reference order's (mode 0) bit for bit, read with spcbpt_read_film, which waits for the shown frame's merge only."""
    shots = {}
    for mode in (0, 2):
        r, cam = _small_renderer(pkg)
        v = pkg.api.Viewer(r, cam["eye"], cam["lookat"], cam["up"], cam["fov"], 64, 64)
        if mode == 0:
            v.set_pipeline(0)
        out, deferred = [], []
        v.frame(); v.frame()
        deferred.append(r.pipeline_state()["deferred"])
        v.mouse_button("left", 1, 30, 30)
        for k in range(5):                                   # a drag in progress: one event per displayed frame
            v.cursor_pos(31 + 2 * k, 30 + k)
            v.frame()
            deferred.append(r.pipeline_state()["deferred"])
            out.append([x.copy() for x in r.read_film()])
        v.mouse_button("left", 0, 41, 35)
        for k in range(3):                                   # steady again
            v.frame()
            deferred.append(r.pipeline_state()["deferred"])
            out.append([x.copy() for x in r.read_film()])
        assert np.array_equal(out[-1][0], r.read_accum()) and np.array_equal(out[-1][1], r.read_frame())
        shots[mode] = (out, deferred)
        v.close()
    assert shots[0][1] == [0] * 9
    assert shots[2][1] == [1] + [0] * 5 + [1] * 3
    for k, (x, y) in enumerate(zip(shots[0][0], shots[2][0])):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]), k


@pytest.mark.gpu
def test_viewer_tolerates_a_host_that_consumes_what_it_queued(gpu, pkg):
    """Between two viewer frames the host drops the deferred frame and resets the light-ahead queue itself: the next viewer frame
    re-validates its flags against the context (spcbpt_get_pipeline_state) instead of failing in the middle."""
    r, cam = _small_renderer(pkg)
    v = pkg.api.Viewer(r, cam["eye"], cam["lookat"], cam["up"], cam["fov"], 64, 64)
    v.frame(); v.frame()
    r.merge_deferred(False)
    r.set_light_ahead(True)            # clears the queue of unbuilt passes
    v.frame(); v.frame()
    a = r.read_accum()
    assert np.isfinite(a).all() and a[..., :3].mean() > 0
    assert v.state()["subframe_index"] == 4
    v.close()


@pytest.mark.gpu
def test_deferred_frame_blocks_everything_that_would_reorder_the_film(gpu, pkg):
    """While a deferred frame is outstanding spcbpt_launch_eye_batch, spcbpt_clear_accum and spcbpt_set_light_ahead are refused
    like every other render launch (include/spcbpt.h); a light pass invalidates the sampler also in light-ahead mode, and
    spcbpt_reuse_sampler brings back the last build's tables while they are intact."""
    r, _ = _small_renderer(pkg)
    r.set_light_ahead(True)
    r.launch("light trace", 1); r.build_sampler()
    r.launch_deferred("SPCBPT_eye", 0)
    with pytest.raises(pkg.SpcbptError, match="deferred"):
        r.launch_eye_batch([1])
    with pytest.raises(pkg.SpcbptError, match="deferred"):
        r.clear_accum()
    with pytest.raises(pkg.SpcbptError, match="deferred"):
        r.set_light_ahead(False)
    r.merge_deferred(True)
    r.launch("light trace", 2)                       # a pass ahead: "build before you render"
    with pytest.raises(pkg.SpcbptError, match="built sampler"):
        r.launch("SPCBPT_eye", 1)
    assert r.pipeline_state()["sampler_intact"] == 1
    r.reuse_sampler()                                # ... unless the host says it means the last build's tables
    r.launch("SPCBPT_eye", 1)
    r.sync()
    want, _ = _small_renderer(pkg)
    want.launch("light trace", 1); want.build_sampler()
    want.launch("SPCBPT_eye", 0); want.launch("SPCBPT_eye", 1); want.sync()
    assert np.array_equal(r.read_accum(), want.read_accum())
    # once a pass has taken the set, the tables are gone
    sets = r.lvc_capacity()[1]
    for k in range(sets):
        r.launch("light trace", 3 + k)
    assert r.pipeline_state()["sampler_intact"] == 0
    with pytest.raises(pkg.SpcbptError, match="reuse_sampler"):
        r.reuse_sampler()
