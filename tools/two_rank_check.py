"""Developer check at the bench scale: two contexts on one GPU run the sharded host loop of bench.py (light passes a batch ahead on two
lanes, export / device gather / import without host waits, sampler build, 4 frames per eye launch on interleaved bands); the sum
of their films must equal the frames one context renders the plain way, bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g
p = g.load_package()
dev = torch.device("cuda", 0)
scene = p.scenes.bedroom()
W, H, M, NF, WORLD, BATCH, DEPTH = 1920, 1080, 100000, 8, 2, 4, 4
VB = p.dist.VERTEX_BYTES
c = scene.camera

def make(batch):
    if batch > 1: os.environ["SPCBPT_EYE_BATCH"] = str(batch)
    else: os.environ.pop("SPCBPT_EYE_BATCH", None)
    r = p.Renderer(scene, 0)
    r.set_camera_lookat(c["eye"], c["lookat"], c["up"], c["fov"], W / H); r.resize(W, H)
    r.set_light_trace(M, 52, 1)
    return r

single = make(1)
single.preprocess(2000000, 2000000, True)
tup = single.get_subspace()
for f in range(NF):
    single.launch("light trace", f + 1); single.build_sampler(); single.launch("SPCBPT_eye", f)
single.sync()
want = single.read_accum().copy()
single.close()

ranks = []
for k in range(WORLD):
    r = make(BATCH)
    r.set_subspace(*tup)
    b, n = p.dist.core_range(M, k, WORLD)
    r.set_light_trace(M, 52, 1, core_begin=b, core_count=n)
    r.set_light_ahead(True)
    for d in range(DEPTH): r.launch("light trace", 1 + d)
    ranks.append(r)
stage = [[None, None] for _ in range(WORLD)]
queued = []
for f in range(NF):
    shards = []
    for r in ranks:
        r.launch("light trace", f + 1 + DEPTH)
        dv, dc, cap = r.lvc_export()
        r.sync_light()
        n = int(p.dist.device_view(dc, 8, dev).view(torch.int32)[0].item())
        shards.append(p.dist.device_view(dv, n * VB, dev))
    gathered = torch.cat(shards)
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
            r.launch_eye_batch(queued, p.dist.band_rows(H, k, WORLD))
        queued = []
for r in ranks: r.sync()
films = [r.read_accum() for r in ranks]
got = films[0] + films[1]
d = np.abs(got.astype(np.float64) - want)
print("two-rank sharded loop vs single context: equal =", np.array_equal(got, want), "differing pixels =", int((d.max(axis=2) > 0).sum()), "max abs =", d.max())
