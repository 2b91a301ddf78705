"""What one rank of an N-GPU job does per frame, on one GPU: 1/N of the light cores, every N-th 8-row band.  Shows how the
step time of a rank scales with N and with the number of frames in flight (SPCBPT_RENDER_STREAMS).
  python tools/rank_sim.py N streams [steps] [--exchange] [--ahead | --ahead=D] [--batch=F] [--lbatch] [--trained]
--lbatch: the light passes of a batch of F frames as one launch too (spcbpt_launch_light_batch), one batch ahead.
--native: the C++ host (libspcbpt_mgpu) at world size 1 -- every call of the real frame loop is there (pack, RCCL all-gather, compaction,
device-count sampler build), but the gather carries only this rank's own shard: what N - 1 peers add is the driver's to measure.
--xbatch (with --native --lbatch): ONE exchange per light batch (spcbpt_comm_exchange_lvc_batch) instead of one per frame.
--bbatch (with --lbatch, and --xbatch if --native): the sampler builds of a batch as one set of launches (spcbpt_build_sampler_batch).
--exchange runs the per-frame host sequence of the real job too (dist.FrameExchanger on a world-size-1 RCCL group: the
all-gathers degenerate to copies, but every host wait of the exchange path is there), which is what bounds a rank's frame
rate when its share of the image is small."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]); streams = int(args[1]); steps = int(args[2]) if len(args) > 2 else 32
exchange, trained, native = "--exchange" in sys.argv, "--trained" in sys.argv, "--native" in sys.argv   # --native: the C++ RCCL host (libspcbpt_mgpu) at world size 1
depth = max([int(a[len("--ahead="):]) for a in sys.argv if a.startswith("--ahead=")] + [1 if "--ahead" in sys.argv else 0])   # light passes ahead
ahead = depth > 0
batch = max([int(a[len("--batch="):]) for a in sys.argv if a.startswith("--batch=")] + [1])   # frames per eye launch
if batch > 1: os.environ["SPCBPT_EYE_BATCH"] = str(batch)
os.environ["SPCBPT_RENDER_STREAMS"] = str(streams)
import __graft_entry__ as g
p = g.load_package()
scene = p.scenes.bedroom()
W, H, M = 1920, 1080, 100000
r = p.Renderer(scene, 0)
c = scene.camera
r.set_camera_lookat(c["eye"], c["lookat"], c["up"], c["fov"], W / H)
r.resize(W, H)
r.set_light_trace(M, 52, 1)
if trained: r.preprocess(2_000_000, 2_000_000, True)
else: r.set_subspace()
r.set_light_trace(M, 52, 1, core_begin=0, core_count=M // N)
ex = None
if exchange:
    import torch, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev, pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True))
    ex = p.dist.FrameExchanger(r, 0, 1, dev)
throttle = "--throttle" in sys.argv   # host waits for the light pass it is about to exchange (what the python exchange does implicitly)
comm = None
if native:
    comm = p.dist.Comm(r, 0, 1, p.dist.unique_id())
    print("shard capacity after calibration:", comm.calibrate(passes=2, slack=1.5))
lbatch = "--lbatch" in sys.argv and batch > 1
xbatch = "--xbatch" in sys.argv and lbatch and comm is not None
bbatch = "--bbatch" in sys.argv and lbatch and (comm is None or xbatch) and ex is None
rows = (0, H, N)
if lbatch:
    r.set_light_ahead(True)
    r.launch_light_batch(1, batch)
    depth = batch
elif ahead:
    r.set_light_ahead(True)
    for k in range(depth): r.launch("light trace", 1 + k)
queued = []
def eye(f):
    if batch == 1: r.launch("SPCBPT_eye", f, rows); return
    queued.append(f)
    if len(queued) == batch:
        if bbatch: r.build_sampler_batch(batch)
        r.launch_eye_batch(queued, rows); queued.clear()
def step(f):
    if lbatch:
        if f % batch == 0:
            r.launch_light_batch(f + 1 + batch, batch)
            if xbatch: comm.exchange_lvc_batch(batch)
    else: r.launch("light trace", f + 1 + depth)
    if ex is not None: ex.allgather_lvc()
    if throttle: r.sync_light()
    if comm is not None and not xbatch: comm.exchange_lvc()
    if not bbatch: r.build_sampler()
    eye(f)
for f in range(batch * max(1, 8 // batch)): step(f)   # a multiple of the batch: nothing is left queued when the clock starts
r.sync()
t0 = time.perf_counter()
host = {"light": 0.0, "exchange": 0.0, "build": 0.0, "eye": 0.0}
def timed_step(f):
    a = time.perf_counter()
    if lbatch:
        if f % batch == 0: r.launch_light_batch(f + 1 + batch, batch)
    else: r.launch("light trace", f + 1 + depth)
    b = time.perf_counter()
    if ex is not None: ex.allgather_lvc()
    if throttle: r.sync_light()
    if xbatch:
        if f % batch == 0: comm.exchange_lvc_batch(batch)
    elif comm is not None: comm.exchange_lvc()
    c2 = time.perf_counter()
    if not bbatch: r.build_sampler()
    d = time.perf_counter(); eye(f)
    e = time.perf_counter()
    host["light"] += b - a; host["exchange"] += c2 - b; host["build"] += d - c2; host["eye"] += e - d
for f in range(steps): timed_step(f)
if queued:                       # a last, shorter eye launch: every frame counted is rendered
    if bbatch: r.build_sampler_batch(len(queued))
    r.launch_eye_batch(queued, rows); queued.clear()
t_host = time.perf_counter() - t0
r.sync()
dt = (time.perf_counter() - t0) / steps
print("host time per frame (ms): " + ", ".join(f"{k} {v / steps * 1e3:.3f}" for k, v in host.items()) + f"; host busy {t_host / steps * 1e3:.3f} of {dt * 1e3:.3f} ms")
print(f"N={N} streams={streams}{' exchange' if exchange else ''}{' native-exchange' + (' (one per light batch)' if xbatch else '') if native else ''}{' ahead=' + str(depth) if ahead else ''}{' batch=' + str(batch) if batch > 1 else ''}{' lbatch' if lbatch else ''}{' bbatch' if bbatch else ''}{' trained' if trained else ''}: {dt * 1e3:.3f} ms per rank-frame  -> {N}-GPU job at {(W * H + M) / dt / 1e6:.1f} Mpaths/s if the exchange were free")
if ex is not None:
    import torch.distributed as dist
    dist.destroy_process_group()
