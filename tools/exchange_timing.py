import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["SPCBPT_RENDER_STREAMS"] = "4"
import __graft_entry__ as g
p = g.load_package()
import torch, torch.distributed as dist
scene = p.scenes.bedroom()
W, H, M, N = 1920, 1080, 100000, 8
r = p.Renderer(scene, 0)
c = scene.camera
r.set_camera_lookat(c["eye"], c["lookat"], c["up"], c["fov"], W / H); r.resize(W, H)
r.set_light_trace(M, 52, 1); r.set_subspace()
r.set_light_trace(M, 52, 1, core_begin=0, core_count=M // N)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dev = torch.device("cuda", 0)
opts = None
if os.environ.get("NCCL_HIGH_PRIO", "1") == "1":
    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev, pg_options=opts)
ex = p.dist.FrameExchanger(r, 0, 1, dev)
T = {}
def tick(name, t0):
    t = time.perf_counter(); T[name] = T.get(name, 0) + t - t0; return t
VB = p.dist.VERTEX_BYTES
def allgather(self):
    t = time.perf_counter()
    dv, dc, cap = r.lvc_export(); t = tick("export", t)
    r.sync_light(); t = tick("sync_light", t)
    my_count = p.dist.device_view(dc, 8, dev).view(torch.int32)[:1]
    dist.all_gather_into_tensor(self.counts, my_count.clone()); t = tick("gather counts", t)
    counts = self.counts.cpu().tolist(); t = tick("counts.cpu", t)
    mx = max(counts); nbytes = mx * VB
    if self.gather_buf is None or self.gather_buf.numel() < self.world * nbytes:
        self.gather_buf = torch.empty(self.world * max(nbytes, VB), dtype=torch.uint8, device=dev)
    shard = p.dist.device_view(dv, cap * VB, dev)[:nbytes]
    out = self.gather_buf[: self.world * nbytes]
    dist.all_gather_into_tensor(out, shard); t = tick("gather shards", t)
    total = sum(counts); k2 = self.frame & 1; self.frame += 1
    r.lvc_import_wait()
    if self.cat_bufs[k2] is None or self.cat_bufs[k2].numel() < total * VB:
        self.cat_bufs[k2] = torch.empty(max(total, 1) * VB * 5 // 4, dtype=torch.uint8, device=dev)
    cat = self.cat_bufs[k2]; off = 0
    for k, cnt in enumerate(counts):
        cat[off: off + cnt * VB].copy_(out[k * nbytes: k * nbytes + cnt * VB]); off += cnt * VB
    t = tick("concat", t)
    torch.cuda.current_stream(dev).synchronize(); t = tick("torch sync", t)
    r.lvc_import_device(cat.data_ptr(), total); t = tick("import", t)
r.set_light_ahead(True); r.launch("light trace", 1)
rows = (0, H, N)
def step(f):
    r.launch("light trace", f + 2); allgather(ex); r.build_sampler(); r.launch("SPCBPT_eye", f, rows)
for f in range(6): step(f)
r.sync(); T.clear()
t0 = time.perf_counter(); n = 64
for f in range(n): step(f)
r.sync(); dt = time.perf_counter() - t0
print({k: round(v / n * 1e3, 3) for k, v in T.items()}, "frame ms", round(dt / n * 1e3, 3))
dist.destroy_process_group()
