#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s of the SPCBPT hot path on the bedroom-class scene at 1920x1080 (BASELINE.json metric;
configs[2] at N=1, configs[3] at N>1).

One "step" = one subframe of the reference's render loop (optixPathTracer.cpp:791-822): light trace (M = 100 000 light
paths) -> device sampler build -> [N>1: RCCL all-gather of the LVC shards] -> SPCBPT eye megakernel over the image
(N>1: every N-th band of 8 rows per rank); N>1 ends the timed region with the RCCL film exchange (band all-gather).
value = (eye paths + light paths) of all ranks / wall time.  Inputs are resident in HBM before the timed region.
The K steps of a run are issued as few, equal launches: the light passes of up to 32 frames in one thin persistent grid a batch
ahead, then per frame its exchange and sampler build, then ONE persistent eye launch over those frames -- every
step still traces its own light pass, builds its own sampler and renders its own subframe inside the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W]      N > 1 started the plain way: this process only spawns the N ranks
                                                            (python -m torch.distributed.run ... as a child, before any GPU call) and relays rank 0's line
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_scene(pkg, name, tris):
    if name == "bedroom":
        return pkg.scenes.bedroom(target_tris=tris)
    if name == "cornell":
        return pkg.scenes.cornell_box()
    if name == "hallway":
        return pkg.scenes.hallway(target_tris=tris)
    raise SystemExit(f"unknown scene {name}")


def light_geometry(args):
    """(num_core, core_padding, m_per_core) of LightTraceParams"""
    if args.light_geometry == "reference":
        return (1000, 800, 100)      # lt_params_setup, optixPathTracer.cpp:464-467
    return (args.light_paths, 52, 1)


def spawn_command(n: int, argv, port: int, script: str = None):
    """The child that runs N ranks of this script: one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int, argv, script: str = None, out=None) -> int:
    """`python bench.py --gpus N` without a launcher: start the ranks as FRESH child processes and relay rank 0's JSON line.
    Runs before torch is imported or the GPU is touched (the parent never initialises HIP; nothing is exec'ed).  Returns the exit
    code: the children's, or 1 when they ended well without printing a result line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = spawn_command(n, argv, free_port(), script)
    print("bench.py: starting", n, "ranks:", " ".join(cmd), file=sys.stderr)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for raw in proc.stdout:        # rank 0 prints exactly one JSON line on stdout; anything else is passed on to stderr
        t = raw.strip()
        if t.startswith("{") and '"metric"' in t:
            try:
                json.loads(t)
                line = t
                continue
            except ValueError:
                pass
        sys.stderr.write(raw)
    rc = proc.wait()
    if rc == 0 and line is None:
        print("bench.py: the ranks ended without a result line", file=sys.stderr)
        rc = 1
    if line is not None and rc == 0:
        (out or sys.stdout).write(line + "\n")
        (out or sys.stdout).flush()
    return rc


def host_cpus(cgroup_root="/sys/fs/cgroup"):
    """The CPUs this process can really run on: the smallest of os.cpu_count(), the scheduler affinity mask and the cgroup CPU quota
    (v2 `cpu.max`, v1 `cpu.cfs_quota_us`).  A one-GPU box of the pool shows all 256 hardware threads of its host in os.cpu_count() but
    is given a 16-CPU share of them: 256 oracle threads then run like 11, and a baseline labelled "256 cores" misleads (round-5 review)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open(os.path.join(cgroup_root, "cpu.max")).read().split()[:2]
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open(os.path.join(cgroup_root, "cpu", "cpu.cfs_quota_us")).read())
            per = int(open(os.path.join(cgroup_root, "cpu", "cpu.cfs_period_us")).read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.999)))
    return max(1, n)


def cpu_baseline(pkg, scene, args, tup):
    """The oracle (scalar CPU restatement, `kind: port`) timed on this host's cores on a bounded sample of the same
    workload: the full light pass + sampler build + every `stride`-th band of the eye pass."""
    from oracle import binding as ob
    threads = args.cpu_threads if args.cpu_threads > 0 else host_cpus()
    o = ob.Oracle(scene, nthreads=threads)
    cam = scene.camera
    o.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], args.width / args.height)
    o.resize(args.width, args.height)
    o.set_light_trace(*light_geometry(args))
    o.set_subspace(*tup)
    o.enable_counters(False)
    o.set_skip_null_connections(True)   # the same work as the product; the image does not depend on it
    stride = args.cpu_band_stride if args.cpu_band_stride > 0 else max(1, 48 // threads)
    rows = sum(1 for y in range(args.height) if (y // 8) % stride == 0)
    frames, dt = 0, 0.0
    t0 = time.perf_counter()
    while frames < 8 and dt < 10.0:   # whole subframes until ~10 s of wall time are covered (256-thread hosts need a few)
        o.launch("light trace", frames + 1)
        o.build_sampler()
        o.launch("SPCBPT_eye", frames, rows=(0, args.height, stride))
        frames += 1
        dt = time.perf_counter() - t0
    paths = frames * (rows * args.width + args.light_paths)
    out = {"value": paths / dt / 1e6, "unit": "Mpaths/s", "cores": threads, "host_cpus_visible": os.cpu_count(), "kind": "port",
           "sample": f"{frames} subframe(s): {args.light_paths} light paths + sampler build + eye pass on every {stride}-th 8-row "
                     f"band ({rows * args.width} eye paths) each, {dt:.1f} s with {threads} threads"}
    # (i) of SURVEY 8(d): the same port on ONE thread.  A whole subframe would take minutes, so its two halves are sampled and put
    # together: the eye pass on every s1-th band over the sampler the run above left (~5 s), then 1/16 of the light pass's cores
    # with their sampler build (~5 s); value = paths of one subframe / (eye time scaled to all bands + light time x 16)
    if threads > 1:
        bands = (args.height + 7) // 8
        s1 = max(stride, bands // 2)             # two bands or so
        rows1 = sum(1 for y in range(args.height) if (y // 8) % s1 == 0)
        t0 = time.perf_counter()
        o.launch("SPCBPT_eye", frames, rows=(0, args.height, s1), nthreads=1)
        t_eye = time.perf_counter() - t0
        nc, pad, mpc = light_geometry(args)
        frac = 16
        o.set_light_trace(max(1, nc // frac), pad, mpc)
        t0 = time.perf_counter()
        o.launch("light trace", frames + 1, nthreads=1)
        o.build_sampler()
        t_light = time.perf_counter() - t0
        t_frame = t_eye * (args.height / rows1) + t_light * frac
        out["single_thread"] = {"value": (args.width * args.height + args.light_paths) / t_frame / 1e6, "unit": "Mpaths/s", "cores": 1, "kind": "port",
                                "sample": f"eye pass on every {s1}-th band ({rows1 * args.width} eye paths, {t_eye:.1f} s) + 1/{frac} of the light pass's cores with "
                                          f"their sampler build ({t_light:.1f} s), scaled to one whole subframe ({t_frame:.0f} s)"}
        out["speedup_over_single_thread"] = round(out["value"] / out["single_thread"]["value"], 2)
    return out


def fast_math_line(pkg, args):
    """A SECOND line, never `value`: the same command on the opt-in approximate-arithmetic library (libspcbpt_hip_fast.so: hardware
    reciprocal / square root like the reference's own --use_fast_math build; image-level bars of its own, tests/test_gpu_fast_build.py),
    in a child interpreter (two libraries with the same exports cannot share a process), after this process's measurements are done."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline",
           "--fast-math-line", "0", "--sync-each-frames", "0", "--long-steps", "0", "--scene", args.scene, "--tris", str(args.tris),
           "--width", str(args.width), "--height", str(args.height), "--light-paths", str(args.light_paths), "--tuple", args.tuple]
    env = dict(os.environ, SPCBPT_LIB=pkg.api.FAST_LIB_PATH)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        d = json.loads(p.stdout.strip().splitlines()[-1])
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "kernel_ms": d["roofline"]["kernel_ms"],
                "library": "libspcbpt_hip_fast.so", "arithmetic": "approx (v_rcp_f32 / v_sqrt_f32: -fno-hip-fp32-correctly-rounded-divide-sqrt)",
                "note": "opt-in build, NOT the shipped default and not `value`: function-level parity and the film hashes are the IEEE build's"}
    except Exception as e:   # the second line is optional: never let it take the contract's line down
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def frames_per_launch(steps: int, max_batch: int = 32) -> int:
    """Frames per eye / light launch for a run of `steps` steps: as few launches as `max_batch` allows, all of (nearly) the same size."""
    launches = max(1, -(-steps // max_batch))
    return max(1, min(max_batch, -(-steps // launches)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs of this node (default: WORLD_SIZE when launched by torch.distributed.run, else 1)")
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="bedroom")
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--light-paths", type=int, default=100_000)
    ap.add_argument("--tuple", default="trained", choices=["minimal", "trained"],
                    help="trained (default): the reference's operating mode -- preprocessing() always runs before the first frame "
                         "(optixPathTracer.cpp:763-766): pretrace, subspace trees of up to 1000 leaves, Q, Adam-trained Gamma; "
                         "minimal: single-leaf trees, Gamma rows = Q (the cheapest valid tuple, no classification work)")
    ap.add_argument("--scene-route", default="gltf", choices=["gltf", "memory"],
                    help="gltf: write the generated scene as glTF 2.0 and read it back with the C++ reader (default); memory: hand the arrays over directly")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--long-steps", type=int, default=256, help="steps of the extra steady-state run reported as ms_per_step_long (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the cpu_baseline leg (0 = the CPUs this process can really use: host_cpus())")
    ap.add_argument("--fast-math-line", type=int, default=1, help="1 (default, one GPU): report the same run on the opt-in approximate-arithmetic library as a second line (fast_math_build)")
    ap.add_argument("--sync-each-frames", type=int, default=16, help="frames of the extra pass in the reference's loop form -- one light pass, one build, one eye "
                    "launch and a device sync per frame (optixPathTracer.cpp:791-822) -- reported as ms_per_frame_sync_each (0 = skip)")
    ap.add_argument("--cpu-band-stride", type=int, default=0, help="0 = choose from the host core count (about 10-30 s of CPU work)")
    ap.add_argument("--render-streams", type=int, default=0,
                    help="render streams per GPU, each with one (batched) eye launch in flight (0 = 1)")
    ap.add_argument("--light-geometry", default="lane", choices=["lane", "reference"],
                    help="lane (default): one light path per core, M cores of 52 slots, BSDF stream decorrelated (DESIGN.md d1); "
                         "reference: the reference's launch geometry lt_params_setup (optixPathTracer.cpp:462-477): 1000 cores x 100 paths, "
                         "800 slots per core, both random streams of a core start equal (q4)")
    ap.add_argument("--write-image", default="")
    ap.add_argument("--eye-batch", type=int, default=0,
                    help="frames per eye launch (spcbpt_launch_eye_batch); 0 = up to 32, equal launches: several frames in one tile queue pay the "
                         "drain phase of the persistent kernel once (and a rank's share of a sharded frame is about one tile per resident wave: "
                         "all drain); 1 = one launch per frame")
    ap.add_argument("--light-batch", type=int, default=-1,
                    help="1: the light passes of a batch of frames as ONE persistent launch too (spcbpt_launch_light_batch), a batch ahead; "
                         "0: one launch per pass (a pass is a ~1.2 ms dependent chain however few paths a rank traces; a batch of them in one thin, long-lived grid costs the eye kernels beside it less)")
    ap.add_argument("--build-ahead", type=int, default=1, help="1 (default, one GPU): the sampler builds of the next eye launch are queued behind this one (they run under it); 0: built when their eye launch is issued")
    ap.add_argument("--build-batch", type=int, default=1, help="1 (default): the sampler builds of a batch of frames as one set of four launches (spcbpt_build_sampler_batch); 0: one build per step")
    ap.add_argument("--light-ahead", type=int, default=0, help="light passes launched ahead of their sampler build (0 = 1, or the batch size when eye launches are batched)")
    ap.add_argument("--no-light-ahead", action="store_true", help="launch each frame's light pass only when its sampler build / exchange is due (the host then waits for it)")
    ap.add_argument("--exchange-batch", type=int, default=1, help="1 (default): ONE all-gather per light batch (spcbpt_comm_exchange_lvc_batch); 0: one exchange per frame")
    ap.add_argument("--force-exchange", action="store_true", help="run the RCCL exchange path even at world size 1 (self-test)")
    ap.add_argument("--exchange", default="native", choices=["native", "python"],
                    help="native (default): the C++ RCCL host (libspcbpt_mgpu.so: all-gather of fixed-capacity shards + device counts, "
                         "device-side compaction, no host wait per frame, film band gather); python: the torch.distributed harness of "
                         "dist.FrameExchanger (per-frame host syncs; kept for comparison)")
    args = ap.parse_args()

    gpus_given = args.gpus is not None
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if gpus_given and "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        # a launcher (or an outer harness that exported WORLD_SIZE) disagrees with the request: running on WORLD_SIZE GPUs and
        # labelling the line --gpus would be a wrong number, and so would the reverse
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']} in the environment: launch with "
                         f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}` or unset WORLD_SIZE")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started the plain way: the ranks are fresh children (one per GPU); this process never touches the GPU
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    # exactly ONE line on stdout: native libraries (RCCL prints a version banner) write to fd 1 too, so fd 1 is pointed
    # at stderr for the run and the JSON line goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus

    import torch
    pkg = entry.load_package()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_exchange:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        # the two small all-gathers of a frame queue behind persistent eye kernels for block slots: high-priority RCCL stream
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device, pg_options=opts)

    # One render stream, 16 frames per persistent eye launch (and per light launch): the kernel-end drain -- the last paths of a
    # launch, up to 50 bounces with ever fewer live lanes -- is paid once per launch, and with the light passes in a thin grid of
    # their own nothing else needs the second stream any more.  Measured on one GPU at the driver's 20 steps / in a 64-step run, ms
    # per step: 2 streams x 4 frames 6.19 / 6.05, 1 x 8 6.00 / 6.01, 1 x 16 6.02 / 5.86 (kernel 5.98 -> 5.76 -> 5.62 ms per frame);
    # a rank's share of a sharded frame gains more (N = 8 simulation: 0.94 -> 0.85 ms per rank-frame from 8 to 16 frames).
    # Up to 32 frames per launch, and a run's launches are kept equal (40 steps = 2 x 20, not 32 + 8: a short last launch pays a whole
    # drain for a few frames).  Per frame the kernel takes 5.98 ms in launches of 4 frames, 5.76 of 8, 5.63 of 16, 5.59 of 20-32.
    batch = args.eye_batch if args.eye_batch > 0 else frames_per_launch(args.steps)
    streams = args.render_streams if args.render_streams > 0 else 1
    os.environ["SPCBPT_RENDER_STREAMS"] = str(streams)   # read by spcbpt_create
    os.environ["SPCBPT_EYE_BATCH"] = str(batch)          # sizes the ring of sampler buffer sets
    scene = make_scene(pkg, args.scene, args.tris)
    if args.scene_route == "gltf":
        # BASELINE config 2: "bedroom-class glTF scene" -- the generated scene goes to disk as glTF 2.0 (binary buffers + PPM
        # textures) and comes back through the library's C++ glTF reader (spcbpt_gltf_load), every rank for itself
        import tempfile
        with tempfile.TemporaryDirectory(prefix="spcbpt_bench_") as tmp:
            scene, warn = pkg.load_gltf(pkg.scenes.write_gltf(scene, tmp, args.scene))
            if warn: print("glTF warnings:", warn, file=sys.stderr)
    r = pkg.Renderer(scene, local_rank)
    cam = scene.camera
    r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], args.width / args.height)
    r.resize(args.width, args.height)
    if args.light_geometry == "reference":
        args.light_paths = 100_000   # 1000 cores x 100 paths
    M = args.light_paths
    ncore, pad, mpc = light_geometry(args)
    # the subspace tuple is computed with the full light pass, then the pass is sharded.  N > 1: rank 0 trains and
    # broadcasts (start-up, outside the timed region), so that every rank labels and weights with the same tuple
    r.set_light_trace(ncore, pad, mpc)
    t_pre = time.perf_counter()
    if rank == 0 or dist is None:
        if args.tuple == "trained":
            r.preprocess(target_paths=2_000_000, target_q_paths=2_000_000, train=True)
        else:
            r.set_subspace()
    tup = r.get_subspace() if (rank == 0 or dist is None) else None
    begin, count = pkg.dist.core_range(ncore, rank, world)
    ex = comm = None
    native_error = None
    if dist is not None and args.exchange != "python":
        # the C++ N-GPU host: its own RCCL communicator per rank (the unique id travels over torch.distributed, which otherwise only
        # provides the barrier and the max-over-ranks of the timing contract)
        try:
            r.set_light_trace(ncore, pad, mpc, core_begin=begin, core_count=count)
            uid = torch.tensor(list(pkg.dist.unique_id() if rank == 0 else bytes(pkg.dist.UNIQUE_ID_BYTES)), dtype=torch.uint8, device=device)
            dist.broadcast(uid, 0)
            comm = pkg.dist.Comm(r, rank, world, bytes(uid.cpu().tolist()))
            comm.broadcast_subspace(0)             # rank 0 trained; ncclBroadcast of trees, Q, Gamma
            comm.calibrate(passes=2, slack=1.5)    # shard capacity of exchange 1 from two light passes (host waits: start-up only)
        except Exception as e:                     # noqa: BLE001 -- reported below, by every rank that saw it
            native_error = e
        # no silent fallback: a scaling number from the torch harness must not pass for the native host's.  If the C++ host could
        # not be set up on ANY rank, every rank ends non-zero (--exchange python selects the harness on purpose)
        flag = torch.tensor([0 if native_error is not None else 1], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            print(f"rank {rank}: C++ N-GPU host (libspcbpt_mgpu) failed: {native_error!r}", file=sys.stderr)
            dist.destroy_process_group()
            raise SystemExit(3)
        comm_rank, comm_world, comm_transport = comm.info()
        if comm_world != world or comm_rank != rank:
            raise SystemExit(f"rank {rank}: RCCL reports rank {comm_rank} of {comm_world}, the launcher said {rank} of {world}")
    if dist is not None and args.exchange == "python":
        tup = pkg.dist.broadcast_subspace(tup, 0, device)
        if rank != 0:
            r.set_subspace(*tup)
        r.set_light_trace(ncore, pad, mpc, core_begin=begin, core_count=count)
        ex = pkg.dist.FrameExchanger(r, rank, world, device)
    elif dist is None:
        r.set_light_trace(ncore, pad, mpc, core_begin=begin, core_count=count)
    t_pre = time.perf_counter() - t_pre
    rows = pkg.dist.band_rows(args.height, rank, world)
    info = r.scene_info()

    # The light pass of the NEXT frame is launched before this frame's shards are gathered and its sampler is built, so the
    # host never waits for a light pass that has only just been queued (spcbpt_set_light_ahead; a light pass is a ~1 ms
    # dependent chain however few paths a rank traces, and it shares the GPU with the previous frame's eye kernel).  Every step still launches exactly one light pass, one exchange, one sampler build
    # and one eye pass; the light pass a step launches is consumed by the next step.
    ahead = not args.no_light_ahead
    depth = args.light_ahead if args.light_ahead > 0 else (batch if batch > 1 else 1)
    state = {"next_light": 1, "primed": False, "lb_left": 0, "phase_left": 0}
    lbatch = ahead and batch > 1 and args.light_batch != 0 and ex is None   # (the torch harness keeps its tested one-pass-per-launch loop)
    xbatch = lbatch and args.exchange_batch != 0     # one LVC exchange per light batch (spcbpt_comm_exchange_lvc_batch)
    if ahead:
        r.set_light_ahead(True)

    def light_batch(n):
        r.launch_light_batch(state["next_light"], n); state["next_light"] += n

    def step(f, isolate=False):
        if not ahead:
            r.launch("light trace", f + 1)
        elif lbatch:                           # every `batch` steps: the passes of the batch after the one being built, as one launch
            if not state["primed"]:
                light_batch(batch); state["primed"] = True    # the stock every phase draws from and refills: one batch
            if state["lb_left"] == 0:                          # a phase of P steps launches exactly P passes (and consumes P of the stock)
                n = max(1, min(batch, state["phase_left"]))
                light_batch(n); state["lb_left"] = n
                if comm is not None and xbatch:
                    # ... and exchanges the n OLDEST pending passes (launched a batch ago) as ONE all-gather + one compaction --
                    # unless that exchange was queued a batch ahead, behind the previous eye launch (build_ahead below)
                    if state["exchanged"] == n:
                        state["exchanged"] = 0
                    elif state["exchanged"] == 0:
                        comm.exchange_lvc_batch(n)
                    else:
                        raise SystemExit(f"bench loop: {state['exchanged']} passes were exchanged ahead for a batch of {n}")
            state["lb_left"] -= 1; state["phase_left"] -= 1
        else:
            if not state["primed"]:
                for _ in range(depth):
                    r.launch("light trace", state["next_light"]); state["next_light"] += 1
                state["primed"] = True
            r.launch("light trace", state["next_light"]); state["next_light"] += 1   # consumed by the next step
        if ex is not None:
            ex.allgather_lvc()
        elif comm is not None:
            # back-pressure, not a data dependency: without it the host queues dozens of frames of light passes and builds ahead of
            # the eye kernels and a rank-frame takes 1.7 instead of 1.4 ms (N = 8 share, rank_sim); the pass waited for was launched
            # `depth` steps ago, so the wait is normally over before it starts
            if not lbatch:                     # (batched passes pace themselves: a batch waits for the eye launch that last read its sets)
                r.sync_light()
            if not (lbatch and xbatch):
                comm.exchange_lvc()            # queues the all-gather + compaction on the communicator's stream; no host wait
        if not bbatch:
            r.build_sampler()
        if isolate and not bbatch and (batch == 1 or len(queued) == batch - 1):
            r.sync()                           # roofline pass: the light pass launched above must not share the GPU with the eye kernel
        if batch == 1:
            r.launch("SPCBPT_eye", f, rows)
        else:                                  # one persistent eye kernel per `batch` frames (tile queue spans the frames)
            queued.append(f)
            if len(queued) == batch:
                flush(isolate)

    queued = []
    # the sampler builds of a batch of frames as ONE set of four launches (spcbpt_build_sampler_batch), issued with the eye launch that
    # needs them: every step still has its build, but 20 builds in a row are 80 small dependent launches in front of a kernel that
    # cannot start before the last (2.4 ms of the 86 a 20-step run takes)
    bbatch = lbatch and args.build_batch != 0 and (comm is None or xbatch)   # (a per-frame exchange addresses the oldest UNBUILT pass: its build cannot wait)

    # ... and a batch AHEAD (one GPU): the builds of the NEXT eye launch are queued right behind this one -- their light passes were
    # launched at this batch's first step, a batch ago by the time they are needed -- so that they run under the eye kernel that does
    # not need them instead of between two eye kernels.  Every step still has its one build (of a later frame, like its light pass);
    # spcbpt_launch_eye_batch renders the samplers of the last n builds, in build order, so the frames keep their passes.
    # A sharded job does the same with its exchange: the all-gather of the next launch's light passes and the builds over the gathered
    # caches are queued behind this eye launch (every rank runs this same sequence of calls, so the collectives keep their order).
    build_ahead = bbatch and ex is None and (comm is None or xbatch) and hasattr(r.lib, "spcbpt_get_pipeline_state") and args.build_ahead != 0
    state["prebuilt"] = 0
    state["exchanged"] = 0

    def build_next(n):
        if comm is not None:
            comm.exchange_lvc_batch(n)
            state["exchanged"] = n
        r.build_sampler_batch(n)
        state["prebuilt"] = n

    def flush(isolate=False, ahead_at_end=False):
        if queued:
            need = len(queued)
            if bbatch:
                have = state["prebuilt"]
                if have not in (0, need):
                    raise SystemExit(f"bench loop: {have} samplers were built ahead for a batch of {need}")
                if have == 0:
                    r.build_sampler_batch(need)
                state["prebuilt"] = 0
                if isolate:
                    r.sync()
            r.launch_eye_batch(queued, rows)
            queued.clear()
            if build_ahead and not isolate and (state["phase_left"] > 0 or ahead_at_end):
                # (at the end of the timed phases too: the builds of the batch AFTER the phase -- a timed step has its build like its
                # light pass, whether or not anything renders it; at the end of the warm-up and isolation phases nothing is built ahead:
                # single launches follow, which would get between these samplers and their eye launch)
                pend = r.pipeline_state()["pending_passes"]
                n_next = min(batch, state["phase_left"]) if state["phase_left"] > 0 else min(batch, pend)
                if 0 < n_next <= pend:
                    build_next(n_next)

    def prebuild_first_batch(steps):
        """The samplers of a phase's first eye launch, built under the phase before it (its light passes were traced there too)."""
        if build_ahead and state["prebuilt"] == 0 and not queued:
            n = min(batch, steps)
            if 0 < n <= r.pipeline_state()["pending_passes"]:
                build_next(n)

    def barrier():
        if dist is not None:
            dist.barrier()
        r.sync()
        torch.cuda.synchronize()

    r.clear_accum()
    state["phase_left"] = args.warmup
    for f in range(args.warmup):
        step(f)
    flush()
    # event counts of ONE launch of the dominant kernel -> algorithmic bytes per launch (untimed)
    r.sync()
    if not ahead:
        r.launch("light trace", 999)
    else:
        r.launch("light trace", state["next_light"]); state["next_light"] += 1
    if ex is not None:
        ex.allgather_lvc()
    elif comm is not None:
        comm.exchange_lvc()
    r.build_sampler()
    # ... counted twice: (1) in the REFERENCE's order (two relabels per connection, one per RMIS update, a bisection per first
    # stage: the contract's table of SURVEY 8(d), kept as roofline.contract_reference_order), and (2) by the instantiation the
    # timed runs use (labels cached per vertex, guided resampling, Gamma / Q from its table): the events that EXECUTE -- roofline.frac is computed from (2)
    r.enable_counters(1)
    r.reset_counters()
    r.launch("SPCBPT_eye", 999, rows)
    r.sync()
    c_ref = r.counters()
    r.enable_counters(2)
    r.reset_counters()
    r.launch("SPCBPT_eye", 999, rows)
    r.sync()
    c_eye = r.counters()
    ph = r.phase_clocks()
    r.enable_counters(False)
    bytes_per_launch = pkg.algorithmic_bytes(c_eye)
    bytes_ref_per_launch = pkg.algorithmic_bytes(c_ref)
    bytes_actual_per_launch = pkg.algorithmic_bytes(c_eye, pkg.api.BYTES_ACTUAL)
    lane_util = {"node_step": round(ph["node_lanes"] / max(ph["node_slots"], 1), 4), "triangle_step": round(ph["tri_lanes"] / max(ph["tri_slots"], 1), 4)}

    # duration of the dominant kernel by itself (roofline): a few frames with a sync after each, so that no neighbouring
    # frame's kernel shares the GPU with it; HIP events on the kernel's own stream (spcbpt_kernel_time)
    r.enable_kernel_timing(True)
    r.reset_kernel_time()
    n_iso = 8 if batch == 1 else 2 * batch   # whole launches only: every timed launch holds `batch` frames
    state["phase_left"] = n_iso
    for f in range(n_iso):
        step(1000 + f, isolate=True)
        if batch == 1 or not queued:      # batched: a sync after each launch = after every `batch` steps
            r.sync()
    flush()
    r.sync()
    k_ms, k_n = r.kernel_time("spcbpt_render")
    bytes_per_launch *= batch             # a batched launch renders `batch` frames (the last one of this pass may hold fewer)
    bytes_ref_per_launch *= batch
    bytes_actual_per_launch *= batch
    lt_ms, _ = r.kernel_time("light_trace")
    sb_ms, _ = r.kernel_time("sampler_build")
    cp_ms, _ = r.kernel_time("lvc_compact")

    r.clear_accum()
    r.reset_kernel_time()
    prebuild_first_batch(args.steps)
    barrier()
    t0 = time.perf_counter()
    state["phase_left"] = args.steps
    for f in range(args.steps):
        step(f)
    flush(ahead_at_end=True)
    if ex is not None:
        ex.reduce_framebuffer()
    elif comm is not None:
        comm.gather_film()                     # read-out: every rank ends up with the whole film (band all-gather, 4 MB per rank)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    k_ms_overlapped, _ = r.kernel_time("spcbpt_render")   # span of the same kernel while frames overlap (timed region)
    r.enable_kernel_timing(False)

    # ---- beside the contract's K steps: a long steady-state run of the same loop, and the reference's own loop form
    ms_long = None
    if args.long_steps > 0:
        nl = -(-args.long_steps // batch) * batch      # whole launches
        barrier()
        t1 = time.perf_counter()
        state["phase_left"] = nl
        for f in range(nl):
            step(2000 + f)
        flush(ahead_at_end=True)
        barrier()
        ms_long = (time.perf_counter() - t1) / nl * 1e3
        if dist is not None:
            t = torch.tensor([ms_long], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms_long = float(t.item())
    ms_sync_each = None
    ms_sync_each_ahead = None
    ms_viewer = None
    ms_viewer_moving = None
    if args.sync_each_frames > 0 and ex is None:
        # optixPathTracer.cpp:791-822: launchLVCTrace (light pass + LVC_Process) then launchSubframe, a device sync after each
        # (513, 634) -- one frame in flight, nothing batched, nothing ahead
        r.sync()
        r.set_light_ahead(False)
        for f in range(2):                             # warm-up of this loop form
            r.launch("light trace", 5000 + f)
            if comm is not None: comm.exchange_lvc()
            r.build_sampler(); r.launch("SPCBPT_eye", 5000 + f, rows); r.sync()
        barrier()
        t1 = time.perf_counter()
        for f in range(args.sync_each_frames):
            r.launch("light trace", 5100 + f)
            if comm is not None: comm.exchange_lvc()
            r.build_sampler()
            r.launch("SPCBPT_eye", 5100 + f, rows)
            r.sync()
        ms_sync_each = (time.perf_counter() - t1) / args.sync_each_frames * 1e3
        if dist is not None:
            t = torch.tensor([ms_sync_each], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms_sync_each = float(t.item())
        if comm is None:
            # the same loop with the viewer's opt-in (spcbpt_viewer_set_light_ahead): the next frame's light pass is launched beside
            # this frame's eye kernel, before the sync -- still one eye launch and one device sync per frame
            r.set_light_ahead(True)
            r.launch("light trace", 5200)
            for f in range(args.sync_each_frames + 2):
                if f == 2:
                    t1 = time.perf_counter()
                r.build_sampler()
                r.launch("SPCBPT_eye", 5200 + f, rows)
                r.launch("light trace", 5201 + f)
                r.sync()
            ms_sync_each_ahead = (time.perf_counter() - t1) / args.sync_each_frames * 1e3
            r.set_light_ahead(False)
            # the interactive loop as the library's viewer runs it by default (csrc/viewer.cpp, pipeline 2): every call shows one
            # complete frame -- the reference loop's frame, bit for bit (tests/test_viewer.py) -- and the next one is traced meanwhile
            # ... on a context of its own, created the way an interactive host creates one (two render streams: the library's
            # default; the bench context above runs ONE stream for its 32-frame launches) with the same tuple installed
            os.environ["SPCBPT_RENDER_STREAMS"] = "2"; os.environ["SPCBPT_EYE_BATCH"] = "1"
            try:
                rv = pkg.Renderer(scene, local_rank)
            finally:
                os.environ["SPCBPT_RENDER_STREAMS"] = str(streams); os.environ["SPCBPT_EYE_BATCH"] = str(batch)
            rv.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], args.width / args.height)
            rv.resize(args.width, args.height)
            rv.set_light_trace(ncore, pad, mpc)
            rv.set_subspace(*tup)
            v = pkg.api.Viewer(rv, cam["eye"], cam["lookat"], cam["up"], cam["fov"], args.width, args.height)
            for f in range(args.sync_each_frames + 3):
                if f == 3:
                    t1 = time.perf_counter()
                v.frame()
            ms_viewer = (time.perf_counter() - t1) / args.sync_each_frames * 1e3
            # ... and while the camera is dragged: one cursor event per displayed frame, so every call restarts the accumulation;
            # the loop then does not speculate (a frame queued in such a call would be dropped by the next event after running to
            # completion beside the real one) and costs what its light-pass-ahead mode costs
            v.mouse_button("left", 1, 400, 300)
            for f in range(args.sync_each_frames + 3):
                if f == 3:
                    t1 = time.perf_counter()
                v.cursor_pos(401 + (f % 16), 300 + (f % 7))
                v.frame()
            ms_viewer_moving = (time.perf_counter() - t1) / args.sync_each_frames * 1e3
            v.mouse_button("left", 0, 400, 300)
            v.close()           # drops what was traced ahead and hands the context back (viewer.cpp: spcbpt_viewer_destroy)
            rv.close()

    eye_paths = args.width * args.height
    total_paths = (eye_paths + M) * args.steps
    value = total_paths / dt / 1e6
    achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0

    if args.write_image and rank == 0:
        img = r.read_accum()
        np.save(args.write_image, img)

    if rank == 0:
        # HBM-side traffic comes from rocprofv3 PMC passes (tools/profile_round.sh -> tools/pmc_summary.py), which cannot run inside
        # this process; the committed summary is quoted only if it profiled THIS code (hash of csrc/) in THIS launch form
        traffic = traffic_low = valu_issue = None
        unit_fracs = {}
        pmc_kernel_ms = None
        traffic_note = "no PMC summary for this code and launch form (profiles/traffic_latest.json)"
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tfile) and world == 1 and ex is None and comm is None and args.light_geometry == "lane" and args.tuple == "trained":
            try:
                t = json.load(open(tfile))
                if t.get("kernel_hash") == pkg.api.kernel_hash():
                    # measured per launch of `frames_per_launch` frames; the bytes go with the frames (r02d: 56.6 GB per 4, r02e: 224.4 GB per 16)
                    per = batch / float(int(t.get("frames_per_launch", 1)))
                    traffic = t.get("spcbpt_render_hbm_bytes_per_launch") * per
                    traffic_low = t.get("spcbpt_render_hbm_bytes_per_launch_low") * per
                    valu_issue = t.get("valu_issue_frac")
                    unit_fracs = t.get("unit_fractions") or {}
                    pmc_kernel_ms = (t.get("hbm_measured_frac") or {}).get("kernel_ms")
                    traffic_note = ("rocprofv3 PMC passes of this kernel code (" + str(t.get("tag")) + f", {t.get('frames_per_launch')} frames per launch, scaled to the {batch} of this run): "
                                    "traffic = 2*FETCH_SIZE + WRITE_SIZE (upper bracket, gfx950 half-count correction for 128-B requests), traffic_low = FETCH_SIZE + WRITE_SIZE; Infinity-Cache hits are included in both")
            except Exception:
                pass
        achieved_actual = bytes_actual_per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # Which unit is nearest saturation (round 4): the memory-side byte counters over THIS run's kernel time, and
        # the unit-busy counters of the PMC passes (tools/pmc_summary.py: unit_fractions; VALU issue priced at the 2.55 cycles per
        # wave-instruction measured by tools/micro/valu_issue.hip at this kernel's 4 waves per SIMD).  No unit is saturated: the
        # vector L1 path (TA busy, one 64-B tag lookup per cycle per CU) is nearest, VALU issue next, HBM well below both.
        hbm_measured = None
        if traffic is not None and k_ms > 0:
            hbm_measured = {"low": round(traffic_low / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "high": round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        units = {k: round(unit_fracs[k], 4) for k in ("ta_busy_frac", "tcp_tag_lookups_per_cu_cycle", "valu_issue_frac", "tcp_pending_stall_frac", "l1_read_miss_rate",
                                                       "valu_lane_utilisation", "wave_wait_frac", "wave_active_frac") if k in unit_fracs}
        if hbm_measured: units["hbm_measured_frac_high"] = hbm_measured["high"]
        saturating = {k: v for k, v in units.items() if k in ("ta_busy_frac", "tcp_tag_lookups_per_cu_cycle", "valu_issue_frac", "hbm_measured_frac_high")}
        nearest = max(saturating, key=saturating.get) if saturating else None
        # bytes no implementation of the path can avoid fetching: per event the LESSER of the reference's order and the order executed
        # (the executed order reads 32 CMF values per first stage where the bisection probes 10, and 1 tree descent per vertex where
        # the reference re-descends per connection: neither order is the lesser on every event)
        bytes_min_per_launch = pkg.algorithmic_bytes({k: min(c_eye[k], c_ref[k]) for k in c_eye}) * batch
        out = {
            "metric": "Mpaths/sec (whole node), SPCBPT, 1920x1080",
            "value": round(value, 3), "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "ms_per_step_long": None if ms_long is None else round(ms_long, 3),
            "ms_per_frame_sync_each": None if ms_sync_each is None else round(ms_sync_each, 3),
            "ms_per_frame_sync_each_light_ahead": None if ms_sync_each_ahead is None else round(ms_sync_each_ahead, 3),
            "ms_per_frame_viewer": None if ms_viewer is None else round(ms_viewer, 3),
            "ms_per_frame_viewer_moving": None if ms_viewer_moving is None else round(ms_viewer_moving, 3),
            "notes": {"ms_per_step_long": f"the same loop over {args.long_steps} more steps (steady state; value / ms_per_step are the contract's {args.steps} steps)",
                      "ms_per_frame_sync_each": "the reference's loop form (optixPathTracer.cpp:791-822): one light pass, one sampler build, one eye launch and a device sync per frame",
                      "ms_per_frame_viewer": "spcbpt_viewer_frame in its default mode: one complete, displayable frame per call (the same frames as the reference's loop), the next frame traced while this one is shown",
                      "ms_per_frame_viewer_moving": "the same loop while the camera is dragged (an event before every call: each frame is subframe 0 of a new view; nothing is traced ahead to be dropped)"},
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} scene{' read from glTF' if args.scene_route == 'gltf' else ''} ({info['n_triangles']} triangles, SAH BVH of {info['n_bvh_nodes']} quantised 4-wide nodes, depth "
                                   f"{info['bvh_depth']}), {args.width}x{args.height}, SPCBPT: {M} light paths + "
                                   f"{eye_paths} eye paths per subframe, CONNECTION_N=3, subspace tuple: {args.tuple}, light pass geometry "
                                   f"{ncore} cores x {mpc} paths x {pad} slots ({args.light_geometry})",
                       "preprocess_s": round(t_pre, 2), "frames_in_flight": streams * batch, "frames_per_eye_launch": batch, "light_passes_ahead": (batch if lbatch else depth) if ahead else 0, "light_passes_per_launch": batch if lbatch else 1, "sampler_builds_ahead": batch if build_ahead else 0, "parallelism": "1 GPU" if world == 1 and comm is None else f"{world} GPUs: interleaved 8-row bands, LVC all-gather + film band gather over RCCL ({'C++ host libspcbpt_mgpu: ' + comm_transport + ' transport, ' + str(comm_world) + ' ranks seen by the communicator' if comm is not None else 'torch.distributed harness (--exchange python)'}" + (f", shard capacity {comm.shard_capacity} vertices, {'one exchange per light batch' if xbatch else 'one exchange per frame'}" if comm is not None else "") + ")",
                       "host": "single context" if dist is None else ("libspcbpt_mgpu" if comm is not None else "torch.distributed harness"), "rccl_ranks": comm_world if comm is not None else (world if dist is not None else 0)},
            # `bound` names the contract's roofline (north_star: the HBM-read roofline; the path has no MFMA work); what really limits the
            # kernel is in `limiter` -- no unit is saturated, the waves' issue slots and their waits for gathers are
            "roofline": {"bound": "hbm", "limiter": "latency / VALU issue of partly filled waves (no unit saturated)" if nearest else None,
                         "bound_note": ("contract roofline = algorithmic HBM bytes / kernel time / 8 TB/s (frac); the kernel is NOT HBM-bound: "
                                        "hbm_measured_frac is the memory-side counters over the same kernel time, unit_busy the direct counters -- "
                                        f"nearest saturation: {nearest} = {saturating[nearest]}" if nearest else
                                        "contract roofline; no PMC summary for this code (profiles/traffic_latest.json), so the nearest unit is not stated"),
                         "kernel": "k_spcbpt (spcbpt_render megakernel)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "hbm_measured_frac": hbm_measured, "unit_busy": units,
                         "frac_min_events": round(bytes_min_per_launch / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if k_ms > 0 else 0.0,
                         "algorithmic_bytes_per_launch": int(bytes_per_launch), "kernel_ms": round(k_ms, 4), "launches": k_n,
                         "events": "as executed by the timed kernel (labels cached per vertex, both resampling stages through guide tables, Gamma / Q from its table) x the record sizes of SURVEY 8(d)",
                         # the contract as the survey wrote it: the reference algorithm's own event order (its relabels and bisections)
                         "contract_reference_order": {"algorithmic_bytes_per_launch": int(bytes_ref_per_launch),
                                                      "achieved": round(bytes_ref_per_launch / (k_ms * 1e-3) / 1e9, 2) if k_ms > 0 else 0.0,
                                                      "frac": round(bytes_ref_per_launch / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if k_ms > 0 else 0.0,
                                                      "events_per_eye_path": {k: round(v / max(c_ref["eye_paths"], 1), 3) for k, v in c_ref.items()
                                                                              if k in ("node_visits", "tri_tests", "tree_nodes", "cmf_probes", "gamma_q_reads", "connections")}},
                         # the honest second reading: the same events at the record sizes this build actually fetches, the
                         # memory-side brackets, and what the kernel is really bound by (VALU issue on partly filled waves)
                         "actual": {"bytes_per_launch": int(bytes_actual_per_launch), "achieved": round(achieved_actual, 2),
                                    "frac": round(achieved_actual / HBM_PEAK_GBS, 5),
                                    "record_bytes": pkg.api.BYTES_ACTUAL, "traffic_low": traffic_low, "traffic_high": traffic,
                                    "traffic_frac_low": None if traffic_low is None or k_ms <= 0 else round(traffic_low / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                    "traffic_frac_high": None if traffic is None or k_ms <= 0 else round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                    "valu_issue_frac": valu_issue, "lane_utilisation": lane_util, "traffic_source": traffic_note}},
            "kernels_ms": {"spcbpt_render": round(k_ms, 4), "light_trace": round(lt_ms, 4), "lvc_compact": round(cp_ms, 4),
                           "sampler_build": round(sb_ms, 4),
                           "spcbpt_render_span_in_timed_region": round(k_ms_overlapped, 4),
                           "note": "HIP-event durations from a pass with nothing else on the GPU (sync before and after each eye launch); in the timed "
                                   "region frames overlap (light pass of f+1 and eye kernel of f+1 under the drain of f), so the "
                                   "span of a kernel there includes sharing the GPU and ms_per_step is shorter than kernel_ms"},
            "events_per_eye_path": {k: round(v / max(c_eye["eye_paths"], 1), 3) for k, v in c_eye.items()
                                    if k in ("closest_rays", "shadow_rays", "node_visits", "tri_tests", "surface_vertices",
                                             "connections", "tree_nodes", "cmf_probes", "gamma_q_reads")},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pkg, scene, args, tup)
        if args.fast_math_line and world == 1 and comm is None and ex is None and not os.environ.get("SPCBPT_LIB") and os.path.exists(pkg.api.FAST_LIB_PATH):
            out["fast_math_build"] = fast_math_line(pkg, args)
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
