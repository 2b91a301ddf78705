"""Multi-GPU decomposition of one SPCBPT frame (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm,
"gloo" on CPU for tests).  The reference is single-GPU; this is new work:

  light pass   rank r traces cores [r*M/N, (r+1)*M/N)  (seeded by GLOBAL core index -> rank-count invariant)
  exchange 1   all-gather of the compact LVC shards (+ their counts); shards concatenate in rank order = global
               (path_id, depth) order, so every rank builds the identical sampler
  eye pass     rank r renders the 8-row bands b with b % N == r
  exchange 2   framebuffer sum over RCCL (bands a rank did not render are zero), only when an image is read out
  start-up     the subspace tuple (trees, Q, CMF-Gamma) is trained on rank 0 and broadcast (SURVEY.md 8(e) "Preprocessing")

PyTorch is plumbing here (device buffers for the collectives); the kernels run behind the C ABI.
"""
from __future__ import annotations

import numpy as np

from .api import LIGHT_VERTEX_DTYPE, NUM_SUBSPACE, TREE_NODE_DTYPE

VERTEX_BYTES = LIGHT_VERTEX_DTYPE.itemsize


def core_range(num_core: int, rank: int, world: int):
    """Contiguous core range of a rank (the last rank takes the remainder)."""
    per = num_core // world
    begin = rank * per
    count = per if rank < world - 1 else num_core - begin
    return begin, count


def band_rows(height: int, rank: int, world: int):
    """(row_begin, row_end, row_step) for spcbpt_launch: every world-th band of 8 rows, starting at band `rank`."""
    return 8 * rank, height, world


def rows_of_rank(height: int, rank: int, world: int):
    return [y for y in range(height) if (y // 8) % world == rank]


class _DevPtr:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can view it without copying."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def device_view(ptr: int, nbytes: int, device):
    import torch
    return torch.as_tensor(_DevPtr(ptr, nbytes), device=device)


def allgather_lvc_host(local: np.ndarray, group=None) -> np.ndarray:
    """CPU/gloo variant of exchange 1 on host arrays of spcbpt_light_vertex records (tests)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64)
    cnts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    cap = int(max(int(c) for c in cnts))
    buf = torch.zeros(cap * VERTEX_BYTES, dtype=torch.uint8)
    raw = np.frombuffer(local.tobytes(), dtype=np.uint8)
    buf[: raw.shape[0]] = torch.from_numpy(raw.copy())
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    parts = [np.frombuffer(out[r][: int(cnts[r]) * VERTEX_BYTES].numpy().tobytes(), dtype=LIGHT_VERTEX_DTYPE) for r in range(world)]
    return np.concatenate(parts)


def allreduce_image_host(img: np.ndarray, group=None) -> np.ndarray:
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(img))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.numpy()


def broadcast_subspace(tup, src: int = 0, device=None, group=None):
    """One-off start-up exchange: the subspace tuple (eye tree, light tree, Q, CMF-Gamma) of rank `src` on every rank.
    The tuple is the product of a training run (device reductions in no fixed order), so ranks that trained for themselves
    would hold slightly different trees and matrices while exchanging light vertices labelled with them; training once and
    broadcasting keeps the labels, Gamma and Q of all ranks identical.  `device` = the rank's GPU for RCCL, None for gloo."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    n = torch.zeros(2, dtype=torch.int64, device=dev)
    if rank == src:
        et, lt, q, g = tup
        et = np.ascontiguousarray(et, dtype=TREE_NODE_DTYPE)
        lt = np.ascontiguousarray(lt, dtype=TREE_NODE_DTYPE)
        n = torch.tensor([et.shape[0], lt.shape[0]], dtype=torch.int64, device=dev)
    dist.broadcast(n, src, group=group)
    ne, nl = (int(v) for v in n.cpu().tolist())
    sizes = [ne * TREE_NODE_DTYPE.itemsize, nl * TREE_NODE_DTYPE.itemsize, NUM_SUBSPACE * 4, NUM_SUBSPACE * NUM_SUBSPACE * 4]
    if rank == src:
        raw = np.concatenate([np.frombuffer(et.tobytes(), np.uint8), np.frombuffer(lt.tobytes(), np.uint8),
                              np.frombuffer(np.ascontiguousarray(q, np.float32).tobytes(), np.uint8),
                              np.frombuffer(np.ascontiguousarray(g, np.float32).tobytes(), np.uint8)])
        buf = torch.from_numpy(raw.copy()).to(dev)
    else:
        buf = torch.zeros(sum(sizes), dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src, group=group)
    raw = buf.cpu().numpy().tobytes()
    o0, o1, o2 = sizes[0], sizes[0] + sizes[1], sizes[0] + sizes[1] + sizes[2]
    return (np.frombuffer(raw[:o0], dtype=TREE_NODE_DTYPE).copy(), np.frombuffer(raw[o0:o1], dtype=TREE_NODE_DTYPE).copy(),
            np.frombuffer(raw[o1:o2], dtype=np.float32).copy(),
            np.frombuffer(raw[o2:], dtype=np.float32).reshape(NUM_SUBSPACE, NUM_SUBSPACE).copy())


class FrameExchanger:
    """GPU (RCCL) exchanges for one Renderer per rank."""

    def __init__(self, renderer, rank: int, world: int, device):
        import torch
        self.r = renderer
        self.rank, self.world = rank, world
        self.device = device
        self.torch = torch
        self.counts = torch.zeros(world, dtype=torch.int32, device=device)
        self.gather_buf = None
        self.cat_bufs = [None, None]   # the import copies out of the staging buffer asynchronously: alternate two of them
        self.frame = 0

    def allgather_lvc(self):
        """After `light trace` on every rank: gathers the shards and installs the global LVC in the context."""
        torch = self.torch
        import torch.distributed as dist
        r = self.r
        dv, dc, cap = r.lvc_export()
        # hand-off from the context's light stream to torch's stream; the render stream is NOT waited for, so the previous
        # frame's eye kernel may still be draining while the shards travel
        r.sync_light()
        my_count = device_view(dc, 8, self.device).view(torch.int32)[:1]
        dist.all_gather_into_tensor(self.counts, my_count.clone())
        counts = self.counts.cpu().tolist()   # one small D2H per frame (the sampler build needs the total on the host anyway)
        mx = max(counts)
        nbytes = mx * VERTEX_BYTES
        if self.gather_buf is None or self.gather_buf.numel() < self.world * nbytes:
            self.gather_buf = torch.empty(self.world * max(nbytes, VERTEX_BYTES), dtype=torch.uint8, device=self.device)
        shard = device_view(dv, cap * VERTEX_BYTES, self.device)[:nbytes]
        out = self.gather_buf[: self.world * nbytes]
        dist.all_gather_into_tensor(out, shard)
        total = sum(counts)
        k2 = self.frame & 1
        self.frame += 1
        r.lvc_import_wait()   # the import two frames ago read staging buffer k2 asynchronously: it must have copied before the
                              # buffer is rewritten (or re-allocated) below
        if self.cat_bufs[k2] is None or self.cat_bufs[k2].numel() < total * VERTEX_BYTES:
            self.cat_bufs[k2] = torch.empty(max(total, 1) * VERTEX_BYTES * 5 // 4, dtype=torch.uint8, device=self.device)
        cat = self.cat_bufs[k2]
        off = 0
        for k, c in enumerate(counts):
            cat[off: off + c * VERTEX_BYTES].copy_(out[k * nbytes: k * nbytes + c * VERTEX_BYTES])
            off += c * VERTEX_BYTES
        torch.cuda.current_stream(self.device).synchronize()   # torch's stream only (a device-wide sync would wait for the eye kernel)
        # queued on the context's light stream, no host wait: `cat` stays untouched until the frame after next (see
        # spcbpt_lvc_import), and the sampler build that follows takes its item count from this call
        r.lvc_import_device(cat.data_ptr(), total)
        return total

    def reduce_framebuffer(self):
        """Sum of the per-rank accum buffers (disjoint bands, zero elsewhere) into a SEPARATE tensor, which is returned: the
        rank's own accum buffer keeps holding its own bands only, so the call may be repeated (a progressive read-out every K
        frames) without summing already-reduced bands again.  (The C++ host gathers bands instead: Comm.gather_film.)"""
        torch = self.torch
        import torch.distributed as dist
        r = self.r
        r.sync()
        acc = device_view(r.accum_device_ptr(), r.width * r.height * 16, self.device).view(torch.float32)
        own = torch.zeros_like(acc)
        own_rows = torch.tensor(rows_of_rank(r.height, self.rank, self.world), dtype=torch.long, device=self.device)
        img = acc.view(r.height, r.width * 4)
        own.view(r.height, r.width * 4)[own_rows] = img[own_rows]     # bands this rank does not own are zero whatever accum holds
        dist.all_reduce(own, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize(self.device)
        return own.view(r.height, r.width, 4)


# ---- the C++ RCCL host (libspcbpt_mgpu.so, include/spcbpt_mgpu.h) -------------------------------------------------------------
import ctypes as _C
import os as _os

MGPU_LIB_PATH = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "csrc", "libspcbpt_mgpu.so")
UNIQUE_ID_BYTES = 128
MGPU_SYMBOLS = ["spcbpt_comm_unique_id", "spcbpt_comm_create", "spcbpt_comm_create_local", "spcbpt_comm_destroy", "spcbpt_comm_last_error",
                "spcbpt_comm_set_shard_capacity", "spcbpt_comm_get_shard_capacity", "spcbpt_comm_calibrate", "spcbpt_comm_exchange_lvc", "spcbpt_comm_exchange_lvc_batch", "spcbpt_comm_info",
                "spcbpt_comm_gather_film", "spcbpt_comm_broadcast_subspace", "spcbpt_comm_barrier", "spcbpt_comm_max_double"]
_mgpu = None


def load_mgpu():
    """ctypes handle of libspcbpt_mgpu.so (needs libspcbpt_hip.so and librccl; loads without a GPU)."""
    global _mgpu
    if _mgpu is None:
        from .api import load_library
        load_library()
        if not _os.path.exists(MGPU_LIB_PATH):
            raise RuntimeError(f"{MGPU_LIB_PATH} missing: run `make -C spcbpt-optix7_amd/csrc` (there is no Python fallback for the N-GPU host)")
        lib = _C.CDLL(MGPU_LIB_PATH, mode=_C.RTLD_GLOBAL)
        vp, i32 = _C.c_void_p, _C.c_int32
        sig = {"spcbpt_comm_unique_id": [_C.c_char_p], "spcbpt_comm_create": [vp, i32, i32, _C.c_char_p, _C.POINTER(vp)],
               "spcbpt_comm_create_local": [_C.POINTER(vp), i32, _C.POINTER(vp)], "spcbpt_comm_destroy": [vp],
               "spcbpt_comm_set_shard_capacity": [vp, i32], "spcbpt_comm_get_shard_capacity": [vp, _C.POINTER(i32)],
               "spcbpt_comm_calibrate": [vp, i32, _C.c_uint32, _C.c_float], "spcbpt_comm_exchange_lvc": [vp],
               "spcbpt_comm_exchange_lvc_batch": [vp, i32], "spcbpt_comm_info": [vp, _C.POINTER(i32), _C.POINTER(i32), _C.POINTER(i32)],
               "spcbpt_comm_gather_film": [vp, vp], "spcbpt_comm_broadcast_subspace": [vp, i32], "spcbpt_comm_barrier": [vp],
               "spcbpt_comm_max_double": [vp, _C.POINTER(_C.c_double)]}
        for name, args in sig.items():
            getattr(lib, name).argtypes = args
            getattr(lib, name).restype = i32
        lib.spcbpt_comm_last_error.argtypes = [vp]
        lib.spcbpt_comm_last_error.restype = _C.c_char_p
        _mgpu = lib
    return _mgpu


def unique_id() -> bytes:
    buf = _C.create_string_buffer(UNIQUE_ID_BYTES)
    if load_mgpu().spcbpt_comm_unique_id(buf) != 0:
        raise RuntimeError("spcbpt_comm_unique_id failed")
    return buf.raw


class Comm:
    """One rank of the C++ N-GPU host.  `Comm(renderer, rank, world, uid)` = RCCL (every rank constructs its own, collectively);
    `Comm.local(renderers)` = ranks sharing one device (tests on a one-GPU box): call a collective on EVERY rank before using its
    result on any of them."""

    def __init__(self, renderer, rank: int, world: int, uid: bytes = None, _handle=None):
        self.lib = load_mgpu()
        self.r, self.rank, self.world = renderer, rank, world
        if _handle is not None:
            self.h = _handle
            return
        h = _C.c_void_p()
        rc = self.lib.spcbpt_comm_create(renderer.h, rank, world, uid, _C.byref(h))
        if rc != 0:
            raise RuntimeError(f"spcbpt_comm_create failed ({rc})")
        self.h = h

    @classmethod
    def local(cls, renderers):
        lib = load_mgpu()
        n = len(renderers)
        ctxs = (_C.c_void_p * n)(*[r.h for r in renderers])
        out = (_C.c_void_p * n)()
        rc = lib.spcbpt_comm_create_local(ctxs, n, out)
        if rc != 0:
            raise RuntimeError(f"spcbpt_comm_create_local failed ({rc})")
        return [cls(r, k, n, _handle=_C.c_void_p(out[k])) for k, r in enumerate(renderers)]

    def _chk(self, rc, what):
        if rc != 0:
            from .api import SpcbptError
            raise SpcbptError(f"{what} failed ({rc}): {self.lib.spcbpt_comm_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.lib.spcbpt_comm_destroy(self.h)
            self.h = None

    def calibrate(self, passes=2, first_frame=900000, slack=1.5):
        self._chk(self.lib.spcbpt_comm_calibrate(self.h, passes, first_frame, slack), "comm_calibrate")
        return self.shard_capacity

    @property
    def shard_capacity(self):
        v = _C.c_int32()
        self._chk(self.lib.spcbpt_comm_get_shard_capacity(self.h, _C.byref(v)), "comm_get_shard_capacity")
        return int(v.value)

    def set_shard_capacity(self, vertices: int):
        self._chk(self.lib.spcbpt_comm_set_shard_capacity(self.h, int(vertices)), "comm_set_shard_capacity")

    def exchange_lvc(self):
        self._chk(self.lib.spcbpt_comm_exchange_lvc(self.h), "comm_exchange_lvc")

    def exchange_lvc_batch(self, n: int):
        """One exchange for the n oldest pending light passes (the passes of one launch_light_batch)."""
        self._chk(self.lib.spcbpt_comm_exchange_lvc_batch(self.h, int(n)), "comm_exchange_lvc_batch")

    def info(self):
        """(rank, world, transport) as the transport reports them; transport "rccl" or "local"."""
        r, w, t = _C.c_int32(), _C.c_int32(), _C.c_int32()
        self._chk(self.lib.spcbpt_comm_info(self.h, _C.byref(r), _C.byref(w), _C.byref(t)), "comm_info")
        return int(r.value), int(w.value), "rccl" if t.value == 0 else "local"

    def gather_film(self, out_device_ptr=None):
        self._chk(self.lib.spcbpt_comm_gather_film(self.h, out_device_ptr), "comm_gather_film")

    def broadcast_subspace(self, root=0):
        self._chk(self.lib.spcbpt_comm_broadcast_subspace(self.h, root), "comm_broadcast_subspace")

    def barrier(self):
        self._chk(self.lib.spcbpt_comm_barrier(self.h), "comm_barrier")

    def max_double(self, v: float) -> float:
        d = _C.c_double(v)
        self._chk(self.lib.spcbpt_comm_max_double(self.h, _C.byref(d)), "comm_max_double")
        return float(d.value)
