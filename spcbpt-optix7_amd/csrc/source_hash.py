"""sha256 (first 16 hex digits) over the sources of libspcbpt_hip.so / libspcbpt_mgpu.so.  The Makefile embeds it in the library
(build_info.cpp -> spcbpt_build_source_hash), api.py recomputes it at load time and refuses a library built from other sources
-- the built .so travels to the GPU box with the tree, and a stale one must not be tested silently.  bench.py uses kernel_hash()
(the device sources of the eye megakernel) to decide whether the committed PMC traffic summary (profiles/traffic_latest.json)
describes the running kernel."""
import hashlib
import os


def source_hash(d=None):
    d = d or os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    for name in ("spcbpt.h", "spcbpt_mgpu.h"):
        h.update(open(os.path.join(d, "..", "..", "include", name), "rb").read())
    return h.hexdigest()[:16]


KERNEL_SOURCES = ("kernels.hip", "kernels.h", "kernel_config.h", "device_lib.h", "dev_traversal.h", "dev_bsdf.h", "dev_sampling.h", "dev_rmis.h", "eye_walk.h", "layout.h", "Makefile")


def kernel_hash(d=None):
    """The same over the DEVICE sources of the eye megakernel only: what a PMC traffic figure of k_spcbpt depends on (host-side
    changes in capi.hip / context.h do not move the kernel's bytes)."""
    d = d or os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
