"""MI355X-native SPCBPT hot path (HIP kernels behind include/spcbpt.h).

The directory name carries a hyphen (it mirrors the reference repository's
name), so import it through `__graft_entry__.load_package()` which registers it
as the module `spcbpt_optix7_amd`.
"""
from .api import (PRETRACE_NODE_DTYPE, PRETRACE_PATH_DTYPE, CONNECTION_N, NUM_SUBSPACE, NUM_SUBSPACE_LIGHTSOURCE, LIGHT_VERTEX_DTYPE, SUBSPACE_DTYPE,
                  TREE_NODE_DTYPE, Renderer, Scene, SpcbptError, algorithmic_bytes, camera_frame, load_library, load_scene_file, load_gltf,
                  single_leaf_tree)
from . import api, dist, scenes

__all__ = ["Renderer", "Scene", "SpcbptError", "scenes", "camera_frame", "single_leaf_tree", "load_library", "load_scene_file", "load_gltf",
           "algorithmic_bytes", "NUM_SUBSPACE", "NUM_SUBSPACE_LIGHTSOURCE", "CONNECTION_N", "LIGHT_VERTEX_DTYPE",
           "SUBSPACE_DTYPE", "TREE_NODE_DTYPE", "PRETRACE_PATH_DTYPE", "PRETRACE_NODE_DTYPE"]
