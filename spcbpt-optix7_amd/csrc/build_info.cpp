// Which sources this library was built from (csrc/source_hash.py; checked by api.load_library at load time), and the sizes of
// the ABI's structs as compiled.
#include "../../include/spcbpt.h"
#ifndef SPCBPT_SOURCE_HASH
#define SPCBPT_SOURCE_HASH "unknown"
#endif
extern "C" const char* spcbpt_build_source_hash() { return SPCBPT_SOURCE_HASH; }
extern "C" int spcbpt_abi_struct_sizes(int32_t* sizes, int capacity) {
    const int32_t s[] = {(int32_t)sizeof(spcbpt_material), (int32_t)sizeof(spcbpt_texture), (int32_t)sizeof(spcbpt_quad_light),
                         (int32_t)sizeof(spcbpt_scene_desc), (int32_t)sizeof(spcbpt_tree_node), (int32_t)sizeof(spcbpt_light_trace_params),
                         (int32_t)sizeof(spcbpt_light_vertex), (int32_t)sizeof(spcbpt_subspace), (int32_t)sizeof(spcbpt_counters),
                         (int32_t)sizeof(spcbpt_unit_eye_vertex), (int32_t)sizeof(spcbpt_pretrace_path), (int32_t)sizeof(spcbpt_pretrace_node),
                         (int32_t)sizeof(spcbpt_viewer_state)};
    const int n = (int)(sizeof(s) / sizeof(s[0]));
    for (int i = 0; sizes && i < n && i < capacity; i++) sizes[i] = s[i];
    return n;
}
