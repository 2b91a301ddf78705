// Which sources this library was built from (csrc/source_hash.py; checked by api.load_library at load time).
#ifndef SPCBPT_SOURCE_HASH
#define SPCBPT_SOURCE_HASH "unknown"
#endif
extern "C" const char* spcbpt_build_source_hash() { return SPCBPT_SOURCE_HASH; }
