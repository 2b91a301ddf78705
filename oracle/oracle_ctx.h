// ORACLE — test infrastructure only (see vec.h).  The oracle's context shared by oracle_capi.cpp and preprocess.cpp.
#pragma once
#include <vector>

#include "spcbpt_ref.h"

struct orc_ctx {
    orc::Scene scene;
    orc::Params P;
    std::vector<orc::tree_node> eye_tree, light_tree;
    std::vector<float> Q, CMFGamma;
    std::vector<orc::float4> accum;
    std::vector<uint32_t> frame;
    std::vector<orc::BDPTVertex> lvc;
    std::vector<uint8_t> lvc_valid;
    orc::SamplerStorage sampler_storage;
    orc::Counters counters;
    bool count_events = true;
    void* pre = nullptr;  // PreState (preprocess.cpp)
};
void orc_free_pre(orc_ctx* c);
