// Host-side set-up of the environment map (env_file.cpp)
#pragma once
#include <vector>

namespace spc {
void env_build(const float* raster, int w, int h, std::vector<float>& tex, std::vector<float>& cmf);
}
