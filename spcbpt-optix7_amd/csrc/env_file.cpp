// The environment map of a scene, host side: the Radiance .hdr (RGBE) reader and the set-up of params.sky.
//   spcbpt_hdr_load      <- HDRLoader (OptiXPathTracer/scene_shift.cpp:334-500): "#?RADIANCE" header, FORMAT=32-bit_rle_rgbe,
//                           EXPOSURE, "-Y h +X w", new-style RLE scanlines (2 2 hi lo) or flat RGBE; RGBE -> float as
//                           (mantissa + 0.5) * 2^(e - 136) / exposure, e = 0 -> black
//   env_build (internal) <- env_params_setup + envMapCMFBuild + surroundsIndex (OptiXPathTracer/optixPathTracer.cpp:381-461) and the
//                           row flip of HDRLoader::loadTexture (scene_shift.cpp:528-540)
// The reader sits in a file that includes the CMake-generated sampleConfig.h upstream, so it cannot be compiled here: restated,
// parity unpinned (tests/test_env_file.py holds it to an independent decoder of the format).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/spcbpt.h"
#include "env_host.h"

namespace {

void get_line(std::ifstream& in, std::string& s) {   // HDRLoader::getLine: skips '#' comment lines, stops at an empty line
    for (;;) {
        if (!std::getline(in, s)) return;
        if (s.empty()) return;
        const std::string::size_type index = s.find_first_not_of("\n\r\t ");
        if (index != std::string::npos && s[index] != '#') break;
    }
}
struct RGBe { unsigned char v[4]; };

bool read_flat(std::ifstream& in, RGBe* line, size_t wid) {
    in.read(reinterpret_cast<char*>(line), (std::streamsize)(wid * sizeof(RGBe)));
    return (bool)in;
}
bool read_scanline(std::ifstream& in, RGBe* line, size_t wid) {
    const size_t MinLen = 8, MaxLen = 0x7fff;
    if (wid < MinLen || wid > MaxLen) return read_flat(in, line, wid);
    char c0, c1, c2, c3;
    in.get(c0); in.get(c1); in.get(c2); in.get(c3);
    if (!in) return false;
    if (c0 != 2 || c1 != 2 || (c2 & 0x80)) {   // an old-format scanline
        in.putback(c3); in.putback(c2); in.putback(c1); in.putback(c0);
        return read_flat(in, line, wid);
    }
    // (upstream: size_t(c2) << 8 | size_t(c3) with plain chars: a low byte >= 0x80 sign-extends there; taken as unsigned here)
    if ((size_t)((size_t)(unsigned char)c2 << 8 | (size_t)(unsigned char)c3) != wid) return false;
    for (unsigned ch = 0; ch < 4; ch++) {
        for (size_t x = 0; x < wid;) {
            unsigned char code;
            in.get(reinterpret_cast<char&>(code));
            if (!in) return false;
            if (code > 0x80) {   // run
                char pix;
                in.get(pix);
                if (!in) return false;
                code = code & 0x7f;
                while (code--) { if (x >= wid) return false; line[x++].v[ch] = (unsigned char)pix; }
            } else {             // literal span
                while (code--) {
                    if (x >= wid) return false;
                    in.get(reinterpret_cast<char&>(line[x++].v[ch]));
                    if (!in) return false;
                }
            }
        }
    }
    return true;
}

}  // namespace

extern "C" int spcbpt_hdr_load(const char* path, int* width, int* height, float* rgba, size_t capacity_floats) {
    if (!path || !width || !height) return SPCBPT_ERR_INVALID_ARG;
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open()) return SPCBPT_ERR_IO;
    std::string magic, comment;
    float exposure = 1.0f;
    std::getline(in, magic);
    if (magic != "#?RADIANCE") return SPCBPT_ERR_IO;
    for (;;) {
        get_line(in, comment);
        if (comment.empty()) break;
        if (comment[0] == '#') continue;
        if (comment.find("FORMAT") != std::string::npos) {
            if (comment != "FORMAT=32-bit_rle_rgbe") return SPCBPT_ERR_IO;   // RGBe only, not XYZe
            continue;
        }
        const size_t ofs = comment.find("EXPOSURE=");
        if (ofs != std::string::npos) exposure = (float)atof(comment.c_str() + ofs + 9);
    }
    std::string major, minor;
    long ny = 0, nx = 0;
    in >> minor >> ny >> major >> nx;
    if (!in || minor != "-Y" || major != "+X" || nx <= 0 || ny <= 0 || nx > 65536 || ny > 65536) return SPCBPT_ERR_IO;
    if ((long long)nx * ny > (1ll << 26)) return SPCBPT_ERR_CAPACITY;   // spcbpt_set_environment's own limit: nothing larger can be used
    get_line(in, comment);   // the rest of the resolution line
    *width = (int)nx; *height = (int)ny;
    if (!rgba) return SPCBPT_OK;   // size query
    if (capacity_floats < (size_t)nx * (size_t)ny * 4) return SPCBPT_ERR_CAPACITY;
    std::vector<RGBe> raster;
    try { raster.resize((size_t)nx * (size_t)ny); }   // <= 256 MiB after the check above; still never throw across the C ABI
    catch (const std::exception&) { return SPCBPT_ERR_CAPACITY; }
    for (long y = 0; y < ny; y++)
        if (!read_scanline(in, raster.data() + (size_t)nx * y, (size_t)nx)) return SPCBPT_ERR_IO;
    const float inv_img_exposure = 1.0f / exposure;
    for (size_t i = 0; i < raster.size(); i++) {   // RGBEtoFloats
        float* f = rgba + i * 4;
        const RGBe& p = raster[i];
        if (p.v[3] == 0) { f[0] = f[1] = f[2] = 0.0f; }
        else {
            float s = (float)ldexp(1.0, (int)p.v[3] - (128 + 8));
            s *= inv_img_exposure;
            f[0] = (p.v[0] + 0.5f) * s; f[1] = (p.v[1] + 0.5f) * s; f[2] = (p.v[2] + 0.5f) * s;
        }
        f[3] = 0.0f;   // (m_raster is new float[n * 4] with the fourth float never written upstream; only the texture copy sets alpha 1)
    }
    return SPCBPT_OK;
}

namespace spc {

// params.sky from the raster (row 0 = top of the image, as the file stores it)
void env_build(const float* raster, int w, int h, std::vector<float>& tex, std::vector<float>& cmf) {
    const int size = w * h;
    tex.resize((size_t)size * 4);
    for (int i = 0; i < w; i++)          // HDRLoader::loadTexture: texture row j = raster row h - 1 - j, alpha 1
        for (int j = 0; j < h; j++) {
            const float* q = raster + ((size_t)(h - j - 1) * w + i) * 4;
            float* t = &tex[((size_t)j * w + i) * 4];
            t[0] = q[0]; t[1] = q[1]; t[2] = q[2]; t[3] = 1.0f;
        }
    // envMapCMFBuild: every texel's luminance (r + g + b) plus the mean of its up-to-12 neighbours within |dx| + |dy| <= 2,
    // accumulated in float over the raster AS READ (not flipped: as written), normalised, 25 % uniform mixed in
    cmf.assign((size_t)size, 0.0f);
    const float uniform_rate = 0.25f;
    const float uniform_pdf = (float)(1.0 / size);
    for (int i = 0; i < size; i++) {
        const int cx = i % w, cy = i / w;
        int n = 0, idx[13];
        for (int dx = -2; dx <= 2; dx++)
            for (int dy = -2; dy <= 2; dy++)
                if (abs(dx) + abs(dy) <= 2) {
                    const int sx = cx + dx, sy = cy + dy;
                    if (sx >= 0 && sy >= 0 && sx < w && sy < h) idx[n++] = sx + sy * w;
                }
        const float* p = raster + (size_t)i * 4;
        float v = p[0] + p[1] + p[2];
        for (int k = 0; k < n; k++) { const float* q = raster + (size_t)idx[k] * 4; v += (q[0] + q[1] + q[2]) / n; }
        cmf[i] = v;
        if (i >= 1) cmf[i] += cmf[i - 1];
    }
    const float sum = cmf[size - 1];
    for (int i = 0; i < size; i++) {
        cmf[i] /= sum;
        cmf[i] = cmf[i] * (1 - uniform_rate) + (uniform_pdf * (i + 1) * uniform_rate);
    }
}

}  // namespace spc
