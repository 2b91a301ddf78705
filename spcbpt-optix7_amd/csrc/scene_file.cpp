// `.scene` + OBJ ingestion: the input format of the reference app.
//   grammar      OptiXPathTracer/sceneLoader.cpp:47-308 (LoadScene): line oriented, '#' in column 0 = comment, a block starts at
//                a line that sscanf's as `material <name>` or merely CONTAINS `light`, `properties`, `cameraSetting` or `mesh`, and
//                runs to the first line containing '}'
//   hand-off     OptiXPathTracer/scene_shift.cpp:32-328: only color / metallic / roughness / albedo texture reach the renderer
//                (q17), the k-th mesh block uses the k-th pushed material, OBJ normals are discarded, missing UVs are zero
// OBJ reading has the semantics of the reference's vendored old-API tinyobj and PPM decoding those of its stb_image; both are
// pinned bit-exactly against those loaders (oracle/_ref, tests/golden/ref_loaders.npz, tests/test_scene_file.py).
// Textures: JPEG, PNG and binary PPM (image_file.cpp, held bit-exactly to the reference's stb_image); a file that cannot be
// decoded is reported in the warnings string and the material renders with its flat colour.
// The `.scene` grammar itself is PARITY UNPINNED: sceneLoader.cpp needs the CMake-generated sampleConfig.h (via sutil.h).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/spcbpt.h"
#include "scene_file.h"

namespace spc_loader {
bool load_ppm(const std::string& path, std::vector<uint8_t>& rgba, int& w, int& h) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0};
    int maxv = 0;
    bool ok = fscanf(f, "%2s", magic) == 1 && strcmp(magic, "P6") == 0;
    auto next_int = [&](int& v) {
        int c;
        while ((c = fgetc(f)) != EOF) {
            if (c == '#') { while ((c = fgetc(f)) != EOF && c != '\n') {} continue; }
            if (c > ' ') { ungetc(c, f); break; }
        }
        return fscanf(f, "%d", &v) == 1;
    };
    ok = ok && next_int(w) && next_int(h) && next_int(maxv) && maxv == 255 && w > 0 && h > 0 && (long long)w * h < (1ll << 28);
    if (ok) {
        fgetc(f);
        std::vector<uint8_t> rgb((size_t)3 * w * h);
        ok = fread(rgb.data(), 1, rgb.size(), f) == rgb.size();
        if (ok) {
            rgba.resize((size_t)4 * w * h);
            for (size_t i = 0; i < (size_t)w * h; i++) { rgba[4 * i] = rgb[3 * i]; rgba[4 * i + 1] = rgb[3 * i + 1]; rgba[4 * i + 2] = rgb[3 * i + 2]; rgba[4 * i + 3] = 255; }
        }
    }
    fclose(f);
    return ok;
}
}  // namespace spc_loader

namespace {
using spc_loader::load_image;

const int kMaxLine = 2048;

struct MatParam {  // MaterialParameter defaults (material_parameters.h:16-32)
    float color[3] = {1, 1, 1};
    float metallic = 0.0f, roughness = 0.5f;
    int brdf = 0;   // BrdfType DISNEY (material_parameters.h:10-12, 30)
    std::string tex = "None";
};

std::string fix_slashes(std::string p) {
    for (auto& ch : p) if (ch == '\\') ch = '/';
    return p;
}

// OBJ reader with the semantics of the reference's vendored (old-API) tinyobj as scene_shift.cpp:187-250 consumes it; pinned
// bit-exactly against that loader by tests/test_scene_file.py (vectors: tests/golden/ref_loaders.npz).  Restated behaviour:
//  * leading blanks are skipped; `#` lines are comments; `v`, `vn` (only counted), `vt`, `f`, and the group breaks
//    `usemtl`, `g`, `o`; everything else is ignored (a missing .mtl only produces a warning there);
//  * face corners are i, i/j, i//k or i/j/k; index 0 means 0, negative indices are relative to the current counts;
//  * faces collect into a group that is flushed into ONE shape at every `usemtl` / `g` / `o` and at end of file;
//    vertices are de-duplicated per flush on the whole (v, vn, vt) triple -- normals are discarded later but still split
//    vertices -- and created in fan order (corner 0, k-1, k);
//  * a texcoord pair is appended only for corners that carry a vt index, so the UV array of a shape with mixed corners is
//    shorter than (and misaligned with) the position array; scene_shift.cpp:204-207 pads it with zeros at the END.  The
//    misalignment is the reference's behaviour and is kept.
struct ObjCorner {
    int v, vn, vt;
    bool operator<(const ObjCorner& o) const {
        if (v != o.v) return v < o.v;
        if (vn != o.vn) return vn < o.vn;
        return vt < o.vt;
    }
};
int obj_fix_index(int idx, int n) { return idx > 0 ? idx - 1 : (idx == 0 ? 0 : n + idx); }
// One number of a `v` / `vt` line.  The reference's loader does not use strtod: its grammar is [sign] digits [. digits]
// [(e|E) [sign] digits] -- ".5" or "-.5" are NOT numbers and read as 0 -- and its value is assembled as
// sign * ldexp(m * 5^e, e) with m accumulated digit by digit in double (fraction digit k adds d * 10^-k).  Restated so
// that coordinates are bit-identical, including the quirks.
float obj_float(const char*& p) {
    p += strspn(p, " \t");
    const char* q = p;
    const char* end = p + strcspn(p, " \t\r");
    p = end;
    if (q >= end) return 0.0f;
    double sign = 1.0;
    if (*q == '+' || *q == '-') { sign = *q == '-' ? -1.0 : 1.0; q++; }
    double m = 0.0;
    int digits = 0;
    while (q != end && *q >= '0' && *q <= '9') { m = m * 10.0 + (double)(*q - '0'); q++; digits++; }
    if (digits == 0) return 0.0f;  // also covers a lone sign and a leading '.'
    int e = 0;
    if (q != end && *q == '.') {
        q++;
        for (int k = 1; q != end && *q >= '0' && *q <= '9'; q++, k++) m += (double)(*q - '0') * pow(10.0, (double)-k);
    }
    if (q != end && (*q == 'e' || *q == 'E')) {
        q++;
        int es = 1;
        if (q != end && (*q == '+' || *q == '-')) { es = *q == '-' ? -1 : 1; q++; }
        else if (!(q != end && *q >= '0' && *q <= '9')) return 0.0f;  // empty exponent: the whole number fails
        int ed = 0;
        while (q != end && *q >= '0' && *q <= '9') { e = e * 10 + (*q - '0'); q++; ed++; }
        if (ed == 0) return 0.0f;
        e *= es;
    }
    return (float)(sign * ldexp(m * pow(5.0, (double)e), e));
}
bool load_obj(const std::string& path, spcbpt_scene_file& s, int material) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    std::vector<float> pos, tc;
    int n_normals = 0;
    std::vector<std::vector<ObjCorner>> group;
    char raw[8192];
    auto flush = [&]() {
        if (group.empty()) return;
        std::map<ObjCorner, uint32_t> cache;
        const size_t v0 = s.V.size() / 3;  // first vertex of this shape in the soup
        std::vector<float> uv;             // this shape's texcoords, tinyobj style (only corners with a vt)
        auto vertex = [&](const ObjCorner& c) -> uint32_t {
            auto it = cache.find(c);
            if (it != cache.end()) return it->second;
            const uint32_t id = (uint32_t)(s.V.size() / 3);
            const bool ok = c.v >= 0 && (size_t)c.v * 3 + 2 < pos.size();
            for (int k = 0; k < 3; k++) s.V.push_back(ok ? pos[3 * (size_t)c.v + k] : 0.0f);
            if (c.vt >= 0 && (size_t)c.vt * 2 + 1 < tc.size()) { uv.push_back(tc[2 * (size_t)c.vt]); uv.push_back(tc[2 * (size_t)c.vt + 1]); }
            cache[c] = id;
            return id;
        };
        for (const auto& face : group) {
            if (face.size() < 3) continue;
            for (size_t k = 2; k < face.size(); k++) {
                const uint32_t a = vertex(face[0]), b = vertex(face[k - 1]), c = vertex(face[k]);
                s.I.push_back(a); s.I.push_back(b); s.I.push_back(c);
                s.M.push_back(material);
            }
        }
        const size_t nv = s.V.size() / 3 - v0;
        for (size_t i = 0; i < nv; i += 3)   // get_aabb(std::vector<float>): floats i, i + 1, i + 2 for i = 0, 3, 6, ... < size / 3
            for (int k = 0; k < 3; k++) {
                const float x = s.V[3 * v0 + i + (size_t)k];
                s.ref_lo[k] = std::min(s.ref_lo[k], x); s.ref_hi[k] = std::max(s.ref_hi[k], x);
            }
        uv.resize(2 * nv, 0.0f);  // scene_shift.cpp:204-207 (pads, never truncates: uv.size() <= 2 nv by construction)
        s.UV.insert(s.UV.end(), uv.begin(), uv.end());
        group.clear();
    };
    while (fgets(raw, sizeof(raw), f)) {
        size_t n = strlen(raw);
        while (n > 0 && (raw[n - 1] == '\n' || raw[n - 1] == '\r')) raw[--n] = 0;
        const char* p = raw + strspn(raw, " \t");
        if (*p == 0 || *p == '#') continue;
        auto blank = [](char c) { return c == ' ' || c == '\t'; };
        if (p[0] == 'v' && blank(p[1])) {
            p += 2;
            const float x = obj_float(p), y = obj_float(p), z = obj_float(p);
            pos.push_back(x); pos.push_back(y); pos.push_back(z);
        } else if (p[0] == 'v' && p[1] == 'n' && blank(p[2])) {
            n_normals++;
        } else if (p[0] == 'v' && p[1] == 't' && blank(p[2])) {
            p += 3;
            const float u = obj_float(p), v = obj_float(p);
            tc.push_back(u); tc.push_back(v);
        } else if (p[0] == 'f' && blank(p[1])) {
            p += 2;
            p += strspn(p, " \t");
            std::vector<ObjCorner> face;
            const int nv = (int)(pos.size() / 3), nt = (int)(tc.size() / 2);
            while (*p && *p != '\r' && *p != '\n') {
                ObjCorner c = {-1, -1, -1};
                c.v = obj_fix_index(atoi(p), nv);
                p += strcspn(p, "/ \t\r");
                if (*p == '/') {
                    p++;
                    if (*p == '/') {  // i//k
                        p++;
                        c.vn = obj_fix_index(atoi(p), n_normals);
                        p += strcspn(p, "/ \t\r");
                    } else {          // i/j or i/j/k
                        c.vt = obj_fix_index(atoi(p), nt);
                        p += strcspn(p, "/ \t\r");
                        if (*p == '/') {
                            p++;
                            c.vn = obj_fix_index(atoi(p), n_normals);
                            p += strcspn(p, "/ \t\r");
                        }
                    }
                }
                face.push_back(c);
                p += strspn(p, " \t\r");
            }
            group.push_back(face);
        } else if ((strncmp(p, "usemtl", 6) == 0 && blank(p[6])) || (p[0] == 'g' && blank(p[1])) || (p[0] == 'o' && blank(p[1]))) {
            flush();
        }
    }
    flush();
    fclose(f);
    return true;
}

}  // namespace

extern "C" {

static int scene_file_load_impl(const char* scene_path, const char* data_root, spcbpt_scene_file** out) {
    if (!scene_path || !out) return SPCBPT_ERR_INVALID_ARG;
    *out = nullptr;
    FILE* file = fopen(scene_path, "r");
    if (!file) return SPCBPT_ERR_INVALID_ARG;
    const std::string root = data_root ? std::string(data_root) : std::string();
    spcbpt_scene_file* s = new spcbpt_scene_file();
    std::map<std::string, MatParam> materials_map;
    std::map<std::string, int> texture_ids;
    std::vector<std::string> texture_paths;
    std::vector<MatParam> mesh_materials;
    std::vector<std::string> mesh_files;
    char line[kMaxLine];
    while (fgets(line, kMaxLine, file)) {
        if (line[0] == '#') continue;
        char name[kMaxLine] = {0};
        if (sscanf(line, " material %2047s", name) == 1) {  // sceneLoader.cpp:77-126
            MatParam m;
            char tex_name[kMaxLine] = "None";
            float dummy[3];
            while (fgets(line, kMaxLine, file)) {
                if (strchr(line, '}')) break;
                sscanf(line, " name %2047s", name);
                sscanf(line, " color %f %f %f", &m.color[0], &m.color[1], &m.color[2]);
                sscanf(line, " albedoTex %2047s", tex_name);
                sscanf(line, " metallic %f", &m.metallic);
                sscanf(line, " roughness %f", &m.roughness);
                sscanf(line, " brdf %i", &m.brdf);   // sceneLoader.cpp:107; reaches the renderer as Pbr::brdf (scene_shift.cpp:75)
                (void)dummy;  // emission / subsurface / specular / ... are parsed by the reference but never reach the renderer (q17)
            }
            m.tex = tex_name;
            materials_map[name] = m;
        }
        if (strstr(line, "light")) {  // sceneLoader.cpp:130-193
            float position[3] = {0, 0, 0}, emission[3] = {0, 0, 0}, v1[3] = {0, 0, 0}, v2[3] = {0, 0, 0};
            char light_type[64] = "None";
            int div = 1;
            while (fgets(line, kMaxLine, file)) {
                if (strchr(line, '}')) break;
                sscanf(line, " position %f %f %f", &position[0], &position[1], &position[2]);
                sscanf(line, " emission %f %f %f", &emission[0], &emission[1], &emission[2]);
                sscanf(line, " v1 %f %f %f", &v1[0], &v1[1], &v1[2]);
                sscanf(line, " v2 %f %f %f", &v2[0], &v2[1], &v2[2]);
                sscanf(line, " type %63s", light_type);
                sscanf(line, " divLevel %d", &div);
            }
            if (strcmp(light_type, "Quad") == 0) {
                spcbpt_quad_light L;
                for (int k = 0; k < 3; k++) { L.position[k] = position[k]; L.u[k] = v1[k] - position[k]; L.v[k] = v2[k] - position[k]; L.emission[k] = emission[k]; }
                L.div_level = div;
                s->lights.push_back(L);
            } else {
                s->warnings += std::string("light of type '") + light_type + "' skipped (only Quad is functional, SURVEY A9); ";
            }
        }
        if (strstr(line, "properties")) {  // sceneLoader.cpp:205-217
            while (fgets(line, kMaxLine, file)) {
                if (strchr(line, '}')) break;
                sscanf(line, " width %i", &s->width);
                sscanf(line, " height %i", &s->height);
            }
        }
        if (strstr(line, "cameraSetting")) {  // sceneLoader.cpp:221-254
            while (fgets(line, kMaxLine, file)) {
                if (strchr(line, '}')) break;
                sscanf(line, " eye %f %f %f", &s->eye[0], &s->eye[1], &s->eye[2]);
                sscanf(line, " lookat %f %f %f", &s->lookat[0], &s->lookat[1], &s->lookat[2]);
                sscanf(line, " up %f %f %f", &s->up[0], &s->up[1], &s->up[2]);
                sscanf(line, " fov %f", &s->fov);
                char ef[256] = "";
                if (sscanf(line, " env_file %255s", ef) == 1) s->env_file = fix_slashes(ef);   // (env_lum is parsed upstream and used nowhere)
            }
        }
        if (strstr(line, "mesh")) {  // sceneLoader.cpp:258-300
            while (fgets(line, kMaxLine, file)) {
                if (strchr(line, '}')) break;
                char path[kMaxLine];
                if (sscanf(line, " file %2047s", path) == 1) mesh_files.push_back(fix_slashes(path));
                if (sscanf(line, " material %2047s", path) == 1) {
                    auto it = materials_map.find(path);
                    if (it != materials_map.end()) mesh_materials.push_back(it->second);
                    else s->warnings += std::string("could not find material ") + path + "; ";
                }
            }
            s->n_mesh_blocks++;
        }
    }
    fclose(file);
    // materials: the k-th mesh uses the k-th pushed material (scene_shift.cpp:235)
    for (size_t k = 0; k < mesh_materials.size(); k++) {
        const MatParam& p = mesh_materials[k];
        spcbpt_material m;
        memset(&m, 0, sizeof(m));
        memcpy(m.base_color, p.color, 12);
        m.metallic = p.metallic; m.roughness = p.roughness; m.brdf = p.brdf;
        m.specular = 0.5f; m.sheen_tint = 0.5f; m.clearcoat_gloss = 1.0f;  // MaterialData() defaults (q17)
        if (p.tex != "None") {
            auto it = texture_ids.find(p.tex);
            if (it != texture_ids.end()) m.albedo_tex = it->second;
            else {
                std::vector<uint8_t> px;
                int w = 0, h = 0;
                const std::string tp = root + "/" + fix_slashes(p.tex);
                if (load_image(tp, px, w, h)) {
                    s->tex_pixels.push_back(std::move(px));
                    spcbpt_texture t;
                    t.rgba = nullptr; t.width = w; t.height = h;
                    s->textures.push_back(t);
                    texture_ids[p.tex] = (int)s->textures.size();
                    m.albedo_tex = (int)s->textures.size();
                } else {
                    texture_ids[p.tex] = 0;
                    s->warnings += "texture " + p.tex + " not loaded (missing, or not a JPEG / PNG / binary PPM this reader decodes); ";
                }
            }
        }
        s->materials.push_back(m);
    }
    for (size_t k = 0; k < s->textures.size(); k++) s->textures[k].rgba = s->tex_pixels[k].data();
    for (size_t k = 0; k < mesh_files.size(); k++) {
        if (k >= s->materials.size()) { s->warnings += "mesh " + mesh_files[k] + " has no material; skipped; "; continue; }
        if (!load_obj(root + "/" + mesh_files[k], *s, (int)k)) s->warnings += "mesh " + mesh_files[k] + " could not be read; ";
    }
    for (const spcbpt_quad_light& L : s->lights)   // the light quads' meshes enter the scene box with all four corners (scene_shift.cpp:276-317)
        for (int c = 0; c < 4; c++)
            for (int k = 0; k < 3; k++) {
                const float x = L.position[k] + ((c & 1) ? L.u[k] : 0.0f) + ((c & 2) ? L.v[k] : 0.0f);
                s->ref_lo[k] = std::min(s->ref_lo[k], x); s->ref_hi[k] = std::max(s->ref_hi[k], x);
            }
    if (!s->env_file.empty()) {   // env_params_setup: <data>/<env_file> through HDRLoader
        const std::string ep = root + "/" + s->env_file;
        int w = 0, h = 0;
        if (spcbpt_hdr_load(ep.c_str(), &w, &h, nullptr, 0) == SPCBPT_OK) {
            s->env_rgba.resize((size_t)w * h * 4);
            if (spcbpt_hdr_load(ep.c_str(), &w, &h, s->env_rgba.data(), s->env_rgba.size()) == SPCBPT_OK) { s->env_w = w; s->env_h = h; }
        }
        if (!s->env_w) { s->env_rgba.clear(); s->warnings += "environment map " + s->env_file + " not loaded (missing, or not a Radiance RGBE file); "; }
    }
    *out = s;
    return SPCBPT_OK;
}

// The scene's environment map (width = 0: none) and the sky.center / sky.r the reference derives from its scene box
// never throw across the C ABI: a file that asks for more memory than there is (std::bad_alloc from the mesh / texture / raster
// vectors) is reported like any other unusable file
int spcbpt_scene_file_load(const char* scene_path, const char* data_root, spcbpt_scene_file** out) {
    try { return scene_file_load_impl(scene_path, data_root, out); }
    catch (const std::bad_alloc&) { return SPCBPT_ERR_CAPACITY; }
    catch (const std::exception&) { return SPCBPT_ERR_IO; }
}

int spcbpt_scene_file_environment(spcbpt_scene_file* s, const float** rgba, int* width, int* height, float center[3], float* radius) {
    if (!s) return SPCBPT_ERR_INVALID_ARG;
    if (rgba) *rgba = s->env_w ? s->env_rgba.data() : nullptr;
    if (width) *width = s->env_w;
    if (height) *height = s->env_h;
    double d2 = 0.0;
    for (int k = 0; k < 3; k++) {
        if (center) center[k] = 0.5f * (s->ref_lo[k] + s->ref_hi[k]);    // Aabb::center
        const double e = (double)s->ref_lo[k] - (double)s->ref_hi[k];
        d2 += e * e;
    }
    if (radius) *radius = (float)sqrt(d2);                                 // length(aabb.m_min - aabb.m_max)
    return SPCBPT_OK;
}

int spcbpt_scene_file_desc(spcbpt_scene_file* s, spcbpt_scene_desc* d) {
    if (!s || !d) return SPCBPT_ERR_INVALID_ARG;
    d->vertices = s->V.data(); d->texcoords = s->UV.data(); d->n_vertices = (int32_t)(s->V.size() / 3);
    d->indices = s->I.data(); d->tri_material = s->M.data(); d->n_triangles = (int32_t)(s->I.size() / 3);
    d->materials = s->materials.data(); d->n_materials = (int32_t)s->materials.size();
    d->textures = s->textures.data(); d->n_textures = (int32_t)s->textures.size();
    d->lights = s->lights.data(); d->n_lights = (int32_t)s->lights.size();
    return SPCBPT_OK;
}

int spcbpt_scene_file_camera(spcbpt_scene_file* s, float eye[3], float lookat[3], float up[3], float* fov, int* width, int* height) {
    if (!s) return SPCBPT_ERR_INVALID_ARG;
    if (eye) memcpy(eye, s->eye, 12);
    if (lookat) memcpy(lookat, s->lookat, 12);
    if (up) memcpy(up, s->up, 12);
    if (fov) *fov = s->fov;
    if (width) *width = s->width;
    if (height) *height = s->height;
    return SPCBPT_OK;
}

const char* spcbpt_scene_file_warnings(spcbpt_scene_file* s) { return s ? s->warnings.c_str() : ""; }

int spcbpt_scene_file_free(spcbpt_scene_file* s) {
    delete s;
    return SPCBPT_OK;
}

}  // extern "C"
