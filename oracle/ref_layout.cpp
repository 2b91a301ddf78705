// ORACLE -- test infrastructure only.  Layout and default-value pins of the hot path's POD types, taken
// from the reference's OWN headers included from where they lie under /root/reference/src (never copied):
//   cuda/Light.h, cuda/MaterialData.h, OptiXPathTracer/light_parameters.h, OptiXPathTracer/material_parameters.h.
// These four are the only type headers of the path that do not pull in <optix.h> (BDPTVertex.h, optixPathTracer.h, whitted.h,
// cuda/BufferView.h, cuda/GeometryData.h, decisionTree/classTree_common.h all do: unbuildable here, no stand-ins are written).
// Built by `make -C oracle ref` into oracle/_ref/libref_layout.so (git-ignored, not gpurun-ignored).
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <string>

#include <cuda_runtime.h>
#include <sutil/vec_math.h>

#include <cuda/Light.h>
#include <cuda/MaterialData.h>
#include <OptiXPathTracer/light_parameters.h>
#include <OptiXPathTracer/material_parameters.h>

namespace {
std::string g_json;
void kv(const char* k, double v, bool last = false) {
    char b[160];
    snprintf(b, sizeof(b), "\"%s\": %.9g%s", k, v, last ? "" : ", ");
    g_json += b;
}
}  // namespace

#define SZ(T) kv("sizeof " #T, (double)sizeof(T))
#define OFF(T, f) kv("offsetof " #T "." #f, (double)offsetof(T, f))

extern "C" const char* ref_layout_json() {
    g_json = "{";
    // ---- cuda/Light.h:31-92
    SZ(Light); SZ(Light::QUAD); SZ(Light::Point); SZ(Light::Directional);
    OFF(Light, type); OFF(Light, id); OFF(Light, divLevel); OFF(Light, ssBase); OFF(Light, quad);
    OFF(Light::QUAD, corner); OFF(Light::QUAD, u); OFF(Light::QUAD, v); OFF(Light::QUAD, emission); OFF(Light::QUAD, normal); OFF(Light::QUAD, area);
    kv("Light::Type::QUAD", (double)(int)Light::Type::QUAD); kv("Light::Type::DIRECTIONAL", (double)(int)Light::Type::DIRECTIONAL);
    kv("Light::Type::ENV", (double)(int)Light::Type::ENV);
    // ---- cuda/MaterialData.h:33-127
    SZ(MaterialData); SZ(MaterialData::Pbr); SZ(MaterialData::Texture);
    OFF(MaterialData::Pbr, base_color); OFF(MaterialData::Pbr, metallic); OFF(MaterialData::Pbr, roughness); OFF(MaterialData::Pbr, specular);
    OFF(MaterialData::Pbr, specularTint); OFF(MaterialData::Pbr, subsurface); OFF(MaterialData::Pbr, anisotropic); OFF(MaterialData::Pbr, sheen);
    OFF(MaterialData::Pbr, sheenTint); OFF(MaterialData::Pbr, clearcoat); OFF(MaterialData::Pbr, clearcoatGloss);
    OFF(MaterialData::Pbr, base_color_tex); OFF(MaterialData::Pbr, metallic_roughness_tex); OFF(MaterialData::Pbr, brdf);
    OFF(MaterialData, emissive_factor); OFF(MaterialData, id); OFF(MaterialData, doubleSided); OFF(MaterialData, pbr); OFF(MaterialData, light_id);
    {
        MaterialData m;   // the defaults every emissive pseudo-material and every untouched Disney parameter runs with (q17)
        kv("MaterialData().pbr.base_color.x", m.pbr.base_color.x); kv("MaterialData().pbr.base_color.y", m.pbr.base_color.y);
        kv("MaterialData().pbr.base_color.z", m.pbr.base_color.z); kv("MaterialData().pbr.base_color.w", m.pbr.base_color.w);
        kv("MaterialData().pbr.metallic", m.pbr.metallic); kv("MaterialData().pbr.roughness", m.pbr.roughness);
        kv("MaterialData().pbr.specular", m.pbr.specular); kv("MaterialData().pbr.specularTint", m.pbr.specularTint);
        kv("MaterialData().pbr.subsurface", m.pbr.subsurface); kv("MaterialData().pbr.anisotropic", m.pbr.anisotropic);
        kv("MaterialData().pbr.sheen", m.pbr.sheen); kv("MaterialData().pbr.sheenTint", m.pbr.sheenTint);
        kv("MaterialData().pbr.clearcoat", m.pbr.clearcoat); kv("MaterialData().pbr.clearcoatGloss", m.pbr.clearcoatGloss);
        // (pbr.brdf is NOT pinned: `bool brdf = false` sits in a union member, the constructor never writes it -- indeterminate;
        // only scene_shift.cpp:86 sets it, for the .scene materials, and every isBrdf branch of the path is dead)
        kv("MaterialData().doubleSided", m.doubleSided ? 1 : 0);
        kv("MaterialData().emissive_factor.x", m.emissive_factor.x); kv("MaterialData().alpha_mode", (double)m.alpha_mode);
        kv("MaterialData().pbr.base_color_tex.tex", (double)m.pbr.base_color_tex.tex);
    }
    // ---- OptiXPathTracer/light_parameters.h:17-45, material_parameters.h:13-50 (the .scene parser's PODs)
    SZ(LightParameter); SZ(LightSample); SZ(MaterialParameter);
    OFF(LightParameter, position); OFF(LightParameter, normal); OFF(LightParameter, emission); OFF(LightParameter, u); OFF(LightParameter, v);
    OFF(LightParameter, direction); OFF(LightParameter, lightType); OFF(LightParameter, area); OFF(LightParameter, radius);
    OFF(LightParameter, divBase); OFF(LightParameter, divLevel); OFF(LightParameter, id);
    kv("LightType::SPHERE", (double)SPHERE); kv("LightType::QUAD", (double)QUAD); kv("LightType::DIRECTION", (double)DIRECTION); kv("LightType::ENV", (double)ENV);
    {
        MaterialParameter p;   // what a `material { }` block of a .scene file starts from (sceneLoader.cpp:77-126)
        kv("MaterialParameter().color.x", p.color.x); kv("MaterialParameter().color.y", p.color.y); kv("MaterialParameter().color.z", p.color.z);
        kv("MaterialParameter().emission.x", p.emission.x); kv("MaterialParameter().metallic", p.metallic);
        kv("MaterialParameter().subsurface", p.subsurface); kv("MaterialParameter().specular", p.specular);
        kv("MaterialParameter().roughness", p.roughness); kv("MaterialParameter().specularTint", p.specularTint);
        kv("MaterialParameter().anisotropic", p.anisotropic); kv("MaterialParameter().sheen", p.sheen);
        kv("MaterialParameter().sheenTint", p.sheenTint); kv("MaterialParameter().clearcoat", p.clearcoat);
        kv("MaterialParameter().clearcoatGloss", p.clearcoatGloss); kv("MaterialParameter().brdf", (double)p.brdf);
        kv("MaterialParameter().albedoID", (double)p.albedoID, true);
    }
    g_json += "}";
    return g_json.c_str();
}
