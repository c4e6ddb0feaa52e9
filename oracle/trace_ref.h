/*
 * oracle/trace_ref.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED.
 *
 * Scalar C++ restatement of the reference's one-bounce indirect-diffuse path tracer
 * with the NVIDIA NRC calls replaced by the reference's own ENABLE_NRC=0 stubs
 * (assets/shaders/rtxgi/Nrc.hlsli:579-621) and nrcMaxPathVertices = 2:
 *   assets/shaders/pathtracer.hlsl:132-143,209-228,299-395,397-625 -> trace_ref_gi
 *   assets/shaders/brdf.hlsli:4-185, rand.hlsli:6-61, sun_disk_sampling.hlsli:45-52
 *   src/nri/GIProcessedScene.h:17-39 (geometry/material tables)   -> trace_ref_geometry/material
 *   assets/shaders/deferred_gbuffers.hlsl:36-104, src/DeferredRenderer.cpp:147-148,
 *   src/core/InspectCamera.h:31-55                                 -> trace_ref_gbuffer
 *
 * The reference's rays run on a vendor DXR driver BVH and its indirect term is produced by a
 * closed neural cache, neither of which can run here; the reference ships no tests or golden
 * vectors.  Deliberate, documented divergences (SURVEY.md quirks 9,12,13 and
 * DESIGN.md): true closest hit; correct instance transform; W x H dispatch; a Cook-Torrance term
 * whose denominator is zero evaluates to 0 instead of 0*inf = NaN.
 */
#ifndef NEB_ORACLE_TRACE_REF_H
#define NEB_ORACLE_TRACE_REF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One submesh: StaticMeshGeometryData (GIProcessedScene.h:17-31) with the bindless buffer
 * index/offset pairs collapsed into host pointers.  Attribute order: position float3,
 * normal float3, texcoord float2, tangent float4 (StaticMesh.h:14-21). */
typedef struct trace_ref_geometry {
    float surfaceToWorld[16]; /* row-major, row-vector convention: world = (p,1) * M (SimpleMath) */
    int32_t materialIndex;    /* -1 = none */
    uint32_t indexStride;     /* 2 or 4 */
    uint32_t numIndices;
    uint32_t numVertices;
    const void* indices;
    const void* attributes[4]; /* NULL = missing (ReconstructSurfaceData then fails) */
    uint32_t attributeStrides[4];
    uint32_t _pad;
} trace_ref_geometry;

/* StaticMeshMaterialData (GIProcessedScene.h:33-39) */
typedef struct trace_ref_material {
    int32_t textureIndices[3]; /* albedo, normal, roughnessMetalness; -1 = none */
    float albedo[4];
    float roughnessMetalness[2];
    uint32_t _pad;
} trace_ref_material;

typedef struct trace_ref_texture {
    const uint8_t* rgba8; /* R8G8B8A8_UNORM, 1 mip, not sRGB-decoded (GLTFSceneImporter.cpp:156) */
    uint32_t width, height;
} trace_ref_texture;

/* GlobalConstants (DeferredRenderer.h:219-238) without the NRC-only members. */
typedef struct trace_ref_constants {
    uint32_t frameIndex;
    uint32_t samplesPerPixel;
    uint32_t maxPathVertices; /* 2 = one bounce */
    float cameraWorldPos[3];
    float skyColor[3];
    float sunLightDirection[3];
    float sunLightRadiance[3];
    float sunTanHalfAngle;
    float throughputThreshold; /* unused by the shader */
} trace_ref_constants;

/* Per-pixel record of the LAST sample's bounce ray (the reference has the equivalent debug UAVs,
 * pathtracer.hlsl:41-42). */
typedef struct trace_ref_hit {
    float t;            /* < 0: miss */
    uint32_t geometry;  /* GeometryIndex() */
    uint32_t primitive; /* PrimitiveIndex() */
    uint32_t flags;     /* bit 0: shadow ray unoccluded */
} trace_ref_hit;

typedef struct trace_ref_camera {
    float eye[3];
    float target[3];
    float up[3];
    float vfov_deg, znear, zfar;
} trace_ref_camera;

typedef struct trace_ref_scene trace_ref_scene;

trace_ref_scene* trace_ref_scene_create(const trace_ref_geometry* geoms, uint32_t n_geoms, const trace_ref_material* mats,
                                        uint32_t n_mats, const trace_ref_texture* texs, uint32_t n_texs);
void trace_ref_scene_destroy(trace_ref_scene* s);
uint32_t trace_ref_scene_triangles(const trace_ref_scene* s);

/* radiance[px].rgb += mean over spp of the path's radiance (replaces NRC Resolve, DeferredRenderer.cpp:586).
 * Returns the number of rays traced (bounce rays + shadow rays). */
uint64_t trace_ref_gi(const trace_ref_scene* s, uint32_t W, uint32_t H, uint32_t row0, uint32_t row1,
                      const trace_ref_constants* c, const uint32_t* albedo_r11g11b10, const uint16_t* rough_metal,
                      const uint16_t* world_pos, const uint16_t* normal, float* radiance, trace_ref_hit* hits, int threads);

/* Primary-visibility G-buffer producer with the reference's encodings (deferred_gbuffers.hlsl:71-103). */
void trace_ref_gbuffer(const trace_ref_scene* s, uint32_t W, uint32_t H, const trace_ref_camera* cam,
                       uint32_t* albedo_r11g11b10, uint16_t* rough_metal, uint16_t* world_pos, uint16_t* normal,
                       uint32_t* depth_stencil, int threads);

/* "next" row f1: direct sun light, assets/shaders/deferred_pbr.hlsl:39-115 -- OVERWRITES radiance (alpha = 1).
 * RNG seeded with the group id as "resolution" exactly as the shader does (:82). */
uint64_t trace_ref_pbr_direct(const trace_ref_scene* s, uint32_t W, uint32_t H, const trace_ref_constants* c,
                              const uint32_t* albedo_r11g11b10, const uint16_t* rough_metal, const uint16_t* world_pos,
                              const uint16_t* normal, float* radiance, int threads);

/* "next" row f3: ACES fit of assets/shaders/tonemapping.hlsl:3-53 into R8G8B8A8_UNORM (alpha = luma). */
void trace_ref_tonemap(uint32_t W, uint32_t H, const float* radiance, uint8_t* rgba8);

/* Format helpers exported for tests. */
uint32_t trace_ref_pack_r11g11b10(const float* rgb);
void trace_ref_unpack_r11g11b10(uint32_t v, float* rgb);

#ifdef __cplusplus
}
#endif
#endif
