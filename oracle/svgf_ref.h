/*
 * oracle/svgf_ref.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar C restatement of the reference's SVGF denoiser (KatanaMajesty/Nebulae):
 *   assets/shaders/svgf_temporal.hlsl:24-68      -> svgf_ref_temporal
 *   assets/shaders/svgf_atrous.hlsl:29-85        -> svgf_ref_atrous
 *   assets/shaders/svgf_common.hlsli:4-35        -> weights / luminance
 *   assets/shaders/octahedron_encoding.hlsli:27-34 -> oct16 unpack
 *   src/SVGFDenoiser.cpp:39-64,66-131,133-203    -> frame state machine
 *   src/DeferredRenderer.cpp:133-146,593-614     -> dynamic-scene / reset policy
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
 * this path and cannot be built or run here (Win32 +
 * D3D12 + DXC + closed NVIDIA DLLs).  This restatement is cross-checked against
 * an independent numpy restatement (oracle/svgf_np.py) written from the HLSL text.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything in oracle/.  The product (nebulae_amd/, libnebulae_hip.so) never does.
 *
 * Plane formats (row-major, pitch == width), identical to the reference's
 * DXGI formats (src/SVGFDenoiser.h:160-168):
 *   radiance  float[4]/px   R32G32B32A32_FLOAT (alpha carried, unused)
 *   normal    uint16[4]/px  R16G16B16A16_FLOAT (.xy oct geom N, .zw oct shading N)
 *   depth     uint32/px     R24G8: bits 0..23 D24_UNORM, bits 24..31 stencil
 *   moments   uint16[2]/px  R16G16_FLOAT  (<Y>, <Y^2>)
 *   variance  uint16/px     R16_FLOAT
 */
#ifndef NEB_ORACLE_SVGF_REF_H
#define NEB_ORACLE_SVGF_REF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/SVGFDenoiser.h:76-92 (defaults) */
typedef struct svgf_ref_params {
    float depthSigma;  /* 0.002 */
    float alpha;       /* 0.9   */
    float varianceEps; /* 1e-4  */
    float phiColor;    /* 4/255 */
    float phiNormal;   /* 128   */
    float phiDepth;    /* 0.002 */
} svgf_ref_params;

void svgf_ref_default_params(svgf_ref_params* p);

/* fp16 <-> fp32, round-to-nearest-even, denormals preserved (D3D typed-UAV store rule). */
uint16_t svgf_ref_f32_to_f16(float f);
float svgf_ref_f16_to_f32(uint16_t h);

/* Rows [row_begin,row_end) of the image are processed; columns/rows beyond
 * (W/8)*8, (H/8)*8 are left untouched (Dispatch(W/8,H/8) floor, SVGFDenoiser.cpp:116,185). */
void svgf_ref_temporal(int W, int H, int row_begin, int row_end,
                       float* radiance_cur, const float* radiance_hist,
                       const uint32_t* depth_cur, const uint32_t* depth_hist,
                       const uint16_t* normal_cur, const uint16_t* normal_hist,
                       const uint16_t* moments_hist, uint16_t* moments_cur,
                       uint16_t* variance, const svgf_ref_params* p);

void svgf_ref_atrous(int W, int H, int row_begin, int row_end,
                     const float* radiance_src, float* radiance_dst,
                     const uint16_t* variance, const uint32_t* depth_cur,
                     const uint16_t* normal_cur, int step, const svgf_ref_params* p);

/* Whole denoiser state, mirroring the SVGFDenoiser resource set. */
typedef struct svgf_ref_state {
    int W, H, levels;
    float* radiance[2];
    uint16_t* normal[2];
    uint32_t* depth[2];
    uint16_t* moments[2];
    uint16_t* variance;
    float* scratch; /* third radiance plane (the reference's unused denoisedOutput) */
    int cur, hist;
    svgf_ref_params params;
    int threads; /* row-parallel worker threads for the cpu_baseline leg (1 = scalar) */
} svgf_ref_state;

svgf_ref_state* svgf_ref_create(int W, int H, int levels);
void svgf_ref_destroy(svgf_ref_state* s);
void svgf_ref_begin_frame(svgf_ref_state* s, uint32_t frame_index); /* SVGFDenoiser.cpp:39-43 */
void svgf_ref_reset_history(svgf_ref_state* s);                     /* SVGFDenoiser.cpp:49-64 */
void svgf_ref_temporal_pass(svgf_ref_state* s);                     /* SVGFDenoiser.cpp:66-131 */
void svgf_ref_atrous_pass(svgf_ref_state* s);                       /* SVGFDenoiser.cpp:133-203 */

#ifdef __cplusplus
}
#endif
#endif
