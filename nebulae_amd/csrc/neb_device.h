// neb_device.h -- device-side helpers shared by the gfx950 kernels.
// Every fused multiply-add is written out (fmaf): the SVGF kernels are compiled with floating-point
// contraction off so that a pixel's result cannot depend on which unrolled copy of the code (tile slot,
// staging slot) produced it -- the multi-GPU path promises N-strip == 1-strip bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace neb {

constexpr float kLog2e = 1.44269504088896340736f;

__device__ __forceinline__ float half_bits_to_float(uint32_t h)
{
    return (float)__builtin_bit_cast(_Float16, (unsigned short)h);
}

// Typed-UAV store rule of R16(G16)_FLOAT: round-to-nearest-even, fp16 denormals kept
// (v_cvt_f16_f32 under the default gfx950 float mode).
__device__ __forceinline__ uint32_t float_to_half_bits(float f)
{
    return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)f);
}

// svgf_common.hlsli:32-35 (ITU-R BT.709)
__device__ __forceinline__ float luminance(float r, float g, float b)
{
    return fmaf(b, 0.0722f, fmaf(g, 0.7152f, r * 0.2126f));
}

// Texture2D<float> view of R24_UNORM_X8_TYPELESS (SVGFDenoiser.h:162): c / (2^24 - 1).
__device__ __forceinline__ float depth_unorm24(uint32_t d)
{
    return (float)(d & 0xffffffu) / 16777215.0f;
}

// Oct16_FastUnpack (octahedron_encoding.hlsli:27-34) of two fp16 values packed in a dword
// (low half = E.x, high half = E.y); normalize() = v * rsqrt(dot(v,v)).
__device__ __forceinline__ float3 oct16_unpack_zw(uint32_t packed)
{
    float ex = half_bits_to_float(packed & 0xffffu);
    float ey = half_bits_to_float(packed >> 16);
    float vz = 1.0f - fabsf(ex) - fabsf(ey);
    float vx = ex, vy = ey;
    if (vz < 0.0f) {
        float sx = (ex > 0.0f) ? 1.0f : -1.0f;
        float sy = (ey > 0.0f) ? 1.0f : -1.0f;
        vx = (1.0f - fabsf(ey)) * sx;
        vy = (1.0f - fabsf(ex)) * sy;
    }
    float inv = __frsqrt_rn(fmaf(vz, vz, fmaf(vy, vy, vx * vx)));
    return make_float3(vx * inv, vy * inv, vz * inv);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

} // namespace neb
