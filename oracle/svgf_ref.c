/*
 * oracle/svgf_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED
 * (see svgf_ref.h).  Scalar C restatement of the reference SVGF passes.
 * Build: gcc -O2 -ffp-contract=off -fopenmp (oracle/Makefile).
 */
#include "svgf_ref.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void svgf_ref_default_params(svgf_ref_params* p)
{
    /* src/SVGFDenoiser.h:79-81,89-91 */
    p->depthSigma = 0.002f;
    p->alpha = 0.9f;
    p->varianceEps = 1e-4f;
    p->phiColor = 4.0f / 255.0f;
    p->phiNormal = 128.0f;
    p->phiDepth = 0.002f;
}

/* ---- fp16 storage model (typed UAV store to R16(G16)_FLOAT: RNE, denormals kept) ---- */
uint16_t svgf_ref_f32_to_f16(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u));
    if (ax >= 0x477ff000u) /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    if (ax < 0x33000001u) /* <= 2^-25 rounds to zero (tie to even -> 0) */
        return (uint16_t)sign;
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u; /* 24-bit significand */
    int shift;                                 /* bits to drop */
    uint32_t base;
    if (e < -14) { /* denormal half: value = m * 2^(e-23); unit = 2^-24 */
        shift = -e - 1; /* 13 + (-14 - e) */
        base = 0;
    } else {
        shift = 13;
        base = (uint32_t)(e + 15) << 10;
        m &= 0x7fffffu;
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u)))
        q++;
    return (uint16_t)(sign | (base + q)); /* carry into exponent is correct by construction */
}

float svgf_ref_f16_to_f32(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else { /* denormal */
            int s = 0;
            while (!(m & 0x400u)) {
                m <<= 1;
                s++;
            }
            m &= 0x3ffu;
            x = sign | ((uint32_t)(127 - 15 - s + 1) << 23) | (m << 13);
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

/* ---- HLSL helper restatements ---- */
static inline float lerpf(float a, float b, float t) { return a + t * (b - a); }
static inline float saturatef(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }

/* svgf_common.hlsli:32-35 */
static inline float luminance(const float* c)
{
    return c[0] * 0.2126f + c[1] * 0.7152f + c[2] * 0.0722f;
}

/* octahedron_encoding.hlsli:8-11,27-34 */
static inline void oct16_fast_unpack(float ex, float ey, float* n)
{
    float vx = ex, vy = ey;
    float vz = 1.0f - fabsf(ex) - fabsf(ey);
    if (vz < 0.0f) {
        float sx = (vx > 0.0f) ? 1.0f : -1.0f;
        float sy = (vy > 0.0f) ? 1.0f : -1.0f;
        float nx = (1.0f - fabsf(vy)) * sx;
        float ny = (1.0f - fabsf(vx)) * sy;
        vx = nx;
        vy = ny;
    }
    float len = sqrtf(vx * vx + vy * vy + vz * vz);
    n[0] = vx / len;
    n[1] = vy / len;
    n[2] = vz / len;
}

/* Texture2D<float> view of R24_UNORM_X8_TYPELESS (SVGFDenoiser.h:162) */
static inline float depth_unorm24(uint32_t d) { return (float)(d & 0xffffffu) / 16777215.0f; }

static inline void shading_normal(const uint16_t* n4, float* n)
{
    /* .zw = shading normal (svgf_temporal.hlsl:36, svgf_atrous.hlsl:44) */
    oct16_fast_unpack(svgf_ref_f16_to_f32(n4[2]), svgf_ref_f16_to_f32(n4[3]), n);
}

/* ---- svgf_temporal.hlsl:24-68 ---- */
void svgf_ref_temporal(int W, int H, int row_begin, int row_end,
                       float* radiance_cur, const float* radiance_hist,
                       const uint32_t* depth_cur, const uint32_t* depth_hist,
                       const uint16_t* normal_cur, const uint16_t* normal_hist,
                       const uint16_t* moments_hist, uint16_t* moments_cur,
                       uint16_t* variance, const svgf_ref_params* p)
{
    const int Wd = (W / 8) * 8, Hd = (H / 8) * 8; /* Dispatch(W/8,H/8): SVGFDenoiser.cpp:116 */
    if (row_end > Hd)
        row_end = Hd;
    for (int y = row_begin; y < row_end; ++y) {
        for (int x = 0; x < Wd; ++x) {
            size_t i = (size_t)y * W + x;
            const float* Ccurr = radiance_cur + 4 * i;
            const float* Chist = radiance_hist + 4 * i;
            float Dcurr = depth_unorm24(depth_cur[i]);
            float Dhist = depth_unorm24(depth_hist[i]);
            float Ncurr[3], Nhist[3];
            shading_normal(normal_cur + 4 * i, Ncurr);
            shading_normal(normal_hist + 4 * i, Nhist);
            float Mh0 = svgf_ref_f16_to_f32(moments_hist[2 * i + 0]);
            float Mh1 = svgf_ref_f16_to_f32(moments_hist[2 * i + 1]);

            /* svgf_common.hlsli:11-15 */
            float dz = fabsf(Dcurr - Dhist);
            float wDepth = expf(-dz * dz / (2.0f * p->depthSigma * p->depthSigma));
            /* svgf_common.hlsli:4-7 */
            float wNormal = saturatef(Ncurr[0] * Nhist[0] + Ncurr[1] * Nhist[1] + Ncurr[2] * Nhist[2]);
            float w = wDepth * wNormal;

            float alpha = lerpf(1.0f, p->alpha, w); /* :51 (quirk 2: w->0 keeps history) */

            float Ycurr = luminance(Ccurr);
            float Yaccum = lerpf(Ycurr, Mh0, alpha);
            float Y2accum = lerpf(Ycurr * Ycurr, Mh1, alpha);
            float var = fmaxf(Y2accum - Yaccum * Yaccum, p->varianceEps);

            float* out = radiance_cur + 4 * i;
            out[0] = lerpf(Ccurr[0], Chist[0], alpha);
            out[1] = lerpf(Ccurr[1], Chist[1], alpha);
            out[2] = lerpf(Ccurr[2], Chist[2], alpha);
            /* out[3]: alpha channel carried unchanged (RWTexture2D<float3> store) */
            moments_cur[2 * i + 0] = svgf_ref_f32_to_f16(Yaccum);
            moments_cur[2 * i + 1] = svgf_ref_f32_to_f16(Y2accum);
            variance[i] = svgf_ref_f32_to_f16(var);
        }
    }
}

/* ---- svgf_atrous.hlsl:29-85 ---- */
void svgf_ref_atrous(int W, int H, int row_begin, int row_end,
                     const float* radiance_src, float* radiance_dst,
                     const uint16_t* variance, const uint32_t* depth_cur,
                     const uint16_t* normal_cur, int step, const svgf_ref_params* p)
{
    static const float K[3] = {1.0f / 16.0f, 1.0f / 4.0f, 3.0f / 8.0f}; /* :35 indexed by abs(d) (quirk 1) */
    const int Wd = (W / 8) * 8, Hd = (H / 8) * 8;                      /* SVGFDenoiser.cpp:185 */
    const float fstep = (float)step;
    if (row_end > Hd)
        row_end = Hd;
    for (int y = row_begin; y < row_end; ++y) {
        for (int x = 0; x < Wd; ++x) {
            size_t i = (size_t)y * W + x;
            const float* c0 = radiance_src + 4 * i;
            float lum0 = luminance(c0);
            float var = svgf_ref_f16_to_f32(variance[i]);
            float varScale = p->phiColor * sqrtf(fmaxf(var, 1e-8f));
            float z0 = depth_unorm24(depth_cur[i]);
            float n0[3];
            shading_normal(normal_cur + 4 * i, n0);

            float sumC[3] = {0.0f, 0.0f, 0.0f};
            float sumW = 0.0f;
            for (int dy = -2; dy <= 2; ++dy) {
                int vy = dy * step;
                float Ky = K[abs(dy)];
                for (int dx = -2; dx <= 2; ++dx) {
                    int vx = dx * step;
                    float Kx = K[abs(dx)];
                    int qx = x + vx, qy = y + vy;
                    qx = qx < 0 ? 0 : (qx > W - 1 ? W - 1 : qx); /* :65 clamp to image */
                    qy = qy < 0 ? 0 : (qy > H - 1 ? H - 1 : qy);
                    size_t q = (size_t)qy * W + qx;
                    const float* c = radiance_src + 4 * q;
                    float lum = luminance(c);
                    float z = depth_unorm24(depth_cur[q]);
                    float n[3];
                    shading_normal(normal_cur + 4 * q, n);

                    float wz = expf(-fabsf(z0 - z) / (p->phiDepth * fstep));
                    float d = n0[0] * n[0] + n0[1] * n[1] + n0[2] * n[2];
                    float wn = powf(fmaxf(0.0f, d), p->phiNormal);
                    float wl = expf(-fabsf(lum0 - lum) / fmaxf(varScale, 1e-6f));
                    float w = Kx * Ky * wz * wn * wl;
                    sumC[0] += w * c[0];
                    sumC[1] += w * c[1];
                    sumC[2] += w * c[2];
                    sumW += w;
                }
            }
            float inv = fmaxf(sumW, 1e-4f);
            float* o = radiance_dst + 4 * i;
            o[0] = sumC[0] / inv;
            o[1] = sumC[1] / inv;
            o[2] = sumC[2] / inv;
            o[3] = c0[3]; /* alpha carried from the centre texel */
        }
    }
}

/* ---- state machine: src/SVGFDenoiser.cpp ---- */
svgf_ref_state* svgf_ref_create(int W, int H, int levels)
{
    svgf_ref_state* s = (svgf_ref_state*)calloc(1, sizeof(*s));
    size_t n = (size_t)W * H;
    s->W = W;
    s->H = H;
    s->levels = levels;
    for (int k = 0; k < 2; ++k) {
        /* quirk 8: the build defines zero-initialised history */
        s->radiance[k] = (float*)calloc(n * 4, sizeof(float));
        s->normal[k] = (uint16_t*)calloc(n * 4, sizeof(uint16_t));
        s->depth[k] = (uint32_t*)calloc(n, sizeof(uint32_t));
        s->moments[k] = (uint16_t*)calloc(n * 2, sizeof(uint16_t));
    }
    s->variance = (uint16_t*)calloc(n, sizeof(uint16_t));
    s->scratch = (float*)calloc(n * 4, sizeof(float));
    s->cur = 0;
    s->hist = 1;
    s->threads = 1;
    svgf_ref_default_params(&s->params);
    return s;
}

void svgf_ref_destroy(svgf_ref_state* s)
{
    if (!s)
        return;
    for (int k = 0; k < 2; ++k) {
        free(s->radiance[k]);
        free(s->normal[k]);
        free(s->depth[k]);
        free(s->moments[k]);
    }
    free(s->variance);
    free(s->scratch);
    free(s);
}

void svgf_ref_begin_frame(svgf_ref_state* s, uint32_t frame_index)
{
    s->cur = (int)(frame_index & 1u);
    s->hist = s->cur ^ 1;
}

void svgf_ref_reset_history(svgf_ref_state* s)
{
    /* CopyResource(hist <- cur); moments/variance are NOT reset (SVGFDenoiser.cpp:57) */
    memcpy(s->radiance[s->hist], s->radiance[s->cur], (size_t)s->W * s->H * 4 * sizeof(float));
}

void svgf_ref_temporal_pass(svgf_ref_state* s)
{
    int H = s->H;
#pragma omp parallel for schedule(static) num_threads(s->threads)
    for (int y = 0; y < H; y += 8)
        svgf_ref_temporal(s->W, s->H, y, (y + 8 < H) ? y + 8 : H, s->radiance[s->cur], s->radiance[s->hist],
                          s->depth[s->cur], s->depth[s->hist], s->normal[s->cur], s->normal[s->hist],
                          s->moments[s->hist], s->moments[s->cur], s->variance, &s->params);
}

void svgf_ref_atrous_pass(svgf_ref_state* s)
{
    /* SVGFDenoiser.cpp:133-203: src = cur, dst = hist, swap per level, step <<= 1.
     * The reference asserts an even level count (:197).  For an odd count the chain
     * cur -> hist -> scratch -> hist -> ... -> cur runs through the third radiance
     * plane so that the final image still lands in radiance[cur];
     * arithmetic per level is unchanged. */
    int L = s->levels, H = s->H;
    float* cur = s->radiance[s->cur];
    float* hist = s->radiance[s->hist];
    const float* src = cur;
    int step = 1;
    for (int i = 0; i < L; ++i) {
        float* dst;
        if (L == 1)
            dst = s->scratch; /* copied back below */
        else if (i == L - 1)
            dst = cur; /* last level always lands in cur */
        else if ((L & 1) == 0)
            dst = (src == cur) ? hist : cur;
        else
            dst = (i == 0) ? hist : ((src == hist) ? s->scratch : hist);
#pragma omp parallel for schedule(static) num_threads(s->threads)
        for (int y = 0; y < H; y += 8)
            svgf_ref_atrous(s->W, s->H, y, (y + 8 < H) ? y + 8 : H, src, dst, s->variance, s->depth[s->cur],
                            s->normal[s->cur], step, &s->params);
        src = dst;
        step <<= 1;
    }
    if (L == 1)
        memcpy(cur, s->scratch, (size_t)s->W * s->H * 4 * sizeof(float));
}
