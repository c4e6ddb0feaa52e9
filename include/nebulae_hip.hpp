// nebulae_hip.hpp -- C++ host-side mirror of the reference classes over the C ABI (nebulae_hip.h).
//
// Same member names and argument meaning as Neb::SVGFDenoiser (src/SVGFDenoiser.h:11-93) and the GI
// methods of Neb::DeferredRenderer (src/DeferredRenderer.h:50-81); ID3D12GraphicsCommandList4* becomes a
// HIP stream.  Error behaviour follows the reference: failures throw (the reference throws HrException
// through ThrowIfFailed/ThrowIfFalse, src/nri/stdafx.h:44-98); here the exception carries neb_last_error().
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>

#include "nebulae_hip.h"

namespace Neb
{

    struct NebException : std::runtime_error
    {
        int Status;
        NebException(int status, const std::string& what) : std::runtime_error(what), Status(status) {}
    };

    inline void ThrowIfFailed(neb_ctx* ctx, int status, const char* what)
    {
        if (status != NEB_OK)
            throw NebException(status, std::string(what) + ": " + neb_last_error(ctx));
    }

    class SVGFDenoiser
    {
    public:
        SVGFDenoiser() = default;
        SVGFDenoiser(const SVGFDenoiser&) = delete;
        SVGFDenoiser& operator=(const SVGFDenoiser&) = delete;
        ~SVGFDenoiser() { neb_destroy(m_ctx); }

        bool IsInitialized() const { return m_ctx != nullptr; }

        // SVGFDenoiser::Init (src/SVGFDenoiser.cpp:14-26).  rowBegin/rowEnd select a multi-GPU row strip.
        bool Init(uint32_t width, uint32_t height, uint32_t numAtrousPasses = NumAtrousPasses, int device = 0,
                  uint32_t rowBegin = 0, uint32_t rowEnd = 0)
        {
            if (IsInitialized())
                throw NebException(NEB_ERR_STATE, "SVGFDenoiser::Init: already initialised");
            neb_create_info info{device, width, height, rowBegin, rowEnd, numAtrousPasses};
            ThrowIfFailed(nullptr, neb_create(&info, &m_ctx), "neb_create");
            // This class orders all its work on the planes through neb_* calls, so it opts into the held-back temporal pass:
            // SubmitTemporalAccumulation + SubmitATrousComputeWavelet back to back run as one fused chain (nebulae_hip.h).
            // A host that also touches cached plane pointers with raw HIP calls between the two: SetOption("svgf_fuse", 0).
            ThrowIfFailed(m_ctx, neb_set_option(m_ctx, "svgf_fuse", 1), "neb_set_option");
            return true;
        }
        // The reference returns false on success (src/SVGFDenoiser.cpp:36) and its caller throws on it; fixed here.
        bool Resize(uint32_t width, uint32_t height)
        {
            ThrowIfFailed(m_ctx, neb_resize(m_ctx, width, height), "neb_resize");
            return true;
        }

        void BeginFrame(uint32_t frameIndex) { ThrowIfFailed(m_ctx, neb_begin_frame(m_ctx, frameIndex), "neb_begin_frame"); }
        void EndFrame() { ThrowIfFailed(m_ctx, neb_end_frame(m_ctx), "neb_end_frame"); }
        uint32_t GetCurrentResourceIndex() const { return (uint32_t)neb_current_index(m_ctx); }
        uint32_t GetHistoryResourceIndex() const { return (uint32_t)neb_history_index(m_ctx); }

        // One accessor instead of the ~25 descriptor getters: device pointer of a plane (borrowed).
        void* GetPlane(neb_plane plane, int slot = NEB_SLOT_CURRENT, size_t* pitchBytes = nullptr, uint32_t* rows = nullptr)
        {
            void* p = nullptr;
            ThrowIfFailed(m_ctx, neb_get_plane(m_ctx, plane, slot, &p, pitchBytes, rows), "neb_get_plane");
            return p;
        }
        void* GetCurrentRadianceTexture() { return GetPlane(NEB_PLANE_RADIANCE, NEB_SLOT_CURRENT); }
        void* GetHistoryRadianceTexture() { return GetPlane(NEB_PLANE_RADIANCE, NEB_SLOT_HISTORY); }

        void ResetHistory(neb_stream commandList) { ThrowIfFailed(m_ctx, neb_svgf_reset_history(m_ctx, commandList), "neb_svgf_reset_history"); }
        void SubmitTemporalAccumulation(neb_stream commandList) { ThrowIfFailed(m_ctx, neb_svgf_temporal(m_ctx, commandList), "neb_svgf_temporal"); }
        // (called right after SubmitTemporalAccumulation on the same command list -- DeferredRenderer::SubmitCommandsSVGFDenoising's
        // order -- the two run as one fused chain: see neb_svgf_atrous in nebulae_hip.h)
        void SubmitATrousComputeWavelet(neb_stream commandList) { ThrowIfFailed(m_ctx, neb_svgf_atrous(m_ctx, commandList), "neb_svgf_atrous"); }
        // Both passes as one explicit call (DeferredRenderer::SubmitCommandsSVGFDenoising, src/DeferredRenderer.cpp:610-611)
        void SubmitDenoising(neb_stream commandList) { ThrowIfFailed(m_ctx, neb_svgf_denoise(m_ctx, commandList), "neb_svgf_denoise"); }
        // Implementation knobs without a reference counterpart ("svgf_fuse", "svgf_profile", "atrous_variant", the "gi_*" options)
        void SetOption(const char* key, int value) { ThrowIfFailed(m_ctx, neb_set_option(m_ctx, key, value), "neb_set_option"); }
        // Durations (us) of the kernels of the last SubmitATrousComputeWavelet chain, after SetOption("svgf_profile", 1); returns how many
        uint32_t LevelTimes(float* outMicroseconds, uint32_t capacity)
        {
            uint32_t n = 0;
            ThrowIfFailed(m_ctx, neb_svgf_level_times(m_ctx, outMicroseconds, capacity, &n), "neb_svgf_level_times");
            return n;
        }

        // GetTemporalConstants()/GetATrousConstants(): one POD with the six tunables (SVGFDenoiser.h:76-92).
        neb_svgf_params GetConstants() const
        {
            neb_svgf_params p;
            ThrowIfFailed(m_ctx, neb_svgf_get_params(m_ctx, &p), "neb_svgf_get_params");
            return p;
        }
        void SetConstants(const neb_svgf_params& p) { ThrowIfFailed(m_ctx, neb_svgf_set_params(m_ctx, &p), "neb_svgf_set_params"); }

        neb_ctx* Context() const { return m_ctx; }
        static constexpr uint32_t NumAtrousPasses = 4; // src/SVGFDenoiser.h:199

    private:
        neb_ctx* m_ctx = nullptr;
    };

    // The GI methods of DeferredRenderer that sit on the hot path (src/DeferredRenderer.h:79, .cpp:396-591,978-1030).
    class GIPathtracer
    {
    public:
        explicit GIPathtracer(SVGFDenoiser& svgf) : m_svgf(svgf) {}
        void InitPathtracerScene(const neb_geometry_desc* geoms, uint32_t nGeoms, const neb_material_desc* mats, uint32_t nMats,
                                 const neb_texture_desc* texs, uint32_t nTexs)
        {
            ThrowIfFailed(m_svgf.Context(), neb_gi_set_scene(m_svgf.Context(), geoms, nGeoms, mats, nMats, texs, nTexs), "neb_gi_set_scene");
        }
        void InitRTAccelerationStructures(neb_stream commandList)
        {
            ThrowIfFailed(m_svgf.Context(), neb_gi_build_bvh(m_svgf.Context(), commandList), "neb_gi_build_bvh");
        }
        void SubmitCommandsGIPathtrace(const neb_gi_constants& globalConstants, neb_stream commandList)
        {
            ThrowIfFailed(m_svgf.Context(), neb_gi_trace(m_svgf.Context(), &globalConstants, commandList), "neb_gi_trace");
        }

        // The same dispatch in two command lists, for a renderer that keeps frames in flight (src/nri/Swapchain.h:15): the first half (ray
        // generation + closest-hit walk) touches only the G-buffer and GI records and may be recorded for frame f + 1 on another stream while
        // the second half of frame f (shade + shadow passes, adds into the radiance target) and its SVGF passes still execute.
        void SubmitCommandsGIPathtraceBegin(const neb_gi_constants& globalConstants, uint32_t row0, uint32_t row1, neb_stream commandList)
        {
            ThrowIfFailed(m_svgf.Context(), neb_gi_trace_begin(m_svgf.Context(), &globalConstants, row0, row1, commandList), "neb_gi_trace_begin");
        }
        void SubmitCommandsGIPathtraceFinish(neb_stream commandList, void* afterShadeEvent = nullptr)
        {
            ThrowIfFailed(m_svgf.Context(), neb_gi_trace_finish(m_svgf.Context(), commandList, afterShadeEvent), "neb_gi_trace_finish");
        }

        // the acceleration structure is built on the device; these report what came out of it
        uint32_t BvhDepth() const
        {
            uint32_t d = 0;
            ThrowIfFailed(m_svgf.Context(), neb_gi_bvh_depth(m_svgf.Context(), &d), "neb_gi_bvh_depth");
            return d;
        }

        // The sun-visibility table (no reference counterpart: the reference's shadow rays go to TraceRay, assets/shaders/pathtracer.hlsl:567-569) is built
        // and kept up to date by SubmitCommandsGIPathtrace itself; these report what it does.
        struct SunTableStats
        {
            uint64_t sidesLitPlus = 0, sidesLitMinus = 0, raysAnswered = 0, builds = 0;
            float lastBuildMs = 0.0f;      // 0 when no table has been built
            int shadowTailMode = -1;       // 0: the rays the table leaves go through the compacted lists, 1: through the sorted pass, -1: not measured yet for this table
        };
        SunTableStats GetSunTableStats(neb_stream commandList) const
        {
            SunTableStats s;
            uint64_t v[4] = {0, 0, 0, 0};
            ThrowIfFailed(m_svgf.Context(), neb_gi_sun_table_stats(m_svgf.Context(), v, commandList), "neb_gi_sun_table_stats");
            s.sidesLitPlus = v[0], s.sidesLitMinus = v[1], s.raysAnswered = v[2], s.builds = v[3];
            if (s.builds)
                ThrowIfFailed(m_svgf.Context(), neb_gi_sun_table_build_ms(m_svgf.Context(), &s.lastBuildMs), "neb_gi_sun_table_build_ms");
            ThrowIfFailed(m_svgf.Context(), neb_gi_shadow_tail_mode(m_svgf.Context(), &s.shadowTailMode, nullptr), "neb_gi_shadow_tail_mode");
            return s;
        }

    private:
        SVGFDenoiser& m_svgf;
    };

    // Halo exchange between the row strips of a multi-GPU frame (no reference counterpart: Nebulae is single-GPU).
    // One communicator per strip group; Exchange() enqueues one RCCL group of sends / receives on the caller's stream.
    // Threading: with one process or thread per GPU nothing else is needed.  ONE thread that drives several GPUs must put
    // the Init() calls of all its ranks, and the Exchange() calls of all its strips for one exchange, between
    // GroupBegin() and GroupEnd() (RCCL's rule for one thread with several devices; otherwise Init blocks for ever).
    class StripExchange
    {
    public:
        static void GroupBegin() { Check(neb_strips_group_begin(), "neb_strips_group_begin"); }
        static void GroupEnd() { Check(neb_strips_group_end(), "neb_strips_group_end"); }
        StripExchange() = default;
        StripExchange(const StripExchange&) = delete;
        StripExchange& operator=(const StripExchange&) = delete;
        ~StripExchange() { neb_strips_comm_destroy(m_comm); }
        static void MakeUniqueId(void* id128) { Check(neb_strips_unique_id(id128), "neb_strips_unique_id"); }
        void Init(int device, int numRanks, int rank, const void* id128)
        {
            Check(neb_strips_comm_create(device, numRanks, rank, id128, &m_comm), "neb_strips_comm_create");
        }
        void Exchange(SVGFDenoiser& strip, const neb_halo_plane* planes, uint32_t nPlanes, const neb_halo_swap* swaps, uint32_t nSwaps, neb_stream commandList)
        {
            ThrowIfFailed(strip.Context(), neb_strips_exchange(strip.Context(), m_comm, planes, nPlanes, swaps, nSwaps, commandList), "neb_strips_exchange");
        }
        // One call per strip frame (round 5): the GI dispatch on the strip's rows (constants == nullptr: none), the temporal pass, the halo
        // exchange(s) of the plan's scheme over this communicator and the a-trous levels on the row ranges the scheme prescribes.
        void SubmitStripFrame(SVGFDenoiser& strip, const neb_gi_constants* constants, const neb_strip_plan& plan, neb_stream commandList)
        {
            ThrowIfFailed(strip.Context(), neb_strip_frame(strip.Context(), constants, m_comm, &plan, commandList), "neb_strip_frame");
        }
        // The same for a host that drives all strips from ONE thread and needs no communicator (the neighbouring strips' contexts instead):
        // Begin for every strip, then Finish for every strip.
        static void SubmitStripFrameBegin(SVGFDenoiser& strip, const neb_gi_constants* constants, const neb_strip_plan& plan, SVGFDenoiser* up, SVGFDenoiser* down,
                                          neb_stream commandList)
        {
            const neb_strip_peers peers{up ? up->Context() : nullptr, down ? down->Context() : nullptr};
            ThrowIfFailed(strip.Context(), neb_strip_frame_begin(strip.Context(), constants, &plan, &peers, commandList), "neb_strip_frame_begin");
        }
        static void SubmitStripFrameFinish(SVGFDenoiser& strip, const neb_strip_plan& plan, SVGFDenoiser* up, SVGFDenoiser* down, neb_stream commandList)
        {
            const neb_strip_peers peers{up ? up->Context() : nullptr, down ? down->Context() : nullptr};
            ThrowIfFailed(strip.Context(), neb_strip_frame_finish(strip.Context(), nullptr, &plan, &peers, commandList), "neb_strip_frame_finish");
        }

    private:
        static void Check(int status, const char* what)
        {
            if (status != NEB_OK)
                throw NebException(status, std::string(what) + ": " + neb_strips_last_error());
        }
        void* m_comm = nullptr;
    };

} // namespace Neb
