// tuning only: what a captured graph of a strip frame's ~19 stream operations costs the HOST per launch, against issuing them one by one.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench_graph.hip -o tools/ubench_graph && tools/ubench_graph
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);             \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

__global__ void touch(float* p, int n, float v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        p[i] = p[i] * 0.5f + v;
}

static int issue(hipStream_t s, hipStream_t side, hipEvent_t a, hipEvent_t b, float* x, float* y, int n, int kernels, float v)
{
    // the shape of a strip frame: a run of dependent launches, a side stream joined by two events, two device-to-device copies
    for (int k = 0; k < kernels; ++k) {
        hipLaunchKernelGGL(touch, dim3((n + 255) / 256), dim3(256), 0, s, x, n, v);
        if (k == kernels / 2) {
            CHECK(hipEventRecord(a, s));
            CHECK(hipStreamWaitEvent(side, a, 0));
            CHECK(hipMemcpyAsync(y, x, 4096, hipMemcpyDeviceToDevice, side));
            CHECK(hipMemcpyAsync(y + 2048, x + 2048, 4096, hipMemcpyDeviceToDevice, side));
            CHECK(hipEventRecord(b, side));
            CHECK(hipStreamWaitEvent(s, b, 0));
        }
    }
    return 0;
}

int main()
{
    const int n = 1 << 16;
    float *x, *y;
    CHECK(hipMalloc(&x, n * sizeof(float)));
    CHECK(hipMalloc(&y, n * sizeof(float)));
    CHECK(hipMemset(x, 0, n * sizeof(float)));
    hipStream_t s, side;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t a, b;
    CHECK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
    for (int kernels : {4, 13, 26}) {
        // one by one
        const int reps = 600;
        double direct_us = 0.0;
        for (int r = 0; r < reps; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            if (issue(s, side, a, b, x, y, n, kernels, 1.0f))
                return 1;
            direct_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (r % 3 == 2)
                CHECK(hipStreamSynchronize(s)); // (drained outside the timed part: back-pressure of a full queue is not the host's cost)
        }
        CHECK(hipDeviceSynchronize());
        // captured
        hipGraph_t g;
        hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        if (issue(s, side, a, b, x, y, n, kernels, 1.0f))
            return 1;
        CHECK(hipStreamEndCapture(s, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        size_t nodes = 0;
        CHECK(hipGraphGetNodes(g, nullptr, &nodes));
        double graph_us = 0.0;
        for (int r = 0; r < reps; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            CHECK(hipGraphLaunch(ge, s));
            graph_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (r % 3 == 2)
                CHECK(hipStreamSynchronize(s));
        }
        CHECK(hipDeviceSynchronize());
        // device side: wall time of a launch sequence, both ways
        auto wall = [&](bool graph) -> double {
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < 200; ++r) {
                if (graph)
                    (void)hipGraphLaunch(ge, s);
                else
                    (void)issue(s, side, a, b, x, y, n, kernels, 1.0f);
            }
            (void)hipStreamSynchronize(s);
            return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200.0;
        };
        const double w_direct = wall(false), w_graph = wall(true);
        printf("%2d launches + 2 copies + 4 event operations (%zu graph nodes): host %.1f us one by one, %.1f us as one hipGraphLaunch; wall per sequence %.1f / %.1f us\n",
               kernels, nodes, direct_us / reps, graph_us / reps, w_direct, w_graph);
        CHECK(hipGraphExecDestroy(ge));
        CHECK(hipGraphDestroy(g));
    }
    return 0;
}
