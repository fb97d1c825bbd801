// EXPERIMENT (not part of libmade_hip.so): what does a cross-stream dependency cost on the device?
// A chain of N dependent tiny kernels alternating between two streams, the hop made (a) with hipEventRecord + hipStreamWaitEvent (what the launch
// tape replays), (b) with hipStreamWriteValue32 on the producing stream + hipStreamWaitValue32 on the consuming one (a value in signal memory the
// command processor polls), against (c) the same chain on one stream.  hipcc --offload-arch=gfx950 -O3 -o tools/probes/stream_value_probe tools/probes/stream_value_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void tiny(float* x) { if (threadIdx.x == 0) x[0] += 1.f; }
int main() {
    const int N = 200, REP = 5;
    float* x; CK(hipMalloc(&x, 256)); CK(hipMemset(x, 0, 256));
    hipStream_t s[2]; CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    hipEvent_t ev[N]; for (int i = 0; i < N; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    uint32_t* flag = nullptr;
    hipError_t fe = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
    if (fe != hipSuccess) { printf("hipMallocSignalMemory: %s\n", hipGetErrorString(fe)); flag = nullptr; }
    else *flag = 0;
    uint32_t epoch = 0;
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 2 && !flag) continue;
        double best = 1e30;
        for (int rep = 0; rep < REP + 1; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            int cur = 0;
            for (int i = 0; i < N; ++i) {
                if (mode == 1) {
                    CK(hipEventRecord(ev[i], s[cur])); cur ^= 1; CK(hipStreamWaitEvent(s[cur], ev[i], 0));
                } else if (mode == 2) {
                    ++epoch;
                    CK(hipStreamWriteValue32(s[cur], flag, epoch, 0)); cur ^= 1;
                    CK(hipStreamWaitValue32(s[cur], flag, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu));
                }
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[cur], x);
            }
            CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1]));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (rep > 0 && us < best) best = us;
        }
        printf("%s: %.2f us per kernel of the chain\n", mode == 0 ? "one stream" : (mode == 1 ? "two streams, event record + stream wait" : "two streams, write value + wait value"), best);
    }
    {   // a chain on stream 0 whose every kernel first waits for an event of stream 1 that completed long ago
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[1], x + 8);
        CK(hipEventRecord(ev[0], s[1])); CK(hipDeviceSynchronize());
        double best = 1e30;
        for (int rep = 0; rep < REP + 1; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) { CK(hipStreamWaitEvent(s[0], ev[0], 0)); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[0], x); }
            CK(hipStreamSynchronize(s[0]));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (rep > 0 && us < best) best = us;
        }
        printf("one stream, every kernel behind a wait for a long-completed event of the other: %.2f us per kernel\n", best);
        // ... and with a fresh record on the (idle) other stream in front of every wait: the hop's cost without any work to wait for
        best = 1e30;
        for (int rep = 0; rep < REP + 1; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) { CK(hipEventRecord(ev[i], s[1])); CK(hipStreamWaitEvent(s[0], ev[i], 0)); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[0], x); }
            CK(hipStreamSynchronize(s[0]));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (rep > 0 && us < best) best = us;
        }
        printf("one stream, every kernel behind record(idle other stream) + wait: %.2f us per kernel\n", best);
    }
    float h; CK(hipMemcpy(&h, x, 4, hipMemcpyDeviceToHost));
    printf("kernels run: %.0f\n", h);
    return 0;
}
// (appended: what a wait costs when the event it waits for completed long ago -- the common case of the step's 20 waits)
