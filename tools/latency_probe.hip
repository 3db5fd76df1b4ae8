// Dependent-kernel latency probe for gfx950: what one kernel boundary costs in a stream and in a hipGraph,
// and what each dependent global-memory round trip inside a tiny kernel adds.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/latency_probe tools/latency_probe.hip && gpurun_out/latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}

// trips dependent loads: p[0] -> p[idx] -> ...
__global__ void k_chain(const int* __restrict__ p, int* __restrict__ out, int trips) {
    int i = threadIdx.x & 1;
    for (int t = 0; t < trips; ++t) i = __builtin_nontemporal_load(p + i * 64 + (threadIdx.x & 1));
    if (i == 12345) out[0] = i;
}

// writes 256 KB, next kernel reads it (producer/consumer across a boundary)
__global__ void k_prod(float* __restrict__ buf) { buf[blockIdx.x * blockDim.x + threadIdx.x] = (float)threadIdx.x; }
__global__ void k_cons(const float* __restrict__ buf, float* __restrict__ out) {
    float v = buf[blockIdx.x * blockDim.x + threadIdx.x];
    if (v == -1.f) out[0] = v;
}

template <class F>
static int time_it(const char* name, hipStream_t st, int n, int reps, F launch, bool graph) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipGraphExec_t ge = nullptr;
    if (graph) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < n; ++i) launch(i);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
    } else {
        for (int i = 0; i < n; ++i) launch(i);
        CK(hipStreamSynchronize(st));
    }
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a, st));
        if (graph) CK(hipGraphLaunch(ge, st));
        else for (int i = 0; i < n; ++i) launch(i);
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("%-44s %s  %7.3f us per kernel\n", name, graph ? "graph " : "stream", best * 1000.f / n);
    return 0;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    int* p; int* out; float* buf; float* fout;
    CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&buf, 1 << 20)); CK(hipMalloc(&fout, 64));
    const int N = 400;
    for (int graph = 0; graph < 2; ++graph) {
        time_it("empty <<<1,64>>>", st, N, 5, [&](int) { hipLaunchKernelGGL(k_empty, 1, 64, 0, st); }, graph);
        time_it("empty <<<256,256>>>", st, N, 5, [&](int) { hipLaunchKernelGGL(k_empty, 256, 256, 0, st); }, graph);
        time_it("empty <<<2048,512>>>", st, N, 5, [&](int) { hipLaunchKernelGGL(k_empty, 2048, 512, 0, st); }, graph);
        for (int trips = 1; trips <= 4; ++trips) {
            char nm[64]; snprintf(nm, 64, "chain of %d dependent loads <<<32,256>>>", trips);
            time_it(nm, st, N, 5, [&](int) { hipLaunchKernelGGL(k_chain, 32, 256, 0, st, p, out, trips); }, graph);
        }
        time_it("producer/consumer pair 256 KB (per kernel)", st, N, 5, [&](int i) {
            if (i & 1) hipLaunchKernelGGL(k_cons, 256, 256, 0, st, buf, fout);
            else hipLaunchKernelGGL(k_prod, 256, 256, 0, st, buf);
        }, graph);
    }
    return 0;
}
