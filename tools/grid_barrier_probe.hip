// What does a device-wide barrier inside ONE persistent kernel cost on gfx950, next to a dependent launch in a hipGraph?
// (tuning aid: decides whether the per-layer chain of small kernels could become one persistent kernel)
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/grid_barrier_probe tools/grid_barrier_probe.hip && gpurun_out/grid_barrier_probe
// Every phase each block writes PAY bytes and reads the PAY bytes the block (b+17) % G wrote in the phase before
// (checked: a stale read is counted), so the barrier carries the cache write-back / invalidate a real chain would need.
// Spins are bounded: a barrier that does not complete sets an error flag and every wave leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Bar { unsigned cnt; unsigned pad0[31]; unsigned gen; unsigned pad1[31]; unsigned xcnt[8][32]; unsigned err; };

__device__ __forceinline__ bool spin_until(unsigned* gen, unsigned g, unsigned* err) {
    for (int i = 0; i < (1 << 22); ++i) {
        if (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= g) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

// flat: one counter that only grows (barrier g completes when it reaches G*g)
template <bool HIER>
__device__ __forceinline__ void grid_barrier(Bar* b, unsigned G, unsigned g) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (!HIER) {
            const unsigned prev = __hip_atomic_fetch_add(&b->cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == G * g - 1) __hip_atomic_store(&b->gen, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else spin_until(&b->gen, g, &b->err);
        } else {
            const unsigned x = blockIdx.x & 7, per = G >> 3;          // round-robin block -> XCD placement
            const unsigned prev = __hip_atomic_fetch_add(&b->xcnt[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool last = false;
            if (prev == per * g - 1) {
                const unsigned p2 = __hip_atomic_fetch_add(&b->cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (p2 == 8 * g - 1) { __hip_atomic_store(&b->gen, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); last = true; }
            }
            if (!last) spin_until(&b->gen, g, &b->err);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

template <bool HIER, int PAY>     // PAY = floats per thread per phase
__global__ __launch_bounds__(256) void k_persistent(Bar* bar, float* buf, unsigned* stale, int phases) {
    const unsigned G = gridDim.x;
    const size_t per_block = (size_t)256 * (PAY ? PAY : 1);
    unsigned bad = 0;
    for (int p = 1; p <= phases; ++p) {
        if (PAY) {
            float* mine = buf + ((size_t)(p & 1) * G + blockIdx.x) * per_block;
            const float* theirs = buf + ((size_t)((p - 1) & 1) * G + (blockIdx.x + 17) % G) * per_block;
            float acc = 0;
#pragma unroll
            for (int j = 0; j < (PAY ? PAY : 1); ++j) {
                const float v = theirs[j * 256 + threadIdx.x];
                if (p > 1 && v != (float)(p - 1)) ++bad;
                acc += v;
            }
#pragma unroll
            for (int j = 0; j < (PAY ? PAY : 1); ++j) mine[j * 256 + threadIdx.x] = (float)p + 0.0f * acc;
        }
        grid_barrier<HIER>(bar, G, (unsigned)p);
        if (__hip_atomic_load(&bar->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    if (bad) atomicAdd(stale, bad);
}

template <int PAY>
__global__ __launch_bounds__(256) void k_phase(float* buf, unsigned* stale, int p) {
    const unsigned G = gridDim.x;
    const size_t per_block = (size_t)256 * (PAY ? PAY : 1);
    if (!PAY) return;
    float* mine = buf + ((size_t)(p & 1) * G + blockIdx.x) * per_block;
    const float* theirs = buf + ((size_t)((p - 1) & 1) * G + (blockIdx.x + 17) % G) * per_block;
    float acc = 0; unsigned bad = 0;
#pragma unroll
    for (int j = 0; j < (PAY ? PAY : 1); ++j) {
        const float v = theirs[j * 256 + threadIdx.x];
        if (p > 1 && v != (float)(p - 1)) ++bad;
        acc += v;
    }
#pragma unroll
    for (int j = 0; j < (PAY ? PAY : 1); ++j) mine[j * 256 + threadIdx.x] = (float)p + 0.0f * acc;
    if (bad) atomicAdd(stale, bad);
}

template <bool HIER, int PAY>
static int run_persistent(const char* name, int G, int phases, Bar* bar, float* buf, unsigned* stale, hipStream_t st) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemsetAsync(bar, 0, sizeof(Bar), st)); CK(hipMemsetAsync(stale, 0, 4, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((k_persistent<HIER, PAY>), dim3(G), dim3(256), 0, st, bar, buf, stale, phases);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    Bar h; unsigned hs = 0;
    CK(hipMemcpy(&h, bar, sizeof(Bar), hipMemcpyDeviceToHost)); CK(hipMemcpy(&hs, stale, 4, hipMemcpyDeviceToHost));
    printf("%-44s G=%4d pay=%5d B/blk  %7.3f us/phase  err=%u stale=%u\n", name, G, PAY * 1024, best * 1000 / phases, h.err, hs);
    fflush(stdout);
    return h.err ? 1 : 0;
}

template <int PAY>
static int run_graph(int G, int phases, float* buf, unsigned* stale, hipStream_t st) {
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int p = 1; p <= phases; ++p) hipLaunchKernelGGL((k_phase<PAY>), dim3(G), dim3(256), 0, st, buf, stale, p);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemsetAsync(stale, 0, 4, st));
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(exec, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    unsigned hs = 0; CK(hipMemcpy(&hs, stale, 4, hipMemcpyDeviceToHost));
    printf("%-44s G=%4d pay=%5d B/blk  %7.3f us/phase  stale=%u\n", "hipGraph of dependent launches", G, PAY * 1024, best * 1000 / phases, hs);
    fflush(stdout);
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    return 0;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    Bar* bar; CK(hipMalloc((void**)&bar, sizeof(Bar)));
    float* buf; CK(hipMalloc((void**)&buf, (size_t)2 * 1024 * 256 * 16 * 4));
    CK(hipMemset(buf, 0, (size_t)2 * 1024 * 256 * 16 * 4));
    unsigned* stale; CK(hipMalloc((void**)&stale, 4));
    const int phases = 400;
    for (int G : {256, 512, 1024}) {
        if (run_persistent<false, 0>("persistent, flat barrier", G, phases, bar, buf, stale, st)) return 2;
        if (run_persistent<true, 0>("persistent, per-XCD then global", G, phases, bar, buf, stale, st)) return 2;
        if (run_persistent<false, 4>("persistent, flat barrier", G, phases, bar, buf, stale, st)) return 2;
        if (run_persistent<true, 4>("persistent, per-XCD then global", G, phases, bar, buf, stale, st)) return 2;
        if (run_persistent<false, 16>("persistent, flat barrier", G, phases, bar, buf, stale, st)) return 2;
        if (run_graph<0>(G, phases, buf, stale, st)) return 1;
        if (run_graph<4>(G, phases, buf, stale, st)) return 1;
        if (run_graph<16>(G, phases, buf, stale, st)) return 1;
    }
    return 0;
}
