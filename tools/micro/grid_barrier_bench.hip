// Micro-benchmark (GPU box): what does a device-wide barrier inside ONE persistent kernel cost on MI355X, against the
// ~4.3 us dependency gap between two kernel nodes of a hipGraph?  Decides whether a persistent cooperative decode step
//  can beat the graph-replayed step.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier_bench.hip -o /tmp/gbb && /tmp/gbb
// The barrier: arrival counter + generation word in global memory, agent-scope atomics (8 XCDs: every arrival crosses the
// fabric).  Every spin is bounded: a workgroup that waits more than ~50 ms raises a flag and leaves, so the grid always drains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Bar { unsigned int count, gen, failed, pad; };

__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned int nblocks, unsigned int& my_gen) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned int target = my_gen + 1;
        const unsigned int prev = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == nblocks - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b->gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != target) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 5000000) {  // 50 ms at 100 MHz: give up, never hang
                    __hip_atomic_store(&b->failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
            }
        }
    }
    my_gen += 1;
    __syncthreads();
    return ok;
}

// n_iter barriers; between barriers every workgroup does one dependent global round trip (reads what its neighbour wrote
// before the barrier), like a phase of the decode step.
__global__ __launch_bounds__(256) void persistent_kernel(Bar* b, float* buf, int n_iter, int work) {
    unsigned int gen = 0;
    const unsigned int nb = gridDim.x;
    float v = (float)blockIdx.x;
    for (int it = 0; it < n_iter; ++it) {
        if (work) {
            if (threadIdx.x == 0) buf[(it & 1) * nb + blockIdx.x] = v;
        }
        if (!grid_barrier(b, nb, gen)) return;
        if (b->failed) return;
        if (work) {
            v += buf[(it & 1) * nb + (blockIdx.x + 1) % nb];  // written by another workgroup (another XCD) before the barrier
        }
    }
    if (threadIdx.x == 0) buf[2 * nb + blockIdx.x] = v;
}

__global__ void tiny_kernel(float* buf, int i) {
    if (threadIdx.x == 0 && blockIdx.x == 0) buf[0] += (float)i;
}

int main() {
    Bar* bar;
    float* buf;
    CHECK(hipMalloc(&bar, sizeof(Bar)));
    CHECK(hipMalloc(&buf, 4096 * sizeof(float)));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int n_iter = 2000;
    for (int nb : {64, 128, 256, 512}) {
        for (int work : {0, 1}) {
            CHECK(hipMemsetAsync(bar, 0, sizeof(Bar), s));
            CHECK(hipMemsetAsync(buf, 0, 4096 * sizeof(float), s));
            hipLaunchKernelGGL(persistent_kernel, dim3(nb), dim3(256), 0, s, bar, buf, 10, work);  // warm
            CHECK(hipMemsetAsync(bar, 0, sizeof(Bar), s));
            CHECK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(persistent_kernel, dim3(nb), dim3(256), 0, s, bar, buf, n_iter, work);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            Bar h;
            CHECK(hipMemcpy(&h, bar, sizeof(Bar), hipMemcpyDeviceToHost));
            printf("persistent kernel, %3d workgroups x 256 threads, %s: %.3f us per barrier%s\n", nb,
                   work ? "barrier + one dependent cross-workgroup round trip" : "barrier only", ms * 1e3 / n_iter,
                   h.failed ? "  [TIMED OUT: grid not co-resident]" : "");
        }
    }
    // the alternative: a chain of dependent kernel nodes replayed from a graph
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(64), dim3(256), 0, s, buf, i);
    CHECK(hipStreamEndCapture(s, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(exec, s));
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(e0, s));
    for (int r = 0; r < 10; ++r) CHECK(hipGraphLaunch(exec, s));
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("hipGraph chain of 200 dependent 64-workgroup kernels: %.3f us per node\n", ms * 1e3 / 2000);
    return 0;
}
