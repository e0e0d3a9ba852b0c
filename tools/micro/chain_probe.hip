// Dependent-launch chains on several HIP streams, timed from INSIDE the kernels.
//
// rocprofv3 --kernel-trace cannot show what several decode loops in flight do to each other: under it the passes of
// `bench.py --phase dec --pipeline 4` run one after another (profiles/r05_decode_gaps.txt: 11 stream changes in 63 285
// dispatches, 86.9 ms per pass traced against 23.9 ms untraced).  This probe needs no tracer: every kernel ("link") writes the
// GPU's constant 100 MHz clock (wall_clock64) when its first workgroup starts and when its last workgroup ends.
//
// A stream's work is the decode step's shape: a hipGraph of N links, each depending on the one before (a link reads what the
// previous link wrote), replayed R times back to back; S streams do so at once, each on its own hardware queue
// (GPU_MAX_HW_QUEUES=8 is set before the runtime starts).  Printed per shape and S: the mean period of a link on its stream,
// split into kernel duration (first start .. last end) and gap (last end of link i .. first start of link i+1) -- the gaps
// INSIDE a replay (graph node -> graph node: the GPU's own doing) apart from the one gap per replay at the graph BOUNDARY
// (which also waits for the host's next hipGraphLaunch), and the host time one hipGraphLaunch takes: without that split a
// host-bound launch loop and a slow command processor look the same.
//
//   hipcc --offload-arch=gfx950 -O3 -o chain_probe tools/micro/chain_probe.hip && ./chain_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

struct Probe {
    unsigned long long* t_start;  // [R * N]
    unsigned long long* t_end;    // [R * N]
    unsigned int* arrived;        // [N] workgroups of link i that have finished (reset by the last one)
    unsigned int* rep;            // replay counter of this stream (advanced by the last link)
    const float* in;              // what the previous link wrote
    float* out;
    int n_links, max_rep;
};

// one link: `work` dependent 16-byte loads per thread from the previous link's output (L2 / fabric round trips, as a small
// decode kernel has), a store, optional LDS footprint (dynamic) so that co-residency can be constrained like the real kernels
__global__ void link_kernel(Probe p, int i, int work, int lds_floats) {
    extern __shared__ float lds[];  // footprint only: touched when the launch asked for any
    __shared__ float s_acc;
    const unsigned int rep = *(volatile unsigned int*)p.rep;
    const int slot = (int)min(rep, (unsigned)p.max_rep - 1) * p.n_links + i;
    if (blockIdx.x == 0 && threadIdx.x == 0) p.t_start[slot] = wall_clock64();
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    int idx = tid & 16383;
    for (int k = 0; k < work; ++k) {  // dependent: the next address comes from the value just loaded
        const float v = __builtin_nontemporal_load(p.in + idx);
        acc += v;
        idx = (idx + 4099 + ((int)v & 1)) & 16383;
    }
    if (threadIdx.x == 0) {
        s_acc = acc;
        if (lds_floats > 0) lds[lds_floats - 1] = acc;  // the last word of the dynamic block: inside the allocation
    }
    __syncthreads();
    p.out[tid & 16383] = acc + s_acc * 0.f + 1.0f;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned int done = atomicAdd(p.arrived + i, 1u);
        if (done == gridDim.x - 1) {  // the last workgroup of the link
            p.arrived[i] = 0;
            p.t_end[slot] = wall_clock64();
            if (i == p.n_links - 1) atomicAdd(p.rep, 1u);
        }
    }
}

struct Shape {
    const char* name;
    int grid, block, lds_bytes, work;
};

// PAD idle streams are created BEFORE the working ones (env CHAIN_PAD): shifts which hardware queue each working stream lands on
// (ROCm hands queues out in creation order); PRIO (env CHAIN_PRIO=1): the working streams alternate between the two priorities
// HIP offers, which also moves them to other queues.
static int g_pad = 0, g_prio = 0;
static void run(const Shape& sh, int S, int N, int R) {
    std::vector<hipStream_t> st(S);
    std::vector<hipGraphExec_t> ge(S);
    std::vector<Probe> pr(S);
    std::vector<float*> bufs;
    double host_us = 0;
    for (int s = 0; s < S; ++s) {
        if (g_prio) {
            int lo = 0, hi = 0;
            CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            CK(hipStreamCreateWithPriority(&st[s], hipStreamNonBlocking, (s & 1) ? hi : lo));
        } else {
            CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
        }
        Probe& p = pr[s];
        p.n_links = N;
        p.max_rep = R;
        CK(hipMalloc(&p.t_start, sizeof(unsigned long long) * R * N));
        CK(hipMalloc(&p.t_end, sizeof(unsigned long long) * R * N));
        CK(hipMalloc(&p.arrived, sizeof(unsigned int) * N));
        CK(hipMalloc(&p.rep, sizeof(unsigned int)));
        CK(hipMemset(p.arrived, 0, sizeof(unsigned int) * N));
        float *a, *b;
        CK(hipMalloc(&a, 16384 * sizeof(float)));
        CK(hipMalloc(&b, 16384 * sizeof(float)));
        CK(hipMemset(a, 0, 16384 * sizeof(float)));
        CK(hipMemset(b, 0, 16384 * sizeof(float)));
        bufs.push_back(a);
        bufs.push_back(b);
        CK(hipFuncSetAttribute((const void*)link_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 148 * 1024));  // + the static word: within the 160 KB of a workgroup
        hipGraph_t g;
        CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) {
            Probe q = p;
            q.in = (i & 1) ? b : a;
            q.out = (i & 1) ? a : b;
            hipLaunchKernelGGL(link_kernel, dim3(sh.grid), dim3(sh.block), sh.lds_bytes, st[s], q, i, sh.work, sh.lds_bytes / 4);
        }
        CK(hipStreamEndCapture(st[s], &g));
        CK(hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    for (int pass = 0; pass < 2; ++pass) {  // pass 0 warms up (uploads the graphs), pass 1 is read
        for (int s = 0; s < S; ++s) CK(hipMemsetAsync(pr[s].rep, 0, sizeof(unsigned int), st[s]));
        CK(hipDeviceSynchronize());
        const auto h0 = std::chrono::steady_clock::now();
        for (int r = 0; r < R; ++r)
            for (int s = 0; s < S; ++s) CK(hipGraphLaunch(ge[s], st[s]));
        host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count() / (R * S);
        CK(hipDeviceSynchronize());
    }
    double dur = 0, gap = 0, period = 0, bgap = 0;
    long n_gap = 0, n_dur = 0, n_bgap = 0;
    double wall = 0;
    for (int s = 0; s < S; ++s) {
        std::vector<unsigned long long> ts(R * N), te(R * N);
        CK(hipMemcpy(ts.data(), pr[s].t_start, sizeof(unsigned long long) * R * N, hipMemcpyDeviceToHost));
        CK(hipMemcpy(te.data(), pr[s].t_end, sizeof(unsigned long long) * R * N, hipMemcpyDeviceToHost));
        const int lo = (R / 4) * N, hi = R * N;  // skip the first quarter (the queues fill)
        for (int k = lo; k < hi; ++k) {
            dur += (double)(te[k] - ts[k]);
            ++n_dur;
            if (k + 1 < hi) {
                const double g = (double)((long long)ts[k + 1] - (long long)te[k]);
                if ((k + 1) % N == 0) {  // link N-1 of a replay -> link 0 of the next: the graph boundary
                    bgap += g;
                    ++n_bgap;
                } else {
                    gap += g;
                    ++n_gap;
                }
            }
        }
        period += (double)(ts[hi - 1] - ts[lo]) / (hi - 1 - lo);
        wall = std::max(wall, (double)(te[hi - 1] - ts[lo]));
    }
    const double tick_us = 0.01;  // 100 MHz
    printf("%-34s S=%d  link period on a stream %6.2f us | kernel %5.2f | gap inside a replay %5.2f | gap at the graph boundary %7.2f | "
           "replay on the GPU %7.1f us, host per hipGraphLaunch %6.1f us x %d streams | chip-wide one link every %5.2f us\n",
           sh.name, S, period / S * tick_us, dur / n_dur * tick_us, gap / n_gap * tick_us, bgap / std::max(n_bgap, 1L) * tick_us,
           period / S * tick_us * N, host_us, S, period / S * tick_us / S);
    for (int s = 0; s < S; ++s) {
        CK(hipGraphExecDestroy(ge[s]));
        CK(hipStreamDestroy(st[s]));
        CK(hipFree(pr[s].t_start));
        CK(hipFree(pr[s].t_end));
        CK(hipFree(pr[s].arrived));
        CK(hipFree(pr[s].rep));
    }
    for (float* b : bufs) CK(hipFree(b));
}

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);  // before the runtime starts: as the package asks (runtime.request_hw_queues)
    const int N = 136, R = argc > 1 ? atoi(argv[1]) : 48;  // the decode step: ~136 dependent launches, replayed once per position
    g_pad = getenv("CHAIN_PAD") ? atoi(getenv("CHAIN_PAD")) : 0;
    g_prio = getenv("CHAIN_PRIO") ? atoi(getenv("CHAIN_PRIO")) : 0;
    const bool quick = getenv("CHAIN_QUICK") != nullptr;  // the two lightest shapes, S = 3 and 4 only
    std::vector<hipStream_t> pad(g_pad);
    unsigned int* touch = nullptr;
    CK(hipMalloc(&touch, sizeof(unsigned int)));
    for (int i = 0; i < g_pad; ++i) {
        CK(hipStreamCreateWithFlags(&pad[i], hipStreamNonBlocking));
        CK(hipMemsetAsync(touch, 0, sizeof(unsigned int), pad[i]));  // use the stream once so that its hardware queue exists
    }
    CK(hipDeviceSynchronize());
    printf("GPU_MAX_HW_QUEUES=%s, %d idle stream(s) created first, priorities %s\n", getenv("GPU_MAX_HW_QUEUES"), g_pad, g_prio ? "alternating" : "default");
    const Shape shapes[] = {
        {"1 wg x 64 thr (dispatch only)", 1, 64, 0, 1},
        {"96 wg x 512 thr, 4 round trips", 96, 512, 0, 4},                 // the prologue / merge kernels' shape
        {"384 wg x 256 thr, 4 round trips", 384, 256, 0, 4},               // a skinny weight-streaming GEMM's shape
        {"384 wg x 256 thr, 64 KB LDS each", 384, 256, 64 * 1024, 4},      // ... that only fits two to a CU
        {"128 wg x 192 thr, 144 KB LDS each", 128, 192, 144 * 1024, 16},   // the half-chip streaming launch's footprint
    };
    int si = 0;
    for (const Shape& sh : shapes) {
        if (quick && si++ >= 2) break;
        for (int S : {1, 2, 3, 4, 5, 6}) {
            if (quick && S < 3) continue;
            if (!quick && S > 4) continue;
            run(sh, S, N, R);
        }
    }
    return 0;
}
