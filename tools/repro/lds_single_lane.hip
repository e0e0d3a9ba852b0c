// Stand-alone attempt to reproduce the merge kernel's single-thread-section failure outside the library:
// thread 0 of a 512-thread workgroup loads split statistics (vector loads: the index is laundered through a VGPR), computes four
// weights for each of the workgroup's four clips, writes them to LDS (4 x ds_write_b128), __syncthreads(), every lane reads the
// weights of "its" clip and stores them.  MODE 0: every lane computes the weights itself (the reference).  Each launch uses the
// same inputs; the host counts workgroups whose stored weights differ from the reference's bits.
//   hipcc --offload-arch=gfx950 -O3 -o build/lds_single_lane tools/repro/lds_single_lane.hip && build/lds_single_lane [launches] [streams]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int H = 12, B = 64, CL = 4, NS = 4, D = 768, NLOAD = 9;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void merge_like(const float* __restrict__ part_m, const float* __restrict__ part_l,
                                                  const float* __restrict__ part_o, float* __restrict__ out, int n_splits) {
    __shared__ __attribute__((aligned(16))) float ws_single[CL][4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b0 = blockIdx.y * CL;
    const int bc = min(b0 + (l15 & (CL - 1)), B - 1);
    // operand loads in flight across the section (the merge kernel has 36 of them)
    f32x4 po[NLOAD][4];
#pragma unroll
    for (int k = 0; k < NLOAD; ++k)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            po[k][s] = *reinterpret_cast<const f32x4*>(part_o + (((int64_t)bc * n_splits + min(s, n_splits - 1)) * 16 + h) * D + w * 96 + 32 * (k % 3) + 8 * g + 4 * (k / 3 % 2));
    float pm[4], pl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        pm[s] = part_m[((int64_t)bc * n_splits + min(s, n_splits - 1)) * 16 + h];
        pl[s] = part_l[((int64_t)bc * n_splits + min(s, n_splits - 1)) * 16 + h];
    }
    __builtin_amdgcn_sched_barrier(0);
    float ws[4];
    if (MODE == 1) {
        if (tid == 0) {
            int launder = 0;
            asm volatile("" : "+v"(launder));
            for (int cl = 0; cl < CL; ++cl) {
                const int bq = min(b0 + cl, B - 1) + launder;
                float mm[4], ll[4], M = -1.0e30f, Lsum = 0.f;
                for (int s = 0; s < 4; ++s) {
                    const int sc = min(s, n_splits - 1);
                    mm[s] = part_m[((int64_t)bq * n_splits + sc) * 16 + h];
                    ll[s] = part_l[((int64_t)bq * n_splits + sc) * 16 + h];
                    if (s < n_splits) M = fmaxf(M, mm[s]);
                }
                for (int s = 0; s < 4; ++s) {
                    mm[s] = s < n_splits ? __expf(mm[s] - M) : 0.f;
                    Lsum += mm[s] * ll[s];
                }
                const float inv = 1.0f / Lsum;
                for (int s = 0; s < 4; ++s) ws_single[cl][s] = mm[s] * inv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) ws[s] = ws_single[l15 & (CL - 1)][s];
    } else {
        float M = -1.0e30f, Lsum = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (s < n_splits) M = fmaxf(M, pm[s]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            ws[s] = s < n_splits ? __expf(pm[s] - M) : 0.f;
            Lsum += ws[s] * pl[s];
        }
        const float inv = 1.0f / Lsum;
#pragma unroll
        for (int s = 0; s < 4; ++s) ws[s] *= inv;
    }
    // merged rows (keeps the operand loads alive, like the A fragments of the value projection) + the weights themselves
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NLOAD; ++k)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc += ws[s] * po[k][s];
    float* o = out + (((int64_t)blockIdx.y * H + h) * 512 + tid) * 8;
    *reinterpret_cast<f32x4*>(o) = f32x4{ws[0], ws[1], ws[2], ws[3]};
    *reinterpret_cast<f32x4*>(o + 4) = acc;
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 2000, n_streams = argc > 2 ? atoi(argv[2]) : 1;
    const size_t n_stat = (size_t)B * NS * 16, n_o = n_stat * D, n_out = (size_t)(B / CL) * H * 512 * 8;
    std::vector<float> hm(n_stat), hl(n_stat), ho(n_o);
    srand(1);
    for (auto& v : hm) v = 4.f * rand() / RAND_MAX - 2.f;
    for (auto& v : hl) v = 50.f + 400.f * rand() / RAND_MAX;
    for (auto& v : ho) v = 2.f * rand() / RAND_MAX - 1.f;
    float *dm, *dl, *dO;
    CK(hipMalloc(&dm, n_stat * 4)); CK(hipMalloc(&dl, n_stat * 4)); CK(hipMalloc(&dO, n_o * 4));
    CK(hipMemcpy(dm, hm.data(), n_stat * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dl, hl.data(), n_stat * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dO, ho.data(), n_o * 4, hipMemcpyHostToDevice));
    std::vector<hipStream_t> st(n_streams);
    std::vector<float*> dout(n_streams);
    for (int i = 0; i < n_streams; ++i) { CK(hipStreamCreate(&st[i])); CK(hipMalloc(&dout[i], n_out * 4)); }
    const dim3 grid(H, B / CL), block(512);
    std::vector<float> ref(n_out), got(n_out);
    hipLaunchKernelGGL(merge_like<0>, grid, block, 0, st[0], dm, dl, dO, dout[0], NS);
    CK(hipStreamSynchronize(st[0]));
    CK(hipMemcpy(ref.data(), dout[0], n_out * 4, hipMemcpyDeviceToHost));
    long bad_launches = 0, bad_wgs = 0;
    const int per_check = argc > 3 ? atoi(argv[3]) : 1;
    for (int it = 0; it < launches; it += per_check) {
        for (int k = 0; k < per_check; ++k)
            for (int i = 0; i < n_streams; ++i) hipLaunchKernelGGL(merge_like<1>, grid, block, 0, st[i], dm, dl, dO, dout[i], NS);
        for (int i = 0; i < n_streams; ++i) {
            CK(hipStreamSynchronize(st[i]));
            CK(hipMemcpy(got.data(), dout[i], n_out * 4, hipMemcpyDeviceToHost));
            if (memcmp(got.data(), ref.data(), n_out * 4) != 0) {
                ++bad_launches;
                for (size_t wg = 0; wg < n_out / (512 * 8); ++wg)
                    if (memcmp(&got[wg * 512 * 8], &ref[wg * 512 * 8], 512 * 8 * 4) != 0) {
                        ++bad_wgs;
                        if (bad_wgs <= 6) {
                            size_t j = wg * 512 * 8;
                            while (got[j] == ref[j]) ++j;
                            printf("iteration %d stream %d: workgroup %zu (clips %zu.., head %zu) differs first at thread %zu float %zu: %.7g vs %.7g\n", it, i, wg,
                                   (wg / H) * CL, wg % H, (j - wg * 512 * 8) / 8, (j - wg * 512 * 8) % 8, got[j], ref[j]);
                        }
                    }
            }
        }
    }
    printf("launches checked (last of every %d per stream): %d x %d streams; launches with a wrong workgroup: %ld; wrong workgroups: %ld\n", per_check,
           launches / per_check, n_streams, bad_launches, bad_wgs);
    return 0;
}
