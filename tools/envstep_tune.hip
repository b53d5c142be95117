// envstep_tune.hip -- development microbenchmark: variants of the SoA CartPole env-step kernel against
// the HBM roof, plus copy kernels with the same stream count as ceilings.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I simple-es_amd/csrc tools/envstep_tune.hip -o tools/envstep_tune
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ses_cartpole.h"

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e = (x);                                                   \
        if (e != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e));                     \
            exit(1);                                                          \
        }                                                                     \
    } while (0)

using namespace ses;

__device__ __forceinline__ void step_one(float &x, float &xd, float &th, float &thd, int action, float &ret,
                                         uint32_t &status, int max_step)
{
    const bool done = (status >> 31) != 0u;
    const uint32_t steps = status & 0x7fffffffu;
    CartPoleState s{x, xd, th, thd};
    const bool term = cartpole_step(s, action);
    x = s.x; xd = s.xd; th = s.th; thd = s.thd;
    const uint32_t nsteps = steps + 1u;
    const bool now_done = term | ((int)nsteps >= max_step);
    ret = done ? ret : ret + 1.0f;
    status = done ? status : (nsteps | ((uint32_t)now_done << 31));
}

typedef float f4v __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld(const float4 *p)
{
    if constexpr (NT) {
        const f4v v = __builtin_nontemporal_load((const f4v *)p);
        return make_float4(v.x, v.y, v.z, v.w);
    } else return *p;
}
template <bool NT>
__device__ __forceinline__ void st(float4 *p, float4 v)
{
    if constexpr (NT) {
        f4v w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, (f4v *)p);
    } else *p = v;
}

// UNROLL float4 per array per thread-iteration; grid-stride
template <bool NT, int UNROLL, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_step(int n4, int max_step, float4 *x, float4 *xd, float4 *th, float4 *thd,
                                                const float4 *action, float4 *ret, float4 *status)
{
    const int stride = gridDim.x * BLOCK;
    for (int i0 = blockIdx.x * BLOCK + threadIdx.x; i0 < n4; i0 += stride * UNROLL) {
        float4 vx[UNROLL], vxd[UNROLL], vth[UNROLL], vthd[UNROLL], vr[UNROLL], va[UNROLL], vs[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int i = i0 + u * stride;
            if (i < n4) {
                vx[u] = ld<NT>(x + i); vxd[u] = ld<NT>(xd + i); vth[u] = ld<NT>(th + i); vthd[u] = ld<NT>(thd + i);
                va[u] = ld<NT>(action + i); vr[u] = ld<NT>(ret + i); vs[u] = ld<NT>(status + i);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int i = i0 + u * stride;
            if (i < n4) {
                uint32_t s0 = __float_as_uint(vs[u].x), s1 = __float_as_uint(vs[u].y), s2 = __float_as_uint(vs[u].z),
                         s3 = __float_as_uint(vs[u].w);
                step_one(vx[u].x, vxd[u].x, vth[u].x, vthd[u].x, __float_as_int(va[u].x), vr[u].x, s0, max_step);
                step_one(vx[u].y, vxd[u].y, vth[u].y, vthd[u].y, __float_as_int(va[u].y), vr[u].y, s1, max_step);
                step_one(vx[u].z, vxd[u].z, vth[u].z, vthd[u].z, __float_as_int(va[u].z), vr[u].z, s2, max_step);
                step_one(vx[u].w, vxd[u].w, vth[u].w, vthd[u].w, __float_as_int(va[u].w), vr[u].w, s3, max_step);
                vs[u] = make_float4(__uint_as_float(s0), __uint_as_float(s1), __uint_as_float(s2), __uint_as_float(s3));
                st<NT>(x + i, vx[u]); st<NT>(xd + i, vxd[u]); st<NT>(th + i, vth[u]); st<NT>(thd + i, vthd[u]);
                st<NT>(ret + i, vr[u]); st<NT>(status + i, vs[u]);
            }
        }
    }
}

// same traffic, no math: 7 read streams, 6 write streams
template <bool NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_copy13(int n4, float4 *x, float4 *xd, float4 *th, float4 *thd,
                                                  const float4 *action, float4 *ret, float4 *status)
{
    const int stride = gridDim.x * BLOCK;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n4; i += stride) {
        float4 a = ld<NT>(x + i), b = ld<NT>(xd + i), c = ld<NT>(th + i), d = ld<NT>(thd + i), e = ld<NT>(action + i),
               f = ld<NT>(ret + i), g = ld<NT>(status + i);
        a.x += e.x; b.x += e.y; c.x += e.z; d.x += e.w;
        st<NT>(x + i, a); st<NT>(xd + i, b); st<NT>(th + i, c); st<NT>(thd + i, d); st<NT>(ret + i, f); st<NT>(status + i, g);
    }
}

template <bool NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_copy2(long n4, const float4 *src, float4 *dst)
{
    const long stride = (long)gridDim.x * BLOCK;
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += stride) st<NT>(dst + i, ld<NT>(src + i));
}

template <typename F>
static float time_it(F launch, int reps = 20)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 24;
    const int n = 1 << lg, n4 = n / 4;
    float *buf[7];
    std::vector<float> h(n);
    for (int k = 0; k < 7; ++k) {
        CK(hipMalloc(&buf[k], (size_t)n * 4));
        for (int i = 0; i < n; ++i) h[i] = k < 4 ? (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f * 0.1f - 0.05f : 0.0f;
        if (k == 4) for (int i = 0; i < n; ++i) { int a = (i * 40503u >> 7) & 1; h[i] = *(float *)&a; }
        CK(hipMemcpy(buf[k], h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    }
    float4 *x = (float4 *)buf[0], *xd = (float4 *)buf[1], *th = (float4 *)buf[2], *thd = (float4 *)buf[3],
           *ac = (float4 *)buf[4], *rt = (float4 *)buf[5], *stt = (float4 *)buf[6];
    const double bytes = 52.0 * n;
    printf("n = 2^%d envs, %.0f MB per launch\n", lg, bytes / 1e6);
#define RUN(NAME, GRID, BLOCK, KERNEL)                                                                               \
    {                                                                                                                \
        float ms = time_it([&] { hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), 0, 0, n4, 1 << 30, x, xd, th, thd, ac, rt, stt); }); \
        printf("%-44s grid %6d block %4d : %8.1f us  %7.1f GB/s\n", NAME, GRID, BLOCK, ms * 1e3, bytes / ms / 1e6);  \
    }
#define RUNC(NAME, GRID, BLOCK, KERNEL)                                                                              \
    {                                                                                                                \
        float ms = time_it([&] { hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), 0, 0, n4, x, xd, th, thd, ac, rt, stt); }); \
        printf("%-44s grid %6d block %4d : %8.1f us  %7.1f GB/s\n", NAME, GRID, BLOCK, ms * 1e3, bytes / ms / 1e6);  \
    }
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        RUN("step plain u1", grid, 256, (k_step<false, 1, 256>));
        RUN("step nt    u1", grid, 256, (k_step<true, 1, 256>));
    }
    for (int grid : {1024, 2048, 4096}) {
        RUN("step plain u2", grid, 256, (k_step<false, 2, 256>));
        RUN("step nt    u2", grid, 256, (k_step<true, 2, 256>));
        RUN("step nt    u1 b512", grid, 512, (k_step<true, 1, 512>));
        RUN("step nt    u1 b1024", grid / 2, 1024, (k_step<true, 1, 1024>));
    }
    RUN("step nt u1 one-shot grid", n4 / 256, 256, (k_step<true, 1, 256>));
    RUN("step plain u1 one-shot grid", n4 / 256, 256, (k_step<false, 1, 256>));
    for (int grid : {2048, 8192}) {
        RUNC("copy13 plain", grid, 256, (k_copy13<false, 256>));
        RUNC("copy13 nt", grid, 256, (k_copy13<true, 256>));
    }
    RUNC("copy13 nt one-shot", n4 / 256, 256, (k_copy13<true, 256>));
    {
        const long c4 = (long)n4 * 3;  // x,xd,th -> thd.. : 3 arrays worth, 2 streams
        float ms = time_it([&] { hipLaunchKernelGGL((k_copy2<false, 256>), dim3(8192), dim3(256), 0, 0, (long)n4, (const float4 *)x, xd); });
        printf("%-44s : %8.1f us  %7.1f GB/s\n", "copy2 plain (1 in 1 out)", ms * 1e3, 8.0 * n / ms / 1e6);
        ms = time_it([&] { hipLaunchKernelGGL((k_copy2<true, 256>), dim3(8192), dim3(256), 0, 0, (long)n4, (const float4 *)x, xd); });
        printf("%-44s : %8.1f us  %7.1f GB/s\n", "copy2 nt (1 in 1 out)", ms * 1e3, 8.0 * n / ms / 1e6);
        (void)c4;
    }
    // stream-skew experiment: the 7 arrays carved out of ONE allocation at n*4 + skew strides
    {
        char *big;
        const size_t stride_max = (size_t)n * 4 + (8u << 20);
        CK(hipMalloc(&big, stride_max * 7));
        for (size_t skew : {(size_t)0, (size_t)4096, (size_t)4096, (size_t)8192, (size_t)12288, (size_t)16384, (size_t)20480,
                            (size_t)28672, (size_t)36864, (size_t)65536, (size_t)69632, (size_t)131072 + 4096,
                            (size_t)262144 + 4096, (size_t)(1u << 20) + 4096, (size_t)4096}) {
            float4 *p[7];
            for (int k = 0; k < 7; ++k) {
                p[k] = (float4 *)(big + k * ((size_t)n * 4 + skew));
                CK(hipMemcpy(p[k], buf[k], (size_t)n * 4, hipMemcpyDeviceToDevice));
            }
            float ms = time_it([&] { hipLaunchKernelGGL((k_step<true, 1, 256>), dim3(n4 / 256), dim3(256), 0, 0, n4, 1 << 30, p[0], p[1], p[2], p[3], p[4], p[5], p[6]); });
            float ms2 = time_it([&] { hipLaunchKernelGGL((k_copy13<true, 256>), dim3(n4 / 256), dim3(256), 0, 0, n4, p[0], p[1], p[2], p[3], p[4], p[5], p[6]); });
            printf("skew %9zu B: step nt one-shot %8.1f us %7.1f GB/s | copy13 nt one-shot %8.1f us %7.1f GB/s\n", skew, ms * 1e3,
                   bytes / ms / 1e6, ms2 * 1e3, bytes / ms2 / 1e6);
        }
    }
    return 0;
}
