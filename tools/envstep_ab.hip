// envstep_ab.hip -- development microbenchmark (not part of the product): interleaved A/B of the SoA CartPole env-step
// kernel's launch shapes against hand-written copies of the same streams, one process, one box.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I simple-es_amd/csrc tools/envstep_ab.hip -o tools/envstep_ab
//   tools/envstep_ab [log2 envs = 24] [rounds = 15] [launches per batch = 20] [pre-roll launches = 200]
// Variants: the product shape (16 B per array per lane, one-shot grid, non-temporal), two 16-B
// groups per lane, persistent grids sized to the CUs, s_setprio, block sizes, cache policies; ceilings: the same 13
// streams with no arithmetic (copy13), and a 2-stream float4 copy of the same byte count.
// Every variant is timed `rounds` times in round-robin order (HIP events around `launches` back-to-back launches);
// the table reports median and minimum per variant, so that box drift is common to all of them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
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

typedef float f4v __attribute__((ext_vector_type(4)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef int32_t i4v __attribute__((ext_vector_type(4)));

template <bool NT, class V>
__device__ __forceinline__ V ld(const V *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, class V>
__device__ __forceinline__ void st(V *p, V v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

__device__ __forceinline__ void step_one(float &x, float &xd, float &th, float &thd, int action, float &ret,
                                         uint32_t &status, int max_step)
{
    const bool done = (status >> 31) != 0u;
    const uint32_t steps = status & 0x7fffffffu;
    CartPoleState s{x, xd, th, thd};
    const bool term = cartpole_step(s, action);
    x = s.x; xd = s.xd; th = s.th; thd = s.thd;              // fixed-length mode: finished envs keep stepping
    const uint32_t nsteps = steps + 1u;
    const bool now_done = term | (max_step > 0 && (int)nsteps >= max_step);
    ret = done ? ret : ret + 1.0f;
    status = done ? status : (nsteps | ((uint32_t)now_done << 31));
}

// U float4 groups per array per lane and trip; grid-stride (a one-shot grid makes exactly one trip)
template <bool NTL, bool NTS, int U, int BLOCK, int PRIO>
__global__ __launch_bounds__(BLOCK) void k_step(int n4, int max_step, f4v *x, f4v *xd, f4v *th, f4v *thd, const i4v *action,
                                                f4v *ret, u4v *status)
{
    if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
    const int stride = gridDim.x * BLOCK;
    for (int i0 = blockIdx.x * BLOCK + threadIdx.x; i0 < n4; i0 += stride * U) {
        f4v vx[U], vxd[U], vth[U], vthd[U], vr[U];
        i4v va[U];
        u4v vs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * stride;
            if (i < n4) {
                vx[u] = ld<NTL>(x + i); vxd[u] = ld<NTL>(xd + i); vth[u] = ld<NTL>(th + i); vthd[u] = ld<NTL>(thd + i);
                va[u] = ld<NTL>(action + i); vr[u] = ld<NTL>(ret + i); vs[u] = ld<NTL>(status + i);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * stride;
            if (i < n4) {
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    float ex = vx[u][l], exd = vxd[u][l], eth = vth[u][l], ethd = vthd[u][l], er = vr[u][l];
                    uint32_t es = vs[u][l];
                    step_one(ex, exd, eth, ethd, va[u][l], er, es, max_step);
                    vx[u][l] = ex; vxd[u][l] = exd; vth[u][l] = eth; vthd[u][l] = ethd; vr[u][l] = er; vs[u][l] = es;
                }
                st<NTS>(x + i, vx[u]); st<NTS>(xd + i, vxd[u]); st<NTS>(th + i, vth[u]); st<NTS>(thd + i, vthd[u]);
                st<NTS>(ret + i, vr[u]); st<NTS>(status + i, vs[u]);
            }
        }
    }
}

// contiguous-chunk persistent form: workgroup b owns the contiguous range [b * per, (b + 1) * per) of float4 groups
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_step_chunk(int n4, int max_step, f4v *x, f4v *xd, f4v *th, f4v *thd,
                                                      const i4v *action, f4v *ret, u4v *status)
{
    const int per = (n4 + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(n4, lo + per);
    for (int i = lo + threadIdx.x; i < hi; i += BLOCK) {
        f4v vx = ld<true>(x + i), vxd = ld<true>(xd + i), vth = ld<true>(th + i), vthd = ld<true>(thd + i), vr = ld<true>(ret + i);
        const i4v va = ld<true>(action + i);
        u4v vs = ld<true>(status + i);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            float ex = vx[l], exd = vxd[l], eth = vth[l], ethd = vthd[l], er = vr[l];
            uint32_t es = vs[l];
            step_one(ex, exd, eth, ethd, va[l], er, es, max_step);
            vx[l] = ex; vxd[l] = exd; vth[l] = eth; vthd[l] = ethd; vr[l] = er; vs[l] = es;
        }
        st<true>(x + i, vx); st<true>(xd + i, vxd); st<true>(th + i, vth); st<true>(thd + i, vthd);
        st<true>(ret + i, vr); st<true>(status + i, vs);
    }
}

// same traffic, no arithmetic: 7 read streams, 6 write streams
template <bool NT, int U, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_copy13(int n4, int, f4v *x, f4v *xd, f4v *th, f4v *thd, const i4v *action, f4v *ret,
                                                  u4v *status)
{
    const int stride = gridDim.x * BLOCK;
    for (int i0 = blockIdx.x * BLOCK + threadIdx.x; i0 < n4; i0 += stride * U) {
        f4v a[U], b[U], c[U], d[U], f[U];
        i4v e[U];
        u4v g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * stride;
            if (i < n4) {
                a[u] = ld<NT>(x + i); b[u] = ld<NT>(xd + i); c[u] = ld<NT>(th + i); d[u] = ld<NT>(thd + i);
                e[u] = ld<NT>(action + i); f[u] = ld<NT>(ret + i); g[u] = ld<NT>(status + i);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * stride;
            if (i < n4) {
                a[u].x += (float)e[u].x;                         // the action stream must be consumed
                st<NT>(x + i, a[u]); st<NT>(xd + i, b[u]); st<NT>(th + i, c[u]); st<NT>(thd + i, d[u]);
                st<NT>(ret + i, f[u]); st<NT>(status + i, g[u]);
            }
        }
    }
}

// Out of place: the six arrays that are written go to a second set (ping-pong), nothing is written where it was read.
template <bool ARITH, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_step_oop(int n4, int max_step, const f4v *x, const f4v *xd, const f4v *th, const f4v *thd,
                                                    const i4v *action, const f4v *ret, const u4v *status, f4v *ox, f4v *oxd,
                                                    f4v *oth, f4v *othd, f4v *oret, u4v *ostatus)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n4) return;
    f4v vx = ld<true>(x + i), vxd = ld<true>(xd + i), vth = ld<true>(th + i), vthd = ld<true>(thd + i), vr = ld<true>(ret + i);
    i4v va = ld<true>(action + i);
    u4v vs = ld<true>(status + i);
    if constexpr (ARITH) {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            float ex = vx[l], exd = vxd[l], eth = vth[l], ethd = vthd[l], er = vr[l];
            uint32_t es = vs[l];
            step_one(ex, exd, eth, ethd, va[l], er, es, max_step);
            vx[l] = ex; vxd[l] = exd; vth[l] = eth; vthd[l] = ethd; vr[l] = er; vs[l] = es;
        }
    } else {
        asm volatile("" : "+v"(vx), "+v"(vxd), "+v"(vth), "+v"(vthd), "+v"(vr), "+v"(vs));
        asm volatile("" ::"v"(va));
    }
    st<true>(ox + i, vx); st<true>(oxd + i, vxd); st<true>(oth + i, vth); st<true>(othd + i, vthd); st<true>(oret + i, vr);
    st<true>(ostatus + i, vs);
}

// A plain copy with the env step's instruction mix: 7 loads and 6 stores of 16 bytes per lane, src -> dst, each workgroup a
// contiguous chunk (tells apart "thirteen memory instructions per lane" from "thirteen address streams")
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_copy_7_6(long n_chunks, const f4v *src, f4v *dst)
{
    const long b = blockIdx.x;
    if (b >= n_chunks) return;
    const f4v *s = src + b * 7 * BLOCK;
    f4v *d = dst + b * 6 * BLOCK;
    f4v v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = ld<true>(s + k * BLOCK + threadIdx.x);
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
    asm volatile("" ::"v"(v[6]));
#pragma unroll
    for (int k = 0; k < 6; ++k) st<true>(d + k * BLOCK + threadIdx.x, v[k]);
}

// the product kernel with a dummy LDS allocation that limits the workgroups per CU (memory-level parallelism)
template <int LDS_BYTES, int BLOCK = 256>
__global__ __launch_bounds__(BLOCK) void k_step_occ(int n4, int max_step, f4v *x, f4v *xd, f4v *th, f4v *thd, const i4v *action,
                                                  f4v *ret, u4v *status)
{
    __shared__ char pad[LDS_BYTES];
    if (n4 < 0) pad[threadIdx.x] = 0;                                       // keeps the allocation
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n4) return;
    f4v vx = ld<true>(x + i), vxd = ld<true>(xd + i), vth = ld<true>(th + i), vthd = ld<true>(thd + i), vr = ld<true>(ret + i);
    i4v va = ld<true>(action + i);
    u4v vs = ld<true>(status + i);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        float ex = vx[l], exd = vxd[l], eth = vth[l], ethd = vthd[l], er = vr[l];
        uint32_t es = vs[l];
        step_one(ex, exd, eth, ethd, va[l], er, es, max_step);
        vx[l] = ex; vxd[l] = exd; vth[l] = eth; vthd[l] = ethd; vr[l] = er; vs[l] = es;
    }
    st<true>(x + i, vx); st<true>(xd + i, vxd); st<true>(th + i, vth); st<true>(thd + i, vthd); st<true>(ret + i, vr);
    st<true>(status + i, vs);
}

// Tiled SoA: the seven arrays of G * BLOCK * 4 envs lie behind one another in one contiguous tile (7 x G x BLOCK float4
// groups); a workgroup owns a tile.  Same 52 bytes per env-step, but a launch is one sequential read stream and one
// sequential write stream over the allocation instead of seven + six concurrent ones.  ARITH = false: the copy ceiling.
template <bool ARITH, int G, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_step_tiled(int n_tiles, int max_step, f4v *tiles)
{
    constexpr int SEG = G * BLOCK;                                          // float4 groups per array per tile
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        f4v *base = tiles + (size_t)t * 7 * SEG;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i = g * BLOCK + threadIdx.x;
            f4v vx = ld<true>(base + i), vxd = ld<true>(base + SEG + i), vth = ld<true>(base + 2 * SEG + i),
                vthd = ld<true>(base + 3 * SEG + i), vr = ld<true>(base + 5 * SEG + i);
            i4v va = ld<true>((const i4v *)(base + 4 * SEG) + i);
            u4v vs = ld<true>((const u4v *)(base + 6 * SEG) + i);
            if constexpr (ARITH) {
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    float ex = vx[l], exd = vxd[l], eth = vth[l], ethd = vthd[l], er = vr[l];
                    uint32_t es = vs[l];
                    step_one(ex, exd, eth, ethd, va[l], er, es, max_step);
                    vx[l] = ex; vxd[l] = exd; vth[l] = eth; vthd[l] = ethd; vr[l] = er; vs[l] = es;
                }
            } else {
                asm volatile("" : "+v"(vx), "+v"(vxd), "+v"(vth), "+v"(vthd), "+v"(vr), "+v"(vs));
                asm volatile("" ::"v"(va));
            }
            st<true>(base + i, vx); st<true>(base + SEG + i, vxd); st<true>(base + 2 * SEG + i, vth);
            st<true>(base + 3 * SEG + i, vthd); st<true>(base + 5 * SEG + i, vr); st<true>((u4v *)(base + 6 * SEG) + i, vs);
        }
    }
}

template <bool NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_copy2(long n4, const f4v *src, f4v *dst)
{
    const long stride = (long)gridDim.x * BLOCK;
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += stride) st<NT>(dst + i, ld<NT>(src + i));
}

struct Variant {
    std::string name;
    std::function<void()> launch;
    double bytes;
    std::vector<float> us;
};

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 24;
    const int rounds = argc > 2 ? atoi(argv[2]) : 15;
    const int launches = argc > 3 ? atoi(argv[3]) : 20;
    const int preroll = argc > 4 ? atoi(argv[4]) : 200;
    const int n = 1 << lg, n4 = n / 4;
    // the product's allocation: seven arrays carved out of one allocation with a 4 KiB skew (HipES.alloc_env_soa)
    const size_t stride = ((size_t)n * 4 + 4096 + 15) / 16 * 16;
    char *pool;
    CK(hipMalloc(&pool, stride * 7));
    std::vector<float> h(n);
    for (int k = 0; k < 7; ++k) {
        for (int i = 0; i < n; ++i) h[i] = k < 4 ? (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f * 0.1f - 0.05f : 0.0f;
        if (k == 4) for (int i = 0; i < n; ++i) { int a = (i * 40503u >> 7) & 1; h[i] = *(float *)&a; }
        CK(hipMemcpy(pool + k * stride, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    }
    f4v *x = (f4v *)(pool + 0 * stride), *xd = (f4v *)(pool + 1 * stride), *th = (f4v *)(pool + 2 * stride),
        *thd = (f4v *)(pool + 3 * stride), *rt = (f4v *)(pool + 5 * stride);
    const i4v *ac = (const i4v *)(pool + 4 * stride);
    u4v *stt = (u4v *)(pool + 6 * stride);
    const double bytes = 52.0 * n;
    f4v *c_src, *c_dst;
    CK(hipMalloc(&c_src, (size_t)(bytes / 2)));
    CK(hipMalloc(&c_dst, (size_t)(bytes / 2)));
    CK(hipMemset(c_src, 1, (size_t)(bytes / 2)));
    const long c4 = (long)(bytes / 2 / 16);

    std::vector<Variant> vs;
#define ADD(NAME, GRID, BLOCK, KERNEL)                                                                                  \
    vs.push_back({NAME, [=] { hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), 0, 0, n4, 0, x, xd, th, thd, ac, rt, stt); }, bytes, {}})
    vs.push_back({"PRODUCT: nt, 16 B/lane, one-shot, b64, 22 KB LDS reserved (7 waves per CU)", [=] { hipLaunchKernelGGL((k_step<true, true, 1, 64, 0>), dim3(n4 / 64), dim3(64), 22528, 0, n4, 0, x, xd, th, thd, ac, rt, stt); }, bytes, {}});
    ADD("until round 3: nt, 16 B/lane, one-shot, b256, unlimited waves", n4 / 256, 256, (k_step<true, true, 1, 256, 0>));
    ADD("2 x 16 B/lane, one-shot, b256", n4 / 512, 256, (k_step<true, true, 2, 256, 0>));
    ADD("persistent 256 CU x 8 wg, 16 B", 2048, 256, (k_step<true, true, 1, 256, 0>));
    ADD("persistent 256 CU x 8 wg, 2 x 16 B", 2048, 256, (k_step<true, true, 2, 256, 0>));
    ADD("persistent 256 CU x 4 wg, 2 x 16 B", 1024, 256, (k_step<true, true, 2, 256, 0>));
    ADD("persistent 256 CU x 16 wg, 16 B", 4096, 256, (k_step<true, true, 1, 256, 0>));
    ADD("persistent contiguous chunks, 2048 wg", 2048, 256, (k_step_chunk<256>));
    ADD("product + s_setprio 3", n4 / 256, 256, (k_step<true, true, 1, 256, 3>));
    ADD("2 x 16 B + s_setprio 3, one-shot", n4 / 512, 256, (k_step<true, true, 2, 256, 3>));
    ADD("one-shot, b512", n4 / 512, 512, (k_step<true, true, 1, 512, 0>));
    ADD("one-shot, b1024", n4 / 1024, 1024, (k_step<true, true, 1, 1024, 0>));
    ADD("one-shot, b64", n4 / 64, 64, (k_step<true, true, 1, 64, 0>));
    ADD("plain loads, nt stores", n4 / 256, 256, (k_step<false, true, 1, 256, 0>));
    ADD("nt loads, plain stores", n4 / 256, 256, (k_step<true, false, 1, 256, 0>));
    ADD("plain loads, plain stores", n4 / 256, 256, (k_step<false, false, 1, 256, 0>));
    ADD("CEILING copy13 nt one-shot b256", n4 / 256, 256, (k_copy13<true, 1, 256>));
    ADD("CEILING copy13 nt 2 x 16 B one-shot", n4 / 512, 256, (k_copy13<true, 2, 256>));
    ADD("CEILING copy13 nt persistent 2048 wg", 2048, 256, (k_copy13<true, 1, 256>));
    // out of place: a second pool for the six written arrays
    char *pool2;
    CK(hipMalloc(&pool2, stride * 7));
    CK(hipMemset(pool2, 0, stride * 7));
    f4v *ox = (f4v *)(pool2 + 0 * stride), *oxd = (f4v *)(pool2 + 1 * stride), *oth = (f4v *)(pool2 + 2 * stride),
        *othd = (f4v *)(pool2 + 3 * stride), *ort = (f4v *)(pool2 + 5 * stride);
    u4v *ostt = (u4v *)(pool2 + 6 * stride);
    vs.push_back({"OUT OF PLACE product (second set of arrays)", [=] { hipLaunchKernelGGL((k_step_oop<true, 256>), dim3(n4 / 256), dim3(256), 0, 0, n4, 0, x, xd, th, thd, ac, rt, stt, ox, oxd, oth, othd, ort, ostt); }, bytes, {}});
    vs.push_back({"CEILING out-of-place copy13", [=] { hipLaunchKernelGGL((k_step_oop<false, 256>), dim3(n4 / 256), dim3(256), 0, 0, n4, 0, x, xd, th, thd, ac, rt, stt, ox, oxd, oth, othd, ort, ostt); }, bytes, {}});
    {
        f4v *s7, *d6;
        CK(hipMalloc(&s7, (size_t)n * 28));
        CK(hipMalloc(&d6, (size_t)n * 24));
        CK(hipMemset(s7, 1, (size_t)n * 28));
        const long chunks = n4 / 256;
        vs.push_back({"CEILING copy 7 loads + 6 stores per lane, chunks", [=] { hipLaunchKernelGGL((k_copy_7_6<256>), dim3((unsigned)chunks), dim3(256), 0, 0, chunks, (const f4v *)s7, d6); }, bytes, {}});
    }
    ADD("product, 40 KB LDS pad (4 wg per CU)", n4 / 256, 256, (k_step_occ<40960>));
    ADD("product, 20 KB LDS pad (8 wg per CU)", n4 / 256, 256, (k_step_occ<20480>));
    ADD("product, 64 KB LDS pad (2 wg per CU)", n4 / 256, 256, (k_step_occ<65536>));
    ADD("product, 53 KB LDS pad (3 wg per CU)", n4 / 256, 256, (k_step_occ<54272>));
    ADD("product, 64 KB pad, b128 (2 wg = 4 waves per CU)", n4 / 128, 128, (k_step_occ<65536, 128>));
    ADD("product, 64 KB pad, b64 (2 wg = 2 waves per CU)", n4 / 64, 64, (k_step_occ<65536, 64>));
    ADD("product, 64 KB pad, b512 (2 wg = 16 waves per CU)", n4 / 512, 512, (k_step_occ<65536, 512>));
    ADD("product, 64 KB pad, b1024 (2 wg = 32 waves/CU)", n4 / 1024, 1024, (k_step_occ<65536, 1024>));
    ADD("product, 40 KB pad, b128 (4 wg = 8 waves per CU)", n4 / 128, 128, (k_step_occ<40960, 128>));
    ADD("product, 20 KB pad, b64 (8 wg = 8 waves per CU)", n4 / 64, 64, (k_step_occ<20480, 64>));
    // occupancy by dynamic LDS (nothing in the kernel reads it): workgroups per CU = 160 KB / bytes
#define ADDL(NAME, BLOCK, LDS)                                                                                          \
    vs.push_back({NAME, [=] { hipLaunchKernelGGL((k_step<true, true, 1, BLOCK, 0>), dim3(n4 / BLOCK), dim3(BLOCK), LDS, 0, n4, 0, x, xd, th, thd, ac, rt, stt); }, bytes, {}})
    ADDL("dyn LDS: b64, 32 KB (5 waves per CU)", 64, 32768);
    ADDL("dyn LDS: b64, 26 KB (6 waves per CU)", 64, 26624);
    ADDL("dyn LDS: b64, 22.8 KB (7 waves per CU)", 64, 23296);
    ADDL("dyn LDS: b64, 20 KB (8 waves per CU)", 64, 20480);
    ADDL("dyn LDS: b64, 17.7 KB (9 waves per CU)", 64, 18176);
    ADDL("dyn LDS: b64, 16 KB (10 waves per CU)", 64, 16384);
    ADDL("dyn LDS: b64, 13.3 KB (12 waves per CU)", 64, 13568);
    ADDL("dyn LDS: b64, 10 KB (16 waves per CU)", 64, 10240);
    ADDL("dyn LDS: b128, 40 KB (4 wg = 8 waves)", 128, 40960);
    ADDL("dyn LDS: b128, 32 KB (5 wg = 10 waves)", 128, 32768);
    ADDL("dyn LDS: b256, 64 KB (2 wg = 8 waves)", 256, 65536);
    vs.push_back({"dyn LDS: out of place b64, 20 KB (8 waves)", [=] { hipLaunchKernelGGL((k_step_oop<true, 64>), dim3(n4 / 64), dim3(64), 20480, 0, n4, 0, x, xd, th, thd, ac, rt, stt, ox, oxd, oth, othd, ort, ostt); }, bytes, {}});
    vs.push_back({"dyn LDS: CEILING copy13 b64, 20 KB (8 waves)", [=] { hipLaunchKernelGGL((k_copy13<true, 1, 64>), dim3(n4 / 64), dim3(64), 20480, 0, n4, 0, x, xd, th, thd, ac, rt, stt); }, bytes, {}});
    // tiled layout: one allocation of n / tile_envs tiles
    f4v *tiles;
    CK(hipMalloc(&tiles, (size_t)n * 28));
    CK(hipMemset(tiles, 0, (size_t)n * 28));
#define ADDT(NAME, ARITH, G, BLOCK, GRID)                                                                                 \
    vs.push_back({NAME, [=] { hipLaunchKernelGGL((k_step_tiled<ARITH, G, BLOCK>), dim3(GRID), dim3(BLOCK), 0, 0, n4 / (G * BLOCK), 0, tiles); }, bytes, {}})
    ADDT("TILED 1024 envs/tile, b256, one-shot", true, 1, 256, n4 / 256);
    ADDT("TILED 256 envs/tile, b64, one-shot", true, 1, 64, n4 / 64);
    ADDT("TILED 4096 envs/tile, b256 x 4, one-shot", true, 4, 256, n4 / 1024);
    ADDT("TILED 4096 envs/tile, b1024, one-shot", true, 1, 1024, n4 / 1024);
    ADDT("TILED 1024 envs/tile, persistent 2048 wg", true, 1, 256, 2048);
    ADDT("CEILING tiled copy 1024 envs/tile", false, 1, 256, n4 / 256);
    ADDT("CEILING tiled copy 4096 envs/tile b256 x 4", false, 4, 256, n4 / 1024);
    vs.push_back({"CEILING copy2 nt (26 B in, 26 B out per env)", [=] { hipLaunchKernelGGL((k_copy2<true, 256>), dim3((unsigned)(c4 / 256)), dim3(256), 0, 0, c4, (const f4v *)c_src, c_dst); }, bytes, {}});
    vs.push_back({"CEILING copy2 plain", [=] { hipLaunchKernelGGL((k_copy2<false, 256>), dim3((unsigned)(c4 / 256)), dim3(256), 0, 0, c4, (const f4v *)c_src, c_dst); }, bytes, {}});
    vs.push_back({"CEILING hipMemcpyDtoD same bytes", [=] { (void)hipMemcpyAsync(c_dst, c_src, (size_t)(bytes / 2), hipMemcpyDeviceToDevice, 0); }, bytes, {}});

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < preroll; ++i) vs[0].launch();                       // clocks and memory system up
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; ++r) {
        for (auto &v : vs) {
            v.launch();                                                     // one untimed launch of this variant
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < launches; ++i) v.launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            v.us.push_back(ms * 1e3f / launches);
        }
    }
    CK(hipGetLastError());
    printf("n = 2^%d envs, %.1f MB per launch, %d rounds x %d launches, pre-roll %d\n", lg, bytes / 1e6, rounds, launches, preroll);
    printf("%-64s %9s %9s %9s %9s %8s\n", "variant", "med us", "min us", "max us", "med GB/s", "frac 8T");
    for (auto &v : vs) {
        std::vector<float> s = v.us;
        std::sort(s.begin(), s.end());
        const float med = s[s.size() / 2];
        printf("%-64s %9.2f %9.2f %9.2f %9.1f %8.4f\n", v.name.c_str(), med, s.front(), s.back(), v.bytes / med / 1e3,
               v.bytes / med / 1e3 / 8000.0);
    }
    return 0;
}
