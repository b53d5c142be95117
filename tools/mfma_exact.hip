// mfma_exact.hip -- development check: is a chain of v_mfma_f32_16x16x4_f32 accumulations bit-identical to the
// k-ascending fmaf chain  acc = c; acc = fmaf(a[k], b[k], acc)  that the canonical GRU arithmetic uses?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/mfma_exact.hip -o tools/mfma_exact
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A[16][K], B[K][16], C[16][16] -> D[16][16], one wave
__global__ void k(const float *A, const float *B, const float *C, int K, float *D)
{
    const int l = threadIdx.x, li = l & 15, lg = l >> 4;
    f32x4 d;
    for (int r = 0; r < 4; ++r) d[r] = C[(4 * lg + r) * 16 + li];
    for (int kb = 0; kb < K / 4; ++kb) {
        const float a = A[li * K + 4 * kb + lg];
        const float b = B[(4 * kb + lg) * 16 + li];
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * lg + r) * 16 + li] = d[r];
}

int main()
{
    const int K = 16, trials = 2000;
    std::vector<float> A(16 * K), B(K * 16), C(256), D(256);
    float *dA, *dB, *dC, *dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 1024)); CK(hipMalloc(&dD, 1024));
    unsigned s = 1;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    long long mism_chain = 0, mism_tree = 0, total = 0;
    for (int t = 0; t < trials; ++t) {
        const float sc = t % 3 == 0 ? 1e-3f : (t % 3 == 1 ? 1.0f : 37.0f);
        for (auto &v : A) v = rnd() * sc;
        for (auto &v : B) v = rnd() * 2.0f;
        for (auto &v : C) v = t % 2 ? rnd() * 0.3f : 0.0f;
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, dD);
        CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float chain = C[i * 16 + j];
                for (int k2 = 0; k2 < K; ++k2) chain = fmaf(A[i * K + k2], B[k2 * 16 + j], chain);
                // alternative: each group of 4 summed exactly then added (what a "dot4" unit would do)
                float tree = C[i * 16 + j];
                for (int kb = 0; kb < K / 4; ++kb) {
                    double acc = 0;
                    for (int q = 0; q < 4; ++q) acc += (double)A[i * K + 4 * kb + q] * (double)B[(4 * kb + q) * 16 + j];
                    tree = (float)((double)tree + acc);
                }
                const float got = D[i * 16 + j];
                mism_chain += memcmp(&got, &chain, 4) != 0;
                mism_tree += memcmp(&got, &tree, 4) != 0;
                ++total;
            }
    }
    printf("v_mfma_f32_16x16x4_f32 over K = %d: %lld outputs, %lld differ from the fmaf chain, %lld differ from dot4-then-add\n",
           K, total, mism_chain, mism_tree);
    return 0;
}
