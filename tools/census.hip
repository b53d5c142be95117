// census.hip -- development tool: where does the dispatcher put G workgroups of B threads?
// Every wave records (XCC_ID, HW_ID) while spinning on ALU work long enough for the whole grid to be
// co-resident; the host prints waves per CU / per SIMD histograms.
//   hipcc --offload-arch=gfx950 -O3 tools/census.hip -o tools/census ; ./tools/census <grid> <block>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void k_census(uint32_t *out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = __builtin_fmaf(a, b, 1e-7f);
    if ((threadIdx.x & 63) == 0) {
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const int w = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
        out[2 * w] = hwid;
        out[2 * w + 1] = xcc;
    }
    if (a == 12345.678f) out[0] = 0;  // keep the loop alive
}

int main(int argc, char **argv)
{
    const int grid = argc > 1 ? atoi(argv[1]) : 1280, block = argc > 2 ? atoi(argv[2]) : 64;
    const int iters = argc > 3 ? atoi(argv[3]) : 200000;
    const int waves = grid * block / 64;
    uint32_t *d;
    hipMalloc(&d, waves * 8);
    hipLaunchKernelGGL(k_census, dim3(grid), dim3(block), 0, 0, d, iters);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(2 * waves);
    hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
    std::map<uint32_t, int> per_cu, per_simd;
    for (int w = 0; w < waves; ++w) {
        const uint32_t hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
        // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
        const uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const uint32_t cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        per_cu[cukey]++;
        per_simd[(cukey << 2) | simd]++;
    }
    std::map<int, int> hist_cu, hist_simd;
    for (auto &kv : per_cu) hist_cu[kv.second]++;
    for (auto &kv : per_simd) hist_simd[kv.second]++;
    printf("grid %d x block %d = %d waves: %zu CUs used, %zu SIMDs used\n", grid, block, waves, per_cu.size(), per_simd.size());
    printf("  waves/CU histogram:");
    for (auto &kv : hist_cu) printf("  %d waves: %d CUs;", kv.first, kv.second);
    printf("\n  waves/SIMD histogram:");
    for (auto &kv : hist_simd) printf("  %d waves: %d SIMDs;", kv.first, kv.second);
    printf("\n");
    return 0;
}
