// Which SIMD does wave w of a workgroup land on?  (HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13])
// Build + run: hipcc --offload-arch=gfx950 tools/probe/simd_map.hip -o gpurun_out/simd_map && gpurun_out/simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}
int main() {
    for (int threads : {256, 512, 768, 1024}) {
        const int nw = threads / 64, blocks = 512;
        unsigned* d; hipMalloc(&d, blocks * nw * 4);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d);
        unsigned h[512 * 16]; hipMemcpy(h, d, blocks * nw * 4, hipMemcpyDeviceToHost);
        printf("threads %d: simd of waves 0..%d, first 6 workgroups:\n", threads, nw - 1);
        for (int b = 0; b < 6; ++b) { for (int w = 0; w < nw; ++w) printf(" %u", (h[b * nw + w] >> 4) & 3); printf("   (cu %u se %u)\n", (h[b * nw] >> 8) & 15, (h[b * nw] >> 13) & 7); }
        int hist[64] = {0};   // pattern histogram: does wave w always sit on simd (w % 4)?
        int rr = 0;
        for (int b = 0; b < blocks; ++b) { bool ok = true; for (int w = 0; w < nw; ++w) ok &= (((h[b * nw + w] >> 4) & 3) == (unsigned)((w + ((h[b * nw] >> 4) & 3)) % 4)); rr += ok; }
        printf("  workgroups whose wave w sits on simd (simd(wave0) + w) %% 4: %d of %d\n", rr, blocks);
        hipFree(d);
    }
    return 0;
}
