// Development probe: which XCD runs workgroup (x, y) of a two-dimensional grid?  Prints the XCC_ID hardware register per workgroup.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/xcd_map_probe tools/xcd_map_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void who(int* out)
{
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);   // HW_REG_XCC_ID, bits 0..3
        out[blockIdx.y * gridDim.x + blockIdx.x] = (int)xcc;
    }
}
int main()
{
    for (auto g : {dim3(24, 1), dim3(24, 8), dim3(88, 8), dim3(26, 4), dim3(6, 128)}) {
        const int n = g.x * g.y;
        int* d; hipMalloc(&d, n * 4); hipMemset(d, 0xff, n * 4);
        hipLaunchKernelGGL(who, g, dim3(512), 0, 0, d);
        std::vector<int> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        int by_linear = 0, by_x = 0;
        for (int i = 0; i < n; ++i) { by_linear += h[i] == i % 8; by_x += h[i] == (i % g.x) % 8; }
        printf("grid %u x %u: XCD == linear %% 8 for %d of %d workgroups, == x %% 8 for %d\n  row 0:", g.x, g.y, by_linear, n, by_x);
        for (unsigned x = 0; x < g.x && x < 32; ++x) printf(" %d", h[x]);
        if (g.y > 1) { printf("\n  row 1:"); for (unsigned x = 0; x < g.x && x < 32; ++x) printf(" %d", h[g.x + x]); }
        printf("\n");
        hipFree(d);
    }
    return 0;
}
