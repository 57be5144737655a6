// How much host time does a level's chain of small dependent kernels cost: 9 direct launches + 1 synchronisation, against one
// hipGraphLaunch of the same 9 nodes (+ one hipGraphExecKernelNodeSetParams) + 1 synchronisation.
//   hipcc --offload-arch=gfx950 -O2 tools/graphbench.hip -o /tmp/graphbench && /tmp/graphbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_small(unsigned *p, unsigned k) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += k; }
__global__ void k_wide(unsigned *p, unsigned k) { unsigned i = blockIdx.x * blockDim.x + threadIdx.x; p[16 + (i & 1023)] = k + i; }
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main() {
    unsigned *p; CHK(hipMalloc(&p, 1 << 16)); CHK(hipMemset(p, 0, 1 << 16));
    hipStream_t st; CHK(hipStreamCreate(&st));
    const int N = 2000, NK = 9;
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int i = 0; i < N; ++i) {
            for (int k = 0; k < NK; ++k) { if (k >= 1 && k <= 3) k_wide<<<256, 256, 0, st>>>(p, i); else k_small<<<1, 64, 0, st>>>(p, i); }
            CHK(hipStreamSynchronize(st));
        }
        printf("direct: %d launches + sync: %.1f us per iteration\n", NK, 1e6 * (now() - t0) / N);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < NK; ++k) { if (k >= 1 && k <= 3) k_wide<<<256, 256, 0, st>>>(p, 0); else k_small<<<1, 64, 0, st>>>(p, 1); }
    CHK(hipStreamEndCapture(st, &g));
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    size_t nn = 0; CHK(hipGraphGetNodes(g, nullptr, &nn));
    hipGraphNode_t nodes[16]; CHK(hipGraphGetNodes(g, nodes, &nn));
    hipGraphNode_t root[16]; size_t nr = 16; CHK(hipGraphGetRootNodes(g, root, &nr));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int i = 0; i < N; ++i) { CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st)); }
        printf("graph: launch + sync: %.1f us per iteration (%zu nodes)\n", 1e6 * (now() - t0) / N, nn);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int i = 0; i < N; ++i) {
            unsigned kk = (unsigned)i; void *args[2] = {&p, &kk};
            hipKernelNodeParams np; memset(&np, 0, sizeof np);
            np.func = (void *)k_small; np.gridDim = dim3(1); np.blockDim = dim3(64); np.kernelParams = args;
            CHK(hipGraphExecKernelNodeSetParams(ge, root[0], &np));
            CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st));
        }
        printf("graph: set params of one node + launch + sync: %.1f us per iteration\n", 1e6 * (now() - t0) / N);
    }
    unsigned h; CHK(hipMemcpy(&h, p, 4, hipMemcpyDeviceToHost)); printf("check %u\n", h);
    return 0;
}
