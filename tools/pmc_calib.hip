// pmc_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access patterns of k_step
// (MI355X_MICROARCH.md, HBM section: "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Each kernel touches every 64-byte line of a 1 GiB table exactly once, in scattered order
// (line = thread * odd constant mod 2^24), so the bytes that must cross the memory side are known: 2^24 lines x 64 B.
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o gpurun_out/pmc_calib
//   rocprofv3 --pmc FETCH_SIZE -d gpurun_out/calib_fetch -o pmc -- gpurun_out/pmc_calib     (and WRITE_SIZE in a run of its own)
//   python tools/pmc_calib.py gpurun_out
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define LINES (1u << 24)
__device__ __forceinline__ size_t line_of() { return (size_t)(((blockIdx.x * 256u + threadIdx.x) * 2654435761u) & (LINES - 1u)); }
// the EDT-line read: one whole 64-byte line per lane (four 16-byte loads)
__global__ void calib_line64_read(const ulonglong2 *tab, unsigned long long *out) {
    const ulonglong2 *p = tab + 4 * line_of();
    ulonglong2 a = p[0], b = p[1], c = p[2], d = p[3];
    if ((a.x ^ b.x ^ c.x ^ d.x ^ a.y ^ b.y ^ c.y ^ d.y) == 0x123456789ULL) out[0] = 1;
}
// the waiter-record read: one 8-byte word per lane, each in a line of its own
__global__ void calib_word8_read(const unsigned long long *tab, unsigned long long *out) {
    unsigned long long v = tab[8 * line_of()];
    if (v == 0x123456789ULL) out[0] = 1;
}
// the slot count: one 4-byte device-scope atomic per lane, each in a line of its own, result used
__global__ void calib_atomic4_returning(unsigned int *tab, unsigned long long *out) {
    unsigned int old = atomicSub(&tab[16 * line_of()], 1u);
    if (old == 0x12345u) out[0] = 1;
}
__global__ void calib_atomic4_noreturn(unsigned int *tab) { atomicSub(&tab[16 * line_of()], 1u); }
// a scattered 4-byte store, each in a line of its own
__global__ void calib_store4(unsigned int *tab) { tab[16 * line_of()] = threadIdx.x; }
// a coalesced 4-byte store stream (the ready-log writes)
__global__ void calib_store4_stream(unsigned int *tab) { tab[blockIdx.x * 256u + threadIdx.x] = threadIdx.x; }
int main() {
    void *tab; unsigned long long *out;
    CHK(hipMalloc(&tab, (size_t)LINES * 64)); CHK(hipMalloc((void **)&out, 64));
    CHK(hipMemset(tab, 0, (size_t)LINES * 64)); CHK(hipMemset(out, 0, 64));
    CHK(hipDeviceSynchronize());
    const unsigned grid = LINES / 256;
    calib_line64_read<<<grid, 256>>>((const ulonglong2 *)tab, out); CHK(hipDeviceSynchronize());
    calib_word8_read<<<grid, 256>>>((const unsigned long long *)tab, out); CHK(hipDeviceSynchronize());
    calib_atomic4_returning<<<grid, 256>>>((unsigned int *)tab, out); CHK(hipDeviceSynchronize());
    calib_atomic4_noreturn<<<grid, 256>>>((unsigned int *)tab); CHK(hipDeviceSynchronize());
    calib_store4<<<grid, 256>>>((unsigned int *)tab); CHK(hipDeviceSynchronize());
    calib_store4_stream<<<grid, 256>>>((unsigned int *)tab); CHK(hipDeviceSynchronize());
    printf("pmc_calib: 6 kernels over %u lines (%.0f MiB)\n", LINES, LINES * 64.0 / (1 << 20));
    return 0;
}
