// Calibration microbenchmarks for the dependence engine's critical path (not product code):
// dependent-load latency, returning-atomic latency, kernel boundary, ticket fan-in.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__global__ void k_chase(const uint32_t* next, uint32_t start, int hops, uint32_t* out, unsigned long long* cyc) {
    uint32_t p = start + threadIdx.x + blockIdx.x * blockDim.x;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < hops; ++i) p = next[p];
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x + blockIdx.x * blockDim.x] = p;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ void k_atomic_chain(uint32_t* ctr, uint32_t mask, int hops, uint32_t* out, unsigned long long* cyc) {
    uint32_t p = (threadIdx.x + blockIdx.x * blockDim.x) * 2654435761u;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < hops; ++i) { uint32_t o = atomicAdd(&ctr[p & mask], 1u); p = p * 1664525u + 1013904223u + o; }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x + blockIdx.x * blockDim.x] = p;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ void k_empty(uint32_t* p) { if (p == nullptr) p[0] = 1; }
__global__ void k_ticket(uint32_t* t, uint32_t* flag) { if (threadIdx.x == 0) { uint32_t v = atomicAdd(t, 1u); if (v == gridDim.x - 1) { *t = 0; *flag += 1; } } }
__global__ void k_ticket_fence(uint32_t* t, uint32_t* flag) { if (threadIdx.x == 0) { __threadfence(); uint32_t v = atomicAdd(t, 1u); if (v == gridDim.x - 1) { *t = 0; *flag += 1; } } }
// scattered atomic throughput: each thread does `per` atomics on random counters
__global__ void k_atomic_tp(uint32_t* ctr, uint32_t mask, int per) {
    uint32_t p = (threadIdx.x + blockIdx.x * blockDim.x) * 2654435761u;
    for (int i = 0; i < per; ++i) { p = p * 1664525u + 1013904223u; atomicSub(&ctr[(p >> 4) & mask], 1u); }
}
__global__ void k_atomic_tp_ret(uint32_t* ctr, uint32_t mask, int per, uint32_t* out) {
    uint32_t p = (threadIdx.x + blockIdx.x * blockDim.x) * 2654435761u, acc = 0;
    for (int i = 0; i < per; ++i) { p = p * 1664525u + 1013904223u; acc += atomicSub(&ctr[(p >> 4) & mask], 1u) == 1u; }
    if (acc == 0xFFFFFFFF) out[0] = acc;
}
int main() {
    const uint32_t N = 1u << 26; // 256 MB of u32: beyond L2, at the edge of the Infinity Cache
    std::vector<uint32_t> h(N);
    uint64_t s = 88172645463325252ull;
    for (uint32_t i = 0; i < N; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % N); }
    uint32_t *next, *out, *ctr, *flag; unsigned long long* cyc;
    CHK(hipMalloc(&next, 4ull * N)); CHK(hipMalloc(&out, 4ull << 20)); CHK(hipMalloc(&ctr, 4ull * N)); CHK(hipMalloc(&flag, 64)); CHK(hipMalloc(&cyc, 8));
    CHK(hipMemcpy(next, h.data(), 4ull * N, hipMemcpyHostToDevice)); CHK(hipMemset(ctr, 0, 4ull * N)); CHK(hipMemset(flag, 0, 64));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    unsigned long long hc; float ms;
    for (int blocks : {1, 40, 256}) {
        k_chase<<<blocks, 64>>>(next, 12345, 64, out, cyc); CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
        printf("dependent random load (256MB table), %d blocks x 64: %.0f ns/hop\n", blocks, hc * 10.0 / 64);
        k_atomic_chain<<<blocks, 64>>>(ctr, N - 1, 64, out, cyc); CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
        printf("dependent returning atomic (random), %d blocks x 64: %.0f ns/hop\n", blocks, hc * 10.0 / 64);
    }
    for (int grid : {1, 256, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipEventRecord(e0)); for (int i = 0; i < 1000; ++i) k_empty<<<grid, 256>>>(out); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep) printf("empty kernel grid %d: %.2f us/launch\n", grid, ms);
            CHK(hipEventRecord(e0)); for (int i = 0; i < 1000; ++i) k_ticket<<<grid, 256>>>(ctr, flag); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep) printf("ticket kernel grid %d: %.2f us/launch\n", grid, ms);
            CHK(hipEventRecord(e0)); for (int i = 0; i < 1000; ++i) k_ticket_fence<<<grid, 256>>>(ctr, flag); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep) printf("ticket+threadfence kernel grid %d: %.2f us/launch\n", grid, ms);
        }
    }
    CHK(hipMemset(ctr, 0x7f, 4ull * N));
    for (uint32_t tab : {1u << 20, 1u << 24, 1u << 26}) {
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipEventRecord(e0)); k_atomic_tp<<<4096, 256>>>(ctr, tab - 1, 64); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep) printf("scattered atomicSub no-return, table %u u32: %.2f G atomics/s\n", tab, 4096.0 * 256 * 64 / ms / 1e6);
            CHK(hipEventRecord(e0)); k_atomic_tp_ret<<<4096, 256>>>(ctr, tab - 1, 64, out); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep) printf("scattered atomicSub returning, table %u u32: %.2f G atomics/s\n", tab, 4096.0 * 256 * 64 / ms / 1e6);
        }
    }
    return 0;
}
