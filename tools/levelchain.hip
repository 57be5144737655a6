// levelchain -- a skeleton with config 2's exact access pattern and NO OCR semantics: measures the
// floor of two work-list structures before any engine code changes (not product code).
//
//   DAG        : L layers x W EDTs; an EDT of layer l > 0 has 1 + popcount(6 random bits) sources, uniform in layer l-1
//                (ocr_amd/opstream.py layered_dag).  One 64-B line per EDT, 8-B waiter records in CSR rows.
//   sync       : one launch per level (ready list -> line -> waiter row -> returning atomic -> append), launches queued
//                back to back: the structure of the engine's k_step without its semantics.
//   async      : ONE resident kernel, no barrier.  A wave keeps the EDTs it made ready and fires them in its next iteration
//                (continuation passing); the level is carried as data: atomicMax on a word in the destination's line issued in
//                front of the returning atomic on its pending word, the lane that takes pending to 0 reads it back.  Surplus
//                goes through sharded ticket queues; termination = double snapshot of per-shard {readied, fired} totals.
//                Bodies (HASH fold over the producers' values) are off the dependence path: a producer publishes {val, DONE}
//                as one 16-B sc1 store (granule), a consumer gathers with 16-B sc1 loads and looks again in the rare case the
//                granule is not there yet.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/levelchain.hip -o tools/_levelchain
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "ocr_opstream.h"
typedef uint32_t u32; typedef uint64_t u64; typedef int32_t s32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct alignas(64) Line { u32 pending, level; u64 param0; u64 val; u32 state, depc; u32 out_row, out_deg; u32 slot[6]; };
#define IROW 8u // waiters kept beside the line, indexed by EDT: their address is known before the line is (prefetch)
static_assert(sizeof(Line) == 64, "one sector");
#define DONE 0x10000u
#define NONE 0xFFFFFFFFu
#define NSH 8u
#define SPILL_CAP 2048u
struct alignas(128) Shard { u32 tail; u32 p0[31]; u32 head; u32 p1[31]; u64 acct; u32 p2[30]; }; // acct = entries consumed << 32 | busy waves
struct P {
    Line *line; u64 *waiters; u64 *irow; u32 *ovf; u32 *q; u32 qcap; Shard *sh; s32 *level_out; u32 n;
    u32 refill, body, sleep, prereserve, prefetch; u64 *stats;
};

__device__ __forceinline__ u32 lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u32 rank_in(u64 m) { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }
__device__ __forceinline__ u32 ld_sc1(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_sc1(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(u32 *p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st16_sc1(void *p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
// eight 16-B sc1 loads in flight, one wait (the block's wait also covers what the wave issued before it)
__device__ __forceinline__ void gather8_sc1(const void *const (&a)[8], u32x4 (&g)[8]) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\t"
                 "global_load_dwordx4 %3, %11, off sc1\n\tglobal_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
}

// ---- sync: one launch per level -------------------------------------------------------------------------------
struct SyncP { Line *line; u64 *waiters; u32 *list[2]; u32 *cnt; s32 *level_out; };
__global__ __launch_bounds__(256) void k_level(SyncP p, u32 level) {
    const u32 n = p.cnt[level & 3];
    if (blockIdx.x == 0 && threadIdx.x == 0) p.cnt[(level + 2) & 3] = 0;
    const u32 *cur = p.list[level & 1]; u32 *nxt = p.list[(level + 1) & 1];
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    u32 row = 0, deg = 0;
    if (i < n) { const u32 e = cur[i]; const Line *L = &p.line[e]; row = L->out_row; deg = L->out_deg; p.level_out[e] = (s32)level; }
    // (thread per EDT, its ~4 edges in a row: the engine balances by edge; here the rows are short)
    for (u32 k = 0; ; ++k) {
        bool ready = false; u32 dst = 0;
        if (k < deg) { dst = (u32)p.waiters[row + k]; ready = atomicSub(&p.line[dst].pending, 1u) == 1u; }
        const u64 m = __ballot(ready);
        if (!__ballot(k < deg)) break;
        if (m) {
            u32 base = 0; const int leader = __ffsll((long long)m) - 1;
            if ((int)lane_id() == leader) base = atomicAdd(&p.cnt[(level + 1) & 3], (u32)__popcll(m));
            base = __shfl(base, leader);
            if (ready) nxt[base + rank_in(m)] = dst;
        }
    }
}

// ---- async: one resident kernel ------------------------------------------------------------------------------
// Queue entry = {EDT + 1, level} (8 B, one sc1 store; 0 = not written yet, SKIP = reserved but not needed).  Per shard: tail
// (entries reserved), head (tickets handed out), acct = entries consumed << 32 | busy waves.  Termination: no wave busy and every
// reserved entry consumed, seen twice in a row (all three are monotone except busy, which only rises together with consumed).
#define SKIP 0xFFFFFFFFu
__device__ bool quiescent(const P &p) {
    const u32 lane = lane_id();
    u64 a = 0, b = 0; u32 ta = 0, tb = 0;
    if (lane < NSH) { a = ld_sc1(&p.sh[lane].acct); ta = ld_sc1(&p.sh[lane].tail); }
    if (lane < NSH) { b = ld_sc1(&p.sh[lane].acct); tb = ld_sc1(&p.sh[lane].tail); }
    if (__ballot(a != b || ta != tb)) return false;
    u64 cons = a >> 32, busy = a & 0xFFFFFFFFull, tl = ta;
    for (int o = 1; o < 8; o <<= 1) { cons += __shfl_xor(cons, o); busy += __shfl_xor(busy, o); tl += __shfl_xor(tl, o); }
    cons = __shfl(cons, 0); busy = __shfl(busy, 0); tl = __shfl(tl, 0);
    return (busy & 0xFFFFFFFFull) == 0 && cons == tl;
}
#define OVF_MARK 0xF0000000u // irow[e][7].hi when the row is longer than 8: lo = index in waiters[] of the 8th waiter, hi = mark | waiters left
#define MAXE 640u            // edges one iteration expands in LDS (64 items x 8 inline waiters + what fits of the long rows)

template <bool BODY>
__global__ __launch_bounds__(64) void k_async(P p) {
    __shared__ u64 lds_w[MAXE];
    __shared__ u32 lds_l[MAXE];
    __shared__ u32 lds_out[MAXE];
    const u32 lane = threadIdx.x, wid = blockIdx.x, sh = wid & (NSH - 1);
    Shard *S = &p.sh[sh];
    u64 *Q = (u64 *)p.q + (size_t)sh * p.qcap;
    u32 item = NONE, lvl = 0, ticket = NONE, spins = 0, iter = 0, sink = 0;
    bool busy = false;
    u64 n_iter = 0, n_fired = 0, n_spill = 0, n_respin = 0, n_idle = 0, t_busy = 0, t_spill = 0, n_spills = 0, n_skip = 0, n_extra = 0;
    for (;;) {
        bool got = false;
        if (ticket != NONE) { const u64 v = ld_sc1(&Q[ticket]); if ((u32)v) { if ((u32)v != SKIP) { item = (u32)v - 1; lvl = (u32)(v >> 32); } ticket = NONE; got = true; } }
        const u64 gm = __ballot(got);
        const u64 hm = __ballot(item != NONE);
        if (gm) { // entries consumed (+ idle -> busy) in ONE atomic
            const bool nb = hm != 0;
            if (lane == 0) atomicAdd((unsigned long long *)&S->acct, ((unsigned long long)__popcll(gm) << 32) + ((nb && !busy) ? 1ull : 0ull));
            busy = busy || nb;
        }
        if (!hm) {
            if (busy) { if (lane == 0) atomicAdd((unsigned long long *)&S->acct, ~0ull); busy = false; }
            const u64 fm = __ballot(lane < p.refill && ticket == NONE);
            if ((u32)__popcll(fm) * 2 >= p.refill) { // at least half of the ticket lanes are free: take tickets for them
                u32 base = 0;
                if (lane == 0) base = atomicAdd(&S->head, (u32)__popcll(fm));
                base = __shfl(base, 0);
                if (lane < p.refill && ticket == NONE && base + rank_in(fm) < p.qcap) ticket = base + rank_in(fm);
                if (base >= p.qcap) { if (quiescent(p)) break; __builtin_amdgcn_s_sleep(64); }
                continue;
            }
            ++n_idle; ++spins;
            if ((spins & 15) == 0 && quiescent(p)) break;
            if (p.sleep) { if (spins < 8) __builtin_amdgcn_s_sleep(2); else if (spins < 32) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(32); }
            continue;
        }
        const u64 t_in = __builtin_amdgcn_s_memrealtime();
        spins = 0; ++n_iter; ++iter;
        const bool has = item != NONE;
        n_fired += __popcll(hm);
        n_skip += __popcll(gm) - __popcll(hm & gm);
        // ---- one trip: the inline waiter row (its address needs the EDT's index only) + the line for the body ----
        u64 rw[IROW]; u32 depc = 0; u64 param0 = 0; u32 sl[6] = {0, 0, 0, 0, 0, 0};
        if (has) {
            const ulonglong2 *R = (const ulonglong2 *)(p.irow + (size_t)item * IROW);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const ulonglong2 x = R[k]; rw[2 * k] = x.x; rw[2 * k + 1] = x.y; }
            if (BODY) { const Line *L = &p.line[item]; depc = L->depc; param0 = L->param0; for (int k = 0; k < 6; ++k) sl[k] = L->slot[k]; }
            p.level_out[item] = (s32)lvl;
        } else { for (int k = 0; k < 8; ++k) rw[k] = 0; }
        u32 cnt = 0, more = 0, more_row = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) cnt += rw[k] != 0;
        if (has && ((u32)(rw[7] >> 32) & OVF_MARK) == OVF_MARK) { more = (u32)(rw[7] >> 32) & ~OVF_MARK; more_row = (u32)rw[7]; cnt = 7; }
        const u32 deg = cnt + more;
        u32 incl = deg;
        for (u32 o = 1; o < p.refill; o <<= 1) { const u32 nb = __shfl_up(incl, o); if (lane >= o) incl += nb; } // (items sit in the ticket lanes)
        const u32 total = __shfl(incl, p.refill - 1);
        const u32 excl = incl - deg;
        // reserve the output entries now (expected ready EDTs + slack); the index is needed only after the atomics have returned
        const u32 tsh = (sh + iter) & (NSH - 1);
        u64 *QT = (u64 *)p.q + (size_t)tsh * p.qcap;
        const u32 rsv = p.prereserve == 99u ? 0u : total / 4 + p.prereserve;
        u32 qb = 0;
        if (rsv && lane == 0) qb = atomicAdd(&p.sh[tsh].tail, rsv);
        // expand the rows into LDS: edge position -> waiter, source level
#pragma unroll
        for (int k = 0; k < 8; ++k) if ((u32)k < cnt && excl + k < MAXE) { lds_w[excl + k] = rw[k]; lds_l[excl + k] = lvl; }
        for (u32 k = 0; k < more; ++k) if (excl + cnt + k < MAXE) { lds_w[excl + cnt + k] = p.waiters[more_row + k]; lds_l[excl + cnt + k] = lvl; } // (long rows: rare, one more trip)
        __syncthreads();
        if (total > MAXE) { if (lane == 0) atomicAdd((unsigned long long *)&p.stats[14], 1ull); }
        const u32 tot = total < MAXE ? total : MAXE;
        u32 nready = 0;
        for (u32 base = 0; base < tot; base += 128) {
            u32 dst[2], old[2], nlv[2]; bool act[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u32 t = base + 64 * q + lane;
                act[q] = t < tot; old[q] = 0; nlv[q] = 0; dst[q] = 0;
                if (act[q]) {
                    dst[q] = (u32)lds_w[t];
                    Line *D = &p.line[dst[q]];
                    atomicMax(&D->level, lds_l[t] + 1u);
                    const u64 o = atomicAdd((unsigned long long *)D, ~0ull);
                    old[q] = (u32)o; nlv[q] = (u32)(o >> 32);
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool ready = act[q] && old[q] == 1u;
                const u64 rm = __ballot(ready);
                const u32 pos = nready + rank_in(rm);
                if (ready && p.prefetch) { sink ^= (u32)p.irow[(size_t)dst[q] * IROW]; if (BODY) sink ^= p.line[dst[q]].out_deg; } // on their way into the memory-side cache for whoever takes the entry
                if (ready && pos < MAXE) { lds_out[pos] = dst[q]; lds_l[pos] = nlv[q]; } // (lds_l[pos]: pos <= t, and every lane has read its lds_l[t] above)
                nready += (u32)__popcll(rm);
            }
            __syncthreads();
        }
        // ---- hand the ready EDTs on ----
        {
            const u64 t_s = __builtin_amdgcn_s_memrealtime();
            qb = __shfl(qb, 0);
            u32 extra = 0, qx = 0;
            if (nready > rsv) { extra = nready - rsv; if (lane == 0) qx = atomicAdd(&p.sh[tsh].tail, extra); qx = __shfl(qx, 0); ++n_extra; }
            for (u32 i = lane; i < (nready > rsv ? nready : rsv); i += 64) {
                const u32 at = i < rsv ? qb + i : qx + (i - rsv);
                const u64 v = i < nready ? ((u64)lds_l[i] << 32) | (u64)(lds_out[i] + 1u) : (u64)SKIP;
                if (at < p.qcap) __hip_atomic_store(&QT[at], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            n_spill += nready; ++n_spills;
            t_spill += __builtin_amdgcn_s_memrealtime() - t_s;
        }
        __syncthreads();
        t_busy += __builtin_amdgcn_s_memrealtime() - t_in;
    if (BODY) { // the body, behind the hand-over: nothing on the dependence path waits for its value (consumers look at the granule's DONE bit)
            const void *a[8]; u32x4 g[8];
            const u32 *ov = depc > 6 ? p.ovf + sl[0] : nullptr;
            u32 pw[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                u32 x = 0;
                if (has && (u32)k < depc) x = depc > 6 ? ov[k] : (k < 6 ? sl[k] : 0u);
                pw[k] = x;
                a[k] = x ? (const void *)&p.line[x - 1].val : (const void *)&p.line[0].val; // (unused lanes: one address, one request)
            }
            for (;;) {
                gather8_sc1(a, g);
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (pw[k] && !(g[k].z & DONE)) ok = false;
                if (!__ballot(!ok)) break;
                ++n_respin;
            }
            if (has) {
                u64 v = ocrops_hash_init(param0);
#pragma unroll
                for (int k = 0; k < 8; ++k) if ((u32)k < depc) { const u64 pv = pw[k] ? ((u64)g[k].y << 32 | g[k].x) : 0ull; v = ocrops_hash_step(v, k, pv, pw[k]); }
                u32x4 out; out.x = (u32)v; out.y = (u32)(v >> 32); out.z = DONE | 1u; out.w = depc;
                st16_sc1((void *)&p.line[item].val, out);
            }
        }
        item = NONE;

    }
    if (lane == 0 && p.stats) {
        atomicAdd((unsigned long long *)&p.stats[0], (unsigned long long)n_iter); atomicAdd((unsigned long long *)&p.stats[1], (unsigned long long)n_fired);
        atomicAdd((unsigned long long *)&p.stats[2], (unsigned long long)n_spill); atomicAdd((unsigned long long *)&p.stats[3], (unsigned long long)n_respin);
        atomicAdd((unsigned long long *)&p.stats[4], (unsigned long long)n_idle); if (n_iter) atomicAdd((unsigned long long *)&p.stats[5], 1ull);
        atomicAdd((unsigned long long *)&p.stats[6], (unsigned long long)t_busy); atomicAdd((unsigned long long *)&p.stats[7], (unsigned long long)t_spill);
        atomicAdd((unsigned long long *)&p.stats[8], (unsigned long long)n_spills); atomicAdd((unsigned long long *)&p.stats[9], (unsigned long long)n_skip);
        atomicAdd((unsigned long long *)&p.stats[10], (unsigned long long)n_extra); if (sink == 0x12345678u) p.stats[15] = sink;
    }
}

static u64 splitmix(u64 seed, u64 i) {
    u64 z = seed + i * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}

int main(int argc, char **argv) {
    u32 Lyr = argc > 1 ? atoi(argv[1]) : 1000, W = argc > 2 ? atoi(argv[2]) : 10000;
    const u32 reps = argc > 3 ? atoi(argv[3]) : 5;
    const u32 N = Lyr * W;
    printf("levelchain: %u layers x %u = %u EDTs\n", Lyr, W, N);
    std::vector<u32> depc(N, 1);
    for (u32 e = W; e < N; ++e) depc[e] = 1 + __builtin_popcountll(splitmix(0x0C5200000002ull, e - W + 1) & 0x3F);
    std::vector<u64> sbase(N + 1, 0);
    for (u32 e = 0; e < N; ++e) sbase[e + 1] = sbase[e] + (e < W ? 0 : depc[e]);
    const u64 E = sbase[N];
    std::vector<u32> src(E);
    std::vector<u32> outdeg(N, 0);
    for (u32 e = W; e < N; ++e)
        for (u32 k = 0; k < depc[e]; ++k) {
            const u64 i = sbase[e] + k;
            const u32 s = (e / W - 1) * W + (u32)(splitmix(0x0C5200000002ull ^ 0xA5A5A5A5ull, i + 1) % W);
            src[i] = s; ++outdeg[s];
        }
    std::vector<u32> row(N + 1, 0);
    for (u32 e = 0; e < N; ++e) row[e + 1] = row[e] + outdeg[e];
    std::vector<u64> waiters(E ? E : 1);
    std::vector<u32> cur(row.begin(), row.end() - 1);
    std::vector<Line> line(N);
    std::vector<u32> ovf;
    u32 maxdeg = 0;
    for (u32 e = 0; e < N; ++e) {
        Line &l = line[e]; memset(&l, 0, sizeof l);
        l.pending = e < W ? 0 : depc[e]; l.level = 0; l.depc = e < W ? 0 : depc[e]; l.param0 = e; l.state = 1;
        l.out_row = row[e]; l.out_deg = outdeg[e]; if (outdeg[e] > maxdeg) maxdeg = outdeg[e];
        if (l.depc > 6) { l.slot[0] = (u32)ovf.size(); for (u32 k = 0; k < l.depc; ++k) ovf.push_back(0); }
    }
    for (u32 e = W; e < N; ++e)
        for (u32 k = 0; k < depc[e]; ++k) {
            const u32 s = src[sbase[e] + k];
            waiters[cur[s]++] = (u64)e | ((u64)k << 32);
            if (depc[e] > 6) ovf[line[e].slot[0] + k] = s + 1; else line[e].slot[k] = s + 1;
        }
    std::vector<u64> irow((size_t)N * IROW, 0);
    for (u32 e = 0; e < N; ++e) {
        for (u32 k = 0; k < outdeg[e] && k < IROW; ++k) irow[(size_t)e * IROW + k] = waiters[row[e] + k];
        if (outdeg[e] > IROW) irow[(size_t)e * IROW + 7] = (u64)(row[e] + 7) | ((u64)(OVF_MARK | (outdeg[e] - 7)) << 32);
    }
    printf("edges %llu, max out-degree %u, overflow slot words %zu\n", (unsigned long long)E, maxdeg, ovf.size());
    // CPU values (HASH fold), level = layer
    std::vector<u64> val(N);
    for (u32 e = 0; e < N; ++e) {
        u64 v = ocrops_hash_init(e);
        for (u32 k = 0; k < line[e].depc; ++k) { const u32 pw = line[e].depc > 6 ? ovf[line[e].slot[0] + k] : line[e].slot[k]; v = ocrops_hash_step(v, k, val[pw - 1], pw); }
        val[e] = v;
    }
    Line *d_line, *d_line0; u64 *d_wait, *d_irow; u32 *d_ovf, *d_q; Shard *d_sh; s32 *d_lvl; u64 *d_stats;
    const u32 qcap = N + 1024;
    CHK(hipMalloc(&d_line, sizeof(Line) * (size_t)N)); CHK(hipMalloc(&d_line0, sizeof(Line) * (size_t)N));
    CHK(hipMalloc(&d_wait, 8 * waiters.size())); CHK(hipMalloc(&d_ovf, 4 * (ovf.size() + 1))); CHK(hipMalloc(&d_q, 8ull * NSH * qcap));
    CHK(hipMalloc(&d_sh, sizeof(Shard) * NSH)); CHK(hipMalloc(&d_lvl, 4ull * N)); CHK(hipMalloc(&d_stats, 128));
    CHK(hipMemcpy(d_line0, line.data(), sizeof(Line) * (size_t)N, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_wait, waiters.data(), 8 * waiters.size(), hipMemcpyHostToDevice));
    CHK(hipMalloc(&d_irow, 8 * irow.size())); CHK(hipMemcpy(d_irow, irow.data(), 8 * irow.size(), hipMemcpyHostToDevice));
    if (ovf.size()) CHK(hipMemcpy(d_ovf, ovf.data(), 4 * ovf.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    std::vector<s32> hl(N); std::vector<Line> hline(N);
    auto check = [&](const char *what, bool vals) {
        CHK(hipMemcpy(hl.data(), d_lvl, 4ull * N, hipMemcpyDeviceToHost));
        u64 badl = 0, badv = 0;
        for (u32 e = 0; e < N; ++e) badl += hl[e] != (s32)(e / W);
        if (vals) { CHK(hipMemcpy(hline.data(), d_line, sizeof(Line) * (size_t)N, hipMemcpyDeviceToHost)); for (u32 e = 0; e < N; ++e) badv += hline[e].val != val[e]; }
        printf("  check %s: %llu wrong levels, %llu wrong values%s\n", what, (unsigned long long)badl, (unsigned long long)badv, vals ? "" : " (not computed)");
    };
    // ---- sync ----
    {
        u32 *d_list[2], *d_cnt; CHK(hipMalloc(&d_list[0], 4ull * W * 2)); CHK(hipMalloc(&d_list[1], 4ull * W * 2)); CHK(hipMalloc(&d_cnt, 16));
        std::vector<u32> l0(W); for (u32 i = 0; i < W; ++i) l0[i] = i;
        SyncP sp{d_line, d_wait, {d_list[0], d_list[1]}, d_cnt, d_lvl};
        float best = 1e9f;
        for (u32 r = 0; r < reps; ++r) {
            CHK(hipMemcpy(d_line, d_line0, sizeof(Line) * (size_t)N, hipMemcpyDeviceToDevice)); CHK(hipMemset(d_lvl, 0xFF, 4ull * N));
            CHK(hipMemcpy(d_list[0], l0.data(), 4ull * W, hipMemcpyHostToDevice)); u32 c4[4] = {W, 0, 0, 0}; CHK(hipMemcpy(d_cnt, c4, 16, hipMemcpyHostToDevice));
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            for (u32 l = 0; l < Lyr; ++l) k_level<<<(W + 255) / 256, 256>>>(sp, l);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("sync  (launch per level, thread per EDT): %.3f ms per resolve, %.2f us per level, %.2fe9 sat/s\n", best, best * 1e3 / Lyr, E / best / 1e6);
        check("sync", false);
    }
    // ---- async ----
    struct Cfg { u32 waves, refill, body, sleep, prereserve, prefetch; };
    std::vector<Cfg> cfgs;
    if (argc > 4) { Cfg c{(u32)atoi(argv[4]), argc > 5 ? (u32)atoi(argv[5]) : 16u, argc > 6 ? (u32)atoi(argv[6]) : 1u, argc > 7 ? (u32)atoi(argv[7]) : 1u, argc > 8 ? (u32)atoi(argv[8]) : 99u, argc > 9 ? (u32)atoi(argv[9]) : 0u}; cfgs.push_back(c); }
    else {
        for (u32 body : {0u, 1u})
            for (u32 waves : {512u, 1024u, 1536u})
                for (u32 pr : {99u, 0u, 1u})
                    for (u32 pf : {0u, 1u}) cfgs.push_back(Cfg{waves, 16, body, 1, pr, pf});
        cfgs.push_back(Cfg{2048, 16, 1, 1, 99, 1});
        cfgs.push_back(Cfg{1024, 32, 1, 1, 99, 1});
        cfgs.push_back(Cfg{1024, 8, 1, 1, 99, 1});
    }
    std::vector<u64> hq((size_t)NSH * qcap, 0);
    for (const Cfg &c : cfgs) {
        float best = 1e9f; u64 st[16];
        for (u32 r = 0; r < reps; ++r) {
            CHK(hipMemcpy(d_line, d_line0, sizeof(Line) * (size_t)N, hipMemcpyDeviceToDevice)); CHK(hipMemset(d_lvl, 0xFF, 4ull * N));
            CHK(hipMemset(d_q, 0, 8ull * NSH * qcap)); CHK(hipMemset(d_stats, 0, 128));
            Shard hs[NSH]; memset(hs, 0, sizeof hs);
            for (u32 i = 0; i < W; ++i) { const u32 s = (i / c.refill) % NSH; hq[(size_t)s * qcap + hs[s].tail++] = i + 1; }
            for (u32 s = 0; s < NSH; ++s) CHK(hipMemcpy((u64 *)d_q + (size_t)s * qcap, hq.data() + (size_t)s * qcap, 8ull * hs[s].tail, hipMemcpyHostToDevice));
            CHK(hipMemcpy(d_sh, hs, sizeof hs, hipMemcpyHostToDevice));
            P p{d_line, d_wait, d_irow, d_ovf, d_q, qcap, d_sh, d_lvl, N, c.refill, c.body, c.sleep, c.prereserve, c.prefetch, d_stats};
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            if (c.body) k_async<true><<<c.waves, 64>>>(p); else k_async<false><<<c.waves, 64>>>(p);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            CHK(hipMemcpy(st, d_stats, 128, hipMemcpyDeviceToHost));
        }
        printf("async waves %4u refill %2u body %u sleep %u prereserve %2u prefetch %u: %.3f ms per resolve, %.2f us per level, %.2fe9 sat/s | iters %llu (%.2f us each, %.1f EDTs) hand-overs %.2f us each, %llu needed more entries, skips consumed %llu, respins %llu idle-polls %llu waves-that-worked %llu, too-long %llu\n",
               c.waves, c.refill, c.body, c.sleep, c.prereserve, c.prefetch, best, best * 1e3 / Lyr, E / best / 1e6,
               (unsigned long long)st[0], st[0] ? st[6] * 0.01 / st[0] : 0.0, st[0] ? (double)st[1] / st[0] : 0.0, st[8] ? st[7] * 0.01 / st[8] : 0.0, (unsigned long long)st[10], (unsigned long long)st[9],
               (unsigned long long)st[3], (unsigned long long)st[4], (unsigned long long)st[5], (unsigned long long)st[14]);
        check("async", c.body != 0);
        fflush(stdout);
    }
    return 0;
}
