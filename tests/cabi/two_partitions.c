/* A plain C host of libocrgpu (no Python, no OCR headers): a chain of N EDTs whose links alternate between two partitions,
 * joined by peer windows inside one process, resolved by the worker loop across partitions (ocrgpu_resolve_exchange_local),
 * and checked against the same chain on ONE partition: same levels, same HASH values (the datablocks travel by value).
 *   gcc two_partitions.c -I include -L ocr_amd -locrgpu -Wl,-rpath,$PWD/ocr_amd -o two_partitions && ./two_partitions
 * The calls are those a comm-platform + policy-domain binding makes: ocrEdtCreate / ocrAddDependence /
 * ocrEventSatisfy batches on the owner of each object, then the loop.                                                  */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ocrgpu.h"
#include "ocr_opstream.h"

#define N 64
#define CHK(call) do { int r_ = (call); if (r_ != OCRGPU_OK) { fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #call, r_, ocrgpu_last_error()); exit(1); } } while (0)

static ocrgpu_engine_t *make(uint32_t location, uint32_t n_loc) {
    ocrgpu_config_t cfg; memset(&cfg, 0, sizeof cfg);
    cfg.max_edt = N + 4; cfg.max_evt = 4; cfg.max_db = 4 * N + 16; cfg.location = location; cfg.max_slots = 2 * N; cfg.max_edges = 4 * N;
    ocrgpu_engine_t *e = NULL;
    CHK(ocrgpu_init(&e, 0, &cfg));
    CHK(ocrgpu_exchange_config(e, n_loc, NULL, 256, 4 * N));
    CHK(ocrgpu_set_hash_tag_base(e, 0));
    return e;
}

/* the chain over `n_loc` partitions: EDT i lives on partition i % n_loc; returns levels and values by EDT id */
static void run(uint32_t n_loc, int32_t *level, uint64_t *val) {
    ocrgpu_engine_t *g[2];
    uint64_t edt[N], outevt[N], start;
    for (uint32_t l = 0; l < n_loc; ++l) g[l] = make(l, n_loc);
    if (n_loc > 1) {
        uint8_t handles[2 * OCRGPU_PEER_HANDLE_BYTES];
        for (uint32_t l = 0; l < n_loc; ++l) CHK(ocrgpu_peer_export(g[l], handles + l * OCRGPU_PEER_HANDLE_BYTES));
        for (uint32_t l = 0; l < n_loc; ++l) CHK(ocrgpu_peer_connect(g[l], handles));
    }
    const uint32_t once = OCREVT_ONCE; const int32_t zero = 0;
    CHK(ocrgpu_event_create(g[0], 1, &once, &zero, &start, OCRGPU_HOST));
    for (uint32_t i = 0; i < N; ++i) {                       /* every partition creates what it owns */
        const uint32_t depc = 1, body = OCRBODY_HASH | OCRBODY_F_OUTEVT; const uint64_t id = i;
        CHK(ocrgpu_edt_create(g[i % n_loc], 1, &depc, &body, &id, &edt[i], &outevt[i], NULL, OCRGPU_HOST));
    }
    for (uint32_t i = 0; i < N; ++i) {                       /* ocrAddDependence runs on the owner of the SOURCE */
        const uint64_t src = i ? outevt[i - 1] : start; const uint32_t slot = 0, mode = i ? 8u /* DB_MODE_RO */ : 0u;
        ocrgpu_engine_t *owner = i ? g[(i - 1) % n_loc] : g[0];
        CHK(ocrgpu_add_dependence(owner, 1, &src, &edt[i], &slot, &mode, OCRGPU_HOST));
    }
    const uint64_t null_guid = OCRGPU_NULL_GUID; const uint32_t slot0 = 0;
    CHK(ocrgpu_event_satisfy_slot(g[0], 1, &start, &null_guid, &slot0, OCRGPU_HOST));
    uint64_t levels = 0;
    if (n_loc > 1) CHK(ocrgpu_resolve_exchange_local(g, n_loc, 0, &levels));
    else { ocrgpu_stats_t st; CHK(ocrgpu_resolve(g[0], 0, &st)); levels = st.levels; }
    if (levels != N) { fprintf(stderr, "%u partitions: %llu levels, expected %d\n", n_loc, (unsigned long long)levels, N); exit(1); }
    for (uint32_t l = 0; l < n_loc; ++l) {
        const uint32_t mine = (N - l + n_loc - 1) / n_loc;
        int32_t lv[N]; uint64_t v[N];
        CHK(ocrgpu_read_edts(g[l], 0, mine, lv, v, OCRGPU_HOST));
        for (uint32_t k = 0; k < mine; ++k) { level[l + k * n_loc] = lv[k]; val[l + k * n_loc] = v[k]; }
        ocrgpu_stats_t st; CHK(ocrgpu_get_stats(g[l], &st));
        for (int k = 0; k < 8; ++k) if (st.err[k]) { fprintf(stderr, "partition %u: err[%d] = %llu\n", l, k, (unsigned long long)st.err[k]); exit(1); }
    }
    for (uint32_t l = 0; l < n_loc; ++l) CHK(ocrgpu_fini(g[l]));
}

int main(void) {
    static int32_t l1[N], l2[N]; static uint64_t v1[N], v2[N];
    run(1, l1, v1);
    run(2, l2, v2);
    for (int i = 0; i < N; ++i) {
        if (l1[i] != i || l2[i] != i) { fprintf(stderr, "EDT %d fired at level %d / %d\n", i, l1[i], l2[i]); return 1; }
        if (v1[i] != v2[i]) { fprintf(stderr, "EDT %d: value %llx on one partition, %llx on two\n", i, (unsigned long long)v1[i], (unsigned long long)v2[i]); return 1; }
    }
    printf("ok: %d-link chain, levels and values agree on 1 and 2 partitions (last value %llx)\n", N, (unsigned long long)v2[N - 1]);
    return 0;
}
