/*
 * fib(n) as an EDT tree -- BASELINE config 1: fib(n) spawns fib(n-1), fib(n-2) and a 2-dep
 * `sum` continuation; results travel in 8-byte datablocks through ONCE events; the root runs in a finish scope.
 * Plain public OCR API only, so the same file builds against the reference runtime (oracle/Makefile.ref) and
 * against include/ocr_shim/ocr.h + libocrshim (the MI355X engine).  For n = 20: 21891 fib EDTs, 10945 sum EDTs,
 * result 6765.
 */
#include <stdlib.h>
#include "ocr.h"

static u64 n_fib, n_sum; /* EDT counters; bumped with atomics because reference workers run bodies concurrently */

static ocrGuid_t tplFib, tplSum;

/* paramv[0] = event to satisfy with the result DB ; depv[0], depv[1] = the two partial results */
ocrGuid_t sumEdt(u32 paramc, u64 *paramv, u32 depc, ocrEdtDep_t depv[]) {
    __sync_fetch_and_add(&n_sum, 1);
    ocrGuid_t out = {.guid = (intptr_t)paramv[0]};
    ocrGuid_t db; u64 *p;
    ocrDbCreate(&db, (void **)&p, sizeof(u64), DB_PROP_NONE, NULL_HINT, NO_ALLOC);
    *p = *(u64 *)depv[0].ptr + *(u64 *)depv[1].ptr;
    ocrDbRelease(db);
    ocrDbDestroy(depv[0].guid);
    ocrDbDestroy(depv[1].guid);
    ocrEventSatisfy(out, db);
    return NULL_GUID;
}

/* paramv[0] = n ; paramv[1] = event to satisfy with the DB holding fib(n) */
ocrGuid_t fibEdt(u32 paramc, u64 *paramv, u32 depc, ocrEdtDep_t depv[]) {
    __sync_fetch_and_add(&n_fib, 1);
    u64 n = paramv[0];
    ocrGuid_t out = {.guid = (intptr_t)paramv[1]};
    if (n < 2) {
        ocrGuid_t db; u64 *p;
        ocrDbCreate(&db, (void **)&p, sizeof(u64), DB_PROP_NONE, NULL_HINT, NO_ALLOC);
        *p = n;
        ocrDbRelease(db);
        ocrEventSatisfy(out, db);
        return NULL_GUID;
    }
    ocrGuid_t e1, e2, sum, f1, f2;
    ocrEventCreate(&e1, OCR_EVENT_ONCE_T, EVT_PROP_TAKES_ARG);
    ocrEventCreate(&e2, OCR_EVENT_ONCE_T, EVT_PROP_TAKES_ARG);
    u64 sp[1] = {(u64)out.guid};
    ocrEdtCreate(&sum, tplSum, EDT_PARAM_DEF, sp, EDT_PARAM_DEF, NULL, EDT_PROP_NONE, NULL_HINT, NULL);
    ocrAddDependence(e1, sum, 0, DB_MODE_RO);
    ocrAddDependence(e2, sum, 1, DB_MODE_RO);
    u64 p1[2] = {n - 1, (u64)e1.guid}, p2[2] = {n - 2, (u64)e2.guid};
    ocrEdtCreate(&f1, tplFib, EDT_PARAM_DEF, p1, EDT_PARAM_DEF, NULL, EDT_PROP_NONE, NULL_HINT, NULL);
    ocrEdtCreate(&f2, tplFib, EDT_PARAM_DEF, p2, EDT_PARAM_DEF, NULL, EDT_PROP_NONE, NULL_HINT, NULL);
    return NULL_GUID;
}

/* finish-scope root: everything spawned below it is done when its output event fires */
ocrGuid_t rootEdt(u32 paramc, u64 *paramv, u32 depc, ocrEdtDep_t depv[]) {
    ocrGuid_t f;
    ocrEdtCreate(&f, tplFib, EDT_PARAM_DEF, paramv, EDT_PARAM_DEF, NULL, EDT_PROP_NONE, NULL_HINT, NULL);
    return NULL_GUID;
}

/* depv[0] = result DB (via the STICKY result event) ; depv[1] = the finish scope's output event */
ocrGuid_t doneEdt(u32 paramc, u64 *paramv, u32 depc, ocrEdtDep_t depv[]) {
    u64 r = *(u64 *)depv[0].ptr;
    PRINTF("FIB n=%" PRIu64 " result=%" PRIu64 " fib_edts=%" PRIu64 " sum_edts=%" PRIu64 "\n", paramv[0], r, n_fib, n_sum);
    ocrDbDestroy(depv[0].guid);
    ocrShutdown();
    return NULL_GUID;
}

ocrGuid_t mainEdt(u32 paramc, u64 *paramv, u32 depc, ocrEdtDep_t depv[]) {
    u64 n = 20;
    if (getArgc(depv[0].ptr) > 1) n = (u64)atoi(getArgv(depv[0].ptr, 1));
    ocrGuid_t tplRoot, tplDone, result, root, rootOut, done;
    ocrEdtTemplateCreate(&tplFib, fibEdt, 2, 0);
    ocrEdtTemplateCreate(&tplSum, sumEdt, 1, 2);
    ocrEdtTemplateCreate(&tplRoot, rootEdt, 2, 1);
    ocrEdtTemplateCreate(&tplDone, doneEdt, 1, 2);
    ocrEventCreate(&result, OCR_EVENT_STICKY_T, EVT_PROP_TAKES_ARG);
    u64 dp[1] = {n}, rp[2] = {n, (u64)result.guid};
    ocrEdtCreate(&done, tplDone, EDT_PARAM_DEF, dp, EDT_PARAM_DEF, NULL, EDT_PROP_NONE, NULL_HINT, NULL);
    ocrEdtCreate(&root, tplRoot, EDT_PARAM_DEF, rp, EDT_PARAM_DEF, NULL, EDT_PROP_FINISH, NULL_HINT, &rootOut);
    ocrAddDependence(result, done, 0, DB_MODE_RO);
    ocrAddDependence(rootOut, done, 1, DB_MODE_NULL);
    ocrAddDependence(NULL_GUID, root, 0, DB_MODE_NULL); /* start the root only once `done` is wired up */
    return NULL_GUID;
}
