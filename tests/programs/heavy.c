/*
 * Worker-thread scaling of libocrshim (OCRSHIM_WORKERS): L levels of W independent EDTs whose bodies do real work
 * (an integer recurrence of `iters` steps on their parameter), joined by a latch per level.  Plain public OCR API.
 * usage: heavy [W] [L] [iters]   prints  HEAVY W=.. L=.. iters=.. sum=..
 */
#include <stdlib.h>
#include "ocr.h"

static u64 W = 4096, L = 8, ITERS = 20000;
static u64 total;
static ocrGuid_t tplWork, tplLevel;

/* paramv[0] = seed ; paramv[1] = latch to DECR */
ocrGuid_t workEdt(u32 paramc, u64 *paramv, u32 depc, ocrEdtDep_t depv[]) {
    u64 x = paramv[0];
    for (u64 i = 0; i < ITERS; ++i) x = x * 6364136223846793005ULL + 1442695040888963407ULL;
    __sync_fetch_and_add(&total, x >> 40);
    ocrGuid_t latch = {.guid = (intptr_t)paramv[1]};
    ocrEventSatisfySlot(latch, NULL_GUID, OCR_EVENT_LATCH_DECR_SLOT);
    return NULL_GUID;
}

/* paramv[0] = level ; depv[0] = the previous level's latch (or the start) */
ocrGuid_t levelEdt(u32 paramc, u64 *paramv, u32 depc, ocrEdtDep_t depv[]) {
    u64 lvl = paramv[0];
    if (lvl == L) { PRINTF("HEAVY W=%" PRIu64 " L=%" PRIu64 " iters=%" PRIu64 " sum=%" PRIu64 "\n", W, L, ITERS, total); ocrShutdown(); return NULL_GUID; }
    ocrGuid_t latch, next;
    ocrEventParams_t p; p.EVENT_LATCH.counter = W;
    ocrEventCreateParams(&latch, OCR_EVENT_LATCH_T, EVT_PROP_NONE, &p);
    u64 np[1] = {lvl + 1};
    ocrEdtCreate(&next, tplLevel, EDT_PARAM_DEF, np, EDT_PARAM_DEF, &latch, EDT_PROP_NONE, NULL_HINT, NULL);
    for (u64 w = 0; w < W; ++w) {
        ocrGuid_t e; u64 wp[2] = {lvl * W + w, (u64)latch.guid};
        ocrEdtCreate(&e, tplWork, EDT_PARAM_DEF, wp, EDT_PARAM_DEF, NULL, EDT_PROP_NONE, NULL_HINT, NULL);
    }
    return NULL_GUID;
}

ocrGuid_t mainEdt(u32 paramc, u64 *paramv, u32 depc, ocrEdtDep_t depv[]) {
    u64 argc = getArgc(depv[0].ptr);
    if (argc > 1) W = (u64)atoll(getArgv(depv[0].ptr, 1));
    if (argc > 2) L = (u64)atoll(getArgv(depv[0].ptr, 2));
    if (argc > 3) ITERS = (u64)atoll(getArgv(depv[0].ptr, 3));
    ocrEdtTemplateCreate(&tplWork, workEdt, 2, 0);
    ocrEdtTemplateCreate(&tplLevel, levelEdt, 1, 1);
    ocrGuid_t first; u64 p0[1] = {0}; ocrGuid_t none = NULL_GUID;
    ocrEdtCreate(&first, tplLevel, EDT_PARAM_DEF, p0, EDT_PARAM_DEF, &none, EDT_PROP_NONE, NULL_HINT, NULL);
    return NULL_GUID;
}
