/*
 * gpu-workpile.c -- workpile "GPU": ocrWorkpileFcts_t (ocr/src/inc/ocr-workpile.h:39-84) over the engine's deque of guids in HBM
 * (ocrgpu_workpile_push / ocrgpu_workpile_pop).  Counterpart of ocr/src/workpile/hc/hc-workpile.c, whose deque_t
 * (ocr/src/utils/deque.c:190-265) it replaces: POP takes from the tail, STEAL from the head, PUSH appends at the tail,
 * PUSHBACK at the head.  Workpiles are driven by the LEGACY "HC" scheduler (ocr/src/scheduler/hc/hc-scheduler.c:207-334).
 *
 * Registration: #define ENABLE_WORKPILE_GPU; workpileGpu_id in workpileType_t (ocr/src/workpile/workpile-all.h); "GPU" in
 * workpile_types[] and `case workpileGpu_id: return newOcrWorkpileFactoryGpu(perType);` (ocr/src/workpile/workpile-all.c:10-42);
 * cfg: [WorkPileType0] name=GPU / [WorkpileInst0] type=GPU.
 */
#include "ocr-config.h"
#ifdef ENABLE_WORKPILE_GPU

#include "debug.h"
#include "ocr-errors.h"
#include "ocr-policy-domain.h"
#include "ocr-sysboot.h"
#include "ocr-workpile.h"
#include "ocrgpu-binding.h"

typedef struct {
    ocrWorkpile_t base;
} ocrWorkpileGpu_t;

typedef struct {
    ocrWorkpileFactory_t base;
} ocrWorkpileFactoryGpu_t;

static void gpuWorkpileDestruct(ocrWorkpile_t *base) {
    runtimeChunkFree((u64)base, PERSISTENT_CHUNK);
}

/* the deque itself lives in the engine the GUID provider made at RL_GUID_OK: nothing to build here (hc-workpile.c:44-58 makes
 * its deque at the same runlevel); the workpile gets its own guid at RL_COMPUTE_OK like every workpile */
static u8 gpuWorkpileSwitchRunlevel(ocrWorkpile_t *self, ocrPolicyDomain_t *PD, ocrRunlevel_t runlevel, phase_t phase, u32 properties,
                                    void (*callback)(ocrPolicyDomain_t *, u64), u64 val) {
    u8 toReturn = 0;
    ASSERT(callback == NULL);
    ASSERT((properties & RL_REQUEST) && !(properties & RL_RESPONSE) && !(properties & RL_RELEASE));
    ASSERT(!(properties & RL_FROM_MSG));
    switch(runlevel) {
    case RL_CONFIG_PARSE:
    case RL_NETWORK_OK:
    case RL_MEMORY_OK:
    case RL_GUID_OK:
    case RL_USER_OK:
        break;
    case RL_PD_OK:
        if(properties & RL_BRING_UP)
            self->pd = PD;
        break;
    case RL_COMPUTE_OK:
        if(properties & RL_BRING_UP) {
            if(RL_IS_FIRST_PHASE_UP(PD, RL_COMPUTE_OK, phase))
                guidify(self->pd, (u64)self, &(self->fguid), OCR_GUID_WORKPILE);
        } else if(RL_IS_LAST_PHASE_DOWN(PD, RL_COMPUTE_OK, phase)) {
            PD_MSG_STACK(msg);
            getCurrentEnv(NULL, NULL, NULL, &msg);
#define PD_MSG (&msg)
#define PD_TYPE PD_MSG_GUID_DESTROY
            msg.type = PD_MSG_GUID_DESTROY | PD_MSG_REQUEST;
            PD_MSG_FIELD_I(guid) = self->fguid;
            PD_MSG_FIELD_I(properties) = 0;
            toReturn |= self->pd->fcts.processMessage(self->pd, &msg, false);
            self->fguid.guid = NULL_GUID;
#undef PD_MSG
#undef PD_TYPE
        }
        break;
    default:
        ASSERT(0);
    }
    return toReturn;
}

/* hcWorkpilePop (hc-workpile.c:103-128): NULL_GUID when there is nothing to take */
static ocrFatGuid_t gpuWorkpilePop(ocrWorkpile_t *base, ocrWorkPopType_t type, ocrCost_t *cost) {
    ocrFatGuid_t fguid;
    u64 g = 0, n = 0;
    ASSERT(type == POP_WORKPOPTYPE || type == STEAL_WORKPOPTYPE);
    OCRGPU_ENTER();
    int rc = ocrgpu_workpile_pop(ocrgpuBinding.engine, type == POP_WORKPOPTYPE ? OCRGPU_WP_POP : OCRGPU_WP_STEAL, 1, &g, &n, OCRGPU_HOST);
    OCRGPU_LEAVE();
    ASSERT(rc == OCRGPU_OK);
    fguid.guid.guid = (rc == OCRGPU_OK && n == 1) ? g : NULL_GUID.guid;
    fguid.metaDataPtr = NULL;
    return fguid;
}

/* hcWorkpilePush (hc-workpile.c:130-150).  A full deque is an assert in the reference (deque.c:194) and one here. */
static void gpuWorkpilePush(ocrWorkpile_t *base, ocrWorkPushType_t type, ocrFatGuid_t g) {
    u64 v = g.guid.guid;
    ASSERT(type == PUSH_WORKPUSHTYPE || type == PUSH_WORKPUSHBACKTYPE);
    OCRGPU_ENTER();
    int rc = ocrgpu_workpile_push(ocrgpuBinding.engine, type == PUSH_WORKPUSHTYPE ? OCRGPU_WP_PUSH : OCRGPU_WP_PUSHBACK, 1, &v, OCRGPU_HOST);
    OCRGPU_LEAVE();
    ASSERT(rc == OCRGPU_OK);
}

static void initializeWorkpileGpu(ocrWorkpileFactory_t *factory, ocrWorkpile_t *self, ocrParamList_t *perInstance) {
    initializeWorkpileOcr(factory, self, perInstance);
}

static ocrWorkpile_t *newWorkpileGpu(ocrWorkpileFactory_t *factory, ocrParamList_t *perInstance) {
    ocrWorkpile_t *derived = (ocrWorkpile_t *)runtimeChunkAlloc(sizeof(ocrWorkpileGpu_t), PERSISTENT_CHUNK);
    factory->initialize(factory, derived, perInstance);
    return derived;
}

static void destructWorkpileFactoryGpu(ocrWorkpileFactory_t *factory) {
    runtimeChunkFree((u64)factory, NONPERSISTENT_CHUNK);
}

ocrWorkpileFactory_t *newOcrWorkpileFactoryGpu(ocrParamList_t *perType) {
    ocrWorkpileFactory_t *base = (ocrWorkpileFactory_t *)runtimeChunkAlloc(sizeof(ocrWorkpileFactoryGpu_t), NONPERSISTENT_CHUNK);
    base->instantiate = &newWorkpileGpu;
    base->initialize = &initializeWorkpileGpu;
    base->destruct = &destructWorkpileFactoryGpu;
    base->workpileFcts.destruct = FUNC_ADDR(void (*)(ocrWorkpile_t *), gpuWorkpileDestruct);
    base->workpileFcts.switchRunlevel = FUNC_ADDR(u8 (*)(ocrWorkpile_t *, ocrPolicyDomain_t *, ocrRunlevel_t, phase_t, u32,
                                                         void (*)(ocrPolicyDomain_t *, u64), u64), gpuWorkpileSwitchRunlevel);
    base->workpileFcts.pop = FUNC_ADDR(ocrFatGuid_t (*)(ocrWorkpile_t *, ocrWorkPopType_t, ocrCost_t *), gpuWorkpilePop);
    base->workpileFcts.push = FUNC_ADDR(void (*)(ocrWorkpile_t *, ocrWorkPushType_t, ocrFatGuid_t), gpuWorkpilePush);
    return base;
}

#endif /* ENABLE_WORKPILE_GPU */
