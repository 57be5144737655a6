/*
 * gpu-scheduler-heuristic.c -- scheduler heuristic "GPU" under the COMMON scheduler: ocrSchedulerHeuristicFcts_t
 * (ocr/src/inc/ocr-scheduler-heuristic.h:126-200) with the engine's work-list in place of the per-worker deques of the WST root
 * scheduler object.  Counterpart of ocr/src/scheduler-heuristic/hc/hc-scheduler-heuristic.c:
 *   NOTIFY(EDT_READY)     -> the ready EDT's guid goes onto the engine's work-list      (hc: deque push, :273-295)
 *   NOTIFY(EDT_SATISFIED) -> OCR_ENOP: the task acquires its datablocks itself           (hc: :322-326)
 *   NOTIFY(EDT_DONE)      -> PD_MSG_WORK_DESTROY                                         (hc: :303-319)
 *   GET_WORK              -> take one guid: the caller's own end first, no steal loop (there is ONE list, so a worker never spins
 *                            over empty victims as hcSchedulerHeuristicGetEdt does, :176-185); NULL_GUID = nothing to do
 * When the event / task factories move onto the engine too (ocrgpu_add_dependence / ocrgpu_event_satisfy_slot), EDT_READY
 * notifications disappear: the engine's ready log is the work-list and GET_WORK becomes ocrgpu_pop_ready (see libocrshim).
 *
 * Registration: #define ENABLE_SCHEDULER_HEURISTIC_GPU; schedulerHeuristicGpu_id in schedulerHeuristicType_t; "GPU" in
 * schedulerHeuristic_types[] and a case in newSchedulerHeuristicFactory (ocr/src/scheduler-heuristic/scheduler-heuristic-all.c:12-36);
 * cfg: [SchedulerHeuristicType0] name=GPU / [SchedulerHeuristicInst0] type=GPU.
 */
#include "ocr-config.h"
#ifdef ENABLE_SCHEDULER_HEURISTIC_GPU

#include "debug.h"
#include "ocr-errors.h"
#include "ocr-policy-domain.h"
#include "ocr-runtime-types.h"
#include "ocr-sysboot.h"
#include "ocr-workpile.h"
#include "ocr-scheduler-heuristic.h"
#include "ocr-scheduler.h"
#include "ocr-worker.h"
#include "ocrgpu-binding.h"

typedef struct {
    ocrSchedulerHeuristic_t base;
} ocrSchedulerHeuristicGpu_t;

typedef struct {
    ocrSchedulerHeuristicFactory_t base;
} ocrSchedulerHeuristicFactoryGpu_t;

/* one context per worker, as the shared-memory HC heuristic keeps (hc-scheduler-heuristic.c:76-110); they carry nothing but
 * their id here: the work-list is not per worker */
static u8 gpuSchedulerHeuristicSwitchRunlevel(ocrSchedulerHeuristic_t *self, ocrPolicyDomain_t *PD, ocrRunlevel_t runlevel, phase_t phase,
                                              u32 properties, void (*callback)(ocrPolicyDomain_t *, u64), u64 val) {
    ASSERT(callback == NULL);
    ASSERT((properties & RL_REQUEST) && !(properties & RL_RESPONSE) && !(properties & RL_RELEASE));
    ASSERT(!(properties & RL_FROM_MSG));
    switch(runlevel) {
    case RL_CONFIG_PARSE:
    case RL_NETWORK_OK:
    case RL_GUID_OK:
    case RL_COMPUTE_OK:
    case RL_USER_OK:
        break;
    case RL_PD_OK:
        ASSERT(self->scheduler);
        self->contextCount = PD->workerCount;
        ASSERT(self->contextCount > 0);
        break;
    case RL_MEMORY_OK:
        if((properties & RL_BRING_UP) && RL_IS_FIRST_PHASE_UP(PD, RL_MEMORY_OK, phase)) {
            u64 i;
            self->contexts = (ocrSchedulerHeuristicContext_t **)PD->fcts.pdMalloc(PD, self->contextCount * sizeof(ocrSchedulerHeuristicContext_t *));
            ocrSchedulerHeuristicContext_t *all = (ocrSchedulerHeuristicContext_t *)PD->fcts.pdMalloc(PD, self->contextCount * sizeof(ocrSchedulerHeuristicContext_t));
            for(i = 0; i < self->contextCount; ++i) {
                self->contexts[i] = &all[i];
                all[i].id = i;
                all[i].location = PD->myLocation;
                all[i].actionSet = NULL;
                all[i].cost = NULL;
                all[i].properties = 0;
            }
        }
        if((properties & RL_TEAR_DOWN) && RL_IS_LAST_PHASE_DOWN(PD, RL_MEMORY_OK, phase)) {
            PD->fcts.pdFree(PD, self->contexts[0]);
            PD->fcts.pdFree(PD, self->contexts);
        }
        break;
    default:
        ASSERT(0);
    }
    return 0;
}

static void gpuSchedulerHeuristicDestruct(ocrSchedulerHeuristic_t *self) {
    runtimeChunkFree((u64)self, PERSISTENT_CHUNK);
}

static u8 gpuSchedulerHeuristicUpdate(ocrSchedulerHeuristic_t *self, u32 properties) {
    return OCR_ENOTSUP;
}

static ocrSchedulerHeuristicContext_t *gpuSchedulerHeuristicGetContext(ocrSchedulerHeuristic_t *self, ocrLocation_t loc) {
    ocrWorker_t *worker = NULL;
    getCurrentEnv(NULL, &worker, NULL, NULL);
    if(worker == NULL) return NULL;
    return self->contexts[worker->id];
}

/* SCHED_GET_WORK (common-scheduler.c:234-245 <- hc-worker.c:82-93) */
static u8 gpuSchedulerHeuristicGetWorkInvoke(ocrSchedulerHeuristic_t *self, ocrSchedulerOpArgs_t *opArgs, ocrRuntimeHint_t *hints) {
    ocrSchedulerOpWorkArgs_t *taskArgs = (ocrSchedulerOpWorkArgs_t *)opArgs;
    if(taskArgs->kind != OCR_SCHED_WORK_EDT_USER) return OCR_ENOTSUP;
    u64 g = 0, n = 0;
    OCRGPU_ENTER();
    int rc = ocrgpu_workpile_pop(ocrgpuBinding.engine, OCRGPU_WP_POP, 1, &g, &n, OCRGPU_HOST);
    OCRGPU_LEAVE();
    if(rc != OCRGPU_OK) return ocrgpuToOcrError(rc);
    if(n == 1) OCRGPU_COUNT(nPopped);
    if(n == 1) { /* the worker dereferences the task at once (hc-worker.c:97-141): resolve its metadata here, as localDeguidify does */
        ocrPolicyDomain_t *pd = self->scheduler->pd;
        ocrFatGuid_t edt;
        u64 md = 0;
        edt.guid.guid = g;
        RESULT_PROPAGATE(pd->guidProviders[0]->fcts.getVal(pd->guidProviders[0], edt.guid, &md, NULL, MD_LOCAL, NULL));
        edt.metaDataPtr = (void *)md;
        taskArgs->OCR_SCHED_ARG_FIELD(OCR_SCHED_WORK_EDT_USER).edt = edt;
    }
    return 0;
}

/* SCHED_NOTIFY (common-scheduler.c:247-313 <- hc-task.c:623-652, hc-worker.c:209-216) */
static u8 gpuSchedulerHeuristicNotifyInvoke(ocrSchedulerHeuristic_t *self, ocrSchedulerOpArgs_t *opArgs, ocrRuntimeHint_t *hints) {
    ocrSchedulerOpNotifyArgs_t *notifyArgs = (ocrSchedulerOpNotifyArgs_t *)opArgs;
    switch(notifyArgs->kind) {
    case OCR_SCHED_NOTIFY_EDT_READY: {
        u64 g = notifyArgs->OCR_SCHED_ARG_FIELD(OCR_SCHED_NOTIFY_EDT_READY).guid.guid.guid;
        OCRGPU_ENTER();
        int rc = ocrgpu_workpile_push(ocrgpuBinding.engine, OCRGPU_WP_PUSH, 1, &g, OCRGPU_HOST);
        OCRGPU_LEAVE();
        OCRGPU_COUNT(nPushed);
        return ocrgpuToOcrError(rc);
    }
    case OCR_SCHED_NOTIFY_EDT_DONE: {
        ocrPolicyDomain_t *pd;
        PD_MSG_STACK(msg);
        getCurrentEnv(&pd, NULL, NULL, &msg);
#define PD_MSG (&msg)
#define PD_TYPE PD_MSG_WORK_DESTROY
        msg.type = PD_MSG_WORK_DESTROY | PD_MSG_REQUEST;
        PD_MSG_FIELD_I(guid) = notifyArgs->OCR_SCHED_ARG_FIELD(OCR_SCHED_NOTIFY_EDT_DONE).guid;
        PD_MSG_FIELD_I(currentEdt) = notifyArgs->OCR_SCHED_ARG_FIELD(OCR_SCHED_NOTIFY_EDT_DONE).guid;
        PD_MSG_FIELD_I(properties) = 0;
        RESULT_ASSERT(pd->fcts.processMessage(pd, &msg, false), ==, 0);
#undef PD_MSG
#undef PD_TYPE
        return 0;
    }
    case OCR_SCHED_NOTIFY_EDT_SATISFIED:
        return OCR_ENOP;
    default:
        return OCR_ENOTSUP;
    }
}

static u8 gpuSchedulerHeuristicNoInvoke(ocrSchedulerHeuristic_t *self, ocrSchedulerOpArgs_t *opArgs, ocrRuntimeHint_t *hints) {
    return OCR_ENOTSUP; /* TRANSACT / ANALYZE: distributed scheduling is not this heuristic's business */
}

static u8 gpuSchedulerHeuristicNoSimulate(ocrSchedulerHeuristic_t *self, ocrSchedulerHeuristicContext_t *context, ocrSchedulerOpArgs_t *opArgs,
                                          ocrRuntimeHint_t *hints) {
    return OCR_ENOTSUP;
}

static ocrSchedulerHeuristic_t *newSchedulerHeuristicGpu(ocrSchedulerHeuristicFactory_t *factory, ocrParamList_t *perInstance) {
    ocrSchedulerHeuristic_t *self = (ocrSchedulerHeuristic_t *)runtimeChunkAlloc(sizeof(ocrSchedulerHeuristicGpu_t), PERSISTENT_CHUNK);
    initializeSchedulerHeuristicOcr(factory, self, perInstance);
    return self;
}

static void destructSchedulerHeuristicFactoryGpu(ocrSchedulerHeuristicFactory_t *factory) {
    runtimeChunkFree((u64)factory, NONPERSISTENT_CHUNK);
}

ocrSchedulerHeuristicFactory_t *newOcrSchedulerHeuristicFactoryGpu(ocrParamList_t *perType, u32 factoryId) {
    ocrSchedulerHeuristicFactory_t *base = (ocrSchedulerHeuristicFactory_t *)runtimeChunkAlloc(sizeof(ocrSchedulerHeuristicFactoryGpu_t), NONPERSISTENT_CHUNK);
    u32 op;
    base->factoryId = factoryId;
    base->instantiate = &newSchedulerHeuristicGpu;
    base->destruct = &destructSchedulerHeuristicFactoryGpu;
    base->fcts.switchRunlevel = FUNC_ADDR(u8 (*)(ocrSchedulerHeuristic_t *, ocrPolicyDomain_t *, ocrRunlevel_t, phase_t, u32,
                                                 void (*)(ocrPolicyDomain_t *, u64), u64), gpuSchedulerHeuristicSwitchRunlevel);
    base->fcts.destruct = FUNC_ADDR(void (*)(ocrSchedulerHeuristic_t *), gpuSchedulerHeuristicDestruct);
    base->fcts.update = FUNC_ADDR(u8 (*)(ocrSchedulerHeuristic_t *, u32), gpuSchedulerHeuristicUpdate);
    base->fcts.getContext = FUNC_ADDR(ocrSchedulerHeuristicContext_t *(*)(ocrSchedulerHeuristic_t *, ocrLocation_t), gpuSchedulerHeuristicGetContext);
    for(op = 0; op < NUM_SCHEDULER_HEURISTIC_OPS; ++op) {
        base->fcts.op[op].invoke = FUNC_ADDR(u8 (*)(ocrSchedulerHeuristic_t *, ocrSchedulerOpArgs_t *, ocrRuntimeHint_t *), gpuSchedulerHeuristicNoInvoke);
        base->fcts.op[op].simulate = FUNC_ADDR(u8 (*)(ocrSchedulerHeuristic_t *, ocrSchedulerHeuristicContext_t *, ocrSchedulerOpArgs_t *,
                                                      ocrRuntimeHint_t *), gpuSchedulerHeuristicNoSimulate);
        base->fcts.op[op].reject = NULL;
    }
    base->fcts.op[OCR_SCHEDULER_HEURISTIC_OP_GET_WORK].invoke = FUNC_ADDR(u8 (*)(ocrSchedulerHeuristic_t *, ocrSchedulerOpArgs_t *, ocrRuntimeHint_t *),
                                                                          gpuSchedulerHeuristicGetWorkInvoke);
    base->fcts.op[OCR_SCHEDULER_HEURISTIC_OP_NOTIFY].invoke = FUNC_ADDR(u8 (*)(ocrSchedulerHeuristic_t *, ocrSchedulerOpArgs_t *, ocrRuntimeHint_t *),
                                                                        gpuSchedulerHeuristicNotifyInvoke);
    return base;
}

#endif /* ENABLE_SCHEDULER_HEURISTIC_GPU */
