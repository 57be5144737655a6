/*
 * gpuhash-guid.c -- GUID provider "GPUHASH": the reference's ocrGuidProviderFcts_t (ocr/src/inc/ocr-guid.h:86-314) over the
 * engine's open-addressing table in HBM.  Counterpart of ocr/src/guid/counted/counted-map-guid.c (COUNTED_MAP) with
 * ocr/src/utils/hashtable.c's bucket-locked chains replaced by ocrgpu_guid_put / get / tryput / remove, and the same GUID bit
 * layout (LOCHOME | KIND | COUNTER, ocr/src/guid/guid-bitmap.h:22-71), so getKind / getLocation stay bit tests.
 *
 * The VALUE stored for a guid stays a HOST pointer (callers dereference fguid->metaDataPtr at once: hc-event.c:1755,
 * hc-task.c:922); the engine's own objects, when the event / task bookkeeping moves onto it too, are found through the same
 * table under the same guid.
 *
 * Registration: #define ENABLE_GUID_GPUHASH in ocr-config.h; guidGpuHash_id in guidType_t
 * (ocr/src/guid/guid-all.h:15-26); "GPUHASH" at the same index of guid_types[] and
 *     case guidGpuHash_id: return newGuidProviderFactoryGpuHash(typeArg, (u32)type);
 * in newGuidProviderFactory (ocr/src/guid/guid-all.c:10-41); cfg: [GuidType0] name=GPUHASH / [GuidInst0] type=GPUHASH.
 */
#include "ocr-config.h"
#ifdef ENABLE_GUID_GPUHASH

#include <stdio.h>
#include <stdlib.h>
#include "debug.h"
#include "ocr-errors.h"
#include "ocr-guid.h"
#include "ocr-policy-domain.h"
#include "ocr-sysboot.h"
#include "ocrgpu-binding.h"

#define DEBUG_TYPE GUID

ocrgpuBinding_t ocrgpuBinding = {NULL, INIT_LOCK, 0, 1, 0, 0, 0, 0, 0};

typedef struct {
    ocrGuidProvider_t base;
    u64 objectCapacity;   /* cfg: the number of live objects the table is sized for (load <= 0.5) */
} ocrGuidProviderGpuHash_t;

typedef struct {
    ocrGuidProviderFactory_t base;
} ocrGuidProviderFactoryGpuHash_t;

static void gpuHashDestruct(ocrGuidProvider_t *self) {
    runtimeChunkFree((u64)self, PERSISTENT_CHUNK);
}

/* table made at RL_GUID_OK on the way up, freed at RL_MEMORY_OK on the way down: the same points as counted-map-guid.c:140-186 */
static u8 gpuHashSwitchRunlevel(ocrGuidProvider_t *self, ocrPolicyDomain_t *PD, ocrRunlevel_t runlevel, phase_t phase, u32 properties,
                                void (*callback)(ocrPolicyDomain_t *, u64), u64 val) {
    u8 toReturn = 0;
    ASSERT(callback == NULL);
    ASSERT((properties & RL_REQUEST) && !(properties & RL_RESPONSE) && !(properties & RL_RELEASE));
    ASSERT(!(properties & RL_FROM_MSG));
    switch(runlevel) {
    case RL_CONFIG_PARSE:
    case RL_NETWORK_OK:
        break;
    case RL_PD_OK:
        if((properties & RL_BRING_UP) && RL_IS_FIRST_PHASE_UP(PD, RL_PD_OK, phase))
            self->pd = PD;
        break;
    case RL_MEMORY_OK:
        if((properties & RL_TEAR_DOWN) && RL_IS_LAST_PHASE_DOWN(PD, RL_MEMORY_OK, phase) && ocrgpuBinding.engine) {
            if(getenv("OCRGPU_BINDING_STATS"))
                fprintf(stderr, "GPUHASH binding: %llu guids made, %llu lookups, %llu released; GPU work-list: %llu pushed, %llu popped\n",
                        (unsigned long long)ocrgpuBinding.nGuids, (unsigned long long)ocrgpuBinding.nLookups, (unsigned long long)ocrgpuBinding.nReleased,
                        (unsigned long long)ocrgpuBinding.nPushed, (unsigned long long)ocrgpuBinding.nPopped);
            toReturn = ocrgpuToOcrError(ocrgpu_fini(ocrgpuBinding.engine));
            ocrgpuBinding.engine = NULL;
        }
        break;
    case RL_GUID_OK:
        if((properties & RL_BRING_UP) && RL_IS_LAST_PHASE_UP(PD, RL_GUID_OK, phase) && !ocrgpuBinding.engine) {
            ocrGuidProviderGpuHash_t *rself = (ocrGuidProviderGpuHash_t *)self;
            ocrgpu_config_t cfg = {0};
            cfg.max_edt = (u32)rself->objectCapacity; cfg.max_evt = (u32)rself->objectCapacity; cfg.max_db = (u32)rself->objectCapacity;
            cfg.max_slots = 8 * rself->objectCapacity; cfg.max_edges = 8 * rself->objectCapacity;
            cfg.guid_slots = 0;                                     /* 2x the object capacity, rounded up to a power of two */
            cfg.location = (u32)PD->myLocation;
            ocrgpuBinding.location = cfg.location;
            toReturn = ocrgpuToOcrError(ocrgpu_init(&ocrgpuBinding.engine, 0, &cfg));
            if(toReturn)
                DPRINTF(DEBUG_LVL_WARN, "GPUHASH: ocrgpu_init failed: %s\n", ocrgpu_last_error());
        }
        break;
    case RL_COMPUTE_OK:
    case RL_USER_OK:
        break;
    default:
        ASSERT(0);
    }
    return toReturn;
}

/* one engine call under the binding's lock */
#define GPUHASH_CALL(expr) ({ OCRGPU_ENTER(); int _rc = (expr); OCRGPU_LEAVE(); ocrgpuToOcrError(_rc); })

/* guidReserve (ocr-guid.h:127-144): numberGuids consecutive counters of one kind; the range is [start, start + n) with step 1.
 * The engine hands the whole range back; only its first guid is needed here. */
static u8 gpuHashGuidReserve(ocrGuidProvider_t *self, ocrGuid_t *startGuid, u64 *skipGuid, u64 numberGuids, ocrGuidKind guidType) {
    if(numberGuids == 0) return OCR_EINVAL;
    ocrPolicyDomain_t *pd = self->pd;
    u64 *range = (u64 *)pd->fcts.pdMalloc(pd, numberGuids * sizeof(u64));
    u8 rc = GPUHASH_CALL(ocrgpu_guid_reserve(ocrgpuBinding.engine, (u32)guidType, numberGuids, range, OCRGPU_HOST));
    if(rc == 0) {
        startGuid->guid = range[0];
        *skipGuid = 1;
    }
    pd->fcts.pdFree(pd, range);
    return rc;
}

static u8 gpuHashGuidUnreserve(ocrGuidProvider_t *self, ocrGuid_t startGuid, u64 skipGuid, u64 numberGuids) {
    return 0; /* counters are not recycled (49 bits of them: counted-map-guid.c:43) */
}

/* getGuid (counted-map-guid.c:222-258): a fresh guid of `kind`; recorded with `val` when asked to */
static u8 gpuHashGetGuid(ocrGuidProvider_t *self, ocrGuid_t *guid, u64 val, ocrGuidKind kind, ocrLocation_t targetLoc, u32 properties) {
    u64 g = 0;
    u8 rc = GPUHASH_CALL(ocrgpu_guid_reserve(ocrgpuBinding.engine, (u32)kind, 1, &g, OCRGPU_HOST));
    if(rc) return rc;
    OCRGPU_COUNT(nGuids);
    guid->guid = g;
    if(properties & GUID_PROP_TORECORD)
        rc = GPUHASH_CALL(ocrgpu_guid_put(ocrgpuBinding.engine, 1, &g, &val, OCRGPU_HOST));
    return rc;
}

/* createGuid (counted-map-guid.c:266-330): `size` bytes of metadata from the policy domain, then a guid for them -- or, with
 * GUID_PROP_ISVALID, the guid the caller brought (labeled creation: GUID_PROP_CHECK makes an existing entry an error) */
static u8 gpuHashCreateGuid(ocrGuidProvider_t *self, ocrFatGuid_t *fguid, u64 size, ocrGuidKind kind, ocrLocation_t targetLoc, u32 properties) {
    PD_MSG_STACK(msg);
    ocrPolicyDomain_t *policy = NULL;
    getCurrentEnv(&policy, NULL, NULL, &msg);
#define PD_MSG (&msg)
#define PD_TYPE PD_MSG_MEM_ALLOC
    msg.type = PD_MSG_MEM_ALLOC | PD_MSG_REQUEST | PD_MSG_REQ_RESPONSE;
    PD_MSG_FIELD_I(size) = size;
    PD_MSG_FIELD_I(properties) = 0;
    PD_MSG_FIELD_I(type) = GUID_MEMTYPE;
    RESULT_PROPAGATE(policy->fcts.processMessage(policy, &msg, true));
    void *ptr = (void *)PD_MSG_FIELD_O(ptr);
#undef PD_MSG
#undef PD_TYPE
    u8 rc = 0;
    if(properties & GUID_PROP_ISVALID) {
        if(properties & GUID_PROP_TORECORD) {
            u64 g = OCRGPU_KEY(fguid->guid), v = (u64)ptr, held = 0;
            if(properties & GUID_PROP_CHECK) { /* hashtableConcBucketLockedTryPut: somebody else may have created it first */
                rc = GPUHASH_CALL(ocrgpu_guid_tryput(ocrgpuBinding.engine, 1, &g, &v, &held, OCRGPU_HOST));
                if(rc == 0 && held != v) rc = OCR_EGUIDEXISTS;
            } else {
                rc = GPUHASH_CALL(ocrgpu_guid_put(ocrgpuBinding.engine, 1, &g, &v, OCRGPU_HOST));
            }
        }
    } else {
        rc = gpuHashGetGuid(self, &(fguid->guid), (u64)ptr, kind, targetLoc, GUID_PROP_TORECORD);
    }
    fguid->metaDataPtr = ptr;
    return rc;
}

/* getVal (counted-map-guid.c:394-406 over hashtable.c:461): the one lookup every message hop pays */
static u8 gpuHashGetVal(ocrGuidProvider_t *self, ocrGuid_t guid, u64 *val, ocrGuidKind *kind, u32 mode, MdProxy_t **proxy) {
    u64 g = OCRGPU_KEY(guid), v = 0;
    u8 rc = GPUHASH_CALL(ocrgpu_guid_get(ocrgpuBinding.engine, 1, &g, &v, OCRGPU_HOST));
    OCRGPU_COUNT(nLookups);
    *val = v;                                   /* 0 when the guid does not resolve, as the reference's NULL entry */
    if(kind) *kind = (ocrGuidKind)OCRGPU_GUID_KIND(g);
    if(proxy) *proxy = NULL;                    /* remote metadata is not cloned by this provider */
    return rc;
}

static u8 gpuHashGetKind(ocrGuidProvider_t *self, ocrGuid_t guid, ocrGuidKind *kind) {
    *kind = (ocrGuidKind)OCRGPU_GUID_KIND(OCRGPU_KEY(guid));      /* getKindFromGuid, guid-bitmap-based.c:13-19 */
    return 0;
}

static u8 gpuHashGetLocation(ocrGuidProvider_t *self, ocrGuid_t guid, ocrLocation_t *location) {
    *location = (ocrLocation_t)OCRGPU_GUID_LOCATION(OCRGPU_KEY(guid)); /* extractLocIdFromGuid, guid-bitmap-based.c:25-33 */
    return 0;
}

static u8 gpuHashRegisterGuid(ocrGuidProvider_t *self, ocrGuid_t guid, u64 val) {
    u64 g = OCRGPU_KEY(guid);
    return GPUHASH_CALL(ocrgpu_guid_put(ocrgpuBinding.engine, 1, &g, &val, OCRGPU_HOST));
}

static u8 gpuHashUnregisterGuid(ocrGuidProvider_t *self, ocrGuid_t guid, u64 **val) {
    u64 g = OCRGPU_KEY(guid), v = 0;
    u8 rc = GPUHASH_CALL(ocrgpu_guid_get(ocrgpuBinding.engine, 1, &g, &v, OCRGPU_HOST));
    if(rc == 0) rc = GPUHASH_CALL(ocrgpu_guid_remove(ocrgpuBinding.engine, 1, &g, OCRGPU_HOST));
    if(val) *val = (u64 *)v;
    return rc;
}

/* releaseGuid (counted-map-guid.c:540-572): the entry goes (its slot is reused by a later put), and with releaseVal the metadata */
static u8 gpuHashReleaseGuid(ocrGuidProvider_t *self, ocrFatGuid_t fguid, bool releaseVal) {
    u64 g = OCRGPU_KEY(fguid.guid);
    u8 rc = GPUHASH_CALL(ocrgpu_guid_remove(ocrgpuBinding.engine, 1, &g, OCRGPU_HOST));
    OCRGPU_COUNT(nReleased);
    if(rc == 0 && releaseVal) {
        ASSERT(fguid.metaDataPtr != NULL);
        PD_MSG_STACK(msg);
        ocrPolicyDomain_t *policy = NULL;
        getCurrentEnv(&policy, NULL, NULL, &msg);
#define PD_MSG (&msg)
#define PD_TYPE PD_MSG_MEM_UNALLOC
        msg.type = PD_MSG_MEM_UNALLOC | PD_MSG_REQUEST;
        PD_MSG_FIELD_I(allocatingPD.guid) = NULL_GUID;
        PD_MSG_FIELD_I(allocatingPD.metaDataPtr) = NULL;
        PD_MSG_FIELD_I(allocator.guid) = NULL_GUID;
        PD_MSG_FIELD_I(allocator.metaDataPtr) = NULL;
        PD_MSG_FIELD_I(ptr) = fguid.metaDataPtr;
        PD_MSG_FIELD_I(type) = GUID_MEMTYPE;
        PD_MSG_FIELD_I(properties) = 0;
        RESULT_PROPAGATE(policy->fcts.processMessage(policy, &msg, true));
#undef PD_MSG
#undef PD_TYPE
    }
    return rc;
}

static ocrGuidProvider_t *newGuidProviderGpuHash(ocrGuidProviderFactory_t *factory, ocrParamList_t *perInstance) {
    ocrGuidProvider_t *base = (ocrGuidProvider_t *)runtimeChunkAlloc(sizeof(ocrGuidProviderGpuHash_t), PERSISTENT_CHUNK);
    base->fcts = factory->providerFcts;
    base->pd = NULL;
    base->id = factory->factoryId;
    ((ocrGuidProviderGpuHash_t *)base)->objectCapacity = 1ULL << 20; /* a cfg key ("capacity") read in populate_inst would set it */
    return base;
}

static void destructGuidProviderFactoryGpuHash(ocrGuidProviderFactory_t *factory) {
    runtimeChunkFree((u64)factory, NONPERSISTENT_CHUNK);
}

ocrGuidProviderFactory_t *newGuidProviderFactoryGpuHash(ocrParamList_t *typeArg, u32 factoryId) {
    ocrGuidProviderFactory_t *base = (ocrGuidProviderFactory_t *)runtimeChunkAlloc(sizeof(ocrGuidProviderFactoryGpuHash_t), NONPERSISTENT_CHUNK);
    base->instantiate = &newGuidProviderGpuHash;
    base->destruct = &destructGuidProviderFactoryGpuHash;
    base->factoryId = factoryId;
    base->providerFcts.destruct = FUNC_ADDR(void (*)(ocrGuidProvider_t *), gpuHashDestruct);
    base->providerFcts.switchRunlevel = FUNC_ADDR(u8 (*)(ocrGuidProvider_t *, ocrPolicyDomain_t *, ocrRunlevel_t, phase_t, u32,
                                                         void (*)(ocrPolicyDomain_t *, u64), u64), gpuHashSwitchRunlevel);
    base->providerFcts.guidReserve = FUNC_ADDR(u8 (*)(ocrGuidProvider_t *, ocrGuid_t *, u64 *, u64, ocrGuidKind), gpuHashGuidReserve);
    base->providerFcts.guidUnreserve = FUNC_ADDR(u8 (*)(ocrGuidProvider_t *, ocrGuid_t, u64, u64), gpuHashGuidUnreserve);
    base->providerFcts.getGuid = FUNC_ADDR(u8 (*)(ocrGuidProvider_t *, ocrGuid_t *, u64, ocrGuidKind, ocrLocation_t, u32), gpuHashGetGuid);
    base->providerFcts.createGuid = FUNC_ADDR(u8 (*)(ocrGuidProvider_t *, ocrFatGuid_t *, u64, ocrGuidKind, ocrLocation_t, u32), gpuHashCreateGuid);
    base->providerFcts.getVal = FUNC_ADDR(u8 (*)(ocrGuidProvider_t *, ocrGuid_t, u64 *, ocrGuidKind *, u32, MdProxy_t **), gpuHashGetVal);
    base->providerFcts.getKind = FUNC_ADDR(u8 (*)(ocrGuidProvider_t *, ocrGuid_t, ocrGuidKind *), gpuHashGetKind);
    base->providerFcts.getLocation = FUNC_ADDR(u8 (*)(ocrGuidProvider_t *, ocrGuid_t, ocrLocation_t *), gpuHashGetLocation);
    base->providerFcts.registerGuid = FUNC_ADDR(u8 (*)(ocrGuidProvider_t *, ocrGuid_t, u64), gpuHashRegisterGuid);
    base->providerFcts.unregisterGuid = FUNC_ADDR(u8 (*)(ocrGuidProvider_t *, ocrGuid_t, u64 **), gpuHashUnregisterGuid);
    base->providerFcts.releaseGuid = FUNC_ADDR(u8 (*)(ocrGuidProvider_t *, ocrFatGuid_t, bool), gpuHashReleaseGuid);
    return base;
}

#endif /* ENABLE_GUID_GPUHASH */
