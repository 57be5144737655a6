/*
 * rccl-comm-platform.c -- comm-platform "RCCL": ocrCommPlatformFcts_t (ocr/src/inc/ocr-comm-platform.h:55-199) over the
 * engine's own transport (ocrgpu_comm_init / ocrgpu_exchange: grouped ncclSend / ncclRecv over xGMI).  Counterpart of
 * ocr/src/comm-platform/mpi/mpi-comm-platform.c for the ONE message class of the dependence path:
 *
 *   sendMessage(PD_MSG_DEP_SATISFY, one-way)  the satisfy is handed to the engine (ocrgpu_event_satisfy_slot), which sees that the
 *                                             destination guid's LOCHOME bits name another partition and queues a 32-byte record
 *                                             for it -- in place of MPI_Isend of the marshalled message (:1055)
 *   pollMessage                               closes what is queued (ocrgpu_step) and runs one exchange round (ocrgpu_exchange):
 *                                             what arrives is applied on the device; nothing is surfaced as an ocrPolicyMsg_t, so
 *                                             the poll always reports POLL_NO_MESSAGE -- in place of MPI_Iprobe / MPI_Testany
 *                                             (:620-642,1193-1202) feeding processRequestEdt
 *   every other message type                  OCR_ENOTSUP: metadata cloning, remote datablock acquire, shutdown hand-shake stay with a
 *                                             comm-platform that carries ocrPolicyMsg_t (MPI); a deployment runs both
 *
 * The 128-byte communicator id goes from location 0 to the others through a file (OCRGPU_COMM_ID_FILE), the one out-of-band
 * channel a single node always has; mpirun-style launchers can hand it over differently.
 *
 * Registration: #define ENABLE_COMM_PLATFORM_RCCL; commPlatformRccl_id in commPlatformType_t; "RCCL" in commplatform_types[] and
 * `case commPlatformRccl_id: return newCommPlatformFactoryRccl(typeArg);` (ocr/src/comm-platform/comm-platform-all.c:12-80);
 * cfg: [CommPlatformType0] name=RCCL / [CommPlatformInst0] type=RCCL.
 */
#include "ocr-config.h"
#ifdef ENABLE_COMM_PLATFORM_RCCL

#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include "debug.h"
#include "ocr-comm-platform.h"
#include "ocr-errors.h"
#include "ocr-policy-domain.h"
#include "ocr-sysboot.h"
#include "ocrgpu-binding.h"

#define DEBUG_TYPE COMM_PLATFORM

typedef struct {
    ocrCommPlatform_t base;
    u64 recordsPerDest;   /* 32-byte records one level may queue for one destination (+1 header slot) */
    u64 maxRemoteDsts;    /* waiters registered here whose destination lives elsewhere */
    u64 sent;             /* satisfies handed to the engine since the last exchange */
} ocrCommPlatformRccl_t;

typedef struct {
    ocrCommPlatformFactory_t base;
} ocrCommPlatformFactoryRccl_t;

static void rcclCommDestruct(ocrCommPlatform_t *self) {
    runtimeChunkFree((u64)self, PERSISTENT_CHUNK);
}

/* location 0 writes the id file (to a temporary name, then renames it); the others wait for it */
static u8 rcclShareUniqueId(u32 location, u8 id[128]) {
    const char *path = getenv("OCRGPU_COMM_ID_FILE");
    if(path == NULL) return OCR_EINVAL;
    if(location == 0) {
        char tmp[512];
        if(ocrgpu_comm_get_unique_id(id) != OCRGPU_OK) return OCR_EIO;
        snprintf(tmp, sizeof tmp, "%s.tmp", path);
        FILE *f = fopen(tmp, "wb");
        if(f == NULL || fwrite(id, 1, 128, f) != 128) return OCR_EIO;
        fclose(f);
        return rename(tmp, path) == 0 ? 0 : OCR_EIO;
    }
    u32 tries;
    for(tries = 0; tries < 6000; ++tries) {
        FILE *f = fopen(path, "rb");
        if(f != NULL) {
            size_t n = fread(id, 1, 128, f);
            fclose(f);
            if(n == 128) return 0;
        }
        usleep(10000);
    }
    return OCR_EAGAIN; /* the id file never appeared */
}

/* OCRGPU_COMM_WINDOWS=1 (all locations on one node): peer windows in place of the communicator.  Every location writes its
 * 128-byte window handle to <OCRGPU_COMM_ID_FILE>.<location> and reads everybody's; then ocrgpu_peer_connect maps the windows */
static u8 rcclShareWindowHandles(u32 location, u32 nLocations) {
    const char *path = getenv("OCRGPU_COMM_ID_FILE");
    if(path == NULL || nLocations > 1024) return OCR_EINVAL;
    u8 *all = (u8 *)runtimeChunkAlloc((u64)nLocations * OCRGPU_PEER_HANDLE_BYTES, PERSISTENT_CHUNK);
    char name[600], tmp[640];
    OCRGPU_ENTER();
    int rc = ocrgpu_peer_export(ocrgpuBinding.engine, all + (u64)location * OCRGPU_PEER_HANDLE_BYTES);
    OCRGPU_LEAVE();
    u8 toReturn = rc == OCRGPU_OK ? 0 : ocrgpuToOcrError(rc);
    if(toReturn == 0) {
        snprintf(name, sizeof name, "%s.%u", path, location);
        snprintf(tmp, sizeof tmp, "%s.tmp", name);
        FILE *f = fopen(tmp, "wb");
        if(f == NULL || fwrite(all + (u64)location * OCRGPU_PEER_HANDLE_BYTES, 1, OCRGPU_PEER_HANDLE_BYTES, f) != OCRGPU_PEER_HANDLE_BYTES) toReturn = OCR_EIO;
        if(f != NULL) fclose(f);
        if(toReturn == 0 && rename(tmp, name) != 0) toReturn = OCR_EIO;
    }
    u32 p;
    for(p = 0; p < nLocations && toReturn == 0; ++p) {
        if(p == location) continue;
        u32 tries; u8 got = 0;
        snprintf(name, sizeof name, "%s.%u", path, p);
        for(tries = 0; tries < 6000 && !got; ++tries) {
            FILE *f = fopen(name, "rb");
            if(f != NULL) {
                got = fread(all + (u64)p * OCRGPU_PEER_HANDLE_BYTES, 1, OCRGPU_PEER_HANDLE_BYTES, f) == OCRGPU_PEER_HANDLE_BYTES;
                fclose(f);
            }
            if(!got) usleep(10000);
        }
        if(!got) toReturn = OCR_EAGAIN; /* that location's handle never appeared */
    }
    if(toReturn == 0) {
        OCRGPU_ENTER();
        rc = ocrgpu_peer_connect(ocrgpuBinding.engine, all);
        OCRGPU_LEAVE();
        if(rc != OCRGPU_OK) {
            DPRINTF(DEBUG_LVL_WARN, "RCCL comm-platform, peer windows: %s\n", ocrgpu_last_error());
            toReturn = ocrgpuToOcrError(rc);
        }
    }
    runtimeChunkFree((u64)all, PERSISTENT_CHUNK);
    return toReturn;
}

/* the communicator comes up once the engine exists (the GUID provider makes it at RL_GUID_OK) and goes away before it does:
 * where mpi-comm-platform.c has MPI_Init / the start barrier (:54,1726-1735) and MPI_Finalize (:60) */
static u8 rcclCommSwitchRunlevel(ocrCommPlatform_t *self, ocrPolicyDomain_t *PD, ocrRunlevel_t runlevel, phase_t phase, u32 properties,
                                 void (*callback)(ocrPolicyDomain_t *, u64), u64 val) {
    ocrCommPlatformRccl_t *rself = (ocrCommPlatformRccl_t *)self;
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
        if(properties & RL_BRING_UP) {
            self->pd = PD;
            ocrgpuBinding.nLocations = (u32)PD->neighborCount + 1;
        }
        break;
    case RL_COMPUTE_OK:
        if((properties & RL_BRING_UP) && RL_IS_FIRST_PHASE_UP(PD, RL_COMPUTE_OK, phase)) {
            u8 id[128];
            ASSERT(ocrgpuBinding.engine != NULL);
            OCRGPU_ENTER();
            /* NULL send buffer: the engine allocates the regions itself (a C host has no device allocator of its own) */
            int rc = ocrgpu_exchange_config(ocrgpuBinding.engine, ocrgpuBinding.nLocations, NULL, rself->recordsPerDest, rself->maxRemoteDsts);
            OCRGPU_LEAVE();
            if(rc != OCRGPU_OK) return ocrgpuToOcrError(rc);
            if(getenv("OCRGPU_COMM_WINDOWS") != NULL) /* one node: no collective, ocrgpu_exchange then runs over the windows */
                return rcclShareWindowHandles(ocrgpuBinding.location, ocrgpuBinding.nLocations);
            toReturn = rcclShareUniqueId(ocrgpuBinding.location, id);
            if(toReturn) return toReturn;
            OCRGPU_ENTER();
            rc = ocrgpu_comm_init(ocrgpuBinding.engine, id, (int)ocrgpuBinding.location, (int)ocrgpuBinding.nLocations);
            OCRGPU_LEAVE();
            if(rc != OCRGPU_OK) {
                DPRINTF(DEBUG_LVL_WARN, "RCCL comm-platform: %s\n", ocrgpu_last_error());
                toReturn = ocrgpuToOcrError(rc);
            }
        }
        if((properties & RL_TEAR_DOWN) && RL_IS_LAST_PHASE_DOWN(PD, RL_COMPUTE_OK, phase) && ocrgpuBinding.engine) {
            OCRGPU_ENTER();
            toReturn = ocrgpuToOcrError(ocrgpu_comm_fini(ocrgpuBinding.engine));
            if(toReturn == 0) toReturn = ocrgpuToOcrError(ocrgpu_peer_fini(ocrgpuBinding.engine));
            OCRGPU_LEAVE();
        }
        break;
    default:
        ASSERT(0);
    }
    return toReturn;
}

static u8 rcclCommSetMaxExpectedMessageSize(ocrCommPlatform_t *self, u64 size, u32 mask) {
    return 0; /* records are 32 bytes whatever the marshalled message would have been */
}

/* sendMessage (ocr-comm-platform.h:111-131): the message is consumed here, so whoever owns the buffer (PERSIST_MSG_PROP or not)
 * may reuse it when the call returns; one-way messages get id 0 */
static u8 rcclCommSendMessage(ocrCommPlatform_t *self, ocrLocation_t target, ocrPolicyMsg_t *message, u64 *id, u32 properties, u32 mask) {
    ocrCommPlatformRccl_t *rself = (ocrCommPlatformRccl_t *)self;
    if((message->type & PD_MSG_TYPE_ONLY) != PD_MSG_DEP_SATISFY || (properties & TWOWAY_MSG_PROP)) {
        DPRINTF(DEBUG_LVL_WARN, "RCCL comm-platform carries one-way PD_MSG_DEP_SATISFY only (got type 0x%"PRIx64")\n", (u64)(message->type & PD_MSG_TYPE_ONLY));
        return OCR_ENOTSUP;
    }
#define PD_MSG (message)
#define PD_TYPE PD_MSG_DEP_SATISFY
    u64 evt = PD_MSG_FIELD_I(guid.guid).guid, db = PD_MSG_FIELD_I(payload.guid).guid;
    u32 slot = PD_MSG_FIELD_I(slot);
#undef PD_MSG
#undef PD_TYPE
    OCRGPU_ENTER();
    int rc = ocrgpu_event_satisfy_slot(ocrgpuBinding.engine, 1, &evt, &db, &slot, OCRGPU_HOST);
    if(rc == OCRGPU_OK) ++rself->sent;
    OCRGPU_LEAVE();
    if(id) *id = 0;
    return ocrgpuToOcrError(rc);
}

/* pollMessage: one exchange round; arrivals are applied by the engine, none becomes a policy message */
static u8 rcclCommPollMessage(ocrCommPlatform_t *self, ocrPolicyMsg_t **msg, u32 properties, u32 *mask) {
    ocrCommPlatformRccl_t *rself = (ocrCommPlatformRccl_t *)self;
    u64 in = 0, activity = 0, sent = 0, queued = 0;
    OCRGPU_ENTER();
    int rc = ocrgpu_step(ocrgpuBinding.engine, 0, 0, NULL, NULL, NULL);           /* close what sendMessage queued */
    if(rc == OCRGPU_OK || rc == OCRGPU_EFAULT) rc = ocrgpu_exchange(ocrgpuBinding.engine, &in, &activity, &sent, &queued);
    rself->sent = 0;
    OCRGPU_LEAVE();
    if(msg) *msg = NULL;
    if(rc != OCRGPU_OK) return ocrgpuToOcrError(rc);
    return POLL_NO_MESSAGE;
}

static u8 rcclCommWaitMessage(ocrCommPlatform_t *self, ocrPolicyMsg_t **msg, u32 properties, u32 *mask) {
    return OCR_ENOTSUP; /* nothing ever arrives as a policy message: there is nothing to wait for */
}

static u8 rcclCommDestructMessage(ocrCommPlatform_t *self, ocrPolicyMsg_t *msg) {
    return 0;
}

static ocrMsgStatus_t rcclCommGetMessageStatus(ocrCommPlatform_t *self, u64 id) {
    return MSG_SEND_OK; /* a satisfy handed to the engine is as good as sent: the next exchange carries it */
}

static u64 rcclCommGetSeqIdAtNeighbor(ocrCommPlatform_t *self, ocrLocation_t neighborLoc, u64 neighborId) {
    return (u64)neighborLoc; /* locations are the communicator's ranks */
}

static void initializeCommPlatformRccl(ocrCommPlatformFactory_t *factory, ocrCommPlatform_t *base, ocrParamList_t *perInstance) {
    ocrCommPlatformRccl_t *rself = (ocrCommPlatformRccl_t *)base;
    initializeCommPlatformOcr(factory, base, perInstance);
    rself->recordsPerDest = 1ULL << 16;   /* cfg keys read in populate_inst would set these two */
    rself->maxRemoteDsts = 1ULL << 20;
    rself->sent = 0;
}

static ocrCommPlatform_t *newCommPlatformRccl(ocrCommPlatformFactory_t *factory, ocrParamList_t *perInstance) {
    ocrCommPlatformRccl_t *self = (ocrCommPlatformRccl_t *)runtimeChunkAlloc(sizeof(ocrCommPlatformRccl_t), PERSISTENT_CHUNK);
    self->base.location = ((paramListCommPlatformInst_t *)perInstance)->location;
    self->base.fcts = factory->platformFcts;
    factory->initialize(factory, (ocrCommPlatform_t *)self, perInstance);
    return (ocrCommPlatform_t *)self;
}

static void destructCommPlatformFactoryRccl(ocrCommPlatformFactory_t *factory) {
    runtimeChunkFree((u64)factory, NONPERSISTENT_CHUNK);
}

ocrCommPlatformFactory_t *newCommPlatformFactoryRccl(ocrParamList_t *perType) {
    ocrCommPlatformFactory_t *base = (ocrCommPlatformFactory_t *)runtimeChunkAlloc(sizeof(ocrCommPlatformFactoryRccl_t), NONPERSISTENT_CHUNK);
    base->instantiate = &newCommPlatformRccl;
    base->initialize = &initializeCommPlatformRccl;
    base->destruct = FUNC_ADDR(void (*)(ocrCommPlatformFactory_t *), destructCommPlatformFactoryRccl);
    base->platformFcts.destruct = FUNC_ADDR(void (*)(ocrCommPlatform_t *), rcclCommDestruct);
    base->platformFcts.switchRunlevel = FUNC_ADDR(u8 (*)(ocrCommPlatform_t *, ocrPolicyDomain_t *, ocrRunlevel_t, phase_t, u32,
                                                         void (*)(ocrPolicyDomain_t *, u64), u64), rcclCommSwitchRunlevel);
    base->platformFcts.setMaxExpectedMessageSize = FUNC_ADDR(u8 (*)(ocrCommPlatform_t *, u64, u32), rcclCommSetMaxExpectedMessageSize);
    base->platformFcts.sendMessage = FUNC_ADDR(u8 (*)(ocrCommPlatform_t *, ocrLocation_t, ocrPolicyMsg_t *, u64 *, u32, u32), rcclCommSendMessage);
    base->platformFcts.pollMessage = FUNC_ADDR(u8 (*)(ocrCommPlatform_t *, ocrPolicyMsg_t **, u32, u32 *), rcclCommPollMessage);
    base->platformFcts.waitMessage = FUNC_ADDR(u8 (*)(ocrCommPlatform_t *, ocrPolicyMsg_t **, u32, u32 *), rcclCommWaitMessage);
    base->platformFcts.sendMessageMT = NULL;
    base->platformFcts.pollMessageMT = NULL;
    base->platformFcts.waitMessageMT = NULL;
    base->platformFcts.destructMessage = FUNC_ADDR(u8 (*)(ocrCommPlatform_t *, ocrPolicyMsg_t *), rcclCommDestructMessage);
    base->platformFcts.getMessageStatus = FUNC_ADDR(ocrMsgStatus_t (*)(ocrCommPlatform_t *, u64), rcclCommGetMessageStatus);
    base->platformFcts.getSeqIdAtNeighbor = FUNC_ADDR(u64 (*)(ocrCommPlatform_t *, ocrLocation_t, u64), rcclCommGetSeqIdAtNeighbor);
    return base;
}

#endif /* ENABLE_COMM_PLATFORM_RCCL */
