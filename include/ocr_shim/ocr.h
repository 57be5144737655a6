/*
 * ocr.h -- the part of the public OCR API that libocrshim implements on top of libocrgpu (include/ocrgpu.h).
 *
 * An OCR program (a file that defines mainEdt and includes "ocr.h") compiles unchanged against this header and
 * links with ocr_amd/libocrshim.a + libocrgpu.so instead of libocr_x86: dependence bookkeeping runs on the MI355X
 * engine, EDT bodies run on the host between levels (the reference's deferred-API execution
 * model, hc-policy.c:1523-1720).  Written from the API's documented contract; each declaration names the reference
 * header it has to stay call-compatible with.  64-bit GUID build only (no OCR_ENABLE_128_BIT_GUID).
 */
#ifndef OCR_SHIM_OCR_H
#define OCR_SHIM_OCR_H
#include <inttypes.h>
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar and guid types (ocr/inc/ocr-types.h:28-68) ---- */
typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;
typedef int64_t s64;
typedef int32_t s32;
typedef int8_t s8;
#ifndef __cplusplus
#ifndef bool
typedef u8 bool;
#define true 1
#define false 0
#endif
#endif
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif

typedef struct { intptr_t guid; } ocrGuid_t;
#define NULL_GUID_INITIALIZER {.guid = 0x0}
#define NULL_GUID ((ocrGuid_t)NULL_GUID_INITIALIZER)
#define UNINITIALIZED_GUID_INITIALIZER {.guid = -2}
#define UNINITIALIZED_GUID ((ocrGuid_t)UNINITIALIZED_GUID_INITIALIZER)
#define ERROR_GUID_INITIALIZER {.guid = -1}
#define ERROR_GUID ((ocrGuid_t)ERROR_GUID_INITIALIZER)

/* guid predicates and printing (ocr/inc/ocr-guid-functions.h:41-168) */
static inline bool ocrGuidIsNull(ocrGuid_t g) { return g.guid == 0; }
static inline bool ocrGuidIsUninitialized(ocrGuid_t g) { return g.guid == -2; }
static inline bool ocrGuidIsError(ocrGuid_t g) { return g.guid == -1; }
static inline bool ocrGuidIsEq(ocrGuid_t a, ocrGuid_t b) { return a.guid == b.guid; }
static inline bool ocrGuidIsLt(ocrGuid_t a, ocrGuid_t b) { return a.guid < b.guid; }
#define IS_GUID_NULL(g) ocrGuidIsNull(g)
#define IS_GUID_UNINITIALIZED(g) ocrGuidIsUninitialized(g)
#define IS_GUID_ERROR(g) ocrGuidIsError(g)
#define IS_GUID_EQUAL(a, b) ocrGuidIsEq(a, b)
#define IS_GUID_LESS_THAN(a, b) ocrGuidIsLt(a, b)
#define GUIDA(v) ((v).guid)
#define GUIDF "0x%" PRIxPTR
#define GUIDSx GUIDF
#define GUIDLx GUIDF
#define GUIDFS(g) GUIDA(g)
#define GUIDFL(g) GUIDA(g)

/* ---- hints: kept in the caller's struct, never acted on (ocr/inc/ocr-types.h:36,346-413) ---- */
typedef enum { OCR_HINT_UNDEF_T, OCR_HINT_EDT_T, OCR_HINT_DB_T, OCR_HINT_EVT_T, OCR_HINT_GROUP_T } ocrHintType_t;
typedef enum {
    OCR_HINT_EDT_PROP_START, OCR_HINT_EDT_PRIORITY, OCR_HINT_EDT_SLOT_MAX_ACCESS, OCR_HINT_EDT_AFFINITY, OCR_HINT_EDT_DISPERSE,
    OCR_HINT_EDT_SPACE, OCR_HINT_EDT_TIME, OCR_HINT_EDT_STATS_HW_CYCLES, OCR_HINT_EDT_STATS_L1_HITS, OCR_HINT_EDT_STATS_L1_MISSES,
    OCR_HINT_EDT_STATS_FLOAT_OPS, OCR_HINT_EDT_PROP_END,
    OCR_HINT_DB_PROP_START, OCR_HINT_DB_AFFINITY, OCR_HINT_DB_NEAR, OCR_HINT_DB_INTER, OCR_HINT_DB_FAR, OCR_HINT_DB_HIGHBW, OCR_HINT_DB_PROP_END,
    OCR_HINT_EVT_PROP_START, OCR_HINT_EVT_PROP_END, OCR_HINT_GROUP_PROP_START, OCR_HINT_GROUP_PROP_END
} ocrHintProp_t;
typedef uint64_t ocrHintVal_t;
typedef struct { ocrHintType_t type; u64 propMask; u64 args[OCR_HINT_GROUP_PROP_END + 1]; } ocrHint_t;
#define NULL_HINT ((ocrHint_t *)0x0)
#define OCR_HINT_EDT_DISPERSE_FAR 0
#define OCR_HINT_EDT_DISPERSE_NEAR 1

/* ---- datablocks (ocr/inc/ocr-types.h:120-174, ocr/inc/ocr-db.h:78-128) ---- */
typedef enum { NO_ALLOC = 0 } ocrInDbAllocator_t;
typedef enum {
    DB_MODE_NULL = 0x0, DB_MODE_CONST = 0x1, DB_MODE_RW = 0x2, DB_MODE_EW = 0x4, DB_MODE_RO = 0x8
} ocrDbAccessMode_t;
#define DB_DEFAULT_MODE ((ocrDbAccessMode_t)DB_MODE_RW)
#define DB_PROP_NONE ((u16)0x0)
#define DB_PROP_NO_ACQUIRE ((u16)0x10)
#define DB_PROP_SINGLE_ASSIGNMENT ((u16)0x20)
u8 ocrDbCreate(ocrGuid_t *db, void **addr, u64 len, u16 flags, ocrHint_t *hint, ocrInDbAllocator_t allocator);
u8 ocrDbDestroy(ocrGuid_t db);
u8 ocrDbRelease(ocrGuid_t db);

/* ---- EDTs (ocr/inc/ocr-types.h:197-252, ocr/inc/ocr-edt.h:209-298) ---- */
typedef struct {
    ocrGuid_t guid;         /* datablock that arrived on this pre-slot, or NULL_GUID */
    void *ptr;              /* its payload, or NULL                                  */
    ocrDbAccessMode_t mode;
} ocrEdtDep_t;
#define EDT_PROP_NONE ((u16)0x0)
#define EDT_PROP_FINISH ((u16)0x1)
#define EDT_PROP_NO_HINT ((u16)0x2)
#define EDT_PROP_LONG ((u16)0x4)
#define EDT_PROP_OEVT_VALID ((u16)0x8) /* *outputEvent names an existing event the EDT's epilogue satisfies */
#define EDT_PARAM_UNK ((u32)-1)
#define EDT_PARAM_DEF ((u32)-2)
typedef ocrGuid_t (*ocrEdt_t)(u32 paramc, u64 *paramv, u32 depc, ocrEdtDep_t depv[]);
u8 ocrEdtTemplateCreate_internal(ocrGuid_t *guid, ocrEdt_t funcPtr, u32 paramc, u32 depc, const char *funcName);
#define ocrEdtTemplateCreate(guid, funcPtr, paramc, depc) \
    ocrEdtTemplateCreate_internal((guid), (funcPtr), (paramc), (depc), #funcPtr)
u8 ocrEdtTemplateDestroy(ocrGuid_t guid);
u8 ocrEdtCreate(ocrGuid_t *guid, ocrGuid_t templateGuid, u32 paramc, u64 *paramv, u32 depc, ocrGuid_t *depv,
                u16 properties, ocrHint_t *hint, ocrGuid_t *outputEvent);
u8 ocrEdtDestroy(ocrGuid_t guid);

/* ---- events and dependences (ocr/inc/ocr-types.h:272-332, ocr/inc/ocr-edt.h:30-166,347) ---- */
typedef enum {
    OCR_EVENT_ONCE_T = 1, OCR_EVENT_IDEM_T = 2, OCR_EVENT_STICKY_T = 3, OCR_EVENT_LATCH_T = 4,
    OCR_EVENT_COUNTED_T = 5, OCR_EVENT_CHANNEL_T = 6, OCR_EVENT_T_MAX
} ocrEventTypes_t;
typedef enum { OCR_EVENT_LATCH_DECR_SLOT = 0, OCR_EVENT_LATCH_INCR_SLOT = 1 } ocrLatchEventSlot_t;
#define EVT_PROP_NONE ((u16)0x0)
#define EVT_PROP_TAKES_ARG ((u16)0x1)
typedef struct { u64 counter; } ocrEventLatchParams_t;
typedef struct { u64 nbDeps; } ocrEventCounterParams_t;
#define EVENT_CHANNEL_UNBOUNDED ((u32)-1)
typedef struct { u32 maxGen, nbSat, nbDeps; } ocrEventChannelParams_t;
typedef struct {
    union {
        ocrEventLatchParams_t EVENT_LATCH;
        ocrEventCounterParams_t EVENT_COUNTED;
        ocrEventChannelParams_t EVENT_CHANNEL;
    };
} ocrEventParams_t;
u8 ocrEventCreate(ocrGuid_t *guid, ocrEventTypes_t eventType, u16 properties);
u8 ocrEventCreateParams(ocrGuid_t *guid, ocrEventTypes_t eventType, u16 properties, ocrEventParams_t *params);
u8 ocrEventDestroy(ocrGuid_t guid);
u8 ocrEventSatisfy(ocrGuid_t eventGuid, ocrGuid_t dataGuid);
u8 ocrEventSatisfySlot(ocrGuid_t eventGuid, ocrGuid_t dataGuid, u32 slot);
u8 ocrAddDependence(ocrGuid_t source, ocrGuid_t destination, u32 slot, ocrDbAccessMode_t mode);

/* ---- program control and console (ocr/inc/ocr.h:54-98, ocr/inc/ocr-std.h:49-120) ---- */
void ocrShutdown(void);
void ocrAbort(u8 errorCode);
u64 getArgc(void *dbPtr);
char *getArgv(void *dbPtr, u64 count);
u32 PRINTF(const char *fmt, ...);
u32 SNPRINTF(char *buf, u32 size, const char *fmt, ...);
void _ocrAssert(bool val, const char *str, const char *file, u32 line);
#define ASSERT(a) do { _ocrAssert((bool)((a) != 0), #a, __FILE__, __LINE__); } while (0);
#define VERIFY(cond, format, ...)                                                                          \
    do {                                                                                                   \
        if (!(cond)) PRINTF("FAILURE @ '%s:%" PRId32 "' " format, __FILE__, __LINE__, ##__VA_ARGS__);      \
        else PRINTF("PASSED @ '%s:%" PRId32 "' " format, __FILE__, __LINE__, ##__VA_ARGS__);               \
    } while (0);

/* error codes used by the calls above (ocr/inc/ocr-errors.h:23-51) */
#define OCR_EINVAL 22
#define OCR_ENOMEM 12
#define OCR_ENOTSUP 95
#define OCR_EGUIDEXISTS 100
#define OCR_ENOENT 2
#define OCR_EPERM 1
#define OCR_EACCES 13
#define OCR_EBUSY 16
#define OCR_ECANCELED 125

/* labeled creation (ocr/inc/ocr-types.h:503-520): the guid passed in names the object to create */
#define GUID_PROP_NONE ((u16)0x0)
#define GUID_PROP_IS_LABELED ((u16)0x1000)
#define GUID_PROP_CHECK ((u16)0x3000)   /* + report OCR_EGUIDEXISTS instead of creating twice */
#define GUID_PROP_BLOCK ((u16)0x5000)
ocrGuid_t addValueToGuid(ocrGuid_t input, u64 value);   /* ocr/inc/ocr-types.h:56 */

/* the program's entry point, defined by the program */
ocrGuid_t mainEdt(u32 paramc, u64 *paramv, u32 depc, ocrEdtDep_t depv[]);

#ifdef __cplusplus
}
#endif
#endif
