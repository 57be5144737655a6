/*
 * ocr-labeling.h -- GUID labeling on libocrshim (call-compatible with ocr/inc/extensions/ocr-labeling.h:24-160).
 * A map / range is a block of consecutive guids reserved on the engine (ocrgpu_guid_reserve == the reference's
 * guidReserve, ocr/src/inc/ocr-guid.h:127-144); an object is created AT such a guid by passing GUID_PROP_IS_LABELED
 * to its create call (OCRGPU_GUIDS_GIVEN at the C-ABI).
 */
#ifndef OCR_SHIM_LABELING_H
#define OCR_SHIM_LABELING_H
#include "ocr.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef enum {
    GUID_USER_NONE = 0, GUID_USER_DB = 2, GUID_USER_EDT = 3, GUID_USER_EDT_TEMPLATE = 4, GUID_USER_EVENT_ONCE = 17,
    GUID_USER_EVENT_COUNTED = 18, GUID_USER_EVENT_IDEM = 19, GUID_USER_EVENT_STICKY = 21, GUID_USER_EVENT_LATCH = 22
} ocrGuidUserKind;
typedef ocrGuid_t (*ocrGuidMapFunc_t)(ocrGuid_t startGuid, u64 skipGuid, s64 *params, s64 *tuple);
u8 ocrGuidMapCreate(ocrGuid_t *mapGuid, u32 numParams, ocrGuidMapFunc_t mapFunc, s64 *params, u64 numberGuid, ocrGuidUserKind kind);
u8 ocrGuidRangeCreate(ocrGuid_t *rangeGuid, u64 numberGuid, ocrGuidUserKind kind);
u8 ocrGuidMapDestroy(ocrGuid_t mapGuid);
u8 ocrGuidFromLabel(ocrGuid_t *outGuid, ocrGuid_t mapGuid, s64 *tuple);
u8 ocrGuidFromIndex(ocrGuid_t *outGuid, ocrGuid_t rangeGuid, u64 idx);
u8 ocrGetGuidKind(ocrGuidUserKind *outKind, ocrGuid_t guid);
#ifdef __cplusplus
}
#endif
#endif
