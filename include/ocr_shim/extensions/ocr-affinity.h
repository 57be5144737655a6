/*
 * ocr-affinity.h -- the affinity extension on libocrshim (call-compatible with ocr/inc/extensions/ocr-affinity.h).
 * One engine = one policy domain, so there is exactly one affinity; placement hints are accepted and ignored.
 */
#ifndef OCR_SHIM_AFFINITY_H
#define OCR_SHIM_AFFINITY_H
#include "ocr.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef enum { AFFINITY_CURRENT, AFFINITY_PD, AFFINITY_PD_MASTER, AFFINITY_SIM, AFFINITY_GUID } ocrAffinityKind;
u8 ocrAffinityCount(ocrAffinityKind kind, u64 *count);
u8 ocrAffinityGet(ocrAffinityKind kind, u64 *count, ocrGuid_t *affinities);
u8 ocrAffinityGetAt(ocrAffinityKind kind, u64 idx, ocrGuid_t *affinity);
u8 ocrAffinityGetCurrent(ocrGuid_t *affinity);
u8 ocrAffinityQuery(ocrGuid_t guid, u64 *count, ocrGuid_t *affinities);
u64 ocrAffinityToHintValue(ocrGuid_t affinity);
#ifdef __cplusplus
}
#endif
#endif
