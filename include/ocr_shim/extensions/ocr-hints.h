/* ocr-hints.h -- hints on libocrshim: stored in the caller's ocrHint_t, never acted on (ocr/inc/extensions/ocr-hints.h:56-142) */
#ifndef OCR_SHIM_HINTS_H
#define OCR_SHIM_HINTS_H
#include "ocr.h"
#ifdef __cplusplus
extern "C" {
#endif
u8 ocrHintInit(ocrHint_t *hint, ocrHintType_t hintType);
u8 ocrSetHintValue(ocrHint_t *hint, ocrHintProp_t hintProp, u64 value);
u8 ocrUnsetHintValue(ocrHint_t *hint, ocrHintProp_t hintProp);
u8 ocrGetHintValue(ocrHint_t *hint, ocrHintProp_t hintProp, u64 *value);
u8 ocrSetHint(ocrGuid_t guid, ocrHint_t *hint);
u8 ocrGetHint(ocrGuid_t guid, ocrHint_t *hint);
#ifdef __cplusplus
}
#endif
#endif
