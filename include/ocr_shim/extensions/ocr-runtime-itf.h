/*
 * ocr-runtime-itf.h -- the runtime-interface extension on libocrshim (call-compatible with
 * ocr/inc/extensions/ocr-runtime-itf.h:44-109).  EDT bodies run on one host thread, so there is one worker; EDT-local
 * storage is OCRSHIM_ELS_SLOTS guids per EDT, zero when the body starts.
 */
#ifndef OCR_SHIM_RUNTIME_ITF_H
#define OCR_SHIM_RUNTIME_ITF_H
#include "ocr.h"
#ifdef __cplusplus
extern "C" {
#endif
#define OCRSHIM_ELS_SLOTS 4
ocrGuid_t ocrElsUserGet(u8 offset);              /* :51 */
void ocrElsUserSet(u8 offset, ocrGuid_t data);   /* :64 */
void currentEdtUserGet(ocrGuid_t *guid);         /* :81  NULL_GUID while mainEdt runs */
u64 ocrNbWorkers(void);                          /* :91 */
void ocrCurrentWorkerGuid(ocrGuid_t *guid);      /* :101 */
u8 ocrInformLegacyCodeBlocking(void);            /* :109 nothing to hand over: the caller IS the worker */
#ifdef __cplusplus
}
#endif
#endif
