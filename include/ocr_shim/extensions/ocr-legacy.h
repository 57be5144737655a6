/*
 * ocr-legacy.h -- legacy mode on libocrshim (call-compatible with ocr/inc/extensions/ocr-legacy.h:42-180): a program
 * with its own main() brings the runtime up, starts EDTs from outside any EDT and waits for what they produce.
 * ocrLegacyBlockProgress may also be called from inside an EDT: the caller then runs the other ready EDTs underneath
 * itself until the event is satisfied (the reference parks the calling worker and lets the others run).
 */
#ifndef OCR_SHIM_LEGACY_H
#define OCR_SHIM_LEGACY_H
#include "ocr.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct _ocrConfig_t {
    int userArgc;          /* argc without the -ocr: arguments */
    char **userArgv;
    const char *iniFile;   /* recorded, not read: capacities come from OCRSHIM_* environment variables */
} ocrConfig_t;
#define LEGACY_PROP_NONE ((u16)0x0)
#define LEGACY_PROP_WAIT_FOR_CREATE ((u16)0x1)
#define LEGACY_PROP_WAIT_IF_CREATED ((u16)0x2)
#define LEGACY_PROP_CHECK ((u16)0x4)
void ocrParseArgs(int argc, const char *argv[], ocrConfig_t *ocrConfig);                    /* :60 */
void ocrLegacyInit(ocrGuid_t *legacyContext, ocrConfig_t *ocrConfig);                       /* :74 */
u8 ocrLegacyFinalize(ocrGuid_t legacyContext, bool runUntilShutdown);                       /* :97 */
u8 ocrLegacySpawnOCR(ocrGuid_t *handle, ocrGuid_t finishEdtTemplate, u64 paramc, u64 *paramv,
                     u64 depc, ocrGuid_t *depv, ocrGuid_t legacyContext);                   /* :130 */
ocrGuid_t ocrWait(ocrGuid_t outputEvent);                                                   /* :146 */
u8 ocrLegacyBlockProgress(ocrGuid_t handle, ocrGuid_t *guid, void **result, u64 *size, u16 properties); /* :172 */
#ifdef __cplusplus
}
#endif
#endif
