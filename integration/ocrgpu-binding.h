/*
 * ocrgpu-binding.h -- what the four reference-side factory files of this directory share.
 *
 * These files are the "five mechanical edits" of SURVEY.md s8b made concrete: each is a component a maintainer drops into the
 * reference tree (ocr/src/<family>/gpu/...), enables with ENABLE_<FAMILY>_<NAME> in ocr/build/<type>/ocr-config.h, adds to the
 * <family>-all.{h,c} tables and names in the machine config.  They include the reference's OWN headers (ocr-guid.h,
 * ocr-scheduler-heuristic.h, ocr-workpile.h, ocr-comm-platform.h, ocr-policy-domain.h) and fill its function tables by forwarding
 * to the C-ABI of include/ocrgpu.h, so every signature is checked by the compiler against the interface it claims to replace:
 * tests/test_integration_compile.py compiles them with gcc against /root/reference (where that tree is present).
 *
 * libocrgpu is batched and takes one caller at a time per engine; the reference calls these tables once per object from every
 * worker thread.  The bindings therefore serialise on one lock and pass batches of one.  That is the correctness shape of the
 * boundary, not its fast shape: a production binding stages the per-object calls of all workers into a ring and lets one
 * drainer issue them as real batches (last paragraph); libocrshim does exactly that for the public API.
 */
#ifndef OCRGPU_BINDING_H
#define OCRGPU_BINDING_H

#include "ocr-config.h"
#include "ocr-types.h"
#include "ocr-hal.h"
#include "ocrgpu.h"

/* the one engine of this policy domain: made by the GUID provider at RL_GUID_OK (the table must exist before any object is
 * created, counted-map-guid.c:140-186), used by the heuristic, the workpile and the comm-platform, freed by the provider */
typedef struct {
    ocrgpu_engine_t *engine;
    lock_t lock;              /* hal_lock: one caller at a time inside libocrgpu */
    u32 location, nLocations; /* this policy domain's LOCHOME id; how many partitions the comm-platform connects */
    u64 nGuids, nLookups, nReleased, nPushed, nPopped; /* calls that reached the engine (updated under the lock); printed at
                                                           tear-down when OCRGPU_BINDING_STATS is set: the evidence a test has
                                                           that a run really went through these factories */
} ocrgpuBinding_t;

extern ocrgpuBinding_t ocrgpuBinding;

#define OCRGPU_ENTER() hal_lock(&ocrgpuBinding.lock)
#define OCRGPU_LEAVE() hal_unlock(&ocrgpuBinding.lock)
#define OCRGPU_COUNT(field) __sync_fetch_and_add(&ocrgpuBinding.field, 1)
/* libocrgpu's negative codes -> the reference's u8 convention (0 ok, OCR_E* from ocr/inc/ocr-errors.h) */
static inline u8 ocrgpuToOcrError(int rc) {
    switch(rc) {
    case OCRGPU_OK:      return 0;
    case OCRGPU_EINVAL:  return OCR_EINVAL;
    case OCRGPU_ENOMEM:  return OCR_ENOMEM;
    case OCRGPU_ENOTSUP: return OCR_ENOTSUP;
    case OCRGPU_EFAULT:  return OCR_EFAULT;
    default:             return OCR_EIO;   /* OCRGPU_EHIP, OCRGPU_ENODEV */
    }
}
/* a guid as the u64 the engine's table is keyed by (GUID_BIT_COUNT == 64 builds: ocr/inc/ocr-types.h) */
#if GUID_BIT_COUNT != 64
#error the gpu bindings are written for 64-bit GUIDs
#endif
#define OCRGPU_KEY(g) ((u64)(g).guid)

#endif
