/*
 * ocr_opstream.h -- the "op stream": one OCR program's API calls as data.
 *
 * An op stream is what a mainEdt would do, written down: create events / datablocks /
 * EDTs, add dependences, satisfy events -- plus, per EDT, a tiny "script" of the satisfies
 * its body performs before it returns.  The same stream is
 *   - replayed through the REAL OCR API on the reference runtime (oracle/ocr_replay.c),
 *   - interpreted by the CPU restatement (oracle/engine_ref.c),
 *   - submitted in batches through the C-ABI of the MI355X engine (include/ocrgpu.h),
 * so all three see bit-identical inputs.  Objects are named by logical index (never by
 * raw GUID bits: PTR GUIDs are heap addresses and COUNTED_MAP GUIDs embed a global
 * atomic counter, ocr/src/guid/guid-bitmap-based.c:91 -- SURVEY.md s7 hard part iv).
 *
 * Every op corresponds 1:1 to a public OCR call:
 *   EVT_CREATE  -> ocrEventCreate / ocrEventCreateParams   (ocr/inc/ocr-edt.h:63,87)
 *   DB_CREATE   -> ocrDbCreate + ocrDbRelease               (ocr/inc/ocr-db.h:78,128)
 *   EDT_CREATE  -> ocrEdtCreate (depv == NULL)              (ocr/inc/ocr-edt.h:282)
 *   ADD_DEP     -> ocrAddDependence                         (ocr/inc/ocr-edt.h:347)
 *   SATISFY     -> ocrEventSatisfySlot                      (ocr/inc/ocr-edt.h:166)
 *   EVT_DESTROY -> ocrEventDestroy                          (ocr/inc/ocr-edt.h:101)
 */
#ifndef OCR_OPSTREAM_H
#define OCR_OPSTREAM_H
#include <stdint.h>

#define OCROPS_MAGIC 0x003153504F52434FULL /* "OCROPS1\0" little-endian */

enum { /* ocrops_op_t.op */
    OCROP_EVT_CREATE = 1, OCROP_DB_CREATE = 2, OCROP_EDT_CREATE = 3,
    OCROP_ADD_DEP = 4, OCROP_SATISFY = 5, OCROP_EVT_DESTROY = 6
};

/* object references: class in the top byte, index in the low 56 bits */
enum {
    OCRREF_NULL = 0,   /* NULL_GUID (whole ref == 0)                                   */
    OCRREF_EDT = 1,    /* EDT #idx                                                      */
    OCRREF_EVT = 2,    /* explicitly created event #idx                                 */
    OCRREF_DB = 3,     /* datablock #idx created by an OCROP_DB_CREATE                  */
    OCRREF_OUTEVT = 4, /* the output event of EDT #idx (ocrEdtCreate's outputEvent)     */
    OCRREF_OUTDB = 5,  /* the datablock EDT #idx's HASH body creates and returns        */
    OCRREF_DEPV = 6,   /* scripts only: the guid received on pre-slot #idx (depv[idx])  */
    OCRREF_UNINIT = 7  /* UNINITIALIZED_GUID                                            */
};
#define OCRREF(cls, idx) (((uint64_t)(cls) << 56) | (uint64_t)(idx))
#define OCRREF_CLS(r) ((uint32_t)((r) >> 56))
#define OCRREF_IDX(r) ((r) & 0x00FFFFFFFFFFFFFFULL)

/* EDT body kinds (low byte of ocrops_edt_t.body) and flags */
enum {
    OCRBODY_NOP = 0,  /* returns NULL_GUID                                              */
    OCRBODY_HASH = 1, /* folds id + every depv[i] payload into a u64, stores it in a new DB, returns that DB */
    OCRBODY_FWD0 = 2, /* returns depv[0].guid                                           */
    OCRBODY_HOST = 3, /* GPU engine only: the body is host code; the EDT is handed out by ocrgpu_pop_ready and its
                         epilogue (output-event satisfy) comes back through ocrgpu_event_satisfy_slot */
    OCRBODY_TILE = 4  /* streams with tile datablocks (hdr.db_words == 64): the pre-slots that carry a datablock carry 1..3
                         tiles; the LAST of them is updated IN PLACE, C -= X * Y^T over wrapping 64-bit integers (X = the first
                         input tile, or C itself when there is none; Y = the second input, or X), and its guid is returned.
                         val = checksum of the updated tile.  (8x8 int64 tiles, integer recurrences) */
};
#define OCRBODY_KIND(b) ((b) & 0xFF)
#define OCRBODY_F_OUTEVT 0x100u /* created with an output event (ONCE, hc-task.c:955)  */
#define OCRBODY_F_SINK   0x200u /* the single sink: dumps results, calls ocrShutdown   */

/* event kinds == ocrEventTypes_t (ocr/inc/ocr-types.h:272-309) */
enum { OCREVT_ONCE = 1, OCREVT_IDEM = 2, OCREVT_STICKY = 3, OCREVT_LATCH = 4,
       OCREVT_COUNTED = 5, OCREVT_CHANNEL = 6 };
/* latch pre-slots (ocr/inc/ocr-types.h:317-319) */
enum { OCRLATCH_DECR = 0, OCRLATCH_INCR = 1 };
/* dependence modes == ocrDbAccessMode_t (ocr/inc/ocr-types.h:150-156) */
enum { OCRMODE_NULL = 0, OCRMODE_CONST = 1, OCRMODE_RW = 2, OCRMODE_EW = 4, OCRMODE_RO = 8 };

typedef struct {
    uint64_t magic;
    uint32_t n_edt, n_evt, n_db, n_main, n_script, sink_edt;
    uint32_t db_words; /* 0: a datablock is {value, id} (16 bytes).  64: every OCROP_DB_CREATE datablock also carries an 8x8 tile of
                          64-bit integers behind those two words (528 bytes), word w initialised to ocrops_tile_init(value, w)  */
    uint32_t pad[7];
} ocrops_hdr_t; /* 64 B */

typedef struct { uint32_t depc, body, script_begin, script_len; } ocrops_edt_t; /* 16 B */
typedef struct { uint32_t kind; int32_t arg; } ocrops_evt_t; /* arg: LATCH initial counter / COUNTED nbDeps */
typedef struct { uint8_t op, mode; uint16_t pad; uint32_t slot; uint64_t a, b; } ocrops_op_t; /* 24 B */

/* file layout: hdr | edt[n_edt] | evt[n_evt] | db_init u64[n_db] | main[n_main] | script[n_script] */

/* ---- result record every executor produces for a stream (the parity contract) ----
 * level[e]     s32  topological level at which EDT e fired (-1: never fired)
 * val[e]       u64  HASH checksum of EDT e (0 for other bodies / unfired)
 * evt_state[v] u64  v in [0,n_evt) explicit events, n_evt+e = output event of EDT e
 * db_val[d]    u64  d in [0,n_db) created DBs, n_db+e = output DB of EDT e (0 if none);
 *                   for a tile datablock (hdr.db_words == 64): ocrops_tile_checksum of its final contents
 */
enum { OCREVS_DESTROYED = 0, OCREVS_LIVE_UNSAT = 1, OCREVS_LIVE_SAT = 2, OCREVS_ABSENT = 3 };
/* evt_state = code | aux<<8 ; aux = (u32)latch counter for a live LATCH,
 * dbid+1 (0 = NULL_GUID) of the stored payload for a satisfied persistent event      */
#define OCREVS(code, aux) ((uint64_t)(code) | ((uint64_t)(uint32_t)(aux) << 8))

#define OCRRES_MAGIC 0x0031534552524F43ULL /* "CORRES1\0" */
typedef struct { uint64_t magic; uint32_t n_edt, n_evt_total, n_db_total, n_fired; uint32_t pad[10]; } ocrres_hdr_t; /* 64 B */
/* file layout: hdr | level s32[n_edt] | val u64[n_edt] | evt_state u64[n_evt_total] | db_val u64[n_db_total] */

#ifdef __HIPCC__
#define OCROPS_HD __host__ __device__
#else
#define OCROPS_HD
#endif
/* The integer mix every executor uses for HASH bodies (SplitMix64 finaliser). */
OCROPS_HD static inline uint64_t ocrops_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
/* HASH body: v0 = mix(id); per slot i in order: v = mix(v ^ (payload_val + (dbid+1)*K + i)) */
#define OCROPS_HASH_K 0xD6E8FEB86659FD93ULL
OCROPS_HD static inline uint64_t ocrops_hash_init(uint64_t edt_id) { return ocrops_mix64(edt_id ^ 0x0C5200000000ULL); }
OCROPS_HD static inline uint64_t ocrops_hash_step(uint64_t v, uint32_t slot, uint64_t payload_val, uint64_t dbid_plus1) {
    return ocrops_mix64(v ^ (payload_val + dbid_plus1 * OCROPS_HASH_K + slot));
}
/* TILE bodies: 8x8 tiles of wrapping 64-bit integers, row-major.  Every executor uses these definitions; the element form
 * (ocrops_tile_elem) is what a GPU lane computes, the whole-tile form is for the CPU executors. */
#define OCROPS_TILE_N 8u
#define OCROPS_TILE_WORDS 64u
OCROPS_HD static inline uint64_t ocrops_tile_init(uint64_t value, uint32_t w) { return ocrops_mix64(value + w); }
/* element (r, c) of C - X * Y^T, all operands read BEFORE anything is written (X and Y may alias C) */
OCROPS_HD static inline uint64_t ocrops_tile_elem(const uint64_t *C, const uint64_t *X, const uint64_t *Y, uint32_t r, uint32_t c) {
    uint64_t acc = C[r * OCROPS_TILE_N + c];
    for (uint32_t t = 0; t < OCROPS_TILE_N; ++t) acc -= X[r * OCROPS_TILE_N + t] * Y[c * OCROPS_TILE_N + t];
    return acc;
}
/* checksum = wrapping sum of the terms (order-free, so a wave can reduce it) */
OCROPS_HD static inline uint64_t ocrops_tile_term(uint64_t word, uint32_t w) { return ocrops_mix64(word + (uint64_t)w * OCROPS_HASH_K); }
OCROPS_HD static inline uint64_t ocrops_tile_checksum(const uint64_t *C) {
    uint64_t s = 0;
    for (uint32_t w = 0; w < OCROPS_TILE_WORDS; ++w) s += ocrops_tile_term(C[w], w);
    return s;
}
/* C -= X * Y^T in place (X = NULL: X = C; Y = NULL: Y = X); returns the checksum of the result */
OCROPS_HD static inline uint64_t ocrops_tile_update(uint64_t *C, const uint64_t *X, const uint64_t *Y) {
    uint64_t out[OCROPS_TILE_WORDS];
    if (!X) X = C;
    if (!Y) Y = X;
    for (uint32_t r = 0; r < OCROPS_TILE_N; ++r)
        for (uint32_t c = 0; c < OCROPS_TILE_N; ++c) out[r * OCROPS_TILE_N + c] = ocrops_tile_elem(C, X, Y, r, c);
    for (uint32_t w = 0; w < OCROPS_TILE_WORDS; ++w) C[w] = out[w];
    return ocrops_tile_checksum(C);
}
#endif
