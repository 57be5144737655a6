/*
 * ocrgpu.h -- C-ABI of libocrgpu, the MI355X (gfx950) dependence-resolution engine.
 *
 * Plain C, opaque handle, plain pointers and sizes.  This is the boundary the OCR component
 * factories bind to (INTEGRATION.md shows the C stubs): a GUID provider, a scheduler
 * heuristic / workpile and the event+task bookkeeping behind the policy domain's DEP_* and
 * WORK_* messages call these entry points instead of walking ocrPolicyMsg_t through
 * hcPolicyDomainProcessMessage (ocr/src/policy-domain/hc/hc-policy.c:1722).
 *
 * Calling convention
 *  - every call returns 0 or a negative OCRGPU_E* code (ocr/inc/ocr-errors.h convention: 0 = ok);
 *  - all calls are BATCHED: n objects per call, struct-of-arrays;  `where` says whether the
 *    arrays live in host memory (OCRGPU_HOST, staged over PCIe by the engine) or are device
 *    pointers on the engine's GPU (OCRGPU_DEVICE, no copy);
 *  - batches are applied in call order on the engine's stream; within one batch the order of
 *    elements is not observable for well-formed programs (the reference gives the same freedom
 *    to concurrent workers);
 *  - one engine = one GPU = one policy-domain partition (location); not thread safe: one
 *    caller at a time per engine (the host shim serialises).
 */
#ifndef OCRGPU_H
#define OCRGPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ocrgpu_engine ocrgpu_engine_t;

enum { OCRGPU_HOST = 0, OCRGPU_DEVICE = 1 };
/* OR into `where` of ocrgpu_event_create / ocrgpu_db_create / ocrgpu_edt_create: the guid arrays are INPUTS holding
 * guids the caller took from ocrgpu_guid_reserve earlier (the reference's guidReserve + labeled create,
 * ocr/src/inc/ocr-guid.h:127-144; GUID_PROP_IS_LABELED ocr/inc/ocr-types.h:504).  Lets a host shim hand a guid back
 * from ocrEdtCreate at once and create the object later in a batch (deferred API, hc-policy.c:1523-1720).
 * ocrgpu_edt_create: guids_out = EDT guids, outevt_guids_out = output-event guids (kind ONCE; 0 or NULL = none),
 * outdb_guids_out is ignored.  ocrgpu_db_create also takes the guid of ANOTHER location's DB: the entry is then this
 * engine's copy of that DB (proxy DB, hc-dist-policy.c:382-440), which satisfies issued here carry by value. */
#define OCRGPU_GUIDS_GIVEN 4

enum {
    OCRGPU_OK = 0,
    OCRGPU_EINVAL = -22,   /* bad argument (OCR_EINVAL)                                  */
    OCRGPU_ENOMEM = -12,   /* a capacity given at init is exhausted (OCR_ENOMEM)         */
    OCRGPU_EHIP = -5,      /* a HIP runtime call failed; ocrgpu_last_error() has the text */
    OCRGPU_ENODEV = -19,   /* no usable gfx950 device                                    */
    OCRGPU_EFAULT = -14,   /* the device flagged a user-level error during this call (stats.err[] moved: a malformed op,
                              a record dropped by a full queue, ...).  The call has done everything else it had to do and
                              its outputs are filled; returned once per movement by ocrgpu_add_dependence,
                              ocrgpu_event_satisfy_slot, ocrgpu_resolve and ocrgpu_step                                 */
    OCRGPU_ENOTSUP = -95
};

/* GUID bit layout == the reference's COUNTED_MAP provider: LOCHOME(10) | KIND(5) | COUNTER(49),
 * MSB->LSB (ocr/src/guid/guid-bitmap.h:22-71, GUID_PROVIDER_LOCID_SIZE=10 ocr/build/common.mk:92,
 * counted-map-guid.c:43).  Kinds == ocrGuidKind (ocr/src/inc/ocr-guid-kind.h:19-44). */
enum {
    OCRGPU_KIND_DB = 2, OCRGPU_KIND_EDT = 3, OCRGPU_KIND_EVENT = 16 /* bit 0x10 */,
    OCRGPU_KIND_EVT_ONCE = 17, OCRGPU_KIND_EVT_COUNTED = 18, OCRGPU_KIND_EVT_IDEM = 19,
    OCRGPU_KIND_EVT_STICKY = 21, OCRGPU_KIND_EVT_LATCH = 22, OCRGPU_KIND_EVT_CHANNEL = 23
};
#define OCRGPU_GUID_KIND(g)     ((uint32_t)(((uint64_t)(g) >> 49) & 0x1F))   /* getKindFromGuid  guid-bitmap-based.c:19 */
#define OCRGPU_GUID_LOCATION(g) ((uint32_t)((uint64_t)(g) >> 54))            /* extractLocIdFromGuid :31 */
#define OCRGPU_GUID_COUNTER(g)  ((uint64_t)(g) & ((1ULL << 49) - 1))
#define OCRGPU_NULL_GUID   0ULL
#define OCRGPU_ERROR_GUID  (~0ULL)          /* ocr/inc/ocr-types.h:61-68 */
#define OCRGPU_UNINIT_GUID (~0ULL - 1)

typedef struct {
    uint32_t max_edt;      /* EDTs this partition can hold                                   */
    uint32_t max_evt;      /* explicitly created events (output events come with the EDTs)   */
    uint32_t max_db;       /* explicitly created datablocks (output DBs come with the EDTs) + proxy copies of other
                              partitions' datablocks that arriving records carry (one per such record, from the top) */
    uint32_t location;     /* LOCHOME id of this partition (== rank)                          */
    uint64_t max_slots;    /* sum of depc over all EDTs                                       */
    uint64_t max_edges;    /* waiter registrations (event -> EDT|event dependences), < 2^32   */
    uint64_t guid_slots;   /* GUID table slots (power of two); 0 = 2x the object capacity     */
    uint32_t max_levels;   /* level boundaries recorded (0 = 1<<20)                           */
    uint32_t flags;        /* OCRGPU_F_*                                                      */
} ocrgpu_config_t;
/* Environment overrides read by ocrgpu_init (capacities the config struct has no field for, and two launch-shape
 * knobs kept for measurements): OCRGPU_MAX_LAZY (held-back COUNTED registrations, default 4*max_evt+1024, at most 4M),
 * OCRGPU_CHAN_SLOTS (queued CHANNEL payloads, default min(max_edges, 4M)), OCRGPU_SPREAD_WGS (workgroups a short work
 * list is dealt over, default 256), OCRGPU_STEP_GRID (grid of a step launch, default 2048). */
#define OCRGPU_F_NANNY 1u  /* detect double satisfy of an EDT pre-slot (hc-task.c:1243); +1 atomic per edge */
#define OCRGPU_F_TIMING 2u /* bracket resolve's kernels with HIP events on the engine stream   */
#define OCRGPU_F_NO_ASYNC 32u /* ocrgpu_resolve always runs level by level (one launch per level).  By default a resolve whose DAG is made of
                              * ONCE / LATCH events, NOP / HASH bodies without scripts and dependences whose payload is known at registration, and
                              * that starts from at most 65 536 ready EDTs, runs as ONE barrier-free kernel with the level carried as data (same
                              * record).  Environment: OCRGPU_ASYNC=0 never, 2 whenever the DAG allows; OCRGPU_ASYNC_WAVES; OCRGPU_ASYNC_MAX_ROW. */
#define OCRGPU_F_WIDE_DB 16u /* every ocrgpu_db_create datablock also carries an 8x8 tile of 64-bit integers (512 bytes, word w =
                              * ocrops_tile_init(value, w)) that OCRBODY_TILE bodies update in place -- the payload SURVEY.md s8d.3 names
                              * for the tiled Cholesky (regular-datablock.c:55-215: a datablock is its guid + a buffer the EDT writes).
                              * ocrgpu_read_dbs reports ocrops_tile_checksum of the tile.  Across partitions (ocrgpu_exchange_config) a
                              * satisfy that carries such a datablock carries the WHOLE datablock, as hc-dist clones it on a remote acquire
                              * (ProxyDb_t, hc-dist-policy.c:382-700): it takes 23 slots of the destination's send region (head + tile +
                              * home guid) instead of one, the receiver keeps a copy under the home guid (or takes the contents in place when
                              * the datablock comes home), and a TILE body that updated a copy sends the contents home (23 more slots: the
                              * write-back of a released RW / EW proxy).  Size records_per_dest and max_db (one copy per arriving tile) for it. */
#define OCRGPU_F_LOCKABLE_DB 8u /* the cfg's datablockfactory=Lockable instead of Regular: an EDT whose pre-slots are all
                                 * satisfied first acquires its datablocks, in ascending guid order, each once, in the mode
                                 * its ocrAddDependence named (DB_MODE_CONST / RW / EW / RO), waits in line on a busy one
                                 * and only then becomes ready; the level that fires it releases them (iterateDbFrontier
                                 * hc-task.c:484-544, lockableAcquireInternal / lockableRelease
                                 * lockable-datablock.c:139-263, 330-507).  At most 4096 pre-slots per EDT (a larger
                                 * ocrgpu_edt_create entry is a malformed op).  Which of several contenders gets a datablock
                                 * first is a race, as it is between the reference's workers; everything but the levels of
                                 * the contenders is deterministic.
                                 * Across partitions (ocrgpu_exchange_config) a datablock made by ocrgpu_db_create is arbitrated
                                 * by the partition that made it, as under hc-dist (the acquire / release messages go to the
                                 * owner's policy domain, hc-dist-policy.c:1027-1031): a satisfy that carries such a datablock
                                 * takes 2 slots of its send region (the second names the original), an EDT whose next
                                 * datablock -- in ascending guid order, one order everywhere -- is a copy asks the owner with
                                 * a record and goes on when the grant arrives, the level that fires it sends the releases.
                                 * Grants and releases are records that set off records: not with OCRGPU_RX_NO_CASCADE;
                                 * ocrgpu_snapshot is refused.  The output
                                 * datablock of an EDT that travels to another partition is a private copy there (arbitrated
                                 * where the copy lives).                                                                 */

typedef struct {
    uint64_t levels;        /* topological levels run by the last resolve                      */
    uint64_t fired;         /* EDTs fired                                                      */
    uint64_t edt_sat;       /* satisfies applied to EDT pre-slots   (satisfyTaskHc calls)      */
    uint64_t evt_sat;       /* satisfies applied to event pre-slots (event.satisfy calls)      */
    uint64_t launches;      /* k_step launches (incl. the no-op ones issued ahead)             */
    uint64_t step_kernel_ns;/* OCRGPU_F_TIMING: HIP-event time over all k_step launches        */
    uint64_t err[8];        /* [0] double slot satisfy [1] waiter on dead non-persistent event
                               [2] STICKY re-satisfied [3] satisfy/destroy of a dead event
                               [4] malformed op / unknown guid [5] queue overflow [6] level log overflow */
    uint64_t async_runs;    /* resolves so far that ran as ONE barrier-free kernel (see OCRGPU_F_NO_ASYNC); such a run counts one launch */
} ocrgpu_stats_t;

/* ---- lifetime ------------------------------------------------------------------------- */
int ocrgpu_init(ocrgpu_engine_t **out, int device, const ocrgpu_config_t *cfg);
int ocrgpu_fini(ocrgpu_engine_t *e);
int ocrgpu_set_stream(ocrgpu_engine_t *e, void *hip_stream);    /* default: the NULL stream */
/* A stream of the engine's own (hipStreamNonBlocking, destroyed with the engine) in place of the NULL stream: for several engines in one
 * process -- partitions that share a GPU, one host thread that drives several GPUs -- whose kernels would otherwise all stand in one
 * line.  The reference's counterpart is one comm / compute worker thread per policy domain (hc-comm-worker.c, hc-worker.c:52-265):
 * policy domains do not wait for each other except through messages.  Device arrays passed with OCRGPU_DEVICE afterwards must be
 * complete when the call is made. */
int ocrgpu_stream_create(ocrgpu_engine_t *e);
const char *ocrgpu_last_error(void);
int ocrgpu_abi_version(void);

/* ---- GUID provider (replaces ocrGuidProviderFcts_t, ocr/src/inc/ocr-guid.h:86-314, COUNTED_MAP impl
 *      ocr/src/guid/counted/counted-map-guid.c:222-572 + ocr/src/utils/hashtable.c:461-554) ------ */
/* getGuid/createGuid: reserve n consecutive counters of `kind`; guids_out[i] = LOC|KIND|counter+i  (:222-300) */
int ocrgpu_guid_reserve(ocrgpu_engine_t *e, uint32_t kind, uint64_t n, uint64_t *guids_out, int where);
/* registerGuid / hashtable put (:302, hashtable.c:498): vals must be non-zero */
int ocrgpu_guid_put(ocrgpu_engine_t *e, uint64_t n, const uint64_t *guids, const uint64_t *vals, int where);
/* hashtableConcBucketLockedTryPut (hashtable.c:527-554; labeled create with GUID_PROP_CHECK, labeled-guid.c): insert
 * guids[i] -> vals[i] unless the guid is already in the table; held_out[i] = the value the table holds for it afterwards
 * (== vals[i] when this call inserted it).  Safe for several elements of one batch naming the same guid: one wins. */
int ocrgpu_guid_tryput(ocrgpu_engine_t *e, uint64_t n, const uint64_t *guids, const uint64_t *vals, uint64_t *held_out, int where);
/* getVal (:394-406, hashtable.c:461): vals_out[i] = 0 when the guid does not resolve */
int ocrgpu_guid_get(ocrgpu_engine_t *e, uint64_t n, const uint64_t *guids, uint64_t *vals_out, int where);
/* releaseGuid / unregisterGuid (:527-572, hashtable.c:527): the slot becomes a tombstone that a later put takes over */
int ocrgpu_guid_remove(ocrgpu_engine_t *e, uint64_t n, const uint64_t *guids, int where);

/* ---- object creation (replace newEventHc hc-event.c:1558-1765, newTaskHc hc-task.c:871-1136,
 *      hcAllocateDb hc-policy.c:752-784).  Objects get engine indices in creation order;
 *      ocrgpu_guid_get() on an engine-made guid returns (index + 1).                       ------ */
/* ocrEventCreateParams (ocr-edt.c:19): kind = ocrEventTypes_t 1..6; arg = LATCH initial counter (hc-event.c:1700),
 * COUNTED nbDeps > 0 (:1600-1612), CHANNEL maxGen = generations that may be in flight (nbSat = nbDeps = 1, the only
 * shape the reference supports, :1629; <= 0 = EVENT_CHANNEL_UNBOUNDED, held to 1024 here).  A CHANNEL pairs its k-th
 * satisfy with its k-th registered waiter (:2081-2156); satisfies of one channel issued in ONE call (or by bodies of
 * one level) take their generations in no particular order, as concurrent workers do in the reference; callers that
 * need an order use separate calls.  Each CHANNEL takes maxGen entries of max_edges. */
int ocrgpu_event_create(ocrgpu_engine_t *e, uint64_t n, const uint32_t *kind, const int32_t *arg,
                        uint64_t *guids_out, int where);
/* ocrDbCreate (ocr-db.h:78): a DB here is a 16-byte payload {value, tag}                    */
int ocrgpu_db_create(ocrgpu_engine_t *e, uint64_t n, const uint64_t *value, const uint64_t *tag,
                     uint64_t *guids_out, int where);
/* ocrEdtCreate with depv == NULL (ocr-edt.c:295): body = OCRBODY_* | OCRBODY_F_OUTEVT (ocr_opstream.h),
 * param0 = paramv[0].  outevt_guids_out[i] = guid of the ONCE output event or NULL_GUID.  depc==0
 * EDTs become ready at once (hc-task.c:1129-1134).  outevt_guids_out / outdb_guids_out may be NULL. */
int ocrgpu_edt_create(ocrgpu_engine_t *e, uint64_t n, const uint32_t *depc, const uint32_t *body,
                      const uint64_t *param0, uint64_t *guids_out, uint64_t *outevt_guids_out,
                      uint64_t *outdb_guids_out, int where);
/* the satisfies an EDT body performs before returning (its own ocrEventSatisfySlot calls):
 * EDT edt[i] runs ops[begin[i] .. begin[i]+len[i]).  target = event guid; payload_kind selects
 * NULL_GUID / a DB guid in `payload` / the EDT's own output DB / depv[payload].guid            */
/* The ops of one script apply in order, as the calls of one EDT body do; every op of a level takes effect in that
 * level's event phase, so a destroy of an event the same script satisfies belongs in a later call, not the script. */
typedef struct { uint64_t target, payload; uint32_t slot; uint8_t op, payload_kind; uint16_t pad; } ocrgpu_script_op_t;
enum { OCRGPU_PAY_NULL = 0, OCRGPU_PAY_GUID = 1, OCRGPU_PAY_OUTDB = 2, OCRGPU_PAY_DEPV = 3 };
enum { OCRGPU_SOP_SATISFY = 5, OCRGPU_SOP_EVT_DESTROY = 6 };
int ocrgpu_edt_set_scripts(ocrgpu_engine_t *e, uint64_t n, const uint64_t *edt, const uint32_t *begin,
                           const uint32_t *len, uint64_t n_ops, const ocrgpu_script_op_t *ops, int where);
/* tag written into the DB a HASH body creates = tag_base + param0 (what the replayed OCR program writes) */
int ocrgpu_set_hash_tag_base(ocrgpu_engine_t *e, uint64_t tag_base);

/* ---- dependence API (replace PD_MSG_DEP_ADD / DEP_REGWAITER / DEP_REGSIGNALER / DEP_SATISFY,
 *      hc-policy.c:2775-3280; registerWaiterEventHc* hc-event.c:1065-1163; satisfy* :437-886) ---- */
/* ocrAddDependence(src, dst, slot, mode) (ocr-edt.c:551) */
int ocrgpu_add_dependence(ocrgpu_engine_t *e, uint64_t n, const uint64_t *src, const uint64_t *dst,
                          const uint32_t *slot, const uint32_t *mode, int where);
/* ocrEventSatisfySlot(evt, db, slot) (ocr-edt.c:127).  Takes effect (fan-out through events) at
 * once; EDTs that become ready wait for ocrgpu_resolve / ocrgpu_step.
 * For a LATCH, slot is OCR_EVENT_LATCH_DECR_SLOT (0), OCR_EVENT_LATCH_INCR_SLOT (1), or OCRGPU_LATCH_DELTA(k): the net
 * of k INCRs (k > 0) or -k DECRs the caller made in a row, applied as one op (the latch fires if it leaves 0; k = 0
 * is an INCR followed by a DECR).  The satisfies of one call are applied concurrently: a caller whose ops on one
 * latch must keep their order (INCR before DECR) passes their net.                                                */
#define OCRGPU_LATCH_DELTA_BIAS 0x40000000u
#define OCRGPU_LATCH_DELTA(k) ((uint32_t)(OCRGPU_LATCH_DELTA_BIAS + (int32_t)(k)))
int ocrgpu_event_satisfy_slot(ocrgpu_engine_t *e, uint64_t n, const uint64_t *evt, const uint64_t *db,
                              const uint32_t *slot, int where);
/* ocrEventDestroy (ocr-edt.c:101) */
int ocrgpu_event_destroy(ocrgpu_engine_t *e, uint64_t n, const uint64_t *evt, int where);

/* ---- capacity is what is alive, not what was ever created ------------------------------------------------------
 * The reference frees an EDT when its body has returned (destructTaskHc, hc-task.c:721-790), an event when it is destroyed
 * (destructEventHc, hc-event.c:185-242: ONCE/LATCH/COUNTED events by their own satisfy) and a datablock on ocrDbDestroy
 * (regular-datablock.c:388-470), each time releasing the guid (counted-map-guid.c:540-572 -> hashtable.c:556-575).  Here
 * objects are rows of fixed-size tables, so the same moments put the row on a free stack that the create calls serve
 * first; the guid leaves the table, and its slot there is reused by later puts.                                          */
/* ocrDbDestroy: the guids stop resolving; explicit datablocks give their index back at once, the output datablock of a HASH
 * EDT marks its EDT's line as free to go.  A datablock some EDT is still to read must not be destroyed (as in the
 * reference: use after free).  Unknown guid / not a datablock: err[4], OCRGPU_EFAULT.                                   */
int ocrgpu_db_destroy(ocrgpu_engine_t *e, uint64_t n, const uint64_t *guids, int where);
typedef struct {
    uint64_t edts_freed, events_freed;  /* rows taken back by this call                                                  */
    uint64_t edts_held;                 /* EDTs that fired but cannot go yet: output event not fired, output datablock of
                                           a HASH body not destroyed, or a frontier table still in use                   */
    uint64_t edges_kept;                /* registrations still waiting on a live event; max_edges - edges_kept are free  */
    uint64_t free_edt, free_evt, free_db;    /* heights of the free stacks after the call                                */
    uint64_t fresh_edt, fresh_evt, fresh_db; /* rows never used so far: a create of n objects succeeds iff n <= free + fresh */
    uint64_t slots_used;                /* overflow pre-slot words in use (start over once no EDT with depc > 6 is left)  */
    uint64_t ready_kept;                /* entries the ready log starts over with                                         */
} ocrgpu_reclaim_t;
/* Take back the rows of every EDT that has fired and is done, and of every destroyed event; drop the registrations of
 * events that have fired; start the ready log and the level numbering over (ocrgpu_read_edts / ocrgpu_read_fired results
 * from before the call are gone, and so is the snapshot).  "Done": built-in bodies have run; for OCRBODY_HOST the caller
 * promises that every EDT ocrgpu_pop_ready has handed out so far has returned (EDTs that fired but were not handed out
 * yet stay in the ready log).  An EDT with OCRBODY_F_OUTEVT goes once that event has fired, one with a HASH body once its
 * output datablock was destroyed.  Call between ocrgpu_step / ocrgpu_resolve calls; pending satisfy records are applied
 * first.  Not reclaimed: CHANNEL rings of destroyed CHANNEL events, remote-destination records (max_remote_dsts).      */
int ocrgpu_reclaim(ocrgpu_engine_t *e, ocrgpu_reclaim_t *out);
/* OCRGPU_F_LOCKABLE_DB with host-run bodies (OCRBODY_HOST): the EDTs named have run -- release every datablock they
 * acquired (taskEpilogue's release loop, hc-task.c:1723-1751 -> lockableRelease lockable-datablock.c:330); EDTs the
 * datablocks are handed to become ready in the next ocrgpu_step / ocrgpu_resolve.  Built-in bodies release by themselves. */
int ocrgpu_edt_release_dbs(ocrgpu_engine_t *e, uint64_t n, const uint64_t *edt, int where);

/* ---- the worker loop (replaces SCHED_NOTIFY(EDT_READY) -> deque push -> SCHED_GET_WORK ->
 *      taskExecute -> taskEpilogue, common-scheduler.c:234-313, hc-scheduler-heuristic.c:148-333,
 *      deque.c:190-265, hc-task.c:1707-2181) for EDTs whose bodies are engine built-ins ---------- */
/* fire ready EDTs level by level until nothing is ready and no satisfy is in flight */
int ocrgpu_resolve(ocrgpu_engine_t *e, uint64_t max_levels, ocrgpu_stats_t *stats_out);
/* fire at most `fire_levels` levels (0 = only close pending satisfies), then stop.  OCRGPU_STEP_FORCE_LEVEL:
 * when nothing is ready here still count one (empty) level, so partitions stepping in lockstep keep the same
 * level numbering.  *has_ready_out = 1 when EDTs are waiting to fire; *pending_records_out = queued satisfies. */
#define OCRGPU_STEP_FORCE_LEVEL 1u
/* before anything else: if the last level recorded is a forced one in which nothing fired, forget it (a lockstep
 * driver that learns only afterwards that every partition was idle uses this to undo its final probe level) */
#define OCRGPU_STEP_DROP_EMPTY_LEVEL 2u
/* OCRGPU_STEP_WRITE_HEADERS: when the step is over, write into the LAST record slot of every destination region of the
 * send buffer the 32-byte header {u64 records queued for that destination, u64 this partition's activity in the step
 * (ready-log entries fired + records sent), 0, 0}: an equal-split all-to-all of the whole buffer then tells every
 * receiver how many records to apply and whether anybody is still working -- one collective per level, no host
 * round trip to build the header.  The caller keeps records_per_dest - 1 slots for records. */
#define OCRGPU_STEP_WRITE_HEADERS 4u
int ocrgpu_step(ocrgpu_engine_t *e, uint32_t fire_levels, uint32_t flags, uint64_t *has_ready_out,
                uint64_t *pending_records_out, uint32_t *send_counts_out /* host [n_locations] or NULL */);

/* facts about the last ocrgpu_step / ocrgpu_resolve, no device access: out4[0] = ready-log entries of the level it
 * fired (0 = it fired nothing), [1] = current level, [2] = levels recorded, [3] = satisfies still queued */
int ocrgpu_step_info(ocrgpu_engine_t *e, uint64_t *out4);

/* EDTs whose body is host code (OCRBODY_HOST): GET_WORK + the depv hand-over of taskExecute
 * (hcSchedulerHeuristicGetEdt hc-scheduler-heuristic.c:148-199, workpile pop ocr-workpile.h:39-84, hc-task.c:1864-1916).
 * Returns the HOST EDTs that fired since the previous call (any order): param0_out[i] = paramv[0] (the caller's task
 * handle), depc_out[i], and depv of EDT i at dep_*_out[dep_first_out[i] .. +depc_out[i]): guid = the `tag` and value =
 * the `value` the datablock was created with (a shim stores guid and host pointer there), both 0 for NULL_GUID.
 * The engine does not run an epilogue for such an EDT: the caller satisfies its output event
 * (ocrgpu_event_satisfy_slot) when the body returns (taskEpilogue, hc-task.c:1788-1831).
 * When more is ready than fits, returns OCRGPU_ENOMEM with the needed sizes in *n_edts_out / *n_deps_out and pops
 * nothing.  All arrays are HOST memory. */
int ocrgpu_pop_ready(ocrgpu_engine_t *e, uint64_t max_edts, uint64_t max_deps, uint64_t *n_edts_out, uint64_t *n_deps_out,
                     uint64_t *param0_out, uint32_t *depc_out, uint32_t *dep_first_out, uint64_t *dep_guid_out,
                     uint64_t *dep_value_out);

/* ocrgpu_pop_ready for callers that cannot allocate.  No counterpart in the reference: it is for worker loops that run
 * where hipMalloc is not allowed (between HIP graph captures, tight serving loops, harnesses that forbid allocation).
 * Same arguments, results and errors as ocrgpu_pop_ready, but the counters and the staging of the five output arrays live
 * in a device array of FIXED size that is part of the library (2.5 MiB per GPU, held until the process ends) instead of the
 * engine's growable staging arena: the call allocates and frees nothing.  Limits: max_edts <= 32768 and
 * max_deps <= 131072, a larger request is refused with OCRGPU_EINVAL (pop in several calls, or use ocrgpu_pop_ready).
 * Every engine on one GPU shares that GPU's array: a call that overlaps another one on the same GPU (two threads, two engines)
 * is refused with OCRGPU_EINVAL. */
int ocrgpu_pop_ready_noalloc(ocrgpu_engine_t *e, uint64_t max_edts, uint64_t max_deps, uint64_t *n_edts_out, uint64_t *n_deps_out,
                             uint64_t *param0_out, uint32_t *depc_out, uint32_t *dep_first_out, uint64_t *dep_guid_out,
                             uint64_t *dep_value_out);

/* ---- cross-partition satisfies (replace ocrCommPlatformFcts_t.sendMessage/pollMessage for PD_MSG_DEP_SATISFY,
 *      ocr/src/inc/ocr-comm-platform.h:55-199, mpi-comm-platform.c:1055, hc-dist-policy.c:2208).  The engine fills
 *      per-destination send buffers with 32-byte records; the caller moves them (RCCL all-to-all over xGMI) and
 *      hands what it received to ocrgpu_exchange_apply.                                                 ---------- */
typedef struct { uint64_t guid; uint32_t slot, flags; uint64_t value, tag; } ocrgpu_sendrec_t;   /* 32 B */
/* send_dev_buf: caller-owned DEVICE buffer of n_locations * records_per_dest records (region r = destination r), or NULL:
 * the engine allocates the regions itself (freed by ocrgpu_fini).
 * After this call API batches no longer drain by themselves: drive the engine with ocrgpu_step.               */
int ocrgpu_exchange_config(ocrgpu_engine_t *e, uint32_t n_locations, void *send_dev_buf, uint64_t records_per_dest,
                           uint64_t max_remote_dsts);
int ocrgpu_exchange_counts(ocrgpu_engine_t *e, uint32_t *counts_out /* host, [n_locations] */);
int ocrgpu_exchange_reset(ocrgpu_engine_t *e);
int ocrgpu_exchange_apply(ocrgpu_engine_t *e, const void *records, uint64_t n, int where);
/* the receive side of an EQUAL-SPLIT all-to-all: records_dev (device) holds n_src regions of stride_records records,
 * region s carrying counts[s] (host array) valid records from partition s */
int ocrgpu_exchange_apply_strided(ocrgpu_engine_t *e, const void *records_dev, uint32_t n_src, uint64_t stride_records,
                                  const uint32_t *counts);

/* ---- the comm-platform itself, over RCCL (replaces ocrCommPlatformFcts_t switchRunlevel / sendMessage / pollMessage of
 *      ocr/src/comm-platform/mpi/mpi-comm-platform.c: MPI_Init :54 -> ocrgpu_comm_init, MPI_Isend of one marshalled
 *      ocrPolicyMsg_t per satisfy :1055 + MPI_Iprobe / MPI_Irecv :287,620-642 -> ONE equal-split all-to-all of the send regions
 *      per level, MPI_Finalize :60 -> ocrgpu_comm_fini; interface ocr/src/inc/ocr-comm-platform.h:55-199).  librccl is opened
 *      with dlopen by the first of these calls; libocrgpu does not link it. -------------------------------------------------- */
/* rank 0 makes the 128-byte id (ncclGetUniqueId); the caller hands it to every rank by whatever channel it has */
int ocrgpu_comm_get_unique_id(void *id128_out);
/* one communicator rank per engine: rank == the engine's location, nranks == n_locations of ocrgpu_exchange_config (which comes
 * first).  Collective: every rank calls it. */
int ocrgpu_comm_init(ocrgpu_engine_t *e, const void *id128, int rank, int nranks);
int ocrgpu_comm_fini(ocrgpu_engine_t *e);
/* the transport alone (grouped ncclSend / ncclRecv on the engine's stream): bytes_per_peer bytes from region p of send_dev to
 * rank p, region p of recv_dev from rank p.  Asynchronous: ocrgpu_sync() or a later call on the engine waits for it. */
int ocrgpu_comm_alltoall(ocrgpu_engine_t *e, const void *send_dev, void *recv_dev, uint64_t bytes_per_peer);
/* ---- the comm-platform within ONE node, without a collective: peer windows.  (In place of the same mpi-comm-platform.c calls as
 *      above -- sendMessage :1055 becomes the sender's kernel STORING its records into the receiver's memory over xGMI, pollMessage
 *      :287,620-642 becomes the receiver's kernel waiting for one arrival word per source.)  Every partition owns a window of
 *      fine-grained device memory; ocrgpu_peer_export makes it and fills a 128-byte handle; the caller hands all handles to all
 *      partitions (any channel: the bytes are plain data); ocrgpu_peer_connect maps the others' windows (hipIpcOpenMemHandle for
 *      another process, the pointer itself inside one process).  From then on ocrgpu_exchange / ocrgpu_resolve_exchange (and
 *      their _local forms) move a level's records this way: only the records there are, no RCCL kernel, the wait on the device.
 *      A source that does not arrive within OCRGPU_PEER_TIMEOUT_MS (default 2000) fails the call (OCRGPU_EHIP).  The round may
 *      then be PARTLY applied (the regions of the sources that had arrived went in), so ocrgpu_step / ocrgpu_resolve / ocrgpu_exchange /
 *      ocrgpu_resolve_exchange refuse from then on (OCRGPU_EHIP) until ocrgpu_restore puts a snapshot back, or the caller joins fresh
 *      windows (ocrgpu_peer_fini + export + connect) and thereby vouches for the state.
 *      All partitions must make the same sequence of exchange calls (as with the communicator).
 *      handles = n_locations x OCRGPU_PEER_HANDLE_BYTES, in location order, this partition's own included. ------------------- */
#define OCRGPU_PEER_HANDLE_BYTES 128
int ocrgpu_peer_export(ocrgpu_engine_t *e, void *handle_out);
int ocrgpu_peer_connect(ocrgpu_engine_t *e, const void *handles);
int ocrgpu_peer_fini(ocrgpu_engine_t *e);
/* One exchange round after ocrgpu_step: headers into the last record slot of every send region {records for that destination,
 * activity = ready-log entries fired + records sent, records sent, satisfies still queued}, the all-to-all of the regions, the
 * apply of what arrived (counts read from the headers ON THE DEVICE), the send counters cleared -- all on the engine's stream,
 * one synchronisation at the end.  Outputs (any may be NULL): records this partition received, and the sums over ALL
 * partitions of activity / records sent / satisfies queued, which is what a lockstep driver needs to decide what comes next.
 * Needs records_per_dest - 1 >= the records one step queues for one destination (a fuller region is an err[5] overflow). */
int ocrgpu_exchange(ocrgpu_engine_t *e, uint64_t *records_in_out, uint64_t *activity_out, uint64_t *sent_out, uint64_t *queued_out);
/* the same round for the n partitions of ONE process on one GPU (locations 0..n-1 in order): device-to-device copies in place
 * of RCCL; how the single-GPU tests run the protocol */
int ocrgpu_exchange_local(ocrgpu_engine_t **engines, uint32_t n, uint64_t *activity_out, uint64_t *sent_out, uint64_t *queued_out);
/* The worker loop across partitions (hc-dist's role for this path, hc-dist-policy.c:1027-1031,2208): settle what the main
 * program sent, then per level {ocrgpu_step(1 level, forced), exchange, settle} until no partition fired or sent anything.  Levels
 * are closed everywhere before the next one fires: the record equals the single-partition one.  Collective over the
 * communicator / the windows.  OCRGPU_RX_NO_CASCADE: no satisfy that arrives from another partition ever sends one on (no event
 * -> remote waiter chains) and what it sets off locally is done in ONE launch, so one exchange per level settles it.
 * One host synchronisation per level; over peer windows none: rounds are queued ahead of the host in batches (OCRGPU_RX_AHEAD,
 * default 8; OCRGPU_RX_NO_AHEAD=1 keeps the host-driven loop) and what a round is -- next level, settle, rest of a step that ran
 * out of launches (OCRGPU_RX_LAUNCHES), idle after the end -- is decided on the device from the exchanged header sums, alike on
 * every partition.  A NO_CASCADE run in which a partition falls a level behind the others is stopped with OCRGPU_EINVAL rather
 * than mislabelled (the host-driven loop catches such a step up with polls instead).
 * So is a NO_CASCADE run in which something that ARRIVED set off records for other partitions (an event whose waiters live elsewhere:
 * the DAG is outside the flag's class, and the records would reach them a round late).
 * OCRGPU_DIST_TIMING=1 in the environment: location 0 prints where a level's wall time went (host-driven loop). */
#define OCRGPU_RX_NO_CASCADE 1u
/* OCRGPU_RX_TIME_STEPS (engines made with OCRGPU_F_TIMING): HIP events around every level's k_step launches, summed into
 * ocrgpu_stats_t.step_kernel_ns.  Two event records per level: for a profiling pass, not for the pass whose wall time is reported. */
#define OCRGPU_RX_TIME_STEPS 2u
int ocrgpu_resolve_exchange(ocrgpu_engine_t *e, uint32_t flags, uint64_t *levels_out);
int ocrgpu_resolve_exchange_local(ocrgpu_engine_t **engines, uint32_t n, uint32_t flags, uint64_t *levels_out);

/* ---- workpile (replaces ocrWorkpileFcts_t.push / pop, ocr/src/inc/ocr-workpile.h:39-84; hcWorkpilePush / hcWorkpilePop
 *      ocr/src/workpile/hc/hc-workpile.c:103-150 on the deque of ocr/src/utils/deque.c:190-265): a deque of guids in HBM.
 *      PUSH = at the tail, PUSHBACK = at the head (PUSH_WORKPUSHTYPE / PUSH_WORKPUSHBACKTYPE); POP = from the tail, newest first,
 *      STEAL = from the head, oldest first (POP_WORKPOPTYPE / STEAL_WORKPOPTYPE).  *n_out = 0 when it is empty (the reference
 *      returns NULL_GUID); a full workpile is OCRGPU_ENOMEM (the reference asserts, deque.c:194).  Capacity max_edt + 1024.
 *      (EDTs the ENGINE makes ready are handed out by ocrgpu_pop_ready; this pair serves schedulers that keep their own
 *      work-list of guids -- the LEGACY HC scheduler, hc-scheduler.c:207-334 -- and a heuristic that wants to re-queue.) ------ */
enum { OCRGPU_WP_PUSH = 0, OCRGPU_WP_PUSHBACK = 1 };
enum { OCRGPU_WP_POP = 0, OCRGPU_WP_STEAL = 1 };
int ocrgpu_workpile_push(ocrgpu_engine_t *e, uint32_t type, uint64_t n, const uint64_t *guids, int where);
int ocrgpu_workpile_pop(ocrgpu_engine_t *e, uint32_t type, uint64_t max_n, uint64_t *guids_out, uint64_t *n_out, int where);

/* snapshot / restore of all mutable dependence state (benchmark loops re-arm a built DAG) */
int ocrgpu_snapshot(ocrgpu_engine_t *e);
int ocrgpu_restore(ocrgpu_engine_t *e);
int ocrgpu_sync(ocrgpu_engine_t *e);

/* ---- results (by engine index = creation order; use ocrgpu_guid_get to go from a guid) -------- */
/* level_out[i] = level EDT first+i fired at (-1 never), val_out[i] = its HASH fold */
int ocrgpu_read_edts(ocrgpu_engine_t *e, uint64_t first, uint64_t n, int32_t *level_out, uint64_t *val_out, int where);
/* The same record folded on the device: *digest_out = sum modulo 2^64 over EDTs [first, first+n) of
 * ocrgpu_digest_term(paramv[0], level, value) -- order-independent, so partitions of one DAG (whose EDTs carry a
 * partition-independent paramv[0]) add their digests up and compare the total with a single-partition resolve of the same
 * DAG.  What the comparison stands for: the satisfies that crossed partitions arrived with their payloads
 * (PD_MSG_DEP_SATISFY carries the datablock, hc-dist-policy.c:2208), not merely in the right number.                     */
#ifdef __HIPCC__
__host__ __device__
#endif
static inline uint64_t ocrgpu_digest_term(uint64_t param0, int32_t level, uint64_t value) {
    uint64_t z = param0 * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)level;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z ^= value;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
int ocrgpu_digest_edts(ocrgpu_engine_t *e, uint64_t first, uint64_t n, uint64_t *digest_out);
/* OCREVS(code, aux) of explicit events (which = 0) or of the output events of EDTs first.. (which = 1) */
int ocrgpu_read_events(ocrgpu_engine_t *e, int which, uint64_t first, uint64_t n, uint64_t *state_out, int where);
/* values of explicit DBs (which = 0) or of the output DBs of EDTs first.. (which = 1; 0 if not created) */
int ocrgpu_read_dbs(ocrgpu_engine_t *e, int which, uint64_t first, uint64_t n, uint64_t *value_out, int where);
/* the fired-EDT log: ids_out[0..n_fired) grouped by level, level_start_out[0..n_levels] offsets.
 * Pass NULL arrays to get the sizes only.                                                      */
int ocrgpu_read_fired(ocrgpu_engine_t *e, uint64_t *n_fired, uint64_t *n_levels, uint32_t *ids_out,
                      uint64_t *level_start_out, int where);
int ocrgpu_get_stats(ocrgpu_engine_t *e, ocrgpu_stats_t *out);

#ifdef __cplusplus
}
#endif
#endif
