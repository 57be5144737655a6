/*
 * libocrshim -- host side of libocrgpu for unmodified OCR programs.
 *
 * The program's EDT bodies are ordinary host C.  All dependence bookkeeping -- events, latches, pre-slot counting,
 * the ready set -- lives in the MI355X engine behind include/ocrgpu.h.  The loop below is the reference's worker
 * loop (hc-worker.c:52-216: get work -> taskExecute -> taskEpilogue) turned level-synchronous:
 *
 *     flush the API calls the last level's bodies made (batched, see below) -> ocrgpu_step(one level)
 *     -> ocrgpu_pop_ready: the EDTs that fired, with paramv handle and depv -> run their bodies -> repeat
 *
 * API calls made inside a body are DEFERRED, the execution model the reference implements behind
 * ENABLE_OCR_API_DEFERRABLE (hc-policy.c:1523-1720, hc-task.c:1672-1698): the call returns at once (guids come from
 * blocks reserved with ocrgpu_guid_reserve), is queued, and takes effect when the EDT has returned.  The queue is
 * flushed as batched C-ABI calls.  Program order INSIDE one EDT is kept where it can matter: every op carries a round
 * number; creates (event, DB, EDT) join the create phase of the current round, and the round grows whenever the EDT
 * issues an add-dependence / satisfy / event destroy / DB destroy of an earlier class than its previous such call, or
 * touches the same event twice in a way that does not commute; the flush runs round by round, class by class.  Order BETWEEN EDTs of one level is not defined by OCR
 * (they run concurrently in the reference).
 *
 * Finish EDTs (EDT_PROP_FINISH) follow hc-task.c:1071-1078,1917-1986,1788-1831: a LATCH (counter 1) per finish scope,
 * INCR when a child is created in the scope, DECR in every child's epilogue and in the finish EDT's own; the latch
 * satisfies the finish EDT's output event and DECRs the enclosing scope's latch.
 *
 * Worker threads (hc-worker.c:52-265): with OCRSHIM_WORKERS=N > 1 the bodies of one level (they are independent by
 * construction) are run by N threads; each body's deferred calls go to its thread's own queue (the per-body program-order
 * state is thread-local) and are merged when the level is done; the engine is still driven by one thread at a time (the
 * shared host tables and the few engine calls a body can cause -- guid blocks -- sit behind one mutex).  Default: 1.
 * No CPU fallback: without a GPU ocrgpu_init fails and the program exits with an error.
 */
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ocr.h"
#include "extensions/ocr-labeling.h"
#include "extensions/ocr-affinity.h"
#include "extensions/ocr-hints.h"
#include "ocrgpu.h"
#include "ocr_opstream.h"

/* ---- small helpers ---------------------------------------------------------------------------------------- */
static void die(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt);
    fprintf(stderr, "ocrshim: "); vfprintf(stderr, fmt, ap); fprintf(stderr, "\n");
    va_end(ap);
    exit(70);
}
static void *xrealloc(void *p, size_t n) { void *q = realloc(p, n ? n : 1); if (!q) die("out of host memory"); return q; }
#define GROW(arr, n, cap) do { if ((n) >= (cap)) { (cap) = (cap) ? 2 * (cap) : 1024; (arr) = xrealloc((arr), (cap) * sizeof *(arr)); } } while (0)
/* OCRGPU_EFAULT = the engine counted a user-level error (what the reference ASSERTs on) and completed the call: the program
 * runs on and ends with exit code 73 and the error counters on stderr (shim_fini) */
#define CHK(call) do { int _rc = (call); if (_rc && _rc != OCRGPU_EFAULT) die("%s failed (%d): %s", #call, _rc, ocrgpu_last_error()); } while (0)

static ocrgpu_engine_t *G;
/* one mutex for the host tables bodies share (tasks, templates, guid pools, datablock map, labels, ranges) and for the
 * engine calls made from inside a body; uncontended (and free of any other thread) unless OCRSHIM_WORKERS > 1 */
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
#define LOCK() pthread_mutex_lock(&g_mu)
#define UNLOCK() pthread_mutex_unlock(&g_mu)
#define TLS __thread
static int g_shutdown, g_exit_code, g_verbose;
#include <time.h>
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static double t_sort, t_calls, t_step, t_pop, t_bodies; /* OCRSHIM_VERBOSE: where the wall time went */

/* ---- guid pools: blocks of reserved guids per kind (guidReserve, ocr-guid.h:127-144) ------------------------ */
enum { POOL_BLOCK = 4096 };
typedef struct { u64 g[POOL_BLOCK]; u32 next, n; } Pool;
static u64 init_epoch;            /* bumped by every bring-up (legacy blocks bring the runtime up more than once) */
/* every thread has blocks of its own: the mutex (and the engine) is only needed when one runs out, once per 4096 guids */
static TLS Pool *pools; static TLS u64 pools_epoch;
static u64 pool_take(u32 kind) {
    if (!pools || pools_epoch != init_epoch) { /* (a new bring-up is a new engine: its guids start over) */
        if (!pools) pools = xrealloc(NULL, 32 * sizeof *pools);
        memset(pools, 0, 32 * sizeof *pools); pools_epoch = init_epoch;
    }
    Pool *p = &pools[kind];
    if (p->next == p->n) { LOCK(); CHK(ocrgpu_guid_reserve(G, kind, POOL_BLOCK, p->g, OCRGPU_HOST)); UNLOCK(); p->next = 0; p->n = POOL_BLOCK; }
    return p->g[p->next++];
}

/* ---- templates and tasks (host metadata: what ocrTaskTemplateHc_t / ocrTaskHc_t keep for taskExecute) ------- */
typedef struct { ocrEdt_t fn; u32 paramc, depc; } Template;
typedef struct {
    ocrEdt_t fn; u32 paramc, depc; u64 *paramv;
    u64 guid, outevt;
    u64 scope_latch;   /* latch of the finish scope this EDT was created in (0 = none): DECR in its epilogue */
    u64 own_latch;     /* EDT_PROP_FINISH: the latch of the scope it opens                                   */
    u32 life;          /* 0 = created, 1 = its body has started, 2 = destroyed before it ran: one compare-and-swap decides between a
                          worker that starts the body and a body elsewhere that calls ocrEdtDestroy */
    u64 creator;       /* serial of the body that created it (its check-in is issued when that body returns) */
} Task;
static Template *tpls; static size_t n_tpl, cap_tpl;
/* the task table is a list of fixed chunks that never move: a body reads and fills its entries without a lock while other
 * bodies append theirs (indices come from one atomic counter; a new chunk is entered under the mutex) */
enum { TASK_CHUNK_BITS = 16, TASK_CHUNKS = 1 << 16 };
static Task *task_chunk[TASK_CHUNKS]; static size_t n_task;
static Task *task_at(size_t i) { return &task_chunk[i >> TASK_CHUNK_BITS][i & ((1u << TASK_CHUNK_BITS) - 1)]; }
static size_t task_new(void) {
    const size_t i = __sync_fetch_and_add(&n_task, 1), c = i >> TASK_CHUNK_BITS;
    if (c >= TASK_CHUNKS) die("more than 2^32 EDTs");
    if (!__atomic_load_n(&task_chunk[c], __ATOMIC_ACQUIRE)) {
        LOCK();
        if (!task_chunk[c]) { Task *t = calloc((size_t)1 << TASK_CHUNK_BITS, sizeof *t); if (!t) die("out of host memory"); __atomic_store_n(&task_chunk[c], t, __ATOMIC_RELEASE); }
        UNLOCK();
    }
    return i;
}
#define KIND_TEMPLATE 4u /* OCR_GUID_EDT_TEMPLATE (ocr-guid-kind.h:24); never enters the engine */
static u64 tpl_guid(size_t i) { return ((u64)KIND_TEMPLATE << 49) | (u64)(i + 1); }

/* ---- datablocks: payloads are host memory; guid -> {ptr} open-addressing map for ocrDbDestroy ---------------- */
typedef struct { u64 guid; void *ptr; } DbEnt;
static DbEnt *dbmap; static size_t dbmap_cap, dbmap_n;
static size_t db_hash(u64 g) { return (size_t)((g * 0x9E3779B97F4A7C15ULL) >> 20); }
static void dbmap_put(u64 g, void *p) {
    if (2 * (dbmap_n + 1) > dbmap_cap) {
        size_t oc = dbmap_cap; DbEnt *old = dbmap;
        dbmap_cap = oc ? 2 * oc : 4096; dbmap = calloc(dbmap_cap, sizeof *dbmap); if (!dbmap) die("out of host memory");
        dbmap_n = 0;
        for (size_t i = 0; i < oc; ++i) if (old[i].guid && old[i].guid != ~0ULL) dbmap_put(old[i].guid, old[i].ptr);
        free(old);
    }
    size_t h = db_hash(g) & (dbmap_cap - 1);
    while (dbmap[h].guid && dbmap[h].guid != ~0ULL) h = (h + 1) & (dbmap_cap - 1);
    dbmap[h].guid = g; dbmap[h].ptr = p; ++dbmap_n;
}
static void *dbmap_take(u64 g) {
    if (!dbmap_cap) return NULL;
    size_t h = db_hash(g) & (dbmap_cap - 1);
    while (dbmap[h].guid) {
        if (dbmap[h].guid == g) { void *p = dbmap[h].ptr; dbmap[h].guid = ~0ULL; dbmap[h].ptr = NULL; return p; }
        h = (h + 1) & (dbmap_cap - 1);
    }
    return NULL;
}

/* a payload carries its length 16 bytes in front of it (ocrLegacyBlockProgress reports it) */
static void *db_alloc(u64 len) { char *b = calloc(1, 16 + (len ? len : 1)); if (!b) return NULL; *(u64 *)b = len; return b + 16; }
static void db_free(void *p) { if (p) free((char *)p - 16); }
static u64 db_len(const void *p) { return p ? *(const u64 *)((const char *)p - 16) : 0; }

/* ---- the deferred-op queue ---------------------------------------------------------------------------------- */
enum { C_EVT_CREATE, C_DB_CREATE, C_EDT_CREATE, C_ADD_DEP, C_SATISFY, C_EVT_DESTROY, C_DB_DESTROY, N_CLASS };
typedef struct { u8 cls, mode; u16 pad; u32 slot; u32 round; s32 arg; u64 a, b, c; u64 seq; } Op;
static TLS Op *ops; static TLS size_t n_ops, cap_ops;   /* the running thread's queue */
static u64 flush_epoch;          /* bumped by every flush(): tells whether a labeled create is still in the queue */
/* per-body program-order state */
static TLS u32 cur_round; static TLS int cur_cls;
typedef struct { u64 evt; u32 slot; } Touched;
static TLS Touched *touched; static TLS size_t n_touched, cap_touched;   /* events satisfied in the current round by this body */
static TLS Task *cur_task;            /* EDT whose body is running (NULL while mainEdt runs) */
static TLS u64 cur_scope;             /* latch of the finish scope children are created in   */
static TLS u32 pending_checkins;      /* children created in cur_scope by this body: INCRs issued when the body returns */
static TLS u64 cur_body;          /* serial of the running body (Task.creator of what it creates): thread number << 40 | count */
static TLS u64 body_serial, thread_no; static u64 n_threads_seen;

static void els_reset(void);
static void body_begin(Task *t) {
    els_reset();
    cur_round = 0; cur_cls = -1; n_touched = 0; cur_task = t; pending_checkins = 0;
    if (!thread_no) thread_no = __sync_add_and_fetch(&n_threads_seen, 1);
    cur_body = (thread_no << 40) | ++body_serial;
    cur_scope = t ? (t->own_latch ? t->own_latch : t->scope_latch) : 0;
}
static Op *emit(int cls) {
    /* creates introduce new names and depend on nothing issued before them, so they join the current round's create
     * phase (which runs first) however late they are issued; only the order among add-dependence / satisfy / destroy
     * calls can matter, and an earlier class after a later one opens a new round */
    if (cls > C_EDT_CREATE) {
        if (cls < cur_cls) { ++cur_round; n_touched = 0; }
        cur_cls = cls;
    }
    GROW(ops, n_ops, cap_ops);
    Op *o = &ops[n_ops++];
    memset(o, 0, sizeof *o);
    o->cls = (u8)cls; o->round = cur_round; o->seq = n_ops;
    return o;
}
static void emit_satisfy(u64 evt, u64 db, u32 slot) {
    /* two satisfies of one event by one EDT keep their order unless they commute (same latch slot) */
    const int latch = OCRGPU_GUID_KIND(evt) == OCRGPU_KIND_EVT_LATCH;
    if (!(n_touched && latch && touched[n_touched - 1].evt == evt && touched[n_touched - 1].slot == slot && n_touched == 1))
        for (size_t i = 0; i < n_touched; ++i)
            if (touched[i].evt == evt && !(latch && touched[i].slot == slot)) { ++cur_round; n_touched = 0; cur_cls = -1; break; }
    Op *o = emit(C_SATISFY);
    o->a = evt; o->b = db; o->slot = slot;
    if (n_touched && touched[n_touched - 1].evt == evt && touched[n_touched - 1].slot == slot) return; /* 200k check-ins of one scope: one entry */
    GROW(touched, n_touched, cap_touched);
    touched[n_touched].evt = evt; touched[n_touched].slot = slot; ++n_touched;
}
/* taskEpilogue (hc-task.c:1784-1831): the returned guid goes to the output event through an add-dependence (a DB or
 * NULL_GUID satisfies it, an EVENT is chained to it); the finish scope this EDT opened, or else the one it lives in,
 * is checked out; a finish EDT's latch takes care of its output event.  The check-ins of the children created by
 * this body (hc-task.c:1071-1078) are issued here, before the check-out. */
static void body_end(Task *t, u64 ret) {
    if (t && !t->own_latch && t->outevt) { Op *d = emit(C_ADD_DEP); d->a = ret; d->b = t->outevt; d->slot = 0; d->mode = DB_MODE_RW; }
    const u64 out_latch = !t ? 0 : t->own_latch ? t->own_latch : t->scope_latch;
    const u32 in = pending_checkins;
    pending_checkins = 0;
    if (in && out_latch == cur_scope) { /* k check-ins and the check-out hit the same latch: one op with their net */
        emit_satisfy(cur_scope, t->own_latch ? ret : 0, in == 2 ? OCRLATCH_INCR : OCRGPU_LATCH_DELTA((s32)in - 1));
        return;
    }
    if (in) emit_satisfy(cur_scope, 0, in == 1 ? OCRLATCH_INCR : OCRGPU_LATCH_DELTA((s32)in));
    if (!t) return;
    if (t->own_latch) emit_satisfy(t->own_latch, ret, OCRLATCH_DECR);
    else if (t->scope_latch) emit_satisfy(t->scope_latch, 0, OCRLATCH_DECR);
}

/* the body is about to wait (ocrLegacyBlockProgress): the children it created so far check in now, so that none of
 * them can check out of the scope first; its own check-out still waits for body_end */
static void body_pause(void) {
    const u32 in = pending_checkins;
    pending_checkins = 0;
    if (in && cur_scope) emit_satisfy(cur_scope, 0, in == 1 ? OCRLATCH_INCR : OCRGPU_LATCH_DELTA((s32)in));
}

static int op_cmp(const void *x, const void *y) {
    const Op *a = x, *b = y;
    if (a->round != b->round) return a->round < b->round ? -1 : 1;
    if (a->cls != b->cls) return a->cls < b->cls ? -1 : 1;
    return a->seq < b->seq ? -1 : (a->seq > b->seq ? 1 : 0); /* issue order */
}

/* order the queue by (round, class), issue order within: a counting sort (few distinct keys), qsort as a fallback */
static Op *ops_tmp; static size_t cap_ops_tmp;
static void sort_ops(void) {
    u32 max_round = 0;
    for (size_t i = 0; i < n_ops; ++i) if (ops[i].round > max_round) max_round = ops[i].round;
    const size_t nkeys = ((size_t)max_round + 1) * N_CLASS;
    if (nkeys > (1u << 16)) { qsort(ops, n_ops, sizeof *ops, op_cmp); return; }
    size_t *cnt = calloc(nkeys + 1, sizeof *cnt); if (!cnt) die("out of host memory");
    for (size_t i = 0; i < n_ops; ++i) ++cnt[(size_t)ops[i].round * N_CLASS + ops[i].cls + 1];
    for (size_t k = 0; k < nkeys; ++k) cnt[k + 1] += cnt[k];
    if (cap_ops_tmp < n_ops) { cap_ops_tmp = n_ops + n_ops / 2; ops_tmp = xrealloc(ops_tmp, cap_ops_tmp * sizeof *ops_tmp); }
    for (size_t i = 0; i < n_ops; ++i) ops_tmp[cnt[(size_t)ops[i].round * N_CLASS + ops[i].cls]++] = ops[i];
    memcpy(ops, ops_tmp, n_ops * sizeof *ops);
    free(cnt);
}

static u64 g_max_edges;
static u64 cap_edges_now(void) { return g_max_edges; }
static u64 *bu64[4]; static u32 *bu32[3]; static s32 *bs32; static size_t bcap;
static void batch_room(size_t n) {
    if (n <= bcap) return;
    bcap = n + n / 2 + 64;
    for (int i = 0; i < 4; ++i) bu64[i] = xrealloc(bu64[i], 8 * bcap);
    for (int i = 0; i < 3; ++i) bu32[i] = xrealloc(bu32[i], 4 * bcap);
    bs32 = xrealloc(bs32, 4 * bcap);
}

/* Engine capacity is what is alive (the reference frees an EDT when its body has returned, hc-task.c:721, an event when it
 * is destroyed, hc-event.c:185-242, a datablock on ocrDbDestroy): when the queue holds more creates than the engine has rows
 * left, the rows of finished EDTs and destroyed events are taken back first (ocrgpu_reclaim).  Only between levels and with
 * no body on the stack (a body blocked in ocrLegacyBlockProgress has been handed out but has not returned). */
static u64 room[4];               /* rows / registrations the engine can still take: EDTs, events, datablocks, edges */
static TLS int run_depth;         /* bodies on this thread's stack */
static u64 n_reclaims, peak_live_edt, cap_edt;
static void make_room(void) {
    u64 want[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n_ops; ++i)
        switch (ops[i].cls) {
        case C_EDT_CREATE: ++want[0]; break;
        case C_EVT_CREATE: ++want[1]; break;
        case C_DB_CREATE: ++want[2]; break;
        case C_ADD_DEP: ++want[3]; break;
        default: break;
        }
    int short_of = 0;
    for (int k = 0; k < 4; ++k) if (want[k] > room[k]) short_of = 1;
    if (short_of && run_depth == 0) {
        ocrgpu_reclaim_t r; memset(&r, 0, sizeof r);
        CHK(ocrgpu_reclaim(G, &r));
        ++n_reclaims;
        room[0] = r.free_edt + r.fresh_edt; room[1] = r.free_evt + r.fresh_evt; room[2] = r.free_db + r.fresh_db;
        room[3] = cap_edges_now() - r.edges_kept;
    }
    if (cap_edt - room[0] + want[0] > peak_live_edt) peak_live_edt = cap_edt - room[0] + want[0];
    for (int k = 0; k < 4; ++k) room[k] = room[k] > want[k] ? room[k] - want[k] : 0; /* short after all: the create call reports it */
}

/* apply the queue: one C-ABI call per (round, class) */
static void flush(void) {
    ++flush_epoch;
    if (!n_ops) return;
    make_room();
    double ts0 = now_s();
    sort_ops();
    t_sort += now_s() - ts0; ts0 = now_s();
    size_t i = 0;
    while (i < n_ops) {
        size_t j = i;
        while (j < n_ops && ops[j].round == ops[i].round && ops[j].cls == ops[i].cls) ++j;
        const size_t n = j - i; const Op *o = ops + i;
        batch_room(n);
        switch (ops[i].cls) {
        case C_EVT_CREATE:
            for (size_t k = 0; k < n; ++k) { bu32[0][k] = o[k].slot; bs32[k] = o[k].arg; bu64[0][k] = o[k].a; }
            CHK(ocrgpu_event_create(G, n, bu32[0], bs32, bu64[0], OCRGPU_HOST | OCRGPU_GUIDS_GIVEN));
            break;
        case C_DB_CREATE: /* value = host pointer, tag = guid: what ocrgpu_pop_ready hands back as depv */
            for (size_t k = 0; k < n; ++k) { bu64[0][k] = o[k].b; bu64[1][k] = o[k].a; bu64[2][k] = o[k].a; }
            for (size_t k = 0; k < n; ++k) dbmap_put(o[k].a, (void *)(uintptr_t)o[k].b); /* guid -> payload, for ocrDbDestroy (entered here, by the one thread that flushes) */
            CHK(ocrgpu_db_create(G, n, bu64[0], bu64[1], bu64[2], OCRGPU_HOST | OCRGPU_GUIDS_GIVEN));
            break;
        case C_EDT_CREATE:
            for (size_t k = 0; k < n; ++k) {
                bu32[0][k] = o[k].slot; bu32[1][k] = OCRBODY_HOST | (o[k].b ? OCRBODY_F_OUTEVT : 0u);
                bu64[0][k] = o[k].c; bu64[1][k] = o[k].a; bu64[2][k] = o[k].b;
            }
            CHK(ocrgpu_edt_create(G, n, bu32[0], bu32[1], bu64[0], bu64[1], bu64[2], NULL, OCRGPU_HOST | OCRGPU_GUIDS_GIVEN));
            break;
        case C_ADD_DEP:
            for (size_t k = 0; k < n; ++k) { bu64[0][k] = o[k].a; bu64[1][k] = o[k].b; bu32[0][k] = o[k].slot; bu32[1][k] = o[k].mode; }
            CHK(ocrgpu_add_dependence(G, n, bu64[0], bu64[1], bu32[0], bu32[1], OCRGPU_HOST));
            break;
        case C_SATISFY:
            for (size_t k = 0; k < n; ++k) { bu64[0][k] = o[k].a; bu64[1][k] = o[k].b; bu32[0][k] = o[k].slot; }
            CHK(ocrgpu_event_satisfy_slot(G, n, bu64[0], bu64[1], bu32[0], OCRGPU_HOST));
            break;
        case C_EVT_DESTROY:
            for (size_t k = 0; k < n; ++k) bu64[0][k] = o[k].a;
            CHK(ocrgpu_event_destroy(G, n, bu64[0], OCRGPU_HOST));
            break;
        case C_DB_DESTROY:
            for (size_t k = 0; k < n; ++k) { bu64[0][k] = o[k].a; db_free(dbmap_take(o[k].a)); }
            CHK(ocrgpu_db_destroy(G, n, bu64[0], OCRGPU_HOST));
            room[2] += n;
            break;
        }
        i = j;
    }
    n_ops = 0;
    t_calls += now_s() - ts0;
}

/* ---- labeled guids (ocr/src/guid/labeled/labeled-guid.c, ocr/src/api/extensions/ocr-labeling.c) ------------- */
typedef struct { u64 first, n; u32 kind; ocrGuidMapFunc_t fn; s64 *params; } Range;
static Range *ranges; static size_t n_range, cap_range;
#define KIND_GUIDMAP 15u /* OCR_GUID_GUIDMAP (ocr-guid-kind.h:35); host-side handle */
/* guids at which a labeled object has been created: open-addressing set */
static u64 *labset; static u8 *labalive; static u32 *labround; static u64 *labepoch; static size_t labset_cap, labset_n;
static size_t labset_find(u64 g) { /* slot of g, or of the free slot where it would go */
    size_t h = db_hash(g) & (labset_cap - 1);
    while (labset[h] && labset[h] != g) h = (h + 1) & (labset_cap - 1);
    return h;
}
/* 1 = the label is free (never used, or its object was destroyed) and is now taken by the running body.  0 = it is
 * in use; if its create is still queued (another EDT of this level made it), the running body's later ops are moved
 * to that create's round so that they reach the engine after it. */
static int labset_add_locked(u64 g, int cls);
static int labset_add(u64 g, int cls) { LOCK(); const int r = labset_add_locked(g, cls); UNLOCK(); return r; }
static int labset_add_locked(u64 g, int cls) {
    if (2 * (labset_n + 1) > labset_cap) {
        size_t oc = labset_cap; u64 *old = labset, *olde = labepoch; u8 *olda = labalive; u32 *oldr = labround;
        labset_cap = oc ? 2 * oc : 1024;
        labset = calloc(labset_cap, sizeof *labset); labalive = calloc(labset_cap, 1);
        labround = calloc(labset_cap, sizeof *labround); labepoch = calloc(labset_cap, sizeof *labepoch);
        if (!labset || !labalive || !labround || !labepoch) die("out of host memory");
        for (size_t i = 0; i < oc; ++i) if (old[i]) { size_t h = labset_find(old[i]); labset[h] = old[i]; labalive[h] = olda[i]; labround[h] = oldr[i]; labepoch[h] = olde[i]; }
        free(old); free(olda); free(oldr); free(olde);
    }
    size_t h = labset_find(g);
    if (labset[h] == g && labalive[h]) {
        if (labepoch[h] == flush_epoch && (labround[h] >> 3) >= cur_round) {
            if ((labround[h] >> 3) > cur_round || (int)(labround[h] & 7u) > cur_cls) { cur_round = labround[h] >> 3; cur_cls = (int)(labround[h] & 7u); n_touched = 0; }
        }
        return 0;
    }
    if (labset[h] != g) { labset[h] = g; ++labset_n; }
    labalive[h] = 1; labround[h] = (cur_round << 3) | (u32)cls; labepoch[h] = flush_epoch; /* where emit(cls) will put it */
    return 1;
}
static void labset_release(u64 g) { LOCK(); if (labset_cap) { size_t h = labset_find(g); if (labset[h] == g) labalive[h] = 0; } UNLOCK(); }
/* what a create call does with GUID_PROP_IS_LABELED: *guid is an input; 0 = go on and create at `out` */
static u8 labeled_guid(const ocrGuid_t *guid, u16 props, u32 want_kind, int cls, u64 *out) {
    if (!(props & GUID_PROP_IS_LABELED)) { *out = 0; return 0; }
    u64 g = (u64)guid->guid;
    if (OCRGPU_GUID_KIND(g) != want_kind) return OCR_EINVAL;
    if (!labset_add(g, cls)) return OCR_EGUIDEXISTS; /* GUID_PROP_CHECK / _BLOCK; without them the reference would create twice: a user error */
    *out = g;
    return 0;
}
static u8 range_new(ocrGuid_t *out, u64 n, ocrGuidUserKind kind, ocrGuidMapFunc_t fn, s64 *params, u32 numParams) {
    if (!out || n == 0 || kind == GUID_USER_NONE || kind == GUID_USER_EDT_TEMPLATE || (u32)kind >= 32) return OCR_EINVAL;
    u64 *tmp = xrealloc(NULL, 8 * (size_t)n);
    LOCK();
    CHK(ocrgpu_guid_reserve(G, (u32)kind, n, tmp, OCRGPU_HOST));   /* consecutive counters: first + i */
    GROW(ranges, n_range, cap_range);
    Range *r = &ranges[n_range];
    r->first = tmp[0]; r->n = n; r->kind = (u32)kind; r->fn = fn; r->params = NULL;
    if (numParams && params) { r->params = xrealloc(NULL, 8 * (size_t)numParams); memcpy(r->params, params, 8 * (size_t)numParams); }
    free(tmp);
    out->guid = (intptr_t)(((u64)KIND_GUIDMAP << 49) | (u64)(++n_range));
    UNLOCK();
    return 0;
}
static Range *range_of(ocrGuid_t g) { /* a copy: the table may move while another body creates a range */
    static TLS Range copy;
    u64 v = (u64)g.guid;
    Range *r = NULL;
    LOCK();
    if (OCRGPU_GUID_KIND(v) == KIND_GUIDMAP && (v & 0xFFFFFFFFu) != 0 && (v & 0xFFFFFFFFu) <= n_range) { copy = ranges[(v & 0xFFFFFFFFu) - 1]; r = &copy; }
    UNLOCK();
    return r;
}
u8 ocrGuidRangeCreate(ocrGuid_t *rangeGuid, u64 n, ocrGuidUserKind kind) { return range_new(rangeGuid, n, kind, NULL, NULL, 0); }
u8 ocrGuidMapCreate(ocrGuid_t *mapGuid, u32 numParams, ocrGuidMapFunc_t fn, s64 *params, u64 n, ocrGuidUserKind kind) {
    if (!fn) return OCR_EINVAL;
    return range_new(mapGuid, n, kind, fn, params, numParams);
}
u8 ocrGuidMapDestroy(ocrGuid_t mapGuid) { return range_of(mapGuid) ? 0 : OCR_EINVAL; }
ocrGuid_t addValueToGuid(ocrGuid_t input, u64 value) { ocrGuid_t r = { input.guid + (intptr_t)value }; return r; }
u8 ocrGuidFromIndex(ocrGuid_t *out, ocrGuid_t rangeGuid, u64 idx) {
    Range *r = range_of(rangeGuid);
    if (!out || !r || idx >= r->n) return OCR_EINVAL;
    out->guid = (intptr_t)(r->first + idx);
    return 0;
}
u8 ocrGuidFromLabel(ocrGuid_t *out, ocrGuid_t mapGuid, s64 *tuple) {
    Range *r = range_of(mapGuid);
    if (!out || !r || !r->fn) return OCR_EINVAL;
    ocrGuid_t start = { (intptr_t)r->first };
    *out = r->fn(start, 1, r->params, tuple);          /* stride 1: counters are consecutive */
    u64 g = (u64)out->guid;
    return (g >= r->first && g < r->first + r->n) ? 0 : OCR_EINVAL;
}
u8 ocrGetGuidKind(ocrGuidUserKind *outKind, ocrGuid_t guid) {
    if (!outKind) return OCR_EINVAL;
    u32 k = OCRGPU_GUID_KIND((u64)guid.guid);
    *outKind = (k == 2 || k == 3 || k == 4 || k == 17 || k == 18 || k == 19 || k == 21 || k == 22) ? (ocrGuidUserKind)k : GUID_USER_NONE;
    return 0;
}

/* ---- affinities and hints: one engine = one policy domain = one affinity; hints are recorded, not acted on ---- */
#define KIND_AFFINITY 12u /* OCR_GUID_AFFINITY (ocr-guid-kind.h:32) */
static ocrGuid_t the_affinity(void) { ocrGuid_t g = { (intptr_t)(((u64)KIND_AFFINITY << 49) | 1u) }; return g; }
u8 ocrAffinityCount(ocrAffinityKind kind, u64 *count) { (void)kind; if (!count) return OCR_EINVAL; *count = 1; return 0; }
u8 ocrAffinityGet(ocrAffinityKind kind, u64 *count, ocrGuid_t *aff) {
    (void)kind;
    if (!count || !aff || *count == 0) return OCR_EINVAL;
    aff[0] = the_affinity(); *count = 1;
    return 0;
}
u8 ocrAffinityGetAt(ocrAffinityKind kind, u64 idx, ocrGuid_t *aff) { (void)kind; if (!aff || idx) return OCR_EINVAL; *aff = the_affinity(); return 0; }
u8 ocrAffinityGetCurrent(ocrGuid_t *aff) { if (!aff) return OCR_EINVAL; *aff = the_affinity(); return 0; }
u8 ocrAffinityQuery(ocrGuid_t guid, u64 *count, ocrGuid_t *aff) { (void)guid; if (count) *count = 1; if (aff) aff[0] = the_affinity(); return 0; }
u64 ocrAffinityToHintValue(ocrGuid_t aff) { return (u64)aff.guid; }
u8 ocrHintInit(ocrHint_t *h, ocrHintType_t t) { if (!h) return OCR_EINVAL; memset(h, 0, sizeof *h); h->type = t; return 0; }
u8 ocrSetHintValue(ocrHint_t *h, ocrHintProp_t p, u64 v) {
    if (!h || (u32)p > OCR_HINT_GROUP_PROP_END) return OCR_EINVAL;
    h->args[p] = v; h->propMask |= 1ull << p;
    return 0;
}
u8 ocrUnsetHintValue(ocrHint_t *h, ocrHintProp_t p) { if (!h || (u32)p > OCR_HINT_GROUP_PROP_END) return OCR_EINVAL; h->propMask &= ~(1ull << p); return 0; }
u8 ocrGetHintValue(ocrHint_t *h, ocrHintProp_t p, u64 *v) {
    if (!h || !v || (u32)p > OCR_HINT_GROUP_PROP_END) return OCR_EINVAL;
    if (!(h->propMask & (1ull << p))) return OCR_ENOENT;
    *v = h->args[p];
    return 0;
}
u8 ocrSetHint(ocrGuid_t guid, ocrHint_t *h) { (void)guid; return h ? 0 : OCR_EINVAL; }
u8 ocrGetHint(ocrGuid_t guid, ocrHint_t *h) { (void)guid; if (!h) return OCR_EINVAL; h->propMask = 0; return 0; }

/* ---- runtime interface extension (ocr/inc/extensions/ocr-runtime-itf.h): one worker, EDT-local storage per body ---- */
#include "extensions/ocr-runtime-itf.h"
static TLS ocrGuid_t els_slots[OCRSHIM_ELS_SLOTS];   /* cleared by body_begin */
static void els_reset(void) { memset(els_slots, 0, sizeof els_slots); }
ocrGuid_t ocrElsUserGet(u8 offset) { return offset < OCRSHIM_ELS_SLOTS ? els_slots[offset] : NULL_GUID; }
void ocrElsUserSet(u8 offset, ocrGuid_t data) { if (offset < OCRSHIM_ELS_SLOTS) els_slots[offset] = data; }
void currentEdtUserGet(ocrGuid_t *guid) { if (guid) guid->guid = cur_task ? (intptr_t)cur_task->guid : 0; }
u64 ocrNbWorkers(void) { return 1; }
void ocrCurrentWorkerGuid(ocrGuid_t *guid) { if (guid) guid->guid = (intptr_t)(((u64)OCRGPU_KIND_EDT - 1) << 49 | 1); } /* a fixed, non-null handle */
u8 ocrInformLegacyCodeBlocking(void) { return 0; }

/* ---- the OCR API -------------------------------------------------------------------------------------------- */
static u32 evt_guid_kind(ocrEventTypes_t t) {
    switch (t) {
    case OCR_EVENT_ONCE_T: return OCRGPU_KIND_EVT_ONCE;
    case OCR_EVENT_IDEM_T: return OCRGPU_KIND_EVT_IDEM;
    case OCR_EVENT_STICKY_T: return OCRGPU_KIND_EVT_STICKY;
    case OCR_EVENT_LATCH_T: return OCRGPU_KIND_EVT_LATCH;
    case OCR_EVENT_COUNTED_T: return OCRGPU_KIND_EVT_COUNTED;
    case OCR_EVENT_CHANNEL_T: return OCRGPU_KIND_EVT_CHANNEL;
    default: return 0;
    }
}
u8 ocrEventCreateParams(ocrGuid_t *guid, ocrEventTypes_t type, u16 properties, ocrEventParams_t *params) {
    u32 gk = evt_guid_kind(type);
    if (!guid) return OCR_EINVAL;
    if (type == OCR_EVENT_CHANNEL_T && params && (params->EVENT_CHANNEL.nbSat != 1 || params->EVENT_CHANNEL.nbDeps != 1)) {
        guid->guid = -1; return OCR_ENOTSUP; /* as hc-event.c:1629-1630 */
    }
    if (!gk) { guid->guid = -1; return OCR_ENOTSUP; }
    if (type == OCR_EVENT_COUNTED_T && (!params || params->EVENT_COUNTED.nbDeps == 0 || params->EVENT_COUNTED.nbDeps > 0xFFFF)) { guid->guid = -1; return OCR_EINVAL; }
    u64 lg; u8 lr = labeled_guid(guid, properties, gk, C_EVT_CREATE, &lg);
    if (lr) return lr;
    Op *o = emit(C_EVT_CREATE);
    o->a = lg ? lg : pool_take(gk); o->slot = (u32)type;
    o->arg = (type == OCR_EVENT_LATCH_T && params) ? (s32)params->EVENT_LATCH.counter        /* hc-event.c:1579-1589 */
           : type == OCR_EVENT_COUNTED_T ? (s32)params->EVENT_COUNTED.nbDeps                 /* hc-event.c:1600-1612 */
           : (type == OCR_EVENT_CHANNEL_T && params && params->EVENT_CHANNEL.maxGen != EVENT_CHANNEL_UNBOUNDED) ? (s32)params->EVENT_CHANNEL.maxGen
           : 0;                                                                              /* CHANNEL: 0 = unbounded */
    guid->guid = (intptr_t)o->a;
    return 0;
}
u8 ocrEventCreate(ocrGuid_t *guid, ocrEventTypes_t type, u16 properties) { return ocrEventCreateParams(guid, type, properties, NULL); }
u8 ocrEventDestroy(ocrGuid_t guid) {
    Op *o = emit(C_EVT_DESTROY); o->a = (u64)guid.guid; labset_release(o->a);
    return 0;
}
u8 ocrEventSatisfySlot(ocrGuid_t evt, ocrGuid_t db, u32 slot) {
    if (ocrGuidIsNull(evt)) return 0; /* ocr-edt.c:171 */
    emit_satisfy((u64)evt.guid, (u64)db.guid, slot);
    return 0;
}
u8 ocrEventSatisfy(ocrGuid_t evt, ocrGuid_t db) { return ocrEventSatisfySlot(evt, db, 0); }
u8 ocrAddDependence(ocrGuid_t src, ocrGuid_t dst, u32 slot, ocrDbAccessMode_t mode) {
    const u64 sg = (u64)src.guid;
    const int chan = sg != 0 && sg < (u64)-2 && OCRGPU_GUID_KIND(sg) == OCRGPU_KIND_EVT_CHANNEL;
    if (chan) /* registrations on one CHANNEL are its generations (hc-event.c:2081-2123): one per engine call, in program order */
        for (size_t i = 0; i < n_touched; ++i)
            if (touched[i].evt == sg) { ++cur_round; n_touched = 0; cur_cls = -1; break; }
    Op *o = emit(C_ADD_DEP);
    o->a = sg; o->b = (u64)dst.guid; o->slot = slot; o->mode = (u8)mode;
    if (chan) { GROW(touched, n_touched, cap_touched); touched[n_touched].evt = sg; touched[n_touched].slot = 0xFFFFFFFFu; ++n_touched; }
    return 0;
}

u8 ocrDbCreate(ocrGuid_t *db, void **addr, u64 len, u16 flags, ocrHint_t *hint, ocrInDbAllocator_t allocator) {
    (void)hint; (void)allocator;
    if (!db || !addr) return OCR_EINVAL;
    u64 lg; u8 lr = labeled_guid(db, flags, OCRGPU_KIND_DB, C_DB_CREATE, &lg);
    if (lr) return lr;
    void *p = db_alloc(len);
    if (!p) return OCR_ENOMEM;
    Op *o = emit(C_DB_CREATE);
    o->a = lg ? lg : pool_take(OCRGPU_KIND_DB); o->b = (u64)(uintptr_t)p;
    db->guid = (intptr_t)o->a; *addr = (flags & DB_PROP_NO_ACQUIRE) ? NULL : p; /* ocr-db.h:60-64 */
    return 0;
}
u8 ocrDbRelease(ocrGuid_t db) { (void)db; return 0; } /* payloads are plain host memory: nothing to write back */
u8 ocrDbDestroy(ocrGuid_t db) { Op *o = emit(C_DB_DESTROY); o->a = (u64)db.guid; labset_release(o->a); return 0; }

u8 ocrEdtTemplateCreate_internal(ocrGuid_t *guid, ocrEdt_t fn, u32 paramc, u32 depc, const char *name) {
    (void)name;
    if (!guid || !fn) return OCR_EINVAL;
    LOCK();
    GROW(tpls, n_tpl, cap_tpl);
    tpls[n_tpl].fn = fn; tpls[n_tpl].paramc = paramc; tpls[n_tpl].depc = depc;
    guid->guid = (intptr_t)tpl_guid(n_tpl++);
    UNLOCK();
    return 0;
}
u8 ocrEdtTemplateDestroy(ocrGuid_t guid) { (void)guid; return 0; }
/* ocrEdtDestroy of an EDT that has not run (destructTaskHc, hc-task.c:760-790): it leaves its finish scope */
u8 ocrEdtDestroy(ocrGuid_t guid) {
    LOCK();
    for (size_t i = n_task; i-- > 0;) {
        if (!task_chunk[i >> TASK_CHUNK_BITS]) continue;
        Task *t = task_at(i);
        if (__atomic_load_n(&t->guid, __ATOMIC_ACQUIRE) != (u64)guid.guid) continue; /* (an entry another body is still filling has guid 0: the guid is published last) */
        if (__sync_val_compare_and_swap(&t->life, 0u, 2u) != 0u) { UNLOCK(); return 0; } /* it has started, or was destroyed already */
        const u64 latch = t->scope_latch; const int mine = t->creator == cur_body;
        UNLOCK();
        if (latch) {
            if (mine && pending_checkins) --pending_checkins; /* its check-in has not been issued yet */
            else emit_satisfy(latch, 0, OCRLATCH_DECR);
        }
        return 0;
    }
    UNLOCK();
    return OCR_EINVAL;
}

/* ocrEdtCreate (ocr-edt.c:295-523, hc-task.c:871-1136) */
u8 ocrEdtCreate(ocrGuid_t *guid, ocrGuid_t tplGuid, u32 paramc, u64 *paramv, u32 depc, ocrGuid_t *depv, u16 props,
                ocrHint_t *hint, ocrGuid_t *outputEvent) {
    (void)hint;
    u64 tg = (u64)tplGuid.guid;
    LOCK();
    if (OCRGPU_GUID_KIND(tg) != KIND_TEMPLATE || (tg & 0xFFFFFFFFu) == 0 || (tg & 0xFFFFFFFFu) > n_tpl) { UNLOCK(); return OCR_EINVAL; }
    const Template tpl_copy = tpls[(tg & 0xFFFFFFFFu) - 1], *tp = &tpl_copy; /* (the table may move while another body creates a template) */
    UNLOCK();
    if (paramc == EDT_PARAM_DEF) paramc = tp->paramc;
    if (depc == EDT_PARAM_DEF) depc = tp->depc;
    if (paramc == EDT_PARAM_UNK || depc == EDT_PARAM_UNK) return OCR_EINVAL;            /* ocr-edt.c:330-360 */
    if ((tp->paramc != EDT_PARAM_UNK && tp->paramc != paramc) || (tp->depc != EDT_PARAM_UNK && tp->depc != depc)) return OCR_EINVAL;
    if ((paramc && !paramv) || (!paramc && paramv)) return OCR_EINVAL;                       /* hc-policy.c:846-855 */
    const int finish = (props & EDT_PROP_FINISH) != 0;
    const int want_out = outputEvent != NULL || finish;

    Task nt, *t = &nt; /* filled here, entered into the table (which other bodies may be growing) in one step below */
    memset(t, 0, sizeof *t);
    t->creator = cur_body;
    t->fn = tp->fn; t->paramc = paramc; t->depc = depc;
    if (paramc) { t->paramv = xrealloc(NULL, 8 * (size_t)paramc); memcpy(t->paramv, paramv, 8 * (size_t)paramc); }
    u64 lg = 0;
    if (props & GUID_PROP_IS_LABELED) { /* the EDT is created at *guid; depv is ignored (ocr-edt.c:323-330) */
        u8 lr = guid ? labeled_guid(guid, props, OCRGPU_KIND_EDT, C_EDT_CREATE, &lg) : OCR_EINVAL;
        if (lr) { free(t->paramv); return lr; }
        depv = NULL;
    }
    t->guid = lg ? lg : pool_take(OCRGPU_KIND_EDT);
    /* EDT_PROP_OEVT_VALID (ocr-edt.c:378-386, hc-task.c:940-959): the caller's own event takes the place of the ONCE */
    const int own_evt = (props & EDT_PROP_OEVT_VALID) && outputEvent && outputEvent->guid != 0 && outputEvent->guid != -1 && outputEvent->guid != -2;
    if ((props & EDT_PROP_OEVT_VALID) && !own_evt) { free(t->paramv); return OCR_EINVAL; }
    t->outevt = own_evt ? (u64)outputEvent->guid : want_out ? pool_take(OCRGPU_KIND_EVT_ONCE) : 0;
    t->scope_latch = cur_scope;
    if (finish) { /* the scope's latch starts at 1: the finish EDT itself (hc-task.c:1931-1933) */
        t->own_latch = pool_take(OCRGPU_KIND_EVT_LATCH);
        Op *l = emit(C_EVT_CREATE); l->a = t->own_latch; l->slot = OCREVT_LATCH; l->arg = 1;
    }
    const size_t ti = task_new();
    { Task *slot = task_at(ti); const u64 g = nt.guid; nt.guid = 0; *slot = nt; nt.guid = g;
      __atomic_store_n(&slot->guid, g, __ATOMIC_RELEASE); } /* the guid last: ocrEdtDestroy's scan matches finished entries only */
    Op *o = emit(C_EDT_CREATE);
    o->a = t->guid; o->b = own_evt ? 0 : t->outevt; o->c = (u64)ti; o->slot = depc;
    if (finish) {
        Op *d = emit(C_ADD_DEP); d->a = t->own_latch; d->b = t->outevt; d->slot = 0; d->mode = DB_MODE_RW;
        if (t->scope_latch) { d = emit(C_ADD_DEP); d->a = t->own_latch; d->b = t->scope_latch; d->slot = OCRLATCH_DECR; d->mode = DB_MODE_RW; }
    }
    if (depv) /* hc-policy.c:2262-2320: UNINITIALIZED_GUID entries are left for a later ocrAddDependence */
        for (u32 k = 0; k < depc; ++k)
            if (!ocrGuidIsUninitialized(depv[k])) { Op *d = emit(C_ADD_DEP); d->a = (u64)depv[k].guid; d->b = t->guid; d->slot = k; d->mode = DB_DEFAULT_MODE; }
    if (t->scope_latch) ++pending_checkins; /* hc-task.c:1071-1078 */
    if (guid) guid->guid = (intptr_t)t->guid;
    if (outputEvent) outputEvent->guid = (intptr_t)t->outevt;
    return 0;
}

void ocrShutdown(void) { g_shutdown = 1; }
void ocrAbort(u8 code) { g_shutdown = 1; g_exit_code = code; }
u64 getArgc(void *dbPtr) { return ((u64 *)dbPtr)[0]; }                                  /* ocr.c:94-100 */
char *getArgv(void *dbPtr, u64 count) { return (char *)dbPtr + ((u64 *)dbPtr)[1 + count]; } /* ocr.c:102-110 */
u32 PRINTF(const char *fmt, ...) { va_list ap; va_start(ap, fmt); int n = vprintf(fmt, ap); va_end(ap); fflush(stdout); return n < 0 ? 0u : (u32)n; }
u32 SNPRINTF(char *buf, u32 size, const char *fmt, ...) { va_list ap; va_start(ap, fmt); int n = vsnprintf(buf, size, fmt, ap); va_end(ap); return n < 0 ? 0u : (u32)n; }
void _ocrAssert(bool val, const char *str, const char *file, u32 line) {
    if (val) return;
    fprintf(stderr, "ASSERT: %s @ %s:%u\n", str, file, line); fflush(stderr);
    abort();
}

/* ---- driver ---------------------------------------------------------------------------------------------------- */
static u64 env_u64(const char *name, u64 dflt) { const char *v = getenv(name); return v && *v ? strtoull(v, NULL, 0) : dflt; }

/* packUserArguments (ocr-driver.c:1034, read back by getArgc/getArgv): u64 argc, u64 offset[argc], the strings */
static void *pack_args(int argc, char **argv, u64 *len_out) {
    u64 len = 8 * (u64)(1 + argc);
    for (int i = 0; i < argc; ++i) len += strlen(argv[i]) + 1;
    char *p = calloc(1, len); if (!p) die("out of host memory");
    u64 off = 8 * (u64)(1 + argc);
    ((u64 *)p)[0] = (u64)argc;
    for (int i = 0; i < argc; ++i) { ((u64 *)p)[1 + i] = off; strcpy(p + off, argv[i]); off += strlen(argv[i]) + 1; }
    *len_out = len;
    return p;
}

/* ---- bring-up, the worker loop, tear-down ----------------------------------------------------------------------
 * One run queue for the whole process: ocrgpu_pop_ready fills it a level at a time, run_one() takes the next EDT.  A
 * body that blocks (ocrLegacyBlockProgress) keeps calling run_one() from inside, so the EDTs popped with it still run. */
static u64 *rq_p0, *rq_dg, *rq_dv; static u32 *rq_dc, *rq_df; static size_t rq_cap_e, rq_cap_d; static u64 rq_n, rq_pos;
static u64 n_levels, n_bodies;
/* OCRSHIM_LOCKABLE=1 (the cfg's datablockfactory=Lockable): the engine arbitrates datablock acquire modes
 * (OCRGPU_F_LOCKABLE_DB); the EDTs whose bodies ran since the last refill give their datablocks back before the next step */
static int g_lockable; static u64 *rel_g; static size_t rel_n, rel_cap;

enum { MAX_WORKERS = 64, PAR_GRAB = 32 };
static u64 par_min = 64;          /* OCRSHIM_PAR_MIN: fewest bodies of a level worth waking the workers for */
static int n_workers = 1, workers_up, in_parallel;   /* worker threads: see run_level_parallel() */
static int shim_init(void) {
    g_verbose = (int)env_u64("OCRSHIM_VERBOSE", 0);
    ocrgpu_config_t cfg; memset(&cfg, 0, sizeof cfg);
    cfg.max_edt = (u32)env_u64("OCRSHIM_MAX_EDT", 1u << 20);
    cfg.max_evt = (u32)env_u64("OCRSHIM_MAX_EVT", 1u << 20);
    cfg.max_db = (u32)env_u64("OCRSHIM_MAX_DB", 1u << 20);
    cfg.max_slots = env_u64("OCRSHIM_MAX_SLOTS", 1u << 22);
    cfg.max_edges = env_u64("OCRSHIM_MAX_EDGES", 1u << 23);
    cfg.location = 0;
    cfg.flags = env_u64("OCRSHIM_NANNY", 1) ? OCRGPU_F_NANNY : 0; /* user programs: report double satisfies (hc-task.c:1243) */
    g_lockable = (int)env_u64("OCRSHIM_LOCKABLE", 0); rel_n = 0;
    if (!workers_up) { n_workers = (int)env_u64("OCRSHIM_WORKERS", 1); if (n_workers < 1) n_workers = 1; if (n_workers > MAX_WORKERS) n_workers = MAX_WORKERS; }
    par_min = env_u64("OCRSHIM_PAR_MIN", 64); if (par_min < 2) par_min = 2;
    if (g_lockable) cfg.flags |= OCRGPU_F_LOCKABLE_DB;
    /* a second bring-up in one process (legacy blocks, ocr-legacy.h:74) starts from nothing */
    ++init_epoch;
    g_shutdown = g_exit_code = 0;
    n_task = n_tpl = n_ops = 0; n_touched = 0; n_range = 0; rq_n = rq_pos = 0; n_levels = n_bodies = 0;
    if (dbmap_cap) { memset(dbmap, 0, dbmap_cap * sizeof *dbmap); dbmap_n = 0; }
    if (labset_cap) { memset(labset, 0, labset_cap * sizeof *labset); memset(labalive, 0, labset_cap); labset_n = 0; }
    room[0] = cap_edt = cfg.max_edt; room[1] = cfg.max_evt; room[2] = cfg.max_db; room[3] = g_max_edges = cfg.max_edges;
    run_depth = 0; n_reclaims = 0; peak_live_edt = 0;
    int rc = ocrgpu_init(&G, (int)env_u64("OCRSHIM_DEVICE", 0), &cfg);
    if (rc) { fprintf(stderr, "ocrshim: ocrgpu_init failed (%d): %s\n", rc, ocrgpu_last_error()); return 71; }
    if (!rq_cap_e) {
        rq_cap_e = 1024; rq_cap_d = 4096;
        rq_p0 = xrealloc(NULL, 8 * rq_cap_e); rq_dg = xrealloc(NULL, 8 * rq_cap_d); rq_dv = xrealloc(NULL, 8 * rq_cap_d);
        rq_dc = xrealloc(NULL, 4 * rq_cap_e); rq_df = xrealloc(NULL, 4 * rq_cap_e);
    }
    return 0;
}

/* deferred calls -> engine, one level, what became ready -> run queue.  0 = nothing left that could ever run */
static double t_refilled;   /* when the run queue was last filled: the time until the next refill went into bodies */
static int refill(void) {
    if (t_refilled != 0.0) t_bodies += now_s() - t_refilled;
    for (;;) {
        if (rel_n) { CHK(ocrgpu_edt_release_dbs(G, rel_n, rel_g, OCRGPU_HOST)); rel_n = 0; } /* taskEpilogue's release loop, hc-task.c:1723-1751 */
        flush();
        u64 has_ready = 0, pending = 0;
        double tt = now_s();
        CHK(ocrgpu_step(G, 1, 0, &has_ready, &pending, NULL));
        t_step += now_s() - tt; tt = now_s();
        ++n_levels;
        u64 ne = 0, nd = 0;
        for (;;) {
            int rc = ocrgpu_pop_ready(G, rq_cap_e, rq_cap_d, &ne, &nd, rq_p0, rq_dc, rq_df, rq_dg, rq_dv);
            if (rc == OCRGPU_ENOMEM && (ne > rq_cap_e || nd > rq_cap_d)) {
                if (ne > rq_cap_e) { rq_cap_e = ne + ne / 2; rq_p0 = xrealloc(rq_p0, 8 * rq_cap_e); rq_dc = xrealloc(rq_dc, 4 * rq_cap_e); rq_df = xrealloc(rq_df, 4 * rq_cap_e); }
                if (nd > rq_cap_d) { rq_cap_d = nd + nd / 2; rq_dg = xrealloc(rq_dg, 8 * rq_cap_d); rq_dv = xrealloc(rq_dv, 8 * rq_cap_d); }
                continue;
            }
            if (rc) die("ocrgpu_pop_ready failed (%d): %s", rc, ocrgpu_last_error());
            break;
        }
        t_pop += now_s() - tt;
        rq_n = ne; rq_pos = 0;
        t_refilled = now_s();
        if (ne) return 1;
        if (!has_ready) return 0; /* else: built-in bodies only on this level */
    }
}

static TLS u64 my_bodies;         /* bodies this thread ran since it last added them to n_bodies */
/* taskExecute for entry i of the run queue (hc-task.c:1864-2181) */
static void run_index(u64 i) {
    if (rq_p0[i] >= n_task) die("engine returned an unknown task handle");
    const u64 ti = rq_p0[i];
    if (__sync_val_compare_and_swap(&task_at(ti)->life, 0u, 1u) != 0u) return; /* destroyed before it could run (exactly one of the two wins) */
    Task t = *task_at(ti); /* a copy: the body's context outlives nothing */
    if (rq_dc[i] != t.depc) die("depc mismatch for task %llu", (unsigned long long)ti);
    /* depv is the body's own: a body that blocks runs other bodies underneath it */
    ocrEdtDep_t small[16], *depv = t.depc <= 16 ? small : xrealloc(NULL, t.depc * sizeof *depv);
    for (u32 k = 0; k < t.depc; ++k) {
        depv[k].guid.guid = (intptr_t)rq_dg[rq_df[i] + k]; depv[k].ptr = (void *)(uintptr_t)rq_dv[rq_df[i] + k]; depv[k].mode = DB_DEFAULT_MODE;
    }
    body_begin(&t);
    ++run_depth;
    ocrGuid_t ret = t.fn(t.paramc, t.paramv, t.depc, depv);
    --run_depth;
    body_end(&t, (u64)ret.guid);
    if (g_lockable && t.depc) {
        LOCK();
        if (rel_n == rel_cap) { rel_cap = rel_cap ? 2 * rel_cap : 1024; rel_g = xrealloc(rel_g, 8 * rel_cap); }
        rel_g[rel_n++] = t.guid;
        UNLOCK();
    }
    free(t.paramv); task_at(ti)->paramv = NULL;
    if (depv != small) free(depv);
    ++my_bodies;
}

/* ---- worker threads (hc-worker.c:52-265: every worker loops get-work -> taskExecute) ----------------------------------
 * The EDTs ocrgpu_pop_ready hands out together fired in one level: none of them depends on another, so OCRSHIM_WORKERS
 * threads take them off the run queue with an atomic cursor.  The calls a body makes are deferred into its thread's own
 * queue; when the queue of the level is empty every worker appends what it collected to the driving thread's queue and
 * parks; the driving thread then flushes, steps the engine and pops the next level alone, as before.  A level of fewer
 * than OCRSHIM_PAR_MIN (64) bodies is run by the driving thread without waking anybody (a wake-up costs more than the bodies). */
static pthread_t worker_tid[MAX_WORKERS];
static pthread_barrier_t bar_start, bar_end;
static volatile u64 par_next, par_end;
static Op *wq_ops[MAX_WORKERS]; static size_t wq_n[MAX_WORKERS];   /* what worker w collected in the level just run */
static void level_share(int w) {
    for (;;) { /* PAR_GRAB entries per visit to the shared cursor: short bodies would otherwise spend their time on its cache line */
        if (g_shutdown) break;
        const u64 i0 = __sync_fetch_and_add(&par_next, PAR_GRAB);
        if (i0 >= par_end) break;
        const u64 i1 = i0 + PAR_GRAB < par_end ? i0 + PAR_GRAB : par_end;
        for (u64 i = i0; i < i1 && !g_shutdown; ++i) run_index(i);
    }
    __sync_fetch_and_add(&n_bodies, my_bodies); my_bodies = 0;
    if (w) { wq_ops[w] = ops; wq_n[w] = n_ops; n_ops = 0; } /* the driver takes them over once everybody is parked (its own queue is in use until then) */
}
static void *worker_main(void *arg) {
    const int w = (int)(intptr_t)arg;
    for (;;) {
        pthread_barrier_wait(&bar_start);
        level_share(w);
        pthread_barrier_wait(&bar_end);
    }
    return NULL;
}
static void run_level_parallel(void) {
    if (!workers_up) {
        pthread_barrier_init(&bar_start, NULL, (unsigned)n_workers); pthread_barrier_init(&bar_end, NULL, (unsigned)n_workers);
        for (int w = 1; w < n_workers; ++w) if (pthread_create(&worker_tid[w], NULL, worker_main, (void *)(intptr_t)w)) die("pthread_create failed");
        workers_up = 1;
    }
    par_next = rq_pos; par_end = rq_n; in_parallel = 1;
    pthread_barrier_wait(&bar_start);
    level_share(0);
    pthread_barrier_wait(&bar_end);
    in_parallel = 0;
    for (int w = 1; w < n_workers; ++w) { /* whole bodies, worker by worker, each in the order it ran them */
        if (!wq_n[w]) continue;
        while (n_ops + wq_n[w] > cap_ops) { cap_ops = cap_ops ? 2 * cap_ops : 1024; ops = xrealloc(ops, cap_ops * sizeof *ops); }
        memcpy(ops + n_ops, wq_ops[w], wq_n[w] * sizeof *ops);
        n_ops += wq_n[w]; wq_n[w] = 0;
    }
    rq_pos = rq_n;
}

/* the next ready EDT (or, with worker threads, the rest of the level).  0 = no EDT can run any more */
static int run_one(void) {
    if (rq_pos == rq_n && !refill()) return 0;
    if (n_workers > 1 && run_depth == 0 && rq_n - rq_pos >= par_min) { run_level_parallel(); return 1; }
    run_index(rq_pos++);
    n_bodies += my_bodies; my_bodies = 0;
    return 1;
}

static int shim_fini(void) {
    ocrgpu_stats_t st; memset(&st, 0, sizeof st);
    CHK(ocrgpu_get_stats(G, &st));
    u64 errs = 0; for (int i = 0; i < 8; ++i) errs += st.err[i];
    if (g_verbose || errs)
        fprintf(stderr, "ocrshim: levels=%llu bodies=%llu edt_sat=%llu evt_sat=%llu err=[%llu %llu %llu %llu %llu %llu %llu]\n",
                (unsigned long long)n_levels, (unsigned long long)n_bodies, (unsigned long long)st.edt_sat, (unsigned long long)st.evt_sat,
                (unsigned long long)st.err[0], (unsigned long long)st.err[1], (unsigned long long)st.err[2], (unsigned long long)st.err[3],
                (unsigned long long)st.err[4], (unsigned long long)st.err[5], (unsigned long long)st.err[6]);
    if (g_verbose)
        fprintf(stderr, "ocrshim: reclaims=%llu peak_live_edt<=%llu of max_edt=%llu workers=%d\n", (unsigned long long)n_reclaims,
                (unsigned long long)peak_live_edt, (unsigned long long)cap_edt, n_workers);
    if (g_verbose)
        fprintf(stderr, "ocrshim: seconds: bodies %.3f | queue sort %.3f | batched C-ABI calls %.3f | ocrgpu_step %.3f | ocrgpu_pop_ready %.3f\n",
                t_bodies, t_sort, t_calls, t_step, t_pop);
    if (errs && !g_exit_code) g_exit_code = 73;
    ocrgpu_fini(G); G = NULL;
    return g_exit_code;
}

static void run_until_shutdown(void) {
    while (!g_shutdown)
        if (!run_one()) {
            fprintf(stderr, "ocrshim: no runnable EDT left and ocrShutdown() was not called (%llu EDTs ran)\n", (unsigned long long)n_bodies);
            g_exit_code = 72;
            break;
        }
}

static int strip_ocr_args(int argc, char **argv, char **uargv, const char **ini) {
    int uargc = 0; /* -ocr:cfg <file> is accepted and ignored: capacities come from the environment */
    for (int i = 0; i < argc; ++i) {
        if (!strncmp(argv[i], "-ocr:", 5)) { if (i + 1 < argc && strcmp(argv[i], "-ocr:version")) { ++i; if (ini) *ini = argv[i]; } continue; }
        uargv[uargc++] = argv[i];
    }
    return uargc;
}

/* a legacy-mode program has its own main() and no mainEdt */
extern ocrGuid_t mainEdt(u32 paramc, u64 *paramv, u32 depc, ocrEdtDep_t depv[]) __attribute__((weak));

int ocrshim_run(int argc, char **argv) {
    if (!mainEdt) { fprintf(stderr, "ocrshim: the program defines neither mainEdt nor main\n"); return 70; }
    char **uargv = calloc((size_t)argc + 1, sizeof *uargv);
    int uargc = strip_ocr_args(argc, argv, uargv, NULL);
    int rc = shim_init(); if (rc) return rc;
    /* mainEdt(paramc 0, depc 1: the argument datablock), ocr-driver.c:1034-1060 */
    u64 alen; void *args = pack_args(uargc, uargv, &alen);
    body_begin(NULL);
    ocrGuid_t adb; void *aptr;
    ocrDbCreate(&adb, &aptr, alen, DB_PROP_NONE, NULL_HINT, NO_ALLOC);
    memcpy(aptr, args, alen); free(args);
    ocrEdtDep_t mdep = { adb, aptr, DB_MODE_RW };
    mainEdt(0, NULL, 1, &mdep);
    body_end(NULL, 0);
    run_until_shutdown();
    return shim_fini();
}

/* ---- legacy mode (ocr/inc/extensions/ocr-legacy.h, ocr/src/api/extensions/ocr-legacy.c): the program keeps its own
 *      main(), brings the runtime up, starts EDTs from outside any EDT and waits for their results ---------------- */
#include "extensions/ocr-legacy.h"
void ocrParseArgs(int argc, const char *argv[], ocrConfig_t *cfg) {
    if (!cfg) return;
    cfg->userArgv = calloc((size_t)argc + 1, sizeof *cfg->userArgv); cfg->iniFile = NULL;
    cfg->userArgc = strip_ocr_args(argc, (char **)argv, cfg->userArgv, &cfg->iniFile);
}
void ocrLegacyInit(ocrGuid_t *legacyContext, ocrConfig_t *cfg) {
    (void)cfg;
    int rc = shim_init(); if (rc) exit(rc);
    body_begin(NULL); /* calls made from main() are queued like mainEdt's */
    if (legacyContext) legacyContext->guid = 1;
}
u8 ocrLegacyFinalize(ocrGuid_t legacyContext, bool runUntilShutdown) {
    (void)legacyContext;
    body_end(NULL, 0);
    if (runUntilShutdown) run_until_shutdown(); /* ocr-legacy.c:60-77 */
    return (u8)shim_fini();
}
u8 ocrLegacySpawnOCR(ocrGuid_t *handle, ocrGuid_t finishEdtTemplate, u64 paramc, u64 *paramv, u64 depc, ocrGuid_t *depv, ocrGuid_t legacyContext) {
    (void)legacyContext;
    if (!handle || depc < 1 || !depv) return OCR_EINVAL; /* ocr-legacy.c:96: the first dependence holds the EDT back */
    ocrGuid_t sticky, edt, out, depv0 = depv[0];
    ocrEventCreate(&sticky, OCR_EVENT_STICKY_T, EVT_PROP_TAKES_ARG);
    depv[0] = UNINITIALIZED_GUID;
    u8 r = ocrEdtCreate(&edt, finishEdtTemplate, (u32)paramc, paramv, (u32)depc, depv, EDT_PROP_FINISH, NULL_HINT, &out);
    depv[0] = depv0;
    if (r) return r;
    ocrAddDependence(out, sticky, 0, DB_DEFAULT_MODE);
    ocrAddDependence(depv0, edt, 0, DB_DEFAULT_MODE);
    *handle = sticky;
    return 0;
}
/* the EDT the blocked caller waits behind: it takes what the event carried and wakes the caller */
typedef struct { int done; ocrGuid_t guid; void *ptr; } Watch;
static ocrGuid_t watchEdt(u32 paramc, u64 *paramv, u32 depc, ocrEdtDep_t depv[]) {
    (void)paramc; (void)depc;
    Watch *w = (Watch *)(uintptr_t)paramv[0];
    w->guid = depv[0].guid; w->ptr = depv[0].ptr; w->done = 1;
    return NULL_GUID;
}
u8 ocrLegacyBlockProgress(ocrGuid_t evtHandle, ocrGuid_t *guid, void **result, u64 *size, u16 properties) {
    (void)properties;
    const u64 g = (u64)evtHandle.guid;
    if (g == 0 || g >= (u64)-2 || !(OCRGPU_GUID_KIND(g) & OCRGPU_KIND_EVENT)) return OCR_EINVAL;
    static ocrGuid_t tpl; static int have_tpl; static u64 tpl_epoch;
    if (!have_tpl || tpl_epoch != init_epoch) { ocrEdtTemplateCreate(&tpl, watchEdt, 1, 1); have_tpl = 1; tpl_epoch = init_epoch; }
    if (in_parallel) die("ocrLegacyBlockProgress from inside an EDT needs OCRSHIM_WORKERS=1 (the waiting body would have to drive the engine while other bodies of its level are running)");
    Watch w = { 0, NULL_GUID_INITIALIZER, NULL };
    u64 pv = (u64)(uintptr_t)&w;
    /* the watcher is no child of the caller's finish scope: it must not hold that scope open, nor be counted in it */
    const u64 scope = cur_scope; cur_scope = 0;
    ocrGuid_t we; ocrEdtCreate(&we, tpl, 1, &pv, 1, &evtHandle, EDT_PROP_NONE, NULL_HINT, NULL);
    cur_scope = scope;
    /* what the caller has issued so far takes effect now (its children check in before they can check out); the
     * caller's own context comes back when the wait is over */
    Task *t = cur_task; /* lives in the caller's run_one() frame (NULL: main) */
    ocrGuid_t els_keep[OCRSHIM_ELS_SLOTS]; memcpy(els_keep, els_slots, sizeof els_keep);
    body_pause();
    while (!w.done && !g_shutdown)
        if (!run_one()) { fprintf(stderr, "ocrshim: ocrLegacyBlockProgress: nothing left to run and the event was not satisfied\n"); g_exit_code = 72; g_shutdown = 1; }
    body_begin(t); /* everything issued before the wait has been flushed: program order starts over */
    memcpy(els_slots, els_keep, sizeof els_keep);
    if (guid) *guid = w.guid;
    if (result) *result = w.ptr;
    if (size) *size = db_len(w.ptr);
    return w.done ? 0 : OCR_ECANCELED;
}
ocrGuid_t ocrWait(ocrGuid_t outputEvent) { /* ocr-legacy.c:115-123 */
    ocrGuid_t out;
    return ocrLegacyBlockProgress(outputEvent, &out, NULL, NULL, LEGACY_PROP_NONE) == 0 ? out : (ocrGuid_t)ERROR_GUID_INITIALIZER;
}

/* libocr provides main() and calls the program's mainEdt (ocr-driver.c:1000-1060); so does the shim */
__attribute__((weak)) int main(int argc, char **argv) { return ocrshim_run(argc, argv); }
