/*
 * engine_ref.c -- TEST INFRASTRUCTURE: the CPU restatement ("oracle port") of the reference's
 * dependence-resolution path.  Single-threaded plain C.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (ocr_amd/csrc) never does.
 *
 * PARITY PINNED: tests/test_oracle_golden.py checks this file against result records produced
 * by the reference runtime itself (oracle/ocr_replay.c linked to oracle/_ref/libocr_x86_*.a),
 * committed under tests/golden/, and -- when /root/reference is present -- re-runs the
 * reference live on the same streams.
 *
 * It interprets an op stream (include/ocr_opstream.h) level-synchronously:
 *   main ops run in order at level -1; every EDT that becomes ready while level k runs
 *   fires at level k+1 (level = 1 + max level of whatever satisfied its pre-slots).
 * Semantics restated, with the reference line each follows:
 *   ONCE   satisfy: fan out to all waiters, then self-destruct, also with zero waiters
 *          (satisfyEventHcOnce, ocr/src/event/hc/hc-event.c:437-465); registering a waiter
 *          after that is a user error (registerWaiterEventHc :1078-1083).
 *   LATCH  counter += (slot==DECR ? -1 : +1); fires iff the new value == 0, INCR included,
 *          forwarding the triggering payload, then self-destructs
 *          (satisfyEventHcLatch :812-868,885); initial counter from params (:1579-1589).
 *   STICKY/IDEM store the payload, fan out, stay alive; a waiter registered later is
 *          satisfied at once with the stored payload (registerWaiterEventHcPersist
 *          :1143-1155); 2nd satisfy: IDEM ignores (:601), STICKY is a user error (:577,624).
 *   EDT    slotSatisfiedCount++; depv[slot].guid = payload, forced NULL when the dependence
 *          mode is DB_MODE_NULL; fires when count == depc (satisfyTaskHc,
 *          ocr/src/task/hc/hc-task.c:1253-1266); depc==0 fires at creation (:1129-1134);
 *          a slot satisfied twice is an error (:1243).
 *   ocrAddDependence classification (ocr/src/policy-domain/hc/hc-policy.c:2857-3074):
 *          NULL src -> satisfy(dst,slot,NULL); DB src -> satisfy(dst,slot,db) (mode recorded
 *          first for an EDT dst, registerSignalerTaskHc hc-task.c:1373,1411-1437);
 *          event src -> waiter registration on the source event.
 *   Epilogue: script satisfies (the body's own ocrEventSatisfySlot calls) happen before the
 *          output event is satisfied with the returned guid (taskEpilogue hc-task.c:1822-1831).
 *   Lockable datablocks (ocrref_run2 flag OCRREF_F_LOCKABLE = the cfg's datablockfactory=Lockable): an EDT whose
 *          pre-slots are all satisfied acquires its datablocks one by one in ascending guid order, duplicates once,
 *          in the mode its ocrAddDependence named (taskAllDepvSatisfied / iterateDbFrontier hc-task.c:662-705,
 *          484-544; stable sort :459-477), queues on the first busy one (lockableAcquireInternal
 *          lockable-datablock.c:139-263) and is scheduled once it holds them all; it releases them when its body has
 *          run, and the last user hands the datablock on (lockableRelease :330-507).  Level-synchronous reading:
 *          releases belong to the level that fires, acquisitions to the satisfies between two levels.  WHO wins a
 *          contended datablock is a scheduling accident in the reference; this file serves contenders in the order
 *          they got there, so `level` is one valid outcome (tests compare it through invariants, everything else exactly).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#include "ocr_opstream.h"

enum { ST_LIVE = 0, ST_SAT = 1, ST_DESTROYED = 2, ST_ABSENT = 3 };
enum { ERR_DOUBLE_SLOT = 0, ERR_ONCE_LATE_WAITER = 1, ERR_STICKY_RESAT = 2, ERR_DEAD_EVENT = 3, ERR_BAD_OP = 4, N_ERR = 8 };

typedef struct { uint32_t dst, slot; uint8_t dst_is_evt, mode_null; } waiter_t;
/* CHANNEL: w[w_head..n_w) = registered waiters not yet paired, sq[sq_head..sq_n) = satisfies not yet paired (FIFO both) */
typedef struct { uint32_t kind, state; int32_t counter; uint32_t data; uint32_t n_w, cap_w; waiter_t *w;
                 uint32_t w_head, sq_head, sq_n, sq_cap, *sq; } event_t;
typedef struct { uint32_t depc, cnt, frontier; uint64_t slot_base; uint8_t created;
                 uint32_t *order, acq_pos, acq_next; } edtst_t; /* Lockable: pre-slots in acquisition order, position, link in a waiter list */
/* lockable-datablock.h: attributes.modeLock / numUsers + the three LIFO waiter lists (heads: EDT index + 1) */
typedef struct { uint32_t lock, users, ew, itw, ro; } dblock_t;
enum { DBL_NONE = 0, DBL_EW = 1, DBL_ITW = 2 };
/* lazy_evt: a persistent event (index+1) named by ocrAddDependence on which the EDT has NOT registered yet: it does
 * so only when its frontier -- the first pre-slot not yet satisfied -- reaches this slot (hc-task.c:1286-1338,1361-1410) */
typedef struct { uint32_t payload, lazy_evt; uint8_t sat, mode_null, mode; } slot_t;
typedef struct { uint32_t idx, slot, payload; uint8_t is_evt, mode_null; } work_t;

typedef struct {
    const ocrops_hdr_t *h; const ocrops_edt_t *edt; const ocrops_evt_t *evt; const ocrops_op_t *script;
    event_t *ev; edtst_t *ed; slot_t *slots;
    work_t *stack; size_t sp, stack_cap;
    uint32_t *ready_next; size_t n_next, cap_next;
    uint64_t *db_val; uint8_t *db_live;
    uint64_t *tiles;     /* hdr.db_words == 64: the 8x8 tile behind every created datablock (OCRBODY_TILE) */
    uint64_t err[N_ERR], n_edge_sat, n_evt_sat;
    int lockable; dblock_t *dbl; uint32_t *acq; size_t n_acq, cap_acq; /* acq: EDTs whose datablock frontier is to be (re)started */
} eng_t;

static void push_work(eng_t *E, work_t w) {
    if (E->sp == E->stack_cap) { E->stack_cap = E->stack_cap ? 2 * E->stack_cap : 1024; E->stack = realloc(E->stack, E->stack_cap * sizeof(work_t)); }
    E->stack[E->sp++] = w;
}
static void push_ready(eng_t *E, uint32_t e) {
    if (E->n_next == E->cap_next) { E->cap_next = E->cap_next ? 2 * E->cap_next : 1024; E->ready_next = realloc(E->ready_next, E->cap_next * 4); }
    E->ready_next[E->n_next++] = e;
}
static void push_acq(eng_t *E, uint32_t e) {
    if (E->n_acq == E->cap_acq) { E->cap_acq = E->cap_acq ? 2 * E->cap_acq : 1024; E->acq = realloc(E->acq, E->cap_acq * 4); }
    E->acq[E->n_acq++] = e;
}
static uint32_t ev_index(const eng_t *E, uint64_t r) {
    return OCRREF_CLS(r) == OCRREF_EVT ? (uint32_t)OCRREF_IDX(r) : E->h->n_evt + (uint32_t)OCRREF_IDX(r);
}
/* payload encoding: dbid+1, 0 = NULL_GUID */
static uint32_t payload_of(const eng_t *E, uint64_t r, uint32_t self, const slot_t *depv) {
    switch (OCRREF_CLS(r)) {
    case OCRREF_DB: return (uint32_t)OCRREF_IDX(r) + 1;
    case OCRREF_OUTDB: return E->h->n_db + (uint32_t)OCRREF_IDX(r) + 1;
    case OCRREF_DEPV: return depv ? depv[OCRREF_IDX(r)].payload : 0;
    default: (void)self; return 0;
    }
}

static void fan_out(eng_t *E, event_t *v, uint32_t payload) {
    for (uint32_t i = 0; i < v->n_w; ++i) {
        work_t w = { v->w[i].dst, v->w[i].slot, payload, v->w[i].dst_is_evt, v->w[i].mode_null };
        push_work(E, w);
    }
}

static void add_waiter(eng_t *E, event_t *v, waiter_t w);
/* registerOnFrontier -> registerWaiterEventHcPersist / ...Counted (hc-task.c:395-413, hc-event.c:1092-1163,1176-1248):
 * the EDT becomes a waiter of a STICKY / IDEM / COUNTED event, or is satisfied at once when the event already is */
static void register_persistent(eng_t *E, uint32_t evt_idx, uint32_t edt, uint32_t slot, uint8_t mode_null) {
    event_t *v = &E->ev[evt_idx];
    if (v->state == ST_DESTROYED || v->state == ST_ABSENT) { E->err[ERR_ONCE_LATE_WAITER]++; return; }
    if (v->state == ST_SAT) {
        work_t w = { edt, slot, v->data, 0, mode_null };
        if (v->kind == OCREVT_COUNTED && --v->counter == 0) v->state = ST_DESTROYED; /* the last registration destroys (:1225-1236) */
        push_work(E, w);
        return;
    }
    waiter_t w = { edt, slot, 0, mode_null };
    add_waiter(E, v, w);
}

/* drain the satisfy work stack (depth-first, like the nested processMessage calls of
 * commonSatisfyRegNode hc-event.c:335-367; the order is not observable in the record) */
static void drain(eng_t *E) {
    while (E->sp) {
        work_t w = E->stack[--E->sp];
        if (!w.is_evt) { /* satisfyTaskHc */
            edtst_t *t = &E->ed[w.idx];
            slot_t *s = &E->slots[t->slot_base + w.slot];
            E->n_edge_sat++;
            if (s->sat) { E->err[ERR_DOUBLE_SLOT]++; continue; }
            s->sat = 1;
            s->payload = (w.mode_null || s->mode_null) ? 0 : w.payload;
            if (++t->cnt == t->depc) { if (E->lockable) push_acq(E, w.idx); else push_ready(E, w.idx); continue; }
            if (w.slot == t->frontier) { /* advance the frontier over satisfied slots (hc-task.c:1288-1300) ... */
                int cond = 1;
                while (t->frontier != t->depc - 1 && cond) { t->frontier++; cond = E->slots[t->slot_base + t->frontier].sat; }
                slot_t *f = &E->slots[t->slot_base + t->frontier];
                if (f->lazy_evt && !f->sat) { /* ... and register on the persistent event found there (:1310-1338) */
                    uint32_t x = f->lazy_evt - 1; f->lazy_evt = 0;
                    register_persistent(E, x, w.idx, t->frontier, f->mode_null);
                }
            }
            continue;
        }
        event_t *v = &E->ev[w.idx];
        E->n_evt_sat++;
        if (v->state == ST_DESTROYED || v->state == ST_ABSENT) { E->err[ERR_DEAD_EVENT]++; continue; }
        switch (v->kind) {
        case OCREVT_ONCE:
            fan_out(E, v, w.payload); v->state = ST_DESTROYED; break;
        case OCREVT_LATCH:
            v->counter += (w.slot == OCRLATCH_DECR) ? -1 : 1;
            if (v->counter == 0) { fan_out(E, v, w.payload); v->state = ST_DESTROYED; }
            break;
        case OCREVT_IDEM: case OCREVT_STICKY:
            if (v->state == ST_SAT) { if (v->kind == OCREVT_STICKY) E->err[ERR_STICKY_RESAT]++; break; }
            v->data = w.payload; v->state = ST_SAT; fan_out(E, v, w.payload); break;
        case OCREVT_COUNTED: /* satisfyEventHcCounted hc-event.c:537-565: sticky, and gone once nbDeps waiters were served */
            if (v->state == ST_SAT) { E->err[ERR_STICKY_RESAT]++; break; }
            v->data = w.payload; v->state = ST_SAT;
            v->counter -= (int32_t)v->n_w;                 /* nbDeps -= waitersCount (:556) */
            fan_out(E, v, w.payload);
            if (v->counter == 0) v->state = ST_DESTROYED;  /* :557,561 */
            break;
        case OCREVT_CHANNEL: /* satisfyEventHcChannel hc-event.c:2125-2156: the oldest waiter takes it, else it queues */
            if (v->w_head < v->n_w) {
                waiter_t *cw = &v->w[v->w_head++];
                work_t d = { cw->dst, cw->slot, w.payload, cw->dst_is_evt, cw->mode_null };
                push_work(E, d);
            } else {
                if (v->counter > 0 && v->sq_n - v->sq_head >= (uint32_t)v->counter) { E->err[ERR_BAD_OP]++; break; } /* ASSERT :2023 */
                if (v->sq_n == v->sq_cap) { v->sq_cap = v->sq_cap ? 2 * v->sq_cap : 4; v->sq = realloc(v->sq, 4 * (size_t)v->sq_cap); }
                v->sq[v->sq_n++] = w.payload;
            }
            break;
        default: E->err[ERR_BAD_OP]++; break;
        }
    }
}

static void add_waiter(eng_t *E, event_t *v, waiter_t w) {
    if (v->n_w == v->cap_w) { v->cap_w = v->cap_w ? 2 * v->cap_w : 4; v->w = realloc(v->w, v->cap_w * sizeof(waiter_t)); }
    v->w[v->n_w++] = w;
}

static void op_add_dep(eng_t *E, const ocrops_op_t *op) {
    uint32_t sc = OCRREF_CLS(op->a), dc = OCRREF_CLS(op->b);
    int dst_is_evt = (dc == OCRREF_EVT || dc == OCRREF_OUTEVT);
    uint32_t dst = dst_is_evt ? ev_index(E, op->b) : (uint32_t)OCRREF_IDX(op->b);
    uint8_t mode_null = (!dst_is_evt && op->mode == OCRMODE_NULL);
    /* a pre-slot the EDT does not have: ASSERT(slot < depc) in the reference (hc-task.c:1229,1374), a malformed op here */
    if (!dst_is_evt && (dc != OCRREF_EDT || dst >= E->h->n_edt || op->slot >= E->edt[dst].depc)) { E->err[ERR_BAD_OP]++; return; }
    if (sc == OCRREF_NULL || sc == OCRREF_DB || sc == OCRREF_OUTDB) {
        /* convertDepAddToSatisfy hc-policy.c:2857-2902; DB->EDT records the mode first */
        uint8_t mn = (sc == OCRREF_NULL) ? 0 : mode_null; /* NULL src: mode not recorded (signalers[].mode stays -1) */
        if (!dst_is_evt && sc != OCRREF_NULL) E->slots[E->ed[dst].slot_base + op->slot].mode = op->mode;
        work_t w = { dst, op->slot, payload_of(E, op->a, 0, NULL), (uint8_t)dst_is_evt, mn };
        push_work(E, w); drain(E);
        return;
    }
    if (sc != OCRREF_EVT && sc != OCRREF_OUTEVT) { E->err[ERR_BAD_OP]++; return; }
    event_t *v = &E->ev[ev_index(E, op->a)];
    if (!dst_is_evt) { slot_t *ds = &E->slots[E->ed[dst].slot_base + op->slot]; ds->mode_null = mode_null; ds->mode = op->mode; } /* registerSignalerTaskHc :1373 */
    if (!dst_is_evt && (v->kind == OCREVT_STICKY || v->kind == OCREVT_IDEM || v->kind == OCREVT_COUNTED)) {
        /* persistent source, EDT destination: register now only on the frontier slot, else when the frontier gets
         * there (registerSignalerTaskHc :1391-1405) */
        edtst_t *t = &E->ed[dst];
        if (op->slot == t->frontier) { register_persistent(E, ev_index(E, op->a), dst, op->slot, mode_null); drain(E); }
        else E->slots[t->slot_base + op->slot].lazy_evt = ev_index(E, op->a) + 1;
        return;
    }
    if (v->state == ST_DESTROYED || v->state == ST_ABSENT) { E->err[ERR_ONCE_LATE_WAITER]++; return; }
    if (v->kind == OCREVT_CHANNEL) { /* registerWaiterEventHcChannel hc-event.c:2081-2123: takes the oldest queued satisfy, else waits in line */
        if (v->sq_head < v->sq_n) {
            work_t w = { dst, op->slot, v->sq[v->sq_head++], (uint8_t)dst_is_evt, mode_null };
            push_work(E, w); drain(E);
        } else {
            if (v->counter > 0 && v->n_w - v->w_head >= (uint32_t)v->counter) { E->err[ERR_BAD_OP]++; return; } /* ASSERT :2005 */
            waiter_t w = { dst, op->slot, (uint8_t)dst_is_evt, mode_null };
            add_waiter(E, v, w);
        }
        return;
    }
    if (v->state == ST_SAT) { /* persistent, already satisfied: immediate satisfy with stored data */
        work_t w = { dst, op->slot, v->data, (uint8_t)dst_is_evt, mode_null };
        /* registerWaiterEventHcCounted hc-event.c:1211-1238: the late registration is counted; the last one destroys */
        if (v->kind == OCREVT_COUNTED && --v->counter == 0) v->state = ST_DESTROYED;
        push_work(E, w); drain(E);
        return;
    }
    waiter_t w = { dst, op->slot, (uint8_t)dst_is_evt, mode_null };
    add_waiter(E, v, w);
}

static void op_satisfy(eng_t *E, const ocrops_op_t *op, uint32_t self, const slot_t *depv) {
    uint32_t c = OCRREF_CLS(op->a);
    uint32_t payload = payload_of(E, op->b, self, depv);
    if (c == OCRREF_EVT || c == OCRREF_OUTEVT) { work_t w = { ev_index(E, op->a), op->slot, payload, 1, 0 }; push_work(E, w); }
    else if (c == OCRREF_EDT) { work_t w = { (uint32_t)OCRREF_IDX(op->a), op->slot, payload, 0, 0 }; push_work(E, w); }
    else if (c == OCRREF_NULL) return; /* ocrEventSatisfy(NULL_GUID, ..) no-op, ocr-edt.c:171 */
    else { E->err[ERR_BAD_OP]++; return; }
    drain(E);
}

static void op_evt_destroy(eng_t *E, uint64_t r) {
    event_t *v = &E->ev[ev_index(E, r)];
    if (v->state == ST_DESTROYED || v->state == ST_ABSENT) { E->err[ERR_DEAD_EVENT]++; return; }
    v->state = ST_DESTROYED;
}

/* ---- Lockable datablocks ------------------------------------------------------------------------------------ */
/* lockableAcquireInternal (lockable-datablock.c:139-263): 0 = acquired, 1 = OCR_EBUSY (queued, LIFO like the reference's lists) */
static int db_acquire(eng_t *E, uint32_t db, uint32_t mode, uint32_t e) {
    dblock_t *L = &E->dbl[db];
    uint32_t *list = NULL;
    if (mode == OCRMODE_CONST) { if (L->lock) list = &L->ro; }                                    /* :173-190 */
    else if (mode == OCRMODE_EW) { if (L->lock || L->users) list = &L->ew; else L->lock = DBL_EW; } /* :192-212 */
    else if (mode == OCRMODE_RW) {                                                                 /* :214-249 */
        if (L->lock == DBL_ITW) { /* one location: joins the writers */ }
        else if (L->users || L->lock == DBL_EW) list = &L->itw;
        else L->lock = DBL_ITW;
    }                                                                                              /* RO: falls through (:171) */
    if (list) { E->ed[e].acq_next = *list; *list = e + 1; return 1; }
    L->users++;
    return 0;
}
/* iterateDbFrontier (hc-task.c:484-544): go on acquiring from where the EDT stands; all held -> scheduleTask */
static void acq_walk(eng_t *E, uint32_t e) {
    edtst_t *t = &E->ed[e];
    const slot_t *sl = &E->slots[t->slot_base];
    if (!t->order && t->depc) { /* taskAllDepvSatisfied :662-693: ascending guid, stable (sortRegNode :459-477) */
        t->order = malloc(4 * (size_t)t->depc);
        for (uint32_t i = 0; i < t->depc; ++i) {
            uint32_t j = i;
            while (j > 0 && sl[t->order[j - 1]].payload > sl[i].payload) { t->order[j] = t->order[j - 1]; --j; }
            t->order[j] = i;
        }
    }
    while (t->acq_pos < t->depc) {
        const uint32_t k = t->order[t->acq_pos++], p = sl[k].payload; /* pre-incremented like frontierSlot (:491) */
        if (p == 0) continue;                                                   /* NULL_GUID (:492) */
        if (t->acq_pos >= 2 && sl[t->order[t->acq_pos - 2]].payload == p) continue; /* duplicate: acquired once (:495-499) */
        if (db_acquire(E, p - 1, sl[k].mode, e)) return;                        /* OCR_EBUSY: resumed by the release (:529-531) */
    }
    push_ready(E, e);
}
static void acq_drain(eng_t *E) {
    for (size_t i = 0; i < E->n_acq; ++i) acq_walk(E, E->acq[i]); /* acq_walk appends nothing to acq */
    E->n_acq = 0;
}
/* lockableRelease (lockable-datablock.c:330-507); a granted waiter goes on with its frontier (dependenceResolvedTaskHc) */
static void db_release(eng_t *E, uint32_t db) {
    dblock_t *L = &E->dbl[db];
    if (L->users == 0) { E->err[ERR_BAD_OP]++; return; }
    if (--L->users) return;
    uint32_t w = 0;
    if (L->lock) { L->lock = DBL_NONE; if (L->ro) { w = L->ro; L->ro = 0;       /* writer(s) gone: every reader in line (:375-382,459-476) */
            for (; w; w = E->ed[w - 1].acq_next) { L->users++; push_acq(E, w - 1); } return; } }
    if (L->itw) { L->lock = DBL_ITW; w = L->itw; L->itw = 0;                      /* every RW waiter of the location (:390-438) */
        for (; w; w = E->ed[w - 1].acq_next) { L->users++; push_acq(E, w - 1); } }
    else if (L->ew) { w = L->ew; L->ew = E->ed[w - 1].acq_next; L->lock = DBL_EW; L->users = 1; push_acq(E, w - 1); } /* one EW waiter (:393-395,439-458) */
}
static void release_all(eng_t *E, uint32_t e) { /* taskEpilogue's release loop over the acquired depv (hc-task.c:1723-1751) */
    edtst_t *t = &E->ed[e];
    const slot_t *sl = &E->slots[t->slot_base];
    uint32_t prev = 0;
    for (uint32_t i = 0; i < t->depc; ++i) { uint32_t p = sl[t->order[i]].payload; if (p && p != prev) db_release(E, p - 1); if (p) prev = p; }
}

/* taskExecute + taskEpilogue for one ready EDT (hc-task.c:1864-2181, 1707-1862) */
static void fire(eng_t *E, uint32_t e, int32_t lvl, int32_t *level, uint64_t *val) {
    const ocrops_edt_t *d = &E->edt[e];
    const slot_t *depv = &E->slots[E->ed[e].slot_base];
    uint64_t v = ocrops_hash_init(e);
    for (uint32_t i = 0; i < d->depc; ++i) {
        uint32_t p = depv[i].payload;
        uint64_t pv = (p && E->db_live[p - 1]) ? E->db_val[p - 1] : 0;
        v = ocrops_hash_step(v, i, pv, p);
    }
    level[e] = lvl; val[e] = v;
    uint32_t ret = 0;
    if (OCRBODY_KIND(d->body) == OCRBODY_TILE) { /* the last tile on the pre-slots is updated in place with the ones before it */
        uint32_t t[3], n = 0;
        for (uint32_t i = 0; i < d->depc; ++i) {
            uint32_t p = depv[i].payload;
            if (!p) continue;
            if (!E->tiles || p - 1 >= E->h->n_db || n == 3) { E->err[ERR_BAD_OP]++; n = 0; break; }
            t[n++] = p;
        }
        if (n) {
            val[e] = ocrops_tile_update(E->tiles + 64 * (size_t)(t[n - 1] - 1), n >= 2 ? E->tiles + 64 * (size_t)(t[0] - 1) : NULL,
                                        n == 3 ? E->tiles + 64 * (size_t)(t[1] - 1) : NULL);
            ret = t[n - 1];
        } else val[e] = 0;
    }
    switch (OCRBODY_KIND(d->body)) {
    case OCRBODY_HASH: ret = E->h->n_db + e + 1; E->db_val[ret - 1] = v; E->db_live[ret - 1] = 1; break;
    case OCRBODY_FWD0: ret = d->depc ? depv[0].payload : 0; break;
    default: break;
    }
    for (uint32_t k = 0; k < d->script_len; ++k) {
        const ocrops_op_t *op = &E->script[d->script_begin + k];
        if (op->op == OCROP_SATISFY) op_satisfy(E, op, e, depv);
        else if (op->op == OCROP_EVT_DESTROY) op_evt_destroy(E, op->a);
        else E->err[ERR_BAD_OP]++;
    }
    if (d->body & OCRBODY_F_OUTEVT) { work_t w = { E->h->n_evt + e, 0, ret, 1, 0 }; push_work(E, w); drain(E); }
    if (E->lockable && d->depc) release_all(E, e);
}

/* Run a whole stream.  Outputs sized as in include/ocr_opstream.h; stats[16]:
 * [0] levels run, [1] EDTs fired, [2] EDT pre-slot satisfies, [3] event pre-slot satisfies, [8+i] error counters */
#define OCRREF_F_LOCKABLE 1u
int ocrref_run2(const ocrops_hdr_t *h, const ocrops_edt_t *edt, const ocrops_evt_t *evt, const uint64_t *db_init,
                const ocrops_op_t *main_ops, const ocrops_op_t *script,
                int32_t *level, uint64_t *val, uint64_t *evt_state, uint64_t *db_val, uint64_t *stats, uint32_t flags);
int ocrref_run(const ocrops_hdr_t *h, const ocrops_edt_t *edt, const ocrops_evt_t *evt, const uint64_t *db_init,
               const ocrops_op_t *main_ops, const ocrops_op_t *script,
               int32_t *level, uint64_t *val, uint64_t *evt_state, uint64_t *db_val, uint64_t *stats) {
    return ocrref_run2(h, edt, evt, db_init, main_ops, script, level, val, evt_state, db_val, stats, 0u);
}
int ocrref_run2(const ocrops_hdr_t *h, const ocrops_edt_t *edt, const ocrops_evt_t *evt, const uint64_t *db_init,
                const ocrops_op_t *main_ops, const ocrops_op_t *script,
                int32_t *level, uint64_t *val, uint64_t *evt_state, uint64_t *db_val, uint64_t *stats, uint32_t flags) {
    eng_t E; memset(&E, 0, sizeof E);
    E.lockable = (flags & OCRREF_F_LOCKABLE) != 0;
    E.h = h; E.edt = edt; E.evt = evt; E.script = script;
    uint32_t n_evt_total = h->n_evt + h->n_edt, n_db_total = h->n_db + h->n_edt;
    E.ev = calloc(n_evt_total + 1, sizeof(event_t));
    E.ed = calloc(h->n_edt + 1, sizeof(edtst_t));
    uint64_t ns = 0;
    for (uint32_t i = 0; i < h->n_edt; ++i) { E.ed[i].depc = edt[i].depc; E.ed[i].slot_base = ns; ns += edt[i].depc; }
    E.slots = calloc(ns + 1, sizeof(slot_t));
    E.db_val = db_val; memset(db_val, 0, 8 * (size_t)n_db_total);
    E.db_live = calloc(n_db_total + 1, 1);
    if (h->db_words == OCROPS_TILE_WORDS) E.tiles = calloc(64 * (size_t)h->n_db + 1, 8);
    if (E.lockable) E.dbl = calloc(n_db_total + 1, sizeof(dblock_t));
    for (uint32_t x = 0; x < n_evt_total; ++x) E.ev[x].state = ST_ABSENT;
    memset(level, 0xFF, 4 * (size_t)h->n_edt); memset(val, 0, 8 * (size_t)h->n_edt);

    /* OCRREF_PROGRESS=<seconds>: a line on stderr at most that often (a 125M-EDT stream runs for minutes; harnesses with a silence
     * limit want to see it move).  Test infrastructure only, like the rest of this file. */
    const char *pg = getenv("OCRREF_PROGRESS");
    const double pg_every = pg ? atof(pg) : 0.0;
    time_t pg_last = time(NULL);
    for (uint32_t k = 0; k < h->n_main; ++k) {
        if (pg_every > 0 && (k & 0xFFFFF) == 0 && difftime(time(NULL), pg_last) >= pg_every) {
            pg_last = time(NULL); fprintf(stderr, "[engine_ref] main op %u of %u\n", k, h->n_main); fflush(stderr);
        }
        const ocrops_op_t *op = &main_ops[k];
        switch (op->op) {
        case OCROP_EVT_CREATE: { event_t *v = &E.ev[op->a]; v->kind = evt[op->a].kind; v->state = ST_LIVE;
                                  v->counter = (v->kind == OCREVT_LATCH || v->kind == OCREVT_COUNTED || v->kind == OCREVT_CHANNEL) ? evt[op->a].arg : 0; } break; /* CHANNEL: maxGen */
        case OCROP_DB_CREATE: E.db_val[op->a] = db_init[op->a]; E.db_live[op->a] = 1;
                              if (E.tiles) for (uint32_t w = 0; w < 64; ++w) E.tiles[64 * (size_t)op->a + w] = ocrops_tile_init(db_init[op->a], w);
                              break;
        case OCROP_EDT_CREATE: { uint32_t e = (uint32_t)op->a; E.ed[e].created = 1;
                                  if (edt[e].body & OCRBODY_F_OUTEVT) { event_t *v = &E.ev[h->n_evt + e]; v->kind = OCREVT_ONCE; v->state = ST_LIVE; }
                                  if (edt[e].depc == 0) push_ready(&E, e); } break; /* nothing to acquire */
        case OCROP_ADD_DEP: op_add_dep(&E, op); break;
        case OCROP_SATISFY: op_satisfy(&E, op, 0, NULL); break;
        case OCROP_EVT_DESTROY: op_evt_destroy(&E, op->a); break;
        default: E.err[ERR_BAD_OP]++; break;
        }
        acq_drain(&E);
    }
    int32_t lvl = 0; uint64_t fired = 0;
    uint32_t *cur = NULL; size_t n_cur = 0;
    while (E.n_next) {
        free(cur); cur = E.ready_next; n_cur = E.n_next;
        E.ready_next = NULL; E.n_next = E.cap_next = 0;
        for (size_t i = 0; i < n_cur; ++i) {
            if (pg_every > 0 && (i & 0xFFFFF) == 0 && difftime(time(NULL), pg_last) >= pg_every) {
                pg_last = time(NULL); fprintf(stderr, "[engine_ref] level %d: EDT %zu of %zu (%llu fired so far)\n", lvl, i, n_cur, (unsigned long long)fired); fflush(stderr);
            }
            fire(&E, cur[i], lvl, level, val);
        }
        acq_drain(&E); /* the level's releases are in: now the acquisitions they and its satisfies set off */
        fired += n_cur; ++lvl;
    }
    free(cur);
    for (uint32_t x = 0; x < n_evt_total; ++x) {
        event_t *v = &E.ev[x];
        switch (v->state) {
        case ST_ABSENT: evt_state[x] = OCREVS(OCREVS_ABSENT, 0); break;
        case ST_DESTROYED: evt_state[x] = OCREVS(OCREVS_DESTROYED, 0); break;
        case ST_SAT: evt_state[x] = OCREVS(OCREVS_LIVE_SAT, v->data); break;
        default: evt_state[x] = OCREVS(OCREVS_LIVE_UNSAT, v->kind == OCREVT_LATCH ? (uint32_t)v->counter
                                       : v->kind == OCREVT_CHANNEL ? ((v->sq_n - v->sq_head) << 16) | (v->n_w - v->w_head) : 0); break;
        }
        free(v->w); free(v->sq);
    }
    if (E.tiles) { for (uint32_t d = 0; d < h->n_db; ++d) if (E.db_live[d]) db_val[d] = ocrops_tile_checksum(E.tiles + 64 * (size_t)d); free(E.tiles); }
    if (stats) { memset(stats, 0, 16 * 8); stats[0] = (uint64_t)lvl; stats[1] = fired; stats[2] = E.n_edge_sat; stats[3] = E.n_evt_sat;
                 for (int i = 0; i < N_ERR; ++i) stats[8 + i] = E.err[i]; }
    for (uint32_t i = 0; i < h->n_edt; ++i) free(E.ed[i].order);
    free(E.dbl); free(E.acq);
    free(E.ev); free(E.ed); free(E.slots); free(E.db_live); free(E.stack); free(E.ready_next);
    return 0;
}
