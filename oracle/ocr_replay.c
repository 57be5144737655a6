/*
 * ocr_replay.c -- TEST INFRASTRUCTURE (the checker), never shipped or measured as product.
 *
 * A plain OCR program that replays an op stream (include/ocr_opstream.h) through the
 * PUBLIC OCR API of the *reference* runtime (linked against oracle/_ref/libocr_x86_*.a,
 * which oracle/Makefile.ref compiles from /root/reference/ocr/src where it lies).  Every
 * dependence decision -- which EDTs fire, what guid each pre-slot receives, which events
 * self-destruct, what a latch counter ends at -- is taken by the reference code:
 *   ocrEdtCreate      ocr/src/api/ocr-edt.c:295   -> newTaskHc        hc-task.c:871
 *   ocrAddDependence  ocr/src/api/ocr-edt.c:551   -> hc-policy.c:2775
 *   ocrEventSatisfySlot ocr-edt.c:127             -> hc-policy.c:3203 -> hc-event.c:437/726/597/620
 *   satisfyTaskHc     hc-task.c:1216, taskExecute hc-task.c:1864, taskEpilogue :1707
 * This file only (a) turns ops into API calls, (b) runs the EDT bodies the stream names,
 * (c) keeps a Lamport clock per event so "topological level" is well defined for pure
 * control dependences (the public API gives a body no way to see who triggered it), and
 * (d) at the sink EDT dumps the parity record: per-EDT level + HASH value, final event
 * states (read from the reference's own metadata through its GUID provider; needs the
 * COUNTED_MAP provider, where a destroyed guid no longer resolves,
 * counted-map-guid.c:394-406,540-548), and datablock values.
 *
 * usage: ocr_replay -ocr:cfg <cfg> <ops-file> <result-file|->
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <sys/time.h>
#include <sched.h>
#include <time.h>
#include "ocr.h"
/* reference internals, only for the final-state peek in dump_event_state() */
#include "ocr-policy-domain.h"
#include "ocr-guid.h"
#include "ocr-sal.h"
#include "event/hc/hc-event.h"
#include "ocr_opstream.h"

#define CLK_UNSET LLONG_MIN
#define CLK_MAIN  (-1LL)

static struct {
    ocrops_hdr_t hdr;
    ocrops_edt_t *edt; ocrops_evt_t *evt; uint64_t *db_init; ocrops_op_t *main_ops, *script;
    uint32_t n_evt_total, n_db_total;
    ocrGuid_t *edt_guid, *evt_guid, *db_guid; uint64_t **db_ptr;
    /* static structure for the Lamport harness */
    uint64_t *slot_base;     /* [n_edt+1] */
    uint64_t *slot_src;      /* [sum depc] source ref of each EDT pre-slot (0 = none/NULL/DB) */
    uint32_t *feed_ptr, *feed_idx; /* CSR: events feeding event x through ocrAddDependence(evt,evt) */
    uint32_t *feed_gen;      /* ... and, when the feeder is a CHANNEL, the generation this registration waits for */
    volatile long long *clk; /* [n_evt_total] */
    /* CHANNEL events pair their k-th satisfy with their k-th waiter (hc-event.c:2081-2156): one clock per generation */
    long long **chan_clk; volatile uint32_t *chan_nsat; uint32_t *chan_nreg; uint32_t *slot_gen; /* slot_gen[pre-slot] = generation it waits for */
    int32_t *level; uint64_t *val;
    volatile uint32_t n_fired;
    const char *res_path; int guid_is_map;
    struct timeval t0; int t0_set;
    uint64_t n_sat_edges;
    /* OCR_REPLAY_LOCKABLE=1 (runs whose cfg names datablockfactory=Lockable): the Lamport clock also counts datablock
     * hand-overs, and bodies watch for holders the arbitration should have kept apart */
    int lockable; uint8_t *slot_mode; volatile long long *db_clk; /* [3*db]: last level of an EW / RW / CONST holder */
    volatile uint64_t *db_busy; volatile uint32_t overlaps;
} G;

static double now_minus(struct timeval *a) {
    struct timeval b; gettimeofday(&b, NULL);
    return (b.tv_sec - a->tv_sec) + 1e-6 * (b.tv_usec - a->tv_usec);
}

static uint32_t evt_index(uint64_t r) { /* unified event index of an EVT / OUTEVT ref */
    return OCRREF_CLS(r) == OCRREF_EVT ? (uint32_t)OCRREF_IDX(r) : G.hdr.n_evt + (uint32_t)OCRREF_IDX(r);
}
static uint32_t evt_kind(uint32_t x) { return x < G.hdr.n_evt ? G.evt[x].kind : OCREVT_ONCE; }

static ocrGuid_t ref_guid(uint64_t r, uint32_t self, ocrEdtDep_t *depv) {
    switch (OCRREF_CLS(r)) {
    case OCRREF_NULL:   return NULL_GUID;
    case OCRREF_EDT:    return G.edt_guid[OCRREF_IDX(r)];
    case OCRREF_EVT:    return G.evt_guid[OCRREF_IDX(r)];
    case OCRREF_OUTEVT: return G.evt_guid[G.hdr.n_evt + OCRREF_IDX(r)];
    case OCRREF_DB:     return G.db_guid[OCRREF_IDX(r)];
    case OCRREF_OUTDB:  return G.db_guid[G.hdr.n_db + OCRREF_IDX(r)];
    case OCRREF_DEPV:   return depv[OCRREF_IDX(r)].guid;
    case OCRREF_UNINIT: return UNINITIALIZED_GUID;
    }
    (void)self; return ERROR_GUID;
}

/* ---- Lamport harness: level of the EDT whose action made event x fire ---- */
static void clk_note(uint32_t x, long long lvl) {
    volatile long long *p = &G.clk[x];
    if (evt_kind(x) == OCREVT_LATCH) { /* fires on the LAST op: max over contributors */
        long long o = *p;
        while (o < lvl && !__sync_bool_compare_and_swap(p, o, lvl)) o = *p;
    } else { /* ONCE / IDEM / STICKY: the first satisfy is the one that fires */
        __sync_bool_compare_and_swap(p, CLK_UNSET, lvl);
    }
}
static long long clk_resolve(uint32_t x);
/* Clock of generation g of CHANNEL x = the (g+1)-th satisfy it has seen.  Its satisfies are the direct ones (noted in chan_clk
 * as they are made) and the firings of the events registered on it (ocrAddDependence(evt, channel): their clocks resolve like
 * any event's).  A well-formed stream orders a channel's satisfies causally, and along a causal chain the clock never
 * decreases, so generation g carries the (g+1)-th smallest clock -- whichever of two equal clocks "really" came first makes
 * no difference to the level computed from it.  CLK_UNSET = fewer than g+1 satisfies so far. */
static long long chan_gen_clk(uint32_t x, uint32_t g) {
    uint32_t nd = G.chan_nsat[x], nf = G.feed_ptr[x + 1] - G.feed_ptr[x], n = 0;
    long long small[64], *v = nd + nf <= 64 ? small : malloc(8 * (size_t)(nd + nf));
    for (uint32_t k = 0; k < nd; ++k) { long long c = G.chan_clk[x][k]; if (c != CLK_UNSET) v[n++] = c; }
    for (uint32_t k = G.feed_ptr[x]; k < G.feed_ptr[x + 1]; ++k) {
        uint32_t f = G.feed_idx[k];
        long long c = evt_kind(f) == OCREVT_CHANNEL ? chan_gen_clk(f, G.feed_gen[k]) : clk_resolve(f);
        if (c != CLK_UNSET) v[n++] = c;
    }
    long long r = CLK_UNSET;
    if (g < n) {
        for (uint32_t i = 1; i < n; ++i) { long long c = v[i]; uint32_t j = i; while (j && v[j - 1] > c) { v[j] = v[j - 1]; --j; } v[j] = c; }
        r = v[g];
    }
    if (v != small) free(v);
    return r;
}
static long long clk_resolve(uint32_t x) {
    long long m = G.clk[x];
    if (evt_kind(x) != OCREVT_LATCH && m != CLK_UNSET) return m;
    for (uint32_t k = G.feed_ptr[x]; k < G.feed_ptr[x + 1]; ++k) {
        uint32_t src = G.feed_idx[k];
        long long f = evt_kind(src) == OCREVT_CHANNEL ? chan_gen_clk(src, G.feed_gen[k]) : clk_resolve(src);
        if (f > m) m = f;
    }
    return m;
}

static void do_satisfy(const ocrops_op_t *op, uint32_t self, long long lvl, ocrEdtDep_t *depv) {
    ocrGuid_t dst = ref_guid(op->a, self, depv), payload = ref_guid(op->b, self, depv);
    uint32_t c = OCRREF_CLS(op->a);
    if (c == OCRREF_EVT && evt_kind(evt_index(op->a)) == OCREVT_CHANNEL) { /* streams keep one satisfier at a time per channel */
        uint32_t x = evt_index(op->a);
        G.chan_clk[x][__sync_fetch_and_add(&G.chan_nsat[x], 1)] = lvl;
    } else
    if (c == OCRREF_EVT || c == OCRREF_OUTEVT) clk_note(evt_index(op->a), lvl);
    ocrEventSatisfySlot(dst, payload, op->slot);
}

static void dump_results(void);

static void clk_max(volatile long long *p, long long v) {
    long long o = *p;
    while (o < v && !__sync_bool_compare_and_swap(p, o, v)) o = *p;
}
/* The datablocks EDT `id` holds while its body runs, as the reference acquired them (iterateDbFrontier hc-task.c:484-544):
 * every non-NULL depv guid once, in the mode of its lowest pre-slot.  Returns how many; db[] / mode[] sized depc. */
static uint32_t held_dbs(uint32_t id, u32 depc, ocrEdtDep_t depv[], uint32_t *db, uint32_t *mode) {
    uint32_t n = 0;
    for (u32 i = 0; i < depc; ++i) {
        if (depv[i].ptr == NULL) continue;
        uint32_t d = (uint32_t)((uint64_t *)depv[i].ptr)[1], seen = 0;
        for (uint32_t k = 0; k < n; ++k) if (db[k] == d) seen = 1;
        if (!seen) { db[n] = d; mode[n++] = G.slot_mode[G.slot_base[id] + i]; }
    }
    return n;
}
#define BUSY_EW (1ULL << 40)
#define BUSY_RW (1ULL << 20)
#define BUSY_CONST 1ULL

static ocrGuid_t bodyEdt(u32 paramc, u64 *paramv, u32 depc, ocrEdtDep_t depv[]) {
    uint32_t id = (uint32_t)paramv[0];
    const ocrops_edt_t *e = &G.edt[id];
    /* level: 1 + max over the pre-slots of the clock of whatever satisfied it */
    long long lvl = 0;
    uint64_t v = ocrops_hash_init(id);
    for (u32 i = 0; i < depc; ++i) {
        uint64_t s = G.slot_src[G.slot_base[id] + i];
        uint32_t c = OCRREF_CLS(s);
        if (c == OCRREF_EVT || c == OCRREF_OUTEVT) {
            long long f = evt_kind(evt_index(s)) == OCREVT_CHANNEL ? chan_gen_clk(evt_index(s), G.slot_gen[G.slot_base[id] + i])
                                                                  : clk_resolve(evt_index(s));
            if (f == CLK_UNSET) { fprintf(stderr, "replay: EDT %u slot %u fired before its source\n", id, i); abort(); }
            if (f + 1 > lvl) lvl = f + 1;
        }
        uint64_t pv = 0, did = 0;
        if (depv[i].ptr != NULL) { pv = ((uint64_t *)depv[i].ptr)[0]; did = ((uint64_t *)depv[i].ptr)[1] + 1; }
        v = ocrops_hash_step(v, i, pv, did);
    }
    uint32_t *hd = NULL, *hm = NULL, nh = 0;
    if (G.lockable && depc) {
        hd = malloc(8 * depc); hm = hd + depc;
        nh = held_dbs(id, depc, depv, hd, hm);
        for (uint32_t k = 0; k < nh; ++k) { /* who else is inside?  EW: nobody; RW: no EW, no CONST; CONST: no writer (RO is never kept out) */
            volatile uint64_t *b = &G.db_busy[hd[k]];
            uint64_t o = 0;
            if (hm[k] == OCRMODE_EW) { o = __sync_fetch_and_add(b, BUSY_EW); if (o) __sync_fetch_and_add(&G.overlaps, 1); }
            else if (hm[k] == OCRMODE_RW) { o = __sync_fetch_and_add(b, BUSY_RW); if (o >= BUSY_EW || (o & (BUSY_RW - 1))) __sync_fetch_and_add(&G.overlaps, 1); }
            else if (hm[k] == OCRMODE_CONST) { o = __sync_fetch_and_add(b, BUSY_CONST); if (o >= BUSY_RW) __sync_fetch_and_add(&G.overlaps, 1); }
        }
        for (uint32_t k = 0; k < nh; ++k) { /* a holder comes after every holder it excludes */
            volatile long long *c = &G.db_clk[3 * hd[k]];
            long long need = -1;
            if (hm[k] == OCRMODE_EW) { need = c[0]; if (c[1] > need) need = c[1]; if (c[2] > need) need = c[2]; }
            else if (hm[k] == OCRMODE_RW) { need = c[0]; if (c[2] > need) need = c[2]; }
            else if (hm[k] == OCRMODE_CONST) { need = c[0]; if (c[1] > need) need = c[1]; }
            if (need + 1 > lvl) lvl = need + 1;
        }
        for (uint32_t k = 0; k < nh; ++k) {
            volatile long long *c = &G.db_clk[3 * hd[k]];
            if (hm[k] == OCRMODE_EW) clk_max(&c[0], lvl); else if (hm[k] == OCRMODE_RW) clk_max(&c[1], lvl); else if (hm[k] == OCRMODE_CONST) clk_max(&c[2], lvl);
        }
        struct timespec ts = { 0, 20000 }; nanosleep(&ts, NULL); /* stay inside long enough for an unkept exclusion to show */
    }
    ocrGuid_t ret = NULL_GUID;
    if (OCRBODY_KIND(e->body) == OCRBODY_TILE) { /* the last tile on the pre-slots is updated in place with the ones before it */
        uint64_t *t[3]; ocrGuid_t tg[3]; uint32_t n = 0;
        for (u32 i = 0; i < depc && n < 3; ++i) if (depv[i].ptr != NULL) { t[n] = (uint64_t *)depv[i].ptr + 2; tg[n++] = depv[i].guid; }
        if (G.hdr.db_words != OCROPS_TILE_WORDS) { fprintf(stderr, "replay: TILE body in a stream without tile datablocks\n"); abort(); }
        v = n ? ocrops_tile_update(t[n - 1], n >= 2 ? t[0] : NULL, n == 3 ? t[1] : NULL) : 0;
        if (n) ret = tg[n - 1];
    }
    switch (OCRBODY_KIND(e->body)) {
    case OCRBODY_HASH: {
        uint64_t *p; ocrGuid_t g;
        ocrDbCreate(&g, (void **)&p, 16, 0, NULL_HINT, NO_ALLOC);
        p[0] = v; p[1] = G.hdr.n_db + id;
        G.db_guid[G.hdr.n_db + id] = g; G.db_ptr[G.hdr.n_db + id] = p;
        ocrDbRelease(g);
        ret = g;
    } break;
    case OCRBODY_FWD0: ret = depc ? depv[0].guid : NULL_GUID; break;
    default: break;
    }
    for (uint32_t k = 0; k < e->script_len; ++k) {
        const ocrops_op_t *op = &G.script[e->script_begin + k];
        if (op->op == OCROP_SATISFY) do_satisfy(op, id, lvl, depv);
        else if (op->op == OCROP_EVT_DESTROY) ocrEventDestroy(ref_guid(op->a, id, depv));
    }
    for (uint32_t k = 0; k < nh; ++k)
        __sync_fetch_and_sub(&G.db_busy[hd[k]], hm[k] == OCRMODE_EW ? BUSY_EW : hm[k] == OCRMODE_RW ? BUSY_RW : hm[k] == OCRMODE_CONST ? BUSY_CONST : 0ULL);
    free(hd);
    if (e->body & OCRBODY_F_OUTEVT) clk_note(G.hdr.n_evt + id, lvl); /* epilogue satisfies it, hc-task.c:1813-1830 */
    G.level[id] = (int32_t)lvl; G.val[id] = v;
    __sync_fetch_and_add(&G.n_fired, 1);
    if (e->body & OCRBODY_F_SINK) {
        double dt = G.t0_set ? now_minus(&G.t0) : 0.0;
        dump_results();
        printf("REPLAY fired=%u edges=%llu resolve_s=%.6f sat_per_s=%.1f\n", G.n_fired,
               (unsigned long long)G.n_sat_edges, dt, dt > 0 ? G.n_sat_edges / dt : 0.0);
        if (G.lockable) printf("REPLAY_LOCKABLE overlaps=%u\n", G.overlaps);
        fflush(stdout);
        ocrShutdown();
    }
    return ret;
}

/* final state of one event, read from the reference's metadata (hc-event.h:61-86) */
static uint64_t dump_event_state(ocrPolicyDomain_t *pd, uint32_t x) {
    ocrGuid_t g = G.evt_guid[x];
    if (ocrGuidIsNull(g)) return OCREVS(OCREVS_ABSENT, 0);
    if (!G.guid_is_map) return OCREVS(OCREVS_ABSENT, 0);
    u64 val = 0; ocrGuidKind kind;
    ocrGuidProvider_t *gp = pd->guidProviders[0];
    gp->fcts.getVal(gp, g, &val, &kind, MD_LOCAL, NULL);
    if (val == 0) return OCREVS(OCREVS_DESTROYED, 0);
    uint32_t k = evt_kind(x);
    if (k == OCREVT_LATCH) return OCREVS(OCREVS_LIVE_UNSAT, (uint32_t)((ocrEventHcLatch_t *)val)->counter);
    if (k == OCREVT_ONCE) return OCREVS(OCREVS_LIVE_UNSAT, 0);
    if (k == OCREVT_CHANNEL) { /* queued satisfies and waiters not yet paired (hc-event.h:105-120) */
        ocrEventHcChannel_t *c = (ocrEventHcChannel_t *)val;
        return OCREVS(OCREVS_LIVE_UNSAT, ((c->tailSat - c->headSat) << 16) | (c->tailWaiter - c->headWaiter));
    }
    ocrGuid_t d = ((ocrEventHcPersist_t *)val)->data; /* UNINITIALIZED_GUID until satisfied, hc-event.c:1616 */
    if (ocrGuidIsUninitialized(d)) return OCREVS(OCREVS_LIVE_UNSAT, 0);
    uint32_t did = 0;
    if (!ocrGuidIsNull(d))
        for (uint32_t i = 0; i < G.n_db_total; ++i) if (ocrGuidIsEq(G.db_guid[i], d)) { did = i + 1; break; }
    return OCREVS(OCREVS_LIVE_SAT, did);
}

/* The sink runs as soon as the last DECR reaches its latch -- which the worker that delivered it does from INSIDE the
 * fan-out of a ONCE output event (satisfyEventHcOnce, hc-event.c:437-465: notify the waiters, THEN destroy the event).
 * With several workers the sink can therefore be dumping while that worker has not yet destroyed the event it is
 * fanning out.  Every EDT that ran has its output event satisfied by its epilogue, so its guid must stop resolving:
 * wait for that (bounded), then give stragglers of the same kind (a latch being destroyed after it fired) a moment. */
static void settle_before_dump(ocrPolicyDomain_t *pd) {
    if (!G.guid_is_map) return;
    ocrGuidProvider_t *gp = pd->guidProviders[0];
    for (int spin = 0; spin < 200000; ++spin) {
        int pending = 0;
        for (uint32_t e = 0; e < G.hdr.n_edt && !pending; ++e) {
            if (G.level[e] < 0 || !(G.edt[e].body & OCRBODY_F_OUTEVT) || (G.edt[e].body & OCRBODY_F_SINK)) continue;
            u64 val = 0; ocrGuidKind kind;
            gp->fcts.getVal(gp, G.evt_guid[G.hdr.n_evt + e], &val, &kind, MD_LOCAL, NULL);
            if (val != 0) pending = 1;
        }
        if (!pending) break;
        sched_yield();
    }
    struct timespec ts = { 0, 2000000 }; nanosleep(&ts, NULL);
}

static void dump_results(void) {
    if (!G.res_path || !strcmp(G.res_path, "-")) return;
    ocrPolicyDomain_t *pd = NULL; getCurrentEnv(&pd, NULL, NULL, NULL);
    settle_before_dump(pd);
    FILE *f = fopen(G.res_path, "wb");
    if (!f) { perror("result file"); abort(); }
    ocrres_hdr_t h; memset(&h, 0, sizeof h);
    h.magic = OCRRES_MAGIC; h.n_edt = G.hdr.n_edt; h.n_evt_total = G.n_evt_total; h.n_db_total = G.n_db_total;
    h.n_fired = G.n_fired;
    fwrite(&h, sizeof h, 1, f);
    fwrite(G.level, 4, G.hdr.n_edt, f);
    fwrite(G.val, 8, G.hdr.n_edt, f);
    for (uint32_t x = 0; x < G.n_evt_total; ++x) { uint64_t s = dump_event_state(pd, x); fwrite(&s, 8, 1, f); }
    for (uint32_t d = 0; d < G.n_db_total; ++d) {
        uint64_t v = G.db_ptr[d] ? G.db_ptr[d][0] : 0;
        if (G.db_ptr[d] && d < G.hdr.n_db && G.hdr.db_words == OCROPS_TILE_WORDS) v = ocrops_tile_checksum(G.db_ptr[d] + 2); /* tile datablocks report their contents */
        fwrite(&v, 8, 1, f);
    }
    fclose(f);
}

static void load_stream(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); abort(); }
    if (fread(&G.hdr, sizeof G.hdr, 1, f) != 1 || G.hdr.magic != OCROPS_MAGIC) { fprintf(stderr, "bad op stream\n"); abort(); }
#define RD(ptr, n, T) do { ptr = (T *)calloc((n) ? (n) : 1, sizeof(T)); if ((n) && fread(ptr, sizeof(T), (n), f) != (n)) { fprintf(stderr, "short read\n"); abort(); } } while (0)
    RD(G.edt, G.hdr.n_edt, ocrops_edt_t); RD(G.evt, G.hdr.n_evt, ocrops_evt_t); RD(G.db_init, G.hdr.n_db, uint64_t);
    RD(G.main_ops, G.hdr.n_main, ocrops_op_t); RD(G.script, G.hdr.n_script, ocrops_op_t);
    fclose(f);
    G.n_evt_total = G.hdr.n_evt + G.hdr.n_edt; G.n_db_total = G.hdr.n_db + G.hdr.n_edt;
    G.edt_guid = calloc(G.hdr.n_edt + 1, sizeof(ocrGuid_t));
    G.evt_guid = calloc(G.n_evt_total + 1, sizeof(ocrGuid_t));
    G.db_guid = calloc(G.n_db_total + 1, sizeof(ocrGuid_t));
    G.db_ptr = calloc(G.n_db_total + 1, sizeof(uint64_t *));
    G.level = malloc(4 * (G.hdr.n_edt + 1)); memset(G.level, 0xFF, 4 * (G.hdr.n_edt + 1));
    G.val = calloc(G.hdr.n_edt + 1, 8);
    G.clk = malloc(8 * (G.n_evt_total + 1));
    for (uint32_t i = 0; i < G.n_evt_total; ++i) G.clk[i] = CLK_UNSET;
    G.slot_base = calloc(G.hdr.n_edt + 1, 8);
    for (uint32_t i = 0; i < G.hdr.n_edt; ++i) G.slot_base[i + 1] = G.slot_base[i] + G.edt[i].depc;
    G.slot_src = calloc(G.slot_base[G.hdr.n_edt] + 1, 8);
    G.lockable = getenv("OCR_REPLAY_LOCKABLE") != NULL;
    G.slot_mode = calloc(G.slot_base[G.hdr.n_edt] + 1, 1);
    G.db_clk = malloc(8 * 3 * (size_t)(G.n_db_total + 1));
    for (size_t i = 0; i < 3 * (size_t)(G.n_db_total + 1); ++i) G.db_clk[i] = -1;
    G.db_busy = calloc(G.n_db_total + 1, 8);
    G.feed_ptr = calloc(G.n_evt_total + 2, 4);
    G.slot_gen = calloc(G.slot_base[G.hdr.n_edt] + 1, 4);
    G.chan_clk = calloc(G.n_evt_total + 1, sizeof(long long *));
    G.chan_nsat = calloc(G.n_evt_total + 1, 4); G.chan_nreg = calloc(G.n_evt_total + 1, 4);
    {   /* one clock per satisfy a channel can ever see */
        uint32_t *cnt = calloc(G.n_evt_total + 1, 4);
        for (uint32_t k = 0; k < G.hdr.n_main; ++k) if (G.main_ops[k].op == OCROP_SATISFY && OCRREF_CLS(G.main_ops[k].a) == OCRREF_EVT) cnt[OCRREF_IDX(G.main_ops[k].a)]++;
        for (uint32_t k = 0; k < G.hdr.n_script; ++k) if (G.script[k].op == OCROP_SATISFY && OCRREF_CLS(G.script[k].a) == OCRREF_EVT) cnt[OCRREF_IDX(G.script[k].a)]++;
        for (uint32_t x = 0; x < G.hdr.n_evt; ++x) if (G.evt[x].kind == OCREVT_CHANNEL) {
            G.chan_clk[x] = malloc(8 * (cnt[x] + 1));
            for (uint32_t k = 0; k <= cnt[x]; ++k) G.chan_clk[x][k] = CLK_UNSET;
        }
        free(cnt);
    }
    /* static pre-pass over main: pre-slot sources and event->event feeders (CSR by destination) */
    uint32_t *cursor = NULL;
    uint32_t *op_gen = calloc(G.hdr.n_main + 1, 4); /* registration k of a CHANNEL waits for its generation k: program order over ALL its waiters */
    for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t k = 0; k < G.hdr.n_main; ++k) {
            const ocrops_op_t *op = &G.main_ops[k];
            if (op->op != OCROP_ADD_DEP) continue;
            uint32_t sc = OCRREF_CLS(op->a), dc = OCRREF_CLS(op->b);
            int src_evt = (sc == OCRREF_EVT || sc == OCRREF_OUTEVT);
            const int src_chan = src_evt && evt_kind(evt_index(op->a)) == OCREVT_CHANNEL;
            if (!pass && src_chan && (dc == OCRREF_EDT || dc == OCRREF_EVT || dc == OCRREF_OUTEVT)) op_gen[k] = G.chan_nreg[evt_index(op->a)]++;
            if (dc == OCRREF_EDT) {
                if (!pass) {
                    G.slot_src[G.slot_base[OCRREF_IDX(op->b)] + op->slot] = op->a; G.n_sat_edges++;
                    G.slot_mode[G.slot_base[OCRREF_IDX(op->b)] + op->slot] = op->mode;
                    if (src_chan) G.slot_gen[G.slot_base[OCRREF_IDX(op->b)] + op->slot] = op_gen[k];
                }
            } else if (src_evt && (dc == OCRREF_EVT || dc == OCRREF_OUTEVT)) {
                uint32_t x = evt_index(op->b);
                if (!pass && evt_kind(x) == OCREVT_LATCH) G.n_sat_edges++;
                if (!pass) G.feed_ptr[x + 1]++;
                else { /* a CHANNEL feeder: this registration is the channel's next one in program order, EDT pre-slots included */
                    const uint32_t src = evt_index(op->a);
                    G.feed_gen[cursor[x]] = op_gen[k];
                    G.feed_idx[cursor[x]++] = src;
                }
            }
        }
        if (!pass) {
            for (uint32_t x = 0; x < G.n_evt_total; ++x) G.feed_ptr[x + 1] += G.feed_ptr[x];
            G.feed_idx = calloc(G.feed_ptr[G.n_evt_total] + 1, 4);
            G.feed_gen = calloc(G.feed_ptr[G.n_evt_total] + 1, 4);
            cursor = malloc(4 * (G.n_evt_total + 1));
            memcpy(cursor, G.feed_ptr, 4 * (G.n_evt_total + 1));
        }
    }
    free(cursor); free(op_gen);
    for (uint32_t k = 0; k < G.hdr.n_script; ++k)
        if (G.script[k].op == OCROP_SATISFY && evt_kind(evt_index(G.script[k].a)) == OCREVT_LATCH) G.n_sat_edges++;
}

ocrGuid_t mainEdt(u32 paramc, u64 *paramv, u32 depc, ocrEdtDep_t depv[]) {
    void *args = depv[0].ptr;
    if (getArgc(args) < 3) { fprintf(stderr, "usage: ocr_replay -ocr:cfg <cfg> <ops> <result|->\n"); ocrAbort(2); return NULL_GUID; }
    load_stream(getArgv(args, 1));
    G.res_path = getArgv(args, 2);
    {   /* COUNTED_MAP guids carry the kind in bits 49..53 (guid-bitmap.h:22-71); PTR guids are heap addresses */
        ocrGuid_t probe; ocrEventCreate(&probe, OCR_EVENT_STICKY_T, EVT_PROP_NONE);
        G.guid_is_map = (((uint64_t)probe.guid >> 49) & 0x1F) == 21 /* OCR_GUID_EVENT_STICKY */ && ((uint64_t)probe.guid >> 54) == 0;
        ocrEventDestroy(probe);
    }
    ocrGuid_t tpl;
    ocrEdtTemplateCreate(&tpl, bodyEdt, 1, EDT_PARAM_UNK);
    for (uint32_t k = 0; k < G.hdr.n_main; ++k) {
        const ocrops_op_t *op = &G.main_ops[k];
        switch (op->op) {
        case OCROP_EVT_CREATE: {
            uint32_t x = (uint32_t)op->a; const ocrops_evt_t *ev = &G.evt[x];
            if ((ev->kind == OCREVT_LATCH && ev->arg != 0) || ev->kind == OCREVT_COUNTED || ev->kind == OCREVT_CHANNEL) {
                ocrEventParams_t p; memset(&p, 0, sizeof p);
                if (ev->kind == OCREVT_LATCH) p.EVENT_LATCH.counter = ev->arg;
                else if (ev->kind == OCREVT_COUNTED) p.EVENT_COUNTED.nbDeps = ev->arg;
                else { p.EVENT_CHANNEL.maxGen = (u32)ev->arg; p.EVENT_CHANNEL.nbSat = 1; p.EVENT_CHANNEL.nbDeps = 1; } /* the only shape supported, hc-event.c:1629 */
                ocrEventCreateParams(&G.evt_guid[x], (ocrEventTypes_t)ev->kind, EVT_PROP_TAKES_ARG, &p);
            } else
                ocrEventCreate(&G.evt_guid[x], (ocrEventTypes_t)ev->kind, EVT_PROP_TAKES_ARG);
        } break;
        case OCROP_DB_CREATE: {
            uint32_t d = (uint32_t)op->a; uint64_t *p;
            ocrDbCreate(&G.db_guid[d], (void **)&p, 16 + 8 * G.hdr.db_words, 0, NULL_HINT, NO_ALLOC);
            p[0] = G.db_init[d]; p[1] = d; G.db_ptr[d] = p;
            for (uint32_t w = 0; w < G.hdr.db_words; ++w) p[2 + w] = ocrops_tile_init(G.db_init[d], w);
            ocrDbRelease(G.db_guid[d]);
        } break;
        case OCROP_EDT_CREATE: {
            uint32_t e = (uint32_t)op->a; u64 id = e;
            int oe = (G.edt[e].body & OCRBODY_F_OUTEVT) != 0;
            ocrEdtCreate(&G.edt_guid[e], tpl, 1, &id, G.edt[e].depc, NULL, EDT_PROP_NONE, NULL_HINT,
                         oe ? &G.evt_guid[G.hdr.n_evt + e] : NULL);
        } break;
        case OCROP_ADD_DEP:
            {   /* (db|NULL, event) is an immediate satisfy (hc-policy.c:2857-2902): note its clock */
                uint32_t sc = OCRREF_CLS(op->a), dc = OCRREF_CLS(op->b);
                if ((sc == OCRREF_NULL || sc == OCRREF_DB || sc == OCRREF_OUTDB) && (dc == OCRREF_EVT || dc == OCRREF_OUTEVT))
                    clk_note(evt_index(op->b), CLK_MAIN);
            }
            ocrAddDependence(ref_guid(op->a, 0, NULL), ref_guid(op->b, 0, NULL), op->slot, (ocrDbAccessMode_t)op->mode);
            break;
        case OCROP_SATISFY:
            if (!G.t0_set) { gettimeofday(&G.t0, NULL); G.t0_set = 1; }
            do_satisfy(op, 0, CLK_MAIN, NULL);
            break;
        case OCROP_EVT_DESTROY: ocrEventDestroy(ref_guid(op->a, 0, NULL)); break;
        default: fprintf(stderr, "bad op %u\n", op->op); abort();
        }
    }
    return NULL_GUID;
}
