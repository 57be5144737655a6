"""Hand-written op streams that pin the semantic rules of SURVEY.md s8(a) (rules 1-7, 11).

Each mirrors what the reference's own non-regression tests exercise (ocr/tests/non-regression-tests/
events/testOcrEventLatch*.c, testOcrEventOnce*.c, testOcrEventIdem*.c, testOcrEventMixed*.c,
testOcrEventToEventDeps0.c, testOcrAddDeps*.c, testOcrDepv*.c, testOcrOutputEvent*.c,
testOcrEventParam*.c, dist/testDuplicateDeps*.c) as data, so all three executors can run them.
Every stream is well formed: its outcome does not depend on the schedule.
"""
import numpy as np

from ocr_amd.opstream import (BODY_FWD0, BODY_HASH, BODY_NOP, EVT_CHANNEL, EVT_COUNTED, EVT_IDEM, EVT_LATCH, EVT_ONCE, EVT_STICKY,
                              LATCH_DECR, LATCH_INCR, MODE_NULL, MODE_RO, MODE_RW, Builder, cholesky_dag, layered_dag,
                              powerlaw_dag, stencil_dag)


def chain(b, n, start):
    """n HASH EDTs in a chain hanging off event `start`; returns the list (levels 0..n-1)."""
    out, prev = [], start
    for _ in range(n):
        e = b.edt(1)
        b.add_dep(prev, e, 0, MODE_RO)
        out.append(e)
        prev = b.out(e)
    return out


def latch_decr_then_incr():
    """Rule 1: fires when the signed counter becomes 0 after ANY op; DECR,INCR on a fresh latch fires."""
    b = Builder("latch_decr_then_incr")
    start = b.evt(EVT_ONCE)
    l = b.evt(EVT_LATCH)
    a = b.edt(1)
    b.add_dep(l, a, 0, MODE_NULL)
    c = chain(b, 2, start)
    b.script_satisfy(c[0], l, 0, LATCH_DECR)   # level 0: counter -1
    b.script_satisfy(c[1], l, 0, LATCH_INCR)   # level 1: counter 0 -> fires
    return b.finish([a] + c, start)


def latch_incr2_decr2():
    """INCR,INCR,DECR,DECR: fires on the last DECR only."""
    b = Builder("latch_incr2_decr2")
    start = b.evt(EVT_ONCE)
    l = b.evt(EVT_LATCH)
    a = b.edt(1)
    b.add_dep(l, a, 0, MODE_RO)
    c = chain(b, 4, start)
    for e, s in zip(c, (LATCH_INCR, LATCH_INCR, LATCH_DECR, LATCH_DECR)):
        b.script_satisfy(e, l, b.outdb(e), s)   # the waiter sees the payload of the triggering DECR
    return b.finish([a] + c, start)


def latch_init_counter():
    """testOcrEventParam: initial counter 3, DECRs at three different levels, payload forwarded."""
    b = Builder("latch_init_counter")
    start = b.evt(EVT_ONCE)
    l = b.evt(EVT_LATCH, 3)
    a = b.edt(2)
    b.add_dep(l, a, 0, MODE_RO)
    b.add_dep(l, a, 1, MODE_NULL)   # same latch, second slot, control only
    c = chain(b, 3, start)
    for e in c:
        b.script_satisfy(e, l, b.outdb(e), LATCH_DECR)
    never = b.evt(EVT_LATCH, 2)      # stays alive with counter 1
    b.script_satisfy(c[0], never, 0, LATCH_DECR)
    return b.finish([a] + c, start)


def once_no_waiters():
    """Rule 2: a ONCE satisfied with zero waiters is dropped and destroyed."""
    b = Builder("once_no_waiters")
    start = b.evt(EVT_ONCE)
    lonely = b.evt(EVT_ONCE)
    untouched = b.evt(EVT_ONCE)  # never satisfied: stays alive
    c = chain(b, 2, start)
    b.script_satisfy(c[0], lonely, b.outdb(c[0]))
    tail = b.edt(1, outevt=True)   # its output event has no waiter except the sink latch
    b.add_dep(b.out(c[1]), tail, 0)
    return b.finish(c + [tail], start)


def sticky_early_and_late():
    """Rule 3: waiters registered before AND after a STICKY's satisfaction all get its payload."""
    b = Builder("sticky_early_and_late")
    start = b.evt(EVT_ONCE)
    d = b.db(0x1111)
    st_main = b.evt(EVT_STICKY)
    b.satisfy(st_main, d)                 # satisfied by main before anyone registers
    late = b.edt(2)
    b.add_dep(st_main, late, 0, MODE_RO)  # immediate satisfy with the stored guid
    b.add_dep(start, late, 1, MODE_NULL)
    st = b.evt(EVT_STICKY)
    early = b.edt(1)
    b.add_dep(st, early, 0, MODE_RO)
    c = chain(b, 2, start)
    b.script_satisfy(c[1], st, b.outdb(c[1]))
    st_null = b.evt(EVT_STICKY)           # satisfied with NULL_GUID
    b.script_satisfy(c[0], st_null, 0)
    en = b.edt(1)
    b.add_dep(st_null, en, 0, MODE_RO)
    st_never = b.evt(EVT_STICKY)          # never satisfied
    blocked = b.edt(1)                    # never fires
    b.add_dep(st_never, blocked, 0)
    return b.finish([late, early, en] + c, start)


def counted_events():
    """COUNTED events (hc-event.c:537-565,1176-1248): sticky until nbDeps waiters have been served, then destroyed;
    waiters before and after the satisfy both count."""
    b = Builder("counted_events")
    start = b.evt(EVT_ONCE)
    c = chain(b, 2, start)
    # all registrations before the satisfy: fires both, destroyed
    c1 = b.evt(EVT_COUNTED, 2)
    w1, w2 = b.edt(1), b.edt(2)
    b.add_dep(c1, w1, 0, MODE_RO)
    b.add_dep(c1, w2, 0, MODE_RO)
    b.add_dep(start, w2, 1, MODE_NULL)
    b.script_satisfy(c[1], c1, b.outdb(c[1]))
    # one early waiter, satisfied by main, two late waiters: the third registration destroys it
    c2 = b.evt(EVT_COUNTED, 3)
    d = b.db(0x2222)
    e1 = b.edt(2)
    b.add_dep(c2, e1, 0, MODE_RO)
    b.add_dep(start, e1, 1, MODE_NULL)
    b.satisfy(c2, d)
    l1, l2 = b.edt(2), b.edt(2)
    b.add_dep(c2, l1, 0, MODE_RO)
    b.add_dep(start, l1, 1, MODE_NULL)
    b.add_dep(c2, l2, 0, MODE_NULL)
    b.add_dep(start, l2, 1, MODE_NULL)
    # satisfied with NULL_GUID, only one of two registrations ever comes: stays satisfied
    c3 = b.evt(EVT_COUNTED, 2)
    b.satisfy(c3, 0)
    h = b.edt(2)
    b.add_dep(c3, h, 0, MODE_RO)
    b.add_dep(start, h, 1, MODE_NULL)
    # never satisfied: its waiter never fires
    c4 = b.evt(EVT_COUNTED, 1)
    blocked = b.edt(1)
    b.add_dep(c4, blocked, 0)
    # a COUNTED event feeding another event
    c5 = b.evt(EVT_COUNTED, 1)
    st = b.evt(EVT_STICKY)
    b.add_dep(c5, st, 0)
    g = b.edt(1)
    b.add_dep(st, g, 0, MODE_RO)
    b.script_satisfy(c[0], c5, b.outdb(c[0]))
    return b.finish([w1, w2, e1, l1, l2, h, g] + c, start)


def counted_after_null_preslot():
    """A COUNTED dependence added AFTER an earlier DB_MODE_NULL pre-slot of the EDT was satisfied by an event: the EDT's
    frontier is already past that pre-slot, so the registration is made (and counted) at once and the satisfy that follows
    serves it and destroys the event (registerSignalerTaskHc hc-task.c:1391-1405; hc-event.c:1211-1238).  Second EDT: the
    same with the pre-slot still unsatisfied -- the registration waits for the frontier and is made when it arrives."""
    b = Builder("counted_after_null_preslot")
    start = b.evt(EVT_ONCE)
    x = b.evt(EVT_ONCE)
    e0 = b.edt(3)
    b.add_dep(x, e0, 0, MODE_NULL)
    b.satisfy(x, 0)                        # pre-slot 0 is satisfied before the COUNTED source exists
    c0 = b.evt(EVT_COUNTED, 1)
    b.add_dep(c0, e0, 1, MODE_RO)
    d = b.db(0x7777)
    b.satisfy(c0, d)                       # serves its one registration: destroyed
    b.add_dep(start, e0, 2, MODE_NULL)     # (e0 waits for the start event like everybody: an EDT that became ready while the main
                                           #  program is still registering waiters on its output event is a race with >1 worker)
    y = b.evt(EVT_ONCE)
    e1 = b.edt(3)
    b.add_dep(y, e1, 0, MODE_NULL)
    b.add_dep(start, e1, 1, MODE_NULL)
    c1 = b.evt(EVT_COUNTED, 1)
    b.add_dep(c1, e1, 2, MODE_RO)          # held back: pre-slot 0 is not satisfied yet
    b.satisfy(c1, d)                       # nothing registered yet: stays satisfied ...
    g = b.edt(1, BODY_NOP)
    b.add_dep(start, g, 0, MODE_RO)
    b.script_satisfy(g, y, 0)              # ... until level 0 satisfies pre-slot 0: frontier reaches 2, counted, destroyed
    # the same on an EDT with more than 6 pre-slots (their words live outside the EDT's line): pre-slots 0..6 are satisfied by
    # DB_MODE_NULL events before the COUNTED source on pre-slot 7 exists; pre-slot 8 comes later
    e2 = b.edt(9)
    zs = [b.evt(EVT_ONCE) for _ in range(7)]
    for k, z in enumerate(zs):
        b.add_dep(z, e2, k, MODE_NULL)
    for z in zs:
        b.satisfy(z, 0)
    c2 = b.evt(EVT_COUNTED, 1)
    b.add_dep(c2, e2, 7, MODE_RO)          # the frontier stands at 7: counted at once
    b.satisfy(c2, d)
    b.add_dep(start, e2, 8, MODE_RO)
    return b.finish([e0, e1, g, e2], start)


def idem_double_satisfy():
    """Rule 3: the 2nd satisfy of an IDEM is ignored; everybody sees the first payload."""
    b = Builder("idem_double_satisfy")
    start = b.evt(EVT_ONCE)
    idem = b.evt(EVT_IDEM)
    w0 = b.edt(1)
    b.add_dep(idem, w0, 0, MODE_RO)
    c = chain(b, 3, start)
    b.script_satisfy(c[0], idem, b.outdb(c[0]))
    b.script_satisfy(c[2], idem, b.outdb(c[2]))   # ignored
    idem2 = b.evt(EVT_IDEM)
    d = b.db(77)
    b.satisfy(idem2, d)
    b.satisfy(idem2, 0)                            # ignored, from main
    w1 = b.edt(2)
    b.add_dep(idem2, w1, 0, MODE_RO)
    b.add_dep(start, w1, 1, MODE_NULL)
    return b.finish([w0, w1] + c, start)


def mode_null_forces_null():
    """Rule 5: depv[slot].guid is forced to NULL_GUID when the dependence mode is DB_MODE_NULL."""
    b = Builder("mode_null_forces_null")
    start = b.evt(EVT_ONCE)
    c = chain(b, 1, start)
    a = b.edt(3)
    b.add_dep(b.out(c[0]), a, 0, MODE_NULL)
    b.add_dep(b.out(c[0]), a, 1, MODE_RO)
    b.add_dep(b.out(c[0]), a, 2, MODE_RW)
    d = b.db(5)
    x = b.edt(3)
    b.add_dep(d, x, 0, MODE_NULL)    # DB -> EDT with NULL mode: recorded, payload dropped
    b.add_dep(d, x, 1, MODE_RO)      # rule 6: same DB on two slots
    b.add_dep(start, x, 2, MODE_NULL)
    return b.finish(c + [a, x], start)


def db_and_null_sources():
    """Rule 7: ocrAddDependence(NULL_GUID, edt, slot) == satisfy(NULL); (db, evt) == satisfy(evt, db)."""
    b = Builder("db_and_null_sources")
    start = b.evt(EVT_ONCE)
    d0, d1 = b.db(10), b.db(20)
    e = b.edt(4)
    b.add_dep(d0, e, 0, MODE_RO)
    b.add_dep(0, e, 1, MODE_RO)       # NULL source
    b.add_dep(d1, e, 2, MODE_RW)
    b.add_dep(start, e, 3, MODE_NULL)
    once = b.evt(EVT_ONCE)
    f = b.edt(2)
    b.add_dep(once, f, 0, MODE_RO)
    b.add_dep(start, f, 1, MODE_NULL)
    b.add_dep(d1, once, 0)            # DB -> event == satisfy
    return b.finish([e, f], start)


def event_chain():
    """testOcrEventToEventDeps0: once -> once -> sticky -> latch(1) -> EDT; payload forwarded unchanged."""
    b = Builder("event_chain")
    start = b.evt(EVT_ONCE)
    c = chain(b, 2, start)
    o1, o2, st, l = b.evt(EVT_ONCE), b.evt(EVT_ONCE), b.evt(EVT_STICKY), b.evt(EVT_LATCH, 1)
    b.add_dep(o1, o2, 0)
    b.add_dep(o2, st, 0)
    b.add_dep(st, l, LATCH_DECR)
    a = b.edt(2)
    b.add_dep(l, a, 0, MODE_RO)
    b.add_dep(st, a, 1, MODE_RO)
    b.add_dep(b.out(c[1]), o1, 0)     # output event -> event
    fan = [b.edt(1) for _ in range(9)]  # > HCEVT_WAITER_STATIC_COUNT waiters on one event
    for f in fan:
        b.add_dep(o2, f, 0, MODE_RO)
    return b.finish(c + [a] + fan, start)


def forward_and_nop():
    """testOcrOutputEvent: the output event carries whatever guid the body returns (FWD0 / NULL)."""
    b = Builder("forward_and_nop")
    start = b.evt(EVT_ONCE)
    d = b.db(0xABCD)
    fw = b.edt(2, BODY_FWD0)
    b.add_dep(d, fw, 0, MODE_RO)
    b.add_dep(start, fw, 1, MODE_NULL)
    nop = b.edt(1, BODY_NOP)
    b.add_dep(start, nop, 0, MODE_NULL)
    a = b.edt(2)
    b.add_dep(b.out(fw), a, 0, MODE_RO)
    b.add_dep(b.out(nop), a, 1, MODE_RO)
    quiet = b.edt(1, BODY_NOP, outevt=False)   # no output event at all
    b.add_dep(b.out(a), quiet, 0, MODE_RO)
    return b.finish([fw, nop, a, quiet], start)


def destroy_and_unfired():
    """ocrEventDestroy on persistent events; EDTs whose slots never fill stay unfired."""
    b = Builder("destroy_and_unfired")
    start = b.evt(EVT_ONCE)
    st = b.evt(EVT_STICKY)
    c = chain(b, 3, start)
    w = b.edt(1)
    b.add_dep(st, w, 0, MODE_RO)
    b.script_satisfy(c[0], st, b.outdb(c[0]))
    b.script_destroy(c[2], st)                 # destroyed two levels after its satisfaction
    st2 = b.evt(EVT_IDEM)
    b.destroy(st2)                             # created and destroyed by main, never used
    half = b.edt(2)                            # only one of two slots is ever satisfied
    b.add_dep(b.out(c[0]), half, 0, MODE_RO)
    return b.finish(c + [w], start)


def wide_fanin():
    """One EDT with 40 pre-slots fed from 40 producers on two levels + a 200-waiter ONCE."""
    b = Builder("wide_fanin")
    start = b.evt(EVT_ONCE)
    prods = []
    for i in range(20):
        c = chain(b, 2, start)
        prods += c
    big = b.edt(40)
    for i, p in enumerate(prods):
        b.add_dep(b.out(p), big, i, MODE_RO if i % 3 else MODE_NULL)
    fan = [b.edt(1) for _ in range(200)]
    for f in fan:
        b.add_dep(b.out(big), f, 0, MODE_RO)
    return b.finish(prods + [big] + fan, start)


def channel_events():
    """CHANNEL events (hc-event.c:2081-2156): the k-th satisfy goes to the k-th registered waiter, whichever comes
    first; satisfies with nobody waiting queue up, waiters with nothing queued wait in line."""
    b = Builder("channel_events")
    start = b.evt(EVT_ONCE)
    d0, d1 = b.db(500), b.db(501)
    ch = b.evt(EVT_CHANNEL, 4)
    b.satisfy(ch, d0)                               # generation 0 queued before anybody waits
    a = b.edt(2, BODY_HASH)
    b.add_dep(start, a, 0)
    b.add_dep(ch, a, 1)                             # takes generation 0 at once
    bb = b.edt(2, BODY_HASH)
    b.add_dep(start, bb, 0)
    b.add_dep(ch, bb, 1)                            # waits for generation 1
    c = b.edt(3, BODY_HASH)
    b.add_dep(start, c, 0)
    b.add_dep(ch, c, 2, MODE_NULL)                  # generation 2, payload dropped by the mode
    b.add_dep(d1, c, 1)
    drv = b.edt(1, BODY_NOP)
    b.add_dep(start, drv, 0)
    b.script_satisfy(drv, ch, d1)                   # generation 1 -> bb (level 1)
    b.script_satisfy(bb, ch, b.outdb(bb))           # generation 2 -> c  (level 2)
    q = b.evt(EVT_CHANNEL, 2)                       # two satisfies nobody takes
    b.satisfy(q, d0)
    b.satisfy(q, 0)
    w = b.evt(EVT_CHANNEL, 2)                       # a waiter nobody serves
    never = b.edt(1, BODY_NOP)
    b.add_dep(w, never, 0)
    gone = b.evt(EVT_CHANNEL, 1)
    b.destroy(gone)
    return b.finish([a, bb, c, drv], start)


def channel_event_deps():
    """CHANNEL events as source AND destination of event dependences (hc-event.c:2081-2156 with an event as the waiter /
    commonSatisfyWaiters reaching a channel): a STICKY's firing is generation 0 of a channel; the channel's registrations --
    in program order: a ONCE event, a LATCH's DECR slot, an EDT pre-slot, a second channel -- each take one generation."""
    b = Builder("channel_event_deps")
    start = b.evt(EVT_ONCE)
    d0, d1 = b.db(900), b.db(901)
    ch = b.evt(EVT_CHANNEL, 8)
    st = b.evt(EVT_STICKY)
    b.add_dep(st, ch, 0)                         # the sticky's satisfy becomes generation 0 of the channel
    once = b.evt(EVT_ONCE)
    lt = b.evt(EVT_LATCH, 1)
    ch2 = b.evt(EVT_CHANNEL, 2)
    b.add_dep(ch, once, 0)                       # registration 0: generation 0 -> ONCE event -> EDT a
    b.add_dep(ch, lt, LATCH_DECR)                # registration 1: generation 1 -> latch DECR -> EDT c
    a = b.edt(2, BODY_HASH)
    b.add_dep(start, a, 0)
    b.add_dep(once, a, 1)
    c = b.edt(2, BODY_NOP)
    b.add_dep(start, c, 0)
    b.add_dep(lt, c, 1)
    e = b.edt(2, BODY_HASH)
    b.add_dep(start, e, 0)
    b.add_dep(ch, e, 1)                          # registration 2: generation 2 -> EDT e's pre-slot
    b.add_dep(ch, ch2, 0)                        # registration 3: generation 3 -> generation 0 of the second channel
    f = b.edt(2, BODY_HASH)
    b.add_dep(start, f, 0)
    b.add_dep(ch2, f, 1)
    drv = b.edt(1, BODY_NOP)
    b.add_dep(start, drv, 0)
    b.script_satisfy(drv, st, d0)                # level 0: sticky fires -> generation 0 -> once -> a at level 1
    b.script_satisfy(a, ch, 0)                   # level 1: generation 1 -> latch -> c at level 2
    b.script_satisfy(c, ch, d1)                  # level 2: generation 2 -> e at level 3 (with d1)
    b.script_satisfy(e, ch, b.outdb(e))          # level 3: generation 3 -> ch2 generation 0 -> f at level 4
    return b.finish([a, c, e, f, drv], start)


SEMANTIC = [channel_event_deps, counted_after_null_preslot, channel_events, latch_decr_then_incr, latch_incr2_decr2, latch_init_counter, once_no_waiters, sticky_early_and_late,
            counted_events, idem_double_satisfy, mode_null_forces_null, db_and_null_sources, event_chain, forward_and_nop,
            destroy_and_unfired, wide_fanin]


def random_stream(seed, n=36, eager_safe=False):
    """A random WELL-FORMED OCR program (no user errors, no payload races): every kind of source on EDT pre-slots
    (start event, datablocks, NULL_GUID, output events, ONCE / STICKY / IDEM / COUNTED / LATCH / CHANNEL events driven by
    EDT scripts, by main, by other events, or by nobody), inline and overflow pre-slots, all three bodies, all payload
    kinds.  Used as fuzz input: reference runtime == restatement on every seed.
    eager_safe=True keeps COUNTED events off pre-slots the EDT's frontier may never reach: the reference registers an
    EDT on a persistent event only when all earlier pre-slots are satisfied (hc-task.c:1286-1338), the GPU engine
    registers at ocrAddDependence; the two differ only in whether such a COUNTED event ends up destroyed."""
    from ocr_amd.opstream import ref_idx
    rng = np.random.default_rng(0x0C52F000 + seed)
    b = Builder(f"fuzz{'_safe' if eager_safe else ''}_{seed}")
    start = b.evt(EVT_ONCE)
    dbs = [b.db(int(rng.integers(1, 1 << 40))) for _ in range(int(rng.integers(1, 5)))]
    edts = []          # (ref, fires, depc, has_outevt, body)
    events = []        # dicts: ref, kind, fires, regs, feeds (event->event edges to add at the end), main_sat
    late = []          # main ops to issue at the very end (event->event dependences, late main satisfies)

    def pick_payload(e_depc, hash_body):
        k = int(rng.integers(0, 4))
        if k == 1:
            return dbs[int(rng.integers(len(dbs)))]
        if k == 2 and hash_body:
            return "outdb"
        if k == 3 and e_depc:
            return b.depv(int(rng.integers(e_depc)))
        return 0

    def new_event(driver_pool, gated_only=False, no_counted=False):
        """Create an explicit event and decide who satisfies it.  Returns its record."""
        kind = [EVT_ONCE, EVT_STICKY, EVT_IDEM, EVT_COUNTED, EVT_LATCH][int(rng.integers(5))]
        if no_counted and kind == EVT_COUNTED:
            kind = EVT_STICKY
        ev = {"kind": kind, "regs": 0, "fires": False, "ref": None, "gated": True}   # gated: cannot fire before `start`
        fired = [d for d in driver_pool if d[1]]
        how = int(rng.integers(0, 10))
        if kind == EVT_LATCH:
            k = int(rng.integers(1, 4))
            drivers = [driver_pool[int(rng.integers(len(driver_pool)))] for _ in range(k)] if driver_pool else []
            ev["ref"] = b.evt(EVT_LATCH, len(drivers))
            for d in drivers:                                  # each contributor checks out once (NULL payload)
                if rng.integers(4) == 0:
                    b.script_satisfy(d[0], ev["ref"], 0, LATCH_INCR)
                    b.script_satisfy(d[0], ev["ref"], 0, LATCH_DECR)
                b.script_satisfy(d[0], ev["ref"], 0, LATCH_DECR)
            ev["fires"] = bool(drivers) and all(d[1] for d in drivers)
            return ev
        ev["ref"] = b.evt(kind, 1 if kind == EVT_COUNTED else 0)
        if how == 0 or not driver_pool:                        # never satisfied
            return ev
        if how <= 2 and kind != EVT_ONCE and not gated_only:   # satisfied by main right away: later waiters are "late"
            b.satisfy(ev["ref"], dbs[int(rng.integers(len(dbs)))] if rng.integers(2) else 0)
            ev["fires"] = True
            ev["gated"] = False
            return ev
        if how == 3 and events:                                # fed by another event (added at the end of main)
            srcs = [x for x in events if x["gated"] or not gated_only]
            if not srcs:
                return ev
            src = srcs[int(rng.integers(len(srcs)))]
            late.append(("dep", src, ev))
            src["regs"] += 1
            ev["fires"] = src["fires"]
            ev["gated"] = src["gated"]
            return ev
        d = driver_pool[int(rng.integers(len(driver_pool)))]   # satisfied by an EDT's script
        body_hash = (d[4] & 0xFF) == BODY_HASH
        pay = pick_payload(d[2], body_hash)
        pay = b.outdb(d[0]) if pay == "outdb" else pay
        b.script_satisfy(d[0], ev["ref"], pay)
        if kind == EVT_IDEM and rng.integers(2):               # a second satisfy from the same body is ignored
            b.script_satisfy(d[0], ev["ref"], 0)
        ev["fires"] = d[1]
        return ev

    chans = []         # CHANNEL events: only on pre-slots k > 0; satisfied by main (early and late) and by at most ONE body

    def channel_source(driver_pool):
        """A CHANNEL event for the next waiter and whether that waiter will be served.  Generation g of a channel goes
        to its g-th registered waiter: main's early satisfies first, then main's late ones, then the one body's."""
        ch = chans[int(rng.integers(len(chans)))] if chans and rng.integers(2) else None
        if ch is None:
            ch = {"ref": b.evt(EVT_CHANNEL, 8), "early": int(rng.integers(0, 3)), "late": int(rng.integers(0, 2)), "drv": None, "regs": 0}
            for _ in range(ch["early"]):
                b.satisfy(ch["ref"], dbs[int(rng.integers(len(dbs)))] if rng.integers(2) else 0)
            for _ in range(ch["late"]):
                late.append(("sat", ch, dbs[int(rng.integers(len(dbs)))] if rng.integers(2) else 0))
            if driver_pool and rng.integers(3):
                d = driver_pool[int(rng.integers(len(driver_pool)))]
                pay = pick_payload(d[2], (d[4] & 0xFF) == BODY_HASH)
                b.script_satisfy(d[0], ch["ref"], b.outdb(d[0]) if pay == "outdb" else pay)
                ch["drv"] = d
            chans.append(ch)
        g = ch["regs"]
        ch["regs"] += 1
        if ch["regs"] > 7:
            chans.remove(ch)                                   # stay below maxGen waiters in line
        served = g < ch["early"] + ch["late"] or (g == ch["early"] + ch["late"] and ch["drv"] is not None and ch["drv"][1])
        return ch["ref"], served

    for i in range(n):
        depc = int([1, 1, 1, 2, 2, 3, 4, 6, 7, 9][int(rng.integers(10))])   # never 0: such an EDT would run while main still builds
        body = [BODY_HASH, BODY_HASH, BODY_NOP, BODY_FWD0][int(rng.integers(4))]
        outevt = bool(rng.integers(5))
        e = b.edt(depc, body, outevt=outevt)
        fires = True
        drivers = [x for x in edts if x[2] > 0]                # a depc==0 EDT runs while main is still building: it drives nothing
        for k in range(depc):
            # slot 0 is always gated by `start` (directly or through something that is), so no EDT that others
            # depend on can run -- on any worker -- before main has finished wiring the program
            t = int(rng.integers(0, 12)) if k else [0, 0, 5, 9][int(rng.integers(4))]
            mode = MODE_NULL if rng.integers(4) == 0 else MODE_RO
            outs = [x for x in edts if x[3] and x[2] > 0]
            if k and rng.integers(10) == 0:
                cref, served = channel_source(drivers)
                b.add_dep(cref, e, k, mode)
                fires &= served
            elif t <= 1:
                b.add_dep(start, e, k, mode)
            elif t == 2:
                b.add_dep(dbs[int(rng.integers(len(dbs)))], e, k, mode)
            elif t == 3:
                b.add_dep(0, e, k, mode)
            elif t <= 7 and outs:
                src = outs[int(rng.integers(len(outs)))]
                b.add_dep(b.out(src[0]), e, k, mode)
                fires &= src[1]
            elif t <= 7:
                b.add_dep(start, e, k, mode)
            else:
                nc = eager_safe and k > 0 and not fires         # the frontier may never get here
                pool = [x for x in events if (x["gated"] or k) and not (nc and x["kind"] == EVT_COUNTED)]
                ev = pool[int(rng.integers(len(pool)))] if pool and rng.integers(2) else None
                if ev is None:
                    ev = new_event(drivers, gated_only=(k == 0), no_counted=nc)
                    events.append(ev)
                b.add_dep(ev["ref"], e, k, mode)
                ev["regs"] += 1
                fires &= ev["fires"]
        edts.append((e, fires, depc, outevt, body))
    # event -> event dependences, after every EDT registration; newest first, so that a destination has all of its
    # own waiters before a source that main already satisfied fires (and, if ONCE, destroys) it
    for what, src, dst in reversed(late):
        if what == "sat":                                      # a late main satisfy of a CHANNEL: the waiters are in line by now
            b.satisfy(src["ref"], dst)
            continue
        slot = LATCH_DECR if dst["kind"] == EVT_LATCH else 0
        b.add_dep(src["ref"], dst["ref"], slot, MODE_RO)
    for ev in events:                                          # COUNTED: nbDeps = the registrations it will see
        if ev["kind"] == EVT_COUNTED:
            b.evts[ref_idx(ev["ref"])] = (EVT_COUNTED, max(1, ev["regs"]))
    return b.finish([x[0] for x in edts if x[1]], start)


FUZZ_SEEDS = 64        # seeds with a committed reference digest in tests/golden/fuzz.json, per generator flavour


def stream_digest(s):
    """Hash of a stream's op arrays (pins the generator against the committed fuzz goldens)."""
    from ocr_amd.opstream import fnv1a64
    return fnv1a64(b"".join(np.ascontiguousarray(getattr(s, f)).tobytes() for f in ("edt", "evt", "db_init", "main", "script")))


def all_streams():
    out = [f() for f in SEMANTIC]
    out.append(layered_dag(1000, 100, name="layered_1k"))
    out.append(layered_dag(100_000, 1000, name="layered_100k"))
    # small twins of BASELINE configs 3, 4 and 5
    out.append(cholesky_dag(8, name="cholesky_T8"))
    out.append(cholesky_dag(32, name="cholesky_T32"))
    # ... and config 3 with the payload s8d.3 names: 8x8 int64 tile datablocks updated in place by TILE bodies
    out.append(cholesky_dag(8, tiles=True))
    out.append(cholesky_dag(32, tiles=True))
    out.append(stencil_dag(16, 10, name="stencil_16x16x10"))
    out.append(powerlaw_dag(100_000, name="powerlaw_100k"))
    return out
