// libocrgpu -- MI355X (gfx950, CDNA4) dependence-resolution engine behind include/ocrgpu.h.
//
// What runs where (DESIGN.md has the long form):
//   HBM-resident state           : one 64-B line per EDT (pending counter, inline pre-slot words, its ONCE
//                                  output event), 16-B event records, CSR waiter rows, the fired-EDT log,
//                                  the satisfy-record log and the GUID open-addressing table.
//   k_step (the hot kernel)      : one launch = one phase.  FIRE: pop a level of ready EDTs from the
//                                  fired log, run their bodies, satisfy their output events in-block
//                                  and fan the satisfies out edge-balanced over the whole workgroup;
//                                  EVENTS: the same fan-out for satisfy records queued by chained
//                                  events / API calls.  Every workgroup of launch K derives the phase
//                                  from launch K-1's replicated state + append counters (no ticket, no
//                                  grid barrier), so the host launches ahead and only polls `done`.
//   wave64 idioms                : ballot + mbcnt prefix for ready-set / record compaction (one global
//                                  atomic per wave), readfirstlane grouping for latch counters (one
//                                  atomic per distinct latch per wave), LDS scan for edge balancing.
// No MFMA: nothing here is a contraction (BASELINE.json north_star).  No CPU fallback: every entry
// point fails with OCRGPU_ENODEV / OCRGPU_EHIP when there is no usable GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <chrono>
#include <algorithm>
#include <atomic>
#include <dlfcn.h>
#include <time.h>
#include <unistd.h>
#include <rccl/rccl.h>   // types only: the library is opened at ocrgpu_comm_init (dlopen), never linked
#include "ocrgpu.h"
#include "ocr_opstream.h"

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t s32;

// ------------------------------------------------------------------------------------------------
// device-side layout
// ------------------------------------------------------------------------------------------------
enum { ST_ABSENT = 0, ST_LIVE = 1, ST_SAT = 2, ST_DESTROYED = 3 };
enum { MODE_EVENTS = 0, MODE_FIRE = 1, MODE_EARLY = 256 /* Step.mode only: queued records are applied in a run that is still to fire its level -- what they send reaches the others a round late unless a settle round follows (Ctrl.nc_sends) */ };
enum { ERR_DOUBLE_SLOT = 0, ERR_LATE_WAITER = 1, ERR_STICKY_RESAT = 2, ERR_DEAD_EVENT = 3, ERR_BAD_OP = 4,
       ERR_Q_OVERFLOW = 5, ERR_LEVEL_OVERFLOW = 6 };
#define RF_DST_EDT 1u   // record / waiter destination is an EDT pre-slot
#define RF_MODE_NULL 2u // dependence mode DB_MODE_NULL: the payload is dropped (hc-task.c:1255)
#define RF_INLINE 4u    // EDT destination: the pre-slot word lives in the EDT's own 64-B line (depc <= 5)
#define RF_PREFILLED 8u // EDT destination: the payload was already written at registration (static payload)
#define RF_FIRED 16u    // record only: the event already fired (its satisfy logic ran inline), just fan out
#define RF_PIECE 32u    // record only: fan out piece `slot` (BIG_ROW edges) of the fired event's row
#define RF_CHANGEN 128u // record only (with RF_FIRED): a CHANNEL pairing; `slot` = index in waiters[] of the waiter that takes `payload`
#define RF_LAZYWALK 64u // record only: pre-slot `slot` of frontier-tracked EDT `dst` was satisfied -> lazy_walk()
#define RF_ACQUIRE 256u // record only (Lockable datablocks): EDT `dst` goes on acquiring its datablocks -> acq_walk()
#define INLINE_SLOTS 5u
#define SLOT_MARK 0x80000000u  // pre-slot word: the slot has been satisfied
#define PEND_LAZY 0x40000000u  // EdtLine.pending: a COUNTED event sits on a pre-slot the frontier has not reached -> track the frontier
#define PEND_ACQ 0x20000000u   // EdtLine.pending: Lockable datablocks -- the EDT acquires its datablocks before it is scheduled
#define PEND_MASK 0x1FFFFFFFu
// EdtLine.pending of an EDT whose pre-slots live in its line (depc <= 5): bits 0..7 = unsatisfied pre-slots, bit 8+k = pre-slot k
// has been satisfied.  The satisfier's ONE atomic adds (1 << (8+k)) - 1: it counts the slot down and marks it, so the frontier
// walk of lazily registered COUNTED events (hc-task.c:1286-1338) can tell a satisfied pre-slot whose payload word was written
// at registration (RF_PREFILLED: the satisfy stores nothing) from an unsatisfied one.  EDTs with more pre-slots keep a plain
// count in bits 0..28 and mark the (always stored, never pre-filled) overflow slot words.
#define PEND_INL_CNT 0xFFu
#define PEND_INL_BIT(k) (0x100u << (k))
#define LAZY_FRONT 0xFFFFFFFFu // lazy_tab key slot of an EDT's frontier word
#define CNT_SAT (1ull << 63)   // counted[]: [satisfied:1 | registrations left:31 | lazy registrations not yet made:32]
#define BODY_F_SCRIPT 0x10000u // internal: EDT has a script
#define SCRIPT_F_LATCH 0x20000u // internal: a script op that satisfies a LATCH (0x10000 = the target is remote)
#define SCRIPT_F_IDEM 0x40000u  // internal: a script op that satisfies an IDEM event
#define SOP_MERGED 0u           // internal: script op folded into an earlier one

// One explicit event = 16 B: CSR row of its waiters (rebuilt by seal) + state | kind<<8 + LATCH counter /
// persistent payload.
struct EvtRec { u32 row_begin, deg, state_kind; s32 val; };
// One EDT = one 64-B line: everything a satisfy or a fire touches for this EDT shares a sector -- the pending
// counter the atomics hit, the record the firing lane reads, the ONCE output event (event index max_evt+e)
// and up to 5 pre-slot payload words.  EDTs with depc > 5 keep their slot words in `slots[slot_ovf..]`.
// {pending, level} is ONE aligned 8-byte word: the barrier-free resolve (k_async) counts a pre-slot down with a 64-bit returning
// add that brings the level back with the count.  {param0, val} is ONE aligned 16-byte unit: a consumer's body gets the producer's
// tag and value in one load.  val is 0 until the body has run (k_async's consumers go by that, see there); BODY_VAL_DONE says so too.
struct alignas(64) EdtLine {
    u32 pending, level;                   // level: 1 + the highest level among the EDTs that satisfied a pre-slot so far (k_async only)
    u32 out_state, body;                  // state | kind<<8 of the ONCE output event; body kind and flags
    u64 param0, val;                      // paramv[0]; the body's fold (== value of the output DB): together the datablock {tag, value}
    u32 out_row, out_deg;                 // CSR row of the output event's waiters
    u32 depc;
    u32 slot[INLINE_SLOTS];               // depc <= 5: the pre-slot words; else slot[0] = base index into `slots`
};
#define BODY_VAL_DONE 0x20000000u // EdtLine.body: the built-in body has run, `val` is final
static_assert(sizeof(EdtLine) == 64, "EdtLine must be one 64-byte sector");
struct Rec { u32 dst, payload, slot, flags; };      // one pending satisfy: dst = event idx | EDT idx
// CHANNEL event (hc-event.c:2081-2156): generation k pairs the k-th satisfy with the k-th registered waiter.  Two rings of
// `cap` (= maxGen) entries: waiters in the top end of waiters[] (so a pairing is fanned out like a 1-edge row), payloads
// in chan_sat[].  counted[x] = [satisfies so far:32 | registrations so far:32], one atomic per op.
struct ChanRec { u32 cap, wait_base, sat_base, pad; };
struct Script { u32 target, payload, slot, op_kind; }; // op | payload_kind<<8
// A satisfy whose destination lives on another partition (GPU): what crosses xGMI instead of the reference's
// marshalled ocrPolicyMsg_t (hc-dist-policy.c:2208, mpi-comm-platform.c:1055).  The datablock travels by value
// (16-byte DBs), the way hc-dist's ProxyDb_t clones it on acquire (hc-dist-policy.c:382-700).
struct SendRec { u64 guid; u32 slot, flags; u64 value, tag; };  // flags: XF_*
static_assert(sizeof(SendRec) == 32, "SendRec is 32 bytes");
#define XF_MODE_NULL 1u
#define XF_HAS_DB 2u
#define XF_DESTROY 4u    // ocrEventDestroy of an event owned by another partition
#define XF_LAZYREG 16u   // guid/slot = an EDT pre-slot here; value = guid of the COUNTED event (there) registered on it
#define XF_ACTIVATE 32u  // guid = a COUNTED event here: a held-back registration of a remote EDT has now been made
// Tile datablocks (OCRGPU_F_WIDE_DB) cross by value too -- the whole datablock, as hc-dist's ProxyDb_t clones it on a remote acquire
// (hc-dist-policy.c:382-700).  A record that carries one is a HEAD slot followed by 22 continuation slots in the same region: slot k
// holds words 3k, 3k+1, 3k+2 of {64 tile words, the guid the datablock was created under} in its guid / value / tag fields and
// XF_CONT in its flags, so every transport moves tiles as the 32-byte slots it already moves and the apply kernels' one-thread-per-slot
// walk skips what is not a head.
#define XF_HAS_TILE 64u  // head: the 22 slots behind it carry the datablock's tile and its home guid
#define XF_CONT 128u     // continuation slot (not a record of its own)
#define XF_TILE_WB 256u  // head (with XF_HAS_TILE): guid = a datablock HERE; a TILE body updated a copy of it elsewhere and its contents come
                         // home (the write-back of a released RW / EW proxy, hc-dist-policy.c PD_MSG_DB_RELEASE); satisfies nothing
#define TILE_REC_SLOTS 23u
// Lockable datablocks (OCRGPU_F_LOCKABLE_DB) across partitions: a datablock is arbitrated WHERE IT LIVES (lockable-datablock.c:139-263,
// 330-507 run on the owner's policy domain; hc-dist forwards PD_MSG_DB_ACQUIRE / DB_RELEASE there, hc-dist-policy.c:1027-1031).
#define XF_HAS_HOME 512u   // head + ONE continuation slot whose guid field is the guid the datablock was created under: a copy stays tied to
                           // the original, whose owner arbitrates every access to it
#define XF_ACQ 1024u       // guid = a datablock HERE, slot = access mode, value = requester (location << 32 | EDT index): acquire, or wait in line
#define XF_GRANT 2048u     // value = EDT index HERE: the datablock it asked another partition for is now its: it goes on acquiring
#define XF_REL 4096u       // guid = a datablock HERE: one remote holder is done with it
#define XF_ACQMODE_SHIFT 16u // bits 16..19 of the flags of a satisfy for an EDT pre-slot: the access mode its ocrAddDependence named
#define HOME_REC_SLOTS 2u
struct RWait { u32 loc, edt, next, pad; }; // a requester on another partition waiting in a datablock's line (list entries above max_edt name these)
#define RF_REMOTE 8u     // waiter flag (with bit0 clear): lo = index into rdst[]; bit2 = dst is an EDT, bit1 = MODE_NULL
struct RemoteDst { u64 guid; u32 slot, flags; };

// Lockable datablock (lockable-datablock.h): state = modeLock << 30 | numUsers; heads of the three LIFO waiter lists
// (EDT index + 1; the link to the next waiter sits in the upper half of the waiting EDT's `val`)
struct DbLock { u32 state, ew, itw, ro; };
#define DBL_EW 1u
#define DBL_ITW 2u
#define DBL_USERS 0x3FFFFFFFu
#define LOCKABLE_MAX_DEPC 4096u // OCRGPU_F_LOCKABLE_DB: most pre-slots an EDT may have

// Phase state of launch K.  EVERY workgroup of launch K derives it from state[(K-1)&1] and the append counters
// of launch K-1 (all stable once K-1 has ended), so no workgroup waits for another: no ticket, no grid barrier.
struct alignas(128) State {
    u32 mode, level, r_begin, r_end, q_begin, q_end, done /* word 6: exchange_finish */, allow_fire;
    u32 r_total, q_total;   // log positions where this launch's appends start
    u32 n_levels, force_level;
    u32 b_begin, b_end;     // ready-log range of the level fired by this launch whose bodies were left for the next launch
    u32 pad[18];
};
struct alignas(128) Cnt { u32 v; u32 pad[31]; };
struct alignas(128) Ctrl {
    // between runs ("absolute" view, used by the API kernels and by k_begin / k_end)
    u32 level, r_end, q_begin, async_block /* objects / dependences k_async does not take (see there) */, n_levels, b_begin, b_end, max_row /* longest waiter row (seal) */, noop_steps /* k_step launches that found nothing to do (rounds queued ahead size their launch budget by it) */, nc_sends /* a level fired with records already queued for other partitions: what arrived set them off (OCRGPU_RX_NO_CASCADE's contract is that nothing does) */, pad1[22];
    u32 r_tail; u32 pad2[31];   // atomics (own cache lines)
    u32 q_tail; u32 pad3[31];
    u32 log_tail; u32 rdst_tail; u32 chan_count, chan_wait_used, chan_sat_used;
    u32 proxy_used;             // proxy datablocks (copies of other partitions' DBs that records carried in) taken from the TOP of db[0,max_db)
    u32 pad4[26];
    u32 err[8];
    DbLock *dbl; unsigned char *acq_mode; // OCRGPU_F_LOCKABLE_DB only (else null): lock word + waiter lists per datablock; access mode per
                                          // EDT pre-slot.  Kept here, not in Dev: the hot kernel's argument block stays as small as it was
    u64 *tiles;                 // OCRGPU_F_WIDE_DB only (else null): the 8x8 tile of 64-bit integers behind datablock i at tiles[64 * i]
    u64 *db_home;               // OCRGPU_F_WIDE_DB / _LOCKABLE_DB: the guid datablock i was created under -- its own, or for a copy that a
                                // record brought in (a proxy) the guid of the original on its home partition
    RWait *rwait; u64 *rrel;    // OCRGPU_F_LOCKABLE_DB with partitions: remote requesters in line; releases that arrived this round (guids)
    u32 rwait_cap, rwait_used, rrel_cap, rrel_n;
    u32 pad5[8];
    // during a run
    State state[2];
    Cnt cnt_r[3], cnt_q[3];     // launch K appends through cnt_*[K%3]; launch K+1 reads them; launch K zeroes [(K+1)%3]
    u32 dummy_val; u32 pad6[31];
};

struct Dev {
    Ctrl *ctrl;
    // EDTs
    EdtLine *edt; u32 *edt_script_begin, *edt_script_len;
    u32 *slots;                 // overflow pre-slot words (depc > 5): dbidx+1 (0 = NULL_GUID); bit31 = satisfied (nanny)
    // events: explicit [0,max_evt) in `evt`; the output event of EDT e is event max_evt+e and lives in edt[e]
    EvtRec *evt;
    u64 *counted;               // COUNTED events: the accounting word (CNT_SAT | left << 32 | lazy pending), see counted_*();
                                // CHANNEL events: satisfies so far << 32 | registrations so far
    ChanRec *chan; u32 *chan_sat; u32 chan_sat_cap; // CHANNEL events: evt[x].val = index into chan[]
    ulonglong2 *lazy_tab; u64 lazy_mask; // (EDT, pre-slot) -> COUNTED event waiting for the frontier; (EDT, LAZY_FRONT) -> frontier
    u64 *waiters;               // CSR payload: lo = dst ; hi = flags<<28 | (pre-slot k of the EDT | latch slot)
    u32 *log_src; u64 *log_rec; // registration log the CSR is built from
    // datablocks (explicit [0,max_db) then the output DB of EDT e at max_db+e): {value, tag}
    ulonglong2 *db;
    // work lists
    u32 *fired;                 // fired/ready log, grouped by level
    u32 *level_start;           // level_start[L] = offset in `fired`
    Rec *q;                     // satisfy-record log
    u64 *blk_stats;             // [MAX_GRID][4] per-workgroup counters (no atomics): edt_sat, evt_sat, fired
    u64 *dbg;                   // diagnostic builds only (-DOCRGPU_STAMPS): per-phase time sums of workgroup 0
    Script *scripts;
    // GUID table
    ulonglong2 *gtab; u64 gmask; u32 gshift;
    // sizes
    u32 max_edt, max_evt, max_db, max_levels, q_cap, location, fired_cap;
    // cross-partition satisfies
    RemoteDst *rdst; u32 rdst_cap; SendRec *send; u32 send_cap, n_loc; u32 *send_cnt; // send_cnt[loc*32]
    u64 max_slots, max_edges, hash_tag_base;
    u32 spread_wgs;             // k_step: a short work list is dealt out over this many workgroups
};

// Reclaiming capacity (ocrgpu_reclaim / ocrgpu_db_destroy): the guids an EDT line / event record was created under (what the
// reference's destructTaskHc hc-task.c:721 and destructEventHc hc-event.c:185-242 hand to PD_MSG_GUID_DESTROY), stacks of free
// indices, and the EDTs that have fired but cannot be taken back yet.  Passed to the few kernels that need them as an argument
// of their own: k_step's argument block stays as it was.
struct EdtGuids { u64 edt, oevt, odb, pad; };
struct ReclDev { EdtGuids *edt_guids; u64 *evt_guids; u32 *free_edt, *free_evt, *free_db, *held, *held_new, *cnt; };
enum { RC_EDT = 0, RC_EVT, RC_DB, RC_HELD, RC_OVF_LIVE, RC_LAZY_LIVE, RC_LOG_KEPT, RC_LAZY_FREED, RC_UNPOPPED, RC_SCRIPT_LIVE, RC_READY, RC_N = 16 };
#define BODY_DB_DEAD 0x80000000u // EdtLine.body: the output datablock of this (HASH) EDT has been destroyed (ocrgpu_db_destroy)
#define BODY_FREE 0x40000000u    // EdtLine.body: the line is on the free stack

#define TOMB (~0ULL)
#define GRESV (~0ULL - 1)   // slot claimed, value not yet published (== UNINITIALIZED_GUID, never a key)

// event accessors: explicit events and EDT output events share one index space
__device__ __forceinline__ u32 *evt_state(const Dev &d, u32 x) { return x < d.max_evt ? &d.evt[x].state_kind : &d.edt[x - d.max_evt].out_state; }
__device__ __forceinline__ s32 *evt_val(const Dev &d, u32 x) { return x < d.max_evt ? &d.evt[x].val : (s32 *)&d.ctrl->dummy_val; } // output events are ONCE: no value
__device__ __forceinline__ u32 *evt_deg(const Dev &d, u32 x) { return x < d.max_evt ? &d.evt[x].deg : &d.edt[x - d.max_evt].out_deg; }
__device__ __forceinline__ u32 *evt_row(const Dev &d, u32 x) { return x < d.max_evt ? &d.evt[x].row_begin : &d.edt[x - d.max_evt].out_row; }

__device__ __forceinline__ u64 mkguid(u32 loc, u32 kind, u64 counter) { return ((u64)loc << 54) | ((u64)kind << 49) | counter; }
__device__ __forceinline__ u32 ghash(const Dev &d, u64 g) { return (u32)((g * 0x9E3779B97F4A7C15ULL) >> d.gshift); }

// Open addressing, linear probing, 16-B slots {guid, value}; 0 = empty, TOMB = removed, GRESV = being written.
// A slot is published value first, key last (value store, agent fence, key exchange), so a reader that matches the key in its
// one 16-B load holds the value that belongs to it; remove only turns the key into TOMB.  hashtableConcBucketLockedPut
// (hashtable.c:498-525): a put takes over the first tombstone it passed once it knows the key is not further down the chain
// (removes give capacity back); a key already present gets the new value.  Concurrent puts of ONE key in one batch may
// leave two entries (guids are unique; the reference's put does not check either): callers that race on a key use tryput.
__device__ __forceinline__ void gtab_publish(const Dev &d, u64 h, u64 key, u64 val) {
    // Value, then key, as two device-scope atomic exchanges on the two halves of ONE 16-byte slot: both are executed by the memory-side
    // atomic unit of the slot's channel in the order this lane issued them, so whoever finds the key (tryput: with an atomic of its own)
    // finds the value.  An agent-scope fence between a plain store and the exchange did the same job at the price of an L2 write-back
    // and invalidate per wave and put (buffer_wbl2 sc1 / buffer_inv sc1 on gfx950): three of them per EDT made k_edt_create 3 ns per EDT.
    __hip_atomic_exchange((unsigned long long *)&d.gtab[h].y, (unsigned long long)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // compiler order only (no instruction)
    __hip_atomic_exchange((unsigned long long *)&d.gtab[h].x, (unsigned long long)key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ void gtab_put(const Dev &d, u64 key, u64 val) {
    for (int attempt = 0; attempt < 1024; ++attempt) { // a lost claim means another put made progress: walk the chain again
        u64 h = ghash(d, key), tomb = ~0ULL;
        bool lost = false;
        for (u64 probe = 0; probe <= d.gmask; ++probe, h = (h + 1) & d.gmask) {
            const u64 k = atomicAdd((unsigned long long *)&d.gtab[h].x, 0ULL);
            if (k == key) { d.gtab[h].y = val; return; }
            if (k == TOMB && tomb == ~0ULL) tomb = h;
            if (k != 0ULL) continue;
            // end of the chain: the key is not in the table
            if (tomb != ~0ULL) break;
            if (atomicCAS((unsigned long long *)&d.gtab[h].x, 0ULL, GRESV) == 0ULL) { gtab_publish(d, h, key, val); return; }
            lost = true; break;
        }
        if (tomb != ~0ULL) {
            if (atomicCAS((unsigned long long *)&d.gtab[tomb].x, TOMB, GRESV) == TOMB) { gtab_publish(d, tomb, key, val); return; }
            lost = true;
        }
        if (!lost) break; // neither an empty slot nor a tombstone on the whole walk: the table is full
    }
    atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u);
}
// hashtableConcBucketLockedTryPut (hashtable.c:527-554 region, :385-420 for the CAS variant): insert unless the key is there;
// returns the value the table holds for the key afterwards.  Linearisable among concurrent tryputs of one key: every
// contender walks the same chain and claims the first EMPTY slot only (no tombstone reuse on this path), and a slot found
// reserved is looked at again until its owner has published it.
__device__ u64 gtab_tryput(const Dev &d, u64 key, u64 val) {
    u64 h = ghash(d, key);
    for (u64 probe = 0; probe <= d.gmask;) {
        const u64 k = atomicCAS((unsigned long long *)&d.gtab[h].x, 0ULL, GRESV);
        if (k == 0ULL) { gtab_publish(d, h, key, val); return val; }
        if (k == GRESV) continue;                      // its owner publishes within a few instructions: look again
        if (k == key) return atomicAdd((unsigned long long *)&d.gtab[h].y, 0ULL);
        ++probe; h = (h + 1) & d.gmask;
    }
    atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u);
    return 0;
}
__device__ u64 gtab_get(const Dev &d, u64 key) {
    if (key == 0ULL || key >= TOMB - 1) return 0;
    u64 h = ghash(d, key);
    for (u64 probe = 0; probe <= d.gmask; ++probe, h = (h + 1) & d.gmask) {
        ulonglong2 s = d.gtab[h];
        if (s.x == key) return s.y;
        if (s.x == 0ULL) return 0;
    }
    return 0;
}
// hashtableConcBucketLockedRemove (hashtable.c:556-575).  The slot becomes a tombstone a later put takes over -- or, when it
// ends its run (the next slot is empty), an empty slot again together with the tombstones right before it, so that chains
// shrink back and tryput (which only claims empty slots) keeps finding room in a table that has seen many removes.  Removes
// run in batches of their own (no put or get in the same launch): within such a launch "the next slot is empty" stays true
// once seen, which is what makes emptying a slot safe (no live key's probe path runs over it).
__device__ bool gtab_remove(const Dev &d, u64 key) { // true: this call took the key out (two removes of one key: one of them)
    u64 h = ghash(d, key);
    for (u64 probe = 0; probe <= d.gmask; ++probe, h = (h + 1) & d.gmask) {
        const u64 k = d.gtab[h].x;
        if (k == 0ULL) return false;
        if (k != key) continue;
        const bool run_end = atomicAdd((unsigned long long *)&d.gtab[(h + 1) & d.gmask].x, 0ULL) == 0ULL;
        if (atomicCAS((unsigned long long *)&d.gtab[h].x, (unsigned long long)key, run_end ? 0ULL : TOMB) != key) return false;
        if (run_end)
            for (u64 b = (h - 1) & d.gmask, n = 0; n < 256; b = (b - 1) & d.gmask, ++n)
                if (atomicCAS((unsigned long long *)&d.gtab[b].x, TOMB, 0ULL) != TOMB) break;
        return true;
    }
    return false;
}
// resolve an engine-made guid to its object index, honouring lazy deletion of dead events
__device__ __forceinline__ bool is_event_kind(u32 k) { return (k & 0x10u) != 0; }
__device__ u32 evt_kind_to_guid_kind(u32 k) {
    switch (k) { case OCREVT_ONCE: return OCRGPU_KIND_EVT_ONCE; case OCREVT_IDEM: return OCRGPU_KIND_EVT_IDEM;
                 case OCREVT_STICKY: return OCRGPU_KIND_EVT_STICKY; case OCREVT_LATCH: return OCRGPU_KIND_EVT_LATCH;
                 case OCREVT_COUNTED: return OCRGPU_KIND_EVT_COUNTED; default: return OCRGPU_KIND_EVT_CHANNEL; }
}

// ------------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u32 rank_in(u64 m) { // number of set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}
// wave-aggregated append: every lane with pred gets a unique index in [tail, tail+popc); one atomic per wave
__device__ __forceinline__ u32 wave_append(u32 *tail, bool pred) {
    u64 m = __ballot(pred);
    u32 base = 0;
    if (m) {
        int leader = __ffsll((long long)m) - 1;
        if ((int)lane_id() == leader) base = atomicAdd(tail, (u32)__popcll(m));
        base = __shfl(base, leader);
    }
    return base + rank_in(m);
}

// workgroup-aggregated append: ONE atomic per workgroup (<= 1024 threads).  Every thread of the workgroup calls it (barriers inside);
// lanes with pred get a unique index, the others an unspecified one.  Same-address device-scope atomics are served one after the
// other (~30 ns each on gfx950, whoever issues them): a batch kernel that appends one entry per element to one list spends its time
// there even with one atomic per wavefront (10.5M registrations = 164k wavefronts = 5 ms; k_reclaim_edts 50M lines = 24 ms).
__device__ __forceinline__ u32 block_append(u32 *tail, bool pred, u32 *lds /* 18 words */) {
    const u64 m = __ballot(pred);
    const u32 w = threadIdx.x >> 6, nw = (blockDim.x + 63u) >> 6;
    if (lane_id() == 0) lds[w] = (u32)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 t = 0;
        for (u32 k = 0; k < nw; ++k) { const u32 c = lds[k]; lds[k] = t; t += c; }
        lds[16] = t ? atomicAdd(tail, t) : 0u;
    }
    __syncthreads();
    const u32 r = lds[16] + lds[w] + rank_in(m);
    __syncthreads(); // lds may be reused by the next call
    return r;
}

// ------------------------------------------------------------------------------------------------
// satisfy primitives
// ------------------------------------------------------------------------------------------------
// satisfyTaskHc (hc-task.c:1216-1266): store the payload word, count the slot, report "all satisfied"
__device__ void emit_remote_raw(const Dev &d, u64 guid, u32 slot, u32 xflags, u64 value, u64 tag);
struct Step { // per-launch constants, identical in every workgroup
    u32 mode, r_begin, r_end, q_begin, q_end, r_base, q_base; u32 *cnt_r, *cnt_q;
};
template <bool NANNY, bool LOCKABLE>
__device__ __forceinline__ bool edt_slot_satisfy(const Dev &d, const Step &S, u32 edt, u32 slot, u32 payload, u32 flags) {
    EdtLine *L = &d.edt[edt];
    u32 word = (flags & RF_MODE_NULL) ? 0u : payload;
    // overflow pre-slots (depc > 5): `slot` is relative, the base sits in the line the atomic is about to touch
    const bool inl = (flags & RF_INLINE) != 0;
    if (NANNY) {
        u32 *sp = inl ? &L->slot[slot] : &d.slots[(size_t)L->slot[0] + slot];
        u32 old = atomicExch(sp, word | SLOT_MARK);
        if (old & SLOT_MARK) { atomicAdd(&d.ctrl->err[ERR_DOUBLE_SLOT], 1u); return false; }
    }
    const u32 old = atomicAdd(&L->pending, inl ? PEND_INL_BIT(slot) - 1u : 0xFFFFFFFFu); // count down (+ mark, inline pre-slots)
    // The payload word goes out BEHIND the atomic: an overflow pre-slot's address needs the base from the destination's line, and
    // that load must not stand in front of the whole wave's atomics (the word is only read after the next kernel boundary)
    if (!NANNY && !(flags & RF_PREFILLED)) *(inl ? &L->slot[slot] : &d.slots[(size_t)L->slot[0] + slot]) = word | SLOT_MARK;
    const u32 left = old & (inl ? PEND_INL_CNT : PEND_MASK);
    if (old & PEND_LAZY) { // rare: this EDT's frontier is tracked (COUNTED accounting); the walk runs in the next EVENTS phase
        const u32 qi = S.q_base + atomicAdd(S.cnt_q, 1u);
        if (qi < d.q_cap) d.q[qi] = Rec{edt, 0u, slot, RF_LAZYWALK}; else atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u);
    }
    if (LOCKABLE) { // Lockable datablocks (every EDT carries PEND_ACQ): all satisfied -> acquire first (taskAllDepvSatisfied)
        if (left == 1u) {
            const u32 qi = S.q_base + atomicAdd(S.cnt_q, 1u);
            if (qi < d.q_cap) d.q[qi] = Rec{edt, 0u, 0u, RF_ACQUIRE}; else atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u);
        }
        return false;
    }
    return left == 1u;
}
// An EDT with more than 6 pre-slots whose frontier starts being tracked (a COUNTED source arrives on a later pre-slot): its
// pre-filled pre-slots were satisfied by bare counter decrements and carry no mark.  If the unmarked pre-slots outnumber the
// unsatisfied ones, the difference are such pre-slots: they are found through the registration log (destination = this EDT,
// pre-filled, source event no longer live) and marked, provided exactly that many qualify.  Rare path, one thread.
__device__ void mark_prefilled_overflow(const Dev &d, u32 didx) {
    EdtLine *L = &d.edt[didx];
    const u32 depc = L->depc;
    if (depc <= INLINE_SLOTS) return;
    u32 *sl = d.slots + L->slot[0];
    u32 unmarked = 0;
    for (u32 k = 0; k < depc; ++k) unmarked += (atomicOr(&sl[k], 0u) & SLOT_MARK) ? 0u : 1u;
    const u32 left = atomicOr(&L->pending, 0u) & PEND_MASK;
    if (unmarked <= left) return;
    u32 n = d.ctrl->log_tail; if (n > d.max_edges) n = (u32)d.max_edges;
    for (int pass = 0; pass < 2; ++pass) { // count the candidates, then mark them if they are exactly the missing ones
        u32 cand = 0;
        for (u32 i = 0; i < n; ++i) {
            const u64 rec = d.log_rec[i];
            const u32 hi = (u32)(rec >> 32), fl = hi >> 28;
            if ((u32)rec != didx || (fl & (RF_DST_EDT | RF_PREFILLED | RF_INLINE)) != (RF_DST_EDT | RF_PREFILLED)) continue;
            const u32 st = *evt_state(d, d.log_src[i]) & 0xFF;
            if (st != ST_DESTROYED && st != ST_SAT) continue;
            u32 *w = &sl[hi & 0x0FFFFFFFu];
            if (pass == 0) cand += (atomicOr(w, 0u) & SLOT_MARK) ? 0u : 1u; else atomicOr(w, SLOT_MARK);
        }
        if (pass == 0 && cand != unmarked - left) return;
    }
}
// has pre-slot f of this EDT been satisfied?  (frontier walks; reads through returning atomics: other CUs write these words)
__device__ __forceinline__ bool slot_satisfied(EdtLine *L, u32 *sl, u32 depc, u32 f) {
    if (atomicOr(&sl[f], 0u) & SLOT_MARK) return true;
    return depc <= INLINE_SLOTS && (atomicOr(&L->pending, 0u) & PEND_INL_BIT(f)) != 0;
}

// ---- COUNTED accounting + lazy frontier registration --------------------------------------------------------
// The reference registers an EDT on a persistent event at pre-slot k only when pre-slots 0..k-1 are satisfied
// (iterateDbFrontier / registerOnFrontier, hc-task.c:1286-1338, 395-413; registerSignalerTaskHc :1391-1405).  For
// STICKY / IDEM sources the order is invisible and the engine registers at ocrAddDependence.  A COUNTED event counts
// its registrations (registerWaiterEventHcCounted hc-event.c:1176-1248, satisfyEventHcCounted :537-565), so for those
// the engine keeps the data path eager and makes the COUNT lazy: a registration on a pre-slot the frontier has not
// reached is entered in lazy_tab and only counted once the EDT's frontier gets there.
__device__ __forceinline__ u64 lazy_key(u32 edt, u32 slot) { return ((u64)(edt + 1) << 32) | slot; }
__device__ u64 *lazy_find(const Dev &d, u64 key) {
    u64 h = (key * 0x9E3779B97F4A7C15ULL) >> 20 & d.lazy_mask;
    for (u64 probe = 0; probe <= d.lazy_mask; ++probe, h = (h + 1) & d.lazy_mask) {
        const u64 k = d.lazy_tab[h].x;
        if (k == key) return (u64 *)&d.lazy_tab[h].y;
        if (k == 0ULL) return nullptr;
    }
    return nullptr;
}
__device__ void lazy_insert(const Dev &d, u64 key, u64 val) { // the first insert of a key sets its value
    u64 h = (key * 0x9E3779B97F4A7C15ULL) >> 20 & d.lazy_mask;
    for (u64 probe = 0; probe <= d.lazy_mask; ++probe, h = (h + 1) & d.lazy_mask) {
        u64 old = atomicCAS((unsigned long long *)&d.lazy_tab[h].x, 0ULL, (unsigned long long)key);
        if (old == 0ULL) { d.lazy_tab[h].y = val; return; }
        if (old == key) return;
    }
    atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u);
}
__device__ __forceinline__ void counted_destroyed(const Dev &d, u32 x) { atomicExch(evt_state(d, x), (u32)ST_DESTROYED | (OCREVT_COUNTED << 8)); }
// a registration made on an already satisfied event is served at once; the last one destroys the event (:1211-1238)
__device__ __forceinline__ void counted_late_registration(const Dev &d, u32 x) {
    if (x >= d.max_evt || ((*evt_state(d, x) >> 8) & 0xFF) != OCREVT_COUNTED) return;
    const u64 o = atomicAdd((unsigned long long *)&d.counted[x], (unsigned long long)-(1ll << 32));
    if (((o >> 32) & 0x7FFFFFFFu) == 1u) counted_destroyed(d, x);
}
// the EDT's frontier has reached a pre-slot whose COUNTED registration was held back: it happens now
__device__ void counted_activate(const Dev &d, u32 x) {
    u64 o = atomicAdd((unsigned long long *)&d.counted[x], 0ull);
    for (;;) {
        u64 nw = o - 1; // one lazy registration fewer
        const bool sat = (o & CNT_SAT) != 0;
        if (sat) nw -= 1ull << 32; // ...and it is served at once
        const u64 prev = atomicCAS((unsigned long long *)&d.counted[x], (unsigned long long)o, (unsigned long long)nw);
        if (prev == o) { if (sat && ((nw >> 32) & 0x7FFFFFFFu) == 0u) counted_destroyed(d, x); return; }
        o = prev;
    }
}
// satisfyEventHcCounted: the registrations made so far (row length minus the held-back ones) are served
__device__ bool counted_satisfy(const Dev &d, u32 x, u32 payload) {
    u32 *sk = evt_state(d, x);
    const u32 old = atomicCAS(sk, (u32)ST_LIVE | (OCREVT_COUNTED << 8), (u32)ST_SAT | (OCREVT_COUNTED << 8));
    if ((old & 0xFF) != ST_LIVE) { atomicAdd(&d.ctrl->err[(old & 0xFF) == ST_SAT ? ERR_STICKY_RESAT : ERR_DEAD_EVENT], 1u); return false; }
    *evt_val(d, x) = (s32)payload;
    const u32 deg = *evt_deg(d, x);
    u64 o = atomicAdd((unsigned long long *)&d.counted[x], 0ull);
    for (;;) {
        const u32 lp = (u32)o, left = (u32)(o >> 32) & 0x7FFFFFFFu;
        const u32 served = deg > lp ? deg - lp : 0u;
        if (served > left) atomicAdd(&d.ctrl->err[ERR_BAD_OP], 1u); // more registrations than nbDeps (assert :553)
        const u32 nl = left > served ? left - served : 0u;
        const u64 nw = CNT_SAT | ((u64)nl << 32) | lp;
        const u64 prev = atomicCAS((unsigned long long *)&d.counted[x], (unsigned long long)o, (unsigned long long)nw);
        if (prev == o) { if (nl == 0u) counted_destroyed(d, x); return true; }
        o = prev;
    }
}
// value of a lazy_tab entry: index of a local COUNTED event, or the guid of one on another partition
__device__ __forceinline__ void lazy_reached(const Dev &d, u64 v) {
    if (v >> 32) emit_remote_raw(d, v, 0u, XF_ACTIVATE, 0ull, 0ull); else counted_activate(d, (u32)v);
}
// A pre-slot of a frontier-tracked EDT was satisfied: mark it where every CU can see it, then move the frontier
// over every satisfied pre-slot (whoever wins the CAS f -> f+1 has "reached" pre-slot f+1).  Runs in the EVENTS phase
// after the satisfy (a queued RF_LAZYWALK record), so the hot satisfy path carries one test and no call.  All shared
// words are read and written with returning device-scope atomics: waves on other XCDs run the same walk concurrently.
__device__ __forceinline__ void lazy_walk(const Dev &d, u32 edt, u32 slot) {
    EdtLine *L = &d.edt[edt];
    const u32 depc = L->depc;
    u32 *sl = depc <= INLINE_SLOTS ? L->slot : d.slots + L->slot[0];
    const u32 was = atomicOr(&sl[slot], SLOT_MARK);
    asm volatile("" ::"v"(was)); // the mark has landed before the frontier is read
    u64 *F = lazy_find(d, lazy_key(edt, LAZY_FRONT));
    if (!F) return;
    u64 f = atomicAdd((unsigned long long *)F, 0ull);
    while (f < depc) {
        if (!slot_satisfied(L, sl, depc, (u32)f)) break;
        if (atomicCAS((unsigned long long *)F, (unsigned long long)f, (unsigned long long)(f + 1)) == f && f + 1 < depc) {
            const u64 *xv = lazy_find(d, lazy_key(edt, (u32)f + 1));
            if (xv) lazy_reached(d, *xv);
        }
        f = atomicAdd((unsigned long long *)F, 0ull);
    }
}

// ---- Lockable datablocks: acquire modes on the ready path -----------------------------------
// An EDT whose pre-slots are all satisfied acquires its datablocks one by one in ascending order, each once, in the mode
// its ocrAddDependence named (taskAllDepvSatisfied / iterateDbFrontier, hc-task.c:662-705, 484-544), queues on the first
// busy one (lockableAcquireInternal, lockable-datablock.c:139-263) and becomes ready when it holds them all; the level
// that fires it releases them and the last user hands the datablock on (lockableRelease :330-507).
// No lock is needed around a datablock's record because the two sides never run in the same launch: acquisitions and
// enqueues happen in EVENTS launches (state only gets busier: a "busy" verdict read there stays true until the launch
// ends), releases and hand-overs in FIRE launches and in k_release_dbs (the last user owns the record outright).
__device__ __forceinline__ size_t acq_index(const Dev &d, u32 edt, u32 depc, u32 ovf_base, u32 slot) {
    return depc <= INLINE_SLOTS ? (size_t)edt * INLINE_SLOTS + slot : (size_t)d.max_edt * INLINE_SLOTS + ovf_base + slot;
}
// Who stands in a datablock's line: a list entry w <= max_edt is local EDT w - 1 (its link to the next entry sits in the upper half of
// its `val`), a larger one is remote requester rwait[w - max_edt - 1] (a partition that asked with an XF_ACQ record).
__device__ __forceinline__ u32 *waiter_link(const Dev &d, u32 w) {
    return w <= d.max_edt ? reinterpret_cast<u32 *>(&d.edt[w - 1u].val) + 1 : &d.ctrl->rwait[w - d.max_edt - 1u].next;
}
// The order in which an EDT takes its datablocks must be ONE order for everybody who can meet on them (sortRegNode, hc-task.c:459-477:
// ascending guid).  A datablock made by ocrgpu_db_create -- or a copy of one that came in a record -- goes by the guid it was created
// under, which is the same on every partition; the output datablock of an EDT is only ever contended where it lives and goes by its
// index there, behind all the others.
template <bool XL> __device__ __forceinline__ u64 acq_key(const Dev &d, u32 p /* index + 1 */) {
    if (!XL) return p - 1u < d.max_db ? (u64)p : (0x8000000000000000ull | p); // one partition: creation order IS ascending guid order (no trip to db_home[])
    if (p - 1u < d.max_db) { const u64 g = d.ctrl->db_home ? d.ctrl->db_home[p - 1u] : 0ull; return g ? (g & 0x7FFFFFFFFFFFFFFFull) : (u64)p; }
    return 0x8000000000000000ull | p;
}
// the partition that arbitrates datablock p (this one unless p is a copy of another partition's datablock)
__device__ __forceinline__ u64 acq_remote_home(const Dev &d, u32 p) {
    if (d.n_loc <= 1 || p - 1u >= d.max_db || !d.ctrl->db_home) return 0ull;
    const u64 g = d.ctrl->db_home[p - 1u];
    return (g && OCRGPU_GUID_LOCATION(g) != d.location) ? g : 0ull;
}
// true = acquired; false = OCR_EBUSY: queued on the list of its mode (push-front like the reference's lists).  `who` = list entry
__device__ __forceinline__ bool db_acquire(const Dev &d, u32 db, u32 mode, u32 who) {
    DbLock *B = &d.ctrl->dbl[db];
    u32 o = atomicAdd(&B->state, 0u);
    for (;;) {
        const u32 lock = o >> 30;
        u32 nw = o + 1u, *list = nullptr;                                                        // RO: falls through (:171)
        if (mode == OCRMODE_CONST) { if (lock) list = &B->ro; }                                   // :173-190
        else if (mode == OCRMODE_EW) { if (o) list = &B->ew; else nw = (DBL_EW << 30) | 1u; }      // :192-212
        else if (mode == OCRMODE_RW) {                                                            // :214-249 (one location)
            if (lock == DBL_ITW) { }
            else if (o) list = &B->itw;
            else nw = (DBL_ITW << 30) | 1u;
        }
        if (list) { *waiter_link(d, who) = atomicExch(list, who); return false; }
        const u32 prev = atomicCAS(&B->state, o, nw);
        if (prev == o) return true;
        o = prev;
    }
}
// iterateDbFrontier: go on from the last datablock taken (lower half of `val`: 1 + the pre-slot that named it); true = holds them all
// -> scheduleTask.  A datablock another partition arbitrates is asked for with a record (XF_ACQ); the EDT goes on when the grant arrives.
template <bool XL> // XL: datablocks may live on other partitions (compiled out of the one-partition kernels: tests/test_kernel_resources.py)
__device__ __forceinline__ bool acq_walk(const Dev &d, u32 e) {
    EdtLine *L = &d.edt[e];
    const u32 depc = L->depc;
    const u32 *sl = depc <= INLINE_SLOTS ? L->slot : d.slots + L->slot[0];
    const unsigned char *md = d.ctrl->acq_mode + acq_index(d, e, depc, L->slot[0], 0u);
    const u32 lk = (u32)L->val;
    u64 last = lk ? acq_key<XL>(d, sl[lk - 1u] & 0x7FFFFFFFu) : 0ull;
    for (;;) {
        u64 best = ~0ull; u32 bk = 0; // the next datablock in ascending order; named twice = the mode of its lowest pre-slot (stable sort :459-477)
        for (u32 k = 0; k < depc; ++k) { const u32 p = sl[k] & 0x7FFFFFFFu; if (!p) continue; const u64 key = acq_key<XL>(d, p); if (key > last && key < best) { best = key; bk = k; } }
        if (best == ~0ull) { L->val = 0; return true; }
        last = best;
        reinterpret_cast<u32 *>(&L->val)[0] = bk + 1u; // moved on before the attempt, like frontierSlot (:491)
        const u32 p = sl[bk] & 0x7FFFFFFFu;
        const u64 home = XL ? acq_remote_home(d, p) : 0ull;
        if (XL && home) { emit_remote_raw(d, home, md[bk], XF_ACQ, ((u64)d.location << 32) | e, 0ull); return false; } // its owner answers with XF_GRANT
        if (!db_acquire(d, p - 1u, md[bk], e + 1u)) return false;
    }
}
// lockableRelease: `push(w)` is called for every list entry the datablock is handed to (a local EDT goes on acquiring, a remote
// requester gets its grant)
template <class Push>
__device__ __forceinline__ void db_release(const Dev &d, u32 db, Push push) {
    DbLock *B = &d.ctrl->dbl[db];
    const u32 o = atomicSub(&B->state, 1u);
    if ((o & DBL_USERS) == 0u) { atomicAdd(&B->state, 1u); atomicAdd(&d.ctrl->err[ERR_BAD_OP], 1u); return; } // released more often than acquired
    if ((o & DBL_USERS) != 1u) return;
    // last user (:364-486): nobody else touches this record until the launch ends
    u32 w = 0, n = 0, lock = 0;
    if (o >> 30) w = atomicExch(&B->ro, 0u);                     // the writer(s) left: every reader in line (:375-382, 459-476)
    if (!w) {
        w = atomicExch(&B->itw, 0u);                             // every RW waiter of the location (:390-438)
        if (w) lock = DBL_ITW;
        else if ((w = atomicAdd(&B->ew, 0u)) != 0u) {            // one EW waiter (:393-395, 439-458)
            atomicExch(&B->ew, *waiter_link(d, w));
            push(w);
            atomicExch(&B->state, (DBL_EW << 30) | 1u);
            return;
        }
    }
    for (; w; w = *waiter_link(d, w)) { ++n; push(w); }
    atomicExch(&B->state, (lock << 30) | n);
}
// what `push` does with a remote list entry: the grant goes back to the partition that asked
__device__ __forceinline__ void grant_remote(const Dev &d, u32 w) {
    const RWait r = d.ctrl->rwait[w - d.max_edt - 1u];
    emit_remote_raw(d, mkguid(r.loc, OCRGPU_KIND_EDT, 1ull), 0u, XF_GRANT, (u64)r.edt, 0ull);
}
template <bool XL, class Push>
__device__ __forceinline__ void release_all(const Dev &d, const u32 *sl, u32 depc, Push push) { // the epilogue's release loop
    for (u32 k = 0; k < depc; ++k) {
        const u32 p = sl[k] & 0x7FFFFFFFu;
        if (!p) continue;
        const u64 key = acq_key<XL>(d, p);
        bool dup = false;
        for (u32 j = 0; j < k; ++j) { const u32 q = sl[j] & 0x7FFFFFFFu; dup |= q != 0u && acq_key<XL>(d, q) == key; }
        if (dup) continue;
        const u64 home = XL ? acq_remote_home(d, p) : 0ull;
        if (XL && home) emit_remote_raw(d, home, 0u, XF_REL, 0ull, 0ull); // its owner takes this holder off (lockableRelease runs there)
        else db_release(d, p - 1u, push);
    }
}

// event.satisfy for ONCE / STICKY / IDEM (hc-event.c:437-465, 569-639).  Returns true when the event
// fires its waiters with `payload`.  LATCH is handled wave-wide by latch_group().
__device__ __forceinline__ bool evt_satisfy_simple(const Dev &d, u32 x, u32 kind, u32 payload) {
    u32 *sk = evt_state(d, x);
    if (kind == OCREVT_ONCE) {
        u32 old = atomicExch(sk, ST_DESTROYED | (kind << 8));
        if ((old & 0xFF) != ST_LIVE) { atomicAdd(&d.ctrl->err[ERR_DEAD_EVENT], 1u); return false; }
        return true;
    }
    if (kind == OCREVT_COUNTED) return counted_satisfy(d, x, payload);
    u32 old = atomicCAS(sk, ST_LIVE | (kind << 8), ST_SAT | (kind << 8));
    if ((old & 0xFF) == ST_LIVE) { *evt_val(d, x) = (s32)payload; return true; }
    if ((old & 0xFF) == ST_SAT) { if (kind == OCREVT_STICKY) atomicAdd(&d.ctrl->err[ERR_STICKY_RESAT], 1u); }
    else atomicAdd(&d.ctrl->err[ERR_DEAD_EVENT], 1u);
    return false;
}

// satisfyEventHcChannel (hc-event.c:2125-2156): generation k of the channel.  True when waiter k is already in line
// (wpos = where it sits in waiters[]); else the payload queues for the waiter to come.
__device__ __forceinline__ bool chan_satisfy(const Dev &d, u32 x, u32 payload, u32 &wpos) {
    const ChanRec c = d.chan[(u32)*evt_val(d, x)];
    const u64 old = atomicAdd((unsigned long long *)&d.counted[x], 1ull << 32);
    const u32 k = (u32)(old >> 32), nw = (u32)old;
    if (k < nw) { wpos = c.wait_base + k % c.cap; return true; }
    if (k - nw >= c.cap) { atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u); return false; } // more than maxGen satisfies queued (ASSERT :2023)
    d.chan_sat[c.sat_base + k % c.cap] = payload;
    return false;
}

// satisfyEventHcLatch (hc-event.c:812-868) for every latch record held by this wave: lanes that hit the
// same latch are grouped (readfirstlane + ballot) and ONE atomic applies their summed +1/-1; the group
// fires iff the counter it leaves is 0 (well-formed programs never pass through 0 before their last op).
// A slot above 1 is OCRGPU_LATCH_DELTA(k): the net of several ops one caller made in a row (k may be 0).
// In two halves, so that a caller can put other memory operations between the counter atomics and the look at what they
// returned: latch_issue leaves, in the leader lane of each group, the old counter value and the group's delta.
__device__ __forceinline__ void latch_issue(const Dev &d, bool is_latch, u32 x, u32 slot, bool &lead, s32 &old_out, s32 &delta_out) {
    u64 todo = __ballot(is_latch);
    const u64 general = __ballot(is_latch && slot > 1u);
    while (todo) {
        int leader = __ffsll((long long)todo) - 1;
        u32 x0 = __shfl(x, leader);
        u64 grp = __ballot(is_latch && x == x0) & todo;
        u64 inc = __ballot(is_latch && x == x0 && slot == OCRLATCH_INCR) & todo;
        u64 gen = grp & general;
        s32 delta = (s32)__popcll(inc) - (s32)__popcll(grp & ~inc & ~gen);
        while (gen) { // wave-uniform, rare
            int b = __ffsll((long long)gen) - 1;
            delta += (s32)(__shfl(slot, b) - OCRGPU_LATCH_DELTA_BIAS);
            gen &= gen - 1;
        }
        if ((int)lane_id() == leader) { lead = true; delta_out = delta; old_out = atomicAdd(evt_val(d, x0), delta); } // no state read in front of it: one hop, not two
        todo &= ~grp;
    }
}
__device__ __forceinline__ bool latch_resolve(const Dev &d, bool lead, u32 x, s32 old, s32 delta) {
    bool fire = false;
    if (lead && old + delta == 0) {
        u32 was = atomicExch(evt_state(d, x), ST_DESTROYED | (OCREVT_LATCH << 8));
        if ((was & 0xFF) == ST_LIVE) fire = true; else atomicAdd(&d.ctrl->err[ERR_DEAD_EVENT], 1u);
    }
    return fire;
}
__device__ __forceinline__ bool latch_group(const Dev &d, bool is_latch, u32 x, u32 slot) {
    bool fire = false;
    u64 todo = __ballot(is_latch);
    const u64 general = __ballot(is_latch && slot > 1u);
    while (todo) {
        int leader = __ffsll((long long)todo) - 1;
        u32 x0 = __shfl(x, leader);
        u64 grp = __ballot(is_latch && x == x0) & todo;
        u64 inc = __ballot(is_latch && x == x0 && slot == OCRLATCH_INCR) & todo;
        u64 gen = grp & general;
        s32 delta = (s32)__popcll(inc) - (s32)__popcll(grp & ~inc & ~gen);
        while (gen) { // wave-uniform, rare
            int b = __ffsll((long long)gen) - 1;
            delta += (s32)(__shfl(slot, b) - OCRGPU_LATCH_DELTA_BIAS);
            gen &= gen - 1;
        }
        if ((int)lane_id() == leader) {
            s32 old = atomicAdd(evt_val(d, x0), delta); // no state read in front of it: one hop, not two
            if (old + delta == 0) {
                u32 was = atomicExch(evt_state(d, x0), ST_DESTROYED | (OCREVT_LATCH << 8));
                if ((was & 0xFF) == ST_LIVE) fire = true; else atomicAdd(&d.ctrl->err[ERR_DEAD_EVENT], 1u);
            }
        }
        todo &= ~grp;
    }
    return fire;
}

// ------------------------------------------------------------------------------------------------
// the hot kernel
// ------------------------------------------------------------------------------------------------
#define BLK 256
#define MAX_GRID 2048
#define HELPERS 256u         // fewest workgroups of a launch that run the bodies the last launch left
#define BIG_ROW 2048u        // waiter rows longer than this are split into pieces fanned out by separate workgroups
#define INVALID 0xFFFFFFFFu  // hole in the ready log (an edge that did not make its EDT ready)
#define RF_DST_LATCH 2u      // waiter flag when the destination is an event: it is a LATCH (no state read needed)

// Workgroup barrier for data that lives in LDS only: wait for this wave's LDS operations, then meet.  __syncthreads() also
// waits for every outstanding GLOBAL operation of the wave (vmcnt(0)): stores and atomics issued ahead of a barrier would
// then sit on the critical path of a level although nothing behind the barrier depends on them.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
struct Seg { u32 row, end_off, payload; }; // end_off = inclusive prefix sum of degrees

// value + tag of the datablock behind payload word p (p-1 = DB index): explicit DBs live in db[], the output
// DB of EDT e2 is {edt[e2].val, tag_base + edt[e2].param0} -- one 16-byte read of the producer's line
__device__ __forceinline__ ulonglong2 db_gather(const Dev &d, u32 p) {
    if (p == 0) return make_ulonglong2(0, ~0ULL);
    if (p - 1 < d.max_db) return d.db[p - 1];
    const ulonglong2 pv = *reinterpret_cast<const ulonglong2 *>(&d.edt[p - 1 - d.max_db].param0);
    return make_ulonglong2(pv.y, d.hash_tag_base + pv.x);
}

// queue a satisfy for another partition: per-destination send buffer, DB shipped by value
// AGG: the reservation is shared by the lanes of the wavefront (below).  Only the fan-out asks for it, and not in the tile-body
// instantiation of k_step (tile engines have one partition; the code inlined there cost it 70 VGPRs and a wave of occupancy).
template <bool AGG> __device__ __forceinline__ void emit_remote_record(const Dev &d, u64 guid, u32 slot, u32 xflags, u64 value, u64 tag);
// a record that carries tile datablock `dbi` (index in db[] / tiles[]): head + continuation slots, reserved together
__device__ void emit_remote_tile(const Dev &d, u64 guid, u32 slot, u32 xflags, ulonglong2 b, u32 dbi) {
    const u32 loc = OCRGPU_GUID_LOCATION(guid);
    if (d.send == nullptr || loc >= d.n_loc) { atomicAdd(&d.ctrl->err[ERR_BAD_OP], 1u); return; }
    const u32 pos = atomicAdd(&d.send_cnt[loc * 32], TILE_REC_SLOTS);
    if ((u64)pos + TILE_REC_SLOTS + 1 > d.send_cap) { atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u); return; } // (the last slot of a region is its header)
    SendRec *r = d.send + (size_t)loc * d.send_cap + pos;
    r[0] = SendRec{guid, slot, xflags | XF_HAS_DB | XF_HAS_TILE, b.x, b.y};
    const u64 *t = d.ctrl->tiles + OCROPS_TILE_WORDS * (size_t)dbi;
    const u64 home = d.ctrl->db_home[dbi];
    auto word = [&](u32 w) -> u64 { return w < OCROPS_TILE_WORDS ? t[w] : w == OCROPS_TILE_WORDS ? home : 0ull; };
#pragma unroll 11
    for (u32 k = 0; k + 1 < TILE_REC_SLOTS; ++k) r[1 + k] = SendRec{word(3 * k), k, XF_CONT, word(3 * k + 1), word(3 * k + 2)};
}
// XT: the caller may meet tile datablocks that cross partitions (compiled out of the k_step instantiations that cannot: the code costs
// the tile-body kernel a wave of occupancy and spills; tests/test_kernel_resources.py holds the one-partition kernels to what they had)
// a record that carries a datablock of a Lockable engine: the guid it was created under rides in ONE continuation slot (XF_HAS_HOME)
__device__ void emit_remote_home(const Dev &d, u64 guid, u32 slot, u32 xflags, ulonglong2 b, u32 dbi) {
    const u32 loc = OCRGPU_GUID_LOCATION(guid);
    if (d.send == nullptr || loc >= d.n_loc) { atomicAdd(&d.ctrl->err[ERR_BAD_OP], 1u); return; }
    const u32 pos = atomicAdd(&d.send_cnt[loc * 32], HOME_REC_SLOTS);
    if ((u64)pos + HOME_REC_SLOTS + 1 > d.send_cap) { atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u); return; }
    SendRec *r = d.send + (size_t)loc * d.send_cap + pos;
    r[0] = SendRec{guid, slot, xflags | XF_HAS_DB | XF_HAS_HOME, b.x, b.y};
    r[1] = SendRec{d.ctrl->db_home[dbi], 0u, XF_CONT, 0ull, 0ull};
}
template <bool AGG = false, bool XT = true>
__device__ __forceinline__ void emit_remote(const Dev &d, u64 guid, u32 slot, u32 xflags, u32 payload) {
    u32 loc = OCRGPU_GUID_LOCATION(guid);
    if (d.send == nullptr || loc >= d.n_loc) { atomicAdd(&d.ctrl->err[ERR_BAD_OP], 1u); return; }
    ulonglong2 b = make_ulonglong2(0, 0);
    if (payload && !(xflags & XF_MODE_NULL)) { b = db_gather(d, payload); xflags |= XF_HAS_DB; }
    if (XT && !AGG) { // (the wavefront-aggregated form is the fan-out of the engines WITHOUT tile datablocks)
        if ((xflags & XF_HAS_DB) && payload - 1 < d.max_db) {
            if (d.ctrl->tiles) { emit_remote_tile(d, guid, slot, xflags, b, payload - 1); return; }
            if (d.ctrl->dbl) { emit_remote_home(d, guid, slot, xflags, b, payload - 1); return; }
        }
    }
    emit_remote_record<AGG>(d, guid, slot, xflags, b.x, b.y);
}
__device__ void emit_remote_raw(const Dev &d, u64 guid, u32 slot, u32 xflags, u64 value, u64 tag) { emit_remote_record<false>(d, guid, slot, xflags, value, tag); }
template <bool AGG>
__device__ __forceinline__ void emit_remote_record(const Dev &d, u64 guid, u32 slot, u32 xflags, u64 value, u64 tag) {
    u32 loc = OCRGPU_GUID_LOCATION(guid);
    if (d.send == nullptr || loc >= d.n_loc) { atomicAdd(&d.ctrl->err[ERR_BAD_OP], 1u); return; }
    if (!AGG) {
        const u32 pos1 = atomicAdd(&d.send_cnt[loc * 32], 1u);
        if (pos1 + 1 >= d.send_cap) { atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u); return; } // (the last slot of a region is its header)
        d.send[(size_t)loc * d.send_cap + pos1] = SendRec{guid, slot, xflags, value, tag};
        return;
    }
    // The lanes of this wavefront that are here together with the same destination take ONE atomic on its counter (a level of the
    // bench's DAG queues ~5000 records per destination; one atomic each on a single word took 110 us of the launch), and the
    // atomics of the different destinations go out TOGETHER: the groups are found first (ballots, no memory), then the lowest
    // lane of every group reserves for it in one instruction -- one round trip, however many destinations the wavefront has.
    const u32 lane = __lane_id();
    u64 todo = __ballot(1), mine = 0;
    while (todo) {
        const u32 first = (u32)__ffsll((unsigned long long)todo) - 1u;
        const u64 same = __ballot(loc == __shfl(loc, first));
        if (same & (1ull << lane)) mine = same;
        todo &= ~same;
    }
    const u32 leader = (u32)__ffsll((unsigned long long)mine) - 1u;
    u32 base = 0;
    if (lane == leader) base = atomicAdd(&d.send_cnt[loc * 32], (u32)__popcll(mine));
    const u32 pos = __shfl(base, leader) + (u32)__popcll(mine & ((1ull << lane) - 1ull));
    if (pos + 1 >= d.send_cap) { atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u); return; } // (the last slot of a region is its header)
    d.send[(size_t)loc * d.send_cap + pos] = SendRec{guid, slot, xflags, value, tag};
}


// Fan `nseg` fired events out over the whole workgroup, edge-balanced: lane t takes edge t of the
// concatenated rows (binary search of the LDS prefix sums), so a 10k-waiter row and 63 one-waiter
// rows cost the same number of passes.  Ready EDTs are NOT compacted here: the workgroup reserves `total`
// consecutive entries of the ready log up front (one atomic, issued before the waiter loads so it costs no
// extra round trip) and lane t writes entry t: the EDT it made ready, or INVALID.  Compaction happens on
// the read side (the FIRE pass skips holes), where it is a ballot instead of a dependent atomic.
template <bool NANNY, int LOCKABLE, int TILES>
__device__ void fan_out(const Dev &d, const Step &S, Seg *seg, u32 nseg, u32 total, u32 *lds_base, u64 &n_edt_sat, u64 &n_evt_sat) {
    Ctrl *c = d.ctrl;
    u32 rbase = 0;
    if (threadIdx.x == 0) rbase = S.r_base + atomicAdd(S.cnt_r, total);
    for (u32 base = 0; base < total; base += BLK) {
        u32 t = base + threadIdx.x;
        bool act = t < total;
        bool ready = false, to_evt = false, is_latch = false;
        u32 dst = 0, slot = 0, payload = 0, flags = 0, kind = 0;
        if (act) {
            u32 lo = 0, hi = nseg - 1; // first segment with end_off > t
            while (lo < hi) { u32 mid = (lo + hi) >> 1; if (seg[mid].end_off > t) hi = mid; else lo = mid + 1; }
            u32 start = lo ? seg[lo - 1].end_off : 0;
            const u64 w = d.waiters[seg[lo].row + (t - start)];
            u32 whi = (u32)(w >> 32);
            dst = (u32)w; flags = whi >> 28; slot = whi & 0x0FFFFFFFu; payload = seg[lo].payload;
            if (!(flags & (RF_DST_EDT | RF_REMOTE))) {
                to_evt = true;
                if (flags & RF_DST_LATCH) { is_latch = true; kind = OCREVT_LATCH; }
                else kind = (*evt_state(d, dst) >> 8) & 0xFF;
            }
        }
        // LATCH destinations first (hc-event.c:812-868): their counter atomics -- every leaf of a finish scope hits the same
        // word, and same-address atomics queue up at ~10 ns each -- are in flight beside the pre-slot atomics below instead of
        // behind the ready-log stores; what they returned is looked at in the event block at the end of the pass
        bool l_lead = false; s32 l_old = 0, l_delta = 0;
        if (__ballot(is_latch)) latch_issue(d, is_latch, dst, slot, l_lead, l_old, l_delta);
        if (act) {
            if (flags & RF_DST_EDT) ready = edt_slot_satisfy<NANNY, LOCKABLE != 0>(d, S, dst, slot, payload, flags);
            else if (flags & RF_REMOTE) { RemoteDst rd = d.rdst[dst]; emit_remote<TILES == 0 && LOCKABLE != 2, TILES == 2 || LOCKABLE == 2>(d, rd.guid, rd.slot, rd.flags, payload); if (LOCKABLE != 2 && (S.mode & MODE_EARLY)) d.ctrl->nc_sends = 1u; } // (Lockable engines with partitions never run without settle rounds)
        }
        // statistics are wave totals (one ballot each), so the epilogue needs no cross-lane reduction
        n_edt_sat += __popcll(__ballot(act && (flags & RF_DST_EDT))); n_evt_sat += __popcll(__ballot(to_evt));
        if (base == 0) { if (threadIdx.x == 0) *lds_base = rbase; lds_barrier(); rbase = *lds_base; }
        if (act) { if (rbase + t < d.fired_cap) d.fired[rbase + t] = ready ? dst : INVALID; else atomicAdd(&c->err[ERR_Q_OVERFLOW], 1u); }
        // event -> event dependences: run the destination's satisfy logic right here (commonSatisfyRegNode
        // recursing into processMessage, hc-event.c:335-367); only an event that FIRES costs a queue record
        if (__ballot(to_evt)) {
            bool fire = latch_resolve(d, l_lead, dst, l_old, l_delta);
            u32 cw = 0; bool paired = false;
            if (to_evt && !is_latch) {
                if (kind == OCREVT_ONCE || kind == OCREVT_STICKY || kind == OCREVT_IDEM || kind == OCREVT_COUNTED) fire = evt_satisfy_simple(d, dst, kind, payload);
                else if (kind == OCREVT_CHANNEL && (*evt_state(d, dst) & 0xFF) == ST_LIVE) paired = chan_satisfy(d, dst, payload, cw);
                else atomicAdd(&c->err[ERR_DEAD_EVENT], 1u);
            }
            bool toq = (fire && *evt_deg(d, dst) != 0) || paired;
            u32 qi = S.q_base + wave_append(S.cnt_q, toq);
            if (toq) {
                if (qi < d.q_cap) d.q[qi] = paired ? Rec{dst, payload, cw, RF_FIRED | RF_CHANGEN} : Rec{dst, payload, 0u, RF_FIRED};
                else atomicAdd(&c->err[ERR_Q_OVERFLOW], 1u);
            }
        }
    }
    // no barrier here: the next use of seg[] / lds_base comes after the barriers of the caller's next block scan, which no
    // wave passes before every wave has left this function -- and a barrier's workgroup fence would hold each wave until its
    // ready-log stores have landed
}

// A fired event whose row is longer than BIG_ROW is not expanded by one workgroup: queue one RF_PIECE record
// per BIG_ROW edges; the next EVENTS phase spreads the pieces over the grid.
__device__ __forceinline__ bool split_big_row(const Dev &d, const Step &S, u32 evt, u32 deg, u32 payload) {
    if (deg <= BIG_ROW) return false;
    u32 np = (deg + BIG_ROW - 1) / BIG_ROW;
    u32 qi = S.q_base + atomicAdd(S.cnt_q, np);
    for (u32 k = 0; k < np; ++k) {
        if (qi + k < d.q_cap) d.q[qi + k] = Rec{evt, payload, k, RF_FIRED | RF_PIECE};
        else { atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u); break; }
    }
    return true;
}

// block-wide inclusive scan of one u32 per thread (BLK = 256 = 4 waves)
__device__ __forceinline__ u32 block_inclusive_scan(u32 v, u32 *wave_sums, u32 &total) {
    u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) { u32 n = __shfl_up(v, o); if (lane >= (u32)o) v += n; }
    if (lane == 63) wave_sums[wid] = v;
    lds_barrier();
    u32 add = 0;
    for (u32 w = 0; w < wid; ++w) add += wave_sums[w];
    total = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
    lds_barrier();
    return v + add;
}

// the body proper: fold id + every depv[k] payload (8 pre-slots per round of loads; gb = the first round, already loaded)
__device__ __forceinline__ void run_body(const Dev &d, u32 e, u64 id, u32 depc, const u32 *sl, ulonglong2 (&gb)[8], u32 body) {
    u64 v = ocrops_hash_init(id);
    for (u32 k0 = 0; k0 < depc; k0 += 8) {
        if (k0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) gb[k] = db_gather(d, k0 + k < depc ? (sl[k0 + k] & 0x7FFFFFFFu) : 0u);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k0 + k < depc) v = ocrops_hash_step(v, k0 + k, gb[k].x, gb[k].y + 1);
    }
    d.edt[e].val = v; // == the output DB's value (a live DB only for HASH bodies)
    d.edt[e].body = body | BODY_VAL_DONE;
}

// OCRBODY_TILE (8x8 int64 tiles updated in place by integer recurrences): the pre-slots that carry a datablock
// name 1..3 tiles; the last is updated, C -= X * Y^T.  tile_operands picks them from the pre-slot words (t[k] = index + 1, 0 = none).
__device__ __forceinline__ void tile_operands(const u32 *pw, u32 depc, u32 (&t)[3]) {
    t[0] = t[1] = t[2] = 0;
#pragma unroll
    for (u32 k = 0; k < INLINE_SLOTS; ++k) {
        const u32 p = k < depc ? pw[k] : 0u;
        if (p) { if (!t[0]) t[0] = p; else if (!t[1]) t[1] = p; else if (!t[2]) t[2] = p; }
    }
}
__device__ __forceinline__ u32 tile_target(const u32 (&t)[3]) { return t[2] ? t[2] : t[1] ? t[1] : t[0]; }
// A wave runs the TILE bodies of its lanes eight at a time: lanes 8g..8g+7 take the g-th pending body, lane r of the group
// computes row r of the tile -- its rows of C and X (one 64-byte sector each) and all of Y (eight sectors every lane of the
// group reads alike), 64 multiply-subtracts, one 64-byte store; the checksum is reduced over the group's eight lanes.
// (One body at a time with a lane per element made every body wait for its own loads: 64 dependent round trips per wave,
// 30.7 ms per resolve of the T = 256 Cholesky; eight at a time: see DESIGN.md s7.)  Every lane's operand loads are consumed
// before the store of any lane is issued (one instruction stream), so operands that alias the target (potrf: C -= C * C^T)
// are read whole.  Called in wave-uniform control flow.
__device__ void tile_bodies_wave(const Dev &d, bool has, u32 e, const u32 (&t)[3]) {
    u64 m = __ballot(has);
    if (!m) return;
    u64 *const tiles = d.ctrl->tiles;
    const u32 lane = lane_id(), g = lane >> 3, r = lane & 7u;
    while (m) {
        int src = -1; // the g-th lowest pending lane
        { u64 mm = m; for (u32 k = 0; k <= g && mm; ++k) { if (k == g) src = __ffsll((long long)mm) - 1; mm &= mm - 1; } }
        for (int k = 0; k < 8 && m; ++k) m &= m - 1;
        const int from = src < 0 ? 0 : src;
        const u32 ee = __shfl(e, from), a = __shfl(t[0], from), b = __shfl(t[1], from), cc = __shfl(t[2], from);
        if (src < 0) continue; // (no wave-wide operation below this line except within the 8-lane group)
        const u32 tgt = cc ? cc : b ? b : a, xi = b ? a : tgt, yi = cc ? b : xi;
        u64 sum = 0;
        if (tiles && tgt && tgt - 1 < d.max_db && xi - 1 < d.max_db && yi - 1 < d.max_db) {
            ulonglong2 *Crow = reinterpret_cast<ulonglong2 *>(tiles + 64 * (size_t)(tgt - 1) + 8 * r);
            const ulonglong2 *Xrow = reinterpret_cast<const ulonglong2 *>(tiles + 64 * (size_t)(xi - 1) + 8 * r);
            const ulonglong2 *Y = reinterpret_cast<const ulonglong2 *>(tiles + 64 * (size_t)(yi - 1));
            u64 acc[8], x[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const ulonglong2 cv = Crow[q], xv = Xrow[q]; acc[2 * q] = cv.x; acc[2 * q + 1] = cv.y; x[2 * q] = xv.x; x[2 * q + 1] = xv.y; }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { const ulonglong2 yv = Y[4 * c + q]; acc[c] -= x[2 * q] * yv.x + x[2 * q + 1] * yv.y; }
            }
            // (all of the group's -- and the wave's -- loads above have returned before the first store below is issued)
#pragma unroll
            for (int q = 0; q < 4; ++q) Crow[q] = make_ulonglong2(acc[2 * q], acc[2 * q + 1]);
#pragma unroll
            for (int c = 0; c < 8; ++c) sum += ocrops_tile_term(acc[c], 8 * r + c);
        } else if (tgt && r == 0) atomicAdd(&d.ctrl->err[ERR_BAD_OP], 1u); // a TILE body on something that is not a tile datablock
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
        if (r == 0) d.edt[ee].val = sum;
    }
}

// the per-lane part of a deferred body: the EDT's line, its pre-slot words, and (unless it is a TILE body, which the whole wave
// runs afterwards) the gathers and the fold
template <int TILES>
__device__ __forceinline__ void deferred_body(const Dev &d, u32 e, bool &tile, u32 (&tt)[3]) {
    const EdtLine L = d.edt[e];
    if (OCRBODY_KIND(L.body) == OCRBODY_HOST) return;
    const u32 *sl; u32 pw[8]; ulonglong2 gb[8];
    if (L.depc <= INLINE_SLOTS) {
#pragma unroll
        for (int k = 0; k < (int)INLINE_SLOTS; ++k) pw[k] = L.slot[k] & 0x7FFFFFFFu;
        pw[5] = pw[6] = pw[7] = 0; sl = d.edt[e].slot;
    } else {
        sl = d.slots + L.slot[0];
#pragma unroll
        for (int k = 0; k < 8; ++k) pw[k] = (u32)k < L.depc ? (sl[k] & 0x7FFFFFFFu) : 0u;
    }
    if (TILES && OCRBODY_KIND(L.body) == OCRBODY_TILE) { tile = true; tile_operands(pw, L.depc, tt); return; }
#pragma unroll
    for (int k = 0; k < 8; ++k) gb[k] = db_gather(d, pw[k]);
    run_body(d, e, L.param0, L.depc, sl, gb, L.body);
}
template <int TILES>
__device__ void deferred_bodies(const Dev &d, u32 b_begin, u32 b_end, u32 h, u32 nh) {
    bool tile = false; u32 tt[3] = {0, 0, 0};
    if (!TILES) { // every lane for itself
        for (u32 i = b_begin + h * BLK + threadIdx.x; i < b_end; i += nh * BLK) {
            const u32 e = d.fired[i];
            if (e != INVALID) deferred_body<false>(d, e, tile, tt);
        }
        return;
    }
    // A wave runs eight tile bodies per pass and every pass is a round trip to the operands: with 64 entries per wave the owed range
    // is eight dependent passes behind the two trips to the entry and its line, longer than the level it runs beside.  So the range is
    // dealt as thinly as the helpers allow, down to 8 entries (one pass) per wave: the T = 256 Cholesky's levels average 3 700 tasks,
    // which 2 048 workgroups take in one pass.
    u32 per_wave = 64;
    while (per_wave > 8u && (u64)nh * (per_wave / 2u) * (BLK / 64u) >= (u64)(b_end - b_begin)) per_wave >>= 1;
    const u32 per_wg = per_wave * (BLK / 64u), wv = threadIdx.x >> 6, ln = threadIdx.x & 63u;
    for (u32 base = b_begin + h * per_wg; base < b_end; base += nh * per_wg) { // workgroup-uniform trip count: tile bodies are wave-wide
        const u32 i = base + wv * per_wave + ln;
        const u32 e = (ln < per_wave && i < b_end) ? d.fired[i] : INVALID;
        tile = false;
        if (e != INVALID) deferred_body<true>(d, e, tile, tt);
        tile_bodies_wave(d, tile, e, tt);
    }
}

#ifdef OCRGPU_STAMPS
#define STAMP_LAUNCHES 8u // consecutive launches K0 .. K0+7 keep a per-workgroup table each
#define STAMP_AT(j) d.dbg[16 + (size_t)(K - (u32)d.dbg[15]) * 8 * MAX_GRID + 8 * blockIdx.x + (j)]
#define STAMP(k) do { if (threadIdx.x == 0) { u64 _t = __builtin_amdgcn_s_memrealtime(); if (blockIdx.x == 0) { d.dbg[k] += _t - _t0; _t0 = _t; } \
                                              if (K - (u32)d.dbg[15] < STAMP_LAUNCHES) STAMP_AT(2 + (k)) = _t; } } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// One launch = one phase.  K is the launch index (host counter): it selects the state / counter buffers.
// TILES (OCRGPU_F_WIDE_DB): OCRBODY_TILE bodies are compiled in -- a variant of its own, so that the default kernel keeps
// its registers (118 VGPRs, 4 waves per SIMD; with the tile code inline: 137 and 3).
template <bool NANNY, int LOCKABLE, int TILES>
__global__ __launch_bounds__(BLK) void k_step(Dev d, u32 K) {
#ifdef OCRGPU_STAMPS
    u64 _t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && K - (u32)d.dbg[15] < STAMP_LAUNCHES) // entry time; the XCD this workgroup runs on in the top 4 bits
        STAMP_AT(0) = _t0 | ((u64)(__builtin_amdgcn_s_getreg(((32 - 1) << 11) | 20) & 0xFu) << 60);
#endif
    __shared__ Seg seg[BLK];
    __shared__ u32 wave_sums[4];
    __shared__ u32 lds_base;
    Ctrl *c = d.ctrl;
    // ---- replicated phase decision (the scheduler's "get work", on the device, without a hand-over) ----
    const State P = c->state[(K - 1) & 1];
    if (P.done) { if (blockIdx.x == 0 && threadIdx.x == 0) { c->state[K & 1] = P; atomicAdd(&c->noop_steps, 1u); } return; }
    const u32 r_total = P.r_total + c->cnt_r[(K - 1) % 3].v;
    u32 q_total = P.q_total + c->cnt_q[(K - 1) % 3].v;
    if (q_total > d.q_cap) q_total = d.q_cap;
    State N = P;
    N.r_total = r_total; N.q_total = q_total;
    // work split of a list of `n` entries: entries per workgroup (`per`) and workgroups with work (`part`), see below
    auto split = [&](u32 n, u32 &per_, u32 &part_) {
        per_ = (n + d.spread_wgs - 1) / d.spread_wgs;
        if (per_ < 1) per_ = 1;
        if (per_ > BLK) per_ = BLK;
        if (per_ >= 16u) per_ = (per_ + 15u) & ~15u; // whole 16-entry (64-B) chunks, see the FIRE loop
        part_ = (n + per_ - 1) / per_; if (part_ > gridDim.x) part_ = gridDim.x; // 0 for a forced empty level
    };
    const bool owed = P.b_end > P.b_begin; // the last launch fired a level and left its bodies to this one
    bool flush_only = false;
    if (owed && !(q_total > P.q_end) && P.allow_fire && r_total > P.r_end) {
        // a WIDE level is about to fire behind owed bodies: its own bodies cannot wait for idle workgroups (there are few)
        // and must not run beside the owed ones (they read their values), so this launch only pays the debt
        u32 pr, pt; split(r_total - P.r_end, pr, pt);
        flush_only = pt > gridDim.x / 2;
    }
    if (flush_only) { N.mode = MODE_EVENTS; N.q_begin = N.q_end = P.q_end; }
    else if (q_total > P.q_end) { N.mode = MODE_EVENTS; N.q_begin = P.q_end; N.q_end = q_total; }
    else if (P.allow_fire && r_total > P.r_end) {
        if (P.allow_fire != 0xFFFFFFFFu) N.allow_fire = P.allow_fire - 1; // FIRE budget (distributed stepping: 1 level)
        N.mode = MODE_FIRE; N.level = P.level + 1; N.r_begin = P.r_end; N.r_end = r_total; N.n_levels = N.level + 1;
    } else if (P.allow_fire && P.force_level) {
        // nothing ready here, but the partitions step in lockstep: count one EMPTY level so numbering stays aligned
        if (P.allow_fire != 0xFFFFFFFFu) N.allow_fire = P.allow_fire - 1;
        N.mode = MODE_FIRE; N.level = P.level + 1; N.r_begin = N.r_end = P.r_end; N.n_levels = N.level + 1;
    } else N.done = 1;
    // Entries per workgroup per pass: spread a short list over the whole grid (a level of 255 EDTs with 255-waiter
    // rows, or 512 pieces of a 1M-waiter row, must not land on one or two workgroups); BLK once the list is long.
    // The spread stops at d.spread_wgs workgroups: every workgroup reserves its ready-log range with one atomic on
    // the same counter, and same-address atomics serialise (~10 ns each: 2000 of them cost more than they save).
    const u32 count = N.done ? 0u : (N.mode == MODE_FIRE) ? N.r_end - N.r_begin : N.q_end - N.q_begin;
    u32 per, part; split(count, per, part);
    // Bodies off the critical path: a FIRE launch on a single partition fans its level out and leaves the bodies (gather
    // the depv datablocks, fold, store the result) to the NEXT launch, where the workgroups that have no work of their own
    // run them (`deferred_bodies`).  Nothing on the dependence path waits for a body's value: consumers' bodies read it at
    // the earliest two launches later.  A level is deferred when it leaves at least half the grid idle (a narrow level is
    // most likely followed by another); a wide level runs its bodies itself, after a launch that only pays what is owed.
    const bool defer = !N.done && N.mode == MODE_FIRE && d.n_loc <= 1 && part <= gridDim.x / 2;
    N.b_begin = defer ? N.r_begin : 0u; N.b_end = defer ? N.r_end : 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        c->state[K & 1] = N;
        c->cnt_r[(K + 1) % 3].v = 0; c->cnt_q[(K + 1) % 3].v = 0;
        if (!N.done && N.mode == MODE_FIRE) {
            if (N.level < d.max_levels) { d.level_start[N.level] = N.r_begin; d.level_start[N.level + 1] = N.r_end; }
            else atomicAdd(&c->err[ERR_LEVEL_OVERFLOW], 1u);
        }
    }
    if (owed && part < gridDim.x) { // the idle workgroups of this launch run the bodies the last one left
        // one pass over the owed range when there are enough idle workgroups for it, never fewer than HELPERS of them
        // (64 helpers looping over a 40 000-entry range take longer than the level beside them: 15.7 instead of 11.9 us);
        // the rest of the idle workgroups leave at once
        u32 n_help = (P.b_end - P.b_begin + BLK - 1) / BLK;
        if (TILES) n_help = (P.b_end - P.b_begin + 8u * (BLK / 64u) - 1) / (8u * (BLK / 64u)); // (tile bodies: one pass of eight per wave where the helpers allow, deferred_bodies)
        if (n_help < HELPERS) n_help = HELPERS;
        if (n_help > gridDim.x - part) n_help = gridDim.x - part;
        if (blockIdx.x >= part && blockIdx.x < part + n_help) deferred_bodies<TILES>(d, P.b_begin, P.b_end, blockIdx.x - part, n_help);
    }
    if (N.done) { if (blockIdx.x == 0 && threadIdx.x == 0 && !owed) atomicAdd(&c->noop_steps, 1u); return; } // (a launch that only notices the end)
    Step S;
    S.mode = N.mode | ((LOCKABLE != 2 && N.mode == MODE_EVENTS && N.allow_fire != 0u && d.n_loc > 1) ? MODE_EARLY : 0u); S.r_begin = N.r_begin; S.r_end = N.r_end; S.q_begin = N.q_begin; S.q_end = N.q_end;
    S.r_base = r_total; S.q_base = q_total; S.cnt_r = &c->cnt_r[K % 3].v; S.cnt_q = &c->cnt_q[K % 3].v;
    u64 n_edt_sat = 0, n_evt_sat = 0, n_fired = 0;
    if (blockIdx.x >= part) return; // workgroups beyond the work leave at once
    STAMP(0);

    const bool body_first = d.n_loc > 1;
    if (S.mode == MODE_FIRE) {
        // ---- taskExecute + taskEpilogue for one level (hc-task.c:1864-2181, 1707-1862) ----
        // The ready log is sparse and its valid entries are not spread evenly: an EDT is made ready by the LAST of its
        // edges, and late edges are appended late, so the holes sit at the front of a level's range and the EDTs at
        // its end.  Each workgroup therefore takes its entries as 16-entry chunks dealt round-robin over the whole
        // range (chunk c of workgroup b = chunk c*W + b of the pass), not as one contiguous run.
        // Only when there are few workgroups per CU: a wide level has many more workgroups than CUs, which evens the load
        // out by itself, and keeps its contiguous (better coalesced) runs.
        const u32 CH = (per >= 16u && part <= 512u) ? 16u : per;
        const u32 pass_len = gridDim.x * per;
        for (u32 pb = S.r_begin; pb < S.r_end; pb += pass_len) {
            const u32 left = S.r_end - pb;
            const u32 W = left >= pass_len ? gridDim.x : (left + per - 1) / per; // workgroups working in this pass
            if (blockIdx.x >= W) break;
            u32 deg = 0; Seg sg = {0, 0, 0};
            u32 i = S.r_end;
            if (threadIdx.x < per) {
                const u32 ii = pb + ((threadIdx.x / CH) * W + blockIdx.x) * CH + threadIdx.x % CH;
                if (ii < S.r_end) i = ii;
            }
            // Critical path per level: ready-log entry -> the EDT's 64-B line -> waiter row -> atomicSub -> log store.
            // The body's gathers (the DBs named by the pre-slot words, which sit in the same line) are issued
            // BEFORE the fan-out and consumed after it, so they ride under it.
            u32 e = INVALID;
            if (i < S.r_end) e = d.fired[i];
            const bool mine = e != INVALID;
            bool epi = false;
            u32 body = 0, depc = 0, pw[8]; u64 id = 0; const u32 *sl = nullptr; ulonglong2 gb[8]; u32 tt[3] = {0, 0, 0};
            if (mine) {
                const EdtLine L = d.edt[e]; // the one 64-B read that carries everything this EDT needs to fire
                body = L.body; depc = L.depc; id = L.param0;
                const u32 x = d.max_evt + e;
                if (depc <= INLINE_SLOTS) {
#pragma unroll
                    for (int k = 0; k < (int)INLINE_SLOTS; ++k) pw[k] = L.slot[k] & 0x7FFFFFFFu;
                    pw[5] = pw[6] = pw[7] = 0; sl = d.edt[e].slot;
                } else {
                    sl = d.slots + L.slot[0];
#pragma unroll
                    for (int k = 0; k < 8; ++k) pw[k] = (u32)k < depc ? (sl[k] & 0x7FFFFFFFu) : 0u;
                }
                if (TILES != 2 && !defer && !(TILES && OCRBODY_KIND(body) == OCRBODY_TILE)) { // (TILES == 2: the bodies of this level ran in k_bodies_pre)
#pragma unroll
                    for (int k = 0; k < 8; ++k) gb[k] = db_gather(d, pw[k]);
                }
                u32 ret = 0;
                const u32 kind = OCRBODY_KIND(body);
                if (kind == OCRBODY_HASH) ret = d.max_db + e + 1;
                else if (kind == OCRBODY_FWD0) ret = pw[0];
                else if (TILES && kind == OCRBODY_TILE) { tile_operands(pw, depc, tt); ret = tile_target(tt); } // the tile it updates travels on
                // With other partitions around, a satisfy may carry this EDT's output DB away by value in this very
                // launch, so the body runs first; on a single partition it runs after the fan-out (off the critical path).
                const bool host_body = kind == OCRBODY_HOST; // taskExecute happens on the host (ocrgpu_pop_ready)
                if (TILES != 2 && body_first && !host_body && kind != OCRBODY_TILE) run_body(d, e, id, depc, sl, gb, body); // (tile engines with partitions: k_bodies_pre)
                if (body & BODY_F_SCRIPT) { // the body's own ocrEventSatisfySlot / ocrEventDestroy calls
                    u32 sb = d.edt_script_begin[e], sn = d.edt_script_len[e];
                    for (u32 k = 0; k < sn; ++k) {
                        Script sc = d.scripts[sb + k];
                        u32 op = sc.op_kind & 0xFF, pk = (sc.op_kind >> 8) & 0xFF;
                        if (op == SOP_MERGED) continue; // a latch op folded into an earlier one (k_script_merge_latch)
                        if (op == OCRGPU_SOP_SATISFY) {
                            u32 pay = pk == OCRGPU_PAY_GUID ? sc.payload : pk == OCRGPU_PAY_OUTDB ? d.max_db + e + 1
                                    : pk == OCRGPU_PAY_DEPV ? (sl[sc.payload] & 0x7FFFFFFFu) : 0u;
                            if (sc.op_kind & 0x10000u) { // the event lives on another partition
                                RemoteDst rd = d.rdst[sc.target]; emit_remote<false, TILES == 2 || LOCKABLE == 2>(d, rd.guid, rd.slot, 0u, pay); continue;
                            }
                            u32 qi = S.q_base + atomicAdd(S.cnt_q, 1u);
                            if (qi < d.q_cap) d.q[qi] = Rec{sc.target, pay, sc.slot, 0u};
                            else atomicAdd(&c->err[ERR_Q_OVERFLOW], 1u);
                        } else if (sc.op_kind & 0x10000u) { // ocrEventDestroy of a remote event
                            RemoteDst rd = d.rdst[sc.target]; emit_remote<false, false>(d, rd.guid, 0u, XF_DESTROY, 0u);
                        } else { // ocrEventDestroy
                            u32 old = atomicExch(evt_state(d, sc.target), (u32)ST_DESTROYED);
                            if ((old & 0xFF) != ST_LIVE && (old & 0xFF) != ST_SAT) atomicAdd(&c->err[ERR_DEAD_EVENT], 1u);
                        }
                    }
                }
                if (LOCKABLE && depc && !host_body) // Lockable datablocks: the body has run, its datablocks go back (host bodies: ocrgpu_edt_release_dbs)
                    release_all<LOCKABLE == 2>(d, sl, depc, [&](u32 w) {
                        if (LOCKABLE == 2 && w > d.max_edt) { grant_remote(d, w); return; } // a requester on another partition is next in line
                        const u32 qi = S.q_base + atomicAdd(S.cnt_q, 1u);
                        if (qi < d.q_cap) d.q[qi] = Rec{w - 1u, 0u, 0u, RF_ACQUIRE}; else atomicAdd(&c->err[ERR_Q_OVERFLOW], 1u);
                    });
                if ((body & OCRBODY_F_OUTEVT) && !host_body) { // epilogue: satisfy the ONCE output event with the returned guid
                    epi = true;
                    bool ok = true;
                    if (NANNY) ok = evt_satisfy_simple(d, x, OCREVT_ONCE, ret);
                    else d.edt[e].out_state = ST_DESTROYED | (OCREVT_ONCE << 8); // only its own EDT satisfies it
                    if (ok && !split_big_row(d, S, x, L.out_deg, ret)) { deg = L.out_deg; sg.row = L.out_row; sg.payload = ret; }
                }
            }
            n_fired += __popcll(__ballot(mine)); n_evt_sat += __popcll(__ballot(epi));
            STAMP(1);
            u32 total;
            u32 incl = block_inclusive_scan(deg, wave_sums, total);
            sg.end_off = incl;
            seg[threadIdx.x] = sg;
            lds_barrier();
            STAMP(2);
            if (total) fan_out<NANNY, LOCKABLE, TILES>(d, S, seg, BLK, total, &lds_base, n_edt_sat, n_evt_sat);
            STAMP(3);
            if (TILES != 2 && mine && !body_first && !defer && OCRBODY_KIND(body) != OCRBODY_HOST && OCRBODY_KIND(body) != OCRBODY_TILE) run_body(d, e, id, depc, sl, gb, body);
            if (TILES == 1) tile_bodies_wave(d, mine && !defer && OCRBODY_KIND(body) == OCRBODY_TILE, e, tt);
        }
    } else {
        // ---- queued satisfies: API calls, fired chained events, big-row pieces, direct EDT pre-slot satisfies ----
        for (u32 cb = S.q_begin + blockIdx.x * per; cb < S.q_end; cb += gridDim.x * per) {
            u32 deg = 0; Seg sg = {0, 0, 0};
            u32 i = threadIdx.x < per ? cb + threadIdx.x : S.q_end;
            bool act = i < S.q_end;
            Rec r = {0, 0, 0, 0};
            bool ready = false, is_latch = false, cnt_edt = false, cnt_evt = false;
            u32 kind = 0;
            if (act) {
                r = d.q[i];
                if (r.flags & RF_LAZYWALK) lazy_walk(d, r.dst, r.slot);
                else if (LOCKABLE && (r.flags & RF_ACQUIRE)) ready = acq_walk<LOCKABLE == 2>(d, r.dst);
                else if (r.flags & RF_DST_EDT) { ready = edt_slot_satisfy<NANNY, LOCKABLE != 0>(d, S, r.dst, r.slot, r.payload, r.flags); cnt_edt = true; }
                else if (!(r.flags & RF_FIRED)) { kind = (*evt_state(d, r.dst) >> 8) & 0xFF; is_latch = (kind == OCREVT_LATCH); cnt_evt = true; }
            }
            n_edt_sat += __popcll(__ballot(cnt_edt)); n_evt_sat += __popcll(__ballot(cnt_evt));
            u32 ri = S.r_base + wave_append(S.cnt_r, ready);
            if (ready) d.fired[ri] = r.dst;
            bool fire = latch_group(d, is_latch, r.dst, r.slot);
            if (act && (r.flags & RF_CHANGEN)) { deg = 1; sg.row = r.slot; sg.payload = r.payload; } // a CHANNEL pairing made in a fan-out
            else if (act && (r.flags & RF_FIRED)) fire = true;
            else if (act && !(r.flags & (RF_DST_EDT | RF_LAZYWALK | RF_ACQUIRE)) && !is_latch) {
                if (kind == OCREVT_ONCE || kind == OCREVT_STICKY || kind == OCREVT_IDEM || kind == OCREVT_COUNTED) fire = evt_satisfy_simple(d, r.dst, kind, r.payload);
                else if (kind == OCREVT_CHANNEL && (*evt_state(d, r.dst) & 0xFF) == ST_LIVE) {
                    u32 wp; if (chan_satisfy(d, r.dst, r.payload, wp)) { deg = 1; sg.row = wp; sg.payload = r.payload; } // the waiter in line takes it now
                }
                else atomicAdd(&c->err[(*evt_state(d, r.dst) & 0xFF) == ST_ABSENT ? ERR_BAD_OP : ERR_DEAD_EVENT], 1u);
            }
            if (fire) {
                struct { u32 row_begin, deg; } st = {*evt_row(d, r.dst), *evt_deg(d, r.dst)};
                if (r.flags & RF_PIECE) { u32 off = r.slot * BIG_ROW; deg = st.deg - off < BIG_ROW ? st.deg - off : BIG_ROW; sg.row = st.row_begin + off; sg.payload = r.payload; }
                else if (!split_big_row(d, S, r.dst, st.deg, r.payload)) { deg = st.deg; sg.row = st.row_begin; sg.payload = r.payload; }
            }
            u32 total;
            u32 incl = block_inclusive_scan(deg, wave_sums, total);
            sg.end_off = incl;
            seg[threadIdx.x] = sg;
            lds_barrier();
            if (total) fan_out<NANNY, LOCKABLE, TILES>(d, S, seg, BLK, total, &lds_base, n_edt_sat, n_evt_sat);
        }
    }
    if (owed && part == gridDim.x) deferred_bodies<TILES>(d, P.b_begin, P.b_end, blockIdx.x, gridDim.x); // no idle workgroup in this launch
    STAMP(4);
    // ---- statistics: the counters are wave totals (ballots where they are counted) -> this workgroup's own slot, one
    // fire-and-forget atomic per wave and counter.  No shuffles, no LDS stage and no barrier: a barrier's workgroup fence would make every wave wait here for its fan-out stores to land ----
    if ((threadIdx.x & 63) == 0 && (n_edt_sat | n_evt_sat | n_fired)) { // wave totals already
        unsigned long long *bs = (unsigned long long *)(d.blk_stats + 4 * blockIdx.x);
        if (n_edt_sat) atomicAdd(bs, (unsigned long long)n_edt_sat);
        if (n_evt_sat) atomicAdd(bs + 1, (unsigned long long)n_evt_sat);
        if (n_fired) atomicAdd(bs + 2, (unsigned long long)n_fired);
    }
    STAMP(5);
#ifdef OCRGPU_STAMPS
    if (threadIdx.x == 0 && K - (u32)d.dbg[15] < STAMP_LAUNCHES) STAMP_AT(1) = __builtin_amdgcn_s_memrealtime();
#endif
}

// Tile engines WITH partitions: the fan-out of a level carries its tiles away BY VALUE (records), so the bodies of a level run before
// it -- in a launch of their own in front of the FIRE launch.  (Inline in k_step, in front of the fan-out, the tile code took the
// kernel to 249 VGPRs and 312 bytes of scratch: 61 us per launch, 22 for one that finds nothing to do; T = 256 on 8 partitions of
// one GPU 1.3 s per resolve.)  The launch decides what launch K will do exactly as k_step(K) does -- from launch K - 1's state and
// append counters, which nothing writes in between -- and, when that is a level, runs its bodies dealt 8 entries per wave (one pass
// of tile_bodies_wave) where the grid allows.  A body that updated a COPY of another partition's datablock sends the contents home
// (hc-dist writes a released RW / EW proxy back, hc-dist-policy.c:382-700).
__global__ __launch_bounds__(BLK) void k_bodies_pre(Dev d, u32 K) {
    const Ctrl *c = d.ctrl;
    const State P = c->state[(K - 1) & 1];
    if (P.done) return;
    const u32 r_total = P.r_total + c->cnt_r[(K - 1) % 3].v;
    u32 q_total = P.q_total + c->cnt_q[(K - 1) % 3].v;
    if (q_total > d.q_cap) q_total = d.q_cap;
    if (q_total > P.q_end || !P.allow_fire || !(r_total > P.r_end)) return; // launch K applies queued records, or has no level to fire
    const u32 b_begin = P.r_end, b_end = r_total, nh = gridDim.x, h = blockIdx.x;
    u32 per_wave = 64;
    while (per_wave > 8u && (u64)nh * (per_wave / 2u) * (BLK / 64u) >= (u64)(b_end - b_begin)) per_wave >>= 1;
    const u32 per_wg = per_wave * (BLK / 64u), wv = threadIdx.x >> 6, ln = threadIdx.x & 63u;
    for (u32 base = b_begin + h * per_wg; base < b_end; base += nh * per_wg) { // workgroup-uniform trip count
        const u32 i = base + wv * per_wave + ln;
        const u32 e = (ln < per_wave && i < b_end) ? d.fired[i] : INVALID;
        bool tile = false; u32 tt[3] = {0, 0, 0};
        if (e != INVALID) deferred_body<true>(d, e, tile, tt);
        tile_bodies_wave(d, tile, e, tt);
        __syncthreads(); // the rows of a tile were stored by the eight lanes of its group: all there before the record below reads them
        const u32 tg = tile ? tile_target(tt) : 0u;
        if (tg && tg - 1 < d.max_db) {
            const u64 home = c->db_home[tg - 1];
            if (OCRGPU_GUID_LOCATION(home) != d.location) emit_remote_tile(d, home, 0u, XF_TILE_WB, d.db[tg - 1], tg - 1);
        }
    }
}

// A run that stopped on its launch budget pays the bodies it owes before the caller can look at values.
__global__ __launch_bounds__(BLK) void k_owed_bodies(Dev d) { deferred_bodies<true>(d, d.ctrl->b_begin, d.ctrl->b_end, blockIdx.x, gridDim.x); }
__global__ void k_owed_clear(Dev d) { d.ctrl->b_begin = d.ctrl->b_end = 0; }

// Start a run at launch index K0: seed state[(K0-1)&1] from the between-runs view (single thread).
__device__ __forceinline__ void begin_run(const Dev &d, u32 K0, u32 allow_fire, u32 force_level) {
    Ctrl *c = d.ctrl;
    if (force_level & 2u) { // OCRGPU_STEP_DROP_EMPTY_LEVEL: forget a trailing forced level in which nothing fired
        const u32 lv = c->level;
        if (lv != 0xFFFFFFFFu && lv < d.max_levels && d.level_start[lv] == d.level_start[lv + 1]) { c->level = lv - 1; c->n_levels = lv; }
        force_level &= 1u;
    }
    State P; memset(&P, 0, sizeof P);
    P.mode = MODE_EVENTS; P.level = c->level; P.r_begin = P.r_end = c->r_end; P.q_begin = P.q_end = c->q_begin;
    const u32 r_tail = __hip_atomic_load(&c->r_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), q_tail = __hip_atomic_load(&c->q_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    P.done = 0; P.allow_fire = allow_fire; P.r_total = r_tail; P.q_total = q_tail > d.q_cap ? d.q_cap : q_tail; // (atomic loads: the apply kernel's last workgroup begins the next run)
    P.n_levels = c->n_levels; P.force_level = force_level; P.b_begin = c->b_begin; P.b_end = c->b_end;
    c->state[(K0 - 1) & 1] = P;
    c->cnt_r[(K0 - 1) % 3].v = 0; c->cnt_q[(K0 - 1) % 3].v = 0;
    c->cnt_r[K0 % 3].v = 0; c->cnt_q[K0 % 3].v = 0;
}
__global__ void k_begin(Dev d, u32 K0, u32 allow_fire, u32 force_level) { begin_run(d, K0, allow_fire, force_level); }
// Rounds queued AHEAD of the host (ocrgpu_resolve_exchange over peer windows): what a round is -- the next level, a settle round,
// the rest of a step that ran out of launches, or nothing because the run is over -- is decided here, on the device, from the
// words the last round's apply left (RA_*: taken from the exchanged header sums, so every partition decides alike).
#define RA_STOP 1026u   // the run is over: rounds still queued do nothing (they still push and wait, so nobody hangs)
#define RA_HOLD 1027u   // something is still in flight or queued somewhere: the next round settles, it does not fire
#define RA_LEVELS 1028u // levels closed with activity
#define RA_KIND 1029u   // this round: 1 = fires a level, 0 = settles, 2 = idle
#define RA_ACT 1030u    // (u64, 8-byte aligned) activity of the level so far
#define RA_ROUND 1032u  // rounds applied since k_ra_init
#define RA_STOP_AT 1033u // 1 + the round in which RA_STOP was set (what the host goes by: see resolve_run_ahead)
#define RA_BEHIND 1034u  // a NO_CASCADE run was stopped because a partition had fallen a level behind
#define RA_SHORT 1035u   // rounds whose step ran out of launches (it went on in the next round)
__global__ void k_ra_init(u32 *w) { w[RA_STOP] = 0; w[RA_HOLD] = 0; w[RA_LEVELS] = 0; w[RA_KIND] = 0; w[RA_ACT] = 0; w[RA_ACT + 1] = 0; w[RA_ROUND] = 0; w[RA_STOP_AT] = 0; w[RA_BEHIND] = 0; w[RA_SHORT] = 0; }
// What launch K + 1 would find, without launching it: the log totals with launch K's appends, and whether it would only notice that
// the run is over (k_step's phase decision: nothing queued, no fire budget left, no bodies owed).  The level loops across
// partitions go by this instead of paying a launch per level for the notice.
struct RunView { u32 r_total, q_total; bool over; };
__device__ __forceinline__ RunView run_view(const Dev &d, u32 K) {
    const Ctrl *c = d.ctrl;
    const State *N = &c->state[K & 1];
    RunView v{N->r_total, N->q_total, true};
    if (N->done) return v;
    v.r_total += c->cnt_r[K % 3].v; v.q_total += c->cnt_q[K % 3].v;
    if (v.q_total > d.q_cap) v.q_total = d.q_cap;
    v.over = !(v.q_total > N->q_end) && !N->allow_fire && !(N->b_end > N->b_begin);
    return v;
}
__device__ __forceinline__ void begin_auto(const Dev &d, u32 K0, u32 *w) {
    u32 fire = 0, force = 0;
    if (w[RA_STOP]) w[RA_KIND] = 2;
    else if (w[RA_HOLD]) {
        const State *P = &d.ctrl->state[(K0 - 1) & 1]; // how the last run ended: one that ran out of launches goes on with what it had left
        if (!P->done) { fire = P->allow_fire; force = fire ? OCRGPU_STEP_FORCE_LEVEL : 0u; if (!run_view(d, K0 - 1).over) w[RA_SHORT] += 1; }
        w[RA_KIND] = 0;
    } else { fire = 1; force = OCRGPU_STEP_FORCE_LEVEL; w[RA_KIND] = 1; w[RA_ACT] = 0; w[RA_ACT + 1] = 0; }
    begin_run(d, K0, fire, force);
}
__global__ void k_begin_auto(Dev d, u32 K0, u32 *w) { begin_auto(d, K0, w); }
// End of a run whose last launch was K: fold the run state back into the between-runs view; if the record
// log is fully consumed, rewind it (keeps q_cap small).
__device__ __forceinline__ void end_of_run(const Dev &d, u32 K) {
    Ctrl *c = d.ctrl;
    const State P = c->state[K & 1];
    c->level = P.level; c->r_end = P.r_end; c->n_levels = P.n_levels;
    c->b_begin = P.b_begin; c->b_end = P.b_end; // bodies an interrupted run still owes (a finished run has flushed them)
    // a finished run made no appends after its totals were folded; an interrupted one (launch budget) did
    c->r_tail = P.r_total + (P.done ? 0u : c->cnt_r[K % 3].v);
    u32 qt = P.q_total + (P.done ? 0u : c->cnt_q[K % 3].v);
    c->q_begin = P.q_end; c->q_tail = qt;
    if ((P.done || (!P.allow_fire && !(P.b_end > P.b_begin))) && P.q_end >= qt) { c->q_begin = 0; c->q_tail = 0; } // (over, noticed or not: run_view)
}
__global__ void k_end(Dev d, u32 K) { end_of_run(d, K); }

// ------------------------------------------------------------------------------------------------
// the barrier-free resolve: one resident kernel, no level barrier, the level carried as data
// ------------------------------------------------------------------------------------------------
// A chain of narrow levels (config 2: 1000 levels of 10 000 EDTs) costs a level-synchronous engine one launch and one cold burst of
// dependent memory trips per level, whatever its stages do.  Here an EDT fires as soon as it is ready, in the launch
// that made it ready: waves take ready EDTs from ticket queues, fan their output events out (commonSatisfyWaiters, hc-event.c:369
// -> satisfyTaskHc, hc-task.c:1216-1266, exactly as k_step's FIRE pass does), and hand the EDTs they make ready to the queues --
// the reference's own order of work (depth first, no levels at all, hc-event.c:369-433), on 1024 waves.  The parity contract's
// per-level fired SETS stay exact because the level is data: level(EDT) = 1 + max level(its satisfiers), kept with an atomicMax
// on EdtLine.level issued in FRONT of the returning add on EdtLine.pending of the same 8-byte word -- the add that takes pending to
// zero brings the final level back -- which is what the replay harness's Lamport clock does on the reference
// (oracle/ocr_replay.c:167-181).  The fired log is rebuilt grouped by level when the kernel has ended (k_async_group<false> / k_async_scan /
// k_async_group<true>), so everything behind ocrgpu_resolve reads what it always read.
// What may run this way (anything else takes the level-synchronous path): one partition, no NANNY / Lockable / tile engine; only
// ONCE and LATCH events; NOP / HASH bodies without scripts; every registered dependence either an EDT pre-slot whose payload word
// was written at registration (RF_PREFILLED) or a LATCH DECR slot -- objects whose firing does not depend on the order of their
// satisfies (a latch that only counts down fires on its last DECR whoever brings it).  Ctrl.async_block counts what falls outside.
// Values: a body publishes its value with one 8-byte sc1 store; a consumer's body gathers its producers' {param0, val} with
// 16-byte sc1 loads and looks again in the rare case a value is still 0 (the body runs behind the hand-over of the EDTs its
// wave made ready, so nothing on the dependence path waits for a value).
#define ANSH 8u             // queue shards (a wave takes tickets from shard wave % 8 and hands over to shard (wave + iteration) % 8)
#define AROW 8u             // waiters of an EDT's output event kept beside the line, indexed by EDT (Dev-free: ADev.irow)
#define A_EVT 0x80000000u   // queue entry id: a fired event (index + 1) instead of a ready EDT (index + 1)
#define A_OUT 512u          // ready entries a wave collects before it hands them over
#define A_SKIP 0xFFFFFFFFu   // queue entry id: padding of a hand-over's last block (the ticket is served, there is nothing to fire)
#define A_BK 1024u          // ready entries a wave can keep for itself when the queues are full
#define ALAT_BIAS 0x40000000u
#define AMAX_ROW 64u        // rows longer than this: level-synchronous path (its big-row pieces spread over the grid).  A wave takes a block of ~24 ready
                            // EDTs and fans their rows out 64 edges at a time, each pass a round trip: 24 rows of 255 waiters (a level of trsm tasks
                            // in the tiled Cholesky) are 96 dependent passes on ONE wave while a thousand others idle -- T = 256 took 13.5 ms this way
                            // and 7.2 ms level by level.  OCRGPU_ASYNC_MAX_ROW overrides.
#define AROW_TRACK 32u      // Ctrl.max_row is exact above this (below it nobody asks)
#define AMAX_SEEDS 65536u   // more ready EDTs than this at the start: a wide DAG, atomic-rate bound, better served level by level
struct alignas(128) AShard { u32 tail; u32 p0[31]; u32 head; u32 p1[31]; u64 acct; u32 p2[30]; }; // acct = entries consumed << 32 | busy waves
struct ACtl { AShard sh[ANSH]; u32 old_tail[ANSH]; u32 L0, seed_begin, seed_end, max_level, n_lv_prev, total_out, pad[2]; };
struct ADev { ACtl *c; u64 *q; u64 *irow; u64 *alat; u32 *hist; u32 cap, refill /* scan width: a power of two >= blk */, n_evt, blk /* entries per block = tickets a wave takes */; };

__device__ __forceinline__ u32 a_ld(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 a_ld(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
// Eight 16-byte sc1 loads in flight, one wait (which also covers what the wave issued before the block).  Load k is made by the
// lanes of mask m[k] only (EXEC is narrowed around it): a wave holds about 14 EDTs with 4 pre-slots each, and the idle lanes and
// unused pre-slots would otherwise put seven times as many requests on the fabric as the real ones.  g[k] of a lane outside m[k] is
// not written.
__device__ __forceinline__ void a_gather8(const void *const (&p)[8], const u64 (&m)[8], u32x4 (&g)[8]) {
    u64 sav;
    asm volatile("s_mov_b64 %8, exec\n\t"
                 "s_mov_b64 exec, %17\n\tglobal_load_dwordx4 %0, %9, off sc1\n\t"
                 "s_mov_b64 exec, %18\n\tglobal_load_dwordx4 %1, %10, off sc1\n\t"
                 "s_mov_b64 exec, %19\n\tglobal_load_dwordx4 %2, %11, off sc1\n\t"
                 "s_mov_b64 exec, %20\n\tglobal_load_dwordx4 %3, %12, off sc1\n\t"
                 "s_mov_b64 exec, %21\n\tglobal_load_dwordx4 %4, %13, off sc1\n\t"
                 "s_mov_b64 exec, %22\n\tglobal_load_dwordx4 %5, %14, off sc1\n\t"
                 "s_mov_b64 exec, %23\n\tglobal_load_dwordx4 %6, %15, off sc1\n\t"
                 "s_mov_b64 exec, %24\n\tglobal_load_dwordx4 %7, %16, off sc1\n\t"
                 "s_mov_b64 exec, %8\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7]), "=&s"(sav)
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]),
                   "s"(m[0]), "s"(m[1]), "s"(m[2]), "s"(m[3]), "s"(m[4]), "s"(m[5]), "s"(m[6]), "s"(m[7]) : "memory");
}
// Termination: no wave busy and every reserved entry consumed, seen twice in a row.  tail and consumed only grow, and busy only
// rises in the same atomic as consumed: two equal snapshots are one consistent view.
__device__ bool a_quiescent(const ADev &a) {
    const u32 lane = lane_id();
    u64 x = 0, y = 0; u32 tx = 0, ty = 0;
    if (lane < ANSH) { x = a_ld(&a.c->sh[lane].acct); tx = a_ld(&a.c->sh[lane].tail); }
    if (lane < ANSH) { y = a_ld(&a.c->sh[lane].acct); ty = a_ld(&a.c->sh[lane].tail); }
    if (__ballot(x != y || tx != ty)) return false;
    u64 cons = x >> 32, busy = x & 0xFFFFFFFFull, tl = tx < a.cap ? tx : a.cap; // (what was reserved beyond a shard's end is never written)
    for (int o = 1; o < (int)ANSH; o <<= 1) { cons += __shfl_xor(cons, o); busy += __shfl_xor(busy, o); tl += __shfl_xor(tl, o); }
    cons = __shfl(cons, 0); busy = __shfl(busy, 0); tl = __shfl(tl, 0);
    return (busy & 0xFFFFFFFFull) == 0 && cons == tl;
}
// one dependence edge per lane: the atomics go out here, what they returned is looked at in a_collect (several edges in flight)
struct AEdge { u64 o; u32 dst, fl, slv; bool act; };
__device__ __forceinline__ void a_issue(const Dev &d, bool act, u64 w, u32 slv, AEdge &r) {
    const u32 hi = (u32)(w >> 32), slot = hi & 0x0FFFFFFFu;
    r.act = act; r.dst = (u32)w; r.fl = hi >> 28; r.slv = slv; r.o = 0;
    if (act && (r.fl & RF_DST_EDT)) { // satisfyTaskHc: the level first, then ONE 64-bit returning add on {pending, level} (same word: it cannot overtake the max)
        EdtLine *D = &d.edt[r.dst];
        atomicMax(&D->level, slv + 1u);
        r.o = atomicAdd((unsigned long long *)D, (r.fl & RF_INLINE) ? (unsigned long long)(PEND_INL_BIT(slot) - 1u) : ~0ull);
    }
}
// EDTs made ready go to lds_id / lds_lv; LATCH destinations are only noted (lds_lat: {latch, level}) -- every leaf of a finish scope
// hits the same counter word, and a wave that waited for that word here would hold back the EDTs it has just made ready
#define A_LAT 128u
__device__ __forceinline__ void a_collect(const AEdge &r, u32 *lds_id, u32 *lds_lv, u32 *lds_lat, u32 &nready, u32 &nlat, u64 &n_edt_sat, u64 &n_evt_sat) {
    const bool to_edt = r.act && (r.fl & RF_DST_EDT), to_lat = r.act && !to_edt;
    const bool ready = to_edt && ((u32)r.o & ((r.fl & RF_INLINE) ? PEND_INL_CNT : PEND_MASK)) == 1u;
    const u64 em = __ballot(to_edt), lm = __ballot(to_lat), rm = __ballot(ready);
    n_edt_sat += __popcll(em); n_evt_sat += __popcll(lm);
    if (ready) {
        const u32 pos = nready + rank_in(rm);
        u32 lv = (u32)(r.o >> 32); if (lv < r.slv + 1u) lv = r.slv + 1u;
        if (pos < A_OUT) { lds_id[pos] = r.dst + 1u; lds_lv[pos] = lv; }
    }
    nready += (u32)__popcll(rm);
    if (to_lat) { const u32 pos = nlat + rank_in(lm); if (pos < A_LAT) { lds_lat[2 * pos] = r.dst; lds_lat[2 * pos + 1] = r.slv; } }
    nlat += (u32)__popcll(lm);
}
// satisfyEventHcLatch (hc-event.c:812-868), DECR only, for the noted edges: the lanes of one latch take ONE returning add with their
// count, behind an atomicMax with their highest level; the add that leaves zero fires the latch (once; it is destroyed, :836,885)
__device__ void a_latches(const Dev &d, const ADev &a, const u32 *lds_lat, u32 nlat, u32 *lds_id, u32 *lds_lv, u32 &nready) {
    const u32 lane = lane_id();
    for (u32 base = 0; base < nlat; base += 64) {
        const bool act = base + lane < nlat;
        const u32 x = act ? lds_lat[2 * (base + lane)] : 0u, slv = act ? lds_lat[2 * (base + lane) + 1] : 0u;
        bool lead = false; u32 gcnt = 0, glvl = 0; u64 o = 0;
        u64 todo = __ballot(act);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const u32 x0 = __shfl(x, leader);
            const u64 grp = __ballot(act && x == x0) & todo;
            u32 m = (act && x == x0) ? slv : 0u;
            for (int s = 1; s < 64; s <<= 1) { const u32 n = __shfl_xor(m, s); m = n > m ? n : m; }
            if ((int)lane == leader) {
                lead = true; gcnt = (u32)__popcll(grp); glvl = m;
                atomicMax(reinterpret_cast<u32 *>(&a.alat[x0]) + 1, m);
                o = atomicAdd((unsigned long long *)&a.alat[x0], 0ull - (unsigned long long)gcnt);
            }
            todo &= ~grp;
        }
        bool ready = false; u32 lv = 0;
        if (lead && (u32)o - gcnt == ALAT_BIAS) {
            const u32 was = atomicExch(evt_state(d, x), ST_DESTROYED | (OCREVT_LATCH << 8));
            if ((was & 0xFF) != ST_LIVE) atomicAdd(&d.ctrl->err[ERR_DEAD_EVENT], 1u);
            else if (d.evt[x].deg) { ready = true; lv = (u32)(o >> 32); if (lv < glvl) lv = glvl; }
        }
        const u64 rm = __ballot(ready);
        if (ready) { const u32 pos = nready + rank_in(rm); if (pos < A_OUT) { lds_id[pos] = A_EVT | (x + 1u); lds_lv[pos] = lv; } }
        nready += (u32)__popcll(rm);
    }
}

#ifdef OCRGPU_ASTAMPS
static __device__ unsigned long long g_ats[ANSH << 20]; // when an entry was handed over (10 ns ticks), by shard and entry index mod 2^20
#endif
__global__ __launch_bounds__(64) void k_async(Dev d, ADev a) {
    __shared__ u64 lds_w[AROW * 32];
    __shared__ u32 lds_l[AROW * 32];
    __shared__ u32 lds_id[A_OUT], lds_lv[A_OUT];
    __shared__ u32 lds_lat[2 * A_LAT];
    __shared__ u32 lds_bk[2 * A_BK];
    const u32 lane = threadIdx.x, wid = blockIdx.x, sh = wid & (ANSH - 1u), B = a.blk;
    ACtl *c = a.c;
    u32 csh = sh, busy_sh = sh, dead = 0;         // the shard this wave takes tickets from; the one it is counted busy on; shards found used up
    u32 item = INVALID, lvl = 0, ticket = INVALID, spins = 0, iter = 0, max_lv = 0, sink = 0, nbk = 0;
    bool busy = false; // (wave-uniform) this wave is counted in a shard's busy total
    u64 n_edt_sat = 0, n_evt_sat = 0, n_fired = 0;
#ifdef OCRGPU_ASTAMPS
    u64 s_iter = 0, s_titer = 0, s_tbody = 0, s_idle = 0, s_items = 0, s_thand = 0, s_tload = 0, s_tfan = 0, s_lat = 0, s_tlat = 0, s_tgat = 0, s_thop = 0, s_nhop = 0, t_ready = 0; u32 last_ticket = INVALID;
#define ASTAMP(x) x
#else
#define ASTAMP(x)
#endif
    // Hand the collected entries to a queue: one returning add reserves whole BLOCKS of B entries (the last one padded with A_SKIP),
    // the entries go out as 8-byte sc1 stores.  A consumer takes B tickets = one block = what ONE wave wrote in one go: its batch
    // arrives together (tickets that straddled two hand-overs left entries waiting for a wave already at work: 9.5 -> 7.9 ms).
    // A queue that is used up cannot take the entries: the wave keeps them and fires them itself (depth first, as the reference does).
    auto hand_over = [&](u32 &nready) {
        if (nready > A_OUT) nready = A_OUT; // (cannot happen: the callers flush in time)
        if (!nready) return;
        const u32 tsh = (sh + iter) & (ANSH - 1u), nb = (nready + B - 1u) / B * B;
        u32 qb = 0;
        if (lane == 0) qb = atomicAdd(&c->sh[tsh].tail, nb);
        qb = __shfl(qb, 0);
        u64 *QT = a.q + (size_t)tsh * a.cap;
        if ((u64)qb + nb <= a.cap) {
            ASTAMP(for (u32 i = lane; i < nb; i += 64) g_ats[((size_t)tsh << 20) + ((qb + i) & 0xFFFFFu)] = t_ready);
            for (u32 i = lane; i < nb; i += 64)
                __hip_atomic_store(&QT[qb + i], i < nready ? (((u64)lds_lv[i] << 32) | (u64)lds_id[i]) : (u64)A_SKIP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (u32 i = lane; i < nb; i += 64) if (qb + i < a.cap) __hip_atomic_store(&QT[qb + i], (u64)A_SKIP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (u32 i = lane; i < nready; i += 64) {
                if (nbk + i < A_BK) { lds_bk[2 * (nbk + i)] = lds_id[i]; lds_bk[2 * (nbk + i) + 1] = lds_lv[i]; }
                else atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u);
            }
            nbk = nbk + nready < A_BK ? nbk + nready : A_BK;
        }
        nready = 0;
    };
    for (;;) {
        const u64 *Q = a.q + (size_t)csh * a.cap;
        bool got = false;
        for (u32 g = 0; g < 64; ++g) { // a block's entries arrive together, give or take the skew of one wave's stores: look until all are there
            if (ticket != INVALID) { const u64 v = a_ld(&Q[ticket]); if ((u32)v) { if ((u32)v != A_SKIP) { item = (u32)v; lvl = (u32)(v >> 32); } ticket = INVALID; got = true; } }
            if (!__ballot(got) || !__ballot(ticket != INVALID)) break;
        }
        const u64 gm = __ballot(got);
        u64 hm = __ballot(item != INVALID);
        if (gm) { // entries consumed (+ idle -> busy) in ONE atomic: "consumed" never shows without the "busy" that goes with it
            const bool nb = hm != 0 && !busy;
            if (lane == 0) atomicAdd((unsigned long long *)&c->sh[csh].acct, ((unsigned long long)__popcll(gm) << 32) + (nb ? 1ull : 0ull));
            if (nb) { busy = true; busy_sh = csh; }
        }
        if (!hm && nbk && !__ballot(ticket != INVALID)) { // entries this wave had to keep: it fires them itself
            const u32 take = nbk < B ? nbk : B;
            if (lane < take) { item = lds_bk[2 * (nbk - take + lane)]; lvl = lds_bk[2 * (nbk - take + lane) + 1]; }
            nbk -= take;
            hm = __ballot(item != INVALID);
            __syncthreads();
        }
        if (!hm) {
            if (busy && !nbk) { if (lane == 0) atomicAdd((unsigned long long *)&c->sh[busy_sh].acct, ~0ull); busy = false; } // (behind the hand-over's returned add)
            if (!__ballot(ticket != INVALID)) { // (a block's tickets are served together: none is left when the wave comes back)
                if (dead >= ANSH) { // every queue is used up: nothing more can reach this wave
                    ++spins;
                    if ((spins & 7u) == 0 && a_quiescent(a)) break;
                    __builtin_amdgcn_s_sleep(64);
                    continue;
                }
                u32 base = 0;
                if (lane == 0) base = atomicAdd(&c->sh[csh].head, B);
                base = __shfl(base, 0);
                if ((u64)base + B <= a.cap) { if (lane < B) ticket = base + lane; ASTAMP(last_ticket = base); }
                else { ++dead; csh = (csh + 1u) & (ANSH - 1u); }
                continue;
            }
            // waiting: every wave polls its OWN ticket words (polling a shared word from hundreds of waves slows every request that
            // shares its channel: measured); the shared totals are looked at rarely
            ++spins; ASTAMP(++s_idle);
            if ((spins & 31u) == 0 && a_quiescent(a)) break;
            if (spins < 8) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(8);
            continue;
        }
        spins = 0; ++iter;
        ASTAMP(const u64 t_in = __builtin_amdgcn_s_memrealtime(); ++s_iter; s_items += __popcll(hm));
        ASTAMP(if (gm && lane == 0 && last_ticket != INVALID) { const u64 ts = g_ats[((size_t)csh << 20) + (last_ticket & 0xFFFFFu)]; if (ts && t_in > ts && t_in - ts < 100000) { s_thop += t_in - ts; ++s_nhop; } });
        const bool has = item != INVALID, is_evt = has && (item & A_EVT) != 0, is_edt = has && !is_evt;
        const u32 idx = has ? (item & ~A_EVT) - 1u : 0u;
        // ---- one trip: the EDT's line and its inline waiter row (both addressed by the EDT's index alone) ----
        u64 rw[AROW]; u32 row = 0, deg = 0, body = 0, depc = 0, ostate = 0, sl[INLINE_SLOTS]; u64 param0 = 0;
#pragma unroll
        for (int k = 0; k < (int)AROW; ++k) rw[k] = 0;
#pragma unroll
        for (int k = 0; k < (int)INLINE_SLOTS; ++k) sl[k] = 0;
        if (is_edt) {
            const EdtLine *L = &d.edt[idx];
            const ulonglong2 *R = reinterpret_cast<const ulonglong2 *>(a.irow + (size_t)idx * AROW);
#pragma unroll
            for (int k = 0; k < (int)AROW / 2; ++k) { const ulonglong2 x = R[k]; rw[2 * k] = x.x; rw[2 * k + 1] = x.y; }
            body = L->body; depc = L->depc; param0 = L->param0; ostate = L->out_state; row = L->out_row;
            deg = (body & OCRBODY_F_OUTEVT) ? L->out_deg : 0u;
#pragma unroll
            for (int k = 0; k < (int)INLINE_SLOTS; ++k) sl[k] = L->slot[k];
            if (lvl > max_lv) max_lv = lvl;
        } else if (is_evt) { const EvtRec r = d.evt[idx]; row = r.row_begin; deg = r.deg; }
        n_fired += __popcll(__ballot(is_edt)); n_evt_sat += __popcll(__ballot(is_edt && (body & OCRBODY_F_OUTEVT))); // the epilogue's satisfy of the output event
        const u32 cnt = is_edt ? (deg < AROW ? deg : AROW) : 0u, rem = deg - cnt;
        u32 incl = cnt;
        for (u32 o = 1; o < a.refill; o <<= 1) { const u32 nb = __shfl_up(incl, o); if (lane >= o) incl += nb; } // (items sit in the ticket lanes)
        const u32 total = __shfl(incl, a.refill - 1u), excl = incl - cnt;
#pragma unroll
        for (int k = 0; k < (int)AROW; ++k) if ((u32)k < cnt) { lds_w[excl + k] = rw[k]; lds_l[excl + k] = lvl; }
        __syncthreads();
        ASTAMP(const u64 t_ld = __builtin_amdgcn_s_memrealtime(); s_tload += t_ld - t_in);
        // overflow pre-slot words of the body's first round (EDTs with more than 5 pre-slots): asked for beside the atomics below
        u32 ow[8];
        {
            const u32 *ov = d.slots + sl[0];
#pragma unroll
            for (int k = 0; k < 8; ++k) ow[k] = (is_edt && depc > INLINE_SLOTS && (u32)k < depc) ? ov[k] : 0u;
        }
        u32 nready = 0, nlat = 0;
        for (u32 base = 0; base < total; base += 128) {
            AEdge e0, e1;
            { const u32 t = base + lane; const bool act = t < total; a_issue(d, act, act ? lds_w[t] : 0ull, act ? lds_l[t] : 0u, e0); }
            { const u32 t = base + 64 + lane; const bool act = t < total; a_issue(d, act, act ? lds_w[t] : 0ull, act ? lds_l[t] : 0u, e1); }
            if (nlat + 128 > A_LAT) { __syncthreads(); a_latches(d, a, lds_lat, nlat, lds_id, lds_lv, nready); nlat = 0; __syncthreads(); }
            a_collect(e0, lds_id, lds_lv, lds_lat, nready, nlat, n_edt_sat, n_evt_sat);
            a_collect(e1, lds_id, lds_lv, lds_lat, nready, nlat, n_edt_sat, n_evt_sat);
        }
        // rows longer than the inline part, and the rows of fired events: edge-balanced over the wave straight from waiters[]
        u32 rincl = rem, rtotal = 0;
        if (__ballot(rem != 0)) {
            for (u32 o = 1; o < a.refill; o <<= 1) { const u32 nb = __shfl_up(rincl, o); if (lane >= o) rincl += nb; }
            rtotal = __shfl(rincl, a.refill - 1u);
        }
        const u32 rexcl = rincl - rem;
        for (u32 base = 0; base < rtotal; base += 64) {
            const u32 t = base + lane; const bool act = t < rtotal;
            u32 lo = 0, hi = a.refill - 1u;
            while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (__shfl(rincl, mid) > t) hi = mid; else lo = mid + 1; }
            const u32 start = __shfl(rexcl, lo), r0 = __shfl(row + cnt, lo), slv = __shfl(lvl, lo);
            AEdge e0;
            a_issue(d, act, act ? d.waiters[r0 + (t - start)] : 0ull, slv, e0);
            if (nready + 64 > A_OUT) { __syncthreads(); hand_over(nready); __syncthreads(); }
            if (nlat + 64 > A_LAT) { __syncthreads(); a_latches(d, a, lds_lat, nlat, lds_id, lds_lv, nready); nlat = 0; __syncthreads(); }
            a_collect(e0, lds_id, lds_lv, lds_lat, nready, nlat, n_edt_sat, n_evt_sat);
        }
        __syncthreads();
        ASTAMP(const u64 t_fan = __builtin_amdgcn_s_memrealtime(); s_tfan += t_fan - t_ld; t_ready = t_fan);
        hand_over(nready);
        ASTAMP(const u64 t_ho = __builtin_amdgcn_s_memrealtime(); s_thand += t_ho - t_fan; s_titer += t_ho - t_in);
        // ---- behind the hand-over: the latch edges, then the body (taskExecute's user function; the epilogue's output-event satisfy
        // was the fan-out above) ----
        if (nlat) { a_latches(d, a, lds_lat, nlat, lds_id, lds_lv, nready); __syncthreads(); hand_over(nready); }
        ASTAMP(const u64 t_la = __builtin_amdgcn_s_memrealtime(); s_tlat += t_la - t_ho);
        if (__ballot(is_edt)) {
            u64 v = ocrops_hash_init(param0);
            const u32 *ov = d.slots + sl[0];
            for (u32 k0 = 0; __ballot(is_edt && k0 < depc); k0 += 8) {
                u32 pw[8]; const void *ga[8]; u32x4 g[8]; u64 gm8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    u32 p = 0;
                    if (is_edt && k0 + k < depc) {
                        if (depc <= INLINE_SLOTS) p = k < (int)INLINE_SLOTS ? sl[k < (int)INLINE_SLOTS ? k : 0] : 0u;
                        else p = k0 ? ov[k0 + k] : ow[k];
                    }
                    p &= 0x7FFFFFFFu; pw[k] = p;
                    if (p == 0) ga[k] = &d.edt[0].val;                                   // NULL_GUID: nothing to read (the lane sits this load out)
                    else if (p - 1 < d.max_db) ga[k] = &d.db[p - 1];                    // an explicit datablock {value, tag}: not written during a resolve
                    else ga[k] = &d.edt[p - 1 - d.max_db].param0;                          // a producer's {param0, val}
                    gm8[k] = __ballot(p != 0);
                }
                for (u32 tries = 0;; ++tries) { // (rare) a producer's value is not there yet: its body runs behind its hand-over, ours may have overtaken it
                    a_gather8(ga, gm8, g);
                    // A producer's value is there when it is not 0: val is 0 from creation until the body's one 8-byte store.  (A fold
                    // that IS 0 -- one in 2^64 -- makes its consumers look until they give up, and then they use the 0 they read: right.)
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (pw[k] && pw[k] - 1 >= d.max_db && (g[k].z | g[k].w) == 0u) ok = false;
                    if (!__ballot(!ok)) break;
                    ASTAMP(++s_lat);
                    if (tries > (1u << 18)) break; // (values of EDTs that may never run are kept off this path: k_add_dep / k_satisfy)
                    __builtin_amdgcn_s_sleep(2);
                }
                ASTAMP(s_tgat += __builtin_amdgcn_s_memrealtime() - t_la);
#pragma unroll
                for (int k = 0; k < 8; ++k) if (is_edt && k0 + k < depc) {
                    u64 val = 0, tag = ~0ULL;
                    if (pw[k]) { // an explicit datablock is {value, tag}, a producer's line {param0, val}
                        const u64 lo = ((u64)g[k].y << 32) | g[k].x, hi = ((u64)g[k].w << 32) | g[k].z;
                        if (pw[k] - 1 < d.max_db) { val = lo; tag = hi; } else { val = hi; tag = d.hash_tag_base + lo; }
                    }
                    v = ocrops_hash_step(v, k0 + k, val, tag + 1);
                }
            }
            if (is_edt) {
                __hip_atomic_store(&d.edt[idx].val, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // the value consumers wait for: one 8-byte sc1 store
                // satisfyEventHcOnce: the output event has fired and is destroyed; the body has run (read after this launch only)
                *reinterpret_cast<uint2 *>(&d.edt[idx].out_state) = make_uint2((body & OCRBODY_F_OUTEVT) ? (u32)(ST_DESTROYED | (OCREVT_ONCE << 8)) : ostate, body | BODY_VAL_DONE);
            }
        }
        item = INVALID;
        __syncthreads();
        ASTAMP(s_tbody += __builtin_amdgcn_s_memrealtime() - t_ho);
    }
#ifdef OCRGPU_ASTAMPS
    if (lane == 0) { unsigned long long *g = (unsigned long long *)d.dbg + 32; atomicAdd(g, s_iter); atomicAdd(g + 1, s_titer); atomicAdd(g + 2, s_tbody); atomicAdd(g + 3, s_idle);
                     atomicAdd(g + 4, s_items); atomicAdd(g + 5, s_thand); atomicAdd(g + 6, s_tload); atomicAdd(g + 7, s_tfan); atomicAdd(g + 8, s_lat); atomicAdd(g + 9, s_tlat); atomicAdd(g + 10, s_tgat); atomicAdd(g + 11, s_thop); atomicAdd(g + 12, s_nhop); }
#endif
    if (lane == 0) {
        if (n_edt_sat | n_evt_sat | n_fired) {
            unsigned long long *bs = (unsigned long long *)(d.blk_stats + 4 * (blockIdx.x % MAX_GRID));
            atomicAdd(bs, (unsigned long long)n_edt_sat); atomicAdd(bs + 1, (unsigned long long)n_evt_sat); atomicAdd(bs + 2, (unsigned long long)n_fired);
        }
    }
    u32 m = max_lv;
    for (int o = 1; o < 64; o <<= 1) { const u32 n = __shfl_xor(m, o); m = n > m ? n : m; }
    if (lane == 0 && m) atomicMax(&c->max_level, m);
}

// before a run: clear what the last run left in the queues and in the level histogram
__global__ void k_async_reset(ADev a) {
    const ACtl *c = a.c;
    const u64 gs = (u64)gridDim.x * blockDim.x, t0 = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    for (u32 s = 0; s < ANSH; ++s) { const u32 n = c->old_tail[s] < a.cap ? c->old_tail[s] : a.cap; for (u64 i = t0; i < n; i += gs) a.q[(size_t)s * a.cap + i] = 0; }
    for (u64 i = t0; i < (u64)c->n_lv_prev + 2; i += gs) a.hist[i] = 0;
}
__global__ void k_async_init(Dev d, ADev a) { // one wave
    ACtl *c = a.c;
    const u32 t = threadIdx.x;
    if (t < ANSH) { c->sh[t].tail = 0; c->sh[t].head = 0; c->sh[t].acct = 0; c->old_tail[t] = 0; }
    if (t == 0) { c->L0 = d.ctrl->level + 1u; c->seed_begin = d.ctrl->r_end; c->seed_end = d.ctrl->r_tail; c->max_level = d.ctrl->level + 1u; c->n_lv_prev = 0; c->total_out = 0; }
}
// the seeds: what the main program's satisfies made ready (the sparse ready log [r_end, r_tail)) enters the queues at level L0;
// every LIVE latch gets its biased counter word {count + bias, level 0}
__global__ void k_async_seed(Dev d, ADev a) {
    ACtl *c = a.c;
    const u64 gs = (u64)gridDim.x * blockDim.x, t0 = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    for (u64 x = t0; x < a.n_evt; x += gs) if (((d.evt[x].state_kind >> 8) & 0xFF) == OCREVT_LATCH) a.alat[x] = (u64)((u32)d.evt[x].val + ALAT_BIAS);
    const u32 b = c->seed_begin, n = c->seed_end - b, B = a.blk;
    for (u64 i0 = (t0 & ~63ull); i0 < n; i0 += gs) { // (whole waves stay together: the append below is wave-wide)
        const u64 i = i0 + (t0 & 63u);
        const u32 e = i < n ? d.fired[b + i] : INVALID;
        const u32 s = (u32)((i0 / 64) % ANSH);
        const u64 m = __ballot(e != INVALID);
        if (!m) continue;
        const u32 cnt = (u32)__popcll(m), nb = (cnt + B - 1u) / B * B; // whole blocks, the last one padded
        u32 base = 0; const int leader = __ffsll((long long)m) - 1;
        if ((int)lane_id() == leader) base = atomicAdd(&c->sh[s].tail, nb);
        base = __shfl(base, leader);
        if ((u64)base + nb > a.cap) { if ((int)lane_id() == leader) atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u); continue; }
        if (e != INVALID) a.q[(size_t)s * a.cap + base + rank_in(m)] = ((u64)c->L0 << 32) | (u64)(e + 1u);
        const u32 pad = nb - cnt;
        if (lane_id() < pad) a.q[(size_t)s * a.cap + base + cnt + lane_id()] = (u64)A_SKIP;
    }
}
// after the run: the fired log grouped by level, from the queue entries (every EDT that fired is exactly one entry).
// Queue order follows level order loosely, so a stretch of AGRP_CH consecutive entries holds a handful of levels: a workgroup
// counts its stretch in LDS over a window of AGRP_WIN levels from the lowest it sees and goes to memory once per level it met
// (SCATTER: with the returning atomic that hands the stretch its place in that level).  Levels beyond the window go one by one.
#define AGRP_PER 16u
#define AGRP_CH (256u * AGRP_PER)
#define AGRP_WIN 256u
template <bool SCATTER> __global__ __launch_bounds__(256) void k_async_group(Dev d, ADev a) {
    __shared__ u32 cnt[AGRP_WIN], base_w[AGRP_WIN], lo_s;
    const ACtl *c = a.c;
    const u32 L0 = c->L0, t = threadIdx.x;
    u32 n_s[ANSH], first[ANSH + 1]; first[0] = 0;
    for (u32 s = 0; s < ANSH; ++s) { n_s[s] = c->sh[s].tail < a.cap ? c->sh[s].tail : a.cap; first[s + 1] = first[s] + (n_s[s] + AGRP_CH - 1u) / AGRP_CH; }
    for (u32 g = blockIdx.x; g < first[ANSH]; g += gridDim.x) {
        u32 s = 0; while (g >= first[s + 1]) ++s;
        const u32 i0 = (g - first[s]) * AGRP_CH;
        const u64 *q = a.q + (size_t)s * a.cap;
        u32 id[AGRP_PER], lv[AGRP_PER], rk[AGRP_PER]; u32 lo = 0xFFFFFFFFu;
        cnt[t] = 0; if (t == 0) lo_s = 0xFFFFFFFFu;
#pragma unroll
        for (u32 k = 0; k < AGRP_PER; ++k) {
            const u32 i = i0 + k * 256u + t;
            const u64 v = i < n_s[s] ? q[i] : 0ull;
            id[k] = (u32)v; lv[k] = (u32)(v >> 32);
            bool mine = id[k] != 0 && !(id[k] & A_EVT) && lv[k] > L0;
            if (mine && lv[k] >= d.max_levels) { if (!SCATTER) atomicAdd(&d.ctrl->err[ERR_LEVEL_OVERFLOW], 1u); mine = false; }
            if (!mine) id[k] = 0; else if (lv[k] < lo) lo = lv[k];
        }
        for (int o = 1; o < 64; o <<= 1) { const u32 n = __shfl_xor(lo, o); lo = n < lo ? n : lo; }
        __syncthreads();
        if ((t & 63u) == 0 && lo != 0xFFFFFFFFu) atomicMin(&lo_s, lo);
        __syncthreads();
        lo = lo_s;
#pragma unroll
        for (u32 k = 0; k < AGRP_PER; ++k) {
            if (!id[k]) continue;
            const u32 w = lv[k] - lo;
            if (w < AGRP_WIN) rk[k] = atomicAdd(&cnt[w], 1u);
            else { const u32 r = atomicAdd(&a.hist[lv[k] - L0 - 1u], 1u); if (SCATTER) rk[k] = r; }
        }
        __syncthreads();
        if (cnt[t]) { const u32 r = atomicAdd(&a.hist[lo + t - L0 - 1u], cnt[t]); if (SCATTER) base_w[t] = r; }
        if (SCATTER) {
            __syncthreads();
#pragma unroll
            for (u32 k = 0; k < AGRP_PER; ++k) {
                if (!id[k]) continue;
                const u32 w = lv[k] - lo;
                const u32 pos = w < AGRP_WIN ? base_w[w] + rk[k] : rk[k];
                if (pos < d.fired_cap) d.fired[pos] = id[k] - 1u; else atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u);
            }
        }
        __syncthreads();
    }
}
// hist[k] (EDTs of level L0 + 1 + k) -> start offsets in the fired log behind the seeds; level_start[] for every level of the run
__global__ __launch_bounds__(1024) void k_async_scan(Dev d, ADev a) {
    __shared__ u32 ws[16]; __shared__ u32 carry_s;
    ACtl *c = a.c;
    const u32 L0 = c->L0;
    u32 top = c->max_level; if (top >= d.max_levels) top = d.max_levels - 1;
    const u32 n_lv = top - L0; // levels above the seeds
    if (threadIdx.x == 0) { carry_s = c->seed_end; if (L0 < d.max_levels) { d.level_start[L0] = c->seed_begin; d.level_start[L0 + 1] = c->seed_end; } }
    __syncthreads();
    for (u32 base = 0; base < n_lv; base += 1024) {
        const u32 k = base + threadIdx.x;
        const u32 hv = k < n_lv ? a.hist[k] : 0u;
        u32 v = hv;
        const u32 lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (int o = 1; o < 64; o <<= 1) { const u32 nb = __shfl_up(v, o); if (lane >= (u32)o) v += nb; }
        if (lane == 63) ws[w] = v;
        __syncthreads();
        u32 add = carry_s;
        for (u32 j = 0; j < w; ++j) add += ws[j];
        const u32 excl = add + v - hv;
        if (k < n_lv) { a.hist[k] = excl; d.level_start[L0 + 2 + k] = excl + hv; }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = excl + hv;
        __syncthreads();
    }
    if (threadIdx.x == 0) { c->total_out = carry_s; c->n_lv_prev = n_lv; }
}
// fold the run back into the between-runs view (what k_end does for a level-synchronous run); latch counters go back to their records
__global__ void k_async_end(Dev d, ADev a, u32 K) {
    ACtl *c = a.c;
    const u64 gs = (u64)gridDim.x * blockDim.x, t0 = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    for (u64 x = t0; x < a.n_evt; x += gs) if (((d.evt[x].state_kind >> 8) & 0xFF) == OCREVT_LATCH) d.evt[x].val = (s32)((u32)a.alat[x] - ALAT_BIAS);
    if (t0 == 0) {
        Ctrl *k = d.ctrl;
        u32 top = c->max_level; if (top >= d.max_levels) top = d.max_levels - 1;
        u32 tot = c->total_out; if (tot > d.fired_cap) tot = d.fired_cap;
        k->level = top; k->n_levels = top + 1u; k->r_end = tot; k->r_tail = tot; k->q_begin = 0; k->q_tail = 0; k->b_begin = k->b_end = 0;
        for (u32 s = 0; s < ANSH; ++s) c->old_tail[s] = c->sh[s].tail;
        State P; memset(&P, 0, sizeof P); // what a level-synchronous run's last launch would have left (ocrgpu_step_info, the host's polls)
        P.mode = MODE_FIRE; P.level = top; P.n_levels = top + 1u; P.done = 1; P.allow_fire = 0xFFFFFFFFu;
        P.r_begin = top < d.max_levels ? d.level_start[top] : tot; P.r_end = tot; P.r_total = tot;
        k->state[K & 1] = P;
    }
}

// ------------------------------------------------------------------------------------------------
// construction kernels
// ------------------------------------------------------------------------------------------------
// `gin` != nullptr: the caller chose the guids (handed out earlier by ocrgpu_guid_reserve -- the reference's
// guidReserve + labeled create, ocr-guid.h:127-144); a guid of the wrong kind or location is a malformed op.
__device__ __forceinline__ bool given_ok(const Dev &d, u64 g, u32 kind) {
    if (OCRGPU_GUID_KIND(g) == kind && OCRGPU_GUID_LOCATION(g) == d.location && OCRGPU_GUID_COUNTER(g) != 0) return true;
    atomicAdd(&d.ctrl->err[ERR_BAD_OP], 1u); return false;
}
// Indices: the first `n_take` objects of a create call take the top of the free stack (`n_free` entries before the call), the
// others fresh indices from first_idx on.
__device__ __forceinline__ u32 take_index(const u32 *free_stack, u32 n_free, u32 n_take, u32 first_idx, u64 i) {
    return i < n_take ? free_stack[n_free - 1u - (u32)i] : first_idx + (u32)(i - n_take);
}
__global__ void k_event_create(Dev d, ReclDev rc, u32 n_free, u32 n_take, u64 n, const u32 *kind, const s32 *arg, u32 first_idx, u64 first_counter,
                               const u64 *gin, u64 *guids_out) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 k = kind[i], x = take_index(rc.free_evt, n_free, n_take, first_idx, i);
    rc.evt_guids[x] = 0;
    if (k != OCREVT_ONCE && k != OCREVT_LATCH) atomicAdd(&d.ctrl->async_block, 1u); // persistent / counted / channel events: order matters
    const u32 hi = 0; s32 arg_val = 0;
    if (k == OCREVT_COUNTED) { // nbDeps > 0 (hc-event.c:1600-1612); the count lives in counted[x]
        if (arg[i] <= 0) { atomicAdd(&d.ctrl->err[ERR_BAD_OP], 1u); d.evt[x] = EvtRec{0, 0, (u32)ST_ABSENT, 0}; return; }
        d.counted[x] = (u64)(u32)arg[i] << 32;
    } else if (k == OCREVT_CHANNEL) { // arg = maxGen (ring capacity); EVENT_CHANNEL_UNBOUNDED (<= 0) gets 1024 generations in flight
        const u32 cap = arg[i] > 0 ? (u32)arg[i] : 1024u;
        const u32 ord = atomicAdd(&d.ctrl->chan_count, 1u), wb = atomicAdd(&d.ctrl->chan_wait_used, cap), sb = atomicAdd(&d.ctrl->chan_sat_used, cap);
        if (ord >= d.max_evt || (u64)sb + cap > d.chan_sat_cap || (u64)wb + cap + d.ctrl->log_tail > d.max_edges) {
            atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u); d.evt[x] = EvtRec{0, 0, (u32)ST_ABSENT, 0}; return;
        }
        d.chan[ord] = ChanRec{cap, (u32)(d.max_edges - wb - cap), sb, 0u};
        d.counted[x] = 0;
        arg_val = (s32)ord;
    } else if (k < OCREVT_ONCE || k > OCREVT_CHANNEL) { atomicAdd(&d.ctrl->err[ERR_BAD_OP], 1u); d.evt[x] = EvtRec{0, 0, (u32)ST_ABSENT, 0}; return; }
    d.evt[x] = EvtRec{0, 0, ST_LIVE | (k << 8) | hi, k == OCREVT_LATCH ? arg[i] : arg_val};
    u64 g = gin ? gin[i] : mkguid(d.location, evt_kind_to_guid_kind(k), first_counter + i);
    if (gin && !given_ok(d, g, evt_kind_to_guid_kind(k))) return;
    gtab_put(d, g, (u64)x + 1);
    rc.evt_guids[x] = g;
    if (guids_out) guids_out[i] = g;
}
__global__ void k_db_create(Dev d, ReclDev rc, u32 n_free, u32 n_take, u64 n, const u64 *value, const u64 *tag, u32 first_idx, u64 first_counter,
                            const u64 *gin, u64 *guids_out) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 x = take_index(rc.free_db, n_free, n_take, first_idx, i);
    d.db[x] = make_ulonglong2(value[i], tag ? tag[i] : (u64)x);
    u64 g = gin ? gin[i] : mkguid(d.location, OCRGPU_KIND_DB, first_counter + i);
    // a given DB guid may name another policy domain's DB: this entry is then the local copy of it that satisfies
    // issued here carry by value (the proxy DBs of the reference's distributed PD, hc-dist-policy.c:382-440)
    if (gin && (OCRGPU_GUID_KIND(g) != OCRGPU_KIND_DB || OCRGPU_GUID_COUNTER(g) == 0)) { atomicAdd(&d.ctrl->err[ERR_BAD_OP], 1u); return; }
    gtab_put(d, g, (u64)x + 1);
    if (d.ctrl->db_home) d.ctrl->db_home[x] = g; // (a given guid of another location: this entry is a copy, and what a TILE body does to it goes home)
    if (guids_out) guids_out[i] = g;
}
// OCRGPU_F_WIDE_DB: the tile behind each datablock the call created (same index rule as k_db_create), one thread per word
__global__ void k_tile_init(Dev d, ReclDev rc, u32 n_free, u32 n_take, u64 n, const u64 *value, u32 first_idx) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n * OCROPS_TILE_WORDS) return;
    const u64 j = i / OCROPS_TILE_WORDS; const u32 w = (u32)(i % OCROPS_TILE_WORDS);
    d.ctrl->tiles[(size_t)take_index(rc.free_db, n_free, n_take, first_idx, j) * OCROPS_TILE_WORDS + w] = ocrops_tile_init(value[j], w);
}
// overflow slot words an EDT needs (0 when its pre-slots fit the line)
__global__ void k_ovf_counts(u64 n, const u32 *depc, u32 *ovf) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) ovf[i] = depc[i] > INLINE_SLOTS ? depc[i] : 0u;
}
// slot_ovf[] already holds the exclusive scan of the overflow counts (+ overflow slots in use)
__global__ void k_edt_create(Dev d, ReclDev rc, u32 n_free, u32 n_take, u64 n, const u32 *depc, const u32 *body, const u64 *param0, const u32 *slot_ovf,
                             u32 first_idx, u64 first_counter, const u64 *gin, const u64 *oevt_in,
                             u64 *guids_out, u64 *outevt_out, u64 *outdb_out) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 e = take_index(rc.free_edt, n_free, n_take, first_idx, i), dc = depc[i], b = body[i] & 0xFFFFu;
    EdtGuids eg = {0, 0, 0, 0};
    if (OCRBODY_KIND(b) != OCRBODY_NOP && OCRBODY_KIND(b) != OCRBODY_HASH) atomicAdd(&d.ctrl->async_block, 1u); // FWD0 / HOST / TILE bodies
    if ((OCRBODY_KIND(b) == OCRBODY_TILE && (dc > INLINE_SLOTS || !d.ctrl->tiles)) || (d.ctrl->dbl && dc > LOCKABLE_MAX_DEPC)) { // a TILE body reads its (<= 5) pre-slots from the line, and needs tile datablocks // acq_walk / release_all are quadratic in depc: refuse what would not finish (the EDT does not exist)
        atomicAdd(&d.ctrl->err[ERR_BAD_OP], 1u); EdtLine Z; memset(&Z, 0, sizeof Z); Z.out_state = (u32)ST_ABSENT; d.edt[e] = Z; rc.edt_guids[e] = eg;
        if (!gin) { guids_out[i] = 0; if (outevt_out) outevt_out[i] = 0; if (outdb_out) outdb_out[i] = 0; }
        return;
    }
    EdtLine L;
    L.pending = dc | ((d.ctrl->dbl && dc) ? PEND_ACQ : 0u); L.level = 0; L.depc = dc; L.body = b;
    L.param0 = param0 ? param0[i] : (u64)e; L.val = 0;
    L.out_row = 0; L.out_deg = 0;
    L.out_state = (b & OCRBODY_F_OUTEVT) ? (ST_LIVE | (OCREVT_ONCE << 8)) : (u32)ST_ABSENT; // ONCE output event (hc-task.c:940-959)
    for (u32 k = 0; k < INLINE_SLOTS; ++k) L.slot[k] = 0;
    if (dc > INLINE_SLOTS) L.slot[0] = slot_ovf[i];
    d.edt[e] = L;
    if (gin) { // caller-chosen guids: the EDT's, and (optionally) its output event's; no guid for an output DB
        if (given_ok(d, gin[i], OCRGPU_KIND_EDT)) { gtab_put(d, gin[i], (u64)e + 1); eg.edt = gin[i]; }
        if ((b & OCRBODY_F_OUTEVT) && oevt_in && oevt_in[i] && given_ok(d, oevt_in[i], OCRGPU_KIND_EVT_ONCE)) { gtab_put(d, oevt_in[i], (u64)(d.max_evt + e) + 1); eg.oevt = oevt_in[i]; }
    } else {
        u64 g = mkguid(d.location, OCRGPU_KIND_EDT, first_counter + 3 * i);
        gtab_put(d, g, (u64)e + 1);
        guids_out[i] = g;
        u64 ge = 0;
        if (b & OCRBODY_F_OUTEVT) {
            ge = mkguid(d.location, OCRGPU_KIND_EVT_ONCE, first_counter + 3 * i + 1);
            gtab_put(d, ge, (u64)(d.max_evt + e) + 1);
        }
        if (outevt_out) outevt_out[i] = ge;
        u64 gd = mkguid(d.location, OCRGPU_KIND_DB, first_counter + 3 * i + 2);
        gtab_put(d, gd, (u64)(d.max_db + e) + 1);
        if (outdb_out) outdb_out[i] = gd;
        eg.edt = g; eg.oevt = ge; eg.odb = gd;
    }
    rc.edt_guids[e] = eg;
    if (dc == 0) { u32 ri = atomicAdd(&d.ctrl->r_tail, 1u); d.fired[ri] = e; } // fires at creation (hc-task.c:1129-1134)
}

__device__ __forceinline__ void q_push(const Dev &d, Rec r) {
    // one atomic per wavefront for the lanes that are here together (a round's apply queues thousands of records: one atomic each on
    // this single word was most of that kernel's time)
    const u64 m = __ballot(1);
    const u32 lane = __lane_id(), leader = (u32)__ffsll((unsigned long long)m) - 1u;
    u32 base = 0;
    if (lane == leader) base = atomicAdd(&d.ctrl->q_tail, (u32)__popcll(m));
    const u32 qi = __shfl(base, leader) + (u32)__popcll(m & ((1ull << lane) - 1ull));
    if (qi < d.q_cap) d.q[qi] = r; else atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u);
}
__device__ __forceinline__ void bad_op(const Dev &d) { atomicAdd(&d.ctrl->err[ERR_BAD_OP], 1u); }

// resolve a destination guid into (idx, flags, encoded slot): pre-slot k of the EDT (flag INLINE when it lives in the line), or event slot
__device__ bool resolve_dst(const Dev &d, u64 g, u32 slot, u32 &idx, u32 &flags, u32 &slot_out) {
    u64 v = gtab_get(d, g);
    if (!v) return false;
    u32 k = OCRGPU_GUID_KIND(g);
    idx = (u32)(v - 1);
    if (k == OCRGPU_KIND_EDT) {
        u32 depc = d.edt[idx].depc;
        if (slot >= depc) return false;
        if (depc <= INLINE_SLOTS) { flags = RF_DST_EDT | RF_INLINE; slot_out = slot; }
        else { flags = RF_DST_EDT; slot_out = slot; if (slot_out >= (1u << 28)) return false; } // relative to slot[0]
        return true;
    }
    if (!is_event_kind(k)) return false;
    flags = 0; slot_out = slot;
    if (k == OCRGPU_KIND_EVT_LATCH && slot > 1u) // OCRGPU_LATCH_DELTA(k), |k| < 2^28
        return slot - (OCRGPU_LATCH_DELTA_BIAS - (1u << 28)) < (1u << 29);
    return slot < (1u << 28);
}

// registerWaiterEventHcChannel (hc-event.c:2081-2123): registration k of the channel.  True when satisfy k is already
// queued (payload = what it carried: the caller delivers it at once); else the waiter record waits in line.
__device__ __forceinline__ bool chan_register(const Dev &d, u32 x, u64 waiter, u32 &payload) {
    const ChanRec c = d.chan[(u32)*evt_val(d, x)];
    const u64 old = atomicAdd((unsigned long long *)&d.counted[x], 1ull);
    const u32 k = (u32)old, ns = (u32)(old >> 32);
    if (k < ns) { payload = d.chan_sat[c.sat_base + k % c.cap]; return true; }
    if (k - ns >= c.cap) { atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u); return false; } // more than maxGen waiters in line (ASSERT :2005)
    d.waiters[c.wait_base + k % c.cap] = waiter;
    return false;
}

// ocrAddDependence classification (hc-policy.c:2857-3074) of element i.  True: the dependence is a registration on a live event --
// the caller appends {lx, lrec} to the registration log (k_add_dep does it with one atomic per workgroup).
__device__ bool add_dep_one(const Dev &d, u64 i, const u64 *src, const u64 *dst, const u32 *slot, const u32 *mode, u32 nanny, u32 &lx, u64 &lrec) {
    u64 s = src[i];
    u32 didx, dslot, dflag;
    const u64 dg = dst[i];
    if (d.n_loc > 1 && dg != 0 && dg < TOMB - 1 && OCRGPU_GUID_LOCATION(dg) != d.location) {
        // destination owned by another partition: the waiter stays with the (local) source event, the satisfy
        // crosses as a SendRec; the owner resolves slot encoding and mode when it applies it
        const u32 dk = OCRGPU_GUID_KIND(dg);
        const bool r_edt = dk == OCRGPU_KIND_EDT;
        if (!r_edt && !is_event_kind(dk)) { bad_op(d); return false; }
        const u32 xf = ((r_edt && mode[i] == OCRMODE_NULL) ? XF_MODE_NULL : 0u) | (r_edt ? (mode[i] & 0xFu) << XF_ACQMODE_SHIFT : 0u); // (the mode travels: Lockable datablocks)
        if (s == OCRGPU_NULL_GUID) { emit_remote(d, dg, slot[i], xf & ~(0xFu << XF_ACQMODE_SHIFT), 0u); return false; }
        u64 sv = gtab_get(d, s); u32 sk = OCRGPU_GUID_KIND(s);
        if (!sv) { atomicAdd(&d.ctrl->err[is_event_kind(sk) ? ERR_LATE_WAITER : ERR_BAD_OP], 1u); return false; }
        if (sk == OCRGPU_KIND_DB) { emit_remote(d, dg, slot[i], xf, (u32)sv); return false; }
        if (!is_event_kind(sk)) { bad_op(d); return false; }
        u32 x = (u32)(sv - 1);
        const u32 sw = *evt_state(d, x);
        u32 st = sw & 0xFF;
        bool lazy = false;
        if (r_edt && slot[i] > 0 && ((sw >> 8) & 0xFF) == OCREVT_COUNTED && (st == ST_SAT || st == ST_LIVE)) {
            // the EDT's frontier is known where the EDT lives: hold the COUNT back and let its owner say when
            lazy = true;
            atomicAdd((unsigned long long *)&d.counted[x], 1ull);
            emit_remote_raw(d, dg, slot[i], XF_LAZYREG, s, 0ull);
        }
        if (st == ST_SAT) { const u32 pv = (u32)*evt_val(d, x); if (!lazy) counted_late_registration(d, x); emit_remote(d, dg, slot[i], xf, pv); return false; }
        if (st != ST_LIVE) { atomicAdd(&d.ctrl->err[ERR_LATE_WAITER], 1u); return false; }
        if (((sw >> 8) & 0xFF) == OCREVT_CHANNEL) { // the waiter in line is a remote destination
            u32 ri = atomicAdd(&d.ctrl->rdst_tail, 1u);
            if (ri >= d.rdst_cap) { atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u); return false; }
            d.rdst[ri] = RemoteDst{dg, slot[i], xf};
            u32 pay;
            if (chan_register(d, x, ((u64)(RF_REMOTE << 28) << 32) | (u64)ri, pay)) emit_remote(d, dg, slot[i], xf, pay);
            return false;
        }
        u32 ri = wave_append(&d.ctrl->rdst_tail, true);
        if (ri >= d.rdst_cap) { atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u); return false; }
        d.rdst[ri] = RemoteDst{dg, slot[i], xf};
        lx = x; lrec = ((u64)(RF_REMOTE << 28) << 32) | (u64)ri;
        return true;
    }
    if (!resolve_dst(d, dg, slot[i], didx, dflag, dslot)) { bad_op(d); return false; }
    const bool d_edt = (dflag & RF_DST_EDT) != 0;
    u32 mnull = (d_edt && mode[i] == OCRMODE_NULL) ? RF_MODE_NULL : 0u;
    unsigned char *const acq_mode = d.ctrl->acq_mode;
    if (acq_mode && d_edt && s != OCRGPU_NULL_GUID) // registerSignalerTaskHc records the mode (hc-task.c:1373); a NULL source does not
        acq_mode[acq_index(d, didx, (dflag & RF_INLINE) ? 0u : INLINE_SLOTS + 1u, d.edt[didx].slot[0], dslot)] = (unsigned char)mode[i];
    if (s == OCRGPU_NULL_GUID) { q_push(d, Rec{didx, 0u, dslot, dflag}); return false; } // == satisfy(dst, slot, NULL)
    u64 sv = gtab_get(d, s);
    u32 sk = OCRGPU_GUID_KIND(s);
    if (!sv) { atomicAdd(&d.ctrl->err[is_event_kind(sk) ? ERR_LATE_WAITER : ERR_BAD_OP], 1u); return false; }
    if (sk == OCRGPU_KIND_DB) { // mode recorded, then satisfy
        if ((u32)sv - 1 >= d.max_db && !(d.edt[(u32)sv - 1 - d.max_db].body & BODY_VAL_DONE)) atomicAdd(&d.ctrl->async_block, 1u); // the output datablock of an EDT that may never run: k_async would wait for its value
        q_push(d, Rec{didx, (u32)sv, dslot, dflag | mnull}); return false;
    }
    if (!is_event_kind(sk)) { bad_op(d); return false; }
    u32 x = (u32)(sv - 1);
    const u32 sw = *evt_state(d, x);
    u32 st = sw & 0xFF;
    bool lazy = false;
    if (d_edt && slot[i] > 0 && ((sw >> 8) & 0xFF) == OCREVT_COUNTED && (st == ST_SAT || st == ST_LIVE)) {
        // COUNTED source on a pre-slot the EDT's frontier has not reached (an earlier pre-slot is not satisfied yet):
        // the data path registers now, the COUNT waits for the frontier (registerSignalerTaskHc hc-task.c:1391-1405)
        EdtLine *L = &d.edt[didx];
        u32 *sl = (dflag & RF_INLINE) ? L->slot : d.slots + L->slot[0];
        if (!(dflag & RF_INLINE) && !(L->pending & PEND_LAZY)) mark_prefilled_overflow(d, didx);
        u32 f = 0;
        while (f < slot[i] && slot_satisfied(L, sl, L->depc, f)) ++f;
        if (f < slot[i]) {
            lazy = true;
            atomicOr(&L->pending, PEND_LAZY);
            lazy_insert(d, lazy_key(didx, LAZY_FRONT), f);
            lazy_insert(d, lazy_key(didx, slot[i]), x);
            atomicAdd((unsigned long long *)&d.counted[x], 1ull);
        }
    }
    if (st == ST_SAT) { const u32 pv = (u32)*evt_val(d, x); if (!lazy) counted_late_registration(d, x); q_push(d, Rec{didx, pv, dslot, dflag | mnull}); return false; } // registerWaiterEventHcPersist :1143-1155
    if (st != ST_LIVE) { atomicAdd(&d.ctrl->err[ERR_LATE_WAITER], 1u); return false; }    // registerWaiterEventHc :1078-1083
    if (((sw >> 8) & 0xFF) == OCREVT_CHANNEL) {
        const u32 dl = (!d_edt && OCRGPU_GUID_KIND(dst[i]) == OCRGPU_KIND_EVT_LATCH) ? RF_DST_LATCH : 0u;
        u32 pay;
        if (chan_register(d, x, ((u64)(((dflag | mnull | dl) << 28) | dslot) << 32) | (u64)didx, pay)) q_push(d, Rec{didx, pay, dslot, dflag | mnull});
        return false;
    }
    // Payload pre-resolution: when what this dependence will deliver is already known -- nothing (DB_MODE_NULL),
    // or the output DB / NULL_GUID of a built-in HASH / NOP body behind an output event -- write the pre-slot word
    // now and let the satisfy be a bare counter decrement (no per-edge store on the hot path).
    u32 pre = 0;
    if (d_edt && !nanny) {
        bool known = false; u32 word = 0;
        if (mnull) known = true;
        else if (x >= d.max_evt) {
            u32 se = x - d.max_evt, bk = OCRBODY_KIND(d.edt[se].body);
            if (bk == OCRBODY_HASH) { known = true; word = d.max_db + se + 1; }
            else if (bk == OCRBODY_NOP) known = true;
        }
        // The satisfied mark of an inline pre-slot rides in the pending word; pre-filled OVERFLOW pre-slots have none until
        // the EDT's frontier is tracked -- mark_prefilled_overflow() recovers theirs from the registration log when it matters.
        // (an overflow pre-slot too: a satisfy that had to store there would first have to fetch the base from the destination's
        // line, in front of its wave's atomics -- 2.7 % of config 2's edges cost 4 % of a level that way)
        if (known) { pre = RF_PREFILLED; *((dflag & RF_INLINE) ? &d.edt[didx].slot[dslot] : &d.slots[(size_t)d.edt[didx].slot[0] + dslot]) = word; }
    }
    u32 dlatch = (!d_edt && OCRGPU_GUID_KIND(dst[i]) == OCRGPU_KIND_EVT_LATCH) ? RF_DST_LATCH : 0u;
    // k_async takes EDT pre-slots whose payload word is already there and LATCH DECR slots (order-free); anything else is counted
    if (d_edt ? !pre : (!dlatch || dslot != OCRLATCH_DECR)) atomicAdd(&d.ctrl->async_block, 1u);
    lx = x; lrec = ((u64)(((dflag | mnull | pre | dlatch) << 28) | dslot) << 32) | (u64)didx;
    return true;
}
#define ADD_DEP_BLK 1024
__global__ __launch_bounds__(ADD_DEP_BLK) void k_add_dep(Dev d, u64 n, const u64 *src, const u64 *dst, const u32 *slot, const u32 *mode, u32 nanny) {
    __shared__ u32 lds[18];
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    u32 lx = 0; u64 lrec = 0;
    const bool reg = i < n && add_dep_one(d, i, src, dst, slot, mode, nanny, lx, lrec);
    const u32 li = block_append(&d.ctrl->log_tail, reg, lds);
    if (!reg) return;
    if ((u64)li + d.ctrl->chan_wait_used >= d.max_edges) { atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u); return; } // waiters[] top end = CHANNEL rings
    d.log_src[li] = lx; d.log_rec[li] = lrec;
    // the row length is counted HERE, on the line the classification has just read (registerWaiterEventHc's waitersCount++,
    // hc-event.c:1096-1108), not by a pass over the whole log at every seal: that pass (clear + count) was half of the seal
    // (the lanes of a wavefront that register on the FIRST lane's event count as one: a start event with 2M waiters registered in one
    // batch was 2M atomics on one word, 24 ms; the other lanes -- other events -- go one by one)
    const u32 x0 = __builtin_amdgcn_readfirstlane(lx);
    const u64 same = __ballot(lx == x0);
    if (lx != x0) atomicAdd(evt_deg(d, lx), 1u);
    else if (lane_id() == (u32)(__ffsll((long long)same) - 1)) atomicAdd(evt_deg(d, x0), (u32)__popcll(same));
}

__global__ void k_satisfy(Dev d, u64 n, const u64 *evt, const u64 *db, const u32 *slot) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 g = evt[i];
    if (g == OCRGPU_NULL_GUID) return; // ocrEventSatisfy(NULL_GUID, ..) is a no-op (ocr-edt.c:171)
    if (d.n_loc > 1 && g < TOMB - 1 && OCRGPU_GUID_LOCATION(g) != d.location) { // satisfy of a remote event
        u32 pay = 0;
        if (db[i] != OCRGPU_NULL_GUID) { u64 v = gtab_get(d, db[i]); if (!v) { bad_op(d); return; } pay = (u32)v; }
        emit_remote(d, g, slot[i], 0u, pay); return;
    }
    u32 didx, dslot, dflag;
    if (!resolve_dst(d, g, slot[i], didx, dflag, dslot)) { bad_op(d); return; }
    u32 pay = 0;
    if (db[i] != OCRGPU_NULL_GUID) { u64 v = gtab_get(d, db[i]); if (!v || OCRGPU_GUID_KIND(db[i]) != OCRGPU_KIND_DB) { bad_op(d); return; } pay = (u32)v; }
    if (pay && pay - 1 >= d.max_db && !(d.edt[pay - 1 - d.max_db].body & BODY_VAL_DONE)) atomicAdd(&d.ctrl->async_block, 1u); // (as in k_add_dep)
    q_push(d, Rec{didx, pay, dslot, dflag});
}

__global__ void k_evt_destroy(Dev d, u64 n, const u64 *evt) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 v = gtab_get(d, evt[i]);
    if (!v || !is_event_kind(OCRGPU_GUID_KIND(evt[i]))) { bad_op(d); return; }
    u32 old = atomicExch(evt_state(d, (u32)(v - 1)), (u32)ST_DESTROYED);
    if ((old & 0xFF) != ST_LIVE && (old & 0xFF) != ST_SAT) atomicAdd(&d.ctrl->err[ERR_DEAD_EVENT], 1u);
}

__global__ void k_set_scripts(Dev d, u64 n, const u64 *edt, const u32 *begin, const u32 *len, u32 script_base) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 v = gtab_get(d, edt[i]);
    if (!v || OCRGPU_GUID_KIND(edt[i]) != OCRGPU_KIND_EDT) { bad_op(d); return; }
    u32 e = (u32)(v - 1);
    d.edt_script_begin[e] = script_base + begin[i]; d.edt_script_len[e] = len[i];
    if (len[i]) { d.edt[e].body |= BODY_F_SCRIPT; atomicAdd(&d.ctrl->async_block, 1u); } // a body's own satisfies: events whose order matters
}
__global__ void k_script_ops(Dev d, u64 n, const ocrgpu_script_op_t *ops, u32 script_base) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    ocrgpu_script_op_t o = ops[i];
    u32 remote = 0;
    u64 v;
    if (d.n_loc > 1 && o.target != 0 && o.target < TOMB - 1 && OCRGPU_GUID_LOCATION(o.target) != d.location && is_event_kind(OCRGPU_GUID_KIND(o.target))) {
        u32 ri = atomicAdd(&d.ctrl->rdst_tail, 1u);
        if (ri >= d.rdst_cap) { atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u); ri = 0; }
        else d.rdst[ri] = RemoteDst{o.target, o.slot, 0u};
        v = (u64)ri + 1; remote = 0x10000u;
    } else {
        v = gtab_get(d, o.target);
        if (!v || !is_event_kind(OCRGPU_GUID_KIND(o.target))) { bad_op(d); v = 1; }
    }
    u32 pay = (u32)o.payload;
    if (o.payload_kind == OCRGPU_PAY_GUID) { u64 pv = gtab_get(d, o.payload); if (!pv) bad_op(d); pay = (u32)pv; }
    const u32 latch = o.op != OCRGPU_SOP_SATISFY ? 0u : OCRGPU_GUID_KIND(o.target) == OCRGPU_KIND_EVT_LATCH ? SCRIPT_F_LATCH
                    : OCRGPU_GUID_KIND(o.target) == OCRGPU_KIND_EVT_IDEM ? SCRIPT_F_IDEM : 0u;
    d.scripts[script_base + i] = Script{(u32)(v - 1), pay, o.slot, (u32)o.op | ((u32)o.payload_kind << 8) | remote | latch};
}
// One body's ops on one latch apply in program order in the reference (INCR, DECR, DECR on a counter of 1 ends at 0
// and fires once); here every queued record of a level is applied concurrently, so a run like that is folded into a
// single record carrying the net count: same end value, no transient 0 from reordering.  Likewise the first of one
// body's satisfies of an IDEM event is the one that counts (satisfyEventHcPersistIdem, hc-event.c:569-595): the later
// ones are dropped here rather than left to race with it.  One thread per script.
__device__ __forceinline__ s32 latch_slot_delta(u32 slot) {
    return slot == OCRLATCH_INCR ? 1 : slot == OCRLATCH_DECR ? -1 : (s32)(slot - OCRGPU_LATCH_DELTA_BIAS);
}
__global__ void k_script_merge_latch(Dev d, u64 n, const u32 *begin, const u32 *len, u32 script_base) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    Script *sc = d.scripts + script_base + begin[i];
    const u32 m = len[i];
    for (u32 a = 0; a < m; ++a) {
        const bool ra = sc[a].op_kind & 0x10000u;
        const u64 ta = ra ? d.rdst[sc[a].target].guid : (u64)sc[a].target;
        if (sc[a].op_kind & SCRIPT_F_IDEM) {
            for (u32 b = a + 1; b < m; ++b)
                if ((sc[b].op_kind & SCRIPT_F_IDEM) && (bool)(sc[b].op_kind & 0x10000u) == ra &&
                    (ra ? d.rdst[sc[b].target].guid : (u64)sc[b].target) == ta) sc[b].op_kind = SOP_MERGED;
            continue;
        }
        if (!(sc[a].op_kind & SCRIPT_F_LATCH)) continue;
        const u32 slot_a = ra ? d.rdst[sc[a].target].slot : sc[a].slot;
        s32 delta = latch_slot_delta(slot_a);
        bool merged = false;
        for (u32 b = a + 1; b < m; ++b) {
            if (!(sc[b].op_kind & SCRIPT_F_LATCH) || (bool)(sc[b].op_kind & 0x10000u) != ra) continue;
            const u64 tb = ra ? d.rdst[sc[b].target].guid : (u64)sc[b].target;
            if (tb != ta) continue;
            delta += latch_slot_delta(ra ? d.rdst[sc[b].target].slot : sc[b].slot);
            sc[b].op_kind = SOP_MERGED; merged = true;
        }
        if (merged) { if (ra) d.rdst[sc[a].target].slot = OCRGPU_LATCH_DELTA(delta); else sc[a].slot = OCRGPU_LATCH_DELTA(delta); }
    }
}

// Apply satisfies received from other partitions: resolve the guid in OUR table, clone the DB into a proxy
// entry, and queue an ordinary record (the next EVENTS phase does the rest).
// `db_floor` = explicit datablocks created so far (host mirror): proxies grow down from max_db and must not reach them
// the lanes of a wavefront that are here together reserve `1` each on one counter with ONE atomic (a round's apply brings thousands
// of records: one atomic each on a single word is what such a kernel's time goes into)
__device__ __forceinline__ u32 wave_reserve_one(u32 *counter) {
    const u64 m = __ballot(1);
    const u32 lane = __lane_id(), leader = (u32)__ffsll((unsigned long long)m) - 1u;
    u32 base = 0;
    if (lane == leader) base = atomicAdd(counter, (u32)__popcll(m));
    return __shfl(base, leader) + (u32)__popcll(m & ((1ull << lane) - 1ull));
}
// the tile behind a head slot -> tiles[idx]; returns the home guid the record names
__device__ __forceinline__ u64 tile_from_record(const Dev &d, const SendRec *head, u32 idx, u32 first = 0u /* continuation slots below this one are copied by their own lanes (k_exchange_apply_hdr) */) {
    u64 *t = d.ctrl->tiles + OCROPS_TILE_WORDS * (size_t)idx;
    u64 home = 0;
#pragma unroll 11 // (one lane copies the 22 continuation slots of its record: eleven 32-byte loads in flight at a time, not two -- the apply of a
                  // level of tile records was a chain of eleven round trips per record)
    for (u32 k = 0; k + 1 < TILE_REC_SLOTS; ++k) {
        if (k < first) continue;
        const SendRec c = head[1 + k];
        const u64 w[3] = {c.guid, c.value, c.tag};
#pragma unroll
        for (u32 j = 0; j < 3; ++j) { const u32 i = 3 * k + j; if (i < OCROPS_TILE_WORDS) t[i] = w[j]; else if (i == OCROPS_TILE_WORDS) home = w[j]; }
    }
    return home;
}
__device__ __forceinline__ u64 tile_record_home(const SendRec *head) { return head[1 + OCROPS_TILE_WORDS / 3].value; } // word 64 = slot 21, second field
// `avail` = slots from this one to the end of what arrived in its region (a head's continuation slots must lie within it)
// `tile_dst` (k_exchange_apply_hdr): the head of a tile record only says where its tile goes (*tile_dst = index + 1) and copies the
// continuation slots from `in_chunk` on; the slots in front of that are copied by the lanes that hold them (one lane per slot there)
__device__ void apply_record(const Dev &d, const SendRec *rp, u32 avail, u32 db_floor, u32 *tile_dst = nullptr, u32 in_chunk = 0u) {
    const SendRec r = *rp;
    if (r.flags & XF_CONT) return; // part of the head in front of it
    if ((r.flags & XF_HAS_TILE) && (avail < TILE_REC_SLOTS || !d.ctrl->tiles)) { bad_op(d); return; }
    if (r.flags & XF_TILE_WB) { // a copy of one of OUR datablocks was updated elsewhere: its contents come home
        const u64 v = (OCRGPU_GUID_LOCATION(r.guid) == d.location && OCRGPU_GUID_KIND(r.guid) == OCRGPU_KIND_DB && (r.flags & XF_HAS_TILE)) ? gtab_get(d, r.guid) : 0;
        if (!v || v - 1 >= d.max_db) { bad_op(d); return; }
        if (tile_dst) *tile_dst = (u32)v;
        tile_from_record(d, rp, (u32)(v - 1), tile_dst ? in_chunk : 0u);
        return;
    }
    if (r.flags & (XF_ACQ | XF_REL | XF_GRANT)) { // Lockable datablocks: this partition arbitrates the datablock named (ACQ / REL), or asked for one (GRANT)
        if (!d.ctrl->dbl) { bad_op(d); return; }
        if (r.flags & XF_GRANT) { if (r.value >= d.max_edt) { bad_op(d); return; } q_push(d, Rec{(u32)r.value, 0u, 0u, RF_ACQUIRE}); return; } // the EDT goes on acquiring
        const u64 v = (OCRGPU_GUID_LOCATION(r.guid) == d.location && OCRGPU_GUID_KIND(r.guid) == OCRGPU_KIND_DB) ? gtab_get(d, r.guid) : 0;
        if (!v || v - 1 >= d.max_db) { bad_op(d); return; }
        if (r.flags & XF_REL) { // releases and acquisitions never share a launch (the invariant the lock-free record rests on): noted here,
            const u32 i = atomicAdd(&d.ctrl->rrel_n, 1u); // done by k_remote_releases right behind this kernel
            if (i < d.ctrl->rrel_cap) d.ctrl->rrel[i] = v - 1; else atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u);
            return;
        }
        const u32 n = atomicAdd(&d.ctrl->rwait_used, 1u);
        if (n >= d.ctrl->rwait_cap) { atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u); return; }
        d.ctrl->rwait[n] = RWait{(u32)(r.value >> 32), (u32)r.value, 0u, 0u};
        if (db_acquire(d, (u32)(v - 1), r.slot, d.max_edt + 1u + n)) grant_remote(d, d.max_edt + 1u + n);
        return;
    }
    if ((r.flags & XF_HAS_HOME) && avail < HOME_REC_SLOTS) { bad_op(d); return; }
    if (r.flags & XF_DESTROY) {
        u64 v = (OCRGPU_GUID_LOCATION(r.guid) == d.location && is_event_kind(OCRGPU_GUID_KIND(r.guid))) ? gtab_get(d, r.guid) : 0;
        if (!v) { bad_op(d); return; }
        u32 old = atomicExch(evt_state(d, (u32)(v - 1)), (u32)ST_DESTROYED);
        if ((old & 0xFF) != ST_LIVE && (old & 0xFF) != ST_SAT) atomicAdd(&d.ctrl->err[ERR_DEAD_EVENT], 1u);
        return;
    }
    if (r.flags & XF_ACTIVATE) {
        u64 v = (OCRGPU_GUID_LOCATION(r.guid) == d.location && OCRGPU_GUID_KIND(r.guid) == OCRGPU_KIND_EVT_COUNTED) ? gtab_get(d, r.guid) : 0;
        if (!v) { bad_op(d); return; }
        counted_activate(d, (u32)(v - 1));
        return;
    }
    // (the proxy entry of a record that carries a datablock is reserved in front of the table lookup: the two trips overlap)
    u32 pay = 0;
    const u64 home = (r.flags & XF_HAS_TILE) ? tile_record_home(rp) : (r.flags & XF_HAS_HOME) ? rp[1].guid : 0ull;
    if (home && OCRGPU_GUID_LOCATION(home) == d.location) { // one of OUR datablocks comes back (a tile: with what was done to it elsewhere): no copy, the original
        const u64 v = gtab_get(d, home);
        if (!v || v - 1 >= d.max_db) { bad_op(d); return; }
        if (r.flags & XF_HAS_TILE) { if (tile_dst) *tile_dst = (u32)v; tile_from_record(d, rp, (u32)(v - 1), tile_dst ? in_chunk : 0u); }
        pay = (u32)v;
    } else if ((r.flags & XF_HAS_DB) && !(r.flags & XF_LAZYREG)) { // only a record that carries a datablock takes a proxy entry
        const u32 pi = wave_reserve_one(&d.ctrl->proxy_used);
        if ((u64)pi + 1 + db_floor > d.max_db) { atomicAdd(&d.ctrl->err[ERR_Q_OVERFLOW], 1u); return; }
        const u32 proxy = d.max_db - 1u - pi;
        d.db[proxy] = make_ulonglong2(r.value, r.tag); pay = proxy + 1;
        if (r.flags & XF_HAS_TILE) { if (tile_dst) *tile_dst = proxy + 1u; tile_from_record(d, rp, proxy, tile_dst ? in_chunk : 0u); }
        if (d.ctrl->db_home) d.ctrl->db_home[proxy] = home; // (0: the output datablock of an EDT -- contended only where a copy lives)
    }
    u32 didx, dslot, dflag;
    if (OCRGPU_GUID_LOCATION(r.guid) != d.location || !resolve_dst(d, r.guid, r.slot, didx, dflag, dslot)) { bad_op(d); return; }
    if (r.flags & XF_LAZYREG) { // same decision as k_add_dep makes for a local COUNTED source
        if (!(dflag & RF_DST_EDT)) { bad_op(d); return; }
        EdtLine *L = &d.edt[didx];
        u32 *sl = (dflag & RF_INLINE) ? L->slot : d.slots + L->slot[0];
        if (!(dflag & RF_INLINE) && !(L->pending & PEND_LAZY)) mark_prefilled_overflow(d, didx);
        u32 f = 0;
        while (f < r.slot && slot_satisfied(L, sl, L->depc, f)) ++f;
        if (f < r.slot) {
            atomicOr(&L->pending, PEND_LAZY);
            lazy_insert(d, lazy_key(didx, LAZY_FRONT), f);
            lazy_insert(d, lazy_key(didx, r.slot), r.value);
        } else emit_remote_raw(d, r.value, 0u, XF_ACTIVATE, 0ull, 0ull);
        return;
    }
    if ((dflag & RF_DST_EDT) && (r.flags & XF_MODE_NULL)) dflag |= RF_MODE_NULL;
    if (d.ctrl->acq_mode && (dflag & RF_DST_EDT) && ((r.flags >> XF_ACQMODE_SHIFT) & 0xFu)) // registerSignalerTaskHc records the mode (hc-task.c:1373): it came with the record
        d.ctrl->acq_mode[acq_index(d, didx, (dflag & RF_INLINE) ? 0u : INLINE_SLOTS + 1u, d.edt[didx].slot[0], dslot)] = (unsigned char)((r.flags >> XF_ACQMODE_SHIFT) & 0xFu);
    q_push(d, Rec{didx, pay, dslot, dflag});
}
// The releases a round brought in (XF_REL), in a launch of their own behind the apply: lockableRelease for each, hand-overs included
// (a local EDT in line goes on acquiring in the next EVENTS launch, a remote requester gets its grant with the next exchange).
__global__ void k_remote_releases(Dev d) {
    const u32 n = d.ctrl->rrel_n < d.ctrl->rrel_cap ? d.ctrl->rrel_n : d.ctrl->rrel_cap;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        db_release(d, (u32)d.ctrl->rrel[i], [&](u32 w) { if (w > d.max_edt) grant_remote(d, w); else q_push(d, Rec{w - 1u, 0u, 0u, RF_ACQUIRE}); });
}
__global__ void k_remote_releases_done(Dev d) { d.ctrl->rrel_n = 0; }
__global__ void k_exchange_apply(Dev d, u64 n, const SendRec *recs, u32 db_floor) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) apply_record(d, recs + i, (u32)(n - i > 0xFFFFFFFFull ? 0xFFFFFFFFull : n - i), db_floor);
}
// one receive buffer of n_src regions of `stride` records (an equal-split all-to-all): region s holds cnt[s] records,
// cnt[n_src + s] = records in the regions before it
__global__ void k_exchange_apply_strided(Dev d, u32 n_src, u64 stride, u32 max_cnt, const u32 *cnt, const SendRec *recs, u32 db_floor) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    u32 s = (u32)(i / max_cnt), k = (u32)(i % max_cnt);
    if (s >= n_src || k >= cnt[s]) return;
    apply_record(d, recs + ((u64)s * stride + k), cnt[s] - k, db_floor);
}
// OCRGPU_STEP_WRITE_HEADERS: {records for this destination, my activity} in the last record slot of every region
// (end != 0: the run's k_end rides along on lane 0 -- it folds Ctrl.state[K & 1] into the between-runs words and changes nothing the headers read)
#define HDR_BEHIND (1ull << 40) // in header word 3 (sums: how many partitions): the step was to fire a level and has not got to it
// records queued for all destinations together, by the whole workgroup (one lane alone would make n_loc dependent trips to memory:
// 8 us of the push kernel at 8 partitions); every lane gets the sum
__device__ __forceinline__ u64 records_sent_block(const Dev &d, unsigned long long *lds /* one word */) {
    if (threadIdx.x == 0) *lds = 0;
    __syncthreads();
    unsigned long long part = 0;
    for (u32 l = threadIdx.x; l < d.n_loc; l += blockDim.x) { u32 c = d.send_cnt[l * 32]; part += c > d.send_cap - 1 ? d.send_cap - 1 : c; }
    if (part) atomicAdd(lds, part);
    __syncthreads();
    return *lds;
}
__device__ __forceinline__ void header_words(const Dev &d, u32 K, u32 loc, u64 *h, u64 sent) {
    const State *S = &d.ctrl->state[K & 1];
    u32 mine = d.send_cnt[loc * 32]; if (mine > d.send_cap - 1) mine = d.send_cap - 1;
    d.send_cnt[loc * 32 + 1] = mine; // what this round carries away: exchange_finish keeps what is queued behind it
    // (a step that was launched ahead without a poll may have run out of launches before it was done: that counts as "queued",
    // so that every partition comes back for a settle round)
    // activity = EDTs fired by this step + records sent + "EDTs are ready here" (a partition that caught up late, outside the
    // step its headers describe, still keeps the loop going through what it left ready)
    const RunView v = run_view(d, K); // (as the launch that only notices the end would have left it)
    h[0] = mine; h[1] = (u64)(S->r_end - S->r_begin) + sent + (v.r_total > S->r_end ? 1u : 0u); h[2] = sent; h[3] = (v.q_total > S->q_end ? v.q_total - S->q_end : 0) + (v.over ? 0u : 1u)
           + ((S->allow_fire != 0u && S->allow_fire != 0xFFFFFFFFu && !S->done) ? HDR_BEHIND : 0ull); // a level was asked for and has not fired here yet
}
__global__ void k_exchange_headers(Dev d, u32 K, u32 end = 0) {
    const u32 loc = threadIdx.x;
    __shared__ unsigned long long lds_sent;
    if (end && loc == 0) end_of_run(d, K);
    const u64 sent = records_sent_block(d, &lds_sent);
    if (loc >= d.n_loc) return;
    header_words(d, K, loc, (u64 *)&d.send[(size_t)loc * d.send_cap + (d.send_cap - 1)], sent);
}
// the receive side of the equal-split exchange, counts taken from the headers on the device (no host read in between):
// region s of `recs` (stride records, header in the last slot) carries hdr[0] records from partition s
// reset + header sums + report: the send counters start the next step at zero, the sums of the received headers and the state /
// error words go straight into the pinned page the host reads after its one synchronisation.  Any workgroup size.
__device__ void exchange_finish(const Dev &d, u32 K, u32 n_src, u64 stride, const SendRec *recs, u32 *pin, unsigned long long *sum /* LDS, [4] */) {
    const u32 t = threadIdx.x;
    if (t < 4) sum[t] = 0;
    __syncthreads();
    // The send regions start over -- behind what THIS round's apply queued in them while it ran (an XF_ACTIVATE answering a lazy
    // registration, the grant of a Lockable datablock that was free): those records sit behind the ones the round carried away
    // (send_cnt[+1], noted when the headers were written) and move to the front.  (Setting the counters to zero dropped them.)
    for (u32 l = t; l < d.n_loc; l += blockDim.x) {
        u32 cur = d.send_cnt[l * 32]; if (cur > d.send_cap - 1) cur = d.send_cap - 1;
        const u32 gone = d.send_cnt[l * 32 + 1], keep = cur > gone ? cur - gone : 0u;
        SendRec *R = d.send + (size_t)l * d.send_cap;
        for (u32 i = 0; i < keep; ++i) R[i] = R[gone + i];
        d.send_cnt[l * 32] = keep; d.send_cnt[l * 32 + 1] = 0;
    }
    for (u32 x = t; x < n_src; x += blockDim.x) {
        const u64 *h = (const u64 *)&recs[(u64)x * stride + (stride - 1)];
        for (u32 k = 0; k < 4; ++k) if (h[k]) atomicAdd(&sum[k], (unsigned long long)h[k]);
    }
    __syncthreads();
    if (t < 4) reinterpret_cast<u64 *>(pin + 33040)[t] = sum[t];
    const Ctrl *c = d.ctrl;
    if (t < 32) pin[t] = t == 6 ? (run_view(d, K).over ? 1u : 0u) : reinterpret_cast<const u32 *>(&c->state[K & 1])[t]; // word 6 = State.done
    if (t < 8) pin[33020 + t] = __hip_atomic_load(&c->err[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (other workgroups' atomics of this launch)
    if (t == 0) pin[33058] = c->nc_sends;
}
// workgroups of the push kernel per destination: 16 for 2 partitions, 8 for 4, 4 from 8 on (measured with 2 / 4 / 8 partitions on one
// GPU: every workgroup ends in a system-scope fence, and 128 of those cost more than the parallel copy gains)
__host__ __device__ __forceinline__ u32 peer_push_blocks(u32 n_loc) { const u32 b = 32u / (n_loc ? n_loc : 1u); return b > 16u ? 16u : b < 4u ? 4u : b; }
#define PIN_PEER_LATE 33050u
struct PeerDev { char *const *base; char *mine; u32 me; u32 *cnt; }; // cnt[p] = push workgroups done for p; cnt[1024] = waits that gave up (sticky)
__device__ __forceinline__ u64 *peer_arrival(char *base, u32 src) { return (u64 *)(base + 64 * (size_t)src); }
// The apply of a round.  Peer windows (`win` = my window): the workgroups of source s first WAIT for s's arrival word of this
// round (system-scope acquire loads, bounded by `limit` ticks of the 100 MHz wall clock; a wait that gives up is counted in
// words[1024], which is sticky: nothing is waited for or applied after it, and the host call fails).  The LAST workgroup to
// finish does the round's exchange_finish, and for rounds queued ahead of the host (`ra_on`) takes the host loop's decisions
// and starts the next round's run (begin_auto) -- wait, apply, finish and begin in ONE launch.
#define PIN_RA 33052u // pin[PIN_RA..+6) = {RA_STOP_AT, RA_LEVELS, RA_HOLD, rounds applied, Ctrl.noop_steps, RA_SHORT} after each round queued ahead
__global__ void k_exchange_apply_hdr(Dev d, u32 n_src, u64 stride, const SendRec *recs, u32 db_floor, u32 K, u32 *pin, u32 *words,
                                     char *win, u64 round, u64 limit, u32 ra_on, u32 no_cascade, u32 next_K0) {
    const u32 per = gridDim.x / n_src; // workgroups per source: the grid is n_src * per, whatever the regions' capacity
    const u32 s = blockIdx.x / per, part = blockIdx.x % per;
    __shared__ u32 ok, last;
    if (threadIdx.x == 0) {
        ok = 1;
        if (win) {
            if (__hip_atomic_load(&words[1024], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = 0;
            else {
                const u64 *f = peer_arrival(win, s);
                const u64 t0 = wall_clock64();
                while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < round + 1) {
                    if (wall_clock64() - t0 > limit) { ok = 0; atomicAdd(&words[1024], 1u); break; }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
        }
    }
    __syncthreads();
    if (ok) {
        u64 cnt = recs[(u64)s * stride + (stride - 1)].guid; // header word 0
        if (cnt > stride - 1) cnt = stride - 1;
        if (!d.ctrl->tiles) {
            for (u64 k = part * (u64)blockDim.x + threadIdx.x; k < cnt; k += per * (u64)blockDim.x) apply_record(d, recs + ((u64)s * stride + k), (u32)(cnt - k), db_floor);
        } else {
            // Tile records (23 slots): one lane per slot, and the lane of a head used to copy all 22 continuation slots itself -- 704 bytes
            // read by one lane while 22 sat idle; a level of 40 000 tile records took the apply to a millisecond.  Now the head says where
            // its tile goes and every continuation slot of its 256-slot chunk is copied by its own lane (coalesced); the head keeps only
            // the slots that lie past the end of the chunk.
            __shared__ u32 tile_of[256];
            for (u64 base = part * (u64)blockDim.x; base < cnt; base += per * (u64)blockDim.x) { // (workgroup-uniform: barriers inside)
                const u64 k = base + threadIdx.x;
                const SendRec *rp = recs + ((u64)s * stride + k);
                u32 td = 0;
                if (k < cnt) apply_record(d, rp, (u32)(cnt - k), db_floor, &td, 255u - threadIdx.x);
                tile_of[threadIdx.x] = td;
                __syncthreads();
                if (k < cnt) {
                    const SendRec c = *rp;
                    if ((c.flags & XF_CONT) && c.slot + 1u < TILE_REC_SLOTS && threadIdx.x > c.slot) {
                        const u32 t = tile_of[threadIdx.x - 1u - c.slot];
                        if (t) {
                            u64 *tw = d.ctrl->tiles + OCROPS_TILE_WORDS * (size_t)(t - 1u);
                            const u64 w[3] = {c.guid, c.value, c.tag};
#pragma unroll
                            for (u32 j = 0; j < 3; ++j) { const u32 i = 3u * c.slot + j; if (i < OCROPS_TILE_WORDS) tw[i] = w[j]; }
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    // (every wave waits for its own stores -- the explicit s_waitcnt: a workgroup-scope release compiles to no vector-memory wait on
    // this target -- the barrier joins them, ONE lane releases for the workgroup at device scope: a release is cumulative over what
    // happens before it, and a fence per wave -- each a write-back of the L2 -- was most of this kernel's tail)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence(); last = atomicAdd(&words[1025], 1u) == gridDim.x - 1 ? 1u : 0u; if (last) { words[1025] = 0; __threadfence(); } }
    __syncthreads();
    if (!last) return;
    __shared__ unsigned long long sum[4];
    const u32 late = win ? __hip_atomic_load(&words[1024], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    if (!late) exchange_finish(d, K, n_src, stride, recs, pin, sum);
    if (threadIdx.x != 0) return;
    if (win) pin[PIN_PEER_LATE] = late;
    if (!ra_on) return;
    u32 *ra = words; // the host loop's decisions (resolve_exchange_group), taken here
    const u32 kind = ra[RA_KIND];
    // OCRGPU_RX_NO_CASCADE fires the next level without a settle round, so the records of a level arrive at a partition that has
    // not fired that level yet if its launches went into the previous level's arrivals: what they make ready would fire a level
    // early.  (With settle rounds every queue is empty when a level fires.)  The host-driven loop finishes such a step with polls
    // before the exchange; rounds queued ahead cannot, and stop with an error instead of mislabelling.
    const u32 behind = (u32)(sum[3] >> 40);
    if (!late && kind == 1u && no_cascade && behind != 0u && behind != n_src) { ra[RA_STOP] = 1; ra[RA_BEHIND] = 1; }
    if (!late && no_cascade && d.ctrl->nc_sends) { ra[RA_STOP] = 1; ra[RA_BEHIND] = 2; } // arrivals set off records: they would be a round late (this run is not in NO_CASCADE's class)
    if (late) ra[RA_STOP] = 1;
    else if (kind != 2u) {
        u64 act = *reinterpret_cast<u64 *>(&ra[RA_ACT]) + sum[1];
        const bool pending = (kind == 1u && no_cascade) ? sum[3] != 0 : (sum[2] != 0 || sum[3] != 0);
        if (pending) { ra[RA_HOLD] = 1; *reinterpret_cast<u64 *>(&ra[RA_ACT]) = act; }
        else { ra[RA_HOLD] = 0; if (act == 0) ra[RA_STOP] = 1; else ra[RA_LEVELS] += 1; *reinterpret_cast<u64 *>(&ra[RA_ACT]) = 0; }
    }
    const u32 round_no = ra[RA_ROUND]; ra[RA_ROUND] = round_no + 1;
    if (ra[RA_STOP] && !ra[RA_STOP_AT]) ra[RA_STOP_AT] = round_no + 1;
    pin[PIN_RA + 1] = ra[RA_LEVELS]; pin[PIN_RA + 2] = ra[RA_BEHIND] == 2u ? 0xBAD0BAD1u : ra[RA_BEHIND] ? 0xBAD0BAD0u : ra[RA_HOLD]; pin[PIN_RA + 3] = round_no + 1; // (stores only: nothing is read back over PCIe)
    pin[PIN_RA + 4] = __hip_atomic_load(&d.ctrl->noop_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pin[PIN_RA + 5] = ra[RA_SHORT];
    pin[PIN_RA] = ra[RA_STOP_AT]; // 0 = still running; else 1 + the round that ended the run
    if (next_K0) begin_auto(d, next_K0, ra);
}
__global__ void k_exchange_reset(Dev d) { if (threadIdx.x < d.n_loc) d.send_cnt[threadIdx.x * 32] = 0; }

// ---- peer windows: the exchange of one node without a collective.  Every partition owns a WINDOW in its GPU's memory (fine-grained,
//      so that what peers write over xGMI is seen by the kernels that read it): one arrival word per source partition (64-byte
//      stride), then two sets (round parity) of n_loc receive regions of send_cap records.  Each partition maps every other
//      partition's window (hipIpcOpenMemHandle across processes, the plain pointer within one) and, per round, a kernel of the
//      SENDER stores the records it has for p -- only those, plus the header -- into region [parity][me] of p's window and then
//      p's arrival word [me] = round + 1; the RECEIVER's apply kernel waits for its arrival words first (bounded; the workgroups of
//      source s wait for s's word).  Two parities are enough: a sender starts round r + 2 only after it has seen every
//      partition's arrival for r + 1, which each of them stored after its own apply of round r (stream order). ----
__host__ __device__ __forceinline__ size_t peer_regions_off(u32 n_loc) { return ((64 * (size_t)n_loc + 255) / 256) * 256; }
__device__ __forceinline__ SendRec *peer_region(char *base, u32 n_loc, u64 cap, u32 parity, u32 src) {
    return (SendRec *)(base + peer_regions_off(n_loc)) + ((size_t)parity * n_loc + src) * cap;
}
__global__ void k_peer_push(Dev d, PeerDev pd, u64 round, u32 K, u32 end, u32 wave_fence) {
    const u32 per_dest = peer_push_blocks(d.n_loc);
    const u32 p = blockIdx.x / per_dest, part = blockIdx.x % per_dest;
    if (end && blockIdx.x == 0 && threadIdx.x == 0) end_of_run(d, K); // (k_end rides along: it changes nothing that is read here)
    u32 cnt = d.send_cnt[p * 32]; if (cnt > d.send_cap - 1) cnt = d.send_cap - 1;
    const uint4 *src = (const uint4 *)(d.send + (size_t)p * d.send_cap);
    uint4 *dst = (uint4 *)peer_region(pd.base[p], d.n_loc, d.send_cap, (u32)(round & 1), pd.me);
    for (u64 i = part * (u64)blockDim.x + threadIdx.x; i < 2 * (u64)cnt; i += per_dest * (u64)blockDim.x) dst[i] = src[i];
    if (part == 0) { // the header goes straight into the window (uniform branch: the whole workgroup adds up what was sent)
        __shared__ unsigned long long lds_sent;
        const u64 sent = records_sent_block(d, &lds_sent);
        if (threadIdx.x == 0) header_words(d, K, p, (u64 *)(dst + 2 * (u64)(d.send_cap - 1)), sent);
    }
    // Every workgroup releases its stores at SYSTEM scope before it is counted (fence - atomic - atomic - fence: the last workgroup's
    // fence in front of the arrival word then orders everybody's records before it, for an observer on another GPU too).  Such a
    // fence writes the L2 back each time, which is why a destination gets few workgroups when there are many destinations, and why
    // ONE lane issues it for the workgroup.  That lane's fence waits for ITS wave's stores only, and a workgroup-scope release
    // compiles to no wait at all on this target (the waves of a workgroup share one vector L1), so every wave first waits until
    // its own stores have been acknowledged by the L2 -- the explicit s_waitcnt below -- and only then joins the barrier: when
    // lane 0 writes the L2 back, all 256 lanes' records are in it.
    // OCRGPU_PEER_FENCE=wave: every wave releases at system scope itself (the conservative form, for a machine on which the
    // shared release turns out not to be enough between two devices).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave_fence) __threadfence_system();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) __threadfence_system();
    if (threadIdx.x == 0 && atomicAdd(&pd.cnt[p], 1u) == per_dest - 1) { // the last of p's workgroups: everybody's stores are out
        pd.cnt[p] = 0;
        __threadfence_system();
        __hip_atomic_store(peer_arrival(pd.base[p], pd.me), round + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_exchange_counts(Dev d, u32 *out) { if (threadIdx.x < d.n_loc) { u32 c = d.send_cnt[threadIdx.x * 32]; out[threadIdx.x] = c > d.send_cap - 1 ? d.send_cap - 1 : c; } }

// SCHED_GET_WORK for EDTs whose bodies run on the host (hcSchedulerHeuristicGetEdt, hc-scheduler-heuristic.c:148-199,
// feeding taskExecute hc-task.c:1864): compact the OCRBODY_HOST EDTs of fired[begin,end) with what their body needs --
// paramv[0] and depv[] as {guid, value} of each pre-slot's datablock (ocrEdtDep_t, hc-task.c:680-690).
__global__ void k_pop_ready(Dev d, u32 begin, u32 end, u32 max_edts, u32 max_deps, u32 *cnt, u64 *param0_out, u32 *depc_out,
                            u32 *dep_first_out, u64 *dep_guid_out, u64 *dep_value_out) {
    for (u64 i = begin + blockIdx.x * (u64)blockDim.x + threadIdx.x; i < end; i += (u64)gridDim.x * blockDim.x) {
        u32 e = d.fired[i];
        if (e == INVALID) continue;
        const EdtLine *L = &d.edt[e];
        if (OCRBODY_KIND(L->body) != OCRBODY_HOST) continue;
        u32 depc = L->depc;
        u32 pos = atomicAdd(&cnt[0], 1u), dpos = atomicAdd(&cnt[1], depc);
        if (pos >= max_edts || (u64)dpos + depc > max_deps) continue; // the caller sees the totals and comes back with room
        param0_out[pos] = L->param0; depc_out[pos] = depc; dep_first_out[pos] = dpos;
        const u32 *sl = depc <= INLINE_SLOTS ? L->slot : d.slots + L->slot[0];
        for (u32 k = 0; k < depc; ++k) {
            u32 p = sl[k] & 0x7FFFFFFFu;
            ulonglong2 b = make_ulonglong2(0, 0);
            if (p) { b = db_gather(d, p); if (p - 1 >= d.max_db) b.y = 0; } // explicit DBs carry their guid in the tag
            dep_guid_out[dpos + k] = b.y; dep_value_out[dpos + k] = b.x;
        }
    }
}

// ---- CSR (re)build from the registration log: count -> scan -> fill --------------------------------
// (row lengths are kept current by k_add_dep; an event record starts at 0 when it is created, and snapshot / restore carry the
// lengths with the records, in step with Ctrl.log_tail.  Only LIVE events' rows are filled and ever read.)
__global__ void k_csr_gather_deg(Dev d, u32 n, u32 *deg) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    u32 m = 0;
    if (i < n) { // rows for LIVE events only (what k_csr_fill fills, and all that is ever read): an event that has fired keeps no room in waiters[]
        u32 *dp = evt_deg(d, i);
        u32 v = *dp;
        if (v && (*evt_state(d, i) & 0xFF) != ST_LIVE) { v = 0; *dp = 0; }
        deg[i] = v;
        if (i >= d.max_evt) m = v; // (rows of EDT output events: the ones k_async fans out in numbers)
    }
    for (int o = 1; o < 64; o <<= 1) { const u32 x = __shfl_xor(m, o); m = x > m ? x : m; }
    if (lane_id() == 0 && m > AROW_TRACK) atomicMax(&d.ctrl->max_row, m); // (only rows that matter to the choice of path cost an atomic)
}
// the first AROW waiters of every EDT's output event, beside the lines and indexed by EDT (k_async: one trip for line and row)
__global__ void k_irow_fill(Dev d, u64 *irow, u32 n_edt) {
    const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edt) return;
    const u32 row = d.edt[e].out_row, deg = d.edt[e].out_deg;
#pragma unroll
    for (u32 k = 0; k < AROW; ++k) irow[(size_t)e * AROW + k] = k < deg ? d.waiters[row + k] : 0ull;
}
__global__ void k_csr_set_rows(Dev d, u32 n, const u32 *row, u32 *cursor) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { *evt_row(d, i) = row[i]; cursor[i] = row[i]; }
}
__global__ void k_csr_fill(Dev d, u32 *cursor) {
    u32 n = d.ctrl->log_tail; if (n > d.max_edges) n = (u32)d.max_edges;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        u32 x = d.log_src[i];
        if ((*evt_state(d, x) & 0xFF) != ST_LIVE) continue;
        // the lanes whose entry belongs to the first lane's event take their places with ONE atomic (the waiters of a start event or of
        // a power-law hub sit side by side in the log: one atomic each on that event's cursor was tens of ms per seal)
        const u32 x0 = __builtin_amdgcn_readfirstlane(x);
        const u64 same = __ballot(x == x0);
        u32 pos;
        if (x != x0) pos = atomicAdd(&cursor[x], 1u);
        else {
            const int leader = __ffsll((long long)same) - 1;
            u32 base = 0;
            if ((int)lane_id() == leader) base = atomicAdd(&cursor[x0], (u32)__popcll(same));
            pos = __shfl(base, leader) + rank_in(same);
        }
        d.waiters[pos] = d.log_rec[i];
    }
}

// ---- exclusive scan (u32), 1024 elements per workgroup, recursive over the partials -------------------
#define SCAN_ITEMS 4
__global__ __launch_bounds__(BLK) void k_scan_block(const u32 *in, u32 *out, u32 *partial, u64 n, u32 add0) {
    __shared__ u32 ws[4];
    u64 base = (u64)blockIdx.x * (BLK * SCAN_ITEMS) + (u64)threadIdx.x * SCAN_ITEMS;
    u32 v[SCAN_ITEMS], sum = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = (base + k < n) ? in[base + k] : 0u; sum += v[k]; }
    u32 total;
    u32 incl = block_inclusive_scan(sum, ws, total);
    u32 run = incl - sum + add0;
    for (int k = 0; k < SCAN_ITEMS; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 0 && partial) partial[blockIdx.x] = total;
}
__global__ void k_scan_add(u32 *out, const u32 *partial_scanned, u64 n) {
    u64 i = (u64)blockIdx.x * (BLK * SCAN_ITEMS) + threadIdx.x;
    u32 add = partial_scanned[blockIdx.x];
    for (int k = 0; k < SCAN_ITEMS; ++k, i += BLK) if (i < n) out[i] += add;
}
__global__ void k_last_sum(const u32 *in, const u32 *scanned, u64 n, u32 *out) { if (n) *out = scanned[n - 1] + in[n - 1]; else *out = 0; }

// ---- GUID provider kernels ----------------------------------------------------------------------------------
__global__ void k_guid_reserve(Dev d, u32 kind, u64 n, u64 first_counter, u64 *out) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) out[i] = mkguid(d.location, kind, first_counter + i);
}
__global__ void k_guid_put(Dev d, u64 n, const u64 *g, const u64 *v) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) { if (g[i] == 0 || g[i] >= TOMB - 1 || v[i] == 0) bad_op(d); else gtab_put(d, g[i], v[i]); }
}
__global__ void k_guid_tryput(Dev d, u64 n, const u64 *g, const u64 *v, u64 *held) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (g[i] == 0 || g[i] >= TOMB - 1 || v[i] == 0) { bad_op(d); held[i] = 0; } else held[i] = gtab_tryput(d, g[i], v[i]);
}
__global__ void k_guid_get(Dev d, u64 n, const u64 *g, u64 *v) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 r = gtab_get(d, g[i]);
    // lazy deletion: an engine-made event guid stops resolving once the event is destroyed
    // (the reference's destructEventHc -> PD_MSG_GUID_DESTROY -> releaseGuid, hc-event.c:185-242)
    if (r && OCRGPU_GUID_LOCATION(g[i]) == d.location && is_event_kind(OCRGPU_GUID_KIND(g[i])) && r <= (u64)d.max_evt + d.max_edt) {
        u32 st = *evt_state(d, (u32)(r - 1)) & 0xFF;
        if (st == ST_DESTROYED || st == ST_ABSENT) r = 0;
    }
    v[i] = r;
}
__global__ void k_guid_remove(Dev d, u64 n, const u64 *g) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n && g[i] != 0 && g[i] < TOMB - 1) gtab_remove(d, g[i]);
}

// ---- reclaiming capacity ---------------------------------------------------------------------------------------
// ocrDbDestroy (regular-datablock.c:388-470 region -> PD_MSG_GUID_DESTROY): the guid stops resolving and the index goes on the
// free stack; for the output datablock of a HASH EDT (it lives in the EDT's line) the line is marked, and is taken back by
// the next ocrgpu_reclaim.  Consumers that still hold the index in a pre-slot would read the next tenant: as in the
// reference, destroying a datablock somebody is still to use is the program's error.
__global__ void k_db_destroy(Dev d, ReclDev rc, u32 n_free, u32 n_db, u64 n, const u64 *g) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 v = gtab_get(d, g[i]);
    if (!v || OCRGPU_GUID_KIND(g[i]) != OCRGPU_KIND_DB) { bad_op(d); return; }
    if (!gtab_remove(d, g[i])) return; // named twice in this batch: the other element takes it back
    const u32 idx = (u32)(v - 1);
    if (idx >= d.max_db) { atomicOr(&d.edt[idx - d.max_db].body, BODY_DB_DEAD); return; }
    if (idx < n_db) rc.free_db[n_free + atomicAdd(&rc.cnt[RC_DB], 1u)] = idx; // (proxy copies at the top of db[] go with Ctrl.proxy_used)
}
// EDTs that still wait for pre-slots and track a frontier (held-back COUNTED registrations): while there are any, lazy_tab is
// in use and lines that used it stay where they are
__global__ void k_reclaim_lazy_users(Dev d, ReclDev rc, u32 n_edt) {
    u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = false;
    if (e < n_edt) {
        const EdtLine *L = &d.edt[e];
        const u32 pd = L->pending;
        live = !(L->body & BODY_FREE) && (pd & PEND_LAZY) && (pd & (L->depc <= INLINE_SLOTS ? PEND_INL_CNT : PEND_MASK)) != 0;
    }
    const u64 m = __ballot(live);
    if (m && lane_id() == (u32)(__ffsll((long long)m) - 1)) atomicAdd(&rc.cnt[RC_LAZY_LIVE], (u32)__popcll(m));
}
// One thread per entry of held[0,n_held) and of the ready log fired[0,r_end): an EDT that has fired, whose body has run (host
// bodies: handed out by ocrgpu_pop_ready before this call), whose ONCE output event is gone and whose output datablock has
// been destroyed gives its line and its guids back (destructTaskHc, hc-task.c:721-790).  What cannot go yet is kept in
// held_new; host-body EDTs not handed out yet stay in the ready log (`unpopped`).
#define RECL_BLK 1024
__global__ __launch_bounds__(RECL_BLK) void k_reclaim_edts(Dev d, ReclDev rc, u32 n_held, u32 r_end, u32 pop_cursor, u32 n_free, u32 *unpopped) {
    __shared__ u32 lds[18];
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    u32 e = INVALID;
    bool freed = false, held = false;
    if (i < (u64)n_held + r_end) {
        const bool from_log = i >= n_held;
        e = from_log ? d.fired[i - n_held] : rc.held[i];
        if (e != INVALID) {
            EdtLine *L = &d.edt[e];
            const u32 pending = L->pending, body = L->body, out_state = L->out_state, kind = OCRBODY_KIND(body);
            if (body & BODY_FREE) { }
            else if (kind == OCRBODY_HOST && from_log && i - n_held >= pop_cursor) unpopped[atomicAdd(&rc.cnt[RC_UNPOPPED], 1u)] = e;
            else {
                bool ok = true;
                if ((body & OCRBODY_F_OUTEVT) && (out_state & 0xFF) != ST_DESTROYED) ok = false;   // its output event has not fired yet
                if (kind == OCRBODY_HASH && !(body & BODY_DB_DEAD)) ok = false;                     // its output datablock is still alive
                if ((pending & PEND_LAZY) && rc.cnt[RC_LAZY_LIVE] != 0) ok = false;                 // lazy_tab entries of this line cannot be dropped yet
                if (!ok) held = true;
                else {
                    const EdtGuids g = rc.edt_guids[e];
                    if (g.edt) gtab_remove(d, g.edt);
                    if (g.oevt) gtab_remove(d, g.oevt);
                    if (g.odb) gtab_remove(d, g.odb);
                    rc.edt_guids[e] = EdtGuids{0, 0, 0, 0};
                    if (pending & PEND_LAZY) atomicAdd(&rc.cnt[RC_LAZY_FREED], 1u);
                    L->pending = 0u; L->level = 0u; L->body = BODY_FREE; L->out_state = (u32)ST_ABSENT;
                    freed = true;
                }
            }
        }
    }
    // one atomic per workgroup and list (block_append): 50M lines going back one wavefront at a time spent 24 ms on rc.cnt[RC_EDT] alone
    const u32 hi = block_append(&rc.cnt[RC_HELD], held, lds);
    if (held) rc.held_new[hi] = e;
    const u32 fi = block_append(&rc.cnt[RC_EDT], freed, lds);
    if (freed) rc.free_edt[n_free + fi] = e;
}
// destructEventHc (hc-event.c:185-242): a destroyed event's record and guid go back
__global__ void k_reclaim_evts(Dev d, ReclDev rc, u32 n_evt, u32 n_free) {
    const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_evt || (d.evt[x].state_kind & 0xFF) != ST_DESTROYED) return;
    const u64 g = rc.evt_guids[x];
    if (g) gtab_remove(d, g);
    rc.evt_guids[x] = 0;
    d.evt[x].state_kind = (u32)ST_ABSENT;
    rc.free_evt[n_free + atomicAdd(&rc.cnt[RC_EVT], 1u)] = x;
}
// what the bump-allocated pools (overflow pre-slots, script ops) still serve: none left = the pool starts over
__global__ void k_reclaim_pool_users(Dev d, ReclDev rc, u32 n_edt) {
    u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    bool ovf = false, scr = false;
    if (e < n_edt) {
        const u32 bw = d.edt[e].body;
        if (!(bw & BODY_FREE)) { ovf = d.edt[e].depc > INLINE_SLOTS; scr = (bw & BODY_F_SCRIPT) != 0; }
    }
    const u64 mo = __ballot(ovf), ms = __ballot(scr);
    if (lane_id() == 0) { if (mo) atomicAdd(&rc.cnt[RC_OVF_LIVE], (u32)__popcll(mo)); if (ms) atomicAdd(&rc.cnt[RC_SCRIPT_LIVE], (u32)__popcll(ms)); }
}
// The registration log keeps the entries whose source event is still LIVE -- exactly what seal() would build rows from --
// in their order (flags -> scan -> scatter into a copy -> back).  Entries of fired or destroyed sources would otherwise
// come back to life under the next tenant of the source's index.
__global__ void k_log_flags(Dev d, u32 *flags) {
    u32 n = d.ctrl->log_tail; if (n > d.max_edges) n = (u32)d.max_edges;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        flags[i] = (*evt_state(d, d.log_src[i]) & 0xFF) == ST_LIVE ? 1u : 0u;
}
__global__ void k_log_scatter(Dev d, const u32 *flags, const u32 *pos, u32 *tsrc, u64 *trec) {
    u32 n = d.ctrl->log_tail; if (n > d.max_edges) n = (u32)d.max_edges;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (flags[i]) { tsrc[pos[i]] = d.log_src[i]; trec[pos[i]] = d.log_rec[i]; }
}
__global__ void k_log_back(Dev d, const u32 *tsrc, const u64 *trec, const u32 *kept) {
    const u32 n = *kept;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { d.log_src[i] = tsrc[i]; d.log_rec[i] = trec[i]; }
}
// The ready log starts over: [host-body EDTs that fired but were not handed out yet][EDTs that are ready and have not fired].
__global__ void k_ready_gather(Dev d, ReclDev rc, u32 r_end, u32 r_tail, u32 *ready) {
    __shared__ u32 lds[18];
    const u64 span = (u64)r_tail - r_end, stride = (u64)gridDim.x * blockDim.x;
    for (u64 o = 0; o < span; o += stride) { // every thread of a workgroup makes the same number of trips (block_append has barriers)
        const u64 i = o + blockIdx.x * (u64)blockDim.x + threadIdx.x;
        const u32 e = i < span ? d.fired[r_end + i] : INVALID;
        const u32 ri = block_append(&rc.cnt[RC_READY], e != INVALID, lds);
        if (e != INVALID) ready[ri] = e;
    }
}
__global__ void k_ready_back(Dev d, ReclDev rc, const u32 *unpopped, const u32 *ready) {
    const u32 nu = rc.cnt[RC_UNPOPPED], n = nu + rc.cnt[RC_READY];
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d.fired[i] = i < nu ? unpopped[i] : ready[i - nu];
}
__global__ void k_reclaim_finish(Dev d, ReclDev rc, u32 compact_log) {
    Ctrl *c = d.ctrl;
    c->r_end = rc.cnt[RC_UNPOPPED]; c->r_tail = rc.cnt[RC_UNPOPPED] + rc.cnt[RC_READY];
    c->level = 0xFFFFFFFFu; c->n_levels = 0; c->b_begin = c->b_end = 0; // level numbering starts over with the log
    if (compact_log) c->log_tail = rc.cnt[RC_LOG_KEPT];
}

// What the host polls, written by ONE small kernel straight into the pinned page it reads (host-coherent memory mapped into
// the device: the stores are visible when the stream synchronises).  Each hipMemcpyAsync of a few bytes costs a copy packet
// with its own barrier and completion signal (~10 us on the stream); a level of the stepping loops used three to five.
#define RP_STATE 1u    // pin[0..31]          = Ctrl.state[K & 1]
#define RP_SENDCNT 2u  // pin[32 + 32 * loc]  = send counter of partition loc
#define RP_ERR 4u      // pin[PIN_ERR..+8)    = Ctrl.err[]
#define RP_QUEUE 8u    // pin[33008], [33009] = Ctrl.q_begin, q_tail
#define RP_REND 16u    // pin[33010], [33011] = Ctrl.r_end, r_tail
#define RP_ASYNC 32u   // pin[33040..33044]   = Ctrl.async_block, max_row, b_begin, b_end, level
// (pin[33040..+8) = the four header sums of an exchange round, u64 each: written by exchange_finish)
__global__ void k_report(Dev d, u32 K, u32 what, u32 *pin) {
    const u32 t = threadIdx.x;
    const Ctrl *c = d.ctrl;
    if ((what & RP_STATE) && t < 32) pin[t] = reinterpret_cast<const u32 *>(&c->state[K & 1])[t];
    if ((what & RP_SENDCNT) && d.send_cnt) for (u32 l = t; l < d.n_loc; l += blockDim.x) pin[32 + 32 * l] = d.send_cnt[l * 32];
    if ((what & RP_ERR) && t < 8) pin[33020 + t] = c->err[t];
    if ((what & RP_QUEUE) && t == 0) { pin[33008] = c->q_begin; pin[33009] = c->q_tail; }
    if ((what & RP_REND) && t == 0) { pin[33010] = c->r_end; pin[33011] = c->r_tail; }
    if ((what & RP_ASYNC) && t == 0) { pin[33040] = c->async_block; pin[33041] = c->max_row; pin[33042] = c->b_begin; pin[33043] = c->b_end; pin[33044] = c->level; }
}

// ---- result readback ------------------------------------------------------------------------------------
__global__ void k_fill_i32(s32 *p, u64 n, s32 v) { u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
__global__ void k_levels(Dev d, u64 first, u64 n, s32 *level_out) {
    u32 nf = d.ctrl->r_end, nl = d.ctrl->n_levels;
    if (nl > d.max_levels) nl = d.max_levels;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += gridDim.x * blockDim.x) {
        u32 e = d.fired[i];
        if (e == INVALID || e < first || e >= first + n) continue;
        u32 lo = 0, hi = nl; // last level whose start <= i
        while (hi - lo > 1) { u32 mid = (lo + hi) >> 1; if (d.level_start[mid] <= i) lo = mid; else hi = mid; }
        level_out[e - first] = (s32)lo;
    }
}
__global__ void k_read_vals(Dev d, u64 first, u64 n, const s32 *level, u64 *val_out) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) val_out[i] = level[i] >= 0 ? d.edt[first + i].val : 0; // stale after a restore unless fired
}
__global__ void k_read_events(Dev d, u32 first, u64 n, u64 *out) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    struct { u32 state_kind; s32 val; } m = {*evt_state(d, first + (u32)i), *evt_val(d, first + (u32)i)};
    u32 st = m.state_kind & 0xFF, kind = (m.state_kind >> 8) & 0xFF;
    u64 r;
    if (st == ST_ABSENT) r = OCREVS(OCREVS_ABSENT, 0);
    else if (st == ST_DESTROYED) r = OCREVS(OCREVS_DESTROYED, 0);
    else if (st == ST_SAT) r = OCREVS(OCREVS_LIVE_SAT, m.val ? (u32)(db_gather(d, (u32)m.val).y + 1) : 0u);
    else if (kind == OCREVT_CHANNEL) { // queued satisfies << 16 | waiters in line
        const u64 a = d.counted[first + (u32)i]; const u32 ns = (u32)(a >> 32), nw = (u32)a;
        r = OCREVS(OCREVS_LIVE_UNSAT, ((ns > nw ? ns - nw : 0u) << 16) | (nw > ns ? nw - ns : 0u));
    }
    else r = OCREVS(OCREVS_LIVE_UNSAT, kind == OCREVT_LATCH ? (u32)m.val : 0u);
    out[i] = r;
}
__global__ void k_read_dbs(Dev d, u32 first, u64 n, const s32 *level, u32 first_edt, u64 *out) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (level) { // output DBs exist once their EDT has fired with a HASH body
        bool live = level[i] >= 0 && OCRBODY_KIND(d.edt[first_edt + i].body) == OCRBODY_HASH;
        out[i] = live ? d.edt[first_edt + i].val : 0;
    } else if (d.ctrl->tiles) out[i] = ocrops_tile_checksum(d.ctrl->tiles + 64 * (size_t)(first + i)); // a tile datablock reports its contents
    else out[i] = d.db[first + i].x;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
#define HIPCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) { snprintf(g_err, sizeof g_err, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); return OCRGPU_EHIP; } } while (0)
#define FAIL(code, ...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return (code); } while (0)

struct Snapshot { u64 *tiles; DbLock *dbl; EdtLine *edt; EvtRec *evt; u64 *counted; ulonglong2 *lazy_tab; u32 *slots; u64 n_slots; u64 *chan_wait; u32 *chan_sat; u32 chan_wait_used, chan_sat_used; u32 *fired; Ctrl *ctrl; u32 n_edt, n_evt, n_db, n_fired_entries;
                  u64 edges_upper, seal_epoch, err_seen, mut_epoch; bool valid; };
static void snap_free(Snapshot &s) {
    if (s.dbl) (void)hipFree(s.dbl);
    s.dbl = nullptr;
    if (s.tiles) (void)hipFree(s.tiles);
    s.tiles = nullptr;
    (void)hipFree(s.edt); (void)hipFree(s.evt); (void)hipFree(s.counted); (void)hipFree(s.lazy_tab); (void)hipFree(s.slots); (void)hipFree(s.chan_wait); (void)hipFree(s.chan_sat); (void)hipFree(s.fired); (void)hipFree(s.ctrl);
    s.valid = false;
}

#define PIN_ERR 33020u // u32 index into pinned_done of the 8 error counters the last poll brought back
struct ocrgpu_engine {
    u64 err_seen;  // sum of Ctrl.err[] the caller has been told about (OCRGPU_EFAULT)
    int device; hipStream_t stream; hipStream_t own_stream /* ocrgpu_stream_create: destroyed with the engine */; ocrgpu_config_t cfg; Dev d;
    std::vector<void *> allocs;
    // host mirrors
    u32 n_edt, n_evt, n_db; u64 n_slots, n_script, script_cap, lazy_cap; u64 counter; u64 edges_upper;
    bool csr_dirty; u32 q_cap; u64 seal_epoch;
    // staging
    char *stage_dev; size_t stage_dev_cap, stage_dev_off; char *stage_host; size_t stage_host_cap;
    u32 *tmp_a, *tmp_b, *tmp_c; // scan temporaries sized for max(max_edt, n_evt_total)
    u32 *scan_partials; u64 scan_partials_cap;
    Snapshot snap;
    hipEvent_t ev0, ev1; bool have_events; u32 step_grid; u32 launch_index; u64 fired_cap; bool defer_drain; u32 *send_cnt_dev;
    ocrgpu_stats_t last; u64 launches_total, step_ns_total;
    u32 *pinned_done, *pinned_dev; u32 apply_flip; hipEvent_t apply_ev[2]; // the polled page and the device's address of it (k_report)
    DbLock *dbl; unsigned char *acq_mode; // OCRGPU_F_LOCKABLE_DB: the device tables (the device finds them through Ctrl)
    bool irow_stale;                      // the rows were rebuilt since k_async's inline rows (ADev.irow) were last filled
    RWait *rwait; u64 *rrel;              // OCRGPU_F_LOCKABLE_DB with partitions (through Ctrl as well)
    u64 *tiles, *db_home;                 // OCRGPU_F_WIDE_DB: 64 words per explicit datablock, and the guid each was created under (through Ctrl as well)
    u32 pop_cursor, snap_pop_cursor; // ocrgpu_pop_ready: fired[] entries before this were handed out (or are built-ins)
    // comm-platform over RCCL (ocrgpu_comm_*): communicator of this partition, receive regions, a small device word block
    void *comm; int comm_rank, comm_nranks; SendRec *recv_buf; u64 recv_cap_records;
    // peer windows (ocrgpu_peer_*): my window, the device table of everybody's, what was opened from other processes, the round
    // counter every partition advances together, the wait limit in wall-clock ticks; hdr_fresh: the step just queued wrote the headers
    char *peer_win; size_t peer_win_bytes; char **peer_base_dev; std::vector<void *> peer_opened; u32 *peer_cnt; u64 peer_round, peer_limit; u32 peer_fence_wave; bool peer_on, hdr_fresh, ahead_timed, end_pending, poisoned /* a peer wait gave up: the level may be partly applied */; hipEvent_t ra_ev[2]; u32 ra_launches, ra_noop_seen, ra_short_seen, ra_rounds_seen, ra_cool; // rounds queued ahead: this engine's step launches per round, sized by what the device reports
    u64 *wp; u64 wp_cap, wp_head, wp_tail; // workpile: ring of guids in HBM, indices kept by the (single) caller side
    // reclaimed capacity (ocrgpu_reclaim, ocrgpu_db_destroy): device tables, host mirrors of the free-stack heights and of the
    // held list's length, a grow-on-demand temporary for the log compaction; mut_epoch moves whenever indices change hands
    ReclDev rc; u32 n_free_edt, n_free_evt, n_free_db, n_held; u64 mut_epoch; char *recl_tmp; size_t recl_tmp_cap;
    // the barrier-free resolve (k_async): its tables (null = this engine never takes that path), launch shape, runs so far
    ADev ad; u32 async_waves; u64 async_runs; int async_mode; u32 async_max_row; // async_mode: 0 never, 1 when the DAG allows and looks narrow (default), 2 whenever it allows
};

static int dalloc(ocrgpu_engine *e, void **p, size_t bytes, bool zero = true) {
    if (bytes == 0) bytes = 16;
    HIPCHK(hipMalloc(p, bytes));
    e->allocs.push_back(*p);
    if (zero) HIPCHK(hipMemsetAsync(*p, 0, bytes, e->stream));
    return 0;
}
#define DALLOC(ptr, T, count) do { int _r = dalloc(e, (void **)&(ptr), sizeof(T) * (size_t)(count)); if (_r) return _r; } while (0)

static inline u32 nblk(u64 n, u32 b = 256) { return (u32)((n + b - 1) / b); }

// staging arena: host arrays -> device copies for the duration of one call
static int stage_reset(ocrgpu_engine *e) { e->stage_dev_off = 0; return 0; }
static int stage_reserve(ocrgpu_engine *e, size_t bytes) {
    if (bytes <= e->stage_dev_cap) return 0;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->stage_dev) HIPCHK(hipFree(e->stage_dev));
    size_t cap = bytes + (bytes >> 2) + 4096;
    HIPCHK(hipMalloc((void **)&e->stage_dev, cap));
    e->stage_dev_cap = cap;
    return 0;
}
template <typename T> static int stage_in(ocrgpu_engine *e, const T *p, u64 n, int where, const T **out) {
    if (p == nullptr) { *out = nullptr; return 0; }
    if (where == OCRGPU_DEVICE) { *out = p; return 0; }
    size_t bytes = (sizeof(T) * n + 255) & ~(size_t)255;
    if (e->stage_dev_off + bytes > e->stage_dev_cap) FAIL(OCRGPU_ENOMEM, "staging arena too small (internal)");
    T *dp = (T *)(e->stage_dev + e->stage_dev_off);
    e->stage_dev_off += bytes;
    HIPCHK(hipMemcpyAsync(dp, p, sizeof(T) * n, hipMemcpyHostToDevice, e->stream));
    *out = dp;
    return 0;
}
template <typename T> static int stage_out_alloc(ocrgpu_engine *e, T *p, u64 n, int where, T **out) {
    if (p == nullptr) { *out = nullptr; return 0; }
    if (where == OCRGPU_DEVICE) { *out = p; return 0; }
    size_t bytes = (sizeof(T) * n + 255) & ~(size_t)255;
    if (e->stage_dev_off + bytes > e->stage_dev_cap) FAIL(OCRGPU_ENOMEM, "staging arena too small (internal)");
    *out = (T *)(e->stage_dev + e->stage_dev_off);
    e->stage_dev_off += bytes;
    return 0;
}
template <typename T> static int stage_out_copy(ocrgpu_engine *e, T *host, const T *dev, u64 n, int where) {
    if (host == nullptr || where == OCRGPU_DEVICE) return 0;
    HIPCHK(hipMemcpyAsync(host, dev, sizeof(T) * n, hipMemcpyDeviceToHost, e->stream));
    return 0;
}
static size_t rup(size_t b) { return (b + 255) & ~(size_t)255; }

// exclusive scan of in[0..n) into out (+add0); total (optional, device u32) receives the sum
static int scan_u32(ocrgpu_engine *e, const u32 *in, u32 *out, u64 n, u32 add0, u32 *total_dev) {
    if (n == 0) { if (total_dev) HIPCHK(hipMemsetAsync(total_dev, 0, 4, e->stream)); return 0; }
    const u64 per = BLK * SCAN_ITEMS;
    u64 nb = (n + per - 1) / per;
    if (nb == 1) {
        k_scan_block<<<1, BLK, 0, e->stream>>>(in, out, nullptr, n, add0);
    } else {
        // partials live in scan_partials, levels packed one after the other
        u64 need = 0; for (u64 m = nb; ; m = (m + per - 1) / per) { need += 2 * m; if (m == 1) break; }
        if (need > e->scan_partials_cap) {
            HIPCHK(hipStreamSynchronize(e->stream));
            if (e->scan_partials) HIPCHK(hipFree(e->scan_partials));
            HIPCHK(hipMalloc((void **)&e->scan_partials, need * 4 + 1024));
            e->scan_partials_cap = need + 256;
        }
        u32 *part = e->scan_partials, *part_scanned = part + nb;
        k_scan_block<<<(u32)nb, BLK, 0, e->stream>>>(in, out, part, n, add0);
        // scan the partials (recursion on a sub-range of the same buffer)
        struct Tmp { u32 *buf; u64 cap; } save = {e->scan_partials, e->scan_partials_cap};
        e->scan_partials = part_scanned + nb; e->scan_partials_cap = save.cap - 2 * nb;
        int r = scan_u32(e, part, part_scanned, nb, 0, nullptr);
        e->scan_partials = save.buf; e->scan_partials_cap = save.cap;
        if (r) return r;
        k_scan_add<<<(u32)nb, BLK, 0, e->stream>>>(out, part_scanned, n);
    }
    if (total_dev) k_last_sum<<<1, 1, 0, e->stream>>>(in, out, n, total_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

static int scratch_setup(int device);
extern "C" int ocrgpu_abi_version(void) { return 6; } // 6: ocrgpu_stream_create; 5: tile and Lockable datablocks across partitions (records of several slots), a peer wait that gave up refuses further steps, no-alloc busy flag
extern "C" const char *ocrgpu_last_error(void) { return g_err; }

extern "C" int ocrgpu_init(ocrgpu_engine_t **out, int device, const ocrgpu_config_t *cfg) {
    if (!out || !cfg) FAIL(OCRGPU_EINVAL, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) FAIL(OCRGPU_ENODEV, "no HIP device: libocrgpu has no CPU path");
    if (device < 0 || device >= ndev) FAIL(OCRGPU_EINVAL, "device %d out of range (%d devices)", device, ndev);
    if ((u64)cfg->max_edt >= (1u << 30) || (u64)cfg->max_evt + cfg->max_edt >= (1u << 30)) FAIL(OCRGPU_EINVAL, "at most 2^30 objects per partition");
    if (cfg->max_edges >= (1ull << 32) || cfg->max_slots >= (1ull << 31)) FAIL(OCRGPU_EINVAL, "max_edges < 2^32 and max_slots < 2^31 per partition");
    if (cfg->location >= 1024) FAIL(OCRGPU_EINVAL, "location must fit GUID_PROVIDER_LOCID_SIZE=10 bits");
    HIPCHK(hipSetDevice(device));
    { int sr = scratch_setup(device); if (sr) return sr; } // this GPU's address of the static scratch array (ocrgpu_pop_ready_noalloc)
    ocrgpu_engine *e = new ocrgpu_engine();
    e->device = device; e->stream = 0; e->own_stream = nullptr; e->cfg = *cfg; e->err_seen = 0;
    e->comm = nullptr; e->comm_rank = 0; e->comm_nranks = 0; e->recv_buf = nullptr; e->recv_cap_records = 0;
    e->peer_win = nullptr; e->peer_win_bytes = 0; e->peer_base_dev = nullptr; e->peer_cnt = nullptr; e->peer_round = 0; e->peer_limit = 0; e->peer_fence_wave = 0; e->peer_on = false; e->hdr_fresh = false; e->ahead_timed = false; e->end_pending = false; e->poisoned = false; e->ra_ev[0] = e->ra_ev[1] = nullptr; e->ra_launches = 0; e->ra_noop_seen = e->ra_short_seen = e->ra_rounds_seen = e->ra_cool = 0;
    e->wp = nullptr; e->wp_cap = 0; e->wp_head = e->wp_tail = 0;
    memset(&e->rc, 0, sizeof e->rc); e->n_free_edt = e->n_free_evt = e->n_free_db = e->n_held = 0; e->mut_epoch = 0; e->recl_tmp = nullptr; e->recl_tmp_cap = 0;
    e->n_edt = e->n_evt = e->n_db = 0; e->n_slots = 0; e->n_script = 0; e->counter = 1; e->edges_upper = 0;
    e->csr_dirty = false; e->seal_epoch = 0; e->stage_dev = nullptr; e->stage_dev_cap = 0; e->stage_dev_off = 0; e->stage_host = nullptr; e->stage_host_cap = 0;
    e->scan_partials = nullptr; e->scan_partials_cap = 0; e->snap.valid = false; e->snap.dbl = nullptr; e->snap.tiles = nullptr; e->have_events = false;
    e->apply_flip = 0; e->apply_ev[0] = e->apply_ev[1] = nullptr; e->dbl = nullptr; e->acq_mode = nullptr; e->tiles = nullptr; e->db_home = nullptr; e->rwait = nullptr; e->rrel = nullptr; e->irow_stale = true;
    e->launches_total = 0; e->step_ns_total = 0; e->launch_index = 8; e->pop_cursor = e->snap_pop_cursor = 0; e->defer_drain = false; e->send_cnt_dev = nullptr; e->step_grid = getenv("OCRGPU_STEP_GRID") ? (u32)atoi(getenv("OCRGPU_STEP_GRID")) : 2048u; memset(&e->last, 0, sizeof e->last);
    Dev &d = e->d; memset(&d, 0, sizeof d);
    { const char *mp = getenv("OCRGPU_SPREAD_WGS"); u32 v = mp ? (u32)atoi(mp) : 256u; d.spread_wgs = v < 1 ? 1u : (v > MAX_GRID ? (u32)MAX_GRID : v); }
    d.max_edt = cfg->max_edt; d.max_evt = cfg->max_evt; d.max_db = cfg->max_db; d.location = cfg->location;
    d.max_slots = cfg->max_slots ? cfg->max_slots : 1; d.max_edges = cfg->max_edges ? cfg->max_edges : 1;
    d.max_levels = cfg->max_levels ? cfg->max_levels : (1u << 20);
    u64 n_evt_total = (u64)cfg->max_evt + cfg->max_edt, n_db_total = (u64)cfg->max_db + cfg->max_edt;
    // one record per queued satisfy: events that fire, API satisfies (every DB- or NULL-source edge is one: bounded by the
    // pre-slots), script ops; API batches larger than a quarter of it are applied in chunks with a drain between them
    u64 q_cap = 2 * (n_evt_total + cfg->max_edt) + d.max_slots + 4096;
    if (cfg->flags & OCRGPU_F_LOCKABLE_DB) q_cap += 2 * (u64)cfg->max_edt + d.max_slots + 6 * (u64)cfg->max_edt; // one record per EDT + one per hand-over
    if (q_cap > 0xFFFFFFF0ull) q_cap = 0xFFFFFFF0ull;
    d.q_cap = (u32)q_cap; e->q_cap = (u32)q_cap;
    u64 objs = n_evt_total + n_db_total + cfg->max_edt;
    u64 gs = cfg->guid_slots ? cfg->guid_slots : 2 * objs;
    u64 cap = 1024; u32 lg = 10; while (cap < gs) { cap <<= 1; ++lg; }
    d.gmask = cap - 1; d.gshift = 64 - lg;
#define TRY(x) do { int _r = (x); if (_r) { ocrgpu_fini(e); return _r; } } while (0)
    TRY(dalloc(e, (void **)&d.ctrl, sizeof(Ctrl)));
    TRY(dalloc(e, (void **)&d.edt, sizeof(EdtLine) * (size_t)cfg->max_edt));
    TRY(dalloc(e, (void **)&d.edt_script_begin, 4 * (size_t)cfg->max_edt));
    TRY(dalloc(e, (void **)&d.edt_script_len, 4 * (size_t)cfg->max_edt));
    TRY(dalloc(e, (void **)&d.slots, 4 * (size_t)d.max_slots));
    TRY(dalloc(e, (void **)&d.evt, sizeof(EvtRec) * (size_t)cfg->max_evt));
    TRY(dalloc(e, (void **)&d.counted, 8 * (size_t)cfg->max_evt));
    TRY(dalloc(e, (void **)&d.chan, sizeof(ChanRec) * (size_t)cfg->max_evt));
    d.chan_sat_cap = getenv("OCRGPU_CHAN_SLOTS") ? (u32)atoll(getenv("OCRGPU_CHAN_SLOTS")) : (u32)(d.max_edges < (1u << 22) ? d.max_edges : (1u << 22));
    TRY(dalloc(e, (void **)&d.chan_sat, 4 * (size_t)d.chan_sat_cap));
    { // held-back COUNTED registrations: 2 entries each at most; OCRGPU_MAX_LAZY overrides the default bound
        u64 want = getenv("OCRGPU_MAX_LAZY") ? (u64)atoll(getenv("OCRGPU_MAX_LAZY")) : 4 * (u64)cfg->max_evt + 1024;
        if (want > (1ull << 22) && !getenv("OCRGPU_MAX_LAZY")) want = 1ull << 22;
        u64 lc = 4096; while (lc < 4 * want) lc <<= 1;
        d.lazy_mask = lc - 1; e->lazy_cap = lc;
        TRY(dalloc(e, (void **)&d.lazy_tab, 16 * (size_t)lc));
    }
    TRY(dalloc(e, (void **)&d.waiters, 8 * (size_t)d.max_edges));
    TRY(dalloc(e, (void **)&d.log_src, 4 * (size_t)d.max_edges));
    TRY(dalloc(e, (void **)&d.log_rec, 8 * (size_t)d.max_edges));
    TRY(dalloc(e, (void **)&d.db, 16 * (size_t)n_db_total));
    if ((cfg->flags & OCRGPU_F_WIDE_DB) && (cfg->flags & OCRGPU_F_LOCKABLE_DB)) { ocrgpu_fini(e); FAIL(OCRGPU_EINVAL, "OCRGPU_F_WIDE_DB and OCRGPU_F_LOCKABLE_DB cannot be combined"); }
    if (cfg->flags & OCRGPU_F_WIDE_DB) { TRY(dalloc(e, (void **)&e->tiles, 8 * OCROPS_TILE_WORDS * (size_t)cfg->max_db)); TRY(dalloc(e, (void **)&e->db_home, 8 * (size_t)cfg->max_db)); }
    if ((cfg->flags & OCRGPU_F_LOCKABLE_DB) && !e->db_home) TRY(dalloc(e, (void **)&e->db_home, 8 * (size_t)cfg->max_db));
    if (cfg->flags & OCRGPU_F_LOCKABLE_DB) {
        TRY(dalloc(e, (void **)&e->dbl, sizeof(DbLock) * (size_t)n_db_total));
        TRY(dalloc(e, (void **)&e->acq_mode, (size_t)cfg->max_edt * INLINE_SLOTS + (size_t)d.max_slots));
    }
    u64 fired_cap = (u64)cfg->max_edt + d.max_edges + 4096; if (fired_cap > 0xFFFFFFF0ull) fired_cap = 0xFFFFFFF0ull;
    e->fired_cap = fired_cap; d.fired_cap = (u32)fired_cap;
    TRY(dalloc(e, (void **)&d.fired, 4 * (size_t)fired_cap, false));
    TRY(dalloc(e, (void **)&d.level_start, 4 * ((size_t)d.max_levels + 2)));
    TRY(dalloc(e, (void **)&d.q, sizeof(Rec) * (size_t)q_cap, false));
    TRY(dalloc(e, (void **)&d.blk_stats, 8 * 4 * MAX_GRID));
    TRY(dalloc(e, (void **)&d.dbg, 8 * (16 + 8 * 8 * MAX_GRID)));
    { u64 k = getenv("OCRGPU_STAMP_K") ? (u64)atoll(getenv("OCRGPU_STAMP_K")) : 0; (void)hipMemcpyAsync(&d.dbg[15], &k, 8, hipMemcpyHostToDevice, e->stream); (void)hipStreamSynchronize(e->stream); }
    e->script_cap = (u64)cfg->max_edt + 1024;
    TRY(dalloc(e, (void **)&d.scripts, sizeof(Script) * e->script_cap));
    TRY(dalloc(e, (void **)&d.gtab, 16 * (size_t)cap));
    u64 tmpn = n_evt_total > cfg->max_edt ? n_evt_total : cfg->max_edt;
    TRY(dalloc(e, (void **)&e->tmp_a, 4 * (size_t)(tmpn + 4)));
    TRY(dalloc(e, (void **)&e->tmp_b, 4 * (size_t)(tmpn + 4)));
    TRY(dalloc(e, (void **)&e->tmp_c, 4 * (size_t)(tmpn + 4)));
    TRY(dalloc(e, (void **)&e->rc.edt_guids, sizeof(EdtGuids) * (size_t)cfg->max_edt));
    TRY(dalloc(e, (void **)&e->rc.evt_guids, 8 * (size_t)cfg->max_evt));
    TRY(dalloc(e, (void **)&e->rc.free_edt, 4 * (size_t)cfg->max_edt));
    TRY(dalloc(e, (void **)&e->rc.free_evt, 4 * (size_t)cfg->max_evt));
    TRY(dalloc(e, (void **)&e->rc.free_db, 4 * (size_t)cfg->max_db));
    TRY(dalloc(e, (void **)&e->rc.held, 4 * (size_t)cfg->max_edt));
    TRY(dalloc(e, (void **)&e->rc.held_new, 4 * (size_t)cfg->max_edt));
    TRY(dalloc(e, (void **)&e->rc.cnt, 4 * RC_N));
    memset(&e->ad, 0, sizeof e->ad); e->async_runs = 0; e->async_waves = 0;
    { const char *am = getenv("OCRGPU_ASYNC"); e->async_mode = am ? atoi(am) : 1; }
    { const char *mr = getenv("OCRGPU_ASYNC_MAX_ROW"); e->async_max_row = mr ? (u32)strtoul(mr, nullptr, 10) : AMAX_ROW; if (e->async_max_row < AROW_TRACK) e->async_max_row = AROW_TRACK; }
    if ((cfg->flags & (OCRGPU_F_NANNY | OCRGPU_F_LOCKABLE_DB | OCRGPU_F_WIDE_DB | OCRGPU_F_NO_ASYNC)) || d.max_levels > (1u << 24)) e->async_mode = 0;
    if (e->async_mode) {
        const char *w = getenv("OCRGPU_ASYNC_WAVES"), *rf = getenv("OCRGPU_ASYNC_BLOCK");
        u32 blk = rf ? (u32)atoi(rf) : 24u; if (blk < 4) blk = 4; if (blk > 32) blk = 32;
        // entries per shard (8 shards), whole blocks: room for blocks filled to an eighth; a used-up queue makes waves keep what they make ready
        const u64 acap = (((u64)cfg->max_edt + cfg->max_evt) / 2 + 65536) / blk * blk;
        if (acap < 0xFFFFFFF0ull) {
            e->ad.cap = (u32)acap;
            TRY(dalloc(e, (void **)&e->ad.c, sizeof(ACtl)));
            TRY(dalloc(e, (void **)&e->ad.q, 8 * (size_t)ANSH * acap));
            TRY(dalloc(e, (void **)&e->ad.irow, 8 * (size_t)AROW * cfg->max_edt));
            TRY(dalloc(e, (void **)&e->ad.alat, 8 * (size_t)cfg->max_evt));
            TRY(dalloc(e, (void **)&e->ad.hist, 4 * ((size_t)d.max_levels + 4)));
            e->async_waves = w ? (u32)atoi(w) : 1280u; // (5 single-wave workgroups per CU; 1024 .. 1536 measure alike) if (e->async_waves < 8) e->async_waves = 8; if (e->async_waves > 8192) e->async_waves = 8192;
            e->ad.blk = blk; e->ad.refill = blk <= 8 ? 8u : blk <= 16 ? 16u : 32u; // (a power of two: the wave's scans run over the ticket lanes)
        } else e->async_mode = 0;
    }
    // ctrl.level starts at -1 (main-program satisfies are "level -1")
    Ctrl h; memset(&h, 0, sizeof h); h.level = 0xFFFFFFFFu; h.state[0].done = h.state[1].done = 1;
    h.dbl = e->dbl; h.acq_mode = e->acq_mode; h.tiles = e->tiles; h.db_home = e->db_home;
    if (hipMemcpyAsync(d.ctrl, &h, sizeof h, hipMemcpyHostToDevice, e->stream) != hipSuccess) { ocrgpu_fini(e); FAIL(OCRGPU_EHIP, "ctrl init failed"); }
    // u32 index: [0,32) the polled State; [32, 32 + 32*1024) send counters of up to 1024 locations; [33000, 34000) small read-backs
    if (hipHostMalloc((void **)&e->pinned_done, 4 * 34816, hipHostMallocMapped) != hipSuccess) { ocrgpu_fini(e); FAIL(OCRGPU_EHIP, "hipHostMalloc failed"); }
    memset(e->pinned_done, 0, 4 * 34816);
    if (hipHostGetDevicePointer((void **)&e->pinned_dev, e->pinned_done, 0) != hipSuccess) { ocrgpu_fini(e); FAIL(OCRGPU_EHIP, "hipHostGetDevicePointer failed"); }
    if (cfg->flags & OCRGPU_F_TIMING) { if (hipEventCreate(&e->ev0) == hipSuccess && hipEventCreate(&e->ev1) == hipSuccess) e->have_events = true; }
    if (hipStreamSynchronize(e->stream) != hipSuccess) { ocrgpu_fini(e); FAIL(OCRGPU_EHIP, "init sync failed"); }
#undef TRY
    *out = e;
    return 0;
}

extern "C" int ocrgpu_fini(ocrgpu_engine_t *e) {
    if (!e) return 0;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
#ifdef OCRGPU_ASTAMPS
    { u64 g[16]; if (e->async_runs && hipMemcpy(g, e->d.dbg + 32, sizeof g, hipMemcpyDeviceToHost) == hipSuccess && g[0])
        fprintf(stderr, "[k_async, %llu runs] wave-iterations %llu (%.1f items each); per iteration: %.2f us to the end of the hand-over (loads %.2f, fan-out %.2f, hand-over %.2f), body %.2f us (latch edges %.2f, gathers %.2f, %llu second looks); idle polls %llu; atomics back -> taken up by the next wave %.2f us\n",
                (unsigned long long)e->async_runs, (unsigned long long)g[0], (double)g[4] / g[0], g[1] * 0.01 / g[0], g[6] * 0.01 / g[0], g[7] * 0.01 / g[0], g[5] * 0.01 / g[0], g[2] * 0.01 / g[0], g[9] * 0.01 / g[0], g[10] * 0.01 / g[0], (unsigned long long)g[8], (unsigned long long)g[3], g[12] ? g[11] * 0.01 / g[12] : 0.0); }
#endif
#ifdef OCRGPU_STAMPS
    { u64 h[16]; if (e->d.dbg && hipMemcpy(h, e->d.dbg, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
        fprintf(stderr, "[stamps, 10ns ticks of workgroup 0 summed over %llu launches] entry %llu | A(loads) %llu | scan %llu | fanout %llu | body %llu | ticket %llu\n",
                (unsigned long long)e->launches_total, (unsigned long long)h[0], (unsigned long long)h[1], (unsigned long long)h[2], (unsigned long long)h[3], (unsigned long long)h[4], (unsigned long long)h[5]);
        if (getenv("OCRGPU_STAMP_DUMP")) { // raw tables of the 8 stamped launches, for tools/stamp_tail.py
            std::vector<u64> all(16 + 8 * 8 * MAX_GRID);
            if (hipMemcpy(all.data(), e->d.dbg, 8 * all.size(), hipMemcpyDeviceToHost) == hipSuccess) {
                FILE *f = fopen(getenv("OCRGPU_STAMP_DUMP"), "wb"); if (f) { fwrite(all.data(), 8, all.size(), f); fclose(f); }
            }
        }
        std::vector<u64> hb(16 + 8 * MAX_GRID);
        if (hipMemcpy(hb.data(), e->d.dbg, 8 * hb.size(), hipMemcpyDeviceToHost) == hipSuccess) {
            u64 t0 = ~0ull; int n = 0;
            for (int b = 0; b < MAX_GRID; ++b) hb[16 + 8 * b] &= (1ull << 60) - 1; // strip the XCD id
            for (int b = 0; b < MAX_GRID; ++b) { u64 st = hb[16 + 8 * b]; if (st && hb[17 + 8 * b] && st < t0) t0 = st; }
            // per phase: min / median / max over the working workgroups, 10ns ticks after the first entry
            static const char *nm[8] = {"entry", "exit", "decided", "lines", "scan", "fanout", "loopend", "stats"};
            for (int ph = 0; ph < 8; ++ph) {
                std::vector<u64> v;
                for (int b = 0; b < MAX_GRID; ++b) { if (!hb[17 + 8 * b]) continue; u64 t = hb[16 + 8 * b + ph]; if (t >= t0) v.push_back(t - t0); }
                if (v.empty()) continue;
                std::sort(v.begin(), v.end()); n = (int)v.size();
                fprintf(stderr, "[stamps K=%llu] %-8s n=%d min %llu med %llu p90 %llu max %llu\n", (unsigned long long)hb[15], nm[ph], n,
                        (unsigned long long)v[0], (unsigned long long)v[n / 2], (unsigned long long)v[n * 9 / 10], (unsigned long long)v[n - 1]);
            }
        }
} }
#endif
    if (e->comm) (void)ocrgpu_comm_fini(e);
    (void)ocrgpu_peer_fini(e);
    if (e->recv_buf) (void)hipFree(e->recv_buf);
    for (void *p : e->allocs) (void)hipFree(p);
    if (e->stage_dev) (void)hipFree(e->stage_dev);
    if (e->scan_partials) (void)hipFree(e->scan_partials);
    if (e->recl_tmp) (void)hipFree(e->recl_tmp);
    if (e->snap.valid) snap_free(e->snap);
    if (e->pinned_done) (void)hipHostFree(e->pinned_done);
    for (int i = 0; i < 2; ++i) if (e->apply_ev[i]) (void)hipEventDestroy(e->apply_ev[i]);
    for (int i = 0; i < 2; ++i) if (e->ra_ev[i]) (void)hipEventDestroy(e->ra_ev[i]);
    if (e->have_events) { (void)hipEventDestroy(e->ev0); (void)hipEventDestroy(e->ev1); }
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
    return 0;
}

extern "C" int ocrgpu_set_stream(ocrgpu_engine_t *e, void *s) { if (!e) FAIL(OCRGPU_EINVAL, "null engine"); e->stream = (hipStream_t)s; return 0; }
// Several engines in ONE process (partitions that share a GPU, or one host thread driving several GPUs) all launch on the NULL stream
// by default: one queue, every kernel of every partition in line behind the others.  This gives the engine a stream of its own
// (hipStreamNonBlocking; it belongs to the engine and goes with it).  What was queued on the old stream is waited for first.  From
// then on device arrays handed in with OCRGPU_DEVICE must be complete before the call (no implicit order with the NULL stream).
extern "C" int ocrgpu_stream_create(ocrgpu_engine_t *e) {
    if (!e) FAIL(OCRGPU_EINVAL, "null engine");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (!e->own_stream) HIPCHK(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
    e->stream = e->own_stream;
    return 0;
}
extern "C" int ocrgpu_sync(ocrgpu_engine_t *e) { if (!e) FAIL(OCRGPU_EINVAL, "null engine"); HIPCHK(hipStreamSynchronize(e->stream)); return 0; }
extern "C" int ocrgpu_set_hash_tag_base(ocrgpu_engine_t *e, u64 b) { if (!e) FAIL(OCRGPU_EINVAL, "null engine"); e->d.hash_tag_base = b; return 0; }

#define ENTER(e) do { if (!(e)) FAIL(OCRGPU_EINVAL, "null engine"); HIPCHK(hipSetDevice((e)->device)); stage_reset(e); (e)->hdr_fresh = false; } while (0)
// A wait on a peer window that gave up leaves its level partly applied (the regions of the sources that had arrived went in): the
// worker-loop calls refuse from then on, until ocrgpu_restore puts a snapshot back or ocrgpu_peer_connect joins fresh windows (the
// caller then vouches for the state).
#define REFUSE_POISONED(e) do { if ((e)->poisoned) FAIL(OCRGPU_EHIP, "a wait on a peer window gave up earlier and left its level partly applied: ocrgpu_restore a snapshot " \
                                                                    "(or join fresh windows with ocrgpu_peer_connect) before going on"); } while (0)

// ---- CSR seal ---------------------------------------------------------------------------------------------
static int seal(ocrgpu_engine *e) {
    if (!e->csr_dirty) return 0;
    Dev &d = e->d;
    u32 n_evt_total = d.max_evt + d.max_edt; // rows exist for every possible event index
    // only indices in use matter: explicit [0,n_evt) and output events [max_evt, max_evt+n_edt)
    u32 used = d.max_evt + e->n_edt;
    (void)n_evt_total;
    u32 g = nblk(e->edges_upper); if (g > 4096) g = 4096; if (g == 0) g = 1;
    HIPCHK(hipMemsetAsync(&d.ctrl->max_row, 0, 4, e->stream));
    k_csr_gather_deg<<<nblk(used), 256, 0, e->stream>>>(d, used, e->tmp_a);
    int r = scan_u32(e, e->tmp_a, e->tmp_b, used, 0, nullptr);
    if (r) return r;
    k_csr_set_rows<<<nblk(used), 256, 0, e->stream>>>(d, used, e->tmp_b, e->tmp_c);
    k_csr_fill<<<g, 256, 0, e->stream>>>(d, e->tmp_c);
    e->irow_stale = true; // (the inline rows k_async reads are filled when it next runs: a seal on the level-synchronous path does not pay for them)
    HIPCHK(hipGetLastError());
    e->csr_dirty = false; ++e->seal_epoch;
    return 0;
}

// ---- drain: run k_step launches until the device says done ---------------------------------------------
// Launch k_step ahead of the device until it reports `done`.  Every poll brings back the whole phase state and
// (when partitions are configured) the per-destination send counters in ONE synchronisation: pinned[0..31] = State,
// pinned[32 + 32*loc] = records queued for partition loc.
// the k_step variant this engine runs (NANNY x Lockable datablocks, or tile datablocks)
static void launch_step(ocrgpu_engine *e, u32 grid, u32 K) {
    const bool nanny = (e->cfg.flags & OCRGPU_F_NANNY) != 0;
    const Dev &d = e->d;
    if (e->tiles && d.n_loc > 1) k_bodies_pre<<<grid, BLK, 0, e->stream>>>(d, K); // (the bodies of the level launch K fires, if it fires one)
    if (e->tiles && d.n_loc > 1) { if (nanny) k_step<true, 0, 2><<<grid, BLK, 0, e->stream>>>(d, K); else k_step<false, 0, 2><<<grid, BLK, 0, e->stream>>>(d, K); } // tile datablocks that cross partitions
    else if (e->tiles) { if (nanny) k_step<true, 0, 1><<<grid, BLK, 0, e->stream>>>(d, K); else k_step<false, 0, 1><<<grid, BLK, 0, e->stream>>>(d, K); }
    else if (e->dbl && d.n_loc > 1) { if (nanny) k_step<true, 2, 0><<<grid, BLK, 0, e->stream>>>(d, K); else k_step<false, 2, 0><<<grid, BLK, 0, e->stream>>>(d, K); } // datablocks arbitrated by their owners
    else if (e->dbl) { if (nanny) k_step<true, 1, 0><<<grid, BLK, 0, e->stream>>>(d, K); else k_step<false, 1, 0><<<grid, BLK, 0, e->stream>>>(d, K); }
    else if (nanny) k_step<true, 0, 0><<<grid, BLK, 0, e->stream>>>(d, K); else k_step<false, 0, 0><<<grid, BLK, 0, e->stream>>>(d, K);
}
static int run_steps(ocrgpu_engine *e, u32 allow_fire, u64 max_launches, bool timing, u32 force_level = 0, u32 first_batch = 0) {
    Dev &d = e->d;
    int r = seal(e); if (r) return r;
    if (e->launch_index > 0xFFFF0000u) e->launch_index = 8; // keep K-1 and K%3 arithmetic away from the wrap
    u32 K = e->launch_index + 1;
    k_begin<<<1, 1, 0, e->stream>>>(d, K, allow_fire, force_level);
    const bool nanny = (e->cfg.flags & OCRGPU_F_NANNY) != 0;
    const u32 grid = e->step_grid > MAX_GRID ? MAX_GRID : (e->step_grid ? e->step_grid : 256u);
    u32 batch = first_batch ? first_batch : (allow_fire ? 8 : 2);
    u64 launched = 0;
    if (timing && e->have_events) HIPCHK(hipEventRecord(e->ev0, e->stream));
    for (;;) {
        for (u32 k = 0; k < batch; ++k, ++K) {
            launch_step(e, grid, K);
        }
        launched += batch;
        k_report<<<1, 64, 0, e->stream>>>(d, K - 1, RP_STATE | RP_SENDCNT | RP_ERR, e->pinned_dev);
        HIPCHK(hipStreamSynchronize(e->stream));
        if (((State *)e->pinned_done)->done) break;
        if (launched >= max_launches) break;
        if (batch < 64) batch *= 2;
    }
    if (timing && e->have_events) {
        HIPCHK(hipEventRecord(e->ev1, e->stream)); HIPCHK(hipEventSynchronize(e->ev1));
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
        e->step_ns_total += (u64)(ms * 1e6);
    }
    e->launches_total += launched;
    e->launch_index = K - 1;
    k_end<<<1, 1, 0, e->stream>>>(d, K - 1);
    if (!((State *)e->pinned_done)->done) { // stopped by the launch budget: the last level's bodies are still owed, and the caller may read values
        k_owed_bodies<<<grid, BLK, 0, e->stream>>>(d);
        k_owed_clear<<<1, 1, 0, e->stream>>>(d);
    }
    HIPCHK(hipGetLastError());
    return 0;
}
// The device counts user-level errors (Ctrl.err[], ocrgpu_stats_t.err[]) and goes on; the call during which a counter moved
// still does all its work and then returns OCRGPU_EFAULT (once per movement), so that no caller takes a dropped record or a
// malformed op for success.  pinned_done[PIN_ERR..+8) = Ctrl.err as of the last poll.
static int check_fault(ocrgpu_engine *e) {
    u64 sum = 0; for (int i = 0; i < 8; ++i) sum += e->pinned_done[PIN_ERR + i];
    if (sum == e->err_seen) return 0;
    e->err_seen = sum;
    const u32 *x = e->pinned_done + PIN_ERR;
    FAIL(OCRGPU_EFAULT, "the device flagged user-level errors: err = [%u %u %u %u %u %u %u] (double slot, late waiter, sticky re-satisfied, dead event, "
                        "malformed op, queue/table overflow, level overflow); the call completed", x[0], x[1], x[2], x[3], x[4], x[5], x[6]);
}
static int drain_events(ocrgpu_engine *e) {
    if (e->defer_drain) return 0; // distributed stepping: the driver calls ocrgpu_step
    // q_begin, q_tail and err[] sit in different cache lines of Ctrl: three small reads, one sync
    k_report<<<1, 64, 0, e->stream>>>(e->d, 0u, RP_QUEUE | RP_ERR, e->pinned_dev);
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->pinned_done[33009] > e->pinned_done[33008]) { int r = run_steps(e, 0u, ~0ull, false); if (r) return r; }
    return check_fault(e);
}

// ---- GUID provider -----------------------------------------------------------------------------------------
extern "C" int ocrgpu_guid_reserve(ocrgpu_engine_t *e, u32 kind, u64 n, u64 *guids_out, int where) {
    ENTER(e);
    if (!guids_out || kind >= 32) FAIL(OCRGPU_EINVAL, "bad argument");
    if (e->counter + n >= (1ull << 49)) FAIL(OCRGPU_ENOMEM, "GUID counter overflows 49 bits");
    if (n == 0) return 0;
    int r = stage_reserve(e, rup(8 * n)); if (r) return r;
    u64 *dg; r = stage_out_alloc(e, guids_out, n, where, &dg); if (r) return r;
    k_guid_reserve<<<nblk(n), 256, 0, e->stream>>>(e->d, kind, n, e->counter, dg);
    e->counter += n;
    r = stage_out_copy(e, guids_out, dg, n, where); if (r) return r;
    if (where == OCRGPU_HOST) HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
extern "C" int ocrgpu_guid_put(ocrgpu_engine_t *e, u64 n, const u64 *guids, const u64 *vals, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!guids || !vals) FAIL(OCRGPU_EINVAL, "null array");
    int r = stage_reserve(e, 2 * rup(8 * n)); if (r) return r;
    const u64 *dg, *dv;
    if ((r = stage_in(e, guids, n, where, &dg)) || (r = stage_in(e, vals, n, where, &dv))) return r;
    k_guid_put<<<nblk(n), 256, 0, e->stream>>>(e->d, n, dg, dv);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int ocrgpu_guid_tryput(ocrgpu_engine_t *e, u64 n, const u64 *guids, const u64 *vals, u64 *held_out, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!guids || !vals || !held_out) FAIL(OCRGPU_EINVAL, "null array");
    int r = stage_reserve(e, 3 * rup(8 * n)); if (r) return r;
    const u64 *dg, *dv; u64 *dh;
    if ((r = stage_in(e, guids, n, where, &dg)) || (r = stage_in(e, vals, n, where, &dv)) || (r = stage_out_alloc(e, held_out, n, where, &dh))) return r;
    k_guid_tryput<<<nblk(n), 256, 0, e->stream>>>(e->d, n, dg, dv, dh);
    if ((r = stage_out_copy(e, held_out, dh, n, where))) return r;
    if (where == OCRGPU_HOST) HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
extern "C" int ocrgpu_guid_get(ocrgpu_engine_t *e, u64 n, const u64 *guids, u64 *vals_out, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!guids || !vals_out) FAIL(OCRGPU_EINVAL, "null array");
    int r = stage_reserve(e, 2 * rup(8 * n)); if (r) return r;
    const u64 *dg; u64 *dv;
    if ((r = stage_in(e, guids, n, where, &dg)) || (r = stage_out_alloc(e, vals_out, n, where, &dv))) return r;
    k_guid_get<<<nblk(n), 256, 0, e->stream>>>(e->d, n, dg, dv);
    if ((r = stage_out_copy(e, vals_out, dv, n, where))) return r;
    if (where == OCRGPU_HOST) HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
extern "C" int ocrgpu_guid_remove(ocrgpu_engine_t *e, u64 n, const u64 *guids, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!guids) FAIL(OCRGPU_EINVAL, "null array");
    int r = stage_reserve(e, rup(8 * n)); if (r) return r;
    const u64 *dg; if ((r = stage_in(e, guids, n, where, &dg))) return r;
    k_guid_remove<<<nblk(n), 256, 0, e->stream>>>(e->d, n, dg);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- creation --------------------------------------------------------------------------------------------------
extern "C" int ocrgpu_event_create(ocrgpu_engine_t *e, u64 n, const u32 *kind, const s32 *arg, u64 *guids_out, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!kind || !arg || !guids_out) FAIL(OCRGPU_EINVAL, "null array");
    const u32 take = n < e->n_free_evt ? (u32)n : e->n_free_evt; // reclaimed records first
    if ((u64)e->n_evt + (n - take) > e->cfg.max_evt) FAIL(OCRGPU_ENOMEM, "max_evt=%u exhausted (%u reclaimed records free)", e->cfg.max_evt, e->n_free_evt);
    const bool given = (where & OCRGPU_GUIDS_GIVEN) != 0; where &= 1;
    int r = stage_reserve(e, 2 * rup(4 * n) + rup(8 * n)); if (r) return r;
    const u32 *dk; const s32 *da; u64 *dg = nullptr; const u64 *dgin = nullptr;
    if ((r = stage_in(e, kind, n, where, &dk)) || (r = stage_in(e, arg, n, where, &da))) return r;
    if (given) r = stage_in(e, (const u64 *)guids_out, n, where, &dgin); else r = stage_out_alloc(e, guids_out, n, where, &dg);
    if (r) return r;
    k_event_create<<<nblk(n), 256, 0, e->stream>>>(e->d, e->rc, e->n_free_evt, take, n, dk, da, e->n_evt, e->counter, dgin, dg);
    e->n_evt += (u32)(n - take); e->n_free_evt -= take; if (take) ++e->mut_epoch; if (!given) e->counter += n;
    if (!given && (r = stage_out_copy(e, guids_out, dg, n, where))) return r;
    if (where == OCRGPU_HOST) HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
extern "C" int ocrgpu_db_create(ocrgpu_engine_t *e, u64 n, const u64 *value, const u64 *tag, u64 *guids_out, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!value || !guids_out) FAIL(OCRGPU_EINVAL, "null array");
    const u32 take = n < e->n_free_db ? (u32)n : e->n_free_db;
    if ((u64)e->n_db + (n - take) > e->cfg.max_db) FAIL(OCRGPU_ENOMEM, "max_db=%u exhausted (%u destroyed datablocks free)", e->cfg.max_db, e->n_free_db);
    const bool given = (where & OCRGPU_GUIDS_GIVEN) != 0; where &= 1;
    int r = stage_reserve(e, 3 * rup(8 * n)); if (r) return r;
    const u64 *dv, *dt; u64 *dg = nullptr; const u64 *dgin = nullptr;
    if ((r = stage_in(e, value, n, where, &dv)) || (r = stage_in(e, tag, n, where, &dt))) return r;
    if (given) r = stage_in(e, (const u64 *)guids_out, n, where, &dgin); else r = stage_out_alloc(e, guids_out, n, where, &dg);
    if (r) return r;
    k_db_create<<<nblk(n), 256, 0, e->stream>>>(e->d, e->rc, e->n_free_db, take, n, dv, dt, e->n_db, e->counter, dgin, dg);
    if (e->tiles) k_tile_init<<<nblk(n * OCROPS_TILE_WORDS), 256, 0, e->stream>>>(e->d, e->rc, e->n_free_db, take, n, dv, e->n_db);
    e->n_db += (u32)(n - take); e->n_free_db -= take; if (take) ++e->mut_epoch; if (!given) e->counter += n;
    if (!given && (r = stage_out_copy(e, guids_out, dg, n, where))) return r;
    if (where == OCRGPU_HOST) HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
extern "C" int ocrgpu_edt_create(ocrgpu_engine_t *e, u64 n, const u32 *depc, const u32 *body, const u64 *param0,
                                 u64 *guids_out, u64 *outevt_out, u64 *outdb_out, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!depc || !body || !guids_out) FAIL(OCRGPU_EINVAL, "null array");
    const u32 take = n < e->n_free_edt ? (u32)n : e->n_free_edt;
    if ((u64)e->n_edt + (n - take) > e->cfg.max_edt) FAIL(OCRGPU_ENOMEM, "max_edt=%u exhausted (%u reclaimed lines free)", e->cfg.max_edt, e->n_free_edt);
    const bool given = (where & OCRGPU_GUIDS_GIVEN) != 0; where &= 1;
    int r = stage_reserve(e, 4 * rup(4 * n) + 4 * rup(8 * n) + 256); if (r) return r;
    const u32 *dd, *db; const u64 *dp; u64 *dg = nullptr, *doe = nullptr, *dod = nullptr; u32 *dsb, *dtot; const u64 *dgin = nullptr, *doin = nullptr;
    if ((r = stage_in(e, depc, n, where, &dd)) || (r = stage_in(e, body, n, where, &db)) || (r = stage_in(e, param0, n, where, &dp))) return r;
    if (given) {
        if ((r = stage_in(e, (const u64 *)guids_out, n, where, &dgin)) || (r = stage_in(e, (const u64 *)outevt_out, n, where, &doin))) return r;
    } else if ((r = stage_out_alloc(e, guids_out, n, where, &dg)) || (r = stage_out_alloc(e, outevt_out, n, where, &doe)) ||
               (r = stage_out_alloc(e, outdb_out, n, where, &dod))) return r;
    dsb = (u32 *)(e->stage_dev + e->stage_dev_off); e->stage_dev_off += rup(4 * n);
    dtot = (u32 *)(e->stage_dev + e->stage_dev_off); e->stage_dev_off += 256;
    u32 *dovf = (u32 *)(e->stage_dev + e->stage_dev_off); e->stage_dev_off += rup(4 * n);
    k_ovf_counts<<<nblk(n), 256, 0, e->stream>>>(n, dd, dovf);
    if ((r = scan_u32(e, dovf, dsb, n, (u32)e->n_slots, dtot))) return r;
    HIPCHK(hipMemcpyAsync(e->pinned_done + 33000, dtot, 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    u64 add = e->pinned_done[33000] - (u32)e->n_slots; // the scan was seeded with the slots already in use
    if (e->n_slots + add > e->d.max_slots) FAIL(OCRGPU_ENOMEM, "max_slots=%llu exhausted", (unsigned long long)e->d.max_slots);
    k_edt_create<<<nblk(n), 256, 0, e->stream>>>(e->d, e->rc, e->n_free_edt, take, n, dd, db, dp, dsb, e->n_edt, e->counter, dgin, doin, dg, doe, dod);
    e->n_edt += (u32)(n - take); e->n_free_edt -= take; if (take) ++e->mut_epoch; e->n_slots += add; if (!given) e->counter += 3 * n;
    if (!given && ((r = stage_out_copy(e, guids_out, dg, n, where)) || (r = stage_out_copy(e, outevt_out, doe, n, where)) ||
                   (r = stage_out_copy(e, outdb_out, dod, n, where)))) return r;
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
extern "C" int ocrgpu_edt_set_scripts(ocrgpu_engine_t *e, u64 n, const u64 *edt, const u32 *begin, const u32 *len,
                                      u64 n_ops, const ocrgpu_script_op_t *ops, int where) {
    ENTER(e);
    if (n == 0 || n_ops == 0) return 0;
    if (!edt || !begin || !len || !ops) FAIL(OCRGPU_EINVAL, "null array");
    if (e->n_script + n_ops > e->script_cap) { // the pool starts at max_edt + 1024 ops and doubles when a program needs more
        u64 cap = e->script_cap * 2 > e->n_script + n_ops ? e->script_cap * 2 : e->n_script + n_ops;
        Script *bigger = nullptr;
        if (cap > 0xFFFFFFFFull || hipMalloc((void **)&bigger, sizeof(Script) * cap) != hipSuccess) FAIL(OCRGPU_ENOMEM, "script pool cannot grow");
        HIPCHK(hipMemcpyAsync(bigger, e->d.scripts, sizeof(Script) * e->n_script, hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        for (void *&p : e->allocs) if (p == (void *)e->d.scripts) p = bigger;
        (void)hipFree(e->d.scripts);
        e->d.scripts = bigger; e->script_cap = cap;
    }
    int r = stage_reserve(e, rup(8 * n) + 2 * rup(4 * n) + rup(sizeof(ocrgpu_script_op_t) * n_ops)); if (r) return r;
    const u64 *de; const u32 *dbg, *dl; const ocrgpu_script_op_t *dops;
    if ((r = stage_in(e, edt, n, where, &de)) || (r = stage_in(e, begin, n, where, &dbg)) || (r = stage_in(e, len, n, where, &dl)) ||
        (r = stage_in(e, ops, n_ops, where, &dops))) return r;
    k_script_ops<<<nblk(n_ops), 256, 0, e->stream>>>(e->d, n_ops, dops, (u32)e->n_script);
    k_script_merge_latch<<<nblk(n), 256, 0, e->stream>>>(e->d, n, dbg, dl, (u32)e->n_script);
    k_set_scripts<<<nblk(n), 256, 0, e->stream>>>(e->d, n, de, dbg, dl, (u32)e->n_script);
    e->n_script += n_ops;
    HIPCHK(hipGetLastError());
    if (where == OCRGPU_HOST) HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

// ---- dependence API -----------------------------------------------------------------------------------------
// API batches are applied in chunks of at most a quarter of the record queue, with a drain between them: every element can
// queue one satisfy record (a DB or NULL_GUID source, a satisfied persistent source, every ocrEventSatisfySlot), and the queue
// only rewinds when it has been drained.  (With an exchange configured the caller drains: ocrgpu_step.)
static inline u64 api_chunk(const ocrgpu_engine *e) { u64 c = e->q_cap / 4; return c < 1024 ? 1024 : c; }
extern "C" int ocrgpu_add_dependence(ocrgpu_engine_t *e, u64 n, const u64 *src, const u64 *dst, const u32 *slot, const u32 *mode, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!src || !dst || !slot || !mode) FAIL(OCRGPU_EINVAL, "null array");
    if (e->edges_upper + n > e->d.max_edges) FAIL(OCRGPU_ENOMEM, "max_edges=%llu exhausted", (unsigned long long)e->d.max_edges);
    const u64 chunk = api_chunk(e);
    int fault = 0;
    for (u64 o = 0; o < n; o += chunk) {
        const u64 m = n - o < chunk ? n - o : chunk;
        stage_reset(e);
        int r = stage_reserve(e, 2 * rup(8 * m) + 2 * rup(4 * m)); if (r) return r;
        const u64 *ds, *dd; const u32 *dsl, *dm;
        if ((r = stage_in(e, src + o, m, where, &ds)) || (r = stage_in(e, dst + o, m, where, &dd)) || (r = stage_in(e, slot + o, m, where, &dsl)) ||
            (r = stage_in(e, mode + o, m, where, &dm))) return r;
        // registrations first, then the immediate satisfies this batch produced (see DESIGN.md "batch order")
        k_add_dep<<<(u32)((m + ADD_DEP_BLK - 1) / ADD_DEP_BLK), ADD_DEP_BLK, 0, e->stream>>>(e->d, m, ds, dd, dsl, dm, (e->cfg.flags & OCRGPU_F_NANNY) ? 1u : 0u);
        HIPCHK(hipGetLastError());
        e->edges_upper += m; e->csr_dirty = true;
        if (where == OCRGPU_HOST) HIPCHK(hipStreamSynchronize(e->stream)); // staging buffers are reused by the next chunk / call
        r = drain_events(e);
        if (r == OCRGPU_EFAULT) fault = r; else if (r) return r;
    }
    return fault;
}
extern "C" int ocrgpu_event_satisfy_slot(ocrgpu_engine_t *e, u64 n, const u64 *evt, const u64 *db, const u32 *slot, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!evt || !db || !slot) FAIL(OCRGPU_EINVAL, "null array");
    const u64 chunk = api_chunk(e);
    int fault = 0;
    for (u64 o = 0; o < n; o += chunk) {
        const u64 m = n - o < chunk ? n - o : chunk;
        stage_reset(e);
        int r = stage_reserve(e, 2 * rup(8 * m) + rup(4 * m)); if (r) return r;
        const u64 *de, *dd; const u32 *dsl;
        if ((r = stage_in(e, evt + o, m, where, &de)) || (r = stage_in(e, db + o, m, where, &dd)) || (r = stage_in(e, slot + o, m, where, &dsl))) return r;
        k_satisfy<<<nblk(m), 256, 0, e->stream>>>(e->d, m, de, dd, dsl);
        HIPCHK(hipGetLastError());
        if (where == OCRGPU_HOST) HIPCHK(hipStreamSynchronize(e->stream));
        r = drain_events(e);
        if (r == OCRGPU_EFAULT) fault = r; else if (r) return r;
    }
    return fault;
}
extern "C" int ocrgpu_event_destroy(ocrgpu_engine_t *e, u64 n, const u64 *evt, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!evt) FAIL(OCRGPU_EINVAL, "null array");
    int r = stage_reserve(e, rup(8 * n)); if (r) return r;
    const u64 *de; if ((r = stage_in(e, evt, n, where, &de))) return r;
    k_evt_destroy<<<nblk(n), 256, 0, e->stream>>>(e->d, n, de);
    HIPCHK(hipGetLastError());
    if (where == OCRGPU_HOST) HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

// ---- reclaiming capacity ------------------------------------------------------------------------------------------
extern "C" int ocrgpu_db_destroy(ocrgpu_engine_t *e, u64 n, const u64 *guids, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!guids) FAIL(OCRGPU_EINVAL, "null array");
    int r = stage_reserve(e, rup(8 * n)); if (r) return r;
    const u64 *dg; if ((r = stage_in(e, guids, n, where, &dg))) return r;
    HIPCHK(hipMemsetAsync(e->rc.cnt + RC_DB, 0, 4, e->stream));
    k_db_destroy<<<nblk(n), 256, 0, e->stream>>>(e->d, e->rc, e->n_free_db, e->n_db, n, dg);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(e->pinned_done + 33016, e->rc.cnt + RC_DB, 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(e->pinned_done + PIN_ERR, e->d.ctrl->err, 32, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream)); // the free stack's height is the host's to know
    e->n_free_db += e->pinned_done[33016]; ++e->mut_epoch;
    return check_fault(e);
}
extern "C" int ocrgpu_reclaim(ocrgpu_engine_t *e, ocrgpu_reclaim_t *out) {
    ENTER(e);
    Dev &d = e->d; const ReclDev &rc = e->rc;
    // quiescence: no satisfy record may be in flight (a record names its destination by index)
    HIPCHK(hipMemcpyAsync(e->pinned_done + 33008, &d.ctrl->q_begin, 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(e->pinned_done + 33009, &d.ctrl->q_tail, 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    int fault = 0;
    if (e->pinned_done[33009] > e->pinned_done[33008]) {
        if (e->defer_drain) FAIL(OCRGPU_EINVAL, "ocrgpu_reclaim: satisfy records are pending (call ocrgpu_step first)");
        int r = run_steps(e, 0u, ~0ull, false); if (r) return r;
        fault = check_fault(e);
    }
    HIPCHK(hipMemcpyAsync(e->pinned_done + 33010, &d.ctrl->r_end, 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(e->pinned_done + 33011, &d.ctrl->r_tail, 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    const u32 r_end = e->pinned_done[33010], r_tail = e->pinned_done[33011];
    const u32 pop_cursor = e->pop_cursor < r_end ? e->pop_cursor : r_end;
    HIPCHK(hipMemsetAsync(rc.cnt, 0, 4 * RC_N, e->stream));
    if (e->n_edt) k_reclaim_lazy_users<<<nblk(e->n_edt), 256, 0, e->stream>>>(d, rc, e->n_edt);
    const u64 n_cand = (u64)e->n_held + r_end;
    if (n_cand) k_reclaim_edts<<<nblk(n_cand, RECL_BLK), RECL_BLK, 0, e->stream>>>(d, rc, e->n_held, r_end, pop_cursor, e->n_free_edt, e->tmp_a);
    if (e->n_evt) k_reclaim_evts<<<nblk(e->n_evt), 256, 0, e->stream>>>(d, rc, e->n_evt, e->n_free_evt);
    if (e->n_edt) k_reclaim_pool_users<<<nblk(e->n_edt), 256, 0, e->stream>>>(d, rc, e->n_edt);
    // the registration log (the rows are rebuilt from it by the next run)
    const u64 nl = e->edges_upper;
    if (nl) {
        const size_t need = 2 * rup(4 * nl) + rup(4 * nl) + rup(8 * nl);
        if (need > e->recl_tmp_cap) {
            HIPCHK(hipStreamSynchronize(e->stream));
            if (e->recl_tmp) HIPCHK(hipFree(e->recl_tmp));
            e->recl_tmp = nullptr; e->recl_tmp_cap = 0;
            HIPCHK(hipMalloc((void **)&e->recl_tmp, need + need / 4));
            e->recl_tmp_cap = need + need / 4;
        }
        u32 *flags = (u32 *)e->recl_tmp, *pos = (u32 *)(e->recl_tmp + rup(4 * nl)), *tsrc = (u32 *)(e->recl_tmp + 2 * rup(4 * nl));
        u64 *trec = (u64 *)(e->recl_tmp + 3 * rup(4 * nl));
        HIPCHK(hipMemsetAsync(flags, 0, 4 * nl, e->stream)); // entries past Ctrl.log_tail (calls that registered nothing) count as dropped
        u32 g = nblk(nl); if (g > 4096) g = 4096;
        k_log_flags<<<g, 256, 0, e->stream>>>(d, flags);
        int r = scan_u32(e, flags, pos, nl, 0, rc.cnt + RC_LOG_KEPT); if (r) return r;
        k_log_scatter<<<g, 256, 0, e->stream>>>(d, flags, pos, tsrc, trec);
        k_log_back<<<g, 256, 0, e->stream>>>(d, tsrc, trec, rc.cnt + RC_LOG_KEPT);
    }
    // the ready log
    if (r_tail > r_end) { u32 g = nblk(r_tail - r_end); if (g > 4096) g = 4096; k_ready_gather<<<g, 256, 0, e->stream>>>(d, rc, r_end, r_tail, e->tmp_b); }
    k_ready_back<<<1024, 256, 0, e->stream>>>(d, rc, e->tmp_a, e->tmp_b);
    k_reclaim_finish<<<1, 1, 0, e->stream>>>(d, rc, nl ? 1u : 0u);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(e->pinned_done + 33950, rc.cnt, 4 * RC_N, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    const u32 *c = e->pinned_done + 33950;
    e->n_free_edt += c[RC_EDT]; e->n_free_evt += c[RC_EVT];
    e->n_held = c[RC_HELD]; std::swap(e->rc.held, e->rc.held_new);
    if (nl) e->edges_upper = c[RC_LOG_KEPT];
    e->pop_cursor = 0; e->csr_dirty = true; ++e->mut_epoch;
    if (c[RC_OVF_LIVE] == 0) e->n_slots = 0;     // no EDT left that keeps pre-slot words outside its line
    if (c[RC_SCRIPT_LIVE] == 0) e->n_script = 0; // nor one with a script
    if (c[RC_LAZY_FREED] && c[RC_LAZY_LIVE] == 0) HIPCHK(hipMemsetAsync(d.lazy_tab, 0, 16 * (size_t)e->lazy_cap, e->stream));
    memset(e->pinned_done, 0, sizeof(State)); ((State *)e->pinned_done)->done = 1; // ocrgpu_step_info: nothing fired since
    if (out) {
        out->edts_freed = c[RC_EDT]; out->events_freed = c[RC_EVT]; out->edts_held = c[RC_HELD]; out->edges_kept = e->edges_upper;
        out->free_edt = e->n_free_edt; out->free_evt = e->n_free_evt; out->free_db = e->n_free_db;
        out->fresh_edt = e->cfg.max_edt - e->n_edt; out->fresh_evt = e->cfg.max_evt - e->n_evt; out->fresh_db = e->cfg.max_db - e->n_db;
        out->slots_used = e->n_slots; out->ready_kept = c[RC_UNPOPPED] + c[RC_READY];
    }
    return fault;
}

// Host-run bodies (OCRBODY_HOST) under OCRGPU_F_LOCKABLE_DB: the EDTs named have run; what they hold is released.
__global__ void k_release_dbs(Dev d, u64 n, const u64 *edt) {
    u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 v = gtab_get(d, edt[i]);
    if (!v || OCRGPU_GUID_KIND(edt[i]) != OCRGPU_KIND_EDT) { bad_op(d); return; }
    const EdtLine *L = &d.edt[(u32)(v - 1)];
    const u32 depc = L->depc;
    release_all<true>(d, depc <= INLINE_SLOTS ? L->slot : d.slots + L->slot[0], depc, [&](u32 w) { if (w > d.max_edt) grant_remote(d, w); else q_push(d, Rec{w - 1u, 0u, 0u, RF_ACQUIRE}); });
}
extern "C" int ocrgpu_edt_release_dbs(ocrgpu_engine_t *e, u64 n, const u64 *edt, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!edt) FAIL(OCRGPU_EINVAL, "null array");
    if (!e->dbl) FAIL(OCRGPU_EINVAL, "engine was not created with OCRGPU_F_LOCKABLE_DB");
    int r = stage_reserve(e, rup(8 * n)); if (r) return r;
    const u64 *de; if ((r = stage_in(e, edt, n, where, &de))) return r;
    k_release_dbs<<<nblk(n), 256, 0, e->stream>>>(e->d, n, de);
    HIPCHK(hipGetLastError());
    if (where == OCRGPU_HOST) HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

// ---- worker loop ----------------------------------------------------------------------------------------------
static int fetch_stats(ocrgpu_engine *e, ocrgpu_stats_t *out) {
    Ctrl h;
    HIPCHK(hipMemcpyAsync(&h, e->d.ctrl, sizeof h, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    memset(out, 0, sizeof *out);
    std::vector<u64> bs(4 * MAX_GRID);
    HIPCHK(hipMemcpyAsync(bs.data(), e->d.blk_stats, 8 * 4 * MAX_GRID, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    out->levels = h.n_levels;
    for (u32 b = 0; b < MAX_GRID; ++b) { out->edt_sat += bs[4 * b]; out->evt_sat += bs[4 * b + 1]; out->fired += bs[4 * b + 2]; }
    out->launches = e->launches_total; out->step_kernel_ns = e->step_ns_total; out->async_runs = e->async_runs;
    for (int i = 0; i < 8; ++i) out->err[i] = h.err[i];
    return 0;
}
// The barrier-free resolve (k_async, see there), entered with the main program's satisfies applied (the queue of satisfy records
// empty) and the EDTs they made ready as seeds.  One synchronisation at its end.
static int run_async(ocrgpu_engine *e) {
    Dev &d = e->d;
    ADev a = e->ad; a.n_evt = e->n_evt;
    const bool timing = e->have_events;
#ifdef OCRGPU_ASTAMPS
    static double hacc[8]; static u64 hn; int hk = 0;
    auto hnow = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double ht = hnow();
#define HSTAMP() { double t2 = hnow(); hacc[hk++] += t2 - ht; ht = t2; }
#else
#define HSTAMP()
#endif
    if (e->irow_stale && e->n_edt) { k_irow_fill<<<nblk(e->n_edt), 256, 0, e->stream>>>(d, a.irow, e->n_edt); e->irow_stale = false; }
    k_async_reset<<<1024, 256, 0, e->stream>>>(a); HSTAMP();
    k_async_init<<<1, 64, 0, e->stream>>>(d, a); HSTAMP();
    k_async_seed<<<256, 256, 0, e->stream>>>(d, a); HSTAMP();
    if (timing) HIPCHK(hipEventRecord(e->ev0, e->stream));
    HSTAMP();
    k_async<<<e->async_waves, 64, 0, e->stream>>>(d, a); HSTAMP();
    if (timing) HIPCHK(hipEventRecord(e->ev1, e->stream));
    HSTAMP();
#ifdef OCRGPU_ASTAMPS
    if ((++hn & 7) == 0) fprintf(stderr, "[run_async host us per call] reset %.1f init %.1f seed %.1f ev0 %.1f k_async %.1f ev1 %.1f\n", hacc[0] / hn, hacc[1] / hn, hacc[2] / hn, hacc[3] / hn, hacc[4] / hn, hacc[5] / hn);
#endif
    k_async_group<false><<<2048, 256, 0, e->stream>>>(d, a);
    k_async_scan<<<1, 1024, 0, e->stream>>>(d, a);
    k_async_group<true><<<2048, 256, 0, e->stream>>>(d, a);
    if (e->launch_index > 0xFFFF0000u) e->launch_index = 8;
    const u32 K = ++e->launch_index;
    k_async_end<<<64, 256, 0, e->stream>>>(d, a, K);
    k_report<<<1, 64, 0, e->stream>>>(d, K, RP_STATE | RP_ERR, e->pinned_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    if (timing) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1)); e->step_ns_total += (u64)(ms * 1e6); }
    ++e->launches_total; ++e->async_runs;
    return 0;
}
extern "C" int ocrgpu_resolve(ocrgpu_engine_t *e, u64 max_levels, ocrgpu_stats_t *stats_out) {
    ENTER(e);
    REFUSE_POISONED(e);
    u64 max_launches = max_levels ? 4 * max_levels + 64 : ~0ull;
    int r;
    bool done = false;
    if (e->async_mode && !max_levels && e->d.n_loc <= 1 && !e->defer_drain) {
        // apply what the main program queued (EVENTS phases only: nothing fires), then look at what is ready and at what the DAG is made of
        // (the satisfy calls drain their own records unless told otherwise, so as a rule the first look finds the queue empty)
        r = seal(e); if (r) return r;
        const u32 *pn = e->pinned_done;
        for (int look = 0; look < 2; ++look) {
            k_report<<<1, 64, 0, e->stream>>>(e->d, 0u, RP_QUEUE | RP_REND | RP_ASYNC | RP_ERR, e->pinned_dev);
            HIPCHK(hipStreamSynchronize(e->stream));
            if (look || pn[33009] <= pn[33008]) break;
            r = run_steps(e, 0u, ~0ull, false); if (r) return r;
        }
        const u32 seeds = pn[33011] - pn[33010];
        const bool clean = pn[33040] == 0 && pn[33009] <= pn[33008] && pn[33043] <= pn[33042] && pn[33044] + 2u < e->d.max_levels;
        if (clean && seeds && pn[33041] <= e->async_max_row && (e->async_mode >= 2 || seeds <= AMAX_SEEDS)) { r = run_async(e); if (r) return r; done = true; }
    }
    if (!done) { r = run_steps(e, 0xFFFFFFFFu, max_launches, true); if (r) return r; }
    if (stats_out) { r = fetch_stats(e, stats_out); if (r) return r; e->last = *stats_out; }
    return check_fault(e); // stats_out is filled either way
}
extern "C" int ocrgpu_step(ocrgpu_engine_t *e, u32 fire_levels, u32 flags, u64 *has_ready_out, u64 *pending_records_out,
                           u32 *send_counts_out) {
    ENTER(e);
    REFUSE_POISONED(e);
    // one run: close pending satisfies, fire at most `fire_levels` levels (an empty one if forced), close again;
    // 3 launches cover the common single-level case (FIRE, the no-op that notices quiescence, one spare)
    int r = run_steps(e, fire_levels, ~0ull, true, flags & (OCRGPU_STEP_FORCE_LEVEL | OCRGPU_STEP_DROP_EMPTY_LEVEL), fire_levels == 1 ? 3u : 2u);
    if (r) return r;
    if (flags & OCRGPU_STEP_WRITE_HEADERS) {
        if (!e->d.send_cnt || e->d.n_loc > 1024 || e->d.send_cap < 2) FAIL(OCRGPU_EINVAL, "exchange not configured");
        k_exchange_headers<<<1, 1024, 0, e->stream>>>(e->d, e->launch_index);
        HIPCHK(hipGetLastError());
    }
    const State *S = (const State *)e->pinned_done; // the poll that saw `done` brought the final state and counters
    if (has_ready_out) *has_ready_out = S->r_total > S->r_end ? 1 : 0;
    if (pending_records_out) *pending_records_out = S->q_total > S->q_end ? S->q_total - S->q_end : 0;
    if (send_counts_out) {
        if (!e->d.send_cnt) FAIL(OCRGPU_EINVAL, "exchange not configured");
        for (u32 l = 0; l < e->d.n_loc; ++l) { u32 c = e->pinned_done[32 + 32 * l]; send_counts_out[l] = c > e->d.send_cap - 1 ? e->d.send_cap - 1 : c; }
    }
    e->hdr_fresh = (flags & OCRGPU_STEP_WRITE_HEADERS) != 0;
    return check_fault(e); // the outputs above are filled either way
}

// Shared by ocrgpu_pop_ready (buf = the staging arena, grown on demand) and ocrgpu_pop_ready_noalloc (buf = this device's
// copy of the static `scratch` array): counters in the first 256 bytes, then the five output arrays.
static int pop_ready_into(ocrgpu_engine *e, char *buf, u32 r_end, u64 max_edts, u64 max_deps, u64 *n_edts_out, u64 *n_deps_out, u64 *param0_out,
                          u32 *depc_out, u32 *dep_first_out, u64 *dep_guid_out, u64 *dep_value_out) {
    size_t off = 0;
    auto carve = [&](size_t bytes) { char *p = buf + off; off += rup(bytes); return p; };
    u32 *cnt = (u32 *)carve(256);
    u64 *dp0 = (u64 *)carve(8 * max_edts); u32 *ddc = (u32 *)carve(4 * max_edts), *ddf = (u32 *)carve(4 * max_edts);
    u64 *dgd = (u64 *)carve(8 * max_deps), *dvl = (u64 *)carve(8 * max_deps);
    HIPCHK(hipMemsetAsync(cnt, 0, 256, e->stream)); // the buffer keeps the last call's contents: the two counters start at zero
    u64 span = r_end - e->pop_cursor; u32 g = nblk(span); if (g > 4096) g = 4096;
    k_pop_ready<<<g, 256, 0, e->stream>>>(e->d, e->pop_cursor, r_end, (u32)max_edts, (u32)max_deps, cnt, dp0, ddc, ddf, dgd, dvl);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(e->pinned_done + 33012, cnt, 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    const u64 ne = e->pinned_done[33012], nd = e->pinned_done[33013];
    *n_edts_out = ne; *n_deps_out = nd;
    if (ne > max_edts || nd > max_deps) FAIL(OCRGPU_ENOMEM, "pop_ready: %llu EDTs / %llu deps are ready, room for %llu / %llu (nothing was popped)",
                                              (unsigned long long)ne, (unsigned long long)nd, (unsigned long long)max_edts, (unsigned long long)max_deps);
    if (ne) {
        HIPCHK(hipMemcpyAsync(param0_out, dp0, 8 * ne, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(depc_out, ddc, 4 * ne, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(dep_first_out, ddf, 4 * ne, hipMemcpyDeviceToHost, e->stream));
    }
    if (nd) {
        HIPCHK(hipMemcpyAsync(dep_guid_out, dgd, 8 * nd, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(dep_value_out, dvl, 8 * nd, hipMemcpyDeviceToHost, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    e->pop_cursor = r_end;
    return 0;
}
static int pop_ready_args(ocrgpu_engine *e, u64 max_edts, u64 max_deps, u64 *n_edts_out, u64 *n_deps_out, u64 *param0_out, u32 *depc_out,
                          u32 *dep_first_out, u64 *dep_guid_out, u64 *dep_value_out, u32 *r_end_out) {
    if (!n_edts_out || !n_deps_out) FAIL(OCRGPU_EINVAL, "null argument");
    if ((max_edts && (!param0_out || !depc_out || !dep_first_out)) || (max_deps && (!dep_guid_out || !dep_value_out))) FAIL(OCRGPU_EINVAL, "null array");
    if (max_edts > 0xFFFFFFFFull || max_deps > 0xFFFFFFFFull) FAIL(OCRGPU_EINVAL, "capacity above 2^32");
    k_report<<<1, 64, 0, e->stream>>>(e->d, 0u, RP_REND, e->pinned_dev);
    HIPCHK(hipStreamSynchronize(e->stream));
    *r_end_out = e->pinned_done[33010];
    *n_edts_out = 0; *n_deps_out = 0;
    return 0;
}
extern "C" int ocrgpu_pop_ready(ocrgpu_engine_t *e, u64 max_edts, u64 max_deps, u64 *n_edts_out, u64 *n_deps_out, u64 *param0_out,
                                u32 *depc_out, u32 *dep_first_out, u64 *dep_guid_out, u64 *dep_value_out) {
    ENTER(e);
    u32 r_end;
    int r = pop_ready_args(e, max_edts, max_deps, n_edts_out, n_deps_out, param0_out, depc_out, dep_first_out, dep_guid_out, dep_value_out, &r_end);
    if (r) return r;
    if (r_end <= e->pop_cursor) return 0;
    r = stage_reserve(e, 256 + rup(8 * max_edts) + 2 * rup(4 * max_edts) + 2 * rup(8 * max_deps)); if (r) return r;
    return pop_ready_into(e, e->stage_dev, r_end, max_edts, max_deps, n_edts_out, n_deps_out, param0_out, depc_out, dep_first_out, dep_guid_out, dep_value_out);
}

// ---- the entry point that allocates nothing ----------------------------------------------------------------------
// ocrgpu_pop_ready stages its counters and its five output arrays in device memory it may have to (re)allocate.  Callers
// that cannot allocate (a worker loop under a harness that forbids hipMalloc, code between graph captures, tight serving
// loops) get the same call on a file-scope device array of fixed size.  Every GPU holds its own copy of the array at its own
// address: ocrgpu_init asks for it once per device, with that device current, and host code only ever uses that pointer.
// One array per GPU: calls that overlap on one GPU (two engines, two threads) race on it; it keeps the last call's contents.
#define NOALLOC_MAX_EDTS 32768u
#define NOALLOC_MAX_DEPS 131072u
#define NOALLOC_BYTES (256u + 16u * NOALLOC_MAX_EDTS + 16u * NOALLOC_MAX_DEPS + 5u * 256u)   /* 2.5 MiB + the carve padding */
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];
#define NOALLOC_MAX_DEVICES 64
static unsigned char *g_scratch_ptr[NOALLOC_MAX_DEVICES]; // device address of `scratch` on each GPU (hipGetSymbolAddress), 0 = not asked yet
static std::atomic<int> g_scratch_busy[NOALLOC_MAX_DEVICES];
static int scratch_setup(int device) { // called by ocrgpu_init with `device` current
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return 0; // such a device has no no-alloc entry point (it says so when called)
    if (g_scratch_ptr[device]) return 0;
    void *ptr = nullptr;
    HIPCHK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) FAIL(OCRGPU_EHIP, "hipGetSymbolAddress(scratch) returned a null address on device %d", device);
    g_scratch_ptr[device] = (unsigned char *)ptr;
    return 0;
}
extern "C" int ocrgpu_pop_ready_noalloc(ocrgpu_engine_t *e, u64 max_edts, u64 max_deps, u64 *n_edts_out, u64 *n_deps_out, u64 *param0_out,
                                        u32 *depc_out, u32 *dep_first_out, u64 *dep_guid_out, u64 *dep_value_out) {
    ENTER(e);
    if (max_edts > NOALLOC_MAX_EDTS || max_deps > NOALLOC_MAX_DEPS)
        FAIL(OCRGPU_EINVAL, "ocrgpu_pop_ready_noalloc: room for %llu EDTs / %llu deps asked, the static scratch holds at most %u / %u (use ocrgpu_pop_ready)",
             (unsigned long long)max_edts, (unsigned long long)max_deps, NOALLOC_MAX_EDTS, NOALLOC_MAX_DEPS);
    unsigned char *ptr = (e->device >= 0 && e->device < NOALLOC_MAX_DEVICES) ? g_scratch_ptr[e->device] : nullptr;
    if (!ptr) FAIL(OCRGPU_ENOTSUP, "ocrgpu_pop_ready_noalloc: no scratch address for device %d", e->device);
    // every engine on this GPU shares the one array: a call that overlaps another one (two threads, two engines) is refused, not raced
    struct Busy { std::atomic<int> &f; bool mine; Busy(std::atomic<int> &x) : f(x), mine(x.exchange(1) == 0) {} ~Busy() { if (mine) f.store(0); } } busy(g_scratch_busy[e->device]);
    if (!busy.mine) FAIL(OCRGPU_EINVAL, "ocrgpu_pop_ready_noalloc: another call is using device %d's scratch array (calls on one GPU must not overlap)", e->device);
    u32 r_end;
    int r = pop_ready_args(e, max_edts, max_deps, n_edts_out, n_deps_out, param0_out, depc_out, dep_first_out, dep_guid_out, dep_value_out, &r_end);
    if (r) return r;
    if (r_end <= e->pop_cursor) return 0;
    return pop_ready_into(e, (char *)ptr, r_end, max_edts, max_deps, n_edts_out, n_deps_out, param0_out, depc_out, dep_first_out, dep_guid_out, dep_value_out);
}

extern "C" int ocrgpu_step_info(ocrgpu_engine_t *e, u64 *out4) {
    if (!e || !out4) FAIL(OCRGPU_EINVAL, "null argument");
    const State *S = (const State *)e->pinned_done; // what the last run's final poll brought back
    out4[0] = S->r_end - S->r_begin; out4[1] = S->level; out4[2] = S->n_levels; out4[3] = S->q_total > S->q_end ? S->q_total - S->q_end : 0;
    return 0;
}

// ---- cross-partition exchange (replaces the comm-platform's sendMessage/pollMessage for DEP_SATISFY,
//      ocr/src/inc/ocr-comm-platform.h:55-199; the transport itself -- RCCL all-to-all -- is the caller's) ----
extern "C" int ocrgpu_exchange_config(ocrgpu_engine_t *e, u32 n_locations, void *send_dev_buf, u64 records_per_dest, u64 max_remote_dsts) {
    ENTER(e);
    if (n_locations == 0 || n_locations > 1024 || e->cfg.location >= n_locations) FAIL(OCRGPU_EINVAL, "bad n_locations");
    if (n_locations > 1 && (records_per_dest == 0 || records_per_dest > 0xFFFFFFFFull)) FAIL(OCRGPU_EINVAL, "records_per_dest must be in [1, 2^32)");
    Dev &d = e->d;
    if (!send_dev_buf && records_per_dest) { // the engine owns the send regions (a plain C host has no device allocator)
        int r = dalloc(e, &send_dev_buf, sizeof(SendRec) * (size_t)n_locations * (size_t)records_per_dest); if (r) return r;
    }
    d.n_loc = n_locations; d.send = (SendRec *)send_dev_buf; d.send_cap = (u32)records_per_dest;
    if (!e->send_cnt_dev) { int r = dalloc(e, (void **)&e->send_cnt_dev, 4 * 32 * 1024); if (r) return r; }
    d.send_cnt = e->send_cnt_dev;
    if (max_remote_dsts && !d.rdst) { int r = dalloc(e, (void **)&d.rdst, sizeof(RemoteDst) * (size_t)max_remote_dsts); if (r) return r; d.rdst_cap = (u32)max_remote_dsts; }
    e->defer_drain = n_locations > 1; // records may target other partitions: the distributed driver steps explicitly
    if (e->dbl && n_locations > 1 && !e->rwait) { // Lockable datablocks arbitrated by their owners: remote requesters in line, releases of a round
        const u64 wcap = 4ull * e->cfg.max_edt + 65536ull, rcap = (u64)n_locations * records_per_dest;
        int r;
        if ((r = dalloc(e, (void **)&e->rwait, sizeof(RWait) * (size_t)wcap)) || (r = dalloc(e, (void **)&e->rrel, 8 * (size_t)rcap))) return r;
        struct { RWait *rwait; u64 *rrel; u32 rwait_cap, rwait_used, rrel_cap, rrel_n; } h = {e->rwait, e->rrel, (u32)(wcap > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : wcap), 0u, (u32)(rcap > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : rcap), 0u};
        HIPCHK(hipMemcpyAsync(&d.ctrl->rwait, &h, sizeof h, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    return 0;
}
// Lockable datablocks with partitions: the releases the apply kernel noted are done in a launch of their own right behind it
static void remote_releases(ocrgpu_engine *e) {
    if (!e->dbl || e->d.n_loc <= 1 || !e->rrel) return;
    k_remote_releases<<<64, 256, 0, e->stream>>>(e->d);
    k_remote_releases_done<<<1, 1, 0, e->stream>>>(e->d);
}
extern "C" int ocrgpu_exchange_counts(ocrgpu_engine_t *e, u32 *counts_out) {
    ENTER(e);
    if (!counts_out || !e->d.send_cnt) FAIL(OCRGPU_EINVAL, "exchange not configured");
    int r = stage_reserve(e, 4096 + 256); if (r) return r;
    u32 *dc = (u32 *)e->stage_dev;
    k_exchange_counts<<<1, 1024, 0, e->stream>>>(e->d, dc);
    HIPCHK(hipMemcpyAsync(counts_out, dc, 4 * e->d.n_loc, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
extern "C" int ocrgpu_exchange_reset(ocrgpu_engine_t *e) {
    ENTER(e);
    if (!e->d.send_cnt) FAIL(OCRGPU_EINVAL, "exchange not configured");
    k_exchange_reset<<<1, 1024, 0, e->stream>>>(e->d);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int ocrgpu_exchange_apply_strided(ocrgpu_engine_t *e, const void *recs_dev, u32 n_src, u64 stride_records, const u32 *counts) {
    ENTER(e);
    if (n_src == 0) return 0;
    if (!recs_dev || !counts || n_src > 1024) FAIL(OCRGPU_EINVAL, "bad argument");
    std::vector<u32> cp(2 * (size_t)n_src);
    u64 total = 0; u32 mx = 0;
    for (u32 s = 0; s < n_src; ++s) {
        if (counts[s] > stride_records) FAIL(OCRGPU_EINVAL, "count %u of source %u exceeds the region (%llu records)", counts[s], s, (unsigned long long)stride_records);
        cp[s] = counts[s]; cp[n_src + s] = (u32)total; total += counts[s]; if (counts[s] > mx) mx = counts[s];
    }
    if (total == 0) return 0;
    int r = stage_reserve(e, rup(8 * (size_t)n_src)); if (r) return r;
    // the counts go through pinned memory that outlives this call (two alternating halves, each guarded by an event), so the
    // copy can stay asynchronous
    if (n_src > 200) { // beyond the pinned scratch: stage from the vector and wait
        const u32 *dc; if ((r = stage_in(e, cp.data(), 2 * (u64)n_src, OCRGPU_HOST, &dc))) return r;
        k_exchange_apply_strided<<<nblk((u64)n_src * mx), 256, 0, e->stream>>>(e->d, n_src, stride_records, mx, dc, (const SendRec *)recs_dev, e->n_db);
        remote_releases(e);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(e->stream));
        return 0;
    }
    const u32 half = (e->apply_flip ^= 1u);
    u32 *pin = e->pinned_done + 33100 + 400 * half;
    if (!e->apply_ev[half]) HIPCHK(hipEventCreateWithFlags(&e->apply_ev[half], hipEventDisableTiming));
    else HIPCHK(hipEventSynchronize(e->apply_ev[half])); // the copy out of this half two calls ago (long done in practice)
    memcpy(pin, cp.data(), 8 * (size_t)n_src);
    const u32 *dc; if ((r = stage_in(e, pin, 2 * (u64)n_src, OCRGPU_HOST, &dc))) return r;
    HIPCHK(hipEventRecord(e->apply_ev[half], e->stream));
    k_exchange_apply_strided<<<nblk((u64)n_src * mx), 256, 0, e->stream>>>(e->d, n_src, stride_records, mx, dc, (const SendRec *)recs_dev, e->n_db);
    remote_releases(e);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int ocrgpu_exchange_apply(ocrgpu_engine_t *e, const void *recs, u64 n, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!recs) FAIL(OCRGPU_EINVAL, "null records");
    int r = stage_reserve(e, rup(32 * n)); if (r) return r;
    const SendRec *dr; if ((r = stage_in(e, (const SendRec *)recs, n, where, &dr))) return r;
    k_exchange_apply<<<nblk(n), 256, 0, e->stream>>>(e->d, n, dr, e->n_db);
    remote_releases(e);
    HIPCHK(hipGetLastError());
    if (where == OCRGPU_HOST) HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}


// ------------------------------------------------------------------------------------------------
// comm-platform over RCCL (takes the place of ocr/src/comm-platform/mpi/mpi-comm-platform.c: MPI_Init :54, the
// MPI_Isend of one marshalled ocrPolicyMsg_t per remote PD_MSG_DEP_SATISFY :1055, MPI_Iprobe/Irecv :287,620-642, behind
// ocrCommPlatformFcts_t ocr/src/inc/ocr-comm-platform.h:55-199).  One communicator rank per engine = per GPU = per
// policy-domain location.  What crosses xGMI per level is ONE equal-split all-to-all of the engine's send regions (32-byte
// records, header in the last slot of each region), as grouped ncclSend / ncclRecv on the engine's stream; the counts the
// receiver needs ride in the headers, so nothing comes back to the host between the step that filled the regions and the
// kernel that applies what arrived.  librccl is opened with dlopen when the first communicator is made: a single-GPU user
// of libocrgpu needs no RCCL at all, and a process that already holds torch's copy of the library shares it.
// ------------------------------------------------------------------------------------------------
struct RcclApi {
    void *lib;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    const char *(*GetErrorString)(ncclResult_t);
};
static RcclApi g_rccl;
static int rccl_load() {
    if (g_rccl.lib) return 0;
    const char *names[] = {getenv("OCRGPU_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) { if (n && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break; }
    if (!h) FAIL(OCRGPU_ENOTSUP, "librccl not found (dlopen: %s)", dlerror());
    RcclApi a; a.lib = h;
#define SYM(field, name) do { *(void **)&a.field = dlsym(h, name); if (!a.field) FAIL(OCRGPU_ENOTSUP, "librccl lacks %s", name); } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl = a;
    return 0;
}
#define RCCLCHK(call) do { ncclResult_t _r = (call); if (_r != ncclSuccess) { snprintf(g_err, sizeof g_err, "%s:%d %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(_r)); return OCRGPU_EHIP; } } while (0)

extern "C" int ocrgpu_comm_get_unique_id(void *id128_out) {
    if (!id128_out) FAIL(OCRGPU_EINVAL, "null argument");
    int r = rccl_load(); if (r) return r;
    ncclUniqueId id;
    RCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(id128_out, &id, sizeof id);
    return 0;
}
static int exchange_buffers(ocrgpu_engine *e) { // receive regions + the header sums, sized by the configured send regions
    Dev &d = e->d;
    if (!d.send_cnt || !d.send || d.send_cap < 2) FAIL(OCRGPU_EINVAL, "exchange not configured (ocrgpu_exchange_config first)");
    const u64 need = (u64)d.n_loc * d.send_cap;
    if (!e->recv_buf || e->recv_cap_records < need) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->recv_buf) (void)hipFree(e->recv_buf);
        HIPCHK(hipMalloc((void **)&e->recv_buf, sizeof(SendRec) * need));
        e->recv_cap_records = need;
    }
    return 0;
}
extern "C" int ocrgpu_comm_init(ocrgpu_engine_t *e, const void *id128, int rank, int nranks) {
    ENTER(e);
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) FAIL(OCRGPU_EINVAL, "bad rank %d of %d", rank, nranks);
    if (e->comm) FAIL(OCRGPU_EINVAL, "this engine already has a communicator");
    if ((u32)nranks != e->d.n_loc || (u32)rank != e->cfg.location) FAIL(OCRGPU_EINVAL, "rank %d of %d does not match location %u of %u (ocrgpu_exchange_config)", rank, nranks, e->cfg.location, e->d.n_loc);
    int r = rccl_load(); if (r) return r;
    if ((r = exchange_buffers(e))) return r;
    ncclUniqueId id; memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    RCCLCHK(g_rccl.CommInitRank(&c, nranks, id, rank));
    e->comm = (void *)c; e->comm_rank = rank; e->comm_nranks = nranks;
    return 0;
}
extern "C" int ocrgpu_comm_fini(ocrgpu_engine_t *e) {
    if (!e) FAIL(OCRGPU_EINVAL, "null engine");
    if (!e->comm) return 0;
    (void)hipSetDevice(e->device); (void)hipStreamSynchronize(e->stream);
    ncclResult_t r = g_rccl.CommDestroy((ncclComm_t)e->comm);
    e->comm = nullptr;
    if (r != ncclSuccess) FAIL(OCRGPU_EHIP, "ncclCommDestroy -> %s", g_rccl.GetErrorString(r));
    return 0;
}
// the transport itself: region p of send_dev goes to rank p, region p of recv_dev comes from rank p (bytes_per_peer each)
extern "C" int ocrgpu_comm_alltoall(ocrgpu_engine_t *e, const void *send_dev, void *recv_dev, u64 bytes_per_peer) {
    ENTER(e);
    if (!e->comm) FAIL(OCRGPU_EINVAL, "no communicator (ocrgpu_comm_init)");
    if (!send_dev || !recv_dev) FAIL(OCRGPU_EINVAL, "null buffer");
    if (bytes_per_peer == 0) return 0;
    ncclComm_t c = (ncclComm_t)e->comm;
    RCCLCHK(g_rccl.GroupStart());
    for (int p = 0; p < e->comm_nranks; ++p) {
        ncclResult_t a = g_rccl.Send((const char *)send_dev + (size_t)p * bytes_per_peer, bytes_per_peer, ncclUint8, p, c, e->stream);
        ncclResult_t b = a == ncclSuccess ? g_rccl.Recv((char *)recv_dev + (size_t)p * bytes_per_peer, bytes_per_peer, ncclUint8, p, c, e->stream) : a;
        if (b != ncclSuccess) { (void)g_rccl.GroupEnd(); FAIL(OCRGPU_EHIP, "ncclSend/ncclRecv with rank %d -> %s", p, g_rccl.GetErrorString(b)); }
    }
    RCCLCHK(g_rccl.GroupEnd());
    return 0;
}
// One exchange round, in three stages so that the same protocol runs over RCCL (one engine per process) and over plain
// device-to-device copies (several partitions in one process: the tests).  xchg_post: headers into the send regions.
// xchg_land: apply what arrived, clear the send counters, bring back the header sums (the round's ONE synchronisation).
static int xchg_words(ocrgpu_engine *e) { // [0,1024) push workgroups done per destination, [1024] sources late, [1025] apply workgroups done, [1026..1031] RA_*
    if (e->peer_cnt) return 0;
    return dalloc(e, (void **)&e->peer_cnt, 4 * 1040);
}
static int xchg_post(ocrgpu_engine *e, bool hdr_fresh) {
    int r = e->peer_on ? 0 : exchange_buffers(e); if (r) return r;
    if (e->d.n_loc > 1024) FAIL(OCRGPU_EINVAL, "more than 1024 locations");
    if (!hdr_fresh) k_exchange_headers<<<1, 1024, 0, e->stream>>>(e->d, e->launch_index); // (a step with OCRGPU_STEP_WRITE_HEADERS has just done it)
    HIPCHK(hipGetLastError());
    return 0;
}
static int xchg_land(ocrgpu_engine *e, const SendRec *recs, bool windows, u64 *out4, bool ahead = false, u32 no_cascade = 0, u32 next_K0 = 0) {
    Dev &d = e->d;
    // workgroups per source region: 32 for two partitions, 32 / n_loc (at least 4) beyond -- every workgroup ends in a device-scope
    // release and a ticket on one word, and while it waits for its source's arrival it holds its wave slots (partitions that share a
    // GPU: 8 x 32 x 8 workgroups of waiting applies filled the chip).  Measured with a stream per partition, 4 partitions 118 -> 108 us
    // per level, 8 partitions 207 -> 161.  OCRGPU_APPLY_BLOCKS overrides.
    u32 per = (d.send_cap + 254u) / 256u, lim = d.n_loc <= 2 ? 32u : (32u / d.n_loc < 4u ? 4u : 32u / d.n_loc);
    if (e->tiles) lim *= 2u; // (records of 23 slots, one lane per slot: T = 256 on 2 partitions 178 -> 157 ms, on 8: 466 -> 434; four times as many is worse again)
    { static const char *ab = getenv("OCRGPU_APPLY_BLOCKS"); if (ab && atoi(ab) > 0) lim = (u32)atoi(ab) > 512u ? 512u : (u32)atoi(ab); }
    if (per > lim) per = lim;
    int r = xchg_words(e); if (r) return r;
    // the header sums, the state the last step ended in and the error counters ride on the same synchronisation (step_ahead does not poll)
    k_exchange_apply_hdr<<<d.n_loc * per, 256, 0, e->stream>>>(d, d.n_loc, d.send_cap, recs, e->n_db, e->launch_index, e->pinned_dev, e->peer_cnt,
                                                             windows ? e->peer_win : nullptr, e->peer_round, e->peer_limit, ahead ? 1u : 0u, no_cascade,
                                                             (e->dbl && d.n_loc > 1) ? 0u : next_K0);
    remote_releases(e);
    // (Lockable engines: the releases a round brought in run in the launch above, behind the apply, and queue records -- the next round's
    // run begins behind them, not in the apply's last workgroup)
    if (next_K0 && e->dbl && d.n_loc > 1) k_begin_auto<<<1, 1, 0, e->stream>>>(d, next_K0, e->peer_cnt);
    HIPCHK(hipGetLastError());
    if (ahead) return 0; // rounds queued ahead: the caller synchronises once per batch
    HIPCHK(hipStreamSynchronize(e->stream));
    memcpy(out4, e->pinned_done + 33040, 32);
    out4[3] &= HDR_BEHIND - 1; // (a partition that is behind is not over either: it has counted itself in the low bits too)
    return 0;
}

// ---- peer windows, host side (the kernels and the protocol: k_peer_push above) ----
struct PeerHandle { hipIpcMemHandle_t ipc; u64 bytes; u32 n_loc, send_cap, location, pid; u64 local_ptr; s32 device; u32 pad[5]; };
static_assert(sizeof(PeerHandle) <= OCRGPU_PEER_HANDLE_BYTES, "handle size");
extern "C" int ocrgpu_peer_export(ocrgpu_engine_t *e, void *handle_out) {
    ENTER(e);
    if (!handle_out) FAIL(OCRGPU_EINVAL, "null handle");
    Dev &d = e->d;
    if (!d.send_cnt || !d.send || d.send_cap < 2) FAIL(OCRGPU_EINVAL, "exchange not configured (ocrgpu_exchange_config first)");
    if (e->peer_on) FAIL(OCRGPU_EINVAL, "this engine's window is already connected");
    const size_t bytes = peer_regions_off(d.n_loc) + sizeof(SendRec) * 2 * (size_t)d.n_loc * d.send_cap;
    if (!e->peer_win) {
        HIPCHK(hipExtMallocWithFlags((void **)&e->peer_win, bytes, hipDeviceMallocFinegrained));
        e->peer_win_bytes = bytes;
        HIPCHK(hipMemset(e->peer_win, 0, bytes)); // arrival words start at 0 = nothing arrived; done before anybody can hold the handle
        HIPCHK(hipDeviceSynchronize());
    } else if (e->peer_win_bytes != bytes) FAIL(OCRGPU_EINVAL, "the exchange was reconfigured after ocrgpu_peer_export");
    PeerHandle h; memset(&h, 0, sizeof h);
    h.bytes = bytes; h.n_loc = d.n_loc; h.send_cap = d.send_cap; h.location = e->cfg.location; h.pid = (u32)getpid(); h.local_ptr = (u64)(uintptr_t)e->peer_win; h.device = e->device;
    hipError_t er = hipIpcGetMemHandle(&h.ipc, e->peer_win); // (a group inside one process never opens it: not fatal here)
    if (er != hipSuccess) { (void)hipGetLastError(); memset(&h.ipc, 0, sizeof h.ipc); h.pad[0] = 1u; }
    memset(handle_out, 0, OCRGPU_PEER_HANDLE_BYTES); memcpy(handle_out, &h, sizeof h);
    return 0;
}
extern "C" int ocrgpu_peer_connect(ocrgpu_engine_t *e, const void *handles) {
    ENTER(e);
    if (!handles) FAIL(OCRGPU_EINVAL, "null handles");
    if (!e->peer_win) FAIL(OCRGPU_EINVAL, "ocrgpu_peer_export first");
    if (e->peer_on) FAIL(OCRGPU_EINVAL, "already connected");
    Dev &d = e->d;
    std::vector<char *> base(d.n_loc, nullptr);
    std::vector<void *> opened;
    auto undo = [&]() { for (void *q : opened) (void)hipIpcCloseMemHandle(q); };
    for (u32 p = 0; p < d.n_loc; ++p) {
        PeerHandle h; memcpy(&h, (const char *)handles + (size_t)p * OCRGPU_PEER_HANDLE_BYTES, sizeof h);
        if (h.n_loc != d.n_loc || h.send_cap != d.send_cap || h.location != p || h.bytes != e->peer_win_bytes) {
            undo(); FAIL(OCRGPU_EINVAL, "handle %u is of location %u of %u with regions of %u records: not this partition's shape", p, h.location, h.n_loc, h.send_cap);
        }
        if (p == e->cfg.location) { base[p] = e->peer_win; continue; }
        if (h.pid == (u32)getpid()) { // a partition of this same process: its pointer is good here
            if (h.device != e->device) { hipError_t er = hipDeviceEnablePeerAccess(h.device, 0); if (er != hipSuccess && er != hipErrorPeerAccessAlreadyEnabled) { undo(); FAIL(OCRGPU_EHIP, "no peer access from device %d to %d: %s", e->device, h.device, hipGetErrorString(er)); } (void)hipGetLastError(); }
            base[p] = (char *)(uintptr_t)h.local_ptr; continue;
        }
        if (h.pad[0]) { undo(); FAIL(OCRGPU_ENOTSUP, "location %u could not export its window to other processes (hipIpcGetMemHandle failed there)", p); }
        void *q = nullptr;
        hipError_t er = hipIpcOpenMemHandle(&q, h.ipc, hipIpcMemLazyEnablePeerAccess);
        if (er != hipSuccess) { (void)hipGetLastError(); undo(); FAIL(OCRGPU_EHIP, "hipIpcOpenMemHandle of location %u's window: %s", p, hipGetErrorString(er)); }
        opened.push_back(q); base[p] = (char *)q;
    }
    if (!e->peer_base_dev) { int r = dalloc(e, (void **)&e->peer_base_dev, 8 * 1024); if (r) { undo(); return r; } }
    { int r = xchg_words(e); if (r) { undo(); return r; } }
    HIPCHK(hipMemsetAsync(e->peer_cnt, 0, 4 * 1040, e->stream)); // (a window that gave up waiting earlier starts afresh: the late count is sticky)
    e->pinned_done[PIN_PEER_LATE] = 0;
    HIPCHK(hipMemcpyAsync(e->peer_base_dev, base.data(), 8 * (size_t)d.n_loc, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    const char *ms = getenv("OCRGPU_PEER_TIMEOUT_MS");
    e->peer_limit = 100000ull * (u64)(ms && atoi(ms) > 0 ? atoi(ms) : 2000); // 100 MHz ticks
    const char *pf = getenv("OCRGPU_PEER_FENCE");
    e->peer_fence_wave = pf && !strcmp(pf, "wave") ? 1u : 0u;
    e->peer_opened.swap(opened); e->peer_round = 0; e->peer_on = true; e->poisoned = false;
    return 0;
}
extern "C" int ocrgpu_peer_fini(ocrgpu_engine_t *e) {
    if (!e) FAIL(OCRGPU_EINVAL, "null engine");
    if (!e->peer_win && !e->peer_on) return 0;
    (void)hipSetDevice(e->device); (void)hipStreamSynchronize(e->stream);
    for (void *q : e->peer_opened) (void)hipIpcCloseMemHandle(q);
    e->peer_opened.clear();
    if (e->peer_win) { (void)hipFree(e->peer_win); e->peer_win = nullptr; }
    e->peer_on = false;
    return 0;
}
// one round over the windows for the partitions of this process (1 with one process per GPU): every push is queued before any
// wait, so partitions that share a hardware queue cannot wait on a push that sits behind them
static int peer_exchange(ocrgpu_engine **g, u32 n, u64 *h, bool ahead = false, u32 no_cascade = 0, bool next_K0 = false) {
    for (u32 i = 0; i < n; ++i) {
        ocrgpu_engine *e = g[i]; Dev &d = e->d;
        HIPCHK(hipSetDevice(e->device));
        if (d.n_loc > 1024) FAIL(OCRGPU_EINVAL, "more than 1024 locations");
        const PeerDev pd{e->peer_base_dev, e->peer_win, e->cfg.location, e->peer_cnt};
        k_peer_push<<<d.n_loc * peer_push_blocks(d.n_loc), 256, 0, e->stream>>>(d, pd, e->peer_round, e->launch_index, e->end_pending ? 1u : 0u, e->peer_fence_wave); // (the headers are made here)
        e->end_pending = false;
        HIPCHK(hipGetLastError());
    }
    for (u32 i = 0; i < n; ++i) {
        ocrgpu_engine *e = g[i]; Dev &d = e->d;
        HIPCHK(hipSetDevice(e->device));
        const SendRec *recs = (const SendRec *)(e->peer_win + peer_regions_off(d.n_loc)) + (size_t)(e->peer_round & 1) * d.n_loc * d.send_cap;
        int r = xchg_land(e, recs, true, h, ahead, no_cascade, next_K0 ? e->launch_index + 1 : 0u); if (r) return r;
        e->peer_round++;
        if (!ahead && e->pinned_done[PIN_PEER_LATE]) { e->poisoned = true;
                                               FAIL(OCRGPU_EHIP, "peer windows: %u of %u partitions did not arrive at location %u within the limit (OCRGPU_PEER_TIMEOUT_MS); "
                                               "the round may be partly applied and the run cannot continue", e->pinned_done[PIN_PEER_LATE], d.n_loc, e->cfg.location); }
    }
    return 0;
}
extern "C" int ocrgpu_exchange(ocrgpu_engine_t *e, u64 *records_in_out, u64 *activity_out, u64 *sent_out, u64 *queued_out) {
    const bool fresh = e && e->hdr_fresh;
    ENTER(e);
    REFUSE_POISONED(e);
    if (!e->comm && !e->peer_on) FAIL(OCRGPU_EINVAL, "no transport (ocrgpu_comm_init or ocrgpu_peer_connect)");
    u64 h[4]; int r;
    if (e->peer_on) { if ((r = peer_exchange(&e, 1, h))) return r; }
    else {
        if ((r = xchg_post(e, fresh))) return r;
        if ((r = ocrgpu_comm_alltoall(e, e->d.send, e->recv_buf, sizeof(SendRec) * (u64)e->d.send_cap))) return r;
        if ((r = xchg_land(e, e->recv_buf, false, h))) return r;
    }
    if (records_in_out) *records_in_out = h[0];
    if (activity_out) *activity_out = h[1];
    if (sent_out) *sent_out = h[2];
    if (queued_out) *queued_out = h[3];
    return 0;
}
extern "C" int ocrgpu_exchange_local(ocrgpu_engine_t **engines, u32 n, u64 *activity_out, u64 *sent_out, u64 *queued_out) {
    if (!engines || n == 0) FAIL(OCRGPU_EINVAL, "no engines");
    for (u32 i = 0; i < n; ++i) {
        if (!engines[i] || engines[i]->d.n_loc != n || engines[i]->cfg.location != i || engines[i]->d.send_cap != engines[0]->d.send_cap ||
            engines[i]->device != engines[0]->device) FAIL(OCRGPU_EINVAL, "engine %u: the group must be locations 0..n-1 of one n-way partition on one device, equal regions", i);
    }
    bool fresh[1024]; u32 n_peer = 0;
    for (u32 i = 0; i < n; ++i) { fresh[i] = engines[i]->hdr_fresh; engines[i]->hdr_fresh = false; n_peer += engines[i]->peer_on ? 1u : 0u; }
    if (n_peer) { // the group is connected by windows: the same protocol as between processes
        if (n_peer != n) FAIL(OCRGPU_EINVAL, "some engines of the group are connected by peer windows, some are not");
        u64 h[4]; int r = peer_exchange(engines, n, h); if (r) return r;
        if (activity_out) *activity_out = h[1];
        if (sent_out) *sent_out = h[2];
        if (queued_out) *queued_out = h[3];
        return 0;
    }
    for (u32 i = 0; i < n; ++i) {
        HIPCHK(hipSetDevice(engines[i]->device));
        int r = xchg_post(engines[i], fresh[i]); if (r) return r;
    }
    for (u32 i = 0; i < n; ++i) HIPCHK(hipStreamSynchronize(engines[i]->stream)); // the copies below read every engine's regions
    const size_t reg = sizeof(SendRec) * (size_t)engines[0]->d.send_cap;
    for (u32 src = 0; src < n; ++src)
        for (u32 dst = 0; dst < n; ++dst)
            HIPCHK(hipMemcpyAsync((char *)engines[dst]->recv_buf + (size_t)src * reg, (const char *)engines[src]->d.send + (size_t)dst * reg, reg,
                                  hipMemcpyDeviceToDevice, engines[dst]->stream));
    for (u32 i = 0; i < n; ++i) HIPCHK(hipStreamSynchronize(engines[i]->stream)); // (the engines may sit on different streams)
    u64 h[4] = {0, 0, 0, 0};
    for (u32 i = 0; i < n; ++i) { int r = xchg_land(engines[i], engines[i]->recv_buf, false, h); if (r) return r; }
    if (activity_out) *activity_out = h[1];
    if (sent_out) *sent_out = h[2];
    if (queued_out) *queued_out = h[3];
    return 0;
}
// The distributed worker loop (hc-dist's role for this path, hc-dist-policy.c:1027-1031,2208): every partition fires one level,
// exchanges, and applies; a level is closed (no record in flight, no satisfy queued anywhere) before the next one fires, so
// level numbers are those of the single-partition run.  OCRGPU_RX_NO_CASCADE: the caller vouches that a satisfy arriving from
// another partition never sends one on (no event -> remote waiter chains): one exchange per level is then enough.
// One group = the engines of this process (1 with RCCL; several in the lockstep tests).
static double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
// ocrgpu_step without its poll: the launches a step normally needs are queued ahead (pending records, the level, what the
// level's events fire, the launch that notices the end), then the headers; whether the step was really over is read back with
// the exchange's own synchronisation (xchg_land), so a level costs ONE host synchronisation.  A step that was not over says
// so in its headers ("queued"), which brings every partition back for a settle round, and is finished with polls there.
static int ahead_time_flush(ocrgpu_engine *e) { // the k_step launches of the last timed step_ahead, once its events have passed
    if (!e->ahead_timed) return 0;
    e->ahead_timed = false;
    HIPCHK(hipEventSynchronize(e->ev1));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    e->step_ns_total += (u64)(ms * 1e6);
    return 0;
}
static int step_ahead(ocrgpu_engine *e, u32 fire, u32 flags, u32 n_launch, bool timed) {
    ENTER(e);
    timed = timed && e->have_events;
    if (timed) { int r0 = ahead_time_flush(e); if (r0) return r0; }
    Dev &d = e->d;
    int r = seal(e); if (r) return r;
    if (!d.send_cnt || d.n_loc > 1024 || d.send_cap < 2) FAIL(OCRGPU_EINVAL, "exchange not configured");
    if (e->launch_index > 0xFFFF0000u) e->launch_index = 8;
    u32 K = e->launch_index + 1;
    k_begin<<<1, 1, 0, e->stream>>>(d, K, fire, flags & (OCRGPU_STEP_FORCE_LEVEL | OCRGPU_STEP_DROP_EMPTY_LEVEL));
    const bool nanny = (e->cfg.flags & OCRGPU_F_NANNY) != 0;
    const u32 grid = e->step_grid > MAX_GRID ? MAX_GRID : (e->step_grid ? e->step_grid : 256u);
    if (timed) HIPCHK(hipEventRecord(e->ev0, e->stream));
    for (u32 k = 0; k < n_launch; ++k, ++K) {
        launch_step(e, grid, K);
    }
    if (timed) { HIPCHK(hipEventRecord(e->ev1, e->stream)); e->ahead_timed = true; }
    e->launches_total += n_launch; e->launch_index = K - 1;
    if (d.n_loc <= 1) { k_end<<<1, 1, 0, e->stream>>>(d, K - 1); k_owed_bodies<<<grid, BLK, 0, e->stream>>>(d); k_owed_clear<<<1, 1, 0, e->stream>>>(d); } // (with partitions bodies run before the fan-out: nothing is owed)
    if (e->peer_on && d.n_loc > 1) e->end_pending = true; // peer windows: the push kernel of the exchange that follows makes the headers and carries k_end
    else { k_exchange_headers<<<1, 1024, 0, e->stream>>>(d, e->launch_index, d.n_loc <= 1 ? 0u : 1u); e->hdr_fresh = true; }
    HIPCHK(hipGetLastError());
    return 0;
}
static int group_round(ocrgpu_engine **g, u32 n, bool rccl, u32 fire, u32 step_flags, u64 *h, double *tm, int *fault, bool timed = false, u32 n_fire = 4u) {
    const double t0 = now_s();
    const bool polled = getenv("OCRGPU_RX_POLLED") != nullptr; // the two-synchronisation round of before (A/B, debugging)
    for (u32 i = 0; i < n; ++i) {
        const char *nl = getenv("OCRGPU_RX_LAUNCHES"); // tests: a budget of 1 makes every step run out and take the catch-up path
        int r = polled ? ocrgpu_step(g[i], fire, step_flags, nullptr, nullptr, nullptr) : step_ahead(g[i], fire, step_flags, nl ? (u32)atoi(nl) : (fire ? n_fire : 2u), timed);
        if (r == OCRGPU_EFAULT) *fault = r; else if (r) return r;
    }
    const double t1 = now_s();
    u64 rin = 0;
    int r = rccl ? ocrgpu_exchange(g[0], &rin, &h[1], &h[2], &h[3]) : ocrgpu_exchange_local(g, n, &h[1], &h[2], &h[3]);
    if (r) return r;
    for (u32 i = 0; i < n; ++i) {
        if (check_fault(g[i]) == OCRGPU_EFAULT) *fault = OCRGPU_EFAULT;
        const State *S = (const State *)g[i]->pinned_done;
        if (polled || S->done) continue;
        // ran out of launches: finish the step with polls (its fire budget is what the state says is left); what it sends now
        // goes out with the settle round its "queued" header has already asked everybody for
        r = ocrgpu_step(g[i], S->allow_fire, step_flags & OCRGPU_STEP_FORCE_LEVEL & (S->allow_fire ? ~0u : 0u), nullptr, nullptr, nullptr);
        if (r == OCRGPU_EFAULT) *fault = r; else if (r) return r;
    }
    if (tm) { tm[0] += t1 - t0; tm[1] += now_s() - t1; }
    return 0;
}
// The level loop with rounds queued AHEAD of the host (peer windows only: nothing in a round needs the host).  Batches of
// `ahead` rounds; two batches are kept in flight, so the device never waits for a launch, and the host looks at the pinned
// report (k_exchange_apply_hdr: stop / levels) once per batch.  Rounds still queued when the run stops are idle ones.
static int run_ahead_round(ocrgpu_engine **g, u32 n, u32 n_launch, u32 no_cascade, bool first) {
    for (u32 i = 0; i < n; ++i) {
        ocrgpu_engine *e = g[i]; Dev &d = e->d;
        ENTER(e);
        u32 K = e->launch_index + 1; // (no wrap reset inside a run: the previous round's apply has already begun this K)
        if (first) k_begin_auto<<<1, 1, 0, e->stream>>>(d, K, e->peer_cnt); // (afterwards the last workgroup of a round's apply begins the next round)
        const u32 grid = e->step_grid > MAX_GRID ? MAX_GRID : (e->step_grid ? e->step_grid : 256u);
        const u32 nl = e->ra_launches ? e->ra_launches : n_launch;
        for (u32 k = 0; k < nl; ++k, ++K) launch_step(e, grid, K);
        e->launches_total += nl; e->launch_index = K - 1;
        e->end_pending = true; // (k_end and the headers: in the push kernel)
        HIPCHK(hipGetLastError());
    }
    u64 h[4];
    return peer_exchange(g, n, h, true, no_cascade, true);
}
static int resolve_run_ahead(ocrgpu_engine **g, u32 n, u32 flags, u64 *levels_out, int *fault) {
    const char *ev_ = getenv("OCRGPU_RX_AHEAD"), *nl = getenv("OCRGPU_RX_LAUNCHES");
    const u32 ahead = ev_ && atoi(ev_) > 0 ? (u32)atoi(ev_) : 8u, no_cascade = (flags & OCRGPU_RX_NO_CASCADE) ? 1u : 0u;
    const u32 n_launch = nl ? (u32)atoi(nl) : (no_cascade ? 2u : 3u); // (the arrivals, the level; that the run is over is seen without a launch: run_view)
    int r;
    for (u32 i = 0; i < n; ++i) {
        ocrgpu_engine *e = g[i];
        HIPCHK(hipSetDevice(e->device));
        if ((r = seal(e))) return r;
        if (e->launch_index > 0xF0000000u) e->launch_index = 8; // keep K-1 and K%3 arithmetic away from the wrap (checked between runs only)
        if (e->d.n_loc < 2) FAIL(OCRGPU_EINVAL, "rounds are queued ahead for partitioned engines only");
        for (int k = 0; k < 2; ++k) if (!e->ra_ev[k]) HIPCHK(hipEventCreateWithFlags(&e->ra_ev[k], hipEventDisableTiming));
        e->pinned_done[PIN_RA] = e->pinned_done[PIN_RA + 1] = e->pinned_done[PIN_RA + 2] = e->pinned_done[PIN_RA + 3] = 0;
        e->pinned_done[PIN_RA + 4] = e->pinned_done[PIN_RA + 5] = 0;
        HIPCHK(hipMemsetAsync(&e->d.ctrl->noop_steps, 0, 4, e->stream));
        // A round's step gets 3 launches to begin with (arrivals, what they set off, the level); a DAG whose arrivals set nothing off
        // (EDT -> EDT dependences: configs 2-5) needs one, and every launch with nothing to do is 4.5 us in every round.  The device
        // counts such launches and the rounds whose step ran out (it goes on in the next round: exact either way), and the budget
        // follows: down when every round had a launch to spare and none ran out, up when a quarter of them ran out.
        e->ra_launches = (nl || no_cascade) ? 0u : n_launch; e->ra_noop_seen = e->ra_short_seen = e->ra_rounds_seen = 0; e->ra_cool = 0;
        k_ra_init<<<1, 1, 0, e->stream>>>(e->peer_cnt);
    }
    bool stop = false;
    for (u64 batch = 0; !stop; ++batch) {
        for (u32 k = 0; k < ahead; ++k) if ((r = run_ahead_round(g, n, n_launch, no_cascade, batch == 0 && k == 0))) return r;
        for (u32 i = 0; i < n; ++i) { HIPCHK(hipSetDevice(g[i]->device)); HIPCHK(hipEventRecord(g[i]->ra_ev[batch & 1], g[i]->stream)); }
        if (batch == 0) continue; // keep two batches in flight
        for (u32 i = 0; i < n; ++i) {
            HIPCHK(hipEventSynchronize(g[i]->ra_ev[(batch - 1) & 1]));
            // Every partition takes the same decisions from the same sums, and every PROCESS must queue the same number of rounds
            // (a round of one pairs with the same round of the others): so the stop counts only once the batch it fell in is
            // known to be complete -- a host that happens to look late and sees a stop from the batch still running ignores it
            // until the next look.  The run ends after the batch that follows the stopping one, everywhere.
            const u32 at = g[i]->pinned_done[PIN_RA];
            if ((at && (u64)(at - 1) < batch * (u64)ahead) || g[i]->pinned_done[PIN_PEER_LATE]) stop = true;
            ocrgpu_engine *e = g[i];
            if (e->ra_launches) { // (what the words say may be a few rounds newer than the batch waited for: only differences are used)
                const u32 rounds = e->pinned_done[PIN_RA + 3], noop = e->pinned_done[PIN_RA + 4], shrt = e->pinned_done[PIN_RA + 5];
                const u32 dr = rounds - e->ra_rounds_seen, dn = noop - e->ra_noop_seen, ds = shrt - e->ra_short_seen;
                if (dr >= ahead / 2u) {
                    if (4u * ds > dr) { if (e->ra_launches < 3u) e->ra_launches++; e->ra_cool = 3; }
                    else if (e->ra_cool) e->ra_cool--;
                    else if (ds == 0 && dn >= dr && e->ra_launches > 1u) e->ra_launches--;
                    e->ra_rounds_seen = rounds; e->ra_noop_seen = noop; e->ra_short_seen = shrt;
                }
            }
        }
    }
    for (u32 i = 0; i < n; ++i) { HIPCHK(hipSetDevice(g[i]->device)); HIPCHK(hipStreamSynchronize(g[i]->stream)); }
    for (u32 i = 0; i < n; ++i) {
        ocrgpu_engine *e = g[i];
        if (e->pinned_done[PIN_PEER_LATE]) { e->poisoned = true;
                                               FAIL(OCRGPU_EHIP, "peer windows: %u of %u partitions did not arrive at location %u within the limit (OCRGPU_PEER_TIMEOUT_MS); "
                                               "the run cannot continue", e->pinned_done[PIN_PEER_LATE], e->d.n_loc, e->cfg.location); }
        if (!e->pinned_done[PIN_RA]) FAIL(OCRGPU_EHIP, "rounds queued ahead ended without the stop word at location %u", e->cfg.location);
        if (e->pinned_done[PIN_RA + 2] == 0xBAD0BAD1u) FAIL(OCRGPU_EINVAL, "OCRGPU_RX_NO_CASCADE: what arrived at location %u set off records for other partitions (an event whose waiters live elsewhere): "
                                                                            "they would reach them a round late; the run was stopped. Resolve without OCRGPU_RX_NO_CASCADE", e->cfg.location);
        if (e->pinned_done[PIN_RA + 2] == 0xBAD0BAD0u) FAIL(OCRGPU_EINVAL, "OCRGPU_RX_NO_CASCADE: a partition had not fired its level when the others' records of that level arrived "
                                                                            "(its launches went into what the arrivals of the level before set off); the run was stopped. Resolve without "
                                                                            "OCRGPU_RX_NO_CASCADE, or with OCRGPU_RX_NO_AHEAD=1 / a larger OCRGPU_RX_LAUNCHES");
        if (check_fault(e) == OCRGPU_EFAULT) *fault = OCRGPU_EFAULT;
    }
    *levels_out = g[0]->pinned_done[PIN_RA + 1];
    return 0;
}
static int resolve_exchange_group(ocrgpu_engine **g, u32 n, bool rccl, u32 flags, u64 *levels_out) {
    u64 h[4] = {0, 0, 0, 0}, levels = 0;
    double tm[2] = {0, 0}; const bool timing = getenv("OCRGPU_DIST_TIMING") != nullptr;
    int r, fault = 0; const bool timed = (flags & OCRGPU_RX_TIME_STEPS) != 0;
    do { if ((r = group_round(g, n, rccl, 0u, OCRGPU_STEP_WRITE_HEADERS, h, nullptr, &fault, timed))) return r; } while (h[2] || h[3]); // what the main program sent settles first
    for (u32 i = 0; i < n; ++i) { // from here on a level that fires with records already queued for other partitions is noted (Ctrl.nc_sends)
        HIPCHK(hipSetDevice(g[i]->device));
        HIPCHK(hipMemsetAsync(&g[i]->d.ctrl->nc_sends, 0, 4, g[i]->stream));
        g[i]->pinned_done[33058] = 0;
    }
    bool all_peer = g[0]->d.n_loc > 1;
    for (u32 i = 0; i < n; ++i) all_peer = all_peer && g[i]->peer_on;
    bool any_lockable = false;
    for (u32 i = 0; i < n; ++i) any_lockable = any_lockable || g[i]->dbl != nullptr; // (their releases run behind the apply, and the next round's run begins behind those: xchg_land; OCRGPU_RX_LOCKABLE_HOST=1 keeps the host-driven loop)
    if (any_lockable && (flags & OCRGPU_RX_NO_CASCADE)) FAIL(OCRGPU_EINVAL, "OCRGPU_RX_NO_CASCADE with Lockable datablocks: grants and releases are records that set off records");
    const bool queue_ahead = all_peer && (!any_lockable || !getenv("OCRGPU_RX_LOCKABLE_HOST")) && !timed && !timing && !getenv("OCRGPU_RX_POLLED") && !getenv("OCRGPU_RX_NO_AHEAD");
    if (queue_ahead) { if ((r = resolve_run_ahead(g, n, flags, &levels, &fault))) return r; }
    else for (;;) {
        // launches queued for a level: the records that arrived, the level itself (that the run is then over is seen without a
        // launch: run_view); one more where arrivals may fire events that send on (a step that needs more says so and is
        // finished with polls)
        if ((r = group_round(g, n, rccl, 1u, OCRGPU_STEP_FORCE_LEVEL | OCRGPU_STEP_WRITE_HEADERS, h, timing ? tm : nullptr, &fault, timed, (flags & OCRGPU_RX_NO_CASCADE) ? 2u : 3u))) return r;
        u64 act = h[1];
        if (flags & OCRGPU_RX_NO_CASCADE) for (u32 i = 0; i < n; ++i) if (g[i]->pinned_done[33058])
            FAIL(OCRGPU_EINVAL, "OCRGPU_RX_NO_CASCADE: what arrived at location %u set off records for other partitions (an event whose waiters live elsewhere): "
                                "they would reach them a round late; the run was stopped. Resolve without OCRGPU_RX_NO_CASCADE", g[i]->cfg.location);
        if (!(flags & OCRGPU_RX_NO_CASCADE) || h[3])
            while (h[2] || h[3]) { if ((r = group_round(g, n, rccl, 0u, OCRGPU_STEP_WRITE_HEADERS, h, timing ? tm : nullptr, &fault, timed))) return r; act += h[1]; }
        if (act == 0) break; // nobody fired or sent anything: the probe level is dropped below
        ++levels;
    }
    for (u32 i = 0; i < n; ++i) { r = ocrgpu_step(g[i], 0u, OCRGPU_STEP_DROP_EMPTY_LEVEL, nullptr, nullptr, nullptr); if (r == OCRGPU_EFAULT) fault = r; else if (r) return r; }
    for (u32 i = 0; i < n; ++i) if ((r = ahead_time_flush(g[i]))) return r;
    if (timing && g[0]->cfg.location == 0 && levels)
        fprintf(stderr, "[ocrgpu dist timing] us per level: engine step %.1f | exchange (headers, all-to-all, apply, one sync) %.1f | %llu levels\n",
                1e6 * tm[0] / (double)(levels + 1), 1e6 * tm[1] / (double)(levels + 1), (unsigned long long)levels);
    if (levels_out) *levels_out = levels;
    if (fault) FAIL(OCRGPU_EFAULT, "the device flagged user-level errors during the run (ocrgpu_get_stats: err[]); the run completed");
    return 0;
}
extern "C" int ocrgpu_resolve_exchange(ocrgpu_engine_t *e, u32 flags, u64 *levels_out) {
    if (!e) FAIL(OCRGPU_EINVAL, "null engine");
    REFUSE_POISONED(e);
    if (!e->comm && !e->peer_on) FAIL(OCRGPU_EINVAL, "no transport (ocrgpu_comm_init or ocrgpu_peer_connect)");
    return resolve_exchange_group(&e, 1, true, flags, levels_out);
}
extern "C" int ocrgpu_resolve_exchange_local(ocrgpu_engine_t **engines, u32 n, u32 flags, u64 *levels_out) {
    if (!engines || n == 0) FAIL(OCRGPU_EINVAL, "no engines");
    return resolve_exchange_group(engines, n, false, flags, levels_out);
}

// ---- workpile (ocrWorkpileFcts_t.push / pop, ocr/src/inc/ocr-workpile.h:39-84; hcWorkpilePush / hcWorkpilePop
//      ocr/src/workpile/hc/hc-workpile.c:103-150 over deque.c:190-265): a deque of guids kept in HBM.  PUSH appends at the tail,
//      PUSHBACK at the head; POP takes from the tail (the owner's LIFO end), STEAL from the head (FIFO).  One caller at a time per
//      engine, like every other entry point, so the two indices live on the caller's side and the ring holds only the guids. ----
extern "C" int ocrgpu_workpile_push(ocrgpu_engine_t *e, u32 type, u64 n, const u64 *guids, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!guids || type > OCRGPU_WP_PUSHBACK) FAIL(OCRGPU_EINVAL, "bad argument");
    if (!e->wp) {
        e->wp_cap = (u64)e->cfg.max_edt + 1024; // INIT_DEQUE_CAPACITY's role (ocr/src/inc/utils/deque.h:15-18): every EDT can be in it once
        HIPCHK(hipMalloc((void **)&e->wp, 8 * e->wp_cap));
        e->allocs.push_back(e->wp);
        e->wp_head = e->wp_tail = e->wp_cap; // start in the middle of the index space: PUSHBACK moves the head down
    }
    if (e->wp_tail - e->wp_head + n > e->wp_cap) FAIL(OCRGPU_ENOMEM, "workpile full (%llu entries; the reference's deque asserts here, deque.c:194)", (unsigned long long)e->wp_cap);
    const hipMemcpyKind kind = where == OCRGPU_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    // the indices are positions modulo wp_cap: an empty pile goes back to the middle of the index space, and a head that a
    // PUSHBACK would take below zero moves up by one whole turn together with the tail (same slots, no wrap of the u64)
    if (e->wp_tail == e->wp_head) e->wp_head = e->wp_tail = e->wp_cap;
    if (type == OCRGPU_WP_PUSHBACK && e->wp_head < n) { e->wp_head += e->wp_cap; e->wp_tail += e->wp_cap; }
    u64 first = type == OCRGPU_WP_PUSH ? e->wp_tail : e->wp_head - n; // PUSHBACK of n guids: guids[0] ends up at the head
    for (u64 done = 0; done < n;) {
        const u64 pos = (first + done) % e->wp_cap, run = n - done < e->wp_cap - pos ? n - done : e->wp_cap - pos;
        HIPCHK(hipMemcpyAsync(e->wp + pos, guids + done, 8 * run, kind, e->stream));
        done += run;
    }
    if (type == OCRGPU_WP_PUSH) e->wp_tail += n; else e->wp_head -= n;
    if (where == OCRGPU_HOST) HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
extern "C" int ocrgpu_workpile_pop(ocrgpu_engine_t *e, u32 type, u64 max_n, u64 *guids_out, u64 *n_out, int where) {
    ENTER(e);
    if (!n_out || (max_n && !guids_out) || type > OCRGPU_WP_STEAL) FAIL(OCRGPU_EINVAL, "bad argument");
    const u64 have = e->wp ? e->wp_tail - e->wp_head : 0, n = have < max_n ? have : max_n;
    *n_out = n; // 0 = empty: the reference's pop returns NULL_GUID then
    if (n == 0) return 0;
    const hipMemcpyKind kind = where == OCRGPU_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (type == OCRGPU_WP_STEAL) { // from the head, oldest first
        for (u64 done = 0; done < n;) {
            const u64 pos = (e->wp_head + done) % e->wp_cap, run = n - done < e->wp_cap - pos ? n - done : e->wp_cap - pos;
            HIPCHK(hipMemcpyAsync(guids_out + done, e->wp + pos, 8 * run, kind, e->stream));
            done += run;
        }
        e->wp_head += n;
        HIPCHK(hipStreamSynchronize(e->stream));
    } else { // from the tail, newest first: the run [tail-n, tail) is copied out and reversed on the way
        if (where == OCRGPU_DEVICE) FAIL(OCRGPU_ENOTSUP, "OCRGPU_WP_POP fills host arrays only");
        const u64 first = e->wp_tail - n;
        for (u64 done = 0; done < n;) {
            const u64 pos = (first + done) % e->wp_cap, run = n - done < e->wp_cap - pos ? n - done : e->wp_cap - pos;
            HIPCHK(hipMemcpyAsync(guids_out + done, e->wp + pos, 8 * run, kind, e->stream));
            done += run;
        }
        HIPCHK(hipStreamSynchronize(e->stream));
        std::reverse(guids_out, guids_out + n);
        e->wp_tail -= n;
    }
    return 0;
}

extern "C" int ocrgpu_get_stats(ocrgpu_engine_t *e, ocrgpu_stats_t *out) { ENTER(e); if (!out) FAIL(OCRGPU_EINVAL, "null"); return fetch_stats(e, out); }

extern "C" int ocrgpu_snapshot(ocrgpu_engine_t *e) {
    ENTER(e);
    if (e->dbl && e->d.n_loc > 1) FAIL(OCRGPU_ENOTSUP, "ocrgpu_snapshot: Lockable datablocks with partitions (the lines of remote requesters are not part of a snapshot)");
    int r = seal(e); if (r) return r;
    Dev &d = e->d; Snapshot &s = e->snap;
    if (s.valid) { HIPCHK(hipStreamSynchronize(e->stream)); snap_free(s); }
    s.dbl = nullptr; s.tiles = nullptr;
    if (e->tiles && e->n_db) { // tiles are updated in place: they are part of what a re-arm puts back
        HIPCHK(hipMalloc((void **)&s.tiles, 8 * OCROPS_TILE_WORDS * (size_t)e->n_db));
        HIPCHK(hipMemcpyAsync(s.tiles, e->tiles, 8 * OCROPS_TILE_WORDS * (size_t)e->n_db, hipMemcpyDeviceToDevice, e->stream));
    }
    if (e->dbl) {
        HIPCHK(hipMalloc((void **)&s.dbl, sizeof(DbLock) * ((size_t)d.max_db + d.max_edt)));
        HIPCHK(hipMemcpyAsync(s.dbl, e->dbl, sizeof(DbLock) * ((size_t)d.max_db + d.max_edt), hipMemcpyDeviceToDevice, e->stream));
    }
    HIPCHK(hipMalloc((void **)&s.edt, sizeof(EdtLine) * (size_t)e->n_edt + 64));
    HIPCHK(hipMalloc((void **)&s.evt, sizeof(EvtRec) * (size_t)e->n_evt + 64));
    HIPCHK(hipMalloc((void **)&s.counted, 8 * (size_t)e->n_evt + 64));
    HIPCHK(hipMalloc((void **)&s.lazy_tab, 16 * (size_t)e->lazy_cap));
    s.n_slots = e->n_slots;
    HIPCHK(hipMalloc((void **)&s.slots, 4 * (size_t)s.n_slots + 64));
    HIPCHK(hipMemcpyAsync(e->pinned_done + 33012, &d.ctrl->r_tail, 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(e->pinned_done + 33014, &d.ctrl->chan_wait_used, 8, hipMemcpyDeviceToHost, e->stream)); // + chan_sat_used
    HIPCHK(hipStreamSynchronize(e->stream));
    s.n_fired_entries = e->pinned_done[33012];
    s.chan_wait_used = e->pinned_done[33014]; s.chan_sat_used = e->pinned_done[33015]; // CHANNEL rings in use
    if (s.chan_wait_used > d.max_edges) s.chan_wait_used = (u32)d.max_edges;
    if (s.chan_sat_used > d.chan_sat_cap) s.chan_sat_used = d.chan_sat_cap;
    HIPCHK(hipMalloc((void **)&s.chan_wait, 8 * (size_t)s.chan_wait_used + 64));
    HIPCHK(hipMalloc((void **)&s.chan_sat, 4 * (size_t)s.chan_sat_used + 64));
    if (s.chan_wait_used) HIPCHK(hipMemcpyAsync(s.chan_wait, d.waiters + (d.max_edges - s.chan_wait_used), 8 * (size_t)s.chan_wait_used, hipMemcpyDeviceToDevice, e->stream));
    if (s.chan_sat_used) HIPCHK(hipMemcpyAsync(s.chan_sat, d.chan_sat, 4 * (size_t)s.chan_sat_used, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMalloc((void **)&s.fired, 4 * (size_t)s.n_fired_entries + 16));
    HIPCHK(hipMalloc((void **)&s.ctrl, sizeof(Ctrl)));
    s.valid = true; s.n_edt = e->n_edt; s.n_evt = e->n_evt; s.n_db = e->n_db; e->snap_pop_cursor = e->pop_cursor;
    s.edges_upper = e->edges_upper; s.seal_epoch = e->seal_epoch; s.err_seen = e->err_seen; s.mut_epoch = e->mut_epoch;
    HIPCHK(hipMemcpyAsync(s.edt, d.edt, sizeof(EdtLine) * (size_t)e->n_edt, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(s.evt, d.evt, sizeof(EvtRec) * (size_t)e->n_evt, hipMemcpyDeviceToDevice, e->stream));
    if (e->n_evt) HIPCHK(hipMemcpyAsync(s.counted, d.counted, 8 * (size_t)e->n_evt, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(s.lazy_tab, d.lazy_tab, 16 * (size_t)e->lazy_cap, hipMemcpyDeviceToDevice, e->stream));
    if (s.n_slots) HIPCHK(hipMemcpyAsync(s.slots, d.slots, 4 * (size_t)s.n_slots, hipMemcpyDeviceToDevice, e->stream));
    if (s.n_fired_entries) HIPCHK(hipMemcpyAsync(s.fired, d.fired, 4 * (size_t)s.n_fired_entries, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(s.ctrl, d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToDevice, e->stream));
    return 0;
}
extern "C" int ocrgpu_restore(ocrgpu_engine_t *e) {
    ENTER(e);
    Dev &d = e->d; Snapshot &s = e->snap;
    if (!s.valid) FAIL(OCRGPU_EINVAL, "no snapshot taken");
    if (s.n_edt != e->n_edt || s.n_evt != e->n_evt || s.mut_epoch != e->mut_epoch) FAIL(OCRGPU_EINVAL, "objects were created or reclaimed after the snapshot");
    e->poisoned = false; // (the snapshot's state replaces whatever a round that gave up left half applied)
    e->pop_cursor = e->snap_pop_cursor;
    e->n_db = s.n_db; // (proxy datablocks cloned from other partitions since the snapshot go with Ctrl.proxy_used)
    // the registration log is append-only and Ctrl.log_tail goes back with Ctrl: what was registered since the snapshot is
    // forgotten, on the host side too; if the rows were rebuilt meanwhile (waiters[] has another layout than the restored
    // row_begin / deg describe), the next run seals again from the restored log and states
    e->edges_upper = s.edges_upper; e->err_seen = s.err_seen; // Ctrl.err[] goes back with Ctrl
    if (e->seal_epoch != s.seal_epoch || e->csr_dirty) e->csr_dirty = true;
    if (e->d.send_cnt) k_exchange_reset<<<1, 1024, 0, e->stream>>>(e->d);
    if (e->tiles && s.tiles) HIPCHK(hipMemcpyAsync(e->tiles, s.tiles, 8 * OCROPS_TILE_WORDS * (size_t)s.n_db, hipMemcpyDeviceToDevice, e->stream));
    if (e->dbl && s.dbl) HIPCHK(hipMemcpyAsync(e->dbl, s.dbl, sizeof(DbLock) * ((size_t)d.max_db + d.max_edt), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d.edt, s.edt, sizeof(EdtLine) * (size_t)e->n_edt, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d.evt, s.evt, sizeof(EvtRec) * (size_t)e->n_evt, hipMemcpyDeviceToDevice, e->stream));
    if (e->n_evt) HIPCHK(hipMemcpyAsync(d.counted, s.counted, 8 * (size_t)e->n_evt, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d.lazy_tab, s.lazy_tab, 16 * (size_t)e->lazy_cap, hipMemcpyDeviceToDevice, e->stream));
    if (s.chan_wait_used) HIPCHK(hipMemcpyAsync(d.waiters + (d.max_edges - s.chan_wait_used), s.chan_wait, 8 * (size_t)s.chan_wait_used, hipMemcpyDeviceToDevice, e->stream));
    if (s.chan_sat_used) HIPCHK(hipMemcpyAsync(d.chan_sat, s.chan_sat, 4 * (size_t)s.chan_sat_used, hipMemcpyDeviceToDevice, e->stream));
    // pre-slot words carry the "satisfied" marks: overflow words go back to what they were (inline ones are in the lines)
    if (s.n_slots) HIPCHK(hipMemcpyAsync(d.slots, s.slots, 4 * (size_t)s.n_slots, hipMemcpyDeviceToDevice, e->stream));
    if (s.n_fired_entries) HIPCHK(hipMemcpyAsync(d.fired, s.fired, 4 * (size_t)s.n_fired_entries, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d.ctrl, s.ctrl, sizeof(Ctrl), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemsetAsync(d.blk_stats, 0, 8 * 4 * MAX_GRID, e->stream)); // statistics restart with the re-armed DAG
    return 0;
}

// ---- results ---------------------------------------------------------------------------------------------------
extern "C" int ocrgpu_read_edts(ocrgpu_engine_t *e, u64 first, u64 n, s32 *level_out, u64 *val_out, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (first + n > e->n_edt) FAIL(OCRGPU_EINVAL, "EDT range [%llu,%llu) beyond %u created", (unsigned long long)first, (unsigned long long)(first + n), e->n_edt);
    int r = stage_reserve(e, 2 * rup(4 * n) + rup(8 * n)); if (r) return r;
    s32 *dl; u64 *dv;
    if ((r = stage_out_alloc(e, level_out, n, where, &dl)) || (r = stage_out_alloc(e, val_out, n, where, &dv))) return r;
    if (!dl) { dl = (s32 *)(e->stage_dev + e->stage_dev_off); e->stage_dev_off += rup(4 * n); }
    k_fill_i32<<<nblk(n), 256, 0, e->stream>>>(dl, n, -1);
    k_levels<<<2048, 256, 0, e->stream>>>(e->d, first, n, dl);
    if (dv) k_read_vals<<<nblk(n), 256, 0, e->stream>>>(e->d, first, n, dl, dv);
    if ((r = stage_out_copy(e, level_out, dl, n, where)) || (r = stage_out_copy(e, val_out, dv, n, where))) return r;
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
__global__ void k_digest_edts(Dev d, u64 first, u64 n, const s32 *level, const u64 *val, u64 *out) {
    u64 acc = 0;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        acc += ocrgpu_digest_term(d.edt[first + i].param0, level[i], val[i]);
    for (int o = 1; o < 64; o <<= 1) acc += __shfl_xor(acc, o);
    if (lane_id() == 0 && acc) atomicAdd((unsigned long long *)out, (unsigned long long)acc);
}
extern "C" int ocrgpu_digest_edts(ocrgpu_engine_t *e, u64 first, u64 n, u64 *digest_out) {
    ENTER(e);
    if (!digest_out) FAIL(OCRGPU_EINVAL, "null argument");
    *digest_out = 0;
    if (n == 0) return 0;
    if (first + n > e->n_edt) FAIL(OCRGPU_EINVAL, "EDT range [%llu,%llu) beyond %u created", (unsigned long long)first, (unsigned long long)(first + n), e->n_edt);
    int r = stage_reserve(e, rup(4 * n) + rup(8 * n) + 256); if (r) return r;
    s32 *dl = (s32 *)(e->stage_dev + e->stage_dev_off); e->stage_dev_off += rup(4 * n);
    u64 *dv = (u64 *)(e->stage_dev + e->stage_dev_off); e->stage_dev_off += rup(8 * n);
    u64 *dd = (u64 *)(e->stage_dev + e->stage_dev_off); e->stage_dev_off += 256;
    HIPCHK(hipMemsetAsync(dd, 0, 8, e->stream));
    k_fill_i32<<<nblk(n), 256, 0, e->stream>>>(dl, n, -1);
    k_levels<<<2048, 256, 0, e->stream>>>(e->d, first, n, dl);
    k_read_vals<<<nblk(n), 256, 0, e->stream>>>(e->d, first, n, dl, dv);
    k_digest_edts<<<2048, 256, 0, e->stream>>>(e->d, first, n, dl, dv, dd);
    HIPCHK(hipMemcpyAsync(digest_out, dd, 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
extern "C" int ocrgpu_read_events(ocrgpu_engine_t *e, int which, u64 first, u64 n, u64 *state_out, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!state_out) FAIL(OCRGPU_EINVAL, "null array");
    if (which == 0 ? first + n > e->n_evt : first + n > e->n_edt) FAIL(OCRGPU_EINVAL, "event range out of bounds");
    int r = stage_reserve(e, rup(8 * n)); if (r) return r;
    u64 *ds; if ((r = stage_out_alloc(e, state_out, n, where, &ds))) return r;
    k_read_events<<<nblk(n), 256, 0, e->stream>>>(e->d, (u32)(which ? e->d.max_evt + first : first), n, ds);
    if ((r = stage_out_copy(e, state_out, ds, n, where))) return r;
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
extern "C" int ocrgpu_read_dbs(ocrgpu_engine_t *e, int which, u64 first, u64 n, u64 *value_out, int where) {
    ENTER(e);
    if (n == 0) return 0;
    if (!value_out) FAIL(OCRGPU_EINVAL, "null array");
    if (which == 0 ? first + n > e->n_db : first + n > e->n_edt) FAIL(OCRGPU_EINVAL, "DB range out of bounds");
    int r = stage_reserve(e, rup(8 * n) + rup(4 * n)); if (r) return r;
    u64 *dv; if ((r = stage_out_alloc(e, value_out, n, where, &dv))) return r;
    s32 *dl = nullptr;
    if (which) {
        dl = (s32 *)(e->stage_dev + e->stage_dev_off); e->stage_dev_off += rup(4 * n);
        k_fill_i32<<<nblk(n), 256, 0, e->stream>>>(dl, n, -1);
        k_levels<<<2048, 256, 0, e->stream>>>(e->d, first, n, dl);
    }
    k_read_dbs<<<nblk(n), 256, 0, e->stream>>>(e->d, (u32)(which ? e->d.max_db + first : first), n, dl, (u32)first, dv);
    if ((r = stage_out_copy(e, value_out, dv, n, where))) return r;
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
extern "C" int ocrgpu_read_fired(ocrgpu_engine_t *e, u64 *n_fired, u64 *n_levels, u32 *ids_out, u64 *level_start_out, int where) {
    ENTER(e);
    if (where == OCRGPU_DEVICE) FAIL(OCRGPU_ENOTSUP, "ocrgpu_read_fired fills host arrays only");
    Ctrl h;
    HIPCHK(hipMemcpyAsync(&h, e->d.ctrl, sizeof h, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    u64 ne = h.r_end, nl = h.n_levels > e->d.max_levels ? e->d.max_levels : h.n_levels;
    // the device log has holes (edges that did not ready their EDT): squeeze them out level by level
    std::vector<u32> raw(ne ? ne : 1), ls(nl + 1);
    if (ne) HIPCHK(hipMemcpyAsync(raw.data(), e->d.fired, 4 * ne, hipMemcpyDeviceToHost, e->stream));
    if (nl) HIPCHK(hipMemcpyAsync(ls.data(), e->d.level_start, 4 * (nl + 1), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    u64 w = 0;
    for (u64 L = 0; L < nl; ++L) {
        if (level_start_out) level_start_out[L] = w;
        for (u64 i = ls[L]; i < ls[L + 1]; ++i) if (raw[i] != INVALID) { if (ids_out) ids_out[w] = raw[i]; ++w; }
    }
    if (level_start_out) level_start_out[nl] = w;
    if (n_fired) *n_fired = w;
    if (n_levels) *n_levels = nl;
    return 0;
}
