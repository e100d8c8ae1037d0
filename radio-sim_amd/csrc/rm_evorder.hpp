// rm_evorder.hpp -- the pop order of the reference's event queue as a sort key (host + device)
// (part of libradiomedium_hip.so; overview at the top of rm_engine.h)
//
// The reference keeps its reception / transmission events in a three-tier "ladder" queue
// (com/botbox/scheduler/EventQueue.java) and drains it at the end of a tick with
// Simulator.processAllEvents (Simulator.java:213-228: pop while nextTime < time).  Equal timestamps are
// common -- all receivers of a frame share its start and its end -- and their order is observable (the
// order of deliverRadioPacket calls; "A ends at t" vs "B starts at t" on one node).  The queue's pop
// order has a closed form, which is what lets the reception stage run as a parallel sort:
//
//     (time ascending, ladder ascending, insertion DEscending)
//
//  * insertBottom puts an event before the first queued one with time >= its own (EventQueue.java:215-231),
//    bucket -> bottom transfers re-insert one by one (:186-192), buckets and the top list keep insertion
//    order (:89-91, :112-117, :311-317): inside one ladder equal timestamps pop in reverse insertion order;
//  * an event whose time is >= topStart when it is added waits in the top list (:83) until the NEXT
//    moveTop builds a ladder from it (:329-337, topStart = maxTS), and that only happens once the
//    current ladder is empty (:156-167) -- so it pops after every equal-time event of the current ladder.
//    "ladder" = moveTop calls before the insertion, plus one if time >= topStart at that moment.
//  * moveTop runs in a drain to time T iff the top list is not empty and the current ladder holds no
//    event >= T, i.e. its maxTS (= topStart) is below T -- or no ladder exists yet.
//
// The tests check this closed form against a literal restatement of the Java queue on randomized
// schedules, and THIS header against it through rm_evq_* (tests/test_host_logic.py).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define RM_EVHD __host__ __device__ inline
#else
#define RM_EVHD inline
#endif

namespace rm {

struct EvOrder {
    int64_t top_start;    // EventQueue.topStart
    int64_t top_max;      // maxTS of the events now in the top list
    int32_t ladders;      // moveTop calls so far
    int32_t top_nonempty; // numTop > 0
};

// the ladder an event added now with time `t` belongs to (the caller also notes it in the top list: ev_note_top)
RM_EVHD int32_t ev_ladder(const EvOrder &o, int64_t t) { return (t >= o.top_start) ? o.ladders + 1 : o.ladders; }

RM_EVHD void ev_note_top(EvOrder &o, int64_t t)
{
    if (t < o.top_start) return;
    if (!o.top_nonempty || t > o.top_max) o.top_max = t;
    o.top_nonempty = 1;
}

// Simulator.processAllEvents(T) seen from the queue's structure: does it build a new ladder?
RM_EVHD void ev_drain(EvOrder &o, int64_t T)
{
    if (o.top_nonempty && (o.ladders == 0 || o.top_start < T)) {
        o.ladders += 1;
        o.top_start = o.top_max;
        o.top_nonempty = 0;
    }
}

} // namespace rm
