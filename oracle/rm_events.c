/*
 * rm_events.c -- CPU oracle, part 2 (TEST INFRASTRUCTURE; see rm_oracle.h for the rules):
 * what happens to the verdicts after RadioMedium.transmit -- the reference's event queue, the
 * simulator's event generation / tick-end drain and the receiver state machine, restated line by
 * line in plain C as a SERIAL replay.  Reference paths relative to /root/reference/radio-medium/java/:
 *
 *   com/botbox/scheduler/EventQueue.java:35-339   three-tier "ladder" queue (top list, rungs of
 *                                                 20 buckets, sorted bottom list)
 *   com/botbox/scheduler/Rung.java, TimeEvent.java
 *   se/sics/emul8/radiomedium/Simulator.java:155-165 (emulatorTimeStepDone), :213-228 (nextEvent,
 *       processAllEvents: pops while nextTime < time, STRICT), :321-350 (generate*Events)
 *   se/sics/emul8/radiomedium/events/ReceptionEvent.java:35-46, TransmissionEvent.java:18-26
 *   se/sics/emul8/radiomedium/Transciever.java:52-113
 *
 * "parity unpinned": the reference has no tests for any of this; the restatement is pinned by
 * source-derived known answers (tests/test_oracle_events.py): equal timestamps pop in reverse
 * insertion order (insertBottom inserts before the first element with time >= t, :215-231), a
 * zero-air-time packet executes its end before its start and leaves the receiver RECEIVING
 * (SURVEY.md section 3.2), processAllEvents' strict "<", and the moveTop boundary (topStart = maxTS,
 * :329-337) that makes a later event with time == maxTS pop AFTER the earlier ones.
 */
#include "rm_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ TimeEvent / Rung */

typedef struct orc_tev {
    int64_t time;              /* TimeEvent.time */
    struct orc_tev *nextEvent; /* TimeEvent.nextEvent */
    struct orc_tev *prevEvent; /* TimeEvent.prevEvent (never set by the queue; checked by addEvent) */
    /* payload: ReceptionEvent / TransmissionEvent fields */
    int32_t node, pkt, kind;
    double rssi;
    int64_t seq;               /* insertion number (for tests) */
} orc_tev_t;

#define BUCKET_NUMBER 20   /* EventQueue.java:42 */
#define SPAWN_THRESHOLD 20 /* EventQueue.java:39 */

typedef struct {
    int64_t bucketWidth;     /* Rung: int bucketWidth (1 + (int)...) -- kept in 64 bits, same values */
    int64_t bucketStartTime; /* "cursor" of the current bucket */
    int64_t startTime;
    int numBucket[BUCKET_NUMBER];
    int numTotal;
    orc_tev_t *bucketFirst[BUCKET_NUMBER];
    orc_tev_t *bucketLast[BUCKET_NUMBER];
} orc_rung_t;

typedef struct {
    int64_t maxTS, minTS; /* EventQueue.java:45-47 */
    int numTop;
    int64_t topStart;
    orc_tev_t *topFirst, *topLast;
    orc_rung_t **rungs; /* Rung[] rungs = new Rung[20] */
    int rungsLength;
    int rungCount;
    orc_rung_t *currentRung;
    int numRung;
    int numBottom;
    orc_tev_t *firstBottom;
    int64_t lastPopTime;
    int error; /* an exception the Java code would have thrown (1 already scheduled, 2 backwards in time, 3 index) */
    int64_t moveTops; /* statistics for the tests: how often moveTop ran */
} orc_evq_t;

static void evq_init(orc_evq_t *q)
{
    memset(q, 0, sizeof(*q));
    q->rungsLength = 20;
    q->rungs = (orc_rung_t **)calloc((size_t)q->rungsLength, sizeof(orc_rung_t *));
}

static void evq_free(orc_evq_t *q)
{
    for (int i = 0; i < q->rungCount; ++i) free(q->rungs[i]);
    free(q->rungs);
}

/* EventQueue.insertBottom :206-244 */
static void insertBottom(orc_evq_t *q, orc_tev_t *event)
{
    int64_t time = event->time;
    if (q->numBottom == 0) {
        q->firstBottom = event;
        event->nextEvent = NULL;
    } else {
        orc_tev_t *evt = q->firstBottom;
        orc_tev_t *last = NULL;
        while (evt != NULL && evt->time < time) {
            last = evt;
            evt = evt->nextEvent;
        }
        if (last == NULL) {
            last = q->firstBottom;
            q->firstBottom = event;
            event->nextEvent = last;
        } else {
            last->nextEvent = event;
            event->nextEvent = evt;
        }
    }
    q->numBottom++;
}

/* EventQueue.addEvent :70-127 */
static void addEvent(orc_evq_t *q, orc_tev_t *event)
{
    if (event->nextEvent != NULL || event->prevEvent != NULL) {
        q->error = 1; /* IllegalStateException("Event already scheduled") */
        return;
    }
    if (event->time < q->lastPopTime) {
        q->error = 2; /* IllegalArgumentException("Can not insert a time value backwards in time") */
        return;
    }
    int64_t time = event->time;
    if (time >= q->topStart) {
        if (q->topFirst == NULL) {
            q->topFirst = event;
            q->topLast = event;
            q->maxTS = q->minTS = time;
        } else {
            q->topLast->nextEvent = event;
            q->topLast = event;
        }
        if (time > q->maxTS) q->maxTS = time;
        if (time < q->minTS) q->minTS = time;
        q->numTop++;
    } else {
        int rung = 0;
        while (rung < q->numRung && time < q->rungs[rung]->bucketStartTime) rung++;
        if (rung < q->numRung) {
            orc_rung_t *cRung = q->rungs[rung];
            int64_t bi = (time - cRung->startTime) / cRung->bucketWidth;
            if (bi < 0 || bi >= BUCKET_NUMBER) {
                q->error = 3; /* ArrayIndexOutOfBoundsException */
                return;
            }
            int bucketIndex = (int)bi;
            if (cRung->numBucket[bucketIndex] == 0) {
                cRung->bucketFirst[bucketIndex] = cRung->bucketLast[bucketIndex] = event;
            } else {
                cRung->bucketLast[bucketIndex]->nextEvent = event;
                cRung->bucketLast[bucketIndex] = event;
            }
            cRung->numBucket[bucketIndex]++;
            cRung->numTotal++;
        } else {
            insertBottom(q, event);
        }
    }
}

/* EventQueue.createRung(TimeEvent first, long rs, int bw) :285-326 */
static void createRung(orc_evq_t *q, orc_tev_t *first, int64_t rs, int64_t bw)
{
    q->numRung++;
    orc_tev_t *last;
    if (q->rungCount < q->numRung) {
        if (q->rungCount == q->rungsLength) {
            int newCapacity = (q->rungCount * 3) / 2 + 1;
            q->rungs = (orc_rung_t **)realloc(q->rungs, (size_t)newCapacity * sizeof(orc_rung_t *));
            for (int i = q->rungsLength; i < newCapacity; ++i) q->rungs[i] = NULL;
            q->rungsLength = newCapacity;
        }
        q->rungs[q->rungCount++] = (orc_rung_t *)calloc(1, sizeof(orc_rung_t)); /* new Rung(BUCKET_NUMBER) */
    }
    q->currentRung = q->rungs[q->numRung - 1];
    q->currentRung->startTime = rs;
    q->currentRung->bucketWidth = bw;
    while (first != NULL) {
        int64_t bi = (first->time - rs) / bw;
        if (bi < 0 || bi >= BUCKET_NUMBER) {
            q->error = 3;
            return;
        }
        int bucketIndex = (int)bi;
        if (q->currentRung->bucketFirst[bucketIndex] == NULL) {
            q->currentRung->bucketFirst[bucketIndex] = q->currentRung->bucketLast[bucketIndex] = first;
        } else {
            q->currentRung->bucketLast[bucketIndex]->nextEvent = first;
            q->currentRung->bucketLast[bucketIndex] = first;
        }
        last = first;
        first = first->nextEvent;
        last->nextEvent = NULL;
        q->currentRung->numBucket[bucketIndex]++;
        q->currentRung->numTotal++;
    }
}

static int findBucket(orc_evq_t *q);

/* EventQueue.createRung(int bucketIndex) :272-283 */
static void createRungFromBucket(orc_evq_t *q, int bucketIndex)
{
    orc_rung_t *cur = q->currentRung;
    orc_tev_t *newRungStart = cur->bucketFirst[bucketIndex];
    cur->bucketFirst[bucketIndex] = cur->bucketLast[bucketIndex] = NULL;
    cur->numTotal -= cur->numBucket[bucketIndex];
    cur->numBucket[bucketIndex] = 0;
    orc_rung_t *oldCurrent = cur;
    createRung(q, newRungStart, cur->bucketStartTime, 1 + (cur->bucketWidth / BUCKET_NUMBER));
    oldCurrent->bucketStartTime += oldCurrent->bucketWidth;
}

/* EventQueue.findBucket :246-270 */
static int findBucket(orc_evq_t *q)
{
    int bucketIndex = 0;
    orc_rung_t *cur = q->currentRung;
    cur->bucketStartTime = cur->startTime;
    while (cur->numBucket[bucketIndex] == 0) {
        bucketIndex++;
        cur->bucketStartTime += cur->bucketWidth;
        if (bucketIndex >= BUCKET_NUMBER) { /* Java: ArrayIndexOutOfBoundsException on the next read */
            q->error = 3;
            return -1;
        }
    }
    if (cur->bucketWidth > 1 && cur->numBucket[bucketIndex] > SPAWN_THRESHOLD) {
        createRungFromBucket(q, bucketIndex);
        if (q->error) return -1;
        return findBucket(q);
    }
    return bucketIndex;
}

/* EventQueue.moveBucket :173-204 */
static void moveBucket(orc_evq_t *q)
{
    int bucket = findBucket(q);
    if (bucket < 0) return;
    orc_rung_t *cur = q->currentRung;
    orc_tev_t *evt = cur->bucketFirst[bucket];
    cur->bucketFirst[bucket] = cur->bucketLast[bucket] = NULL;
    cur->numTotal -= cur->numBucket[bucket];
    cur->numBucket[bucket] = 0;
    cur->bucketStartTime += cur->bucketWidth;
    orc_tev_t *next = NULL;
    while (evt != NULL) {
        next = evt->nextEvent;
        insertBottom(q, evt);
        evt = next;
    }
    while (q->currentRung != NULL && q->currentRung->numTotal == 0) {
        q->numRung--;
        if (q->numRung == 0) q->currentRung = NULL;
        else q->currentRung = q->rungs[q->numRung - 1];
    }
}

/* EventQueue.moveTop :329-337 */
static void moveTop(orc_evq_t *q)
{
    int64_t bw = 1 + (int64_t)(int32_t)((q->maxTS - q->minTS) / BUCKET_NUMBER); /* 1 + (int) (...) */
    q->topStart = q->maxTS;
    q->numTop = 0;
    createRung(q, q->topFirst, q->minTS, bw);
    q->topFirst = NULL;
    q->moveTops++;
}

/* EventQueue.getFirst :145-171 */
static orc_tev_t *getFirst(orc_evq_t *q, int remove)
{
    if (q->error) return NULL;
    if (q->numBottom > 0) {
        orc_tev_t *retVal = q->firstBottom;
        if (remove) {
            q->firstBottom = q->firstBottom->nextEvent;
            q->numBottom--;
            q->lastPopTime = retVal->time;
            retVal->nextEvent = NULL;
            retVal->prevEvent = NULL;
        }
        return retVal;
    } else if (q->numRung > 0) {
        moveBucket(q);
        if (q->numBottom > 0) return getFirst(q, remove);
    } else if (q->numTop > 0) {
        moveTop(q);
        if (q->error) return NULL;
        moveBucket(q);
        if (q->numBottom > 0) return getFirst(q, remove);
    }
    return NULL;
}

/* EventQueue.nextTime :137-143 */
static int64_t nextTime(orc_evq_t *q)
{
    orc_tev_t *e = getFirst(q, 0);
    if (e != NULL) return e->time;
    return -1;
}

/* ------------------------------------------------------------------ Simulator + Transciever */

struct orc_sim {
    orc_evq_t eventQueue;   /* Simulator.java:57 */
    int64_t currentTime;    /* :69 */
    int32_t n;
    /* Transciever.java:14-16 per node: receivingPacket / sendingPacket (packet ids, -1 = null), receivingRSSI */
    int32_t *receivingPacket, *sendingPacket;
    double *receivingRSSI;
    int64_t seq;
    int64_t live; /* events allocated and not yet executed */
};

orc_sim_t *orc_sim_create(int32_t n_nodes)
{
    orc_sim_t *s = (orc_sim_t *)calloc(1, sizeof(orc_sim_t));
    evq_init(&s->eventQueue);
    s->n = n_nodes;
    s->receivingPacket = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_nodes > 0 ? n_nodes : 1));
    s->sendingPacket = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_nodes > 0 ? n_nodes : 1));
    s->receivingRSSI = (double *)calloc((size_t)(n_nodes > 0 ? n_nodes : 1), sizeof(double));
    for (int32_t i = 0; i < n_nodes; ++i) s->receivingPacket[i] = s->sendingPacket[i] = -1;
    return s;
}

void orc_sim_destroy(orc_sim_t *s)
{
    if (!s) return;
    /* events still queued */
    orc_evq_t *q = &s->eventQueue;
    orc_tev_t *lists[2] = {q->topFirst, q->firstBottom};
    for (int l = 0; l < 2; ++l)
        for (orc_tev_t *e = lists[l]; e;) {
            orc_tev_t *n = e->nextEvent;
            free(e);
            e = n;
        }
    for (int r = 0; r < q->rungCount; ++r)
        for (int b = 0; b < BUCKET_NUMBER; ++b)
            for (orc_tev_t *e = q->rungs[r]->bucketFirst[b]; e;) {
                orc_tev_t *n = e->nextEvent;
                free(e);
                e = n;
            }
    evq_free(q);
    free(s->receivingPacket);
    free(s->sendingPacket);
    free(s->receivingRSSI);
    free(s);
}

int64_t orc_sim_time(const orc_sim_t *s) { return s->currentTime; }
int32_t orc_sim_error(const orc_sim_t *s) { return s->eventQueue.error; }
int64_t orc_sim_move_tops(const orc_sim_t *s) { return s->eventQueue.moveTops; }
int64_t orc_sim_top_start(const orc_sim_t *s) { return s->eventQueue.topStart; }

static orc_tev_t *new_event(orc_sim_t *s, int64_t time, int32_t node, int32_t pkt, int32_t kind, double rssi)
{
    orc_tev_t *e = (orc_tev_t *)calloc(1, sizeof(orc_tev_t));
    e->time = time;
    e->node = node;
    e->pkt = pkt;
    e->kind = kind;
    e->rssi = rssi;
    e->seq = s->seq++;
    s->live++;
    return e;
}

/* Simulator.generateReceptionEvents :321-335 */
void orc_sim_reception_events(orc_sim_t *s, int32_t pkt, int32_t dst, int64_t start_us, int64_t air_us, double rssi, int32_t do_deliver)
{
    int64_t packetTime = start_us;
    if (packetTime < s->currentTime) packetTime = s->currentTime;
    orc_tev_t *teStart = new_event(s, packetTime, dst, pkt, ORC_EV_RX_START, rssi);
    orc_tev_t *teEnd = new_event(s, packetTime + air_us, dst, pkt, do_deliver ? ORC_EV_RX_END_DELIVERY : ORC_EV_RX_END_INTERFERENCE, rssi);
    addEvent(&s->eventQueue, teStart);
    addEvent(&s->eventQueue, teEnd);
}

/* Simulator.generateTransmissionEvents :337-350 */
void orc_sim_transmission_events(orc_sim_t *s, int32_t pkt, int32_t src, int64_t start_us, int64_t air_us)
{
    int64_t packetTime = start_us;
    if (packetTime < s->currentTime) packetTime = s->currentTime;
    orc_tev_t *teStart = new_event(s, packetTime, src, pkt, ORC_EV_TX_START, 0.0);
    orc_tev_t *teEnd = new_event(s, packetTime + air_us, src, pkt, ORC_EV_TX_END, 0.0);
    addEvent(&s->eventQueue, teStart);
    addEvent(&s->eventQueue, teEnd);
}

/* Transciever.java:80-88, 106-113 */
static void clearSending(orc_sim_t *s, int32_t node) { s->sendingPacket[node] = -1; }
static void clearReceiving(orc_sim_t *s, int32_t node) { s->receivingPacket[node] = -1; }
static void setReceiving(orc_sim_t *s, int32_t node, int32_t pkt, double rssi)
{
    clearSending(s, node);
    s->receivingPacket[node] = pkt;
    s->receivingRSSI[node] = rssi;
}
static void setSending(orc_sim_t *s, int32_t node, int32_t pkt)
{
    clearReceiving(s, node);
    s->sendingPacket[node] = pkt;
}

/* ReceptionEvent.execute :35-46 ; TransmissionEvent.execute :18-26 */
static void execute(orc_sim_t *s, const orc_tev_t *e)
{
    switch (e->kind) {
    case ORC_EV_RX_START: setReceiving(s, e->node, e->pkt, e->rssi); break;
    case ORC_EV_RX_END_INTERFERENCE: clearReceiving(s, e->node); break;
    case ORC_EV_RX_END_DELIVERY:
        clearReceiving(s, e->node);
        /* simulator.deliverRadioPacket(packet, destination, rssi): reported through out_events */
        break;
    case ORC_EV_TX_START: setSending(s, e->node, e->pkt); break;
    case ORC_EV_TX_END: clearSending(s, e->node); break;
    default: break;
    }
}

/* Simulator.emulatorTimeStepDone :155-165 = currentTime = stepTime; processAllEvents(currentTime) :224-228,
 * with nextEvent :213-221 (pop while 0 <= nextTime < time).  Executed events are reported in pop order. */
int64_t orc_sim_step(orc_sim_t *s, int64_t time, orc_event_t *out_events, int64_t cap)
{
    s->currentTime = time;
    int64_t n = 0;
    for (;;) {
        int64_t nt = nextTime(&s->eventQueue);
        if (!(nt >= 0 && nt < time)) break;
        orc_tev_t *e = getFirst(&s->eventQueue, 1);
        if (!e) break;
        execute(s, e);
        if (out_events && n < cap) {
            out_events[n].time = e->time;
            out_events[n].node = e->node;
            out_events[n].pkt = e->pkt;
            out_events[n].kind = e->kind;
            out_events[n].rssi = e->rssi;
        }
        n++;
        s->live--;
        free(e);
    }
    return n;
}

int64_t orc_sim_pending(const orc_sim_t *s) { return s->live; }

/* What a reference medium does with one evaluated pass (orc_tick's output), in the reference's order: per
 * packet generateTransmissionEvents (UDGMRadioMedium.java:97, N2NRadioMedium.java:53, NullRadioMedium.java:59),
 * then generateReceptionEvents per heard receiver in node order (:99-111).  The constant-loss medium queues
 * nothing (UDGMConstantLossRadioMedium.java:25-33: deliverRadioPacket at once); its links are returned as they are. */
void orc_sim_medium_calls(orc_sim_t *s, const orc_packet_t *packets, int32_t n_packets, int32_t pkt_base, int64_t n_links,
                          const int32_t *out_pkt, const int32_t *out_dst, const uint8_t *out_verdict, const double *out_rssi,
                          int32_t const_loss)
{
    int64_t k = 0;
    for (int32_t q = 0; q < n_packets; ++q) {
        const orc_packet_t *p = &packets[q];
        if (!const_loss) orc_sim_transmission_events(s, pkt_base + q, p->src, p->start_us, p->air_us);
        while (k < n_links && out_pkt[k] == q) {
            if (!const_loss)
                orc_sim_reception_events(s, pkt_base + q, out_dst[k], p->start_us, p->air_us, out_rssi[k],
                                         out_verdict[k] == ORC_DELIVERED);
            k++;
        }
    }
}

/* Transciever.getRSSI :52-61 (medium present: its base RSSI) */
double orc_sim_rssi(const orc_sim_t *s, int32_t node, double base_rssi)
{
    if (s->receivingPacket[node] >= 0) return s->receivingRSSI[node];
    return base_rssi;
}

/* Transciever.getReceivingState :67-78 */
int32_t orc_sim_receiving_state(const orc_sim_t *s, int32_t node, int32_t enabled)
{
    if (!enabled) return 3;                         /* DISABLED */
    if (s->receivingPacket[node] >= 0) return 2;    /* RECEIVING */
    if (s->sendingPacket[node] >= 0) return 1;      /* TRANSMITTING */
    return 0;                                       /* LISTENING */
}

int32_t orc_sim_receiving_packet(const orc_sim_t *s, int32_t node) { return s->receivingPacket[node]; }
int32_t orc_sim_sending_packet(const orc_sim_t *s, int32_t node) { return s->sendingPacket[node]; }

/* a bare queue for the ordering tests: events are (time, id) pairs */
int64_t orc_evq_replay(const int64_t *op_time, const int32_t *op_kind, int64_t n_ops, int64_t *out_id, int64_t cap)
{
    /* op_kind 0: addEvent(time = op_time[i]) with id = running insertion number
     * op_kind 1: pop everything with time < op_time[i]  (Simulator.processAllEvents) */
    orc_sim_t *s = orc_sim_create(1);
    int64_t n = 0;
    for (int64_t i = 0; i < n_ops && !s->eventQueue.error; ++i) {
        if (op_kind[i] == 0) {
            addEvent(&s->eventQueue, new_event(s, op_time[i], 0, 0, -1, 0.0));
        } else {
            for (;;) {
                int64_t nt = nextTime(&s->eventQueue);
                if (!(nt >= 0 && nt < op_time[i])) break;
                orc_tev_t *e = getFirst(&s->eventQueue, 1);
                if (!e) break;
                if (n < cap) out_id[n] = e->seq;
                n++;
                s->live--;
                free(e);
            }
        }
    }
    if (s->eventQueue.error) n = -(int64_t)s->eventQueue.error;
    orc_sim_destroy(s);
    return n;
}
