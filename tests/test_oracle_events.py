"""Known answers for the oracle's serial replay of the reference's event path (oracle/rm_events.c):
com/botbox/scheduler/EventQueue.java (literal restatement), Simulator.generate*Events / processAllEvents
(Simulator.java:213-228, 321-350), ReceptionEvent / TransmissionEvent.execute and the Transciever state machine.
The reference has no tests for any of it; the answers below are derived by hand from the cited lines.

The last tests pin the closed form the engine's reception stage uses instead of a queue: the pop order is
(time ascending, ladder ascending, insertion DEscending), where an event's "ladder" is the number of
EventQueue.moveTop calls before its insertion, plus one if its time >= topStart at that moment (it then waits in
the top list for the next moveTop).  Checked against the literal queue on randomized schedules."""
import numpy as np
import pytest

from oracle import oracle as O


def test_equal_timestamps_pop_in_reverse_insertion_order():
    # insertBottom inserts before the first element with time >= t (EventQueue.java:215-231)
    got = O.evq_replay([("add", 100), ("add", 100), ("add", 50), ("add", 100), ("pop", 1000)])
    assert list(got) == [2, 3, 1, 0]


def test_process_all_events_is_strict():
    # Simulator.nextEvent: nextTime < time (Simulator.java:216): an event AT the step time stays queued
    got = O.evq_replay([("add", 10), ("add", 20), ("pop", 20), ("pop", 21)])
    assert list(got) == [0, 1]
    s = O.Sim(2)
    s.reception_events(7, 1, 0, 320, -3.5, True)
    ev = s.step(320)
    assert [int(e["kind"]) for e in ev] == [O.EV_RX_START]
    assert s.receiving_state(1) == 2 and s.rssi(1) == -3.5          # RECEIVING, latched rssi (Transciever.java:52-61)
    ev = s.step(321)
    assert [int(e["kind"]) for e in ev] == [O.EV_RX_END_DELIVERY]
    assert s.receiving_state(1) == 0 and s.rssi(1, -100.0) == -100.0  # LISTENING, the medium's base RSSI
    s.close()


def test_zero_air_time_packet_leaves_the_receiver_receiving():
    # start and end carry the same time, the end was inserted last and pops first:
    # clearReceiving, then setReceiving -> stuck in RECEIVING; the delivery still happens (on the end flank)
    s = O.Sim(3)
    s.transmission_events(0, 0, 1000, 0)
    s.reception_events(0, 2, 1000, 0, 1.5, True)
    ev = s.step(2000)
    assert [(int(e["kind"]), int(e["node"])) for e in ev] == [
        (O.EV_RX_END_DELIVERY, 2), (O.EV_RX_START, 2), (O.EV_TX_END, 0), (O.EV_TX_START, 0)]
    assert s.receiving_state(2) == 2 and s.rssi(2) == 1.5
    assert s.receiving_state(0) == 1        # TRANSMITTING: the end ran before the start
    s.close()


def test_event_time_is_max_of_start_and_current_time():
    # Simulator.java:323-326
    s = O.Sim(2)
    s.step(5000)
    s.reception_events(0, 1, 1000, 320, 0.0, True)
    ev = s.step(5321)
    assert [(int(e["time"]), int(e["kind"])) for e in ev] == [(5000, O.EV_RX_START), (5320, O.EV_RX_END_DELIVERY)]
    s.close()


def test_set_receiving_clears_sending_and_set_sending_clears_receiving():
    # Transciever.java:80-84, 106-109
    s = O.Sim(2)
    s.transmission_events(0, 1, 0, 1000)     # node 1 sends 0..1000
    s.reception_events(1, 1, 500, 100, -7.0, False)   # and hears a frame 500..600
    s.step(501)
    assert s.receiving_state(1) == 2 and s.rssi(1) == -7.0
    s.step(601)
    assert s.receiving_state(1) == 0          # reception over, and the start flank had cleared "sending"
    s.step(2000)
    assert s.receiving_state(1) == 0
    s.close()


def test_overlapping_receptions_are_both_delivered():
    # no collision rule in radio-medium/ -- the later start overwrites, both ends deliver
    s = O.Sim(3)
    s.reception_events(0, 2, 0, 1000, -1.0, True)
    s.reception_events(1, 2, 500, 1000, -2.0, True)
    ev = s.step(10_000)
    assert [(int(e["pkt"]), int(e["kind"])) for e in ev] == [(0, 0), (1, 0), (0, 2), (1, 2)]
    s.close()


def test_move_top_boundary_later_equal_time_event_pops_after():
    # EventQueue.moveTop sets topStart = maxTS (:329-337).  An event added later with time == maxTS goes to the
    # top list (time >= topStart, :83) and pops AFTER the equal-time events of the ladder -- insertion order there,
    # not the reverse
    got = O.evq_replay([("add", 10), ("add", 50), ("add", 50), ("pop", 20),   # moveTop: topStart = 50; 10 popped
                        ("add", 50), ("add", 50), ("add", 30), ("pop", 100)])
    assert list(got) == [0, 5, 2, 1, 4, 3]


def test_spawned_rungs_keep_the_order():
    # more than SPAWN_THRESHOLD events in one bucket of width > 1: a finer rung is spawned (:258-265); order stays
    # time-sorted with equal times reversed
    times = [1000 + (i * 7) % 40 for i in range(200)] + [5000]
    got = O.evq_replay([("add", t) for t in times] + [("pop", 10_000)])
    want = sorted(range(len(times)), key=lambda i: (times[i], -i))
    assert list(got) == want


def closed_form_order(ops):
    """pop order of every ('pop', T) of `ops` by the closed form: sort by (time, ladder, -insertion)."""
    L, S = 0, 0              # moveTop calls so far, EventQueue.topStart
    top_nonempty, top_max = False, 0
    pending = []             # (time, ladder, -id)
    out = []
    nid = 0
    for kind, t in ops:
        if kind == "add":
            if t >= S:       # waits in the top list for the next moveTop
                pending.append((t, L + 1, -nid))
                top_max = t if not top_nonempty else max(top_max, t)
                top_nonempty = True
            else:
                pending.append((t, L, -nid))
            nid += 1
        else:
            # the drain empties the current ladder iff its largest time (S, the maxTS it was built with) is below T --
            # or there is none yet; the next peek then builds a new ladder from the top list (EventQueue.java:161-167)
            if top_nonempty and (L == 0 or S < t):
                L += 1
                S = top_max
                top_nonempty = False
            pending.sort()
            k = 0
            while k < len(pending) and pending[k][0] < t:
                out.append(-pending[k][2])
                k += 1
            pending = pending[k:]
    return out


@pytest.mark.parametrize("seed", range(200))
def test_closed_form_order_equals_the_literal_queue(seed):
    rng = np.random.default_rng(seed)
    now = 0
    ops = []
    style = seed % 4
    for tick in range(int(rng.integers(3, 30))):
        n_add = int(rng.integers(0, 120 if style != 3 else 12))
        for _ in range(n_add):
            if style == 0:      # frames of a tick: starts now, ends now + air (few distinct values -> many ties)
                t = now + int(rng.choice([0, 0, 320, 320, 8128, 1000, 2000]))
            elif style == 1:    # spread starts
                t = now + int(rng.integers(0, 1000)) + int(rng.choice([0, 4064, 8128]))
            elif style == 2:    # everything at few instants, far horizon
                t = now + int(rng.choice([0, 1000, 1000, 3000, 9000, 9000]))
            else:
                t = now + int(rng.integers(0, 5)) * 1000
            ops.append(("add", t))
        now += int(rng.choice([1000, 1000, 1000, 1, 10, 5000]))
        ops.append(("pop", now))
    ops.append(("pop", now + 10 ** 9))
    got = list(O.evq_replay(ops))
    assert got == closed_form_order(ops)
