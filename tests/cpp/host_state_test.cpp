// host_state_test.cpp -- the receiver-side state machine of the C++ mirror (no GPU needed):
// Transciever.java:52-113 + events/ReceptionEvent.java:35-46 + events/TransmissionEvent.java:18-26.
// Prints "ok" or the first failed expectation.
#include <cstdio>

#include "../../radio-sim_amd/host/radiomedium.hpp"

using namespace emul8;

struct FixedMedium : AbstractRadioMedium {
    std::string getName() override { return "fixed"; }
    void transmit(RadioPacket &) override {}
};

#define EXPECT(cond)                                                                      \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            std::printf("FAILED line %d: %s\n", __LINE__, #cond);                         \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

int main()
{
    Simulator sim(1);
    Node *a = sim.addNode("1"), *b = sim.addNode("2"), *c = sim.addNode("node-c");
    EXPECT(sim.addNode("1") == a && sim.getNodes().size() == 3);      // Simulator.java:249-255: idempotent
    EXPECT(a->getIdAsInteger() == 1 && c->getIdAsInteger() == -1);    // Node.java:52-58
    EXPECT(b->getRadio().getRSSI() == -99.99);                        // no medium yet, Transciever.java:60
    FixedMedium medium;
    sim.setRadioMedium(&medium);
    EXPECT(b->getRadio().getRSSI() == -100.0);                        // base RSSI, AbstractRadioMedium.java:38
    medium.setBaseRSSI(-91.5);
    EXPECT(b->getRadio().getRSSI() == -91.5);
    EXPECT(b->getRadio().getReceivingState() == Transciever::LISTENING);
    EXPECT(b->getRadio().getWirelessChannel() == 26 && b->getRadio().getTransmitPower() == 0.0);

    RadioPacket p(a, 1000, "0102030405");
    EXPECT(p.getPacketAirTime() == 320 && p.getEndTime() == 1320);    // 32 us per hex character
    sim.setTime(5000);
    sim.generateReceptionEvents(p, b, -55.0, true);
    EXPECT(sim.calls.back().timeStart == 5000 && sim.calls.back().timeEnd == 5320);   // max(start, currentTime)

    ReceptionEvent s{}, e{};
    makeReceptionEvents(sim, sim.calls.back(), p, s, e);
    TransmissionEvent ts{5000, &p, true}, te{5320, &p, false};
    ts.execute(5000);
    EXPECT(a->getRadio().getReceivingState() == Transciever::TRANSMITTING);
    s.execute(5000);
    EXPECT(b->getRadio().isReceiving() && b->getRadio().getRSSI() == -55.0);          // latched rssi
    EXPECT(b->getRadio().getReceivingState() == Transciever::RECEIVING);
    const size_t before = sim.calls.size();
    e.execute(5320);
    EXPECT(!b->getRadio().isReceiving() && b->getRadio().getRSSI() == -91.5);
    EXPECT(sim.calls.size() == before + 1 && sim.calls.back().kind == MediumCall::DELIVER);   // delivery on the end flank
    te.execute(5320);
    EXPECT(a->getRadio().getReceivingState() == Transciever::LISTENING);

    // interference mode: the end flank clears, nothing is delivered
    sim.generateReceptionEvents(p, b, -60.0, false);
    makeReceptionEvents(sim, sim.calls.back(), p, s, e);
    s.execute(0);
    const size_t n0 = sim.calls.size();
    e.execute(0);
    EXPECT(sim.calls.size() == n0 && !b->getRadio().isReceiving());

    // a node that starts sending while receiving drops the reception (setSending -> clearReceiving)
    RadioPacket q(b, 9000, "00");
    s.execute(0);
    TransmissionEvent qs{9000, &q, true};
    qs.execute(9000);
    EXPECT(!b->getRadio().isReceiving() && b->getRadio().getReceivingState() == Transciever::TRANSMITTING);
    // ... and a reception start clears a pending transmission (setReceiving -> clearSending)
    s.execute(0);
    EXPECT(b->getRadio().getReceivingState() == Transciever::RECEIVING);

    // zero-length payload: start == end; executed end-before-start (the reference's equal-timestamp
    // order) the receiver is left in RECEIVING
    RadioPacket z(a, 20000, "");
    sim.setTime(0);
    sim.generateReceptionEvents(z, c, -70.0, true);
    EXPECT(sim.calls.back().timeStart == sim.calls.back().timeEnd);
    makeReceptionEvents(sim, sim.calls.back(), z, s, e);
    e.execute(20000);
    s.execute(20000);
    EXPECT(c->getRadio().getReceivingState() == Transciever::RECEIVING);
    c->getRadio().setEnabled(false);
    EXPECT(c->getRadio().getReceivingState() == Transciever::DISABLED);
    std::printf("ok\n");
    return 0;
}
