// host_trace_test.cpp -- packet traces of the C++ mirror (no GPU needed): the pcap bytes of
// util/PcapExporter.java:47-91 / util/PcapListener.java:40-58 and the compact replay trace.
// Usage: host_trace_test <pcap-out> <trace-out>; prints "ok" or the first failed expectation.
#include <cstdio>

#include "../../radio-sim_amd/host/radiomedium.hpp"

using namespace emul8;

#define EXPECT(cond)                                                                      \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            std::printf("FAILED line %d: %s\n", __LINE__, #cond);                         \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    Simulator sim(1);
    Node *a = sim.addNode("1"), *b = sim.addNode("2");
    b->getRadio().setTransmitPower(-3.5);
    b->getRadio().setWirelessChannel(15);
    {
        PcapListener pcap(argv[1]);
        TraceListener trace(argv[2]);
        sim.addRadioListener(&pcap);
        sim.addRadioListener(&trace);
        RadioPacket p1(a, 1000, "0102030405");            // 5 bytes at t = 1000 us
        RadioPacket p2(b, 3000123, "FFfe7f");             // 3 bytes at t = 3 s + 123 us, mixed-case hex
        RadioPacket p3(a, 4294967296000000LL, "");        // seconds beyond 32 bits are truncated as (int) does
        EXPECT(p1.getPacketDataAsBytes().size() == 5 && p1.getPacketDataAsBytes()[4] == 5);
        EXPECT(p2.getPacketDataAsBytes()[0] == 0xFF && p2.getPacketDataAsBytes()[1] == 0xFE && p2.getPacketDataAsBytes()[2] == 0x7F);
        sim.notifyRadioListeners(p1);                     // what SimulatorJSONHandler.java:92 does after transmit
        sim.notifyRadioListeners(p2);
        sim.notifyRadioListeners(p3);
        bool threw = false;
        try {
            RadioPacket bad(a, 0, "abc");
            bad.getPacketDataAsBytes();
        } catch (const std::invalid_argument &) {
            threw = true;                                 // parseHexBinary: odd length
        }
        EXPECT(threw);
    }
    std::printf("ok\n");
    return 0;
}
