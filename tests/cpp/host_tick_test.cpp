// host_tick_test.cpp -- the C++ mirror (radio-sim_amd/host/radiomedium.hpp) driven tick by tick in its
// three modes: "packet" (one rm_transmit per RadioMedium.transmit), "tick" (transmit() queues, ONE
// evaluation in Simulator::emulatorTimeStepDone before the time moves) and "device" (tick mode with the
// events kept on the device: deliveries and node-info come back).  Prints every Simulator call the medium
// made ("call kind packet dst rssi doDeliver t0 t1"), after every tick the node-info of all nodes
// ("info tick node rssi state channel", device mode) -- tests/test_gpu_host_tick.py compares the modes
// with each other and with the oracle.
// Input: host_mirror_test's scenario header (model, seed, nodes, model parameters), then
//   <ticks> ; per tick: <stepTime> <packets> ; per packet: <id> <start> <hex|-> <has_override> [txpower channel]
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "../../radio-sim_amd/host/radiomedium.hpp"

using namespace emul8;

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const std::string mode = argv[2];
    std::ifstream in(argv[1]);
    std::string model;
    long long seed;
    int n;
    in >> model >> seed >> n;
    Simulator sim(seed);
    for (int i = 0; i < n; ++i) {
        std::string id;
        double x, y, z, tp, rp, xp;
        int ch, en;
        in >> id >> x >> y >> z >> tp >> ch >> en >> rp >> xp;
        Node *nd = sim.addNode(id);
        nd->getPosition().set(x, y, z);
        nd->getRadio().setTransmitPower(tp);
        nd->getRadio().setWirelessChannel(ch);
        nd->getRadio().setEnabled(en != 0);
        nd->getRadio().setRxProbability(rp);
        nd->getRadio().setTxProbability(xp);
    }
    std::unique_ptr<GpuRadioMedium> medium;
    try {
        if (model == "udgm") {
            double ratioRx, range;
            in >> ratioRx >> range;
            auto *m = new UDGMRadioMedium();
            medium.reset(m);
            m->setSuccessRatioRx(ratioRx);
            m->setTransmissionRange(range);
        } else if (model == "const") {
            medium.reset(new UDGMConstantLossRadioMedium());
        } else if (model == "null") {
            medium.reset(new NullRadioMedium());
        } else {
            return 2;
        }
        medium->setTickMode(mode != "packet");
        medium->setDeviceEvents(mode == "device");
    } catch (const std::exception &e) {
        std::fprintf(stderr, "medium: %s\n", e.what());
        return 3;
    }
    sim.setRadioMedium(medium.get());
    int ticks;
    in >> ticks;
    std::vector<std::unique_ptr<RadioPacket>> packets;
    std::vector<int32_t> all(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) all[size_t(i)] = i;
    size_t printed = 0;
    for (int t = 0; t < ticks; ++t) {
        long long stepTime;
        int np;
        in >> stepTime >> np;
        for (int p = 0; p < np; ++p) {
            std::string id, hex;
            long long start;
            int has_override;
            in >> id >> start >> hex >> has_override;
            packets.emplace_back(new RadioPacket(sim.getNode(id), start, hex == "-" ? std::string() : hex));
            if (has_override) {
                double tp;
                int ch;
                in >> tp >> ch;
                packets.back()->setTransmitPower(tp);
                packets.back()->setWirelessChannel(ch);
            }
            medium->transmit(*packets.back());
            if (!medium->lastError.empty()) std::printf("error %s\n", medium->lastError.c_str());
        }
        sim.emulatorTimeStepDone(stepTime);
        if (!medium->lastError.empty()) std::printf("error %s\n", medium->lastError.c_str());
        for (; printed < sim.calls.size(); ++printed) {
            const MediumCall &c = sim.calls[printed];
            size_t pid = 0;
            while (pid < packets.size() && packets[pid].get() != c.packet) ++pid;
            std::printf("call %d %zu %d %.17g %d %lld %lld\n", int(c.kind), pid, c.destination ? c.destination->index : -1, c.rssi,
                        int(c.doDeliver), (long long)c.timeStart, (long long)c.timeEnd);
        }
        std::printf("tick %d %lld\n", t, stepTime);
        if (mode == "device") {
            std::vector<double> rssi;
            std::vector<int32_t> rx, ch;
            if (!medium->nodeInfo(all, rssi, rx, ch)) { std::printf("error %s\n", medium->lastError.c_str()); continue; }
            for (int i = 0; i < n; ++i)
                if (rx[size_t(i)] != 0 || rssi[size_t(i)] != -100.0)
                    std::printf("info %d %d %.17g %d %d\n", t, i, rssi[size_t(i)], rx[size_t(i)], ch[size_t(i)]);
        }
    }
    return 0;
}
