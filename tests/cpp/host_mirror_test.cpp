// host_mirror_test.cpp -- drives the C++ mirror of the reference's RadioMedium API
// (radio-sim_amd/host/radiomedium.hpp) the way a reference-side test would: build a Simulator,
// add nodes, install a medium, transmit packets; prints every Simulator call the medium made as
// one line "kind packet dst rssi doDeliver t0 t1" for tests/test_gpu_host_mirror.py to compare
// with the oracle.  Input: a scenario file (see the Python test).
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../radio-sim_amd/host/radiomedium.hpp"

using namespace emul8;

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
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
    std::unique_ptr<RadioMedium> medium;
    // "udgm@2": the same medium as a group of 2 members on device 0 (receivers split over two contexts)
    int members = 0;
    if (const size_t at = model.find('@'); at != std::string::npos) {
        members = std::atoi(model.c_str() + at + 1);
        model.resize(at);
    }
    try {
        if (members > 0) {
            const std::vector<int32_t> devices(size_t(members), 0);
            if (model == "udgm") {
                double ratioRx, range;
                in >> ratioRx >> range;
                auto *m = new GroupRadioMedium(RM_MODEL_UDGM, devices);
                medium.reset(m);
                m->params().udgm_success_ratio_rx = ratioRx;
                m->params().udgm_transmission_range = range;
                m->params().udgm_success_ratio_tx = 0.0;
                m->apply();
            } else if (model == "const") {
                medium.reset(new GroupRadioMedium(RM_MODEL_UDGM_CONST, devices));
            } else if (model == "null") {
                medium.reset(new GroupRadioMedium(RM_MODEL_NULL, devices));
            } else if (model == "n2n") {
                int m;
                in >> m;
                std::vector<std::vector<double>> mat(m, std::vector<double>(m));
                for (auto &row : mat)
                    for (auto &v : row) in >> v;
                auto *g = new GroupRadioMedium(RM_MODEL_N2N, devices);
                medium.reset(g);
                g->setMatrix(mat);
            } else {
                return 2;
            }
        } else if (model == "udgm") {
            double ratioRx, range;
            in >> ratioRx >> range;
            auto *m = new UDGMRadioMedium();
            medium.reset(m);
            m->setSuccessRatioRx(ratioRx);
            m->setTransmissionRange(range);
            m->setSuccessRatioTx(0.0); // dead field in the reference: must change nothing (K8)
        } else if (model == "const") {
            medium.reset(new UDGMConstantLossRadioMedium());
        } else if (model == "null") {
            medium.reset(new NullRadioMedium());
        } else if (model == "n2n") {
            int m;
            in >> m;
            std::vector<std::vector<double>> mat(m, std::vector<double>(m));
            for (auto &row : mat)
                for (auto &v : row) in >> v;
            medium.reset(new N2NRadioMedium(mat));
        } else {
            return 2;
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "medium: %s\n", e.what());
        return 3;
    }
    sim.setRadioMedium(medium.get());
    std::printf("name %s\n", medium->getName().c_str());
    std::printf("base %.17g %.17g\n", medium->getBaseRSSI(*sim.getNodes()[0]), sim.getNodes()[0]->getRadio().getRSSI());
    int np;
    in >> np;
    std::vector<std::unique_ptr<RadioPacket>> packets;
    for (int p = 0; p < np; ++p) {
        std::string id, hex;
        long long start, now;
        int has_override;
        in >> id;
        if (id == "@move") { // node-config-set with a new position between two packets
            std::string who;
            double x, y, z;
            in >> who >> x >> y >> z;
            Node *nd = sim.getNode(who);
            nd->getPosition().set(x, y, z);
            sim.nodeChanged(nd);
            packets.emplace_back(nullptr);
            continue;
        }
        in >> start >> now >> hex >> has_override;
        Node *src = sim.getNode(id);
        if (!src) { std::printf("error could not find source node\n"); continue; } // SimulatorJSONHandler.java:75-77
        packets.emplace_back(new RadioPacket(src, start, hex == "-" ? std::string() : hex));
        if (has_override) {
            double tp;
            int ch;
            in >> tp >> ch;
            packets.back()->setTransmitPower(tp);
            packets.back()->setWirelessChannel(ch);
        }
        sim.setTime(now);
        const size_t before = sim.calls.size();
        medium->transmit(*packets.back());
        if (auto *g = dynamic_cast<GpuRadioMedium *>(medium.get())) {
            if (!g->lastError.empty()) std::printf("error %s\n", g->lastError.c_str());
        } else if (auto *gg = dynamic_cast<GroupRadioMedium *>(medium.get())) {
            if (!gg->lastError.empty()) std::printf("error %s\n", gg->lastError.c_str());
        }
        for (size_t i = before; i < sim.calls.size(); ++i) {
            const MediumCall &c = sim.calls[i];
            std::printf("call %d %d %d %.17g %d %lld %lld\n", int(c.kind), p, c.destination ? c.destination->index : -1, c.rssi,
                        int(c.doDeliver), (long long)c.timeStart, (long long)c.timeEnd);
        }
    }
    return 0;
}
