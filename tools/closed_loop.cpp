// closed_loop.cpp -- what a host written against the reference's API sees per simulated tick, without an interpreter in
// the loop: the C++ mirror of the plug-in API (radio-sim_amd/host/radiomedium.hpp) driven like the reference drives a
// medium -- T RadioMedium.transmit calls, then Simulator.emulatorTimeStepDone -- on the shape of BASELINE configs[2]
// (100 k nodes, 1000 frames per tick, log-distance + shadowing) or a smaller one.
//
//   g++ -std=c++17 -O2 tools/closed_loop.cpp -Lradio-sim_amd/csrc -lradiomedium_hip -Wl,-rpath,$PWD/radio-sim_amd/csrc -o closed_loop
//   ./closed_loop [nodes] [frames per tick] [ticks]
//
// Modes: per packet (one rm_transmit per call: the reference's own pattern), tick (transmit queues, ONE evaluation per
// tick, the heard links read in place and turned into generate*Events calls), tick + device events (only deliveries
// come back).  PCIe-inclusive by nature; never bench.py's `value`.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../radio-sim_amd/host/radiomedium.hpp"

using namespace emul8;

struct LogDistMedium : GpuRadioMedium { // the extension medium of DESIGN.md section 6 behind the same contract
    explicit LogDistMedium(double sigma, long long seed) : GpuRadioMedium(RM_MODEL_LOGDIST)
    {
        params_.ld_sigma_db = sigma;
        params_.ld_seed = uint64_t(seed);
        apply();
    }
};

int main(int argc, char **argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 100000;
    const int t = argc > 2 ? std::atoi(argv[2]) : 1000;
    const int ticks = argc > 3 ? std::atoi(argv[3]) : 200;
    const double side = 50.0 * std::sqrt(3.14159265358979323846 * n / 20.0);
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> u(0.0, side);
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 0 && t > 200) continue; // a thousand 25-us calls per tick: the small shape shows it
        Simulator sim(1);
        sim.recording = false; // count the medium's calls, do not keep them: what is timed is the medium, not the stub's list
        for (int i = 0; i < n; ++i) {
            Node *nd = sim.addNode(std::to_string(i + 1));
            nd->getPosition().set(u(rng), u(rng));
        }
        LogDistMedium medium(4.0, 0xC0FFEE);
        sim.setRadioMedium(&medium);
        medium.setTickMode(mode >= 1);
        if (mode == 2) medium.setDeviceEvents(true);
        std::vector<std::unique_ptr<RadioPacket>> packets;
        std::uniform_int_distribution<int> pick(0, n - 1);
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < ticks + 20; ++k) {
            if (k == 20) {
                sim.callCount = 0;
                t0 = std::chrono::steady_clock::now();
            }
            const long long now = 1000LL * k;
            for (int f = 0; f < t; ++f) {
                packets.emplace_back(new RadioPacket(sim.getNodes()[size_t(pick(rng))], now, "0102030405060708090a0b0c0d0e0f10"));
                medium.transmit(*packets.back());
            }
            sim.emulatorTimeStepDone(now + 1000);
            if (!medium.lastError.empty()) {
                std::fprintf(stderr, "medium: %s\n", medium.lastError.c_str());
                return 1;
            }
            if (packets.size() >= size_t(40 * t)) packets.erase(packets.begin(), packets.begin() + 20 * t); // (frames of 1 ms: long gone)
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / ticks;
        static const char *names[3] = {"per packet (rm_transmit)", "tick mode (one evaluation per tick, generate*Events on the host)",
                                       "tick mode + device events (deliveries only)"};
        std::printf("{\"nodes\": %d, \"frames_per_tick\": %d, \"mode\": \"%s\", \"us_per_tick\": %.1f, \"simulator_calls_per_tick\": %.0f, "
                    "\"links_per_s\": %.3e}\n",
                    n, t, names[mode], us, double(sim.callCount) / ticks, double(t) * (n - 1) / (us * 1e-6));
    }
    return 0;
}
