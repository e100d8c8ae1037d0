// loop_latency.cpp -- the PCIe-inclusive closed loop through the C ABI, no interpreter in the loop: what one simulated tick
// costs a host that (A) hands the tick's records in and reads the heard links in place (rm_tick_begin / rm_enqueue_tx_records /
// rm_tick_flush_view), or (B) keeps packets, links and events on the device and takes the drain's deliveries
// (rm_tick_run_sources_device + rm_events_process).  Shape of BASELINE configs[2]: 100 k nodes, 1000 frames per tick,
// log-distance + shadowing, frames of 8128 us over 1000 us ticks.
//
//   g++ -std=c++17 -O2 tools/loop_latency.cpp -Iinclude -Lradio-sim_amd/csrc -lradiomedium_hip -Wl,-rpath,$PWD/radio-sim_amd/csrc -o tools/loop_latency
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "radiomedium_hip.h"

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { if ((x) != RM_OK) { std::fprintf(stderr, "%s: %s\n", #x, rm_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 100000, t = argc > 2 ? std::atoi(argv[2]) : 1000, ticks = argc > 3 ? std::atoi(argv[3]) : 300;
    const double side = 50.0 * std::sqrt(3.14159265358979323846 * n / 20.0);
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> u(0.0, side);
    std::vector<double> x(n), y(n);
    for (int i = 0; i < n; ++i) { x[i] = u(rng); y[i] = u(rng); }
    rm_context *c = nullptr;
    CK(rm_create(0, &c));
    CK(rm_nodes_upload(c, n, x.data(), y.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
    rm_model_params p;
    rm_model_defaults(&p, RM_MODEL_LOGDIST);
    p.ld_sigma_db = 4.0;
    p.ld_seed = 0xC0FFEE;
    CK(rm_set_model(c, &p));
    CK(rm_set_link_capacity(c, 1u << 21));
    const int pool = 16;
    std::vector<std::vector<rm_tx_record>> recs(pool);
    std::vector<int32_t *> dev_src(pool);
    std::uniform_int_distribution<int> pick(0, n - 1);
    for (int k = 0; k < pool; ++k) {
        std::vector<int32_t> src(t);
        recs[k].resize(t);
        for (int f = 0; f < t; ++f) {
            const int s = pick(rng);
            src[f] = s;
            recs[k][f] = rm_tx_record{x[s], y[s], 0.0, 0.0, 1.0, 0, 8128, s, 26};
        }
        if (hipMalloc(reinterpret_cast<void **>(&dev_src[k]), t * 4) != hipSuccess) return 1;
        if (hipMemcpy(dev_src[k], src.data(), t * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
    }
    // (A) records in, heard links read in place
    double t0 = 0;
    uint64_t links = 0;
    for (int k = 0; k < ticks + 20; ++k) {
        if (k == 20) { t0 = now_us(); links = 0; }
        const int64_t tb = 1000LL * k;
        for (auto &r : recs[k % pool]) r.start_us = tb;
        CK(rm_tick_begin(c, tb, tb + 1000));
        CK(rm_enqueue_tx_records(c, recs[k % pool].data(), t));
        rm_host_result r;
        CK(rm_tick_flush_view(c, &r));
        links += r.count;
    }
    const double a_us = (now_us() - t0) / ticks;
    std::printf("{\"mode\": \"tick_flush_view\", \"nodes\": %d, \"frames_per_tick\": %d, \"us_per_tick\": %.1f, \"heard_links_per_tick\": %.0f, "
                "\"links_per_s\": %.3e}\n", n, t, a_us, double(links) / ticks, double(t) * (n - 1) / (a_us * 1e-6));
    // (B) device events: only the deliveries come back
    CK(rm_events_enable(c, 1u << 16, 1u << 21));
    CK(rm_set_time(c, 1000LL * (ticks + 20)));
    uint64_t deliveries = 0;
    double tick_call = 0, drain_call = 0;
    const int64_t base = 1000LL * (ticks + 20);
    for (int k = 0; k < ticks + 30; ++k) {
        if (k == 30) { t0 = now_us(); deliveries = 0; tick_call = drain_call = 0; }
        const int64_t tb = base + 1000LL * k;
        const double a = now_us();
        CK(rm_tick_run_sources_device(c, tb, tb + 1000, dev_src[k % pool], t, tb, 8128));
        const double b = now_us();
        rm_delivery_view v;
        CK(rm_events_process(c, tb + 1000, &v));
        drain_call += now_us() - b;
        tick_call += b - a;
        deliveries += v.count;
    }
    const double b_us = (now_us() - t0) / ticks;
    std::printf("{\"mode\": \"tick_events\", \"nodes\": %d, \"frames_per_tick\": %d, \"us_per_tick\": %.1f, \"deliveries_per_tick\": %.0f, "
                "\"tick_call_us\": %.1f, \"drain_call_us\": %.1f, \"links_per_s\": %.3e}\n", n, t, b_us, double(deliveries) / ticks,
                tick_call / ticks, drain_call / ticks, double(t) * (n - 1) / (b_us * 1e-6));
    rm_destroy(c);
    return 0;
}
