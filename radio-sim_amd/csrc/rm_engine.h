// rm_engine.h -- internal interface between the C-ABI host code (rm_api.cpp) and the
// gfx950 kernels (rm_kernels.hip).  Not part of the public boundary (include/radiomedium_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/radiomedium_hip.h"

namespace rm {

constexpr int kTxChunk = 64;        // transmitters per LDS tile == wave width (one count lane per tx)
constexpr int kWavesPerBlock = 4;   // 256-thread workgroups
constexpr int kBlock = 64 * kWavesPerBlock;

// staging-entry flags
constexpr uint8_t kFlagHeardNew = 1;   // gets an output record (has a slab rank)
constexpr uint8_t kFlagInterferer = 2; // rssi >= interference floor (SINR mode)
constexpr uint8_t kFlagSelf = 4;       // receiver is the source of this on-air frame (half duplex)

// model constants as the kernels need them
struct ModelDev {
    int kind;
    int flags;
    double udgm_ratio_rx;     // successRatioRx
    double udgm_range;        // transmissionRange
    double const_range;
    const double *n2n;        // row-major m x m (device)
    int n2n_m;
    double ld_pl0, ld_exp, ld_d0, ld_sigma, ld_clip;
    uint64_t ld_seed_mixed;   // mix64(seed + golden)
    double ld_sens, ld_noise, ld_capture, ld_ifloor;
    double ld_noise_lin;      // det_pow10(noise/10)
    double ld_level;          // candidate level L = sens, or min(sens, ifloor) with SINR
    // prefilter
    double org_x, org_y, org_z; // origin of the fp32 frame
    double coord_bound;         // max |coord - origin| the fp32 slack was computed for
    double f32_slack;           // Delta (metres) added to every cut-off distance
    double geo_cut;             // cut-off distance for UDGM / CONST (metres), <0 = nobody
};

struct NodesDev {
    int n;
    const double *x, *y, *z, *rxprob, *txprob, *txpower;
    const int32_t *channel;
    const uint8_t *enabled;
    const int32_t *int_id;
    float4 *rxf; // prefilter record per node: (fx, fy, fz, channel bits); NaN position = never a candidate
};

struct TickDev {
    const rm_tx_record *tx; // on-air list, canonical order [n_active]
    float4 *txf;            // prefilter record per frame: (fx, fy, fz, threshold on d^2)
    double *txd;            // fp64 filter: thr64 per frame
    int n_active;
    int first_new;          // frames [first_new, n_active) get verdicts
    int first_eval;         // frames [first_eval, n_active) are swept by the all-pairs kernel
    int cnt_base;           // eval-relative index of the first counted slot (multiple of 64)
    int shift;              // slot of new packet 0 = shift
    int n_cnt;              // counted slots (multiple of 64)
    int rx_first, rx_count; // receiver partition
    int rpt;                // receivers per thread in the all-pairs kernel
    int n_slabs;            // ceil(rx_count / (64*rpt))
    // per (slot, slab) heard counts / offsets, layout [(chunk*n_slabs + slab)*64 + lane]
    uint32_t *cnt, *off;
    uint32_t *partial;      // [n_cnt * n_groups]
    int n_groups, slabs_per_group;
    uint32_t *slot_off;     // [n_cnt + 1] exclusive scan of per-slot totals
    // staging (unordered, filled by the all-pairs kernel)
    uint32_t *stage_count;  // [0] entries appended, [1] dropped-for-capacity flag
    uint32_t cap;
    int32_t *st_pkt;        // eval-relative frame index
    int32_t *st_dst;
    uint32_t *st_rank;
    double *st_aux;         // probability (UDGM/N2N) or rssi (logdist)
    double *st_lin;         // linear power (SINR)
    double *st_sinr;
    int32_t *st_next;       // per-receiver list (SINR)
    uint8_t *st_flags;
    uint8_t *st_coll;
    int32_t *head;          // [rx_count]
    // final, ordered records
    uint32_t *out_count;    // [0] heard links stored, [1] dropped flag, [2] heard links total
    int32_t *out_pkt, *out_dst;
    uint8_t *out_verdict;
    double *out_rssi, *out_sinr, *out_prob;
    uint8_t *pkt_interference; // [n_new]
    // stochastic part
    uint32_t *draw_scan;    // [cap+1]
    uint32_t *scan_block;   // scratch for the generic scan
    uint64_t *rng_state;    // [1] java.util.Random state (48 bit)
    uint64_t *pkt_rng;      // [n_new] state before the packet's receiver draws
};

struct LaunchCfg {
    bool f64_filter;
    bool stochastic;
};

// kernels' host launchers (rm_kernels.hip)
hipError_t launch_prep_rx(hipStream_t s, const NodesDev &nd, const ModelDev &m);
hipError_t launch_prep_tx(hipStream_t s, const ModelDev &m, const TickDev &t);
hipError_t launch_pack_tx(hipStream_t s, const NodesDev &nd, const int32_t *dev_src, int n, int64_t start_us,
                          int64_t air_us, rm_tx_record *out);
hipError_t launch_allpairs(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t,
                           const LaunchCfg &cfg);
hipError_t launch_self_entries(hipStream_t s, const TickDev &t);
hipError_t launch_offsets(hipStream_t s, const TickDev &t);
hipError_t launch_sinr(hipStream_t s, const ModelDev &m, const TickDev &t);
hipError_t launch_finalize(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t,
                           const LaunchCfg &cfg);
hipError_t launch_draws(hipStream_t s, const ModelDev &m, const TickDev &t);

// host-side mirrors of device math used for constants (rm_kernels.hip, __host__ __device__)
double host_det_pow10(double y);
uint64_t host_mix64(uint64_t z);
void host_lcg_jump_map(uint64_t steps, uint64_t *A, uint64_t *C);

} // namespace rm
