// rm_tick.hip -- the closed-loop tick: ONE frame per workgroup, filter + exact evaluation in one launch
// (part of libradiomedium_hip.so; gfx950 only, -ffp-contract=off, no fast-math; overview at the top of rm_engine.h)
//
// A lone tick is what the reference's call pattern produces: the emulators step, their transmit()
// calls are evaluated, the events are consumed in emulatorTimeStepDone (Simulator.java:155-165), and
// only then does the next tick begin.  Its few MB of work cannot fill the chip, so what it costs is
// the chain of dependent launches and memory round trips.  The sweep kernels (rm_filter / rm_exact)
// need three dependent launches because the candidate list crosses workgroups; here a frame's whole
// evaluation stays inside one workgroup, as k_transmit_one (rm_transmit.hip) does for one packet:
//
//   level 1   the frame against the boxes of 1024 receivers (one box per thread)        -> LDS list
//   level 2   the frame against the group boxes (64 receivers) of the near ones          -> LDS list
//   level 3   the receivers of the near groups: fp32 pre-filter, exact channel test      -> LDS candidates
//   exact     the candidates with full lanes: the reference's fp64 arithmetic (eval_link)
//             UDGMRadioMedium.java:99-111, UDGMConstantLossRadioMedium.java:25-33          -> LDS links
//   write     the frame's heard links into its own fixed segment of the A records
//
// k_reorder (rm_reorder.hip) then ranks every frame's links by node index and compacts them: two
// launches per tick instead of three, no candidate list, no atomics on global memory.  Every level
// works in rounds, so nothing is bounded by the LDS lists: a frame with more heard links than its
// segment holds takes its room from an overflow allocator and evaluates its candidates a second time.
#include "rm_device.hpp"

namespace rm {

constexpr int kFrBoxes = 1024; // level-1 boxes tested per round (4 per thread)
constexpr int kFrGroups = 1024; // level-2 group boxes tested per round
constexpr int kFrRound = 16;   // near groups per level-3 round: at most 1024 candidates
constexpr int kFrCand = kFrRound * kGroup;

template <int MODEL, bool STOCH>
__global__ void __launch_bounds__(256) k_tick_frames(const NodesDev nd, const ModelDev m, const TickDev t, const int seg_len)
{
    __shared__ int s_l1[kFrBoxes], s_l2[kFrGroups], s_cand[kFrCand];
    __shared__ int s_orig[kFrameSegMax];
    __shared__ double s_rssi[kFrameSegMax];
    __shared__ double s_prob[STOCH ? kFrameSegMax : 1];
    __shared__ uint32_t s_n1[2], s_n2[2], s_nc[2], s_nres, s_base; // the lists' fill counts, by round parity

    const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
    const int slot = blockIdx.x;
    const int n_new = t.n_active - t.first_new;

    // what the sweep's first kernel does for the tick after this one (rm_filter.hip, tick_prep_body)
    if (blockIdx.x == 0) {
        if (tid < 8) t.next_counters[tid] = 0u;
        t.next_shard_count[tid * kShardStride] = 0u; // kBlock == kShards
    }
    for (int i = blockIdx.x * blockDim.x + tid; i < t.zero_len; i += gridDim.x * blockDim.x) t.cand_tot_next[i] = 0u;

    const int q = slot - t.shift;
    if (q < 0 || q >= n_new) { // padding slot of the per-frame counters
        if (tid == 0) {
            t.cursor[slot] = 0u;
            t.seg_off[slot] = uint32_t(slot) * uint32_t(seg_len);
        }
        return;
    }
    rm_tx_record tx;
    const int abs_i = t.first_new + q;
    if (t.src_list) { // RadioPacket(node, time, data) copies txpower / channel from its source, RadioPacket.java:46-52
        tx = make_tx_record(nd, t.src_list[q], t.src_start_us, t.src_air_us);
        if (tid == 0) t.tx_build[abs_i] = tx;
    } else {
        tx = t.tx[abs_i];
    }
    float4 f;
    double thr64;
    tx_prefilter(m, tx, f, thr64);
    if (tid == 0) s_n1[0] = s_n1[1] = s_n2[0] = s_n2[1] = s_nc[0] = s_nc[1] = s_nres = s_base = 0u;
    __syncthreads();
    // A list's fill count of one round is cleared while the next round's -- the other parity -- is in use: every
    // reuse of a list or a count is separated from its last reader by a barrier without extra barriers for the clearing.
    int r1 = 0, r2 = 0, rc = 0;

    const int n_groups = (nd.n_rx + kGroup - 1) / kGroup;
    const int n_boxes = (n_groups + 15) / 16;
    const bool dead = (MODEL != RM_MODEL_UDGM_CONST) && tx_success(m, tx) <= 0.0; // UDGMRadioMedium.java:88
    const uint32_t seg = uint32_t(seg_len);
    const uint32_t fixed_base = uint32_t(slot) * seg;

    // pass 0 keeps the links in LDS; pass 1 only runs for a frame that heard more than its segment holds
    // and writes them straight to the room it got from the overflow allocator
    for (int pass = 0; pass < 2; ++pass) { // block-uniform
        if (f.w >= 0.f) {
            for (int b0 = 0; b0 < n_boxes; b0 += kFrBoxes) {
                // level 1
#pragma unroll
                for (int k = 0; k < kFrBoxes / 256; ++k) {
                    const int b = b0 + k * 256 + tid;
                    bool hit = false;
                    if (b < n_boxes) {
                        const float4 qb = nd.wg_box_xy[b];
                        const float2 qz = nd.wg_box_z[b];
                        const float dx = fmaxf(fmaxf(qb.x - f.x, f.x - qb.z), 0.f);
                        const float dy = fmaxf(fmaxf(qb.y - f.y, f.y - qb.w), 0.f);
                        const float dz = fmaxf(fmaxf(qz.x - f.z, f.z - qz.y), 0.f);
                        hit = dist2_f32(dx, dy, dz) <= f.w;
                    }
                    const uint64_t hm = ballot64(hit);
                    if (hm) {
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(&s_n1[r1 & 1], uint32_t(__popcll(hm)));
                        base = uniform_u(base);
                        if (hit) s_l1[base + lane_prefix(hm)] = b;
                    }
                }
                __syncthreads();
                const int n1 = uniform_i(int(s_n1[r1 & 1]));
                if (tid == 0) s_n1[(r1 + 1) & 1] = 0u;
                ++r1;
                for (int i0 = 0; i0 < n1 * 16; i0 += kFrGroups) {
                    // level 2
#pragma unroll
                    for (int k = 0; k < kFrGroups / 256; ++k) {
                        const int i = i0 + k * 256 + tid;
                        bool hit = false;
                        int g = 0;
                        if (i < n1 * 16) {
                            g = s_l1[i >> 4] * 16 + (i & 15);
                            if (g < n_groups) {
                                const float4 qb = nd.bbox_xy[g];
                                const float2 qz = nd.bbox_z[g];
                                const float dx = fmaxf(fmaxf(qb.x - f.x, f.x - qb.z), 0.f);
                                const float dy = fmaxf(fmaxf(qb.y - f.y, f.y - qb.w), 0.f);
                                const float dz = fmaxf(fmaxf(qz.x - f.z, f.z - qz.y), 0.f);
                                hit = dist2_f32(dx, dy, dz) <= f.w;
                            }
                        }
                        const uint64_t hm = ballot64(hit);
                        if (hm) {
                            uint32_t base = 0;
                            if (lane == 0) base = atomicAdd(&s_n2[r2 & 1], uint32_t(__popcll(hm)));
                            base = uniform_u(base);
                            if (hit) s_l2[base + lane_prefix(hm)] = g;
                        }
                    }
                    __syncthreads();
                    const int n2 = uniform_i(int(s_n2[r2 & 1]));
                    if (tid == 0) s_n2[(r2 + 1) & 1] = 0u;
                    ++r2;
                    for (int gi0 = 0; gi0 < n2; gi0 += kFrRound) {
                        // level 3: every wave takes four of the round's groups, their records requested together
                        float4 v[kFrRound / 4];
                        int jj[kFrRound / 4];
#pragma unroll
                        for (int k = 0; k < kFrRound / 4; ++k) {
                            const int gi = gi0 + k * 4 + wave;
                            jj[k] = -1;
                            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (gi < n2) {
                                const int j = s_l2[gi] * kGroup + lane;
                                if (j < nd.n_rx) {
                                    jj[k] = j;
                                    v[k] = nd.rxf[j];
                                }
                            }
                        }
#pragma unroll
                        for (int k = 0; k < kFrRound / 4; ++k) {
                            const float s2 = dist2_f32(v[k].x - f.x, v[k].y - f.y, v[k].z - f.z);
                            const bool hit = jj[k] >= 0 && s2 <= f.w && __float_as_int(v[k].w) == tx.channel;
                            const uint64_t hm = ballot64(hit);
                            if (hm) {
                                uint32_t base = 0;
                                if (lane == 0) base = atomicAdd(&s_nc[rc & 1], uint32_t(__popcll(hm)));
                                base = uniform_u(base);
                                if (hit) s_cand[base + lane_prefix(hm)] = jj[k];
                            }
                        }
                        __syncthreads();
                        // exact: full lanes over the round's candidates
                        const int nc = uniform_i(int(s_nc[rc & 1]));
                        if (tid == 0) s_nc[(rc + 1) & 1] = 0u;
                        ++rc;
                        for (int c0 = 0; c0 < nc; c0 += 256) {
                            const int c = c0 + tid;
                            bool heard = false;
                            int orig = 0;
                            double rssi = 0.0, prob = 1.0;
                            if (c < nc) {
                                const int pos = s_cand[c];
                                RxRecord rx_;
                                if (nd.rec32 == nullptr) {
                                    rx_ = nd.rec[pos];
                                } else { // the 32-byte record: channel and radio state were tested above
                                    const RxCompact rc = nd.rec32[pos];
                                    rx_.x = rc.x;
                                    rx_.y = rc.y;
                                    rx_.z = rc.z;
                                    rx_.orig = rc.orig;
                                    rx_.int_id = 0;
                                    rx_.channel = tx.channel;
                                    rx_.enabled = 1;
                                    rx_.rxprob = (rc.flags & 1u) ? nd.rxprob[pos] : 1.0;
                                }
                                const LinkEval ev = eval_link<MODEL, false>(m, nd, tx, rx_, true);
                                if (ev.wanted) {
                                    heard = true;
                                    orig = rx_.orig;
                                    if (MODEL == RM_MODEL_LOGDIST) {
                                        rssi = ev.aux;
                                        prob = rx_.rxprob;
                                    } else {
                                        rssi = tx.txpower; // reference media hand the packet's transmit power through
                                        prob = (MODEL == RM_MODEL_UDGM) ? ev.aux : 1.0;
                                    }
                                }
                            }
                            const uint64_t hm = ballot64(heard);
                            if (hm) {
                                uint32_t base = 0;
                                if (lane == 0) base = atomicAdd(&s_nres, uint32_t(__popcll(hm)));
                                base = uniform_u(base);
                                if (heard) {
                                    const uint32_t k = base + lane_prefix(hm);
                                    if (pass == 0) {
                                        if (k < seg) {
                                            s_orig[k] = orig;
                                            s_rssi[k] = rssi;
                                            if (STOCH) s_prob[k] = prob;
                                        }
                                    } else {
                                        const uint32_t o = s_base + k;
                                        t.a_dst[o] = orig;
                                        t.a_rssi[o] = rssi;
                                        if (STOCH) {
                                            t.a_prob[o] = prob;
                                            t.a_verdict[o] = uint8_t(0); // pending: k_apply_draws decides
                                        } else {
                                            t.a_verdict[o] = dead ? uint8_t(RM_INTERFERED) : uint8_t(RM_DELIVERED);
                                        }
                                    }
                                }
                            }
                        }
                        __syncthreads(); // the candidates are overwritten by the next round
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t total = s_nres;
        if (pass == 1) break;
        if (total <= seg) {
            for (uint32_t i = tid; i < total; i += blockDim.x) {
                const uint32_t o = fixed_base + i;
                t.a_dst[o] = s_orig[i];
                t.a_rssi[o] = s_rssi[i];
                if (STOCH) {
                    t.a_prob[o] = s_prob[i];
                    t.a_verdict[o] = uint8_t(0);
                } else {
                    t.a_verdict[o] = dead ? uint8_t(RM_INTERFERED) : uint8_t(RM_DELIVERED);
                }
            }
            if (tid == 0) {
                t.cursor[slot] = total;
                t.seg_off[slot] = fixed_base;
            }
            break;
        }
        // more links than the segment holds: room behind the fixed segments, then the candidates once more
        __syncthreads();
        if (tid == 0) {
            const uint32_t first = uint32_t(t.n_cnt) * seg;
            const uint32_t room = t.cap - first; // frame_tick_segment keeps the fixed segments within half of the records
            const uint32_t got = atomicAdd(&t.stage_count[7], total);
            uint32_t base = first + got;
            uint32_t keep = total;
            if (got > room || total > room - got) { // no room: the tick reports RM_ERR_CAPACITY
                t.stage_count[1] = 1u;
                keep = 0u;
                base = fixed_base;
            }
            t.cursor[slot] = keep;
            t.seg_off[slot] = base;
            s_base = base;
            s_nres = keep ? 0u : 0xFFFFFFFFu;
        }
        __syncthreads();
        if (s_nres == 0xFFFFFFFFu) break; // dropped
    }
}

// the largest per-frame segment the A records (capacity `cap`) allow for n_cnt frame slots, 0 = not this path
int frame_tick_segment(const TickDev &t, const LaunchCfg &cfg, const ModelDev &m)
{
    const bool geometric = (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_UDGM_CONST || m.kind == RM_MODEL_LOGDIST);
    const bool sinr = m.kind == RM_MODEL_LOGDIST && (m.flags & RM_LD_SINR);
    if (!geometric || sinr || !cfg.sorted || !cfg.bbox || cfg.f64_filter || t.use_matrix) return 0;
    if (t.n_cnt <= 0 || t.n_cnt > kFusedScanMax || t.n_rx <= 0 || t.n_active <= t.first_new) return 0;
    // half of the records for the fixed segments, the rest for frames that outgrow theirs
    const uint32_t per = (t.cap / 2u) / uint32_t(t.n_cnt);
    if (per < 64u) return 0;
    return int(per < uint32_t(kFrameSegMax) ? (per / 64u) * 64u : uint32_t(kFrameSegMax));
}

hipError_t launch_tick_frames(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t, const LaunchCfg &cfg, int seg_len)
{
    const dim3 grid(t.n_cnt), block(256);
#define RM_FR(MODEL)                                                                                             \
    do {                                                                                                         \
        if (cfg.stochastic) hipLaunchKernelGGL((k_tick_frames<MODEL, true>), grid, block, 0, s, nd, m, t, seg_len); \
        else hipLaunchKernelGGL((k_tick_frames<MODEL, false>), grid, block, 0, s, nd, m, t, seg_len);             \
    } while (0)
    switch (m.kind) {
    case RM_MODEL_UDGM: RM_FR(RM_MODEL_UDGM); break;
    case RM_MODEL_UDGM_CONST: RM_FR(RM_MODEL_UDGM_CONST); break;
    case RM_MODEL_LOGDIST: RM_FR(RM_MODEL_LOGDIST); break;
    default: return hipErrorInvalidValue;
    }
#undef RM_FR
    return hipGetLastError();
}

} // namespace rm
