// rm_tick.hip -- the closed-loop tick: ONE frame per workgroup, filter + exact evaluation + node order in one launch
// (part of libradiomedium_hip.so; gfx950 only, -ffp-contract=off, no fast-math; overview at the top of rm_engine.h)
//
// A lone tick is what the reference's call pattern produces: the emulators step, their transmit()
// calls are evaluated, the events are consumed in emulatorTimeStepDone (Simulator.java:155-165), and
// only then does the next tick begin.  Its few MB of work cannot fill the chip, so what it costs is
// the chain of dependent launches and memory round trips (a kernel of this runtime cannot finish in
// less than 3-4 us, whatever it does).  The sweep kernels (rm_filter / rm_exact / rm_reorder) are
// three dependent launches because the candidate list crosses workgroups; here a frame's whole
// evaluation stays inside one workgroup, as k_transmit_one (rm_transmit.hip) does for one packet:
//
//   boxes     the frame against the boxes of 16 receiver groups (1024 receivers), then the group boxes
//             (64 receivers) of the near ones; small tables test every group box directly              -> LDS lists
//   filter    the receivers of the near groups: fp32 pre-filter, exact channel test, and for the
//             shadowed medium the sweep's second-level test on the link hash                         -> LDS candidates
//   exact     the candidates with full lanes: the reference's fp64 arithmetic (eval_link),
//             UDGMRadioMedium.java:99-111, UDGMConstantLossRadioMedium.java:25-33                     -> LDS links
//   order     rank by node index -- the order the reference's loop visits receivers in
//             (UDGMRadioMedium.java:99) -- and write the frame's links into its own segment of the A records
//
// The result of the tick is then complete: per frame an ordered list (seg_off, cursor = count) in HBM.
// Consumers that walk it frame by frame (the host-mapped pack k_pack_frames below, the reception
// stage) read the segments as they are; the compact packet-major arrays of rm_device_result are
// produced by k_reorder only when somebody asks for them (rm_result_device / rm_result_copy), and
// before the java.util.Random kernels, which scan the compact records.  No candidate list, no
// atomics on global memory.  Every level works in rounds, so nothing is bounded by the LDS lists:
// a frame with more heard links than its segment holds takes its room from an overflow allocator,
// evaluates its candidates a second time and sorts its links through global memory.
#include "rm_device.hpp"

#include <stdlib.h>

namespace rm {

constexpr int kFrBoxes = 1024;      // level-1 boxes tested per round (4 per thread)
constexpr int kFrGroups = 2048;     // group boxes tested per round (8 per thread)
constexpr int kFrFlatGroups = 256;  // up to here every group box is tested directly (no level 1)
constexpr int kFrRound = 16;        // near groups per filter round: at most 1024 candidates
constexpr int kFrCand = kFrRound * kGroup;

// Diagnostic build only (make stamps; never the shipped library): s_memtime at the phase boundaries of
// k_tick_frames, written to the unused tail of the compact rssi array (tools/tick_stamps.py reads them).
#ifdef RM_STAMPS
#define RM_STAMP(k)                                                                                          \
    do {                                                                                                     \
        if (threadIdx.x == 0) stamps[k] = __builtin_amdgcn_s_memtime();                                      \
    } while (0)
#else
#define RM_STAMP(k)                                                                                          \
    do {                                                                                                     \
    } while (0)
#endif

RM_D bool box_near(const float4 &qb, const float2 &qz, const float4 &f)
{
    const float dx = fmaxf(fmaxf(qb.x - f.x, f.x - qb.z), 0.f);
    const float dy = fmaxf(fmaxf(qb.y - f.y, f.y - qb.w), 0.f);
    const float dz = fmaxf(fmaxf(qz.x - f.z, f.z - qz.y), 0.f);
    return dist2_f32(dx, dy, dz) <= f.w;
}

// Which frame does workgroup b take?  The hardware deals workgroups to the eight XCDs in turn (b % 8), each with its own L2.  With
// xcd_map the frames of a tick are dealt in EIGHTHS instead -- XCD x takes the x-th eighth of the slots -- so that, where the
// caller's frame order is a spatial order (node ids assigned along a space-filling curve), the frames that share receiver groups
// share an L2 as well and a group's records leave HBM for one XCD, not for up to eight: 14.0 -> 5.9 MB of counter traffic per tick
// of configs[2] (4.85 MB algorithmic; 12.8 MB with arbitrary ids, which the renaming neither helps nor hurts), 11.09 -> 11.03 us --
// the tick is a chain of round trips, not of bytes (profiles/r05_c3_tick_spatial_xcd*).  A pure renaming of the workgroups: every
// slot is still taken exactly once.  RM_TICK_XCD_MAP=0: workgroup b takes slot b.
RM_D int xcd_slot(const int b, const int n)
{
    const int per = n >> 3, rem = n & 7, x = b & 7;
    return x * per + min(x, rem) + (b >> 3);
}

template <int MODEL, bool STOCH, bool SHADOW, bool FLAT, bool SINR = false>
RM_D void tick_frames_body(const NodesDev &nd, const ModelDev &m, const TickDev &t, const int seg_len, const ScanDev *sdp = nullptr,
                           const int xcd_map = 0)
{
    // one LDS block, carved by hand: the lists are dead when a frame that outgrew its segment orders its links,
    // and that ordering wants all of it for a bitmap over the node indices (below)
    constexpr int kWl1 = FLAT ? 0 : kFrBoxes, kWprob = STOCH ? 2 * kFrameSegMax : 0, kWtbl = SHADOW ? kShadowBins : 0;
    constexpr int kWords = kWl1 + kFrGroups + kFrCand + kFrameSegMax + 2 * kFrameSegMax + kWprob + kWtbl;
    static_assert((kWl1 + kFrGroups + kFrCand + kFrameSegMax) % 2 == 0, "the doubles behind the lists are 8-byte aligned");
    __shared__ __attribute__((aligned(16))) uint32_t s_mem[kWords];
    int *const s_l1 = reinterpret_cast<int *>(s_mem);
    int *const s_l2 = s_l1 + kWl1;
    int *const s_cand = s_l2 + kFrGroups;
    int *const s_orig = s_cand + kFrCand;
    double *const s_rssi = reinterpret_cast<double *>(s_orig + kFrameSegMax);
    double *const s_prob = s_rssi + kFrameSegMax; // (STOCH only)
    uint32_t *const s_tbl = s_mem + (kWords - kWtbl); // (SHADOW only)
    __shared__ uint32_t s_n1[2], s_n2[2], s_nc[2], s_nres, s_base; // the lists' fill counts, by round parity

    const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
    const int slot = (xcd_map && int(blockIdx.x) < t.n_cnt) ? xcd_slot(int(blockIdx.x), t.n_cnt) : int(blockIdx.x);
    const int n_new = t.n_active - t.first_new;
    const int n_groups = (nd.n_rx + kGroup - 1) / kGroup;
    const int n_boxes = (n_groups + 15) / 16;
#ifdef RM_STAMPS
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(t.out_rssi + (size_t(t.cap) - 16u * (size_t(slot) + 1u)));
    if (threadIdx.x == 0) stamps[15] = __builtin_amdgcn_s_memrealtime();
#endif
    RM_STAMP(0);

    // Program order = issue order: the frame's record first (two dependent round trips when it is built from a
    // source index), then everything that does not depend on it, so that it all flies under those round trips.
    const int q = slot - t.shift;
    const bool real = q >= 0 && q < n_new;
    const int abs_i = t.first_new + (real ? q : 0);
    const bool build = t.src_list != nullptr;
    rm_tx_record tx;
    int s_idx = -1;
    const bool from_host = !build && t.gather_src != nullptr; // the records still lie in the host's pinned block (rm_tick_flush*)
    if (build) s_idx = t.src_list[real ? q : 0];
    else if (from_host) tx = t.gather_src[size_t(abs_i / t.gather_slots) * size_t(t.gather_stride) + size_t(abs_i % t.gather_slots)];
    else tx = t.tx[abs_i];

    // the first round's boxes do not depend on the frame (branch-free: the index is clamped, validity is tested at use)
    constexpr int kPre = FLAT ? kFrGroups / 256 : kFrBoxes / 256;
    float4 pre_xy[kPre];
    float2 pre_z[kPre];
    {
        const int nb = max(FLAT ? n_groups : n_boxes, 1);
        const float4 *__restrict__ bxy = FLAT ? nd.bbox_xy : nd.wg_box_xy;
        const float2 *__restrict__ bz = FLAT ? nd.bbox_z : nd.wg_box_z;
#pragma unroll
        for (int k = 0; k < kPre; ++k) {
            const int b = min(k * 256 + tid, nb - 1);
            pre_xy[k] = bxy[b];
            pre_z[k] = bz[b];
        }
    }
    uint32_t tbl_word = 0u;
    if (SHADOW) tbl_word = m.shadow_tbl[tid]; // kBlock == kShadowBins; needed by the filter phase only

    if (build) { // RadioPacket(node, time, data) copies txpower / channel from its source, RadioPacket.java:46-52
        const bool pad = s_idx < 0 || s_idx >= nd.n;
        const int sc = pad ? 0 : s_idx;
        const SrcRecord sr = nd.srec[sc]; // (one line of the source table, not six)
        tx.x = sr.x;
        tx.y = sr.y;
        tx.z = sr.z;
        tx.txpower = sr.txpower;
        tx.txprob = sr.txprob;
        tx.channel = sr.channel;
        tx.start_us = t.src_start_us;
        tx.air_us = t.src_air_us;
        tx.src = s_idx;
        if (pad) tx = make_tx_record(nd, -1, t.src_start_us, t.src_air_us);
    }

    // what the sweep's first kernel does for the tick after this one (rm_filter.hip, tick_prep_body)
    if (blockIdx.x == 0) {
        if (tid < 8) t.next_counters[tid] = 0u;
        t.next_shard_count[tid * kShardStride] = 0u; // kBlock == kShards
        if (SINR && tid == 0 && t.air.bad[0]) t.stage_count[1] = 1u; // the on-air lists are broken until the host rebuilds them
    }
    for (int i = blockIdx.x * blockDim.x + tid; i < t.zero_len; i += gridDim.x * blockDim.x) t.cand_tot_next[i] = 0u;

    // the SINR medium's tick by scan (rm_airscan.hip): every frame on the air indexed for the second launch
    if (MODEL == RM_MODEL_LOGDIST && !SINR && sdp != nullptr) {
        const ScanDev &sd = *sdp;
        ModelDev ml = m;
        ml.ld_level = sd.level;
        if (slot >= t.n_cnt) { // (the launch has extra workgroups for this behind the frames' own: one frame on the air per thread)
            const int i = (slot - t.n_cnt) * int(blockDim.x) + tid;
            if (i < t.n_active) {
                // (a new frame given as a source index: the same record its own workgroup builds)
                const rm_tx_record r = (build && i >= t.first_new) ? make_tx_record(nd, t.src_list[i - t.first_new], t.src_start_us, t.src_air_us)
                                                                   : t.tx[i];
                scan_index(ml, t, sd, nd.n, i, r);
            }
        }
    }

    if (!real) { // padding slot of the per-frame counters (a batch launches the largest tick's grid for every tick)
        if (tid == 0 && slot < t.n_cnt) {
            t.cursor[slot] = 0u;
            t.seg_off[slot] = uint32_t(slot) * uint32_t(seg_len);
        }
        return;
    }
    if ((build || from_host) && tid == 0) t.tx_build[abs_i] = tx; // where every later consumer of the tick finds the record
    if (!STOCH && t.check_txprob && tid == 0 && tx.src >= 0 && tx.txprob > 0.0 && tx.txprob < 1.0) t.stage_count[6] = 2u;
    float4 f;
    double thr64;
    tx_prefilter(m, tx, f, thr64);
    RM_STAMP(1); // the frame's record is there
    // shadowed medium: can the link still reach the level with its own deviate?  The sweep's second-level
    // filter (rm_filter.hip): conservative table of the largest link hash that can, per bin of d^2 / cut^2
    float shadow_inv = 0.f;
    if (SHADOW) {
        s_tbl[tid] = tbl_word;
        if (f.w > 0.f && f.w < __builtin_inff()) {
            // is the fp32 frame error small against the distances where the table decides anything (d > 0.15 cut)?
            // Otherwise bin 0 (always pass).  An fp32 evaluation with a 1 % margin: the test only chooses between
            // two conservative filters.
            const float cut = __builtin_sqrtf(f.w);
            if (1.01f * (2.0f * float(m.f32_slack)) / (0.15f * cut) + 1e-5f <= float(kShadowPad)) shadow_inv = float(kShadowBins) / f.w;
        }
    }
    if (tid == 0) s_n1[0] = s_n1[1] = s_n2[0] = s_n2[1] = s_nc[0] = s_nc[1] = s_nres = s_base = 0u;
    __syncthreads();
    // A list's fill count of one round is cleared while the next round's -- the other parity -- is in use: every
    // reuse of a list or a count is separated from its last reader by a barrier, without barriers for the clearing.
    int r1 = 0, r2 = 0, rc = 0;

    const bool dead = (MODEL != RM_MODEL_UDGM_CONST) && tx_success(m, tx) <= 0.0; // UDGMRadioMedium.java:88
    const uint32_t seg = uint32_t(seg_len);
    const uint32_t fixed_base = uint32_t(slot) * seg;

    // pass 0 keeps the links in LDS; pass 1 only runs for a frame that heard more than its segment holds
    // and writes them straight to the room it got from the overflow allocator
    for (int pass = 0; pass < 2; ++pass) { // block-uniform
        if (f.w >= 0.f) {
            for (int b0 = 0; b0 < (FLAT ? 1 : n_boxes); b0 += kFrBoxes) {
                int n1 = 0;
                if (!FLAT) {
                    // level 1: the boxes of 16 groups
#pragma unroll
                    for (int k = 0; k < kFrBoxes / 256; ++k) {
                        const int b = b0 + k * 256 + tid;
                        bool hit = false;
                        if (b < n_boxes) {
                            if (b0 == 0 && pass == 0) hit = box_near(pre_xy[k], pre_z[k], f);
                            else hit = box_near(nd.wg_box_xy[b], nd.wg_box_z[b], f);
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
                    RM_STAMP(2); // level 1 done
                    n1 = uniform_i(int(s_n1[r1 & 1]));
                    if (tid == 0) s_n1[(r1 + 1) & 1] = 0u;
                    ++r1;
                }
                const int n_l2 = FLAT ? n_groups : n1 * 16;
                for (int i0 = 0; i0 < n_l2; i0 += kFrGroups) {
                    // level 2: the group boxes
#pragma unroll
                    for (int k = 0; k < kFrGroups / 256; ++k) {
                        if (i0 + k * 256 >= n_l2) break; // block-uniform
                        const int i = i0 + k * 256 + tid;
                        bool hit = false;
                        int g = 0;
                        if (i < n_l2) {
                            g = FLAT ? i : s_l1[i >> 4] * 16 + (i & 15);
                            if (g < n_groups) {
                                if (FLAT && i0 == 0 && pass == 0) hit = box_near(pre_xy[k < kPre ? k : 0], pre_z[k < kPre ? k : 0], f);
                                else hit = box_near(nd.bbox_xy[g], nd.bbox_z[g], f);
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
                    RM_STAMP(3); // level 2 done
                    const int n2 = uniform_i(int(s_n2[r2 & 1]));
                    if (tid == 0) s_n2[(r2 + 1) & 1] = 0u;
                    ++r2;
                    for (int gi0 = 0; gi0 < n2; gi0 += kFrRound) {
                        // filter: every wave takes four of the round's groups, their records requested together
                        float4 v[kFrRound / 4];
                        int jj[kFrRound / 4], oo[kFrRound / 4];
#pragma unroll
                        for (int k = 0; k < kFrRound / 4; ++k) {
                            const int gi = gi0 + k * 4 + wave;
                            jj[k] = -1;
                            oo[k] = 0;
                            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (gi < n2) {
                                const int j = s_l2[gi] * kGroup + lane;
                                if (j < nd.n_rx) {
                                    jj[k] = j;
                                    v[k] = nd.rxf[j];
                                    if (SHADOW) oo[k] = nd.orig[j];
                                }
                            }
                        }
                        uint64_t hms[kFrRound / 4];
                        uint32_t wave_hits = 0;
#pragma unroll
                        for (int k = 0; k < kFrRound / 4; ++k) {
                            const float s2 = dist2_f32(v[k].x - f.x, v[k].y - f.y, v[k].z - f.z);
                            bool hit = jj[k] >= 0 && s2 <= f.w && __float_as_int(v[k].w) == tx.channel;
                            if (SHADOW && hit) {
                                const int bin = min(kShadowBins - 1, int(s2 * shadow_inv));
                                const uint32_t a = uint32_t(tx.src), b = uint32_t(oo[k]);
                                const uint64_t key = (uint64_t(a < b ? a : b) << 32) | uint64_t(a < b ? b : a);
                                hit = uint32_t(mix64(m.ld_seed_mixed ^ key) >> 32) <= s_tbl[bin];
                            }
                            hms[k] = ballot64(hit);
                            wave_hits += uint32_t(__popcll(hms[k]));
                        }
                        if (wave_hits) { // one LDS atomic per wave and round
                            uint32_t base = 0;
                            if (lane == 0) base = atomicAdd(&s_nc[rc & 1], wave_hits);
                            base = uniform_u(base);
#pragma unroll
                            for (int k = 0; k < kFrRound / 4; ++k) {
                                if ((hms[k] >> lane) & 1ull) s_cand[base + lane_prefix(hms[k])] = jj[k];
                                base += uint32_t(__popcll(hms[k]));
                            }
                        }
                        __syncthreads();
                        RM_STAMP(4); // filter done
                        // exact: full lanes over the round's candidates
                        const int nc = uniform_i(int(s_nc[rc & 1]));
                        if (tid == 0) s_nc[(rc + 1) & 1] = 0u;
                        ++rc;
                        for (int c0 = 0; c0 < nc; c0 += 256) {
                            const int c = c0 + tid;
                            bool heard = false;
                            int orig = 0;
                            double rssi = 0.0, prob = 1.0;
                            bool ins = false; // (SINR) the link is significant at its receiver: an entry of that receiver's on-air list
                            double ins_lin = 0.0;
                            int ins_pos = 0;
                            if (c < nc) {
                                const int pos = s_cand[c];
                                RxRecord rx_;
                                if (nd.rec32 == nullptr) {
                                    rx_ = nd.rec[pos];
                                } else { // the 32-byte record: channel and radio state were tested above
                                    const RxCompact rc32 = nd.rec32[pos];
                                    rx_.x = rc32.x;
                                    rx_.y = rc32.y;
                                    rx_.z = rc32.z;
                                    rx_.orig = rc32.orig;
                                    rx_.int_id = 0;
                                    rx_.channel = tx.channel;
                                    rx_.enabled = 1;
                                    rx_.rxprob = (rc32.flags & 1u) ? nd.rxprob[pos] : 1.0;
                                }
                                const LinkEval ev = eval_link<MODEL, SINR>(m, nd, tx, rx_, true);
                                if (SINR) {
                                    ins = pass == 0 && ev.append && (ev.flags & kFlagInterferer) != 0;
                                    ins_lin = ev.lin;
                                    ins_pos = pos;
                                }
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
                            if (SINR) { // every wave of the workgroup is here: one allocation per wave (first pass only)
                                // (gathering a frame's entries in LDS and inserting them in one go was measured: no faster)
                                const int aidx = air_alloc(t, ins, air_sub(t));
                                if (ins) air_link(t, aidx, ins_pos, tx.start_us, tx.air_us, ins_lin, kAirInterferer);
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
                                    } else { // unordered, behind the fixed segments; sorted below
                                        const uint32_t o = s_base + k;
                                        t.a_dst[o] = orig;
                                        t.a_rssi[o] = rssi;
                                        if (STOCH) t.a_prob[o] = prob;
                                    }
                                }
                            }
                        }
                        __syncthreads(); // the candidates are overwritten by the next round
                        RM_STAMP(5); // exact done
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t total = s_nres;
        if (pass == 1) {
            // the frame's links sit unordered at s_base: rank by node index, permute into the (so far unused) compact
            // arrays of the same range, copy back.
            const uint32_t base = s_base;
            // Node indices are unique inside a frame: a bitmap over the partition's nodes in LDS ranks all of them in
            // O(links + nodes / 32): rank(j) = bits set below j = prefix of its 256-bit block + the words before it.
            const uint32_t n_bits = uint32_t(nd.pos_span), nw = (n_bits + 31u) >> 5, nblk = (nw + 7u) >> 3;
            const bool bitmap = nw + nblk + 8u <= uint32_t(kWords);
            if (bitmap) {
                uint32_t *const bits = s_mem, *const bpre = s_mem + ((nw + 7u) & ~7u);
                __syncthreads();
                for (uint32_t w = tid; w < ((nw + 7u) & ~7u) + nblk; w += 256) s_mem[w] = 0u;
                __syncthreads();
                for (uint32_t i = tid; i < total; i += 256) {
                    const uint32_t j = uint32_t(t.a_dst[base + i] - nd.rx_first);
                    atomicOr(&bits[j >> 5], 1u << (j & 31u));
                }
                __syncthreads();
                for (uint32_t b = tid; b < nblk; b += 256) { // bits per block of eight words
                    uint32_t c = 0;
#pragma unroll
                    for (int w = 0; w < 8; ++w) c += uint32_t(__popc(bits[b * 8u + w]));
                    bpre[b] = c;
                }
                __syncthreads();
                if (tid < 64) { // exclusive scan of the block counts by one wave
                    uint32_t carry = 0;
                    for (uint32_t b0 = 0; b0 < nblk; b0 += 64) { // wave-uniform
                        const uint32_t b = b0 + lane;
                        const uint32_t v = (b < nblk) ? bpre[b] : 0u;
                        const uint32_t inc = wave_inclusive_scan(v, lane);
                        if (b < nblk) bpre[b] = carry + inc - v;
                        carry += uint32_t(__shfl(int(inc), 63));
                    }
                }
                __syncthreads();
                for (uint32_t i = tid; i < total; i += 256) {
                    const int node = t.a_dst[base + i];
                    const uint32_t j = uint32_t(node - nd.rx_first);
                    uint32_t rank = bpre[j >> 8];
                    for (uint32_t w = (j >> 8) << 3; w < (j >> 5); ++w) rank += uint32_t(__popc(bits[w]));
                    rank += uint32_t(__popc(bits[j >> 5] & ((1u << (j & 31u)) - 1u)));
                    t.out_dst[base + rank] = node;
                    t.out_rssi[base + rank] = t.a_rssi[base + i];
                    if (STOCH) t.out_prob[base + rank] = t.a_prob[base + i];
                }
            }
            for (uint32_t i0 = 0; !bitmap && i0 < total; i0 += 256) { // block-uniform: tables too large for the bitmap
                const uint32_t i = i0 + tid;
                const int mine = (i < total) ? t.a_dst[base + i] : 0x7fffffff;
                uint32_t rank = 0;
                for (uint32_t k0 = 0; k0 < total; k0 += kFrCand) {
                    __syncthreads();
                    for (uint32_t k = tid; k < uint32_t(kFrCand) && k0 + k < total; k += 256) s_cand[k] = t.a_dst[base + k0 + k];
                    __syncthreads();
                    const uint32_t nk = min(uint32_t(kFrCand), total - k0);
                    for (uint32_t k = 0; k < nk; ++k) rank += (s_cand[k] < mine) ? 1u : 0u;
                }
                if (i < total) {
                    t.out_dst[base + rank] = mine;
                    t.out_rssi[base + rank] = t.a_rssi[base + i];
                    if (STOCH) t.out_prob[base + rank] = t.a_prob[base + i];
                }
            }
            __syncthreads();
            for (uint32_t i = tid; i < total; i += 256) {
                t.a_dst[base + i] = t.out_dst[base + i];
                t.a_rssi[base + i] = t.out_rssi[base + i];
                if (STOCH) {
                    t.a_prob[base + i] = t.out_prob[base + i];
                    t.a_verdict[base + i] = uint8_t(0); // pending: k_apply_draws decides
                } else {
                    t.a_verdict[base + i] = dead ? uint8_t(RM_INTERFERED) : uint8_t(RM_DELIVERED);
                }
            }
            break;
        }
        if (total <= seg) {
            // node order: every link is written to its final place of the segment.  Up to 64 links (the usual
            // frame) sit one per lane of the first wave and are ranked with a readlane loop; more by counting in LDS.
            if (total <= 64u && tid < 64) {
                // every lane of the first wave takes part (lanes without a link hold the largest key), so that the
                // loop can be unrolled: eight independent readlanes in flight instead of one SALU -> VALU round trip each
                const bool have = uint32_t(tid) < total;
                const int mine = have ? s_orig[tid] : 0x7fffffff;
                uint32_t rank = 0;
                const int n8 = uniform_i(int((total + 7u) & ~7u));
                for (int k0 = 0; k0 < n8; k0 += 8) {
                    int v8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v8[u] = __builtin_amdgcn_readlane(mine, k0 + u);
#pragma unroll
                    for (int u = 0; u < 8; ++u) rank += (v8[u] < mine) ? 1u : 0u;
                }
                if (have) {
                    const uint32_t o = fixed_base + rank;
                    t.a_dst[o] = mine;
                    t.a_rssi[o] = s_rssi[tid];
                    if (STOCH) {
                        t.a_prob[o] = s_prob[tid];
                        t.a_verdict[o] = uint8_t(0);
                    } else {
                        t.a_verdict[o] = dead ? uint8_t(RM_INTERFERED) : uint8_t(RM_DELIVERED);
                    }
                }
            }
            for (uint32_t i = tid; total > 64u && i < total; i += blockDim.x) {
                const int mine = s_orig[i];
                uint32_t rank = 0;
                {
                    uint32_t k = 0;
                    for (; k + 8u <= total; k += 8u) { // eight independent LDS reads in flight
                        int v8[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v8[u] = s_orig[k + u];
#pragma unroll
                        for (int u = 0; u < 8; ++u) rank += (v8[u] < mine) ? 1u : 0u;
                    }
                    for (; k < total; ++k) rank += (s_orig[k] < mine) ? 1u : 0u;
                }
                const uint32_t o = fixed_base + rank;
                t.a_dst[o] = mine;
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
            RM_STAMP(6); // ordered and written
#ifdef RM_STAMPS
            if (threadIdx.x == 0) stamps[14] = __builtin_amdgcn_s_memrealtime();
#endif
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
    if (SINR && wave == 0) { // half duplex: the frame's SELF entry in its source's list (two atomics in a row: last)
        const int self_pos = (lane == 0) ? engine_pos(nd, tx.src) : -1;
        const bool want = self_pos >= 0;
        const int aidx = air_alloc(t, want, air_sub(t));
        if (want) air_link(t, aidx, self_pos, tx.start_us, tx.air_us, 0.0, kAirSelf);
    }
}

template <int MODEL, bool STOCH, bool SHADOW, bool FLAT>
__global__ void __launch_bounds__(256) k_tick_frames(const NodesDev nd, const ModelDev m, const TickDev t, const int seg_len, const int xcd_map)
{
    tick_frames_body<MODEL, STOCH, SHADOW, FLAT>(nd, m, t, seg_len, nullptr, xcd_map);
}

// ... and as the first launch of the SINR medium's tick by scan (rm_airscan.hip): the heard links as the medium without SINR
// finds them, and every frame on the air indexed for the second launch
template <bool STOCH, bool SHADOW, bool FLAT>
__global__ void __launch_bounds__(256)
k_tick_frames_scan(const NodesDev nd, const ModelDev m, const TickDev t, const int seg_len, const ScanDev sd)
{
    tick_frames_body<RM_MODEL_LOGDIST, STOCH, SHADOW, FLAT>(nd, m, t, seg_len, &sd);
}

// The SINR medium in the same form (a lone tick that only adds its new frames to the on-air lists, section 4.4 of
// DESIGN.md): the exact phase also leaves the frame's significant links in their receivers' lists, and a second launch --
// the lists are complete only when every frame of the tick has been evaluated -- walks them for the heard links.
template <bool STOCH, bool SHADOW, bool FLAT>
__global__ void __launch_bounds__(256) k_tick_frames_sinr(const NodesDev nd, const ModelDev m, const TickDev t, const int seg_len)
{
    tick_frames_body<RM_MODEL_LOGDIST, STOCH, SHADOW, FLAT, true>(nd, m, t, seg_len);
}

// one wave per frame, one lane per heard link of its (ordered) segment: sinr and the capture / half-duplex verdict
__global__ void __launch_bounds__(256) k_sinr_frames(const NodesDev nd, const ModelDev m, const TickDev t)
{
    const int lane = threadIdx.x & 63;
    const int n_new = t.n_active - t.first_new;
    for (int q = blockIdx.x * 4 + wave_index(); q < n_new; q += gridDim.x * 4) { // wave-uniform
        const int slot = q + t.shift;
        const uint32_t src0 = uniform_u(t.seg_off[slot]);
        const uint32_t len = uniform_u(t.cursor[slot]);
        const rm_tx_record &w = t.tx[t.first_new + q];
        for (uint32_t c = lane; c < len; c += 64) {
            const uint32_t o = src0 + c;
            const double rssi = t.a_rssi[o];
            const int pos = engine_pos(nd, t.a_dst[o]); // (a heard link's receiver is one of this partition's)
            const SinrOut so = air_sinr(m, t, pos, kAirOwnInSum, w.start_us, w.air_us, rssi);
            t.a_sinr[o] = so.sinr;
            if (so.collided) t.a_verdict[o] = uint8_t(RM_INTERFERED);
        }
    }
    if (blockIdx.x == 0) air_end(t); // the next tick's entries begin where the sub-rings' tails are now
}

// the same for the ticks of a batch (blockIdx.z = tick; descriptors in device memory, as the sweep's batched kernels)
template <int MODEL, bool STOCH, bool SHADOW, bool FLAT>
__global__ void __launch_bounds__(256)
k_tick_frames_batch(const NodesDev nd, const ModelDev m, const TickDev *__restrict__ ticks, const int seg_len)
{
    tick_frames_body<MODEL, STOCH, SHADOW, FLAT>(nd, m, ticks[blockIdx.z], seg_len);
}

// ---------------------------------------------------------------------------------------------------
// The sweep's candidate list, one frame per workgroup: for a lone tick over a table so large that the sweep's
// filter tiles the receivers (k_tick_prep + k_filter_wg: every tile of 1024 receivers looks at every frame, then the
// waves of a tile walk its near frames one after the other -- about 30 us at a million receivers however few the
// frames).  Here a frame finds its own receivers through the two box levels, as k_tick_frames does, and appends them
// to the candidate list k_exact walks.  Used by the SINR medium, whose links cross workgroups (the per-receiver
// interferer lists), and does what k_tick_prep does besides: the frame's record in build mode, the counters of the
// next tick, the SELF entry of the on-air lists.
template <bool SHADOW>
__global__ void __launch_bounds__(256) k_frames_cand(const NodesDev nd, const ModelDev m, const TickDev t)
{
    __shared__ int s_l1[kFrBoxes];
    __shared__ int s_l2[kFrGroups];
    constexpr int kRound = 32;                  // near groups per filter round: eight per wave, their records requested together
    constexpr int kRoundCand = kRound * kGroup; // what a round can add at most
    constexpr int kCandLds = 3 * kRoundCand;    // candidates gathered between two appends to the list
    __shared__ int s_cand[kCandLds];
    __shared__ uint32_t s_tbl[SHADOW ? kShadowBins : 1];
    __shared__ uint32_t s_n1[2], s_n2[2], s_nc[2], s_base;

    const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
    const int e = blockIdx.x; // eval-relative frame
    const int n_groups = (nd.n_rx + kGroup - 1) / kGroup;
    const int n_boxes = (n_groups + 15) / 16;
    const int abs_i = t.first_eval + e;
    const bool build = t.src_list != nullptr && abs_i >= t.first_new;
    rm_tx_record tx;
    int s_idx = -1;
    if (build) s_idx = t.src_list[abs_i - t.first_new];
    else tx = t.tx[abs_i];

    // the first round's level-1 boxes do not depend on the frame: requested under the record's round trips
    constexpr int kPre = kFrBoxes / 256;
    float4 pre_xy[kPre];
    float2 pre_z[kPre];
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
        const int b = min(k * 256 + tid, max(n_boxes, 1) - 1);
        pre_xy[k] = nd.wg_box_xy[b];
        pre_z[k] = nd.wg_box_z[b];
    }
    if (SHADOW) s_tbl[tid] = m.shadow_tbl[tid]; // kBlock == kShadowBins
    if (build) tx = make_tx_record(nd, s_idx, t.src_start_us, t.src_air_us);

    // what k_tick_prep does for the tick after this one
    if (blockIdx.x == 0) {
        if (tid < 8) t.next_counters[tid] = 0u;
        t.next_shard_count[tid * kShardStride] = 0u; // kBlock == kShards
        if (t.air.pool != nullptr && tid == 0 && t.air.bad[0]) t.stage_count[1] = 1u; // the lists are broken until rebuilt
    }
    for (int i = blockIdx.x * blockDim.x + tid; i < t.zero_len; i += gridDim.x * blockDim.x) {
        t.cursor[i] = 0u;
        t.cand_tot_next[i] = 0u;
    }
    if (build && tid == 0) t.tx_build[abs_i] = tx;
    if (t.check_span && tid == 0 && tx.src >= 0 && (tx.start_us < t.span_begin || tx.start_us + tx.air_us > t.span_end))
        t.stage_count[6] = 1u;
    if (t.check_txprob && tid == 0 && tx.src >= 0 && tx.txprob > 0.0 && tx.txprob < 1.0) t.stage_count[6] = 2u;
    float4 f;
    double thr64;
    tx_prefilter(m, tx, f, thr64);
    float shadow_inv = 0.f;
    if (SHADOW && f.w > 0.f && f.w < __builtin_inff()) {
        const float cut = __builtin_sqrtf(f.w);
        if (1.01f * (2.0f * float(m.f32_slack)) / (0.15f * cut) + 1e-5f <= float(kShadowPad)) shadow_inv = float(kShadowBins) / f.w;
    }
    if (tid == 0) s_n1[0] = s_n1[1] = s_n2[0] = s_n2[1] = s_nc[0] = s_nc[1] = 0u;
    __syncthreads();
    int r1 = 0, r2 = 0, rc = 0;
    uint32_t total = 0, flushes = 0;
    const bool counted = abs_i >= t.first_new; // frames that get verdicts: their candidates size the frame's segment
    int pending_rounds = 0;
    // the gathered candidates join the list k_exact walks: one atomic reserves their run in a shard (block-uniform call)
    auto flush = [&]() {
        __syncthreads(); // the rounds' LDS writes
        const int nc = uniform_i(int(s_nc[rc & 1]));
        if (nc > 0) {
            const uint32_t shard = (uint32_t(e) * 7u + flushes * 37u) & t.shard_mask;
            if (tid == 0) {
                s_base = atomicAdd(&t.shard_count[shard * kShardStride], uint32_t(nc));
                s_nc[(rc + 1) & 1] = 0u;
            }
            __syncthreads();
            const uint32_t base = uniform_u(s_base);
            if (base + uint32_t(nc) > t.seg_cap) { // the shard is full: drop the run, flag the tick
                if (tid == 0) t.stage_count[1] = 1u;
            } else {
                for (int c = tid; c < nc; c += 256) {
                    const uint32_t idx = shard * t.seg_cap + base + uint32_t(c);
                    t.st_pkt[idx] = e;
                    t.st_dst[idx] = s_cand[c];
                }
                total += uint32_t(nc);
            }
            ++flushes;
            ++rc;
            __syncthreads(); // s_cand and s_base are reused
        }
    };

    if (f.w >= 0.f) {
        for (int b0 = 0; b0 < n_boxes; b0 += kFrBoxes) {
            // level 1: the boxes of 16 groups
#pragma unroll
            for (int k = 0; k < kFrBoxes / 256; ++k) {
                const int b = b0 + k * 256 + tid;
                bool hit = false;
                if (b < n_boxes) hit = (b0 == 0) ? box_near(pre_xy[k], pre_z[k], f) : box_near(nd.wg_box_xy[b], nd.wg_box_z[b], f);
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
            const int n_l2 = n1 * 16;
            for (int i0 = 0; i0 < n_l2; i0 += kFrGroups) {
                // level 2: the group boxes
#pragma unroll
                for (int k = 0; k < kFrGroups / 256; ++k) {
                    if (i0 + k * 256 >= n_l2) break; // block-uniform
                    const int i = i0 + k * 256 + tid;
                    bool hit = false;
                    int g = 0;
                    if (i < n_l2) {
                        g = s_l1[i >> 4] * 16 + (i & 15);
                        if (g < n_groups) hit = box_near(nd.bbox_xy[g], nd.bbox_z[g], f);
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
                for (int gi0 = 0; gi0 < n2; gi0 += kRound) {
                    // filter: every wave takes eight of the round's groups, their records requested together
                    float4 v[kRound / 4];
                    int jj[kRound / 4], oo[kRound / 4];
#pragma unroll
                    for (int k = 0; k < kRound / 4; ++k) {
                        const int gi = gi0 + k * 4 + wave;
                        jj[k] = -1;
                        oo[k] = 0;
                        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (gi < n2) {
                            const int j = s_l2[gi] * kGroup + lane;
                            if (j < nd.n_rx) {
                                jj[k] = j;
                                v[k] = nd.rxf[j];
                                if (SHADOW) oo[k] = nd.orig[j];
                            }
                        }
                    }
                    uint64_t hms[kRound / 4];
                    uint32_t wave_hits = 0;
#pragma unroll
                    for (int k = 0; k < kRound / 4; ++k) {
                        const float s2 = dist2_f32(v[k].x - f.x, v[k].y - f.y, v[k].z - f.z);
                        bool hit = jj[k] >= 0 && s2 <= f.w && __float_as_int(v[k].w) == tx.channel;
                        if (SHADOW && hit) {
                            const int bin = min(kShadowBins - 1, int(s2 * shadow_inv));
                            const uint32_t a = uint32_t(tx.src), b = uint32_t(oo[k]);
                            const uint64_t key = (uint64_t(a < b ? a : b) << 32) | uint64_t(a < b ? b : a);
                            hit = uint32_t(mix64(m.ld_seed_mixed ^ key) >> 32) <= s_tbl[bin];
                        }
                        hms[k] = ballot64(hit);
                        wave_hits += uint32_t(__popcll(hms[k]));
                    }
                    if (wave_hits) { // one LDS atomic per wave and round
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(&s_nc[rc & 1], wave_hits);
                        base = uniform_u(base);
#pragma unroll
                        for (int k = 0; k < kRound / 4; ++k) {
                            if ((hms[k] >> lane) & 1ull) s_cand[base + lane_prefix(hms[k])] = jj[k];
                            base += uint32_t(__popcll(hms[k]));
                        }
                    }
                    // no barrier between rounds: the waves stream through their groups; the candidates join the
                    // list when another round might not fit any more (block-uniform: a round adds at most kRoundCand)
                    if (++pending_rounds * kRoundCand + kRoundCand > kCandLds) {
                        flush();
                        pending_rounds = 0;
                    }
                }
                // (the lists s_l2 / s_l1 are rewritten by the next chunk: everybody is done reading them)
                __syncthreads();
            }
        }
    }
    flush();
    if (counted && tid == 0 && total) t.cand_tot[e - t.cnt_base] = total; // (zeroed by the tick before; one workgroup per frame)
    if (t.air.pool != nullptr && wave == 0) { // half duplex: the frame's SELF entry in its source's list (two atomics in a row: last)
        const int self_pos = (lane == 0) ? engine_pos(nd, tx.src) : -1;
        const bool want = self_pos >= 0;
        const int aidx = air_alloc(t, want, air_sub(t));
        if (want) air_link(t, aidx, self_pos, tx.start_us, tx.air_us, 0.0, kAirSelf);
    }
}

// ---------------------------------------------------------------------------------------------------
// The tick's result written straight from the frames' segments into the host-mapped block of
// rm_tick_flush* (header, packet offsets, Tx-failure flags, records): every workgroup redoes the scan
// of the per-frame counts in LDS and copies runs of 256 consecutive compact records, the workgroup
// that finishes last publishes the sequence number the host polls.  Same layout as k_pack_tick
// (rm_transmit.hip) writes from the compact arrays.
template <int SCAN>
__global__ void __launch_bounds__(256)
k_pack_frames(const ModelDev m, const TickDev t, int n_new, HostView v, uint32_t *done_counter, uint32_t seq)
{
    __shared__ uint32_t s_off[scan_lds(SCAN)];
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_last;
    constexpr bool kRegScan = (SCAN == 3 || SCAN == 4);
    SmallCounts<scan_per(SCAN)> pre{};
    if (kRegScan) pre = small_scan_load<scan_per(SCAN)>(t.cursor, t.n_cnt);
    const uint32_t total = kRegScan ? small_scan(pre, t.n_cnt, s_off, s_wave, nullptr, nullptr)
                                    : block_scan_counts(t.cursor, t.n_cnt, s_off, s_wave, nullptr, nullptr);
    const uint32_t dropped = (total > t.cap || t.stage_count[1] != 0u) ? 1u : 0u;
    const uint32_t room = dropped ? 0u : v.links;
    const bool draws_possible = (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_N2N || m.kind == RM_MODEL_LOGDIST);
    const uint32_t np = min(uint32_t(max(n_new, 0)), v.packets);
    // The copy is laid out by its OUTPUT: a workgroup takes 256 consecutive records of the compact arrays at a time, every
    // thread finds its record's frame in the scanned offsets (a binary search in LDS) and its place in that frame's segment.
    // Every store instruction of a wave then writes 64 consecutive records -- whole, aligned lines of the host's block, from
    // ONE workgroup (one XCD's L2) -- where a wave per frame wrote the frame's 44 records wherever they fell: two frames, two
    // waves, often two XCDs shared most lines, and every shared line crossed the link as two partial writes.
    {
        const uint32_t limit = min(total, room);
        const int n_cnt = t.n_cnt;
        for (uint32_t o0 = blockIdx.x * 256u; o0 < limit; o0 += gridDim.x * 256u) { // block-uniform
            const uint32_t o = o0 + threadIdx.x;
            if (o >= limit) continue;
            int lo = 0, hi = n_cnt; // the last slot whose offset is <= o (slots without links share their successor's offset)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_off[mid] <= o) lo = mid;
                else hi = mid;
            }
            const uint32_t src = t.seg_off[lo] + (o - s_off[lo]);
            v.dst[o] = t.a_dst[src];
            if (v.rssi) v.rssi[o] = t.a_rssi[src]; // (nullptr: the packet's transmit power, once per packet below)
            v.verdict[o] = t.a_verdict[src];
            if (v.sinr != nullptr && t.out_sinr != nullptr) v.sinr[o] = t.a_sinr[src]; // (the SINR extension only)
        }
        for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < np; q += gridDim.x * 256u) {
            v.pkt_offset[q] = s_off[int(q) + t.shift];
            const rm_tx_record txq = t.tx[t.first_new + int(q)];
            v.pkt_interference[q] = (draws_possible && tx_success(m, txq) <= 0.0) ? 1 : 0;
            if (!v.rssi) v.pkt_rssi[q] = txq.txpower;
            if (q + 1 == np) v.pkt_offset[np] = total;
        }
    }
    if (np == 0 && blockIdx.x == 0 && threadIdx.x == 0) v.pkt_offset[0] = 0u;
    // Ordinary stores + one release fence per workgroup: the stores gather in this XCD's L2 and the fence writes them out in
    // bursts (33 GB/s on the measured box).  Write-through stores at system scope -- what k_ev_apply uses for its scattered
    // deliveries, with far more waves in flight -- were measured here too: each is acknowledged from the far end of the link,
    // 71 us per tick instead of 45; drained ordinary stores without the fence arrive after the sequence number (tests fail).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains its stores, ...
    __syncthreads();                                  // ... the workgroup meets, one lane releases them to the host
    if (threadIdx.x == 0) {
        __threadfence_system();
        s_last = (atomicAdd(done_counter, 1u) == gridDim.x - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        *done_counter = 0u;
        v.hdr->stored = dropped ? 0u : min(total, v.links);
        v.hdr->dropped = dropped;
        v.hdr->total = total;
        v.hdr->span_flag = 0u;
        v.hdr->n_packets = uint32_t(max(n_new, 0));
        __threadfence_system();
        __hip_atomic_store(&v.hdr->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the largest per-frame segment the A records (capacity `cap`) allow for n_cnt frame slots, 0 = not this path
int frame_tick_segment(const TickDev &t, const LaunchCfg &cfg, const ModelDev &m)
{
    const bool geometric = (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_UDGM_CONST || m.kind == RM_MODEL_LOGDIST);
    const bool sinr = m.kind == RM_MODEL_LOGDIST && (m.flags & RM_LD_SINR);
    const char *e_sf = getenv("RM_SINR_FRAMES"); // 0: the SINR medium's lone ticks through the sweep kernels (read per tick: tests switch it)
    const bool no_sinr_frames = e_sf && atoi(e_sf) == 0;
    if (sinr && !t.air_scan && (no_sinr_frames || t.air.pool == nullptr || t.first_eval != t.first_new)) return 0; // (a rebuild sweeps old frames too)
    if (!geometric || !cfg.sorted || !cfg.bbox || cfg.f64_filter || t.use_matrix) return 0;
    if (t.n_cnt <= 0 || t.n_cnt > kFusedScanMax || t.n_rx <= 0 || t.n_active <= t.first_new) return 0;
    // half of the records for the fixed segments, the rest for frames that outgrow theirs
    const uint32_t per = (t.cap / 2u) / uint32_t(t.n_cnt);
    if (per < 64u) return 0;
    return int(per < uint32_t(kFrameSegMax) ? (per / 64u) * 64u : uint32_t(kFrameSegMax));
}

hipError_t launch_tick_frames(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t, const LaunchCfg &cfg, int seg_len,
                              const ScanDev *scan)
{
    dim3 grid(t.n_cnt);
    const dim3 block(256);
    const int n_groups = cdiv(nd.n_rx, kGroup);
    const char *e_xcd = getenv("RM_TICK_XCD_MAP"); // 0: workgroup b takes slot b (read per tick: tests and experiments switch it)
    const int xcd_map = (e_xcd && atoi(e_xcd) == 0) ? 0 : 1;
    static const int flat_max = [] {
        const char *e = getenv("RM_FR_FLAT_MAX");
        return e ? max(0, min(kFrGroups * 2, atoi(e))) : kFrFlatGroups;
    }();
#define RM_FR2(M_, MODEL, ST, SH)                                                                                      \
    do {                                                                                                               \
        if (n_groups <= flat_max) RM_KLAUNCH((k_tick_frames<MODEL, ST, SH, true>), grid, block, 0, s, nd, M_, t, seg_len, xcd_map); \
        else RM_KLAUNCH((k_tick_frames<MODEL, ST, SH, false>), grid, block, 0, s, nd, M_, t, seg_len, xcd_map);         \
    } while (0)
#define RM_FR3(M_, MODEL, SH)                                                                                          \
    do {                                                                                                               \
        if (cfg.stochastic) RM_FR2(M_, MODEL, true, SH);                                                               \
        else RM_FR2(M_, MODEL, false, SH);                                                                             \
    } while (0)
#define RM_FR(MODEL, SH) RM_FR3(m, MODEL, SH)
    static const bool no_shadow = getenv("RM_FR_NO_SHADOW") != nullptr;
    if (m.kind == RM_MODEL_LOGDIST && (m.flags & RM_LD_SINR) && t.air_scan) {
        // the tick by scan: the heard links as the medium without SINR finds them (cut-off at the sensitivity, nothing
        // left per receiver), then the interference sums from the frames on the air themselves (rm_airscan.hip)
        if (scan == nullptr) return hipErrorInvalidValue;
        ModelDev ms = m;
        ms.ld_level = m.ld_sens;
        grid.x += unsigned(cdiv(t.n_active, 256)); // (workgroups that index the frames on the air)
        const bool sh = cfg.shadow && m.shadow_tbl && !no_shadow;
        const bool flat = n_groups <= flat_max;
#define RM_FSC(ST, SH, FL) RM_KLAUNCH((k_tick_frames_scan<ST, SH, FL>), grid, block, 0, s, nd, ms, t, seg_len, *scan)
        if (cfg.stochastic) {
            if (sh) { if (flat) RM_FSC(true, true, true); else RM_FSC(true, true, false); }
            else { if (flat) RM_FSC(true, false, true); else RM_FSC(true, false, false); }
        } else {
            if (sh) { if (flat) RM_FSC(false, true, true); else RM_FSC(false, true, false); }
            else { if (flat) RM_FSC(false, false, true); else RM_FSC(false, false, false); }
        }
#undef RM_FSC
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        return launch_sinr_scan(s, nd, m, t, *scan, cfg);
    }
    if (m.kind == RM_MODEL_LOGDIST && (m.flags & RM_LD_SINR)) {
        const bool sh = cfg.shadow && m.shadow_tbl && !no_shadow;
        const bool flat = n_groups <= flat_max;
#define RM_FRS(ST, SH, FL) RM_KLAUNCH((k_tick_frames_sinr<ST, SH, FL>), grid, block, 0, s, nd, m, t, seg_len)
        if (cfg.stochastic) {
            if (sh) { if (flat) RM_FRS(true, true, true); else RM_FRS(true, true, false); }
            else { if (flat) RM_FRS(true, false, true); else RM_FRS(true, false, false); }
        } else {
            if (sh) { if (flat) RM_FRS(false, true, true); else RM_FRS(false, true, false); }
            else { if (flat) RM_FRS(false, false, true); else RM_FRS(false, false, false); }
        }
#undef RM_FRS
        const int n_new = t.n_active - t.first_new;
        RM_KLAUNCH(k_sinr_frames, dim3(max(1, min(4096, cdiv(n_new, 4)))), dim3(256), 0, s, nd, m, t);
        return hipGetLastError();
    }
    switch (m.kind) {
    case RM_MODEL_UDGM: RM_FR(RM_MODEL_UDGM, false); break;
    case RM_MODEL_UDGM_CONST: RM_FR(RM_MODEL_UDGM_CONST, false); break;
    case RM_MODEL_LOGDIST:
        if (cfg.shadow && m.shadow_tbl && !no_shadow) RM_FR(RM_MODEL_LOGDIST, true);
        else RM_FR(RM_MODEL_LOGDIST, false);
        break;
    default: return hipErrorInvalidValue;
    }
#undef RM_FR
#undef RM_FR3
#undef RM_FR2
    return hipGetLastError();
}

// is the per-frame candidate kernel the better filter for this tick?  (a lone tick over a table the sweep would tile)
bool frames_cand_applies(const TickDev &t, const LaunchCfg &cfg)
{
    static const bool off = [] {
        const char *e = getenv("RM_FRAMES_CAND");
        return e && atoi(e) == 0;
    }();
    return !off && t.air.pool != nullptr && t.filter_mode == kFilterWg && cfg.sorted && cfg.bbox && !cfg.f64_filter && !t.use_matrix &&
           !t.reset_heads;
}

hipError_t launch_frames_cand(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t, const LaunchCfg &cfg)
{
    const int n_eval = t.n_active - t.first_eval;
    if (n_eval <= 0 || t.n_rx <= 0) return hipSuccess;
    if (cfg.shadow && m.shadow_tbl) RM_KLAUNCH((k_frames_cand<true>), dim3(n_eval), dim3(256), 0, s, nd, m, t);
    else RM_KLAUNCH((k_frames_cand<false>), dim3(n_eval), dim3(256), 0, s, nd, m, t);
    return hipGetLastError();
}

hipError_t launch_tick_frames_batch(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev *ticks, int n,
                                    const TickDev *dev_ticks, const LaunchCfg &cfg, int seg_len)
{
    int max_cnt = 0;
    for (int i = 0; i < n; ++i) max_cnt = max(max_cnt, ticks[i].n_cnt);
    const dim3 grid(max_cnt, 1, n), block(256);
    const int n_groups = cdiv(nd.n_rx, kGroup);
    const bool flat = n_groups <= kFrFlatGroups;
#define RM_FRB(MODEL, SH)                                                                                              \
    do {                                                                                                               \
        if (flat) RM_KLAUNCH((k_tick_frames_batch<MODEL, false, SH, true>), grid, block, 0, s, nd, m, dev_ticks, seg_len); \
        else RM_KLAUNCH((k_tick_frames_batch<MODEL, false, SH, false>), grid, block, 0, s, nd, m, dev_ticks, seg_len);     \
    } while (0)
    switch (m.kind) {
    case RM_MODEL_UDGM: RM_FRB(RM_MODEL_UDGM, false); break;
    case RM_MODEL_UDGM_CONST: RM_FRB(RM_MODEL_UDGM_CONST, false); break;
    case RM_MODEL_LOGDIST:
        if (cfg.shadow && m.shadow_tbl) RM_FRB(RM_MODEL_LOGDIST, true);
        else RM_FRB(RM_MODEL_LOGDIST, false);
        break;
    default: return hipErrorInvalidValue;
    }
#undef RM_FRB
    return hipGetLastError();
}

hipError_t launch_pack_frames(hipStream_t s, const ModelDev &m, const TickDev &t, int n_new, const HostView &v, uint32_t *done_counter,
                              uint32_t seq)
{
    static const int wgs = [] { // (developer knob: workgroups of the pack, a multiple of the eight XCDs)
        const char *e = getenv("RM_PACK_WGS");
        return e ? max(8, min(1024, atoi(e) & ~7)) : 128;
    }();
    const dim3 grid(wgs), block(256);
    const int scan = scan_variant(t.n_cnt);
    if (scan == 3) RM_KLAUNCH(k_pack_frames<3>, grid, block, 0, s, m, t, n_new, v, done_counter, seq);
    else if (scan == 4) RM_KLAUNCH(k_pack_frames<4>, grid, block, 0, s, m, t, n_new, v, done_counter, seq);
    else RM_KLAUNCH(k_pack_frames<1>, grid, block, 0, s, m, t, n_new, v, done_counter, seq);
    return hipGetLastError();
}

} // namespace rm
