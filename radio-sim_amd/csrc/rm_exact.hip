// rm_exact.hip -- exact evaluation of the candidates, SINR, and the ordered scatter of unsorted tables
// (part of libradiomedium_hip.so; gfx950 only, -ffp-contract=off, no fast-math; overview at the top of rm_engine.h)
#include "rm_device.hpp"

#include <stdlib.h>

namespace rm {

// grid-filter ticks have no k_tick_prep: the sticky overflow flag is looked at here
__global__ void __launch_bounds__(64) k_air_begin(TickDev t)
{
    if (threadIdx.x == 0 && t.air.bad[0]) t.stage_count[1] = 1u;
}

// One lane per candidate link (full waves): the reference's fp64 arithmetic.
// SEG 0: unsorted table, heard links are counted per (frame, slab) cell (ordered scatter later).
// SEG 1/2: sorted table, heard links go straight into the frame's segment of the A records (any
// order inside it); the segment offsets are the scan of the per-frame candidate counts, redone in
// LDS by every workgroup (1; 3 / 4 = the same for at most kSmallScan / kMediumScan frames) or read from k_scan_counts' output (2).
// PACKED: a 1-D grid whose workgroups walk the started 256-entry chunks of all shards (no workgroup
// without entries); otherwise blockIdx.y is the shard and blockIdx.x strides over its entries.
template <int MODEL, bool SINR, bool STOCH, int SEG, bool PACKED = false>
RM_D void exact_body(const NodesDev &nd, const ModelDev &m, const TickDev &t)
{
    __shared__ uint32_t s_seg[scan_lds(SEG)];
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_cs[PACKED ? kShards + 1 : 1]; // exclusive scan of the shards' chunk counts
    __shared__ uint32_t s_sn[PACKED ? kShards : 1];     // entries per shard
    const bool publisher = (blockIdx.x == 0 && blockIdx.y == 0);
    bool scanned = false; // the scan is only needed by the scatter: it runs after the first evaluation
    // A tick whose candidates did not fit the link capacity is reported as RM_ERR_CAPACITY; its shards
    // hold gaps where runs were dropped, so none of its entries is touched (no link of it is reported).
    const bool dropped = t.stage_count[1] != 0u;
    // (shards beyond the tick's mask are never appended to -- a batch uses 64 of the 256, a receiver partition 8: their counters
    // are not even read, each is a cache line of its own)
    const uint32_t sh_own = PACKED ? threadIdx.x : blockIdx.y; // kBlock == kShards
    const uint32_t n_own = (dropped || sh_own > t.shard_mask) ? 0u : min(t.shard_count[sh_own * kShardStride], t.seg_cap);
    constexpr bool kRegScan = (SEG == 3 || SEG == 4);
    const bool acc_mode = SINR && t.acc_lo != nullptr; // block-uniform: interference summed per receiver, no lists (TickDev::acc_lo)
    if (acc_mode && publisher && t.src_air_us > 0) {
        // half duplex: every frame's source that is a receiver here is on the air itself (the pre-pass zeroed the words; the
        // sums' additions never reach bit 63)
        const int n_eval = t.n_active - t.first_eval;
        for (int e = int(threadIdx.x); e < n_eval; e += int(blockDim.x)) {
            const int pos = engine_pos(nd, t.tx[t.first_eval + e].src);
            if (pos >= 0) atomicOr(&t.acc_hi[pos], 1ull << 63);
        }
    }
    SmallCounts<scan_per(SEG)> pre{};
    if (kRegScan && !PACKED && (blockIdx.x == 0 || blockIdx.x * blockDim.x < n_own)) pre = small_scan_load<scan_per(SEG)>(t.cand_tot, t.n_cnt);
    const uint32_t stride = gridDim.x * blockDim.x;
    const int per_slab = kGroup * t.rpt;
    const int lane = threadIdx.x & 63;
    uint32_t n_chunks = 0;
    if (PACKED) {
        const uint32_t mine = (n_own + 255u) >> 8;
        uint32_t inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) s_wave[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t run = inc - mine;
        for (int w = 0; w < int(threadIdx.x >> 6); ++w) run += s_wave[w];
        s_cs[threadIdx.x] = run;
        s_sn[threadIdx.x] = n_own;
        n_chunks = uniform_u(s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3]);
        if (threadIdx.x == 0) s_cs[PACKED ? kShards : 0] = n_chunks;
        __syncthreads();
        // a workgroup without a chunk leaves before touching the per-frame counts (a receiver
        // partition has few candidates per tick: most of the grid); the publisher stays for seg_off
        if (blockIdx.x >= n_chunks && !publisher) return;
        if (kRegScan) pre = small_scan_load<scan_per(SEG)>(t.cand_tot, t.n_cnt);
    }
    // an entry's inputs, requested ahead of its evaluation (PACKED walks request the next chunk's before
    // they evaluate the current one: the gathers of one chunk fly under the arithmetic of the other)
    struct Loaded {
        uint32_t idx;
        bool valid;
        int erel, pos;
        rm_tx_record tx;
        RxRecord rx;
    };
    auto load = [&](const uint32_t shard, const uint32_t it, const uint32_t n) -> Loaded {
        Loaded L;
        const uint32_t i = it + threadIdx.x;
        L.valid = i < n;
        L.idx = shard * t.seg_cap + i;
        L.erel = 0;
        L.pos = 0;
        if (L.valid) {
            L.erel = t.st_pkt[L.idx];
            L.pos = t.st_dst[L.idx];
            L.tx = t.tx[t.first_eval + L.erel];
            if (SEG == 0 || nd.rec32 == nullptr) {
                L.rx = nd.rec[L.pos];
            } else { // sorted table: the 32-byte record; channel and radio state were tested by the sweep
                const RxCompact c = nd.rec32[L.pos];
                L.rx.x = c.x;
                L.rx.y = c.y;
                L.rx.z = c.z;
                L.rx.orig = c.orig;
                L.rx.int_id = 0;
                L.rx.channel = L.tx.channel;
                L.rx.enabled = 1;
                L.rx.rxprob = (c.flags & 1u) ? nd.rxprob[L.pos] : 1.0;
            }
        }
        return L;
    };
    auto finish = [&](const Loaded &L) {
        const bool valid = L.valid;
        const uint32_t idx = L.idx;
        bool wanted = false;
        int slot = -1, key = -1;
        uint8_t fl = 0;
        double rssi = 0.0, prob = 1.0, lin = 0.0;
        int orig = 0;
        if (valid) {
            const int erel = L.erel;
            const int pos = L.pos;
            const rm_tx_record &tx = L.tx;
            const bool is_new = (t.first_eval + erel) >= t.first_new;
            const RxRecord &rx_ = L.rx;
            const LinkEval ev = eval_link<MODEL, SINR>(m, nd, tx, rx_, is_new);
            fl = ev.append ? ev.flags : uint8_t(0);
            if (ev.append && MODEL != RM_MODEL_NULL && MODEL != RM_MODEL_UDGM_CONST && tx_success(m, tx) <= 0.0) fl |= kFlagTxDead;
            if (SEG == 0 || (SINR && !acc_mode)) t.st_flags[idx] = fl; // read by the ordered scatter / the SINR pass only
            if (ev.append && acc_mode) {
                // the interferer's power joins its receiver's sum (Q80: exact, any order); nothing of the entry is kept
                if ((fl & kFlagInterferer) && t.src_air_us > 0) {
                    const U128 v = q80_from_double(ev.lin);
                    if ((v.lo | v.hi) != 0ull) {
                        const unsigned long long old = atomicAdd(&t.acc_lo[pos], (unsigned long long)v.lo);
                        const unsigned long long carry = (old + v.lo < old) ? 1ull : 0ull; // (the low words' running sum is exact mod 2^64: so is the carry count)
                        if (v.hi + carry) atomicAdd(&t.acc_hi[pos], (unsigned long long)(v.hi + carry));
                    }
                }
                orig = rx_.orig;
                rssi = ev.aux;
                prob = rx_.rxprob;
                wanted = ev.wanted;
            } else if (ev.append) {
                orig = rx_.orig;
                if (MODEL == RM_MODEL_LOGDIST) {
                    rssi = ev.aux;
                    prob = rx_.rxprob;
                } else {
                    rssi = tx.txpower; // reference media hand the packet's transmit power through
                    prob = (MODEL == RM_MODEL_UDGM || MODEL == RM_MODEL_N2N) ? ev.aux : 1.0;
                }
                if (SEG == 0 || SINR) { // the ordered scatter / the SINR pass read these from the entry
                    t.st_orig[idx] = orig;
                    t.st_aux[idx] = rssi;
                    t.st_prob[idx] = prob;
                }
                if (SINR) {
                    lin = ev.lin;
                    if (t.air.pool == nullptr) { // the tick's own lists
                        t.st_lin[idx] = ev.lin;
                        t.st_next[idx] = atomicExch(&t.head[pos], int(idx));
                    }
                }
                wanted = ev.wanted;
            }
            slot = erel - t.cnt_base;
            key = (SEG == 0) ? int((size_t(slot >> 6) * t.n_slabs + pos / per_slab) * 64 + (slot & 63)) : slot;
        }
        if (SINR && t.air.pool != nullptr) { // block-uniform: the interferers join the lists that outlive the tick
            const bool want = valid && (fl & kFlagInterferer);
            const int aidx = air_alloc(t, want, air_sub(t));
            if (valid && fl) t.st_next[idx] = aidx; // k_sinr skips the link's own entry
            if (want) air_link(t, aidx, L.pos, L.tx.start_us, L.tx.air_us, lin, kAirInterferer);
        }
        if (SEG == 1 && !scanned) { // block-uniform
            block_scan_counts(t.cand_tot, t.n_cnt, s_seg, s_wave, publisher ? t.seg_off : nullptr, nullptr);
            scanned = true;
        }
        if (kRegScan && !scanned) {
            small_scan(pre, t.n_cnt, s_seg, s_wave, publisher ? t.seg_off : nullptr, nullptr);
            scanned = true;
        }
        // one atomic per run of same-frame (same-cell) entries
        const RunInfo ri = run_prefix(key, wanted, lane);
        if (SEG == 0) {
            if (valid && lane == ri.start && ri.total) atomicAdd(&t.cnt[key], ri.total);
        } else {
            uint32_t base = 0;
            if (valid && lane == ri.start && ri.total) base = atomicAdd(&t.cursor[slot], ri.total);
            base = __shfl(base, ri.start);
            if (acc_mode) {
                // the heard links' places in the A records, one list per tick: k_sinr_acc takes them with full lanes once every
                // receiver's sum is complete (one atomic per wave)
                const uint64_t wm = ballot64(wanted);
                if (wm) {
                    uint32_t hb = 0;
                    if (lane == __ffsll((long long)wm) - 1) hb = atomicAdd(&t.stage_count[0], uint32_t(__popcll(wm)));
                    hb = uint32_t(__shfl(int(hb), __ffsll((long long)wm) - 1));
                    if (wanted) t.st_next[hb + lane_prefix(wm)] = int(((SEG == 1 || kRegScan) ? s_seg[slot] : t.seg_off[slot]) + base + ri.before);
                }
            }
            if (wanted) {
                const uint32_t o = ((SEG == 1 || kRegScan) ? s_seg[slot] : t.seg_off[slot]) + base + ri.before;
                t.a_dst[o] = orig;
                t.a_rssi[o] = rssi;
                if (SINR) t.a_e[o] = acc_mode ? L.pos : int(idx); // (the sums are looked up by receiver, the lists by entry)
                if (STOCH) {
                    t.a_prob[o] = prob;
                    t.a_verdict[o] = uint8_t(0); // pending: k_apply_draws decides
                } else {
                    t.a_verdict[o] = (fl & kFlagTxDead) ? uint8_t(RM_INTERFERED) : uint8_t(RM_DELIVERED);
                }
            }
        }
    };
    if (PACKED) {
        auto load_chunk = [&](const uint32_t u) -> Loaded {
            uint32_t lo = 0, hi = kShards; // the shard whose chunk range holds u: s_cs[lo] <= u < s_cs[lo + 1]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (uniform_u(s_cs[mid]) <= u) lo = mid; else hi = mid;
            }
            return load(lo, (u - uniform_u(s_cs[lo])) << 8, uniform_u(s_sn[lo]));
        };
        uint32_t u = blockIdx.x;
        if (u < n_chunks) {
            Loaded cur = load_chunk(u);
            for (;;) { // block-uniform
                const uint32_t un = u + gridDim.x;
                if (un < n_chunks) {
                    const Loaded nxt = load_chunk(un); // requested before the current chunk is evaluated
                    finish(cur);
                    cur = nxt;
                    u = un;
                } else {
                    finish(cur);
                    break;
                }
            }
        }
    } else {
        uint32_t it = blockIdx.x * blockDim.x; // block-uniform trip count; pipelined like the packed walk
        if (it < n_own) {
            Loaded cur = load(blockIdx.y, it, n_own);
            for (;;) {
                const uint32_t nx = it + stride;
                if (nx < n_own) {
                    const Loaded nxt = load(blockIdx.y, nx, n_own);
                    finish(cur);
                    cur = nxt;
                    it = nx;
                } else {
                    finish(cur);
                    break;
                }
            }
        }
    }
    if (SEG == 1 && !scanned && publisher) // seg_off is published even if this shard was empty
        block_scan_counts(t.cand_tot, t.n_cnt, s_seg, s_wave, t.seg_off, nullptr);
    if (kRegScan && !scanned && publisher) small_scan(pre, t.n_cnt, s_seg, s_wave, t.seg_off, nullptr);
}

template <int MODEL, bool SINR, bool STOCH, int SEG>
__global__ void __launch_bounds__(256) k_exact(const NodesDev nd, const ModelDev m, const TickDev t)
{
    exact_body<MODEL, SINR, STOCH, SEG>(nd, m, t);
}

// (the variant with the scan of up to 2048 per-frame counts in registers -- what a rank's batch of gathered ticks takes, by the
// HOST's frame count -- came to 129-133 VGPRs: three workgroups per CU instead of four for a stage that is bound by round trips
// times resident waves; held to 128)
template <int MODEL, bool STOCH, int SCAN, bool SINR = false>
__global__ void __launch_bounds__(256, (SCAN == 4 && !SINR) ? 4 : 1) k_exact_batch(const NodesDev nd, const ModelDev m, const TickDev *__restrict__ ticks)
{
    exact_body<MODEL, SINR, STOCH, SCAN, true>(nd, m, ticks[blockIdx.z]);
}

// half duplex (SINR mode): every frame on the air leaves a SELF entry in its source's list
RM_D void self_entries_body(const NodesDev &nd, const TickDev &t)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_eval = t.n_active - t.first_eval;
    if (t.air.pool != nullptr) { // block-uniform
        bool want = false;
        int pos = 0;
        rm_tx_record tx{};
        if (e < n_eval) {
            tx = t.tx[t.first_eval + e];
            pos = engine_pos(nd, tx.src);
            want = pos >= 0;
        }
        const int aidx = air_alloc(t, want, air_sub(t));
        if (want) air_link(t, aidx, pos, tx.start_us, tx.air_us, 0.0, kAirSelf);
        return;
    }
    if (e >= n_eval) return;
    const int src = t.tx[t.first_eval + e].src;
    const int pos = engine_pos(nd, src);
    if (pos < 0) return;
    const uint32_t shard = (blockIdx.x * 4 + (threadIdx.x >> 6)) & t.shard_mask;
    const uint32_t local = atomicAdd(&t.shard_count[shard * kShardStride], 1u);
    if (local >= t.seg_cap) {
        t.stage_count[1] = 1u;
        return;
    }
    const uint32_t idx = shard * t.seg_cap + local;
    t.st_pkt[idx] = e;
    t.st_dst[idx] = pos;
    t.st_blk[idx] = idx;
    t.st_flags[idx] = kFlagSelf;
    t.st_aux[idx] = 0.0;
    t.st_prob[idx] = 0.0;
    t.st_orig[idx] = src;
    t.st_lin[idx] = 0.0;
    t.st_next[idx] = atomicExch(&t.head[pos], int(idx));
}

__global__ void __launch_bounds__(256) k_self_entries(NodesDev nd, TickDev t) { self_entries_body(nd, t); }

__global__ void __launch_bounds__(256) k_self_entries_batch(const NodesDev nd, const TickDev *__restrict__ ticks)
{
    self_entries_body(nd, ticks[blockIdx.z]);
}

// off[cell] = heard links of the same frame in lower slabs; slot_tot[slot] = heard links of the
// frame.  One 1024-thread workgroup per tile of 64 frames: lane = frame, each wave owns a
// contiguous range of slabs (coalesced 256-byte rows).
__global__ void __launch_bounds__(1024) k_cell_off(TickDev t)
{
    __shared__ uint32_t s_part[16][64];
    const int cc = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (t.n_slabs + 15) / 16;
    const int s0 = min(t.n_slabs, wave * per), s1 = min(t.n_slabs, s0 + per);
    const size_t row0 = size_t(cc) * t.n_slabs;
    uint32_t sum = 0;
    for (int s = s0; s < s1; ++s) sum += t.cnt[(row0 + s) * 64 + lane];
    s_part[wave][lane] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t x = s_part[w][lane];
        if (w < wave) run += x;
        total += x;
    }
    for (int s = s0; s < s1; ++s) {
        const size_t c = (row0 + s) * 64 + lane;
        t.off[c] = run;
        run += t.cnt[c];
    }
    if (wave == 0) t.slot_tot[cc * 64 + lane] = total;
}

// slot_off = exclusive scan over frames of their heard-link totals; publishes the link count
__global__ void __launch_bounds__(1024) k_slot_scan(TickDev t)
{
    __shared__ uint32_t s_wave[16];
    uint32_t carry = 0;
    uint32_t vmax = 0;
    for (int base = 0; base < t.n_cnt; base += 1024) {
        const int slot = base + threadIdx.x;
        const uint32_t v = (slot < t.n_cnt) ? t.slot_tot[slot] : 0u;
        vmax = max(vmax, v);
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_1024(v, s_wave, total);
        if (slot < t.n_cnt) t.slot_off[slot] = carry + ex;
        carry += total;
    }
    for (int d = 32; d >= 1; d >>= 1) vmax = max(vmax, uint32_t(__shfl_xor(int(vmax), d)));
    if ((threadIdx.x & 63) == 0 && vmax) atomicMax(&t.out_count[3], vmax);
    if (threadIdx.x == 0) {
        t.slot_off[t.n_cnt] = carry;
        t.out_count[0] = carry < t.cap ? carry : t.cap;
        t.out_count[1] = (carry > t.cap || t.stage_count[1] != 0u) ? 1u : 0u;
        t.out_count[2] = carry;
    }
}

// ============================================================================ SINR (O(heard links))

RM_D bool frames_overlap(const rm_tx_record &w, const rm_tx_record &k)
{
    return k.start_us < w.start_us + w.air_us && k.start_us + k.air_us > w.start_us;
}

// one thread per link entry that is a heard link of a new frame: walk the receiver's list,
// sum the co-channel, time-overlapping interferers exactly (Q80), apply capture + half duplex
RM_D void sinr_body(const ModelDev &m, const TickDev &t)
{
    const uint32_t n = t.stage_count[1] ? 0u : min(t.shard_count[blockIdx.y * kShardStride], t.seg_cap); // nothing of a dropped tick
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t e = blockIdx.y * t.seg_cap + i;
        if (!(t.st_flags[e] & kFlagHeardNew)) continue;
        const int pos = t.st_dst[e];
        const rm_tx_record w = t.tx[t.first_eval + t.st_pkt[e]];
        U128 acc = {0, 0};
        bool half_duplex = false;
        if (t.air.pool != nullptr) { // the lists that live across ticks (unsorted tables: sorted ones do this in k_reorder)
            const SinrOut so = air_sinr(m, t, pos, t.st_next[e], w.start_us, w.air_us, t.st_aux[e]);
            t.st_sinr[e] = so.sinr;
            t.st_coll[e] = so.collided ? 1 : 0;
            continue;
        }
        for (int idx = t.head[pos]; idx >= 0; idx = t.st_next[idx]) {
            if (uint32_t(idx) == e) continue;
            const rm_tx_record k = t.tx[t.first_eval + t.st_pkt[idx]];
            if (!frames_overlap(w, k)) continue;
            const uint8_t fl = t.st_flags[idx];
            if (fl & kFlagSelf) {
                half_duplex = true;
                continue;
            }
            if (!(fl & kFlagInterferer)) continue;
            acc = u128_add(acc, q80_from_double(t.st_lin[idx]));
        }
        const double I = q80_to_double(acc);
        const double denom = I + m.ld_noise_lin;
        const double sinr = t.st_aux[e] - 10.0 * det_log10(denom);
        t.st_sinr[e] = sinr;
        t.st_coll[e] = (half_duplex || !(sinr >= m.ld_capture)) ? 1 : 0;
    }
}

__global__ void __launch_bounds__(256) k_sinr(ModelDev m, TickDev t)
{
    sinr_body(m, t);
    if (t.air.pool != nullptr && blockIdx.x == 0 && blockIdx.y == 0) air_end(t);
}

__global__ void __launch_bounds__(256) k_sinr_batch(const ModelDev m, const TickDev *__restrict__ ticks)
{
    sinr_body(m, ticks[blockIdx.z]);
}

// Interference summed per receiver (TickDev::acc_lo): one lane per heard link of the tick, full lanes, once the exact stage has
// completed every receiver's sum -- the receiver's sum less this link's own power (it joined the sum iff it reaches the
// interference floor: eval_link's rule and its arithmetic), then air_sinr's formula; sinr and a collision's verdict go into the
// link's A record, where the reorder stage finds them.  (Done inside the reorder stage's per-frame waves it ran three lanes
// of 64 through the logarithm: 184 us per 32 ticks of configs[3] for that stage alone.)
__global__ void __launch_bounds__(256) k_sinr_acc_batch(const ModelDev m, const TickDev *__restrict__ ticks)
{
    const TickDev &t = ticks[blockIdx.z];
    if (t.acc_lo == nullptr || t.stage_count[1] != 0u) return; // (nothing of a dropped tick)
    const uint32_t n = min(t.stage_count[0], t.cap);
    const bool on_air = t.src_air_us > 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t o = uint32_t(t.st_next[i]);
        const int pos = t.a_e[o];
        const double rssi = t.a_rssi[o];
        U128 acc;
        acc.lo = t.acc_lo[pos];
        const unsigned long long hi = t.acc_hi[pos];
        acc.hi = hi & ~(1ull << 63);
        if (rssi >= m.ld_ifloor && on_air) acc = u128_sub(acc, q80_from_double(det_pow10(rssi / 10.0)));
        const double sinr = rssi - 10.0 * det_log10(q80_to_double(acc) + m.ld_noise_lin);
        t.a_sinr[o] = sinr;
        if (((hi >> 63) != 0ull && on_air) || !(sinr >= m.ld_capture)) t.a_verdict[o] = uint8_t(RM_INTERFERED);
    }
}

// ============================================================================ ordered scatter

// frames beyond what the fused scans hold: seg_off / slot_off from a one-workgroup scan kernel
__global__ void __launch_bounds__(1024) k_scan_counts(const uint32_t *cnt, uint32_t *off, int n)
{
    __shared__ uint32_t s_wave[16];
    uint32_t carry = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = (i < n) ? cnt[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_1024(v, s_wave, total);
        if (i < n) off[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) off[n] = carry;
}

// Unsorted tables (Null / N2N media): link entry -> cell offset + rank inside the (frame, slab)
// block -- directly the final (packet, node index) order; verdict for everything that needs no draw.
template <bool STOCH>
__global__ void __launch_bounds__(256) k_finalize(ModelDev m, TickDev t)
{
    const uint32_t n = t.stage_count[1] ? 0u : min(t.shard_count[blockIdx.y * kShardStride], t.seg_cap); // nothing of a dropped tick
    const uint32_t stride = gridDim.x * blockDim.x;
    const int per_slab = kGroup * t.rpt;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t e = blockIdx.y * t.seg_cap + i;
        const uint8_t fl = t.st_flags[e];
        if (!(fl & kFlagHeardNew)) continue;
        const int slot = t.st_pkt[e] - t.cnt_base;
        const int slab = t.st_dst[e] / per_slab;
        uint32_t rank = 0;
        for (uint32_t k = t.st_blk[e]; k < e; ++k) rank += (t.st_flags[k] & kFlagHeardNew) ? 1u : 0u;
        const uint32_t o = t.slot_off[slot] + t.off[(size_t(slot >> 6) * t.n_slabs + slab) * 64 + (slot & 63)] + rank;
        if (o >= t.cap) continue;
        t.out_pkt[o] = slot - t.shift;
        t.out_dst[o] = t.st_orig[e];
        t.out_rssi[o] = t.st_aux[e];
        const bool sinr = (m.kind == RM_MODEL_LOGDIST) && (m.flags & RM_LD_SINR);
        if (t.out_sinr) t.out_sinr[o] = sinr ? t.st_sinr[e] : 0.0;
        const bool collided = sinr && t.st_coll[e];
        if (STOCH) {
            t.out_prob[o] = t.st_prob[e];
            t.out_verdict[o] = collided ? uint8_t(RM_INTERFERED) : uint8_t(0); // 0 = pending
        } else {
            t.out_verdict[o] = ((fl & kFlagTxDead) || collided) ? uint8_t(RM_INTERFERED) : uint8_t(RM_DELIVERED);
        }
    }
}

// ============================================================================ launchers

template <int MODEL, bool SINR>
static void launch_exact_m(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t, const LaunchCfg &cfg)
{
    const dim3 grid(4, kShards), block(256);
    const int seg = t.use_matrix ? 0 : scan_variant(t.n_cnt);
#define RM_EX(ST, SG) RM_KLAUNCH((k_exact<MODEL, SINR, ST, SG>), grid, block, 0, s, nd, m, t)
    if (cfg.stochastic) {
        if (seg == 0) RM_EX(true, 0); else if (seg == 1) RM_EX(true, 1); else if (seg == 3) RM_EX(true, 3); else if (seg == 4) RM_EX(true, 4); else RM_EX(true, 2);
    } else {
        if (seg == 0) RM_EX(false, 0); else if (seg == 1) RM_EX(false, 1); else if (seg == 3) RM_EX(false, 3); else if (seg == 4) RM_EX(false, 4); else RM_EX(false, 2);
    }
#undef RM_EX
}

hipError_t launch_exact(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t, const LaunchCfg &cfg)
{
    switch (m.kind) {
    case RM_MODEL_NULL: launch_exact_m<RM_MODEL_NULL, false>(s, nd, m, t, cfg); break;
    case RM_MODEL_UDGM: launch_exact_m<RM_MODEL_UDGM, false>(s, nd, m, t, cfg); break;
    case RM_MODEL_UDGM_CONST: launch_exact_m<RM_MODEL_UDGM_CONST, false>(s, nd, m, t, cfg); break;
    case RM_MODEL_N2N: launch_exact_m<RM_MODEL_N2N, false>(s, nd, m, t, cfg); break;
    case RM_MODEL_LOGDIST:
        if (m.flags & RM_LD_SINR) launch_exact_m<RM_MODEL_LOGDIST, true>(s, nd, m, t, cfg);
        else launch_exact_m<RM_MODEL_LOGDIST, false>(s, nd, m, t, cfg);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// sorted tables with more frames than the fused scans hold: seg_off before k_exact
hipError_t launch_seg_scan(hipStream_t s, const TickDev &t)
{
    if (!t.use_matrix && t.n_cnt > kFusedScanMax)
        RM_KLAUNCH(k_scan_counts, dim3(1), dim3(1024), 0, s, t.cand_tot, t.seg_off, t.n_cnt);
    return hipGetLastError();
}

hipError_t launch_air_begin(hipStream_t s, const TickDev &t)
{
    RM_KLAUNCH(k_air_begin, dim3(1), dim3(64), 0, s, t);
    return hipGetLastError();
}

hipError_t launch_self_entries(hipStream_t s, const NodesDev &nd, const TickDev &t)
{
    const int n_eval = t.n_active - t.first_eval;
    if (n_eval <= 0) return hipSuccess;
    RM_KLAUNCH(k_self_entries, dim3(cdiv(n_eval, 256)), dim3(256), 0, s, nd, t);
    return hipGetLastError();
}

// unsorted tables: cell offsets + frame scan; sorted tables with very many frames: slot_off
hipError_t launch_offsets(hipStream_t s, const TickDev &t)
{
    if (t.use_matrix) {
        if (t.n_cnt > 0 && t.n_slabs > 0) RM_KLAUNCH(k_cell_off, dim3(t.n_cnt / 64), dim3(1024), 0, s, t);
        RM_KLAUNCH(k_slot_scan, dim3(1), dim3(1024), 0, s, t);
    } else if (t.n_cnt > kFusedScanMax) {
        RM_KLAUNCH(k_scan_counts, dim3(1), dim3(1024), 0, s, t.cursor, t.slot_off, t.n_cnt);
    }
    return hipGetLastError();
}

hipError_t launch_sinr(hipStream_t s, const ModelDev &m, const TickDev &t)
{
    RM_KLAUNCH(k_sinr, dim3(4, kShards), dim3(256), 0, s, m, t);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_pkt_interference(ModelDev m, TickDev t)
{
    write_pkt_interference(m, t, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// unsorted tables only
hipError_t launch_finalize(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t,
                           const LaunchCfg &cfg)
{
    (void)nd;
    const dim3 grid(4, kShards), block(256);
    if (cfg.stochastic) {
        RM_KLAUNCH(k_finalize<true>, grid, block, 0, s, m, t);
    } else {
        RM_KLAUNCH(k_finalize<false>, grid, block, 0, s, m, t);
        RM_KLAUNCH(k_pkt_interference, dim3(8), dim3(256), 0, s, m, t);
    }
    return hipGetLastError();
}

// rm_batch_*, stage 1
hipError_t launch_exact_batch(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev *ticks, int n, const TickDev *b,
                              const LaunchCfg &cfg)
{
    const int scan = batch_scan_variant(ticks, n);
    // packed walk: a workgroup takes every gridDim.x-th started chunk of 256 entries and requests the next
    // chunk's records before it evaluates the current one, so it wants several chunks: about one
    // workgroup per CU and tick slot in flight (1024 in all), the ticks of the batch bring the rest
    // (hundreds of ticks per launch: fewer workgroups per tick, each with more chunks to pipeline -- one rank's share of an
    // 8-GPU tick at 512 ticks per launch: 8 -> 4 workgroups per tick 0.58 -> 0.55 us per tick)
    int gx = max(n >= 256 ? 4 : 8, min(int(kShards), 1024 / n));
    // (ticks of thousands of frames have hundreds of chunks each -- configs[3]: 5000 frames, ~590 chunks: 8 / 12 / 16 / 32 workgroups
    // per tick -> 12.1 / 11.3 / 11.7 / 13.1 us per tick with three contexts in flight; configs[2], 1000 frames: 8 / 16 -> 2.86 / 2.94)
    if (n < 256) {
        int max_eval = 0;
        for (int i = 0; i < n; ++i) max_eval = max(max_eval, ticks[i].n_active - ticks[i].first_eval);
        gx = max(gx, min(16, max_eval / 400));
    }
    if (const char *e = getenv("RM_EXACT_GRID")) gx = max(1, min(int(kShards), atoi(e)));
    const dim3 grid(gx, 1, n), block(256);
#define RM_EXB(MODEL)                                                                                                \
do {                                                                                                             \
    if (cfg.stochastic) {                                                                                        \
        if (scan == 3) RM_KLAUNCH((k_exact_batch<MODEL, true, 3>), grid, block, 0, s, nd, m, b);         \
        else if (scan == 4) RM_KLAUNCH((k_exact_batch<MODEL, true, 4>), grid, block, 0, s, nd, m, b);    \
        else RM_KLAUNCH((k_exact_batch<MODEL, true, 1>), grid, block, 0, s, nd, m, b);                   \
    } else {                                                                                                     \
        if (scan == 3) RM_KLAUNCH((k_exact_batch<MODEL, false, 3>), grid, block, 0, s, nd, m, b);        \
        else if (scan == 4) RM_KLAUNCH((k_exact_batch<MODEL, false, 4>), grid, block, 0, s, nd, m, b);   \
        else RM_KLAUNCH((k_exact_batch<MODEL, false, 1>), grid, block, 0, s, nd, m, b);                  \
    }                                                                                                            \
} while (0)
    if (m.kind == RM_MODEL_LOGDIST && (m.flags & RM_LD_SINR)) { // self-contained ticks of the SINR extension (no draws)
        if (scan == 3) RM_KLAUNCH((k_exact_batch<RM_MODEL_LOGDIST, false, 3, true>), grid, block, 0, s, nd, m, b);
        else if (scan == 4) RM_KLAUNCH((k_exact_batch<RM_MODEL_LOGDIST, false, 4, true>), grid, block, 0, s, nd, m, b);
        else RM_KLAUNCH((k_exact_batch<RM_MODEL_LOGDIST, false, 1, true>), grid, block, 0, s, nd, m, b);
        return hipGetLastError();
    }
    switch (m.kind) {
    case RM_MODEL_NULL: RM_EXB(RM_MODEL_NULL); break;
    case RM_MODEL_UDGM: RM_EXB(RM_MODEL_UDGM); break;
    case RM_MODEL_UDGM_CONST: RM_EXB(RM_MODEL_UDGM_CONST); break;
    case RM_MODEL_N2N: RM_EXB(RM_MODEL_N2N); break;
    case RM_MODEL_LOGDIST: RM_EXB(RM_MODEL_LOGDIST); break;
    default: return hipErrorInvalidValue;
    }
#undef RM_EXB
    return hipGetLastError();
}

hipError_t launch_sinr_acc_batch(hipStream_t s, const ModelDev &m, int n, const TickDev *b, int max_links, int share)
{
    // (a receiver partition hears 1 / share of a tick's links: fewer workgroups per tick, each with something to do)
    RM_KLAUNCH(k_sinr_acc_batch, dim3(max(1, min(max(4, 64 / max(share, 1)), cdiv(max(max_links, 1), 1024))), 1, n), dim3(256), 0, s, m, b);
    return hipGetLastError();
}

// rm_batch_*, SINR stage: every on-air frame's SELF entry, then the interference sums
hipError_t launch_sinr_batch(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev *ticks, int n, const TickDev *b)
{
    int max_eval = 0;
    for (int i = 0; i < n; ++i) max_eval = max(max_eval, ticks[i].n_active - ticks[i].first_eval);
    RM_KLAUNCH(k_self_entries_batch, dim3(cdiv(max(max_eval, 1), 256), 1, n), dim3(256), 0, s, nd, b);
    // one row of workgroups per shard IN USE (a batch appends to 64 of the 256 shards, a receiver partition to 8): the rows of
    // the others would be a thousand workgroups per tick that find nothing
    const int shards = int(ticks[0].shard_mask) + 1;
    int gx = 2; // (measured 1 / 2 / 4 / 8: 5.10 / 5.03 / 5.09 / 5.26 us per tick on a rank's share of configs[3])
    if (const char *e = getenv("RM_SINR_GX")) gx = max(1, atoi(e));
    RM_KLAUNCH(k_sinr_batch, dim3(gx, shards, n), dim3(256), 0, s, m, b);
    return hipGetLastError();
}

} // namespace rm
