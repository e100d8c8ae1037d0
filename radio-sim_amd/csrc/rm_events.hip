// rm_events.hip -- the reception stage: what the reference does with the verdicts after RadioMedium.transmit
// (part of libradiomedium_hip.so; gfx950 only; overview at the top of rm_engine.h)
//
// Reference (paths relative to /root/reference/radio-medium/java/se/sics/emul8/radiomedium/):
//   Simulator.generateTransmissionEvents / generateReceptionEvents (Simulator.java:321-350): two queued events
//     per packet and per heard link, at max(start, currentTime) and + air time -- 2 allocations and 2 locked
//     ladder-queue inserts per heard link, the reference's real bottleneck;
//   Simulator.emulatorTimeStepDone -> processAllEvents (:155-165, :213-228): at the end of a tick every
//     event with time < currentTime is popped and executed;
//   events/ReceptionEvent.java:35-46, events/TransmissionEvent.java:18-26, Transciever.java:52-113: the start
//     flank latches packet + rssi (and clears "sending"), the end flank clears the reception and, in
//     delivery mode, calls Simulator.deliverRadioPacket; setSending clears the reception.
//
// Here the packets of the evaluated ticks stay on the device (a ring of EvPacket + a ring of their heard
// links) and a drain is a handful of data-parallel kernels instead of a queue:
//   * the pop order of the reference's queue is a sort key (rm_evorder.hpp), constant per (packet, flank):
//     a drain ranks the few thousand fired (packet, flank) groups, not the events;
//   * deliveries do not depend on the radio state (a delivery-mode end flank always delivers): the
//     delivery list is the delivered links of the fired end groups, group by group in rank order, inside
//     a group in reverse node order (reverse insertion);
//   * every event ASSIGNS the radio fields it touches (receiving <- true / false, sending <- true / false,
//     rssi <- r), so a node's state after the drain is decided by the LAST event that touched the field:
//     one 64-bit atomicMax per event on (rank, event) keys, then the winners write the state.
// Nothing is approximated: tests/test_gpu_events.py compares delivery lists and node states with a literal
// serial replay of the Java queue kept with the tests, ties included.
#include "rm_device.hpp"
#include "rm_evorder.hpp"

#include <stdlib.h>

namespace rm {

constexpr int64_t kI64Min = int64_t(0x8000000000000000ull);

RM_D void amax_i64(int64_t *p, int64_t v) { (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RM_D unsigned long long ev_key(uint32_t rank, uint32_t kind, uint32_t ref)
{
    return ((unsigned long long)(rank + 1u) << 32) | ((unsigned long long)kind << 29) | (unsigned long long)(ref & 0x1FFFFFFFu);
}
RM_D void amax_key(unsigned long long *p, unsigned long long k) { (void)__hip_atomic_fetch_max(p, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RM_D bool owned(const EvDev &e, int node)
{
    if (e.member) return node >= 0 && node < e.n_nodes && e.member[node] != 0;
    return node >= e.own_first && node < e.own_first + e.own_count;
}

// ============================================================================ append
// One evaluated tick handed to the event stage: per packet an EvPacket (event times, ladders, packet
// number) and a copy of its heard links.  What Simulator.generate*Events does per call, for the whole tick.
template <bool SEG>
RM_D void ev_append_body(const EvDev &e, const EvLinkSrc &ls, const rm_tx_record *__restrict__ tx, int n_new, int64_t now, int immediate,
                         const uint32_t *dropped_flag, const int blk, const int n_blk)
{
    __shared__ uint32_t s_off[SEG ? kFusedScanMax + 1 : 1];
    __shared__ uint32_t s_wave[4];
    __shared__ int64_t s_top[4];
    EvState &st = *e.st;
    const EvTails tl = st.tails[e.par];
    const uint32_t pk_head = st.pk_head, pk_tail = tl.pk_tail, pool_head = st.pool_head, pool_tail = tl.pool_tail;
    const int64_t gseq0 = tl.gseq_next;
    EvOrder ord;
    ord.top_start = st.top_start;
    ord.ladders = st.ladders;
    ord.top_max = 0;
    ord.top_nonempty = 0;
    uint32_t total;
    if (SEG) total = block_scan_counts(ls.cnt, ls.n_scan, s_off, s_wave, nullptr, nullptr);
    else total = ls.off[n_new];
    uint32_t err = 0;
    if (dropped_flag && *dropped_flag != 0u) err |= 8u; // the tick itself overflowed the link capacity: it has no links
    if (pk_tail - pk_head + uint32_t(n_new) > e.pk_mask + 1u) err |= 1u;
    if (pool_tail - pool_head + total > e.pool_mask + 1u) err |= 2u;
    const int lane = threadIdx.x & 63;
    int64_t wave_top = kI64Min; // latest event of this wave's packets that waits in the queue's top list
    if (!err) {
        for (int q = blk * 4 + wave_index(); q < n_new; q += n_blk * 4) { // wave-uniform
            const rm_tx_record r = tx[q];
            const uint32_t cnt = uniform_u(SEG ? ls.cnt[q] : (ls.off[q + 1] - ls.off[q]));
            const uint32_t src0 = uniform_u(ls.off[q]);
            const uint32_t dst0 = pool_tail + uniform_u(SEG ? s_off[q] : ls.off[q]);
            uint32_t n_deliver = 0;
            for (uint32_t j0 = 0; j0 < cnt; j0 += 64) { // wave-uniform
                const uint32_t j = j0 + lane;
                bool deliver = false;
                if (j < cnt) {
                    const uint32_t o = (dst0 + j) & e.pool_mask;
                    const uint8_t v = ls.verdict[src0 + j];
                    e.l_dst[o] = ls.dst[src0 + j];
                    e.l_rssi[o] = ls.rssi[src0 + j];
                    e.l_verdict[o] = v;
                    deliver = v == RM_DELIVERED;
                }
                n_deliver += uint32_t(__popcll(ballot64(deliver)));
            }
            if (lane == 0) {
                EvPacket p;
                int64_t t0 = r.start_us;           // Simulator.java:323-326
                if (t0 < now) t0 = now;
                p.t0 = t0;
                p.t1 = t0 + r.air_us;
                p.gseq = gseq0 + q;
                p.link_off = dst0;
                p.link_cnt = cnt;
                p.src = r.src;
                p.lad0 = ev_ladder(ord, p.t0);
                p.lad1 = ev_ladder(ord, p.t1);
                p.flags = 0u;
                if (immediate) p.flags |= kEvImmediate | kEvNoTx;          // UDGMConstantLossRadioMedium.java:30: no events at all
                else if (r.src < 0 || !owned(e, r.src)) p.flags |= kEvNoTx; // the source's Transciever lives on another rank
                if (r.src < 0) p.flags |= kEvStartDone | kEvDone;           // a padding record is no packet
                p.n_deliver = n_deliver;
                p.pad[0] = p.pad[1] = p.pad[2] = 0u;
                e.pk[(pk_tail + uint32_t(q)) & e.pk_mask] = p;
                // every packet's transmission events sit in the reference's (one, global) queue, whoever owns the source
                if (!immediate && r.src >= 0) {
                    if (p.t0 >= ord.top_start && p.t0 > wave_top) wave_top = p.t0;
                    if (p.t1 >= ord.top_start && p.t1 > wave_top) wave_top = p.t1;
                }
            }
        }
    }
    // one atomic per workgroup (a single word takes ~88 atomics per microsecond: one per packet would cost more
    // than everything else in this kernel)
    if (lane == 0) s_top[wave_index()] = wave_top;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t m4 = s_top[0];
        for (int w = 1; w < 4; ++w) m4 = s_top[w] > m4 ? s_top[w] : m4;
        if (m4 != kI64Min) amax_i64(&st.top_max, m4);
    }
    // the tails after this tick, for the next launch (see EvTails): every workgroup computed the same values
    if (blk == 0 && threadIdx.x == 0) {
        EvTails nt = tl;
        nt.err = tl.err | err;
        if (!err) {
            nt.pk_tail = pk_tail + uint32_t(n_new);
            nt.pool_tail = pool_tail + total;
        }
        nt.gseq_next = gseq0 + n_new; // the packets were transmitted, kept or not
        st.tails[e.par ^ 1] = nt;
    }
}

template <bool SEG>
__global__ void __launch_bounds__(256)
k_ev_append(const EvDev e, const EvLinkSrc ls, const rm_tx_record *__restrict__ tx, int n_new, int64_t now, int immediate,
            const uint32_t *dropped_flag)
{
    ev_append_body<SEG>(e, ls, tx, n_new, now, immediate, dropped_flag, int(blockIdx.x), int(gridDim.x));
}

// ============================================================================ drain
// k_ev_select: which (packet, flank) groups fire in processAllEvents(T)?  Sort key = (time, meta).
RM_D uint64_t ev_meta(uint32_t lad_rel, uint32_t order, uint32_t phase) { return (uint64_t(lad_rel) << 40) | (uint64_t(order) << 1) | phase; }

// The tick whose append runs in the SAME launch as the selection (k_ev_append_select): its packets are not in the ring yet, so
// the selection derives what it needs of them -- event times, ladders, flags, deliveries -- from the tick's own records, the
// way the append does.
struct EvFresh {
    EvLinkSrc ls;
    const rm_tx_record *tx;
    int n_new;
    int64_t now;
    int immediate;
    const uint32_t *dropped_flag;
};

template <bool FRESH, bool SEG>
RM_D void ev_select_body(const EvDev &e, const int64_t T, const uint32_t blk, const EvFresh &fr)
{
    EvState &st = *e.st;
    const EvTails tl = st.tails[e.par]; // (FRESH: the tails before the tick's append, which writes the other set)
    const uint32_t head = st.pk_head, w_old = tl.pk_tail - head;
    uint32_t w = w_old;
    EvOrder ord;
    if (FRESH) { // does the append keep the tick?  (its own tests: ev_append_body)
        __shared__ uint32_t s_tot[4];
        uint32_t total = 0;
        if (SEG) {
            uint32_t part = 0;
            for (int i = threadIdx.x; i < fr.ls.n_scan; i += 256) part += fr.ls.cnt[i];
            for (int d = 32; d >= 1; d >>= 1) part += uint32_t(__shfl_xor(int(part), d));
            if ((threadIdx.x & 63) == 0) s_tot[threadIdx.x >> 6] = part;
            __syncthreads();
            total = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
        } else {
            total = fr.ls.off[fr.n_new];
        }
        bool err = fr.dropped_flag && *fr.dropped_flag != 0u;
        err = err || (tl.pk_tail - head + uint32_t(fr.n_new) > e.pk_mask + 1u);
        err = err || (tl.pool_tail - st.pool_head + total > e.pool_mask + 1u);
        if (!err) w += uint32_t(fr.n_new);
        ord.top_start = st.top_start;
        ord.ladders = st.ladders;
        ord.top_max = 0;
        ord.top_nonempty = 0;
    }
    const int lane = threadIdx.x & 63;
    if (blk * blockDim.x >= ((w + 63u) & ~63u)) return; // (the grid is sized for the host's bound on the window)
    const uint32_t i = blk * blockDim.x + threadIdx.x;
    const uint32_t idx = (head + i) & e.pk_mask;
    EvPacket p{};
    bool valid = false;
    if (i < w_old) {
        p = e.pk[idx];
        valid = !(p.flags & kEvDone);
    } else if (FRESH && i < w) { // a packet of the tick being appended
        const int q = int(i - w_old);
        const rm_tx_record r = fr.tx[q];
        const uint32_t cnt = SEG ? fr.ls.cnt[q] : (fr.ls.off[q + 1] - fr.ls.off[q]);
        const uint32_t src0 = fr.ls.off[q];
        uint32_t n_deliver = 0;
        if (fr.ls.per_frame_verdict) { // (one load instead of a chain of them: this thread has the packet to itself)
            if (cnt != 0u && fr.ls.verdict[src0] == RM_DELIVERED) n_deliver = cnt;
        } else {
            uint32_t j = 0;
            for (; j + 8u <= cnt; j += 8u) { // eight loads in flight
                uint8_t v8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v8[u] = fr.ls.verdict[src0 + j + uint32_t(u)];
#pragma unroll
                for (int u = 0; u < 8; ++u) n_deliver += (v8[u] == RM_DELIVERED) ? 1u : 0u;
            }
            for (; j < cnt; ++j) n_deliver += (fr.ls.verdict[src0 + j] == RM_DELIVERED) ? 1u : 0u;
        }
        int64_t t0 = r.start_us; // Simulator.java:323-326
        if (t0 < fr.now) t0 = fr.now;
        p.t0 = t0;
        p.t1 = t0 + r.air_us;
        p.gseq = tl.gseq_next + q;
        p.src = r.src;
        p.lad0 = ev_ladder(ord, p.t0);
        p.lad1 = ev_ladder(ord, p.t1);
        p.flags = 0u;
        if (fr.immediate) p.flags |= kEvImmediate | kEvNoTx;
        else if (r.src < 0 || !owned(e, r.src)) p.flags |= kEvNoTx;
        if (r.src < 0) p.flags |= kEvStartDone | kEvDone;
        p.n_deliver = n_deliver;
        valid = !(p.flags & kEvDone);
    }
    // the oldest packet with events still queued BEFORE this drain: k_ev_finish moves the ring heads up to it (the
    // packets this drain finishes are passed over by the next one -- a window scan per drain costs more than it frees)
    {
        __shared__ uint32_t s_live;
        if (threadIdx.x == 0) s_live = 0xFFFFFFFFu;
        __syncthreads();
        const uint64_t live = ballot64(valid);
        if (live && lane == 0) atomicMin(&s_live, (i & ~63u) + uint32_t(__ffsll((long long)live) - 1));
        __syncthreads();
        if (threadIdx.x == 0 && s_live != 0xFFFFFFFFu) atomicMin(&st.first_live, s_live); // one per workgroup
    }
    const int64_t gseq_head = (FRESH && w_old == 0u) ? tl.gseq_next : e.pk[head & e.pk_mask].gseq; // (an empty ring: the tick's first packet)
    const uint32_t rel = uint32_t(p.gseq - gseq_head);
    const int32_t lb = st.ladders;
    const bool imm = (p.flags & kEvImmediate) != 0u;
    const bool fire_start = valid && !imm && !(p.flags & kEvStartDone) && p.t0 < T;
    const bool fire_end = valid && (imm || p.t1 < T);
    { // the drain's deliveries and the runs they come in (a packet's), one atomic per wave each
        uint32_t nd = fire_end ? p.n_deliver : 0u;
        const uint32_t runs = uint32_t(__popcll(ballot64(nd != 0u)));
        for (int d = 32; d >= 1; d >>= 1) nd += uint32_t(__shfl_xor(int(nd), d));
        if (lane == 0 && nd) {
            atomicAdd(&st.n_deliv, nd);
            atomicAdd(&st.n_dgroups, runs);
        }
    }
    for (int ph = 0; ph < 2; ++ph) { // 0 end, 1 start
        const bool fire = ph ? fire_start : fire_end;
        const uint64_t hm = ballot64(fire);
        if (hm == 0) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&st.n_groups, uint32_t(__popcll(hm)));
        base = uniform_u(base);
        if (base + uint32_t(__popcll(hm)) > e.g_cap) {
            if (lane == 0) st.err |= 4u;
            continue;
        }
        if (fire) {
            const uint32_t k = base + lane_prefix(hm);
            if (imm) { // synchronous deliveries: before anything queued, in call order
                e.g_time[k] = kI64Min;
                e.g_meta[k] = ev_meta(0u, rel, 0u);
            } else {
                e.g_time[k] = ph ? p.t0 : p.t1;
                e.g_meta[k] = ev_meta(uint32_t((ph ? p.lad0 : p.lad1) - lb), 0xFFFFFFFFu - rel, uint32_t(ph));
            }
            e.g_ref[k] = (idx << 1) | uint32_t(ph);
            e.cnt_by_rank[k] = ph ? 0u : p.n_deliver; // (by group here: an end group delivers its packet's delivery-mode links)
            // a node that starts to send in this drain: reception starts on it must take part in the "sending"
            // field's last-writer contest (k_ev_emit skips that for every other idle node)
            if (ph && !(p.flags & kEvNoTx)) e.send_key[p.src] = 1ull;
        }
    }
}

__global__ void __launch_bounds__(256) k_ev_select(const EvDev e, int64_t T)
{
    ev_select_body<false, false>(e, T, blockIdx.x, EvFresh{});
}

// The closed loop's tick and drain follow each other at once (Simulator.java:155-165): the tick's append and the drain's
// selection in ONE launch -- the first n_append workgroups append, the others select; neither reads what the other writes.
template <bool SEG>
__global__ void __launch_bounds__(256) k_ev_append_select(const EvDev e, const EvFresh fr, const int64_t T, const int n_append)
{
    if (int(blockIdx.x) < n_append)
        ev_append_body<SEG>(e, fr.ls, fr.tx, fr.n_new, fr.now, fr.immediate, fr.dropped_flag, int(blockIdx.x), n_append);
    else
        ev_select_body<true, SEG>(e, T, blockIdx.x - uint32_t(n_append), fr);
}

// A fired end group's deliveries, into the host-mapped list at the place its rank gives them, and its run record.  One wave.
// (verdict_first / d_first / rssi_first: the first 64 links' records, asked for by the caller in front of whatever it had to
// wait for.)
RM_D void ev_write_deliveries(const EvDev &e, const EvOut &out, const int64_t gseq, const uint32_t cnt, const uint32_t off0, const bool imm,
                              const uint32_t n_del, const uint32_t first, const uint32_t run, const int lane, const uint8_t verdict_first,
                              const int d_first, const double rssi_first)
{
    if (n_del == 0u) return;
    if (lane == 0) { // the run: the packet's number once, not with each delivery -- left in device memory
        // (write-through: the workgroup that finishes the drain, on whatever XCD, copies all runs to the host in one go;
        // a thousand scattered 16-byte writes over PCIe cost more than the deliveries' packet numbers had)
        if (run < out.run_cap) {
            __hip_atomic_store(&e.run_rec[2u * run], (unsigned long long)gseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&e.run_rec[2u * run + 1u], (unsigned long long)first | ((unsigned long long)n_del << 32), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    uint32_t seen = 0;
    for (uint32_t j0 = 0; j0 < cnt; j0 += 64) { // wave-uniform
        const uint32_t j = j0 + lane;
        bool deliver = false;
        uint32_t o = 0;
        if (j < cnt) {
            o = (off0 + j) & e.pool_mask;
            deliver = (j0 == 0u ? verdict_first : e.l_verdict[o]) == RM_DELIVERED;
        }
        const uint64_t dm = ballot64(deliver);
        if (deliver) {
            const uint32_t nth = seen + lane_prefix(dm); // n-th delivered link of the packet in node order
            // queued end events pop in reverse insertion order = reverse node order; the constant-loss
            // medium delivers synchronously in node order
            const uint32_t pos = first + (imm ? nth : (n_del - 1u - nth));
            if (pos < out.cap) { // host-mapped memory: write-through stores at system scope (drained by the storing workgroup
                // before it reports in or ends: no release fence, which would also write back this XCD's whole L2)
                __hip_atomic_store(&out.dst[pos], j0 == 0u ? d_first : e.l_dst[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&out.rssi[pos], j0 == 0u ? rssi_first : e.l_rssi[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        seen += uint32_t(__popcll(dm));
    }
}

// k_ev_emit: one wave per fired group: its place in the pop order, and every event's (rank, event) key to the fields of
// the node it touches (the last-writer contest).  The deliveries themselves are written by k_ev_apply: they depend on
// nothing the contest decides, and their stores into host-mapped memory (PCIe-bound) then run under the state update.
__global__ void __launch_bounds__(256) k_ev_emit(const EvDev e, const EvOut out, const int share, const int lds_keys)
{
    const uint32_t G = min(e.st->n_groups, e.g_cap);
    const int lane = threadIdx.x & 63;
    // the groups' keys, once per workgroup: every wave's pass below reads all of them, and four waves fetching the same 40 KB
    // from the L2 each was most of the pass (a closed-loop drain has ~2000 groups; more than kEmitKeys: the pass reads global memory)
    constexpr uint32_t kEmitKeys = 2048;
    __shared__ int64_t s_time[kEmitKeys];
    __shared__ uint64_t s_meta[kEmitKeys];
    __shared__ uint32_t s_cnt[kEmitKeys];
    const bool in_lds = lds_keys != 0 && G <= kEmitKeys;
    // (this wave's first group is asked for in front of the staging: its chain -- group -> packet -> links -- starts a round trip earlier)
    const uint32_t g_first = blockIdx.x * 4 + wave_index();
    const uint32_t ref_first = (g_first < G) ? e.g_ref[g_first] : 0u;
    if (in_lds && blockIdx.x * 4u < G) { // eight keys per thread, every load in flight before the first store: ONE round trip
        constexpr int kPer = int(kEmitKeys / 256u);
        int64_t tk[kPer];
        uint64_t mk[kPer];
        uint32_t ck[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const uint32_t k = uint32_t(u) * 256u + threadIdx.x;
            const bool in = k < G;
            tk[u] = in ? e.g_time[k] : 0;
            mk[u] = in ? e.g_meta[k] : 0ull;
            ck[u] = in ? e.cnt_by_rank[k] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const uint32_t k = uint32_t(u) * 256u + threadIdx.x;
            s_time[k] = tk[u];
            s_meta[k] = mk[u];
            s_cnt[k] = ck[u];
        }
    }
    __syncthreads();
    for (uint32_t g = g_first; g < G; g += gridDim.x * 4) { // wave-uniform
        const uint32_t ref = uniform_u(g == g_first ? ref_first : e.g_ref[g]);
        // (the packet's fields and its first 64 links' nodes are asked for here, in front of the pass over the keys: they
        // depend on nothing the pass finds, and behind it they were two more dependent round trips per group)
        const EvPacket &p = e.pk[ref >> 1];
        const uint32_t cnt = uniform_u(p.link_cnt), off0 = uniform_u(p.link_off), fl = uniform_u(p.flags);
        const int src = uniform_i(p.src);
        const bool start = (ref & 1u) != 0u;
        const bool imm = (fl & kEvImmediate) != 0u;
        // (share: every share-th end group's deliveries go out from here -- the host-mapped list is PCIe-bound, 0.5 MB per
        // drain, and half of it under this kernel's pass and contest leaves k_ev_apply the other half)
        const bool mine = !start && share > 0 && (g % uint32_t(share)) == 0u;
        const uint32_t n_del_g = mine ? uniform_u(e.cnt_by_rank[g]) : 0u;
        int d_first = 0;
        uint8_t sending_first = 0, verdict_first = 0;
        double rssi_first = 0.0;
        if (uint32_t(lane) < cnt && (!(imm && !start) || n_del_g != 0u)) {
            const uint32_t o = (off0 + uint32_t(lane)) & e.pool_mask;
            d_first = e.l_dst[o];
            if (start) sending_first = e.sending[d_first];
            if (n_del_g != 0u) {
                verdict_first = e.l_verdict[o];
                rssi_first = e.l_rssi[o];
            }
        }
        // The group's place in the queue's pop order = the number of fired groups with a smaller key (keys are unique), and
        // the place of its deliveries in the list = the deliveries of those groups: one pass of the wave over all groups'
        // keys (a few thousand, L2-resident) -- no sort, no scan, no launch in between.
        uint32_t r = 0, first = 0, run = 0;
        {
            const int64_t tg = in_lds ? s_time[g] : e.g_time[g];
            const uint64_t mg = in_lds ? s_meta[g] : e.g_meta[g];
            for (uint32_t k0 = 0; k0 < G; k0 += 64 * 8) { // eight groups per lane in flight: from global memory the pass is a chain of L2 round trips
                int64_t tk[8];
                uint64_t mk[8];
                uint32_t ck[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t k = k0 + uint32_t(u) * 64u + uint32_t(lane);
                    const bool in = k < G;
                    if (in_lds) { // (wave-uniform)
                        tk[u] = in ? s_time[k] : int64_t(0x7FFFFFFFFFFFFFFFll);
                        mk[u] = in ? s_meta[k] : ~0ull;
                        ck[u] = in ? s_cnt[k] : 0u;
                    } else {
                        tk[u] = in ? e.g_time[k] : int64_t(0x7FFFFFFFFFFFFFFFll);
                        mk[u] = in ? e.g_meta[k] : ~0ull;
                        ck[u] = in ? e.cnt_by_rank[k] : 0u;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool less = tk[u] < tg || (tk[u] == tg && mk[u] < mg);
                    r += less ? 1u : 0u;
                    first += less ? ck[u] : 0u;
                    run += (less && ck[u] != 0u) ? 1u : 0u;
                }
            }
            for (int d = 32; d >= 1; d >>= 1) {
                r += uint32_t(__shfl_xor(int(r), d));
                first += uint32_t(__shfl_xor(int(first), d));
                run += uint32_t(__shfl_xor(int(run), d));
            }
            r = uniform_u(r);
            first = uniform_u(first);
            if (lane == 0) {
                e.g_rank[g] = r;        // k_ev_apply recomputes the events' keys from it ...
                e.off_by_rank[g] = first; // ... and writes the group's deliveries from here on ...
                e.g_run[g] = run;       // ... as this run of the list
            }
        }
        if (mine) ev_write_deliveries(e, out, p.gseq, cnt, off0, imm, n_del_g, first, uniform_u(run), lane, verdict_first, d_first, rssi_first);
        if (start) {
            // ReceptionEvent start flank: setReceiving = clearSending + latch (Transciever.java:80-84)
            for (uint32_t j = lane; j < cnt; j += 64) {
                const uint32_t o = (off0 + j) & e.pool_mask;
                const unsigned long long k = ev_key(r, kEvRxStart, o);
                const int d = (j < 64u) ? d_first : e.l_dst[o];
                amax_key(&e.recv_key[d], k);
                // clearSending only matters on a node that is sending or starts to in this drain (marked by k_ev_select)
                if (((j < 64u) ? sending_first : e.sending[d]) || __hip_atomic_load(&e.send_key[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull)
                    amax_key(&e.send_key[d], k);
            }
            // TransmissionEvent start: setSending = clearReceiving + sendingPacket (Transciever.java:106-109)
            if (lane == 0 && !(fl & kEvNoTx)) {
                const unsigned long long k = ev_key(r, kEvTxStart, 0u);
                amax_key(&e.send_key[src], k);
                amax_key(&e.recv_key[src], k);
            }
        } else {
            if (!imm)
                for (uint32_t j = lane; j < cnt; j += 64) { // end flank: clearReceiving, whichever packet
                    const uint32_t o = (off0 + j) & e.pool_mask;
                    amax_key(&e.recv_key[(j < 64u) ? d_first : e.l_dst[o]], ev_key(r, kEvRxEnd, o));
                }
            if (lane == 0 && !(fl & kEvNoTx)) amax_key(&e.send_key[src], ev_key(r, kEvTxEnd, 0u)); // clearSending
        }
    }
}

RM_D void ev_finish_body(const EvDev &e, const EvOut &out, int64_t T, uint32_t seq);

// k_ev_apply: the last writer of a field writes it (and clears its key); the workgroup that is done last finishes the drain.
__global__ void __launch_bounds__(256) k_ev_apply(const EvDev e, const EvOut out, int64_t T, uint32_t seq, const int share)
{
    __shared__ uint32_t s_lastwg;
    const uint32_t G = min(e.st->n_groups, e.g_cap);
    const int lane = threadIdx.x & 63;
    for (uint32_t g = blockIdx.x * 4 + wave_index(); g < G; g += gridDim.x * 4) { // wave-uniform
        // (everything indexed by the group at once, then the packet, then the first 64 links' records at once: four dependent
        // round trips per group instead of seven)
        const uint32_t ref = uniform_u(e.g_ref[g]);
        const uint32_t r = uniform_u(e.g_rank[g]);
        const uint32_t n_del_g = uniform_u(e.cnt_by_rank[g]), first_g = uniform_u(e.off_by_rank[g]), run_g = uniform_u(e.g_run[g]);
        EvPacket &p = e.pk[ref >> 1];
        const uint32_t cnt = uniform_u(p.link_cnt), off0 = uniform_u(p.link_off), fl = uniform_u(p.flags);
        const int src = uniform_i(p.src);
        const bool start = (ref & 1u) != 0u;
        const bool imm = (fl & kEvImmediate) != 0u;
        int d_first = 0;
        double rssi_first = 0.0;
        uint8_t verdict_first = 0;
        if (uint32_t(lane) < cnt) {
            const uint32_t o = (off0 + uint32_t(lane)) & e.pool_mask;
            d_first = e.l_dst[o];
            if (start || n_del_g != 0u) rssi_first = e.l_rssi[o];
            if (!start) verdict_first = e.l_verdict[o];
        }
        if (!start && !(share > 0 && (g % uint32_t(share)) == 0u)) // (the other end groups' deliveries left with k_ev_emit)
            ev_write_deliveries(e, out, p.gseq, cnt, off0, imm, n_del_g, first_g, run_g, lane, verdict_first, d_first, rssi_first);
        // the fired flank is done (k_ev_select of the next drain reads this; the other flank's wave reads only the bits
        // that never change)
        if (lane == 0) atomicOr(&p.flags, start ? kEvStartDone : (kEvDone | kEvStartDone));
        if (start) {
            for (uint32_t j = lane; j < cnt; j += 64) {
                const uint32_t o = (off0 + j) & e.pool_mask;
                const unsigned long long k = ev_key(r, kEvRxStart, o);
                const int d = (j < 64u) ? d_first : e.l_dst[o];
                if (e.recv_key[d] == k) {
                    e.receiving[d] = 1;
                    e.latched[d] = (j < 64u) ? rssi_first : e.l_rssi[o];
                    e.recv_key[d] = 0ull;
                }
                if (e.send_key[d] == k) {
                    e.sending[d] = 0;
                    e.send_key[d] = 0ull;
                }
            }
            if (lane == 0 && !(fl & kEvNoTx)) {
                const unsigned long long k = ev_key(r, kEvTxStart, 0u);
                if (e.send_key[src] == k) {
                    e.sending[src] = 1;
                    e.send_key[src] = 0ull;
                }
                if (e.recv_key[src] == k) {
                    e.receiving[src] = 0;
                    e.recv_key[src] = 0ull;
                }
            }
        } else {
            if (!imm) {
                for (uint32_t j = lane; j < cnt; j += 64) {
                    const uint32_t o = (off0 + j) & e.pool_mask;
                    const unsigned long long k = ev_key(r, kEvRxEnd, o);
                    const int d = (j < 64u) ? d_first : e.l_dst[o];
                    if (e.recv_key[d] == k) {
                        e.receiving[d] = 0;
                        e.recv_key[d] = 0ull;
                    }
                }
                if (lane == 0 && !(fl & kEvNoTx)) {
                    const unsigned long long k = ev_key(r, kEvTxEnd, 0u);
                    if (e.send_key[src] == k) {
                        e.sending[src] = 0;
                        e.send_key[src] = 0ull;
                    }
                }
            }
        }
    }
    {
        // (the radio state and the deliveries this workgroup wrote are read by later launches / by the host after the
        // header's sequence number: every storing wave drains its stores before the workgroup reports in)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) { // two levels (EvState::done_sub): the last of its sixteenth reports to the one word
            const uint32_t sub = blockIdx.x & uint32_t(kEvDoneSub - 1);
            const uint32_t in_sub = (gridDim.x - sub + uint32_t(kEvDoneSub - 1)) / uint32_t(kEvDoneSub); // workgroups with this residue
            uint32_t last = 0u;
            if (atomicAdd(&e.st->done_sub[sub * 32u], 1u) == in_sub - 1u) {
                e.st->done_sub[sub * 32u] = 0u;
                last = (atomicAdd(&e.st->done_apply, 1u) == min(gridDim.x, uint32_t(kEvDoneSub)) - 1u) ? 1u : 0u;
            }
            s_lastwg = last;
        }
        __syncthreads();
        if (!s_lastwg) return;
        if (threadIdx.x == 0) e.st->done_apply = 0u;
        ev_finish_body(e, out, T, seq);
    }
}

// k_ev_finish (one workgroup, after k_ev_apply): the fired groups' packets are marked, the ring heads move, the
// queue's ladder rule for this drain (rm_evorder.hpp) is applied and the delivery list's header is published.
RM_D void ev_finish_body(const EvDev &e, const EvOut &out, int64_t T, uint32_t seq)
{
    EvState &st = *e.st;
    const EvTails tl = st.tails[e.par];
    const uint32_t head = st.pk_head, tail = tl.pk_tail;
    // thread 0's part first -- loads that depend on each other (the ring heads, the oldest packet still queued) -- so that it
    // flies under the copy of the runs and its wait, not behind them
    uint32_t total = 0, runs = 0, err = 0, new_head = 0;
    int64_t oldest = 0;
    if (threadIdx.x == 0) {
        // ring heads: up to the oldest packet that still had events queued when this drain began (k_ev_select)
        const uint32_t live = st.first_live;
        st.first_live = 0xFFFFFFFFu;
        new_head = head + min(live, tail - head);
        st.pk_head = new_head;
        st.pool_head = (new_head == tail) ? tl.pool_tail
                                          : __hip_atomic_load(&e.pk[new_head & e.pk_mask].link_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        EvOrder o;
        o.top_start = st.top_start;
        o.top_max = st.top_max;
        o.ladders = st.ladders;
        o.top_nonempty = (st.top_max != kI64Min) ? 1 : 0;
        ev_drain(o, T);
        st.top_start = o.top_start;
        st.ladders = o.ladders;
        if (!o.top_nonempty) st.top_max = kI64Min;
        st.t_prev = T;
        total = st.n_deliv;
        runs = __hip_atomic_load(&st.n_dgroups, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (st.n_groups > e.g_cap) st.err |= 4u;
        err = st.err | tl.err;
        // the ring head's own number: a tick that did not fit the rings was numbered but left nothing here, so
        // "next - pending" would skip packets that are still queued
        oldest = (new_head == tail) ? tl.gseq_next
                                    : __hip_atomic_load(&e.pk[new_head & e.pk_mask].gseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    { // the runs of the delivery list, from where the groups' waves left them: one coalesced copy to the host
        const uint32_t runs_all = min(__hip_atomic_load(&st.n_dgroups, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), out.run_cap);
        for (uint32_t r = threadIdx.x; r < runs_all; r += blockDim.x) {
            const unsigned long long pk = __hip_atomic_load(&e.run_rec[2u * r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long fc = __hip_atomic_load(&e.run_rec[2u * r + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&out.run_packet[r], (int64_t)pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&out.run_first[r], uint32_t(fc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&out.run_count[r], uint32_t(fc >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    st.n_groups = 0u;
    st.n_deliv = 0u;
    st.n_dgroups = 0u;
    // an overflow is reported by the drain that follows it, once: the rings take the next ticks again
    st.err = 0u;
    st.tails[e.par].err = 0u;
    // The header lives in host-mapped memory: four 16-byte stores, each with the sequence number the host polls (EvHeader) --
    // no waiting between the fields and "the" number.  (Nor a system-scope release fence: it would also write back every dirty
    // line the drain's kernels left in this XCD's L2 -- microseconds, and nothing the host reads: the delivery records and
    // the runs were stored, and waited for, before this.)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 *const q = reinterpret_cast<u32x4 *>(out.hdr);
    auto store16 = [](u32x4 *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { // one write-through store of 16 bytes at system scope
        const u32x4 v = {a, b, c, d};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    };
    store16(q + 0, seq, min(total, out.cap), total, err);
    store16(q + 1, seq, tail - new_head, min(runs, out.run_cap), 0u);
    store16(q + 2, seq, 0u, uint32_t(uint64_t(tl.gseq_next)), uint32_t(uint64_t(tl.gseq_next) >> 32));
    store16(q + 3, seq, 0u, uint32_t(uint64_t(oldest)), uint32_t(uint64_t(oldest) >> 32));
}

// ============================================================================ node-info
// The per-node fields of a time-step message (net/JSONClientConnection.java:331-341): Transciever.getRSSI
// (:52-61: the latched rssi while receiving, else the medium's base RSSI), getReceivingState (:67-78),
// getWirelessChannel -- from the device-resident radio state.
__global__ void __launch_bounds__(256)
k_node_info(const EvDev e, const NodesDev nd, const int32_t *__restrict__ nodes, int n, double base_rssi, NodeInfoOut out, uint32_t seq,
            uint32_t *done_counter)
{
    __shared__ uint32_t s_last;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const int i = nodes ? nodes[k] : k;
        double rssi = base_rssi;
        int state = 0, ch = 0;
        if (i >= 0 && i < e.n_nodes) {
            const bool rx = e.receiving[i] != 0;
            if (rx) rssi = e.latched[i];
            state = !nd.senabled[i] ? 3 : (rx ? 2 : (e.sending[i] ? 1 : 0)); // DISABLED / RECEIVING / TRANSMITTING / LISTENING
            ch = nd.schannel[i];
        }
        out.rssi[k] = rssi;
        out.receiving[k] = state;
        out.channel[k] = ch;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        s_last = (atomicAdd(done_counter, 1u) == gridDim.x - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        *done_counter = 0u;
        __threadfence_system();
        __hip_atomic_store(out.seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The same, incrementally: only the nodes whose fields differ from what was REPORTED for them last (rep_*: a device-resident
// copy of the last report; state -1 = never reported).  The changed nodes go to the host-mapped block in whatever order the
// waves find them (one atomic per wave); the workgroup that is done last publishes the count and the sequence number.
// A time-step message repeats every node's fields whether they changed or not (net/JSONClientConnection.java:326-353): a
// host that keeps the text it sent last only needs these.
__global__ void __launch_bounds__(256)
k_node_info_changed(const EvDev e, const NodesDev nd, int n, double base_rssi, double *rep_rssi, int2 *rep_sc, NodeChangeOut out,
                    uint32_t cap, uint32_t seq, uint32_t *counters /* [0] done, [1] changed */)
{
    __shared__ uint32_t s_last;
    const int lane = threadIdx.x & 63;
    for (int i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) { // (whole waves: they count together)
        const int i = i0 + int(threadIdx.x);
        double rssi = base_rssi;
        int state = 0, ch = 0;
        bool changed = false;
        if (i < n) {
            if (i < e.n_nodes) {
                const bool rx = e.receiving[i] != 0;
                if (rx) rssi = e.latched[i];
                state = !nd.senabled[i] ? 3 : (rx ? 2 : (e.sending[i] ? 1 : 0));
                ch = nd.schannel[i];
            }
            const int2 was = rep_sc[i];
            changed = was.x != state || was.y != ch || __double_as_longlong(rep_rssi[i]) != __double_as_longlong(rssi);
        }
        const uint64_t cm = ballot64(changed);
        if (cm == 0ull) continue;
        uint32_t base = 0;
        if (lane == __ffsll((long long)cm) - 1) base = atomicAdd(&counters[1], uint32_t(__popcll(cm)));
        base = uint32_t(__shfl(int(base), __ffsll((long long)cm) - 1));
        if (changed) {
            const uint32_t k = base + lane_prefix(cm);
            if (k < cap) { // (always: the host's block holds every node)
                __hip_atomic_store(&out.node[k], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&out.rssi[k], rssi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&out.receiving[k], state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&out.channel[k], ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                rep_rssi[i] = rssi;
                rep_sc[i] = make_int2(state, ch);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(&counters[0], 1u) == gridDim.x - 1u) ? 1u : 0u;
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        const uint32_t total = __hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        counters[0] = 0u;
        counters[1] = 0u;
        __hip_atomic_store(out.count, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(out.seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ============================================================================ launchers

hipError_t launch_ev_append(hipStream_t s, const EvDev &e, const EvLinkSrc &ls, const rm_tx_record *tx, int n_new, int64_t now,
                            int immediate, const uint32_t *dropped_flag)
{
    if (n_new <= 0) return hipSuccess;
    const dim3 grid(max(1, min(256, cdiv(n_new, 4)))), block(256);
    if (ls.n_scan > 0) {
        if (ls.n_scan > kFusedScanMax) return hipErrorInvalidValue;
        RM_KLAUNCH(k_ev_append<true>, grid, block, 0, s, e, ls, tx, n_new, now, immediate, dropped_flag);
    } else {
        RM_KLAUNCH(k_ev_append<false>, grid, block, 0, s, e, ls, tx, n_new, now, immediate, dropped_flag);
    }
    return hipGetLastError();
}

hipError_t launch_ev_drain(hipStream_t s, const EvDev &e, const EvOut &out, int64_t time_us, uint32_t seq, uint32_t window,
                           const EvLinkSrc *fresh_ls, const rm_tx_record *fresh_tx, int fresh_n, int64_t fresh_now, int fresh_immediate,
                           const uint32_t *fresh_dropped)
{
    // Three launches.  Selection (what fires before time_us; `window`: the host's bound on the pending packets); every
    // fired group's wave finds its place in the pop order by counting, writes its deliveries and enters the last-writer
    // contest of the radio fields; the winners write the state and the workgroup that is done last finishes the drain.
    const uint32_t pk_cap = e.pk_mask + 1u;
    const uint32_t w = (window == 0u || window > pk_cap) ? pk_cap : window;
    EvDev after = e;
    if (fresh_ls && fresh_n > 0) {
        // the tick that was evaluated last has not been appended yet (rm_api_events.cpp: ev_append defers it to here): e.par is
        // the parity its append reads, the stages behind it take the other one
        EvFresh fr{};
        fr.ls = *fresh_ls;
        fr.tx = fresh_tx;
        fr.n_new = fresh_n;
        fr.now = fresh_now;
        fr.immediate = fresh_immediate;
        fr.dropped_flag = fresh_dropped;
        const int n_append = max(1, min(256, cdiv(fresh_n, 4)));
        const dim3 grid(n_append + cdiv(int(w), 256));
        if (fr.ls.n_scan > 0) {
            if (fr.ls.n_scan > kFusedScanMax) return hipErrorInvalidValue;
            RM_KLAUNCH(k_ev_append_select<true>, grid, dim3(256), 0, s, e, fr, time_us, n_append);
        } else {
            RM_KLAUNCH(k_ev_append_select<false>, grid, dim3(256), 0, s, e, fr, time_us, n_append);
        }
        after.par = e.par ^ 1;
    } else {
        RM_KLAUNCH(k_ev_select, dim3(cdiv(int(w), 256)), dim3(256), 0, s, e, time_us);
    }
    static const int share = [] { const char *v = getenv("RM_EV_SHARE"); return v ? atoi(v) : 2; }(); // (0: every delivery from k_ev_apply)
    static const int lds_keys = [] { const char *v = getenv("RM_EV_EMIT_LDS"); return v ? atoi(v) : 1; }(); // (0: the rank pass reads global memory)
    RM_KLAUNCH(k_ev_emit, dim3(512), dim3(256), 0, s, after, out, share, lds_keys);
    RM_KLAUNCH(k_ev_apply, dim3(512), dim3(256), 0, s, after, out, time_us, seq, share);
    return hipGetLastError();
}

static uint32_t *g_dummy = nullptr;

hipError_t launch_node_info(hipStream_t s, const EvDev &e, const NodesDev &nd, const int32_t *dev_nodes, int n, double base_rssi,
                            const NodeInfoOut &out, uint32_t seq)
{
    (void)g_dummy;
    if (n <= 0) return hipSuccess;
    RM_KLAUNCH(k_node_info, dim3(max(1, min(256, cdiv(n, 256)))), dim3(256), 0, s, e, nd, dev_nodes, n, base_rssi, out, seq,
                       &e.st->done_a);
    return hipGetLastError();
}

hipError_t launch_node_info_changed(hipStream_t s, const EvDev &e, const NodesDev &nd, int n, double base_rssi, double *rep_rssi,
                                    int2 *rep_sc, const NodeChangeOut &out, uint32_t cap, uint32_t seq, uint32_t *counters)
{
    if (n <= 0) return hipSuccess;
    RM_KLAUNCH(k_node_info_changed, dim3(max(1, min(256, cdiv(n, 256)))), dim3(256), 0, s, e, nd, n, base_rssi, rep_rssi, rep_sc, out, cap,
               seq, counters);
    return hipGetLastError();
}

} // namespace rm
