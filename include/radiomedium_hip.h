/*
 * radiomedium_hip.h -- C ABI of libradiomedium_hip.so, the MI355X (gfx950) engine for
 * radio-sim's per-packet propagation / delivery-verdict pass.
 *
 * This is the drop-in boundary for the reference's RadioMedium plug-in contract
 * (reference paths relative to /root/reference/radio-medium/java/se/sics/emul8/radiomedium/):
 *
 *     public interface RadioMedium {            RadioMedium.java:35-45
 *         String getName();                     -> rm_get_name
 *         void   setSimulator(Simulator sim);   -> rm_create / rm_nodes_upload / rm_seed / rm_set_time
 *         void   transmit(RadioPacket packet);  -> rm_transmit   (or rm_enqueue_tx + rm_tick_flush)
 *         double getBaseRSSI(Node node);        -> rm_get_base_rssi
 *     }
 *
 * Plain C types only (pointers + sizes); no exceptions cross it; every function returns an
 * int status (RM_OK or a negative RM_ERR_*) unless stated, and rm_last_error() gives the
 * thread-local message of the last failure.  A context is NOT re-entrant: the caller
 * serialises calls per context (the reference enters transmit() from per-socket reader
 * threads, net/JSONClientConnection.java:118-131; the Java shim in INTEGRATION.md holds
 * one lock).  There is no CPU fallback: without a usable HIP device rm_create fails with
 * RM_ERR_NO_DEVICE.
 *
 * Node indices are positions in Simulator.getNodes() (registration order,
 * Simulator.java:245,274) -- the order the reference's loop visits receivers in, and the
 * order heard links are returned in.
 */
#ifndef RADIOMEDIUM_HIP_H
#define RADIOMEDIUM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5: rm_host_result.rssi is NULL for the reference's four media and pkt_rssi carries one value per packet;
 *    a partitioned context that is handed all ranks' source indices keeps, per tick, only the frames that can matter to
 *    its receivers (results unchanged: packets keep their numbers); the ranks' node-table digests ride in the all-gather of
 *    rm_dist_batch_run_sources_device (rm_table_digest, rm_batch_run_gathered_blocks_device, RM_GATHER_TRAILER): a rank
 *    whose copy of the node table differs makes the batch RM_ERR_STATE on every rank.
 * 4: rm_profile_enable times every kernel by its own dispatch (rm_profile_kernels; RM_STAGE_EMPTY is always 0);
 *    rm_batch_run_sources_device / rm_batch_run_gathered_sources_device / rm_dist_batch_run_sources_device take ticks of
 *    the SINR medium whose frames outlive their tick (rm_air_batch_stats); rm_node_info_changed.
 * 3: rm_host_result.pkt is NULL (a link's packet follows from pkt_offset: the column no longer crosses PCIe) and so is
 *    rm_delivery_view.packet (the packet numbers come once per run of deliveries: n_runs, run_*); rm_air_scan_ticks,
 *    rm_batch_run_gathered_sources_device.
 * 2: rm_delivery_view.oldest_packet; rm_host_result.sinr / rm_device_result.sinr are NULL without the SINR extension;
 *    rm_group_*, rm_events_*, rm_node_info, rm_tick_run_records_device, rm_set_partition_spatial and the draw-node
 *    exchange were added; rm_tick_run_device refuses the SINR medium (rm_tick_run_records_device takes it).  A host
 *    built against another version must not load this library: compare rm_abi_version() with RM_ABI_VERSION. */
#define RM_ABI_VERSION 5

#define RM_OK 0
#define RM_ERR_INVALID (-1)   /* bad argument */
#define RM_ERR_NO_DEVICE (-2) /* no usable HIP device / library built without one */
#define RM_ERR_HIP (-3)       /* a HIP runtime call failed */
#define RM_ERR_CAPACITY (-4)  /* caller buffer or link capacity too small (count is still returned) */
#define RM_ERR_STATE (-5)     /* call sequence / configuration not valid for this call */

typedef struct rm_context rm_context;

/* which RadioMedium implementation the context behaves as */
enum rm_model_kind {
    RM_MODEL_NULL = 0,       /* NullRadioMedium.java:47-77 */
    RM_MODEL_UDGM = 1,       /* UDGMRadioMedium.java:63-117 */
    RM_MODEL_UDGM_CONST = 2, /* UDGMConstantLossRadioMedium.java:16-36 */
    RM_MODEL_N2N = 3,        /* N2NRadioMedium.java:24-73 */
    RM_MODEL_LOGDIST = 4     /* build-defined extension, DESIGN.md "Extension spec" */
};

/* events/ReceptionEvent.java:12-16: unheard links are never reported */
enum rm_verdict { RM_UNHEARD = 0, RM_INTERFERED = 1, RM_DELIVERED = 2 };

#define RM_LD_SINR 1 /* logdist: co-channel SINR capture + half duplex over the on-air list */

typedef struct rm_model_params {
    int32_t kind;  /* enum rm_model_kind */
    int32_t flags; /* RM_LD_* */
    /* UDGMRadioMedium.java:18-24 (setters :31-61) */
    double udgm_success_ratio_tx;   /* declared by the reference, never used (kept for parity) */
    double udgm_success_ratio_rx;
    double udgm_transmission_range;
    double udgm_interference_range; /* declared by the reference, never read */
    /* UDGMConstantLossRadioMedium.java:8 */
    double const_range;
    /* extension */
    double ld_pl0_db, ld_exponent, ld_d0;
    double ld_sigma_db, ld_clip;
    uint64_t ld_seed;
    double ld_sensitivity_dbm, ld_noise_dbm, ld_capture_db, ld_ifloor_dbm;
} rm_model_params;

/* One frame on the air: RadioPacket.java:40-52 plus the source state transmit() reads
 * (source position, Position.java; txProbability, Transciever.java:18).  This is also the
 * record the ranks all-gather each tick in the receiver-sharded multi-GPU mode. */
typedef struct rm_tx_record {
    double x, y, z;
    double txpower;
    double txprob;
    int64_t start_us;
    int64_t air_us;
    int32_t src;
    int32_t channel;
} rm_tx_record; /* 64 bytes */

/* device-resident result of the last evaluated tick (all pointers are device memory owned by
 * the context, valid until the next rm_tick_* / rm_transmit call) */
typedef struct rm_device_result {
    const uint32_t *count;      /* [1] heard links */
    const uint32_t *pkt_offset; /* [n_new+1] first link of each new packet */
    const int32_t *pkt;         /* [count] index into this tick's new packets */
    const int32_t *dst;         /* [count] receiver node index */
    const uint8_t *verdict;     /* [count] RM_INTERFERED / RM_DELIVERED */
    const double *rssi;         /* [count] */
    const double *sinr;         /* [count] with the SINR extension, else NULL (host copies deliver 0) */
    uint32_t capacity;
} rm_device_result;

/* ---- life cycle ---------------------------------------------------------------------- */
int rm_abi_version(void);
int rm_device_count(void);                              /* >=0, or RM_ERR_* */
int rm_create(int device_ordinal, rm_context **out);    /* Simulator.setRadioMedium(...) -> setSimulator */
void rm_destroy(rm_context *ctx);
const char *rm_last_error(void);
const char *rm_get_name(const rm_context *ctx);         /* RadioMedium.getName() */
int rm_set_stream(rm_context *ctx, void *hip_stream);   /* all work is enqueued on this stream */

/* ---- model ---------------------------------------------------------------------------- */
void rm_model_defaults(rm_model_params *p, int32_t kind); /* the reference's field defaults */
int rm_set_model(rm_context *ctx, const rm_model_params *p);
int rm_get_model(const rm_context *ctx, rm_model_params *out);
/* N2NRadioMedium(double[][] m): row-major m x m (net/SimulatorJSONHandler.java:183-206) */
int rm_set_n2n_matrix(rm_context *ctx, int32_t m, const double *row_major);
int rm_set_base_rssi(rm_context *ctx, double rssi);       /* AbstractRadioMedium.java:51-53 */
double rm_get_base_rssi(const rm_context *ctx, int32_t node); /* RadioMedium.getBaseRSSI */

/* ---- Simulator.getRandom(): one java.util.Random shared by all packets ------------------ */
int rm_seed(rm_context *ctx, int64_t seed);               /* new java.util.Random(seed) */
int rm_get_rng_state(rm_context *ctx, uint64_t *state48);
int rm_set_rng_state(rm_context *ctx, uint64_t state48);

/* ---- node state (Simulator.getNodes() snapshot; Node/Position/Transciever fields) -------- */
int rm_nodes_upload(rm_context *ctx, int32_t n,
                    const double *x, const double *y, const double *z,
                    const double *txpower, const int32_t *channel, const uint8_t *enabled,
                    const double *rxprob, const double *txprob,
                    const int32_t *int_id /* Node.getIdAsInteger(); NULL = index+1 */);
/* One changed node (node-config-set, SimulatorJSONHandler.java:105-143: position, rf-power,
 * wireless-channel, rx-loss, tx-loss, radio-state).  Written in place on the device by one small
 * launch, no synchronisation; the receiver table is sorted again only after enough receivers have
 * left the box their group of 64 had (results never depend on that order). */
int rm_node_update(rm_context *ctx, int32_t node, double x, double y, double z, double txpower,
                   int32_t channel, uint8_t enabled, double rxprob, double txprob);
/* New positions of `count` nodes in one call (Position.set, Position.java:44-52); z may be NULL
 * (z = 0, as Position.set(x, y) does).  The shim's "dirty list" flushed at the start of a tick. */
int rm_nodes_move(rm_context *ctx, int32_t count, const int32_t *nodes, const double *x, const double *y,
                  const double *z);
/* how often the receiver table has been (re)built and spatially sorted so far (observability) */
int64_t rm_receiver_table_builds(const rm_context *ctx);
int rm_node_count(const rm_context *ctx);
/* receiver range owned by this context (multi-GPU range partitioning); default = all */
int rm_set_partition(rm_context *ctx, int32_t first, int32_t count);
/* Receivers partitioned by REGION instead of by index range: this context owns part `part` of the `n_parts` regions the
 * k-d split of ALL node positions yields (groups of 64 nodes in proportion; ties by node index, so every rank computes the
 * same cut from the same table for itself).  A rank's receivers then lie together, its filter drops the frames far from
 * its region, and a packet's heard links -- still ascending in node index inside every rank -- are merged by node index
 * across the ranks.  The members are fixed until the next rm_nodes_upload / partition call: a node that moves stays with
 * its rank.  n_parts == 1: no partition.  rm_partition_of_nodes labels every node with its part (which rank owns a source,
 * for the all-gather of Tx records); rm_partition_nodes lists this context's receivers in ascending order. */
int rm_set_partition_spatial(rm_context *ctx, int32_t part, int32_t n_parts);
int rm_partition_of_nodes(rm_context *ctx, int32_t n_parts, int32_t *part_of /* [n] */);
int rm_partition_nodes(rm_context *ctx, int32_t *nodes, int32_t cap, int32_t *count);
/* the same cut from positions alone (host only, no context, no device): what rm_partition_of_nodes answers for a table
 * with these positions (z may be NULL: 0) */
int rm_region_split(int32_t n, const double *x, const double *y, const double *z, int32_t n_parts, int32_t *part_of /* [n] */);
int rm_set_link_capacity(rm_context *ctx, uint32_t max_links);

/* ---- time ------------------------------------------------------------------------------ */
int rm_set_time(rm_context *ctx, int64_t current_time_us);          /* Simulator.getTime() */
int64_t rm_air_time_us(int64_t hex_length);                         /* RadioPacket.java:67-75 */
/* Simulator.java:321-335: event times of a packet given the simulator's current time */
void rm_event_times(int64_t start_us, int64_t air_us, int64_t current_time_us,
                    int64_t *t_start, int64_t *t_end);

/* ---- transmit(): one packet, reference semantics ------------------------------------------
 * txpower / channel: optional overrides ("rf-power", "wireless-channel",
 * net/SimulatorJSONHandler.java:83-90); NULL = the source radio's values.
 * Heard receivers are returned in node order.  *count gets the number of heard links even
 * when it exceeds cap (then RM_ERR_CAPACITY).  interference (may be NULL) gets the packet
 * level Tx-failure flag (UDGMRadioMedium.java:88-92). */
int rm_transmit(rm_context *ctx, int32_t src, int64_t start_us, int64_t hex_length,
                const double *txpower, const int32_t *channel,
                int32_t *dst, uint8_t *verdict, double *rssi, double *sinr, uint32_t cap,
                uint32_t *count, uint8_t *interference);

/* ---- batched: all frames of one simulated tick in one pass -------------------------------- */
int rm_tick_begin(rm_context *ctx, int64_t t_begin_us, int64_t t_end_us);
int rm_enqueue_tx(rm_context *ctx, int32_t src, int64_t start_us, int64_t air_us,
                  const double *txpower, const int32_t *channel);
/* records built by the caller: a record's txprob decides the packet's Tx draw (UDGMRadioMedium.java:87-92), also
 * where it differs from the node table */
int rm_enqueue_tx_records(rm_context *ctx, const rm_tx_record *recs, int32_t n);
/* evaluates the tick and copies the heard links (packet-major, receiver ascending) out */
int rm_tick_flush(rm_context *ctx, int32_t *pkt, int32_t *dst, uint8_t *verdict,
                  double *rssi, double *sinr, uint32_t cap, uint32_t *count,
                  uint8_t *pkt_interference /* [n_new] or NULL */,
                  uint32_t *pkt_offset /* [n_new+1] or NULL */);

/* The same without the copy: the engine's last kernel writes the tick's result into pinned,
 * host-mapped memory the context owns, and the caller reads it in place (a JNI shim wraps the
 * arrays in direct ByteBuffers).  The pointers stay valid until the next call that evaluates
 * anything on this context.  `count` records are there; an error is reported as by rm_tick_flush. */
typedef struct rm_host_result {
    uint32_t count;                  /* heard links */
    uint32_t n_packets;              /* frames of this tick */
    const uint32_t *pkt_offset;      /* [n_packets + 1] first link of every packet */
    const uint8_t *pkt_interference; /* [n_packets] Tx-failure flag (UDGMRadioMedium.java:88-92) */
    const int32_t *pkt;              /* NULL since ABI version 3: link i belongs to the packet q with pkt_offset[q] <= i < pkt_offset[q + 1]
                                      * (4 of a record's 17 bytes that need not cross PCIe; rm_tick_flush still fills its caller's array) */
    const int32_t *dst;              /* [count] receiver node index (ascending per packet) */
    const uint8_t *verdict;          /* [count] RM_INTERFERED / RM_DELIVERED */
    const double *rssi;              /* [count] -- or NULL (ABI version 5) for the reference's four media: a heard link's rssi is its
                                      * packet's transmit power there (UDGMRadioMedium.java:95, NullRadioMedium.java:57,
                                      * N2NRadioMedium.java:51, UDGMConstantLossRadioMedium.java:22), and it crosses PCIe once per packet: */
    const double *sinr;              /* [count] with the SINR extension, else NULL (as rm_device_result.sinr) */
    const double *pkt_rssi;          /* [n_packets] when rssi is NULL: link i of packet q has rssi pkt_rssi[q] (5 bytes per link
                                      * instead of 13); NULL when rssi is not (rm_tick_flush still fills its caller's array) */
} rm_host_result;
int rm_tick_flush_view(rm_context *ctx, rm_host_result *out);

/* A medium in which a frame is heard by a large share of all nodes -- the reference's default NullRadioMedium (every same-channel
 * node: NullRadioMedium.java:62-73), a lossless N2N matrix, a unit disc over a small field -- is evaluated in node order, and
 * what its tick leaves IS its result: per (packet, chunk of 1024 consecutive nodes) cell sixteen 64-bit lane masks -- bit l
 * of mask k set: node rx_first + 1024 * chunk + 64 * k + l heard the packet -- and the cell's count.  A heard link's rssi is
 * its packet's transmit power, its verdict its packet's (pkt_interference): nothing else distinguishes links of such a medium,
 * so the 17-byte records (68 MB for 4 M links) are written only when rm_result_device / rm_result_copy / a host view asks.
 * The tick itself ends with the masks and the cells' counts (one launch); the packets' offsets and the total are laid out on the
 * context's stream when this call (or rm_result_count) first asks for them.
 * All pointers are device memory, valid until the next evaluating call; RM_ERR_STATE when the last tick took another form. */
typedef struct rm_dense_result {
    const unsigned long long *cell_mask; /* [n_packets][chunks][16] */
    const uint32_t *cell_count;          /* [n_packets][chunks] */
    const uint32_t *count;               /* [1] heard links of the tick */
    const uint32_t *pkt_offset;          /* [n_packets + 1] */
    const uint8_t *pkt_interference;     /* [n_packets] */
    int32_t n_packets, chunks, rx_first;
} rm_dense_result;
int rm_result_dense(rm_context *ctx, rm_dense_result *out);

/* evaluate the enqueued tick without copying anything out (then rm_result_copy / rm_result_device) */
int rm_tick_run(rm_context *ctx);

/* ---- receiver partitions with probabilistic links --------------------------------------------
 * The shared java.util.Random is consumed in packet order, then node order = rank order, so a
 * rank needs every rank's per-packet draw counts before it can place its own draws.  After
 * rm_tick_run / rm_tick_run_device on a partitioned context whose links may draw,
 * rm_draws_pending() is 1: all-gather rm_draw_counts_device (uint32[n_new] per rank, rank-major)
 * and call rm_tick_finish_draws; only then are verdicts, Tx-failure flags and the generator
 * state final (identical on all ranks). */
int rm_draws_pending(const rm_context *ctx);
int rm_draw_counts_device(rm_context *ctx, const uint32_t **dev_counts, int32_t *n_new);
int rm_draw_counts_to(rm_context *ctx, uint32_t *dev_out); /* async copy into a caller's device buffer */
int rm_tick_finish_draws(rm_context *ctx, const uint32_t *all_counts /* [world][n_new] */, int32_t world,
                         int32_t rank, int on_device);
/* Spatial partitions (rm_set_partition_spatial): the ranks' node sets interleave in node order, so the counts are not
 * enough -- every rank also publishes the node index of each of its links that will draw, packet-major, ascending inside
 * a packet (rm_draw_nodes_device: sum of its counts entries), the lists are all-gathered into rows of `stride` entries
 * per rank, and a link's place among its packet's draws is the number of listed nodes below it over all ranks. */
int rm_draw_nodes_device(rm_context *ctx, const int32_t **dev_nodes);
int rm_tick_finish_draws_nodes(rm_context *ctx, const uint32_t *all_counts /* [world][n_new] */,
                               const int32_t *all_nodes /* [world][stride] */, uint32_t stride, int32_t world, int on_device);

/* ---- device-resident path (bench, multi-GPU): no host copies ------------------------------ */
/* build tx records for sources `dev_src[0..n)` from the resident node state */
int rm_pack_tx_device(rm_context *ctx, const int32_t *dev_src, int32_t n, int64_t start_us,
                      int64_t air_us, rm_tx_record *dev_out);
/* the same on another stream (packing + all-gather of tick t+1 can overlap the sweep of tick t) */
int rm_pack_tx_device_on(rm_context *ctx, void *hip_stream, const int32_t *dev_src, int32_t n,
                         int64_t start_us, int64_t air_us, rm_tx_record *dev_out);
/* the same for n_ticks ticks in one launch: dev_src / dev_out hold n_ticks rows of n entries, row b
 * starts at start_us[b] (host array) -- feeds one RCCL all-gather per batch of ticks */
int rm_pack_tx_batch_device_on(rm_context *ctx, void *hip_stream, const int32_t *dev_src, int32_t n_ticks, int32_t n,
                               const int64_t *start_us, int64_t air_us, rm_tx_record *dev_out);
/* evaluate one tick whose new frames are `dev_new[0..n_new)` (device memory, canonical order).  Records in device
 * memory are not inspected by the host: their txprob has to be the source node's (as rm_pack_tx_device builds
 * them) -- whether a java.util.Random draw can happen is decided from the node table and the model. */
int rm_tick_run_device(rm_context *ctx, int64_t t_begin_us, int64_t t_end_us,
                       const rm_tx_record *dev_new, int32_t n_new);
/* the same with the new frames given as source node indices (device int32[n], -1 = padding): the
 * Tx records are built from the resident node state inside the sweep -- RadioPacket(node, time,
 * data) copies txpower / channel from its source, RadioPacket.java:46-52 -- one call per tick */
int rm_tick_run_sources_device(rm_context *ctx, int64_t t_begin_us, int64_t t_end_us,
                               const int32_t *dev_src, int32_t n, int64_t start_us, int64_t air_us);
/* as rm_tick_run_device for the SINR extension, whose frames stay on the air over several ticks (the gathered records
 * of a receiver-sharded tick): `latest_end_us` is an upper bound of start + air over the records
 * -- the host never reads them and has to know how long their entries can matter.  The engine keeps a copy of the
 * records while they are on the air.  Without the SINR extension this is rm_tick_run_device. */
int rm_tick_run_records_device(rm_context *ctx, int64_t t_begin_us, int64_t t_end_us, const rm_tx_record *dev_new,
                               int32_t n_new, int64_t latest_end_us);
int rm_result_device(rm_context *ctx, rm_device_result *out);
int rm_result_count(rm_context *ctx, uint32_t *count, uint32_t *dropped); /* synchronises */
/* copy the last evaluated tick's heard links to host buffers (same layout as rm_tick_flush) */
int rm_result_copy(rm_context *ctx, int32_t *pkt, int32_t *dst, uint8_t *verdict, double *rssi,
                   double *sinr, uint32_t cap, uint32_t *count, uint8_t *pkt_interference,
                   uint32_t *pkt_offset);
int rm_sync(rm_context *ctx);

/* ---- several independent ticks per pass ---------------------------------------------------------
 * RadioMedium.transmit treats every packet on its own (UDGMRadioMedium.java:63-117 reads nothing
 * but the packet, the node table and the shared Random), so the frames of n_ticks consecutive
 * ticks can be swept together: one launch sequence for all of them instead of one per tick,
 * which is what fills an MI355X at the 100k-node sizes (a single tick is a few dependent
 * launches of a few microseconds each).  Tick b's results live in result slot b of the context
 * (slot 0 is also what rm_result_* read) until the next rm_batch_* / rm_tick_* call.  The
 * java.util.Random draws are consumed tick by tick in slot order, i.e. exactly as n_ticks
 * single rm_tick_run_sources_device calls would.
 * The RM_LD_SINR extension looks at every frame on the air.  With source indices (rm_batch_run_sources_device and the
 * gathered-sources forms) the frames' time spans are in the arguments, and both kinds of batch are taken:
 *  - self-contained ticks (no frame of an earlier call or of an earlier tick of the batch still on the air when a tick
 *    begins: air time <= tick length) keep per-tick interferer lists inside the sweep;
 *  - ticks whose frames OUTLIVE them (BASELINE configs[4]: 8128 us frames over 1000 us ticks), or that begin while frames
 *    of earlier calls are on the air: the heard links of all ticks come from the sweep of the medium without SINR, then
 *    every frame the batch can see -- the context's on-air window, then the batch's ticks -- is indexed once and the
 *    interference sums of all ticks are formed in bulk (rm_airbatch.hip).  A frame's verdicts are decided against the
 *    frames of its own and earlier ticks, exactly as n_ticks single calls would decide them; the batch's frames join the
 *    on-air window for the calls that follow.  The ticks have to be in time order and their links must not draw.
 * With records (rm_batch_run_device, rm_batch_run_gathered_device) the host cannot see the time spans: the ticks
 * [t_begin, t_end] must not overlap and every frame has to lie inside its tick -- verified on the device, a violation is
 * reported as RM_ERR_STATE when the tick's result is read.  Anything else of it, and partitioned contexts whose links draw,
 * are refused with RM_ERR_STATE -- run those one tick at a time. */
#define RM_MAX_BATCH 512
int rm_batch_run_sources_device(rm_context *ctx, int32_t n_ticks, const int64_t *t_begin_us /* [n_ticks] */,
                                const int64_t *t_end_us, const int32_t *const *dev_src /* device int32[n_src[b]] each */,
                                const int32_t *n_src, const int64_t *start_us, const int64_t *air_us);
/* the same with the ticks' Tx records given (device memory, canonical order), as rm_tick_run_device */
int rm_batch_run_device(rm_context *ctx, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us,
                        const rm_tx_record *const *dev_new, const int32_t *n_new);
/* the same with the ticks' records where an all-gather of per-rank blocks left them: dev_gathered[rank][tick][slot]
 * (`world` ranks, each packed `slots` records per tick for all n_ticks ticks, src = -1 = padding); tick b's packets are
 * the world * slots records dev_gathered[(r * n_ticks + b) * slots + s], rank-major -- no transposition in between */
int rm_batch_run_gathered_device(rm_context *ctx, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us,
                                 const rm_tx_record *dev_gathered, int32_t world, int32_t slots);
/* the same from the ticks' SOURCE INDICES where an all-gather of per-rank blocks left them: dev_src_all[rank][tick][slot]
 * (-1: padding; all frames of tick b start at start_us[b] and last air_us).  Every rank holds the whole node table, so
 * it builds the records of all ranks' frames itself: what has to cross the links between the GPUs is 4 bytes per frame
 * instead of a 64-byte record (rm_dist_batch_run_sources_device does exactly this around its ncclAllGather) */
int rm_batch_run_gathered_sources_device(rm_context *ctx, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us,
                                         const int32_t *dev_src_all, int32_t world, int32_t slots, const int64_t *start_us,
                                         int64_t air_us);
/* the same with every rank's block as rm_dist_batch_run_sources_device's own all-gather leaves it: n_ticks * slots source
 * indices followed by RM_GATHER_TRAILER words, the first two of them the rank's rm_table_digest (low word, high word).
 * Every rank builds the other ranks' records from ITS copy of the node table (the reference has no change hook --
 * net/SimulatorJSONHandler.java:105-143 -- so a host may miss an update): a block whose digest differs from this
 * context's makes every tick of the batch read as RM_ERR_STATE. */
#define RM_GATHER_TRAILER 4
int rm_batch_run_gathered_blocks_device(rm_context *ctx, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us,
                                        const int32_t *dev_blocks /* [world][n_ticks * slots + RM_GATHER_TRAILER] */, int32_t world,
                                        int32_t slots, const int64_t *start_us, int64_t air_us);
/* digest of the node table as this context holds it: a function of its content (node count and every node's fields),
 * whatever sequence of rm_nodes_upload / rm_node_update / rm_nodes_move calls produced it */
int rm_table_digest(const rm_context *ctx, uint64_t *digest);
/* ticks of the last batch a filter workgroup swept with one load of its 1024 receivers: the receiver table left HBM once per
 * that many ticks of the launch (what a roofline has to charge the launch for the table) */
int rm_batch_tile_reuse(const rm_context *ctx);
int rm_batch_result_device(rm_context *ctx, int32_t slot, rm_device_result *out);
int rm_batch_result_count(rm_context *ctx, int32_t slot, uint32_t *count, uint32_t *dropped); /* synchronises */
int rm_batch_result_copy(rm_context *ctx, int32_t slot, int32_t *pkt, int32_t *dst, uint8_t *verdict, double *rssi,
                         double *sinr, uint32_t cap, uint32_t *count, uint8_t *pkt_interference,
                         uint32_t *pkt_offset);
/* The results of slots 0 .. n_slots-1 in the context's pinned, host-mapped block: one packing
 * launch, one wait; out[b] points into the block (valid until the next evaluating call).
 * status (may be NULL) gets every slot's own RM_OK / RM_ERR_CAPACITY / RM_ERR_STATE; the return
 * value is the first of them that is not RM_OK. */
int rm_batch_result_view(rm_context *ctx, int32_t n_slots, rm_host_result *out, int32_t *status);

/* Kernel timing on the context's stream: on every `every_n`-th launch sequence (0 = off) every kernel launch carries its
 * own pair of HIP events (hipExtLaunchKernelGGL), which take the start and the end of THAT dispatch -- the interval a
 * rocprofv3 kernel trace reports for it, without the launch gap before it and without whatever other contexts have in
 * flight.  rm_profile_read returns the number of sampled sequences and the summed kernel milliseconds per stage,
 * rm_profile_kernels the same per kernel, under the name rocprofv3 prints (template arguments as the launch site spells them). */
enum rm_profile_stage {
    RM_STAGE_FILTER = 0,  /* k_filter (or k_tick_prep + k_filter_wg): all (frame, receiver) pairs, conservative */
    RM_STAGE_EXACT = 1,   /* k_exact: the reference's fp64 arithmetic on the candidates */
    RM_STAGE_SELF = 2,    /* k_self_entries (SINR) */
    RM_STAGE_OFFSETS = 3, /* k_cell_off + k_slot_scan */
    RM_STAGE_SINR = 4,    /* k_sinr */
    RM_STAGE_SCATTER = 5, /* k_finalize */
    RM_STAGE_REORDER = 6, /* k_reorder */
    RM_STAGE_DRAWS = 7,   /* java.util.Random kernels */
    RM_STAGE_EMPTY = 8,   /* unused since ABI version 4 (was: the cost of an event bracket); always 0 */
    RM_PROFILE_STAGES = 9
};
int rm_profile_enable(rm_context *ctx, int every_n);
int rm_profile_read(rm_context *ctx, uint32_t *samples, double *stage_ms /* [RM_PROFILE_STAGES] */);
typedef struct rm_kernel_time {
    char name[96];      /* e.g. "k_filter_wg_batch<4, true>" */
    int32_t stage;      /* rm_profile_stage the launch belongs to */
    uint32_t launches;  /* sampled launches */
    double total_ms;    /* their summed duration */
} rm_kernel_time;
/* count gets the number of distinct kernels sampled since rm_profile_enable; at most cap entries are written */
int rm_profile_kernels(rm_context *ctx, rm_kernel_time *out, int32_t cap, int32_t *count);
/* number of Tx->Rx link evaluations resolved by the last tick ( T * (N_loc) minus self links ) */
int64_t rm_last_link_evaluations(const rm_context *ctx);
/* observability (synchronises): the candidate links the sweep's conservative filter handed to the exact stage for the
 * tick of result slot `slot` (0 on the one-launch tick path, which keeps no candidate list) and its heard links */
int rm_slot_stats(rm_context *ctx, int32_t slot, uint64_t *candidates, uint64_t *heard);
/* the SINR extension's on-air lists (build-defined, DESIGN.md "Extension spec" E4): how many ticks added only their new
 * frames to the per-receiver interferer lists kept on the device, and how many rebuilt the lists from every frame on the
 * air (first tick, after a node / model / partition / capacity change, after a dropped tick, when t_begin went back) */
int rm_air_list_stats(const rm_context *ctx, uint64_t *incremental_ticks, uint64_t *rebuilt_ticks);
/* ... and how many ticks needed no lists at all: a tick of at most 4096 new frames over a spatially sorted table finds the
 * interferers of its heard links among the frames on the air themselves (rm_airscan.hip; RM_SINR_SCAN=0 keeps the lists) */
int rm_air_scan_ticks(const rm_context *ctx, uint64_t *scan_ticks);
/* ... and the batches of ticks whose frames outlive their tick (rm_batch_run_sources_device and the gathered-sources forms take
 * them since ABI version 4: heard links by the batch sweep, interference over the whole batch, rm_airbatch.hip), and their ticks */
int rm_air_batch_stats(const rm_context *ctx, uint64_t *batches, uint64_t *ticks);
/* observability (synchronises): the (heard link, frame on the air) pairs the last such batch evaluated exactly, the frames
 * its index held (on-air window + the batch's own), and how many of the pairs turned out to interfere */
int rm_air_batch_pairs(rm_context *ctx, uint64_t *pairs, uint64_t *frames, uint64_t *interferers);
/* the entry ring behind those lists (synchronises): entries allocated since the lists were last rebuilt in the busiest of the
 * 256 sub-rings, and the entries a sub-ring holds -- more allocated than held: the ring has gone round (old entries were reclaimed) */
int rm_air_ring_stats(rm_context *ctx, uint64_t *max_allocated, uint64_t *sub_ring_entries);

/* ---- several devices behind one caller --------------------------------------------------------------
 * The reference host is ONE process (Main.java:65-73): a group drives n contexts from one host thread, one
 * per device (an ordinal may repeat: several partitions on one GPU).  Receivers are partitioned over the
 * members by region (rm_set_partition_spatial; or by node index range, rm_group_set_partitioning); a tick's
 * Tx records are on the host already, so they are simply handed to every member -- no all-gather; every
 * member evaluates them against its receivers, the launches of all members are enqueued before any result
 * is waited for, and the heard links are merged packet-major / node ascending (a k-way merge by node index).
 * Probabilistic links: the per-packet draw counts (and, for regions, the drawing links' nodes) are exchanged
 * through the host and every member places its draws among the other members' (rm_tick_finish_draws*), so
 * verdicts, Tx-failure flags and the java.util.Random state are those of one context.  Everything else of a member (reception stage, node-info, device-resident results) is reached
 * through rm_group_context. */
typedef struct rm_group rm_group;
int rm_group_create(int32_t n_members, const int32_t *device_ordinals, rm_group **out);
void rm_group_destroy(rm_group *g);
/* members own regions of the plane (default, rm_set_partition_spatial) or ranges of node indices (spatial = 0);
 * before rm_group_nodes_upload */
int rm_group_set_partitioning(rm_group *g, int32_t spatial);
int rm_group_size(const rm_group *g);
rm_context *rm_group_context(rm_group *g, int32_t member);
int rm_group_set_model(rm_group *g, const rm_model_params *p);
int rm_group_set_n2n_matrix(rm_group *g, int32_t m, const double *row_major);
int rm_group_seed(rm_group *g, int64_t seed);
int rm_group_get_rng_state(rm_group *g, uint64_t *state48);
int rm_group_set_link_capacity(rm_group *g, uint32_t max_links_per_member);
int rm_group_nodes_upload(rm_group *g, int32_t n, const double *x, const double *y, const double *z,
                          const double *txpower, const int32_t *channel, const uint8_t *enabled,
                          const double *rxprob, const double *txprob, const int32_t *int_id);
int rm_group_node_update(rm_group *g, int32_t node, double x, double y, double z, double txpower,
                         int32_t channel, uint8_t enabled, double rxprob, double txprob);
int rm_group_set_time(rm_group *g, int64_t current_time_us);
int rm_group_tick_begin(rm_group *g, int64_t t_begin_us, int64_t t_end_us);
int rm_group_enqueue_tx(rm_group *g, int32_t src, int64_t start_us, int64_t air_us, const double *txpower,
                        const int32_t *channel);
int rm_group_enqueue_tx_records(rm_group *g, const rm_tx_record *recs, int32_t n);
/* as rm_tick_flush, over all members */
int rm_group_tick_flush(rm_group *g, int32_t *pkt, int32_t *dst, uint8_t *verdict, double *rssi, double *sinr,
                        uint32_t cap, uint32_t *count, uint8_t *pkt_interference, uint32_t *pkt_offset);

/* the device-resident tick of a group: dev_src[r] = `slots` source node indices in member r's device memory (the
 * transmitters member r owns, -1 = padding).  Every member packs the Tx records of its transmitters from its resident node
 * state, the members all-gather the packed blocks -- RCCL (ncclAllGather over a communicator from ncclCommInitAll, xGMI
 * between the devices; rm_group_uses_rccl tells) when every member has its own device, copies on the device when several
 * members share one -- and every member sweeps the gathered frames (packet order: member after member, slot after slot)
 * against its receivers.  Nothing crosses PCIe; rm_group_result_copy merges the members' heard links like
 * rm_group_tick_flush, rm_group_context(g, r) + rm_result_device leave them on the devices. */
int rm_group_tick_run_sources_device(rm_group *g, int64_t t_begin_us, int64_t t_end_us, const int32_t *const *dev_src,
                                     int32_t slots, int64_t start_us, int64_t air_us);
int rm_group_result_copy(rm_group *g, int32_t *pkt, int32_t *dst, uint8_t *verdict, double *rssi, double *sinr, uint32_t cap,
                         uint32_t *count, uint8_t *pkt_interference, uint32_t *pkt_offset);
int rm_group_uses_rccl(rm_group *g); /* 1: the members exchange over RCCL, 0: copies on one device, < 0: error */

/* ---- RCCL inside the library: one process per GPU without a framework in between -------------------------------
 * The receiver-sharded tick of SURVEY.md section 8e as ONE call per rank: pack the Tx records of the transmitters this
 * rank owns (from its resident node state), ncclAllGather of the packed blocks over xGMI, sweep of the gathered frames
 * against this rank's receivers (rm_set_partition_spatial / rm_set_partition), all on the context's stream.  RCCL is
 * bound at run time (dlopen: a process that already holds an RCCL shares it; RM_RCCL_LIB names another file);
 * rm_comm_available() tells whether it could be.  Rank 0 makes the id (ncclGetUniqueId) and hands it to the other ranks by
 * whatever means the host has (a file, a socket, MPI, torch.distributed's store); every rank then calls rm_comm_init_rank
 * on its context.  A context without a communicator is a world of one (no collective).
 *   dev_src: n_ticks rows of `slots` source node indices (this rank's transmitters of every tick, -1 = padding).
 *   Packet order of a tick: rank after rank, slot after slot (world * slots packets, padding included).
 * rm_dist_tick_run_sources_device also exchanges the java.util.Random draw counts (regions: the drawing links' nodes)
 * over the same communicator and finishes the draws; it is the form for media with draws and for the SINR medium with
 * frames that stay on the air.  Results: rm_result_* / rm_batch_result_*, as after rm_tick_run_device / rm_batch_run_device. */
#define RM_COMM_ID_BYTES 128
int rm_comm_available(void);
int rm_comm_get_unique_id(uint8_t *id /* [RM_COMM_ID_BYTES] */);
int rm_comm_init_rank(rm_context *ctx, const uint8_t *id, int32_t world, int32_t rank);
int rm_comm_destroy(rm_context *ctx);
int rm_comm_world(const rm_context *ctx);
int rm_comm_rank(const rm_context *ctx);
int rm_dist_batch_run_sources_device(rm_context *ctx, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us,
                                     const int32_t *dev_src, int32_t slots, const int64_t *start_us, int64_t air_us);
int rm_dist_tick_run_sources_device(rm_context *ctx, int64_t t_begin_us, int64_t t_end_us, const int32_t *dev_src, int32_t slots,
                                    int64_t start_us, int64_t air_us);

/* ---- reception stage: what the reference does with the verdicts, on the device -------------------------
 * SURVEY.md section 8f-1 / 8f-3.  After rm_events_enable every evaluated tick (rm_transmit, rm_tick_flush*,
 * rm_tick_run*, not rm_batch_*) also hands its packets and heard links to the event stage, exactly as the
 * reference's media call Simulator.generateTransmissionEvents / generateReceptionEvents (Simulator.java:321-350)
 * per packet and heard link: event times max(start, rm_set_time value) and + air time.  The constant-loss
 * medium queues nothing and delivers synchronously (UDGMConstantLossRadioMedium.java:30): its links come
 * out of the next rm_events_process call first, in call order.
 *
 * rm_events_process(t) is Simulator.emulatorTimeStepDone (:155-165): currentTime = t; processAllEvents(t)
 * pops every event with time < t (strict, :213-228) in the order of the reference's ladder queue
 * (com/botbox/scheduler/EventQueue.java; equal timestamps included, see csrc/rm_evorder.hpp) and executes
 * it (events/ReceptionEvent.java:35-46, events/TransmissionEvent.java:18-26, Transciever.java:80-113).  The
 * deliveries -- the Simulator.deliverRadioPacket(packet, destination, rssi) calls of that drain, in call
 * order -- are returned in pinned, host-mapped memory (valid until the next rm_events_process).  Packets are
 * numbered in the order they were handed in since rm_events_enable (rm_events_next_packet tells the number
 * the next one gets; every record of a tick counts, padding records too).
 *
 * rm_node_info gives the per-node fields of a time-step message (net/JSONClientConnection.java:331-341):
 * Transciever.getRSSI (:52-61), getReceivingState (:67-78: 0 listening, 1 transmitting, 2 receiving,
 * 3 disabled) and the wireless channel, from the device-resident radio state; nodes == NULL: nodes 0..n-1.
 * On a receiver partition (rm_set_partition) a context keeps the events of its own nodes only. */
typedef struct rm_delivery_view {
    uint32_t count;            /* deliveries of this drain */
    uint32_t pending_packets;  /* packets with events still queued */
    int64_t oldest_packet;     /* number of the oldest of them (== rm_events_next_packet: none): every packet
                                * below it has fired its last event and may be forgotten by the host */
    const int64_t *packet;     /* NULL since ABI version 3: the packet numbers come once per run (below) */
    const int32_t *dst;        /* [count] destination node index */
    const double *rssi;        /* [count] */
    /* The list in runs: all deliveries of one packet in this drain are adjacent (one fired end group of the queue), so the
     * packet number crosses the link once per run instead of once per delivery (8 of 20 bytes).  Run r covers the
     * deliveries [run_first[r], run_first[r] + run_count[r]); the runs are in list order and cover it completely. */
    uint32_t n_runs;
    const int64_t *run_packet; /* [n_runs] */
    const uint32_t *run_first; /* [n_runs] */
    const uint32_t *run_count; /* [n_runs] */
} rm_delivery_view;
int rm_events_enable(rm_context *ctx, uint32_t max_pending_packets, uint32_t max_pending_links); /* 0, 0: defaults */
int rm_events_disable(rm_context *ctx);
int64_t rm_events_next_packet(rm_context *ctx);
int rm_events_process(rm_context *ctx, int64_t time_us, rm_delivery_view *out);
int rm_node_info(rm_context *ctx, const int32_t *nodes, int32_t n, double *rssi, int32_t *receiving, int32_t *channel);
/* The same, incrementally (ABI version 4): only the nodes whose (rssi, receiving state, channel) differ from what THIS call
 * reported for them last -- the first call after rm_events_enable or a new node table reports every node.  A time-step
 * message repeats every node's fields each step (net/JSONClientConnection.java:326-353); a host that keeps the text it sent
 * last re-serialises only these.  nodes / rssi / receiving / channel take up to cap entries (cap >= the node count), in no
 * particular order; count gets how many there are. */
int rm_node_info_changed(rm_context *ctx, int32_t *nodes, double *rssi, int32_t *receiving, int32_t *channel, int32_t cap, int32_t *count);

/* ---- host-side helpers exported for tests ------------------------------------------------- */
/* java.util.Random LCG: state after `steps` next() calls */
uint64_t rm_lcg_jump(uint64_t state48, uint64_t steps);
double rm_lcg_next_double(uint64_t *state48);
/* the exact-arithmetic layer of the extension spec as the HOST compiler builds it from the kernels' header
 * (csrc/rm_math.hpp; no device needed): fn 0 det_log2, 1 det_exp2, 2 det_log10, 3 det_pow10, 4 det_normal,
 * 5 Q80 truncation and back; the per-link shadowing hash and its uniform deviate */
double rm_det_math(int32_t fn, double x);
uint64_t rm_link_hash(uint64_t seed, uint32_t a, uint32_t b, double *u);
/* the pop order of the reference's event queue as a sort key (csrc/rm_evorder.hpp; no device needed):
 * rm_evq_add returns the ladder of an event added now with time t, rm_evq_drain is processAllEvents(t) */
typedef struct rm_evq_order {
    int64_t top_start, top_max;
    int32_t ladders, top_nonempty;
} rm_evq_order;
void rm_evq_init(rm_evq_order *o);
int32_t rm_evq_add(rm_evq_order *o, int64_t time_us);
void rm_evq_drain(rm_evq_order *o, int64_t time_us);

#ifdef __cplusplus
}
#endif
#endif /* RADIOMEDIUM_HIP_H */
