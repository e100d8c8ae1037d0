/*
 * rm_oracle.h -- CPU oracle for the radio-medium propagation / delivery-verdict pass.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (radio-sim_amd/) never
 * includes, links or calls anything in this directory.
 *
 * It is a plain-C restatement of the reference's Java arithmetic for the hot path
 * (paths relative to /root/reference/radio-medium/java/se/sics/emul8/radiomedium/):
 *   UDGMRadioMedium.java:63-117, UDGMConstantLossRadioMedium.java:16-36,
 *   N2NRadioMedium.java:24-73, NullRadioMedium.java:47-77, Position.java:56-64,
 *   RadioPacket.java:67-75, Simulator.java:321-350, events/ReceptionEvent.java:35-46,
 *   Transciever.java:52-88, plus java.util.Random as fixed by the Java SE specification.
 *
 * PARITY STATUS: "parity unpinned" by reference tests -- the reference ships no unit
 * tests, assertions, golden vectors or fixtures for this path,
 * and it cannot be compiled or run here (Java, no JDK).  The oracle is pinned instead by
 * the source-derived known-answer tests K1..K10 of SURVEY.md section 8c
 * (tests/test_oracle_kats.py) and by the Java SE LCG known answers.
 *
 * The "logdist" model (log-distance path loss, log-normal shadowing, co-channel SINR
 * capture, multi-tick overlap) does NOT exist in the reference; it is a build-defined
 * extension whose normative text is DESIGN.md section "Extension spec".  The oracle
 * holds an independent implementation of that text.
 */
#ifndef RM_ORACLE_H
#define RM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ORC_MODEL_NULL = 0,          /* NullRadioMedium */
    ORC_MODEL_UDGM = 1,          /* UDGMRadioMedium */
    ORC_MODEL_UDGM_CONST = 2,    /* UDGMConstantLossRadioMedium */
    ORC_MODEL_N2N = 3,           /* N2NRadioMedium */
    ORC_MODEL_LOGDIST = 4        /* extension (not in the reference) */
};

enum { ORC_UNHEARD = 0, ORC_INTERFERED = 1, ORC_DELIVERED = 2 };

enum { ORC_LD_SINR = 1 };       /* logdist flag: co-channel SINR capture + half duplex */

typedef struct {
    int32_t n;
    const double *x, *y, *z;     /* Position.java:37-39 */
    const double *txpower;       /* Transciever.java:11 */
    const int32_t *channel;      /* Transciever.java:12 */
    const uint8_t *enabled;      /* Transciever.java:13 */
    const double *rxprob;        /* Transciever.java:17 */
    const double *txprob;        /* Transciever.java:18 */
    const int32_t *int_id;       /* Node.java:52-58 (Integer.parseInt(id), -1 if not numeric) */
} orc_nodes_t;

typedef struct {
    int32_t kind;
    /* UDGMRadioMedium.java:18-24 */
    double udgm_success_ratio_tx;   /* declared, never used by the reference (K8) */
    double udgm_success_ratio_rx;
    double udgm_transmission_range;
    double udgm_interference_range; /* declared, never read by the reference */
    /* UDGMConstantLossRadioMedium.java:8 */
    double const_range;
    /* N2NRadioMedium.java:9 : row-major m x m */
    const double *n2n_matrix;
    int32_t n2n_m;
    /* extension */
    double ld_pl0_db, ld_exponent, ld_d0;
    double ld_sigma_db, ld_clip;
    uint64_t ld_seed;
    double ld_sensitivity_dbm, ld_noise_dbm, ld_capture_db, ld_ifloor_dbm;
    int32_t ld_flags;
} orc_model_t;

/* one frame on the air; position/txprob are the source's state when it was transmitted */
typedef struct {
    int32_t src;        /* node index (registration order) */
    int32_t channel;    /* RadioPacket.java:50,89 */
    double x, y, z;
    double txpower;     /* RadioPacket.java:49,81 */
    double txprob;
    int64_t start_us;   /* RadioPacket.java:58 */
    int64_t air_us;     /* RadioPacket.java:67-75 */
} orc_packet_t;

void orc_model_defaults(orc_model_t *m, int32_t kind);

/* java.util.Random */
uint64_t orc_jrandom_seed(int64_t seed);
int32_t orc_jrandom_next(uint64_t *state, int bits);
int32_t orc_jrandom_next_int(uint64_t *state);
double orc_jrandom_next_double(uint64_t *state);

/* Position.getDistance, Position.java:56-64 */
double orc_distance(double x1, double y1, double z1, double x2, double y2, double z2);
/* UDGMRadioMedium.getRxSuccessProbability :67-81 / getTxSuccessProbability :63-65 */
double orc_udgm_rx_probability(const orc_model_t *m, const orc_nodes_t *nd, const orc_packet_t *p, int32_t dst);
double orc_udgm_tx_probability(const orc_model_t *m, const orc_packet_t *p);
/* N2NRadioMedium :24-37 */
double orc_n2n_rx_probability(const orc_model_t *m, const orc_nodes_t *nd, const orc_packet_t *p, int32_t dst);

/* RadioPacket.getPacketAirTime :67-75 ; Simulator.generateReceptionEvents :321-335 */
int64_t orc_air_time_us(int64_t hex_length);
void orc_event_times(int64_t start_us, int64_t air_us, int64_t current_time, int64_t *t_start, int64_t *t_end);

void orc_fill_packet(const orc_nodes_t *nd, int32_t src, int64_t start_us, int64_t air_us, orc_packet_t *out);

/*
 * One evaluation pass.  `active[0..n_active)` is the on-air list in canonical order;
 * entries [first_new, n_active) are the frames whose verdicts are decided now (for the four
 * reference models only these are looked at).  Heard links are written packet-major,
 * receiver index ascending -- the order in which the reference calls
 * Simulator.generateReceptionEvents.  out_pkt is relative to first_new.
 * pkt_interference / pkt_draws (length n_active-first_new, may be NULL) receive the packet
 * level Tx-failure flag and the number of nextDouble() calls the packet consumed.
 * Returns the number of heard links (may exceed cap; only cap are stored).
 */
int64_t orc_tick(const orc_model_t *m, const orc_nodes_t *nd, uint64_t *rng_state,
                 const orc_packet_t *active, int32_t n_active, int32_t first_new,
                 int32_t *out_pkt, int32_t *out_dst, uint8_t *out_verdict,
                 double *out_rssi, double *out_sinr, int64_t cap,
                 uint8_t *pkt_interference, int32_t *pkt_draws);
/* the same pass over `threads` threads, for ticks without java.util.Random draws (returns -2 if one would be needed) */
int64_t orc_tick_mt(const orc_model_t *m, const orc_nodes_t *nd, const orc_packet_t *active, int32_t n_active, int32_t first_new,
                    int32_t threads, int32_t *out_pkt, int32_t *out_dst, uint8_t *out_verdict, double *out_rssi, double *out_sinr,
                    int64_t cap, uint8_t *pkt_interference);

/*
 * CPU-baseline leg: verdict pass without record storage, `threads` OpenMP threads over
 * packets (1 = the reference's one-thread-per-packet shape).  Only for models/settings that
 * consume no random draws.  Returns the number of heard links; *delivered gets the count of
 * delivered ones.
 */
int64_t orc_count_links(const orc_model_t *m, const orc_nodes_t *nd,
                        const orc_packet_t *active, int32_t n_active, int32_t first_new,
                        int32_t threads, int64_t *delivered);
int32_t orc_max_threads(void);

/* extension math (DESIGN.md "Extension spec") */
double orc_det_log2(double x);
double orc_det_exp2(double y);
double orc_det_log10(double x);
double orc_det_pow10(double y);
double orc_det_normal(double u);
uint64_t orc_shadow_hash(uint64_t seed, uint32_t a, uint32_t b);
double orc_shadow_gauss(const orc_model_t *m, uint32_t a, uint32_t b);
double orc_logdist_rssi(const orc_model_t *m, const orc_packet_t *p, const orc_nodes_t *nd, int32_t dst);
double orc_fixed_roundtrip(double lin); /* to Q80 fixed point and back (test hook) */

/*
 * Receiver state machine driven by the verdicts (next-1 row; ReceptionEvent.java:35-46,
 * TransmissionEvent.java:18-26, Transciever.java:52-113, Simulator.java:213-228 and the
 * equal-timestamp pop order of com/botbox/scheduler/EventQueue.java:206-244).
 */
enum {
    ORC_EV_RX_START = 0,            /* ReceptionMode.start */
    ORC_EV_RX_END_INTERFERENCE = 1, /* ReceptionMode.interference */
    ORC_EV_RX_END_DELIVERY = 2,     /* ReceptionMode.delivery: simulator.deliverRadioPacket on the end flank */
    ORC_EV_TX_START = 3,            /* TransmissionEvent isStart */
    ORC_EV_TX_END = 4
};

typedef struct {
    int64_t time;
    int32_t node;       /* destination (reception) or source (transmission) */
    int32_t pkt;        /* caller's packet id */
    int32_t kind;       /* ORC_EV_* */
    double rssi;
} orc_event_t;

/* Serial replay (oracle/rm_events.c): the reference's ladder queue (EventQueue.java, literal), the
 * simulator's event generation and tick-end drain, the Transciever state machine. */
typedef struct orc_sim orc_sim_t;
orc_sim_t *orc_sim_create(int32_t n_nodes);
void orc_sim_destroy(orc_sim_t *s);
int64_t orc_sim_time(const orc_sim_t *s);
int32_t orc_sim_error(const orc_sim_t *s);      /* != 0: the Java queue would have thrown */
int64_t orc_sim_move_tops(const orc_sim_t *s);  /* times EventQueue.moveTop ran (test hook) */
int64_t orc_sim_top_start(const orc_sim_t *s);  /* EventQueue.topStart (test hook) */
int64_t orc_sim_pending(const orc_sim_t *s);    /* events still queued */
/* Simulator.generateTransmissionEvents :337-350 / generateReceptionEvents :321-335 (event time = max(start, currentTime)) */
void orc_sim_transmission_events(orc_sim_t *s, int32_t pkt, int32_t src, int64_t start_us, int64_t air_us);
void orc_sim_reception_events(orc_sim_t *s, int32_t pkt, int32_t dst, int64_t start_us, int64_t air_us, double rssi,
                              int32_t do_deliver);
void orc_sim_medium_calls(orc_sim_t *s, const orc_packet_t *packets, int32_t n_packets, int32_t pkt_base, int64_t n_links,
                          const int32_t *out_pkt, const int32_t *out_dst, const uint8_t *out_verdict, const double *out_rssi,
                          int32_t const_loss);
/* Simulator.emulatorTimeStepDone :155-165: currentTime = time; processAllEvents(time).  The executed events in
 * pop order (at most cap stored); deliveries are the ORC_EV_RX_END_DELIVERY entries.  Returns their number. */
int64_t orc_sim_step(orc_sim_t *s, int64_t time, orc_event_t *out_events, int64_t cap);
/* Transciever.getRSSI :52-61 / getReceivingState :67-78 */
double orc_sim_rssi(const orc_sim_t *s, int32_t node, double base_rssi);
int32_t orc_sim_receiving_state(const orc_sim_t *s, int32_t node, int32_t enabled);
int32_t orc_sim_receiving_packet(const orc_sim_t *s, int32_t node);
int32_t orc_sim_sending_packet(const orc_sim_t *s, int32_t node);
/* TEST-ONLY (rm_oracle.c): what Math.pow(v, 2.0) != v * v by whole ulps would change, link by link */
void orc_udgm_pow_sensitivity(const orc_model_t *m, const orc_nodes_t *nd, const orc_packet_t *pk, int32_t n_pk, int32_t d2_ulp,
                              int32_t dmax2_ulp, int64_t *out /* [4] */, double *max_rel);
/* bare queue: op_kind 0 = addEvent(op_time), 1 = pop everything with time < op_time; returns the popped
 * events' insertion numbers in pop order (or -error) */
int64_t orc_evq_replay(const int64_t *op_time, const int32_t *op_kind, int64_t n_ops, int64_t *out_id, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
