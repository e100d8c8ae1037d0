/*
 * rm_oracle.c -- CPU oracle (TEST INFRASTRUCTURE; see rm_oracle.h for the rules).
 *
 * Plain C11, fp64, no contraction (-ffp-contract=off), no fast-math: every arithmetic
 * operation below is one IEEE-754 correctly rounded operation, in the order the reference's
 * Java source performs it.  Reference paths are relative to
 * /root/reference/radio-medium/java/se/sics/emul8/radiomedium/.
 *
 * "parity unpinned": the reference has no tests or golden vectors for this path; the
 * restatement is pinned by the known-answer tests K1..K10 of SURVEY.md section 8c.
 */
#include "rm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ java.util.Random */
/* Java SE specification of java.util.Random (the JDK is not in the reference tree):
 *   seed  = (s ^ 0x5DEECE66D) & ((1<<48)-1)
 *   next(bits): seed = (seed * 0x5DEECE66D + 0xB) & ((1<<48)-1); return (int)(seed >>> (48-bits))
 *   nextDouble(): (((long)next(26) << 27) + next(27)) * 0x1.0p-53
 * Call sites: UDGMRadioMedium.java:85,89,106 ; N2NRadioMedium.java:41,45,62. */
#define JR_MULT 0x5DEECE66DULL
#define JR_ADD 0xBULL
#define JR_MASK ((1ULL << 48) - 1)

uint64_t orc_jrandom_seed(int64_t seed) { return ((uint64_t)seed ^ JR_MULT) & JR_MASK; }

int32_t orc_jrandom_next(uint64_t *state, int bits)
{
    *state = (*state * JR_MULT + JR_ADD) & JR_MASK;
    /* (int)(seed >>> (48 - bits)) : Java narrows to 32 bits, keeping the low ones */
    return (int32_t)(uint32_t)(*state >> (48 - bits));
}

int32_t orc_jrandom_next_int(uint64_t *state) { return orc_jrandom_next(state, 32); }

double orc_jrandom_next_double(uint64_t *state)
{
    int64_t hi = (int64_t)orc_jrandom_next(state, 26);
    int64_t lo = (int64_t)orc_jrandom_next(state, 27);
    return (double)((hi << 27) + lo) * 0x1.0p-53;
}

/* ------------------------------------------------------------------ reference arithmetic */

/* Position.java:56-64 -- this = first point, p2 = second; (dx^2 + dy^2) + dz^2 ; Math.sqrt */
double orc_distance(double x1, double y1, double z1, double x2, double y2, double z2)
{
    double dx = x1 - x2;
    double dy = y1 - y2;
    double dz = z1 - z2;
    dx = dx * dx;
    dy = dy * dy;
    dz = dz * dz;
    return sqrt(dx + dy + dz);
}

/* UDGMRadioMedium.java:63-65 : successRatioRx (sic) * source txProbability */
double orc_udgm_tx_probability(const orc_model_t *m, const orc_packet_t *p)
{
    return m->udgm_success_ratio_rx * p->txprob;
}

/* UDGMRadioMedium.java:67-81.  Math.pow(v, 2.0) is v*v (fdlibm e_pow and the HotSpot
 * intrinsic both special-case y == 2). */
double orc_udgm_rx_probability(const orc_model_t *m, const orc_nodes_t *nd, const orc_packet_t *p, int32_t dst)
{
    double distance = orc_distance(p->x, p->y, p->z, nd->x[dst], nd->y[dst], nd->z[dst]);
    double distanceSquared = distance * distance;
    double distanceMax = m->udgm_transmission_range;
    if (distanceMax == 0.0) {
        return 0.0;
    }
    double distanceMaxSquared = distanceMax * distanceMax;
    double ratio = distanceSquared / distanceMaxSquared;
    if (ratio > 1.0) {
        return 0.0;
    }
    ratio = 1.0 - ratio * (1.0 - m->udgm_success_ratio_rx);
    return ratio * nd->rxprob[dst];
}

/* TEST-ONLY: what the one JDK assumption this oracle takes on faith can cost.  UDGMRadioMedium.java:69,74 square through
 * Math.pow(v, 2.0); this file restates them as v * v (fdlibm's e_pow and HotSpot's intrinsic special-case y == 2), but the
 * Java SE specification only promises a result within 1 ulp.  The same function with distanceSquared and
 * distanceMaxSquared moved by whole ulps (d2_ulp, dmax2_ulp in {-1, 0, +1}: what a pow that is not special-cased may
 * legally return), compared link by link with the unperturbed one over every (packet, receiver) pair:
 *   out[0] links evaluated (filters of UDGMRadioMedium.java:102 applied)
 *   out[1] links heard unperturbed (p > 0)
 *   out[2] links whose heard / unheard status differs
 *   out[3] links heard both ways whose p differs at all
 *  *max_rel  largest |p' - p| / p over the links heard both ways
 * tests/test_oracle_pow_ulp.py holds the counts for the golden scenarios and the BASELINE layouts. */
static double ulp_step(double v, int k)
{
    while (k > 0) { v = nextafter(v, INFINITY); k--; }
    while (k < 0) { v = nextafter(v, -INFINITY); k++; }
    return v;
}
static double udgm_rx_probability_ulp(const orc_model_t *m, const orc_nodes_t *nd, const orc_packet_t *p, int32_t dst, int d2_ulp, int dmax2_ulp)
{
    double distance = orc_distance(p->x, p->y, p->z, nd->x[dst], nd->y[dst], nd->z[dst]);
    double distanceSquared = ulp_step(distance * distance, d2_ulp);
    double distanceMax = m->udgm_transmission_range;
    if (distanceMax == 0.0) return 0.0;
    double distanceMaxSquared = ulp_step(distanceMax * distanceMax, dmax2_ulp);
    double ratio = distanceSquared / distanceMaxSquared;
    if (ratio > 1.0) return 0.0;
    ratio = 1.0 - ratio * (1.0 - m->udgm_success_ratio_rx);
    return ratio * nd->rxprob[dst];
}
void orc_udgm_pow_sensitivity(const orc_model_t *m, const orc_nodes_t *nd, const orc_packet_t *pk, int32_t n_pk, int32_t d2_ulp,
                              int32_t dmax2_ulp, int64_t *out, double *max_rel)
{
    int64_t n_eval = 0, n_heard = 0, n_flip = 0, n_diff = 0;
    double worst = 0.0;
    for (int32_t pi = 0; pi < n_pk; pi++) {
        const orc_packet_t *p = &pk[pi];
        for (int32_t j = 0; j < nd->n; j++) {
            if (j == p->src || !nd->enabled[j] || nd->channel[j] != p->channel) continue;
            const double p0 = orc_udgm_rx_probability(m, nd, p, j);
            const double p1 = udgm_rx_probability_ulp(m, nd, p, j, d2_ulp, dmax2_ulp);
            n_eval++;
            if (p0 > 0.0) n_heard++;
            if ((p0 > 0.0) != (p1 > 0.0)) n_flip++;
            else if (p0 > 0.0 && p1 != p0) {
                n_diff++;
                const double rel = fabs(p1 - p0) / p0;
                if (rel > worst) worst = rel;
            }
        }
    }
    out[0] = n_eval; out[1] = n_heard; out[2] = n_flip; out[3] = n_diff;
    *max_rel = worst;
}

/* N2NRadioMedium.java:28-37 */
double orc_n2n_rx_probability(const orc_model_t *m, const orc_nodes_t *nd, const orc_packet_t *p, int32_t dst)
{
    int32_t sourceID = nd->int_id[p->src];
    int32_t destID = nd->int_id[dst];
    if (m->n2n_matrix == NULL || sourceID <= 0 || destID <= 0 || sourceID > m->n2n_m || destID > m->n2n_m) {
        return 0.0;
    }
    return m->n2n_matrix[(int64_t)(sourceID - 1) * m->n2n_m + (destID - 1)] * nd->rxprob[dst];
}

/* RadioPacket.java:67-75 : 32 us per hex character */
int64_t orc_air_time_us(int64_t hex_length) { return hex_length * 32; }

/* Simulator.java:321-335 / 337-350 */
void orc_event_times(int64_t start_us, int64_t air_us, int64_t current_time, int64_t *t_start, int64_t *t_end)
{
    int64_t packetTime = start_us;
    if (packetTime < current_time) {
        packetTime = current_time;
    }
    *t_start = packetTime;
    *t_end = packetTime + air_us;
}

void orc_fill_packet(const orc_nodes_t *nd, int32_t src, int64_t start_us, int64_t air_us, orc_packet_t *out)
{
    /* RadioPacket.java:46-52 copies txpower and channel from the source radio */
    out->src = src;
    out->channel = nd->channel[src];
    out->x = nd->x[src];
    out->y = nd->y[src];
    out->z = nd->z[src];
    out->txpower = nd->txpower[src];
    out->txprob = nd->txprob[src];
    out->start_us = start_us;
    out->air_us = air_us;
}

void orc_model_defaults(orc_model_t *m, int32_t kind)
{
    memset(m, 0, sizeof(*m));
    m->kind = kind;
    m->udgm_success_ratio_tx = 1.0;     /* UDGMRadioMedium.java:18 */
    m->udgm_success_ratio_rx = 1.0;     /* :20 */
    m->udgm_transmission_range = 50.0;  /* :22 */
    m->udgm_interference_range = 100.0; /* :24 */
    m->const_range = 100.0;             /* UDGMConstantLossRadioMedium.java:8 */
    m->ld_pl0_db = 40.0;
    m->ld_exponent = 3.0;
    m->ld_d0 = 1.0;
    m->ld_sigma_db = 0.0;
    m->ld_clip = 3.0;
    m->ld_seed = 0;
    m->ld_sensitivity_dbm = -95.0;
    m->ld_noise_dbm = -100.0;           /* AbstractRadioMedium.java:38 base RSSI */
    m->ld_capture_db = 3.0;
    m->ld_ifloor_dbm = -110.0;
    m->ld_flags = 0;
}

/* ------------------------------------------------------------------ extension math */
/* DESIGN.md "Extension spec" E-math.  Only + - * / sqrt floor and integer bit operations. */

static inline uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

double orc_det_log2(double x)
{
    uint64_t bits = d2u(x);
    int e = (int)((bits >> 52) & 0x7FF) - 1023;
    double m = u2d((bits & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL);
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e = e + 1;
    }
    double f = (m - 1.0) / (m + 1.0);
    double s = f * f;
    double q = 1.0 / 23.0;
    q = q * s + 1.0 / 21.0;
    q = q * s + 1.0 / 19.0;
    q = q * s + 1.0 / 17.0;
    q = q * s + 1.0 / 15.0;
    q = q * s + 1.0 / 13.0;
    q = q * s + 1.0 / 11.0;
    q = q * s + 1.0 / 9.0;
    q = q * s + 1.0 / 7.0;
    q = q * s + 1.0 / 5.0;
    q = q * s + 1.0 / 3.0;
    q = q * s;
    double r = f + f * q;
    return (double)e + (2.0 * r) * 1.4426950408889634;
}

double orc_det_exp2(double y)
{
    if (y != y) return y;
    if (!(y >= -1022.0)) return 0.0;
    if (y > 1023.0) return INFINITY;
    double k = floor(y + 0.5);
    double r = y - k;
    double t = r * 0.6931471805599453;
    double q = 1.0 / 6227020800.0;
    q = q * t + 1.0 / 479001600.0;
    q = q * t + 1.0 / 39916800.0;
    q = q * t + 1.0 / 3628800.0;
    q = q * t + 1.0 / 362880.0;
    q = q * t + 1.0 / 40320.0;
    q = q * t + 1.0 / 5040.0;
    q = q * t + 1.0 / 720.0;
    q = q * t + 1.0 / 120.0;
    q = q * t + 1.0 / 24.0;
    q = q * t + 1.0 / 6.0;
    q = q * t + 0.5;
    q = q * t + 1.0;
    q = q * t + 1.0;
    double scale = u2d((uint64_t)((int64_t)k + 1023) << 52);
    return q * scale;
}

double orc_det_log10(double x) { return orc_det_log2(x) * 0.30102999566398120; }
double orc_det_pow10(double y) { return orc_det_exp2(y * 3.3219280948873622); }

/* Acklam's rational approximation of the standard normal quantile (rel. err 1.2e-9) */
double orc_det_normal(double u)
{
    const double a1 = -3.969683028665376e+01, a2 = 2.209460984245205e+02, a3 = -2.759285104469687e+02,
                 a4 = 1.383577518672690e+02, a5 = -3.066479806614716e+01, a6 = 2.506628277459239e+00;
    const double b1 = -5.447609879822406e+01, b2 = 1.615858368580409e+02, b3 = -1.556989798598866e+02,
                 b4 = 6.680131188771972e+01, b5 = -1.328068155288572e+01;
    const double c1 = -7.784894002430293e-03, c2 = -3.223964580411365e-01, c3 = -2.400758277161838e+00,
                 c4 = -2.549732539343734e+00, c5 = 4.374664141464968e+00, c6 = 2.938163982698783e+00;
    const double d1 = 7.784695709041462e-03, d2 = 3.224671290700398e-01, d3 = 2.445134137142996e+00,
                 d4 = 3.754408661907416e+00;
    if (u < 0.02425) {
        double q = sqrt(-2.0 * (orc_det_log2(u) * 0.6931471805599453));
        return (((((c1 * q + c2) * q + c3) * q + c4) * q + c5) * q + c6) /
               ((((d1 * q + d2) * q + d3) * q + d4) * q + 1.0);
    }
    if (u <= 0.97575) {
        double q = u - 0.5;
        double r = q * q;
        return (((((a1 * r + a2) * r + a3) * r + a4) * r + a5) * r + a6) * q /
               (((((b1 * r + b2) * r + b3) * r + b4) * r + b5) * r + 1.0);
    }
    double q = sqrt(-2.0 * (orc_det_log2(1.0 - u) * 0.6931471805599453));
    return -((((((c1 * q + c2) * q + c3) * q + c4) * q + c5) * q + c6) /
             ((((d1 * q + d2) * q + d3) * q + d4) * q + 1.0));
}

static inline uint64_t mix64(uint64_t z)
{
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

uint64_t orc_shadow_hash(uint64_t seed, uint32_t a, uint32_t b)
{
    uint32_t lo = a < b ? a : b;
    uint32_t hi = a < b ? b : a;
    uint64_t key = ((uint64_t)lo << 32) | (uint64_t)hi;
    return mix64(mix64(seed + 0x9E3779B97F4A7C15ULL) ^ key);
}

double orc_shadow_gauss(const orc_model_t *m, uint32_t a, uint32_t b)
{
    uint64_t h = orc_shadow_hash(m->ld_seed, a, b);
    double u = ((double)(h >> 12) + 0.5) * 0x1.0p-52;
    double g = orc_det_normal(u);
    if (g > m->ld_clip) g = m->ld_clip;
    if (g < -m->ld_clip) g = -m->ld_clip;
    return g;
}

double orc_logdist_rssi(const orc_model_t *m, const orc_packet_t *p, const orc_nodes_t *nd, int32_t dst)
{
    double d = orc_distance(p->x, p->y, p->z, nd->x[dst], nd->y[dst], nd->z[dst]);
    double dd = (d > m->ld_d0) ? d : m->ld_d0;
    double t1 = p->txpower - m->ld_pl0_db;
    double t2 = 10.0 * m->ld_exponent;
    double l = orc_det_log10(dd / m->ld_d0);
    double rssi = t1 - t2 * l;
    if (m->ld_sigma_db > 0.0) {
        rssi = rssi - m->ld_sigma_db * orc_shadow_gauss(m, (uint32_t)p->src, (uint32_t)dst);
    }
    return rssi;
}

/* Q80 fixed point: exact, order-independent interference sums */
typedef unsigned __int128 u128;

static u128 to_fixed(double lin)
{
    if (!(lin > 0.0)) return 0;
    uint64_t bits = d2u(lin);
    int ex = (int)((bits >> 52) & 0x7FF);
    if (ex == 0x7FF) return ((u128)1 << 127) - 1;
    uint64_t man = bits & 0x000FFFFFFFFFFFFFULL;
    if (ex == 0) return 0; /* subnormal: below 2^-80 anyway */
    man |= 0x0010000000000000ULL;
    int shift = ex - 1075 + 80; /* value = man * 2^(ex-1075) ; Q80 = value * 2^80 */
    if (shift >= 0) {
        if (shift > 74) return ((u128)1 << 127) - 1;
        return (u128)man << shift;
    }
    if (-shift >= 64) return 0;
    return (u128)(man >> (-shift));
}

static double from_fixed(u128 q)
{
    if (q == 0) return 0.0;
    int top = 127;
    while (!((q >> top) & 1)) top--;
    /* keep 53 bits, round to nearest even on the rest */
    if (top <= 52) {
        return ldexp((double)(uint64_t)q, -80);
    }
    int drop = top - 52;
    uint64_t keep = (uint64_t)(q >> drop);
    u128 rem = q & (((u128)1 << drop) - 1);
    u128 half = (u128)1 << (drop - 1);
    if (rem > half || (rem == half && (keep & 1))) keep++;
    return ldexp((double)keep, drop - 80); /* keep may be 2^53: still exact */
}

double orc_fixed_roundtrip(double lin) { return from_fixed(to_fixed(lin)); }

static inline int overlaps(const orc_packet_t *i, const orc_packet_t *k)
{
    return k->start_us < i->start_us + i->air_us && k->start_us + k->air_us > i->start_us;
}

/* ------------------------------------------------------------------ the pass */

typedef struct {
    int32_t *pkt, *dst;
    uint8_t *verdict;
    double *rssi, *sinr;
    int64_t cap, n;
} sink_t;

static inline void emit(sink_t *s, int32_t pkt, int32_t dst, int verdict, double rssi, double sinr)
{
    if (s->n < s->cap) {
        if (s->pkt) s->pkt[s->n] = pkt;
        if (s->dst) s->dst[s->n] = dst;
        if (s->verdict) s->verdict[s->n] = (uint8_t)verdict;
        if (s->rssi) s->rssi[s->n] = rssi;
        if (s->sinr) s->sinr[s->n] = sinr;
    }
    s->n++;
}

/* logdist: SINR of wanted frame `wi` at receiver j; returns 1 if the link is collided */
static int logdist_collided(const orc_model_t *m, const orc_nodes_t *nd, const orc_packet_t *active,
                            int32_t n_active, int32_t wi, int32_t j, double rssi, double *sinr_out)
{
    const orc_packet_t *w = &active[wi];
    u128 acc = 0;
    int half_duplex = 0;
    for (int32_t k = 0; k < n_active; k++) {
        if (k == wi) continue;
        const orc_packet_t *p = &active[k];
        if (!overlaps(w, p)) continue;
        if (p->src == j) {
            half_duplex = 1;
            continue;
        }
        if (p->channel != w->channel) continue;
        double r = orc_logdist_rssi(m, p, nd, j);
        if (!(r >= m->ld_ifloor_dbm)) continue;
        acc += to_fixed(orc_det_pow10(r / 10.0));
    }
    double I = from_fixed(acc);
    double noise_lin = orc_det_pow10(m->ld_noise_dbm / 10.0);
    double denom = I + noise_lin;
    double sinr = rssi - 10.0 * orc_det_log10(denom);
    *sinr_out = sinr;
    return half_duplex || !(sinr >= m->ld_capture_db);
}

/* One packet of the pass: the body of the reference's transmit() (UDGMRadioMedium.java:83-117 and siblings) for the
 * active frame `pi`.  rs == NULL: the caller evaluates packets out of order (orc_tick_mt), which is only the reference's
 * result when no java.util.Random draw happens -- a draw that would be needed sets *need_draw instead. */
static void packet_pass(const orc_model_t *m, const orc_nodes_t *nd, uint64_t *rs, int *need_draw,
                        const orc_packet_t *active, int32_t n_active, int32_t first_new, int32_t pi, sink_t *sp,
                        uint8_t *pkt_interference, int32_t *pkt_draws)
{
    sink_t s = *sp;
    {
        const orc_packet_t *p = &active[pi];
        int32_t rel = pi - first_new;
        int draws = 0;
        int interference = 0;

        if (m->kind == ORC_MODEL_UDGM || m->kind == ORC_MODEL_N2N || m->kind == ORC_MODEL_LOGDIST) {
            /* UDGMRadioMedium.java:87-92 ; N2NRadioMedium.java:43-48 */
            double txSuccess = (m->kind == ORC_MODEL_UDGM) ? orc_udgm_tx_probability(m, p) : p->txprob;
            if (txSuccess <= 0.0) {
                interference = 1;
            } else if (txSuccess < 1.0) {
                draws++;
                if (!rs) *need_draw = 1;
                else if (orc_jrandom_next_double(rs) > txSuccess) interference = 1;
            }
        }
        double pkt_rssi = p->txpower; /* UDGMRadioMedium.java:95 etc.: rssi = packet txpower */

        for (int32_t j = 0; j < nd->n; j++) {
            if (j == p->src) continue;                       /* node != source */
            if (!nd->enabled[j]) continue;                   /* radio.isEnabled() */
            if (nd->channel[j] != p->channel) continue;      /* getWirelessChannel() == channel */
            switch (m->kind) {
            case ORC_MODEL_NULL: /* NullRadioMedium.java:72 */
                emit(&s, rel, j, ORC_DELIVERED, pkt_rssi, 0.0);
                break;
            case ORC_MODEL_UDGM_CONST: { /* UDGMConstantLossRadioMedium.java:29-32, strict < */
                double distance = orc_distance(p->x, p->y, p->z, nd->x[j], nd->y[j], nd->z[j]);
                if (distance < m->const_range) emit(&s, rel, j, ORC_DELIVERED, pkt_rssi, 0.0);
                break;
            }
            case ORC_MODEL_UDGM:
            case ORC_MODEL_N2N: { /* UDGMRadioMedium.java:103-111 ; N2NRadioMedium.java:59-67 */
                double rxSuccess = (m->kind == ORC_MODEL_UDGM) ? orc_udgm_rx_probability(m, nd, p, j)
                                                               : orc_n2n_rx_probability(m, nd, p, j);
                if (rxSuccess <= 0.0) {
                    /* The receiver can not hear the sender */
                } else {
                    int failed = interference;
                    if (!failed && rxSuccess < 1.0) {
                        draws++;
                        if (!rs) *need_draw = 1;
                        else failed = orc_jrandom_next_double(rs) > rxSuccess;
                    }
                    emit(&s, rel, j, failed ? ORC_INTERFERED : ORC_DELIVERED, pkt_rssi, 0.0);
                }
                break;
            }
            case ORC_MODEL_LOGDIST: { /* extension */
                double rssi = orc_logdist_rssi(m, p, nd, j);
                if (!(rssi >= m->ld_sensitivity_dbm)) break;
                double rx = nd->rxprob[j];
                if (rx <= 0.0) break;
                double sinr = 0.0;
                int collided = 0;
                if (m->ld_flags & ORC_LD_SINR) {
                    collided = logdist_collided(m, nd, active, n_active, pi, j, rssi, &sinr);
                }
                int failed = interference || collided;
                if (!failed && rx < 1.0) {
                    draws++;
                    if (!rs) *need_draw = 1;
                    else failed = orc_jrandom_next_double(rs) > rx;
                }
                emit(&s, rel, j, failed ? ORC_INTERFERED : ORC_DELIVERED, rssi, sinr);
                break;
            }
            default:
                break;
            }
        }
        if (pkt_interference) pkt_interference[rel] = (uint8_t)interference;
        if (pkt_draws) pkt_draws[rel] = draws;
    }
    *sp = s;
}

int64_t orc_tick(const orc_model_t *m, const orc_nodes_t *nd, uint64_t *rng_state,
                 const orc_packet_t *active, int32_t n_active, int32_t first_new,
                 int32_t *out_pkt, int32_t *out_dst, uint8_t *out_verdict,
                 double *out_rssi, double *out_sinr, int64_t cap,
                 uint8_t *pkt_interference, int32_t *pkt_draws)
{
    sink_t s = { out_pkt, out_dst, out_verdict, out_rssi, out_sinr, cap, 0 };
    uint64_t local_state = 0;
    uint64_t *rs = rng_state ? rng_state : &local_state;
    int unused = 0;
    /* packets in arrival order, one after the other: the shared generator chains them */
    for (int32_t pi = first_new; pi < n_active; pi++)
        packet_pass(m, nd, rs, &unused, active, n_active, first_new, pi, &s, pkt_interference, pkt_draws);
    return s.n;
}

/* The same pass with the packets spread over `threads` threads, for ticks in which NO java.util.Random draw happens (every
 * probability that is looked at is 0 or 1): the packets are then independent of each other -- the only thing that chains them
 * in the reference is the shared generator -- and their links are written packet by packet in arrival order all the same.
 * Returns the number of heard links, or -2 if a draw would have been needed (the result is then not the reference's: use
 * orc_tick).  Full-size parity checks (tests/test_gpu_fullsize.py) need it: a whole configs[3] tick is 25 s on one thread. */
int64_t orc_tick_mt(const orc_model_t *m, const orc_nodes_t *nd, const orc_packet_t *active, int32_t n_active, int32_t first_new,
                    int32_t threads, int32_t *out_pkt, int32_t *out_dst, uint8_t *out_verdict, double *out_rssi, double *out_sinr,
                    int64_t cap, uint8_t *pkt_interference)
{
    const int32_t n_new = n_active - first_new;
    if (n_new <= 0) return 0;
    if (threads < 1) threads = 1;
    /* every packet into its own small buffer (a frame is heard by tens to hundreds of receivers; grown when needed) */
    typedef struct { int32_t *dst; uint8_t *verdict; double *rssi, *sinr; int64_t n, cap; } part_t;
    part_t *parts = (part_t *)calloc((size_t)n_new, sizeof(part_t));
    int need_draw = 0, oom = 0;
    if (!parts) return -1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
    for (int32_t rel = 0; rel < n_new; rel++) {
        part_t *pt = &parts[rel];
        int my_draw = 0;
        for (int64_t want = 256;; want = pt->n + 64) {
            free(pt->dst); free(pt->verdict); free(pt->rssi); free(pt->sinr);
            pt->cap = want;
            pt->dst = (int32_t *)malloc((size_t)want * 4);
            pt->verdict = (uint8_t *)malloc((size_t)want);
            pt->rssi = (double *)malloc((size_t)want * 8);
            pt->sinr = (double *)malloc((size_t)want * 8);
            if (!pt->dst || !pt->verdict || !pt->rssi || !pt->sinr) { oom = 1; break; }
            sink_t s = { NULL, pt->dst, pt->verdict, pt->rssi, pt->sinr, want, 0 };
            packet_pass(m, nd, NULL, &my_draw, active, n_active, first_new, first_new + rel, &s, pkt_interference, NULL);
            pt->n = s.n;
            if (s.n <= want) break; /* everything fitted; otherwise once more with room for all of it */
        }
        if (my_draw) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            need_draw = 1;
        }
    }
    int64_t total = 0;
    for (int32_t rel = 0; rel < n_new; rel++) {
        const part_t *pt = &parts[rel];
        for (int64_t i = 0; i < pt->n && !oom; i++, total++) {
            if (total >= cap) continue;
            if (out_pkt) out_pkt[total] = rel;
            if (out_dst) out_dst[total] = pt->dst[i];
            if (out_verdict) out_verdict[total] = pt->verdict[i];
            if (out_rssi) out_rssi[total] = pt->rssi[i];
            if (out_sinr) out_sinr[total] = pt->sinr[i];
        }
        free(pt->dst); free(pt->verdict); free(pt->rssi); free(pt->sinr);
    }
    free(parts);
    if (oom) return -1;
    return need_draw ? -2 : total;
}

int32_t orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int64_t orc_count_links(const orc_model_t *m, const orc_nodes_t *nd,
                        const orc_packet_t *active, int32_t n_active, int32_t first_new,
                        int32_t threads, int64_t *delivered)
{
    int64_t heard = 0, deliv = 0;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4) reduction(+ : heard, deliv)
#endif
    for (int32_t pi = first_new; pi < n_active; pi++) {
        const orc_packet_t *p = &active[pi];
        int interference = 0;
        if (m->kind == ORC_MODEL_UDGM) interference = orc_udgm_tx_probability(m, p) <= 0.0;
        if (m->kind == ORC_MODEL_N2N || m->kind == ORC_MODEL_LOGDIST) interference = p->txprob <= 0.0;
        for (int32_t j = 0; j < nd->n; j++) {
            if (j == p->src) continue;
            if (!nd->enabled[j]) continue;
            if (nd->channel[j] != p->channel) continue;
            switch (m->kind) {
            case ORC_MODEL_NULL:
                heard++; deliv++;
                break;
            case ORC_MODEL_UDGM_CONST:
                if (orc_distance(p->x, p->y, p->z, nd->x[j], nd->y[j], nd->z[j]) < m->const_range) { heard++; deliv++; }
                break;
            case ORC_MODEL_UDGM:
            case ORC_MODEL_N2N: {
                double rx = (m->kind == ORC_MODEL_UDGM) ? orc_udgm_rx_probability(m, nd, p, j)
                                                        : orc_n2n_rx_probability(m, nd, p, j);
                if (rx > 0.0) { heard++; if (!interference) deliv++; }
                break;
            }
            case ORC_MODEL_LOGDIST: {
                double rssi = orc_logdist_rssi(m, p, nd, j);
                if (!(rssi >= m->ld_sensitivity_dbm)) break;
                if (nd->rxprob[j] <= 0.0) break;
                double sinr = 0.0;
                int collided = 0;
                if (m->ld_flags & ORC_LD_SINR) collided = logdist_collided(m, nd, active, n_active, pi, j, rssi, &sinr);
                heard++;
                if (!interference && !collided) deliv++;
                break;
            }
            default:
                break;
            }
        }
    }
    if (delivered) *delivered = deliv;
    return heard;
}
