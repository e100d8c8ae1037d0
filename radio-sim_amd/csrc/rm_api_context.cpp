// rm_api_context.cpp -- C ABI (include/radiomedium_hip.h): context life cycle, model, derived constants, generator state, exported host helpers.
#include "rm_host.hpp"

using namespace rmh;

namespace rmh {

thread_local std::string g_err;
} // namespace rmh
namespace rm {
thread_local KernelProbe g_probe = {nullptr, nullptr};
} // namespace rm
namespace rmh {

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

const char *model_name(int kind)
{
    switch (kind) {
    case RM_MODEL_NULL: return "Null radio medium - just forwards incoming packets to all other nodes"; // NullRadioMedium.java:44
    case RM_MODEL_UDGM: return "UDGM Radio Medium";                                                   // UDGMRadioMedium.java:28
    case RM_MODEL_UDGM_CONST: return "UDGM Constant Loss Radio Medium";                               // UDGMConstantLossRadioMedium.java:12
    case RM_MODEL_N2N: return "Matrix Radio Medium";                                                  // N2NRadioMedium.java:17
    case RM_MODEL_LOGDIST: return "Log-distance SINR Radio Medium (MI355X)";
    default: return "?";
    }
}

bool is_sinr(const rm_context *c) { return c->params.kind == RM_MODEL_LOGDIST && (c->params.flags & RM_LD_SINR); }

bool part_spatial(const rm_context *c) { return c->sp_parts > 1; }
int part_first(const rm_context *c) { return (part_spatial(c) || c->rx_count < 0) ? 0 : c->rx_first; }
int part_count(const rm_context *c) { return part_spatial(c) ? int(c->sp_nodes.size()) : (c->rx_count < 0 ? c->n : c->rx_count); }
int pos_span(const rm_context *c) { return part_spatial(c) ? c->n : part_count(c); }

bool frac(double p) { return p > 0.0 && p < 1.0; }

// can a java.util.Random draw ever be consumed with the current model + node table?
// (the O(N) scan of the probabilities is cached until the node table changes)
bool maybe_draws(rm_context *c)
{
    const int k = c->params.kind;
    if (k == RM_MODEL_NULL || k == RM_MODEL_UDGM_CONST) return false;
    if (k == RM_MODEL_N2N) return true;
    if (c->tick_frac_records) return true; // a record of this tick carries its own fractional txProbability
    if (k == RM_MODEL_UDGM && c->params.udgm_success_ratio_rx != 1.0) return true;
    if (c->frac_probs < 0) {
        c->frac_probs = 0;
        for (int i = 0; i < c->n; ++i)
            if (frac(c->rxprob[i]) || frac(c->txprob[i])) {
                c->frac_probs = 1;
                break;
            }
    }
    return c->frac_probs == 1;
}

int validate_model(const rm_model_params *p)
{
    if (p->kind < RM_MODEL_NULL || p->kind > RM_MODEL_LOGDIST) return fail(RM_ERR_INVALID, "unknown model kind");
    if (p->kind == RM_MODEL_LOGDIST) {
        if (!(p->ld_d0 > 0.0) || !(p->ld_exponent >= 0.0) || !(p->ld_sigma_db >= 0.0) || !(p->ld_clip >= 0.0))
            return fail(RM_ERR_INVALID, "logdist: need d0 > 0, exponent >= 0, sigma >= 0, clip >= 0");
    }
    return RM_OK;
}

void recompute_frame(rm_context *c)
{
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (int i = 0; i < c->n; ++i) {
        const double v[3] = {c->x[i], c->y[i], c->z[i]};
        for (int a = 0; a < 3; ++a) {
            if (i == 0 || v[a] < lo[a]) lo[a] = v[a];
            if (i == 0 || v[a] > hi[a]) hi[a] = v[a];
        }
    }
    double bound = 0;
    for (int a = 0; a < 3; ++a) {
        c->org[a] = 0.5 * (lo[a] + hi[a]);
        bound = std::max(bound, std::max(hi[a] - c->org[a], c->org[a] - lo[a]));
    }
    c->coord_bound = bound * (1.0 + 1e-9) + 1e-300;
    // fp32 frame: each coordinate is off by at most 2^-24 * bound; see DESIGN.md "Pre-filter"
    c->f32_slack = 4.0 * std::sqrt(3.0) * std::ldexp(1.0, -24) * c->coord_bound;
    c->prefilter_dirty = true;
}

rm::ModelDev model_dev(const rm_context *c)
{
    // planned once per tick of a batch (and more than once): the derived values (a software pow10, the mixed seed) are
    // kept as long as everything they come from is unchanged
    struct Key {
        rm_model_params p;
        double org[3], coord_bound, f32_slack;
        const void *n2n, *shadow;
        int n2n_m;
    };
    Key key;
    std::memset(&key, 0, sizeof(key)); // (padding bytes take part in the comparison)
    key.p = c->params;
    key.org[0] = c->org[0]; key.org[1] = c->org[1]; key.org[2] = c->org[2];
    key.coord_bound = c->coord_bound;
    key.f32_slack = c->f32_slack;
    key.n2n = c->d_n2n.p;
    key.shadow = c->shadow_tbl_valid ? c->d_shadow_tbl.p : nullptr;
    key.n2n_m = c->n2n_m;
    static_assert(sizeof(Key) <= sizeof(c->mdev_key), "model key buffer");
    if (c->mdev_valid && std::memcmp(&key, c->mdev_key, sizeof(key)) == 0) return c->mdev;
    rm::ModelDev m{};
    const rm_model_params &p = c->params;
    m.kind = p.kind;
    m.flags = p.flags;
    m.udgm_ratio_rx = p.udgm_success_ratio_rx;
    m.udgm_range = p.udgm_transmission_range;
    m.const_range = p.const_range;
    m.n2n = c->d_n2n.p;
    m.n2n_m = c->n2n_m;
    m.ld_pl0 = p.ld_pl0_db;
    m.ld_exp = p.ld_exponent;
    m.ld_d0 = p.ld_d0;
    m.ld_sigma = p.ld_sigma_db;
    m.ld_clip = p.ld_clip;
    m.ld_seed_mixed = rm::host_mix64(p.ld_seed + 0x9E3779B97F4A7C15ull);
    m.ld_sens = p.ld_sensitivity_dbm;
    m.ld_noise = p.ld_noise_dbm;
    m.ld_capture = p.ld_capture_db;
    m.ld_ifloor = p.ld_ifloor_dbm;
    m.ld_noise_lin = rm::host_det_pow10(p.ld_noise_dbm / 10.0);
    m.ld_level = p.ld_sensitivity_dbm;
    m.ld_cut_scale = (p.ld_exponent > 0.0) ? 3.3219280948873622 / (10.0 * p.ld_exponent) : 0.0;
    if ((p.flags & RM_LD_SINR) && p.ld_ifloor_dbm < m.ld_level) m.ld_level = p.ld_ifloor_dbm;
    m.org_x = c->org[0];
    m.org_y = c->org[1];
    m.org_z = c->org[2];
    m.coord_bound = c->coord_bound;
    m.f32_slack = c->f32_slack;
    m.geo_cut = -1.0;
    m.shadow_tbl = c->shadow_tbl_valid ? c->d_shadow_tbl.p : nullptr;
    if (p.kind == RM_MODEL_UDGM) {
        const double r = std::fabs(p.udgm_transmission_range);
        m.geo_cut = (r == 0.0) ? -1.0 : r * (1.0 + 1e-9); // ratio > 1.0 -> unheard, d == range is in
    } else if (p.kind == RM_MODEL_UDGM_CONST) {
        m.geo_cut = (p.const_range > 0.0) ? p.const_range * (1.0 + 1e-9) : -1.0; // strict distance < range
    }
    std::memcpy(c->mdev_key, &key, sizeof(key));
    c->mdev = m;
    c->mdev_valid = true;
    return m;
}

rm::NodesDev nodes_dev(rm_context *c)
{
    rm::NodesDev nd{};
    nd.n = c->n;
    nd.srec = c->d_srec.p;
    nd.sx = c->d_x.p;
    nd.sy = c->d_y.p;
    nd.sz = c->d_z.p;
    nd.stxpower = c->d_txpower.p;
    nd.stxprob = c->d_txprob.p;
    nd.srxprob = c->d_rxprob_node.p;
    nd.schannel = c->d_channel.p;
    nd.sint_id = c->d_int_id.p;
    nd.senabled = c->d_enabled.p;
    nd.n_rx = c->n_rx;
    nd.x = c->d_rx_x.p;
    nd.y = c->d_rx_y.p;
    nd.z = c->d_rx_z.p;
    nd.rxprob = c->d_rx_rxprob.p;
    nd.channel = c->d_rx_channel.p;
    nd.int_id = c->d_rx_int_id.p;
    nd.orig = c->d_rx_orig.p;
    nd.enabled = c->d_rx_enabled.p;
    nd.rec = c->d_rx_rec.p;
    static const bool no_rec32 = std::getenv("RM_NO_REC32") != nullptr;
    nd.rec32 = no_rec32 ? nullptr : c->d_rx_rec32.p;
    nd.pos_of = c->d_pos_of.p;
    nd.rx_first = part_first(c);
    nd.pos_span = pos_span(c);
    nd.rxf = c->d_rxf.p;
    nd.bbox_xy = c->d_bbox_xy.p;
    nd.bbox_z = c->d_bbox_z.p;
    nd.wg_box_xy = c->d_wg_box_xy.p;
    nd.grp_chmask = c->d_grp_chmask.p;
    nd.wg_chmask = c->d_wg_chmask.p;
    nd.wg_box_z = c->d_wg_box_z.p;
    return nd;
}

bool is_geometric(const rm_context *c)
{
    const int k = c->params.kind;
    return k == RM_MODEL_UDGM || k == RM_MODEL_UDGM_CONST || k == RM_MODEL_LOGDIST;
}

// Second-level filter of the shadowed log-distance medium: for a link at rho = d^2/cut^2 the
// deviate may be at most x(rho) = (5 n log10(1/rho) - sigma*clip)/sigma for the link to reach the
// candidate level, i.e. the hash's uniform at most Phi(x).  One conservative 32-bit threshold per
// bin (lower bin edge, rho padded by kShadowPad, x padded for the quantile approximation).
int build_shadow_table(rm_context *c)
{
    const rm_model_params &p = c->params;
    c->shadow_tbl_valid = false;
    if (p.kind != RM_MODEL_LOGDIST || !(p.ld_sigma_db > 0.0) || !(p.ld_exponent > 0.0)) return RM_OK;
    std::vector<uint32_t> tbl(rm::kShadowBins);
    for (int b = 0; b < rm::kShadowBins; ++b) {
        const double rho = (double(b) / rm::kShadowBins) * (1.0 - rm::kShadowPad);
        double umax = 1.0;
        if (rho > 0.0) {
            const double margin = 5.0 * p.ld_exponent * std::log10(1.0 / rho) - p.ld_sigma_db * p.ld_clip;
            const double x = margin / p.ld_sigma_db + 1e-6;
            if (x < p.ld_clip) umax = 0.5 * std::erfc(-x / std::sqrt(2.0));
        }
        const double v = std::floor(umax * 4294967296.0) + 2.0;
        tbl[b] = v >= 4294967295.0 ? 0xFFFFFFFFu : uint32_t(v);
    }
    RM_HIP(c->d_shadow_tbl.ensure(rm::kShadowBins));
    RM_HIP(hipMemcpyAsync(c->d_shadow_tbl.p, tbl.data(), tbl.size() * 4, hipMemcpyHostToDevice, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    c->shadow_tbl_valid = true;
    return RM_OK;
}

} // namespace rmh

void TickSlot::release_all()
{
    d_tx.release(); d_p_txf.release(); d_p_ch.release(); d_p_src.release(); d_p_inv.release();
    d_cnt.release(); d_off.release(); d_dense_mask.release(); d_slot_tot.release(); d_slot_off.release();
    d_fl_map.release(); d_fl_lb.release(); d_slot_off_loc.release();
    d_counters.release(); d_shards.release(); d_cursor.release(); d_cand_tot.release(); d_seg_off.release(); d_a_e.release();
    d_st_pkt.release(); d_st_dst.release(); d_st_next.release(); d_head.release(); d_st_blk.release(); d_st_aux.release();
    d_st_lin.release(); d_st_sinr.release(); d_st_prob.release(); d_st_orig.release(); d_st_flags.release(); d_st_coll.release();
    d_out_pkt.release(); d_out_dst.release(); d_a_pkt.release(); d_a_dst.release(); d_out_verdict.release(); d_pkt_interf.release();
    d_a_verdict.release(); d_out_rssi.release(); d_out_sinr.release(); d_out_prob.release(); d_a_rssi.release(); d_a_sinr.release();
    d_a_prob.release(); d_draw_scan.release(); d_scan_block.release(); d_scan_xyzr.release(); d_scan_ch.release(); d_sg_cnt.release(); d_sg_bxyzr.release(); d_sg_bci.release(); d_sg_every.release();
    d_self_next.release(); d_self_slot.release(); sg_clean[0] = sg_clean[1] = false; d_pkt_rng.release(); d_pkt_draw_cnt.release(); d_all_cnt.release();
    alloc_cap = 0;
    alloc_feat = 0;
    have_result = false;
}

extern "C" {

int rm_abi_version(void) { return RM_ABI_VERSION; }

const char *rm_last_error(void) { return g_err.c_str(); }

int rm_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(RM_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n;
}

int rm_create(int device_ordinal, rm_context **out)
{
    if (!out) return fail(RM_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(RM_ERR_NO_DEVICE, std::string("no HIP device available (") +
                                          (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                                          "); this engine has no CPU fallback");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(RM_ERR_INVALID, "device ordinal out of range");
    RM_HIP(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    RM_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RM_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    rm_context *c = new rm_context();
    c->device = device_ordinal;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(RM_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    c->own_stream = true;
    if (const char *g = std::getenv("RM_GRAPH")) c->use_graphs = (g[0] == '1');
    rm_model_defaults(&c->params, RM_MODEL_NULL); // Main.java:66-70: NullRadioMedium is the default
    *out = c;
    return RM_OK;
}

void rm_destroy(rm_context *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &g : c->graphs) (void)hipGraphExecDestroy(g.exec);
    for (auto &sm : c->ev_pool)
        for (auto &k : sm.k) {
            (void)hipEventDestroy(k.a);
            (void)hipEventDestroy(k.b);
        }
    c->d_x.release(); c->d_y.release(); c->d_z.release(); c->d_txpower.release(); c->d_txprob.release(); c->d_rxprob_node.release();
    c->d_channel.release(); c->d_int_id.release(); c->d_srec.release(); c->d_rx_x.release(); c->d_rx_y.release(); c->d_rx_z.release();
    c->d_rx_rxprob.release(); c->d_rx_channel.release(); c->d_rx_int_id.release(); c->d_rx_orig.release();
    c->d_pos_of.release(); c->d_rx_enabled.release(); c->d_rx_rec.release(); c->d_rx_rec32.release(); c->d_rxf.release(); c->d_bbox_xy.release();
    c->d_bbox_z.release(); c->d_wg_box_xy.release(); c->d_wg_box_z.release(); c->d_grp_chmask.release(); c->d_wg_chmask.release();
    c->d_n2n.release(); c->d_shadow_tbl.release(); c->d_air.release(); c->d_air_alt.release(); c->d_cull_ring.release(); c->d_rng.release(); c->d_ticks.release(); c->d_near_list.release(); c->d_near_cnt.release();
    c->air.pool.release(); c->air.head.release(); c->air.tail.release(); c->air.mark.release(); c->air.bad.release();
    {
        rm_context::Overlap &o = c->ov;
        o.fr_f.release(); o.e_f.release(); o.fr_m.release(); o.e_m.release(); o.fr_t.release(); o.e_t.release();
        o.fr_bin.release(); o.bin_cnt.release(); o.bin_off.release(); o.block_sum.release(); o.every.release(); o.misc.release();
        o.pair_tail.release(); o.items.release(); o.self_next.release(); o.defer.release(); o.slot_first.release(); o.ticks.release(); o.pairs.release();
        for (int g = 0; g < 2; ++g) {
            if (o.h_ev[g]) (void)hipEventDestroy(o.h_ev[g]);
            if (o.h_desc[g]) (void)hipHostFree(o.h_desc[g]);
            if (o.h_flag_ev[g]) (void)hipEventDestroy(o.h_flag_ev[g]);
        }
        if (o.h_flag) (void)hipHostFree(o.h_flag);
    }
    c->d_patch.release();
    c->d_enabled.release();
    c->d_member.release(); c->d_draw_nodes.release(); c->d_all_off.release(); c->d_all_nodes.release();
    (void)rm_comm_destroy(c);
    c->d_dist_mine.release(); c->d_dist_all.release(); c->d_dist_idx.release(); c->d_dist_stage.release();
    (void)rm_events_disable(c);
    c->release_all();
    for (auto &sl : c->extra_slots) sl->release_all();
    for (int g = 0; g < 2; ++g) {
        if (c->h_ticks_ev[g]) (void)hipEventDestroy(c->h_ticks_ev[g]);
        if (c->h_ticks[g]) (void)hipHostFree(c->h_ticks[g]);
    }
    if (c->h_transmit) (void)hipHostFree(c->h_transmit);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_pack) (void)hipHostFree(c->h_pack);
    c->d_pack.release();
    c->d_pack_done.release();
    for (int g = 0; g < 2; ++g) {
        if (c->h_tx_ev[g]) (void)hipEventDestroy(c->h_tx_ev[g]);
        if (c->h_tx[g]) (void)hipHostFree(c->h_tx[g]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *rm_get_name(const rm_context *c) { return c ? model_name(c->params.kind) : ""; }

int rm_set_stream(rm_context *c, void *hip_stream)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    RM_HIP(hipStreamSynchronize(c->stream));
    if (c->own_stream) RM_HIP(hipStreamDestroy(c->stream));
    c->stream = static_cast<hipStream_t>(hip_stream);
    c->own_stream = false;
    return RM_OK;
}

void rm_model_defaults(rm_model_params *p, int32_t kind)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->kind = kind;
    p->udgm_success_ratio_tx = 1.0;     // UDGMRadioMedium.java:18
    p->udgm_success_ratio_rx = 1.0;     // :20
    p->udgm_transmission_range = 50.0;  // :22
    p->udgm_interference_range = 100.0; // :24
    p->const_range = 100.0;             // UDGMConstantLossRadioMedium.java:8
    p->ld_pl0_db = 40.0;
    p->ld_exponent = 3.0;
    p->ld_d0 = 1.0;
    p->ld_sigma_db = 0.0;
    p->ld_clip = 3.0;
    p->ld_seed = 0;
    p->ld_sensitivity_dbm = -95.0;
    p->ld_noise_dbm = -100.0;           // AbstractRadioMedium.java:38
    p->ld_capture_db = 3.0;
    p->ld_ifloor_dbm = -110.0;
}

int rm_set_model(rm_context *c, const rm_model_params *p)
{
    if (!c || !p) return fail(RM_ERR_INVALID, "NULL argument");
    RM_TRY(validate_model(p));
    RM_TRY(ev_flush_append(c)); // (the tick before was evaluated by the old medium: its append does not wait for the next drain)
    const bool was_geo = is_geometric(c);
    c->params = *p;
    if (was_geo != is_geometric(c)) c->rx_dirty = true;
    c->prefilter_dirty = true;
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(build_shadow_table(c));
    c->air_batches.clear();
    c->air_head = c->air_tail = 0;
    c->air.valid = false;
    c->pending.clear();
    return RM_OK;
}

int rm_get_model(const rm_context *c, rm_model_params *out)
{
    if (!c || !out) return fail(RM_ERR_INVALID, "NULL argument");
    *out = c->params;
    return RM_OK;
}

int rm_set_n2n_matrix(rm_context *c, int32_t m, const double *row_major)
{
    if (!c || m < 0 || (m > 0 && !row_major)) return fail(RM_ERR_INVALID, "bad matrix");
    RM_HIP(hipSetDevice(c->device));
    c->n2n_m = m;
    if (m == 0) {
        c->d_n2n.release();
        return RM_OK;
    }
    RM_HIP(c->d_n2n.ensure(size_t(m) * m));
    RM_HIP(hipMemcpyAsync(c->d_n2n.p, row_major, size_t(m) * m * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    return RM_OK;
}

int rm_set_base_rssi(rm_context *c, double rssi)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    c->base_rssi = rssi;
    return RM_OK;
}

double rm_get_base_rssi(const rm_context *c, int32_t node)
{
    (void)node; // AbstractRadioMedium.java:46-48: the same value for every node
    return c ? c->base_rssi : -100.0;
}

int rm_seed(rm_context *c, int64_t seed)
{
    return rm_set_rng_state(c, (uint64_t(seed) ^ 0x5DEECE66Dull) & ((1ull << 48) - 1));
}

int rm_set_rng_state(rm_context *c, uint64_t state48)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    RM_HIP(hipSetDevice(c->device));
    RM_HIP(c->d_rng.ensure(1));
    state48 &= (1ull << 48) - 1;
    RM_HIP(hipMemcpyAsync(c->d_rng.p, &state48, 8, hipMemcpyHostToDevice, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    return RM_OK;
}

int rm_get_rng_state(rm_context *c, uint64_t *state48)
{
    if (!c || !state48) return fail(RM_ERR_INVALID, "NULL argument");
    RM_HIP(hipSetDevice(c->device));
    if (!c->d_rng.p) RM_TRY(rm_seed(c, 0));
    RM_HIP(hipMemcpyAsync(state48, c->d_rng.p, 8, hipMemcpyDeviceToHost, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    return RM_OK;
}

int rm_set_link_capacity(rm_context *c, uint32_t max_links)
{
    if (!c || max_links == 0) return fail(RM_ERR_INVALID, "bad capacity");
    c->cap = max_links;
    return RM_OK;
}

int rm_air_list_stats(const rm_context *c, uint64_t *incremental_ticks, uint64_t *rebuilt_ticks)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    if (incremental_ticks) *incremental_ticks = c->air.incremental;
    if (rebuilt_ticks) *rebuilt_ticks = c->air.rebuilds;
    return RM_OK;
}

int rm_air_scan_ticks(const rm_context *c, uint64_t *scan_ticks)
{
    if (!c || !scan_ticks) return fail(RM_ERR_INVALID, "NULL argument");
    *scan_ticks = c->air.scans;
    return RM_OK;
}

int rm_air_batch_stats(const rm_context *c, uint64_t *batches, uint64_t *ticks)
{
    if (!c) return fail(RM_ERR_INVALID, "NULL argument");
    if (batches) *batches = c->ov.batches;
    if (ticks) *ticks = c->ov.ticks_done;
    return RM_OK;
}

int rm_air_batch_pairs(rm_context *c, uint64_t *pairs, uint64_t *frames, uint64_t *interferers)
{
    if (!c || !pairs) return fail(RM_ERR_INVALID, "NULL argument");
    *pairs = 0;
    if (frames) *frames = 0;
    if (interferers) *interferers = 0;
    if (!c->ov.pair_tail.p || c->ov.batches == 0) return RM_OK;
    RM_HIP(hipSetDevice(c->device));
    std::vector<uint32_t> tails(size_t(rm::kShards) * rm::kShardStride);
    RM_HIP(hipMemcpyAsync(tails.data(), c->ov.pair_tail.p, tails.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    for (int k = 0; k < rm::kShards; ++k) *pairs += std::min<uint64_t>(tails[size_t(k) * rm::kShardStride], c->ov.pair_cap / rm::kShards);
    if (frames) *frames = c->ov.last_frames;
    if (interferers) {
        uint32_t v = 0;
        RM_HIP(hipMemcpyAsync(&v, c->ov.misc.p + 2, sizeof(v), hipMemcpyDeviceToHost, c->stream));
        RM_HIP(hipStreamSynchronize(c->stream));
        *interferers = v;
    }
    return RM_OK;
}

int rm_air_ring_stats(rm_context *c, uint64_t *max_allocated, uint64_t *sub_ring_entries)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    uint64_t mx = 0;
    if (c->air.tail.p && c->air.valid) {
        RM_HIP(hipSetDevice(c->device));
        std::vector<uint32_t> tails(size_t(rm::kShards) * rm::kShardStride);
        RM_HIP(hipMemcpyAsync(tails.data(), c->air.tail.p, tails.size() * 4, hipMemcpyDeviceToHost, c->stream));
        RM_HIP(hipStreamSynchronize(c->stream));
        for (int k = 0; k < rm::kShards; ++k) mx = std::max<uint64_t>(mx, tails[size_t(k) * rm::kShardStride]);
    }
    if (max_allocated) *max_allocated = mx;
    if (sub_ring_entries) *sub_ring_entries = c->air.sub_cap;
    return RM_OK;
}

int rm_set_time(rm_context *c, int64_t t)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    c->current_time = t;
    return RM_OK;
}

int64_t rm_air_time_us(int64_t hex_length) { return hex_length * 32; } // RadioPacket.java:72

void rm_event_times(int64_t start_us, int64_t air_us, int64_t current_time_us, int64_t *t_start, int64_t *t_end)
{
    int64_t packetTime = start_us; // Simulator.java:323-326
    if (packetTime < current_time_us) packetTime = current_time_us;
    if (t_start) *t_start = packetTime;
    if (t_end) *t_end = packetTime + air_us;
}

double rm_det_math(int32_t fn, double x) { return rm::host_det_math(fn, x); }

uint64_t rm_link_hash(uint64_t seed, uint32_t a, uint32_t b, double *u) { return rm::host_link_hash(seed, a, b, u); }

void rm_evq_init(rm_evq_order *o)
{
    if (!o) return;
    o->top_start = 0; // EventQueue.java:51
    o->top_max = 0;
    o->ladders = 0;
    o->top_nonempty = 0;
}

int32_t rm_evq_add(rm_evq_order *o, int64_t time_us)
{
    rm::EvOrder e{o->top_start, o->top_max, o->ladders, o->top_nonempty};
    const int32_t lad = rm::ev_ladder(e, time_us);
    rm::ev_note_top(e, time_us);
    o->top_max = e.top_max;
    o->top_nonempty = e.top_nonempty;
    return lad;
}

void rm_evq_drain(rm_evq_order *o, int64_t time_us)
{
    rm::EvOrder e{o->top_start, o->top_max, o->ladders, o->top_nonempty};
    rm::ev_drain(e, time_us);
    o->top_start = e.top_start;
    o->top_max = e.top_max;
    o->ladders = e.ladders;
    o->top_nonempty = e.top_nonempty;
}

int rm_sync(rm_context *c)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    RM_HIP(hipSetDevice(c->device));
    RM_HIP(hipStreamSynchronize(c->stream));
    return RM_OK;
}

uint64_t rm_lcg_jump(uint64_t state48, uint64_t steps)
{
    uint64_t A, C;
    rm::host_lcg_jump_map(steps, &A, &C);
    return (A * state48 + C) & ((1ull << 48) - 1);
}

double rm_lcg_next_double(uint64_t *state48)
{
    const uint64_t a = 0x5DEECE66Dull, cc = 0xBull, mask = (1ull << 48) - 1;
    uint64_t s = *state48;
    s = (s * a + cc) & mask;
    const int64_t hi = int64_t(s >> 22);
    s = (s * a + cc) & mask;
    const int64_t lo = int64_t(s >> 21);
    *state48 = s;
    return double((hi << 27) + lo) * 0x1.0p-53;
}

} // extern "C"
