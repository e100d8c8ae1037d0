// rm_api_comm.cpp -- C ABI: the all-gather of active-transmitter records INSIDE the library (rm_comm_*, rm_dist_*):
// RCCL over xGMI, called from C -- no framework between a host (the JNI shim, a C++ host, a rank of bench.py) and the
// collective.  A receiver-sharded tick is then ONE call: pack this rank's transmitters, ncclAllGather, sweep.
//
// RCCL is bound at run time (dlopen): the library loads and every other entry point works on a box without RCCL, and a
// process that already holds an RCCL (a torch process: its bundled librccl.so) shares that instance instead of loading a
// second one.  RM_RCCL_LIB names another library file.
#include "rm_host.hpp"

#include <dlfcn.h>

using namespace rmh;

namespace {

// the few RCCL entry points used, with the signatures of <rccl/rccl.h> (ROCm 7.2: NCCL 2.x API)
struct UniqueId {
    char internal[RM_COMM_ID_BYTES];
};
static_assert(RM_COMM_ID_BYTES == 128, "NCCL_UNIQUE_ID_BYTES");
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why; // why it could not be had
};

Rccl *rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return &r;
    tried = true;
    const char *env = std::getenv("RM_RCCL_LIB");
    const char *names[] = {"librccl.so", "librccl.so.1"};
    if (env) r.lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    for (const char *n : names) // an RCCL this process already holds (a torch process: its bundled one)
        if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) {
        r.why = std::string("RCCL could not be loaded (") + (dlerror() ? dlerror() : "librccl.so.1 not found") + ")";
        return &r;
    }
    bool ok = true;
    auto sym = [&](const char *name) -> void * {
        void *p = dlsym(r.lib, name);
        if (!p) {
            ok = false;
            r.why = std::string("RCCL lacks ") + name;
        }
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) r.lib = nullptr;
    return &r;
}

int need_rccl(Rccl *&out)
{
    out = rccl();
    if (!out->lib) return fail(RM_ERR_NO_DEVICE, out->why);
    return RM_OK;
}

#define RM_NCCL(r, call)                                                                               \
    do {                                                                                               \
        const int e_ = (call);                                                                         \
        if (e_ != 0) return fail(RM_ERR_HIP, std::string(#call) + ": " + (r)->GetErrorString(e_));      \
    } while (0)

constexpr int kNcclChar = 0; // ncclInt8 / ncclChar

} // namespace

namespace rmh {

// the all-gather of `bytes` per rank on the context's stream (a one-rank context without a communicator: a copy)
int comm_all_gather(rm_context *c, const void *mine, void *all, size_t bytes)
{
    if (!c->comm) {
        if (c->comm_world != 1) return fail(RM_ERR_STATE, "no communicator: rm_comm_init_rank first");
        if (all != mine) RM_HIP(hipMemcpyAsync(all, mine, bytes, hipMemcpyDeviceToDevice, c->stream));
        return RM_OK;
    }
    Rccl *r = nullptr;
    RM_TRY(need_rccl(r));
    RM_NCCL(r, r->AllGather(mine, all, bytes, kNcclChar, c->comm, c->stream));
    return RM_OK;
}

// java.util.Random draws of a receiver-sharded tick: the per-packet counts go round (and, for regions, the drawing
// links' nodes), every rank places its draws among the others'
int comm_finish_draws(rm_context *c)
{
    if (!c->draws_pending) return RM_OK;
    const int world = c->comm_world, rank = c->comm_rank, n_new = c->last_n_new;
    RM_HIP(c->d_all_cnt.ensure(size_t(world) * std::max(n_new, 1)));
    RM_TRY(comm_all_gather(c, c->d_pkt_draw_cnt.p, c->d_all_cnt.p, size_t(n_new) * 4));
    if (!part_spatial(c)) return rm_tick_finish_draws(c, c->d_all_cnt.p, world, rank, 1);
    // regions: rows as long as the longest rank's list (the one host read-back of this path)
    std::vector<uint32_t> cnt(size_t(world) * size_t(std::max(n_new, 1)));
    RM_HIP(hipMemcpyAsync(cnt.data(), c->d_all_cnt.p, size_t(world) * n_new * 4, hipMemcpyDeviceToHost, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    uint64_t stride = 1;
    for (int r = 0; r < world; ++r) {
        uint64_t tot = 0;
        for (int q = 0; q < n_new; ++q) tot += cnt[size_t(r) * n_new + q];
        stride = std::max(stride, tot);
    }
    RM_HIP(c->d_all_nodes.ensure(size_t(world) * stride));
    RM_HIP(c->d_draw_nodes.ensure(stride)); // (its tail beyond this rank's own list is padding nobody reads)
    RM_TRY(comm_all_gather(c, c->d_draw_nodes.p, c->d_all_nodes.p, stride * 4));
    return rm_tick_finish_draws_nodes(c, c->d_all_cnt.p, c->d_all_nodes.p, uint32_t(stride), world, 1);
}

} // namespace rmh

extern "C" {

int rm_comm_available(void)
{
    Rccl *r = rccl();
    if (!r->lib) g_err = r->why;
    return r->lib ? 1 : 0;
}

int rm_comm_get_unique_id(uint8_t *id)
{
    if (!id) return fail(RM_ERR_INVALID, "id is NULL");
    Rccl *r = nullptr;
    RM_TRY(need_rccl(r));
    UniqueId u;
    RM_NCCL(r, r->GetUniqueId(&u));
    std::memcpy(id, u.internal, RM_COMM_ID_BYTES);
    return RM_OK;
}

int rm_comm_destroy(rm_context *c)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    if (c->comm && c->comm_owned) {
        Rccl *r = rccl();
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (r->lib) (void)r->CommDestroy(c->comm);
    }
    c->comm = nullptr;
    c->comm_owned = false;
    c->comm_world = 1;
    c->comm_rank = 0;
    return RM_OK;
}

int rm_comm_init_rank(rm_context *c, const uint8_t *id, int32_t world, int32_t rank)
{
    if (!c || !id || world < 1 || rank < 0 || rank >= world) return fail(RM_ERR_INVALID, "bad arguments");
    Rccl *r = nullptr;
    RM_TRY(need_rccl(r));
    RM_TRY(rm_comm_destroy(c));
    RM_HIP(hipSetDevice(c->device));
    UniqueId u;
    std::memcpy(u.internal, id, RM_COMM_ID_BYTES);
    void *comm = nullptr;
    RM_NCCL(r, r->CommInitRank(&comm, world, u, rank));
    c->comm = comm;
    c->comm_owned = true;
    c->comm_world = world;
    c->comm_rank = rank;
    return RM_OK;
}

int rm_comm_world(const rm_context *c) { return c ? c->comm_world : fail(RM_ERR_INVALID, "ctx is NULL"); }
int rm_comm_rank(const rm_context *c) { return c ? c->comm_rank : fail(RM_ERR_INVALID, "ctx is NULL"); }

// The ticks' transmitters as every rank's own source indices, already gathered: dev_src_all[rank][tick][slot].  Every rank
// builds ALL the records itself from its copy of the node table (the same bytes a rank's own pack would have sent), in
// the layout rm_batch_run_gathered_device reads: what crosses the links between the GPUs is 4 bytes per frame, not 64.
int rm_batch_run_gathered_sources_device(rm_context *c, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us,
                                         const int32_t *dev_src_all, int32_t world, int32_t slots, const int64_t *start_us, int64_t air_us)
{
    if (!c || n_ticks < 1 || n_ticks > RM_MAX_BATCH || slots < 1 || world < 1 || !dev_src_all || !start_us || !t_begin_us || !t_end_us || air_us < 0)
        return fail(RM_ERR_INVALID, "bad arguments");
    RM_HIP(hipSetDevice(c->device));
    // (the sweep's pre-pass builds every frame's record from its source index where it needs it: no packed copy in between;
    // the frames' time spans travel with the call -- the SINR medium's ticks may outlive each other)
    static thread_local std::vector<int64_t> air_v;
    air_v.assign(size_t(n_ticks), air_us);
    return batch_run(c, n_ticks, t_begin_us, t_end_us, nullptr, nullptr, nullptr, start_us, air_v.data(), nullptr, world, slots, dev_src_all);
}

// The same with every rank's block as the library's own all-gather leaves it: n_ticks * slots source indices, then
// RM_GATHER_TRAILER words -- the rank's node-table digest (rm_table_digest) in the first two.  Every rank builds the other
// ranks' records from ITS copy of the node table; a rank whose copy differs from this context's (it missed an
// rm_node_update: the reference has no change hook, net/SimulatorJSONHandler.java:105-143) would yield silently wrong
// verdicts -- here every tick of the batch reads as RM_ERR_STATE instead, on every rank.
int rm_batch_run_gathered_blocks_device(rm_context *c, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us,
                                        const int32_t *dev_blocks, int32_t world, int32_t slots, const int64_t *start_us, int64_t air_us)
{
    if (!c || n_ticks < 1 || n_ticks > RM_MAX_BATCH || slots < 1 || world < 1 || !dev_blocks || !start_us || !t_begin_us || !t_end_us || air_us < 0)
        return fail(RM_ERR_INVALID, "bad arguments");
    RM_HIP(hipSetDevice(c->device));
    static thread_local std::vector<int64_t> air_v;
    air_v.assign(size_t(n_ticks), air_us);
    const int mine = n_ticks * slots;
    return batch_run(c, n_ticks, t_begin_us, t_end_us, nullptr, nullptr, nullptr, start_us, air_v.data(), nullptr, world, slots, dev_blocks,
                     mine + rm::kGatherTrailer, mine);
}

int rm_dist_batch_run_sources_device(rm_context *c, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us,
                                     const int32_t *dev_src, int32_t slots, const int64_t *start_us, int64_t air_us)
{
    if (!c || n_ticks < 1 || n_ticks > RM_MAX_BATCH || slots < 1 || !dev_src || !start_us || !t_begin_us || !t_end_us || air_us < 0)
        return fail(RM_ERR_INVALID, "bad arguments");
    RM_HIP(hipSetDevice(c->device));
    // the all-gather carries the source INDICES (4 bytes per frame): every rank has the whole node table and builds the
    // records of all ranks' frames itself -- so the tables have to agree: each rank's block ends with its table's digest
    const size_t mine = size_t(n_ticks) * size_t(slots);
    if (!c->comm) return rm_batch_run_gathered_sources_device(c, n_ticks, t_begin_us, t_end_us, dev_src, 1, slots, start_us, air_us);
    const size_t block = mine + size_t(rm::kGatherTrailer);
    RM_HIP(c->d_dist_stage.ensure(block));
    RM_HIP(c->d_dist_idx.ensure(block * size_t(c->comm_world)));
    RM_HIP(rm::launch_stage_block(c->stream, dev_src, int(mine), c->table_digest, c->d_dist_stage.p));
    RM_TRY(comm_all_gather(c, c->d_dist_stage.p, c->d_dist_idx.p, block * sizeof(int32_t)));
    return rm_batch_run_gathered_blocks_device(c, n_ticks, t_begin_us, t_end_us, c->d_dist_idx.p, c->comm_world, slots, start_us, air_us);
}

int rm_dist_tick_run_sources_device(rm_context *c, int64_t t_begin_us, int64_t t_end_us, const int32_t *dev_src, int32_t slots,
                                    int64_t start_us, int64_t air_us)
{
    if (!c || slots < 1 || !dev_src || air_us < 0) return fail(RM_ERR_INVALID, "bad arguments");
    RM_HIP(hipSetDevice(c->device));
    const int world = c->comm ? c->comm_world : 1;
    const int32_t *all = dev_src;
    if (c->comm) { // (indices, as above)
        RM_HIP(c->d_dist_idx.ensure(size_t(slots) * size_t(world)));
        RM_TRY(comm_all_gather(c, dev_src, c->d_dist_idx.p, size_t(slots) * sizeof(int32_t)));
        all = c->d_dist_idx.p;
    }
    RM_HIP(c->d_dist_all.ensure(size_t(slots) * size_t(world)));
    RM_HIP(rm::launch_pack_tx(c->stream, nodes_dev(c), all, slots * world, start_us, air_us, c->d_dist_all.p));
    // (the SINR medium keeps the frames on the air: the packed frames all end at start + air)
    RM_TRY(rm_tick_run_records_device(c, t_begin_us, t_end_us, c->d_dist_all.p, slots * world, start_us + air_us));
    return comm_finish_draws(c);
}

} // extern "C"

// ---- the group's device-resident tick: one host thread, one communicator over all members (ncclCommInitAll) ----------------

namespace rmh {

int group_comm_init(rm_context *const *members, int n, void **comms_out)
{
    Rccl *r = nullptr;
    RM_TRY(need_rccl(r));
    std::vector<int> devs(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) devs[size_t(i)] = members[i]->device;
    std::vector<void *> comms(static_cast<size_t>(n), nullptr);
    RM_NCCL(r, r->CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) {
        rm_context *c = members[i];
        c->comm = comms[size_t(i)];
        c->comm_owned = true;
        c->comm_world = n;
        c->comm_rank = i;
        if (comms_out) comms_out[i] = comms[size_t(i)];
    }
    return RM_OK;
}

// every member's all-gather enqueued as ONE group call (a single host thread drives all ranks of the communicator)
int group_all_gather(rm_context *const *members, int n, const void *const *mine, void *const *all, size_t bytes)
{
    Rccl *r = nullptr;
    RM_TRY(need_rccl(r));
    RM_NCCL(r, r->GroupStart());
    for (int i = 0; i < n; ++i) {
        rm_context *c = members[i];
        const int e = r->AllGather(mine[i], all[i], bytes, kNcclChar, c->comm, c->stream);
        if (e != 0) {
            (void)r->GroupEnd();
            return fail(RM_ERR_HIP, std::string("ncclAllGather: ") + r->GetErrorString(e));
        }
    }
    RM_NCCL(r, r->GroupEnd());
    return RM_OK;
}

} // namespace rmh
