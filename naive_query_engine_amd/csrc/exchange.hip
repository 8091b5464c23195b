// exchange.hip — the exchange steps of the sharded operators behind the C ABI (no reference analogue: the reference
// is one process, one thread).  One process per GPU, every rank holds a contiguous row range:
//   * filter / projection / join probe: rows are independent; rank order == row order.  The only exchange is the optional
//     materialisation of the whole result on every rank: an ORDERED variable-length all-gather — the row counts travel first
//     (one 8-byte all-gather + the one host wait of the operation), then every rank sends each column straight out of its
//     local table to every peer and receives each peer's rows straight into their place in the output column (direct peer
//     sends over the xGMI mesh inside one RCCL group; no staging copies, no ring);
//   * hash aggregate: every rank's partial {key, count, sum, min, max} table travels in ONE fixed-size all-gather whose last
//     word is the group count, and the merge reads those counts on the device — pack, collective and merge are enqueued on the
//     context's stream back to back, the host waits once (for the merged group count).
// The transport is a small vtable: the default implementation binds RCCL at run time (dlopen of librccl.so.1, so that a
// single-GPU user never loads the 570 MB library and a process that already carries PyTorch's RCCL shares it), and a host may
// plug in its own (the tests drive the very same sharding code with two ranks on one GPU through a host-staged transport).
#include <dlfcn.h>

#include <rccl/rccl.h>

#include "device_utils.hpp"
#include "nqe_internal.hpp"

struct nqe_comm {
    nqe_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    nqe_transport tr{};
    bool owns_user = false;
};

namespace nqe {
namespace {

// ---------------------------------------------------------------- RCCL binding (run-time)
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};

Rccl &rccl() {
    static Rccl r;
    if (r.handle) return r;
    // the soname first: a librccl.so.1 that is already mapped (PyTorch's) is returned as is
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) fail(NQE_ERR_RCCL, std::string("cannot load RCCL (librccl.so.1): ") + dlerror());
    auto sym = [&](const char *name) {
        void *p = dlsym(h, name);
        if (!p) fail(NQE_ERR_RCCL, std::string("RCCL symbol missing: ") + name);
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
    r.handle = h;
    return r;
}

struct RcclState {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

#define NQE_RCCL_RET(expr)                                                                                   \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess) return int32_t(r_);                                                           \
    } while (0)

int32_t rccl_all_gather(void *user, const void *send, void *recv, size_t bytes, void *stream) {
    auto *s = static_cast<RcclState *>(user);
    NQE_RCCL_RET(rccl().AllGather(send, recv, bytes, ncclInt8, s->comm, static_cast<hipStream_t>(stream)));
    return 0;
}
// every rank sends its `send_bytes` to every peer and receives peer r's recv_bytes[r] at recv + recv_offsets[r]: direct
// point-to-point transfers over the full xGMI mesh in one group; the rank's own part is a device copy.  Empty parts are
// skipped on both sides (sender and receiver know all the counts).
int32_t rccl_all_gather_v(void *user, const void *send, size_t send_bytes, void *recv, const size_t *recv_offsets, const size_t *recv_bytes,
                          void *stream) {
    auto *s = static_cast<RcclState *>(user);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (send_bytes && hipMemcpyAsync(static_cast<char *>(recv) + recv_offsets[s->rank], send, send_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return -1;
    NQE_RCCL_RET(rccl().GroupStart());
    for (int r = 0; r < s->world; ++r) {
        if (r == s->rank) continue;
        if (send_bytes) NQE_RCCL_RET(rccl().Send(send, send_bytes, ncclInt8, r, s->comm, st));
        if (recv_bytes[r]) NQE_RCCL_RET(rccl().Recv(static_cast<char *>(recv) + recv_offsets[r], recv_bytes[r], ncclInt8, r, s->comm, st));
    }
    NQE_RCCL_RET(rccl().GroupEnd());
    return 0;
}
int32_t rccl_group_begin(void *) { return int32_t(rccl().GroupStart()); }
int32_t rccl_group_end(void *) { return int32_t(rccl().GroupEnd()); }
void rccl_destroy(void *user) {
    auto *s = static_cast<RcclState *>(user);
    if (s->comm) (void)rccl().CommDestroy(s->comm);
    delete s;
}

void tr_check(nqe_comm *c, int32_t rc, const char *what) {
    if (rc == 0) return;
    std::string msg = std::string("exchange: ") + what + " failed";
    if (c->owns_user) msg += std::string(": ") + rccl().GetErrorString(ncclResult_t(rc));
    else msg += " (transport status " + std::to_string(rc) + ")";
    fail(NQE_ERR_RCCL, msg);
}

__global__ void store_word_kernel(uint64_t *dst, uint64_t v) { *dst = v; }

// the columns an exchange can move: 8-byte words without validity (what partial aggregate states, join outputs over plain
// columns and projected integer/float expressions are)
void require_plain(const nqe_table *t, const char *who) {
    for (auto &c : t->cols)
        if (!is_word_type(c.dtype) || c.validity)
            fail(NQE_ERR_NOT_SUPPORTED, std::string(who) + ": only 8-byte columns without validity can be exchanged");
}

std::unique_ptr<nqe_table> all_gather_table(nqe_comm *cm, const nqe_table *local) {
    nqe_ctx *ctx = cm->ctx;
    require_plain(local, "nqe_table_all_gather");
    const int world = cm->world;
    // ---- row counts first (the one host wait)
    BufRef cnt = dev_alloc(ctx, 8), cnts = dev_alloc(ctx, size_t(world) * 8);
    launch(ctx, "exchange_store_word", store_word_kernel, dim3(1), dim3(1), 0, (uint64_t *)cnt->ptr, uint64_t(local->rows));
    tr_check(cm, cm->tr.all_gather(cm->tr.user, cnt->ptr, cnts->ptr, 8, ctx->stream), "all_gather(row counts)");
    std::vector<uint64_t> counts(size_t(world), 0);
    NQE_HIP_CHECK(hipMemcpyAsync(counts.data(), cnts->ptr, size_t(world) * 8, hipMemcpyDeviceToHost, ctx->stream));
    sync(ctx);
    std::vector<size_t> off(size_t(world) + 1, 0), bytes(size_t(world), 0), boff(size_t(world), 0);
    for (int r = 0; r < world; ++r) {
        off[size_t(r) + 1] = off[size_t(r)] + size_t(counts[size_t(r)]);
        bytes[size_t(r)] = size_t(counts[size_t(r)]) * 8;
        boff[size_t(r)] = off[size_t(r)] * 8;
    }
    if (counts[size_t(cm->rank)] != uint64_t(local->rows)) fail(NQE_ERR_RCCL, "exchange: gathered row count of this rank differs from its table");
    const size_t total = off[size_t(world)];
    auto out = std::make_unique<nqe_table>();
    out->ctx = ctx;
    out->rows = int64_t(total);
    // columns that share one buffer locally (the two key columns of an equi-join's output) travel once and share it again
    std::vector<const void *> seen;
    std::vector<size_t> seen_at;
    size_t distinct = 0;
    for (auto &c : local->cols) {
        const void *p = c.values ? c.values->ptr : nullptr;
        bool dup = false;
        for (const void *q : seen) dup = dup || (p && q == p);
        if (!dup) ++distinct;
        seen.push_back(p);
    }
    BufRef all = dev_alloc(ctx, total * distinct * 8 + 8);
    seen.clear();
    size_t next = 0;
    tr_check(cm, cm->tr.group_begin(cm->tr.user), "group_begin");
    for (size_t ci = 0; ci < local->cols.size(); ++ci) {
        const DevColumn &c = local->cols[ci];
        const void *p = c.values ? c.values->ptr : nullptr;
        DevColumn d;
        d.dtype = c.dtype;
        d.length = int64_t(total);
        d.null_count = 0;
        size_t hit = size_t(-1);
        for (size_t k = 0; k < seen.size(); ++k)
            if (p && seen[k] == p) hit = k;
        if (hit != size_t(-1)) {
            d.values = out->cols[seen_at[hit]].values;
        } else {
            d.values = dev_view(all, next * total * 8, total * 8);
            ++next;
            if (total)
                tr_check(cm, cm->tr.all_gather_v(cm->tr.user, p, size_t(local->rows) * 8, d.values->ptr, boff.data(), bytes.data(), ctx->stream),
                         "all_gather_v(column)");
            seen.push_back(p);
            seen_at.push_back(ci);
        }
        out->cols.push_back(std::move(d));
    }
    tr_check(cm, cm->tr.group_end(cm->tr.user), "group_end");
    sync(ctx); // the local table may be released by the caller as soon as this returns
    return out;
}

void check_status(nqe_ctx *ctx, nqe_status st) {
    if (st != NQE_OK) fail(st, ctx->last_error);
}

} // namespace
} // namespace nqe

using namespace nqe;

extern "C" {

nqe_status nqe_comm_get_unique_id(void *id_out) {
    NQE_API_BEGIN(nullptr)
    if (!id_out) fail(NQE_ERR_INVALID_ARGUMENT, "id_out is NULL");
    static_assert(NQE_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    ncclResult_t r = rccl().GetUniqueId(&id);
    if (r != ncclSuccess) fail(NQE_ERR_RCCL, std::string("ncclGetUniqueId: ") + rccl().GetErrorString(r));
    std::memcpy(id_out, &id, NQE_COMM_ID_BYTES);
    NQE_API_END()
}

nqe_status nqe_comm_rccl_version(int32_t *version_out) {
    NQE_API_BEGIN(nullptr)
    if (!version_out) fail(NQE_ERR_INVALID_ARGUMENT, "version_out is NULL");
    int v = 0;
    ncclResult_t r = rccl().GetVersion(&v);
    if (r != ncclSuccess) fail(NQE_ERR_RCCL, std::string("ncclGetVersion: ") + rccl().GetErrorString(r));
    *version_out = v;
    NQE_API_END()
}

nqe_status nqe_comm_create(nqe_ctx *ctx, const void *unique_id, int32_t rank, int32_t world, nqe_comm **out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !unique_id || !out || world < 1 || rank < 0 || rank >= world) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    NQE_HIP_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, NQE_COMM_ID_BYTES);
    auto st = std::make_unique<RcclState>();
    st->rank = rank;
    st->world = world;
    ncclResult_t r = rccl().CommInitRank(&st->comm, world, id, rank);
    if (r != ncclSuccess) fail(NQE_ERR_RCCL, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
    auto cm = std::make_unique<nqe_comm>();
    cm->ctx = ctx;
    cm->rank = rank;
    cm->world = world;
    cm->tr.user = st.release();
    cm->tr.all_gather = rccl_all_gather;
    cm->tr.all_gather_v = rccl_all_gather_v;
    cm->tr.group_begin = rccl_group_begin;
    cm->tr.group_end = rccl_group_end;
    cm->tr.destroy = rccl_destroy;
    cm->owns_user = true;
    *out = cm.release();
    NQE_API_END()
}

nqe_status nqe_comm_create_custom(nqe_ctx *ctx, const nqe_transport *transport, int32_t rank, int32_t world, nqe_comm **out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !transport || !out || world < 1 || rank < 0 || rank >= world || !transport->all_gather || !transport->all_gather_v)
        fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    auto cm = std::make_unique<nqe_comm>();
    cm->ctx = ctx;
    cm->rank = rank;
    cm->world = world;
    cm->tr = *transport;
    static auto nop = +[](void *) -> int32_t { return 0; };
    if (!cm->tr.group_begin) cm->tr.group_begin = nop;
    if (!cm->tr.group_end) cm->tr.group_end = nop;
    *out = cm.release();
    NQE_API_END()
}

nqe_status nqe_comm_destroy(nqe_comm *comm) {
    if (!comm) return NQE_OK;
    if (comm->ctx) {
        (void)hipSetDevice(comm->ctx->device);
        (void)hipStreamSynchronize(comm->ctx->stream);
    }
    if (comm->tr.destroy) comm->tr.destroy(comm->tr.user);
    delete comm;
    return NQE_OK;
}

int32_t nqe_comm_rank(const nqe_comm *comm) { return comm ? comm->rank : 0; }
int32_t nqe_comm_world(const nqe_comm *comm) { return comm ? comm->world : 1; }

nqe_status nqe_table_all_gather(nqe_comm *comm, const nqe_table *local, nqe_table **out) {
    NQE_API_BEGIN(comm ? comm->ctx : nullptr)
    if (!comm || !local || !out) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    *out = all_gather_table(comm, local).release();
    NQE_API_END()
}

nqe_status nqe_sharded_aggregate_execute(nqe_comm *comm, const nqe_table *in, const nqe_expr_node *pred, int32_t pred_nodes,
                                         const nqe_expr_node *group, int32_t group_nodes, const nqe_aggregate *aggs, int32_t num_aggs,
                                         nqe_table **out, nqe_table **keys_out) {
    NQE_API_BEGIN(comm ? comm->ctx : nullptr)
    if (!comm || !in || !out || (num_aggs > 0 && !aggs)) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    nqe_ctx *ctx = comm->ctx;
    const bool grouped = group && group_nodes > 0;
    nqe_table *state_raw = nullptr, *keys_raw = nullptr;
    check_status(ctx, nqe_aggregate_partial(ctx, in, pred, pred_nodes, group, group_nodes, aggs, num_aggs, &state_raw, &keys_raw));
    std::unique_ptr<nqe_table> state(state_raw), keys(keys_raw);
    if (grouped && keys && !keys->cols.empty() && keys->cols[0].dtype == NQE_UTF8)
        // the partial's Utf8 keys are strings of rank-local representative rows; exchanging them needs a byte exchange and a global
        // re-encoding that this path does not have
        fail(NQE_ERR_NOT_SUPPORTED, "sharded aggregate: Utf8 group keys are not exchanged (aggregate each shard's strings on one rank)");
    const int nk = grouped ? 1 : 0;
    const int ncols = nk + int(state->cols.size());
    const int key_dtype = grouped ? keys->cols[0].dtype : NQE_INT64;
    const int64_t stride = NQE_EXCHANGE_ROWS;
    const size_t words = size_t(ncols) * size_t(stride) + 1;
    BufRef buf = dev_alloc(ctx, words * 8), gathered = dev_alloc(ctx, words * 8 * size_t(comm->world));
    const int64_t rows = state->rows;
    if (rows <= stride) {
        const nqe_table *tabs[2];
        int nt = 0;
        if (grouped) tabs[nt++] = keys.get();
        tabs[nt++] = state.get();
        check_status(ctx, nqe_table_pack_words(ctx, tabs, nt, stride, buf->ptr));
    } else {
        // header only: tells the peers that this exchange takes the exact-size path
        launch(ctx, "exchange_store_word", store_word_kernel, dim3(1), dim3(1), 0, (uint64_t *)buf->ptr + (words - 1), uint64_t(rows));
    }
    tr_check(comm, comm->tr.all_gather(comm->tr.user, buf->ptr, gathered->ptr, words * 8, ctx->stream), "all_gather(partial aggregate states)");
    nqe_table *merged = nullptr, *merged_keys = nullptr;
    check_status(ctx, nqe_aggregate_merge_packed(ctx, gathered->ptr, comm->world, stride, grouped ? 1 : 0, key_dtype, aggs, num_aggs, &merged,
                                                 grouped ? &merged_keys : nullptr));
    if (!merged) {
        // ---- some rank holds more groups than the fixed buffer: exact-size exchange of the (keys, state) tables
        nqe_table both;
        both.ctx = ctx;
        both.rows = rows;
        if (grouped) both.cols.push_back(keys->cols[0]);
        for (auto &c : state->cols) both.cols.push_back(c);
        std::unique_ptr<nqe_table> all = all_gather_table(comm, &both);
        nqe_table kt, stt;
        kt.ctx = stt.ctx = ctx;
        kt.rows = stt.rows = all->rows;
        if (grouped) kt.cols.push_back(all->cols[0]);
        for (size_t c = size_t(nk); c < all->cols.size(); ++c) stt.cols.push_back(all->cols[c]);
        const nqe_table *sp = &stt, *kp = &kt;
        check_status(ctx, nqe_aggregate_merge(ctx, &sp, grouped ? &kp : nullptr, 1, aggs, num_aggs, &merged, &merged_keys));
        sync(ctx);
    }
    *out = merged;
    if (keys_out) *keys_out = merged_keys;
    else if (merged_keys) nqe_table_release(merged_keys);
    NQE_API_END()
}

nqe_status nqe_sharded_hash_join_probe(nqe_comm *comm, const nqe_join_table *build, const nqe_table *right_local, int32_t right_key,
                                       int32_t gather, nqe_table **out) {
    NQE_API_BEGIN(comm ? comm->ctx : nullptr)
    if (!comm || !build || !right_local || !out) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    nqe_ctx *ctx = comm->ctx;
    nqe_table *local = nullptr;
    check_status(ctx, nqe_hash_join_probe(ctx, build, right_local, right_key, &local));
    std::unique_ptr<nqe_table> guard(local);
    if (!gather) {
        *out = guard.release();
    } else {
        *out = all_gather_table(comm, local).release();
    }
    NQE_API_END()
}

nqe_status nqe_sharded_selection_projection_execute(nqe_comm *comm, const nqe_table *in_local, const nqe_expr_node *pred, int32_t pred_nodes,
                                                    const nqe_expr_node *nodes, const int32_t *expr_offsets, int32_t num_exprs, int32_t gather,
                                                    nqe_table **out) {
    NQE_API_BEGIN(comm ? comm->ctx : nullptr)
    if (!comm || !in_local || !out) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    nqe_ctx *ctx = comm->ctx;
    nqe_table *local = nullptr;
    check_status(ctx, nqe_selection_projection_execute(ctx, in_local, pred, pred_nodes, nodes, expr_offsets, num_exprs, &local));
    std::unique_ptr<nqe_table> guard(local);
    if (!gather) {
        *out = guard.release();
    } else {
        *out = all_gather_table(comm, local).release();
    }
    NQE_API_END()
}

} // extern "C"
