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

#include "device_utils.hpp"
#include "nqe_internal.hpp"

// The handful of RCCL declarations this file binds by name at run time (rccl/rccl.h, NCCL's stable C API): a single-GPU build of
// the library needs neither the header nor the 570 MB shared object.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;   // ncclSuccess == 0
typedef int ncclDataType_t; // ncclInt8 == 0
}
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclInt8 = 0;

struct nqe_comm {
    nqe_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    nqe_transport tr{};
    bool owns_user = false; // tr.user is one of this file's P2PState objects
    bool is_rccl = false;
};

namespace nqe {
namespace {

// ---------------------------------------------------------------- RCCL binding (run-time)
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
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

// ---------------------------------------------------------------- collectives over point-to-point primitives
// One state object behind every communicator this file builds itself: the p2p primitives (RCCL's, or the host's through
// nqe_comm_create_p2p) and, for RCCL, the communicator whose native ncclAllGather serves the fixed-size collective.
struct P2PState {
    nqe_p2p p{};
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr; // RCCL only
};

int32_t rccl_send(void *user, const void *buf, size_t bytes, int32_t peer, void *stream) {
    return int32_t(rccl().Send(buf, bytes, ncclInt8, peer, static_cast<P2PState *>(user)->comm, static_cast<hipStream_t>(stream)));
}
int32_t rccl_recv(void *user, void *buf, size_t bytes, int32_t peer, void *stream) {
    return int32_t(rccl().Recv(buf, bytes, ncclInt8, peer, static_cast<P2PState *>(user)->comm, static_cast<hipStream_t>(stream)));
}
int32_t rccl_group_begin(void *) { return int32_t(rccl().GroupStart()); }
int32_t rccl_group_end(void *) { return int32_t(rccl().GroupEnd()); }

// closes a group on every path out of the scope that opened it: a failure between group_begin and group_end must not leave the
// (thread-wide) RCCL group open — that would swallow every later RCCL call of the thread
struct P2PGroup {
    P2PState *s;
    bool open = false;
    explicit P2PGroup(P2PState *st) : s(st) {}
    int32_t begin() {
        int32_t rc = s->p.group_begin ? s->p.group_begin(s->p.user) : 0;
        open = rc == 0;
        return rc;
    }
    int32_t end() {
        open = false;
        return s->p.group_end ? s->p.group_end(s->p.user) : 0;
    }
    ~P2PGroup() {
        if (open) (void)end();
    }
};

// every rank sends its `send_bytes` to every peer and receives peer r's recv_bytes[r] at recv + recv_offsets[r]: direct
// point-to-point transfers over the full xGMI mesh in one group; the rank's own part is a device copy.  Empty parts are
// skipped on both sides (sender and receiver know all the counts): rank a posts send(a→b) iff its own part is non-empty, and
// rank b posts recv(b←a) iff recv_bytes[a] != 0 — the same condition seen from both ends, so every send has its receive.
int32_t p2p_all_gather_v(void *user, const void *send, size_t send_bytes, void *recv, const size_t *recv_offsets, const size_t *recv_bytes,
                         void *stream) {
    auto *s = static_cast<P2PState *>(user);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (recv_bytes[s->rank] != send_bytes) return -2; // the caller's counts disagree with what it sends
    if (send_bytes && hipMemcpyAsync(static_cast<char *>(recv) + recv_offsets[s->rank], send, send_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return -1;
    P2PGroup g(s);
    int32_t rc = g.begin();
    if (rc) return rc;
    for (int r = 0; r < s->world; ++r) {
        if (r == s->rank) continue;
        if (send_bytes && (rc = s->p.send(s->p.user, send, send_bytes, r, stream))) return rc;
        if (recv_bytes[r] && (rc = s->p.recv(s->p.user, static_cast<char *>(recv) + recv_offsets[r], recv_bytes[r], r, stream))) return rc;
    }
    return g.end();
}
int32_t p2p_all_gather(void *user, const void *send, void *recv, size_t bytes, void *stream) {
    auto *s = static_cast<P2PState *>(user);
    std::vector<size_t> off(size_t(s->world)), sz(size_t(s->world), bytes);
    for (int r = 0; r < s->world; ++r) off[size_t(r)] = size_t(r) * bytes;
    return p2p_all_gather_v(user, send, bytes, recv, off.data(), sz.data(), stream);
}
int32_t rccl_all_gather(void *user, const void *send, void *recv, size_t bytes, void *stream) {
    auto *s = static_cast<P2PState *>(user);
    return int32_t(rccl().AllGather(send, recv, bytes, ncclInt8, s->comm, static_cast<hipStream_t>(stream)));
}
int32_t p2p_group_begin(void *user) {
    auto *s = static_cast<P2PState *>(user);
    return s->p.group_begin ? s->p.group_begin(s->p.user) : 0;
}
int32_t p2p_group_end(void *user) {
    auto *s = static_cast<P2PState *>(user);
    return s->p.group_end ? s->p.group_end(s->p.user) : 0;
}
void p2p_destroy(void *user) {
    auto *s = static_cast<P2PState *>(user);
    if (s->comm) (void)rccl().CommDestroy(s->comm);
    else if (s->p.destroy) s->p.destroy(s->p.user);
    delete s;
}

void tr_check(nqe_comm *c, int32_t rc, const char *what) {
    if (rc == 0) return;
    std::string msg = std::string("exchange: ") + what + " failed";
    if (c->is_rccl && rc > 0) msg += std::string(": ") + rccl().GetErrorString(ncclResult_t(rc));
    else msg += " (transport status " + std::to_string(rc) + ")";
    fail(NQE_ERR_RCCL, msg);
}

// the brackets around the transfers of one table, closed on every path (see P2PGroup)
struct TableGroup {
    nqe_comm *c;
    bool open = false;
    explicit TableGroup(nqe_comm *cm) : c(cm) {}
    void begin() {
        tr_check(c, c->tr.group_begin(c->tr.user), "group_begin");
        open = true;
    }
    void end() {
        open = false;
        tr_check(c, c->tr.group_end(c->tr.user), "group_end");
    }
    ~TableGroup() {
        if (open) (void)c->tr.group_end(c->tr.user);
    }
};

__global__ void store_word_kernel(uint64_t *dst, uint64_t v) { *dst = v; }

// ---------------------------------------------------------------- the ordered variable-length all-gather of a table
// what a rank brings to an exchange besides its table: the outcome of the local work that produced it
struct LocalStatus {
    int code = NQE_OK;
    std::string msg;
};

// Header every rank contributes (fixed size, so that a rank whose local work failed — and which therefore has no table — can
// still take part): [0] status, [1] rows, [2] columns, then per column {dtype | has-validity << 8, Utf8 data bytes}.
constexpr int XCH_MAX_COLS = 62;
constexpr int XCH_HDR_WORDS = 128;
static_assert(3 + 2 * XCH_MAX_COLS <= XCH_HDR_WORDS, "header size");

// status word of a partial aggregate sent header-only by a rank whose local work failed (any count above NQE_EXCHANGE_ROWS sends
// the peers to the exact-size exchange, whose header then carries the status)
constexpr uint64_t XCH_FAILED_COUNT = uint64_t(1) << 62;

[[noreturn]] void fail_together(nqe_comm *cm, const std::vector<uint64_t> &hdr, const LocalStatus &ls) {
    if (ls.code != NQE_OK) fail(ls.code, ls.msg); // this rank's own error, as it would have been reported without the exchange
    for (int r = 0; r < cm->world; ++r) {
        const uint64_t st = hdr[size_t(r) * XCH_HDR_WORDS];
        if (st != 0)
            fail(NQE_ERR_RCCL, "exchange: rank " + std::to_string(r) + " failed its local part of the operator with status " + std::to_string(int64_t(st)) +
                                   "; every rank returns an error");
    }
    fail(NQE_ERR_RCCL, "exchange: internal error (fail_together without a failed rank)");
}

std::unique_ptr<nqe_table> all_gather_table(nqe_comm *cm, const nqe_table *local, LocalStatus ls = LocalStatus()) {
    nqe_ctx *ctx = cm->ctx;
    const int world = cm->world;
    const size_t W = size_t(world);
    // ---- this rank's header
    std::vector<uint64_t> mine(XCH_HDR_WORDS, 0);
    std::vector<int32_t> ufirst; // first offset of every local Utf8 column (its bytes are sent from there)
    if (ls.code == NQE_OK && (!local || local->cols.size() > size_t(XCH_MAX_COLS))) {
        ls.code = NQE_ERR_NOT_SUPPORTED;
        ls.msg = "exchange: a table of more than " + std::to_string(XCH_MAX_COLS) + " columns cannot be gathered";
    }
    if (ls.code == NQE_OK) {
        const size_t nc = local->cols.size();
        ufirst.assign(nc, 0);
        std::vector<int32_t> ulast(nc, 0);
        bool any_utf8 = false;
        try {
            for (size_t c = 0; c < nc; ++c) {
                const DevColumn &dc = local->cols[c];
                if (dc.dtype != NQE_UTF8 || local->rows == 0) continue;
                any_utf8 = true;
                NQE_HIP_CHECK(hipMemcpyAsync(&ufirst[c], dc.values->ptr, 4, hipMemcpyDeviceToHost, ctx->stream));
                NQE_HIP_CHECK(hipMemcpyAsync(&ulast[c], (const int32_t *)dc.values->ptr + local->rows, 4, hipMemcpyDeviceToHost, ctx->stream));
            }
            if (any_utf8) sync(ctx);
        } catch (const Error &e) {
            ls.code = e.code;
            ls.msg = e.msg;
        }
        mine[1] = uint64_t(local->rows);
        mine[2] = uint64_t(nc);
        for (size_t c = 0; c < nc; ++c) {
            const DevColumn &dc = local->cols[c];
            mine[3 + 2 * c] = uint64_t(uint32_t(dc.dtype) & 0xffu) | (dc.validity ? 0x100u : 0u);
            mine[4 + 2 * c] = uint64_t(ulast[c] - ufirst[c]);
        }
    }
    mine[0] = uint64_t(int64_t(ls.code));
    // ---- headers first (the one host wait)
    std::vector<uint64_t> hdr(W * XCH_HDR_WORDS, 0);
    {
        BufRef dmine = dev_alloc(ctx, XCH_HDR_WORDS * 8), dall = dev_alloc(ctx, W * XCH_HDR_WORDS * 8);
        NQE_HIP_CHECK(hipMemcpyAsync(dmine->ptr, mine.data(), XCH_HDR_WORDS * 8, hipMemcpyHostToDevice, ctx->stream));
        tr_check(cm, cm->tr.all_gather(cm->tr.user, dmine->ptr, dall->ptr, XCH_HDR_WORDS * 8, ctx->stream), "all_gather(headers)");
        NQE_HIP_CHECK(hipMemcpyAsync(hdr.data(), dall->ptr, W * XCH_HDR_WORDS * 8, hipMemcpyDeviceToHost, ctx->stream));
        sync(ctx);
    }
    auto H = [&](int r, int w) { return hdr[size_t(r) * XCH_HDR_WORDS + size_t(w)]; };
    for (int r = 0; r < world; ++r)
        if (H(r, 0) != 0) fail_together(cm, hdr, ls);
    if (ls.code != NQE_OK) fail(ls.code, ls.msg); // (the all_gather lost this rank's status: a broken transport)
    if (H(cm->rank, 1) != uint64_t(local->rows)) fail(NQE_ERR_RCCL, "exchange: gathered row count of this rank differs from its table");
    const size_t nc = local->cols.size();
    for (int r = 0; r < world; ++r) {
        bool same = H(r, 2) == uint64_t(nc);
        for (size_t c = 0; same && c < nc; ++c) same = (H(r, int(3 + 2 * c)) & 0xff) == (mine[3 + 2 * c] & 0xff);
        if (!same) fail(NQE_ERR_ARROW, "exchange: the ranks' tables have different schemas (rank " + std::to_string(r) + ")");
    }
    std::vector<size_t> rows(W), roff(W + 1, 0);
    for (int r = 0; r < world; ++r) {
        rows[size_t(r)] = size_t(H(r, 1));
        roff[size_t(r) + 1] = roff[size_t(r)] + rows[size_t(r)];
    }
    const size_t total = roff[W];
    auto out = std::make_unique<nqe_table>();
    out->ctx = ctx;
    out->rows = int64_t(total);

    // a bitmap (validity, Boolean values) of every rank is received into its own word-aligned slot and shifted into place after
    // the transfers; `has[r]` = rank r sends one (a column is nullable in the output when it is on any rank)
    struct StagedBits {
        BufRef stage;
        size_t stride = 0;
        uint64_t *dst = nullptr;
        std::vector<char> has;
    };
    std::vector<StagedBits> staged_bits;
    struct StagedOffsets {
        BufRef stage;
        size_t stride = 0;
        int32_t *dst = nullptr;
        std::vector<size_t> base; // byte position of every rank's strings in the output
    };
    std::vector<StagedOffsets> staged_offsets;
    std::vector<size_t> off(W), sz(W);
    size_t max_bits_bytes = 0;
    for (int r = 0; r < world; ++r) max_bits_bytes = std::max(max_bits_bytes, (bitmap_bytes(int64_t(rows[size_t(r)])) + 7) / 8 * 8);

    auto gather_bits = [&](const uint8_t *src, BufRef &dst_buf, const std::vector<char> &has, const char *what) {
        StagedBits sb;
        sb.stride = max_bits_bytes;
        sb.stage = dev_alloc(ctx, W * sb.stride + 8);
        dst_buf = dev_alloc_zero(ctx, bitmap_alloc_bytes(int64_t(total)) + 8);
        sb.dst = (uint64_t *)dst_buf->ptr;
        sb.has = has;
        for (int r = 0; r < world; ++r) {
            off[size_t(r)] = size_t(r) * sb.stride;
            sz[size_t(r)] = has[size_t(r)] ? bitmap_bytes(int64_t(rows[size_t(r)])) : 0;
        }
        tr_check(cm, cm->tr.all_gather_v(cm->tr.user, src, sz[size_t(cm->rank)], sb.stage->ptr, off.data(), sz.data(), ctx->stream), what);
        staged_bits.push_back(std::move(sb));
    };

    // columns that share their buffers locally (the two key columns of an equi-join's output) travel once and share them again
    std::vector<size_t> sent; // indices of the word columns already moved
    TableGroup grp(cm);
    grp.begin();
    for (size_t ci = 0; ci < nc; ++ci) {
        const DevColumn &c = local->cols[ci];
        DevColumn d;
        d.dtype = c.dtype;
        d.length = int64_t(total);
        d.null_count = 0;
        std::vector<char> vhas(W, 0);
        bool any_valid = false;
        for (int r = 0; r < world; ++r) {
            vhas[size_t(r)] = rows[size_t(r)] && (H(r, int(3 + 2 * ci)) & 0x100) ? 1 : 0;
            any_valid = any_valid || vhas[size_t(r)];
        }
        if (is_word_type(c.dtype)) {
            size_t hit = size_t(-1);
            for (size_t k : sent) {
                const DevColumn &o = local->cols[k];
                if (c.values && o.values && o.values->ptr == c.values->ptr && o.dtype == c.dtype && !any_valid && !out->cols[k].validity) hit = k;
            }
            if (hit != size_t(-1)) {
                d.values = out->cols[hit].values;
                out->cols.push_back(std::move(d));
                continue;
            }
            d.values = dev_alloc(ctx, total * 8 + 8);
            for (int r = 0; r < world; ++r) {
                off[size_t(r)] = roff[size_t(r)] * 8;
                sz[size_t(r)] = rows[size_t(r)] * 8;
            }
            if (total)
                tr_check(cm, cm->tr.all_gather_v(cm->tr.user, c.values ? c.values->ptr : nullptr, size_t(local->rows) * 8, d.values->ptr, off.data(), sz.data(),
                                                 ctx->stream),
                         "all_gather_v(column)");
            if (!any_valid) sent.push_back(ci);
        } else if (c.dtype == NQE_BOOLEAN) {
            std::vector<char> all(W);
            for (int r = 0; r < world; ++r) all[size_t(r)] = rows[size_t(r)] ? 1 : 0;
            if (total) gather_bits(c.bits(), d.values, all, "all_gather_v(Boolean values)");
            else d.values = dev_alloc_zero(ctx, 16);
        } else if (c.dtype == NQE_UTF8) {
            StagedOffsets so;
            so.base.assign(W + 1, 0);
            for (int r = 0; r < world; ++r) so.base[size_t(r) + 1] = so.base[size_t(r)] + size_t(H(r, int(4 + 2 * ci)));
            const size_t bytes = so.base[W];
            if (bytes >= (size_t(1) << 31)) fail(NQE_ERR_ARROW, "Utf8 exchange: offsets overflow int32");
            d.values = dev_alloc_zero(ctx, (total + 1) * 4 + 8);
            d.data = dev_alloc(ctx, bytes + 8);
            d.data_length = int64_t(bytes);
            so.dst = (int32_t *)d.values->ptr;
            size_t mx = 0;
            for (int r = 0; r < world; ++r) mx = std::max(mx, ((rows[size_t(r)] + 1) * 4 + 7) / 8 * 8);
            so.stride = mx;
            so.stage = dev_alloc(ctx, W * mx + 8);
            if (total) {
                for (int r = 0; r < world; ++r) {
                    off[size_t(r)] = size_t(r) * so.stride;
                    sz[size_t(r)] = rows[size_t(r)] ? (rows[size_t(r)] + 1) * 4 : 0;
                }
                tr_check(cm, cm->tr.all_gather_v(cm->tr.user, c.values ? c.values->ptr : nullptr, sz[size_t(cm->rank)], so.stage->ptr, off.data(), sz.data(),
                                                 ctx->stream),
                         "all_gather_v(Utf8 offsets)");
                for (int r = 0; r < world; ++r) {
                    off[size_t(r)] = so.base[size_t(r)];
                    sz[size_t(r)] = so.base[size_t(r) + 1] - so.base[size_t(r)];
                }
                if (bytes)
                    tr_check(cm, cm->tr.all_gather_v(cm->tr.user, c.data ? (const uint8_t *)c.data->ptr + ufirst[ci] : nullptr, sz[size_t(cm->rank)], d.data->ptr,
                                                     off.data(), sz.data(), ctx->stream),
                             "all_gather_v(Utf8 bytes)");
            }
            staged_offsets.push_back(std::move(so));
        } else {
            fail(NQE_ERR_NOT_SUPPORTED, "exchange: unsupported column type");
        }
        if (any_valid) {
            gather_bits(c.valid(), d.validity, vhas, "all_gather_v(validity)");
            d.null_count = -1;
        }
        out->cols.push_back(std::move(d));
    }
    grp.end();
    // ---- bitmaps to their bit offsets (a rank without a validity bitmap contributes ones), Utf8 offsets behind the earlier ranks' bytes
    for (auto &sb : staged_bits)
        for (int r = 0; r < world; ++r)
            if (rows[size_t(r)])
                bitmap_place(ctx, sb.has[size_t(r)] ? (const uint8_t *)sb.stage->ptr + size_t(r) * sb.stride : (const uint8_t *)nullptr, sb.dst,
                             int64_t(roff[size_t(r)]), int64_t(rows[size_t(r)]));
    for (auto &so : staged_offsets)
        for (int r = 0; r < world; ++r)
            if (rows[size_t(r)])
                utf8_rebase_offsets(ctx, (const int32_t *)((const uint8_t *)so.stage->ptr + size_t(r) * so.stride), int64_t(rows[size_t(r)]),
                                    int32_t(so.base[size_t(r)]), so.dst + roff[size_t(r)]);
    // The sends read the local table's buffers on the context's stream.  Memory of the context's pool is recycled in stream order,
    // so the caller may release such a table at once; memory the caller lent (NQE_DEVICE columns) must stay untouched until the
    // transfers are done, and only the host knows what it does with it next: wait for those.
    bool lent = false;
    for (auto &c : local->cols)
        for (const BufRef *b : {&c.values, &c.validity, &c.data}) lent = lent || (*b && !(*b)->lib_memory);
    if (lent) sync(ctx);
    return out;
}

void check_status(nqe_ctx *ctx, nqe_status st) {
    if (st != NQE_OK) fail(st, ctx->last_error);
}

// runs the local, fallible part of a sharded operator; a failure becomes the status this rank brings to the exchange
template <typename Fn> LocalStatus run_local(Fn &&fn) {
    LocalStatus ls;
    try {
        fn();
    } catch (const Error &e) {
        ls.code = e.code;
        ls.msg = e.msg;
    } catch (const std::bad_alloc &) {
        ls.code = NQE_ERR_OUT_OF_MEMORY;
        ls.msg = "host out of memory";
    } catch (const std::exception &e) {
        ls.code = NQE_ERR_OTHERS;
        ls.msg = e.what();
    }
    return ls;
}

std::unique_ptr<nqe_comm> make_p2p_comm(nqe_ctx *ctx, std::unique_ptr<P2PState> st, bool is_rccl) {
    auto cm = std::make_unique<nqe_comm>();
    cm->ctx = ctx;
    cm->rank = st->rank;
    cm->world = st->world;
    cm->tr.all_gather = is_rccl ? rccl_all_gather : p2p_all_gather;
    cm->tr.all_gather_v = p2p_all_gather_v;
    cm->tr.group_begin = p2p_group_begin;
    cm->tr.group_end = p2p_group_end;
    cm->tr.destroy = p2p_destroy;
    cm->tr.user = st.release();
    cm->owns_user = true;
    cm->is_rccl = is_rccl;
    return cm;
}

} // namespace
} // namespace nqe

using namespace nqe;

extern "C" {

nqe_status nqe_comm_get_unique_id(void *id_out) {
    NQE_API_BEGIN(nullptr)
    if (!id_out) fail(NQE_ERR_INVALID_ARGUMENT, "id_out is NULL");
    static_assert(NQE_COMM_ID_BYTES == sizeof(ncclUniqueId), "unique id size");
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
    auto st = std::make_unique<P2PState>();
    st->rank = rank;
    st->world = world;
    ncclResult_t r = rccl().CommInitRank(&st->comm, world, id, rank);
    if (r != ncclSuccess) fail(NQE_ERR_RCCL, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
    st->p.user = st.get();
    st->p.send = rccl_send;
    st->p.recv = rccl_recv;
    st->p.group_begin = rccl_group_begin;
    st->p.group_end = rccl_group_end;
    *out = make_p2p_comm(ctx, std::move(st), true).release();
    NQE_API_END()
}

nqe_status nqe_comm_create_p2p(nqe_ctx *ctx, const nqe_p2p *p2p, int32_t rank, int32_t world, nqe_comm **out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !p2p || !out || world < 1 || rank < 0 || rank >= world || !p2p->send || !p2p->recv) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    auto st = std::make_unique<P2PState>();
    st->rank = rank;
    st->world = world;
    st->p = *p2p;
    *out = make_p2p_comm(ctx, std::move(st), false).release();
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
    if (!comm || !in || !out || num_aggs < 0 || num_aggs > 16 || (num_aggs > 0 && !aggs)) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    nqe_ctx *ctx = comm->ctx;
    const bool grouped = group && group_nodes > 0;
    // ---- local part: the partial aggregate (may fail on the data: DivideByZero, overflow, out of memory, ...)
    std::unique_ptr<nqe_table> state, keys;
    LocalStatus ls = run_local([&] {
        nqe_table *state_raw = nullptr, *keys_raw = nullptr;
        check_status(ctx, nqe_aggregate_partial(ctx, in, pred, pred_nodes, group, group_nodes, aggs, num_aggs, &state_raw, &keys_raw));
        state.reset(state_raw);
        keys.reset(keys_raw);
    });
    // the layout of the packed buffer follows from the aggregate list alone (a failed rank has no partial to look at): one key
    // column when grouped + {count, sum, min, max} per DISTINCT aggregated column
    int V = 0;
    for (int i = 0; i < num_aggs; ++i) {
        bool seen = false;
        for (int j = 0; j < i; ++j) seen = seen || aggs[j].column == aggs[i].column;
        V += seen ? 0 : 1;
    }
    const int nk = grouped ? 1 : 0;
    const int ncols = nk + 4 * V;
    const bool utf8_keys = ls.code == NQE_OK && grouped && keys && !keys->cols.empty() && keys->cols[0].dtype == NQE_UTF8;
    const int key_dtype = (ls.code == NQE_OK && grouped && !utf8_keys) ? keys->cols[0].dtype : NQE_INT64;
    const int64_t stride = NQE_EXCHANGE_ROWS;
    const size_t words = size_t(ncols) * size_t(stride) + 1;
    BufRef buf = dev_alloc(ctx, words * 8), gathered = dev_alloc(ctx, words * 8 * size_t(comm->world));
    const int64_t rows = ls.code == NQE_OK ? state->rows : 0;
    if (ls.code == NQE_OK && !utf8_keys && rows <= stride) {
        const nqe_table *tabs[2];
        int nt = 0;
        if (grouped) tabs[nt++] = keys.get();
        tabs[nt++] = state.get();
        LocalStatus ps = run_local([&] { check_status(ctx, nqe_table_pack_words(ctx, tabs, nt, stride, buf->ptr)); });
        if (ps.code != NQE_OK) ls = ps;
    }
    if (ls.code != NQE_OK || utf8_keys || rows > stride) {
        // header only: tells the peers that this exchange takes the exact-size path (more groups than the fixed buffer holds;
        // string keys, whose bytes the fixed buffer cannot carry; a failed rank, whose status travels in that path's header)
        const uint64_t count = ls.code != NQE_OK ? XCH_FAILED_COUNT : uint64_t(std::max<int64_t>(rows, stride + 1));
        launch(ctx, "exchange_store_word", store_word_kernel, dim3(1), dim3(1), 0, (uint64_t *)buf->ptr + (words - 1), count);
    }
    tr_check(comm, comm->tr.all_gather(comm->tr.user, buf->ptr, gathered->ptr, words * 8, ctx->stream), "all_gather(partial aggregate states)");
    nqe_table *merged = nullptr, *merged_keys = nullptr;
    if (ls.code == NQE_OK) {
        LocalStatus ms = run_local([&] {
            check_status(ctx, nqe_aggregate_merge_packed(ctx, gathered->ptr, comm->world, stride, grouped ? 1 : 0, key_dtype, aggs, num_aggs, &merged,
                                                         grouped ? &merged_keys : nullptr));
        });
        if (ms.code != NQE_OK) {
            // the merge failed HERE (out of memory, say).  Whether the peers go on to the exact-size exchange is a function of the
            // gathered counts, which this rank holds too: join them there with the failure, or fail alone when they are done
            std::vector<uint64_t> cnt(size_t(comm->world), 0);
            for (int r = 0; r < comm->world; ++r)
                (void)hipMemcpyAsync(&cnt[size_t(r)], (const uint64_t *)gathered->ptr + (size_t(r) + 1) * words - 1, 8, hipMemcpyDeviceToHost, ctx->stream);
            (void)hipStreamSynchronize(ctx->stream);
            bool exact = false;
            for (uint64_t c : cnt) exact = exact || c > uint64_t(stride);
            if (!exact) fail(ms.code, ms.msg);
            ls = ms;
            merged = nullptr;
        }
    }
    if (!merged) {
        // ---- exact-size exchange of the (keys, state) tables
        nqe_table both;
        both.ctx = ctx;
        both.rows = rows;
        if (ls.code == NQE_OK) {
            if (grouped) both.cols.push_back(keys->cols[0]);
            for (auto &c : state->cols) both.cols.push_back(c);
        }
        std::unique_ptr<nqe_table> all = all_gather_table(comm, &both, ls); // throws on every rank when any rank failed
        nqe_table kt, stt;
        kt.ctx = stt.ctx = ctx;
        kt.rows = stt.rows = all->rows;
        DevColumn strings;
        if (grouped) {
            if (utf8_keys) {
                // merge by string: every gathered key string → the first row that holds an equal string (exact, strings.hip); the
                // Int64 merge then runs on those codes, and the output keys are the representatives' strings
                strings = all->cols[0];
                Utf8Dict dict;
                kt.cols.push_back(utf8_encode_build(ctx, strings, &dict));
            } else
                kt.cols.push_back(all->cols[0]);
        }
        for (size_t c = size_t(nk); c < all->cols.size(); ++c) stt.cols.push_back(all->cols[c]);
        const nqe_table *sp = &stt, *kp = &kt;
        check_status(ctx, nqe_aggregate_merge(ctx, &sp, grouped ? &kp : nullptr, 1, aggs, num_aggs, &merged, &merged_keys));
        if (utf8_keys && merged_keys) {
            const DevColumn codes = merged_keys->cols[0];
            merged_keys->cols[0] = take_utf8(ctx, strings, (const int64_t *)codes.words(), codes.length, false);
        }
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
    std::unique_ptr<nqe_table> guard;
    LocalStatus ls = run_local([&] {
        nqe_table *local = nullptr;
        check_status(ctx, nqe_hash_join_probe(ctx, build, right_local, right_key, &local));
        guard.reset(local);
    });
    if (!gather) {
        if (ls.code != NQE_OK) fail(ls.code, ls.msg); // no exchange: nobody waits for this rank
        *out = guard.release();
    } else {
        *out = all_gather_table(comm, guard.get(), ls).release();
    }
    NQE_API_END()
}

nqe_status nqe_sharded_selection_projection_execute(nqe_comm *comm, const nqe_table *in_local, const nqe_expr_node *pred, int32_t pred_nodes,
                                                    const nqe_expr_node *nodes, const int32_t *expr_offsets, int32_t num_exprs, int32_t gather,
                                                    nqe_table **out) {
    NQE_API_BEGIN(comm ? comm->ctx : nullptr)
    if (!comm || !in_local || !out) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    nqe_ctx *ctx = comm->ctx;
    std::unique_ptr<nqe_table> guard;
    LocalStatus ls = run_local([&] {
        nqe_table *local = nullptr;
        check_status(ctx, nqe_selection_projection_execute(ctx, in_local, pred, pred_nodes, nodes, expr_offsets, num_exprs, &local));
        guard.reset(local);
    });
    if (!gather) {
        if (ls.code != NQE_OK) fail(ls.code, ls.msg);
        *out = guard.release();
    } else {
        *out = all_gather_table(comm, guard.get(), ls).release();
    }
    NQE_API_END()
}

} // extern "C"

// this translation unit's code object is loaded when a context is created, not by the first query that needs it (context.hip: load_modules)
NQE_MODULE_PROBE(nqe::store_word_kernel);
