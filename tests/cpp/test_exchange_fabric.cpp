// test_exchange_fabric.cpp — the sharded operators (csrc/exchange.hip) with world = 2, 3 and 8 ranks on ONE GPU.
//
// Ranks are threads, each with its own nqe context and a communicator made by nqe_comm_create_p2p over a test fabric that has
// RCCL's point-to-point semantics: a send is matched, per (sender, receiver) pair and in order, by the peer's receive of the SAME
// byte count; a size mismatch fails; a send or receive that never finds its partner fails after a timeout (RCCL would hang); calls
// between group_begin / group_end (nested) are issued together.  The library builds its collectives on these primitives with the
// very code that runs over ncclSend / ncclRecv (p2p_all_gather_v), so the pairing logic is exercised without a second GPU —
// which a collective-level transport (parallel.HostStagedTransport) cannot do.
//
// Checked: nqe_table_all_gather over every column type of the path (8-byte words, validity on some ranks only, Boolean, Utf8,
// empty ranks), the sharded aggregate (one-collective path, > NQE_EXCHANGE_ROWS groups, Utf8 keys, un-grouped), the sharded join
// and selection+projection gathered in rank order — all against the single-context operator over the whole table — and that a
// rank whose local work fails makes EVERY rank return an error (nobody blocks), with every message matched afterwards.
//
// Build: g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include test_exchange_fabric.cpp -lnqe_hip -lamdhip64 -lpthread
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nqe.h"

// ------------------------------------------------------------------------------------------------ the fabric
struct Msg {
    const void *buf;
    size_t bytes;
    bool done = false;
};

struct Fabric {
    int world;
    double timeout_s;
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::deque<std::shared_ptr<Msg>>> q; // [src * world + dst]: sends posted by src, not yet received by dst
    std::vector<std::string> errors;
    long sends = 0, recvs = 0;
    Fabric(int w, double t) : world(w), timeout_s(t), q(size_t(w) * size_t(w)) {}
    bool drained() {
        std::lock_guard<std::mutex> l(m);
        for (auto &d : q)
            if (!d.empty()) return false;
        return true;
    }
    void error(const std::string &e) { errors.push_back(e); }
};

struct Op {
    bool is_send;
    const void *sbuf;
    void *rbuf;
    size_t bytes;
    int peer;
    void *stream;
};

struct RankEnd {
    Fabric *f;
    int rank;
    int depth = 0;
    std::vector<Op> pending;
    hipStream_t copy_stream = nullptr;

    int32_t flush() {
        std::vector<Op> ops;
        ops.swap(pending);
        if (ops.empty()) return 0;
        // what this rank sends must have been produced: the operations are ordered on its stream
        for (auto &o : ops)
            if (hipStreamSynchronize(static_cast<hipStream_t>(o.stream)) != hipSuccess) return 90;
        std::vector<std::shared_ptr<Msg>> mine;
        {
            std::lock_guard<std::mutex> l(f->m);
            for (auto &o : ops)
                if (o.is_send) {
                    auto msg = std::make_shared<Msg>();
                    msg->buf = o.sbuf;
                    msg->bytes = o.bytes;
                    f->q[size_t(rank) * size_t(f->world) + size_t(o.peer)].push_back(msg);
                    mine.push_back(msg);
                    ++f->sends;
                }
        }
        f->cv.notify_all();
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(f->timeout_s);
        int32_t rc = 0;
        for (auto &o : ops) {
            if (o.is_send) continue;
            std::shared_ptr<Msg> msg;
            {
                std::unique_lock<std::mutex> l(f->m);
                auto &dq = f->q[size_t(o.peer) * size_t(f->world) + size_t(rank)];
                if (!f->cv.wait_until(l, deadline, [&] { return !dq.empty(); })) {
                    f->error("rank " + std::to_string(rank) + ": recv of " + std::to_string(o.bytes) + " bytes from rank " + std::to_string(o.peer) + " never matched");
                    rc = 91;
                    break;
                }
                msg = dq.front();
                dq.pop_front();
                ++f->recvs;
                if (msg->bytes != o.bytes) {
                    f->error("rank " + std::to_string(rank) + ": recv of " + std::to_string(o.bytes) + " bytes matched a send of " + std::to_string(msg->bytes) +
                             " from rank " + std::to_string(o.peer));
                    msg->done = true;
                    rc = 92;
                }
            }
            if (rc) {
                f->cv.notify_all();
                break;
            }
            // (a device-to-device hipMemcpy may return before the copy has run: copy on a stream of the fabric's own and wait for it —
            // the sender reuses its buffer, and this rank's stream reads the destination, as soon as the message counts as done)
            if (!copy_stream && hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess) return 95;
            hipError_t e = hipMemcpyAsync(o.rbuf, msg->buf, o.bytes, hipMemcpyDeviceToDevice, copy_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(copy_stream);
            {
                std::lock_guard<std::mutex> l(f->m);
                msg->done = true;
                if (e != hipSuccess) {
                    f->error("hipMemcpy failed");
                    rc = 93;
                }
            }
            f->cv.notify_all();
            if (rc) break;
        }
        // a send completes when its receiver has copied it (its buffer may be reused after group_end returns)
        {
            std::unique_lock<std::mutex> l(f->m);
            for (auto &msg : mine)
                if (!f->cv.wait_until(l, deadline, [&] { return msg->done; })) {
                    f->error("rank " + std::to_string(rank) + ": a send of " + std::to_string(msg->bytes) + " bytes was never received");
                    if (!rc) rc = 94;
                    break;
                }
        }
        return rc;
    }
};

static int32_t fab_send(void *u, const void *buf, size_t bytes, int32_t peer, void *stream) {
    auto *r = static_cast<RankEnd *>(u);
    r->pending.push_back(Op{true, buf, nullptr, bytes, peer, stream});
    return r->depth == 0 ? r->flush() : 0;
}
static int32_t fab_recv(void *u, void *buf, size_t bytes, int32_t peer, void *stream) {
    auto *r = static_cast<RankEnd *>(u);
    r->pending.push_back(Op{false, nullptr, buf, bytes, peer, stream});
    return r->depth == 0 ? r->flush() : 0;
}
static int32_t fab_begin(void *u) {
    ++static_cast<RankEnd *>(u)->depth;
    return 0;
}
static int32_t fab_end(void *u) {
    auto *r = static_cast<RankEnd *>(u);
    if (--r->depth > 0) return 0;
    return r->flush();
}

// ------------------------------------------------------------------------------------------------ host columns
struct HostCol {
    int dtype = NQE_INT64;
    int64_t n = 0;
    std::vector<uint64_t> words;  // word types
    std::vector<uint8_t> bits;    // Boolean values
    std::vector<int32_t> offsets; // Utf8
    std::string data;             // Utf8
    std::vector<uint8_t> valid;   // empty = no nulls
    bool is_valid(int64_t i) const { return valid.empty() || ((valid[size_t(i >> 3)] >> (i & 7)) & 1); }
    bool bit(int64_t i) const { return (bits[size_t(i >> 3)] >> (i & 7)) & 1; }
    std::string str(int64_t i) const { return data.substr(size_t(offsets[size_t(i)]), size_t(offsets[size_t(i) + 1] - offsets[size_t(i)])); }
};

static uint64_t mix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

static void set_bit(std::vector<uint8_t> &b, int64_t i, bool v) {
    if (v) b[size_t(i >> 3)] |= uint8_t(1u << (i & 7));
}

// rows [lo, hi) of `c` as a column of its own (offset-free, as a rank would hold its shard)
static HostCol slice(const HostCol &c, int64_t lo, int64_t hi) {
    HostCol s;
    s.dtype = c.dtype;
    s.n = hi - lo;
    bool any_null = false;
    for (int64_t i = lo; i < hi; ++i) any_null = any_null || !c.is_valid(i);
    if (any_null) {
        s.valid.assign(size_t((s.n + 7) / 8), 0);
        for (int64_t i = lo; i < hi; ++i) set_bit(s.valid, i - lo, c.is_valid(i));
    }
    if (c.dtype == NQE_BOOLEAN) {
        s.bits.assign(size_t((s.n + 7) / 8) + 1, 0);
        for (int64_t i = lo; i < hi; ++i) set_bit(s.bits, i - lo, c.bit(i));
    } else if (c.dtype == NQE_UTF8) {
        s.offsets.push_back(0);
        for (int64_t i = lo; i < hi; ++i) {
            s.data += c.str(i);
            s.offsets.push_back(int32_t(s.data.size()));
        }
    } else
        s.words.assign(c.words.begin() + lo, c.words.begin() + hi);
    return s;
}

static nqe_column describe(const HostCol &c) {
    nqe_column d;
    std::memset(&d, 0, sizeof(d));
    d.dtype = c.dtype;
    d.location = NQE_HOST;
    d.length = c.n;
    d.null_count = c.valid.empty() ? 0 : -1;
    d.validity = c.valid.empty() ? nullptr : c.valid.data();
    if (c.dtype == NQE_BOOLEAN) d.values = c.bits.data();
    else if (c.dtype == NQE_UTF8) {
        d.values = c.offsets.data();
        d.data = c.data.data();
        d.data_length = int64_t(c.data.size());
    } else
        d.values = c.words.data();
    return d;
}

static nqe_table *upload(nqe_ctx *ctx, const std::vector<HostCol> &cols) {
    std::vector<nqe_column> d;
    for (auto &c : cols) d.push_back(describe(c));
    nqe_table *t = nullptr;
    if (nqe_table_create(ctx, d.data(), int32_t(d.size()), &t) != NQE_OK) {
        std::fprintf(stderr, "nqe_table_create: %s\n", nqe_last_error(ctx));
        std::abort();
    }
    return t;
}

static HostCol download(const nqe_table *t, int i) {
    nqe_column d;
    if (nqe_table_column(t, i, &d) != NQE_OK) std::abort();
    HostCol c;
    c.dtype = d.dtype;
    c.n = d.length;
    if (d.validity) c.valid.assign(size_t((c.n + 7) / 8), 0);
    void *vals = nullptr;
    if (c.dtype == NQE_BOOLEAN) {
        c.bits.assign(size_t((c.n + 7) / 8) + 1, 0);
        vals = c.bits.data();
    } else if (c.dtype == NQE_UTF8) {
        c.offsets.assign(size_t(c.n) + 1, 0);
        c.data.assign(size_t(d.data_length), '\0');
        vals = c.offsets.data();
    } else {
        c.words.assign(size_t(c.n), 0);
        vals = c.words.data();
    }
    if (nqe_table_download_column(t, i, vals, c.valid.empty() ? nullptr : c.valid.data(), c.dtype == NQE_UTF8 ? &c.data[0] : nullptr) != NQE_OK) std::abort();
    return c;
}

// row-wise equality of the valid rows, validity bit for bit
static bool same_column(const HostCol &a, const HostCol &b, std::string *why, double rtol = 0.0) {
    if (a.dtype != b.dtype || a.n != b.n) {
        *why = "dtype/length differ (" + std::to_string(a.n) + " vs " + std::to_string(b.n) + ")";
        return false;
    }
    for (int64_t i = 0; i < a.n; ++i) {
        if (a.is_valid(i) != b.is_valid(i)) {
            *why = "validity differs at row " + std::to_string(i);
            return false;
        }
        if (!a.is_valid(i)) continue;
        bool eq;
        if (a.dtype == NQE_BOOLEAN) eq = a.bit(i) == b.bit(i);
        else if (a.dtype == NQE_UTF8) eq = a.str(i) == b.str(i);
        else if (rtol > 0 && a.dtype == NQE_FLOAT64) {
            double x, y;
            std::memcpy(&x, &a.words[size_t(i)], 8);
            std::memcpy(&y, &b.words[size_t(i)], 8);
            eq = (x == y) || (std::isnan(x) && std::isnan(y)) || std::fabs(x - y) <= rtol * std::fmax(std::fabs(x), std::fabs(y));
        } else
            eq = a.words[size_t(i)] == b.words[size_t(i)];
        if (!eq) {
            *why = "value differs at row " + std::to_string(i);
            return false;
        }
    }
    return true;
}

// rows of a grouped aggregate (keys + output columns of 8-byte words) reordered by key string: Utf8 group keys come out ordered by a
// representative row that the insertion race picks, so the two sides are compared as maps string -> row
static bool same_by_string_key(const nqe_table *got_keys, const nqe_table *got, const nqe_table *exp_keys, const nqe_table *exp, std::string *why, double rtol) {
    HostCol gk = download(got_keys, 0), ek = download(exp_keys, 0);
    if (gk.n != ek.n) {
        *why = "group count differs (" + std::to_string(gk.n) + " vs " + std::to_string(ek.n) + ")";
        return false;
    }
    auto order = [](const HostCol &k) {
        std::vector<int64_t> p(size_t(k.n));
        for (int64_t i = 0; i < k.n; ++i) p[size_t(i)] = i;
        std::sort(p.begin(), p.end(), [&](int64_t a, int64_t b) { return k.str(a) < k.str(b); });
        return p;
    };
    const std::vector<int64_t> gp = order(gk), ep = order(ek);
    for (int64_t i = 0; i < gk.n; ++i)
        if (gk.str(gp[size_t(i)]) != ek.str(ep[size_t(i)])) {
            *why = "key sets differ";
            return false;
        }
    for (int c = 0; c < nqe_table_num_columns(got); ++c) {
        HostCol g = download(got, c), e = download(exp, c), g2 = g, e2 = e;
        for (int64_t i = 0; i < gk.n; ++i) {
            g2.words[size_t(i)] = g.words[size_t(gp[size_t(i)])];
            e2.words[size_t(i)] = e.words[size_t(ep[size_t(i)])];
        }
        std::string w;
        if (!same_column(g2, e2, &w, rtol)) {
            *why = "column " + std::to_string(c) + ": " + w;
            return false;
        }
    }
    return true;
}

static bool same_table(const nqe_table *got, const nqe_table *exp, std::string *why, double rtol = 0.0) {
    if (nqe_table_num_columns(got) != nqe_table_num_columns(exp)) {
        *why = "column count differs";
        return false;
    }
    for (int c = 0; c < nqe_table_num_columns(got); ++c) {
        std::string w;
        if (!same_column(download(got, c), download(exp, c), &w, rtol)) {
            *why = "column " + std::to_string(c) + ": " + w;
            return false;
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ data
// id Int64 (row number) | v Float64 nullable, nulls only in the first 40 % of the rows | b Boolean | nb Boolean nullable | s Utf8 (few
// distinct values, some empty) | ns Utf8 nullable | k Int64 (id % 7000)
static std::vector<HostCol> make_table(int64_t n) {
    std::vector<HostCol> t(7);
    const char *names[] = {"", "lynne", "bob", "alice", "véé", "grandmaster", "x"};
    for (auto &c : t) c.n = n;
    t[0].dtype = NQE_INT64;
    t[1].dtype = NQE_FLOAT64;
    t[2].dtype = t[3].dtype = NQE_BOOLEAN;
    t[4].dtype = t[5].dtype = NQE_UTF8;
    t[6].dtype = NQE_INT64;
    t[1].valid.assign(size_t((n + 7) / 8), 0);
    t[3].valid.assign(size_t((n + 7) / 8), 0);
    t[5].valid.assign(size_t((n + 7) / 8), 0);
    t[2].bits.assign(size_t((n + 7) / 8) + 1, 0);
    t[3].bits.assign(size_t((n + 7) / 8) + 1, 0);
    t[4].offsets.push_back(0);
    t[5].offsets.push_back(0);
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t h = mix(uint64_t(i));
        t[0].words.push_back(uint64_t(i));
        double v = double(h >> 11) * 0x1.0p-53 * 200.0 - 100.0;
        uint64_t w;
        std::memcpy(&w, &v, 8);
        t[1].words.push_back(w);
        set_bit(t[1].valid, i, !(i < n * 2 / 5 && h % 13 == 0));
        set_bit(t[2].bits, i, (h >> 5) & 1);
        set_bit(t[3].bits, i, (h >> 6) & 1);
        set_bit(t[3].valid, i, h % 11 != 0);
        t[4].data += names[(h >> 8) % 7];
        if ((h >> 12) % 3 == 0) t[4].data += std::to_string((h >> 16) % 50);
        t[4].offsets.push_back(int32_t(t[4].data.size()));
        const bool sv = h % 17 != 0;
        set_bit(t[5].valid, i, sv);
        if (sv) t[5].data += names[(h >> 20) % 7];
        t[5].offsets.push_back(int32_t(t[5].data.size()));
        t[6].words.push_back(uint64_t(i % 7000));
    }
    return t;
}

// ragged shards, one of them EMPTY when there are at least three ranks
static std::vector<int64_t> shard_bounds(int64_t n, int world) {
    std::vector<int64_t> b(static_cast<size_t>(world) + 1, 0);
    std::vector<int64_t> w(static_cast<size_t>(world), 0);
    int64_t tot = 0;
    for (int r = 0; r < world; ++r) {
        w[size_t(r)] = (world >= 3 && r == 1) ? 0 : 3 + (r * 7) % 5;
        tot += w[size_t(r)];
    }
    int64_t acc = 0;
    for (int r = 0; r < world; ++r) {
        acc += w[size_t(r)];
        b[size_t(r) + 1] = n * acc / tot;
    }
    b[size_t(world)] = n;
    return b;
}

// ------------------------------------------------------------------------------------------------ expression helpers
static nqe_expr_node e_col(int c) {
    nqe_expr_node n;
    std::memset(&n, 0, sizeof(n));
    n.kind = NQE_EXPR_COLUMN;
    n.column = c;
    return n;
}
static nqe_expr_node e_i64(int64_t v) {
    nqe_expr_node n;
    std::memset(&n, 0, sizeof(n));
    n.kind = NQE_EXPR_LITERAL;
    n.dtype = NQE_INT64;
    n.value.i64 = v;
    return n;
}
static nqe_expr_node e_op(int op) {
    nqe_expr_node n;
    std::memset(&n, 0, sizeof(n));
    n.kind = NQE_EXPR_BINARY;
    n.op = op;
    return n;
}

// ------------------------------------------------------------------------------------------------ running ranks
struct RankCtx {
    int rank, world;
    nqe_ctx *ctx;
    nqe_comm *comm;
    std::vector<HostCol> shard;
    nqe_table *table;
};

static int failures = 0, checks = 0;
static std::mutex report_m;
static void check(bool ok, const std::string &what) {
    std::lock_guard<std::mutex> l(report_m);
    ++checks;
    if (!ok) {
        ++failures;
        std::printf("  FAIL: %s\n", what.c_str());
    }
}

// runs fn(rank context) on `world` threads over one fabric; returns the fabric's error list; verifies that every message found
// its partner
static std::vector<std::string> run_ranks(int world, const std::vector<HostCol> &whole, double timeout_s, const std::function<void(RankCtx &)> &fn,
                                          bool expect_drained = true) {
    Fabric fab(world, timeout_s);
    std::vector<RankEnd> ends(static_cast<size_t>(world), RankEnd{&fab, 0, 0, {}, nullptr});
    const std::vector<int64_t> b = shard_bounds(whole[0].n, world);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            (void)hipSetDevice(0);
            RankCtx rc;
            rc.rank = r;
            rc.world = world;
            if (nqe_ctx_create(0, nullptr, &rc.ctx) != NQE_OK) std::abort();
            ends[size_t(r)].rank = r;
            nqe_p2p p;
            std::memset(&p, 0, sizeof(p));
            p.user = &ends[size_t(r)];
            p.send = fab_send;
            p.recv = fab_recv;
            p.group_begin = fab_begin;
            p.group_end = fab_end;
            if (nqe_comm_create_p2p(rc.ctx, &p, r, world, &rc.comm) != NQE_OK) std::abort();
            for (auto &c : whole) rc.shard.push_back(slice(c, b[size_t(r)], b[size_t(r) + 1]));
            rc.table = upload(rc.ctx, rc.shard);
            fn(rc);
            nqe_table_release(rc.table);
            nqe_comm_destroy(rc.comm);
            nqe_ctx_destroy(rc.ctx);
        });
    for (auto &t : th) t.join();
    if (expect_drained) check(fab.drained(), "world " + std::to_string(world) + ": every send was matched by a receive");
    return fab.errors;
}

int main() {
    (void)hipSetDevice(0);
    nqe_ctx *ref = nullptr;
    if (nqe_ctx_create(0, nullptr, &ref) != NQE_OK) {
        std::fprintf(stderr, "no device: %s\n", nqe_last_global_error());
        return 2;
    }
    const int64_t N = 50021;
    const std::vector<HostCol> whole = make_table(N);
    nqe_table *whole_t = upload(ref, whole);
    const nqe_aggregate aggs[] = {{NQE_AGG_COUNT, 1}, {NQE_AGG_SUM, 1}, {NQE_AGG_AVG, 1}, {NQE_AGG_MIN, 1}, {NQE_AGG_MAX, 1}, {NQE_AGG_COUNT, 4}};
    const int naggs = 6;

    for (int world : {2, 3, 8}) {
        std::printf("world = %d\n", world);
        const std::string W = "world " + std::to_string(world) + ": ";
        // ---- 1. the table itself, every column type, gathered in rank order
        auto errs = run_ranks(world, whole, 30.0, [&](RankCtx &rc) {
            nqe_table *all = nullptr;
            nqe_status st = nqe_table_all_gather(rc.comm, rc.table, &all);
            check(st == NQE_OK, W + "nqe_table_all_gather: " + (st == NQE_OK ? "" : nqe_last_error(rc.ctx)));
            if (st != NQE_OK) return;
            std::string why;
            bool ok = nqe_table_num_rows(all) == N;
            for (int c = 0; ok && c < int(whole.size()); ++c) ok = same_column(download(all, c), whole[size_t(c)], &why);
            check(ok, W + "rank " + std::to_string(rc.rank) + " gathered table equals the whole table " + why);
            nqe_table_release(all);
        });
        check(errs.empty(), W + "no fabric errors (table) " + (errs.empty() ? "" : errs[0]));

        // ---- 2. sharded aggregate: few groups (one collective), > NQE_EXCHANGE_ROWS groups (exact exchange), Utf8 keys, un-grouped
        struct AggCase {
            const char *name;
            std::vector<nqe_expr_node> key;
        };
        std::vector<AggCase> cases = {{"id % 100", {e_col(0), e_i64(100), e_op(NQE_OP_MODULOS)}},
                                      {"k (7000 groups)", {e_col(6)}},
                                      {"s (Utf8 keys)", {e_col(4)}},
                                      {"un-grouped", {}}};
        const nqe_expr_node pred[] = {e_col(0), e_i64(N - 1000), e_op(NQE_OP_LT)};
        for (auto &cs : cases) {
            nqe_table *exp = nullptr, *exp_keys = nullptr;
            const bool grouped = !cs.key.empty();
            if (nqe_aggregate_execute(ref, whole_t, pred, 3, grouped ? cs.key.data() : nullptr, int32_t(cs.key.size()), aggs, naggs, &exp, &exp_keys) != NQE_OK) {
                std::printf("reference aggregate failed: %s\n", nqe_last_error(ref));
                return 1;
            }
            errs = run_ranks(world, whole, 30.0, [&](RankCtx &rc) {
                nqe_table *out = nullptr, *keys = nullptr;
                nqe_status st = nqe_sharded_aggregate_execute(rc.comm, rc.table, pred, 3, grouped ? cs.key.data() : nullptr, int32_t(cs.key.size()), aggs, naggs, &out,
                                                              &keys);
                check(st == NQE_OK, W + "sharded aggregate [" + cs.name + "]: " + (st == NQE_OK ? "" : nqe_last_error(rc.ctx)));
                if (st != NQE_OK) return;
                std::string why;
                bool ok;
                if (grouped && cs.key[0].kind == NQE_EXPR_COLUMN && cs.key.size() == 1 && whole[size_t(cs.key[0].column)].dtype == NQE_UTF8) {
                    ok = same_by_string_key(keys, out, exp_keys, exp, &why, 1e-9);
                } else
                    ok = (!grouped || same_table(keys, exp_keys, &why)) && same_table(out, exp, &why, 1e-9);
                check(ok, W + "rank " + std::to_string(rc.rank) + " sharded aggregate [" + cs.name + "] equals the single-context result " + why);
                nqe_table_release(out);
                if (keys) nqe_table_release(keys);
            });
            check(errs.empty(), W + "no fabric errors (aggregate " + cs.name + ") " + (errs.empty() ? "" : errs[0]));
            nqe_table_release(exp);
            if (exp_keys) nqe_table_release(exp_keys);
        }

        // ---- 3. join (build replicated: dim(k unique, name Utf8, w Float64 nullable); probe = the shard, key column 6) and
        //         selection + projection, gathered
        std::vector<HostCol> dim(3);
        const int64_t NB = 6500; // keys 6500..6999 of the probe side find no partner
        dim[0].dtype = NQE_INT64;
        dim[1].dtype = NQE_UTF8;
        dim[2].dtype = NQE_FLOAT64;
        dim[1].offsets.push_back(0);
        dim[2].valid.assign(size_t((NB + 7) / 8), 0);
        for (int64_t i = 0; i < NB; ++i) {
            const int64_t key = int64_t(mix(uint64_t(i) + 99) % 1000003) * 0 + (i * 4999) % NB; // a permutation of 0..NB-1 (4999 and 6500 coprime)
            dim[0].words.push_back(uint64_t(key));
            dim[1].data += "dept" + std::to_string(key % 37);
            dim[1].offsets.push_back(int32_t(dim[1].data.size()));
            double w = double(key) * 0.5;
            uint64_t ww;
            std::memcpy(&ww, &w, 8);
            dim[2].words.push_back(ww);
            set_bit(dim[2].valid, i, key % 9 != 0);
        }
        for (auto &c : dim) c.n = NB;
        nqe_table *dim_ref = upload(ref, dim), *join_exp = nullptr;
        if (nqe_hash_join_execute(ref, dim_ref, whole_t, 0, 6, &join_exp) != NQE_OK) {
            std::printf("reference join failed: %s\n", nqe_last_error(ref));
            return 1;
        }
        const nqe_expr_node spred[] = {e_col(6), e_i64(3000), e_op(NQE_OP_LT)};
        const nqe_expr_node sproj[] = {e_col(0), e_i64(100), e_op(NQE_OP_PLUS), e_col(4), e_col(1), e_col(3)};
        const int32_t soff[] = {0, 3, 4, 5, 6};
        nqe_table *sp_exp = nullptr;
        if (nqe_selection_projection_execute(ref, whole_t, spred, 3, sproj, soff, 4, &sp_exp) != NQE_OK) {
            std::printf("reference selection failed: %s\n", nqe_last_error(ref));
            return 1;
        }
        errs = run_ranks(world, whole, 30.0, [&](RankCtx &rc) {
            nqe_table *dim_t = upload(rc.ctx, dim);
            nqe_join_table *jt = nullptr;
            if (nqe_hash_join_build(rc.ctx, dim_t, 0, &jt) != NQE_OK) std::abort();
            nqe_table *out = nullptr;
            nqe_status st = nqe_sharded_hash_join_probe(rc.comm, jt, rc.table, 6, 1, &out);
            check(st == NQE_OK, W + "sharded join: " + (st == NQE_OK ? "" : nqe_last_error(rc.ctx)));
            std::string why;
            if (st == NQE_OK) {
                check(same_table(out, join_exp, &why), W + "rank " + std::to_string(rc.rank) + " gathered join output equals the single-context join, row order included " + why);
                nqe_table_release(out);
            }
            nqe_join_table_release(jt);
            nqe_table_release(dim_t);
            st = nqe_sharded_selection_projection_execute(rc.comm, rc.table, spred, 3, sproj, soff, 4, 1, &out);
            check(st == NQE_OK, W + "sharded selection+projection: " + (st == NQE_OK ? "" : nqe_last_error(rc.ctx)));
            if (st == NQE_OK) {
                check(same_table(out, sp_exp, &why), W + "rank " + std::to_string(rc.rank) + " gathered selection+projection equals the single-context result " + why);
                nqe_table_release(out);
            }
        });
        check(errs.empty(), W + "no fabric errors (join, selection) " + (errs.empty() ? "" : errs[0]));
        nqe_table_release(join_exp);
        nqe_table_release(sp_exp);
        nqe_table_release(dim_ref);

        // ---- 4. a rank whose local work fails: `100 / (id - K)` divides by zero on the rank that holds row K only.  Every rank must
        //         return an error — the failing one its own (ArrowError), the others NQE_ERR_RCCL — and nobody may block
        const std::vector<int64_t> b = shard_bounds(N, world);
        const int bad_rank = world - 1;
        const int64_t K = (b[size_t(bad_rank)] + b[size_t(bad_rank) + 1]) / 2;
        const nqe_expr_node fproj[] = {e_i64(100), e_col(0), e_i64(K), e_op(NQE_OP_MINUS), e_op(NQE_OP_DIVIDE)};
        const int32_t foff[] = {0, 5};
        const nqe_expr_node fpred[] = {e_i64(100), e_col(0), e_i64(K), e_op(NQE_OP_MINUS), e_op(NQE_OP_DIVIDE), e_i64(1000), e_op(NQE_OP_LT)};
        std::atomic<int> own_error{0}, peer_error{0}, ok_count{0};
        const auto t0 = std::chrono::steady_clock::now();
        errs = run_ranks(world, whole, 20.0, [&](RankCtx &rc) {
            nqe_table *out = nullptr;
            const nqe_expr_node all_rows[] = {e_col(0), e_i64(0), e_op(NQE_OP_GT_EQ)};
            nqe_status st = nqe_sharded_selection_projection_execute(rc.comm, rc.table, all_rows, 3, fproj, foff, 1, 1, &out);
            if (st == NQE_OK) {
                ++ok_count;
                nqe_table_release(out);
            } else if (rc.rank == bad_rank && st == NQE_ERR_ARROW) ++own_error;
            else if (rc.rank != bad_rank && st == NQE_ERR_RCCL) ++peer_error;
            // the same through the aggregate (its packed exchange carries the failure to the exact-size path's header)
            nqe_table *o2 = nullptr, *k2 = nullptr;
            const nqe_expr_node key[] = {e_col(0), e_i64(100), e_op(NQE_OP_MODULOS)};
            st = nqe_sharded_aggregate_execute(rc.comm, rc.table, fpred, 7, key, 3, aggs, naggs, &o2, &k2);
            if (st == NQE_OK) {
                ++ok_count;
                nqe_table_release(o2);
                if (k2) nqe_table_release(k2);
            } else if (rc.rank == bad_rank && st == NQE_ERR_ARROW) ++own_error;
            else if (rc.rank != bad_rank && st == NQE_ERR_RCCL) ++peer_error;
            // and the communicator still works afterwards
            nqe_table *all = nullptr;
            st = nqe_table_all_gather(rc.comm, rc.table, &all);
            check(st == NQE_OK && nqe_table_num_rows(all) == N, W + "the communicator works after a collective failure");
            if (st == NQE_OK) nqe_table_release(all);
        });
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        check(ok_count == 0 && own_error == 2 && peer_error == 2 * (world - 1),
              W + "a failing rank fails every rank (own " + std::to_string(own_error.load()) + ", peers " + std::to_string(peer_error.load()) + ", ok " +
                  std::to_string(ok_count.load()) + ")");
        check(errs.empty() && secs < 15.0, W + "nobody waited for the failed rank (" + std::to_string(secs) + " s) " + (errs.empty() ? "" : errs[0]));
    }

    // ---- 5. the fabric itself notices what RCCL would hang on: one rank never joins the exchange
    {
        auto errs = run_ranks(2, whole, 1.5, [&](RankCtx &rc) {
            if (rc.rank == 1) return;
            nqe_table *all = nullptr;
            nqe_status st = nqe_table_all_gather(rc.comm, rc.table, &all);
            check(st == NQE_ERR_RCCL, "an exchange a peer never joins fails with NQE_ERR_RCCL instead of hanging");
            if (st == NQE_OK) nqe_table_release(all);
        }, false);
        check(!errs.empty(), "the fabric reports the unmatched message");
    }

    nqe_table_release(whole_t);
    nqe_ctx_destroy(ref);
    std::printf("%d/%d checks passed\n", checks - failures, checks);
    return failures ? 1 : 0;
}
