// expr_jit.hpp — run-time specialisation of the expression machine (included by expr.hip, host code only).
//
// The stack machine of expr.hip (expr_tree_kernel) interprets an ExProgram per 256-row chunk: operand fetch, stack moves and the
// (operator, dtype) dispatch are wave-uniform branches around register copies — ≈ 10-14 vector instructions per program step per
// row against 1-2 for the arithmetic itself, so a tree of more than ~4 operators is issue-bound (an 8-operator chain ran at
// 2.7 TB/s, `(id % 1000) * 3 + id / 7` at 2.9).  For such trees over large inputs the SAME program is turned into straight-line
// HIP source — one SSA value per step, operands named at generation time, literal divisors baked in so that the compiler emits
// its own multiply-shift sequences — and compiled for gfx950 with hipRTC (libhiprtc.so, loaded on first use).
//
// Compilation (≈ 0.3 s, 2-3 s for the first one of a process) happens on a worker thread that makes no HIP calls; executions
// of the query shape keep using the interpreter until the code object is ready, then the calling thread loads it
// (hipModuleLoadData) and launches it with hipModuleLaunchKernel on the context's stream.  Kernels are cached in the context by
// a hash of everything baked into the source (steps, dtypes, which columns carry validity, output form, baked divisors); other
// literals and all pointers are kernel arguments, so `v * 2 > K` does not compile once per K.  Any failure — no libhiprtc, a
// compile error — marks the entry failed and the interpreter stays: results never depend on the specialisation.
// Semantics are those of apply_binary / ex_combine (device_utils.hpp, expr.hip), restated in the generated source; the parity
// tests run every tree through both forms.
#pragma once

#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <sstream>
#include <thread>

namespace nqe {
namespace {

struct HipRtcApi {
    decltype(&hiprtcCreateProgram) create = nullptr;
    decltype(&hiprtcCompileProgram) compile = nullptr;
    decltype(&hiprtcGetCodeSize) code_size = nullptr;
    decltype(&hiprtcGetCode) code = nullptr;
    decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
    decltype(&hiprtcGetProgramLog) log = nullptr;
    decltype(&hiprtcDestroyProgram) destroy = nullptr;
    bool ok = false;
};

// (called from worker threads: initialised once)
const HipRtcApi &hiprtc_api() {
    static HipRtcApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = dlopen("libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        api.create = reinterpret_cast<decltype(api.create)>(dlsym(h, "hiprtcCreateProgram"));
        api.compile = reinterpret_cast<decltype(api.compile)>(dlsym(h, "hiprtcCompileProgram"));
        api.code_size = reinterpret_cast<decltype(api.code_size)>(dlsym(h, "hiprtcGetCodeSize"));
        api.code = reinterpret_cast<decltype(api.code)>(dlsym(h, "hiprtcGetCode"));
        api.log_size = reinterpret_cast<decltype(api.log_size)>(dlsym(h, "hiprtcGetProgramLogSize"));
        api.log = reinterpret_cast<decltype(api.log)>(dlsym(h, "hiprtcGetProgramLog"));
        api.destroy = reinterpret_cast<decltype(api.destroy)>(dlsym(h, "hiprtcDestroyProgram"));
        api.ok = api.create && api.compile && api.code_size && api.code && api.log_size && api.log && api.destroy;
    });
    return api;
}

constexpr int JIT_MAX_LITS = 2 * EX_MAX_INSTR;
// the generated kernel's one argument (restated in the source: keep in step with gen_source)
struct JitArgs {
    const void *col[EX_MAX_COLS];
    const uint8_t *valid[EX_MAX_COLS];
    uint64_t lit[JIT_MAX_LITS]; // lit[2 * step] = lit_a, lit[2 * step + 1] = lit_b
    int64_t n;
    uint64_t *out_words, *out_bits, *out_valid;
    int *flags;
};

struct JitEntry {
    std::thread worker;
    std::atomic<int> state{0}; // 0: compiling, 1: code ready, 2: loaded, -1: failed
    std::string disk_path; // where the code object is kept between processes (empty: nowhere)
    std::string extra_opt; // one more hipRTC option (the aggregate kernel: -munsafe-fp-atomics → ds_add_f64 / ds_min_f64 / ds_max_f64)
    std::string source, log, arch; // source: kept for the entry's lifetime — a lookup compares it (a 64-bit hash alone could collide)
    std::vector<char> code;
    bool from_disk = false; // `code` was read from disk_path (not compiled by this process): a failed load deletes the file and compiles
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
};

struct JitCache {
    std::map<uint64_t, std::unique_ptr<JitEntry>> entries;
    bool unavailable = false;
    ~JitCache() {
        for (auto &kv : entries) {
            if (kv.second->worker.joinable()) kv.second->worker.join();
            if (kv.second->mod) (void)hipModuleUnload(kv.second->mod);
        }
    }
};

JitCache *jit_cache(nqe_ctx *ctx) {
    if (!ctx->jit) ctx->jit = std::shared_ptr<void>(new JitCache, [](void *p) { delete static_cast<JitCache *>(p); });
    return static_cast<JitCache *>(ctx->jit.get());
}

// is this literal operand baked into the source?  Only the divisors of integer `/` and `%` (the compiler's multiply-shift forms)
bool jit_bakes_b(const ExInstr &in) {
    return in.b_src == EX_LIT && (in.op == NQE_OP_DIVIDE || in.op == NQE_OP_MODULOS) && in.dt != NQE_FLOAT64;
}

uint64_t jit_hash(const ExProgram &P, bool nulls, bool bool_out) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t nb) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < nb; ++i) h = (h ^ b[i]) * 1099511628211ull;
    };
    const int32_t head[4] = {P.n, P.ncols, nulls ? 1 : 0, bool_out ? 1 : 0};
    mix(head, sizeof(head));
    for (int c = 0; c < P.ncols; ++c) {
        const int32_t cd[2] = {P.col_dtype[c], (nulls && P.col_valid[c]) ? 1 : 0};
        mix(cd, sizeof(cd));
    }
    for (int i = 0; i < P.n; ++i) {
        const ExInstr &in = P.ins[i];
        const int32_t w[4] = {in.op, in.dt, in.a_src, in.b_src};
        mix(w, sizeof(w));
        if (jit_bakes_b(in)) mix(&in.lit_b, 8);
    }
    return h ? h : 1;
}

// ---- source generation
// Emits the steps of P (column slots already in the kernel's numbering: values c<k>[r], validity c<k>v[r] where col_has_valid[k])
// as one unrolled R-row loop per step; returns the names of the result's value array and validity expression.
// `in[r]`: the row takes part (and its column values count); lit_valid: the validity of a non-NULL literal.
std::pair<std::string, std::string> emit_steps(std::ostringstream &s, const ExProgram &P, const bool *col_has_valid, bool nulls, const std::string &prefix,
                                               const std::string &lit_array, const std::string &lit_valid) {
    std::vector<std::string> st, stv;
    auto hex = [](uint64_t v) {
        std::ostringstream o;
        o << "0x" << std::hex << v << "ull";
        return o.str();
    };
    for (int i = 0; i < P.n; ++i) {
        const ExInstr &in = P.ins[i];
        const bool a_st = in.a_src == EX_STACK, b_st = in.b_src == EX_STACK;
        std::string a, av, b, bv;
        auto operand = [&](int src, int which, std::string &w, std::string &v) {
            if (src >= EX_COL) {
                const int c = src - EX_COL;
                w = "c" + std::to_string(c) + "[r]";
                v = (nulls && col_has_valid[c]) ? "c" + std::to_string(c) + "v[r]" : "in[r]";
            } else {
                w = (which == 1 && jit_bakes_b(in)) ? hex(in.lit_b) : lit_array + "[" + std::to_string(2 * i + which) + "]";
                v = src == EX_LIT ? lit_valid : "false";
            }
        };
        if (a_st && b_st) {
            b = st.back() + "[r]"; bv = stv.back(); st.pop_back(); stv.pop_back();
            a = st.back() + "[r]"; av = stv.back(); st.pop_back(); stv.pop_back();
        } else if (a_st) {
            a = st.back() + "[r]"; av = stv.back(); st.pop_back(); stv.pop_back();
            operand(in.b_src, 1, b, bv);
        } else if (b_st) {
            b = st.back() + "[r]"; bv = stv.back(); st.pop_back(); stv.pop_back();
            operand(in.a_src, 0, a, av);
        } else {
            operand(in.a_src, 0, a, av);
            operand(in.b_src, 1, b, bv);
        }
        const std::string t = prefix + std::to_string(i), tv = t + "v";
        s << "    u64 " << t << "[R];";
        if (nulls) s << " bool " << tv << "[R];";
        s << "\n#pragma unroll\n    for (int r = 0; r < R; ++r) {\n      const u64 a = " << a << ", b = " << b << ";\n";
        if (nulls) s << "      const bool av = " << av << ", bv = " << bv << ";\n";
        const std::string ok = nulls ? "(av && bv)" : "in[r]";
        const int op = in.op, dt = in.dt;
        if (op == NQE_OP_AND || op == NQE_OP_OR) {
            if (nulls) { // and_kleene / or_kleene (ex_combine)
                s << "      const bool lb = av && a, rb = bv && b;\n";
                if (op == NQE_OP_AND) s << "      const bool ok = (av && bv) || (av && !lb) || (bv && !rb), res = ok && lb && rb;\n";
                else s << "      const bool ok = (av && bv) || lb || rb, res = ok && (lb || rb);\n";
                s << "      " << t << "[r] = res ? 1ull : 0ull; " << tv << "[r] = ok;\n";
            } else
                s << "      " << t << "[r] = a " << (op == NQE_OP_AND ? "&" : "|") << " b;\n";
        } else {
            if (nulls) s << "      " << tv << "[r] = av && bv;\n";
            if (op <= NQE_OP_GT_EQ) {
                static const char *cmp[6] = {"==", "!=", "<", "<=", ">", ">="}; // NQE_OP_EQ … NQE_OP_GT_EQ
                const char *c = cmp[op - NQE_OP_EQ];
                if (dt == NQE_INT64) s << "      " << t << "[r] = ((i64)a " << c << " (i64)b) ? 1ull : 0ull;\n";
                else if (dt == NQE_FLOAT64) s << "      " << t << "[r] = (u2d(a) " << c << " u2d(b)) ? 1ull : 0ull;\n";
                else s << "      " << t << "[r] = (a " << c << " b) ? 1ull : 0ull;\n";
            } else if (op == NQE_OP_PLUS || op == NQE_OP_MINUS || op == NQE_OP_MULTIPLY) {
                const char *c = op == NQE_OP_PLUS ? "+" : op == NQE_OP_MINUS ? "-" : "*";
                if (dt == NQE_FLOAT64) s << "      " << t << "[r] = d2u(u2d(a) " << c << " u2d(b));\n";
                else s << "      " << t << "[r] = a " << c << " b;\n";
            } else { // divide / modulus: a zero divisor (and MIN / -1) of a VALID row raises the flag, the word becomes 0
                const bool div = op == NQE_OP_DIVIDE;
                if (dt == NQE_FLOAT64) {
                    s << "      if (u2d(b) == 0.0) { if (" << ok << ") atomicOr(&A.flags[" << NQE_FLAG_DIV_ZERO << "], 1); " << t << "[r] = 0; }\n"
                      << "      else " << t << "[r] = d2u(" << (div ? "u2d(a) / u2d(b)" : "fmod(u2d(a), u2d(b))") << ");\n";
                } else {
                    s << "      if (b == 0) { if (" << ok << ") atomicOr(&A.flags[" << NQE_FLAG_DIV_ZERO << "], 1); " << t << "[r] = 0; }\n";
                    if (dt == NQE_INT64) {
                        s << "      else if ((i64)a == (i64)0x8000000000000000ull && (i64)b == -1) { if (" << ok << ") atomicOr(&A.flags[" << NQE_FLAG_OVERFLOW
                          << "], 1); " << t << "[r] = 0; }\n"
                          << "      else " << t << "[r] = (u64)((i64)a " << (div ? "/" : "%") << " (i64)b);\n";
                    } else
                        s << "      else " << t << "[r] = a " << (div ? "/" : "%") << " b;\n";
                }
            }
        }
        s << "    }\n";
        st.push_back(t);
        stv.push_back(nulls ? tv + "[r]" : "in[r]");
    }
    return {st.back(), stv.back()};
}

std::string gen_source(const ExProgram &P, bool nulls, bool bool_out) {
    std::ostringstream s;
    s << "#pragma clang fp contract(off)\n" // (hipRTC's own -ffp-contract=fast-honor-pragmas comes after the caller's options)
      << "typedef unsigned long long u64; typedef long long i64; typedef unsigned int u32;\n"
      << "#define R " << EX_ROWS << "\n"
      << "struct Args { const void *col[" << EX_MAX_COLS << "]; const unsigned char *valid[" << EX_MAX_COLS << "]; u64 lit[" << JIT_MAX_LITS
      << "]; i64 n; u64 *out_words, *out_bits, *out_valid; int *flags; };\n"
      << "static __device__ __forceinline__ double u2d(u64 w) { return __longlong_as_double((i64)w); }\n"
      << "static __device__ __forceinline__ u64 d2u(double d) { return (u64)__double_as_longlong(d); }\n"
      << "extern \"C\" __global__ void __launch_bounds__(256) nqe_jit_expr(Args A) {\n"
      << "  const int lane = threadIdx.x & 63;\n"
      << "  const i64 n = A.n, n_chunks = (n + 64 * R - 1) / (64 * R);\n"
      << "  const i64 wave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((i64)gridDim.x * blockDim.x) >> 6;\n"
      << "  for (i64 chunk = wave; chunk < n_chunks; chunk += n_waves) {\n"
      << "    const i64 row0 = chunk * (64 * R) + lane;\n"
      << "    i64 rc[R]; bool in[R];\n"
      << "#pragma unroll\n"
      << "    for (int r = 0; r < R; ++r) { const i64 row = row0 + r * 64; in[r] = row < n; rc[r] = row < n - 1 ? row : n - 1; }\n";
    // every load of the chunk first
    for (int c = 0; c < P.ncols; ++c) {
        s << "    u64 c" << c << "[R];\n#pragma unroll\n    for (int r = 0; r < R; ++r) ";
        if (P.col_dtype[c] == NQE_BOOLEAN) s << "c" << c << "[r] = ((const unsigned char *)A.col[" << c << "])[rc[r] >> 3];\n";
        else s << "c" << c << "[r] = __builtin_nontemporal_load((const u64 *)A.col[" << c << "] + rc[r]);\n";
        if (nulls && P.col_valid[c])
            s << "    u32 vb" << c << "[R];\n#pragma unroll\n    for (int r = 0; r < R; ++r) vb" << c << "[r] = A.valid[" << c << "][rc[r] >> 3];\n";
    }
    for (int c = 0; c < P.ncols; ++c) {
        if (P.col_dtype[c] == NQE_BOOLEAN) s << "#pragma unroll\n    for (int r = 0; r < R; ++r) c" << c << "[r] = (c" << c << "[r] >> ((int)rc[r] & 7)) & 1ull;\n";
        if (nulls && P.col_valid[c])
            s << "    bool c" << c << "v[R];\n#pragma unroll\n    for (int r = 0; r < R; ++r) c" << c << "v[r] = in[r] && ((vb" << c << "[r] >> ((int)rc[r] & 7)) & 1u);\n";
    }
    bool has_valid[EX_MAX_COLS];
    for (int c = 0; c < EX_MAX_COLS; ++c) has_valid[c] = c < P.ncols && P.col_valid[c] != nullptr;
    const auto result = emit_steps(s, P, has_valid, nulls, "t", "A.lit", "in[r]");
    const std::string res = result.first, resv = result.second;
    s << "#pragma unroll\n    for (int r = 0; r < R; ++r) {\n"
      << "      const i64 row = row0 + r * 64;\n"
      << "      if (row - lane >= n) break;\n"
      << "      const bool ok = " << resv << ";\n";
    if (!bool_out) s << "      if (row < n) __builtin_nontemporal_store(ok ? " << res << "[r] : 0ull, A.out_words + row);\n";
    else s << "      { const u64 w = __ballot(ok && " << res << "[r]); if (lane == 0) A.out_bits[row >> 6] = w; }\n";
    s << "      if (A.out_valid) { const u64 w = __ballot(ok); if (lane == 0) A.out_valid[row >> 6] = w; }\n"
      << "    }\n  }\n}\n";
    return s.str();
}

// ---- code objects on disk: a NEW process finds the kernels an earlier one compiled (hipRTC takes 0.3-3 s per tree shape, during
// which a one-shot process would only ever interpret).  One file per (source hash, ISA, HIP runtime version) under NQE_JIT_CACHE_DIR
// (default $XDG_CACHE_HOME/nqe_jit, $HOME/.cache/nqe_jit, /tmp/nqe_jit-<uid>): magic, the generated SOURCE (compared on load: a
// hash collision or a changed generator never hands out another tree's kernel), a checksum of the code object, and the code object;
// written to a temporary name and renamed.  NQE_NO_JIT_DISK_CACHE=1 switches it off.  Failures of any kind just mean "compile".
// The directory is trusted only when it is a real directory (not a symlink) OWNED BY THIS USER with no group / other permissions
// (the /tmp fallback has a predictable name another local user could create first and fill with crafted code objects);
// files are opened O_NOFOLLOW and must be regular files of this user; a file whose checksum or load fails is deleted and recompiled.
inline uint64_t jit_checksum(const char *p, size_t n) { // FNV-1a over 8-byte words (+ the tail bytes)
    uint64_t h = 1469598103934665603ull;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        std::memcpy(&w, p + i, 8);
        h = (h ^ w) * 1099511628211ull;
    }
    for (; i < n; ++i) h = (h ^ uint64_t((unsigned char)p[i])) * 1099511628211ull;
    return h;
}
std::string jit_disk_path(uint64_t key, const std::string &arch) {
    if (getenv("NQE_NO_JIT_DISK_CACHE")) return std::string();
    std::string dir;
    if (const char *d = getenv("NQE_JIT_CACHE_DIR")) dir = d;
    else if (const char *x = getenv("XDG_CACHE_HOME")) dir = std::string(x) + "/nqe_jit";
    else if (const char *h = getenv("HOME")) dir = std::string(h) + "/.cache/nqe_jit";
    else dir = "/tmp/nqe_jit-" + std::to_string((unsigned long)getuid());
    while (dir.size() > 1 && dir.back() == '/') dir.pop_back();
    for (size_t i = 1; i <= dir.size(); ++i) // mkdir -p
        if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), 0700);
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & 077) != 0) return std::string(); // not ours alone: no disk cache
    int rtv = 0;
    (void)hipRuntimeGetVersion(&rtv);
    char name[96];
    snprintf(name, sizeof(name), "/%016llx-%s-%d.nqejit", (unsigned long long)key, arch.c_str(), rtv);
    return dir + name;
}
constexpr uint64_t JIT_DISK_MAGIC = 0x3230544a5145514eull; // "NQEQJT02" (02: + checksum)
bool jit_disk_load(const std::string &path, const std::string &source, std::vector<char> *code) {
    if (path.empty()) return false;
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    FILE *f = (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == getuid()) ? fdopen(fd, "rb") : nullptr;
    if (!f) {
        close(fd);
        return false;
    }
    bool ok = false, mine = false;
    uint64_t head[4];
    if (fread(head, 8, 4, f) == 4 && head[0] == JIT_DISK_MAGIC && head[1] == source.size() && head[2] > 0 && head[2] < (uint64_t(1) << 28)) {
        std::string src(size_t(head[1]), '\0');
        code->resize(size_t(head[2]));
        mine = fread(&src[0], 1, src.size(), f) == src.size() && src == source; // (another tree under the same hash: not this entry's file to judge)
        ok = mine && fread(code->data(), 1, code->size(), f) == code->size() && jit_checksum(code->data(), code->size()) == head[3];
    }
    fclose(f);
    if (!ok) {
        code->clear();
        if (mine) (void)unlink(path.c_str()); // truncated or corrupt: the next store replaces it
    }
    return ok;
}
void jit_disk_store(const std::string &path, const std::string &source, const std::vector<char> &code) {
    if (path.empty() || code.empty()) return;
    const std::string tmp = path + ".tmp" + std::to_string((unsigned long)getpid());
    (void)unlink(tmp.c_str());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    FILE *f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
    if (!f) {
        if (fd >= 0) close(fd);
        return;
    }
    const uint64_t head[4] = {JIT_DISK_MAGIC, source.size(), code.size(), jit_checksum(code.data(), code.size())};
    const bool ok = fwrite(head, 8, 4, f) == 4 && fwrite(source.data(), 1, source.size(), f) == source.size() && fwrite(code.data(), 1, code.size(), f) == code.size();
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
}

void jit_compile(JitEntry *e) {
    const HipRtcApi &rt = hiprtc_api();
    if (!rt.ok) {
        e->log = "libhiprtc.so not available";
        e->state.store(-1, std::memory_order_release);
        return;
    }
    hiprtcProgram prog = nullptr;
    bool good = rt.create(&prog, e->source.c_str(), "nqe_jit_expr.hip", 0, nullptr, nullptr) == HIPRTC_SUCCESS;
    if (good) {
        // -ffp-contract=off: `v * v + w` is two roundings in arrow (and in the interpreter, whose steps are separate), never an fma
        const std::string target = "--offload-arch=" + e->arch; // the context's device (this library ships gfx950 code objects only, but
                                                                 // a run-time kernel for another target would never load)
        const char *opts[] = {target.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", e->extra_opt.c_str()};
        good = rt.compile(prog, e->extra_opt.empty() ? 4 : 5, opts) == HIPRTC_SUCCESS;
        size_t ls = 0;
        if (rt.log_size(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
            e->log.resize(ls);
            (void)rt.log(prog, &e->log[0]);
        }
        size_t cs = 0;
        if (good) good = rt.code_size(prog, &cs) == HIPRTC_SUCCESS && cs > 0;
        if (good) {
            e->code.resize(cs);
            good = rt.code(prog, e->code.data()) == HIPRTC_SUCCESS;
        }
    }
    if (prog) (void)rt.destroy(&prog);
    if (good) jit_disk_store(e->disk_path, e->source, e->code);
    e->state.store(good ? 1 : -1, std::memory_order_release);
}

// The specialised kernel of P, if it is ready; starts its compilation otherwise (null: use the interpreter this time).
template <class MakeSource> JitEntry *jit_get(nqe_ctx *ctx, uint64_t key, const char *kernel_name, MakeSource &&make_source, const char *extra_opt = nullptr) {
    JitCache *cache = jit_cache(ctx);
    if (cache->unavailable) return nullptr; // a code object failed to load on this device: no further compilations
    auto it = cache->entries.find(key);
    if (it != cache->entries.end() && it->second->source != make_source()) return nullptr; // hash collision: interpret, never launch another tree's kernel
    if (it == cache->entries.end()) {
        if (cache->entries.size() >= 128) return nullptr; // (entries hold threads and modules: no eviction, just stop specialising)
        auto e = std::make_unique<JitEntry>();
        e->source = make_source();
        e->arch = ctx->arch;
        if (extra_opt) e->extra_opt = extra_opt;
        JitEntry *raw = e.get();
        if (const char *dump = getenv("NQE_JIT_DUMP")) { // diagnostics: the generated source, appended to this file
            if (FILE *f = fopen(dump, "a")) {
                fprintf(f, "// ---- %016llx\n%s\n", (unsigned long long)key, e->source.c_str());
                fclose(f);
            }
        }
        e->disk_path = jit_disk_path(key, e->arch);
        if (jit_disk_load(e->disk_path, e->source, &e->code)) {
            e->from_disk = true;
            e->state.store(1, std::memory_order_release); // an earlier process compiled this very source: loaded below, used by THIS execution
        } else {
            e->worker = std::thread(jit_compile, raw);
            if (getenv("NQE_JIT_SYNC")) raw->worker.join(); // tests / benchmarks: compile before the first execution
        }
        it = cache->entries.emplace(key, std::move(e)).first;
    }
    JitEntry *e = it->second.get();
    int st = e->state.load(std::memory_order_acquire);
    if (st == 1) { // code ready: load it on this thread (the only one that makes HIP calls for the context)
        if (e->worker.joinable()) e->worker.join();
        const bool ok = hipModuleLoadData(&e->mod, e->code.data()) == hipSuccess && hipModuleGetFunction(&e->fn, e->mod, kernel_name) == hipSuccess;
        e->code.clear();
        e->code.shrink_to_fit();
        if (!ok && e->from_disk) {
            // a code object from DISK that does not load (another driver's leftovers, a damaged file that kept its checksum): delete it and
            // compile — interpreting meanwhile — instead of switching specialisation off for the whole context and for every later process
            (void)hipGetLastError();
            if (e->mod) (void)hipModuleUnload(e->mod);
            e->mod = nullptr;
            e->fn = nullptr;
            e->from_disk = false;
            (void)unlink(e->disk_path.c_str());
            e->state.store(0, std::memory_order_release);
            e->worker = std::thread(jit_compile, e);
            if (getenv("NQE_JIT_SYNC")) e->worker.join();
            return nullptr;
        }
        if (!ok) {
            (void)hipGetLastError();
            e->log = "hipModuleLoadData failed";
            cache->unavailable = true;
        }
        st = ok ? 2 : -1;
        e->state.store(st, std::memory_order_release);
    }
    if (st == -1 && getenv("NQE_DEBUG") && !e->log.empty()) {
        fprintf(stderr, "[nqe] expression specialisation failed: %s\n", e->log.c_str());
        e->log.clear();
    }
    return st == 2 ? e : nullptr;
}

// Runs P specialised when that pays and the kernel is ready.  Returns false when the caller should interpret.
bool jit_expr_tree(nqe_ctx *ctx, const ExProgram &P, bool nulls, int64_t rows, uint64_t *ow, uint64_t *ob, uint64_t *ov) {
    // (read per call, not once per process: tests switch them around single calls)
    const bool off = getenv("NQE_NO_JIT") != nullptr; // diagnostics (A/B)
    const char *mr = getenv("NQE_JIT_MIN_ROWS");
    const int64_t min_rows = mr ? atoll(mr) : (int64_t(1) << 22);
    if (off || P.n < 3 || rows < min_rows) return false; // one or two steps run at the memory system's rate interpreted
    const bool bool_out = ob != nullptr;
    JitEntry *e = jit_get(ctx, jit_hash(P, nulls, bool_out), "nqe_jit_expr", [&] { return gen_source(P, nulls, bool_out); });
    if (!e) return false;
    JitArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int c = 0; c < P.ncols; ++c) {
        a.col[c] = P.col_values[c];
        a.valid[c] = P.col_valid[c];
    }
    for (int i = 0; i < P.n; ++i) {
        a.lit[2 * i] = P.ins[i].lit_a;
        a.lit[2 * i + 1] = P.ins[i].lit_b;
    }
    a.n = rows;
    a.out_words = ow;
    a.out_bits = ob;
    a.out_valid = ov;
    a.flags = ctx->d_flags;
    void *params[] = {&a};
    const unsigned grid = unsigned(stream_grid(ctx, (rows + EX_ROWS - 1) / EX_ROWS, 256));
    TimerScope t(ctx, "expr_jit");
    ctx->flags_clean = false;
    NQE_HIP_CHECK(hipModuleLaunchKernel(e->fn, grid, 1, 1, 256, 1, 1, 0, ctx->stream, params, nullptr));
    return true;
}


// ---- the projection behind a selection, every output in ONE pass over the kept rows
// (expr_tree_compact_kernel's job — one wave per 4096-row tile of the keep bitmap, results written straight to their compacted
// positions, a NULL predicate emits a NULL row, dropped rows never evaluate — for the whole projection list at once: each
// referenced column is read once whatever the number of outputs that use it; bare columns are outputs like any other)
constexpr int JP_MAX_COLS = 8, JP_MAX_OUTS = 4;
struct JitProjOut {
    bool is_column = false; // a bare column: slot `col`
    int col = 0;
    bool bool_out = false, needs_valid = false;
    int out_dtype = NQE_INT64;
    ExProgram P; // column operands renumbered to the projection's slots (col_values etc. unused)
};
struct JitProj {
    int ncols = 0;
    const void *col_values[JP_MAX_COLS];
    const uint8_t *col_valid[JP_MAX_COLS];
    int32_t col_dtype[JP_MAX_COLS];
    std::vector<JitProjOut> outs;
    bool nulls = false; // some column carries validity, some literal is NULL, or the predicate was NULL on some row
};
struct JitProjArgs {
    const void *col[JP_MAX_COLS];
    const uint8_t *valid[JP_MAX_COLS];
    uint64_t lit[JP_MAX_OUTS][JIT_MAX_LITS];
    const uint64_t *keep, *pvalid, *tile_offsets;
    int64_t n, ntiles;
    uint64_t *out_words[JP_MAX_OUTS];
    uint8_t *out_bool[JP_MAX_OUTS], *out_valid[JP_MAX_OUTS];
    int *flags;
};

uint64_t jit_hash_proj(const JitProj &J) {
    uint64_t h = 1469598103934665603ull ^ 0x70726f6aull;
    auto mix = [&](const void *p, size_t nb) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < nb; ++i) h = (h ^ b[i]) * 1099511628211ull;
    };
    const int32_t head[3] = {J.ncols, int32_t(J.outs.size()), J.nulls ? 1 : 0};
    mix(head, sizeof(head));
    for (int c = 0; c < J.ncols; ++c) {
        const int32_t cd[2] = {J.col_dtype[c], (J.nulls && J.col_valid[c]) ? 1 : 0};
        mix(cd, sizeof(cd));
    }
    for (const JitProjOut &o : J.outs) {
        const int32_t od[5] = {o.is_column ? 1 : 0, o.col, o.bool_out ? 1 : 0, o.needs_valid ? 1 : 0, o.is_column ? 0 : o.P.n};
        mix(od, sizeof(od));
        if (o.is_column) continue;
        for (int i = 0; i < o.P.n; ++i) {
            const ExInstr &in = o.P.ins[i];
            const int32_t w[4] = {in.op, in.dt, in.a_src, in.b_src};
            mix(w, sizeof(w));
            if (jit_bakes_b(in)) mix(&in.lit_b, 8);
        }
    }
    return h ? h : 1;
}

std::string gen_source_proj(const JitProj &J) {
    std::ostringstream s;
    const bool nulls = J.nulls;
    s << "#pragma clang fp contract(off)\n"
      << "typedef unsigned long long u64; typedef long long i64; typedef unsigned int u32;\n"
      << "#define R " << EX_ROWS << "\n"
      << "struct Args { const void *col[" << JP_MAX_COLS << "]; const unsigned char *valid[" << JP_MAX_COLS << "]; u64 lit[" << JP_MAX_OUTS << "][" << JIT_MAX_LITS
      << "]; const u64 *keep, *pvalid, *tile_offsets; i64 n, ntiles; u64 *out_words[" << JP_MAX_OUTS << "]; unsigned char *out_bool[" << JP_MAX_OUTS
      << "], *out_valid[" << JP_MAX_OUTS << "]; int *flags; };\n"
      << "static __device__ __forceinline__ double u2d(u64 w) { return __longlong_as_double((i64)w); }\n"
      << "static __device__ __forceinline__ u64 d2u(double d) { return (u64)__double_as_longlong(d); }\n"
      << "extern \"C\" __global__ void __launch_bounds__(256) nqe_jit_proj(Args A) {\n"
      << "  const int lane = threadIdx.x & 63, wpb = blockDim.x / 64;\n"
      << "  const i64 n = A.n, nwords = (n + 63) / 64;\n"
      << "  const u64 lt = (1ull << lane) - 1ull;\n"
      << "  for (i64 tile = (i64)blockIdx.x * wpb + threadIdx.x / 64; tile < A.ntiles; tile += (i64)gridDim.x * wpb) {\n"
      << "    const i64 w = tile * 64 + lane;\n"
      << "    const u64 my_word = w < nwords ? A.keep[w] : 0ull;\n"
      << "    const u64 my_pv = (A.pvalid && w < nwords) ? A.pvalid[w] : ~0ull;\n"
      << "    const u32 cnt = (u32)__popcll(my_word);\n"
      << "    u32 off = cnt;\n"
      << "    for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(off, d, 64); if (lane >= d) off += t; }\n"
      << "    const u32 tot = __shfl(off, 63, 64);\n"
      << "    off -= cnt;\n"
      << "    if (tot == 0) continue;\n"
      << "    const u64 base = A.tile_offsets[tile];\n"
      << "    for (int k0 = 0; k0 < 64; k0 += R) {\n"
      << "    u64 kw[R]; u32 po[R]; bool in[R], emit[R]; bool anyk = false;\n"
      << "#pragma unroll\n"
      << "    for (int r = 0; r < R; ++r) {\n"
      << "      kw[r] = __shfl(my_word, k0 + r, 64);\n"
      << "      const u64 pv = __shfl(my_pv, k0 + r, 64);\n"
      << "      anyk = anyk || kw[r] != 0;\n"
      << "      emit[r] = (kw[r] >> lane) & 1ull;\n"          // every emitted row: literals are valid there
      << "      in[r] = ((kw[r] & pv) >> lane) & 1ull;\n"      // … whose predicate was valid: column values count
      << "      po[r] = __shfl(off, k0 + r, 64) + (u32)__popcll(kw[r] & lt);\n" // position inside the tile's output range (all lanes active here)
      << "    }\n"
      << "    if (!anyk) continue;\n"
      << "    const i64 row0 = (tile * 64 + k0) * 64 + lane;\n"
      << "    i64 rc[R];\n"
      << "#pragma unroll\n"
      << "    for (int r = 0; r < R; ++r) { const i64 row = row0 + r * 64; rc[r] = row < n - 1 ? row : n - 1; }\n";
    bool has_valid[JP_MAX_COLS];
    for (int c = 0; c < JP_MAX_COLS; ++c) has_valid[c] = c < J.ncols && nulls && J.col_valid[c] != nullptr;
    for (int c = 0; c < J.ncols; ++c) {
        s << "    u64 c" << c << "[R];\n#pragma unroll\n    for (int r = 0; r < R; ++r) ";
        if (J.col_dtype[c] == NQE_BOOLEAN) s << "c" << c << "[r] = ((const unsigned char *)A.col[" << c << "])[rc[r] >> 3];\n";
        else s << "c" << c << "[r] = __builtin_nontemporal_load((const u64 *)A.col[" << c << "] + rc[r]);\n";
        if (has_valid[c]) s << "    u32 vb" << c << "[R];\n#pragma unroll\n    for (int r = 0; r < R; ++r) vb" << c << "[r] = A.valid[" << c << "][rc[r] >> 3];\n";
    }
    for (int c = 0; c < J.ncols; ++c) {
        if (J.col_dtype[c] == NQE_BOOLEAN) s << "#pragma unroll\n    for (int r = 0; r < R; ++r) c" << c << "[r] = (c" << c << "[r] >> ((int)rc[r] & 7)) & 1ull;\n";
        if (has_valid[c])
            s << "    bool c" << c << "v[R];\n#pragma unroll\n    for (int r = 0; r < R; ++r) c" << c << "v[r] = in[r] && ((vb" << c << "[r] >> ((int)rc[r] & 7)) & 1u);\n";
    }
    std::vector<std::pair<std::string, std::string>> results;
    for (size_t o = 0; o < J.outs.size(); ++o) {
        const JitProjOut &out = J.outs[o];
        if (out.is_column) results.push_back({"c" + std::to_string(out.col), has_valid[out.col] ? "c" + std::to_string(out.col) + "v[r]" : "in[r]"});
        else results.push_back(emit_steps(s, out.P, has_valid, nulls, "t" + std::to_string(o) + "_", "A.lit[" + std::to_string(o) + "]", "emit[r]"));
    }
    s << "#pragma unroll\n    for (int r = 0; r < R; ++r) {\n"
      << "      if (!emit[r]) continue;\n"
      << "      const u64 pos = base + po[r];\n";
    std::string body;
    for (size_t o = 0; o < J.outs.size(); ++o) {
        const JitProjOut &out = J.outs[o];
        std::ostringstream b;
        b << "      { const bool ok = " << results[o].second << "; const u64 v = " << results[o].first << "[r];\n";
        if (!out.bool_out) b << "        __builtin_nontemporal_store(ok ? v : 0ull, A.out_words[" << o << "] + pos);\n";
        else b << "        A.out_bool[" << o << "][pos] = (ok && v) ? 1 : 0;\n";
        if (out.needs_valid) b << "        A.out_valid[" << o << "][pos] = ok ? 1 : 0;\n";
        b << "      }\n";
        body += b.str();
    }
    s << body << "    }\n    }\n  }\n}\n";
    return s.str();
}

// launches the fused projection when its kernel is ready; false: the caller takes the per-expression path
bool jit_project(nqe_ctx *ctx, const JitProj &J, const KeepMask &km, uint64_t *const *out_words, uint8_t *const *out_bool, uint8_t *const *out_valid) {
    JitEntry *e = jit_get(ctx, jit_hash_proj(J), "nqe_jit_proj", [&] { return gen_source_proj(J); });
    if (!e) return false;
    JitProjArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int c = 0; c < J.ncols; ++c) {
        a.col[c] = J.col_values[c];
        a.valid[c] = J.col_valid[c];
    }
    for (size_t o = 0; o < J.outs.size(); ++o) {
        if (!J.outs[o].is_column)
            for (int i = 0; i < J.outs[o].P.n; ++i) {
                a.lit[o][2 * i] = J.outs[o].P.ins[i].lit_a;
                a.lit[o][2 * i + 1] = J.outs[o].P.ins[i].lit_b;
            }
        a.out_words[o] = out_words[o];
        a.out_bool[o] = out_bool[o];
        a.out_valid[o] = out_valid[o];
    }
    a.keep = (const uint64_t *)km.keep->ptr;
    a.pvalid = km.pvalid ? (const uint64_t *)km.pvalid->ptr : nullptr;
    a.tile_offsets = (const uint64_t *)km.tile_offsets->ptr;
    a.n = km.n;
    a.ntiles = km.ntiles;
    a.flags = ctx->d_flags;
    void *params[] = {&a};
    const unsigned grid = unsigned(stream_grid(ctx, km.ntiles, 4));
    TimerScope t(ctx, "proj_jit");
    ctx->flags_clean = false;
    NQE_HIP_CHECK(hipModuleLaunchKernel(e->fn, grid, 1, 1, 256, 1, 1, 0, ctx->stream, params, nullptr));
    return true;
}

// ---- selection + projection in ONE pass over the table (a tree predicate, a projection list, inputs without NULLs)
// The two-kernel form reads the predicate's columns (mask kernel), then — behind a scan and a host wait for the row count — the
// projection's columns for the kept rows: a column both sides use is read twice (`select v * v + v / 4, id … where (id + 1) % 10 < 5`
// moved 1.37x its algorithmic bytes).  Here every wave takes 512-row chunks in ticket order (one ticket per 16-wave workgroup and step), evaluates the predicate on its rows,
// counts the kept ones with ballots, publishes the count and obtains the number of rows kept before its chunk by decoupled
// look-back over the chunks' status words (flag in the top two bits: 1 = this chunk's count, 2 = the inclusive prefix; a wave
// inspects 64 predecessors per step; tickets are drawn in order, so every predecessor is already running), then evaluates the
// projection list on the kept rows and writes them at prefix + position: stable order, each column read once, no mask, no scan
// kernel.  The output columns are allocated for the worst case (every row kept); the last chunk's inclusive prefix is the row
// count the host reads back.  Columns only the projection uses are loaded after the count is known to be non-zero.
// rows per lane per chunk, threads per workgroup (twelve chunks = 6144 rows per ticket and status word; 74 VGPRs: two such workgroups
// per CU).  Other shapes measured (profiles/r04, r05) — per 10^8 rows of `select v * v + v / 4, id … where (id + 1) % 10 < 5`: 256x8 0.78 ms,
// 512x4 0.76, 512x8 0.60, 512x16 0.60, 1024x8 0.60, 768x8 0.56 (the two-kernel form: 0.67 + its scan and host wait)
constexpr int SP_R = 8, SP_BLOCK = 768;
struct JitSelProj {
    JitProj proj;       // the union of the predicate's and the projection's columns, the outputs
    ExProgram pred;     // column operands renumbered to proj's slots
    uint32_t pred_cols = 0, proj_cols = 0; // slot bit masks: read by the predicate / by some output
};
struct JitSelProjArgs {
    const void *col[JP_MAX_COLS];
    uint64_t lit[JP_MAX_OUTS][JIT_MAX_LITS];
    uint64_t plit[JIT_MAX_LITS];
    int64_t n, n_steps;
    uint64_t *out_words[JP_MAX_OUTS];
    unsigned long long *status; // [n_steps], zeroed
    uint32_t *ticket;           // zeroed
    unsigned long long *total;  // rows kept
    int *flags;
};

uint64_t jit_hash_selproj(const JitSelProj &S) {
    uint64_t h = jit_hash_proj(S.proj) ^ 0x73656c70726f6aull;
    auto mix = [&](const void *p, size_t nb) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < nb; ++i) h = (h ^ b[i]) * 1099511628211ull;
    };
    const int32_t head[5] = {S.pred.n, int32_t(S.pred_cols), int32_t(S.proj_cols), SP_R, SP_BLOCK};
    mix(head, sizeof(head));
    for (int i = 0; i < S.pred.n; ++i) {
        const ExInstr &in = S.pred.ins[i];
        const int32_t w[4] = {in.op, in.dt, in.a_src, in.b_src};
        mix(w, sizeof(w));
        if (jit_bakes_b(in)) mix(&in.lit_b, 8);
    }
    return h ? h : 1;
}

std::string gen_source_selproj(const JitSelProj &S) {
    const JitProj &J = S.proj;
    std::ostringstream s;
    s << "#pragma clang fp contract(off)\n"
      << "typedef unsigned long long u64; typedef long long i64; typedef unsigned int u32;\n"
      << "#define R " << SP_R << "\n"
      << "struct Args { const void *col[" << JP_MAX_COLS << "]; u64 lit[" << JP_MAX_OUTS << "][" << JIT_MAX_LITS << "]; u64 plit[" << JIT_MAX_LITS
      << "]; i64 n, n_steps; u64 *out_words[" << JP_MAX_OUTS << "]; u64 *status; u32 *ticket; u64 *total; int *flags; };\n"
      << "static __device__ __forceinline__ double u2d(u64 w) { return __longlong_as_double((i64)w); }\n"
      << "static __device__ __forceinline__ u64 d2u(double d) { return (u64)__double_as_longlong(d); }\n"
      << "static __device__ __forceinline__ u64 wave_sum(u64 v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64); return v; }\n"
      << "static __device__ __forceinline__ u64 st_load(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }\n"
      << "static __device__ __forceinline__ void st_store(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }\n"
      << "extern \"C\" __global__ void __launch_bounds__(" << SP_BLOCK << ") nqe_jit_selproj(Args A) {\n"
      << "  __shared__ u32 s_base; __shared__ u32 s_cnt[" << SP_BLOCK / 64 << "]; __shared__ u64 s_excl;\n"
      << "  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;\n"
      << "  const i64 n = A.n;\n"
      << "  const u64 lt = (1ull << lane) - 1ull, VAL = (1ull << 62) - 1ull;\n"
      // one ticket per WORKGROUP per step (its waves take consecutive chunks): a ticket per wave was 195 000 atomics on one word
      // for 10^8 rows — at ~12 ns each they alone took 2.4 ms.  (Drawing the NEXT step's ticket while this step runs hid the
      // atomic's latency and lengthened the look-back chains — a workgroup then holds a step it has not started: 0.60 -> 0.82 ms.)
      << "  for (;;) {\n"
      << "    if (threadIdx.x == 0) s_base = atomicAdd(A.ticket, 1u);\n"
      << "    __syncthreads();\n"
      << "    const i64 step = (i64)s_base;\n"
      << "    if (step >= A.n_steps) break;\n"
      << "    const i64 chunk = step * wpb + wave;\n" // (a chunk past the table: no row exists, nothing kept — it still takes part in the barriers)
      << "    const i64 row0 = chunk * (64 * R) + lane;\n"
      << "    i64 rc[R]; bool in[R];\n"
      << "#pragma unroll\n"
      << "    for (int r = 0; r < R; ++r) { const i64 row = row0 + r * 64; in[r] = row < n; rc[r] = row < n - 1 ? row : n - 1; }\n";
    auto load_col = [&](int c) {
        s << "    u64 c" << c << "[R];\n#pragma unroll\n    for (int r = 0; r < R; ++r) c" << c << "[r] = __builtin_nontemporal_load((const u64 *)A.col[" << c << "] + rc[r]);\n";
    };
    bool has_valid[JP_MAX_COLS];
    for (int c = 0; c < JP_MAX_COLS; ++c) has_valid[c] = false;
    for (int c = 0; c < J.ncols; ++c)
        if (S.pred_cols & (1u << c)) load_col(c);
    const auto pres = emit_steps(s, S.pred, has_valid, false, "p", "A.plit", "in[r]");
    s << "    u64 kw[R]; u32 po[R]; u32 cnt = 0;\n"
      << "#pragma unroll\n"
      << "    for (int r = 0; r < R; ++r) { kw[r] = __ballot(in[r] && " << pres.first << "[r] != 0); po[r] = cnt + (u32)__popcll(kw[r] & lt); cnt += (u32)__popcll(kw[r]); }\n"
      << "    if (lane == 0) s_cnt[wave] = cnt;\n";
    // the projection's own columns: requested before the look-back, so that their latency overlaps it — and only when rows were
    // kept (cnt is wave-uniform: a chunk that keeps nothing never touches them)
    for (int c = 0; c < J.ncols; ++c)
        if (!(S.pred_cols & (1u << c)))
            s << "    u64 c" << c << "[R];\n#pragma unroll\n    for (int r = 0; r < R; ++r) c" << c << "[r] = cnt ? __builtin_nontemporal_load((const u64 *)A.col[" << c
              << "] + rc[r]) : 0ull;\n";
    // The workgroup's kept rows are summed through LDS and ONE wave looks back over the statuses of the steps before this one (a
    // status per 512-row chunk had every wave walk back through the ~4000 chunks in flight, 64 per memory round trip: 0.94 ms per
    // 10^8 rows against 0.72 for the two kernels); the other waves wait at the barrier with their projection loads in flight.
    s << "    __syncthreads();\n"
      << "    if (wave == 0) {\n"
      << "      const u32 mine = lane < wpb ? s_cnt[lane] : 0u;\n"
      << "      const u64 tot = wave_sum((u64)mine);\n"
      << "      u64 ex = 0;\n"
      << "      if (step > 0) {\n"
      << "        if (lane == 0) st_store(A.status + step, (1ull << 62) | tot);\n"
      << "        i64 look = step - 1;\n"
      << "        for (;;) {\n"
      << "          const i64 idx = look - lane;\n"
      << "          u64 sv = idx >= 0 ? st_load(A.status + idx) : (2ull << 62);\n" // (before step 0: an inclusive prefix of zero rows)
      << "          while (__ballot((sv >> 62) == 0ull) != 0ull) { __builtin_amdgcn_s_sleep(1); if (idx >= 0) sv = st_load(A.status + idx); }\n"
      << "          const u64 pm = __ballot((sv >> 62) == 2ull);\n"
      << "          if (pm != 0ull) {\n"
      << "            const int nearest = __ffsll((i64)pm) - 1;\n"
      << "            ex += wave_sum(lane <= nearest ? (sv & VAL) : 0ull);\n"
      << "            break;\n"
      << "          }\n"
      << "          ex += wave_sum(sv & VAL);\n"
      << "          look -= 64;\n"
      << "        }\n"
      << "      }\n"
      << "      if (lane == 0) { st_store(A.status + step, (2ull << 62) | (ex + tot)); s_excl = ex; if (step == A.n_steps - 1) *A.total = ex + tot; }\n"
      << "    }\n"
      << "    __syncthreads();\n"
      << "    u64 excl = s_excl;\n"
      << "    for (int w = 0; w < wave; ++w) excl += s_cnt[w];\n"
      << "    if (cnt == 0) continue;\n"
      << "    bool keep[R];\n"
      << "#pragma unroll\n"
      << "    for (int r = 0; r < R; ++r) keep[r] = (kw[r] >> lane) & 1ull;\n"
      << "    {\n"
      << "    bool in[R];\n" // the projection's rows: the kept ones (a dropped row never evaluates: no DivideByZero from it)
      << "#pragma unroll\n"
      << "    for (int r = 0; r < R; ++r) in[r] = keep[r];\n";
    std::vector<std::pair<std::string, std::string>> results;
    for (size_t o = 0; o < J.outs.size(); ++o) {
        const JitProjOut &out = J.outs[o];
        if (out.is_column) results.push_back({"c" + std::to_string(out.col), "in[r]"});
        else results.push_back(emit_steps(s, out.P, has_valid, false, "t" + std::to_string(o) + "_", "A.lit[" + std::to_string(o) + "]", "in[r]"));
    }
    s << "#pragma unroll\n    for (int r = 0; r < R; ++r) {\n"
      << "      if (!in[r]) continue;\n"
      << "      const u64 pos = excl + po[r];\n";
    for (size_t o = 0; o < J.outs.size(); ++o) s << "      __builtin_nontemporal_store(" << results[o].first << "[r], A.out_words[" << o << "] + pos);\n";
    s << "    }\n    }\n  }\n}\n";
    return s.str();
}

// the one-pass selection + projection kernel of S when it is ready (null: the caller takes the mask + compaction path this time)
JitEntry *jit_select_project_entry(nqe_ctx *ctx, const JitSelProj &S) {
    return jit_get(ctx, jit_hash_selproj(S), "nqe_jit_selproj", [&] { return gen_source_selproj(S); });
}
void jit_select_project(nqe_ctx *ctx, JitEntry *e, const JitSelProj &S, int64_t n, uint64_t *const *out_words, unsigned long long *status, uint32_t *ticket,
                        unsigned long long *total) {
    JitSelProjArgs a;
    std::memset(&a, 0, sizeof(a));
    const JitProj &J = S.proj;
    for (int c = 0; c < J.ncols; ++c) a.col[c] = J.col_values[c];
    for (size_t o = 0; o < J.outs.size(); ++o) {
        if (!J.outs[o].is_column)
            for (int i = 0; i < J.outs[o].P.n; ++i) {
                a.lit[o][2 * i] = J.outs[o].P.ins[i].lit_a;
                a.lit[o][2 * i + 1] = J.outs[o].P.ins[i].lit_b;
            }
        a.out_words[o] = out_words[o];
    }
    for (int i = 0; i < S.pred.n; ++i) {
        a.plit[2 * i] = S.pred.ins[i].lit_a;
        a.plit[2 * i + 1] = S.pred.ins[i].lit_b;
    }
    a.n = n;
    a.n_steps = (n + SP_BLOCK * SP_R - 1) / (SP_BLOCK * SP_R);
    a.status = status;
    a.ticket = ticket;
    a.total = total;
    a.flags = ctx->d_flags;
    void *params[] = {&a};
    const unsigned grid = unsigned(stream_grid(ctx, a.n_steps, 1, 4));
    TimerScope t(ctx, "select_project_jit");
    ctx->flags_clean = false;
    NQE_HIP_CHECK(hipModuleLaunchKernel(e->fn, grid, 1, 1, SP_BLOCK, 1, 1, 0, ctx->stream, params, nullptr));
}

// ---- the streaming aggregate under a predicate TREE, specialised (nqe_jit_agg)
// The static streaming kernel evaluates tree predicates with interpreters (PRED 5 / 6 of aggregate_fast_kernel.hpp: wave-uniform
// dispatch per test or per stack-machine step) — `v < 20 or id % 3 = 0` ran at 0.55 of peak against 0.81 for a plain range test,
// and specialising THAT kernel costs ~20 s of compilation per instance.  This is a lean kernel for the common shape instead:
// group key `col % m` with a literal m (direct-mapped LDS table of m or 2m-1 slots, 512 <= slots <= 4096), ONE value column,
// columns without validity: tiles of 4 rows per lane with a second tile in flight, the predicate as straight-line code
// (emit_steps), the key's modulus and the value conversion baked in, the per-thread run cache (rows of a thread whose key repeats
// accumulate in registers; a changed key flushes to the LDS table: ds_add_u32 / ds_add_f64 / ds_min_f64 / ds_max_f64 behind
// read-before-atomic compares).  Every workgroup writes its table to a partials buffer [workgroup][slot]; a small static kernel
// (aggregate.hip: agg_merge_partials_kernel) folds them into the group table the rest of the operator works on — the generated
// source knows nothing of that table's layout.  Compiles in ~1 s; until then, and for every other shape, the interpreters run.
constexpr int JA_TU = 4, JA_BLOCK = 1024, JA_MAX_COLS = EX_MAX_COLS;
struct JitAgg {
    ExProgram pred;          // the predicate (n = 0: none); column operands renumbered to the slots of `col`
    ExProgram key;           // the group key: any fault-free integer program whose LAST step is `… % m` with a literal m (`id % 1024`,
                             // `(id + 1) % 1000`, `id / 7 % 50`)
    int ncols = 0;           // columns loaded: the union of what predicate, key and value reference
    const void *col[JA_MAX_COLS];
    int val_slot = 0, val_dtype = NQE_FLOAT64;
    bool key_signed = true;
    uint64_t modulus = 1;    // |m|
    uint32_t span = 1;       // table slots: m (unsigned keys) or 2m - 1 (signed: keys in (-m, m)); slot = key + (m - 1) resp. key
};
struct JitAggArgs {
    const void *col[JA_MAX_COLS];
    uint64_t plit[JIT_MAX_LITS], klit[JIT_MAX_LITS];
    int64_t n;
    double *psum, *pmn, *pmx;
    uint32_t *pcnt;
    int *flags;
};
uint64_t jit_hash_agg(const JitAgg &G) {
    uint64_t h = jit_hash(G.pred, false, true) ^ (jit_hash(G.key, false, false) * 0x9E3779B97F4A7C15ull) ^ 0x616767ull;
    auto mix = [&](const void *p, size_t nb) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < nb; ++i) h = (h ^ b[i]) * 1099511628211ull;
    };
    const int64_t head[6] = {G.ncols, G.val_slot, G.val_dtype, G.key_signed ? 1 : 0, int64_t(G.modulus), int64_t(G.span)};
    mix(head, sizeof(head));
    return h ? h : 1;
}
std::string gen_source_agg(const JitAgg &G) {
    std::ostringstream s;
    const std::string vs = std::to_string(G.val_slot);
    s << "#pragma clang fp contract(off)\n"
      << "typedef unsigned long long u64; typedef long long i64; typedef unsigned int u32;\n"
      << "#define R " << JA_TU << "\n#define TU " << JA_TU << "\n#define BLOCK " << JA_BLOCK << "\n#define NC " << G.ncols << "\n#define S " << G.span << "u\n"
      << "#define NAN_BIT 0x80000000u\n"
      << "struct Args { const void *col[" << JA_MAX_COLS << "]; u64 plit[" << JIT_MAX_LITS << "], klit[" << JIT_MAX_LITS << "]; i64 n; double *psum, *pmn, *pmx; u32 *pcnt; int *flags; };\n"
      << "struct Tile { u64 c[NC][TU]; };\n"
      << "static __device__ __forceinline__ double u2d(u64 w) { return __longlong_as_double((i64)w); }\n"
      << "static __device__ __forceinline__ u64 d2u(double d) { return (u64)__double_as_longlong(d); }\n"
      << "extern \"C\" __global__ void __launch_bounds__(BLOCK) nqe_jit_agg(Args A) {\n"
      << "  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];\n"
      << "  double *lsum = (double *)smem, *lmn = lsum + S, *lmx = lmn + S;\n"
      << "  u32 *lcnt = (u32 *)(lmx + S);\n"
      << "  for (u32 i = threadIdx.x; i < S; i += BLOCK) { lsum[i] = 0.0; lmn[i] = 1.7976931348623157e308; lmx[i] = -1.7976931348623157e308; lcnt[i] = 0u; }\n"
      << "  __syncthreads();\n"
      << "  const i64 n = A.n, last = n - 1, step = (i64)BLOCK * TU, stride = (i64)gridDim.x * step;\n"
      << "  const u64 *__restrict__ cp[NC];\n"
      << "#pragma unroll\n  for (int k = 0; k < NC; ++k) cp[k] = (const u64 *)A.col[k];\n"
      << "  u32 lane_row[TU];\n"
      << "#pragma unroll\n  for (int u = 0; u < TU; ++u) lane_row[u] = (u32)u * BLOCK + threadIdx.x;\n"
      // whole tiles: scalar tile pointer + loop-invariant 32-bit lane offsets; the last tile: clamped rows
      << "  auto load = [&](Tile &t, i64 base) {\n"
      << "    if (base + step <= n) {\n"
      << "#pragma unroll\n      for (int k = 0; k < NC; ++k) { const u64 *__restrict__ p = cp[k] + base;\n"
      << "#pragma unroll\n        for (int u = 0; u < TU; ++u) t.c[k][u] = __builtin_nontemporal_load(&p[lane_row[u]]); }\n"
      << "    } else {\n"
      << "#pragma unroll\n      for (int u = 0; u < TU; ++u) { i64 row = base + lane_row[u]; row = row < last ? row : last;\n"
      << "#pragma unroll\n        for (int k = 0; k < NC; ++k) t.c[k][u] = __builtin_nontemporal_load(&cp[k][row]); }\n"
      << "    }\n"
      << "  };\n"
      << "  bool run_live = false; u32 run_slot = 0, rcnt = 0; double rsum = 0.0, rmn = 1.7976931348623157e308, rmx = -1.7976931348623157e308; bool rnan = false;\n"
      << "  auto flush = [&]() {\n"
      << "    atomicAdd(&lcnt[run_slot], rcnt);\n"
      << "    unsafeAtomicAdd(&lsum[run_slot], rsum);\n"
      << "    if (rnan) atomicOr(&lcnt[run_slot], NAN_BIT);\n"
      << "    if (rmn < lmn[run_slot]) unsafeAtomicMin(&lmn[run_slot], rmn);\n"
      << "    if (rmx > lmx[run_slot]) unsafeAtomicMax(&lmx[run_slot], rmx);\n"
      << "    rcnt = 0; rsum = 0.0; rmn = 1.7976931348623157e308; rmx = -1.7976931348623157e308; rnan = false;\n"
      << "  };\n"
      << "  auto process = [&](const Tile &t, i64 base) {\n"
      << "    bool in[TU];\n"
      << "#pragma unroll\n    for (int u = 0; u < TU; ++u) in[u] = base + lane_row[u] < n;\n";
    for (int c = 0; c < G.ncols; ++c) s << "    const u64 (&c" << c << ")[TU] = t.c[" << c << "];\n";
    bool has_valid[JP_MAX_COLS];
    for (int c = 0; c < JP_MAX_COLS; ++c) has_valid[c] = false;
    std::string pass = "in[u]";
    if (G.pred.n > 0) {
        const auto pres = emit_steps(s, G.pred, has_valid, false, "p", "A.plit", "in[r]");
        pass = "(in[u] && " + pres.first + "[u] != 0ull)";
    }
    // the key: its steps as straight-line code (the literal modulus of the last step is baked in: multiply-shift sequences); rows the
    // predicate drops compute a key nobody uses
    const auto kres = emit_steps(s, G.key, has_valid, false, "k", "A.klit", "in[r]");
    s << "#pragma unroll\n    for (int u = 0; u < TU; ++u) {\n"
      << "      if (!" << pass << ") continue;\n";
    if (G.key_signed) s << "      const u32 slot = (u32)((i64)" << kres.first << "[u] + (i64)" << (G.modulus - 1) << "ll);\n";
    else s << "      const u32 slot = (u32)" << kres.first << "[u];\n";
    if (G.val_dtype == NQE_FLOAT64) s << "      const double x = u2d(c" << vs << "[u]);\n";
    else if (G.val_dtype == NQE_INT64) s << "      const double x = (double)(i64)c" << vs << "[u];\n";
    else s << "      const double x = (double)c" << vs << "[u];\n";
    s << "      if (!run_live || slot != run_slot) { if (run_live) flush(); run_slot = slot; run_live = true; }\n"
      << "      rcnt += 1u; rsum += x; rnan = rnan || (x != x); rmn = fmin(rmn, x); rmx = fmax(rmx, x);\n" // (fmin / fmax ignore a NaN operand)
      << "    }\n"
      << "  };\n"
      << "  i64 base = (i64)blockIdx.x * step;\n"
      << "  if (base < n) {\n"
      << "    Tile T0, T1;\n"
      << "    load(T0, base);\n"
      << "    for (;;) {\n"
      << "      load(T1, base + stride);\n" // (a tile past the table reads clamped rows it then ignores)
      << "      process(T0, base); base += stride; if (base >= n) break;\n"
      << "      load(T0, base + stride);\n"
      << "      process(T1, base); base += stride; if (base >= n) break;\n"
      << "    }\n"
      << "  }\n"
      << "  if (run_live) flush();\n"
      << "  __syncthreads();\n"
      << "  const size_t o = (size_t)blockIdx.x * S;\n"
      << "  for (u32 i = threadIdx.x; i < S; i += BLOCK) { A.psum[o + i] = lsum[i]; A.pmn[o + i] = lmn[i]; A.pmx[o + i] = lmx[i]; A.pcnt[o + i] = lcnt[i]; }\n"
      << "}\n";
    return s.str();
}
JitEntry *jit_aggregate_entry(nqe_ctx *ctx, const JitAgg &G) {
    return jit_get(ctx, jit_hash_agg(G), "nqe_jit_agg", [&] { return gen_source_agg(G); }, "-munsafe-fp-atomics");
}
void jit_aggregate_launch(nqe_ctx *ctx, JitEntry *e, const JitAgg &G, int64_t n, unsigned grid, double *psum, double *pmn, double *pmx, uint32_t *pcnt) {
    JitAggArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int c = 0; c < G.ncols; ++c) a.col[c] = G.col[c];
    for (int i = 0; i < G.pred.n; ++i) {
        a.plit[2 * i] = G.pred.ins[i].lit_a;
        a.plit[2 * i + 1] = G.pred.ins[i].lit_b;
    }
    for (int i = 0; i < G.key.n; ++i) {
        a.klit[2 * i] = G.key.ins[i].lit_a;
        a.klit[2 * i + 1] = G.key.ins[i].lit_b;
    }
    a.n = n;
    a.psum = psum; a.pmn = pmn; a.pmx = pmx; a.pcnt = pcnt;
    a.flags = ctx->d_flags;
    void *params[] = {&a};
    TimerScope t(ctx, "agg_grouped_jit");
    ctx->flags_clean = false;
    NQE_HIP_CHECK(hipModuleLaunchKernel(e->fn, grid, 1, 1, JA_BLOCK, 1, 1, unsigned(size_t(G.span) * 28), ctx->stream, params, nullptr));
}

// blocks until every compilation in flight has finished (tests: the next execution takes the specialised kernels)
void jit_wait_all(nqe_ctx *ctx) {
    if (!ctx->jit) return;
    for (auto &kv : static_cast<JitCache *>(ctx->jit.get())->entries)
        if (kv.second->worker.joinable()) kv.second->worker.join();
}

} // namespace
} // namespace nqe
