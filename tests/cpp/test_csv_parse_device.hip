// Device run of csrc/csv_parse.hpp over the shared corpus: every string is parsed by one GPU thread and compared bit for
// bit with glibc strtod on the host (and with the host build of the same parser).  Built and run by tests/test_gpu_csv.py.
#include <hip/hip_runtime.h>

#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../naive_query_engine_amd/csrc/csv_parse.hpp"
#include "csv_float_corpus.hpp"

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } \
    } while (0)

__global__ void parse_kernel(const char *blob, const int64_t *offs, int64_t n, uint64_t *bits, uint8_t *ok) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = 0;
    ok[i] = nqe::csvp::parse_f64(blob + offs[i], int(offs[i + 1] - offs[i]), &v) ? 1 : 0;
    uint64_t b;
    __builtin_memcpy(&b, &v, 8);
    bits[i] = b;
}

int main() {
    std::vector<std::string> corpus = csv_float_corpus();
    const char *extra[] = {"1e3", "1E+2", "12e1", "-0.0e0", "+.5E-3", "5.", "1e-400", "1e400", "nan", "-Infinity", "INF"};
    for (const char *e : extra) corpus.push_back(e);
    std::string blob;
    std::vector<int64_t> offs{0};
    for (auto &s : corpus) {
        blob += s;
        offs.push_back(int64_t(blob.size()));
    }
    const int64_t n = int64_t(corpus.size());
    char *d_blob;
    int64_t *d_offs;
    uint64_t *d_bits;
    uint8_t *d_ok;
    CK(hipMalloc(&d_blob, blob.size() + 8));
    CK(hipMalloc(&d_offs, offs.size() * 8));
    CK(hipMalloc(&d_bits, size_t(n) * 8));
    CK(hipMalloc(&d_ok, size_t(n)));
    CK(hipMemcpy(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_offs, offs.data(), offs.size() * 8, hipMemcpyHostToDevice));
    parse_kernel<<<dim3(unsigned((n + 255) / 256)), dim3(256)>>>(d_blob, d_offs, n, d_bits, d_ok);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> bits(static_cast<size_t>(n));
    std::vector<uint8_t> ok(static_cast<size_t>(n));
    CK(hipMemcpy(bits.data(), d_bits, size_t(n) * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ok.data(), d_ok, size_t(n), hipMemcpyDeviceToHost));
    long long failed = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double exp = strtod(corpus[size_t(i)].c_str(), nullptr);
        uint64_t eb;
        memcpy(&eb, &exp, 8);
        const bool both_nan = exp != exp && ((bits[size_t(i)] >> 52) & 0x7ff) == 0x7ff && (bits[size_t(i)] & 0xfffffffffffffull);
        if (!ok[size_t(i)] || (bits[size_t(i)] != eb && !both_nan)) {
            if (++failed < 20) printf("MISMATCH '%s': device ok=%d %016" PRIx64 ", strtod %016" PRIx64 "\n", corpus[size_t(i)].c_str(), int(ok[size_t(i)]), bits[size_t(i)], eb);
        }
    }
    printf("%lld device checks, %lld failures\n", (long long)n, failed);
    return failed ? 1 : 0;
}
