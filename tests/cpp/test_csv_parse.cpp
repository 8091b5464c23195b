// CPU unit test of csrc/csv_parse.hpp (the same code runs in the device kernel): parse_f64 / parse_i64 against glibc
// strtod / strtoll (correctly rounded) on random and adversarial inputs.  Built and run by tests/test_csv_host.py.
#include <cerrno>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../naive_query_engine_amd/csrc/csv_parse.hpp"
#include "csv_float_corpus.hpp"

static long long checked = 0, failed = 0;

static void check(const std::string &s) {
    double got = 0;
    bool ok = nqe::csvp::parse_f64(s.data(), int(s.size()), &got);
    char *end = nullptr;
    double exp = strtod(s.c_str(), &end);
    ++checked;
    uint64_t a, b;
    memcpy(&a, &got, 8);
    memcpy(&b, &exp, 8);
    if (!ok || (a != b && !(std::isnan(got) && std::isnan(exp)))) {
        if (++failed < 20) printf("MISMATCH '%s': got %.17g (%016" PRIx64 ") ok=%d, strtod %.17g (%016" PRIx64 ")\n", s.c_str(), got, a, int(ok), exp, b);
    }
}

int main() {
    for (const std::string &s : csv_float_corpus()) check(s);
    std::mt19937_64 rng(777);
    // 6. grammar rejections / specials (lexical-core rules)
    const char *bad[] = {"", "+", "-", ".", "e5", "1e", "1e+", "1.2.3", "1x", " 1", "1 ", "0x10", "--1", "1e5.5", "in", "nanx"};
    for (const char *b : bad) {
        double d;
        ++checked;
        if (nqe::csvp::parse_f64(b, int(strlen(b)), &d)) { ++failed; printf("ACCEPTED '%s'\n", b); }
    }
    const char *special[] = {"nan", "NaN", "-nan", "inf", "-inf", "+Infinity", "INF"};
    for (const char *sp : special) {
        double d = 0;
        ++checked;
        bool ok = nqe::csvp::parse_f64(sp, int(strlen(sp)), &d);
        if (!ok || !(std::isnan(d) || std::isinf(d))) { ++failed; printf("SPECIAL '%s' -> %g\n", sp, d); }
    }
    // 7. integers
    for (int i = 0; i < 200000; ++i) {
        int nd = 1 + int(rng() % 21);
        std::string s;
        if (rng() % 3 == 0) s += '-';
        for (int k = 0; k < nd; ++k) s += char('0' + rng() % 10);
        int64_t got = 0;
        bool ok = nqe::csvp::parse_i64(s.data(), int(s.size()), &got);
        errno = 0;
        long long exp = strtoll(s.c_str(), nullptr, 10);
        bool eok = errno == 0;
        ++checked;
        if (ok != eok || (ok && got != exp)) { if (++failed < 20) printf("INT '%s': got %lld ok=%d, strtoll %lld ok=%d\n", s.c_str(), (long long)got, ok, exp, eok); }
    }
    const char *ints[] = {"9223372036854775807", "-9223372036854775808", "+5", "0", "-0"};
    const long long intv[] = {INT64_MAX, INT64_MIN, 5, 0, 0};
    for (int i = 0; i < 5; ++i) {
        int64_t g;
        ++checked;
        if (!nqe::csvp::parse_i64(ints[i], int(strlen(ints[i])), &g) || g != intv[i]) { ++failed; printf("INT edge '%s'\n", ints[i]); }
    }
    const char *ibad[] = {"9223372036854775808", "-9223372036854775809", "", "-", "1.0", "1e3", " 1", "12a"};
    for (const char *b : ibad) {
        int64_t g;
        ++checked;
        if (nqe::csvp::parse_i64(b, int(strlen(b)), &g)) { ++failed; printf("INT accepted '%s'\n", b); }
    }
    printf("%lld checks, %lld failures\n", checked, failed);
    return failed ? 1 : 0;
}
