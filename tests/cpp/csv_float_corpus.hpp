// Corpus of decimal strings for the number-parser tests (host: test_csv_parse.cpp, device: test_csv_parse_device.hip):
// random bit patterns at several precisions, human decimals, random digit strings with exponents, exact halfway cases
// (up to 70 digits) and edge values.  Every string is valid for both strtod and the lexical-core grammar.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

inline std::vector<std::string> csv_float_corpus() {
    std::vector<std::string> out;
    std::mt19937_64 rng(12345);
    char buf[512];
    // 1. random bit patterns printed with various precisions
    for (int i = 0; i < 400000; ++i) {
        uint64_t bits = rng();
        double d;
        memcpy(&d, &bits, 8);
        if (std::isnan(d) || std::isinf(d)) continue;
        static const char *fmts[] = {"%.17g", "%.15g", "%.9g", "%.20e", "%.30e", "%.45e", "%.3f", "%.17e"};
        snprintf(buf, sizeof buf, fmts[i % 8], d);
        if (strlen(buf) < 400) out.push_back(buf);
    }
    // 2. "human" decimals
    for (int i = 0; i < 400000; ++i) {
        long long ip = (long long)(rng() % 2000000000ull) - 1000000000ll;
        unsigned fp = unsigned(rng() % 100000000u);
        int w = int(rng() % 9);
        if (w) snprintf(buf, sizeof buf, "%lld.%0*u", ip, w, fp % unsigned(std::pow(10, w)));
        else snprintf(buf, sizeof buf, "%lld", ip);
        out.push_back(buf);
    }
    // 3. random digit strings (1..60 digits) with random exponents and dot positions
    for (int i = 0; i < 400000; ++i) {
        int nd = 1 + int(rng() % 60);
        std::string s;
        if (rng() & 1) s += '-';
        int dot = int(rng() % (nd + 1));
        for (int k = 0; k < nd; ++k) {
            if (k == dot && k != 0) s += '.';
            s += char('0' + rng() % 10);
        }
        if (rng() % 3) {
            int e = int(rng() % 700) - 350;
            s += (rng() & 1) ? 'e' : 'E';
            s += std::to_string(e);
        }
        out.push_back(s);
    }
    // 4. halfway cases: midpoint between two adjacent doubles, written exactly (they have finite decimal expansions)
    for (int i = 0; i < 20000; ++i) {
        int e = int(rng() % 120) - 60;
        uint64_t m = (1ull << 52) | (rng() & ((1ull << 52) - 1));
        // value = (2m+1) * 2^(e-1): print exactly via long double/bigint-free trick: use %.*Lf of ldexpl for |e| small
        long double v = ldexpl((long double)(2 * m + 1), e - 1 - 52);
        snprintf(buf, sizeof buf, "%.70Lf", v); // 64-bit significand long double holds 2m+1 (54 bits) exactly
        out.push_back(buf);
        snprintf(buf, sizeof buf, "%.40Le", v);
        out.push_back(buf);
    }
    // 5. edge cases
    const char *edges[] = {"0", "-0", "0.0", "000.000", "1", "+1", "1.", ".5", "-.5e1", "1e0", "1E+2", "1e-2", "4.9e-324", "2.4703282292062327e-324",
                           "2.4703282292062328e-324", "2.47e-324", "1e-400", "1e-323", "2.2250738585072014e-308", "2.2250738585072011e-308",
                           "1.7976931348623157e308", "1.7976931348623158e308", "1.7976931348623159e308", "1e309", "1e400", "123456789012345678901234567890",
                           "0.000000000000000000000000000001", "9007199254740993", "9007199254740992", "9007199254740991", "18014398509481985",
                           "8.5", "81.09666666666668", "100.0", "1e22", "1e23", "8.41e21", "2.2250738585072012e-308", "0.1", "0.3", "1e-22", "1e-23",
                           "179769313486231580793728971405303415079934132710037826936173778980444968292764750946649017977587207096330286416692887910946555547851940402630657488671505820681908902000708383676273854845817711531764475730270069855571366959622842914819860834936475292719074168444365510704342711559699508093042880177904174497791.9999999999999999999999999999999999999999999999999999999999999999999999"};
    for (const char *e : edges) out.push_back(e);
    return out;
}
