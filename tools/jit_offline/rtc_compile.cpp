// compiles a source file with hipRTC exactly as the library does (options from argv), prints the log and resource usage is not available here
#include <hip/hiprtc.h>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <vector>
int main(int argc, char **argv) {
    std::ifstream f(argv[1]); std::stringstream ss; ss << f.rdbuf(); std::string src = ss.str();
    hiprtcProgram p; if (hiprtcCreateProgram(&p, src.c_str(), "t.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return 2;
    std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};
    for (int i = 2; i < argc; ++i) opts.push_back(argv[i]);
    hiprtcResult r = hiprtcCompileProgram(p, int(opts.size()), opts.data());
    size_t ls = 0; hiprtcGetProgramLogSize(p, &ls); if (ls > 1) { std::string log(ls, 0); hiprtcGetProgramLog(p, &log[0]); printf("%s\n", log.c_str()); }
    size_t cs = 0; hiprtcGetCodeSize(p, &cs); printf("result %d code %zu bytes\n", int(r), cs);
    if (r == HIPRTC_SUCCESS && argc > 0) { std::vector<char> code(cs); hiprtcGetCode(p, code.data()); const char *op = getenv("NQE_RTC_OUT"); if (op) { FILE *o = fopen(op, "wb"); if (o) { fwrite(code.data(), 1, cs, o); fclose(o); } } }
    return r == HIPRTC_SUCCESS ? 0 : 1;
}
