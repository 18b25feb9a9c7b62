// Sanitizer run of the product's parameter-file reader / writer / rescaler (rafft_amd/csrc/rafft_params.h) on the CPU:
// the one parser of the library that takes untrusted input (rafft_load_params, entered from rafft_api.hip).  GPU
// AddressSanitizer is not available on the pool; this code needs no GPU.  Built host-only with
// -fsanitize=address,undefined,float-cast-overflow by tests/test_params.py::test_parameter_reader_under_sanitizers,
// which feeds it well-formed files, the malformed corpus of the loader tests, truncations, byte flips and oversized files.
//
// usage: params_san_driver FILE...      prints one line per file: "ok <checksum>" or "rejected: <message>"
#include "../../rafft_amd/csrc/rafft_params.h"

#include <fstream>
#include <memory>
#include <sstream>

static unsigned long long checksum(const EnergyTables &t)
{
    unsigned long long h = 1469598103934665603ULL;
    const unsigned char *p = (const unsigned char *)&t;
    for (size_t i = 0; i < sizeof t; i++) { h ^= p[i]; h *= 1099511628211ULL; }
    return h;
}

int main(int argc, char **argv)
{
    int bad = 0;
    for (int a = 1; a < argc; a++) {
        std::ifstream f(argv[a], std::ios::binary);
        std::stringstream ss;
        ss << f.rdbuf();
        const std::string text = ss.str();
        rafft_par::ParamSet P;
        std::string err;
        if (!rafft_par::parse(text, P, err)) { printf("rejected: %s\n", err.c_str()); continue; }
        // what the library does with an accepted set: device tables at the temperatures asked for, and the writer
        std::unique_ptr<EnergyTables> h(new EnergyTables());
        unsigned long long sum = 0;
        const double temps[4] = {37.0, 4.5, 25.0, 60.0};
        bool ok = true;
        for (int k = 0; k < (P.has_dH ? 4 : 1) && ok; k++) {
            ok = rafft_par::scaled_tables(P, temps[k], h.get(), err);
            if (ok) sum ^= checksum(*h) + (unsigned long long)k;
        }
        if (!ok) { printf("rejected: %s\n", err.c_str()); continue; }
        // written and read again it is the same set (the writer emits what the reader takes)
        const std::string again = rafft_par::format(P);
        rafft_par::ParamSet Q;
        if (!rafft_par::parse(again, Q, err)) { printf("ROUND TRIP FAILED: %s\n", err.c_str()); bad = 1; continue; }
        std::unique_ptr<EnergyTables> h2(new EnergyTables());
        if (!rafft_par::scaled_tables(Q, 37.0, h2.get(), err) || !rafft_par::scaled_tables(P, 37.0, h.get(), err) || checksum(*h) != checksum(*h2)) {
            printf("ROUND TRIP CHANGED THE TABLES\n"); bad = 1; continue;
        }
        printf("ok %016llx\n", sum);
    }
    return bad;
}
