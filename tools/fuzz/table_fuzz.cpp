// The tile-table writer of libbiscuit_io (csrc/table_writer.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer: random float32 /
// float64 values of every magnitude and the specials through the formatter (each must read back to the same double with strtod), and
// tables with long and awkward slide names, with and without locations, written and read back line by line.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined tools/fuzz/table_fuzz.cpp -o table_fuzz && ./table_fuzz 200000
#include "../../biscuit_amd/csrc/table_writer.cpp"

#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <string>
#include <vector>

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 100000;
    std::mt19937_64 rng(7);
    long bad = 0;
    char buf[64];
    for (long i = 0; i < n; ++i) {
        double v;
        const uint64_t bits = rng();
        if (i % 3 == 0) memcpy(&v, &bits, 8);                                        // any double
        else if (i % 3 == 1) { uint32_t b32 = (uint32_t)bits; float f; memcpy(&f, &b32, 4); v = (double)f; }   // any float32, widened
        else v = (double)(float)((bits >> 11) * (1.0 / 9007199254740992.0)) * pow(10.0, (double)((int)(bits % 12) - 9));
        const int k = bqio_format_f64(v, buf, (int)sizeof buf);
        if (k < 0) { ++bad; continue; }
        if (v != v) { if (k != 0) ++bad; continue; }
        const double back = strtod(buf, nullptr);
        if (memcmp(&back, &v, 8) != 0 && !(back == 0 && v == 0)) ++bad;
        if (bqio_format_f64(v, buf, k) != BQIO_ERR_ARG) ++bad;                       // a buffer one byte short is refused, not overrun
    }
    // tables
    const char* path = "/tmp/bq_table_fuzz.csv";
    for (int with_loc = 0; with_loc < 2; ++with_loc) {
        bqio_table* t = bqio_table_open(path, "out,come \"x\"", with_loc, 0);
        if (!t) return 2;
        long rows = 0;
        for (int s = 0; s < 300; ++s) {
            std::string name(1 + rng() % 700, 'a');
            for (auto& c : name) c = "ab,\"\n\r x-_0"[rng() % 11];
            const int cnt = (int)(rng() % 600);
            std::vector<float> m(2 * cnt), sd(2 * cnt);
            std::vector<int64_t> loc(2 * cnt);
            for (int i = 0; i < 2 * cnt; ++i) {
                uint32_t b = (uint32_t)rng(); float f; memcpy(&f, &b, 4);
                m[i] = (f != f) ? 0.5f : f;
                b = (uint32_t)rng(); memcpy(&f, &b, 4); sd[i] = f;                  // (NaN / inf uncertainties are legal cells)
                loc[i] = (int64_t)rng();
            }
            const int e = bqio_table_rows(t, name.c_str(), (int64_t)rng(), with_loc ? loc.data() : nullptr, m.data(), sd.data(), cnt);
            if (e != BQIO_OK) ++bad;
            rows += cnt;
            if (s == 150) {                                                          // a NaN prediction writes nothing
                float nanm[2] = {0.5f, NAN}, one[2] = {0.f, 0.f};
                int64_t l2[2] = {0, 0};
                if (bqio_table_rows(t, "n", 0, with_loc ? l2 : nullptr, nanm, one, 1) != BQIO_ERR_NAN) ++bad;
                if (bqio_table_rows(t, "n", 0, with_loc ? nullptr : l2, one, one, 1) != BQIO_ERR_ARG) ++bad;     // locations must match the header
            }
        }
        int64_t r = 0, b = 0;
        if (bqio_table_close(t, &r, &b) != BQIO_OK || r != rows) ++bad;
        bqio_table* u = bqio_table_open("/tmp/bq_table_fuzz2.csv", "o", with_loc, 0);
        if (!u || bqio_table_append_file(u, path, 0, b) != BQIO_OK || bqio_table_append_file(u, path, b - 1, 5) == BQIO_OK) ++bad;   // past the end: refused
        int64_t b2 = 0;
        bqio_table_close(u, nullptr, &b2);
    }
    remove(path); remove("/tmp/bq_table_fuzz2.csv");
    printf("values %ld, problems %ld\n", n, bad);
    return bad ? 1 : 0;
}
