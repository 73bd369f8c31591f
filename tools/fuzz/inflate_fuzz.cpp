// Sanitizer fuzz of csrc/inflate_fast.h (CPU only): g++ -O1 -g -fsanitize=address,undefined inflate_fuzz.cpp -lz
// Mutates valid zlib streams (bit flips, byte splices, truncations) and checks that the decompressor never reads or
// writes outside its buffers and agrees with zlib's uncompress() on accept/reject and on the bytes.
#include "../../biscuit_amd/csrc/inflate_fast.h"

#include <stdio.h>
#include <stdlib.h>
#include <zlib.h>

#include <vector>

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    srand(12345);
    std::vector<std::vector<uint8_t>> datas;
    for (int kind = 0; kind < 4; ++kind) {
        std::vector<uint8_t> d(20000 + kind * 7919);
        int v = 128;
        for (auto& x : d) {
            if (kind == 0) x = (uint8_t)rand();
            else if (kind == 1) x = (uint8_t)(rand() & 1);
            else if (kind == 2) { v += rand() % 7 - 3; v = v < 0 ? 0 : v > 255 ? 255 : v; x = (uint8_t)v; }
            else x = (uint8_t)("hello world, "[(&x - d.data()) % 13]);
        }
        datas.push_back(d);
    }
    static bqinf::Tables T;
    long accepted = 0, rejected = 0, disagreements = 0;
    for (int it = 0; it < iters; ++it) {
        const auto& d = datas[(it / 4) % datas.size()];
        uLongf zn = (uLongf)d.size() * 2 + 1024;         // Z_FIXED expands random bytes by an eighth
        std::vector<uint8_t> z(zn);
        const int level = it % 10, strat[5] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED};
        z_stream zs{};
        deflateInit2(&zs, level, Z_DEFLATED, 9 + it % 7, 8, strat[(it / 10) % 5]);
        zs.next_in = (Bytef*)d.data(); zs.avail_in = (uInt)d.size(); zs.next_out = z.data(); zs.avail_out = (uInt)z.size();
        deflate(&zs, Z_FINISH);
        z.resize(zs.total_out);
        deflateEnd(&zs);
        if (it % 4) {
            const int nmut = 1 + rand() % 4;
            for (int m = 0; m < nmut; ++m) {
                const size_t k = (size_t)rand() % z.size();
                if (rand() & 1) z[k] ^= (uint8_t)(1 << (rand() & 7)); else z[k] = (uint8_t)rand();
            }
            if (rand() % 6 == 0) z.resize((size_t)rand() % z.size() + 1);
        }
        // exact-size buffers so that ASan sees any overrun: input + 16 zero bytes, output + slack
        std::vector<uint8_t> zin(z.size() + 16, 0), out(d.size() + bqinf::OUT_SLACK), ref(d.size());
        memcpy(zin.data(), z.data(), z.size());
        bool ok;
        if (it & 1) {                                    // odd iterations: through the two-stream loop, next to the previous stream
            static std::vector<uint8_t> pin, pout; static size_t pn = 0, plen = 0; static bqinf::Tables T2;
            bool pok = false;
            if (pin.empty()) ok = bqinf::inflate_zlib(zin.data(), z.size(), out.data(), d.size(), T);
            else {
                std::vector<uint8_t> pin2(pin), pout2(plen + bqinf::OUT_SLACK);
                bqinf::inflate_zlib2(pin2.data(), pn, pout2.data(), plen, T2, pok, zin.data(), z.size(), out.data(), d.size(), T, ok);
            }
            pin = zin; pn = z.size(); plen = d.size();
        } else {
            ok = bqinf::inflate_zlib(zin.data(), z.size(), out.data(), d.size(), T);
        }
        uLongf got = (uLongf)ref.size();
        const bool ref_ok = uncompress(ref.data(), &got, z.data(), (uLong)z.size()) == Z_OK && got == d.size();
        if (ok != ref_ok || (ok && memcmp(out.data(), ref.data(), d.size()) != 0)) {
            ++disagreements;
            fprintf(stderr, "disagreement at iteration %d (ours %d, zlib %d)\n", it, (int)ok, (int)ref_ok);
        }
        if (it % 4 == 0 && !ok) { ++disagreements; fprintf(stderr, "valid stream rejected at iteration %d\n", it); }
        ok ? ++accepted : ++rejected;
    }
    printf("iterations %d accepted %ld rejected %ld disagreements %ld\n", iters, accepted, rejected, disagreements);
    return disagreements != 0;
}
