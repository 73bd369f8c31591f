// Sanitizer fuzz of csrc/jpeg_baseline.h (CPU only):
//   g++ -O1 -g -fsanitize=address,undefined jpeg_fuzz.cpp -o jpeg_fuzz && ./jpeg_fuzz ITERS file.jpg [file.jpg ...]
// (tools/fuzz/make_jpeg_corpus.py writes the files).  Mutates valid JPEG files (byte flips, splices, truncations, marker
// injections) and decodes each from a heap buffer of exactly the file's size into an output of exactly px*px*3 bytes:
// the decoder may accept or refuse, never read or write outside those buffers.  Agreement of accepted streams with
// libjpeg is what tests/test_jpeg.py checks.
#include "../../biscuit_amd/csrc/jpeg_baseline.h"

#include <stdio.h>
#include <stdlib.h>

#include <fstream>
#include <iterator>
#include <memory>

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: jpeg_fuzz ITERS file.jpg ...\n"); return 2; }
    const int iters = atoi(argv[1]);
    srand(4242);
    std::vector<std::vector<uint8_t>> files;
    std::vector<int> px;
    for (int a = 2; a < argc; ++a) {
        std::ifstream f(argv[a], std::ios::binary);
        files.emplace_back((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const auto& d = files.back();
        int w = 0;
        for (size_t i = 2; i + 9 < d.size(); ++i)
            if (d[i] == 0xFF && (d[i + 1] == 0xC0 || d[i + 1] == 0xC1 || d[i + 1] == 0xC2)) { w = (d[i + 7] << 8) | d[i + 8]; break; }
        px.push_back(w);
    }
    std::unique_ptr<bqjpg::Scratch> S(new bqjpg::Scratch());
    long ok = 0, refused = 0, wrong = 0;
    for (int it = 0; it < iters; ++it) {
        // a fresh Scratch every few files: with one reused object the unstuffed-scan vector keeps the capacity of the
        // largest file seen, and a read past its size() lands in memory the sanitizer considers valid (the build also
        // defines _GLIBCXX_SANITIZE_VECTOR, which poisons size()..capacity())
        if (it % 4 == 0) S.reset(new bqjpg::Scratch());
        const size_t fi = (size_t)it % files.size();
        std::vector<uint8_t> d = files[fi];
        const int kind = rand() % 9;
        if (kind < 4) {
            for (int k = rand() % 4 + 1; k > 0; --k) d[(size_t)rand() % d.size()] = (uint8_t)rand();
        } else if (kind == 4) {
            d.resize((size_t)rand() % d.size() + 1);
        } else if (kind == 5) {                           // splice a stretch of the file over another place
            const size_t n = (size_t)rand() % 64 + 1, a = (size_t)rand() % (d.size() - n), b = (size_t)rand() % (d.size() - n);
            memmove(d.data() + a, d.data() + b, n);
        } else if (kind == 8) {                           // the scan cut down to a few bytes, the EOI kept: header valid,
            size_t sos = 0;                               // entropy-coded segment far shorter than its MCU count
            for (size_t i = 2; i + 3 < d.size(); ++i) if (d[i] == 0xFF && d[i + 1] == 0xDA) { sos = i; break; }
            if (sos) {
                const size_t hdr = sos + 2 + ((d[sos + 2] << 8) | d[sos + 3]);
                if (hdr < d.size()) {
                    const size_t keep = (size_t)rand() % 40;
                    std::vector<uint8_t> e(d.begin(), d.begin() + (hdr + keep < d.size() ? hdr + keep : hdr));
                    for (size_t i = hdr; i < e.size(); ++i) if (e[i] == 0xFF) e[i] = 0x7F;      // no markers inside what is kept
                    e.push_back(0xFF); e.push_back(0xD9);
                    d.swap(e);
                }
            }
        } else if (kind == 6) {                           // a marker where there was data
            const size_t a = (size_t)rand() % (d.size() - 1);
            d[a] = 0xFF; d[a + 1] = (uint8_t)(0xC0 + rand() % 64);
        } else {                                          // header fields: sizes, sampling, table ids
            const size_t a = 2 + (size_t)rand() % (d.size() < 700 ? d.size() - 2 : 700);
            d[a] = (uint8_t)rand();
        }
        // exact-size heap copies: any overrun is the sanitizer's
        std::unique_ptr<uint8_t[]> in(new uint8_t[d.size()]);
        memcpy(in.get(), d.data(), d.size());
        const int p = px[fi];
        std::unique_ptr<uint8_t[]> out(new uint8_t[(size_t)p * p * 3]);
        const int e = bqjpg::decode(in.get(), d.size(), p, out.get(), *S);
        if (e == bqjpg::OK) ++ok; else if (e == bqjpg::UNSUPPORTED) ++refused; else ++wrong;
    }
    printf("%d mutated files: %ld decoded, %ld refused, %ld of another size\n", iters, ok, refused, wrong);
    return 0;
}
