// libbiscuit_io.so: Slideflow tile TFRecords -> uint8 RGB tiles on the host (include/biscuit_io.h).
// TFRecord framing + CRC-32C, the subset of protobuf a tf.train.Example needs, and a PNG decoder
// (zlib inflate + scanline unfilter).  Host code only: g++, -lz, -lpthread.
#include "../../include/biscuit_io.h"
#include "inflate_fast.h"
#include "jpeg_baseline.h"

#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

std::string g_open_error;

// ---- CRC-32C (Castagnoli), slicing-by-8 ------------------------------------------------
uint32_t g_crc[8][256];
bool g_crc_ready = false;

void crc_init() {
    if (g_crc_ready) return;
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
        g_crc[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) g_crc[t][i] = (g_crc[t - 1][i] >> 8) ^ g_crc[0][g_crc[t - 1][i] & 255];
    g_crc_ready = true;
}

uint32_t crc32c(const uint8_t* p, size_t n) {
    uint32_t c = 0xFFFFFFFFu;
    while (n && ((uintptr_t)p & 7)) { c = g_crc[0][(c ^ *p++) & 255] ^ (c >> 8); --n; }
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        v ^= c;
        c = g_crc[7][v & 255] ^ g_crc[6][(v >> 8) & 255] ^ g_crc[5][(v >> 16) & 255] ^ g_crc[4][(v >> 24) & 255] ^
            g_crc[3][(v >> 32) & 255] ^ g_crc[2][(v >> 40) & 255] ^ g_crc[1][(v >> 48) & 255] ^ g_crc[0][v >> 56];
        p += 8; n -= 8;
    }
    while (n--) c = g_crc[0][(c ^ *p++) & 255] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

uint32_t masked(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xA282EAD8u; }

// ---- protobuf subset ----------------------------------------------------------------------
struct Span { const uint8_t* p = nullptr; size_t n = 0; };

bool varint(const uint8_t*& p, const uint8_t* end, uint64_t& v) {
    v = 0;
    for (int shift = 0; p < end && shift < 64; shift += 7) {
        const uint8_t b = *p++;
        v |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) return true;
    }
    return false;
}

// iterate the fields of one message; returns false on malformed input
template <typename F>
bool fields(Span m, F&& f) {
    const uint8_t* p = m.p;
    const uint8_t* end = m.p + m.n;
    while (p < end) {
        uint64_t key;
        if (!varint(p, end, key)) return false;
        const int fn = (int)(key >> 3), wt = (int)(key & 7);
        if (wt == 0) {
            uint64_t v;
            if (!varint(p, end, v)) return false;
            f(fn, wt, v, Span{});
        } else if (wt == 2) {
            uint64_t len;
            if (!varint(p, end, len) || len > (uint64_t)(end - p)) return false;
            f(fn, wt, 0, Span{p, (size_t)len});
            p += len;
        } else if (wt == 1) {
            if (end - p < 8) return false;
            p += 8;
        } else if (wt == 5) {
            if (end - p < 4) return false;
            p += 4;
        } else {
            return false;
        }
    }
    return true;
}

struct Example {
    Span image, slide;
    int64_t loc_x = 0, loc_y = 0;
};

bool parse_example(Span rec, Example& ex) {
    bool ok = true;
    ok &= fields(rec, [&](int fn, int, uint64_t, Span features) {
        if (fn != 1) return;
        ok &= fields(features, [&](int fn2, int, uint64_t, Span entry) {
            if (fn2 != 1) return;
            Span key, feat;
            ok &= fields(entry, [&](int fn3, int, uint64_t, Span v) {
                if (fn3 == 1) key = v;
                else if (fn3 == 2) feat = v;
            });
            if (!key.p || !feat.p) return;
            const std::string k((const char*)key.p, key.n);
            ok &= fields(feat, [&](int kind, int, uint64_t, Span lst) {
                if (kind == 1) {               // BytesList
                    Span first;
                    ok &= fields(lst, [&](int f, int, uint64_t, Span v) { if (f == 1 && !first.p) first = v; });
                    if (k == "image_raw") ex.image = first;
                    else if (k == "slide") ex.slide = first;
                } else if (kind == 3) {        // Int64List, packed or not
                    int64_t val = 0;
                    bool have = false;
                    ok &= fields(lst, [&](int f, int wt, uint64_t v, Span packed) {
                        if (f != 1 || have) return;
                        if (wt == 0) { val = (int64_t)v; have = true; }
                        else {
                            const uint8_t* q = packed.p;
                            uint64_t x;
                            if (varint(q, packed.p + packed.n, x)) { val = (int64_t)x; have = true; }
                        }
                    });
                    if (k == "loc_x") ex.loc_x = val;
                    else if (k == "loc_y") ex.loc_y = val;
                }
            });
        });
    });
    return ok;
}

// ---- PNG ------------------------------------------------------------------------------------
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int image_format(Span img) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (img.n >= 8 && memcmp(img.p, sig, 8) == 0) return BQIO_IMG_PNG;
    if (img.n >= 3 && img.p[0] == 0xFF && img.p[1] == 0xD8 && img.p[2] == 0xFF) return BQIO_IMG_JPEG;
    return BQIO_IMG_UNKNOWN;
}

inline uint8_t paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (uint8_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}

// Average and Paeth scanlines are serial in the pixel to the left: the byte loops above, with their data-dependent
// branches, ran at ~14 cycles per byte (1.8 ms of a 2.6 ms tile whose encoder chose Paeth).  One pixel per step, its
// channels as independent chains, every select a conditional move: 0.35 ms per tile.  (An SSE2 form -- channels in
// 16-bit lanes -- measured 1.6 ms: the 3-byte loads and stores cost more than the arithmetic saves.)
template <int BPP>
void unfilter_paeth(uint8_t* cur, const uint8_t* up, size_t stride) {
    int a[BPP] = {0}, c[BPP] = {0};
    for (size_t i = 0; i < stride; i += BPP) {
#pragma GCC unroll 4
        for (int k = 0; k < BPP; ++k) {
            const int b = up[i + k];
            const int pa = b - c[k], pb = a[k] - c[k], pc = pa + pb;             // p - a, p - b, p - c with p = a + b - c
            const int aa = pa < 0 ? -pa : pa, ab = pb < 0 ? -pb : pb, ac = pc < 0 ? -pc : pc;
            int pred = c[k];
            pred = ab <= ac ? b : pred;
            pred = (aa <= ab && aa <= ac) ? a[k] : pred;                        // ties: a, then b, then c
            a[k] = (cur[i + k] + pred) & 0xFF;
            cur[i + k] = (uint8_t)a[k];
            c[k] = b;
        }
    }
}

template <int BPP>
void unfilter_avg(uint8_t* cur, const uint8_t* up, size_t stride) {
    int a[BPP] = {0};
    for (size_t i = 0; i < stride; i += BPP) {
#pragma GCC unroll 4
        for (int k = 0; k < BPP; ++k) {
            a[k] = (cur[i + k] + ((a[k] + up[i + k]) >> 1)) & 0xFF;
            cur[i + k] = (uint8_t)a[k];
        }
    }
}

// returns BQIO_OK / BQIO_ERR_FORMAT (wrong size) / BQIO_ERR_UNSUPPORTED / BQIO_ERR_CORRUPT
std::atomic<long long> g_inflate_fallbacks{0};     // streams zlib accepted after inflate_fast.h refused them (a bug if ever > 0)

// One PNG tile in three steps, so that a worker can inflate two tiles' streams in one loop (inflate_fast.h:
// inflate_zlib2): parse the chunks, inflate, unfilter into the RGB output.
struct PngJob {
    std::vector<uint8_t> zbuf, raw;          // IDAT payloads (+ 16 zero bytes), scanlines (+ slack)
    bqinf::Tables tables;
    uint8_t pal[256][3];
    int npal = 0, ctype = -1, bpp = 0;
    uint32_t w = 0, h = 0;
    size_t stride = 0, raw_n = 0, z_n = 0;
};

// returns BQIO_OK / BQIO_ERR_FORMAT (wrong size) / BQIO_ERR_UNSUPPORTED / BQIO_ERR_CORRUPT
int png_parse(Span img, int px, PngJob& J) {
    std::vector<uint8_t>& zbuf = J.zbuf;
    std::vector<uint8_t>& raw = J.raw;
    uint8_t (&pal)[256][3] = J.pal;
    int& npal = J.npal;
    const uint8_t* p = img.p + 8;
    const uint8_t* end = img.p + img.n;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    npal = 0;
    zbuf.clear();
    bool seen_end = false;
    while (end - p >= 12 && !seen_end) {
        const uint32_t len = be32(p);
        if ((uint64_t)len + 12 > (uint64_t)(end - p)) return BQIO_ERR_CORRUPT;
        const uint8_t* type = p + 4;
        const uint8_t* data = p + 8;
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) return BQIO_ERR_CORRUPT;
            w = be32(data); h = be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            npal = (int)(len / 3 < 256 ? len / 3 : 256);
            memcpy(pal, data, (size_t)npal * 3);
        } else if (!memcmp(type, "IDAT", 4)) {
            zbuf.insert(zbuf.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            seen_end = true;
        }
        p += (size_t)len + 12;
    }
    if (ctype < 0 || zbuf.empty()) return BQIO_ERR_CORRUPT;
    if (depth != 8 || interlace != 0 || !(ctype == 0 || ctype == 2 || ctype == 3 || ctype == 6))
        return BQIO_ERR_UNSUPPORTED;
    if ((int)w != px || (int)h != px) return BQIO_ERR_FORMAT;
    J.ctype = ctype; J.w = w; J.h = h;
    J.bpp = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : 4;
    J.stride = (size_t)w * J.bpp;
    J.raw_n = (J.stride + 1) * h; J.z_n = zbuf.size();
    raw.resize(J.raw_n + bqinf::OUT_SLACK);
    zbuf.insert(zbuf.end(), 16, (uint8_t)0);              // the bit reader loads 8 bytes at a time
    return BQIO_OK;
}

// zlib as the second opinion on a stream inflate_fast.h refused
bool inflate_second_opinion(PngJob& J) {
    uLongf got = (uLongf)J.raw_n;
    if (uncompress(J.raw.data(), &got, J.zbuf.data(), (uLong)J.z_n) != Z_OK || got != J.raw_n) return false;
    ++g_inflate_fallbacks;
    return true;
}

int png_finish(PngJob& J, uint8_t* out) {
    std::vector<uint8_t>& raw = J.raw;
    const uint8_t (&pal)[256][3] = J.pal;
    const int npal = J.npal, ctype = J.ctype, bpp = J.bpp;
    const uint32_t w = J.w, h = J.h;
    const size_t stride = J.stride;
    // unfilter in place (the filter byte stays in front of every scanline)
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t* cur = raw.data() + (stride + 1) * y + 1;
        const uint8_t* up = y ? cur - (stride + 1) : nullptr;
        const int ft = cur[-1];
        switch (ft) {
        case 0: break;
        case 1: for (size_t i = bpp; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
        case 2: if (up) for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + up[i]); break;
        case 3:
            if (up && bpp == 3) { unfilter_avg<3>(cur, up, stride); break; }
            if (up && bpp == 4) { unfilter_avg<4>(cur, up, stride); break; }
            if (up && bpp == 1) { unfilter_avg<1>(cur, up, stride); break; }
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0;
                cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
            }
            break;
        case 4:
            if (up && bpp == 3) { unfilter_paeth<3>(cur, up, stride); break; }
            if (up && bpp == 4) { unfilter_paeth<4>(cur, up, stride); break; }
            if (up && bpp == 1) { unfilter_paeth<1>(cur, up, stride); break; }
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0;
                const int c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
                cur[i] = (uint8_t)(cur[i] + paeth(a, b, c));
            }
            break;
        default: return BQIO_ERR_CORRUPT;
        }
        uint8_t* o = out + (size_t)y * w * 3;
        if (ctype == 2) {
            memcpy(o, cur, stride);
        } else if (ctype == 6) {
            for (uint32_t x = 0; x < w; ++x) { o[3 * x] = cur[4 * x]; o[3 * x + 1] = cur[4 * x + 1]; o[3 * x + 2] = cur[4 * x + 2]; }
        } else if (ctype == 0) {
            for (uint32_t x = 0; x < w; ++x) o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = cur[x];
        } else {
            for (uint32_t x = 0; x < w; ++x) {
                const int idx = cur[x];
                if (idx >= npal) return BQIO_ERR_CORRUPT;
                o[3 * x] = pal[idx][0]; o[3 * x + 1] = pal[idx][1]; o[3 * x + 2] = pal[idx][2];
            }
        }
    }
    return BQIO_OK;
}

}  // namespace

struct bqio_reader {
    int fd = -1;
    const uint8_t* base = nullptr;
    size_t size = 0;
    std::vector<Span> records;
    std::string err;
    int verify = 0;                          // BQIO_VERIFY_*: FULL also checks the PNG chunk CRCs where a caller takes raw streams
};

extern "C" {

const char* bqio_last_error(bqio_reader* r) { return r ? r->err.c_str() : g_open_error.c_str(); }

uint32_t bqio_masked_crc32c(const uint8_t* data, size_t len) {
    crc_init();
    return masked(crc32c(data, len));
}

bqio_reader* bqio_open(const char* path, int verify) {
    crc_init();
    if (!path) { g_open_error = "path is null"; return nullptr; }
    bqio_reader* r = new (std::nothrow) bqio_reader();
    if (r) r->verify = verify;
    if (!r) { g_open_error = "out of memory"; return nullptr; }
    auto fail = [&](const std::string& m) { g_open_error = std::string(path) + ": " + m; bqio_close(r); return nullptr; };
    r->fd = open(path, O_RDONLY);
    if (r->fd < 0) return fail("cannot open");
    struct stat st;
    if (fstat(r->fd, &st) != 0) return fail("cannot stat");
    r->size = (size_t)st.st_size;
    if (r->size) {
        void* m = mmap(nullptr, r->size, PROT_READ, MAP_PRIVATE, r->fd, 0);
        if (m == MAP_FAILED) return fail("cannot map");
        r->base = (const uint8_t*)m;
    }
    size_t off = 0;
    while (off < r->size) {
        if (r->size - off < 12) return fail("truncated record header");
        uint64_t len;
        uint32_t lcrc;
        memcpy(&len, r->base + off, 8);
        memcpy(&lcrc, r->base + off + 8, 4);
        if (verify != BQIO_VERIFY_NONE && masked(crc32c(r->base + off, 8)) != lcrc) return fail("corrupt record length");
        if (len > r->size - off - 12 || r->size - off - 12 - len < 4) return fail("truncated record");
        const uint8_t* data = r->base + off + 12;
        if (verify == BQIO_VERIFY_FULL) {
            uint32_t dcrc;
            memcpy(&dcrc, data + len, 4);
            if (masked(crc32c(data, (size_t)len)) != dcrc) return fail("corrupt record data");
        }
        r->records.push_back(Span{data, (size_t)len});
        off += 12 + (size_t)len + 4;
    }
    return r;
}

void bqio_close(bqio_reader* r) {
    if (!r) return;
    if (r->base) munmap((void*)r->base, r->size);
    if (r->fd >= 0) close(r->fd);
    delete r;
}

int64_t bqio_count(bqio_reader* r) { return r ? (int64_t)r->records.size() : BQIO_ERR_ARG; }

int bqio_slide_name(bqio_reader* r, char* buf, int buflen) {
    if (!r || !buf || buflen <= 0) return BQIO_ERR_ARG;
    buf[0] = 0;
    if (r->records.empty()) return 0;
    Example ex;
    if (!parse_example(r->records[0], ex)) { r->err = "malformed Example"; return BQIO_ERR_CORRUPT; }
    const int n = (int)(ex.slide.n < (size_t)(buflen - 1) ? ex.slide.n : (size_t)(buflen - 1));
    if (n) memcpy(buf, ex.slide.p, (size_t)n);
    buf[n] = 0;
    return n;
}

int bqio_image_bytes(bqio_reader* r, int64_t index, const uint8_t** data, size_t* len) {
    if (!r || index < 0 || index >= (int64_t)r->records.size() || !data || !len) return BQIO_ERR_ARG;
    Example ex;
    if (!parse_example(r->records[(size_t)index], ex) || !ex.image.p) { r->err = "record without image_raw"; return BQIO_ERR_CORRUPT; }
    *data = ex.image.p;
    *len = ex.image.n;
    return BQIO_OK;
}

int bqio_image_format(bqio_reader* r, int64_t index) {
    const uint8_t* d;
    size_t n;
    const int e = bqio_image_bytes(r, index, &d, &n);
    return e < 0 ? e : image_format(Span{d, n});
}

int bqio_inflate(const uint8_t* zdata, size_t n, uint8_t* out, size_t out_len) {
    if (!zdata || (!out && out_len)) return BQIO_ERR_ARG;
    std::vector<uint8_t> z(zdata, zdata + n), raw(out_len + bqinf::OUT_SLACK);
    z.insert(z.end(), 16, (uint8_t)0);
    std::vector<bqinf::Tables> tables(1);
    if (!bqinf::inflate_zlib(z.data(), n, raw.data(), out_len, tables[0])) return BQIO_ERR_CORRUPT;
    if (out_len) memcpy(out, raw.data(), out_len);
    return BQIO_OK;
}

int bqio_inflate2(const uint8_t* za, size_t na, uint8_t* out_a, size_t len_a, const uint8_t* zb, size_t nb, uint8_t* out_b,
                  size_t len_b, int* ok_a, int* ok_b) {
    if (!za || !zb || !ok_a || !ok_b || (!out_a && len_a) || (!out_b && len_b)) return BQIO_ERR_ARG;
    std::vector<uint8_t> a(za, za + na), b(zb, zb + nb), ra(len_a + bqinf::OUT_SLACK), rb(len_b + bqinf::OUT_SLACK);
    a.insert(a.end(), 16, (uint8_t)0);
    b.insert(b.end(), 16, (uint8_t)0);
    std::vector<bqinf::Tables> tables(2);
    bool oa = false, ob = false;
    bqinf::inflate_zlib2(a.data(), na, ra.data(), len_a, tables[0], oa, b.data(), nb, rb.data(), len_b, tables[1], ob);
    if (oa && len_a) memcpy(out_a, ra.data(), len_a);
    if (ob && len_b) memcpy(out_b, rb.data(), len_b);
    *ok_a = oa; *ok_b = ob;
    return BQIO_OK;
}

int64_t bqio_inflate_fallbacks(void) { return (int64_t)g_inflate_fallbacks.load(); }

// rows: leave the PNG scanline filters in (bqio_decode_rows); a tile's output is then px rows of 1 + 3 px bytes
static int decode_impl(bqio_reader* r, int64_t first, int64_t count, int tile_px, uint8_t* out, int64_t* loc, int n_threads,
                       int64_t* bad_index, bool rows) {
    if (!r || first < 0 || count < 0 || first + count > (int64_t)r->records.size() || tile_px <= 0 || (count && !out))
        return BQIO_ERR_ARG;
    if (bad_index) *bad_index = -1;
    if (count == 0) return BQIO_OK;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > (count + 1) / 2) n_threads = (int)((count + 1) / 2);
    std::atomic<int64_t> next(0);
    std::atomic<int> status(BQIO_OK);
    std::atomic<int64_t> bad(-1);
    const size_t row_bytes = (size_t)tile_px * 3 + (rows ? 1 : 0);
    const size_t tile_bytes = (size_t)tile_px * row_bytes;
    auto work = [&]() {
        std::vector<PngJob> jobs(2);                      // two tiles at a time: their streams are inflated in one loop
        std::unique_ptr<bqjpg::Scratch> jpg;              // built on the first JPEG record
        std::vector<uint8_t> rgb;                         // rows mode: a tile decoded here on the host, before it is re-rowed
        // rows mode, a tile that is not an 8-bit RGB PNG: its pixels as px rows of filter type 0
        auto as_plain_rows = [&](const uint8_t* src, uint8_t* dst) {
            for (int y = 0; y < tile_px; ++y) {
                dst[(size_t)y * row_bytes] = 0;
                memcpy(dst + (size_t)y * row_bytes + 1, src + (size_t)y * tile_px * 3, (size_t)tile_px * 3);
            }
        };
        auto fail = [&](int e, int64_t i) {
            int expect = BQIO_OK;
            if (status.compare_exchange_strong(expect, e)) bad.store(first + i);
        };
        for (;;) {
            const int64_t i0 = next.fetch_add(2);
            if (i0 >= count || status.load() != BQIO_OK) return;
            const int nj = i0 + 1 < count ? 2 : 1;
            Example ex[2];
            int png[2], npng = 0;                          // the records of this pair that are PNG -> jobs[0..npng)
            for (int k = 0; k < nj; ++k) {
                int e = BQIO_OK;
                if (!parse_example(r->records[(size_t)(first + i0 + k)], ex[k]) || !ex[k].image.p) e = BQIO_ERR_CORRUPT;
                else switch (image_format(ex[k].image)) {
                    case BQIO_IMG_PNG:
                        e = png_parse(ex[k].image, tile_px, jobs[npng]);
                        png[npng++] = k;
                        break;
                    case BQIO_IMG_JPEG: {
                        if (!jpg) jpg.reset(new bqjpg::Scratch());
                        uint8_t* o = out + (size_t)(i0 + k) * tile_bytes;
                        if (rows) rgb.resize((size_t)tile_px * tile_px * 3);
                        const int j = bqjpg::decode(ex[k].image.p, ex[k].image.n, tile_px, rows ? rgb.data() : o, *jpg);
                        e = j == bqjpg::OK ? BQIO_OK : j == bqjpg::WRONG_SIZE ? BQIO_ERR_FORMAT : BQIO_ERR_UNSUPPORTED;
                        if (rows && e == BQIO_OK) as_plain_rows(rgb.data(), o);
                        break;
                    }
                    default: e = BQIO_ERR_UNSUPPORTED;
                }
                if (e != BQIO_OK) { fail(e, i0 + k); return; }
                if (loc) { loc[2 * (i0 + k)] = ex[k].loc_x; loc[2 * (i0 + k) + 1] = ex[k].loc_y; }
            }
            bool ok[2] = {false, false};
            if (npng == 2)
                bqinf::inflate_zlib2(jobs[0].zbuf.data(), jobs[0].z_n, jobs[0].raw.data(), jobs[0].raw_n, jobs[0].tables, ok[0],
                                     jobs[1].zbuf.data(), jobs[1].z_n, jobs[1].raw.data(), jobs[1].raw_n, jobs[1].tables, ok[1]);
            else if (npng == 1)
                ok[0] = bqinf::inflate_zlib(jobs[0].zbuf.data(), jobs[0].z_n, jobs[0].raw.data(), jobs[0].raw_n, jobs[0].tables);
            for (int j = 0; j < npng; ++j) {
                const int k = png[j];
                uint8_t* o = out + (size_t)(i0 + k) * tile_bytes;
                int e = BQIO_ERR_CORRUPT;
                if (ok[j] || inflate_second_opinion(jobs[j])) {
                    PngJob& J = jobs[j];
                    if (!rows) e = png_finish(J, o);
                    else if (J.ctype == 2) {              // 8-bit RGB: the inflated stream IS the row format; only the filter types are checked
                        e = BQIO_OK;
                        for (uint32_t y = 0; y < J.h; ++y)
                            if (J.raw[(J.stride + 1) * y] > 4) e = BQIO_ERR_CORRUPT;
                        if (e == BQIO_OK) memcpy(o, J.raw.data(), J.raw_n);
                    } else {                              // grey, palette, RGBA: un-filtered and converted here
                        rgb.resize((size_t)tile_px * tile_px * 3);
                        e = png_finish(J, rgb.data());
                        if (e == BQIO_OK) as_plain_rows(rgb.data(), o);
                    }
                }
                if (e != BQIO_OK) { fail(e, i0 + k); return; }
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    const int e = status.load();
    if (e != BQIO_OK) {
        if (bad_index) *bad_index = bad.load();
        r->err = e == BQIO_ERR_UNSUPPORTED ? "image_raw is not a PNG or baseline JPEG this decoder handles"
                 : e == BQIO_ERR_FORMAT    ? "tile size differs from tile_px"
                                           : "corrupt record or PNG";
    }
    return e;
}

int bqio_probe(bqio_reader* r, int64_t first, int64_t count, int tile_px, int64_t* bad_index) {
    if (!r || first < 0 || count < 0 || first + count > (int64_t)r->records.size() || tile_px <= 0) return BQIO_ERR_ARG;
    if (bad_index) *bad_index = -1;
    std::unique_ptr<bqjpg::Scratch> jpg;
    for (int64_t i = first; i < first + count; ++i) {
        Example ex;
        int e = BQIO_OK;
        if (!parse_example(r->records[(size_t)i], ex) || !ex.image.p) e = BQIO_ERR_CORRUPT;
        else switch (image_format(ex.image)) {
            case BQIO_IMG_PNG: break;                      // lossless: whichever decoder takes it, the pixels are the same
            case BQIO_IMG_JPEG: {
                if (!jpg) jpg.reset(new bqjpg::Scratch());
                const int j = bqjpg::decode(ex.image.p, ex.image.n, tile_px, nullptr, *jpg, true);
                e = j == bqjpg::OK ? BQIO_OK : j == bqjpg::WRONG_SIZE ? BQIO_ERR_FORMAT : BQIO_ERR_UNSUPPORTED;
                break;
            }
            default: e = BQIO_ERR_UNSUPPORTED;
        }
        if (e != BQIO_OK) {
            if (bad_index) *bad_index = i;
            r->err = e == BQIO_ERR_UNSUPPORTED ? "image_raw is not a PNG or baseline JPEG this decoder handles"
                     : e == BQIO_ERR_FORMAT    ? "tile size differs from tile_px" : "corrupt record";
            return e;
        }
    }
    return BQIO_OK;
}

// The zlib streams of records [first, first + count), packed for the device inflate (bq_png_inflate): stream i = the concatenated
// IDAT payloads of record i at out_z + off[i], len[i] bytes, every off[i] a multiple of 16 and at least 32 zero bytes behind every
// stream (the device bit reader loads two dwords ahead).  Only what the device path handles passes: 8-bit RGB, non-interlaced PNG tiles of tile_px x tile_px (anything else:
// BQIO_ERR_UNSUPPORTED / BQIO_ERR_FORMAT with *bad_index, as bqio_decode would answer for a record outside ITS subset -- the caller
// falls back to bqio_decode for the slide).  *used = bytes of out_z written (<= cap, else BQIO_ERR_ARG with *used = bytes needed).
int bqio_extract_z(bqio_reader* r, int64_t first, int64_t count, int tile_px, uint8_t* out_z, size_t cap, uint32_t* off, uint32_t* len,
                   int64_t* loc, size_t* used, int n_threads, int64_t* bad_index) {
    if (!r || first < 0 || count < 0 || first + count > (int64_t)r->records.size() || tile_px <= 0 || !off || !len || !used ||
        (count && !out_z)) return BQIO_ERR_ARG;
    if (bad_index) *bad_index = -1;
    *used = 0;
    struct Piece { const uint8_t* p; uint32_t n; };
    std::vector<std::vector<Piece>> pieces((size_t)count);
    size_t at = 0;
    for (int64_t i = 0; i < count; ++i) {
        Example ex;
        int e = BQIO_OK;
        if (!parse_example(r->records[(size_t)(first + i)], ex) || !ex.image.p) e = BQIO_ERR_CORRUPT;
        else if (image_format(ex.image) != BQIO_IMG_PNG) e = BQIO_ERR_UNSUPPORTED;
        else {
            const uint8_t* p = ex.image.p + 8;
            const uint8_t* end = ex.image.p + ex.image.n;
            bool ihdr = false, seen_end = false;
            int idat = 0;                                      // 0: none yet, 1: inside the run of IDAT chunks, 2: behind it
            uint64_t total = 0;
            // stricter than the host decoder (round-5 advisory): the stream goes to the device as it is, so what a PNG reader
            // is entitled to refuse is refused HERE and the slide takes the host path -- IHDR first, compression and filter
            // method 0, IDAT chunks consecutive, chunk CRCs when the reader was opened with BQIO_VERIFY_FULL
            for (int nchunk = 0; end - p >= 12 && !seen_end && e == BQIO_OK; ++nchunk) {
                const uint32_t n = be32(p);
                if ((uint64_t)n + 12 > (uint64_t)(end - p)) { e = BQIO_ERR_CORRUPT; break; }
                const uint8_t* type = p + 4;
                const uint8_t* data = p + 8;
                if (r->verify == BQIO_VERIFY_FULL && (uint32_t)crc32(0L, type, (uInt)(n + 4)) != be32(data + n)) {
                    e = BQIO_ERR_CORRUPT; break;
                }
                const bool is_ihdr = !memcmp(type, "IHDR", 4);
                if ((nchunk == 0) != is_ihdr) { e = BQIO_ERR_CORRUPT; break; }            // IHDR first, and once
                if (is_ihdr) {
                    if (n != 13) { e = BQIO_ERR_CORRUPT; break; }
                    const uint32_t w = be32(data), h = be32(data + 4);
                    if (data[10] != 0 || data[11] != 0) e = BQIO_ERR_CORRUPT;                // compression / filter method
                    else if (data[8] != 8 || data[9] != 2 || data[12] != 0) e = BQIO_ERR_UNSUPPORTED;   // 8-bit RGB, not interlaced
                    else if ((int)w != tile_px || (int)h != tile_px) e = BQIO_ERR_FORMAT;
                    ihdr = true;
                } else if (!memcmp(type, "IDAT", 4)) {
                    if (idat == 2) { e = BQIO_ERR_CORRUPT; break; }                          // IDAT chunks must be consecutive
                    idat = 1;
                    if (n) pieces[(size_t)i].push_back(Piece{data, n});
                    total += n;
                } else {
                    if (idat == 1) idat = 2;
                    if (!memcmp(type, "IEND", 4)) seen_end = true;
                }
                p += (size_t)n + 12;
            }
            if (e == BQIO_OK && (!ihdr || total == 0 || total > 0x7fffffffull)) e = BQIO_ERR_CORRUPT;
            if (e == BQIO_OK) {
                off[i] = (uint32_t)at;
                len[i] = (uint32_t)total;
                if (loc) { loc[2 * i] = ex.loc_x; loc[2 * i + 1] = ex.loc_y; }
                at = (at + (size_t)total + 32 + 15) & ~(size_t)15;
                if (at > 0xfffffff0ull) e = BQIO_ERR_ARG;
            }
        }
        if (e != BQIO_OK) {
            if (bad_index) *bad_index = first + i;
            r->err = e == BQIO_ERR_UNSUPPORTED ? "image_raw is not an 8-bit RGB PNG the device inflate handles"
                     : e == BQIO_ERR_FORMAT    ? "tile size differs from tile_px" : "corrupt record or PNG";
            return e;
        }
    }
    *used = at;
    if (at > cap) return BQIO_ERR_ARG;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > count) n_threads = (int)(count ? count : 1);
    std::atomic<int64_t> next(0);
    auto work = [&]() {
        for (;;) {
            const int64_t i = next.fetch_add(8);
            if (i >= count) return;
            for (int64_t k = i; k < i + 8 && k < count; ++k) {
                uint8_t* d = out_z + off[k];
                for (const Piece& pc : pieces[(size_t)k]) { memcpy(d, pc.p, pc.n); d += pc.n; }
                const size_t stop = k + 1 < count ? off[k + 1] : at;
                memset(d, 0, (size_t)(out_z + stop - d));
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    return BQIO_OK;
}

int bqio_decode(bqio_reader* r, int64_t first, int64_t count, int tile_px, uint8_t* out, int64_t* loc, int n_threads,
                int64_t* bad_index) {
    return decode_impl(r, first, count, tile_px, out, loc, n_threads, bad_index, false);
}

int bqio_decode_rows(bqio_reader* r, int64_t first, int64_t count, int tile_px, uint8_t* out_rows, int64_t* loc, int n_threads,
                     int64_t* bad_index) {
    return decode_impl(r, first, count, tile_px, out_rows, loc, n_threads, bad_index, true);
}

int bqio_decode_jpeg(const uint8_t* data, size_t len, int tile_px, uint8_t* out) {
    if (!data || !out || tile_px <= 0) return BQIO_ERR_ARG;
    std::unique_ptr<bqjpg::Scratch> S(new bqjpg::Scratch());
    const int j = bqjpg::decode(data, len, tile_px, out, *S);
    return j == bqjpg::OK ? BQIO_OK : j == bqjpg::WRONG_SIZE ? BQIO_ERR_FORMAT : BQIO_ERR_UNSUPPORTED;
}

}  // extern "C"
