// libbiscuit_io.so, output side: the tile-prediction table (include/biscuit_io.h, "bqio_table_*").
//
// What leaves the hot path is ONE file, `tile_predictions_eval.csv`, that biscuit reads back with
// `pd.read_csv(path, dtype={'slide': str})` (biscuit/experiment.py:688-699) and renames by the column contract of
// biscuit/utils.py:19-53.  Slideflow writes it with `DataFrame.to_csv(index=False)` after the last batch; here rows are
// appended batch by batch from a host thread while the GPU runs, in the same bytes pandas would write:
//   * header  slide[,loc_x,loc_y],{outcome}-y_true0,{outcome}-y_pred0,{outcome}-y_pred1,{outcome}-uncertainty0,{outcome}-uncertainty1
//   * float64 cells as the SHORTEST decimal string that reads back to the same double (std::to_chars), laid out by the rules of
//     Python's repr(float) -- fixed notation for 1e-4 <= |x| < 1e16, otherwise d[.ddd]e[+-]XX -- which is what pandas emits;
//   * NaN as the empty cell, +-inf as "inf" / "-inf"; a NaN PREDICTION is refused (threshold.py:141-142 would raise on it);
//   * the slide name quoted by csv.QUOTE_MINIMAL's rule (only when it holds , " CR or LF).
// Host code only.
#include "../../include/biscuit_io.h"

#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <string.h>
#include <unistd.h>

#include <charconv>
#include <new>
#include <string>
#include <vector>

struct bqio_table {
    int fd = -1;
    bool with_loc = false;
    int64_t rows = 0, bytes = 0;
    std::vector<char> buf;
    size_t used = 0;
    std::string err;
};

namespace {

std::string g_table_error;
constexpr size_t kFlushAt = 1u << 20;
constexpr size_t kRowMax = 4 * 32 + 3 * 24 + 16;       // four float cells, three integer cells, separators

int flush(bqio_table* t) {
    size_t at = 0;
    while (at < t->used) {
        const ssize_t w = ::write(t->fd, t->buf.data() + at, t->used - at);
        if (w < 0) {
            if (errno == EINTR) continue;
            t->err = std::string("write: ") + strerror(errno);
            return BQIO_ERR_IO;
        }
        at += (size_t)w;
    }
    t->bytes += (int64_t)t->used;
    t->used = 0;
    return BQIO_OK;
}

inline char* put_int(char* p, int64_t v) { return std::to_chars(p, p + 24, v).ptr; }

// repr(float) of a double: shortest round-trip digits (to_chars), Python's choice between fixed and scientific layout
char* put_f64(char* p, double v) {
    if (v != v) return p;                                  // NaN: pandas' na_rep is the empty string
    if (signbit(v)) { *p++ = '-'; v = -v; }
    if (isinf(v)) { memcpy(p, "inf", 3); return p + 3; }
    char s[40];
    char* e = std::to_chars(s, s + sizeof s, v, std::chars_format::scientific).ptr;   // d[.ddd]e[+-]XX, shortest
    char digits[24];
    int nd = 0;
    char* q = s;
    digits[nd++] = *q++;
    if (*q == '.') { ++q; while (*q != 'e') digits[nd++] = *q++; }
    ++q;                                                   // 'e'
    const bool eneg = (*q == '-');
    ++q;
    int ex = 0;
    while (q < e) ex = ex * 10 + (*q++ - '0');
    if (eneg) ex = -ex;
    if (ex >= -4 && ex < 16) {                             // fixed notation, always with a fractional part
        if (ex < 0) {
            *p++ = '0'; *p++ = '.';
            for (int i = 0; i < -ex - 1; ++i) *p++ = '0';
            memcpy(p, digits, nd); p += nd;
        } else {
            const int ip = ex + 1;                         // digits in front of the point
            if (nd <= ip) {
                memcpy(p, digits, nd); p += nd;
                for (int i = nd; i < ip; ++i) *p++ = '0';
                *p++ = '.'; *p++ = '0';
            } else {
                memcpy(p, digits, ip); p += ip;
                *p++ = '.';
                memcpy(p, digits + ip, nd - ip); p += nd - ip;
            }
        }
        return p;
    }
    *p++ = digits[0];
    if (nd > 1) { *p++ = '.'; memcpy(p, digits + 1, nd - 1); p += nd - 1; }
    *p++ = 'e';
    *p++ = ex < 0 ? '-' : '+';
    const int ax = ex < 0 ? -ex : ex;
    if (ax < 10) *p++ = '0';
    return std::to_chars(p, p + 8, ax).ptr;
}

std::string csv_field(const char* s) {
    const std::string v(s ? s : "");
    if (v.find_first_of(",\"\r\n") == std::string::npos) return v;
    std::string o = "\"";
    for (char c : v) { if (c == '"') o += '"'; o += c; }
    return o + "\"";
}

}  // namespace

extern "C" {

const char* bqio_table_last_error(bqio_table* t) { return t ? t->err.c_str() : g_table_error.c_str(); }

bqio_table* bqio_table_open(const char* path, const char* outcome, int with_loc, int append) {
    if (!path || !outcome) { g_table_error = "path / outcome is null"; return nullptr; }
    bqio_table* t = new (std::nothrow) bqio_table();
    if (!t) { g_table_error = "out of memory"; return nullptr; }
    t->fd = ::open(path, O_WRONLY | O_CREAT | (append ? O_APPEND : O_TRUNC) | O_CLOEXEC, 0644);
    if (t->fd < 0) {
        g_table_error = std::string(path) + ": " + strerror(errno);
        delete t;
        return nullptr;
    }
    t->with_loc = with_loc != 0;
    t->buf.resize(kFlushAt + (64u << 10));
    if (!append) {
        const std::string o = outcome;
        std::string h = "slide";
        if (t->with_loc) h += ",loc_x,loc_y";
        // a header that needs quoting is quoted as pandas quotes it
        for (const char* col : {"-y_true0", "-y_pred0", "-y_pred1", "-uncertainty0", "-uncertainty1"})
            h += "," + csv_field((o + col).c_str());
        h += "\n";
        memcpy(t->buf.data(), h.data(), h.size());
        t->used = h.size();
    }
    return t;
}

int bqio_table_rows(bqio_table* t, const char* slide, int64_t y_true, const int64_t* loc, const float* mean2, const float* std2,
                    int64_t count) {
    if (!t || t->fd < 0 || !slide || count < 0 || (count && (!mean2 || !std2))) return BQIO_ERR_ARG;
    if (t->with_loc != (loc != nullptr) && count) {
        t->err = t->with_loc ? "the table has loc_x / loc_y columns but no locations were given"
                             : "locations given to a table without loc_x / loc_y columns";
        return BQIO_ERR_ARG;
    }
    for (int64_t i = 0; i < 2 * count; ++i)
        if (mean2[i] != mean2[i]) {                        // before anything of this call is written
            t->err = "MC-dropout means contain NaN (slide " + std::string(slide) + ", row " + std::to_string(i / 2) + ")";
            return BQIO_ERR_NAN;
        }
    const std::string name = csv_field(slide);
    char yt[24];
    const size_t ytn = (size_t)(put_int(yt, y_true) - yt);
    const size_t row_max = name.size() + kRowMax;
    if (t->buf.size() < kFlushAt + row_max) t->buf.resize(kFlushAt + row_max);
    for (int64_t i = 0; i < count; ++i) {
        char* p = t->buf.data() + t->used;
        memcpy(p, name.data(), name.size()); p += name.size();
        if (loc) {
            *p++ = ','; p = put_int(p, loc[2 * i]);
            *p++ = ','; p = put_int(p, loc[2 * i + 1]);
        }
        *p++ = ','; memcpy(p, yt, ytn); p += ytn;
        *p++ = ','; p = put_f64(p, (double)mean2[2 * i]);
        *p++ = ','; p = put_f64(p, (double)mean2[2 * i + 1]);
        *p++ = ','; p = put_f64(p, (double)std2[2 * i]);
        *p++ = ','; p = put_f64(p, (double)std2[2 * i + 1]);
        *p++ = '\n';
        t->used = (size_t)(p - t->buf.data());
        if (t->used >= kFlushAt) {
            const int e = flush(t);
            if (e) return e;
        }
    }
    t->rows += count;
    return BQIO_OK;
}

int64_t bqio_table_tell(bqio_table* t) { return t ? t->bytes + (int64_t)t->used : -1; }

int bqio_table_append_file(bqio_table* t, const char* src_path, int64_t offset, int64_t length) {
    if (!t || t->fd < 0 || !src_path || offset < 0 || length < 0) return BQIO_ERR_ARG;
    int e = flush(t);
    if (e) return e;
    const int src = ::open(src_path, O_RDONLY | O_CLOEXEC);
    if (src < 0) { t->err = std::string(src_path) + ": " + strerror(errno); return BQIO_ERR_IO; }
    int64_t left = length, at = offset;
    while (left > 0) {
        const size_t want = (size_t)(left < (int64_t)t->buf.size() ? left : (int64_t)t->buf.size());
        const ssize_t r = ::pread(src, t->buf.data(), want, at);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) {
            t->err = std::string(src_path) + (r < 0 ? std::string(": ") + strerror(errno) : ": shorter than its index says");
            ::close(src);
            return r < 0 ? BQIO_ERR_IO : BQIO_ERR_CORRUPT;
        }
        t->used = (size_t)r;
        e = flush(t);
        if (e) { ::close(src); return e; }
        at += r; left -= r;
    }
    ::close(src);
    return BQIO_OK;
}

int bqio_table_close(bqio_table* t, int64_t* rows, int64_t* bytes) {
    if (!t) return BQIO_ERR_ARG;
    int e = t->fd >= 0 ? flush(t) : BQIO_OK;
    if (t->fd >= 0 && ::close(t->fd) != 0 && !e) { g_table_error = std::string("close: ") + strerror(errno); e = BQIO_ERR_IO; }
    else if (e) g_table_error = t->err;
    if (rows) *rows = t->rows;
    if (bytes) *bytes = t->bytes;
    delete t;
    return e;
}

int bqio_format_f64(double v, char* out, int cap) {
    char tmp[48];
    const int n = (int)(put_f64(tmp, v) - tmp);
    if (!out || cap <= n) return BQIO_ERR_ARG;
    memcpy(out, tmp, n);
    out[n] = 0;
    return n;
}

}  // extern "C"
