/* biscuit_io.h -- C ABI of libbiscuit_io.so: host-side reader of Slideflow tile TFRecords
 * (SURVEY.md section 8f row 1), the data format in front of the staging kernel.
 *
 * Replaces, for the hot path's input side, what the reference gets from Slideflow's tf.data
 * pipeline (`Project.evaluate(...)`, experiment.py:917-922; tiles are PNG, 299 px / 302 um,
 * configure.py:118-124): TFRecord framing
 *     uint64 length | uint32 masked_crc32c(length) | bytes[length] | uint32 masked_crc32c(data)
 * a `tf.train.Example` holding `slide` (bytes), `image_raw` (bytes, PNG or JPEG), `loc_x`,
 * `loc_y` (int64), the PNG decode (zlib inflate + scanline unfilter written here; the image
 * has no libpng) and the baseline-JPEG decode (csrc/jpeg_baseline.h: the arithmetic of libjpeg's
 * defaults -- islow IDCT, fancy upsampling -- which is what TensorFlow's decode_jpeg runs, so the
 * bytes are the ones the reference's pipeline saw).  JPEG streams outside that subset are reported,
 * not decoded: the caller hands those bytes to its own decoder.  No TensorFlow, no Slideflow.
 * Plain pointers and sizes only.
 */
#ifndef BISCUIT_IO_H
#define BISCUIT_IO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bqio_reader bqio_reader;

enum { BQIO_OK = 0, BQIO_ERR_ARG = -1, BQIO_ERR_IO = -2, BQIO_ERR_FORMAT = -3, BQIO_ERR_CORRUPT = -4,
       BQIO_ERR_UNSUPPORTED = -5, BQIO_ERR_NAN = -6 };
enum { BQIO_VERIFY_NONE = 0, BQIO_VERIFY_LENGTH = 1, BQIO_VERIFY_FULL = 2 };
enum { BQIO_IMG_UNKNOWN = 0, BQIO_IMG_PNG = 1, BQIO_IMG_JPEG = 2 };

/* Map the file and index its records (checking the CRCs `verify` asks for).  NULL on failure;
 * bqio_last_error(NULL) then says why. */
bqio_reader* bqio_open(const char* path, int verify);
void bqio_close(bqio_reader* r);
const char* bqio_last_error(bqio_reader* r);

/* Number of records (= tiles of the slide). */
int64_t bqio_count(bqio_reader* r);

/* `slide` feature of the first record, NUL-terminated into buf; returns its length or <0. */
int bqio_slide_name(bqio_reader* r, char* buf, int buflen);

/* Format of record `index`'s image_raw payload (BQIO_IMG_*), and a pointer to / length of the
 * payload inside the mapping (valid until bqio_close). */
int bqio_image_format(bqio_reader* r, int64_t index);
int bqio_image_bytes(bqio_reader* r, int64_t index, const uint8_t** data, size_t* len);

/* Decode records [first, first+count): RGB uint8 tiles into out[count][tile_px][tile_px][3],
 * loc_x/loc_y into loc[count][2] (may be NULL), with n_threads worker threads.  PNG (8-bit grey,
 * RGB, palette, RGBA -- alpha dropped; non-interlaced) and baseline JPEG (8-bit, Huffman, one
 * interleaved scan, grey or YCbCr at 4:4:4 / 4:2:2 / 4:2:0, restart markers allowed).  Any other
 * payload -- progressive JPEG, a damaged JPEG stream, ... -- returns BQIO_ERR_UNSUPPORTED and
 * leaves the index of the first such record in *bad_index (may be NULL); a tile of the wrong size
 * returns BQIO_ERR_FORMAT. */
int bqio_decode(bqio_reader* r, int64_t first, int64_t count, int tile_px, uint8_t* out, int64_t* loc,
                int n_threads, int64_t* bad_index);

/* The same records with the PNG scanline filters LEFT IN, for a caller that reverses them on the GPU (bq_png_unfilter,
 * include/biscuit_hip.h): out_rows[count][tile_px][1 + 3*tile_px] -- per row the filter-type byte (0..4) and the filtered RGB
 * bytes, which for an 8-bit RGB non-interlaced PNG is the inflated IDAT stream as it is (a third of a photo-like tile's decode
 * time stays off the host).  Tiles of any other kind -- grey / palette / RGBA PNGs, JPEGs -- are decoded completely here and
 * delivered as rows of filter type 0.  Errors as bqio_decode. */
int bqio_decode_rows(bqio_reader* r, int64_t first, int64_t count, int tile_px, uint8_t* out_rows, int64_t* loc,
                     int n_threads, int64_t* bad_index);

/* Would bqio_decode take records [first, first + count)?  Checks what can be checked without decoding: every record parses,
 * every image is a PNG or a JPEG whose markers and scan structure the baseline decoder accepts (one pass over the bytes of the
 * JPEG records; PNG records are not looked into -- PNG is lossless, any decoder gives the same pixels).  A caller that falls
 * back to another JPEG decoder for what this library refuses can so decide ONCE per slide, before the first chunk, and never
 * mixes two decoders' IDCTs inside a slide.  BQIO_OK, or the error bqio_decode would report, *bad_index = the record. */
int bqio_probe(bqio_reader* r, int64_t first, int64_t count, int tile_px, int64_t* bad_index);

/* The compressed side of the PNG tiles for the DEVICE inflate (libbiscuit_hip: bq_png_inflate): the zlib streams (concatenated
 * IDAT payloads) of records [first, first + count), packed into out_z -- stream i at off[i] (a multiple of 16), len[i] bytes, at
 * least 32 zero bytes behind each -- plus the records' loc_x / loc_y (loc may be NULL).  The host does no decompression: it walks the
 * record framing and the PNG chunk headers and copies.  Only 8-bit RGB non-interlaced tiles of tile_px x tile_px pass
 * (BQIO_ERR_UNSUPPORTED / BQIO_ERR_FORMAT / BQIO_ERR_CORRUPT with *bad_index otherwise: decode that slide with bqio_decode).
 * *used = bytes written; a capacity `cap` that is too small returns BQIO_ERR_ARG with *used = the bytes needed. */
int bqio_extract_z(bqio_reader* r, int64_t first, int64_t count, int tile_px, uint8_t* out_z, size_t cap, uint32_t* off,
                   uint32_t* len, int64_t* loc, size_t* used, int n_threads, int64_t* bad_index);

/* One JPEG file (as bqio_image_bytes returns it) -> out[tile_px][tile_px][3], the decoder
 * bqio_decode uses, exported for tests.  BQIO_OK / BQIO_ERR_UNSUPPORTED / BQIO_ERR_FORMAT. */
int bqio_decode_jpeg(const uint8_t* data, size_t len, int tile_px, uint8_t* out);

/* masked CRC-32C of a buffer (the TFRecord checksum), exported for tests and writers. */
uint32_t bqio_masked_crc32c(const uint8_t* data, size_t len);

/* The reader's own zlib-stream decompressor (csrc/inflate_fast.h), exported for tests: inflates `n` bytes at zdata into
 * exactly out_len bytes at out.  BQIO_OK, or BQIO_ERR_CORRUPT for anything zlib's uncompress() would refuse (bad header,
 * invalid or over-subscribed codes, a distance before the start, wrong length, Adler-32 mismatch, trailing bytes). */
int bqio_inflate(const uint8_t* zdata, size_t n, uint8_t* out, size_t out_len);

/* The same for two streams decoded in one loop (what bqio_decode does with pairs of tiles): *ok_a / *ok_b = 1 where
 * bqio_inflate would have returned BQIO_OK.  Returns BQIO_OK unless an argument is bad. */
int bqio_inflate2(const uint8_t* za, size_t na, uint8_t* out_a, size_t len_a, const uint8_t* zb, size_t nb, uint8_t* out_b,
                  size_t len_b, int* ok_a, int* ok_b);

/* How many PNG streams bqio_decode handed to zlib after the decompressor above refused them and zlib accepted them
 * (process-wide).  Always 0 unless that decompressor has a bug; the tests assert it. */
int64_t bqio_inflate_fallbacks(void);

/* ---- The output side: the tile-prediction table -------------------------------------------------------------------------
 * Replaces the `DataFrame.to_csv(index=False)` with which Slideflow's `Project.evaluate(..., save_predictions=True)`
 * (experiment.py:917-922) leaves `tile_predictions_eval.csv`, the file biscuit reads back with
 * `pd.read_csv(path, dtype={'slide': str})` (experiment.py:688-699; validation: `tile_predictions_val_epoch1.csv`,
 * experiment.py:982-988, utils.py:216) and renames by the column contract of utils.py:19-53.  Rows are appended while the GPU
 * works (one call per run of tiles of one slide), in the bytes pandas would write: float64 cells as the shortest string that
 * reads back to the same double, laid out as Python's repr(float); NaN = empty cell; slide names quoted only when they must be. */
typedef struct bqio_table bqio_table;

/* Create (append = 0: truncate and write the header line
 *     slide[,loc_x,loc_y],{outcome}-y_true0,{outcome}-y_pred0,{outcome}-y_pred1,{outcome}-uncertainty0,{outcome}-uncertainty1)
 * or extend (append = 1: no header) the table at `path`.  NULL on failure; bqio_table_last_error(NULL) says why. */
bqio_table* bqio_table_open(const char* path, const char* outcome, int with_loc, int append);
const char* bqio_table_last_error(bqio_table* t);

/* Append `count` rows of ONE slide: mean2 / std2 = float32 [count][2] (the MC mean and population std of the two class
 * probabilities, widened to float64 exactly as the in-memory table holds them), loc = int64 [count][2] (loc_x, loc_y) exactly
 * when the table was opened with_loc.  A NaN in mean2 returns BQIO_ERR_NAN and writes nothing of this call (threshold.py:141-142
 * refuses such a table: the Python wrapper raises PredsContainNaNError). */
int bqio_table_rows(bqio_table* t, const char* slide, int64_t y_true, const int64_t* loc, const float* mean2, const float* std2,
                    int64_t count);

/* Bytes of the table so far (written or buffered): a rank notes it per slide, so that the per-rank shards of a multi-rank run
 * can be spliced into ONE table in dataset order without parsing them. */
int64_t bqio_table_tell(bqio_table* t);

/* Append bytes [offset, offset + length) of the file at src_path (a slide's rows in another rank's shard). */
int bqio_table_append_file(bqio_table* t, const char* src_path, int64_t offset, int64_t length);

/* Flush and close; *rows / *bytes (may be NULL) = what bqio_table_rows wrote / the file's size increase.  Frees t. */
int bqio_table_close(bqio_table* t, int64_t* rows, int64_t* bytes);

/* repr(float) of one double into out (NUL-terminated), exported for tests.  Returns its length or BQIO_ERR_ARG. */
int bqio_format_f64(double v, char* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
