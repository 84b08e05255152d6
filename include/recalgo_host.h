/* recalgo_host.h — C-ABI of librecalgo_host.so: native host-side data plumbing in front of the
 * MI355X hot path.  Replaces what the reference obtains from TensorFlow's C++
 * runtime: tf.data.TFRecordDataset + tf.parse_example (/root/reference algorithm/utils.py:18-24,
 * algorithm/DeepFM/deepfm.py:102-117) and the vocabulary-file HashTable behind
 * fc.categorical_column_with_vocabulary_file (deepfm.py:56-64).  Record / Example layout as written
 * by dataset/wechat_algo_data1/DataGenerator.py:390-447.  Host pointers only; no GPU involved.
 */
#ifndef RECALGO_HOST_H_
#define RECALGO_HOST_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RECALGO_HOST_ABI_VERSION 1
int recalgo_host_abi_version(void);

/* CRC-32C (Castagnoli) of a buffer, unmasked. */
uint32_t recalgo_crc32c(const void* data, uint64_t n);

/* Vocabulary file: one key per line, id = 0-based line number (first occurrence wins), absent
 * key -> -1 (tf: num_oov_buckets=0, default_value=-1).  NULL if the file cannot be read. */
void* recalgo_vocab_open(const char* path);
int64_t recalgo_vocab_size(const void* vocab);
int64_t recalgo_vocab_lookup(const void* vocab, const char* key, uint64_t len);
void recalgo_vocab_close(void* vocab);

/* TFRecord reader with a batch buffer.  verify_crc != 0 checks both masked crc32c words. */
void* recalgo_reader_open(const char* path, int verify_crc);
void recalgo_reader_close(void* reader);
int recalgo_reader_rewind(void* reader);
/* dataset.shuffle(buffer_size).repeat(num_epochs), the reference's order (algorithm/utils.py:19-21): the shuffle buffer
 * is drained at every epoch boundary, so records of adjacent epochs never mix.  num_epochs < 0: forever;
 * buffer_size 0: no shuffle; tf.data's buffer semantics, own generator seeded with `seed`.  Call before the first
 * recalgo_reader_next_batch. */
void recalgo_reader_configure(void* reader, int64_t num_epochs, int64_t shuffle_buffer_size, uint64_t seed);
const char* recalgo_reader_error(const void* reader);
/* Read up to max_records records; returns the count (0 = end of file, -1 = error). */
int64_t recalgo_reader_next_batch(void* reader, int64_t max_records);
/* Decode one feature of the current batch (tf.parse_example semantics; SequenceExample
 * feature_lists are not read):
 *   float:  FixedLenFeature((n,), float32, default) -> out [B, n];  -1 if required and missing
 *   id:     VarLenFeature(string) looked up in `vocab` -> offsets [B+1], values [nnz]; returns nnz
 *           (only offsets are written when values_cap < nnz) */
int recalgo_reader_float_feature(void* reader, const char* key, int n, float default_value, int has_default,
                                 float* out);
int64_t recalgo_reader_id_feature(void* reader, const char* key, const void* vocab, int64_t* offsets,
                                  int64_t* values, int64_t values_cap);

/* All single-valued id features of the current batch in one parallel pass: out [B, n_keys] int64 row-major,
 * column f = keys[f] looked up in vocabs[f]; -1 where a record has no value (or the value is not in the
 * vocabulary).  multi[f] is set to 1 if some record holds more than one value for keys[f] (the column then
 * holds the first one: decode that key with recalgo_reader_id_feature instead).  Returns 0, -1 on error. */
int recalgo_reader_id_matrix(void* reader, int n_keys, const char* const* keys, const void* const* vocabs,
                             int64_t* out, int32_t* multi);

/* ---- asynchronous batch pipeline: TFRecordDataset(file)[.shuffle(buffer)].repeat(epochs).batch(B).map(parse_example)
 * .prefetch(depth) (/root/reference algorithm/utils.py:18-24) as ONE object.  A producer thread frames and shuffles the records
 * of batch after batch into a ring of `depth` slots; worker threads decode chunks of 64 records of whichever batches are in the
 * ring (no fork / join per batch).  id_keys[n_ids] / vocabs[n_ids]: single-valued string features -> int64 [B, n_ids] (column
 * order as given, -1 = missing / not in the vocabulary); float_keys[n_floats] with float_n / float_default /
 * float_has_default: FixedLenFeature((n,), float32[, default]) -> float32 [B, sum n].  ids_slots / float_slots: `depth`
 * caller-owned buffers each (e.g. page-locked), or NULL for the pipeline's own.  threads <= 0: RECALGO_READER_THREADS, else half
 * the hardware threads (2 .. 32).  NULL on error. */
void* recalgo_pipeline_open(const char* path, int verify_crc, int64_t num_epochs, int64_t shuffle_buffer_size, uint64_t seed,
                            int64_t batch_size, int n_ids, const char* const* id_keys, const void* const* vocabs,
                            int n_floats, const char* const* float_keys, const int32_t* float_n, const float* float_default,
                            const int32_t* float_has_default, int depth, int64_t* const* ids_slots, float* const* float_slots,
                            int threads);
/* The next batch, in order: its number of records (> 0) and *slot = the ring slot holding it, valid until
 * recalgo_pipeline_release(slot); 0 = end of the data; -1 = framing error / malformed Example, -3 = a required float feature
 * is missing (recalgo_pipeline_error says which); -2 = an id feature holds several values in some record (decode that file with
 * the synchronous accessors). */
int64_t recalgo_pipeline_next(void* pipeline, int* slot);
void recalgo_pipeline_release(void* pipeline, int slot);
const int64_t* recalgo_pipeline_ids(void* pipeline, int slot);
const float* recalgo_pipeline_floats(void* pipeline, int slot);
const char* recalgo_pipeline_error(void* pipeline);
int recalgo_pipeline_threads(void* pipeline);
void recalgo_pipeline_close(void* pipeline);

#ifdef __cplusplus
}
#endif
#endif /* RECALGO_HOST_H_ */
