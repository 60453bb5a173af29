/* srwn_io.h -- host-side data path of the reference: NSynth TFRecord files (nsynth.py:6-46, written by
 * filter_tfrecord.py:40-58) without TensorFlow.  Plain C ABI, host pointers only, no GPU involved; implemented in
 * sr-wavenet_amd/csrc/srwn_tfrecord.cpp -> sr-wavenet_amd/libsrwn_io.so.
 *
 * Replaces: tf.data.TFRecordDataset(filepath) + tf.parse_single_example(example, features) (nsynth.py:9-37):
 *   TFRecord framing  u64 length | u32 masked_crc32c(length) | payload | u32 masked_crc32c(payload)
 *   payload           tf.train.Example protobuf; the 13 NSynth features of nsynth.py:10-25 are float_list "audio",
 *                     int64_list "pitch" / "velocity" / "instrument" / ... , bytes_list "note_str" / ...
 * All functions return 0 on success or a negative SRWN_IO_E_* code; srwn_io_last_error() gives the message
 * (thread-local). */
#ifndef SRWN_IO_H
#define SRWN_IO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SRWN_IO_E_ARG    (-1)
#define SRWN_IO_E_IO     (-2)
#define SRWN_IO_E_FORMAT (-3)   /* truncated framing */
#define SRWN_IO_E_CRC    (-4)   /* tf.errors.DataLossError in the reference */
#define SRWN_IO_E_PARSE  (-5)   /* malformed protobuf */
#define SRWN_IO_E_NOKEY  (-6)   /* feature missing (tf.parse_single_example raises for a FixedLenFeature without default) */
#define SRWN_IO_E_TYPE   (-7)   /* feature holds another list kind */
#define SRWN_IO_E_SHAPE  (-8)   /* FixedLenFeature length mismatch */

const char* srwn_io_last_error(void);
/* CRC-32C (Castagnoli) of a host buffer and TensorFlow's mask ((crc >> 15 | crc << 17) + 0xa282ead8): TFRecord
 * frames, checkpoint index blocks and tensor-bundle entries store the masked value (tf_checkpoint.py uses these). */
uint32_t srwn_crc32c(const void* data, uint64_t n);
uint32_t srwn_crc32c_mask(uint32_t crc);

/* maps the file and indexes every record; verify_crc != 0 checks both CRCs of every record.  NULL on failure. */
void* srwn_tfr_open(const char* path, int32_t verify_crc);
void srwn_tfr_close(void* handle);
int64_t srwn_tfr_count(void* handle);

/* kind (1 bytes_list, 2 float_list, 3 int64_list, 0 empty) and number of values of feature `key` in record idx */
int srwn_tfr_feature(void* handle, int64_t idx, const char* key, int32_t* kind, int64_t* count);
/* copy up to max_n values; *n_out = number of values the feature holds (bytes: length of the first value) */
int srwn_tfr_read_floats(void* handle, int64_t idx, const char* key, float* out, int64_t max_n, int64_t* n_out);
int srwn_tfr_read_int64s(void* handle, int64_t idx, const char* key, int64_t* out, int64_t max_n, int64_t* n_out);
int srwn_tfr_read_bytes(void* handle, int64_t idx, const char* key, char* out, int64_t max_n, int64_t* n_out);

/* one minibatch (nsynth.py:27-33 "reduced" mode): audio[b, :num_samples] = first num_samples floats of audio_key
 * (which must hold exactly audio_len floats when audio_len > 0: tf.FixedLenFeature([audio_max_length]));
 * label[b] = first value of label_key (may be NULL).  Records are decoded by `nthreads` threads. */
int srwn_tfr_read_batch(void* handle, const int64_t* idx, int32_t B, const char* audio_key, int64_t audio_len,
                        int32_t num_samples, float* audio, const char* label_key, int64_t* label, int32_t nthreads);

#ifdef __cplusplus
}
#endif
#endif /* SRWN_IO_H */
