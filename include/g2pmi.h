/*
 * g2pmi.h — C ABI of the ByT5 G2P engine inside libvitsmi.so: the byte-level T5 encoder-decoder that
 * phoonnx's multilingual phonemizer runs through onnxruntime, as hand-written gfx950 kernels.
 *
 * Reference interface replaced (paths relative to the phoonnx checkout):
 *   - session construction      phoonnx/phonemizers/mul.py:106      -> g2p_open()
 *   - session.get_outputs()     phoonnx/phonemizers/mul.py:183      -> g2p_output_name()
 *   - session.run(names, feed)  phoonnx/phonemizers/mul.py:199-211  -> g2p_run()   (logits of every decoder position)
 *   - the greedy loop around it phoonnx/phonemizers/mul.py:192-230  -> g2p_generate() (one call, KV cache, on the device:
 *     the reference re-runs the whole graph, encoder included, for every generated token)
 * The graph is Hugging Face transformers' T5ForConditionalGeneration exported to ONNX (inputs input_ids, attention_mask,
 * decoder_input_ids; output logits); weights are read from the same .onnx file.
 *
 * Conventions as in vitsmi.h: plain pointers and sizes; int functions return 0 or a negative VITS_E_* code
 * (vitsmi.h), message from g2p_last_error().  g2p_run: batch size 1, as the reference calls it.
 */
#ifndef G2PMI_H
#define G2PMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct g2p_handle g2p_handle;

/* Parse `onnx_path`, derive the model description (d_model, heads, layers, feed-forward type ... from shapes and node
 * names), pack the weights on GPU `device_id`.  device_id < 0: host-only handle (description only; g2p_run fails). */
int g2p_open(const char *onnx_path, int device_id, g2p_handle **out);
void g2p_close(g2p_handle *h);
const char *g2p_last_error(g2p_handle *h);

/* "vocab","d_model","heads","d_kv","d_ff","n_enc","n_dec","num_buckets","max_distance","gated","act","scale_out" */
int g2p_hparam(g2p_handle *h, const char *key, int64_t *out);
int g2p_num_outputs(g2p_handle *h);
const char *g2p_output_name(g2p_handle *h, int i);  /* "logits" */

/* Relative-position bucket of (key position - query position) = rel, as the graph computes it (an integer: exact).
 * decoder != 0: the causal variant. */
int g2p_bucket(g2p_handle *h, int decoder, int rel);

/* What session.run returns (mul.py:210-211): logits [1, T, vocab] (host float32, T * vocab values) for
 * input_ids [1, S] and decoder_input_ids [1, T].  attention_mask may be NULL (all ones, what mul.py:187 builds) or
 * right-padded (ones, then zeros: the padded positions are dropped, which is what masking them amounts to); a mask
 * with holes is rejected. */
int g2p_run(g2p_handle *h, const int64_t *input_ids, int S, const int64_t *attention_mask, const int64_t *decoder_input_ids,
            int T, float *logits);

/* The greedy loop of mul.py:192-230 in one call: encoder once, then one decoder step per token with a key/value cache,
 * argmax on the device; stops at eos_id (included in the output, as the reference appends it before breaking) or after
 * max_length tokens.  out_ids receives at most max_length ids; *n_out their count. */
int g2p_generate(g2p_handle *h, const int64_t *input_ids, int S, int max_length, int64_t start_id, int64_t eos_id,
                 int64_t *out_ids, int *n_out);

/* The same loop for B independent inputs side by side (extension: the reference phonemizes chunk after chunk,
 * base.py:66-70): input_ids holds the B sequences back to back, lens[b] their lengths.  One encoder pass over the padded
 * batch (padding is masked out of every attention), then every decoder step streams the weights once for all B
 * sequences; a sequence that has produced eos_id stops collecting tokens while the others finish.  Each sequence's ids
 * are exactly what g2p_generate returns for it alone.  out_ids: [B][max_length]; n_out: [B].  B <= G2P_MAX_BATCH. */
#define G2P_MAX_BATCH 64
int g2p_generate_batch(g2p_handle *h, const int64_t *input_ids, const int *lens, int B, int max_length, int64_t start_id,
                       int64_t eos_id, int64_t *out_ids, int *n_out);

/* Test hook: the decoder-STEP path (the kernels g2p_generate runs per token: matrix-vector products, one-query attention
 * over the key / value caches) with GIVEN decoder inputs instead of its own argmax fed back, returning the logits of every
 * step: logits[b][t] = what g2p_run returns at position t for decoder_input_ids[b][0 .. t].  B <= 4 inputs back to back in
 * input_ids (lens[b] each), decoder_input_ids [B][T] (column 0 = the start token), logits [B][T][vocab].  The greedy ids of a
 * randomly initialised model are nearly constant sequences; this compares the step path number by number. */
int g2p_test_forced_steps(g2p_handle *h, const int64_t *input_ids, const int *lens, int B, const int64_t *decoder_input_ids, int T,
                          float *logits);

#ifdef __cplusplus
}
#endif
#endif /* G2PMI_H */
