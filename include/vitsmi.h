/*
 * vitsmi.h — C ABI of libvitsmi.so, the MI355X-native (gfx950) VITS inference engine
 * that replaces the onnxruntime session on phoonnx's hot path.
 *
 * Reference interface replaced (paths relative to the phoonnx checkout):
 *   - session construction      phoonnx/voice.py:167-171  -> vits_open()
 *   - session.get_inputs()      phoonnx/voice.py:347      -> vits_num_inputs()/vits_input_name()
 *   - session.run(None, feed)   phoonnx/voice.py:374-377  -> vits_run()
 * The computation is the graph phoonnx_train/export_onnx.py:250-327 traces from
 * SynthesizerTrn.infer (phoonnx_train/vits/models.py:681-722); weights are read from
 * the same .onnx file onnxruntime would load.
 *
 * Conventions: plain pointers and sizes, no C++ or torch types.  All functions
 * returning int return 0 on success and a negative VITS_E_* code on failure; the
 * message is available from vits_last_error().  A handle owns one HIP stream, the
 * device weight arena and a growable activation workspace; calls on one handle are
 * serialised by an internal mutex, different handles (one per GPU) run concurrently.
 */
#ifndef VITSMI_H
#define VITSMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vits_handle vits_handle;

enum {
    VITS_OK = 0,
    VITS_E_IO = -1,        /* cannot open / read the .onnx file */
    VITS_E_FORMAT = -2,    /* not a VITS graph this engine understands */
    VITS_E_ARG = -3,       /* invalid argument (shape, id out of range, missing sid, ...) */
    VITS_E_DEVICE = -4,    /* HIP error / no gfx950 device */
    VITS_E_NOMEM = -5,
    VITS_E_RANGE = -6      /* f16x3 arithmetic only: an activation left the range of the fp16 operand planes (|x| > 65504,
                              or a non-finite value entered the generator); the engine reports it instead of returning
                              clamped audio.  Reopen with gen_precision "bf16x6" (fp32 range). */
};

/* Stage taps for parity tests ("emb","x","m_p","logs_p","logw","w_ceil","z_p","z"). */
#define VITS_MAX_DIMS 4

/* ---- lifetime ------------------------------------------------------------------- */

/* Parse `onnx_path`, derive the model description, pack the weights into one
 * contiguous device arena on GPU `device_id` (replaces InferenceSession(path, ...),
 * voice.py:167-171). */
int vits_open(const char *onnx_path, int device_id, vits_handle **out);

/* Same, but the packed weight arena is supplied by the caller (already resident on the
 * device, e.g. received by an RCCL broadcast from the rank that read the file, or owned by
 * another handle on the same GPU).  The file is parsed for the model description and the arena
 * LAYOUT only: no weight is packed again.  `arena_dev` must stay valid for the life of the
 * handle.  See vits_arena_* below. */
int vits_open_with_arena(const char *onnx_path, int device_id, void *arena_dev, size_t arena_bytes,
                         vits_handle **out);

/* Host-only open: parses and packs but touches no GPU (device_id ignored).  Only the
 * metadata / arena / hparam calls work on such a handle.  Used by CPU-side tests and
 * by non-root ranks that only need the arena size. */
int vits_open_host(const char *onnx_path, vits_handle **out);

/* Host-only and layout-only: the model description and the arena layout (vits_arena_bytes, vits_hparam,
 * vits_meta) without packing a single weight - exactly what vits_open_with_arena computes before it adopts the
 * caller's device arena.  vits_arena_host() is NULL on such a handle. */
int vits_open_layout(const char *onnx_path, vits_handle **out);

/* Everything above in one call, with the generator's arithmetic chosen explicitly instead of through
 * VITSMI_GEN_PRECISION in the environment. */
typedef struct {
    int device_id;
    const char *gen_precision;  /* NULL / "": environment or default ("f16x3"); "f16x3", "bf16x6", "f16" */
    void *arena_dev;            /* as vits_open_with_arena, or NULL */
    size_t arena_bytes;
    int host_only;              /* as vits_open_host */
    int layout_only;            /* as vits_open_layout (with host_only) */
} vits_open_options;
int vits_open_opts(const char *onnx_path, const vits_open_options *opts, vits_handle **out);

void vits_close(vits_handle *h);

/* Last error message for this handle (or, with h == NULL, of the last failed open on
 * this thread).  Never NULL. */
const char *vits_last_error(vits_handle *h);

/* ---- model description (session.get_inputs(), metadata_props) ---------------------- */

int vits_num_inputs(vits_handle *h);                 /* 3, or 4 with "sid" */
/* "input","input_lengths","scales"[,"sid"]; a third-party graph may also declare "langid" (voice.py:369): it is
 * listed here so that the caller's feed filter (voice.py:373) keeps it, and ignored by the engine (a graph that
 * really consumes a language table is rejected at open). */
const char *vits_input_name(vits_handle *h, int i);

/* metadata_props written by export_onnx.py:335-350 (sample_rate, n_speakers, ...).
 * Returns the value length, or VITS_E_ARG if the key is absent.
 * One key is the loader's own: "vitsmi.name_warnings" - present only when node names and graph structure disagree about
 * which module a Conv belongs to (the names win, the file loads): every such node, one per line. */
int vits_meta(vits_handle *h, const char *key, char *buf, size_t n);

/* Derived hyper-parameters: "hidden","inter","filter","n_heads","n_layers","n_vocab",
 * "n_speakers","gin","use_sdp","hop" (= product of upsample rates),"n_ups","resblock",
 * "gen_sx" (1: the generator runs on the split-operand matrix-core engine, 0: on the f32-MFMA engine),
 * "enc_sx" (1: the text encoder's convs run on the split-operand engine too - f16x3 voices; VITSMI_ENC_ENGINE=f32 keeps
 *   the f32-MFMA engine),
 * "workspace_bytes" (device memory the handle's workspaces hold right now: grows with the largest request, vits_reserve),
 * "gen_rf_frames" (one-sided receptive field of the generator in frames: the context chunked rendering adds),
 * "gen_nprod" (the generator's arithmetic, chosen by VITSMI_GEN_PRECISION in the environment at open time:
 *   2 = "f16x3", the default: fp32 operands as two fp16 planes, three MFMA products per fp32 product, fp32
 *       accumulation; error no larger than the f32-MFMA engine's,
 *   6 = "bf16x6": three bf16 planes, six products, every fp32 product exact to 2^-24,
 *   1 = "f16": BASELINE config 4's reduced-precision vocoder - ONE fp16 plane per operand, one MFMA product per fp32
 *       product, fp32 accumulation, generator activations STORED as fp16 (2 bytes per element instead of 8); everything
 *       in front of z is the default arithmetic.  Range-guarded like f16x3 (VITS_E_RANGE, never clamped audio)). */
int vits_hparam(vits_handle *h, const char *key, int64_t *out);

/* ---- weight arena (multi-GPU: one rank reads + packs, RCCL broadcasts the bytes) ---- */

size_t vits_arena_bytes(vits_handle *h);
/* Host copy of the packed arena (valid until vits_close); NULL for a handle opened with
 * vits_open_with_arena, which never materialises one. */
const void *vits_arena_host(vits_handle *h);
/* Device copy (NULL for a host-only handle). */
void *vits_arena_device(vits_handle *h);

/* ---- the hot call ------------------------------------------------------------------ */

typedef struct {
    /* Optional injected noise for parity with the reference graph's two RandomNormalLike
     * nodes (models.py:111 and :718).  NULL -> generated on the device (Philox) from
     * `seed`.  With scales[2]==0 / scales[0]==0 the respective noise is not used. */
    const float *noise_dp;   /* [B, 2, T] host (vits_run) or device (vits_run_device) */
    const float *noise_z;    /* [B, inter, noise_z_stride] */
    int64_t noise_z_stride;  /* frames per row in noise_z; must be >= max frames */
    uint64_t seed;
} vits_noise;

typedef struct {
    float *data;             /* [B,1,1,S] float32, C-contiguous */
    int64_t dims[4];
    int64_t *y_lengths;      /* [B] frames per utterance (valid samples = y_lengths*hop) */
} vits_output;

/* Host buffers in, host buffers out (what session.run does, voice.py:374).
 *   ids   int64 [B,T]   "input"          (voice.py:350)
 *   lens  int64 [B]     "input_lengths"  (voice.py:351)
 *   scales float32 [3]  [noise_scale, length_scale, noise_w]  (voice.py:364-367)
 *   sid   int64 [B] or NULL              (voice.py:370)
 * `out->data` / `out->y_lengths` are allocated by the library (pinned host memory);
 * release with vits_free_output().  `out` may be NULL: the batch is rendered and kept on the device, to be
 * fetched as 16-bit PCM with vits_last_pcm16() (no fp32 copy to the host at all) together with
 * vits_last_y_lengths(). */
int vits_run(vits_handle *h, const int64_t *ids, const int64_t *lens, int B, int T, const float scales[3],
             const int64_t *sid, const vits_noise *noise, vits_output *out);
void vits_free_output(vits_handle *h, vits_output *out);

/* vits_run without the wait at the end and without an output: host buffers in, the whole path enqueued on the
 * handle's stream.  Returns once the input copies and the one mid-pipeline readback (frame counts:
 * vits_last_y_lengths) are done, i.e. while the generator is still rendering; complete it with vits_fetch_output,
 * vits_last_pcm16 or vits_sync (any of them reports VITS_E_RANGE). */
int vits_run_async(vits_handle *h, const int64_t *ids, const int64_t *lens, int B, int T, const float scales[3],
                   const int64_t *sid, const vits_noise *noise);

/* The last run's waveform -> caller-owned host memory (pageable, or pinned memory from vits_host_alloc for full PCIe
 * rate): row b of the [B, S] result goes to dst + b * row_elems (row_elems >= S; columns [S, row_elems) are zeroed),
 * so that several handles - the sub-batches of one batch, rendered concurrently - can deliver into ONE [B,1,1,S_max]
 * array without a gather on the host.  Typical use: vits_run(..., out = NULL), then vits_last_y_lengths() to size the
 * array, then this.  Waits for the run; VITS_E_RANGE as vits_sync. */
int vits_fetch_output(vits_handle *h, float *dst, size_t row_elems, size_t dst_elems);
/* Pinned (page-locked) host memory for vits_fetch_output / input staging; NULL on failure. */
void *vits_host_alloc(size_t bytes);
void vits_host_free(void *p);

/* Device-resident variant: every pointer (ids, lens, sid, noise arrays) is a device
 * pointer on the handle's GPU; `out->data` and `out->y_lengths` are device pointers into
 * the handle's workspace, valid until the next call on this handle.  Returns after the
 * work has been enqueued on the handle's stream and the one mid-pipeline readback
 * (max frame count) has completed; call vits_sync() before reading `out`. */
int vits_run_device(vits_handle *h, const int64_t *ids, const int64_t *lens, int B, int T,
                    const float scales[3], const int64_t *sid, const vits_noise *noise, vits_output *out);
/* Waits for the handle's stream; VITS_E_RANGE if the run it completes left the fp16 planes' range. */
int vits_sync(vits_handle *h);

/* Chunked (streaming) rendering - SURVEY §8 f1; the reference renders a text sentence by sentence and hands each
 * sentence's audio on as soon as it exists (voice.py:261-269); this does the same INSIDE an utterance batch.
 * Encoder, duration predictor and flow run once; the generator (models.py:348-368) then renders `chunk_frames`
 * frames at a time, each chunk together with vits_hparam "gen_rf_frames" frames of context on either side (the
 * generator's receptive field), of which only the interior is kept: every sample is bit-identical to the one an
 * unchunked vits_run returns.  `fn` is called once per chunk, in order, from the calling thread, while the next chunk
 * renders: samples is host memory [B][n_samples] (row b = utterance b, valid during the call), covering samples
 * [first_sample, first_sample + n_samples) of each row of the [B,1,1,total_samples] output; rows shorter than the
 * longest utterance carry the generator's rendering of their padding, as in vits_run.  A non-zero return stops the
 * run early.  Frame counts: vits_last_y_lengths().  The generator workspace is sized by the chunk, not by the
 * utterance. */
typedef int (*vits_chunk_fn)(void *user, const float *samples, int B, int64_t first_sample, int64_t n_samples,
                             int64_t total_samples);
int vits_run_chunked(vits_handle *h, const int64_t *ids, const int64_t *lens, int B, int T, const float scales[3],
                     const int64_t *sid, const vits_noise *noise, int chunk_frames, vits_chunk_fn fn, void *user);
/* ... and for the vocoder-only entry (z as in vits_run_vocoder). */
int vits_run_vocoder_chunked(vits_handle *h, const float *z, int B, int F, const int64_t *sid, int chunk_frames,
                             vits_chunk_fn fn, void *user);

/* Frame counts of the last run, from the host copy made by the mid-pipeline readback
 * (no synchronisation).  Writes min(n, B) values, returns B. */
int vits_last_y_lengths(vits_handle *h, int64_t *buf, int n);

/* synthesize()'s post-processing on the device, for the LAST run (phoonnx/voice.py:271-282 and AudioChunk,
 * voice.py:88-91): per utterance, over its y_lengths*hop valid samples: peak-normalise (if `normalize`),
 * scale by `volume`, clip, convert to int16 exactly as the NumPy code does.  `out` is host int16 [B, S];
 * samples past an utterance's length are 0. */
int vits_last_pcm16(vits_handle *h, int normalize, float volume, int16_t *out, size_t out_elems);

/* Vocoder only (BASELINE config 2, and teacher-forced parity): z is [B, inter, F] host
 * float32, already masked; output as vits_run. */
int vits_run_vocoder(vits_handle *h, const float *z, int B, int F, const int64_t *sid, vits_output *out);

/* Copy a stage tensor of the LAST run to host: name in {"emb","x","m_p","logs_p","logw",
 * "w_ceil","z_p","z"}; "emb" is emb[ids] * sqrt(hidden) * mask as [B,hidden,T] (models.py:199): the integer
 * gather, bit-exact.  dims receives the shape (rank returned). buf may be NULL to
 * query the shape only. */
int vits_tap(vits_handle *h, const char *name, float *buf, size_t buf_elems, int64_t dims[VITS_MAX_DIMS]);

/* ---- measurement ------------------------------------------------------------------- */

typedef struct {
    double conv_flops;      /* algorithmic FLOPs issued through the conv engine in the last run */
    double conv_bytes;      /* layer-granular bytes (each conv reads input once, writes output once) */
    double dec_flops, dec_bytes;   /* same, HiFi-GAN generator only */
    double flow_flops, enc_flops, dp_flops;
    float conv_ms;          /* HIP-event time of all conv-engine launches (needs timing enabled) */
    float dec_ms, flow_ms, enc_ms, dp_ms, total_ms;
    int conv_launches, total_launches;
    /* the subset of the conv launches that ran on the split-exact bf16 engine (the generator's convs) */
    double sx_flops;
    float sx_ms;
    int sx_launches;
    /* f16x3 range guard (valid after vits_sync / vits_get_stats): every launch that splits fp32 values into fp16
     * operand planes records the largest magnitude it split.  f16_peak_max: the largest over the run (above 65504 =
     * clamped -> f16_saturated = 1 and VITS_E_RANGE); f16_peak_min: the smallest per-launch peak (a whole tensor below
     * ~2^-12 would lose relative precision: planes resolve 2^-36 absolute); f16_tracked: launches recorded. */
    float f16_peak_max, f16_peak_min;
    int f16_tracked, f16_saturated;
    double sx_bytes;        /* layer-granular bytes of the split-engine launches (as conv_bytes) */
} vits_stats;

/* Enable HIP-event timing on the handle's stream: 1 = stage marks plus events around every conv launch (vits_get_stats
 * conv_ms / sx_ms, vits_launch_records; the event records serialise the launches a little), 2 = stage marks only
 * (enc_ms .. total_ms), 0 = off. */
int vits_set_timing(vits_handle *h, int enable);

/* Padded batches (B > 1 with unequal frame counts; the reference itself only ever runs B = 1, voice.py:350-351).  The
 * exported graph does not mask its generator (models.py:348-368, 720): it renders every utterance to the longest one's
 * length, and the samples behind utterance b's end (y_lengths[b] * hop) are the generator's response to zeros.
 *   reference = 0 (default): those samples are NOT rendered.  Every generator launch ends utterance b's tensors
 *       "gen_rf_frames" behind y_lengths[b], so each VALID sample is bit-identical to the padded rendering, and the
 *       output holds 0.0 from y_lengths[b] * hop on.
 *   reference = 1 (or VITSMI_TAILS=reference in the environment at open time): the graph's padded rendering, tails
 *       included - what onnxruntime returns for the same padded feed.
 * Applies to the following runs of this handle (whole and chunked). */
int vits_set_tails(vits_handle *h, int reference);

/* Size the handle's device workspaces NOW for requests of up to B utterances x T tokens that render up to F frames each
 * (the batch's longest utterance; T = 0 or F = 0 leaves that domain alone).  A run grows a workspace when a request
 * needs more than any before it - hipFree + hipMalloc of tens of GB at batch 32, a device-wide synchronisation that was
 * measured at 0.3 ms to 5 s - so a serving process calls this once at start-up with the largest request it admits (the
 * frame count of a batch depends on the durations the model predicts, i.e. on the noise as well: leave headroom), and no
 * request up to that size allocates device memory afterwards (vits_last_pcm16's int16 staging of such a request
 * included).  onnxruntime has no counterpart (its arena grows the same way, voice.py:167-171 passes default
 * SessionOptions); nothing in the reference needs to call it.
 * INVALIDATES THE LAST RUN'S RESULTS when a workspace actually grows: the device waveform, frame counts and taps of the
 * last run live in those workspaces, so after a growing vits_reserve (or any run that grows one) vits_fetch_output /
 * vits_last_pcm16 / vits_tap return VITS_E_ARG ("no completed run") until the next run, and an out->data / out->y_lengths
 * pointer still held from vits_run_device / vits_run_async must not be read any more.  Fetch first, reserve afterwards. */
int vits_reserve(vits_handle *h, int B, int T, int F);
int vits_get_stats(vits_handle *h, vits_stats *out);

/* One record per conv-engine launch of the last run made with timing enabled, in launch order (call after
 * vits_get_stats, which reads the events): the kernel instantiation as rocprofv3 spells it (without "void vitsmi::"
 * and the argument list), its HIP-event time, algorithmic FLOPs and layer-granular bytes (fp32 input read once +
 * output written once), pipeline stage (0 encoder, 1 duration predictor, 2 flow, 3 generator), shape.  Writes min(n, count)
 * records, returns count. */
typedef struct {
    char kernel[128];
    double flops, bytes;
    float ms;
    int stage;
    int cin, cout, k, dil, t;   /* the conv's shape (a fused pair: its first conv) and input length */
} vits_launch_record;
int vits_launch_records(vits_handle *h, vits_launch_record *buf, int n);

/* The HIP stream the handle launches on (hipStream_t as void*), for callers that want
 * to order their own work or events against it. */
void *vits_stream(vits_handle *h);

/* ---- kernel-level test hooks (used by tests/ to localise parity failures) ------------ */

/* out[B,Cout,T] = conv1d(x[B,Cin,T], w[Cout,Cin,K], bias) through the MFMA conv engine,
 * same padding pad_l/pad_r with pad_l + pad_r == dil*(K-1); flags: bit0 leaky-relu(slope)
 * on the input, bit1 relu on the output.  Host pointers. */
int vits_test_conv1d(int device_id, const float *x, int B, int Cin, int T, const float *w, const float *bias,
                     int Cout, int K, int dil, int pad_l, int flags, float slope, float *out);
/* Kernel tuning: average launch time (ms) of one same-padded conv shape on random data through the
 * engine; ms_out[0] = ms, [1] = tile config used, [2] = channel chunk.  cfg/ck_override < 0 = automatic. */
int vits_bench_conv1d(int device_id, int B, int Cin, int Cout, int T, int K, int dil, int hint, int iters,
                      int cfg_override, int ck_override, float *ms_out);
/* out[B,Cout,T*stride] = conv_transpose1d(x, w[Cin,Cout,K], bias, stride, pad=(K-stride)/2). */
int vits_test_conv_transpose1d(int device_id, const float *x, int B, int Cin, int T, const float *w,
                               const float *bias, int Cout, int K, int stride, float *out);
/* The same three hooks through the split-operand engine the generator runs on when all of its channel counts
 * are multiples of 32 (csrc/conv_sx_engine.hip.hpp); needs Cin % 16 == 0 and Cout % 32 == 0.
 * vits_test_conv1d_sx flags: bit0 -> out = leaky_relu(conv, slope) read back from the 16-bit output planes
 * (else the fp32 raw output), bit2 -> residual epilogue with res = x (Cin == Cout), bit3 -> leaky_relu(slope) on the
 * input (raw-input kernels, Cin <= 64; with arithmetic 2: the input plane holds leaky_relu(x) and the residual is
 * recovered from it), bits 4-5 -> arithmetic: 0 bf16x6 (six exact bf16 plane products), 2 f16 (one fp16 plane, one
 * product: the reduced-precision vocoder), 3 f16x3 (two fp16 planes, three products: the generator's default,
 * VITSMI_GEN_PRECISION), bit7 -> the planes are the only output (needs bit0).  vits_test_conv_transpose1d_sx: a negative
 * stride selects f16x3.
 * vits_bench_conv1d_sx dbg bits: 1 no DMA after the first step, 2 no epilogue, 8 residual epilogue, 16 in-kernel
 * cycle breakdown (128-row tiles only), 64 f16 (single plane), 128 f16x3, 256 the v_mfma_f32_32x32x16 main loop even where
 * the 16x16x32 one applies.  ms_out holds 8 floats: [0] ms per launch, [1] tile config, [3..7] with
 * bit 16: s_memtime ticks per pipeline step spent in {LDS wait, DMA wait, barrier, DMA issue, loads + MFMA}. */
int vits_test_conv1d_sx(int device_id, const float *x, int B, int Cin, int T, const float *w, const float *bias,
                        int Cout, int K, int dil, int pad_l, int flags, float slope, float *out);
/* The planar epilogue of the split-operand engine (f16x3 arithmetic, 16x16x32 loop; "same" padding), as the flow's
 * res_skip / pre / post convs and the text encoder's convs use it: o = old + act(conv(x) + bias) * mask, rows
 * [0, row_split) into a first planar tensor, the rest into a second.  flags: 1 ReLU, 2 mask (t < lens[b]), 4 residual
 * (old [B][row_split][T] is added to the first tensor's rows), 8 accumulate (old [B][Cout][T] = the outputs' previous
 * contents), 16 coupling update o = (old - value * mask) * mask (with 8), 32 the second tensor's rows are stored, not
 * accumulated (with 8), 64 the operand planes are those of the second tensor's rows.  out: [B][Cout][T]; planes_out
 * (nullable): [B][pl_rows][T] as read back from the fp16 operand planes the epilogue wrote. */
int vits_test_conv1d_sx_planar(int device_id, const float *x, int B, int Cin, int T, const float *w, const float *bias,
                               int Cout, int K, int dil, int flags, const int64_t *lens, const float *old, int row_split,
                               int pl_rows, float *out, float *planes_out);
/* A/B hook: largest conv launch (workgroups of csrc/conv_sx_small.hip.hpp's kernel) that takes the short-launch kernel;
 * 0 = never (process-wide; default 1536 or VITSMI_SX_SMALL_MAX).  Returns the previous value. */
long long vits_test_set_sx_small_max(long long wgs);
/* The WN in-layer conv with the gate epilogue (tanh(a + g_a) * sigmoid(b + g_b)): w / bias in the packed row order (32 tanh
 * rows, their 32 sigmoid partners, ...), bias_b [B][Cout] in the module's order; flags bit 0: the short-launch kernel, bit 1:
 * the result through the fp16 operand planes.  out: [B][Cout / 2][T]. */
int vits_test_conv1d_sx_gate(int device_id, const float *x, int B, int Cin, int T, const float *w, const float *bias,
                             const float *bias_b, int Cout, int K, int dil, int flags, float *out);
int vits_test_conv_transpose1d_sx(int device_id, const float *x, int B, int Cin, int T, const float *w,
                                  const float *bias, int Cout, int K, int stride, float *out);
int vits_bench_conv1d_sx(int device_id, int B, int Cin, int Cout, int T, int K, int dil, int dbg, int iters,
                         float *ms_out);
/* Two dependent convs of a ResBlock in one fused launch on a raw-format stage (csrc/conv_sx_pair.hip.hpp), C in {32, 64},
 * f16x3 arithmetic, c1 = Conv1d(C, C, K, dilation dil1), c2 = Conv1d(C, C, K, dilation dil2), both "same"-padded:
 *   chain == 0 (ResBlock1 step):   out = c2(leaky_relu(c1(leaky_relu(x, slope)), slope)) + x
 *   chain != 0 (two ResBlock2 steps): x1 = c1(leaky_relu(x, slope)) + x ; out = c2(leaky_relu(x1, slope)) + x1
 * ms_out (nullable): average launch time over 10 launches. */
int vits_test_conv_pair_sx(int device_id, const float *x, int B, int C, int T, const float *w1, const float *b1,
                           const float *w2, const float *b2, int K, int dil1, int dil2, int chain, float slope, float *out,
                           float *ms_out);
/* Relative-position multi-head self-attention core (attentions.py:225-272): q,k,v
 * [B,C,T] host, emb_rel_k/v [2w+1, dk], lens int64[B]; out [B,C,T]. */
int vits_test_attention(int device_id, const float *qkv, int B, int C, int T, int n_heads, const float *rel_k,
                        const float *rel_v, int window, const int64_t *lens, float *out);
/* ... the f16x3 16x16x32 kernel (kernel = 1; head width 32 / 64 / 96) or the fp32-MFMA one (kernel = 0): out_planes (may be
 * NULL) receives the output's fp16 operand planes [B][3][C/8][T][8]; reps > 0 times reps launches into ms_out[0] (ms each). */
int vits_test_attention16(int device_id, const float *qkv, int B, int C, int T, int n_heads, const float *rel_k,
                          const float *rel_v, int window, const int64_t *lens, float *out, uint16_t *out_planes, int kernel,
                          int reps, float *ms_out);

#ifdef __cplusplus
}
#endif
#endif /* VITSMI_H */
