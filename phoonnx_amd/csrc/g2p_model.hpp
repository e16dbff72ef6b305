// g2p_model.hpp — byte-level T5 (ByT5) G2P model description derived from the exported .onnx + packed weight arena
// (reference call site phoonnx/phonemizers/mul.py:106, 192-230).  Host-side C++17, no HIP types.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "model.hpp"

namespace vitsmi {

// A Linear layer without bias, row-major W[out][in]: what both the short-sequence kernel (encoder, prefixes) and the
// matrix-vector kernel of the greedy loop's one-column steps stream.
struct T5Linear {
    int64_t rowmajor = -1;  // arena offset of W[out][in]
    int in = 0, out = 0;
};

struct T5AttnDesc {
    T5Linear q, k, v, o;
};

struct T5FfnDesc {
    bool gated = true;    // T5DenseGatedActDense (wi_0, wi_1) vs T5DenseActDense (wi)
    T5Linear wi0, wi1, wo;
};

struct T5BlockDesc {
    int64_t ln_self = -1, ln_cross = -1, ln_ffn = -1;  // RMS-norm weights [d_model]
    T5AttnDesc self, cross;                            // cross: decoder only
    T5FfnDesc ffn;
};

struct G2PModel {
    int vocab = 0, d_model = 0, heads = 0, d_kv = 0, inner = 0, d_ff = 0, num_buckets = 0, max_distance = 128;
    int act = 0;             // feed-forward activation: 0 gelu_new (tanh form: "gated-gelu"), 1 relu, 2 gelu (erf)
    bool scale_out = false;  // decoder output * d_model^-0.5 before lm_head (tied-embedding checkpoints)
    float eps = 1e-6f;
    int64_t shared = -1;                 // embedding table [vocab][d_model]
    int64_t enc_bias = -1, dec_bias = -1;  // relative_attention_bias [num_buckets][heads] of block 0 of each stack
    int64_t enc_final_ln = -1, dec_final_ln = -1;
    std::vector<T5BlockDesc> enc, dec;
    T5Linear lm_head;                    // d_model -> vocab
    std::vector<float> arena;
    int64_t arena_floats = 0;
    int64_t zeros_off = 0;
    std::vector<std::string> input_names, output_names;
    // relative-position bucket of (memory - context) = d, d in [-(kMaxPos-1), kMaxPos-1], float32 arithmetic of
    // modeling_t5.py `_relative_position_bucket`: bucket_enc[d + kMaxPos - 1] (bidirectional), bucket_dec (causal)
    static constexpr int kMaxPos = 1024;
    std::vector<int> bucket_enc, bucket_dec;

    std::string build(const OnnxModel &om);
};

}  // namespace vitsmi
