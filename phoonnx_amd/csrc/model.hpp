// model.hpp — VITS model description derived from the .onnx graph + packed weight arena.
// Host-side C++17; no HIP types here so CPU-only tests can exercise it.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "onnx_reader.hpp"

namespace vitsmi {

// One dense conv as executed by the MFMA conv engine (conv_engine.hip).
// Weights are packed as Wp[mblock][step/4][lane 0..63][4] with
//   step = (chunk*K + tap)*(CK/2) + pair,
//   value = W[co = mblock*32 + (lane&31)][ci = chunk*CK + 2*pair + (lane>>5)][tap]  (0 if out of range)
// i.e. exactly the A-operand lane layout of v_mfma_f32_32x32x2_f32, four k-steps per 16-byte load.
struct ConvDesc {
    int64_t w_off = -1;   // arena offset (floats) of packed weights
    int64_t b_off = -1;   // arena offset of bias [Cout] (virtual channels), -1 if none
    int Cin = 0, Cout = 0;  // Cout = virtual output channels (real Cout * ups for transposed conv)
    int K = 1, dil = 1, padL = 0;
    int CK = 8, nchunks = 0;
    int mblocks = 0;      // packed 32-row blocks (padded to the tile config)
    int ups = 1;          // pixel-shuffle factor (transposed conv), real Cout = Cout/ups
    int cfg = 0;          // tile: 0: 32x512, 1: 64x256, 2: 128x128, 3: 64x64, 4: 32x128
    // sx = packed for the split-operand engine (conv_sx_engine.hip.hpp) instead: weights as three bf16
    // planes [m-tile][chunk of 16 ci][tap][32-row block][plane][lane][8]; cfg then indexes the sx tiles
    // (0: 128x256, 1: 64x256, 2: 32x256) and a transposed conv's virtual rows are r-major (r*Cr + co).
    bool sx = false;
    // sx only: the input tensor is in the fp32 raw layout and is split into planes inside the kernel (tensors of
    // <= 64 channels, see sx_raw_format); such convs use the 64- or 32-row tiles
    bool rawin = false;
    // sx only: weights are TWO fp16 planes (per 32-row block) of g = w * 2^k, k per tensor: max |g| in [2^14, 2^15);
    // wscale = 2^-k is applied to the accumulators (conv_sx_engine.hip.hpp, f16 mode)
    bool f16 = false;
    float wscale = 1.f;
    // sx + f16 only: packed for the v_mfma_f32_16x16x32_f16 main loop: chunks of 32 input channels, per 32-row block and
    // step [16-row sub-block a][plane][lane][8] with lane = (row c = lane & 15, channel group g = lane >> 4) and the rows
    // of a sub-block permuted (row 4 q + r of sub-block a = row r + 8 (2 a + (q & 1)) + 4 (q >> 1) of the 32-row block)
    // so that the accumulators reach the epilogue's 32 x 32 layout with one half-row swap (conv_sx_engine.hip.hpp)
    bool s16 = false;
    // sx only: ONE scaled fp16 plane (s16 layout, 2 KiB per 32-row block and step), one MFMA product per fp32 product; the
    // conv's tensors are single fp16 planes holding the consumer's leaky-ReLU (VITSMI_GEN_PRECISION=f16, BASELINE config 4)
    bool h1 = false;
    // sx only (flow WN in-layers): rows packed as [32 tanh | 32 sigmoid] per 64-row tile, the conv's epilogue applies
    // the gate and writes planar acts (conv_sx_engine.hip.hpp SX_GATE)
    bool gate = false;
    // sx transposed convs of kernel 2 * ups (dense 3-tap form): the padding of the transposed conv - which of the three taps
    // is all zeros follows from it per output phase (SxArgs::zt_p); -1: not such a conv
    int zt_p = -1;
    double macs_per_t = 0;   // algorithmic MACs per input time step (reference definition)
    bool valid() const { return w_off >= 0; }
};

struct DDSDesc {  // modules.py:81-129
    int n_layers = 0, K = 3;
    struct L {
        int64_t dw_w = -1, dw_b = -1;  // depthwise [C,K], [C]
        int dil = 1;
        int64_t ln1_g = -1, ln1_b = -1, ln2_g = -1, ln2_b = -1;
        ConvDesc pw;  // 1x1
        int64_t pw16 = -1;  // the same weights as the A operand of v_mfma_f32_16x16x4_f32 (dds_layer16_kernel): [C/16][4][16][C/4]
    } l[4];
};

struct ConvFlowDesc {  // modules.py:469-527
    int64_t pre_w = -1, pre_b = -1;  // [C], [C]
    DDSDesc convs;
    ConvDesc proj;  // C -> 3*nb-1
    int64_t proj16 = -1;  // ... once more in dds_layer16_kernel's A-operand layout (rows padded to 16): the layer's fused tail
    int nb = 10;
};

struct EncLayerDesc {
    ConvDesc qkv, o, ffn1, ffn2;
    // the same four convs packed for the split-operand engine (16x16x32 loop, planar epilogue): read when Model::enc_sx
    ConvDesc qkv_sx, o_sx, ffn1_sx, ffn2_sx;
    int64_t rel_k = -1, rel_v = -1;  // [2w+1, dk]
    int64_t ln1_g = -1, ln1_b = -1, ln2_g = -1, ln2_b = -1;
};

struct CouplingDesc {  // one ResidualCouplingLayer with the preceding Flip folded in
    ConvDesc pre, post;
    // pre / post once more for the split-operand engine (planar epilogue; taken when every WN layer of the coupling runs
    // there): pre_sx.sx says whether they exist
    ConvDesc pre_sx, post_sx;
    int n_wn = 0;
    struct {
        ConvDesc in, rs;
        // the res_skip conv once more for the split-operand engine (16x16x32 loop, K = 1; used when the in-layer is gated
        // on that engine and hands its acts over as operand planes): rs_sx.sx says whether it exists
        ConvDesc rs_sx;
    } wn[8];
    int64_t cond_w = -1, cond_b = -1;  // [2*H*n_wn, gin]
    bool swapped = false;  // true: x0 is the physical upper half, x1 the lower
};

struct ResBlockDesc {
    int n = 0;          // pairs (ResBlock1) or single convs (ResBlock2)
    bool type1 = true;
    ConvDesc c1[4], c2[4];
};

struct UpStageDesc {
    ConvDesc up;
    int u = 1, C = 0;
    std::vector<ResBlockDesc> rbs;
};

struct Model {
    // ---- hyper-parameters recovered from the graph
    int n_vocab = 0, H = 0, C = 0, FF = 0, n_heads = 0, dk = 0, n_layers = 0, window = 0;
    int n_speakers = 1, gin = 0;
    bool use_sdp = true;
    int hop = 1;  // product of upsample rates
    std::vector<std::string> input_names;
    std::map<std::string, std::string> meta;

    // ---- text encoder
    int64_t emb = -1;
    std::vector<EncLayerDesc> enc;
    ConvDesc enc_proj;
    ConvDesc enc_proj_sx;  // (split-operand engine, see EncLayerDesc)
    bool enc_sx = false;   // the encoder's convs run on the split-operand engine (f16x3; VITSMI_ENC_ENGINE=f32 keeps the f32 engine)

    // ---- speaker conditioning
    int64_t emb_g = -1;
    int64_t dp_cond_w = -1, dp_cond_b = -1;    // [Cdp_in, gin]
    int dp_cond_rows = 0;
    int64_t dec_cond_w = -1, dec_cond_b = -1;  // [C0, gin]

    // ---- stochastic duration predictor (reverse)
    ConvDesc dp_pre, dp_proj;
    int64_t dp_proj16 = -1;  // dp_proj in dds_layer16_kernel's A-operand layout (the fused tail of the last DDSConv layer)
    DDSDesc dp_convs;
    ConvFlowDesc cf[3];  // execution order: flows.7, flows.5, flows.3
    float ea_m0 = 0.f, ea_logs0 = 0.f;
    // ---- plain duration predictor
    ConvDesc dpp_conv1, dpp_conv2, dpp_proj;
    int64_t dpp_n1_g = -1, dpp_n1_b = -1, dpp_n2_g = -1, dpp_n2_b = -1;
    int dpp_F = 0;

    // ---- flow (execution order: flows.6, .4, .2, .0)
    std::vector<CouplingDesc> flow;
    int flow_H = 0;

    // ---- generator
    ConvDesc conv_pre;
    int C0 = 0;
    bool gen_f16 = false; // ... in its fp16 two-plane mode (VITSMI_GEN_PRECISION=f16x3)
    bool gen_planes = false;  // gen_f16 || gen_h1: the plane-stream generator (every inter-conv tensor stored once, as operand planes)
    bool gen_h1 = false;  // ... in its fp16 single-plane, single-product mode (VITSMI_GEN_PRECISION=f16; 16-bit activations)
    bool gen_sx = false;  // generator packed for the split-operand engine (all channel counts % 32 == 0)
    std::vector<UpStageDesc> ups;
    int64_t post_w = -1;  // [Cin, K] conv_post weight (Cout = 1, no bias)
    int post_cin = 0, post_k = 7;
    int gen_rf_frames = 0;  // one-sided receptive field of the whole generator, in input frames (rounded up, + 1)
    std::vector<int> gen_rf_stage;  // ... of what follows the INPUT of upsampling stage s (its transposed conv included)

    // ---- packed arena (host copy; empty after a layout-only build)
    std::vector<float> arena;
    int64_t arena_floats = 0;  // size of the packed arena, materialised or not
    int64_t zeros_off = 0;  // >= 1024 zero floats

    // Build from a parsed file.  Returns "" or an error message.  layout_only: compute every offset and descriptor
    // but materialise no weight (the packed bytes already live on the device: vits_open_with_arena).
    std::string build(const OnnxModel &om, bool layout_only = false);

    // reference-definition work per unit: MACs per frame / per token
    double dec_macs_per_frame = 0, flow_macs_per_frame = 0, enc_macs_per_token = 0, dp_macs_per_token = 0;
    double dec_elems_per_frame = 0;  // conv input + output elements per frame (layer-granular bytes / 4)
};

// kernel-level test hooks: pack one conv / transposed conv into a private arena
std::string pack_test_conv(const float *w, const float *bias, int Cin, int Cout, int K, int dil, int pad_l, int hint,
                           ConvDesc *d, std::vector<float> *arena);
void set_tiling_override(int cfg, int ck);  // -1,-1 = automatic (kernel tuning only)
std::string pack_test_convT(const float *w, const float *bias, int Cin, int Cout, int K, int stride, ConvDesc *d,
                            std::vector<float> *arena, bool sx = false);

// fp32 -> bf16 planes of the split-exact engine (host mirror of split3 in conv_sx_engine.hip.hpp)
uint16_t bf16_rne(float f);
float bf16_to_f32(uint16_t h);
void split3_host(float v, uint16_t p[3]);
// ... and of the f16 mode: fp16 round-to-nearest-even (clamped to +-65504), two planes (p[2] = 0)
uint16_t f16_rne(float f);
float f16_to_f32(uint16_t h);
void split2h_host(float v, uint16_t p[3]);
void set_sx_force16(bool on);  // ... the 16x16x32 layout also for <= 64 input channels (the fused pair kernel's operand)
void set_sx_h1(bool on);   // ... one fp16 plane in the 16x16x32 layout (the NP = 1 mode)
void set_sx_f16(bool on);  // pack_conv_sx format for the calls that follow on this thread (test hooks)
void set_sx_shape32(bool on);  // ... never the 16x16x32 (s16) packing (bench hooks: ablation flags, A/B of the MFMA shapes)
// generator arithmetic for the Model::build calls that follow on this thread: explicit name, or nullptr = take
// VITSMI_GEN_PRECISION from the environment (default f16x3)
void set_gen_precision_override(const char *name);
const char *gen_precision_name();
// may this conv shape run on the sx engine (channel multiples, LDS budget)?
bool sx_supported(int Cin, int Cout_virtual, int Cr, int K, int dil);
// Storage format of a generator tensor with C channels on the sx path: true = fp32 raw only (its consumers
// split it on the fly; layers this narrow are HBM-bound), false = 16-bit planes (+ raw where it is a residual).
inline bool sx_raw_format(int C) {
    static const int maxc = [] {
        const char *e = std::getenv("VITSMI_SX_RAW_MAXC");  // tuning experiments only
        return e ? std::atoi(e) : 64;
    }();
    return C <= maxc;
}

}  // namespace vitsmi
