// vitsmi.hip — host side of libvitsmi.so: handle, workspace, the VITS pipeline as a sequence of
// kernel launches on one HIP stream, and the C ABI declared in include/vitsmi.h.
//
// Pipeline = SynthesizerTrn.infer (phoonnx_train/vits/models.py:681-722):
//   text encoder -> (stochastic) duration predictor -> length regulator -> inverse coupling flow
//   -> HiFi-GAN generator.  The only host synchronisation inside a run is the readback of the frame
//   counts (data-dependent output length).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vitsmi.h"
#include "conv_sx_engine.hip.hpp"
#include "conv_sx_pair.hip.hpp"
#include "conv_sx_pair16.hip.hpp"
#include "attention16.hip.hpp"
#include "conv_sx_small.hip.hpp"
#include "kernels.hip.hpp"
#include "model.hpp"

using namespace vitsmi;

namespace {

thread_local std::string g_open_error;

struct Slab {
    char *base = nullptr;
    size_t cap = 0, used = 0;
};

// Pinned host buffer handed out by vits_run()/vits_run_vocoder() and returned by vits_free_output():
// one per handle, reused across calls (hipHostMalloc costs more than a whole B=1 run).
struct PinnedPool {
    char *base = nullptr;
    size_t cap = 0;
    bool busy = false;
};

}  // namespace

struct vits_handle {
    Model model;
    int device = -1;
    bool host_only = false;
    hipStream_t stream = nullptr;
    float *arena_dev = nullptr;
    bool arena_owned = false;
    Slab tok, frm;  // token-domain and frame-domain workspaces
    Slab io;        // device staging of vits_run()'s host inputs
    PinnedPool pin;
    std::mutex mu;
    std::string err;
    // last-run state (for taps / outputs)
    int B = 0, T = 0, F = 0, S = 0;
    int Fpitch = 0;  // row pitch of the frame-domain flow tensors (F rounded up to 4)
    float *d_emb = nullptr, *d_x = nullptr, *d_mp = nullptr, *d_logs = nullptr, *d_logw = nullptr, *d_wceil = nullptr;
    float *d_zp = nullptr, *d_z = nullptr, *d_out = nullptr;
    int *d_len = nullptr, *d_ylen = nullptr, *d_cum = nullptr;
    int64_t *d_ylen64 = nullptr;
    std::vector<int> h_ylen;
    char *h_in_pin = nullptr;      // pinned staging of a call's ids | lens | sid: one host-to-device copy instead of three
    size_t h_in_pin_bytes = 0;     // staged pageable ones
    int *h_ylen_pin = nullptr;     // pinned landing buffer of the one mid-run readback (a pageable destination makes the copy a
    int h_ylen_pin_n = 0;          // staged, synchronous one: the stream then waits for the host twice)
    // stats
    int timing = 0;  // vits_set_timing: 0 off, 1 stage marks + events around every conv launch, 2 stage marks only
    vits_stats stats{};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> conv_events;
    std::vector<char> conv_event_sx;  // 1: that launch went through the split-exact engine
    std::vector<vits_launch_record> conv_recs;  // what each timed launch was (vits_launch_records)
    size_t conv_events_used = 0;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int cur_stage = 0;  // 0 enc, 1 dp, 2 flow, 3 dec
    uint64_t run_counter = 0;
    // Padded batches (B > 1, unequal frame counts).  The reference graph does not mask its generator (models.py:348-368, 720):
    // it renders every utterance to the longest one's length and the samples behind an utterance's end are the generator's
    // response to zeros.  tails_reference = false (default; VITSMI_TAILS=reference or vits_set_tails(h, 1) for the other):
    // those samples are NOT rendered - every generator launch ends utterance b's tensors gen_rf_frames behind y_len[b]
    // (SxRagged), so each valid sample is bit-identical to the padded rendering - and the output holds zeros there.
    bool tails_reference = false;
    int gen_nprod = 2;  // generator arithmetic (VITSMI_GEN_PRECISION): 2 = two fp16 planes / three products (default), 6 = six
                        // exact bf16 plane products, 1 = one fp16 plane / one product with fp16 activations (config 4)
    // f16 range guard: every launch that splits values into fp16 planes publishes the largest magnitude it saw into its
    // own 64 slots of d_range; range_reduce_kernel folds them at the end of a run into d_range_res = {max, min over
    // launches of the per-launch peak, launches tracked}, copied to the pinned h_range (read after the next sync).
    unsigned *d_range = nullptr;
    float *d_range_res = nullptr;
    float *h_range = nullptr;
    int range_launch = 0;
    int range_used = -1;          // slots groups the last run wrote (-1: unknown, clear all)
    int h_range_launches = 0;     // launches behind the values in h_range
    bool range_pending = false;   // h_range holds the result of a run that has not been checked yet
    bool range_failed = false;    // the last run saturated (sticky until the next run starts)
    // chunked rendering: two pinned host buffers the finished chunks are copied into, with their completion events
    char *ring[2] = {nullptr, nullptr};
    size_t ring_cap = 0;
    hipEvent_t ring_ev[2] = {nullptr, nullptr};
};

// chunked rendering (vits_run_chunked / vits_run_vocoder_chunked): where the audio goes
struct ChunkSink {
    int chunk_frames;
    vits_chunk_fn fn;
    void *user;
};

constexpr int kMaxRangeLaunches = 512;

namespace {

int fail(vits_handle *h, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_open_error = buf;
    return code;
}

// this launch's slots of the f16 range guard (nullptr when the arithmetic has no fp16 planes)
unsigned *range_slots(vits_handle *h, bool f16) {
    if (!f16 || !h->d_range) return nullptr;
    const int i = h->range_launch < kMaxRangeLaunches ? h->range_launch : kMaxRangeLaunches - 1;
    h->range_launch++;
    return h->d_range + (size_t)i * kSxPeakSlots * kSxPeakStride;
}

// one wave per launch: peak = max over its 64 slots; out[0] = max over launches, out[1] = min over the launches that
// recorded anything (both as float bit patterns, which order like unsigned integers for non-negative floats; preset
// to 0 / 0xffffffff by range_end)
__global__ void range_reduce_kernel(const unsigned *slots, unsigned *out) {
    unsigned b = slots[((size_t)blockIdx.x * kSxPeakSlots + threadIdx.x) * kSxPeakStride];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o, 64));
    if (threadIdx.x == 0 && b) {  // (an all-zero tensor - e.g. fully masked rows - says nothing about range)
        atomicMax(out, b);
        atomicMin(out + 1, b);
    }
}

void range_begin(vits_handle *h) {
    h->range_launch = 0;
    // (only the slots the previous run used need clearing)
    if (h->d_range)
        hipMemsetAsync(h->d_range, 0, (size_t)(h->range_used > 0 ? h->range_used : kMaxRangeLaunches) * kSxPeakSlots *
                                          kSxPeakStride * sizeof(unsigned), h->stream);
}

// fold the slots and start the 12-byte copy to the host; evaluated by range_check() after the next synchronisation
void range_end(vits_handle *h) {
    if (!h->d_range) return;
    const int n = h->range_launch < kMaxRangeLaunches ? h->range_launch : kMaxRangeLaunches;
    h->range_used = n;
    unsigned *res = reinterpret_cast<unsigned *>(h->d_range_res);
    hipMemsetAsync(res, 0, 4, h->stream);
    hipMemsetAsync(res + 1, 0xff, 4, h->stream);
    if (n > 0) range_reduce_kernel<<<n, kSxPeakSlots, 0, h->stream>>>(h->d_range, res);
    hipMemcpyAsync(h->h_range, h->d_range_res, 2 * sizeof(float), hipMemcpyDeviceToHost, h->stream);
    h->h_range_launches = n;
    h->range_pending = true;
}

// after a synchronisation: did the last run leave the fp16 planes' range?
int range_check(vits_handle *h) {
    if (!h->range_pending) return 0;
    h->range_pending = false;
    unsigned bits[2];
    std::memcpy(bits, h->h_range, sizeof bits);
    float pmax, pmin;
    std::memcpy(&pmax, &bits[0], 4);
    if (bits[1] == 0xffffffffu) pmin = 0.f;  // nothing recorded
    else std::memcpy(&pmin, &bits[1], 4);
    h->stats.f16_peak_max = pmax;
    h->stats.f16_peak_min = pmin;
    h->stats.f16_tracked = h->h_range_launches;
    h->stats.f16_saturated = !(pmax <= kF16Max) ? 1 : 0;
    h->range_failed = h->stats.f16_saturated != 0;
    if (h->stats.f16_saturated)
        return fail(h, VITS_E_RANGE,
                    "an activation of magnitude %g (or a non-finite value) left the range of the generator's fp16 operand "
                    "planes (65504): the f16x3 arithmetic would clamp it.  Open the voice with gen_precision \"bf16x6\" "
                    "(VITSMI_GEN_PRECISION=bf16x6), whose bf16 planes have the fp32 range",
                    (double)h->stats.f16_peak_max);
    return 0;
}

#define HIPCHECK(h, expr)                                                                       \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return fail(h, VITS_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                                    \
    } while (0)

// (slack: a request that needs more than any before it gets a quarter on top - the frame count of the SAME batch moves by
// +-7 % with the noise of a pass - so that growth stops after the first requests; vits_reserve asks for exactly what its
// caller said)
// A workspace that is about to be freed takes the last run's results with it: everything vits_fetch_output /
// vits_last_pcm16 / vits_tap (and a caller of vits_run_device / vits_run_async still holding out->data) would read lives
// in tok or frm.  Those pointers are dropped here, so the next such call fails with "no completed run" instead of reading
// freed device memory (a run that grows a slab sets them again itself, behind the growth).
void forget_results_in(vits_handle *h, const Slab &s) {
    if (&s == &h->tok) {
        h->d_emb = h->d_x = h->d_mp = h->d_logs = h->d_logw = h->d_wceil = nullptr;
        h->d_len = h->d_ylen = h->d_cum = nullptr;
        h->d_ylen64 = nullptr;
    } else if (&s == &h->frm) {
        h->d_zp = h->d_z = h->d_out = nullptr;
    }
}

int slab_reserve(vits_handle *h, Slab &s, size_t bytes, bool slack = true) {
    if (bytes <= s.cap) return 0;
    if (s.base) {
        HIPCHECK(h, hipStreamSynchronize(h->stream));
        forget_results_in(h, s);
        HIPCHECK(h, hipFree(s.base));
        s.base = nullptr;
        s.cap = 0;
    }
    size_t want = bytes + (slack ? bytes / 4 : 0) + (1 << 20);
    static const bool trace = std::getenv("VITSMI_TRACE_ALLOC") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc((void **)&s.base, want);
    if (trace)
        fprintf(stderr, "vitsmi: handle %p slab %p grows to %.1f MB (hipMalloc %.1f ms)\n", (void *)h, (void *)&s, want / 1e6,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (e != hipSuccess) {
        (void)hipGetLastError();  // (the error is reported HERE: the next launch check must not find it)
        return fail(h, VITS_E_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    s.cap = want;
    return 0;
}

template <class Tp>
Tp *slab_take(Slab &s, size_t n) {
    size_t off = (s.used + 255) & ~size_t(255);
    s.used = off + n * sizeof(Tp);
    return reinterpret_cast<Tp *>(s.base + off);
}

inline size_t al(size_t nfloats) { return ((nfloats * 4 + 255) & ~size_t(255)) + 256; }

struct Ctx {
    vits_handle *h;
    const Model &m;
    hipStream_t st;
    const float *A;  // device arena
    int B;
    hipError_t err = hipSuccess;
    // ragged generator (run_generator*): device frame counts + margin, the frame count F the launches' T are multiples of, and
    // the share of B * F frames that lies inside the utterances' (margin-extended) ends - what the FLOP / byte accounting of a
    // generator launch is scaled by (host copy of the frame counts: h_len)
    SxRagged rag{nullptr, 0, 0};
    int rag_F = 0;
    double rag_frac = 1.0;
    const int *h_len = nullptr;
    const float *P(int64_t off) const { return off >= 0 ? A + off : nullptr; }
    // the SxRagged of a generator launch whose input tensors have T columns per utterance
    SxRagged rag_at(int T) const {
        if (!rag.len || rag_F <= 0 || T % rag_F) return SxRagged{nullptr, 0, 0};
        return SxRagged{rag.len, rag.add, T / rag_F};
    }
    double work_frac() const { return rag.len ? rag_frac : 1.0; }
    // the flow's convs on a padded batch (conv_sx): frames behind y_len[b] are neither read nor written (device frame counts;
    // the share of B * F frames inside the utterances, for the accounting)
    const int *flow_len = nullptr;
    double flow_frac = 1.0;
    void note(hipError_t e) {
        if (err == hipSuccess && e != hipSuccess) err = e;
    }
};

// algorithmic FLOPs / layer-granular bytes of one conv launch, per pipeline stage
void conv_account(Ctx &c, const ConvDesc &d, int T) {
    vits_handle *h = c.h;
    // (ragged generator launches: only the columns inside the utterances' ends are worked on)
    const double wf = c.h->cur_stage == 2 && c.flow_len ? c.flow_frac : c.work_frac();
    double fl = 2.0 * d.macs_per_t * (double)T * c.B * wf;
    double by = (d.h1 ? 2.0 : 4.0) * c.B * ((double)d.Cin * T + (double)d.Cout * T) * wf;  // (stored dtype: SURVEY 8d)
    h->stats.conv_flops += fl;
    h->stats.conv_bytes += by;
    h->stats.conv_launches++;
    h->stats.total_launches++;
    switch (h->cur_stage) {
        case 0: h->stats.enc_flops += fl; break;
        case 1: h->stats.dp_flops += fl; break;
        case 2: h->stats.flow_flops += fl; break;
        default:
            h->stats.dec_flops += fl;
            h->stats.dec_bytes += by;
            break;
    }
}

// with timing enabled: record the start event of the next conv launch (the caller records the end event)
bool conv_event_begin(Ctx &c) {
    vits_handle *h = c.h;
    if (h->timing != 1) return false;
    if (h->conv_events_used == h->conv_events.size()) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        h->conv_events.push_back({e0, e1});
    }
    if (h->conv_event_sx.size() < h->conv_events.size()) h->conv_event_sx.resize(h->conv_events.size(), 0);
    if (h->conv_recs.size() < h->conv_events.size()) h->conv_recs.resize(h->conv_events.size());
    h->conv_event_sx[h->conv_events_used] = 0;
    g_launch_name[0] = 0;
    hipEventRecord(h->conv_events[h->conv_events_used].first, c.st);
    return true;
}

// ... and its end event, with what the launch was: the instantiation the launcher picked (g_launch_name), algorithmic
// FLOPs and layer-granular bytes
void conv_event_end(Ctx &c, bool sx, double flops, double bytes, const ConvDesc &d, int T) {
    vits_handle *h = c.h;
    const size_t i = h->conv_events_used++;
    h->conv_event_sx[i] = sx ? 1 : 0;
    vits_launch_record &r = h->conv_recs[i];
    std::snprintf(r.kernel, sizeof r.kernel, "%s", g_launch_name);
    r.flops = flops;
    r.bytes = bytes;
    r.stage = h->cur_stage;
    r.ms = 0.f;
    r.cin = d.Cin;
    r.cout = d.Cout;
    r.k = d.K;
    r.dil = d.dil;
    r.t = T;
    hipEventRecord(h->conv_events[i].second, c.st);
}

// Launch one conv through the engine; accounts algorithmic FLOPs/bytes per stage.
struct ConvExtra {           // rarely used arguments of conv()
    int wn_split = 0;        // EPI_WN: first skip row
    uint16_t *out_pl = nullptr;  // fp16 operand planes of the rows [0, pl_rows) of `out` for a following sx conv
    int pl_rows = 0;
};

void conv(Ctx &c, const ConvDesc &d, const float *x, int64_t x_bstride, int T, float *out, int64_t out_bstride,
          int flags, const int *len = nullptr, const float *res = nullptr, int64_t res_bstride = 0,
          const float *bias_b = nullptr, int bias_b_stride = 0, float slope = 0.1f, float div = 1.f,
          float oslope = 1.f, float *out2 = nullptr, float oslope2 = 1.f, int x_cstride = 0, int out_cstride = 0,
          const ConvExtra *ex = nullptr) {
    ConvArgs a{};
    a.x = x;
    a.x_bstride = x_bstride;
    a.T = T;
    a.len = len;
    a.wp = c.P(d.w_off);
    a.bias = c.P(d.b_off);
    a.bias_b = bias_b;
    a.bias_b_stride = bias_b_stride;
    a.out = out;
    a.out_bstride = out_bstride;
    a.res = res;
    a.res_bstride = res_bstride;
    a.zeros = c.P(c.m.zeros_off);
    a.Cin = d.Cin;
    a.Cout = d.Cout;
    a.K = d.K;
    a.dil = d.dil;
    a.padL = d.padL;
    a.CK = d.CK;
    a.nchunks = d.nchunks;
    a.ups = d.ups;
    a.flags = flags;
    a.slope = slope;
    a.div = div;
    a.oslope = oslope;
    a.out2 = out2;
    a.oslope2 = oslope2;
    a.x_cstride = x_cstride;
    a.out_cstride = out_cstride;
    vits_handle *h = c.h;
    if (ex) {
        a.wn_split = ex->wn_split;
        a.out_pl = ex->out_pl;
        a.pl_rows = ex->pl_rows;
        a.peak = ex->out_pl ? range_slots(h, true) : nullptr;
    }
    const bool ev = conv_event_begin(c);
    c.note(launch_conv(a, d.cfg, c.B, c.st));
    if (ev) conv_event_end(c, false, 2.0 * d.macs_per_t * (double)T * c.B, 4.0 * c.B * ((double)d.Cin * T + (double)d.Cout * T), d, T);
    conv_account(c, d, T);
}

// One conv through the split-operand engine (conv_sx_engine.hip.hpp; 16-bit planes: fp16 x 2 by default, bf16 x 3).  Tensors are whole utterance
// batches in the engine's layouts: input planes [B][3][Cin/8][T][8], outputs raw [B][Cr/8][T*u][8] and/or
// planes [B][3][Cr/8][T*u][8]; `res` has the raw layout of the output.
// `x` is the plane tensor, or (d.rawin) the fp32 raw tensor, to which the kernel applies leaky_relu(islope).
struct SxWn {               // SX_WN_RMW arguments of conv_sx()
    const int *len = nullptr;
    float *out_raw2 = nullptr;
    int row_split = 0, pl_rows = 0;
    bool pl_of2 = false;          // planes of out_raw2's rows instead of out_raw's
    int64_t planar_bstride = 0;   // batch stride of out_raw / res (0: row_split * T)
};

// largest launch (in workgroups of the short-launch kernel) that conv_sx() hands to conv_sx_small_kernel; process-wide,
// VITSMI_SX_SMALL_MAX at start-up, vits_test_set_sx_small_max() for A/B tests
std::atomic<long long> &sx_small_max() {
    static std::atomic<long long> v{[] { const char *e = std::getenv("VITSMI_SX_SMALL_MAX"); return e ? std::atoll(e) : 1536ll; }()};
    return v;
}

void conv_sx(Ctx &c, const ConvDesc &d, const void *x, int T, float *out_raw, uint16_t *out_pl, int flags,
             const float *res = nullptr, const float *bias_b = nullptr, int bias_b_stride = 0, float div = 1.f,
             float oslope = 1.f, float oslope2 = 1.f, float islope = 1.f, const SxWn *wn = nullptr,
             const uint16_t *res_pl = nullptr, float res_slope = 1.f) {
    // res_pl (single-plane mode, d.h1): the residual is the plane tensor that holds leaky_relu(residual, res_slope)
    SxArgs a{};
    const int Cr = d.Cout / d.ups;
    const int64_t Tout = (int64_t)T * d.ups;
    if (d.rawin) a.xr = static_cast<const float *>(x);
    else a.xp = static_cast<const u32x4 *>(x);
    a.islope = islope;
    a.x_bstride = (int64_t)3 * (d.Cin / 8) * T;
    a.T = T;
    a.wp = reinterpret_cast<const u32x4 *>(c.P(d.w_off));
    a.bias = c.P(d.b_off);
    a.bias_b = bias_b;
    a.bias_b_stride = bias_b_stride;
    a.out_raw = out_raw;
    a.raw_bstride = (flags & SX_GATE) ? (int64_t)(Cr / 2) * Tout : (int64_t)Cr * Tout;  // (gate: planar acts [H][T])
    a.out_pl = out_pl;
    a.pl_bstride = (flags & SX_GATE) ? (int64_t)3 * (Cr / 2) * Tout : (int64_t)3 * Cr * Tout;  // (gate: planes of the H acts)
    if (wn) {  // SX_WN_RMW: planes of the first pl_rows rows only
        a.len = wn->len;
        a.out_raw2 = wn->out_raw2;
        a.row_split = wn->row_split;
        a.pl_rows = wn->pl_rows;
        a.pl_bstride = (int64_t)3 * wn->pl_rows * Tout;
        a.pl_of2 = wn->pl_of2 ? 1 : 0;
        a.planar_bstride = wn->planar_bstride;
    }
    a.res = res;
    a.res_pl = res_pl;
    a.res_unslope = 1.f / res_slope;
    a.zeros = c.P(c.m.zeros_off);
    a.Cin = d.Cin;
    a.Cout = d.Cout;
    a.Cr = Cr;
    a.K = d.K;
    a.dil = d.dil;
    a.padL = d.padL;
    a.nchunks = d.nchunks;
    a.ups = d.ups;
    a.flags = flags;
    a.div = div;
    a.oslope = oslope;
    a.oslope2 = oslope2;
    a.wscale = d.wscale;
    a.s16 = d.s16 ? 1 : 0;
    {
        static const bool zt_off = std::getenv("VITSMI_NO_ZERO_TAP_SKIP") != nullptr;  // A/B timing
        a.zt_p = zt_off ? -1 : d.zt_p;
    }
    vits_handle *h = c.h;
    if (h->cur_stage == 3) a.rag = c.rag_at(T);
    // the flow's tensors are masked by y_len (modules.py:447-466: every conv's input is x * mask, every result * mask): a
    // padded batch's frames behind an utterance's end are zeros the reference computes and this engine neither reads nor
    // writes (run_frames: Ctx::flow_len, when every launch of the flow is one of these)
    else if (h->cur_stage == 2 && c.flow_len) a.rag = SxRagged{c.flow_len, 0, 1};
    a.peak = range_slots(h, (d.f16 || d.h1) && (d.rawin || out_pl));  // launches that turn values into fp16 planes
    const bool ev = conv_event_begin(c);
    // Short grids (a single utterance, a streaming chunk): the 128-row packing is read by the 64- or 32-row kernel -
    // 2-4x the workgroups, each with half / a quarter of the reduction work per step.  Same arithmetic.
    static const bool tall_only = std::getenv("VITSMI_SX_NO_SHORT_TILES") != nullptr;  // A/B timing only
    int run_cfg = d.cfg;
    if (d.cfg == 0 && !d.rawin && !tall_only) {
        const long long wgs0 = (long long)((T + 255) / 256) * c.B * (d.Cout / 128);
        if (wgs0 <= 128) run_cfg = 2;
        else if (wgs0 <= 256) run_cfg = 1;
        // token- / frame-domain convs with the planar epilogue (short reductions, bound by their prologue / epilogue
        // latencies): 64 x 128 tiles put 4-7x the workgroups on the chip (res_skip 51 -> 42 us, FFN conv_1 43 -> 36 us)
        if (d.s16 && (flags & SX_WN_RMW) && wgs0 > 128 && wgs0 <= 1024) run_cfg = 3;
    }
    // 64-row layers on the 16x16x32 loop: 128-column tiles where the grid of 256-column ones fills the chip badly (the
    // flow's WN in-layers at batch 32: 768 workgroups on 512 slots = two rounds of which the second is half empty, and 24 %
    // padding columns; as 1344 half-size workgroups on 768 slots: the time of one full round).  Same arithmetic.
    static const bool no_narrow = std::getenv("VITSMI_SX_NO_NARROW_TILES") != nullptr;  // A/B timing only
    if (d.cfg == 1 && d.s16 && !d.rawin && !no_narrow) {
        const long long mt = d.Cout / 64;
        const long long w256 = ((long long)((T + 255) / 256) * c.B + 7) / 8 * 8 * mt, w128 = ((long long)((T + 127) / 128) * c.B + 7) / 8 * 8 * mt;
        const long long r256 = (w256 + 511) / 512 * 2, r128 = (w128 + 767) / 768;  // rounds, in units of a 128-column tile
        if (r128 < r256) run_cfg = 3;
    }
    {
        // (experiments only: VITSMI_SX_FORCE_CFG=<stage><cfg>, e.g. 22 = the flow's plane-input convs on the 32-row tile)
        static const int force = [] { const char *e = std::getenv("VITSMI_SX_FORCE_CFG"); return e ? std::atoi(e) : -1; }();
        if (force >= 0 && d.s16 && !d.rawin && h->cur_stage == force / 10 && (!(flags & SX_GATE) || force % 10 == 1 || force % 10 == 3) &&
            sx_tile_m(force % 10) <= sx_tile_m(d.cfg) && d.Cout % sx_tile_m(force % 10) == 0)
            run_cfg = force % 10;
    }
    // Short launches of the token / frame domain (one utterance, a streaming chunk): the reduction-splitting kernel of
    // conv_sx_small.hip.hpp.  VITSMI_SX_SMALL_MAX = largest launch in workgroups that takes it (0: never; A/B timing).
    const long long small_max = sx_small_max().load(std::memory_order_relaxed);
    const int nprod = d.h1 ? 1 : (d.f16 ? 2 : 6);
    // (generator convs - neither planar nor gated - stay on the engine whatever the launch size: a chunked rendering
    // (vits_run_chunked) must equal the unchunked one bit for bit, so a conv's kernel must not depend on how many frames a launch
    // covers.  VITSMI_SX_SMALL_GEN=1 lifts that for single-shot latency: 1.31 -> 1.27 ms on `medium`, 3.34 -> 3.24 on `high`.)
    static const bool small_gen = [] { const char *e = std::getenv("VITSMI_SX_SMALL_GEN"); return e && e[0] == '1'; }();
    const bool small_kind = (a.flags & (SX_WN_RMW | SX_GATE)) != 0 || small_gen;
    if (small_max > 0 && small_kind && conv_sx_small_ok(a, d.rawin, nprod) && conv_sx_small_wgs(a, c.B) <= small_max)
        c.note(launch_conv_sx_small(a, c.B, d.cfg, c.st));
    else
        c.note(launch_conv_sx(a, run_cfg, c.B, c.st, d.rawin, nprod, d.cfg));
    // layer-granular bytes in the STORED dtype ("bf16 storage halves these"): 2 bytes per element in the
    // single-plane mode, 4 otherwise
    const double ebytes = d.h1 ? 2.0 : 4.0;
    const double wf = !a.rag.len ? 1.0 : (h->cur_stage == 2 ? c.flow_frac : c.rag_frac);
    const double fl = 2.0 * d.macs_per_t * (double)T * c.B * wf, by = ebytes * c.B * ((double)d.Cin * T + (double)d.Cout * T) * wf;
    if (ev) conv_event_end(c, true, fl, by, d, T);
    conv_account(c, d, T);
    h->stats.sx_flops += fl;
    h->stats.sx_bytes += by;
    h->stats.sx_launches++;
}

// a3: relative-position attention core, one launch (kernels.hip.hpp).  Two key halves per workgroup unless
// VITSMI_ATT_NOSPLIT is set (A/B timing only).
void launch_attention(hipStream_t st, int B, int T, int n_heads, int dk, int window, const float *qkv, float *att,
                      const float *rel_k, const float *rel_v, const int *len, int H, uint16_t *planes = nullptr,
                      unsigned *peak = nullptr) {
    static const bool nosplit = std::getenv("VITSMI_ATT_NOSPLIT") != nullptr;
    dim3 ag((T + 127) / 128, n_heads, B);
    const int dkb = (dk + 31) / 32;
#define VITSMI_ATT(DKB)                                                                                                       \
    do {                                                                                                                      \
        if (nosplit) attention_relpos_kernel<DKB, 1><<<ag, 256, 0, st>>>(qkv, att, rel_k, rel_v, len, H, T, dk, window, planes, peak); \
        else attention_relpos_kernel<DKB, 2><<<ag, 512, 0, st>>>(qkv, att, rel_k, rel_v, len, H, T, dk, window, planes, peak);         \
    } while (0)
    switch (dkb) {
        case 1: VITSMI_ATT(1); break;
        case 2: VITSMI_ATT(2); break;
        case 3: VITSMI_ATT(3); break;
        default: VITSMI_ATT(4); break;
    }
#undef VITSMI_ATT
}

// ... on the 16-bit matrix pipe as f16x3 products (attention16.hip.hpp): head widths of 32 / 64 / 96, q | k | v given as
// operand planes too (the q|k|v conv's planar epilogue writes them).  VITSMI_ATT16=0 keeps the fp32-MFMA kernel (A/B timing).
bool attention16_ok(int dk, int window) {
    static const bool off = [] { const char *e = std::getenv("VITSMI_ATT16"); return e && e[0] == '0'; }();
    return !off && dk % 32 == 0 && dk <= 96 && window <= 4;
}
template <int DKS, int NS, int QT>
hipError_t launch_attention16_t(hipStream_t st, const Att16Args &a) {
    constexpr int lds = NS * 4 * (DKS * 2) * 1024 + 9 * DKS * 32 * 4 + 16 + 64 * QT * kAtt16RelPitch * 4;
    static std::atomic<uint64_t> done{0};
    if (lds > 64 * 1024)
        if (hipError_t e = sx_allow_big_lds(reinterpret_cast<const void *>(&attention_relpos16_kernel<DKS, NS, QT>), done)) return e;
    const int ntile = (a.T + 64 * QT - 1) / (64 * QT), npair = a.nh * a.B;
    attention_relpos16_kernel<DKS, NS, QT><<<dim3(((npair + 7) / 8) * 8 * ntile), 256, lds, st>>>(a);
    return hipSuccess;
}
// stages / query tiles per wave: two stages, 64 queries per workgroup; VITSMI_ATT16_NS / VITSMI_ATT16_QT override (A/B timing)
hipError_t launch_attention16(hipStream_t st, int B, int T, int n_heads, int dk, int window, const uint16_t *qkv_pl, float *att,
                              uint16_t *att_pl, const float *rel_k, const float *rel_v, const int *len, int H, unsigned *peak,
                              unsigned long long *prof = nullptr) {
    static const int env_ns = [] { const char *e = std::getenv("VITSMI_ATT16_NS"); return e ? std::atoi(e) : 0; }();
    static const int env_qt = [] { const char *e = std::getenv("VITSMI_ATT16_QT"); return e ? std::atoi(e) : 0; }();
    Att16Args a{};
    a.qkv_pl = qkv_pl;
    a.out = att;
    a.out_pl = att_pl;
    a.relk = rel_k;
    a.relv = rel_v;
    a.len = len;
    a.Hc = H;
    a.T = T;
    a.dk = dk;
    a.win = window;
    a.nh = n_heads;
    a.B = B;
    a.peak = att_pl ? peak : nullptr;
    a.prof = prof;
    int ns = env_ns ? env_ns : 2, qt = env_qt ? env_qt : 1;  // (measured: 2, 3 and 4 stages time alike)
    if (dk == 32) return launch_attention16_t<1, 3, 1>(st, a);
    if (dk == 64) return launch_attention16_t<2, 3, 1>(st, a);
    if (qt == 2) return ns >= 3 ? launch_attention16_t<3, 3, 2>(st, a) : launch_attention16_t<3, 2, 2>(st, a);
    if (ns >= 4) return launch_attention16_t<3, 4, 1>(st, a);
    return ns == 3 ? launch_attention16_t<3, 3, 1>(st, a) : launch_attention16_t<3, 2, 1>(st, a);
}

// A token- / frame-domain conv on the split-operand engine with the planar epilogue (SX_WN_RMW): x_pl = fp16 operand planes
// of the input; out (may be nullptr when out_pl is given) = planar fp32 [B][Cout][T]; out_pl = operand planes of the output
// for the next conv; flags = EPI_RELU | EPI_MASK | EPI_ACC | EPI_RES (res: planar, the shape of out).
void conv_sx_planar(Ctx &c, const ConvDesc &d, const uint16_t *x_pl, int T, float *out, uint16_t *out_pl, int flags,
                    const int *len = nullptr, const float *res = nullptr) {
    SxWn w;
    w.len = len;
    w.row_split = d.Cout;
    w.pl_rows = out_pl ? d.Cout : 0;
    conv_sx(c, d, x_pl, T, out, out_pl, SX_WN_RMW | flags, res, nullptr, 0, 1.f, 1.f, 1.f, 1.f, &w);
}

// planar fp32 [B][C][T] (masked by len when given) -> fp16 operand planes of the split-operand engine
void split_planes(Ctx &c, const float *x, uint16_t *pl, int C, int T, const int *len) {
    sx_split_planes_kernel<<<dim3((T + 255) / 256, C / 8, c.B), 256, 0, c.st>>>(x, (int64_t)C * T, T, len, pl, C, T, 1,
                                                                                range_slots(c.h, true));
    c.note(hipGetLastError());
    c.h->stats.total_launches++;
}

// Two dependent convs of a ResBlock as ONE launch (conv_sx_pair.hip.hpp), raw-format stages only (x, out: fp32 raw
// [B][C/8][T][8]): a ResBlock1 step, out = c2(lrelu(c1(lrelu(x)))) + x, or (chain) two ResBlock2 steps,
// x1 = c1(lrelu(x)) + x, out = c2(lrelu(x1)) + x1; then [+ out] [/ div].
bool sx_pair_ok(const vits_handle *h, const ConvDesc &c1, const ConvDesc &c2) {
    static const bool off = std::getenv("VITSMI_SX_NO_PAIR") != nullptr;  // A/B timing only
    return !off && h->gen_nprod == 2 && c1.f16 && c2.f16 && c1.rawin && c2.rawin && c1.cfg == c2.cfg && c1.ups == 1 &&
           c2.ups == 1 && c1.Cin == c1.Cout && c2.Cin == c2.Cout && c1.Cin == c2.Cin && c2.padL * 2 == (c2.K - 1) * c2.dil &&
           sx_pair_supported(c1.Cin, c1.cfg, c1.K, c1.dil, c2.K, c2.dil);
}

void conv_sx_pair(Ctx &c, const ConvDesc &c1, const ConvDesc &c2, const float *x, int T, float *out, int flags, float div,
                  float slope, bool chain = false) {
    SxPairArgs a{};
    a.xr = x;
    a.islope = slope;
    a.mslope = slope;
    a.T = T;
    a.wp1 = reinterpret_cast<const u32x4 *>(c.P(c1.w_off));
    a.wp2 = reinterpret_cast<const u32x4 *>(c.P(c2.w_off));
    a.bias1 = c.P(c1.b_off);
    a.bias2 = c.P(c2.b_off);
    a.wscale1 = c1.wscale;
    a.wscale2 = c2.wscale;
    a.out_raw = out;
    a.zeros = c.P(c.m.zeros_off);
    a.C = c1.Cin;
    a.K1 = c1.K;
    a.dil1 = c1.dil;
    a.pad1 = c1.padL;
    a.K2 = c2.K;
    a.dil2 = c2.dil;
    a.pad2 = c2.padL;
    a.flags = flags & (EPI_ACC | EPI_DIV);
    a.div = div;
    vits_handle *h = c.h;
    a.rag = c.rag_at(T);
    a.peak = range_slots(h, true);
#if SX_PAIR_PROF
    {
        // diagnostic build: one row of 8 counters per pair launch of a run, printed by the next launch_table / bench run
        static unsigned long long *rows = nullptr;
        static int idx = 0;
        if (!rows) {
            hipMalloc((void **)&rows, 64 * 8 * sizeof(unsigned long long));
            hipMemset(rows, 0, 64 * 8 * sizeof(unsigned long long));
        }
        if (h->stats.sx_launches == 0 || idx >= 64) idx = 0;
        a.prof = rows + (idx++ % 64) * 8;
        if (std::getenv("VITSMI_PAIR_PROF_DUMP") && idx == 1) {
            unsigned long long hrows[64 * 8];
            hipDeviceSynchronize();
            hipMemcpy(hrows, rows, sizeof hrows, hipMemcpyDeviceToHost);
            for (int r = 0; r < 64; r++)
                if (hrows[r * 8 + 7]) {
                    fprintf(stderr, "pairprof row %d wgs %llu:", r, hrows[r * 8 + 7]);
                    for (int i = 0; i < 7; i++) fprintf(stderr, " %.0f", (double)hrows[r * 8 + i] / (double)hrows[r * 8 + 7]);
                    fprintf(stderr, "\n");
                }
            hipMemset(rows, 0, sizeof hrows);
        }
    }
#endif
    const bool ev = conv_event_begin(c);
    c.note(launch_conv_sx_pair(a, c1.cfg, c.B, c.st, chain));
    const double wf = c.work_frac();
    const double fl = 2.0 * (c1.macs_per_t + c2.macs_per_t) * (double)T * c.B * wf;
    const double by = 4.0 * c.B * ((double)(c1.Cin + c1.Cout) * T + (double)(c2.Cin + c2.Cout) * T) * wf;
    if (ev) conv_event_end(c, true, fl, by, c1, T);
    conv_account(c, c1, T);
    conv_account(c, c2, T);
    h->stats.conv_launches--;  // (two convs, one launch)
    h->stats.total_launches--;
    h->stats.sx_flops += fl;
    h->stats.sx_bytes += by;
    h->stats.sx_launches++;
}

// Two dependent convs of a ResBlock on the 16x16x32 loop (conv_sx_pair16.hip.hpp): f16x3 (fp32 raw tensors; the convs'
// second weight copy, ConvDesc::w16_off) or the single-plane arithmetic (fp16 plane tensors).  VITSMI_PAIR16=0: off (A/B).
bool sx_pair16_ok(const vits_handle *h, const ConvDesc &c1, const ConvDesc &c2) {
    static const bool off = [] {
        const char *e = std::getenv("VITSMI_PAIR16");
        return (e && e[0] == '0') || std::getenv("VITSMI_SX_NO_PAIR") != nullptr;
    }();
    if (off || c1.Cin != c1.Cout || c2.Cin != c2.Cout || c1.Cin != c2.Cin || c1.ups != 1 || c2.ups != 1) return false;
    if (c1.padL * 2 != (c1.K - 1) * c1.dil || c2.padL * 2 != (c2.K - 1) * c2.dil) return false;
    const bool h1 = c1.h1 && c2.h1, f3 = c1.f16 && c2.f16 && c1.s16 && c2.s16 && !c1.rawin && !c2.rawin;
    if (!h1 && !f3) return false;
    if (c1.cfg != c2.cfg || c1.cfg != (c1.Cin == 64 ? 1 : 2)) return false;  // (weights packed for a tile of all C rows)
    (void)h;
    return sx_pair16_plan(c1.Cin, h1 ? 1 : 2, c1.K, c1.dil, c2.K, c2.dil, nullptr) != 0;
}

// x: the operand-plane tensor holding leaky_relu(x, slope) (two fp16 planes: f16x3; one: the single-plane arithmetic).
// out_raw: fp32 raw destination (flags & P16_HAS_RAW) and / or EPI_ACC operand; out_pl (flags & P16_HAS_PL): operand planes of
// leaky_relu(result, slope).
void conv_sx_pair16(Ctx &c, const ConvDesc &c1, const ConvDesc &c2, const uint16_t *x, int T, float *out_raw, uint16_t *out_pl, int flags,
                    float div, float slope, bool chain) {
    SxPair16Args a{};
    const bool h1 = c1.h1;
    const int C = c1.Cin;
    a.xpl = x;
    a.x_bstride = (int64_t)3 * C * T;
    a.islope = a.mslope = a.oslope = slope;
    a.T = T;
    a.wp1 = reinterpret_cast<const u32x4 *>(c.P(c1.w_off));
    a.wp2 = reinterpret_cast<const u32x4 *>(c.P(c2.w_off));
    a.bias1 = c.P(c1.b_off);
    a.bias2 = c.P(c2.b_off);
    a.wscale1 = c1.wscale;
    a.wscale2 = c2.wscale;
    a.out_raw = out_raw;
    a.raw_bstride = (int64_t)C * T;
    a.out_pl = out_pl;
    a.pl_bstride = (int64_t)3 * C * T;
    a.zeros = c.P(c.m.zeros_off);
    a.K1 = c1.K; a.dil1 = c1.dil; a.pad1 = c1.padL;
    a.K2 = c2.K; a.dil2 = c2.dil; a.pad2 = c2.padL;
    a.flags = flags;
    a.div = div;
    vits_handle *h = c.h;
    a.rag = c.rag_at(T);
    a.peak = range_slots(h, true);
    const double ebytes = h1 ? 2.0 : 4.0;
    const bool ev = conv_event_begin(c);
    c.note(launch_conv_sx_pair16(a, C, h1 ? 1 : 2, c.B, c.st, chain));
    const double fl = 2.0 * (c1.macs_per_t + c2.macs_per_t) * (double)T * c.B * c.work_frac();
    const double by = ebytes * c.B * ((double)(c1.Cin + c1.Cout) * T + (double)(c2.Cin + c2.Cout) * T) * c.work_frac();
    if (ev) conv_event_end(c, true, fl, by, c1, T);
    conv_account(c, c1, T);
    conv_account(c, c2, T);
    h->stats.conv_launches--;  // (two convs, one launch)
    h->stats.total_launches--;
    h->stats.sx_flops += fl;
    h->stats.sx_bytes += by;
    h->stats.sx_launches++;
}

// The multi-receptive-field sum of a 32-channel ResBlock2 stage, xs = (rb_0(x) + .. + rb_{n-1}(x)) / n with every rb a
// two-step chain (models.py:356-363, modules.py:355-364), as ONE launch (conv_sx_pair_kernel<.., NCH = n>): can it?
bool sx_mrf_ok(const vits_handle *h, const UpStageDesc &stg) {
    // Opt-in (VITSMI_SX_MRF=1, read per run): measured on the default voice, the fused stage moves 4x fewer bytes and takes
    // the same time on one handle (2.39 vs 2.46 ms: the 32-row tiles are bound by their own dependent chains, not by HBM) and
    // loses 4-10 % of the step under the three-handle schedule (two long workgroups per CU instead of three short ones).
    const bool off = std::getenv("VITSMI_SX_MRF") == nullptr;
    const int n = (int)stg.rbs.size();
    if (off || n < 2 || n > 3) return false;
    int K1[3], d1[3], K2[3], d2[3];
    for (int j = 0; j < n; j++) {
        const auto &rb = stg.rbs[j];
        if (rb.type1 || rb.n != 2 || !sx_pair_ok(h, rb.c1[0], rb.c1[1]) || rb.c1[0].padL * 2 != (rb.c1[0].K - 1) * rb.c1[0].dil)
            return false;
        K1[j] = rb.c1[0].K;
        d1[j] = rb.c1[0].dil;
        K2[j] = rb.c1[1].K;
        d2[j] = rb.c1[1].dil;
    }
    return sx_mrf_geom(stg.rbs[0].c1[0].Cin, n, K1, d1, K2, d2, nullptr);
}

void conv_sx_mrf(Ctx &c, const UpStageDesc &stg, const float *x, int T, float *out, float slope) {
    const int n = (int)stg.rbs.size();
    SxPairArgs a{};
    a.xr = x;
    a.islope = slope;
    a.mslope = slope;
    a.T = T;
    a.out_raw = out;
    a.zeros = c.P(c.m.zeros_off);
    a.C = stg.rbs[0].c1[0].Cin;
    a.div = (float)n;
    a.nchain = n;
    double macs = 0, bytes = 0;
    for (int j = 0; j < n; j++) {
        const ConvDesc &c1 = stg.rbs[j].c1[0], &c2 = stg.rbs[j].c1[1];
        auto &ch = a.ch[j];
        ch.wp1 = reinterpret_cast<const u32x4 *>(c.P(c1.w_off));
        ch.wp2 = reinterpret_cast<const u32x4 *>(c.P(c2.w_off));
        ch.bias1 = c.P(c1.b_off);
        ch.bias2 = c.P(c2.b_off);
        ch.wscale1 = c1.wscale;
        ch.wscale2 = c2.wscale;
        ch.K1 = c1.K;
        ch.dil1 = c1.dil;
        ch.K2 = c2.K;
        ch.dil2 = c2.dil;
        macs += c1.macs_per_t + c2.macs_per_t;
        bytes += 4.0 * ((double)(c1.Cin + c1.Cout) + (double)(c2.Cin + c2.Cout));  // (layer-granular, as the separate launches)
    }
    vits_handle *h = c.h;
    a.rag = c.rag_at(T);
    a.peak = range_slots(h, true);
    macs *= c.work_frac();
    bytes *= c.work_frac();
    const bool ev = conv_event_begin(c);
    c.note(launch_conv_sx_mrf(a, c.B, c.st));
    if (ev) conv_event_end(c, true, 2.0 * macs * (double)T * c.B, bytes * T * c.B, stg.rbs[n - 1].c1[0], T);
    for (int j = 0; j < n; j++) {
        conv_account(c, stg.rbs[j].c1[0], T);
        conv_account(c, stg.rbs[j].c1[1], T);
    }
    h->stats.conv_launches -= 2 * n - 1;  // (2 n convs, one launch)
    h->stats.total_launches -= 2 * n - 1;
    h->stats.sx_flops += 2.0 * macs * (double)T * c.B;
    h->stats.sx_bytes += bytes * T * c.B;
    h->stats.sx_launches++;
}

// Pinned output buffer of `bytes` bytes: the handle's pool when it is free (grown on demand), else a fresh
// allocation (the caller still holds an earlier output).  Released by pinned_put().
void *pinned_get(vits_handle *h, size_t bytes) {
    PinnedPool &p = h->pin;
    if (!p.busy) {
        if (bytes > p.cap) {
            if (p.base) hipHostFree(p.base);
            p.base = nullptr;
            p.cap = 0;
            size_t want = bytes + bytes / 4 + 4096;
            if (hipHostMalloc((void **)&p.base, want) != hipSuccess) return nullptr;
            p.cap = want;
        }
        p.busy = true;
        return p.base;
    }
    void *q = nullptr;
    return hipHostMalloc(&q, bytes) == hipSuccess ? q : nullptr;
}

void pinned_put(vits_handle *h, void *q) {
    if (!q) return;
    if (h && q == h->pin.base) h->pin.busy = false;
    else hipHostFree(q);
}

// planes (optional): the result once more as fp16 operand planes of the split-operand engine (see split_planes)
void layernorm(Ctx &c, const float *in, float *out, int64_t g, int64_t b, const int *len, int C, int T, int flags,
               uint16_t *planes = nullptr) {
    // (16 time steps per workgroup: VITSMI_LN_TS=32 keeps the 32-step form, A/B timing)
    static const bool ts32 = [] { const char *e = std::getenv("VITSMI_LN_TS"); return e && std::atoi(e) == 32; }();
    if (C <= 256 && planes && C % 8 == 0 && !ts32)
        ln_tile_kernel<0, true, 16><<<dim3((T + 15) / 16, c.B), 256, 0, c.st>>>(in, out, c.P(g), c.P(b), len, C, T, flags, nullptr,
                                                                                nullptr, 1, 1, planes, range_slots(c.h, true));
    else if (C <= 256 && planes && C % 8 == 0)
        ln_tile_kernel<0, true><<<dim3((T + 31) / 32, c.B), 256, 0, c.st>>>(in, out, c.P(g), c.P(b), len, C, T, flags, nullptr,
                                                                            nullptr, 1, 1, planes, range_slots(c.h, true));
    else if (C <= 256 && !ts32)
        ln_tile_kernel<0, false, 16><<<dim3((T + 15) / 16, c.B), 256, 0, c.st>>>(in, out, c.P(g), c.P(b), len, C, T, flags, nullptr,
                                                                                 nullptr, 1, 1);
    else if (C <= 256)
        ln_tile_kernel<0><<<dim3((T + 31) / 32, c.B), 256, 0, c.st>>>(in, out, c.P(g), c.P(b), len, C, T, flags, nullptr,
                                                                      nullptr, 1, 1);
    else
        layernorm_c_kernel<<<dim3((T + 63) / 64, c.B), 64, 0, c.st>>>(in, out, c.P(g), c.P(b), len, C, T, flags);
    c.note(hipGetLastError());
    c.h->stats.total_launches++;
    if (planes && !(C <= 256 && C % 8 == 0)) split_planes(c, out, planes, C, T, nullptr);
}

// DDSConv (modules.py:117-129) in place on h [B,C,T]; y,y2 are scratch of the same size.
// tail (optional): the masked 1 x 1 conv `tail` (weights once more at tail16 in the 16-column kernel's layout) applied to the
// stack's result inside the last layer's launch, written to tail_out [B][tail->Cout][T]; hbuf does then NOT receive the stack's
// result.  Returns whether the tail was taken (false: the caller runs the conv itself).
// head (optional, ConvFlow stacks): the stack's input is head->w[c] * z + head->b[c] + head->in (ConvFlow.pre + conditioning),
// formed by the first layer while it loads; hbuf is then never read.  *head_done tells whether that happened (false: the caller
// must have filled hbuf, e.g. by cf_pre_kernel).
struct DdsHead {
    const float *in, *z, *w, *b;  // conditioning tensor [B][C][T], z channel row of utterance 0 (rows 2 T apart), pre weights
};
bool dds16_head_ok(const DDSDesc &d, int C) {
    static const bool off = [] { const char *e = std::getenv("VITSMI_DDS16"); return e && e[0] == '0'; }();
    static const bool head_off = [] { const char *e = std::getenv("VITSMI_DDS_HEAD"); return e && e[0] == '0'; }();  // A/B timing only
    static const bool unfused = std::getenv("VITSMI_DDS_UNFUSED") != nullptr;
    const int nblk = C / 32;
    return !off && !head_off && !unfused && C <= 256 && C % 32 == 0 && d.K == 3 && d.n_layers > 1 && d.l[0].pw16 >= 0 &&
           (nblk == 2 || nblk == 4 || nblk == 6 || nblk == 8);
}
bool ddsconv(Ctx &c, const DDSDesc &d, float *hbuf, float *y, float *y2, const int *len, int C, int T,
             const ConvDesc *tail = nullptr, int64_t tail16 = -1, float *tail_out = nullptr, const DdsHead *head = nullptr) {
    static const bool unfused = std::getenv("VITSMI_DDS_UNFUSED") != nullptr;  // A/B timing only
    static const bool tail_off = [] { const char *e = std::getenv("VITSMI_DDS_TAIL"); return e && e[0] == '0'; }();  // A/B timing only
    bool tail_done = false;
    const int nblk = C / 32;
    if (!unfused && C <= 256 && C % 32 == 0 && nblk != 5 && nblk != 7 && d.K == 3 && d.n_layers > 0) {
        // one launch per layer (dds_layer_kernel), ping-ponging between the three buffers so that the result of the
        // last layer lands in hbuf: each layer must write a buffer other than the one it reads
        float *bufs[3] = {hbuf, y, y2};
        int cur = 0;
        for (int l = 0; l < d.n_layers; l++) {
            const auto &L = d.l[l];
            const int left = d.n_layers - 1 - l;          // layers after this one
            int nxt = left == 0 ? 0 : (cur == 1 ? 2 : 1); // last layer writes hbuf ...
            if (nxt == cur) {                             // ... unless it would read it too (n_layers == 1): detour
                nxt = 1;
            }
            // 16 columns per workgroup on v_mfma_f32_16x16x4_f32 (kernels.hip.hpp dds_layer16_kernel; VITSMI_DDS16=0: the
            // 32-column kernel, A/B timing)
            static const bool dds16_off = [] { const char *e = std::getenv("VITSMI_DDS16"); return e && e[0] == '0'; }();
            if (!dds16_off && L.pw16 >= 0 && (nblk == 2 || nblk == 4 || nblk == 6 || nblk == 8)) {
                DdsLayer16Args q{};
                q.in = bufs[cur];
                q.out = bufs[nxt];
                if (l == 0 && head) {  // (dds16_head_ok: the caller checked that this path is taken)
                    q.in = head->in;
                    q.head_z = head->z;
                    q.head_w = head->w;
                    q.head_b = head->b;
                    q.head_zstride = (int64_t)2 * T;
                }
                q.len = len;
                q.dw_w = c.P(L.dw_w);
                q.dw_b = c.P(L.dw_b);
                q.ln1_g = c.P(L.ln1_g);
                q.ln1_b = c.P(L.ln1_b);
                q.ln2_g = c.P(L.ln2_g);
                q.ln2_b = c.P(L.ln2_b);
                q.pw16 = c.P(L.pw16);
                q.pw_bias = L.pw.b_off >= 0 ? c.P(L.pw.b_off) : c.P(c.m.zeros_off);
                q.T = T;
                q.dil = L.dil;
                q.mask_out = l == d.n_layers - 1;
                if (left == 0 && tail && tail16 >= 0 && tail_out && !tail_off && tail->Cin == C && tail->K == 1 && tail->Cout <= C) {
                    q.tail_w16 = c.P(tail16);
                    q.tail_b = tail->b_off >= 0 ? c.P(tail->b_off) : nullptr;
                    q.tail_out = tail_out;
                    q.tail_rows = tail->Cout;
                    q.tail_mask = 1;
                    tail_done = true;
                    const double tfl = 2.0 * tail->macs_per_t * (double)T * c.B;  // (accounted as conv() would)
                    c.h->stats.conv_flops += tfl;
                    (c.h->cur_stage == 1 ? c.h->stats.dp_flops : c.h->stats.enc_flops) += tfl;
                }
                const dim3 dg16((T + 15) / 16, c.B);
                switch (nblk) {
                    case 2: dds_layer16_kernel<2><<<dg16, 256, 0, c.st>>>(q); break;
                    case 4: dds_layer16_kernel<4><<<dg16, 256, 0, c.st>>>(q); break;
                    case 6: dds_layer16_kernel<6><<<dg16, 256, 0, c.st>>>(q); break;
                    default: dds_layer16_kernel<8><<<dg16, 256, 0, c.st>>>(q); break;
                }
                c.note(hipGetLastError());
                c.h->stats.total_launches++;
                {
                    const double fl = 2.0 * L.pw.macs_per_t * (double)T * c.B;
                    c.h->stats.conv_flops += fl;
                    (c.h->cur_stage == 1 ? c.h->stats.dp_flops : c.h->stats.enc_flops) += fl;
                }
                cur = nxt;
                continue;
            }
            DdsLayerArgs a{};
            a.in = bufs[cur];
            a.out = bufs[nxt];
            a.len = len;
            a.dw_w = c.P(L.dw_w);
            a.dw_b = c.P(L.dw_b);
            a.ln1_g = c.P(L.ln1_g);
            a.ln1_b = c.P(L.ln1_b);
            a.ln2_g = c.P(L.ln2_g);
            a.ln2_b = c.P(L.ln2_b);
            a.pw = c.P(L.pw.w_off);
            a.pw_bias = L.pw.b_off >= 0 ? c.P(L.pw.b_off) : c.P(c.m.zeros_off);
            a.C = C;
            a.T = T;
            a.dil = L.dil;
            a.mask_out = l == d.n_layers - 1;
            a.CK = L.pw.CK;
            a.nchunks = L.pw.nchunks;
            a.MB = (L.pw.cfg == 2 ? 128 : ((L.pw.cfg == 1 || L.pw.cfg == 3) ? 64 : 32)) / 32;
            const dim3 dg((T + 31) / 32, c.B);
            switch (nblk) {  // (channel count at compile time: straight-line channel loops)
                case 1: dds_layer_kernel<1><<<dg, 256, 0, c.st>>>(a); break;
                case 2: dds_layer_kernel<2><<<dg, 256, 0, c.st>>>(a); break;
                case 3: dds_layer_kernel<3><<<dg, 256, 0, c.st>>>(a); break;
                case 4: dds_layer_kernel<4><<<dg, 256, 0, c.st>>>(a); break;
                case 6: dds_layer_kernel<6><<<dg, 256, 0, c.st>>>(a); break;
                default: dds_layer_kernel<8><<<dg, 256, 0, c.st>>>(a); break;
            }
            c.note(hipGetLastError());
            c.h->stats.total_launches++;
            {   // the 1x1 conv's algorithmic work, as conv() would account it
                const double fl = 2.0 * L.pw.macs_per_t * (double)T * c.B;
                c.h->stats.conv_flops += fl;
                (c.h->cur_stage == 1 ? c.h->stats.dp_flops : c.h->stats.enc_flops) += fl;
            }
            cur = nxt;
        }
        if (cur != 0 && !tail_done) c.note(hipMemcpyAsync(hbuf, bufs[cur], (size_t)c.B * C * T * 4, hipMemcpyDeviceToDevice, c.st));
        return tail_done;
    }
    for (int l = 0; l < d.n_layers; l++) {
        const auto &L = d.l[l];
        if (C <= 256 && d.K == 3)
            ln_tile_kernel<3><<<dim3((T + 31) / 32, c.B), 256, 0, c.st>>>(hbuf, y, c.P(L.ln1_g), c.P(L.ln1_b), len, C, T,
                                                                          LN_GELU, c.P(L.dw_w), c.P(L.dw_b), d.K, L.dil);
        else if (C <= 256)
            ln_tile_kernel<1><<<dim3((T + 31) / 32, c.B), 256, 0, c.st>>>(hbuf, y, c.P(L.ln1_g), c.P(L.ln1_b), len, C, T,
                                                                          LN_GELU, c.P(L.dw_w), c.P(L.dw_b), d.K, L.dil);
        else
            dds_dw_ln_gelu_kernel<<<dim3((T + 63) / 64, c.B), 64, 0, c.st>>>(hbuf, y, c.P(L.dw_w), c.P(L.dw_b),
                                                                             c.P(L.ln1_g), c.P(L.ln1_b), len, C, T, d.K,
                                                                             L.dil);
        c.note(hipGetLastError());
        c.h->stats.total_launches++;
        conv(c, L.pw, y, (int64_t)C * T, T, y2, (int64_t)C * T, 0);
        int fl = LN_GELU | LN_ACCUM | (l == d.n_layers - 1 ? LN_MASK : 0);
        layernorm(c, y2, hbuf, L.ln2_g, L.ln2_b, len, C, T, fl);
    }
    return false;
}

void stage_mark(vits_handle *h, int idx) {
    if (h->timing) hipEventRecord(h->ev[idx], h->stream);
}

// bytes of the token-domain slab for B utterances of T tokens (run_tokens' plan)
size_t tokens_ws_bytes(const Model &m, int B, int T) {
    const int H = m.H, C = m.C;
    const size_t nHT = (size_t)B * H * T;
    const int Cdp = m.use_sdp ? m.dp_pre.Cout : m.dpp_F;
    size_t need = 0;
    need += al(nHT) * 4;                                   // x, attn out, embedding, spare
    need += al((size_t)B * 3 * H * T);                     // qkv
    need += al((size_t)B * m.FF * T);                      // ffn hidden
    if (m.enc_sx) need += 2 * al(nHT * 3 / 2 + 64) + al((size_t)B * m.FF * T * 3 / 2 + 64);  // operand planes: x, attn out, ffn hidden
    if (m.enc_sx && attention16_ok(m.dk, m.window)) need += al(nHT * 9 / 2 + 64);  // operand planes: q | k | v
    need += al((size_t)B * 2 * C * T) + 2 * al((size_t)B * C * T);  // stats, m_p, logs_p
    need += al((size_t)B * Cdp * T) * 5;                   // dp buffers
    int pr_rows = 32;  // spline parameters per position: 3 * bins - 1 (29 for the reference's 10 bins, up to 47)
    if (m.use_sdp)
        for (const auto &cfd : m.cf) pr_rows = cfd.proj.Cout > pr_rows ? cfd.proj.Cout : pr_rows;
    need += al((size_t)B * pr_rows * T) + al((size_t)B * 2 * T) * 2 + al((size_t)B * T) * 4;
    need += al((size_t)B * (m.gin + m.dp_cond_rows + m.C0 + 16)) + (1 << 16);
    for (auto &cd : m.flow) need += al((size_t)B * 2 * m.flow_H * cd.n_wn);
    return need;
}

// ---- token-domain part: encoder + duration predictor + durations.  Leaves y_len on the host.
int run_tokens(vits_handle *h, const int64_t *d_ids, const int64_t *d_lens, int B, int T, const float *scales,
               const int64_t *d_sid, const float *d_noise_dp, uint64_t seed) {
    const Model &m = h->model;
    const int H = m.H, C = m.C;
    // workspace plan (floats)
    const size_t nHT = (size_t)B * H * T;
    const int Cdp = m.use_sdp ? m.dp_pre.Cout : m.dpp_F;
    const size_t need = tokens_ws_bytes(m, B, T);
    int pr_rows = 32;  // spline parameters per position: 3 * bins - 1 (29 for the reference's 10 bins, up to 47)
    if (m.use_sdp)
        for (const auto &cfd : m.cf) pr_rows = cfd.proj.Cout > pr_rows ? cfd.proj.Cout : pr_rows;
    if (int rc = slab_reserve(h, h->tok, need)) return rc;
    Slab &s = h->tok;
    s.used = 0;
    Ctx c{h, m, h->stream, h->arena_dev, B};
    hipStream_t st = h->stream;

    h->d_len = slab_take<int>(s, B);
    h->d_ylen = slab_take<int>(s, B);
    h->d_ylen64 = slab_take<int64_t>(s, B);
    h->d_cum = slab_take<int>(s, (size_t)B * T);
    float *x = slab_take<float>(s, nHT), *att = slab_take<float>(s, nHT);
    // the embedded ids keep their own buffer (tap "emb": the integer gather, checked bit for bit); layer 0 reads it
    // and writes `x`, so nothing is copied
    float *xe = slab_take<float>(s, nHT);
    h->d_emb = xe;
    float *qkv = slab_take<float>(s, (size_t)B * 3 * H * T);
    float *ffh = slab_take<float>(s, (size_t)B * m.FF * T);
    float *stats = slab_take<float>(s, (size_t)B * 2 * C * T);
    h->d_x = x;
    h->d_logw = slab_take<float>(s, (size_t)B * T);
    h->d_wceil = slab_take<float>(s, (size_t)B * T);

    lens_to_i32<<<(B + 63) / 64, 64, 0, st>>>(d_lens, h->d_len, B, T);
    const int *len = h->d_len;
    h->cur_stage = 0;
    stage_mark(h, 0);
    embed_kernel<<<dim3((T + 63) / 64, (H + 15) / 16, B), 64, 0, st>>>(d_ids, len, c.P(m.emb), xe, H, T, m.n_vocab,
                                                        (float)std::sqrt((double)H));
    h->stats.total_launches += 2;
    const int64_t sHT = (int64_t)H * T;
    const float *xin = xe;  // layer input: the embedding for layer 0, x afterwards
    // split-operand engine (Model::enc_sx): every conv reads fp16 operand planes and writes planar fp32 (what attention,
    // LayerNorm and the duration predictor read) or planes for the next conv
    uint16_t *x_pl = nullptr, *att_pl = nullptr, *ff_pl = nullptr, *qkv_pl = nullptr;
    if (m.enc_sx) {
        x_pl = reinterpret_cast<uint16_t *>(slab_take<float>(s, nHT * 3 / 2 + 64));
        att_pl = reinterpret_cast<uint16_t *>(slab_take<float>(s, nHT * 3 / 2 + 64));
        ff_pl = reinterpret_cast<uint16_t *>(slab_take<float>(s, (size_t)B * m.FF * T * 3 / 2 + 64));
        if (attention16_ok(m.dk, m.window)) qkv_pl = reinterpret_cast<uint16_t *>(slab_take<float>(s, nHT * 9 / 2 + 64));
        split_planes(c, xe, x_pl, H, T, len);
    }
    for (auto &L : m.enc) {
        if (m.enc_sx) {
            const bool a16 = attention16_ok(m.dk, m.window);
            conv_sx_planar(c, L.qkv_sx, x_pl, T, a16 ? nullptr : qkv, a16 ? qkv_pl : nullptr, 0);  // (a16 reads the planes only)
            // (the attention kernel writes its output as conv_o's operand planes too when head widths are whole cells)
            uint16_t *apl = m.dk % 8 == 0 ? att_pl : nullptr;
            if (a16)
                c.note(launch_attention16(st, B, T, m.n_heads, m.dk, m.window, qkv_pl, nullptr, apl, c.P(L.rel_k), c.P(L.rel_v), len,
                                          H, range_slots(h, true)));
            else
                launch_attention(st, B, T, m.n_heads, m.dk, m.window, qkv, att, c.P(L.rel_k), c.P(L.rel_v), len, H, apl,
                                 apl ? range_slots(h, true) : nullptr);
            c.note(hipGetLastError());
            h->stats.total_launches++;
            h->stats.enc_flops += 2.0 * B * m.n_heads * (2.0 * m.dk * T * (double)T);
            if (!apl) split_planes(c, att, att_pl, H, T, nullptr);
            conv_sx_planar(c, L.o_sx, att_pl, T, x, nullptr, EPI_RES, nullptr, xin);
            xin = x;
            layernorm(c, x, x, L.ln1_g, L.ln1_b, len, H, T, LN_MASK, x_pl);
            conv_sx_planar(c, L.ffn1_sx, x_pl, T, nullptr, ff_pl, EPI_RELU | EPI_MASK, len);
            conv_sx_planar(c, L.ffn2_sx, ff_pl, T, x, nullptr, EPI_MASK | EPI_ACC, len);
            layernorm(c, x, x, L.ln2_g, L.ln2_b, len, H, T, LN_MASK, x_pl);
            continue;
        }
        // q|k|v = 1x1 convs (attentions.py:216-218), fused into one [3H,H] GEMM
        conv(c, L.qkv, xin, sHT, T, qkv, 3 * sHT, 0);
        launch_attention(st, B, T, m.n_heads, m.dk, m.window, qkv, att, c.P(L.rel_k), c.P(L.rel_v), len, H);
        c.note(hipGetLastError());
        h->stats.total_launches++;
        h->stats.enc_flops += 2.0 * B * m.n_heads * (2.0 * m.dk * T * (double)T);
        // x = LN(x + conv_o(att))  (attentions.py:66-68)
        conv(c, L.o, att, sHT, T, x, sHT, EPI_RES, nullptr, xin, sHT);
        xin = x;
        // Padded positions never reach valid ones (keys are masked, every other op is per-position or reads
        // x*mask), so masking the LayerNorm outputs changes no observable value and lets the FFN convs read
        // their input without a ragged mask (16-byte LDS-DMA path).
        layernorm(c, x, x, L.ln1_g, L.ln1_b, len, H, T, LN_MASK);
        // FFN (attentions.py:386-407): conv(x*mask) -> relu -> conv(h*mask) -> *mask ; x = LN(x + y)
        conv(c, L.ffn1, x, sHT, T, ffh, (int64_t)m.FF * T, EPI_RELU | EPI_MASK, len);
        conv(c, L.ffn2, ffh, (int64_t)m.FF * T, T, x, sHT, EPI_MASK | EPI_RES, len, x, sHT);
        layernorm(c, x, x, L.ln2_g, L.ln2_b, len, H, T, LN_MASK);
    }
    // x = x * mask ; stats = proj(x) * mask (models.py:205-208)
    mask_kernel<<<dim3((T + 255) / 256, H, B), 256, 0, st>>>(x, len, H, T);
    h->stats.total_launches++;
    if (m.enc_sx) conv_sx_planar(c, m.enc_proj_sx, x_pl, T, stats, nullptr, EPI_MASK, len);
    else conv(c, m.enc_proj, x, sHT, T, stats, (int64_t)2 * C * T, EPI_MASK, len);
    h->d_mp = stats;                       // [B, 2C, T] : m_p = rows [0,C), logs_p = rows [C,2C)
    h->d_logs = stats + (int64_t)C * T;    // batch stride 2*C*T for both

    // ---- speaker conditioning vectors
    float *dp_cond = nullptr;
    if (m.gin) {
        if (!d_sid) return fail(h, VITS_E_ARG, "Missing speaker id");
        dp_cond = slab_take<float>(s, (size_t)B * m.dp_cond_rows);
        cond_matvec_kernel<<<dim3((m.dp_cond_rows + 63) / 64, B), 64, 0, st>>>(
            c.P(m.emb_g), d_sid, m.n_speakers, c.P(m.dp_cond_w), c.P(m.dp_cond_b), dp_cond, m.dp_cond_rows, m.gin);
        h->stats.total_launches++;
    }

    // ---- duration predictor
    h->cur_stage = 1;
    stage_mark(h, 1);
    const float noise_w = scales[2];
    if (m.use_sdp) {
        const int Cd = m.dp_pre.Cout;
        const int64_t sC = (int64_t)Cd * T;
        float *hb = slab_take<float>(s, (size_t)B * Cd * T), *y = slab_take<float>(s, (size_t)B * Cd * T);
        float *y2 = slab_take<float>(s, (size_t)B * Cd * T), *cond = slab_take<float>(s, (size_t)B * Cd * T);
        float *h2 = slab_take<float>(s, (size_t)B * Cd * T);
        float *pr = slab_take<float>(s, (size_t)B * pr_rows * T);
        float *z = slab_take<float>(s, (size_t)B * 2 * T);
        // h = pre(x) [+ cond(g)] ; DDSConv ; cond = proj(h)*mask (models.py:65-70)
        conv(c, m.dp_pre, x, sHT, T, hb, sC, 0, nullptr, nullptr, 0, dp_cond, m.dp_cond_rows);
        if (!ddsconv(c, m.dp_convs, hb, y, y2, len, Cd, T, &m.dp_proj, m.dp_proj16, cond))
            conv(c, m.dp_proj, hb, sC, T, cond, sC, EPI_MASK, len);
        // z = randn * noise_scale_w (models.py:111)
        int64_t nz = (int64_t)B * 2 * T;
        if (noise_w == 0.f) {
            c.note(hipMemsetAsync(z, 0, nz * 4, st));
        } else if (d_noise_dp) {
            scale_kernel<<<(unsigned)((nz + 255) / 256), 256, 0, st>>>(d_noise_dp, z, noise_w, nz);
        } else {
            fill_normal_kernel<<<(unsigned)((nz / 4 + 256) / 256), 256, 0, st>>>(z, nz, seed, 1);
            scale_kernel<<<(unsigned)((nz + 255) / 256), 256, 0, st>>>(z, z, noise_w, nz);
        }
        h->stats.total_launches += 2;
        int swapped = 0;  // logical channel 0 lives in physical channel `swapped`
        for (int f = 0; f < 3; f++) {
            swapped ^= 1;  // Flip (modules.py:384-391)
            const auto &cf = m.cf[f];
            int ch0 = swapped, ch1 = swapped ^ 1;
            DdsHead hd{cond, z + (int64_t)ch0 * T, c.P(cf.pre_w), c.P(cf.pre_b)};
            const bool use_head = dds16_head_ok(cf.convs, Cd);
            if (!use_head) {
                cf_pre_kernel<<<dim3((T + 255) / 256, Cd, B), 256, 0, st>>>(z, ch0, c.P(cf.pre_w), c.P(cf.pre_b), cond, h2, Cd, T);
                h->stats.total_launches++;
            }
            if (!ddsconv(c, cf.convs, h2, y, y2, len, Cd, T, &cf.proj, cf.proj16, pr, use_head ? &hd : nullptr))
                conv(c, cf.proj, h2, sC, T, pr, (int64_t)cf.proj.Cout * T, EPI_MASK, len);
            float sqc = std::sqrt((float)Cd);
            if (cf.nb <= 10)
                rqs_inverse_kernel<10><<<dim3((T + 63) / 64, B), 64, 0, st>>>(pr, z, len, ch0, ch1, cf.nb, T, sqc);
            else
                rqs_inverse_kernel<16><<<dim3((T + 63) / 64, B), 64, 0, st>>>(pr, z, len, ch0, ch1, cf.nb, T, sqc);
            c.note(hipGetLastError());
            h->stats.total_launches++;
        }
        swapped ^= 1;
        ea_logw_kernel<<<dim3((T + 63) / 64, B), 64, 0, st>>>(z, swapped, m.ea_m0, m.ea_logs0, len, h->d_logw, T);
        h->stats.total_launches++;
    } else {
        const int Fd = m.dpp_F;
        float *xi = x;
        if (dp_cond) {  // x = x + cond(g) (models.py:153-155)
            xi = slab_take<float>(s, nHT);
            add_bias_b_kernel<<<dim3((T + 255) / 256, H, B), 256, 0, st>>>(x, xi, dp_cond, m.dp_cond_rows, H, T);
            h->stats.total_launches++;
        }
        float *h1 = slab_take<float>(s, (size_t)B * Fd * T), *h2 = slab_take<float>(s, (size_t)B * Fd * T);
        const int64_t sF = (int64_t)Fd * T;
        conv(c, m.dpp_conv1, xi, sHT, T, h1, sF, PRO_MASK | EPI_RELU, len);
        layernorm(c, h1, h1, m.dpp_n1_g, m.dpp_n1_b, len, Fd, T, 0);
        conv(c, m.dpp_conv2, h1, sF, T, h2, sF, PRO_MASK | EPI_RELU, len);
        layernorm(c, h2, h2, m.dpp_n2_g, m.dpp_n2_b, len, Fd, T, 0);
        conv(c, m.dpp_proj, h2, sF, T, h->d_logw, T, PRO_MASK | EPI_MASK, len);
    }
    // ---- durations (models.py:702-704)
    duration_kernel<<<B, 256, 0, st>>>(h->d_logw, len, scales[1], h->d_wceil, h->d_cum, h->d_ylen, T);
    h->stats.total_launches++;
    c.note(hipGetLastError());
    if (c.err != hipSuccess) return fail(h, VITS_E_DEVICE, "kernel launch failed: %s", hipGetErrorString(c.err));
    h->h_ylen.resize(B);
    if (h->h_ylen_pin_n < B) {
        if (h->h_ylen_pin) hipHostFree(h->h_ylen_pin);
        h->h_ylen_pin = nullptr;
        h->h_ylen_pin_n = 0;
        const int cap = B < 64 ? 64 : B;
        if (hipHostMalloc((void **)&h->h_ylen_pin, sizeof(int) * cap) == hipSuccess) h->h_ylen_pin_n = cap;
    }
    int *ydst = h->h_ylen_pin_n >= B ? h->h_ylen_pin : h->h_ylen.data();
    HIPCHECK(h, hipMemcpyAsync(ydst, h->d_ylen, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    HIPCHECK(h, hipStreamSynchronize(st));  // the one data-dependent readback: output length
    if (ydst != h->h_ylen.data()) std::memcpy(h->h_ylen.data(), ydst, sizeof(int) * B);
    if (h->range_pending) {  // an earlier asynchronous run whose range verdict nobody has looked at
        const vits_stats keep = h->stats;
        const int rr = range_check(h);
        h->stats = keep;
        if (rr) return fail(h, VITS_E_RANGE, "the previous run on this handle left the range of the fp16 operand planes");
    }
    int F = 1;
    for (int b = 0; b < B; b++) F = h->h_ylen[b] > F ? h->h_ylen[b] : F;
    h->F = F;
    return 0;
}

// frame-domain layout: flow buffers + generator ping-pong regions
size_t gen_region_floats(const Model &m, int B, int F) {
    size_t mx = (size_t)B * (m.C0 > m.C ? m.C0 : m.C) * ((F + 3) & ~3);
    int64_t t = F;
    for (auto &st : m.ups) {
        t *= st.u;
        size_t n = (size_t)B * st.C * t;
        mx = n > mx ? n : mx;
    }
    return mx;
}

constexpr int kGenRegions = 10;     // f32 engine: ten fp32 regions
constexpr int kGenRegionsSx = 15;   // sx engine: 4 raw + 6 plane tensors (1.5 regions each) + the waveform, rounded up

// bytes of generator workspace (the largest tensor decides the region size)
size_t gen_ws_bytes(const Model &m, int B, int F) {
    return (size_t)(m.gen_sx ? kGenRegionsSx : kGenRegions) * al(gen_region_floats(m, B, F));
}

// Ragged rendering of a padded batch (vits_handle::tails_reference == false): from here on every generator launch ends
// utterance b's tensors gen_rf_frames behind ylen[b] (Ctx::rag_at), and the tail kernel writes zeros behind ylen[b] * hop.
// Needs the host copy of the frame counts (Ctx::h_len) for the FLOP / byte accounting; B = 1 has no padding.
// the margin of the following launches (frames behind an utterance's end that its tensors still cover), and with it the
// share of B * F frames they work on
void rag_margin(Ctx &c, int add) {
    if (!c.rag.len) return;
    const int B = c.B, F = c.rag_F;
    double cols = 0;
    for (int b = 0; b < B; b++) {
        const int n = c.h_len[b];
        cols += n > 0 ? (n + add < F ? n + add : F) : 0;
    }
    c.rag.add = add;
    c.rag_frac = cols / ((double)B * F);
}
void rag_begin(vits_handle *h, Ctx &c, const int *ylen, int B, int F) {
    c.rag = SxRagged{nullptr, 0, 0};
    c.rag_F = F;
    c.rag_frac = 1.0;
    if (!ylen || !c.h_len || B < 2 || B != c.B || h->tails_reference) return;
    c.rag = SxRagged{ylen, 0, 1};
    rag_margin(c, h->model.gen_rf_frames);  // (conv_pre: the whole receptive field; the stages set their own, smaller ones)
}
// the frame counts the tail kernel zeroes behind (nullptr: the reference's padded rendering is kept as it is)
const int *tail_len(const vits_handle *h, const int *ylen) { return h->tails_reference ? nullptr : ylen; }

// The PLANE-STREAM generator of the two fp16 arithmetics: f16x3 (the default: two fp16 planes per operand, three MFMA
// products per fp32 product) and f16 (VITSMI_GEN_PRECISION=f16, BASELINE config 4's reduced-precision vocoder: one plane, one
// product).  Every tensor between two convs exists ONCE, as the operand planes of the consumer's leaky_relu
// ([planes][C/8][T][8]: 4 / 2 bytes per element): a conv reads them as its B operand straight from LDS (by LDS-DMA), and
// a residual add recovers x from them by undoing the leaky_relu (x = p >= 0 ? p : p / 0.1; exact up to the planes' own
// resolution - 22 bits in f16x3).  No fp32 copy of the residual stream is written or read; only the multi-receptive-field
// sum xs (models.py:356-363; three read-modify-writes per stage) is fp32.  Everything in front of z is unchanged.  Same
// dataflow as run_generator_sx (which now serves the exact bf16x6 arithmetic only).
int run_generator_planes(vits_handle *h, Ctx &c, const float *z, int64_t z_bstride, int z_cstride, const int *ylen, int B,
                     int F, const float *dec_cond, Slab &s) {
    const Model &m = h->model;
    hipStream_t st = h->stream;
    const size_t R = gen_region_floats(m, B, F);
    const size_t RP = R + R / 2 + 64;  // (plane tensors keep the three-slot batch stride of the other modes)
    auto planes = [&]() { return reinterpret_cast<uint16_t *>(slab_take<float>(s, RP)); };
    uint16_t *stage_in[2] = {planes(), planes()}, *y_pl = planes(), *raa[2] = {planes(), planes()}, *tmp_pl = planes();
    float *xs_raw = slab_take<float>(s, R);
    h->cur_stage = 3;
    stage_mark(h, 3);
    rag_begin(h, c, ylen, B, F);
    const float S = 0.1f;  // Generator.LRELU_SLOPE / ResBlock LRELU_SLOPE
    const int nst = (int)m.ups.size();
    // z * y_mask (models.py:349) as conv_pre's operand plane (no activation in front of conv_pre)
    sx_split_planes_kernel<<<dim3((F + 255) / 256, m.C / 8, B), 256, 0, st>>>(z, z_bstride, z_cstride, ylen, tmp_pl, m.C, F,
                                                                              m.gen_h1 ? 2 : 1, range_slots(h, true));
    c.note(hipGetLastError());
    h->stats.total_launches++;
    // xa = leaky_relu(conv_pre(z) [+ cond(g)], 0.1) (models.py:349-354)
    conv_sx(c, m.conv_pre, tmp_pl, F, nullptr, stage_in[0], 0, nullptr, dec_cond, m.C0, 1.f, 1.f, S);
    const uint16_t *xa = stage_in[0];
    int T = F;
    for (int si = 0; si < nst; si++) {
        const auto &stg = m.ups[si];
        rag_margin(c, m.gen_rf_stage[si]);  // (what is left of the receptive field from this stage's input on)
        // y = up(xa), stored as leaky_relu(y, 0.1): the resblocks' first operand and, un-activated, their residual
        conv_sx(c, stg.up, xa, T, nullptr, y_pl, 0, nullptr, nullptr, 0, 1.f, 1.f, S);
        T *= stg.u;
        uint16_t *xs_pl = stage_in[(si + 1) & 1];
        const int nk = (int)stg.rbs.size();
        const bool last_stage = si == nst - 1;
        for (int j = 0; j < nk; j++) {
            const auto &rbk = stg.rbs[j];
            const uint16_t *cur = y_pl;
            const bool final_rb = j == nk - 1;
            for (int q = 0; q < rbk.n; q++) {
                const bool last = q == rbk.n - 1;
                int fl = EPI_RES;
                float *dst = nullptr;          // fp32 destination: the running sum xs
                uint16_t *dsta = raa[q & 1];   // plane destination: the block's stream
                if (last) {
                    // xs = rb0(x); xs += rb1(x); ...; x = xs / nk.  The stage output feeds the next upsampler as a plane
                    // (leaky_relu 0.1); the last stage's feeds conv_post as fp32 (its kernel applies leaky_relu 0.01)
                    fl |= (j == 0 ? 0 : EPI_ACC) | (final_rb && nk > 1 ? EPI_DIV : 0);
                    dst = xs_raw;
                    dsta = nullptr;
                    if (final_rb && !last_stage) {
                        dsta = xs_pl;
                        if (nk > 1) fl |= SX_NO_RAW_STORE;
                        else dst = nullptr;
                    }
                }
                // flags of a fused launch: the multi-receptive-field arithmetic + which of the two destinations it writes
                auto p16_flags = [&](int f, float *d_raw, uint16_t *d_pl) {
                    return (f & (EPI_ACC | EPI_DIV)) | (d_raw && !(f & SX_NO_RAW_STORE) ? P16_HAS_RAW : 0) | (d_pl ? P16_HAS_PL : 0);
                };
                if (rbk.type1) {  // modules.py:301-314: x = c2(lrelu(c1(lrelu(x)))) + x
                    if (sx_pair16_ok(h, rbk.c1[q], rbk.c2[q]))
                        conv_sx_pair16(c, rbk.c1[q], rbk.c2[q], cur, T, dst, dsta, p16_flags(fl, dst, dsta), (float)nk, S, false);
                    else {
                        conv_sx(c, rbk.c1[q], cur, T, nullptr, tmp_pl, 0, nullptr, nullptr, 0, 1.f, 1.f, S);
                        conv_sx(c, rbk.c2[q], tmp_pl, T, dst, dsta, fl, nullptr, nullptr, 0, (float)nk, 1.f, S, 1.f, nullptr, cur, S);
                    }
                } else if (q + 1 < rbk.n && sx_pair16_ok(h, rbk.c1[q], rbk.c1[q + 1])) {
                    // modules.py:355-364, two steps in one launch: x1 = c(lrelu(x)) + x ; x = c'(lrelu(x1)) + x1
                    q++;
                    const bool last2 = q == rbk.n - 1;
                    int fl2 = EPI_RES;
                    float *dst2 = nullptr;
                    uint16_t *dsta2 = raa[q & 1];
                    if (last2) {
                        fl2 |= (j == 0 ? 0 : EPI_ACC) | (final_rb && nk > 1 ? EPI_DIV : 0);
                        dst2 = xs_raw;
                        dsta2 = nullptr;
                        if (final_rb && !last_stage) {
                            dsta2 = xs_pl;
                            if (nk > 1) fl2 |= SX_NO_RAW_STORE;
                            else dst2 = nullptr;
                        }
                    }
                    conv_sx_pair16(c, rbk.c1[q - 1], rbk.c1[q], cur, T, dst2, dsta2, p16_flags(fl2, dst2, dsta2), (float)nk, S, true);
                    dsta = dsta2;
                } else  // modules.py:355-364: x = c(lrelu(x)) + x
                    conv_sx(c, rbk.c1[q], cur, T, dst, dsta, fl, nullptr, nullptr, 0, (float)nk, 1.f, S, 1.f, nullptr, cur, S);
                cur = dsta;
            }
        }
        xa = xs_pl;
    }
    // leaky_relu(0.01), conv_post, tanh (models.py:364-366) from the fp32 stage output
    h->S = T;
    h->d_out = slab_take<float>(s, (size_t)B * T);
    {
        const size_t lds = (size_t)m.post_cin * (256 + m.post_k - 1) * sizeof(float);
        if (m.post_k == 7)
            post_conv_tanh_blocked_kernel<7><<<dim3((T + 255) / 256, B), 256, lds, st>>>(xs_raw, c.P(m.post_w), h->d_out,
                                                                                        m.post_cin, m.post_k, T, 0.01f,
                                                                                        tail_len(h, ylen), T / F);
        else
            post_conv_tanh_blocked_kernel<0><<<dim3((T + 255) / 256, B), 256, lds, st>>>(xs_raw, c.P(m.post_w), h->d_out,
                                                                                        m.post_cin, m.post_k, T, 0.01f,
                                                                                        tail_len(h, ylen), T / F);
    }
    c.note(hipGetLastError());
    h->stats.total_launches++;
    {
        double fl = 2.0 * m.post_cin * m.post_k * (double)T * B * c.work_frac(), by = 4.0 * B * ((double)m.post_cin * T + T) * c.work_frac();
        h->stats.dec_flops += fl;
        h->stats.dec_bytes += by;
    }
    c.rag = SxRagged{nullptr, 0, 0};
    stage_mark(h, 4);
    return 0;
}

// f16x3, plane-format stages: residual stream as operand planes only (run_generator_sx); VITSMI_F16X3_RES=raw for the fp32 one
bool res_planes_on() {  // (read per run: tests switch it inside one process)
    // default ON (r05e, same box, two runs each: 138.9 / 137.4 M samples/s against 136.3 / 136.3 with the fp32 residual stream;
    // 128-channel k = 3 residual conv 652 -> 588-600 us, stride-8 upsampler 983 -> 830 us, k = 11 residual convs +2-3 %)
    const char *e = std::getenv("VITSMI_F16X3_RES");
    return e ? std::string(e) != "raw" : true;
}

// The generator on the split-operand engine.  Same dataflow as run_generator below; tensors that feed a conv
// are stored as 16-bit planes (already leaky-ReLU'd by their producer), the residual stream as fp32 raw cells.
int run_generator_sx(vits_handle *h, Ctx &c, const float *z, int64_t z_bstride, int z_cstride, const int *ylen, int B,
                     int F, const float *dec_cond, Slab &s) {
    const Model &m = h->model;
    if (m.gen_planes) return run_generator_planes(h, c, z, z_bstride, z_cstride, ylen, B, F, dec_cond, s);
    hipStream_t st = h->stream;
    const size_t R = gen_region_floats(m, B, F);
    const size_t RP = R + R / 2 + 64;  // floats holding R elements as three 16-bit plane slots (the fp16 mode uses two)
    auto planes = [&]() { return reinterpret_cast<uint16_t *>(slab_take<float>(s, RP)); };
    uint16_t *stage_in[2] = {planes(), planes()}, *y_pl = planes(), *raa[2] = {planes(), planes()}, *tmp_pl = planes();
    float *y_raw = slab_take<float>(s, R), *ra[2] = {slab_take<float>(s, R), slab_take<float>(s, R)};
    float *xs_raw = slab_take<float>(s, R);
    // the plane regions double as fp32 raw buffers where a stage uses the raw format (RP >= R floats)
    float *tmp_raw = reinterpret_cast<float *>(tmp_pl), *xin_raw = reinterpret_cast<float *>(stage_in[0]);
    h->cur_stage = 3;
    stage_mark(h, 3);
    rag_begin(h, c, ylen, B, F);
    const float S = 0.1f;  // Generator.LRELU_SLOPE / ResBlock LRELU_SLOPE
    const int nst = (int)m.ups.size();
    // Tensor formats (model.hpp sx_raw_format): > 64 channels: 16-bit planes that already carry the consumer's
    // leaky_relu (+ fp32 raw where the tensor is also a residual); <= 64 channels: fp32 raw only, the consuming
    // conv applies the leaky_relu and the split while loading (its `islope`).
    // ---- z * y_mask (models.py:349) in conv_pre's input format
    const void *zin;
    if (sx_raw_format(m.C)) {
        sx_block_kernel<<<dim3((F + 255) / 256, m.C / 8, B), 256, 0, st>>>(z, z_bstride, z_cstride, ylen, tmp_raw, m.C, F,
                                                                           range_slots(h, m.gen_f16));
        zin = tmp_raw;
    } else {
        sx_split_planes_kernel<<<dim3((F + 255) / 256, m.C / 8, B), 256, 0, st>>>(z, z_bstride, z_cstride, ylen, tmp_pl, m.C, F,
                                                                                  m.gen_f16 ? 1 : 0, range_slots(h, m.gen_f16));
        zin = tmp_pl;
    }
    c.note(hipGetLastError());
    h->stats.total_launches++;
    // ---- xa = conv_pre(z) [+ cond(g)]; leaky_relu(0.1) follows (models.py:349-354)
    bool xa_is_raw = sx_raw_format(m.C0);
    const void *xa;
    if (xa_is_raw) {
        conv_sx(c, m.conv_pre, zin, F, xin_raw, nullptr, 0, nullptr, dec_cond, m.C0);
        xa = xin_raw;
    } else {
        conv_sx(c, m.conv_pre, zin, F, nullptr, stage_in[0], 0, nullptr, dec_cond, m.C0, 1.f, 1.f, S);
        xa = stage_in[0];
    }
    int T = F;
    for (int si = 0; si < nst; si++) {
        const auto &stg = m.ups[si];
        rag_margin(c, m.gen_rf_stage[si]);  // (what is left of the receptive field from this stage's input on)
        const bool fr = sx_raw_format(stg.C);  // format of this stage's tensors
        // Plane-format stages (> 64 channels) in f16x3: the residual stream as operand planes ONLY (SX_RES_PL: a residual is
        // recovered from the planes of leaky_relu(x), 22 bits, as in run_generator_planes) - no fp32 copy of y and of the
        // blocks' intermediate x is written or read: 12 instead of 16 bytes per element on the residual convs, which at 128
        // channels and k = 3 are HBM-bound (launch table r05c: 0.65 ms = 4.9 TB/s of real traffic).  Needs every conv of the
        // stage on the 16x16x32 loop (the SX_RES_PL instantiations); VITSMI_F16X3_RES=raw keeps the fp32 residual stream (A/B).
        bool rpl = !fr && h->gen_nprod == 2 && res_planes_on() && stg.up.s16;
        for (const auto &rbk : stg.rbs)
            for (int q = 0; q < rbk.n && rpl; q++) rpl = rbk.c1[q].s16 && rbk.c1[q].f16 && (!rbk.type1 || (rbk.c2[q].s16 && rbk.c2[q].f16));
        // y = up(leaky_relu(xa)): pixel-shuffled dense conv; raw (residual / raw-format input) [+ planes]
        conv_sx(c, stg.up, xa, T, rpl ? nullptr : y_raw, fr ? nullptr : y_pl, 0, nullptr, nullptr, 0, 1.f, 1.f, S, S);
        T *= stg.u;
        uint16_t *xs_pl = stage_in[(si + 1) & 1];
        const int nk = (int)stg.rbs.size();
        const bool last_stage = si == nst - 1;
        if (fr && sx_mrf_ok(h, stg)) {
            // every ResBlock2 of the stage in one launch: x read once, the sum formed in registers, one store
            conv_sx_mrf(c, stg, y_raw, T, xs_raw, S);
            xa_is_raw = true;
            xa = xs_raw;
            continue;
        }
        for (int j = 0; j < nk; j++) {
            const auto &rbk = stg.rbs[j];
            const float *cur = y_raw;          // residual operand
            const uint16_t *cura = y_pl;       // plane format: leaky_relu(cur) as planes
            // MRF accumulation (models.py:356-363): xs = rb0(x); xs += rb1(x); ... ; x = xs / nk
            const bool final_rb = j == nk - 1;
            for (int q = 0; q < rbk.n; q++) {
                const bool last = q == rbk.n - 1;
                int fl = EPI_RES;
                float *dst = ra[q & 1];
                uint16_t *dsta = fr ? nullptr : raa[q & 1];
                if (last) {
                    fl |= (j == 0 ? 0 : EPI_ACC) | (final_rb && nk > 1 ? EPI_DIV : 0);
                    dst = xs_raw;
                    dsta = nullptr;
                    // the stage output x = xs / nk feeds the next upsampler (leaky_relu 0.1) or conv_post (0.01):
                    // planes carry the activation, raw tensors get it from their consumer
                    if (final_rb && !last_stage && !fr) {
                        dsta = xs_pl;
                        fl |= SX_NO_RAW_STORE;
                    }
                }
                const void *in = fr ? static_cast<const void *>(cur) : static_cast<const void *>(cura);
                if (rpl) {
                    // (the block's stream exists as planes only: an inner step writes no raw tensor, every step takes its
                    // residual from the planes of its own input)
                    if (!last) dst = nullptr;
                    if (rbk.type1) {
                        conv_sx(c, rbk.c1[q], in, T, nullptr, tmp_pl, 0, nullptr, nullptr, 0, 1.f, 1.f, S);
                        conv_sx(c, rbk.c2[q], tmp_pl, T, dst, dsta, fl, nullptr, nullptr, 0, (float)nk, 1.f, S, 1.f, nullptr, cura, S);
                    } else
                        conv_sx(c, rbk.c1[q], in, T, dst, dsta, fl, nullptr, nullptr, 0, (float)nk, 1.f, S, 1.f, nullptr, cura, S);
                    cur = dst;
                    cura = dsta;
                    continue;
                }
                if (rbk.type1) {  // modules.py:301-314: x = c2(lrelu(c1(lrelu(x)))) + x
                    if (fr && sx_pair_ok(h, rbk.c1[q], rbk.c2[q])) {
                        // both convs in one launch: the intermediate stays in LDS, x is read once
                        conv_sx_pair(c, rbk.c1[q], rbk.c2[q], cur, T, dst, fl, (float)nk, S);
                    } else if (fr) {
                        conv_sx(c, rbk.c1[q], in, T, tmp_raw, nullptr, 0, nullptr, nullptr, 0, 1.f, 1.f, 1.f, S);
                        conv_sx(c, rbk.c2[q], tmp_raw, T, dst, nullptr, fl, cur, nullptr, 0, (float)nk, 1.f, 1.f, S);
                    } else {
                        conv_sx(c, rbk.c1[q], in, T, nullptr, tmp_pl, 0, nullptr, nullptr, 0, 1.f, 1.f, S);
                        conv_sx(c, rbk.c2[q], tmp_pl, T, dst, dsta, fl, cur, nullptr, 0, (float)nk, 1.f, S);
                    }
                } else if (fr && q + 1 < rbk.n && sx_pair_ok(h, rbk.c1[q], rbk.c1[q + 1])) {
                    // modules.py:355-364, two steps in one launch: x1 = c(lrelu(x)) + x ; x = c'(lrelu(x1)) + x1
                    q++;
                    const bool last2 = q == rbk.n - 1;
                    int fl2 = EPI_RES;
                    float *dst2 = ra[q & 1];
                    if (last2) {
                        fl2 |= (j == 0 ? 0 : EPI_ACC) | (final_rb && nk > 1 ? EPI_DIV : 0);
                        dst2 = xs_raw;
                    }
                    conv_sx_pair(c, rbk.c1[q - 1], rbk.c1[q], cur, T, dst2, fl2, (float)nk, S, /*chain=*/true);
                    dst = dst2;
                    dsta = nullptr;
                } else  // modules.py:355-364: x = c(lrelu(x)) + x
                    conv_sx(c, rbk.c1[q], in, T, dst, dsta, fl, cur, nullptr, 0, (float)nk, 1.f, S, S);
                cur = dst;
                cura = dsta;
            }
        }
        // next stage input: the planes written by the final conv, or the raw x = xs / nk itself
        xa_is_raw = fr || last_stage;
        xa = (fr || last_stage) ? static_cast<const void *>(xs_raw) : static_cast<const void *>(xs_pl);
    }
    // leaky_relu(0.01), conv_post, tanh (models.py:364-366) from the raw stage output
    h->S = T;
    h->d_out = slab_take<float>(s, (size_t)B * T);
    {
        const size_t lds = (size_t)m.post_cin * (256 + m.post_k - 1) * sizeof(float);
        if (m.post_k == 7)
            post_conv_tanh_blocked_kernel<7><<<dim3((T + 255) / 256, B), 256, lds, st>>>(xs_raw, c.P(m.post_w), h->d_out,
                                                                                        m.post_cin, m.post_k, T, 0.01f,
                                                                                        tail_len(h, ylen), T / F);
        else
            post_conv_tanh_blocked_kernel<0><<<dim3((T + 255) / 256, B), 256, lds, st>>>(xs_raw, c.P(m.post_w), h->d_out,
                                                                                        m.post_cin, m.post_k, T, 0.01f,
                                                                                        tail_len(h, ylen), T / F);
    }
    c.note(hipGetLastError());
    h->stats.total_launches++;
    {
        double fl = 2.0 * m.post_cin * m.post_k * (double)T * B * c.work_frac(), by = 4.0 * B * ((double)m.post_cin * T + T) * c.work_frac();
        h->stats.dec_flops += fl;
        h->stats.dec_bytes += by;
    }
    c.rag = SxRagged{nullptr, 0, 0};
    stage_mark(h, 4);
    (void)xa_is_raw;
    return 0;
}

int run_generator(vits_handle *h, Ctx &c, const float *z, int64_t z_bstride, int z_cstride, const int *ylen, int B,
                  int F, const float *dec_cond, Slab &s) {
    const Model &m = h->model;
    if (m.gen_sx) return run_generator_sx(h, c, z, z_bstride, z_cstride, ylen, B, F, dec_cond, s);
    hipStream_t st = h->stream;
    const size_t R = gen_region_floats(m, B, F);
    float *reg[kGenRegions];
    for (int i = 0; i < kGenRegions; i++) reg[i] = slab_take<float>(s, R);
    h->cur_stage = 3;
    stage_mark(h, 3);
    // Every leaky_relu of the generator (models.py:354,364; modules.py:303,307,357) is applied by the
    // PRODUCER's epilogue, so no conv carries activation math in its MFMA loop (conv_engine.hip.hpp).
    // A tensor that is needed both raw (residual) and activated (next conv input) is stored twice.
    const float S = 0.1f;  // Generator.LRELU_SLOPE / ResBlock LRELU_SLOPE
    const int nst = (int)m.ups.size();
    // xa = leaky_relu(conv_pre(z * y_mask) [+ cond(g)], 0.1)            (models.py:349-354)
    float *xa = reg[0];
    const int Fp = (F + 3) & ~3;  // pitch of conv_pre's output rows: lets ups[0] use the 16-byte DMA path
    conv(c, m.conv_pre, z, z_bstride, F, xa, (int64_t)m.C0 * Fp, ylen ? PRO_MASK : 0, ylen, nullptr, 0, dec_cond, m.C0,
         1.f, 1.f, S, nullptr, 1.f, z_cstride, Fp);
    int T = F, Cc = m.C0, xs_idx = 0;
    int in_pitch = Fp;
    for (int si = 0; si < nst; si++) {
        const auto &stg = m.ups[si];
        // y = up(xa) as a pixel-shuffled dense conv; stored raw (residual) and activated (conv input)
        float *y = reg[2], *ya = reg[3];
        const int To = T * stg.u;
        conv(c, stg.up, xa, (int64_t)Cc * in_pitch, T, y, (int64_t)stg.C * To, 0, nullptr, nullptr, 0, nullptr, 0, 1.f, 1.f,
             1.f, ya, S, in_pitch, 0);
        in_pitch = To;
        T = To;
        Cc = stg.C;
        const int64_t sCT = (int64_t)Cc * T;
        xs_idx ^= 1;
        float *xs = reg[xs_idx];
        float *ra[2] = {reg[4], reg[6]}, *raa[2] = {reg[5], reg[7]}, *tmp = reg[8];
        const int nk = (int)stg.rbs.size();
        // slope applied to the stage output: 0.1 before the next upsampler, 0.01 before conv_post (:364)
        const float out_slope = si == nst - 1 ? 0.01f : S;
        for (int j = 0; j < nk; j++) {
            const auto &rbk = stg.rbs[j];
            const float *cur = y, *cura = ya;
            // MRF accumulation (models.py:356-363): xs = rb0(x); xs += rb1(x); ... ; x = xs / nk
            const bool final_rb = j == nk - 1;
            const int last_flags = (j == 0 ? 0 : EPI_ACC) | (final_rb && nk > 1 ? EPI_DIV : 0);
            const float last_oslope = final_rb ? out_slope : 1.f;
            for (int q = 0; q < rbk.n; q++) {
                const bool last = q == rbk.n - 1;
                float *dst = last ? xs : ra[q & 1];
                float *dsta = last ? nullptr : raa[q & 1];
                const int fl = EPI_RES | (last ? last_flags : 0);
                const float osl = last ? last_oslope : 1.f;
                if (rbk.type1) {  // modules.py:301-314: x = c2(lrelu(c1(lrelu(x)))) + x
                    conv(c, rbk.c1[q], cura, sCT, T, tmp, sCT, 0, nullptr, nullptr, 0, nullptr, 0, 1.f, 1.f, S);
                    conv(c, rbk.c2[q], tmp, sCT, T, dst, sCT, fl, nullptr, cur, sCT, nullptr, 0, 1.f, (float)nk, osl, dsta,
                         S);
                } else {  // modules.py:355-364: x = c(lrelu(x)) + x
                    conv(c, rbk.c1[q], cura, sCT, T, dst, sCT, fl, nullptr, cur, sCT, nullptr, 0, 1.f, (float)nk, osl, dsta,
                         S);
                }
                cur = dst;
                cura = dsta;
            }
        }
        xa = xs;  // already activated for its only consumer
    }
    float *x = xa;
    // x = leaky_relu(x) [slope 0.01]; conv_post; tanh (models.py:364-366)
    h->S = T;
    h->d_out = reg[9];
    size_t lds = ((size_t)m.post_cin * (256 + m.post_k - 1) + (size_t)m.post_cin * m.post_k) * sizeof(float);
    // (the leaky_relu(0.01) of models.py:364 was applied by the last stage's epilogue)
    post_conv_tanh_kernel<<<dim3((T + 255) / 256, B), 256, lds, st>>>(x, c.P(m.post_w), h->d_out, m.post_cin, m.post_k,
                                                                      T, 1.0f, tail_len(h, ylen), T / F);
    c.note(hipGetLastError());
    h->stats.total_launches++;
    {
        double fl = 2.0 * m.post_cin * m.post_k * (double)T * B * c.work_frac(), by = 4.0 * B * ((double)m.post_cin * T + T) * c.work_frac();
        h->stats.dec_flops += fl;
        h->stats.dec_bytes += by;
    }
    c.rag = SxRagged{nullptr, 0, 0};
    stage_mark(h, 4);
    return 0;
}

__global__ void chunk_len_kernel(const int *ylen, int *out, int B, int lo, int n) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        const int v = ylen ? ylen[b] - lo : n;  // (no lengths: every frame of the chunk is valid)
        out[b] = v < 0 ? 0 : (v > n ? n : v);
    }
}

// Frames [0, F) of z rendered in chunks of sink->chunk_frames: each chunk is rendered together with gen_rf_frames of
// context on either side (clipped at the utterance ends, where the generator really sees zero padding) and only its
// interior is kept, so every sample equals the one an unchunked run produces (the convolutions accumulate in the same
// order wherever a column sits in a tile).  Finished chunks go to the host through two pinned buffers; the callback
// for chunk i runs while chunk i + 1 renders.
int render_chunks(vits_handle *h, Ctx &c, const float *z, int64_t z_bstride, int z_cstride, const int *ylen, int B, int F,
                  const float *dec_cond, Slab &s, const ChunkSink &sink) {
    const Model &m = h->model;
    hipStream_t st = h->stream;
    const int ov = m.gen_rf_frames, hop = m.hop;
    const int chunk = sink.chunk_frames;
    const size_t need = (size_t)B * chunk * hop * sizeof(float);
    if (need > h->ring_cap) {
        for (auto &r : h->ring) {
            if (r) hipHostFree(r);
            r = nullptr;
        }
        h->ring_cap = 0;
        for (auto &r : h->ring)
            if (hipHostMalloc((void **)&r, need) != hipSuccess) return fail(h, VITS_E_NOMEM, "pinned chunk buffer (%zu bytes)", need);
        h->ring_cap = need;
    }
    for (auto &e : h->ring_ev)
        if (!e) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    // per-chunk valid lengths, ALWAYS passed: a chunk is a window into rows whose neighbours are real data, and only a
    // length mask makes the conv engines bound their reads by the chunk instead of by the row pitch
    int *yl = slab_take<int>(s, B);
    const size_t mark = s.used;
    const int *const whole_len = c.h_len;  // host copy of ylen (or nullptr)
    const int64_t total = (int64_t)F * hop;
    int64_t pend_first = 0, pend_n = 0;
    int pend = -1, k = 0, stop = 0;
    auto deliver = [&]() -> int {  // hand the pending chunk to the caller
        if (pend < 0) return 0;
        if (hipEventSynchronize(h->ring_ev[pend]) != hipSuccess) return fail(h, VITS_E_DEVICE, "chunk copy failed");
        stop = sink.fn ? sink.fn(sink.user, reinterpret_cast<const float *>(h->ring[pend]), B, pend_first, pend_n, total) : 0;
        pend = -1;
        return 0;
    };
    for (int f0 = 0; f0 < F && !stop; f0 += chunk, k ^= 1) {
        const int f1 = f0 + chunk < F ? f0 + chunk : F;
        const int lo = f0 - ov > 0 ? f0 - ov : 0, hi = f1 + ov < F ? f1 + ov : F, n = hi - lo;
        s.used = mark;  // the previous chunk's workspace (its copy-out precedes this chunk on the stream)
        chunk_len_kernel<<<(B + 63) / 64, 64, 0, st>>>(ylen, yl, B, lo, n);
        // (host mirror of chunk_len_kernel for the ragged accounting; without frame counts every frame is valid)
        std::vector<int> hyl((size_t)B, n);
        if (ylen && whole_len)
            for (int b = 0; b < B; b++) hyl[b] = whole_len[b] - lo < 0 ? 0 : (whole_len[b] - lo > n ? n : whole_len[b] - lo);
        c.h_len = ylen && !whole_len ? nullptr : hyl.data();
        const int rc_gen = run_generator(h, c, z + lo, z_bstride, z_cstride, yl, B, n, dec_cond, s);
        c.h_len = whole_len;
        if (rc_gen) return rc_gen;
        if (c.err != hipSuccess) return fail(h, VITS_E_DEVICE, "kernel launch failed: %s", hipGetErrorString(c.err));
        const int64_t ns = (int64_t)(f1 - f0) * hop;
        // interior of this chunk: samples [(f0 - lo) * hop, (f1 - lo) * hop) of every row -> ring[k] as [B, ns]
        HIPCHECK(h, hipMemcpy2DAsync(h->ring[k], (size_t)ns * 4, h->d_out + (int64_t)(f0 - lo) * hop, (size_t)h->S * 4,
                                     (size_t)ns * 4, B, hipMemcpyDeviceToHost, st));
        HIPCHECK(h, hipEventRecord(h->ring_ev[k], st));
        if (int rc = deliver()) return rc;  // (the previous chunk, while this one renders)
        pend = k;
        pend_first = (int64_t)f0 * hop;
        pend_n = ns;
    }
    if (!stop)
        if (int rc = deliver()) return rc;
    h->S = (int)total;
    h->d_out = nullptr;  // (no whole waveform exists on the device after a chunked run)
    return 0;
}

// bytes of the frame-domain slab (flow + generator) for B utterances of F frames (a multiple of 4), the generator rendering
// Fgen frames at a time (run_frames' plan)
size_t frames_ws_bytes(const Model &m, int B, int F, int Fgen) {
    const size_t nCF = (size_t)B * m.C * F, nHF = (size_t)B * m.flow_H * F;
    size_t need = al(nCF) * 3 + al(nHF) * 3 + al(nHF * 2) * 2 + al(nHF * 2) + gen_ws_bytes(m, B, Fgen) + (1 << 16);
    for (auto &cd : m.flow) need += al((size_t)B * 2 * m.flow_H * cd.n_wn);
    need += al((size_t)B * m.C0);
    need += al(nCF) + al(nHF * 2);  // operand planes of a coupling's x0 and of its skip sum (pre / post on the split-operand engine)
    return need;
}

int run_frames(vits_handle *h, int B, int T, const float *scales, const int64_t *d_sid, const float *d_noise_z,
               int64_t noise_z_stride, uint64_t seed, const ChunkSink *sink = nullptr) {
    const Model &m = h->model;
    const int C = m.C, Freal = h->F, Hf = m.flow_H;
    if (d_noise_z && noise_z_stride < Freal)
        return fail(h, VITS_E_ARG, "noise_z has %lld frames per row but %d are needed", (long long)noise_z_stride, Freal);
    // The flow runs on F rounded up to a multiple of 4 frames: every tensor in it is masked by y_len, so the
    // extra (masked) frames change nothing, and all its rows become 16-byte aligned for the conv engine's
    // 16-byte LDS-DMA.  The generator, which is NOT masked, runs on exactly Freal frames and reads z through
    // its row pitch.
    const int F = (Freal + 3) & ~3;
    h->Fpitch = F;
    const size_t nCF = (size_t)B * C * F, nHF = (size_t)B * Hf * F;
    // frames the generator renders at a time: everything, or one chunk with its context
    const int Fgen = sink && sink->chunk_frames + 2 * m.gen_rf_frames < F ? sink->chunk_frames + 2 * m.gen_rf_frames : F;
    if (int rc = slab_reserve(h, h->frm, frames_ws_bytes(m, B, F, Fgen))) return rc;
    Slab &s = h->frm;
    s.used = 0;
    Ctx c{h, m, h->stream, h->arena_dev, B};
    if ((int)h->h_ylen.size() == B) c.h_len = h->h_ylen.data();
    hipStream_t st = h->stream;
    const int *len = h->d_len, *ylen = h->d_ylen;
    float *zp = slab_take<float>(s, nCF), *z = slab_take<float>(s, nCF);
    h->d_zp = zp;
    h->d_z = z;
    h->cur_stage = 2;
    stage_mark(h, 2);
    {
        // Ragged flow: only when EVERY launch of the flow is a conv_sx() launch (pre / gated in-layer / res_skip / post on the
        // split-operand engine: every f16x3 voice) - a kernel without the per-utterance end would read what its predecessor
        // did not write.  Masked either way, so results are those of the padded form, bit for bit.
        static const bool off = std::getenv("VITSMI_NO_RAGGED_FLOW") != nullptr;  // A/B timing
        bool all_sx = !off && B > 1 && c.h_len && (C / 2) % 32 == 0 && Hf % 32 == 0;
        for (const auto &cd : m.flow) {
            all_sx = all_sx && cd.pre_sx.sx && cd.post_sx.sx && cd.n_wn > 0;
            for (int i = 0; i < cd.n_wn && all_sx; i++)
                all_sx = cd.wn[i].in.sx && cd.wn[i].in.f16 && cd.wn[i].in.gate && cd.wn[i].rs_sx.sx;
        }
        if (all_sx) {
            double fr = 0;
            for (int b = 0; b < B; b++) fr += c.h_len[b] < F ? c.h_len[b] : F;
            c.flow_len = ylen;
            c.flow_frac = fr / ((double)B * F);
        }
    }
    const float noise_scale = scales[0];
    const float *nz = nullptr;
    int64_t nzs = F;
    if (noise_scale != 0.f) {
        if (d_noise_z) {
            nz = d_noise_z;
            nzs = noise_z_stride;
        } else {
            float *g = slab_take<float>(s, nCF);
            fill_normal_kernel<<<(unsigned)((nCF / 4 + 256) / 256), 256, 0, st>>>(g, (int64_t)nCF, seed, 2);
            h->stats.total_launches++;
            nz = g;
        }
    }
    // m_p / logs_p are the two halves of the proj output: channel stride T, batch stride 2*C*T
    expand_prior_strided_kernel<<<dim3((F + 63) / 64, (C + 15) / 16, B), 64, 0, st>>>(h->d_mp, h->d_logs, (int64_t)2 * C * T, h->d_cum,
                                                                       len, ylen, nz, nzs, noise_scale, zp, C, T, F,
                                                                       nz == d_noise_z ? Freal : F);
    h->stats.total_launches++;
    if (c.flow_len) {
        // z = z_p * y_mask: the ragged flow leaves the frames behind an utterance's end alone, where the reference's couplings
        // zero them one half at a time ((x1 - m) * mask, modules.py:464) - same z wherever it is valid, and the zeros the
        // reference ends up with behind
        masked_copy_kernel<<<dim3((F + 255) / 256, C, B), 256, 0, st>>>(zp, z, ylen, C, F);
        h->stats.total_launches++;
    } else
        HIPCHECK(h, hipMemcpyAsync(z, zp, nCF * 4, hipMemcpyDeviceToDevice, st));

    // ---- inverse coupling flow (models.py:247-254, modules.py:447-466); Flips folded at pack time
    float *hx = slab_take<float>(s, nHF), *skip = slab_take<float>(s, nHF), *acts = slab_take<float>(s, nHF);
    float *a2 = slab_take<float>(s, nHF * 2), *rs = slab_take<float>(s, nHF * 2);
    uint16_t *hx_pl = reinterpret_cast<uint16_t *>(slab_take<float>(s, nHF * 2));  // planes of hx (sx in-layers)
    const int half = C / 2;
    const int64_t sCF = (int64_t)C * F, sHF = (int64_t)Hf * F;
    uint16_t *acts_pl = reinterpret_cast<uint16_t *>(a2);  // (a2 is unused where the gate writes operand planes)
    // pre / post on the split-operand engine (CouplingDesc::pre_sx): planes of x0 (written by the previous coupling's post:
    // its x1 is this one's x0; split here for the first) and of the skip sum (written by the last res_skip conv)
    uint16_t *x0_pl = reinterpret_cast<uint16_t *>(slab_take<float>(s, nCF));
    uint16_t *skip_pl = reinterpret_cast<uint16_t *>(slab_take<float>(s, nHF * 2));
    bool x0_planes_ready = false;
    for (size_t ci = 0; ci < m.flow.size(); ci++) {
        const auto &cd = m.flow[ci];
        const bool pp = cd.pre_sx.sx && cd.post_sx.sx && half % 32 == 0;
        const bool pp_next = ci + 1 < m.flow.size() && m.flow[ci + 1].pre_sx.sx && m.flow[ci + 1].post_sx.sx;
        float *gc = nullptr;
        const int gc_rows = 2 * Hf * cd.n_wn;
        if (m.gin) {
            gc = slab_take<float>(s, (size_t)B * gc_rows);
            cond_matvec_kernel<<<dim3((gc_rows + 63) / 64, B), 64, 0, st>>>(c.P(m.emb_g), d_sid, m.n_speakers, c.P(cd.cond_w),
                                                                           c.P(cd.cond_b), gc, gc_rows, m.gin);
            h->stats.total_launches++;
        }
        const float *x0 = z + (cd.swapped ? (int64_t)half * F : 0);
        float *x1 = z + (cd.swapped ? 0 : (int64_t)half * F);
        // Launches per WN layer: [in-layer conv on the split engine] [tanh * sigmoid gate] [res_skip 1x1 conv].  The
        // res_skip conv's epilogue applies the update (x = (x + res) * mask, skip += ...: EPI_WN) and, when the next
        // in-layer runs on the split engine, also emits x as that conv's fp16 operand planes - as `pre` does for the
        // first layer - so neither an update nor a split launch is left (they were 2 of the 5 launches of a layer).
        const bool planes0 = cd.n_wn > 0 && cd.wn[0].in.sx && cd.wn[0].in.f16 && Hf % 32 == 0;
        ConvExtra ex0;
        ex0.out_pl = planes0 ? hx_pl : nullptr;
        ex0.pl_rows = Hf;
        // h = pre(x0) * mask
        if (pp) {
            if (!x0_planes_ready) {
                sx_split_planes_kernel<<<dim3((F + 255) / 256, half / 8, B), 256, 0, st>>>(x0, sCF, F, nullptr, x0_pl, half, F, 1,
                                                                                        range_slots(h, true));
                h->stats.total_launches++;
            }
            conv_sx_planar(c, cd.pre_sx, x0_pl, F, hx, planes0 ? hx_pl : nullptr, EPI_MASK, ylen);
        } else
            conv(c, cd.pre, x0, sCF, F, hx, sHF, EPI_MASK, ylen, nullptr, 0, nullptr, 0, 0.1f, 1.f, 1.f, nullptr, 1.f, 0, 0, &ex0);
        x0_planes_ready = false;
        for (int i = 0; i < cd.n_wn; i++) {
            const bool last = i == cd.n_wn - 1;
            // x_in = in_layer(h) + g_l ; acts = tanh * sigmoid ; rs = res_skip(acts)
            if (cd.wn[i].in.sx) {
                // split-operand engine: planes of hx (from the producing epilogue, or split here), conv to the raw cell
                // layout, gate reads that layout
                const bool have_planes = cd.wn[i].in.f16 && Hf % 32 == 0;
                if (!have_planes) {
                    sx_split_planes_kernel<<<dim3((F + 255) / 256, Hf / 8, B), 256, 0, st>>>(hx, sHF, F, nullptr, hx_pl, Hf, F,
                                                                                             cd.wn[i].in.f16 ? 1 : 0,
                                                                                             range_slots(h, cd.wn[i].in.f16));
                    h->stats.total_launches++;
                }
                if (cd.wn[i].in.gate && cd.wn[i].rs_sx.sx) {
                    // gate in the in-layer's epilogue, acts handed over as fp16 operand planes; the res_skip 1 x 1 conv on
                    // the same engine reads them and folds the update in: x += res * mask (+ the next in-layer's planes),
                    // skip += .. (the first layer stores it)
                    conv_sx(c, cd.wn[i].in, hx_pl, F, nullptr, acts_pl, SX_GATE, nullptr, gc ? gc + (int64_t)i * 2 * Hf : nullptr,
                            gc_rows);
                    SxWn w;
                    w.len = ylen;
                    w.out_raw2 = skip;
                    w.row_split = last ? 0 : Hf;
                    const bool np = !last && cd.wn[i + 1].in.sx && cd.wn[i + 1].in.f16 && Hf % 32 == 0;
                    w.pl_rows = np ? Hf : 0;
                    uint16_t *rs_pl = np ? hx_pl : nullptr;
                    if (last && pp) {  // the finished skip sum as post's operand planes
                        w.pl_rows = Hf;
                        w.pl_of2 = true;
                        rs_pl = skip_pl;
                    }
                    // (the first layer stores its skip rows: no zero fill of skip, no read of it)
                    conv_sx(c, cd.wn[i].rs_sx, acts_pl, F, hx, rs_pl,
                            SX_WN_RMW | EPI_ACC | EPI_MASK | (i == 0 ? SX_PLANAR_STORE2 : 0), nullptr, nullptr, 0, 1.f, 1.f, 1.f, 1.f, &w);
                    continue;
                } else if (cd.wn[i].in.gate) {  // tanh * sigmoid in the conv's epilogue: acts directly
                    conv_sx(c, cd.wn[i].in, hx_pl, F, acts, nullptr, SX_GATE, nullptr, gc ? gc + (int64_t)i * 2 * Hf : nullptr,
                            gc_rows);
                    h->stats.total_launches--;  // (no gate launch: undo the increment below)
                } else {
                    conv_sx(c, cd.wn[i].in, hx_pl, F, a2, nullptr, 0, nullptr, gc ? gc + (int64_t)i * 2 * Hf : nullptr, gc_rows);
                    wn_gate_blocked_kernel<<<dim3((F + 255) / 256, Hf / 8, B), 256, 0, st>>>(a2, acts, Hf, F);
                }
            } else {
                conv(c, cd.wn[i].in, hx, sHF, F, a2, 2 * sHF, 0, nullptr, nullptr, 0,
                     gc ? gc + (int64_t)i * 2 * Hf : nullptr, gc_rows);
                wn_gate_kernel<<<dim3((F + 255) / 256, Hf, B), 256, 0, st>>>(a2, acts, Hf, F);
            }
            h->stats.total_launches++;
            // res_skip conv + update: rows [0, Hf) -> hx (residual half), rows [Hf, 2 Hf) -> skip; last layer: skip only
            ConvExtra ex;
            ex.wn_split = last ? 0 : Hf;
            const bool next_planes = !last && cd.wn[i + 1].in.sx && cd.wn[i + 1].in.f16 && Hf % 32 == 0;
            ex.out_pl = next_planes ? hx_pl : nullptr;
            ex.pl_rows = Hf;
            conv(c, cd.wn[i].rs, acts, sHF, F, hx, sHF, EPI_WN | EPI_MASK | (i == 0 ? EPI_WN_FIRST : 0), ylen, nullptr, 0,
                 nullptr, 0, 0.1f, 1.f, 1.f, skip, 1.f, 0, 0, &ex);
        }
        // x1 = (x1 - post(skip)*mask) * mask
        if (pp) {
            SxWn w;
            w.len = ylen;
            w.row_split = half;
            w.planar_bstride = sCF;           // (x1's rows live inside z)
            w.pl_rows = pp_next ? half : 0;   // ... and are the next coupling's x0: its operand planes
            conv_sx(c, cd.post_sx, skip_pl, F, x1, pp_next ? x0_pl : nullptr, SX_WN_RMW | EPI_ACC | EPI_MASK | SX_PLANAR_COUPLING,
                    nullptr, nullptr, 0, 1.f, 1.f, 1.f, 1.f, &w);
            x0_planes_ready = pp_next;
        } else
            conv(c, cd.post, skip, sHF, F, x1, sCF, EPI_COUPLING, ylen);
    }
    c.note(hipGetLastError());

    float *dec_cond = nullptr;
    if (m.gin) {
        dec_cond = slab_take<float>(s, (size_t)B * m.C0);
        cond_matvec_kernel<<<dim3((m.C0 + 63) / 64, B), 64, 0, st>>>(c.P(m.emb_g), d_sid, m.n_speakers, c.P(m.dec_cond_w),
                                                                    c.P(m.dec_cond_b), dec_cond, m.C0, m.gin);
        h->stats.total_launches++;
    }
    if (sink) return render_chunks(h, c, z, sCF, F, ylen, B, Freal, dec_cond, s, *sink);
    if (int rc = run_generator(h, c, z, sCF, F, ylen, B, Freal, dec_cond, s)) return rc;
    if (c.err != hipSuccess) return fail(h, VITS_E_DEVICE, "kernel launch failed: %s", hipGetErrorString(c.err));
    return 0;
}

}  // namespace

// ================================================================================== C ABI

static int open_common(const char *path, vits_handle **out, bool host_only, int device, void *ext_arena,
                       size_t ext_bytes, bool layout_only = false) {
    if (!out || !path) return fail(nullptr, VITS_E_ARG, "null argument");
    *out = nullptr;
    OnnxModel om;
    std::string e = om.load(path);
    if (!e.empty()) return fail(nullptr, e.rfind("cannot open", 0) == 0 ? VITS_E_IO : VITS_E_FORMAT, "%s", e.c_str());
    vits_handle *h = new vits_handle();
    // a handle that adopts a resident arena only needs the layout (offsets, descriptors): no weight is re-packed
    e = h->model.build(om, /*layout_only=*/ext_arena != nullptr || layout_only);
    if (!e.empty()) {
        delete h;
        return fail(nullptr, VITS_E_FORMAT, "%s: %s", path, e.c_str());
    }
    if (h->model.window > 4) {
        delete h;
        return fail(nullptr, VITS_E_FORMAT, "attention window %d > 4 is unsupported", h->model.window);
    }
    h->host_only = host_only;
    {
        const char *te = std::getenv("VITSMI_TAILS");  // "reference": the graph's padded rendering (see vits_set_tails)
        h->tails_reference = te && std::string(te) == "reference";
    }
    {
        // Arithmetic of the generator's convs on the split-exact engine (fp32 operands and results in every mode):
        //   f16x3  (default) two fp16 planes per operand, three MFMA products: each product within ~3 * 2^-24
        //   bf16x6 three bf16 planes, six products: each product exact to 2^-24
        //   f16    the reduced-precision vocoder of BASELINE config 4: ONE fp16 plane per operand, one product, fp32
        //          accumulation, generator activations stored as fp16 (everything in front of z unchanged)
        // gen_nprod: 2 = f16x3, 1 = f16 (Model::build packed the weights for either), 6 = the six bf16 plane products.
        const char *pe = gen_precision_name();
        const std::string ps = pe ? pe : "";
        if (ps.empty() || ps == "f16x3") h->gen_nprod = 2;
        else if (ps == "bf16x6") h->gen_nprod = 6;
        else if (ps == "f16") h->gen_nprod = 1;
        else {
            delete h;
            return fail(nullptr, VITS_E_ARG, "VITSMI_GEN_PRECISION must be f16x3, bf16x6 or f16 (got '%s')", pe);
        }
        if (!h->model.gen_sx) {
            // (a generator the split-operand engine cannot take - a channel count that is not a multiple of 32 - runs on the f32
            // engine: fine for the two fp32-grade requests, whose results it matches, but NOT a reduced-precision vocoder: an
            // explicit "f16" request is refused with the reason, as a conv that fails the single-plane constraints already is)
            if (h->gen_nprod == 1) {
                delete h;
                return fail(nullptr, VITS_E_ARG, "gen_precision \"f16\" needs a generator on the split-operand engine (every channel "
                                                 "count a multiple of 32); this voice's runs on the f32 engine");
            }
            h->gen_nprod = 6;
        }
    }
    if (!host_only) {
        int n = 0;
        hipError_t er = hipGetDeviceCount(&n);
        if (er != hipSuccess || device < 0 || device >= n) {
            delete h;
            return fail(nullptr, VITS_E_DEVICE, "no usable HIP device %d (count %d): %s", device, n,
                        er == hipSuccess ? "index out of range" : hipGetErrorString(er));
        }
        h->device = device;
        if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
            delete h;
            return fail(nullptr, VITS_E_DEVICE, "cannot create stream on device %d", device);
        }
        size_t bytes = (size_t)h->model.arena_floats * 4;
        if (ext_arena) {
            if (ext_bytes != bytes) {
                delete h;
                return fail(nullptr, VITS_E_ARG, "arena size mismatch: got %zu, model needs %zu", ext_bytes, bytes);
            }
            h->arena_dev = (float *)ext_arena;
        } else {
            if (hipMalloc((void **)&h->arena_dev, bytes) != hipSuccess ||
                hipMemcpy(h->arena_dev, h->model.arena.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
                delete h;
                return fail(nullptr, VITS_E_NOMEM, "cannot place %zu-byte weight arena on device %d", bytes, device);
            }
            h->arena_owned = true;
        }
        for (auto &e2 : h->ev) hipEventCreate(&e2);
        if (!std::getenv("VITSMI_NO_RANGE_GUARD")) {  // range guard of the fp16 operand planes (generator, flow WN convs);
                                                     // the switch exists for A/B timing of its cost only
            const size_t nb = (size_t)kMaxRangeLaunches * kSxPeakSlots * kSxPeakStride * sizeof(unsigned);
            if (hipMalloc((void **)&h->d_range, nb + 64) != hipSuccess || hipHostMalloc((void **)&h->h_range, 64) != hipSuccess) {
                vits_close(h);
                return fail(nullptr, VITS_E_NOMEM, "cannot allocate the range-guard buffers");
            }
            h->d_range_res = reinterpret_cast<float *>(reinterpret_cast<char *>(h->d_range) + nb);
            hipMemset(h->d_range, 0, nb + 64);
            std::memset(h->h_range, 0, 64);
        }
    }
    *out = h;
    return VITS_OK;
}

extern "C" {

int vits_open(const char *p, int dev, vits_handle **out) { return open_common(p, out, false, dev, nullptr, 0); }
int vits_open_with_arena(const char *p, int dev, void *arena, size_t bytes, vits_handle **out) {
    if (!arena) return fail(nullptr, VITS_E_ARG, "null arena");
    return open_common(p, out, false, dev, arena, bytes);
}
int vits_open_host(const char *p, vits_handle **out) { return open_common(p, out, true, -1, nullptr, 0); }
int vits_open_layout(const char *p, vits_handle **out) { return open_common(p, out, true, -1, nullptr, 0, true); }

int vits_open_opts(const char *p, const vits_open_options *o, vits_handle **out) {
    if (!o) return fail(nullptr, VITS_E_ARG, "null options");
    if (o->arena_dev && o->host_only) return fail(nullptr, VITS_E_ARG, "host_only excludes arena_dev");
    set_gen_precision_override(o->gen_precision && *o->gen_precision ? o->gen_precision : nullptr);
    const int rc = open_common(p, out, o->host_only != 0, o->host_only ? -1 : o->device_id, o->arena_dev, o->arena_bytes,
                               o->layout_only != 0);
    set_gen_precision_override(nullptr);
    return rc;
}

void vits_close(vits_handle *h) {
    if (!h) return;
    if (!h->host_only) {
        hipSetDevice(h->device);
        if (h->stream) hipStreamSynchronize(h->stream);
        if (h->arena_owned && h->arena_dev) hipFree(h->arena_dev);
        if (h->h_ylen_pin) hipHostFree(h->h_ylen_pin);
        if (h->h_in_pin) hipHostFree(h->h_in_pin);
        if (h->tok.base) hipFree(h->tok.base);
        if (h->frm.base) hipFree(h->frm.base);
        if (h->io.base) hipFree(h->io.base);
        if (h->pin.base) hipHostFree(h->pin.base);
        if (h->d_range) hipFree(h->d_range);
        if (h->h_range) hipHostFree(h->h_range);
        for (auto &r : h->ring)
            if (r) hipHostFree(r);
        for (auto &e : h->ring_ev)
            if (e) hipEventDestroy(e);
        for (auto &p : h->conv_events) {
            hipEventDestroy(p.first);
            hipEventDestroy(p.second);
        }
        for (auto &e : h->ev)
            if (e) hipEventDestroy(e);
        if (h->stream) hipStreamDestroy(h->stream);
    }
    delete h;
}

const char *vits_last_error(vits_handle *h) { return h ? h->err.c_str() : g_open_error.c_str(); }

int vits_num_inputs(vits_handle *h) { return h ? (int)h->model.input_names.size() : 0; }
const char *vits_input_name(vits_handle *h, int i) {
    if (!h || i < 0 || i >= (int)h->model.input_names.size()) return nullptr;
    return h->model.input_names[i].c_str();
}

int vits_meta(vits_handle *h, const char *key, char *buf, size_t n) {
    if (!h || !key) return VITS_E_ARG;
    auto it = h->model.meta.find(key);
    if (it == h->model.meta.end()) return fail(h, VITS_E_ARG, "no metadata key %s", key);
    if (buf && n) {
        size_t k = it->second.size() < n - 1 ? it->second.size() : n - 1;
        std::memcpy(buf, it->second.data(), k);
        buf[k] = 0;
    }
    return (int)it->second.size();
}

int vits_hparam(vits_handle *h, const char *key, int64_t *out) {
    if (!h || !key || !out) return VITS_E_ARG;
    const Model &m = h->model;
    std::string k = key;
    if (k == "hidden") *out = m.H;
    else if (k == "inter") *out = m.C;
    else if (k == "filter") *out = m.FF;
    else if (k == "n_heads") *out = m.n_heads;
    else if (k == "n_layers") *out = m.n_layers;
    else if (k == "n_vocab") *out = m.n_vocab;
    else if (k == "n_speakers") *out = m.n_speakers;
    else if (k == "gin") *out = m.gin;
    else if (k == "gen_sx") *out = m.gen_sx ? 1 : 0;
    else if (k == "gen_nprod") *out = h->gen_nprod;
    else if (k == "enc_sx") *out = m.enc_sx ? 1 : 0;
    else if (k == "use_sdp") *out = m.use_sdp;
    else if (k == "hop") *out = m.hop;
    else if (k == "n_ups") *out = (int64_t)m.ups.size();
    else if (k == "resblock") *out = m.ups.empty() || m.ups[0].rbs.empty() ? 0 : (m.ups[0].rbs[0].type1 ? 1 : 2);
    else if (k == "window") *out = m.window;
    else if (k == "gen_rf_frames") *out = m.gen_rf_frames;
    else if (k == "upsample_initial_channel") *out = m.C0;
    else if (k == "dec_macs_per_frame") *out = (int64_t)m.dec_macs_per_frame;
    else if (k == "flow_macs_per_frame") *out = (int64_t)m.flow_macs_per_frame;
    else if (k == "enc_macs_per_token") *out = (int64_t)m.enc_macs_per_token;
    else if (k == "dec_bytes_per_frame") *out = (int64_t)(m.dec_elems_per_frame * 4);
    else if (k == "workspace_bytes") *out = (int64_t)(h->tok.cap + h->frm.cap + h->io.cap);
    else return fail(h, VITS_E_ARG, "unknown hparam %s", key);
    return VITS_OK;
}

size_t vits_arena_bytes(vits_handle *h) { return h ? (size_t)h->model.arena_floats * 4 : 0; }
const void *vits_arena_host(vits_handle *h) { return h && !h->model.arena.empty() ? h->model.arena.data() : nullptr; }
void *vits_arena_device(vits_handle *h) { return h ? h->arena_dev : nullptr; }
void *vits_stream(vits_handle *h) { return h ? (void *)h->stream : nullptr; }

int vits_set_timing(vits_handle *h, int enable) {
    if (!h) return VITS_E_ARG;
    h->timing = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    return VITS_OK;
}

int vits_set_tails(vits_handle *h, int reference) {
    if (!h) return VITS_E_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    h->tails_reference = reference != 0;
    return VITS_OK;
}

static int check_dev(vits_handle *h) {
    if (!h) return VITS_E_ARG;
    if (h->host_only) return fail(h, VITS_E_DEVICE, "handle was opened host-only");
    if (hipSetDevice(h->device) != hipSuccess) return fail(h, VITS_E_DEVICE, "hipSetDevice(%d) failed", h->device);
    return 0;
}

int vits_reserve(vits_handle *h, int B, int T, int F) {
    if (int rc = check_dev(h)) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    const Model &m = h->model;
    if (B <= 0 || T < 0 || F < 0) return fail(h, VITS_E_ARG, "vits_reserve: B=%d T=%d F=%d", B, T, F);
    if (T > 0) {
        if (int rc = slab_reserve(h, h->tok, tokens_ws_bytes(m, B, T), false)) return rc;
        // the staging slab: ids | lens | sid, and injected noises where a caller passes them (vits_noise)
        // ... or, between runs, vits_last_pcm16's int16 waveform and per-utterance peaks (the larger of the two uses)
        const size_t io_in = (size_t)B * T * 8 + (size_t)B * 16 + (size_t)B * 2 * T * 4 + (size_t)B * m.C * (size_t)((F + 3) & ~3) * 4 + 4096;
        const size_t io_pcm = (((size_t)B * F * m.hop * 2 + 255) & ~size_t(255)) + (size_t)B * 4 + 256;
        const size_t io = io_in > io_pcm ? io_in : io_pcm;
        if (int rc = slab_reserve(h, h->io, io, false)) return rc;
    }
    if (F > 0) {
        const int Fp = (F + 3) & ~3;
        if (int rc = slab_reserve(h, h->frm, frames_ws_bytes(m, B, Fp, Fp), false)) return rc;
    }
    return VITS_OK;
}

static int run_device_locked(vits_handle *h, const int64_t *ids, const int64_t *lens, int B, int T,
                             const float scales[3], const int64_t *sid, const vits_noise *noise, vits_output *out,
                             const ChunkSink *sink = nullptr) {
    if (B <= 0 || T <= 0) return fail(h, VITS_E_ARG, "empty batch or sequence (B=%d, T=%d)", B, T);
    if (!ids || !lens || !scales || (!out && !sink)) return fail(h, VITS_E_ARG, "null argument");
    if (h->model.gin && !sid) return fail(h, VITS_E_ARG, "Missing speaker id");
    std::memset(&h->stats, 0, sizeof h->stats);
    h->conv_events_used = 0;
    g_launch_name_on = h->timing == 1;
    h->B = B;
    h->T = T;
    h->range_failed = false;
    uint64_t seed = noise ? noise->seed : 0;
    seed = seed * 0x9E3779B97F4A7C15ull + (++h->run_counter);
    // (run_tokens' one synchronisation also completes the previous run on this handle: its range verdict, if nobody
    // asked for it yet, is looked at there, before this run's guard slots are cleared)
    if (int rc = run_tokens(h, ids, lens, B, T, scales, sid, noise ? noise->noise_dp : nullptr, seed)) return rc;
    range_begin(h);
    if (int rc = run_frames(h, B, T, scales, sid, noise ? noise->noise_z : nullptr, noise ? noise->noise_z_stride : 0,
                            seed, sink))
        return rc;
    ylen_to_i64<<<(B + 63) / 64, 64, 0, h->stream>>>(h->d_ylen, h->d_ylen64, B);
    range_end(h);
    if (out) {
        out->data = h->d_out;
        out->dims[0] = B;
        out->dims[1] = 1;
        out->dims[2] = 1;
        out->dims[3] = h->S;
        out->y_lengths = h->d_ylen64;
    }
    return VITS_OK;
}

int vits_run_device(vits_handle *h, const int64_t *ids, const int64_t *lens, int B, int T, const float scales[3],
                    const int64_t *sid, const vits_noise *noise, vits_output *out) {
    if (int rc = check_dev(h)) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    if (!out) return fail(h, VITS_E_ARG, "null argument");
    return run_device_locked(h, ids, lens, B, T, scales, sid, noise, out);
}

int vits_last_y_lengths(vits_handle *h, int64_t *buf, int n) {
    if (!h) return VITS_E_ARG;
    int B = (int)h->h_ylen.size();
    for (int b = 0; b < B && b < n && buf; b++) buf[b] = h->h_ylen[b];
    return B;
}

int vits_sync(vits_handle *h) {
    if (int rc = check_dev(h)) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    if (int rc = range_check(h)) return rc;
    if (h->range_failed) return fail(h, VITS_E_RANGE, "the last run left the range of the fp16 operand planes (see vits_get_stats)");
    return VITS_OK;
}

// vits_run / vits_run_chunked: validate the host inputs and stage them on the device (one slab, reused across calls)
struct Staged {
    int64_t *d_ids = nullptr, *d_lens = nullptr, *d_sid = nullptr;
    vits_noise dn{};
    bool has_noise = false;
};

static int stage_inputs(vits_handle *h, const int64_t *ids, const int64_t *lens, int B, int T, const int64_t *sid,
                        const vits_noise *noise, Staged &sg) {
    if (B <= 0 || T <= 0) return fail(h, VITS_E_ARG, "empty batch or sequence (B=%d, T=%d)", B, T);
    if (!ids || !lens) return fail(h, VITS_E_ARG, "null argument");
    const Model &m = h->model;
    for (int b = 0; b < B; b++) {
        if (lens[b] < 0 || lens[b] > T)
            return fail(h, VITS_E_ARG, "input_lengths[%d]=%lld outside [0,%d]", b, (long long)lens[b], T);
        for (int t = 0; t < T; t++) {
            int64_t id = ids[(int64_t)b * T + t];
            if (id < 0 || id >= m.n_vocab)
                return fail(h, VITS_E_ARG, "phoneme id %lld at [%d,%d] is out of range [0,%d)", (long long)id, b, t,
                            m.n_vocab);
        }
        if (sid && m.gin && (sid[b] < 0 || sid[b] >= m.n_speakers))
            return fail(h, VITS_E_ARG, "sid[%d]=%lld is out of range [0,%d)", b, (long long)sid[b], m.n_speakers);
    }
    if (m.gin && !sid) return fail(h, VITS_E_ARG, "Missing speaker id");
    size_t nb = (size_t)B * T * 8 + (size_t)B * 16 + 1024;
    size_t ndp = noise && noise->noise_dp ? (size_t)B * 2 * T * 4 : 0;
    size_t nz = noise && noise->noise_z ? (size_t)B * m.C * (size_t)noise->noise_z_stride * 4 : 0;
    if (int rc0 = slab_reserve(h, h->io, nb + ndp + nz + 1024)) return rc0;
    char *stage = h->io.base;
    sg.d_ids = (int64_t *)stage;
    sg.d_lens = sg.d_ids + (size_t)B * T;
    int64_t *d_sid = sg.d_lens + B;
    float *d_ndp = (float *)(stage + ((nb + 255) & ~size_t(255)));
    float *d_nz = (float *)((char *)d_ndp + ((ndp + 255) & ~size_t(255)));
    hipStream_t st = h->stream;
    int rc = VITS_OK;
    auto cp = [&](void *d, const void *s, size_t n) {
        if (rc == VITS_OK && hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st) != hipSuccess)
            rc = fail(h, VITS_E_DEVICE, "host-to-device copy failed");
    };
    {
        // ids | lens | sid are contiguous on the device: gather them in pinned memory and send them as one copy
        const size_t n_ids = (size_t)B * T * 8, n_b = (size_t)B * 8, n_all = n_ids + n_b + (sid ? n_b : 0);
        if (h->h_in_pin_bytes < n_all) {
            if (h->h_in_pin) hipHostFree(h->h_in_pin);
            h->h_in_pin = nullptr;
            h->h_in_pin_bytes = 0;
            const size_t cap = n_all < 65536 ? 65536 : n_all;
            if (hipHostMalloc((void **)&h->h_in_pin, cap) == hipSuccess) h->h_in_pin_bytes = cap;
        }
        if (h->h_in_pin_bytes >= n_all) {
            // (the previous call's copy out of this buffer has completed: every call ends its input phase with a stream sync)
            std::memcpy(h->h_in_pin, ids, n_ids);
            std::memcpy(h->h_in_pin + n_ids, lens, n_b);
            if (sid) std::memcpy(h->h_in_pin + n_ids + n_b, sid, n_b);
            cp(sg.d_ids, h->h_in_pin, n_all);
        } else {
            cp(sg.d_ids, ids, n_ids);
            cp(sg.d_lens, lens, n_b);
            if (sid) cp(d_sid, sid, n_b);
        }
        if (sid) sg.d_sid = d_sid;
    }
    if (noise) {
        sg.has_noise = true;
        sg.dn = *noise;
        if (noise->noise_dp) {
            cp(d_ndp, noise->noise_dp, ndp);
            sg.dn.noise_dp = d_ndp;
        }
        if (noise->noise_z) {
            cp(d_nz, noise->noise_z, nz);
            sg.dn.noise_z = d_nz;
        }
    }
    return rc;
}

// host inputs in; returns once everything is enqueued (the mid-pipeline frame-count readback has completed, and with it
// the input copies: the caller's buffers are free)
static int run_async_locked(vits_handle *h, const int64_t *ids, const int64_t *lens, int B, int T, const float scales[3],
                            const int64_t *sid, const vits_noise *noise, vits_output *dev) {
    if (!scales) return fail(h, VITS_E_ARG, "null argument");
    Staged sg;
    int rc = stage_inputs(h, ids, lens, B, T, sid, noise, sg);
    if (rc == VITS_OK) rc = run_device_locked(h, sg.d_ids, sg.d_lens, B, T, scales, sg.d_sid, sg.has_noise ? &sg.dn : nullptr, dev);
    if (rc != VITS_OK) hipStreamSynchronize(h->stream);  // the staging slab is reused by the next call
    return rc;
}

int vits_run_async(vits_handle *h, const int64_t *ids, const int64_t *lens, int B, int T, const float scales[3],
                   const int64_t *sid, const vits_noise *noise) {
    if (int rc = check_dev(h)) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    vits_output dev{};
    return run_async_locked(h, ids, lens, B, T, scales, sid, noise, &dev);
}

int vits_run(vits_handle *h, const int64_t *ids, const int64_t *lens, int B, int T, const float scales[3],
             const int64_t *sid, const vits_noise *noise, vits_output *out) {
    if (int rc = check_dev(h)) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t st = h->stream;
    vits_output dev{};
    int rc = run_async_locked(h, ids, lens, B, T, scales, sid, noise, &dev);  // (out == NULL: run only)
    if (rc == VITS_OK && !out) {
        // run only: the caller fetches what it needs afterwards (vits_last_pcm16, vits_last_y_lengths, vits_tap)
        if (hipStreamSynchronize(st) != hipSuccess)
            rc = fail(h, VITS_E_DEVICE, "synchronisation failed: %s", hipGetErrorString(hipGetLastError()));
        else
            rc = range_check(h);
    } else if (rc == VITS_OK) {
        // one pinned block: [samples | frame counts]
        const size_t n = (size_t)B * h->S, ybase = (n * 4 + 63) & ~size_t(63);
        char *host = (char *)pinned_get(h, ybase + (size_t)B * 8);
        if (!host)
            rc = fail(h, VITS_E_NOMEM, "cannot allocate pinned output (%zu bytes)", n * 4);
        else if (hipMemcpyAsync(host, dev.data, n * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                 hipStreamSynchronize(st) != hipSuccess) {
            pinned_put(h, host);
            rc = fail(h, VITS_E_DEVICE, "device-to-host copy failed: %s", hipGetErrorString(hipGetLastError()));
        } else if ((rc = range_check(h)) != VITS_OK) {
            pinned_put(h, host);  // clamped audio is not handed out
        } else {
            int64_t *hy = (int64_t *)(host + ybase);
            for (int b = 0; b < B; b++) hy[b] = h->h_ylen[b];
            out->data = (float *)host;
            std::memcpy(out->dims, dev.dims, sizeof dev.dims);
            out->y_lengths = hy;
        }
        if (rc != VITS_OK) hipStreamSynchronize(st);
    }
    return rc;
}

int vits_run_chunked(vits_handle *h, const int64_t *ids, const int64_t *lens, int B, int T, const float scales[3],
                     const int64_t *sid, const vits_noise *noise, int chunk_frames, vits_chunk_fn fn, void *user) {
    if (int rc = check_dev(h)) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    if (!scales || chunk_frames < 1) return fail(h, VITS_E_ARG, "bad chunked-run arguments");
    Staged sg;
    int rc = stage_inputs(h, ids, lens, B, T, sid, noise, sg);
    const ChunkSink sink{chunk_frames, fn, user};
    if (rc == VITS_OK)
        rc = run_device_locked(h, sg.d_ids, sg.d_lens, B, T, scales, sg.d_sid, sg.has_noise ? &sg.dn : nullptr, nullptr, &sink);
    if (hipStreamSynchronize(h->stream) != hipSuccess && rc == VITS_OK)
        rc = fail(h, VITS_E_DEVICE, "synchronisation failed: %s", hipGetErrorString(hipGetLastError()));
    // (chunks are handed out as they finish; a range violation is therefore reported after the fact)
    if (rc == VITS_OK) rc = range_check(h);
    return rc;
}

void vits_free_output(vits_handle *h, vits_output *out) {
    if (!out) return;
    std::unique_lock<std::mutex> lk;
    if (h) lk = std::unique_lock<std::mutex>(h->mu);
    pinned_put(h, out->data);  // y_lengths lives in the same block
    out->data = nullptr;
    out->y_lengths = nullptr;
}

int vits_last_pcm16(vits_handle *h, int normalize, float volume, int16_t *out, size_t out_elems) {
    if (int rc = check_dev(h)) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    const int B = h->B, S = h->S;
    if (!h->d_out || !h->d_ylen || B <= 0 || S <= 0) return fail(h, VITS_E_ARG, "no completed run to post-process");
    const size_t n = (size_t)B * S;
    if (!out || out_elems < n) return fail(h, VITS_E_ARG, "pcm16 buffer too small: %zu < %zu", out_elems, n);
    // the staging slab is idle between runs (vits_run has consumed its inputs before it returns)
    const size_t pcm_bytes = (n * 2 + 255) & ~size_t(255);
    if (int rc = slab_reserve(h, h->io, pcm_bytes + (size_t)B * 4 + 256)) return rc;
    int16_t *d_pcm = (int16_t *)h->io.base;
    unsigned *d_peak = (unsigned *)(h->io.base + pcm_bytes);
    hipStream_t st = h->stream;
    hipError_t e = hipMemsetAsync(d_peak, 0, (size_t)B * 4, st);
    const int hop = h->model.hop;
    if (e == hipSuccess) {
        int gx = (S + 255) / 256;
        gx = gx > 256 ? 256 : gx;
        peak_abs_kernel<<<dim3(gx, B), 256, 0, st>>>(h->d_out, h->d_ylen, hop, S, d_peak);
        pcm16_kernel<<<dim3((S + 255) / 256, B), 256, 0, st>>>(h->d_out, h->d_ylen, hop, S, d_peak, normalize, volume, d_pcm);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_pcm, n * 2, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(h, VITS_E_DEVICE, "pcm16 post-processing failed: %s", hipGetErrorString(e));
    if (int rc = range_check(h)) return rc;  // (after vits_run_async this is the first synchronisation of that run)
    if (h->range_failed) return fail(h, VITS_E_RANGE, "the last run left the range of the fp16 operand planes (see vits_get_stats)");
    return VITS_OK;
}

// vocoder-only entry points: z (host, [B, inter, F], already masked) -> device, speaker bias; then either the whole
// waveform or chunks
static int vocoder_common(vits_handle *h, const float *z, int B, int F, const int64_t *sid, vits_output *out,
                          const ChunkSink *sink) {
    const Model &m = h->model;
    if (!z || (!out && !sink) || B <= 0 || F <= 0) return fail(h, VITS_E_ARG, "bad vocoder arguments");
    if (m.gin && !sid) return fail(h, VITS_E_ARG, "Missing speaker id");
    for (int b = 0; sid && m.gin && b < B; b++)
        if (sid[b] < 0 || sid[b] >= m.n_speakers)
            return fail(h, VITS_E_ARG, "sid[%d]=%lld is out of range [0,%d)", b, (long long)sid[b], m.n_speakers);
    std::memset(&h->stats, 0, sizeof h->stats);
    h->conv_events_used = 0;
    g_launch_name_on = h->timing == 1;
    h->range_failed = false;
    const size_t nCF = (size_t)B * m.C * F;
    const int Fgen = sink && sink->chunk_frames + 2 * m.gen_rf_frames < F ? sink->chunk_frames + 2 * m.gen_rf_frames : F;
    size_t need = al(nCF) + gen_ws_bytes(m, B, Fgen) + al((size_t)B * m.C0) + (1 << 16);
    if (int rc = slab_reserve(h, h->frm, need)) return rc;
    Slab &s = h->frm;
    s.used = 0;
    hipStream_t st = h->stream;
    float *dz = slab_take<float>(s, nCF);
    HIPCHECK(h, hipMemcpyAsync(dz, z, nCF * 4, hipMemcpyHostToDevice, st));
    Ctx c{h, m, st, h->arena_dev, B};
    float *dec_cond = nullptr;
    int64_t *d_sid = nullptr;
    if (m.gin) {
        d_sid = slab_take<int64_t>(s, B);
        HIPCHECK(h, hipMemcpyAsync(d_sid, sid, (size_t)B * 8, hipMemcpyHostToDevice, st));
        dec_cond = slab_take<float>(s, (size_t)B * m.C0);
        cond_matvec_kernel<<<dim3((m.C0 + 63) / 64, B), 64, 0, st>>>(c.P(m.emb_g), d_sid, m.n_speakers, c.P(m.dec_cond_w),
                                                                    c.P(m.dec_cond_b), dec_cond, m.C0, m.gin);
    }
    h->B = B;
    h->F = F;
    h->d_ylen = nullptr;  // (no frame counts: vits_last_pcm16 does not apply to a vocoder-only run)
    range_begin(h);
    int rc;
    if (sink) rc = render_chunks(h, c, dz, (int64_t)m.C * F, F, nullptr, B, F, dec_cond, s, *sink);
    else rc = run_generator(h, c, dz, (int64_t)m.C * F, F, nullptr, B, F, dec_cond, s);
    if (rc) return rc;
    if (c.err != hipSuccess) return fail(h, VITS_E_DEVICE, "kernel launch failed: %s", hipGetErrorString(c.err));
    range_end(h);
    if (sink) {
        HIPCHECK(h, hipStreamSynchronize(st));
        return range_check(h);
    }
    size_t n = (size_t)B * h->S;
    float *host = (float *)pinned_get(h, n * 4 + 64);
    if (!host) return fail(h, VITS_E_NOMEM, "pinned alloc failed");
    hipError_t ce = hipMemcpyAsync(host, h->d_out, n * 4, hipMemcpyDeviceToHost, st);
    if (ce == hipSuccess) ce = hipStreamSynchronize(st);
    if (ce != hipSuccess) {
        pinned_put(h, host);
        return fail(h, VITS_E_DEVICE, "device-to-host copy failed: %s", hipGetErrorString(ce));
    }
    if (int rr = range_check(h)) {
        pinned_put(h, host);
        return rr;
    }
    out->data = host;
    out->dims[0] = B;
    out->dims[1] = 1;
    out->dims[2] = 1;
    out->dims[3] = h->S;
    out->y_lengths = nullptr;
    return VITS_OK;
}

int vits_run_vocoder(vits_handle *h, const float *z, int B, int F, const int64_t *sid, vits_output *out) {
    if (int rc = check_dev(h)) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    if (!out) return fail(h, VITS_E_ARG, "bad vocoder arguments");
    return vocoder_common(h, z, B, F, sid, out, nullptr);
}

int vits_run_vocoder_chunked(vits_handle *h, const float *z, int B, int F, const int64_t *sid, int chunk_frames,
                             vits_chunk_fn fn, void *user) {
    if (int rc = check_dev(h)) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    if (chunk_frames < 1) return fail(h, VITS_E_ARG, "bad vocoder arguments");
    const ChunkSink sink{chunk_frames, fn, user};
    return vocoder_common(h, z, B, F, sid, nullptr, &sink);
}

int vits_tap(vits_handle *h, const char *name, float *buf, size_t buf_elems, int64_t dims[VITS_MAX_DIMS]) {
    if (int rc = check_dev(h)) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    if (!name || !dims) return fail(h, VITS_E_ARG, "null argument");
    const Model &m = h->model;
    const int B = h->B, T = h->T, F = h->F;
    std::string k = name;
    const float *src = nullptr;
    int64_t bstride = 0;
    int nd = 3, C = 0, L = 0, cstride = 0;
    if (k == "x") { src = h->d_x; C = m.H; L = T; bstride = (int64_t)C * L; }
    else if (k == "emb") { src = h->d_emb; C = m.H; L = T; bstride = (int64_t)C * L; }
    else if (k == "m_p") { src = h->d_mp; C = m.C; L = T; bstride = (int64_t)2 * C * L; }
    else if (k == "logs_p") { src = h->d_logs; C = m.C; L = T; bstride = (int64_t)2 * C * L; }
    else if (k == "logw") { src = h->d_logw; C = 1; L = T; bstride = L; }
    else if (k == "w_ceil") { src = h->d_wceil; C = 1; L = T; bstride = L; nd = 2; }
    else if (k == "z_p") { src = h->d_zp; C = m.C; L = F; cstride = h->Fpitch; bstride = (int64_t)C * cstride; }
    else if (k == "z") { src = h->d_z; C = m.C; L = F; cstride = h->Fpitch; bstride = (int64_t)C * cstride; }
    else return fail(h, VITS_E_ARG, "unknown tap %s", name);
    if (!src || B == 0) return fail(h, VITS_E_ARG, "no completed run to tap");
    if (nd == 2) { dims[0] = B; dims[1] = L; }
    else { dims[0] = B; dims[1] = C; dims[2] = L; }
    size_t n = (size_t)B * C * L;
    if (!buf) return nd;
    if (buf_elems < n) return fail(h, VITS_E_ARG, "tap buffer too small: %zu < %zu", buf_elems, n);
    float *tmp = nullptr;
    HIPCHECK(h, hipMalloc((void **)&tmp, n * 4 + 16));
    gather_view_kernel<<<dim3((L + 255) / 256, C, B), 256, 0, h->stream>>>(src, bstride, cstride ? cstride : L, tmp, C, L);
    hipError_t e = hipMemcpyAsync(buf, tmp, n * 4, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(tmp);
    if (e != hipSuccess) return fail(h, VITS_E_DEVICE, "tap copy failed: %s", hipGetErrorString(e));
    return nd;
}

int vits_get_stats(vits_handle *h, vits_stats *out) {
    if (!h || !out) return VITS_E_ARG;
    if (!h->host_only) {
        hipSetDevice(h->device);
        hipStreamSynchronize(h->stream);
        range_check(h);  // fills stats.f16_*; a violation stays on record for vits_sync (range_failed)
        if (h->timing) {
            float ms = 0.f, tot = 0.f, tot_sx = 0.f;
            for (size_t i = 0; i < h->conv_events_used; i++) {
                if (hipEventElapsedTime(&ms, h->conv_events[i].first, h->conv_events[i].second) != hipSuccess) continue;
                h->conv_recs[i].ms = ms;
                tot += ms;
                if (i < h->conv_event_sx.size() && h->conv_event_sx[i]) tot_sx += ms;
            }
            h->stats.conv_ms = tot;
            h->stats.sx_ms = tot_sx;
            auto el = [&](int a, int b) {
                float v = 0.f;
                if (hipEventElapsedTime(&v, h->ev[a], h->ev[b]) != hipSuccess) v = 0.f;
                return v;
            };
            h->stats.enc_ms = el(0, 1);
            h->stats.dp_ms = el(1, 2);
            h->stats.flow_ms = el(2, 3);
            h->stats.dec_ms = el(3, 4);
            h->stats.total_ms = el(0, 4);
        }
    }
    *out = h->stats;
    return VITS_OK;
}

int vits_launch_records(vits_handle *h, vits_launch_record *buf, int n) {
    if (!h) return VITS_E_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    const int have = (int)h->conv_events_used;
    for (int i = 0; i < have && i < n && buf; i++) buf[i] = h->conv_recs[i];
    return have;
}

void *vits_host_alloc(size_t bytes) {
    void *p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1) == hipSuccess ? p : nullptr;
}

void vits_host_free(void *p) {
    if (p) hipHostFree(p);
}

int vits_fetch_output(vits_handle *h, float *dst, size_t row_elems, size_t dst_elems) {
    if (int rc = check_dev(h)) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    const int B = h->B, S = h->S;
    if (!h->d_out || B <= 0 || S <= 0) return fail(h, VITS_E_ARG, "no completed run to fetch");
    if (!dst || row_elems < (size_t)S || dst_elems < (size_t)B * row_elems)
        return fail(h, VITS_E_ARG, "output buffer too small: rows of %zu (need %d), %zu elements (need %zu)", row_elems, S,
                    dst_elems, (size_t)B * row_elems);
    hipStream_t st = h->stream;
    hipError_t e = row_elems == (size_t)S
                       ? hipMemcpyAsync(dst, h->d_out, (size_t)B * S * 4, hipMemcpyDeviceToHost, st)
                       : hipMemcpy2DAsync(dst, row_elems * 4, h->d_out, (size_t)S * 4, (size_t)S * 4, B, hipMemcpyDeviceToHost, st);
    if (row_elems > (size_t)S)  // (host work while the copy runs: the tail columns belong to nobody else)
        for (int b = 0; b < B; b++) std::memset(dst + (size_t)b * row_elems + S, 0, (row_elems - S) * 4);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(h, VITS_E_DEVICE, "device-to-host copy failed: %s", hipGetErrorString(e));
    if (int rc = range_check(h)) return rc;
    if (h->range_failed) return fail(h, VITS_E_RANGE, "the last run left the range of the fp16 operand planes (see vits_get_stats)");
    return VITS_OK;
}

// ---------------------------------------------------------------- kernel-level test hooks

static int test_dev(int device_id) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n)
        return fail(nullptr, VITS_E_DEVICE, "no usable HIP device %d", device_id);
    if (hipSetDevice(device_id) != hipSuccess) return fail(nullptr, VITS_E_DEVICE, "hipSetDevice failed");
    return 0;
}

#define TCHECK(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) return fail(nullptr, VITS_E_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

static int run_test_conv(const ConvDesc &d, const std::vector<float> &arena, const float *x, int B, int T,
                         int flags, float slope, float *out, size_t out_elems, int64_t out_bstride) {
    float *dA = nullptr, *dx = nullptr, *dout = nullptr;
    size_t nx = (size_t)B * d.Cin * T;
    TCHECK(hipMalloc((void **)&dA, arena.size() * 4));
    TCHECK(hipMalloc((void **)&dx, nx * 4 + 16));
    TCHECK(hipMalloc((void **)&dout, out_elems * 4 + 16));
    TCHECK(hipMemcpy(dA, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(dx, x, nx * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemset(dout, 0, out_elems * 4));
    ConvArgs a{};
    a.x = dx;
    a.x_bstride = (int64_t)d.Cin * T;
    a.T = T;
    a.wp = dA + d.w_off;
    a.bias = d.b_off >= 0 ? dA + d.b_off : nullptr;
    a.out = dout;
    a.out_bstride = out_bstride;
    a.zeros = dA;  // pack_test_* reserve a zero page at offset 0
    a.Cin = d.Cin; a.Cout = d.Cout; a.K = d.K; a.dil = d.dil; a.padL = d.padL; a.CK = d.CK;
    a.nchunks = d.nchunks; a.ups = d.ups;
    a.flags = ((flags & 1) ? PRO_LRELU : 0) | ((flags & 2) ? EPI_RELU : 0);
    a.slope = slope;
    a.div = 1.f;
    TCHECK(launch_conv(a, d.cfg, B, nullptr));
    TCHECK(hipDeviceSynchronize());
    TCHECK(hipMemcpy(out, dout, out_elems * 4, hipMemcpyDeviceToHost));
    hipFree(dA);
    hipFree(dx);
    hipFree(dout);
    return VITS_OK;
}

int vits_test_conv1d(int device_id, const float *x, int B, int Cin, int T, const float *w, const float *bias, int Cout,
                     int K, int dil, int pad_l, int flags, float slope, float *out) {
    if (int rc = test_dev(device_id)) return rc;
    ConvDesc d;
    std::vector<float> arena;
    { std::string e = pack_test_conv(w, bias, Cin, Cout, K, dil, pad_l, (flags >> 8) & 3, &d, &arena); if (!e.empty()) return fail(nullptr, VITS_E_ARG, "%s", e.c_str()); }
    return run_test_conv(d, arena, x, B, T, flags, slope, out, (size_t)B * Cout * T, (int64_t)Cout * T);
}

// Micro-benchmark of one conv shape on random data (kernel tuning; tools/conv_bench.py): returns the
// average launch time in ms over `iters` back-to-back launches (HIP events on the null stream).
int vits_bench_conv1d(int device_id, int B, int Cin, int Cout, int T, int K, int dil, int hint, int iters,
                      int cfg_override, int ck_override, float *ms_out) {
    if (int rc = test_dev(device_id)) return rc;
    std::vector<float> w((size_t)Cout * Cin * K), x((size_t)B * Cin * T);
    uint32_t s = 12345u;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f;
    };
    for (auto &v : w) v = rnd() * 0.05f;
    for (auto &v : x) v = rnd();
    ConvDesc d;
    std::vector<float> arena;
    set_tiling_override(cfg_override, ck_override);
    const int dbg = hint >> 8;  // bit0: no DMA after warm-up chunks, bit1: no epilogue, bit2: no lrelu prologue
    hint &= 3;
    std::string e = pack_test_conv(w.data(), nullptr, Cin, Cout, K, dil, dil * (K - 1) / 2, hint, &d, &arena);
    set_tiling_override(-1, -1);
    if (!e.empty()) return fail(nullptr, VITS_E_ARG, "%s", e.c_str());
    float *dA = nullptr, *dx = nullptr, *dout = nullptr;
    TCHECK(hipMalloc((void **)&dA, arena.size() * 4));
    TCHECK(hipMalloc((void **)&dx, x.size() * 4));
    TCHECK(hipMalloc((void **)&dout, (size_t)B * Cout * T * 4));
    TCHECK(hipMemcpy(dA, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    ConvArgs a{};
    a.x = dx;
    a.x_bstride = (int64_t)Cin * T;
    a.T = T;
    a.wp = dA + d.w_off;
    a.out = dout;
    a.out_bstride = (int64_t)Cout * T;
    a.zeros = dA;
    a.Cin = d.Cin; a.Cout = d.Cout; a.K = d.K; a.dil = d.dil; a.padL = d.padL; a.CK = d.CK;
    a.nchunks = d.nchunks; a.ups = 1;
    a.flags = ((dbg & 4) ? 0 : PRO_LRELU) | ((dbg & 1) ? DBG_NO_DMA : 0) | ((dbg & 2) ? DBG_NO_EPI : 0);
    if ((dbg & 8) && Cin == Cout) {  // residual epilogue + pre-activated second output, as the generator runs it
        a.flags |= EPI_RES;
        a.res = dx;
        a.res_bstride = (int64_t)Cin * T;
        a.oslope2 = 0.1f;
    }
    a.slope = 0.1f;
    a.div = 1.f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) TCHECK(launch_conv(a, d.cfg, B, nullptr));
    TCHECK(hipDeviceSynchronize());
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; i++) TCHECK(launch_conv(a, d.cfg, B, nullptr));
    hipEventRecord(e1, nullptr);
    TCHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms_out) {
        ms_out[0] = ms / iters;
        ms_out[1] = (float)d.cfg;
        ms_out[2] = (float)d.CK;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(dA);
    hipFree(dx);
    hipFree(dout);
    return VITS_OK;
}

int vits_test_conv_transpose1d(int device_id, const float *x, int B, int Cin, int T, const float *w, const float *bias,
                               int Cout, int K, int stride, float *out) {
    if (int rc = test_dev(device_id)) return rc;
    ConvDesc d;
    std::vector<float> arena;
    { std::string e = pack_test_convT(w, bias, Cin, Cout, K, stride, &d, &arena); if (!e.empty()) return fail(nullptr, VITS_E_ARG, "%s", e.c_str()); }
    return run_test_conv(d, arena, x, B, T, 0, 0.f, out, (size_t)B * Cout * T * stride, (int64_t)Cout * T * stride);
}

// ---- the same hooks through the split-operand engine: planar host tensors are converted to the engine's
// plane / raw layouts on the device, the result is converted back.
static void fill_sx_args(SxArgs &a, const ConvDesc &d, const float *dA, int T) {
    const int Cr = d.Cout / d.ups;
    a.x_bstride = (int64_t)3 * (d.Cin / 8) * T;
    a.T = T;
    a.wp = reinterpret_cast<const u32x4 *>(dA + d.w_off);
    a.bias = d.b_off >= 0 ? dA + d.b_off : nullptr;
    a.raw_bstride = (int64_t)Cr * T * d.ups;
    a.pl_bstride = 3 * a.raw_bstride;
    a.zeros = dA;  // pack_test_* reserve a zero page at offset 0
    a.Cin = d.Cin; a.Cout = d.Cout; a.Cr = Cr; a.K = d.K; a.dil = d.dil; a.padL = d.padL;
    a.nchunks = d.nchunks; a.ups = d.ups;
    a.div = 1.f;
    a.s16 = d.s16 ? 1 : 0;
    a.zt_p = std::getenv("VITSMI_NO_ZERO_TAP_SKIP") ? -1 : d.zt_p;
}

static int run_test_conv_sx(const ConvDesc &d, const std::vector<float> &arena, const float *x, int B, int T, int flags,
                            float slope, float *out) {
    const int Cr = d.Cout / d.ups, To = T * d.ups;
    const size_t nx = (size_t)B * d.Cin * T, no = (size_t)B * Cr * To;
    float *dA = nullptr, *dx = nullptr, *dres = nullptr, *draw = nullptr, *dout = nullptr;
    uint16_t *dxp = nullptr, *dop = nullptr;
    TCHECK(hipMalloc((void **)&dA, arena.size() * 4));
    TCHECK(hipMalloc((void **)&dx, nx * 4 + 16));
    TCHECK(hipMalloc((void **)&dxp, nx * 6 + 16));
    TCHECK(hipMalloc((void **)&draw, no * 4 + 16));
    TCHECK(hipMalloc((void **)&dop, no * 6 + 16));
    TCHECK(hipMalloc((void **)&dout, no * 4 + 16));
    TCHECK(hipMemcpy(dA, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(dx, x, nx * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemset(draw, 0, no * 4));
    TCHECK(hipMemset(dop, 0, no * 6));
    if (d.h1 && (flags & 8)) {
        // single-plane mode with an input slope: the plane holds leaky_relu(x, slope) - what the conv consumes - and the
        // residual x is recovered from it (out = conv(lrelu(x)) + x): activate on the host, then store
        std::vector<float> xa(nx);
        for (size_t i = 0; i < nx; i++) xa[i] = x[i] >= 0.f ? x[i] : x[i] * slope;
        TCHECK(hipMemcpy(dx, xa.data(), nx * 4, hipMemcpyHostToDevice));
    }
    sx_split_planes_kernel<<<dim3((T + 255) / 256, d.Cin / 8, B), 256>>>(dx, (int64_t)d.Cin * T, T, nullptr, dxp, d.Cin, T,
                                                                         d.h1 ? 2 : (d.f16 ? 1 : 0));
    TCHECK(hipMalloc((void **)&dres, nx * 4 + 16));  // x in the raw layout: raw-input operand and residual
    sx_block_kernel<<<dim3((T + 255) / 256, d.Cin / 8, B), 256>>>(dx, (int64_t)d.Cin * T, T, nullptr, dres, d.Cin, T);
    SxArgs a{};
    fill_sx_args(a, d, dA, T);
    a.xp = reinterpret_cast<const u32x4 *>(dxp);
    a.xr = dres;
    a.islope = (flags & 8) ? slope : 1.f;  // (raw-input convs only)
    a.out_raw = (flags & 128) ? nullptr : draw;  // (bit 7: planes are the only output - the specialised plane epilogues)
    a.out_pl = dop;
    a.oslope = 1.f;
    a.oslope2 = (flags & 1) ? slope : 1.f;
    if (flags & 4) {  // residual: res = x (same shape only)
        if (d.Cin != d.Cout || d.ups != 1) return fail(nullptr, VITS_E_ARG, "residual test needs Cin == Cout");
        if (d.h1) {
            a.res_pl = dxp;  // the residual is the input plane itself, un-activated on the way in
            a.res_unslope = (flags & 8) ? 1.f / slope : 1.f;
        } else
            a.res = dres;
        a.flags |= EPI_RES;
    }
    const int nprod = d.h1 ? 1 : (d.f16 ? 2 : 6);
    a.wscale = d.wscale;
    if (flags & 256) {  // the short-launch kernel (conv_sx_small.hip.hpp) with the generator's epilogue
        a.s16 = d.s16 ? 1 : 0;
        if (!conv_sx_small_ok(a, d.rawin, nprod)) return fail(nullptr, VITS_E_ARG, "arguments not taken by the short-launch kernel");
        TCHECK(launch_conv_sx_small(a, B, d.cfg, nullptr));
    } else
        TCHECK(launch_conv_sx(a, d.cfg, B, nullptr, d.rawin, nprod));
    sx_unblock_kernel<<<dim3((To + 255) / 256, Cr / 8, B), 256>>>(draw, (flags & 1) ? dop : nullptr, dout, Cr, To,
                                                                  d.h1 ? 2 : (d.f16 ? 1 : 0));
    TCHECK(hipGetLastError());
    TCHECK(hipDeviceSynchronize());
    TCHECK(hipMemcpy(out, dout, no * 4, hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dx); hipFree(dxp); hipFree(draw); hipFree(dop); hipFree(dout);
    if (dres) hipFree(dres);
    return VITS_OK;
}

int vits_test_conv1d_sx(int device_id, const float *x, int B, int Cin, int T, const float *w, const float *bias, int Cout,
                        int K, int dil, int pad_l, int flags, float slope, float *out) {
    if (int rc = test_dev(device_id)) return rc;
    ConvDesc d;
    std::vector<float> arena;
    const int prec = (flags >> 4) & 3;  // 0: bf16x6, 3: f16x3 (two fp16 planes), 2: f16 (one fp16 plane, one product)
    if (prec == 1) return fail(nullptr, VITS_E_ARG, "precision code 1 (bf16x3) was retired");
    if ((flags & 128) && !(flags & 1)) return fail(nullptr, VITS_E_ARG, "planes-only output is read back from the planes (bit 0)");
    set_sx_f16(prec == 3);
    set_sx_h1(prec == 2);
    std::string e = pack_test_conv(w, bias, Cin, Cout, K, dil, pad_l, 3, &d, &arena);
    set_sx_f16(false);
    set_sx_h1(false);
    if (!e.empty()) return fail(nullptr, VITS_E_ARG, "%s", e.c_str());
    return run_test_conv_sx(d, arena, x, B, T, flags, slope, out);
}

int vits_test_conv1d_sx_planar(int device_id, const float *x, int B, int Cin, int T, const float *w, const float *bias,
                               int Cout, int K, int dil, int flags, const int64_t *lens, const float *old, int row_split,
                               int pl_rows, float *out, float *planes_out) {
    // the planar epilogue of the split-operand engine (f16x3, 16x16x32 loop), as the flow and the text encoder use it:
    // flags bit 0 ReLU, 1 mask (t < lens[b]), 2 residual (old: planar [B][row_split][T]), 3 accumulate (old: the
    // outputs' previous contents, [B][Cout][T]), 4 coupling update, 5 the rows behind row_split are stored (not
    // accumulated), 6 the planes are those of the rows behind row_split, 7 run the short-launch kernel.  out = [B][Cout][T] (rows behind row_split from
    // the second tensor); planes_out (nullable) = [B][pl_rows][T] read back from the operand planes.
    if (int rc = test_dev(device_id)) return rc;
    if (Cin % 32 || Cout % 32 || row_split % 32 || row_split > Cout || pl_rows % 32) return fail(nullptr, VITS_E_ARG, "bad planar test shape");
    ConvDesc d;
    std::vector<float> arena;
    set_sx_f16(true);
    std::string e = pack_test_conv(w, bias, Cin, Cout, K, dil, dil * (K - 1) / 2, 3, &d, &arena);
    set_sx_f16(false);
    if (!e.empty()) return fail(nullptr, VITS_E_ARG, "%s", e.c_str());
    if (!d.s16) return fail(nullptr, VITS_E_ARG, "shape not taken by the 16x16x32 loop");
    const int srows = Cout - row_split;
    const size_t nx = (size_t)B * Cin * T, n1 = (size_t)B * (row_split ? row_split : 1) * T, n2 = (size_t)B * (srows ? srows : 1) * T;
    float *dA = nullptr, *dx = nullptr, *d1 = nullptr, *d2 = nullptr, *dres = nullptr, *dpo = nullptr;
    uint16_t *dxp = nullptr, *dpl = nullptr;
    int *dlen = nullptr;
    TCHECK(hipMalloc((void **)&dA, arena.size() * 4));
    TCHECK(hipMalloc((void **)&dx, nx * 4 + 16));
    TCHECK(hipMalloc((void **)&dxp, nx * 6 + 64));
    TCHECK(hipMalloc((void **)&d1, n1 * 4 + 16));
    TCHECK(hipMalloc((void **)&d2, n2 * 4 + 16));
    TCHECK(hipMalloc((void **)&dres, n1 * 4 + 16));
    TCHECK(hipMalloc((void **)&dpl, (size_t)B * 3 * (pl_rows ? pl_rows : 32) * T * 2 + 64));
    TCHECK(hipMalloc((void **)&dpo, (size_t)B * (pl_rows ? pl_rows : 32) * T * 4 + 16));
    TCHECK(hipMalloc((void **)&dlen, (size_t)B * 4));
    std::vector<int> l32(B);
    for (int b = 0; b < B; b++) l32[b] = lens ? (int)lens[b] : T;
    TCHECK(hipMemcpy(dlen, l32.data(), (size_t)B * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(dA, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(dx, x, nx * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemset(d1, 0, n1 * 4));
    TCHECK(hipMemset(d2, 0, n2 * 4));
    TCHECK(hipMemset(dres, 0, n1 * 4));
    if (old) {
        // [B][Cout][T] -> the two planar tensors (accumulate / coupling) or the residual (rows of the first tensor)
        for (int b = 0; b < B; b++) {
            if (row_split)
                TCHECK(hipMemcpy(((flags & 4) ? dres : d1) + (size_t)b * row_split * T, old + (size_t)b * Cout * T, (size_t)row_split * T * 4,
                                 hipMemcpyHostToDevice));
            if (srows && !(flags & 4))
                TCHECK(hipMemcpy(d2 + (size_t)b * srows * T, old + ((size_t)b * Cout + row_split) * T, (size_t)srows * T * 4,
                                 hipMemcpyHostToDevice));
        }
    }
    sx_split_planes_kernel<<<dim3((T + 255) / 256, Cin / 8, B), 256>>>(dx, (int64_t)Cin * T, T, nullptr, dxp, Cin, T, 1);
    SxArgs a{};
    fill_sx_args(a, d, dA, T);
    a.wscale = d.wscale;
    a.s16 = 1;
    a.xp = reinterpret_cast<const u32x4 *>(dxp);
    a.out_raw = row_split ? d1 : nullptr;
    a.out_raw2 = srows ? d2 : nullptr;
    a.row_split = row_split;
    a.len = dlen;
    a.res = (flags & 4) ? dres : nullptr;
    a.out_pl = pl_rows ? dpl : nullptr;
    a.pl_rows = pl_rows;
    a.pl_of2 = (flags & 64) ? 1 : 0;
    a.pl_bstride = (int64_t)3 * pl_rows * T;
    a.flags = SX_WN_RMW | ((flags & 1) ? EPI_RELU : 0) | ((flags & 2) ? EPI_MASK : 0) | ((flags & 4) ? EPI_RES : 0) |
              ((flags & 8) ? EPI_ACC : 0) | ((flags & 16) ? SX_PLANAR_COUPLING : 0) | ((flags & 32) ? SX_PLANAR_STORE2 : 0);
    if (flags & 128) {  // the short-launch kernel (conv_sx_small.hip.hpp) instead of the engine's
        if (!conv_sx_small_ok(a, false, 2)) return fail(nullptr, VITS_E_ARG, "arguments not taken by the short-launch kernel");
        TCHECK(launch_conv_sx_small(a, B, d.cfg, nullptr));
    } else
        TCHECK(launch_conv_sx(a, d.cfg, B, nullptr, false, 2));
    if (pl_rows)
        sx_unblock_kernel<<<dim3((T + 255) / 256, pl_rows / 8, B), 256>>>(nullptr, dpl, dpo, pl_rows, T, 1);
    TCHECK(hipGetLastError());
    TCHECK(hipDeviceSynchronize());
    for (int b = 0; b < B; b++) {
        if (row_split)
            TCHECK(hipMemcpy(out + (size_t)b * Cout * T, d1 + (size_t)b * row_split * T, (size_t)row_split * T * 4, hipMemcpyDeviceToHost));
        if (srows)
            TCHECK(hipMemcpy(out + ((size_t)b * Cout + row_split) * T, d2 + (size_t)b * srows * T, (size_t)srows * T * 4, hipMemcpyDeviceToHost));
    }
    if (pl_rows && planes_out) TCHECK(hipMemcpy(planes_out, dpo, (size_t)B * pl_rows * T * 4, hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dx); hipFree(dxp); hipFree(d1); hipFree(d2); hipFree(dres); hipFree(dpl); hipFree(dpo); hipFree(dlen);
    return VITS_OK;
}

// A/B hook: the short-launch kernel's launch-size limit (0 = every launch on the engine); returns the previous value
long long vits_test_set_sx_small_max(long long wgs) { return sx_small_max().exchange(wgs); }

// The WN in-layer with its gate epilogue (SX_GATE; f16x3, 16x16x32 loop): acts = tanh(a + g_a) * sigmoid(b + g_b), a / b = the
// conv's rows.  w / bias arrive in the PACKED row order (32 tanh rows, their 32 sigmoid partners, the next 32 tanh rows, ..: what
// model.cpp's out_perm produces), bias_b [B][Cout] (the per-utterance conditioning) in the module's order (tanh half, sigmoid
// half).  flags bit 0: the short-launch kernel (conv_sx_small.hip.hpp) instead of the engine's; bit 1: acts through the fp16
// operand planes instead of the planar fp32 output.  out = [B][Cout / 2][T].
int vits_test_conv1d_sx_gate(int device_id, const float *x, int B, int Cin, int T, const float *w, const float *bias,
                             const float *bias_b, int Cout, int K, int dil, int flags, float *out) {
    if (int rc = test_dev(device_id)) return rc;
    if (Cin % 32 || Cout % 64) return fail(nullptr, VITS_E_ARG, "bad gate test shape");
    ConvDesc d;
    std::vector<float> arena;
    set_sx_f16(true);
    std::string e = pack_test_conv(w, bias, Cin, Cout, K, dil, dil * (K - 1) / 2, 3, &d, &arena);
    set_sx_f16(false);
    if (!e.empty()) return fail(nullptr, VITS_E_ARG, "%s", e.c_str());
    if (!d.s16) return fail(nullptr, VITS_E_ARG, "shape not taken by the 16x16x32 loop");
    const int H = Cout / 2;
    const size_t nx = (size_t)B * Cin * T, no = (size_t)B * H * T;
    float *dA = nullptr, *dx = nullptr, *dact = nullptr, *dbb = nullptr;
    uint16_t *dxp = nullptr, *dpl = nullptr;
    TCHECK(hipMalloc((void **)&dA, arena.size() * 4));
    TCHECK(hipMalloc((void **)&dx, nx * 4 + 16));
    TCHECK(hipMalloc((void **)&dxp, nx * 6 + 64));
    TCHECK(hipMalloc((void **)&dact, no * 4 + 16));
    TCHECK(hipMalloc((void **)&dpl, no * 6 + 64));
    TCHECK(hipMalloc((void **)&dbb, (size_t)B * Cout * 4));
    TCHECK(hipMemcpy(dA, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(dx, x, nx * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(dbb, bias_b, (size_t)B * Cout * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemset(dact, 0xff, no * 4));
    sx_split_planes_kernel<<<dim3((T + 255) / 256, Cin / 8, B), 256>>>(dx, (int64_t)Cin * T, T, nullptr, dxp, Cin, T, 1);
    SxArgs a{};
    fill_sx_args(a, d, dA, T);
    a.wscale = d.wscale;
    a.s16 = 1;
    a.xp = reinterpret_cast<const u32x4 *>(dxp);
    a.bias_b = dbb;
    a.bias_b_stride = Cout;
    a.flags = SX_GATE;
    a.raw_bstride = (int64_t)H * T;
    a.pl_bstride = (int64_t)3 * H * T;
    if (flags & 2) a.out_pl = dpl;
    else a.out_raw = dact;
    if (flags & 1) {
        if (!conv_sx_small_ok(a, false, 2)) return fail(nullptr, VITS_E_ARG, "arguments not taken by the short-launch kernel");
        TCHECK(launch_conv_sx_small(a, B, d.cfg, nullptr));
    } else
        TCHECK(launch_conv_sx(a, d.cfg == 0 ? 3 : d.cfg, B, nullptr, false, 2, d.cfg));
    if (flags & 2) sx_unblock_kernel<<<dim3((T + 255) / 256, H / 8, B), 256>>>(nullptr, dpl, dact, H, T, 1);
    TCHECK(hipGetLastError());
    TCHECK(hipDeviceSynchronize());
    TCHECK(hipMemcpy(out, dact, no * 4, hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dx); hipFree(dxp); hipFree(dact); hipFree(dpl); hipFree(dbb);
    return VITS_OK;
}

int vits_test_conv_transpose1d_sx(int device_id, const float *x, int B, int Cin, int T, const float *w, const float *bias,
                                  int Cout, int K, int stride, float *out) {
    if (int rc = test_dev(device_id)) return rc;
    ConvDesc d;
    std::vector<float> arena;
    const bool f16 = stride < 0;  // (test hook convention: negative stride = the fp16 two-plane mode)
    if (f16) stride = -stride;
    set_sx_f16(f16);
    std::string e = pack_test_convT(w, bias, Cin, Cout, K, stride, &d, &arena, true);
    set_sx_f16(false);
    if (!e.empty()) return fail(nullptr, VITS_E_ARG, "%s", e.c_str());
    return run_test_conv_sx(d, arena, x, B, T, 0, 0.f, out);
}

// out = c2(lrelu(c1(lrelu(x, slope)), slope)) + x through ONE fused launch (conv_sx_pair.hip.hpp; f16x3 arithmetic).
// x, out: [B, C, T] host; w1, w2: [C, C, K]; c1 dilated by dil1, c2 dilation 1; flags bit0: also time it (ms_out).
int vits_test_conv_pair_sx(int device_id, const float *x, int B, int C, int T, const float *w1, const float *b1,
                           const float *w2, const float *b2, int K, int dil1, int dil2, int chain, float slope, float *out,
                           float *ms_out) {
    if (int rc = test_dev(device_id)) return rc;
    // chain: bit 0 = CHAIN (two ResBlock2 steps), bits 1-2 = kernel: 0 conv_sx_pair_kernel (32x32x16 form, f16x3, fp32 raw
    // tensors), 1 conv_sx_pair16_kernel in f16x3, 2 the same in the single-plane arithmetic (both: operand planes of
    // leaky_relu(x) in); bit 3 (pair16 only): the result is read back from the output PLANES (leaky_relu(out, slope)) instead
    // of the fp32 output
    const int kern = (chain >> 1) & 3;
    const bool from_plane = (chain & 8) != 0;
    chain &= 1;
    if (kern) {
        const bool h1 = kern == 2;
        ConvDesc d1, d2;
        std::vector<float> arena;
        set_sx_f16(!h1);
        set_sx_force16(!h1);
        set_sx_h1(h1);
        std::string e = pack_test_conv(w1, b1, C, C, K, dil1, dil1 * (K - 1) / 2, 3, &d1, &arena);
        if (e.empty()) e = pack_test_conv(w2, b2, C, C, K, dil2, dil2 * (K - 1) / 2, 3, &d2, &arena);
        set_sx_f16(false);
        set_sx_force16(false);
        set_sx_h1(false);
        if (!e.empty()) return fail(nullptr, VITS_E_ARG, "%s", e.c_str());
        if (!d1.s16 || !d2.s16 || d1.rawin || !sx_pair16_plan(C, h1 ? 1 : 2, d1.K, d1.dil, d2.K, d2.dil, nullptr))
            return fail(nullptr, VITS_E_ARG, "this conv pair cannot run fused on the 16x16x32 loop (C %d, kernel %d, dilation %d)", C, K, dil1);
        const size_t n = (size_t)B * C * T;
        float *dA = nullptr, *dx = nullptr, *dxr = nullptr, *draw = nullptr, *dout = nullptr;
        uint16_t *dxp = nullptr, *dop = nullptr;
        TCHECK(hipMalloc((void **)&dA, arena.size() * 4));
        TCHECK(hipMalloc((void **)&dx, n * 4 + 16));
        TCHECK(hipMalloc((void **)&dxr, n * 4 + 16));
        TCHECK(hipMalloc((void **)&dxp, n * 6 + 16));
        TCHECK(hipMalloc((void **)&dop, n * 6 + 16));
        TCHECK(hipMalloc((void **)&draw, n * 4 + 16));
        TCHECK(hipMalloc((void **)&dout, n * 4 + 16));
        TCHECK(hipMemcpy(dA, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
        TCHECK(hipMemset(draw, 0, n * 4));
        TCHECK(hipMemset(dop, 0, n * 6));
        {  // the input planes hold leaky_relu(x, slope)
            std::vector<float> xa(n);
            for (size_t i = 0; i < n; i++) xa[i] = x[i] >= 0.f ? x[i] : x[i] * slope;
            TCHECK(hipMemcpy(dx, xa.data(), n * 4, hipMemcpyHostToDevice));
            sx_split_planes_kernel<<<dim3((T + 255) / 256, C / 8, B), 256>>>(dx, (int64_t)C * T, T, nullptr, dxp, C, T, h1 ? 2 : 1);
        }
        SxPair16Args a{};
        a.xpl = dxp;
        a.x_bstride = (int64_t)3 * C * T;
        a.islope = a.mslope = a.oslope = slope;
        a.T = T;
        a.wp1 = reinterpret_cast<const u32x4 *>(dA + d1.w_off);
        a.wp2 = reinterpret_cast<const u32x4 *>(dA + d2.w_off);
        a.bias1 = d1.b_off >= 0 ? dA + d1.b_off : nullptr;
        a.bias2 = d2.b_off >= 0 ? dA + d2.b_off : nullptr;
        a.wscale1 = d1.wscale;
        a.wscale2 = d2.wscale;
        a.out_raw = draw;
        a.raw_bstride = (int64_t)C * T;
        a.out_pl = dop;
        a.pl_bstride = (int64_t)3 * C * T;
        a.zeros = dA;
        a.K1 = d1.K; a.dil1 = d1.dil; a.pad1 = d1.padL;
        a.K2 = d2.K; a.dil2 = d2.dil; a.pad2 = d2.padL;
        a.flags = from_plane ? P16_HAS_PL : P16_HAS_RAW;
        a.div = 1.f;
        TCHECK(launch_conv_sx_pair16(a, C, h1 ? 1 : 2, B, nullptr, chain != 0));
        TCHECK(hipDeviceSynchronize());
#if P16_PROF
        {
            unsigned long long *dprof = nullptr;
            const size_t nwg = 1 << 17;  // (>= the launch's workgroups)
            TCHECK(hipMalloc((void **)&dprof, nwg * 64));
            TCHECK(hipMemset(dprof, 0, nwg * 64));
            a.prof = dprof;
            TCHECK(launch_conv_sx_pair16(a, C, h1 ? 1 : 2, B, nullptr, chain != 0));
            TCHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> hp(nwg * 8);
            TCHECK(hipMemcpy(hp.data(), dprof, nwg * 64, hipMemcpyDeviceToHost));
            // (a row per workgroup: phase sums over its tiles, [7] = tiles - one in the one-shot form, many in the persistent one)
            double sum[7] = {0, 0, 0, 0, 0, 0, 0};
            unsigned long long n = 0, wgs = 0;
            for (size_t w = 0; w < nwg; w++)
                if (hp[w * 8 + 7]) {
                    wgs++;
                    n += hp[w * 8 + 7];
                    for (int i = 0; i < 7; i++) sum[i] += (double)hp[w * 8 + i];
                }
            fprintf(stderr, "p16prof C %d npl %d K %d wgs %llu tiles %llu | x wait %.0f  barrier %.0f  phase1 %.0f  hand-over %.0f  phase2 %.0f  epilogue %.0f  drain %.0f (cycles per TILE)\n",
                    C, h1 ? 1 : 2, K, wgs, n, sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n, sum[5] / n, sum[6] / n);
            a.prof = nullptr;
            hipFree(dprof);
        }
#endif
        if (ms_out) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, nullptr);
            for (int i = 0; i < 10; i++) TCHECK(launch_conv_sx_pair16(a, C, h1 ? 1 : 2, B, nullptr, chain != 0));
            hipEventRecord(e1, nullptr);
            TCHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            *ms_out = ms / 10;
            hipEventDestroy(e0);
            hipEventDestroy(e1);
        }
        sx_unblock_kernel<<<dim3((T + 255) / 256, C / 8, B), 256>>>(draw, from_plane ? dop : nullptr, dout, C, T, h1 ? 2 : 1);
        TCHECK(hipGetLastError());
        TCHECK(hipDeviceSynchronize());
        TCHECK(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
        hipFree(dA); hipFree(dx); hipFree(dxr); hipFree(dxp); hipFree(dop); hipFree(draw); hipFree(dout);
        return VITS_OK;
    }
    ConvDesc d1, d2;
    std::vector<float> arena;
    set_sx_f16(true);
    std::string e = pack_test_conv(w1, b1, C, C, K, dil1, dil1 * (K - 1) / 2, 3, &d1, &arena);
    if (e.empty()) e = pack_test_conv(w2, b2, C, C, K, dil2, dil2 * (K - 1) / 2, 3, &d2, &arena);
    set_sx_f16(false);
    if (!e.empty()) return fail(nullptr, VITS_E_ARG, "%s", e.c_str());
    if (!d1.rawin || !d2.rawin || d1.cfg != d2.cfg || !sx_pair_supported(C, d1.cfg, d1.K, d1.dil, d2.K, d2.dil))
        return fail(nullptr, VITS_E_ARG, "this conv pair cannot run fused (C %d, kernel %d, dilation %d)", C, K, dil1);
    const size_t n = (size_t)B * C * T;
    float *dA = nullptr, *dx = nullptr, *dxr = nullptr, *draw = nullptr, *dout = nullptr;
    TCHECK(hipMalloc((void **)&dA, arena.size() * 4));
    TCHECK(hipMalloc((void **)&dx, n * 4 + 16));
    TCHECK(hipMalloc((void **)&dxr, n * 4 + 16));
    TCHECK(hipMalloc((void **)&draw, n * 4 + 16));
    TCHECK(hipMalloc((void **)&dout, n * 4 + 16));
    TCHECK(hipMemcpy(dA, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemset(draw, 0, n * 4));
    sx_block_kernel<<<dim3((T + 255) / 256, C / 8, B), 256>>>(dx, (int64_t)C * T, T, nullptr, dxr, C, T);
    SxPairArgs a{};
    a.xr = dxr;
    a.islope = a.mslope = slope;
    a.T = T;
    a.wp1 = reinterpret_cast<const u32x4 *>(dA + d1.w_off);
    a.wp2 = reinterpret_cast<const u32x4 *>(dA + d2.w_off);
    a.bias1 = d1.b_off >= 0 ? dA + d1.b_off : nullptr;
    a.bias2 = d2.b_off >= 0 ? dA + d2.b_off : nullptr;
    a.wscale1 = d1.wscale;
    a.wscale2 = d2.wscale;
    a.out_raw = draw;
    a.zeros = dA;
    a.C = C;
    a.K1 = d1.K; a.dil1 = d1.dil; a.pad1 = d1.padL;
    a.K2 = d2.K; a.dil2 = d2.dil; a.pad2 = d2.padL;
    a.div = 1.f;
    TCHECK(launch_conv_sx_pair(a, d1.cfg, B, nullptr, chain != 0));
    TCHECK(hipDeviceSynchronize());
    if (ms_out) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < 10; i++) TCHECK(launch_conv_sx_pair(a, d1.cfg, B, nullptr, chain != 0));
        hipEventRecord(e1, nullptr);
        TCHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        *ms_out = ms / 10;
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
    sx_unblock_kernel<<<dim3((T + 255) / 256, C / 8, B), 256>>>(draw, nullptr, dout, C, T, 1);
    TCHECK(hipGetLastError());
    TCHECK(hipDeviceSynchronize());
    TCHECK(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dx); hipFree(dxr); hipFree(draw); hipFree(dout);
    return VITS_OK;
}

int vits_bench_conv1d_sx(int device_id, int B, int Cin, int Cout, int T, int K, int dil, int dbg, int iters,
                         float *ms_out) {
    if (int rc = test_dev(device_id)) return rc;
    std::vector<float> w((size_t)Cout * Cin * K), x((size_t)B * Cin * T);
    uint32_t s = 12345u;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f;
    };
    for (auto &v : w) v = rnd() * 0.05f;
    for (auto &v : x) v = rnd();
    ConvDesc d;
    std::vector<float> arena;
    set_sx_f16((dbg & 128) != 0);
    set_sx_h1((dbg & 64) != 0);  // one fp16 plane, one product (VITSMI_GEN_PRECISION=f16)
    set_sx_shape32((dbg & (1 | 2 | 16 | 256)) != 0);  // ablation / cycle-breakdown builds exist for the 32x32x16 loop only
    std::string e = pack_test_conv(w.data(), nullptr, Cin, Cout, K, dil, dil * (K - 1) / 2, 3, &d, &arena);
    set_sx_f16(false);
    set_sx_h1(false);
    set_sx_shape32(false);
    if (!e.empty()) return fail(nullptr, VITS_E_ARG, "%s", e.c_str());
    float *dA = nullptr, *dx = nullptr, *draw = nullptr, *dres = nullptr;
    uint16_t *dxp = nullptr, *dop = nullptr;
    const size_t nx = x.size(), no = (size_t)B * Cout * T;
    TCHECK(hipMalloc((void **)&dA, arena.size() * 4));
    TCHECK(hipMalloc((void **)&dx, nx * 4));
    TCHECK(hipMalloc((void **)&dxp, nx * 6));
    TCHECK(hipMalloc((void **)&draw, no * 4));
    TCHECK(hipMalloc((void **)&dres, no * 4));
    TCHECK(hipMalloc((void **)&dop, no * 6));
    TCHECK(hipMemcpy(dA, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(dx, x.data(), nx * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemset(dres, 0, no * 4));
    sx_split_planes_kernel<<<dim3((T + 255) / 256, Cin / 8, B), 256>>>(dx, (int64_t)Cin * T, T, nullptr, dxp, Cin, T,
                                                                       d.h1 ? 2 : (d.f16 ? 1 : 0));
    SxArgs a{};
    fill_sx_args(a, d, dA, T);
    a.wscale = d.wscale;
    a.xp = reinterpret_cast<const u32x4 *>(dxp);
    a.out_pl = dop;  // as the generator's inner convs: planes out
    a.oslope2 = 0.1f;
    a.flags = ((dbg & 1) ? DBG_NO_DMA : 0) | ((dbg & 2) ? DBG_NO_EPI : 0);
    if (dbg & 8) {  // residual epilogue with raw + planes outputs (ResBlock tail)
        a.flags |= EPI_RES;
        if (d.h1) {  // (single-plane mode: plane in, residual from a plane, plane out - the ResBlock tail of that mode)
            a.res_pl = reinterpret_cast<const uint16_t *>(dres);
            a.res_unslope = 10.f;
        } else {
            a.res = dres;
            a.out_raw = draw;
        }
    }
    if (d.rawin) {  // <= 64 input channels: as the generator runs such layers: fp32 raw in (lrelu on load), raw out
        a.xr = dx;
        a.islope = 0.1f;
        a.out_pl = nullptr;
        a.out_raw = draw;
    }
    unsigned long long *dprof = nullptr;
    if (dbg & 16) {  // per-step cycle breakdown (128x128 tile only), returned in ms_out[3..8]
        TCHECK(hipMalloc((void **)&dprof, 64));
        TCHECK(hipMemset(dprof, 0, 64));
        a.prof = dprof;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) TCHECK(launch_conv_sx(a, d.cfg, B, nullptr, d.rawin, d.h1 ? 1 : (d.f16 ? 2 : 6)));
    TCHECK(hipDeviceSynchronize());
    if (dprof) TCHECK(hipMemset(dprof, 0, 64));
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; i++) TCHECK(launch_conv_sx(a, d.cfg, B, nullptr, d.rawin, d.h1 ? 1 : (d.f16 ? 2 : 6)));
    hipEventRecord(e1, nullptr);
    TCHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms_out) {
        ms_out[0] = ms / iters;
        ms_out[1] = (float)d.cfg;
        ms_out[2] = 0.f;
        if (dprof) {
            unsigned long long hp[8];
            TCHECK(hipMemcpy(hp, dprof, sizeof hp, hipMemcpyDeviceToHost));
            for (int i = 0; i < 5; i++) ms_out[3 + i] = hp[5] ? (float)((double)hp[i] / (double)hp[5]) : 0.f;  // per step
            ms_out[2] = hp[7] ? (float)((double)hp[6] / (double)hp[7] * 0.1) : 0.f;  // shader clock, GHz
        }
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (dprof) hipFree(dprof);
    hipFree(dA); hipFree(dx); hipFree(dxp); hipFree(draw); hipFree(dres); hipFree(dop);
    return VITS_OK;
}

int vits_test_attention(int device_id, const float *qkv, int B, int C, int T, int n_heads, const float *rel_k,
                        const float *rel_v, int window, const int64_t *lens, float *out) {
    if (int rc = test_dev(device_id)) return rc;
    if (window > 4 || C % n_heads) return fail(nullptr, VITS_E_ARG, "bad attention test arguments");
    int dk = C / n_heads;
    float *dq = nullptr, *dout = nullptr, *drk = nullptr, *drv = nullptr;
    int *dlen = nullptr;
    size_t nq = (size_t)B * 3 * C * T, no = (size_t)B * C * T, nr = (size_t)(2 * window + 1) * dk;
    std::vector<int> l32(B);
    for (int b = 0; b < B; b++) l32[b] = (int)lens[b];
    TCHECK(hipMalloc((void **)&dq, nq * 4));
    TCHECK(hipMalloc((void **)&dout, no * 4));
    TCHECK(hipMalloc((void **)&drk, nr * 4));
    TCHECK(hipMalloc((void **)&drv, nr * 4));
    TCHECK(hipMalloc((void **)&dlen, B * 4));
    TCHECK(hipMemcpy(dq, qkv, nq * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(drk, rel_k, nr * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(drv, rel_v, nr * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(dlen, l32.data(), B * 4, hipMemcpyHostToDevice));
    launch_attention(nullptr, B, T, n_heads, dk, window, dq, dout, drk, drv, dlen, C);
    TCHECK(hipGetLastError());
    TCHECK(hipDeviceSynchronize());
    TCHECK(hipMemcpy(out, dout, no * 4, hipMemcpyDeviceToHost));
    hipFree(dq); hipFree(dout); hipFree(drk); hipFree(drv); hipFree(dlen);
    return VITS_OK;
}

// ... the 16x16x32 f16x3 kernel (kernel = 1) or the fp32-MFMA one (0) with timing: the q | k | v tensor is split into
// operand planes on the device first (what the q|k|v conv's epilogue does in the pipeline); out_planes (optional) receives the
// output's operand planes [B][3][C/8][T][8]; reps > 0: ms_out[0] = mean launch duration over reps launches (HIP events).
int vits_test_attention16(int device_id, const float *qkv, int B, int C, int T, int n_heads, const float *rel_k,
                          const float *rel_v, int window, const int64_t *lens, float *out, uint16_t *out_planes, int kernel,
                          int reps, float *ms_out) {
    if (int rc = test_dev(device_id)) return rc;
    if (window > 4 || C % n_heads || C % 8) return fail(nullptr, VITS_E_ARG, "bad attention test arguments");
    int dk = C / n_heads;
    if (kernel == 1 && !(dk % 32 == 0 && dk <= 96)) return fail(nullptr, VITS_E_ARG, "attention16 needs a head width of 32, 64 or 96");
    float *dq = nullptr, *dout = nullptr, *drk = nullptr, *drv = nullptr;
    uint16_t *dqp = nullptr, *dop = nullptr;
    unsigned *dpk = nullptr;
    int *dlen = nullptr;
    size_t nq = (size_t)B * 3 * C * T, no = (size_t)B * C * T, nr = (size_t)(2 * window + 1) * dk;
    std::vector<int> l32(B);
    for (int b = 0; b < B; b++) l32[b] = (int)lens[b];
    TCHECK(hipMalloc((void **)&dq, nq * 4));
    TCHECK(hipMalloc((void **)&dqp, nq * 3 * 2));
    TCHECK(hipMalloc((void **)&dop, no * 3 * 2));
    TCHECK(hipMalloc((void **)&dpk, kSxPeakSlots * kSxPeakStride * 4));
    TCHECK(hipMalloc((void **)&dout, no * 4));
    TCHECK(hipMalloc((void **)&drk, nr * 4));
    TCHECK(hipMalloc((void **)&drv, nr * 4));
    TCHECK(hipMalloc((void **)&dlen, B * 4));
    TCHECK(hipMemset(dpk, 0, kSxPeakSlots * kSxPeakStride * 4));
    TCHECK(hipMemset(dop, 0xff, no * 3 * 2));
    TCHECK(hipMemset(dout, 0xff, no * 4));
    TCHECK(hipMemcpy(dq, qkv, nq * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(drk, rel_k, nr * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(drv, rel_v, nr * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(dlen, l32.data(), B * 4, hipMemcpyHostToDevice));
    sx_split_planes_kernel<<<dim3((T + 255) / 256, 3 * C / 8, B), 256>>>(dq, (int64_t)3 * C * T, T, nullptr, dqp, 3 * C, T, 1, dpk);
    unsigned long long *dprof = nullptr;
    const int prof_wgs = ((n_heads * B + 7) / 8) * 8 * ((T + 63) / 64);
#if ATT16_PROF
    TCHECK(hipMalloc((void **)&dprof, (size_t)prof_wgs * 4 * 8));
    TCHECK(hipMemset(dprof, 0, (size_t)prof_wgs * 4 * 8));
#endif
    auto go = [&] {
        if (kernel == 1) (void)launch_attention16(nullptr, B, T, n_heads, dk, window, dqp, dout, dop, drk, drv, dlen, C, dpk, dprof);
        else launch_attention(nullptr, B, T, n_heads, dk, window, dq, dout, drk, drv, dlen, C, dk % 8 == 0 ? dop : nullptr, dpk);
    };
    go();
    TCHECK(hipGetLastError());
    TCHECK(hipDeviceSynchronize());
    if (reps > 0 && ms_out) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, nullptr);
        for (int r = 0; r < reps; r++) go();
        hipEventRecord(e1, nullptr);
        TCHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        ms_out[0] = ms / reps;
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
    if (dprof && kernel == 1) {  // ATT16_PROF builds: where a workgroup's time goes (shader-clock cycles, means over the launch)
        std::vector<unsigned long long> hp((size_t)prof_wgs * 4);
        TCHECK(hipMemcpy(hp.data(), dprof, hp.size() * 8, hipMemcpyDeviceToHost));
        double d[3] = {0, 0, 0};
        unsigned long long t0 = ~0ull, t1 = 0;
        int n = 0;
        for (int w = 0; w < prof_wgs; w++) {
            if (!hp[w * 4 + 3]) continue;
            for (int k = 0; k < 3; k++) d[k] += (double)(hp[w * 4 + k + 1] - hp[w * 4 + k]);
            t0 = hp[w * 4] < t0 ? hp[w * 4] : t0;
            t1 = hp[w * 4 + 3] > t1 ? hp[w * 4 + 3] : t1;
            n++;
        }
        if (n) std::fprintf(stderr, "att16 stamps B=%d T=%d: %d workgroups; cycles prologue %.0f loop %.0f epilogue %.0f; first start -> last end %llu\n",
                            B, T, n, d[0] / n, d[1] / n, d[2] / n, t1 - t0);
        hipFree(dprof);
    }
    TCHECK(hipMemcpy(out, dout, no * 4, hipMemcpyDeviceToHost));
    if (out_planes) TCHECK(hipMemcpy(out_planes, dop, no * 3 * 2, hipMemcpyDeviceToHost));
    hipFree(dq); hipFree(dqp); hipFree(dop); hipFree(dpk); hipFree(dout); hipFree(drk); hipFree(drv); hipFree(dlen);
    return VITS_OK;
}

}  // extern "C"
