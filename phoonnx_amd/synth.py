"""Synthetic VITS voices for benchmarking and full-size parity tests.

The reference tree ships no trained voice and the GPU box has no network, so `bench.py`
and the full-size tests need `.onnx` files with the exact structure `export_onnx.py`
produces (node names carrying module paths, weight-normed flow convs folded into
anonymous `onnx::Conv_N` initializers, `Neg(logs)` folded into `onnx::Exp_N`,
metadata_props) but seeded random weights.  Only the nodes that carry parameters are
written; the ~5 000 shape-plumbing nodes of a real export hold no information the engine
uses.  Real exports (tests/golden/*.onnx were produced by the reference's own exporter)
load through the same reader.

Weights follow the PyTorch initialisers of the reference modules, with the zero/one
initialised tensors perturbed (as oracle/gen_golden.py does) so that flows and splines are
not identities, and the generator weights scaled so the waveform peaks around 0.5.
"""
import json
import os
import struct

import numpy as np

PRESETS = {
    # phoonnx default ("medium" quality): lightning.py:26-35
    "medium": dict(resblock="2", resblock_kernel_sizes=(3, 5, 7), resblock_dilation_sizes=((1, 2), (2, 6), (3, 12)),
                   upsample_rates=(8, 8, 4), upsample_initial_channel=256, upsample_kernel_sizes=(16, 16, 8),
                   dec_gain=5.6),
    # original LJSpeech VITS ("high"): phoonnx_train/vits/config.py:43-56
    "high": dict(resblock="1", resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3, 5),) * 3,
                 upsample_rates=(8, 8, 2, 2), upsample_initial_channel=512, upsample_kernel_sizes=(16, 16, 4, 4),
                 dec_gain=3.9),
    "small": dict(resblock="2", resblock_kernel_sizes=(3, 5, 7), resblock_dilation_sizes=((1, 2), (2, 6), (3, 12)),
                  upsample_rates=(8, 4, 2), upsample_initial_channel=64, upsample_kernel_sizes=(16, 8, 4),
                  inter_channels=64, hidden_channels=64, filter_channels=128, n_layers=2, dec_gain=11.0),
}
BASE = dict(n_vocab=256, inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6,
            kernel_size=3, n_speakers=1, gin_channels=0, use_sdp=True, window=4, flow_layers=4, flow_wn_layers=4,
            flow_kernel=5, dp_kernel=3, n_bins=10)


# ---------------------------------------------------------------- protobuf encoding helpers
def _vint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):
    return _vint((field << 3) | 2) + _vint(len(payload)) + payload


def _vi(field, v):
    return _vint(field << 3) + _vint(v)


def _tensor(name, arr):
    arr = np.ascontiguousarray(arr, np.float32)
    b = b"".join(_vi(1, d) for d in arr.shape) + _vi(2, 1) + _ld(8, name.encode()) + _ld(9, arr.tobytes())
    return b


def _attr_ints(name, vals):
    return _ld(5, _ld(1, name.encode()) + b"".join(_vi(8, v) for v in vals) + _vi(20, 7))


def _attr_int(name, v):
    return _ld(5, _ld(1, name.encode()) + _vi(3, v) + _vi(20, 2))


def _node(op, name, inputs, outputs, attrs=b""):
    b = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    return b + _ld(3, name.encode()) + _ld(4, op.encode()) + attrs


class _Graph:
    def __init__(self, rng):
        self.rng = rng
        self.nodes = []
        self.inits = []
        self.anon = 6000
        self.act = 0

    def _a(self):
        self.act += 1
        return f"/act_{self.act}"

    def init(self, name, arr):
        self.inits.append(_ld(5, _tensor(name, arr)))
        return name

    def conv(self, mod, w, b=None, dil=1, groups=1, fold=False, transpose=False, stride=1, pad=0):
        path = "/" + mod.replace(".", "/")
        wname = mod + ".weight"
        if fold:  # weight-normed module: exporter constant-folds g*v/|v| into an anonymous tensor
            self.anon += 3
            wname = f"onnx::Conv_{self.anon}"
        ins = [self._a(), self.init(wname, w)]
        if b is not None:
            ins.append(self.init(mod + ".bias", b))
        k = w.shape[2]
        attrs = _attr_ints("dilations", [dil]) + _attr_int("group", groups) + _attr_ints("kernel_shape", [k])
        if transpose:
            attrs += _attr_ints("pads", [pad, pad]) + _attr_ints("strides", [stride])
            self.nodes.append(_ld(1, _node("ConvTranspose", path + "/ConvTranspose", ins, [self._a()], attrs)))
        else:
            p = (k * dil - dil) // 2
            attrs += _attr_ints("pads", [p, p]) + _attr_ints("strides", [1])
            self.nodes.append(_ld(1, _node("Conv", path + "/Conv", ins, [self._a()], attrs)))

    def layernorm(self, mod, gamma, beta):
        path = "/" + mod.replace(".", "/")
        self.nodes.append(_ld(1, _node("Mul", path + "/Mul", [self._a(), self.init(mod + ".gamma", gamma)], [self._a()])))
        self.nodes.append(_ld(1, _node("Add", path + "/Add_1", [self._a(), self.init(mod + ".beta", beta)], [self._a()])))

    def gather(self, mod, table):
        path = "/" + mod.replace(".", "/")
        self.nodes.append(_ld(1, _node("Gather", path + "/Gather", [self.init(mod + ".weight", table), self._a()],
                                       [self._a()])))


def _uconv(rng, cout, cin, k, groups=1):
    """nn.Conv1d default init: kaiming_uniform(a=sqrt(5)) -> U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for w and b."""
    fan_in = (cin // groups) * k
    bound = 1.0 / np.sqrt(fan_in)
    w = rng.uniform(-bound, bound, size=(cout, cin // groups, k)).astype(np.float32)
    b = rng.uniform(-bound, bound, size=(cout,)).astype(np.float32)
    return w, b


def _xavier(rng, cout, cin, k=1):
    bound = np.sqrt(6.0 / (cin * k + cout * k))
    return rng.uniform(-bound, bound, size=(cout, cin, k)).astype(np.float32)


def hparams(preset="medium", **over):
    hp = dict(BASE)
    hp.update(PRESETS[preset])
    hp.update(over)
    if hp["n_speakers"] > 1 and not hp["gin_channels"]:
        hp["gin_channels"] = 512  # lightning.py:82-84
    return hp


def write_voice(path, preset="medium", seed=1234, extra_inputs=(), **over):
    """Write `<path>` (.onnx) and `<path>.json` (voice config).  Returns the hyper-parameter dict.
    extra_inputs: further graph input names to declare (e.g. "langid", as third-party exports do, voice.py:369)."""
    hp = hparams(preset, **over)
    rng = np.random.default_rng(seed)
    g = _Graph(rng)
    H, C, FF, V = hp["hidden_channels"], hp["inter_channels"], hp["filter_channels"], hp["n_vocab"]
    heads, win, gin = hp["n_heads"], hp["window"], hp["gin_channels"]
    dk = H // heads
    N = lambda *s, std=1.0: (rng.standard_normal(s) * std).astype(np.float32)
    ln = lambda c: ((1 + N(c, std=0.1)), N(c, std=0.1))

    # ---- text encoder (models.py:168-209)
    g.gather("enc_p.emb", N(V, H, std=H ** -0.5))
    for l in range(hp["n_layers"]):
        a = f"enc_p.encoder.attn_layers.{l}"
        apath = "/" + a.replace(".", "/")
        g.nodes.append(_ld(1, _node("Pad", apath + "/Pad", [g.init(a + ".emb_rel_k", N(1, 2 * win + 1, dk, std=dk ** -0.5)),
                                                           g._a()], [g._a()])))
        for nm in ("conv_q", "conv_k", "conv_v"):
            _, b = _uconv(rng, H, H, 1)
            g.conv(f"{a}.{nm}", _xavier(rng, H, H), b)
        g.nodes.append(_ld(1, _node("Pad", apath + "/Pad_5", [g.init(a + ".emb_rel_v", N(1, 2 * win + 1, dk, std=dk ** -0.5)),
                                                             g._a()], [g._a()])))
        g.conv(f"{a}.conv_o", *_uconv(rng, H, H, 1))
        g.layernorm(f"enc_p.encoder.norm_layers_1.{l}", *ln(H))
        f = f"enc_p.encoder.ffn_layers.{l}"
        g.conv(f"{f}.conv_1", *_uconv(rng, FF, H, hp["kernel_size"]))
        g.conv(f"{f}.conv_2", *_uconv(rng, H, FF, hp["kernel_size"]))
        g.layernorm(f"enc_p.encoder.norm_layers_2.{l}", *ln(H))
    g.conv("enc_p.proj", *_uconv(rng, 2 * C, H, 1))
    if gin:
        g.gather("emb_g", N(hp["n_speakers"], gin))

    # ---- duration predictor
    def dds(pfx, ch, k):
        for i in range(3):
            g.conv(f"{pfx}.convs_sep.{i}", *_uconv(rng, ch, ch, k, groups=ch), dil=k ** i, groups=ch)
            g.layernorm(f"{pfx}.norms_1.{i}", *ln(ch))
            g.conv(f"{pfx}.convs_1x1.{i}", *_uconv(rng, ch, ch, 1))
            g.layernorm(f"{pfx}.norms_2.{i}", *ln(ch))

    if hp["use_sdp"]:
        Cd, k = H, hp["dp_kernel"]  # models.py:25: filter_channels = in_channels
        g.conv("dp.pre", *_uconv(rng, Cd, H, 1))
        if gin:
            g.conv("dp.cond", *_uconv(rng, Cd, gin, 1))
        dds("dp.convs", Cd, k)
        g.conv("dp.proj", *_uconv(rng, Cd, Cd, 1))
        for fl in (7, 5, 3):
            s = f"dp.flows.{fl}"
            g.conv(f"{s}.pre", *_uconv(rng, Cd, 1, 1))
            dds(f"{s}.convs", Cd, k)
            nb = hp["n_bins"]
            g.conv(f"{s}.proj", N(3 * nb - 1, Cd, 1, std=0.1), N(3 * nb - 1, std=0.1))
        g.nodes.append(_ld(1, _node("Sub", "/dp/flows.0/Sub", [g._a(), g.init("dp.flows.0.m", N(2, 1, std=0.1))], [g._a()])))
        g.anon += 5
        g.nodes.append(_ld(1, _node("Exp", "/dp/flows.0/Exp", [g.init(f"onnx::Exp_{g.anon}", -N(2, 1, std=0.1))], [g._a()])))
    else:
        if gin:
            g.conv("dp.cond", *_uconv(rng, H, gin, 1))
        g.conv("dp.conv_1", *_uconv(rng, 256, H, 3))
        g.layernorm("dp.norm_1", *ln(256))
        g.conv("dp.conv_2", *_uconv(rng, 256, 256, 3))
        g.layernorm("dp.norm_2", *ln(256))
        g.conv("dp.proj", *_uconv(rng, 1, 256, 1))

    # ---- flow (exported in execution order 6,4,2,0; weight-normed convs folded)
    half = C // 2
    nl, fk = hp["flow_wn_layers"], hp["flow_kernel"]
    for idx in reversed(range(0, 2 * hp["flow_layers"], 2)):
        s = f"flow.flows.{idx}"
        g.conv(f"{s}.pre", *_uconv(rng, H, half, 1))
        if gin:
            g.conv(f"{s}.enc.cond_layer", *_uconv(rng, 2 * H * nl, gin, 1), fold=True)
        for i in range(nl):
            g.conv(f"{s}.enc.in_layers.{i}", *_uconv(rng, 2 * H, H, fk), fold=True)
            g.conv(f"{s}.enc.res_skip_layers.{i}", *_uconv(rng, 2 * H if i < nl - 1 else H, H, 1), fold=True)
        g.conv(f"{s}.post", N(half, H, 1, std=0.1), N(half, std=0.1))

    # ---- generator (models.py:299-368); weights ~ N(0, 0.01) (commons.py:11-14) times a gain
    gain = hp["dec_gain"]
    C0 = hp["upsample_initial_channel"]
    w, b = _uconv(rng, C0, C, 7)
    g.conv("dec.conv_pre", w, b)
    if gin:
        g.conv("dec.cond", *_uconv(rng, C0, gin, 1))
    ch = C0
    nk = len(hp["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(hp["upsample_rates"], hp["upsample_kernel_sizes"])):
        fan = (ch // 2) * k
        g.conv(f"dec.ups.{i}", N(ch, ch // 2, k, std=0.01 * gain), rng.uniform(-1, 1, ch // 2).astype(np.float32) / np.sqrt(fan),
               transpose=True, stride=u, pad=(k - u) // 2)
        ch //= 2
        for j, (rk, rd) in enumerate(zip(hp["resblock_kernel_sizes"], hp["resblock_dilation_sizes"])):
            rb = f"dec.resblocks.{i * nk + j}"
            bb = lambda: (rng.uniform(-1, 1, ch) / np.sqrt(ch * rk)).astype(np.float32)
            for q, d in enumerate(rd):
                if hp["resblock"] == "1":
                    g.conv(f"{rb}.convs1.{q}", N(ch, ch, rk, std=0.01 * gain), bb(), dil=d)
                    g.conv(f"{rb}.convs2.{q}", N(ch, ch, rk, std=0.01 * gain), bb(), dil=1)
                else:
                    g.conv(f"{rb}.convs.{q}", N(ch, ch, rk, std=0.01 * gain), bb(), dil=d)
    wp, _ = _uconv(rng, 1, ch, 7)
    g.conv("dec.conv_post", wp * np.float32(hp.get("post_gain", 1.0)), None)

    # ---- ModelProto
    inputs = ["input", "input_lengths", "scales"] + (["sid"] if gin else []) + list(extra_inputs)
    graph = b"".join(g.nodes) + _ld(2, b"main_graph") + b"".join(g.inits)
    graph += b"".join(_ld(11, _ld(1, n.encode())) for n in inputs) + _ld(12, _ld(1, b"output"))
    meta = {"model_type": "vits", "n_speakers": hp["n_speakers"], "n_vocab": V, "sample_rate": 22050,
            "alphabet": "ipa", "phoneme_type": "raw", "phonemizer_model": "", "phoneme_id_map": json.dumps({}),
            "has_espeak": False}
    model = _vi(1, 8) + _ld(2, b"pytorch") + _ld(3, b"2.10.0") + _ld(7, graph) + _ld(8, _ld(1, b"") + _vi(2, 15))
    model += b"".join(_ld(14, _ld(1, k.encode()) + _ld(2, str(v).encode())) for k, v in meta.items())
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    with open(path, "wb") as f:
        f.write(model)
    cfg = {"audio": {"sample_rate": 22050}, "phoneme_type": "raw", "alphabet": "ipa", "lang_code": "en-us",
           "num_symbols": V, "num_speakers": hp["n_speakers"], "speaker_id_map": {},
           "inference": {"noise_scale": 0.667, "length_scale": 1.0, "noise_w": 0.8},
           "phoneme_id_map": {chr(97 + i): i + 1 for i in range(26)}}
    with open(path + ".json", "w") as f:
        json.dump(cfg, f)
    return hp


if __name__ == "__main__":
    import sys
    print(write_voice(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "medium"))
