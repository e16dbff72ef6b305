#!/usr/bin/env python3
"""Golden-vector generator.  TEST INFRASTRUCTURE — runs ONLY in the build
container, where the reference checkout exists at /root/reference.

It imports the reference's own PyTorch definition of the exported graph
(`phoonnx_train/vits/models.py:522-722`, `SynthesizerTrn.infer`), builds small
seeded presets, exports them to `.onnx` with the same call the reference uses
(`phoonnx_train/export_onnx.py:240-327`: eval, `dec.remove_weight_norm()`,
forward := infer_forward, opset 15, names input/input_lengths/scales[/sid] ->
output) and records inputs + per-stage outputs as `.npz` fixtures.

Nothing from the reference is copied into the fixtures except numbers: the
`.onnx` holds seeded random weights and the traced graph, the `.npz` holds
inputs and expected outputs.

Noise: the graph has two RandomNormalLike nodes (models.py:111, models.py:718)
whose stream no other backend can reproduce, so goldens are taken (a) at
scales=[0, ls, 0] and (b) with noise arrays injected by patching torch.randn /
torch.randn_like and saved next to the outputs.

Usage:  python oracle/gen_golden.py [--out tests/golden] [--preset NAME ...]
        python oracle/gen_golden.py --big /tmp/vits_big   (medium/high/ms .onnx +
                                     goldens for in-container validation only)
"""
import argparse
import json
import os
import struct
import sys
from unittest import mock

import numpy as np

REF = "/root/reference"

PRESETS = {
    # name: (SynthesizerTrn kwargs overrides)
    "tiny_rb1": dict(
        inter_channels=32, hidden_channels=32, filter_channels=64, n_heads=2, n_layers=2,
        resblock="1", resblock_kernel_sizes=(3, 7, 11),
        resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)),
        upsample_rates=(4, 4, 2, 2), upsample_initial_channel=64,
        upsample_kernel_sizes=(8, 8, 4, 4), n_speakers=1, gin_channels=0, use_sdp=True),
    "tiny_rb2_ms": dict(
        inter_channels=32, hidden_channels=32, filter_channels=64, n_heads=2, n_layers=2,
        resblock="2", resblock_kernel_sizes=(3, 5, 7),
        resblock_dilation_sizes=((1, 2), (2, 6), (3, 12)),
        upsample_rates=(8, 4, 2), upsample_initial_channel=32,
        upsample_kernel_sizes=(16, 8, 4), n_speakers=4, gin_channels=16, use_sdp=True),
    "tiny_dp": dict(
        inter_channels=32, hidden_channels=32, filter_channels=64, n_heads=4, n_layers=1,
        resblock="2", resblock_kernel_sizes=(3, 5),
        resblock_dilation_sizes=((1, 2), (2, 6)),
        upsample_rates=(4, 4), upsample_initial_channel=16,
        upsample_kernel_sizes=(8, 8), n_speakers=1, gin_channels=0, use_sdp=False),
    # Generators whose channel counts are all multiples of 32: these run on the split-operand matrix-core engine
    # (conv_sx_kernel, `vits_hparam "gen_sx"` = 1) that every full-size voice uses.  sx_rb1: a 128-channel stage in the
    # 16-bit plane format (ResBlock1 conv pairs: planes -> planes -> residual) followed by 64- and 32-channel stages in
    # the fp32 raw format.  sx_rb2_ms: ResBlock2 with the speaker path (dec.cond bias into conv_pre).
    "sx_rb1": dict(
        inter_channels=32, hidden_channels=32, filter_channels=64, n_heads=2, n_layers=2,
        resblock="1", resblock_kernel_sizes=(3, 7),
        resblock_dilation_sizes=((1, 3, 5), (1, 3, 5)),
        upsample_rates=(4, 4, 2), upsample_initial_channel=256,
        upsample_kernel_sizes=(8, 8, 4), n_speakers=1, gin_channels=0, use_sdp=True),
    "sx_rb2_ms": dict(
        inter_channels=32, hidden_channels=32, filter_channels=64, n_heads=2, n_layers=2,
        resblock="2", resblock_kernel_sizes=(3, 5, 7),
        resblock_dilation_sizes=((1, 2), (2, 6), (3, 12)),
        upsample_rates=(4, 4, 2), upsample_initial_channel=256,
        upsample_kernel_sizes=(8, 8, 4), n_speakers=4, gin_channels=16, use_sdp=True),
    # full-size presets (not committed; --big only)
    "medium": dict(
        resblock="2", resblock_kernel_sizes=(3, 5, 7),
        resblock_dilation_sizes=((1, 2), (2, 6), (3, 12)),
        upsample_rates=(8, 8, 4), upsample_initial_channel=256,
        upsample_kernel_sizes=(16, 16, 8)),
    "high": dict(
        resblock="1", resblock_kernel_sizes=(3, 7, 11),
        resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)),
        upsample_rates=(8, 8, 2, 2), upsample_initial_channel=512,
        upsample_kernel_sizes=(16, 16, 4, 4)),
    "medium_ms": dict(
        resblock="2", resblock_kernel_sizes=(3, 5, 7),
        resblock_dilation_sizes=((1, 2), (2, 6), (3, 12)),
        upsample_rates=(8, 8, 4), upsample_initial_channel=256,
        upsample_kernel_sizes=(16, 16, 8), n_speakers=4, gin_channels=512),
}

# metadata_props["phoneme_id_map"] of the sx_* fixtures (export_onnx.py:335-345 writes the voice's map there): a voice
# that loads from the .onnx alone, no JSON next to it (tests: TTSVoice.load -> synthesize on the GPU)
META_ID_MAP = {"_": 0, "^": 1, "$": 2, " ": 3, **{chr(97 + i): 4 + i for i in range(26)}, ".": 30, ",": 31, "?": 32}

BASE = dict(  # lightning.py:86-106 defaults
    n_vocab=256, spec_channels=513, segment_size=32, inter_channels=192,
    hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6,
    kernel_size=3, p_dropout=0.1, n_speakers=1, gin_channels=0, use_sdp=True)


def _import_reference():
    if not os.path.isdir(REF):
        sys.exit("gen_golden.py needs the reference checkout at /root/reference")
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import torch  # noqa
    from phoonnx_train.vits import models  # noqa
    return torch, models


def build_model(torch, models, name, seed=1234):
    kw = dict(BASE)
    kw.update(PRESETS[name])
    torch.manual_seed(seed)
    m = models.SynthesizerTrn(**kw).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        # zero / one initialised parameters make flow and spline identities
        # (modules.py:444-445, 493-494): perturb them so they are exercised.
        for n, p in m.named_parameters():
            if n.startswith("enc_q."):
                continue
            if (".post." in n and n.startswith("flow.")) or \
               (n.startswith("dp.flows.") and ".proj." in n) or \
               n.endswith(".gamma") or n.endswith(".beta") or \
               n in ("dp.flows.0.m", "dp.flows.0.logs"):
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
        m.dec.remove_weight_norm()  # export_onnx.py:244
        # as initialised the generator's output is tiny: scale dec weights until
        # the waveform sits in the un-saturated part of tanh
        z = torch.randn(1, kw["inter_channels"], 40, generator=g)
        gg = None
        if kw["n_speakers"] > 1:
            gg = m.emb_g(torch.tensor([1])).unsqueeze(-1)
        for _ in range(60):
            peak = m.dec(z, g=gg).abs().max().item()
            if peak > 0.5:
                break
            for n, p in m.dec.named_parameters():
                if n.endswith("weight") and "cond" not in n:
                    p.mul_(1.08)
        peak = m.dec(z, g=gg).abs().max().item()
        assert 0.05 < peak < 0.97, peak
    return m, kw


def export_onnx(torch, m, kw, path, name="", opset=15, constant_folding=True, keep_initializers=False):
    """Mirror of export_onnx.py:250-327 (this container has no `onnx` package;
    the legacy exporter's only use of it is a post-step that is a no-op here)."""
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda b, c: b

    def infer_forward(text, text_lengths, scales, sid=None):
        audio = m.infer(text, text_lengths, noise_scale=scales[0], length_scale=scales[1],
                        noise_scale_w=scales[2], sid=sid)[0].unsqueeze(1)
        return audio

    old_forward = m.forward
    m.forward = infer_forward
    torch.manual_seed(1234)
    seqs = torch.randint(0, kw["n_vocab"], (1, 50), dtype=torch.long)
    lens = torch.LongTensor([50])
    names = ["input", "input_lengths", "scales"]
    dyn = {"input": {0: "batch_size", 1: "phonemes"}, "input_lengths": {0: "batch_size"},
           "output": {0: "batch_size", 1: "time"}}
    sid = None
    if kw["n_speakers"] > 1:
        sid = torch.LongTensor([0])
        names.append("sid")
        dyn["sid"] = {0: "batch_size"}
    scales = torch.FloatTensor([0.667, 1.0, 0.8])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(model=m, args=(seqs, lens, scales, sid), f=path, verbose=False,
                          opset_version=opset, input_names=names, output_names=["output"],
                          dynamic_axes=dyn, dynamo=False, do_constant_folding=constant_folding,
                          keep_initializers_as_inputs=keep_initializers)
    m.forward = old_forward
    # metadata_props (export_onnx.py:335-350): ModelProto field 14, appended
    meta = {"model_type": "vits", "n_speakers": kw["n_speakers"], "n_vocab": kw["n_vocab"],
            "sample_rate": 22050, "alphabet": "ipa", "phoneme_type": "raw",
            "phonemizer_model": "", "phoneme_id_map": json.dumps(META_ID_MAP if name.startswith("sx_") else {}),
            "has_espeak": False}

    def vint(v):
        out = b""
        while True:
            b7 = v & 0x7F
            v >>= 7
            if v:
                out += bytes([b7 | 0x80])
            else:
                return out + bytes([b7])

    def ld(field, payload):
        return vint((field << 3) | 2) + vint(len(payload)) + payload

    with open(path, "ab") as f:
        for k, v in meta.items():
            f.write(ld(14, ld(1, k.encode()) + ld(2, str(v).encode())))


def run_case(torch, m, kw, ids, lens, scales, sid, noise_dp, noise_z):
    """One `infer` call with injected noise; returns stage tensors."""
    real_randn = torch.randn
    taps = {}

    def fake_randn(*shape, **k):
        if noise_dp is None:
            return real_randn(*shape, **k) * 0
        return torch.from_numpy(noise_dp)

    def fake_randn_like(t, **k):
        if noise_z is None:
            return torch.zeros_like(t)
        assert tuple(t.shape) == noise_z.shape, (t.shape, noise_z.shape)
        return torch.from_numpy(noise_z)

    # hooks for stage taps
    hs = []
    hs.append(m.enc_p.register_forward_hook(
        lambda mod, i, o: taps.update(x=o[0], m_p=o[1], logs_p=o[2])))
    hs.append(m.dp.register_forward_hook(lambda mod, i, o: taps.update(logw=o)))
    with torch.no_grad(), mock.patch.object(torch, "randn", fake_randn), \
            mock.patch.object(torch, "randn_like", fake_randn_like):
        o, attn, y_mask, (z, z_p, m_p, logs_p) = m.infer(
            torch.from_numpy(ids), torch.from_numpy(lens),
            sid=None if sid is None else torch.from_numpy(sid),
            noise_scale=float(scales[0]), length_scale=float(scales[1]),
            noise_scale_w=float(scales[2]))
    for h in hs:
        h.remove()
    w_ceil = attn.sum(2).squeeze(1)  # [B,Tx] durations after masking
    y_lengths = y_mask.sum((1, 2)).long()
    out = dict(
        x=taps["x"], m_p=taps["m_p"], logs_p=taps["logs_p"], logw=taps["logw"],
        w_ceil=w_ceil, y_lengths=y_lengths, z_p=z_p, z=z,
        output=o.unsqueeze(1))
    return {k: v.numpy() for k, v in out.items()}


def frames_for(torch, m, kw, ids, lens, scales, sid, noise_dp):
    r = run_case(torch, m, kw, ids, lens, scales, sid, noise_dp, None)
    return int(r["y_lengths"].max())


def make_cases(torch, m, kw, rng, big=False, long_case=False):
    nv = kw["n_vocab"]
    C = kw["inter_channels"]
    cases = {}

    def add(name, ids, lens, scales, sid, noisy):
        B, T = ids.shape
        scales = np.asarray(scales, np.float32)
        ndp = rng.standard_normal((B, 2, T)).astype(np.float32) if noisy else None
        nz = None
        if noisy:
            F = frames_for(torch, m, kw, ids, lens, scales, sid, ndp)
            nz = rng.standard_normal((B, C, F)).astype(np.float32)
        r = run_case(torch, m, kw, ids, lens, scales, sid, ndp, nz)
        case = dict(ids=ids, lens=lens, scales=scales)
        if sid is not None:
            case["sid"] = sid
        if noisy:
            case["noise_dp"] = ndp
            case["noise_z"] = nz
        for k, v in r.items():
            case["out_" + k] = v
        cases[name] = case
        print(f"   case {name}: B={B} T={T} frames={r['y_lengths'].tolist()} "
              f"peak={np.abs(r['output']).max():.3f}")

    def sids(B):
        if kw["n_speakers"] > 1:
            return rng.integers(0, kw["n_speakers"], size=(B,)).astype(np.int64)
        return None

    def padded(lengths):
        T = max(lengths)
        ids = np.zeros((len(lengths), T), np.int64)
        for b, L in enumerate(lengths):
            ids[b, :L] = rng.integers(0, nv, size=(L,))
        return ids, np.asarray(lengths, np.int64)

    if big:
        ids, lens = padded([40, 33, 17])
        add("b3_noise", ids, lens, [0.667, 1.3, 0.8], sids(3), True)
        ids, lens = padded([64])
        add("b1_zero", ids, lens, [0.0, 1.5, 0.0], sids(1), False)
        # a batch at the bench's length scale: ~400-600 frames per utterance, mixed lengths (every generator stage is many
        # tiles wide, the padding mask is in play) - the real exporter graph at full size
        ids, lens = padded([200, 137, 96, 180])
        add("b4_long_noise", ids, lens, [0.667, 1.95, 0.8], sids(4), True)
        return cases

    ids, lens = padded([40, 33, 17])
    add("b3_noise", ids, lens, [0.667, 1.3, 0.8], sids(3), True)
    ids, lens = padded([23])
    add("b1_zero", ids, lens, [0.0, 1.0, 0.0], sids(1), False)
    ids, lens = padded([7, 12])
    add("b2_zero_ls2", ids, lens, [0.0, 2.0, 0.0], sids(2), False)
    ids, lens = padded([3])   # T < window+1: reference slices rel-emb (attentions.py:295-297)
    add("b1_t3_noise", ids, lens, [0.5, 1.7, 0.6], sids(1), True)
    ids, lens = padded([1, 5])
    add("b2_t1_t5_noise", ids, lens, [0.667, 1.0, 0.8], sids(2), True)
    if long_case:
        # (added last: the earlier cases draw the same random numbers as before)  >= 200 frames: with upsample rates
        # (4, 4, 2) the 128-channel plane-format stage is then >= 800 columns = several 256-column tiles wide, the 64- and
        # 32-channel raw-format stages 3200 / 6400 columns: tile seams, halos and the fused pair / chain kernels' overlap
        # columns are all inside the tensor, against the reference itself
        ids, lens = padded([128, 93])
        add("b2_long_noise", ids, lens, [0.667, 1.6, 0.8], sids(2), True)
    return cases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--preset", nargs="*", default=["tiny_rb1", "tiny_rb2_ms", "tiny_dp"])
    ap.add_argument("--big", default=None, help="directory for medium/high/medium_ms (not committed)")
    # exporter variants: the same seeded model written the ways third-party exports differ from
    # export_onnx.py:318-327 - other opsets, no constant folding, initializers listed as graph inputs.  `--variants DIR`
    # writes <preset>.<variant>.onnx for every --preset (no goldens: the weights, hence the packed arena, are the base file's)
    ap.add_argument("--variants", default=None)
    a = ap.parse_args()
    torch, models = _import_reference()
    torch.set_num_threads(8)
    out = a.big or a.out
    presets = ["medium", "high", "medium_ms"] if a.big and a.preset == ["tiny_rb1", "tiny_rb2_ms", "tiny_dp"] else a.preset
    os.makedirs(out, exist_ok=True)
    if a.variants:
        os.makedirs(a.variants, exist_ok=True)
        for name in a.preset:
            m, kw = build_model(torch, models, name)
            for tag, opt in (("opset11", dict(opset=11)), ("opset13", dict(opset=13)), ("opset17", dict(opset=17)),
                             ("nofold", dict(constant_folding=False)), ("initinputs", dict(keep_initializers=True))):
                path = os.path.join(a.variants, f"{name}.{tag}.onnx")
                try:
                    export_onnx(torch, m, kw, path, name, **opt)
                    print("   wrote", path, os.path.getsize(path), "bytes")
                except Exception as e:  # noqa: BLE001 - an opset the exporter refuses for this graph is a finding, not a failure
                    print("   export failed", tag, type(e).__name__, str(e)[:200])
        return
    for name in presets:
        print("preset", name)
        m, kw = build_model(torch, models, name)
        path = os.path.join(out, f"{name}.onnx")
        export_onnx(torch, m, kw, path, name)
        print("   wrote", path, os.path.getsize(path), "bytes")
        rng = np.random.default_rng(4321)
        cases = make_cases(torch, m, kw, rng, big=bool(a.big), long_case=name.startswith("sx_"))
        flat = {}
        for cname, c in cases.items():
            for k, v in c.items():
                flat[f"{cname}/{k}"] = v
        np.savez_compressed(os.path.join(out, f"{name}.npz"), **flat)
        cfg = {k: (list(map(list, v)) if k == "resblock_dilation_sizes" else
                   list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
        with open(os.path.join(out, f"{name}.hparams.json"), "w") as f:
            json.dump(cfg, f, indent=1)


if __name__ == "__main__":
    main()
