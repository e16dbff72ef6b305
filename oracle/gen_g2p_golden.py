#!/usr/bin/env python3
"""Golden vectors for the ByT5 G2P path.  TEST INFRASTRUCTURE - runs ONLY in the build container.

The reference's G2P hot call is `onnxruntime.InferenceSession.run` on a byte-level T5 encoder-decoder
(`phoonnx/phonemizers/mul.py:106` builds the session, `:192-230` is the greedy loop that re-runs the whole graph for every
generated token).  The model itself is downloaded from Hugging Face at first use (`mul.py:25-29, 71-83`) and is not in
the tree; the arithmetic of the graph is that of Hugging Face transformers' `T5ForConditionalGeneration` (third-party
dependency; transformers 5.15.0 is installed in this image), exported to ONNX.  So the anchor is built here the same way:
a small seeded ByT5-configured model (vocab 384 = 256 bytes + 3 specials + 125 sentinels, gated-GELU feed-forward, untied
lm_head, relative-position buckets 32 / 128), exported with torch.onnx (input_ids, attention_mask, decoder_input_ids ->
logits, the names `mul.py:199-203` feeds), plus inputs and outputs of the transformers model itself:
  * logits of teacher-forced decoder prefixes,
  * the greedy token sequence `mul.py:192-230` would generate,
  * the relative-position bucket tables,
and, from the reference's own file, the outputs of `ByT5Phonemizer._encode_text` / `_decode_phones` (`mul.py:135-170`),
pulled out of mul.py with `ast` (the module itself does not import here: onnxruntime is absent).

Usage: python oracle/gen_g2p_golden.py [--out tests/golden]
"""
import argparse
import ast
import json
import os
import sys
import warnings

import numpy as np

REF = "/root/reference"


# second fixture (round 6): heads of width 64 - every ByT5 / mT5 size has d_kv = 64, and the engine's attention has a
# body of its own for it (g2p.hip g2p_attention_body_t<64>) that the 16-wide tiny model never reaches
VARIANTS = {"byt5_tiny": dict(d_model=96, d_kv=16, d_ff=160, num_layers=3, num_decoder_layers=2, num_heads=4),
            "byt5_dk64": dict(d_model=128, d_kv=64, d_ff=192, num_layers=2, num_decoder_layers=2, num_heads=2)}


def build(seed=7, name="byt5_tiny"):
    import torch
    from transformers import T5Config, T5ForConditionalGeneration
    cfg = T5Config(vocab_size=384, **VARIANTS[name],
                   relative_attention_num_buckets=32, relative_attention_max_distance=128, dropout_rate=0.0,
                   feed_forward_proj="gated-gelu", tie_word_embeddings=False, decoder_start_token_id=0, pad_token_id=0,
                   eos_token_id=1, layer_norm_epsilon=1e-6)
    torch.manual_seed(seed)
    m = T5ForConditionalGeneration(cfg).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():  # as initialised every layer-norm weight is 1 (and would be de-duplicated by the exporter)
        for n, p in m.named_parameters():
            if "layer_norm" in n:
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
            elif "relative_attention_bias" in n:
                p.mul_(3.0)
        # make the greedy path end by itself on some inputs: the EOS row (id 1) shadows the row of the token the
        # untouched model emits sixth for the probe input - wherever that token would have won, EOS now wins
        if name != "byt5_tiny":   # (the second fixture's sequences run to their length limit: long decoder prefixes)
            return torch, m, cfg
        probe = torch.tensor([[3 + b for b in b"<de-DE>: x"]])
        gen = [0]
        for _ in range(16):
            gen.append(int(m(input_ids=probe, decoder_input_ids=torch.tensor([gen])).logits[0, -1].argmax()))
        seq = gen[1:]
        firsts = {t: seq.index(t) for t in seq}
        k = max(firsts, key=lambda t: firsts[t])          # the token whose first appearance is latest
        m.lm_head.weight[1] = m.lm_head.weight[k] * 1.02
    return torch, m, cfg


def export(torch, m, path):
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda b, c: b

    class Wrap(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, input_ids, attention_mask, decoder_input_ids):
            return self.m(input_ids=input_ids, attention_mask=attention_mask, decoder_input_ids=decoder_input_ids,
                          use_cache=False, return_dict=False)[0]

    ids = torch.randint(3, 259, (1, 9))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(Wrap(m), (ids, torch.ones_like(ids), torch.tensor([[0, 7, 8]])), path, opset_version=15,
                          input_names=["input_ids", "attention_mask", "decoder_input_ids"], output_names=["logits"],
                          dynamic_axes={"input_ids": {0: "batch", 1: "src"}, "attention_mask": {0: "batch", 1: "src"},
                                        "decoder_input_ids": {0: "batch", 1: "tgt"}, "logits": {0: "batch", 1: "tgt"}},
                          dynamo=False)


def reference_text_functions():
    """_encode_text / _decode_phones / BYT5_LANGS of the reference, taken from its source file with ast."""
    src = open(os.path.join(REF, "phoonnx", "phonemizers", "mul.py"), encoding="utf-8").read()
    cls = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "ByT5Phonemizer"][0]
    keep = [n for n in cls.body if (isinstance(n, ast.FunctionDef) and n.name in ("_decode_phones", "_encode_text")) or
            (isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "BYT5_LANGS")]
    stub = ast.parse("class ByT5Phonemizer:\n    @classmethod\n    def get_lang(cls, lang):\n        return lang\n")
    stub.body[0].body.extend(keep)
    ns = {"np": np, "List": list, "Dict": dict}
    exec(compile(stub, "mul_subset", "exec"), ns)
    return ns["ByT5Phonemizer"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--name", default="byt5_tiny", choices=sorted(VARIANTS),
                    help="byt5_tiny: the round-2 fixture (also writes byt5_frontend.json); byt5_dk64: heads of width 64, one input "
                         "longer than 256 bytes (cross attention past the engine's 256-key prefetch), model + vectors only")
    a = ap.parse_args()
    torch, m, cfg = build(name=a.name)
    from transformers.models.t5.modeling_t5 import T5Attention
    path = os.path.join(a.out, a.name + ".onnx")
    export(torch, m, path)
    print("wrote", path, os.path.getsize(path), "bytes")
    P = reference_text_functions()
    out = {}
    texts = [("hello world", "en-US"), ("olá, tudo bem?", "pt-PT"), ("x", "de-DE"),
             ("The quick brown fox jumps over the lazy dog near the bank of the river.", "en-GB")]
    if a.name != "byt5_tiny":
        texts = [texts[0], texts[3], ("Pack my box with five dozen liquor jugs; how vexingly quick daft zebras jump! " * 4, "en-US")]
    for i, (text, lang) in enumerate(texts):
        ids = P._encode_text(text, lang)                              # reference: mul.py:152-170
        out[f"c{i}/input_ids"] = ids
        with torch.no_grad():
            tid = torch.from_numpy(ids)
            mask = torch.ones_like(tid)
            gen = [0]
            for _ in range(64):                                       # the greedy loop of mul.py:192-230
                logits = m(input_ids=tid, attention_mask=mask, decoder_input_ids=torch.tensor([gen])).logits
                nxt = int(logits[0, -1].argmax())
                gen.append(nxt)
                if nxt == 1:
                    break
            out[f"c{i}/greedy"] = np.asarray(gen[1:], np.int64)
            dec = torch.tensor([gen[:min(len(gen), 12)]])
            out[f"c{i}/decoder_input_ids"] = dec.numpy()
            out[f"c{i}/logits"] = m(input_ids=tid, attention_mask=mask, decoder_input_ids=dec).logits.numpy()
            enc = m.encoder(input_ids=tid, attention_mask=mask).last_hidden_state
            out[f"c{i}/encoder_out"] = enc.numpy()
        print(f"   case {i}: {ids.shape[1]} input ids, greedy {len(gen) - 1} tokens, ends with EOS: {gen[-1] == 1}")
    rp = torch.arange(-520, 521)[None, :]
    out["bucket/rel"] = rp[0].numpy()
    out["bucket/enc"] = T5Attention._relative_position_bucket(rp, True, 32, 128)[0].numpy()
    out["bucket/dec"] = T5Attention._relative_position_bucket(rp, False, 32, 128)[0].numpy()
    np.savez_compressed(os.path.join(a.out, a.name + ".npz"), **out)
    with open(os.path.join(a.out, a.name + ".hparams.json"), "w") as f:
        json.dump({k: getattr(cfg, k) for k in ("vocab_size", "d_model", "d_kv", "d_ff", "num_layers", "num_decoder_layers",
                                                "num_heads", "relative_attention_num_buckets",
                                                "relative_attention_max_distance", "feed_forward_proj",
                                                "tie_word_embeddings", "layer_norm_epsilon")}, f, indent=1)
    if a.name != "byt5_tiny":
        return
    # the reference's text <-> id functions
    tok = {"added_tokens_decoder": {str(i): {"content": c} for i, c in ((0, "<pad>"), (1, "</s>"), (2, "<unk>"), (259, "<extra_id_0>"))}}
    dec_obj = P.__new__(P)
    dec_obj.tokens = tok["added_tokens_decoder"]
    front = {"langs": P.BYT5_LANGS, "tokenizer_config": tok,
             "encode": [{"text": t, "lang": l, "ids": P._encode_text(t, l)[0].tolist()} for t, l in texts],
             "decode": [{"ids": ids, "text": dec_obj._decode_phones(ids)} for ids in
                        ([107, 104, 111, 1], [3 + b for b in "həˈloʊ".encode()] + [1], [0, 2, 259, 200, 130, 50])]}
    with open(os.path.join(a.out, "byt5_frontend.json"), "w", encoding="utf-8") as f:
        json.dump(front, f, ensure_ascii=False, indent=1)


if __name__ == "__main__":
    main()
