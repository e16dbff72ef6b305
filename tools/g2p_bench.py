#!/usr/bin/env python3
"""ByT5 G2P at full size (run on the GPU box; needs `transformers`, which the image carries):
builds a seeded ByT5-small-shaped T5 (d_model 1472, d_ff 3584, 6 heads x 64, 12 encoder / 4 decoder layers - the shape of
the `g2p-mbyt5-12l` models mul.py:25-29 downloads), exports it to ONNX as the reference's model files are, then
  * parity: logits of the engine vs the transformers model on the same ids (teacher-forced prefix), generated ids equal,
  * speed: ms per generated token of `g2p_generate` (device loop, KV cache) and of the reference's call pattern through
    `session.run` (whole graph per token), next to the transformers model on the host CPU run the way mul.py:192-230 runs
    onnxruntime (whole graph per token, no cache).
Prints one JSON line."""
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from transformers import T5Config, T5ForConditionalGeneration
    small = "--small" in sys.argv
    cfg = T5Config(vocab_size=384, d_model=256 if small else 1472, d_kv=64, d_ff=512 if small else 3584,
                   num_layers=3 if small else 12, num_decoder_layers=2 if small else 4, num_heads=4 if small else 6,
                   relative_attention_num_buckets=32, relative_attention_max_distance=128, dropout_rate=0.0,
                   feed_forward_proj="gated-gelu", tie_word_embeddings=False, decoder_start_token_id=0, pad_token_id=0,
                   eos_token_id=1)
    torch.manual_seed(3)
    m = T5ForConditionalGeneration(cfg).eval()
    path = "/tmp/byt5_bench.onnx"
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda b, c: b

    class Wrap(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, input_ids, attention_mask, decoder_input_ids):
            return self.m(input_ids=input_ids, attention_mask=attention_mask, decoder_input_ids=decoder_input_ids,
                          use_cache=False, return_dict=False)[0]
    ids0 = torch.randint(3, 259, (1, 9))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(Wrap(m), (ids0, torch.ones_like(ids0), torch.tensor([[0, 7, 8]])), path, opset_version=15,
                          input_names=["input_ids", "attention_mask", "decoder_input_ids"], output_names=["logits"],
                          dynamic_axes={"input_ids": {0: "b", 1: "s"}, "attention_mask": {0: "b", 1: "s"},
                                        "decoder_input_ids": {0: "b", 1: "t"}, "logits": {0: "b", 1: "t"}}, dynamo=False)
    from phoonnx_amd.g2p import MiG2PSession, encode_text
    s = MiG2PSession(path)
    text = "The quick brown fox jumps over the lazy dog near the bank of the river."
    ids = encode_text(text, "en-US")
    n_tok = 48
    # --- parity at this size
    with torch.no_grad():
        gen = [0]
        t0 = time.perf_counter()
        for _ in range(n_tok):   # the reference's pattern: the whole graph per token
            lg = m(input_ids=torch.from_numpy(ids), attention_mask=torch.ones(ids.shape, dtype=torch.long),
                   decoder_input_ids=torch.tensor([gen])).logits
            gen.append(int(lg[0, -1].argmax()))
        cpu_s = time.perf_counter() - t0
        ref_logits = m(input_ids=torch.from_numpy(ids), decoder_input_ids=torch.tensor([gen[:16]])).logits.numpy()
    got_logits = s.run(None, {"input_ids": ids, "decoder_input_ids": np.array([gen[:16]], np.int64)})[0]
    err = float(np.abs(got_logits - ref_logits).max())
    dev = s.generate(ids[0], max_length=n_tok, eos_id=-1)
    s.generate(ids[0], max_length=8, eos_id=-1)
    t0 = time.perf_counter()
    dev = s.generate(ids[0], max_length=n_tok, eos_id=-1)
    gen_s = time.perf_counter() - t0
    # encoder + cross keys / values + one decoder step, and the marginal cost of a token (two lengths' difference)
    s.generate(ids[0], max_length=1, eos_id=-1)
    t0 = time.perf_counter()
    for _ in range(5):
        s.generate(ids[0], max_length=1, eos_id=-1)
    first_s = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(3):
        s.generate(ids[0], max_length=4 * n_tok, eos_id=-1)
    long_s = (time.perf_counter() - t0) / 3
    # batched loop: 8 and 16 inputs side by side (same length here), tokens per second over all of them
    batch = {}
    for nb in (4, 8, 16, 64):
        many = [np.roll(ids[0], i) for i in range(nb)]
        s.generate_batch(many, max_length=8, eos_id=-1)
        t0 = time.perf_counter()
        s.generate_batch(many, max_length=n_tok, eos_id=-1)
        dt = time.perf_counter() - t0
        s.generate_batch(many, max_length=1, eos_id=-1)
        t0 = time.perf_counter()
        s.generate_batch(many, max_length=1, eos_id=-1)
        d1 = time.perf_counter() - t0
        batch[f"batch {nb}"] = {"ms_first_step (encoder)": d1 * 1e3, "ms_per_further_step": (dt - d1) / (n_tok - 1) * 1e3,
                                "tokens_per_s": nb * n_tok / dt}
    dec = np.array([[0]], np.int64)
    t0 = time.perf_counter()
    for i in range(n_tok):
        lg = s.run(None, {"input_ids": ids, "decoder_input_ids": dec})[0]
        dec = np.concatenate((dec, np.array([[int(np.argmax(lg[0, -1]))]], np.int64)), axis=1)
    run_s = time.perf_counter() - t0
    print(json.dumps({
        "model": f"ByT5-shaped T5, d_model {cfg.d_model}, d_ff {cfg.d_ff}, {cfg.num_layers}+{cfg.num_decoder_layers} layers, "
                 f"{sum(p.numel() for p in m.parameters()) / 1e6:.0f} M parameters, seeded random weights",
        "input_bytes": int(ids.shape[1]), "tokens": n_tok,
        "logits_max_abs_err_vs_transformers": err, "logits_scale": float(np.abs(ref_logits).max()),
        "generated_ids_equal": dev == gen[1:],
        "ms_first_token (encoder + cross keys/values + one step)": first_s * 1e3,
        "ms_per_further_token": (long_s - first_s) / (4 * n_tok - 1) * 1e3,
        "generate_batch": batch,
        "ms_per_token": {"g2p_generate (device loop, KV cache)": gen_s / n_tok * 1e3,
                         "session.run per token (the reference's call pattern, on the GPU)": run_s / n_tok * 1e3,
                         f"transformers on the host CPU, whole graph per token ({torch.get_num_threads()} threads)": cpu_s / n_tok * 1e3}}))
    s.close()


if __name__ == "__main__":
    main()
