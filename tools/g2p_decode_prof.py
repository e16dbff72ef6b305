#!/usr/bin/env python3
"""The G2P decoder step alone, for a kernel trace: builds the seeded ByT5-small-shaped model of tools/g2p_bench.py (or reuses
/tmp/byt5_bench.onnx), then ONLY runs `generate` (device loop, KV cache) for --tokens tokens, --reps times.

    rocprofv3 --kernel-trace --output-format csv -d out -o g -- python3 tools/g2p_decode_prof.py --tokens 192
    python3 tools/g2p_decode_prof.py --summarise out        # per-kernel table of the trace (calls per token, us, share)
"""
import argparse
import csv
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def export(path):
    import warnings
    import torch
    from transformers import T5Config, T5ForConditionalGeneration
    cfg = T5Config(vocab_size=384, d_model=1472, d_kv=64, d_ff=3584, num_layers=12, num_decoder_layers=4, num_heads=6,
                   relative_attention_num_buckets=32, relative_attention_max_distance=128, dropout_rate=0.0,
                   feed_forward_proj="gated-gelu", tie_word_embeddings=False, decoder_start_token_id=0, pad_token_id=0,
                   eos_token_id=1)
    torch.manual_seed(3)
    m = T5ForConditionalGeneration(cfg).eval()
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


def summarise(d, tokens):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    agg = {}
    for r in rows:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    # one token's launches in order (the last complete token of the trace): kernel, grid, duration
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for r in rows]
    ends = [i for i, n in enumerate(names) if n.startswith("g2p_argmax")]
    if len(ends) >= 3:
        lo, hi = ends[-3] + 1, ends[-2] + 1
        t0 = int(rows[lo]["Start_Timestamp"])
        print(f"one token = {hi - lo} launches, {(int(rows[hi - 1]['End_Timestamp']) - t0) / 1e3:.1f} us from the first start to the last end:")
        for i in range(lo, hi):
            r = rows[i]
            gap = (int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3
            print(f"    {names[i][:28]:28s} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8s} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>4s}"
                  f"  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.2f} us  (gap before {gap:5.2f})")
    tot = sum(a[1] for a in agg.values())
    print(f"{len(rows)} dispatches, {tot / 1e3:.2f} ms of kernel time" + (f", {tot / tokens:.1f} us per token" if tokens else ""))
    for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n[:70]:70s} {c:7d} x {us / c:8.2f} us  {100 * us / tot:5.1f} %" + (f"  {c / tokens:6.2f} per token" if tokens else ""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=192)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--summarise", default=None)
    ap.add_argument("--summarise-tokens", type=int, default=0)
    a = ap.parse_args()
    if a.summarise:
        return summarise(a.summarise, a.summarise_tokens)
    path = "/tmp/byt5_bench.onnx"
    if not os.path.exists(path):
        export(path)
    from phoonnx_amd.g2p import MiG2PSession, encode_text
    s = MiG2PSession(path)
    ids = encode_text("The quick brown fox jumps over the lazy dog near the bank of the river.", "en-US")
    s.generate(ids[0], max_length=8, eos_id=-1)
    t1 = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        s.generate(ids[0], max_length=1, eos_id=-1)
        t1.append(time.perf_counter() - t0)
    tn = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        s.generate(ids[0], max_length=a.tokens, eos_id=-1)
        tn.append(time.perf_counter() - t0)
    print(json.dumps({"tokens": a.tokens, "ms_first_token": 1e3 * float(np.median(t1)),
                      "ms_per_further_token": 1e3 * (float(np.median(tn)) - float(np.median(t1))) / (a.tokens - 1)}))
    s.close()


if __name__ == "__main__":
    main()
