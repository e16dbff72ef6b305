"""CPU restatement (NumPy) of the ByT5 G2P graph - SURVEY §8 f4.  TEST INFRASTRUCTURE.

Only tests/ import this.  The reference runs the graph through onnxruntime (`phoonnx/phonemizers/mul.py:106, 210`); the
arithmetic of the graph is Hugging Face transformers' T5 (`transformers/models/t5/modeling_t5.py`, version 5.15.0 in this
image: `T5LayerNorm`, `T5Attention.forward`, `_relative_position_bucket`, `T5DenseGatedActDense`, `T5Stack`,
`T5ForConditionalGeneration.forward`), which is third-party and not in the reference tree; this file restates that
published algorithm and is pinned to outputs of the transformers model itself (`oracle/gen_g2p_golden.py` ->
`tests/golden/byt5_tiny.*`, checked by `tests/test_g2p_oracle.py`).

Weights are read from the `.onnx` with the oracle's own walker by following nodes (`/encoder/block.N/layer.M/.../MatMul`),
as for the VITS graph: Linear weights appear as anonymous transposed `onnx::MatMul_N` initializers.
"""
import math
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from onnx_walk import OnnxModel  # noqa: E402


def relative_position_bucket(rel, bidirectional, num_buckets=32, max_distance=128):
    """modeling_t5.py `T5Attention._relative_position_bucket`, with its float32 arithmetic (the bucket is an integer:
    it has to come out the same, not close)."""
    rel = np.asarray(rel, np.int64)
    out = np.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        out += (rel > 0).astype(np.int64) * num_buckets
        n = np.abs(rel)
    else:
        n = -np.minimum(rel, 0)
    max_exact = num_buckets // 2
    small = n < max_exact
    with np.errstate(divide="ignore"):
        v = np.log(n.astype(np.float32) / np.float32(max_exact)) / np.float32(math.log(max_distance / max_exact)) * \
            np.float32(num_buckets - max_exact)
    large = max_exact + np.where(np.isfinite(v), v, 0).astype(np.int64)
    large = np.minimum(large, num_buckets - 1)
    return out + np.where(small, n, large)


def gelu_new(x):
    return 0.5 * x * (1.0 + np.tanh(np.float32(math.sqrt(2.0 / math.pi)) * (x + np.float32(0.044715) * x * x * x)))


class T5Oracle:
    def __init__(self, onnx_path, max_distance=128, eps=1e-6):
        self.model = OnnxModel(onnx_path)
        self.max_distance, self.eps = max_distance, np.float32(eps)
        w = {}
        for n in self.model.nodes:
            if not n.name:
                continue
            m = re.search(r"(encoder|decoder)/block\.(\d+)/layer\.(\d+)/(\w+)/(\w+)/(MatMul|Gather)$", n.name)
            if m and n.op in ("MatMul", "Gather"):
                stack, blk, lay, mod, leaf, _ = m.groups()
                t = self.model.tensor(n.inputs[1] if n.op == "MatMul" else n.inputs[0])
                if t is not None:
                    w[f"{stack}.{blk}.{lay}.{mod}.{leaf}"] = np.asarray(t, np.float32)
                continue
            m = re.search(r"(encoder|decoder)/block\.(\d+)/layer\.(\d+)/layer_norm/Mul(_\d+)?$", n.name)
            if m and n.op == "Mul":
                for i in n.inputs:
                    t = self.model.init.get(i)
                    if t is not None and t.ndim == 1:
                        w[f"{m.group(1)}.{m.group(2)}.{m.group(3)}.layer_norm"] = np.asarray(t, np.float32)
                continue
            m = re.search(r"(encoder|decoder)/final_layer_norm/Mul(_\d+)?$", n.name)
            if m and n.op == "Mul":
                for i in n.inputs:
                    t = self.model.init.get(i)
                    if t is not None and t.ndim == 1:
                        w[f"{m.group(1)}.final_layer_norm"] = np.asarray(t, np.float32)
                continue
            if n.op == "Gather" and n.name.endswith("encoder/embed_tokens/Gather"):
                w["shared"] = np.asarray(self.model.tensor(n.inputs[0]), np.float32)
            if n.op == "MatMul" and n.name.endswith("lm_head/MatMul"):
                w["lm_head"] = np.asarray(self.model.tensor(n.inputs[1]), np.float32)
        self.w = w
        self.d_model = w["shared"].shape[1]
        self.n_enc = 1 + max(int(k.split(".")[1]) for k in w if k.startswith("encoder.") and k[8].isdigit())
        self.n_dec = 1 + max(int(k.split(".")[1]) for k in w if k.startswith("decoder.") and k[8].isdigit())
        rb = w["encoder.0.0.SelfAttention.relative_attention_bias"]
        self.num_buckets, self.heads = rb.shape
        self.inner = w["encoder.0.0.SelfAttention.q"].shape[1]
        self.d_kv = self.inner // self.heads
        # T5 v1.0 checkpoints (tied embeddings) scale the decoder output by d_model^-0.5 before lm_head
        # (modeling_t5.py, T5ForConditionalGeneration.forward): in the graph that is a Mul between the decoder's final
        # layer norm and /lm_head/MatMul.  ByT5 / v1.1 graphs have none.
        prod = {o: n for n in self.model.nodes for o in n.outputs}
        lm = [n for n in self.model.nodes if n.op == "MatMul" and n.name.endswith("lm_head/MatMul")][0]
        p = prod.get(lm.inputs[0])
        self.tied = p is not None and p.op == "Mul" and "final_layer_norm" not in p.name

    def rms(self, x, g):
        var = np.mean(x.astype(np.float32) ** 2, axis=-1, keepdims=True, dtype=np.float32)
        return x * (1.0 / np.sqrt(var + self.eps)).astype(np.float32) * g

    def attention(self, pfx, xq, xkv, bias):
        """xq [Tq, d], xkv [Tk, d], bias [heads, Tq, Tk] -> [Tq, d]   (no 1/sqrt(d) scaling in T5)"""
        w = self.w
        q = (xq @ w[pfx + ".q"]).reshape(-1, self.heads, self.d_kv).transpose(1, 0, 2)
        k = (xkv @ w[pfx + ".k"]).reshape(-1, self.heads, self.d_kv).transpose(1, 0, 2)
        v = (xkv @ w[pfx + ".v"]).reshape(-1, self.heads, self.d_kv).transpose(1, 0, 2)
        s = q @ k.transpose(0, 2, 1) + bias
        s = s - s.max(-1, keepdims=True)
        p = np.exp(s)
        p /= p.sum(-1, keepdims=True)
        o = (p @ v).transpose(1, 0, 2).reshape(-1, self.inner)
        return (o @ w[pfx + ".o"]).astype(np.float32)

    def ffn(self, pfx, x):
        w = self.w
        if pfx + ".wi_0" in w:  # T5DenseGatedActDense, gated-gelu = gelu_new
            h = gelu_new(x @ w[pfx + ".wi_0"]) * (x @ w[pfx + ".wi_1"])
        else:                   # T5DenseActDense, relu
            h = np.maximum(x @ w[pfx + ".wi"], 0)
        return (h @ w[pfx + ".wo"]).astype(np.float32)

    def position_bias(self, stack, Tq, Tk, q_off=0):
        rel = np.arange(Tk)[None, :] - (np.arange(Tq)[:, None] + q_off)       # memory - context
        b = relative_position_bucket(rel, stack == "encoder", self.num_buckets, self.max_distance)
        table = self.w[f"{stack}.0.0.SelfAttention.relative_attention_bias"]   # [buckets, heads]; block 0's, shared
        return table[b].transpose(2, 0, 1).astype(np.float32)

    def encode(self, input_ids):
        x = self.w["shared"][np.asarray(input_ids, np.int64)]
        T = x.shape[0]
        bias = self.position_bias("encoder", T, T)
        for b in range(self.n_enc):
            x = x + self.attention(f"encoder.{b}.0.SelfAttention", self.rms(x, self.w[f"encoder.{b}.0.layer_norm"]),
                                   self.rms(x, self.w[f"encoder.{b}.0.layer_norm"]), bias)
            x = x + self.ffn(f"encoder.{b}.1.DenseReluDense", self.rms(x, self.w[f"encoder.{b}.1.layer_norm"]))
        return self.rms(x, self.w["encoder.final_layer_norm"]).astype(np.float32)

    def decode(self, enc, decoder_input_ids):
        x = self.w["shared"][np.asarray(decoder_input_ids, np.int64)]
        T, S = x.shape[0], enc.shape[0]
        bias = self.position_bias("decoder", T, T) + np.triu(np.full((T, T), -np.inf, np.float32), 1)[None]
        zero = np.zeros((self.heads, T, S), np.float32)
        for b in range(self.n_dec):
            h = self.rms(x, self.w[f"decoder.{b}.0.layer_norm"])
            x = x + self.attention(f"decoder.{b}.0.SelfAttention", h, h, bias)
            x = x + self.attention(f"decoder.{b}.1.EncDecAttention", self.rms(x, self.w[f"decoder.{b}.1.layer_norm"]), enc, zero)
            x = x + self.ffn(f"decoder.{b}.2.DenseReluDense", self.rms(x, self.w[f"decoder.{b}.2.layer_norm"]))
        x = self.rms(x, self.w["decoder.final_layer_norm"])
        if self.tied:
            x = x * np.float32(self.d_model ** -0.5)
        return (x @ self.w["lm_head"]).astype(np.float32)

    def logits(self, input_ids, decoder_input_ids):
        """What `session.run(..., {"input_ids", "attention_mask", "decoder_input_ids"})[0]` returns for batch 1."""
        return self.decode(self.encode(input_ids), decoder_input_ids)[None]

    def greedy(self, input_ids, max_length=512, start=0, eos=1):
        """The loop of mul.py:192-230."""
        enc = self.encode(input_ids)
        dec, out = [start], []
        for _ in range(max_length):
            nxt = int(np.argmax(self.decode(enc, dec)[-1]))
            out.append(nxt)
            if nxt == eos:
                break
            dec.append(nxt)
        return out
