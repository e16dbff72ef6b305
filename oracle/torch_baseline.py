"""Torch-CPU functional restatement of the exported graph.  TEST / MEASUREMENT INFRASTRUCTURE.

Why a second CPU implementation next to oracle/vits_oracle.c: the reference's hot call is onnxruntime's CPU
execution provider with default session options, i.e. an optimised multi-threaded conv/GEMM library on all host
cores (phoonnx/voice.py:167-171).  onnxruntime is not installed on the build or GPU images, and the C oracle is a
checker written for clarity, not speed.  This file gives `bench.py`'s `cpu_baseline` leg the closest stand-in
available: the same graph evaluated op by op with PyTorch's CPU kernels (oneDNN / MKL convolutions, all cores) on
the same `.onnx` weights.  It is written from SURVEY.md App. A (the op-level semantics of
phoonnx_train/vits/{models,modules,attentions,commons,transforms}.py, cited per function below), never from the
reference's files, and is pinned to the reference-generated fixtures by tests/test_torch_baseline.py.

Only tests/ and bench.py's cpu_baseline leg import this; the product path never does.
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from onnx_walk import OnnxModel  # noqa: E402
from vits_oracle import resolve_weights  # noqa: E402  (pure-Python graph walk; no C involved)


def _seq_mask(lens, T):
    return (torch.arange(T)[None, :] < lens[:, None]).to(torch.float32)[:, None, :]  # commons.py:109-113


class TorchVits:
    def __init__(self, onnx_path, threads=None):
        if threads:
            torch.set_num_threads(int(threads))
        model = OnnxModel(onnx_path)
        tensors, self.ints = resolve_weights(model)
        self.t = {k: torch.from_numpy(np.array(v, dtype=np.float32, copy=True)) for k, v in tensors.items()}
        t = self.t
        self.H = t["enc_p.emb.weight"].shape[1]
        self.C = t["enc_p.proj.weight"].shape[0] // 2
        self.n_layers = sum(1 for k in t if k.startswith("enc_p.encoder.attn_layers.") and k.endswith("conv_q.weight"))
        self.dk = t["enc_p.encoder.attn_layers.0.emb_rel_k"].shape[2]
        self.heads = self.H // self.dk
        self.window = (t["enc_p.encoder.attn_layers.0.emb_rel_k"].shape[1] - 1) // 2
        self.gin = t["emb_g.weight"].shape[1] if "emb_g.weight" in t else 0
        self.use_sdp = "dp.flows.0.m" in t
        self.n_flows = sum(1 for k in t if k.startswith("flow.flows.") and k.endswith(".pre.weight"))
        self.n_ups = sum(1 for k in t if k.startswith("dec.ups.") and k.endswith(".weight"))
        self.rb1 = "dec.resblocks.0.convs1.0.weight" in t
        self.n_rb = 0
        while (f"dec.resblocks.{self.n_rb}.convs1.0.weight" in t) or (f"dec.resblocks.{self.n_rb}.convs.0.weight" in t):
            self.n_rb += 1
        self.hop = 1
        for i in range(self.n_ups):
            self.hop *= self.ints[f"dec.ups.{i}.stride"]

    # ------------------------------------------------------------------ primitives
    def conv(self, name, x, dil=None, pad=None):
        w = self.t[name + ".weight"]
        b = self.t.get(name + ".bias")
        d = self.ints.get(name + ".dilation", 1) if dil is None else dil
        k = w.shape[2]
        p = d * (k - 1) // 2 if pad is None else pad
        return F.conv1d(x, w, b, dilation=d, padding=p, groups=self.ints.get(name + ".group", 1))

    def ln(self, name, x):  # modules.py:14-26: over channels, biased variance, eps 1e-5
        return F.layer_norm(x.transpose(1, 2), (x.shape[1],), self.t[name + ".gamma"], self.t[name + ".beta"],
                            1e-5).transpose(1, 2)

    # ------------------------------------------------------------------ A1-A4 text encoder
    def mha(self, pfx, x, mask):  # attentions.py:215-272
        B, H, T = x.shape
        h, dk, w = self.heads, self.dk, self.window
        q = self.conv(pfx + ".conv_q", x).view(B, h, dk, T).transpose(2, 3) / math.sqrt(dk)
        k = self.conv(pfx + ".conv_k", x).view(B, h, dk, T).transpose(2, 3)
        v = self.conv(pfx + ".conv_v", x).view(B, h, dk, T).transpose(2, 3)
        scores = q @ k.transpose(2, 3)
        ek, ev = self.t[pfx + ".emb_rel_k"][0], self.t[pfx + ".emb_rel_v"][0]      # [2w+1, dk], shared across heads
        rel = q @ ek.t()                                                           # [B,h,T,2w+1]: q_i . E_k[d+w]
        idx = torch.arange(T)
        d = idx[None, :] - idx[:, None]                                            # j - i
        near = (d.abs() <= w)
        scores = scores + torch.where(near, rel.gather(3, (d.clamp(-w, w) + w).expand(B, h, T, T)), torch.zeros(()))
        am = (mask.unsqueeze(2) * mask.unsqueeze(-1))                              # [B,1,T,T]
        scores = scores.masked_fill(am == 0, -1e4)
        p = torch.softmax(scores, dim=-1)
        out = p @ v
        # relative values: out_i += sum_{|j-i|<=w} p[i,j] E_v[j-i+w]
        pw = torch.zeros(B, h, T, 2 * w + 1)
        pw.scatter_add_(3, (d.clamp(-w, w) + w).expand(B, h, T, T), torch.where(near, p, torch.zeros(())))
        out = out + pw @ ev
        out = out.transpose(2, 3).reshape(B, H, T)
        return self.conv(pfx + ".conv_o", out)

    def text_encoder(self, ids, lens):
        T = ids.shape[1]
        mask = _seq_mask(lens, T)
        x = (self.t["enc_p.emb.weight"][ids] * math.sqrt(self.H)).transpose(1, 2) * mask   # models.py:199-203
        for l in range(self.n_layers):
            x = self.ln(f"enc_p.encoder.norm_layers_1.{l}", x + self.mha(f"enc_p.encoder.attn_layers.{l}", x, mask))
            f = f"enc_p.encoder.ffn_layers.{l}"
            y = self.conv(f + ".conv_2", torch.relu(self.conv(f + ".conv_1", x * mask)) * mask) * mask
            x = self.ln(f"enc_p.encoder.norm_layers_2.{l}", x + y)
        x = x * mask
        stats = self.conv("enc_p.proj", x) * mask
        return x, stats[:, :self.C], stats[:, self.C:], mask

    # ------------------------------------------------------------------ A5 duration predictors
    def ddsconv(self, pfx, x, mask, g=None):  # modules.py:117-129
        if g is not None:
            x = x + g
        for i in range(3):
            if f"{pfx}.convs_sep.{i}.weight" not in self.t:
                break
            y = self.conv(f"{pfx}.convs_sep.{i}", x * mask)
            y = F.gelu(self.ln(f"{pfx}.norms_1.{i}", y))
            y = self.conv(f"{pfx}.convs_1x1.{i}", y)
            y = F.gelu(self.ln(f"{pfx}.norms_2.{i}", y))
            x = x + y
        return x * mask

    @staticmethod
    def rqs_inverse(x, W, Hh, D, tail=5.0):  # transforms.py:50-191, inverse branch
        nb = W.shape[-1]
        inside = (x >= -tail) & (x <= tail)
        const = math.log(math.exp(1 - 1e-3) - 1)
        D = F.pad(D, (1, 1), value=const)
        mbw = mbh = md = 1e-3

        def knots(u, m):
            w = m + (1 - m * nb) * torch.softmax(u, dim=-1)
            c = F.pad(torch.cumsum(w, dim=-1), (1, 0)) * (2 * tail) - tail
            c[..., 0], c[..., -1] = -tail, tail
            return c, c[..., 1:] - c[..., :-1]

        cw, widths = knots(W, mbw)
        ch, heights = knots(Hh, mbh)
        derivs = md + F.softplus(D)
        loc = ch.clone()
        loc[..., -1] += 1e-6
        xin = x.clamp(-tail, tail)
        b = ((xin[..., None] >= loc).sum(-1) - 1).clamp(0, nb - 1)[..., None]
        g = lambda a: a.gather(-1, b)[..., 0]
        icw, iw, ich, ih = g(cw), g(widths), g(ch), g(heights)
        delta = ih / iw
        d0, d1 = g(derivs), g(derivs[..., 1:])
        a_ = (xin - ich) * (d0 + d1 - 2 * delta) + ih * (delta - d0)
        b_ = ih * d0 - (xin - ich) * (d0 + d1 - 2 * delta)
        c_ = -delta * (xin - ich)
        root = (2 * c_) / (-b_ - torch.sqrt(b_ * b_ - 4 * a_ * c_))
        return torch.where(inside, root * iw + icw, x)

    def convflow_reverse(self, pfx, z, cond, mask):  # modules.py:496-527
        x0, x1 = z[:, :1], z[:, 1:]
        h = self.conv(pfx + ".pre", x0)
        h = self.ddsconv(pfx + ".convs", h, mask, g=cond)
        h = self.conv(pfx + ".proj", h) * mask                    # [B, 3nb-1, T]
        nb = (h.shape[1] + 1) // 3
        Cf = self.t[pfx + ".pre.weight"].shape[0]
        hp = h.permute(0, 2, 1)
        W, Hh, D = hp[..., :nb] / math.sqrt(Cf), hp[..., nb:2 * nb] / math.sqrt(Cf), hp[..., 2 * nb:]
        x1 = self.rqs_inverse(x1[:, 0], W, Hh, D)[:, None, :]
        return torch.cat([x0, x1], 1) * mask

    def sdp_reverse(self, x, mask, g, noise_dp, noise_w):  # models.py:63-70,108-117
        h = self.conv("dp.pre", x)
        if g is not None:
            h = h + self.conv("dp.cond", g)
        h = self.ddsconv("dp.convs", h, mask)
        cond = self.conv("dp.proj", h) * mask
        z = noise_dp * noise_w
        for fl in (7, 5, 3):
            z = torch.flip(z, [1])
            z = self.convflow_reverse(f"dp.flows.{fl}", z, cond, mask)
        z = torch.flip(z, [1])
        z = (z - self.t["dp.flows.0.m"].view(1, 2, 1)) * torch.exp(-self.t["dp.flows.0.logs"].view(1, 2, 1)) * mask
        return z[:, :1]

    def dp_plain(self, x, mask, g):  # models.py:138-165
        if g is not None:
            x = x + self.conv("dp.cond", g)
        h = self.ln("dp.norm_1", torch.relu(self.conv("dp.conv_1", x * mask)))
        h = self.ln("dp.norm_2", torch.relu(self.conv("dp.conv_2", h * mask)))
        return self.conv("dp.proj", h * mask) * mask

    # ------------------------------------------------------------------ A7 flow
    def wn(self, pfx, x, mask, g):  # modules.py:184-209
        Hf = x.shape[1]
        out = torch.zeros_like(x)
        gc = self.conv(pfx + ".cond_layer", g) if g is not None else None
        i = 0
        while f"{pfx}.in_layers.{i}.weight" in self.t:
            a = self.conv(f"{pfx}.in_layers.{i}", x)
            if gc is not None:
                a = a + gc[:, 2 * Hf * i:2 * Hf * (i + 1)]
            acts = torch.tanh(a[:, :Hf]) * torch.sigmoid(a[:, Hf:])
            rs = self.conv(f"{pfx}.res_skip_layers.{i}", acts)
            if rs.shape[1] == 2 * Hf:
                x = (x + rs[:, :Hf]) * mask
                out = out + rs[:, Hf:]
            else:
                out = out + rs
            i += 1
        return out * mask

    def flow_reverse(self, z, mask, g):  # models.py:247-254, modules.py:447-466
        half = self.C // 2
        for idx in reversed(range(0, 2 * self.n_flows, 2)):
            z = torch.flip(z, [1])
            x0, x1 = z[:, :half], z[:, half:]
            h = self.conv(f"flow.flows.{idx}.pre", x0) * mask
            h = self.wn(f"flow.flows.{idx}.enc", h, mask, g)
            m = self.conv(f"flow.flows.{idx}.post", h) * mask
            z = torch.cat([x0, (x1 - m) * mask], 1)
        return z

    # ------------------------------------------------------------------ A8 generator
    def generator(self, z, g):  # models.py:348-368
        x = self.conv("dec.conv_pre", z)
        if g is not None:
            x = x + self.conv("dec.cond", g)
        nk = self.n_rb // self.n_ups
        for i in range(self.n_ups):
            x = F.leaky_relu(x, 0.1)
            w, b = self.t[f"dec.ups.{i}.weight"], self.t.get(f"dec.ups.{i}.bias")
            x = F.conv_transpose1d(x, w, b, stride=self.ints[f"dec.ups.{i}.stride"], padding=self.ints[f"dec.ups.{i}.pad"])
            xs = None
            for j in range(nk):
                rb = f"dec.resblocks.{i * nk + j}"
                y = x
                q = 0
                while (f"{rb}.convs1.{q}.weight" if self.rb1 else f"{rb}.convs.{q}.weight") in self.t:
                    if self.rb1:  # modules.py:301-314
                        t_ = self.conv(f"{rb}.convs1.{q}", F.leaky_relu(y, 0.1))
                        y = self.conv(f"{rb}.convs2.{q}", F.leaky_relu(t_, 0.1)) + y
                    else:         # modules.py:355-364
                        y = self.conv(f"{rb}.convs.{q}", F.leaky_relu(y, 0.1)) + y
                    q += 1
                xs = y if xs is None else xs + y
            x = xs / nk
        x = F.leaky_relu(x)  # default slope 0.01 (:364)
        return torch.tanh(self.conv("dec.conv_post", x))

    # ------------------------------------------------------------------ SynthesizerTrn.infer (models.py:681-722)
    @torch.no_grad()
    def infer(self, ids, lens, scales, sid=None, noise_dp=None, noise_z=None):
        ids = torch.as_tensor(np.asarray(ids, np.int64))
        lens = torch.as_tensor(np.asarray(lens, np.int64))
        noise_scale, length_scale, noise_w = (float(v) for v in scales)
        B, T = ids.shape
        x, m_p, logs_p, mask = self.text_encoder(ids, lens)
        g = None
        if self.gin:
            if sid is None:
                raise RuntimeError("Missing speaker id")
            g = self.t["emb_g.weight"][torch.as_tensor(np.asarray(sid, np.int64))].unsqueeze(-1)
        if self.use_sdp:
            ndp = torch.zeros(B, 2, T) if noise_dp is None else torch.as_tensor(np.asarray(noise_dp, np.float32))
            logw = self.sdp_reverse(x, mask, g, ndp, noise_w)
        else:
            logw = self.dp_plain(x, mask, g)
        w_ceil = torch.ceil(torch.exp(logw) * mask * length_scale)
        y_len = torch.clamp_min(w_ceil.sum((1, 2)), 1).long()
        Fm = int(y_len.max())
        y_mask = _seq_mask(y_len, Fm)
        cum = torch.cumsum(w_ceil[:, 0], 1)                                   # [B,T]
        f = torch.arange(Fm, dtype=torch.float32)
        tok = (f[None, :, None] >= cum[:, None, :]).sum(-1).clamp(max=T - 1)  # frame -> token (commons.py:116-129)
        valid = y_mask * mask.gather(2, tok[:, None, :])
        m_e = m_p.gather(2, tok[:, None, :].expand(B, self.C, Fm)) * valid
        l_e = logs_p.gather(2, tok[:, None, :].expand(B, self.C, Fm)) * valid
        if noise_z is None:
            eps = torch.zeros(B, self.C, Fm)
        else:
            eps = torch.as_tensor(np.asarray(noise_z, np.float32))[:, :, :Fm]
        z_p = m_e + eps * torch.exp(l_e) * noise_scale
        z = self.flow_reverse(z_p, y_mask, g)
        o = self.generator(z * y_mask, g)
        return {"output": o.unsqueeze(1).numpy(), "y_lengths": y_len.numpy(), "w_ceil": w_ceil[:, 0].numpy(),
                "z": z.numpy(), "z_p": z_p.numpy(), "logw": logw.numpy(), "x": x.numpy(), "m_p": m_p.numpy(),
                "logs_p": logs_p.numpy()}


def _worker(argv):
    """`python torch_baseline.py --worker <voice.onnx> <tokens> <length_scale> <threads> <seconds> <start_epoch>`: one of P
    processes that bench.py's cpu_baseline starts side by side (P x `threads` host threads): utterance after utterance at
    B = 1 - the reference's own call shape (voice.py:350-351, sentences one by one) - from `start_epoch` for `seconds`;
    prints "samples seconds"."""
    import time
    voice, tokens, ls, threads, seconds, start = argv[0], int(argv[1]), float(argv[2]), int(argv[3]), float(argv[4]), float(argv[5])
    torch.set_num_threads(threads)
    m = TorchVits(voice)
    rng = np.random.default_rng(os.getpid())
    scales = np.array([0.667, ls, 0.8], np.float32)
    ids = rng.integers(0, 256, size=(1, tokens)).astype(np.int64)
    lens = np.full((1,), tokens, np.int64)
    ndp = rng.standard_normal((1, 2, tokens)).astype(np.float32)
    nz = rng.standard_normal((1, m.C, tokens * 12)).astype(np.float32)
    m.infer(ids[:, :64], np.full((1,), 64, np.int64), scales, None, ndp[:, :, :64], nz)  # warm-up
    while time.time() < start:
        time.sleep(0.01)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        r = m.infer(ids, lens, scales, None, ndp, nz)
        n += int(np.asarray(r["y_lengths"]).sum()) * m.hop
    print(n, time.perf_counter() - t0, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        _worker(sys.argv[2:])

