"""Experiment (GPU box): how the speed of a freshly opened voice depends on what else is allocated on the device.
Finding that motivated it: with the `high` voice's handles still open, `medium` measured 404 M samples/s; after closing
them 523 M.  bench.py therefore closes the headline handles before its secondary measurement."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
torch.cuda.set_device(0)
from phoonnx_amd import MiSession, PipelinedSession
from phoonnx_amd.synth import write_voice
cache = "/tmp/vitsmi_bench"
os.makedirs(cache, exist_ok=True)


def voice(p):
    f = os.path.join(cache, f"synth_{p}.onnx")
    if not os.path.exists(f):
        write_voice(f, p, seed=1234)
    return f


def run(preset, steps=10, parts=2, keep=None):
    s = MiSession(voice(preset))
    pipe = PipelinedSession(s, parts)
    pipe.set_seed(1)
    g = torch.Generator().manual_seed(1234)
    ids = torch.randint(0, 256, (32, 256), generator=g).cuda()
    lens = torch.full((32,), 256, dtype=torch.int64).cuda()
    sc = np.array([0.667, 1.95, 0.8], np.float32)
    pipe.run_device_steps(ids.data_ptr(), lens.data_ptr(), 32, 256, sc, 3)
    pipe.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = pipe.run_device_steps(ids.data_ptr(), lens.data_ptr(), 32, 256, sc, steps)
    pipe.sync()
    dt = time.perf_counter() - t0
    v = int(y.sum()) * s.hparam("hop") / dt
    if keep is None:
        pipe.close()
    else:
        keep.append(pipe)
    return round(v / 1e6, 1), round(dt / steps * 1e3, 2)


print("medium cold", run("medium"))
print("high cold", run("high"))
for gb in (20, 60, 120):
    dummy = torch.empty(gb << 30, dtype=torch.uint8, device="cuda")
    dummy.zero_()
    torch.cuda.synchronize()
    print(f"medium with a {gb} GB torch allocation alive", run("medium"))
    print(f"high   with a {gb} GB torch allocation alive", run("high"))
    del dummy
    torch.cuda.empty_cache()
keep = []
print("high (kept open)", run("high", keep=keep))
print("medium after high, high handles open", run("medium"))
print("high again, first high handles open", run("high"))
for p in keep:
    p.close()
print("medium after closing high", run("medium"))
