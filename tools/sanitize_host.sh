#!/bin/bash
# AddressSanitizer + UBSan over the host-side code that touches untrusted bytes: the .onnx protobuf walker
# and the weight packer (csrc/onnx_reader.cpp, csrc/model.cpp), on the committed fixtures plus damaged copies
# (truncations, random byte flips).  CPU only (GPU sanitizers are not available on the pool).
#   tools/sanitize_host.sh [n_damaged=300]
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
N=${1:-300}
cat > "$W/drv.cpp" <<'EOF'
#include "model.hpp"
#include "g2p_model.hpp"
#include <cstdio>
int main(int argc, char **argv) {
    int bad = 0, ok = 0, g2p_ok = 0;
    for (int i = 1; i < argc; i++) {
        vitsmi::OnnxModel om;
        std::string e = om.load(argv[i]);
        if (!e.empty()) { bad++; continue; }
        vitsmi::Model m;
        e = m.build(om);
        if (e.empty()) ok++; else bad++;
        vitsmi::Model lay;
        lay.build(om, true);      // the layout-only path (vits_open_with_arena)
        vitsmi::G2PModel g;       // ... and the T5 reader on the same bytes
        if (g.build(om).empty()) g2p_ok++;
    }
    printf("loaded %d rejected %d (as T5: %d)\n", ok, bad, g2p_ok);
    return 0;
}
EOF
# -ftrivial-auto-var-init=pattern: ASan does not see reads of uninitialised locals; with pattern-initialised stack
# variables such a read turns into a wild pointer / absurd length that ASan or the reader's own checks then catch
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}   # (g++ 11 lacks -ftrivial-auto-var-init)
"$CXX" -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer \
    -ftrivial-auto-var-init=pattern \
    -I"$R/phoonnx_amd/csrc" "$W/drv.cpp" "$R/phoonnx_amd/csrc/model.cpp" "$R/phoonnx_amd/csrc/onnx_reader.cpp" -o "$W/drv"
python3 - "$R" "$W" "$N" <<'EOF'
import random, sys
root, w, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
rng = random.Random(12)
t5 = open(f"{root}/tests/golden/byt5_tiny.onnx", "rb").read()
for i in range(n // 2):   # damaged copies of the T5 fixture
    b = bytearray(t5)
    if i % 3 == 0:
        b = b[:rng.randrange(len(b))]
    else:
        for _ in range(rng.randrange(1, 12)):
            b[rng.randrange(len(b))] = rng.randrange(256)
    open(f"{w}/t{i}.onnx", "wb").write(bytes(b))
src = open(f"{root}/tests/golden/tiny_rb2_ms.onnx", "rb").read()
for i in range(n):
    b = bytearray(src)
    if i % 3 == 0:
        b = b[:rng.randrange(len(b))]
    else:
        for _ in range(rng.randrange(1, 12)):
            b[rng.randrange(len(b))] = rng.randrange(256)
    open(f"{w}/d{i}.onnx", "wb").write(bytes(b))
# targeted damage (what random flips rarely hit): key bytes switched to another wire type, dims fields rewritten
k = 0
pos = [m for m in range(len(src) - 3) if src[m:m + 3] == b"\x10\x01\x42"]
for m in pos[::3]:
    for repl in (b"\x10\x01\x40", b"\x10\x01\x45", b"\x10\x01\x41"):
        b = bytearray(src); b[m:m + 3] = repl
        open(f"{w}/dk{k}.onnx", "wb").write(bytes(b)); k += 1
    j = m
    while j >= 2 and src[j - 2] == 0x08 and src[j - 1] < 0x80:
        for v in (0, 1, 97):
            b = bytearray(src); b[j - 1] = v
            open(f"{w}/dk{k}.onnx", "wb").write(bytes(b)); k += 1
        j -= 2
for blob in (bytes([0x3a, 4, 0x2a, 2, 0x40, 1]), bytes([0x3a, 4, 0x2a, 2, 0x48, 1]), bytes([0x3a, 4, 0x0a, 2, 0x18, 1]),
             bytes([0x72, 2, 0x08, 1])):
    open(f"{w}/dk{k}.onnx", "wb").write(blob); k += 1
print("targeted cases", k)
EOF
"$W/drv" "$R"/tests/golden/*.onnx "$W"/d*.onnx "$W"/t*.onnx
rm -rf "$W"
