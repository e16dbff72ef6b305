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
#include <cstdio>
int main(int argc, char **argv) {
    int bad = 0, ok = 0;
    for (int i = 1; i < argc; i++) {
        vitsmi::OnnxModel om;
        std::string e = om.load(argv[i]);
        if (!e.empty()) { bad++; continue; }
        vitsmi::Model m;
        e = m.build(om);
        if (e.empty()) ok++; else bad++;
    }
    printf("loaded %d rejected %d\n", ok, bad);
    return 0;
}
EOF
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer \
    -I"$R/phoonnx_amd/csrc" "$W/drv.cpp" "$R/phoonnx_amd/csrc/model.cpp" "$R/phoonnx_amd/csrc/onnx_reader.cpp" -o "$W/drv"
python3 - "$R" "$W" "$N" <<'EOF'
import random, sys
root, w, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
src = open(f"{root}/tests/golden/tiny_rb2_ms.onnx", "rb").read()
rng = random.Random(12)
for i in range(n):
    b = bytearray(src)
    if i % 3 == 0:
        b = b[:rng.randrange(len(b))]
    else:
        for _ in range(rng.randrange(1, 12)):
            b[rng.randrange(len(b))] = rng.randrange(256)
    open(f"{w}/d{i}.onnx", "wb").write(bytes(b))
EOF
"$W/drv" "$R"/tests/golden/*.onnx "$W"/d*.onnx
rm -rf "$W"
