#!/bin/bash
# Sanitizer + mutation run of the host-side file code (CPU only; GPU sanitizers are not available on the pool).
#   bash tools/fuzz/run.sh [mutations per file, default 200]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=$(mktemp -d /tmp/ptfuzz.XXXXXX)
mkdir -p $W/img
python3 - "$ROOT" "$W/img" <<'PY'
import sys, os, numpy as np
root, out = sys.argv[1], sys.argv[2]
for npz in ("png_textures.npz", "jpeg_textures.npz"):
    g = np.load(os.path.join(root, "tests", "golden", npz))
    for k in g.files:
        if k.startswith("file_"):
            open(os.path.join(out, npz[:3] + "_" + k[5:]), "wb").write(g[k].tobytes())
PY
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -o $W/fuzz_loader $ROOT/tools/fuzz/fuzz_loader.cpp
cd $ROOT/scenes && ASAN_OPTIONS=detect_leaks=1 timeout 1500 $W/fuzz_loader $W/img $ROOT/scenes ${1:-200} $W
rm -rf $W
