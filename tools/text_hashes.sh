#!/bin/bash
# sha256 of the .text section of every gfx950 code object of the release library, one line per source: the check that a source
# clean-up left the SHIPPED device code unchanged (compare two runs' outputs).   bash tools/text_hashes.sh [csrc dir]
CSRC=${1:-$(dirname "$0")/../zhusuan-pytorch_amd/csrc}
TMP=$(mktemp -d)
for f in "$CSRC"/*.hip; do
  b=$(basename "$f" .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function --cuda-device-only --no-gpu-bundle-output -c "$f" -o "$TMP/$b.co" 2>/dev/null || { echo "$b: build failed"; continue; }
  /opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.text "$TMP/$b.co" "$TMP/$b.text"
  printf "%-16s %8d bytes  %s\n" "$b" "$(stat -c %s "$TMP/$b.text")" "$(sha256sum "$TMP/$b.text" | cut -c1-16)"
done
rm -rf "$TMP"
