#!/bin/bash
# Build the library of another git revision next to the working tree's, for same-box A/B runs (tools/ab_lib.py):
#   tools/build_baseline.sh <rev> [name]   ->  tools/bin/librerank_<name>.so   (tools/bin is git-ignored, shipped by gpurun)
set -e
rev=${1:-HEAD}; name=${2:-base}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" reranking-multimodal-retrievers_amd include | tar -x -C "$tmp"
python3 "$tmp/reranking-multimodal-retrievers_amd/build.py" > /dev/null
mkdir -p "$root/tools/bin"
cp "$tmp/reranking-multimodal-retrievers_amd/librerank_mi355.so" "$root/tools/bin/librerank_$name.so"
rm -rf "$tmp"
echo "$root/tools/bin/librerank_$name.so"
