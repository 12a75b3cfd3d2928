#!/bin/bash
# Builds libekf_engine.so of another commit into variants/libekf_engine_<tag>.so (A/B timing on the GPU box:
# EKF_ENGINE_LIB=variants/libekf_engine_<tag>.so python bench.py ...).  usage: scripts/build_variant.sh <tag> <commit>
set -e
tag=$1; commit=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$commit" openekfmonoslam_amd/csrc openekfmonoslam_amd/build.py include | tar -x -C "$tmp"
(cd "$tmp" && python - <<PY
import importlib.util, sys
spec = importlib.util.spec_from_file_location("b", "openekfmonoslam_amd/build.py")
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print(b.build_engine(force=True))
PY
)
mkdir -p "$root/variants"
cp "$tmp/openekfmonoslam_amd/libekf_engine.so" "$root/variants/libekf_engine_$tag.so"
rm -rf "$tmp"
echo "variants/libekf_engine_$tag.so"
