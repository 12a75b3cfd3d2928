#!/bin/bash
# A/B builds of the working tree with extra macros: scripts/build_variant_flags.sh <name> "<flags>"  -> variants/libekf_engine_<name>.so
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/openekfmonoslam_amd" "$tmp/include"
cp -r "$root/openekfmonoslam_amd/csrc" "$tmp/openekfmonoslam_amd/"
rm -f "$tmp"/openekfmonoslam_amd/csrc/*.o
cp "$root/openekfmonoslam_amd/build.py" "$tmp/openekfmonoslam_amd/"
cp "$root"/include/*.h "$tmp/include/"
(cd "$tmp" && EKF_EXTRA_FLAGS="$2" python - <<PY
import importlib.util
spec = importlib.util.spec_from_file_location("b", "openekfmonoslam_amd/build.py")
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print(b.build_engine(force=True))
PY
)
mkdir -p "$root/variants"
cp "$tmp/openekfmonoslam_amd/libekf_engine.so" "$root/variants/libekf_engine_$1.so"
rm -rf "$tmp"
echo "variants/libekf_engine_$1.so"
