#!/bin/bash
# Builds the library of a git revision (default HEAD) into ab/libvkrhip_<name>.so,
# for tools/ab_bench.sh; the working tree is left alone.
#   tools/build_variant.sh <name> [rev]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; REV=${2:-HEAD}
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" vulkan_renderer_amd/csrc include | tar -x -C "$TMP"
mkdir -p "$ROOT/ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt \
  -fPIC -shared -std=c++17 -o "$ROOT/ab/libvkrhip_$NAME.so" \
  "$TMP/vulkan_renderer_amd/csrc/vkrhip.hip" "$TMP"/vulkan_renderer_amd/csrc/*.cpp
rm -rf "$TMP"
echo "$ROOT/ab/libvkrhip_$NAME.so"
