#!/usr/bin/env bash
# Build compile-time variants of the library into svtyper_amd/csrc/variants/ (git-ignored, shipped by gpurun):
#   tools/stream_variants.sh name1:"-DFOO=1 -DBAR=2" name2:"..."
set -eu
cd "$(dirname "$0")/../svtyper_amd/csrc"
mkdir -p variants
for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    make -s -B OUT=variants/lib_$name.so EXTRA="$flags" libsvtyper_hip.so &
done
wait
ls -la variants/
