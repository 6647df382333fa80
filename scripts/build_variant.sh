# build a variant of the library for A/B runs: bash scripts/build_variant.sh NAME [-DFLAG=..]...  ->  simka_amd/lib/libsimka_NAME.so
# (then: LIBS="NAME=simka_amd/lib/libsimka_NAME.so ..." bash scripts/ab.sh on the GPU box)
set -e
name=$1; shift
cd "$(dirname "$0")/../simka_amd/csrc"

hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -Wno-unused-result "$@" -o ../lib/libsimka_$name.so simka_ctx.hip simka_wide.hip simka_host.cpp -lz -ldl 2>&1 | grep -E "error|Error" || true
ls -la ../lib/libsimka_$name.so
