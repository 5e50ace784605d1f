#!/bin/bash
# Builds an experimental variant of the product library into build/<name>/libhypergreco.so (git-ignored; travels to the GPU box).
#   scripts/build_variant.sh strict "-DHG_STRICT_TICKETS"      then run with HG_LIB=build/strict/libhypergreco.so
set -e
name=$1; shift
flags="$*"
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/hyper-greco_amd/csrc
out=$root/build/$name
mkdir -p "$out"
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fopenmp --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -Wno-sign-compare $flags"
pids=()
for f in kernels.hip prover.hip prover_seq.hip capi.hip bn254.hip comm.hip verifier_dev.hip; do $CXX -c "$src/$f" -o "$out/${f%.hip}.o" & pids+=($!); done
for f in host.cpp verifier.cpp; do $CXX -x hip -c "$src/$f" -o "$out/${f%.cpp}.o" & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
# sanitizer flags (make asan) must reach the link too
lflags=$(echo "$flags" | tr ' ' '\n' | grep -E '^-fsanitize|^-shared-libsan|^-fno-gpu-sanitize' | tr '\n' ' ' || true)
/opt/rocm/bin/hipcc -shared -fopenmp --offload-arch=gfx950 $lflags -o "$out/libhypergreco.so" "$out"/*.o -ldl
echo "$out/libhypergreco.so"
