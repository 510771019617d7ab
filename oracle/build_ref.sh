#!/usr/bin/env bash
# build_ref.sh — TEST INFRASTRUCTURE.  Compiles the REFERENCE's own OpenCL kernels
# (/root/reference/src/cl/wavefront.cl and the files it includes, unmodified, where they lie)
# for gfx950 with the ROCm OpenCL compiler and ROCm's own OpenCL builtin library, one code
# object per kernel variant, into oracle/_ref/ (git-ignored; travels to the GPU box).
# Nothing from the reference is copied into the repository and no stand-in headers, libraries
# or generated code are involved: the compile line below is the whole recipe.
#
# Flags: the reference builds with -cl-fast-relaxed-math -cl-mad-enable (template.cpp:1254),
# which is not reproducible; the pinned build is IEEE: -O2 -ffp-contract=off
# -cl-fp32-correctly-rounded-divide-sqrt (see oracle/README.md).
# Resolution is the reference's compile-time 1280x720 (src/constants.h:3-4).
set -euo pipefail
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
CLANG=${CLANG:-/opt/rocm/lib/llvm/bin/clang}
FORCE=0; [[ "${1:-}" == "--force" ]] && FORCE=1
[[ -d "$REF/src/cl" ]] || { echo "build_ref.sh: $REF not present, nothing to do"; exit 0; }
mkdir -p "$OUT"
cd "$REF"
for shading in SHADING_NEE SHADING_SIMPLE; do
 for sampling in SAMPLING_COSINE SAMPLING_HEMISPHERE; do
  for bvh in USE_BVH2 USE_BVH4; do
   for rr in 1 0; do
    for ff in 1 0; do
      name="wf_${shading#SHADING_}_${sampling#SAMPLING_}_${bvh#USE_}_rr${rr}_ff${ff}.co"
      name=$(echo "$name" | tr 'A-Z' 'a-z')
      [[ $FORCE -eq 0 && -s "$OUT/$name" ]] && continue
      defs=(-D$shading -D$sampling -D$bvh)
      [[ $rr -eq 1 ]] && defs+=(-DRUSSIAN_ROULETTE)
      [[ $ff -eq 1 ]] && defs+=(-DFILTER_FIREFLIES)
      extra=()
      # tlas.cl:9 hard-codes BVHNode2* in instanceIntersect: a pointer-type warning-as-error under clang for USE_BVH4
      [[ $bvh == USE_BVH4 ]] && extra+=(-Wno-incompatible-pointer-types)
      "$CLANG" -x cl -cl-std=CL2.0 -target amdgcn-amd-amdhsa -mcpu=gfx950 -I "$REF" "${defs[@]}" \
          -O2 -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt -w "${extra[@]}" \
          src/cl/wavefront.cl -o "$OUT/$name"
    done
   done
  done
 done
done
# post-processing chain (src/cl/postproc.cl: prep, vignetting, gammaCorr, chromatic are buffer kernels; display / saveImage need a GL image)
if [[ $FORCE -eq 1 || ! -s "$OUT/postproc.co" ]]; then
  "$CLANG" -x cl -cl-std=CL2.0 -target amdgcn-amd-amdhsa -mcpu=gfx950 -I "$REF" \
      -O2 -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt -w src/cl/postproc.cl -o "$OUT/postproc.co"
fi
# the runner that loads these code objects (own code, oracle/ref_runner.cpp)
if [[ $FORCE -eq 1 || ! -s "$OUT/libref_runner.so" || "$HERE/ref_runner.cpp" -nt "$OUT/libref_runner.so" ]]; then
  /opt/rocm/bin/hipcc -O2 -fPIC -shared -x c++ "$HERE/ref_runner.cpp" -o "$OUT/libref_runner.so" -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -L/opt/rocm/lib -lamdhip64
fi
# the reference's asset readers: its vendored header-only libraries (lib/stb_image.h, lib/stb_image_write.h, src/tiny_obj_loader.h),
# compiled where they lie behind own glue code (oracle/ref_io_runner.cpp) - pins Scene::LoadTexture / LoadModel / SavePNG
if [[ $FORCE -eq 1 || ! -s "$OUT/libref_io.so" || "$HERE/ref_io_runner.cpp" -nt "$OUT/libref_io.so" ]]; then
  g++ -O2 -fPIC -shared -w -I "$REF/lib" -I "$REF/src" "$HERE/ref_io_runner.cpp" -o "$OUT/libref_io.so"
fi
ls "$OUT" | wc -l
