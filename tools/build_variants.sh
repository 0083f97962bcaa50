#!/bin/bash
# Builds experiment variants of libwmhip.so into tools/bin/ (git-ignored, travels with gpurun).
#   tools/build_variants.sh name1:"-DFLAG..." name2:"..."      (name "base" = no flags)
set -e
cd "$(dirname "$0")/.."
C=digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd/csrc
mkdir -p tools/bin
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  ( /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -shared -fPIC $flags -o tools/bin/libwmhip_$name.so $C/wmhip.hip $C/wm_ref.hip $C/wm_pixel.hip $C/wm_route.hip && echo "built $name [$flags]" ) &
done
wait
