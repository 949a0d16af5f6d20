#!/bin/bash
# Timing experiments: build variants of the library with different -D flags into tools/_ablate/.
# usage: tools/ablate_decode.sh "name1:-DFOO=1" "name2:-DBAR=2 -DBAZ" ...
set -e
mkdir -p tools/_ablate
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared $flags -x hip libhuffman_amd/csrc/hufgpu_api.hip -x hip libhuffman_amd/csrc/huf_host.cpp -x hip libhuffman_amd/csrc/hufgpu_sharded.hip -o tools/_ablate/lib_$name.so -lpthread -ldl
done
