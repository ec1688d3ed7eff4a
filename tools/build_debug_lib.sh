#!/bin/bash
# Build an alternative libthinktwice_hip.so whose LDS-DMA conv kernel honours the timing-experiment activation codes
# (TT_MB_ACT=97: skip the weight DMA after the first K tile, 98: skip the activation DMA; results are wrong by design).
#   tools/build_debug_lib.sh && TT_LIB_PATH=$PWD/tools/_dbg/libthinktwice_hip.so TT_MB_ACT=97 python tools/conv_microbench.py ...
set -e
cd "$(dirname "$0")/.."
python -m thinktwice_amd.build > /dev/null
mkdir -p tools/_dbg
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wno-unused-result"
/opt/rocm/bin/hipcc $F -DTT_GLDS_DEBUG=1 -x hip -c thinktwice_amd/csrc/conv_igemm_glds.hip -o tools/_dbg/conv_igemm_glds.dbg.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_dbg/libthinktwice_hip.so \
    $(ls thinktwice_amd/csrc/_obj/*.o | grep -v conv_igemm_glds) tools/_dbg/conv_igemm_glds.dbg.o
echo tools/_dbg/libthinktwice_hip.so
