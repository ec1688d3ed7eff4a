#!/bin/bash
# Alternative builds of libthinktwice_hip.so whose hand-pipelined h2 conv kernel (csrc/conv_h2.hip, conv_h2_pipe_kernel) has one part
# of its K loop removed (timing ablations; results are wrong by design).  tools/build_h2_debug.sh 1 2 4 ...  ->  tools/_dbg/libtt_h2_<N>.so
set -e
cd "$(dirname "$0")/.."
python -m thinktwice_amd.build > /dev/null
mkdir -p tools/_dbg
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wno-unused-result"
for n in "$@"; do
  /opt/rocm/bin/hipcc $F -DTT_H2_DEBUG=$n -x hip -c thinktwice_amd/csrc/conv_h2.hip -o tools/_dbg/conv_h2.dbg$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_dbg/libtt_h2_$n.so \
      $(ls thinktwice_amd/csrc/_obj/*.o | grep -v conv_h2) tools/_dbg/conv_h2.dbg$n.o
  echo tools/_dbg/libtt_h2_$n.so
done
