#!/bin/bash
# Alternative builds of libthinktwice_hip.so whose hand-pipelined bf16x3 conv kernel (csrc/conv_x3_pipe.hip) has one part of its
# K loop removed (timing ablations; results are wrong by design).  tools/build_pipe_debug.sh 1 2 4 ...  ->  tools/_dbg/libtt_pipe<N>.so
set -e
cd "$(dirname "$0")/.."
python -m thinktwice_amd.build > /dev/null
mkdir -p tools/_dbg
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wno-unused-result -fno-slp-vectorize"
for n in "$@"; do
  /opt/rocm/bin/hipcc $F -DTT_PIPE_DEBUG=$n -x hip -c thinktwice_amd/csrc/conv_x3_pipe.hip -o tools/_dbg/conv_x3_pipe.dbg$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_dbg/libtt_pipe$n.so \
      $(ls thinktwice_amd/csrc/_obj/*.o | grep -v conv_x3_pipe) tools/_dbg/conv_x3_pipe.dbg$n.o
  echo tools/_dbg/libtt_pipe$n.so
done
