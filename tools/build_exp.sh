#!/bin/bash
# Builds experiment variants of the engine: tools/build_exp.sh name "-DFLAG ..." -> tools/exp/lib_name.so
set -e
cd "$(dirname "$0")/../tfhe-aes_amd"
mkdir -p ../tools/exp
make -s csrc/client.o host/boolean_tree.o host/sbox.o host/tfhe_boolean.o host/primitive.o host/state.o host/modes.o host/capi.o
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value $2 -c csrc/engine.hip -o ../tools/exp/engine_$1.o
hipcc --offload-arch=gfx950 -shared -o ../tools/exp/lib_$1.so ../tools/exp/engine_$1.o csrc/client.o host/*.o -lgomp
rm ../tools/exp/engine_$1.o
