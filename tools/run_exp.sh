#!/bin/bash
# Runs tools/prof_pbs.py against the shipped library and each experiment build given as arguments
# (names under tools/exp/lib_<name>.so); stops at the first failure.  usage: tools/run_exp.sh G name...
G=$1; shift
echo "lib=shipped"; timeout -k 10 120 python3 tools/prof_pbs.py $G || exit 1
for n in "$@"; do
  echo "lib=$n"; TFHE_AES_HIP_LIB=tools/exp/lib_$n.so timeout -k 10 120 python3 tools/prof_pbs.py $G || exit 1
done
