#!/bin/bash
# Timing variants of the four-wave body of the wide loop: tools/asm4_variants.sh name=flag1,flag2 ...
# (PBS_ASM_EXP flags of tools/pbs_asm/gen.py; results of these builds are garbage, they are for timing only;
# "name=" alone rebuilds the shipped loop, "name=opt:a,b" sets PBS_ASM_OPT instead and stays bit-exact)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/exp
for v in "$@"; do
  n=${v%%=*}; f=${v#*=}
  if [[ $f == opt:* ]]; then PBS_ASM_WAVES=4 PBS_ASM_OPT=${f#opt:} python3 tools/pbs_asm/gen.py tools/exp/loop4_$n.inc > /dev/null
  else PBS_ASM_WAVES=4 PBS_ASM_EXP=$f python3 tools/pbs_asm/gen.py tools/exp/loop4_$n.inc > /dev/null; fi
  (tools/build_exp.sh asm4_$n "-DTFA_PBS_ASM4_INC=\"../../tools/exp/loop4_$n.inc\"" 2>&1 | grep -iE "error" || true) &
done
wait
