#!/bin/bash
# Builds one experiment library per timing variant of the one-bootstrap-per-workgroup loop:
# tools/lat_variants.sh name=flag1,flag2 ...   (PBS_ASM_EXP flags of tools/pbs_asm/lat_gen.py: nokey nobar nowait noho nospec)
# PER_CU=2 tools/lat_variants.sh ...  does the same for the loop of two workgroups per CU (libraries lib_lat2_<name>.so).
# The results of these builds are garbage by construction; they are for timing only.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/exp
per_cu=${PER_CU:-1}
for v in "$@"; do
  n=${v%%=*}; f=${v#*=}
  if [ "$per_cu" = 2 ]; then
    PBS_LAT_PER_CU=2 PBS_ASM_EXP=$f python3 tools/pbs_asm/lat_gen.py tools/exp/lat2_loop_$n.inc > /dev/null
    (tools/build_exp.sh lat2_$n "-DTFA_LAT_ASM2_INC=\"../../tools/exp/lat2_loop_$n.inc\"" 2>&1 | grep -iE "error" || true) &
  else
    PBS_ASM_EXP=$f python3 tools/pbs_asm/lat_gen.py tools/exp/lat_loop_$n.inc > /dev/null
    (tools/build_exp.sh lat_$n "-DTFA_LAT_ASM_INC=\"../../tools/exp/lat_loop_$n.inc\"" 2>&1 | grep -iE "error" || true) &
  fi
done
wait
