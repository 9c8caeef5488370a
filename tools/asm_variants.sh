#!/bin/bash
# Builds one experiment library per schedule variant of the hand-scheduled loop: tools/asm_variants.sh name=opt1,opt2 ...
# (PBS_ASM_OPT flags of tools/pbs_asm/gen.py; "name=" alone is the default schedule), checks each on the CPU emulator first.
# "name=pad:class:count" adds harmless instructions of one class (PBS_ASM_PAD of gen.py; results stay bit-exact).
# "name=exp:flag1,flag2" builds a timing variant of the default schedule instead (PBS_ASM_EXP flags; results are garbage).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/exp
for v in "$@"; do
  n=${v%%=*}; f=${v#*=}
  if [[ $f == pad:* ]]; then
    PBS_ASM_PAD=${f#pad:} python3 tools/pbs_asm/gen.py tools/exp/loop_$n.inc > /dev/null
    (PBS_ASM_PAD=${f#pad:} python3 tools/pbs_asm/emu.py 1 | grep -q "mismatching words 0" || { echo "variant $n fails on the emulator"; exit 1; }
     tools/build_exp.sh asm_$n "-DTFA_PBS_ASM_INC=\"../../tools/exp/loop_$n.inc\"" 2>&1 | grep -iE "error" || true) &
    continue
  fi
  if [[ $f == exp:* ]]; then
    PBS_ASM_EXP=${f#exp:} python3 tools/pbs_asm/gen.py tools/exp/loop_$n.inc > /dev/null
    (tools/build_exp.sh asm_$n "-DTFA_PBS_ASM_INC=\"../../tools/exp/loop_$n.inc\"" 2>&1 | grep -iE "error" || true) &
    continue
  fi
  PBS_ASM_OPT=$f python3 tools/pbs_asm/gen.py tools/exp/loop_$n.inc > /dev/null
  PBS_ASM_OPT=$f PBS_ASM_WAVES=4 python3 tools/pbs_asm/gen.py tools/exp/loop4_$n.inc > /dev/null      # both loops read one key
  (PBS_ASM_OPT=$f python3 tools/pbs_asm/emu.py 1 | grep -q "mismatching words 0" || { echo "variant $n fails on the emulator"; exit 1; }
   ls=1; [[ -n "$f" && ",$f," != *,lanesum,* ]] && ls=0      # the key layout and twiddle table follow the loop's "lanesum" option
   tools/build_exp.sh asm_$n "-DTFA_LANESUM=$ls -DTFA_PBS_ASM_INC=\"../../tools/exp/loop_$n.inc\" -DTFA_PBS_ASM4_INC=\"../../tools/exp/loop4_$n.inc\"" 2>&1 | grep -iE "error" || true) &
done
wait
