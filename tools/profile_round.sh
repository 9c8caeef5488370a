#!/bin/bash
# Collects the round's profiles on the GPU box: kernel-trace stats of the bench, then one rocprofv3 --pmc pass per
# counter group over tools/prof_pbs.py.  Every pass runs on its own, under its own time limit; the first failure stops.
# usage: tools/profile_round.sh <tag>   -> gpurun_out/prof_<tag>/...
set -e
tag=${1:-r2}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
echo "[kt noaes]"; timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/kt_noaes --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aes > $out/kt_noaes.log 2>&1
echo "[kt full]";  timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $out/kt_full --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/kt_full.log 2>&1
echo "[kt aes block]"; timeout -s KILL 200 rocprofv3 --kernel-trace -d $out/kt_aes --output-format csv -- python3 tools/aes_trace.py > $out/kt_aes.log 2>&1
pass() {  # name bootstraps counters...
  local name=$1 n=$2; shift 2
  echo "[pmc $name]"; timeout -s KILL 120 rocprofv3 --pmc "$@" -d $out/pmc_$name --output-format csv -- python3 tools/prof_pbs.py $n > $out/pmc_$name.log 2>&1
}
pass fetch 65536 FETCH_SIZE
pass write 65536 WRITE_SIZE
pass sq_a 8192 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU
pass sq_b 8192 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
echo done
