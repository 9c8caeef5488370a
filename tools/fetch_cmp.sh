set -e
export TMPDIR=/tmp
out=gpurun_out/r3/fetch_cmp; mkdir -p $out
for i in 1 2; do
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/cur_$i --output-format csv -- python3 tools/prof_pbs.py 65536 > $out/cur_$i.log 2>&1
  TFHE_AES_HIP_LIB=$PWD/tools/exp/lib_nodo.so timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/nodo_$i --output-format csv -- python3 tools/prof_pbs.py 65536 > $out/nodo_$i.log 2>&1
done
python3 - <<'PY'
import csv,glob
for d in sorted(glob.glob("gpurun_out/r3/fetch_cmp/*_?")):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "pbs_asm_kernel<0, 8>" in r["Kernel_Name"] and int(r["Grid_Size"])>=4000000]
        print(d, [round(x*2*1024/1e9,1) for x in v], "GB")
PY
