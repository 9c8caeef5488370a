"""Reduces the rocprofv3 --pmc passes of tools/profile_round.sh to mean counter values per dispatch and kernel.
usage: python3 tools/pmc_summary.py gpurun_out/prof_<tag> > summary.json"""
import csv, glob, json, os, sys
root = sys.argv[1]
bootstraps = {"fetch": 65536, "write": 65536, "sq_a": 8192, "sq_b": 8192}
out = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    n = bootstraps.get(os.path.basename(d)[4:], 0)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].split("(")[0]
                if "pbs_kernel" not in k and "keyswitch" not in k:
                    continue
                acc.setdefault((k, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
        for (k, c), v in acc.items():
            out.setdefault(k, {})[f"{c}@{n}"] = sum(v) / len(v)
print(json.dumps(out, indent=1, sort_keys=True))
