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
                if ("pbs_" not in k and "keyswitch" not in k) or "pbs_asm_kernel<1>" in k:   # <1>: the calibration's launches
                    continue
                acc.setdefault((k, row["Counter_Name"]), []).append((int(row["Grid_Size"]), float(row["Counter_Value"])))
        for (k, c), v in acc.items():
            # the workload's launches are the widest ones of a kernel; the context's round-time calibration also runs
            # every bootstrap mapping a few times on one round of workgroups
            widest = max(g for g, _ in v)
            vals = [x for g, x in v if g == widest]
            out.setdefault(k, {})[f"{c}@{n}"] = sum(vals) / len(vals)
print(json.dumps(out, indent=1, sort_keys=True))
