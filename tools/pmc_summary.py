"""Reduces the rocprofv3 --pmc passes of tools/profile_round.sh to mean counter values per dispatch and kernel.
usage: python3 tools/pmc_summary.py gpurun_out/prof_<tag> [commit] > summary.json   (commit: the revision the profiled library was
built from; default: HEAD of the repository this script sits in)"""
import csv, glob, json, os, subprocess, sys
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
                if ("pbs_" not in k and "keyswitch" not in k) or "pbs_asm_kernel<1" in k:   # <1, ...>: the calibration's launches
                    continue
                acc.setdefault((k, row["Counter_Name"]), []).append((int(row["Grid_Size"]), float(row["Counter_Value"])))
        for (k, c), v in acc.items():
            # the workload's launches are the widest ones of a kernel; the context's round-time calibration also runs
            # every bootstrap mapping a few times on one round of workgroups
            widest = max(g for g, _ in v)
            vals = [x for g, x in v if g == widest]
            out.setdefault(k, {})[f"{c}@{n}"] = sum(vals) / len(vals)


def derived(c, waves_per_bootstrap, steps=805, n=8192, ratios_only=False):
    """ratios the text quotes; SQ_* wave counters share one unit, so their ratios need no conversion"""
    wc = c["SQ_WAVE_CYCLES@%d" % n]
    d = {k + "/SQ_WAVE_CYCLES": round(c["%s@%d" % (k, n)] / wc, 4)
         for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS")}
    d["lds_bank_conflict_cycles/lds_active_cycles"] = round(c["SQ_LDS_BANK_CONFLICT@%d" % n] / c["SQ_LDS_IDX_ACTIVE@%d" % n], 4)
    if ratios_only:
        return d
    per = n * waves_per_bootstrap * steps
    d["valu_instructions_per_wave_step"] = round(c["SQ_INSTS_VALU@%d" % n] / per, 1)
    d["lds_instructions_per_wave_step"] = round(c["SQ_INSTS_LDS@%d" % n] / per, 1)
    if "FETCH_SIZE@65536" in c:
        d["hbm_bytes_per_65536_launch"] = int((2 * c["FETCH_SIZE@65536"] + c["WRITE_SIZE@65536"]) * 1024)
    return d


def fp64_mix(lines):
    ops = [ln.split()[0] for ln in lines if not ln.endswith(":")]
    n = {k: sum(1 for o in ops if o.startswith(k)) for k in ("v_fma_f64", "v_add_f64", "v_mul_f64")}
    n["fma_share"] = round(n["v_fma_f64"] / sum(n.values()), 3)
    n["source"] = "instruction counts of the generated loop (tools/pbs_asm)"
    return n


res = {"note": "rocprofv3 --pmc, separate passes (tools/profile_round.sh), reduced by tools/pmc_summary.py: mean per dispatch over the "
               "workload's launches (the round-time calibration's launches are left out). FETCH_SIZE/WRITE_SIZE in KiB as reported "
               "(gfx950: double FETCH_SIZE for 16-byte-per-lane reads). @N = bootstraps per dispatch (tools/prof_pbs.py N). "
               "SQ_* wave counters in quad-cycles. The one-bootstrap-per-workgroup kernels (pbs_lat_*) are launched here only by the "
               "round-time calibration (one round of workgroups, a quarter and three quarters of the steps), so only their ratios are meaningful.",
       "counters_current_kernels": out}
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "pbs_asm"))
for k, c in out.items():
    try:
        if "pbs_asm_kernel<0, 8>" in k or "pbs_asm_kernel<0>" in k:
            import gen
            res["derived_pbs_asm_kernel"] = derived(c, 1)
            res["derived_pbs_asm_kernel"]["fp64_instructions_per_wave_step"] = fp64_mix(gen.Loop().build())
        elif "pbs_lat_asm_kernel<1>" in k:
            import lat_gen
            res["derived_pbs_lat_asm_kernel_1"] = derived(c, 4, ratios_only=True)
            res["derived_pbs_lat_asm_kernel_1"]["fp64_instructions_per_wave_step"] = fp64_mix(lat_gen.Lat().build())
        elif "pbs_lat_asm_kernel<2>" in k:
            res["derived_pbs_lat_asm_kernel_2"] = derived(c, 4, ratios_only=True)
    except KeyError:
        pass
try:
    res["commit"] = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(
        ["git", "-C", os.path.dirname(os.path.abspath(__file__)), "rev-parse", "HEAD"], text=True).strip()
except (OSError, subprocess.CalledProcessError):
    res["commit"] = None
res["kernel"] = next((k for k in out if "pbs_asm_kernel<0, 8>" in k or "pbs_asm_kernel<0>" in k), None)
print(json.dumps(res, indent=1, sort_keys=True))
