"""profiles/r01_pmc_summary.json from three rocprofv3 --pmc passes (SQ counters | FETCH_SIZE | WRITE_SIZE) of bench.py:
    python tools/pmc_summary.py gpurun_out/pmc_r1c_ profiles/r01_pmc_summary.json"""
import collections, csv, glob, json, sys


def load(pat):
    return list(csv.DictReader(open(glob.glob(pat)[0])))


def main(prefix, out_path):
    out = {"source": "rocprofv3 --pmc, three separate passes (SQ counters | FETCH_SIZE | WRITE_SIZE), --kernel-include-regex 'k_macro_step|k_observe|k_conv1_u8', "
                     "command: python3 bench.py --no-cpu-baseline --steps 40 --warmup 20 (time-sliced schedule, bench defaults otherwise)",
           "notes": ["SQ_* counters in quad-cycles summed over all waves, averaged per launch",
                     "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of wide "
                     "(16 B/lane) streaming reads and is uncalibrated for other widths -- these kernels read 4 B per lane, so both the raw and the "
                     "doubled figure are given"], "kernels": {}}
    agg = collections.defaultdict(collections.Counter); disp = collections.defaultdict(set)
    for r in load(prefix + "SQ_WAVE_CYCLES/*/*counter_collection.csv"):
        k = r["Kernel_Name"].split("(")[0]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k, c in agg.items():
        n = len(disp[k]); d = {name: v / n for name, v in c.items()}; wc = d["SQ_WAVE_CYCLES"]
        d.update(launches=n, active_inst_any_frac=d["SQ_ACTIVE_INST_ANY"] / wc, active_inst_valu_frac=d["SQ_ACTIVE_INST_VALU"] / wc,
                 wait_any_frac=d["SQ_WAIT_ANY"] / wc, wait_inst_any_frac=d["SQ_WAIT_INST_ANY"] / wc)
        out["kernels"][k] = d
    for nm in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.Counter(); disp = collections.defaultdict(set)
        for r in load(prefix + nm + "/*/*counter_collection.csv"):
            k = r["Kernel_Name"].split("(")[0]; agg[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            out["kernels"][k][nm + "_KiB_per_launch"] = v / len(disp[k])
    for d in out["kernels"].values():
        d["hbm_bytes_per_launch_raw"] = (d["FETCH_SIZE_KiB_per_launch"] + d["WRITE_SIZE_KiB_per_launch"]) * 1024
        d["hbm_bytes_per_launch_fetch_doubled"] = (2 * d["FETCH_SIZE_KiB_per_launch"] + d["WRITE_SIZE_KiB_per_launch"]) * 1024
    json.dump(out, open(out_path, "w"), indent=1)
    for k, d in out["kernels"].items():
        print(k, {x: round(d[x], 3) for x in ("active_inst_valu_frac", "wait_any_frac", "wait_inst_any_frac")},
              round(d["FETCH_SIZE_KiB_per_launch"]), round(d["WRITE_SIZE_KiB_per_launch"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
