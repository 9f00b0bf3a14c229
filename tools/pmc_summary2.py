"""profiles/<name>_pmc_summary.json from the small per-kernel CSVs tools/pmc_run.sh copies back (gpurun_out/pmc_<tag>/{SQ_WAVE_CYCLES,FETCH_SIZE,WRITE_SIZE}.csv):
    python tools/pmc_summary2.py gpurun_out/pmc_r02b profiles/r02_pmc_summary.json [previous summary to take the physics-only probe block from]
The physics-only probe (tools/physics_rate.py under an SQ pass: instructions per physics.step()) is carried over from the previous summary when the
macro-step kernel has not changed since; the note says which."""
import csv, json, sys, os


def main(src, out_path, prev_path=None):
    k = {}
    for r in csv.DictReader(open(os.path.join(src, "SQ_WAVE_CYCLES.csv"))):
        d = k.setdefault(r["kernel"], {}); n = int(r["dispatches"]); d[r["counter"]] = float(r["sum"]) / n; d["launches"] = n
    for nm in ("FETCH_SIZE", "WRITE_SIZE"):
        for r in csv.DictReader(open(os.path.join(src, nm + ".csv"))):
            if r["kernel"] in k:
                k[r["kernel"]][nm + "_KiB_per_launch"] = float(r["sum"]) / int(r["dispatches"])
    for d in k.values():
        wc = d["SQ_WAVE_CYCLES"]
        d.update(active_inst_any_frac=d["SQ_ACTIVE_INST_ANY"] / wc, active_inst_valu_frac=d["SQ_ACTIVE_INST_VALU"] / wc, wait_any_frac=d["SQ_WAIT_ANY"] / wc,
                 wait_inst_any_frac=d["SQ_WAIT_INST_ANY"] / wc)
        if "FETCH_SIZE_KiB_per_launch" in d and "WRITE_SIZE_KiB_per_launch" in d:
            d["hbm_bytes_per_launch_raw"] = (d["FETCH_SIZE_KiB_per_launch"] + d["WRITE_SIZE_KiB_per_launch"]) * 1024
            d["hbm_bytes_per_launch_fetch_doubled"] = (2 * d["FETCH_SIZE_KiB_per_launch"] + d["WRITE_SIZE_KiB_per_launch"]) * 1024
    out = {"source": "rocprofv3 --pmc, three separate passes (8 SQ counters | FETCH_SIZE | WRITE_SIZE; tools/pmc_run.sh), --kernel-include-regex "
                     "'k_macro_step|k_observe|k_conv1_u8', command: python3 bench.py --no-cpu-baseline --steps 40 --warmup 20 (bench defaults otherwise: acorn_env, "
                     "4096 envs, f32 state, time-sliced schedule, synthetic U(-1,1) action stream, 800 pre-roll steps); summarised by tools/pmc_summary2.py",
           "config": {"object": "acorn", "envs": 4096, "state_dtype": "f32", "mixed": False},
           "notes": ["SQ_* counters in quad-cycles (SQ_WAVE_CYCLES, SQ_ACTIVE_*, SQ_WAIT_*) or instructions (SQ_INSTS_*), summed over all waves, averaged per launch",
                     "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) "
                     "streaming reads and is uncalibrated for other widths -- these kernels read 4 B per lane, so both the raw and the doubled figure are given"],
           "kernels": k}
    sha = os.path.join(src, "csrc_sha16.txt")
    if os.path.exists(sha):                    # the tree the counters were collected on (engine.source_fingerprint(), written by tools/pmc_run.sh on the GPU box)
        out["csrc_sha16"] = open(sha).read().strip()
        psha = os.path.join(src, "probe_csrc_sha16.txt")
        if os.path.exists(psha) and open(psha).read().strip() != out["csrc_sha16"]:
            raise SystemExit("the physics-only probe and the bench passes were taken on different sources: " + open(psha).read().strip() + " / " + out["csrc_sha16"])
    probe_csv, probe_json = os.path.join(src, "probe_SQ.csv"), os.path.join(src, "probe.json")
    if os.path.exists(probe_csv) and os.path.exists(probe_json):          # tools/pmc_probe.sh: the physics-only probe of THIS kernel, measured, not carried over
        c = {r["counter"]: (float(r["sum"]), int(r["dispatches"])) for r in csv.DictReader(open(probe_csv))}
        pj = json.loads(open(probe_json).read().strip().splitlines()[-1])
        subs = float(pj["all_substeps_of_finished_macro_steps"]); n = c["SQ_INSTS_VALU"][1]
        k["k_macro_step"]["physics_only_probe"] = {
            "command": "python3 tools/physics_rate.py - acorn 144 3000 1500 2048 under rocprofv3 --pmc (SQ pass; tools/pmc_probe.sh)", "launches": n, "env_substeps": subs,
            "substeps_per_s_under_the_profiler": pj["substeps_per_s"],
            "valu_wave_instructions_per_env_substep": c["SQ_INSTS_VALU"][0] / subs, "wave_cycles_per_env_substep": 4 * c["SQ_WAVE_CYCLES"][0] / subs,
            "active_inst_valu_frac": c["SQ_ACTIVE_INST_VALU"][0] / c["SQ_WAVE_CYCLES"][0], "wait_any_frac": c["SQ_WAIT_ANY"][0] / c["SQ_WAVE_CYCLES"][0],
            "salu_per_valu": c["SQ_INSTS_SALU"][0] / c["SQ_INSTS_VALU"][0], "lds_per_valu": c["SQ_INSTS_LDS"][0] / c["SQ_INSTS_VALU"][0]}
        k["k_macro_step"]["valu_lane_ops_per_env_substep"] = 64.0 * c["SQ_INSTS_VALU"][0] / subs
        out["notes"].append("k_macro_step.physics_only_probe: measured on this kernel (tools/pmc_probe.sh); env_substeps = physics.step() calls of the macro steps that finished "
                            "during the run (all ticks, pre-roll included, as the counters are)")
        prev_path = None
    if prev_path:
        prev = json.load(open(prev_path))["kernels"]["k_macro_step"]
        for key in ("physics_only_probe", "valu_lane_ops_per_env_substep"):
            if key in prev:
                k["k_macro_step"][key] = prev[key]
        out["notes"].append("k_macro_step.physics_only_probe / valu_lane_ops_per_env_substep: carried over from the previous summary (same kernel source; "
                            "tools/physics_rate.py under an SQ pass)")
    json.dump(out, open(out_path, "w"), indent=1)
    for name, d in k.items():
        print(name, {x: round(d[x], 3) for x in ("active_inst_valu_frac", "wait_any_frac")}, "VALU/launch", round(d["SQ_INSTS_VALU"]), "launches", d["launches"])


if __name__ == "__main__":
    main(*sys.argv[1:4])
