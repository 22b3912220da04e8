"""Developer tool: one csv of per-launch PMC counters and derived utilisations from tools/profile_round.sh's passes.

    python tools/pmc_report.py gpurun_out/prof_r02 > profiles/r02_pmc_kernels.csv

Per variant (list scan / accelerated x fp32 / fp64) and kernel (render, resume pass, tail) it prints the mean counter
values per launch, the kernel's duration inside the passes (counter-collection timestamps) and the derived figures:
VALU issue (SQ_INSTS_VALU / 1024 SIMDs / cycles against 0.5), lane utilisation, scalar-cache and LDS activity, HBM
bytes (FETCH_SIZE x 1024 x 2 on gfx950, WRITE_SIZE x 1024; MI355X_MICROARCH.md "HBM")."""
import collections, csv, glob, os, sys

d = sys.argv[1]
VARIANTS = [("list_f32", "rrt -b: list scan, fp32 (BASELINE configuration 3 - the headline kernel)"), ("accel_f32", "rrt: use_bvh (acceleration grid), fp32 - the CLI's default mode"),
            ("list_f64", "rrtd -b: list scan, fp64 (BASELINE configuration 4)"), ("accel_f64", "rrtd: use_bvh, fp64")]
print("# rocprofv3 --pmc passes (one counter group per pass, no tracing; tools/profile_round.sh) of")
print("#   ./rrt[d] [-b] -i scenes/final.txt -w 1200 -h 800 -s 500 -d 50 -o frame.png      (MI355X, gfx950, ROCm 7.2)")
print("# values: mean per launch of the kernel named; duration = mean of End - Start timestamps of its dispatches inside the passes")
for var, what in VARIANTS:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.Counter())
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, var, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "rrtx::" not in k:
                continue
            k = k[k.index("rrtx::") + 6:].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    for k in sorted(acc, key=lambda k: -sum(dur[k]) / len(dur[k])):
        v = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
        ms = sum(dur[k]) / len(dur[k])
        if ms < 0.3:
            continue
        print("\n# %s | rrtx::%s (%.3f ms)" % (what, k, ms))
        print("counter,value")
        for c in sorted(v):
            print("%s,%g" % (c, v[c]))
        print("# derived")
        if "GRBM_GUI_ACTIVE" not in v or "SQ_INSTS_VALU" not in v:
            continue
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        print("shader_clock_GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration),%.3f" % (cyc / (ms * 1e-3) / 1e9))
        print("valu_wave_instructions_per_cycle_per_SIMD (peak 0.5),%.4f" % (v["SQ_INSTS_VALU"] / (1024.0 * cyc)))
        print("valu_issue_utilisation,%.4f" % (v["SQ_INSTS_VALU"] / (1024.0 * cyc) / 0.5))
        if v.get("SQ_ACTIVE_INST_VALU"):
            print("valu_lane_utilisation (THREAD_CYCLES_VALU / (64 ACTIVE_INST_VALU)),%.4f" % (v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"])))
        if "SQ_INSTS_SALU" in v:
            print("salu_per_valu_instruction,%.4f" % (v["SQ_INSTS_SALU"] / v["SQ_INSTS_VALU"]))
        if "SQC_DCACHE_BUSY_CYCLES" in v:
            print("scalar_cache_busy (SQC_DCACHE_BUSY_CYCLES / (128 SQC x cycles)),%.4f" % (v["SQC_DCACHE_BUSY_CYCLES"] / (128 * cyc)))
        if "SQ_ACTIVE_INST_LDS" in v:
            print("lds_instruction_issue (SQ_ACTIVE_INST_LDS / (256 CU x cycles)),%.5f" % (v["SQ_ACTIVE_INST_LDS"] / (256 * cyc)))
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            rd, wr = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
            print("hbm_read_bytes (FETCH_SIZE KB x 1024 x 2, gfx950 correction),%.4g" % rd)
            print("hbm_write_bytes (WRITE_SIZE KB x 1024),%.4g" % wr)
            print("hbm_bandwidth_GBs,%.2f" % ((rd + wr) / (ms * 1e-3) / 1e9))
