"""Developer tool: profiles/r01_pmc_render_kernel.csv from tools/profile_round.sh's output directory.

  python tools/pmc_to_csv.py gpurun_out/prof_final > profiles/r01_pmc_render_kernel.csv
Reads pmc_summary.txt (per-launch counter means per kernel) and the kernel-trace stats csv (durations) and
adds the derived utilisations; also prints the HBM traffic of the headline kernel as JSON on stderr."""
import csv, glob, json, sys
d = sys.argv[1]
KERNELS = [("render_kernel<float, true, 1, false, 0, false>", "list scan (conservative filter, hybrid scalar/LDS operands, camera-ray lists) - the headline kernel"),
           ("render_kernel<float, true, 0, false, 2, false>", "use_bvh: accelerated closest hit (uniform grid + always-list in LDS, sliced walk)"),
           ("render_kernel<float, true, 0, false, 2, true>", "use_bvh: resume pass over the work units parked at the end of the queue"),
           ("tail_kernel", "tail kernel of the list-scan launch (8 lanes per ray)")]
dur = {}
for f in glob.glob(d + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = float(r["AverageNs"]) * 1e-6
vals, cur = {}, None
for line in open(d + "/pmc_summary.txt"):
    if line.startswith("== "):
        cur = line[3:].strip(); vals[cur] = {}
    elif cur and line.strip():
        p = line.split(); vals[cur][p[1]] = float(p[2])
print("# rocprofv3 --pmc passes (one counter group per pass, no tracing; tools/profile_round.sh) of")
print("#   python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline      (final.txt 1200x800 spp=500 d=50 fp32)")
print("# values: per launch, MI355X, end of round 1.  Kernel durations: rocprofv3 kernel-trace averages, profiles/r01_bench_kernel_stats.csv")
for k, what in KERNELS:
    v = vals.get(k)
    if not v: continue
    ms = next((t for n, t in dur.items() if k in n), None)
    print("\n# rrtx::%s : %s (%.3f ms)" % (k, what, ms))
    print("counter,value")
    for c in sorted(v): print("%s,%g" % (c, v[c]))
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0  # per XCD
    kcyc = ms * 1e-3 * 2.4e9 if ms else cyc  # cycles of the kernel itself at the nominal clock (under --pmc GUI_ACTIVE includes the serialisation)
    print("# derived")
    if ms > 5: print("shader_clock_GHz (GRBM_GUI_ACTIVE/8 XCDs/duration),%.3f" % (cyc / (ms * 1e-3) / 1e9))
    c = cyc if ms > 5 else kcyc
    print("valu_wave_instructions_per_cycle_per_SIMD (peak 0.5),%.4f" % (v["SQ_ACTIVE_INST_VALU"] / 4 / (1024 * c) * 4 / 4 if False else v["SQ_INSTS_VALU"] / (1024.0 * c)))
    print("valu_issue_utilisation,%.4f" % (v["SQ_INSTS_VALU"] / (1024.0 * c) / 0.5))
    print("valu_lane_utilisation (THREAD_CYCLES_VALU/(64*ACTIVE_INST_VALU)),%.4f" % (v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"])))
    print("salu_per_valu_instruction,%.4f" % (v["SQ_INSTS_SALU"] / v["SQ_INSTS_VALU"]))
    print("scalar_cache_busy (SQC_DCACHE_BUSY_CYCLES/(128 SQC*cycles)),%.4f" % (v["SQC_DCACHE_BUSY_CYCLES"] / (128 * c)))
    print("lds_instruction_issue (SQ_ACTIVE_INST_LDS/(256 CU*cycles)),%.5f" % (v["SQ_ACTIVE_INST_LDS"] / (256 * c)))
    rd, wr = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
    print("hbm_read_bytes (FETCH_SIZE KB x1024 x2 gfx950 correction),%.4g" % rd)
    print("hbm_write_bytes (WRITE_SIZE KB x1024),%.4g" % wr)
    print("hbm_bandwidth_GBs,%.2f" % ((rd + wr) / (ms * 1e-3) / 1e9))
    if k == KERNELS[0][0]:
        json.dump({"hbm_bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr),
                   "source": "profiles/r01_pmc_render_kernel.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled per the gfx950 note)",
                   "workload": "scenes/final.txt 1200x800 spp=500 d=50 fp32"}, sys.stderr, indent=1)
