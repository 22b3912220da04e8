"""Developer script (GPU box): wall time of whole `rrt` processes - what a Makefile that starts one process per frame (the
reference's scenes/final_anim/Makefile) pays besides the render - with the phases the process reports under RRTX_TIMING=1."""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
def run(args, n=3, env=None):
    for _ in range(n):
        t = time.perf_counter()
        r = subprocess.run([os.path.join(ROOT, "rrt")] + args, capture_output=True, env=env)
        dt = time.perf_counter() - t
        took = re.search(rb"took ([0-9.]+) seconds", r.stderr)
        print("%.3f s wall, render %s s: rrt %s" % (dt, took.group(1).decode() if took else "?", " ".join(args)), flush=True)
        extra = [l for l in r.stderr.decode().splitlines() if l.startswith("timing,")]
        if extra: print("   " + extra[-1])
final, test1 = os.path.join(ROOT, "scenes", "final.txt"), os.path.join(ROOT, "scenes", "test1.txt")
run(["-i", final, "-o", os.path.join(out, "x.png"), "-w", "1280", "-h", "720", "-s", "50"])
run(["-i", test1, "-o", os.path.join(out, "y.png"), "-w", "400", "-h", "266", "-s", "4"])
run(["-b", "-i", test1, "-o", os.path.join(out, "y.png"), "-w", "400", "-h", "266", "-s", "4"], 2)
run(["-i", test1, "-o", os.path.join(out, "y.png"), "-w", "400", "-h", "266", "-s", "4"], 2, dict(os.environ, HIP_VISIBLE_DEVICES="0"))
run(["-h"], 1)
