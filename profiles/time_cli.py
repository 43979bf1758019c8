#!/usr/bin/env python3
"""Process wall time of the CLIs beside their own account of it:  python profiles/time_cli.py [runs=5] [pause_s=0]
pause_s: seconds of sleep in front of every run (the kernel driver finishes tearing the previous process's GPU context down in the background:
a process started right behind another one waits for that inside its first HIP call)."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pause = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
for tool, seq in (("ba", "fr1xyz"), ("slam", "fr2robot2"), ("ba", "fr2robot2")):
    for r in range(runs):
        d = tempfile.mkdtemp()
        time.sleep(pause)
        t0 = time.perf_counter()
        p = subprocess.run([os.path.join(ROOT, "gbp_poplar_amd", "bin", tool), "--bal_file", os.path.join(ROOT, "data", "sequences", seq + ".txt"), "--profile", "1"],
                           env=dict(os.environ, GC_PROFILE_LOG_DIR=d), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        wall = time.perf_counter() - t0
        st = json.load(open(os.path.join(d, "gbp_profile.json")))["startup"]
        print("%s %s run %d: process wall %.3f s | to the end of main %.3f s (loader %.3f, file %.4f, runtime %.3f, create %.3f, loop %.4f, teardown %.4f) | after main %.3f s | %s"
              % (tool, seq, r, wall, st["process_s"], st["exec_to_main_s"], st["file_parse_s"], st["runtime_init_s"], st["create_s"], st["loop_s"], st["teardown_s"],
                 wall - st["process_s"], st["create"][6:]), flush=True)
