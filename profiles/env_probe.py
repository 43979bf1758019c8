"""Does an environment switch of the HIP / ROCr runtime shorten `bin/ba fr1xyz`?  python profiles/env_probe.py   (0.5 s idle in front of every run, 3 runs each)"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cases = [("default", {}), ("HSA_ENABLE_SDMA=0", {"HSA_ENABLE_SDMA": "0"}), ("GPU_MAX_HW_QUEUES=1", {"GPU_MAX_HW_QUEUES": "1"}), ("HSA_ENABLE_INTERRUPT=0", {"HSA_ENABLE_INTERRUPT": "0"}),
         ("HIP_INITIAL_DM_SIZE=0", {"HIP_INITIAL_DM_SIZE": "0"}), ("HSA_SCRATCH_MEM=0 (unset scratch)", {"HSA_NO_SCRATCH_RECLAIM": "1"}), ("default", {})]
for label, extra in cases:
    for r in range(3):
        d = tempfile.mkdtemp()
        time.sleep(0.5)
        t0 = time.perf_counter()
        p = subprocess.run([os.path.join(ROOT, "gbp_poplar_amd", "bin", "ba"), "--bal_file", os.path.join(ROOT, "data", "sequences", "fr1xyz.txt"), "--profile", "1"],
                           env=dict(os.environ, GC_PROFILE_LOG_DIR=d, **extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        wall = time.perf_counter() - t0
        if p.returncode != 0:
            print("%-34s FAILED rc %d: %s" % (label, p.returncode, p.stderr[-200:].replace("\n", " ")), flush=True)
            continue
        st = json.load(open(os.path.join(d, "gbp_profile.json")))["startup"]
        print("%-34s wall %.3f | runtime %.3f create %.3f loop %.4f teardown %.4f | after main %.3f" % (label, wall, st["runtime_init_s"], st["create_s"], st["loop_s"], st["teardown_s"], wall - st["process_s"]), flush=True)
