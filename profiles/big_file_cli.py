#!/usr/bin/env python3
"""`bin/ba` on a file of a million factors: does bringing the HIP runtime up beside the file's parsing pay?

    python profiles/big_file_cli.py [runs=3] [n_cams=1000] [n_lmks=100000]

Writes the synthetic S1 graph (gbp_synth_generate, the bench's workload) as a text file in the reference's format under /tmp, then
runs `bin/ba --n_iters 10 --profile 1` on it three ways, alternating — one host thread and no overlap (GBP_HOST_THREADS=1, GBP_CLI_NO_WARMUP=1),
the file read by every core, and that plus the runtime coming up beside it (the default) — and prints each run's own phases."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gbp_poplar_amd import hostlib  # noqa: E402
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_cams = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n_lmks = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
tmp = tempfile.mkdtemp(prefix="gbp_big_")
path = os.path.join(tmp, "s1.txt")
t0 = time.perf_counter()
hostlib.bal_write(path, hostlib.synth_generate(n_cams, n_lmks))
print("file: %d cameras, %d landmarks, %.1f MB, written in %.2f s" % (n_cams, n_lmks, os.path.getsize(path) / 1e6, time.perf_counter() - t0), flush=True)
md5 = set()
for r in range(runs):
    for label, extra in (("r06-start", {"GBP_CLI_NO_WARMUP": "1", "GBP_HOST_THREADS": "1"}),      # one host thread reads the file, then the runtime comes up
                         ("threads", {"GBP_CLI_NO_WARMUP": "1"}),                                   # every host core reads its piece of the file
                         ("both", {})):                                                            # ... and the runtime comes up beside that
        d = tempfile.mkdtemp(dir=tmp)
        time.sleep(0.5)      # not on the heels of the previous process (its GPU context is torn down in the background: r06_cli_pause.txt)
        t0 = time.perf_counter()
        p = subprocess.run([os.path.join(ROOT, "gbp_poplar_amd", "bin", "ba"), "--bal_file", path, "--n_iters", "10", "--profile", "1"],
                           env=dict(os.environ, GC_PROFILE_LOG_DIR=d, **extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        wall = time.perf_counter() - t0
        if p.returncode != 0:
            sys.exit("ba failed: " + p.stderr[-2000:])
        st = json.load(open(os.path.join(d, "gbp_profile.json")))["startup"]
        md5.add("\n".join(l for l in p.stdout.splitlines() if "Total time" not in l and "Profile written" not in l))
        print("%-9s run %d: process wall %.3f s | loader %.3f, file %.3f, runtime (what was left to wait for) %.3f, create %.3f, upload %.3f, linearise + first metric %.3f, loop %.3f, teardown %.3f, after main %.3f"
              % (label, r, wall, st["exec_to_main_s"], st["file_parse_s"], st["runtime_init_s"], st["create_s"], st.get("upload_s", 0.0), st.get("linearise_s", 0.0), st["loop_s"],
                 st["teardown_s"], wall - st["process_s"]), flush=True)
print("stdout identical across all runs: %s" % (len(md5) == 1))
print("the last run's %s" % st.get("create", "")[6:])
p = subprocess.run([os.path.join(ROOT, "gbp_poplar_amd", "bin", "ba"), "--bal_file", path, "--n_iters", "1"], env=dict(os.environ, GBP_HOST_TRACE="1"),
                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
print("".join(l + "\n" for l in p.stderr.splitlines() if "read_number_file" in l or "load_problem" in l or "gbp_upload" in l), end="")
