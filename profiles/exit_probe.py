"""Process wall of `bin/ba fr1xyz` with and without the address-space priming of cli_common.hpp (prime_address_space): python profiles/exit_probe.py [--n_iters N]"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for label, extra, pre in (("GBP_CLI_NO_PRIME=1", {"GBP_CLI_NO_PRIME": "1"}, []), ("default (primed)", {}, []), ("primed, 8 threads on ONE cpu", {"GBP_CLI_PRIME_PIN": "1"}, []),
                          ("GBP_CLI_NO_PRIME=1", {"GBP_CLI_NO_PRIME": "1"}, []), ("default (primed)", {}, []), ("primed, 8 threads on ONE cpu", {"GBP_CLI_PRIME_PIN": "1"}, [])):
    for r in range(3):
        d = tempfile.mkdtemp()
        time.sleep(0.5)
        t0 = time.perf_counter()
        p = subprocess.run(pre + [os.path.join(ROOT, "gbp_poplar_amd", "bin", "ba"), "--bal_file", os.path.join(ROOT, "data", "sequences", "fr1xyz.txt"), "--profile", "1"] + sys.argv[1:],
                           env=dict(os.environ, GC_PROFILE_LOG_DIR=d, **extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        wall = time.perf_counter() - t0
        st = json.load(open(os.path.join(d, "gbp_profile.json")))["startup"]
        print("%-34s wall %.3f | main %.3f (create %.3f loop %.4f teardown %.4f) | after main %.3f" % (label, wall, st["process_s"], st["create_s"], st["loop_s"], st["teardown_s"], wall - st["process_s"]), flush=True)
