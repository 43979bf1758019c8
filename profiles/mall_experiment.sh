cd $GRAFT_REPO_ROOT
for flags in "" "-DGBP_FAC_TEMPORAL" "-DGBP_FAC_TEMPORAL -DGBP_LMSG_NT_STORE" "-DGBP_LMSG_NT_STORE"; do
  GBP_EXTRA_HIPFLAGS="$flags" python -m gbp_poplar_amd.build --force > /dev/null 2>&1
  for lm in 100000 60000; do
    python bench.py --cpu-seconds 0 --lmks $lm 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags=[$flags] lmks=$lm ms/step', d['ms_per_step'], 'sweep_us', d['roofline']['avg_launch_us'], 'beliefs_us', d['roofline']['belief_kernels_avg_us'])"
  done
done
python -m gbp_poplar_amd.build --force > /dev/null 2>&1
