# A/B on one box with one binary: env switch (forward profile, batch 32 at 640x640)
for r in 1 2; do for v in 0 1; do
  RTPE_NT_STORE=$v timeout -k 10 300 python tools/forward_profile.py 32 640 gpurun_out/ab_${v}_$r.txt > /dev/null 2>&1 || exit 1
  echo "nt_store=$v run $r: $(grep stem gpurun_out/ab_${v}_$r.txt | cut -c1-120) | $(tail -1 gpurun_out/ab_${v}_$r.txt)"
done; done
