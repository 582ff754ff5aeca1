# A/B on one box with one binary: RTPE_PLANE_MAJOR=0 vs 1 (forward profile, batch 32 at 640x640)
for r in 1 2; do for v in 0 1; do
  RTPE_PLANE_MAJOR=$v timeout -k 10 300 python tools/forward_profile.py 32 640 gpurun_out/ab_${v}_$r.txt > /dev/null 2>&1 || exit 1
  echo "plane_major=$v run $r: $(tail -1 gpurun_out/ab_${v}_$r.txt)"
done; done
