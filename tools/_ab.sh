# A/B on one box: ab/head.so vs ab/new.so (forward profile, batch 32 at 640x640)
for r in 1 2; do for v in head new; do
  cp ab/$v.so realtime-pose-estimation_amd/librtpe_hip.so
  timeout -k 10 300 python tools/forward_profile.py 32 640 gpurun_out/ab_${v}_$r.txt > /dev/null 2>&1 || exit 1
  echo "$v $r: $(tail -1 gpurun_out/ab_${v}_$r.txt)"
done; done
