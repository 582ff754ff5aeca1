export RTPE_LIBRARY=$PWD/realtime-pose-estimation_amd/librtpe_diag.so
mkdir -p gpurun_out/stag
for abl in 0 1024 2048 3072 4096 0; do
  RTPE_STREAM_ABL=$abl timeout -k 10 120 python tools/forward_profile.py 32 640 gpurun_out/stag/abl_$abl.txt > gpurun_out/stag/abl_$abl.log 2>&1 || exit 1
  echo "== ABL $abl"; grep -E "conv (96->96|192->192|384->384) k3s1|forward total" gpurun_out/stag/abl_$abl.txt
done
