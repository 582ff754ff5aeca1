export RTPE_LIBRARY=$PWD/realtime-pose-estimation_amd/librtpe_diag.so
mkdir -p gpurun_out/wt
for abl in 0 16 0 16; do
  RTPE_STREAM_ABL=$abl timeout -k 10 120 python tools/forward_profile.py 32 640 gpurun_out/wt/abl_$abl.txt > gpurun_out/wt/abl_$abl.log 2>&1 || exit 1
  echo "== ABL $abl"; grep -E "conv (96->96|192->192|384->384) k3s1|conv 48->(96|48) k3s2|forward total" gpurun_out/wt/abl_$abl.txt
done
