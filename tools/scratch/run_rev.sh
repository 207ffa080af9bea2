cd /root/repo
L=clip_calibration_amd/csrc
for lib in libclipmi_tuning.so libclipmi_revgemm.so libclipmi_revattn.so libclipmi_tuning.so; do
  echo "--- $lib"
  CLIPMI_LIBRARY=$L/$lib OPTION=gemm_band VALUES=0 ROUNDS=5 python tools/block_ab.py 2>&1 | grep -v amdgpu.ids | tail -1
done
