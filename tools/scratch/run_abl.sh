cd /root/repo
L=clip_calibration_amd/csrc
python -m pytest tests/test_gpu_ops.py -q -k "attention" 2>&1 | tail -2
OPTION=attn_loader VALUES=1,2 python tools/block_ab.py 2>&1 | grep -v amdgpu.ids
echo "--- tuning build (plain stores in the GEMMs)"
CLIPMI_LIBRARY=$L/libclipmi_tuning.so OPTION=attn_loader VALUES=1,2 python tools/block_ab.py 2>&1 | grep -v amdgpu.ids
echo "--- GEMM output stores non-temporal"
CLIPMI_LIBRARY=$L/libclipmi_gemmnt.so OPTION=attn_loader VALUES=1,2 python tools/block_ab.py 2>&1 | grep -v amdgpu.ids
