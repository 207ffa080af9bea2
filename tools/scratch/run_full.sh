cd /root/repo
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r3_full5.log; cat gpurun_out/r3_full5.log
