cd /root/repo
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r3_bench2.json
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python3 /root/repo/bench.py --steps 10 --warmup 3 > /tmp/prof.log 2>&1
cd /root/repo
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3_kernel_stats.csv \;
head -40 gpurun_out/r3_kernel_stats.csv | cut -c1-200
