cd /root/repo
python bench.py --workload coop_dac --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r3_coop_dac.json
python -c "
import json; d=json.load(open('gpurun_out/r3_coop_dac.json')); print(d['value'], json.dumps(d['coop_dac']['per_batch_text_tower'], indent=1))"
