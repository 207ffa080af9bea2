#!/bin/bash
# Diagnostic: which power / clock files does this box expose to an ordinary user?  (bench.py's `power` object reads them.)
for d in /sys/class/drm/card*/device; do
  [ -d "$d/hwmon" ] || continue
  echo "== $d -> $(readlink -f $d)"
  for h in $d/hwmon/hwmon*; do
    echo "  $h: $(ls $h | tr '\n' ' ')"
    for f in name power1_average power1_input power1_cap power1_cap_max freq1_input freq1_label freq2_input freq2_label temp1_input; do
      [ -r "$h/$f" ] && echo "     $f = $(cat $h/$f 2>&1 | head -c 80)"
    done
  done
  [ -r "$d/pp_dpm_sclk" ] && echo "  pp_dpm_sclk: $(cat $d/pp_dpm_sclk | tr '\n' ' ')"
  [ -r "$d/gpu_metrics" ] && echo "  gpu_metrics: $(stat -c %s $d/gpu_metrics) bytes"
  [ -r "$d/unique_id" ] && echo "  unique_id: $(cat $d/unique_id)"
done
python3 - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print("torch props:", {k: getattr(p, k) for k in dir(p) if "pci" in k or k in ("name", "uuid", "multi_processor_count")})
PY
